// Register-resident FFT building blocks shared by the native engines (qi_native.hip, qi_block.hip).
#pragma once
#include <utility>

#include "qi_common.hpp"

#define QI_LAUNCH_CHECK()                                                                \
  do {                                                                                   \
    hipError_t e_ = hipGetLastError();                                                   \
    if (e_ != hipSuccess) {                                                              \
      set_error("%s:%d kernel launch -> %s", __FILE__, __LINE__, hipGetErrorString(e_)); \
      return QI_ERR_HIP;                                                                 \
    }                                                                                    \
  } while (0)

namespace qi {
namespace native {
namespace {

// cos / sin of 2 pi k / 64, exact at the quadrant points so that trivial twiddles fold away
constexpr double kCos64[33] = {1.0, 0.9951847266721969, 0.9807852804032304, 0.9569403357322088, 0.9238795325112867,
                               0.881921264348355, 0.8314696123025452, 0.773010453362737, 0.7071067811865476,
                               0.6343932841636455, 0.5555702330196023, 0.4713967368259978, 0.38268343236508984,
                               0.29028467725446233, 0.19509032201612833, 0.09801714032956077, 0.0,
                               -0.09801714032956065, -0.1950903220161282, -0.29028467725446216, -0.3826834323650897,
                               -0.4713967368259977, -0.555570233019602, -0.6343932841636454, -0.7071067811865475,
                               -0.773010453362737, -0.8314696123025453, -0.8819212643483549, -0.9238795325112867,
                               -0.9569403357322088, -0.9807852804032304, -0.9951847266721968, -1.0};
constexpr double kSin64[33] = {0.0, 0.0980171403295606, 0.19509032201612825, 0.29028467725446233, 0.3826834323650898,
                               0.47139673682599764, 0.5555702330196022, 0.6343932841636455, 0.7071067811865475,
                               0.773010453362737, 0.8314696123025452, 0.8819212643483549, 0.9238795325112867,
                               0.9569403357322089, 0.9807852804032304, 0.9951847266721968, 1.0, 0.9951847266721969,
                               0.9807852804032304, 0.9569403357322089, 0.9238795325112867, 0.881921264348355,
                               0.8314696123025455, 0.7730104533627371, 0.7071067811865476, 0.6343932841636455,
                               0.5555702330196022, 0.47139673682599786, 0.3826834323650899, 0.2902846772544624,
                               0.1950903220161286, 0.09801714032956083, 0.0};

// v * W_64^(DIR * K), K in [0, 32)
template <typename T, int K, int DIR>
__device__ __forceinline__ cplx<T> mul_tw64(cplx<T> v) {
  static_assert(K >= 0 && K < 32, "twiddle exponent");
  if constexpr (K == 0) {
    return v;
  } else if constexpr (K == 16) {
    return DIR > 0 ? mk<T>(-v.y, v.x) : mk<T>(v.y, -v.x);
  } else {
    constexpr T c = (T)kCos64[K];
    constexpr T s = (T)(DIR * kSin64[K]);
    return mk<T>(v.x * c - v.y * s, v.x * s + v.y * c);
  }
}

constexpr int brev(int x, int bits) {
  int r = 0;
  for (int i = 0; i < bits; ++i) r |= ((x >> i) & 1) << (bits - 1 - i);
  return r;
}
constexpr int ilog2(int x) { return x <= 1 ? 0 : 1 + ilog2(x / 2); }

// one radix-2 decimation-in-frequency butterfly of stage S (half span) on register array v[R]
template <typename T, int R, int S, int DIR, int I>
__device__ __forceinline__ void bfly(cplx<T> (&v)[R]) {
  constexpr int i = I % S;
  constexpr int p = (I / S) * 2 * S;
  const cplx<T> a = v[p + i], b = v[p + i + S];
  v[p + i] = mk<T>(a.x + b.x, a.y + b.y);
  v[p + i + S] = mul_tw64<T, i*(32 / S), DIR>(mk<T>(a.x - b.x, a.y - b.y));
}
template <typename T, int R, int S, int DIR, int... Is>
__device__ __forceinline__ void stage(cplx<T> (&v)[R], std::integer_sequence<int, Is...>) {
  (bfly<T, R, S, DIR, Is>(v), ...);
}
// In-register FFT of R points (R = 2^m <= 64); output index d ends up in v[brev(d)].
template <typename T, int R, int DIR, int S = R / 2>
__device__ __forceinline__ void fft_reg(cplx<T> (&v)[R]) {
  stage<T, R, S, DIR>(v, std::make_integer_sequence<int, R / 2>{});
  if constexpr (S > 1) fft_reg<T, R, DIR, S / 2>(v);
}

// v[brev(c)] *= W_64^c for c = 0..31 (the radix-2 combination twiddles of a 2048-point row)
template <typename T, int... Cs>
__device__ __forceinline__ void mul_w64_powers(cplx<T> (&v)[32], std::integer_sequence<int, Cs...>) {
  ((v[brev(Cs, 5)] = mul_tw64<T, Cs, 1>(v[brev(Cs, 5)])), ...);
}

// exp(+2 pi i m / Lf) for an exact integer phase m in [0, Lf), Lf = 2^p <= 2^24: the float argument
// 2 m / Lf is exact, so the seeds are accurate to single precision whatever the size of m
__device__ __forceinline__ void unit_root(uint32_t m, float two_over_len, double* c, double* s) {
  float sf, cf;
  sincospif((float)m * two_over_len, &sf, &cf);
  *c = cf;
  *s = sf;
}

// the same in the arithmetic of the engine's element type: float64 engines take the double-precision sincospi of the
// (exact) argument
template <typename T>
__device__ __forceinline__ void unit_root_t(uint32_t m, float two_over_len, double* c, double* s) {
  if constexpr (sizeof(T) == 8) {
    double sd, cd;
    sincospi((double)m * (double)two_over_len, &sd, &cd);
    *c = cd;
    *s = sd;
  } else {
    unit_root(m, two_over_len, c, s);
  }
}
// sin / cos of pi x in the precision of T (x is formed exactly by the callers)
template <typename T>
__device__ __forceinline__ void sincospi_as(double x, T* s, T* c) {
  if constexpr (sizeof(T) == 8) {
    double sd, cd;
    sincospi(x, &sd, &cd);
    *s = sd;
    *c = cd;
  } else {
    float sf, cf;
    sincospif((float)x, &sf, &cf);
    *s = sf;
    *c = cf;
  }
}

// running maximum of non-negative powers in one v_max (a compare-and-select takes two instructions)
__device__ __forceinline__ float max_t(float a, float b) { return __builtin_fmaxf(a, b); }
__device__ __forceinline__ double max_t(double a, double b) { return __builtin_fmax(a, b); }
__device__ __forceinline__ float plog2p(float p) { return p * __log2f(fmaxf(p, 1e-37f)); }
__device__ __forceinline__ double plog2p(double p) { return p > 0.0 ? p * log2(p) : 0.0; }

}  // namespace
}  // namespace native
}  // namespace qi
