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
// log2 of a positive normal double for the entropy sums: p = 2^e m, m in [1, 2); the top seven mantissa bits pick the
// centre c of m's interval from a table of (1 / c rounded, -log2 of that rounded value), r = m / c - 1 (|r| <= 2^-8, one
// fma) and log2(1 + r) is a degree-5 series.  Absolute error below 1e-15 (checked against 120-bit arithmetic over
// 1e-300 .. 1e300); about a third of the instructions of the library log2, which was 39 % of the float64 zoom kernel.
__device__ __constant__ double kLog2Tab[128][2] = {
    {0x1.fe01fe01fe020p-1, 0x1.709c46d7aac60p-8},
    {0x1.fa11caa01fa12p-1, 0x1.1363117a97b03p-6},
    {0x1.f6310aca0dbb5p-1, 0x1.c9363ba850f9cp-6},
    {0x1.f25f644230ab5p-1, 0x1.3ed3094685a27p-5},
    {0x1.ee9c7f8458e02p-1, 0x1.985bfc3495193p-5},
    {0x1.eae807aba01ebp-1, 0x1.f13898332539dp-5},
    {0x1.e741aa59750e4p-1, 0x1.24b5b7e135a41p-4},
    {0x1.e3a9179dc1a73p-1, 0x1.507b836033bbap-4},
    {0x1.e01e01e01e01ep-1, 0x1.7beee96b8a281p-4},
    {0x1.dca01dca01dcap-1, 0x1.a7111df348494p-4},
    {0x1.d92f2231e7f8ap-1, 0x1.d1e34e35b82d7p-4},
    {0x1.d5cac807572b2p-1, 0x1.fc66a0f0b00a5p-4},
    {0x1.d272ca3fc5b1ap-1, 0x1.134e1b4890631p-3},
    {0x1.cf26e5c44bfc6p-1, 0x1.284294b07a640p-3},
    {0x1.cbe6d9601cbe7p-1, 0x1.3d1146d9a8a63p-3},
    {0x1.c8b265afb8a42p-1, 0x1.51bab907a5c8ap-3},
    {0x1.c5894d10d4986p-1, 0x1.663f6fac91315p-3},
    {0x1.c26b5392ea01cp-1, 0x1.7a9fec7d05de0p-3},
    {0x1.bf583ee868d8bp-1, 0x1.8edcae8352b6bp-3},
    {0x1.bc4fd65883e7bp-1, 0x1.a2f632320b86cp-3},
    {0x1.b951e2b18ff23p-1, 0x1.b6ecf175f95ecp-3},
    {0x1.b65e2e3beee05p-1, 0x1.cac163c770dcap-3},
    {0x1.b37484ad806cep-1, 0x1.de73fe3b1480ep-3},
    {0x1.b094b31d922a4p-1, 0x1.f205339208f27p-3},
    {0x1.adbe87f94905ep-1, 0x1.02baba24d0664p-2},
    {0x1.aaf1d2f87ebfdp-1, 0x1.0c62975542a8dp-2},
    {0x1.a82e65130e159p-1, 0x1.15fa676bb08fep-2},
    {0x1.a574107688a4ap-1, 0x1.1f825f6d88e13p-2},
    {0x1.a2c2a87c51ca0p-1, 0x1.28fab35b32684p-2},
    {0x1.a01a01a01a01ap-1, 0x1.32639636b2836p-2},
    {0x1.9d79f176b682dp-1, 0x1.3bbd3a0a1dcfbp-2},
    {0x1.9ae24ea5510dap-1, 0x1.4507cfedd4fc5p-2},
    {0x1.9852f0d8ec0ffp-1, 0x1.4e43880e8fb6bp-2},
    {0x1.95cbb0be377aep-1, 0x1.577091b3378c9p-2},
    {0x1.934c67f9b2ce6p-1, 0x1.608f1b42948aep-2},
    {0x1.90d4f120190d5p-1, 0x1.699f5248cd4b8p-2},
    {0x1.8e6527af1373fp-1, 0x1.72a1637cbc183p-2},
    {0x1.8bfce8062ff3ap-1, 0x1.7b957ac51aac4p-2},
    {0x1.899c0f601899cp-1, 0x1.847bc33d8618ep-2},
    {0x1.87427bcc092b9p-1, 0x1.8d54673b5c371p-2},
    {0x1.84f00c2780614p-1, 0x1.961f90527409bp-2},
    {0x1.82a4a0182a4a0p-1, 0x1.9edd6759b25e0p-2},
    {0x1.8060180601806p-1, 0x1.a78e146f7bef4p-2},
    {0x1.7e225515a4f1dp-1, 0x1.b031befe06435p-2},
    {0x1.7beb3922e017cp-1, 0x1.b8c88dbf88679p-2},
    {0x1.79baa6bb6398bp-1, 0x1.c152a6c24cae7p-2},
    {0x1.77908119ac60dp-1, 0x1.c9d02f6ca47b5p-2},
    {0x1.756cac201756dp-1, 0x1.d2414c80bf27cp-2},
    {0x1.734f0c541fe8dp-1, 0x1.daa6222064fb8p-2},
    {0x1.713786d9c7c09p-1, 0x1.e2fed3d097297p-2},
    {0x1.6f26016f26017p-1, 0x1.eb4b847d15bcep-2},
    {0x1.6d1a62681c861p-1, 0x1.f38c567bcc541p-2},
    {0x1.6b1490aa31a3dp-1, 0x1.fbc16b902680ap-2},
    {0x1.691473a88d0c0p-1, 0x1.01f57277264e0p-1},
    {0x1.6719f3601671ap-1, 0x1.0604719f24eb2p-1},
    {0x1.6524f853b4aa3p-1, 0x1.0a0dc34f8e1fcp-1},
    {0x1.63356b88ac0dep-1, 0x1.0e117754d7c11p-1},
    {0x1.614b36831ae94p-1, 0x1.120f9d39e1806p-1},
    {0x1.5f66434292dfcp-1, 0x1.160844495e006p-1},
    {0x1.5d867c3ece2a5p-1, 0x1.19fb7b8f32422p-1},
    {0x1.5babcc647fa91p-1, 0x1.1de951d9cbba7p-1},
    {0x1.59d61f123ccaap-1, 0x1.21d1d5bb6d59bp-1},
    {0x1.5805601580560p-1, 0x1.25b5158b73d05p-1},
    {0x1.56397ba7c52e2p-1, 0x1.29931f6791560p-1},
    {0x1.54725e6bb82fep-1, 0x1.2d6c013501380p-1},
    {0x1.52aff56a8054bp-1, 0x1.313fc8a1b36f2p-1},
    {0x1.50f22e111c4c5p-1, 0x1.350e8325707dap-1},
    {0x1.4f38f62dd4c9bp-1, 0x1.38d83e02f5d08p-1},
    {0x1.4d843bedc2c4cp-1, 0x1.3c9d06490ae11p-1},
    {0x1.4bd3edda68fe1p-1, 0x1.405ce8d38f4bcp-1},
    {0x1.4a27fad76014ap-1, 0x1.4417f24c82165p-1},
    {0x1.4880522014880p-1, 0x1.47ce2f2d02588p-1},
    {0x1.46dce34596066p-1, 0x1.4b7fabbe49796p-1},
    {0x1.453d9e2c776cap-1, 0x1.4f2c741a9f33ep-1},
    {0x1.43a2730abee4dp-1, 0x1.52d4942e4790ap-1},
    {0x1.420b5265e5951p-1, 0x1.567817b86b02dp-1},
    {0x1.40782d10e6566p-1, 0x1.5a170a4bf8d5cp-1},
    {0x1.3ee8f42a5af07p-1, 0x1.5db177508413cp-1},
    {0x1.3d5d991aa75c6p-1, 0x1.61476a031b108p-1},
    {0x1.3bd60d9232955p-1, 0x1.64d8ed7719beep-1},
    {0x1.3a524387ac822p-1, 0x1.68660c96f6f88p-1},
    {0x1.38d22d366088ep-1, 0x1.6beed2250cdadp-1},
    {0x1.3755bd1c945eep-1, 0x1.6f7348bc5c617p-1},
    {0x1.35dce5f9f2af8p-1, 0x1.72f37ad14c5b0p-1},
    {0x1.34679ace01346p-1, 0x1.766f72b263defp-1},
    {0x1.32f5ced6a1dfap-1, 0x1.79e73a8900620p-1},
    {0x1.3187758e9ebb6p-1, 0x1.7d5adc5a078a4p-1},
    {0x1.301c82ac40260p-1, 0x1.80ca620694df9p-1},
    {0x1.2eb4ea1fed14bp-1, 0x1.8435d54ca3774p-1},
    {0x1.2d50a012d50a0p-1, 0x1.879d3fc7b3b71p-1},
    {0x1.2bef98e5a3711p-1, 0x1.8b00aaf16d4a9p-1},
    {0x1.2a91c92f3c105p-1, 0x1.8e6020223d661p-1},
    {0x1.293725bb804a5p-1, 0x1.91bba891f1708p-1},
    {0x1.27dfa38a1ce4dp-1, 0x1.95134d584e2e4p-1},
    {0x1.268b37cd60127p-1, 0x1.9867176da382dp-1},
    {0x1.2539d7e9177b2p-1, 0x1.9bb70fab5ce4dp-1},
    {0x1.23eb79717605bp-1, 0x1.9f033ecc8e957p-1},
    {0x1.22a0122a0122ap-1, 0x1.a24bad6e7fb77p-1},
    {0x1.21579804855e6p-1, 0x1.a590641131564p-1},
    {0x1.2012012012012p-1, 0x1.a8d16b17e2745p-1},
    {0x1.1ecf43c7fb84cp-1, 0x1.ac0ecac99133cp-1},
    {0x1.1d8f5672e4abdp-1, 0x1.af488b51792d4p-1},
    {0x1.1c522fc1ce059p-1, 0x1.b27eb4bf8f08ap-1},
    {0x1.1b17c67f2bae3p-1, 0x1.b5b14f08f9665p-1},
    {0x1.19e0119e0119ep-1, 0x1.b8e0620887309p-1},
    {0x1.18ab083902bdbp-1, 0x1.bc0bf57f23605p-1},
    {0x1.1778a191bd684p-1, 0x1.bf341114464a7p-1},
    {0x1.1648d50fc3201p-1, 0x1.c258bc5664829p-1},
    {0x1.151b9a3fdd5c9p-1, 0x1.c579febb5b657p-1},
    {0x1.13f0e8d344724p-1, 0x1.c897dfa0db58ep-1},
    {0x1.12c8b89edc0acp-1, 0x1.cbb2664ccfcf5p-1},
    {0x1.11a3019a74826p-1, 0x1.cec999edc5204p-1},
    {0x1.107fbbe011080p-1, 0x1.d1dd819b4c3f0p-1},
    {0x1.0f5edfab325a2p-1, 0x1.d4ee24565c62ap-1},
    {0x1.0e40655826011p-1, 0x1.d7fb8909b2a6cp-1},
    {0x1.0d24456359e3ap-1, 0x1.db05b68a2fb64p-1},
    {0x1.0c0a7868b4171p-1, 0x1.de0cb397338a3p-1},
    {0x1.0af2f722eecb5p-1, 0x1.e11086daf7497p-1},
    {0x1.09ddba6af8360p-1, 0x1.e41136eae553dp-1},
    {0x1.08cabb37565e2p-1, 0x1.e70eca47ef86fp-1},
    {0x1.07b9f29b8eae2p-1, 0x1.ea09475ee3c39p-1},
    {0x1.06ab59c7912fbp-1, 0x1.ed00b488bec24p-1},
    {0x1.059eea0727586p-1, 0x1.eff5180afd3e5p-1},
    {0x1.04949cc1664c5p-1, 0x1.f2e67817eb846p-1},
    {0x1.038c6b78247fcp-1, 0x1.f5d4dacef36bep-1},
    {0x1.02864fc7729e9p-1, 0x1.f8c0463ce8c68p-1},
    {0x1.0182436517a37p-1, 0x1.fba8c05c544e0p-1},
    {0x1.0080402010080p-1, 0x1.fe8e4f15bd1a1p-1},
};
// (`tab`: kLog2Tab itself, or a copy of it in LDS -- a kernel that takes one logarithm per output cannot wait for a
// vector-memory round trip each time)
__device__ __forceinline__ double log2_pos(double p, const double (*tab)[2] = kLog2Tab) {
  const unsigned long long bits = (unsigned long long)__double_as_longlong(p);
  const int e = (int)((bits >> 52) & 0x7ffull) - 1023;
  const int idx = (int)((bits >> 45) & 127ull);
  const double m = __longlong_as_double((long long)((bits & 0x000fffffffffffffull) | 0x3ff0000000000000ull));
  const double r = fma(m, tab[idx][0], -1.0);
  const double q = r * fma(r, fma(r, fma(r, fma(r, 0.2, -0.25), 1.0 / 3.0), -0.5), 1.0);
  return (double)e + fma(q, 0x1.71547652b82fep+0, tab[idx][1]);
}
__device__ __forceinline__ float plog2p(float p, const double (*)[2]) { return plog2p(p); }
__device__ __forceinline__ double plog2p(double p, const double (*tab)[2]) {
  return p >= 2.2250738585072014e-308 ? p * log2_pos(p, tab) : 0.0;
}
__device__ __forceinline__ double plog2p(double p) { return p >= 2.2250738585072014e-308 ? p * log2_pos(p) : 0.0; }
// the same without a branch: p log2(max(p, smallest normal)) -- equal for every normal p and for p = 0 (0 x -1022 = 0); a
// denormal power contributes -1022 p instead of p log2 p, below 1e-304 either way
__device__ __forceinline__ double plog2p_flat(double p, const double (*tab)[2]) {
  return p * log2_pos(__builtin_fmax(p, 2.2250738585072014e-308), tab);
}

}  // namespace
}  // namespace native
}  // namespace qi
