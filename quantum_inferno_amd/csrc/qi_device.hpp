// Device-side helpers shared by the kernel files (wave = 64 lanes on gfx950).
#pragma once
#include "qi_common.hpp"

namespace qi {

constexpr int kWave = 64;

// Wave-wide reductions without the LDS crossbar: four DPP steps inside each row of sixteen lanes (quad permutes, half-row
// and row mirrors: afterwards every lane of a row holds the row's result), then the four rows by v_readlane.  The total is
// returned in EVERY lane; the order of the additions is fixed.  (The __shfl_down tree this replaces is twelve
// ds_bpermute_b32 for a double, each a round trip through the LDS pipe in a dependent chain -- once per band and wave in
// the block, zoom and two-pass kernels.)
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
template <int CTRL>
__device__ __forceinline__ double dpp_mov(double v) {
  const long long u = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_update_dpp(0, (int)(u & 0xffffffffll), CTRL, 0xF, 0xF, true);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(u >> 32), CTRL, 0xF, 0xF, true);
  return __longlong_as_double(((long long)hi << 32) | (long long)(unsigned)lo);
}
__device__ __forceinline__ float lane_get(float v, int lane) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}
__device__ __forceinline__ double lane_get(double v, int lane) {
  const long long u = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_readlane((int)(u & 0xffffffffll), lane);
  const int hi = __builtin_amdgcn_readlane((int)(u >> 32), lane);
  return __longlong_as_double(((long long)hi << 32) | (long long)(unsigned)lo);
}
#ifdef QI_WAVE_SHFL  // (A/B: the shuffle trees of rounds 1-2)
template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
  for (int o = kWave / 2; o > 0; o >>= 1) v += __shfl_down(v, o, kWave);
  return v;
}
template <typename T>
__device__ __forceinline__ T wave_max(T v) {
#pragma unroll
  for (int o = kWave / 2; o > 0; o >>= 1) {
    T w = __shfl_down(v, o, kWave);
    v = w > v ? w : v;
  }
  return v;
}
#else
// PRECONDITION of wave_sum / wave_max below: the WHOLE wave is active (call them after the loops have reconverged, never
// inside a divergent branch).  The row reductions run under EXEC, but the last step reads lanes 0, 16, 32 and 48 with
// v_readlane, which ignores EXEC: an inactive one of those lanes would hand back stale register contents.  -DQI_WAVE_SHFL
// selects the shuffle trees above, which only need lane 0; -DQI_NATIVE_DEBUG builds trap on a partial wave.
__device__ __forceinline__ void wave_full_check() {
#ifdef QI_NATIVE_DEBUG
  if (__builtin_amdgcn_read_exec() != ~0ull) __builtin_trap();
#endif
}
template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
  wave_full_check();
  v += dpp_mov<0xB1>(v);   // quad_perm [1, 0, 3, 2]
  v += dpp_mov<0x4E>(v);   // quad_perm [2, 3, 0, 1]
  v += dpp_mov<0x141>(v);  // row_half_mirror
  v += dpp_mov<0x140>(v);  // row_mirror
  return (lane_get(v, 0) + lane_get(v, 16)) + (lane_get(v, 32) + lane_get(v, 48));
}
template <typename T>
__device__ __forceinline__ T wave_max(T v) {
  wave_full_check();
  T w = dpp_mov<0xB1>(v);
  v = w > v ? w : v;
  w = dpp_mov<0x4E>(v);
  v = w > v ? w : v;
  w = dpp_mov<0x141>(v);
  v = w > v ? w : v;
  w = dpp_mov<0x140>(v);
  v = w > v ? w : v;
  const T a = lane_get(v, 0), b = lane_get(v, 16), c = lane_get(v, 32), d = lane_get(v, 48);
  const T ab = b > a ? b : a, cd = d > c ? d : c;
  return cd > ab ? cd : ab;
}
#endif

// Read-only global data through the constant address space: a load whose address is the same for the whole wave becomes a
// scalar load (s_load_dword*): its result lives in scalar registers and is an operand of the lanes' arithmetic.  For small
// wave-uniform tables only -- the scalar cache is no streaming path.
template <typename V>
using qi_cptr = const V __attribute__((address_space(4)))*;
template <typename V>
__device__ __forceinline__ qi_cptr<V> as_const(const V* p) {
  return reinterpret_cast<qi_cptr<V>>(reinterpret_cast<uintptr_t>(p));
}

// A struct at a wave-uniform address (a work item, a band descriptor) by scalar loads: its fields live in scalar registers,
// not in vector registers of every lane (a band descriptor fetched a band ahead was 24 vector registers of the block kernels).
// QI_NO_UNIFORM_LOADS: the plain (vector) loads, for A/B.
template <typename S>
__device__ __forceinline__ S load_uniform(const S* p) {
#ifdef QI_NO_UNIFORM_LOADS
  return *p;
#else
  static_assert(sizeof(S) % 4 == 0, "whole dwords");
  S out;
  const auto src = as_const(reinterpret_cast<const uint32_t*>(p));
  uint32_t* dst = reinterpret_cast<uint32_t*>(&out);
#pragma unroll
  for (size_t i = 0; i < sizeof(S) / 4; ++i) dst[i] = src[i];
  return out;
#endif
}

// streaming stores of panel data that is never read back by this launch (nontemporal: no allocation in the caches)
typedef float qi_f2 __attribute__((ext_vector_type(2)));
typedef float qi_f4 __attribute__((ext_vector_type(4)));
typedef double qi_d2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void stream_store(float2* p, float2 v) {
#ifdef QI_NO_STREAM_STORES
  *p = v;
#else
  __builtin_nontemporal_store((qi_f2){v.x, v.y}, reinterpret_cast<qi_f2*>(p));
#endif
}
__device__ __forceinline__ void stream_store(double2* p, double2 v) {
#ifdef QI_NO_STREAM_STORES
  *p = v;
#else
  __builtin_nontemporal_store((qi_d2){v.x, v.y}, reinterpret_cast<qi_d2*>(p));
#endif
}
__device__ __forceinline__ void stream_store(float4* p, float4 v) {
#ifdef QI_NO_STREAM_STORES
  *p = v;
#else
  __builtin_nontemporal_store((qi_f4){v.x, v.y, v.z, v.w}, reinterpret_cast<qi_f4*>(p));
#endif
}

// |z|^2 and scaled power in one fixed instruction form, so that every kernel variant (with or without the
// coefficient / bits stores) rounds the reductions identically
__device__ __forceinline__ float mul_rn(float a, float b) { return __fmul_rn(a, b); }
__device__ __forceinline__ double mul_rn(double a, double b) { return __dmul_rn(a, b); }
__device__ __forceinline__ float norm2(float x, float y) { return fmaf(x, x, __fmul_rn(y, y)); }
__device__ __forceinline__ double norm2(double x, double y) { return fma(x, x, __dmul_rn(y, y)); }

// complex product in one fixed instruction form (two multiplies, two fused multiply-adds)
__device__ __forceinline__ float2 cmul_rn(float2 a, float2 b) {
  return make_float2(fmaf(a.x, b.x, -__fmul_rn(a.y, b.y)), fmaf(a.x, b.y, __fmul_rn(a.y, b.x)));
}
__device__ __forceinline__ double2 cmul_rn(double2 a, double2 b) {
  return make_double2(fma(a.x, b.x, -__dmul_rn(a.y, b.y)), fma(a.x, b.y, __dmul_rn(a.y, b.x)));
}

__device__ __forceinline__ float fma_t(float a, float b, float c) { return fmaf(a, b, c); }
__device__ __forceinline__ double fma_t(double a, double b, double c) { return fma(a, b, c); }
__device__ __forceinline__ float log2_t(float v) { return log2f(v); }
__device__ __forceinline__ double log2_t(double v) { return log2(v); }
__device__ __forceinline__ float sqrt_t(float v) { return sqrtf(v); }
__device__ __forceinline__ double sqrt_t(double v) { return sqrt(v); }
// v_exp_f32 without the denormal-range fix-up of exp2f (results below 2^-126 are not needed where this is used)
__device__ __forceinline__ float fast_exp2(float v) { return __builtin_amdgcn_exp2f(v); }
__device__ __forceinline__ double fast_exp2(double v) { return exp2(v); }
__device__ __forceinline__ float exp2_t(float v) { return exp2f(v); }
__device__ __forceinline__ double exp2_t(double v) { return exp2(v); }

// Split of an atom that is longer than the record (styx bank: the reference truncates it at |x| = n / 2,
// styx_cwt.py:113-144): atom = taper * atom + (1 - taper) * atom.  The taper is 1 up to `e` samples before the
// truncation and falls to 0 (4e-11) at it as a Gaussian distribution function (+-6.5 standard deviations over `e`
// samples): the first part has a narrow spectrum (zoom engine), the second is two pieces of `e` taps at lags +-n / 2
// (k_block_edge).  x = sample position relative to the atom's centre.
__device__ __forceinline__ double split_taper(double x, int64_t n, double e) {
  const double r = (0.5 * (double)n - fabs(x)) / e;
  if (r >= 1.5) return 1.0;
  return 0.5 * erfc(-(r - 0.5) * (13.0 * 0.70710678118654752440));
}


}  // namespace qi
