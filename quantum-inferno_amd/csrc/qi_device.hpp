// Device-side helpers shared by the kernel files (wave = 64 lanes on gfx950).
#pragma once
#include "qi_common.hpp"

namespace qi {

constexpr int kWave = 64;

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

__device__ __forceinline__ float log2_t(float v) { return log2f(v); }
__device__ __forceinline__ double log2_t(double v) { return log2(v); }
__device__ __forceinline__ float sqrt_t(float v) { return sqrtf(v); }
__device__ __forceinline__ double sqrt_t(double v) { return sqrt(v); }
__device__ __forceinline__ float exp2_t(float v) { return exp2f(v); }
__device__ __forceinline__ double exp2_t(double v) { return exp2(v); }


}  // namespace qi
