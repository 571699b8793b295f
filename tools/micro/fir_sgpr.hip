// Micro-benchmark: sliding FIR with wave-uniform samples in scalar registers (v_readlane) vs in vector registers.
// hipcc -O3 -std=c++17 --offload-arch=gfx950 fir_sgpr.hip -o fir_sgpr
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int STEPS = 16, TAPS = 37;
__device__ __forceinline__ float lane_value(float v, int lane) {
  return __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v), lane));
}
template <bool SGPR>
__global__ void __launch_bounds__(256) k(const float* __restrict__ in, float* __restrict__ out, const float* __restrict__ w, int iters) {
  const int lane = threadIdx.x & 63;
  float wgt[TAPS];
  for (int j = 0; j < TAPS; ++j) wgt[j] = w[j * 64 + lane];
  float tot = 0.f;
  for (int it = 0; it < iters; ++it) {
    const float smp = in[(it * 64 + lane) & 4095];
    float acc[STEPS];
    if (SGPR) {
      float sx[STEPS + TAPS - 1];
#pragma unroll
      for (int i = 0; i < TAPS - 1; ++i) sx[i] = lane_value(smp, i);
#pragma unroll
      for (int s = 0; s < STEPS; ++s) {
        sx[s + TAPS - 1] = lane_value(smp, s + TAPS - 1);
        float a = 0.f;
#pragma unroll
        for (int j = 0; j < TAPS; ++j) a = fmaf(wgt[j], sx[s + j], a);
        acc[s] = a;
      }
    } else {
      float sx[STEPS + TAPS - 1];
#pragma unroll
      for (int i = 0; i < STEPS + TAPS - 1; ++i) sx[i] = __shfl(smp, i, 64);
#pragma unroll
      for (int s = 0; s < STEPS; ++s) {
        float a = 0.f;
#pragma unroll
        for (int j = 0; j < TAPS; ++j) a = fmaf(wgt[j], sx[s + j], a);
        acc[s] = a;
      }
    }
#pragma unroll
    for (int s = 0; s < STEPS; ++s) tot += acc[s];
  }
  out[blockIdx.x * 256 + threadIdx.x] = tot;
}
int main() {
  float *in, *out, *w;
  hipMalloc(&in, 4096 * 4); hipMalloc(&w, 64 * TAPS * 4); hipMalloc(&out, 2048 * 256 * 4);
  hipMemset(in, 0, 4096 * 4); hipMemset(w, 0, 64 * TAPS * 4);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int waves_per_simd = 1; waves_per_simd <= 4; waves_per_simd *= 2) {
    const int blocks = 256 * waves_per_simd, iters = 400;
    for (int rep = 0; rep < 2; ++rep) {
      float ms;
      hipEventRecord(a); k<true><<<blocks, 256>>>(in, out, w, iters); hipEventRecord(b); hipEventSynchronize(b);
      hipEventElapsedTime(&ms, a, b);
      const double instr = (double)iters * STEPS * TAPS;  // fmacs per wave
      printf("waves/simd %d sgpr: %.3f ms  -> %.2f cycles per fmac per wave (2.4 GHz)\n", waves_per_simd, ms, ms * 1e-3 * 2.4e9 / instr);
      hipEventRecord(a); k<false><<<blocks, 256>>>(in, out, w, iters); hipEventRecord(b); hipEventSynchronize(b);
      hipEventElapsedTime(&ms, a, b);
      printf("waves/simd %d vgpr: %.3f ms  -> %.2f cycles per fmac per wave\n", waves_per_simd, ms, ms * 1e-3 * 2.4e9 / instr);
    }
  }
  return 0;
}
