// Micro-benchmark: register FFT16 + 15 twiddle multiplies per iteration, scalar f32 code vs packed (v_pk_*_f32) code.
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -fno-slp-vectorize -I ../../include -I ../../quantum_inferno_amd/csrc pk_fft.hip -o pk_fft
#include <hip/hip_runtime.h>
#include <cstdio>
#include <utility>
#include <vector>
#include "qi_fft_reg.hpp"
typedef float v2f __attribute__((ext_vector_type(2)));
using namespace qi; using namespace qi::native;
template <int K, int DIR>
__device__ __forceinline__ v2f mul_tw(v2f v) {
  if constexpr (K == 0) return v;
  else if constexpr (K == 16) return DIR > 0 ? (v2f){-v.y, v.x} : (v2f){v.y, -v.x};
  else {
    constexpr float c = (float)kCos64[K];
    constexpr float s = (float)(DIR * kSin64[K]);
    return __builtin_elementwise_fma(v.yx, (v2f){-s, s}, v * (v2f){c, c});
  }
}
template <int R, int S, int DIR, int I>
__device__ __forceinline__ void bflyp(v2f (&v)[R]) {
  constexpr int i = I % S;
  constexpr int p = (I / S) * 2 * S;
  const v2f a = v[p + i], b = v[p + i + S];
  v[p + i] = a + b;
  v[p + i + S] = mul_tw<i*(32 / S), DIR>(a - b);
}
template <int R, int S, int DIR, int... Is>
__device__ __forceinline__ void stagep(v2f (&v)[R], std::integer_sequence<int, Is...>) { (bflyp<R, S, DIR, Is>(v), ...); }
template <int R, int DIR, int S = R / 2>
__device__ __forceinline__ void fftp(v2f (&v)[R]) {
  stagep<R, S, DIR>(v, std::make_integer_sequence<int, R / 2>{});
  if constexpr (S > 1) fftp<R, DIR, S / 2>(v);
}
__global__ void __launch_bounds__(256) k_packed(const v2f* in, v2f* out, const v2f* tw, int iters) {
  v2f v[16], w[16];
  for (int b = 0; b < 16; ++b) { v[b] = in[threadIdx.x + 256 * b]; w[b] = tw[threadIdx.x + 256 * b]; }
  for (int it = 0; it < iters; ++it) {
    fftp<16, 1>(v);
#pragma unroll
    for (int b = 1; b < 16; ++b) { v2f x = v[b]; v[b] = __builtin_elementwise_fma(x.yx, (v2f){-w[b].y, w[b].y}, x * w[b].xx); }
  }
  for (int b = 0; b < 16; ++b) out[(size_t)blockIdx.x * 4096 + threadIdx.x + 256 * b] = v[b];
}
__global__ void __launch_bounds__(256) k_scalar(const float2* in, float2* out, const float2* tw, int iters) {
  float2 v[16], w[16];
  for (int b = 0; b < 16; ++b) { v[b] = in[threadIdx.x + 256 * b]; w[b] = tw[threadIdx.x + 256 * b]; }
  for (int it = 0; it < iters; ++it) {
    fft_reg<float, 16, 1>(v);
#pragma unroll
    for (int b = 1; b < 16; ++b) v[b] = cmul(v[b], w[b]);
  }
  for (int b = 0; b < 16; ++b) out[(size_t)blockIdx.x * 4096 + threadIdx.x + 256 * b] = v[b];
}
int main() {
  const int blocks = 256 * 8, iters = 2000;
  float2 *in, *out, *tw;
  hipMalloc(&in, 4096 * 8); hipMalloc(&tw, 4096 * 8); hipMalloc(&out, (size_t)blocks * 4096 * 8);
  std::vector<float2> h(4096);
  for (int i = 0; i < 4096; ++i) h[i] = make_float2(cosf(i * 0.37f) * 0.25f, sinf(i * 0.37f) * 0.25f);
  hipMemcpy(in, h.data(), 4096 * 8, hipMemcpyHostToDevice);
  hipMemcpy(tw, h.data(), 4096 * 8, hipMemcpyHostToDevice);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int rep = 0; rep < 2; ++rep) {
    float ms;
    hipEventRecord(a); k_scalar<<<blocks, 256>>>(in, out, tw, iters); hipEventRecord(b); hipEventSynchronize(b);
    hipEventElapsedTime(&ms, a, b); printf("scalar %.3f ms\n", ms);
    hipEventRecord(a); k_packed<<<blocks, 256>>>((v2f*)in, (v2f*)out, (v2f*)tw, iters); hipEventRecord(b); hipEventSynchronize(b);
    hipEventElapsedTime(&ms, a, b); printf("packed %.3f ms\n", ms);
  }
  // per wave-iteration cycles at 8 waves/CU... report instr-normalised: blocks*4 waves*iters / (1024 SIMDs)
  return 0;
}
