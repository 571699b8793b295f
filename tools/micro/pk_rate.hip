// Micro-benchmark (round 4): issue rate of v_pk_fma_f32 against v_fma_f32 on gfx950 -- independent accumulator chains, 1 / 2 / 4
// waves per SIMD -- and of the two-instruction packed complex product (v_pk_mul_f32 + v_pk_fma_f32 with op_sel / neg_lo) against
// the four-instruction scalar one, with a bit-for-bit check of the packed form.
// hipcc -O3 -std=c++17 --offload-arch=gfx950 pk_rate.hip -o pk_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
typedef float v2f __attribute__((ext_vector_type(2)));
constexpr int NACC = 8;

template <int MODE>
__global__ void __launch_bounds__(256) k_rate(const float* __restrict__ in, float* __restrict__ out, int iters) {
  const float a0 = in[threadIdx.x], b0 = in[threadIdx.x + 256];
  if (MODE == 0) {  // scalar fma: NACC independent chains
    float acc[NACC];
    for (int j = 0; j < NACC; ++j) acc[j] = a0 + j;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int j = 0; j < NACC; ++j) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(acc[j]) : "v"(a0), "v"(b0));
    }
    float s = 0;
    for (int j = 0; j < NACC; ++j) s += acc[j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
  } else {  // packed fma: NACC independent chains of register pairs
    v2f acc[NACC];
    const v2f a = {a0, b0}, b = {b0, a0};
    for (int j = 0; j < NACC; ++j) acc[j] = (v2f){a0 + j, b0 - j};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int j = 0; j < NACC; ++j) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(acc[j]) : "v"(a), "v"(b));
    }
    float s = 0;
    for (int j = 0; j < NACC; ++j) s += acc[j].x + acc[j].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
  }
}

__device__ __forceinline__ v2f cmul_pk(v2f z, v2f w) {
  v2f t, r;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0]" : "=v"(t) : "v"(z), "v"(w));
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1] neg_lo:[0,0,1] neg_hi:[0,0,0]" : "=v"(r) : "v"(z), "v"(w), "v"(t));
  return r;
}
__device__ __forceinline__ v2f cmul_sc(v2f z, v2f w) {
  return (v2f){fmaf(z.x, w.x, -__fmul_rn(z.y, w.y)), fmaf(z.x, w.y, __fmul_rn(z.y, w.x))};
}
// x + (-i) y and x - (-i) y in one instruction each
__device__ __forceinline__ v2f add_mi(v2f x, v2f y) {
  v2f r;
  asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,0] neg_hi:[0,1]" : "=v"(r) : "v"(x), "v"(y));
  return r;
}
__device__ __forceinline__ v2f sub_mi(v2f x, v2f y) {
  v2f r;
  asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,0]" : "=v"(r) : "v"(x), "v"(y));
  return r;
}
template <int MODE>
__global__ void __launch_bounds__(256) k_cmul(const v2f* __restrict__ in, v2f* __restrict__ out, int iters) {
  v2f z[NACC];
  const v2f w = in[threadIdx.x + 256];
  for (int j = 0; j < NACC; ++j) z[j] = in[(threadIdx.x + 17 * j) & 255];
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < NACC; ++j) z[j] = MODE ? cmul_pk(z[j], w) : cmul_sc(z[j], w);
#pragma unroll
    for (int j = 0; j < NACC; j += 2) {  // a butterfly with a -i rotation on the second input
      const v2f a = z[j], b = z[j + 1];
      if (MODE) {
        z[j] = add_mi(a, b);
        z[j + 1] = sub_mi(a, b);
      } else {
        z[j] = (v2f){a.x + b.y, a.y - b.x};
        z[j + 1] = (v2f){a.x - b.y, a.y + b.x};
      }
    }
  }
  for (int j = 0; j < NACC; ++j) out[((size_t)blockIdx.x * NACC + j) * 256 + threadIdx.x] = z[j];
}

int main() {
  const int iters = 4000;
  float *in, *out;
  hipMalloc(&in, 512 * 8);
  hipMalloc(&out, (size_t)4096 * NACC * 256 * 8);
  std::vector<float> h(1024);
  for (int i = 0; i < 1024; ++i) h[i] = 0.999f * cosf(0.37f * i + 0.1f);
  for (int i = 512; i < 1024; i += 2) { float a = 0.37f * i; h[i] = cosf(a); h[i + 1] = sinf(a); }  // unit-modulus w
  hipMemcpy(in, h.data(), 1024 * 4, hipMemcpyHostToDevice);
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  for (int wps = 1; wps <= 4; wps *= 2) {  // waves per SIMD: 256 CUs x wps workgroups of 4 waves
    const int blocks = 256 * wps;
    float ms[2];
    for (int m = 0; m < 2; ++m) {
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(a);
        if (m == 0) k_rate<0><<<blocks, 256>>>(in, out, iters); else k_rate<1><<<blocks, 256>>>(in, out, iters);
        hipEventRecord(b);
        hipEventSynchronize(b);
        hipEventElapsedTime(&ms[m], a, b);
      }
    }
    // cycles per instruction and SIMD at 2.4 GHz: time * 2.4e6 / (iters * NACC * wps)
    printf("waves/SIMD %d: v_fma_f32 %.3f ms = %.2f cycles per instruction and SIMD; v_pk_fma_f32 %.3f ms = %.2f\n", wps, ms[0],
           ms[0] * 2.4e6 / (iters * NACC * wps), ms[1], ms[1] * 2.4e6 / (iters * NACC * wps));
  }
  // complex product + rotated butterfly: scalar (4 + 2 instructions per element) against packed (2 + 1)
  std::vector<float> r0((size_t)2 * NACC * 256 * 2), r1(r0.size());
  for (int wps = 1; wps <= 4; wps *= 2) {
    const int blocks = 256 * wps;
    float ms[2];
    for (int m = 0; m < 2; ++m)
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(a);
        if (m == 0) k_cmul<0><<<blocks, 256>>>((v2f*)in, (v2f*)out, 1000); else k_cmul<1><<<blocks, 256>>>((v2f*)in, (v2f*)out, 1000);
        hipEventRecord(b);
        hipEventSynchronize(b);
        hipEventElapsedTime(&ms[m], a, b);
      }
    printf("waves/SIMD %d: complex product + rotated butterfly, scalar %.3f ms, packed %.3f ms (x %.2f)\n", wps, ms[0], ms[1], ms[0] / ms[1]);
  }
  k_cmul<0><<<2, 256>>>((v2f*)in, (v2f*)out, 37);
  hipMemcpy(r0.data(), out, r0.size() * 4, hipMemcpyDeviceToHost);
  k_cmul<1><<<2, 256>>>((v2f*)in, (v2f*)out, 37);
  hipMemcpy(r1.data(), out, r1.size() * 4, hipMemcpyDeviceToHost);
  size_t bad = 0;
  for (size_t i = 0; i < r0.size(); ++i) bad += std::memcmp(&r0[i], &r1[i], 4) != 0;
  printf("packed against scalar after 37 iterations: %zu of %zu values differ (first %g %g)\n", bad, r0.size(), r0[0], r1[0]);
  return 0;
}
