// Micro-benchmark (round 4): the float64 fine stage's inner arithmetic -- a sliding FIR in double precision with the
// wave-uniform samples (a) in scalar registers (two v_readlane_b32 per double, v_fma_f64 with a scalar operand), (b) in vector
// registers (all-vector v_fma_f64) -- and the cost of v_readlane itself.
// hipcc -O3 -std=c++17 --offload-arch=gfx950 fir_sgpr64.hip -o fir_sgpr64
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int STEPS = 8;
__device__ __forceinline__ double lane_value64(double v, int lane) {
  const unsigned long long u = (unsigned long long)__double_as_longlong(v);
  const unsigned lo = __builtin_amdgcn_readlane((unsigned)u, lane), hi = __builtin_amdgcn_readlane((unsigned)(u >> 32), lane);
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
// MODE 0: scalar-register samples (readlane); 1: vector-register samples (broadcast once per iteration with shuffles, all-vector
// FMAs: the FMA rate alone); 2: readlanes only (the FMAs take a vector copy of the first sample)
template <int MODE, int TAPS>
__global__ void __launch_bounds__(256) k(const double* __restrict__ in, double* __restrict__ out, const double* __restrict__ w, int iters) {
  const int lane = threadIdx.x & 63;
  double wgt[TAPS];
  for (int j = 0; j < TAPS; ++j) wgt[j] = w[j * 64 + lane];
  double tot = 0.0;
  double smp = in[lane];
  for (int it = 0; it < iters; ++it) {
    double acc[STEPS];
    constexpr int WIN = STEPS + TAPS - 1;
    if (MODE == 0) {
      double sx[WIN];
#pragma unroll
      for (int i = 0; i < WIN; ++i) sx[i] = lane_value64(smp, i);
#pragma unroll
      for (int s = 0; s < STEPS; ++s) {
        double a = 0.0;
#pragma unroll
        for (int j = 0; j < TAPS; ++j) a = fma(wgt[j], sx[s + j], a);
        acc[s] = a;
      }
    } else if (MODE == 1) {
      double sx[WIN];
#pragma unroll
      for (int i = 0; i < WIN; ++i) sx[i] = smp + (double)i;  // vector values (one add each)
#pragma unroll
      for (int s = 0; s < STEPS; ++s) {
        double a = 0.0;
#pragma unroll
        for (int j = 0; j < TAPS; ++j) a = fma(wgt[j], sx[s + j], a);
        acc[s] = a;
      }
    } else {
      double sx[WIN];
#pragma unroll
      for (int i = 0; i < WIN; ++i) sx[i] = lane_value64(smp, i);
      double t = 0.0;
#pragma unroll
      for (int i = 0; i < WIN; ++i) t += sx[i];  // one vector add with a scalar operand per sample
#pragma unroll
      for (int s = 0; s < STEPS; ++s) acc[s] = t;
    }
#pragma unroll
    for (int s = 0; s < STEPS; ++s) tot += acc[s];
    smp = tot * 1e-30 + smp;  // (dependence across iterations: nothing is hoisted)
  }
  out[blockIdx.x * 256 + threadIdx.x] = tot;
}
template <int MODE, int TAPS>
void run(const char* name, const double* in, double* out, const double* w, hipEvent_t a, hipEvent_t b) {
  for (int wps = 1; wps <= 4; wps *= 2) {
    const int blocks = 256 * wps, iters = 2000;
    float ms = 0.f;
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(a); k<MODE, TAPS><<<blocks, 256>>>(in, out, w, iters); hipEventRecord(b); hipEventSynchronize(b);
      hipEventElapsedTime(&ms, a, b);
    }
    // per SIMD: wps waves, each `iters` iterations
    const double cyc_iter = ms * 1e-3 * 2.4e9 / ((double)iters * wps);
    const double fma = (double)STEPS * TAPS, rl = 2.0 * (STEPS + TAPS - 1);
    printf("%-28s taps %2d waves/SIMD %d: %8.3f ms, %7.1f SIMD cycles per wave-iteration (%3.0f fma + %3.0f readlane%s) -> %.2f cycles per fma\n",
           name, TAPS, wps, ms, cyc_iter, MODE == 2 ? 0.0 : fma, MODE == 1 ? 0.0 : rl, MODE == 2 ? " + adds" : "", MODE == 2 ? 0.0 : cyc_iter / fma);
  }
}
int main() {
  double *in, *out, *w;
  hipMalloc(&in, 4096 * 8); hipMalloc(&w, 64 * 16 * 8); hipMalloc(&out, 2048 * 256 * 8);
  hipMemset(in, 0, 4096 * 8); hipMemset(w, 0, 64 * 16 * 8);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  run<0, 16>("scalar-operand fma + readlane", in, out, w, a, b);
  run<1, 16>("vector-operand fma", in, out, w, a, b);
  run<2, 16>("readlane + scalar-operand add", in, out, w, a, b);
  run<0, 6>("scalar-operand fma + readlane", in, out, w, a, b);
  run<1, 6>("vector-operand fma", in, out, w, a, b);
  return 0;
}
