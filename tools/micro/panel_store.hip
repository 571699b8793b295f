// What the chip sustains for the traffic pattern of the panel-writing kernels: per output sample one 8-byte coefficient
// into one array and one 4-byte value into another (the zoom / block epilogues), written once, never read back --
// against a float4 copy of the same number of bytes.  The HBM roofline of DESIGN.md prices every kernel against the
// 8 TB/s of the data sheet; this is the ceiling a kernel with NO arithmetic reaches on the same box.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/panel_store.hip -o tools/micro/panel_store && tools/micro/panel_store
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CK(x)                                                                   \
  do {                                                                          \
    hipError_t e = (x);                                                         \
    if (e != hipSuccess) {                                                      \
      printf("%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e));           \
      exit(1);                                                                  \
    }                                                                           \
  } while (0)

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

// rows of `n` samples (a band of a record), a workgroup writes `per` consecutive samples of one row per sweep:
// lane i of the workgroup holds the sample pair (2 i, 2 i + 1), as the engines' epilogues do
template <bool NT, bool BITS>
__global__ void __launch_bounds__(256) k_panel(f4* __restrict__ coef, f2* __restrict__ bits, int64_t pairs, float v) {
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < pairs; i += stride) {
    const f4 c = {v, v + 1.0f, v + 2.0f, (float)i};
    const f2 b = {v, (float)i};
    if (NT) {
      __builtin_nontemporal_store(c, coef + i);
      if (BITS) __builtin_nontemporal_store(b, bits + i);
    } else {
      coef[i] = c;
      if (BITS) bits[i] = b;
    }
  }
}

__global__ void __launch_bounds__(256) k_copy(const f4* __restrict__ src, f4* __restrict__ dst, int64_t n4) {
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) dst[i] = src[i];
}

__global__ void __launch_bounds__(256) k_read(const f4* __restrict__ src, float* __restrict__ out, int64_t n4) {
  const int64_t stride = (int64_t)gridDim.x * 256;
  f4 acc = {0, 0, 0, 0};
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) acc += src[i];
  if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = acc.x;
}

template <typename F>
static double time_ms(F launch, int reps) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) launch();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0, 0));
  for (int i = 0; i < reps; ++i) launch();
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / reps;
}

int main(int argc, char** argv) {
  const int64_t points = (argc > 1 ? atoll(argv[1]) : 2048) * (1ll << 20);  // output samples per launch (default 2^31)
  const int64_t pairs = points / 2;
  f4* coef;
  f2* bits;
  CK(hipMalloc((void**)&coef, pairs * sizeof(f4)));
  CK(hipMalloc((void**)&bits, pairs * sizeof(f2)));
  const int grids[] = {256 * 8, 256 * 16, 256 * 64};
  printf("%lld output samples per launch: %.2f GB coefficients + %.2f GB values\n", (long long)points, pairs * 16e-9, pairs * 8e-9);
  for (int g : grids) {
    const double a = time_ms([&] { k_panel<false, true><<<g, 256>>>(coef, bits, pairs, 1.0f); }, 5);
    const double b = time_ms([&] { k_panel<true, true><<<g, 256>>>(coef, bits, pairs, 1.0f); }, 5);
    const double c = time_ms([&] { k_panel<true, false><<<g, 256>>>(coef, bits, pairs, 1.0f); }, 5);
    printf("grid %6d: 8 + 4 bytes per sample, plain stores %.3f ms = %.0f GB/s | nontemporal %.3f ms = %.0f GB/s | coefficients only, nontemporal %.3f ms = %.0f GB/s\n",
           g, a, points * 12e-6 / a, b, points * 12e-6 / b, c, points * 8e-6 / c);
  }
  // copy and read of the same arrays (float4 lanes)
  const int64_t n4 = pairs / 2;
  for (int g : grids) {
    const double c = time_ms([&] { k_copy<<<g, 256>>>(coef, coef + n4, n4); }, 5);
    const double r = time_ms([&] { k_read<<<g, 256>>>(coef, (float*)bits, pairs); }, 5);
    printf("grid %6d: float4 copy %.3f ms = %.0f GB/s (read + write) | float4 read %.3f ms = %.0f GB/s\n", g, c, n4 * 32e-6 / c, r,
           pairs * 16e-6 / r);
  }
  return 0;
}
