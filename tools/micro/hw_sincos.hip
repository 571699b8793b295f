// Accuracy of v_sin_f32 / v_cos_f32 (argument in revolutions) against sincospi in double for the phases k / M the coarse
// stage's twiddles use:  hipcc -O3 --offload-arch=gfx950 tools/micro/hw_sincos.hip -o tools/micro/hw_sincos
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
__global__ void k(float* s, float* c, float* s2, float* c2, int M) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= M) return;
  const float x = (float)i / (float)M;
  s[i] = __builtin_amdgcn_sinf(x);
  c[i] = __builtin_amdgcn_cosf(x);
  sincospif(2.0f * x, &s2[i], &c2[i]);
}
int main() {
  for (int M : {256, 4096, 1 << 17, 1 << 21}) {
    float *s, *c, *s2, *c2;
    hipMalloc(&s, M * 4); hipMalloc(&c, M * 4); hipMalloc(&s2, M * 4); hipMalloc(&c2, M * 4);
    k<<<(M + 255) / 256, 256>>>(s, c, s2, c2, M);
    std::vector<float> hs(M), hc(M), hs2(M), hc2(M);
    hipMemcpy(hs.data(), s, M * 4, hipMemcpyDeviceToHost); hipMemcpy(hc.data(), c, M * 4, hipMemcpyDeviceToHost);
    hipMemcpy(hs2.data(), s2, M * 4, hipMemcpyDeviceToHost); hipMemcpy(hc2.data(), c2, M * 4, hipMemcpyDeviceToHost);
    double e_hw = 0, e_lib = 0;
    for (int i = 0; i < M; ++i) {
      const double ph = 2.0 * M_PI * (double)i / (double)M, rs = sin(ph), rc = cos(ph);
      e_hw = fmax(e_hw, fmax(fabs(hs[i] - rs), fabs(hc[i] - rc)));
      e_lib = fmax(e_lib, fmax(fabs(hs2[i] - rs), fabs(hc2[i] - rc)));
    }
    printf("M = %8d: max abs error v_sin/v_cos %.3e, sincospif %.3e\n", M, e_hw, e_lib);
    hipFree(s); hipFree(c); hipFree(s2); hipFree(c2);
  }
  return 0;
}
