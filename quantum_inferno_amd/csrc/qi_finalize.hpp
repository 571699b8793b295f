// Fixed-order finalisation of the fused reductions, shared by k_finalize (qi_kernels.hip) and the native engine's
// tail kernel (qi_native.hip).
#pragma once
#include "qi_common.hpp"
#include "qi_device.hpp"

namespace qi {

// Workgroup (j, c) of 256 threads: j < B sums the partials of band j (power_band[c][j]); j == B reduces the stat
// entries (max, sum, sum p log2 p) of record c.  Fixed-order sums: the reductions are reproducible run to run (no
// float atomics anywhere on the path).  `s` = shared scratch [3][256 / kWave].
__device__ __forceinline__ void finalize_block(const double* __restrict__ part_band, const double* __restrict__ part_stat,
                                               double* __restrict__ power_band, double* __restrict__ stats, int64_t B,
                                               int64_t nblk, int64_t nstat, const int32_t* __restrict__ band_slots,
                                               int64_t j, int64_t c, double (*s)[256 / kWave]) {
  const int tid = threadIdx.x, lane = tid & (kWave - 1), wv = tid / kWave;
  double a0 = 0.0, a1 = 0.0, a2 = 0.0;
  if (j < B) {
    if (!part_band || !power_band) return;
    const double* p = part_band + (c * B + j) * nblk;
    const int64_t used = band_slots ? band_slots[j] : nblk;  // slots the band's engine wrote (the rest is never read)
    for (int64_t i = tid; i < used; i += 256) a1 += p[i];
  } else {
    if (!part_stat || !stats) return;
    const double* p = part_stat + c * nstat * 3;
    int64_t i = tid;
    for (; i + 3 * 256 < nstat; i += 4 * 256) {  // four independent entries in flight, summed in index order
      double m[4], s1[4], s2[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        m[u] = p[3 * (i + 256 * u)];
        s1[u] = p[3 * (i + 256 * u) + 1];
        s2[u] = p[3 * (i + 256 * u) + 2];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        a0 = m[u] > a0 ? m[u] : a0;
        a1 += s1[u];
        a2 += s2[u];
      }
    }
    for (; i < nstat; i += 256) {
      a0 = p[3 * i] > a0 ? p[3 * i] : a0;
      a1 += p[3 * i + 1];
      a2 += p[3 * i + 2];
    }
  }
  a0 = wave_max(a0);
  a1 = wave_sum(a1);
  a2 = wave_sum(a2);
  if (lane == 0) {
    s[0][wv] = a0;
    s[1][wv] = a1;
    s[2][wv] = a2;
  }
  __syncthreads();
  if (tid == 0) {
    double m = 0.0, s1 = 0.0, s2 = 0.0;
    for (int w = 0; w < 256 / kWave; ++w) {
      m = s[0][w] > m ? s[0][w] : m;
      s1 += s[1][w];
      s2 += s[2][w];
    }
    if (j < B) {
      power_band[c * B + j] = s1;
    } else {
      stats[c * 4 + 0] = m;
      stats[c * 4 + 1] = s1;
      stats[c * 4 + 2] = s2;
      stats[c * 4 + 3] = 0.0;
    }
  }
}

}  // namespace qi
