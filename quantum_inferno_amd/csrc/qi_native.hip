// Native gfx950 FFT passes for the inverse transforms of the TFR panels.
//
// An inverse FFT of length Lf = N1 * N2 (N2 = 1024) is evaluated as out[t1 + N1 t2] =
//   sum_k2 A[t1][k2] W_N2^(k2 t2),   A[t1][k2] = sum_{k == k2 mod N2} Y[k] W_Lf^(k t1).
// One workgroup (1024 threads, 64 per row) owns G = 16 consecutive t1 (so every store is a run of 16 consecutive
// time samples), transforms the 16 rows of N2 points as 16 x 16 x 4 with one exchange through LDS, and applies the
// crop / power / entropy epilogue from registers.
//  * A band whose spectrum product Y has a short support (a Gaussian atom far from DC) gets A
//    straight from the spectrum: "pruned" loader, no intermediate in HBM at all.
//  * A wide band goes through pass 1 (the same row kernel over k1 for G1 consecutive k2) which writes
//    A once to an intermediate [k2][t1]; pass 2 reads it back transposed through LDS.
// Wave = 64 lanes; no MFMA (there is no dense contraction on this path).
#include <utility>

#include "qi_common.hpp"
#include "qi_device.hpp"
#include "qi_native.hpp"
#include "qi_fft_reg.hpp"
#include "qi_finalize.hpp"

namespace qi {
namespace native {

namespace {

// Geometry of one workgroup: G = 16 rows of 1024 points, 64 threads per row (1024 threads = 16 waves, <= 128 VGPRs:
// four waves per SIMD).  The 1024-point row transform is 16 x 16 x 4: step 1, thread (row g, a) transforms
// x[a + 64 b] over b; after the exchange through LDS, thread (d, c2, row g) -- wave = d, lanes = (c2, g) so that
// stores run over the 16 rows -- folds the radix-4 over q (a = a1 + 16 q) for its own c2 and transforms over a1.
template <typename T, int G_ = 16>
struct Cfg {
  static constexpr int NR = 1024, G = G_, TH = 64 * G_;
  static constexpr int SR = NR + 1;      // row stride of the natural-order image A[g][k]
  static constexpr int SA = 16 * G + 1;  // a-stride of the exchange image E[a][d][g]
  static constexpr int BUF = (G * SR > 64 * SA) ? G * SR : 64 * SA;
  static constexpr int TW1 = (kMaxPrunedTerms + 1) * G;  // inner twiddles of the pruned loader
  static constexpr int TW2 = 64;                         // W_64^(a1 c2)
  static constexpr size_t LDS_BYTES = ((size_t)BUF + NR + TW2 + TW1) * sizeof(cplx<T>) + 256;
};

#ifdef QI_NATIVE_DEBUG
#define QI_DBG(bit) (a.debug & (bit))
#else
#define QI_DBG(bit) false
#endif

// In-kernel phase stamps (diagnostic build only, -DQI_NATIVE_STAMPS): wave 0 of every workgroup sums the cycles it
// spends in each phase of the band loop; never enabled in the shipped library.
#ifdef QI_NATIVE_STAMPS
#define QI_STAMP(k)                                                        \
  do {                                                                     \
    __builtin_amdgcn_sched_barrier(0);                                     \
    const unsigned long long now_ = __builtin_amdgcn_s_memtime();          \
    __builtin_amdgcn_s_waitcnt(0xC07F);                                    \
    st_acc[k] += now_ - st_last;                                           \
    st_last = now_;                                                        \
    __builtin_amdgcn_sched_barrier(0);                                     \
  } while (0)
#else
#define QI_STAMP(k)
#endif

// ---- loaders: fill A[g][k] (row stride SR) for the G rows of this workgroup ---------------------------------------
// pruned: A[g][s] = W_Lf^(s t1_g) * sum_m Y[s + 1024 (k1_min + m)] W_N1^((k1_min + m) t1_g),  t1_g = t1_0 + g.
// The inner twiddles depend on (m, g) only: they are computed once per band into `tw1` (LDS, NTERM x G values)
// and broadcast; the outer twiddle is one float64 recurrence over g per slot from exact-phase seeds.  Each thread
// owns the slots tid and tid + TH/…; all global loads of a slot pair are issued before any arithmetic.
template <typename T, class C, bool STX>
__device__ __forceinline__ void load_pruned(cplx<T>* A, cplx<T>* tw1, const RowArgs<T>& a, const BandDesc& bd,
                                            const cplx<T>* __restrict__ X, uint32_t t1_0) {
  constexpr int MAXM = kMaxPrunedTerms + 1;  // distinct k1 = floor(k / 1024) a support of <= 8192 bins can touch
  constexpr int SPT = C::NR / C::TH;         // slots per thread (1)
  const int tid = threadIdx.x;
  const uint32_t mask = (uint32_t)a.Lf - 1u;  // Lf is a power of two: x mod Lf == x & mask, also for negative x
  const int32_t k_end = bd.k_lo + bd.k_len;
  const int32_t k1_min = bd.k_lo >> 10;            // arithmetic shift = floor for negative k_lo
  const int32_t nterm = ((k_end - 1) >> 10) - k1_min + 1;
  const cplx<T>* __restrict__ Hc = STX ? nullptr : a.Hc + bd.src_off;
  // inner twiddles W_Lf^(1024 k1 t1_g)
  if (tid < MAXM * C::G) {
    const int m = tid / C::G, g = tid % C::G;
    double wr, wi;
    unit_root_t<T>(((uint32_t)((k1_min + m) * 1024) * (t1_0 + (uint32_t)g)) & mask, a.two_over_len, &wr, &wi);
    tw1[tid] = mk<T>((T)wr, (T)wi);
  }
  __syncthreads();
  cplx<T> acc[SPT][C::G];
  bool has[SPT];
#pragma unroll
  for (int q = 0; q < SPT; ++q) {
    has[q] = false;
#pragma unroll
    for (int g = 0; g < C::G; ++g) acc[q][g] = mk<T>(T(0), T(0));
  }
#pragma unroll 1
  for (int m = 0; m < nterm; ++m) {
    cplx<T> y[SPT];
#pragma unroll
    for (int q = 0; q < SPT; ++q) {
      const int32_t k = tid + q * C::TH + 1024 * (k1_min + m);
      y[q] = mk<T>(T(0), T(0));
      if (k >= bd.k_lo && k < k_end) {
        has[q] = true;
        if constexpr (STX) {
          const cplx<T> x = X[(uint32_t)(k + (int32_t)bd.shift) & mask];
          const T e = (T)bd.coef * (T)k;
          const T w = exp2_t(-e * e) * a.inv_len;
          y[q] = mk<T>(x.x * w, x.y * w);
        } else {
          y[q] = cmul(X[(uint32_t)k & mask], Hc[k - bd.k_lo]);
        }
      }
    }
    bool any = false;
#pragma unroll
    for (int q = 0; q < SPT; ++q) any = any || (y[q].x != T(0) || y[q].y != T(0));
    if (any) {
#pragma unroll
      for (int g = 0; g < C::G; ++g) {
        const cplx<T> w = tw1[m * C::G + g];
#pragma unroll
        for (int q = 0; q < SPT; ++q) {
          acc[q][g].x += y[q].x * w.x - y[q].y * w.y;
          acc[q][g].y += y[q].x * w.y + y[q].y * w.x;
        }
      }
    }
  }
  // outer twiddle W_Lf^(s t1_g) and store; slots outside the support only store zeros.  The twiddles do not depend
  // on the band: t1 is made opaque so that they are recomputed where needed instead of being hoisted out of the band
  // loop for every slot (64 registers that would spill).
  __builtin_amdgcn_sched_barrier(0);
  uint32_t t1v = t1_0;
  asm volatile("" : "+v"(t1v));
#pragma unroll
  for (int q = 0; q < SPT; ++q) {
    const uint32_t sl = (uint32_t)(tid + q * C::TH);
    if (has[q]) {
      double wr, wi, sr, si;
      unit_root_t<T>((sl * t1v) & mask, a.two_over_len, &wr, &wi);
      unit_root_t<T>(sl & mask, a.two_over_len, &sr, &si);
#pragma unroll
      for (int g = 0; g < C::G; ++g) {
        const T cr = (T)wr, ci = (T)wi;
        A[g * C::SR + sl] = mk<T>(acc[q][g].x * cr - acc[q][g].y * ci, acc[q][g].x * ci + acc[q][g].y * cr);
        const double nr = wr * sr - wi * si;
        wi = wr * si + wi * sr;
        wr = nr;
      }
    } else {
#pragma unroll
      for (int g = 0; g < C::G; ++g) A[g * C::SR + sl] = mk<T>(T(0), T(0));
    }
  }
}

// general pass 2: the intermediate is stored transposed, imdT[r][k2] (one 8 KB row per time residue), so thread
// (row g, a) takes its 16 step-1 inputs k2 = a + 64 b straight into registers: 512-byte runs per row, no LDS image
template <typename T>
__device__ __forceinline__ void load_imd_direct(cplx<T> (&v)[16], const cplx<T>* __restrict__ imdT, uint32_t row, int a1) {
  const cplx<T>* __restrict__ src = imdT + (size_t)row * kN2 + a1;
#pragma unroll
  for (int b = 0; b < 16; ++b) v[b] = src[64 * b];
}

// pass-1 operand Y[k] of one band: spectrum x stored bank row (SRC 0), shifted spectrum x Gaussian (SRC 1), or the
// real record itself, zero beyond n (SRC 2: forward transform of the records)
template <typename T, int SRC>
__device__ __forceinline__ cplx<T> pass1_operand(const RowArgs<T>& a, const BandDesc& bd, const cplx<T>* __restrict__ X,
                                                 const cplx<T>* __restrict__ H, const T* __restrict__ sig, uint32_t k,
                                                 uint32_t mask) {
  if constexpr (SRC == 1) {
    const cplx<T> x = X[(k + (uint32_t)bd.shift) & mask];
    const int32_t ks = (k <= (mask >> 1)) ? (int32_t)k : (int32_t)k - (int32_t)(mask + 1u);
    const T e = (T)bd.coef * (T)ks;
    const T w = exp2_t(-e * e) * a.inv_len;
    return mk<T>(x.x * w, x.y * w);
  } else if constexpr (SRC == 0) {
    return cmul(X[k], H[k]);
  } else {
    return mk<T>(k < (uint32_t)a.n ? sig[k] : T(0), T(0));
  }
}

// pass 1 loader: rows are the G consecutive k2 = k2_0 + r, columns the 1024 inputs of phase `ph` of the transform
// over k1.  N1 = 1024: the column is Y[k2 + 1024 m].  N1 = 2048 (decimation in frequency): phase 0 transforms
// Y[m] + Y[m + 1024] (outputs t1 = 2 c), phase 1 (Y[m] - Y[m + 1024]) W_2048^m (outputs t1 = 2 c + 1); both phases
// read the operands again instead of holding a second result set in registers.
template <typename T, class C, int SRC, int NPH>
__device__ __forceinline__ void load_full(cplx<T>* A, const RowArgs<T>& a, const BandDesc& bd,
                                          const cplx<T>* __restrict__ X, const T* __restrict__ sig, uint32_t k2_0,
                                          int ph) {
  const int tid = threadIdx.x;
  const int r = tid % C::G;
  constexpr int STEP = C::TH / C::G, ITERS = C::NR / STEP, BATCH = NPH == 1 ? 8 : 4;
  static_assert(ITERS % BATCH == 0, "batch");
  const uint32_t mask = (uint32_t)a.Lf - 1u;
  const cplx<T>* __restrict__ H = SRC == 0 ? a.Hfull + (int64_t)bd.bank_row * a.Lf : nullptr;
#pragma unroll 1
  for (int it = 0; it < ITERS; it += BATCH) {
    cplx<T> ys[BATCH][NPH];
#pragma unroll
    for (int u = 0; u < BATCH; ++u) {
      const uint32_t m = (uint32_t)(tid / C::G + (it + u) * STEP);
#pragma unroll
      for (int h = 0; h < NPH; ++h)
        ys[u][h] = pass1_operand<T, SRC>(a, bd, X, H, sig, k2_0 + r + (uint32_t)kN2 * (m + 1024u * h), mask);
    }
#pragma unroll
    for (int u = 0; u < BATCH; ++u) {
      const uint32_t m = (uint32_t)(tid / C::G + (it + u) * STEP);
      cplx<T> y = ys[u][0];
      if constexpr (NPH == 2) {
        if (ph == 0) {
          y = mk<T>(ys[u][0].x + ys[u][1].x, ys[u][0].y + ys[u][1].y);
        } else {
          T sn, cs;
          sincospi_as<T>((double)m * (1.0 / 1024.0), &sn, &cs);  // W_2048^m
          y = cmul(mk<T>(ys[u][0].x - ys[u][1].x, ys[u][0].y - ys[u][1].y), mk<T>(cs, sn));
        }
      }
      A[r * C::SR + m] = y;
    }
  }
}

// tw[d * 64 + a] = W_1024^(a d) (between step 1 and step 2, lanes run over a) and tw[1024 + c2 * 16 + a1] =
// W_64^(a1 c2) (inside step 2)
template <typename T, class C>
__device__ __forceinline__ void fill_step_twiddles(cplx<T>* tw) {
  for (int i = threadIdx.x; i < C::NR + C::TW2; i += C::TH) {
    T sf, cf;
    if (i < C::NR) {
      const int d = i / 64, aa = i % 64;
      sincospi_as<T>((double)(2 * (aa * d)) / (double)C::NR, &sf, &cf);
    } else {
      const int c2 = (i - C::NR) / 16, a1 = (i - C::NR) % 16;
      sincospi_as<T>((double)(2 * (a1 * c2)) / 64.0, &sf, &cf);
    }
    tw[i] = mk<T>(cf, sf);
  }
}

// 1024-point inverse transform of G rows, all threads of the workgroup.  Thread (g1 = tid / 64, a1 = tid % 64) enters
// with its 16 step-1 inputs x[a1 + 64 b] of row g1 in v; on return thread (d2 = tid / 64, c2 = (tid % 64) / 16,
// g2 = tid % 16) holds out[d2 + 16 c2 + 64 c1] of row g2 in u[brev(c1, 4)] and buf is free.  `between` runs on every
// thread right after the first barrier (flush of a pending reduction), `before_step2` after the last one.
template <typename T, class C, class F, class F2>
__device__ __forceinline__ void rows_fft1024_regs(cplx<T> (&v)[16], cplx<T>* buf, const cplx<T>* tw, cplx<T> (&u)[16],
                                                  bool skip, F between, F2 before_step2
#ifdef QI_NATIVE_STAMPS
                                                  , unsigned long long (&st_acc)[8], unsigned long long& st_last
#endif
) {
  const int tid = threadIdx.x;
  const int g1 = tid / 64, a1 = tid % 64;
  const int d2 = tid / (4 * C::G), c2 = (tid % (4 * C::G)) / C::G, g2 = tid % C::G;
  if (!skip) {
    fft_reg<T, 16, 1>(v);
#pragma unroll
    for (int d = 1; d < 16; ++d) v[brev(d, 4)] = cmul(v[brev(d, 4)], tw[d * 64 + a1]);
  }
  __builtin_amdgcn_sched_barrier(0);
  QI_STAMP(2);
  __syncthreads();  // every thread has taken its inputs (from the image or earlier), buf may be overwritten
  QI_STAMP(3);
  between();
#pragma unroll
  for (int d = 0; d < 16; ++d) buf[a1 * C::SA + d * C::G + g1] = v[brev(d, 4)];
  __builtin_amdgcn_sched_barrier(0);
  __syncthreads();
  // step 2: X[d + 16 (c2 + 4 c1)] = sum_a1 W_16^(a1 c1) W_64^(a1 c2) sum_q T'[a1 + 16 q][d] W_4^(q c2)
  const T sg = (c2 & 1) ? T(-1) : T(1), tau = (c2 & 2) ? T(-1) : T(1);
  const bool odd = c2 & 1;
  const cplx<T>* __restrict__ col = buf + d2 * C::G + g2;
#pragma unroll
  for (int a = 0; a < 16; ++a) {
    const cplx<T> t0 = col[a * C::SA], t1 = col[(a + 16) * C::SA], t2 = col[(a + 32) * C::SA], t3 = col[(a + 48) * C::SA];
    const cplx<T> u02 = mk<T>(t0.x + sg * t2.x, t0.y + sg * t2.y), u13 = mk<T>(t1.x + sg * t3.x, t1.y + sg * t3.y);
    const cplx<T> w13 = odd ? mk<T>(-u13.y, u13.x) : u13;  // times i for the odd residues
    u[a] = mk<T>(u02.x + tau * w13.x, u02.y + tau * w13.y);
  }
  __builtin_amdgcn_sched_barrier(0);
  __syncthreads();
  QI_STAMP(4);
  before_step2();
  if (!skip) {
#pragma unroll
    for (int a = 1; a < 16; ++a) u[a] = cmul(u[a], tw[C::NR + c2 * 16 + a]);
    fft_reg<T, 16, 1>(u);
  }
  __builtin_amdgcn_sched_barrier(0);
  QI_STAMP(5);
}

// same, starting from the natural-order LDS image A[g][k] (row stride SR) the caller filled and synchronised
template <typename T, class C>
__device__ __forceinline__ void rows_fft1024(cplx<T>* buf, const cplx<T>* tw, cplx<T> (&u)[16], bool skip
#ifdef QI_NATIVE_STAMPS
                                             , unsigned long long (&st_acc)[8], unsigned long long& st_last
#endif
) {
  const int g1 = threadIdx.x / 64, a1 = threadIdx.x % 64;
  cplx<T> v[16];
#pragma unroll
  for (int b = 0; b < 16; ++b) v[b] = buf[g1 * C::SR + a1 + 64 * b];
#ifdef QI_NATIVE_STAMPS
  rows_fft1024_regs<T, C>(v, buf, tw, u, skip, [] {}, [] {}, st_acc, st_last);
#else
  rows_fft1024_regs<T, C>(v, buf, tw, u, skip, [] {}, [] {});
#endif
}

// ---- pass 1 (wide bands, and the forward transform of the records) -------------------------------------------------
// Rows are G consecutive k2; the transform runs over k1 (N1 = 1024 NPH points) and the result, multiplied by the
// pass twiddle W_Lf^(k2 t1), is written transposed, imdT[t1][k2], in runs of G consecutive k2 (128 bytes).
template <typename T, class C, int SRC, int NPH>
__global__ void __launch_bounds__(C::TH) k_pass1(RowArgs<T> a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  cplx<T>* buf = reinterpret_cast<cplx<T>*>(smem);
  cplx<T>* tw = buf + C::BUF;
  const int tid = threadIdx.x;
  const int64_t ch = blockIdx.z;
  // with phase_split the decimation-in-frequency phases of a 2048-point pass are separate workgroups (blockIdx.y =
  // band * NPH + phase): twice the workgroups for launches that would not cover the chip
  const int split = (NPH > 1 && a.phase_split) ? NPH : 1;
  uint32_t bx = blockIdx.x, by = blockIdx.y;
  if (SRC == 2 && gridDim.y == 1 && (gridDim.x & 7) == 0) {
    bx = (bx & 7u) * (gridDim.x >> 3) + (bx >> 3);  // (one workgroup per row group: see below)
  } else if (SRC == 2 && NPH == 2 && split == 2 && gridDim.y == 2 && (gridDim.x & 7) == 0) {
    // forward transform of real records: a row group reads G floats (64 bytes at G = 16) of every 4 KB of the record, and
    // both phases read the same ones.  Consecutive workgroups go to consecutive XCDs (eight L2s): every XCD takes a
    // contiguous range of row groups, the two phases of a group back to back, so that the other half of a 128-byte line
    // and the second phase's reads are served by the same L2 (4.0 x the record's bytes were fetched before)
    const uint32_t lin = bx + gridDim.x * by, per = gridDim.x >> 3;
    bx = (lin & 7u) * per + ((lin >> 3) >> 1);
    by = (lin >> 3) & 1u;
  }
  const uint32_t row0 = bx * C::G;
  const int d2 = tid / (4 * C::G), c2 = (tid % (4 * C::G)) / C::G, g2 = tid % C::G;  // lanes run over the G rows
  fill_step_twiddles<T, C>(tw);
  const int ph_first = split > 1 ? (int)(by % NPH) : 0, ph_last = split > 1 ? ph_first + 1 : NPH;
  BandDesc bd{};
  if constexpr (SRC != 2) bd = a.bands[a.gen_list[by / split]];
  const cplx<T>* Xc = SRC == 2 ? nullptr : a.X + ch * a.Lf;
  const T* sigc = SRC == 2 ? a.sig + ch * a.n : nullptr;
  const uint32_t mask = (uint32_t)a.Lf - 1u;
  const uint32_t k2 = row0 + g2;
  // transposed intermediate imdT[r][k2]: one row of 1024 values per time residue.  The linear kind's pass 2 works
  // on residues t1 = r - 1 (r = 0 is t1 = -1 == N1 - 1 with the pass twiddle taken at -1): columns are stored at
  // r = (t1 + 1) mod N1 so that pass 2 reads whole rows.
  cplx<T>* __restrict__ dst = a.imd + ((int64_t)ch * a.imd_slots + bd.gen_slot) * a.Lf + k2;
  const uint32_t roll = a.neg_last_row ? 1u : 0u, cmask = (uint32_t)a.N1 - 1u;
#ifdef QI_NATIVE_STAMPS
  unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long st_last = __builtin_amdgcn_s_memtime();
#endif
#pragma unroll 1
  for (int ph = ph_first; ph < ph_last; ++ph) {
    if (!QI_DBG(2)) load_full<T, C, SRC, NPH>(buf, a, bd, Xc, sigc, row0, ph);
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    cplx<T> u[16];
#ifdef QI_NATIVE_STAMPS
    rows_fft1024<T, C>(buf, tw, u, QI_DBG(4), st_acc, st_last);
#else
    rows_fft1024<T, C>(buf, tw, u, QI_DBG(4));
#endif
    // pass twiddle W_Lf^(k2 t1) along t1 = NPH (d2 + 16 c2 + 64 c1) + ph by a float64 recurrence over c1
    double wr, wi, sr, si;
    unit_root_t<T>((k2 * (uint32_t)(NPH * (d2 + 16 * c2) + ph)) & mask, a.two_over_len, &wr, &wi);
    unit_root_t<T>((k2 * (uint32_t)(64 * NPH)) & mask, a.two_over_len, &sr, &si);
#pragma unroll
    for (int c1 = 0; c1 < 16; ++c1) {
      const cplx<T> z = u[brev(c1, 4)];
      const uint32_t t1 = (uint32_t)(NPH * (d2 + 16 * c2 + 64 * c1)) + (uint32_t)ph;
      if (c1 == 15 && a.neg_last_row && t1 == cmask)  // t1 = N1 - 1 is used by pass 2 as t1 = -1
        unit_root_t<T>((0u - k2) & mask, a.two_over_len, &wr, &wi);
      const T cr = (T)wr, ci = (T)wi;
      if (!QI_DBG(1)) dst[(size_t)((t1 + roll) & cmask) * kN2] = mk<T>(z.x * cr - z.y * ci, z.x * ci + z.y * cr);
      const double nr = wr * sr - wi * si;
      wi = wr * si + wi * sr;
      wr = nr;
    }
  }
}

// ---- forward transform of the records, second pass -----------------------------------------------------------------
// X = conj(IDFT(x)) for a real record x: pass 1 (SRC 2) ran the inverse machinery on the record; here rows are G
// consecutive frequency residues f1, the transform runs over t2 and X[f1 + N1 f2] = conj(result) is written in
// natural order (runs of G consecutive bins).
template <typename T, class C>
__global__ void __launch_bounds__(C::TH) k_fwd2(RowArgs<T> a, cplx<T>* __restrict__ Xout) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  cplx<T>* buf = reinterpret_cast<cplx<T>*>(smem);
  cplx<T>* tw = buf + C::BUF;
  const int tid = threadIdx.x;
  const int64_t ch = blockIdx.z;
  const uint32_t row0 = blockIdx.x * C::G;
  const int d2 = tid / (4 * C::G), c2 = (tid % (4 * C::G)) / C::G, g2 = tid % C::G;
  fill_step_twiddles<T, C>(tw);
  cplx<T> v[16], u[16];
  load_imd_direct<T>(v, a.imd + (int64_t)ch * a.imd_slots * a.Lf, row0 + tid / 64, tid % 64);
  __syncthreads();  // step twiddles ready
#ifdef QI_NATIVE_STAMPS
  unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long st_last = 0;
  rows_fft1024_regs<T, C>(v, buf, tw, u, false, [] {}, [] {}, st_acc, st_last);
#else
  rows_fft1024_regs<T, C>(v, buf, tw, u, false, [] {}, [] {});
#endif
  cplx<T>* __restrict__ dst = Xout + ch * a.Lf + row0 + g2 + (uint32_t)a.N1 * (d2 + 16 * c2);
  const uint32_t tstep = 64u * (uint32_t)a.N1;
#pragma unroll
  for (int c1 = 0; c1 < 16; ++c1) {
    const cplx<T> z = u[brev(c1, 4)];
    dst[(size_t)c1 * tstep] = mk<T>(z.x, -z.y);
  }
}

// ---- pass 2 (every band) ---------------------------------------------------------------------------------------------
// Rows are G consecutive time residues t1; the transform runs over k2 and the epilogue crops / rolls into the
// panel and takes the tfr_info reductions from registers.
// KIND: 0 zero-padded linear correlation (Lf = 2n, keep [n/2 - 1, n/2 - 1 + n)), 1 circular correlation rolled by
// n/2 (Lf = n), 2 Stockwell (Lf = n).  With t = t1 + N1 (d + 16 c2 + 64 c1) the crop / roll is a compile-time map
// of c1:  KIND 2: panel position i = c1;  KIND 1: i = (c1 + 8) mod 16;  KIND 0: i = c1 - 4 for c1 in [4, 12) (8 of
// the 16 outputs of a thread are kept, the others are never computed: the dead butterflies are eliminated).
template <typename T, class C, int KIND, bool COEF, bool BITS>
__global__ void __launch_bounds__(C::TH) k_pass2(RowArgs<T> a) {
  constexpr bool STX = KIND == 2;
  constexpr int NOUT = KIND == 0 ? 8 : 16;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  cplx<T>* buf = reinterpret_cast<cplx<T>*>(smem);
  cplx<T>* tw = buf + C::BUF;
  __shared__ double s_red[2][C::TH / kWave];
  __shared__ double s_fin[3][C::TH / kWave];
  __shared__ double ltab[sizeof(T) == 8 ? 128 : 1][2];  // float64: the log2 table of the entropy sums in LDS (log2_pos)
  const int tid = threadIdx.x, lane = tid & (kWave - 1), wv = tid / kWave;
  if (sizeof(T) == 8 && tid < 128) {
    ltab[tid][0] = kLog2Tab[tid][0];
    ltab[tid][1] = kLog2Tab[tid][1];
  }
  const int64_t grp = blockIdx.x, ch = blockIdx.z;
  const uint32_t row0 = (uint32_t)grp * C::G;
  const int d2 = tid / (4 * C::G), c2 = (tid % (4 * C::G)) / C::G, g2 = tid % C::G;
  fill_step_twiddles<T, C>(tw);
  __syncthreads();

  T col[NOUT];
#pragma unroll
  for (int c = 0; c < NOUT; ++c) col[c] = T(0);
  T mx = T(0);
  double plogp = 0.0;
  // The linear kind works on the time residues t1 = r - 1, r = row0 + g (so r = 0 is t1 = -1, i.e. the samples
  // N1 t2 - 1): the kept samples [n/2 - 1, 3n/2 - 1) are then exactly t2 in [256, 768), i.e. c1 in [4, 12), for every
  // row, and each run of G outputs starts on a 128-byte boundary of the panel.
  const uint32_t t1_first = row0 - (KIND == 0 ? 1u : 0u);
  // panel offset of this thread's first output and the stride between its outputs
  const uint32_t tbase = row0 + g2 + (uint32_t)a.N1 * (d2 + 16 * c2);
  const uint32_t tstep = 64u * (uint32_t)a.N1;
  const cplx<T>* Xc = a.X + ch * a.Lf;
  int64_t pending = -1;  // band whose row sum sits in s_red waiting for a barrier
  int par = 0;

#ifdef QI_NATIVE_STAMPS
  unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long st_last = __builtin_amdgcn_s_memtime();
#endif
  // list entries blockIdx.y, blockIdx.y + nchunk, ...: every chunk gets the same mix of narrow and wide bands
  for (int64_t jj = blockIdx.y; jj < a.nbands; jj += gridDim.y) {
    const BandDesc bd = a.bands[jj];
    const int64_t j = bd.out_band;  // row of the panel this band writes
    QI_STAMP(7);
    cplx<T> v[16], u[16];
    auto flush = [&] {
      if (pending >= 0 && tid == 0) {
        double s = 0.0;
        for (int w = 0; w < C::TH / kWave; ++w) s += s_red[par ^ 1][w];
        a.part_band[((int64_t)ch * a.panel_bands + pending) * a.nblk + grp] = s;
      }
    };
    if (bd.mode == 0) {
      if (!QI_DBG(2)) load_pruned<T, C, STX>(buf, tw + C::NR + C::TW2, a, bd, Xc, t1_first);
      __builtin_amdgcn_sched_barrier(0);
      QI_STAMP(0);
      __syncthreads();
      QI_STAMP(1);
#pragma unroll
      for (int b = 0; b < 16; ++b) v[b] = buf[(tid / 64) * C::SR + (tid % 64) + 64 * b];
    } else {
      if (!QI_DBG(2))
        load_imd_direct<T>(v, a.imd + ((int64_t)ch * a.imd_slots + bd.gen_slot) * a.Lf, row0 + tid / 64, tid % 64);
      QI_STAMP(0);
    }
#ifdef QI_NATIVE_STAMPS
    rows_fft1024_regs<T, C>(v, buf, tw, u, QI_DBG(4), flush, [] {}, st_acc, st_last);
#else
    rows_fft1024_regs<T, C>(v, buf, tw, u, QI_DBG(4), flush, [] {});
#endif

    const int64_t orow = ((int64_t)ch * a.panel_bands + j) * a.n;
    char* __restrict__ coef_row = reinterpret_cast<char*>(a.coef ? a.coef + orow : nullptr);
    char* __restrict__ bits_row = reinterpret_cast<char*>(a.bits ? a.bits + orow : nullptr);
    // the output offsets do not depend on the band; hide that from the optimiser, which would otherwise hoist
    // every address out of the band loop and spill them
    uint32_t tb = tbase;
    asm volatile("" : "+v"(tb));
    T rowacc = T(0), pl = T(0);
    if (KIND == 1 && !COEF && bd.edge > 0 && a.edge_z) {
      // no panel to correct in place: hand the circular values of the edge samples (panel positions i = 0 and 15,
      // edge <= n / 16) to k_edge_fix through edge_z
      cplx<T>* ez = a.edge_z + ((int64_t)ch * a.nedge + bd.edge_slot) * 2 * a.edge_wmax;
      const uint32_t t_lo = tb, t_hi = (uint32_t)a.n - 1u - (tb + 15u * tstep);
      if (t_lo < (uint32_t)bd.edge) ez[t_lo] = u[brev(8, 4)];
      if (t_hi < (uint32_t)bd.edge) ez[a.edge_wmax + t_hi] = u[brev(7, 4)];
    }
#pragma unroll
    for (int i = 0; i < NOUT; ++i) {
      const int c = KIND == 0 ? i + 4 : (KIND == 1 ? ((i + 8) & 15) : i);
      const cplx<T> z = u[brev(c, 4)];
      const uint32_t tt = tb + (uint32_t)i * tstep;
      if (COEF && !QI_DBG(1)) stream_store(reinterpret_cast<cplx<T>*>(coef_row + (size_t)(tt * (uint32_t)sizeof(cplx<T>))), z);
      const T m2 = norm2(z.x, z.y);
      if (BITS) *reinterpret_cast<T*>(bits_row + (size_t)(tt * (uint32_t)sizeof(T))) = log2_t(sqrt_t(m2) + a.eps);
      T p = mul_rn(a.power_scale, m2);
      if (KIND == 1) {
        // short-atom bands evaluated circularly: the first / last `edge` samples are corrected (and reduced) by
        // k_edge_fix afterwards, so they are left out of the sums here.  edge = 0 keeps every sample.
        const bool inside = (uint32_t)(tt - (uint32_t)bd.edge) < (uint32_t)(a.n - 2 * (int64_t)bd.edge);
        p = inside ? p : T(0);
      }
      col[i] += p;
      if (!QI_DBG(8)) {
        rowacc += p;
        mx = max_t(mx, p);
        pl += plog2p(p, ltab);
      }
    }
    plogp += (double)pl;
    if (a.part_band) {
      const double r = wave_sum((double)rowacc);
      if (lane == 0) s_red[par][wv] = r;
      pending = j;
      par ^= 1;
    }
    QI_STAMP(6);
  }
#ifdef QI_NATIVE_STAMPS
  if (a.stamps && tid == 0) {
    unsigned long long* o = a.stamps + ((((int64_t)ch * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 8);
    for (int k = 0; k < 8; ++k) o[k] = st_acc[k];
  }
#endif

  __syncthreads();
  if (pending >= 0 && tid == 0) {
    double s = 0.0;
    for (int w = 0; w < C::TH / kWave; ++w) s += s_red[par ^ 1][w];
    a.part_band[((int64_t)ch * a.panel_bands + pending) * a.nblk + grp] = s;
  }
  T tot = T(0);
  char* __restrict__ time_row =
      reinterpret_cast<char*>(a.time_part ? a.time_part + ((int64_t)ch * a.chunk_total + a.chunk_base + blockIdx.y) * a.n
                                          : nullptr);
#pragma unroll
  for (int i = 0; i < NOUT; ++i) {
    tot += col[i];
    const uint32_t tt = tbase + (uint32_t)i * tstep;
    if (time_row) *reinterpret_cast<T*>(time_row + (size_t)(tt * (uint32_t)sizeof(T))) = col[i];
  }
  if (a.part_stat) {
    const double r0 = wave_max((double)mx), r1 = wave_sum((double)tot), r2 = wave_sum(plogp);
    if (lane == 0) {
      s_fin[0][wv] = r0;
      s_fin[1][wv] = r1;
      s_fin[2][wv] = r2;
    }
    __syncthreads();
    if (tid == 0) {
      double m = 0.0, s1 = 0.0, s2 = 0.0;
      for (int w = 0; w < C::TH / kWave; ++w) {
        m = s_fin[0][w] > m ? s_fin[0][w] : m;
        s1 += s_fin[1][w];
        s2 += s_fin[2][w];
      }
      double* o = a.part_stat + ((int64_t)ch * a.stat_stride + (int64_t)(a.chunk_base + blockIdx.y) * a.stat_nblk + grp) * 3;
      o[0] = m;
      o[1] = s1;
      o[2] = s2;
    }
  }
}

// power_time[c][t] = sum over chunks of time_part[c][q][t] (+ the corrected edge samples of the short-atom bands);
// workgroup bx of gx
template <typename T, int VEC>  // VEC samples per thread and plane (4: one 16-byte float load; needs aligned rows)
__device__ __forceinline__ void time_reduce_block(const T* __restrict__ part, T* __restrict__ out, int64_t n, int nchunk,
                                                  const T* __restrict__ edge_time, int64_t wmax, int64_t bx, int64_t gx,
                                                  int64_t c) {
  struct alignas(sizeof(T) * VEC) Pack {
    T v[VEC];
  };
  for (int64_t t = (bx * blockDim.x + threadIdx.x) * VEC; t < n; t += gx * blockDim.x * VEC) {
    if (t + VEC <= n) {
      Pack s = *reinterpret_cast<const Pack*>(part + (c * nchunk) * n + t);
      for (int q = 1; q < nchunk; ++q) {
        const Pack x = *reinterpret_cast<const Pack*>(part + (c * nchunk + q) * n + t);
#pragma unroll
        for (int i = 0; i < VEC; ++i) s.v[i] += x.v[i];
      }
      if (edge_time) {
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
          if (t + i < wmax) s.v[i] += edge_time[(c * 2 + 0) * wmax + t + i];
          if (t + i >= n - wmax) s.v[i] += edge_time[(c * 2 + 1) * wmax + (n - 1 - t - i)];
        }
      }
      *reinterpret_cast<Pack*>(out + c * n + t) = s;
    } else {
      for (int64_t u = t; u < n; ++u) {
        T s = T(0);
        for (int q = 0; q < nchunk; ++q) s += part[(c * nchunk + q) * n + u];
        if (edge_time) {
          if (u < wmax) s += edge_time[(c * 2 + 0) * wmax + u];
          if (u >= n - wmax) s += edge_time[(c * 2 + 1) * wmax + (n - 1 - u)];
        }
        out[c * n + u] = s;
      }
    }
  }
}
template <typename T, int VEC>
__global__ void __launch_bounds__(256) k_time_reduce(const T* __restrict__ part, T* __restrict__ out, int64_t n, int nchunk,
                                                     const T* __restrict__ edge_time, int64_t wmax) {
  time_reduce_block<T, VEC>(part, out, n, nchunk, edge_time, wmax, blockIdx.x, gridDim.x, blockIdx.y);
}

// The tail of a transform in one launch: the first nfin workgroups of a record finalise the per-band and whole-panel
// reductions (finalize_block), the others sum the per-time planes.
struct TailFin {
  const double* part_band;
  const double* part_stat;
  double* power_band;
  double* stats;
  int64_t B, nblk, nstat;
  const int32_t* band_slots;
};
template <typename T, int VEC>
__global__ void __launch_bounds__(256) k_tail(const T* __restrict__ part, T* __restrict__ out, int64_t n, int nchunk,
                                              const T* __restrict__ edge_time, int64_t wmax, TailFin f, int nfin) {
  __shared__ double s[3][256 / kWave];
  if ((int)blockIdx.x < nfin) {
    finalize_block(f.part_band, f.part_stat, f.power_band, f.stats, f.B, f.nblk, f.nstat, f.band_slots, blockIdx.x,
                   blockIdx.y, s);
    return;
  }
  time_reduce_block<T, VEC>(part, out, n, nchunk, edge_time, wmax, (int64_t)blockIdx.x - nfin, (int64_t)gridDim.x - nfin,
                            blockIdx.y);
}

// Xn[c][k] = X2n[c][2k]: the n-point spectrum of a record is the even bins of its zero-padded 2n-point spectrum
template <typename T>
__global__ void k_even_bins(const cplx<T>* __restrict__ x2, cplx<T>* __restrict__ x1, int64_t n) {
  const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, c = blockIdx.y;
  if (k < n) x1[c * n + k] = x2[c * 2 * n + 2 * k];
}

// Edge correction of the short-atom bands that pass 2 evaluated as a circular correlation of length n: the
// reference's zero-padded (linear) correlation differs only in the first / last W samples, by the taps that wrapped
// around the record end.  One wave per output sample subtracts sum_m sig[m] conj(psi(u)) over the wrapped taps,
// psi(u) = amp exp(-(p_re + i p_im) x^2) exp(i omega x), x = u + 1/2 (the centred atom of styx_cwt.py:113-144),
// rewrites the coefficient (and bits) and leaves its power in edge_p for the fixed-order reduction below.
template <typename T>
__global__ void __launch_bounds__(256) k_edge_fix(EdgeArgs<T> a) {
  const int lane = threadIdx.x & (kWave - 1), wv = threadIdx.x / kWave;
  const int64_t e = blockIdx.y, c = blockIdx.z;
  const EdgeBand eb = a.bands[e];
  const int64_t o = (int64_t)blockIdx.x * (256 / kWave) + wv;
  const int side = o >= a.wmax ? 1 : 0;
  const int64_t tloc = o - (int64_t)side * a.wmax;
  if (tloc >= eb.w) return;
  const int64_t cnt = eb.w - tloc;                      // wrapped taps of this output
  const int64_t t = side ? a.n - 1 - tloc : tloc;       // output sample
  const T* __restrict__ sig = a.sig + c * a.n;
  T sr = T(0), si = T(0);
  // head: m = n - W + t + i, u = m - t - n = i - W;  tail: m = i, u = m - t + n = i + 1 + tloc;  x = u + 1/2
  const int64_t m0 = side ? 0 : a.n - eb.w + t;
  const double x0 = (side ? (double)(1 + tloc) : (double)(-eb.w)) + 0.5;
  const int64_t chunk = (cnt + kWave - 1) / kWave;
  if constexpr (sizeof(T) == 8) {
    // float64: every tap in double (a few hundred taps per sample at most: the bands are wide-spectrum, short atoms)
    for (int64_t i = lane; i < cnt; i += kWave) {
      const double x = x0 + (double)i;
      const double ph = eb.omega * x - eb.p_im * x * x;
      const double turns = ph * 0.15915494309189535;  // / 2 pi
      double sn, cs;
      sincospi(2.0 * (turns - rint(turns)), &sn, &cs);
      const double env = eb.amp * exp(-(eb.p_re * x * x));
      const double v = (double)sig[m0 + i] * env;
      sr += (T)(v * cs);
      si -= (T)(v * sn);
    }
  } else if (eb.p_im == 0.0 && chunk >= 4) {
    // each lane takes `chunk` consecutive taps: phasor and Gaussian envelope by recurrences from exact seeds,
    // conj(psi(x + 1)) = conj(psi(x)) e^{-i omega} e^{-p (2 x + 1)}
    const int64_t i0 = (int64_t)lane * chunk;
    const int64_t i1 = i0 + chunk < cnt ? i0 + chunk : cnt;
    if (i0 < i1) {
      const double x = x0 + (double)i0;
      const double turns = eb.omega * x * 0.15915494309189535;
      float sn, cs, sw, cw;
      sincospif(2.0f * (float)(turns - rint(turns)), &sn, &cs);
      const double tw_ = eb.omega * 0.15915494309189535;
      sincospif(2.0f * (float)(tw_ - rint(tw_)), &sw, &cw);
      float env = (float)eb.amp * expf(-(float)(eb.p_re * x * x));
      float ratio = expf(-(float)(eb.p_re * (2.0 * x + 1.0)));
      const float ratio2 = expf(-(float)(2.0 * eb.p_re));
      float pr = cs, pi = -sn;  // conj phasor
      for (int64_t i = i0; i < i1; ++i) {
        const T v = sig[m0 + i] * (T)env;
        sr += v * (T)pr;
        si += v * (T)pi;
        const float nr = pr * cw + pi * sw;  // times e^{-i omega}
        pi = pi * cw - pr * sw;
        pr = nr;
        env *= ratio;
        ratio *= ratio2;
      }
    }
  } else {
    for (int64_t i = lane; i < cnt; i += kWave) {
      const double x = x0 + (double)i;
      const double ph = eb.omega * x - eb.p_im * x * x;
      const double turns = ph * 0.15915494309189535;  // / 2 pi
      float sn, cs;
      sincospif(2.0f * (float)(turns - rint(turns)), &sn, &cs);
      const float env = (float)eb.amp * expf(-(float)(eb.p_re * x * x));
      const T v = sig[m0 + i] * (T)env;
      sr += v * (T)cs;  // sig * conj(psi): conj(e^{i ph}) = cos - i sin
      si -= v * (T)sn;
    }
  }
  sr = wave_sum(sr);
  si = wave_sum(si);
  if (lane == 0) {
    const int64_t row = (c * a.panel_bands + eb.out_band) * a.n;
    cplx<T> z = a.coef ? a.coef[row + t] : a.edge_z[((c * a.nedge + e) * 2 + side) * a.wmax + tloc];
    z.x -= sr;
    z.y -= si;
    if (a.coef) a.coef[row + t] = z;
    const T m2 = norm2(z.x, z.y);
    if (a.bits) a.bits[row + t] = log2_t(sqrt_t(m2) + a.eps);
    a.edge_p[((c * a.nedge + e) * 2 + side) * a.wmax + tloc] = mul_rn(a.power_scale, m2);
  }
}

// Fixed-order reduction of the corrected edge powers.  Workgroup (e, c) with e < nedge: the edge samples of band e
// -> one extra partial of part_band and one extra part_stat entry; workgroups e >= nedge: per-sample sums over the
// bands -> edge_time.
template <typename T>
__global__ void __launch_bounds__(256) k_edge_reduce(EdgeArgs<T> a, T* __restrict__ edge_time,
                                                     double* __restrict__ part_band, int64_t nblk, int64_t band_slot,
                                                     double* __restrict__ part_stat, int64_t stat_slot0) {
  __shared__ double s[3][256 / kWave];
  const int tid = threadIdx.x, lane = tid & (kWave - 1), wv = tid / kWave;
  const int64_t c = blockIdx.y;
  const T* __restrict__ ep = a.edge_p + c * a.nedge * 2 * a.wmax;
  if ((int64_t)blockIdx.x >= a.nedge) {
    if (!edge_time) return;
    const int64_t i = ((int64_t)blockIdx.x - a.nedge) * 256 + tid;
    if (i >= 2 * a.wmax) return;
    const int side = i >= a.wmax ? 1 : 0;
    const int64_t tloc = i - (int64_t)side * a.wmax;
    T r = T(0);
    for (int64_t e = 0; e < a.nedge; ++e)
      if (tloc < a.bands[e].w) r += ep[(e * 2 + side) * a.wmax + tloc];
    edge_time[(c * 2 + side) * a.wmax + tloc] = r;
    return;
  }
  const int64_t e = blockIdx.x, w = a.bands[e].w;
  double mx = 0.0, sum = 0.0, pl = 0.0;
  for (int64_t i = tid; i < 2 * w; i += 256) {
    const double p = (double)ep[(e * 2 + (i >= w ? 1 : 0)) * a.wmax + (i >= w ? i - w : i)];
    sum += p;
    mx = max_t(mx, p);
    pl += (double)plog2p((T)p);
  }
  mx = wave_max(mx);
  sum = wave_sum(sum);
  pl = wave_sum(pl);
  if (lane == 0) {
    s[0][wv] = mx;
    s[1][wv] = sum;
    s[2][wv] = pl;
  }
  __syncthreads();
  if (tid == 0) {
    double m = 0.0, s1 = 0.0, s2 = 0.0;
    for (int q = 0; q < 256 / kWave; ++q) {
      m = s[0][q] > m ? s[0][q] : m;
      s1 += s[1][q];
      s2 += s[2][q];
    }
    if (part_band) part_band[(c * a.panel_bands + a.bands[e].out_band) * nblk + band_slot] = s1;
    if (part_stat) {
      double* o = part_stat + (c * a.stat_slots + stat_slot0 + e) * 3;
      o[0] = m;
      o[1] = s1;
      o[2] = s2;
    }
  }
}

// per band: max |F|^2 and the first / last bin at or above thr2 * max (plan-time support analysis)
__global__ void __launch_bounds__(256) k_band_support(const double2* __restrict__ F, int64_t L, double thr2,
                                                      double* __restrict__ out /*[nb][3]*/) {
  __shared__ double s[256 / kWave];
  __shared__ double s_max;
  const int tid = threadIdx.x, lane = tid & (kWave - 1), wv = tid / kWave;
  const double2* f = F + (int64_t)blockIdx.x * L;
  double m = 0.0;
  for (int64_t k = tid; k < L; k += 256) {
    const double p = f[k].x * f[k].x + f[k].y * f[k].y;
    m = p > m ? p : m;
  }
  m = wave_max(m);
  if (lane == 0) s[wv] = m;
  __syncthreads();
  if (tid == 0) {
    double t = 0.0;
    for (int w = 0; w < 256 / kWave; ++w) t = s[w] > t ? s[w] : t;
    s_max = t;
  }
  __syncthreads();
  const double cut = s_max * thr2;
  // plain support [lo, hi], and the support of a spectrum that straddles bin 0: [lo2 - L, hi1] with hi1 the last bin
  // above the cut in the lower half and lo2 the first one in the upper half; the shorter of the two is reported
  double lo = (double)L, hi = -1.0, hi1 = -1.0, lo2 = (double)L;
  for (int64_t k = tid; k < L; k += 256) {
    const double p = f[k].x * f[k].x + f[k].y * f[k].y;
    if (p >= cut && p > 0.0) {
      lo = (double)k < lo ? (double)k : lo;
      hi = (double)k > hi ? (double)k : hi;
      if (k < L / 2) hi1 = (double)k > hi1 ? (double)k : hi1;
      else lo2 = (double)k < lo2 ? (double)k : lo2;
    }
  }
  lo = -wave_max(-lo);
  hi = wave_max(hi);
  hi1 = wave_max(hi1);
  lo2 = -wave_max(-lo2);
  __shared__ double s_lo[256 / kWave], s_hi[256 / kWave], s_hi1[256 / kWave], s_lo2[256 / kWave];
  if (lane == 0) {
    s_lo[wv] = lo;
    s_hi[wv] = hi;
    s_hi1[wv] = hi1;
    s_lo2[wv] = lo2;
  }
  __syncthreads();
  if (tid == 0) {
    for (int w = 0; w < 256 / kWave; ++w) {
      lo = s_lo[w] < lo ? s_lo[w] : lo;
      hi = s_hi[w] > hi ? s_hi[w] : hi;
      hi1 = s_hi1[w] > hi1 ? s_hi1[w] : hi1;
      lo2 = s_lo2[w] < lo2 ? s_lo2[w] : lo2;
    }
    if (hi1 >= 0.0 && lo2 < (double)L && (hi1 + 1.0) + ((double)L - lo2) < hi - lo + 1.0) {
      lo = lo2 - (double)L;  // negative: bins are taken modulo L
      hi = hi1;
    }
    out[blockIdx.x * 3 + 0] = s_max;
    out[blockIdx.x * 3 + 1] = lo;
    out[blockIdx.x * 3 + 2] = hi;
  }
}

// copy a window of a float64 spectrum row into working precision, scaled (and conjugated for the circular bank)
template <typename T>
__global__ void k_copy_window(const double2* __restrict__ F, cplx<T>* __restrict__ dst, int64_t k_lo, int64_t count,
                              int conj, double scale, int64_t row_len, double ramp, int64_t ramp_center) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  int64_t k = k_lo + i;  // the window may start at a negative bin: bins are taken modulo the row length
  if (k < 0) k += row_len;
  if (k >= row_len) k -= row_len;
  const double2 v = F[k];
  double re = v.x * scale, im = (conj ? -v.y : v.y) * scale;
  if (ramp != 0.0) {  // times exp(2 pi i ramp (i - ramp_center)): a time shift of the band's envelope (zoom engine)
    double s, c;
    sincospi(2.0 * ramp * (double)(i - ramp_center), &s, &c);
    const double r2 = re * c - im * s;
    im = re * s + im * c;
    re = r2;
  }
  dst[i] = mk<T>((T)re, (T)im);
}

}  // namespace


template <class Kern, typename T>
static int launch_lds(Kern kern, size_t lds, const RowArgs<T>& a, dim3 grid, int threads, hipStream_t st) {
  QI_TRY(allow_dynamic_lds(reinterpret_cast<const void*>(kern), lds));
  kern<<<grid, threads, lds, st>>>(a);
  QI_LAUNCH_CHECK();
  return QI_OK;
}

template <typename T, class C, int SRC, int NPH>
static int launch_p1(const RowArgs<T>& a, dim3 grid, hipStream_t st) {
  return launch_lds(k_pass1<T, C, SRC, NPH>, C::LDS_BYTES, a, grid, C::TH, st);
}
template <typename T, class C, int KIND, bool COEF, bool BITS>
static int launch_p2v(const RowArgs<T>& a, dim3 grid, hipStream_t st) {
  return launch_lds(k_pass2<T, C, KIND, COEF, BITS>, C::LDS_BYTES, a, grid, C::TH, st);
}
// the optional outputs are compile-time variants: no per-sample branches in the epilogue
template <typename T, class C, int KIND>
static int launch_p2(const RowArgs<T>& a, dim3 grid, hipStream_t st) {
  if (a.coef) return a.bits ? launch_p2v<T, C, KIND, true, true>(a, grid, st) : launch_p2v<T, C, KIND, true, false>(a, grid, st);
  return a.bits ? launch_p2v<T, C, KIND, false, true>(a, grid, st) : launch_p2v<T, C, KIND, false, false>(a, grid, st);
}

template <typename T, class C>
static int launch_pass1_cfg(const RowArgs<T>& a0, int kind, int64_t n_channels, hipStream_t st) {
  RowArgs<T> a = a0;
  a.phase_split = a.N1 == 2048 && (a.N2 / C::G) * a.ngen_launch * n_channels < 256 ? 1 : 0;
  dim3 grid((unsigned)(a.N2 / C::G), (unsigned)(a.ngen_launch * (a.phase_split ? 2 : 1)), (unsigned)n_channels);
  const bool stx = kind == 2;
  if (a.N1 == 1024) return stx ? launch_p1<T, C, 1, 1>(a, grid, st) : launch_p1<T, C, 0, 1>(a, grid, st);
  if (a.N1 == 2048) return stx ? launch_p1<T, C, 1, 2>(a, grid, st) : launch_p1<T, C, 0, 2>(a, grid, st);
  set_error("native pass 1 supports N1 = 1024 or 2048, got %lld", (long long)a.N1);
  return QI_ERR_UNSUPPORTED;
}
template <>
int launch_pass1<float>(const RowArgs<float>& a, int kind, int64_t n_channels, hipStream_t st) {
  if (a.ngen_launch <= 0) return QI_OK;
  // (8-row workgroups were measured for launches that do not cover the chip: slower, their 64-byte runs cost more)
  return launch_pass1_cfg<float, Cfg<float, 16>>(a, kind, n_channels, st);
}
// float64: 8-row workgroups of 512 threads (the row transform holds 2 x 16 complex doubles per thread: 256-register
// budget; LDS image 152 KB, one workgroup per CU)
template <>
int launch_pass1<double>(const RowArgs<double>& a, int kind, int64_t n_channels, hipStream_t st) {
  if (a.ngen_launch <= 0) return QI_OK;
  return launch_pass1_cfg<double, Cfg<double, 8>>(a, kind, n_channels, st);
}

// forward transform of n_channels real records (a.sig) into Xout [C][Lf], through a.imd (one slot per channel)
template <typename T, class C1>
static int launch_forward_cfg(const RowArgs<T>& a0, cplx<T>* Xout, int64_t n_channels, hipStream_t st) {
  using C2 = C1;
  RowArgs<T> a = a0;
  a.phase_split = a.N1 == 2048 && (a.N2 / C1::G) * n_channels < 256 ? 1 : 0;
  dim3 g1((unsigned)(a.N2 / C1::G), a.phase_split ? 2u : 1u, (unsigned)n_channels);
  if (a.N1 == 1024)
    QI_TRY((launch_p1<T, C1, 2, 1>(a, g1, st)));
  else if (a.N1 == 2048)
    QI_TRY((launch_p1<T, C1, 2, 2>(a, g1, st)));
  else {
    set_error("native forward transform supports N1 = 1024 or 2048, got %lld", (long long)a.N1);
    return QI_ERR_UNSUPPORTED;
  }
  auto kern = k_fwd2<T, C2>;
  QI_TRY(allow_dynamic_lds(reinterpret_cast<const void*>(kern), C2::LDS_BYTES));
  dim3 g2((unsigned)(a.N1 / C2::G), 1, (unsigned)n_channels);
  kern<<<g2, C2::TH, C2::LDS_BYTES, st>>>(a, Xout);
  QI_LAUNCH_CHECK();
  return QI_OK;
}
template <>
int launch_forward<float>(const RowArgs<float>& a, float2* Xout, int64_t n_channels, hipStream_t st) {
  // few records: 8-row workgroups, twice as many of them (a launch of 16-row workgroups would leave most CUs idle)
  if (n_channels * (a.N2 / 16) < 256) return launch_forward_cfg<float, Cfg<float, 8>>(a, Xout, n_channels, st);
  return launch_forward_cfg<float, Cfg<float, 16>>(a, Xout, n_channels, st);
}
// float64 (round 4): the same two launches in double instead of hipFFT's five passes over the padded records
template <>
int launch_forward<double>(const RowArgs<double>& a, double2* Xout, int64_t n_channels, hipStream_t st) {
  return launch_forward_cfg<double, Cfg<double, 8>>(a, Xout, n_channels, st);
}

template <typename T, class C>
static int launch_pass2_cfg(const RowArgs<T>& a, int kind, int nchunk, int64_t n_channels, hipStream_t st) {
  dim3 grid((unsigned)(a.N1 / C::G), (unsigned)nchunk, (unsigned)n_channels);
  switch (kind) {
    case 0: return launch_p2<T, C, 0>(a, grid, st);
    case 1: return launch_p2<T, C, 1>(a, grid, st);
    default: return launch_p2<T, C, 2>(a, grid, st);
  }
}
template <>
int launch_pass2<double>(const RowArgs<double>& a, int kind, int rows_per_group, int nchunk, int64_t n_channels,
                         hipStream_t st) {
  if (rows_per_group != 8) {
    set_error("float64 pass 2 runs 8 rows per workgroup, got %d", rows_per_group);
    return QI_ERR_UNSUPPORTED;
  }
  return launch_pass2_cfg<double, Cfg<double, 8>>(a, kind, nchunk, n_channels, st);
}

template <>
int launch_pass2<float>(const RowArgs<float>& a, int kind, int rows_per_group, int nchunk, int64_t n_channels,
                        hipStream_t st) {
  if (rows_per_group == 16) return launch_pass2_cfg<float, Cfg<float, 16>>(a, kind, nchunk, n_channels, st);
  // 8 rows: half the LDS image, two workgroups per CU that hide each other's barriers, but 64-byte store runs
  if (rows_per_group == 8) return launch_pass2_cfg<float, Cfg<float, 8>>(a, kind, nchunk, n_channels, st);
  set_error("pass 2 supports 8 or 16 rows per workgroup, got %d", rows_per_group);
  return QI_ERR_UNSUPPORTED;
}

template <typename T>
int launch_time_reduce(const T* part, T* out, int64_t C, int64_t n, int nchunk, const T* edge_time, int64_t wmax,
                       hipStream_t st) {
  const bool aligned = n % 4 == 0 && reinterpret_cast<uintptr_t>(out) % (4 * sizeof(T)) == 0 &&
                       reinterpret_cast<uintptr_t>(part) % (4 * sizeof(T)) == 0;
  if (aligned) {
    dim3 g((unsigned)(ceil_div(n, 1024) > 4096 ? 4096 : ceil_div(n, 1024)), (unsigned)C);
    k_time_reduce<T, 4><<<g, 256, 0, st>>>(part, out, n, nchunk, edge_time, wmax);
  } else {
    dim3 g((unsigned)(ceil_div(n, 256) > 4096 ? 4096 : ceil_div(n, 256)), (unsigned)C);
    k_time_reduce<T, 1><<<g, 256, 0, st>>>(part, out, n, nchunk, edge_time, wmax);
  }
  QI_LAUNCH_CHECK();
  return QI_OK;
}
template <typename T>
int launch_tail(const T* part, T* out, int64_t C, int64_t n, int nchunk, const T* edge_time, int64_t wmax,
                const double* part_band, const double* part_stat, double* power_band, double* stats, int64_t B,
                int64_t nblk, int64_t nstat, const int32_t* band_slots, hipStream_t st);
// two tails (the two transforms of qi_cwt_stx) in one launch: blockIdx.z selects the transform
template <typename T>
struct TailPack {
  const T* part;
  T* out;
  int64_t n;
  int nchunk, nfin;
  TailFin f;
};
template <typename T, int VEC>
__global__ void __launch_bounds__(256) k_tail2(TailPack<T> p0, TailPack<T> p1) {
  __shared__ double s[3][256 / kWave];
  const TailPack<T>& p = blockIdx.z == 0 ? p0 : p1;
  if ((int)blockIdx.x < p.nfin) {
    finalize_block(p.f.part_band, p.f.part_stat, p.f.power_band, p.f.stats, p.f.B, p.f.nblk, p.f.nstat, p.f.band_slots,
                   blockIdx.x, blockIdx.y, s);
    return;
  }
  time_reduce_block<T, VEC>(p.part, p.out, p.n, p.nchunk, nullptr, 0, (int64_t)blockIdx.x - p.nfin,
                            (int64_t)gridDim.x - p.nfin, blockIdx.y);
}

template <typename T>
int launch_tail2(const T* part0, T* out0, int nchunk0, const double* part_band0, const double* part_stat0,
                 double* power_band0, double* stats0, int64_t B0, int64_t nblk0, int64_t nstat0,
                 const int32_t* band_slots0, const T* part1, T* out1, int nchunk1, const double* part_band1,
                 const double* part_stat1, double* power_band1, double* stats1, int64_t B1, int64_t nblk1, int64_t nstat1,
                 const int32_t* band_slots1, int64_t C, int64_t n, hipStream_t st) {
  const bool aligned = n % 4 == 0 && reinterpret_cast<uintptr_t>(out0) % 16 == 0 && reinterpret_cast<uintptr_t>(part0) % 16 == 0 &&
                       reinterpret_cast<uintptr_t>(out1) % 16 == 0 && reinterpret_cast<uintptr_t>(part1) % 16 == 0;
  if (!aligned) {
    if (int rc = launch_tail<T>(part0, out0, C, n, nchunk0, nullptr, 0, part_band0, part_stat0, power_band0, stats0, B0, nblk0,
                                nstat0, band_slots0, st))
      return rc;
    return launch_tail<T>(part1, out1, C, n, nchunk1, nullptr, 0, part_band1, part_stat1, power_band1, stats1, B1, nblk1,
                          nstat1, band_slots1, st);
  }
  TailPack<T> p0{part0, out0, n, nchunk0, (int)B0 + 1, TailFin{part_band0, part_stat0, power_band0, stats0, B0, nblk0, nstat0, band_slots0}};
  TailPack<T> p1{part1, out1, n, nchunk1, (int)B1 + 1, TailFin{part_band1, part_stat1, power_band1, stats1, B1, nblk1, nstat1, band_slots1}};
  const int64_t gt = ceil_div(n, 1024) > 4096 ? 4096 : ceil_div(n, 1024);
  const int nfin = p0.nfin > p1.nfin ? p0.nfin : p1.nfin;
  dim3 g((unsigned)(gt + nfin), (unsigned)C, 2);
  k_tail2<T, 4><<<g, 256, 0, st>>>(p0, p1);
  QI_LAUNCH_CHECK();
  return QI_OK;
}

template <typename T>
int launch_tail(const T* part, T* out, int64_t C, int64_t n, int nchunk, const T* edge_time, int64_t wmax,
                const double* part_band, const double* part_stat, double* power_band, double* stats, int64_t B,
                int64_t nblk, int64_t nstat, const int32_t* band_slots, hipStream_t st) {
  const bool aligned = n % 4 == 0 && reinterpret_cast<uintptr_t>(out) % (4 * sizeof(T)) == 0 &&
                       reinterpret_cast<uintptr_t>(part) % (4 * sizeof(T)) == 0;
  if (!aligned) {  // (cannot happen for the engine's power-of-two records: two launches)
    if (int rc = launch_time_reduce<T>(part, out, C, n, nchunk, edge_time, wmax, st)) return rc;
    return launch_finalize(part_band, part_stat, power_band, stats, C, B, nblk, nstat, st, band_slots);
  }
  const TailFin f{part_band, part_stat, power_band, stats, B, nblk, nstat, band_slots};
  const int nfin = (int)B + 1;
  const int64_t gt = ceil_div(n, 1024) > 4096 ? 4096 : ceil_div(n, 1024);
  dim3 g((unsigned)(gt + nfin), (unsigned)C);
  k_tail<T, 4><<<g, 256, 0, st>>>(part, out, n, nchunk, edge_time, wmax, f, nfin);
  QI_LAUNCH_CHECK();
  return QI_OK;
}
template int launch_tail2<float>(const float*, float*, int, const double*, const double*, double*, double*, int64_t, int64_t,
                                 int64_t, const int32_t*, const float*, float*, int, const double*, const double*, double*,
                                 double*, int64_t, int64_t, int64_t, const int32_t*, int64_t, int64_t, hipStream_t);
template int launch_tail<float>(const float*, float*, int64_t, int64_t, int, const float*, int64_t, const double*,
                                const double*, double*, double*, int64_t, int64_t, int64_t, const int32_t*, hipStream_t);
template int launch_tail<double>(const double*, double*, int64_t, int64_t, int, const double*, int64_t, const double*,
                                 const double*, double*, double*, int64_t, int64_t, int64_t, const int32_t*, hipStream_t);
template int launch_time_reduce<float>(const float*, float*, int64_t, int64_t, int, const float*, int64_t, hipStream_t);
template int launch_time_reduce<double>(const double*, double*, int64_t, int64_t, int, const double*, int64_t,
                                        hipStream_t);

template <typename T>
int launch_even_bins(const cplx<T>* x2, cplx<T>* x1, int64_t C, int64_t n, hipStream_t st) {
  dim3 g((unsigned)ceil_div(n, 256), (unsigned)C);
  k_even_bins<T><<<g, 256, 0, st>>>(x2, x1, n);
  QI_LAUNCH_CHECK();
  return QI_OK;
}
template int launch_even_bins<float>(const float2*, float2*, int64_t, int64_t, hipStream_t);
template int launch_even_bins<double>(const double2*, double2*, int64_t, int64_t, hipStream_t);

template <typename T>
int launch_edge(const EdgeArgs<T>& a, int64_t C, T* edge_time, double* part_band, int64_t nblk, int64_t band_slot,
                double* part_stat, int64_t stat_slot, hipStream_t st) {
  if (a.nedge <= 0) return QI_OK;
  dim3 g((unsigned)ceil_div(2 * a.wmax, 256 / kWave), (unsigned)a.nedge, (unsigned)C);
  k_edge_fix<T><<<g, 256, 0, st>>>(a);
  QI_LAUNCH_CHECK();
  dim3 gr((unsigned)(a.nedge + ceil_div(2 * a.wmax, 256)), (unsigned)C);
  k_edge_reduce<T><<<gr, 256, 0, st>>>(a, edge_time, part_band, nblk, band_slot, part_stat, stat_slot);
  QI_LAUNCH_CHECK();
  return QI_OK;
}
template int launch_edge<float>(const EdgeArgs<float>&, int64_t, float*, double*, int64_t, int64_t, double*, int64_t,
                                hipStream_t);
template int launch_edge<double>(const EdgeArgs<double>&, int64_t, double*, double*, int64_t, int64_t, double*, int64_t,
                                 hipStream_t);

int launch_band_support(const double2* F, int64_t L, int nb, double thr2, double* out, hipStream_t st) {
  k_band_support<<<nb, 256, 0, st>>>(F, L, thr2, out);
  QI_LAUNCH_CHECK();
  return QI_OK;
}

template <typename T>
int launch_copy_window(const double2* F, cplx<T>* dst, int64_t k_lo, int64_t count, int conj, double scale,
                       int64_t row_len, hipStream_t st, double ramp, int64_t ramp_center) {
  k_copy_window<T><<<(unsigned)ceil_div(count, 256), 256, 0, st>>>(F, dst, k_lo, count, conj, scale, row_len, ramp, ramp_center);
  QI_LAUNCH_CHECK();
  return QI_OK;
}
template int launch_copy_window<float>(const double2*, float2*, int64_t, int64_t, int, double, int64_t, hipStream_t, double,
                                       int64_t);
template int launch_copy_window<double>(const double2*, double2*, int64_t, int64_t, int, double, int64_t, hipStream_t,
                                        double, int64_t);

}  // namespace native
}  // namespace qi
