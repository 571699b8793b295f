// Fused short-time Fourier transform (styx_fft.stft_complex_pow2 / stft_from_sig, ref styx_fft.py:14-57,152-187;
// scipy.signal.stft with boundary="zeros", padded=True, detrend="constant", one-sided) for power-of-two transform
// lengths up to 4096: ONE kernel instead of frames -> batched R2C FFT -> transpose.
//
// A workgroup owns G consecutive segments of one record.  Per segment a wave loads the (zero-extended) samples,
// removes their mean (float64 sum, fixed order), applies the window and leaves the nfft real values in LDS packed as
// M = nfft / 2 complex numbers z[m] = v[2m] + i v[2m+1]; the G transforms of M points run in place in LDS (radix-2^2
// decimation in frequency, twiddles from an LDS table of exp(-2 pi i k / nfft) copied from a per-device table, results in
// bit-reversed order); the real-input spectra X[k], k = 0..M, are untangled from Z[k], Z[M - k] on the way out and
// written [frequency][time]: the G segments of a bin are consecutive in memory, so every store is a run of G
// coefficients (G = 16: 128 B).  HBM traffic = the record once (the 50 % overlap is served by the caches) + the
// panel once (+ the bits panel when asked for): the algorithmic bytes of SURVEY s8(d).  No MFMA: an FFT has no dense
// contraction.
#include <map>
#include <mutex>
#include <vector>

#include "qi_common.hpp"
#include "qi_device.hpp"
#include "qi_fft_reg.hpp"

namespace qi {

namespace {

#ifndef QI_STFT_THREADS
#define QI_STFT_THREADS 256
#endif
constexpr int kStftThreads = QI_STFT_THREADS;
#ifdef QI_STFT_WAVES  // experiments: waves per SIMD the register budget must allow
#define QI_STFT_BOUNDS __launch_bounds__(kStftThreads, QI_STFT_WAVES)
#else
#define QI_STFT_BOUNDS __launch_bounds__(kStftThreads)
#endif

template <typename T>
__device__ __forceinline__ void sincospi_t(T x, T* s, T* c);
template <>
__device__ __forceinline__ void sincospi_t<float>(float x, float* s, float* c) {
  sincospif(x, s, c);
}
template <>
__device__ __forceinline__ void sincospi_t<double>(double x, double* s, double* c) {
  sincospi(x, s, c);
}

template <typename T>
__device__ __forceinline__ cplx<T> cadd(cplx<T> a, cplx<T> b) {
  return mk<T>(a.x + b.x, a.y + b.y);
}
template <typename T>
__device__ __forceinline__ cplx<T> csub(cplx<T> a, cplx<T> b) {
  return mk<T>(a.x - b.x, a.y - b.y);
}

// sample k of the record extended past its ends (np.pad: constant 0 / edge / reflect even / odd; qi_stft_sliding.hip)
template <typename T>
__device__ __forceinline__ T stft_sample(const T* __restrict__ x, int64_t n, int64_t k, int mode) {
  if (k >= 0 && k < n) return x[k];
  if (mode == 0) return T(0);
  if (mode == 1) return k < 0 ? x[0] : x[n - 1];
  const int64_t j = k < 0 ? -k : 2 * (n - 1) - k;
  if (mode == 2) return x[j];
  return T(2) * (k < 0 ? x[0] : x[n - 1]) - x[j];
}

struct StftFusedArgs {
  int64_t n, seg, hop, nseg, lead;  // lead: zero-extended samples in front of the record (seg / 2 for the STFT, 0 Welch)
  int32_t log2g, G;                 // G = 1 << log2g segments per workgroup
  int32_t ngroups, per_xcd;         // segment groups per record; work items (record, group) per XCD
  int64_t nitems;                   // records x groups
  double scale, eps;
  double* welch_part;  // Welch (welch_power_pow2): [C][ngroups][nfft / 2 + 1] sums of |X|^2 over a group's segments; no panel
  // scipy.signal.ShortTimeFFT's convention (qi_sliding_stft): how the record is extended past its ends (0 zeros, 1 edge
  // values, 2 / 3 even / odd reflection), whether the slice's mean is removed, the left rotation of the windowed slice
  // (a phase ramp exp(2 pi i k roll / nfft) on the coefficients) and what the real output holds (0: log2 bits, 1: |X|,
  // 2: |X|^2)
  int32_t pad_mode, detrend, real_kind;
  int64_t roll;
  int32_t dbg;  // -DQI_STFT_DBG builds: timing ablations (1: no panel stores, 2: no transforms, 4: no record loads, 8: no bits)
};

// dynamic LDS: data [G][R (C + 1) + 1] complex | twiddles [M + 1] complex (exp(-2 pi i k / (2 M)))
//
// The M-point transform of a segment is a four-step transform on the R x C matrix z[r][c] = z[c + C r] (rows padded
// by one element, segments by one more: every access pattern below is free of bank conflicts):
//   step 1: thread = column c: R-point transform over r in registers, times W_M^(c k1), back to row k1;
//   step 2: thread = row k1:   C-point transform over c in registers: row k1, column k2 holds Z[k1 + R k2].
//
// PLAIN: the styx_fft product (zero extension, both panels, log2 bits, no slice rotation) with its loops specialised:
// whole segments come in as aligned pairs, a thread of the store pass keeps its segment and walks the bins with running
// addresses.  The general form serves Welch and the ShortTimeFFT convention.
template <typename T, int LOG2R, int LOG2C, bool PLAIN>
__global__ void QI_STFT_BOUNDS k_stft_fused(const T* __restrict__ sig, const T* __restrict__ win,
                                                            const cplx<T>* __restrict__ twg, cplx<T>* __restrict__ Z,
                                                            T* __restrict__ bits, StftFusedArgs a) {
  extern __shared__ __align__(16) unsigned char lds_raw[];
  constexpr int R = 1 << LOG2R, C = 1 << LOG2C, M = R * C, RS = C + 1, TILE = R * RS + 1;
  const int G = a.G;
  cplx<T>* __restrict__ data = reinterpret_cast<cplx<T>*>(lds_raw);
  cplx<T>* __restrict__ tw = data + (size_t)G * TILE;
  const int tid = threadIdx.x, lane = tid & (kWave - 1), wv = tid / kWave;
  // XCD-aware mapping: consecutive workgroup ids go to consecutive XCDs (eight L2s), and a workgroup writes only G
  // consecutive time samples of each frequency row (64-byte runs).  Every XCD takes a CONTIGUOUS range of the (record,
  // segment group) items, so the runs that complete a 128-byte line come from workgroups behind the same L2, a few
  // dispatch slots apart, and leave it as full lines.
  const int64_t item = (int64_t)(blockIdx.x & 7) * a.per_xcd + (blockIdx.x >> 3);
  if ((int)(blockIdx.x >> 3) >= a.per_xcd || item >= a.nitems) return;
  const int64_t c = item / a.ngroups, m0 = (item % a.ngroups) * G;
  const T* __restrict__ x = sig + c * a.n;

  // exp(-i pi k / M), k = 0..M: requested first, left in LDS after the segments (one round trip, not one per sweep)
  constexpr int NT = (M + kStftThreads) / kStftThreads;
  cplx<T> twv[NT];
#pragma unroll
  for (int i = 0; i < NT; ++i) {
    const int k = tid + i * kStftThreads;
    twv[i] = k <= M ? twg[k] : mk<T>(T(0), T(0));
  }
  // segments: one wave per segment (mean by wave shuffles, no workgroup barrier)
  constexpr int NW = kStftThreads / kWave, NP = (M + kWave - 1) / kWave;
  constexpr bool kFast = NP * sizeof(cplx<T>) <= 128;  // the paths that hold a segment and its window in registers
  auto one_segment = [=](int g) {
    const int64_t m = m0 + g;
    cplx<T>* __restrict__ d = data + (size_t)g * TILE;
    if (m >= a.nseg) {  // past the last segment: zeros (never stored)
      for (int j = lane; j < M; j += kWave) d[(j >> LOG2C) * RS + (j & (C - 1))] = mk<T>(T(0), T(0));
      return;
    }
    // the segment's samples, once, into registers: lane l holds the pairs (2 j, 2 j + 1), j = l + 64 t (out-of-record
    // samples of the zero extension are zeros and count in the mean, as in the reference)
    const int64_t base = m * a.hop - a.lead;
    const int pad_mode = PLAIN ? 0 : a.pad_mode;
    T v0[NP], v1[NP];
    // a segment of nfft samples inside the record, 2-sample aligned: one load per pair
    const bool whole = kFast && a.seg == 2 * M && base >= 0 && base + 2 * M <= a.n &&
                       ((reinterpret_cast<uintptr_t>(x + base) | reinterpret_cast<uintptr_t>(win)) & (2 * sizeof(T) - 1)) == 0;
    if (whole) {
      const cplx<T>* __restrict__ xp = reinterpret_cast<const cplx<T>*>(x + base);
#pragma unroll
      for (int t = 0; t < NP; ++t) {
        const int j = lane + kWave * t;
        const cplx<T> q = (NP * kWave == M || j < M) ? xp[j] : mk<T>(T(0), T(0));
        v0[t] = q.x;
        v1[t] = q.y;
      }
    } else {
#pragma unroll
      for (int t = 0; t < NP; ++t) {
        const int64_t i0 = 2 * (int64_t)(lane + kWave * t), k0 = base + i0;
        v0[t] = i0 < a.seg ? stft_sample<T>(x, a.n, k0, pad_mode) : T(0);
        v1[t] = i0 + 1 < a.seg ? stft_sample<T>(x, a.n, k0 + 1, pad_mode) : T(0);
      }
    }
    double acc = 0.0;
#pragma unroll
    for (int t = 0; t < NP; ++t) acc += (double)v0[t] + (double)v1[t];
    acc = wave_sum(acc);
    const T mean = a.detrend ? (T)(acc / (double)a.seg) : T(0);
#pragma unroll
    for (int t = 0; t < NP; ++t) {
      const int j = lane + kWave * t;
      if (NP * kWave == M || j < M) {
        T w0, w1;
        if (whole) {
          const cplx<T> w = reinterpret_cast<const cplx<T>*>(win)[j];
          w0 = w.x;
          w1 = w.y;
        } else {
          const int64_t i0 = 2 * (int64_t)j;
          w0 = i0 < a.seg ? win[i0] : T(0);
          w1 = i0 + 1 < a.seg ? win[i0 + 1] : T(0);
        }
        d[(j >> LOG2C) * RS + (j & (C - 1))] = mk<T>((v0[t] - mean) * w0, (v1[t] - mean) * w1);
      }
    }
  };
  // Two consecutive segments at half overlap (the product's geometry) share their middle half: a wave loads the three
  // halves once, all in flight together, and windows them twice.
#ifdef QI_STFT_NO_PAIRS
  constexpr bool kPairs = false;
#else
  constexpr bool kPairs = PLAIN && NP >= 2 && NP * kWave == M && kFast;
#endif
  bool pairs = false;
  if constexpr (kPairs)
    pairs = a.hop == M && a.seg == 2 * M && (G & 1) == 0 && (reinterpret_cast<uintptr_t>(win) & (2 * sizeof(T) - 1)) == 0;
  if (pairs) {
    if constexpr (kPairs) {
      constexpr int NH = NP / 2;  // pairs of samples per lane and half segment
      for (int g = 2 * wv; g < G; g += 2 * NW) {
        const int64_t m = m0 + g, base = m * a.hop - a.lead;
        if (!(m + 1 < a.nseg && base >= 0 && base + 3 * M <= a.n &&
              (reinterpret_cast<uintptr_t>(x + base) & (2 * sizeof(T) - 1)) == 0)) {
          one_segment(g);
          one_segment(g + 1);
          continue;
        }
        const cplx<T>* __restrict__ xp = reinterpret_cast<const cplx<T>*>(x + base);
        const cplx<T>* __restrict__ wp = reinterpret_cast<const cplx<T>*>(win);
        cplx<T> h[3][NH], w[2][NH];
#pragma unroll
        for (int q = 0; q < 3; ++q)
#pragma unroll
          for (int t = 0; t < NH; ++t) h[q][t] = xp[q * (M / 2) + lane + kWave * t];
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
          for (int t = 0; t < NH; ++t) w[q][t] = wp[q * (M / 2) + lane + kWave * t];
        double sh[3];
#pragma unroll
        for (int q = 0; q < 3; ++q) {
          sh[q] = 0.0;
#pragma unroll
          for (int t = 0; t < NH; ++t) sh[q] += (double)h[q][t].x + (double)h[q][t].y;
        }
        const double accA = wave_sum(sh[0] + sh[1]), accB = wave_sum(sh[1] + sh[2]);
        const T meanA = a.detrend ? (T)(accA / (double)a.seg) : T(0), meanB = a.detrend ? (T)(accB / (double)a.seg) : T(0);
        cplx<T>* __restrict__ dA = data + (size_t)g * TILE;
        cplx<T>* __restrict__ dB = dA + TILE;
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
          for (int t = 0; t < NH; ++t) {
            const int j = q * (M / 2) + lane + kWave * t, at = (j >> LOG2C) * RS + (j & (C - 1));
            dA[at] = mk<T>((h[q][t].x - meanA) * w[q][t].x, (h[q][t].y - meanA) * w[q][t].y);
            dB[at] = mk<T>((h[q + 1][t].x - meanB) * w[q][t].x, (h[q + 1][t].y - meanB) * w[q][t].y);
          }
      }
    }
  } else {
    for (int g = wv; g < G; g += NW) one_segment(g);
  }
#pragma unroll
  for (int i = 0; i < NT; ++i) {
    const int k = tid + i * kStftThreads;
    if (k <= M) tw[k] = twv[i];
  }
  __syncthreads();

  // step 1: columns
#ifdef QI_STFT_DBG
  if (!(a.dbg & 2))
#endif
  for (int q = tid; q < G * C; q += kStftThreads) {
    const int g = q >> LOG2C, cc = q & (C - 1);
    cplx<T>* __restrict__ d = data + (size_t)g * TILE + cc;
    cplx<T> v[R];
#pragma unroll
    for (int r = 0; r < R; ++r) v[r] = d[r * RS];
    native::fft_reg<T, R, -1>(v);
#pragma unroll
    for (int k1 = 0; k1 < R; ++k1) {
      cplx<T> y = v[native::brev(k1, LOG2R)];
      if (k1 > 0) {
        const int j2 = 2 * cc * k1;  // W_M^(c k1) = exp(-i pi (2 c k1) / M); the table covers [0, M)
        const cplx<T> w = tw[j2 & (M - 1)];
        y = cmul(y, (j2 & M) ? mk<T>(-w.x, -w.y) : w);
      }
      d[k1 * RS] = y;
    }
  }
  __syncthreads();
  // step 2: rows
#ifdef QI_STFT_DBG
  if (!(a.dbg & 2))
#endif
  for (int q = tid; q < G * R; q += kStftThreads) {
    const int g = q >> LOG2R, k1 = q & (R - 1);
    cplx<T>* __restrict__ d = data + (size_t)g * TILE + k1 * RS;
    cplx<T> v[C];
#pragma unroll
    for (int j = 0; j < C; ++j) v[j] = d[j];
    native::fft_reg<T, C, -1>(v);
#pragma unroll
    for (int k2 = 0; k2 < C; ++k2) d[k2] = v[native::brev(k2, LOG2C)];
  }
  __syncthreads();

  // untangle and store: X[k] = (Z[k] + conj Z[M-k]) / 2 - (i / 2) exp(-i pi k / M) (Z[k] - conj Z[M-k]), Z[M] = Z[0];
  // Z[k] = row k mod R, column k / R.  Consecutive threads take consecutive segments of one bin.
  const int nf = M + 1, lg = a.log2g;
  const T scale = (T)a.scale, eps = (T)a.eps;
  if constexpr (PLAIN) {
    // thread = (bin k0 of a sweep of S = threads / G bins, segment g): its segment, its tile and its place in a row stay;
    // bins advance by S, the panel addresses by S rows
    const int g = tid & (G - 1), S = kStftThreads >> lg;
    const int64_t m = m0 + g;
    if (m >= a.nseg) return;
    const cplx<T>* __restrict__ d = data + (size_t)g * TILE;
    const T hs = T(0.5) * scale;
    const int64_t row0 = (c * nf + (tid >> lg)) * a.nseg + m, step = (int64_t)S * a.nseg;
    cplx<T>* __restrict__ zp = Z + row0;
    T* __restrict__ bp = bits + row0;
    // the bins k and M - k together: one pair of reads, one product (exp(-i pi (M - k) / M) = -conj exp(-i pi k / M)) and
    // the sums shared -- X[M - k] = scale ((e.x - (w o).y), -(e.y) - (w o).x) in the terms of X[k]; two row streams, one up,
    // one down (0.395 -> 0.38 ms at configs[2])
    cplx<T>* __restrict__ zq = Z + (c * nf + (M - (tid >> lg))) * a.nseg + m;
    T* __restrict__ bq = bits + (c * nf + (M - (tid >> lg))) * a.nseg + m;
    for (int k = tid >> lg; k <= M / 2; k += S, zp += step, bp += step, zq -= step, bq -= step) {
      const int ka = k, kb = (M - k) & (M - 1);
      const cplx<T> za = d[(ka & (R - 1)) * RS + (ka >> LOG2R)], zb = d[(kb & (R - 1)) * RS + (kb >> LOG2R)];
      const cplx<T> w = tw[k];
      const T A = za.x + zb.x, B = za.y - zb.y;
      const cplx<T> wo = cmul(mk<T>(za.x - zb.x, za.y + zb.y), w);
      const cplx<T> X = mk<T>(hs * (A + wo.y), hs * (B - wo.x)), Y = mk<T>(hs * (A - wo.y), hs * (-B - wo.x));
#ifdef QI_STFT_DBG
      if (a.dbg & 1) {  // no panel stores (the arithmetic is kept alive by a store that never happens)
        const T p = X.x * X.x + X.y * X.y + Y.x * Y.x + Y.y * Y.y;
        const T b = (a.dbg & 8) ? p : log2_t(sqrt_t(p) + eps);
        if (b == T(12345.678)) *zp = X;
        continue;
      }
      if (a.dbg & 8) {  // coefficients only
        *zp = X;
        if (2 * k != M) *zq = Y;
        continue;
      }
#endif
      *zp = X;
      *bp = log2_t(sqrt_t(X.x * X.x + X.y * X.y) + eps);
      if (2 * k != M) {
        *zq = Y;
        *bq = log2_t(sqrt_t(Y.x * Y.x + Y.y * Y.y) + eps);
      }
    }
    return;
  }
  if (a.welch_part) {
    // Welch mean (styx_fft.py:230-266): the sum over this workgroup's segments of |X[k]|^2, one partial per (record,
    // group, bin); k_welch_reduce adds the groups in index order
    double* __restrict__ part = a.welch_part + ((size_t)c * a.ngroups + (size_t)(item % a.ngroups)) * nf;
    for (int k = tid; k < nf; k += kStftThreads) {
      const int ka = k & (M - 1), kb = (M - k) & (M - 1);
      const cplx<T> w = tw[k];
      double acc = 0.0;
      for (int g = 0; g < G && m0 + g < a.nseg; ++g) {
        const cplx<T>* __restrict__ d = data + (size_t)g * TILE;
        const cplx<T> za = d[(ka & (R - 1)) * RS + (ka >> LOG2R)], zb = d[(kb & (R - 1)) * RS + (kb >> LOG2R)];
        const cplx<T> e = mk<T>(T(0.5) * (za.x + zb.x), T(0.5) * (za.y - zb.y));
        const cplx<T> o = mk<T>(T(0.5) * (za.x - zb.x), T(0.5) * (za.y + zb.y));
        const cplx<T> wo = cmul(o, w);
        const T xr = e.x + wo.y, xi = e.y - wo.x;
        acc += (double)(xr * xr + xi * xi);
      }
      part[k] = acc;
    }
    return;
  }
  for (int q = tid; q < (nf << lg); q += kStftThreads) {
    const int k = q >> lg, g = q & (G - 1);
    const int64_t m = m0 + g;
    if (m >= a.nseg) continue;
    const cplx<T>* __restrict__ d = data + (size_t)g * TILE;
    const int ka = k & (M - 1), kb = (M - k) & (M - 1);
    const cplx<T> za = d[(ka & (R - 1)) * RS + (ka >> LOG2R)], zb = d[(kb & (R - 1)) * RS + (kb >> LOG2R)];
    const cplx<T> e = mk<T>(T(0.5) * (za.x + zb.x), T(0.5) * (za.y - zb.y));   // (Z[k] + conj Z[M-k]) / 2
    const cplx<T> o = mk<T>(T(0.5) * (za.x - zb.x), T(0.5) * (za.y + zb.y));   // (Z[k] - conj Z[M-k]) / 2
    const cplx<T> wo = cmul(o, tw[k]);
    cplx<T> X = mk<T>(e.x + wo.y, e.y - wo.x);  // e - i (w o)
    X.x *= scale;
    X.y *= scale;
    if (a.roll) {  // the slice rotated left by `roll` samples: X[k] exp(2 pi i k roll / nfft), phase from the table (exact index)
      const int j = (int)(((int64_t)k * a.roll) & (2 * M - 1));
      const cplx<T> r = j < M ? mk<T>(tw[j].x, -tw[j].y) : mk<T>(-tw[j - M].x, tw[j - M].y);
      X = cmul(X, r);
    }
    const int64_t at = (c * nf + k) * a.nseg + m;
    if (Z) Z[at] = X;
    if (bits) {
      const T p = X.x * X.x + X.y * X.y;
      bits[at] = a.real_kind == 0 ? log2_t(sqrt_t(p) + eps) : (a.real_kind == 1 ? sqrt_t(p) : p);
    }
  }
}

// Inverse of the ShortTimeFFT-convention transform (qi_sliding_istft; ref utilities/short_time_fft.py:106-137, scipy's
// ShortTimeFFT.istft) in ONE kernel instead of un-transpose -> batched C2R FFT -> overlap-add (round 4).  A workgroup owns
// GOWN consecutive hops of the output and holds the GOWN + H slices that cover them (H = ceil(seg / hop) - 1 slices of the
// previous workgroup's range are transformed again: the gather form of the overlap-add, no atomics, fixed order).  Per
// slice: the one-sided spectrum X[0..M] (M = nfft / 2; the imaginary parts of X[0] and X[M] are dropped as irfft does)
// is folded into the M complex numbers Z'[k] = A[k] + i B[k], A[k] = X[k] + conj X[M - k], B[k] = (X[k] - conj X[M - k])
// exp(+i pi k / M), whose unnormalised inverse M-point transform is x[2 m] + i x[2 m + 1] -- the mirror image of the forward
// kernel's untangling, same four-step transform in LDS with the conjugate twiddles.  Output sample k = the sum over its
// slices q of x_q[(k - first - q hop - roll) mod nfft] dual[k - first - q hop], times 1 / nfft.
struct IstftArgs {
  int64_t seg, hop, nseg, first, roll, k0, k1;
  int32_t gown, halo, ngroups, log2gp;  // log2gp: log2 of the slice count of a workgroup rounded up to a power of two
  int32_t per_xcd;                      // work items (record, group) per XCD
  int64_t nitems;
};
template <typename T, int LOG2R, int LOG2C>
__global__ void __launch_bounds__(kStftThreads) k_istft_fused(const cplx<T>* __restrict__ S, const T* __restrict__ dual,
                                                              const cplx<T>* __restrict__ twg, T* __restrict__ out, IstftArgs a) {
  extern __shared__ __align__(16) unsigned char lds_raw[];
  constexpr int R = 1 << LOG2R, C = 1 << LOG2C, M = R * C, RS = C + 1, TILE = R * RS + 1;
  const int G = a.gown + a.halo;
  cplx<T>* __restrict__ data = reinterpret_cast<cplx<T>*>(lds_raw);
  cplx<T>* __restrict__ tw = data + (size_t)G * TILE;
  const int tid = threadIdx.x;
  // XCD-aware mapping as in k_stft_fused: every XCD takes a contiguous range of the (record, group) items, so that the 56- /
  // 64-byte pieces of neighbouring groups, which share 128-byte lines of a frequency row, meet in one L2
  const int64_t item = (int64_t)(blockIdx.x & 7) * a.per_xcd + (blockIdx.x >> 3);
  if ((int)(blockIdx.x >> 3) >= a.per_xcd || item >= a.nitems) return;
  const int64_t c = item / a.ngroups, nf = M + 1;
  const int64_t m_own = (item % a.ngroups) * a.gown, m_lo = m_own - a.halo;  // slices m_lo .. m_lo + G - 1 (some may not exist)
  for (int k = tid; k <= M; k += kStftThreads) tw[k] = twg[k];
  // fold: thread = (bin pair (k, M - k), slice): consecutive threads read consecutive slices of a frequency row
  // (slices along the low bits of the thread index, padded to a power of two: no division by G)
  const int lgp = a.log2gp, gs = tid & ((1 << lgp) - 1);  // this thread's slice of the workgroup
  const int64_t m = m_lo + gs;
  const bool live = gs < G && m >= 0 && m < a.nseg;
  const cplx<T>* __restrict__ Sa = S + c * nf * a.nseg + m;
  // (the loads of UN sweeps are issued before any of them is used: the loop is bound by their latency)
  constexpr int UN = sizeof(T) == 4 ? 8 : 4;
  const int kstep = kStftThreads >> lgp;
  cplx<T>* __restrict__ dg = data + (size_t)gs * TILE;
  for (int kb0 = tid >> lgp; gs < G && kb0 <= M / 2; kb0 += UN * kstep) {
    cplx<T> xa[UN], xb[UN], wk[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int k = kb0 + u * kstep;
      const bool on = live && k <= M / 2;
      xa[u] = on ? Sa[(int64_t)k * a.nseg] : mk<T>(T(0), T(0));
      xb[u] = on ? Sa[(int64_t)(M - k) * a.nseg] : mk<T>(T(0), T(0));
      wk[u] = twg[k <= M / 2 ? k : 0];  // exp(-i pi k / M): B = D conj(w)
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int k = kb0 + u * kstep;
      if (k > M / 2) break;
      if (k == 0) {
        xa[u].y = T(0);
        xb[u].y = T(0);
      }
      const cplx<T> w = wk[u];
      const cplx<T> A = mk<T>(xa[u].x + xb[u].x, xa[u].y - xb[u].y), D = mk<T>(xa[u].x - xb[u].x, xa[u].y + xb[u].y);
      const cplx<T> B = mk<T>(D.x * w.x + D.y * w.y, D.y * w.x - D.x * w.y);
      dg[(k >> LOG2C) * RS + (k & (C - 1))] = mk<T>(A.x - B.y, A.y + B.x);  // A + i B
      if (k > 0 && 2 * k != M) {
        const int kb = M - k;
        dg[(kb >> LOG2C) * RS + (kb & (C - 1))] = mk<T>(A.x + B.y, B.x - A.y);  // conj(A - i B)
      }
    }
  }
  __syncthreads();
  // step 1: columns (inverse sign: conjugate twiddles)
  for (int q = tid; q < G * C; q += kStftThreads) {
    const int g = q >> LOG2C, cc = q & (C - 1);
    cplx<T>* __restrict__ d = data + (size_t)g * TILE + cc;
    cplx<T> v[R];
#pragma unroll
    for (int r = 0; r < R; ++r) v[r] = d[r * RS];
    native::fft_reg<T, R, 1>(v);
#pragma unroll
    for (int k1 = 0; k1 < R; ++k1) {
      cplx<T> y = v[native::brev(k1, LOG2R)];
      if (k1 > 0) {
        const int j2 = 2 * cc * k1;
        const cplx<T> w = tw[j2 & (M - 1)];
        y = cmul(y, (j2 & M) ? mk<T>(-w.x, w.y) : mk<T>(w.x, -w.y));
      }
      d[k1 * RS] = y;
    }
  }
  __syncthreads();
  // step 2: rows; row k1, column k2 then holds the pair e = k1 + R k2: (x[2 e], x[2 e + 1])
  for (int q = tid; q < G * R; q += kStftThreads) {
    const int g = q >> LOG2R, k1 = q & (R - 1);
    cplx<T>* __restrict__ d = data + (size_t)g * TILE + k1 * RS;
    cplx<T> v[C];
#pragma unroll
    for (int j = 0; j < C; ++j) v[j] = d[j];
    native::fft_reg<T, C, 1>(v);
#pragma unroll
    for (int k2 = 0; k2 < C; ++k2) d[k2] = v[native::brev(k2, LOG2C)];
  }
  __syncthreads();
  // overlap-add of the owned hops: sample k' = k - first in [m_own hop, (m_own + gown) hop)
  // (hop by hop, sample j of hop h: its slices are q = m_own + h, m_own + h - 1, ... at offsets i = j, j + hop, ... < seg, in
  // ascending q like the three-kernel path -- no division)
  const T inv = T(1) / (T)(2 * M);
  const T* __restrict__ flat = reinterpret_cast<const T*>(data);
  const int hop = (int)a.hop, seg = (int)a.seg, roll = (int)a.roll;
  for (int h = 0; h < a.gown; ++h) {
    const int64_t qh = m_own + h;
    for (int j = tid; j < hop; j += kStftThreads) {
      const int64_t k = qh * a.hop + j + a.first;
      if (k < a.k0 || k >= a.k1) continue;
      int dmax = (seg - 1 - j) / hop;  // slices qh - dmax .. qh cover the sample
      if (dmax > qh) dmax = (int)qh;
      T acc = T(0);
      for (int dq = dmax; dq >= 0; --dq) {
        const int64_t q = qh - dq;
        if (q > a.nseg - 1) continue;
        const int i = j + dq * hop;
        int dd = i - roll;
        if (dd < 0) dd += 2 * M;
        const int e = dd >> 1, gs = (int)(q - m_lo);
        acc += flat[2 * ((size_t)gs * TILE + (e & (R - 1)) * RS + (e >> LOG2R)) + (dd & 1)] * dual[i];
      }
      out[c * (a.k1 - a.k0) + (k - a.k0)] = acc * inv;
    }
  }
}

// Pxx[c][f] = w_f scale^2 / nseg * sum over the groups (index order) of their partial sums, w_f = 2 except at DC and Nyquist
template <typename T>
__global__ void k_welch_reduce(const double* __restrict__ part, T* __restrict__ pxx, int ngroups, int64_t nseg, int nf,
                               T scale2) {
  const int f = blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t c = blockIdx.y;
  if (f >= nf) return;
  double acc = 0.0;
  for (int g = 0; g < ngroups; ++g) acc += part[((size_t)c * ngroups + g) * nf + f];
  const bool paired = f > 0 && f != nf - 1;  // (nfft is even here)
  pxx[c * nf + f] = (T)(acc / (double)nseg) * scale2 * (paired ? T(2) : T(1));
}

}  // namespace

// M = R x C per transform length (R <= C <= 64: register transforms of at most 64 points; float64 keeps to 32)
template <typename T>
static bool stft_shape(int64_t M, int* lr, int* lc) {
  int lm = 0;
  while ((1ll << lm) < M) ++lm;
  *lr = lm / 2;
  *lc = lm - *lr;
  return (1ll << lm) == M && *lc <= (sizeof(T) == 8 ? 5 : 6) && *lr >= 2;
}

// segments per workgroup (a power of two <= 16) so that the tiles and the twiddles stay within `budget` bytes of LDS
static int stft_fused_group(int64_t M, int lr, int lc, size_t esz, size_t budget) {
  const size_t tile = ((size_t)1 << lr) * (((size_t)1 << lc) + 1) + 1;
  for (int G = 16; G >= 1; G >>= 1)
    if (((size_t)G * tile + M + 1) * esz <= budget) return G;
  return 0;
}

bool stft_fused_supported(int dtype, int64_t seg, int64_t hop, int64_t nfft) {
  int lr, lc;
  if (!(nfft >= 64 && nfft <= 4096 && (nfft & (nfft - 1)) == 0 && seg <= nfft && seg >= 2 && hop >= 1)) return false;
  return dtype == QI_F64 ? stft_shape<double>(nfft / 2, &lr, &lc) : stft_shape<float>(nfft / 2, &lr, &lc);
}

// exp(-i pi k / M), k = 0..M, per device and transform length (built on the host in long double, kept for the process)
template <typename T>
static int stft_twiddles(int64_t M, const cplx<T>** out) {
  static std::mutex mu;
  static std::map<std::pair<int, int64_t>, cplx<T>*> tables;
  int dev = 0;
  QI_HIP(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lk(mu);
  auto it = tables.find({dev, M});
  if (it == tables.end()) {
    std::vector<cplx<T>> h((size_t)M + 1);
    for (int64_t k = 0; k <= M; ++k) {
      const long double ph = 3.14159265358979323846264338327950288L * (long double)k / (long double)M;
      h[(size_t)k] = mk<T>((T)cosl(ph), (T)-sinl(ph));
    }
    h[(size_t)M] = mk<T>(T(-1), T(0));
    if (M % 2 == 0) h[(size_t)(M / 2)] = mk<T>(T(0), T(-1));
    cplx<T>* d = nullptr;
    QI_HIP(hipMalloc(reinterpret_cast<void**>(&d), h.size() * sizeof(cplx<T>)));
    QI_HIP(hipMemcpy(d, h.data(), h.size() * sizeof(cplx<T>), hipMemcpyHostToDevice));
    it = tables.emplace(std::make_pair(dev, M), d).first;
  }
  *out = it->second;
  return QI_OK;
}

static thread_local int32_t g_last_ngroups = 0;  // segment groups per record of this thread's last launch (Welch partials)

template <typename T, int LR, int LC>
static int launch_stft_shape(const T* sig, const T* win, cplx<T>* Z, T* bits, int64_t C, int64_t nseg, StftFusedArgs a,
                             hipStream_t st) {
  const int64_t M = 1ll << (LR + LC);
  size_t budget = 80 * 1024;  // two workgroups per CU
  if (const char* e = tune_env("QI_STFT_LDS_KB")) budget = (size_t)atoi(e) * 1024;
  const int G = stft_fused_group(M, LR, LC, sizeof(cplx<T>), budget);
  if (G < 1) {
    set_error("fused STFT: a transform of %lld points does not fit the LDS tile", (long long)(2 * M));
    return QI_ERR_UNSUPPORTED;
  }
  a.G = G;
  a.log2g = 0;
  while ((1 << a.log2g) < G) ++a.log2g;
  const size_t tile = ((size_t)1 << LR) * (((size_t)1 << LC) + 1) + 1;
  const size_t lds = ((size_t)G * tile + M + 1) * sizeof(cplx<T>);
  const cplx<T>* twg = nullptr;
  QI_TRY(stft_twiddles<T>(M, &twg));
  // the product's own call (styx_fft: zeros beyond the record, both panels, log2 bits) runs the specialised loops
  const bool plain = Z && bits && !a.welch_part && a.pad_mode == 0 && a.real_kind == 0 && a.roll == 0;
  const void* fn = plain ? reinterpret_cast<const void*>(&k_stft_fused<T, LR, LC, true>)
                         : reinterpret_cast<const void*>(&k_stft_fused<T, LR, LC, false>);
  QI_TRY(allow_dynamic_lds(fn, lds));
  a.ngroups = (int32_t)ceil_div(nseg, G);
  g_last_ngroups = a.ngroups;
  a.nitems = (int64_t)a.ngroups * C;
  a.per_xcd = (int32_t)ceil_div(a.nitems, 8);
  dim3 grid((unsigned)(8 * a.per_xcd));
  if (plain)
    k_stft_fused<T, LR, LC, true><<<grid, kStftThreads, lds, st>>>(sig, win, twg, Z, bits, a);
  else
    k_stft_fused<T, LR, LC, false><<<grid, kStftThreads, lds, st>>>(sig, win, twg, Z, bits, a);
  QI_LAUNCH_CHECK();
  return QI_OK;
}

template <typename T>
int launch_stft_fused(const T* sig, const T* win, cplx<T>* Z, T* bits, int64_t C, int64_t n, int64_t seg, int64_t hop,
                      int64_t nfft, int64_t nseg, int64_t lead, double scale, double eps, hipStream_t st,
                      double* welch_part, const StftSliding* sl) {
  StftFusedArgs a;
  a.welch_part = welch_part;
  a.pad_mode = sl ? sl->pad_mode : 0;
  a.detrend = sl ? sl->detrend : 1;
  a.real_kind = sl ? sl->real_kind : 0;
  a.roll = sl ? sl->roll : 0;
  a.dbg = 0;
#ifdef QI_STFT_DBG
  if (const char* e = tune_env("QI_STFT_DBG")) a.dbg = atoi(e);
#endif
  a.n = n;
  a.seg = seg;
  a.hop = hop;
  a.nseg = nseg;
  a.lead = lead;
  a.scale = scale;
  a.eps = eps;
  a.G = a.log2g = 0;
  int lr, lc;
  if (!stft_shape<T>(nfft / 2, &lr, &lc)) {
    set_error("fused STFT: transform length %lld is not supported", (long long)nfft);
    return QI_ERR_UNSUPPORTED;
  }
  switch (lr * 8 + lc) {  // M = 32 ... 2048 (float64: ... 1024)
    case 2 * 8 + 3: return launch_stft_shape<T, 2, 3>(sig, win, Z, bits, C, nseg, a, st);
    case 3 * 8 + 3: return launch_stft_shape<T, 3, 3>(sig, win, Z, bits, C, nseg, a, st);
    case 3 * 8 + 4: return launch_stft_shape<T, 3, 4>(sig, win, Z, bits, C, nseg, a, st);
    case 4 * 8 + 4: return launch_stft_shape<T, 4, 4>(sig, win, Z, bits, C, nseg, a, st);
    case 4 * 8 + 5: return launch_stft_shape<T, 4, 5>(sig, win, Z, bits, C, nseg, a, st);
    case 5 * 8 + 5: return launch_stft_shape<T, 5, 5>(sig, win, Z, bits, C, nseg, a, st);
    case 5 * 8 + 6:
      if constexpr (sizeof(T) == 4) return launch_stft_shape<T, 5, 6>(sig, win, Z, bits, C, nseg, a, st);
      break;
    default: break;
  }
  set_error("fused STFT: transform length %lld is not supported", (long long)nfft);
  return QI_ERR_UNSUPPORTED;
}

template int launch_stft_fused<float>(const float*, const float*, float2*, float*, int64_t, int64_t, int64_t, int64_t,
                                      int64_t, int64_t, int64_t, double, double, hipStream_t, double*, const StftSliding*);
template int launch_stft_fused<double>(const double*, const double*, double2*, double*, int64_t, int64_t, int64_t, int64_t,
                                       int64_t, int64_t, int64_t, double, double, hipStream_t, double*, const StftSliding*);

// Welch power spectrum on the fused kernel: partial sums per segment group in `part` ([C][groups][nfft / 2 + 1] doubles,
// groups <= nseg), then the mean over the segments with scipy's one-sided "spectrum" weights
template <typename T>
int launch_welch_fused(const T* sig, const T* win, T* pxx, double* part, int64_t C, int64_t n, int64_t seg, int64_t hop,
                       int64_t nfft, int64_t nseg, double scale2, hipStream_t st) {
  QI_TRY(launch_stft_fused<T>(sig, win, nullptr, nullptr, C, n, seg, hop, nfft, nseg, 0, 1.0, 0.0, st, part, nullptr));
  const int nf = (int)(nfft / 2 + 1);
  dim3 g((unsigned)ceil_div(nf, 256), (unsigned)C);
  k_welch_reduce<T><<<g, 256, 0, st>>>(part, pxx, g_last_ngroups, nseg, nf, (T)scale2);
  QI_LAUNCH_CHECK();
  return QI_OK;
}
template int launch_welch_fused<float>(const float*, const float*, float*, double*, int64_t, int64_t, int64_t, int64_t, int64_t,
                                       int64_t, double, hipStream_t);
template int launch_welch_fused<double>(const double*, const double*, double*, double*, int64_t, int64_t, int64_t, int64_t,
                                        int64_t, int64_t, double, hipStream_t);

// the fused inverse: supported for the fused kernel's transform lengths when a workgroup's LDS holds at least one owned hop
// beside the halo slices
template <typename T, int LR, int LC>
static int launch_istft_shape(const cplx<T>* S, const T* dual, T* out, int64_t C, IstftArgs a, hipStream_t st) {
  const int64_t M = 1ll << (LR + LC);
  const size_t tile = ((size_t)1 << LR) * (((size_t)1 << LC) + 1) + 1, budget = 80 * 1024;
  int G = (int)((budget / sizeof(cplx<T>) - (size_t)(M + 1)) / tile);
  if (G > 16) G = 16;
  if (G - a.halo < 1) return QI_ERR_UNSUPPORTED;
  a.gown = G - a.halo;
  // owned hops: every sample up to the end of the last slice
  const int64_t hops = a.nseg + (a.seg - 1) / a.hop;
  if (hops < a.gown) a.gown = (int32_t)hops;
  a.ngroups = (int32_t)ceil_div(hops, a.gown);
  a.log2gp = 0;
  while ((1 << a.log2gp) < a.gown + a.halo) ++a.log2gp;
  const size_t lds = ((size_t)(a.gown + a.halo) * tile + M + 1) * sizeof(cplx<T>);
  const cplx<T>* twg = nullptr;
  QI_TRY(stft_twiddles<T>(M, &twg));
  QI_TRY(allow_dynamic_lds(reinterpret_cast<const void*>(&k_istft_fused<T, LR, LC>), lds));
  if (a.k0 < a.first || a.k1 > a.first + hops * a.hop)  // (samples outside the hops the workgroups own)
    QI_HIP(hipMemsetAsync(out, 0, (size_t)C * (size_t)(a.k1 - a.k0) * sizeof(T), st));
  a.nitems = (int64_t)a.ngroups * C;
  a.per_xcd = (int32_t)ceil_div(a.nitems, 8);
  k_istft_fused<T, LR, LC><<<dim3((unsigned)(8 * a.per_xcd)), kStftThreads, lds, st>>>(S, dual, twg, out, a);
  QI_LAUNCH_CHECK();
  return QI_OK;
}

template <typename T>
int launch_istft_fused(const cplx<T>* S, const T* dual, T* out, int64_t C, int64_t seg, int64_t hop, int64_t nfft, int64_t first,
                       int64_t nseg, int64_t roll, int64_t k0, int64_t k1, hipStream_t st) {
  int lr, lc;
  if (!(nfft >= 64 && nfft <= 4096 && (nfft & (nfft - 1)) == 0 && seg <= nfft && seg >= 2 && hop >= 1 && hop <= seg) ||
      !stft_shape<T>(nfft / 2, &lr, &lc))
    return QI_ERR_UNSUPPORTED;
  IstftArgs a{};
  a.seg = seg;
  a.hop = hop;
  a.nseg = nseg;
  a.first = first;
  a.roll = roll;
  a.k0 = k0;
  a.k1 = k1;
  a.halo = (int32_t)(ceil_div(seg, hop) - 1);
  switch (lr * 8 + lc) {
    case 2 * 8 + 3: return launch_istft_shape<T, 2, 3>(S, dual, out, C, a, st);
    case 3 * 8 + 3: return launch_istft_shape<T, 3, 3>(S, dual, out, C, a, st);
    case 3 * 8 + 4: return launch_istft_shape<T, 3, 4>(S, dual, out, C, a, st);
    case 4 * 8 + 4: return launch_istft_shape<T, 4, 4>(S, dual, out, C, a, st);
    case 4 * 8 + 5: return launch_istft_shape<T, 4, 5>(S, dual, out, C, a, st);
    case 5 * 8 + 5: return launch_istft_shape<T, 5, 5>(S, dual, out, C, a, st);
    case 5 * 8 + 6:
      if constexpr (sizeof(T) == 4) return launch_istft_shape<T, 5, 6>(S, dual, out, C, a, st);
      break;
    default: break;
  }
  return QI_ERR_UNSUPPORTED;
}
template int launch_istft_fused<float>(const float2*, const float*, float*, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t,
                                       int64_t, int64_t, int64_t, hipStream_t);
template int launch_istft_fused<double>(const double2*, const double*, double*, int64_t, int64_t, int64_t, int64_t, int64_t,
                                        int64_t, int64_t, int64_t, int64_t, hipStream_t);

}  // namespace qi
