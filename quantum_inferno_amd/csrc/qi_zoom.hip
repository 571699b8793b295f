// Zoom engine: the narrow-spectrum bands of a panel (long atoms: a few hundred to a few ten thousand occupied bins
// out of a million) are slowly varying envelopes on a carrier.  Their occupied bins, moved to baseband, are
// transformed on a COARSE time grid on which the band is oversampled at least 4 times (one small inverse FFT per
// band: k_zoom_gather + k_zoom_coarse), and the panel is produced from that by band-limited interpolation -- N-tap
// interpolators that are exact at the Chebyshev nodes of the band (10 taps at 4 x oversampling: 9e-8 of a unit tone
// anywhere in the band; 6 and 4 taps for bands oversampled 8 and 32 times: see zoom_weights) -- times the carrier
// phasor.  About 35-60 instructions per output instead of a
// share of a million-point transform, nothing discarded (the zero-padded half of the linear correlation is simply not
// evaluated), no intermediate: the kernel is bound by the panel write.
//
// The coarse grid has D = 64 >> level fine samples per coarse sample (level 0: the narrowest bands, D = 64; each
// level doubles the bandwidth that can be carried).  One wave-step = 64 consecutive outputs = the 64 lanes of a wave;
// it spans S = 1 << level coarse intervals, so its window holds N + S - 1 coarse samples, which are uniform over the
// wave (scalar registers, taken from a vector register by v_readlane as the window slides); each lane carries the
// N + S - 1 weights of its own position in the window (registers; zero outside its N taps).  Every store is a 512-byte
// run.  Waves only meet for the per-band power sum (one barrier per band).  No MFMA: there is no dense contraction.
#include "qi_common.hpp"
#include "qi_device.hpp"
#include "qi_native.hpp"
#include "qi_fft_reg.hpp"
#include "qi_zoom_gather.hpp"

namespace qi {
namespace native {

namespace {

constexpr int kZoomThreads = 256;

__device__ __forceinline__ float lane_value(float v, int lane) {
  return __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v), lane));
}

// One thread per (kappa0, tau1) of the coarse stage's input (zoom_gather_value), written to the band's planes
// [tau1][kappa0], transformed in place by the coarse stage -- the stand-alone form; qi_cwt_stx's joint launch gathers
// inside the plane transforms (k_zoom_coarse2g, qi_block.hip).
template <typename T, bool STX>
__device__ __forceinline__ void zoom_gather_plane(const ZoomArgs<T>& a, const uint32_t plane) {
  const BandDesc bd = load_uniform(a.bands + *as_const(a.plane_band + plane));  // plane of the record's coarse storage
  const int32_t kappa0 = (int32_t)(blockIdx.x * 256 + threadIdx.x);
  const uint32_t tau1 = plane - (uint32_t)bd.edge;
  const int64_t ch = blockIdx.z;
  const cplx<T>* __restrict__ X = a.X + ch * (a.Lf << a.x_shift);
  a.coarse[((int64_t)ch * a.planes + bd.edge) * kBlk + (int32_t)tau1 * kBlk + kappa0] =
      zoom_gather_value<T, STX>(a, bd, tau1, kappa0, X);
}
template <typename T, bool STX>
__global__ void __launch_bounds__(256) k_zoom_gather(ZoomArgs<T> a) {
  zoom_gather_plane<T, STX>(a, blockIdx.y);
}
// qi_cwt_stx: the planes of the styx table (a0) and of the Stockwell table (a2) in one launch
template <typename T>
__global__ void __launch_bounds__(256) k_zoom_gather2(ZoomArgs<T> a0, ZoomArgs<T> a2) {
  if (blockIdx.y < (uint32_t)a0.planes) zoom_gather_plane<T, false>(a0, blockIdx.y);
  else zoom_gather_plane<T, true>(a2, blockIdx.y - (uint32_t)a0.planes);
}

// Fine stage of one level.  PHASOR: multiply by the carrier exp(2 pi i k_c f / Lf) (Gabor banks; the Stockwell bands
// are at baseband already).  Output sample t is the full-length sample f = t + off, off = 64 A - e (e = 0 or 1).
// SDESC: the band descriptors by scalar loads (load_uniform: no vector registers for the descriptor fetched a band ahead;
// -3 % of the interpolation launch at 64 records).  With one or two records a workgroup has few bands and the first
// descriptor's scalar-cache miss shows (+1 % of the step): those calls keep the vector loads.
template <typename T, int LEVEL, bool PHASOR, bool COEF, bool BITS, bool SDESC>
__device__ __forceinline__ void zoom_level(const ZoomArgs<T>& a, int chunk, int row, double (*s_red)[kZoomThreads / kWave],
                                           double (*s_fin)[kZoomThreads / kWave]) {
  constexpr int NW = kZoomThreads / kWave, S = zoom_span(LEVEL), TAPS = zoom_taps(LEVEL), STEPS = zoom_steps(LEVEL);
  if ((int64_t)blockIdx.x >= a.n / ((int64_t)kZoomD * STEPS * NW)) return;  // this level has fewer groups along time
  const int nchunk = a.lvl_nchunk[LEVEL];
  const float* __restrict__ weights = a.lvl_weights[LEVEL];
  constexpr int GRID = zoom_grid(LEVEL), HALF = zoom_ntap(LEVEL) / 2 - 1;
  constexpr int WIN = (STEPS - 1) * S + TAPS;  // coarse samples one wave needs per band
  static_assert(WIN <= 2 * kWave, "the window of one wave must fit two registers of its lanes");
  constexpr bool TWO = WIN > kWave;  // the window spills into a second vector register
  const int tid = threadIdx.x, lane = tid & (kWave - 1), wv = tid / kWave;
  const int64_t ch = blockIdx.z;
  const int64_t gw = (int64_t)blockIdx.x * NW + wv;  // wave index along time
  const uint32_t step_a = (uint32_t)gw * STEPS;      // first wave-step of this wave
  const uint32_t mmask = (uint32_t)((a.Lf / kZoomD) << GRID) - 1u, lmask = (uint32_t)a.Lf - 1u;
  const int plog2 = ilog2((int)(((a.Lf / kZoomD) << GRID) / kBlk));  // log2 of the planes per band
  float wgt[TAPS];
#pragma unroll
  for (int j = 0; j < TAPS; ++j) wgt[j] = weights[j * kWave + lane];  // [tap][lane]: coalesced
  T colp[STEPS];
#pragma unroll
  for (int s = 0; s < STEPS; ++s) colp[s] = T(0);
  T mx = T(0);
  double plogp = 0.0;
  const uint32_t t_base = step_a * kZoomD + (uint32_t)lane;  // output sample of wave-step s: t_base + 64 s
  int par = 0;  // s_red is double-buffered so that one barrier per band is enough

  // lane i holds coarse sample S (step_a + A) - HALF + i of the band: wave-step s interpolates from lanes s S ..
  // s S + TAPS - 1.  Coarse sample tau = P tau2 + tau1 sits at plane tau1, position tau2 (the coarse stage's layout).
  // The samples (and the descriptor) of the next band are requested a band ahead.
  const uint32_t wtau = ((uint32_t)S * (step_a + (uint32_t)a.tau_off) - (uint32_t)HALF + (uint32_t)lane) & mmask;
  const uint32_t widx = (wtau & ((1u << plog2) - 1u)) * (uint32_t)kBlk + (wtau >> plog2);
  const uint32_t wtau2 = (wtau + (uint32_t)kWave) & mmask;  // window samples 64 .. 127 (wide windows only)
  const uint32_t widx2 = (wtau2 & ((1u << plog2) - 1u)) * (uint32_t)kBlk + (wtau2 >> plog2);
  const int jj0 = a.lvl_first[LEVEL] + chunk, jj_end = a.lvl_first[LEVEL] + a.lvl_count[LEVEL];
  cplx<T> smp_next = mk<T>(T(0), T(0)), smq_next = mk<T>(T(0), T(0));
  const BandDesc* first_bd = a.bands + (jj0 < jj_end ? jj0 : a.lvl_first[LEVEL]);
  BandDesc bd_next = SDESC ? load_uniform(first_bd) : *first_bd;
  if (jj0 < jj_end) {
    const cplx<T>* __restrict__ b = a.coarse + ((int64_t)ch * a.planes + bd_next.edge) * kBlk;
    smp_next = b[widx];
    if (TWO) smq_next = b[widx2];
  }
  for (int jj = jj0; jj < jj_end; jj += nchunk) {
    const BandDesc bd = bd_next;
    const cplx<T> smp = smp_next, smq = smq_next;
    if (jj + nchunk < jj_end) {
      bd_next = SDESC ? load_uniform(a.bands + jj + nchunk) : a.bands[jj + nchunk];
      const cplx<T>* __restrict__ b = a.coarse + ((int64_t)ch * a.planes + bd_next.edge) * kBlk;
      smp_next = b[widx];
      if (TWO) smq_next = b[widx2];
    }
    cplx<T> P = mk<T>(T(1), T(0)), Q = mk<T>(T(1), T(0));
    if (PHASOR) {
      // carrier: exp(2 pi i kc f / Lf), f = 64 (step + A) + lane - e: per-lane factor P, per-step factor Q (lane i
      // holds the factor of wave-step i); the phases are exact integers modulo Lf
      const uint32_t kc = (uint32_t)(bd.k_lo + bd.k_len / 2);
      double c, s;
      unit_root((kc * (uint32_t)(lane - a.lane_off)) & lmask, a.two_over_len, &c, &s);
      P = mk<T>((T)c, (T)s);
      unit_root((kc * kZoomD * (step_a + (uint32_t)a.tau_off + (uint32_t)lane)) & lmask, a.two_over_len, &c, &s);
      Q = mk<T>((T)c, (T)s);
    }
    const int64_t orow = ((int64_t)ch * a.panel_bands + bd.out_band) * a.n;
    // split band (bd.add_row): this is only the tapered part of its atom -- the samples go to row add_row - 1 of
    // split_part and count for nothing here; the edge items of the block launch add their part and finish the band
    const bool part = bd.add_row != 0;
    char* __restrict__ coef_row = reinterpret_cast<char*>(
        part ? a.split_part + ((int64_t)ch * a.split_rows + (bd.add_row - 1)) * a.n : (a.coef ? a.coef + orow : nullptr));
    char* __restrict__ bits_row = reinterpret_cast<char*>(a.bits ? a.bits + orow : nullptr);
    const T pscale = part ? T(0) : a.power_scale;
    uint32_t tb = t_base;
    asm volatile("" : "+v"(tb));  // keep the band-invariant addresses out of the loop-invariant hoisting
    T rowacc = T(0), pl = T(0);
    // interpolation: real parts of all wave-steps, then imaginary parts (the wave-uniform coarse samples sit in
    // scalar registers; one part at a time keeps them within budget)
    T zr[STEPS], zi[STEPS];
    {
      float sx[WIN];
#pragma unroll
      for (int i = 0; i < WIN; ++i) sx[i] = i < kWave ? lane_value(smp.x, i) : lane_value(smq.x, i - kWave);
#pragma unroll
      for (int s = 0; s < STEPS; ++s) {
        T acc = T(0);
#pragma unroll
        for (int j = 0; j < TAPS; ++j) acc = fmaf(wgt[j], sx[s * S + j], acc);
        zr[s] = acc;
      }
    }
    {
      float sy[WIN];
#pragma unroll
      for (int i = 0; i < WIN; ++i) sy[i] = i < kWave ? lane_value(smp.y, i) : lane_value(smq.y, i - kWave);
#pragma unroll
      for (int s = 0; s < STEPS; ++s) {
        T acc = T(0);
#pragma unroll
        for (int j = 0; j < TAPS; ++j) acc = fmaf(wgt[j], sy[s * S + j], acc);
        zi[s] = acc;
      }
    }
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
      cplx<T> z = mk<T>(zr[s], zi[s]);
      if (PHASOR) {
        const cplx<T> q = mk<T>(lane_value(Q.x, s), lane_value(Q.y, s));
        z = cmul_rn(z, cmul_rn(P, q));
      }
      const uint32_t tt = tb + (uint32_t)(kZoomD * s);
      if (COEF || part) stream_store(reinterpret_cast<cplx<T>*>(coef_row + (size_t)(tt * (uint32_t)sizeof(cplx<T>))), z);
      const T m2 = norm2(z.x, z.y);
      if (BITS) *reinterpret_cast<T*>(bits_row + (size_t)(tt * (uint32_t)sizeof(T))) = log2_t(sqrt_t(m2) + a.eps);
      const T p = mul_rn(pscale, m2);
      colp[s] += p;
      rowacc += p;
      mx = max_t(mx, p);
      pl += plog2p(p);
    }
    plogp += (double)pl;
    if (a.part_band) {
      // one partial per workgroup and band: wave sums through LDS
      const double r = wave_sum((double)rowacc);
      if (lane == 0) s_red[par][wv] = r;
      __syncthreads();
      if (tid == 0 && !part) {
        double t = 0.0;
        for (int q = 0; q < NW; ++q) t += s_red[par][q];
        a.part_band[((int64_t)ch * a.panel_bands + bd.out_band) * a.nblk + blockIdx.x] = t;
      }
      par ^= 1;
    }
  }

  T tot = T(0);
  char* __restrict__ time_row = reinterpret_cast<char*>(
      a.time_part ? a.time_part + ((int64_t)ch * a.chunk_total + a.chunk_base + row) * a.n : nullptr);
#pragma unroll
  for (int s = 0; s < STEPS; ++s) {
    tot += colp[s];
    const uint32_t tt = t_base + (uint32_t)(kZoomD * s);
    if (time_row) *reinterpret_cast<T*>(time_row + (size_t)(tt * (uint32_t)sizeof(T))) = colp[s];
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
      for (int q = 0; q < NW; ++q) {
        m = s_fin[0][q] > m ? s_fin[0][q] : m;
        s1 += s_fin[1][q];
        s2 += s_fin[2][q];
      }
      const int64_t groups = a.n / ((int64_t)kZoomD * STEPS * NW);
      double* o = a.part_stat + ((int64_t)ch * a.stat_stride + a.lvl_stat_base[LEVEL] + (int64_t)chunk * groups + blockIdx.x) * 3;
      o[0] = m;
      o[1] = s1;
      o[2] = s2;
    }
  }
}

// row y of a table's fine launch = (level, chunk of the level's band list)
template <typename T, bool PHASOR, bool COEF, bool BITS, bool SDESC>
__device__ __forceinline__ void zoom_row(const ZoomArgs<T>& a, int y, double (*s_red)[kZoomThreads / kWave],
                                         double (*s_fin)[kZoomThreads / kWave]) {
  if (y < a.lvl_chunk0[0] + a.lvl_nchunk[0]) zoom_level<T, 0, PHASOR, COEF, BITS, SDESC>(a, y - a.lvl_chunk0[0], y, s_red, s_fin);
  else if (y < a.lvl_chunk0[1] + a.lvl_nchunk[1]) zoom_level<T, 1, PHASOR, COEF, BITS, SDESC>(a, y - a.lvl_chunk0[1], y, s_red, s_fin);
  else if (y < a.lvl_chunk0[2] + a.lvl_nchunk[2]) zoom_level<T, 2, PHASOR, COEF, BITS, SDESC>(a, y - a.lvl_chunk0[2], y, s_red, s_fin);
  else if (y < a.lvl_chunk0[3] + a.lvl_nchunk[3]) zoom_level<T, 3, PHASOR, COEF, BITS, SDESC>(a, y - a.lvl_chunk0[3], y, s_red, s_fin);
  else if (y < a.lvl_chunk0[4] + a.lvl_nchunk[4]) zoom_level<T, 4, PHASOR, COEF, BITS, SDESC>(a, y - a.lvl_chunk0[4], y, s_red, s_fin);
  else if (y < a.lvl_chunk0[5] + a.lvl_nchunk[5]) zoom_level<T, 5, PHASOR, COEF, BITS, SDESC>(a, y - a.lvl_chunk0[5], y, s_red, s_fin);
  else zoom_level<T, 6, PHASOR, COEF, BITS, SDESC>(a, y - a.lvl_chunk0[6], y, s_red, s_fin);
}

// one launch for every level: blockIdx.y is the row
template <typename T, bool PHASOR, bool COEF, bool BITS>
__global__ void __launch_bounds__(kZoomThreads) k_zoom(ZoomArgs<T> a) {
  __shared__ double s_red[2][kZoomThreads / kWave];
  __shared__ double s_fin[3][kZoomThreads / kWave];
  zoom_row<T, PHASOR, COEF, BITS, false>(a, blockIdx.y, s_red, s_fin);
}

// qi_cwt_stx: the rows of the styx table (a0, with carrier) and of the Stockwell table (a2) in one launch
template <typename T, bool COEF, bool BITS, bool SDESC>
__global__ void __launch_bounds__(kZoomThreads) k_zoom2(ZoomArgs<T> a0, ZoomArgs<T> a2, int rows0) {
  __shared__ double s_red[2][kZoomThreads / kWave];
  __shared__ double s_fin[3][kZoomThreads / kWave];
  if ((int)blockIdx.y < rows0) zoom_row<T, true, COEF, BITS, SDESC>(a0, blockIdx.y, s_red, s_fin);
  else zoom_row<T, false, COEF, BITS, SDESC>(a2, (int)blockIdx.y - rows0, s_red, s_fin);
}

template <typename T, bool PHASOR>
int launch_zoom_v(const ZoomArgs<T>& a, dim3 grid, hipStream_t st) {
  const bool coef = a.coef != nullptr, bits = a.bits != nullptr;
  if (coef && bits) k_zoom<T, PHASOR, true, true><<<grid, kZoomThreads, 0, st>>>(a);
  else if (coef) k_zoom<T, PHASOR, true, false><<<grid, kZoomThreads, 0, st>>>(a);
  else if (bits) k_zoom<T, PHASOR, false, true><<<grid, kZoomThreads, 0, st>>>(a);
  else k_zoom<T, PHASOR, false, false><<<grid, kZoomThreads, 0, st>>>(a);
  QI_LAUNCH_CHECK();
  return QI_OK;
}

}  // namespace

int64_t zoom_groups(int64_t n, int level) {
  return n / ((int64_t)kZoomD * zoom_steps(level) * (kZoomThreads / kWave));
}

template <>
int launch_zoom_gather<float>(const ZoomArgs<float>& a, int max_level, int64_t n_channels, hipStream_t st) {
  if (a.nbands <= 0) return QI_OK;
  (void)max_level;
  dim3 grid((unsigned)(kBlk / 256), (unsigned)a.planes, (unsigned)n_channels);
  if (a.stx) k_zoom_gather<float, true><<<grid, 256, 0, st>>>(a);
  else k_zoom_gather<float, false><<<grid, 256, 0, st>>>(a);
  QI_LAUNCH_CHECK();
  return QI_OK;
}

template <>
int launch_zoom_gather2<float>(const ZoomArgs<float>& a0, const ZoomArgs<float>& a2, int64_t n_channels, hipStream_t st) {
  if (a0.stx || !a2.stx || a0.nbands <= 0 || a2.nbands <= 0) {
    set_error("zoom engine: the joint gather takes a styx table and a Stockwell table");
    return QI_ERR_STATE;
  }
  dim3 grid((unsigned)(kBlk / 256), (unsigned)(a0.planes + a2.planes), (unsigned)n_channels);
  k_zoom_gather2<float><<<grid, 256, 0, st>>>(a0, a2);
  QI_LAUNCH_CHECK();
  return QI_OK;
}

// grid of a table's fine launch: workgroups along time (the level with the most) x rows
static int zoom_grid(const ZoomArgs<float>& a, int64_t* groups_out, int* rows_out) {
  int64_t groups = 0;
  int chunks = 0;
  for (int g = 0; g < kZoomClasses; ++g) {
    if (a.lvl_nchunk[g] <= 0) continue;
    const int64_t gg = zoom_groups(a.n, g);
    if (gg < 1 || gg * kZoomD * zoom_steps(g) * (kZoomThreads / kWave) != a.n) {
      set_error("zoom engine: record length %lld is not a multiple of %d samples", (long long)a.n,
                kZoomD * zoom_steps(g) * (kZoomThreads / kWave));
      return QI_ERR_UNSUPPORTED;
    }
    if (gg > groups) groups = gg;
    chunks = a.lvl_chunk0[g] + a.lvl_nchunk[g];
  }
  *groups_out = groups;
  *rows_out = chunks;
  return QI_OK;
}

template <>
int launch_zoom<float>(const ZoomArgs<float>& a, int64_t n_channels, hipStream_t st) {
  int64_t groups = 0;
  int chunks = 0;
  if (int rc = zoom_grid(a, &groups, &chunks)) return rc;
  if (chunks <= 0) return QI_OK;
  dim3 grid((unsigned)groups, (unsigned)chunks, (unsigned)n_channels);
  return a.stx ? launch_zoom_v<float, false>(a, grid, st) : launch_zoom_v<float, true>(a, grid, st);
}

template <>
int launch_zoom2<float>(const ZoomArgs<float>& a0, const ZoomArgs<float>& a2, int64_t n_channels, hipStream_t st) {
  int64_t g0 = 0, g2 = 0;
  int r0 = 0, r2 = 0;
  if (int rc = zoom_grid(a0, &g0, &r0)) return rc;
  if (int rc = zoom_grid(a2, &g2, &r2)) return rc;
  const bool coef = a0.coef != nullptr, bits = a0.bits != nullptr;
  if (a0.stx || !a2.stx || r0 <= 0 || r2 <= 0 || coef != (a2.coef != nullptr) || bits != (a2.bits != nullptr)) {
    set_error("zoom engine: the joint launch takes a styx table and a Stockwell table with the same panels");
    return QI_ERR_STATE;
  }
  dim3 grid((unsigned)(g0 > g2 ? g0 : g2), (unsigned)(r0 + r2), (unsigned)n_channels);
  const bool sdesc = n_channels >= 4;  // (see zoom_level)
#define QI_ZOOM2(C, B)                                                                  \
  do {                                                                                  \
    if (sdesc) k_zoom2<float, C, B, true><<<grid, kZoomThreads, 0, st>>>(a0, a2, r0);   \
    else k_zoom2<float, C, B, false><<<grid, kZoomThreads, 0, st>>>(a0, a2, r0);        \
  } while (0)
  if (coef && bits) QI_ZOOM2(true, true);
  else if (coef) QI_ZOOM2(true, false);
  else if (bits) QI_ZOOM2(false, true);
  else QI_ZOOM2(false, false);
#undef QI_ZOOM2
  QI_LAUNCH_CHECK();
  return QI_OK;
}

// Interpolation weights of lane L for window sample j of a wave-step of class `cls`: the lane sits pos = L / D coarse
// samples after the window's reference sample (index N / 2 - 1), D = 64 >> grid; with q = floor(pos), x = pos - q the
// N taps are the coarse samples q - N/2 + 1 .. q + N/2 (window samples j = q .. q + N - 1; zero elsewhere).  The taps
// are the interpolator that is EXACT for the N / 2 tones +-omega_k at the Chebyshev nodes of the band [-pi / r, pi / r]
// (r = the class's design oversampling): sum_j w_j exp(i omega_k j) = exp(i omega_k x) -- N real equations for N real
// weights, solved in long double.  Worst-case error of a unit tone anywhere in the band, float32 weights included:
// 9e-8 (N = 10, r = 4), 6e-7 (N = 6, r = 8), 3e-7 (N = 4, r = 32); the 12-tap Kaiser-windowed sinc it replaces: 6e-7.
// Layout [tap][lane].
// N taps (nodes -N/2 + 1 .. N/2) of the interpolator to the fraction x in [0, 1) that is exact at the Chebyshev nodes of
// the band [-band, band] (radians per coarse sample); out[c] belongs to node c - N/2 + 1
static void interp_taps(int N, long double band, long double x, long double* out) {
  const int half = N / 2;
  long double M[10][11];
  for (int k = 0; k < half; ++k) {
    const long double om = band * std::cos((long double)(2 * k + 1) * 3.14159265358979323846264338327950288L / (long double)(2 * N));
    for (int c = 0; c < N; ++c) {
      const long double node = (long double)(c - half + 1);
      M[k][c] = std::cos(om * node);
      M[half + k][c] = std::sin(om * node);
    }
    M[k][N] = std::cos(om * x);
    M[half + k][N] = std::sin(om * x);
  }
  for (int c = 0; c < N; ++c) {  // Gauss-Jordan with partial pivoting
    int piv = c;
    for (int r = c + 1; r < N; ++r)
      if (std::fabs((double)M[r][c]) > std::fabs((double)M[piv][c])) piv = r;
    if (piv != c)
      for (int k = 0; k <= N; ++k) std::swap(M[c][k], M[piv][k]);
    const long double d = M[c][c];
    for (int k = 0; k <= N; ++k) M[c][k] /= d;
    for (int r = 0; r < N; ++r) {
      if (r == c) continue;
      const long double f = M[r][c];
      if (f != 0.0L)
        for (int k = 0; k <= N; ++k) M[r][k] -= f * M[c][k];
    }
  }
  for (int c = 0; c < N; ++c) out[c] = M[c][N];
}

void zoom_weights(int cls, int lane_off, float* w) {
  const int taps = zoom_taps(cls), N = zoom_ntap(cls), half = N / 2;
  const long double D = (long double)(kZoomD >> zoom_grid(cls));
  const long double band = 3.14159265358979323846264338327950288L / (long double)zoom_design_oversampling(cls);
  for (int lane = 0; lane < kWave; ++lane) {
    const long double pos = (long double)(lane - lane_off) / D;
    const int q = (int)std::floor((double)pos);
    const long double x = pos - (long double)q;
    long double t[10];
    interp_taps(N, band, x, t);
    for (int j = 0; j < taps; ++j) w[j * kWave + lane] = 0.0f;
    for (int c = 0; c < N; ++c) {
      const int j = (half - 1) + q + (c - half + 1);
      if (j >= 0 && j < taps) w[j * kWave + lane] = (float)t[c];
    }
  }
}

}  // namespace native
}  // namespace qi
