// Zoom engine: the narrow-spectrum bands of a panel (long atoms: a few hundred to a few thousand occupied bins out of
// a million) are slowly varying envelopes on a carrier.  Their occupied bins, moved to baseband, are transformed on a
// COARSE time grid of M = Lf / 64 samples (one small batched inverse FFT per band), and the panel is produced from
// that by band-limited interpolation -- a 12-tap Kaiser-windowed sinc on the >= 4x oversampled coarse grid, error
// below 6e-7 of a unit tone at the band edge (where the spectrum is < 2^-30 of its peak), < 1e-9 in the bulk -- times
// the carrier phasor.  About 50 instructions per output instead of a share of a million-point transform, nothing
// discarded (the zero-padded half of the linear correlation is simply not evaluated), no intermediate: the kernel
// is bound by the panel write.
//
// One wave = 64 lanes = the 64 fine positions between two coarse samples: lane L produces sample 64 tau + L for
// kSteps consecutive tau; the coarse samples a step needs are uniform over the wave (scalar registers, taken from a
// vector register by v_readlane), the 13 interpolation weights are per-lane constants, every store is a 512-byte
// run.  Waves only meet for the per-band power sum (one barrier per band).  No MFMA: there is no dense contraction here.
#include "qi_common.hpp"
#include "qi_device.hpp"
#include "qi_native.hpp"
#include "qi_fft_reg.hpp"

namespace qi {
namespace native {

namespace {

constexpr int kZoomThreads = 256;

__device__ __forceinline__ float lane_value(float v, int lane) {
  return __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v), lane));
}

// Input of the coarse stage (k_zoom_coarse, qi_block.hip): with the coarse grid of M = P * 4096 samples and
// tau = P tau2 + tau1, the envelope b[tau] = sum_kappa Yc[kappa] exp(2 pi i kappa tau / M) is, for each tau1 < P, one
// 4096-point transform of  in[tau1][kappa0] = exp(2 pi i kappa0 tau1 / M) sum_r Yc[kappa0 + 4096 r] exp(2 pi i r tau1 / P).
// Yc = the band's occupied bins moved to baseband (Y as the one-pass loader of qi_native.hip forms it: spectrum x
// compact bank, or shifted spectrum x Gaussian).  One thread per (kappa0, tau1), every block r that can hold occupied
// bins visited; written to coarse[c][j][tau1][kappa0], transformed in place by the coarse stage.
template <typename T, bool STX>
__global__ void __launch_bounds__(256) k_zoom_gather(ZoomArgs<T> a) {
  const int32_t M = (int32_t)a.M, P = M / kBlk;
  const int32_t e = (int32_t)(blockIdx.x * 256 + threadIdx.x);  // = tau1 * 4096 + kappa0
  if (e >= M) return;
  const int32_t kappa0 = e & (kBlk - 1);
  const uint32_t tau1 = (uint32_t)(e / kBlk);
  const int j = blockIdx.y;
  const int64_t ch = blockIdx.z;
  const BandDesc bd = a.bands[j];
  const int32_t kc = STX ? 0 : bd.k_lo + bd.k_len / 2;
  const int32_t ks_lo = bd.k_lo - kc, ks_hi = ks_lo + bd.k_len;  // support in baseband bins
  const uint32_t lmask = (uint32_t)a.Lf - 1u;
  const cplx<T>* __restrict__ X = a.X + ch * a.Lf;
  cplx<T> acc = mk<T>(T(0), T(0));
  for (int32_t r = 0; r < P; ++r) {
    const int32_t kappa = r * kBlk + kappa0;
    const int32_t ks = kappa < M / 2 ? kappa : kappa - M;
    if (ks < ks_lo || ks >= ks_hi) continue;
    const int32_t k = kc + ks;
    cplx<T> y;
    if (STX) {
      const cplx<T> x = X[(uint32_t)(k + (int32_t)bd.shift) & lmask];
      const T g0 = (T)bd.coef * (T)k;
      const T g = exp2_t(-g0 * g0) * a.inv_len;
      y = mk<T>(x.x * g, x.y * g);
    } else {
      y = cmul(X[(uint32_t)k & lmask], a.Hc[bd.src_off + (k - bd.k_lo)]);  // k < 0: bins modulo Lf
    }
    float sr, cr;
    sincospif(2.0f * (float)(((uint32_t)r * tau1) & (uint32_t)(P - 1)) / (float)P, &sr, &cr);
    const cplx<T> t = cmul(y, mk<T>((T)cr, (T)sr));
    acc.x += t.x;
    acc.y += t.y;
  }
  float s, c;
  sincospif(2.0f * (float)(((uint32_t)kappa0 * tau1) & ((uint32_t)M - 1u)) / (float)M, &s, &c);
  a.coarse[((int64_t)ch * a.nbands + j) * a.M + e] = cmul(acc, mk<T>((T)c, (T)s));
}

// Fine stage.  PHASOR: multiply by the carrier exp(2 pi i k_c f / Lf) (Gabor banks; the Stockwell bands are at
// baseband already).  Output sample t is the full-length sample f = t + off, off = 64 A - e (e = 0 or 1).
template <typename T, int STEPS, int TAPS, bool PHASOR, bool COEF, bool BITS>
__global__ void __launch_bounds__(kZoomThreads) k_zoom(ZoomArgs<T> a) {
  constexpr int NW = kZoomThreads / kWave, HALF = (TAPS - 1) / 2;
  static_assert(STEPS + TAPS - 1 <= kWave, "the window of one wave must fit its lanes");
  const int tid = threadIdx.x, lane = tid & (kWave - 1), wv = tid / kWave;
  const int64_t ch = blockIdx.z;
  const int64_t gw = (int64_t)blockIdx.x * NW + wv;  // wave index along time
  const uint32_t tau_a = (uint32_t)gw * STEPS;       // first coarse step of this wave
  const uint32_t mmask = (uint32_t)a.M - 1u, lmask = (uint32_t)a.Lf - 1u;
  float wgt[TAPS];
#pragma unroll
  for (int j = 0; j < TAPS; ++j) wgt[j] = a.weights[j * kZoomD + lane];  // [tap][lane]: coalesced
  T colp[STEPS];
#pragma unroll
  for (int s = 0; s < STEPS; ++s) colp[s] = T(0);
  T mx = T(0);
  double plogp = 0.0;
  const uint32_t t_base = tau_a * kZoomD + (uint32_t)lane;  // output sample of step s: t_base + 64 s
  __shared__ double s_red[2][NW];
  int par = 0;  // double-buffered so that one barrier per band is enough

  // lane i holds coarse sample tau_a + A - HALF + i of the band: step s interpolates from lanes s .. s + TAPS - 1.
  // The samples (and the descriptor) of the next band are requested a band ahead: two registers hide the one
  // memory latency of the band loop.
  const uint32_t wtau = (tau_a + (uint32_t)a.tau_off - (uint32_t)HALF + (uint32_t)lane) & mmask;
  // coarse sample tau = P tau2 + tau1 sits at plane tau1, position tau2 (the layout the coarse stage writes)
  const uint32_t widx = (wtau & ((1u << a.coarse_planes_log2) - 1u)) * (uint32_t)kBlk + (wtau >> a.coarse_planes_log2);
  const int jj0 = a.band_first + blockIdx.y, jj_end = a.band_first + a.band_count;
  cplx<T> smp_next = mk<T>(T(0), T(0));
  BandDesc bd_next = a.bands[jj0 < jj_end ? jj0 : a.band_first];
  if (jj0 < jj_end) smp_next = a.coarse[((int64_t)ch * a.nbands + jj0) * a.M + widx];
  for (int jj = jj0; jj < jj_end; jj += gridDim.y) {
    const BandDesc bd = bd_next;
    const cplx<T> smp = smp_next;
    if (jj + (int)gridDim.y < jj_end) {
      bd_next = a.bands[jj + gridDim.y];
      smp_next = a.coarse[((int64_t)ch * a.nbands + jj + gridDim.y) * a.M + widx];
    }
    cplx<T> P = mk<T>(T(1), T(0)), Q = mk<T>(T(1), T(0));
    if (PHASOR) {
      // carrier: exp(2 pi i kc f / Lf), f = 64 (tau + A) + lane - e: per-lane factor P, per-step factor Q (lane i
      // holds the factor of step i); the phases are exact integers modulo Lf
      const uint32_t kc = (uint32_t)(bd.k_lo + bd.k_len / 2);
      double c, s;
      unit_root((kc * (uint32_t)(lane - a.lane_off)) & lmask, a.two_over_len, &c, &s);
      P = mk<T>((T)c, (T)s);
      unit_root((kc * kZoomD * (tau_a + (uint32_t)a.tau_off + (uint32_t)lane)) & lmask, a.two_over_len, &c, &s);
      Q = mk<T>((T)c, (T)s);
    }
    const int64_t orow = ((int64_t)ch * a.panel_bands + bd.out_band) * a.n;
    char* __restrict__ coef_row = reinterpret_cast<char*>(a.coef ? a.coef + orow : nullptr);
    char* __restrict__ bits_row = reinterpret_cast<char*>(a.bits ? a.bits + orow : nullptr);
    uint32_t tb = t_base;
    asm volatile("" : "+v"(tb));  // keep the band-invariant addresses out of the loop-invariant hoisting
    T rowacc = T(0), pl = T(0);
    // interpolation: real parts of all steps, then imaginary parts (the wave-uniform coarse samples sit in scalar
    // registers, taken from `smp` by v_readlane as the window slides; one part at a time keeps them within budget)
    T zr[STEPS], zi[STEPS];
    {
      float sx[STEPS + TAPS - 1];
#pragma unroll
      for (int i = 0; i < TAPS - 1; ++i) sx[i] = lane_value(smp.x, i);
#pragma unroll
      for (int s = 0; s < STEPS; ++s) {
        sx[s + TAPS - 1] = lane_value(smp.x, s + TAPS - 1);
        T acc = T(0);
#pragma unroll
        for (int j = 0; j < TAPS; ++j) acc = fmaf(wgt[j], sx[s + j], acc);
        zr[s] = acc;
      }
    }
    {
      float sy[STEPS + TAPS - 1];
#pragma unroll
      for (int i = 0; i < TAPS - 1; ++i) sy[i] = lane_value(smp.y, i);
#pragma unroll
      for (int s = 0; s < STEPS; ++s) {
        sy[s + TAPS - 1] = lane_value(smp.y, s + TAPS - 1);
        T acc = T(0);
#pragma unroll
        for (int j = 0; j < TAPS; ++j) acc = fmaf(wgt[j], sy[s + j], acc);
        zi[s] = acc;
      }
    }
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
      const T br = zr[s], bi = zi[s];
      cplx<T> z = mk<T>(br, bi);
      if (PHASOR) {
        const cplx<T> q = mk<T>(lane_value(Q.x, s), lane_value(Q.y, s));
        z = cmul_rn(z, cmul_rn(P, q));
      }
      const uint32_t tt = tb + (uint32_t)(kZoomD * s);
      if (COEF) *reinterpret_cast<cplx<T>*>(coef_row + (size_t)(tt * (uint32_t)sizeof(cplx<T>))) = z;
      const T m2 = norm2(z.x, z.y);
      if (BITS) *reinterpret_cast<T*>(bits_row + (size_t)(tt * (uint32_t)sizeof(T))) = log2_t(sqrt_t(m2) + a.eps);
      const T p = mul_rn(a.power_scale, m2);
      colp[s] += p;
      rowacc += p;
      mx = p > mx ? p : mx;
      pl += plog2p(p);
    }
    plogp += (double)pl;
    if (a.part_band) {
      // one partial per workgroup and band: wave sums through LDS, written by thread 0 a band later
      const double r = wave_sum((double)rowacc);
      if (lane == 0) s_red[par][wv] = r;
      __syncthreads();
      if (tid == 0) {
        double t = 0.0;
        for (int q = 0; q < NW; ++q) t += s_red[par][q];
        a.part_band[((int64_t)ch * a.panel_bands + bd.out_band) * a.nblk + blockIdx.x] = t;
      }
      par ^= 1;
    }
  }

  T tot = T(0);
  char* __restrict__ time_row = reinterpret_cast<char*>(
      a.time_part ? a.time_part + ((int64_t)ch * a.chunk_total + a.chunk_base + blockIdx.y) * a.n : nullptr);
#pragma unroll
  for (int s = 0; s < STEPS; ++s) {
    tot += colp[s];
    const uint32_t tt = t_base + (uint32_t)(kZoomD * s);
    if (time_row) {
      T* dst = reinterpret_cast<T*>(time_row + (size_t)(tt * (uint32_t)sizeof(T)));
      *dst = a.time_accumulate ? *dst + colp[s] : colp[s];
    }
  }
  if (a.part_stat) {
    __shared__ double s_fin[3][NW];
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
      double* o = a.part_stat + ((int64_t)ch * a.stat_stride + a.stat_base + (int64_t)blockIdx.y * gridDim.x + blockIdx.x) * 3;
      o[0] = m;
      o[1] = s1;
      o[2] = s2;
    }
  }
}

template <typename T, int TAPS, bool PHASOR>
int launch_zoom_v(const ZoomArgs<T>& a, dim3 grid, hipStream_t st) {
  const bool coef = a.coef != nullptr, bits = a.bits != nullptr;
  if (coef && bits) k_zoom<T, kZoomSteps, TAPS, PHASOR, true, true><<<grid, kZoomThreads, 0, st>>>(a);
  else if (coef) k_zoom<T, kZoomSteps, TAPS, PHASOR, true, false><<<grid, kZoomThreads, 0, st>>>(a);
  else if (bits) k_zoom<T, kZoomSteps, TAPS, PHASOR, false, true><<<grid, kZoomThreads, 0, st>>>(a);
  else k_zoom<T, kZoomSteps, TAPS, PHASOR, false, false><<<grid, kZoomThreads, 0, st>>>(a);
  QI_LAUNCH_CHECK();
  return QI_OK;
}
template <typename T, int TAPS>
int launch_zoom_t(const ZoomArgs<T>& a, dim3 grid, hipStream_t st) {
  return a.stx ? launch_zoom_v<T, TAPS, false>(a, grid, st) : launch_zoom_v<T, TAPS, true>(a, grid, st);
}

}  // namespace

int64_t zoom_groups(int64_t n) { return n / ((int64_t)kZoomD * kZoomSteps * (kZoomThreads / kWave)); }

template <>
int launch_zoom_gather<float>(const ZoomArgs<float>& a, int64_t n_channels, hipStream_t st) {
  if (a.nbands <= 0) return QI_OK;
  dim3 grid((unsigned)ceil_div(a.M, 256), (unsigned)a.nbands, (unsigned)n_channels);
  if (a.stx) k_zoom_gather<float, true><<<grid, 256, 0, st>>>(a);
  else k_zoom_gather<float, false><<<grid, 256, 0, st>>>(a);
  QI_LAUNCH_CHECK();
  return QI_OK;
}

template <>
int launch_zoom<float>(const ZoomArgs<float>& a, int cls, int nchunk, int64_t n_channels, hipStream_t st) {
  if (a.band_count <= 0) return QI_OK;
  const int64_t groups = zoom_groups(a.n);
  if (groups * kZoomD * kZoomSteps * (kZoomThreads / kWave) != a.n) {
    set_error("zoom engine: record length %lld is not a multiple of %d samples", (long long)a.n,
              kZoomD * kZoomSteps * (kZoomThreads / kWave));
    return QI_ERR_UNSUPPORTED;
  }
  dim3 grid((unsigned)groups, (unsigned)nchunk, (unsigned)n_channels);
  switch (cls) {
    case 0: return launch_zoom_t<float, zoom_taps(0)>(a, grid, st);
    case 1: return launch_zoom_t<float, zoom_taps(1)>(a, grid, st);
    default: return launch_zoom_t<float, zoom_taps(2)>(a, grid, st);
  }
}

// Interpolation weights of lane L for tap j: h((j - half) - (L - e) / 64), h = Kaiser-windowed sinc (beta = 14) of
// 2 half taps: half = 6, 9, 18 for the classes 0, 1, 2 (errors 6e-7, 4e-7, 3e-7 of a unit tone at the band edge)
void zoom_weights(int cls, int lane_off, float* w) {
  const int taps = zoom_taps(cls);
  const double beta = 14.0, half = (double)((taps - 1) / 2);
  auto bessel_i0 = [](double x) {
    double sum = 1.0, term = 1.0;
    for (int k = 1; k < 64; ++k) {
      term *= (x / (2.0 * k)) * (x / (2.0 * k));
      sum += term;
      if (term < 1e-18 * sum) break;
    }
    return sum;
  };
  const double i0b = bessel_i0(beta);
  for (int lane = 0; lane < kZoomD; ++lane) {
    const double phi = (double)(lane - lane_off) / kZoomD;
    for (int j = 0; j < taps; ++j) {
      const double x = (double)j - half - phi;
      double v = 0.0;
      if (std::fabs(x) < half) {
        const double r = x / half;
        const double win = bessel_i0(beta * std::sqrt(1.0 - r * r)) / i0b;
        const double sinc = std::fabs(x) < 1e-12 ? 1.0 : std::sin(M_PI * x) / (M_PI * x);
        v = sinc * win;
      }
      w[j * kZoomD + lane] = (float)v;
    }
  }
}

}  // namespace native
}  // namespace qi
