// float64 zoom: the narrow-spectrum bands of a float64 panel at the decimated rate.
//
// A band whose spectrum occupies `len` of the Lf bins around bin k_c is a slow envelope on a carrier:
//   full[tau] = sum_k Y[k] W^(k tau) = W^(k_c tau) env(tau),   env(tau) = sum_i Y[k_lo + i] W^((i - len / 2) tau),
// W = exp(2 pi i / Lf), Y = record spectrum x band spectrum (the same operand the one-pass loader of the two-pass engine
// forms).  On the coarse grid tau = m D (D = Lf / M fine samples per coarse sample, M >= 4 len: at least 4 times
// oversampled) the envelope is ONE M-point inverse transform of the len occupied bins moved to baseband:
//   k_z64_gather  writes them (zero elsewhere) for every band of a level, hipFFT (Z2Z, batched) transforms them,
//   k_z64_interp  produces the panel rows: 16-tap band-optimal interpolation (exact at the 16 Chebyshev nodes of the band
//                 [-pi / 4, pi / 4]; worst-case error of a unit tone anywhere in the band 2.8e-12, double weights) x
//                 carrier phasor (exact-phase seed per band and thread, advanced by an exact-phase step), crop / roll of
//                 the transform kind as an offset of tau, and the tfr_info reductions from registers in the partial
//                 layout of the two-pass kernels (so that their tail launch finishes both).
// Against the exact two-pass kernels this trades a 2^20 / 2^21-point transform per band for M <= Lf / 4 points plus 32
// fused multiply-adds per output.  float32 has its own zoom engine (qi_zoom.hip: scalar-register windows, 4 - 10 taps).
#include "qi_common.hpp"
#include "qi_device.hpp"
#include "qi_native.hpp"
#include "qi_fft_reg.hpp"

namespace qi {
namespace native {

namespace {

using cd = cplx<double>;
constexpr int kZ64Threads = 256;

template <bool STX>
__global__ void __launch_bounds__(256) k_z64_gather(Z64Args a) {
  const BandDesc bd = a.bands[blockIdx.y];
  const int64_t ch = blockIdx.z;
  const uint32_t q = blockIdx.x * 256u + threadIdx.x;  // baseband bin of the coarse spectrum
  if (q >= (uint32_t)a.M) return;
  const uint32_t i = (q + (uint32_t)(bd.k_len / 2)) & ((uint32_t)a.M - 1u);  // bin k_lo + i of the band's support
  cd y = mk<double>(0.0, 0.0);
  if (i < (uint32_t)bd.k_len) {
    const int32_t k = bd.k_lo + (int32_t)i;
    const cd* __restrict__ X = a.X + ch * a.Lf;
    const uint32_t mask = (uint32_t)a.Lf - 1u;
    if (STX) {
      const cd x = X[(uint32_t)(k + (int32_t)bd.shift) & mask];
      const double e = bd.coef * (double)k;
      const double w = exp2_t(-e * e) * a.inv_len;
      y = mk<double>(x.x * w, x.y * w);
    } else {
      y = cmul(X[(uint32_t)k & mask], a.Hc[bd.src_off + i]);
    }
  }
  a.Z[((int64_t)ch * a.nbands + blockIdx.y) * a.M + q] = y;
}

// KIND: 0 zero-padded linear correlation (Lf = 2 n, panel sample t = full-length sample t + n / 2 - 1), 1 circular
// correlation rolled by n / 2, 2 Stockwell (no carrier: its bands are centred on bin 0 after the shift).
// One workgroup = one tile of kZ64Tile consecutive panel samples (one partial slot per band and tile) of the bands
// blockIdx.y, blockIdx.y + gridDim.y, ... of the level; a thread owns samples tid + 256 r: consecutive lanes, consecutive
// samples (every store a contiguous run), and -- 256 being a multiple of D -- one interpolation phase for all of them,
// so its 16 weights stay in registers.  The coarse samples of the tile sit in LDS (lanes of one coarse interval read the
// same address: broadcast).
// LOG2D (the grid's coarse step) is a compile-time constant: consecutive samples of a thread then read overlapping
// windows at known offsets (on the coarsest grid 12 of the 16 taps are the previous sample's), and the loads are shared.
template <int KIND, int LOG2D>
__global__ void __launch_bounds__(kZ64Threads) k_z64_interp(Z64Args a) {
  constexpr int NW = kZ64Threads / kWave, N = kZ64Taps;
  constexpr int R = kZ64Tile / kZ64Threads;  // samples per thread and band
  __shared__ cd win[((R * kZ64Threads) >> LOG2D) + N + 1];
  __shared__ double s_red[NW];
  __shared__ double s_fin[3][NW];
  __shared__ double ltab[128][2];  // log2 table of the entropy sums (see log2_pos)
  const int tid = threadIdx.x, lane = tid & (kWave - 1), wv = tid / kWave;
  if (tid < 128) {
    ltab[tid][0] = kLog2Tab[tid][0];
    ltab[tid][1] = kLog2Tab[tid][1];
  }
  const int64_t ch = blockIdx.z, tile = blockIdx.x;
  const int64_t n = a.n;
  constexpr uint32_t TT = (uint32_t)R * kZ64Threads;
  const uint32_t off = KIND == 0 ? (uint32_t)(n / 2 - 1) : (KIND == 1 ? (uint32_t)(n / 2) : 0u);
  const uint32_t lmask = (uint32_t)a.Lf - 1u, mmask = (uint32_t)a.M - 1u;
  constexpr uint32_t D = 1u << LOG2D;
  const uint32_t t_first = (uint32_t)tile * TT + (uint32_t)tid;   // this thread's first panel sample
  const uint32_t tau_tile = ((uint32_t)tile * TT + off) & lmask;  // full-length sample of the tile's first one
  const uint32_t tau_first = (t_first + off) & lmask;
  const uint32_t m_first = (tau_tile >> LOG2D) - (uint32_t)(N / 2 - 1);  // first coarse sample of the window (mod M)
  const int nwin = (int)(TT >> LOG2D) + N + 1;
  const uint32_t phase = tau_first & (D - 1u);
  // window index of this thread's first tap at r = 0; it advances by 256 / D per r
  const uint32_t idx0 = (tau_first >> LOG2D) - (tau_tile >> LOG2D);
  constexpr uint32_t istep = kZ64Threads >> LOG2D;
  double w[N];
#pragma unroll
  for (int j = 0; j < N; ++j) w[j] = a.weights[phase * N + j];
  // per-time power sums of the tile over this workgroup's bands: in LDS (32 KB), not in 2 R registers per thread -- with
  // the sample loop unrolled by two the kernel then fits three waves per SIMD (1.22 against 1.55 ms per launch of 80
  // bands x 4 records with the sums in registers and two waves)
  __shared__ double colv[R * kZ64Threads];
  double* __restrict__ col0 = colv + tid;
#pragma unroll
  for (int r = 0; r < R; ++r) col0[r * kZ64Threads] = 0.0;
  double mx = 0.0, plogp = 0.0;
  for (int jj = blockIdx.y; jj < a.nbands; jj += gridDim.y) {
    const BandDesc bd = a.bands[jj];
    const cd* __restrict__ C = a.Z + ((int64_t)ch * a.nbands + jj) * a.M;
    __syncthreads();  // the previous band's readers are done with the window (and with s_red)
    for (int i = tid; i < nwin; i += kZ64Threads) win[i] = C[(m_first + (uint32_t)i) & mmask];
    // carrier: exp(2 pi i k_c tau / Lf) at this thread's first sample, advanced by exp(2 pi i k_c 256 / Lf); the
    // phases are exact integers modulo Lf
    cd ph = mk<double>(1.0, 0.0), st = ph;
    if (KIND != 2) {
      const uint32_t kc = (uint32_t)(bd.k_lo + bd.k_len / 2);
      double c, s;
      unit_root_t<double>((kc * tau_first) & lmask, a.two_over_len, &c, &s);
      ph = mk<double>(c, s);
      unit_root_t<double>((kc * (uint32_t)kZ64Threads) & lmask, a.two_over_len, &c, &s);
      st = mk<double>(c, s);
    }
    const int64_t orow = ((int64_t)ch * a.panel_bands + bd.out_band) * n;
    // split band (bd.add_row): this is only the tapered part of its atom -- the samples go to row add_row - 1 of
    // split_part and count for nothing here; the edge items of the block launch add their part and finish the band
    const bool part = bd.add_row != 0;
    cd* __restrict__ coef_row =
        part ? a.split_part + ((int64_t)ch * a.split_rows + (bd.add_row - 1)) * n : (a.coef ? a.coef + orow : nullptr);
    double* __restrict__ bits_row = a.bits && !part ? a.bits + orow : nullptr;
    const double pscale = part ? 0.0 : a.power_scale;
    double rowacc = 0.0, pl = 0.0;
    __syncthreads();
#pragma unroll 2
    for (int r = 0; r < R; ++r) {
      const cd* __restrict__ s = win + idx0 + (uint32_t)r * istep;
      double zr[2] = {0.0, 0.0}, zi[2] = {0.0, 0.0};  // two accumulation chains per part (the taps are independent)
#pragma unroll
      for (int j = 0; j < N; ++j) {
        const cd x = s[j];
        zr[j & 1] = fma(w[j], x.x, zr[j & 1]);
        zi[j & 1] = fma(w[j], x.y, zi[j & 1]);
      }
      cd z = mk<double>(zr[0] + zr[1], zi[0] + zi[1]);
      if (KIND != 2) {
        z = cmul_rn(z, ph);
        ph = cmul_rn(ph, st);
      }
      const uint32_t tt = t_first + (uint32_t)r * kZ64Threads;
      if (coef_row) stream_store(coef_row + tt, z);
      const double m2 = norm2(z.x, z.y);
      if (bits_row) bits_row[tt] = log2_t(sqrt_t(m2) + a.eps);
      const double p = mul_rn(pscale, m2);
      col0[r * kZ64Threads] += p;
      rowacc += p;
      mx = max_t(mx, p);
      pl += plog2p(p, ltab);
    }
    plogp += pl;
    if (a.part_band && !part) {  // (the same for every thread)
      const double rs = wave_sum(rowacc);
      if (lane == 0) s_red[wv] = rs;
      __syncthreads();
      if (tid == 0) {
        double t = 0.0;
        for (int q = 0; q < NW; ++q) t += s_red[q];
        a.part_band[((int64_t)ch * a.panel_bands + bd.out_band) * a.pb_stride + tile] = t;
      }
    }
  }
  double tot = 0.0;
  double* __restrict__ time_row = a.time_part ? a.time_part + ((int64_t)ch * a.chunk_total + a.chunk_base + blockIdx.y) * n : nullptr;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const double v = col0[r * kZ64Threads];
    tot += v;
    if (time_row) time_row[t_first + (uint32_t)r * kZ64Threads] = v;
  }
  if (a.part_stat) {
    const double r0 = wave_max(mx), r1 = wave_sum(tot), r2 = wave_sum(plogp);
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
      double* o = a.part_stat + ((int64_t)ch * a.stat_stride + (int64_t)(a.chunk_base + blockIdx.y) * a.stat_nblk + tile) * 3;
      o[0] = m;
      o[1] = s1;
      o[2] = s2;
    }
  }
}

template <int KIND>
int launch_interp_v(const Z64Args& a, dim3 grid, hipStream_t st) {
  switch (a.log2d) {
    case 6: k_z64_interp<KIND, 6><<<grid, kZ64Threads, 0, st>>>(a); break;
    case 5: k_z64_interp<KIND, 5><<<grid, kZ64Threads, 0, st>>>(a); break;
    case 4: k_z64_interp<KIND, 4><<<grid, kZ64Threads, 0, st>>>(a); break;
    case 3: k_z64_interp<KIND, 3><<<grid, kZ64Threads, 0, st>>>(a); break;
    default: k_z64_interp<KIND, 2><<<grid, kZ64Threads, 0, st>>>(a); break;
  }
  QI_LAUNCH_CHECK();
  return QI_OK;
}

}  // namespace

int launch_z64_gather(const Z64Args& a, int64_t n_channels, hipStream_t st) {
  if (a.nbands <= 0) return QI_OK;
  dim3 grid((unsigned)((a.M + 255) / 256), (unsigned)a.nbands, (unsigned)n_channels);
  if (a.kind == 2) k_z64_gather<true><<<grid, 256, 0, st>>>(a);
  else k_z64_gather<false><<<grid, 256, 0, st>>>(a);
  QI_LAUNCH_CHECK();
  return QI_OK;
}

int launch_z64_interp(const Z64Args& a, int nchunk, int64_t n_channels, hipStream_t st) {
  if (a.nbands <= 0) return QI_OK;
  const int64_t TT = a.n / a.nblk;
  if (TT != kZ64Tile || TT * a.nblk != a.n || a.log2d < 2 || a.log2d > 6 || (a.M & (a.M - 1)) != 0) {
    set_error("float64 zoom: tile of %lld samples / coarse step %d not supported", (long long)TT, 1 << a.log2d);
    return QI_ERR_UNSUPPORTED;
  }
  dim3 grid((unsigned)a.nblk, (unsigned)nchunk, (unsigned)n_channels);
  return a.kind == 0 ? launch_interp_v<0>(a, grid, st) : (a.kind == 1 ? launch_interp_v<1>(a, grid, st) : launch_interp_v<2>(a, grid, st));
}

// weights[phase][tap] of coarse step D = 1 << log2d: the 16-tap interpolator at x = phase / D that is exact for the
// tones at the Chebyshev nodes of [-pi / 4, pi / 4] (the same design as zoom_weights, solved in long double)
void z64_weights(int log2d, double* w) {
  constexpr int N = kZ64Taps, half = N / 2;
  const int D = 1 << log2d;
  const long double pi = 3.14159265358979323846264338327950288L;
  const long double band = pi / 4.0L;
  for (int ph = 0; ph < D; ++ph) {
    const long double x = (long double)ph / (long double)D;
    long double M[N][N + 1];
    for (int k = 0; k < half; ++k) {
      const long double om = band * std::cos((long double)(2 * k + 1) * pi / (long double)(2 * N));
      for (int c = 0; c < N; ++c) {
        const long double node = (long double)(c - half + 1);
        M[k][c] = std::cos(om * node);
        M[half + k][c] = std::sin(om * node);
      }
      M[k][N] = std::cos(om * x);
      M[half + k][N] = std::sin(om * x);
    }
    for (int i = 0; i < N; ++i) {  // Gauss-Jordan with partial pivoting
      int piv = i;
      for (int r = i + 1; r < N; ++r)
        if (std::fabs((double)M[r][i]) > std::fabs((double)M[piv][i])) piv = r;
      if (piv != i)
        for (int c = 0; c <= N; ++c) std::swap(M[i][c], M[piv][c]);
      const long double d = M[i][i];
      for (int c = 0; c <= N; ++c) M[i][c] /= d;
      for (int r = 0; r < N; ++r) {
        if (r == i) continue;
        const long double f = M[r][i];
        if (f == 0.0L) continue;
        for (int c = 0; c <= N; ++c) M[r][c] -= f * M[i][c];
      }
    }
    for (int c = 0; c < N; ++c) w[ph * N + c] = (double)M[c][N];
  }
}

}  // namespace native
}  // namespace qi
