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
      // (zero-padded kind: panel sample t is full-length sample t + n / 2 - 1.  The odd sample is taken out of the envelope
      // -- env'(tau) = env(tau - 1): a phase ramp exp(-2 pi i (i - len / 2) / Lf) on the band's baseband bins, folded into
      // its compact bank row at plan time (fill_native_bank) -- so that the fine stage reads an aligned window
      // (tau = t + n / 2); the carrier keeps its phase at tau - 1)
      y = cmul(X[(uint32_t)k & mask], a.Hc[bd.src_off + i]);
    }
  }
  a.Z[((int64_t)ch * a.nbands + blockIdx.y) * (a.M + 2 * kZ64Pad) + kZ64Pad + q] = y;
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
#ifndef QI_Z64_INTERP_WAVES
#define QI_Z64_INTERP_WAVES 1
#endif
#ifndef QI_Z64_INTERP_UNROLL
#define QI_Z64_INTERP_UNROLL 2
#endif
template <int KIND, int LOG2D>
__global__ void __launch_bounds__(kZ64Threads, QI_Z64_INTERP_WAVES) k_z64_interp(Z64Args a) {
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
  // (KIND 0: the odd sample of t + n / 2 - 1 sits in the envelope's phase ramp, see k_z64_gather; the carrier below takes it)
  const uint32_t off = KIND == 2 ? 0u : (uint32_t)(n / 2);
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
  double mx = 0.0, plogp = 0.0, ptot = 0.0;  // (ptot: the total power when no per-time sums are kept)
  const bool want_time = a.time_part != nullptr;
  for (int jj = blockIdx.y; jj < a.nbands; jj += gridDim.y) {
    const BandDesc bd = load_uniform(a.bands + jj);
    const cd* __restrict__ C = a.Z + ((int64_t)ch * a.nbands + jj) * (a.M + 2 * kZ64Pad) + kZ64Pad;
    __syncthreads();  // the previous band's readers are done with the window (and with s_red)
    for (int i = tid; i < nwin; i += kZ64Threads) win[i] = C[(m_first + (uint32_t)i) & mmask];
    // carrier: exp(2 pi i k_c tau / Lf) at this thread's first sample, advanced by exp(2 pi i k_c 256 / Lf); the
    // phases are exact integers modulo Lf
    const int64_t orow = ((int64_t)ch * a.panel_bands + bd.out_band) * n;
    // split band (bd.add_row): this is only the tapered part of its atom -- the samples go to row add_row - 1 of
    // split_part and count for nothing here; the edge items of the block launch add their part and finish the band
    const bool part = bd.add_row != 0;
    cd* __restrict__ coef_row =
        part ? a.split_part + ((int64_t)ch * a.split_rows + (bd.add_row - 1)) * n : (a.coef ? a.coef + orow : nullptr);
    // (no sample leaves the kernel -- reductions only --: the carrier, of modulus 1, changes no power; the envelope will do)
    const bool carrier = KIND != 2 && coef_row != nullptr;
    cd ph = mk<double>(1.0, 0.0), st = ph;
    if (carrier) {
      const uint32_t kc = (uint32_t)(bd.k_lo + bd.k_len / 2);
      double c, s;
      unit_root_t<double>((kc * (tau_first - (KIND == 0 ? 1u : 0u))) & lmask, a.two_over_len, &c, &s);
      ph = mk<double>(c, s);
      unit_root_t<double>((kc * (uint32_t)kZ64Threads) & lmask, a.two_over_len, &c, &s);
      st = mk<double>(c, s);
    }
    double* __restrict__ bits_row = a.bits && !part ? a.bits + orow : nullptr;
    const double pscale = part ? 0.0 : a.power_scale;
    double rowacc = 0.0, pl = 0.0;
    __syncthreads();
#pragma unroll QI_Z64_INTERP_UNROLL
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
      if (carrier) {
        z = cmul_rn(z, ph);
        ph = cmul_rn(ph, st);
      }
      const uint32_t tt = t_first + (uint32_t)r * kZ64Threads;
      if (coef_row) stream_store(coef_row + tt, z);
      const double m2 = norm2(z.x, z.y);
      if (bits_row) bits_row[tt] = log2_t(sqrt_t(m2) + a.eps);
      const double p = mul_rn(pscale, m2);
      if (want_time) col0[r * kZ64Threads] += p;
      rowacc += p;
      mx = max_t(mx, p);
      pl += plog2p_flat(p, ltab);
    }
    plogp += pl;
    ptot += rowacc;
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
  if (!want_time) tot = ptot;
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

// ---- interpolation on the matrix pipe (round 5): the grids of Lf / 16 and Lf / 8 samples ---------------------------------
// The 16-tap interpolation is a small dense contraction: for 16 consecutive coarse intervals i and the D phases ph of an
// interval, out[i][ph] = sum_j c[m0 - 7 + i + j] w[ph][j] = (Hankel matrix of the window, 16 x 16) x (weights^T, 16 x D).
// With D = 16 that is v_mfma_f64_16x16x4_f64 four times per part (real, imaginary), and the result layout of that
// instruction -- lane l, register r holds row (l >> 4) + 4 r, column l & 15 -- IS the panel order: sample
// t0 + 16 ((l >> 4) + 4 r) + (l & 15) = t0 + l + 64 r, consecutive lanes = consecutive samples, register r = wave-step r.  D = 8:
// the columns are (interval parity e, phase), the rows every second interval, the contraction runs over j' = j + e < 17
// (padded to 20: five instructions) with weights w[ph][j' - e].  Per 256 outputs and part: 4 (5) ds_read_b64 of a lane's
// Hankel element from the wave's window in LDS + 4 (5) MFMA, against 64 x (16 ds_read_b128 + 16 fused multiply-adds) per
// lane in k_z64_interp -- the vector pipe keeps the epilogue (carrier, powers, entropy logarithm, stores), which was 45 of
// that kernel's 82 instructions per output.  Same launch geometry, partial-sum layout and arithmetic per tap as
// k_z64_interp (one workgroup = one tile of kZ64Tile samples, a wave owns a quarter of it); the sums of products are formed
// in the order of the instruction (k = 0 .. 3 within a block of four taps, blocks in turn) -- float64 rounding, no more.
// No other kernel of the path is a contraction large enough for the matrix pipe.
typedef double qi_d4 __attribute__((ext_vector_type(4)));
#ifndef QI_Z64_MX_WAVES
#define QI_Z64_MX_WAVES 4
#endif
template <int KIND, int LOG2D, bool COEF>
__global__ void __launch_bounds__(kZ64Threads, QI_Z64_MX_WAVES) k_z64_mfma(Z64Args a) {
  static_assert(LOG2D == 4 || LOG2D == 3, "grids of Lf / 16 and Lf / 8 samples");
  constexpr int NW = kZ64Threads / kWave, N = kZ64Taps;
  constexpr int WSAMP = kZ64Tile / NW;           // panel samples per wave and band
  constexpr int STEPS = WSAMP / kWave;           // wave-steps (64 samples) per wave and band
  constexpr int GROUPS = STEPS / 4;              // 256-sample groups = one accumulator tile each
  constexpr int KB = LOG2D == 4 ? 4 : 5;         // blocks of four taps per contraction
  constexpr int ROWMUL = LOG2D == 4 ? 1 : 2;     // coarse samples per row of the tile
  constexpr int GSTEP = 256 >> LOG2D;            // coarse samples per group
  constexpr int NWIN = (WSAMP >> LOG2D) + 4 * KB;  // window of one wave (>= WSAMP / D + N - 1 + what the padded taps reach)
  constexpr int NPRE = (NWIN + kWave - 1) / kWave;
  static_assert(STEPS % 4 == 0 && GROUPS >= 1, "a wave's share is a whole number of 256-sample groups");
  __shared__ double win_re[NW][NWIN], win_im[NW][NWIN];
  __shared__ double s_red[NW];
  __shared__ double s_fin[3][NW];
  __shared__ double ltab[128][2];
  __shared__ double colv[kZ64Tile];
  const int tid = threadIdx.x, lane = tid & (kWave - 1);
  const int wv = __builtin_amdgcn_readfirstlane(tid / kWave);
  if (tid < 128) {
    ltab[tid][0] = kLog2Tab[tid][0];
    ltab[tid][1] = kLog2Tab[tid][1];
  }
  const int64_t ch = blockIdx.z, tile = blockIdx.x, n = a.n;
  const uint32_t off = KIND == 2 ? 0u : (uint32_t)(n / 2);
  const uint32_t lmask = (uint32_t)a.Lf - 1u, mmask = (uint32_t)a.M - 1u;
  const uint32_t t_wave = (uint32_t)tile * (uint32_t)kZ64Tile + (uint32_t)(wv * WSAMP);  // first panel sample of this wave
  const uint32_t tau_wave = (t_wave + off) & lmask;       // (a multiple of 256: every group starts on a coarse sample)
  const uint32_t m_wave = (tau_wave >> LOG2D) - (uint32_t)(N / 2 - 1);  // first window sample (mod M)
  // B operand: lane l supplies weights^T[k = 4 q + (l >> 4)][column l & 15]
  double wb[KB];
  {
    const int kq = lane >> 4, col = lane & 15;
#pragma unroll
    for (int q = 0; q < KB; ++q) {
      const int jp = 4 * q + kq;  // tap index j' of the (padded) contraction
      if (LOG2D == 4) {
        wb[q] = a.weights[col * N + jp];
      } else {
        const int e = col >> 3, ph = col & 7, j = jp - e;
        wb[q] = (j >= 0 && j < N) ? a.weights[ph * N + j] : 0.0;
      }
    }
  }
  // A operand: lane l reads window sample ROWMUL (l & 15) + (l >> 4) + 4 q of its group
  const int a_off = ROWMUL * (lane & 15) + (lane >> 4);
  double* __restrict__ col0 = colv + wv * WSAMP + lane;
  const bool want_time = a.time_part != nullptr;
#pragma unroll
  for (int s = 0; s < STEPS; ++s) col0[s * kWave] = 0.0;
  double mx = 0.0, plogp = 0.0, ptot = 0.0;
  const uint32_t tt0 = t_wave + (uint32_t)lane;
  // a wave's window comes through registers, requested ONE BAND AHEAD: the global loads of band j + 1 fly while band j is
  // interpolated (k_z64_interp waits for them at the top of every band)
  const int64_t zstride = a.M + 2 * kZ64Pad;
  cd pre[NPRE];
  auto request = [&](int band) {
    const cd* __restrict__ C = a.Z + ((int64_t)ch * a.nbands + band) * zstride + kZ64Pad;
#pragma unroll
    for (int q = 0; q < NPRE; ++q) {
      const int i = lane + q * kWave;
      pre[q] = C[(m_wave + (uint32_t)(i < NWIN ? i : NWIN - 1)) & mmask];
    }
  };
  if ((int)blockIdx.y < a.nbands) request(blockIdx.y);
  for (int jj = blockIdx.y; jj < a.nbands; jj += gridDim.y) {
    const BandDesc bd = load_uniform(a.bands + jj);
    __syncthreads();  // the previous band's readers are done with the windows (and with s_red)
#pragma unroll
    for (int q = 0; q < NPRE; ++q) {
      const int i = lane + q * kWave;
      if (i < NWIN) {
        win_re[wv][i] = pre[q].x;
        win_im[wv][i] = pre[q].y;
      }
    }
    if (jj + (int)gridDim.y < a.nbands) request(jj + gridDim.y);
    const int64_t orow = ((int64_t)ch * a.panel_bands + bd.out_band) * n;
    const bool part = bd.add_row != 0;  // split band: the tapered part only, to split_part, counts for nothing here
    cd* __restrict__ coef_row =
        part ? a.split_part + ((int64_t)ch * a.split_rows + (bd.add_row - 1)) * n : (COEF ? a.coef + orow : nullptr);
    const bool carrier = KIND != 2 && coef_row != nullptr;
    cd ph = mk<double>(1.0, 0.0), st = ph;
    if (carrier) {
      const uint32_t kc = (uint32_t)(bd.k_lo + bd.k_len / 2);
      double c, s;
      unit_root_t<double>((kc * (tau_wave + (uint32_t)lane - (KIND == 0 ? 1u : 0u))) & lmask, a.two_over_len, &c, &s);
      ph = mk<double>(c, s);
      unit_root_t<double>((kc * (uint32_t)kWave) & lmask, a.two_over_len, &c, &s);
      st = mk<double>(c, s);
    }
    double* __restrict__ bits_row = a.bits && !part ? a.bits + orow : nullptr;
    const double pscale = part ? 0.0 : a.power_scale;
    double rowacc = 0.0, pl = 0.0;
    __syncthreads();
#pragma unroll 1  // (unrolled, the compiler hoists every group's window reads to the top: 256 registers)
    for (int g = 0; g < GROUPS; ++g) {
      qi_d4 zr = {0.0, 0.0, 0.0, 0.0}, zi = {0.0, 0.0, 0.0, 0.0};
      const double* __restrict__ wr = &win_re[wv][g * GSTEP + a_off];
      const double* __restrict__ wi = &win_im[wv][g * GSTEP + a_off];
#pragma unroll
      for (int q = 0; q < KB; ++q) {
        zr = __builtin_amdgcn_mfma_f64_16x16x4f64(wr[4 * q], wb[q], zr, 0, 0, 0);
        zi = __builtin_amdgcn_mfma_f64_16x16x4f64(wi[4 * q], wb[q], zi, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {  // register r = wave-step 4 g + r: sample tt0 + 64 (4 g + r)
        cd z = mk<double>(zr[r], zi[r]);
        if (carrier) {
          z = cmul_rn(z, ph);
          ph = cmul_rn(ph, st);
        }
        const uint32_t tt = tt0 + (uint32_t)(kWave * (4 * g + r));
        if (coef_row) stream_store(coef_row + tt, z);
        const double m2 = norm2(z.x, z.y);
        if (bits_row) bits_row[tt] = log2_t(sqrt_t(m2) + a.eps);
        const double p = mul_rn(pscale, m2);
        if (want_time) col0[(4 * g + r) * kWave] += p;
        rowacc += p;
        mx = max_t(mx, p);
        pl += plog2p_flat(p, ltab);
      }
    }
    plogp += pl;
    ptot += rowacc;
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
  for (int s = 0; s < STEPS; ++s) {
    const double v = col0[s * kWave];
    tot += v;
    if (time_row) time_row[tt0 + (uint32_t)(kWave * s)] = v;
  }
  if (!want_time) tot = ptot;
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

// ---- fine stage with wave-uniform windows (see Z64FineArgs in qi_native.hpp) -----------------------------------------
// Read-only global data through the constant address space: a load whose address is the same for the whole wave then
// becomes a scalar load (s_load_dword*) -- used for the band descriptors and the wave-uniform carrier factors.  (The
// WINDOWS were first read this way too: correct, and 20 % slower than k_z64_interp -- the scalar cache is no streaming
// path: waves waited 60-70 % of their life for s_load data.  They come through a vector register and v_readlane now.)

#ifdef QI_NATIVE_DEBUG
#define QI_ZDBG(bit) (a.debug & (bit))
#else
#define QI_ZDBG(bit) false
#endif
__device__ __forceinline__ double lane_value64(double v, int lane) {
  const unsigned long long u = (unsigned long long)__double_as_longlong(v);
  const unsigned lo = __builtin_amdgcn_readlane((unsigned)u, lane), hi = __builtin_amdgcn_readlane((unsigned)(u >> 32), lane);
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

// steps whose windows are held in scalar registers together: (GS S + N - 1) coarse samples x 2 registers per part
constexpr int z64f_group(int cls) {
  const int s = z64f_span(cls), nt = z64f_ntap(cls);
  int gs = 8;
  while (gs > 1 && gs * s + nt - 1 > 24) gs /= 2;
  return gs;
}

// One wave: kZ64FineWave consecutive panel samples of the bands row, row + nrow, ... of class CLS.  Output sample
// t = t_wave + 64 s + lane is the coarse-grid position (tau >> LOG2D) + (tau & (D - 1)) / D, tau = t + off (off = n / 2 for
// the Gabor kinds: crop / roll; the zero-padded kind's odd sample is in the envelope, k_z64_gather); its N taps are window
// samples s S + q .. s S + q + N - 1 of the wave's window, q = lane >> LOG2D, which starts N / 2 - 1 coarse samples before
// the wave's first one.  A band's coarse array is [kZ64Pad | M | kZ64Pad] samples, the pads holding the other end's samples
// (k_z64_pad): a wave's window is always a run of consecutive samples -- lane i loads sample m_wave + i (16 bytes, a band
// ahead), and the lanes' arithmetic takes sample e from lane e through scalar registers (v_readlane).
#ifndef QI_Z64_FLIGHT
#define QI_Z64_FLIGHT 2
#endif
template <int KIND, int CLS, bool COEF>
__device__ __forceinline__ void z64_fine_class(const Z64FineArgs& a, const int row, const double (*ltab)[2], double* __restrict__ colv) {
  constexpr int LEVEL = z64f_level(CLS), LOG2D = 6 - LEVEL, S = 1 << LEVEL, N = z64f_ntap(CLS), WL = N + S - 1;
  constexpr int STEPS = kZ64FineSteps, GS = z64f_group(CLS), WIN = GS * S + N - 1;
  static_assert(N / 2 <= kZ64Pad && STEPS % GS == 0, "window reach / group size");
  const int lane = threadIdx.x & (kWave - 1);
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / kWave));
  const int64_t ch = blockIdx.z, n = a.z.n;
  const uint32_t slot = blockIdx.x * (kZ64Threads / kWave) + (uint32_t)wv;  // wave along time = partial slot of its bands
  const uint32_t t_wave = slot * (uint32_t)kZ64FineWave;
  const uint32_t tau_wave = (t_wave + (KIND == 2 ? 0u : (uint32_t)(n / 2))) & ((uint32_t)a.z.Lf - 1u);
  const int32_t m_wave = (int32_t)(tau_wave >> LOG2D) - (N / 2 - 1);  // first window sample of the wave (>= -kZ64Pad)
  double w[WL];
#pragma unroll
  for (int j = 0; j < WL; ++j) w[j] = a.w[j * kWave + lane];
  // per-time power sums of the wave's samples over its bands: in LDS (8 KB per wave), not in 2 x 16 registers per lane
  double* __restrict__ col0 = colv + wv * kZ64FineWave + lane;
#pragma unroll
  for (int s = 0; s < STEPS; ++s) col0[s * kWave] = 0.0;
  double mx = 0.0, plogp = 0.0, ptot = 0.0;  // (ptot: the total power when no per-time sums are kept)
  const bool want_time = a.z.time_part != nullptr;
  const uint32_t tt0 = t_wave + (uint32_t)lane;
  const int64_t zstride = a.z.M + 2 * kZ64Pad;
  constexpr int NEED = STEPS * S + N - 1;  // coarse samples one wave needs per band
  constexpr bool TWO = NEED > kWave;       // ... in two registers of its lanes
  static_assert(NEED <= 2 * kWave, "the window of one wave must fit two registers of its lanes");
  cd smp_next = mk<double>(0.0, 0.0), smq_next = smp_next;
  if (row < a.z.nbands) {
    const cd* __restrict__ Cn = a.z.Z + ((int64_t)ch * a.lvl_bands + a.lvl_index0 + row) * zstride + kZ64Pad + m_wave;
    smp_next = Cn[lane < NEED ? lane : NEED - 1];
    if (TWO) smq_next = Cn[kWave + (lane < NEED - kWave ? lane : NEED - kWave - 1)];
  }
  for (int jj = row; jj < a.z.nbands; jj += a.nrow) {
    const auto bd = as_const(a.z.bands + jj);
    const int32_t out_band = bd->out_band, add_row = bd->add_row;
    // lane i holds coarse sample m_wave + i of the band (requested a band ahead); wave-step s interpolates from samples
    // s S .. s S + N + S - 2 of that window, handed to the lanes' arithmetic through scalar registers (v_readlane)
    const cd smp = smp_next, smq = smq_next;
    if (jj + a.nrow < a.z.nbands) {
      const cd* __restrict__ Cn = a.z.Z + ((int64_t)ch * a.lvl_bands + a.lvl_index0 + jj + a.nrow) * zstride + kZ64Pad + m_wave;
      smp_next = Cn[lane < NEED ? lane : NEED - 1];
      if (TWO) smq_next = Cn[kWave + (lane < NEED - kWave ? lane : NEED - kWave - 1)];
    }
    // (reductions only, no split band: the carrier, of modulus 1, changes no power -- the envelope will do)
    const bool carrier = KIND != 2 && (COEF || add_row != 0);
    cd ph = mk<double>(1.0, 0.0), st = ph;
    if (carrier) {
      // carrier exp(2 pi i k_c (tau - e) / Lf), tau = tau_wave + 64 s + lane: the wave's factor (tau_wave is a multiple of
      // kZ64FineWave: exact phase, table) x the lane's (table) at step 0, advanced by the exact step of 64 samples
      const uint32_t kc = (uint32_t)(bd->k_lo + bd->k_len / 2);
      const auto lp = a.lane_ph + (int64_t)jj * 65;
      const auto wp = as_const(reinterpret_cast<const double*>(a.wave_ph)) +
                      2 * ((kc * (tau_wave / (uint32_t)kZ64FineWave)) & (uint32_t)(a.z.Lf / kZ64FineWave - 1));
      const auto sp = as_const(reinterpret_cast<const double*>(lp + 64));
      ph = cmul_rn(lp[lane], mk<double>(wp[0], wp[1]));
      st = mk<double>(sp[0], sp[1]);
    }
    const int64_t orow = ((int64_t)ch * a.z.panel_bands + out_band) * n;
    // split band (add_row): only the tapered part of its atom -- the samples go to split_part and count for nothing here
    const bool part = add_row != 0;
    cd* __restrict__ coef_row = part ? a.z.split_part + ((int64_t)ch * a.z.split_rows + (add_row - 1)) * n
                                     : (COEF ? a.z.coef + orow : nullptr);
    double* __restrict__ bits_row = a.z.bits && !part ? a.z.bits + orow : nullptr;
    const double pscale = part ? 0.0 : a.z.power_scale;
    double rowacc = 0.0, pl = 0.0;
#pragma unroll
    for (int g = 0; g < STEPS / GS; ++g) {
      double zr[GS], zi[GS];
      // the window of one group and one part at a time (2 (GS S + N - 1) scalar registers)
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int part_i = 0; part_i < 2; ++part_i) {
        double sx[WIN];
#pragma unroll
        for (int i = 0; i < WIN; ++i) {
          const int e = g * GS * S + i;  // window sample (compile-time): lane e of smp, or lane e - 64 of smq
          sx[i] = e < kWave ? lane_value64(part_i == 0 ? smp.x : smp.y, e) : lane_value64(part_i == 0 ? smq.x : smq.y, e - kWave);
        }
#pragma unroll
        for (int s = 0; s < GS; ++s) {
          double acc = 0.0;
#pragma unroll
          for (int j = 0; j < WL; ++j)
            if (j == 0 || !QI_ZDBG(2)) acc = fma(w[j], sx[s * S + j], acc);
          if (part_i == 0) zr[s] = acc;
          else zi[s] = acc;
        }
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int s = 0; s < GS; ++s) {
        cd z = mk<double>(zr[s], zi[s]);
        if (carrier && !QI_ZDBG(16)) {
          z = cmul_rn(z, ph);
          ph = cmul_rn(ph, st);
          zr[s] = z.x;
          zi[s] = z.y;
        }
        // (with the panel stored the store follows its output instead of eight stores back to back at the end of a group:
        // measured neutral to 1 % faster at order 12 x 4 records)
        if (COEF && !QI_ZDBG(1)) stream_store(coef_row + tt0 + (uint32_t)(kWave * (g * GS + s)), z);
        const double p = mul_rn(pscale, norm2(z.x, z.y));
        if (want_time && !QI_ZDBG(8)) col0[(g * GS + s) * kWave] += p;  // (no per-time marginal asked for: no LDS traffic for it)
        rowacc += p;
        mx = max_t(mx, p);
        if (!QI_ZDBG(4)) pl += plog2p_flat(p, ltab);
        if ((s & (QI_Z64_FLIGHT - 1)) == QI_Z64_FLIGHT - 1) {  // two outputs in flight: more only cost registers (round 5, four / eight: zoom stage +3 % / +20 %) (the fence pins the running sums: without it every
          // power of the band stays in registers until a deferred chain of maxima at the end of the loop)
          asm volatile("" : "+v"(mx), "+v"(pl), "+v"(rowacc));
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      // (the stores of a group behind ONE wave-uniform branch each, not one per output)
      if (!COEF && part && !QI_ZDBG(1)) {
#pragma unroll
        for (int s = 0; s < GS; ++s) stream_store(coef_row + tt0 + (uint32_t)(kWave * (g * GS + s)), mk<double>(zr[s], zi[s]));
      }
      if (bits_row) {
#pragma unroll
        for (int s = 0; s < GS; ++s)
          bits_row[tt0 + (uint32_t)(kWave * (g * GS + s))] = log2_t(sqrt_t(norm2(zr[s], zi[s])) + a.z.eps);
      }
    }
    plogp += pl;
    ptot += rowacc;
    if (a.z.part_band && !part) {
      const double rs = wave_sum(rowacc);
      if (lane == 0) a.z.part_band[((int64_t)ch * a.z.panel_bands + out_band) * a.z.pb_stride + slot] = rs;
    }
  }
  double tot = 0.0;
  double* __restrict__ time_row =
      a.z.time_part ? a.z.time_part + ((int64_t)ch * a.z.chunk_total + a.z.chunk_base + row) * n : nullptr;
#pragma unroll
  for (int s = 0; s < STEPS; ++s) {
    const double v = col0[s * kWave];
    tot += v;
    if (time_row) time_row[tt0 + (uint32_t)(kWave * s)] = v;
  }
  if (!want_time) tot = ptot;
  if (a.z.part_stat) {
    const double r0 = wave_max(mx), r1 = wave_sum(tot), r2 = wave_sum(plogp);
    if (lane == 0) {
      double* o = a.z.part_stat + ((int64_t)ch * a.z.stat_stride + (int64_t)(a.z.chunk_base + row) * a.z.stat_nblk + slot) * 3;
      o[0] = r0;
      o[1] = r1;
      o[2] = r2;
    }
  }
}

// one launch per class: blockIdx.y is the row (chunk of the class's band list); registers as the class needs them
template <int KIND, int CLS, bool COEF>
__global__ void __launch_bounds__(kZ64Threads, z64f_ntap(CLS) >= 16 ? 3 : 4) k_z64_fine(Z64FineArgs a) {
  __shared__ double ltab[128][2];  // log2 table of the entropy sums (see log2_pos)
  __shared__ double colv[(kZ64Threads / kWave) * kZ64FineWave];
  const int tid = threadIdx.x;
  if (tid < 128) {
    ltab[tid][0] = kLog2Tab[tid][0];
    ltab[tid][1] = kLog2Tab[tid][1];
  }
  __syncthreads();
  z64_fine_class<KIND, CLS, COEF>(a, blockIdx.y, ltab, colv);
}

// the pads of the coarse arrays: [kZ64Pad | M | kZ64Pad] per band and record, pads = the samples of the other end
__global__ void __launch_bounds__(64) k_z64_pad(cd* __restrict__ Z, int64_t M) {
  cd* z = Z + (int64_t)blockIdx.x * (M + 2 * kZ64Pad);
  const int i = threadIdx.x;
  if (i < kZ64Pad) z[i] = z[M + i];                      // front pad <- last samples
  else if (i < 2 * kZ64Pad) z[M + i] = z[i];             // back pad <- first samples (i - kZ64Pad + kZ64Pad)
}

template <int KIND, bool COEF, int CLS = 0>
void launch_fine_cls(const Z64FineArgs& a, dim3 grid, hipStream_t st) {
  if constexpr (CLS < kZ64FineClasses) {
    if (a.cls == CLS) k_z64_fine<KIND, CLS, COEF><<<grid, kZ64Threads, 0, st>>>(a);
    else launch_fine_cls<KIND, COEF, CLS + 1>(a, grid, st);
  }
}

template <int KIND>
int launch_interp_v(const Z64Args& a, dim3 grid, hipStream_t st) {
  // the grids of Lf / 16 and Lf / 8 samples: interpolation on the matrix pipe (QI_NATIVE_Z64_MFMA=0: the LDS-window kernel)
  static const bool use_mfma = !(tune_env("QI_NATIVE_Z64_MFMA") && atoi(tune_env("QI_NATIVE_Z64_MFMA")) == 0);
  if (use_mfma && (a.log2d == 4 || a.log2d == 3) && kZ64Tile % (4 * 256) == 0 && (a.n / 2) % 256 == 0) {
    const bool coef = a.coef != nullptr;
    if (a.log2d == 4) {
      if (coef) k_z64_mfma<KIND, 4, true><<<grid, kZ64Threads, 0, st>>>(a);
      else k_z64_mfma<KIND, 4, false><<<grid, kZ64Threads, 0, st>>>(a);
    } else {
      if (coef) k_z64_mfma<KIND, 3, true><<<grid, kZ64Threads, 0, st>>>(a);
      else k_z64_mfma<KIND, 3, false><<<grid, kZ64Threads, 0, st>>>(a);
    }
    QI_LAUNCH_CHECK();
    return QI_OK;
  }
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

int launch_z64_pad(cplx<double>* Z, int64_t M, int64_t rows, hipStream_t st) {
  if (rows <= 0) return QI_OK;
  k_z64_pad<<<(unsigned)rows, 64, 0, st>>>(Z, M);
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


int launch_z64_fine(const Z64FineArgs& a, int64_t n_channels, hipStream_t st) {
  if (a.nrow <= 0 || a.z.nbands <= 0) return QI_OK;
  const int64_t per_wg = (int64_t)kZ64FineWave * (kZ64Threads / kWave);
  if (a.z.n % per_wg != 0 || a.z.nblk * kZ64FineWave != a.z.n || (a.z.n / 2) % kZ64FineWave != 0 || (a.z.Lf & (a.z.Lf - 1)) != 0 ||
      a.cls < 0 || a.cls >= kZ64FineClasses || a.z.M != (a.z.Lf >> (6 - z64f_level(a.cls)))) {
    set_error("float64 zoom: a record of %lld samples / class %d not supported", (long long)a.z.n, a.cls);
    return QI_ERR_UNSUPPORTED;
  }
  dim3 grid((unsigned)(a.z.n / per_wg), (unsigned)a.nrow, (unsigned)n_channels);
  const bool coef = a.z.coef != nullptr;
  if (a.z.kind == 0) coef ? launch_fine_cls<0, true>(a, grid, st) : launch_fine_cls<0, false>(a, grid, st);
  else if (a.z.kind == 1) coef ? launch_fine_cls<1, true>(a, grid, st) : launch_fine_cls<1, false>(a, grid, st);
  else coef ? launch_fine_cls<2, true>(a, grid, st) : launch_fine_cls<2, false>(a, grid, st);
  QI_LAUNCH_CHECK();
  return QI_OK;
}

// N taps (nodes -N/2 + 1 .. N/2) of the interpolator to the fraction x in [0, 1) that is exact for the tones at the N / 2
// Chebyshev nodes of the band [-band, band] (radians per coarse sample); solved in long double
static void z64_taps(int N, long double band, long double x, long double* out) {
  const int half = N / 2;
  const long double pi = 3.14159265358979323846264338327950288L;
  long double M[16][17];
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
  for (int c = 0; c < N; ++c) out[c] = M[c][N];
}

// weights of the lanes for class `cls`, layout [window sample j][lane]: lane sits q = lane / D coarse intervals after the
// window's reference sample (index N / 2 - 1) at the fraction x = (lane mod D) / D; its N taps are window samples q .. q + N - 1
void z64_fine_weights(int cls, double* w) {
  const int N = z64f_ntap(cls), WL = z64f_win(cls), log2d = 6 - z64f_level(cls), D = 1 << log2d;
  const long double band = 3.14159265358979323846264338327950288L / (long double)z64f_oversampling(cls);
  for (int lane = 0; lane < kWave; ++lane) {
    const int q = lane >> log2d;
    long double t[16];
    z64_taps(N, band, (long double)(lane & (D - 1)) / (long double)D, t);
    for (int j = 0; j < WL; ++j) w[j * kWave + lane] = (j >= q && j - q < N) ? (double)t[j - q] : 0.0;
  }
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
