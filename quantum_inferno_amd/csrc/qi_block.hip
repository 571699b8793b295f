// Block engine: the wide-spectrum bands of a panel have SHORT atoms in time (a few tens to a thousand taps), so
// they are evaluated as a short-time filter by overlap-save with 4096-point transforms that never leave the CU.
//
// One workgroup (256 threads, 16 values per thread) owns kBlk = 4096 consecutive record samples starting W before
// its first output: it transforms them once (forward, kept in registers) and then, for every band of its list,
// multiplies by the band's 4096-point filter spectrum (a [bands][4096] table that stays in L2), transforms back
// through LDS and keeps the V = 4096 - 2 W outputs that no wrapped tap touched.  Outputs of a thread are the
// samples a + 256 c: every store is a 512-byte run of the panel.  The zero padding of the reference's linear
// correlation (styx_cwt.py:147-198) is the zero extension of the block loads at the record ends; the circular
// Stockwell transform (styx_stx.py:195-236) wraps the loads instead and multiplies the outputs by the
// demodulation phasor exp(-2 pi i idx t / n).  No intermediate in HBM, no pass 1, no edge correction.
//
// 4096 = 16 x 16 x 16 (decimation in frequency): k = k0 + 16 k1 + 256 k2, q = 256 q0 + 16 q1 + q2,
//   y[q] = sum_k0 W16^(k0 q0) W256^(k0 q1) sum_k1 W16^(k1 q1) W4096^((k0 + 16 k1) q2) sum_k2 W16^(k2 q2) Y[k].
// Thread a = k0 + 16 k1 holds Y[a + 256 k2] on entry and y[a + 256 q0] on exit, so forward and inverse chain
// without a reordering.  No MFMA: there is no dense contraction here.
#include "qi_common.hpp"
#include "qi_device.hpp"
#include "qi_native.hpp"
#include "qi_fft_reg.hpp"
#include "qi_zoom_gather.hpp"

namespace qi {
namespace native {

namespace {

constexpr int kBlkThreads = 256;
#ifndef QI_BLK_WAVES
#define QI_BLK_WAVES 3  // waves per SIMD the block kernel is compiled for (register budget 512 / QI_BLK_WAVES)
#endif
#ifndef QI_BLK_LONG_WAVES
#define QI_BLK_LONG_WAVES 2  // the same for the long-block kernels (they hold the even samples of a band while its odd samples are transformed)
#endif
#ifdef QI_NATIVE_DEBUG
#define QI_BDBG(bit) (a.debug & (bit))
#else
#define QI_BDBG(bit) false
#endif
#ifdef QI_NATIVE_STAMPS
#define QI_BSTAMP(k)                                               \
  do {                                                             \
    __builtin_amdgcn_sched_barrier(0);                             \
    const unsigned long long now_ = __builtin_amdgcn_s_memtime();  \
    __builtin_amdgcn_s_waitcnt(0xC07F);                            \
    st_acc[k] += now_ - st_last;                                   \
    st_last = now_;                                                \
    __builtin_amdgcn_sched_barrier(0);                             \
  } while (0)
#else
#define QI_BSTAMP(k)
#endif
constexpr int kBlkPad = 257;  // k0-stride of the second exchange image (conflict-free transposed reads)
constexpr int kBlkRow1 = 272;  // row stride of the first exchange image (see fft4096_tail)
constexpr int kBlkBuf = 16 * kBlkRow1;  // elements of the exchange buffer (>= 16 * kBlkPad)

template <typename T>
__device__ __forceinline__ cplx<T> cconj(cplx<T> v) {
  return mk<T>(v.x, -v.y);
}
// e + W_64^(-K) o
template <typename T, int K>
__device__ __forceinline__ cplx<T> cadd_tw(cplx<T> e, cplx<T> o) {
  const cplx<T> t = mul_tw64<T, K, -1>(o);
  return mk<T>(e.x + t.x, e.y + t.y);
}

// exchange between the half-waves: afterwards lanes 0-31 hold (their own a, the a of lane + 32) and lanes 32-63
// (the b of lane - 32, their own b)
__device__ __forceinline__ void half_swap(float& a, float& b) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  a = __uint_as_float(r[0]);
  b = __uint_as_float(r[1]);
}

__device__ __forceinline__ void half_swap(double& a, double& b) {
  const unsigned long long ua = (unsigned long long)__double_as_longlong(a), ub = (unsigned long long)__double_as_longlong(b);
  const auto lo = __builtin_amdgcn_permlane32_swap((unsigned)ua, (unsigned)ub, false, false);
  const auto hi = __builtin_amdgcn_permlane32_swap((unsigned)(ua >> 32), (unsigned)(ub >> 32), false, false);
  a = __longlong_as_double((long long)(((unsigned long long)hi[0] << 32) | lo[0]));
  b = __longlong_as_double((long long)(((unsigned long long)hi[1] << 32) | lo[1]));
}
// two adjacent complex samples / two adjacent reals of a panel row at sample tt (float: one 16- / 8-byte store)
__device__ __forceinline__ void store_pair(char* row, uint32_t tt, float2 z0, float2 z1) {
  stream_store(reinterpret_cast<float4*>(row + (size_t)(tt * (uint32_t)sizeof(float2))), make_float4(z0.x, z0.y, z1.x, z1.y));
}
__device__ __forceinline__ void store_pair(char* row, uint32_t tt, double2 z0, double2 z1) {
  double2* q = reinterpret_cast<double2*>(row + (size_t)tt * sizeof(double2));
  stream_store(q, z0);
  stream_store(q + 1, z1);
}
__device__ __forceinline__ void store_real_pair(char* row, uint32_t tt, float a, float b) {
  *reinterpret_cast<float2*>(row + (size_t)(tt * (uint32_t)sizeof(float))) = make_float2(a, b);
}
__device__ __forceinline__ void store_real_pair(char* row, uint32_t tt, double a, double b) {
  *reinterpret_cast<double2*>(row + (size_t)tt * sizeof(double)) = make_double2(a, b);
}
// exp(2 pi i x), x formed exactly by the caller (float: x in single precision)
template <typename T>
__device__ __forceinline__ cplx<T> unit_phasor(double two_x) {
  T s, c;
  sincospi_as<T>(two_x, &s, &c);
  return mk<T>(c, s);
}

// S[c] *= r0 * exp(-i pi c / 16), c = 0..15 (the second factor is a compile-time constant: W_64^(-2c))
template <typename T, int... Cs>
__device__ __forceinline__ void rotate_rows16(cplx<T> (&S)[16], cplx<T> r0, std::integer_sequence<int, Cs...>) {
  ((S[Cs] = mul_tw64<T, 2 * Cs, -1>(cmul(S[Cs], r0))), ...);
}

// X[c] = E[c] + conj-twiddle * W_64^(-2 c) * O[c]   (long blocks: the 8192-point spectrum from its two 4096-point halves)
template <typename T, int... Cs>
__device__ __forceinline__ void long_combine(cplx<T> (&X)[16], const cplx<T> (&E)[16], const cplx<T> (&O)[16], cplx<T> w8c,
                                             std::integer_sequence<int, Cs...>) {
  (((void)(X[Cs] = cadd_tw<T, 2 * Cs>(E[Cs], cmul(O[Cs], w8c)))), ...);
}
// S[c] *= r0 * exp(-i pi c / 32), c = 0..15 (W_64^(-c))
template <typename T, int... Cs>
__device__ __forceinline__ void long_rotate(cplx<T> (&S)[16], cplx<T> r0, std::integer_sequence<int, Cs...>) {
  ((S[Cs] = mul_tw64<T, Cs, -1>(cmul(S[Cs], r0))), ...);
}

// v[brev(q)] *= w^q, q = 1..15, powers by products of w, w^2, w^4, w^8 (depth <= 4 roundings)
template <typename T>
__device__ __forceinline__ void mul_powers16(cplx<T> (&v)[16], cplx<T> w) {
  const cplx<T> p1 = w, p2 = cmul(p1, p1), p4 = cmul(p2, p2), p8 = cmul(p4, p4);
  const cplx<T> p3 = cmul(p2, p1), p5 = cmul(p4, p1), p6 = cmul(p4, p2), p7 = cmul(p4, p3);
  v[brev(1, 4)] = cmul(v[brev(1, 4)], p1);
  v[brev(2, 4)] = cmul(v[brev(2, 4)], p2);
  v[brev(3, 4)] = cmul(v[brev(3, 4)], p3);
  v[brev(4, 4)] = cmul(v[brev(4, 4)], p4);
  v[brev(5, 4)] = cmul(v[brev(5, 4)], p5);
  v[brev(6, 4)] = cmul(v[brev(6, 4)], p6);
  v[brev(7, 4)] = cmul(v[brev(7, 4)], p7);
  v[brev(8, 4)] = cmul(v[brev(8, 4)], p8);
  v[brev(9, 4)] = cmul(v[brev(9, 4)], cmul(p8, p1));
  v[brev(10, 4)] = cmul(v[brev(10, 4)], cmul(p8, p2));
  v[brev(11, 4)] = cmul(v[brev(11, 4)], cmul(p8, p3));
  v[brev(12, 4)] = cmul(v[brev(12, 4)], cmul(p8, p4));
  v[brev(13, 4)] = cmul(v[brev(13, 4)], cmul(p8, p5));
  v[brev(14, 4)] = cmul(v[brev(14, 4)], cmul(p8, p6));
  v[brev(15, 4)] = cmul(v[brev(15, 4)], cmul(p8, p7));
}

// 4096-point transform of the workgroup's block.  Entry: v[b] = in[tid + 256 b]; exit: v[brev(c)] = out[tid + 256 c].
// `w` = W4096^tid (inverse sign), `tw256[m]` = W256^m (inverse sign); DIR = -1 conjugates both.
// Column order: the thread of lane L in wave wv owns column col = 64 wv + (L < 32 ? 2 L : 2 (L - 32) + 1) on entry
// and on exit, i.e. lanes L and L + 32 hold ADJACENT samples -- after one v_permlane32_swap per register pair a lane
// holds two adjacent outputs and stores them as 16 bytes (the epilogue is store-issue bound: half the store
// instructions).  `w` = W4096^col.
template <typename T, int DIR>
__device__ __forceinline__ void fft4096_tail(cplx<T> (&v)[16], cplx<T>* __restrict__ buf,
                                             const cplx<T>* __restrict__ tw256, int tid, int col);
template <typename T, int DIR>
__device__ __forceinline__ void fft4096(cplx<T> (&v)[16], cplx<T>* __restrict__ buf, const cplx<T>* __restrict__ tw256,
                                        cplx<T> w, int tid, int col) {
  fft_reg<T, 16, DIR>(v);  // over k2 -> q2
  // the powers of w do not depend on the band: hide that from the optimiser, which would otherwise keep all fifteen
  // in registers across the band loop
  asm volatile("" : "+v"(w.x), "+v"(w.y));
  mul_powers16<T>(v, DIR > 0 ? w : cconj<T>(w));
  fft4096_tail<T, DIR>(v, buf, tw256, tid, col);
}

// A spectrum with at most ONE non-zero value y among a thread's sixteen (bin k = col + 256 b0): the first pass of the
// inverse transform and its twiddles collapse to v[brev(q)] = y om^q, om = exp(2 pi i k / 4096) (binary products,
// depth <= 4 roundings like mul_powers16).
template <typename T>
__device__ __forceinline__ void sparse_head16(cplx<T> (&v)[16], cplx<T> y, cplx<T> om) {
  const cplx<T> u1 = om, u2 = cmul(u1, u1), u4 = cmul(u2, u2), u8 = cmul(u4, u4);
  v[brev(0, 4)] = y;
  v[brev(1, 4)] = cmul(y, u1);
  v[brev(2, 4)] = cmul(y, u2);
  v[brev(3, 4)] = cmul(v[brev(2, 4)], u1);
  v[brev(4, 4)] = cmul(y, u4);
  v[brev(5, 4)] = cmul(v[brev(4, 4)], u1);
  v[brev(6, 4)] = cmul(v[brev(4, 4)], u2);
  v[brev(7, 4)] = cmul(v[brev(6, 4)], u1);
  v[brev(8, 4)] = cmul(y, u8);
  v[brev(9, 4)] = cmul(v[brev(8, 4)], u1);
  v[brev(10, 4)] = cmul(v[brev(8, 4)], u2);
  v[brev(11, 4)] = cmul(v[brev(10, 4)], u1);
  v[brev(12, 4)] = cmul(v[brev(8, 4)], u4);
  v[brev(13, 4)] = cmul(v[brev(12, 4)], u1);
  v[brev(14, 4)] = cmul(v[brev(12, 4)], u2);
  v[brev(15, 4)] = cmul(v[brev(14, 4)], u1);
}

// the value of a thread's sixteen that index b0 selects, b0 in {ba, ba + 1 mod 16} with ba the same for the workgroup
// (a switch over constant indices: a computed index would move the sixteen values to scratch memory)
template <typename T>
__device__ __forceinline__ cplx<T> pick_pair16(const cplx<T> (&S)[16], int ba, bool first) {
  cplx<T> xa, xb;
#define QI_PICK(B)                                                                \
  case B:                                                                         \
    xa = S[B];                                                                    \
    xb = S[(B + 1) & 15];                                                         \
    asm volatile("" : "+v"(xa.x), "+v"(xa.y), "+v"(xb.x), "+v"(xb.y)); /* keeps the cases apart (no index table) */ \
    break;
  switch (ba) {
    QI_PICK(0) QI_PICK(1) QI_PICK(2) QI_PICK(3) QI_PICK(4) QI_PICK(5) QI_PICK(6) QI_PICK(7)
    QI_PICK(8) QI_PICK(9) QI_PICK(10) QI_PICK(11) QI_PICK(12) QI_PICK(13) QI_PICK(14)
    default:
      xa = S[15];
      xb = S[0];
      asm volatile("" : "+v"(xa.x), "+v"(xa.y), "+v"(xb.x), "+v"(xb.y));
      break;
  }
#undef QI_PICK
  return first ? xa : xb;
}

// everything after the first radix-16 pass and its twiddles: two exchanges through LDS, two register passes
template <typename T, int DIR>
__device__ __forceinline__ void fft4096_tail(cplx<T> (&v)[16], cplx<T>* __restrict__ buf,
                                             const cplx<T>* __restrict__ tw256, int tid, int col) {
  __syncthreads();  // the previous transform's readers are done with buf
  // First exchange image: row q2 (stride kBlkRow1 = 272: a row and the next one sit 16 elements apart modulo 32), column
  // c at position c ^ ((c >> 4) & 1).  With the column order of this kernel (lanes 0-31 even, 32-63 odd columns) the
  // sixteen lanes of a ds_write_b64 group then hit sixteen different bank pairs, and so do the 2 x 16 lanes of a
  // ds_read_b64 group below (k0 = 0..15 of rows q2t and q2t + 1): no bank conflicts on either side.
  {
    cplx<T> t[16];
#pragma unroll
    for (int q2 = 0; q2 < 16; ++q2) t[q2] = v[brev(q2, 4)];
    // (16-byte elements -- double2 --: a ds_write_b128 is served in groups of EIGHT contiguous lanes over 32 banks, i.e. the
    // eight columns 0, 2, ... 14 of a group must fall on eight different 16-byte slots modulo 128 bytes: the swap bit is bit 3
    // of the column there, bit 4 for the 8-byte elements' groups of sixteen.  Round 4's counters had 35 % of k_block64's LDS
    // cycles as bank conflicts, all of them this store's 2-way ones.)
    constexpr int SW = sizeof(cplx<T>) == 16 ? 3 : 4;
    const int pos = col ^ ((col >> SW) & 1);
#pragma unroll
    for (int q2 = 0; q2 < 16; ++q2) buf[q2 * kBlkRow1 + pos] = t[q2];
  }
  __syncthreads();
  const int k0 = tid & 15, q2t = tid >> 4;
#pragma unroll
  for (int k1 = 0; k1 < 16; ++k1)  // column 16 k1 + k0 sits at the position the store's swap gave it
    v[k1] = buf[q2t * kBlkRow1 + 16 * k1 + (sizeof(cplx<T>) == 16 ? (k0 ^ ((k0 >> 3) & 1)) : (k0 ^ (k1 & 1)))];
  fft_reg<T, 16, DIR>(v);  // over k1 -> q1
#pragma unroll
  for (int q1 = 1; q1 < 16; ++q1) {
    const cplx<T> t = tw256[(k0 * q1) & 255];
    v[brev(q1, 4)] = cmul(v[brev(q1, 4)], DIR > 0 ? t : cconj<T>(t));
  }
  __syncthreads();
#pragma unroll
  for (int q1 = 0; q1 < 16; ++q1)  // column 16 q1 + q2t goes to the slot of the thread that owns it (see above)
    buf[k0 * kBlkPad + 64 * (q1 >> 2) + 8 * (q1 & 3) + 32 * (q2t & 1) + (q2t >> 1)] = v[brev(q1, 4)];
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 16; ++k) v[k] = buf[k * kBlkPad + tid];
  fft_reg<T, 16, DIR>(v);  // over k0 -> q0
}

// Spectrum of record samples [t0, t0 + 4096) (CIRC: the record wraps -- Stockwell; else it is zero outside -- the
// reference's zero padding), in natural order: S[c] = bin col + 256 c.
template <typename T, bool CIRC>
__device__ __forceinline__ void block_forward(const T* __restrict__ sig, int64_t n, int64_t t0, cplx<T> (&S)[16],
                                              cplx<T>* __restrict__ buf, const cplx<T>* __restrict__ tw256, cplx<T> w,
                                              int tid, int col) {
#pragma unroll
  for (int b = 0; b < 16; ++b) {
    int64_t t = t0 + col + 256 * b;
    T x;
    if (CIRC) {
      t = (t + n) & (n - 1);  // circular (n is a power of two, t >= -n)
      x = sig[t];
    } else {
      x = (t >= 0 && t < n) ? sig[t] : T(0);
    }
    S[b] = mk<T>(x, T(0));
  }
  fft4096<T, -1>(S, buf, tw256, w, tid, col);
  cplx<T> t[16];
#pragma unroll
  for (int c = 0; c < 16; ++c) t[c] = S[brev(c, 4)];
#pragma unroll
  for (int c = 0; c < 16; ++c) S[c] = t[c];
}

// The bands [band_first, band_first + band_count) of the launch's list on block `blk` of reach group WQ, from the
// block's spectrum S (natural order; rotated in place here for the Gabor banks).  DEMOD: Stockwell (demodulated outputs).
// EDGE: the bands are the split bands band_first ... of the styx table (no descriptors: panel rows from a.edge_band, filter
// spectra from piece `edge_piece` of a.edge_bank), S is the spectrum of that far piece of the record, and the zoom engine's
// part of every output (a.edge_part) is added before the band is finished (k_block_edge).
template <typename T, int WQ, bool DEMOD, bool COEF, bool BITS, bool EDGE = false>
__device__ __forceinline__ void block_bands(const BlockArgs<T>& a, int32_t blk_i, int32_t band_first, int32_t band_count,
                                            int32_t plane, int32_t stat_slot, cplx<T> (&S)[16], cplx<T>* __restrict__ buf,
                                            const cplx<T>* __restrict__ tw256, double (*s_red)[kBlkThreads / kWave],
                                            cplx<T> w, int edge_piece = 0) {
  static_assert(!EDGE || !DEMOD, "split bands belong to the styx table");
  constexpr int W = 256 * WQ, V = kBlk - 2 * W, NOUT = 16 - 2 * WQ, NW = kBlkThreads / kWave;
  constexpr bool F64 = sizeof(T) == 8;
  const int tid = threadIdx.x, lane = tid & (kWave - 1), wv = tid / kWave;
  const int col = kWave * wv + (lane < 32 ? 2 * lane : 2 * (lane - 32) + 1);  // the column this thread owns
  const int64_t blk = blk_i, ch = blockIdx.z;
  const int64_t n = a.n;
#ifdef QI_NATIVE_STAMPS
  unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long st_last = __builtin_amdgcn_s_memtime();
#endif
  // record samples [t0, t0 + 4096), outputs [t0 + W, t0 + W + V)
  const int64_t t0 = blk * V - W;
  if (!DEMOD && !EDGE) {
    // Gabor banks: the half-sample offset of the atoms (styx_cwt.py:113-144) is the factor exp(-i theta_k / 2) of
    // every filter spectrum; it goes into the block spectrum once, which leaves REAL Gaussian weights per band
    const cplx<T> r0 = unit_phasor<T>(-(double)col / (double)kBlk);
    rotate_rows16<T>(S, r0, std::make_integer_sequence<int, 16>{});
  }

  QI_BSTAMP(0);
  T col_p[NOUT];  // per-time power sums of this thread's outputs
#pragma unroll
  for (int i = 0; i < NOUT; ++i) col_p[i] = T(0);
  T mx = T(0);
  double plogp = 0.0;
  const uint32_t tb0 = (uint32_t)(t0 + W + col);  // this thread's first output sample
  // the 16-byte stores: lanes 0-31 write the pair (own, lane + 32) of output i, lanes 32-63 the pair of output i + 1
  const uint32_t tb_pair = (uint32_t)(t0 + W + kWave * wv + 2 * (lane & 31)) + (lane < 32 ? 0u : 256u);
  int pending = -1, par = 0;  // band whose wave sums sit in s_red[par ^ 1] until a barrier has passed

  // the band descriptor is fetched one band ahead: its load would otherwise sit in front of the filter loads
  BlockBandT<T> bd_next{};
  if constexpr (!EDGE) bd_next = load_uniform(a.bands + band_first);
  for (int jj = 0; jj < band_count; ++jj) {
    BlockBandT<T> bd = bd_next;
    if constexpr (EDGE) {
      bd.out_band = a.edge_band[band_first + jj];
    } else {
      if (jj + 1 < band_count) bd_next = load_uniform(a.bands + band_first + jj + 1);
    }
    cplx<T> v[16];  // (F64: float64 tables hold analytic bands only)
    if (!EDGE && bd.narrow == 1) {
      // narrow filter spectrum (<= 256 bins from klo): this thread's only bin with a weight above 2^-30 of the peak is
      // k = klo + ((col - klo) mod 256); the first pass of the inverse transform is y om^q (sparse_head16)
      QI_BSTAMP(1);
      QI_BSTAMP(2);
      const int kres = (col - bd.klo) & 255;
      const int k = (bd.klo + kres) & (kBlk - 1);
      const bool first = (k >> 8) == (bd.klo >> 8);
      const cplx<T> x = pick_pair16<T>(S, __builtin_amdgcn_readfirstlane(bd.klo >> 8), first);
      T r;
      if constexpr (F64) {
        // (float64: the plan-time weight table holds this bin's weight -- block_bands' formula in double, aliases and signs
        // included; `narrow` is set there only when every weight above 2^-52 of the peak lies inside the 256-bin window)
        r = a.gauss_w[(int64_t)(band_first + jj) * kBlk + k];
      } else {
        T dk = (T)(k - bd.kappa_int) - (T)bd.kappa_frac;
        T amp = (T)bd.amp;
        if (dk > (T)(kBlk / 2)) {
          dk -= (T)kBlk;
          if (!DEMOD) amp = -amp;  // half-integer sample grid: the aliases alternate in sign
        }
        if (DEMOD && dk < -(T)(kBlk / 2)) dk += (T)kBlk;
        const T e = (T)bd.cw * dk;
        r = amp * fast_exp2(-e * e);
      }
      cplx<T> wq = w;
      asm volatile("" : "+v"(wq.x), "+v"(wq.y));
      const cplx<T> om = cmul(wq, first ? mk<T>((T)bd.rot_a[0], (T)bd.rot_a[1]) : mk<T>((T)bd.rot_b[0], (T)bd.rot_b[1]));
      sparse_head16<T>(v, mk<T>(x.x * r, x.y * r), om);
      if (!QI_BDBG(4)) fft4096_tail<T, 1>(v, buf, tw256, tid, col);
    } else if (!EDGE && F64 && a.gauss_w) {
      // float64: the band's 4096 real Gaussian weights from a table made at plan time (32 KB per band, L2-resident: every
      // block and record multiplies by the same weights) -- in double an exponential is ~22 instructions, sixteen of them
      // per band and thread were a fifth of the kernel's arithmetic (float32 evaluates them in registers: v_exp_f32)
      const T* __restrict__ gw = a.gauss_w + (int64_t)(band_first + jj) * kBlk + col;
      T r[16];
#pragma unroll
      for (int b = 0; b < 16; ++b) r[b] = gw[256 * b];
#pragma unroll
      for (int b = 0; b < 16; ++b) v[b] = mk<T>(S[b].x * r[b], S[b].y * r[b]);
    } else if (!EDGE && (F64 || bd.analytic)) {
      // Gaussian filter spectrum in registers: no table traffic (the table rows cost as much L2 bandwidth as the
      // panel costs HBM bandwidth)
      QI_BSTAMP(1);
      QI_BSTAMP(2);
      // (narrow = 2: every weight above 2^-30 of the peak belongs to a bin below kBlk / 2 -- no wrap-around among the
      // lower eight values of a thread, and the upper eight are taken as zero)
      const bool lower = bd.narrow == 2;
      if (bd.nowrap) {
        // no alias of the filter spectrum matters: weight(k) = exp2(la - (cw (k - kappa))^2) straight from the bin index
        // (k - kappa_int is an exact float; the fractional part of kappa enters through the fused multiply-add)
        const T kf0 = (T)(col - bd.kappa_int), cwf = -(T)bd.cw * (T)bd.kappa_frac, cw = (T)bd.cw, la = (T)bd.la;
#pragma unroll
        for (int b = 0; b < 8; ++b) {
          const T e = fma_t(cw, kf0 + (T)(256 * b), cwf);
          const T r = fast_exp2(fma_t(-e, e, la));
          v[b] = mk<T>(S[b].x * r, S[b].y * r);
        }
        if (lower) {
#pragma unroll
          for (int b = 8; b < 16; ++b) v[b] = mk<T>(T(0), T(0));
        } else {
#pragma unroll
          for (int b = 8; b < 16; ++b) {
            const T e = fma_t(cw, kf0 + (T)(256 * b), cwf);
            const T r = fast_exp2(fma_t(-e, e, la));
            v[b] = mk<T>(S[b].x * r, S[b].y * r);
          }
        }
      } else {
#pragma unroll
      for (int b = 0; b < 8; ++b) {
        T dk = (T)(col + 256 * b - bd.kappa_int) - (T)bd.kappa_frac;
        T amp = (T)bd.amp;
        if (!lower && dk > (T)(kBlk / 2)) {
          dk -= (T)kBlk;
          if (!DEMOD) amp = -amp;  // half-integer sample grid: the aliases alternate in sign
        }
        if (DEMOD && !lower && dk < -(T)(kBlk / 2)) dk += (T)kBlk;
        const T e = (T)bd.cw * dk;
        const T r = amp * fast_exp2(-e * e);
        v[b] = mk<T>(S[b].x * r, S[b].y * r);
      }
      if (lower) {
#pragma unroll
        for (int b = 8; b < 16; ++b) v[b] = mk<T>(T(0), T(0));
      } else {
#pragma unroll
        for (int b = 8; b < 16; ++b) {
          T dk = (T)(col + 256 * b - bd.kappa_int) - (T)bd.kappa_frac;
          T amp = (T)bd.amp;
          if (dk > (T)(kBlk / 2)) {
            dk -= (T)kBlk;
            if (!DEMOD) amp = -amp;
          }
          if (DEMOD && dk < -(T)(kBlk / 2)) dk += (T)kBlk;
          const T e = (T)bd.cw * dk;
          const T r = amp * fast_exp2(-e * e);
          v[b] = mk<T>(S[b].x * r, S[b].y * r);
        }
      }
      }
    } else {
      const cplx<T>* __restrict__ H =
          (EDGE ? a.edge_bank + ((int64_t)(band_first + jj) * 2 + edge_piece) * kBlk : a.bank + (int64_t)bd.bank_row * kBlk) + col;
      cplx<T> h[16];
#pragma unroll
      for (int b = 0; b < 16; ++b) h[b] = QI_BDBG(2) ? S[(b + 1) & 15] : H[256 * b];
      QI_BSTAMP(1);
#ifdef QI_NATIVE_STAMPS
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
      QI_BSTAMP(2);
#pragma unroll
      for (int b = 0; b < 16; ++b) v[b] = cmul(S[b], h[b]);
    }
    if ((EDGE || bd.narrow != 1) && !QI_BDBG(4)) fft4096<T, 1>(v, buf, tw256, w, tid, col);
    QI_BSTAMP(3);
    if (pending >= 0 && tid == 0) {
      double r = 0.0;
      for (int q = 0; q < NW; ++q) r += s_red[par ^ 1][q];
      a.part_band[((int64_t)ch * a.panel_bands + pending) * a.nblk + blk] = r;
    }

    cplx<T> ph = mk<T>(T(1), T(0));
    if (DEMOD && !F64) {
      // exp(-2 pi i idx t / n) at this thread's first output; idx * t mod n is exact in 32-bit wraparound
      const uint32_t m = (0u - (uint32_t)bd.shift * tb0) & (uint32_t)(n - 1);
      double s, c;
      unit_root_t<T>(m, a.two_over_n, &c, &s);
      ph = mk<T>((T)c, (T)s);
    }
    if constexpr (DEMOD && F64 && COEF) {  // (reductions only: the demodulation, of modulus 1, changes no power)
      // float64: exp(-2 pi i idx t / n) of output i (t = V blk + col + 256 i) as a product of three exact-phase factors from
      // tables -- no double-precision sincospi per band and thread, and no phasor state carried through the epilogue (seed,
      // running power and two steps were 16 registers that did not fit beside the block spectrum: 270-390 bytes of scratch
      // per lane in round 3): the block's and the column's factors (two table entries each, exp(-2 pi i m / n) =
      // t1[m >> 10] t2[m & 1023]) give `ph` here, the 256 i samples' factor is a wave-uniform table entry per output.
      const uint32_t nm = (uint32_t)(n - 1);
      const uint32_t mA = ((uint32_t)bd.shift * (uint32_t)(blk * V)) & nm;  // (the same for the whole workgroup; idx t mod n
      const uint32_t mB = ((uint32_t)bd.shift * (uint32_t)col) & nm;            // is exact in 32-bit wraparound)
      const auto t1s = as_const(reinterpret_cast<const T*>(a.demod_t1) + 2 * (mA >> 10));
      const auto t2s = as_const(reinterpret_cast<const T*>(a.demod_t2) + 2 * (mA & 1023u));
      const cplx<T> fa = cmul_rn(mk<T>(t1s[0], t1s[1]), mk<T>(t2s[0], t2s[1]));
      const cplx<T> fb = cmul_rn(a.demod_t1[mB >> 10], a.demod_t2[mB & 1023u]);
      ph = cmul_rn(fa, fb);
    }
    [[maybe_unused]] const auto demod_pow = as_const(reinterpret_cast<const T*>(a.demod_pow) + 32 * (int64_t)(band_first + jj));
    const int64_t orow = ((int64_t)ch * a.panel_bands + bd.out_band) * n;
    char* __restrict__ coef_row = reinterpret_cast<char*>(a.coef ? a.coef + orow : nullptr);
    char* __restrict__ bits_row = reinterpret_cast<char*>(a.bits ? a.bits + orow : nullptr);
    const cplx<T>* __restrict__ part = EDGE ? a.edge_part + (ch * a.nsplit + band_first + jj) * n : nullptr;
    T rowacc = T(0), pl = T(0);
    // demodulation phasor of output i: ph * r^i (r: 256 samples); every fourth one from the exact power r^4 (at most
    // three roundings), the ones between advance by r -- held one at a time, not as an array of NOUT
    [[maybe_unused]] const cplx<T> R1 = mk<T>((T)bd.rot[0], (T)bd.rot[1]), R4 = mk<T>((T)bd.rot[4], (T)bd.rot[5]);
    uint32_t tp = tb_pair;
    asm volatile("" : "+v"(tp));  // keep the band-invariant addresses out of the loop-invariant hoisting
    // Two outputs at a time: demodulate, bring the two ADJACENT samples of a pair into one lane (half_swap), then
    // everything else -- powers, sums, the 16-byte store -- on the pair.  `guard`: the block reaches past the end of the
    // record (its last block only): pairs outside are neither stored nor summed.
    auto finish_band = [&](auto guard) {
      constexpr bool GUARD = decltype(guard)::value;
      cplx<T> seed = ph, rcur = ph;
#pragma unroll
      for (int i = 0; i < NOUT; i += 2) {
        cplx<T> z[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          z[h] = v[brev(i + h + WQ, 4)];
          if constexpr (DEMOD && F64) {
            if constexpr (COEF)
              if (!QI_BDBG(32)) z[h] = cmul_rn(z[h], cmul_rn(ph, mk<T>(demod_pow[2 * (i + h)], demod_pow[2 * (i + h) + 1])));
          } else if (DEMOD && !QI_BDBG(32)) {
            if (i + h > 0) {
              if (((i + h) & 3) == 0) {
                seed = cmul_rn(seed, R4);
                rcur = seed;
              } else {
                rcur = cmul_rn(rcur, R1);
              }
            }
            z[h] = cmul_rn(z[h], rcur);
          }
        }
        half_swap(z[0].x, z[1].x);
        half_swap(z[0].y, z[1].y);
        const uint32_t tt = tp + 256u * (uint32_t)i;  // first sample of this lane's pair
        const bool inside = !GUARD || tt < (uint32_t)n;
        if constexpr (EDGE) {
          if (inside) {  // (16 bytes per lane: the pair is adjacent in the zoom engine's row, too)
            const cplx<T> pa = part[tt], pb = part[tt + 1];
            z[0].x += pa.x;
            z[0].y += pa.y;
            z[1].x += pb.x;
            z[1].y += pb.y;
          }
        }
        T lg[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const T m2 = norm2(z[h].x, z[h].y);
          if (BITS) lg[h] = log2_t(sqrt_t(m2) + a.eps);
          const T p = inside ? mul_rn(a.power_scale, m2) : T(0);
          col_p[i + h] += p;
          rowacc += p;
          mx = max_t(mx, p);
          if constexpr (F64) {  // (the entropy logarithm's table in LDS behind the twiddles -- k_block64 puts it there --, no branch)
            if (!QI_BDBG(16)) pl += plog2p_flat(p, reinterpret_cast<const double (*)[2]>(tw256 + 256));
          } else {
            if (!QI_BDBG(16)) pl += plog2p(p);
          }
        }
        if (COEF && inside && !QI_BDBG(1)) store_pair(coef_row, tt, z[0], z[1]);
        if (BITS && inside) store_real_pair(bits_row, tt, lg[0], lg[1]);
      }
    };
    if (t0 + W + V > n) finish_band(std::true_type{});
    else finish_band(std::false_type{});
    plogp += (double)pl;
    if (a.part_band && !QI_BDBG(64)) {
      const double r = wave_sum((double)rowacc);
      if (lane == 0) s_red[par][wv] = r;
      pending = bd.out_band;
      par ^= 1;
    }
    QI_BSTAMP(4);
  }
#ifdef QI_NATIVE_STAMPS
  if (a.stamps && tid == 0) {
    unsigned long long* o = a.stamps + ((int64_t)blockIdx.z * gridDim.x + blockIdx.x) * 8;
    for (int k = 0; k < 5; ++k) o[k] = st_acc[k];
    o[5] = (unsigned long long)band_count;
  }
#endif

  T tot = T(0);
  char* __restrict__ time_row = reinterpret_cast<char*>(
      a.time_part ? a.time_part + ((int64_t)ch * a.chunk_total + a.chunk_base + plane) * n : nullptr);
#pragma unroll
  for (int i = 0; i < NOUT; i += 2) {  // (col_p holds the sums of this lane's PAIRS: see finish_band)
    tot += col_p[i] + col_p[i + 1];
    const uint32_t tt = tb_pair + 256u * (uint32_t)i;
    if (time_row && tt < (uint32_t)n) store_real_pair(time_row, tt, col_p[i], col_p[i + 1]);
  }
  const double r0 = wave_max((double)mx), r1 = wave_sum((double)tot), r2 = wave_sum(plogp);
  __syncthreads();  // the last band's wave sums are visible; buf is free
  if (pending >= 0 && tid == 0) {
    double r = 0.0;
    for (int q = 0; q < NW; ++q) r += s_red[par ^ 1][q];
    a.part_band[((int64_t)ch * a.panel_bands + pending) * a.nblk + blk] = r;
  }
  if (a.part_stat) {
    double* fin = reinterpret_cast<double*>(buf);
    if (lane == 0) {
      fin[wv] = r0;
      fin[NW + wv] = r1;
      fin[2 * NW + wv] = r2;
    }
    __syncthreads();
    if (tid == 0) {
      double m = 0.0, s1 = 0.0, s2 = 0.0;
      for (int q = 0; q < NW; ++q) {
        m = fin[q] > m ? fin[q] : m;
        s1 += fin[NW + q];
        s2 += fin[2 * NW + q];
      }
      double* o = a.part_stat + ((int64_t)ch * a.stat_stride + a.stat_base + stat_slot) * 3;
      o[0] = m;
      o[1] = s1;
      o[2] = s2;
    }
  }
}

// One work item: block `it.block` of the reach group WQ (taps within W = 256 WQ samples), bands
// [it.band_first, it.band_first + it.band_count) of the launch's list.  DEMOD: Stockwell (circular loads, demodulated
// outputs).
template <typename T, int WQ, bool DEMOD, bool COEF, bool BITS>
__device__ __forceinline__ void block_item(const BlockArgs<T>& a, const BlockItem& it, cplx<T>* __restrict__ buf,
                                           const cplx<T>* __restrict__ tw256, double (*s_red)[kBlkThreads / kWave],
                                           cplx<T> w) {
  const int tid = threadIdx.x, lane = tid & (kWave - 1);
  const int col = (tid & ~(kWave - 1)) + (lane < 32 ? 2 * lane : 2 * (lane - 32) + 1);
  cplx<T> S[16];
  block_forward<T, DEMOD>(a.sig + (int64_t)blockIdx.z * a.n, a.n, (int64_t)it.block * (kBlk - 512 * WQ) - 256 * WQ, S, buf, tw256,
                          w, tid, col);
  block_bands<T, WQ, DEMOD, COEF, BITS>(a, it.block, it.band_first, it.band_count, it.plane, it.stat_slot, S, buf, tw256,
                                        s_red, w);
}


// ---- long blocks: 8192 record samples, the narrow Gaussian bands of the 1024-sample reach group ----------------------
// Half of a 4096-sample block of that group is overlap.  An 8192-sample block keeps 6144 of its outputs (75 %) and costs
// no more per point: the spectrum of the 8192 samples (bins k < 4096 only: the bands are analytic) is
// X[k] = E[k] + W_8192^-k O[k] from the 4096-point transforms of the even and the odd samples, and -- the upper half of
// the filtered spectrum being empty -- the 8192-point inverse is two 4096-point inverses, of Y[k] for the even output
// samples and of Y[k] W_8192^k for the odd ones.  A thread ends up with both samples of a pair (2 q', 2 q' + 1): one
// 16-byte store per pair, a wave's stores are one kilobyte run.
template <typename T, bool CIRC>
__device__ __forceinline__ void block_forward8(const T* __restrict__ sig, int64_t n, int64_t t0, cplx<T> (&S)[16],
                                               cplx<T>* __restrict__ buf, const cplx<T>* __restrict__ tw256, cplx<T> w,
                                               cplx<T> w8, int tid, int col) {
  cplx<T> O[16];
#pragma unroll
  for (int par = 1; par >= 0; --par) {  // odd samples first (kept in O), then the even ones (in S)
#pragma unroll
    for (int b = 0; b < 16; ++b) {
      int64_t t = t0 + 2 * (col + 256 * b) + par;
      T x;
      if (CIRC) {
        t = (t + n) & (n - 1);
        x = sig[t];
      } else {
        x = (t >= 0 && t < n) ? sig[t] : T(0);
      }
      S[b] = mk<T>(x, T(0));
    }
    fft4096<T, -1>(S, buf, tw256, w, tid, col);
    if (par == 1) {
#pragma unroll
      for (int c = 0; c < 16; ++c) O[c] = S[brev(c, 4)];
    }
  }
  // X[col + 256 c] = E + W_8192^-(col + 256 c) O,  W_8192^-(256 c) = W_64^-(2 c)
  const cplx<T> w8c = cconj<T>(w8);
  cplx<T> t[16];
#pragma unroll
  for (int c = 0; c < 16; ++c) t[c] = S[brev(c, 4)];
  long_combine<T>(S, t, O, w8c, std::make_integer_sequence<int, 16>{});
}

// The bands [band_first, band_first + band_count) of the launch's list on long block `blk_i`, from its spectrum S
// (natural order, bins col + 256 c of the 8192-bin grid).
template <typename T, bool DEMOD, bool COEF, bool BITS>
__device__ __forceinline__ void long_bands(const BlockArgs<T>& a, int32_t blk_i, int32_t band_first, int32_t band_count,
                                           int32_t plane, int32_t stat_slot, cplx<T> (&S)[16], cplx<T>* __restrict__ buf,
                                           const cplx<T>* __restrict__ tw256, double (*s_red)[kBlkThreads / kWave],
                                           cplx<T> w, cplx<T> w8) {
  constexpr int W = 1024, V = kBlkLongValid, NOUT = 12, C0 = 2, NW = kBlkThreads / kWave;  // pairs c = 2 .. 13 of a thread are kept
  const int tid = threadIdx.x, lane = tid & (kWave - 1), wv = tid / kWave;
  const int col = kWave * wv + (lane < 32 ? 2 * lane : 2 * (lane - 32) + 1);
  const int64_t blk = blk_i, ch = blockIdx.z, n = a.n;
  const int64_t t0 = blk * V - W;  // record samples [t0, t0 + 8192), outputs [t0 + W, t0 + W + V)
  if (!DEMOD) {
    // half-sample offset of the Gabor atoms: exp(-i theta_k / 2) = exp(-i pi k / 8192), k = col + 256 c
    float sn, cs;
    sincospif(-(float)col * (1.0f / (float)kBlkLong), &sn, &cs);
    long_rotate<T>(S, mk<T>((T)cs, (T)sn), std::make_integer_sequence<int, 16>{});
  }
  T col_p[2 * NOUT];
#pragma unroll
  for (int i = 0; i < 2 * NOUT; ++i) col_p[i] = T(0);
  T mx = T(0);
  double plogp = 0.0;
  const uint32_t tb0 = (uint32_t)(t0 + 2 * (col + 256 * C0));  // first sample of this thread's first kept pair
  int pending = -1, par = 0;
  BlockBand bd_next = load_uniform(a.bands + band_first);
  for (int jj = 0; jj < band_count; ++jj) {
    const BlockBand bd = bd_next;
    if (jj + 1 < band_count) bd_next = load_uniform(a.bands + band_first + jj + 1);
    // this thread's only bin with a weight above 2^-30 of the peak
    const int kres = (col - bd.klo) & 255;
    const int k = (bd.klo + kres) & (kBlk - 1);  // (the band lies in the lower half of the 8192-bin grid)
    const bool first = (k >> 8) == (bd.klo >> 8);
    const cplx<T> x = pick_pair16<T>(S, __builtin_amdgcn_readfirstlane(bd.klo >> 8), first);
    const T dk = (T)(k - bd.kappa_int) - (T)bd.kappa_frac;
    const T e = (T)bd.cw * dk;
    const T r = (T)bd.amp * fast_exp2(-e * e);
    const cplx<T> y = mk<T>(x.x * r, x.y * r);
    cplx<T> wq = w, wq8 = w8;
    asm volatile("" : "+v"(wq.x), "+v"(wq.y), "+v"(wq8.x), "+v"(wq8.y));
    const cplx<T> om = cmul(wq, first ? mk<T>((T)bd.rot_a[0], (T)bd.rot_a[1]) : mk<T>((T)bd.rot_b[0], (T)bd.rot_b[1]));
    const cplx<T> tw_odd = cmul(wq8, first ? mk<T>((T)bd.rot8_a[0], (T)bd.rot8_a[1]) : mk<T>((T)bd.rot8_b[0], (T)bd.rot8_b[1]));
    cplx<T> v[16];
    cplx<T> ph = mk<T>(T(1), T(0)), ph1 = ph;
    if (DEMOD) {
      const uint32_t m = (0u - (uint32_t)bd.shift * tb0) & (uint32_t)(n - 1);
      float s, c;
      sincospif((float)m * a.two_over_n, &s, &c);
      ph = mk<T>((T)c, (T)s);
      ph1 = mk<T>((T)bd.rot1[0], (T)bd.rot1[1]);
    }
    const int64_t orow = ((int64_t)ch * a.panel_bands + bd.out_band) * n;
    char* __restrict__ coef_row = reinterpret_cast<char*>(a.coef ? a.coef + orow : nullptr);
    char* __restrict__ bits_row = reinterpret_cast<char*>(a.bits ? a.bits + orow : nullptr);
    uint32_t tb = tb0;
    asm volatile("" : "+v"(tb));
    T rowacc = T(0), pl = T(0);
    // demodulation phasor of pair i (its even sample): ph * R^i, R = r^2 (512 samples): groups of four pairs start from
    // the exact-power seeds ph, ph R^4, ph R^8 (depth <= 4 roundings), inside a group the phasor advances by R
    const cplx<T> R1 = mk<T>((T)bd.rot[2], (T)bd.rot[3]), R4 = mk<T>((T)bd.rot[6], (T)bd.rot[7]);
    {
      // both samples of a pair leave in one 16-byte store: the even samples wait in registers for the odd ones (taking
      // one parity at a time with 8-byte stores fits 3 waves / SIMD only with spills and measured 40 % slower)
      cplx<T> ze[NOUT];
      sparse_head16<T>(v, y, om);  // even output samples
      fft4096_tail<T, 1>(v, buf, tw256, tid, col);
#pragma unroll
      for (int i = 0; i < NOUT; ++i) ze[i] = v[brev(i + C0, 4)];
      sparse_head16<T>(v, cmul(y, tw_odd), om);  // odd output samples: Y[k] W_8192^k
      fft4096_tail<T, 1>(v, buf, tw256, tid, col);
      if (pending >= 0 && tid == 0) {
        double rs = 0.0;
        for (int q = 0; q < NW; ++q) rs += s_red[par ^ 1][q];
        a.part_band[((int64_t)ch * a.panel_bands + pending) * a.nblk + blk] = rs;
      }
      auto finish_band = [&](auto guard) {  // guard: the block reaches past the end of the record (see block_bands)
      constexpr bool GUARD = decltype(guard)::value;
      cplx<T> seed = ph, rcur = ph;
#pragma unroll
      for (int i = 0; i < NOUT; ++i) {
        cplx<T> z[2] = {ze[i], v[brev(i + C0, 4)]};
        if (DEMOD) {
          if (i > 0) {
            if ((i & 3) == 0) {
              seed = cmul_rn(seed, R4);
              rcur = seed;
            } else {
              rcur = cmul_rn(rcur, R1);
            }
          }
          z[0] = cmul_rn(z[0], rcur);
          z[1] = cmul_rn(z[1], cmul_rn(rcur, ph1));
        }
        const uint32_t tt = tb + 512u * (uint32_t)i;  // even sample of the pair
        const bool inside = !GUARD || tt < (uint32_t)n;  // (n is even: the odd sample is inside with it)
        T lg[2];
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
          const T m2 = norm2(z[hh].x, z[hh].y);
          if (BITS) lg[hh] = log2_t(sqrt_t(m2) + a.eps);
          const T p = inside ? mul_rn(a.power_scale, m2) : T(0);
          col_p[2 * i + hh] += p;
          rowacc += p;
          mx = max_t(mx, p);
          pl += plog2p(p);
        }
        if (COEF && inside)
          stream_store(reinterpret_cast<float4*>(coef_row + (size_t)(tt * (uint32_t)sizeof(cplx<T>))),
                       make_float4(z[0].x, z[0].y, z[1].x, z[1].y));
        if (BITS && inside) *reinterpret_cast<float2*>(bits_row + (size_t)(tt * (uint32_t)sizeof(T))) = make_float2(lg[0], lg[1]);
      }
      };
      if (t0 + W + V > n) finish_band(std::true_type{});
      else finish_band(std::false_type{});
    }
    plogp += (double)pl;
    if (a.part_band) {
      const double rs = wave_sum((double)rowacc);
      if (lane == 0) s_red[par][wv] = rs;
      pending = bd.out_band;
      par ^= 1;
    }
  }
  T tot = T(0);
  char* __restrict__ time_row = reinterpret_cast<char*>(
      a.time_part ? a.time_part + ((int64_t)ch * a.chunk_total + a.chunk_base + plane) * n : nullptr);
#pragma unroll
  for (int i = 0; i < NOUT; ++i) {
    tot += col_p[2 * i] + col_p[2 * i + 1];
    const uint32_t tt = tb0 + 512u * (uint32_t)i;
    if (time_row && tt < (uint32_t)n)
      *reinterpret_cast<float2*>(time_row + (size_t)(tt * (uint32_t)sizeof(T))) = make_float2(col_p[2 * i], col_p[2 * i + 1]);
  }
  const double r0 = wave_max((double)mx), r1 = wave_sum((double)tot), r2 = wave_sum(plogp);
  __syncthreads();  // the last band's wave sums are visible; buf is free
  if (pending >= 0 && tid == 0) {
    double rs = 0.0;
    for (int q = 0; q < NW; ++q) rs += s_red[par ^ 1][q];
    a.part_band[((int64_t)ch * a.panel_bands + pending) * a.nblk + blk] = rs;
  }
  if (a.part_stat) {
    double* fin = reinterpret_cast<double*>(buf);
    if (lane == 0) {
      fin[wv] = r0;
      fin[NW + wv] = r1;
      fin[2 * NW + wv] = r2;
    }
    __syncthreads();
    if (tid == 0) {
      double m = 0.0, s1 = 0.0, s2 = 0.0;
      for (int q = 0; q < NW; ++q) {
        m = fin[q] > m ? fin[q] : m;
        s1 += fin[NW + q];
        s2 += fin[2 * NW + q];
      }
      double* o = a.part_stat + ((int64_t)ch * a.stat_stride + a.stat_base + stat_slot) * 3;
      o[0] = m;
      o[1] = s1;
      o[2] = s2;
    }
  }
}

template <typename T, bool DEMOD, bool COEF, bool BITS>
__device__ __forceinline__ void long_item(const BlockArgs<T>& a, const BlockItem& it, cplx<T>* __restrict__ buf,
                                          const cplx<T>* __restrict__ tw256, double (*s_red)[kBlkThreads / kWave], cplx<T> w,
                                          cplx<T> w8) {
  const int tid = threadIdx.x, lane = tid & (kWave - 1);
  const int col = (tid & ~(kWave - 1)) + (lane < 32 ? 2 * lane : 2 * (lane - 32) + 1);
  cplx<T> S[16];
  block_forward8<T, DEMOD>(a.sig + (int64_t)blockIdx.z * a.n, a.n, (int64_t)it.block * kBlkLongValid - 1024, S, buf, tw256, w, w8,
                           tid, col);
  long_bands<T, DEMOD, COEF, BITS>(a, it.block, it.band_first, it.band_count, it.plane, it.stat_slot, S, buf, tw256, s_red, w, w8);
}

// joint launch: the Stockwell bands and the styx bands of the same long block (see dual_item)
template <typename T, bool COEF, bool BITS>
__device__ __forceinline__ void long_dual_item(const BlockArgs<T>& a0, const BlockArgs<T>& a2, const DualItem& it,
                                               cplx<T>* __restrict__ buf, const cplx<T>* __restrict__ tw256,
                                               double (*s_red)[kBlkThreads / kWave], cplx<T> w, cplx<T> w8) {
  const int tid = threadIdx.x, lane = tid & (kWave - 1);
  const int col = (tid & ~(kWave - 1)) + (lane < 32 ? 2 * lane : 2 * (lane - 32) + 1);
  const int64_t n = a0.n, t0 = (int64_t)it.block * kBlkLongValid - 1024;
  const bool inside = t0 >= 0 && t0 + kBlkLong <= n;
  const T* sig = a0.sig + (int64_t)blockIdx.z * n;
  cplx<T> S[16];
  if (it.count2 > 0 || inside) block_forward8<T, true>(sig, n, t0, S, buf, tw256, w, w8, tid, col);
  if (it.count2 > 0)
    long_bands<T, true, COEF, BITS>(a2, it.block, it.first2, it.count2, it.plane2, it.slot2, S, buf, tw256, s_red, w, w8);
  if (it.count0 > 0) {
    if (!inside) block_forward8<T, false>(sig, n, t0, S, buf, tw256, w, w8, tid, col);
    long_bands<T, false, COEF, BITS>(a0, it.block, it.first0, it.count0, it.plane0, it.slot0, S, buf, tw256, s_red, w, w8);
  }
}

// qi_cwt_stx: the Stockwell bands (a2) and the styx bands (a0) of the same block from ONE forward transform -- inside
// the record its wrapped and its zero-extended loads are the same samples; the blocks that reach over a record end are
// transformed a second time.
template <typename T, int WQ, bool COEF, bool BITS>
__device__ __forceinline__ void dual_item(const BlockArgs<T>& a0, const BlockArgs<T>& a2, const DualItem& it,
                                          cplx<T>* __restrict__ buf, const cplx<T>* __restrict__ tw256,
                                          double (*s_red)[kBlkThreads / kWave], cplx<T> w) {
  const int tid = threadIdx.x, lane = tid & (kWave - 1);
  const int col = (tid & ~(kWave - 1)) + (lane < 32 ? 2 * lane : 2 * (lane - 32) + 1);
  const int64_t n = a0.n, t0 = (int64_t)it.block * (kBlk - 512 * WQ) - 256 * WQ;
  const bool inside = t0 >= 0 && t0 + kBlk <= n;
  const T* sig = a0.sig + (int64_t)blockIdx.z * n;
  cplx<T> S[16];
  if (it.count2 > 0 || inside) block_forward<T, true>(sig, n, t0, S, buf, tw256, w, tid, col);
  if (it.count2 > 0)
    block_bands<T, WQ, true, COEF, BITS>(a2, it.block, it.first2, it.count2, it.plane2, it.slot2, S, buf, tw256, s_red, w);
  if (it.count0 > 0) {
    if (!inside) block_forward<T, false>(sig, n, t0, S, buf, tw256, w, tid, col);
    block_bands<T, WQ, false, COEF, BITS>(a0, it.block, it.first0, it.count0, it.plane0, it.slot0, S, buf, tw256, s_red, w);
  }
}

// Edge item of a split band (see the note on split bands in qi_native.hpp) for output block it.block: the two
// pieces read the record n/2 - W samples after / before the output block (zero outside the record; a piece whose 4096
// input samples all lie outside is skipped), their filtered spectra are summed and transformed back once; the zoom
// engine's part of the band is added and the band is finished like any block band (panel rows, reductions).
// `follows`: the item's plane and statistics slot already hold the bands before it of a merged item (k_block_edge's blocks
// with two pieces): added to instead of written.
template <typename T, int WQ, bool COEF, bool BITS>
__device__ __forceinline__ void edge_item(const BlockArgs<T>& a, const BlockItem& it, cplx<T>* __restrict__ buf,
                                          const cplx<T>* __restrict__ tw256, cplx<T> w, bool follows = false) {
  constexpr int W = 256 * WQ, V = kBlk - 2 * W, NOUT = 16 - 2 * WQ, NW = kBlkThreads / kWave;
  const int tid = threadIdx.x, lane = tid & (kWave - 1), wv = tid / kWave;
  const int col = kWave * wv + (lane < 32 ? 2 * lane : 2 * (lane - 32) + 1);  // fft4096's column order
  const int64_t n = a.n, ch = blockIdx.z, blk = it.block;
  const int sb = it.band_first, out_band = a.edge_band[sb];
  const int64_t t0 = blk * V - W;  // outputs [t0 + W, t0 + W + V)
  const T* __restrict__ sig = a.sig + ch * n;
  const cplx<T>* __restrict__ bank = a.edge_bank + (int64_t)sb * 2 * kBlk;
  // the zoom engine's part of this thread's outputs, requested first
  const cplx<T>* __restrict__ part = a.edge_part + (ch * a.nsplit + sb) * n;
  const uint32_t tb = (uint32_t)(t0 + W + col);
  cplx<T> z0[NOUT];
#pragma unroll
  for (int i = 0; i < NOUT; ++i) {
    const uint32_t t = tb + 256u * (uint32_t)i;
    z0[i] = t < (uint32_t)n ? part[t] : mk<T>(T(0), T(0));
  }
  cplx<T> acc[16];
#pragma unroll
  for (int b = 0; b < 16; ++b) acc[b] = mk<T>(T(0), T(0));
  for (int piece = 0; piece < 2; ++piece) {
    const int64_t in0 = t0 + (piece == 0 ? n / 2 - W : -(n / 2 - W));
    if (in0 >= n || in0 + kBlk <= 0) continue;  // the same for every thread of the workgroup
    cplx<T> S[16];
#pragma unroll
    for (int b = 0; b < 16; ++b) {
      const int64_t t = in0 + col + 256 * b;
      S[b] = mk<T>((t >= 0 && t < n) ? sig[t] : T(0), T(0));
    }
    fft4096<T, -1>(S, buf, tw256, w, tid, col);
    const cplx<T>* __restrict__ H = bank + (int64_t)piece * kBlk + col;
#pragma unroll
    for (int b = 0; b < 16; ++b) {
      const cplx<T> y = cmul(S[brev(b, 4)], H[256 * b]);
      acc[b].x += y.x;
      acc[b].y += y.y;
    }
  }
  fft4096<T, 1>(acc, buf, tw256, w, tid, col);
  const int64_t orow = ((int64_t)ch * a.panel_bands + out_band) * n;
  cplx<T>* __restrict__ coef_row = a.coef ? a.coef + orow : nullptr;
  T* __restrict__ bits_row = a.bits ? a.bits + orow : nullptr;
  T* __restrict__ time_row = a.time_part ? a.time_part + ((int64_t)ch * a.chunk_total + a.chunk_base + it.plane) * n : nullptr;
  T rowacc = T(0), pl = T(0), mx = T(0);
#pragma unroll
  for (int i = 0; i < NOUT; ++i) {
    const uint32_t t = tb + 256u * (uint32_t)i;
    const bool inside = t < (uint32_t)n;
    cplx<T> z = acc[brev(i + WQ, 4)];
    z.x += z0[i].x;
    z.y += z0[i].y;
    const T m2 = norm2(z.x, z.y);
    if (COEF && inside) stream_store(coef_row + t, z);
    if (BITS && inside) bits_row[t] = log2_t(sqrt_t(m2) + a.eps);
    const T p = inside ? mul_rn(a.power_scale, m2) : T(0);
    if (time_row && inside) time_row[t] = follows ? time_row[t] + p : p;
    rowacc += p;
    mx = max_t(mx, p);
    if constexpr (sizeof(T) == 8) pl += plog2p_flat(p, reinterpret_cast<const double (*)[2]>(tw256 + 256));
    else pl += plog2p(p);
  }
  const double r0 = wave_max((double)mx), r1 = wave_sum((double)rowacc), r2 = wave_sum((double)pl);
  __syncthreads();  // buf is free
  double* fin = reinterpret_cast<double*>(buf);
  if (lane == 0) {
    fin[wv] = r0;
    fin[NW + wv] = r1;
    fin[2 * NW + wv] = r2;
  }
  __syncthreads();
  if (tid == 0) {
    double m = 0.0, s1 = 0.0, s2 = 0.0;
    for (int q = 0; q < NW; ++q) {
      m = fin[q] > m ? fin[q] : m;
      s1 += fin[NW + q];
      s2 += fin[2 * NW + q];
    }
    if (a.part_band) a.part_band[((int64_t)ch * a.panel_bands + out_band) * a.nblk + blk] = s1;
    if (a.part_stat) {
      double* o = a.part_stat + ((int64_t)ch * a.stat_stride + a.stat_base + it.stat_slot) * 3;
      o[0] = follows ? (o[0] > m ? o[0] : m) : m;
      o[1] = follows ? o[1] + s1 : s1;
      o[2] = follows ? o[2] + s2 : s2;
    }
  }
}

// The same for ALL split bands of an output block (it.band_first ... + it.band_count): the block's far piece is read and
// transformed once and the bands run through block_bands (EDGE) like the bands of any item -- one filter multiply, inverse
// transform and pair epilogue each, one per-time plane and one statistics slot for all of them.  The two blocks around the
// middle of the record, which see both pieces, go band by band through edge_item.  A kernel of its own (k_block_edge):
// inside k_block / k_block_dual the band loop costs every variant its register budget (0 -> 236-396 bytes of scratch per
// lane, measured).
// PATH: 0 both kinds of block in one kernel (float32); float64 compiles them apart -- the two-piece blocks' band-by-band
// path (1) costs the one-piece path (2) its register allocation when they share a kernel (500-600 bytes of scratch per lane
// in round 3): a block of the other kind is left to the other kernel's launch.
template <typename T, int WQ, bool COEF, bool BITS, int PATH = 0>
__device__ __forceinline__ void edge_block_item(const BlockArgs<T>& a, const BlockItem& it, cplx<T>* __restrict__ buf,
                                                const cplx<T>* __restrict__ tw256, double (*s_red)[kBlkThreads / kWave],
                                                cplx<T> w) {
  constexpr int W = 256 * WQ, V = kBlk - 2 * W;
  const int tid = threadIdx.x, lane = tid & (kWave - 1), wv = tid / kWave;
  const int col = kWave * wv + (lane < 32 ? 2 * lane : 2 * (lane - 32) + 1);  // fft4096's column order
  const int64_t n = a.n, t0 = (int64_t)it.block * V - W;
  const int64_t in_p[2] = {t0 + (n / 2 - W), t0 - (n / 2 - W)};
  const bool has_p[2] = {!(in_p[0] >= n || in_p[0] + kBlk <= 0), !(in_p[1] >= n || in_p[1] + kBlk <= 0)};
  if (has_p[0] && has_p[1]) {  // (the same for every thread of the workgroup)
    if constexpr (PATH != 2) {
      for (int32_t q = 0; q < it.band_count; ++q) {
        const BlockItem one{it.wq, it.block, it.band_first + q, 0, it.plane, it.stat_slot};
        if (q > 0) __syncthreads();  // the previous band's statistics have left the exchange buffer
        edge_item<T, WQ, COEF, BITS>(a, one, buf, tw256, w, q > 0);
      }
    }
    return;
  }
  if constexpr (PATH == 1) return;
  const int piece = has_p[0] ? 0 : 1;
  const T* __restrict__ sig = a.sig + (int64_t)blockIdx.z * n;
  cplx<T> S[16];
#pragma unroll
  for (int b = 0; b < 16; ++b) {
    const int64_t t = in_p[piece] + col + 256 * b;
    S[b] = mk<T>((t >= 0 && t < n) ? sig[t] : T(0), T(0));
  }
  fft4096<T, -1>(S, buf, tw256, w, tid, col);
  cplx<T> Sn[16];  // block_bands' order of the block spectrum
#pragma unroll
  for (int b = 0; b < 16; ++b) Sn[b] = S[brev(b, 4)];
  block_bands<T, WQ, false, COEF, BITS, true>(a, it.block, it.band_first, it.band_count, it.plane, it.stat_slot, Sn, buf, tw256,
                                              s_red, w, piece);
}

#ifndef QI_BLK_EDGE_WAVES
#define QI_BLK_EDGE_WAVES QI_BLK_WAVES
#endif
// merged edge items (plan tables for many records): items of the band launch's list (`items`) or of the joint list (`dual`)
template <typename T, bool COEF, bool BITS>
__global__ void __launch_bounds__(kBlkThreads, QI_BLK_EDGE_WAVES) k_block_edge(BlockArgs<T> a, const BlockItem* __restrict__ items,
                                                                              const DualItem* __restrict__ dual) {
  __shared__ cplx<T> buf[kBlkBuf];
  __shared__ cplx<T> tw256[256];
  __shared__ double s_red[2][kBlkThreads / kWave];
  const int tid = threadIdx.x, lane = tid & (kWave - 1);
  const int col = (tid & ~(kWave - 1)) + (lane < 32 ? 2 * lane : 2 * (lane - 32) + 1);
  float sn, cs;
  sincospif((float)tid * (2.0f / 256.0f), &sn, &cs);
  tw256[tid] = mk<T>((T)cs, (T)sn);
  sincospif((float)col * (2.0f / 4096.0f), &sn, &cs);
  const cplx<T> w = mk<T>((T)cs, (T)sn);
  BlockItem it;
  if (items) {
    it = load_uniform(items + blockIdx.x);
  } else {
    const DualItem d = load_uniform(dual + blockIdx.x);
    it = BlockItem{d.wq, d.block, d.first0, d.count0, d.plane0, d.slot0};
  }
  switch (-it.wq) {
    case 1: edge_block_item<T, 1, COEF, BITS>(a, it, buf, tw256, s_red, w); break;
    case 2: edge_block_item<T, 2, COEF, BITS>(a, it, buf, tw256, s_red, w); break;
    default: edge_block_item<T, 4, COEF, BITS>(a, it, buf, tw256, s_red, w); break;
  }
}

template <typename T, bool DEMOD, bool COEF, bool BITS>
__global__ void __launch_bounds__(kBlkThreads, QI_BLK_WAVES) k_block(BlockArgs<T> a) {
  __shared__ cplx<T> buf[kBlkBuf];
  __shared__ cplx<T> tw256[256];
  __shared__ double s_red[2][kBlkThreads / kWave];
  const int tid = threadIdx.x;
  {
    float s, c;
    sincospif((float)tid * (2.0f / 256.0f), &s, &c);
    tw256[tid] = mk<T>((T)c, (T)s);
  }
  cplx<T> w;
  {
    float s, c;
    const int lane = tid & (kWave - 1);
    const int col = (tid & ~(kWave - 1)) + (lane < 32 ? 2 * lane : 2 * (lane - 32) + 1);  // block_item's column order
    sincospif((float)col * (2.0f / 4096.0f), &s, &c);
    w = mk<T>((T)c, (T)s);
  }
  const BlockItem it = load_uniform(a.items + blockIdx.x);
  if (it.wq < 0) {
    // edge item of a split band (styx bank): light items at the end of the list, they fill the tail of the launch
    if constexpr (!DEMOD) {
      switch (-it.wq) {
        case 1: edge_item<T, 1, COEF, BITS>(a, it, buf, tw256, w); break;
        case 2: edge_item<T, 2, COEF, BITS>(a, it, buf, tw256, w); break;
        default: edge_item<T, 4, COEF, BITS>(a, it, buf, tw256, w); break;
      }
    }
    return;
  }
  switch (it.wq) {
    case 1: block_item<T, 1, DEMOD, COEF, BITS>(a, it, buf, tw256, s_red, w); break;
    case 2: block_item<T, 2, DEMOD, COEF, BITS>(a, it, buf, tw256, s_red, w); break;
    default: block_item<T, 4, DEMOD, COEF, BITS>(a, it, buf, tw256, s_red, w); break;
  }
}

// float64 records (run_native64): the same band items in double arithmetic -- Gaussian filter spectra in registers, no
// narrow-spectrum shortcuts (those drop weights below 2^-30 of the peak), no long blocks; split bands through k_block64_edge.  The exchange
// buffer and the twiddle table take 72 KB of (dynamic) LDS: two workgroups per CU, compiled for two waves per SIMD.
constexpr size_t kBlk64Lds = (size_t)(kBlkBuf + 256 + 128) * sizeof(double2);  // exchange buffer | W256 table | log2 table
template <bool DEMOD, bool COEF, bool BITS>
__global__ void __launch_bounds__(kBlkThreads, 2) k_block64(BlockArgs<double> a) {
  extern __shared__ __attribute__((aligned(16))) char smem64[];
  double2* buf = reinterpret_cast<double2*>(smem64);
  double2* tw256 = buf + kBlkBuf;
  __shared__ double s_red[2][kBlkThreads / kWave];
  const int tid = threadIdx.x, lane = tid & (kWave - 1);
  const int col = (tid & ~(kWave - 1)) + (lane < 32 ? 2 * lane : 2 * (lane - 32) + 1);  // block_item's column order
  {
    double s, c;
    sincospi((double)tid * (2.0 / 256.0), &s, &c);
    tw256[tid] = make_double2(c, s);
    if (tid < 128) tw256[256 + tid] = make_double2(kLog2Tab[tid][0], kLog2Tab[tid][1]);  // (block_bands' entropy logarithm)
  }
  double2 w;
  {
    double s, c;
    sincospi((double)col * (2.0 / 4096.0), &s, &c);
    w = make_double2(c, s);
  }
  const BlockItem it = load_uniform(a.items + blockIdx.x);
  switch (it.wq) {
    case 1: block_item<double, 1, DEMOD, COEF, BITS>(a, it, buf, tw256, s_red, w); break;
    case 2: block_item<double, 2, DEMOD, COEF, BITS>(a, it, buf, tw256, s_red, w); break;
    default: block_item<double, 4, DEMOD, COEF, BITS>(a, it, buf, tw256, s_red, w); break;
  }
}

// the split bands of a float64 styx table: one edge item per block (edge_block_item), always a launch of its own (PATH 2:
// the blocks that see one piece of the record; PATH 1: a second, small launch for the blocks around the middle that see both)
template <bool COEF, bool BITS, int PATH>
__global__ void __launch_bounds__(kBlkThreads, 2) k_block64_edge(BlockArgs<double> a, const BlockItem* __restrict__ items) {
  extern __shared__ __attribute__((aligned(16))) char smem64[];
  double2* buf = reinterpret_cast<double2*>(smem64);
  double2* tw256 = buf + kBlkBuf;
  __shared__ double s_red[2][kBlkThreads / kWave];
  const int tid = threadIdx.x, lane = tid & (kWave - 1);
  const int col = (tid & ~(kWave - 1)) + (lane < 32 ? 2 * lane : 2 * (lane - 32) + 1);
  double s, c;
  sincospi((double)tid * (2.0 / 256.0), &s, &c);
  tw256[tid] = make_double2(c, s);
  if (tid < 128) tw256[256 + tid] = make_double2(kLog2Tab[tid][0], kLog2Tab[tid][1]);
  sincospi((double)col * (2.0 / 4096.0), &s, &c);
  const double2 w = make_double2(c, s);
  const BlockItem it = load_uniform(items + blockIdx.x);
  switch (-it.wq) {
    case 1: edge_block_item<double, 1, COEF, BITS, PATH>(a, it, buf, tw256, s_red, w); break;
    case 2: edge_block_item<double, 2, COEF, BITS, PATH>(a, it, buf, tw256, s_red, w); break;
    default: edge_block_item<double, 4, COEF, BITS, PATH>(a, it, buf, tw256, s_red, w); break;
  }
}

// ---- float64 zoom, coarse stage in LDS (round 4) ---------------------------------------------------------------------
// The coarse samples of a band -- b[tau] = sum_q Yb[q] exp(2 pi i q tau / M), M = 4096 P, Yb the band's occupied bins at
// baseband -- for one residue tau1 = tau mod P are ONE 4096-point transform of the bins folded modulo 4096 and twiddled
// (zoom_gather16 of the float32 engine, here in double): a workgroup forms the 4096 inputs of its plane in registers from
// the record's spectrum and the compact bank (or the Stockwell window), transforms them in LDS and stores the samples
// tau = P tau2 + tau1 into the band's coarse array [kZ64Pad | M | kZ64Pad] (natural order: what k_z64_fine reads), pads
// included.  Against gather launch + batched hipFFT (3-5 passes over zero-filled M-point arrays) + pad launch this moves
// the coarse array through HBM once.  The fold costs ~ len / 4096 terms per input: it pays on the three coarsest grids
// (P <= 32 at Lf = 2^21); the two finest (len up to 131 072 bins: 33 terms, more arithmetic than a full transform) stay
// with hipFFT.
template <bool STX>
__device__ __forceinline__ void z64_gather16(const Z64Args& a, const BandDesc& bd, const uint32_t tau1, const int col,
                                             const double2* __restrict__ X, double2 (&v)[16]) {
  const int32_t M = (int32_t)a.M, P = M / kBlk;
  const int32_t kc = STX ? 0 : bd.k_lo + bd.k_len / 2;
  const int32_t ks_lo = bd.k_lo - kc, ks_hi = ks_lo + bd.k_len;  // support in baseband bins
  const uint32_t lmask = (uint32_t)a.Lf - 1u;
  const int nterm = (bd.k_len + kBlk - 1) / kBlk;
  int32_t ks0[16];
#pragma unroll
  for (int b = 0; b < 16; ++b) {
    v[b] = make_double2(0.0, 0.0);
    ks0[b] = ks_lo + (((col + 256 * b) - ks_lo) & (kBlk - 1));
  }
  // block twiddles exp(2 pi i r tau1 / P): r = r_a for the elements whose first bin lies in the block of ks_lo, r_a + 1 for
  // the others, advanced by one block per term
  const uint32_t r_a = ((uint32_t)ks_lo & ((uint32_t)M - 1u)) / kBlk;
  const uint32_t edge = (((uint32_t)ks_lo & ((uint32_t)M - 1u)) | (uint32_t)(kBlk - 1)) + 1u;
  double sf, cf;
  sincospi(2.0 * (double)((r_a * tau1) & (uint32_t)(P - 1)) / (double)P, &sf, &cf);
  double2 w_a = make_double2(cf, sf);
  sincospi(2.0 * (double)(tau1 & (uint32_t)(P - 1)) / (double)P, &sf, &cf);
  const double2 w_step = make_double2(cf, sf);
  double2 w_b = cmul(w_a, w_step);
  const uint32_t base_mod = (uint32_t)ks_lo & ((uint32_t)M - 1u);
  for (int m = 0; m < nterm; ++m) {
#pragma unroll
    for (int hb = 0; hb < 16; hb += 8) {  // eight elements' loads in flight at a time
      double2 x[8], h[8];
      bool on[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int32_t ks = ks0[hb + q] + kBlk * m;
        on[q] = ks < ks_hi;
        const int32_t k = kc + ks;
        x[q] = make_double2(0.0, 0.0);
        h[q] = x[q];
        if (on[q]) {
          if (STX) {
            x[q] = X[(uint32_t)(k + (int32_t)bd.shift) & lmask];
          } else {
            x[q] = X[(uint32_t)k & lmask];
            h[q] = a.Hc[bd.src_off + (k - bd.k_lo)];
          }
        }
      }
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int b = hb + q;
        if (!on[q]) continue;
        double2 y;
        if (STX) {
          const double g0 = bd.coef * (double)(kc + ks0[b] + kBlk * m);
          const double g = exp2_t(-g0 * g0) * a.inv_len;
          y = make_double2(x[q].x * g, x[q].y * g);
        } else {
          y = cmul(x[q], h[q]);
        }
        const bool second = base_mod + (uint32_t)(ks0[b] - ks_lo) >= edge;
        const double2 t = cmul(y, second ? w_b : w_a);
        v[b].x += t.x;
        v[b].y += t.y;
      }
    }
    w_a = w_b;
    w_b = cmul(w_b, w_step);
  }
  // outer twiddle exp(2 pi i (col + 256 b) tau1 / M) = e0 s^b (exact integer phases), powers by binary products
  sincospi(2.0 * (double)(((uint32_t)col * tau1) & ((uint32_t)M - 1u)) / (double)M, &sf, &cf);
  const double2 e0 = make_double2(cf, sf);
  sincospi(2.0 * (double)((256u * tau1) & ((uint32_t)M - 1u)) / (double)M, &sf, &cf);
  const double2 s1 = make_double2(cf, sf), s2 = cmul(s1, s1), s4 = cmul(s2, s2), s8 = cmul(s4, s4);
  double2 pw = e0;
#pragma unroll
  for (int b = 0; b < 16; ++b) {  // (Gray-code-free walk: pw(b) from pw(b - lowbit(b)) needs sixteen live values; b -> b + 1 by s1,
    v[b] = cmul(v[b], pw);        //  fifteen roundings deep, is far inside the float64 tolerance and holds one)
    pw = cmul(pw, s1);
  }
  (void)s2; (void)s4; (void)s8;
}

template <bool STX>
__global__ void __launch_bounds__(kBlkThreads, 2) k_z64_coarse(Z64Args a) {
  extern __shared__ __attribute__((aligned(16))) char smem64[];
  double2* buf = reinterpret_cast<double2*>(smem64);
  double2* tw256 = buf + kBlkBuf;
  const int tid = threadIdx.x, lane = tid & (kWave - 1);
  const int col = (tid & ~(kWave - 1)) + (lane < 32 ? 2 * lane : 2 * (lane - 32) + 1);  // fft4096's column order
  const int32_t P = (int32_t)(a.M / kBlk);
  // Consecutive workgroups go to consecutive XCDs; the P planes of a band fill the same 128-byte lines of its coarse array
  // (sample tau = P tau2 + tau1, 16 bytes each) and gather the same bins: consecutive planes sit behind ONE L2 -- workgroup
  // w takes plane (w mod 8) ceil(planes / 8) + w / 8 of the launch (the grid is rounded up to a multiple of 8)
  const uint32_t planes = (uint32_t)a.nbands * (uint32_t)P;
  const uint32_t pi = (blockIdx.x & 7u) * ((planes + 7u) >> 3) + (blockIdx.x >> 3);
  if (pi >= planes) return;  // (the same for the whole workgroup)
  const uint32_t band = pi / (uint32_t)P, tau1 = pi - band * (uint32_t)P;
  const BandDesc bd = a.bands[band];
  const int64_t ch = blockIdx.y;
  double2 v[16];
  z64_gather16<STX>(a, bd, tau1, col, a.X + ch * a.Lf, v);
  {
    double s, c;
    sincospi((double)tid * (2.0 / 256.0), &s, &c);
    tw256[tid] = make_double2(c, s);
  }
  double2 w;
  {
    double s, c;
    sincospi((double)col * (2.0 / 4096.0), &s, &c);
    w = make_double2(c, s);
  }
  fft4096<double, 1>(v, buf, tw256, w, tid, col);
  double2* __restrict__ z = a.Z + ((int64_t)ch * a.nbands + band) * (a.M + 2 * kZ64Pad) + kZ64Pad;
#pragma unroll
  for (int c = 0; c < 16; ++c) {
    const int64_t tau = (int64_t)P * (col + 256 * c) + tau1;
    const double2 val = v[brev(c, 4)];
    z[tau] = val;
    if (tau < kZ64Pad) z[a.M + tau] = val;            // back pad <- first samples
    if (tau >= a.M - kZ64Pad) z[tau - a.M] = val;     // front pad <- last samples
  }
}

// long-block items (BlockItem::wq = kBlkLongWq) have kernels of their own: their register budget (the even samples of a
// band are held while its odd samples are transformed) would spill inside k_block / k_block_dual
template <typename T, bool DEMOD, bool COEF, bool BITS>
__global__ void __launch_bounds__(kBlkThreads, QI_BLK_LONG_WAVES) k_block_long(BlockArgs<T> a, const BlockItem* __restrict__ items) {
  __shared__ cplx<T> buf[kBlkBuf];
  __shared__ cplx<T> tw256[256];
  __shared__ double s_red[2][kBlkThreads / kWave];
  const int tid = threadIdx.x, lane = tid & (kWave - 1);
  const int col = (tid & ~(kWave - 1)) + (lane < 32 ? 2 * lane : 2 * (lane - 32) + 1);
  float s, c;
  sincospif((float)tid * (2.0f / 256.0f), &s, &c);
  tw256[tid] = mk<T>((T)c, (T)s);
  sincospif((float)col * (2.0f / 4096.0f), &s, &c);
  const cplx<T> w = mk<T>((T)c, (T)s);
  sincospif((float)col * (2.0f / 8192.0f), &s, &c);
  const cplx<T> w8 = mk<T>((T)c, (T)s);
  long_item<T, DEMOD, COEF, BITS>(a, load_uniform(items + blockIdx.x), buf, tw256, s_red, w, w8);
}
template <typename T, bool COEF, bool BITS>
__global__ void __launch_bounds__(kBlkThreads, QI_BLK_LONG_WAVES) k_block_long_dual(BlockArgs<T> a0, BlockArgs<T> a2,
                                                                                    const DualItem* __restrict__ items) {
  __shared__ cplx<T> buf[kBlkBuf];
  __shared__ cplx<T> tw256[256];
  __shared__ double s_red[2][kBlkThreads / kWave];
  const int tid = threadIdx.x, lane = tid & (kWave - 1);
  const int col = (tid & ~(kWave - 1)) + (lane < 32 ? 2 * lane : 2 * (lane - 32) + 1);
  float s, c;
  sincospif((float)tid * (2.0f / 256.0f), &s, &c);
  tw256[tid] = mk<T>((T)c, (T)s);
  sincospif((float)col * (2.0f / 4096.0f), &s, &c);
  const cplx<T> w = mk<T>((T)c, (T)s);
  sincospif((float)col * (2.0f / 8192.0f), &s, &c);
  const cplx<T> w8 = mk<T>((T)c, (T)s);
  long_dual_item<T, COEF, BITS>(a0, a2, load_uniform(items + blockIdx.x), buf, tw256, s_red, w, w8);
}

template <typename T, bool COEF, bool BITS>
__global__ void __launch_bounds__(kBlkThreads, QI_BLK_WAVES) k_block_dual(BlockArgs<T> a0, BlockArgs<T> a2,
                                                                         const DualItem* __restrict__ items) {
  __shared__ cplx<T> buf[kBlkBuf];
  __shared__ cplx<T> tw256[256];
  __shared__ double s_red[2][kBlkThreads / kWave];
  const int tid = threadIdx.x;
  {
    float s, c;
    sincospif((float)tid * (2.0f / 256.0f), &s, &c);
    tw256[tid] = mk<T>((T)c, (T)s);
  }
  cplx<T> w;
  {
    float s, c;
    const int lane = tid & (kWave - 1);
    const int col = (tid & ~(kWave - 1)) + (lane < 32 ? 2 * lane : 2 * (lane - 32) + 1);  // block_bands' column order
    sincospif((float)col * (2.0f / 4096.0f), &s, &c);
    w = mk<T>((T)c, (T)s);
  }
  const DualItem it = load_uniform(items + blockIdx.x);
#ifdef QI_NATIVE_STAMPS
  const unsigned long long wall0 = __builtin_amdgcn_s_memrealtime();  // (100 MHz) the launch's dispatch timeline
#endif
  if (it.wq < 0) {
    const BlockItem e{it.wq, it.block, it.first0, it.count0, it.plane0, it.slot0};
    switch (-it.wq) {
      case 1: edge_item<T, 1, COEF, BITS>(a0, e, buf, tw256, w); break;
      case 2: edge_item<T, 2, COEF, BITS>(a0, e, buf, tw256, w); break;
      default: edge_item<T, 4, COEF, BITS>(a0, e, buf, tw256, w); break;
    }
  } else {
    switch (it.wq) {
      case 1: dual_item<T, 1, COEF, BITS>(a0, a2, it, buf, tw256, s_red, w); break;
      case 2: dual_item<T, 2, COEF, BITS>(a0, a2, it, buf, tw256, s_red, w); break;
      default: dual_item<T, 4, COEF, BITS>(a0, a2, it, buf, tw256, s_red, w); break;
    }
  }
#ifdef QI_NATIVE_STAMPS
  if (a0.stamps && tid == 0) {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    unsigned long long* o = a0.stamps + ((int64_t)blockIdx.z * gridDim.x + blockIdx.x) * 8;
    unsigned hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    // (48 bits of time; above them the XCD and the CU's place in it, the item's reach code and band count)
    o[6] = (wall0 & 0xffffffffffffull) | ((unsigned long long)(xcc & 15) << 56) | ((unsigned long long)((hw >> 8) & 0xff) << 48);
    o[7] = (__builtin_amdgcn_s_memrealtime() & 0xffffffffffffull) | ((unsigned long long)(it.wq & 0xff) << 56) |
           ((unsigned long long)((it.count0 + it.count2) & 0xff) << 48);
  }
#endif
}

// Coarse stage of the zoom engine (qi_zoom.hip): workgroup (tau1, band, record) transforms the 4096 folded and
// twiddled baseband bins that k_zoom_gather left in plane tau1 of the band, in place: afterwards
// plane tau1 of the band holds the envelope samples tau = P tau2 + tau1 of its coarse grid at [tau2].
template <typename T>
__device__ __forceinline__ void zoom_coarse_plane(cplx<T>* __restrict__ plane0) {
  __shared__ cplx<T> buf[kBlkBuf];
  __shared__ cplx<T> tw256[256];
  const int tid = threadIdx.x, lane = tid & (kWave - 1);
  const int col = (tid & ~(kWave - 1)) + (lane < 32 ? 2 * lane : 2 * (lane - 32) + 1);  // fft4096's column order
  cplx<T>* __restrict__ plane = plane0 + col;
  cplx<T> v[16];
#pragma unroll
  for (int b = 0; b < 16; ++b) v[b] = plane[256 * b];
  {
    float s, c;
    sincospif((float)tid * (2.0f / 256.0f), &s, &c);
    tw256[tid] = mk<T>((T)c, (T)s);
  }
  cplx<T> w;
  {
    float s, c;
    sincospif((float)col * (2.0f / 4096.0f), &s, &c);
    w = mk<T>((T)c, (T)s);
  }
  fft4096<T, 1>(v, buf, tw256, w, tid, col);
#pragma unroll
  for (int c = 0; c < 16; ++c) plane[256 * c] = v[brev(c, 4)];
}
// The same with the gather step inside: the plane's 4096 inputs are formed in registers (zoom_gather_value) instead of
// being written by a gather launch and read back -- one launch and two passes over the coarse storage fewer.
template <typename T, bool STX, int NF>
__device__ __forceinline__ void zoom_coarse_plane_gather(const ZoomArgs<T>& a, const uint32_t plane_i,
                                                         cplx<T>* __restrict__ buf, cplx<T>* __restrict__ tw256) {
  const int tid = threadIdx.x, lane = tid & (kWave - 1);
  const int col = (tid & ~(kWave - 1)) + (lane < 32 ? 2 * lane : 2 * (lane - 32) + 1);  // fft4096's column order
  const BandDesc bd = load_uniform(a.bands + *as_const(a.plane_band + plane_i));
  const uint32_t tau1 = plane_i - (uint32_t)bd.edge;
  const int64_t ch = blockIdx.z;
  const cplx<T>* __restrict__ X = a.X + ch * (a.Lf << a.x_shift);
  cplx<T> v[16];
#ifdef QI_NATIVE_DEBUG
  if (a.debug & 256) {
#pragma unroll
    for (int b = 0; b < 16; ++b) v[b] = mk<T>((T)(col + b), (T)tau1);
  } else
#endif
  zoom_gather16<T, STX, NF>(a, bd, tau1, col, X, v);
  {
    float s, c;
    sincospif((float)tid * (2.0f / 256.0f), &s, &c);
    tw256[tid] = mk<T>((T)c, (T)s);
  }
  cplx<T> w;
  {
    float s, c;
    sincospif((float)col * (2.0f / 4096.0f), &s, &c);
    w = mk<T>((T)c, (T)s);
  }
#ifdef QI_NATIVE_DEBUG
  if (!(a.debug & 512))
#endif
  fft4096<T, 1>(v, buf, tw256, w, tid, col);
  cplx<T>* __restrict__ plane = a.coarse + ((int64_t)ch * a.planes + plane_i) * kBlk + col;
#ifdef QI_NATIVE_DEBUG
  if ((a.debug & 1024) && v[3].x != (T)12345.678f) return;
#endif
#pragma unroll
  for (int c = 0; c < 16; ++c) plane[256 * c] = v[brev(c, 4)];
}
template <typename T, int NF>
__global__ void __launch_bounds__(kBlkThreads) k_zoom_coarse2g(ZoomArgs<T> a0, ZoomArgs<T> a2) {
  __shared__ cplx<T> buf[kBlkBuf];
  __shared__ cplx<T> tw256[256];
  // Consecutive workgroups go to consecutive XCDs; the planes of a band (>= 4 or 8 consecutive plane indices) all gather the
  // same bins of the spectrum and the same filter row: runs of eight consecutive planes stay behind one L2, the runs take
  // turns over the XCDs (the grid is rounded up to a multiple of 64).
#ifdef QI_COARSE_PLAIN_ORDER
  const uint32_t pi = blockIdx.x;
#else
  const uint32_t q = blockIdx.x >> 3, pi = (q >> 3) * 64u + (blockIdx.x & 7u) * 8u + (q & 7u);
#endif
  if (pi >= (uint32_t)(a0.planes + a2.planes)) return;
  if (pi < (uint32_t)a0.planes) zoom_coarse_plane_gather<T, false, NF>(a0, pi, buf, tw256);
  else zoom_coarse_plane_gather<T, true, NF>(a2, pi - (uint32_t)a0.planes, buf, tw256);
}
template <typename T, bool STX>
__global__ void __launch_bounds__(kBlkThreads) k_zoom_coarse_g(ZoomArgs<T> a) {
  __shared__ cplx<T> buf[kBlkBuf];
  __shared__ cplx<T> tw256[256];
  zoom_coarse_plane_gather<T, STX, 8>(a, blockIdx.x, buf, tw256);
}

template <typename T>
__global__ void __launch_bounds__(kBlkThreads) k_zoom_coarse(ZoomArgs<T> a) {
  zoom_coarse_plane<T>(a.coarse + ((int64_t)blockIdx.z * a.planes + blockIdx.x) * kBlk);
}
// qi_cwt_stx: the planes of both tables in one launch
template <typename T>
__global__ void __launch_bounds__(kBlkThreads) k_zoom_coarse2(ZoomArgs<T> a0, ZoomArgs<T> a2) {
  const bool first = blockIdx.x < (uint32_t)a0.planes;
  const ZoomArgs<T>& a = first ? a0 : a2;
  const int64_t plane = first ? blockIdx.x : blockIdx.x - (uint32_t)a0.planes;
  zoom_coarse_plane<T>(a.coarse + ((int64_t)blockIdx.z * a.planes + plane) * kBlk);
}

// taps of a Gabor atom as a 4096-point circular-convolution kernel: g[(-u) mod 4096] = conj(psi(u + 1/2)), |u| <= W
// (psi of styx_cwt.py:113-144 on the half-integer grid of an even-length record; out[t] = sum_u sig[t + u] conj(psi))
__global__ void k_block_taps_gabor(double2* __restrict__ g, int w, const double* __restrict__ par, int nb_total,
                                   const int32_t* __restrict__ ids) {
  const int j = ids[blockIdx.y];
  const double p_re = par[j], p_im = par[nb_total + j], omega = par[2 * nb_total + j], amp = par[3 * nb_total + j];
  double2* row = g + (int64_t)blockIdx.y * kBlk;
  for (int m = blockIdx.x * blockDim.x + threadIdx.x; m < kBlk; m += gridDim.x * blockDim.x) {
    // m = (-u) mod 4096  ->  u = -m for m <= w, u = 4096 - m for m >= 4096 - w + 1 ... u in [-w, w)
    int u;
    bool on = true;
    if (m <= w) u = -m;
    else if (m > kBlk - w) u = kBlk - m;
    else { u = 0; on = false; }
    double2 v = make_double2(0.0, 0.0);
    if (on) {
      const double x = (double)u + 0.5;
      const double env = amp * exp(-p_re * x * x);
      const double ph = omega * x - p_im * x * x;
      double s, c;
      sincos(ph, &s, &c);
      v = make_double2(env * c, -env * s);  // conj(psi)
    }
    row[m] = v;
  }
}


// taps of the edge pieces of a split band as 4096-point circular-convolution kernels (the layout of
// k_block_taps_gabor): row 2 s + piece, g[(-v) mod 4096] = (1 - taper(x)) conj(psi(x)), x = u + 1/2,
// u = centre_piece + v, centre = +-(n/2 - w), -w <= v < w
__global__ void k_block_taps_edge(double2* __restrict__ g, int w, int64_t n, double taper_e,
                                  const double* __restrict__ par, int nb_total, const int32_t* __restrict__ ids) {
  const int j = ids[blockIdx.y >> 1], piece = blockIdx.y & 1;
  const double p_re = par[j], p_im = par[nb_total + j], omega = par[2 * nb_total + j], amp = par[3 * nb_total + j];
  double2* row = g + (int64_t)blockIdx.y * kBlk;
  const int64_t centre = piece == 0 ? n / 2 - w : -(n / 2 - w);
  for (int m = blockIdx.x * blockDim.x + threadIdx.x; m < kBlk; m += gridDim.x * blockDim.x) {
    int v;
    bool on = true;
    if (m <= w) v = -m;
    else if (m > kBlk - w) v = kBlk - m;
    else { v = 0; on = false; }
    const int64_t u = centre + v;
    if (u < -(n / 2) || u >= n / 2) on = false;  // the atom has n samples
    double2 val = make_double2(0.0, 0.0);
    if (on) {
      const double x = (double)u + 0.5;
      const double env = (1.0 - split_taper(x, n, taper_e)) * amp * exp(-p_re * x * x);
      const double ph = omega * x - p_im * x * x;
      double s, c;
      sincos(ph, &s, &c);
      val = make_double2(env * c, -env * s);  // conj(psi)
    }
    row[m] = val;
  }
}

// taps of a Stockwell band: out[t] = e^{-2 pi i idx t / n} sum_tau x[t - tau] wt(tau),  wt(tau) = e^{2 pi i idx tau / n}
// om(tau), om = IDFT_n of the band's Gaussian window (`om` holds n * om, the unnormalised inverse transform)
__global__ void k_block_taps_stx(double2* __restrict__ g, int w, const double2* __restrict__ om, int64_t n,
                                 int64_t idx) {
  for (int m = blockIdx.x * blockDim.x + threadIdx.x; m < kBlk; m += gridDim.x * blockDim.x) {
    int tau;
    bool on = true;
    if (m < w) tau = m;
    else if (m > kBlk - w) tau = m - kBlk;
    else { tau = 0; on = false; }
    double2 v = make_double2(0.0, 0.0);
    if (on) {
      const double2 o = om[tau >= 0 ? tau : n + tau];
      const int64_t ph = ((idx * (int64_t)tau) % n + n) % n;
      double s, c;
      sincospi(2.0 * (double)ph / (double)n, &s, &c);
      const double inv = 1.0 / (double)n;
      v = make_double2((o.x * c - o.y * s) * inv, (o.x * s + o.y * c) * inv);
    }
    g[m] = v;
  }
}

// the Gaussian window of one Stockwell band on the signed FFT bins (styx_stx.py:195-236), as a full row
__global__ void k_stx_window_row(double2* __restrict__ row, int64_t n, double coef) {
  for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (int64_t)gridDim.x * blockDim.x) {
    const double ks = (double)(k < (n + 1) / 2 ? k : k - n);
    const double e = coef * ks;
    row[k] = make_double2(exp2(-e * e), 0.0);
  }
}

// rows[r][k] *= exp(+i pi k / kBlk): the table rows of a Gabor bank, made to match the block spectrum that carries
// the factor exp(-i pi k / kBlk) (see block_item)
__global__ void k_block_rotate_rows(double2* __restrict__ rows) {
  double2* row = rows + (int64_t)blockIdx.y * kBlk;
  for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < kBlk; k += gridDim.x * blockDim.x) {
    double s, c;
    sincospi((double)k / (double)kBlk, &s, &c);
    const double2 v = row[k];
    row[k] = make_double2(v.x * c - v.y * s, v.x * s + v.y * c);
  }
}

template <typename T, bool DEMOD>
int launch_block_v(const BlockArgs<T>& a, dim3 grid, hipStream_t st) {
  const bool coef = a.coef != nullptr, bits = a.bits != nullptr;
  if (coef && bits) k_block<T, DEMOD, true, true><<<grid, kBlkThreads, 0, st>>>(a);
  else if (coef) k_block<T, DEMOD, true, false><<<grid, kBlkThreads, 0, st>>>(a);
  else if (bits) k_block<T, DEMOD, false, true><<<grid, kBlkThreads, 0, st>>>(a);
  else k_block<T, DEMOD, false, false><<<grid, kBlkThreads, 0, st>>>(a);
  QI_LAUNCH_CHECK();
  return QI_OK;
}

}  // namespace

int block_valid(int wq) { return wq == kBlkLongWq ? kBlkLongValid : kBlk - 512 * (wq & 15); }

static int launch_block_long(const BlockArgs<float>& a, int demod, const BlockItem* items, int32_t nitems, int64_t n_channels,
                             hipStream_t st) {
  if (nitems <= 0) return QI_OK;
  dim3 grid((unsigned)nitems, 1, (unsigned)n_channels);
  const bool coef = a.coef != nullptr, bits = a.bits != nullptr;
#define QI_LONG(D, C, B) k_block_long<float, D, C, B><<<grid, kBlkThreads, 0, st>>>(a, items)
  if (demod) {
    if (coef && bits) QI_LONG(true, true, true);
    else if (coef) QI_LONG(true, true, false);
    else if (bits) QI_LONG(true, false, true);
    else QI_LONG(true, false, false);
  } else {
    if (coef && bits) QI_LONG(false, true, true);
    else if (coef) QI_LONG(false, true, false);
    else if (bits) QI_LONG(false, false, true);
    else QI_LONG(false, false, false);
  }
#undef QI_LONG
  QI_LAUNCH_CHECK();
  return QI_OK;
}

static int launch_block_long_dual(const BlockArgs<float>& a0, const BlockArgs<float>& a2, const DualItem* items, int32_t nitems,
                                  int64_t n_channels, hipStream_t st) {
  if (nitems <= 0) return QI_OK;
  const bool coef = a0.coef != nullptr, bits = a0.bits != nullptr;
  dim3 grid((unsigned)nitems, 1, (unsigned)n_channels);
#define QI_LONG2(C, B) k_block_long_dual<float, C, B><<<grid, kBlkThreads, 0, st>>>(a0, a2, items)
  if (coef && bits) QI_LONG2(true, true);
  else if (coef) QI_LONG2(true, false);
  else if (bits) QI_LONG2(false, true);
  else QI_LONG2(false, false);
#undef QI_LONG2
  QI_LAUNCH_CHECK();
  return QI_OK;
}

// the merged edge items of a table built for many records: their own launch behind the band items
static int launch_block_edge(const BlockArgs<float>& a, const BlockItem* items, const DualItem* dual, int32_t count,
                             int64_t n_channels, hipStream_t st) {
  if (count <= 0) return QI_OK;
  const bool coef = a.coef != nullptr, bits = a.bits != nullptr;
  dim3 grid((unsigned)count, 1, (unsigned)n_channels);
  if (coef && bits) k_block_edge<float, true, true><<<grid, kBlkThreads, 0, st>>>(a, items, dual);
  else if (coef) k_block_edge<float, true, false><<<grid, kBlkThreads, 0, st>>>(a, items, dual);
  else if (bits) k_block_edge<float, false, true><<<grid, kBlkThreads, 0, st>>>(a, items, dual);
  else k_block_edge<float, false, false><<<grid, kBlkThreads, 0, st>>>(a, items, dual);
  QI_LAUNCH_CHECK();
  return QI_OK;
}

template <>
int launch_block<float>(const BlockArgs<float>& a, int demod, int64_t n_channels, hipStream_t st, hipStream_t, hipEvent_t, hipEvent_t) {
  if (a.nitems + a.nedge_items <= 0) return QI_OK;
  if (a.nlong > 0) QI_TRY(launch_block_long(a, demod, a.items, a.nlong, n_channels, st));
  BlockArgs<float> rest = a;
  rest.items += a.nlong;
  rest.nitems -= a.nlong;
  rest.nlong = 0;
  const int32_t riding = a.edge_merged ? 0 : rest.nedge_items;  // single-band edge items ride at the end of the band launch
  if (rest.nitems + riding > 0) {
    dim3 grid((unsigned)(rest.nitems + riding), 1, (unsigned)n_channels);
    QI_TRY((demod ? launch_block_v<float, true>(rest, grid, st) : launch_block_v<float, false>(rest, grid, st)));
  }
  if (a.edge_merged) QI_TRY(launch_block_edge(a, rest.items + rest.nitems, nullptr, a.nedge_items, n_channels, st));
  return QI_OK;
}

template <bool DEMOD>
static int launch_block64_v(const BlockArgs<double>& a, dim3 grid, hipStream_t st) {
  const bool coef = a.coef != nullptr, bits = a.bits != nullptr;
#define QI_B64(C, B)                                                                                   \
  do {                                                                                                 \
    QI_TRY(allow_dynamic_lds(reinterpret_cast<const void*>(&k_block64<DEMOD, C, B>), kBlk64Lds));      \
    k_block64<DEMOD, C, B><<<grid, kBlkThreads, kBlk64Lds, st>>>(a);                                   \
  } while (0)
  if (coef && bits) QI_B64(true, true);
  else if (coef) QI_B64(true, false);
  else if (bits) QI_B64(false, true);
  else QI_B64(false, false);
#undef QI_B64
  QI_LAUNCH_CHECK();
  return QI_OK;
}
template <>
int launch_block<double>(const BlockArgs<double>& a, int demod, int64_t n_channels, hipStream_t st, hipStream_t side, hipEvent_t fork,
                         hipEvent_t join) {
  if (a.nitems + a.nedge_items <= 0) return QI_OK;
  if (a.nlong > 0 || (a.nedge_items > 0 && (!a.edge_merged || a.edge_wq < 1 || a.edge_wq > 4))) {
    set_error("block engine: float64 tables have no long blocks, and their split bands one edge item per block");
    return QI_ERR_STATE;
  }
  const bool coef = a.coef != nullptr, bits = a.bits != nullptr;
  const BlockItem* items = a.items + a.nitems;
  // the blocks around the middle of the record see both far pieces: items [mid_lo, mid_hi] of the (block-ordered) edge list,
  // a small launch of the band-by-band kernel; every other block goes through the one-piece kernel
  int64_t mid_lo = -1, mid_hi = -2;
  if (a.nedge_items > 0) {
    const int64_t W = 256 * (int64_t)a.edge_wq, V = kBlk - 2 * W;
    for (int64_t b = 0; b < a.nedge_items; ++b) {
      const int64_t t0 = b * V - W, p0 = t0 + (a.n / 2 - W), p1 = t0 - (a.n / 2 - W);
      if (!(p0 >= a.n || p0 + kBlk <= 0) && !(p1 >= a.n || p1 + kBlk <= 0)) {
        if (mid_lo < 0) mid_lo = b;
        mid_hi = b;
      }
    }
  }
#define QI_E64(C, B, PATH, GRID, ITEMS, STREAM)                                                         \
  do {                                                                                                  \
    QI_TRY(allow_dynamic_lds(reinterpret_cast<const void*>(&k_block64_edge<C, B, PATH>), kBlk64Lds));   \
    k_block64_edge<C, B, PATH><<<GRID, kBlkThreads, kBlk64Lds, STREAM>>>(a, ITEMS);                     \
  } while (0)
#define QI_E64_ALL(PATH, GRID, ITEMS, STREAM)                          \
  do {                                                                 \
    if (coef && bits) QI_E64(true, true, PATH, GRID, ITEMS, STREAM);   \
    else if (coef) QI_E64(true, false, PATH, GRID, ITEMS, STREAM);     \
    else if (bits) QI_E64(false, true, PATH, GRID, ITEMS, STREAM);     \
    else QI_E64(false, false, PATH, GRID, ITEMS, STREAM);              \
  } while (0)
  const bool mids = mid_hi >= mid_lo;
  const bool beside = mids && side && fork && join && a.nitems > 0;  // the two-piece items beside the band items
  if (mids) {
    const dim3 grid_mid((unsigned)(mid_hi - mid_lo + 1), 1, (unsigned)n_channels);
    hipStream_t ms = beside ? side : st;
    if (beside) {
      QI_HIP(hipEventRecord(fork, st));
      QI_HIP(hipStreamWaitEvent(side, fork, 0));
    }
    QI_E64_ALL(1, grid_mid, items + mid_lo, ms);
    QI_LAUNCH_CHECK();
    if (beside) QI_HIP(hipEventRecord(join, side));
  }
  // (whatever fails below: the caller's stream still waits for the side stream's launch -- its rows and partial slots must
  // not be left unordered against the next call's reuse of the scratch)
  struct Rejoin {
    hipStream_t st; hipEvent_t join; bool on;
    ~Rejoin() { if (on && hipStreamWaitEvent(st, join, 0) != hipSuccess) (void)hipGetLastError(); }
  } rejoin{st, join, beside};
  if (a.nitems > 0) {
    dim3 grid((unsigned)a.nitems, 1, (unsigned)n_channels);
    QI_TRY((demod ? launch_block64_v<true>(a, grid, st) : launch_block64_v<false>(a, grid, st)));
  }
  if (a.nedge_items > 0) {
    dim3 grid((unsigned)a.nedge_items, 1, (unsigned)n_channels);
    QI_E64_ALL(2, grid, items, st);
    QI_LAUNCH_CHECK();
  }
#undef QI_E64_ALL
#undef QI_E64
  return QI_OK;
}

int launch_z64_coarse(const Z64Args& a, int64_t n_channels, hipStream_t st) {
  if (a.nbands <= 0) return QI_OK;
  const int64_t P = a.M / kBlk;
  if (P < 1 || P * kBlk != a.M || (a.M & (a.M - 1)) != 0) {
    set_error("float64 zoom: a coarse grid of %lld samples has no in-LDS coarse stage", (long long)a.M);
    return QI_ERR_UNSUPPORTED;
  }
  dim3 grid((unsigned)(8 * ceil_div(a.nbands * P, 8)), (unsigned)n_channels, 1);
  if (a.kind == 2) {
    QI_TRY(allow_dynamic_lds(reinterpret_cast<const void*>(&k_z64_coarse<true>), kBlk64Lds));
    k_z64_coarse<true><<<grid, kBlkThreads, kBlk64Lds, st>>>(a);
  } else {
    QI_TRY(allow_dynamic_lds(reinterpret_cast<const void*>(&k_z64_coarse<false>), kBlk64Lds));
    k_z64_coarse<false><<<grid, kBlkThreads, kBlk64Lds, st>>>(a);
  }
  QI_LAUNCH_CHECK();
  return QI_OK;
}

template <>
int launch_block_dual<float>(const BlockArgs<float>& a0, const BlockArgs<float>& a2, const DualItem* items, int32_t nitems,
                             int32_t nlong, int32_t n_edge, int64_t n_channels, hipStream_t st) {
  if (nitems <= 0) return QI_OK;
  const bool coef = a0.coef != nullptr, bits = a0.bits != nullptr;
  if (coef != (a2.coef != nullptr) || bits != (a2.bits != nullptr) || a0.n != a2.n) {
    set_error("block engine: the joint launch needs the same panels from both transforms");
    return QI_ERR_STATE;
  }
  if (nlong > 0) QI_TRY(launch_block_long_dual(a0, a2, items, nlong, n_channels, st));
  items += nlong;
  nitems -= nlong;
  if (nitems <= 0) return QI_OK;
  // merged edge items (the last n_edge of the list): a launch of their own
  const int32_t own = a0.edge_merged ? (n_edge < nitems ? n_edge : nitems) : 0;
  if (nitems - own > 0) {
    dim3 grid((unsigned)(nitems - own), 1, (unsigned)n_channels);
    if (coef && bits) k_block_dual<float, true, true><<<grid, kBlkThreads, 0, st>>>(a0, a2, items);
    else if (coef) k_block_dual<float, true, false><<<grid, kBlkThreads, 0, st>>>(a0, a2, items);
    else if (bits) k_block_dual<float, false, true><<<grid, kBlkThreads, 0, st>>>(a0, a2, items);
    else k_block_dual<float, false, false><<<grid, kBlkThreads, 0, st>>>(a0, a2, items);
    QI_LAUNCH_CHECK();
  }
  return launch_block_edge(a0, nullptr, items + (nitems - own), own, n_channels, st);
}

template <>
int launch_zoom_coarse<float>(const ZoomArgs<float>& a, int max_level, int64_t n_channels, hipStream_t st) {
  if (a.nbands <= 0) return QI_OK;
  (void)max_level;
  dim3 grid((unsigned)a.planes, 1, (unsigned)n_channels);  // every plane of every band is one 4096-point transform
  k_zoom_coarse<float><<<grid, kBlkThreads, 0, st>>>(a);
  QI_LAUNCH_CHECK();
  return QI_OK;
}

template <>
int launch_zoom_coarse_gather<float>(const ZoomArgs<float>& a, int64_t n_channels, hipStream_t st) {
  if (a.nbands <= 0) return QI_OK;
  dim3 grid((unsigned)a.planes, 1, (unsigned)n_channels);
  if (a.stx) k_zoom_coarse_g<float, true><<<grid, kBlkThreads, 0, st>>>(a);
  else k_zoom_coarse_g<float, false><<<grid, kBlkThreads, 0, st>>>(a);
  QI_LAUNCH_CHECK();
  return QI_OK;
}

template <>
int launch_zoom_coarse_gather2<float>(const ZoomArgs<float>& a0, const ZoomArgs<float>& a2, int64_t n_channels,
                                      hipStream_t st) {
  if (a0.stx || !a2.stx || a0.nbands <= 0 || a2.nbands <= 0) {
    set_error("zoom engine: the joint coarse stage takes a styx table and a Stockwell table");
    return QI_ERR_STATE;
  }
  dim3 grid((unsigned)(((a0.planes + a2.planes) + 63) / 64 * 64), 1, (unsigned)n_channels);
  // few records: a couple of workgroups per CU, each waiting on its loads -- twice as many in flight
  static const int nf_env = tune_env("QI_NATIVE_COARSE_NF") ? atoi(tune_env("QI_NATIVE_COARSE_NF")) : 0;
  if (nf_env == 8 || (nf_env == 0 && n_channels > 2)) k_zoom_coarse2g<float, 8><<<grid, kBlkThreads, 0, st>>>(a0, a2);
  else k_zoom_coarse2g<float, 16><<<grid, kBlkThreads, 0, st>>>(a0, a2);
  QI_LAUNCH_CHECK();
  return QI_OK;
}

template <>
int launch_zoom_coarse2<float>(const ZoomArgs<float>& a0, const ZoomArgs<float>& a2, int64_t n_channels, hipStream_t st) {
  dim3 grid((unsigned)(a0.planes + a2.planes), 1, (unsigned)n_channels);
  k_zoom_coarse2<float><<<grid, kBlkThreads, 0, st>>>(a0, a2);
  QI_LAUNCH_CHECK();
  return QI_OK;
}

int launch_block_taps_gabor(double2* g, int w, const double* d_par, int nb_total, const int32_t* d_ids, int count,
                            hipStream_t st) {
  dim3 grid(4, (unsigned)count);
  k_block_taps_gabor<<<grid, 256, 0, st>>>(g, w, d_par, nb_total, d_ids);
  QI_LAUNCH_CHECK();
  return QI_OK;
}

int launch_block_taps_edge(double2* g, int w, int64_t n, double taper_e, const double* d_par, int nb_total,
                           const int32_t* d_ids, int count, hipStream_t st) {
  dim3 grid(4, (unsigned)(2 * count));
  k_block_taps_edge<<<grid, 256, 0, st>>>(g, w, n, taper_e, d_par, nb_total, d_ids);
  QI_LAUNCH_CHECK();
  return QI_OK;
}

int launch_block_taps_stx(double2* g, int w, const double2* om, int64_t n, int64_t idx, hipStream_t st) {
  k_block_taps_stx<<<4, 256, 0, st>>>(g, w, om, n, idx);
  QI_LAUNCH_CHECK();
  return QI_OK;
}

int launch_block_rotate_rows(double2* rows, int count, hipStream_t st) {
  if (count <= 0) return QI_OK;
  k_block_rotate_rows<<<dim3(4, (unsigned)count), 256, 0, st>>>(rows);
  QI_LAUNCH_CHECK();
  return QI_OK;
}

int launch_stx_window_row(double2* row, int64_t n, double coef, hipStream_t st) {
  k_stx_window_row<<<(unsigned)(ceil_div(n, 256) > 1024 ? 1024 : ceil_div(n, 256)), 256, 0, st>>>(row, n, coef);
  QI_LAUNCH_CHECK();
  return QI_OK;
}

}  // namespace native
}  // namespace qi
