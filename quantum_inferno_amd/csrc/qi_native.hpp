// Interface of the native (hand-written FFT pass) engine, see qi_native.hip.
#pragma once
#include "qi_common.hpp"

namespace qi {
namespace native {

constexpr int kN2 = 1024;  // points of the in-LDS row transform of pass 2
constexpr int kMaxPrunedTerms = 16;  // widest pruned support = kMaxPrunedTerms * kN2 spectrum bins

struct BandDesc {
  int32_t mode;      // 0: pruned (spectrum support short enough for the one-pass loader), 1: general
  int32_t k_lo;      // first bin of the support (negative for the Stockwell window, centred on 0)
  int32_t k_len;     // number of support bins
  int32_t gen_slot;  // general bands: slot of the intermediate buffer used by its launch group
  int32_t out_band;  // row of the panel this band writes
  int32_t bank_row;  // general Gabor bands: row of the full-spectrum bank
  int32_t edge;      // circularly evaluated short-atom band: samples at each record end fixed by k_edge_fix
  int32_t edge_slot; // index of that band in the edge list (rows of RowArgs::edge_z)
  int32_t add_row;   // zoom engine, split band: 1 + its row of ZoomArgs::split_part (0: an ordinary band)
  int32_t pad_;
  int64_t src_off;   // pruned Gabor bands: offset of H[k_lo] in the compact bank
  int64_t shift;     // Stockwell: shift index idx_j
  double coef;       // Stockwell: window coefficient (exp2(-(coef k)^2))
};

template <typename T>
struct RowArgs {
  int64_t Lf, n, N1, N2;
  int32_t nbands;           // entries of `bands` handled by this launch
  int32_t panel_bands;      // bands of the whole panel (row stride of the outputs)
  const BandDesc* bands;    // [nbands] device: the launch group's band list
  const int32_t* gen_list;  // [ngen_launch] indices into `bands` of its general bands (pass 1 launch order)
  int32_t ngen_launch, imd_slots;
  int32_t chunk_base, chunk_total;  // this launch owns chunks [chunk_base, chunk_base + gridDim.y) of chunk_total
  const T* sig;          // [C][n] records (forward transform only)
  const cplx<T>* X;      // [C][Lf] spectra of the records
  const cplx<T>* Hc;     // compact bank of the pruned Gabor bands
  const cplx<T>* Hfull;  // [ngen][Lf] full spectra of the general Gabor bands
  cplx<T>* imd;          // [C][imd_slots][N2][N1] intermediate of the launch group's general bands
  T inv_len;
  float two_over_len;  // 2 / Lf (exact)
  int32_t neg_last_row;  // pass 1 of the linear kind: twiddle of t1 = N1 - 1 taken at t1 = -1
  int32_t phase_split;   // pass 1 at N1 = 2048: one workgroup per decimation phase (set by the launcher)
  // pass 2 outputs
  cplx<T>* coef;
  T* bits;
  cplx<T>* edge_z;    // [C][nedge][2][edge_wmax]: edge samples of the short-atom bands when no panel is stored
  int64_t edge_wmax;
  int32_t nedge;
  T* time_part;       // [C][chunk_total][n]
  double* part_band;  // [C][panel_bands][nblk]
  double* part_stat;  // [C][chunk_total][nblk][3]
  int64_t nblk;       // partial slots per band (stride of part_band rows)
  int64_t stat_nblk;  // part_stat entries per chunk of this engine
  int64_t stat_stride;  // part_stat entries per channel
  int32_t bands_per_chunk;
  T power_scale, eps;
  unsigned long long* stamps;  // diagnostic builds only (-DQI_NATIVE_STAMPS): [workgroup][8] phase cycles
  int32_t debug;  // QI_NATIVE_DEBUG bit mask (timing experiments only: 1 no stores, 2 no loads, 4 no FFT, 8 no reductions)
};

template <typename T>
int launch_pass1(const RowArgs<T>& a, int kind, int64_t n_channels, hipStream_t st);
template <typename T>
int launch_forward(const RowArgs<T>& a, cplx<T>* Xout, int64_t n_channels, hipStream_t st);
template <typename T>
int launch_pass2(const RowArgs<T>& a, int kind, int rows_per_group, int nchunk, int64_t n_channels, hipStream_t st);
// one short-atom band of the styx bank evaluated circularly (see k_edge_fix)
struct EdgeBand {
  int32_t out_band, w;  // panel row; taps |u| <= w of the atom are above 2^-30 of its peak
  double p_re, p_im, omega, amp;
};
template <typename T>
struct EdgeArgs {
  const EdgeBand* bands;  // [nedge] device
  int32_t nedge, panel_bands;
  int64_t n, wmax, stat_slots;
  const T* sig;       // [C][n]
  cplx<T>* coef;      // panel, or null: the edge samples are then in edge_z
  cplx<T>* edge_z;    // [C][nedge][2][wmax]
  T* bits;            // optional
  T* edge_p;          // [C][nedge][2][wmax] corrected powers
  T power_scale, eps;
};

// ---- block engine (qi_block.hip): short-atom bands by overlap-save with 4096-point transforms -------------------
constexpr int kBlk = 4096;
// (R: float for the float32 engine, double for the float64 one -- the block kernels compute in the record's precision)
template <typename R>
struct BlockBandT {
  int32_t out_band;  // row of the panel
  int32_t bank_row;  // row of the [rows][kBlk] filter-spectrum table
  int32_t shift;     // Stockwell: shift index idx_j (outputs are multiplied by exp(-2 pi i idx t / n))
  int32_t analytic;  // the filter spectrum is a Gaussian evaluated in registers (no table row is read)
  R rot[8];      // Stockwell: r^(2^k), k = 0..3, r = exp(-2 pi i idx 256 / n)
  // analytic filter: weight(k) = amp * exp2(-(cw * dk)^2), dk = k - (kappa_int + kappa_frac) wrapped to +-kBlk / 2
  int32_t kappa_int;
  R kappa_frac, cw, amp;
  // narrow = 1: the weights above 2^-30 of the peak lie within the 256 bins from `klo` (mod kBlk), so a thread holds at
  // most one non-zero value of the filtered spectrum and the first radix-16 pass of the inverse transform is a
  // product with powers of one phasor; rot_a / rot_b = exp(2 pi i b / 16) for b = klo / 256 and the next one
  // narrow = 2: the weights above 2^-30 of the peak lie within bins (0, kBlk / 2): eight weights per thread and a first
  // pass of the inverse transform without its first radix-2 stage
  int32_t narrow, klo;
  R rot_a[2], rot_b[2];
  // long blocks (BlockItem::wq = kBlkLongWq: 8192 record samples per block, 6144 outputs kept; narrow Gaussian bands of
  // the 1024-sample reach group): the band on the 8192-bin grid -- kappa_int / kappa_frac / cw / amp / klo / rot_a /
  // rot_b above are then in units of THAT grid --, exp(i pi b / 16) for b = klo / 256 and the next one (the twiddle of the
  // odd output samples), and for the Stockwell demodulation exp(-2 pi i idx / n) (one sample)
  R rot8_a[2], rot8_b[2];
  R rot1[2];
  // nowrap = 1: every weight above 2^-30 of the peak belongs to a bin k in [0, kBlk) at distance k - kappa (no alias is
  // nearer), and amp > 0: weight(k) = exp2(la - (cw (k - kappa))^2), la = log2(amp) -- no wrap-around logic per weight
  int32_t nowrap;
  R la;
};
using BlockBand = BlockBandT<float>;

constexpr int kBlkLongWq = 8;       // BlockItem::wq of a long block
constexpr int kBlkLong = 2 * kBlk;  // record samples of a long block
constexpr int kBlkLongValid = kBlkLong - 2048;  // outputs kept (taps within 1024 samples)
struct BlockItem {  // one workgroup of the block launch
  int32_t wq;          // reach group: taps within 256 * wq samples; negative: the edge pieces (reach group -wq) of
                       // split band `band_first` for output block `block` (see EdgeSplitArgs)
  int32_t block;       // block index: outputs [block * V, (block + 1) * V), V = kBlk - 512 wq
  int32_t band_first, band_count;  // its bands in BlockArgs::bands
  int32_t plane;       // time_part plane (relative to chunk_base) it writes
  int32_t stat_slot;   // part_stat slot (relative to stat_base)
};
struct DualItem {  // one workgroup of the joint block launch of qi_cwt_stx: the styx bands (0) and the Stockwell bands (2)
  int32_t wq, block;  // of one block; wq < 0: an edge item of the styx table (fields 0 as in BlockItem)
  int32_t first0, count0, plane0, slot0;
  int32_t first2, count2, plane2, slot2;
};
template <typename T>
struct BlockArgs {
  int64_t n;
  int32_t nitems, panel_bands;
  int32_t nedge_items, nsplit;  // edge items of the split bands, after the nitems band items
  int32_t edge_merged;          // the edge items cover all split bands of a block each and run as a launch of their own
  int32_t edge_wq;              // reach group of the edge pieces (taps within 256 * edge_wq samples)
  int32_t nlong;                // of the nitems band items, the first nlong are long-block items (a launch of their own)
  const int32_t* edge_band;     // [nsplit] device: panel row of each split band
  const cplx<T>* edge_bank;     // [nsplit][2][kBlk]
  const cplx<T>* edge_part;     // [C][nsplit][n]: the zoom engine's part of the split bands
  const BlockItem* items;  // [nitems + nedge_items] device, most expensive first
  const BlockBandT<T>* bands;  // device, all reach groups
  const cplx<T>* bank;     // [rows][kBlk], scaled by 1 / kBlk
  // float64 Stockwell tables: demodulation factors from tables (see block_bands): per band of `bands` the sixteen
  // exp(-2 pi i idx 256 i / n), and exp(-2 pi i m / n) = demod_t1[m >> 10] demod_t2[m & 1023]
  const T* gauss_w;          // float64 tables: [bands][kBlk] the bands' real Gaussian filter weights (null: evaluated in registers)
  const cplx<T>* demod_pow;  // [bands][16]
  const cplx<T>* demod_t1;   // [n / 1024] exp(-2 pi i 1024 j / n)
  const cplx<T>* demod_t2;   // [1024] exp(-2 pi i j / n)
  const T* sig;            // [C][n]
  cplx<T>* coef;
  T* bits;
  T* time_part;       // [C][chunk_total][n]
  double* part_band;  // [C][panel_bands][nblk]: slot = block
  double* part_stat;  // [C][stat_stride][3]: slot = stat_base + item's stat_slot
  int64_t nblk, stat_stride, stat_base;
  int32_t chunk_base, chunk_total;
  T power_scale, eps;
  float two_over_n;
  int32_t debug;  // QI_NATIVE_DEBUG bit mask (timing experiments only: 1 no stores, 2 no filter loads, 4 no inverse FFT)
  unsigned long long* stamps;  // diagnostic builds only (-DQI_NATIVE_STAMPS): [workgroup][8] phase cycles
};
int block_valid(int wq);  // outputs per block for taps within 256 * wq samples
// (float64: `side` / `fork` / `join` -- a second stream and two events of the plan -- let the few two-piece edge items of the
// split bands, a dozen long workgroups, run beside the band items instead of alone on the chip)
template <typename T>
int launch_block(const BlockArgs<T>& a, int demod, int64_t n_channels, hipStream_t st, hipStream_t side = nullptr,
                 hipEvent_t fork = nullptr, hipEvent_t join = nullptr);
// a0: styx table, a2: Stockwell table; both must agree on which panels (coefficients, bits) are stored
template <typename T>
// (n_edge: how many of the list's last items are edge items of the split bands)
int launch_block_dual(const BlockArgs<T>& a0, const BlockArgs<T>& a2, const DualItem* items, int32_t nitems, int32_t nlong,
                      int32_t n_edge, int64_t n_channels, hipStream_t st);
int launch_block_taps_gabor(double2* g, int w, const double* d_par, int nb_total, const int32_t* d_ids, int count,
                            hipStream_t st);
int launch_block_taps_stx(double2* g, int w, const double2* om, int64_t n, int64_t idx, hipStream_t st);
int launch_stx_window_row(double2* row, int64_t n, double coef, hipStream_t st);
int launch_block_rotate_rows(double2* rows, int count, hipStream_t st);

// Split bands (atoms longer than the record, see split_taper in qi_device.hpp): the zoom engine produces the tapered
// part of band s into part[c][s][t]; the edge items of the block launch (BlockItem::wq < 0) add
//   sum_piece sum_|v|<=W sig[t + centre_piece + v] taps_piece(v),  centre = +-(n/2 - W), W = 256 wq,
// taps = (1 - taper) conj(psi) at lags [n/2 - 2W, n/2) and [-n/2, -n/2 + 2W) given as two 4096-point filter spectra
// bank[s][piece][4096] (scaled by 1 / 4096), by overlap-save like the band items, and finish the band (panel rows,
// reductions).
int launch_block_taps_edge(double2* g, int w, int64_t n, double taper_e, const double* d_par, int nb_total,
                           const int32_t* d_ids, int count, hipStream_t st);

// ---- zoom engine (qi_zoom.hip): narrow-spectrum bands from a coarse inverse transform + band-limited interpolation
// A band with K occupied bins is assigned the coarsest grid "level" g on which it is oversampled >= 4 times:
// D = 64 >> g fine samples per coarse sample, M_g = (Lf / 64) << g coarse samples, S = 1 << g coarse samples per
// wave-step (a wave-step = 64 consecutive outputs = the 64 lanes).  The interpolator has N = 10, 6 or 4 taps in
// coarse-sample units (by the band's oversampling, see the classes below); a wave-step spans S coarse intervals, so its
// window has N + S - 1 samples and every lane carries the N + S - 1 weights of its own position in it.  (The one-sample
// offset of the zero-padded linear correlation, output t = full-length sample t + n/2 - 1, is a phase ramp folded
// into the band's compact bank at plan time, so a lane's position in its window is lane / D >= 0 for every kind.)
constexpr int kZoomD = 64;  // fine samples per coarse sample at level 0 (= the lanes of a wave)
constexpr int kZoomLevels = 5;   // coarse-grid levels
// A band's CLASS = (grid level, interpolator).  Classes 0..4: grid level 0..4, bands oversampled >= 4 times on their
// grid, 10-tap interpolator.  Classes 5 and 6: grid level 0 (the coarsest grid: every narrower band lands there), bands
// oversampled >= 8 / >= 32 times, 6- / 4-tap interpolators -- the more a band is oversampled, the shorter the
// interpolator that reaches the same error (zoom_weights).  `level` below is a class index.
constexpr int kZoomClasses = 7;
constexpr int zoom_grid(int cls) { return cls < kZoomLevels ? cls : 0; }
constexpr int zoom_ntap(int cls) { return cls < kZoomLevels ? 10 : (cls == 5 ? 6 : 4); }
constexpr int zoom_design_oversampling(int cls) { return cls < kZoomLevels ? 4 : (cls == 5 ? 8 : 32); }
constexpr int zoom_span(int cls) { return 1 << zoom_grid(cls); }                   // S
constexpr int zoom_taps(int cls) { return zoom_ntap(cls) + zoom_span(cls) - 1; }   // window samples per wave-step
constexpr int zoom_steps(int cls) { return zoom_grid(cls) <= 2 ? 16 : (64 >> zoom_grid(cls)); }  // wave-steps per wave and band
                                                                                   // (window of <= 128 coarse samples)
constexpr int kZoomOversample = 4;
template <typename T>
struct ZoomArgs {
  int64_t n, Lf;
  int64_t planes;          // 4096-sample planes of coarse storage per record (all bands)
  int32_t nbands, panel_bands;
  const BandDesc* bands;   // [nbands] device: one-pass descriptors of the zoom bands, ordered by level; for these
                           // bands `edge` = first plane of the band's coarse array, `edge_slot` = its level
  const int32_t* plane_band;       // [planes] device: band (index into `bands`) that owns each coarse plane
  // the fine launch covers every level: blockIdx.y in [lvl_chunk0[g], lvl_chunk0[g] + lvl_nchunk[g]) works on level g,
  // bands [lvl_first[g], lvl_first[g] + lvl_count[g]) of `bands`, and writes per-time plane chunk_base + blockIdx.y
  int32_t lvl_first[kZoomClasses], lvl_count[kZoomClasses], lvl_chunk0[kZoomClasses], lvl_nchunk[kZoomClasses];
  int64_t lvl_stat_base[kZoomClasses];
  const float* lvl_weights[kZoomClasses];  // [taps][64] interpolation weights of the lanes, per class
  const cplx<T>* X;        // [C][Lf << x_shift] spectra of the records
  int32_t x_shift;         // 1: X is the spectrum of the records zero-padded to twice Lf (bin k of Lf = bin 2k)
  const cplx<T>* Hc;       // compact bank (Gabor kinds)
  cplx<T>* split_part;     // [C][split_rows][n]: the split bands (BandDesc::add_row) leave their samples here, see k_zoom
  int32_t split_rows;
  int32_t debug;  // QI_NATIVE_DEBUG builds, timing experiments of the coarse stage: 256 no gather, 512 no transform, 1024 no stores
  cplx<T>* coarse;         // [C][planes][4096]: per band [P][4096], P = M_g / 4096: sample tau = P tau2 + tau1 at [tau1][tau2]
  int32_t stx;             // Stockwell: bands are at baseband already, no carrier
  int32_t lane_off;        // output sample t is full-length sample f = 64 (tau + tau_off) + lane - lane_off
  int64_t tau_off;
  T inv_len;
  float two_over_len;
  cplx<T>* coef;
  T* bits;
  T* time_part;       // [C][chunk_total][n]
  double* part_band;  // [C][panel_bands][nblk]: slot = workgroup index along time
  double* part_stat;  // [C][stat_stride][3]: slot = stat_base + chunk * groups + group
  int64_t nblk, stat_stride;
  int32_t chunk_base, chunk_total;
  T power_scale, eps;
};
int64_t zoom_groups(int64_t n, int level);  // workgroups along time (partial slots per band, stat slots per chunk)
template <typename T>
int launch_zoom_gather(const ZoomArgs<T>& a, int max_level, int64_t n_channels, hipStream_t st);  // folded baseband bins, then
template <typename T>
int launch_zoom_coarse(const ZoomArgs<T>& a, int max_level, int64_t n_channels, hipStream_t st);  // their 4096-point transforms, in place (qi_block.hip)
template <typename T>
int launch_zoom(const ZoomArgs<T>& a, int64_t n_channels, hipStream_t st);
// qi_cwt_stx: gather / coarse stage of the styx table (a0) and the Stockwell table (a2) in one launch each
template <typename T>
int launch_zoom_gather2(const ZoomArgs<T>& a0, const ZoomArgs<T>& a2, int64_t n_channels, hipStream_t st);
template <typename T>
int launch_zoom2(const ZoomArgs<T>& a0, const ZoomArgs<T>& a2, int64_t n_channels, hipStream_t st);  // and the interpolation
template <typename T>
int launch_zoom_coarse2(const ZoomArgs<T>& a0, const ZoomArgs<T>& a2, int64_t n_channels, hipStream_t st);
// gather and plane transforms in one launch (the inputs of a plane are formed in registers)
template <typename T>
int launch_zoom_coarse_gather(const ZoomArgs<T>& a, int64_t n_channels, hipStream_t st);
template <typename T>
int launch_zoom_coarse_gather2(const ZoomArgs<T>& a0, const ZoomArgs<T>& a2, int64_t n_channels, hipStream_t st);
// ---- float64 zoom (qi_zoom64.hip) ------------------------------------------------------------------------------------
constexpr int kZ64Taps = 16;       // interpolator taps (oversampling >= 4: 2.8e-12 of a unit tone)
#ifndef QI_Z64_TILE
#define QI_Z64_TILE 2048  // (round 5: 4096 -> 2048 halves the kernel's LDS, four workgroups per CU instead of three: -1.6 % of the zoom stage; 1024: slower)
#endif
constexpr int kZ64Tile = QI_Z64_TILE;  // panel samples per workgroup and band of k_z64_interp (one partial slot per band and tile)
constexpr int kZ64Levels = 5;      // coarse grids of Lf / 64 ... Lf / 4 samples (a band is oversampled >= 4 times on its grid)
constexpr int kZ64Pad = 16;        // a band's coarse array is [kZ64Pad | M | kZ64Pad] samples: the pads repeat the other end (k_z64_pad)
struct Z64Args {
  int64_t Lf, n, M;             // transform length, record length, coarse grid (M = Lf >> log2d)
  int32_t log2d, kind;          // fine samples per coarse sample D = 1 << log2d (64 ... 4); table kind 0 / 1 / 2
  int32_t nbands, panel_bands;  // bands of this level's list; rows of the panel
  const BandDesc* bands;        // [nbands] device (k_lo, k_len, src_off | shift, coef, out_band)
  const cplx<double>* X;        // [C][Lf] spectra of the records
  const cplx<double>* Hc;       // compact bank (Gabor tables)
  cplx<double>* Z;              // [C][nbands][kZ64Pad + M + kZ64Pad] coarse spectra, transformed in place to coarse samples
  const double* weights;        // [D][kZ64Taps] device
  double inv_len;
  float two_over_len;
  cplx<double>* coef;           // [C][panel_bands][n] or null
  double* bits;
  cplx<double>* split_part;     // [C][split_rows][n]: the split bands (BandDesc::add_row) leave their samples here, see k_z64_interp
  int32_t split_rows;
  double* time_part;            // [C][chunk_total][n] per-time planes (or the output row itself when chunk_total = 1)
  double* part_band;            // [C][panel_bands][nblk]
  double* part_stat;            // [C][stat_stride][3]
  int64_t nblk, stat_stride;    // tiles per record (= partial slots a band fills); stat slots per record
  int64_t pb_stride, stat_nblk; // partial slots per band in part_band; blocks per chunk in part_stat
  int32_t chunk_base, chunk_total;
  double power_scale, eps;
};
int launch_z64_gather(const Z64Args& a, int64_t n_channels, hipStream_t st);
int launch_z64_interp(const Z64Args& a, int nchunk, int64_t n_channels, hipStream_t st);
int launch_z64_pad(cplx<double>* Z, int64_t M, int64_t rows, hipStream_t st);  // fills the pads of `rows` coarse arrays
// gather + M-point transform + pads of one level in ONE launch of 4096-point plane transforms in LDS (qi_block.hip; M >= 4096)
int launch_z64_coarse(const Z64Args& a, int64_t n_channels, hipStream_t st);
void z64_weights(int log2d, double* w /*[1 << log2d][kZ64Taps]*/);

// Fine stage with wave-uniform windows (k_z64_fine, round 4).  A wave-step = 64 consecutive outputs = the 64 lanes; it spans
// S = 64 / D coarse intervals, so its window holds N + S - 1 coarse samples -- the same for every lane: one copy per wave
// in scalar registers (taken with v_readlane from a vector register whose lane i holds coarse sample i of the wave's window),
// operands of the lanes' fused multiply-adds; every lane carries the N + S - 1 weights of its own position in the window.  No LDS traffic per output (k_z64_interp reads 16 x 16 bytes of LDS
// per output and is bound by that).  A band's CLASS = (coarse grid, interpolator length): the narrower a band is against
// its grid, the shorter the interpolator that reaches the same error (worst case of a unit tone anywhere in the band, double
// weights: 16 taps at >= 4 x oversampling 2.8e-12, 12 at >= 8 x 3.9e-13, 10 at >= 16 x 4.1e-14, 8 at >= 32 x 7.2e-14, 6 at
// >= 64 x 2.1e-12) -- on the coarsest grid, where every narrower band lands, most bands of an order-12 table take 6 - 10 taps.
// (measured per grid against k_z64_interp, 4 records, order 12: Lf / 64 -8 %, Lf / 32 -5 %, Lf / 16 +15 % -- a wave-step that
// spans four coarse intervals pays 46 v_readlane per output -- so the fine kernel takes the two coarsest grids)
constexpr int kZ64FineLevels = 2;   // coarse grids the fine kernel takes (the finer three stay with k_z64_interp)
constexpr int kZ64FineClasses = kZ64FineLevels + 4;  // [0, kZ64FineLevels): grids of Lf / 64, Lf / 32 samples, 16 taps; then the
                                                     // coarsest grid with 12 / 10 / 8 / 6 taps
constexpr int z64f_level(int c) { return c < kZ64FineLevels ? c : 0; }
constexpr int z64f_ntap(int c) { return c < kZ64FineLevels ? 16 : 12 - 2 * (c - kZ64FineLevels); }
constexpr int z64f_oversampling(int c) { return c < kZ64FineLevels ? 4 : (8 << (c - kZ64FineLevels)); }
constexpr int z64f_span(int c) { return 1 << z64f_level(c); }                  // S
constexpr int z64f_win(int c) { return z64f_ntap(c) + z64f_span(c) - 1; }      // weights per lane
constexpr int kZ64FineSteps = 16;                   // wave-steps per wave and band
constexpr int kZ64FineWave = 64 * kZ64FineSteps;    // panel samples per wave and band (one partial slot per band and wave)
struct Z64FineArgs {  // one launch = one class of one table
  Z64Args z;  // n, Lf, kind, panel outputs, partial layouts; `bands` = the class's bands; nbands = their number; nblk = n /
              // kZ64FineWave partial slots per band (pb_stride, stat_nblk: the strides of the partial arrays); chunk_base =
              // the per-time plane of the launch's first row; Z = coarse samples of the class's grid level [C][lvl_bands][M]
  int32_t cls;         // class index
  int32_t nrow;        // rows (gridDim.y): row r takes bands r, r + nrow, ...
  int32_t lvl_bands;   // bands of the class's grid level (record stride of its coarse storage)
  int32_t lvl_index0;  // index of the class's first band in its level's list
  const double* w;     // [z64f_win(cls)][64] weights of the lanes
  const cplx<double>* lane_ph;  // Gabor kinds: [nbands][65] exp(2 pi i k_c (lane - e) / Lf), lane = 0..63, and at [64] the step
                                // exp(2 pi i k_c 64 / Lf) (e = 1 for the zero-padded kind: the carrier of sample tau - 1)
  const cplx<double>* wave_ph;  // [Lf / kZ64FineWave] exp(2 pi i j kZ64FineWave / Lf)
  int32_t debug;  // QI_NATIVE_DEBUG bit mask (-DQI_NATIVE_DEBUG builds, timing experiments only: 1 no panel stores, 2 no
                  // interpolation, 4 no entropy logarithm, 8 no per-time sums, 16 no carrier)
};
int launch_z64_fine(const Z64FineArgs& a, int64_t n_channels, hipStream_t st);
void z64_fine_weights(int cls, double* w /*[z64f_win(cls)][64]*/);

void zoom_weights(int level, int lane_off, float* w /*[zoom_taps(level)][64]*/);

template <typename T>
int launch_time_reduce(const T* part, T* out, int64_t C, int64_t n, int nchunk, const T* edge_time, int64_t wmax,
                       hipStream_t st);
// time reduction + finalisation of the reductions in one launch (both requested)
template <typename T>
int launch_tail(const T* part, T* out, int64_t C, int64_t n, int nchunk, const T* edge_time, int64_t wmax,
                const double* part_band, const double* part_stat, double* power_band, double* stats, int64_t B,
                int64_t nblk, int64_t nstat, const int32_t* band_slots, hipStream_t st);
template <typename T>
int launch_tail2(const T* part0, T* out0, int nchunk0, const double* part_band0, const double* part_stat0,
                 double* power_band0, double* stats0, int64_t B0, int64_t nblk0, int64_t nstat0,
                 const int32_t* band_slots0, const T* part1, T* out1, int nchunk1, const double* part_band1,
                 const double* part_stat1, double* power_band1, double* stats1, int64_t B1, int64_t nblk1, int64_t nstat1,
                 const int32_t* band_slots1, int64_t C, int64_t n, hipStream_t st);
template <typename T>
int launch_even_bins(const cplx<T>* x2, cplx<T>* x1, int64_t C, int64_t n, hipStream_t st);
template <typename T>
int launch_edge(const EdgeArgs<T>& a, int64_t C, T* edge_time, double* part_band, int64_t nblk, int64_t band_slot,
                double* part_stat, int64_t stat_slot, hipStream_t st);
int launch_band_support(const double2* F, int64_t L, int nb, double thr2, double* out, hipStream_t st);
template <typename T>
int launch_copy_window(const double2* F, cplx<T>* dst, int64_t k_lo, int64_t count, int conj, double scale,
                       int64_t row_len, hipStream_t st, double ramp = 0.0, int64_t ramp_center = 0);

}  // namespace native
}  // namespace qi
