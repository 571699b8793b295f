// Host side of libqi_tfr.so shared by its translation units: the plan object, the hipFFT plan cache, the stage profiler
// and the functions that cross files (qi_plan_build.hip: tables of a plan; qi_run.hip: launch sequences of the engines;
// qi_api.hip: the plan C ABI; qi_api_ops.hip: the plan-less C ABI).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstring>
#include <map>
#include <mutex>
#include <new>
#include <tuple>
#include <vector>

#include <cstdlib>

#include "qi_common.hpp"
#include "qi_native.hpp"

namespace qi {

struct DeviceGuard {
  int prev = -1;
  bool ok = true;
  explicit DeviceGuard(int dev) {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != dev) ok = hipSetDevice(dev) == hipSuccess;
  }
  ~DeviceGuard() {
    int cur = -1;
    if (prev >= 0 && hipGetDevice(&cur) == hipSuccess && cur != prev) (void)hipSetDevice(prev);
  }
};

// ---- hipFFT plan cache (one per qi_plan, plus a process-wide one for the plan-less STFT entry) ----
struct FftCache {
  using Key = std::tuple<int, int64_t, int64_t, int64_t>;  // hipfftType, length, batch, distance between transforms
  std::map<Key, hipfftHandle> plans;
  void* work = nullptr;
  size_t work_bytes = 0;
  std::vector<void*> retired;  // outgrown work areas, freed with the cache

  int get(hipfftType type, int64_t len, int64_t batch, hipfftHandle* out, int64_t dist = 0) {
    if (dist <= 0) dist = len;
    Key k{(int)type, len, batch, dist};
    auto it = plans.find(k);
    if (it != plans.end()) {
      *out = it->second;
      return QI_OK;
    }
    QI_REQUIRE(len > 0 && len < (1ll << 31) && batch > 0 && batch < (1ll << 31), "fft size out of range");
    hipfftHandle h;
    QI_FFT(hipfftCreate(&h));
    QI_FFT(hipfftSetAutoAllocation(h, 0));
    int nn[1] = {(int)len};
    size_t ws = 0;
    QI_REQUIRE(dist >= len && dist < (1ll << 31), "fft distance out of range");
    if (dist == len) QI_FFT(hipfftMakePlanMany(h, 1, nn, nullptr, 1, (int)len, nullptr, 1, (int)len, type, (int)batch, &ws));
    else QI_FFT(hipfftMakePlanMany(h, 1, nn, nn, 1, (int)dist, nn, 1, (int)dist, type, (int)batch, &ws));  // (in place, padded rows)
    if (ws > work_bytes) {
      // growing the shared work area happens while a plan warms up, never in steady state.  No synchronisation: the old
      // area stays allocated (transforms already queued keep using it) until the cache is cleared
      if (work) retired.push_back(work);
      work = nullptr;
      work_bytes = 0;
      QI_HIP(hipMalloc(&work, ws));
      work_bytes = ws;
      for (auto& kv : plans) QI_FFT(hipfftSetWorkArea(kv.second, work));
    }
    if (work) QI_FFT(hipfftSetWorkArea(h, work));
    plans[k] = h;
    *out = h;
    return QI_OK;
  }
  void clear() {
    for (auto& kv : plans) hipfftDestroy(kv.second);
    plans.clear();
    if (work) (void)hipFree(work);
    for (void* w : retired) (void)hipFree(w);
    retired.clear();
    work = nullptr;
    work_bytes = 0;
  }
};

// in-place double-precision transforms of `batch` rows of `len` points that start `dist` elements apart
int fft_z2z_rows(FftCache& fc, double2* data, int64_t len, int64_t dist, int64_t batch, int dir, hipStream_t st);
template <typename T>
int fft_c2c(FftCache& fc, cplx<T>* data, int64_t len, int64_t batch, int dir, hipStream_t st);
template <>
int fft_c2c<float>(FftCache& fc, float2* data, int64_t len, int64_t batch, int dir, hipStream_t st);
template <>
int fft_c2c<double>(FftCache& fc, double2* data, int64_t len, int64_t batch, int dir, hipStream_t st);
template <typename T>
int fft_r2c(FftCache& fc, T* in, cplx<T>* out, int64_t len, int64_t batch, hipStream_t st);
template <>
int fft_r2c<float>(FftCache& fc, float* in, float2* out, int64_t len, int64_t batch, hipStream_t st);
template <>
int fft_r2c<double>(FftCache& fc, double* in, double2* out, int64_t len, int64_t batch, hipStream_t st);
template <typename T>
int fft_c2r(FftCache& fc, cplx<T>* in, T* out, int64_t len, int64_t batch, hipStream_t st);
template <>
int fft_c2r<float>(FftCache& fc, float2* in, float* out, int64_t len, int64_t batch, hipStream_t st);
template <>
int fft_c2r<double>(FftCache& fc, double2* in, double* out, int64_t len, int64_t batch, hipStream_t st);

// ---- optional per-stage timing with HIP events on the caller's stream (bench.py's roofline leg) ----
struct Profiler {
  static constexpr int kStages = QI_STAGE_COUNT;
  bool on = false;
  uint32_t mask = ~0u;  // stages that are timed (bit = stage)
  struct Span {
    int a, b;  // indices into `used`
    int stage;
  };
  std::vector<Span> spans;
  std::vector<hipEvent_t> used;  // events recorded since the last read
  std::vector<hipEvent_t> pool;
  int cur = -1;
  int last = -1;  // the event that closed the previous span, while nothing has been launched since: the next
                  // span starts on it instead of recording another one (every record is a bubble in the stream)

  int record(hipStream_t st) {
    hipEvent_t e = nullptr;
    if (!pool.empty()) {
      e = pool.back();
      pool.pop_back();
    } else if (hipEventCreate(&e) != hipSuccess) {
      return -1;
    }
    (void)hipEventRecord(e, st);
    used.push_back(e);
    return (int)used.size() - 1;
  }
  int period = 1;      // time every period-th transform call only (the others run without any event)
  int64_t tick = 0;
  bool sampled = true;
  void unchain() {  // called at every transform entry point
    last = -1;
    sampled = period <= 1 || tick % period == 0;
    ++tick;
  }
  void unchain_span() { last = -1; }  // the next span is on another stream: do not share an event with it
  void begin(hipStream_t st, int stage) {
    cur = -1;
    if (!on || !sampled || !((mask >> stage) & 1u)) {
      last = -1;
      return;
    }
    cur = last >= 0 ? last : record(st);
  }
  void end(int stage, hipStream_t st) {
    last = -1;
    if (!on || cur < 0) return;
    const int b = record(st);
    if (b >= 0) {
      spans.push_back({cur, b, stage});
      last = b;
    }
    cur = -1;
  }
  void read(double* ms, int64_t* count) {
    for (int i = 0; i < kStages; ++i) {
      ms[i] = 0.0;
      count[i] = 0;
    }
    for (auto& s : spans) {
      float t = 0.f;
      if (hipEventSynchronize(used[s.b]) == hipSuccess && hipEventElapsedTime(&t, used[s.a], used[s.b]) == hipSuccess) {
        ms[s.stage] += t;
        count[s.stage] += 1;
      }
    }
    for (auto e : used) pool.push_back(e);
    used.clear();
    spans.clear();
    cur = last = -1;
  }
  void clear() {
    double ms[kStages];
    int64_t c[kStages];
    read(ms, c);
    for (auto e : pool) (void)hipEventDestroy(e);
    pool.clear();
  }
};

// hipFFT plans of the plan-less entry points (STFT, Welch, sliding STFT, ShannonFFT): per device, under one mutex
extern std::mutex g_stft_mu;
extern std::map<int, FftCache> g_stft_fft;
const char* last_error();

}  // namespace qi

using namespace qi;  // (internal header of the host translation units only)

// qi_cwt_stx: the CWT run leaves its block launch and its tail to the Stockwell run, which issues them together with
// its own (one launch each: the two block launches share the forward transform of every block)
struct TailCall {  // the arguments of one native::launch_tail
  const float* time_part = nullptr;
  float* out_time = nullptr;
  int64_t ct = 0, n = 0;
  int chunk_total = 0;
  const double* part_band = nullptr;
  const double* part_stat = nullptr;
  double* power_band = nullptr;
  double* stats = nullptr;
  int64_t B = 0, nbk = 0, stat_slots = 0;
  const int32_t* band_slots = nullptr;
};
struct FusedCarry {
  bool active = false;
  size_t ws_used = 0;  // bytes of the workspace the CWT run's scratch occupies (kept until its deferred launches ran)
  native::BlockArgs<float> blk{};
  int demod = 0;
  int64_t ct = 0;
  TailCall tail;
  bool has_zoom = false;  // the gather / coarse / interpolation launches of the CWT run are deferred as well
  native::ZoomArgs<float> zoom{};
};

struct qi_plan {
  qi_plan_desc d{};
  int64_t n = 0;
  int64_t L = 0;  // zero-padded length of the linear (styx_cwt) correlation
  void* bank[2] = {nullptr, nullptr};
  int32_t nb[2] = {0, 0};
  int64_t* d_stx_idx = nullptr;
  double* d_stx_coef = nullptr;
  int32_t nb_stx = 0;
  char* ws = nullptr;
  size_t ws_bytes = 0;
  FftCache fft;
  Profiler prof;
  // native engine: per transform kind (0 styx bank, 1 atoms bank, 2 Stockwell) the band descriptors
  struct NativeGroup {  // bands launched together: their wide members share one intermediate buffer
    int32_t first = 0, count = 0;    // range of d_bands
    int32_t gen_first = 0, ngen = 0; // range of d_gen_list (indices relative to `first`)
  };
  struct NativeTable {
    bool ready = false;
    int64_t Lf = 0;
    int32_t nbands = 0, ngen = 0, imd_slots = 0;
    native::BandDesc* d_bands = nullptr;  // grouped order
    int32_t* d_gen_list = nullptr;
    std::vector<NativeGroup> groups;
    void* Hc = nullptr;
    void* Hfull = nullptr;
    native::BandDesc* d_zoom = nullptr;  // bands produced by the zoom engine (qi_zoom.hip), by level
    int32_t* d_zoom_plane_band = nullptr;  // owner band of every coarse plane
    std::vector<std::pair<int32_t, int32_t>> h_zoom;  // (panel row, level) of the zoom bands
    std::vector<int32_t> h_rows;                      // panel rows of the pass-2 bands
    int32_t nzoom = 0, zoom_count[native::kZoomClasses] = {0, 0, 0, 0, 0, 0, 0};
    int64_t zoom_planes = 0;  // 4096-sample planes of coarse storage per record
    int zoom_max_level = 0;
    // float64 zoom (qi_zoom64.hip): the narrow-spectrum bands of a float64 table, by coarse-grid level
    native::BandDesc* d_z64 = nullptr;
    int32_t nz64 = 0, z64_first[native::kZ64Levels] = {}, z64_count[native::kZ64Levels] = {};
    // ... the bands of the three coarsest grids by class of the fine kernel (k_z64_fine: grid, interpolator length); ranges of
    // d_z64, a class's bands lie inside its level's range
    int32_t zf_first[native::kZ64FineClasses] = {}, zf_count[native::kZ64FineClasses] = {};
    double2* d_z64_lane_ph = nullptr;  // [nz64][65] carrier factors of the lanes and the step, per band of d_z64 (Gabor kinds)
    double2* d_z64_wave_ph = nullptr;  // [Lf / kZ64FineWave] carrier factors of the waves
    void release() {
      if (d_z64) (void)hipFree(d_z64);
      if (d_z64_lane_ph) (void)hipFree(d_z64_lane_ph);
      if (d_z64_wave_ph) (void)hipFree(d_z64_wave_ph);
      if (d_zoom) (void)hipFree(d_zoom);
      if (d_zoom_plane_band) (void)hipFree(d_zoom_plane_band);
      if (d_bands) (void)hipFree(d_bands);
      if (d_gen_list) (void)hipFree(d_gen_list);
      if (Hc) (void)hipFree(Hc);
      if (Hfull) (void)hipFree(Hfull);
      *this = NativeTable();
    }
  } nat[4];  // 0 styx bank (linear, Lf = 2n), 1 atoms bank (circular), 2 Stockwell, 3 styx short-atom bands (circular n)
  // block engine (qi_block.hip): bands whose atoms reach at most 1024 samples, per transform kind, in three
  // groups by reach (256, 512, 1024 samples)
  struct BlockTable {
    bool ready = false;
    int demod = 0;
    void* bank = nullptr;  // [rows][kBlk] complex filter spectra
    int32_t rows = 0;
    // Work items (one per workgroup, most expensive first) in two cuts: [0] few bands per workgroup -- many workgroups,
    // for calls with one or two records --, [1] many bands per workgroup -- fewer forward transforms of the same block
    // and fewer per-time planes, for batches that fill the chip anyway.
    struct ItemList {
      void* d_bands = nullptr;  // native::BlockBandT<T>[]: all reach groups, group by group (cut 1 keeps some bands on long blocks)
      void* d_demod_pow = nullptr;  // float64 Stockwell tables: [bands][16] demodulation factors (BlockArgs::demod_pow)
      void* d_gauss_w = nullptr;    // float64 tables: [bands][kBlk] real Gaussian filter weights (BlockArgs::gauss_w)
      std::vector<std::pair<int32_t, int32_t>> h_bands;  // (panel row, blocks) of the block bands
      native::BlockItem* d_items = nullptr;
      int32_t nitems = 0, nplanes = 0;
      int32_t nlong = 0;        // long-block items, at the front of the list
      int32_t nedge_items = 0;  // edge pieces of the split bands, appended to the item list (styx bank)
      bool edge_merged = false;  // one edge item per block for all split bands (k_block_edge) instead of one per band and block
      std::vector<native::BlockItem> h_items;  // host copy of d_items (the joint launch list is made from it)
    } var[2];
    int64_t max_blocks = 0;  // partial slots a band row needs
    void release() {
      if (bank) (void)hipFree(bank);
      for (auto& v : var) {
        if (v.d_bands) (void)hipFree(v.d_bands);
        if (v.d_demod_pow) (void)hipFree(v.d_demod_pow);
        if (v.d_gauss_w) (void)hipFree(v.d_gauss_w);
        if (v.d_items) (void)hipFree(v.d_items);
      }
      *this = BlockTable();
    }
  } blk[3];
  // qi_cwt_stx: the settled tile (records per joint tile) of the last request shape; any table change bumps table_gen
  struct {
    int64_t C = -1, tile = 0;
    unsigned flags = 0;
    uint64_t gen = 0;
  } tile_cache;
  uint64_t table_gen = 1;
  int native_block = 1;        // use the block engine for short-atom bands (0: two-pass paths only)
  // qi_cwt_stx: the CWT leaves the zero-padded spectra of the records at the start of the scratch
  const void* shared_sig = nullptr;
  int64_t shared_C = 0;
  bool shared_valid = false;
  int32_t* d_band_slots[3][2] = {};  // per table kind and item cut: partial slots each band's engine writes
  int native_zoom = 1;         // use the zoom engine for narrow-spectrum bands (0: one-pass loader of pass 2)
  int native_zoom_short = 1;      // bands oversampled >= 8 / >= 32 times on the coarsest grid use 6- / 4-tap interpolators
  int native_zoom_short_from = 4; // ... in calls (tiles) of at least this many records; below, they run with the 10-tap class
  int native_zoom_max_level = 3;  // finest coarse grid the zoom engine may use (level 4 costs more in the coarse stage than two-pass saves)
  int native_zoom_waves = 2048; // native_zoom_wgs = 0: waves each level of a zoom launch should have at least
  int native_zoom_wgs_joint = 768;   // the same budget per table in the joint launch of qi_cwt_stx (512 .. 1024 measured within 1.5 %)
  int native_zoom_wgs = 0;      // > 0: workgroups of a zoom launch, dealt to the levels by work (measured: 1.5 % slower than the per-level rule)
  float* d_zoom_w[native::kZoomClasses][2] = {};  // interpolation weights [class][lane offset]
  int native_z64 = 1;      // float64: narrow-spectrum bands at the decimated rate (coarse inverse FFT + 16-tap interpolation)
  int native_z64_levels = native::kZ64Levels;  // ... on coarse grids of Lf / 64 ... Lf / (64 >> (levels - 1)) samples
  double* d_z64_w[native::kZ64Levels] = {};  // interpolation weights per coarse-grid level
  double* d_z64f_w[native::kZ64FineClasses] = {};  // lane weights per class of the fine kernel
  double2* d_demod_t1 = nullptr;  // float64 block engine, Stockwell demodulation: exp(-2 pi i 1024 j / n), j < n / 1024
  double2* d_demod_t2 = nullptr;  // ... exp(-2 pi i j / n), j < 1024
  int native_z64_fine = 1;  // 0: every level on k_z64_interp (windows through LDS), as in round 3
  // Stockwell bands that no native engine takes at this record length (the two-pass kernels run 2^20 / 2^21 samples only): when
  // they are the last rows of the table -- the top band of an order-1 or order-2 table, whose frequency window is cut at the
  // Nyquist bins -- the native run leaves them out and a pass of the hipFFT engine over just these rows follows it
  // (run_stx_leftover); any other case hands the whole table to the hipFFT engine as before
  int32_t stx_left_lo = -1, stx_left_n = 0;
  int native_min_log2n = 14;    // shortest power-of-two record the zoom / block engines are tried on (below: the hipFFT engine).
                                // float32: 2^14 (round 5: 54 / 69 us per call against 97 / 230 on the hipFFT engine at orders 3 / 12);
                                // float64: 2^15 (at 2^14 the float64 zoom's small grids go through hipFFT launch by launch: slower)
  int native_blk64_narrow = 1;  // float64 block engine: bands whose weights above 2^-52 of the peak span <= 256 bins skip the first pass of the inverse transform (sparse_head16)
  int native_blk64_wtab = 1;  // float64 block engine: Gaussian filter weights from a plan-time table instead of sixteen exp2 per band and thread
  int native_z64_block_from = 4;  // a band that needs coarse-grid level >= this (0-based) goes to the block engine when its atom is short enough
  int native_z64_coarse = 3;  // coarse-grid levels whose coarse stage is one launch of in-LDS plane transforms (the finer ones: hipFFT)
  int native_z64_rows = 0;  // rows (band chunks = per-time planes) of the fine launches of a call together, at least
  int native_f64 = 1;      // float64 plans run on the native engines in double arithmetic (2^20 / 2^21-point transforms)
  int native_gather_fused = 1;  // zoom engine: from this many records per tile the coarse stage forms its inputs in registers
                                // (no gather launch, two passes over the coarse storage fewer, the loads of a thread's sixteen
                                // inputs batched: -35 % of that stage at 16 records, -20 % at one); 0: never
  hipStream_t side = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  // split bands of the styx bank (atoms longer than the record): zoom engine + edge pieces, see split_taper
  int native_split = 1;        // 0: such bands stay on the two-pass kernels
  int64_t native_split_e = 1024;  // taper length in samples (512, 1024 or 2048: the edge pieces' reach group)
  void* split_bank = nullptr;  // [nsplit][2][kBlk] filter spectra of the edge pieces
  int32_t* d_split_bands = nullptr;  // [nsplit] panel rows of the split bands
  std::vector<int32_t> h_split_bands;
  int32_t nsplit = 0;
  int native_blk_analytic = 1; // evaluate Gaussian filter spectra in registers instead of reading their table rows
  FusedCarry carry;
  native::DualItem* d_dual[2] = {nullptr, nullptr};  // joint block launch of qi_cwt_stx (styx + Stockwell tables) per item cut, built on first use
  int32_t n_dual[2] = {0, 0};
  int32_t n_dual_long[2] = {0, 0};  // long-block items at the front of d_dual
  bool dual_valid[2] = {false, false};
  int native_fuse = 4;         // qi_cwt_stx: 1 the block launches and the tails of the two transforms go out back to back, 2 as one
                               // launch each, 3 also the gather and the coarse stage of the zoom engine, 4 and its interpolation
  int native_blk_narrow = 1;   // block bands whose filter spectrum spans <= 256 bins skip the first radix-16 pass of the inverse transform
  int native_blk_fastw = 1;    // Gaussian weights without wrap-around logic where no alias of the filter spectrum matters
  int native_edge_merge = 1;   // tables for many records (cut 1): the split bands of a block share one edge item and its forward transforms
  int native_blk_long = 1;     // narrow Gaussian bands of the 1024-sample reach group in 8192-sample blocks (75 % of the outputs kept instead of 50 %)
  int native_blk_half = 1;     // block bands whose filter spectrum lies in the lower half of the block spectrum: eight weights, pruned first pass
  int native_tail = 1;         // time reduction and finalisation of the reductions in one launch
  int64_t native_tile = 0;     // qi_cwt_stx: at most this many records per joint tile (0: as many as the scratch holds)
  int native_blk_maxwq = 4;    // reach groups above this one (1, 2, 4) prefer the zoom engine when their spectrum fits it
  int native_blk_bands = 6;    // bands one block workgroup walks at most (each workgroup pays one forward transform)
  int native_blk_bands_batch = 12;  // the same for batches of native_blk_batch_from records or more (item cut 1)
  int native_blk_batch_from = 0;    // 0: 4 records, 8 for tables with few block bands (batch_from())
  native::EdgeBand* d_edge = nullptr;  // short-atom bands of table 3
  int32_t nedge = 0;
  int64_t edge_wmax = 0;
  int native_short = 1;  // evaluate short-atom styx bands circularly at length n (0: everything at 2n)
  int64_t native_kmax = 12288;  // widest spectrum support handled by the one-pass (pruned) loader
  int native_debug = 0;
  int native_fwd = 1;          // forward transform of the records on the native kernels (0: hipFFT)
  int native_wgs = 256;        // workgroups a pass-2 launch should have at least (band chunks are sized for it)
  unsigned long long* stamps = nullptr;  // diagnostic builds: phase cycle counters of the last pass-2 launch
  unsigned long long* blk_stamps = nullptr;  // idem, last block launch
  int native_group = 0;        // wide bands per launch group (0: all in one group)
  int native_rows = 16;        // consecutive time residues (rows) per pass-2 workgroup: 8 or 16
};

namespace qi {
namespace host {

inline size_t align_up(size_t v, size_t a = 256) { return (v + a - 1) / a * a; }
enum class Kind { Linear, Circular, Stockwell };
// Host sanitizer build (tests/sanitize: the library's host code on a stand-in HIP runtime under AddressSanitizer /
// UBSan): every region a run carves out of the plan's scratch is reported here and checked -- inside the workspace, and
// disjoint from every other live region (`shared`: the spectra a joint qi_cwt_stx tile's second run takes over from its
// first).  The product build compiles these away.
#ifdef QI_HOST_SANITIZE
void layout_begin(const qi_plan* p, const char* run, bool keep_previous);
void layout_note(const qi_plan* p, const char* what, const void* ptr, size_t bytes, bool shared = false);
#define QI_LAYOUT_BEGIN(p, run, keep) ::qi::host::layout_begin(p, run, keep)
#define QI_LAYOUT_NOTE(p, what, ptr, bytes, ...) ::qi::host::layout_note(p, what, ptr, bytes, ##__VA_ARGS__)
#else
#define QI_LAYOUT_BEGIN(p, run, keep) ((void)0)
#define QI_LAYOUT_NOTE(p, what, ptr, bytes, ...) ((void)0)
#endif
// order of the zoom classes in a table's band list (classes 6, 5 and 0 share the coarsest grid)
constexpr int kZoomListOrder[native::kZoomClasses] = {6, 5, 0, 1, 2, 3, 4};

// ---- qi_plan_build.hip: the tables of a plan ---------------------------------------------------------------------------
bool native_len_ok(int64_t Lf);
bool native_wanted(const qi_plan* p, int kind);
bool z64_table(const qi_plan* p, int table);
int64_t narrow_limit(const qi_plan* p, int table, int64_t Lf);
int batch_from(const qi_plan* p);
int zoom_class(const qi_plan* p, int table, int64_t Lf, int64_t len);
int upload_native_table(qi_plan* p, int kind, int64_t Lf, std::vector<native::BandDesc> bands);
template <typename T>
int build_native_bank(qi_plan* p, int bank, int32_t B, const double* d_par, const double* h_par, hipStream_t st);
template <typename T>
int build_bank(qi_plan* p, int bank, int32_t B, const double* d_par, hipStream_t st);
// the Stockwell table of a plan (native classification, block picks; nat[2] / blk[2]); `sigma`, `shift_index`: host [B]
int build_stx_tables(qi_plan* p, int32_t B, const int64_t* shift_index, const double* sigma, const std::vector<double>& coef);

// ---- qi_run.hip: launch sequences ----------------------------------------------------------------------------------------
template <typename T>
int run_transform(qi_plan* p, Kind kind, const void* sig_v, int64_t C, const qi_tfr_out* out, hipStream_t st);
template <typename T>
int run_native(qi_plan* p, int kind, const void* sig_v, int64_t C, const qi_tfr_out* out, hipStream_t st,
               bool may_share = false, FusedCarry* defer = nullptr, FusedCarry* finish = nullptr, size_t* probe = nullptr);
template <typename T>
int run_stx_leftover(qi_plan* p, const void* sig, int64_t C, const qi_tfr_out* out, hipStream_t st);
int run_native64(qi_plan* p, int kind, const void* sig_v, int64_t C, const qi_tfr_out* out, hipStream_t st);
int flush_carry(qi_plan* p, FusedCarry* c, hipStream_t st);

}  // namespace host
}  // namespace qi
