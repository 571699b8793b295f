// C ABI of libqi_tfr.so: plans, tiling over (channel, band) tiles, the hipFFT engine.
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

static thread_local char g_err[512] = "";

// Development switches (QI_NATIVE_*, QI_STFT_FUSED: engine ablations and launch-geometry experiments)
// are read only when QI_TUNE is set in the environment: a production process never consults them.
const char* tune_env(const char* name) {
  static const bool on = std::getenv("QI_TUNE") != nullptr;
  return on ? std::getenv(name) : nullptr;
}

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

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
  using Key = std::tuple<int, int64_t, int64_t>;  // hipfftType, length, batch
  std::map<Key, hipfftHandle> plans;
  void* work = nullptr;
  size_t work_bytes = 0;
  std::vector<void*> retired;  // outgrown work areas, freed with the cache

  int get(hipfftType type, int64_t len, int64_t batch, hipfftHandle* out) {
    Key k{(int)type, len, batch};
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
    QI_FFT(hipfftMakePlanMany(h, 1, nn, nullptr, 1, (int)len, nullptr, 1, (int)len, type, (int)batch, &ws));
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

template <typename T>
int fft_c2c(FftCache& fc, cplx<T>* data, int64_t len, int64_t batch, int dir, hipStream_t st);
template <>
int fft_c2c<float>(FftCache& fc, float2* data, int64_t len, int64_t batch, int dir, hipStream_t st) {
  hipfftHandle h;
  QI_TRY(fc.get(HIPFFT_C2C, len, batch, &h));
  QI_FFT(hipfftSetStream(h, st));
  QI_FFT(hipfftExecC2C(h, (hipfftComplex*)data, (hipfftComplex*)data, dir));
  return QI_OK;
}
template <>
int fft_c2c<double>(FftCache& fc, double2* data, int64_t len, int64_t batch, int dir, hipStream_t st) {
  hipfftHandle h;
  QI_TRY(fc.get(HIPFFT_Z2Z, len, batch, &h));
  QI_FFT(hipfftSetStream(h, st));
  QI_FFT(hipfftExecZ2Z(h, (hipfftDoubleComplex*)data, (hipfftDoubleComplex*)data, dir));
  return QI_OK;
}
template <typename T>
int fft_r2c(FftCache& fc, T* in, cplx<T>* out, int64_t len, int64_t batch, hipStream_t st);
template <>
int fft_r2c<float>(FftCache& fc, float* in, float2* out, int64_t len, int64_t batch, hipStream_t st) {
  hipfftHandle h;
  QI_TRY(fc.get(HIPFFT_R2C, len, batch, &h));
  QI_FFT(hipfftSetStream(h, st));
  QI_FFT(hipfftExecR2C(h, in, (hipfftComplex*)out));
  return QI_OK;
}
template <>
int fft_r2c<double>(FftCache& fc, double* in, double2* out, int64_t len, int64_t batch, hipStream_t st) {
  hipfftHandle h;
  QI_TRY(fc.get(HIPFFT_D2Z, len, batch, &h));
  QI_FFT(hipfftSetStream(h, st));
  QI_FFT(hipfftExecD2Z(h, in, (hipfftDoubleComplex*)out));
  return QI_OK;
}

template <typename T>
int fft_c2r(FftCache& fc, cplx<T>* in, T* out, int64_t len, int64_t batch, hipStream_t st);
template <>
int fft_c2r<float>(FftCache& fc, float2* in, float* out, int64_t len, int64_t batch, hipStream_t st) {
  hipfftHandle h;
  QI_TRY(fc.get(HIPFFT_C2R, len, batch, &h));
  QI_FFT(hipfftSetStream(h, st));
  QI_FFT(hipfftExecC2R(h, (hipfftComplex*)in, out));
  return QI_OK;
}
template <>
int fft_c2r<double>(FftCache& fc, double2* in, double* out, int64_t len, int64_t batch, hipStream_t st) {
  hipfftHandle h;
  QI_TRY(fc.get(HIPFFT_Z2D, len, batch, &h));
  QI_FFT(hipfftSetStream(h, st));
  QI_FFT(hipfftExecZ2D(h, (hipfftDoubleComplex*)in, out));
  return QI_OK;
}

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

static std::mutex g_stft_mu;
static std::map<int, FftCache> g_stft_fft;  // per device

}  // namespace qi

using namespace qi;

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
    void release() {
      if (d_z64) (void)hipFree(d_z64);
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
      std::vector<std::pair<int32_t, int32_t>> h_bands;  // (panel row, blocks) of the block bands
      native::BlockItem* d_items = nullptr;
      int32_t nitems = 0, nplanes = 0;
      int32_t nlong = 0;        // long-block items, at the front of the list
      int32_t nedge_items = 0;  // edge pieces of the split bands, appended to the item list (styx bank)
      std::vector<native::BlockItem> h_items;  // host copy of d_items (the joint launch list is made from it)
    } var[2];
    int64_t max_blocks = 0;  // partial slots a band row needs
    void release() {
      if (bank) (void)hipFree(bank);
      for (auto& v : var) {
        if (v.d_bands) (void)hipFree(v.d_bands);
        if (v.d_items) (void)hipFree(v.d_items);
      }
      *this = BlockTable();
    }
  } blk[3];
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
  // the block engine needs only the records, not their spectra: its launch runs on a side stream, concurrently with
  // the forward transform / pass 1 / coarse zoom stages, which leave most of the chip idle
  int native_overlap = 0;  // measured: no gain (the block launch fills the chip by itself), kept as an option
  int native_pair = 0;     // qi_cwt_stx: the joint block launch runs beside the zoom engine's launches (side stream).  Measured:
                           // +1.5 % at 16 records x 167 bands, -0.5 % at one record -- both kernels are bound by vector issue and
                           // by registers (3-4 waves per SIMD either way), so sharing the chip gains nothing; kept as an option
  int native_z64 = 1;      // float64: narrow-spectrum bands at the decimated rate (coarse inverse FFT + 16-tap interpolation)
  int native_z64_levels = native::kZ64Levels;  // ... on coarse grids of Lf / 64 ... Lf / (64 >> (levels - 1)) samples
  double* d_z64_w[native::kZ64Levels] = {};  // interpolation weights per coarse-grid level
  int native_f64 = 1;      // float64 plans run the two-pass kernels (exact algorithm, double arithmetic) at 2^20 / 2^21-point transforms
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
#ifdef QI_BLK_LZ
  int native_blk_lz = 3;
#else
  int native_blk_lz = 0;
#endif
  // ^ local zoom, an experiment that needs a -DQI_BLK_LZ build (bit 0: the 512-sample reach group, bit 1: the 1024-sample group): block bands of the 512- / 1024-sample reach groups with <= 256 / 128 - 16 spectrum bins from coarse samples + interpolation (qi_block.hip, lz_bands)
  float* d_lz_w = nullptr;     // [2][8][kBlkLzTaps] interpolation weights of the local zoom
  int native_blk_fastw = 1;    // Gaussian weights without wrap-around logic where no alias of the filter spectrum matters
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

namespace {

size_t align_up(size_t v, size_t a = 256) { return (v + a - 1) / a * a; }

struct Tile {
  int64_t Ct, Bt, ntb, nblk;
  size_t off_x, off_y, off_pb, off_ps;
};

// Split the plan's scratch into X [Ct][L], Y [Ct][Bt][L] and the reduction partials.
template <typename T>
int plan_tiles(const qi_plan* p, int64_t C, int64_t B, int64_t L, Tile* t) {
  const size_t row = (size_t)L * sizeof(cplx<T>);
  const int64_t nblk = ceil_div(p->n, kEpiSpan);
  const size_t part = align_up((size_t)B * nblk * 8) + align_up((size_t)B * nblk * 24);  // per channel, ntb <= B
  const size_t per_chan_full = row * (size_t)(B + 1) + part + 2048;
  int64_t Ct, Bt;
  if (per_chan_full <= p->ws_bytes) {
    Bt = B;
    Ct = (int64_t)(p->ws_bytes / per_chan_full);
    if (Ct > C) Ct = C;
  } else {
    Ct = 1;
    if (p->ws_bytes < part + 2 * row + 2048) {
      set_error("workspace of %zu bytes cannot hold one (channel, band) tile of %zu bytes", p->ws_bytes,
                part + 2 * row + 2048);
      return QI_ERR_NOMEM;
    }
    Bt = (int64_t)((p->ws_bytes - part - 2048) / row) - 1;
    if (Bt > B) Bt = B;
  }
  t->Ct = Ct;
  t->Bt = Bt;
  t->ntb = ceil_div(B, Bt);
  t->nblk = nblk;
  size_t o = 0;
  t->off_x = o;
  o += align_up(row * (size_t)Ct);
  t->off_y = o;
  o += align_up(row * (size_t)Ct * (size_t)Bt);
  t->off_pb = o;
  o += align_up((size_t)Ct * B * nblk * 8);
  t->off_ps = o;
  return QI_OK;
}

enum class Kind { Linear, Circular, Stockwell };

// order of the zoom classes in a table's band list (classes 6, 5 and 0 share the coarsest grid)
constexpr int kZoomListOrder[native::kZoomClasses] = {6, 5, 0, 1, 2, 3, 4};

template <typename T>
int run_transform(qi_plan* p, Kind kind, const void* sig_v, int64_t C, const qi_tfr_out* out, hipStream_t st) {
  const int64_t n = p->n;
  const T* sig = static_cast<const T*>(sig_v);
  int64_t L, B, off;
  const cplx<T>* H = nullptr;
  if (kind == Kind::Linear) {
    L = p->L;
    B = p->nb[QI_BANK_STYX];
    off = (n - 1) / 2;
    H = static_cast<const cplx<T>*>(p->bank[QI_BANK_STYX]);
  } else if (kind == Kind::Circular) {
    L = n;
    B = p->nb[QI_BANK_ATOMS];
    off = n / 2;
    H = static_cast<const cplx<T>*>(p->bank[QI_BANK_ATOMS]);
  } else {
    L = n;
    B = p->nb_stx;
    off = 0;
  }
  if (B <= 0) {
    set_error("plan has no band table for this transform");
    return QI_ERR_STATE;
  }
  Tile tl;
  QI_TRY(plan_tiles<T>(p, C, B, L, &tl));
  cplx<T>* X = reinterpret_cast<cplx<T>*>(p->ws + tl.off_x);
  cplx<T>* Y = reinterpret_cast<cplx<T>*>(p->ws + tl.off_y);
  double* part_band = reinterpret_cast<double*>(p->ws + tl.off_pb);
  double* part_stat = reinterpret_cast<double*>(p->ws + tl.off_ps);
  const bool want_band = out->power_band != nullptr;
  const bool want_stat = out->stats != nullptr;

  for (int64_t c0 = 0; c0 < C; c0 += tl.Ct) {
    const int64_t ct = (C - c0 < tl.Ct) ? C - c0 : tl.Ct;
    p->prof.begin(st, QI_STAGE_FORWARD);
    QI_TRY(launch_pack_pad<T>(sig + c0 * n, X, ct, n, L, st));
    QI_TRY(fft_c2c<T>(p->fft, X, L, ct, HIPFFT_FORWARD, st));
    p->prof.end(QI_STAGE_FORWARD, st);
    int64_t tb = 0;
    for (int64_t j0 = 0; j0 < B; j0 += tl.Bt, ++tb) {
      const int64_t bt = (B - j0 < tl.Bt) ? B - j0 : tl.Bt;
      p->prof.begin(st, QI_STAGE_MULTIPLY);
      if (kind == Kind::Stockwell)
        QI_TRY(launch_stx_window<T>(X, Y, ct, bt, n, p->d_stx_idx + j0, p->d_stx_coef + j0, st));
      else
        QI_TRY(launch_mul_bank<T>(X, H + j0 * L, Y, ct, bt, L, st));
      p->prof.end(QI_STAGE_MULTIPLY, st);
      p->prof.begin(st, QI_STAGE_INVERSE);
      QI_TRY(fft_c2c<T>(p->fft, Y, L, ct * bt, HIPFFT_BACKWARD, st));
      p->prof.end(QI_STAGE_INVERSE, st);
      p->prof.begin(st, QI_STAGE_EPILOGUE);
      EpiArgs<T> a{};
      a.Y = Y;
      a.L = L;
      a.n = n;
      a.off = off;
      a.Ct = ct;
      a.Bt = bt;
      a.B = B;
      a.j0 = j0;
      a.coef = out->coef ? static_cast<cplx<T>*>(out->coef) + c0 * B * n : nullptr;
      a.bits = out->bits ? static_cast<T*>(out->bits) + c0 * B * n : nullptr;
      a.power_time = out->power_time ? static_cast<T*>(out->power_time) + c0 * n : nullptr;
      a.part_band = want_band ? part_band : nullptr;
      a.part_stat = want_stat ? part_stat : nullptr;
      a.tile_b = tb;
      a.ntile_b = tl.ntb;
      a.power_scale = (T)(out->power_scale == 0.0 ? 1.0 : out->power_scale);
      a.eps = (T)(out->eps == 0.0 ? 2.220446049250313e-16 : out->eps);
      QI_TRY(launch_epilogue<T>(a, st));
      p->prof.end(QI_STAGE_EPILOGUE, st);
    }
    if (want_band || want_stat)
      QI_TRY(launch_finalize(want_band ? part_band : nullptr, want_stat ? part_stat : nullptr,
                             want_band ? static_cast<double*>(out->power_band) + c0 * B : nullptr,
                             want_stat ? static_cast<double*>(out->stats) + c0 * 4 : nullptr, ct, B, tl.nblk,
                             tl.ntb * tl.nblk, st));
    p->prof.unchain();
  }
  return QI_OK;
}

bool native_len_ok(int64_t Lf) { return Lf == (1ll << 20) || Lf == (1ll << 21); }

// does this plan run transform `kind` (0 styx bank, 1 atoms bank, 2 Stockwell) on the native engine?
bool native_wanted(const qi_plan* p, int kind) {
  if (p->d.engine == QI_ENGINE_HIPFFT) return false;
  const int64_t Lf = kind == 0 ? p->L : p->n;
  // float64: the exact two-pass kernels only (their transform lengths); no zoom / block / split approximations
  if (p->d.dtype == QI_F64) return p->native_f64 && is_pow2(p->n) && native_len_ok(Lf);
  if (is_pow2(p->n) && native_len_ok(Lf)) return true;
  // Stockwell and styx tables usually have no band for the two-pass kernels (every band is a zoom, block or split
  // band), and those engines take any power-of-two length from 2^15: the table build decides
  return kind != 1 && is_pow2(p->n) && p->n >= (1 << 15) && Lf <= (1ll << 26);
}

// Widest spectrum support (bins) of a band that keeps a compact bank row: the one-pass loader's limit, or -- float64 with
// the float64 zoom engine, which then takes every such band -- the widest band its finest grid (Lf / 4 samples) still
// oversamples four times.
bool z64_table(const qi_plan* p, int table) { return p->d.dtype == QI_F64 && p->native_z64 && table != 3; }
int64_t narrow_limit(const qi_plan* p, int table, int64_t Lf) {
  return z64_table(p, table) ? std::max<int64_t>(p->native_kmax, Lf >> (9 - p->native_z64_levels)) : p->native_kmax;
}

// Records per call from which the block launches use their batch geometry (12 bands per workgroup, long blocks): fewer
// forward transforms and per-time planes against fewer, heavier workgroups.  Measured on one box: with the 18 block bands
// of an order-3 table the batch geometry pays from 8 records (+2 % at 4 and 6 records without it), with the 35 / 46 of
// orders 6 / 12 from 4 (+2-3 % with it).  The same answer for both tables of a joint call.
int batch_from(const qi_plan* p) {
  if (p->native_blk_batch_from > 0) return p->native_blk_batch_from;
  const int32_t rows = std::max(p->blk[0].ready ? p->blk[0].rows : 0, p->blk[2].ready ? p->blk[2].rows : 0);
  return rows <= 24 ? 8 : 4;
}

// Order the bands into launch groups: the wide bands are dealt out `native_group` per group (all in one group when
// 0) so that a group's intermediate is small enough to stay in the last-level cache between pass 1 and pass 2; the
// narrow bands are spread evenly over the groups.  `bands[j].out_band` must be set by the caller.
int upload_native_table(qi_plan* p, int kind, int64_t Lf, std::vector<native::BandDesc> bands) {
  auto& t = p->nat[kind];
  if (tune_env("QI_NATIVE_VERBOSE"))
    for (const auto& d : bands)
      fprintf(stderr, "[qi plan] table %d (Lf = %lld) band %d: %s, support [%d, +%d)\n", kind, (long long)Lf, d.out_band,
              d.mode == 0 ? "one-pass loader" : (d.mode == 1 ? "two-pass" : "zoom"), d.k_lo, d.k_len);
  // bands marked for the zoom engine (mode 2 + level) leave the pass-2 list, ordered by level
  {
    std::vector<native::BandDesc> rest;
    std::vector<std::vector<native::BandDesc>> by_level(native::kZoomClasses);
    for (const auto& d : bands) {
      if (d.mode >= 2) by_level[d.mode - 2].push_back(d);
      else rest.push_back(d);
    }
    // a class with only a few bands is not worth rows of its own in the launch: they join the next class that can carry
    // them -- the 4-tap class the 6-tap one, the 6-tap class the 10-tap class of the same grid, a grid level the next
    // occupied level up (at most two up: each level doubles their coarse grid and adds window samples)
    auto join = [&](int from, int to) {
      by_level[to].insert(by_level[to].begin(), by_level[from].begin(), by_level[from].end());
      by_level[from].clear();
    };
    if (!by_level[6].empty() && by_level[6].size() < 6) join(6, 5);
    if (!by_level[5].empty() && by_level[5].size() < 6) join(5, 0);
    for (int g = 0; g + 1 < native::kZoomLevels; ++g) {
      if (by_level[g].empty() || by_level[g].size() >= 6) continue;
      for (int h = g + 1; h <= g + 2 && h < native::kZoomLevels; ++h)
        if (!by_level[h].empty()) {
          join(g, h);
          break;
        }
    }
    std::vector<native::BandDesc> zoom;
    t.h_zoom.clear();
    t.zoom_planes = 0;
    t.zoom_max_level = 0;
    for (int g = 0; g < native::kZoomClasses; ++g) t.zoom_count[g] = (int32_t)by_level[g].size();
    for (int gi = 0; gi < native::kZoomClasses; ++gi) {
      // list order: the short-interpolator classes first, next to the 10-tap class of their grid, so that a call with
      // few records can run all three as one class (kZoomListOrder)
      const int g = kZoomListOrder[gi];
      const int grid = native::zoom_grid(g);
      for (auto d : by_level[g]) {
        d.edge_slot = grid;                 // level of the band's coarse grid
        d.edge = (int32_t)t.zoom_planes;    // first plane of its coarse array
        t.zoom_planes += ((Lf / native::kZoomD) << grid) / native::kBlk;
        if (grid > t.zoom_max_level) t.zoom_max_level = grid;
        zoom.push_back(d);
        t.h_zoom.push_back({d.out_band, g});
      }
    }
    if (!zoom.empty()) {
      QI_HIP(hipMalloc((void**)&t.d_zoom, zoom.size() * sizeof(native::BandDesc)));
      QI_HIP(hipMemcpy(t.d_zoom, zoom.data(), zoom.size() * sizeof(native::BandDesc), hipMemcpyHostToDevice));
      t.nzoom = (int32_t)zoom.size();
      std::vector<int32_t> owner((size_t)t.zoom_planes);
      for (size_t j = 0; j < zoom.size(); ++j) {
        const int64_t planes = ((Lf / native::kZoomD) << zoom[j].edge_slot) / native::kBlk;
        for (int64_t q = 0; q < planes; ++q) owner[(size_t)(zoom[j].edge + q)] = (int32_t)j;
      }
      QI_HIP(hipMalloc((void**)&t.d_zoom_plane_band, owner.size() * sizeof(int32_t)));
      QI_HIP(hipMemcpy(t.d_zoom_plane_band, owner.data(), owner.size() * sizeof(int32_t), hipMemcpyHostToDevice));
      for (int g = 0; g < native::kZoomClasses; ++g)
        for (int e = 0; e < 1; ++e) {
          // (class 0 also serves the bands of classes 5 and 6 in calls with few records)
          const bool needed = t.zoom_count[g] > 0 || (g == 0 && t.zoom_count[5] + t.zoom_count[6] > 0);
          if (p->d_zoom_w[g][e] || !needed) continue;
          std::vector<float> w((size_t)64 * native::zoom_taps(g));
          native::zoom_weights(g, e, w.data());
          QI_HIP(hipMalloc((void**)&p->d_zoom_w[g][e], w.size() * sizeof(float)));
          QI_HIP(hipMemcpy(p->d_zoom_w[g][e], w.data(), w.size() * sizeof(float), hipMemcpyHostToDevice));
        }
    }
    bands.swap(rest);
  }
  if (z64_table(p, kind)) {
    // float64: every band with a compact row goes to the float64 zoom engine, on the coarsest grid that oversamples it
    // four times
    std::vector<native::BandDesc> rest;
    std::vector<std::vector<native::BandDesc>> lvl(native::kZ64Levels);
    for (const auto& d : bands) {
      int g = -1;
      if (d.mode == 0)
        for (int q = 0; q < p->native_z64_levels && g < 0; ++q)
          if (4 * (int64_t)d.k_len <= ((Lf / 64) << q)) g = q;
      if (g >= 0) lvl[g].push_back(d);
      else rest.push_back(d);
    }
    std::vector<native::BandDesc> z;
    for (int g = 0; g < native::kZ64Levels; ++g) {
      t.z64_first[g] = (int32_t)z.size();
      t.z64_count[g] = (int32_t)lvl[g].size();
      z.insert(z.end(), lvl[g].begin(), lvl[g].end());
      if (!lvl[g].empty() && !p->d_z64_w[g]) {
        const int log2d = 6 - g;
        std::vector<double> w((size_t)(1 << log2d) * native::kZ64Taps);
        native::z64_weights(log2d, w.data());
        QI_HIP(hipMalloc((void**)&p->d_z64_w[g], w.size() * sizeof(double)));
        QI_HIP(hipMemcpy(p->d_z64_w[g], w.data(), w.size() * sizeof(double), hipMemcpyHostToDevice));
      }
    }
    t.nz64 = (int32_t)z.size();
    if (!z.empty()) {
      QI_HIP(hipMalloc((void**)&t.d_z64, z.size() * sizeof(native::BandDesc)));
      QI_HIP(hipMemcpy(t.d_z64, z.data(), z.size() * sizeof(native::BandDesc), hipMemcpyHostToDevice));
    }
    if (tune_env("QI_NATIVE_VERBOSE"))
      fprintf(stderr, "[qi plan] table %d: float64 zoom bands per level %d %d %d %d %d, two-pass bands %zu\n", kind, t.z64_count[0],
              t.z64_count[1], t.z64_count[2], t.z64_count[3], t.z64_count[4], rest.size());
    bands.swap(rest);
  }
  t.h_rows.clear();
  for (const auto& d : bands) t.h_rows.push_back(d.out_band);
  if (bands.empty()) {  // every band is produced by the block / zoom engines: an empty but valid table
    t.Lf = Lf;
    t.ready = true;
    return QI_OK;
  }
  std::vector<int32_t> wide, narrow;
  for (size_t j = 0; j < bands.size(); ++j) (bands[j].mode == 1 ? wide : narrow).push_back((int32_t)j);
  const int32_t per = p->native_group > 0 ? p->native_group : (int32_t)wide.size();
  const int32_t ngroups = wide.empty() ? 1 : (int32_t)ceil_div((int64_t)wide.size(), per);
  std::vector<native::BandDesc> ordered;
  std::vector<int32_t> gen;
  t.groups.clear();
  size_t wi = 0, ni = 0;
  for (int32_t g = 0; g < ngroups; ++g) {
    qi_plan::NativeGroup grp;
    grp.first = (int32_t)ordered.size();
    grp.gen_first = (int32_t)gen.size();
    int32_t slot = 0;
    for (int32_t q = 0; q < per && wi < wide.size(); ++q, ++wi) {
      native::BandDesc d = bands[wide[wi]];
      d.gen_slot = slot++;
      gen.push_back((int32_t)ordered.size() - grp.first);
      ordered.push_back(d);
    }
    const size_t share = (narrow.size() * (size_t)(g + 1)) / (size_t)ngroups;
    for (; ni < share; ++ni) ordered.push_back(bands[narrow[ni]]);
    grp.count = (int32_t)ordered.size() - grp.first;
    grp.ngen = (int32_t)gen.size() - grp.gen_first;
    if (grp.count > 0) t.groups.push_back(grp);
  }
  QI_HIP(hipMalloc((void**)&t.d_bands, ordered.size() * sizeof(native::BandDesc)));
  QI_HIP(hipMemcpy(t.d_bands, ordered.data(), ordered.size() * sizeof(native::BandDesc), hipMemcpyHostToDevice));
  if (!gen.empty()) {
    QI_HIP(hipMalloc((void**)&t.d_gen_list, gen.size() * sizeof(int32_t)));
    QI_HIP(hipMemcpy(t.d_gen_list, gen.data(), gen.size() * sizeof(int32_t), hipMemcpyHostToDevice));
  }
  t.Lf = Lf;
  t.nbands = (int32_t)bands.size();
  t.ngen = (int32_t)wide.size();
  t.imd_slots = wide.empty() ? 0 : (per < (int32_t)wide.size() ? per : (int32_t)wide.size());
  t.ready = true;
  return QI_OK;
}

// Support analysis of `count` atom spectra starting at band j0 (rows built in `circular` or linear form).
int analyse_support(qi_plan* p, int circular, int64_t L, int32_t B, int32_t j0, int32_t count, const double* d_par,
                    std::vector<double>* sup, hipStream_t st, double taper_e = 0.0) {
  const size_t row64 = (size_t)L * sizeof(double2);
  int64_t chunk = (int64_t)((p->ws_bytes - 4096) / row64);
  if (chunk < 1) {
    set_error("workspace too small to build one bank row (%zu bytes needed)", row64);
    return QI_ERR_NOMEM;
  }
  double2* rows = reinterpret_cast<double2*>(p->ws);
  double* d_sup = nullptr;
  QI_HIP(hipMalloc((void**)&d_sup, (size_t)count * 3 * sizeof(double)));
  // |H| below 2^-30 of the row maximum is dropped (float32 engines); float64 keeps everything above 2^-50
  const double thr2 = p->d.dtype == QI_F64 ? std::ldexp(1.0, -100) : std::ldexp(1.0, -60);
  int rc = QI_OK;
  for (int32_t q = 0; q < count && rc == QI_OK; q += (int32_t)chunk) {
    const int nbk = (count - q < chunk) ? count - q : (int)chunk;
    rc = launch_bank_rows(rows, p->n, L, circular, d_par, d_par + B, d_par + 2 * B, d_par + 3 * B, j0 + q, nbk, st,
                          taper_e);
    if (rc == QI_OK) rc = fft_c2c<double>(p->fft, rows, L, nbk, HIPFFT_FORWARD, st);
    if (rc == QI_OK) rc = native::launch_band_support(rows, L, nbk, thr2, d_sup + (size_t)q * 3, st);
  }
  sup->assign((size_t)count * 3, 0.0);
  if (rc == QI_OK && (hipStreamSynchronize(st) != hipSuccess ||
                      hipMemcpy(sup->data(), d_sup, sup->size() * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess)) {
    set_error("support analysis failed: %s", hipGetErrorString(hipGetLastError()));
    rc = QI_ERR_HIP;
  }
  (void)hipFree(d_sup);
  return rc;
}

// Fill the compact / full-row banks of table `t` for the bands listed in `ids` (global band ids; descriptors in
// `bands`, same order) from freshly built float64 spectra.
template <typename T>
int fill_native_bank(qi_plan* p, qi_plan::NativeTable& t, int circular, int64_t L, int32_t B,
                     const std::vector<int32_t>& ids, const std::vector<native::BandDesc>& bands, const double* d_par,
                     hipStream_t st) {
  const size_t row64 = (size_t)L * sizeof(double2);
  int64_t chunk = (int64_t)((p->ws_bytes - 4096) / row64);
  double2* rows = reinterpret_cast<double2*>(p->ws);
  size_t q = 0;
  while (q < ids.size()) {
    // a run of consecutive band ids, at most `chunk` long (split bands -- tapered rows -- apart from the others)
    size_t r = q + 1;
    const bool tapered = bands[q].add_row != 0;
    while (r < ids.size() && ids[r] == ids[r - 1] + 1 && (int64_t)(r - q) < chunk && (bands[r].add_row != 0) == tapered) ++r;
    const int nbk = (int)(r - q);
    QI_TRY(launch_bank_rows(rows, p->n, L, circular, d_par, d_par + B, d_par + 2 * B, d_par + 3 * B, ids[q], nbk, st,
                            tapered ? (double)p->native_split_e : 0.0));
    QI_TRY(fft_c2c<double>(p->fft, rows, L, nbk, HIPFFT_FORWARD, st));
    for (int jj = 0; jj < nbk; ++jj) {
      const native::BandDesc& d = bands[q + jj];
      if (d.mode != 1) {
        // zoom bands of the linear (styx) table: panel sample t is full-length sample t + n/2 - 1; the odd sample is a
        // phase ramp on the band's baseband bins, exp(-2 pi i (k - k_c) / L), folded into the compact bank here
        const double ramp = (d.mode >= 2 && !circular) ? -1.0 / (double)L : 0.0;
        QI_TRY(native::launch_copy_window<T>(rows + (int64_t)jj * L, static_cast<cplx<T>*>(t.Hc) + d.src_off, d.k_lo,
                                              d.k_len, circular, 1.0 / (double)L, L, st, ramp, d.k_len / 2));
      }
      else
        QI_TRY(native::launch_copy_window<T>(rows + (int64_t)jj * L,
                                              static_cast<cplx<T>*>(t.Hfull) + (int64_t)d.bank_row * L, 0, L, circular,
                                              1.0 / (double)L, L, st));
    }
    q = r;
  }
  return QI_OK;
}

// ---- block engine tables ------------------------------------------------------------------------------------------
struct BlockPick {
  int32_t band;   // panel row
  int wq;         // reach group: taps within 256 * wq samples (1, 2 or 4)
  int64_t shift;  // Stockwell shift index (0 for Gabor banks)
  // analytic Gaussian filter spectrum (0: read the table row): weight(k) = amp exp2(-(cw (k - kappa))^2)
  int analytic = 0;
  double kappa = 0.0, cw = 0.0, amp = 0.0;
};
int block_group_of(double reach) { return reach <= 256.0 ? 1 : (reach <= 512.0 ? 2 : (reach <= 1024.0 ? 4 : 0)); }

// `taps` holds one 4096-sample circular-convolution kernel per pick (float64, on the device, same order):
// transform them, convert to the engine's precision and upload the per-group band lists.
template <typename T>
int finish_block_table(qi_plan* p, int kind, int demod, const std::vector<BlockPick>& picks, double2* taps,
                       hipStream_t st) {
  auto& bt = p->blk[kind];
  bt.release();
  p->dual_valid[0] = p->dual_valid[1] = false;
  if (picks.empty()) return QI_OK;
  const int32_t rows = (int32_t)picks.size();
  QI_TRY(fft_c2c<double>(p->fft, taps, native::kBlk, rows, HIPFFT_FORWARD, st));
  if (!demod) QI_TRY(native::launch_block_rotate_rows(taps, rows, st));
  QI_HIP(hipMalloc(&bt.bank, (size_t)rows * native::kBlk * sizeof(cplx<T>)));
  QI_TRY(launch_bank_convert<T>(taps, static_cast<cplx<T>*>(bt.bank), (int64_t)rows * native::kBlk, 0,
                                1.0 / (double)native::kBlk, st));
  bt.rows = rows;
  bt.demod = demod;
  // reach groups: taps within 256, 512, 1024 samples (4096-sample blocks), and the long blocks (8192 samples) for the
  // narrow Gaussian bands of the 1024-sample group whose spectrum lies in the lower half of the 8192-bin grid
  // ... and the local zoom items (narrow Gaussian bands of the 512- and 1024-sample groups at the decimated rate)
  constexpr int NG = 6;
  const int wqs[NG] = {1, 2, 4, native::kBlkLongWq, native::kBlkLzA, native::kBlkLzB};
  // local zoom: the weights above 2^-30 of the peak within 4096 / (8 D) bins of the baseband centre (the band is then
  // oversampled >= 4 times on the coarse grid); the centre is the band's own for a Stockwell band and the next multiple of
  // 16 bins for a Gabor band (<= 8 bins off)
  // (float64 tables: Gaussian weights in double from every bin -- the shortcuts below drop weights under 2^-30 of the peak)
  constexpr bool F64 = sizeof(T) == 8;
  const double drop_bits = F64 ? 52.0 : 30.0;
  auto lz_kind = [&](const BlockPick& pk) -> int {
    if (F64) return 0;
    if (!p->native_blk_lz || !p->native_blk_analytic || !p->native_blk_narrow || !pk.analytic) return 0;
    const double half = std::ceil(std::sqrt(30.0) / pk.cw);
    if (2.0 * half + 2.0 > 256.0) return 0;
    const double margin = demod ? 2.0 : 10.0;
    if ((p->native_blk_lz & 1) && pk.wq == 2 && half + margin <= 128.0) return native::kBlkLzA;
    if ((p->native_blk_lz & 2) && pk.wq == 4 && half + margin <= 64.0) return native::kBlkLzB;
    return 0;
  };
  if (p->native_blk_lz && !p->d_lz_w) {
    std::vector<float> wts(2 * 8 * native::kBlkLzTaps, 0.0f);
    native::lz_weights(2, wts.data());
    native::lz_weights(3, wts.data() + 8 * native::kBlkLzTaps);
    QI_HIP(hipMalloc((void**)&p->d_lz_w, wts.size() * sizeof(float)));
    QI_HIP(hipMemcpy(p->d_lz_w, wts.data(), wts.size() * sizeof(float), hipMemcpyHostToDevice));
  }
  // first bin of the 256-bin window of a long band: centred on the band, kept inside the lower half of the 8192-bin grid
  // (the half a long block holds)
  auto long_window = [&](const BlockPick& pk) {
    return std::min<int64_t>(std::max<int64_t>((int64_t)std::llround(2.0 * pk.kappa) - 128, 0), native::kBlk - 256);
  };
  auto long_ok = [&](const BlockPick& pk, int cut) {
    if (F64 || lz_kind(pk)) return false;
    if (cut == 0) return false;  // few records: the long blocks' own launch would cost more than the blocks save
    if (!p->native_blk_long || !p->native_blk_analytic || !p->native_blk_narrow || pk.wq != 4 || !pk.analytic) return false;
    if (p->n < 4 * native::kBlkLong) return false;
    const double half8 = std::ceil(std::sqrt(30.0) / (0.5 * pk.cw));  // weights >= 2^-30 of the peak on the 8192-bin grid
    const int64_t klo8 = long_window(pk);
    return 2.0 * pk.kappa - half8 - 1.0 >= (double)klo8 && 2.0 * pk.kappa + half8 + 1.0 <= (double)(klo8 + 255);
  };
  int64_t split_blocks = 0;
  if (kind == 0 && p->nsplit > 0) {
    split_blocks = ceil_div(p->n, native::block_valid((int)(p->native_split_e / 512)));
    if (split_blocks > bt.max_blocks) bt.max_blocks = split_blocks;
    for (auto& il : bt.var)
      for (int32_t sb = 0; sb < p->nsplit; ++sb) il.h_bands.push_back({p->h_split_bands[sb], (int32_t)split_blocks});
  }
  for (int v = 0; v < 2; ++v) {
    auto& il = bt.var[v];
    std::vector<native::BlockBandT<T>> list;
    int32_t group_first[NG] = {0, 0, 0, 0}, group_count[NG] = {0, 0, 0, 0};
    for (int g = 0; g < NG; ++g) {
      const int32_t first = (int32_t)list.size();
      for (int32_t r = 0; r < rows; ++r) {
        const bool is_long = long_ok(picks[r], v);
        const int lz = lz_kind(picks[r]);
        const int home = lz ? lz : (is_long ? native::kBlkLongWq : picks[r].wq);  // the group that takes this band
        if (home != wqs[g]) continue;
        native::BlockBandT<T> b;
        memset(&b, 0, sizeof(b));
        b.out_band = picks[r].band;
        b.bank_row = r;
        b.shift = (int32_t)picks[r].shift;
        b.analytic = p->native_blk_analytic ? picks[r].analytic : 0;
        const double grid = is_long ? 2.0 : 1.0;  // the band on the 8192-bin grid of a long block: twice the bins
        const double kappa = grid * picks[r].kappa, cw = picks[r].cw / grid;
        b.kappa_int = (int32_t)std::floor(kappa);
        b.kappa_frac = (T)(kappa - std::floor(kappa));
        b.cw = (T)cw;
        b.amp = (T)(picks[r].amp / grid);
        // weights >= 2^-30 of the peak: |cw dk| <= sqrt(30) (float64: 2^-52)
        const double half = std::ceil(std::sqrt(drop_bits) / cw);
        if (F64 && !b.analytic) {
          set_error("block engine: float64 tables take analytic (Gaussian) bands only");
          return QI_ERR_STATE;
        }
        if (!F64 && b.analytic && p->native_blk_narrow && 2.0 * half + 2.0 <= 256.0) {
          b.narrow = 1;
          b.klo = is_long ? (int32_t)long_window(picks[r])
                          : (int32_t)((((int64_t)std::llround(kappa) - 128) % native::kBlk + native::kBlk) % native::kBlk);
          const int ba = b.klo >> 8;
          b.rot_a[0] = (T)std::cos(2.0 * M_PI * ba / 16.0);
          b.rot_a[1] = (T)std::sin(2.0 * M_PI * ba / 16.0);
          b.rot_b[0] = (T)std::cos(2.0 * M_PI * ((ba + 1) & 15) / 16.0);
          b.rot_b[1] = (T)std::sin(2.0 * M_PI * ((ba + 1) & 15) / 16.0);
          b.rot8_a[0] = (T)std::cos(M_PI * ba / 16.0);  // exp(2 pi i 256 b / 8192)
          b.rot8_a[1] = (T)std::sin(M_PI * ba / 16.0);
          b.rot8_b[0] = (T)std::cos(M_PI * (ba + 1) / 16.0);
          b.rot8_b[1] = (T)std::sin(M_PI * (ba + 1) / 16.0);
        } else if (!F64 && b.analytic && p->native_blk_half && kappa - half - 1.0 >= 0.0 && kappa + half + 1.0 < (double)(native::kBlk / 2)) {
          b.narrow = 2;  // every weight above 2^-30 of the peak lies in the lower half of the block spectrum
        }
        if (b.analytic && p->native_blk_fastw && b.amp > (T)0 && kappa - half - 1.0 >= 0.0 && kappa + half + 1.0 < (double)native::kBlk) {
          b.nowrap = 1;
          b.la = (T)std::log2(picks[r].amp / grid);
        }
        if (lz) {
          if (b.narrow != 1) {
            set_error("block engine: a local-zoom band without a 256-bin window");
            return QI_ERR_STATE;
          }
          b.narrow = 3;
          b.kc = (int32_t)(((16 * (int64_t)std::llround(kappa / 16.0)) % native::kBlk + native::kBlk) % native::kBlk);
          b.rot_lz[0] = (T)std::cos(2.0 * M_PI * (double)b.kc / (double)native::kBlk);
          b.rot_lz[1] = (T)std::sin(2.0 * M_PI * (double)b.kc / (double)native::kBlk);
        }
        for (int k = 0; k < 4; ++k) {
          // r^(2^k), r = exp(-2 pi i idx 256 / n), from the exact integer phase
          const int64_t m = (int64_t)(((__int128)picks[r].shift * 256 * (1 << k)) % p->n);
          const double ang = -2.0 * M_PI * (double)m / (double)p->n;
          b.rot[2 * k] = (T)std::cos(ang);
          b.rot[2 * k + 1] = (T)std::sin(ang);
        }
        {
          const int64_t m1 = picks[r].shift % p->n;  // one sample: the odd sample of a long block's pair
          b.rot1[0] = (T)std::cos(-2.0 * M_PI * (double)m1 / (double)p->n);
          b.rot1[1] = (T)std::sin(-2.0 * M_PI * (double)m1 / (double)p->n);
        }
        list.push_back(b);
      }
      group_first[g] = first;
      group_count[g] = (int32_t)list.size() - first;
      if (group_count[g] == 0) continue;
      const int64_t nblocks = ceil_div(p->n, native::block_valid(wqs[g]));
      if (nblocks > bt.max_blocks) bt.max_blocks = nblocks;
      for (int32_t q = first; q < (int32_t)list.size(); ++q) il.h_bands.push_back({list[q].out_band, (int32_t)nblocks});
    }
    std::vector<native::BlockItem> items;
    for (int g = 0; g < NG; ++g) {
      const int32_t first = group_first[g], count = group_count[g];
      if (count == 0) continue;
      // the group's bands are dealt to `nchunk` workgroups per block (each pays one forward transform of the block); a
      // local-zoom workgroup takes its bands D at a time (one run of the LDS passes per D bands)
      int per_wg = v == 0 ? p->native_blk_bands : p->native_blk_bands_batch;
      if (const int l2d = native::lz_log2d(wqs[g])) per_wg = v == 0 ? (1 << l2d) : (l2d == 2 ? 12 : 8);
      const int32_t nchunk = (int32_t)ceil_div(count, per_wg);
      const int64_t nblocks = ceil_div(p->n, native::block_valid(wqs[g]));
      if (tune_env("QI_NATIVE_VERBOSE"))
        fprintf(stderr, "[qi plan] block table %d cut %d, reach <= %d%s: %d bands (%d analytic, %d narrow, %d half) in %d workgroups x %lld blocks\n", kind, v,
                g == 3 ? 1024 : 256 * (wqs[g] & 15), g == 3 ? " (8192-sample blocks)" : (g > 3 ? " (local zoom)" : ""), count,
                (int)std::count_if(list.begin() + first, list.begin() + first + count, [](const native::BlockBandT<T>& b) { return b.analytic != 0; }),
                (int)std::count_if(list.begin() + first, list.begin() + first + count, [](const native::BlockBandT<T>& b) { return b.narrow == 1; }),
                (int)std::count_if(list.begin() + first, list.begin() + first + count, [](const native::BlockBandT<T>& b) { return b.narrow == 2; }),
                nchunk, (long long)nblocks);
      for (int32_t c = 0; c < nchunk; ++c) {
        const int32_t lo = first + (int32_t)((int64_t)count * c / nchunk);
        const int32_t hi = first + (int32_t)((int64_t)count * (c + 1) / nchunk);
        for (int64_t b = 0; b < nblocks; ++b) {
          native::BlockItem it;
          it.wq = wqs[g];
          it.block = (int32_t)b;
          it.band_first = lo;
          it.band_count = hi - lo;
          it.plane = il.nplanes;
          it.stat_slot = 0;
          items.push_back(it);
        }
        il.nplanes += 1;
      }
    }
    std::stable_sort(items.begin(), items.end(), [](const native::BlockItem& x, const native::BlockItem& y) {
      const bool lx = x.wq == native::kBlkLongWq, ly = y.wq == native::kBlkLongWq;
      return lx != ly ? lx : x.band_count > y.band_count;
    });
    for (size_t i = 0; i < items.size(); ++i) items[i].stat_slot = (int32_t)i;
    il.nitems = (int32_t)items.size();
    il.nlong = (int32_t)std::count_if(items.begin(), items.end(), [](const native::BlockItem& x) { return x.wq == native::kBlkLongWq; });
    if (kind == 0 && p->nsplit > 0) {
      // the edge items of the split bands ride at the end of the launch (light items: they fill its tail); each split
      // band has a per-time plane and one partial slot per block like the other bands of the launch
      const int wq = (int)(p->native_split_e / 512);
      for (int32_t sb = 0; sb < p->nsplit; ++sb) {
        for (int64_t b = 0; b < split_blocks; ++b)
          items.push_back({-wq, (int32_t)b, sb, 0, il.nplanes, (int32_t)items.size()});
        il.nplanes += 1;
      }
      il.nedge_items = (int32_t)items.size() - il.nitems;
    }
    il.h_items = items;
    QI_HIP(hipMalloc((void**)&il.d_bands, list.size() * sizeof(native::BlockBandT<T>)));
    QI_HIP(hipMemcpy(il.d_bands, list.data(), list.size() * sizeof(native::BlockBandT<T>), hipMemcpyHostToDevice));
    QI_HIP(hipMalloc((void**)&il.d_items, items.size() * sizeof(native::BlockItem)));
    QI_HIP(hipMemcpy(il.d_items, items.data(), items.size() * sizeof(native::BlockItem), hipMemcpyHostToDevice));
  }
  QI_HIP(hipStreamSynchronize(st));
  bt.ready = true;
  return QI_OK;
}

// Gabor bands (styx bank): taps straight from the atom parameters.
template <typename T>
int build_block_gabor(qi_plan* p, int kind, int32_t B, const std::vector<BlockPick>& picks, const double* d_par,
                      hipStream_t st) {
  if (picks.empty()) {
    p->blk[kind].release();
    return QI_OK;
  }
  double2* taps = reinterpret_cast<double2*>(p->ws);
  if (p->ws_bytes < picks.size() * native::kBlk * sizeof(double2)) {
    set_error("workspace too small for the block-engine taps");
    return QI_ERR_NOMEM;
  }
  const int wqs[3] = {1, 2, 4};
  int32_t* d_ids = nullptr;
  QI_HIP(hipMalloc((void**)&d_ids, picks.size() * sizeof(int32_t)));
  int rc = QI_OK;
  size_t r = 0;
  // picks are ordered by group, so each group is one run of rows
  for (int g = 0; g < 3 && rc == QI_OK; ++g) {
    std::vector<int32_t> ids;
    for (const auto& pk : picks)
      if (pk.wq == wqs[g]) ids.push_back(pk.band);
    if (ids.empty()) continue;
    if (hipMemcpy(d_ids + r, ids.data(), ids.size() * sizeof(int32_t), hipMemcpyHostToDevice) != hipSuccess) {
      set_error("hipMemcpy of block band ids failed");
      rc = QI_ERR_HIP;
      break;
    }
    rc = native::launch_block_taps_gabor(taps + r * native::kBlk, 256 * wqs[g], d_par, B, d_ids + r, (int)ids.size(), st);
    r += ids.size();
  }
  if (rc == QI_OK) rc = finish_block_table<T>(p, kind, 0, picks, taps, st);
  (void)hipStreamSynchronize(st);
  (void)hipFree(d_ids);
  return rc;
}

// Stockwell bands: the time-domain kernel is the inverse transform of the band's Gaussian window, computed as the
// reference defines it (on the n signed FFT bins) and modulated to the band's shift index.
template <typename T>
int build_block_stx(qi_plan* p, const std::vector<BlockPick>& picks, const std::vector<double>& coef, hipStream_t st) {
  if (picks.empty()) {
    p->blk[2].release();
    return QI_OK;
  }
  const size_t need = ((size_t)p->n + picks.size() * native::kBlk) * sizeof(double2);
  if (p->ws_bytes < need) {
    set_error("workspace too small for the block-engine taps (%zu bytes needed)", need);
    return QI_ERR_NOMEM;
  }
  double2* row = reinterpret_cast<double2*>(p->ws);
  double2* taps = row + p->n;
  for (size_t r = 0; r < picks.size(); ++r) {
    QI_TRY(native::launch_stx_window_row(row, p->n, coef[picks[r].band], st));
    QI_TRY(fft_c2c<double>(p->fft, row, p->n, 1, HIPFFT_BACKWARD, st));
    QI_TRY(native::launch_block_taps_stx(taps + r * native::kBlk, 256 * picks[r].wq, row, p->n, picks[r].shift, st));
  }
  return finish_block_table<T>(p, 2, 1, picks, taps, st);
}

// Zoom engine level of a band with `len` occupied bins out of Lf (-1: not eligible): the coarsest grid
// M_g = (Lf / 64) << g on which the band is oversampled at least 4 times.
int zoom_class(const qi_plan* p, int table, int64_t Lf, int64_t len) {
  if (!p->native_zoom || table == 3 || len <= 0 || Lf % native::kZoomD != 0) return -1;
  const int64_t M0 = Lf / native::kZoomD;
  if (!is_pow2(M0)) return -1;
  // the coarse stage works in 4096-point planes: a short record starts at the first grid level that fills one
  int g_min = 0;
  while ((M0 << g_min) < native::kBlk) ++g_min;
  for (int g = g_min; g < native::kZoomLevels; ++g) {
    if (p->n % ((int64_t)native::kZoomD * native::zoom_steps(g) * 4) != 0) return -1;
    if (native::kZoomOversample * len <= (M0 << g)) {
      // (the finest grid costs more in the coarse stage than the two-pass kernels save -- where those exist; at other
      // lengths it keeps the table off the hipFFT engine)
      if (g > p->native_zoom_max_level && native_len_ok(Lf)) return -1;
      // on the coarsest grid the band may be oversampled far more than 4 times: shorter interpolators (classes 5, 6)
      if (g == 0 && p->native_zoom_short) {
        if ((int64_t)native::zoom_design_oversampling(6) * len <= M0) return 6;
        if ((int64_t)native::zoom_design_oversampling(5) * len <= M0) return 5;
      }
      return g;
    }
  }
  return -1;
}

// Classify bands by spectrum support, allocate and fill one table.
template <typename T>
int make_native_table(qi_plan* p, int table, int circular, int64_t L, int32_t B, const std::vector<int32_t>& ids,
                      const std::vector<double>& sup /*[ids][3]*/, const std::vector<int32_t>& edge_w,
                      const double* d_par, hipStream_t st, const std::vector<int32_t>& add_row = {}) {
  std::vector<native::BandDesc> bands(ids.size());
  int64_t compact = 0;
  int32_t ngen = 0;
  for (size_t q = 0; q < ids.size(); ++q) {
    native::BandDesc& d = bands[q];
    memset(&d, 0, sizeof(d));
    const int64_t lo = (int64_t)sup[3 * q + 1], hi = (int64_t)sup[3 * q + 2];
    const int64_t len = hi >= lo ? hi - lo + 1 : 0;
    d.out_band = ids[q];
    d.edge = edge_w.empty() ? 0 : edge_w[q];
    d.edge_slot = (int32_t)q;  // the edge list is in the order of `ids`
    d.add_row = add_row.empty() ? 0 : add_row[q];
    const int zc = zoom_class(p, table, L, len);
    if (zc >= 0 || (len > 0 && len <= narrow_limit(p, table, L))) {
      d.mode = zc >= 0 ? 2 + zc : 0;  // 0: one-pass loader of pass 2; 2 + c: zoom engine, class c
      d.k_lo = (int32_t)lo;
      d.k_len = (int32_t)len;
      d.src_off = compact;
      compact += len;
    } else {
      d.mode = 1;
      d.bank_row = ngen++;
    }
  }
  auto& t = p->nat[table];
  t.release();
  if (ids.empty()) {  // every band is produced elsewhere (block engine): an empty but valid table
    t.Lf = L;
    t.ready = table != 3;
    return QI_OK;
  }
  if (compact > 0) QI_HIP(hipMalloc(&t.Hc, (size_t)compact * sizeof(cplx<T>)));
  if (ngen > 0) QI_HIP(hipMalloc(&t.Hfull, (size_t)ngen * L * sizeof(cplx<T>)));
  QI_TRY(fill_native_bank<T>(p, t, circular, L, B, ids, bands, d_par, st));
  return upload_native_table(p, table, L, bands);
}

// Native bank.  Every atom spectrum is analysed for its support: a narrow one keeps a compact window (one-pass
// "pruned" bands), a wide one its full row.  For the styx bank (zero-padded linear correlation, Lf = 2n) a band whose
// spectrum is wide but whose ATOM is short in time is not run at 2n at all: it is evaluated as a circular
// correlation of length n (half the bank row, half the intermediate, no discarded outputs) and its first / last W
// samples -- the only ones where circular and linear differ -- are corrected by k_edge_fix.
template <typename T>
int build_native_bank(qi_plan* p, int bank, int32_t B, const double* d_par, const double* h_par, hipStream_t st) {
  const int64_t n = p->n;
  const int circular = bank == QI_BANK_ATOMS;
  const int64_t L = circular ? n : p->L;
  std::vector<double> sup;
  QI_TRY(analyse_support(p, circular, L, B, 0, B, d_par, &sup, st));
  std::vector<int32_t> keep, shorts, short_w;
  std::vector<BlockPick> picks;
  const bool can_short = !circular && p->native_short && native_len_ok(n) && is_pow2(n);
  const bool can_block = !circular && p->native_block && is_pow2(n) && n >= 4 * native::kBlk;
  for (int32_t j = 0; j < B; ++j) {
    const int64_t lo = (int64_t)sup[3 * j + 1], hi = (int64_t)sup[3 * j + 2];
    const int64_t len = hi >= lo ? hi - lo + 1 : 0;
    // taps with |x| <= w are above 2^-30 of the atom's peak: exp(-p_re x^2) >= 2^-30 (float64: 2^-52)
    const double w = std::ceil(std::sqrt((p->d.dtype == QI_F64 ? 52.0 : 30.0) * M_LN2 / h_par[j])) + 1.0;
    // (a band of the widest reach groups -- half of each 4096-sample block is overlap there -- goes to the zoom
    // engine instead when its spectrum fits one of its grids)
    // (float64: a band the float64 zoom takes -- support within Lf / 16 bins -- stays there)
    const bool z64_first = p->d.dtype == QI_F64 && z64_table(p, bank) && len > 0 && len <= narrow_limit(p, bank, L);
    if (can_block && block_group_of(w) > 0 && !z64_first &&
        !(block_group_of(w) > p->native_blk_maxwq && zoom_class(p, bank, L, len) >= 0)) {
      BlockPick pk{j, block_group_of(w), 0};
      const double p_re = h_par[j], p_im = h_par[B + j], om = h_par[2 * B + j], am = h_par[3 * B + j];
      // a pure Gabor atom at least 2.75 samples wide (no alias of its Gaussian spectrum above 1e-16) with its centre
      // frequency inside (0, pi): its 4096-point filter spectrum is amp sqrt(pi / p) exp(-d^2 / 4p) exp(-i theta / 2)
      if (p_im == 0.0 && p_re > 0.0 && p_re <= 1.0 / (2.0 * 2.75 * 2.75) && om > 0.0 && om < M_PI) {
        pk.analytic = 1;
        pk.kappa = om * (double)native::kBlk / (2.0 * M_PI);
        pk.cw = (2.0 * M_PI / (double)native::kBlk) * std::sqrt(M_LOG2E / (4.0 * p_re));
        pk.amp = am * std::sqrt(M_PI / p_re) / (double)native::kBlk;
      }
      if (p->d.dtype == QI_F64 && !pk.analytic) {  // (the float64 block kernels evaluate Gaussians only)
        if (can_short && w <= 8192.0 && w < (double)n / 8) {
          shorts.push_back(j);
          short_w.push_back((int32_t)w);
        } else {
          keep.push_back(j);
        }
        continue;
      }
      picks.push_back(pk);
    } else if (can_short && zoom_class(p, bank, L, len) < 0 && !(len > 0 && len <= narrow_limit(p, bank, L)) && w <= 8192.0 &&
               w < (double)n / 8) {
      shorts.push_back(j);
      short_w.push_back((int32_t)w);
    } else {
      keep.push_back(j);
    }
  }
  // Bands left for the two-pass kernels because the reference cuts their atoms off at |x| = n / 2 (a spectrum with
  // 1 / k side lobes): with the last `e` samples before the cut tapered away the spectrum is narrow enough for the
  // zoom engine; what the taper removed is a pair of e-tap filters at lags +-n / 2 (k_block_edge), added back by the
  // zoom kernel.
  std::vector<int32_t> split;
  if (bank == QI_BANK_STYX) {
    if (p->split_bank) (void)hipFree(p->split_bank);
    if (p->d_split_bands) (void)hipFree(p->d_split_bands);
    p->split_bank = nullptr;
    p->d_split_bands = nullptr;
    p->h_split_bands.clear();
    p->nsplit = 0;
  }
  const int64_t se = p->native_split_e;
  if (bank == QI_BANK_STYX && p->native_split && can_block && !picks.empty() &&  // (their edge items ride in the block launch)
      (se == 512 || se == 1024 || se == 2048) && n >= 8 * se) {
    for (int32_t j : keep) {
      const int64_t lo = (int64_t)sup[3 * j + 1], hi = (int64_t)sup[3 * j + 2];
      const int64_t len = hi >= lo ? hi - lo + 1 : 0;
      // (a band the one-pass loader of the two-pass kernels would take stays there only where those kernels exist)
      if (zoom_class(p, bank, L, len) >= 0 || (len > 0 && len <= p->native_kmax && native_len_ok(L))) continue;
      std::vector<double> part;
      QI_TRY(analyse_support(p, 0, L, B, j, 1, d_par, &part, st, (double)se));
      const int64_t tlo = (int64_t)part[1], thi = (int64_t)part[2];
      const int64_t tlen = thi >= tlo ? thi - tlo + 1 : 0;
      if (tune_env("QI_NATIVE_VERBOSE"))
        fprintf(stderr, "[qi plan] band %d: support %lld bins as the reference cuts it, %lld bins tapered over %lld samples\n",
                j, (long long)len, (long long)tlen, (long long)se);
      if (zoom_class(p, bank, L, tlen) < 0) continue;
      std::copy(part.begin(), part.end(), sup.begin() + 3 * j);
      split.push_back(j);
    }
  }
  std::vector<double> sup_keep;
  std::vector<int32_t> add_row;
  for (int32_t j : keep) {
    sup_keep.insert(sup_keep.end(), sup.begin() + 3 * j, sup.begin() + 3 * j + 3);
    const auto it = std::find(split.begin(), split.end(), j);
    add_row.push_back(it == split.end() ? 0 : (int32_t)(it - split.begin()) + 1);
  }
  QI_TRY(make_native_table<T>(p, bank, circular, L, B, keep, sup_keep, {}, d_par, st, add_row));
  if (!split.empty()) {
    // filter spectra of the edge pieces: taps in float64, transformed, scaled by 1 / 4096
    const size_t rows = split.size() * 2;
    if (p->ws_bytes < rows * native::kBlk * sizeof(double2) + 4096) {
      set_error("workspace too small for the taps of the split bands");
      return QI_ERR_NOMEM;
    }
    double2* taps = reinterpret_cast<double2*>(p->ws);
    QI_HIP(hipMalloc((void**)&p->d_split_bands, split.size() * sizeof(int32_t)));
    int32_t* d_ids = p->d_split_bands;
    int rc = hipMemcpy(d_ids, split.data(), split.size() * sizeof(int32_t), hipMemcpyHostToDevice) == hipSuccess
                 ? QI_OK : QI_ERR_HIP;
    if (rc == QI_OK)
      rc = native::launch_block_taps_edge(taps, (int)(se / 2), n, (double)se, d_par, B, d_ids, (int)split.size(), st);
    if (rc == QI_OK) rc = fft_c2c<double>(p->fft, taps, native::kBlk, (int64_t)rows, HIPFFT_FORWARD, st);
    if (rc == QI_OK && hipMalloc(&p->split_bank, rows * native::kBlk * sizeof(cplx<T>)) != hipSuccess) rc = QI_ERR_NOMEM;
    if (rc == QI_OK)
      rc = launch_bank_convert<T>(taps, static_cast<cplx<T>*>(p->split_bank), (int64_t)rows * native::kBlk, 0,
                                  1.0 / (double)native::kBlk, st);
    (void)hipStreamSynchronize(st);
    if (rc != QI_OK) {
      if (rc == QI_ERR_HIP) set_error("building the edge pieces of the split bands failed");
      return rc;
    }
    p->nsplit = (int32_t)split.size();
    p->h_split_bands = split;
  }
  p->nat[bank].nbands = B;  // the table's panel has all B rows even when some are produced by table 3 / the block engine
  if (bank == QI_BANK_STYX) {
    std::stable_sort(picks.begin(), picks.end(), [](const BlockPick& x, const BlockPick& y) { return x.wq < y.wq; });
    QI_TRY(build_block_gabor<T>(p, 0, B, picks, d_par, st));
  }
  if (bank == QI_BANK_STYX) {
    p->nat[3].release();
    if (p->d_edge) (void)hipFree(p->d_edge);
    p->d_edge = nullptr;
    p->nedge = 0;
    p->edge_wmax = 0;
    if (!shorts.empty()) {
      // spectra of the circular (length n) form of the short atoms
      std::vector<double> sup_s((size_t)shorts.size() * 3);
      size_t q = 0;
      while (q < shorts.size()) {
        size_t r = q + 1;
        while (r < shorts.size() && shorts[r] == shorts[r - 1] + 1) ++r;
        std::vector<double> part;
        QI_TRY(analyse_support(p, 1, n, B, shorts[q], (int32_t)(r - q), d_par, &part, st));
        std::copy(part.begin(), part.end(), sup_s.begin() + 3 * q);
        q = r;
      }
      QI_TRY(make_native_table<T>(p, 3, 1, n, B, shorts, sup_s, short_w, d_par, st));
      p->nat[3].nbands = B;
      std::vector<native::EdgeBand> eb(shorts.size());
      for (size_t i = 0; i < shorts.size(); ++i) {
        const int32_t j = shorts[i];
        eb[i].out_band = j;
        eb[i].w = short_w[i];
        eb[i].p_re = h_par[j];
        eb[i].p_im = h_par[B + j];
        eb[i].omega = h_par[2 * B + j];
        eb[i].amp = h_par[3 * B + j];
        if (short_w[i] > p->edge_wmax) p->edge_wmax = short_w[i];
      }
      QI_HIP(hipMalloc((void**)&p->d_edge, eb.size() * sizeof(native::EdgeBand)));
      QI_HIP(hipMemcpy(p->d_edge, eb.data(), eb.size() * sizeof(native::EdgeBand), hipMemcpyHostToDevice));
      p->nedge = (int32_t)eb.size();
    }
  }
  return QI_OK;
}

// One transform on the native engine: forward FFT of the records (hipFFT), then per table (the styx bank has two:
// the 2n-point linear part and the n-point circular part for short atoms) pass 1 for the wide bands and pass 2 with
// the fused epilogue for every band, the edge correction of the short-atom bands, and a fixed-order finalisation of
// the reductions.
// Work items of the joint block launch: the items of the styx table (0) and of the Stockwell table (2) on the same
// (reach group, block) are paired chunk by chunk -- one forward transform serves both; what has no partner stays single;
// the edge items of the styx table keep their place at the end.
int build_dual_items(qi_plan* p, int cut) {
  if (p->dual_valid[cut]) return QI_OK;
  if (p->d_dual[cut]) (void)hipFree(p->d_dual[cut]);
  p->d_dual[cut] = nullptr;
  p->n_dual[cut] = 0;
  std::map<std::pair<int32_t, int32_t>, std::pair<std::vector<native::BlockItem>, std::vector<native::BlockItem>>> at;
  std::vector<native::DualItem> dual, edge;
  for (const auto& it : p->blk[0].var[cut].h_items) {
    if (it.wq < 0) edge.push_back({it.wq, it.block, it.band_first, it.band_count, it.plane, it.stat_slot, 0, 0, 0, 0});
    else at[{it.wq, it.block}].first.push_back(it);
  }
  for (const auto& it : p->blk[2].var[cut].h_items) at[{it.wq, it.block}].second.push_back(it);
  for (const auto& kv : at) {
    const auto& a = kv.second.first;
    const auto& b = kv.second.second;
    for (size_t i = 0; i < std::max(a.size(), b.size()); ++i) {
      native::DualItem d{kv.first.first, kv.first.second, 0, 0, 0, 0, 0, 0, 0, 0};
      if (i < a.size()) {
        d.first0 = a[i].band_first;
        d.count0 = a[i].band_count;
        d.plane0 = a[i].plane;
        d.slot0 = a[i].stat_slot;
      }
      if (i < b.size()) {
        d.first2 = b[i].band_first;
        d.count2 = b[i].band_count;
        d.plane2 = b[i].plane;
        d.slot2 = b[i].stat_slot;
      }
      dual.push_back(d);
    }
  }
  std::stable_sort(dual.begin(), dual.end(), [](const native::DualItem& x, const native::DualItem& y) {
    const bool lx = x.wq == native::kBlkLongWq, ly = y.wq == native::kBlkLongWq;
    return lx != ly ? lx : x.count0 + x.count2 > y.count0 + y.count2;
  });
  p->n_dual_long[cut] = (int32_t)std::count_if(dual.begin(), dual.end(), [](const native::DualItem& x) { return x.wq == native::kBlkLongWq; });
  dual.insert(dual.end(), edge.begin(), edge.end());
  if (dual.empty()) return QI_OK;
  QI_HIP(hipMalloc((void**)&p->d_dual[cut], dual.size() * sizeof(native::DualItem)));
  QI_HIP(hipMemcpy(p->d_dual[cut], dual.data(), dual.size() * sizeof(native::DualItem), hipMemcpyHostToDevice));
  p->n_dual[cut] = (int32_t)dual.size();
  p->dual_valid[cut] = true;
  return QI_OK;
}

int launch_tail_call(const TailCall& t, hipStream_t st) {
  return native::launch_tail<float>(t.time_part, t.out_time, t.ct, t.n, t.chunk_total, nullptr, 0, t.part_band, t.part_stat,
                                    t.power_band, t.stats, t.B, t.nbk, t.stat_slots, t.band_slots, st);
}

int launch_zoom_all(qi_plan* p, const native::ZoomArgs<float>& z, int64_t ct, hipStream_t st) {
  p->prof.begin(st, QI_STAGE_ZOOM_COARSE);
  if (p->native_gather_fused > 0 && ct >= p->native_gather_fused) {
    QI_TRY(native::launch_zoom_coarse_gather<float>(z, ct, st));
  } else {
    QI_TRY(native::launch_zoom_gather<float>(z, 0, ct, st));
    QI_TRY(native::launch_zoom_coarse<float>(z, 0, ct, st));
  }
  p->prof.end(QI_STAGE_ZOOM_COARSE, st);
  p->prof.begin(st, QI_STAGE_ZOOM);
  QI_TRY(native::launch_zoom<float>(z, ct, st));
  p->prof.end(QI_STAGE_ZOOM, st);
  return QI_OK;
}

// the deferred launches of a CWT run, on their own
int flush_carry(qi_plan* p, FusedCarry* c, hipStream_t st) {
  if (!c || !c->active) return QI_OK;
  c->active = false;
  if (c->has_zoom) {
    c->has_zoom = false;
    QI_TRY(launch_zoom_all(p, c->zoom, c->ct, st));
  }
  p->prof.begin(st, QI_STAGE_BLOCK);
  QI_TRY(native::launch_block<float>(c->blk, c->demod, c->ct, st));
  p->prof.end(QI_STAGE_BLOCK, st);
  p->prof.begin(st, QI_STAGE_EPILOGUE);
  QI_TRY(launch_tail_call(c->tail, st));
  p->prof.end(QI_STAGE_EPILOGUE, st);
  return QI_OK;
}

template <typename T>
int run_native(qi_plan* p, int kind, const void* sig_v, int64_t C, const qi_tfr_out* out, hipStream_t st,
               bool may_share = false, FusedCarry* defer = nullptr, FusedCarry* finish = nullptr,
               size_t* probe = nullptr) {
  // probe: only report the scratch bytes one record needs when this run is the `defer` (CWT) or the `finish`
  // (Stockwell, spectra shared) half of a joint qi_cwt_stx tile of C records; nothing is launched
  static_assert(std::is_same<T, float>::value, "the native engine is float32");
  struct Sub {
    const qi_plan::NativeTable* t;
    int kernel_kind;
    int64_t N1, nblk;
    std::vector<int> nchunk;
  };
  std::vector<Sub> subs;
  subs.push_back({&p->nat[kind], kind, 0, 0, {}});
  const bool shorts = kind == 0 && p->nat[3].ready && p->nedge > 0;
  if (shorts) subs.push_back({&p->nat[3], 1, 0, 0, {}});
  const int64_t n = p->n, B = p->nat[kind].nbands, Lf0 = p->nat[kind].Lf;
  const T* sig = static_cast<const T*>(sig_v);
  const int G = p->native_rows;
  int64_t nblk_max = 0, imd_elems = 0;
  int chunk_total = 0;
  for (auto& sb : subs) {
    sb.N1 = sb.t->Lf / native::kN2;
    sb.nblk = sb.N1 / G;
    if (sb.nblk > nblk_max) nblk_max = sb.nblk;
    if ((int64_t)sb.t->imd_slots * sb.t->Lf > imd_elems) imd_elems = (int64_t)sb.t->imd_slots * sb.t->Lf;
    for (const auto& grp : sb.t->groups) {
      // chunks (workgroups along the band list): enough workgroups to fill the chip
      int nc = (int)ceil_div(p->native_wgs, sb.nblk * C);
      if (nc < 1) nc = 1;
      if (nc > grp.count) nc = grp.count;
      sb.nchunk.push_back(nc);
      chunk_total += nc;
    }
  }
  if (imd_elems < Lf0) imd_elems = Lf0;  // the forward transform of the records stages through one slot
  // block engine launches (one per reach group): their chunks come after the pass-2 chunks
  const auto& bt = p->blk[kind];
  const bool blocks = kind != 1 && bt.ready;
  const int cut = C >= batch_from(p) ? 1 : 0;  // (both halves of a joint tile see the same C and the same tables)
  const auto& il = bt.var[cut];
  const int chunk_p2 = chunk_total;
  int64_t blk_stats = 0, blk_slots = 0;
  if (blocks) {
    chunk_total += il.nplanes;
    blk_stats = il.nitems + il.nedge_items;
    blk_slots = bt.max_blocks;
  }
  // zoom engine launch (narrow bands of the main table): its chunks come last
  const auto& zt = p->nat[kind];
  const bool zoom = zt.nzoom > 0;
  constexpr int NL = native::kZoomClasses;
  // per-class band counts of this call: with few records the launch cannot afford rows for every class (its workgroup
  // budget is dealt over the rows), so the short-interpolator classes run as part of the 10-tap class of their grid
  // (their bands are oversampled enough for any of the three interpolators)
  int zcount[NL];
  for (int g = 0; g < NL; ++g) zcount[g] = zt.zoom_count[g];
  if (C < p->native_zoom_short_from) {
    zcount[0] += zcount[5] + zcount[6];
    zcount[5] = zcount[6] = 0;
  }
  int znchunk[NL] = {}, zplanes = 0;
  int64_t zstat_base[NL] = {}, zgroups[NL] = {}, zoom_stats = 0, zslots = 0;
  const int chunk_z0 = chunk_total;
  if (zoom) {
    // one launch for every level: each (level, chunk) pair is a row of the grid and owns a per-time plane
    // All workgroups of the launch should be resident at once (native_zoom_wgs of them) and finish together: every
    // level starts with one row, then the level whose rows carry the most work per workgroup gets the next one
    // (per band: a little more at the higher levels, half at level 3 and up where a workgroup covers half the samples).
    const double level_cost[NL] = {1.0, 1.08, 1.25, 0.75, 1.0, 0.85, 0.75};
    int64_t wgs = 0;
    for (int g = 0; g < NL; ++g) {
      if (zcount[g] <= 0) continue;
      zgroups[g] = native::zoom_groups(n, g);
      znchunk[g] = 1;
      wgs += zgroups[g] * C;
    }
    // (in the joint launch of qi_cwt_stx the rows of both tables queue behind each other: there the split by work wins,
    // measured 3 %; in a launch of one table the per-level rule does, 1.5 %)
    const int64_t zoom_wgs = p->native_zoom_wgs > 0 ? p->native_zoom_wgs
                             : ((defer || (finish && (finish->active || probe))) && p->native_fuse > 3 ? p->native_zoom_wgs_joint : 0);
    if (zoom_wgs > 0) {
      for (;;) {
        int best = -1;
        double best_load = 0.0;
        for (int g = 0; g < NL; ++g) {
          if (zcount[g] <= 0 || znchunk[g] >= zcount[g]) continue;
          const double load = level_cost[g] * (double)ceil_div(zcount[g], znchunk[g]);
          if (load > best_load) {
            best_load = load;
            best = g;
          }
        }
        if (best < 0 || wgs + zgroups[best] * C > zoom_wgs) break;
        // (a level that cannot grow any more but carries the largest load ends the search: more rows elsewhere would
        // not shorten the launch)
        bool is_max = true;
        for (int g = 0; g < NL; ++g)
          if (zcount[g] > 0 && level_cost[g] * (double)ceil_div(zcount[g], znchunk[g]) > best_load) is_max = false;
        if (!is_max) break;
        znchunk[best] += 1;
        wgs += zgroups[best] * C;
      }
    } else {
      for (int g = 0; g < NL; ++g) {
        if (zcount[g] <= 0) continue;
        int nc = (int)ceil_div(p->native_zoom_waves, 4 * zgroups[g] * C);
        if (nc < 1) nc = 1;
        if (nc > zcount[g]) nc = zcount[g];
        znchunk[g] = nc;
      }
    }
    for (int g = 0; g < NL; ++g) {
      if (zcount[g] <= 0) continue;
      zplanes += znchunk[g];
      zstat_base[g] = zoom_stats;
      zoom_stats += (int64_t)znchunk[g] * zgroups[g];
      if (zgroups[g] > zslots) zslots = zgroups[g];
    }
    if (tune_env("QI_NATIVE_VERBOSE"))
      fprintf(stderr, "[qi run] zoom launch of table %d: bands per class %d %d %d %d %d | 6-tap %d 4-tap %d in rows %d %d %d %d %d | %d %d\n", kind,
              zcount[0], zcount[1], zcount[2], zcount[3], zcount[4], zcount[5],
              zcount[6], znchunk[0], znchunk[1], znchunk[2], znchunk[3], znchunk[4], znchunk[5], znchunk[6]);
    chunk_total += zplanes;
  }
  int64_t nbk = nblk_max + (shorts ? 1 : 0);          // partial slots per band (last one: edge samples)
  if (blk_slots > nbk) nbk = blk_slots;
  if (zslots > nbk) nbk = zslots;
  const int64_t p2_stats = (int64_t)chunk_p2 * nblk_max;
  const int64_t stat_slots = p2_stats + blk_stats + zoom_stats + (shorts ? p->nedge : 0);
  const bool want_band = out->power_band != nullptr, want_stat = out->stats != nullptr;
  const bool want_time = out->power_time != nullptr;
  const bool time_via_part = want_time && (chunk_total > 1 || shorts);
  // Every engine writes a dense prefix of its bands' partial slots and all of its stat slots, so nothing has to be
  // cleared when the finalisation knows each band's slot count; only the short-atom table (a second pass-2 geometry
  // plus the edge slot at the end of the row) keeps the cleared layout.
  const bool clear_parts = shorts;
  if (!shorts && !p->d_band_slots[kind][cut]) {
    std::vector<int32_t> slots((size_t)B, 0);
    for (int32_t r : p->nat[kind].h_rows) slots[r] = (int32_t)nblk_max;
    for (const auto& z : p->nat[kind].h_zoom) slots[z.first] = (int32_t)native::zoom_groups(n, z.second);
    if (blocks)
      for (const auto& b : bt.var[cut].h_bands) slots[b.first] = b.second;
    QI_HIP(hipMalloc((void**)&p->d_band_slots[kind][cut], slots.size() * sizeof(int32_t)));
    QI_HIP(hipMemcpy(p->d_band_slots[kind][cut], slots.data(), slots.size() * sizeof(int32_t), hipMemcpyHostToDevice));
  }
  // scratch regions, each [Ct][...] without per-channel padding
  // qi_cwt_stx: the Stockwell call can take its spectra from the even bins of the zero-padded spectra the CWT call
  // left at the start of the scratch -- when nothing of this table needs the n-point spectrum as an array (every band
  // on the zoom / block engines) and both calls hold all records in one tile
  bool share = may_share && kind == 2 && (probe || (p->shared_valid && p->shared_sig == sig_v && p->shared_C == C)) &&
               p->nat[kind].h_rows.empty() && !shorts;
  const size_t e_x = (size_t)(share ? 2 * Lf0 : Lf0) * sizeof(cplx<T>);
  const size_t e_xn = shorts ? (size_t)n * sizeof(cplx<T>) : 0;
  const size_t e_imd = (size_t)imd_elems * sizeof(cplx<T>);
  const size_t e_pb = (size_t)B * nbk * 8;
  const size_t e_ps = (size_t)stat_slots * 24;
  const size_t e_tp = time_via_part ? (size_t)chunk_total * n * sizeof(T) : 0;
  const size_t e_ep = shorts ? (size_t)p->nedge * 2 * p->edge_wmax * sizeof(T) : 0;
  const size_t e_et = shorts ? (size_t)2 * p->edge_wmax * sizeof(T) : 0;
  const size_t e_ez = shorts && !out->coef ? (size_t)p->nedge * 2 * p->edge_wmax * sizeof(cplx<T>) : 0;
  const size_t e_zc = zoom ? (size_t)zt.zoom_planes * native::kBlk * sizeof(cplx<T>) : 0;
  const int32_t nsplit = kind == 0 ? p->nsplit : 0;  // split bands: the zoom launch hands its part to the block launch
  const size_t e_add = (size_t)nsplit * n * sizeof(cplx<T>);
  const size_t per_chan = e_x + e_xn + e_imd + e_pb + e_ps + e_tp + e_ep + e_et + e_ez + e_zc + e_add;
  if (probe) {
    *probe = per_chan;
    return QI_OK;
  }
  if (p->ws_bytes < per_chan + 4096) {
    set_error("workspace of %zu bytes cannot hold one record's native scratch of %zu bytes", p->ws_bytes,
              per_chan + 4096);
    return QI_ERR_NOMEM;
  }
  int64_t Ct = (int64_t)((p->ws_bytes - 4096) / per_chan);
  if (Ct > C) Ct = C;
  if (share && Ct != C) {
    set_error("internal: shared spectra need all records in one tile");  // cannot happen: the CWT scratch is larger
    return QI_ERR_STATE;
  }
  const bool tail_one = time_via_part && (want_band || want_stat) && p->native_tail;
  const bool overlap = blocks && p->native_overlap && nsplit == 0;  // (edge items need the zoom launch's output)
  // qi_cwt_stx: a CWT run whose records fit one tile leaves its block launch and tail to the Stockwell run ...
  const bool deferring = defer && kind == 0 && Ct == C && blocks && !overlap && !shorts && tail_one;
  // ... which keeps the CWT run's scratch intact (its own follows it; only the spectra are shared) and finishes both
  bool finishing = finish && finish->active && kind == 2 && share && blocks && !overlap && tail_one;
  if (finishing) {
    const size_t need = align_up(e_x * (size_t)C) + align_up(finish->ws_used) + (per_chan - e_x) * (size_t)C + 64 * 256;
    if (need > p->ws_bytes) finishing = false;
  }
  if (finish && finish->active && !finishing) QI_TRY(flush_carry(p, finish, st));  // before this run reuses the scratch
  if (kind == 0) {  // what this call will leave behind for a following qi_cwt_stx Stockwell call
    p->shared_valid = Ct == C;
    p->shared_sig = sig_v;
    p->shared_C = C;
  } else if (!share) {
    p->shared_valid = false;  // the scratch is about to be overwritten
  }
  char* w = p->ws;
  auto carve = [&](size_t bytes) {
    char* r = w;
    w += align_up(bytes * Ct);
    return r;
  };
  cplx<T>* X = reinterpret_cast<cplx<T>*>(carve(e_x));
  if (finishing) w = p->ws + align_up(finish->ws_used);
  cplx<T>* Xn = reinterpret_cast<cplx<T>*>(carve(e_xn));
  cplx<T>* imd = reinterpret_cast<cplx<T>*>(carve(e_imd));
  char* parts0 = w;
  double* part_band = reinterpret_cast<double*>(carve(e_pb));
  double* part_stat = reinterpret_cast<double*>(carve(e_ps));
  const size_t parts_bytes = (size_t)(w - parts0);
  T* time_part = reinterpret_cast<T*>(carve(e_tp));
  T* edge_p = reinterpret_cast<T*>(carve(e_ep));
  T* edge_time = reinterpret_cast<T*>(carve(e_et));
  cplx<T>* edge_z = e_ez ? reinterpret_cast<cplx<T>*>(carve(e_ez)) : nullptr;
  cplx<T>* zcoarse = e_zc ? reinterpret_cast<cplx<T>*>(carve(e_zc)) : nullptr;
  cplx<T>* zadd = e_add ? reinterpret_cast<cplx<T>*>(carve(e_add)) : nullptr;

  if (overlap && !p->side) {
    QI_HIP(hipStreamCreateWithFlags(&p->side, hipStreamNonBlocking));
    QI_HIP(hipEventCreateWithFlags(&p->ev_fork, hipEventDisableTiming));
    QI_HIP(hipEventCreateWithFlags(&p->ev_join, hipEventDisableTiming));
  }
  for (int64_t c0 = 0; c0 < C; c0 += Ct) {
    const int64_t ct = (C - c0 < Ct) ? C - c0 : Ct;
    // qi_cwt_stx, joint block launch: its band items need nothing but the records, so they run on a side stream BESIDE
    // the zoom engine's launches (neither kernel fills the vector pipes by itself: ~47 % issue each); the edge items of the
    // split bands follow the interpolation launch, whose output they add to
    const bool joint_blk = finishing && blocks && p->native_fuse > 1 && finish->ct == ct && !finish->demod && bt.demod &&
                           (finish->blk.coef != nullptr) == (out->coef != nullptr) &&
                           (finish->blk.bits != nullptr) == (out->bits != nullptr);
    const bool pair = joint_blk && p->native_pair && !overlap;
    if (pair && !p->side) {
      QI_HIP(hipStreamCreateWithFlags(&p->side, hipStreamNonBlocking));
      QI_HIP(hipEventCreateWithFlags(&p->ev_fork, hipEventDisableTiming));
      QI_HIP(hipEventCreateWithFlags(&p->ev_join, hipEventDisableTiming));
    }
    auto launch_blocks = [&](hipStream_t bs, int phase = 0) -> int {  // phase 1: band items only, 2: edge items only (joint launch)
      native::BlockArgs<T> b{};
      b.n = n;
      b.nitems = il.nitems;
      b.nlong = il.nlong;
      b.nedge_items = il.nedge_items;
      b.nsplit = nsplit;
      b.edge_band = p->d_split_bands;
      b.edge_bank = static_cast<const cplx<T>*>(p->split_bank);
      b.edge_part = zadd;
      b.panel_bands = (int32_t)B;
      b.items = il.d_items;
      b.bands = static_cast<const native::BlockBandT<T>*>(il.d_bands);
      b.bank = static_cast<const cplx<T>*>(bt.bank);
      b.lz_w = p->d_lz_w;
      b.sig = sig + c0 * n;
      b.coef = out->coef ? static_cast<cplx<T>*>(out->coef) + c0 * B * n : nullptr;
      b.bits = out->bits ? static_cast<T*>(out->bits) + c0 * B * n : nullptr;
      b.time_part = !want_time ? nullptr : (time_via_part ? time_part : static_cast<T*>(out->power_time) + c0 * n);
      b.part_band = want_band ? part_band : nullptr;
      b.part_stat = want_stat ? part_stat : nullptr;
      b.nblk = nbk;
      b.stat_stride = stat_slots;
      b.stat_base = p2_stats;
      b.chunk_base = chunk_p2;
      b.chunk_total = chunk_total;
      b.power_scale = (T)(out->power_scale == 0.0 ? 1.0 : out->power_scale);
      b.eps = (T)(out->eps == 0.0 ? 2.220446049250313e-16 : out->eps);
      b.two_over_n = (float)(2.0 / (double)n);
      b.debug = p->native_debug;
      b.stamps = p->blk_stamps;
      if (deferring) {  // the Stockwell run of qi_cwt_stx launches it
        defer->blk = b;
        defer->demod = bt.demod;
        defer->ct = ct;
        return QI_OK;
      }
      p->prof.unchain_span();
      p->prof.begin(bs, QI_STAGE_BLOCK);
      if (joint_blk) {
        QI_TRY(build_dual_items(p, cut));
        const int32_t n_edge = p->blk[0].var[cut].nedge_items;
        const native::DualItem* items = p->d_dual[cut];
        int32_t count = p->n_dual[cut], nlong = p->n_dual_long[cut];
        if (phase == 1) count -= n_edge;
        if (phase == 2) {
          items += count - n_edge;
          count = n_edge;
          nlong = 0;
        }
        QI_TRY(native::launch_block_dual<T>(finish->blk, b, items, count, nlong, ct, bs));
      } else {
        if (finishing) QI_TRY(native::launch_block<T>(finish->blk, finish->demod, finish->ct, bs));
        QI_TRY(native::launch_block<T>(b, bt.demod, ct, bs));
      }
      p->prof.end(QI_STAGE_BLOCK, bs);
      p->prof.unchain_span();
      return QI_OK;
    };
    if (clear_parts) QI_HIP(hipMemsetAsync(parts0, 0, parts_bytes, st));
    if (pair) {
      QI_HIP(hipEventRecord(p->ev_fork, st));
      QI_HIP(hipStreamWaitEvent(p->side, p->ev_fork, 0));
      QI_TRY(launch_blocks(p->side, 1));
      QI_HIP(hipEventRecord(p->ev_join, p->side));
    }
    if (overlap) {  // fork: the block launch follows the clearing of the partials and nothing else
      QI_HIP(hipEventRecord(p->ev_fork, st));
      QI_HIP(hipStreamWaitEvent(p->side, p->ev_fork, 0));
      QI_TRY(launch_blocks(p->side));
      QI_HIP(hipEventRecord(p->ev_join, p->side));
    }
    p->prof.begin(st, QI_STAGE_FORWARD);
    if (share) {
      // X already holds the zero-padded spectra of these records
    } else if (p->native_fwd && native_len_ok(Lf0)) {
      native::RowArgs<T> f{};
      f.Lf = Lf0;
      f.n = n;
      f.N1 = Lf0 / native::kN2;
      f.N2 = native::kN2;
      f.imd_slots = 1;
      f.imd = imd;
      f.sig = sig + c0 * n;
      f.two_over_len = (float)(2.0 / (double)Lf0);
      f.debug = 0;
      QI_TRY(native::launch_forward<T>(f, X, ct, st));
    } else {
      QI_TRY(launch_pack_pad<T>(sig + c0 * n, X, ct, n, Lf0, st));
      QI_TRY(fft_c2c<T>(p->fft, X, Lf0, ct, HIPFFT_FORWARD, st));
    }
    if (shorts) QI_TRY(native::launch_even_bins<T>(X, Xn, ct, n, st));
    p->prof.end(QI_STAGE_FORWARD, st);
    int chunk_base = 0;
    for (size_t si = 0; si < subs.size(); ++si) {
      const Sub& sb = subs[si];
      const auto& t = *sb.t;
      native::RowArgs<T> a{};
      a.Lf = t.Lf;
      a.n = n;
      a.N1 = sb.N1;
      a.N2 = native::kN2;
      a.panel_bands = (int32_t)B;
      a.imd_slots = t.imd_slots;
      a.chunk_total = chunk_total;
      a.X = si == 0 ? X : Xn;
      a.Hc = static_cast<const cplx<T>*>(t.Hc);
      a.Hfull = static_cast<const cplx<T>*>(t.Hfull);
      a.imd = imd;
      a.inv_len = (T)(1.0 / (double)t.Lf);
      a.two_over_len = (float)(2.0 / (double)t.Lf);
      a.debug = p->native_debug;
      a.stamps = p->stamps;
      a.neg_last_row = sb.kernel_kind == 0 ? 1 : 0;
      a.coef = out->coef ? static_cast<cplx<T>*>(out->coef) + c0 * B * n : nullptr;
      a.bits = out->bits ? static_cast<T*>(out->bits) + c0 * B * n : nullptr;
      a.edge_z = edge_z;
      a.edge_wmax = p->edge_wmax;
      a.nedge = p->nedge;
      a.time_part = !want_time ? nullptr : (time_via_part ? time_part : static_cast<T*>(out->power_time) + c0 * n);
      a.part_band = want_band ? part_band : nullptr;
      a.part_stat = want_stat ? part_stat : nullptr;
      a.nblk = nbk;
      a.stat_nblk = nblk_max;
      a.stat_stride = stat_slots;
      a.power_scale = (T)(out->power_scale == 0.0 ? 1.0 : out->power_scale);
      a.eps = (T)(out->eps == 0.0 ? 2.220446049250313e-16 : out->eps);
      for (size_t g = 0; g < t.groups.size(); ++g) {
        const auto& grp = t.groups[g];
        a.bands = t.d_bands + grp.first;
        a.nbands = grp.count;
        a.gen_list = t.d_gen_list ? t.d_gen_list + grp.gen_first : nullptr;
        a.ngen_launch = grp.ngen;
        a.chunk_base = chunk_base;
        if (grp.ngen > 0) {
          p->prof.begin(st, QI_STAGE_PASS1);
          QI_TRY(native::launch_pass1<T>(a, sb.kernel_kind, ct, st));
          p->prof.end(QI_STAGE_PASS1, st);
        }
        p->prof.begin(st, QI_STAGE_PASS2);
        QI_TRY(native::launch_pass2<T>(a, sb.kernel_kind, G, sb.nchunk[g], ct, st));
        p->prof.end(QI_STAGE_PASS2, st);
        chunk_base += sb.nchunk[g];
      }
    }
    if (zoom) {
      native::ZoomArgs<T> z{};
      z.n = n;
      z.Lf = zt.Lf;
      z.planes = zt.zoom_planes;
      z.nbands = zt.nzoom;
      z.panel_bands = (int32_t)B;
      z.bands = zt.d_zoom;
      z.plane_band = zt.d_zoom_plane_band;
      z.X = X;
      z.x_shift = share ? 1 : 0;
      z.Hc = static_cast<const cplx<T>*>(zt.Hc);
      z.coarse = zcoarse;
      z.stx = kind == 2 ? 1 : 0;
      // panel sample t is full-length sample t + off: linear correlation off = n/2 - 1, rolled circular n/2, Stockwell 0
      z.lane_off = kind == 0 ? 1 : 0;
      z.tau_off = kind == 2 ? 0 : n / 2 / native::kZoomD;
      z.inv_len = (T)(1.0 / (double)zt.Lf);
      z.two_over_len = (float)(2.0 / (double)zt.Lf);
      z.coef = out->coef ? static_cast<cplx<T>*>(out->coef) + c0 * B * n : nullptr;
      z.bits = out->bits ? static_cast<T*>(out->bits) + c0 * B * n : nullptr;
      z.time_part = !want_time ? nullptr : (time_via_part ? time_part : static_cast<T*>(out->power_time) + c0 * n);
      z.part_band = want_band ? part_band : nullptr;
      z.part_stat = want_stat ? part_stat : nullptr;
      z.nblk = nbk;
      z.stat_stride = stat_slots;
      z.chunk_base = chunk_z0;
      z.chunk_total = chunk_total;
      z.power_scale = (T)(out->power_scale == 0.0 ? 1.0 : out->power_scale);
      z.eps = (T)(out->eps == 0.0 ? 2.220446049250313e-16 : out->eps);
      z.split_part = zadd;
      z.split_rows = nsplit;
      int chunk0 = 0;
      for (int g = 0; g < NL; ++g) {
        z.lvl_count[g] = zcount[g];
        z.lvl_chunk0[g] = chunk0;
        z.lvl_nchunk[g] = znchunk[g];
        z.lvl_stat_base[g] = p2_stats + blk_stats + zstat_base[g];
        z.lvl_weights[g] = p->d_zoom_w[g][0];  // (a lane's position in its window does not depend on the kind)
        chunk0 += znchunk[g];
      }
      int first = 0;
      for (int gi = 0; gi < NL; ++gi) {  // positions in the band list (kZoomListOrder): a merged class 0 starts where class 6 does
        const int g = kZoomListOrder[gi];
        z.lvl_first[g] = first;
        first += zcount[g];
        if (g == 0 && zcount[0] != zt.zoom_count[0]) z.lvl_first[0] = 0;
      }
      if (deferring && p->native_fuse > 2) {  // the Stockwell run of qi_cwt_stx launches them with its own
        defer->zoom = z;
        defer->has_zoom = true;
        defer->ct = ct;
      } else {
        const bool joint = finishing && finish->has_zoom && finish->ct == ct;
        p->prof.begin(st, QI_STAGE_ZOOM_COARSE);
        const bool gfused = p->native_gather_fused > 0 && ct >= p->native_gather_fused;
        if (joint && gfused) {
          QI_TRY(native::launch_zoom_coarse_gather2<T>(finish->zoom, z, ct, st));
        } else if (joint) {
          QI_TRY(native::launch_zoom_gather2<T>(finish->zoom, z, ct, st));
          QI_TRY(native::launch_zoom_coarse2<T>(finish->zoom, z, ct, st));
        } else if (gfused) {
          QI_TRY(native::launch_zoom_coarse_gather<T>(z, ct, st));
        } else {
          QI_TRY(native::launch_zoom_gather<T>(z, zt.zoom_max_level, ct, st));
          QI_TRY(native::launch_zoom_coarse<T>(z, zt.zoom_max_level, ct, st));
        }
        p->prof.end(QI_STAGE_ZOOM_COARSE, st);
        if (pair && p->native_pair == 2) {  // the block launch has covered the coarse stage; the interpolation launch runs alone
          QI_HIP(hipStreamWaitEvent(st, p->ev_join, 0));
        }
        p->prof.begin(st, QI_STAGE_ZOOM);
        const bool joint_fine = joint && p->native_fuse > 3 && (finish->zoom.coef != nullptr) == (z.coef != nullptr) &&
                                (finish->zoom.bits != nullptr) == (z.bits != nullptr);
        if (joint_fine) {
          QI_TRY(native::launch_zoom2<T>(finish->zoom, z, ct, st));
        } else {
          if (joint) QI_TRY(native::launch_zoom<T>(finish->zoom, ct, st));
          QI_TRY(native::launch_zoom<T>(z, ct, st));
        }
        if (joint) finish->has_zoom = false;
        p->prof.end(QI_STAGE_ZOOM, st);
      }
    }
    if (finishing && finish->has_zoom) {  // (this table has no zoom band, or another tiling: the deferred launches alone)
      QI_TRY(launch_zoom_all(p, finish->zoom, finish->ct, st));
      finish->has_zoom = false;
    }
    // (the edge items of the block launch finish the split bands the zoom launch began: it comes after it)
    if (pair) {
      QI_HIP(hipStreamWaitEvent(st, p->ev_join, 0));
      if (p->blk[0].var[cut].nedge_items > 0) QI_TRY(launch_blocks(st, 2));
    } else if (blocks && !overlap) {
      QI_TRY(launch_blocks(st));
    }
    if (overlap) QI_HIP(hipStreamWaitEvent(st, p->ev_join, 0));  // join before the reductions are finalised
    p->prof.begin(st, QI_STAGE_EPILOGUE);
    if (shorts) {
      native::EdgeArgs<T> e{};
      e.bands = p->d_edge;
      e.nedge = p->nedge;
      e.panel_bands = (int32_t)B;
      e.n = n;
      e.wmax = p->edge_wmax;
      e.stat_slots = stat_slots;
      e.sig = sig + c0 * n;
      e.coef = out->coef ? static_cast<cplx<T>*>(out->coef) + c0 * B * n : nullptr;
      e.edge_z = edge_z;
      e.bits = out->bits ? static_cast<T*>(out->bits) + c0 * B * n : nullptr;
      e.edge_p = edge_p;
      e.power_scale = (T)(out->power_scale == 0.0 ? 1.0 : out->power_scale);
      e.eps = (T)(out->eps == 0.0 ? 2.220446049250313e-16 : out->eps);
      QI_TRY(native::launch_edge<T>(e, ct, want_time ? edge_time : nullptr, want_band ? part_band : nullptr, nbk,
                                    nbk - 1, want_stat ? part_stat : nullptr, stat_slots - p->nedge, st));
    }
    if (deferring || finishing) {
      TailCall tc;
      tc.time_part = time_part;
      tc.out_time = static_cast<T*>(out->power_time) + c0 * n;
      tc.ct = ct;
      tc.n = n;
      tc.chunk_total = chunk_total;
      tc.part_band = want_band ? part_band : nullptr;
      tc.part_stat = want_stat ? part_stat : nullptr;
      tc.power_band = want_band ? static_cast<double*>(out->power_band) + c0 * B : nullptr;
      tc.stats = want_stat ? static_cast<double*>(out->stats) + c0 * 4 : nullptr;
      tc.B = B;
      tc.nbk = nbk;
      tc.stat_slots = stat_slots;
      tc.band_slots = p->d_band_slots[kind][cut];
      if (deferring) {
        defer->tail = tc;
        defer->ws_used = (size_t)(w - p->ws);
        defer->active = true;
      } else {
        finish->active = false;
        const TailCall& t0 = finish->tail;
        if (t0.ct == tc.ct && t0.n == tc.n) {
          QI_TRY(native::launch_tail2<float>(t0.time_part, t0.out_time, t0.chunk_total, t0.part_band, t0.part_stat,
                                             t0.power_band, t0.stats, t0.B, t0.nbk, t0.stat_slots, t0.band_slots,
                                             tc.time_part, tc.out_time, tc.chunk_total, tc.part_band, tc.part_stat,
                                             tc.power_band, tc.stats, tc.B, tc.nbk, tc.stat_slots, tc.band_slots, tc.ct,
                                             tc.n, st));
        } else {
          QI_TRY(launch_tail_call(t0, st));
          QI_TRY(launch_tail_call(tc, st));
        }
      }
    } else if (tail_one)
      QI_TRY(native::launch_tail<T>(time_part, static_cast<T*>(out->power_time) + c0 * n, ct, n, chunk_total,
                                    shorts ? edge_time : nullptr, p->edge_wmax, want_band ? part_band : nullptr,
                                    want_stat ? part_stat : nullptr,
                                    want_band ? static_cast<double*>(out->power_band) + c0 * B : nullptr,
                                    want_stat ? static_cast<double*>(out->stats) + c0 * 4 : nullptr, B, nbk, stat_slots,
                                    shorts ? nullptr : p->d_band_slots[kind][cut], st));
    else if (time_via_part)
      QI_TRY(native::launch_time_reduce<T>(time_part, static_cast<T*>(out->power_time) + c0 * n, ct, n, chunk_total,
                                           shorts ? edge_time : nullptr, p->edge_wmax, st));
    if ((want_band || want_stat) && !tail_one)
      QI_TRY(launch_finalize(want_band ? part_band : nullptr, want_stat ? part_stat : nullptr,
                             want_band ? static_cast<double*>(out->power_band) + c0 * B : nullptr,
                             want_stat ? static_cast<double*>(out->stats) + c0 * 4 : nullptr, ct, B, nbk, stat_slots,
                             st, shorts ? nullptr : p->d_band_slots[kind][cut]));
    p->prof.end(QI_STAGE_EPILOGUE, st);
  }
  return QI_OK;
}

// float64 records on the native two-pass kernels (exact algorithm: no truncated atoms, no interpolation): forward
// transform of the records by hipFFT, then per launch group pass 1 for the wide bands and pass 2 with the pruned loader
// and the fused epilogue for every band, one tail launch.  8-row workgroups (Cfg<double, 8>).
int run_native64(qi_plan* p, int kind, const void* sig_v, int64_t C, const qi_tfr_out* out, hipStream_t st) {
  using T = double;
  const auto& t = p->nat[kind];
  const int64_t n = p->n, B = kind == 2 ? p->nb_stx : p->nb[kind], Lf = t.Lf;
  constexpr int G = 8;
  const T* sig = static_cast<const T*>(sig_v);
  // two-pass sub-tables: the table itself and, for the styx bank, its wide-spectrum short-atom bands evaluated as
  // circular correlations of length n (table 3: half the bank row and intermediate; k_edge_fix restores the zero-padded
  // result on their first / last samples)
  struct Sub {
    const qi_plan::NativeTable* t;
    int kernel_kind;
    int64_t N1, nblk;
    std::vector<int> nchunk;
  };
  std::vector<Sub> subs;
  subs.push_back({&t, kind, 0, 0, {}});
  const bool shorts = kind == 0 && p->nat[3].ready && p->nedge > 0;
  if (shorts) subs.push_back({&p->nat[3], 1, 0, 0, {}});
  int chunk_total = 0;
  int64_t imd_elems = 0;
  for (auto& sb : subs) {
    sb.N1 = sb.t->Lf / native::kN2;
    sb.nblk = sb.N1 / G;
    if ((int64_t)sb.t->imd_slots * sb.t->Lf > imd_elems) imd_elems = (int64_t)sb.t->imd_slots * sb.t->Lf;
    for (const auto& grp : sb.t->groups) {
      int nc = (int)ceil_div(p->native_wgs, sb.nblk * C);
      nc = nc < 1 ? 1 : (nc > grp.count ? grp.count : nc);
      sb.nchunk.push_back(nc);
      chunk_total += nc;
    }
  }
  // partial slots per band: the row groups of the two-pass kernels (the circular sub-table has half as many) or the tiles
  // of the float64 zoom, whichever is more
  const int64_t nblk_z = n / native::kZ64Tile;
  int64_t nblk = subs[0].nblk > nblk_z ? subs[0].nblk : nblk_z;
  if (p->blk[kind].ready && kind != 1 && p->blk[kind].max_blocks > nblk) nblk = p->blk[kind].max_blocks;
  // (some bands leave slots unwritten: the block bands fill one slot per block of their reach group)
  const bool clear_parts = shorts || subs[0].nblk != nblk || (p->blk[kind].ready && kind != 1);
  // float64 zoom bands: one launch per coarse-grid level, its bands dealt to `zchunk` workgroups per tile
  int zchunk[native::kZ64Levels] = {};
  size_t e_z = 0;  // coarse storage of the largest level (the levels run one after the other)
  for (int g = 0; g < native::kZ64Levels; ++g) {
    if (t.z64_count[g] == 0) continue;
    int nc = (int)ceil_div(p->native_wgs, nblk_z * C);
    zchunk[g] = nc < 1 ? 1 : (nc > t.z64_count[g] ? t.z64_count[g] : nc);
    chunk_total += zchunk[g];
    const size_t bytes = (size_t)t.z64_count[g] * (size_t)((Lf / 64) << g) * sizeof(cplx<T>);
    if (bytes > e_z) e_z = bytes;
  }
  // block engine (short-atom bands with wide spectra, double arithmetic): its planes and stat slots come last
  const auto& bt = p->blk[kind];
  const bool blocks = kind != 1 && bt.ready;
  const auto& il = bt.var[C >= 4 ? 1 : 0];
  const int chunk_blk = chunk_total;
  int64_t blk_stats = 0;
  if (blocks) {
    chunk_total += il.nplanes;
    blk_stats = il.nitems;
  }
  // stat slots: [chunks of the two-pass and zoom launches][nblk], then one per block item, then the edge bands
  const int64_t blk_stat_base = (int64_t)chunk_blk * nblk;
  const int64_t stat_slots = blk_stat_base + blk_stats + (shorts ? p->nedge : 0);
  if (chunk_total == 0) {
    set_error("float64 native table has no band");
    return QI_ERR_STATE;
  }
  const bool want_band = out->power_band != nullptr, want_stat = out->stats != nullptr, want_time = out->power_time != nullptr;
  const bool time_via_part = want_time && (chunk_total > 1 || shorts);
  const int64_t nbk = nblk + (shorts ? 1 : 0);  // partial slots per band (last one: the corrected edge samples)
  const size_t e_x = (size_t)Lf * sizeof(cplx<T>);
  const size_t e_xn = shorts ? (size_t)n * sizeof(cplx<T>) : 0;
  const size_t e_imd = (size_t)imd_elems * sizeof(cplx<T>);
  const size_t e_pb = (size_t)B * nbk * 8, e_ps = (size_t)stat_slots * 24;
  const size_t e_tp = time_via_part ? (size_t)chunk_total * n * sizeof(T) : 0;
  const size_t e_ep = shorts ? (size_t)p->nedge * 2 * p->edge_wmax * sizeof(T) : 0;
  const size_t e_et = shorts ? (size_t)2 * p->edge_wmax * sizeof(T) : 0;
  const size_t e_ez = shorts && !out->coef ? (size_t)p->nedge * 2 * p->edge_wmax * sizeof(cplx<T>) : 0;
  const size_t per_chan = e_x + e_xn + e_imd + e_z + e_pb + e_ps + e_tp + e_ep + e_et + e_ez;
  if (p->ws_bytes < per_chan + 16384) {
    set_error("workspace of %zu bytes cannot hold one record's float64 scratch of %zu bytes", p->ws_bytes, per_chan + 16384);
    return QI_ERR_NOMEM;
  }
  int64_t Ct = (int64_t)((p->ws_bytes - 16384) / per_chan);
  if (Ct > C) Ct = C;
  p->shared_valid = false;
  char* w = p->ws;
  auto carve = [&](size_t bytes) {
    char* r = w;
    w += align_up(bytes * Ct);
    return r;
  };
  cplx<T>* X = reinterpret_cast<cplx<T>*>(carve(e_x));
  cplx<T>* Xn = reinterpret_cast<cplx<T>*>(carve(e_xn));
  cplx<T>* imd = reinterpret_cast<cplx<T>*>(carve(e_imd));
  cplx<T>* Z = reinterpret_cast<cplx<T>*>(carve(e_z));
  char* parts0 = w;  // the partial sums: cleared per tile when the sub-tables fill different numbers of slots
  double* part_band = reinterpret_cast<double*>(carve(e_pb));
  double* part_stat = reinterpret_cast<double*>(carve(e_ps));
  const size_t parts_bytes = (size_t)(w - parts0);
  T* time_part = reinterpret_cast<T*>(carve(e_tp));
  T* edge_p = reinterpret_cast<T*>(carve(e_ep));
  T* edge_time = reinterpret_cast<T*>(carve(e_et));
  cplx<T>* edge_z = e_ez ? reinterpret_cast<cplx<T>*>(carve(e_ez)) : nullptr;
  for (int64_t c0 = 0; c0 < C; c0 += Ct) {
    const int64_t ct = (C - c0 < Ct) ? C - c0 : Ct;
    if (clear_parts) QI_HIP(hipMemsetAsync(parts0, 0, parts_bytes, st));
    p->prof.begin(st, QI_STAGE_FORWARD);
    QI_TRY(launch_pack_pad<T>(sig + c0 * n, X, ct, n, Lf, st));
    QI_TRY(fft_c2c<T>(p->fft, X, Lf, ct, HIPFFT_FORWARD, st));
    if (shorts) QI_TRY(native::launch_even_bins<T>(X, Xn, ct, n, st));
    p->prof.end(QI_STAGE_FORWARD, st);
    cplx<T>* coef = out->coef ? static_cast<cplx<T>*>(out->coef) + c0 * B * n : nullptr;
    T* bits = out->bits ? static_cast<T*>(out->bits) + c0 * B * n : nullptr;
    T* tpart = !want_time ? nullptr : (time_via_part ? time_part : static_cast<T*>(out->power_time) + c0 * n);
    const double power_scale = out->power_scale == 0.0 ? 1.0 : out->power_scale;
    const double eps = out->eps == 0.0 ? 2.220446049250313e-16 : out->eps;
    int chunk_base = 0;
    for (size_t si = 0; si < subs.size(); ++si) {
      const Sub& sb = subs[si];
      const auto& tt = *sb.t;
      native::RowArgs<T> a{};
      a.Lf = tt.Lf;
      a.n = n;
      a.N1 = sb.N1;
      a.N2 = native::kN2;
      a.panel_bands = (int32_t)B;
      a.imd_slots = tt.imd_slots;
      a.chunk_total = chunk_total;
      a.X = si == 0 ? X : Xn;
      a.Hc = static_cast<const cplx<T>*>(tt.Hc);
      a.Hfull = static_cast<const cplx<T>*>(tt.Hfull);
      a.imd = imd;
      a.inv_len = 1.0 / (double)tt.Lf;
      a.two_over_len = (float)(2.0 / (double)tt.Lf);
      a.neg_last_row = sb.kernel_kind == 0 ? 1 : 0;
      a.coef = coef;
      a.bits = bits;
      a.edge_z = edge_z;
      a.edge_wmax = p->edge_wmax;
      a.nedge = p->nedge;
      a.time_part = tpart;
      a.part_band = want_band ? part_band : nullptr;
      a.part_stat = want_stat ? part_stat : nullptr;
      a.nblk = nbk;
      a.stat_nblk = nblk;
      a.stat_stride = stat_slots;
      a.power_scale = power_scale;
      a.eps = eps;
      for (size_t g = 0; g < tt.groups.size(); ++g) {
        const auto& grp = tt.groups[g];
        a.bands = tt.d_bands + grp.first;
        a.nbands = grp.count;
        a.gen_list = tt.d_gen_list ? tt.d_gen_list + grp.gen_first : nullptr;
        a.ngen_launch = grp.ngen;
        a.chunk_base = chunk_base;
        if (grp.ngen > 0) {
          p->prof.begin(st, QI_STAGE_PASS1);
          QI_TRY(native::launch_pass1<T>(a, sb.kernel_kind, ct, st));
          p->prof.end(QI_STAGE_PASS1, st);
        }
        p->prof.begin(st, QI_STAGE_PASS2);
        QI_TRY(native::launch_pass2<T>(a, sb.kernel_kind, G, sb.nchunk[g], ct, st));
        p->prof.end(QI_STAGE_PASS2, st);
        chunk_base += sb.nchunk[g];
      }
    }
    for (int g = 0; g < native::kZ64Levels; ++g) {
      if (t.z64_count[g] == 0) continue;
      native::Z64Args z{};
      z.Lf = Lf;
      z.n = n;
      z.log2d = 6 - g;
      z.M = Lf >> z.log2d;
      z.kind = kind;
      z.nbands = t.z64_count[g];
      z.panel_bands = (int32_t)B;
      z.bands = t.d_z64 + t.z64_first[g];
      z.X = X;
      z.Hc = static_cast<const cplx<T>*>(t.Hc);
      z.Z = Z;
      z.weights = p->d_z64_w[g];
      z.inv_len = 1.0 / (double)Lf;
      z.two_over_len = (float)(2.0 / (double)Lf);
      z.coef = coef;
      z.bits = bits;
      z.time_part = tpart;
      z.part_band = want_band ? part_band : nullptr;
      z.part_stat = want_stat ? part_stat : nullptr;
      z.nblk = nblk_z;
      z.pb_stride = nbk;
      z.stat_nblk = nblk;
      z.stat_stride = stat_slots;
      z.chunk_base = chunk_base;
      z.chunk_total = chunk_total;
      z.power_scale = power_scale;
      z.eps = eps;
      p->prof.begin(st, QI_STAGE_ZOOM_COARSE);
      QI_TRY(native::launch_z64_gather(z, ct, st));
      QI_TRY(fft_c2c<T>(p->fft, Z, z.M, (int64_t)z.nbands * ct, HIPFFT_BACKWARD, st));
      p->prof.end(QI_STAGE_ZOOM_COARSE, st);
      p->prof.begin(st, QI_STAGE_ZOOM);
      QI_TRY(native::launch_z64_interp(z, zchunk[g], ct, st));
      p->prof.end(QI_STAGE_ZOOM, st);
      chunk_base += zchunk[g];
    }
    if (blocks) {
      native::BlockArgs<T> b{};
      b.n = n;
      b.nitems = il.nitems;
      b.panel_bands = (int32_t)B;
      b.items = il.d_items;
      b.bands = static_cast<const native::BlockBandT<T>*>(il.d_bands);
      b.bank = static_cast<const cplx<T>*>(bt.bank);
      b.sig = sig + c0 * n;
      b.coef = coef;
      b.bits = bits;
      b.time_part = tpart;
      b.part_band = want_band ? part_band : nullptr;
      b.part_stat = want_stat ? part_stat : nullptr;
      b.nblk = nbk;
      b.stat_stride = stat_slots;
      b.stat_base = blk_stat_base;
      b.chunk_base = chunk_blk;
      b.chunk_total = chunk_total;
      b.power_scale = power_scale;
      b.eps = eps;
      b.two_over_n = (float)(2.0 / (double)n);
      p->prof.begin(st, QI_STAGE_BLOCK);
      QI_TRY(native::launch_block<T>(b, bt.demod, ct, st));
      p->prof.end(QI_STAGE_BLOCK, st);
    }
    p->prof.begin(st, QI_STAGE_EPILOGUE);
    if (shorts) {
      native::EdgeArgs<T> e{};
      e.bands = p->d_edge;
      e.nedge = p->nedge;
      e.panel_bands = (int32_t)B;
      e.n = n;
      e.wmax = p->edge_wmax;
      e.stat_slots = stat_slots;
      e.sig = sig + c0 * n;
      e.coef = coef;
      e.edge_z = edge_z;
      e.bits = bits;
      e.edge_p = edge_p;
      e.power_scale = power_scale;
      e.eps = eps;
      QI_TRY(native::launch_edge<T>(e, ct, want_time ? edge_time : nullptr, want_band ? part_band : nullptr, nbk, nbk - 1,
                                    want_stat ? part_stat : nullptr, stat_slots - p->nedge, st));
    }
    double* pb_out = want_band ? static_cast<double*>(out->power_band) + c0 * B : nullptr;
    double* st_out = want_stat ? static_cast<double*>(out->stats) + c0 * 4 : nullptr;
    if (time_via_part && (want_band || want_stat)) {
      QI_TRY(native::launch_tail<T>(time_part, static_cast<T*>(out->power_time) + c0 * n, ct, n, chunk_total,
                                    shorts ? edge_time : nullptr, p->edge_wmax, want_band ? part_band : nullptr,
                                    want_stat ? part_stat : nullptr, pb_out, st_out, B, nbk, stat_slots, nullptr, st));
    } else {
      if (time_via_part)
        QI_TRY(native::launch_time_reduce<T>(time_part, static_cast<T*>(out->power_time) + c0 * n, ct, n, chunk_total,
                                             shorts ? edge_time : nullptr, p->edge_wmax, st));
      if (want_band || want_stat)
        QI_TRY(launch_finalize(want_band ? part_band : nullptr, want_stat ? part_stat : nullptr, pb_out, st_out, ct, B, nbk,
                               stat_slots, st, nullptr));
    }
    p->prof.end(QI_STAGE_EPILOGUE, st);
    p->prof.unchain();
  }
  return QI_OK;
}

template <typename T>
int build_bank(qi_plan* p, int bank, int32_t B, const double* d_par, hipStream_t st) {
  const int64_t n = p->n;
  const int circular = bank == QI_BANK_ATOMS;
  const int64_t L = circular ? n : p->L;
  const size_t row64 = (size_t)L * sizeof(double2);
  int64_t chunk = (int64_t)(p->ws_bytes / row64);
  if (chunk < 1) {
    set_error("workspace too small to build one bank row (%zu bytes needed)", row64);
    return QI_ERR_NOMEM;
  }
  if (chunk > B) chunk = B;
  double2* rows = reinterpret_cast<double2*>(p->ws);
  cplx<T>* dst = static_cast<cplx<T>*>(p->bank[bank]);
  for (int32_t j0 = 0; j0 < B; j0 += (int32_t)chunk) {
    const int nbk = (B - j0 < chunk) ? B - j0 : (int)chunk;
    QI_TRY(launch_bank_rows(rows, n, L, circular, d_par, d_par + B, d_par + 2 * B, d_par + 3 * B, j0, nbk, st));
    QI_TRY(fft_c2c<double>(p->fft, rows, L, nbk, HIPFFT_FORWARD, st));
    QI_TRY(launch_bank_convert<T>(rows, dst + (int64_t)j0 * L, (int64_t)nbk * L, circular, 1.0 / (double)L, st));
  }
  return QI_OK;
}

template <typename T>
int stft_impl(int device, const void* sig, int64_t C, int64_t n, const void* window, int64_t seg, int64_t hop,
                     int64_t nfft, double scale, void* Z, void* bits, double eps, char* scratch, hipStream_t st) {
  const int64_t nseg = qi_stft_segments(n, seg, hop);
  const int64_t nf = nfft / 2 + 1;
  static const bool fused_off = tune_env("QI_STFT_FUSED") && atoi(tune_env("QI_STFT_FUSED")) == 0;
  if (!fused_off && stft_fused_supported(sizeof(T) == 8 ? QI_F64 : QI_F32, seg, hop, nfft))  // one kernel: segments, transform and store from LDS
    return launch_stft_fused<T>(static_cast<const T*>(sig), static_cast<const T*>(window), static_cast<cplx<T>*>(Z),
                                static_cast<T*>(bits), C, n, seg, hop, nfft, nseg, seg / 2, scale,
                                eps == 0.0 ? 2.220446049250313e-16 : eps, st);
  T* frames = reinterpret_cast<T*>(scratch);
  cplx<T>* F = reinterpret_cast<cplx<T>*>(scratch + align_up((size_t)C * nseg * nfft * sizeof(T)));
  QI_TRY(launch_stft_frames<T>(static_cast<const T*>(sig), static_cast<const T*>(window), frames, C, n, seg, hop,
                               nfft, nseg, seg / 2, st));
  {
    std::lock_guard<std::mutex> lk(g_stft_mu);
    QI_TRY(fft_r2c<T>(g_stft_fft[device], frames, F, nfft, C * nseg, st));
  }
  return launch_stft_transpose<T>(F, static_cast<cplx<T>*>(Z), static_cast<T*>(bits), C, nseg, nf, (T)scale,
                                  (T)(eps == 0.0 ? 2.220446049250313e-16 : eps), st);
}

template <typename T>
int welch_impl(int device, const void* sig, int64_t C, int64_t n, const void* window, int64_t seg, int64_t hop,
               int64_t nfft, double scale, void* pxx, char* scratch, hipStream_t st) {
  const int64_t nseg = (n - seg) / hop + 1;
  const int64_t nf = nfft / 2 + 1;
  T* frames = reinterpret_cast<T*>(scratch);
  cplx<T>* F = reinterpret_cast<cplx<T>*>(scratch + align_up((size_t)C * nseg * nfft * sizeof(T)));
  QI_TRY(launch_stft_frames<T>(static_cast<const T*>(sig), static_cast<const T*>(window), frames, C, n, seg, hop,
                               nfft, nseg, 0, st));
  {
    std::lock_guard<std::mutex> lk(g_stft_mu);
    QI_TRY(fft_r2c<T>(g_stft_fft[device], frames, F, nfft, C * nseg, st));
  }
  return launch_welch_mean<T>(F, static_cast<T*>(pxx), C, nseg, nf, nfft, (T)(scale * scale), st);
}

}  // namespace

namespace {
template <typename T>
int sliding_stft_impl(int device, const T* sig, int64_t C, int64_t n, const T* window, int64_t seg, int64_t hop,
                      int64_t nfft, int64_t first, int64_t nseg, int pad_mode, int detrend, int64_t roll, cplx<T>* Z, T* R,
                      int kind, char* scratch, hipStream_t st) {
  const int64_t nf = nfft / 2 + 1;
  T* frames = reinterpret_cast<T*>(scratch);
  cplx<T>* F = reinterpret_cast<cplx<T>*>(scratch + align_up((size_t)C * nseg * nfft * sizeof(T)));
  QI_TRY(launch_sliding_frames<T>(sig, window, frames, C, n, seg, hop, nfft, nseg, first, pad_mode, detrend, roll, st));
  {
    std::lock_guard<std::mutex> lk(g_stft_mu);
    QI_TRY(fft_r2c<T>(g_stft_fft[device], frames, F, nfft, C * nseg, st));
  }
  return launch_sliding_transpose<T>(F, Z, R, kind, C, nseg, nf, st);
}

template <typename T>
int sliding_istft_impl(int device, const cplx<T>* S, int64_t C, const T* dual, int64_t seg, int64_t hop, int64_t nfft,
                       int64_t first, int64_t nseg, int64_t roll, int64_t k0, int64_t k1, T* out, char* scratch,
                       hipStream_t st) {
  const int64_t nf = nfft / 2 + 1;
  T* slices = reinterpret_cast<T*>(scratch);
  cplx<T>* F = reinterpret_cast<cplx<T>*>(scratch + align_up((size_t)C * nseg * nfft * sizeof(T)));
  QI_TRY(launch_sliding_untranspose<T>(S, F, C, nseg, nf, st));
  {
    std::lock_guard<std::mutex> lk(g_stft_mu);
    QI_TRY(fft_c2r<T>(g_stft_fft[device], F, slices, nfft, C * nseg, st));
  }
  return launch_sliding_overlap_add<T>(slices, dual, out, C, k0, k1, seg, hop, nfft, nseg, first, roll, st);
}
}  // namespace

namespace {
template <typename T>
int shannon_fft_impl(int device, const T* sig, int64_t C, int64_t n, cplx<T>* spectrum, T* angle, T* marginal,
                     char* scratch, hipStream_t st) {
  const int64_t nf = n / 2 + 1;
  double* partial = reinterpret_cast<double*>(scratch);
  int32_t* turns = reinterpret_cast<int32_t*>(scratch + align_up((size_t)C * shannon_spans(n) * 8));
  T* copy = reinterpret_cast<T*>(scratch + align_up((size_t)C * shannon_spans(n) * 8) + align_up((size_t)C * nf * 4));
  QI_HIP(hipMemcpyAsync(copy, sig, (size_t)C * n * sizeof(T), hipMemcpyDeviceToDevice, st));
  {
    std::lock_guard<std::mutex> lock(g_stft_mu);
    QI_TRY(fft_r2c<T>(g_stft_fft[device], copy, spectrum, n, C, st));
  }
  return launch_fft_marginal<T>(spectrum, C, nf, angle, marginal, partial, turns, st);
}
}  // namespace

extern "C" {

int qi_abi_version(void) { return QI_TFR_ABI_VERSION; }
const char* qi_last_error(void) { return g_err; }

int qi_device_info(int device, char* name, size_t name_len, int64_t* hbm_bytes, int32_t* compute_units) {
  hipDeviceProp_t prop;
  QI_HIP(hipGetDeviceProperties(&prop, device));
  if (name && name_len) snprintf(name, name_len, "%s (%s)", prop.name, prop.gcnArchName);
  if (hbm_bytes) *hbm_bytes = (int64_t)prop.totalGlobalMem;
  if (compute_units) *compute_units = prop.multiProcessorCount;
  return QI_OK;
}

int qi_plan_create(qi_plan** plan, const qi_plan_desc* desc) {
  QI_REQUIRE(plan && desc, "null plan/desc");
  *plan = nullptr;
  QI_REQUIRE(desc->n >= 2 && desc->n <= (1ll << 28), "n = %lld out of range", (long long)desc->n);
  QI_REQUIRE(desc->dtype == QI_F32 || desc->dtype == QI_F64, "bad dtype %d", desc->dtype);
  QI_REQUIRE(desc->engine >= QI_ENGINE_AUTO && desc->engine <= QI_ENGINE_NATIVE, "bad engine %d", desc->engine);
  if (desc->engine == QI_ENGINE_NATIVE && !(is_pow2(desc->n) && desc->n >= (1 << 18) &&
                                            (desc->dtype == QI_F32 || desc->n == (1 << 20)))) {
    set_error("native engine: float32 records of a power-of-two length >= 2^18, float64 records of 2^20 samples (got n = %lld, dtype %d)",
              (long long)desc->n, desc->dtype);
    return QI_ERR_UNSUPPORTED;
  }
  DeviceGuard g(desc->device);
  if (!g.ok) {
    set_error("hipSetDevice(%d) failed", desc->device);
    return QI_ERR_HIP;
  }
  qi_plan* p = new (std::nothrow) qi_plan();
  QI_REQUIRE(p, "out of host memory");
  p->d = *desc;
  p->n = desc->n;
  // scipy.signal.fftconvolve pads to next_fast_len(2n-1) (= 2n when n = 2^k); any L >= 2n-1 gives the
  // same linear correlation, so other n use the next power of two.
  p->L = is_pow2(desc->n) ? 2 * desc->n : next_pow2(2 * desc->n - 1);
  if (const char* e = tune_env("QI_NATIVE_KMAX")) {
    const long v = atol(e);
    if (v >= 0) p->native_kmax = v;
  }
  if (p->native_kmax > (int64_t)native::kMaxPrunedTerms * native::kN2)
    p->native_kmax = (int64_t)native::kMaxPrunedTerms * native::kN2;
  if (const char* e = tune_env("QI_NATIVE_DEBUG")) p->native_debug = atoi(e);
  if (const char* e = tune_env("QI_NATIVE_GROUP")) p->native_group = atoi(e);
  if (const char* e = tune_env("QI_NATIVE_WGS")) p->native_wgs = atoi(e) > 0 ? atoi(e) : 256;
  if (const char* e = tune_env("QI_NATIVE_FWD")) p->native_fwd = atoi(e);
#ifdef QI_NATIVE_STAMPS
  if (tune_env("QI_NATIVE_STAMPS")) {
    if (hipMalloc((void**)&p->stamps, 65536 * 8 * sizeof(unsigned long long)) != hipSuccess) p->stamps = nullptr;
    if (p->stamps) (void)hipMemset(p->stamps, 0, 65536 * 8 * sizeof(unsigned long long));
    if (hipMalloc((void**)&p->blk_stamps, 65536 * 8 * sizeof(unsigned long long)) != hipSuccess) p->blk_stamps = nullptr;
    if (p->blk_stamps) (void)hipMemset(p->blk_stamps, 0, 65536 * 8 * sizeof(unsigned long long));
  }
#endif
  if (const char* e = tune_env("QI_NATIVE_SHORT")) p->native_short = atoi(e);
  if (const char* e = tune_env("QI_NATIVE_BLOCK")) p->native_block = atoi(e);
  if (const char* e = tune_env("QI_NATIVE_ZOOM")) p->native_zoom = atoi(e);
  if (const char* e = tune_env("QI_NATIVE_ZOOM_LEVELS")) p->native_zoom_max_level = atoi(e);
  if (const char* e = tune_env("QI_NATIVE_ZOOM_SHORT")) p->native_zoom_short = atoi(e);
  if (const char* e = tune_env("QI_NATIVE_ZOOM_SHORT_FROM")) p->native_zoom_short_from = atoi(e);
  if (const char* e = tune_env("QI_NATIVE_ZOOM_WGS")) p->native_zoom_wgs = atoi(e);
  if (const char* e = tune_env("QI_NATIVE_ZOOM_WGS_JOINT")) p->native_zoom_wgs_joint = atoi(e);
  if (const char* e = tune_env("QI_NATIVE_ZOOM_WAVES")) p->native_zoom_waves = atoi(e) > 0 ? atoi(e) : p->native_zoom_waves;
  if (const char* e = tune_env("QI_NATIVE_BLK_ANALYTIC")) p->native_blk_analytic = atoi(e);
  if (const char* e = tune_env("QI_NATIVE_OVERLAP")) p->native_overlap = atoi(e);
  if (const char* e = tune_env("QI_NATIVE_PAIR")) p->native_pair = atoi(e);
  if (const char* e = tune_env("QI_NATIVE_GATHER_FUSED")) p->native_gather_fused = atoi(e);
  if (const char* e = tune_env("QI_NATIVE_SPLIT")) p->native_split = atoi(e);
  if (const char* e = tune_env("QI_NATIVE_SPLIT_E")) p->native_split_e = atoll(e);
  if (const char* e = tune_env("QI_NATIVE_FUSE")) p->native_fuse = atoi(e);
  if (const char* e = tune_env("QI_NATIVE_BLK_NARROW")) p->native_blk_narrow = atoi(e);
  if (const char* e = tune_env("QI_NATIVE_TAIL")) p->native_tail = atoi(e);
  if (const char* e = tune_env("QI_NATIVE_TILE")) p->native_tile = atoll(e);
  if (const char* e = tune_env("QI_NATIVE_BLK_MAXWQ")) p->native_blk_maxwq = atoi(e);
  if (const char* e = tune_env("QI_NATIVE_BLK_BANDS")) p->native_blk_bands = atoi(e) > 0 ? atoi(e) : p->native_blk_bands;
  if (const char* e = tune_env("QI_NATIVE_BLK_BANDS_BATCH")) p->native_blk_bands_batch = atoi(e) > 0 ? atoi(e) : p->native_blk_bands_batch;
  if (const char* e = tune_env("QI_NATIVE_BLK_BATCH_FROM")) p->native_blk_batch_from = atoi(e);
  if (const char* e = tune_env("QI_NATIVE_BLK_HALF")) p->native_blk_half = atoi(e);
  if (const char* e = tune_env("QI_NATIVE_BLK_LONG")) p->native_blk_long = atoi(e);
  if (const char* e = tune_env("QI_NATIVE_BLK_FASTW")) p->native_blk_fastw = atoi(e);
#ifdef QI_BLK_LZ
  if (const char* e = tune_env("QI_NATIVE_BLK_LZ")) p->native_blk_lz = atoi(e);
#endif
  if (const char* e = tune_env("QI_NATIVE_ROWS")) {
    const long v = atol(e);
    if (v == 8 || v == 16) p->native_rows = (int)v;
  }
  if (const char* e = tune_env("QI_NATIVE_F64")) p->native_f64 = atoi(e);
  if (const char* e = tune_env("QI_NATIVE_Z64")) p->native_z64 = atoi(e);
  if (const char* e = tune_env("QI_NATIVE_Z64_LEVELS")) {
    const int v = atoi(e);
    if (v >= 1 && v <= native::kZ64Levels) p->native_z64_levels = v;
  }
  if (desc->dtype == QI_F64) {  // the float32 zoom / block / split engines are sized for the float32 tolerance
    const int short64 = p->native_short && !(tune_env("QI_NATIVE_SHORT64") && atoi(tune_env("QI_NATIVE_SHORT64")) == 0);
    // (the block engine runs float64 tables in double arithmetic: analytic Gaussian bands, no narrow-spectrum shortcuts)
    const int block64 = p->native_block && !(tune_env("QI_NATIVE_BLOCK64") && atoi(tune_env("QI_NATIVE_BLOCK64")) == 0);
    p->native_zoom = p->native_split = 0;
    p->native_block = block64;
    p->native_short = short64;  // wide-spectrum, short-atom styx bands as circular correlations of length n + edge fix
    p->native_rows = 8;
    if (p->native_group <= 0) p->native_group = 8;  // wide bands per launch group: bounds the intermediate (32 MB per band and record)
  }
  p->ws_bytes = desc->workspace_bytes > 0 ? (size_t)desc->workspace_bytes : ((size_t)2 << 30);
  if (hipMalloc((void**)&p->ws, p->ws_bytes) != hipSuccess) {
    set_error("hipMalloc of %zu workspace bytes failed", p->ws_bytes);
    delete p;
    return QI_ERR_HIP;
  }
  *plan = p;
  return QI_OK;
}

int qi_plan_destroy(qi_plan* p) {
  if (!p) return QI_OK;
  DeviceGuard g(p->d.device);
  (void)hipDeviceSynchronize();
  p->fft.clear();
  p->prof.clear();
#ifdef QI_NATIVE_STAMPS
  if (p->stamps) {
    std::vector<unsigned long long> h(65536 * 8);
    if (hipMemcpy(h.data(), p->stamps, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost) == hipSuccess) {
      double sum[8] = {0};
      long cnt = 0;
      for (size_t w = 0; w < 65536; ++w) {
        unsigned long long tot = 0;
        for (int k = 0; k < 8; ++k) tot += h[w * 8 + k];
        if (!tot) continue;
        ++cnt;
        for (int k = 0; k < 8; ++k) sum[k] += (double)h[w * 8 + k];
      }
      fprintf(stderr, "[qi stamps] last pass-2 launch, %ld workgroups, mean cycles per workgroup: load %.0f | barrier %.0f | "
              "step1 %.0f | barrier %.0f | exchange %.0f | step2 %.0f | epilogue %.0f | loop head %.0f\n", cnt,
              sum[0] / cnt, sum[1] / cnt, sum[2] / cnt, sum[3] / cnt, sum[4] / cnt, sum[5] / cnt, sum[6] / cnt, sum[7] / cnt);
    }
    (void)hipFree(p->stamps);
  }
  if (p->blk_stamps) {
    std::vector<unsigned long long> h(65536 * 8);
    if (hipMemcpy(h.data(), p->blk_stamps, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost) == hipSuccess) {
      double sum[8] = {0};
      long cnt = 0;
      for (size_t w = 0; w < 65536; ++w) {
        if (!h[w * 8 + 5]) continue;
        ++cnt;
        for (int k = 0; k < 8; ++k) sum[k] += (double)h[w * 8 + k];
      }
      if (cnt)
        fprintf(stderr, "[qi stamps] last block launch, %ld workgroups, %.2f bands each; mean cycles per workgroup: prologue "
                "(load + forward) %.0f | per band: filter loads issued %.0f | wait for them %.0f | multiply + inverse "
                "transform %.0f | epilogue %.0f\n", cnt, sum[5] / cnt, sum[0] / cnt, sum[1] / sum[5], sum[2] / sum[5],
                sum[3] / sum[5], sum[4] / sum[5]);
    }
    (void)hipFree(p->blk_stamps);
  }
#endif
  for (auto& t : p->nat) t.release();
  for (auto& t : p->blk) t.release();
  if (p->side) (void)hipStreamDestroy(p->side);
  if (p->ev_fork) (void)hipEventDestroy(p->ev_fork);
  if (p->ev_join) (void)hipEventDestroy(p->ev_join);
  for (auto& per_cut : p->d_band_slots)
    for (auto* b : per_cut)
      if (b) (void)hipFree(b);
  for (auto* w : p->d_z64_w)
    if (w) (void)hipFree(w);
  for (auto& wc : p->d_zoom_w)
    for (auto* w : wc)
      if (w) (void)hipFree(w);
  if (p->d_edge) (void)hipFree(p->d_edge);
  if (p->split_bank) (void)hipFree(p->split_bank);
  if (p->d_lz_w) (void)hipFree(p->d_lz_w);
  if (p->d_split_bands) (void)hipFree(p->d_split_bands);
  for (auto* d : p->d_dual)
    if (d) (void)hipFree(d);
  for (int b = 0; b < 2; ++b)
    if (p->bank[b]) (void)hipFree(p->bank[b]);
  if (p->d_stx_idx) (void)hipFree(p->d_stx_idx);
  if (p->d_stx_coef) (void)hipFree(p->d_stx_coef);
  if (p->ws) (void)hipFree(p->ws);
  delete p;
  return QI_OK;
}

int qi_plan_set_gabor_bank(qi_plan* p, int bank, int32_t B, const double* p_re, const double* p_im,
                           const double* omega, const double* amp, qi_stream stream) {
  QI_REQUIRE(p && p_re && p_im && omega && amp, "null argument");
  QI_REQUIRE(bank == QI_BANK_STYX || bank == QI_BANK_ATOMS, "bad bank %d", bank);
  QI_REQUIRE(B > 0 && B <= 65535, "band count %d out of range", B);
  DeviceGuard g(p->d.device);
  hipStream_t st = (hipStream_t)stream;
  const int64_t L = bank == QI_BANK_ATOMS ? p->n : p->L;
  const size_t esz = p->d.dtype == QI_F64 ? sizeof(double2) : sizeof(float2);
  if (p->bank[bank]) {
    QI_HIP(hipDeviceSynchronize());
    QI_HIP(hipFree(p->bank[bank]));
    p->bank[bank] = nullptr;
    p->nb[bank] = 0;
  }
  bool use_native = native_wanted(p, bank);
  if (!use_native && p->d.engine == QI_ENGINE_NATIVE) {
    set_error("native engine does not support this bank at n = %lld", (long long)p->n);
    return QI_ERR_UNSUPPORTED;
  }
  p->nat[bank].release();
  if (bank == QI_BANK_STYX) p->blk[0].release();
  for (auto*& b : p->d_band_slots[bank]) {
      if (b) (void)hipFree(b);
    b = nullptr;
  }
  double* d_par = nullptr;
  QI_HIP(hipMalloc((void**)&d_par, (size_t)4 * B * sizeof(double)));
  std::vector<double> host((size_t)4 * B);
  memcpy(&host[0], p_re, B * sizeof(double));
  memcpy(&host[B], p_im, B * sizeof(double));
  memcpy(&host[2 * B], omega, B * sizeof(double));
  memcpy(&host[3 * B], amp, B * sizeof(double));
  int rc = QI_OK;
  if (hipMemcpy(d_par, host.data(), host.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) {
    set_error("hipMemcpy of band parameters failed");
    rc = QI_ERR_HIP;
  }
  if (rc == QI_OK && use_native) {
    rc = p->d.dtype == QI_F64 ? build_native_bank<double>(p, bank, B, d_par, host.data(), st)
                              : build_native_bank<float>(p, bank, B, d_par, host.data(), st);
    // The zoom and block engines take any power-of-two record from 2^18 samples; the two-pass kernels run
    // 2^20 / 2^21-point transforms only.  A table that still has bands for them at another length goes to the hipFFT engine.
    if (rc == QI_OK && !native_len_ok(L) && !p->nat[bank].h_rows.empty()) {
      (void)hipStreamSynchronize(st);
      p->nat[bank].release();
      if (bank == QI_BANK_STYX) {
        p->blk[0].release();
        p->nat[3].release();
        p->nsplit = 0;
      }
      use_native = false;
      if (p->d.engine == QI_ENGINE_NATIVE) {
        set_error("native engine: this band table needs the two-pass kernels, which run 2^20 / 2^21-point transforms only");
        rc = QI_ERR_UNSUPPORTED;
      }
    }
  }
  if (rc == QI_OK && !use_native) {
    if (hipMalloc(&p->bank[bank], (size_t)B * L * esz) != hipSuccess) {
      set_error("hipMalloc of the %zu-byte atom-spectrum bank failed", (size_t)B * L * esz);
      rc = QI_ERR_NOMEM;
    } else {
      rc = p->d.dtype == QI_F64 ? build_bank<double>(p, bank, B, d_par, st) : build_bank<float>(p, bank, B, d_par, st);
    }
  }
  if (rc == QI_OK && hipStreamSynchronize(st) != hipSuccess) {
    set_error("bank build failed on the device: %s", hipGetErrorString(hipGetLastError()));
    rc = QI_ERR_HIP;
  }
  (void)hipFree(d_par);
  if (rc == QI_OK) {
    p->nb[bank] = B;
  } else {  // nothing half-built stays behind (a ready table without its block / split producers would leave rows unwritten)
    p->nat[bank].release();
    if (bank == QI_BANK_STYX) {
      p->blk[0].release();
      p->nat[3].release();
      p->nsplit = 0;
    }
    if (p->bank[bank]) {
      (void)hipFree(p->bank[bank]);
      p->bank[bank] = nullptr;
    }
  }
  return rc;
}

int qi_gabor_atoms(int device, int64_t n, int32_t B, const double* p_re, const double* p_im, const double* omega,
                   const double* amp, void* out, qi_stream stream) {
  QI_REQUIRE(p_re && p_im && omega && amp && out, "null argument");
  QI_REQUIRE(n >= 2 && B > 0 && B <= 65535, "bad atom bank shape");
  DeviceGuard g(device);
  double* d_par = nullptr;
  QI_HIP(hipMalloc((void**)&d_par, (size_t)4 * B * sizeof(double)));
  std::vector<double> host((size_t)4 * B);
  memcpy(&host[0], p_re, B * sizeof(double));
  memcpy(&host[B], p_im, B * sizeof(double));
  memcpy(&host[2 * B], omega, B * sizeof(double));
  memcpy(&host[3 * B], amp, B * sizeof(double));
  int rc = QI_OK;
  if (hipMemcpy(d_par, host.data(), host.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) {
    set_error("hipMemcpy of band parameters failed");
    rc = QI_ERR_HIP;
  }
  hipStream_t st = (hipStream_t)stream;
  if (rc == QI_OK)
    rc = launch_bank_rows((double2*)out, n, n, 1, d_par, d_par + B, d_par + 2 * B, d_par + 3 * B, 0, B, st);
  if (rc == QI_OK && hipStreamSynchronize(st) != hipSuccess) {
    set_error("atom kernel failed: %s", hipGetErrorString(hipGetLastError()));
    rc = QI_ERR_HIP;
  }
  (void)hipFree(d_par);
  return rc;
}

int qi_gabor_atoms_at(int device, int64_t n, int32_t B, const double* p_re, const double* p_im, const double* omega,
                      const double* amp, const void* x, void* out, qi_stream stream) {
  QI_REQUIRE(p_re && p_im && omega && amp && x && out, "null argument");
  QI_REQUIRE(n >= 1 && B > 0 && B <= 65535, "bad atom bank shape");
  DeviceGuard g(device);
  double* d_par = nullptr;
  QI_HIP(hipMalloc((void**)&d_par, (size_t)4 * B * sizeof(double)));
  std::vector<double> host((size_t)4 * B);
  memcpy(&host[0], p_re, B * sizeof(double));
  memcpy(&host[B], p_im, B * sizeof(double));
  memcpy(&host[2 * B], omega, B * sizeof(double));
  memcpy(&host[3 * B], amp, B * sizeof(double));
  int rc = QI_OK;
  if (hipMemcpy(d_par, host.data(), host.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) {
    set_error("hipMemcpy of band parameters failed");
    rc = QI_ERR_HIP;
  }
  hipStream_t st = (hipStream_t)stream;
  if (rc == QI_OK)
    rc = launch_bank_rows((double2*)out, n, n, 1, d_par, d_par + B, d_par + 2 * B, d_par + 3 * B, 0, B, st, 0.0,
                          static_cast<const double*>(x));
  if (rc == QI_OK && hipStreamSynchronize(st) != hipSuccess) {
    set_error("atom kernel failed: %s", hipGetErrorString(hipGetLastError()));
    rc = QI_ERR_HIP;
  }
  (void)hipFree(d_par);
  return rc;
}

int qi_plan_set_stx_bands(qi_plan* p, int32_t B, const int64_t* shift_index, const double* sigma) {
  QI_REQUIRE(p && shift_index && sigma, "null argument");
  QI_REQUIRE(B > 0 && B <= 65535, "band count %d out of range", B);
  for (int32_t j = 0; j < B; ++j)
    QI_REQUIRE(shift_index[j] >= 0 && shift_index[j] < p->n, "shift_index[%d] = %lld outside [0, n)", j,
               (long long)shift_index[j]);
  DeviceGuard g(p->d.device);
  if (p->d_stx_idx) {
    QI_HIP(hipDeviceSynchronize());
    QI_HIP(hipFree(p->d_stx_idx));
    QI_HIP(hipFree(p->d_stx_coef));
    p->d_stx_idx = nullptr;
    p->d_stx_coef = nullptr;
    p->nb_stx = 0;
  }
  std::vector<double> coef(B);
  const double k = 2.0 * M_PI / (double)p->n * std::sqrt(0.5 * M_LOG2E);
  for (int32_t j = 0; j < B; ++j) coef[j] = sigma[j] * k;
  QI_HIP(hipMalloc((void**)&p->d_stx_idx, B * sizeof(int64_t)));
  QI_HIP(hipMalloc((void**)&p->d_stx_coef, B * sizeof(double)));
  QI_HIP(hipMemcpy(p->d_stx_idx, shift_index, B * sizeof(int64_t), hipMemcpyHostToDevice));
  QI_HIP(hipMemcpy(p->d_stx_coef, coef.data(), B * sizeof(double), hipMemcpyHostToDevice));
  p->nb_stx = 0;  // committed below, once every table of the native engine has been built
  p->nat[2].release();
  p->blk[2].release();
  for (auto*& b : p->d_band_slots[2]) {
      if (b) (void)hipFree(b);
    b = nullptr;
  }
  if (native_wanted(p, 2)) {
    // support of exp2(-(coef k)^2) above 2^-30: |k| <= sqrt(30) / coef (float64: above 2^-50)
    const double cut = p->d.dtype == QI_F64 ? std::sqrt(50.0) : std::sqrt(30.0);
    std::vector<native::BandDesc> bands;
    std::vector<BlockPick> picks;
    const bool can_block = p->native_block && p->n >= 4 * native::kBlk;
    int32_t ngen = 0;
    (void)ngen;
    for (int32_t j = 0; j < B; ++j) {
      // the band's time-domain kernel is a Gaussian of standard deviation sigma_j samples (above 2^-30 of its peak
      // within sqrt(60 ln 2) sigma); it is only that short if the frequency window has decayed before Nyquist
      const double reach = std::ceil(std::sqrt((p->d.dtype == QI_F64 ? 104.0 : 60.0) * M_LN2) * sigma[j]) + 1.0;
      const double kh0 = std::floor(std::sqrt(30.0) / coef[j]);
      const bool zoom_first = block_group_of(reach) > p->native_blk_maxwq && 2 * kh0 + 1 < (double)p->n &&
                              zoom_class(p, 2, p->n, (int64_t)(2 * kh0 + 1)) >= 0;
      const double kh64 = std::floor(cut / coef[j]);
      const bool z64_first = p->d.dtype == QI_F64 && z64_table(p, 2) && 2 * kh64 + 1 <= (double)narrow_limit(p, 2, p->n) &&
                             2 * kh64 + 1 < (double)p->n;
      if (can_block && sigma[j] >= 2.75 && block_group_of(reach) > 0 && !zoom_first && !z64_first) {
        BlockPick pk{j, block_group_of(reach), shift_index[j]};
        // the band's filter spectrum is the Gaussian window itself, centred on the band's shift index
        pk.analytic = 1;
        pk.kappa = (double)shift_index[j] * (double)native::kBlk / (double)p->n;
        pk.cw = (2.0 * M_PI / (double)native::kBlk) * sigma[j] * std::sqrt(M_LOG2E / 2.0);
        pk.amp = 1.0 / (double)native::kBlk;
        picks.push_back(pk);
        continue;
      }
      bands.emplace_back();
      native::BandDesc& d = bands.back();
      memset(&d, 0, sizeof(d));
      d.shift = shift_index[j];
      d.coef = coef[j];
      d.out_band = j;
      const double kh = std::floor(cut / coef[j]);
      const int zc = 2 * kh + 1 < (double)p->n ? zoom_class(p, 2, p->n, (int64_t)(2 * kh + 1)) : -1;
      if (zc >= 0 || (2 * kh + 1 <= (double)narrow_limit(p, 2, p->n) && 2 * kh + 1 < (double)p->n)) {
        d.mode = zc >= 0 ? 2 + zc : 0;
        d.k_lo = -(int32_t)kh;
        d.k_len = 2 * (int32_t)kh + 1;
      } else {
        d.mode = 1;
        ngen++;
      }
    }
    bool two_pass_free = true;  // no band for pass 1 / pass 2 (their transform lengths are 2^20 and 2^21 only)
    for (const auto& d : bands) two_pass_free = two_pass_free && d.mode >= 2;
    if (native_len_ok(p->n) || two_pass_free) {
      int rc = upload_native_table(p, 2, p->n, bands);
      if (rc == QI_OK) {
        p->nat[2].nbands = B;
        rc = p->d.dtype == QI_F64 ? build_block_stx<double>(p, picks, coef, nullptr) : build_block_stx<float>(p, picks, coef, nullptr);
      }
      if (rc != QI_OK) {  // no half-built table: a ready table whose block bands have no producer would leave panel rows unwritten
        p->nat[2].release();
        p->blk[2].release();
        return rc;
      }
    } else if (p->d.engine == QI_ENGINE_NATIVE) {
      set_error("native engine: this Stockwell band table needs the two-pass kernels, which run 2^20 / 2^21 samples only");
      return QI_ERR_UNSUPPORTED;
    }  // else: the hipFFT engine runs it (nat[2] stays empty)
  } else if (p->d.engine == QI_ENGINE_NATIVE) {
    set_error("native engine does not support the Stockwell transform at n = %lld", (long long)p->n);
    return QI_ERR_UNSUPPORTED;
  }
  p->nb_stx = B;
  return QI_OK;
}

int64_t qi_plan_bands(const qi_plan* p, int which) {
  if (!p) return 0;
  if (which == QI_BANK_STYX || which == QI_BANK_ATOMS) return p->nb[which];
  return which == 2 ? p->nb_stx : 0;
}

int64_t qi_plan_stage_bands(const qi_plan* p, int which, int stage) {
  if (!p || which < 0 || which > 2) return 0;
  const int64_t total = qi_plan_bands(p, which);
  if (!p->nat[which].ready) return stage == QI_STAGE_INVERSE ? total : 0;
  int64_t blk = 0;
  if (which != 1 && p->blk[which].ready) blk = p->blk[which].rows;
  const int64_t zoom = p->nat[which].nzoom + p->nat[which].nz64;
  if (stage == QI_STAGE_BLOCK) return blk;
  if (stage == QI_STAGE_ZOOM) return zoom;
  return stage == QI_STAGE_PASS2 ? total - blk - zoom : 0;
}

int qi_plan_profile(qi_plan* p, int enable) {
  QI_REQUIRE(p, "null plan");
  DeviceGuard g(p->d.device);
  double ms[Profiler::kStages];
  int64_t c[Profiler::kStages];
  p->prof.read(ms, c);
  p->prof.on = enable != 0;
  p->prof.mask = (enable & 0xFFFF) == 1 ? ~0u : (uint32_t)(enable & 0xFFFF) >> 1;
  p->prof.period = (enable >> 16) > 0 ? (enable >> 16) : 1;
  p->prof.tick = 0;
  return QI_OK;
}

int qi_plan_profile_read(qi_plan* p, double* stage_ms, int64_t* stage_launches, int32_t n_stages) {
  QI_REQUIRE(p && stage_ms && stage_launches, "null argument");
  QI_REQUIRE(n_stages == QI_STAGE_COUNT, "n_stages must be %d", (int)QI_STAGE_COUNT);
  DeviceGuard g(p->d.device);
  p->prof.read(stage_ms, stage_launches);
  return QI_OK;
}

int qi_cwt(qi_plan* p, int bank, const void* sig, int64_t C, const qi_tfr_out* out, qi_stream stream) {
  QI_REQUIRE(p && sig && out, "null argument");
  QI_REQUIRE(bank == QI_BANK_STYX || bank == QI_BANK_ATOMS, "bad bank %d", bank);
  QI_REQUIRE(C > 0, "n_channels must be positive");
  DeviceGuard g(p->d.device);
  const Kind k = bank == QI_BANK_STYX ? Kind::Linear : Kind::Circular;
  p->prof.unchain();
  if (p->nat[bank].ready)
    return p->d.dtype == QI_F64 ? run_native64(p, bank, sig, C, out, (hipStream_t)stream)
                                : run_native<float>(p, bank, sig, C, out, (hipStream_t)stream);
  return p->d.dtype == QI_F64 ? run_transform<double>(p, k, sig, C, out, (hipStream_t)stream)
                              : run_transform<float>(p, k, sig, C, out, (hipStream_t)stream);
}

int qi_stx(qi_plan* p, const void* sig, int64_t C, const qi_tfr_out* out, qi_stream stream) {
  QI_REQUIRE(p && sig && out, "null argument");
  QI_REQUIRE(C > 0, "n_channels must be positive");
  DeviceGuard g(p->d.device);
  p->prof.unchain();
  if (p->nat[2].ready)
    return p->d.dtype == QI_F64 ? run_native64(p, 2, sig, C, out, (hipStream_t)stream)
                                : run_native<float>(p, 2, sig, C, out, (hipStream_t)stream);
  return p->d.dtype == QI_F64 ? run_transform<double>(p, Kind::Stockwell, sig, C, out, (hipStream_t)stream)
                              : run_transform<float>(p, Kind::Stockwell, sig, C, out, (hipStream_t)stream);
}

// the outputs of records [c0, ...) of a call whose panels have B bands
static qi_tfr_out shift_out(const qi_tfr_out& o, int64_t c0, int64_t B, int64_t n, int dtype) {
  const size_t r = dtype == QI_F64 ? 8 : 4;
  qi_tfr_out s = o;
  if (o.coef) s.coef = static_cast<char*>(o.coef) + (size_t)c0 * B * n * 2 * r;
  if (o.bits) s.bits = static_cast<char*>(o.bits) + (size_t)c0 * B * n * r;
  if (o.power_band) s.power_band = static_cast<char*>(o.power_band) + (size_t)c0 * B * 8;
  if (o.power_time) s.power_time = static_cast<char*>(o.power_time) + (size_t)c0 * n * r;
  if (o.stats) s.stats = static_cast<char*>(o.stats) + (size_t)c0 * 4 * 8;
  return s;
}

int qi_cwt_stx(qi_plan* p, int bank, const void* sig, int64_t C, const qi_tfr_out* out_cwt, const qi_tfr_out* out_stx,
               qi_stream stream) {
  QI_REQUIRE(p && sig && out_cwt && out_stx, "null argument");
  QI_REQUIRE(bank == QI_BANK_STYX, "qi_cwt_stx runs the styx bank (bank %d given)", bank);
  const bool fuse = p->native_fuse && p->d.dtype == QI_F32 && p->nat[bank].ready && p->nat[2].ready;
  p->carry.active = false;
  p->carry.has_zoom = false;
  QI_REQUIRE(C > 0, "n_channels must be positive");
  hipStream_t st = (hipStream_t)stream;
  if (fuse) {
    // Joint launches need the scratch of both transforms of a tile side by side: the records go through in tiles of
    // as many as fit (the scratch per record depends a little on the tile's size -- rows of the zoom launch --, so the
    // size is settled by iteration; it only ever shrinks).
    DeviceGuard g0(p->d.device);
    int64_t tile = C;
    if (p->native_tile > 0 && tile > p->native_tile) tile = p->native_tile;
    for (int it = 0; it < 8 && tile >= 1; ++it) {
      size_t pc0 = 0, pc2 = 0;
      QI_TRY(run_native<float>(p, bank, sig, tile, out_cwt, st, false, &p->carry, nullptr, &pc0));
      QI_TRY(run_native<float>(p, 2, sig, tile, out_stx, st, true, nullptr, &p->carry, &pc2));
      const int64_t fit = p->ws_bytes > (1u << 16) ? (int64_t)((p->ws_bytes - (1u << 16)) / (pc0 + pc2)) : 0;
      if (fit >= tile) break;
      tile = fit;
    }
    if (tile >= 1) {
      const int64_t n = p->n, B0 = p->nb[bank], B2 = p->nb_stx;
      const size_t r = p->d.dtype == QI_F64 ? 8 : 4;
      for (int64_t c0 = 0; c0 < C; c0 += tile) {
        const int64_t ct = C - c0 < tile ? C - c0 : tile;
        const void* s = static_cast<const char*>(sig) + (size_t)c0 * n * r;
        const qi_tfr_out oc = shift_out(*out_cwt, c0, B0, n, p->d.dtype), os = shift_out(*out_stx, c0, B2, n, p->d.dtype);
        p->carry.active = false;
        p->carry.has_zoom = false;
        p->prof.unchain();
        QI_TRY(run_native<float>(p, bank, s, ct, &oc, st, false, &p->carry, nullptr));
        p->prof.unchain();
        int rc = run_native<float>(p, 2, s, ct, &os, st, /*may_share=*/true, nullptr, &p->carry);
        if (p->carry.active) {  // the Stockwell run failed before it reached the deferred launches
          const int rc2 = flush_carry(p, &p->carry, st);
          if (rc == QI_OK) rc = rc2;
        }
        p->shared_valid = false;
        QI_TRY(rc);
      }
      return QI_OK;
    }
  }
  QI_TRY(qi_cwt(p, bank, sig, C, out_cwt, stream));
  if (p->d.dtype == QI_F32 && p->nat[bank].ready && p->nat[2].ready) {  // separate launches, but the Stockwell run may still use the CWT's spectra
    DeviceGuard g(p->d.device);
    p->prof.unchain();
    const int rc = run_native<float>(p, 2, sig, C, out_stx, st, /*may_share=*/true, nullptr, nullptr);
    p->shared_valid = false;
    return rc;
  }
  return qi_stx(p, sig, C, out_stx, stream);
}

// ---- STFT ----------------------------------------------------------------------------------------
int64_t qi_stft_segments(int64_t n, int64_t seg, int64_t hop) {
  if (n <= 0 || seg <= 0 || hop <= 0 || hop > seg) return 0;
  const int64_t len0 = n + 2 * (seg / 2);  // boundary='zeros' extends by seg//2 on both sides
  const int64_t nadd = ((hop - ((len0 - seg) % hop)) % hop) % seg;  // padded=True
  return (len0 + nadd - seg) / hop + 1;
}

int64_t qi_stft_scratch_bytes(int dtype, int64_t C, int64_t n, int64_t seg, int64_t hop, int64_t nfft) {
  const int64_t nseg = qi_stft_segments(n, seg, hop);
  const size_t e = dtype == QI_F64 ? 8 : 4;
  return (int64_t)(align_up((size_t)C * nseg * nfft * e) + align_up((size_t)C * nseg * (nfft / 2 + 1) * 2 * e));
}

int qi_stft(int dtype, int device, const void* sig, int64_t C, int64_t n, const void* window, int64_t seg,
            int64_t hop, int64_t nfft, double scale, void* Z, void* bits, double eps, void* scratch,
            int64_t scratch_bytes, qi_stream stream) {
  QI_REQUIRE(sig && window && Z && scratch, "null argument");
  QI_REQUIRE(dtype == QI_F32 || dtype == QI_F64, "bad dtype %d", dtype);
  QI_REQUIRE(C > 0 && n > 0 && seg > 0 && hop > 0 && hop <= seg && nfft >= seg, "bad STFT geometry");
  QI_REQUIRE(scratch_bytes >= qi_stft_scratch_bytes(dtype, C, n, seg, hop, nfft), "scratch too small");
  DeviceGuard g(device);
  return dtype == QI_F64 ? stft_impl<double>(device, sig, C, n, window, seg, hop, nfft, scale, Z, bits, eps,
                                             (char*)scratch, (hipStream_t)stream)
                         : stft_impl<float>(device, sig, C, n, window, seg, hop, nfft, scale, Z, bits, eps,
                                            (char*)scratch, (hipStream_t)stream);
}

int64_t qi_welch_scratch_bytes(int dtype, int64_t C, int64_t n, int64_t seg, int64_t hop, int64_t nfft) {
  if (n < seg || seg <= 0 || hop <= 0) return 0;
  const int64_t nseg = (n - seg) / hop + 1;
  const size_t e = dtype == QI_F64 ? 8 : 4;
  return (int64_t)(align_up((size_t)C * nseg * nfft * e) + align_up((size_t)C * nseg * (nfft / 2 + 1) * 2 * e));
}

int qi_welch(int dtype, int device, const void* sig, int64_t C, int64_t n, const void* window, int64_t seg,
             int64_t hop, int64_t nfft, double scale, void* pxx, void* scratch, int64_t scratch_bytes,
             qi_stream stream) {
  QI_REQUIRE(sig && window && pxx && scratch, "null argument");
  QI_REQUIRE(dtype == QI_F32 || dtype == QI_F64, "bad dtype %d", dtype);
  QI_REQUIRE(C > 0 && seg > 0 && n >= seg && hop > 0 && hop <= seg && nfft >= seg, "bad Welch geometry");
  QI_REQUIRE(scratch_bytes >= qi_welch_scratch_bytes(dtype, C, n, seg, hop, nfft), "scratch too small");
  DeviceGuard g(device);
  return dtype == QI_F64 ? welch_impl<double>(device, sig, C, n, window, seg, hop, nfft, scale, pxx, (char*)scratch,
                                              (hipStream_t)stream)
                         : welch_impl<float>(device, sig, C, n, window, seg, hop, nfft, scale, pxx, (char*)scratch,
                                             (hipStream_t)stream);
}

// ---- tfr_info -------------------------------------------------------------------------------------
int64_t qi_power_marginals_scratch_bytes(int64_t C, int64_t B, int64_t n) {
  const int64_t nblk = ceil_div(n, kEpiSpan);
  return (int64_t)(align_up((size_t)C * B * nblk * 8) + align_up((size_t)C * nblk * 24));
}

int qi_power_marginals(int dtype, int device, const void* power, int64_t C, int64_t B, int64_t n, void* power_band,
                       void* power_time, void* stats, void* scratch, int64_t scratch_bytes, qi_stream stream) {
  QI_REQUIRE(power && scratch, "null argument");
  QI_REQUIRE(dtype == QI_F32 || dtype == QI_F64, "bad dtype %d", dtype);
  QI_REQUIRE(C > 0 && B > 0 && n > 0, "bad panel shape");
  QI_REQUIRE(scratch_bytes >= qi_power_marginals_scratch_bytes(C, B, n), "scratch too small");
  DeviceGuard g(device);
  hipStream_t st = (hipStream_t)stream;
  const int64_t nblk = ceil_div(n, kEpiSpan);
  double* pb = reinterpret_cast<double*>(scratch);
  double* ps = reinterpret_cast<double*>((char*)scratch + align_up((size_t)C * B * nblk * 8));
  if (dtype == QI_F64)
    QI_TRY(launch_power_marginals<double>((const double*)power, C, B, n, (double*)power_time, pb, ps, st));
  else
    QI_TRY(launch_power_marginals<float>((const float*)power, C, B, n, (float*)power_time, pb, ps, st));
  return launch_finalize(power_band ? pb : nullptr, stats ? ps : nullptr, (double*)power_band, (double*)stats, C, B,
                         nblk, nblk, st);
}

int qi_log2_offset(int dtype, int device, const void* in, void* out, int64_t C, int64_t count, double eps,
                   const void* ref, qi_stream stream) {
  QI_REQUIRE(in && out, "null argument");
  QI_REQUIRE(dtype == QI_F32 || dtype == QI_F64, "bad dtype %d", dtype);
  QI_REQUIRE(C > 0 && count > 0, "bad shape");
  DeviceGuard g(device);
  return dtype == QI_F64 ? launch_log2_offset<double>((const double*)in, (double*)out, C, count, eps,
                                                      (const double*)ref, (hipStream_t)stream)
                         : launch_log2_offset<float>((const float*)in, (float*)out, C, count, (float)eps,
                                                     (const double*)ref, (hipStream_t)stream);
}

int qi_widen(int device, const void* in, void* out, int64_t count, qi_stream stream) {
  QI_REQUIRE(in && out && count > 0, "bad argument");
  DeviceGuard g(device);
  return launch_widen((const float*)in, (double*)out, count, (hipStream_t)stream);
}

int qi_log2_abs(int dtype, int device, const void* in, int is_complex, void* out, int64_t count, double eps,
                qi_stream stream) {
  QI_REQUIRE(in && out, "null argument");
  QI_REQUIRE(dtype == QI_F32 || dtype == QI_F64, "bad dtype %d", dtype);
  QI_REQUIRE(count > 0, "bad shape");
  DeviceGuard g(device);
  return dtype == QI_F64 ? launch_log2_abs<double>((const double*)in, is_complex, (double*)out, count, eps, (hipStream_t)stream)
                         : launch_log2_abs<float>((const float*)in, is_complex, (float*)out, count, (float)eps, (hipStream_t)stream);
}

int qi_shannon_panel(int dtype, int device, const void* power, const void* mult, int mode, int64_t C, int64_t B,
                     int64_t n, double deg_free, void* info, void* shannon_bits, void* isnr, void* esnr,
                     qi_stream stream) {
  QI_REQUIRE(power && mult, "null argument");
  QI_REQUIRE(dtype == QI_F32 || dtype == QI_F64, "bad dtype %d", dtype);
  QI_REQUIRE(mode >= 0 && mode <= 2, "bad mode %d", mode);
  QI_REQUIRE(C > 0 && B > 0 && n > 0 && deg_free > 1.0, "bad shape");
  DeviceGuard g(device);
  return dtype == QI_F64
             ? launch_shannon<double>((const double*)power, (const double*)mult, mode, C, B, n, deg_free,
                                      (double*)info, (double*)shannon_bits, (double*)isnr, (double*)esnr,
                                      (hipStream_t)stream)
             : launch_shannon<float>((const float*)power, (const float*)mult, mode, C, B, n, deg_free, (float*)info,
                                     (float*)shannon_bits, (float*)isnr, (float*)esnr, (hipStream_t)stream);
}

// ---- 1-D Shannon family ---------------------------------------------------------------------------------------------
int qi_shannon_1d(int dtype, int device, const void* marginal, int64_t C, int64_t n, void* info, void* entropy,
                  void* isnr, void* esnr, qi_stream stream) {
  QI_REQUIRE(marginal, "null argument");
  QI_REQUIRE(dtype == QI_F32 || dtype == QI_F64, "bad dtype %d", dtype);
  QI_REQUIRE(C > 0 && n > 1, "bad shape");
  DeviceGuard g(device);
  return dtype == QI_F64 ? launch_shannon_1d<double>((const double*)marginal, C, n, (double*)info, (double*)entropy,
                                                     (double*)isnr, (double*)esnr, (hipStream_t)stream)
                         : launch_shannon_1d<float>((const float*)marginal, C, n, (float*)info, (float*)entropy,
                                                    (float*)isnr, (float*)esnr, (hipStream_t)stream);
}

int64_t qi_shannon_scratch_bytes(int dtype, int64_t C, int64_t n) {
  if (C <= 0 || n <= 1) return 0;
  const int64_t nf = n / 2 + 1, esz = dtype == QI_F64 ? 8 : 4;
  // partial sums | unwrap turns | a copy of the records (the real-to-complex transform may overwrite its input)
  return (int64_t)(align_up((size_t)C * shannon_spans(n) * 8) + align_up((size_t)C * nf * 4) + align_up((size_t)C * n * esz));
}

int qi_shannon_tdr(int dtype, int device, const void* sig, int64_t C, int64_t n, void* sig_norm, void* marginal,
                   void* scratch, int64_t scratch_bytes, qi_stream stream) {
  QI_REQUIRE(sig && marginal && scratch, "null argument");
  QI_REQUIRE(dtype == QI_F32 || dtype == QI_F64, "bad dtype %d", dtype);
  QI_REQUIRE(C > 0 && n > 1, "bad shape");
  QI_REQUIRE(scratch_bytes >= qi_shannon_scratch_bytes(dtype, C, n), "scratch too small");
  DeviceGuard g(device);
  double* partial = static_cast<double*>(scratch);
  return dtype == QI_F64 ? launch_tdr_marginal<double>((const double*)sig, C, n, (double*)sig_norm, (double*)marginal,
                                                       partial, (hipStream_t)stream)
                         : launch_tdr_marginal<float>((const float*)sig, C, n, (float*)sig_norm, (float*)marginal,
                                                      partial, (hipStream_t)stream);
}

int qi_shannon_fft(int dtype, int device, const void* sig, int64_t C, int64_t n, void* spectrum, void* angle,
                   void* marginal, void* scratch, int64_t scratch_bytes, qi_stream stream) {
  QI_REQUIRE(sig && spectrum && marginal && scratch, "null argument");
  QI_REQUIRE(dtype == QI_F32 || dtype == QI_F64, "bad dtype %d", dtype);
  QI_REQUIRE(C > 0 && n > 1, "bad shape");
  QI_REQUIRE(scratch_bytes >= qi_shannon_scratch_bytes(dtype, C, n), "scratch too small");
  DeviceGuard g(device);
  return dtype == QI_F64 ? shannon_fft_impl<double>(device, (const double*)sig, C, n, (double2*)spectrum, (double*)angle,
                                                    (double*)marginal, (char*)scratch, (hipStream_t)stream)
                         : shannon_fft_impl<float>(device, (const float*)sig, C, n, (float2*)spectrum, (float*)angle,
                                                   (float*)marginal, (char*)scratch, (hipStream_t)stream);
}

// ---- sliding-window STFT in scipy.signal.ShortTimeFFT's convention ---------------------------------------------------
int64_t qi_sliding_scratch_bytes(int dtype, int64_t C, int64_t nfft, int64_t n_slices) {
  if (C <= 0 || nfft <= 0 || n_slices <= 0) return 0;
  const size_t esz = dtype == QI_F64 ? 8 : 4;
  return (int64_t)(align_up((size_t)C * n_slices * nfft * esz) + align_up((size_t)C * n_slices * (nfft / 2 + 1) * 2 * esz));
}

int qi_sliding_stft(int dtype, int device, const void* sig, int64_t C, int64_t n, const void* window, int64_t seg,
                    int64_t hop, int64_t nfft, int64_t first, int64_t n_slices, int pad_mode, int detrend, int64_t roll,
                    void* Z, void* real_out, int real_kind, void* scratch, int64_t scratch_bytes, qi_stream stream) {
  QI_REQUIRE(sig && window && scratch && (Z || real_out), "null argument");
  QI_REQUIRE(dtype == QI_F32 || dtype == QI_F64, "bad dtype %d", dtype);
  QI_REQUIRE(C > 0 && n > 0 && seg > 0 && hop > 0 && nfft >= seg && n_slices > 0 && roll >= 0 && roll < nfft, "bad shape");
  QI_REQUIRE(pad_mode >= 0 && pad_mode <= 3 && (real_kind == 1 || real_kind == 2 || !real_out), "bad mode");
  QI_REQUIRE(pad_mode < 2 || (-first <= n - 1 && first + (n_slices - 1) * hop + seg - n <= n - 1),
             "reflective padding reaches further than the record is long");
  QI_REQUIRE(scratch_bytes >= qi_sliding_scratch_bytes(dtype, C, nfft, n_slices), "scratch too small");
  DeviceGuard g(device);
  return dtype == QI_F64
             ? sliding_stft_impl<double>(device, (const double*)sig, C, n, (const double*)window, seg, hop, nfft, first,
                                         n_slices, pad_mode, detrend, roll, (double2*)Z, (double*)real_out, real_kind,
                                         (char*)scratch, (hipStream_t)stream)
             : sliding_stft_impl<float>(device, (const float*)sig, C, n, (const float*)window, seg, hop, nfft, first,
                                        n_slices, pad_mode, detrend, roll, (float2*)Z, (float*)real_out, real_kind,
                                        (char*)scratch, (hipStream_t)stream);
}

int qi_sliding_istft(int dtype, int device, const void* S, int64_t C, const void* dual_window, int64_t seg, int64_t hop,
                     int64_t nfft, int64_t first, int64_t n_slices, int64_t roll, int64_t k0, int64_t k1, void* out,
                     void* scratch, int64_t scratch_bytes, qi_stream stream) {
  QI_REQUIRE(S && dual_window && out && scratch, "null argument");
  QI_REQUIRE(dtype == QI_F32 || dtype == QI_F64, "bad dtype %d", dtype);
  QI_REQUIRE(C > 0 && seg > 0 && hop > 0 && nfft >= seg && n_slices > 0 && k1 > k0 && roll >= 0 && roll < nfft, "bad shape");
  QI_REQUIRE(scratch_bytes >= qi_sliding_scratch_bytes(dtype, C, nfft, n_slices), "scratch too small");
  DeviceGuard g(device);
  return dtype == QI_F64
             ? sliding_istft_impl<double>(device, (const double2*)S, C, (const double*)dual_window, seg, hop, nfft, first,
                                          n_slices, roll, k0, k1, (double*)out, (char*)scratch, (hipStream_t)stream)
             : sliding_istft_impl<float>(device, (const float2*)S, C, (const float*)dual_window, seg, hop, nfft, first,
                                         n_slices, roll, k0, k1, (float*)out, (char*)scratch, (hipStream_t)stream);
}

}  // extern "C"
