// The tables of a plan: support analysis of the atom spectra, assignment of every band to an engine (zoom classes, block
// reach groups, split bands, two-pass groups), the device tables and work-item lists of those engines.
#include <algorithm>

#include "qi_host.hpp"

using namespace qi;

namespace qi {
namespace host {

bool native_len_ok(int64_t Lf) { return Lf == (1ll << 20) || Lf == (1ll << 21); }

// does this plan run transform `kind` (0 styx bank, 1 atoms bank, 2 Stockwell) on the native engine?
bool native_wanted(const qi_plan* p, int kind) {
  if (p->d.engine == QI_ENGINE_HIPFFT) return false;
  const int64_t Lf = kind == 0 ? p->L : p->n;
  if (p->d.dtype == QI_F64 && !p->native_f64) return false;
  if (is_pow2(p->n) && native_len_ok(Lf)) return true;
  // Stockwell and styx tables usually have no band for the two-pass kernels (every band is a zoom, block or split
  // band), and those engines -- in float32 and, since round 4, their float64 twins (float64 zoom, k_block64, split bands) --
  // take any power-of-two length from 2^15: the table build decides
  return kind != 1 && is_pow2(p->n) && p->n >= ((int64_t)1 << p->native_min_log2n) && Lf <= (1ll << 26);
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
    // The three coarsest grids go through the fine kernel with wave-uniform windows (k_z64_fine), by CLASS = (grid,
    // interpolator length): on the coarsest grid, where every narrower band lands, a band oversampled >= 8 / 16 / 32 / 64
    // times takes 12 / 10 / 8 / 6 taps instead of 16 (classes 3..6; the same error bound, see z64f_ntap).  A class of fewer
    // than four bands joins the next longer interpolator.
    for (int c = 0; c < native::kZ64FineClasses; ++c) t.zf_first[c] = t.zf_count[c] = 0;
    if (p->native_z64_fine) {
      const int64_t M0 = Lf / 64;
      std::vector<std::vector<native::BandDesc>> cls(native::kZ64FineClasses);
      for (const auto& d : lvl[0]) {
        int c = 0;
        for (int q = native::kZ64FineClasses - 1; q >= native::kZ64FineLevels && c == 0; --q)
          if ((int64_t)native::z64f_oversampling(q) * d.k_len <= M0) c = q;
        cls[c].push_back(d);
      }
      for (int q = native::kZ64FineClasses - 1; q >= native::kZ64FineLevels; --q) {
        if (cls[q].empty() || cls[q].size() >= 4) continue;
        const int to = q == native::kZ64FineLevels ? 0 : q - 1;
        cls[to].insert(cls[to].end(), cls[q].begin(), cls[q].end());
        cls[q].clear();
      }
      lvl[0].clear();
      int32_t pos = 0;
      std::vector<int> order0{0};  // list order of the coarsest grid: the 16-tap class, then the shorter interpolators
      for (int c = native::kZ64FineLevels; c < native::kZ64FineClasses; ++c) order0.push_back(c);
      for (int c : order0) {
        t.zf_first[c] = pos;
        t.zf_count[c] = (int32_t)cls[c].size();
        pos += t.zf_count[c];
        lvl[0].insert(lvl[0].end(), cls[c].begin(), cls[c].end());
      }
      for (int g = 1; g < native::kZ64FineLevels && g < native::kZ64Levels; ++g) {
        t.zf_first[g] = pos;
        t.zf_count[g] = (int32_t)lvl[g].size();
        pos += t.zf_count[g];
      }
      for (int c = 0; c < native::kZ64FineClasses; ++c) {
        if (t.zf_count[c] == 0 || p->d_z64f_w[c]) continue;
        std::vector<double> w((size_t)native::z64f_win(c) * 64);
        native::z64_fine_weights(c, w.data());
        QI_HIP(hipMalloc((void**)&p->d_z64f_w[c], w.size() * sizeof(double)));
        QI_HIP(hipMemcpy(p->d_z64f_w[c], w.data(), w.size() * sizeof(double), hipMemcpyHostToDevice));
      }
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
    if (!z.empty() && p->native_z64_fine && kind != 2) {
      // carrier factors of the fine kernel (Gabor kinds), exact integer phases evaluated in long double: per band
      // exp(2 pi i k_c (lane - e) / Lf) for the 64 lanes and the step exp(2 pi i k_c 64 / Lf); per table the waves' factors
      // exp(2 pi i j kZ64FineWave / Lf).  e = 1 for the zero-padded kind (the carrier of full-length sample tau - 1).
      const long double two_pi = 6.283185307179586476925286766559005768L;
      const int e = kind == 0 ? 1 : 0;
      auto root = [&](int64_t m) {
        m = ((m % Lf) + Lf) % Lf;
        const long double ang = two_pi * (long double)m / (long double)Lf;
        return make_double2((double)cosl(ang), (double)sinl(ang));
      };
      std::vector<double2> lane(z.size() * 65);
      for (size_t j = 0; j < z.size(); ++j) {
        const int64_t kc = (int64_t)z[j].k_lo + z[j].k_len / 2;
        for (int l = 0; l < 64; ++l) lane[j * 65 + l] = root(kc * (l - e));
        lane[j * 65 + 64] = root(kc * 64);
      }
      const int64_t nw = Lf / native::kZ64FineWave;
      std::vector<double2> wave((size_t)nw);
      for (int64_t j = 0; j < nw; ++j) wave[(size_t)j] = root(j * native::kZ64FineWave);
      QI_HIP(hipMalloc((void**)&t.d_z64_lane_ph, lane.size() * sizeof(double2)));
      QI_HIP(hipMemcpy(t.d_z64_lane_ph, lane.data(), lane.size() * sizeof(double2), hipMemcpyHostToDevice));
      QI_HIP(hipMalloc((void**)&t.d_z64_wave_ph, wave.size() * sizeof(double2)));
      QI_HIP(hipMemcpy(t.d_z64_wave_ph, wave.data(), wave.size() * sizeof(double2), hipMemcpyHostToDevice));
    }
    if (tune_env("QI_NATIVE_VERBOSE")) {
      fprintf(stderr, "[qi plan] table %d: float64 zoom bands per level %d %d %d %d %d, two-pass bands %zu; fine classes (taps: bands)", kind,
              t.z64_count[0], t.z64_count[1], t.z64_count[2], t.z64_count[3], t.z64_count[4], rest.size());
      for (int c = 0; c < native::kZ64FineClasses; ++c)
        fprintf(stderr, " %d@L%d: %d", native::z64f_ntap(c), native::z64f_level(c), t.zf_count[c]);
      fprintf(stderr, "\n");
    }
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
#ifdef QI_HOST_SANITIZE
  // Host sanitizer build (tests/sanitize): no kernel runs there, so the supports are the Gaussian atoms' own -- centre
  // omega L / 2 pi, half-width where exp(-d^2 / 4 p_re) falls below the threshold -- with the whole row for an atom the
  // record cuts off, and a taper's widening (~ 6.6 L / e bins at 2^-30, 13.2 at 2^-50, as the GPU analysis measures it).
  // "Device" memory is host memory in that build: d_par is read directly.
  (void)st;
  (void)circular;
  const double bits = p->d.dtype == QI_F64 ? 50.0 : 30.0;
  sup->assign((size_t)count * 3, 0.0);
  for (int32_t q = 0; q < count; ++q) {
    const double p_re = d_par[j0 + q], om = d_par[2 * (int64_t)B + j0 + q];
    const double kc = om * (double)L / (2.0 * M_PI), hw = std::sqrt(4.0 * p_re * bits * M_LN2) * (double)L / (2.0 * M_PI);
    const bool cut = p_re * 0.25 * (double)p->n * (double)p->n <= bits * M_LN2;
    double lo = 0.0, hi = (double)(L - 1);
    if (!cut || taper_e > 0.0) {
      const double extra = cut ? (p->d.dtype == QI_F64 ? 13.2 : 6.6) * (double)L / taper_e : 0.0;
      lo = std::floor(kc - hw - extra);
      hi = std::ceil(kc + hw + extra);
      if (hi - lo + 1.0 >= (double)L) {
        lo = 0.0;
        hi = (double)(L - 1);
      }
    }
    (*sup)[3 * (size_t)q] = 1.0;
    (*sup)[3 * (size_t)q + 1] = lo;
    (*sup)[3 * (size_t)q + 2] = hi;
  }
  return QI_OK;
#endif
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
        // (float64: the same for the bands the float64 zoom takes -- upload_native_table's rule)
        const bool z64_band = p->d.dtype == QI_F64 && z64_table(p, 0) && d.mode == 0 &&
                              4 * (int64_t)d.k_len <= ((L / 64) << (p->native_z64_levels - 1));
        const double ramp = ((d.mode >= 2 || z64_band) && !circular) ? -1.0 / (double)L : 0.0;
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
  constexpr int NG = 4;
  const int wqs[NG] = {1, 2, 4, native::kBlkLongWq};
  // (float64 tables: Gaussian weights in double from every bin -- the shortcuts below drop weights under 2^-30 of the peak)
  constexpr bool F64 = sizeof(T) == 8;
  const double drop_bits = F64 ? 52.0 : 30.0;
  // first bin of the 256-bin window of a long band: centred on the band, kept inside the lower half of the 8192-bin grid
  // (the half a long block holds)
  auto long_window = [&](const BlockPick& pk) {
    return std::min<int64_t>(std::max<int64_t>((int64_t)std::llround(2.0 * pk.kappa) - 128, 0), native::kBlk - 256);
  };
  auto long_ok = [&](const BlockPick& pk, int cut) {
    if (F64) return false;
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
        const int home = is_long ? native::kBlkLongWq : picks[r].wq;  // the group that takes this band
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
        // (float64 since round 5: `half` is then the 2^-52 half-width, the weight comes from the table -- bands of the 512- and
        // 1024-sample reach groups; analytic = 2, an aliased spectrum, is not narrow)
        if ((!F64 || (b.analytic == 1 && p->native_blk64_wtab && p->native_blk64_narrow)) && b.analytic && p->native_blk_narrow &&
            2.0 * half + 2.0 <= 256.0) {
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
      // the group's bands are dealt to `nchunk` workgroups per block (each pays one forward transform of the block)
      const int per_wg = v == 0 ? p->native_blk_bands : p->native_blk_bands_batch;
      const int32_t nchunk = (int32_t)ceil_div(count, per_wg);
      const int64_t nblocks = ceil_div(p->n, native::block_valid(wqs[g]));
      if (tune_env("QI_NATIVE_VERBOSE"))
        fprintf(stderr, "[qi plan] block table %d cut %d, reach <= %d%s: %d bands (%d analytic, %d narrow, %d half) in %d workgroups x %lld blocks\n", kind, v,
                g == 3 ? 1024 : 256 * (wqs[g] & 15), g == 3 ? " (8192-sample blocks)" : "", count,
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
      // -- in the table for many records one item per block covers all of them (one plane, one launch of its own)
      const int wq = (int)(p->native_split_e / 512);
      il.edge_merged = (v == 1 && p->native_edge_merge != 0) || F64;  // (float64: always, k_block64_edge)
      if (il.edge_merged) {
        for (int64_t b = 0; b < split_blocks; ++b)
          items.push_back({-wq, (int32_t)b, 0, p->nsplit, il.nplanes, (int32_t)items.size()});
        il.nplanes += 1;
      } else {
        for (int32_t sb = 0; sb < p->nsplit; ++sb) {
          for (int64_t b = 0; b < split_blocks; ++b)
            items.push_back({-wq, (int32_t)b, sb, 0, il.nplanes, (int32_t)items.size()});
          il.nplanes += 1;
        }
      }
      il.nedge_items = (int32_t)items.size() - il.nitems;
    }
    il.h_items = items;
    if (F64 && p->native_blk64_wtab && !list.empty()) {
      // float64: the bands' real Gaussian filter weights, weight(k) = amp exp2(-(cw dk)^2) with dk = k - kappa wrapped to
      // +-kBlk / 2 (Gabor banks: the aliases of the half-sample grid alternate in sign) -- block_bands' formula, in double
      std::vector<double> gw(list.size() * (size_t)native::kBlk);
      for (size_t q = 0; q < list.size(); ++q) {
        const auto& b = list[q];
        for (int k = 0; k < native::kBlk; ++k) {
          if (b.analytic == 2) {  // an atom shorter than 2.75 samples: every alias that matters, alternating in sign
            double acc = 0.0;
            for (int m = -6; m <= 6; ++m) {
              const double e = (double)b.cw * ((double)(k - b.kappa_int) - (double)b.kappa_frac + (double)native::kBlk * m);
              acc += ((m & 1) && !demod ? -1.0 : 1.0) * std::exp2(-e * e);
            }
            gw[q * native::kBlk + k] = (double)b.amp * acc;
            continue;
          }
          double dk = (double)(k - b.kappa_int) - (double)b.kappa_frac, amp = (double)b.amp;
          if (dk > (double)(native::kBlk / 2)) {
            dk -= (double)native::kBlk;
            if (!demod) amp = -amp;
          }
          if (demod && dk < -(double)(native::kBlk / 2)) dk += (double)native::kBlk;
          const double e = (double)b.cw * dk;
          gw[q * native::kBlk + k] = amp * std::exp2(-e * e);
        }
      }
      QI_HIP(hipMalloc((void**)&il.d_gauss_w, gw.size() * sizeof(double)));
      QI_HIP(hipMemcpy(il.d_gauss_w, gw.data(), gw.size() * sizeof(double), hipMemcpyHostToDevice));
    }
    if (F64 && demod && !list.empty()) {
      // demodulation tables of the float64 Stockwell bands (exact integer phases, long double): per band exp(-2 pi i idx 256 i / n)
      const long double two_pi = 6.283185307179586476925286766559005768L;
      auto root = [&](int64_t m) {
        const long double ang = -two_pi * (long double)(((m % p->n) + p->n) % p->n) / (long double)p->n;
        return make_double2((double)cosl(ang), (double)sinl(ang));
      };
      std::vector<double2> pw(list.size() * 16);
      for (size_t q = 0; q < list.size(); ++q)
        for (int i = 0; i < 16; ++i) pw[q * 16 + i] = root((int64_t)(((__int128)list[q].shift * 256 * i) % p->n));
      QI_HIP(hipMalloc((void**)&il.d_demod_pow, pw.size() * sizeof(double2)));
      QI_HIP(hipMemcpy(il.d_demod_pow, pw.data(), pw.size() * sizeof(double2), hipMemcpyHostToDevice));
      if (!p->d_demod_t1 && p->n >= 1024) {
        std::vector<double2> t1((size_t)(p->n / 1024)), t2(1024);
        for (int64_t j = 0; j < p->n / 1024; ++j) t1[(size_t)j] = root(1024 * j);
        for (int64_t j = 0; j < 1024; ++j) t2[(size_t)j] = root(j);
        QI_HIP(hipMalloc((void**)&p->d_demod_t1, t1.size() * sizeof(double2)));
        QI_HIP(hipMemcpy(p->d_demod_t1, t1.data(), t1.size() * sizeof(double2), hipMemcpyHostToDevice));
        QI_HIP(hipMalloc((void**)&p->d_demod_t2, t2.size() * sizeof(double2)));
        QI_HIP(hipMemcpy(p->d_demod_t2, t2.data(), t2.size() * sizeof(double2), hipMemcpyHostToDevice));
      }
    }
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
    // (... its finest grid, Lf / 4 samples, oversamples it four times: at transform lengths below 2^19 the one-pass loader's
    // limit is wider than that, and such a band belongs to the block engine)
    // (... and not on one of its finest grids when the block engine can take the band: a band of 65 536 - 131 072 bins costs
    // the float64 zoom 13.5 us per record -- a 2^19-point coarse transform and the LDS-window interpolation kernel --
    // against 7.4 us on the block engine, measured at order 12 x 4 records: native_z64_block_from)
    bool z64_first = p->d.dtype == QI_F64 && z64_table(p, bank) && len > 0 && len <= narrow_limit(p, bank, L) &&
                     4 * len <= ((L / 64) << (p->native_z64_levels - 1));
    if (z64_first && can_block && block_group_of(w) > 0 && 4 * len > ((L / 64) << (p->native_z64_block_from - 1))) z64_first = false;
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
      // float64, an atom SHORTER than 2.75 samples (the top band of an order-1 or order-2 table): its sampled spectrum is the
      // Gaussian plus its aliases, sum over m of (-1)^m G(theta + 2 pi m) after the half-sample factor -- still real weights,
      // which the plan-time weight table holds summed (analytic = 2: table only; float32 reads such a band's bank row)
      if (p->d.dtype == QI_F64 && !pk.analytic && p->native_blk64_wtab && p_im == 0.0 && p_re > 0.0 && om > 0.0 && om < M_PI) {
        pk.analytic = 2;
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
    // float64: "the zoom engine" is the float64 zoom, which takes a band whose support its finest grid oversamples four times
    const bool z64 = p->d.dtype == QI_F64;
    auto z64_takes = [&](int64_t len) {
      return z64_table(p, bank) && len > 0 && len <= narrow_limit(p, bank, L) &&
             4 * len <= ((L / 64) << (p->native_z64_levels - 1));
    };
    for (int32_t j : keep) {
      const int64_t lo = (int64_t)sup[3 * j + 1], hi = (int64_t)sup[3 * j + 2];
      const int64_t len = hi >= lo ? hi - lo + 1 : 0;
      // (a band the one-pass loader of the two-pass kernels would take stays there only where those kernels exist)
      if (z64 ? z64_takes(len)
              : (zoom_class(p, bank, L, len) >= 0 || (len > 0 && len <= p->native_kmax && native_len_ok(L))))
        continue;
      std::vector<double> part;
      QI_TRY(analyse_support(p, 0, L, B, j, 1, d_par, &part, st, (double)se));
      const int64_t tlo = (int64_t)part[1], thi = (int64_t)part[2];
      const int64_t tlen = thi >= tlo ? thi - tlo + 1 : 0;
      if (tune_env("QI_NATIVE_VERBOSE"))
        fprintf(stderr, "[qi plan] band %d: support %lld bins as the reference cuts it, %lld bins tapered over %lld samples\n",
                j, (long long)len, (long long)tlen, (long long)se);
      if (z64 ? !z64_takes(tlen) : zoom_class(p, bank, L, tlen) < 0) continue;
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

// The Stockwell table of a plan: every band is assigned to the zoom / float64 zoom, block or two-pass engines from its
// window (shift index, sigma -> coef); nat[2] and blk[2] are left ready, or empty when the hipFFT engine runs the table.
int build_stx_tables(qi_plan* p, int32_t B, const int64_t* shift_index, const double* sigma, const std::vector<double>& coef) {
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
      bool z64_first = p->d.dtype == QI_F64 && z64_table(p, 2) && 2 * kh64 + 1 <= (double)narrow_limit(p, 2, p->n) &&
                       2 * kh64 + 1 < (double)p->n && 4.0 * (2 * kh64 + 1) <= (double)((p->n / 64) << (p->native_z64_levels - 1));
      // (a band of the float64 zoom's finest grids goes to the block engine when that can take it: native_z64_block_from)
      if (z64_first && can_block && sigma[j] >= 2.75 && block_group_of(reach) > 0 &&
          4.0 * (2 * kh64 + 1) > (double)((p->n / 64) << (p->native_z64_block_from - 1)))
        z64_first = false;
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
    p->stx_left_lo = -1;
    p->stx_left_n = 0;
    if (!native_len_ok(p->n) && p->d.engine != QI_ENGINE_NATIVE) {
      // bands for the two-pass kernels at a length they do not run: if they are the last (at most four) rows of the table, a
      // pass of the hipFFT engine over those rows follows the native run (run_stx_leftover)
      int32_t lo = B, cnt = 0;
      for (const auto& d : bands)
        if (d.mode == 1) {
          ++cnt;
          lo = d.out_band < lo ? d.out_band : lo;
        }
      // (the pass tiles the plan's scratch like the hipFFT engine: one record's spectrum, `cnt` rows and the partial sums
      // of the whole table must fit -- else the whole table goes to the hipFFT engine, which tiles over bands)
      const size_t row = (size_t)p->n * (p->d.dtype == QI_F64 ? sizeof(double2) : sizeof(float2));
      const int64_t nblk_e = ceil_div(p->n, kEpiSpan);
      const size_t part = align_up((size_t)B * nblk_e * 8) + align_up((size_t)B * nblk_e * 24);
      const bool left_fits = p->ws_bytes >= part + 2048 + row * (size_t)(cnt + 1);
      if (cnt > 0 && cnt <= 4 && lo == B - cnt && cnt < B && left_fits) {
        bands.erase(std::remove_if(bands.begin(), bands.end(), [](const native::BandDesc& d) { return d.mode == 1; }), bands.end());
        p->stx_left_lo = lo;
        p->stx_left_n = cnt;
      }
    }
    bool two_pass_free = true;  // no band for pass 1 / pass 2 (their transform lengths are 2^20 and 2^21 only)
    for (const auto& d : bands) two_pass_free = two_pass_free && d.mode >= 2;
    // (float64: the float64 zoom takes its bands inside upload_native_table -- what it leaves is known afterwards)
    const bool f64_maybe = p->d.dtype == QI_F64 && z64_table(p, 2);
    if (native_len_ok(p->n) || two_pass_free || f64_maybe) {
      int rc = upload_native_table(p, 2, p->n, bands);
      if (rc == QI_OK && !native_len_ok(p->n) && !p->nat[2].h_rows.empty()) {
        p->nat[2].release();  // bands are left for the two-pass kernels at a length they do not run: the hipFFT engine takes the table
        p->blk[2].release();
        p->stx_left_n = 0;
        if (p->d.engine == QI_ENGINE_NATIVE) {
          set_error("native engine: this Stockwell band table needs the two-pass kernels, which run 2^20 / 2^21 samples only");
          return QI_ERR_UNSUPPORTED;
        }
        return QI_OK;
      }
      if (rc == QI_OK) {
        p->nat[2].nbands = B;
        rc = p->d.dtype == QI_F64 ? build_block_stx<double>(p, picks, coef, nullptr) : build_block_stx<float>(p, picks, coef, nullptr);
      }
      if (rc != QI_OK) {  // no half-built table: a ready table whose block bands have no producer would leave panel rows unwritten
        p->nat[2].release();
        p->blk[2].release();
        p->stx_left_n = 0;
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
  return QI_OK;
}

template int build_native_bank<float>(qi_plan*, int, int32_t, const double*, const double*, hipStream_t);
template int build_native_bank<double>(qi_plan*, int, int32_t, const double*, const double*, hipStream_t);
template int build_bank<float>(qi_plan*, int, int32_t, const double*, hipStream_t);
template int build_bank<double>(qi_plan*, int, int32_t, const double*, hipStream_t);

}  // namespace host
}  // namespace qi
