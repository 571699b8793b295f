// Launch sequences of the engines: the hipFFT engine (any length), the native float32 engines (two-pass, zoom, block,
// joint launches of qi_cwt_stx) and the float64 native path.
#include "qi_host.hpp"

using namespace qi;

namespace qi {
namespace host {



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
  QI_LAYOUT_BEGIN(p, "hipFFT-engine tile", false);
  QI_LAYOUT_NOTE(p, "X", X, (size_t)tl.Ct * L * sizeof(cplx<T>));
  QI_LAYOUT_NOTE(p, "Y", Y, (size_t)tl.Ct * tl.Bt * L * sizeof(cplx<T>));
  QI_LAYOUT_NOTE(p, "part_band", part_band, (size_t)tl.Ct * B * tl.nblk * 8);
  QI_LAYOUT_NOTE(p, "part_stat", part_stat, (size_t)tl.Ct * B * tl.nblk * 24);
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

// Stockwell rows [stx_left_lo, nb_stx) behind a native run that left them out (qi_plan::stx_left_*): the hipFFT engine's
// window -> inverse transform -> epilogue over just these rows; the per-time sums are added to the native run's, the rows'
// powers and the panel statistics merged by k_left_merge (fixed order).
namespace {
__global__ void __launch_bounds__(256) k_left_merge(const double* __restrict__ part_band, const double* __restrict__ part_stat,
                                                    double* __restrict__ power_band, double* __restrict__ stats, int64_t B,
                                                    int64_t j0, int64_t bt, int64_t nblk) {
  const int64_t c = blockIdx.x;
  if (threadIdx.x < bt && power_band) {
    const double* q = part_band + (c * B + j0 + threadIdx.x) * nblk;
    double s = 0.0;
    for (int64_t b = 0; b < nblk; ++b) s += q[b];
    power_band[c * B + j0 + threadIdx.x] = s;
  }
  if (threadIdx.x == 64 && stats) {
    const double* q = part_stat + ((c * 2 + 1) * nblk) * 3;  // (the epilogue ran as tile 1 of 2)
    double m = stats[c * 4 + 0], s1 = 0.0, s2 = 0.0;
    for (int64_t b = 0; b < nblk; ++b) {
      m = q[3 * b] > m ? q[3 * b] : m;
      s1 += q[3 * b + 1];
      s2 += q[3 * b + 2];
    }
    stats[c * 4 + 0] = m;
    stats[c * 4 + 1] += s1;
    stats[c * 4 + 2] += s2;
  }
}
}  // namespace

template <typename T>
int run_stx_leftover(qi_plan* p, const void* sig_v, int64_t C, const qi_tfr_out* out, hipStream_t st) {
  const int64_t n = p->n, B = p->nb_stx, j0 = p->stx_left_lo, bt = p->stx_left_n;
  const T* sig = static_cast<const T*>(sig_v);
  Tile tl;
  QI_TRY(plan_tiles<T>(p, C, B, n, &tl));
  if (tl.Bt < bt) {
    set_error("workspace too small for the %lld Stockwell rows behind the native run", (long long)bt);
    return QI_ERR_NOMEM;
  }
  cplx<T>* X = reinterpret_cast<cplx<T>*>(p->ws + tl.off_x);
  cplx<T>* Y = reinterpret_cast<cplx<T>*>(p->ws + tl.off_y);
  double* part_band = reinterpret_cast<double*>(p->ws + tl.off_pb);
  double* part_stat = reinterpret_cast<double*>(p->ws + tl.off_ps);
  QI_LAYOUT_BEGIN(p, "hipFFT-engine tile", false);
  QI_LAYOUT_NOTE(p, "X", X, (size_t)tl.Ct * n * sizeof(cplx<T>));
  QI_LAYOUT_NOTE(p, "Y", Y, (size_t)tl.Ct * tl.Bt * n * sizeof(cplx<T>));
  QI_LAYOUT_NOTE(p, "part_band", part_band, (size_t)tl.Ct * B * tl.nblk * 8);
  QI_LAYOUT_NOTE(p, "part_stat", part_stat, (size_t)tl.Ct * B * tl.nblk * 24);
  const bool want_band = out->power_band != nullptr, want_stat = out->stats != nullptr;
  for (int64_t c0 = 0; c0 < C; c0 += tl.Ct) {
    const int64_t ct = (C - c0 < tl.Ct) ? C - c0 : tl.Ct;
    p->prof.begin(st, QI_STAGE_MULTIPLY);
    QI_TRY(launch_pack_pad<T>(sig + c0 * n, X, ct, n, n, st));
    QI_TRY(fft_c2c<T>(p->fft, X, n, ct, HIPFFT_FORWARD, st));
    QI_TRY(launch_stx_window<T>(X, Y, ct, bt, n, p->d_stx_idx + j0, p->d_stx_coef + j0, st));
    p->prof.end(QI_STAGE_MULTIPLY, st);
    p->prof.begin(st, QI_STAGE_INVERSE);
    QI_TRY(fft_c2c<T>(p->fft, Y, n, ct * bt, HIPFFT_BACKWARD, st));
    p->prof.end(QI_STAGE_INVERSE, st);
    p->prof.begin(st, QI_STAGE_EPILOGUE);
    EpiArgs<T> a{};
    a.Y = Y;
    a.L = n;
    a.n = n;
    a.off = 0;
    a.Ct = ct;
    a.Bt = bt;
    a.B = B;
    a.j0 = j0;
    a.coef = out->coef ? static_cast<cplx<T>*>(out->coef) + c0 * B * n : nullptr;
    a.bits = out->bits ? static_cast<T*>(out->bits) + c0 * B * n : nullptr;
    a.power_time = out->power_time ? static_cast<T*>(out->power_time) + c0 * n : nullptr;
    a.part_band = want_band ? part_band : nullptr;
    a.part_stat = want_stat ? part_stat : nullptr;
    a.tile_b = 1;  // (not the first tile of the panel: the per-time sums are ADDED to what the native run wrote)
    a.ntile_b = 2;
    a.power_scale = (T)(out->power_scale == 0.0 ? 1.0 : out->power_scale);
    a.eps = (T)(out->eps == 0.0 ? 2.220446049250313e-16 : out->eps);
    QI_TRY(launch_epilogue<T>(a, st));
    if (want_band || want_stat) {
      k_left_merge<<<dim3((unsigned)ct), 256, 0, st>>>(part_band, part_stat,
                                                     want_band ? static_cast<double*>(out->power_band) + c0 * B : nullptr,
                                                     want_stat ? static_cast<double*>(out->stats) + c0 * 4 : nullptr, B, j0, bt,
                                                     tl.nblk);
      QI_HIP(hipGetLastError());
    }
    p->prof.end(QI_STAGE_EPILOGUE, st);
  }
  return QI_OK;
}
template int run_stx_leftover<float>(qi_plan*, const void*, int64_t, const qi_tfr_out*, hipStream_t);
template int run_stx_leftover<double>(qi_plan*, const void*, int64_t, const qi_tfr_out*, hipStream_t);

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
int run_native(qi_plan* p, int kind, const void* sig_v, int64_t C, const qi_tfr_out* out, hipStream_t st, bool may_share,
               FusedCarry* defer, FusedCarry* finish, size_t* probe) {
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
  // qi_cwt_stx: a CWT run whose records fit one tile leaves its block launch and tail to the Stockwell run ...
  const bool deferring = defer && kind == 0 && Ct == C && blocks && !shorts && tail_one;
  // ... which keeps the CWT run's scratch intact (its own follows it; only the spectra are shared) and finishes both
  bool finishing = finish && finish->active && kind == 2 && share && blocks && tail_one;
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
  QI_LAYOUT_BEGIN(p, kind == 0 ? "run_native styx" : (kind == 1 ? "run_native atoms" : "run_native stx"), finishing);
  [[maybe_unused]] bool carving_shared = share;  // (the first region: spectra the CWT run of a joint tile left behind)
  auto carve = [&](size_t bytes) {
    char* r = w;
    w += align_up(bytes * Ct);
    QI_LAYOUT_NOTE(p, "native scratch", r, bytes * Ct, carving_shared);
    carving_shared = false;
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

  for (int64_t c0 = 0; c0 < C; c0 += Ct) {
    const int64_t ct = (C - c0 < Ct) ? C - c0 : Ct;
    // qi_cwt_stx, joint block launch: the styx and the Stockwell bands of a block from one forward transform; the edge items
    // of the split bands ride at its end (they add to the interpolation launch's output, which has run by then)
    const bool joint_blk = finishing && blocks && p->native_fuse > 1 && finish->ct == ct && !finish->demod && bt.demod &&
                           (finish->blk.coef != nullptr) == (out->coef != nullptr) &&
                           (finish->blk.bits != nullptr) == (out->bits != nullptr);
    auto launch_blocks = [&](hipStream_t bs) -> int {
      native::BlockArgs<T> b{};
      b.n = n;
      b.nitems = il.nitems;
      b.nlong = il.nlong;
      b.nedge_items = il.nedge_items;
      b.edge_merged = il.edge_merged ? 1 : 0;
      b.nsplit = nsplit;
      b.edge_band = p->d_split_bands;
      b.edge_bank = static_cast<const cplx<T>*>(p->split_bank);
      b.edge_part = zadd;
      b.panel_bands = (int32_t)B;
      b.items = il.d_items;
      b.bands = static_cast<const native::BlockBandT<T>*>(il.d_bands);
      b.bank = static_cast<const cplx<T>*>(bt.bank);
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
        QI_TRY(native::launch_block_dual<T>(finish->blk, b, p->d_dual[cut], p->n_dual[cut], p->n_dual_long[cut],
                                            p->blk[0].var[cut].nedge_items, ct, bs));
      } else {
        if (finishing) QI_TRY(native::launch_block<T>(finish->blk, finish->demod, finish->ct, bs));
        QI_TRY(native::launch_block<T>(b, bt.demod, ct, bs));
      }
      p->prof.end(QI_STAGE_BLOCK, bs);
      p->prof.unchain_span();
      return QI_OK;
    };
    if (clear_parts) QI_HIP(hipMemsetAsync(parts0, 0, parts_bytes, st));
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
      z.debug = p->native_debug;
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
    if (blocks) QI_TRY(launch_blocks(st));
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

// float64 records on the native engines in double arithmetic: forward transform of the records by hipFFT; the bands the
// float64 zoom takes (one gather + batched hipFFT + interpolation launch per coarse-grid level; split bands leave their
// tapered part in scratch), the block engine's bands (k_block64) and the split bands' edge items (k_block64_edge); whatever
// is left on the exact two-pass kernels (per launch group pass 1 for the wide bands, pass 2 with the pruned loader and
// the fused epilogue, 8-row workgroups, Cfg<double, 8>); one tail launch.
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
  // (the fine kernel's waves fill one slot per band and kZ64FineWave samples, k_z64_interp's workgroups one per kZ64Tile)
  bool fine = false;
  for (int c = 0; c < native::kZ64FineClasses; ++c) fine = fine || t.zf_count[c] > 0;
  const int64_t nblk_z = fine ? n / native::kZ64FineWave : n / native::kZ64Tile;
  int64_t nblk = subs[0].nblk > nblk_z ? subs[0].nblk : nblk_z;
  if (p->blk[kind].ready && kind != 1 && p->blk[kind].max_blocks > nblk) nblk = p->blk[kind].max_blocks;
  // (some bands leave slots unwritten: the block bands fill one slot per block of their reach group)
  const bool clear_parts = shorts || subs[0].nblk != nblk || (p->blk[kind].ready && kind != 1);
  // float64 zoom bands.  Coarse stage per grid level (gather, batched transform, pads; the levels' coarse arrays lie side by
  // side); fine stage: one k_z64_fine launch per class of the three coarsest grids, its bands dealt to `frow` rows, and one
  // k_z64_interp launch per finer level, its bands dealt to `zchunk` workgroups per tile.
  int zchunk[native::kZ64Levels] = {};
  int frow[native::kZ64FineClasses] = {};
  size_t z_off[native::kZ64Levels] = {};  // per record: offset (elements) of a level's coarse arrays
  size_t e_z = 0;
  const int64_t tiles_f = n / ((int64_t)native::kZ64FineWave * 4);
  for (int g = 0; g < native::kZ64Levels; ++g) {
    if (t.z64_count[g] == 0) continue;
    z_off[g] = e_z / sizeof(cplx<T>);
    e_z += (size_t)t.z64_count[g] * (size_t)(((Lf / 64) << g) + 2 * native::kZ64Pad) * sizeof(cplx<T>);
    if (fine && g < native::kZ64FineLevels) continue;
    int nc = (int)ceil_div(p->native_wgs, (n / native::kZ64Tile) * C);
    zchunk[g] = nc < 1 ? 1 : (nc > t.z64_count[g] ? t.z64_count[g] : nc);
    chunk_total += zchunk[g];
  }
  for (int c = 0; c < native::kZ64FineClasses; ++c) {
    if (t.zf_count[c] == 0) continue;
    // rows by work: a class's share of the launch budget (a band costs two fused multiply-adds per window sample plus the
    // epilogue), at least one row
    double work = 0.0, mine = (double)t.zf_count[c] * (2.0 * native::z64f_win(c) + 40.0);
    for (int q = 0; q < native::kZ64FineClasses; ++q) work += (double)t.zf_count[q] * (2.0 * native::z64f_win(q) + 40.0);
    // (native_z64_rows: rows of all classes together a call should have at least)
    const double want_rows = std::max<double>((double)p->native_z64_rows, (double)p->native_wgs / (double)(tiles_f * C));
    int nr = (int)std::ceil(want_rows * (mine / work));
    frow[c] = nr < 1 ? 1 : (nr > t.zf_count[c] ? t.zf_count[c] : nr);
    chunk_total += frow[c];
  }
  // block engine (short-atom bands with wide spectra, double arithmetic): its planes and stat slots come last
  const auto& bt = p->blk[kind];
  const bool blocks = kind != 1 && bt.ready;
  const auto& il = bt.var[C >= 4 ? 1 : 0];
  const int chunk_blk = chunk_total;
  int64_t blk_stats = 0;
  if (blocks) {
    chunk_total += il.nplanes;
    blk_stats = il.nitems + il.nedge_items;
  }
  const int32_t nsplit = kind == 0 && blocks ? p->nsplit : 0;  // split bands: the zoom launches hand their part to the edge items
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
  // forward transform of the records: the two-pass kernels in double where they exist (2^20- and 2^21-point transforms),
  // staged through one slot of the intermediate; hipFFT elsewhere
  const bool fwd_native = p->native_fwd && native_len_ok(Lf);
  if (fwd_native && imd_elems < Lf) imd_elems = Lf;
  const size_t e_x = (size_t)Lf * sizeof(cplx<T>);
  const size_t e_xn = shorts ? (size_t)n * sizeof(cplx<T>) : 0;
  const size_t e_imd = (size_t)imd_elems * sizeof(cplx<T>);
  const size_t e_pb = (size_t)B * nbk * 8, e_ps = (size_t)stat_slots * 24;
  const size_t e_tp = time_via_part ? (size_t)chunk_total * n * sizeof(T) : 0;
  const size_t e_ep = shorts ? (size_t)p->nedge * 2 * p->edge_wmax * sizeof(T) : 0;
  const size_t e_et = shorts ? (size_t)2 * p->edge_wmax * sizeof(T) : 0;
  const size_t e_ez = shorts && !out->coef ? (size_t)p->nedge * 2 * p->edge_wmax * sizeof(cplx<T>) : 0;
  const size_t e_add = (size_t)nsplit * n * sizeof(cplx<T>);
  const size_t per_chan = e_x + e_xn + e_imd + e_z + e_pb + e_ps + e_tp + e_ep + e_et + e_ez + e_add;
  if (p->ws_bytes < per_chan + 16384) {
    set_error("workspace of %zu bytes cannot hold one record's float64 scratch of %zu bytes", p->ws_bytes, per_chan + 16384);
    return QI_ERR_NOMEM;
  }
  int64_t Ct = (int64_t)((p->ws_bytes - 16384) / per_chan);
  if (Ct > C) Ct = C;
  p->shared_valid = false;
  char* w = p->ws;
  QI_LAYOUT_BEGIN(p, kind == 2 ? "run_native64 stx" : "run_native64 gabor", false);
  auto carve = [&](size_t bytes) {
    char* r = w;
    w += align_up(bytes * Ct);
    QI_LAYOUT_NOTE(p, "native64 scratch", r, bytes * Ct);
    return r;
  };
  cplx<T>* X = reinterpret_cast<cplx<T>*>(carve(e_x));
  cplx<T>* Xn = reinterpret_cast<cplx<T>*>(carve(e_xn));
  cplx<T>* imd = reinterpret_cast<cplx<T>*>(carve(e_imd));
  cplx<T>* Z = reinterpret_cast<cplx<T>*>(carve(e_z));
  cplx<T>* zadd = e_add ? reinterpret_cast<cplx<T>*>(carve(e_add)) : nullptr;
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
    if (fwd_native) {
      native::RowArgs<T> f{};
      f.Lf = Lf;
      f.n = n;
      f.N1 = Lf / native::kN2;
      f.N2 = native::kN2;
      f.imd_slots = 1;
      f.imd = imd;
      f.sig = sig + c0 * n;
      f.two_over_len = (float)(2.0 / (double)Lf);
      f.debug = 0;
      QI_TRY(native::launch_forward<T>(f, X, ct, st));
    } else {
      QI_TRY(launch_pack_pad<T>(sig + c0 * n, X, ct, n, Lf, st));
      QI_TRY(fft_c2c<T>(p->fft, X, Lf, ct, HIPFFT_FORWARD, st));
    }
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
    // coarse stage of every level, then the fine launches (heaviest classes first)
    native::Z64Args zl[native::kZ64Levels];
    for (int g = 0; g < native::kZ64Levels; ++g) {
      if (t.z64_count[g] == 0) continue;
      native::Z64Args& z = zl[g];
      z = native::Z64Args{};
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
      z.Z = Z + z_off[g] * (size_t)ct;  // (levels side by side: [level][record][band][pad | M | pad])
      z.weights = p->d_z64_w[g];
      z.inv_len = 1.0 / (double)Lf;
      z.two_over_len = (float)(2.0 / (double)Lf);
      z.coef = coef;
      z.bits = bits;
      z.split_part = zadd;
      z.split_rows = nsplit;
      z.time_part = tpart;
      z.part_band = want_band ? part_band : nullptr;
      z.part_stat = want_stat ? part_stat : nullptr;
      z.nblk = n / native::kZ64Tile;
      z.pb_stride = nbk;
      z.stat_nblk = nblk;
      z.stat_stride = stat_slots;
      z.chunk_total = chunk_total;
      z.power_scale = power_scale;
      z.eps = eps;
      p->prof.begin(st, QI_STAGE_ZOOM_COARSE);
      if (g < p->native_z64_coarse && z.M >= native::kBlk) {  // the coarsest grids: gather, transform and pads in one launch
        QI_TRY(native::launch_z64_coarse(z, ct, st));
      } else {
        QI_TRY(native::launch_z64_gather(z, ct, st));
        QI_TRY(fft_z2z_rows(p->fft, z.Z + native::kZ64Pad, z.M, z.M + 2 * native::kZ64Pad, (int64_t)z.nbands * ct, HIPFFT_BACKWARD, st));
        QI_TRY(native::launch_z64_pad(z.Z, z.M, (int64_t)z.nbands * ct, st));
      }
      p->prof.end(QI_STAGE_ZOOM_COARSE, st);
    }
    p->prof.begin(st, QI_STAGE_ZOOM);
    for (int ci = 0; ci < native::kZ64FineClasses; ++ci) {
      const int c = native::kZ64FineClasses - 1 - ci;  // (the shortest interpolators -- the classes with the most bands -- first)
      if (t.zf_count[c] == 0) continue;
      const int g = native::z64f_level(c);
      native::Z64FineArgs f{};
      f.z = zl[g];
      f.z.bands = t.d_z64 + t.zf_first[c];
      f.z.nbands = t.zf_count[c];
      f.z.nblk = n / native::kZ64FineWave;
      f.z.chunk_base = chunk_base;
      f.cls = c;
      f.nrow = frow[c];
      f.lvl_bands = t.z64_count[g];
      f.lvl_index0 = t.zf_first[c] - t.z64_first[g];
      f.w = p->d_z64f_w[c];
      f.lane_ph = t.d_z64_lane_ph ? t.d_z64_lane_ph + (int64_t)t.zf_first[c] * 65 : nullptr;
      f.wave_ph = t.d_z64_wave_ph;
      f.debug = p->native_debug;
      QI_TRY(native::launch_z64_fine(f, ct, st));
      chunk_base += frow[c];
    }
    for (int g = 0; g < native::kZ64Levels; ++g) {
      if (zchunk[g] == 0) continue;
      zl[g].chunk_base = chunk_base;
      QI_TRY(native::launch_z64_interp(zl[g], zchunk[g], ct, st));
      chunk_base += zchunk[g];
    }
    p->prof.end(QI_STAGE_ZOOM, st);
    if (blocks) {
      native::BlockArgs<T> b{};
      b.n = n;
      b.nitems = il.nitems;
      b.nedge_items = nsplit > 0 ? il.nedge_items : 0;
      b.edge_merged = il.edge_merged ? 1 : 0;
      b.nsplit = nsplit;
      b.edge_band = p->d_split_bands;
      b.edge_bank = static_cast<const cplx<T>*>(p->split_bank);
      b.edge_part = zadd;
      b.panel_bands = (int32_t)B;
      b.items = il.d_items;
      b.bands = static_cast<const native::BlockBandT<T>*>(il.d_bands);
      b.bank = static_cast<const cplx<T>*>(bt.bank);
      b.edge_wq = (int32_t)(p->native_split_e / 512);
      b.gauss_w = static_cast<const T*>(il.d_gauss_w);
      b.demod_pow = static_cast<const cplx<T>*>(il.d_demod_pow);
      b.demod_t1 = p->d_demod_t1;
      b.demod_t2 = p->d_demod_t2;
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
      b.debug = p->native_debug;
      b.stamps = p->blk_stamps;
      p->prof.begin(st, QI_STAGE_BLOCK);
      if (b.nedge_items > 0 && !p->side) {
        QI_HIP(hipStreamCreateWithFlags(&p->side, hipStreamNonBlocking));
        QI_HIP(hipEventCreateWithFlags(&p->ev_fork, hipEventDisableTiming));
        QI_HIP(hipEventCreateWithFlags(&p->ev_join, hipEventDisableTiming));
      }
      QI_TRY(native::launch_block<T>(b, bt.demod, ct, st, p->side, p->ev_fork, p->ev_join));
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

template int run_transform<float>(qi_plan*, Kind, const void*, int64_t, const qi_tfr_out*, hipStream_t);
template int run_transform<double>(qi_plan*, Kind, const void*, int64_t, const qi_tfr_out*, hipStream_t);
template int run_native<float>(qi_plan*, int, const void*, int64_t, const qi_tfr_out*, hipStream_t, bool, FusedCarry*, FusedCarry*,
                               size_t*);

}  // namespace host
}  // namespace qi
