// Host sanitizer walk ("build variants with -fsanitize=address for host code").  The library's translation units,
// compiled for the HOST ONLY under AddressSanitizer + UndefinedBehaviorSanitizer and linked with a stand-in HIP runtime
// (fake_hip.cpp: kernels do not run, device memory is host memory), are driven through the C ABI over every
//   order 1 .. 12  x  record length 2^14 .. 2^22  x  float32 / float64  x  1 / 4 / 16 / 64 records
// with the band tables the Python host code makes (gen_tables.py) and the scratch TfrPlan.workspace_for sizes for that batch:
// plan build (band assignment, zoom classes, block item lists, split bands), qi_cwt_stx / qi_cwt / qi_stx with several output
// sets -- panels, bits, full and band-only reductions -- (scratch carving, tiles, joint launches, launch geometry).  Checked on the way:
//   * every scratch region a run carves lies inside the workspace and no two live regions overlap (QI_LAYOUT_* hooks in
//     qi_run.hip -> layout_note, qi_host_util.hip);
//   * every table upload stays inside its allocation (AddressSanitizer on the malloc'ed "device" tables);
//   * the block engine's work-item lists: each band's blocks cover the record exactly once, planes and statistics slots are
//     in range and unique, the joint list of qi_cwt_stx holds every item of both tables exactly once;
//   * launch geometry (fake hipLaunchKernel), signed overflow / shifts / misaligned access in the host arithmetic (UBSan).
// Test infrastructure: built and run by tests/test_host_sanitize.py on the CPU container; never part of libqi_tfr.so.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <set>
#include <vector>

#include "qi_host.hpp"

extern "C" size_t qi_fake_hip_launches();
extern "C" size_t qi_layout_regions_checked();

namespace {

struct Config {
  int32_t order, log2n, dtype, B;
  std::vector<int64_t> records, ws;
  std::vector<double> p_re, p_im, omega, amp, sigma;
  std::vector<int64_t> idx;
};

[[noreturn]] void die(const char* what, const Config& c, int64_t C) {
  fprintf(stderr, "walk: %s (order %d, 2^%d samples, %s, %lld records): %s\n", what, c.order, c.log2n, c.dtype ? "f64" : "f32",
          (long long)C, qi_last_error());
  exit(1);
}

std::vector<Config> read_tables(const char* path) {
  FILE* f = fopen(path, "rb");
  if (!f) {
    perror(path);
    exit(2);
  }
  int32_t head[4];
  if (fread(head, 4, 4, f) != 4 || head[0] != 0x51495354) exit(2);
  std::vector<Config> out((size_t)head[1]);
  for (auto& c : out) {
    int32_t h[4];
    if (fread(h, 4, 4, f) != 4) exit(2);
    c.order = h[0], c.log2n = h[1], c.dtype = h[2], c.B = h[3];
    c.records.resize((size_t)head[2]);
    c.ws.resize((size_t)head[2]);
    if (fread(c.records.data(), 8, c.records.size(), f) != c.records.size()) exit(2);
    if (fread(c.ws.data(), 8, c.ws.size(), f) != c.ws.size()) exit(2);
    for (auto* v : {&c.p_re, &c.p_im, &c.omega, &c.amp, &c.sigma}) {
      v->resize((size_t)c.B);
      if (fread(v->data(), 8, (size_t)c.B, f) != (size_t)c.B) exit(2);
    }
    c.idx.resize((size_t)c.B);
    if (fread(c.idx.data(), 8, (size_t)c.B, f) != (size_t)c.B) exit(2);
  }
  fclose(f);
  return out;
}

// the block engine's item lists of table `kind`, cut `v`
void check_block_items(const qi_plan* p, int kind, int v, const Config& c) {
  const auto& bt = p->blk[kind];
  if (!bt.ready) return;
  const auto& il = bt.var[v];
  const int64_t n = p->n;
  const size_t total = (size_t)il.nitems + (size_t)il.nedge_items;
  if (il.h_items.size() != total) die("block item list: host copy and counts disagree", c, 0);
  std::set<int32_t> slots;
  // (band position in the list, block) pairs seen, per reach code
  std::set<std::pair<int64_t, int64_t>> seen;
  std::vector<int64_t> blocks_of_band;  // by list position
  int32_t nlong = 0;
  for (size_t i = 0; i < total; ++i) {
    const auto& it = il.h_items[i];
    if (!slots.insert(it.stat_slot).second || it.stat_slot < 0 || it.stat_slot >= (int32_t)total) die("block items: statistics slot", c, 0);
    if (it.plane < 0 || it.plane >= il.nplanes) die("block items: per-time plane out of range", c, 0);
    if (i < (size_t)il.nitems) {
      if (it.wq != 1 && it.wq != 2 && it.wq != 4 && it.wq != qi::native::kBlkLongWq) die("block items: reach code", c, 0);
      if (it.wq == qi::native::kBlkLongWq) {
        if ((int32_t)i != nlong) die("block items: long-block items must lead the list", c, 0);
        ++nlong;
      }
      const int64_t V = qi::native::block_valid(it.wq), nb = (n + V - 1) / V;
      if (it.block < 0 || it.block >= nb || it.band_count <= 0 || it.band_first < 0) die("block items: block / band range", c, 0);
      for (int32_t b = it.band_first; b < it.band_first + it.band_count; ++b) {
        if (!seen.insert({b, it.block}).second) die("block items: a (band, block) pair twice", c, 0);
        if ((size_t)b >= blocks_of_band.size()) blocks_of_band.resize((size_t)b + 1, 0);
        blocks_of_band[(size_t)b] += 1;
      }
    } else {
      if (it.wq >= 0) die("block items: an edge item without a negative reach code", c, 0);
      const int64_t V = qi::native::block_valid(-it.wq), nb = (n + V - 1) / V;
      if (it.block < 0 || it.block >= nb) die("edge items: block out of range", c, 0);
    }
  }
  if (nlong != il.nlong) die("block items: long-block count", c, 0);
  // every band of the list has ALL blocks of its reach group: band positions are dense, counts equal a group's block count
  for (size_t b = 0; b < blocks_of_band.size(); ++b) {
    const int64_t got = blocks_of_band[b];
    bool ok = false;
    for (int wq : {1, 2, 4, (int)qi::native::kBlkLongWq}) ok = ok || got == (n + qi::native::block_valid(wq) - 1) / qi::native::block_valid(wq);
    if (!ok) die("block items: a band's blocks do not cover the record", c, 0);
  }
  if ((int32_t)blocks_of_band.size() != bt.rows && il.nitems > 0) die("block items: bands of the list vs rows of the table", c, 0);
}

void check_dual_items(const qi_plan* p, int cut, const Config& c) {
  if (!p->dual_valid[cut] || !p->d_dual[cut]) return;
  const auto& l0 = p->blk[0].var[cut];
  const auto& l2 = p->blk[2].var[cut];
  std::set<std::tuple<int, int, int, int>> want0, want2;  // (wq, block, first, count)
  for (int32_t i = 0; i < l0.nitems + l0.nedge_items; ++i) {
    const auto& it = l0.h_items[(size_t)i];
    want0.insert({it.wq, it.block, it.band_first, it.band_count});
  }
  for (int32_t i = 0; i < l2.nitems; ++i) {
    const auto& it = l2.h_items[(size_t)i];
    want2.insert({it.wq, it.block, it.band_first, it.band_count});
  }
  const qi::native::DualItem* d = p->d_dual[cut];  // ("device" memory is host memory here)
  for (int32_t i = 0; i < p->n_dual[cut]; ++i) {
    const auto& it = d[i];
    if (it.wq < 0 || it.count0 > 0) {
      if (want0.erase({it.wq, it.block, it.first0, it.count0}) != 1) die("joint items: a styx item that is not in the styx list (or twice)", c, 0);
    }
    if (it.wq > 0 && it.count2 > 0) {
      if (want2.erase({it.wq, it.block, it.first2, it.count2}) != 1) die("joint items: a Stockwell item that is not in its list (or twice)", c, 0);
    }
  }
  if (!want0.empty() || !want2.empty()) die("joint items: items of a table missing from the joint list", c, 0);
}

void check_tables(const qi_plan* p, const Config& c) {
  for (int kind : {0, 2}) {
    const auto& t = p->nat[kind];
    if (!t.ready) continue;
    int64_t planes = 0;
    for (const auto& z : t.h_zoom) planes += ((t.Lf / qi::native::kZoomD) << qi::native::zoom_grid(z.second)) / qi::native::kBlk;
    if (planes != t.zoom_planes || (int32_t)t.h_zoom.size() != t.nzoom) die("zoom table: planes / band count", c, 0);
    // every panel row has exactly one producer: zoom / float64 zoom / block / two-pass (or the hipFFT pass behind the run)
    const int64_t B = kind == 2 ? p->nb_stx : p->nb[kind];
    const int64_t blk = p->blk[kind].ready ? p->blk[kind].rows : 0;
    const int64_t left = kind == 2 ? p->stx_left_n : 0;
    const int64_t shorts = kind == 0 && p->nat[3].ready ? p->nedge : 0;
    if (t.nzoom + t.nz64 + blk + (int64_t)t.h_rows.size() + left + shorts != B) die("table: producers do not add up to the panel's rows", c, 0);
    for (int v = 0; v < 2; ++v) check_block_items(p, kind, v, c);
  }
}

}  // namespace

int main(int argc, char** argv) {
  if (argc < 2) {
    fprintf(stderr, "usage: walk tables.bin [max configs]\n");
    return 2;
  }
  const auto configs = read_tables(argv[1]);
  const size_t limit = argc > 2 ? (size_t)atoll(argv[2]) : configs.size();
  size_t plans = 0, calls = 0, native = 0;
  for (size_t ci = 0; ci < configs.size() && ci < limit; ++ci) {
    const Config& c = configs[ci];
    const int64_t n = (int64_t)1 << c.log2n;
    const size_t rsz = c.dtype ? 8 : 4;
    for (size_t r = 0; r < c.records.size(); ++r) {
      const int64_t C = c.records[r];
      qi_plan_desc desc{};
      desc.n = n;
      desc.dtype = c.dtype;
      desc.device = 0;
      desc.engine = QI_ENGINE_AUTO;
      desc.workspace_bytes = c.ws[r];
      qi_plan* p = nullptr;
      if (qi_plan_create(&p, &desc) != QI_OK) die("qi_plan_create", c, C);
      if (qi_plan_set_gabor_bank(p, QI_BANK_STYX, c.B, c.p_re.data(), c.p_im.data(), c.omega.data(), c.amp.data(), nullptr) != QI_OK)
        die("qi_plan_set_gabor_bank", c, C);
      if (qi_plan_set_stx_bands(p, c.B, c.idx.data(), c.sigma.data()) != QI_OK) die("qi_plan_set_stx_bands", c, C);
      ++plans;
      native += p->nat[0].ready && p->nat[2].ready;
      check_tables(p, c);
      // caller buffers: records, panels (address-space reservations), reduced products
      void *sig = nullptr, *coef0 = nullptr, *coef2 = nullptr, *bits = nullptr, *red = nullptr;
      const size_t panel = (size_t)C * c.B * n;
      if (hipMalloc(&sig, (size_t)C * n * rsz) != hipSuccess || hipMalloc(&coef0, panel * 2 * rsz) != hipSuccess ||
          hipMalloc(&coef2, panel * 2 * rsz) != hipSuccess || hipMalloc(&bits, panel * rsz) != hipSuccess ||
          hipMalloc(&red, 2 * ((size_t)C * (c.B + 4) * 8 + (size_t)C * n * rsz)) != hipSuccess)
        die("caller buffers", c, C);
      char* rp = static_cast<char*>(red);
      auto outs = [&](void* coef, void* b, bool reductions, int which) {
        qi_tfr_out o{};
        o.coef = coef;
        o.bits = b;
        if (reductions) {
          char* base = rp + (size_t)which * ((size_t)C * (c.B + 4) * 8 + (size_t)C * n * rsz);
          o.power_time = base;
          o.power_band = base + (size_t)C * n * rsz;
          o.stats = base + (size_t)C * n * rsz + (size_t)C * c.B * 8;
        }
        return o;
      };
      const qi_tfr_out a0 = outs(coef0, nullptr, true, 0), a2 = outs(coef2, nullptr, true, 1);
      if (qi_cwt_stx(p, QI_BANK_STYX, sig, C, &a0, &a2, nullptr) != QI_OK) die("qi_cwt_stx", c, C);
      for (int cut = 0; cut < 2; ++cut) check_dual_items(p, cut, c);
      const qi_tfr_out lean0 = outs(nullptr, nullptr, true, 0), lean2 = outs(nullptr, nullptr, true, 1);
      if (qi_cwt_stx(p, QI_BANK_STYX, sig, C, &lean0, &lean2, nullptr) != QI_OK) die("qi_cwt_stx (reductions only)", c, C);
      const qi_tfr_out full = outs(coef0, bits, true, 0), bare = outs(coef2, nullptr, false, 1);
      if (qi_cwt(p, QI_BANK_STYX, sig, C, &full, nullptr) != QI_OK) die("qi_cwt (coefficients, bits, reductions)", c, C);
      if (qi_stx(p, sig, C, &bare, nullptr) != QI_OK) die("qi_stx (coefficients only)", c, C);
      if (qi_stx(p, sig, C, &full, nullptr) != QI_OK) die("qi_stx (coefficients, bits, reductions)", c, C);
      calls += 5;
      {  // band powers and statistics without the per-time marginal (engine: reductions="band", the streaming pipeline's request)
        qi_tfr_out b0 = lean0, b2 = lean2, bf = full;
        b0.power_time = b2.power_time = bf.power_time = nullptr;
        if (qi_cwt_stx(p, QI_BANK_STYX, sig, C, &b0, &b2, nullptr) != QI_OK) die("qi_cwt_stx (band-only reductions)", c, C);
        if (qi_cwt(p, QI_BANK_STYX, sig, C, &bf, nullptr) != QI_OK) die("qi_cwt (coefficients, bits, band-only reductions)", c, C);
        if (qi_stx(p, sig, C, &b2, nullptr) != QI_OK) die("qi_stx (band-only reductions)", c, C);
        calls += 3;
      }
      if (C == 4 && c.order == 3) {  // the atoms bank (cwt_atoms: circular kind) on a few shapes, then the plan's tables again
        if (qi_plan_set_gabor_bank(p, QI_BANK_ATOMS, c.B, c.p_re.data(), c.p_im.data(), c.omega.data(), c.amp.data(), nullptr) != QI_OK)
          die("qi_plan_set_gabor_bank (atoms)", c, C);
        if (qi_cwt(p, QI_BANK_ATOMS, sig, C, &full, nullptr) != QI_OK) die("qi_cwt (atoms bank)", c, C);
        ++calls;
      }
      for (void* q : {sig, coef0, coef2, bits, red}) (void)hipFree(q);
      if (qi_plan_destroy(p) != QI_OK) die("qi_plan_destroy", c, C);
    }
    if ((ci & 15) == 15) fprintf(stderr, "walk: %zu of %zu tables done\n", ci + 1, configs.size());
  }
  printf("{\"ok\": true, \"plans\": %zu, \"plans_on_native_engines\": %zu, \"calls\": %zu, \"kernel_launches\": %zu, \"scratch_regions_checked\": %zu}\n",
         plans, native, calls, qi_fake_hip_launches(), qi_layout_regions_checked());
  return 0;
}
