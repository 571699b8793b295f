// C ABI of libqi_tfr.so, plan entry points: create / destroy, band tables, transforms, profiling.
#include "qi_host.hpp"

using namespace qi;
using namespace qi::host;

extern "C" {

int qi_abi_version(void) { return QI_TFR_ABI_VERSION; }
const char* qi_last_error(void) { return qi::last_error(); }

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
  QI_REQUIRE(desc->flags == 0, "qi_plan_desc.flags = %d: the field is reserved and must be 0", desc->flags);
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
  if (p->native_kmax > (int64_t)native::kMaxPrunedTerms * native::kN2)
    p->native_kmax = (int64_t)native::kMaxPrunedTerms * native::kN2;
  if (const char* e = tune_env("QI_NATIVE_DEBUG")) p->native_debug = atoi(e);
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
  if (const char* e = tune_env("QI_NATIVE_SPLIT")) p->native_split = atoi(e);
  if (const char* e = tune_env("QI_NATIVE_FUSE")) p->native_fuse = atoi(e);
  if (const char* e = tune_env("QI_NATIVE_F64")) p->native_f64 = atoi(e);
  if (const char* e = tune_env("QI_NATIVE_Z64")) p->native_z64 = atoi(e);
  if (const char* e = tune_env("QI_NATIVE_Z64_FINE")) p->native_z64_fine = atoi(e);
  if (const char* e = tune_env("QI_NATIVE_Z64_COARSE")) p->native_z64_coarse = atoi(e);
  if (const char* e = tune_env("QI_NATIVE_Z64_BLOCK_FROM")) p->native_z64_block_from = atoi(e);
  if (const char* e = tune_env("QI_NATIVE_BLK64_WTAB")) p->native_blk64_wtab = atoi(e);
  if (const char* e = tune_env("QI_NATIVE_BLK64_NARROW")) p->native_blk64_narrow = atoi(e);
  if (const char* e = tune_env("QI_NATIVE_MIN_LOG2N")) p->native_min_log2n = atoi(e);
  if (desc->dtype == QI_F64) {  // the float32 zoom / block / split engines are sized for the float32 tolerance
    if (!tune_env("QI_NATIVE_MIN_LOG2N")) p->native_min_log2n = 15;
    const int short64 = p->native_short && !(tune_env("QI_NATIVE_SHORT64") && atoi(tune_env("QI_NATIVE_SHORT64")) == 0);
    // (the block engine runs float64 tables in double arithmetic: analytic Gaussian bands, no narrow-spectrum shortcuts)
    const int block64 = p->native_block && !(tune_env("QI_NATIVE_BLOCK64") && atoi(tune_env("QI_NATIVE_BLOCK64")) == 0);
    // (split bands: tapered part on the float64 zoom, edge pieces on the float64 block engine)
    const int split64 = p->native_split && block64 && p->native_z64 &&
                        !(tune_env("QI_NATIVE_SPLIT64") && atoi(tune_env("QI_NATIVE_SPLIT64")) == 0);
    p->native_zoom = 0;
    p->native_split = split64;
    // (a longer taper than the float32 engines': the tapered spectrum is half as wide at the 2^-50 level the float64 zoom
    // keeps -- a coarser grid for every split band; measured 10.03 against 10.29 ms at order 12 x 4 records)
    p->native_split_e = 2048;
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
      if (const char* path = tune_env("QI_NATIVE_TIMELINE")) {  // the joint block launch's dispatch timeline (k_block_dual)
        if (FILE* f = fopen(path, "w")) {
          fprintf(f, "wg start_10ns end_10ns xcc cu bands wq\n");
          const unsigned long long m48 = 0xffffffffffffull;
          for (size_t w = 0; w < 65536; ++w)
            if (h[w * 8 + 7])
              fprintf(f, "%zu %llu %llu %llu %llu %llu %lld\n", w, h[w * 8 + 6] & m48, h[w * 8 + 7] & m48, h[w * 8 + 6] >> 56,
                      (h[w * 8 + 6] >> 48) & 0xff, (h[w * 8 + 7] >> 48) & 0xff, (long long)(signed char)(h[w * 8 + 7] >> 56));
          fclose(f);
        }
      }
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
  for (auto* w : p->d_z64f_w)
    if (w) (void)hipFree(w);
  if (p->d_demod_t1) (void)hipFree(p->d_demod_t1);
  if (p->d_demod_t2) (void)hipFree(p->d_demod_t2);
  for (auto& wc : p->d_zoom_w)
    for (auto* w : wc)
      if (w) (void)hipFree(w);
  if (p->d_edge) (void)hipFree(p->d_edge);
  if (p->split_bank) (void)hipFree(p->split_bank);
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
  p->table_gen++;
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
  p->table_gen++;
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
  QI_TRY(build_stx_tables(p, B, shift_index, sigma, coef));
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
  const int64_t left = which == 2 ? p->stx_left_n : 0;  // rows the hipFFT engine's pass behind the native run produces
  if (stage == QI_STAGE_BLOCK) return blk;
  if (stage == QI_STAGE_ZOOM) return zoom;
  if (stage == QI_STAGE_INVERSE) return left;
  return stage == QI_STAGE_PASS2 ? total - blk - zoom - left : 0;
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
  if (p->nat[2].ready) {
    int rc = p->d.dtype == QI_F64 ? run_native64(p, 2, sig, C, out, (hipStream_t)stream)
                                  : run_native<float>(p, 2, sig, C, out, (hipStream_t)stream);
    if (rc == QI_OK && p->stx_left_n > 0)  // the rows no native engine takes at this length: the hipFFT engine's pass over them
      rc = p->d.dtype == QI_F64 ? run_stx_leftover<double>(p, sig, C, out, (hipStream_t)stream)
                                : run_stx_leftover<float>(p, sig, C, out, (hipStream_t)stream);
    return rc;
  }
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
  const bool fuse = p->native_fuse && p->d.dtype == QI_F32 && p->nat[bank].ready && p->nat[2].ready && p->stx_left_n == 0;
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
    // (the settled size is kept per request shape: the probes are pure host work, but a step of one record is a quarter
    // of a millisecond)
    auto want = [](const qi_tfr_out* o) {
      return (o->coef ? 1u : 0u) | (o->bits ? 2u : 0u) | (o->power_band ? 4u : 0u) | (o->power_time ? 8u : 0u) | (o->stats ? 16u : 0u);
    };
    const unsigned flags = want(out_cwt) | (want(out_stx) << 8);
    if (p->tile_cache.C == C && p->tile_cache.flags == flags && p->tile_cache.gen == p->table_gen) {
      tile = p->tile_cache.tile;
    } else {
      for (int it = 0; it < 8 && tile >= 1; ++it) {
        size_t pc0 = 0, pc2 = 0;
        QI_TRY(run_native<float>(p, bank, sig, tile, out_cwt, st, false, &p->carry, nullptr, &pc0));
        QI_TRY(run_native<float>(p, 2, sig, tile, out_stx, st, true, nullptr, &p->carry, &pc2));
        const int64_t fit = p->ws_bytes > (1u << 16) ? (int64_t)((p->ws_bytes - (1u << 16)) / (pc0 + pc2)) : 0;
        if (fit >= tile) break;
        tile = fit;
      }
      p->tile_cache.C = C;
      p->tile_cache.flags = flags;
      p->tile_cache.gen = p->table_gen;
      p->tile_cache.tile = tile;
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
  if (p->d.dtype == QI_F32 && p->nat[bank].ready && p->nat[2].ready && p->stx_left_n == 0) {  // separate launches, but the Stockwell run may still use the CWT's spectra
    DeviceGuard g(p->d.device);
    p->prof.unchain();
    const int rc = run_native<float>(p, 2, sig, C, out_stx, st, /*may_share=*/true, nullptr, nullptr);
    p->shared_valid = false;
    return rc;
  }
  return qi_stx(p, sig, C, out_stx, stream);
}

}  // extern "C"
