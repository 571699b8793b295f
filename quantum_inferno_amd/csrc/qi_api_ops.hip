// C ABI of libqi_tfr.so, plan-less entry points: STFT / Welch / sliding STFT, tfr_info on caller panels, the 1-D Shannon
// family, widening and log2 helpers, Gabor atoms.
#include "qi_host.hpp"

using namespace qi;
using qi::host::align_up;

extern "C" int64_t qi_stft_segments(int64_t n, int64_t seg, int64_t hop);

namespace {

template <typename T>
int stft_impl(int device, const void* sig, int64_t C, int64_t n, const void* window, int64_t seg, int64_t hop,
                     int64_t nfft, double scale, void* Z, void* bits, double eps, char* scratch, hipStream_t st) {
  const int64_t nseg = qi_stft_segments(n, seg, hop);
  const int64_t nf = nfft / 2 + 1;
  static const bool fused_off = tune_env("QI_STFT_FUSED") && atoi(tune_env("QI_STFT_FUSED")) == 0;
  if (!fused_off && stft_fused_supported(sizeof(T) == 8 ? QI_F64 : QI_F32, seg, hop, nfft)) {  // one kernel: segments, transform and store from LDS
    const int rc = launch_stft_fused<T>(static_cast<const T*>(sig), static_cast<const T*>(window), static_cast<cplx<T>*>(Z),
                                        static_cast<T*>(bits), C, n, seg, hop, nfft, nseg, seg / 2, scale,
                                        eps == 0.0 ? 2.220446049250313e-16 : eps, st);
    if (rc != QI_ERR_UNSUPPORTED) return rc;  // (a device with less LDS per workgroup than the tile needs: the three-kernel path below)
  }
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
  static const bool fused_off = tune_env("QI_STFT_FUSED") && atoi(tune_env("QI_STFT_FUSED")) == 0;
  if (!fused_off && stft_fused_supported(sizeof(T) == 8 ? QI_F64 : QI_F32, seg, hop, nfft)) {  // segments, transform, |X|^2 sums in one kernel
    const int rc = launch_welch_fused<T>(static_cast<const T*>(sig), static_cast<const T*>(window), static_cast<T*>(pxx),
                                         reinterpret_cast<double*>(scratch), C, n, seg, hop, nfft, nseg, scale * scale, st);
    if (rc != QI_ERR_UNSUPPORTED) return rc;
  }
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
  static const bool fused_off = tune_env("QI_STFT_FUSED") && atoi(tune_env("QI_STFT_FUSED")) == 0;
  if (!fused_off && stft_fused_supported(sizeof(T) == 8 ? QI_F64 : QI_F32, seg, hop, nfft)) {
    // one kernel: slices (padding mode, optional detrend), transform, phase roll, [frequency][slice] store
    const StftSliding sl{pad_mode, detrend, R ? kind : 0, roll};
    const int rc = launch_stft_fused<T>(sig, window, Z, R, C, n, seg, hop, nfft, nseg, -first, 1.0, 0.0, st, nullptr, &sl);
    if (rc != QI_ERR_UNSUPPORTED) return rc;
  }
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
  static const bool fused_off = tune_env("QI_STFT_FUSED") && atoi(tune_env("QI_STFT_FUSED")) == 0;
  if (!fused_off) {  // one kernel: fold, inverse transform in LDS, overlap-add in gather form
    const int rc = launch_istft_fused<T>(S, dual, out, C, seg, hop, nfft, first, nseg, roll, k0, k1, st);
    if (rc != QI_ERR_UNSUPPORTED) return rc;  // (other transform lengths, or a device without the LDS: the three-kernel path)
  }
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
