// Error text, development switches and the hipFFT wrappers shared by the host files of libqi_tfr.so.
#include "qi_host.hpp"

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

const char* last_error() { return g_err; }

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
int fft_z2z_rows(FftCache& fc, double2* data, int64_t len, int64_t dist, int64_t batch, int dir, hipStream_t st) {
  hipfftHandle h;
  QI_TRY(fc.get(HIPFFT_Z2Z, len, batch, &h, dist));
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

std::mutex g_stft_mu;
std::map<int, FftCache> g_stft_fft;  // per device

}  // namespace qi

#ifdef QI_HOST_SANITIZE
// Host sanitizer build only (tests/sanitize): the regions a run carves out of the plan's scratch, checked as they are reported.
namespace qi {
namespace host {
namespace {
struct LayoutRegion {
  const char* base;
  size_t bytes;
  std::string what;
  int gen;
};
std::vector<LayoutRegion> g_regions;
int g_layout_gen = 0;
size_t g_layout_checked = 0;
}  // namespace

void layout_begin(const qi_plan* p, const char* run, bool keep_previous) {
  (void)p;
  (void)run;
  if (!keep_previous) g_regions.clear();
  ++g_layout_gen;
}

void layout_note(const qi_plan* p, const char* what, const void* ptr, size_t bytes, bool shared) {
  const char* q = static_cast<const char*>(ptr);
  ++g_layout_checked;
  if (bytes == 0) return;
  if (q < p->ws || q + bytes > p->ws + p->ws_bytes) {
    fprintf(stderr, "layout: region '%s' [%td, +%zu) leaves the workspace of %zu bytes\n", what, q - p->ws, bytes, p->ws_bytes);
    abort();
  }
  for (const auto& r : g_regions) {
    const bool overlap = q < r.base + r.bytes && r.base < q + bytes;
    if (!overlap) continue;
    if (shared && r.gen != g_layout_gen && q == r.base && bytes <= r.bytes) continue;  // the spectra a joint tile shares
    fprintf(stderr, "layout: region '%s' [%td, +%zu) overlaps '%s' [%td, +%zu)\n", what, q - p->ws, bytes, r.what.c_str(),
            r.base - p->ws, r.bytes);
    abort();
  }
  g_regions.push_back({q, bytes, what, g_layout_gen});
}
}  // namespace host
}  // namespace qi
extern "C" size_t qi_layout_regions_checked() { return qi::host::g_layout_checked; }
#endif
