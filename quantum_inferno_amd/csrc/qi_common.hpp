// Shared helpers for libqi_tfr.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hipfft/hipfft.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "qi_tfr.h"

namespace qi {

void set_error(const char* fmt, ...);
const char* tune_env(const char* name);  // development switch (read only when QI_TUNE is set; qi_api.hip)

#define QI_HIP(call)                                                                        \
  do {                                                                                      \
    hipError_t e_ = (call);                                                                 \
    if (e_ != hipSuccess) {                                                                 \
      ::qi::set_error("%s:%d %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(e_)); \
      return QI_ERR_HIP;                                                                    \
    }                                                                                       \
  } while (0)

#define QI_FFT(call)                                                              \
  do {                                                                            \
    hipfftResult r_ = (call);                                                     \
    if (r_ != HIPFFT_SUCCESS) {                                                   \
      ::qi::set_error("%s:%d %s -> hipfft %d", __FILE__, __LINE__, #call, (int)r_); \
      return QI_ERR_FFT;                                                          \
    }                                                                             \
  } while (0)

#define QI_TRY(call)        \
  do {                      \
    int s_ = (call);        \
    if (s_ != QI_OK) return s_; \
  } while (0)

#define QI_REQUIRE(cond, ...)       \
  do {                              \
    if (!(cond)) {                  \
      ::qi::set_error(__VA_ARGS__); \
      return QI_ERR_ARG;            \
    }                               \
  } while (0)

template <typename T>
struct c2;
template <>
struct c2<float> {
  using type = float2;
};
template <>
struct c2<double> {
  using type = double2;
};
template <typename T>
using cplx = typename c2<T>::type;

template <typename T>
__host__ __device__ inline cplx<T> mk(T re, T im) {
  cplx<T> v;
  v.x = re;
  v.y = im;
  return v;
}
template <typename C>
__host__ __device__ inline C cmul(C a, C b) {
  C r;
  r.x = a.x * b.x - a.y * b.y;
  r.y = a.x * b.y + a.y * b.x;
  return r;
}

// Allow kernel `fn` `bytes` of dynamic LDS on the CURRENT device.  hipFuncSetAttribute acts on the current device's
// function object only, so the raised limits are kept per (device, function); a request above what the device offers
// returns QI_ERR_UNSUPPORTED instead of a launch error (qi_kernels.hip).
int allow_dynamic_lds(const void* fn, size_t bytes);

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }
inline bool is_pow2(int64_t v) { return v > 0 && (v & (v - 1)) == 0; }
inline int64_t next_pow2(int64_t v) {
  int64_t p = 1;
  while (p < v) p <<= 1;
  return p;
}

// Time samples one epilogue workgroup owns (256 threads x 4 consecutive samples).
constexpr int kEpiThreads = 256;
constexpr int kEpiVec = 4;
constexpr int kEpiSpan = kEpiThreads * kEpiVec;

// ---- launchers implemented in qi_kernels.hip (all asynchronous on `st`) ------------------------
template <typename T>
int launch_pack_pad(const T* sig, cplx<T>* X, int64_t C, int64_t n, int64_t L, hipStream_t st);

int launch_bank_rows(double2* rows, int64_t n, int64_t L, int circular, const double* p_re, const double* p_im,
                     const double* omega, const double* amp, int j0, int nb, hipStream_t st, double taper_e = 0.0,
                     const double* xs = nullptr);
template <typename T>
int launch_bank_convert(const double2* F, cplx<T>* bank, int64_t count, int conj, double scale, hipStream_t st);

template <typename T>
int launch_mul_bank(const cplx<T>* X, const cplx<T>* H, cplx<T>* Y, int64_t Ct, int64_t Bt, int64_t L, hipStream_t st);
template <typename T>
int launch_stx_window(const cplx<T>* X, cplx<T>* Y, int64_t Ct, int64_t Bt, int64_t n, const int64_t* idx,
                      const double* coef, hipStream_t st);

template <typename T>
struct EpiArgs {
  const cplx<T>* Y;  // [Ct][Bt][L]
  int64_t L, n, off; // out[t] = Y[(t + off) mod L]
  int64_t Ct, Bt;    // tile extents
  int64_t B;         // bands in the whole panel
  int64_t j0;        // first band of this tile
  cplx<T>* coef;     // [Ct][B][n] (already offset to the tile's first channel) or null
  T* bits;           // idem
  T* power_time;     // [Ct][n] or null
  double* part_band; // [Ct][B][nblk] or null
  double* part_stat; // [Ct][ntile_b][nblk][3] or null
  int64_t tile_b;    // index of this band tile
  int64_t ntile_b;
  T power_scale;
  T eps;
};
template <typename T>
int launch_epilogue(const EpiArgs<T>& a, hipStream_t st);
// sliding-window STFT in scipy.signal.ShortTimeFFT's convention (qi_stft_sliding.hip)
template <typename T>
int launch_sliding_frames(const T* sig, const T* win, T* frames, int64_t C, int64_t n, int64_t seg, int64_t hop,
                          int64_t nfft, int64_t nseg, int64_t first, int pad_mode, int detrend, int64_t roll,
                          hipStream_t st);
template <typename T>
int launch_sliding_transpose(const cplx<T>* F, cplx<T>* Z, T* R, int kind, int64_t C, int64_t nseg, int64_t nf,
                             hipStream_t st);
template <typename T>
int launch_sliding_untranspose(const cplx<T>* S, cplx<T>* F, int64_t C, int64_t nseg, int64_t nf, hipStream_t st);
template <typename T>
int launch_sliding_overlap_add(const T* slices, const T* dual, T* out, int64_t C, int64_t k0, int64_t k1, int64_t seg,
                               int64_t hop, int64_t nfft, int64_t nseg, int64_t first, int64_t roll, hipStream_t st);

// 1-D Shannon family (qi_shannon1d.hip)
int64_t shannon_spans(int64_t n);
template <typename T>
int launch_shannon_1d(const T* m, int64_t C, int64_t n, T* info, T* entropy, T* isnr, T* esnr, hipStream_t st);
template <typename T>
int launch_tdr_marginal(const T* sig, int64_t C, int64_t n, T* sig_norm, T* marginal, double* partial, hipStream_t st);
template <typename T>
int launch_fft_marginal(const cplx<T>* X, int64_t C, int64_t nf, T* angle, T* marginal, double* partial, int32_t* turns,
                        hipStream_t st);

int launch_finalize(const double* part_band, const double* part_stat, double* power_band, double* stats, int64_t C,
                    int64_t B, int64_t nblk, int64_t nstat, hipStream_t st, const int32_t* band_slots = nullptr);

template <typename T>
int launch_stft_frames(const T* sig, const T* win, T* frames, int64_t C, int64_t n, int64_t seg, int64_t hop,
                       int64_t nfft, int64_t nseg, int64_t lead, hipStream_t st);
// options of scipy.signal.ShortTimeFFT's convention on the fused STFT kernel (qi_sliding_stft): record extension past its
// ends (0 zeros, 1 edge, 2 / 3 even / odd reflection), mean removal, left rotation of the slice, real output (1 |X|, 2 |X|^2)
struct StftSliding {
  int32_t pad_mode, detrend, real_kind;
  int64_t roll;
};
// fused STFT (qi_stft_fused.hip): frames, transform and [frequency][time] store in one kernel
bool stft_fused_supported(int dtype, int64_t seg, int64_t hop, int64_t nfft);
template <typename T>
int launch_stft_fused(const T* sig, const T* win, cplx<T>* Z, T* bits, int64_t C, int64_t n, int64_t seg, int64_t hop,
                      int64_t nfft, int64_t nseg, int64_t lead, double scale, double eps, hipStream_t st,
                      double* welch_part = nullptr, const StftSliding* sliding = nullptr);
// fused inverse of the ShortTimeFFT-convention transform (QI_ERR_UNSUPPORTED, with no error text, where it does not apply)
template <typename T>
int launch_istft_fused(const cplx<T>* S, const T* dual, T* out, int64_t C, int64_t seg, int64_t hop, int64_t nfft, int64_t first,
                       int64_t nseg, int64_t roll, int64_t k0, int64_t k1, hipStream_t st);
// Welch mean on the fused kernel (`part`: [C][<= nseg][nfft / 2 + 1] doubles of scratch)
template <typename T>
int launch_welch_fused(const T* sig, const T* win, T* pxx, double* part, int64_t C, int64_t n, int64_t seg, int64_t hop,
                       int64_t nfft, int64_t nseg, double scale2, hipStream_t st);
template <typename T>
int launch_welch_mean(const cplx<T>* F, T* pxx, int64_t C, int64_t nseg, int64_t nf, int64_t nfft, T scale2,
                      hipStream_t st);
template <typename T>
int launch_stft_transpose(const cplx<T>* F, cplx<T>* Z, T* bits, int64_t C, int64_t nseg, int64_t nf, T scale, T eps,
                          hipStream_t st);

template <typename T>
int launch_power_marginals(const T* P, int64_t C, int64_t B, int64_t n, T* power_time, double* part_band,
                           double* part_stat, hipStream_t st);
template <typename T>
int launch_log2_offset(const T* in, T* out, int64_t C, int64_t count, T eps, const double* ref, hipStream_t st);
int launch_widen(const float* in, double* out, int64_t count, hipStream_t st);
template <typename T>
int launch_log2_abs(const T* in, int is_complex, T* out, int64_t count, T eps, hipStream_t st);
template <typename T>
int launch_shannon(const T* P, const T* mult, int mode, int64_t C, int64_t B, int64_t n, double deg, T* info, T* sb,
                   T* isnr, T* esnr, hipStream_t st);

}  // namespace qi
