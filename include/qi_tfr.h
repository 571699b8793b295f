/*
 * qi_tfr.h -- C ABI of libqi_tfr.so, the MI355X (gfx950) time-frequency hot path.
 *
 * The reference (ISLA-UH/quantum-inferno v1.1.3) is pure Python and has no FFI; the
 * boundary a maintainer would bind is therefore the set of public Python functions
 * cited on each entry point below (paths relative to quantum_inferno/ in the reference).
 * INTEGRATION.md shows the ctypes stub for each.
 *
 * Conventions
 *  - plain C: pointers, sizes, doubles.  No torch / hip types in any signature
 *    (qi_stream is a hipStream_t passed as void*; NULL = the default stream).
 *  - every data pointer is a CALLER-OWNED DEVICE pointer on the plan's device; band
 *    tables passed to the qi_plan_set_* calls are HOST pointers (float64 / int64),
 *    because band and index selection stays on the host in float64 (bit-exact with the
 *    reference, scales_dyadic.py:355-393, styx_stx.py:233).
 *  - every call returns 0 on success or a negative qi_status; qi_last_error() gives the
 *    message for the calling thread.  Nothing throws, nothing aborts.
 *  - transforms are asynchronous on the given stream and do no host synchronisation;
 *    a plan is thread-compatible (one plan per host thread / stream).
 *  - panels are [channel][band][time], C-contiguous, band frequency ascending, as the
 *    reference returns them for one channel (styx_cwt.py:198, styx_stx.py:236).
 *  - dtype selects the arithmetic: QI_F32 (float / float2) or QI_F64 (double / double2).
 */
#ifndef QI_TFR_H
#define QI_TFR_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define QI_TFR_ABI_VERSION 1

typedef struct qi_plan qi_plan; /* opaque */
typedef void* qi_stream;        /* hipStream_t */

typedef enum { QI_F32 = 0, QI_F64 = 1 } qi_dtype;

typedef enum {
  QI_OK = 0,
  QI_ERR_ARG = -1,     /* bad argument (null pointer, size, dtype, band count)        */
  QI_ERR_STATE = -2,   /* plan is missing the table this call needs                   */
  QI_ERR_HIP = -3,     /* a HIP runtime call failed                                   */
  QI_ERR_FFT = -4,     /* hipFFT failed                                               */
  QI_ERR_NOMEM = -5,   /* workspace budget too small for one (channel, band) tile     */
  QI_ERR_UNSUPPORTED = -6
} qi_status;

/* Gabor-bank slots of a plan */
typedef enum {
  QI_BANK_STYX = 0,  /* styx_cwt.py:147-198  zero-padded linear correlation, L = 2n      */
  QI_BANK_ATOMS = 1  /* cwt_atoms.py:406-421 circular correlation of length n, roll n/2  */
} qi_bank;

/* FFT engine behind a plan */
typedef enum {
  QI_ENGINE_AUTO = 0,   /* native kernels when n is a supported power of two, else hipFFT */
  QI_ENGINE_HIPFFT = 1, /* batched hipFFT + hand-written multiply / epilogue kernels      */
  QI_ENGINE_NATIVE = 2  /* hand-written LDS FFT passes with fused multiply and epilogue   */
} qi_engine;

typedef struct {
  int64_t n;               /* samples per record (any n >= 2 for hipFFT; 2^k for native)  */
  int32_t dtype;           /* qi_dtype                                                     */
  int32_t device;          /* HIP device ordinal                                           */
  int32_t engine;          /* qi_engine                                                    */
  int32_t flags;           /* reserved: must be 0 (qi_plan_create rejects anything else)   */
  int64_t workspace_bytes; /* scratch budget owned by the plan; 0 = default (2 GiB)        */
} qi_plan_desc;

/* What a transform should produce.  Every pointer is optional (NULL = not produced).
 * P = power_scale * |z|^2 is the power the reductions are taken over (the reference's
 * callers use 2|z|^2, docs/examples_tutorial/e00_intro_set/s04_tone_tfr.py:92). */
typedef struct {
  void* coef;         /* [C][B][n] complex  TFR coefficients                                      */
  void* bits;         /* [C][B][n] real     log2(|z| + eps)        utilities/rescaling.py:13-20   */
  void* power_band;   /* [C][B]    float64  sum over time of P     tfr_info.py:93 (before log2)   */
  void* power_time;   /* [C][n]    real     sum over bands of P    tfr_info.py:91 (before log2)   */
  void* stats;        /* [C][4]    float64  {max P, sum P, sum P*log2(P), 0}  tfr_info.py:65-79,231 */
  double power_scale; /* 0 is read as 1                                                           */
  double eps;         /* epsilon inside the log2 of `bits`; 0 is read as 2^-52 (scales_dyadic.py:16) */
} qi_tfr_out;

/* ---- library ------------------------------------------------------------------------ */
int qi_abi_version(void);
const char* qi_last_error(void);
/* Name of the code object's GPU target ("gfx950") and of the device actually present. */
int qi_device_info(int device, char* name, size_t name_len, int64_t* hbm_bytes, int32_t* compute_units);

/* ---- plans -------------------------------------------------------------------------- */
int qi_plan_create(qi_plan** plan, const qi_plan_desc* desc);
int qi_plan_destroy(qi_plan* plan);

/* Gabor atom bank, built on the device in float64 exactly as the reference builds it in
 * the time domain and transformed once per plan:
 *   atom_j[k] = amp_j * exp(-(p_re_j + i p_im_j) x_k^2) * exp(i omega_j x_k),  x_k = k - (n-1)/2
 * styx_cwt.py:68-144 (p_re = 1/(2 s^2), p_im = 0, omega = 2 pi f / fs, amp by dictionary_type);
 * cwt_atoms.py:16-50,202-238 (p = (1 - i shift gamma/pi)/(2 s^2), omega = M_q / s).
 * Host arrays of length B (float64). */
int qi_plan_set_gabor_bank(qi_plan* plan, int bank, int32_t n_bands, const double* p_re, const double* p_im,
                           const double* omega, const double* amp, qi_stream stream);

/* The same atoms in the time domain, float64, for inspection (styx_cwt.wavelet_centered_4cwt,
 * styx_cwt.py:113-144; cwt_atoms.chirp_centered_4cwt, cwt_atoms.py:303-340): out [B][n] complex128
 * (device).  Parameter arrays are host float64 of length B, as above. */
int qi_gabor_atoms(int device, int64_t n, int32_t n_bands, const double* p_re, const double* p_im,
                   const double* omega, const double* amp, void* out, qi_stream stream);

/* The same on the caller's own sample positions x [n] (device, float64; x_k = fs * (t_k - t_offset) in samples):
 * styx_cwt.wavelet_complex on an arbitrary time axis (styx_cwt.py:58-110), cwt_atoms.chirp_complex (cwt_atoms.py:16-50). */
int qi_gabor_atoms_at(int device, int64_t n, int32_t n_bands, const double* p_re, const double* p_im,
                      const double* omega, const double* amp, const void* x, void* out, qi_stream stream);

/* Stockwell band table: shift index idx_j = argmin_k |fftfreq_k - f_j| and Gaussian width
 * sigma_j = M / (2 pi f_j / fs); the window exp(-sigma_j^2 omega_k^2 / 2), omega_k = 2 pi fftfreq_k / fs,
 * is regenerated in registers.  styx_stx.py:216-234.  Host arrays of length B. */
int qi_plan_set_stx_bands(qi_plan* plan, int32_t n_bands, const int64_t* shift_index, const double* sigma);

int64_t qi_plan_bands(const qi_plan* plan, int which /* qi_bank, or 2 for the STX table */);
/* Bands of table `which` whose coefficients are produced by the kernels of profiling stage `stage` (qi_stage below:
 * PASS2 or BLOCK on the native engine, INVERSE on the hipFFT engine); used to price a stage's algorithmic bytes. */
int64_t qi_plan_stage_bands(const qi_plan* plan, int which, int stage);

/* ---- measurement ------------------------------------------------------------------- */
/* Stages of one transform call, timed with HIP events on the caller's stream when profiling is on
 * (bench.py's roofline leg; off by default, two event records per stage launch when on). */
typedef enum {
  QI_STAGE_FORWARD = 0,  /* pack + forward FFT of the records                                  */
  QI_STAGE_MULTIPLY = 1, /* spectrum x atom bank, or shifted spectrum x Gaussian (hipFFT engine) */
  QI_STAGE_INVERSE = 2,  /* batched inverse FFT (hipFFT engine)                                  */
  QI_STAGE_EPILOGUE = 3, /* crop / power / entropy epilogue (hipFFT engine)                      */
  QI_STAGE_PASS1 = 4,    /* native engine: fused multiply + first FFT pass                       */
  QI_STAGE_PASS2 = 5,    /* native engine: second FFT pass + fused epilogue                      */
  QI_STAGE_BLOCK = 6,    /* native engine: short-atom bands by overlap-save blocks (forward, filter, inverse,  */
                         /* epilogue in one kernel)                                                             */
  QI_STAGE_ZOOM = 7,     /* native engine: narrow-band panels, interpolation kernel (one span per launch)       */
  QI_STAGE_ZOOM_COARSE = 8, /* native engine: baseband gather + batched coarse inverse FFT of the zoom bands     */
  QI_STAGE_COUNT = 9
} qi_stage;
/* enable: 0 off; low 16 bits: 1 every stage, otherwise a mask with bit (stage + 1) set for each stage to time (every
 * recorded event is a small bubble in the stream, so a caller that wants one stage asks for that one); high 16 bits:
 * sampling period P (0 or 1: every transform call; P: every P-th call of qi_cwt / qi_stx is timed, the others run
 * without events).  Enabling or disabling also clears the counters. */
int qi_plan_profile(qi_plan* plan, int enable);
/* Sum of elapsed milliseconds and number of launches per stage since the last read; waits for the
 * recorded events.  Arrays of QI_STAGE_COUNT entries. */
int qi_plan_profile_read(qi_plan* plan, double* stage_ms, int64_t* stage_launches, int32_t n_stages);

/* ---- transforms --------------------------------------------------------------------- */
/* styx_cwt.cwt_complex_any_scale_pow2 (styx_cwt.py:147-198, cwt_type="fft") when bank = QI_BANK_STYX;
 * cwt_atoms.cwt_chirp_complex (cwt_atoms.py:343-444, cwt_type="fft") when bank = QI_BANK_ATOMS.
 * sig: [C][n] real. */
int qi_cwt(qi_plan* plan, int bank, const void* sig, int64_t n_channels, const qi_tfr_out* out, qi_stream stream);

/* styx_stx.stx_complex_any_scale_pow2 (styx_stx.py:195-236).  sig: [C][n] real. */
int qi_stx(qi_plan* plan, const void* sig, int64_t n_channels, const qi_tfr_out* out, qi_stream stream);

/* Both of the above on the same records in one call (bank = QI_BANK_STYX): the results are those of qi_cwt followed by
 * qi_stx to within float rounding -- the Stockwell bands may be formed from the even bins of the zero-padded spectrum
 * the CWT has just made instead of a second forward transform, and when the records fit one workspace tile the two
 * transforms share their kernel launches stage by stage (one forward transform per 4096-sample block for the bands of
 * both), so out_cwt is complete only when the call's work on `stream` is (the tutorials run both on every record,
 * s04_tone_tfr.py:84-112). */
int qi_cwt_stx(qi_plan* plan, int bank, const void* sig, int64_t n_channels, const qi_tfr_out* out_cwt,
               const qi_tfr_out* out_stx, qi_stream stream);

/* styx_fft.stft_complex_pow2 / stft_from_sig (styx_fft.py:14-57,152-187): scipy.signal.stft with
 * boundary="zeros", padded=True, detrend="constant", one-sided.  window: [seg] real (device);
 * scale multiplies every coefficient (1/sum(window), times 2 sqrt(pi)/seg for stft_from_sig).
 * Z: [C][nfft/2+1][n_seg] complex, bits: same shape real or NULL.  n_seg as qi_stft_segments().
 * scratch: caller-owned device buffer of qi_stft_scratch_bytes() (windowed frames + their spectra). */
int64_t qi_stft_segments(int64_t n, int64_t seg, int64_t hop);
int64_t qi_stft_scratch_bytes(int dtype, int64_t n_channels, int64_t n, int64_t seg, int64_t hop, int64_t nfft);
int qi_stft(int dtype, int device, const void* sig, int64_t n_channels, int64_t n, const void* window, int64_t seg,
            int64_t hop, int64_t nfft, double scale, void* Z, void* bits, double eps, void* scratch,
            int64_t scratch_bytes, qi_stream stream);

/* styx_fft.welch_power_pow2 (styx_fft.py:230-266): scipy.signal.welch, detrend="constant", scaling="spectrum",
 * average="mean", one-sided, no boundary extension.  Pxx: [C][nfft/2+1] real.  scale = 1/sum(window). */
int64_t qi_welch_scratch_bytes(int dtype, int64_t n_channels, int64_t n, int64_t seg, int64_t hop, int64_t nfft);
int qi_welch(int dtype, int device, const void* sig, int64_t n_channels, int64_t n, const void* window, int64_t seg,
             int64_t hop, int64_t nfft, double scale, void* pxx, void* scratch, int64_t scratch_bytes,
             qi_stream stream);

/* ---- tfr_info reductions on a caller-supplied power panel P [C][B][n] (real) ---------- */
/* Marginals in one pass: power_band [C][B] f64, power_time [C][n] real, stats [C][4] f64
 * {max, sum, sum P log2 P, 0}.  tfr_info.py:82-94 (the sums / max under the log2). */
int qi_power_marginals(int dtype, int device, const void* power, int64_t n_channels, int64_t n_bands, int64_t n,
                       void* power_band, void* power_time, void* stats, void* scratch, int64_t scratch_bytes,
                       qi_stream stream);
int64_t qi_power_marginals_scratch_bytes(int64_t n_channels, int64_t n_bands, int64_t n);

/* out[i] = log2(in[i] + eps) - ref[c]   (tfr_info.py:65-79 with ref = log2(max + eps));
 * `ref` is a device array [C] float64 or NULL (= 0).  Elements per channel = count. */
int qi_log2_offset(int dtype, int device, const void* in, void* out, int64_t n_channels, int64_t count, double eps,
                   const void* ref, qi_stream stream);

/* out[i] = (double) in[i] for `count` float32 values (a complex64 panel is 2 * count floats): the reference returns
 * complex128 panels / float64 bits whatever the record's dtype (styx_cwt.py:195-198, styx_stx.py:228, cwt_atoms.py:408,442);
 * the wrappers compute float32 records in float32 and widen on the device before the copy to the host. */
int qi_widen(int device, const void* in, void* out, int64_t count, qi_stream stream);

/* out[i] = log2(|in[i]| + eps) for `count` real (is_complex = 0) or complex (1; dtype is the real type) values:
 * utilities.rescaling.to_log2_with_epsilon (utilities/rescaling.py:13-20) on a device array. */
int qi_log2_abs(int dtype, int device, const void* in, int is_complex, void* out, int64_t count, double eps,
                qi_stream stream);

/* ShannonStft family (tfr_info.py:203-260) on P [C][B][n]:
 *   pdf = P * mult, mult = 1/sum(P)            (mode 0, shannon_stft_from_tfr_power)
 *                        = 1/sum_axis0 + eps64  (mode 1, ShannonStftPerTime,  deg_free = B)
 *                        = 1/sum_axis1 + eps64  (mode 2, ShannonStftPerFreq,  deg_free = n)
 *   info = -log2(pdf + eps64); shannon_bits = pdf*info; isnr = log2(D) - info; esnr = shannon_bits/(log2(D)/D)
 * mult: device [C] (mode 0), [C][n] (mode 1) or [C][B] (mode 2), real.  Outputs optional. */
int qi_shannon_panel(int dtype, int device, const void* power, const void* mult, int mode, int64_t n_channels,
                     int64_t n_bands, int64_t n, double deg_free, void* info, void* shannon_bits, void* isnr,
                     void* esnr, qi_stream stream);

/* ---- sliding-window STFT in scipy.signal.ShortTimeFFT's convention (utilities/short_time_fft.py:20-175) ------------
 * Slice q (0 <= q < n_slices) covers the padded record from sample first + q * hop (first <= 0: ShortTimeFFT's
 * p_min * hop - seg // 2); pad_mode 0 zeros, 1 edge, 2 even, 3 odd reflection; detrend 1 removes each slice's mean
 * (stft_detrend(detr="constant")); the windowed slice is rotated left by `roll` before the transform (ShortTimeFFT's
 * phase_shift = 0 convention: roll = seg // 2).  window: [seg] real, already scaled (ShortTimeFFT.scale_to).  Z: [C][nfft/2+1]
 * [n_slices] complex or NULL; real_out: same shape real or NULL, real_kind 1 = |Z| (stft_tukey), 2 = |Z|^2
 * (spectrogram_tukey). */
int64_t qi_sliding_scratch_bytes(int dtype, int64_t n_channels, int64_t nfft, int64_t n_slices);
int qi_sliding_stft(int dtype, int device, const void* sig, int64_t n_channels, int64_t n, const void* window,
                    int64_t seg, int64_t hop, int64_t nfft, int64_t first, int64_t n_slices, int pad_mode, int detrend,
                    int64_t roll, void* Z, void* real_out, int real_kind, void* scratch, int64_t scratch_bytes,
                    qi_stream stream);
/* ShortTimeFFT.istft (istft_tukey, short_time_fft.py:112-137): out[c][k - k0] for k0 <= k < k1 = sum over the slices
 * covering k of irfft(S[:, q]) (rotated back by `roll`) [k - (first + q hop)] * dual_window[k - (first + q hop)].
 * S: [C][nfft/2+1][n_slices]. */
int qi_sliding_istft(int dtype, int device, const void* S, int64_t n_channels, const void* dual_window, int64_t seg,
                     int64_t hop, int64_t nfft, int64_t first, int64_t n_slices, int64_t roll, int64_t k0, int64_t k1,
                     void* out, void* scratch, int64_t scratch_bytes, qi_stream stream);

/* ---- 1-D Shannon information of a record and of its spectrum (tfr_info.py:97-200) --------------------------------
 * Shannon / get_info_and_entropy_32 (tfr_info.py:97-133) on marginals [C][n]: info = -log2(m + eps32),
 * entropy = m * info, isnr = log2(n) - info, esnr = entropy / (log2(n) / n).  Any output may be NULL. */
int qi_shannon_1d(int dtype, int device, const void* marginal, int64_t n_channels, int64_t n, void* info, void* entropy,
                  void* isnr, void* esnr, qi_stream stream);
/* scratch for the two calls below */
int64_t qi_shannon_scratch_bytes(int dtype, int64_t n_channels, int64_t n);
/* ShannonTDR (tfr_info.py:138-147): sig_norm = sig / sqrt(sum sig^2) (may be NULL), marginal = sig_norm^2.  [C][n]. */
int qi_shannon_tdr(int dtype, int device, const void* sig, int64_t n_channels, int64_t n, void* sig_norm, void* marginal,
                   void* scratch, int64_t scratch_bytes, qi_stream stream);
/* ShannonFFT (tfr_info.py:163-183): spectrum = rfft(sig) [C][n/2+1] complex, angle = np.unwrap(np.angle(spectrum))
 * (may be NULL), marginal = |spectrum|^2 / sum |spectrum|^2. */
int qi_shannon_fft(int dtype, int device, const void* sig, int64_t n_channels, int64_t n, void* spectrum, void* angle,
                   void* marginal, void* scratch, int64_t scratch_bytes, qi_stream stream);

#ifdef __cplusplus
}
#endif
#endif /* QI_TFR_H */
