// Hand-written gfx950 kernels around the FFT passes: packing, Gabor bank construction,
// spectrum multiply, Stockwell shift x Gaussian, crop / power / entropy epilogue, STFT framing,
// tfr_info reductions.  Wave = 64 lanes everywhere.
#include <map>
#include <mutex>
#include <utility>
#include "qi_common.hpp"
#include "qi_device.hpp"
#include "qi_finalize.hpp"

namespace qi {

namespace {

// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ void k_pack_pad(const T* __restrict__ sig, cplx<T>* __restrict__ X, int64_t n, int64_t L) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t c = blockIdx.y;
  if (i < L) X[c * L + i] = mk<T>(i < n ? sig[c * n + i] : T(0), T(0));
}

// Time-domain rows of the bank in float64.  Linear mode stores conj(atom) reversed and zero-padded
// (the operand scipy.signal.fftconvolve receives, styx_cwt.py:195-196); circular mode the atom itself.
__global__ void k_bank_rows(double2* __restrict__ rows, int64_t n, int64_t L, int circular,
                            const double* __restrict__ p_re, const double* __restrict__ p_im,
                            const double* __restrict__ omega, const double* __restrict__ amp, int j0, double taper_e,
                            const double* __restrict__ xs) {
  int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int jj = blockIdx.y;
  int j = j0 + jj;
  if (m >= L) return;
  double2 v = make_double2(0.0, 0.0);
  if (m < n) {
    int64_t k = circular ? m : (n - 1 - m);
    double x = xs ? xs[k] : (double)k - 0.5 * (double)(n - 1);  // xs: the caller's own sample positions (atoms for inspection)
    double g = amp[j] * exp(-p_re[j] * x * x);
    if (taper_e > 0.0) g *= split_taper(x, n, taper_e);  // the zoom-engine part of a split band
    double ph = omega[j] * x - p_im[j] * x * x;
    double s, c;
    sincos(ph, &s, &c);
    v.x = g * c;
    v.y = circular ? g * s : -g * s;
  }
  rows[(int64_t)jj * L + m] = v;
}

template <typename T>
__global__ void k_bank_convert(const double2* __restrict__ F, cplx<T>* __restrict__ bank, int64_t count, int conj,
                               double scale) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  double2 v = F[i];
  bank[i] = mk<T>((T)(v.x * scale), (T)((conj ? -v.y : v.y) * scale));
}

// Y[c][j][k] = X[c][k] * H[j][k]   (1/L is folded into H)
template <typename T>
__global__ void k_mul_bank(const cplx<T>* __restrict__ X, const cplx<T>* __restrict__ H, cplx<T>* __restrict__ Y,
                           int64_t Bt, int64_t L) {
  int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t j = blockIdx.y, c = blockIdx.z;
  if (k >= L) return;
  Y[(c * Bt + j) * L + k] = cmul(X[c * L + k], H[j * L + k]);
}

// Stockwell: Y[c][j][k] = X[c][(k + idx_j) mod n] * exp2(-(coef_j * ks)^2) / n, ks = signed bin of k.
// coef_j = sigma_j * 2 pi / n * sqrt(log2(e) / 2) (host, float64).  styx_stx.py:213-234.
template <typename T>
__global__ void k_stx_window(const cplx<T>* __restrict__ X, cplx<T>* __restrict__ Y, int64_t Bt, int64_t n,
                             const int64_t* __restrict__ idx, const double* __restrict__ coef) {
  int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t j = blockIdx.y, c = blockIdx.z;
  if (k >= n) return;
  int64_t ks = (k <= (n - 1) / 2) ? k : k - n;
  T a = (T)coef[j] * (T)ks;
  T w = exp2_t(-a * a) * (T(1) / (T)n);
  int64_t src = k + idx[j];
  if (src >= n) src -= n;
  cplx<T> x = X[c * n + src];
  Y[(c * Bt + j) * n + k] = mk<T>(x.x * w, x.y * w);
}

// ------------------------------------------------------------------------------------------------
// Epilogue: crop / roll one tile of inverse-transformed rows into the caller's panel and take every
// reduction tfr_info needs in the same pass.  One workgroup owns kEpiSpan consecutive time samples
// of one channel and walks the tile's bands, so the per-time sums stay in registers.
template <typename T, bool REAL_IN>
__global__ void __launch_bounds__(kEpiThreads) k_epilogue(EpiArgs<T> a) {
  __shared__ double s_red[2][kEpiThreads / kWave];
  __shared__ double s_fin[3][kEpiThreads / kWave];
  const int tid = threadIdx.x, lane = tid & (kWave - 1), wv = tid / kWave;
  const int64_t blk = blockIdx.x, c = blockIdx.y, nblk = gridDim.x;
  const int64_t t0 = blk * kEpiSpan + (int64_t)tid * kEpiVec;
  T col[kEpiVec];
#pragma unroll
  for (int v = 0; v < kEpiVec; ++v) col[v] = T(0);
  T mx = T(0);
  double plogp = 0.0;

  for (int64_t j = 0; j < a.Bt; ++j) {
    T rowacc = T(0);
    const int64_t row = (c * a.Bt + j) * a.L;
    const int64_t orow = (c * a.B + a.j0 + j) * a.n;
#pragma unroll
    for (int v = 0; v < kEpiVec; ++v) {
      int64_t t = t0 + v;
      if (t < a.n) {
        T p;
        if constexpr (REAL_IN) {
          p = reinterpret_cast<const T*>(a.Y)[row + t];
        } else {
          int64_t src = t + a.off;
          if (src >= a.L) src -= a.L;
          cplx<T> z = a.Y[row + src];
          if (a.coef) a.coef[orow + t] = z;
          T m2 = z.x * z.x + z.y * z.y;
          if (a.bits) a.bits[orow + t] = log2_t(sqrt_t(m2) + a.eps);
          p = a.power_scale * m2;
        }
        col[v] += p;
        rowacc += p;
        mx = p > mx ? p : mx;
        if (p > T(0)) plogp += (double)(p * log2_t(p));
      }
    }
    if (a.part_band) {
      double r = wave_sum((double)rowacc);
      if (lane == 0) s_red[j & 1][wv] = r;
      __syncthreads();
      if (tid == 0) {
        double s = 0.0;
#pragma unroll
        for (int w = 0; w < kEpiThreads / kWave; ++w) s += s_red[j & 1][w];
        a.part_band[(c * a.B + a.j0 + j) * nblk + blk] = s;
      }
    }
  }
  T tot = T(0);
#pragma unroll
  for (int v = 0; v < kEpiVec; ++v) {
    int64_t t = t0 + v;
    tot += col[v];
    if (a.power_time && t < a.n) {
      if (a.tile_b == 0)
        a.power_time[c * a.n + t] = col[v];
      else
        a.power_time[c * a.n + t] += col[v];
    }
  }
  if (a.part_stat) {
    double r0 = wave_max((double)mx), r1 = wave_sum((double)tot), r2 = wave_sum(plogp);
    if (lane == 0) {
      s_fin[0][wv] = r0;
      s_fin[1][wv] = r1;
      s_fin[2][wv] = r2;
    }
    __syncthreads();
    if (tid == 0) {
      double m = 0.0, s1 = 0.0, s2 = 0.0;
#pragma unroll
      for (int w = 0; w < kEpiThreads / kWave; ++w) {
        m = s_fin[0][w] > m ? s_fin[0][w] : m;
        s1 += s_fin[1][w];
        s2 += s_fin[2][w];
      }
      double* o = a.part_stat + ((c * a.ntile_b + a.tile_b) * nblk + blk) * 3;
      o[0] = m;
      o[1] = s1;
      o[2] = s2;
    }
  }
}

// One workgroup per (band or the stats slot, channel): fixed-order sums of the partials (finalize_block).
__global__ void __launch_bounds__(256) k_finalize(const double* __restrict__ part_band,
                                                  const double* __restrict__ part_stat,
                                                  double* __restrict__ power_band, double* __restrict__ stats,
                                                  int64_t B, int64_t nblk, int64_t nstat,
                                                  const int32_t* __restrict__ band_slots) {
  __shared__ double s[3][256 / kWave];
  finalize_block(part_band, part_stat, power_band, stats, B, nblk, nstat, band_slots, blockIdx.x, blockIdx.y, s);
}

// ------------------------------------------------------------------------------------------------
// STFT framing: zero-extended segment, mean removal, window (scipy.signal._spectral_helper with
// boundary='zeros', padded=True, detrend='constant').  One workgroup per (segment, channel).
template <typename T>
__global__ void __launch_bounds__(256) k_stft_frames(const T* __restrict__ sig, const T* __restrict__ win,
                                                     T* __restrict__ frames, int64_t n, int64_t seg, int64_t hop,
                                                     int64_t nfft, int64_t nseg, int64_t lead) {
  __shared__ double s[256 / kWave];
  __shared__ double s_mean;
  const int tid = threadIdx.x, lane = tid & (kWave - 1), wv = tid / kWave;
  const int64_t m = blockIdx.x, c = blockIdx.y;
  const int64_t base = m * hop - lead;  // lead = seg/2 zero-extended samples (stft) or 0 (welch)
  const T* x = sig + c * n;
  double acc = 0.0;
  for (int64_t i = tid; i < seg; i += 256) {
    int64_t k = base + i;
    if (k >= 0 && k < n) acc += (double)x[k];
  }
  acc = wave_sum(acc);
  if (lane == 0) s[wv] = acc;
  __syncthreads();
  if (tid == 0) {
    double t = 0.0;
    for (int w = 0; w < 256 / kWave; ++w) t += s[w];
    s_mean = t / (double)seg;
  }
  __syncthreads();
  const T mean = (T)s_mean;
  T* f = frames + (c * nseg + m) * nfft;
  for (int64_t i = tid; i < nfft; i += 256) {
    T v = T(0);
    if (i < seg) {
      int64_t k = base + i;
      T xv = (k >= 0 && k < n) ? x[k] : T(0);
      v = (xv - mean) * win[i];
    }
    f[i] = v;
  }
}

// F [C*nseg][nf] -> Z [C][nf][nseg] (frequency x time as the reference returns it), scaled; bits optional.
template <typename T>
__global__ void k_stft_transpose(const cplx<T>* __restrict__ F, cplx<T>* __restrict__ Z, T* __restrict__ bits,
                                 int64_t nseg, int64_t nf, T scale, T eps) {
  __shared__ cplx<T> tile[32][33];
  const int64_t c = blockIdx.z;
  const int64_t f0 = (int64_t)blockIdx.x * 32, m0 = (int64_t)blockIdx.y * 32;
  for (int r = threadIdx.y; r < 32; r += blockDim.y) {
    int64_t m = m0 + r, f = f0 + threadIdx.x;
    if (m < nseg && f < nf) tile[r][threadIdx.x] = F[(c * nseg + m) * nf + f];
  }
  __syncthreads();
  for (int r = threadIdx.y; r < 32; r += blockDim.y) {
    int64_t f = f0 + r, m = m0 + threadIdx.x;
    if (m < nseg && f < nf) {
      cplx<T> z = tile[threadIdx.x][r];
      z.x *= scale;
      z.y *= scale;
      Z[(c * nf + f) * nseg + m] = z;
      if (bits) bits[(c * nf + f) * nseg + m] = log2_t(sqrt_t(z.x * z.x + z.y * z.y) + eps);
    }
  }
}

// Welch: Pxx[c][f] = w_f * scale^2 * mean over segments of |F[c][m][f]|^2, w_f = 2 except at DC and (even nfft) Nyquist
// (scipy.signal.welch, scaling="spectrum", average="mean", one-sided; call site styx_fft.py:253-266).
template <typename T>
__global__ void k_welch_mean(const cplx<T>* __restrict__ F, T* __restrict__ pxx, int64_t nseg, int64_t nf, int64_t nfft,
                             T scale2) {
  const int64_t f = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, c = blockIdx.y;
  if (f >= nf) return;
  double acc = 0.0;
  for (int64_t m = 0; m < nseg; ++m) {
    const cplx<T> z = F[(c * nseg + m) * nf + f];
    acc += (double)(z.x * z.x + z.y * z.y);
  }
  const bool paired = f > 0 && !(nfft % 2 == 0 && f == nf - 1);
  pxx[c * nf + f] = (T)(acc / (double)nseg) * scale2 * (paired ? T(2) : T(1));
}

// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ void k_log2_offset(const T* __restrict__ in, T* __restrict__ out, int64_t count, T eps,
                              const double* __restrict__ ref) {
  const int64_t c = blockIdx.y;
  const T r = ref ? (T)ref[c] : T(0);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x)
    out[c * count + i] = log2_t(in[c * count + i] + eps) - r;
}

// out[i] = log2(|in[i]| + eps), in real (CPLX = false) or complex (utilities/rescaling.py:13-20)
template <typename T, bool CPLX>
__global__ void k_log2_abs(const T* __restrict__ in, T* __restrict__ out, int64_t count, T eps) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x) {
    T m;
    if (CPLX) {
      const T re = in[2 * i], im = in[2 * i + 1];
      m = sqrt_t(re * re + im * im);
    } else {
      m = in[i] < T(0) ? -in[i] : in[i];
    }
    out[i] = log2_t(m + eps);
  }
}

template <typename T>
__global__ void k_shannon(const T* __restrict__ P, const T* __restrict__ mult, int mode, int64_t B, int64_t n,
                          T log2d, T inv_ref, T* __restrict__ info, T* __restrict__ sb, T* __restrict__ isnr,
                          T* __restrict__ esnr) {
  const int64_t c = blockIdx.z, j = blockIdx.y;
  const T eps = (T)2.220446049250313e-16;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = (c * B + j) * n + t;
    const T m = mode == 0 ? mult[c] : (mode == 1 ? mult[c * n + t] : mult[c * B + j]);
    const T pdf = P[i] * m;
    const T inf = -log2_t(pdf + eps);
    const T s = pdf * inf;
    if (info) info[i] = inf;
    if (sb) sb[i] = s;
    if (isnr) isnr[i] = log2d - inf;
    if (esnr) esnr[i] = s * inv_ref;
  }
}

inline dim3 grid1(int64_t count, int threads) { return dim3((unsigned)ceil_div(count, threads)); }

}  // namespace

// ------------------------------------------------------------------------------------------------
#define QI_LAUNCH_CHECK()                                                                      \
  do {                                                                                         \
    hipError_t e_ = hipGetLastError();                                                         \
    if (e_ != hipSuccess) {                                                                    \
      set_error("%s:%d kernel launch -> %s", __FILE__, __LINE__, hipGetErrorString(e_));       \
      return QI_ERR_HIP;                                                                       \
    }                                                                                          \
  } while (0)

template <typename T>
int launch_pack_pad(const T* sig, cplx<T>* X, int64_t C, int64_t n, int64_t L, hipStream_t st) {
  dim3 g((unsigned)ceil_div(L, 256), (unsigned)C);
  k_pack_pad<T><<<g, 256, 0, st>>>(sig, X, n, L);
  QI_LAUNCH_CHECK();
  return QI_OK;
}

int launch_bank_rows(double2* rows, int64_t n, int64_t L, int circular, const double* p_re, const double* p_im,
                     const double* omega, const double* amp, int j0, int nb, hipStream_t st, double taper_e, const double* xs) {
  dim3 g((unsigned)ceil_div(L, 256), (unsigned)nb);
  k_bank_rows<<<g, 256, 0, st>>>(rows, n, L, circular, p_re, p_im, omega, amp, j0, taper_e, xs);
  QI_LAUNCH_CHECK();
  return QI_OK;
}

template <typename T>
int launch_bank_convert(const double2* F, cplx<T>* bank, int64_t count, int conj, double scale, hipStream_t st) {
  k_bank_convert<T><<<grid1(count, 256), 256, 0, st>>>(F, bank, count, conj, scale);
  QI_LAUNCH_CHECK();
  return QI_OK;
}

template <typename T>
int launch_mul_bank(const cplx<T>* X, const cplx<T>* H, cplx<T>* Y, int64_t Ct, int64_t Bt, int64_t L,
                    hipStream_t st) {
  dim3 g((unsigned)ceil_div(L, 256), (unsigned)Bt, (unsigned)Ct);
  k_mul_bank<T><<<g, 256, 0, st>>>(X, H, Y, Bt, L);
  QI_LAUNCH_CHECK();
  return QI_OK;
}

template <typename T>
int launch_stx_window(const cplx<T>* X, cplx<T>* Y, int64_t Ct, int64_t Bt, int64_t n, const int64_t* idx,
                      const double* coef, hipStream_t st) {
  dim3 g((unsigned)ceil_div(n, 256), (unsigned)Bt, (unsigned)Ct);
  k_stx_window<T><<<g, 256, 0, st>>>(X, Y, Bt, n, idx, coef);
  QI_LAUNCH_CHECK();
  return QI_OK;
}

template <typename T>
int launch_epilogue(const EpiArgs<T>& a, hipStream_t st) {
  dim3 g((unsigned)ceil_div(a.n, kEpiSpan), (unsigned)a.Ct);
  k_epilogue<T, false><<<g, kEpiThreads, 0, st>>>(a);
  QI_LAUNCH_CHECK();
  return QI_OK;
}

int launch_finalize(const double* part_band, const double* part_stat, double* power_band, double* stats, int64_t C,
                    int64_t B, int64_t nblk, int64_t nstat, hipStream_t st, const int32_t* band_slots) {
  dim3 g((unsigned)(B + 1), (unsigned)C);
  k_finalize<<<g, 256, 0, st>>>(part_band, part_stat, power_band, stats, B, nblk, nstat, band_slots);
  QI_LAUNCH_CHECK();
  return QI_OK;
}

template <typename T>
int launch_stft_frames(const T* sig, const T* win, T* frames, int64_t C, int64_t n, int64_t seg, int64_t hop,
                       int64_t nfft, int64_t nseg, int64_t lead, hipStream_t st) {
  dim3 g((unsigned)nseg, (unsigned)C);
  k_stft_frames<T><<<g, 256, 0, st>>>(sig, win, frames, n, seg, hop, nfft, nseg, lead);
  QI_LAUNCH_CHECK();
  return QI_OK;
}

template <typename T>
int launch_stft_transpose(const cplx<T>* F, cplx<T>* Z, T* bits, int64_t C, int64_t nseg, int64_t nf, T scale, T eps,
                          hipStream_t st) {
  dim3 g((unsigned)ceil_div(nf, 32), (unsigned)ceil_div(nseg, 32), (unsigned)C);
  k_stft_transpose<T><<<g, dim3(32, 8), 0, st>>>(F, Z, bits, nseg, nf, scale, eps);
  QI_LAUNCH_CHECK();
  return QI_OK;
}

template <typename T>
int launch_welch_mean(const cplx<T>* F, T* pxx, int64_t C, int64_t nseg, int64_t nf, int64_t nfft, T scale2,
                      hipStream_t st) {
  dim3 g((unsigned)ceil_div(nf, 256), (unsigned)C);
  k_welch_mean<T><<<g, 256, 0, st>>>(F, pxx, nseg, nf, nfft, scale2);
  QI_LAUNCH_CHECK();
  return QI_OK;
}

template <typename T>
int launch_power_marginals(const T* P, int64_t C, int64_t B, int64_t n, T* power_time, double* part_band,
                           double* part_stat, hipStream_t st) {
  EpiArgs<T> a{};
  a.Y = reinterpret_cast<const cplx<T>*>(P);
  a.L = n;
  a.n = n;
  a.off = 0;
  a.Ct = C;
  a.Bt = B;
  a.B = B;
  a.j0 = 0;
  a.power_time = power_time;
  a.part_band = part_band;
  a.part_stat = part_stat;
  a.tile_b = 0;
  a.ntile_b = 1;
  a.power_scale = T(1);
  a.eps = T(0);
  dim3 g((unsigned)ceil_div(n, kEpiSpan), (unsigned)C);
  k_epilogue<T, true><<<g, kEpiThreads, 0, st>>>(a);
  QI_LAUNCH_CHECK();
  return QI_OK;
}

template <typename T>
int launch_log2_offset(const T* in, T* out, int64_t C, int64_t count, T eps, const double* ref, hipStream_t st) {
  int64_t blocks = ceil_div(count, 256);
  if (blocks > 4096) blocks = 4096;
  dim3 g((unsigned)blocks, (unsigned)C);
  k_log2_offset<T><<<g, 256, 0, st>>>(in, out, count, eps, ref);
  QI_LAUNCH_CHECK();
  return QI_OK;
}

// out[i] = (double)in[i]: float32 results handed to callers that expect the reference's float64 / complex128
__global__ void k_widen(const float* __restrict__ in, double* __restrict__ out, int64_t count) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x)
    out[i] = (double)in[i];
}
int allow_dynamic_lds(const void* fn, size_t bytes) {
  if (bytes <= 48 * 1024) return QI_OK;  // inside every kernel's default limit
  static std::mutex mu;
  static std::map<std::pair<int, const void*>, size_t> raised;  // (device, function) -> limit set so far
  static std::map<int, size_t> device_max;
  int dev = 0;
  QI_HIP(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lock(mu);
  auto dm = device_max.find(dev);
  if (dm == device_max.end()) {
    int v = 0;
    QI_HIP(hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerBlock, dev));
    dm = device_max.emplace(dev, (size_t)v).first;
  }
  if (bytes > dm->second) {
    set_error("a kernel asks for %zu bytes of LDS, device %d offers %zu per workgroup", bytes, dev, dm->second);
    return QI_ERR_UNSUPPORTED;
  }
  size_t& have = raised[{dev, fn}];
  if (bytes > have) {
    QI_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    have = bytes;
  }
  return QI_OK;
}

int launch_widen(const float* in, double* out, int64_t count, hipStream_t st) {
  int64_t blocks = ceil_div(count, 256);
  if (blocks > 16384) blocks = 16384;
  k_widen<<<(unsigned)blocks, 256, 0, st>>>(in, out, count);
  QI_LAUNCH_CHECK();
  return QI_OK;
}

template <typename T>
int launch_log2_abs(const T* in, int is_complex, T* out, int64_t count, T eps, hipStream_t st) {
  int64_t blocks = ceil_div(count, 256);
  if (blocks > 8192) blocks = 8192;
  if (is_complex) k_log2_abs<T, true><<<(unsigned)blocks, 256, 0, st>>>(in, out, count, eps);
  else k_log2_abs<T, false><<<(unsigned)blocks, 256, 0, st>>>(in, out, count, eps);
  QI_LAUNCH_CHECK();
  return QI_OK;
}
template int launch_log2_abs<float>(const float*, int, float*, int64_t, float, hipStream_t);
template int launch_log2_abs<double>(const double*, int, double*, int64_t, double, hipStream_t);

template <typename T>
int launch_shannon(const T* P, const T* mult, int mode, int64_t C, int64_t B, int64_t n, double deg, T* info, T* sb,
                   T* isnr, T* esnr, hipStream_t st) {
  int64_t blocks = ceil_div(n, 256);
  if (blocks > 256) blocks = 256;
  dim3 g((unsigned)blocks, (unsigned)B, (unsigned)C);
  const double l2 = log2(deg);
  k_shannon<T><<<g, 256, 0, st>>>(P, mult, mode, B, n, (T)l2, (T)(deg / l2), info, sb, isnr, esnr);
  QI_LAUNCH_CHECK();
  return QI_OK;
}

#define QI_INSTANTIATE(T)                                                                                          \
  template int launch_pack_pad<T>(const T*, cplx<T>*, int64_t, int64_t, int64_t, hipStream_t);                     \
  template int launch_bank_convert<T>(const double2*, cplx<T>*, int64_t, int, double, hipStream_t);                \
  template int launch_mul_bank<T>(const cplx<T>*, const cplx<T>*, cplx<T>*, int64_t, int64_t, int64_t,             \
                                  hipStream_t);                                                                    \
  template int launch_stx_window<T>(const cplx<T>*, cplx<T>*, int64_t, int64_t, int64_t, const int64_t*,           \
                                    const double*, hipStream_t);                                                   \
  template int launch_epilogue<T>(const EpiArgs<T>&, hipStream_t);                                                 \
  template int launch_stft_frames<T>(const T*, const T*, T*, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, \
                                     int64_t, hipStream_t);                                                        \
  template int launch_welch_mean<T>(const cplx<T>*, T*, int64_t, int64_t, int64_t, int64_t, T, hipStream_t);       \
  template int launch_stft_transpose<T>(const cplx<T>*, cplx<T>*, T*, int64_t, int64_t, int64_t, T, T,             \
                                        hipStream_t);                                                              \
  template int launch_power_marginals<T>(const T*, int64_t, int64_t, int64_t, T*, double*, double*, hipStream_t);  \
  template int launch_log2_offset<T>(const T*, T*, int64_t, int64_t, T, const double*, hipStream_t);               \
  template int launch_shannon<T>(const T*, const T*, int, int64_t, int64_t, int64_t, double, T*, T*, T*, T*,       \
                                 hipStream_t);
QI_INSTANTIATE(float)
QI_INSTANTIATE(double)

}  // namespace qi
