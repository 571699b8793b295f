// 1-D Shannon information of a record and of its spectrum (tfr_info.py:97-200: get_info_and_entropy_32, Shannon,
// ShannonTDR, ShannonFFT).  Streaming kernels over [C][n] rows; the two sums (sum sig^2, sum |X|^2) are reduced in
// fixed order (per-workgroup partials in float64, summed in index order by every consumer); np.unwrap is a
// workgroup-wide prefix sum of the 2 pi corrections, one workgroup per record.
#include "qi_common.hpp"
#include "qi_device.hpp"
#include "qi_fft_reg.hpp"  // QI_LAUNCH_CHECK

namespace qi {

namespace {

constexpr int kRedSpan = 4096;  // elements one partial covers

// partial[c][b] = sum over the b-th span of |x|^2 (x real, or complex when CPLX)
template <typename T, bool CPLX>
__global__ void __launch_bounds__(256) k_sumsq_partials(const T* __restrict__ x, int64_t n, double* __restrict__ partial,
                                                        int64_t nspan) {
  __shared__ double s[256 / kWave];
  const int tid = threadIdx.x, lane = tid & (kWave - 1), wv = tid / kWave;
  const int64_t c = blockIdx.y, b = blockIdx.x;
  const T* row = x + c * n * (CPLX ? 2 : 1);
  double acc = 0.0;
  for (int64_t i = b * kRedSpan + tid; i < (b + 1) * kRedSpan && i < n; i += 256) {
    if (CPLX) {
      const double re = (double)row[2 * i], im = (double)row[2 * i + 1];
      acc += re * re + im * im;
    } else {
      const double v = (double)row[i];
      acc += v * v;
    }
  }
  acc = wave_sum(acc);
  if (lane == 0) s[wv] = acc;
  __syncthreads();
  if (tid == 0) {
    double t = 0.0;
    for (int w = 0; w < 256 / kWave; ++w) t += s[w];
    partial[c * nspan + b] = t;
  }
}

__device__ __forceinline__ double total_of(const double* __restrict__ partial, int64_t nspan) {
  double t = 0.0;
  for (int64_t b = 0; b < nspan; ++b) t += partial[b];  // same order in every thread: identical totals
  return t;
}

// ShannonTDR (tfr_info.py:146-147): sig_norm = sig / sqrt(sum sig^2), marginal = sig_norm^2
template <typename T>
__global__ void __launch_bounds__(256) k_tdr_marginal(const T* __restrict__ sig, int64_t n,
                                                      const double* __restrict__ partial, int64_t nspan,
                                                      T* __restrict__ sig_norm, T* __restrict__ marginal) {
  const int64_t c = blockIdx.y;
  const T inv = (T)(1.0 / sqrt(total_of(partial + c * nspan, nspan)));
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const T v = sig[c * n + i] * inv;
    if (sig_norm) sig_norm[c * n + i] = v;
    marginal[c * n + i] = v * v;
  }
}

// ShannonFFT (tfr_info.py:179-183): angle = arg(X), marginal = |X|^2 / sum |X|^2
template <typename T>
__global__ void __launch_bounds__(256) k_fft_marginal(const cplx<T>* __restrict__ X, int64_t nf,
                                                      const double* __restrict__ partial, int64_t nspan,
                                                      T* __restrict__ angle, T* __restrict__ marginal) {
  const int64_t c = blockIdx.y;
  const T inv = (T)(1.0 / total_of(partial + c * nspan, nspan));
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nf; i += (int64_t)gridDim.x * 256) {
    const cplx<T> z = X[c * nf + i];
    if (angle) angle[c * nf + i] = (T)atan2((double)z.y, (double)z.x);
    marginal[c * nf + i] = (z.x * z.x + z.y * z.y) * inv;
  }
}

// np.unwrap(p) (period 2 pi): up[0] = p[0], up[i] = p[i] + cumsum(corr)[i], corr[i] = ddmod - dd for |dd| >= pi with
// dd = p[i] - p[i-1], ddmod = mod(dd + pi, 2 pi) - pi (and +pi instead of -pi when dd > 0).  The corrections are whole
// periods, so they are counted as integers: turns[i] (first kernel, reads only the wrapped angles), then a workgroup-wide
// prefix sum per record and one rewrite pass (second kernel) -- exact, unlike a floating-point cumulative sum.
template <typename T>
__global__ void __launch_bounds__(256) k_unwrap_turns(const T* __restrict__ p, int64_t nf, int32_t* __restrict__ turns) {
  const int64_t c = blockIdx.y;
  const double kPi = 3.14159265358979323846, kTwoPi = 2.0 * kPi;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nf; i += (int64_t)gridDim.x * 256) {
    int32_t t = 0;
    if (i > 0) {
      const double dd = (double)p[c * nf + i] - (double)p[c * nf + i - 1];
      if (fabs(dd) >= kPi) {
        double ddmod = fmod(dd + kPi, kTwoPi);
        if (ddmod < 0.0) ddmod += kTwoPi;
        ddmod -= kPi;
        if (ddmod == -kPi && dd > 0.0) ddmod = kPi;
        t = (int32_t)llrint((ddmod - dd) / kTwoPi);
      }
    }
    turns[c * nf + i] = t;
  }
}

template <typename T>
__global__ void __launch_bounds__(1024) k_unwrap_apply(T* __restrict__ p, int64_t nf, const int32_t* __restrict__ turns) {
  __shared__ int s_wave[1024 / kWave];
  __shared__ int s_carry;
  const int tid = threadIdx.x, lane = tid & (kWave - 1), wv = tid / kWave;
  const int64_t c = blockIdx.x;
  const double kTwoPi = 2.0 * 3.14159265358979323846;
  if (tid == 0) s_carry = 0;
  __syncthreads();
  for (int64_t base = 0; base < nf; base += 1024) {
    const int64_t i = base + tid;
    int v = i < nf ? turns[c * nf + i] : 0;
    // inclusive scan inside the wave, then across the 16 waves
#pragma unroll
    for (int o = 1; o < kWave; o <<= 1) {
      const int u = __shfl_up(v, o, kWave);
      if (lane >= o) v += u;
    }
    if (lane == kWave - 1) s_wave[wv] = v;
    __syncthreads();
    int before = s_carry;
    for (int w = 0; w < wv; ++w) before += s_wave[w];
    const int total = before + v;
    if (i < nf) p[c * nf + i] = (T)((double)p[c * nf + i] + (double)total * kTwoPi);
    __syncthreads();
    if (tid == 1023) s_carry = total;
    __syncthreads();
  }
}

// Shannon / get_info_and_entropy_32 (tfr_info.py:97-133) on [C][n] marginals: info = -log2(m + eps32),
// entropy = m info, isnr = log2(n) - info, esnr = entropy / (log2(n) / n)
template <typename T>
__global__ void __launch_bounds__(256) k_shannon_1d(const T* __restrict__ m, int64_t total, int64_t n, T* __restrict__ info,
                                                    T* __restrict__ entropy, T* __restrict__ isnr, T* __restrict__ esnr) {
  const T eps32 = (T)1.1920928955078125e-07;
  const T log2n = (T)log2((double)n);
  const T inv_ref = (T)((double)n / log2((double)n));
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const T v = m[i];
    const T inf = -log2_t(v + eps32);
    const T ent = v * inf;
    if (info) info[i] = inf;
    if (entropy) entropy[i] = ent;
    if (isnr) isnr[i] = log2n - inf;
    if (esnr) esnr[i] = ent * inv_ref;
  }
}

inline unsigned grid_for(int64_t count) {
  const int64_t g = ceil_div(count, 256);
  return (unsigned)(g > 4096 ? 4096 : (g < 1 ? 1 : g));
}

}  // namespace

int64_t shannon_spans(int64_t n) { return ceil_div(n, kRedSpan); }

template <typename T>
int launch_shannon_1d(const T* m, int64_t C, int64_t n, T* info, T* entropy, T* isnr, T* esnr, hipStream_t st) {
  k_shannon_1d<T><<<grid_for(C * n), 256, 0, st>>>(m, C * n, n, info, entropy, isnr, esnr);
  QI_LAUNCH_CHECK();
  return QI_OK;
}

template <typename T>
int launch_tdr_marginal(const T* sig, int64_t C, int64_t n, T* sig_norm, T* marginal, double* partial, hipStream_t st) {
  const int64_t nspan = shannon_spans(n);
  k_sumsq_partials<T, false><<<dim3((unsigned)nspan, (unsigned)C), 256, 0, st>>>(sig, n, partial, nspan);
  QI_LAUNCH_CHECK();
  k_tdr_marginal<T><<<dim3(grid_for(n), (unsigned)C), 256, 0, st>>>(sig, n, partial, nspan, sig_norm, marginal);
  QI_LAUNCH_CHECK();
  return QI_OK;
}

template <typename T>
int launch_fft_marginal(const cplx<T>* X, int64_t C, int64_t nf, T* angle, T* marginal, double* partial, int32_t* turns,
                        hipStream_t st) {
  const int64_t nspan = shannon_spans(nf);
  k_sumsq_partials<T, true><<<dim3((unsigned)nspan, (unsigned)C), 256, 0, st>>>(reinterpret_cast<const T*>(X), nf,
                                                                                 partial, nspan);
  QI_LAUNCH_CHECK();
  k_fft_marginal<T><<<dim3(grid_for(nf), (unsigned)C), 256, 0, st>>>(X, nf, partial, nspan, angle, marginal);
  QI_LAUNCH_CHECK();
  if (angle) {
    k_unwrap_turns<T><<<dim3(grid_for(nf), (unsigned)C), 256, 0, st>>>(angle, nf, turns);
    QI_LAUNCH_CHECK();
    k_unwrap_apply<T><<<(unsigned)C, 1024, 0, st>>>(angle, nf, turns);
    QI_LAUNCH_CHECK();
  }
  return QI_OK;
}

#define QI_INSTANTIATE_S1D(T)                                                                               \
  template int launch_shannon_1d<T>(const T*, int64_t, int64_t, T*, T*, T*, T*, hipStream_t);               \
  template int launch_tdr_marginal<T>(const T*, int64_t, int64_t, T*, T*, double*, hipStream_t);            \
  template int launch_fft_marginal<T>(const cplx<T>*, int64_t, int64_t, T*, T*, double*, int32_t*, hipStream_t);
QI_INSTANTIATE_S1D(float)
QI_INSTANTIATE_S1D(double)

}  // namespace qi
