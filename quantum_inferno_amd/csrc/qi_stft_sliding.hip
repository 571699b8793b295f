// Sliding-window STFT in the convention of scipy.signal.ShortTimeFFT (the API of the reference's
// utilities/short_time_fft.py:20-175): slice p is centred on sample p * hop, i.e. it covers the padded record from
// p * hop - seg // 2; the record is extended by zeros, its edge values, or its even / odd reflection; each slice is
// optionally detrended (mean removed), windowed, zero-padded to nfft and transformed (hipFFT, batched over slices).
// The inverse multiplies the inverse transforms of the slices by the dual window and overlap-adds them: every output
// sample gathers its <= ceil(seg / hop) contributions in slice order (no atomics).
#include "qi_common.hpp"
#include "qi_device.hpp"
#include "qi_fft_reg.hpp"  // QI_LAUNCH_CHECK

namespace qi {

namespace {

// sample k of the record extended past its ends by at most n - 1 samples (np.pad: constant 0 / edge / reflect with
// reflect_type even / odd)
template <typename T>
__device__ __forceinline__ T padded_sample(const T* __restrict__ x, int64_t n, int64_t k, int mode) {
  if (k >= 0 && k < n) return x[k];
  if (mode == 0) return T(0);
  if (mode == 1) return k < 0 ? x[0] : x[n - 1];
  const int64_t j = k < 0 ? -k : 2 * (n - 1) - k;  // mirrored about the end sample (not repeated)
  if (mode == 2) return x[j];
  return T(2) * (k < 0 ? x[0] : x[n - 1]) - x[j];  // point-symmetric about the end value
}

template <typename T>
__global__ void __launch_bounds__(256) k_sliding_frames(const T* __restrict__ sig, const T* __restrict__ win,
                                                        T* __restrict__ frames, int64_t n, int64_t seg, int64_t hop,
                                                        int64_t nfft, int64_t nseg, int64_t first, int pad_mode,
                                                        int detrend, int64_t roll) {
  __shared__ double s[256 / kWave];
  __shared__ double s_mean;
  const int tid = threadIdx.x, lane = tid & (kWave - 1), wv = tid / kWave;
  const int64_t m = blockIdx.x, c = blockIdx.y;
  const int64_t base = first + m * hop;  // record index of the slice's first sample (negative before the record)
  const T* x = sig + c * n;
  T mean = T(0);
  if (detrend) {
    double acc = 0.0;
    for (int64_t i = tid; i < seg; i += 256) acc += (double)padded_sample<T>(x, n, base + i, pad_mode);
    acc = wave_sum(acc);
    if (lane == 0) s[wv] = acc;
    __syncthreads();
    if (tid == 0) {
      double t = 0.0;
      for (int w = 0; w < 256 / kWave; ++w) t += s[w];
      s_mean = t / (double)seg;
    }
    __syncthreads();
    mean = (T)s_mean;
  }
  // the windowed slice, zero-padded to nfft, is rotated left by `roll` (ShortTimeFFT's phase_shift convention: with
  // roll = seg // 2 the window centre sits at index 0 of the transform)
  T* f = frames + (c * nseg + m) * nfft;
  for (int64_t i = tid; i < nfft; i += 256) {
    const T v = i < seg ? (padded_sample<T>(x, n, base + i, pad_mode) - mean) * win[i] : T(0);
    int64_t d = i - roll;
    if (d < 0) d += nfft;
    f[d] = v;
  }
}

// F [C * nseg][nf] -> out [C][nf][nseg]: complex coefficients, or their magnitude / squared magnitude
template <typename T>
__global__ void k_sliding_transpose(const cplx<T>* __restrict__ F, cplx<T>* __restrict__ Z, T* __restrict__ R, int kind,
                                    int64_t nseg, int64_t nf) {
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
      const cplx<T> z = tile[threadIdx.x][r];
      if (Z) Z[(c * nf + f) * nseg + m] = z;
      if (R) {
        const T p = z.x * z.x + z.y * z.y;
        R[(c * nf + f) * nseg + m] = kind == 2 ? p : sqrt_t(p);
      }
    }
  }
}

// S [C][nf][nseg] -> F [C * nseg][nf] (slice-major, what the batched inverse transform reads)
template <typename T>
__global__ void k_sliding_untranspose(const cplx<T>* __restrict__ S, cplx<T>* __restrict__ F, int64_t nseg, int64_t nf) {
  __shared__ cplx<T> tile[32][33];
  const int64_t c = blockIdx.z;
  const int64_t m0 = (int64_t)blockIdx.x * 32, f0 = (int64_t)blockIdx.y * 32;
  for (int r = threadIdx.y; r < 32; r += blockDim.y) {
    int64_t f = f0 + r, m = m0 + threadIdx.x;
    if (m < nseg && f < nf) tile[r][threadIdx.x] = S[(c * nf + f) * nseg + m];
  }
  __syncthreads();
  for (int r = threadIdx.y; r < 32; r += blockDim.y) {
    int64_t m = m0 + r, f = f0 + threadIdx.x;
    if (m < nseg && f < nf) F[(c * nseg + m) * nf + f] = tile[threadIdx.x][r];
  }
}

// x[k] = sum over the slices q that cover k of slice_q[k - (first + q hop)] * dual[k - (first + q hop)], k0 <= k < k1;
// `first` is the record index of slice 0's first sample; the unnormalised inverse transform is scaled by 1 / nfft.
template <typename T>
__global__ void __launch_bounds__(256) k_sliding_overlap_add(const T* __restrict__ slices, const T* __restrict__ dual,
                                                             T* __restrict__ out, int64_t k0, int64_t k1, int64_t seg,
                                                             int64_t hop, int64_t nfft, int64_t nseg, int64_t first,
                                                             int64_t roll) {
  const int64_t c = blockIdx.y;
  const T inv = T(1) / (T)nfft;
  for (int64_t k = k0 + (int64_t)blockIdx.x * 256 + threadIdx.x; k < k1; k += (int64_t)gridDim.x * 256) {
    // slices with first + q hop <= k < first + q hop + seg
    int64_t q_hi = (k - first) / hop;
    if (k - first < 0) q_hi = -1;
    int64_t q_lo = (k - first - seg) / hop + 1;
    if (k - first - seg < 0) q_lo = 0;
    if (q_hi > nseg - 1) q_hi = nseg - 1;
    T acc = T(0);
    for (int64_t q = q_lo; q <= q_hi; ++q) {
      const int64_t i = k - (first + q * hop);
      int64_t d = i - roll;  // undo the rotation of the forward transform
      if (d < 0) d += nfft;
      acc += slices[(c * nseg + q) * nfft + d] * dual[i];
    }
    out[c * (k1 - k0) + (k - k0)] = acc * inv;
  }
}

}  // namespace

template <typename T>
int launch_sliding_frames(const T* sig, const T* win, T* frames, int64_t C, int64_t n, int64_t seg, int64_t hop,
                          int64_t nfft, int64_t nseg, int64_t first, int pad_mode, int detrend, int64_t roll,
                          hipStream_t st) {
  k_sliding_frames<T><<<dim3((unsigned)nseg, (unsigned)C), 256, 0, st>>>(sig, win, frames, n, seg, hop, nfft, nseg, first,
                                                                          pad_mode, detrend, roll);
  QI_LAUNCH_CHECK();
  return QI_OK;
}

template <typename T>
int launch_sliding_transpose(const cplx<T>* F, cplx<T>* Z, T* R, int kind, int64_t C, int64_t nseg, int64_t nf,
                             hipStream_t st) {
  dim3 g((unsigned)ceil_div(nf, 32), (unsigned)ceil_div(nseg, 32), (unsigned)C);
  k_sliding_transpose<T><<<g, dim3(32, 8), 0, st>>>(F, Z, R, kind, nseg, nf);
  QI_LAUNCH_CHECK();
  return QI_OK;
}

template <typename T>
int launch_sliding_untranspose(const cplx<T>* S, cplx<T>* F, int64_t C, int64_t nseg, int64_t nf, hipStream_t st) {
  dim3 g((unsigned)ceil_div(nseg, 32), (unsigned)ceil_div(nf, 32), (unsigned)C);
  k_sliding_untranspose<T><<<g, dim3(32, 8), 0, st>>>(S, F, nseg, nf);
  QI_LAUNCH_CHECK();
  return QI_OK;
}

template <typename T>
int launch_sliding_overlap_add(const T* slices, const T* dual, T* out, int64_t C, int64_t k0, int64_t k1, int64_t seg,
                               int64_t hop, int64_t nfft, int64_t nseg, int64_t first, int64_t roll, hipStream_t st) {
  const int64_t g = ceil_div(k1 - k0, 256);
  k_sliding_overlap_add<T><<<dim3((unsigned)(g > 4096 ? 4096 : (g < 1 ? 1 : g)), (unsigned)C), 256, 0, st>>>(
      slices, dual, out, k0, k1, seg, hop, nfft, nseg, first, roll);
  QI_LAUNCH_CHECK();
  return QI_OK;
}

#define QI_INSTANTIATE_SLIDING(T)                                                                                     \
  template int launch_sliding_frames<T>(const T*, const T*, T*, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, \
                                        int64_t, int, int, int64_t, hipStream_t);                                     \
  template int launch_sliding_transpose<T>(const cplx<T>*, cplx<T>*, T*, int, int64_t, int64_t, int64_t, hipStream_t); \
  template int launch_sliding_untranspose<T>(const cplx<T>*, cplx<T>*, int64_t, int64_t, int64_t, hipStream_t);       \
  template int launch_sliding_overlap_add<T>(const T*, const T*, T*, int64_t, int64_t, int64_t, int64_t, int64_t,     \
                                             int64_t, int64_t, int64_t, int64_t, hipStream_t);
QI_INSTANTIATE_SLIDING(float)
QI_INSTANTIATE_SLIDING(double)

}  // namespace qi
