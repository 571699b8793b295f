// Gather step of the zoom engine's coarse stage, shared by qi_zoom.hip (stand-alone gather launch) and qi_block.hip
// (gather fused into the 4096-point plane transforms).
#pragma once
#include "qi_common.hpp"
#include "qi_device.hpp"
#include "qi_native.hpp"

namespace qi {
namespace native {
namespace {

// Input of the coarse stage: with the band's coarse grid of M = P * 4096 samples and tau = P tau2 + tau1, the envelope
// b[tau] = sum_kappa Yc[kappa] exp(2 pi i kappa tau / M) is, for each tau1 < P, one 4096-point transform of
//   in[tau1][kappa0] = exp(2 pi i kappa0 tau1 / M) sum_r Yc[kappa0 + 4096 r] exp(2 pi i r tau1 / P).
// Yc = the band's occupied bins moved to baseband (Y as the one-pass loader of qi_native.hip forms it: spectrum x
// compact bank, or shifted spectrum x Gaussian).  One value per (kappa0, tau1), every block r that can hold occupied
// bins visited.
template <typename T, bool STX>
__device__ __forceinline__ cplx<T> zoom_gather_value(const ZoomArgs<T>& a, const BandDesc& bd, const uint32_t tau1,
                                                     const int32_t kappa0, const cplx<T>* __restrict__ X) {
  const int32_t M = (int32_t)((a.Lf / kZoomD) << bd.edge_slot), P = M / kBlk;
  const int32_t kc = STX ? 0 : bd.k_lo + bd.k_len / 2;
  const int32_t ks_lo = bd.k_lo - kc, ks_hi = ks_lo + bd.k_len;  // support in baseband bins
  const uint32_t lmask = (uint32_t)a.Lf - 1u;
  cplx<T> acc = mk<T>(T(0), T(0));
  // the occupied baseband bins congruent to kappa0 modulo 4096: ks = ks0, ks0 + 4096, ... < ks_hi
  const int32_t ks0 = ks_lo + ((kappa0 - ks_lo) & (kBlk - 1));
  if (ks0 >= ks_hi) return acc;
  for (int32_t ks = ks0; ks < ks_hi; ks += kBlk) {
    const uint32_t r = ((uint32_t)ks & ((uint32_t)M - 1u)) / kBlk;  // block of the M-point grid that holds bin ks
    const int32_t k = kc + ks;
    cplx<T> y;
    if (STX) {
      const cplx<T> x = X[((uint32_t)(k + (int32_t)bd.shift) & lmask) << a.x_shift];
      const T g0 = (T)bd.coef * (T)k;
      const T g = exp2_t(-g0 * g0) * a.inv_len;
      y = mk<T>(x.x * g, x.y * g);
    } else {
      y = cmul(X[((uint32_t)k & lmask) << a.x_shift], a.Hc[bd.src_off + (k - bd.k_lo)]);  // k < 0: bins modulo Lf
    }
    float sr, cr;
    sincospif(2.0f * (float)((r * tau1) & (uint32_t)(P - 1)) / (float)P, &sr, &cr);
    const cplx<T> t = cmul(y, mk<T>((T)cr, (T)sr));
    acc.x += t.x;
    acc.y += t.y;
  }
  float s, c;
  sincospif(2.0f * (float)(((uint32_t)kappa0 * tau1) & ((uint32_t)M - 1u)) / (float)M, &s, &c);
  return cmul(acc, mk<T>((T)c, (T)s));
}

// The sixteen inputs kappa0 = col + 256 b, b = 0..15, of one thread of the fused coarse stage (k_zoom_coarse2g): the
// same values as zoom_gather_value, with the global loads of a term issued for all sixteen before any arithmetic (a
// thread that walks its sixteen values one after the other waits sixteen times for memory), the block twiddle
// exp(2 pi i r tau1 / P) by a recurrence over the terms from two seeds (the sixteen first bins of a thread lie in at most
// two 4096-bin blocks of the grid), and the outer twiddle exp(2 pi i kappa0 tau1 / M) by binary powers from two seeds.
template <typename T, bool STX, int NF = 8>  // NF: loads in flight per thread (8 or 16)
__device__ __forceinline__ void zoom_gather16(const ZoomArgs<T>& a, const BandDesc& bd, const uint32_t tau1, const int col,
                                              const cplx<T>* __restrict__ X, cplx<T> (&v)[16]) {
  const int32_t M = (int32_t)((a.Lf / kZoomD) << bd.edge_slot), P = M / kBlk;
  const int32_t kc = STX ? 0 : bd.k_lo + bd.k_len / 2;
  const int32_t ks_lo = bd.k_lo - kc, ks_hi = ks_lo + bd.k_len;
  const uint32_t lmask = (uint32_t)a.Lf - 1u;
  const int nterm = (bd.k_len + kBlk - 1) / kBlk;  // terms of the element whose first bin is the support's first
  int32_t ks0[16];
#pragma unroll
  for (int b = 0; b < 16; ++b) {
    v[b] = mk<T>(T(0), T(0));
    ks0[b] = ks_lo + (((col + 256 * b) - ks_lo) & (kBlk - 1));
  }
  // block twiddles: r = r_a for the elements whose first bin lies in the block of ks_lo, r_a + 1 for the others
  const uint32_t r_a = ((uint32_t)ks_lo & ((uint32_t)M - 1u)) / kBlk;
  const uint32_t edge = (((uint32_t)ks_lo & ((uint32_t)M - 1u)) | (uint32_t)(kBlk - 1)) + 1u;  // first bin (mod M, unwrapped) of the next block
  float sf, cf;
  sincospif(2.0f * (float)((r_a * tau1) & (uint32_t)(P - 1)) / (float)P, &sf, &cf);
  cplx<T> w_a = mk<T>((T)cf, (T)sf);
  sincospif(2.0f * (float)(tau1 & (uint32_t)(P - 1)) / (float)P, &sf, &cf);
  const cplx<T> w_step = mk<T>((T)cf, (T)sf);
  cplx<T> w_b = cmul(w_a, w_step);
  const uint32_t base_mod = (uint32_t)ks_lo & ((uint32_t)M - 1u);
  for (int m = 0; m < nterm; ++m) {
#pragma unroll
    for (int hb = 0; hb < 16; hb += NF) {  // NF loads in flight at a time (sixteen cost a wave per SIMD in registers:
                                           // taken only by calls of few records, whose few workgroups wait on latency)
      cplx<T> x[NF], h[NF];
      bool on[NF];
#pragma unroll
      for (int q = 0; q < NF; ++q) {
        const int b = hb + q;
        const int32_t ks = ks0[b] + kBlk * m;
        on[q] = ks < ks_hi;
        const int32_t k = kc + ks;
        x[q] = mk<T>(T(0), T(0));
        h[q] = mk<T>(T(0), T(0));
        if (on[q]) {
          if (STX) {
            x[q] = X[((uint32_t)(k + (int32_t)bd.shift) & lmask) << a.x_shift];
          } else {
            x[q] = X[((uint32_t)k & lmask) << a.x_shift];
            h[q] = a.Hc[bd.src_off + (k - bd.k_lo)];
          }
        }
      }
#pragma unroll
      for (int q = 0; q < NF; ++q) {
        const int b = hb + q;
        if (!on[q]) continue;
        cplx<T> y;
        if (STX) {
          const T g0 = (T)bd.coef * (T)(kc + ks0[b] + kBlk * m);
          const T g = exp2_t(-g0 * g0) * a.inv_len;
          y = mk<T>(x[q].x * g, x[q].y * g);
        } else {
          y = cmul(x[q], h[q]);
        }
        const bool second = base_mod + (uint32_t)(ks0[b] - ks_lo) >= edge;  // this element's first bin is in the next block
        const cplx<T> t = cmul(y, second ? w_b : w_a);
        v[b].x += t.x;
        v[b].y += t.y;
      }
    }
    w_a = w_b;
    w_b = cmul(w_b, w_step);
  }
  // outer twiddle exp(2 pi i (col + 256 b) tau1 / M) = e0 * s^b, s = exp(2 pi i 256 tau1 / M), powers by binary products
  sincospif(2.0f * (float)(((uint32_t)col * tau1) & ((uint32_t)M - 1u)) / (float)M, &sf, &cf);
  const cplx<T> e0 = mk<T>((T)cf, (T)sf);
  sincospif(2.0f * (float)((256u * tau1) & ((uint32_t)M - 1u)) / (float)M, &sf, &cf);
  const cplx<T> s1 = mk<T>((T)cf, (T)sf), s2 = cmul(s1, s1), s4 = cmul(s2, s2), s8 = cmul(s4, s4);
  cplx<T> pw[16];
  pw[0] = e0;
#pragma unroll
  for (int b = 1; b < 16; ++b) {
    const int low = b & -b;
    pw[b] = cmul(pw[b - low], low == 1 ? s1 : (low == 2 ? s2 : (low == 4 ? s4 : s8)));
  }
#pragma unroll
  for (int b = 0; b < 16; ++b) v[b] = cmul(v[b], pw[b]);
}

}  // namespace
}  // namespace native
}  // namespace qi
