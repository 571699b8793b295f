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

}  // namespace
}  // namespace native
}  // namespace qi
