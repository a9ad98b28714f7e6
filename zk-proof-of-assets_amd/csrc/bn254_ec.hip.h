// BN254 G1 / G2 group arithmetic on the device, generic over the coordinate field
// (F = Fq for G1: y^2 = x^3 + 3;  F = Fq2 for G2: y^2 = x^3 + 3/(9+u)).
//
// Replaces the curve layer under G1.multiExpAffine / G2.multiExpAffine of the reference's
// external provers. Bases arrive exactly as a .zkey stores them:
// affine, Montgomery form, x then y, infinity = all-zero bytes.
//
// Accumulators use extended Jacobian "XYZZ" coordinates (x = X/ZZ, y = Y/ZZZ, ZZ^3 = ZZZ^2):
// mixed addition of an affine base costs 8M + 2S, with no inversion and no Z multiplication.
// The curve coefficient a is 0 for both groups, so no formula below needs b.
#pragma once
#include "bn254_field.hip.h"
#include <type_traits>

#define ZK_HD __host__ __device__ __forceinline__

namespace zkpoa {

template <class F>
struct Affine {
  F x, y;
  ZK_HD bool is_inf() const { return x.is_zero() && y.is_zero(); }
};

template <class F>
struct XYZZ {
  F x, y, zz, zzz;
  static ZK_HD XYZZ inf() { return {F::zero(), F::zero(), F::zero(), F::zero()}; }
  ZK_HD bool is_inf() const { return zz.is_zero(); }
  static ZK_HD XYZZ from_affine(const Affine<F>& p) {
    if (p.is_inf()) return inf();
    return {p.x, p.y, F::one(), F::one()};
  }
};

// 2 * (affine p), p != inf   [mdbl-2008-s-1]
template <class F>
ZK_HD XYZZ<F> xyzz_dbl_affine(const Affine<F>& p) {
  if (p.y.is_zero()) return XYZZ<F>::inf();
  F u = p.y.dbl();
  F v = u.sqr();
  F w = u * v;
  F s = p.x * v;
  F xx = p.x.sqr();
  F m = xx.dbl() + xx;
  XYZZ<F> r;
  r.x = m.sqr() - s.dbl();
  r.y = m * (s - r.x) - w * p.y;
  r.zz = v;
  r.zzz = w;
  return r;
}

// 2 * a   [dbl-2008-s-1]
template <class F>
ZK_HD XYZZ<F> xyzz_dbl(const XYZZ<F>& a) {
  if (a.is_inf() || a.y.is_zero()) return XYZZ<F>::inf();
  F u = a.y.dbl();
  F v = u.sqr();
  F w = u * v;
  F s = a.x * v;
  F xx = a.x.sqr();
  F m = xx.dbl() + xx;
  XYZZ<F> r;
  r.x = m.sqr() - s.dbl();
  r.y = m * (s - r.x) - w * a.y;
  r.zz = v * a.zz;
  r.zzz = w * a.zzz;
  return r;
}

// acc += p (affine); `negate` adds -p.   [madd-2008-s], with all exceptional cases
template <class F>
ZK_HD void xyzz_add_affine(XYZZ<F>& acc, const Affine<F>& p_in, bool negate) {
  if (p_in.is_inf()) return;
  Affine<F> p = p_in;
  if (negate) p.y = p.y.neg();
  if (acc.is_inf()) {
    acc = {p.x, p.y, F::one(), F::one()};
    return;
  }
#if defined(__HIP_DEVICE_COMPILE__) && !defined(ZKPOA_G1_UNPAIRED)
  // r04: the independent products of the G1 addition taken in pairs (Fq::mul_pair, two accumulator chains in lockstep):
  // u2 | s2, pp | r^2, ppp | q, r (q - x3) | y ppp, zz pp | zzz ppp. Same instructions and registers (149 VGPRs in the
  // accumulation kernel); the register-only loop gains 1.4 % at three waves per SIMD, 10 % at one (tools/microbench4.hip;
  // -DZKPOA_G1_UNPAIRED builds the r03 form).
  if constexpr (std::is_same<F, Fq>::value) {
    F u2, s2;
    Fq::mul_pair(p.x, acc.zz, p.y, acc.zzz, u2, s2);
    F pp_ = u2 - acc.x;
    F r = s2 - acc.y;
    if (pp_.is_zero()) {
      if (r.is_zero()) acc = xyzz_dbl_affine(p);
      else acc = XYZZ<F>::inf();
      return;
    }
    F pp, rr, ppp, q, t0, t1;
    Fq::sqr_pair(pp_, r, pp, rr);
    Fq::mul_pair(pp_, pp, acc.x, pp, ppp, q);
    F x3 = rr - ppp - q.dbl();
#if defined(ZKPOA_G1_Y3_TWO_PRODUCTS)
    Fq::mul_pair(r, q - x3, acc.y, ppp, t0, t1);
    acc.y = t0 - t1;
#else
    // y3 = r (q - x3) - y1 ppp as ONE sum of two products with one Montgomery reduction (Fq::dot2, as the Fq2 product):
    // 9 reductions per addition instead of 10, and no subtraction afterwards
    (void)t0;
    (void)t1;
    acc.y = Fq::dot2(r, q - x3, acc.y.neg_2p(), ppp);
#endif
    acc.x = x3;
    Fq::mul_pair(acc.zz, pp, acc.zzz, ppp, acc.zz, acc.zzz);
    return;
  }
#endif
  F u2 = p.x * acc.zz;
  F s2 = p.y * acc.zzz;
  F pp_ = u2 - acc.x;
  F r = s2 - acc.y;
  if (pp_.is_zero()) {
    if (r.is_zero()) acc = xyzz_dbl_affine(p);
    else acc = XYZZ<F>::inf();
    return;
  }
  F pp = pp_.sqr();
  F ppp = pp_ * pp;
  F q = acc.x * pp;
  F x3 = r.sqr() - ppp - q.dbl();
  F y3 = r * (q - x3) - acc.y * ppp;
  acc.x = x3;
  acc.y = y3;
  acc.zz = acc.zz * pp;
  acc.zzz = acc.zzz * ppp;
}

// acc += b   [add-2008-s], with all exceptional cases
template <class F>
ZK_HD void xyzz_add(XYZZ<F>& acc, const XYZZ<F>& b) {
  if (b.is_inf()) return;
  if (acc.is_inf()) {
    acc = b;
    return;
  }
  F u1 = acc.x * b.zz;
  F u2 = b.x * acc.zz;
  F s1 = acc.y * b.zzz;
  F s2 = b.y * acc.zzz;
  F pp_ = u2 - u1;
  F r = s2 - s1;
  if (pp_.is_zero()) {
    if (r.is_zero()) acc = xyzz_dbl(acc);
    else acc = XYZZ<F>::inf();
    return;
  }
  F pp = pp_.sqr();
  F ppp = pp_ * pp;
  F q = u1 * pp;
  F x3 = r.sqr() - ppp - q.dbl();
  F y3;
#if defined(__HIP_DEVICE_COMPILE__) && !defined(ZKPOA_G1_Y3_TWO_PRODUCTS)
  if constexpr (std::is_same<F, Fq>::value) y3 = Fq::dot2(r, q - x3, s1.neg_2p(), ppp);   // one reduction, as xyzz_add_affine
  else
#endif
    y3 = r * (q - x3) - s1 * ppp;
  acc.x = x3;
  acc.y = y3;
  acc.zz = acc.zz * b.zz * pp;
  acc.zzz = acc.zzz * b.zzz * ppp;
}

template <class F>
ZK_HD XYZZ<F> xyzz_neg(const XYZZ<F>& a) {
  return {a.x, a.y.neg(), a.zz, a.zzz};
}

// k * a for a small unsigned k (double-and-add, MSB first); used for bucket-index weights.
template <class F>
ZK_HD XYZZ<F> xyzz_mul_small(const XYZZ<F>& a, uint32_t k) {
  XYZZ<F> r = XYZZ<F>::inf();
  if (k == 0) return r;
  int top = 31 - __builtin_clz(k);
  for (int b = top; b >= 0; b--) {
    r = xyzz_dbl(r);
    if ((k >> b) & 1) xyzz_add(r, a);
  }
  return r;
}

// ---- memory forms ------------------------------------------------------------------------------
template <class F>
ZK_DEV Affine<F> load_affine(const void* base, size_t idx) {
  constexpr int FB = FieldBytes<F>::N;
  const char* p = reinterpret_cast<const char*>(base) + idx * (2 * FB);
  return {load_field<F>(p), load_field<F>(p + FB)};
}
template <class F>
ZK_DEV XYZZ<F> load_xyzz(const void* base, size_t idx) {
  constexpr int FB = FieldBytes<F>::N;
  const char* p = reinterpret_cast<const char*>(base) + idx * (4 * FB);
  return {load_field<F>(p), load_field<F>(p + FB), load_field<F>(p + 2 * FB), load_field<F>(p + 3 * FB)};
}
template <class F>
ZK_DEV void store_xyzz(void* base, size_t idx, const XYZZ<F>& v) {
  constexpr int FB = FieldBytes<F>::N;
  char* p = reinterpret_cast<char*>(base) + idx * (4 * FB);
  store_field(p, v.x);
  store_field(p + FB, v.y);
  store_field(p + 2 * FB, v.zz);
  store_field(p + 3 * FB, v.zzz);
}

}  // namespace zkpoa
