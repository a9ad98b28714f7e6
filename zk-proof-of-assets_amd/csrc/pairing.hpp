// Optimal-ate pairing on BN254 and Groth16 verification, host side (C++, no GPU).
//
// Next row after the prove path: the reference verifies every proof right after
// proving it with `npx snarkjs groth16 verify vkey public proof` (scripts/g16_verify.sh:213-216) and
// re-formats it for the next recursion layer with scripts/sanitize_groth16_proof.py:39-124, which
// needs e(-alpha, beta). Both are pinned byte-for-byte by the reference's committed fixtures
// (*_vkey.json, proof.json, public.json, sanitized_proof.json).
//
// Tower: Fq2 = Fq[u]/(u^2+1), Fq6 = Fq2[v]/(v^3 - xi), xi = 9+u, Fq12 = Fq6[w]/(w^2 - v). With t = w
// this is Fq[t]/(t^12 - 18 t^6 + 82), the basis py_ecc / the sanitizer use: the coefficient of t^i is the
// Fq2 coefficient of v^(i/2) w^(i%2) re-expressed through u = t^6 - 9.
// Miller loop: affine coordinates on the twist E'(Fq2): y^2 = x^3 + 3/xi; line through T with slope L
// evaluated at P = (xp, yp) in G1:  l = yp - L*xp * w + (L*xT - yT) * w^3   (w^3 = v w).
// Several pairings share one Miller accumulator: one Fq12 squaring per loop step whatever the number of pairs,
// and the slopes of all pairs at a step are divided with ONE field inversion (Montgomery's trick).
// Final exponentiation: f^((q^12-1)/r), exactly the value py_ecc produces, computed as the easy part
// f^((q^6-1)(q^2+1)) (a conjugation, one inversion, one q^2-Frobenius) followed by a plain square-and-multiply
// with the 761-bit exponent (q^4-q^2+1)/r -- a quarter of the work of the 2790-bit exponent, same result.
#pragma once
#include "bn254_ec.hip.h"
#include "host_field.hpp"

namespace zkpoa {
namespace pairing {

inline HFq2 fq2_xi() { return HFq2{HFq::from_u64(9), HFq::from_u64(1)}; }
inline HFq2 fq2_mul_xi(const HFq2& a) {  // (a0 + a1 u)(9 + u) = (9a0 - a1) + (a0 + 9a1) u
  HFq n0 = a.c0.dbl().dbl().dbl() + a.c0, n1 = a.c1.dbl().dbl().dbl() + a.c1;
  return HFq2{n0 - a.c1, a.c0 + n1};
}
inline HFq2 fq2_conj(const HFq2& a) { return HFq2{a.c0, a.c1.neg()}; }
inline HFq2 fq2_pow(const HFq2& a, const uint64_t e[4]) {
  HFq2 res = HFq2::one(), base = a;
  for (int i = 0; i < 4; i++) {
    uint64_t w = e[i];
    for (int b = 0; b < 64; b++) {
      if (w & 1) res = res * base;
      base = base.sqr();
      w >>= 1;
    }
  }
  return res;
}

struct Fq6 {
  HFq2 c0, c1, c2;
  static Fq6 zero() { return {HFq2::zero(), HFq2::zero(), HFq2::zero()}; }
  static Fq6 one() { return {HFq2::one(), HFq2::zero(), HFq2::zero()}; }
  friend Fq6 operator+(const Fq6& a, const Fq6& b) { return {a.c0 + b.c0, a.c1 + b.c1, a.c2 + b.c2}; }
  friend Fq6 operator-(const Fq6& a, const Fq6& b) { return {a.c0 - b.c0, a.c1 - b.c1, a.c2 - b.c2}; }
  friend Fq6 operator*(const Fq6& a, const Fq6& b) {  // schoolbook, v^3 = xi
    HFq2 t00 = a.c0 * b.c0, t11 = a.c1 * b.c1, t22 = a.c2 * b.c2;
    HFq2 t01 = a.c0 * b.c1 + a.c1 * b.c0;
    HFq2 t02 = a.c0 * b.c2 + a.c2 * b.c0;
    HFq2 t12 = a.c1 * b.c2 + a.c2 * b.c1;
    return {t00 + fq2_mul_xi(t12), t01 + fq2_mul_xi(t22), t02 + t11};
  }
  Fq6 mul_by_v() const { return {fq2_mul_xi(c2), c0, c1}; }
  Fq6 neg() const { return {c0.neg(), c1.neg(), c2.neg()}; }
  Fq6 inv() const {
    HFq2 t0 = c0.sqr() - fq2_mul_xi(c1 * c2);
    HFq2 t1 = fq2_mul_xi(c2.sqr()) - c0 * c1;
    HFq2 t2 = c1.sqr() - c0 * c2;
    HFq2 n = (c0 * t0 + fq2_mul_xi(c2 * t1 + c1 * t2)).inv();
    return {t0 * n, t1 * n, t2 * n};
  }
  bool is_zero() const { return c0.is_zero() && c1.is_zero() && c2.is_zero(); }
  bool operator==(const Fq6& b) const { return c0 == b.c0 && c1 == b.c1 && c2 == b.c2; }
};

struct Fq12 {
  Fq6 c0, c1;  // c0 + c1 w
  static Fq12 one() { return {Fq6::one(), Fq6::zero()}; }
  friend Fq12 operator*(const Fq12& a, const Fq12& b) {
    Fq6 t0 = a.c0 * b.c0, t1 = a.c1 * b.c1;
    Fq6 t2 = (a.c0 + a.c1) * (b.c0 + b.c1);
    return {t0 + t1.mul_by_v(), t2 - t0 - t1};
  }
  Fq12 sqr() const {  // (c0 + c1 w)^2 = (c0^2 + v c1^2) + 2 c0 c1 w, two Fq6 products
    Fq6 ab = c0 * c1;
    Fq6 t = (c0 + c1) * (c0 + c1.mul_by_v());          // c0^2 + v c1^2 + (1 + v) c0 c1
    return {t - ab - ab.mul_by_v(), ab + ab};
  }
  Fq12 conj() const { return {c0, c1.neg()}; }          // the q^6 Frobenius
  Fq12 inv() const {
    Fq6 n = (c0 * c0 - (c1 * c1).mul_by_v()).inv();
    return {c0 * n, (c1 * n).neg()};
  }
  // the q^2 Frobenius: Fq2 coefficients are fixed, w^(q^2) = gamma * w with gamma = xi^((q^2-1)/6) in Fq
  Fq12 frob2(const HFq gamma_pow[6]) const {
    auto sc = [](const HFq2& a, const HFq& k) { return HFq2{a.c0 * k, a.c1 * k}; };
    Fq12 r;
    r.c0 = {c0.c0, sc(c0.c1, gamma_pow[2]), sc(c0.c2, gamma_pow[4])};                       // w^0, w^2, w^4
    r.c1 = {sc(c1.c0, gamma_pow[1]), sc(c1.c1, gamma_pow[3]), sc(c1.c2, gamma_pow[5])};   // w^1, w^3, w^5
    return r;
  }
  bool operator==(const Fq12& b) const { return c0 == b.c0 && c1 == b.c1; }
  bool is_one() const { return *this == one(); }
  // the 6 Fq2 coefficients in the order the sanitizer emits them: t^0 .. t^5 (see header)
  void fq2_coeffs(HFq2 out[6]) const {
    out[0] = c0.c0; out[1] = c1.c0; out[2] = c0.c1; out[3] = c1.c1; out[4] = c0.c2; out[5] = c1.c2;
  }
};

typedef Affine<HFq> G1;    // infinity = (0, 0)
typedef Affine<HFq2> G2;

inline HFq2 twist_b() { return HFq2{HFq::from_u64(3), HFq::zero()} * fq2_xi().inv(); }
inline bool g1_on_curve(const G1& p) {
  if (p.is_inf()) return true;
  return p.y.sqr() == p.x.sqr() * p.x + HFq::from_u64(3);
}
inline bool g2_on_curve(const G2& p) {
  if (p.is_inf()) return true;
  return p.y.sqr() == p.x.sqr() * p.x + twist_b();
}

// f *= l, l = a + b w + c w^3 with a in Fq, b, c in Fq2 (sparse): as a full Fq12 element
inline Fq12 line_value(const HFq& yp, const HFq2& b_w, const HFq2& c_w3) {
  Fq12 l;
  l.c0 = {HFq2{yp, HFq::zero()}, HFq2::zero(), HFq2::zero()};
  l.c1 = {b_w, c_w3, HFq2::zero()};   // w and v*w = w^3
  return l;
}

// Frobenius constants on twisted coordinates: pi(x, y) = (conj(x) * g2, conj(y) * g3), g2 = xi^((q-1)/3),
// g3 = xi^((q-1)/2); gamma^i for the q^2 Frobenius of Fq12, gamma = norm(xi^((q-1)/6)) = norm(g3 / g2)
struct FrobConsts {
  HFq2 g2c, g3c;
  HFq gamma_pow[6];
  FrobConsts() {
    static const uint64_t e13[4] = {0x69602eb24829a9c2ull, 0xdd2b2385cd7b4384ull, 0xe81ac1e7808072c9ull, 0x10216f7ba065e00dull};
    static const uint64_t e12[4] = {0x9e10460b6c3e7ea3ull, 0xcbc0b548b438e546ull, 0xdc2822db40c0ac2eull, 0x183227397098d014ull};
    g2c = fq2_pow(fq2_xi(), e13);
    g3c = fq2_pow(fq2_xi(), e12);
    HFq2 g = g3c * g2c.inv();
    HFq gamma = g.c0.sqr() + g.c1.sqr();
    gamma_pow[0] = HFq::one();
    for (int i = 1; i < 6; i++) gamma_pow[i] = gamma_pow[i - 1] * gamma;
  }
};
inline const FrobConsts& frob_consts() {
  static const FrobConsts k;
  return k;
}

// out[i] = 1 / d[i] with one inversion (Montgomery's trick); every d[i] must be non-zero
inline void fq2_batch_inv(const HFq2* d, HFq2* out, int n) {
  HFq2 pre[8];
  HFq2 acc = HFq2::one();
  for (int i = 0; i < n; i++) {
    pre[i] = acc;
    acc = acc * d[i];
  }
  HFq2 inv = acc.inv();
  for (int i = n - 1; i >= 0; i--) {
    out[i] = inv * pre[i];
    inv = inv * d[i];
  }
}

// f = prod_i MillerLoop(Q_i, P_i), up to 8 pairs in lock step (pairs with an infinity on either side are skipped).
// T_i != +-Q_i holds on the r-torsion for the loop's indices, so no slope denominator is zero for valid inputs;
// a zero denominator (a point outside the r-torsion) makes the result 0, which no check accepts.
inline Fq12 multi_miller_loop(const G2* Qs, const G1* Ps, int count) {
  static const char* kAte = "11001110101111001011100000011100110111110011101100011101110101000";  // 6x+2, MSB first
  const FrobConsts& fc = frob_consts();
  G2 Q[8], T[8];
  G1 P[8];
  int n = 0;
  for (int i = 0; i < count && n < 8; i++)
    if (!Qs[i].is_inf() && !Ps[i].is_inf()) {
      Q[n] = T[n] = Qs[i];
      P[n] = Ps[i];
      n++;
    }
  Fq12 f = Fq12::one();
  if (n == 0) return f;
  HFq2 den[8], lam[8];
  bool degenerate = false;
  auto apply = [&](const G2* other) {   // other == nullptr: doubling step; else addition of other[i]
    for (int i = 0; i < n; i++) {
      den[i] = other ? other[i].x - T[i].x : T[i].y.dbl();
      if (den[i].is_zero()) degenerate = true;   // a point outside the r-torsion: no valid input gets here
    }
    if (degenerate) return;
    fq2_batch_inv(den, lam, n);
    for (int i = 0; i < n; i++) {
      HFq2 num;
      if (other) num = other[i].y - T[i].y;
      else {
        HFq2 xx = T[i].x.sqr();
        num = xx.dbl() + xx;
      }
      HFq2 L = num * lam[i];
      HFq2 x3 = L.sqr() - T[i].x - (other ? other[i].x : T[i].x);
      HFq2 y3 = L * (T[i].x - x3) - T[i].y;
      HFq2 b = HFq2{(L.c0 * P[i].x).neg(), (L.c1 * P[i].x).neg()};
      HFq2 c = L * T[i].x - T[i].y;
      T[i] = {x3, y3};
      f = f * line_value(P[i].y, b, c);
    }
  };
  for (const char* bit = kAte + 1; *bit; bit++) {
    f = f.sqr();
    apply(nullptr);
    if (*bit == '1') apply(Q);
  }
  G2 Q1[8], Q2[8];
  for (int i = 0; i < n; i++) {
    Q1[i] = {fq2_conj(Q[i].x) * fc.g2c, fq2_conj(Q[i].y) * fc.g3c};
    Q2[i] = {fq2_conj(Q1[i].x) * fc.g2c, (fq2_conj(Q1[i].y) * fc.g3c).neg()};
  }
  apply(Q1);
  apply(Q2);
  if (degenerate) return {Fq6::zero(), Fq6::zero()};   // final exponentiation of 0 is 0: never "== 1"
  return f;
}

inline Fq12 miller_loop(const G2& Q, const G1& P) { return multi_miller_loop(&Q, &P, 1); }

inline Fq12 final_exponentiation(const Fq12& f) {
  // (q^4 - q^2 + 1) / r, 761 bits, little-endian 64-bit limbs
  static const uint64_t kHard[12] = {
    0xe81bb482ccdf42b1ull, 0x5abf5cc4f49c36d4ull, 0xf1154e7e1da014fdull, 0xdcc7b44c87cdbacfull,
    0xaaa441e3954bcf8aull, 0x6b887d56d5095f23ull, 0x79581e16f3fd90c6ull, 0x3b1b1355d189227dull,
    0x4e529a5861876f6bull, 0x6c0eb522d5b12278ull, 0x331ec15183177fafull, 0x01baaa710b0759adull,
  };
  Fq12 f1 = f.conj() * f.inv();                                // ^(q^6 - 1)
  Fq12 f2 = f1.frob2(frob_consts().gamma_pow) * f1;            // ^(q^2 + 1)
  Fq12 res = Fq12::one();
  bool started = false;
  for (int i = 11; i >= 0; i--)
    for (int b = 63; b >= 0; b--) {
      if (started) res = res.sqr();
      if ((kHard[i] >> b) & 1) {
        res = started ? res * f2 : f2;
        started = true;
      }
    }
  return res;
}

inline Fq12 pairing(const G2& Q, const G1& P) { return final_exponentiation(miller_loop(Q, P)); }

}  // namespace pairing
}  // namespace zkpoa
