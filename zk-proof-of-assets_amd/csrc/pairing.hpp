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
// Final exponentiation: plain f^((q^12-1)/r) (2790-bit exponent), exactly the value py_ecc produces.
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
  Fq12 sqr() const { return (*this) * (*this); }
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

struct MillerState {
  G2 T;
};
// doubling step: returns the line through T,T at P and sets T = 2T
inline Fq12 step_double(G2& T, const G1& P) {
  HFq2 xx = T.x.sqr();
  HFq2 lam = (xx.dbl() + xx) * T.y.dbl().inv();
  HFq2 x3 = lam.sqr() - T.x.dbl();
  HFq2 y3 = lam * (T.x - x3) - T.y;
  HFq2 b = HFq2{(lam.c0 * P.x).neg(), (lam.c1 * P.x).neg()};
  HFq2 c = lam * T.x - T.y;
  T = {x3, y3};
  return line_value(P.y, b, c);
}
// addition step: line through T,Q at P, T = T + Q (T != +-Q on the r-torsion for the loop's indices)
inline Fq12 step_add(G2& T, const G2& Qp, const G1& P) {
  HFq2 lam = (Qp.y - T.y) * (Qp.x - T.x).inv();
  HFq2 x3 = lam.sqr() - T.x - Qp.x;
  HFq2 y3 = lam * (T.x - x3) - T.y;
  HFq2 b = HFq2{(lam.c0 * P.x).neg(), (lam.c1 * P.x).neg()};
  HFq2 c = lam * T.x - T.y;
  T = {x3, y3};
  return line_value(P.y, b, c);
}

inline Fq12 miller_loop(const G2& Q, const G1& P) {
  if (Q.is_inf() || P.is_inf()) return Fq12::one();
  static const char* kAte = "11001110101111001011100000011100110111110011101100011101110101000";  // 6x+2, MSB first
  // Frobenius on twisted coordinates: pi(x, y) = (conj(x) * g2, conj(y) * g3), g2 = xi^((q-1)/3), g3 = xi^((q-1)/2)
  static const uint64_t e13[4] = {0x69602eb24829a9c2ull, 0xdd2b2385cd7b4384ull, 0xe81ac1e7808072c9ull, 0x10216f7ba065e00dull};
  static const uint64_t e12[4] = {0x9e10460b6c3e7ea3ull, 0xcbc0b548b438e546ull, 0xdc2822db40c0ac2eull, 0x183227397098d014ull};
  const HFq2 g2c = fq2_pow(fq2_xi(), e13), g3c = fq2_pow(fq2_xi(), e12);
  Fq12 f = Fq12::one();
  G2 T = Q;
  for (const char* b = kAte + 1; *b; b++) {
    f = f.sqr() * step_double(T, P);
    if (*b == '1') f = f * step_add(T, Q, P);
  }
  G2 Q1 = {fq2_conj(Q.x) * g2c, fq2_conj(Q.y) * g3c};
  G2 Q2 = {fq2_conj(Q1.x) * g2c, (fq2_conj(Q1.y) * g3c).neg()};
  f = f * step_add(T, Q1, P);
  f = f * step_add(T, Q2, P);
  return f;
}

inline Fq12 final_exponentiation(const Fq12& f) {
  static const uint64_t kExp[44] = {
    0x86964b64ca86f120ull, 0x40a4efb7e54523a4ull, 0x837fa97896e84abbull, 0x361102b6b9b2b918ull,
    0xc0de81def35692daull, 0xbe04c7e8a6c3c760ull, 0xd766f9c9d570bb7full, 0xc230974d83561841ull,
    0x5bba1668c3be69a3ull, 0x7f3811c410526294ull, 0x29baee7ddadda71cull, 0xbf813b8d145da900ull,
    0x641bbadf423f9a2cull, 0xa80bb4ea44eacc5eull, 0xcd65664814fde37cull, 0x4a0364b9580291d2ull,
    0xee93dfb10826f0ddull, 0x6b42db8dc5514724ull, 0xbb10cf430b0f3785ull, 0x40494e406f804216ull,
    0x55cfe107acf3aafbull, 0x2088ec80e0ebae87ull, 0x846a3ed011a337a0ull, 0x48a45a4a1e3a5195ull,
    0xe5664568dfc50e16ull, 0xab6a41294c0cc4ebull, 0x82d0d602d268c7daull, 0x6668449aed3cc48aull,
    0x5062cd0fb2015dfcull, 0x7f2940a8b1ddb3d1ull, 0x77f5b63a2a226448ull, 0xfef0781361e443aeull,
    0xf977870e88d5c6c8ull, 0x790364a61f676baaull, 0x5887e72eceaddea3ull, 0x1377e563a09a1b70ull,
    0x0c54efee1bd8c3b2ull, 0x3ec3d15ad524d8f7ull, 0xdaf15466b2383a5dull, 0xe1e30a73bb94fec0ull,
    0x6a1c71015f3f7be2ull, 0x842d43bf6369b1ffull, 0x20fddadf107d20bcull, 0x0000002f4b6dc970ull,
  };
  Fq12 res = Fq12::one(), base = f;
  for (int i = 0; i < 44; i++) {
    uint64_t w = kExp[i];
    for (int b = 0; b < 64; b++) {
      if (w & 1) res = res * base;
      base = base.sqr();
      w >>= 1;
    }
  }
  return res;
}

inline Fq12 pairing(const G2& Q, const G1& P) { return final_exponentiation(miller_loop(Q, P)); }

}  // namespace pairing
}  // namespace zkpoa
