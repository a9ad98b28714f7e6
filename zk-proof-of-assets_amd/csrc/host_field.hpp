// Host-side BN254 field arithmetic (4 x u64 limbs, unsigned __int128 products) for the O(1)
// tail of the prove path: window combination of MSM results, the randomised proof assembly
// (SURVEY.md 3.2 steps 6-7 / 8a row a10), Jacobian -> affine, and decimal JSON output.
// Same class interface as the device Fp so the curve templates in bn254_ec.hip.h serve both.
// This is product code (it never runs an MSM/NTT on the CPU: the hot path is HIP-only).
#pragma once
#include <stdint.h>
#include <string.h>
#include <string>

namespace zkpoa {

typedef unsigned __int128 u128;

struct HFqParams {
  static constexpr uint64_t P[4] = {0x3c208c16d87cfd47ull, 0x97816a916871ca8dull, 0xb85045b68181585dull,
                                    0x30644e72e131a029ull};
  static constexpr uint64_t INV = 0x87d20782e4866389ull;
  static constexpr uint64_t ONE[4] = {0xd35d438dc58f0d9dull, 0x0a78eb28f5c70b3dull, 0x666ea36f7879462cull,
                                      0x0e0a77c19a07df2full};
  static constexpr uint64_t R2[4] = {0xf32cfc5b538afa89ull, 0xb5e71911d44501fbull, 0x47ab1eff0a417ff6ull,
                                     0x06d89f71cab8351full};
};
struct HFrParams {
  static constexpr uint64_t P[4] = {0x43e1f593f0000001ull, 0x2833e84879b97091ull, 0xb85045b68181585dull,
                                    0x30644e72e131a029ull};
  static constexpr uint64_t INV = 0xc2e1f593efffffffull;
  static constexpr uint64_t ONE[4] = {0xac96341c4ffffffbull, 0x36fc76959f60cd29ull, 0x666ea36f7879462eull,
                                      0x0e0a77c19a07df2full};
  static constexpr uint64_t R2[4] = {0x1bb8e645ae216da7ull, 0x53fe3ab1e35c59e3ull, 0x8c49833d53bb8085ull,
                                     0x0216d0b17f4e44a5ull};
};

template <class PRM>
struct HFp {
  uint64_t l[4];

  static HFp zero() { return HFp{{0, 0, 0, 0}}; }
  static HFp one() { return HFp{{PRM::ONE[0], PRM::ONE[1], PRM::ONE[2], PRM::ONE[3]}}; }
  static HFp r2() { return HFp{{PRM::R2[0], PRM::R2[1], PRM::R2[2], PRM::R2[3]}}; }
  static HFp from_bytes(const void* p) {
    HFp r;
    memcpy(r.l, p, 32);
    return r;
  }
  void to_bytes(void* p) const { memcpy(p, l, 32); }
  static HFp from_u64(uint64_t v) {  // standard form small value -> Montgomery
    HFp r{{v, 0, 0, 0}};
    return r.to_mont();
  }
  bool is_zero() const { return (l[0] | l[1] | l[2] | l[3]) == 0; }
  bool operator==(const HFp& b) const { return l[0] == b.l[0] && l[1] == b.l[1] && l[2] == b.l[2] && l[3] == b.l[3]; }
  bool operator!=(const HFp& b) const { return !(*this == b); }

  static bool geq_p(const uint64_t* a) {
    for (int i = 3; i >= 0; i--) {
      if (a[i] > PRM::P[i]) return true;
      if (a[i] < PRM::P[i]) return false;
    }
    return true;
  }
  static void sub_p(uint64_t* a) {
    u128 bw = 0;
    for (int i = 0; i < 4; i++) {
      u128 d = (u128)a[i] - PRM::P[i] - bw;
      a[i] = (uint64_t)d;
      bw = (d >> 64) & 1;
    }
  }
  friend HFp operator+(const HFp& a, const HFp& b) {
    HFp r;
    u128 c = 0;
    for (int i = 0; i < 4; i++) {
      c += (u128)a.l[i] + b.l[i];
      r.l[i] = (uint64_t)c;
      c >>= 64;
    }
    if (geq_p(r.l)) sub_p(r.l);
    return r;
  }
  friend HFp operator-(const HFp& a, const HFp& b) {
    HFp r;
    u128 bw = 0;
    for (int i = 0; i < 4; i++) {
      u128 d = (u128)a.l[i] - b.l[i] - bw;
      r.l[i] = (uint64_t)d;
      bw = (d >> 64) & 1;
    }
    if (bw) {
      u128 c = 0;
      for (int i = 0; i < 4; i++) {
        c += (u128)r.l[i] + PRM::P[i];
        r.l[i] = (uint64_t)c;
        c >>= 64;
      }
    }
    return r;
  }
  HFp neg() const {
    if (is_zero()) return *this;
    HFp z = zero();
    return z - *this;
  }
  HFp dbl() const { return *this + *this; }

  friend HFp operator*(const HFp& a, const HFp& b) {
    uint64_t t[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; i++) {
      u128 c = 0;
      for (int j = 0; j < 4; j++) {
        c += (u128)a.l[j] * b.l[i] + t[j];
        t[j] = (uint64_t)c;
        c >>= 64;
      }
      c += t[4];
      t[4] = (uint64_t)c;
      t[5] = (uint64_t)(c >> 64);
      uint64_t m = t[0] * PRM::INV;
      c = (u128)m * PRM::P[0] + t[0];
      c >>= 64;
      for (int j = 1; j < 4; j++) {
        c += (u128)m * PRM::P[j] + t[j];
        t[j - 1] = (uint64_t)c;
        c >>= 64;
      }
      c += t[4];
      t[3] = (uint64_t)c;
      t[4] = t[5] + (uint64_t)(c >> 64);
    }
    HFp r{{t[0], t[1], t[2], t[3]}};
    if (t[4] || geq_p(r.l)) sub_p(r.l);
    return r;
  }
  HFp sqr() const { return (*this) * (*this); }
  HFp to_mont() const { return (*this) * r2(); }
  HFp from_mont() const {
    HFp o{{1, 0, 0, 0}};
    return (*this) * o;
  }
  HFp pow(const uint64_t e[4]) const {
    HFp res = one(), base = *this;
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
  HFp inv() const {
    uint64_t e[4] = {PRM::P[0] - 2, PRM::P[1], PRM::P[2], PRM::P[3]};
    return pow(e);
  }
  // decimal string of the standard-form value
  std::string to_dec() const {
    HFp s = from_mont();
    uint64_t v[4] = {s.l[0], s.l[1], s.l[2], s.l[3]};
    std::string out;
    while (v[0] | v[1] | v[2] | v[3]) {
      u128 rem = 0;
      for (int i = 3; i >= 0; i--) {
        u128 cur = (rem << 64) | v[i];
        v[i] = (uint64_t)(cur / 10000000000000000000ull);
        rem = cur % 10000000000000000000ull;
      }
      uint64_t chunk = (uint64_t)rem;
      bool more = (v[0] | v[1] | v[2] | v[3]) != 0;
      for (int d = 0; d < 19 && (more || chunk); d++) {
        out.push_back('0' + (chunk % 10));
        chunk /= 10;
      }
    }
    if (out.empty()) out = "0";
    return std::string(out.rbegin(), out.rend());
  }
};

using HFq = HFp<HFqParams>;
using HFr = HFp<HFrParams>;

struct HFq2 {
  HFq c0, c1;
  static HFq2 zero() { return {HFq::zero(), HFq::zero()}; }
  static HFq2 one() { return {HFq::one(), HFq::zero()}; }
  static HFq2 from_bytes(const void* p) {
    return {HFq::from_bytes(p), HFq::from_bytes(reinterpret_cast<const char*>(p) + 32)};
  }
  void to_bytes(void* p) const {
    c0.to_bytes(p);
    c1.to_bytes(reinterpret_cast<char*>(p) + 32);
  }
  bool is_zero() const { return c0.is_zero() && c1.is_zero(); }
  bool operator==(const HFq2& b) const { return c0 == b.c0 && c1 == b.c1; }
  bool operator!=(const HFq2& b) const { return !(*this == b); }
  friend HFq2 operator+(const HFq2& a, const HFq2& b) { return {a.c0 + b.c0, a.c1 + b.c1}; }
  friend HFq2 operator-(const HFq2& a, const HFq2& b) { return {a.c0 - b.c0, a.c1 - b.c1}; }
  HFq2 neg() const { return {c0.neg(), c1.neg()}; }
  HFq2 dbl() const { return {c0.dbl(), c1.dbl()}; }
  friend HFq2 operator*(const HFq2& a, const HFq2& b) {
    HFq t0 = a.c0 * b.c0, t1 = a.c1 * b.c1;
    HFq t2 = (a.c0 + a.c1) * (b.c0 + b.c1);
    return {t0 - t1, t2 - t0 - t1};
  }
  HFq2 sqr() const {
    HFq t = c0 * c1;
    return {(c0 + c1) * (c0 - c1), t + t};
  }
  HFq2 inv() const {
    HFq d = (c0.sqr() + c1.sqr()).inv();
    return {c0 * d, (c1 * d).neg()};
  }
};

}  // namespace zkpoa
