// BN254 prime-field arithmetic for gfx950 (CDNA4): Fq (base field) and Fr (scalar field).
//
// Replaces, on the device, the field layer the reference's external provers use
// (rapidsnark fq.asm/fr.asm, ffjavascript WasmField1 -- SURVEY.md 8a row a4): 254-bit primes,
// elements stored as 4 x u64 little-endian (= 8 x u32 here), Montgomery radix R = 2^256.
//
// CDNA4 has no 64x64 multiplier; the widest integer multiply is v_mad_u64_u32
// (32x32 + 64 -> 64, with carry-out). So "4-limb 64-bit" is the storage layout and the
// arithmetic is 8 x 32-bit limbs: product-scanning (column-wise) Montgomery multiplication,
// each product = one v_mad_u64_u32 into a 64-bit column accumulator + one v_addc into the
// third accumulator word.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace zkpoa {

#define ZK_DEV __device__ __forceinline__

struct FqParams {
  // q = 0x30644e72e131a029b85045b68181585d97816a916871ca8d3c208c16d87cfd47
  static constexpr uint32_t P[8] = {0xd87cfd47u, 0x3c208c16u, 0x6871ca8du, 0x97816a91u,
                                    0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
  // 2q (elements are kept lazily in [0, 2q): 4q < 2^256)
  static constexpr uint32_t P2[8] = {0xb0f9fa8eu, 0x7841182du, 0xd0e3951au, 0x2f02d522u,
                                     0x0302b0bbu, 0x70a08b6du, 0xc2634053u, 0x60c89ce5u};
  static constexpr uint32_t INV = 0xe4866389u;  // -q^-1 mod 2^32
  // R mod q
  static constexpr uint32_t ONE[8] = {0xc58f0d9du, 0xd35d438du, 0xf5c70b3du, 0x0a78eb28u,
                                      0x7879462cu, 0x666ea36fu, 0x9a07df2fu, 0x0e0a77c1u};
  // R^2 mod q
  static constexpr uint32_t R2[8] = {0x538afa89u, 0xf32cfc5bu, 0xd44501fbu, 0xb5e71911u,
                                     0x0a417ff6u, 0x47ab1effu, 0xcab8351fu, 0x06d89f71u};
};

struct FrParams {
  // r = 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001
  static constexpr uint32_t P[8] = {0xf0000001u, 0x43e1f593u, 0x79b97091u, 0x2833e848u,
                                    0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
  static constexpr uint32_t P2[8] = {0xe0000002u, 0x87c3eb27u, 0xf372e122u, 0x5067d090u,
                                     0x0302b0bau, 0x70a08b6du, 0xc2634053u, 0x60c89ce5u};
  static constexpr uint32_t INV = 0xefffffffu;  // -r^-1 mod 2^32
  static constexpr uint32_t ONE[8] = {0x4ffffffbu, 0xac96341cu, 0x9f60cd29u, 0x36fc7695u,
                                      0x7879462eu, 0x666ea36fu, 0x9a07df2fu, 0x0e0a77c1u};
  static constexpr uint32_t R2[8] = {0xae216da7u, 0x1bb8e645u, 0xe35c59e3u, 0x53fe3ab1u,
                                     0x53bb8085u, 0x8c49833du, 0x7f4e44a5u, 0x0216d0b1u};
};

// ---- carry helpers ---------------------------------------------------------------------------
ZK_DEV uint32_t addc(uint32_t a, uint32_t b, uint32_t& carry) {
  uint32_t co;
  uint32_t r = __builtin_addc(a, b, carry, &co);
  carry = co;
  return r;
}
ZK_DEV uint32_t subb(uint32_t a, uint32_t b, uint32_t& borrow) {
  uint32_t bo;
  uint32_t r = __builtin_subc(a, b, borrow, &bo);
  borrow = bo;
  return r;
}

// acc (96 bit: lo64 + hi32) += a * b
ZK_DEV void mac96(uint64_t& lo, uint32_t& hi, uint32_t a, uint32_t b) {
  asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\t"
      "v_addc_co_u32 %1, vcc, 0, %1, vcc"
      : "+v"(lo), "+v"(hi)
      : "v"(a), "v"(b)
      : "vcc");
}

// two such accumulators advanced together: the second carry travels in an SGPR pair, so the two mads issue back to
// back and neither addc waits for the other chain (a wave's own issue rate is what limits the multiplier at two waves
// per SIMD -- the G2 kernels: tools/microbench4.hip, +6 % per Fq2 product there, nothing at eight waves)
ZK_DEV void mac96x2(uint64_t& lo0, uint32_t& hi0, uint32_t a0, uint32_t b0, uint64_t& lo1, uint32_t& hi1, uint32_t a1,
                    uint32_t b1) {
  uint64_t c1;
  asm("v_mad_u64_u32 %0, vcc, %5, %6, %0\n\t"
      "v_mad_u64_u32 %2, %4, %7, %8, %2\n\t"
      "v_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
      "v_addc_co_u32 %3, %4, 0, %3, %4"
      : "+v"(lo0), "+v"(hi0), "+v"(lo1), "+v"(hi1), "=&s"(c1)
      : "v"(a0), "v"(b0), "v"(a1), "v"(b1)
      : "vcc");
}

template <class PRM>
struct Fp {
  uint32_t l[8];

  static ZK_DEV Fp zero() {
    Fp r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.l[i] = 0;
    return r;
  }
  static ZK_DEV Fp one() {  // Montgomery form of 1
    Fp r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.l[i] = PRM::ONE[i];
    return r;
  }
  static ZK_DEV Fp r2() {
    Fp r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.l[i] = PRM::R2[i];
    return r;
  }
  // Lazy reduction: every value of this class lies in [0, 2p) (4p < 2^256 for both BN254 primes), so the
  // Montgomery product needs no final conditional subtraction: (a*b + m*p)/R < 2p whenever a, b < 2p.
  // Values are brought to the canonical range [0, p) only where representation matters: stores,
  // comparisons, bit scans (canon()).
  ZK_DEV Fp canon() const {  // [0, 2p) -> [0, p)
    Fp d;
    uint32_t bw = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) d.l[i] = subb(l[i], PRM::P[i], bw);
    Fp r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.l[i] = bw ? l[i] : d.l[i];
    return r;
  }
  ZK_DEV bool is_zero() const {  // value == 0 mod p, i.e. representation 0 or p
    uint32_t o = 0, e = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
      o |= l[i];
      e |= (l[i] ^ PRM::P[i]);
    }
    return o == 0 || e == 0;
  }
  ZK_DEV bool operator==(const Fp& b) const { return (*this - b).is_zero(); }
  ZK_DEV bool operator!=(const Fp& b) const { return !(*this == b); }

  // r = (a >= 2p) ? a - 2p : a     (a < 4p)
  static ZK_DEV Fp reduce_2p(const Fp& a) {
    Fp d;
    uint32_t bw = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) d.l[i] = subb(a.l[i], PRM::P2[i], bw);
    Fp r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.l[i] = bw ? a.l[i] : d.l[i];
    return r;
  }

  friend ZK_DEV Fp operator+(const Fp& a, const Fp& b) {
    Fp s;
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) s.l[i] = addc(a.l[i], b.l[i], c);
    // a + b < 4p < 2^256: no carry out of the top limb
    return reduce_2p(s);
  }
  friend ZK_DEV Fp operator-(const Fp& a, const Fp& b) {
    Fp d;
    uint32_t bw = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) d.l[i] = subb(a.l[i], b.l[i], bw);
    uint32_t mask = 0u - bw;
    uint32_t c = 0;
    Fp r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.l[i] = addc(d.l[i], PRM::P2[i] & mask, c);
    return r;
  }
  ZK_DEV Fp neg() const {  // 2p - a, except 0 -> 0 (keeps the result below 2p)
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) o |= l[i];
    if (o == 0) return *this;
    Fp r;
    uint32_t bw = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) r.l[i] = subb(PRM::P2[i], l[i], bw);
    return r;
  }
  ZK_DEV Fp dbl() const { return *this + *this; }

  // Montgomery product a*b/R mod p, finely-integrated product scanning.
  friend ZK_DEV Fp operator*(const Fp& a, const Fp& b) {
    uint64_t lo = 0;
    uint32_t hi = 0;
    uint32_t m[8];
    Fp r;
#pragma unroll
    for (int k = 0; k < 8; k++) {
#pragma unroll
      for (int i = 0; i <= k; i++) mac96(lo, hi, a.l[i], b.l[k - i]);
#pragma unroll
      for (int i = 0; i < k; i++) mac96(lo, hi, m[i], PRM::P[k - i]);
      m[k] = (uint32_t)lo * PRM::INV;
      mac96(lo, hi, m[k], PRM::P[0]);
      lo = (lo >> 32) | ((uint64_t)hi << 32);
      hi = 0;
    }
#pragma unroll
    for (int k = 8; k < 16; k++) {
#pragma unroll
      for (int i = k - 7; i < 8; i++) mac96(lo, hi, a.l[i], b.l[k - i]);
#pragma unroll
      for (int i = k - 7; i < 8; i++) mac96(lo, hi, m[i], PRM::P[k - i]);
      r.l[k - 8] = (uint32_t)lo;
      lo = (lo >> 32) | ((uint64_t)hi << 32);
      hi = 0;
    }
    return r;  // < 2p (lazy reduction)
  }
  // Montgomery square: the 28 cross products a_i a_j (i < j) are taken once against a pre-doubled operand instead of
  // twice -- 36 + 72 products instead of 64 + 72 (every product is a mad + addc pair: -20 % per squaring).
  // 2 * sum_{i<j} a_i a_j X^(i+j) = sum_i a_i X^i * (2 * floor(a / X^(i+1)) * X^(i+1)), and the limbs of the doubled tail
  // are (a_(i+1) << 1) for j = i + 1 (no bit comes in from below: the tail was cut there) and (a_j << 1) | (a_(j-1) >> 31)
  // for j > i + 1; the limb above the top is a_7 >> 31 = 0 because a < 2p < 2^255.
  ZK_DEV Fp sqr() const {
    uint32_t lo2[8], d2[8];
#pragma unroll
    for (int j = 1; j < 8; j++) {
      lo2[j] = l[j] << 1;
      d2[j] = __builtin_amdgcn_alignbit(l[j], l[j - 1], 31);   // (l[j] << 1) | (l[j-1] >> 31)
    }
    uint64_t lo = 0;
    uint32_t hi = 0;
    uint32_t m[8];
    Fp r;
#pragma unroll
    for (int k = 0; k < 8; k++) {
#pragma unroll
      for (int i = 0; 2 * i < k; i++) mac96(lo, hi, l[i], (k - i == i + 1) ? lo2[k - i] : d2[k - i]);
      if ((k & 1) == 0) mac96(lo, hi, l[k / 2], l[k / 2]);
#pragma unroll
      for (int i = 0; i < k; i++) mac96(lo, hi, m[i], PRM::P[k - i]);
      m[k] = (uint32_t)lo * PRM::INV;
      mac96(lo, hi, m[k], PRM::P[0]);
      lo = (lo >> 32) | ((uint64_t)hi << 32);
      hi = 0;
    }
#pragma unroll
    for (int k = 8; k < 16; k++) {
#pragma unroll
      for (int i = k - 7; 2 * i < k; i++) mac96(lo, hi, l[i], (k - i == i + 1) ? lo2[k - i] : d2[k - i]);
      if ((k & 1) == 0 && k / 2 < 8) mac96(lo, hi, l[k / 2], l[k / 2]);
#pragma unroll
      for (int i = k - 7; i < 8; i++) mac96(lo, hi, m[i], PRM::P[k - i]);
      r.l[k - 8] = (uint32_t)lo;
      lo = (lo >> 32) | ((uint64_t)hi << 32);
      hi = 0;
    }
    return r;  // < 2p, as operator*
  }

  // (a0 * b0 + a1 * b1) / R mod p with ONE Montgomery reduction: both products are accumulated column by column
  // before the shared m * p terms (128 + 72 products instead of 2 x 136, and no addition afterwards). Inputs < 2p
  // (b1 may equal 2p): T < 8p^2, so (T + m p) / R < p (8p / R + 1) < 2.6 p < 2^256 and one conditional subtraction of
  // 2p restores the lazy range [0, 2p). Fq2 products are two of these (lazy reduction over the extension field).
  static ZK_DEV Fp dot2(const Fp& a0, const Fp& b0, const Fp& a1, const Fp& b1) {
    uint64_t lo = 0;
    uint32_t hi = 0;
    uint32_t m[8];
    Fp r;
#pragma unroll
    for (int k = 0; k < 8; k++) {
#pragma unroll
      for (int i = 0; i <= k; i++) {
        mac96(lo, hi, a0.l[i], b0.l[k - i]);
        mac96(lo, hi, a1.l[i], b1.l[k - i]);
      }
#pragma unroll
      for (int i = 0; i < k; i++) mac96(lo, hi, m[i], PRM::P[k - i]);
      m[k] = (uint32_t)lo * PRM::INV;
      mac96(lo, hi, m[k], PRM::P[0]);
      lo = (lo >> 32) | ((uint64_t)hi << 32);
      hi = 0;
    }
#pragma unroll
    for (int k = 8; k < 16; k++) {
#pragma unroll
      for (int i = k - 7; i < 8; i++) {
        mac96(lo, hi, a0.l[i], b0.l[k - i]);
        mac96(lo, hi, a1.l[i], b1.l[k - i]);
      }
#pragma unroll
      for (int i = k - 7; i < 8; i++) mac96(lo, hi, m[i], PRM::P[k - i]);
      r.l[k - 8] = (uint32_t)lo;
      lo = (lo >> 32) | ((uint64_t)hi << 32);
      hi = 0;
    }
    return reduce_2p(r);
  }
  // (a0 * b0 + a1 * b1 + a2 * b2) / R mod p with one reduction (3 x 64 + 72 products instead of 3 x 136): a row of
  // Poseidon's 3 x 3 MDS product. The b operands must be fully reduced (< p: the round matrices are constants stored
  // that way), the a operands < 2p: T < 6p^2, (T + m p) / R < p (6p / R + 1) < 2.14 p for both BN254 moduli
  // (p / R < 0.1891), so one conditional subtraction of 2p restores [0, 2p).
  static ZK_DEV Fp dot3(const Fp& a0, const Fp& b0, const Fp& a1, const Fp& b1, const Fp& a2, const Fp& b2) {
    uint64_t lo = 0;
    uint32_t hi = 0;
    uint32_t m[8];
    Fp r;
#pragma unroll
    for (int k = 0; k < 8; k++) {
#pragma unroll
      for (int i = 0; i <= k; i++) {
        mac96(lo, hi, a0.l[i], b0.l[k - i]);
        mac96(lo, hi, a1.l[i], b1.l[k - i]);
        mac96(lo, hi, a2.l[i], b2.l[k - i]);
      }
#pragma unroll
      for (int i = 0; i < k; i++) mac96(lo, hi, m[i], PRM::P[k - i]);
      m[k] = (uint32_t)lo * PRM::INV;
      mac96(lo, hi, m[k], PRM::P[0]);
      lo = (lo >> 32) | ((uint64_t)hi << 32);
      hi = 0;
    }
#pragma unroll
    for (int k = 8; k < 16; k++) {
#pragma unroll
      for (int i = k - 7; i < 8; i++) {
        mac96(lo, hi, a0.l[i], b0.l[k - i]);
        mac96(lo, hi, a1.l[i], b1.l[k - i]);
        mac96(lo, hi, a2.l[i], b2.l[k - i]);
      }
#pragma unroll
      for (int i = k - 7; i < 8; i++) mac96(lo, hi, m[i], PRM::P[k - i]);
      r.l[k - 8] = (uint32_t)lo;
      lo = (lo >> 32) | ((uint64_t)hi << 32);
      hi = 0;
    }
    return reduce_2p(r);
  }
  // (a0 b0 + a1 b1, c0 d0 + c1 d1): two dot2 advanced column by column in lockstep (mac96x2) -- the two coordinates
  // of an Fq2 product
  static ZK_DEV void dot2_pair(const Fp& a0, const Fp& b0, const Fp& a1, const Fp& b1, const Fp& c0, const Fp& d0,
                               const Fp& c1, const Fp& d1, Fp& r0, Fp& r1) {
    uint64_t lo0 = 0, lo1 = 0;
    uint32_t hi0 = 0, hi1 = 0, m0[8], m1[8];
    Fp x, y;
#pragma unroll
    for (int k = 0; k < 8; k++) {
#pragma unroll
      for (int i = 0; i <= k; i++) {
        mac96x2(lo0, hi0, a0.l[i], b0.l[k - i], lo1, hi1, c0.l[i], d0.l[k - i]);
        mac96x2(lo0, hi0, a1.l[i], b1.l[k - i], lo1, hi1, c1.l[i], d1.l[k - i]);
      }
#pragma unroll
      for (int i = 0; i < k; i++) mac96x2(lo0, hi0, m0[i], PRM::P[k - i], lo1, hi1, m1[i], PRM::P[k - i]);
      m0[k] = (uint32_t)lo0 * PRM::INV;
      m1[k] = (uint32_t)lo1 * PRM::INV;
      mac96x2(lo0, hi0, m0[k], PRM::P[0], lo1, hi1, m1[k], PRM::P[0]);
      lo0 = (lo0 >> 32) | ((uint64_t)hi0 << 32);
      hi0 = 0;
      lo1 = (lo1 >> 32) | ((uint64_t)hi1 << 32);
      hi1 = 0;
    }
#pragma unroll
    for (int k = 8; k < 16; k++) {
#pragma unroll
      for (int i = k - 7; i < 8; i++) {
        mac96x2(lo0, hi0, a0.l[i], b0.l[k - i], lo1, hi1, c0.l[i], d0.l[k - i]);
        mac96x2(lo0, hi0, a1.l[i], b1.l[k - i], lo1, hi1, c1.l[i], d1.l[k - i]);
      }
#pragma unroll
      for (int i = k - 7; i < 8; i++) mac96x2(lo0, hi0, m0[i], PRM::P[k - i], lo1, hi1, m1[i], PRM::P[k - i]);
      x.l[k - 8] = (uint32_t)lo0;
      y.l[k - 8] = (uint32_t)lo1;
      lo0 = (lo0 >> 32) | ((uint64_t)hi0 << 32);
      hi0 = 0;
      lo1 = (lo1 >> 32) | ((uint64_t)hi1 << 32);
      hi1 = 0;
    }
    r0 = reduce_2p(x);
    r1 = reduce_2p(y);
  }
  // (a b, c d): two independent Montgomery products in lockstep (results < 2p, as operator*)
  static ZK_DEV void mul_pair(const Fp& a, const Fp& b, const Fp& c, const Fp& d, Fp& r0, Fp& r1) {
    uint64_t lo0 = 0, lo1 = 0;
    uint32_t hi0 = 0, hi1 = 0, m0[8], m1[8];
#pragma unroll
    for (int k = 0; k < 8; k++) {
#pragma unroll
      for (int i = 0; i <= k; i++) mac96x2(lo0, hi0, a.l[i], b.l[k - i], lo1, hi1, c.l[i], d.l[k - i]);
#pragma unroll
      for (int i = 0; i < k; i++) mac96x2(lo0, hi0, m0[i], PRM::P[k - i], lo1, hi1, m1[i], PRM::P[k - i]);
      m0[k] = (uint32_t)lo0 * PRM::INV;
      m1[k] = (uint32_t)lo1 * PRM::INV;
      mac96x2(lo0, hi0, m0[k], PRM::P[0], lo1, hi1, m1[k], PRM::P[0]);
      lo0 = (lo0 >> 32) | ((uint64_t)hi0 << 32);
      hi0 = 0;
      lo1 = (lo1 >> 32) | ((uint64_t)hi1 << 32);
      hi1 = 0;
    }
#pragma unroll
    for (int k = 8; k < 16; k++) {
#pragma unroll
      for (int i = k - 7; i < 8; i++) mac96x2(lo0, hi0, a.l[i], b.l[k - i], lo1, hi1, c.l[i], d.l[k - i]);
#pragma unroll
      for (int i = k - 7; i < 8; i++) mac96x2(lo0, hi0, m0[i], PRM::P[k - i], lo1, hi1, m1[i], PRM::P[k - i]);
      r0.l[k - 8] = (uint32_t)lo0;
      r1.l[k - 8] = (uint32_t)lo1;
      lo0 = (lo0 >> 32) | ((uint64_t)hi0 << 32);
      hi0 = 0;
      lo1 = (lo1 >> 32) | ((uint64_t)hi1 << 32);
      hi1 = 0;
    }
  }
  // (a^2, b^2): two dedicated squarings (see sqr()) in lockstep
  static ZK_DEV void sqr_pair(const Fp& a, const Fp& b, Fp& r0, Fp& r1) {
    uint32_t al[8], ad[8], bl[8], bd[8];
#pragma unroll
    for (int j = 1; j < 8; j++) {
      al[j] = a.l[j] << 1;
      ad[j] = __builtin_amdgcn_alignbit(a.l[j], a.l[j - 1], 31);
      bl[j] = b.l[j] << 1;
      bd[j] = __builtin_amdgcn_alignbit(b.l[j], b.l[j - 1], 31);
    }
    uint64_t lo0 = 0, lo1 = 0;
    uint32_t hi0 = 0, hi1 = 0, m0[8], m1[8];
#pragma unroll
    for (int k = 0; k < 8; k++) {
#pragma unroll
      for (int i = 0; 2 * i < k; i++)
        mac96x2(lo0, hi0, a.l[i], (k - i == i + 1) ? al[k - i] : ad[k - i], lo1, hi1, b.l[i], (k - i == i + 1) ? bl[k - i] : bd[k - i]);
      if ((k & 1) == 0) mac96x2(lo0, hi0, a.l[k / 2], a.l[k / 2], lo1, hi1, b.l[k / 2], b.l[k / 2]);
#pragma unroll
      for (int i = 0; i < k; i++) mac96x2(lo0, hi0, m0[i], PRM::P[k - i], lo1, hi1, m1[i], PRM::P[k - i]);
      m0[k] = (uint32_t)lo0 * PRM::INV;
      m1[k] = (uint32_t)lo1 * PRM::INV;
      mac96x2(lo0, hi0, m0[k], PRM::P[0], lo1, hi1, m1[k], PRM::P[0]);
      lo0 = (lo0 >> 32) | ((uint64_t)hi0 << 32);
      hi0 = 0;
      lo1 = (lo1 >> 32) | ((uint64_t)hi1 << 32);
      hi1 = 0;
    }
#pragma unroll
    for (int k = 8; k < 16; k++) {
#pragma unroll
      for (int i = k - 7; 2 * i < k; i++)
        mac96x2(lo0, hi0, a.l[i], (k - i == i + 1) ? al[k - i] : ad[k - i], lo1, hi1, b.l[i], (k - i == i + 1) ? bl[k - i] : bd[k - i]);
      if ((k & 1) == 0 && k / 2 < 8) mac96x2(lo0, hi0, a.l[k / 2], a.l[k / 2], lo1, hi1, b.l[k / 2], b.l[k / 2]);
#pragma unroll
      for (int i = k - 7; i < 8; i++) mac96x2(lo0, hi0, m0[i], PRM::P[k - i], lo1, hi1, m1[i], PRM::P[k - i]);
      r0.l[k - 8] = (uint32_t)lo0;
      r1.l[k - 8] = (uint32_t)lo1;
      lo0 = (lo0 >> 32) | ((uint64_t)hi0 << 32);
      hi0 = 0;
      lo1 = (lo1 >> 32) | ((uint64_t)hi1 << 32);
      hi1 = 0;
    }
  }
  ZK_DEV Fp neg_2p() const {  // 2p - a in [1, 2p] (no zero test): only as an operand of dot2
    Fp r;
    uint32_t bw = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) r.l[i] = subb(PRM::P2[i], l[i], bw);
    return r;
  }

  // standard form <-> Montgomery form
  ZK_DEV Fp to_mont() const { return (*this) * r2(); }
  ZK_DEV Fp from_mont() const {
    Fp o = zero();
    o.l[0] = 1;
    return (*this) * o;
  }

  // a^(p-2); exponent p-2 taken from PRM::P (rarely used on device: batch inversions)
  ZK_DEV Fp inv() const {
    Fp res = one();
    Fp base = *this;
    // e = p - 2
    uint32_t e[8];
#pragma unroll
    for (int i = 0; i < 8; i++) e[i] = PRM::P[i];
    e[0] -= 2;  // P[0] >= 2 for both primes
    for (int i = 0; i < 8; i++) {
      uint32_t w = e[i];
      for (int bit = 0; bit < 32; bit++) {
        if (w & 1) res = res * base;
        base = base.sqr();
        w >>= 1;
      }
    }
    return res;
  }
};

using Fq = Fp<FqParams>;
using Fr = Fp<FrParams>;

// ---- Fq2 = Fq[u]/(u^2+1) ---------------------------------------------------------------------
struct Fq2 {
  Fq c0, c1;
  static ZK_DEV Fq2 zero() { return {Fq::zero(), Fq::zero()}; }
  static ZK_DEV Fq2 one() { return {Fq::one(), Fq::zero()}; }
  ZK_DEV bool is_zero() const { return c0.is_zero() && c1.is_zero(); }
  ZK_DEV bool operator==(const Fq2& b) const { return c0 == b.c0 && c1 == b.c1; }
  ZK_DEV bool operator!=(const Fq2& b) const { return !(*this == b); }
  friend ZK_DEV Fq2 operator+(const Fq2& a, const Fq2& b) { return {a.c0 + b.c0, a.c1 + b.c1}; }
  friend ZK_DEV Fq2 operator-(const Fq2& a, const Fq2& b) { return {a.c0 - b.c0, a.c1 - b.c1}; }
  ZK_DEV Fq2 neg() const { return {c0.neg(), c1.neg()}; }
  ZK_DEV Fq2 dbl() const { return {c0.dbl(), c1.dbl()}; }
  // (a0 + a1 u)(b0 + b1 u) = (a0 b0 - a1 b1) + (a0 b1 + a1 b0) u, each coordinate ONE lazily reduced sum of two
  // products (Fq::dot2): 2 x (128 + 72) mad pairs and two conditional subtractions. The Karatsuba form it replaces
  // (3 x 136 pairs + 5 additions / subtractions of 24 instructions each) was 10 % more instructions.
  // r04: the two coordinates advance in lockstep (Fq::dot2_pair), which at the two waves per SIMD of the G2 kernels is
  // worth +6 % per product (tools/microbench4.hip); same instructions, three more registers.
  friend ZK_DEV Fq2 operator*(const Fq2& a, const Fq2& b) {
    Fq2 r;
    Fq::dot2_pair(a.c0, b.c0, a.c1, b.c1.neg_2p(), a.c0, b.c1, a.c1, b.c0, r.c0, r.c1);
    return r;
  }
  // (a0+a1 u)^2 = (a0+a1)(a0-a1) + 2 a0 a1 u: two independent products, in lockstep too
  ZK_DEV Fq2 sqr() const {
    Fq2 r;
    Fq t;
    Fq::mul_pair(c0 + c1, c0 - c1, c0, c1, r.c0, t);
    r.c1 = t + t;
    return r;
  }
  ZK_DEV Fq2 inv() const {
    Fq d = (c0.sqr() + c1.sqr()).inv();
    return {c0 * d, (c1 * d).neg()};
  }
};

template <class F>
struct FieldBytes;
template <>
struct FieldBytes<Fq> { static constexpr int N = 32; };
template <>
struct FieldBytes<Fr> { static constexpr int N = 32; };
template <>
struct FieldBytes<Fq2> { static constexpr int N = 64; };

// 16-byte vector loads/stores of field elements (coalescing sweet spot: 16 B per lane-instruction)
template <class PRM>
ZK_DEV Fp<PRM> load_fp(const void* p) {
  const uint4* q = reinterpret_cast<const uint4*>(p);
  uint4 a = q[0], b = q[1];
  Fp<PRM> r;
  r.l[0] = a.x; r.l[1] = a.y; r.l[2] = a.z; r.l[3] = a.w;
  r.l[4] = b.x; r.l[5] = b.y; r.l[6] = b.z; r.l[7] = b.w;
  return r;
}
template <class PRM>
ZK_DEV void store_fp(void* p, const Fp<PRM>& vin) {  // memory always holds the canonical value
  const Fp<PRM> v = vin.canon();
  uint4* q = reinterpret_cast<uint4*>(p);
  q[0] = make_uint4(v.l[0], v.l[1], v.l[2], v.l[3]);
  q[1] = make_uint4(v.l[4], v.l[5], v.l[6], v.l[7]);
}
ZK_DEV Fq2 load_fq2(const void* p) {
  return {load_fp<FqParams>(p), load_fp<FqParams>(reinterpret_cast<const char*>(p) + 32)};
}
ZK_DEV void store_fq2(void* p, const Fq2& v) {
  store_fp<FqParams>(p, v.c0);
  store_fp<FqParams>(reinterpret_cast<char*>(p) + 32, v.c1);
}

template <class F> ZK_DEV F load_field(const void* p);
template <> ZK_DEV Fq load_field<Fq>(const void* p) { return load_fp<FqParams>(p); }
template <> ZK_DEV Fr load_field<Fr>(const void* p) { return load_fp<FrParams>(p); }
template <> ZK_DEV Fq2 load_field<Fq2>(const void* p) { return load_fq2(p); }
ZK_DEV void store_field(void* p, const Fq& v) { store_fp<FqParams>(p, v); }
ZK_DEV void store_field(void* p, const Fr& v) { store_fp<FrParams>(p, v); }
ZK_DEV void store_field(void* p, const Fq2& v) { store_fq2(p, v); }

}  // namespace zkpoa
