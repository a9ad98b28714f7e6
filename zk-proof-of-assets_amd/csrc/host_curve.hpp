// Host-side group helpers on top of host_field.hpp + the shared XYZZ templates: the O(1) tail
// of each MSM (window Horner), affine conversion, small scalar multiplications of the proof
// assembly. Product code; no MSM/NTT ever runs here.
#pragma once
#include "bn254_ec.hip.h"
#include "host_field.hpp"

namespace zkpoa {

template <class HF>
struct HostBytes;
template <>
struct HostBytes<HFq> { static constexpr size_t N = 32; };
template <>
struct HostBytes<HFq2> { static constexpr size_t N = 64; };

template <class HF>
inline Affine<HF> h_affine_from_bytes(const void* p) {
  const char* c = reinterpret_cast<const char*>(p);
  return {HF::from_bytes(c), HF::from_bytes(c + HostBytes<HF>::N)};
}
template <class HF>
inline void h_affine_to_bytes(const Affine<HF>& a, void* p) {
  char* c = reinterpret_cast<char*>(p);
  a.x.to_bytes(c);
  a.y.to_bytes(c + HostBytes<HF>::N);
}
template <class HF>
inline XYZZ<HF> h_xyzz_from_bytes(const void* p) {
  const char* c = reinterpret_cast<const char*>(p);
  constexpr size_t N = HostBytes<HF>::N;
  return {HF::from_bytes(c), HF::from_bytes(c + N), HF::from_bytes(c + 2 * N), HF::from_bytes(c + 3 * N)};
}
template <class HF>
inline Affine<HF> h_to_affine(const XYZZ<HF>& a) {
  if (a.is_inf()) return {HF::zero(), HF::zero()};
  // 1/ZZZ = i3;  1/ZZ = (ZZ * i3)^2   (ZZ^3 = ZZZ^2)
  HF i3 = a.zzz.inv();
  HF i2 = (a.zz * i3).sqr();
  return {a.x * i2, a.y * i3};
}

// k * P, k = 4 x u64 little-endian standard-form scalar
template <class HF>
inline XYZZ<HF> h_mul(const XYZZ<HF>& p, const uint64_t k[4]) {
  XYZZ<HF> r = XYZZ<HF>::inf();
  bool started = false;
  for (int i = 3; i >= 0; i--) {
    for (int b = 63; b >= 0; b--) {
      if (started) r = xyzz_dbl(r);
      if ((k[i] >> b) & 1) {
        xyzz_add(r, p);
        started = true;
      }
    }
  }
  return r;
}

// result = sum_w 2^(c*w) * S[w] from the device's bucket reduction: per window and group (0 = row sums, 1 = column
// sums) the logS + 1 per-bit totals T_b; weighted total = sum_b 2^b T_b, window sum S[w] = 2^logS * U(rows) + V(columns).
// Every total has one power of two, e = c*w + (rows ? logS : 0) + b: ONE Horner pass over e (<= 254 + 2 logS
// doublings and one addition per total; nested Horners per group and per window cost twice the doublings).
template <class HF>
inline XYZZ<HF> h_combine_windows(const void* window_sums, uint32_t W, uint32_t c, uint32_t logS) {
  constexpr size_t X = 4 * HostBytes<HF>::N;
  const uint32_t nbits = logS + 1;
  const char* base = reinterpret_cast<const char*>(window_sums);
  const uint32_t top = c * (W ? W - 1 : 0) + 2 * logS;   // largest exponent that occurs
  XYZZ<HF> acc = XYZZ<HF>::inf();
  for (int e = (int)top; e >= 0; e--) {
    acc = xyzz_dbl(acc);
    // totals with exponent e: window w, rows with b = e - c*w - logS, columns with b = e - c*w
    for (uint32_t w = 0; w < W && c * w <= (uint32_t)e; w++) {
      const uint32_t rel = (uint32_t)e - c * w;
      if (rel < nbits) {
        XYZZ<HF> t = h_xyzz_from_bytes<HF>(base + ((size_t)(2 * w + 1) * nbits + rel) * X);
        xyzz_add(acc, t);
      }
      if (rel >= logS && rel - logS < nbits) {
        XYZZ<HF> t = h_xyzz_from_bytes<HF>(base + ((size_t)(2 * w) * nbits + (rel - logS)) * X);
        xyzz_add(acc, t);
      }
    }
  }
  return acc;
}

}  // namespace zkpoa
