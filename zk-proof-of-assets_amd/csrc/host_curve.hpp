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

// Horner over the per-window sums of one MSM: result = sum_w 2^(c*w) * S[w]
template <class HF>
inline XYZZ<HF> h_combine_windows(const void* window_sums, uint32_t W, uint32_t c, uint32_t logS) {
  // window_sums: per window and group (0 = row sums, 1 = column sums) the logS + 1 per-bit totals T_b of the device's
  // bucket reduction; weighted total = sum_b 2^b T_b, window sum = 2^logS * U(rows) + V(columns)
  constexpr size_t X = 4 * HostBytes<HF>::N;
  const uint32_t nbits = logS + 1;
  const char* base = reinterpret_cast<const char*>(window_sums);
  auto weighted = [&](uint32_t group) {
    XYZZ<HF> t = XYZZ<HF>::inf();
    for (int b = (int)nbits - 1; b >= 0; b--) {
      t = xyzz_dbl(t);
      XYZZ<HF> tb = h_xyzz_from_bytes<HF>(base + ((size_t)group * nbits + (uint32_t)b) * X);
      xyzz_add(t, tb);
    }
    return t;
  };
  XYZZ<HF> acc = XYZZ<HF>::inf();
  for (int w = (int)W - 1; w >= 0; w--) {
    for (uint32_t k = 0; k < c; k++) acc = xyzz_dbl(acc);
    XYZZ<HF> u = weighted(2u * (uint32_t)w);
    for (uint32_t k = 0; k < logS; k++) u = xyzz_dbl(u);
    XYZZ<HF> v = weighted(2u * (uint32_t)w + 1u);
    xyzz_add(acc, u);
    xyzz_add(acc, v);
  }
  return acc;
}

}  // namespace zkpoa
