// Bucket sort of the MSM's (point, window) entries: two-level MSD counting sort on LDS histograms.
//
// Replaces the first version's per-entry global atomics (one returning atomic + a 4-byte rank per
// entry, ~50 % of a 2^26 MSM). Every (window, bucket) list of point indices is produced with:
//   K0 digits   : scalar -> W signed digits (mag | sign<<31), written once, window-major.
//   level 1     : bucket id split as (coarse bin : fine key), fine = low FB bits.
//     K1        : task = (chunk of points, window): LDS histogram over coarse bins; ONE global atomic
//                 per (task, non-empty bin) reserves the task's slice of that bin  -> base1
//     scan      : bin totals -> bin offsets; ceil(size / CH2) -> level-2 task offsets
//     K3        : same tasks: LDS cursors = bin offset + base1; entries (idx|sign, fine) scattered
//                 to their bin (ranks from LDS atomics; wave-ballot aggregation of the leading key)
//   level 2     : task = (bin, split of <= CH2 entries): bins of any size stay load-balanced
//     K4a       : LDS histogram over fine keys; one global atomic per (task, key) into the bucket counts
//     scan      : bucket counts -> bucket offsets (off0), piece offsets
//     K4b       : LDS cursors = off0 + base2; point indices scattered into their bucket's list
// HBM traffic ~40 B per entry, all atomics except O(tasks x bins) of them are LDS atomics. Order inside a
// bucket is arbitrary, which the (commutative) bucket sum does not see.
#pragma once
// included from msm.hip.h after the scalar-recoding helpers, the u32 scan and find_bucket
#include "bn254_field.hip.h"

namespace zkpoa {

struct SortPlan {
  uint32_t n, W, c, Nb;
  uint32_t FB, F;        // fine bits / keys per bin
  uint32_t bins;         // coarse bins per window
  uint32_t SB;           // W * bins
  uint32_t CH1, chunks1; // level-1 task size / tasks per window
  uint32_t CH2;          // level-2 task size
  uint64_t tasks2_max;
};

inline SortPlan make_sort_plan(uint32_t n, uint32_t W, uint32_t c) {
  SortPlan s;
  s.n = n; s.W = W; s.c = c; s.Nb = 1u << (c - 1);
  uint32_t tb = c - 1;
  uint32_t cb = tb > 8 ? tb - 8 : 0;
  if (cb > 11) cb = 11;
  s.FB = tb - cb;
  s.F = 1u << s.FB;
  s.bins = 1u << cb;
  s.SB = W * s.bins;
  s.CH1 = n > (1u << 22) ? (1u << 16) : (1u << 14);
  s.chunks1 = n ? (n + s.CH1 - 1) / s.CH1 : 1;
  s.CH2 = 1u << 14;
  s.tasks2_max = ((uint64_t)n * W) / s.CH2 + s.SB + 1;
  return s;
}

// lds[key] += 1 for every valid lane; returns the lane's rank (value before its own increment).
// Lanes sharing the first valid lane's key are aggregated into one LDS atomic via a 64-bit ballot,
// which turns the hot-key case (witness bits: most lanes on one key) from a 64-way serialisation
// into a single atomic.
ZK_DEV uint32_t lds_count_rank(uint32_t* lds, uint32_t key, bool valid) {
  const uint32_t lane = threadIdx.x & 63u;
  unsigned long long vm = __ballot(valid);
  uint32_t rank = 0;
  bool done = !valid;
  if (vm) {
    int first = __ffsll((long long)vm) - 1;
    uint32_t k0 = __shfl(key, first);
    unsigned long long m = __ballot(valid && key == k0);
    if (valid && key == k0) {
      uint32_t base = 0;
      if ((int)lane == first) base = atomicAdd(&lds[k0], (uint32_t)__popcll(m));
      base = __shfl(base, first);
      rank = base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
      done = true;
    }
  }
  if (!done) rank = atomicAdd(&lds[key], 1u);
  return rank;
}

// ---- K0 ----------------------------------------------------------------------------------------------
static __global__ __launch_bounds__(256) void msm_digits_kernel(const void* __restrict__ scalars, uint32_t n,
                                                                uint32_t c, uint32_t W,
                                                                uint32_t* __restrict__ digits) {
  uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n) return;
  uint32_t s[8];
  load_scalar(scalars, i, s);
  bool neg = scalar_normalize(s);
  const uint32_t Nb = 1u << (c - 1), mask = (1u << c) - 1u;
  uint32_t carry = 0;
  for (uint32_t w = 0; w < W; w++) {
    uint32_t d = (s[0] & mask) + carry;
    scalar_shr(s, c);
    carry = d > Nb ? 1u : 0u;
    uint32_t mag = carry ? (mask + 1u - d) : d;
    bool sign = (carry != 0) != neg;
    digits[(size_t)w * n + i] = mag | ((sign && mag) ? 0x80000000u : 0u);
  }
}

// ---- level 1 ---------------------------------------------------------------------------------------------
constexpr uint32_t kMaxBins = 2048;

template <bool SCATTER>
static __global__ __launch_bounds__(256) void msm_sort_coarse_kernel(const uint32_t* __restrict__ digits, SortPlan sp,
                                                                     uint32_t* __restrict__ bin_cnt,
                                                                     const uint32_t* __restrict__ bin_off,
                                                                     uint32_t* __restrict__ base1,
                                                                     uint2* __restrict__ coarse) {
  __shared__ uint32_t h[kMaxBins];
  const uint32_t w = blockIdx.y, chunk = blockIdx.x, tid = threadIdx.x;
  const size_t task = (size_t)w * sp.chunks1 + chunk;
  for (uint32_t b = tid; b < sp.bins; b += 256u)
    h[b] = SCATTER ? bin_off[w * sp.bins + b] + base1[task * sp.bins + b] : 0u;
  __syncthreads();
  const uint32_t start = chunk * sp.CH1;
  const uint32_t end = (sp.n - start) < sp.CH1 ? sp.n : start + sp.CH1;
  const uint32_t fmask = sp.F - 1u;
  for (uint32_t off = 0; off < sp.CH1; off += 256u) {
    uint32_t i = start + off + tid;
    bool valid = i < end;
    uint32_t d = valid ? digits[(size_t)w * sp.n + i] : 0u;
    uint32_t mag = d & 0x7fffffffu;
    valid = valid && mag != 0;
    uint32_t key = valid ? (mag - 1u) >> sp.FB : 0u;
    uint32_t pos = lds_count_rank(h, key, valid);
    if (SCATTER && valid) coarse[pos] = make_uint2(i | (d & 0x80000000u), (mag - 1u) & fmask);
    if (start + off + 256u >= end) break;    // uniform: the remaining iterations have no valid lane
  }
  if (!SCATTER) {
    __syncthreads();
    for (uint32_t b = tid; b < sp.bins; b += 256u) {
      uint32_t cnt = h[b];
      base1[task * sp.bins + b] = cnt ? atomicAdd(&bin_cnt[w * sp.bins + b], cnt) : 0u;
    }
  }
}

// ---- level 2 ---------------------------------------------------------------------------------------------
constexpr uint32_t kMaxFine = 1024;

template <bool SCATTER>
static __global__ __launch_bounds__(256) void msm_sort_fine_kernel(const uint2* __restrict__ coarse, SortPlan sp,
                                                                   const uint32_t* __restrict__ bin_off,
                                                                   const uint32_t* __restrict__ tpo,
                                                                   uint32_t* __restrict__ cnt0,
                                                                   const uint32_t* __restrict__ off0,
                                                                   uint32_t* __restrict__ base2,
                                                                   uint32_t* __restrict__ sorted) {
  __shared__ uint32_t h[kMaxFine];
  const uint32_t t = blockIdx.x, tid = threadIdx.x;
  if (t >= tpo[sp.SB]) return;
  const uint32_t sb = find_bucket(tpo, sp.SB, t);
  const uint32_t j = t - tpo[sb];
  const uint32_t start = bin_off[sb] + j * sp.CH2;
  uint32_t end = bin_off[sb + 1];
  if (end - start > sp.CH2) end = start + sp.CH2;
  const size_t kb = (size_t)sb * sp.F;       // first bucket of this bin (global bucket index)
  for (uint32_t k = tid; k < sp.F; k += 256u) h[k] = SCATTER ? off0[kb + k] + base2[(size_t)t * sp.F + k] : 0u;
  __syncthreads();
  for (uint32_t off = 0; off < sp.CH2; off += 256u) {
    uint32_t e = start + off + tid;
    bool valid = e < end;
    uint2 v = valid ? coarse[e] : make_uint2(0u, 0u);
    uint32_t pos = lds_count_rank(h, v.y, valid);
    if (SCATTER && valid) sorted[pos] = v.x;
    if (start + off + 256u >= end) break;    // uniform: the remaining iterations have no valid lane
  }
  if (!SCATTER) {
    __syncthreads();
    for (uint32_t k = tid; k < sp.F; k += 256u) {
      uint32_t cnt = h[k];
      base2[(size_t)t * sp.F + k] = cnt ? atomicAdd(&cnt0[kb + k], cnt) : 0u;
    }
  }
}

}  // namespace zkpoa
