// Bucket sort of the MSM's (point, window) entries: MSD counting sort in passes of <= 8 key bits, every
// pass staged through LDS so that HBM only sees coalesced runs.
//
// Replaces the first version's per-entry global atomics (one returning atomic + a 4-byte rank per entry,
// ~50 % of a 2^26 MSM) and the second version's two unstaged scatter passes (2048 bins + 256 keys: every
// lane stored to its own cache line, ~100 G stores/s whatever the bin count, 25 of 108 ms at 2^26).
// Every (window, bucket) list of point indices is produced with:
//   K0 digits : scalar -> W signed digits (mag | sign<<31), written once, window-major.
//   pass l    : the bucket id (c-1 bits) is consumed from the top, bits[l] <= 8 bits per pass (2 passes up
//               to 16 bits, 3 up to 21: 7 + 6 + 6 at c = 20). Input of pass l is partitioned into segments (pass 0: one per window;
//               later: one per (window, key prefix)); a task is <= CH entries of one segment.
//     count   : LDS histogram of the task over the 2^bits[l] bins, ONE global atomic per (task, non-empty bin)
//               reserves the task's slice of the output segment -> base
//     scan    : segment totals -> output segment offsets, ceil(size / CH) -> task offsets of the next pass
//     scatter : per tile of 2048 entries: LDS histogram + ranks (wave-ballot aggregation of the leading key,
//               so a hot bucket costs one LDS atomic per wave), exclusive scan of the tile histogram, entries
//               written to LDS grouped by bin, then copied out with consecutive lanes on consecutive
//               addresses of each bin's run (avg >= 16 entries = 128-256 B per (tile, bin)).
//   The last pass writes the 4-byte point index | sign only; its segment offsets are the bucket offsets.
// HBM traffic ~12-16 B per entry and pass; all atomics except O(tasks x bins) of them are LDS atomics. Order
// inside a bucket is arbitrary, which the (commutative) bucket sum does not see.
#pragma once
// included from msm.hip.h after the scalar-recoding helpers, the u32 scan and find_bucket
#include "bn254_field.hip.h"

namespace zkpoa {

constexpr uint32_t kSortMaxBits = 8;                 // key bits per pass (one bin per thread of the workgroup)
constexpr uint32_t kSortMaxBins = 1u << kSortMaxBits;
constexpr uint32_t kSortEpt = 8;                     // entries per thread and tile
constexpr uint32_t kSortTile = 256u * kSortEpt;      // 2048 entries = 16 KiB of LDS staging
constexpr uint32_t kSortMaxPasses = 3;

struct SortPlan {
  uint32_t n, W, c, Nb;
  uint32_t npass;
  uint32_t bits[kSortMaxPasses];    // key bits consumed by pass l (from the top)
  uint32_t rem[kSortMaxPasses];     // key bits still in the entry when pass l starts (rem[0] = c - 1)
  uint32_t segs[kSortMaxPasses + 1];// segments entering pass l; segs[npass] = W * Nb buckets
  uint32_t CH;                      // task size (entries)
  uint32_t chunks0;                 // pass-0 tasks per window
  uint64_t tasks_max[kSortMaxPasses];
};

inline SortPlan make_sort_plan(uint32_t n, uint32_t W, uint32_t c) {
  SortPlan s{};
  s.n = n; s.W = W; s.c = c; s.Nb = 1u << (c - 1);
  const uint32_t tb = c - 1;
  s.npass = tb <= kSortMaxBits ? 1u : (tb + kSortMaxBits - 1) / kSortMaxBits;
  uint32_t left = tb;
  s.segs[0] = W;
  for (uint32_t l = 0; l < s.npass; l++) {
    uint32_t passes_left = s.npass - l;
    s.bits[l] = (left + passes_left - 1) / passes_left;
    s.rem[l] = left;
    left -= s.bits[l];
    s.segs[l + 1] = s.segs[l] << s.bits[l];
  }
  s.CH = 8u * kSortTile;   // 16384 entries per task
  s.chunks0 = n ? (n + s.CH - 1) / s.CH : 1;
  s.tasks_max[0] = (uint64_t)s.chunks0 * W;
  for (uint32_t l = 1; l < s.npass; l++) s.tasks_max[l] = ((uint64_t)n * W) / s.CH + s.segs[l] + 1;
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

// ---- K0', sparse scalars, fixed-base form -----------------------------------------------------------------
// A witness is mostly bits and short limbs: at the c = 17 of a witness table a scalar has ~3 non-zero digits of 15, and
// the dense digit array above (n x W words written once, read by pass 0's count and again by its scatter) is four
// fifths zeros. When the digit density says so (msm_sort_phase), the digits go straight into a COMPACT entry list
// instead -- (table index w * n + i | sign << 31, |digit| - 1), 8 bytes per non-zero digit, in no particular order (a
// bucket sum does not care) -- and pass 0 runs over that list exactly as the later passes run over theirs. One
// block-wide scan + one global atomic per block reserve the block's range of the list.
constexpr uint32_t kEntriesThreads = 1024, kEntriesPerThread = 2;          // 2048 scalars per workgroup
constexpr uint32_t kEntriesStage = 8192;                                    // entries a workgroup stages in LDS (64 KiB)
// signed c-bit digits of one sign-normalised scalar, lowest window first: f(w, magnitude, sign) for every non-zero one
template <class Fn>
ZK_DEV void for_nonzero_digits(uint32_t (&s)[8], bool neg, uint32_t c, uint32_t W, Fn f) {
  const uint32_t Nb = 1u << (c - 1), mask = (1u << c) - 1u;
  uint32_t carry = 0;
  for (uint32_t w = 0; w < W; w++) {
    const uint32_t d = (s[0] & mask) + carry;
    scalar_shr(s, c);
    carry = d > Nb ? 1u : 0u;
    const uint32_t mag = carry ? (mask + 1u - d) : d;
    if (mag) f(w, mag, (carry != 0) != neg);
  }
}
// One global atomic per WORKGROUP reserves its range of the list, so the workgroups are large: with 256-scalar
// workgroups the 70 k atomics of an 18 M-scalar query on one counter were the whole kernel (0.89 ms for 0.35 ms of bytes).
static __global__ __launch_bounds__(kEntriesThreads) void msm_entries_kernel(const void* __restrict__ scalars, uint32_t n,
                                                                            uint32_t c, uint32_t W, uint2* __restrict__ out,
                                                                            uint32_t* __restrict__ counter) {
  __shared__ uint32_t wave_tot[kEntriesThreads / 64];
  __shared__ uint32_t s_base, s_total;
  __shared__ uint2 stage[kEntriesStage];
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  uint32_t sc[kEntriesPerThread][8];
  bool neg[kEntriesPerThread];
  uint32_t idx[kEntriesPerThread], k = 0;
#pragma unroll
  for (uint32_t q = 0; q < kEntriesPerThread; q++) {
    idx[q] = (blockIdx.x * kEntriesPerThread + q) * kEntriesThreads + tid;
    neg[q] = false;
    if (idx[q] < n) {
      load_scalar(scalars, idx[q], sc[q]);
      neg[q] = scalar_normalize(sc[q]);
      uint32_t t[8];
#pragma unroll
      for (int j = 0; j < 8; j++) t[j] = sc[q][j];
      for_nonzero_digits(t, neg[q], c, W, [&](uint32_t, uint32_t, bool) { k++; });
    }
  }
  // exclusive scan of k over the workgroup
  uint32_t x = k;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t y = __shfl_up(x, o);
    if (lane >= (uint32_t)o) x += y;
  }
  if (lane == 63) wave_tot[wave] = x;
  __syncthreads();
  uint32_t before = 0, total = 0;
  for (uint32_t v = 0; v < kEntriesThreads / 64; v++) {
    const uint32_t t = wave_tot[v];
    if (v < wave) before += t;
    total += t;
  }
  uint32_t at = before + x - k;
  if (tid == 0) {
    s_total = total;
    s_base = total ? atomicAdd(counter, total) : 0u;
  }
  __syncthreads();
  const bool staged = s_total <= kEntriesStage;
  const uint32_t base = s_base;
#pragma unroll
  for (uint32_t q = 0; q < kEntriesPerThread; q++)
    if (idx[q] < n)
      for_nonzero_digits(sc[q], neg[q], c, W, [&](uint32_t w, uint32_t mag, bool sign) {
        const uint2 e = make_uint2((w * n + idx[q]) | (sign ? 0x80000000u : 0u), mag - 1u);
        if (staged) stage[at] = e;
        else out[base + at] = e;
        at++;
      });
  if (staged) {
    __syncthreads();
    for (uint32_t e = tid; e < s_total; e += kEntriesThreads) out[base + e] = stage[e];
  }
}
// the list as pass 0's one input segment: offsets {0, T}, task offsets {0, ceil(T / CH)}
static __global__ void msm_entries_finish_kernel(const uint32_t* __restrict__ counter, uint32_t CH, uint32_t* __restrict__ off,
                                                 uint32_t* __restrict__ tpo) {
  const uint32_t T = *counter;
  off[0] = 0;
  off[1] = T;
  tpo[0] = 0;
  tpo[1] = (T + CH - 1u) / CH;
}

// ---- one pass ------------------------------------------------------------------------------------------
// Task geometry. FIRST: task = (chunk blockIdx.x, window blockIdx.y) of the digit array; otherwise task
// t = blockIdx.x over the segments of `in_off` with task offsets `tpo`. Returns false when the block has no task.
struct SortTask {
  uint32_t seg, start, end;
  size_t id;
};

template <bool FIRST>
ZK_DEV bool sort_task(const SortPlan& sp, uint32_t pass, const uint32_t* __restrict__ in_off,
                      const uint32_t* __restrict__ tpo, SortTask& t) {
  if (FIRST) {
    t.seg = blockIdx.y;
    t.id = (size_t)blockIdx.y * sp.chunks0 + blockIdx.x;
    t.start = blockIdx.x * sp.CH;
    t.end = (sp.n - t.start) < sp.CH ? sp.n : t.start + sp.CH;   // indices inside the window's digit row
    return true;
  }
  const uint32_t S = sp.segs[pass];
  if (blockIdx.x >= tpo[S]) return false;
  t.seg = find_bucket(tpo, S, blockIdx.x);
  t.id = blockIdx.x;
  t.start = in_off[t.seg] + (blockIdx.x - tpo[t.seg]) * sp.CH;
  t.end = in_off[t.seg + 1];
  if (t.end - t.start > sp.CH) t.end = t.start + sp.CH;
  return true;
}

// entry e of the task's input -> (payload = point index | sign << 31, key = remaining bucket bits); false = skip
template <bool FIRST>
ZK_DEV bool sort_load(const SortPlan& sp, const uint32_t* __restrict__ digits, const uint2* __restrict__ in,
                      const SortTask& t, uint32_t e, uint32_t& payload, uint32_t& key) {
  payload = 0;
  key = 0;
  if (e >= t.end) return false;
  if (FIRST) {
    uint32_t d = digits[(size_t)t.seg * sp.n + e];
    uint32_t mag = d & 0x7fffffffu;
    payload = e | (d & 0x80000000u);
    key = mag - 1u;
    return mag != 0;
  }
  uint2 v = in[e];
  payload = v.x;
  key = v.y;
  return true;
}

// count: base[task][bin] = this task's offset inside output segment (seg, bin)
template <bool FIRST>
static __global__ __launch_bounds__(256) void msm_sort_count_kernel(SortPlan sp, uint32_t pass,
                                                                    const uint32_t* __restrict__ digits,
                                                                    const uint2* __restrict__ in,
                                                                    const uint32_t* __restrict__ in_off,
                                                                    const uint32_t* __restrict__ tpo,
                                                                    uint32_t* __restrict__ out_cnt,
                                                                    uint32_t* __restrict__ base) {
  __shared__ uint32_t h[kSortMaxBins];
  SortTask t;
  if (!sort_task<FIRST>(sp, pass, in_off, tpo, t)) return;
  const uint32_t tid = threadIdx.x, nb = 1u << sp.bits[pass], shift = sp.rem[pass] - sp.bits[pass];
  if (tid < nb) h[tid] = 0;
  __syncthreads();
  for (uint32_t off = t.start; off < t.end; off += 256u) {
    uint32_t payload, key;
    bool valid = sort_load<FIRST>(sp, digits, in, t, off + tid, payload, key);
    (void)lds_count_rank(h, key >> shift, valid);
  }
  __syncthreads();
  if (tid < nb) {
    uint32_t cnt = h[tid];
    base[t.id * nb + tid] = cnt ? atomicAdd(&out_cnt[(size_t)t.seg * nb + tid], cnt) : 0u;
  }
}

// scatter. LAST: the output is the 4-byte payload at its final (bucket-sorted) position, else (payload, key
// without the consumed bits) in the next pass's segment.
template <bool FIRST, bool LAST>
static __global__ __launch_bounds__(256) void msm_sort_scatter_kernel(SortPlan sp, uint32_t pass,
                                                                      const uint32_t* __restrict__ digits,
                                                                      const uint2* __restrict__ in,
                                                                      const uint32_t* __restrict__ in_off,
                                                                      const uint32_t* __restrict__ tpo,
                                                                      const uint32_t* __restrict__ out_off,
                                                                      const uint32_t* __restrict__ base,
                                                                      uint2* __restrict__ out,
                                                                      uint32_t* __restrict__ sorted) {
  __shared__ uint32_t hist[kSortMaxBins], toff[kSortMaxBins + 1], gpos[kSortMaxBins], cur[kSortMaxBins];
  __shared__ uint32_t wave_tot[4];
  __shared__ uint2 stage[kSortTile];
  SortTask t;
  if (!sort_task<FIRST>(sp, pass, in_off, tpo, t)) return;
  const uint32_t tid = threadIdx.x, nb = 1u << sp.bits[pass], shift = sp.rem[pass] - sp.bits[pass];
  const uint32_t keymask = (1u << shift) - 1u;
  static_assert(kSortMaxBins == 256, "one bin per thread");
  cur[tid] = tid < nb ? out_off[(size_t)t.seg * nb + tid] + base[t.id * nb + tid] : 0u;
  hist[tid] = 0;
  __syncthreads();
  for (uint32_t tile = t.start; tile < t.end; tile += kSortTile) {
    uint32_t payload[kSortEpt], key[kSortEpt], rank[kSortEpt];
    bool valid[kSortEpt];
#pragma unroll
    for (uint32_t e = 0; e < kSortEpt; e++) {
      valid[e] = sort_load<FIRST>(sp, digits, in, t, tile + e * 256u + tid, payload[e], key[e]);
      rank[e] = lds_count_rank(hist, key[e] >> shift, valid[e]);
    }
    __syncthreads();
    // exclusive scan of the tile histogram (one bin per thread, shuffle scan per wave), advance the run cursors
    {
      uint32_t v = hist[tid], x = v;
#pragma unroll
      for (uint32_t d = 1; d < 64u; d <<= 1) {
        uint32_t y = __shfl_up(x, d);
        if ((tid & 63u) >= d) x += y;
      }
      if ((tid & 63u) == 63u) wave_tot[tid >> 6] = x;
      toff[tid] = x - v;             // exclusive inside the wave
      gpos[tid] = cur[tid];
      cur[tid] += v;
      hist[tid] = 0;
    }
    __syncthreads();
    {
      uint32_t add = 0;
      for (uint32_t w = 0; w < (tid >> 6); w++) add += wave_tot[w];
      toff[tid] += add;
      if (tid == 255u) toff[kSortMaxBins] = add + wave_tot[3];
    }
    __syncthreads();
#pragma unroll
    for (uint32_t e = 0; e < kSortEpt; e++)
      if (valid[e]) stage[toff[key[e] >> shift] + rank[e]] = make_uint2(payload[e], key[e]);
    __syncthreads();
    const uint32_t total = toff[kSortMaxBins];
#pragma unroll
    for (uint32_t e = 0; e < kSortEpt; e++) {
      uint32_t s = e * 256u + tid;
      if (s < total) {
        uint2 v = stage[s];
        uint32_t b = v.y >> shift;
        uint32_t pos = gpos[b] + (s - toff[b]);
        if (LAST) sorted[pos] = v.x;
        else out[pos] = make_uint2(v.x, v.y & keymask);
      }
    }
    __syncthreads();
  }
}

}  // namespace zkpoa
