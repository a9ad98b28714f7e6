// Pippenger multi-scalar multiplication over BN254 G1 / G2 for gfx950, generic over the
// coordinate field F (Fq -> G1, Fq2 -> G2).
//
// Replaces G1.multiExpAffine / G2.multiExpAffine of snarkjs (groth16_prove.js, five call sites:
// SURVEY.md 3.2 step 5) and rapidsnark's ParallelMultiexp, the step
// that runs behind the reference call site scripts/g16_prove.sh:248-252.
//
// Inputs are exactly the reference's wire formats: bases affine / Montgomery / 64 B (G1) or
// 128 B (G2) as in .zkey sections 5-9, scalars 32 B little-endian standard form as in .wtns.
//
// Pipeline (all kernels on one stream; one 16-byte read-back to size the launches):
//   1 digits   : scalar -> sign-normalised (s > r/2 -> r-s, negate base) signed c-bit digits.
//   2 sort     : MSD counting sort of the (point, window) entries by bucket in passes of <= 8 key bits, every
//                pass staged through LDS (msm_sort.hip.h): per-task LDS histogram, one global atomic per
//                (task, bin) to reserve ranges, scan, LDS-regrouped coalesced scatter. The hot key of real
//                witnesses (|digit| = 1, i.e. scalars 0/1/-1) is aggregated per wavefront with a 64-bit ballot.
//   3 scan     : exclusive scans (bucket offsets, piece offsets).
//   4 accumulate (level 0): buckets are cut into pieces of <= K0 entries, pieces are ordered by length
//                (longest first) and one thread sums one piece with XYZZ mixed additions (gather of
//                64/128-B affine bases). Work per thread is bounded whatever the scalar distribution;
//                buckets with more than one piece are finished by further levels over the partial sums.
//   5 reduce   : sum_b (b+1)*B[b] per bucket set from the row and column sums of the rows x S bucket matrix
//                (2 independent additions per bucket), small double-and-add weights, LDS tree.
//   6 host     : per bucket set 2^logS * U + V, then Horner over 2^c across the windows (O(W*c) group ops).
// Fixed-base form (MsmTable, below): with 2^(c*j) * P_i precomputed for every window, all windows share one
// bucket set and the same kernels run over n * W entries.
#pragma once
#include "bn254_ec.hip.h"
#include "device_ctx.hpp"
#include <string.h>
#include <chrono>
#include <utility>

namespace zkpoa {

constexpr uint32_t kMaxPieceLenPlan = 512;   // = kMaxPieceLen of the piece-ordering kernels
struct MsmPlan {
  uint32_t n = 0;   // points
  uint32_t c = 0;   // window bits
  uint32_t W = 0;   // digit windows per scalar
  uint32_t Nb = 0;  // buckets per bucket set = 2^(c-1)
  // Fixed-base form (MsmTable below): the bases 2^(c*j) * P_i of every window j are precomputed, so the digits of
  // ALL windows fall into ONE bucket set: Wb = 1 and the (point, window) entry w * n + i is itself the index into
  // the table. Classic form: one bucket set per window, Wb = W.
  bool merged = false;
  uint32_t Wb = 0;  // bucket sets
  uint32_t ne = 0;  // entries per bucket set: n (classic) or n * W (merged)
  uint32_t TB = 0;  // total buckets = Wb * Nb
  uint32_t K0 = 0;  // level-0 piece length
  uint32_t K = 4;   // fan-in per thread for levels >= 1: hot buckets (witness bits) are latency-bound chains of full
                    // additions, so a small fan-in with more levels beats 64 sequential additions per thread
  // bucket reduction: the Nb buckets of a window form a rows x S matrix, b = hi * S + lo
  uint32_t logS = 0, logRows = 0;
};

inline uint32_t msm_windows(uint32_t c) { return (254 + c - 1) / c; }
// Density hint of the calling thread (the prover's witness stages set it): density[c] = expected non-zero digits per
// scalar at window width c, 4 <= c <= 25, measured on this proof's witness (msm_density_kernel). Real witnesses are
// mostly bits and short limbs, so a witness MSM has a fraction of the n * W entries of a uniform one -- and a window
// sized for n * W entries then pays for millions of near-empty buckets (2^26 shape: the A query took 34 ms for 10 ms
// of additions). nullptr = uniform scalars (every digit non-zero).
constexpr int kDensityLo = 4, kDensityHi = 25;
inline const double*& msm_density_hint() {
  static thread_local const double* hint = nullptr;
  return hint;
}
// experiments (zkpoa_set_option "msm_k0"): force the level-0 piece length; 0 = the rule below
inline int& msm_forced_k0() {
  static int k0 = 0;
  return k0;
}

// Window width. Costs in units of one mixed addition: every (point, window) entry is one; a bucket costs ~3 in the
// reduction (a row and a column full addition at 1.4 each + tree levels); one counting-sort pass moves ~20 B per
// entry at ~3 TB/s, i.e. ~0.03 of an addition per entry and pass (measured per-kernel times at 2^20, r02).
inline MsmPlan msm_make_plan(size_t n, int force_c = 0, bool for_g2 = false, bool merged = false) {
  MsmPlan p;
  p.n = (uint32_t)n;
  p.merged = merged;
  int best_c = 4;
  double best = 1e300;
  const double* density = msm_density_hint();   // merged form: the width is the table's (force_c) at run time, but a
                                                // table for witness scalars is SIZED with the density of a witness
  for (int c = 4; c <= (merged ? 25 : 22); c++) {
    double W = (double)msm_windows((uint32_t)c);
    double sets = merged ? 1.0 : W;
    double passes = (double)((c - 1 + 7) / 8);
    // entries that exist; the dense n x W digit matrix is still written once and read twice (0.04 of an addition
    // per slot), and every bucket costs a slot in the piece list besides its two additions in the reduction
    double entries = density ? density[c] * (double)n : W * (double)n;
    double cost = entries * (1.0 + 0.03 * passes) + (density ? 0.04 * W * (double)n : 0.0) +
                  (density ? 5.0 : 3.0) * sets * (double)(1u << (c - 1));
    if (cost < best) {
      best = cost;
      best_c = c;
    }
  }
  if (force_c >= 4 && force_c <= (merged ? 25 : 22)) best_c = force_c;
  p.c = best_c;
  p.W = msm_windows(p.c);
  p.Nb = 1u << (p.c - 1);
  p.Wb = merged ? 1u : p.W;
  p.ne = merged ? p.n * p.W : p.n;
  p.TB = p.Wb * p.Nb;
  p.logS = (uint32_t)(p.c - 1 + 1) / 2;
  p.logRows = (uint32_t)(p.c - 1) - p.logS;
  // Level-0 piece length. One thread adds one piece sequentially: about twice the mean bucket occupancy, so
  // most buckets are a single piece (uniform scalars: 128 at 2^20, 256 at 2^26). The kernel cannot end before
  // K0 dependent mixed additions have run, and a G2 addition is ~15 us for a lone wave (256 of them: 4 ms), so
  // a sort whose result also feeds a G2 accumulation (the prover's B query) uses a quarter of that; buckets
  // longer than K0 continue in the partial-sum levels.
  const double dens_entries = density ? density[p.c] * (double)p.n : (double)p.n * p.W;
  double avg = (merged ? dens_entries : dens_entries / p.W) / (double)p.Nb;
  uint32_t k0 = 32;
  while (k0 < 2 * avg && k0 < 256) k0 <<= 1;
  if (for_g2 && k0 > 32) k0 = k0 >= 128 ? k0 / 4 : 32;
  // ... and short enough that the pieces fill the chip about 2.5 times over (256 CUs x 12 waves x 64 lanes at the
  // accumulation kernel's 3 waves per SIMD; 2 for G2): with fewer, longer pieces the grid is one partial wave of
  // work and the CUs idle behind its longest pieces (2^20 points: 128-long pieces ran at 0.84 of the ALU ceiling,
  // 32-long ones at 0.95-1.0; the extra partial sums cost ~3 % more additions).
  const double fill = dens_entries / (2.5 * 256.0 * (for_g2 ? 8.0 : 12.0) * 64.0);
  while (k0 > 32 && (double)k0 > fill) k0 >>= 1;
  if (msm_forced_k0() >= 8 && msm_forced_k0() <= (int)kMaxPieceLenPlan) k0 = (uint32_t)msm_forced_k0();
  p.K0 = k0;
  return p;
}

// ---- scalar recoding ------------------------------------------------------------------------
// (r-1)/2 and r, 8 x u32 little-endian
__device__ static const uint32_t kHalfR[8] = {0xf8000000u, 0xa1f0fac9u, 0x3cdcb848u, 0x9419f424u,
                                              0x40c0ac2eu, 0xdc2822dbu, 0x7098d014u, 0x18322739u};

// s <- min(s, r - s); returns true when r - s was taken (the base must then be negated).
ZK_DEV bool scalar_normalize(uint32_t (&s)[8]) {
  uint32_t bw = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) (void)subb(kHalfR[i], s[i], bw);
  if (!bw) return false;  // s <= (r-1)/2
  uint32_t b2 = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) s[i] = subb(FrParams::P[i], s[i], b2);
  return true;
}

ZK_DEV void scalar_shr(uint32_t (&s)[8], uint32_t c) {
#pragma unroll
  for (int i = 0; i < 7; i++) s[i] = __builtin_amdgcn_alignbit(s[i + 1], s[i], c);
  s[7] >>= c;
}

ZK_DEV void load_scalar(const void* scalars, uint32_t i, uint32_t (&s)[8]) {
  const uint4* p = reinterpret_cast<const uint4*>(scalars) + 2 * (size_t)i;
  uint4 a = p[0], b = p[1];
  s[0] = a.x; s[1] = a.y; s[2] = a.z; s[3] = a.w;
  s[4] = b.x; s[5] = b.y; s[6] = b.z; s[7] = b.w;
}

// non-zero c-bit digits of the sign-normalised scalars, summed over the array, for every c in [kDensityLo, kDensityHi]:
// ~ceil(bitlength / c) each. out[c - kDensityLo] (u64, zeroed by the caller).
static __global__ __launch_bounds__(256) void msm_density_kernel(const void* __restrict__ scalars, uint64_t n,
                                                                 unsigned long long* __restrict__ out) {
  __shared__ unsigned int acc[kDensityHi - kDensityLo + 1];
  if (threadIdx.x <= (unsigned)(kDensityHi - kDensityLo)) acc[threadIdx.x] = 0;
  __syncthreads();
  uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
  uint32_t L = 0;
  if (i < n) {
    uint32_t s[8];
    const uint4* p = reinterpret_cast<const uint4*>(scalars) + 2 * i;
    uint4 a = p[0], b = p[1];
    s[0] = a.x; s[1] = a.y; s[2] = a.z; s[3] = a.w;
    s[4] = b.x; s[5] = b.y; s[6] = b.z; s[7] = b.w;
    (void)scalar_normalize(s);
#pragma unroll
    for (int k = 7; k >= 0; k--)
      if (L == 0 && s[k]) L = 32u * k + (32u - __builtin_clz(s[k]));
  }
  for (int c = kDensityLo; c <= kDensityHi; c++) {
    uint32_t d = (L + c - 1) / c;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) d += __shfl_xor(d, o);
    if ((threadIdx.x & 63u) == 0 && d) atomicAdd(&acc[c - kDensityLo], d);
  }
  __syncthreads();
  if (threadIdx.x <= (unsigned)(kDensityHi - kDensityLo) && acc[threadIdx.x])
    atomicAdd(&out[threadIdx.x], (unsigned long long)acc[threadIdx.x]);
}

// ---- 2: scan (u32, exclusive, n+1 outputs) ----------------------------------------------------
// mode 0: v = in[i];  mode 1: v = ceil(in[i]/K);
// mode 3: `in` is an offsets array with n+1 entries: x = in[i+1]-in[i]; v = x <= 1 ? 0 : ceil(x/K)
ZK_DEV uint32_t scan_input(const uint32_t* __restrict__ in, uint32_t idx, uint32_t n, int mode, uint32_t K,
                           uint32_t& raw) {
  if (idx >= n) {
    raw = 0;
    return 0;
  }
  uint32_t x = in[idx];
  if (mode == 3) x = in[idx + 1] - x;
  raw = x;
  if (mode == 0) return x;
  if (mode == 3 && x <= 1) return 0;
  return (x + K - 1) / K;
}
constexpr int kScanBlock = 256;
constexpr int kScanItems = 8;
constexpr int kScanTile = kScanBlock * kScanItems;

ZK_DEV uint32_t block_exclusive_scan(uint32_t v, uint32_t* lds, uint32_t& total) {
  // wave scan via shuffles, then 4 wave totals through LDS
  uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  uint32_t x = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    uint32_t y = __shfl_up(x, o);
    if (lane >= (uint32_t)o) x += y;
  }
  if (lane == 63) lds[wave] = x;
  __syncthreads();
  uint32_t base = 0, tot = 0;
#pragma unroll
  for (int k = 0; k < kScanBlock / 64; k++) {
    uint32_t t = lds[k];
    if ((uint32_t)k < wave) base += t;
    tot += t;
  }
  __syncthreads();
  total = tot;
  return base + x - v;
}

static __global__ __launch_bounds__(kScanBlock) void scan_reduce_kernel(const uint32_t* __restrict__ in, uint32_t n, int mode,
                                                                 uint32_t K, uint32_t* __restrict__ block_sums,
                                                                 uint32_t* __restrict__ max_out) {
  __shared__ uint32_t lds[8];
  uint32_t base = blockIdx.x * kScanTile + threadIdx.x * kScanItems;
  uint32_t sum = 0, mx = 0;
#pragma unroll
  for (int k = 0; k < kScanItems; k++) {
    uint32_t x;
    sum += scan_input(in, base + k, n, mode, K, x);
    mx = x > mx ? x : mx;
  }
  uint32_t total;
  (void)block_exclusive_scan(sum, lds, total);
  if (threadIdx.x == 0) block_sums[blockIdx.x] = total;
  if (max_out) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      uint32_t y = __shfl_xor(mx, o);
      mx = y > mx ? y : mx;
    }
    if ((threadIdx.x & 63u) == 0 && mx) atomicMax(max_out, mx);
  }
}

// single block: exclusive scan of block_sums in place, total -> *total_out
static __global__ __launch_bounds__(kScanBlock) void scan_spine_kernel(uint32_t* __restrict__ block_sums, uint32_t nblocks,
                                                                uint32_t* __restrict__ total_out) {
  __shared__ uint32_t lds[8];
  uint32_t running = 0;
  for (uint32_t start = 0; start < nblocks; start += kScanBlock) {
    uint32_t idx = start + threadIdx.x;
    uint32_t v = idx < nblocks ? block_sums[idx] : 0;
    uint32_t total;
    uint32_t ex = block_exclusive_scan(v, lds, total);
    if (idx < nblocks) block_sums[idx] = running + ex;
    running += total;
  }
  if (threadIdx.x == 0) *total_out = running;
}

static __global__ __launch_bounds__(kScanBlock) void scan_apply_kernel(const uint32_t* __restrict__ in, uint32_t n, int mode,
                                                                uint32_t K, const uint32_t* __restrict__ block_sums,
                                                                const uint32_t* __restrict__ total,
                                                                uint32_t* __restrict__ out) {
  __shared__ uint32_t lds[8];
  uint32_t base = blockIdx.x * kScanTile + threadIdx.x * kScanItems;
  uint32_t v[kScanItems];
  uint32_t sum = 0;
#pragma unroll
  for (int k = 0; k < kScanItems; k++) {
    uint32_t raw;
    v[k] = scan_input(in, base + k, n, mode, K, raw);
    sum += v[k];
  }
  uint32_t tot;
  uint32_t ex = block_exclusive_scan(sum, lds, tot) + block_sums[blockIdx.x];
#pragma unroll
  for (int k = 0; k < kScanItems; k++) {
    uint32_t idx = base + k;
    if (idx < n) out[idx] = ex;
    ex += v[k];
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) out[n] = *total;
}

// out[0..n] = exclusive scan of transform(in[0..n)); out[n] = total (also in *total_dev).
inline void scan_u32(hipStream_t st, const uint32_t* in, uint32_t n, int mode, uint32_t K, uint32_t* out,
                     uint32_t* block_sums, uint32_t* total_dev, uint32_t* max_dev) {
  uint32_t nblocks = (n + kScanTile - 1) / kScanTile;
  hipLaunchKernelGGL(scan_reduce_kernel, dim3(nblocks), dim3(kScanBlock), 0, st, in, n, mode, K, block_sums, max_dev);
  hipLaunchKernelGGL(scan_spine_kernel, dim3(1), dim3(kScanBlock), 0, st, block_sums, nblocks, total_dev);
  hipLaunchKernelGGL(scan_apply_kernel, dim3(nblocks), dim3(kScanBlock), 0, st, in, n, mode, K, block_sums, total_dev,
                     out);
}

// ---- 2b: the same scans in ONE launch (decoupled look-back), two at a time ------------------------------------
// The MSM's hot path runs its scans in pairs over one input (bucket counts -> entry offsets and piece offsets; segment
// counts -> entry offsets and task offsets) and each scan_u32 is three launches: 12 of the ~45 launches of a 2^20 MSM,
// every one a few microseconds of kernel plus a dependency bubble on a lane whose proof is latency-shaped (VERDICT r02:
// 2205 launches of scan_reduce / spine / apply in a 26-proof profile). scan_pair_kernel does both scans (and the
// optional maximum) in a single pass: a workgroup takes the next tile from a ticket counter (tiles start in index
// order, so a tile never waits for one that has not been scheduled), publishes its aggregates, and a whole wavefront
// looks back over the published prefixes of its predecessors, 64 tiles per step. Status words carry the call's
// generation number, so the scratch is never cleared between calls; the last workgroup to finish rewinds the tickets.
struct ScanPair {
  const uint32_t* in;
  uint32_t n;
  int mode_a;
  uint32_t K_a;
  uint32_t* out_a;      // n + 1 entries
  uint32_t* total_a;    // optional
  int mode_b;           // < 0: no second scan
  uint32_t K_b;
  uint32_t* out_b;
  uint32_t* total_b;
  uint32_t* max_out;    // optional: atomicMax of the raw inputs
  // optional: histogram of the level-0 piece lengths the raw inputs (bucket counts) imply -- a bucket of x entries is
  // x / hist_K pieces of hist_K entries and one of x % hist_K: hist[len] += pieces of that length (len <= hist_K <= 512).
  // The piece-ordering kernel needs it complete before it runs, and the scan reads every bucket count anyway.
  uint32_t* hist = nullptr;
  uint32_t hist_K = 0;
};

ZK_DEV uint64_t scan_status_load(const uint64_t* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
ZK_DEV void scan_status_store(uint64_t* p, uint64_t v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// exclusive prefix of tile `tile` for one of the two scans, by the first wavefront of the workgroup: publishes the
// tile's aggregate, walks back 64 predecessors at a time until one with an inclusive prefix is met, then publishes
// its own inclusive prefix. status: [ntiles] words of (generation << 2 | state) << 32 | value; state 1 = aggregate,
// 2 = inclusive prefix.
// guard: a word that is never published (host-side ordering mistake, wiped scratch) ends the wait after poll_limit
// polls: the lane raises the fault word in the lane's pinned host memory, takes the missing prefix as zero and goes on,
// so every wavefront still reaches the end of the kernel; the host turns the fault word into an error (msm_read_back).
struct ScanGuard {
  uint32_t poll_limit;
  uint32_t withhold;     // tests only: tile 0 does not publish
  uint32_t* fault;       // host-pinned, device-addressable
};
ZK_DEV uint32_t scan_look_back(uint64_t* status, uint32_t tile, uint32_t aggregate, uint32_t gen, const ScanGuard& guard) {
  const uint32_t lane = threadIdx.x & 63u;
  const uint64_t tag1 = ((uint64_t)((gen << 2) | 1u)) << 32, tag2 = ((uint64_t)((gen << 2) | 2u)) << 32;
  if (tile == 0) {
    if (lane == 0 && !guard.withhold) scan_status_store(&status[0], tag2 | aggregate);
    return 0;
  }
  if (lane == 0) scan_status_store(&status[tile], tag1 | aggregate);
  uint32_t running = 0;
  int64_t top = (int64_t)tile - 1;   // highest tile not yet folded in
  for (;;) {
    const int64_t t = top - (int64_t)lane;
    uint64_t w = 0;
    uint32_t state = 2, val = 0;     // lanes before tile 0: an inclusive prefix of nothing
    if (t >= 0) {
      uint32_t polls = 0;
      bool gave_up = false;
      do {
        w = scan_status_load(&status[t]);
        if (++polls > guard.poll_limit) {
          gave_up = true;
          break;
        }
      } while ((uint32_t)(w >> 34) != gen || (((uint32_t)(w >> 32)) & 3u) == 0);
      state = ((uint32_t)(w >> 32)) & 3u;
      val = (uint32_t)w;
      if (gave_up) {   // counts as "prefix of nothing": the walk ends here, the result is wrong and flagged as such
        __hip_atomic_store(guard.fault, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        state = 2;
        val = 0;
      }
    }
    const uint64_t incl = __ballot(state == 2);          // non-zero: lanes below tile 0 count too
    const uint32_t stop = incl ? (uint32_t)__builtin_ctzll(incl) : 63u;   // nearest predecessor that knows its whole prefix
    uint32_t x = lane <= stop ? val : 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o);
    running += x;
    if (incl) break;
    top -= 64;
  }
  if (lane == 0) scan_status_store(&status[tile], tag2 | (running + aggregate));
  return running;
}

static __global__ __launch_bounds__(kScanBlock) void scan_pair_kernel(ScanPair a, uint64_t* __restrict__ status_a,
                                                                      uint64_t* __restrict__ status_b,
                                                                      uint32_t* __restrict__ tickets, uint32_t gen,
                                                                      uint32_t ntiles, ScanGuard guard) {
  __shared__ uint32_t lds[8];
  __shared__ uint32_t s_tile, s_pre_a, s_pre_b;
  __shared__ uint32_t s_hist[kMaxPieceLenPlan + 1];
  if (threadIdx.x == 0) s_tile = atomicAdd(&tickets[0], 1u);
  if (a.hist)
    for (uint32_t i = threadIdx.x; i <= a.hist_K; i += kScanBlock) s_hist[i] = 0;
  __syncthreads();
  const uint32_t tile = s_tile;
  const uint32_t base = tile * kScanTile + threadIdx.x * kScanItems;
  const bool two = a.mode_b >= 0;
  uint32_t va[kScanItems], vb[kScanItems];
  uint32_t sum_a = 0, sum_b = 0, mx = 0;
#pragma unroll
  for (int k = 0; k < kScanItems; k++) {
    uint32_t raw;
    va[k] = scan_input(a.in, base + k, a.n, a.mode_a, a.K_a, raw);
    sum_a += va[k];
    mx = raw > mx ? raw : mx;
    if (a.hist && raw) {
      const uint32_t full = raw / a.hist_K, rem = raw - full * a.hist_K;
      if (full) atomicAdd(&s_hist[a.hist_K], full);
      if (rem) atomicAdd(&s_hist[rem], 1u);
    }
    vb[k] = 0;
    if (two) {
      if (a.mode_b == 0) vb[k] = raw;
      else if (a.mode_b == 3 && raw <= 1) vb[k] = 0;
      else vb[k] = (raw + a.K_b - 1) / a.K_b;
      sum_b += vb[k];
    }
  }
  uint32_t tot_a, tot_b = 0;
  uint32_t ex_a = block_exclusive_scan(sum_a, lds, tot_a);
  uint32_t ex_b = two ? block_exclusive_scan(sum_b, lds, tot_b) : 0;
  if (threadIdx.x < 64) {
    uint32_t pa = scan_look_back(status_a, tile, tot_a, gen, guard);
    uint32_t pb = two ? scan_look_back(status_b, tile, tot_b, gen, guard) : 0;
    if (threadIdx.x == 0) {
      s_pre_a = pa;
      s_pre_b = pb;
    }
  }
  __syncthreads();
  ex_a += s_pre_a;
  ex_b += s_pre_b;
#pragma unroll
  for (int k = 0; k < kScanItems; k++) {
    const uint32_t idx = base + k;
    if (idx < a.n) {
      a.out_a[idx] = ex_a;
      if (two) a.out_b[idx] = ex_b;
    }
    ex_a += va[k];
    ex_b += vb[k];
  }
  if (tile == ntiles - 1 && threadIdx.x == kScanBlock - 1) {   // the last thread of the last tile holds the totals
    a.out_a[a.n] = ex_a;
    if (a.total_a) *a.total_a = ex_a;
    if (two) {
      a.out_b[a.n] = ex_b;
      if (a.total_b) *a.total_b = ex_b;
    }
  }
  if (a.max_out) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      uint32_t y = __shfl_xor(mx, o);
      mx = y > mx ? y : mx;
    }
    if ((threadIdx.x & 63u) == 0 && mx) atomicMax(a.max_out, mx);
  }
  if (a.hist) {   // (the barrier of the block scans above has ordered every LDS atomic before this)
    for (uint32_t i = threadIdx.x; i <= a.hist_K; i += kScanBlock) {
      const uint32_t c = s_hist[i];
      if (c) atomicAdd(&a.hist[i], c);
    }
  }
  // rewind the tickets for the next call (stream order: it cannot start before this grid has drained)
  if (threadIdx.x == 0 && atomicAdd(&tickets[1], 1u) == ntiles - 1) {
    tickets[0] = 0;
    tickets[1] = 0;
  }
}

inline bool scan_three_kernel() {
  static const bool v = [] {
    const char* e = getenv("ZKPOA_SCAN");
    return e && !strcmp(e, "3");
  }();
  return v;
}

// Per-lane scratch of scan_pair (device_ctx.hpp Lane::scan_*): two status arrays + the ticket counters, zeroed when
// (re)allocated only; gen counts the calls on the lane.
inline void scan_pair(Lane& lane, const ScanPair& a) {
  const uint32_t ntiles = (a.n + kScanTile - 1) / kScanTile;
  if (ntiles == 0) {   // n == 0: the n + 1 outputs are the empty sums
    ZK_HIP(hipMemsetAsync(a.out_a, 0, 4, lane.stream));
    if (a.total_a) ZK_HIP(hipMemsetAsync(a.total_a, 0, 4, lane.stream));
    if (a.mode_b >= 0) {
      ZK_HIP(hipMemsetAsync(a.out_b, 0, 4, lane.stream));
      if (a.total_b) ZK_HIP(hipMemsetAsync(a.total_b, 0, 4, lane.stream));
    }
    return;
  }
  if (lane.scan_tiles < ntiles || lane.scan_gen >= 0x3ffffff0u) {
    ZK_HIP(hipStreamSynchronize(lane.stream));
    if (lane.scan_scratch) ZK_HIP(hipFree(lane.scan_scratch));
    lane.scan_scratch = nullptr;
    const uint32_t cap = ntiles < 1024 ? 1024 : ntiles * 2;
    const size_t bytes = (size_t)cap * 16 + 256;
    ZK_HIP(hipMalloc(&lane.scan_scratch, bytes));
    // on the lane's own stream: a null-stream hipMemset is not ordered against a non-blocking stream, and a clear that
    // lands after the first scan has started would wipe live tickets and status words
    ZK_HIP(hipMemsetAsync(lane.scan_scratch, 0, bytes, lane.stream));
    lane.scan_tiles = cap;
    lane.scan_gen = 0;
  }
  const uint32_t gen = ++lane.scan_gen;   // generation 0 never appears in a live status word
  uint32_t* tickets = reinterpret_cast<uint32_t*>(lane.scan_scratch);
  uint64_t* st_a = reinterpret_cast<uint64_t*>(reinterpret_cast<char*>(lane.scan_scratch) + 256);
  uint64_t* st_b = st_a + lane.scan_tiles;
  const ScanGuard guard{lane.scan_poll_limit, lane.scan_test_withhold, lane.fault_word_dev()};
  hipLaunchKernelGGL(scan_pair_kernel, dim3(ntiles), dim3(kScanBlock), 0, lane.stream, a, st_a, st_b, tickets, gen, ntiles,
                     guard);
}

// After a synchronisation of the lane's stream: did a scan of this lane give up waiting? (The word is cleared, so the
// lane is usable again -- the caller decides what the failure means for the context.)
inline void lane_check_fault(Lane& lane) {
  volatile uint32_t* f = lane.fault_word();
  if (*f) {
    *f = 0;
    throw HipError("msm: a scan look-back gave up waiting for a tile prefix that was never published (result discarded)");
  }
}

// ---- 4: accumulate --------------------------------------------------------------------------
// largest b in [0, nb) with po[b] <= t  (po has nb+1 entries, non-decreasing, po[nb] > t)
ZK_DEV uint32_t find_bucket(const uint32_t* __restrict__ po, uint32_t nb, uint32_t t) {
  uint32_t lo = 0, hi = nb;  // invariant: po[lo] <= t < po[hi]
  while (hi - lo > 1) {
    uint32_t mid = (lo + hi) >> 1;
    if (po[mid] <= t) lo = mid; else hi = mid;
  }
  return lo;
}

}  // namespace zkpoa
#include "msm_sort.hip.h"   // 1 + 3: digits and the two-level LDS bucket sort
namespace zkpoa {

// ---- 4a: order the pieces by length, longest first ----------------------------------------------
// Piece lengths follow the bucket-count distribution (Poisson for uniform scalars), so lanes of one
// wave would idle behind its longest piece and the last workgroups of the grid would run alone.
// A counting sort of piece ids by length (<= K0 distinct keys; LDS histogram per workgroup, one
// global atomic per (workgroup, length)) gives every wave equal-length pieces and schedules the
// longest first.
constexpr uint32_t kMaxPieceLen = 512;

// order[pos] = t with the pieces sorted by descending length, pbkt[t] = the piece's bucket. The histogram of the
// lengths comes from the scan that produced the piece offsets (ScanPair::hist). Every thread takes kPiecesPerThread
// pieces (r03 took one, in two kernels: 100 k workgroups of a 2^26 MSM each reserved its slice of ~64 length classes
// with one global atomic per class -- 6 M atomics on 64 addresses, 4.1 ms of an 87 ms MSM -- and the bucket of every
// piece was searched twice).
constexpr uint32_t kPiecesPerThread = 8;
// the piece-length histogram on its own (the three-launch scans of ZKPOA_SCAN=3 do not carry it)
static __global__ __launch_bounds__(256) void msm_piece_hist_kernel(const uint32_t* __restrict__ cnt0, uint32_t TB,
                                                                    uint32_t K0, uint32_t* __restrict__ len_hist) {
  __shared__ uint32_t h[kMaxPieceLen + 1];
  for (uint32_t i = threadIdx.x; i <= K0; i += 256u) h[i] = 0;
  __syncthreads();
  const uint32_t b = blockIdx.x * 256u + threadIdx.x;
  const uint32_t x = b < TB ? cnt0[b] : 0u;
  if (x) {
    const uint32_t full = x / K0, rem = x - full * K0;
    if (full) atomicAdd(&h[K0], full);
    if (rem) atomicAdd(&h[rem], 1u);
  }
  __syncthreads();
  for (uint32_t i = threadIdx.x; i <= K0; i += 256u)
    if (h[i]) atomicAdd(&len_hist[i], h[i]);
}
static __global__ __launch_bounds__(256) void msm_piece_order_kernel(const uint32_t* __restrict__ cnt0,
                                                                     const uint32_t* __restrict__ po1, uint32_t TB,
                                                                     uint32_t K0, const uint32_t* __restrict__ len_hist,
                                                                     uint32_t* __restrict__ cursor,
                                                                     uint32_t* __restrict__ pbkt,
                                                                     uint32_t* __restrict__ order) {
  __shared__ uint32_t base[kMaxPieceLen + 2];   // start of each length class in descending order
  __shared__ uint32_t h[kMaxPieceLen + 1];      // local count, then reserved global start
  __shared__ uint32_t wsum[4];
  // descending exclusive prefix over lengths: base[L] = sum_{l > L} hist[l]
  // (513 entries: three per thread, block scan over reversed index)
  uint32_t tid = threadIdx.x;
  uint32_t v0 = 0, v1 = 0, v2 = 0;
  uint32_t r0 = 3 * tid, r1 = 3 * tid + 1, r2 = 3 * tid + 2;   // reversed positions: length = kMax - r
  if (r0 <= kMaxPieceLen) v0 = len_hist[kMaxPieceLen - r0];
  if (r1 <= kMaxPieceLen) v1 = len_hist[kMaxPieceLen - r1];
  if (r2 <= kMaxPieceLen) v2 = len_hist[kMaxPieceLen - r2];
  uint32_t tot;
  uint32_t ex = block_exclusive_scan(v0 + v1 + v2, wsum, tot);
  if (r0 <= kMaxPieceLen) base[kMaxPieceLen - r0] = ex;
  if (r1 <= kMaxPieceLen) base[kMaxPieceLen - r1] = ex + v0;
  if (r2 <= kMaxPieceLen) base[kMaxPieceLen - r2] = ex + v0 + v1;
  for (uint32_t i = tid; i <= kMaxPieceLen; i += 256u) h[i] = 0;
  __syncthreads();
  const uint32_t total = po1[TB];
  uint32_t len[kPiecesPerThread], local[kPiecesPerThread];
#pragma unroll
  for (uint32_t k = 0; k < kPiecesPerThread; k++) {
    const uint32_t t = (blockIdx.x * kPiecesPerThread + k) * 256u + tid;
    len[k] = 0;
    local[k] = 0;
    if (t < total) {
      const uint32_t b = find_bucket(po1, TB, t);
      const uint32_t j = t - po1[b];
      uint32_t l = cnt0[b] - j * K0;
      if (l > K0) l = K0;
      pbkt[t] = b;
      len[k] = l;
      local[k] = atomicAdd(&h[l], 1u);
    }
  }
  __syncthreads();
  for (uint32_t i = tid; i <= kMaxPieceLen; i += 256u) {
    uint32_t c = h[i];
    if (c) h[i] = atomicAdd(&cursor[i], c);
  }
  __syncthreads();
#pragma unroll
  for (uint32_t k = 0; k < kPiecesPerThread; k++) {
    const uint32_t t = (blockIdx.x * kPiecesPerThread + k) * 256u + tid;
    if (t < total) order[base[len[k]] + h[len[k]] + local[k]] = t;
  }
}

// waves per SIMD the accumulation kernel is compiled for (register budget 512 / waves):
// G1: 132 VGPRs -> 3 waves (4 waves = 128 VGPRs measured no faster: the kernel is at the ALU ceiling); G2: 256 -> 2
template <class F> struct AccumWaves { static constexpr int N = 3; };
template <> struct AccumWaves<Fq2> { static constexpr int N = 2; };

// level 0: items are sorted point indices; piece j of bucket b covers entries
// [off0[b] + j*K0, min(off0[b] + cnt0[b], +K0)). Threads take pieces in `order` (longest first).
template <class F>
constexpr bool kAccumPrefetch = FieldBytes<F>::N <= 32;
template <class F>
static __global__ __launch_bounds__(256, AccumWaves<F>::N) void msm_accum0_kernel(const void* __restrict__ bases,
                                                         const uint32_t* __restrict__ sorted,
                                                         const uint32_t* __restrict__ cnt0,
                                                         const uint32_t* __restrict__ off0,
                                                         const uint32_t* __restrict__ po1,
                                                         const uint32_t* __restrict__ order,
                                                         const uint32_t* __restrict__ pbkt, uint32_t TB, uint32_t K0,
                                                         void* __restrict__ buckets, void* __restrict__ P1) {
  uint32_t slot = blockIdx.x * 256u + threadIdx.x;
  uint32_t total = po1[TB];
  if (slot >= total) return;
  uint32_t t = order[slot];
  uint32_t b = pbkt[t];
  uint32_t j = t - po1[b];
  uint32_t cnt = cnt0[b];
  uint32_t start = off0[b] + j * K0;
  uint32_t end = off0[b] + cnt;
  if (end - start > K0) end = start + K0;
  XYZZ<F> acc = XYZZ<F>::inf();
  // The piece's entry indices are read FOUR at a time (one aligned 16-byte load per four additions). Every lane walks
  // its own run of `sorted`, so a 4-byte read per addition touched the lane's 64-byte line sixteen separate times,
  // ~5 us apart, with the base gathers of 200 k other lanes streaming through the L2 in between: the line was fetched
  // again most of the time (r02 PMC: 1.63 GB fetched per 2^20 launch against 1.07 GB of bases + 0.07 GB of indices,
  // unchanged with four times longer pieces -- so it was the walk, not the per-piece metadata; 1.31 GB with this).
  // The words of a block that lie outside [start, end) belong to neighbouring pieces and are simply not used; `sorted`
  // is 256-byte aligned and padded (Arena::take).
  const uint4* sorted4 = reinterpret_cast<const uint4*>(sorted);
  uint32_t blk = start >> 2;
  uint4 cache = FieldBytes<F>::N > 32 ? make_uint4(0, 0, 0, 0) : sorted4[blk];
  auto entry = [&](uint32_t k) -> uint32_t {
    // G2: the accumulation already needs every one of the 256 VGPRs two waves per SIMD leave it; the five registers
    // of the block cache would go to scratch, and a G2 addition is long enough for the line to be worth re-reading
    if (FieldBytes<F>::N > 32) return sorted[k];
    if ((k >> 2) != blk) {
      blk = k >> 2;
      cache = sorted4[blk];
    }
    const uint32_t i = k & 3u;
    return i == 0u ? cache.x : (i == 1u ? cache.y : (i == 2u ? cache.z : cache.w));
  };
  if (kAccumPrefetch<F>) {
    uint32_t e = entry(start);
    Affine<F> p = load_affine<F>(bases, e & 0x7fffffffu);
    for (uint32_t k = start; k < end; k++) {
      uint32_t e_cur = e;
      Affine<F> p_cur = p;
      // prefetch the next base under this addition (r04: TWO bases in flight measured no better -- 1.256 vs 1.208 ms at
      // 2^20, 1.095 vs 1.075 ms through a 0.9 GB table: the gathers are not what the kernel waits for)
      if (k + 1 < end) {
        e = entry(k + 1);
        p = load_affine<F>(bases, e & 0x7fffffffu);
      }
      xyzz_add_affine(acc, p_cur, (e_cur >> 31) != 0);
    }
  } else {   // G2: no room for a second 128-byte point in registers; the other wave of the SIMD covers the gather
    for (uint32_t k = start; k < end; k++) {
      const uint32_t e = entry(k);
      const Affine<F> p = load_affine<F>(bases, e & 0x7fffffffu);
      xyzz_add_affine(acc, p, (e >> 31) != 0);
    }
  }
  uint32_t pieces = (cnt + K0 - 1) / K0;
  if (pieces == 1) store_xyzz(buckets, b, acc);
  else store_xyzz(P1, t, acc);
}

// level l >= 1: items are XYZZ partial sums. Input items of bucket b: pin = pc_in[b] > 1 ? pc_in[b] : 0,
// stored at Pin[po_in[b] ...]; output pieces enumerated by po_out.
template <class F>
static __global__ __launch_bounds__(256) void msm_accumN_kernel(const void* __restrict__ Pin,
                                                         const uint32_t* __restrict__ po_in,
                                                         const uint32_t* __restrict__ po_out, uint32_t TB, uint32_t K,
                                                         void* __restrict__ buckets, void* __restrict__ Pout) {
  uint32_t t = blockIdx.x * 256u + threadIdx.x;
  uint32_t total = po_out[TB];
  if (t >= total) return;
  uint32_t b = find_bucket(po_out, TB, t);
  uint32_t j = t - po_out[b];
  uint32_t cnt = po_in[b + 1] - po_in[b];  // > 1 here, or this bucket would have no output piece
  uint32_t start = po_in[b] + j * K;
  uint32_t end = po_in[b] + cnt;
  if (end - start > K) end = start + K;
  XYZZ<F> acc = load_xyzz<F>(Pin, start);
  for (uint32_t k = start + 1; k < end; k++) {
    XYZZ<F> q = load_xyzz<F>(Pin, k);
    xyzz_add(acc, q);
  }
  uint32_t pieces = po_out[b + 1] - po_out[b];
  if (pieces == 1) store_xyzz(buckets, b, acc);
  else store_xyzz(Pout, t, acc);
}

// ---- 5: bucket reduction ----------------------------------------------------------------------
// Window sum = sum_b (b + 1) * B_b. With b = hi * S + lo (rows x S matrix of the window's buckets):
//   sum_b (b + 1) B_b = S * sum_hi hi * R_hi + sum_lo (lo + 1) * C_lo,   R_hi = row sums, C_lo = column sums.
// All 2 * Nb additions of the row / column sums are independent: 2^logParts threads per sum add `per` (1-16)
// buckets each, then an LDS tree over the parts -- enough threads to fill the chip whatever the bucket count (a
// fixed-base MSM at c = 20 has 2^19 buckets: 1536 sums x 64 parts), every thread a short chain. The weighted totals
// U = sum hi R_hi, V = sum (lo+1) C_lo are NOT formed by multiplying every sum by its weight (a double-and-add of up
// to logS + 1 bits per thread: 2 (logS + 1) dependent group operations, ~10 us each on a lone wave): for every bit
// b of the weight one LDS tree adds the sums whose weight has that bit (msm_bit_tree_sum_kernel, all bits in one
// launch) and the host forms sum_b 2^b T_b -- half the group operations, and a dependent chain of 8 per 256 sums.
// With one bucket set (fixed-base form) the whole reduction is latency: a single 2^21 witness MSM spent 0.62 ms in
// it (sums 0.30 + weights 0.18 + trees 0.15) next to 0.46 ms of bucket accumulation; chains of <= 2 + 7 (sums, more
// parts when the launch would not fill the chip) and 8 (trees) bring that to ~0.2 ms.
// Every kernel below has ONE inlined call site per group operation (a loop whose operand comes from HBM first and
// from LDS afterwards): no scratch memory, ~140 VGPRs, so these waves co-reside with the accumulation kernel's.
// out[(w * 2 + grp) * E + idx]: grp 0 = R (idx = hi < rows), grp 1 = C (idx = lo < S); E >= max(rows, S).
constexpr uint32_t kReduceMaxLogParts = 8;   // one sum per 256-thread workgroup at most
constexpr uint32_t kReduceMaxBits = 16;      // weight bits (logS + 1 <= 14 for c <= 26)
inline uint32_t msm_reduce_log_parts(uint32_t logRows, uint32_t logS = 0, uint32_t sets = 0) {
  uint32_t lp = logRows > 3 ? logRows - 3 : 0;   // per = 8 column terms (16 row terms when S = 2 * rows)
  if (lp < 4) lp = logRows < 4 ? logRows : 4;
  if (lp > kReduceMaxLogParts) lp = kReduceMaxLogParts;
  // few bucket sets (fixed-base form: one): shorter chains until the launch has 2^16 threads
  const uint32_t cap = logRows < kReduceMaxLogParts ? logRows : kReduceMaxLogParts;
  while (sets && lp < cap && (((uint64_t)sets * ((1ull << logRows) + (1ull << logS))) << lp) < (1ull << 16)) lp++;
  return lp;
}
inline uint32_t msm_reduce_bits(uint32_t logS) { return logS + 1; }   // column weights go up to S = 2^logS

template <class F>
static __global__ __launch_bounds__(256) void msm_bucket_sums_kernel(const void* __restrict__ buckets,
                                                                     const uint32_t* __restrict__ counts, uint32_t W,
                                                                     uint32_t Nb, uint32_t logS, uint32_t logRows,
                                                                     uint32_t logParts, uint32_t E,
                                                                     void* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) char lds_raw[];
  const uint32_t gid = blockIdx.x * 256u + threadIdx.x;
  const uint32_t parts = 1u << logParts, part = gid & (parts - 1u), o = gid >> logParts;
  const uint32_t S = 1u << logS, rows = 1u << logRows, per_window = rows + S;
  const uint32_t w = o / per_window, r = o % per_window;
  const bool valid = w < W;
  const bool is_row = r < rows;
  const uint32_t idx = is_row ? r : r - rows;
  const uint32_t len = is_row ? S : rows, stride = is_row ? 1u : S;
  const size_t base = (size_t)w * Nb + (is_row ? (size_t)idx * S : (size_t)idx);
  const uint32_t per = len >> logParts;   // logParts <= min(logS, logRows)
  const uint32_t per_max = S >> logParts; // uniform trip count (S >= rows)
  XYZZ<F> acc = XYZZ<F>::inf();
  // iterations [0, per_max): this thread's buckets from HBM; then logParts tree levels through LDS.
  // (Tried and dropped, r02: handing each tree level's additions to the first threads of the workgroup so that idle
  // lanes form whole waves that skip the addition -- fewer wave-additions, same depth: no gain alone or six in flight.)
  for (uint32_t it = 0; it < per_max + logParts; it++) {
    XYZZ<F> other = XYZZ<F>::inf();
    if (it < per_max) {
      // an empty bucket was never written (the array is not cleared: 3.2 GB per 2^26 MSM): its count says so
      if (valid && it < per) {
        const size_t bi = base + (size_t)(part * per + it) * stride;
        if (counts[bi]) other = load_xyzz<F>(buckets, bi);
      }
    } else {
      const uint32_t st = parts >> (it - per_max + 1u);
      store_xyzz(lds_raw, threadIdx.x, acc);
      __syncthreads();
      if (part < st) other = load_xyzz<F>(lds_raw, threadIdx.x + st);
      __syncthreads();
    }
    xyzz_add(acc, other);
  }
  if (valid && part == 0) store_xyzz(out, ((size_t)w * 2u + (is_row ? 0u : 1u)) * E + idx, acc);
}

// Y[(g * nbits + b) * S_out + blockIdx.x] = sum of the X[g * E + idx], idx in this block's 256, whose weight has bit b
// (g = w * 2 + grp; weight = idx for row sums, idx + 1 for column sums, 0 for padding entries).
// ONE wave per workgroup: each lane adds its 4 entries, then 6 LDS tree levels (a 256-thread workgroup would keep
// three of its four waves parked at the barriers of the tree, each holding ~140 VGPRs).
template <class F>
static __global__ __launch_bounds__(64) void msm_bit_tree_sum_kernel(const void* __restrict__ X, uint32_t E,
                                                                     uint32_t logS, uint32_t logRows, uint32_t nbits,
                                                                     uint32_t S_out, void* __restrict__ Y) {
  extern __shared__ __attribute__((aligned(16))) char lds_raw[];
  const uint32_t g = blockIdx.y / nbits, b = blockIdx.y % nbits, grp = g & 1u;
  const uint32_t cnt = grp ? (1u << logS) : (1u << logRows);
  XYZZ<F> v = XYZZ<F>::inf();
  // iterations 0..3: this lane's entries from HBM; 4..9: tree levels through LDS (one inlined addition site)
  for (uint32_t it = 0; it < 10; it++) {
    XYZZ<F> o = XYZZ<F>::inf();
    if (it < 4) {
      const uint32_t idx = blockIdx.x * 256u + it * 64u + threadIdx.x;
      const uint32_t k = idx < cnt ? (grp ? idx + 1u : idx) : 0u;
      if ((k >> b) & 1u) o = load_xyzz<F>(X, (size_t)g * E + idx);
    } else {
      const uint32_t stride = 32u >> (it - 4);
      store_xyzz(lds_raw, threadIdx.x, v);
      __syncthreads();
      if (threadIdx.x < stride) o = load_xyzz<F>(lds_raw, threadIdx.x + stride);
      __syncthreads();
    }
    xyzz_add(v, o);
  }
  if (threadIdx.x == 0) store_xyzz(Y, (size_t)blockIdx.y * S_out + blockIdx.x, v);
}

// Y[w][blockIdx.x] = sum of X[w][blockIdx.x*256 .. +256) (S entries per window); same one-wave shape.
template <class F>
static __global__ __launch_bounds__(64) void msm_tree_sum_kernel(const void* __restrict__ X, uint32_t S, uint32_t S_out,
                                                           void* __restrict__ Y) {
  extern __shared__ __attribute__((aligned(16))) char lds_raw[];
  const uint32_t w = blockIdx.y;
  XYZZ<F> v = XYZZ<F>::inf();
  for (uint32_t it = 0; it < 10; it++) {
    XYZZ<F> o = XYZZ<F>::inf();
    if (it < 4) {
      const uint32_t idx = blockIdx.x * 256u + it * 64u + threadIdx.x;
      if (idx < S) o = load_xyzz<F>(X, (size_t)w * S + idx);
    } else {
      const uint32_t stride = 32u >> (it - 4);
      store_xyzz(lds_raw, threadIdx.x, v);
      __syncthreads();
      if (threadIdx.x < stride) o = load_xyzz<F>(lds_raw, threadIdx.x + stride);
      __syncthreads();
    }
    xyzz_add(v, o);
  }
  if (threadIdx.x == 0) store_xyzz(Y, (size_t)w * S_out + blockIdx.x, v);
}

// ---- host driver ------------------------------------------------------------------------------
template <class F>
struct MsmSizes {
  static constexpr size_t kAffine = 2 * FieldBytes<F>::N;
  static constexpr size_t kXyzz = 4 * FieldBytes<F>::N;
};

// Read-back of a few bytes to KB: a kernel stores them into the lane's pinned host buffer. (hipMemcpyAsync device ->
// host goes through the SDMA engine: measured ~0.15 ms per MSM for the 37 KB of per-bit totals with six MSMs in
// flight -- 8 % of the 2^20 rate -- and HSA_ENABLE_SDMA=0 is not ours to set for the caller's process.)
static __global__ __launch_bounds__(256) void msm_to_host_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst,
                                                                 uint32_t n16) {
  for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n16; i += gridDim.x * 256u) dst[i] = src[i];
}
inline void msm_read_back(Lane& lane, const void* d_src, size_t bytes) {
  if (bytes > lane.pinned_cap) throw HipError("msm: read-back larger than the pinned staging area");
  const uint32_t n16 = (uint32_t)((bytes + 15) / 16);
  const uint32_t grid = n16 > 4096 ? 16u : (n16 + 255) / 256;
  hipLaunchKernelGGL(msm_to_host_kernel, dim3(grid ? grid : 1), dim3(256), 0, lane.stream, (const uint4*)d_src,
                     (uint4*)lane.pinned_dev, n16);
  ZK_HIP(hipStreamSynchronize(lane.stream));
  lane_check_fault(lane);
}

// ---- host driver: two phases ---------------------------------------------------------------------
// Phase A (scalars only): digits, bucket sort, scans, piece ordering. Its result can serve several
// accumulations over DIFFERENT base arrays with the SAME scalars -- the prover's A, B1 and B2 queries all
// use the witness -- so those three MSMs sort once.
// Phase B (bases): bucket accumulation, further levels, bucket reduction, window sums to the host.
struct MsmSorted {
  MsmPlan p;
  uint32_t* counts = nullptr;   // entries per bucket
  uint32_t* off0 = nullptr;     // bucket offsets into `sorted`
  uint32_t* po_a = nullptr;     // level-0 piece offsets
  uint32_t* sorted = nullptr;   // point index | sign << 31, grouped by bucket
  uint32_t* order = nullptr;    // piece ids, longest first
  uint32_t* pbkt = nullptr;     // piece -> bucket
  uint32_t total1 = 0;          // level-0 pieces
  uint32_t total_entries = 0;   // (point, window) entries with a non-zero digit = mixed additions of the level-0 kernel
  uint32_t max_count = 0;       // largest bucket
};

inline size_t al256(size_t b) { return (b + 255) & ~size_t(255); }

inline size_t msm_sort_workspace_bytes(const MsmPlan& p) {
  size_t T = (size_t)p.n * p.W;
  size_t p1 = T / p.K0 + p.TB + 1;
  SortPlan sp = make_sort_plan(p.ne, p.Wb, p.c);
  size_t bytes = 0;
  bytes += al256((size_t)p.TB * 4);                  // counts
  bytes += al256(((size_t)p.TB + 1) * 4) * 2;        // off0, po_a
  bytes += al256(T * 4) + al256(T * 8);              // sorted; digits (T words) or the compact entry list (T pairs)
  if (sp.npass > 1) bytes += al256(T * 8);           // entries between passes (ping)
  if (sp.npass > 2) bytes += al256(T * 8);           // (pong)
  for (uint32_t l = 0; l < sp.npass; l++) {
    if (l + 1 < sp.npass) bytes += al256(((size_t)sp.segs[l + 1] + 1) * 4) * 3;   // seg_cnt, seg_off, tpo
    bytes += al256((size_t)sp.tasks_max[l] * ((size_t)1 << sp.bits[l]) * 4);       // base
  }
  bytes += al256(((size_t)p.TB / kScanTile + 2) * 4);
  bytes += al256(64);
  bytes += al256(p1 * 4) * 2;                        // pbkt, order
  bytes += al256((kMaxPieceLen + 1) * 4 * 2);        // len_hist, cursor
  return bytes + (1 << 12);
}

template <class F>
inline size_t msm_accum_workspace_bytes(const MsmPlan& p) {
  size_t T = (size_t)p.n * p.W;
  size_t p1 = T / p.K0 + p.TB + 1;
  size_t p2 = p1 / 2 + 1;
  size_t E = (size_t)1 << p.logS;   // logS >= logRows
  size_t bytes = 0;
  bytes += al256(((size_t)p.TB + 1) * 4) * 2;        // po_b, po_c
  bytes += al256(((size_t)p.TB / kScanTile + 2) * 4);
  bytes += al256(64);
  bytes += al256((size_t)p.TB * MsmSizes<F>::kXyzz);  // buckets
  bytes += al256(p1 * MsmSizes<F>::kXyzz);            // P1
  bytes += al256(p2 * MsmSizes<F>::kXyzz);            // P2
  bytes += al256(2 * p.Wb * E * MsmSizes<F>::kXyzz);                    // X: row / column sums
  bytes += al256(2 * p.Wb * msm_reduce_bits(p.logS) * ((E + 255) / 256) * MsmSizes<F>::kXyzz) * 2;  // tree-sum levels
  return bytes + (1 << 12);
}

inline void lane_reserve(Lane& lane, size_t need) {
  if (lane.ws.cap < need) {
    ZK_HIP(hipStreamSynchronize(lane.stream));
    const auto t0 = std::chrono::steady_clock::now();
    lane.ws.reserve(need);
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (ms > 20.0 && getenv("ZKPOA_VERBOSE"))   // (only when a workspace grows: first proof on a key, or after precompute)
      fprintf(stderr, "zkpoa:   lane workspace grows to %.2f GB: hipFree + hipMalloc %.1f ms\n", need / 1e9, ms);
  }
}

// Phase A on lane.stream. `extra_bytes` is reserved on top (for a following phase B on the same lane).
// The arena is reset here; the result stays valid until the lane's next msm_sort_phase. With
// sync_at_end the stream is synchronised on return, so other lanes may read the result.
inline MsmSorted msm_sort_phase(Lane& lane, const void* d_scalars, size_t n, int force_c,
                                size_t (*extra)(const MsmPlan&), bool sync_at_end = false, bool for_g2 = false,
                                bool merged = false) {
  MsmSorted sr;
  sr.p = msm_make_plan(n, force_c, for_g2, merged);
  const MsmPlan& p = sr.p;
  // entry positions are 32-bit: n * W (+ a few per-bucket slots) must stay below 2^32; a merged entry index also
  // carries the sign in bit 31
  if ((uint64_t)p.n * p.W + p.TB >= 0xffff0000ull) throw HipError("msm: n * windows exceeds the 32-bit entry index");
  if (merged && (uint64_t)p.n * p.W >= 0x80000000ull) throw HipError("msm: n * windows exceeds the fixed-base entry index");
  hipStream_t st = lane.stream;
  lane_reserve(lane, msm_sort_workspace_bytes(p) + (extra ? extra(p) : 0));
  Arena& ws = lane.ws;
  ws.reset();
  const size_t T_max = (size_t)p.n * p.W;
  const SortPlan sp = make_sort_plan(p.ne, p.Wb, p.c);
  // everything that must start at zero sits in ONE block at the head of the arena and is cleared by one memset (r03:
  // four, each a launch of its own on a lane whose MSM is ~45 launches): bucket counts, the segment counts of the
  // intermediate passes, the misc words, the piece-length histogram and its cursors
  char* zero_lo = ws.base + ws.off;
  sr.counts = ws.take<uint32_t>(p.TB);
  uint32_t* seg_cnt[kSortMaxPasses] = {};
  for (uint32_t l = 0; l + 1 < sp.npass; l++) seg_cnt[l] = ws.take<uint32_t>(sp.segs[l + 1] + 1);
  uint32_t* misc = ws.take<uint32_t>(16);  // [0]=T, [1]=max count, [2]=total pieces, ...
  uint32_t* len_hist = ws.take<uint32_t>(2 * (kMaxPieceLen + 1));
  uint32_t* len_cursor = len_hist + (kMaxPieceLen + 1);
  const size_t zero_bytes = (size_t)((ws.base + ws.off) - zero_lo);
  sr.off0 = ws.take<uint32_t>(p.TB + 1);
  sr.po_a = ws.take<uint32_t>(p.TB + 1);
  // sparse scalars in the fixed-base form (a witness through its tables): a compact entry list instead of the dense
  // digit array (msm_sort.hip.h msm_entries_kernel); ZKPOA_NO_SPARSE=1 keeps the dense form (measurement)
  static const bool no_sparse = [] {
    const char* e = getenv("ZKPOA_NO_SPARSE");
    return e && *e == '1';
  }();
  const double* dens = msm_density_hint();
  const bool sparse = merged && p.n && dens && dens[p.c] < 0.6 * (double)p.W && !no_sparse;
  uint32_t* digits = sparse ? nullptr : ws.take<uint32_t>(T_max);
  uint2* elist = sparse ? ws.take<uint2>(T_max) : nullptr;
  uint32_t* pre = sparse ? ws.take<uint32_t>(8) : nullptr;   // [0..1] offsets, [4..5] task offsets of pass 0's one segment
  uint2* ebuf[2] = {sp.npass > 1 ? ws.take<uint2>(T_max) : nullptr, sp.npass > 2 ? ws.take<uint2>(T_max) : nullptr};
  // per pass: output segment counts / offsets / task offsets (the last pass writes the bucket arrays) and bases
  uint32_t *seg_off[kSortMaxPasses] = {}, *seg_tpo[kSortMaxPasses] = {}, *base[kSortMaxPasses] = {};
  for (uint32_t l = 0; l < sp.npass; l++) {
    if (l + 1 < sp.npass) {
      seg_off[l] = ws.take<uint32_t>(sp.segs[l + 1] + 1);
      seg_tpo[l] = ws.take<uint32_t>(sp.segs[l + 1] + 1);
    }
    base[l] = ws.take<uint32_t>((size_t)sp.tasks_max[l] << sp.bits[l]);
  }
  sr.sorted = ws.take<uint32_t>(T_max);
  uint32_t* block_sums = ws.take<uint32_t>(p.TB / kScanTile + 2);
  size_t p1_cap = T_max / p.K0 + p.TB + 1;
  sr.pbkt = ws.take<uint32_t>(p1_cap);
  sr.order = ws.take<uint32_t>(p1_cap);

  ZK_HIP(hipMemsetAsync(zero_lo, 0, zero_bytes, st));
  const uint32_t nblk = (p.n + 255) / 256;
  if (p.n && !sparse) hipLaunchKernelGGL(msm_digits_kernel, dim3(nblk), dim3(256), 0, st, d_scalars, p.n, p.c, p.W, digits);
  if (sparse) {
    const uint32_t per_wg = kEntriesThreads * kEntriesPerThread;
    hipLaunchKernelGGL(msm_entries_kernel, dim3((p.n + per_wg - 1) / per_wg), dim3(kEntriesThreads), 0, st, d_scalars, p.n,
                       p.c, p.W, elist, misc + 8);
    hipLaunchKernelGGL(msm_entries_finish_kernel, dim3(1), dim3(1), 0, st, (const uint32_t*)(misc + 8), sp.CH, pre, pre + 4);
  }
  for (uint32_t l = 0; l < sp.npass; l++) {
    const bool first = l == 0 && !sparse, last = l + 1 == sp.npass;
    const uint2* in = l == 0 ? (const uint2*)elist : ebuf[(l - 1) & 1];
    uint2* out = last ? nullptr : ebuf[l & 1];
    const uint32_t* in_off = l == 0 ? (const uint32_t*)pre : seg_off[l - 1];
    const uint32_t* tpo = l == 0 ? (const uint32_t*)(pre ? pre + 4 : nullptr) : seg_tpo[l - 1];
    uint32_t* out_cnt = last ? sr.counts : seg_cnt[l];
    uint32_t* out_off = last ? sr.off0 : seg_off[l];
    const dim3 grid = first ? dim3(sp.chunks0, sp.W) : dim3((uint32_t)sp.tasks_max[l]);
    if (p.n) {
      if (first)
        hipLaunchKernelGGL((msm_sort_count_kernel<true>), grid, dim3(256), 0, st, sp, l, (const uint32_t*)digits, in,
                           in_off, tpo, out_cnt, base[l]);
      else
        hipLaunchKernelGGL((msm_sort_count_kernel<false>), grid, dim3(256), 0, st, sp, l, (const uint32_t*)digits, in,
                           in_off, tpo, out_cnt, base[l]);
    }
    if (scan_three_kernel()) {   // ZKPOA_SCAN=3: the three-launch scans (A/B measurement)
      if (last) {
        scan_u32(st, sr.counts, p.TB, 0, 0, sr.off0, block_sums, misc + 0, misc + 1);
        scan_u32(st, sr.counts, p.TB, 1, p.K0, sr.po_a, block_sums, misc + 2, nullptr);
        hipLaunchKernelGGL(msm_piece_hist_kernel, dim3((p.TB + 255) / 256), dim3(256), 0, st, (const uint32_t*)sr.counts,
                           p.TB, p.K0, len_hist);
      } else {
        scan_u32(st, seg_cnt[l], sp.segs[l + 1], 0, 0, seg_off[l], block_sums, misc + 4 + 2 * l, nullptr);
        scan_u32(st, seg_cnt[l], sp.segs[l + 1], 1, sp.CH, seg_tpo[l], block_sums, misc + 5 + 2 * l, nullptr);
      }
    } else if (last) {   // bucket counts -> entry offsets + piece offsets (+ the largest bucket, the piece lengths), one launch
      scan_pair(lane, ScanPair{sr.counts, p.TB, 0, 0, sr.off0, misc + 0, 1, p.K0, sr.po_a, misc + 2, misc + 1, len_hist, p.K0});
    } else {             // segment counts -> entry offsets + task offsets
      scan_pair(lane, ScanPair{seg_cnt[l], sp.segs[l + 1], 0, 0, seg_off[l], misc + 4 + 2 * l, 1, sp.CH, seg_tpo[l],
                               misc + 5 + 2 * l, nullptr});
    }
    if (p.n) {
#define ZK_SORT_SCATTER(F_, L_)                                                                                       \
  hipLaunchKernelGGL((msm_sort_scatter_kernel<F_, L_>), grid, dim3(256), 0, st, sp, l, (const uint32_t*)digits, in,  \
                     in_off, tpo, (const uint32_t*)out_off, (const uint32_t*)base[l], out, sr.sorted)
      if (first && last) ZK_SORT_SCATTER(true, true);
      else if (first) ZK_SORT_SCATTER(true, false);
      else if (last) ZK_SORT_SCATTER(false, true);
      else ZK_SORT_SCATTER(false, false);
#undef ZK_SORT_SCATTER
    }
  }
  uint32_t* hb = reinterpret_cast<uint32_t*>(lane.pinned);
  msm_read_back(lane, misc, 16);
  sr.total_entries = hb[0];
  sr.max_count = hb[1];
  sr.total1 = hb[2];
  if (sr.total1) {
    const uint32_t per_block = 256u * kPiecesPerThread;
    hipLaunchKernelGGL(msm_piece_order_kernel, dim3((sr.total1 + per_block - 1) / per_block), dim3(256), 0, st,
                       (const uint32_t*)sr.counts, (const uint32_t*)sr.po_a, p.TB, p.K0, (const uint32_t*)len_hist,
                       len_cursor, sr.pbkt, sr.order);
  }
  if (sync_at_end) ZK_HIP(hipStreamSynchronize(st));  // other lanes are about to read the result
  return sr;
}

// Phase B on lane.stream for base array d_bases. `own_arena`: true when `sr` was produced on this lane
// by the immediately preceding msm_sort_phase (the accumulation buffers are then taken after it);
// false when `sr` lives on another lane (that lane's stream must have finished phase A): this lane's arena
// is reset and only holds the accumulation buffers. On return window_sums_host = 2 W (logS + 1) XYZZ: per window
// and group (rows, columns) the totals T_b of the sums whose weight has bit b; h_combine_windows forms
// U = sum_b 2^b T_b (rows), V (columns) and the window sum 2^logS * U + V. Room for 2 * 64 * kReduceMaxBits points.
template <class F>
inline void msm_accum_phase(Lane& lane, const MsmSorted& sr, const void* d_bases, void* window_sums_host,
                            bool own_arena, float* accum_ms = nullptr) {
  const MsmPlan& p = sr.p;
  hipStream_t st = lane.stream;
  Arena& ws = lane.ws;
  if (!own_arena) {
    lane_reserve(lane, msm_accum_workspace_bytes<F>(p));
    ws.reset();
  }
  const size_t T_max = (size_t)p.n * p.W;
  uint32_t* po_b = ws.take<uint32_t>(p.TB + 1);
  uint32_t* po_c = ws.take<uint32_t>(p.TB + 1);
  uint32_t* block_sums = ws.take<uint32_t>(p.TB / kScanTile + 2);
  uint32_t* misc = ws.take<uint32_t>(16);
  char* buckets = ws.take<char>((size_t)p.TB * MsmSizes<F>::kXyzz);
  size_t p1_cap = T_max / p.K0 + p.TB + 1;
  size_t p2_cap = p1_cap / 2 + 1;
  char* P1 = ws.take<char>(p1_cap * MsmSizes<F>::kXyzz);
  char* P2 = ws.take<char>(p2_cap * MsmSizes<F>::kXyzz);
  const uint32_t E = 1u << p.logS;            // entries per (window, row|column) group; logS >= logRows
  const uint32_t groups = 2u * p.Wb;
  char* X = ws.take<char>((size_t)groups * E * MsmSizes<F>::kXyzz);
  const uint32_t S1 = (E + 255) / 256;
  const uint32_t nbits = msm_reduce_bits(p.logS), sums = groups * nbits;   // per-bit totals of every group
  char* Y1 = ws.take<char>((size_t)(S1 * sums) * MsmSizes<F>::kXyzz);
  char* Y2 = ws.take<char>((size_t)(S1 * sums) * MsmSizes<F>::kXyzz);

  if (accum_ms) ZK_HIP(hipEventRecord(lane.ev0, st));
  if (sr.total1) {
    const uint32_t pgrid = (sr.total1 + 255) / 256;
    hipLaunchKernelGGL((msm_accum0_kernel<F>), dim3(pgrid), dim3(256), 0, st, d_bases, (const uint32_t*)sr.sorted,
                       (const uint32_t*)sr.counts, (const uint32_t*)sr.off0, (const uint32_t*)sr.po_a,
                       (const uint32_t*)sr.order, (const uint32_t*)sr.pbkt, p.TB, p.K0, (void*)buckets, (void*)P1);
  }
  if (accum_ms) ZK_HIP(hipEventRecord(lane.ev1, st));
  // further levels while some bucket still has more than one partial sum; the shared po_a is read-only,
  // the level offsets ping-pong between this phase's own po_b / po_c
  uint64_t max_items = ((uint64_t)sr.max_count + p.K0 - 1) / p.K0;  // max items per bucket entering level 1
  uint64_t total_in = sr.total1;                                       // upper bound of items entering the level
  const uint32_t* po_in = sr.po_a;
  uint32_t* po_out = po_b;
  char* Pin = P1;
  char* Pout = P2;
  size_t cap_out = p2_cap, cap_in = p1_cap;
  while (max_items > 1) {
    uint64_t bound = total_in / 2 + 1;  // every unfinished bucket holds >= 2 items and emits <= items/2 + 1 pieces
    if (bound > cap_out) throw HipError("msm: level buffer too small");
    if (scan_three_kernel()) scan_u32(st, po_in, p.TB, 3, p.K, po_out, block_sums, misc + 3, nullptr);
    else scan_pair(lane, ScanPair{po_in, p.TB, 3, p.K, po_out, misc + 3, -1, 0, nullptr, nullptr, nullptr});
    hipLaunchKernelGGL((msm_accumN_kernel<F>), dim3((uint32_t)((bound + 255) / 256)), dim3(256), 0, st, (const void*)Pin,
                       po_in, (const uint32_t*)po_out, p.TB, p.K, (void*)buckets, (void*)Pout);
    max_items = (max_items + p.K - 1) / p.K;
    total_in = bound;
    po_in = po_out;
    po_out = (po_out == po_b) ? po_c : po_b;
    std::swap(Pin, Pout);
    std::swap(cap_in, cap_out);
  }
  // bucket reduction: row / column sums, then per (window, group, weight bit) the total of the sums with that bit
  {
    uint32_t log_parts = msm_reduce_log_parts(p.logRows, p.logS, p.Wb);
    uint64_t threads = ((uint64_t)p.Wb << log_parts) * ((1u << p.logRows) + E);
    hipLaunchKernelGGL((msm_bucket_sums_kernel<F>), dim3((uint32_t)((threads + 255) / 256)), dim3(256),
                       256 * MsmSizes<F>::kXyzz, st, (const void*)buckets, (const uint32_t*)sr.counts, p.Wb, p.Nb, p.logS,
                       p.logRows, log_parts, E, (void*)X);
  }
  char* ybuf[2] = {Y1, Y2};
  int yi = 0;
  uint32_t S = S1;
  hipLaunchKernelGGL((msm_bit_tree_sum_kernel<F>), dim3(S1, sums), dim3(64), 64 * MsmSizes<F>::kXyzz, st,
                     (const void*)X, E, p.logS, p.logRows, nbits, S1, (void*)ybuf[yi]);
  const char* cur = ybuf[yi];
  yi ^= 1;
  while (S > 1) {
    uint32_t S_out = (S + 255) / 256;
    hipLaunchKernelGGL((msm_tree_sum_kernel<F>), dim3(S_out, sums), dim3(64), 64 * MsmSizes<F>::kXyzz, st,
                       (const void*)cur, S, S_out, (void*)ybuf[yi]);
    cur = ybuf[yi];
    yi ^= 1;
    S = S_out;
  }
  msm_read_back(lane, cur, (size_t)sums * MsmSizes<F>::kXyzz);
  ZK_HIP(hipGetLastError());
  memcpy(window_sums_host, lane.pinned, (size_t)sums * MsmSizes<F>::kXyzz);
  if (accum_ms) ZK_HIP(hipEventElapsedTime(accum_ms, lane.ev0, lane.ev1));
}

// ---- fixed-base tables -------------------------------------------------------------------------------------
// A proving key's bases never change (zkey sections 5-9), and an MI355X has 288 GB of HBM: store 2^(c*j) * P_i for
// every window j once per key (window-major, affine Montgomery wire format, so table[j * n + i] is an ordinary
// base array). Then all windows share ONE bucket set, which (a) lets the window grow (2^20: c 15 -> 17..19, 17 ->
// 14..15 windows; 2^26: c 20 -> 24, 13 -> 11 windows: that many fewer mixed additions in the kernel that is >90 %
// of an MSM), (b) divides the bucket-reduction work by W, (c) leaves the host a single window to combine.
struct MsmTable {
  void* d = nullptr;     // W * n points
  uint64_t n = 0;        // points per window (= the base array it was built from)
  uint32_t c = 0, W = 0;
  size_t bytes = 0;
};

// table[j * n + i] = 2^(c * j) * P_i: one thread per point walks the windows (c doublings each) and writes XYZZ into
// `scratch` (slab x W points), converted to affine afterwards (msm_table_affine_kernel; once per key).
template <class F>
static __global__ __launch_bounds__(256) void msm_table_chain_kernel(const void* __restrict__ bases, uint64_t i0,
                                                                      uint64_t cnt, uint32_t c, uint32_t W,
                                                                      void* __restrict__ scratch) {
  uint64_t t = (uint64_t)blockIdx.x * 256u + threadIdx.x;
  if (t >= cnt) return;
  XYZZ<F> acc = XYZZ<F>::from_affine(load_affine<F>(bases, i0 + t));
  for (uint32_t j = 0; j < W; j++) {
    store_xyzz(scratch, (size_t)j * cnt + t, acc);
    if (j + 1 < W)
      for (uint32_t k = 0; k < c; k++) acc = xyzz_dbl(acc);
  }
}

// scratch (XYZZ, window-major within the slab) -> table rows in the zkey wire format. One thread per POINT converts
// its W window entries with ONE field inversion (Montgomery's trick over the ZZZ coordinates: prefix products kept in
// `prefix`, then unwound), ~40 multiplications per entry instead of the ~400 of an inversion each.
template <class F>
static __global__ __launch_bounds__(256) void msm_table_affine_kernel(const void* __restrict__ scratch,
                                                                       void* __restrict__ prefix, uint64_t i0,
                                                                       uint64_t cnt, uint64_t n, uint32_t W,
                                                                       void* __restrict__ table) {
  uint64_t t = (uint64_t)blockIdx.x * 256u + threadIdx.x;
  if (t >= cnt) return;
  constexpr int FB = FieldBytes<F>::N;
  char* pre = reinterpret_cast<char*>(prefix);
  // a point at infinity stays at infinity in every window
  XYZZ<F> p0 = load_xyzz<F>(scratch, t);
  if (p0.is_inf()) {
    for (uint32_t j = 0; j < W; j++) {
      char* o = reinterpret_cast<char*>(table) + ((size_t)j * n + i0 + t) * (size_t)(2 * FB);
      store_field(o, F::zero());
      store_field(o + FB, F::zero());
    }
    return;
  }
  F acc = F::one();
  for (uint32_t j = 0; j < W; j++) {   // prefix[j] = zzz_0 * ... * zzz_(j-1)
    store_field(pre + ((size_t)j * cnt + t) * FB, acc);
    XYZZ<F> p = load_xyzz<F>(scratch, (size_t)j * cnt + t);
    if (!p.is_inf()) acc = acc * p.zzz;   // (a doubling can only give infinity for a point outside the prime-order group)
  }
  F inv = acc.inv();
  for (uint32_t jj = W; jj-- > 0;) {
    XYZZ<F> p = load_xyzz<F>(scratch, (size_t)jj * cnt + t);
    char* o = reinterpret_cast<char*>(table) + ((size_t)jj * n + i0 + t) * (size_t)(2 * FB);
    if (p.is_inf()) {
      store_field(o, F::zero());
      store_field(o + FB, F::zero());
      continue;
    }
    F i3 = inv * load_field<F>(pre + ((size_t)jj * cnt + t) * FB);   // 1 / zzz_jj
    inv = inv * p.zzz;
    F i2 = (p.zz * i3).sqr();
    store_field(o, p.x * i2);
    store_field(o + FB, p.y * i3);
  }
}

template <class F>
inline size_t msm_table_bytes(uint64_t n, uint32_t c) {
  return (size_t)n * msm_windows(c) * MsmSizes<F>::kAffine;
}

// Window width of a table over n bases (the merged-bucket cost model), or `force_c`.
inline uint32_t msm_table_c(uint64_t n, int force_c = 0, bool for_g2 = false) {
  return msm_make_plan((size_t)n, force_c, for_g2, true).c;
}

// Builds the table on `st` (synchronised on return). Slabs of 2^20 points bound the scratch memory.
template <class F>
inline MsmTable msm_table_build(hipStream_t st, const void* d_bases, uint64_t n, uint32_t c) {
  MsmTable t;
  t.n = n;
  t.c = c;
  t.W = msm_windows(c);
  t.bytes = msm_table_bytes<F>(n, c);
  if ((uint64_t)n * t.W >= 0x80000000ull) throw HipError("msm table: n * windows exceeds the fixed-base entry index");
  ZK_HIP(hipMalloc(&t.d, t.bytes ? t.bytes : 1));
  const uint64_t slab = 1ull << 20;
  void* scratch = nullptr;
  void* prefix = nullptr;
  try {
    ZK_HIP(hipMalloc(&scratch, (size_t)(n < slab ? (n ? n : 1) : slab) * t.W * MsmSizes<F>::kXyzz));
    ZK_HIP(hipMalloc(&prefix, (size_t)(n < slab ? (n ? n : 1) : slab) * t.W * FieldBytes<F>::N));
    for (uint64_t off = 0; off < n; off += slab) {
      const uint64_t cnt = n - off < slab ? n - off : slab;
      hipLaunchKernelGGL((msm_table_chain_kernel<F>), dim3((uint32_t)((cnt + 255) / 256)), dim3(256), 0, st, d_bases, off,
                         cnt, t.c, t.W, scratch);
      hipLaunchKernelGGL((msm_table_affine_kernel<F>), dim3((uint32_t)((cnt + 255) / 256)), dim3(256), 0, st,
                         (const void*)scratch, prefix, off, cnt, n, t.W, t.d);
    }
    ZK_HIP(hipStreamSynchronize(st));
    ZK_HIP(hipGetLastError());
  } catch (...) {
    if (scratch) (void)hipFree(scratch);
    if (prefix) (void)hipFree(prefix);
    (void)hipFree(t.d);
    throw;
  }
  (void)hipFree(scratch);
  (void)hipFree(prefix);
  return t;
}

inline void msm_table_free(MsmTable& t) {
  if (t.d) (void)hipFree(t.d);
  t = MsmTable();
}

// One complete MSM on one lane (phase A + phase B). Returns the plan used. With a table (built from exactly these
// n bases) the fixed-base form runs: d_bases is then not read.
template <class F>
inline MsmPlan msm_device(Lane& lane, const void* d_bases, const void* d_scalars, size_t n, void* window_sums_host,
                          int force_c = 0, float* accum_ms = nullptr, const MsmTable* table = nullptr,
                          uint32_t* entries = nullptr) {
  if (table && table->n != n) throw HipError("msm: the fixed-base table was built for another base array");
  MsmSorted sr = msm_sort_phase(lane, d_scalars, n, table ? (int)table->c : force_c, &msm_accum_workspace_bytes<F>,
                                false, FieldBytes<F>::N > 32, table != nullptr);
  msm_accum_phase<F>(lane, sr, table ? table->d : d_bases, window_sums_host, true, accum_ms);
  if (entries) *entries = sr.total_entries;
  return sr.p;
}

}  // namespace zkpoa
