// Phase-2 setup arithmetic on the device (first half): the point sections of a fresh proving key.
//
// `snarkjs zkey new circuit.r1cs pot.ptau circuit_0000.zkey` (scripts/g16_setup.sh:243-246; "34 h" for the layer-three
// circuit, README.md:179) is, for every signal s, a sum over the R1CS coefficients that mention it:
//   A[s]  = sum_{(c, k) in A column s}  k * L_c(tau) G1                              (zkey section 5)
//   B1[s], B2[s]: the same over the B matrix, in G1 and in G2                        (sections 6, 7)
//   IC / C[s] = sum_A k * beta L_c(tau) G1 + sum_B k * alpha L_c(tau) G1 + sum_C k * L_c(tau) G1   (sections 3, 8)
// with L_c(tau) G the powers-of-tau points in Lagrange form (the prepared .ptau's sections 12-15). snarkjs is not
// vendored in the reference (package.json dependency); this restates the published algorithm (snarkjs zkey_new.js).
// Each sum is a transposed sparse-matrix x point-vector product: the coefficients are mostly 1, -1 and small
// constants, the columns short except for a few signals (the constant 1) that occur in a large part of the rows.
//
// Device mapping (one call per section; zkpoa_setup_accumulate):
//   1. histogram of the entries by signal (atomics), exclusive scan -> segment offsets, scatter of the entry ids;
//   2. one lane per entry, in segment order: k * P by MSB-first double-and-add on the sign-normalised coefficient
//      (k > r/2 -> (r - k) * (-P): "-1" is one addition, not 254 doublings), XYZZ result into the segment's slot;
//   3. the MSM's partial-sum levels (msm_accumN_kernel, fan-in 4: a hot segment is a chain of full additions, so
//      depth matters) until every segment is one point; 4. XYZZ -> affine, zkey wire format.
// The order inside a segment depends on the atomics; the sum does not, and the affine output is canonical.
#include "msm.hip.h"
#include "hooks.hip.h"
#include "zkpoa_internal.hpp"

using namespace zkpoa;

namespace {

// counts[sig[e]]++; flags |= 1 for an out-of-range index, |= 2 for a coefficient >= r
static __global__ __launch_bounds__(256) void setup_hist_kernel(const uint32_t* __restrict__ sig,
                                                                const uint32_t* __restrict__ pidx,
                                                                const void* __restrict__ coefs, uint32_t nnz,
                                                                uint32_t n_signals, uint32_t n_points,
                                                                uint32_t* __restrict__ counts, uint32_t* __restrict__ flags) {
  uint32_t e = blockIdx.x * 256u + threadIdx.x;
  if (e >= nnz) return;
  uint32_t s = sig[e];
  uint32_t k[8];
  load_scalar(coefs, e, k);
  uint32_t bw = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) (void)subb(k[i], FrParams::P[i], bw);
  if (s >= n_signals || pidx[e] >= n_points) {
    atomicOr(flags, 1u);
    return;
  }
  if (!bw) atomicOr(flags, 2u);
  atomicAdd(&counts[s], 1u);
}

static __global__ __launch_bounds__(256) void setup_scatter_kernel(const uint32_t* __restrict__ sig, uint32_t nnz,
                                                                   const uint32_t* __restrict__ off,
                                                                   uint32_t* __restrict__ cursor,
                                                                   uint32_t* __restrict__ order) {
  uint32_t e = blockIdx.x * 256u + threadIdx.x;
  if (e >= nnz) return;
  uint32_t s = sig[e];
  order[off[s] + atomicAdd(&cursor[s], 1u)] = e;
}

// slot j (segment order): Q = coef * P; a segment of one entry goes straight to its bucket
template <class F>
static __global__ __launch_bounds__(256) void setup_mul_kernel(const void* __restrict__ points,
                                                               const void* __restrict__ coefs,
                                                               const uint32_t* __restrict__ pidx,
                                                               const uint32_t* __restrict__ sig,
                                                               const uint32_t* __restrict__ order,
                                                               const uint32_t* __restrict__ off, uint32_t nnz,
                                                               void* __restrict__ buckets, void* __restrict__ items) {
  uint32_t j = blockIdx.x * 256u + threadIdx.x;
  if (j >= nnz) return;
  const uint32_t e = order[j], s = sig[e];
  uint32_t k[8];
  load_scalar(coefs, e, k);
  const bool neg = scalar_normalize(k);
  const Affine<F> p = load_affine<F>(points, pidx[e]);
  int top = -1;
#pragma unroll
  for (int i = 7; i >= 0; i--)
    if (top < 0 && k[i]) top = 32 * i + (31 - __builtin_clz(k[i]));
  XYZZ<F> acc = XYZZ<F>::inf();
  for (int bit = top; bit >= 0; bit--) {   // one doubling site, one addition site
    acc = xyzz_dbl(acc);
    if ((k[bit >> 5] >> (bit & 31)) & 1u) xyzz_add_affine(acc, p, neg);
  }
  if (off[s + 1] - off[s] == 1u) store_xyzz(buckets, s, acc);
  else store_xyzz(items, j, acc);
}

template <class F>
void setup_accumulate(zkpoa_context* ctx, const void* d_points, uint64_t n_points, const void* d_coefs,
                      const uint32_t* d_pidx, const uint32_t* d_sig, uint64_t nnz, uint64_t n_signals, void* d_out) {
  if (n_signals == 0) return;
  if (nnz >= (1ull << 31) || n_signals >= (1ull << 31) || n_points >= (1ull << 32))
    throw HipError("setup_accumulate: more than 2^31 entries or signals in one call");
  Lane& lane = ctx->dev.lanes[0];
  hipStream_t st = lane.stream;
  const uint32_t S = (uint32_t)n_signals, N = (uint32_t)nnz;
  constexpr size_t X = MsmSizes<F>::kXyzz;
  DevBuf counts((size_t)S * 4), off(((size_t)S + 1) * 4), po_b(((size_t)S + 1) * 4), po_c(((size_t)S + 1) * 4),
      block_sums(((size_t)S / kScanTile + 2) * 4), misc(64), order((size_t)(N ? N : 1) * 4), buckets((size_t)S * X),
      items((size_t)(N ? N : 1) * X), items2(((size_t)N / 2 + 2) * X);
  uint32_t* m = reinterpret_cast<uint32_t*>(misc.p);   // [0] flags, [1] total, [2] max segment, [3] level total
  ZK_HIP(hipMemsetAsync(counts.p, 0, (size_t)S * 4, st));
  ZK_HIP(hipMemsetAsync(misc.p, 0, 64, st));
  ZK_HIP(hipMemsetAsync(buckets.p, 0, (size_t)S * X, st));   // the all-zero XYZZ is the point at infinity
  const uint32_t grid = (N + 255) / 256;
  if (N)
    hipLaunchKernelGGL(setup_hist_kernel, dim3(grid), dim3(256), 0, st, d_sig, d_pidx, d_coefs, N, S, (uint32_t)n_points,
                       (uint32_t*)counts.p, m);
  scan_u32(st, (const uint32_t*)counts.p, S, 0, 0, (uint32_t*)off.p, (uint32_t*)block_sums.p, m + 1, m + 2);
  msm_read_back(lane, m, 16);
  const uint32_t* hb = reinterpret_cast<const uint32_t*>(lane.pinned);
  const uint32_t flags = hb[0], total = hb[1], max_seg = hb[2];
  if (flags & 1u) throw HipError("setup_accumulate: signal or point index out of range");
  if (flags & 2u) throw HipError("setup_accumulate: coefficient is not a field element (>= r)");
  if (total != N) throw HipError("setup_accumulate: internal: histogram does not add up");
  if (N) {
    ZK_HIP(hipMemsetAsync(counts.p, 0, (size_t)S * 4, st));   // reused as the scatter cursors
    hipLaunchKernelGGL(setup_scatter_kernel, dim3(grid), dim3(256), 0, st, d_sig, N, (const uint32_t*)off.p,
                       (uint32_t*)counts.p, (uint32_t*)order.p);
    hipLaunchKernelGGL((setup_mul_kernel<F>), dim3(grid), dim3(256), 0, st, d_points, d_coefs, d_pidx, d_sig,
                       (const uint32_t*)order.p, (const uint32_t*)off.p, N, buckets.p, items.p);
  }
  // partial-sum levels, as after the MSM's level 0 (msm_accum_phase): fan-in 4 until every segment is one point
  const uint32_t K = 4;
  uint64_t max_items = max_seg, total_in = N;
  const uint32_t* po_in = (const uint32_t*)off.p;
  uint32_t* po_out = (uint32_t*)po_b.p;
  char *Pin = (char*)items.p, *Pout = (char*)items2.p;
  while (max_items > 1) {
    uint64_t bound = total_in / 2 + 1;
    scan_u32(st, po_in, S, 3, K, po_out, (uint32_t*)block_sums.p, m + 3, nullptr);
    hipLaunchKernelGGL((msm_accumN_kernel<F>), dim3((uint32_t)((bound + 255) / 256)), dim3(256), 0, st, (const void*)Pin,
                       po_in, (const uint32_t*)po_out, S, K, buckets.p, (void*)Pout);
    max_items = (max_items + K - 1) / K;
    total_in = bound;
    po_in = po_out;
    po_out = (po_out == (uint32_t*)po_b.p) ? (uint32_t*)po_c.p : (uint32_t*)po_b.p;
    std::swap(Pin, Pout);
  }
  hipLaunchKernelGGL((xyzz_to_affine_kernel<F>), dim3((S + 255) / 256), dim3(256), 0, st, (const void*)buckets.p, d_out,
                     (uint64_t)S);
  ZK_HIP(hipStreamSynchronize(st));
  ZK_HIP(hipGetLastError());
}

}  // namespace

extern "C" int zkpoa_setup_accumulate(zkpoa_context* ctx, int group, const void* d_points, uint64_t n_points,
                                      const void* d_coefs, const uint32_t* d_point_index, const uint32_t* d_signal,
                                      uint64_t nnz, uint64_t n_signals, void* d_out) {
  ZK_API_BEGIN(ctx)
  if (group != 1 && group != 2) throw HipError("setup_accumulate: group must be 1 (G1) or 2 (G2)");
  if ((nnz && (!d_points || !d_coefs || !d_point_index || !d_signal)) || (n_signals && !d_out))
    throw HipError("setup_accumulate: null pointer");
  if (group == 1) setup_accumulate<Fq>(ctx, d_points, n_points, d_coefs, d_point_index, d_signal, nnz, n_signals, d_out);
  else setup_accumulate<Fq2>(ctx, d_points, n_points, d_coefs, d_point_index, d_signal, nnz, n_signals, d_out);
  ZK_API_END(ctx)
}
