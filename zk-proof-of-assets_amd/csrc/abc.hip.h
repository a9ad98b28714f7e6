// buildABC / joinABC on the device.
//
// Replaces snarkjs groth16_prove.js::buildABC1 + joinABC (SURVEY.md 3.2 steps 2 and 4, 8a rows
// a5/a7) and the coefficient loop of rapidsnark's groth16.cpp:
//   out[m][c] += coef (x) w[s]      for every record (m, c, s, coef*R^2) of zkey section 4
//   C[c] = A[c] (x) B[c]
//   P[i] = fromMontgomery(A'[i] (x) B'[i] - C'[i])
// (x) = Montgomery product; witness values are standard form, so the products come out in
// Montgomery form of coef*w.
//
// The 44-byte records arrive in file order. Field elements have no atomic add, so the records
// are bucketed by output row once per proving key (counting sort on key = 2*c + m, done at
// zkey load: it depends on the key only) into CSR arrays; per proof one thread per constraint
// walks its two rows. HBM-bound: 36 B (value + signal index) + 32 B gathered witness per record.
#pragma once
#include "bn254_field.hip.h"
#include "device_ctx.hpp"

namespace zkpoa {

struct CoefRec {  // 44 bytes, 4-byte aligned, as stored in zkey section 4 after the u32 count
  uint32_t m, c, s;
  uint32_t val[8];
};
static_assert(sizeof(CoefRec) == 44, "coef record");

// ---- range validation of untrusted field elements -------------------------------------------------------------
// The device keeps field elements lazily in [0, 2p) and takes file data raw, so a value >= the modulus in a .wtns
// or .zkey would flow through (a witness value >= r breaks the scalar recoding's r - s: a silently wrong proof with
// exit code 0). One streaming pass ORs a flag instead: count elements of 32 B each, all must be < P.
template <class PRM>
static __global__ __launch_bounds__(256) void range_check_kernel(const void* __restrict__ data, uint64_t count,
                                                                 uint32_t* __restrict__ flag) {
  uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
  if (i >= count) return;
  const uint4* q = reinterpret_cast<const uint4*>(data) + 2 * i;
  uint4 a = q[0], b = q[1];
  const uint32_t l[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
  uint32_t bw = 0;
#pragma unroll
  for (int k = 0; k < 8; k++) (void)subb(l[k], PRM::P[k], bw);
  if (!bw) atomicOr(flag, 1u);   // no borrow: value >= P
}

// Split chain: the stages compute on [polynomial][M] arrays, the exchange buffers are laid out
// [destination rank][polynomial][Q] (Q = M / G) so that ONE all-to-all with equal splits moves all three polynomials.
// pack: work[x][h * Q + pl] -> xbuf[(h * 3 + x) * Q + pl]; unpack is the inverse. One thread per 16-byte half element.
template <bool PACK>
static __global__ __launch_bounds__(256) void split_pack_kernel(const uint4* __restrict__ in, uint4* __restrict__ out,
                                                                uint32_t M, uint32_t Q) {
  uint64_t t = (uint64_t)blockIdx.x * 256u + threadIdx.x;
  if (t >= (uint64_t)6 * M) return;
  const uint32_t half = (uint32_t)(t & 1u);
  const uint64_t e = t >> 1;                       // element in [3][M] order
  const uint32_t x = (uint32_t)(e / M), p = (uint32_t)(e % M);
  const uint32_t h = p / Q, pl = p % Q;
  const uint64_t xe = ((uint64_t)h * 3u + x) * Q + pl;   // element in [G][3][Q] order
  if (PACK) out[2 * xe + half] = in[2 * e + half];
  else out[2 * e + half] = in[2 * xe + half];
}

constexpr uint32_t kAbcSkip = 0xffffffffu;   // rank of a record that belongs to another rank's constraint rows

// pass 1: validate + histogram rows, remember the rank inside the row. err[0] != 0 on bad records.
// Split chain (buildABC row): a rank keeps the constraints c = part (mod 2^log_parts) only and
// numbers them c >> log_parts; log_parts = 0 keeps everything.
static __global__ __launch_bounds__(256) void abc_count_kernel(const CoefRec* __restrict__ recs, uint64_t ncoef,
                                                               uint32_t domain, uint32_t nvars, uint32_t log_parts,
                                                               uint32_t part, uint32_t* __restrict__ row_cnt,
                                                               uint32_t* __restrict__ rank, uint32_t* __restrict__ err) {
  uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
  if (i >= ncoef) return;
  uint32_t m = recs[i].m, c = recs[i].c, s = recs[i].s;
  uint32_t bw = 0;   // the coefficient value must be a canonical Fr element
#pragma unroll
  for (int k = 0; k < 8; k++) (void)subb(recs[i].val[k], FrParams::P[k], bw);
  if (m > 1u || c >= domain || s >= nvars || !bw) {
    atomicOr(err, bw ? 1u : 2u);
    rank[i] = kAbcSkip;
    return;
  }
  if ((c & ((1u << log_parts) - 1u)) != part) {
    rank[i] = kAbcSkip;
    return;
  }
  rank[i] = atomicAdd(&row_cnt[2u * (c >> log_parts) + m], 1u);
}

// pass 2: scatter into CSR order
static __global__ __launch_bounds__(256) void abc_scatter_kernel(const CoefRec* __restrict__ recs, uint64_t ncoef,
                                                                 uint32_t log_parts,
                                                                 const uint32_t* __restrict__ row_ptr,
                                                                 const uint32_t* __restrict__ rank,
                                                                 uint32_t* __restrict__ sig, void* __restrict__ vals) {
  uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
  if (i >= ncoef) return;
  uint32_t m = recs[i].m, c = recs[i].c;
  if (rank[i] == kAbcSkip) return;
  uint32_t pos = row_ptr[2u * (c >> log_parts) + m] + rank[i];
  sig[pos] = recs[i].s;
  Fr v;
#pragma unroll
  for (int k = 0; k < 8; k++) v.l[k] = recs[i].val[k];
  store_field(reinterpret_cast<char*>(vals) + 32 * (size_t)pos, v);
}

ZK_DEV Fr abc_row_sum(const uint32_t* __restrict__ row_ptr, uint32_t row, const uint32_t* __restrict__ sig,
                      const void* __restrict__ vals, const void* __restrict__ witness) {
  Fr acc = Fr::zero();
  uint32_t b = row_ptr[row], e = row_ptr[row + 1];
  for (uint32_t k = b; k < e; k++) {
    Fr v = load_field<Fr>(reinterpret_cast<const char*>(vals) + 32 * (size_t)k);
    Fr w = load_field<Fr>(reinterpret_cast<const char*>(witness) + 32 * (size_t)sig[k]);
    acc = acc + v * w;
  }
  return acc;
}

// A constraint whose two rows hold more than kLongRow coefficients (a Num2Bits sum, a big-integer carry chain:
// hundreds of terms) would keep one lane busy for hundreds of iterations while its 63 neighbours wait, so such
// constraints are listed once per key and handled one WAVE per constraint (abc_long_rows_kernel).
constexpr uint32_t kLongRow = 32;

// one thread per output t < count: CSR row pair of constraint c = row_off + row_stride * t ->
// A_T[t], B_T[t], C_T[t] = A (x) B. (row_off, row_stride) = (0, 1) walks the whole CSR; a rank of the
// split chain walks its cyclic rows of a full CSR with (rank, world). Long constraints are left to the wave kernel.
static __global__ __launch_bounds__(256) void abc_rows_kernel(const uint32_t* __restrict__ row_ptr,
                                                              const uint32_t* __restrict__ sig,
                                                              const void* __restrict__ vals,
                                                              const void* __restrict__ witness, uint32_t count,
                                                              uint32_t row_off, uint32_t row_stride,
                                                              void* __restrict__ A, void* __restrict__ B,
                                                              void* __restrict__ C) {
  uint32_t t = blockIdx.x * 256u + threadIdx.x;
  if (t >= count) return;
  uint32_t c = row_off + row_stride * t;
  if (row_ptr[2u * c + 2u] - row_ptr[2u * c] > kLongRow) return;
  Fr a = abc_row_sum(row_ptr, 2u * c, sig, vals, witness);
  Fr b = abc_row_sum(row_ptr, 2u * c + 1u, sig, vals, witness);
  store_field(reinterpret_cast<char*>(A) + 32 * (size_t)t, a);
  store_field(reinterpret_cast<char*>(B) + 32 * (size_t)t, b);
  store_field(reinterpret_cast<char*>(C) + 32 * (size_t)t, a * b);
}

// key load: list the long constraints (order irrelevant). cnt[0] = how many; list may be null (count only)
static __global__ __launch_bounds__(256) void abc_long_list_kernel(const uint32_t* __restrict__ row_ptr, uint32_t rows2,
                                                                   uint32_t* __restrict__ cnt, uint32_t* __restrict__ list) {
  uint32_t c = blockIdx.x * 256u + threadIdx.x;
  if (c >= rows2 / 2u) return;
  if (row_ptr[2u * c + 2u] - row_ptr[2u * c] > kLongRow) {
    uint32_t slot = atomicAdd(cnt, 1u);
    if (list) list[slot] = c;
  }
}

// sum of one CSR row by a whole wave: lanes stride over the coefficients, then a shuffle tree
ZK_DEV Fr abc_row_sum_wave(const uint32_t* __restrict__ row_ptr, uint32_t row, const uint32_t* __restrict__ sig,
                           const void* __restrict__ vals, const void* __restrict__ witness, uint32_t lane) {
  Fr acc = Fr::zero();
  uint32_t b = row_ptr[row], e = row_ptr[row + 1];
  for (uint32_t k = b + lane; k < e; k += 64u) {
    Fr v = load_field<Fr>(reinterpret_cast<const char*>(vals) + 32 * (size_t)k);
    Fr w = load_field<Fr>(reinterpret_cast<const char*>(witness) + 32 * (size_t)sig[k]);
    acc = acc + v * w;
  }
  for (uint32_t off = 32u; off > 0u; off >>= 1) {
    Fr o;
#pragma unroll
    for (int i = 0; i < 8; i++) o.l[i] = __shfl_down(acc.l[i], off);
    acc = acc + o;
  }
  return acc;   // complete in lane 0
}

// one wave per listed constraint c. Outputs go to index t: c itself, or for a rank of the split chain walking a
// full CSR only the constraints with c = part (mod 2^log_parts), at t = c >> log_parts.
static __global__ __launch_bounds__(256) void abc_long_rows_kernel(const uint32_t* __restrict__ row_ptr,
                                                                   const uint32_t* __restrict__ sig,
                                                                   const void* __restrict__ vals,
                                                                   const void* __restrict__ witness,
                                                                   const uint32_t* __restrict__ list, uint32_t n_long,
                                                                   uint32_t log_parts, uint32_t part,
                                                                   void* __restrict__ A, void* __restrict__ B,
                                                                   void* __restrict__ C) {
  const uint32_t wave = (blockIdx.x * 256u + threadIdx.x) >> 6, lane = threadIdx.x & 63u;
  if (wave >= n_long) return;
  const uint32_t c = list[wave];
  if ((c & ((1u << log_parts) - 1u)) != part) return;
  const uint32_t t = c >> log_parts;
  Fr a = abc_row_sum_wave(row_ptr, 2u * c, sig, vals, witness, lane);
  Fr b = abc_row_sum_wave(row_ptr, 2u * c + 1u, sig, vals, witness, lane);
  if (lane == 0) {
    store_field(reinterpret_cast<char*>(A) + 32 * (size_t)t, a);
    store_field(reinterpret_cast<char*>(B) + 32 * (size_t)t, b);
    store_field(reinterpret_cast<char*>(C) + 32 * (size_t)t, a * b);
  }
}

// P[i] = fromMontgomery(A[i] (x) B[i] - C[i]); out may be A
static __global__ __launch_bounds__(256) void abc_join_kernel(const void* A, const void* __restrict__ B,
                                                              const void* __restrict__ C, uint32_t domain, void* out) {
  uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= domain) return;
  Fr a = load_field<Fr>(reinterpret_cast<const char*>(A) + 32 * (size_t)i);
  Fr b = load_field<Fr>(reinterpret_cast<const char*>(B) + 32 * (size_t)i);
  Fr c = load_field<Fr>(reinterpret_cast<const char*>(C) + 32 * (size_t)i);
  store_field(reinterpret_cast<char*>(out) + 32 * (size_t)i, (a * b - c).from_mont());
}

// dst[i] = src[off + stride * i], 64-byte elements (G1 affine points): the cyclic H-point shard of a split chain
static __global__ __launch_bounds__(256) void strided_copy64_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst,
                                                                    uint64_t count, uint32_t off, uint32_t stride) {
  uint64_t q = (uint64_t)blockIdx.x * 256u + threadIdx.x;   // one 16-byte quarter per thread
  if (q >= count * 4u) return;
  uint64_t i = q >> 2;
  dst[q] = src[((uint64_t)off + (uint64_t)stride * i) * 4u + (q & 3u)];
}

// ---- A / B queries without their points at infinity -----------------------------------------------------
// zkey sections 5 and 6 / 7 hold A_i(tau)*G resp. B_i(tau)*G for EVERY wire, and a wire that never appears in
// that matrix is the point at infinity (all-zero bytes). Those (point, scalar) pairs contribute nothing to
// pi_a / pi_b, yet a lane that meets one still spends a whole mixed addition's time, so the key keeps compacted
// copies of the sections plus the wire index of every kept point, and the MSMs run over gathered scalars.
// keep[i] = 1 unless g1[i] (64 B) and, if given, g2[i] (128 B) are all-zero
static __global__ __launch_bounds__(256) void query_keep_kernel(const uint4* __restrict__ g1, const uint4* __restrict__ g2,
                                                                uint32_t n, uint32_t* __restrict__ keep) {
  uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n) return;
  uint32_t o = 0;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    uint4 v = g1[(size_t)i * 4 + k];
    o |= v.x | v.y | v.z | v.w;
  }
  if (g2) {
#pragma unroll
    for (int k = 0; k < 8; k++) {
      uint4 v = g2[(size_t)i * 8 + k];
      o |= v.x | v.y | v.z | v.w;
    }
  }
  keep[i] = o ? 1u : 0u;
}

// Block-cyclic shards (prover.hip zkpoa_zkey::bc_log): local index j of a rank's concatenated blocks of 2^L items ->
// global index; L == 0: contiguous range starting at base
ZK_DEV uint32_t bc_global(uint32_t j, uint32_t base, uint32_t L, uint32_t rank, uint32_t world) {
  if (L == 0) return base + j;
  return (((j >> L) * world + rank) << L) + (j & ((1u << L) - 1u));
}

// pos = exclusive scan of keep: kept point i goes to slot pos[i]; wire[slot] = the wire of resident point i
static __global__ __launch_bounds__(256) void query_compact_kernel(const uint4* __restrict__ g1, const uint4* __restrict__ g2,
                                                                   const uint32_t* __restrict__ pos, uint32_t n,
                                                                   uint32_t wire0, uint32_t bc_log, uint32_t bc_rank,
                                                                   uint32_t bc_world, uint4* __restrict__ c1,
                                                                   uint4* __restrict__ c2, uint32_t* __restrict__ wire) {
  uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n) return;
  uint32_t s = pos[i];
  if (pos[i + 1] == s) return;
#pragma unroll
  for (int k = 0; k < 4; k++) c1[(size_t)s * 4 + k] = g1[(size_t)i * 4 + k];
  if (g2) {
#pragma unroll
    for (int k = 0; k < 8; k++) c2[(size_t)s * 8 + k] = g2[(size_t)i * 8 + k];
  }
  wire[s] = bc_global(i, wire0, bc_log, bc_rank, bc_world);
}

// out[j] = values[idx[j]], 32-byte elements (two 16-byte quarters per thread pair)
static __global__ __launch_bounds__(256) void gather32_kernel(const uint4* __restrict__ values, const uint32_t* __restrict__ idx,
                                                              uint64_t count, uint4* __restrict__ out) {
  uint64_t q = (uint64_t)blockIdx.x * 256u + threadIdx.x;
  if (q >= count * 2u) return;
  out[q] = values[(uint64_t)idx[q >> 1] * 2u + (q & 1u)];
}

// One exchange of the split chain as ONE kernel (csrc/multi_device.hip.h): chunk h of this rank's send buffer goes
// straight into rank h's receive buffer through its peer-mapped address -- every xGMI link of the GPU carries its own
// pair at the same time (G copies enqueued on one stream would run one after the other, one link at a time), and the
// stores sit on the stage's own stream, so no extra ordering is needed. The walk over the peers starts at the rank's own
// slot so that at any moment the ranks aim at different peers. dst.p[h] = where rank h wants THIS rank's chunk.
struct XchgDst {
  char* p[8];
};
static __global__ __launch_bounds__(256) void xchg_push_kernel(const uint4* __restrict__ src, XchgDst dst, uint64_t chunk16,
                                                               uint32_t G, uint32_t self) {
  const uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
  if (i >= chunk16 * G) return;
  const uint32_t h = (self + (uint32_t)(i / chunk16)) % G;
  const uint64_t o = i % chunk16;
  reinterpret_cast<uint4*>(dst.p[h])[o] = src[(uint64_t)h * chunk16 + o];
}

// The witness of a multi-GPU proof: every rank uploads 1 / G of it over its own PCIe link and this kernel stores that
// slice into the G - 1 other ranks' witness buffers over xGMI (dst.p[h] = the slice's place in rank h's buffer).
static __global__ __launch_bounds__(256) void xchg_bcast_kernel(const uint4* __restrict__ src, XchgDst dst, uint64_t n16,
                                                                uint32_t G, uint32_t self) {
  const uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
  if (i >= n16 * (G - 1u)) return;
  const uint32_t h = (self + 1u + (uint32_t)(i / n16)) % G;
  const uint64_t o = i % n16;
  reinterpret_cast<uint4*>(dst.p[h])[o] = src[o];
}

// out[j] = values[global(j)], 32-byte elements: the C query's witness values of a block-cyclic shard
static __global__ __launch_bounds__(256) void gather_bc32_kernel(const uint4* __restrict__ values, uint64_t count,
                                                                 uint32_t bc_log, uint32_t bc_rank, uint32_t bc_world,
                                                                 uint4* __restrict__ out) {
  uint64_t q = (uint64_t)blockIdx.x * 256u + threadIdx.x;
  if (q >= count * 2u) return;
  out[q] = values[(uint64_t)bc_global((uint32_t)(q >> 1), 0u, bc_log, bc_rank, bc_world) * 2u + (q & 1u)];
}

}  // namespace zkpoa
