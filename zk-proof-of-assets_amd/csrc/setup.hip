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
//   2. one lane per entry: k * P by MSB-first double-and-add on the sign-normalised coefficient (k > r/2 ->
//      (r - k) * (-P): "-1" is one addition, not 254 doublings), XYZZ result into the entry's slot in segment order;
//      lanes take the entries sorted by coefficient length, so a wave's lanes run equally long;
//   3. the MSM's partial-sum levels (msm_accumN_kernel, fan-in 4: a hot segment is a chain of full additions, so
//      depth matters) until every segment is one point; 4. XYZZ -> affine, zkey wire format.
// The order inside a segment depends on the atomics; the sum does not, and the affine output is canonical.
#include "abc.hip.h"
#include "msm.hip.h"
#include "hooks.hip.h"
#include "zkpoa_internal.hpp"

#include <fcntl.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <exception>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

using namespace zkpoa;

namespace zkpoa {
template <> Affine<HFq> host_generator<HFq>();     // hooks_g1.hip
template <> Affine<HFq2> host_generator<HFq2>();   // hooks_g2.hip
}

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

// bit length (0..254) of the sign-normalised coefficient of entry e: the length of its double-and-add
ZK_DEV uint32_t setup_bitlen(const void* __restrict__ coefs, uint32_t e) {
  uint32_t k[8];
  load_scalar(coefs, e, k);
  (void)scalar_normalize(k);
  uint32_t len = 0;
#pragma unroll
  for (int i = 7; i >= 0; i--)
    if (len == 0 && k[i]) len = 32u * i + (32u - __builtin_clz(k[i]));
  return len;
}
// counting sort of the slots by that length, longest first (256 keys): a wave runs as long as its longest
// coefficient, and an R1CS mixes 1 and -1 (one addition) with full-width constants (254 doublings)
static __global__ __launch_bounds__(256) void setup_len_hist_kernel(const void* __restrict__ coefs,
                                                                    const uint32_t* __restrict__ order, uint32_t nnz,
                                                                    uint32_t* __restrict__ hist) {
  __shared__ uint32_t h[256];
  h[threadIdx.x] = 0;
  __syncthreads();
  uint32_t j = blockIdx.x * 256u + threadIdx.x;
  if (j < nnz) atomicAdd(&h[255u - setup_bitlen(coefs, order[j])], 1u);
  __syncthreads();
  if (h[threadIdx.x]) atomicAdd(&hist[threadIdx.x], h[threadIdx.x]);
}
static __global__ __launch_bounds__(256) void setup_len_scatter_kernel(const void* __restrict__ coefs,
                                                                       const uint32_t* __restrict__ order, uint32_t nnz,
                                                                       const uint32_t* __restrict__ start,
                                                                       uint32_t* __restrict__ cursor,
                                                                       uint32_t* __restrict__ by_len) {
  // most entries share one key (|coefficient| = 1): ranks inside the workgroup through LDS, ONE global atomic per
  // (workgroup, key) -- a global atomic per entry on that one cursor took 37 ms for 4 M entries
  __shared__ uint32_t h[256], base[256];
  h[threadIdx.x] = 0;
  __syncthreads();
  const uint32_t j = blockIdx.x * 256u + threadIdx.x;
  uint32_t key = 0, rank = 0;
  if (j < nnz) {
    key = 255u - setup_bitlen(coefs, order[j]);
    rank = atomicAdd(&h[key], 1u);
  }
  __syncthreads();
  if (h[threadIdx.x]) base[threadIdx.x] = start[threadIdx.x] + atomicAdd(&cursor[threadIdx.x], h[threadIdx.x]);
  __syncthreads();
  if (j < nnz) by_len[base[key] + rank] = j;
}

// thread t takes slot j = by_len[t] (segment order): Q = coef * P; a segment of one entry goes straight to its bucket
template <class F>
static __global__ __launch_bounds__(256) void setup_mul_kernel(const void* __restrict__ points,
                                                               const void* __restrict__ coefs,
                                                               const uint32_t* __restrict__ pidx,
                                                               const uint32_t* __restrict__ sig,
                                                               const uint32_t* __restrict__ order,
                                                               const uint32_t* __restrict__ by_len,
                                                               const uint32_t* __restrict__ off, uint32_t nnz,
                                                               void* __restrict__ buckets, void* __restrict__ items) {
  uint32_t t = blockIdx.x * 256u + threadIdx.x;
  if (t >= nnz) return;
  const uint32_t j = by_len[t];
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

// ---- `snarkjs wtns check`: constraint c holds iff <A_c, w> * <B_c, w> == <C_c, w> ------------------------------------
// row_ptr[3 * c + m .. +1): the terms of matrix m of constraint c; coefficients and witness in Montgomery form.
// flags[0] = number of violated constraints, flags[1] = the smallest violated index (atomicMin, starts at ~0).
static __global__ __launch_bounds__(256) void wtns_check_kernel(const uint32_t* __restrict__ row_ptr,
                                                                const uint32_t* __restrict__ sig,
                                                                const void* __restrict__ coef_m,
                                                                const void* __restrict__ w_m, uint32_t n_cons,
                                                                uint32_t* __restrict__ flags) {
  uint32_t c = blockIdx.x * 256u + threadIdx.x;
  if (c >= n_cons) return;
  Fr v[3];
#pragma unroll
  for (int m = 0; m < 3; m++) {
    Fr acc = Fr::zero();
    for (uint32_t t = row_ptr[3 * c + m]; t < row_ptr[3 * c + m + 1]; t++)
      acc = acc + load_field<Fr>(reinterpret_cast<const char*>(coef_m) + 32 * (size_t)t) *
                      load_field<Fr>(reinterpret_cast<const char*>(w_m) + 32 * (size_t)sig[t]);
    v[m] = acc;
  }
  if (!(v[0] * v[1] - v[2]).is_zero()) {
    atomicAdd(&flags[0], 1u);
    atomicMin(&flags[1], c);
  }
}
// in place: standard form (canonical, < r checked by the caller's flag) -> Montgomery
static __global__ __launch_bounds__(256) void fr_to_mont_kernel(void* __restrict__ data, uint64_t n, uint32_t* __restrict__ bad) {
  uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
  if (i >= n) return;
  char* p = reinterpret_cast<char*>(data) + 32 * i;
  Fr v = load_field<Fr>(p);
  uint32_t bw = 0;
#pragma unroll
  for (int k = 0; k < 8; k++) (void)subb(v.l[k], FrParams::P[k], bw);
  if (!bw) atomicOr(bad, 1u);
  store_field(p, v.to_mont());
}

// out[i] = k * in[i] for one scalar k (sign-normalised by the caller: `neg` adds -P), XYZZ into scratch
struct ScalarArg {
  uint32_t l[8];
};
template <class F>
static __global__ __launch_bounds__(256) void setup_scale_kernel(const void* __restrict__ in, uint64_t i0, uint32_t cnt,
                                                                 ScalarArg k, int top, bool neg,
                                                                 void* __restrict__ out_xyzz) {
  uint32_t t = blockIdx.x * 256u + threadIdx.x;
  if (t >= cnt) return;
  const Affine<F> p = load_affine<F>(in, i0 + t);
  XYZZ<F> acc = XYZZ<F>::inf();
  for (int bit = top; bit >= 0; bit--) {   // the scalar is the same in every lane: no divergence
    acc = xyzz_dbl(acc);
    if ((k.l[bit >> 5] >> (bit & 31)) & 1u) xyzz_add_affine(acc, p, neg);
  }
  store_xyzz(out_xyzz, t, acc);
}

// d_out[i] = k * d_in[i], i < n, wire format in and out (k: 32 B little-endian standard form, < r)
template <class F>
void setup_scale(zkpoa_context* ctx, const void* d_in, uint64_t n, const uint8_t k_le[32], void* d_out) {
  if (n == 0) return;
  hipStream_t st = ctx->dev.lanes[0].stream;
  HFr kh = HFr::from_bytes(k_le);
  uint64_t half[4] = {0, 0, 0, 0};   // (r - 1) / 2
  {
    uint64_t c = 0;
    for (int i = 3; i >= 0; i--) {
      half[i] = (HFrParams::P[i] >> 1) | (c << 63);
      c = HFrParams::P[i] & 1;
    }
  }
  bool neg = false;
  for (int i = 3; i >= 0; i--) {
    if (kh.l[i] > half[i]) { neg = true; break; }
    if (kh.l[i] < half[i]) break;
  }
  if (neg) {   // k > r / 2: (r - k) * (-P)
    HFr r_minus = HFr{{HFrParams::P[0], HFrParams::P[1], HFrParams::P[2], HFrParams::P[3]}};
    unsigned __int128 bw = 0;
    for (int i = 0; i < 4; i++) {
      unsigned __int128 d = (unsigned __int128)r_minus.l[i] - kh.l[i] - bw;
      kh.l[i] = (uint64_t)d;
      bw = (d >> 64) & 1;
    }
  }
  ScalarArg ka;
  memcpy(ka.l, kh.l, 32);
  int top = -1;
  for (int i = 7; i >= 0 && top < 0; i--)
    if (ka.l[i]) top = 32 * i + (31 - __builtin_clz(ka.l[i]));
  const uint64_t slab = 1ull << 22;
  DevBuf scratch((size_t)(n < slab ? n : slab) * MsmSizes<F>::kXyzz);
  for (uint64_t off = 0; off < n; off += slab) {
    const uint32_t cnt = (uint32_t)(n - off < slab ? n - off : slab);
    hipLaunchKernelGGL((setup_scale_kernel<F>), dim3((cnt + 255) / 256), dim3(256), 0, st, d_in, off, cnt, ka, top, neg,
                       scratch.p);
    hipLaunchKernelGGL((xyzz_to_affine_kernel<F>), dim3((cnt + 255) / 256), dim3(256), 0, st, (const void*)scratch.p,
                       (void*)(reinterpret_cast<char*>(d_out) + off * MsmSizes<F>::kAffine), (uint64_t)cnt);
  }
  ZK_HIP(hipStreamSynchronize(st));
  ZK_HIP(hipGetLastError());
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
      items((size_t)(N ? N : 1) * X), items2(((size_t)N / 2 + 2) * X), by_len((size_t)(N ? N : 1) * 4), len_hist(1024 * 4);
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
    // slots ordered by the length of their double-and-add (256 keys: histogram, 256-entry scan on one wave, scatter)
    uint32_t* lh = reinterpret_cast<uint32_t*>(len_hist.p);   // [0, 256) counts, [256, 513) starts, [520, 776) cursors
    ZK_HIP(hipMemsetAsync(len_hist.p, 0, 1024 * 4, st));
    hipLaunchKernelGGL(setup_len_hist_kernel, dim3(grid), dim3(256), 0, st, d_coefs, (const uint32_t*)order.p, N, lh);
    scan_u32(st, lh, 256, 0, 0, lh + 256, (uint32_t*)block_sums.p, m + 3, nullptr);
    hipLaunchKernelGGL(setup_len_scatter_kernel, dim3(grid), dim3(256), 0, st, d_coefs, (const uint32_t*)order.p, N,
                       (const uint32_t*)(lh + 256), lh + 520, (uint32_t*)by_len.p);
    hipLaunchKernelGGL((setup_mul_kernel<F>), dim3(grid), dim3(256), 0, st, d_points, d_coefs, d_pidx, d_sig,
                       (const uint32_t*)order.p, (const uint32_t*)by_len.p, (const uint32_t*)off.p, N, buckets.p, items.p);
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

// ---- `snarkjs zkey new <circuit.r1cs> <pot.ptau> <circuit_0.zkey>` (g16_setup.sh:243-246) on files -----------------------
namespace {

struct SetupError : std::runtime_error {
  explicit SetupError(const std::string& m) : std::runtime_error(m) {}
};

uint32_t rd32(const uint8_t* p) { uint32_t v; memcpy(&v, p, 4); return v; }
uint64_t rd64(const uint8_t* p) { uint64_t v; memcpy(&v, p, 8); return v; }

struct MappedFile {
  const uint8_t* p = nullptr;
  uint64_t size = 0;
  int fd = -1;
  explicit MappedFile(const char* path) {
    fd = open(path, O_RDONLY);
    if (fd < 0) throw SetupError(std::string("cannot open ") + path);
    struct stat sb;
    if (fstat(fd, &sb) != 0 || sb.st_size <= 0) {
      close(fd);
      throw SetupError(std::string("cannot stat ") + path);
    }
    size = (uint64_t)sb.st_size;
    void* m = mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
    if (m == MAP_FAILED) {
      close(fd);
      throw SetupError(std::string("cannot map ") + path);
    }
    p = static_cast<const uint8_t*>(m);
  }
  ~MappedFile() {
    if (p) munmap(const_cast<uint8_t*>(p), size);
    if (fd >= 0) close(fd);
  }
  MappedFile(const MappedFile&) = delete;
  MappedFile& operator=(const MappedFile&) = delete;
};

// Output files are written under a temporary name and renamed into place (as prover_main's write_atomic): a failure
// part-way (ENOSPC, HIP error, kill) never leaves a truncated .zkey under the final name for a later "skip if the zkey
// exists" step to pick up, and writing over the input of `zkey contribute` is safe (the mapping keeps the old inode).
// ---- host-side parallelism (r04) ---------------------------------------------------------------------------------------
// At the layer-three shape `zkey new` reads 62 GB and writes 35 GB, and the device's share of the command is 3.5 s: what
// was left was one host thread parsing, copying and converting (54 s in all). Every host stage below is split over
// the host's threads; the arrays they fill are not zero-filled first (UVec), the threads are the first to touch them.
unsigned host_threads() {
  unsigned t = std::thread::hardware_concurrency();
  if (const char* e = getenv("ZKPOA_SETUP_THREADS")) {
    char* end = nullptr;
    const long v = strtol(e, &end, 10);
    if (end != e && !*end && v >= 1 && v <= 256) t = (unsigned)v;
  }
  return t < 1 ? 1 : (t > 32 ? 32 : t);
}
// fn(t, lo, hi) over [0, count) cut into host_threads() ranges; the first exception (in range order) is rethrown
template <class Fn>
void parallel_ranges(uint64_t count, uint64_t min_per_thread, Fn fn) {
  unsigned T = host_threads();
  if (count / (min_per_thread ? min_per_thread : 1) < T) T = (unsigned)(count / (min_per_thread ? min_per_thread : 1));
  if (T <= 1) {
    fn(0u, (uint64_t)0, count);
    return;
  }
  std::vector<std::exception_ptr> errs(T);
  std::vector<std::thread> th;
  for (unsigned t = 0; t < T; t++)
    th.emplace_back([&, t] {
      try {
        fn(t, count * t / T, count * (t + 1) / T);
      } catch (...) {
        errs[t] = std::current_exception();
      }
    });
  for (auto& x : th) x.join();
  for (auto& e : errs)
    if (e) std::rethrow_exception(e);
}
// One-shot commands (zkpoa-setup) leave the tens of GB of host arrays of a `zkey new` to the process's exit: giving
// 100 GB back page by page takes seconds that nobody is waiting for any more (zkpoa_setup_defer_host_frees).
std::atomic<bool>& defer_host_frees() {
  static std::atomic<bool> on{false};
  return on;
}
template <class T>
struct UVec {   // a sized array of trivially copyable elements whose storage is NOT value-initialised
  std::unique_ptr<T[]> p;
  size_t n = 0;
  UVec() = default;
  UVec(UVec&&) = default;
  UVec& operator=(UVec&&) = default;
  ~UVec() {
    if (defer_host_frees().load()) (void)p.release();
  }
  explicit UVec(size_t count) { alloc(count); }
  void alloc(size_t count) {
    p.reset(new T[count ? count : 1]);
    n = count;
  }
  size_t size() const { return n; }
  bool empty() const { return n == 0; }
  T* data() { return p.get(); }
  const T* data() const { return p.get(); }
  T& operator[](size_t i) { return p[i]; }
  const T& operator[](size_t i) const { return p[i]; }
  const T* begin() const { return p.get(); }
  const T* end() const { return p.get() + n; }
};

struct AtomicFile {
  std::string path, tmp;
  FILE* f = nullptr;
  bool ok = true;
  explicit AtomicFile(const char* final_path) : path(final_path), tmp(path + ".tmp." + std::to_string((long)getpid())) {
    f = fopen(tmp.c_str(), "wb");
    if (!f) throw SetupError("cannot create " + tmp);
  }
  void write(const void* p, size_t len) {
    if (ok && len) ok = fwrite(p, 1, len, f) == len;
  }
  // a payload of GBs: the stream is flushed, the file extended, and the bytes are copied by several threads through a
  // shared mapping of their final place (buffered write() calls to one file queue up behind its inode lock -- measured on
  // tmpfs: eight pwrite threads were no faster than one fwrite); the stream then continues behind them
  void write_large(const void* p, size_t len) {
    if (len < (64u << 20)) return write(p, len);
    if (!ok) return;
    if (fflush(f) != 0) {
      ok = false;
      return;
    }
    const off_t base = ftello(f);
    const int fd = fileno(f);
    if (base < 0 || ftruncate(fd, base + (off_t)len) != 0) {
      ok = false;
      return;
    }
    const off_t map_off = base & ~(off_t)4095;
    const size_t lead = (size_t)(base - map_off);
    void* m = mmap(nullptr, lead + len, PROT_READ | PROT_WRITE, MAP_SHARED, fd, map_off);
    if (m == MAP_FAILED) {   // a file system without shared mappings: the plain way
      ok = fseeko(f, base, SEEK_SET) == 0;
      return write(p, len);
    }
    char* dst = static_cast<char*>(m) + lead;
    parallel_ranges(len, 32ull << 20, [&](unsigned, uint64_t lo, uint64_t hi) { memcpy(dst + lo, static_cast<const char*>(p) + lo, hi - lo); });
    ok = munmap(m, lead + len) == 0 && fseeko(f, base + (off_t)len, SEEK_SET) == 0;
  }
  // Absolute placement, for a file whose section offsets are known before their content: reserve() sizes it, put_at()
  // may then be called from several threads for disjoint ranges in any order (small ranges: pwrite; large ones: the
  // shared-mapping copy of write_large). The FILE stream is not used in this mode.
  std::atomic<bool> placed_ok{true};
  void reserve(uint64_t total) {
    if (fflush(f) != 0 || ftruncate(fileno(f), (off_t)total) != 0) ok = false;
  }
  void put_at(uint64_t off, const void* p, uint64_t len) {
    if (!len) return;
    const int fd = fileno(f);
    if (len < (64u << 20)) {
      uint64_t done = 0;
      while (done < len) {
        const ssize_t w = pwrite(fd, static_cast<const char*>(p) + done, len - done, (off_t)(off + done));
        if (w <= 0) {
          placed_ok = false;
          return;
        }
        done += (uint64_t)w;
      }
      return;
    }
    const uint64_t map_off = off & ~4095ull, lead = off - map_off;
    void* m = mmap(nullptr, lead + len, PROT_READ | PROT_WRITE, MAP_SHARED, fd, (off_t)map_off);
    if (m == MAP_FAILED) {
      uint64_t done = 0;
      while (done < len) {
        const ssize_t w = pwrite(fd, static_cast<const char*>(p) + done, len - done, (off_t)(off + done));
        if (w <= 0) {
          placed_ok = false;
          return;
        }
        done += (uint64_t)w;
      }
      return;
    }
    char* dst = static_cast<char*>(m) + lead;
    parallel_ranges(len, 32ull << 20, [&](unsigned, uint64_t lo, uint64_t hi) { memcpy(dst + lo, static_cast<const char*>(p) + lo, hi - lo); });
    if (munmap(m, lead + len) != 0) placed_ok = false;
  }
  void commit() {
    ok = ok && placed_ok.load();
    ok = (fclose(f) == 0) && ok;
    f = nullptr;
    if (!ok || rename(tmp.c_str(), path.c_str()) != 0) {
      unlink(tmp.c_str());
      throw SetupError("write to " + path + " failed");
    }
  }
  ~AtomicFile() {
    if (f) {
      fclose(f);
      unlink(tmp.c_str());
    }
  }
  AtomicFile(const AtomicFile&) = delete;
  AtomicFile& operator=(const AtomicFile&) = delete;
};

// Untrusted inputs: the device keeps Fq elements lazily in [0, 2q) and takes file bytes raw, so every
// coordinate that reaches a kernel or is copied into the key must be a canonical field element (< q).
void host_check_coords(const uint8_t* p, uint64_t count32, const char* what) {
  parallel_ranges(count32, 1u << 20, [&](unsigned, uint64_t lo, uint64_t hi) {
    for (uint64_t i = lo; i < hi; i++) {
      uint64_t v[4];
      memcpy(v, p + 32 * i, 32);
      if (v[3] >= HFqParams::P[3] && HFq::geq_p(v)) throw SetupError(std::string(what) + ": a coordinate is not a field element (>= q)");
    }
  });
}
void dev_check_coords(zkpoa_context* ctx, const void* d, uint64_t count32, const char* what) {
  if (!count32) return;
  hipStream_t st = ctx->dev.lanes[0].stream;
  DevBuf flag(64);
  ZK_HIP(hipMemsetAsync(flag.p, 0, 64, st));
  hipLaunchKernelGGL((range_check_kernel<FqParams>), dim3((uint32_t)((count32 + 255) / 256)), dim3(256), 0, st, d, count32,
                     (uint32_t*)flag.p);
  uint32_t bad = 0;
  ZK_HIP(hipMemcpyAsync(&bad, flag.p, 4, hipMemcpyDeviceToHost, st));
  ZK_HIP(hipStreamSynchronize(st));
  ZK_HIP(hipGetLastError());
  if (bad) throw SetupError(std::string(what) + ": a coordinate is not a field element (>= q)");
}

struct Sec {
  uint64_t off = 0, len = 0;
  bool present = false;
};

// iden3 binary container: magic(4) version(u32) nSections(u32) then { type(u32) size(u64) payload }
std::map<uint32_t, Sec> bin_sections(const MappedFile& f, const char* magic, uint32_t max_version, const char* what) {
  if (f.size < 12 || memcmp(f.p, magic, 4) != 0) throw SetupError(std::string(what) + ": bad magic");
  if (rd32(f.p + 4) > max_version) throw SetupError(std::string(what) + ": unsupported version");
  const uint32_t n = rd32(f.p + 8);
  std::map<uint32_t, Sec> out;
  uint64_t pos = 12;
  for (uint32_t i = 0; i < n; i++) {
    if (pos + 12 > f.size) throw SetupError(std::string(what) + ": truncated section table");
    const uint32_t type = rd32(f.p + pos);
    const uint64_t len = rd64(f.p + pos + 4);
    pos += 12;
    if (len > f.size - pos) throw SetupError(std::string(what) + ": section runs past the end of the file");
    if (!out.count(type)) out[type] = Sec{pos, len, true};   // the first section of a type (as snarkjs' readers)
    pos += len;
  }
  return out;
}

struct Term {
  uint32_t c, s;
  uint8_t coef[32];
};

struct R1cs {
  uint32_t nWires = 0, nPublic = 0, nConstraints = 0;
  UVec<Term> A, B, C;   // per constraint in file order, terms of one linear combination ascending by signal
};

// Two passes: the first walks the term counts only (three words per constraint) and notes, every 2^14 constraints, where
// the constraint starts in the file and how many A / B / C terms precede it; the second parses those blocks in parallel
// straight into their places.
R1cs parse_r1cs(const MappedFile& f) {
  auto secs = bin_sections(f, "r1cs", 1, "r1cs");
  if (!secs.count(1) || !secs.count(2)) throw SetupError("r1cs: header or constraint section missing");
  const Sec h = secs[1], cs = secs[2];
  if (h.len < 4 + 32 + 4 * 4 + 8 + 4 || rd32(f.p + h.off) != 32) throw SetupError("r1cs: header too short or field size != 32");
  for (int i = 0; i < 4; i++)
    if (rd64(f.p + h.off + 4 + 8 * i) != HFrParams::P[i]) throw SetupError("r1cs: not over the BN254 scalar field");
  const uint8_t* q = f.p + h.off + 36;
  R1cs r;
  r.nWires = rd32(q);
  const uint64_t n_public = (uint64_t)rd32(q + 4) + rd32(q + 8);   // outputs + public inputs (no 32-bit wrap)
  r.nConstraints = rd32(q + 24);
  if (r.nWires == 0 || n_public + 1 > r.nWires) throw SetupError("r1cs: inconsistent wire counts");
  r.nPublic = (uint32_t)n_public;
  const uint64_t end = cs.off + cs.len;
  constexpr uint32_t kBlock = 1u << 14;
  struct Mark { uint64_t pos, cnt[3]; };
  std::vector<Mark> marks;
  marks.reserve(r.nConstraints / kBlock + 2);
  {
    uint64_t pos = cs.off, cnt[3] = {0, 0, 0};
    for (uint32_t c = 0; c < r.nConstraints; c++) {
      if ((c & (kBlock - 1)) == 0) marks.push_back(Mark{pos, {cnt[0], cnt[1], cnt[2]}});
      for (int m = 0; m < 3; m++) {
        if (pos + 4 > end) throw SetupError("r1cs: constraint section truncated");
        const uint32_t nt = rd32(f.p + pos);
        pos += 4;
        if ((uint64_t)nt * 36 > end - pos) throw SetupError("r1cs: constraint section truncated");
        pos += (uint64_t)nt * 36;
        cnt[m] += nt;
      }
    }
    marks.push_back(Mark{pos, {cnt[0], cnt[1], cnt[2]}});
  }
  const Mark& tot = marks.back();
  r.A.alloc(tot.cnt[0]);
  r.B.alloc(tot.cnt[1]);
  r.C.alloc(tot.cnt[2]);
  const uint64_t blocks = marks.size() - 1;
  parallel_ranges(blocks, 1, [&](unsigned, uint64_t b0, uint64_t b1) {
    for (uint64_t b = b0; b < b1; b++) {
      uint64_t pos = marks[b].pos, at[3] = {marks[b].cnt[0], marks[b].cnt[1], marks[b].cnt[2]};
      const uint32_t c1 = (uint32_t)std::min<uint64_t>((b + 1) * kBlock, r.nConstraints);
      for (uint32_t c = (uint32_t)(b * kBlock); c < c1; c++) {
        for (int m = 0; m < 3; m++) {
          const uint32_t nt = rd32(f.p + pos);
          pos += 4;
          Term* dst = (m == 0 ? r.A.data() : (m == 1 ? r.B.data() : r.C.data())) + at[m];
          for (uint32_t t = 0; t < nt; t++, pos += 36) {
            Term& x = dst[t];
            x.c = c;
            x.s = rd32(f.p + pos);
            memcpy(x.coef, f.p + pos + 4, 32);
            if (x.s >= r.nWires) throw SetupError("r1cs: wire index out of range");
            uint64_t v[4];
            memcpy(v, x.coef, 32);
            if (HFr::geq_p(v)) throw SetupError("r1cs: coefficient is not a field element (>= r)");
          }
          // snarkjs holds a linear combination as an object keyed by the signal: iteration is ascending by signal
          std::stable_sort(dst, dst + nt, [](const Term& a, const Term& b) { return a.s < b.s; });
          for (uint32_t t = 1; t < nt; t++)
            if (dst[t].s == dst[t - 1].s) throw SetupError("r1cs: a signal occurs twice in one linear combination");
          at[m] += nt;
        }
      }
    }
  });
  return r;
}

void pread_all(int fd, void* dst, uint64_t len, uint64_t off, const char* what) {
  parallel_ranges(len, 32ull << 20, [&](unsigned, uint64_t lo, uint64_t hi) {
    uint64_t got = lo;
    while (got < hi) {
      ssize_t n = pread(fd, static_cast<char*>(dst) + got, hi - got, (off_t)(off + got));
      if (n <= 0) throw SetupError(std::string("ptau: short read of ") + what);
      got += (uint64_t)n;
    }
  });
}

struct DevArr {
  void* p = nullptr;
  explicit DevArr(size_t bytes) { ZK_HIP(hipMalloc(&p, bytes ? bytes : 1)); }
  ~DevArr() { if (p) (void)hipFree(p); }
  DevArr(const DevArr&) = delete;
  DevArr& operator=(const DevArr&) = delete;
  void up(const void* src, size_t bytes, size_t at = 0) {
    if (bytes) ZK_HIP(hipMemcpy(static_cast<char*>(p) + at, src, bytes, hipMemcpyHostToDevice));
  }
};

struct Entries {   // one zkpoa_setup_accumulate call; filled in place by several threads (set)
  UVec<uint8_t> coef;
  UVec<uint32_t> pidx, sig;
  void alloc(uint64_t count) {
    coef.alloc(count * 32);
    pidx.alloc(count);
    sig.alloc(count);
  }
  void set(uint64_t i, const Term& t, uint32_t point_offset) {
    memcpy(&coef[i * 32], t.coef, 32);
    pidx[i] = point_offset + t.c;
    sig[i] = t.s;
  }
  void set_one(uint64_t i, uint32_t point, uint32_t signal) {
    memset(&coef[i * 32], 0, 32);
    coef[i * 32] = 1;
    pidx[i] = point;
    sig[i] = signal;
  }
};

template <class F>
UVec<uint8_t> run_accumulate(zkpoa_context* ctx, const DevArr& points, uint64_t n_points, const Entries& e,
                                    uint64_t n_signals) {
  constexpr size_t A = MsmSizes<F>::kAffine;
  const uint64_t nnz = e.sig.size();
  DevArr coef(nnz * 32), pidx(nnz * 4), sig(nnz * 4), out(n_signals * A);
  coef.up(e.coef.data(), nnz * 32);
  pidx.up(e.pidx.data(), nnz * 4);
  sig.up(e.sig.data(), nnz * 4);
  setup_accumulate<F>(ctx, points.p, n_points, coef.p, (const uint32_t*)pidx.p, (const uint32_t*)sig.p, nnz, n_signals, out.p);
  UVec<uint8_t> host(n_signals * A);
  if (!host.empty()) ZK_HIP(hipMemcpy(host.data(), out.p, host.size(), hipMemcpyDeviceToHost));
  return host;
}

void put32(std::vector<uint8_t>& v, uint32_t x) { v.insert(v.end(), (uint8_t*)&x, (uint8_t*)&x + 4); }
void put64(std::vector<uint8_t>& v, uint64_t x) { v.insert(v.end(), (uint8_t*)&x, (uint8_t*)&x + 8); }

void zkey_new(zkpoa_context* ctx, const char* r1cs_path, const char* ptau_path, const char* zkey_path) {
  const bool verbose = getenv("ZKPOA_VERBOSE") != nullptr;
  struct timespec tp0;
  clock_gettime(CLOCK_MONOTONIC, &tp0);
  auto phase = [&](const char* what) {   // ZKPOA_VERBOSE: where the command spends its time
    if (!verbose) return;
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    fprintf(stderr, "zkpoa: zkey new: %-38s %8.1f ms\n", what, (t.tv_sec - tp0.tv_sec) * 1e3 + (t.tv_nsec - tp0.tv_nsec) / 1e6);
    tp0 = t;
  };
  MappedFile fr(r1cs_path);
  const R1cs r = parse_r1cs(fr);
  phase("r1cs parsed");
  // domain: the smallest power of two that holds the constraints and the nPublic + 1 extra rows (zkey_new.js)
  uint32_t cp = 0;
  while ((1ull << cp) < (uint64_t)r.nConstraints + r.nPublic + 1) cp++;
  if (cp > 27) throw SetupError("circuit too large (domain above 2^27)");
  const uint64_t n = 1ull << cp;

  MappedFile fp(ptau_path);   // mapped for the section table only; the point ranges are read with pread
  auto ps = bin_sections(fp, "ptau", 1, "ptau");
  for (uint32_t t : {1u, 4u, 5u, 6u, 12u, 13u, 14u, 15u})
    if (!ps.count(t)) throw SetupError("ptau: section " + std::to_string(t) + " missing (the file must be prepared for phase 2)");
  const Sec h = ps[1];
  if (h.len < 4 + 32 + 8 || rd32(fp.p + h.off) != 32) throw SetupError("ptau: header too short or field size != 32");
  for (int i = 0; i < 4; i++)
    if (rd64(fp.p + h.off + 4 + 8 * i) != HFqParams::P[i]) throw SetupError("ptau: not a BN254 ceremony");
  const uint32_t power = rd32(fp.p + h.off + 36);
  if (cp > power) throw SetupError("ptau: ceremony of 2^" + std::to_string(power) + " is too small for a 2^" + std::to_string(cp) + " domain");
  auto level = [&](uint32_t sec, uint32_t lvl, uint64_t unit, uint64_t count, const char* what) {
    const uint64_t off = ((1ull << lvl) - 1) * unit;
    if (off + count * unit > ps[sec].len) throw SetupError(std::string("ptau: section too short for ") + what);
    return ps[sec].off + off;
  };
  UVec<uint8_t> L1(n * 64), L2(n * 128), aL(n * 64), bL(n * 64), Hs(2 * n * 64);
  pread_all(fp.fd, L1.data(), L1.size(), level(12, cp, 64, n, "tau*G1 (Lagrange)"), "tau*G1 (Lagrange)");
  pread_all(fp.fd, L2.data(), L2.size(), level(13, cp, 128, n, "tau*G2 (Lagrange)"), "tau*G2 (Lagrange)");
  pread_all(fp.fd, aL.data(), aL.size(), level(14, cp, 64, n, "alpha*tau*G1 (Lagrange)"), "alpha*tau*G1 (Lagrange)");
  pread_all(fp.fd, bL.data(), bL.size(), level(15, cp, 64, n, "beta*tau*G1 (Lagrange)"), "beta*tau*G1 (Lagrange)");
  pread_all(fp.fd, Hs.data(), Hs.size(), level(12, cp + 1, 64, 2 * n, "tau*G1 (Lagrange, 2n)"), "tau*G1 (Lagrange, 2n)");
  uint8_t alpha1[64], beta1[64], beta2[128];
  if (ps[4].len < 64 || ps[5].len < 64 || ps[6].len < 128) throw SetupError("ptau: alpha / beta sections too short");
  pread_all(fp.fd, alpha1, 64, ps[4].off, "alpha*G1");
  pread_all(fp.fd, beta1, 64, ps[5].off, "beta*G1");
  pread_all(fp.fd, beta2, 128, ps[6].off, "beta*G2");
  host_check_coords(alpha1, 2, "ptau alpha*G1");
  host_check_coords(beta1, 2, "ptau beta*G1");
  host_check_coords(beta2, 4, "ptau beta*G2");
  host_check_coords(Hs.data(), 4 * n, "ptau tau*G1 (Lagrange, 2n)");   // copied into section 9 without touching the device
  phase("ptau ranges read");

  // entries of the three accumulations (+ the nPublic + 1 rows `1 * signal_i` that bind the public inputs)
  Entries eA, eB, eK;
  const uint64_t nA = r.A.size(), nB = r.B.size(), nCt = r.C.size(), nPub1 = (uint64_t)r.nPublic + 1;
  eA.alloc(nA + nPub1);
  eB.alloc(nB);
  eK.alloc(nA + nB + nCt + nPub1);
  parallel_ranges(nA, 1u << 16, [&](unsigned, uint64_t lo, uint64_t hi) {
    for (uint64_t i = lo; i < hi; i++) {
      eA.set(i, r.A[i], 0);
      eK.set(i, r.A[i], 0);                                      // K: A over beta*L  (points [0, n))
    }
  });
  parallel_ranges(nB, 1u << 16, [&](unsigned, uint64_t lo, uint64_t hi) {
    for (uint64_t i = lo; i < hi; i++) {
      eB.set(i, r.B[i], 0);
      eK.set(nA + i, r.B[i], (uint32_t)n);                       //    B over alpha*L ([n, 2n))
    }
  });
  parallel_ranges(nCt, 1u << 16, [&](unsigned, uint64_t lo, uint64_t hi) {
    for (uint64_t i = lo; i < hi; i++) eK.set(nA + nB + i, r.C[i], (uint32_t)(2 * n));   //    C over L       ([2n, 3n))
  });
  for (uint32_t i = 0; i <= r.nPublic; i++) {
    eA.set_one(nA + i, r.nConstraints + i, i);
    eK.set_one(nA + nB + nCt + i, r.nConstraints + i, i);
  }
  phase("entry lists built");
  const uint64_t m = r.nWires;
  // ---- the file: sections 1-10 in the order snarkjs numbers them. Every length follows from the header and the term
  // counts, so the file is sized now and each section is put at its place when it is ready: the host's own sections
  // (header, coefficients, H) by a side thread WHILE the device computes the others.
  const uint64_t nCoefs = r.A.size() + r.B.size() + r.nPublic + 1;
  if (nCoefs > 0xffffffffull) throw SetupError("more than 2^32 coefficients");
  const uint64_t kS2 = 4 + 32 + 4 + 32 + 12 + 64 + 64 + 128 + 128 + 64 + 128, icb = ((uint64_t)r.nPublic + 1) * 64;
  const uint64_t sec_len[11] = {0, 4, kS2, icb, 4 + nCoefs * 44, m * 64, m * 64, m * 128, (m - r.nPublic - 1) * 64, n * 64, 64 + 4};
  uint64_t sec_off[11], total_out = 12;
  for (uint32_t t = 1; t <= 10; t++) {
    sec_off[t] = total_out + 12;
    total_out += 12 + sec_len[t];
  }
  AtomicFile fo(zkey_path);
  fo.reserve(total_out);
  auto put_section = [&](uint32_t id, const void* p, uint64_t len) {
    if (len != sec_len[id]) throw SetupError("internal: section " + std::to_string(id) + " has an unexpected size");
    fo.put_at(sec_off[id] - 12, &id, 4);
    fo.put_at(sec_off[id] - 8, &len, 8);
    fo.put_at(sec_off[id], p, len);
  };
  std::exception_ptr host_err;
  double host_ms = 0;
  std::thread host_sections([&] {
    try {
      struct timespec h0, h1;
      clock_gettime(CLOCK_MONOTONIC, &h0);
      const uint32_t hdr[2] = {1, 10};
      fo.put_at(0, "zkey", 4);
      fo.put_at(4, hdr, 8);
      const uint32_t one_u32 = 1;   // section 1: protocol id 1 = groth16
      put_section(1, &one_u32, 4);
      std::vector<uint8_t> s2, s10(64 + 4, 0);
      put32(s2, 32);
      s2.insert(s2.end(), (const uint8_t*)HFqParams::P, (const uint8_t*)HFqParams::P + 32);
      put32(s2, 32);
      s2.insert(s2.end(), (const uint8_t*)HFrParams::P, (const uint8_t*)HFrParams::P + 32);
      put32(s2, r.nWires);
      put32(s2, r.nPublic);
      put32(s2, (uint32_t)n);
      uint8_t g1[64], g2[128];
      h_affine_to_bytes<HFq>(host_generator<HFq>(), g1);
      h_affine_to_bytes<HFq2>(host_generator<HFq2>(), g2);
      s2.insert(s2.end(), alpha1, alpha1 + 64);
      s2.insert(s2.end(), beta1, beta1 + 64);
      s2.insert(s2.end(), beta2, beta2 + 128);
      s2.insert(s2.end(), g2, g2 + 128);   // gamma2 = the generator until a contribution changes delta
      s2.insert(s2.end(), g1, g1 + 64);    // delta1
      s2.insert(s2.end(), g2, g2 + 128);   // delta2
      put_section(2, s2.data(), s2.size());
      put_section(10, s10.data(), s10.size());
      // coefficients: A and B terms per constraint, then the public rows; values scaled by R^2
      UVec<uint8_t> s4(4 + nCoefs * 44);
      {
        const uint32_t nc32 = (uint32_t)nCoefs;
        memcpy(s4.data(), &nc32, 4);
      }
      auto rec = [&](uint64_t at, uint32_t mtx, uint32_t c, uint32_t sgn, const uint8_t* coef) {
        uint8_t* o = s4.data() + 4 + at * 44;
        memcpy(o, &mtx, 4);
        memcpy(o + 4, &c, 4);
        memcpy(o + 8, &sgn, 4);
        HFr v = HFr::from_bytes(coef).to_mont();   // limbs = coef * R
        HFr w = v.to_mont();                       // limbs = coef * R^2
        memcpy(o + 12, w.l, 32);
      };
      // constraint ranges in parallel: a range starts at the first A / B term of its first constraint, and its records
      // start at the count of A and B terms before that
      parallel_ranges(r.nConstraints, 1u << 14, [&](unsigned, uint64_t c0, uint64_t c1) {
        auto first = [&](const UVec<Term>& v, uint32_t c) {
          return (size_t)(std::lower_bound(v.begin(), v.end(), c, [](const Term& t, uint32_t key) { return t.c < key; }) - v.begin());
        };
        size_t ia = first(r.A, (uint32_t)c0), ib = first(r.B, (uint32_t)c0);
        uint64_t at = ia + ib;
        for (uint32_t c = (uint32_t)c0; c < (uint32_t)c1; c++) {
          for (; ia < r.A.size() && r.A[ia].c == c; ia++) rec(at++, 0, c, r.A[ia].s, r.A[ia].coef);
          for (; ib < r.B.size() && r.B[ib].c == c; ib++) rec(at++, 1, c, r.B[ib].s, r.B[ib].coef);
        }
      });
      {
        const uint8_t one[32] = {1};
        for (uint32_t i = 0; i <= r.nPublic; i++) rec(r.A.size() + r.B.size() + i, 0, r.nConstraints + i, i, one);
      }
      put_section(4, s4.data(), s4.size());
      UVec<uint8_t> s9(n * 64);
      parallel_ranges(n, 1u << 18, [&](unsigned, uint64_t lo, uint64_t hi) {
        for (uint64_t i = lo; i < hi; i++) memcpy(&s9[i * 64], &Hs[(2 * i + 1) * 64], 64);   // odd points of the 2n basis
      });
      put_section(9, s9.data(), s9.size());
      clock_gettime(CLOCK_MONOTONIC, &h1);
      host_ms = (h1.tv_sec - h0.tv_sec) * 1e3 + (h1.tv_nsec - h0.tv_nsec) / 1e6;
    } catch (...) {
      host_err = std::current_exception();
    }
  });
  // a point section is handed to a writer thread as soon as it is back from the device: it goes into the file while the
  // device works on the next one
  std::vector<std::thread> writers;
  std::mutex werr_mutex;
  std::exception_ptr write_err;
  auto write_async = [&](uint32_t id, const uint8_t* p, uint64_t len) {
    writers.emplace_back([&, id, p, len] {
      try {
        put_section(id, p, len);
      } catch (...) {
        std::lock_guard<std::mutex> lk(werr_mutex);
        if (!write_err) write_err = std::current_exception();
      }
    });
  };
  struct Joiner {   // (the device stage below may throw: the side threads are joined whatever happens)
    std::thread& t;
    std::vector<std::thread>& w;
    ~Joiner() {
      if (t.joinable()) t.join();
      for (auto& x : w)
        if (x.joinable()) x.join();
    }
  } joiner{host_sections, writers};
  UVec<uint8_t> secA, secB1, secB2, secK;   // (alive until every writer has been joined)
  {
    DevArr dL1(n * 64);
    dL1.up(L1.data(), L1.size());
    dev_check_coords(ctx, dL1.p, 2 * n, "ptau tau*G1 (Lagrange)");
    secA = run_accumulate<Fq>(ctx, dL1, n, eA, m);
    write_async(5, secA.data(), secA.size());
    secB1 = run_accumulate<Fq>(ctx, dL1, n, eB, m);
    write_async(6, secB1.data(), secB1.size());
  }
  {
    DevArr dL2(n * 128);
    dL2.up(L2.data(), L2.size());
    dev_check_coords(ctx, dL2.p, 4 * n, "ptau tau*G2 (Lagrange)");
    secB2 = run_accumulate<Fq2>(ctx, dL2, n, eB, m);
    write_async(7, secB2.data(), secB2.size());
  }
  {
    DevArr dK(3 * n * 64);
    dK.up(bL.data(), n * 64, 0);
    dK.up(aL.data(), n * 64, n * 64);
    dK.up(L1.data(), n * 64, 2 * n * 64);
    dev_check_coords(ctx, dK.p, 4 * n, "ptau alpha*tau*G1 / beta*tau*G1 (Lagrange)");
    secK = run_accumulate<Fq>(ctx, dK, 3 * n, eK, m);
  }
  phase("point sections (upload, device, download)");
  put_section(3, secK.data(), icb);
  put_section(8, secK.data() + icb, secK.size() - icb);
  host_sections.join();
  for (auto& x : writers) x.join();
  if (host_err) std::rethrow_exception(host_err);
  if (write_err) std::rethrow_exception(write_err);
  if (verbose) fprintf(stderr, "zkpoa: zkey new: (header, coefficient and H sections built and written by a side thread meanwhile: %.1f ms)\n", host_ms);
  fo.commit();
  phase("last sections written, key renamed into place");
}

// ---- `snarkjs wtns check <circuit.r1cs> <witness.wtns>` (g16_verify.sh:205-210) ---------------------------------------------
// returns the number of violated constraints; *first_bad = the smallest violated constraint index
uint64_t wtns_check(zkpoa_context* ctx, const char* r1cs_path, const char* wtns_path, uint64_t* first_bad) {
  MappedFile fr(r1cs_path);
  const R1cs r = parse_r1cs(fr);
  MappedFile fw(wtns_path);
  auto ws = bin_sections(fw, "wtns", 2, "wtns");
  if (!ws.count(1) || !ws.count(2)) throw SetupError("wtns: header or data section missing");
  const Sec wh = ws[1], wd = ws[2];
  if (wh.len != 4 + 32 + 4 || rd32(fw.p + wh.off) != 32) throw SetupError("wtns: header has the wrong size");
  for (int i = 0; i < 4; i++)
    if (rd64(fw.p + wh.off + 4 + 8 * i) != HFrParams::P[i]) throw SetupError("wtns: not over the BN254 scalar field");
  const uint64_t nw = rd32(fw.p + wh.off + 36);
  if (nw != r.nWires) throw SetupError("wtns: " + std::to_string(nw) + " values for a circuit of " + std::to_string(r.nWires) + " wires");
  if (wd.len != nw * 32) throw SetupError("wtns: data section has the wrong size");
  // CSR over (constraint, matrix): the three term lists are already grouped by constraint in file order
  const uint64_t nnz = r.A.size() + r.B.size() + r.C.size();
  if (nnz >= (1ull << 32)) throw SetupError("more than 2^32 coefficients");
  std::vector<uint32_t> row_ptr(3 * (size_t)r.nConstraints + 1, 0), sig(nnz);
  std::vector<uint8_t> coef(nnz * 32);
  {
    size_t ia = 0, ib = 0, ic = 0, t = 0;
    auto take = [&](const UVec<Term>& v, size_t& i, uint32_t c) {
      for (; i < v.size() && v[i].c == c; i++, t++) {
        sig[t] = v[i].s;
        memcpy(&coef[t * 32], v[i].coef, 32);
      }
    };
    for (uint32_t c = 0; c < r.nConstraints; c++) {
      row_ptr[3 * (size_t)c] = (uint32_t)t;
      take(r.A, ia, c);
      row_ptr[3 * (size_t)c + 1] = (uint32_t)t;
      take(r.B, ib, c);
      row_ptr[3 * (size_t)c + 2] = (uint32_t)t;
      take(r.C, ic, c);
    }
    row_ptr[3 * (size_t)r.nConstraints] = (uint32_t)t;
  }
  hipStream_t st = ctx->dev.lanes[0].stream;
  DevArr d_rp(row_ptr.size() * 4), d_sig(nnz * 4), d_coef(nnz * 32), d_w(nw * 32), d_flags(64);
  d_rp.up(row_ptr.data(), row_ptr.size() * 4);
  d_sig.up(sig.data(), nnz * 4);
  d_coef.up(coef.data(), nnz * 32);
  d_w.up(fw.p + wd.off, nw * 32);
  const uint32_t init[4] = {0, 0xffffffffu, 0, 0};
  d_flags.up(init, 16);
  uint32_t* fl = reinterpret_cast<uint32_t*>(d_flags.p);
  if (nnz) hipLaunchKernelGGL(fr_to_mont_kernel, dim3((uint32_t)((nnz + 255) / 256)), dim3(256), 0, st, d_coef.p, nnz, fl + 2);
  hipLaunchKernelGGL(fr_to_mont_kernel, dim3((uint32_t)((nw + 255) / 256)), dim3(256), 0, st, d_w.p, nw, fl + 3);
  if (r.nConstraints)
    hipLaunchKernelGGL(wtns_check_kernel, dim3((r.nConstraints + 255) / 256), dim3(256), 0, st, (const uint32_t*)d_rp.p,
                       (const uint32_t*)d_sig.p, (const void*)d_coef.p, (const void*)d_w.p, r.nConstraints, fl);
  msm_read_back(ctx->dev.lanes[0], fl, 16);
  ZK_HIP(hipGetLastError());
  const uint32_t* hb = reinterpret_cast<const uint32_t*>(ctx->dev.lanes[0].pinned);
  if (hb[3]) throw SetupError("wtns: a value is not a field element (>= r)");
  if (hb[2]) throw SetupError("r1cs: a coefficient is not a field element (>= r)");
  if (first_bad) *first_bad = hb[1];
  return hb[0];
}

// ---- the arithmetic of `snarkjs zkey contribute` (g16_setup.sh:262-266): delta <- d * delta, C and H <- C, H / d ---------
void zkey_contribute(zkpoa_context* ctx, const char* in_path, const char* out_path, const uint8_t* delta_le) {
  MappedFile fi(in_path);
  auto secs = bin_sections(fi, "zkey", 1, "zkey");
  for (uint32_t t = 1; t <= 10; t++)
    if (!secs.count(t)) throw SetupError("zkey: section " + std::to_string(t) + " missing");
  const Sec h = secs[2];
  const uint64_t kHdr = 4 + 32 + 4 + 32 + 12, kDelta1 = kHdr + 64 + 64 + 128 + 128, kDelta2 = kDelta1 + 64;
  if (h.len != kDelta2 + 128 || rd32(fi.p + h.off) != 32 || rd32(fi.p + h.off + 36) != 32)
    throw SetupError("zkey: groth16 header has the wrong size");
  for (int i = 0; i < 4; i++)
    if (rd64(fi.p + h.off + 4 + 8 * i) != HFqParams::P[i] || rd64(fi.p + h.off + 40 + 8 * i) != HFrParams::P[i])
      throw SetupError("zkey: not a BN254 key");
  const uint64_t nVars = rd32(fi.p + h.off + 72), nPublic = rd32(fi.p + h.off + 76), domain = rd32(fi.p + h.off + 80);
  if (nPublic + 1 > nVars || secs[8].len != (nVars - nPublic - 1) * 64 || secs[9].len != domain * 64)
    throw SetupError("zkey: C or H section has the wrong size");
  uint8_t d[32];
  if (delta_le) memcpy(d, delta_le, 32);
  else {   // uniform on [1, r): 254 random bits, rejected while >= r or zero
    int fd = open("/dev/urandom", O_RDONLY);
    if (fd < 0) throw SetupError("cannot open /dev/urandom");
    for (;;) {
      if (read(fd, d, 32) != 32) {
        close(fd);
        throw SetupError("short read from /dev/urandom");
      }
      d[31] &= 0x3f;
      uint64_t v[4];
      memcpy(v, d, 32);
      if (!HFr::geq_p(v) && (v[0] | v[1] | v[2] | v[3])) break;
    }
    close(fd);
  }
  uint64_t dv[4];
  memcpy(dv, d, 32);
  if (HFr::geq_p(dv) || !(dv[0] | dv[1] | dv[2] | dv[3])) throw SetupError("contribute: delta must be in [1, r)");
  HFr dinv = HFr::from_bytes(d).to_mont().inv().from_mont();
  uint8_t dinv_le[32];
  memcpy(dinv_le, dinv.l, 32);

  std::vector<uint8_t> s2(fi.p + h.off, fi.p + h.off + h.len);
  host_check_coords(&s2[kDelta1], 2 + 4, "zkey delta1 / delta2");
  {
    Affine<HFq> d1 = h_affine_from_bytes<HFq>(&s2[kDelta1]);
    Affine<HFq2> d2 = h_affine_from_bytes<HFq2>(&s2[kDelta2]);
    h_affine_to_bytes<HFq>(h_to_affine(h_mul(XYZZ<HFq>::from_affine(d1), dv)), &s2[kDelta1]);
    h_affine_to_bytes<HFq2>(h_to_affine(h_mul(XYZZ<HFq2>::from_affine(d2), dv)), &s2[kDelta2]);
  }
  auto scaled = [&](const Sec& sc) {
    UVec<uint8_t> out(sc.len);
    if (sc.len) {
      DevArr in(sc.len), res(sc.len);
      in.up(fi.p + sc.off, sc.len);
      dev_check_coords(ctx, in.p, sc.len / 32, "zkey C / H section");
      setup_scale<Fq>(ctx, in.p, sc.len / 64, dinv_le, res.p);
      ZK_HIP(hipMemcpy(out.data(), res.p, sc.len, hipMemcpyDeviceToHost));
    }
    return out;
  };
  // the output is sized up front (the section lengths do not change): the sections that are copied as they are go into
  // it from a side thread while the device scales C and H
  AtomicFile fo(out_path);   // temporary name + rename: in_path == out_path is fine (the mapping keeps the old inode)
  uint64_t off_of[11], total_out = 12;
  for (uint32_t t = 1; t <= 10; t++) {
    off_of[t] = total_out + 12;
    total_out += 12 + secs[t].len;
  }
  fo.reserve(total_out);
  auto put_section = [&](uint32_t t, const uint8_t* p) {
    const uint64_t len = secs[t].len;
    fo.put_at(off_of[t] - 12, &t, 4);
    fo.put_at(off_of[t] - 8, &len, 8);
    fo.put_at(off_of[t], p, len);
  };
  std::exception_ptr copy_err;
  std::thread copier([&] {
    try {
      const uint32_t hdr[2] = {1, 10};
      fo.put_at(0, "zkey", 4);
      fo.put_at(4, hdr, 8);
      put_section(2, s2.data());
      for (uint32_t t : {1u, 3u, 4u, 5u, 6u, 7u, 10u}) put_section(t, fi.p + secs[t].off);
    } catch (...) {
      copy_err = std::current_exception();
    }
  });
  struct Joiner {
    std::thread& t;
    ~Joiner() {
      if (t.joinable()) t.join();
    }
  } joiner{copier};
  {
    const UVec<uint8_t> s8 = scaled(secs[8]);
    put_section(8, s8.data());
  }
  {
    const UVec<uint8_t> s9 = scaled(secs[9]);
    put_section(9, s9.data());
  }
  copier.join();
  if (copy_err) std::rethrow_exception(copy_err);
  fo.commit();
}

}  // namespace

extern "C" int zkpoa_wtns_check(zkpoa_context* ctx, const char* r1cs_path, const char* wtns_path, uint64_t* violated,
                                uint64_t* first_violated) {
  ZK_API_BEGIN(ctx)
  if (!r1cs_path || !wtns_path || !violated) throw SetupError("wtns check: null argument");
  *violated = wtns_check(ctx, r1cs_path, wtns_path, first_violated);
  ZK_API_END(ctx)
}

extern "C" int zkpoa_zkey_contribute(zkpoa_context* ctx, const char* zkey_in_path, const char* zkey_out_path,
                                     const uint8_t* delta_le) {
  ZK_API_BEGIN(ctx)
  if (!zkey_in_path || !zkey_out_path) throw SetupError("zkey contribute: null path");
  zkey_contribute(ctx, zkey_in_path, zkey_out_path, delta_le);
  ZK_API_END(ctx)
}

extern "C" void zkpoa_setup_defer_host_frees(int on) { defer_host_frees() = on != 0; }

extern "C" int zkpoa_zkey_new(zkpoa_context* ctx, const char* r1cs_path, const char* ptau_path, const char* zkey_path) {
  ZK_API_BEGIN(ctx)
  if (!r1cs_path || !ptau_path || !zkey_path) throw SetupError("zkey new: null path");
  zkey_new(ctx, r1cs_path, ptau_path, zkey_path);
  ZK_API_END(ctx)
}
