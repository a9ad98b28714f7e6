// Radix-2 number-theoretic transform over the BN254 scalar field Fr for gfx950, LDS-staged.
//
// Replaces Fr.fft / Fr.ifft / Fr.batchApplyKey of ffjavascript as used by snarkjs
// groth16_prove.js and rapidsnark's fft.hpp: six size-n
// transforms per proof, w = w[log2 n] (5^((r-1)/2^28) squared down), elements in Montgomery form.
//
// Structure: the log2(n) butterfly stages are cut into passes of B <= 11 consecutive stages.
// One workgroup stages a tile of 2^B rows x T columns (2048 elements, 64 KiB) in LDS, runs the
// B stages there with one barrier per stage, and writes the tile back: every pass reads and
// writes each element once (HBM-bound part), all butterflies run out of LDS.
//   * DIF (Gentleman-Sande) passes go from the top stage down: natural order in, bit-reversed out.
//   * DIT (Cooley-Tukey) passes go from stage 0 up: bit-reversed in, natural order out.
// Between passes the four-step twiddle W_n^((n/N') * ka * rev_B(m)) is applied on the way out
// (DIF) or in (DIT), looked up as Hi[e >> L] * Lo[e & (2^L - 1)] from two small tables, so inside
// a pass every butterfly twiddle is a plain 2^B-th root from a 2^(B-1)-entry table.
// The prover chain ifft -> coset shift -> fft is DIF(w^-1) -> DIT(w) with the coset factor inc^j / n multiplied in
// as the forward transform's first pass loads its tile: no bit-reversal permutation and no scaling pass at all.
// The natural-order API (Fr.fft / Fr.ifft) stores its last pass at the bit-reversed position (times 1/n).
// Boundary twiddles of a pass whose table fits the cache (2^21 entries) are read as ONE value per element.
#pragma once
#include "bn254_field.hip.h"
#include "device_ctx.hpp"
#include "host_field.hpp"

#include <map>
#include <vector>

namespace zkpoa {

constexpr uint32_t kNttTileLog = 11;  // 2048 elements = 64 KiB of LDS per workgroup
constexpr uint32_t kNttStridedB = 8;  // rows per tile in strided passes (x 8 columns)
constexpr uint32_t kNttMaxStridedB = 10;  // a single strided pass may take up to 10 stages (x 2 columns)
constexpr uint32_t kNttDirectMaxLog = 21;  // boundary twiddles as one table up to 2^21 entries (64 MiB, cache-resident)
// threads per 2048-element tile: 8 waves share the 64 KiB tile, so with 2 tiles per CU every SIMD has 4 waves
// to cover the multiply latency and the per-stage barriers (measured at 2^26: 256 threads +10 %, 1024 +10 %)
constexpr uint32_t kNttThreads = 512;

// T[i] = scale * base^(i * step) for i < count  (all Montgomery)
static __global__ __launch_bounds__(256) void fr_pow_table_kernel(Fr base, Fr scale, uint32_t count, void* out) {
  uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= count) return;
  Fr acc = scale, b = base;
  uint32_t e = i;
  while (e) {
    if (e & 1u) acc = acc * b;
    b = b.sqr();
    e >>= 1;
  }
  store_field(reinterpret_cast<char*>(out) + 32 * (size_t)i, acc);
}

// W^e from the two-level table: e = eh * 2^L + el
ZK_DEV Fr tw_lookup(const void* __restrict__ hi, const void* __restrict__ lo, uint32_t L, uint32_t e) {
  Fr a = load_field<Fr>(reinterpret_cast<const char*>(hi) + 32 * (size_t)(e >> L));
  Fr b = load_field<Fr>(reinterpret_cast<const char*>(lo) + 32 * (size_t)(e & ((1u << L) - 1u)));
  return a * b;
}

// The tile lives in LDS as two planes of 16-byte halves (low limbs of every element, then high limbs): consecutive
// lanes then touch consecutive 16-byte words. With the halves interleaved (32-byte stride) lanes i and i + 8 of a
// ds_read_b128 group fall on the same banks: a two-way conflict on every access of every stage.
ZK_DEV Fr lds_load(const uint4* lds, uint32_t plane, uint32_t idx) {
  uint4 a = lds[idx], b = lds[plane + idx];
  Fr r;
  r.l[0] = a.x; r.l[1] = a.y; r.l[2] = a.z; r.l[3] = a.w;
  r.l[4] = b.x; r.l[5] = b.y; r.l[6] = b.z; r.l[7] = b.w;
  return r;
}
ZK_DEV void lds_store(uint4* lds, uint32_t plane, uint32_t idx, const Fr& v) {
  lds[idx] = make_uint4(v.l[0], v.l[1], v.l[2], v.l[3]);
  lds[plane + idx] = make_uint4(v.l[4], v.l[5], v.l[6], v.l[7]);
}

// What a pass folds into its load / store besides the butterflies (each saves a whole 64 B/element round trip):
//   kPassPlain      nothing
//   kPassPreScale   DIT pass over stages [0, B): every element is multiplied on load by X^j, j = bitrev_k(p), from a
//                   two-level table (the coset shift inc^j / n between the inverse and the forward transform)
//   kPassBitrevOut  DIF pass over stages [0, B): the result is stored at bitrev_k(p), times `post` (1/n of the
//                   inverse; one) -- the natural-order output of Fr.fft / Fr.ifft without a permutation kernel
enum NttPassMode { kPassPlain = 0, kPassPreScale = 1, kPassBitrevOut = 2 };

// boundary twiddle table of a strided pass laid out like the tile rows: D[(r << s_lo) + col] = W^((col * r) << shift)
static __global__ __launch_bounds__(256) void ntt_direct_table_kernel(void* __restrict__ out, uint32_t s_lo, uint32_t B,
                                                                      uint32_t shift, const void* __restrict__ hi,
                                                                      const void* __restrict__ lo, uint32_t L) {
  uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >> (s_lo + B)) return;
  uint32_t col = i & ((1u << s_lo) - 1u), r = i >> s_lo;
  store_field(reinterpret_cast<char*>(out) + 32 * (size_t)i, tw_lookup(hi, lo, L, (col * r) << shift));
}

// One pass over stages [s_lo, s_lo + B). Tile: 2^B rows (stride 2^s_lo elements) x 2^logT columns
// (consecutive elements); requires logT <= s_lo. grid.x = n / 2^(B+logT). tw_direct (optional): the boundary
// twiddles of this pass as one table (a single multiplication per element instead of hi * lo and then the product).
// blockIdx.y = which of `batch` equally sized vectors, `batch_stride` bytes apart (the prover transforms A, B and C
// of a proof together: a third of the launches, three times the workgroups per launch, one set of twiddle tables).
// R4: two butterfly stages per barrier -- every thread takes the four rows of a radix-4 group through both stages in
// registers (half the LDS round trips and barriers; the multiplication count is unchanged: in a prime field the
// "free" rotation by i of a complex radix-4 butterfly is an ordinary product).
template <bool DIF, int MODE, bool R4>
static __global__ __launch_bounds__(kNttThreads) void ntt_pass_kernel(const void* src, void* dst, size_t batch_stride,
                                                              uint32_t k, uint32_t s_lo,
                                                              uint32_t B, uint32_t logT,
                                                              const void* __restrict__ small_tw,
                                                              const void* __restrict__ tw_hi,
                                                              const void* __restrict__ tw_lo, uint32_t L,
                                                              const void* __restrict__ tw_direct,
                                                              const void* __restrict__ sc_hi,
                                                              const void* __restrict__ sc_lo, uint32_t Lc, Fr post) {
  extern __shared__ __attribute__((aligned(16))) uint4 lds[];
  const uint32_t T = 1u << logT, tile = 1u << (B + logT), tid = threadIdx.x, nthr = blockDim.x;
  const uint32_t cg_count = (1u << s_lo) >> logT;
  const uint32_t cg = blockIdx.x % cg_count;
  const uint64_t U = blockIdx.x / cg_count;
  const uint64_t base = (U << (s_lo + B)) + (uint64_t)cg * T;
  const uint32_t shift = k - (s_lo + B);  // log2(n / N')
  const char* sp = reinterpret_cast<const char*>(src) + (size_t)blockIdx.y * batch_stride;
  char* d = reinterpret_cast<char*>(dst) + (size_t)blockIdx.y * batch_stride;

  for (uint32_t e = tid; e < tile; e += nthr) {
    uint32_t c = e & (T - 1u), m = e >> logT;
    uint64_t p = base + ((uint64_t)m << s_lo) + c;
    Fr v = load_field<Fr>(sp + 32 * p);
    if (!DIF && s_lo > 0) {
      uint32_t r = __brev(m) >> (32u - B);
      if (tw_direct) v = v * load_field<Fr>(reinterpret_cast<const char*>(tw_direct) + 32 * (((size_t)r << s_lo) + cg * T + c));
      else v = v * tw_lookup(tw_hi, tw_lo, L, ((cg * T + c) * r) << shift);
    }
    if (MODE == kPassPreScale) v = v * tw_lookup(sc_hi, sc_lo, Lc, __brev((uint32_t)p) >> (32u - k));
    lds_store(lds, tile, e, v);
  }
  __syncthreads();

  const char* stw = reinterpret_cast<const char*>(small_tw);
  uint32_t it = 0;
  while (it < B) {
    if (R4 && B - it >= 2u) {
      // stages (sl, sl + 1), sl = the lower one: rows m0 + {0, h, 2h, 3h}, h = 2^sl, j = m0 mod h.
      // twiddles: t1 = W_{2h}^j (stage sl, both pairs), t2 = W_{4h}^j and t3 = W_{4h}^(j + h) (stage sl + 1)
      const uint32_t sl = DIF ? (B - 2u - it) : it;
      const uint32_t h = 1u << sl;
      for (uint32_t q = tid; q < (tile >> 2); q += nthr) {
        const uint32_t c = q & (T - 1u), mq = q >> logT;
        const uint32_t j = mq & (h - 1u);
        const uint32_t m0 = ((mq >> sl) << (sl + 2u)) | j;
        const uint32_t i0 = (m0 << logT) + c, st = h << logT;
        Fr x0 = lds_load(lds, tile, i0), x1 = lds_load(lds, tile, i0 + st), x2 = lds_load(lds, tile, i0 + 2u * st),
           x3 = lds_load(lds, tile, i0 + 3u * st);
        if (DIF) {
          Fr a0 = x0 + x2, a2 = x0 - x2, a1 = x1 + x3, a3 = x1 - x3;
          a3 = a3 * load_field<Fr>(stw + 32 * (size_t)((j + h) << (B - 2u - sl)));
          if (sl) {   // sl == 0: j = 0, so t2 = t1 = 1 (uniform branch)
            a2 = a2 * load_field<Fr>(stw + 32 * (size_t)(j << (B - 2u - sl)));
            const Fr t1 = load_field<Fr>(stw + 32 * (size_t)(j << (B - 1u - sl)));
            lds_store(lds, tile, i0, a0 + a1);
            lds_store(lds, tile, i0 + st, (a0 - a1) * t1);
            lds_store(lds, tile, i0 + 2u * st, a2 + a3);
            lds_store(lds, tile, i0 + 3u * st, (a2 - a3) * t1);
          } else {
            lds_store(lds, tile, i0, a0 + a1);
            lds_store(lds, tile, i0 + st, a0 - a1);
            lds_store(lds, tile, i0 + 2u * st, a2 + a3);
            lds_store(lds, tile, i0 + 3u * st, a2 - a3);
          }
        } else {
          if (sl) {
            const Fr t1 = load_field<Fr>(stw + 32 * (size_t)(j << (B - 1u - sl)));
            x1 = x1 * t1;
            x3 = x3 * t1;
          }
          Fr a0 = x0 + x1, a1 = x0 - x1, a2 = x2 + x3, a3 = x2 - x3;
          if (sl) a2 = a2 * load_field<Fr>(stw + 32 * (size_t)(j << (B - 2u - sl)));
          a3 = a3 * load_field<Fr>(stw + 32 * (size_t)((j + h) << (B - 2u - sl)));
          lds_store(lds, tile, i0, a0 + a2);
          lds_store(lds, tile, i0 + st, a1 + a3);
          lds_store(lds, tile, i0 + 2u * st, a0 - a2);
          lds_store(lds, tile, i0 + 3u * st, a1 - a3);
        }
      }
      it += 2u;
    } else {
      const uint32_t sl = DIF ? (B - 1u - it) : it;
      const uint32_t half = 1u << sl;
      for (uint32_t b = tid; b < (tile >> 1); b += nthr) {
        uint32_t c = b & (T - 1u), mb = b >> logT;
        uint32_t j = mb & (half - 1u);
        uint32_t m0 = ((mb >> sl) << (sl + 1u)) | j;
        uint32_t i0 = (m0 << logT) + c, i1 = i0 + (half << logT);
        Fr u = lds_load(lds, tile, i0), v = lds_load(lds, tile, i1);
        if (sl == 0u) {  // tile-local stage 0: every twiddle is W^0 = 1, no multiplication (uniform branch)
          lds_store(lds, tile, i0, u + v);
          lds_store(lds, tile, i1, u - v);
        } else {
          Fr tw = load_field<Fr>(stw + 32 * (size_t)(j << (B - 1u - sl)));
          if (DIF) {
            lds_store(lds, tile, i0, u + v);
            lds_store(lds, tile, i1, (u - v) * tw);
          } else {
            v = v * tw;
            lds_store(lds, tile, i0, u + v);
            lds_store(lds, tile, i1, u - v);
          }
        }
      }
      it += 1u;
    }
    __syncthreads();
  }

  for (uint32_t e = tid; e < tile; e += nthr) {
    uint32_t c = e & (T - 1u), m = e >> logT;
    uint64_t p = base + ((uint64_t)m << s_lo) + c;
    Fr v = lds_load(lds, tile, e);
    if (DIF && s_lo > 0) {
      uint32_t r = __brev(m) >> (32u - B);
      if (tw_direct) v = v * load_field<Fr>(reinterpret_cast<const char*>(tw_direct) + 32 * (((size_t)r << s_lo) + cg * T + c));
      else v = v * tw_lookup(tw_hi, tw_lo, L, ((cg * T + c) * r) << shift);
    }
    if (MODE == kPassBitrevOut) {
      // the images of different tiles interleave: with more than one tile, dst must not be the buffer being read
      v = v * post;
      store_field(d + 32 * (size_t)(__brev((uint32_t)p) >> (32u - k)), v);
    } else {
      store_field(d + 32 * p, v);
    }
  }
}

// ---- H-scalar chain split over G ranks (NTT row) -----------------------------------------
// n = G * M. Rank g holds X[g + G t] (cyclic rows), runs a size-M DIF with root w^-G (slot p = bitrev(k1)),
// and the ranks exchange so that rank h owns slots [h*Q, (h+1)*Q), Q = M / G, of every rank's transform.
// This kernel is everything between the two exchanges, per slot p (k1 = bitrev_M(p)):
//   v[g]  = in[g][p] * w^-(g k1)                       four-step twiddle of the inverse transform
//   a[k2] = sum_g v[g] * (w^-M)^(g k2)                 size-G DFT: coefficient k = k1 + M k2 (unscaled)
//   a[k2] *= inc^k / n                                  batchApplyKey(1, inc) and the 1/n of the ifft
//   b[i2] = sum_k2 a[k2] * (w^M)^(k2 i2)               size-G DFT of the forward transform
//   out[i2][p] = b[i2] * w^(k1 i2)                      its four-step twiddle
// after the second exchange rank i2 holds slot-ordered input of a size-M DIT whose output t is the
// evaluation at odd-coset index t*G + i2. G <= 8, so the two size-G DFTs are direct sums.
struct SplitRoots {
  Fr inv[8];   // (w^-M)^e, e < G
  Fr fwd[8];   // (w^M)^e
};

// in / out: rank g's (resp. i2's) Q slots start at element g * rank_stride (exchange buffers are laid out
// [rank][polynomial][Q], so rank_stride = 3 Q and the caller offsets the pointers by polynomial * Q).
template <uint32_t G>
static __global__ __launch_bounds__(256) void ntt_split_mid_kernel(const void* __restrict__ in, void* __restrict__ out,
                                                                   uint32_t Q, uint32_t rank_stride, uint32_t slot0,
                                                                   uint32_t logM,
                                                                   SplitRoots roots, const void* __restrict__ inv_hi,
                                                                   const void* __restrict__ inv_lo,
                                                                   const void* __restrict__ fwd_hi,
                                                                   const void* __restrict__ fwd_lo, uint32_t L,
                                                                   const void* __restrict__ cos_hi,
                                                                   const void* __restrict__ cos_lo, uint32_t Lc) {
  uint32_t pl = blockIdx.x * 256u + threadIdx.x;
  if (pl >= Q) return;
  const uint32_t p = slot0 + pl;
  const uint32_t k1 = logM ? (__brev(p) >> (32u - logM)) : 0u;
  const char* src = reinterpret_cast<const char*>(in);
  char* dst = reinterpret_cast<char*>(out);
  Fr v[G], a[G];
#pragma unroll
  for (uint32_t g = 0; g < G; g++) v[g] = load_field<Fr>(src + 32 * ((size_t)g * rank_stride + pl));
  {
    Fr t = tw_lookup(inv_hi, inv_lo, L, k1), pw = t;
#pragma unroll
    for (uint32_t g = 1; g < G; g++) {
      v[g] = v[g] * pw;
      if (g + 1 < G) pw = pw * t;
    }
  }
#pragma unroll
  for (uint32_t k2 = 0; k2 < G; k2++) {
    Fr acc = v[0];
#pragma unroll
    for (uint32_t g = 1; g < G; g++) acc = acc + v[g] * roots.inv[(g * k2) & (G - 1u)];
    a[k2] = acc * tw_lookup(cos_hi, cos_lo, Lc, k1 + (k2 << logM));
  }
  Fr t = tw_lookup(fwd_hi, fwd_lo, L, k1), pw = Fr::one();
#pragma unroll
  for (uint32_t i2 = 0; i2 < G; i2++) {
    Fr acc = a[0];
#pragma unroll
    for (uint32_t k2 = 1; k2 < G; k2++) acc = acc + a[k2] * roots.fwd[(k2 * i2) & (G - 1u)];
    if (i2) acc = acc * pw;
    store_field(dst + 32 * ((size_t)i2 * rank_stride + pl), acc);
    pw = pw * t;
  }
}

// ---- host side ---------------------------------------------------------------------------------
struct NttPassDesc {
  uint32_t s_lo, B, logT;
};

// passes in DIT order (stage 0 upwards); DIF runs them in reverse
// Tile size of a transform: 2048 elements (64 KiB: two workgroups of 512 threads per CU) up to 2^22, where it makes
// the transform two passes; above that 1024 elements (32 KiB: FOUR workgroups of 256 threads per CU). The pass count
// is the same there (2^26 = 10 + 8 + 8 stages instead of 11 + 8 + 7), and a workgroup's load -> butterflies -> store
// phases do not overlap with each other, only with the other workgroups of the CU: four of them out of step keep the
// multipliers busy while one loads or stores. ZKPOA_NTT_TILE=10|11 forces either (measurement).
inline uint32_t ntt_tile_log(uint32_t k) {
  static const int forced = [] {
    const char* e = getenv("ZKPOA_NTT_TILE");
    return e ? atoi(e) : 0;
  }();
  if (forced == 10 || forced == 11) return (uint32_t)forced;
  return k >= 23 ? 10u : 11u;
}

inline std::vector<NttPassDesc> ntt_plan(uint32_t k) {
  std::vector<NttPassDesc> v;
  if (k == 0) return v;
  const uint32_t kNttTileLog = ntt_tile_log(k);   // shadows the constant: everything below is per transform size
  const uint32_t kNttMaxStridedB = kNttTileLog - 1;
  uint32_t b0 = k < kNttTileLog ? k : kNttTileLog;
  v.push_back({0, b0, 0});
  uint32_t rest = k - b0;
  if (rest && rest <= kNttMaxStridedB) {
    // one strided pass of `rest` stages: tile = 2^rest rows x 2^(11 - rest) columns (>= 64-B row segments).
    // Fewer columns coalesce less well, but a whole pass (64 B/element of traffic + one boundary twiddle,
    // 2 modmuls/element) disappears; these sizes (n <= 2^21) also sit in the Infinity Cache.
    v.push_back({b0, rest, kNttTileLog - rest});
  } else if (rest) {
    uint32_t npass = (rest + kNttStridedB - 1) / kNttStridedB;
    uint32_t s = b0;
    for (uint32_t i = 0; i < npass; i++) {
      uint32_t b = rest / npass + (i < rest % npass ? 1 : 0);
      // always a full 2048-element tile: fewer rows = more columns (r02 kept 8 columns, so a 5- or 6-stage pass ran
      // on 256- or 512-element tiles with most of the workgroup idle: 2^22 and 2^24 were off the curve)
      v.push_back({s, b, kNttTileLog - b});
      s += b;
    }
  }
  return v;
}

inline HFr hfr_root_of_unity(uint32_t k) {  // w[k], Montgomery
  // w[28] = 5^((r-1)/2^28)
  static const uint64_t e28[4] = {0x9b9709143e1f593full, 0x181585d2833e8487ull, 0x131a029b85045b68ull,
                                  0x000000030644e72eull};
  HFr w = HFr::from_u64(5).pow(e28);
  for (uint32_t i = 28; i > k; i--) w = w.sqr();
  return w;
}

struct NttTables {  // device tables for one (k, direction)
  uint32_t k = 0, L = 0;
  void* hi = nullptr;                 // W^(i * 2^L), i < 2^(k-L)
  void* lo = nullptr;                 // W^i, i < 2^L
  std::map<uint32_t, void*> small;    // B -> W_{2^B}^t, t < 2^(B-1)
  std::map<uint32_t, void*> direct;   // s_lo -> boundary twiddles of the strided pass starting there (when small enough)
};

inline Fr to_dev(const HFr& h) {
  Fr f;
  memcpy(&f, &h, 32);
  return f;
}

inline void build_pow_table(hipStream_t st, const HFr& base, const HFr& scale, uint32_t count, void* out) {
  hipLaunchKernelGGL(fr_pow_table_kernel, dim3((count + 255) / 256), dim3(256), 0, st, to_dev(base), to_dev(scale),
                     count, out);
}

struct NttEngine {
  std::map<uint64_t, NttTables> cache;  // key = k*2 + inverse
  std::map<uint64_t, std::pair<void*, void*>> coset_cache;  // k -> (hi, lo) of inc^j / n

  static HFr hpow2(HFr x, uint32_t times) {
    for (uint32_t i = 0; i < times; i++) x = x.sqr();
    return x;
  }

  const NttTables& tables(hipStream_t st, uint32_t k, bool inverse) {
    uint64_t key = (uint64_t)k * 2 + (inverse ? 1 : 0);
    auto it = cache.find(key);
    if (it != cache.end()) return it->second;
    NttTables t;
    t.k = k;
    t.L = (k + 1) / 2;
    HFr w = hfr_root_of_unity(k);
    if (inverse) w = w.inv();
    uint32_t nlo = 1u << t.L, nhi = 1u << (k - t.L);
    ZK_HIP(hipMalloc(&t.hi, (size_t)nhi * 32));
    ZK_HIP(hipMalloc(&t.lo, (size_t)nlo * 32));
    build_pow_table(st, hpow2(w, t.L), HFr::one(), nhi, t.hi);
    build_pow_table(st, w, HFr::one(), nlo, t.lo);
    for (const auto& ps : ntt_plan(k)) {
      if (t.small.count(ps.B)) continue;
      void* p = nullptr;
      uint32_t cnt = ps.B ? (1u << (ps.B - 1)) : 1u;
      ZK_HIP(hipMalloc(&p, (size_t)cnt * 32));
      build_pow_table(st, hpow2(w, k - ps.B), HFr::one(), cnt, p);  // W_{2^B} = W^(2^(k-B))
      t.small[ps.B] = p;
    }
    for (const auto& ps : ntt_plan(k)) {
      if (ps.s_lo == 0 || ps.s_lo + ps.B > kNttDirectMaxLog) continue;
      void* p = nullptr;
      const uint32_t cnt = 1u << (ps.s_lo + ps.B);
      ZK_HIP(hipMalloc(&p, (size_t)cnt * 32));
      hipLaunchKernelGGL(ntt_direct_table_kernel, dim3((cnt + 255) / 256), dim3(256), 0, st, p, ps.s_lo, ps.B,
                         k - (ps.s_lo + ps.B), (const void*)t.hi, (const void*)t.lo, t.L);
      t.direct[ps.s_lo] = p;
    }
    return cache.emplace(key, t).first->second;
  }

  // tables for p -> (inc^j) * scale with j < 2^k
  std::pair<void*, void*> pow_tables(hipStream_t st, uint32_t k, const HFr& g, const HFr& scale, uint64_t cache_key) {
    auto it = coset_cache.find(cache_key);
    if (it != coset_cache.end()) return it->second;
    uint32_t L = (k + 1) / 2;
    void *hi = nullptr, *lo = nullptr;
    ZK_HIP(hipMalloc(&hi, (size_t)(1u << (k - L)) * 32));
    ZK_HIP(hipMalloc(&lo, (size_t)(1u << L) * 32));
    build_pow_table(st, hpow2(g, L), HFr::one(), 1u << (k - L), hi);
    build_pow_table(st, g, scale, 1u << L, lo);
    return coset_cache.emplace(cache_key, std::make_pair(hi, lo)).first->second;
  }

  // mode applies to the pass over stages [0, B) (the last DIF pass / the first DIT pass). kPassPreScale: (sc_hi,
  // sc_lo, Lc) two-level table of the per-element factor. kPassBitrevOut: the transform reads `d_data`, uses `tmp`
  // (n elements) for the intermediate passes and leaves the NATURAL-order result, times `post`, in d_data.
  // batch vectors of 2^k elements, batch_stride bytes apart, go through every pass together (grid.y = batch).
  static bool radix4() {   // ZKPOA_NTT_RADIX=2: one stage per barrier, as before r03 (A/B measurement)
    static const bool v = [] {
      const char* e = getenv("ZKPOA_NTT_RADIX");
      return !(e && !strcmp(e, "2"));
    }();
    return v;
  }
  template <bool DIF>
  void run_passes(hipStream_t st, void* d_data, uint32_t k, bool inverse, int mode = kPassPlain,
                  const void* sc_hi = nullptr, const void* sc_lo = nullptr, uint32_t Lc = 0, void* tmp = nullptr,
                  const HFr* post = nullptr, uint32_t batch = 1, size_t batch_stride = 0) {
    if (k == 0 || batch == 0) return;
    if (mode == kPassBitrevOut && batch != 1) throw HipError("ntt: the natural-order form takes one vector at a time");
    const NttTables& t = tables(st, k, inverse);
    auto plan = ntt_plan(k);
    const Fr post_d = to_dev(post ? *post : HFr::one());
    for (size_t idx = 0; idx < plan.size(); idx++) {
      const NttPassDesc& ps = DIF ? plan[plan.size() - 1 - idx] : plan[idx];
      uint32_t tile_log = ps.B + ps.logT;
      uint32_t grid = 1u << (k - tile_log);
      // a quarter of the tile: one radix-4 group (or two radix-2 butterflies) per thread and stage pair
      const uint32_t nthreads = tile_log >= 8 ? (tile_log >= 11 ? kNttThreads : (1u << (tile_log - 2))) : 64u;
      size_t lds_bytes = (size_t)32 << tile_log;
      auto dit = t.direct.find(ps.s_lo);
      const void* direct = dit == t.direct.end() ? nullptr : dit->second;
      const bool special = ps.s_lo == 0 && mode != kPassPlain;
      const void* src = d_data;
      void* dst = d_data;
      if (mode == kPassBitrevOut && plan.size() > 1) {   // d -> tmp, tmp -> tmp ..., tmp -> d (bit-reversed positions)
        src = idx == 0 ? d_data : tmp;
        dst = idx + 1 == plan.size() ? d_data : tmp;
      }
#define ZK_NTT_PASS_R(MODE_, R4_)                                                                                        \
  hipLaunchKernelGGL((ntt_pass_kernel<DIF, MODE_, R4_>), dim3(grid, batch), dim3(nthreads), lds_bytes, st, src, dst,      \
                     batch_stride, k, ps.s_lo, ps.B, ps.logT, (const void*)t.small.at(ps.B), (const void*)t.hi,          \
                     (const void*)t.lo, t.L, direct, sc_hi, sc_lo, Lc, post_d)
#define ZK_NTT_PASS(MODE_)              \
  do {                                  \
    if (radix4()) ZK_NTT_PASS_R(MODE_, true); \
    else ZK_NTT_PASS_R(MODE_, false);   \
  } while (0)
      if (special && mode == kPassPreScale) ZK_NTT_PASS(kPassPreScale);
      else if (special && mode == kPassBitrevOut) ZK_NTT_PASS(kPassBitrevOut);
      else ZK_NTT_PASS(kPassPlain);
#undef ZK_NTT_PASS
#undef ZK_NTT_PASS_R
    }
  }

  // natural -> bit-reversed, root w (or w^-1)
  void dif(hipStream_t st, void* d, uint32_t k, bool inverse, uint32_t batch = 1, size_t stride = 0) {
    run_passes<true>(st, d, k, inverse, kPassPlain, nullptr, nullptr, 0, nullptr, nullptr, batch, stride);
  }
  // bit-reversed -> natural
  void dit(hipStream_t st, void* d, uint32_t k, bool inverse, uint32_t batch = 1, size_t stride = 0) {
    run_passes<false>(st, d, k, inverse, kPassPlain, nullptr, nullptr, 0, nullptr, nullptr, batch, stride);
  }

  // Fr.fft / Fr.ifft semantics: natural order in and out. The bit-reversal and the 1/n of the inverse ride on the
  // last DIF pass's store (no permutation kernel, no scaling kernel); multi-pass sizes go through a scratch buffer
  // of n elements (grow-only, kept on the engine).
  void* nat_tmp = nullptr;
  size_t nat_tmp_bytes = 0;
  void transform_natural(hipStream_t st, void* d, uint32_t k, bool inverse) {
    if (k == 0) return;
    const uint64_t n = 1ull << k;
    if (ntt_plan(k).size() > 1 && nat_tmp_bytes < n * 32) {
      ZK_HIP(hipStreamSynchronize(st));
      if (nat_tmp) ZK_HIP(hipFree(nat_tmp));
      nat_tmp = nullptr;
      nat_tmp_bytes = 0;
      ZK_HIP(hipMalloc(&nat_tmp, n * 32));
      nat_tmp_bytes = n * 32;
    }
    HFr ninv = HFr::from_u64(n).inv();
    run_passes<true>(st, d, k, inverse, kPassBitrevOut, nullptr, nullptr, 0, nat_tmp, inverse ? &ninv : nullptr);
  }

  // evaluations on the domain -> evaluations on the odd coset (ifft, batchApplyKey(1, inc), fft): the coset shift
  // inc^j / n is applied by the forward transform's first pass as it loads (no separate scaling pass)
  void to_odd_coset(hipStream_t st, void* d, uint32_t k, uint32_t batch = 1, size_t stride = 0) {
    if (k == 0) return;  // n = 1: constant polynomial
    uint64_t n = 1ull << k;
    HFr inc = (k == 28) ? HFr::from_u64(25) : hfr_root_of_unity(k + 1);
    HFr ninv = HFr::from_u64(n).inv();
    auto tb = pow_tables(st, k, inc, ninv, k);
    dif(st, d, k, true, batch, stride);
    run_passes<false>(st, d, k, false, kPassPreScale, tb.first, tb.second, (k + 1) / 2, nullptr, nullptr, batch, stride);
  }

  // the part of to_odd_coset between the two exchanges when the transform is split over G ranks
  // (ntt_split_mid_kernel); k = log2 n of the whole domain, h = this rank, in/out: G blocks of M / G elements,
  // `rank_stride` elements apart.
  void split_mid(hipStream_t st, const void* in, void* out, uint32_t k, uint32_t G, uint32_t h, uint32_t rank_stride) {
    uint32_t lg = 0;
    while ((1u << lg) < G) lg++;
    const uint32_t logM = k - lg, Q = (1u << logM) / G;
    const NttTables& ti = tables(st, k, true);
    const NttTables& tf = tables(st, k, false);
    HFr inc = (k == 28) ? HFr::from_u64(25) : hfr_root_of_unity(k + 1);
    auto tc = pow_tables(st, k, inc, HFr::from_u64(1ull << k).inv(), k);
    SplitRoots roots;
    HFr wf = hpow2(hfr_root_of_unity(k), logM), wi = wf.inv(), af = HFr::one(), ai = HFr::one();
    for (uint32_t e = 0; e < 8; e++) {
      roots.fwd[e] = to_dev(af);
      roots.inv[e] = to_dev(ai);
      af = af * wf;
      ai = ai * wi;
    }
    dim3 grid((Q + 255) / 256), block(256);
    const uint32_t Lc = (k + 1) / 2;
#define ZK_SPLIT_MID(GG)                                                                                            \
  hipLaunchKernelGGL((ntt_split_mid_kernel<GG>), grid, block, 0, st, in, out, Q, rank_stride, h * Q, logM, roots,   \
                     (const void*)ti.hi, (const void*)ti.lo, (const void*)tf.hi, (const void*)tf.lo, ti.L,          \
                     (const void*)tc.first, (const void*)tc.second, Lc)
    if (G == 2) ZK_SPLIT_MID(2);
    else if (G == 4) ZK_SPLIT_MID(4);
    else ZK_SPLIT_MID(8);
#undef ZK_SPLIT_MID
  }

  void release() {
    for (auto& kv : cache) {
      (void)hipFree(kv.second.hi);
      (void)hipFree(kv.second.lo);
      for (auto& s : kv.second.small) (void)hipFree(s.second);
      for (auto& s : kv.second.direct) (void)hipFree(s.second);
    }
    if (nat_tmp) (void)hipFree(nat_tmp);
    nat_tmp = nullptr;
    nat_tmp_bytes = 0;
    for (auto& kv : coset_cache) {
      (void)hipFree(kv.second.first);
      (void)hipFree(kv.second.second);
    }
    cache.clear();
    coset_cache.clear();
  }
};

}  // namespace zkpoa
