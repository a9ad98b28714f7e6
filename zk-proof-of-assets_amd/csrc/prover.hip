// Groth16 prove on the device: zkey/wtns container parsing, device-resident proving key,
// buildABC -> coset NTT chain -> joinABC -> five MSMs -> randomised assembly -> JSON.
//
// This is what runs behind the reference's exec boundary scripts/g16_prove.sh:248-252
// (`prover <zkey> <wtns> <proof.json> <public.json>`); algorithm per snarkjs 0.7.2
// groth16_prove.js as restated in SURVEY.md 3.2 / 8c (the reference vendors no prover source).
#include "abc.hip.h"
#include "msm.hip.h"
#include "ntt.hip.h"
#include "zkpoa_internal.hpp"

#include <fcntl.h>
#include <stdlib.h>
#include <string.h>
#include <sys/file.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <array>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <exception>
#include <functional>
#include <future>
#include <memory>
#include <map>
#include <mutex>
#include <system_error>
#include <thread>

using namespace zkpoa;

namespace {

struct ProverError : std::runtime_error {
  int code;
  ProverError(int c, const std::string& s) : std::runtime_error(s), code(c) {}
};

// ---- per-request options ---------------------------------------------------------------------------------
// The one-shot entry points take r, s, the JSON style and verbosity from the environment (ZKPOA_R / _S / _JSON /
// _VERBOSE). A resident server answers several clients from several threads and must not mutate its process
// environment per request: zkpoa_set_thread_options gives the calling thread values that take precedence over the
// environment for its following calls (empty string = "unset for this thread even if the environment has it").
struct ReqOptions {
  bool active = false;
  std::string r, s, json, verbose;
};
ReqOptions& req_options() {
  static thread_local ReqOptions o;
  return o;
}
const char* req_getenv(const char* name) {
  const ReqOptions& o = req_options();
  if (o.active) {
    const std::string* v = !strcmp(name, "ZKPOA_R") ? &o.r : !strcmp(name, "ZKPOA_S") ? &o.s :
                           !strcmp(name, "ZKPOA_JSON") ? &o.json : !strcmp(name, "ZKPOA_VERBOSE") ? &o.verbose : nullptr;
    if (v) return v->empty() ? nullptr : v->c_str();
  }
  return getenv(name);
}

// ---- binfile container: magic[4] u32 version u32 nSections {u32 id u64 len payload}*
struct Section {
  const uint8_t* p = nullptr;
  uint64_t len = 0;
};
typedef std::map<uint32_t, Section> Sections;

uint32_t rd_u32(const uint8_t* p) {
  uint32_t v;
  memcpy(&v, p, 4);
  return v;
}
uint64_t rd_u64(const uint8_t* p) {
  uint64_t v;
  memcpy(&v, p, 8);
  return v;
}

Sections parse_binfile(const uint8_t* buf, uint64_t size, const char* magic, uint32_t max_version) {
  if (size < 12 || memcmp(buf, magic, 4) != 0)
    throw ProverError(PROVER_ERROR, std::string(magic) + " file: invalid file format (bad magic)");
  uint32_t version = rd_u32(buf + 4), nsec = rd_u32(buf + 8);
  if (version > max_version) throw ProverError(PROVER_ERROR, std::string(magic) + " file: version not supported");
  Sections out;
  uint64_t pos = 12;
  for (uint32_t i = 0; i < nsec; i++) {
    if (pos + 12 > size) throw ProverError(PROVER_ERROR, std::string(magic) + " file: truncated section table");
    uint32_t id = rd_u32(buf + pos);
    uint64_t len = rd_u64(buf + pos + 4);
    pos += 12;
    if (len > size - pos) throw ProverError(PROVER_ERROR, std::string(magic) + " file: truncated section");
    if (!out.count(id)) out[id] = Section{buf + pos, len};
    pos += len;
  }
  return out;
}

const Section& need(const Sections& s, uint32_t id, const char* what) {
  auto it = s.find(id);
  if (it == s.end()) throw ProverError(PROVER_ERROR, std::string("missing section ") + what);
  return it->second;
}

const uint8_t kQ[32] = {0x47, 0xfd, 0x7c, 0xd8, 0x16, 0x8c, 0x20, 0x3c, 0x8d, 0xca, 0x71, 0x68, 0x91, 0x6a, 0x81, 0x97,
                        0x5d, 0x58, 0x81, 0x81, 0xb6, 0x45, 0x50, 0xb8, 0x29, 0xa0, 0x31, 0xe1, 0x72, 0x4e, 0x64, 0x30};
const uint8_t kR[32] = {0x01, 0x00, 0x00, 0xf0, 0x93, 0xf5, 0xe1, 0x43, 0x91, 0x70, 0xb9, 0x79, 0x48, 0xe8, 0x33, 0x28,
                        0x5d, 0x58, 0x81, 0x81, 0xb6, 0x45, 0x50, 0xb8, 0x29, 0xa0, 0x31, 0xe1, 0x72, 0x4e, 0x64, 0x30};

void* dev_upload(zkpoa_context* ctx, const void* src, size_t bytes) {
  void* d = nullptr;
  auto t0 = std::chrono::steady_clock::now();
  ZK_HIP(hipMalloc(&d, bytes ? bytes : 1));
  auto t1 = std::chrono::steady_clock::now();
  try {
    if (bytes) ctx->uploader.upload(d, src, bytes, ctx->dev.device, ctx->dev.lanes[0].stream);
    if (req_getenv("ZKPOA_VERBOSE") && bytes > (16u << 20))
      fprintf(stderr, "zkpoa:   upload %.0f MB: hipMalloc %.1f ms, copy %.1f ms\n", bytes / 1e6,
              std::chrono::duration<double, std::milli>(t1 - t0).count(),
              std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t1).count());
  } catch (...) {
    (void)hipFree(d);
    throw;
  }
  return d;
}

}  // namespace

// ---- device-resident proving key -------------------------------------------------------------------
struct zkpoa_zkey {
  uint32_t nVars = 0, nPublic = 0, domain = 0, power = 0;
  uint64_t nCoefs = 0;
  Affine<HFq> alpha1, beta1, delta1;
  Affine<HFq2> beta2, delta2;
  // Verification key carried by the zkey itself (section 2: alpha1, beta2, gamma2, delta2; section 3: IC), wire
  // format alpha1(64) beta2(128) gamma2(128) delta2(128) IC[(nPublic+1) x 64]. Empty for keys assembled from
  // device buffers (zkpoa_zkey_load_device has no gamma2 / IC). Used by the self-check of the first proof(s).
  std::vector<uint8_t> vkey_points;
  mutable uint64_t selfchecks_done = 0;
  void *dA = nullptr, *dB1 = nullptr, *dB2 = nullptr, *dC = nullptr, *dH = nullptr;
  uint32_t* d_row_ptr = nullptr;
  uint32_t* d_long = nullptr;   // constraints with more than kLongRow coefficients (abc.hip.h), n_long of them
  uint32_t n_long = 0;
  uint32_t* d_sig = nullptr;
  void* d_vals = nullptr;
  uint32_t* d_flag = nullptr; // [0]: a witness value >= r was seen by the last prove (range_check_kernel)
  void* d_abc = nullptr;      // 3 * domain * 32 B work area (A_T, B_T, C_T)
  mutable void* d_witness = nullptr;  // nVars * 32 B: the witness the next prove reads
  // Witness staging of the resident prover (zkey_file_prove): two buffers, so that the witness of the NEXT request is
  // uploaded while the current proof computes; d_witness points at the one being proved. wbuf[0] adopts the buffer the
  // key was loaded with, wbuf[1] is allocated when a second request first overlaps.
  mutable void* wbuf[2] = {nullptr, nullptr};
  mutable bool wbusy[2] = {false, false};
  bool owns_points = true;    // false when the point sections belong to the caller (zkpoa_zkey_load_device)
  // Shard of the MSMs this handle covers: contiguous global index ranges. The device
  // buffers dA/dB1/dB2 start at global wire index `wbase`, dC at C-section index `cbase`, dH at `hbase`.
  uint64_t wlo = 0, wcnt = 0, wbase = 0;   // A, B1, B2: wire indices [wlo, wlo + wcnt)
  uint64_t clo = 0, ccnt = 0, cbase = 0;   // C: section indices (wire nPublic+1+idx)
  uint64_t hlo = 0, hcnt = 0, hbase = 0;   // H: domain indices
  // Split H-scalar chain (SURVEY.md 8e rows NTT / buildABC / joinABC): split_world = G > 1 ranks each own the
  // constraint rows c = split_rank (mod G) and end up with the H scalars of the odd-coset indices
  // i = split_rank (mod G), so the H points are sharded cyclically: dHs[t] = H[t*G + split_rank].
  uint32_t split_world = 0, split_rank = 0, split_log = 0;
  bool csr_local = false;        // CSR holds only this rank's rows, renumbered c >> split_log
  void* dHs = nullptr;           // cyclic H shard (owned), domain / split_world points
  mutable bool h_ready = false;  // d_abc[0 .. domain/split_world) holds this proof's H scalars (stage 3 done)
  uint64_t nCoefsLocal = 0;
  // Block-cyclic shard of sections 5-8 (bc_log > 0; shard handles only): this rank holds the blocks b = bc_rank
  // (mod bc_world) of 2^bc_log consecutive items -- wires for A / B1 / B2, section indices for C -- concatenated. A
  // contiguous range inherits whatever clustering a real witness has (bit-decomposition wires come in runs of cheap
  // 0 / 1 scalars, limbs in runs of full-width ones), so ranks would finish at different times; blocks of 2^16 wires
  // dealt round-robin even that out while every block is still one contiguous byte range of the file. Local index j
  // of a section <-> global index (((j >> L) * world + rank) << L) + (j & (2^L - 1)).
  uint32_t bc_log = 0, bc_rank = 0, bc_world = 1;
  void* d_cscal = nullptr;       // block-cyclic handles: the C query's witness values, gathered per proof
  static uint64_t bc_count(uint64_t n, uint32_t L, uint64_t rank, uint64_t world) {
    const uint64_t B = 1ull << L, nb = (n + B - 1) >> L;
    if (nb <= rank) return 0;
    uint64_t cnt = ((nb - rank + world - 1) / world) << L;
    if ((nb - 1) % world == rank && (n & (B - 1))) cnt -= B - (n & (B - 1));
    return cnt;
  }
  // A and B queries without their points at infinity (abc.hip.h): compacted copies of the resident range of
  // section 5 resp. 6 / 7, the wire of every kept point, pos[i] = kept points before resident wire i (a shard's
  // slice of the compacted arrays is [pos[wlo - wbase], pos[wlo + wcnt - wbase])), and the gathered scalars.
  struct CompactQuery {
    void *g1 = nullptr, *g2 = nullptr, *scalars = nullptr;
    uint32_t *wire = nullptr, *pos = nullptr;
    uint64_t res = 0;          // kept points in the resident range
    uint64_t lo = 0, cnt = 0;  // this shard's slice of them
    void release() {
      void* ptrs[] = {g1, g2, scalars, wire, pos};
      for (void* p : ptrs)
        if (p) (void)hipFree(p);
      g1 = g2 = scalars = nullptr;
      wire = pos = nullptr;
    }
  };
  CompactQuery qA, qB;
  // Fixed-base tables (msm.hip.h MsmTable; zkpoa_zkey_precompute): 2^(c*j) * P for every window j of a whole
  // resident base array -- the compacted A / B queries, section 8, section 9. An MSM uses its table only when it
  // covers exactly that array (a re-pointed shard of a resident key falls back to the classic form).
  MsmTable *tA = nullptr, *tB1 = nullptr, *tB2 = nullptr, *tC = nullptr, *tH = nullptr;
  bool tH_cyclic = false;     // tH was built from the cyclic shard dHs (split handles), not from dH
  uint64_t table_bytes = 0;
  // incremental build (zkey_precompute_step: one table per call, from the resident prover's idle time)
  uint64_t table_budget = 0;  // fixed at the first step (half of the HBM free then)
  bool tables_settled = false;   // every table that is wanted and fits has been built (or an attempt failed)
  bool warmed = false;           // a throw-away proof has run since the tables settled (the lanes' workspaces regrown)
  // digit density of the last witness measured on this handle (msm_density: non-zero digits per scalar for every
  // window width). A circuit's witnesses all look alike (bits stay bits), so it is measured by the first proof only
  // and sizes the windows of later proofs and of the witness tables (zkey_precompute after a proof).
  mutable double witness_density[32];
  mutable bool have_density = false;
  std::atomic<uint64_t> proofs_done{0};   // groth16_prover_zkey_file's cache precomputes when a key is used a second time
  void release_tables() {
    for (MsmTable** t : {&tA, &tB1, &tB2, &tC, &tH}) {
      msm_table_release(*t);
      *t = nullptr;
    }
    table_bytes = 0;
    tH_cyclic = false;
    table_budget = 0;
    tables_settled = false;
    warmed = false;
  }
  void set_full() {
    wlo = 0; wcnt = nVars; clo = 0; ccnt = (uint64_t)nVars - nPublic - 1; hlo = 0; hcnt = domain;
  }
  static void split(uint64_t n, uint64_t rank, uint64_t world, uint64_t& lo, uint64_t& cnt) {
    uint64_t base = n / world, rem = n % world;
    lo = rank * base + (rank < rem ? rank : rem);
    cnt = base + (rank < rem ? 1 : 0);
  }
  void set_shard(uint64_t rank, uint64_t world) {
    split(nVars, rank, world, wlo, wcnt);
    split((uint64_t)nVars - nPublic - 1, rank, world, clo, ccnt);
    split(domain, rank, world, hlo, hcnt);
  }
  void release() {
    release_tables();
    void* pts[] = {dA, dB1, dB2, dC, dH};
    if (owns_points)
      for (void* p : pts)
        if (p) (void)hipFree(p);
    if (wbuf[0] || wbuf[1]) {   // the staging pair owns the witness memory (d_witness points at one of them)
      for (void*& w : wbuf) {
        if (w) (void)hipFree(w);
        w = nullptr;
      }
      d_witness = nullptr;
    }
    void* ptrs[] = {d_row_ptr, d_sig, d_vals, d_abc, d_witness, dHs, d_long, d_flag, d_cscal};
    for (void* p : ptrs)
      if (p) (void)hipFree(p);
    qA.release();
    qB.release();
    dA = dB1 = dB2 = dC = dH = d_vals = d_abc = d_witness = dHs = d_cscal = nullptr;
    d_long = nullptr;
    d_flag = nullptr;
    d_row_ptr = d_sig = nullptr;
  }
};

namespace {

// count x 32-byte elements at d must all be below the modulus (Fq: point coordinates, Fr: witness values)
template <class PRM>
void range_check(hipStream_t st, const void* d, uint64_t count, uint32_t* d_flag) {
  if (count)
    hipLaunchKernelGGL((range_check_kernel<PRM>), dim3((uint32_t)((count + 255) / 256)), dim3(256), 0, st, d, count, d_flag);
}

// list of the constraints too long for one lane (key-dependent only); returns the count, *list owned by the caller
uint32_t build_long_list(hipStream_t st, const uint32_t* d_row_ptr, uint32_t rows, uint32_t** list) {
  DevBuf cnt(64);
  *list = nullptr;
  uint32_t n_long = 0;
  const uint32_t grid = (rows / 2 + 255) / 256;
  ZK_HIP(hipMemsetAsync(cnt.p, 0, 64, st));
  hipLaunchKernelGGL(abc_long_list_kernel, dim3(grid), dim3(256), 0, st, d_row_ptr, rows, (uint32_t*)cnt.p,
                     (uint32_t*)nullptr);
  ZK_HIP(hipMemcpyAsync(&n_long, cnt.p, 4, hipMemcpyDeviceToHost, st));
  ZK_HIP(hipStreamSynchronize(st));
  ZK_HIP(hipMalloc(reinterpret_cast<void**>(list), (size_t)(n_long ? n_long : 1) * 4));
  if (n_long) {
    ZK_HIP(hipMemsetAsync(cnt.p, 0, 64, st));
    hipLaunchKernelGGL(abc_long_list_kernel, dim3(grid), dim3(256), 0, st, d_row_ptr, rows, (uint32_t*)cnt.p, *list);
    ZK_HIP(hipStreamSynchronize(st));
  }
  ZK_HIP(hipGetLastError());
  return n_long;
}

// CSR of the coefficient list by output row (2*c + m); d_recs = device copy of the 44-byte records.
// Also allocates the A/B/C work area and the witness buffer. With split_log > 0 (split chain loaded from a
// file) only the records of this rank's constraints are kept, rows renumbered c >> split_log.
void build_csr(zkpoa_context* ctx, zkpoa_zkey* zk, const void* d_recs, bool local_rows = false, hipStream_t st_in = nullptr) {
  hipStream_t st = st_in ? st_in : ctx->dev.lanes[0].stream;
  const uint32_t lp = local_rows ? zk->split_log : 0u, part = local_rows ? zk->split_rank : 0u;
  const uint32_t rows = 2 * (zk->domain >> lp);
  const uint64_t n = zk->domain, m = zk->nVars;
  // work area: the chain transforms A_T, B_T, C_T in place (a split shard: its n / G rows of each)
  if (!zk->d_abc) ZK_HIP(hipMalloc(&zk->d_abc, (size_t)3 * (n >> lp) * 32));
  if (!zk->d_witness) ZK_HIP(hipMalloc(&zk->d_witness, (size_t)m * 32));
  if (!zk->d_flag) ZK_HIP(hipMalloc(reinterpret_cast<void**>(&zk->d_flag), 1024));   // + scratch of msm_density at +256
  DevBuf d_cnt((size_t)rows * 4), d_rank((size_t)(zk->nCoefs ? zk->nCoefs : 1) * 4),
      d_bs(((size_t)rows / kScanTile + 2) * 4), d_misc(64);
  ZK_HIP(hipMalloc(&zk->d_row_ptr, ((size_t)rows + 1) * 4));
  ZK_HIP(hipMemsetAsync(d_cnt.p, 0, (size_t)rows * 4, st));
  ZK_HIP(hipMemsetAsync(d_misc.p, 0, 64, st));
  uint32_t* misc = (uint32_t*)d_misc.p;
  uint32_t grid = (uint32_t)((zk->nCoefs + 255) / 256);
  uint32_t herr = 0, total = 0;
  if (zk->nCoefs) {
    hipLaunchKernelGGL(abc_count_kernel, dim3(grid), dim3(256), 0, st, (const CoefRec*)d_recs, zk->nCoefs, zk->domain,
                       zk->nVars, lp, part, (uint32_t*)d_cnt.p, (uint32_t*)d_rank.p, misc + 4);
    scan_u32(st, (const uint32_t*)d_cnt.p, rows, 0, 0, zk->d_row_ptr, (uint32_t*)d_bs.p, misc, nullptr);
    ZK_HIP(hipMemcpyAsync(&total, zk->d_row_ptr + rows, 4, hipMemcpyDeviceToHost, st));
  } else {
    ZK_HIP(hipMemsetAsync(zk->d_row_ptr, 0, ((size_t)rows + 1) * 4, st));
  }
  ZK_HIP(hipMemcpyAsync(&herr, misc + 4, 4, hipMemcpyDeviceToHost, st));
  ZK_HIP(hipStreamSynchronize(st));
  ZK_HIP(hipGetLastError());
  if (herr & 2u) throw ProverError(PROVER_ERROR, "zkey coefficient value is not a field element (>= r)");
  if (herr) throw ProverError(PROVER_ERROR, "zkey coefficient record out of range (matrix/constraint/signal)");
  zk->nCoefsLocal = total;
  zk->csr_local = local_rows;
  ZK_HIP(hipMalloc(&zk->d_sig, (size_t)(total ? total : 1) * 4));
  ZK_HIP(hipMalloc(&zk->d_vals, (size_t)(total ? total : 1) * 32));
  if (zk->nCoefs) {
    hipLaunchKernelGGL(abc_scatter_kernel, dim3(grid), dim3(256), 0, st, (const CoefRec*)d_recs, zk->nCoefs, lp,
                       (const uint32_t*)zk->d_row_ptr, (const uint32_t*)d_rank.p, zk->d_sig, zk->d_vals);
    ZK_HIP(hipStreamSynchronize(st));
    ZK_HIP(hipGetLastError());
  }
  zk->n_long = build_long_list(st, zk->d_row_ptr, rows, &zk->d_long);
}

// split chain: world must be a power of two <= 8 with world^2 <= domain (every rank owns whole slots of
// every other rank's transform)
void check_split(const zkpoa_zkey* zk, uint64_t rank, uint64_t world) {
  if (world < 2 || world > 8 || (world & (world - 1)) || rank >= world)
    throw ProverError(PROVER_ERROR, "split chain: world must be 2, 4 or 8 and rank < world");
  if ((uint64_t)zk->domain < world * world)
    throw ProverError(PROVER_ERROR, "split chain: domain smaller than world^2");
}

void set_split(zkpoa_zkey* zk, uint64_t rank, uint64_t world) {
  zk->split_world = (uint32_t)world;
  zk->split_rank = (uint32_t)rank;
  zk->split_log = world == 2 ? 1 : (world == 4 ? 2 : 3);
  zk->h_ready = false;
}

// this shard's slice of a compacted query: two 4-byte reads of the position array
void query_set_slice(const zkpoa_zkey* zk, zkpoa_zkey::CompactQuery& q) {
  uint32_t a = 0, b = 0;
  ZK_HIP(hipMemcpy(&a, q.pos + (zk->wlo - zk->wbase), 4, hipMemcpyDeviceToHost));
  ZK_HIP(hipMemcpy(&b, q.pos + (zk->wlo - zk->wbase + zk->wcnt), 4, hipMemcpyDeviceToHost));
  q.lo = a;
  q.cnt = b - a;
}
void queries_set_slice(zkpoa_zkey* zk) {
  query_set_slice(zk, zk->qA);
  query_set_slice(zk, zk->qB);
}

// Compact the resident range [wbase, wbase + wres) of a query (d1: G1 section, d2: its G2 twin or null) once per key.
void query_compact(zkpoa_context* ctx, zkpoa_zkey* zk, zkpoa_zkey::CompactQuery& q, const void* d1, const void* d2,
                   uint64_t wres, hipStream_t st_in = nullptr) {
  hipStream_t st = st_in ? st_in : ctx->dev.lanes[0].stream;
  const uint32_t n = (uint32_t)wres;
  DevBuf keep(((size_t)n + 1) * 4), bs(((size_t)n / kScanTile + 2) * 4), misc(64);
  ZK_HIP(hipMalloc(reinterpret_cast<void**>(&q.pos), ((size_t)n + 1) * 4));
  ZK_HIP(hipMemsetAsync(misc.p, 0, 64, st));
  uint32_t total = 0;
  if (n) {
    hipLaunchKernelGGL(query_keep_kernel, dim3((n + 255) / 256), dim3(256), 0, st, (const uint4*)d1, (const uint4*)d2, n,
                       (uint32_t*)keep.p);
    scan_u32(st, (const uint32_t*)keep.p, n, 0, 0, q.pos, (uint32_t*)bs.p, (uint32_t*)misc.p, nullptr);
    ZK_HIP(hipMemcpyAsync(&total, q.pos + n, 4, hipMemcpyDeviceToHost, st));
  } else {
    ZK_HIP(hipMemsetAsync(q.pos, 0, 4, st));
  }
  ZK_HIP(hipStreamSynchronize(st));
  ZK_HIP(hipGetLastError());
  q.res = total;
  const size_t cnt = total ? total : 1;
  ZK_HIP(hipMalloc(&q.g1, cnt * 64));
  if (d2) ZK_HIP(hipMalloc(&q.g2, cnt * 128));
  ZK_HIP(hipMalloc(&q.scalars, cnt * 32));
  ZK_HIP(hipMalloc(reinterpret_cast<void**>(&q.wire), cnt * 4));
  if (n) {
    hipLaunchKernelGGL(query_compact_kernel, dim3((n + 255) / 256), dim3(256), 0, st, (const uint4*)d1, (const uint4*)d2,
                       (const uint32_t*)q.pos, n, (uint32_t)zk->wbase, zk->bc_log, zk->bc_rank, zk->bc_world, (uint4*)q.g1,
                       (uint4*)q.g2, q.wire);
    ZK_HIP(hipStreamSynchronize(st));
    ZK_HIP(hipGetLastError());
  }
  query_set_slice(zk, q);
}

// Both witness queries; when the handle owns the sections the originals are freed afterwards (nothing reads
// them again: the MSMs use the compacted copies).
void queries_compact(zkpoa_context* ctx, zkpoa_zkey* zk, uint64_t wres) {
  query_compact(ctx, zk, zk->qA, zk->dA, nullptr, wres);
  query_compact(ctx, zk, zk->qB, zk->dB1, zk->dB2, wres);
  if (zk->owns_points) {
    (void)hipFree(zk->dA);
    (void)hipFree(zk->dB1);
    (void)hipFree(zk->dB2);
    zk->dA = zk->dB1 = zk->dB2 = nullptr;
  }
}

// the five point sections and the coefficient section of a parsed zkey
struct ZkeySections {
  Section s4, s5, s6, s7, s8, s9;
};

// container + header validation; fills the handle's scalar fields, header points and verification key
std::unique_ptr<zkpoa_zkey> zkey_parse(const uint8_t* buf, uint64_t size, ZkeySections& out) {
  Sections secs = parse_binfile(buf, size, "zkey", 2);
  const Section& s1 = need(secs, 1, "1 (protocol)");
  if (s1.len < 4 || rd_u32(s1.p) != 1) throw ProverError(PROVER_ERROR, "zkey file is not groth16");
  const Section& s2 = need(secs, 2, "2 (groth16 header)");
  const uint64_t hdr = 4 + 32 + 4 + 32 + 12 + 64 + 64 + 128 + 128 + 64 + 128;
  if (s2.len < hdr) throw ProverError(PROVER_ERROR, "zkey header too short");
  const uint8_t* p = s2.p;
  if (rd_u32(p) != 32 || memcmp(p + 4, kQ, 32) != 0) throw ProverError(PROVER_ERROR, "zkey curve not supported (q is not BN254)");
  p += 36;
  if (rd_u32(p) != 32 || memcmp(p + 4, kR, 32) != 0) throw ProverError(PROVER_ERROR, "zkey curve not supported (r is not BN254)");
  p += 36;
  std::unique_ptr<zkpoa_zkey> zk(new zkpoa_zkey());
  zk->nVars = rd_u32(p);
  zk->nPublic = rd_u32(p + 4);
  zk->domain = rd_u32(p + 8);
  p += 12;
  if (zk->domain == 0 || (zk->domain & (zk->domain - 1))) throw ProverError(PROVER_ERROR, "zkey domainSize is not a power of two");
  zk->power = 0;
  while ((1u << zk->power) < zk->domain) zk->power++;
  if (zk->power > 28) throw ProverError(PROVER_ERROR, "zkey domainSize exceeds 2^28");
  if (zk->nPublic + 1 > zk->nVars) throw ProverError(PROVER_ERROR, "zkey nPublic >= nVars");
  zk->alpha1 = h_affine_from_bytes<HFq>(p); p += 64;
  zk->beta1 = h_affine_from_bytes<HFq>(p); p += 64;
  const uint8_t* p_alpha1 = p - 128;
  const uint8_t* p_beta2 = p;
  zk->beta2 = h_affine_from_bytes<HFq2>(p); p += 128;
  const uint8_t* p_gamma2 = p;
  p += 128;  // gamma2: verifier only (kept for the self-check)
  zk->delta1 = h_affine_from_bytes<HFq>(p); p += 64;
  const uint8_t* p_delta2 = p;
  zk->delta2 = h_affine_from_bytes<HFq2>(p);
  // section 3 (IC) is optional for proving; with it the handle can verify its own proofs
  {
    auto it3 = secs.find(3);
    if (it3 != secs.end() && it3->second.len == ((uint64_t)zk->nPublic + 1) * 64) {
      zk->vkey_points.resize(448 + it3->second.len);
      uint8_t* v = zk->vkey_points.data();
      memcpy(v, p_alpha1, 64);
      memcpy(v + 64, p_beta2, 128);
      memcpy(v + 192, p_gamma2, 128);
      memcpy(v + 320, p_delta2, 128);
      memcpy(v + 448, it3->second.p, it3->second.len);
    }
  }
  out.s4 = need(secs, 4, "4 (coefficients)");
  if (out.s4.len < 4) throw ProverError(PROVER_ERROR, "zkey coefficient section too short");
  zk->nCoefs = rd_u32(out.s4.p);
  if (out.s4.len != 4 + zk->nCoefs * 44) throw ProverError(PROVER_ERROR, "zkey coefficient section has the wrong size");
  out.s5 = need(secs, 5, "5 (A points)");
  out.s6 = need(secs, 6, "6 (B1 points)");
  out.s7 = need(secs, 7, "7 (B2 points)");
  out.s8 = need(secs, 8, "8 (C points)");
  out.s9 = need(secs, 9, "9 (H points)");
  const uint64_t m = zk->nVars, n = zk->domain;
  if (out.s5.len != m * 64 || out.s6.len != m * 64 || out.s7.len != m * 128 ||
      out.s8.len != (m - zk->nPublic - 1) * 64 || out.s9.len != n * 64)
    throw ProverError(PROVER_ERROR, "zkey point section has the wrong size");
  return zk;
}

// shard flags of the loaders (include/zkpoa_prover.h ZKPOA_SHARD_*): bit 0 = split chain, bits 8-15 = block-cyclic log
void set_block_cyclic(zkpoa_zkey* zk, uint64_t rank, uint64_t world, uint32_t bc_log) {
  if (bc_log == 0 || world <= 1) return;
  if (bc_log < 4 || bc_log > 24) throw ProverError(PROVER_ERROR, "zkey shard: block-cyclic block size must be 2^4 .. 2^24 items");
  zk->bc_log = bc_log;
  zk->bc_rank = (uint32_t)rank;
  zk->bc_world = (uint32_t)world;
  zk->wlo = zk->wbase = 0;
  zk->wcnt = zkpoa_zkey::bc_count(zk->nVars, bc_log, rank, world);
  zk->clo = zk->cbase = 0;
  zk->ccnt = zkpoa_zkey::bc_count((uint64_t)zk->nVars - zk->nPublic - 1, bc_log, rank, world);
}

zkpoa_zkey* zkey_load_impl(zkpoa_context* ctx, const uint8_t* buf, uint64_t size, uint64_t rank = 0,
                           uint64_t world = 1, bool split = false, uint32_t bc_log = 0) {
  ZkeySections zs;
  std::unique_ptr<zkpoa_zkey> zk = zkey_parse(buf, size, zs);
  const Section &s4 = zs.s4, &s5 = zs.s5, &s6 = zs.s6, &s7 = zs.s7, &s8 = zs.s8, &s9 = zs.s9;
  const uint64_t m = zk->nVars, n = zk->domain;

  if (world == 0 || rank >= world) throw ProverError(PROVER_ERROR, "zkey shard: rank/world out of range");
  zk->set_shard(rank, world);   // world == 1: the whole key
  zk->wbase = zk->wlo;
  zk->cbase = zk->clo;
  zk->hbase = zk->hlo;
  if (split) {
    check_split(zk.get(), rank, world);
    set_split(zk.get(), rank, world);
  }
  set_block_cyclic(zk.get(), rank, world, bc_log);
  const bool verbose = req_getenv("ZKPOA_VERBOSE") != nullptr;
  auto tph = std::chrono::steady_clock::now();
  auto phase = [&](const char* what) {   // ZKPOA_VERBOSE: where a key load spends its time
    if (!verbose) return;
    auto now = std::chrono::steady_clock::now();
    fprintf(stderr, "zkpoa: zkey load: %-34s %7.1f ms\n", what,
            std::chrono::duration<double, std::milli>(now - tph).count());
    tph = now;
  };
  try {
    // each rank uploads only its byte range(s) of every point section: one contiguous range, or with block-cyclic
    // shards its blocks of 2^bc_log items one after the other (each still a contiguous byte range of the file)
    auto upload_items = [&](const uint8_t* sec, uint64_t n_items, uint64_t lo, uint64_t cnt, size_t unit) -> void* {
      if (!zk->bc_log) return dev_upload(ctx, sec + lo * unit, cnt * unit);
      void* d = nullptr;
      ZK_HIP(hipMalloc(&d, cnt ? cnt * unit : 1));
      try {
        const uint64_t B = 1ull << zk->bc_log;
        for (uint64_t lb = 0;; lb++) {
          const uint64_t start = (lb * world + rank) << zk->bc_log;
          if (start >= n_items) break;
          const uint64_t len = n_items - start < B ? n_items - start : B;
          ctx->uploader.upload(reinterpret_cast<char*>(d) + (lb << zk->bc_log) * unit, sec + start * unit, len * unit,
                               ctx->dev.device, ctx->dev.lanes[0].stream);
        }
      } catch (...) {
        (void)hipFree(d);
        throw;
      }
      return d;
    };
    zk->dA = upload_items(s5.p, m, zk->wlo, zk->wcnt, 64);
    zk->dB1 = upload_items(s6.p, m, zk->wlo, zk->wcnt, 64);
    zk->dB2 = upload_items(s7.p, m, zk->wlo, zk->wcnt, 128);
    zk->dC = upload_items(s8.p, m - zk->nPublic - 1, zk->clo, zk->ccnt, 64);
    if (zk->bc_log) ZK_HIP(hipMalloc(&zk->d_cscal, zk->ccnt ? zk->ccnt * 32 : 1));
    if (split) {
      // cyclic H shard: H[t * world + rank], gathered on the host (the section is walked once per rank)
      const uint64_t cnt = n / world;
      std::vector<uint8_t> stage(cnt * 64);
      const unsigned nthreads = 6;
      std::vector<std::thread> th;
      for (unsigned k = 0; k < nthreads; k++)
        th.emplace_back([&, k] {
          for (uint64_t t = cnt * k / nthreads; t < cnt * (k + 1) / nthreads; t++)
            memcpy(stage.data() + t * 64, s9.p + (t * world + rank) * 64, 64);
        });
      for (auto& t : th) t.join();
      zk->dHs = dev_upload(ctx, stage.data(), cnt * 64);
      zk->hlo = zk->hbase = 0;
      zk->hcnt = 0;   // no contiguous H range on this handle
    } else {
      zk->dH = dev_upload(ctx, s9.p + zk->hlo * 64, zk->hcnt * 64);
    }
    phase("point sections 5-9 -> HBM");
    {   // every coordinate must be a canonical Fq element (< q)
      hipStream_t st0 = ctx->dev.lanes[0].stream;
      DevBuf flag(64);
      ZK_HIP(hipMemsetAsync(flag.p, 0, 64, st0));
      range_check<FqParams>(st0, zk->dA, zk->wcnt * 2, (uint32_t*)flag.p);
      range_check<FqParams>(st0, zk->dB1, zk->wcnt * 2, (uint32_t*)flag.p);
      range_check<FqParams>(st0, zk->dB2, zk->wcnt * 4, (uint32_t*)flag.p);
      range_check<FqParams>(st0, zk->dC, zk->ccnt * 2, (uint32_t*)flag.p);
      range_check<FqParams>(st0, split ? zk->dHs : zk->dH, (split ? n / world : zk->hcnt) * 2, (uint32_t*)flag.p);
      uint32_t bad = 0;
      ZK_HIP(hipMemcpyAsync(&bad, flag.p, 4, hipMemcpyDeviceToHost, st0));
      ZK_HIP(hipStreamSynchronize(st0));
      if (bad) throw ProverError(PROVER_ERROR, "zkey point coordinate is not a field element (>= q)");
      phase("coordinate range check");
    }
    queries_compact(ctx, zk.get(), zk->wcnt);
    phase("A / B queries without infinity");
    hipStream_t st = ctx->dev.lanes[0].stream;
    void* d_recs = dev_upload(ctx, s4.p + 4, zk->nCoefs * 44);
    phase("coefficient section -> HBM");
    try {
      build_csr(ctx, zk.get(), d_recs, split);
    } catch (...) {
      (void)hipFree(d_recs);
      throw;
    }
    (void)hipFree(d_recs);
    phase("CSR of the coefficients");
    ntt_prepare(ctx, st, zk->power);
    if (split) ntt_prepare(ctx, st, zk->power - zk->split_log);
    ZK_HIP(hipStreamSynchronize(st));
    phase("NTT tables");
  } catch (...) {
    zk->release();
    throw;
  }
  return zk.release();
}

// H-scalar chain on lane.stream: A_T,B_T,C_T -> odd coset -> P (standard form) left in abc[0 .. n)
void h_chain(zkpoa_context* ctx, hipStream_t st, const uint32_t* row_ptr, const uint32_t* sig, const void* vals,
             const uint32_t* long_list, uint32_t n_long, const void* d_witness, uint32_t domain, uint32_t power,
             void* d_abc) {
  char* A = reinterpret_cast<char*>(d_abc);
  char* B = A + (size_t)domain * 32;
  char* C = B + (size_t)domain * 32;
  uint32_t grid = (domain + 255) / 256;
  hipLaunchKernelGGL(abc_rows_kernel, dim3(grid), dim3(256), 0, st, row_ptr, sig, vals, d_witness, domain, 0u, 1u,
                     (void*)A, (void*)B, (void*)C);
  if (n_long)
    hipLaunchKernelGGL(abc_long_rows_kernel, dim3((n_long + 3) / 4), dim3(256), 0, st, row_ptr, sig, vals, d_witness,
                       long_list, n_long, 0u, 0u, (void*)A, (void*)B, (void*)C);
  ntt_to_odd_coset(ctx, st, A, power, 3, (size_t)domain * 32);   // A, B and C together: 4-6 launches instead of 12-18
  hipLaunchKernelGGL(abc_join_kernel, dim3(grid), dim3(256), 0, st, (const void*)A, (const void*)B, (const void*)C,
                     domain, (void*)A);
}

// ---- split chain: three local stages around two all-to-all exchanges the caller runs ----
// Exchange buffers (caller's device memory, e.g. torch tensors handed to RCCL): [G ranks][3 polynomials A, B, C]
// [Q = M / G elements] x 32 B, M = domain / G. Chunk h (3 Q elements, contiguous) is what rank h receives from this
// rank, so each exchange is ONE all_to_all_single with equal splits over the whole buffer. The stages compute in the
// handle's own work area ([3][M]) and pack / unpack at the boundary. All three ENQUEUE on lane 0's stream and return
// (zkpoa_context_stream): a caller that runs its collectives on that stream needs no host synchronisation between
// stage and exchange; anyone else calls zkpoa_context_synchronize first.
void split_stage1(zkpoa_context* ctx, const zkpoa_zkey* zk, void* d_x) {
  hipStream_t st = ctx->dev.lanes[0].stream;
  const uint32_t M = zk->domain >> zk->split_log, kM = zk->power - zk->split_log, Q = M >> zk->split_log;
  char* A = reinterpret_cast<char*>(zk->d_abc);
  char* B = A + (size_t)M * 32;
  char* C = B + (size_t)M * 32;
  zk->h_ready = false;
  // buildABC for the constraint rows c = rank (mod G): a rank-local CSR is already renumbered
  hipLaunchKernelGGL(abc_rows_kernel, dim3((M + 255) / 256), dim3(256), 0, st, (const uint32_t*)zk->d_row_ptr,
                     (const uint32_t*)zk->d_sig, (const void*)zk->d_vals, (const void*)zk->d_witness, M,
                     zk->csr_local ? 0u : zk->split_rank, zk->csr_local ? 1u : zk->split_world, (void*)A, (void*)B,
                     (void*)C);
  if (zk->n_long)   // a rank-local CSR lists its own rows; a full CSR lists all, the kernel keeps c = rank (mod G)
    hipLaunchKernelGGL(abc_long_rows_kernel, dim3((zk->n_long + 3) / 4), dim3(256), 0, st,
                       (const uint32_t*)zk->d_row_ptr, (const uint32_t*)zk->d_sig, (const void*)zk->d_vals,
                       (const void*)zk->d_witness, (const uint32_t*)zk->d_long, zk->n_long,
                       zk->csr_local ? 0u : zk->split_log, zk->csr_local ? 0u : zk->split_rank, (void*)A, (void*)B,
                       (void*)C);
  ntt_dif(ctx, st, A, kM, true, 3, (size_t)M * 32);
  hipLaunchKernelGGL((split_pack_kernel<true>), dim3((uint32_t)(((uint64_t)6 * M + 255) / 256)), dim3(256), 0, st,
                     (const uint4*)zk->d_abc, (uint4*)d_x, M, Q);
  ZK_HIP(hipGetLastError());
}

void split_stage2(zkpoa_context* ctx, const zkpoa_zkey* zk, const void* d_in, void* d_out) {
  hipStream_t st = ctx->dev.lanes[0].stream;
  const size_t M = zk->domain >> zk->split_log, Q = M >> zk->split_log;
  for (int x = 0; x < 3; x++)   // polynomial x of every rank's block: offset x * Q, blocks 3 Q apart
    ntt_split_mid(ctx, st, reinterpret_cast<const char*>(d_in) + x * Q * 32, reinterpret_cast<char*>(d_out) + x * Q * 32,
                  zk->power, zk->split_world, zk->split_rank, (uint32_t)(3 * Q));
  ZK_HIP(hipGetLastError());
}

void split_stage3(zkpoa_context* ctx, const zkpoa_zkey* zk, const void* d_x) {
  hipStream_t st = ctx->dev.lanes[0].stream;
  const uint32_t M = zk->domain >> zk->split_log, kM = zk->power - zk->split_log, Q = M >> zk->split_log;
  char* A = reinterpret_cast<char*>(zk->d_abc);
  char* B = A + (size_t)M * 32;
  char* C = B + (size_t)M * 32;
  hipLaunchKernelGGL((split_pack_kernel<false>), dim3((uint32_t)(((uint64_t)6 * M + 255) / 256)), dim3(256), 0, st,
                     (const uint4*)d_x, (uint4*)zk->d_abc, M, Q);
  ntt_dit(ctx, st, A, kM, false, 3, (size_t)M * 32);
  hipLaunchKernelGGL(abc_join_kernel, dim3((M + 255) / 256), dim3(256), 0, st, (const void*)A, (const void*)B,
                     (const void*)C, M, zk->d_abc);
  ZK_HIP(hipGetLastError());
  zk->h_ready = true;   // in stream order: the H MSM of zkpoa_prove_partials_device runs on the same stream
}

HFr hfr_from_le(const uint8_t* le) { return HFr::from_bytes(le); }

// uniform on [0, r): 254 random bits, rejected while >= r (about one draw in four is), as Groth16's zero-knowledge
// argument assumes and as snarkjs' Fr.random() samples
void random_scalar(uint8_t out[32]) {
  int fd = open("/dev/urandom", O_RDONLY);
  if (fd < 0) throw ProverError(PROVER_ERROR, "cannot open /dev/urandom");
  for (;;) {
    ssize_t got = read(fd, out, 32);
    if (got != 32) {
      close(fd);
      throw ProverError(PROVER_ERROR, "short read from /dev/urandom");
    }
    out[31] &= 0x3f;   // 254 bits
    bool below = false;
    for (int i = 31; i >= 0; i--) {
      if (out[i] < kR[i]) { below = true; break; }
      if (out[i] > kR[i]) break;
    }
    if (below) break;
  }
  close(fd);
}

bool parse_decimal_mod_r(const char* s, uint8_t out[32]) {
  if (!s || !*s) return false;
  HFr acc = HFr::zero(), ten = HFr::from_u64(10);
  for (const char* c = s; *c; c++) {
    if (*c < '0' || *c > '9') return false;
    acc = acc * ten + HFr::from_u64((uint64_t)(*c - '0'));
  }
  acc.from_mont().to_bytes(out);
  return true;
}

// Where a witness comes from: a complete .wtns image in memory (the rapidsnark-shaped entry points), or an open file
// (the `prover` executable, zkpoa_groth16_prover_files): then only the few header bytes are ever read through the CPU
// and the values go from the page cache straight into the uploader's pinned staging buffers (pread) -- mapping a 1.7 GB
// layer-three witness first costs ~25 ms of page-table work before the first byte moves, and as much again to unmap.
struct WtnsSrc {
  const uint8_t* buf = nullptr;
  uint64_t size = 0;
  int fd = -1;
};

struct WtnsView {
  const uint8_t* values = nullptr;   // memory form; null in file form
  uint32_t n = 0;
  int fd = -1;                       // file form: the values start at byte `off` of fd
  uint64_t off = 0;
  // values [first, first + count) -> out
  void read(uint64_t first, uint64_t count, uint8_t* out) const {
    if (fd < 0) {
      memcpy(out, values + first * 32, (size_t)count * 32);
      return;
    }
    size_t got = 0, want = (size_t)count * 32;
    while (got < want) {
      ssize_t r = pread(fd, out + got, want - got, (off_t)(off + first * 32 + got));
      if (r <= 0) throw ProverError(PROVER_ERROR, "wtns file: short read");
      got += (size_t)r;
    }
  }
  // the public signals w[1 .. n_public] (standard form), valid while `store` lives
  const uint8_t* publics(uint32_t n_public, std::vector<uint8_t>& store) const {
    if (fd < 0) return values + 32;
    store.resize((size_t)n_public * 32 + 1);
    if (n_public) read(1, n_public, store.data());
    return store.data();
  }
  // values [first, first + count) -> device memory at dst, on `st` (synchronised on return)
  void upload(zkpoa_context* ctx, void* dst, uint64_t first, uint64_t count, hipStream_t st) const {
    ctx->uploader.upload(dst, fd < 0 ? values + first * 32 : nullptr, (size_t)count * 32, ctx->dev.device, st, fd,
                         off + first * 32);
  }
};

WtnsView parse_wtns(const uint8_t* buf, uint64_t size) {
  Sections secs = parse_binfile(buf, size, "wtns", 2);
  const Section& s1 = need(secs, 1, "1 (wtns header)");
  if (s1.len < 40 || rd_u32(s1.p) != 32) throw ProverError(PROVER_ERROR, "wtns header: unsupported field size");
  if (memcmp(s1.p + 4, kR, 32) != 0)
    throw ProverError(PROVER_ERROR, "Curve of the witness does not match the curve of the proving key");
  WtnsView w;
  w.n = rd_u32(s1.p + 36);
  const Section& s2 = need(secs, 2, "2 (wtns values)");
  if (s2.len != (uint64_t)w.n * 32) throw ProverError(PROVER_ERROR, "wtns value section has the wrong size");
  w.values = s2.p;
  return w;
}

// the same container walk on an open file: 12 bytes per section header are read, payloads are skipped
WtnsView parse_wtns_fd(int fd, uint64_t size) {
  auto rd = [&](uint64_t pos, void* out, size_t len) {
    size_t got = 0;
    while (got < len) {
      ssize_t r = pread(fd, static_cast<char*>(out) + got, len - got, (off_t)(pos + got));
      if (r <= 0) throw ProverError(PROVER_ERROR, "wtns file: short read");
      got += (size_t)r;
    }
  };
  uint8_t head[12];
  if (size < 12) throw ProverError(PROVER_ERROR, "wtns file: invalid file format (bad magic)");
  rd(0, head, 12);
  if (memcmp(head, "wtns", 4) != 0) throw ProverError(PROVER_ERROR, "wtns file: invalid file format (bad magic)");
  if (rd_u32(head + 4) > 2) throw ProverError(PROVER_ERROR, "wtns file: version not supported");
  const uint32_t nsec = rd_u32(head + 8);
  uint64_t pos = 12, off1 = 0, len1 = 0, off2 = 0, len2 = 0;
  bool have1 = false, have2 = false;
  for (uint32_t i = 0; i < nsec; i++) {
    if (pos + 12 > size) throw ProverError(PROVER_ERROR, "wtns file: truncated section table");
    uint8_t sh[12];
    rd(pos, sh, 12);
    const uint32_t id = rd_u32(sh);
    const uint64_t len = rd_u64(sh + 4);
    pos += 12;
    if (len > size - pos) throw ProverError(PROVER_ERROR, "wtns file: truncated section");
    if (id == 1 && !have1) { have1 = true; off1 = pos; len1 = len; }
    if (id == 2 && !have2) { have2 = true; off2 = pos; len2 = len; }
    pos += len;
  }
  if (!have1) throw ProverError(PROVER_ERROR, "missing section 1 (wtns header)");
  uint8_t h1[40];
  if (len1 < 40) throw ProverError(PROVER_ERROR, "wtns header: unsupported field size");
  rd(off1, h1, 40);
  if (rd_u32(h1) != 32) throw ProverError(PROVER_ERROR, "wtns header: unsupported field size");
  if (memcmp(h1 + 4, kR, 32) != 0)
    throw ProverError(PROVER_ERROR, "Curve of the witness does not match the curve of the proving key");
  WtnsView w;
  w.n = rd_u32(h1 + 36);
  if (!have2) throw ProverError(PROVER_ERROR, "missing section 2 (wtns values)");
  if (len2 != (uint64_t)w.n * 32) throw ProverError(PROVER_ERROR, "wtns value section has the wrong size");
  w.fd = fd;
  w.off = off2;
  return w;
}

WtnsView parse_wtns(const WtnsSrc& src) {
  return src.fd >= 0 ? parse_wtns_fd(src.fd, src.size) : parse_wtns(src.buf, src.size);
}

// a table covers the whole resident array: the handle's current range must be that array
bool split_c_partial(const zkpoa_zkey* zk) {
  uint64_t info[4];
  msm_table_info(zk->tC, info);
  return info[0] != zk->ccnt;
}
bool split_h_partial(const zkpoa_zkey* zk) {
  uint64_t info[4];
  msm_table_info(zk->tH, info);
  return info[0] != zk->hcnt;
}

// Fixed-base tables for a resident key, most valuable first, while they fit the budget (bytes; 0 = half of the HBM
// that is free right now -- which leaves room for the per-lane sort / bucket workspaces). H and C first: the H MSM
// closes the critical path and section 8 is the largest G1 array; then the A query; the B query needs both its
// G1 and G2 table (they share one bucket sort). Returns the bytes allocated.
// max_new < 0: all of them at once, replacing what exists (zkpoa_zkey_precompute, the eager policy). max_new >= 0: keep
// what exists and build at most that many more -- the resident prover builds a key's tables one per idle moment, so that
// a request never waits for more than one of them (a whole set is 0.25 s at the layer-one shape, 3-7 s at layers two and
// three: more than twenty proofs' worth, which a workflow of two batches never earns back on the request path).
static size_t lane_workspace_total(const zkpoa_context* ctx, const zkpoa_zkey* zk, const uint64_t lim[5]);   // below
uint64_t zkey_precompute(zkpoa_context* ctx, zkpoa_zkey* zk, uint64_t budget, int max_new = -1) {
  const bool split = zk->split_world > 1;   // H table over the cyclic shard; A / B / C as for any shard
  ctx->dev.wait_lanes();
  ZK_HIP(hipDeviceSynchronize());
  const bool step = max_new >= 0;
  int built = 0;
  bool deferred = false;
  if (!step) zk->release_tables();
  if (step && zk->table_budget) budget = zk->table_budget;
  bool release_lanes = false;
  size_t held = 0;
  if (budget == 0) {
    // the lanes' grow-only workspaces are given back first (they regrow to what the fixed-base plans need) -- counted
    // as free here, released below once it is known that a table will be built at all
    for (auto& l : ctx->dev.lanes) held += l.ws.cap;
    release_lanes = true;
    size_t free_b = 0, total_b = 0;
    ZK_HIP(hipMemGetInfo(&free_b, &total_b));
    free_b += held;
    // ... but never the room the five lanes need for whole MSMs over this key (a chunked MSM cannot use its table, so
    // a table that pushes its own MSM into pieces is HBM spent to be slower): at the reference's shapes the half is the
    // smaller figure (2^26: ~100 GB of workspaces against 230 GB free); on a 2^27 key the workspaces come first
    uint64_t lim[5];
    for (int l = 0; l < 5; l++) lim[l] = ctx->msm_points_limit(l);
    const double room = (double)free_b - 1.1 * (double)lane_workspace_total(ctx, zk, lim);
    budget = (uint64_t)std::max(0.0, std::min(0.5 * (double)free_b, room)) + zk->table_bytes;
  }
  if (step) zk->table_budget = budget;
  const int force_c = ctx->opt_msm_c;
  auto fits = [&](uint64_t bytes) { return zk->table_bytes + bytes <= budget; };
  const uint64_t nH = split ? (zk->dHs ? (uint64_t)(zk->domain >> zk->split_log) : 0) : (zk->dH ? zk->hcnt : 0);
  const uint64_t nC = zk->ccnt, nA = zk->qA.res, nB = zk->qB.res;
  // a table that does not fit after all (allocation failure) ends the list; the ones built so far stay
  auto build = [&](MsmTable** slot, bool g2, const void* bases, uint64_t n, int c) -> bool {
    if (*slot) return true;                     // (step mode: built by an earlier step)
    if (step && built >= max_new) {             // this step's share is done: the rest waits for the next step
      deferred = true;
      return false;
    }
    built++;
    try {
      *slot = g2 ? msm_table_build_g2(ctx, bases, n, c) : msm_table_build_g1(ctx, bases, n, c);
      zk->table_bytes += g2 ? msm_table_bytes_g2(n, c) : msm_table_bytes_g1(n, c);
      return true;
    } catch (const HipError&) {
      (void)hipGetLastError();
      *slot = nullptr;
      return false;
    }
  };
  // H scalars are uniform; the A / B / C tables are sized for witness scalars when a proof has measured some
  auto witness_c = [&](uint64_t n, bool g2) {
    if (const char* e = getenv("ZKPOA_WITNESS_TABLE_C"))   // experiments only
      if (atoi(e) >= 4 && atoi(e) <= 25) return atoi(e);
    msm_set_density_hint(zk->have_density ? zk->witness_density : nullptr);
    const int c = (int)msm_table_width(n, force_c, g2);
    msm_set_density_hint(nullptr);
    return c;
  };
  msm_set_density_hint(nullptr);
  const int cC = nC ? witness_c(nC, false) : 0, cA = nA ? witness_c(nA, false) : 0, cB = nB ? witness_c(nB, true) : 0;
  if (release_lanes) {
    // Nothing to build (a 2^27 key: the lanes' workspaces come first): keep the workspaces. Giving back 150 GB and
    // taking it again costs seconds -- the driver wipes released memory before it hands it out (measured: 5 s per lane).
    const bool any = (!zk->tH && nH && (split || zk->hlo == zk->hbase) && fits(msm_table_bytes_g1(nH, force_c))) ||
                     (!zk->tC && nC && zk->clo == zk->cbase && fits(msm_table_bytes_g1(nC, cC))) ||
                     (!zk->tA && nA && fits(msm_table_bytes_g1(nA, cA))) ||
                     (!(zk->tB1 && zk->tB2) && nB && fits(msm_table_bytes_g1(nB, cB) + msm_table_bytes_g2(nB, cB)));
    if (!any) {
      zk->tables_settled = true;
      return zk->table_bytes;
    }
    for (auto& l : ctx->dev.lanes) l.ws.release();
  }
  bool more = true;
  // (a table that exists already counts as fitting: its bytes are in table_bytes)
  if (more && nH && (split || zk->hlo == zk->hbase) && (zk->tH || fits(msm_table_bytes_g1(nH, force_c)))) {
    more = build(&zk->tH, false, split ? zk->dHs : zk->dH, nH, force_c);
    zk->tH_cyclic = split && zk->tH;
  }
  if (more && nC && zk->clo == zk->cbase && (zk->tC || fits(msm_table_bytes_g1(nC, cC)))) more = build(&zk->tC, false, zk->dC, nC, cC);
  if (more && nA && (zk->tA || fits(msm_table_bytes_g1(nA, cA)))) more = build(&zk->tA, false, zk->qA.g1, nA, cA);
  if (more && nB && !(zk->tB1 && zk->tB2)) {
    // (one window width for both B tables: the G2 model's -- shorter pieces -- as the shared sort is planned for G2)
    if (step && built >= max_new) {   // the pair is one step's work
      more = false;
      deferred = true;
    } else if (fits(msm_table_bytes_g1(nB, cB) + msm_table_bytes_g2(nB, cB))) {
      if (step) max_new = built + 2;              // ... and both halves belong to it
      more = build(&zk->tB1, false, zk->qB.g1, nB, cB) && build(&zk->tB2, true, zk->qB.g2, nB, cB);
      if (!more && zk->tB1) {   // the pair is only usable together
        zk->table_bytes -= msm_table_bytes_g1(nB, cB);
        msm_table_release(zk->tB1);
        zk->tB1 = nullptr;
      }
    }
  }
  // settled: nothing is left for a later step (every wanted table exists, does not fit, or failed to build)
  zk->tables_settled = !deferred;
  return zk->table_bytes;
}

// ---- HBM budget of the five MSM lanes ---------------------------------------------------------------------------
// A lane's workspace is proportional to the points its MSM sorts at once (0.6-0.9 KB per point: 28 GB for the H MSM of a
// 2^26 domain). At the reference's shapes (<= 2^26, 52 M wires) five whole-MSM workspaces, the key and its tables fit in
// 288 GB with room to spare; a key of 2^27 constraints (layer three over four batches) or a card shared with another
// process does not leave that room. The workspaces the stages of the coming proof would reserve (from their plans) are
// therefore added up BEFORE a lane has to grow: if they do not fit in what is free (+ what the lanes hold now), the
// stage with the largest workspace takes its points in halves -- again and again until the sum fits -- and the lanes
// start from empty workspaces. A chunked MSM gives up its fixed-base table and the shared sort of the B query (each is
// indexed by the whole array), so only the stages that must are chunked. What this cannot foresee (a staged one-shot
// prove whose sizes arrive with the file, another process allocating meanwhile) is caught by the failed reservation
// itself: msm_run.
struct LaneNeeds {
  size_t bytes[5] = {0, 0, 0, 0, 0};
  size_t total() const { return bytes[0] + bytes[1] + bytes[2] + bytes[3] + bytes[4]; }
};
static LaneNeeds lane_needs(const zkpoa_context* ctx, const zkpoa_zkey* zk, const uint64_t lim[5], bool with_tables) {
  LaneNeeds w;
  const int fc = ctx->opt_msm_c;
  const bool split = zk->split_world > 1;
  auto table_c = [&](const MsmTable* t, uint64_t n, uint64_t limit) -> int {
    if (!with_tables || !t || n > limit) return 0;
    uint64_t info[4];
    msm_table_info(t, info);
    return info[0] == n ? (int)info[1] : 0;
  };
  const double* dens = zk->have_density ? zk->witness_density : nullptr;
  // H (lane 0): uniform scalars
  msm_set_density_hint(nullptr);
  const uint64_t nH = split ? (uint64_t)(zk->domain >> zk->split_log) : zk->hcnt;
  w.bytes[0] = msm_workspace_g1(std::min(nH, lim[0]), fc, false, table_c(zk->tH, nH, lim[0]), true);
  // A (lane 1), C (lane 4): witness scalars
  msm_set_density_hint(dens);
  w.bytes[1] = msm_workspace_g1(std::min(zk->qA.cnt, lim[1]), fc, false, table_c(zk->tA, zk->qA.cnt, lim[1]), true);
  w.bytes[4] = msm_workspace_g1(std::min(zk->ccnt, lim[4]), fc, false, table_c(zk->tC, zk->ccnt, lim[4]), true);
  // B (lanes 2 and 3): one sort for both when the whole query fits one
  const uint64_t nB = zk->qB.cnt, limB = std::min(lim[2], lim[3]);
  if (nB <= limB) {
    const int tc = (zk->tB1 && zk->tB2) ? table_c(zk->tB1, nB, limB) : 0;
    w.bytes[2] = msm_workspace_g1(nB, fc, true, tc, true);
    w.bytes[3] = msm_workspace_g2(nB, fc, tc, false);
  } else {
    w.bytes[2] = msm_workspace_g1(std::min(nB, lim[2]), fc, false, 0, true);
    w.bytes[3] = msm_workspace_g2(std::min(nB, lim[3]), fc, 0, true);
  }
  msm_set_density_hint(nullptr);
  return w;
}

static size_t lane_workspace_total(const zkpoa_context* ctx, const zkpoa_zkey* zk, const uint64_t lim[5]) {
  return lane_needs(ctx, zk, lim, false).total();   // classic form: what the lanes need if no table is built
}

static void budget_lane_workspaces(zkpoa_context* ctx, const zkpoa_zkey* zk) {
  uint64_t lim[5];
  for (int l = 0; l < 5; l++) lim[l] = ctx->msm_points_limit(l);
  LaneNeeds w = lane_needs(ctx, zk, lim, true);
  size_t held = 0, after = 0;
  bool grow = false, over_cap = false;
  for (int l = 0; l < 5; l++) {
    const Arena& a = ctx->dev.lanes[l].ws;
    held += a.cap;
    after += std::max(a.cap, w.bytes[l]);
    grow = grow || w.bytes[l] > a.cap;
    over_cap = over_cap || (a.limit && w.bytes[l] > a.cap && w.bytes[l] > a.limit);
  }
  if (!grow) return;   // (the steady state: nothing below runs, no driver call)
  size_t free_b = 0, total_b = 0;
  ZK_HIP(hipMemGetInfo(&free_b, &total_b));
  const double keep = 0.95;   // of what is free: the proof's own temporaries are small beside the workspaces
  if (!over_cap && (double)(after - held) <= keep * (double)free_b) return;
  const double avail = keep * ((double)free_b + (double)held);
  auto fits = [&](const LaneNeeds& x) {
    for (int l = 0; l < 5; l++)
      if (ctx->dev.lanes[l].ws.limit && x.bytes[l] > ctx->dev.lanes[l].ws.limit) return false;
    return (double)x.total() <= avail;
  };
  const uint64_t n_of[5] = {zk->split_world > 1 ? (uint64_t)(zk->domain >> zk->split_log) : zk->hcnt, zk->qA.cnt,
                            zk->qB.cnt, zk->qB.cnt, zk->ccnt};
  while (!fits(w)) {
    // the lane that is over its cap first, else the one with the largest workspace; its MSM takes half as many points
    int pick = -1;
    for (int l = 0; l < 5; l++)
      if (ctx->dev.lanes[l].ws.limit && w.bytes[l] > ctx->dev.lanes[l].ws.limit && (pick < 0 || w.bytes[l] > w.bytes[pick]))
        pick = l;
    if (pick < 0)
      for (int l = 0; l < 5; l++)
        if (std::min(n_of[l], lim[l]) > (1ull << 16) && (pick < 0 || w.bytes[l] > w.bytes[pick])) pick = l;
    if (pick < 0 || std::min(n_of[pick], lim[pick]) <= (1ull << 16)) break;   // nothing left to halve: the reservation will say so
    lim[pick] = zkpoa_context::below(std::min(n_of[pick], lim[pick]));
    w = lane_needs(ctx, zk, lim, true);
  }
  for (int l = 0; l < 5; l++)
    if (lim[l] < ctx->msm_points_limit(l)) ctx->oom_max_points[l] = lim[l];
  for (int l = 0; l < 5; l++) {   // every lane starts again from what it needs now (nothing is in flight: prove start)
    ZK_HIP(hipStreamSynchronize(ctx->dev.lanes[l].stream));
    ctx->dev.lanes[l].ws.release();
  }
  if (getenv("ZKPOA_VERBOSE"))
    fprintf(stderr,
            "zkpoa:   HBM budget: %.1f GB free + %.1f GB held by the lanes; MSMs take at most H %llu, A %llu, B1 %llu, B2 %llu, "
            "C %llu points at once (workspaces %.1f GB)\n",
            free_b / 1e9, held / 1e9, (unsigned long long)lim[0], (unsigned long long)lim[1], (unsigned long long)lim[2],
            (unsigned long long)lim[3], (unsigned long long)lim[4], w.total() / 1e9);
}

// Partial MSM results of this handle's shard: A(64) B1(64) B2(128) C(64) H(64). The witness is already
// in zk->d_witness (device). The H-scalar chain runs in full on every rank (replicated).
// Hooks of a one-shot prove that overlaps the key upload with the compute (load_prove_staged): each stage of
// prove_partials first waits for the sections it needs and finishes their key-side preparation on its own lane.
struct Staging {
  std::function<void()> prep_chain, prep_H, prep_A, prep_B, prep_C;
  std::vector<void*> sinks[5];   // temporaries parked until the proof is done (hipFree waits for the whole device)
};

void prove_partials(zkpoa_context* ctx, const zkpoa_zkey* zk, uint8_t out[384], Staging* stg = nullptr) {
  auto t0 = std::chrono::steady_clock::now();
  ctx->dev.wait_lanes();
  Lane& l0 = ctx->dev.lanes[0];
  uint8_t* outA = out;
  uint8_t* outB1 = out + 64;
  uint8_t* outB2 = out + 128;
  uint8_t* outC = out + 256;
  uint8_t* outH = out + 320;
  std::exception_ptr errs[4];
  float msm_ms[5][2] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}, {0, 0}};
  // A, B1 and B2 run over the compacted queries (points at infinity dropped at key load) and gathered scalars. Every
  // device pointer is read INSIDE its stage, after the stage's staging hook: a staged prove allocates and fills the
  // buffers while the stages are already waiting (a pointer taken here could still be null).
  const bool split = zk->split_world > 1;
  if (split && !zk->h_ready)
    throw ProverError(PROVER_ERROR, "split chain: run zkpoa_split_stage1/2/3 for this witness before zkpoa_prove_partials");
  if (ctx->ev_witness_set) {   // an asynchronous witness copy (zkpoa_split_stage1) must land before lanes 1-4 read it
    for (int l = 1; l < 5; l++) ZK_HIP(hipStreamWaitEvent(ctx->dev.lanes[l].stream, ctx->ev_witness, 0));
    ctx->ev_witness_set = false;
  }
  auto guarded = [&](int slot, std::function<void()> fn) {
    return std::thread([&, slot, fn] {
      try {
        ZK_HIP(hipSetDevice(ctx->dev.device));
        if (stg) deferred_free_sink() = &stg->sinks[slot];
        fn();
      } catch (...) {
        errs[slot] = std::current_exception();
      }
    });
  };
  // Witness MSMs on their own lanes (host threads: each MSM has one mid-way read-back). B1 and B2 use the
  // same scalars (the witness values of the wires that occur in the B matrix), so their bucket sort runs once
  // (lane 2) and both accumulations read it; A and C sort on their own lanes.
  std::promise<const MsmSorted*> sorted_promise;
  std::shared_future<const MsmSorted*> sorted_ready = sorted_promise.get_future().share();
  // digit density of this proof's witness (non-zero digits per scalar for every window width), measured once by the
  // first witness stage that gets there, on its own lane; the classic-form witness MSMs size their windows with it
  std::once_flag density_once;
  auto with_density = [&](int lane_id) -> const double* {
    const char* nd = getenv("ZKPOA_NO_DENSITY");
    if ((nd && *nd && strcmp(nd, "0") != 0) || !zk->d_flag || !zk->d_witness) return nullptr;
    std::call_once(density_once, [&] {
      if (zk->have_density) return;
      msm_density(ctx->dev.lanes[lane_id].stream, zk->d_witness, zk->nVars, zk->witness_density,
                  reinterpret_cast<char*>(zk->d_flag) + 256);
      zk->have_density = true;
    });
    return zk->have_density ? zk->witness_density : nullptr;
  };
  // HBM budget of the lanes (a staged prove learns its sizes as the file arrives: msm_run's retry instead). A key's
  // first proof measures the witness's digit density first -- the witness MSMs are planned, and sized, with it
  if (!stg) {
    if (!zk->have_density) (void)with_density(1);
    budget_lane_workspaces(ctx, zk);
  }
  auto gather = [&](int lane_id, const zkpoa_zkey::CompactQuery& q) {
    if (q.cnt)
      hipLaunchKernelGGL(gather32_kernel, dim3((uint32_t)((q.cnt * 2 + 255) / 256)), dim3(256), 0,
                         ctx->dev.lanes[lane_id].stream, (const uint4*)zk->d_witness, (const uint32_t*)q.wire + q.lo,
                         q.cnt, (uint4*)q.scalars);
  };
  // opt_prove_serial (measurement only): every stage runs alone, one after the other, so the per-stage device times
  // are solo times and their sum / the overlapped wall time says how much the five lanes gain (bench.py).
  const bool serial = ctx->opt_prove_serial != 0;
  // fixed-base tables apply when the MSM covers the whole array the table was built from
  std::thread tA = guarded(0, [&] {
    if (stg && stg->prep_A) stg->prep_A();
    const MsmTable* useA = (zk->tA && zk->qA.lo == 0 && zk->qA.cnt == zk->qA.res) ? zk->tA : nullptr;
    const char* pA = reinterpret_cast<const char*>(zk->qA.g1) + zk->qA.lo * 64;
    if (!zk->d_witness || !zk->qA.g1 || !zk->qA.scalars) throw ProverError(PROVER_ERROR, "internal: A stage started before its inputs");
    msm_set_density_hint(with_density(1));
    gather(1, zk->qA);
    msm_run_g1(ctx, 1, pA, zk->qA.scalars, zk->qA.cnt, outA, msm_ms[1], useA);
  });
  if (serial) tA.join();
  // (a B query beyond the 32-bit entry index of one sort cannot share it: two chunked MSMs instead)
  // (nor can one whose whole-query workspace did not fit in HBM on an earlier MSM of this context: msm_points_limit)
  const uint64_t sort_limit = std::min(ctx->msm_points_limit(2), ctx->msm_points_limit(3));
  bool share_b = false;      // set by the B1 stage, read by the B2 stage after the sort has been published
  int table_c_b = 0;
  float sort_b_ms = 0;   // the shared sort of the B query (host clock: the call returns with its stream synchronised)
  std::thread tB1 = guarded(1, [&] {
    MsmSorted* sr = nullptr;
    const char* pB1 = nullptr;
    try {
      if (stg && stg->prep_B) stg->prep_B();
      const bool useB = zk->tB1 && zk->tB2 && zk->qB.lo == 0 && zk->qB.cnt == zk->qB.res;
      share_b = zk->qB.cnt <= sort_limit;
      uint64_t tb_info[4] = {0, 0, 0, 0};
      if (useB) msm_table_info(zk->tB1, tb_info);
      table_c_b = useB && share_b ? (int)tb_info[1] : 0;
      pB1 = reinterpret_cast<const char*>(zk->qB.g1) + zk->qB.lo * 64;
      if (!zk->d_witness || !zk->qB.g1 || !zk->qB.g2 || !zk->qB.scalars)
        throw ProverError(PROVER_ERROR, "internal: B stage started before its inputs");
      msm_set_density_hint(with_density(2));
      gather(2, zk->qB);
      auto ts0 = std::chrono::steady_clock::now();
      if (share_b) {
        try {
          sr = msm_sort_run(ctx, 2, zk->qB.scalars, zk->qB.cnt, true, table_c_b);
        } catch (const OomError& e) {   // no room for the whole query's sort: B1 and B2 each go through it in pieces
          if (!ctx->shrink_after_oom(2, zk->qB.cnt)) throw;
          if (getenv("ZKPOA_VERBOSE")) fprintf(stderr, "zkpoa:   B query: %s; B1 and B2 sort their own pieces\n", e.what());
          share_b = false;
          table_c_b = 0;
        }
      }
      if (!share_b) ZK_HIP(hipStreamSynchronize(ctx->dev.lanes[2].stream));   // the gathered scalars are read on lane 3 too
      sort_b_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - ts0).count();
      sorted_promise.set_value(sr);
    } catch (...) {
      sorted_promise.set_exception(std::current_exception());
      throw;
    }
    if (share_b) msm_accum_g1(ctx, 2, sr, true, table_c_b ? msm_table_data(zk->tB1) : pB1, outB1, msm_ms[2]);
    else msm_run_g1(ctx, 2, pB1, zk->qB.scalars, zk->qB.cnt, outB1, msm_ms[2]);
    if (share_b) msm_ms[2][0] += sort_b_ms;
  });
  if (serial) tB1.join();
  std::thread tB2 = guarded(2, [&] {
    const MsmSorted* sr = sorted_ready.get();
    const char* pB2 = reinterpret_cast<const char*>(zk->qB.g2) + zk->qB.lo * 128;
    bool shared = share_b;
    if (shared) {
      try {
        msm_accum_g2(ctx, 3, sr, false, table_c_b ? msm_table_data(zk->tB2) : pB2, outB2, msm_ms[3]);
      } catch (const OomError& e) {   // the G2 buckets of the whole query did not fit beside the rest: own sort, in pieces
        if (!ctx->shrink_after_oom(3, zk->qB.cnt)) throw;
        if (getenv("ZKPOA_VERBOSE")) fprintf(stderr, "zkpoa:   B2: %s; sorting its own pieces\n", e.what());
        shared = false;
      }
    }
    if (!shared) msm_run_g2(ctx, 3, pB2, zk->qB.scalars, zk->qB.cnt, outB2, msm_ms[3]);
  });
  if (serial) tB2.join();
  std::thread tC = guarded(3, [&] {
    if (stg && stg->prep_C) stg->prep_C();
    const MsmTable* useC = (zk->tC && zk->clo == zk->cbase && !split_c_partial(zk)) ? zk->tC : nullptr;
    const char* pC = reinterpret_cast<const char*>(zk->dC) + (zk->clo - zk->cbase) * 64;
    const char* witC = reinterpret_cast<const char*>(zk->d_witness) + ((uint64_t)zk->nPublic + 1 + zk->clo) * 32;
    if (!zk->d_witness || (zk->ccnt && !zk->dC)) throw ProverError(PROVER_ERROR, "internal: C stage started before its inputs");
    if (zk->bc_log) {   // block-cyclic shard: the scalars of this rank's blocks, gathered in the order of its points
      if (zk->ccnt)
        hipLaunchKernelGGL(gather_bc32_kernel, dim3((uint32_t)((zk->ccnt * 2 + 255) / 256)), dim3(256), 0,
                           ctx->dev.lanes[4].stream,
                           reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(zk->d_witness) + ((uint64_t)zk->nPublic + 1) * 32),
                           zk->ccnt, zk->bc_log, zk->bc_rank, zk->bc_world, (uint4*)zk->d_cscal);
      witC = reinterpret_cast<const char*>(zk->d_cscal);
    }
    msm_set_density_hint(with_density(4));
    msm_run_g1(ctx, 4, pC, witC, zk->ccnt, outC, msm_ms[4], useC);
  });
  if (serial) tC.join();

  std::exception_ptr main_err;
  uint32_t witness_bad = 0;
  try {
    if (stg) deferred_free_sink() = &stg->sinks[4];
    if (stg && stg->prep_chain) stg->prep_chain();
    if (!zk->d_witness || !zk->d_flag || (!split && (!zk->d_abc || !zk->d_row_ptr)))
      throw ProverError(PROVER_ERROR, "internal: chain stage started before its inputs");
    // witness values must be canonical (< r): one streaming pass on the chain's lane, flag read with the H MSM's results
    ZK_HIP(hipMemsetAsync(zk->d_flag, 0, 4, l0.stream));
    range_check<FrParams>(l0.stream, zk->d_witness, zk->nVars, zk->d_flag);
    ZK_HIP(hipEventRecord(ctx->ev_a[5], l0.stream));
    if (!split)
      h_chain(ctx, l0.stream, zk->d_row_ptr, zk->d_sig, zk->d_vals, zk->d_long, zk->n_long, zk->d_witness, zk->domain,
              zk->power, zk->d_abc);
    ZK_HIP(hipEventRecord(ctx->ev_b[5], l0.stream));
    if (stg && stg->prep_H) stg->prep_H();
    const MsmTable* useH = split ? (zk->tH && zk->tH_cyclic ? zk->tH : nullptr)
                                 : ((zk->tH && !zk->tH_cyclic && zk->hlo == zk->hbase && !split_h_partial(zk)) ? zk->tH : nullptr);
    const char* pH = split ? reinterpret_cast<const char*>(zk->dHs)
                           : reinterpret_cast<const char*>(zk->dH) + (zk->hlo - zk->hbase) * 64;
    if ((split ? (zk->dHs == nullptr) : (zk->hcnt && zk->dH == nullptr)))
      throw ProverError(PROVER_ERROR, "internal: H stage started before its inputs");
    if (split) {
      // the three stages left this rank's H scalars (odd-coset indices = rank mod G) in d_abc[0 .. n/G)
      msm_run_g1(ctx, 0, pH, zk->d_abc, zk->domain >> zk->split_log, outH, msm_ms[0], useH);
      zk->h_ready = false;
    } else {
      msm_run_g1(ctx, 0, pH, reinterpret_cast<const char*>(zk->d_abc) + zk->hlo * 32, zk->hcnt, outH, msm_ms[0], useH);
    }
    ZK_HIP(hipEventElapsedTime(&ctx->ms[3], ctx->ev_a[5], ctx->ev_b[5]));
    msm_read_back(l0, zk->d_flag, 4);   // lane 0 is idle (its MSM has returned): the flag through its pinned buffer
    witness_bad = *reinterpret_cast<const uint32_t*>(l0.pinned);
  } catch (...) {
    main_err = std::current_exception();
  }
  deferred_free_sink() = nullptr;
  if (!serial) {
    tA.join();
    tB1.join();
    tB2.join();
    tC.join();
  }
  try {
    msm_sorted_free(const_cast<MsmSorted*>(sorted_ready.get()));
  } catch (...) {
  }
  if (main_err) std::rethrow_exception(main_err);
  for (auto& e : errs)
    if (e) std::rethrow_exception(e);
  if (witness_bad) throw ProverError(PROVER_ERROR, "witness value is not a field element (>= r)");
  auto t1 = std::chrono::steady_clock::now();
  ctx->ms[4] = std::chrono::duration<float, std::milli>(t1 - t0).count();
  ctx->ms[0] = msm_ms[0][0];
  ctx->ms[1] = msm_ms[0][1];
  for (int l = 0; l < 5; l++) {   // per-lane MSM timings (H, A, B1, B2, C): zkpoa_last_ms_lane
    ctx->lane_ms[l][0] = msm_ms[l][0];
    ctx->lane_ms[l][1] = msm_ms[l][1];
  }
}

// header: alpha1(64) beta1(64) beta2(128) delta1(64) delta2(128); sums: A B1 B2 C H summed over all shards.
// Randomised assembly (groth16_prove.js tail). Host only.
void prove_assemble(const uint8_t header[448], const uint8_t sums[384], const uint8_t* r_le, const uint8_t* s_le,
                    uint8_t proof_points[256]) {
  uint8_t rb[32], sb[32];
  if (r_le) memcpy(rb, r_le, 32); else random_scalar(rb);
  if (s_le) memcpy(sb, s_le, 32); else random_scalar(sb);
  Affine<HFq> alpha1 = h_affine_from_bytes<HFq>(header), beta1 = h_affine_from_bytes<HFq>(header + 64);
  Affine<HFq2> beta2 = h_affine_from_bytes<HFq2>(header + 128);
  Affine<HFq> delta1 = h_affine_from_bytes<HFq>(header + 256);
  Affine<HFq2> delta2 = h_affine_from_bytes<HFq2>(header + 320);
  uint64_t rk[4], sk[4], nrs[4];
  memcpy(rk, rb, 32);
  memcpy(sk, sb, 32);
  HFr rs = hfr_from_le(rb).to_mont() * hfr_from_le(sb).to_mont();
  rs.neg().from_mont().to_bytes(nrs);

  XYZZ<HFq> d1 = XYZZ<HFq>::from_affine(delta1);
  XYZZ<HFq2> d2 = XYZZ<HFq2>::from_affine(delta2);
  XYZZ<HFq> pi_a = XYZZ<HFq>::from_affine(h_affine_from_bytes<HFq>(sums));
  xyzz_add_affine(pi_a, alpha1, false);
  xyzz_add(pi_a, h_mul(d1, rk));
  XYZZ<HFq2> pi_b = XYZZ<HFq2>::from_affine(h_affine_from_bytes<HFq2>(sums + 128));
  xyzz_add_affine(pi_b, beta2, false);
  xyzz_add(pi_b, h_mul(d2, sk));
  XYZZ<HFq> pib1 = XYZZ<HFq>::from_affine(h_affine_from_bytes<HFq>(sums + 64));
  xyzz_add_affine(pib1, beta1, false);
  xyzz_add(pib1, h_mul(d1, sk));
  XYZZ<HFq> pi_c = XYZZ<HFq>::from_affine(h_affine_from_bytes<HFq>(sums + 256));
  xyzz_add_affine(pi_c, h_affine_from_bytes<HFq>(sums + 320), false);
  xyzz_add(pi_c, h_mul(pi_a, sk));
  xyzz_add(pi_c, h_mul(pib1, rk));
  xyzz_add(pi_c, h_mul(d1, nrs));

  h_affine_to_bytes<HFq>(h_to_affine(pi_a), proof_points);
  h_affine_to_bytes<HFq2>(h_to_affine(pi_b), proof_points + 64);
  h_affine_to_bytes<HFq>(h_to_affine(pi_c), proof_points + 192);
}

void zkey_header_bytes(const zkpoa_zkey* zk, uint8_t out[448]) {
  h_affine_to_bytes<HFq>(zk->alpha1, out);
  h_affine_to_bytes<HFq>(zk->beta1, out + 64);
  h_affine_to_bytes<HFq2>(zk->beta2, out + 128);
  h_affine_to_bytes<HFq>(zk->delta1, out + 256);
  h_affine_to_bytes<HFq2>(zk->delta2, out + 320);
}

bool is_full_key(const zkpoa_zkey* zk) {
  return zk->split_world <= 1 && zk->wlo == 0 && zk->wcnt == zk->nVars && zk->clo == 0 && zk->ccnt == (uint64_t)zk->nVars - zk->nPublic - 1 &&
         zk->hlo == 0 && zk->hcnt == zk->domain;
}

// unsharded prove = partials of the whole key + assembly
void prove_core(zkpoa_context* ctx, const zkpoa_zkey* zk, const uint8_t* r_le, const uint8_t* s_le,
                uint8_t proof_points[256]) {
  auto t0 = std::chrono::steady_clock::now();
  if (!is_full_key(zk))
    throw ProverError(PROVER_ERROR, "this key handle is a shard: use zkpoa_prove_partials + zkpoa_prove_assemble");
  uint8_t parts[384], header[448];
  prove_partials(ctx, zk, parts);
  zkey_header_bytes(zk, header);
  prove_assemble(header, parts, r_le, s_le, proof_points);
  ctx->ms[5] = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
}

// Self-check (ZKPOA_SELFCHECK): the reference verifies every proof right after proving it
// (scripts/g16_verify.sh:213-216, full_workflow.sh:503-504); a zkey carries its own verification key, so the prover
// can run the same pairing check before anything is written. "0" = never, unset / "1" = the first proof of every key
// handle (a key whose conventions differ from SURVEY.md 8c fails on first contact, not downstream), "all" = every
// proof. Host only (csrc/verify.hip, ~7 ms). Keys without section 3 / assembled from device buffers are skipped.
int selfcheck_mode() {
  const char* e = getenv("ZKPOA_SELFCHECK");
  if (!e || !*e) return 1;
  if (!strcmp(e, "0") || !strcmp(e, "off")) return 0;
  if (!strcmp(e, "all") || !strcmp(e, "2")) return 2;
  return 1;
}

struct SelfCheckFailed : ProverError {   // the proof was computed and does not verify (as opposed to "could not be checked")
  explicit SelfCheckFailed(const std::string& s) : ProverError(PROVER_ERROR, s) {}
};

// first_n: in the default mode the first that many proofs of a key are checked (1; the multi-GPU path checks 3: a stale
// exchange buffer can only show from the second proof on)
void selfcheck(zkpoa_context* ctx, const zkpoa_zkey* zk, const uint8_t proof_points[256], const uint8_t* public_le,
               uint64_t first_n = 1) {
  ctx->ms[6] = 0;
  const int mode = selfcheck_mode();
  if (mode == 0 || zk->vkey_points.empty() || (mode == 1 && zk->selfchecks_done >= first_n)) return;
  auto t0 = std::chrono::steady_clock::now();
  char msg[256] = {0};
  int rc = zkpoa_groth16_verify_points(zk->vkey_points.data(), (unsigned long)zk->vkey_points.size(), proof_points,
                                       public_le, zk->nPublic, msg, sizeof(msg));
  ctx->ms[6] = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
  if (rc == PROVER_OK) {
    zk->selfchecks_done++;
    if (req_getenv("ZKPOA_VERBOSE")) fprintf(stderr, "zkpoa: self-check: proof verifies against the zkey's own verification key (%.1f ms)\n", ctx->ms[6]);
    return;
  }
  if (rc == ZKPOA_VERIFY_INVALID_PROOF)
    throw SelfCheckFailed(
                      "self-check failed: the proof does not verify against the verification key inside the zkey "
                      "(sections 2-3). Either the witness does not satisfy the circuit, or this zkey does not follow the "
                      "conventions the prover assumes: section 4 coefficients stored as coef*R^2 mod r; section 9 H points = "
                      "Lagrange basis of the odd coset of the 2n-th roots (no division by Z); section 8 C points starting at "
                      "wire nPublic+1; all points affine in Montgomery form. Set ZKPOA_SELFCHECK=0 to write the proof anyway.");
  throw ProverError(PROVER_ERROR, std::string("self-check could not run: ") + msg);
}

void prove_impl(zkpoa_context* ctx, const zkpoa_zkey* zk, const WtnsSrc& wsrc,
                const uint8_t* r_le, const uint8_t* s_le, uint8_t proof_points[256], uint8_t* public_le,
                uint64_t public_cap) {
  WtnsView w = parse_wtns(wsrc);
  if (w.n != zk->nVars)
    throw ProverError(PROVER_INVALID_WITNESS_LENGTH, "Invalid witness length. Circuit: " + std::to_string(zk->nVars) +
                                                         ", witness: " + std::to_string(w.n));
  if (public_cap < (uint64_t)zk->nPublic * 32) throw ProverError(PROVER_ERROR_SHORT_BUFFER, "public buffer too small");
  Lane& l0 = ctx->dev.lanes[0];
  (void)l0;
  {
    const auto tu = std::chrono::steady_clock::now();
    w.upload(ctx, zk->d_witness, 0, w.n, ctx->dev.lanes[0].stream);
    ctx->io_ms[0] = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - tu).count();
    ctx->io_ms[1] = (float)((double)w.n * 32 / 1e6);
  }
  prove_core(ctx, zk, r_le, s_le, proof_points);
  std::vector<uint8_t> pub_store;
  const uint8_t* pubs = w.publics(zk->nPublic, pub_store);
  memcpy(public_le, pubs, (size_t)zk->nPublic * 32);
  selfcheck(ctx, zk, proof_points, pubs);
}

// ---- one-shot prove with the key upload overlapped ------------------------------------------
// A one-shot `prover` run spends most of its time moving the key over PCIe (1 GB for layer one, 21 GB for layer
// three), and none of the compute needs the whole key: the H-scalar chain needs the witness and section 4, the H MSM
// section 9, the B MSMs sections 6 + 7, A section 5, C section 8. So the sections go up on a stream of their own in
// the order  witness, 4, 9, 6, 7, 5, 8  (the stage with the least work after its last byte arrives goes last) while
// every stage of prove_partials starts as soon as its own inputs are resident and does the key-side preparation of
// that section (CSR build, dropping the points at infinity, range checks) on its own lane. Wall time tends to
// max(upload, compute) + the C MSM instead of upload + compute. Returns the loaded key (complete, reusable: the cache
// keeps it) with `parts` = the five MSM results; nullptr when no copy stream could be created (caller: plain path).
// zkey_fd >= 0: the file `buf` maps; the sections are then read with pread instead of through the mapping.
zkpoa_zkey* load_prove_staged(zkpoa_context* ctx, const uint8_t* buf, uint64_t size, const WtnsView& w, uint8_t parts[384],
                              int zkey_fd = -1) {
  hipStream_t cs = ctx->dev.copy_stream_wait();
  if (!cs) return nullptr;
  ZkeySections zs;
  std::unique_ptr<zkpoa_zkey> zk = zkey_parse(buf, size, zs);
  if (w.n != zk->nVars)
    throw ProverError(PROVER_INVALID_WITNESS_LENGTH, "Invalid witness length. Circuit: " + std::to_string(zk->nVars) +
                                                         ", witness: " + std::to_string(w.n));
  zk->set_full();
  const uint64_t m = zk->nVars, n = zk->domain, nC = m - zk->nPublic - 1;
  {
    // What the overlapped form holds at its peak: the sections AND the compacted copies of the A / B queries (a hipFree
    // would wait for the whole device, so the originals are only given back after the proof), the coefficient records
    // AND their CSR form, the chain's work area, the witness -- plus the lanes' workspaces. A key for which that leaves
    // the lanes less than a quarter of the card (2^28: 247 of 309 GB; 2^27: 125) is loaded first and proved afterwards (the caller's
    // sequential path: temporaries freed as it goes, then the lanes' HBM budget of prove_partials).
    size_t free_b = 0, total_b = 0;
    ZK_HIP(hipMemGetInfo(&free_b, &total_b));
    const double peak = (double)n * 64 + (double)m * (64 + 128 + 64) + (double)nC * 64 + (double)m * 32 +
                        (double)zk->nCoefs * (44 + 36) + (double)n * (8 + 96) + (double)m * (108 + 236);
    if (peak > 0.72 * (double)total_b || peak > 0.8 * (double)free_b) {
      if (req_getenv("ZKPOA_VERBOSE"))
        fprintf(stderr, "zkpoa: the overlapped load would hold %.0f GB at its peak (%.0f GB free of %.0f): loading the key first\n",
                peak / 1e9, free_b / 1e9, total_b / 1e9);
      return nullptr;
    }
  }
  // What of that is still to be allocated while the lanes already reserve their workspaces: every part is taken off as
  // it is allocated, and a lane that would eat into the rest halves its piece instead (Arena::reserve, msm_run).
  struct HoldBack {
    std::atomic<int64_t>& v;
    explicit HoldBack(std::atomic<int64_t>& x, double bytes) : v(x) { v = (int64_t)bytes; }
    void done(double bytes) {
      if (v.fetch_sub((int64_t)bytes) - (int64_t)bytes < 0) v = 0;
    }
    ~HoldBack() { v = 0; }
  } hold(ctx->key_hold_back, (double)n * 64 + (double)m * 256 + (double)nC * 64 + (double)zk->nCoefs * 36 +
                               (double)n * (8 + 96) + (double)m * (108 + 236));
  zkpoa_zkey* k = zk.get();
  void* d_recs = nullptr;
  uint32_t* d_cflag = nullptr;
  enum { S_WIT, S_COEF, S_H, S_B1, S_B2, S_A, S_C, S_COUNT };
  std::promise<void> ready[S_COUNT];
  std::shared_future<void> have[S_COUNT];
  for (int i = 0; i < S_COUNT; i++) have[i] = ready[i].get_future().share();
  std::atomic<bool> cancel{false};
  std::thread uploader;
  Staging stg;
  const bool verbose = req_getenv("ZKPOA_VERBOSE") != nullptr;
  const auto t_start = std::chrono::steady_clock::now();
  auto mark = [&, verbose](const char* what) {   // ZKPOA_VERBOSE: the timeline of an overlapped load + prove
    if (verbose)
      fprintf(stderr, "zkpoa: staged %7.1f ms  %s\n",
              std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count(), what);
  };
  auto cleanup_temps = [&] {
    (void)hipDeviceSynchronize();
    for (auto& v : stg.sinks) {
      for (void* p : v) (void)hipFree(p);
      v.clear();
    }
    if (d_recs) (void)hipFree(d_recs);
    if (d_cflag) (void)hipFree(d_cflag);
    d_recs = nullptr;
    d_cflag = nullptr;
  };
  try {
    auto alloc = [](void** p, uint64_t bytes) { ZK_HIP(hipMalloc(p, bytes ? bytes : 1)); };
    alloc(reinterpret_cast<void**>(&d_cflag), 64);
    ZK_HIP(hipMemsetAsync(d_cflag, 0, 64, ctx->dev.lanes[0].stream));   // stream-ordered, then waited for: the range checks
    ZK_HIP(hipStreamSynchronize(ctx->dev.lanes[0].stream));              // that OR into it run on other (non-blocking) streams
    alloc(&k->d_witness, m * 32);
    alloc(&d_recs, zk->nCoefs * 44);
    alloc(reinterpret_cast<void**>(&k->d_flag), 1024);
    // each buffer is allocated by the uploader right before its section moves (hipMalloc of GBs is milliseconds each)
    struct Item { int id; void** dst; const uint8_t* src; uint64_t bytes; };
    const Item items[S_COUNT] = {
        {S_WIT, &k->d_witness, w.values /* null: from w.fd */, m * 32}, {S_COEF, &d_recs, zs.s4.p + 4, zk->nCoefs * 44},
        {S_H, &k->dH, zs.s9.p, n * 64},           {S_B1, &k->dB1, zs.s6.p, m * 64},
        {S_B2, &k->dB2, zs.s7.p, m * 128},        {S_A, &k->dA, zs.s5.p, m * 64},
        {S_C, &k->dC, zs.s8.p, nC * 64}};
    uploader = std::thread([&] {
      int i = 0;
      try {
        ZK_HIP(hipSetDevice(ctx->dev.device));
        for (; i < S_COUNT; i++) {
          if (cancel.load()) throw ProverError(PROVER_ERROR, "upload cancelled");
          if (!*items[i].dst) {
            ZK_HIP(hipMalloc(items[i].dst, items[i].bytes ? items[i].bytes : 1));
            hold.done((double)items[i].bytes);
          }
          if (items[i].bytes) {
            if (items[i].id == S_WIT && w.fd >= 0) {   // file-backed witness: pread into the pinned staging buffers
              std::vector<uint8_t> small;
              if (items[i].bytes < (4u << 20)) {
                small.resize(items[i].bytes);
                w.read(0, w.n, small.data());
                ZK_HIP(hipMemcpyAsync(*items[i].dst, small.data(), items[i].bytes, hipMemcpyHostToDevice, cs));
                ZK_HIP(hipStreamSynchronize(cs));
              } else {
                ctx->uploader.upload(*items[i].dst, nullptr, items[i].bytes, ctx->dev.device, cs, w.fd, w.off);
              }
            } else if (items[i].bytes < (4u << 20)) {   // small: one asynchronous copy on the copy stream (no null-stream copy here)
              ZK_HIP(hipMemcpyAsync(*items[i].dst, items[i].src, items[i].bytes, hipMemcpyHostToDevice, cs));
              ZK_HIP(hipStreamSynchronize(cs));
            } else {
              const bool from_file = zkey_fd >= 0 && items[i].src >= buf && items[i].src < buf + size;
              ctx->uploader.upload(*items[i].dst, items[i].src, items[i].bytes, ctx->dev.device, cs,
                                   from_file ? zkey_fd : -1, from_file ? (uint64_t)(items[i].src - buf) : 0);
            }
          }
          ready[items[i].id].set_value();
          static const char* kNames[S_COUNT] = {"witness resident", "section 4 resident", "section 9 (H) resident",
                                                "section 6 (B1) resident", "section 7 (B2) resident",
                                                "section 5 (A) resident", "section 8 (C) resident"};
          mark(kNames[items[i].id]);
        }
      } catch (...) {
        for (; i < S_COUNT; i++) ready[items[i].id].set_exception(std::current_exception());
      }
    });
    ctx->dev.wait_lanes();   // the other lanes come up in the background while the first sections are already moving
    Lane* L = ctx->dev.lanes;
    stg.prep_chain = [&, k] {
      have[S_WIT].get();
      have[S_COEF].get();
      build_csr(ctx, k, d_recs, false, L[0].stream);
      ntt_prepare(ctx, L[0].stream, k->power);
      hold.done((double)k->nCoefs * 36 + (double)n * (8 + 96));
      mark("CSR + NTT tables built, H-scalar chain starts");
    };
    stg.prep_H = [&, k] {
      mark("H-scalar chain enqueued");
      have[S_H].get();
      range_check<FqParams>(L[0].stream, k->dH, n * 2, d_cflag);
    };
    stg.prep_A = [&, k] {
      have[S_WIT].get();
      have[S_A].get();
      range_check<FqParams>(L[1].stream, k->dA, m * 2, d_cflag);
      query_compact(ctx, k, k->qA, k->dA, nullptr, m, L[1].stream);
      hold.done((double)m * 108);
      mark("A query compacted, A MSM starts");
    };
    stg.prep_B = [&, k] {
      have[S_WIT].get();
      have[S_B1].get();
      have[S_B2].get();
      range_check<FqParams>(L[2].stream, k->dB1, m * 2, d_cflag);
      range_check<FqParams>(L[2].stream, k->dB2, m * 4, d_cflag);
      query_compact(ctx, k, k->qB, k->dB1, k->dB2, m, L[2].stream);
      hold.done((double)m * 236);
      mark("B query compacted, B MSMs start");
    };
    stg.prep_C = [&, k] {
      have[S_WIT].get();
      have[S_C].get();
      range_check<FqParams>(L[4].stream, k->dC, nC * 2, d_cflag);
    };
    mark("lanes up");
    prove_partials(ctx, k, parts, &stg);
    mark("all five MSMs done");
    uploader.join();
    uint32_t bad = 0;
    ZK_HIP(hipDeviceSynchronize());
    ZK_HIP(hipMemcpy(&bad, d_cflag, 4, hipMemcpyDeviceToHost));
    if (bad) throw ProverError(PROVER_ERROR, "zkey point coordinate is not a field element (>= q)");
    // the originals of the compacted queries are not read again
    stg.sinks[4].push_back(k->dA);
    stg.sinks[4].push_back(k->dB1);
    stg.sinks[4].push_back(k->dB2);
    k->dA = k->dB1 = k->dB2 = nullptr;
    cleanup_temps();
    mark("temporaries freed");
  } catch (...) {
    cancel.store(true);
    if (uploader.joinable()) uploader.join();
    cleanup_temps();
    zk->release();
    throw;
  }
  return zk.release();
}

// ---- JSON (byte formats pinned by the reference's committed fixtures) --------
std::string fq_dec(const uint8_t* le_mont) { return HFq::from_bytes(le_mont).to_dec(); }
bool all_zero(const uint8_t* p, size_t n) {
  for (size_t i = 0; i < n; i++)
    if (p[i]) return false;
  return true;
}

std::string proof_json(const uint8_t pts[256], int style) {
  // coordinates as decimal strings
  std::string a[3], b[3][2], c[3];
  auto g1 = [&](const uint8_t* p, std::string out[3]) {
    if (all_zero(p, 64)) { out[0] = "0"; out[1] = "1"; out[2] = "0"; return; }
    out[0] = fq_dec(p); out[1] = fq_dec(p + 32); out[2] = "1";
  };
  g1(pts, a);
  g1(pts + 192, c);
  const uint8_t* pb = pts + 64;
  if (all_zero(pb, 128)) {
    b[0][0] = "0"; b[0][1] = "0"; b[1][0] = "1"; b[1][1] = "0"; b[2][0] = "0"; b[2][1] = "0";
  } else {
    b[0][0] = fq_dec(pb); b[0][1] = fq_dec(pb + 32); b[1][0] = fq_dec(pb + 64); b[1][1] = fq_dec(pb + 96);
    b[2][0] = "1"; b[2][1] = "0";
  }
  std::string o;
  auto q = [](const std::string& s) { return "\"" + s + "\""; };
  if (style == 0) {  // rapidsnark / nlohmann dump(): one line, no spaces
    o += "{\"pi_a\":[" + q(a[0]) + "," + q(a[1]) + "," + q(a[2]) + "],";
    o += "\"pi_b\":[[" + q(b[0][0]) + "," + q(b[0][1]) + "],[" + q(b[1][0]) + "," + q(b[1][1]) + "],[" + q(b[2][0]) +
         "," + q(b[2][1]) + "]],";
    o += "\"pi_c\":[" + q(c[0]) + "," + q(c[1]) + "," + q(c[2]) + "],";
    o += "\"protocol\":\"groth16\"}";
  } else {  // snarkjs: JSON.stringify(obj, null, 1)
    auto g1s = [&](const char* key, const std::string v[3]) {
      return std::string(" \"") + key + "\": [\n  " + q(v[0]) + ",\n  " + q(v[1]) + ",\n  " + q(v[2]) + "\n ],\n";
    };
    o += "{\n";
    o += g1s("pi_a", a);
    o += " \"pi_b\": [\n";
    for (int i = 0; i < 3; i++) {
      o += "  [\n   " + q(b[i][0]) + ",\n   " + q(b[i][1]) + "\n  ]";
      o += (i < 2) ? ",\n" : "\n";
    }
    o += " ],\n";
    o += g1s("pi_c", c);
    o += " \"protocol\": \"groth16\",\n \"curve\": \"bn128\"\n}";
  }
  return o;
}

std::string public_json(const uint8_t* pub, uint64_t n, int style) {
  std::string o;
  if (style == 0) {
    o = "[";
    for (uint64_t i = 0; i < n; i++) {
      HFr v = HFr::from_bytes(pub + 32 * i).to_mont();
      o += (i ? ",\"" : "\"") + v.to_dec() + "\"";
    }
    o += "]";
  } else {
    if (n == 0) return "[]";
    o = "[\n";
    for (uint64_t i = 0; i < n; i++) {
      HFr v = HFr::from_bytes(pub + 32 * i).to_mont();
      o += " \"" + v.to_dec() + "\"" + (i + 1 < n ? ",\n" : "\n");
    }
    o += "]";
  }
  return o;
}

int emit(const std::string& s, char* buffer, unsigned long* size) {
  if (!size) return PROVER_ERROR;
  unsigned long needed = (unsigned long)s.size() + 1;
  if (!buffer || *size < needed) {
    *size = needed;
    return PROVER_ERROR_SHORT_BUFFER;
  }
  memcpy(buffer, s.c_str(), needed);
  *size = needed;
  return PROVER_OK;
}

zkpoa_context* g_ctx = nullptr;
std::mutex g_prove_mutex;   // one-shot entry points share the process-wide context: one proof at a time
// The file entry point is entered by several threads of a resident server. Two stages, two locks: g_stage_mutex covers
// the key cache and the upload of a request's witness into a free staging buffer of its key; g_prove_mutex the proof
// itself. A request whose key is resident and already has its tables stages its witness while the request before it
// is still proving -- at the layer-three size that is 31 ms of PCIe time per proof taken off the proof-to-proof period.
// Anything that changes the cache or a key (a load, the second-use table build, an eviction) waits until no staged
// request is pending and then holds both locks.
std::mutex g_stage_mutex;
std::condition_variable g_stage_cv;
int g_staged_users = 0;

// the code of a failure for the two file entry points + its message; call inside a catch (...) block
int classify_current_exception(char* error_msg, unsigned long error_msg_maxsize) {
  try {
    throw;
  } catch (const ProverError& e) {
    set_err(error_msg, error_msg_maxsize, e.what());
    return e.code;
  } catch (const HipError& e) {            // HIP runtime failure: the context and its cached keys are suspect
    set_err(error_msg, error_msg_maxsize, e.what());
    return PROVER_ERROR_RUNTIME;
  } catch (const std::bad_alloc&) {
    set_err(error_msg, error_msg_maxsize, "out of host memory");
    return PROVER_ERROR_RUNTIME;
  } catch (const std::system_error& e) {   // a stage thread could not be started
    set_err(error_msg, error_msg_maxsize, e.what());
    return PROVER_ERROR_RUNTIME;
  } catch (const std::exception& e) {
    set_err(error_msg, error_msg_maxsize, e.what());
    return PROVER_ERROR;
  }
}

struct DeviceSet;
DeviceSet* process_devices(uint32_t power, std::string& err, int* code);   // multi_device.hip.h: the device list of this process

// r, s from the environment (ZKPOA_R / ZKPOA_S, decimal; test use) -> pointers, or null for /dev/urandom
void env_blinding(uint8_t rb[32], uint8_t sb[32], const uint8_t*& rp, const uint8_t*& sp) {
  rp = sp = nullptr;
  if (req_getenv("ZKPOA_R") || req_getenv("ZKPOA_S")) {
    static bool warned = false;
    if (!warned) {
      warned = true;
      fprintf(stderr, "zkpoa: WARNING: blinding scalars fixed by ZKPOA_R / ZKPOA_S (test use): proofs made this way are "
                      "not zero-knowledge; unset them in production\n");
    }
  }
  if (const char* e = req_getenv("ZKPOA_R")) {
    if (!parse_decimal_mod_r(e, rb)) throw ProverError(PROVER_ERROR, "ZKPOA_R is not a decimal number");
    rp = rb;
  }
  if (const char* e = req_getenv("ZKPOA_S")) {
    if (!parse_decimal_mod_r(e, sb)) throw ProverError(PROVER_ERROR, "ZKPOA_S is not a decimal number");
    sp = sb;
  }
}

// proof points + public values -> the two JSON texts (ZKPOA_JSON style) + the ZKPOA_VERBOSE phase line
int emit_outputs(zkpoa_context* ctx, const zkpoa_zkey* zk, const uint8_t pts[256], const uint8_t* pub, char* proof_buffer,
                 unsigned long* proof_size, char* public_buffer, unsigned long* public_size, char* error_msg,
                 unsigned long error_msg_maxsize, double load_ms, uint64_t zkey_size, const char* how) {
  int rc = PROVER_OK;
  int style = 0;
  if (const char* e = req_getenv("ZKPOA_JSON")) style = (strcmp(e, "snarkjs") == 0) ? 1 : 0;
  std::string pj = proof_json(pts, style), uj = public_json(pub, zk->nPublic, style);
  int r1 = emit(pj, proof_buffer, proof_size);
  int r2 = emit(uj, public_buffer, public_size);
  if (r1 != PROVER_OK || r2 != PROVER_OK) {
    rc = PROVER_ERROR_SHORT_BUFFER;
    set_err(error_msg, error_msg_maxsize, "output buffer too small");
  }
  if (req_getenv("ZKPOA_VERBOSE")) {
    fprintf(stderr,
            "zkpoa: nVars=%u nPublic=%u domain=2^%u nCoefs=%llu | zkey %s %.1f ms (%.2f GB/s) | witness -> HBM %.2f ms "
            "(%.0f MB, %.1f GB/s) | h-chain %.2f ms, msm phase %.2f ms, prove %.2f ms, self-check %.2f ms\n",
            zk->nVars, zk->nPublic, zk->power, (unsigned long long)zk->nCoefs, how, load_ms,
            load_ms > 0 ? (double)zkey_size / load_ms / 1e6 : 0.0, ctx->io_ms[0], ctx->io_ms[1],
            ctx->io_ms[0] > 0 ? ctx->io_ms[1] / ctx->io_ms[0] : 0.0, ctx->ms[3], ctx->ms[4], ctx->ms[5], ctx->ms[6]);
  }
  return rc;
}

// prove with a resident key, JSON out; options from the environment (ZKPOA_R / ZKPOA_S / ZKPOA_JSON / ZKPOA_VERBOSE)
int prove_to_json(zkpoa_context* ctx, const zkpoa_zkey* zk, const WtnsSrc& wsrc, char* proof_buffer,
                  unsigned long* proof_size, char* public_buffer, unsigned long* public_size, char* error_msg,
                  unsigned long error_msg_maxsize, double load_ms, uint64_t zkey_size, bool cache_hit) {
  uint8_t rb[32], sb[32];
  const uint8_t *rp = nullptr, *sp = nullptr;
  env_blinding(rb, sb, rp, sp);
  uint8_t pts[256];
  std::vector<uint8_t> pub((size_t)zk->nPublic * 32 + 1);
  prove_impl(ctx, zk, wsrc, rp, sp, pts, pub.data(), pub.size());
  return emit_outputs(ctx, zk, pts, pub.data(), proof_buffer, proof_size, public_buffer, public_size, error_msg,
                      error_msg_maxsize, load_ms, zkey_size, cache_hit ? "cached," : "load");
}

// One-shot: load the key and prove, with the upload overlapped unless ZKPOA_OVERLAP=0. *out_zk <- the loaded key
// (the caller frees or caches it). load_ms <- time to the end of the proof (load and prove are one phase here).
int load_and_prove_to_json(zkpoa_context* ctx, const uint8_t* zkey, uint64_t zkey_size, const WtnsSrc& wsrc,
                           char* proof_buffer, unsigned long* proof_size, char* public_buffer,
                           unsigned long* public_size, char* error_msg, unsigned long error_msg_maxsize,
                           zkpoa_zkey** out_zk, int zkey_fd = -1) {
  *out_zk = nullptr;
  const char* ov = getenv("ZKPOA_OVERLAP");
  auto tl0 = std::chrono::steady_clock::now();
  if (!ov || strcmp(ov, "0") != 0) {
    WtnsView w = parse_wtns(wsrc);
    uint8_t rb[32], sb[32], parts[384], header[448], pts[256];
    const uint8_t *rp = nullptr, *sp = nullptr;
    env_blinding(rb, sb, rp, sp);
    auto t0 = std::chrono::steady_clock::now();
    zkpoa_zkey* zk = load_prove_staged(ctx, zkey, zkey_size, w, parts, zkey_fd);
    if (zk) {
      *out_zk = zk;
      zkey_header_bytes(zk, header);
      prove_assemble(header, parts, rp, sp, pts);
      const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
      ctx->ms[5] = (float)ms;
      std::vector<uint8_t> pub_store;
      const uint8_t* pubs = w.publics(zk->nPublic, pub_store);
      selfcheck(ctx, zk, pts, pubs);
      return emit_outputs(ctx, zk, pts, pubs, proof_buffer, proof_size, public_buffer, public_size, error_msg,
                          error_msg_maxsize, ms, zkey_size, "load overlapped with the prove:");
    }
  }
  zkpoa_zkey* zk = zkey_load_impl(ctx, zkey, zkey_size);
  *out_zk = zk;
  const double load_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tl0).count();
  return prove_to_json(ctx, zk, wsrc, proof_buffer, proof_size, public_buffer, public_size, error_msg,
                       error_msg_maxsize, load_ms, zkey_size, false);
}

#include "multi_device.hip.h"

// log2(domain) of a zkey image, for the automatic device selection (throws what zkey_parse throws)
uint32_t zkey_power(const uint8_t* buf, uint64_t size) {
  ZkeySections zs;
  return zkey_parse(buf, size, zs)->power;
}

int one_shot(const uint8_t* zkey, uint64_t zkey_size, const WtnsSrc& wsrc, char* proof_buffer,
             unsigned long* proof_size, char* public_buffer, unsigned long* public_size, char* error_msg,
             unsigned long error_msg_maxsize) {
  std::string err;
  DeviceSet* ds = nullptr;
  int dcode = PROVER_ERROR;
  try {
    ds = process_devices(zkey_power(zkey, zkey_size), err, &dcode);
  } catch (const std::exception& e) {   // malformed key: nothing touches a GPU
    set_err(error_msg, error_msg_maxsize, e.what());
    return PROVER_ERROR;
  }
  if (!ds) {
    set_err(error_msg, error_msg_maxsize, err);
    return dcode;
  }
  zkpoa_context* ctx = ds->ctx[0];
  std::lock_guard<std::mutex> lk(g_prove_mutex);
  zkpoa_zkey* zk = nullptr;
  MultiKey* mk = nullptr;
  int rc = PROVER_OK;
  try {
    ZK_HIP(hipSetDevice(ctx->dev.device));
    if (ds->ids.size() > 1) {   // one proof over all ranks of the process
      mk = multi_key_load(ds, zkey, zkey_size);
      rc = multi_prove_to_json(ds, mk, wsrc, proof_buffer, proof_size, public_buffer, public_size, error_msg,
                               error_msg_maxsize, zkey_size, false);
    } else {
      rc = load_and_prove_to_json(ctx, zkey, zkey_size, wsrc, proof_buffer, proof_size, public_buffer,
                                  public_size, error_msg, error_msg_maxsize, &zk);
    }
  } catch (const ProverError& e) {
    set_err(error_msg, error_msg_maxsize, e.what());
    rc = e.code;
  } catch (const HipError& e) {            // HIP runtime failure: the context is suspect (PROVER_ERROR_RUNTIME)
    set_err(error_msg, error_msg_maxsize, e.what());
    rc = PROVER_ERROR_RUNTIME;
  } catch (const std::bad_alloc&) {
    set_err(error_msg, error_msg_maxsize, "out of host memory");
    rc = PROVER_ERROR_RUNTIME;
  } catch (const std::system_error& e) {   // a stage thread could not be started
    set_err(error_msg, error_msg_maxsize, e.what());
    rc = PROVER_ERROR_RUNTIME;
  } catch (const std::exception& e) {
    set_err(error_msg, error_msg_maxsize, e.what());
    rc = PROVER_ERROR;
  }
  if (zk) {
    zk->release();
    delete zk;
  }
  if (mk) multi_key_release(ds, mk);
  return rc;
}

// ---- device-resident key cache of groth16_prover_zkey_file ("optional device-resident zkey
// cache keyed by path+mtime") ----------------------------------------------------------------------------
// A long-lived caller (the prover server behind the CLI, or an FFI host process) proves many witnesses against
// the same few keys (full_workflow.sh: layer one and two once per batch); uploading 1-21 GB and rebuilding the
// CSR each time is most of a call. Keyed by (device, inode, size, mtime): a rewritten file is a different key.
// ZKPOA_KEY_CACHE = number of keys kept (default 2, 0 = off); least recently used goes first, and everything
// goes when an upload runs out of HBM.
// When a resident key gets its fixed-base tables (ZKPOA_PRECOMP). A whole set costs 0.25 s at the layer-one shape and
// 3-7 s at layers two and three -- twenty to thirty proofs' worth -- and saves ~10 % per proof, so it pays after a few
// hundred proofs. r02 / r03 built it on the request path at the key's SECOND use: a workflow of two batches
// (tests/4_sigs_2_batches_12_height) then spent 3.4 s on a 0.12 s proof. r04 default ("idle"): never on the request path
// -- zkpoa_idle_work builds one table per call when the library has nothing else to do (the `prover` server calls it
// after 300 ms without a request) -- except for a key that has served ZKPOA_PRECOMP_AFTER proofs (default 64) in a host
// that never calls it. "eager" = the r03 behaviour, "0" = never.
enum PrecompPolicy { kPrecompOff, kPrecompEager, kPrecompIdle };
PrecompPolicy precomp_policy() {
  const char* e = getenv("ZKPOA_PRECOMP");
  if (!e || !*e) return kPrecompIdle;
  if (!strcmp(e, "0") || !strcmp(e, "off")) return kPrecompOff;
  if (!strcmp(e, "eager")) return kPrecompEager;
  return kPrecompIdle;
}
uint64_t precomp_after() {
  const char* e = getenv("ZKPOA_PRECOMP_AFTER");
  const long v = e && *e ? atol(e) : 64;
  return v < 1 ? 1 : (uint64_t)v;
}
// must this request build the key's tables before it proves? (done = proofs the key has served, settled / bytes = its tables)
bool tables_due_now(uint64_t done, bool settled, uint64_t bytes) {
  switch (precomp_policy()) {
    case kPrecompEager: return done == 1 && bytes == 0;
    case kPrecompIdle: return done >= precomp_after() && !settled && bytes == 0;
    default: return false;
  }
}

struct CachedKey {
  dev_t dev;
  ino_t ino;
  off_t size;
  struct timespec mtime;
  zkpoa_zkey* zk;
  uint64_t last_use;
};
std::vector<CachedKey> g_key_cache;
uint64_t g_key_clock = 0;

size_t key_cache_capacity() {
  const char* e = getenv("ZKPOA_KEY_CACHE");
  if (!e || !*e) return 2;
  long v = atol(e);
  return v < 0 ? 0 : (size_t)v;
}

void key_cache_drop(size_t idx) {
  (void)hipDeviceSynchronize();
  g_key_cache[idx].zk->release();
  delete g_key_cache[idx].zk;
  g_key_cache.erase(g_key_cache.begin() + (long)idx);
}

void key_cache_clear() {
  while (!g_key_cache.empty()) key_cache_drop(g_key_cache.size() - 1);
}

// The same cache for keys sharded over the ranks of a multi-GPU process (one entry = G shard handles + their exchange
// buffers); the second use of a key builds every shard's fixed-base tables, in parallel on the G devices.
struct CachedMultiKey {
  dev_t dev;
  ino_t ino;
  off_t size;
  struct timespec mtime;
  MultiKey* mk;
  uint64_t last_use;
};
std::vector<CachedMultiKey> g_multi_cache;

int multi_file_prove(DeviceSet* ds, int fd, const struct stat& sb, const char* path, const WtnsSrc& wsrc,
                     char* proof_buffer, unsigned long* proof_size, char* public_buffer, unsigned long* public_size,
                     char* error_msg, unsigned long error_msg_maxsize) {
  int rc = PROVER_OK;
  MultiKey* mk = nullptr;
  bool cached = false, hit = false;
  try {
    const size_t cap = key_cache_capacity();
    for (auto& c : g_multi_cache)
      if (c.dev == sb.st_dev && c.ino == sb.st_ino && c.size == sb.st_size && c.mtime.tv_sec == sb.st_mtim.tv_sec &&
          c.mtime.tv_nsec == sb.st_mtim.tv_nsec) {
        mk = c.mk;
        c.last_use = ++g_key_clock;
        hit = cached = true;
      }
    auto drop = [&](size_t idx) {
      multi_key_release(ds, g_multi_cache[idx].mk);
      g_multi_cache.erase(g_multi_cache.begin() + (long)idx);
    };
    if (!mk) {
      void* map = mmap(nullptr, (size_t)sb.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
      if (map == MAP_FAILED) throw ProverError(PROVER_ERROR, std::string("cannot mmap zkey file ") + path);
      try {
        while (cap && g_multi_cache.size() >= cap) {
          size_t lru = 0;
          for (size_t i = 1; i < g_multi_cache.size(); i++)
            if (g_multi_cache[i].last_use < g_multi_cache[lru].last_use) lru = i;
          drop(lru);
        }
        try {
          mk = multi_key_load(ds, reinterpret_cast<const uint8_t*>(map), (uint64_t)sb.st_size);
        } catch (const HipError&) {
          if (g_multi_cache.empty()) throw;
          while (!g_multi_cache.empty()) drop(g_multi_cache.size() - 1);   // probably out of HBM: retry alone
          for (int d : ds->ids) {
            (void)hipSetDevice(d);
            (void)hipGetLastError();
          }
          mk = multi_key_load(ds, reinterpret_cast<const uint8_t*>(map), (uint64_t)sb.st_size);
        }
      } catch (...) {
        munmap(map, (size_t)sb.st_size);
        throw;
      }
      munmap(map, (size_t)sb.st_size);
      if (cap) {
        g_multi_cache.push_back({sb.st_dev, sb.st_ino, sb.st_size, sb.st_mtim, mk, ++g_key_clock});
        cached = true;
      }
    }
    if (hit && tables_due_now(mk->proofs_done, mk->tables_tried, mk->table_bytes)) {
      {
        auto tp0 = std::chrono::steady_clock::now();
        multi_precompute(ds, mk);
        mk->tables_tried = true;
        if (req_getenv("ZKPOA_VERBOSE"))
          fprintf(stderr, "zkpoa: fixed-base tables for the cached key on %zu ranks: %.2f GB in %.0f ms\n", ds->ids.size(),
                  mk->table_bytes / 1e9,
                  std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tp0).count());
      }
    }
    rc = multi_prove_to_json(ds, mk, wsrc, proof_buffer, proof_size, public_buffer, public_size, error_msg,
                             error_msg_maxsize, (uint64_t)sb.st_size, hit);
    mk->proofs_done++;
  } catch (const ProverError& e) {
    set_err(error_msg, error_msg_maxsize, e.what());
    rc = e.code;
  } catch (const HipError& e) {
    set_err(error_msg, error_msg_maxsize, e.what());
    rc = PROVER_ERROR_RUNTIME;
    // the shards of a key whose proof died in the HIP runtime are not kept for the next request
    for (size_t i = 0; i < g_multi_cache.size(); i++)
      if (g_multi_cache[i].mk == mk) {
        g_multi_cache.erase(g_multi_cache.begin() + (long)i);
        cached = false;
        break;
      }
  } catch (const std::bad_alloc&) {
    set_err(error_msg, error_msg_maxsize, "out of host memory");
    rc = PROVER_ERROR_RUNTIME;
  } catch (const std::system_error& e) {
    set_err(error_msg, error_msg_maxsize, e.what());
    rc = PROVER_ERROR_RUNTIME;
  } catch (const std::exception& e) {
    set_err(error_msg, error_msg_maxsize, e.what());
    rc = PROVER_ERROR;
  }
  if (mk && !cached) multi_key_release(ds, mk);
  return rc;
}

// A request on a resident key in steady state (cached, tables built or not wanted): the witness goes into a free
// staging buffer under the stage lock -- which is then dropped -- and the proof runs under the prove lock. Enters with
// stage_lk held, leaves with it released.
int staged_prove(zkpoa_context* ctx, const zkpoa_zkey* zk, const WtnsSrc& wsrc, uint64_t zkey_size,
                 std::unique_lock<std::mutex>& stage_lk, hipStream_t cs, char* proof_buffer, unsigned long* proof_size,
                 char* public_buffer, unsigned long* public_size, char* error_msg, unsigned long error_msg_maxsize) {
  int rc = PROVER_OK, slot = -1;
  try {
    ZK_HIP(hipSetDevice(ctx->dev.device));
    WtnsView w = parse_wtns(wsrc);
    if (w.n != zk->nVars)
      throw ProverError(PROVER_INVALID_WITNESS_LENGTH, "Invalid witness length. Circuit: " + std::to_string(zk->nVars) +
                                                           ", witness: " + std::to_string(w.n));
    if (!zk->wbuf[0]) zk->wbuf[0] = zk->d_witness;   // adopt the buffer the key came with
    g_stage_cv.wait(stage_lk, [&] { return !zk->wbusy[0] || !zk->wbusy[1]; });
    slot = !zk->wbusy[0] ? 0 : 1;
    if (!zk->wbuf[slot]) ZK_HIP(hipMalloc(&zk->wbuf[slot], (size_t)zk->nVars * 32));
    zk->wbusy[slot] = true;
    g_staged_users++;
    const auto tu = std::chrono::steady_clock::now();
    w.upload(ctx, zk->wbuf[slot], 0, w.n, cs);      // on the copy stream(s): lane 0 may be busy with another proof's chain
    const float up_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - tu).count();
    std::vector<uint8_t> pub_store;
    const uint8_t* pubs = w.publics(zk->nPublic, pub_store);
    stage_lk.unlock();
    {
      std::lock_guard<std::mutex> lk(g_prove_mutex);
      zk->d_witness = zk->wbuf[slot];
      uint8_t rb[32], sb[32], pts[256];
      const uint8_t *rp = nullptr, *sp = nullptr;
      env_blinding(rb, sb, rp, sp);
      prove_core(ctx, zk, rp, sp, pts);
      selfcheck(ctx, zk, pts, pubs);
      ctx->io_ms[0] = up_ms;
      ctx->io_ms[1] = (float)((double)w.n * 32 / 1e6);
      rc = emit_outputs(ctx, zk, pts, pubs, proof_buffer, proof_size, public_buffer, public_size, error_msg,
                        error_msg_maxsize, 0.0, zkey_size, "cached,");
      const_cast<zkpoa_zkey*>(zk)->proofs_done++;
    }
  } catch (...) {
    rc = classify_current_exception(error_msg, error_msg_maxsize);
  }
  if (!stage_lk.owns_lock()) stage_lk.lock();
  if (slot >= 0) {
    zk->wbusy[slot] = false;
    g_staged_users--;
  }
  stage_lk.unlock();
  g_stage_cv.notify_all();
  return rc;
}

int zkey_file_prove(const char* path, const WtnsSrc& wsrc, char* proof_buffer,
                    unsigned long* proof_size, char* public_buffer, unsigned long* public_size, char* error_msg,
                    unsigned long error_msg_maxsize) {
  int fd = open(path, O_RDONLY);
  if (fd < 0) {
    set_err(error_msg, error_msg_maxsize, std::string("cannot open zkey file ") + path);
    return PROVER_ERROR;
  }
  struct stat sb;
  if (fstat(fd, &sb) != 0 || sb.st_size == 0) {
    close(fd);
    set_err(error_msg, error_msg_maxsize, std::string("cannot stat zkey file ") + path);
    return PROVER_ERROR;
  }
  std::string err;
  DeviceSet* ds = nullptr;
  int dcode = PROVER_ERROR;
  try {
    uint32_t power = 0;
    if (!devices_ready()) {   // the first key of the process decides the device list: its domain size is in the header
      void* map = mmap(nullptr, (size_t)sb.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
      if (map == MAP_FAILED) throw ProverError(PROVER_ERROR, std::string("cannot mmap zkey file ") + path);
      try {
        power = zkey_power(reinterpret_cast<const uint8_t*>(map), (uint64_t)sb.st_size);
      } catch (...) {
        munmap(map, (size_t)sb.st_size);
        throw;
      }
      munmap(map, (size_t)sb.st_size);
    }
    ds = process_devices(power, err, &dcode);
  } catch (const std::exception& e) {   // malformed key: nothing touches a GPU
    close(fd);
    set_err(error_msg, error_msg_maxsize, e.what());
    return PROVER_ERROR;
  }
  if (!ds) {
    close(fd);
    set_err(error_msg, error_msg_maxsize, err);
    return dcode;
  }
  zkpoa_context* ctx = ds->ctx[0];
  std::unique_lock<std::mutex> stage_lk(g_stage_mutex);
  if (ds->ids.size() == 1) {   // steady state of a resident single-GPU key: stage the witness, then prove (two locks)
    for (auto& c : g_key_cache)
      if (c.dev == sb.st_dev && c.ino == sb.st_ino && c.size == sb.st_size && c.mtime.tv_sec == sb.st_mtim.tv_sec &&
          c.mtime.tv_nsec == sb.st_mtim.tv_nsec) {
        const uint64_t done = c.zk->proofs_done.load();
        const bool tables_due = tables_due_now(done, c.zk->tables_settled, c.zk->table_bytes);
        hipStream_t cs = ctx->dev.copy_stream_wait();
        if (done >= 1 && !tables_due && cs) {
          c.last_use = ++g_key_clock;
          close(fd);
          return staged_prove(ctx, c.zk, wsrc, (uint64_t)sb.st_size, stage_lk, cs, proof_buffer, proof_size, public_buffer,
                              public_size, error_msg, error_msg_maxsize);
        }
      }
  }
  // everything else changes the cache or a key: alone, with both locks
  g_stage_cv.wait(stage_lk, [] { return g_staged_users == 0; });
  std::lock_guard<std::mutex> lk(g_prove_mutex);
  if (ds->ids.size() > 1) {
    int rc = multi_file_prove(ds, fd, sb, path, wsrc, proof_buffer, proof_size, public_buffer, public_size,
                              error_msg, error_msg_maxsize);
    close(fd);
    return rc;
  }
  int rc = PROVER_OK;
  zkpoa_zkey* zk = nullptr;
  bool owned = false, hit = false, proved = false;
  double load_ms = 0;
  try {
    ZK_HIP(hipSetDevice(ctx->dev.device));
    const size_t cap = key_cache_capacity();
    for (auto& c : g_key_cache)
      if (c.dev == sb.st_dev && c.ino == sb.st_ino && c.size == sb.st_size && c.mtime.tv_sec == sb.st_mtim.tv_sec &&
          c.mtime.tv_nsec == sb.st_mtim.tv_nsec) {
        zk = c.zk;
        c.last_use = ++g_key_clock;
        hit = true;
      }
    if (zk && zk->wbuf[0]) zk->d_witness = zk->wbuf[0];   // nothing is staged now: back to the first buffer
    if (!zk) {
      void* map = mmap(nullptr, (size_t)sb.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
      if (map == MAP_FAILED) throw ProverError(PROVER_ERROR, std::string("cannot mmap zkey file ") + path);
      auto tl0 = std::chrono::steady_clock::now();
      try {
        while (cap && g_key_cache.size() >= cap) {   // make room first: least recently used
          size_t lru = 0;
          for (size_t i = 1; i < g_key_cache.size(); i++)
            if (g_key_cache[i].last_use < g_key_cache[lru].last_use) lru = i;
          key_cache_drop(lru);
        }
        // load and prove in one overlapped phase (the mapping must outlive it: the uploader streams from it)
        try {
          rc = load_and_prove_to_json(ctx, reinterpret_cast<const uint8_t*>(map), (uint64_t)sb.st_size, wsrc,
                                      proof_buffer, proof_size, public_buffer, public_size, error_msg, error_msg_maxsize,
                                      &zk, fd);
        } catch (const HipError&) {
          if (g_key_cache.empty()) throw;
          key_cache_clear();                          // probably out of HBM: retry with nothing else resident
          (void)hipGetLastError();
          rc = load_and_prove_to_json(ctx, reinterpret_cast<const uint8_t*>(map), (uint64_t)sb.st_size, wsrc,
                                      proof_buffer, proof_size, public_buffer, public_size, error_msg, error_msg_maxsize,
                                      &zk, fd);
        }
        proved = true;
      } catch (...) {
        munmap(map, (size_t)sb.st_size);
        if (zk) {   // loaded, but the proof failed afterwards (self-check, output): not cached, not leaked
          (void)hipDeviceSynchronize();
          zk->release();
          delete zk;
          zk = nullptr;
        }
        throw;
      }
      munmap(map, (size_t)sb.st_size);
      load_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tl0).count();
      if (cap) g_key_cache.push_back({sb.st_dev, sb.st_ino, sb.st_size, sb.st_mtim, zk, ++g_key_clock});
      else owned = true;
    }
    // a key that comes out of the cache is being reused: build its fixed-base tables now, once (ZKPOA_PRECOMP=0 off)
    if (hit && !owned && tables_due_now(zk->proofs_done, zk->tables_settled, zk->table_bytes)) {
      {
        auto tp0 = std::chrono::steady_clock::now();
        try {
          uint64_t used = zkey_precompute(ctx, zk, 0);
          if (req_getenv("ZKPOA_VERBOSE"))
            fprintf(stderr, "zkpoa: fixed-base tables for the cached key: %.2f GB in %.0f ms\n", used / 1e9,
                    std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tp0).count());
        } catch (const HipError&) {   // out of HBM: the classic form keeps working
          zk->release_tables();
          (void)hipGetLastError();
        }
        zk->tables_settled = true;
      }
    }
    if (!proved)
      rc = prove_to_json(ctx, zk, wsrc, proof_buffer, proof_size, public_buffer, public_size, error_msg,
                         error_msg_maxsize, load_ms, (uint64_t)sb.st_size, hit);
    zk->proofs_done++;
  } catch (...) {
    rc = classify_current_exception(error_msg, error_msg_maxsize);
  }
  close(fd);
  if (owned && zk) {
    zk->release();
    delete zk;
  }
  return rc;
}

}  // namespace

// ---- C ABI -------------------------------------------------------------------------------------------
#define ZK_PROVER_CATCH(ctx)          \
  catch (const ProverError& e) {      \
    (ctx)->last_error = e.what();     \
    return e.code;                    \
  }                                   \
  catch (const std::exception& e) {   \
    (ctx)->last_error = e.what();     \
    return PROVER_ERROR;              \
  }

extern "C" int zkpoa_zkey_load(zkpoa_context* ctx, const void* zkey_buffer, unsigned long zkey_size, zkpoa_zkey** out) {
  if (!ctx || !out || !zkey_buffer) return PROVER_ERROR;
  *out = nullptr;
  try {
    ZK_HIP(hipSetDevice(ctx->dev.device));
    *out = zkey_load_impl(ctx, reinterpret_cast<const uint8_t*>(zkey_buffer), zkey_size);
  }
  ZK_PROVER_CATCH(ctx)
  return PROVER_OK;
}

extern "C" int zkpoa_zkey_load_shard(zkpoa_context* ctx, const void* zkey_buffer, unsigned long zkey_size,
                                     uint64_t rank, uint64_t world, zkpoa_zkey** out) {
  if (!ctx || !out || !zkey_buffer) return PROVER_ERROR;
  *out = nullptr;
  try {
    ZK_HIP(hipSetDevice(ctx->dev.device));
    *out = zkey_load_impl(ctx, reinterpret_cast<const uint8_t*>(zkey_buffer), zkey_size, rank, world);
  }
  ZK_PROVER_CATCH(ctx)
  return PROVER_OK;
}

extern "C" int zkpoa_zkey_set_shard(zkpoa_zkey* zkey, uint64_t rank, uint64_t world) {
  if (!zkey || world == 0 || rank >= world) return PROVER_ERROR;
  if (zkey->wbase || zkey->cbase || zkey->hbase || zkey->csr_local || !zkey->dH || zkey->bc_log)
    return PROVER_ERROR;  // only a fully resident key can be re-sharded
  zkey->set_shard(rank, world);
  zkey->split_world = zkey->split_rank = zkey->split_log = 0;
  zkey->h_ready = false;
  try {
    queries_set_slice(zkey);
  } catch (const std::exception&) {
    return PROVER_ERROR;
  }
  return PROVER_OK;
}

extern "C" int zkpoa_zkey_load_shard_split(zkpoa_context* ctx, const void* zkey_buffer, unsigned long zkey_size,
                                           uint64_t rank, uint64_t world, zkpoa_zkey** out) {
  if (!ctx || !out || !zkey_buffer) return PROVER_ERROR;
  *out = nullptr;
  try {
    ZK_HIP(hipSetDevice(ctx->dev.device));
    *out = zkey_load_impl(ctx, reinterpret_cast<const uint8_t*>(zkey_buffer), zkey_size, rank, world, true);
  }
  ZK_PROVER_CATCH(ctx)
  return PROVER_OK;
}

extern "C" int zkpoa_zkey_load_shard_ex(zkpoa_context* ctx, const void* zkey_buffer, unsigned long zkey_size,
                                        uint64_t rank, uint64_t world, int flags, zkpoa_zkey** out) {
  if (!ctx || !out || !zkey_buffer) return PROVER_ERROR;
  *out = nullptr;
  try {
    ZK_HIP(hipSetDevice(ctx->dev.device));
    *out = zkey_load_impl(ctx, reinterpret_cast<const uint8_t*>(zkey_buffer), zkey_size, rank, world,
                          (flags & ZKPOA_SHARD_SPLIT_CHAIN) != 0, ZKPOA_SHARD_BLOCK_LOG(flags));
  }
  ZK_PROVER_CATCH(ctx)
  return PROVER_OK;
}

extern "C" int zkpoa_zkey_set_shard_split(zkpoa_context* ctx, zkpoa_zkey* zkey, uint64_t rank, uint64_t world) {
  if (!ctx || !zkey) return PROVER_ERROR;
  try {
    ZK_HIP(hipSetDevice(ctx->dev.device));
    if (zkey->wbase || zkey->cbase || zkey->hbase || zkey->csr_local || !zkey->dH || zkey->bc_log)
      throw ProverError(PROVER_ERROR, "only a fully resident key can be re-sharded");
    check_split(zkey, rank, world);
    hipStream_t st = ctx->dev.lanes[0].stream;
    const uint64_t cnt = zkey->domain / world;
    if (zkey->tH_cyclic) {   // built from the cyclic shard that is replaced now
      uint64_t info[4];
      msm_table_info(zkey->tH, info);
      zkey->table_bytes -= info[3] < zkey->table_bytes ? info[3] : zkey->table_bytes;
      msm_table_release(zkey->tH);
      zkey->tH = nullptr;
      zkey->tH_cyclic = false;
    }
    if (zkey->dHs) {
      ZK_HIP(hipFree(zkey->dHs));
      zkey->dHs = nullptr;
    }
    ZK_HIP(hipMalloc(&zkey->dHs, cnt * 64));
    hipLaunchKernelGGL(strided_copy64_kernel, dim3((uint32_t)((cnt * 4 + 255) / 256)), dim3(256), 0, st,
                       (const uint4*)zkey->dH, (uint4*)zkey->dHs, cnt, (uint32_t)rank, (uint32_t)world);
    zkey->set_shard(rank, world);
    queries_set_slice(zkey);
    set_split(zkey, rank, world);
    ntt_prepare(ctx, st, zkey->power - zkey->split_log);
    ZK_HIP(hipStreamSynchronize(st));
    ZK_HIP(hipGetLastError());
  }
  ZK_PROVER_CATCH(ctx)
  return PROVER_OK;
}

extern "C" int zkpoa_witness_load(zkpoa_context* ctx, const zkpoa_zkey* zkey, const void* wtns_buffer,
                                  unsigned long wtns_size, uint8_t* public_le, unsigned long public_capacity) {
  if (!ctx || !zkey || !wtns_buffer) return PROVER_ERROR;
  try {
    ZK_HIP(hipSetDevice(ctx->dev.device));
    WtnsView w = parse_wtns(reinterpret_cast<const uint8_t*>(wtns_buffer), wtns_size);
    if (w.n != zkey->nVars)
      throw ProverError(PROVER_INVALID_WITNESS_LENGTH, "Invalid witness length. Circuit: " +
                                                           std::to_string(zkey->nVars) + ", witness: " + std::to_string(w.n));
    if (public_le && public_capacity < (unsigned long)zkey->nPublic * 32)
      throw ProverError(PROVER_ERROR_SHORT_BUFFER, "public buffer too small");
    ctx->uploader.upload(zkey->d_witness, w.values, (size_t)w.n * 32, ctx->dev.device, ctx->dev.lanes[0].stream);
    zkey->h_ready = false;
    if (public_le) memcpy(public_le, w.values + 32, (size_t)zkey->nPublic * 32);
  }
  ZK_PROVER_CATCH(ctx)
  return PROVER_OK;
}

extern "C" int zkpoa_split_stage1(zkpoa_context* ctx, const zkpoa_zkey* zkey, const void* d_witness, void* d_exchange) {
  if (!ctx || !zkey || !d_exchange) return PROVER_ERROR;
  try {
    ZK_HIP(hipSetDevice(ctx->dev.device));
    if (zkey->split_world < 2) throw ProverError(PROVER_ERROR, "split chain: the key handle is not a split shard");
    hipStream_t st = ctx->dev.lanes[0].stream;
    if (d_witness && d_witness != zkey->d_witness) {
      ZK_HIP(hipMemcpyAsync(zkey->d_witness, d_witness, (size_t)zkey->nVars * 32, hipMemcpyDeviceToDevice, st));
      if (!ctx->ev_witness) ZK_HIP(hipEventCreateWithFlags(&ctx->ev_witness, hipEventDisableTiming));
      ZK_HIP(hipEventRecord(ctx->ev_witness, st));   // the witness MSMs run on other lanes: they wait for this copy
      ctx->ev_witness_set = true;
    }
    split_stage1(ctx, zkey, d_exchange);
  }
  ZK_PROVER_CATCH(ctx)
  return PROVER_OK;
}

extern "C" int zkpoa_split_stage2(zkpoa_context* ctx, const zkpoa_zkey* zkey, const void* d_received, void* d_exchange) {
  if (!ctx || !zkey || !d_received || !d_exchange || d_received == d_exchange) return PROVER_ERROR;
  try {
    ZK_HIP(hipSetDevice(ctx->dev.device));
    if (zkey->split_world < 2) throw ProverError(PROVER_ERROR, "split chain: the key handle is not a split shard");
    split_stage2(ctx, zkey, d_received, d_exchange);
  }
  ZK_PROVER_CATCH(ctx)
  return PROVER_OK;
}

extern "C" int zkpoa_split_stage3(zkpoa_context* ctx, const zkpoa_zkey* zkey, void* d_received) {
  if (!ctx || !zkey || !d_received) return PROVER_ERROR;
  try {
    ZK_HIP(hipSetDevice(ctx->dev.device));
    if (zkey->split_world < 2) throw ProverError(PROVER_ERROR, "split chain: the key handle is not a split shard");
    split_stage3(ctx, zkey, d_received);
  }
  ZK_PROVER_CATCH(ctx)
  return PROVER_OK;
}

extern "C" void* zkpoa_context_stream(zkpoa_context* ctx, int lane) {
  if (!ctx || lane < 0 || lane >= DeviceCtx::kLanes) return nullptr;
  try {
    if (lane) ctx->dev.wait_lanes();
    ctx->dev.ensure_lane(lane);   // lanes beyond the eager ones are created on first use: never hand out a null stream
  } catch (const std::exception&) {
    return nullptr;
  }
  return reinterpret_cast<void*>(ctx->dev.lanes[lane].stream);
}

extern "C" int zkpoa_context_synchronize(zkpoa_context* ctx) {
  if (!ctx) return PROVER_ERROR;
  try {
    ZK_HIP(hipSetDevice(ctx->dev.device));
    ZK_HIP(hipStreamSynchronize(ctx->dev.lanes[0].stream));
    ZK_HIP(hipGetLastError());
  }
  ZK_PROVER_CATCH(ctx)
  return PROVER_OK;
}

extern "C" int zkpoa_zkey_header(const zkpoa_zkey* zkey, uint8_t header_points[448]) {
  if (!zkey || !header_points) return PROVER_ERROR;
  zkey_header_bytes(zkey, header_points);
  return PROVER_OK;
}

extern "C" int zkpoa_prove_partials(zkpoa_context* ctx, const zkpoa_zkey* zkey, const void* wtns_buffer,
                                    unsigned long wtns_size, uint8_t partials[384], uint8_t* public_le,
                                    unsigned long public_capacity) {
  if (!ctx || !zkey || !wtns_buffer || !partials) return PROVER_ERROR;
  try {
    ZK_HIP(hipSetDevice(ctx->dev.device));
    WtnsView w = parse_wtns(reinterpret_cast<const uint8_t*>(wtns_buffer), wtns_size);
    if (w.n != zkey->nVars)
      throw ProverError(PROVER_INVALID_WITNESS_LENGTH, "Invalid witness length. Circuit: " +
                                                           std::to_string(zkey->nVars) + ", witness: " + std::to_string(w.n));
    if (public_le && public_capacity < (unsigned long)zkey->nPublic * 32)
      throw ProverError(PROVER_ERROR_SHORT_BUFFER, "public buffer too small");
    if (zkey->split_world > 1)
      throw ProverError(PROVER_ERROR, "split chain: use zkpoa_witness_load, zkpoa_split_stage1/2/3, then "
                                      "zkpoa_prove_partials_device with a NULL witness");
    ctx->uploader.upload(zkey->d_witness, w.values, (size_t)w.n * 32, ctx->dev.device, ctx->dev.lanes[0].stream);
    prove_partials(ctx, zkey, partials);
    if (public_le) memcpy(public_le, w.values + 32, (size_t)zkey->nPublic * 32);
  }
  ZK_PROVER_CATCH(ctx)
  return PROVER_OK;
}

extern "C" int zkpoa_prove_partials_device(zkpoa_context* ctx, const zkpoa_zkey* zkey, const void* d_witness,
                                           uint8_t partials[384]) {
  if (!ctx || !zkey || !partials) return PROVER_ERROR;
  try {
    ZK_HIP(hipSetDevice(ctx->dev.device));
    hipStream_t st = ctx->dev.lanes[0].stream;
    if (d_witness && d_witness != zkey->d_witness) {
      ZK_HIP(hipMemcpyAsync(zkey->d_witness, d_witness, (size_t)zkey->nVars * 32, hipMemcpyDeviceToDevice, st));
      ZK_HIP(hipStreamSynchronize(st));
    }
    prove_partials(ctx, zkey, partials);
  }
  ZK_PROVER_CATCH(ctx)
  return PROVER_OK;
}

extern "C" int zkpoa_prove_assemble(const uint8_t header_points[448], const uint8_t partial_sums[384],
                                    const uint8_t* r_le, const uint8_t* s_le, uint8_t proof_points[256]) {
  if (!header_points || !partial_sums || !proof_points) return PROVER_ERROR;
  try {
    prove_assemble(header_points, partial_sums, r_le, s_le, proof_points);
  } catch (const std::exception&) {
    return PROVER_ERROR;
  }
  return PROVER_OK;
}

extern "C" void zkpoa_zkey_free(zkpoa_context* ctx, zkpoa_zkey* zkey) {
  if (!zkey) return;
  if (ctx) {
    (void)hipSetDevice(ctx->dev.device);
    (void)hipDeviceSynchronize();
  }
  zkey->release();
  delete zkey;
}

extern "C" int zkpoa_zkey_info(const zkpoa_zkey* zkey, uint64_t out[4]) {
  if (!zkey || !out) return PROVER_ERROR;
  out[0] = zkey->nVars;
  out[1] = zkey->nPublic;
  out[2] = zkey->domain;
  out[3] = zkey->nCoefs;
  return PROVER_OK;
}

extern "C" int zkpoa_prove(zkpoa_context* ctx, const zkpoa_zkey* zkey, const void* wtns_buffer, unsigned long wtns_size,
                           const uint8_t* r_le, const uint8_t* s_le, uint8_t proof_points[256], uint8_t* public_le,
                           unsigned long public_capacity) {
  if (!ctx || !zkey || !wtns_buffer || !proof_points) return PROVER_ERROR;
  try {
    ZK_HIP(hipSetDevice(ctx->dev.device));
    uint8_t dummy[1];
    prove_impl(ctx, zkey, WtnsSrc{reinterpret_cast<const uint8_t*>(wtns_buffer), wtns_size, -1}, r_le, s_le, proof_points,
               public_le ? public_le : dummy, public_le ? public_capacity : (zkey->nPublic ? 0 : 1));
  }
  ZK_PROVER_CATCH(ctx)
  return PROVER_OK;
}

// Key (or one rank's shard of it) from sections already in HBM. world == 1: the whole key. world > 1: the point
// buffers hold only this rank's index ranges (zkpoa_zkey::split: the ranges zkpoa_zkey_load_shard uploads); with
// `split`, d_H is the cyclic shard H[t * world + rank] and the records are those of the constraints
// c = rank (mod world) (records of other constraints are ignored).
static zkpoa_zkey* zkey_load_device_impl(zkpoa_context* ctx, uint64_t n_vars, uint64_t n_public, unsigned log_domain,
                                         uint64_t rank, uint64_t world, bool split, uint32_t bc_log, const void* d_A, const void* d_B1,
                                         const void* d_B2, const void* d_C, const void* d_H,
                                         const void* d_coef_records, uint64_t n_coefs, const uint8_t header_points[448]) {
  std::unique_ptr<zkpoa_zkey> zk(new zkpoa_zkey());
  try {
    ZK_HIP(hipSetDevice(ctx->dev.device));
    if (log_domain > 28 || n_vars == 0 || n_vars > (1ull << 28) || n_public + 1 > n_vars || n_coefs > 0xffffffffull)
      throw ProverError(PROVER_ERROR, "zkey_load_device: size out of range");
    if (world == 0 || rank >= world) throw ProverError(PROVER_ERROR, "zkey shard: rank/world out of range");
    zk->nVars = (uint32_t)n_vars;
    zk->nPublic = (uint32_t)n_public;
    zk->power = log_domain;
    zk->domain = 1u << log_domain;
    zk->nCoefs = n_coefs;
    zk->owns_points = false;
    zk->dA = const_cast<void*>(d_A);
    zk->dB1 = const_cast<void*>(d_B1);
    zk->dB2 = const_cast<void*>(d_B2);
    zk->dC = const_cast<void*>(d_C);
    zk->alpha1 = h_affine_from_bytes<HFq>(header_points);
    zk->beta1 = h_affine_from_bytes<HFq>(header_points + 64);
    zk->beta2 = h_affine_from_bytes<HFq2>(header_points + 128);
    zk->delta1 = h_affine_from_bytes<HFq>(header_points + 256);
    zk->delta2 = h_affine_from_bytes<HFq2>(header_points + 320);
    zk->set_shard(rank, world);   // world == 1: the whole key
    zk->wbase = zk->wlo;
    zk->cbase = zk->clo;
    zk->hbase = zk->hlo;
    if (split) {
      check_split(zk.get(), rank, world);
      set_split(zk.get(), rank, world);
      // the handle owns its cyclic H shard (release() frees it): a copy of the caller's, device to device
      const size_t bytes = (size_t)(zk->domain / world) * 64;
      ZK_HIP(hipMalloc(&zk->dHs, bytes));
      ZK_HIP(hipMemcpy(zk->dHs, d_H, bytes, hipMemcpyDeviceToDevice));
      zk->hlo = zk->hbase = 0;
      zk->hcnt = 0;   // no contiguous H range on this handle
    } else {
      zk->dH = const_cast<void*>(d_H);
    }
    set_block_cyclic(zk.get(), rank, world, bc_log);
    if (zk->bc_log) ZK_HIP(hipMalloc(&zk->d_cscal, zk->ccnt ? zk->ccnt * 32 : 1));
    queries_compact(ctx, zk.get(), zk->wcnt);
    build_csr(ctx, zk.get(), d_coef_records, split);
    ntt_prepare(ctx, ctx->dev.lanes[0].stream, zk->power);
    if (split) ntt_prepare(ctx, ctx->dev.lanes[0].stream, zk->power - zk->split_log);
    ZK_HIP(hipStreamSynchronize(ctx->dev.lanes[0].stream));
  } catch (...) {
    zk->release();
    throw;
  }
  return zk.release();
}

extern "C" int zkpoa_zkey_load_device(zkpoa_context* ctx, uint64_t n_vars, uint64_t n_public, unsigned log_domain,
                                      const void* d_A, const void* d_B1, const void* d_B2, const void* d_C,
                                      const void* d_H, const void* d_coef_records, uint64_t n_coefs,
                                      const uint8_t header_points[448], zkpoa_zkey** out) {
  if (!ctx || !out || !header_points) return PROVER_ERROR;
  *out = nullptr;
  try {
    ZK_HIP(hipSetDevice(ctx->dev.device));
    *out = zkey_load_device_impl(ctx, n_vars, n_public, log_domain, 0, 1, false, 0, d_A, d_B1, d_B2, d_C, d_H,
                                 d_coef_records, n_coefs, header_points);
  }
  ZK_PROVER_CATCH(ctx)
  return PROVER_OK;
}

extern "C" int zkpoa_zkey_load_device_shard(zkpoa_context* ctx, uint64_t n_vars, uint64_t n_public, unsigned log_domain,
                                            uint64_t rank, uint64_t world, int split, const void* d_A,
                                            const void* d_B1, const void* d_B2, const void* d_C, const void* d_H,
                                            const void* d_coef_records, uint64_t n_coefs,
                                            const uint8_t header_points[448], zkpoa_zkey** out) {
  if (!ctx || !out || !header_points) return PROVER_ERROR;
  *out = nullptr;
  try {
    ZK_HIP(hipSetDevice(ctx->dev.device));
    *out = zkey_load_device_impl(ctx, n_vars, n_public, log_domain, rank, world, (split & ZKPOA_SHARD_SPLIT_CHAIN) != 0,
                                 ZKPOA_SHARD_BLOCK_LOG(split), d_A, d_B1, d_B2, d_C, d_H,
                                 d_coef_records, n_coefs, header_points);
  }
  ZK_PROVER_CATCH(ctx)
  return PROVER_OK;
}

extern "C" int zkpoa_prove_device(zkpoa_context* ctx, const zkpoa_zkey* zkey, const void* d_witness,
                                  const uint8_t* r_le, const uint8_t* s_le, uint8_t proof_points[256],
                                  uint8_t* public_le, unsigned long public_capacity) {
  if (!ctx || !zkey || !d_witness || !proof_points) return PROVER_ERROR;
  try {
    ZK_HIP(hipSetDevice(ctx->dev.device));
    if (public_le && public_capacity < (unsigned long)zkey->nPublic * 32)
      throw ProverError(PROVER_ERROR_SHORT_BUFFER, "public buffer too small");
    hipStream_t st = ctx->dev.lanes[0].stream;
    if (d_witness != zkey->d_witness)
      ZK_HIP(hipMemcpyAsync(zkey->d_witness, d_witness, (size_t)zkey->nVars * 32, hipMemcpyDeviceToDevice, st));
    if (public_le && zkey->nPublic)
      ZK_HIP(hipMemcpyAsync(public_le, reinterpret_cast<const char*>(d_witness) + 32, (size_t)zkey->nPublic * 32,
                            hipMemcpyDeviceToHost, st));
    ZK_HIP(hipStreamSynchronize(st));
    prove_core(ctx, zkey, r_le, s_le, proof_points);
    if (!zkey->vkey_points.empty() && selfcheck_mode()) {
      std::vector<uint8_t> pub((size_t)zkey->nPublic * 32 + 1);
      if (zkey->nPublic)
        ZK_HIP(hipMemcpy(pub.data(), reinterpret_cast<const char*>(zkey->d_witness) + 32, (size_t)zkey->nPublic * 32,
                         hipMemcpyDeviceToHost));
      selfcheck(ctx, zkey, proof_points, pub.data());
    }
  }
  ZK_PROVER_CATCH(ctx)
  return PROVER_OK;
}

extern "C" int zkpoa_zkey_precompute(zkpoa_context* ctx, zkpoa_zkey* zkey, uint64_t budget_bytes, uint64_t* used_bytes) {
  if (!ctx || !zkey) return PROVER_ERROR;
  try {
    ZK_HIP(hipSetDevice(ctx->dev.device));
    uint64_t used = zkey_precompute(ctx, zkey, budget_bytes);
    if (used_bytes) *used_bytes = used;
  }
  ZK_PROVER_CATCH(ctx)
  return PROVER_OK;
}

extern "C" int zkpoa_zkey_read_h_scalars(zkpoa_context* ctx, const zkpoa_zkey* zkey, void* out, unsigned long capacity) {
  if (!ctx || !zkey || !out) return PROVER_ERROR;
  try {
    ZK_HIP(hipSetDevice(ctx->dev.device));
    const uint64_t cnt = zkey->split_world > 1 ? (uint64_t)zkey->domain >> zkey->split_log : zkey->domain;
    if (capacity < cnt * 32) throw ProverError(PROVER_ERROR_SHORT_BUFFER, "h-scalar buffer too small");
    if (!zkey->d_abc) throw ProverError(PROVER_ERROR, "no H scalars on this handle");
    ZK_HIP(hipDeviceSynchronize());
    ZK_HIP(hipMemcpy(out, zkey->d_abc, cnt * 32, hipMemcpyDeviceToHost));
  }
  ZK_PROVER_CATCH(ctx)
  return PROVER_OK;
}

extern "C" int zkpoa_zkey_vkey(const zkpoa_zkey* zkey, uint8_t* buffer, unsigned long* size) {
  if (!zkey || !size) return PROVER_ERROR;
  const unsigned long needed = (unsigned long)zkey->vkey_points.size();
  if (needed == 0) return PROVER_ERROR;   // the handle carries no verification key
  if (!buffer || *size < needed) {
    *size = needed;
    return PROVER_ERROR_SHORT_BUFFER;
  }
  memcpy(buffer, zkey->vkey_points.data(), needed);
  *size = needed;
  return PROVER_OK;
}

extern "C" int zkpoa_proof_to_json(const uint8_t proof_points[256], int style, char* buffer, unsigned long* size) {
  if (!proof_points) return PROVER_ERROR;
  return emit(proof_json(proof_points, style), buffer, size);
}

extern "C" int zkpoa_public_to_json(const uint8_t* public_le, unsigned long n_public, int style, char* buffer,
                                    unsigned long* size) {
  if (!public_le && n_public) return PROVER_ERROR;
  return emit(public_json(public_le, n_public, style), buffer, size);
}

extern "C" int zkpoa_h_scalars(zkpoa_context* ctx, const void* coeffs, unsigned long coeffs_size, const void* witness,
                               uint64_t n_vars, unsigned log_domain, void* out) {
  if (!ctx || !coeffs || !witness || !out) return PROVER_ERROR;
  try {
    ZK_HIP(hipSetDevice(ctx->dev.device));
    if (log_domain > 28) throw ProverError(PROVER_ERROR, "h_scalars: log_domain > 28");
    const uint8_t* cb = reinterpret_cast<const uint8_t*>(coeffs);
    if (coeffs_size < 4) throw ProverError(PROVER_ERROR, "h_scalars: coefficient payload too short");
    uint64_t ncoef = rd_u32(cb);
    if (coeffs_size != 4 + ncoef * 44) throw ProverError(PROVER_ERROR, "h_scalars: coefficient payload has the wrong size");
    const uint32_t domain = 1u << log_domain, rows = 2 * domain;
    hipStream_t st = ctx->dev.lanes[0].stream;
    DevBuf recs(ncoef * 44), cnt((size_t)rows * 4), rank((ncoef ? ncoef : 1) * 4), bs(((size_t)rows / kScanTile + 2) * 4),
        misc(64), row_ptr(((size_t)rows + 1) * 4), sig((ncoef ? ncoef : 1) * 4), vals((ncoef ? ncoef : 1) * 32),
        abc((size_t)3 * domain * 32), wit(n_vars * 32);
    ZK_HIP(hipMemcpy(recs.p, cb + 4, ncoef * 44, hipMemcpyHostToDevice));
    ZK_HIP(hipMemcpy(wit.p, witness, n_vars * 32, hipMemcpyHostToDevice));
    ZK_HIP(hipMemsetAsync(cnt.p, 0, (size_t)rows * 4, st));
    ZK_HIP(hipMemsetAsync(misc.p, 0, 64, st));
    ZK_HIP(hipMemsetAsync(row_ptr.p, 0, ((size_t)rows + 1) * 4, st));
    if (ncoef) {
      uint32_t grid = (uint32_t)((ncoef + 255) / 256);
      hipLaunchKernelGGL(abc_count_kernel, dim3(grid), dim3(256), 0, st, (const CoefRec*)recs.p, ncoef, domain,
                         (uint32_t)n_vars, 0u, 0u, (uint32_t*)cnt.p, (uint32_t*)rank.p, (uint32_t*)misc.p + 4);
      scan_u32(st, (const uint32_t*)cnt.p, rows, 0, 0, (uint32_t*)row_ptr.p, (uint32_t*)bs.p, (uint32_t*)misc.p,
               nullptr);
      hipLaunchKernelGGL(abc_scatter_kernel, dim3(grid), dim3(256), 0, st, (const CoefRec*)recs.p, ncoef, 0u,
                         (const uint32_t*)row_ptr.p, (const uint32_t*)rank.p, (uint32_t*)sig.p, vals.p);
    }
    uint32_t herr = 0;
    ZK_HIP(hipMemcpyAsync(&herr, (uint32_t*)misc.p + 4, 4, hipMemcpyDeviceToHost, st));
    ZK_HIP(hipStreamSynchronize(st));
    if (herr) throw ProverError(PROVER_ERROR, "h_scalars: coefficient record out of range or value >= r");
    ntt_prepare(ctx, st, log_domain);
    uint32_t* long_list = nullptr;
    uint32_t n_long = build_long_list(st, (const uint32_t*)row_ptr.p, rows, &long_list);
    ZK_HIP(hipEventRecord(ctx->ev_a[5], st));
    try {
      h_chain(ctx, st, (const uint32_t*)row_ptr.p, (const uint32_t*)sig.p, vals.p, long_list, n_long, wit.p, domain,
              log_domain, abc.p);
    } catch (...) {
      (void)hipFree(long_list);
      throw;
    }
    ZK_HIP(hipEventRecord(ctx->ev_b[5], st));
    ZK_HIP(hipStreamSynchronize(st));
    ZK_HIP(hipGetLastError());
    ZK_HIP(hipEventElapsedTime(&ctx->ms[3], ctx->ev_a[5], ctx->ev_b[5]));
    (void)hipFree(long_list);
    ZK_HIP(hipMemcpy(out, abc.p, (size_t)domain * 32, hipMemcpyDeviceToHost));
  }
  ZK_PROVER_CATCH(ctx)
  return PROVER_OK;
}

// Test hook (not in the header): the automatic GPU choice of the multi-GPU drop-in for a node of `count` GPUs of which
// those in busy_mask are locked by other provers. out <- the chosen devices; returns how many.
extern "C" int zkpoa_test_auto_pick_devices(int count, unsigned power, unsigned min_power, unsigned long pid,
                                            unsigned busy_mask, int out[8]) {
  if (count <= 0 || count > 32 || !out) return -1;
  unsigned taken = 0;
  std::vector<int> order;
  std::vector<int> ids = auto_pick_devices(count, power, min_power, pid, [&](int d, bool block) {
    if (!block && ((busy_mask | taken) >> d) & 1u) return false;
    taken |= 1u << d;
    order.push_back(d);
    return true;
  }, [&](size_t keep) {
    while (order.size() > keep) {
      taken &= ~(1u << order.back());
      order.pop_back();
    }
  });
  for (size_t i = 0; i < ids.size() && i < 8; i++) out[i] = ids[i];
  return (int)ids.size();
}

// test hook (no GPU): the block size the multi-GPU loader deals sections 5-8 out with (multi_block_log_default)
extern "C" unsigned zkpoa_test_multi_block_log(uint64_t n_vars, unsigned ranks) {
  return multi_block_log_default(n_vars, ranks);
}

extern "C" int groth16_prover(const void* zkey_buffer, unsigned long zkey_size, const void* wtns_buffer,
                              unsigned long wtns_size, char* proof_buffer, unsigned long* proof_size,
                              char* public_buffer, unsigned long* public_size, char* error_msg,
                              unsigned long error_msg_maxsize) {
  if (!zkey_buffer || !wtns_buffer || !proof_size || !public_size) {
    set_err(error_msg, error_msg_maxsize, "null argument");
    return PROVER_ERROR;
  }
  return one_shot(reinterpret_cast<const uint8_t*>(zkey_buffer), zkey_size,
                  WtnsSrc{reinterpret_cast<const uint8_t*>(wtns_buffer), wtns_size, -1}, proof_buffer, proof_size, public_buffer,
                  public_size, error_msg, error_msg_maxsize);
}

extern "C" int groth16_prover_zkey_file(const char* zkey_file_path, const void* wtns_buffer, unsigned long wtns_size,
                                        char* proof_buffer, unsigned long* proof_size, char* public_buffer,
                                        unsigned long* public_size, char* error_msg, unsigned long error_msg_maxsize) {
  if (!zkey_file_path || !wtns_buffer || !proof_size || !public_size) {
    set_err(error_msg, error_msg_maxsize, "null argument");
    return PROVER_ERROR;
  }
  return zkey_file_prove(zkey_file_path, WtnsSrc{reinterpret_cast<const uint8_t*>(wtns_buffer), wtns_size, -1}, proof_buffer,
                         proof_size, public_buffer, public_size, error_msg, error_msg_maxsize);
}

extern "C" int zkpoa_groth16_prover_files(const char* zkey_file_path, const char* wtns_file_path, char* proof_buffer,
                                          unsigned long* proof_size, char* public_buffer, unsigned long* public_size,
                                          char* error_msg, unsigned long error_msg_maxsize) {
  if (!zkey_file_path || !wtns_file_path || !proof_size || !public_size) {
    set_err(error_msg, error_msg_maxsize, "null argument");
    return PROVER_ERROR;
  }
  int wfd = open(wtns_file_path, O_RDONLY | O_CLOEXEC);
  struct stat wsb;
  if (wfd < 0 || fstat(wfd, &wsb) != 0 || !S_ISREG(wsb.st_mode)) {
    if (wfd >= 0) close(wfd);
    set_err(error_msg, error_msg_maxsize, std::string("cannot read witness file ") + wtns_file_path);
    return PROVER_ERROR;
  }
  int rc = zkey_file_prove(zkey_file_path, WtnsSrc{nullptr, (uint64_t)wsb.st_size, wfd}, proof_buffer, proof_size,
                           public_buffer, public_size, error_msg, error_msg_maxsize);
  close(wfd);
  return rc;
}

extern "C" int zkpoa_set_thread_options(const char* r_dec, const char* s_dec, const char* json_style, int verbose) {
  ReqOptions& o = req_options();
  o.active = true;
  o.r = r_dec ? r_dec : "";
  o.s = s_dec ? s_dec : "";
  o.json = json_style ? json_style : "";
  o.verbose = verbose ? "1" : "";
  return PROVER_OK;
}

extern "C" int zkpoa_clear_thread_options(void) {
  req_options() = ReqOptions();
  return PROVER_OK;
}

// The host has nothing to do right now: the library may use the time. One step of background work per call -- today: the
// next fixed-base table of the most recently used resident key that has none yet (ZKPOA_PRECOMP policy "idle") -- so that
// a request arriving meanwhile waits for one table at most. Returns 1 when a step was done (call again while idle), 0
// when there is nothing to do, the policy says no, or a request is in flight (never blocks behind one).
extern "C" int zkpoa_idle_work(void) {
  if (precomp_policy() != kPrecompIdle) return 0;
  std::unique_lock<std::mutex> stage_lk(g_stage_mutex, std::try_to_lock);
  if (!stage_lk.owns_lock() || g_staged_users) return 0;
  std::unique_lock<std::mutex> lk(g_prove_mutex, std::try_to_lock);
  if (!lk.owns_lock()) return 0;
  DeviceSet* ds = nullptr;
  {
    std::lock_guard<std::mutex> dl(g_devset_mutex);
    ds = g_devset;
  }
  if (!ds) return 0;
  const bool verbose = getenv("ZKPOA_VERBOSE") != nullptr;
  try {
    if (ds->ids.size() > 1) {
      for (auto& c : g_multi_cache)
        if (c.mk->proofs_done >= 1 && c.mk->table_bytes == 0 && !c.mk->tables_tried) {
          auto t0 = std::chrono::steady_clock::now();
          c.mk->tables_tried = true;
          multi_precompute(ds, c.mk);
          if (verbose)
            fprintf(stderr, "zkpoa: idle: fixed-base tables for the cached key on %zu ranks: %.2f GB in %.0f ms\n", ds->ids.size(),
                    c.mk->table_bytes / 1e9, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
          return 1;
        }
      return 0;
    }
    CachedKey* pick = nullptr;
    for (auto& c : g_key_cache)
      if (c.zk->proofs_done.load() >= 1 && !c.zk->tables_settled && (!pick || c.last_use > pick->last_use)) pick = &c;
    zkpoa_context* ctx = ds->ctx[0];
    ZK_HIP(hipSetDevice(ctx->dev.device));
    if (!pick) {
      // Tables complete: one throw-away proof on the witness the key still holds. Building the tables gave the lanes'
      // workspaces back to the allocator, and the first proof through the tables would otherwise pay for regrowing them
      // (0.5-0.6 s at the layer-two / -three shapes) inside a request.
      for (auto& c : g_key_cache)
        if (c.zk->tables_settled && c.zk->table_bytes && !c.zk->warmed && c.zk->d_witness && is_full_key(c.zk)) {
          auto t0 = std::chrono::steady_clock::now();
          c.zk->warmed = true;
          uint8_t parts[384];
          prove_partials(ctx, c.zk, parts);
          if (verbose)
            fprintf(stderr, "zkpoa: idle: warm-up proof through the new tables: %.0f ms\n",
                    std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
          return 1;
        }
      return 0;
    }
    const uint64_t before = pick->zk->table_bytes;
    auto t0 = std::chrono::steady_clock::now();
    try {
      (void)zkey_precompute(ctx, pick->zk, 0, 1);
    } catch (const HipError&) {   // out of HBM: what exists stays, nothing more is tried
      (void)hipGetLastError();
      pick->zk->tables_settled = true;
    }
    if (verbose)
      fprintf(stderr, "zkpoa: idle: fixed-base table step for the cached key: +%.2f GB in %.0f ms (%.2f GB so far%s)\n",
              (pick->zk->table_bytes - before) / 1e9,
              std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(),
              pick->zk->table_bytes / 1e9, pick->zk->tables_settled ? ", complete" : "");
    return 1;
  } catch (const std::exception&) {
    return 0;
  }
}
