// Poseidon(2) over the BN254 scalar field and the anonymity-set Merkle tree on the device.
//
// Replaces the reference's only native code, the Rust binary `merkle-tree` (scripts/merkle_tree.rs, built from
// Cargo.toml:18-20, run at scripts/full_workflow.sh:371-380; "2.5 hrs to generate a tree for a set of size 10M",
// merkle_tree.rs:3-5): leaf = Poseidon(address, balance) (:206-269), zero-valued padding leaves up to a power of two
// (:261-266), node = Poseidon(left, right) (:138-178), rs_merkle tree (:411), sibling paths for the owned addresses
// (:290-396). The hash is light-poseidon 0.2.0 `new_circom(2)` = circomlib's Poseidon, width 3, x^5, 8 full + 57
// partial rounds; its parameters come from the Poseidon reference generator (Grain LFSR), regenerated here on the
// host (the crates are not vendored) and pinned by the reference's committed anonymity set / Merkle root.
//
// Device mapping: one hash per lane, the 3-element state in registers (24 VGPRs), round constants and matrices read
// through wave-uniform (scalar) loads from a 14 KB table. Integer-VALU-bound like everything else on this path; 64 B
// in / 32 B out per hash is nowhere near HBM-bound. A level of the tree is one launch.
//
// The 57 partial rounds run in the equivalent sparse form of the Poseidon paper (appendix B; the form circomlib's own
// poseidon_opt uses): round constants of the partial rounds are pushed through the inverse MDS matrix so that only the
// S-box lane carries one, and the MDS product is factored into one dense 2 x 2 product up front plus, per round, a
// matrix with a full first row, a full first column and an identity block -- 5 products instead of 9. A hash is
// 8 x (9 + 9) + 4 + 57 x (3 + 5) = 604 Montgomery products (+3 for the form changes) instead of 828. The tables are
// derived on the host from the plain parameters (host_params); zkpoa_poseidon_params still exports the plain ones,
// and the parity tests compare the device hashes with the oracle's textbook permutation.
#include "bn254_field.hip.h"
#include "zkpoa_internal.hpp"

#include <string.h>

#include <array>
#include <memory>
#include <stdexcept>
#include <utility>
#include <vector>

using namespace zkpoa;

namespace {

constexpr int kT = 3, kRF = 8, kRP = 57, kRounds = kRF + kRP;

struct PoseidonParams {   // Montgomery form; the textbook parameters (host side, zkpoa_poseidon_params)
  Fr C[kRounds * kT];
  Fr M[kT][kT];
};

struct PoseidonTables {   // Montgomery form, device layout: the sparse-partial-round form derived from PoseidonParams
  Fr Cf[kRF][kT];         // constants of the 4 + 4 full rounds
  Fr Cmid[kT];            // added once before the partial rounds
  Fr Mhat[kT - 1][kT - 1];  // (s1, s2) <- (s1, s2) . Mhat before the partial rounds
  Fr C0[kRP];             // constant of the S-box lane after the S-box of partial round r (the last one is zero)
  Fr W[kRP][kT - 1];      // first row of round r's sparse matrix (without M00)
  Fr V[kRP][kT - 1];      // first column
  Fr M[kT][kT];           // MDS matrix (full rounds; M[0][0] also in the partial rounds)
};

// ---- parameter generation (host): generate_parameters_grain.sage 1 0 254 3 8 57 ------------------------------
struct Grain {
  uint8_t st[80];
  int step() {
    int nb = st[62] ^ st[51] ^ st[38] ^ st[23] ^ st[13] ^ st[0];
    memmove(st, st + 1, 79);
    st[79] = (uint8_t)nb;
    return nb;
  }
  int bit() {   // self-shrinking generator: a 0 discards the bit that follows it
    int nb = step();
    while (nb == 0) {
      step();
      nb = step();
    }
    return step();
  }
  Grain(unsigned t, unsigned rf, unsigned rp) {
    const unsigned vals[6] = {1, 0, 254, t, rf, rp}, widths[6] = {2, 4, 12, 12, 10, 10};   // field, sbox, n, t, R_F, R_P
    unsigned pos = 0;
    for (int f = 0; f < 6; f++)
      for (int b = (int)widths[f] - 1; b >= 0; b--) st[pos++] = (vals[f] >> b) & 1u;
    while (pos < 80) st[pos++] = 1;
    for (int i = 0; i < 160; i++) step();
  }
  void bits254(uint64_t out[4]) {   // most significant bit first
    out[0] = out[1] = out[2] = out[3] = 0;
    for (int i = 253; i >= 0; i--)
      if (bit()) out[i >> 6] |= 1ull << (i & 63);
  }
};

bool below_r(const uint64_t v[4]) {
  for (int i = 3; i >= 0; i--) {
    if (v[i] < HFrParams::P[i]) return true;
    if (v[i] > HFrParams::P[i]) return false;
  }
  return false;
}

void host_params(PoseidonParams& out) {
  Grain g(kT, kRF, kRP);
  for (int i = 0; i < kRounds * kT;) {   // round constants: rejection sampling
    uint64_t v[4];
    g.bits254(v);
    if (!below_r(v)) continue;
    HFr c = HFr{{v[0], v[1], v[2], v[3]}}.to_mont();
    memcpy(&out.C[i++], &c, 32);
  }
  for (;;) {   // MDS: Cauchy matrix 1 / (x_i + y_j) over 2t distinct elements (reduced, not rejected)
    HFr xy[2 * kT];
    bool ok = true;
    for (int i = 0; i < 2 * kT; i++) {
      uint64_t v[4];
      g.bits254(v);
      if (!below_r(v)) {   // F(bits): reduced, not rejected (2^254 < 2r: one subtraction)
        unsigned __int128 bw = 0;
        for (int q = 0; q < 4; q++) {
          unsigned __int128 d = (unsigned __int128)v[q] - HFrParams::P[q] - bw;
          v[q] = (uint64_t)d;
          bw = (d >> 64) & 1;
        }
      }
      HFr e{{v[0], v[1], v[2], v[3]}};
      xy[i] = e.to_mont();
    }
    for (int i = 0; i < 2 * kT && ok; i++)
      for (int j = 0; j < i; j++)
        if (xy[i] == xy[j]) ok = false;
    HFr m[kT][kT];
    for (int i = 0; i < kT && ok; i++)
      for (int j = 0; j < kT; j++) {
        HFr s = xy[i] + xy[kT + j];
        if (s.is_zero()) {
          ok = false;
          break;
        }
        m[i][j] = s.inv();
      }
    if (!ok) continue;
    for (int i = 0; i < kT; i++)
      for (int j = 0; j < kT; j++) memcpy(&out.M[i][j], &m[i][j], 32);
    return;
  }
}

// ---- the sparse form of the partial rounds (Poseidon paper, appendix B), derived on the host ---------------------
// n x n inverse by Gauss-Jordan (n <= 3); throws on a singular matrix (cannot happen for an MDS matrix)
template <int N>
void mat_inv(const HFr (&a)[N][N], HFr (&out)[N][N]) {
  HFr w[N][2 * N];
  for (int i = 0; i < N; i++)
    for (int j = 0; j < N; j++) {
      w[i][j] = a[i][j];
      w[i][N + j] = i == j ? HFr::one() : HFr::zero();
    }
  for (int c = 0; c < N; c++) {
    int p = c;
    while (p < N && w[p][c].is_zero()) p++;
    if (p == N) throw std::runtime_error("poseidon: singular matrix while deriving the sparse round form");
    for (int j = 0; j < 2 * N; j++) std::swap(w[c][j], w[p][j]);
    HFr iv = w[c][c].inv();
    for (int j = 0; j < 2 * N; j++) w[c][j] = w[c][j] * iv;
    for (int r = 0; r < N; r++) {
      if (r == c || w[r][c].is_zero()) continue;
      HFr f = w[r][c];
      for (int j = 0; j < 2 * N; j++) w[r][j] = w[r][j] - f * w[c][j];
    }
  }
  for (int i = 0; i < N; i++)
    for (int j = 0; j < N; j++) out[i][j] = w[i][N + j];
}

void host_tables(const PoseidonParams& plain, PoseidonTables& out) {
  auto get = [](const Fr& f) {
    HFr v;
    memcpy(&v, &f, 32);
    return v;
  };
  auto put = [](Fr& f, const HFr& v) { memcpy(&f, &v, 32); };
  HFr M[kT][kT], MT[kT][kT], MTinv[kT][kT];
  for (int i = 0; i < kT; i++)
    for (int j = 0; j < kT; j++) {
      M[i][j] = get(plain.M[i][j]);
      MT[j][i] = M[i][j];
    }
  mat_inv<kT>(MT, MTinv);
  // 1. constants: walking backwards over the partial rounds, c_(i+1) moves through M^-1 to the round before it;
  //    only its S-box-lane component stays behind (to be added after that round's S-box)
  std::vector<std::array<HFr, kT>> ct(kRounds);
  for (int r = 0; r < kRounds; r++)
    for (int i = 0; i < kT; i++) ct[r][i] = get(plain.C[r * kT + i]);
  const int Rf = kRF / 2;
  for (int i = kRounds - 2 - Rf; i >= Rf; i--) {
    HFr moved[kT];   // row vector ct[i+1] times MT^-1
    for (int j = 0; j < kT; j++) {
      HFr acc = HFr::zero();
      for (int k = 0; k < kT; k++) acc = acc + ct[i + 1][k] * MTinv[k][j];
      moved[j] = acc;
    }
    for (int j = 1; j < kT; j++) ct[i][j] = ct[i][j] + moved[j];
    ct[i + 1][0] = moved[0];
    for (int j = 1; j < kT; j++) ct[i + 1][j] = HFr::zero();
  }
  // 2. matrices: M = M' . M'' with M'' sparse (first row, first column, identity block), repeated from the last
  //    partial round backwards; what is left over at the front is one dense (t-1) x (t-1) block
  HFr Mmul[kT][kT], Mi[kT][kT];
  for (int i = 0; i < kT; i++)
    for (int j = 0; j < kT; j++) Mmul[i][j] = MT[i][j];
  for (int j = kRP - 1; j >= 0; j--) {   // j counts the rounds from the back: tables are stored in execution order
    HFr hat[kT - 1][kT - 1], hat_inv[kT - 1][kT - 1];
    for (int a = 1; a < kT; a++)
      for (int b = 1; b < kT; b++) hat[a - 1][b - 1] = Mmul[a][b];
    mat_inv<kT - 1>(hat, hat_inv);
    for (int a = 1; a < kT; a++) {
      put(out.V[j][a - 1], Mmul[0][a]);
      HFr acc = HFr::zero();   // (hat^-1 . w)[a-1], w = first column of Mmul below the corner
      for (int b = 1; b < kT; b++) acc = acc + hat_inv[a - 1][b - 1] * Mmul[b][0];
      put(out.W[j][a - 1], acc);
    }
    for (int a = 0; a < kT; a++)
      for (int b = 0; b < kT; b++) Mi[a][b] = a == b ? HFr::one() : HFr::zero();
    for (int a = 1; a < kT; a++)
      for (int b = 1; b < kT; b++) Mi[a][b] = hat[a - 1][b - 1];
    for (int a = 0; a < kT; a++)
      for (int b = 0; b < kT; b++) {
        HFr acc = HFr::zero();
        for (int k = 0; k < kT; k++) acc = acc + MT[a][k] * Mi[k][b];
        Mmul[a][b] = acc;
      }
  }
  for (int a = 1; a < kT; a++)
    for (int b = 1; b < kT; b++) put(out.Mhat[a - 1][b - 1], Mi[a][b]);
  for (int k = 0; k < Rf; k++)
    for (int i = 0; i < kT; i++) {
      put(out.Cf[k][i], ct[k][i]);
      put(out.Cf[Rf + k][i], ct[Rf + kRP + k][i]);
    }
  for (int i = 0; i < kT; i++) put(out.Cmid[i], ct[Rf][i]);
  for (int r = 0; r < kRP; r++) put(out.C0[r], r < kRP - 1 ? ct[Rf + 1 + r][0] : HFr::zero());
  for (int i = 0; i < kT; i++)
    for (int j = 0; j < kT; j++) put(out.M[i][j], M[i][j]);
}

// ---- device ----------------------------------------------------------------------------------------------------
ZK_DEV Fr pow5(const Fr& x) {
  Fr x2 = x.sqr();
  Fr x4 = x2.sqr();
  return x4 * x;
}

ZK_DEV void full_round(const PoseidonTables* __restrict__ prm, int k, Fr& s0, Fr& s1, Fr& s2) {
  s0 = pow5(s0 + prm->Cf[k][0]);
  s1 = pow5(s1 + prm->Cf[k][1]);
  s2 = pow5(s2 + prm->Cf[k][2]);
  // every row of the MDS product is one sum of three products under a single Montgomery reduction (Fr::dot3: the
  // matrix entries are canonical, the state lazily reduced) -- 3 reductions per round instead of 9
  Fr n0 = Fr::dot3(s0, prm->M[0][0], s1, prm->M[0][1], s2, prm->M[0][2]);
  Fr n1 = Fr::dot3(s0, prm->M[1][0], s1, prm->M[1][1], s2, prm->M[1][2]);
  Fr n2 = Fr::dot3(s0, prm->M[2][0], s1, prm->M[2][1], s2, prm->M[2][2]);
  s0 = n0;
  s1 = n1;
  s2 = n2;
}

// standard-form inputs -> standard-form hash
ZK_DEV Fr poseidon2(const PoseidonTables* __restrict__ prm, const Fr& left, const Fr& right) {
  Fr s0 = Fr::zero(), s1 = left.to_mont(), s2 = right.to_mont();
  for (int k = 0; k < kRF / 2; k++) full_round(prm, k, s0, s1, s2);
  s0 = s0 + prm->Cmid[0];
  s1 = s1 + prm->Cmid[1];
  s2 = s2 + prm->Cmid[2];
  {
    Fr n1 = Fr::dot2(s1, prm->Mhat[0][0], s2, prm->Mhat[1][0]);
    Fr n2 = Fr::dot2(s1, prm->Mhat[0][1], s2, prm->Mhat[1][1]);
    s1 = n1;
    s2 = n2;
  }
  for (int r = 0; r < kRP; r++) {
    s0 = pow5(s0) + prm->C0[r];
    Fr n0 = Fr::dot3(s0, prm->M[0][0], s1, prm->W[r][0], s2, prm->W[r][1]);
    s1 = s1 + s0 * prm->V[r][0];
    s2 = s2 + s0 * prm->V[r][1];
    s0 = n0;
  }
  for (int k = kRF / 2; k < kRF; k++) full_round(prm, k, s0, s1, s2);
  return s0.from_mont();
}

// out[i] = H(left[i * stride], right[i * stride]) for i < n; elements are 32 B LE standard form.
// stride 1: independent pairs (leaves: addresses / balances); stride 2 with right = left + 1: one tree level.
__global__ __launch_bounds__(256) void poseidon2_kernel(const PoseidonTables* __restrict__ prm, const void* left,
                                                        const void* right, uint64_t stride, uint64_t n, void* out) {
  uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
  if (i >= n) return;
  Fr l = load_field<Fr>(reinterpret_cast<const char*>(left) + 32 * i * stride);
  Fr r = load_field<Fr>(reinterpret_cast<const char*>(right) + 32 * i * stride);
  store_field(reinterpret_cast<char*>(out) + 32 * i, poseidon2(prm, l, r));
}

void launch_hash(hipStream_t st, const PoseidonTables* prm, const void* l, const void* r, uint64_t stride, uint64_t n,
                 void* out) {
  if (n)
    hipLaunchKernelGGL(poseidon2_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, st, prm, l, r, stride, n, out);
}

}  // namespace

// parameters live with the context (uploaded on first use)
struct zkpoa_poseidon_state {
  PoseidonTables* d = nullptr;
};

static const PoseidonTables* device_params(zkpoa_context* ctx) {
  if (!ctx->poseidon) {
    std::unique_ptr<zkpoa_poseidon_state> s(new zkpoa_poseidon_state());
    std::unique_ptr<PoseidonParams> h(new PoseidonParams());
    std::unique_ptr<PoseidonTables> tb(new PoseidonTables());
    host_params(*h);
    host_tables(*h, *tb);
    ZK_HIP(hipMalloc(reinterpret_cast<void**>(&s->d), sizeof(PoseidonTables)));
    ZK_HIP(hipMemcpy(s->d, tb.get(), sizeof(PoseidonTables), hipMemcpyHostToDevice));
    ctx->poseidon = s.release();
  }
  return ctx->poseidon->d;
}

namespace zkpoa {
void poseidon_release(zkpoa_context* ctx) {
  if (ctx->poseidon) {
    if (ctx->poseidon->d) (void)hipFree(ctx->poseidon->d);
    delete ctx->poseidon;
    ctx->poseidon = nullptr;
  }
}
}  // namespace zkpoa

// ---- C ABI ---------------------------------------------------------------------------------------------------------
// host only (no GPU): the parameters as generated here, standard form: 195 round constants then the 3 x 3 MDS matrix
// row-major, 204 x 32 B LE -- what the CPU tests compare with circomlib's published values
extern "C" int zkpoa_poseidon_params(uint8_t out[204 * 32]) {
  if (!out) return PROVER_ERROR;
  std::unique_ptr<PoseidonParams> h(new PoseidonParams());
  host_params(*h);
  const Fr* all = h->C;
  for (int i = 0; i < kRounds * kT + kT * kT; i++) {
    const Fr* src = i < kRounds * kT ? &all[i] : &h->M[(i - kRounds * kT) / kT][(i - kRounds * kT) % kT];
    HFr v;
    memcpy(&v, src, 32);
    v = v.from_mont();
    memcpy(out + 32 * i, &v, 32);
  }
  return PROVER_OK;
}

extern "C" int zkpoa_poseidon2_device(zkpoa_context* ctx, const void* d_left, const void* d_right, uint64_t n, void* d_out) {
  ZK_API_BEGIN(ctx)
  hipStream_t st = ctx->dev.lanes[0].stream;
  launch_hash(st, device_params(ctx), d_left, d_right, 1, n, d_out);
  ZK_HIP(hipStreamSynchronize(st));
  ZK_HIP(hipGetLastError());
  ZK_API_END(ctx)
}

extern "C" int zkpoa_poseidon2(zkpoa_context* ctx, const void* left, const void* right, uint64_t n, void* out) {
  ZK_API_BEGIN(ctx)
  DevBuf dl(n * 32), dr(n * 32), dout(n * 32);
  ZK_HIP(hipMemcpy(dl.p, left, n * 32, hipMemcpyHostToDevice));
  ZK_HIP(hipMemcpy(dr.p, right, n * 32, hipMemcpyHostToDevice));
  hipStream_t st = ctx->dev.lanes[0].stream;
  launch_hash(st, device_params(ctx), dl.p, dr.p, 1, n, dout.p);
  ZK_HIP(hipStreamSynchronize(st));
  ZK_HIP(hipGetLastError());
  ZK_HIP(hipMemcpy(out, dout.p, n * 32, hipMemcpyDeviceToHost));
  ZK_API_END(ctx)
}

struct zkpoa_merkle {
  void* d_levels = nullptr;   // (2^(k+1) - 1) nodes, leaves first, root last; 32 B LE standard form
  unsigned log_leaves = 0;
  uint64_t n_set = 0;         // leaves that came from the anonymity set (the rest are zero padding)
  float build_ms = 0;
};

// level l (0 = leaves) starts at node offset 2^(k+1) - 2^(k+1-l)
static uint64_t level_offset(unsigned k, unsigned l) { return (2ull << k) - (2ull << (k - l)); }

extern "C" int zkpoa_merkle_build_device(zkpoa_context* ctx, const void* d_addresses, const void* d_balances, uint64_t n,
                                         zkpoa_merkle** out) {
  if (!out) return PROVER_ERROR;
  *out = nullptr;
  ZK_API_BEGIN(ctx)
  if (n == 0 || n > (1ull << 30)) throw HipError("merkle: anonymity set size must be in [1, 2^30]");
  unsigned k = 0;
  while ((1ull << k) < n) k++;   // merkle_tree.rs:261-266: height = ceil(log2(size)), zero leaves up to 2^height
  std::unique_ptr<zkpoa_merkle> t(new zkpoa_merkle());
  t->log_leaves = k;
  t->n_set = n;
  const uint64_t N = 1ull << k, nodes = 2 * N - 1;
  ZK_HIP(hipMalloc(&t->d_levels, nodes * 32));
  hipStream_t st = ctx->dev.lanes[0].stream;
  const PoseidonTables* prm = device_params(ctx);
  char* lv = reinterpret_cast<char*>(t->d_levels);
  try {
    ZK_HIP(hipEventRecord(ctx->ev_a[0], st));
    if (N > n) ZK_HIP(hipMemsetAsync(lv + n * 32, 0, (N - n) * 32, st));
    launch_hash(st, prm, d_addresses, d_balances, 1, n, lv);
    for (unsigned l = 0; l < k; l++) {
      const char* in = lv + level_offset(k, l) * 32;
      launch_hash(st, prm, in, in + 32, 2, N >> (l + 1), lv + level_offset(k, l + 1) * 32);
    }
    ZK_HIP(hipEventRecord(ctx->ev_b[0], st));
    ZK_HIP(hipStreamSynchronize(st));
    ZK_HIP(hipGetLastError());
    ZK_HIP(hipEventElapsedTime(&t->build_ms, ctx->ev_a[0], ctx->ev_b[0]));
    ctx->ms[7] = t->build_ms;
  } catch (...) {
    (void)hipFree(t->d_levels);
    throw;
  }
  *out = t.release();
  ZK_API_END(ctx)
}

extern "C" int zkpoa_merkle_build(zkpoa_context* ctx, const void* addresses, const void* balances, uint64_t n,
                                  zkpoa_merkle** out) {
  if (!out) return PROVER_ERROR;
  *out = nullptr;
  ZK_API_BEGIN(ctx)
  DevBuf da(n * 32), db(n * 32);
  ctx->uploader.upload(da.p, addresses, n * 32, ctx->dev.device, ctx->dev.lanes[0].stream);
  ctx->uploader.upload(db.p, balances, n * 32, ctx->dev.device, ctx->dev.lanes[0].stream);
  int rc = zkpoa_merkle_build_device(ctx, da.p, db.p, n, out);
  if (rc != PROVER_OK) return rc;
  ZK_API_END(ctx)
}

extern "C" void zkpoa_merkle_free(zkpoa_context* ctx, zkpoa_merkle* tree) {
  if (!tree) return;
  if (ctx) (void)hipSetDevice(ctx->dev.device);
  if (tree->d_levels) (void)hipFree(tree->d_levels);
  delete tree;
}

extern "C" int zkpoa_merkle_info(const zkpoa_merkle* tree, uint64_t out[3]) {
  if (!tree || !out) return PROVER_ERROR;
  out[0] = tree->n_set;
  out[1] = tree->log_leaves;          // path length; rs_merkle's depth() is this + 1
  out[2] = (2ull << tree->log_leaves) - 1;
  return PROVER_OK;
}

extern "C" int zkpoa_merkle_root(zkpoa_context* ctx, const zkpoa_merkle* tree, uint8_t root_le[32]) {
  if (!tree || !root_le) return PROVER_ERROR;
  ZK_API_BEGIN(ctx)
  const uint64_t nodes = (2ull << tree->log_leaves) - 1;
  ZK_HIP(hipMemcpy(root_le, reinterpret_cast<const char*>(tree->d_levels) + (nodes - 1) * 32, 32, hipMemcpyDeviceToHost));
  ZK_API_END(ctx)
}

// leaves [first, first + count) as the tree stores them (hashes; zero for padding)
extern "C" int zkpoa_merkle_leaves(zkpoa_context* ctx, const zkpoa_merkle* tree, uint64_t first, uint64_t count, void* out) {
  if (!tree || !out) return PROVER_ERROR;
  ZK_API_BEGIN(ctx)
  const uint64_t n_leaves = 1ull << tree->log_leaves;
  if (first > n_leaves || count > n_leaves - first) throw HipError("merkle: leaf range out of bounds");
  ZK_HIP(hipMemcpy(out, reinterpret_cast<const char*>(tree->d_levels) + first * 32, count * 32, hipMemcpyDeviceToHost));
  ZK_API_END(ctx)
}

// sibling path of a leaf, from the leaves up: path_le = log_leaves x 32 B, path_indices = log_leaves bytes (the index
// bit per level: merkle_tree.rs build_path_indices)
extern "C" int zkpoa_merkle_path(zkpoa_context* ctx, const zkpoa_merkle* tree, uint64_t leaf_index, uint8_t* path_le,
                                 uint8_t* path_indices) {
  if (!tree || !path_le) return PROVER_ERROR;
  ZK_API_BEGIN(ctx)
  const unsigned k = tree->log_leaves;
  if (leaf_index >> k) throw HipError("merkle: leaf index out of range");
  uint64_t i = leaf_index;
  for (unsigned l = 0; l < k; l++) {
    const char* src = reinterpret_cast<const char*>(tree->d_levels) + (level_offset(k, l) + (i ^ 1)) * 32;
    ZK_HIP(hipMemcpyAsync(path_le + 32 * l, src, 32, hipMemcpyDeviceToHost, ctx->dev.lanes[0].stream));
    if (path_indices) path_indices[l] = (uint8_t)(i & 1);
    i >>= 1;
  }
  ZK_HIP(hipStreamSynchronize(ctx->dev.lanes[0].stream));
  ZK_API_END(ctx)
}
