/* ORACLE (test infrastructure, not product code) -- plain-C restatement of the Groth16 prove path.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 * The product (zk-proof-of-assets_amd/) never links or calls it.
 *
 * Restates, for the CPU, what the reference's external provers compute behind
 * scripts/g16_prove.sh:248-259 (rapidsnark `prover` / `snarkjs groth16 prove`): the algorithm
 * of snarkjs 0.7.2 groth16_prove.js (pnpm-lock.yaml:2231) over ffjavascript 0.2.62 field/curve
 * semantics (pnpm-lock.yaml:1406) -- none of which is vendored under /root/reference -- as
 * specified in SURVEY.md 3.2 and 8c. Function by function:
 *   orc_h_scalars   = buildABC1 + 3 x (Fr.ifft, batchApplyKey(1, inc), Fr.fft) + joinABC
 *   orc_msm_g1/g2   = G1/G2.multiExpAffine (here: textbook Pippenger, unsigned c-bit windows,
 *                     threaded by window like rapidsnark's ParallelMultiexp)
 *   orc_prove       = groth16Prove (five MSMs + randomised assembly)
 *   orc_poseidon2 / orc_merkle_levels = the anonymity-set Poseidon Merkle tree (scripts/merkle_tree.rs)
 *   orc_quotient_check = an NTT-free polynomial-identity test of a complete H-scalar vector (full-size pi_c checks)
 * Pinning: this file is checked against oracle/py (big-int Python), which is itself pinned by the
 * reference's committed proof/vkey fixtures through the pairing verifier; the (zkey, wtns) -> proof
 * map has no golden vector in the reference ("prover parity unpinned" beyond that chain).
 *
 * Arithmetic: 4 x u64 limbs, unsigned __int128 products, Montgomery R = 2^256. Written
 * independently of the product's host_field.hpp (different reduction schedule, Jacobian
 * instead of XYZZ coordinates) so the two do not share bugs.
 */
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef unsigned __int128 u128;
typedef struct { uint64_t v[4]; } fe;           /* field element, Montgomery form unless said otherwise */
typedef struct { const uint64_t* p; uint64_t inv; fe one; fe r2; } field;

static const uint64_t QP[4] = {0x3c208c16d87cfd47ull, 0x97816a916871ca8dull, 0xb85045b68181585dull, 0x30644e72e131a029ull};
static const uint64_t RP[4] = {0x43e1f593f0000001ull, 0x2833e84879b97091ull, 0xb85045b68181585dull, 0x30644e72e131a029ull};
static const field FQ = {QP, 0x87d20782e4866389ull,
  {{0xd35d438dc58f0d9dull, 0x0a78eb28f5c70b3dull, 0x666ea36f7879462cull, 0x0e0a77c19a07df2full}},
  {{0xf32cfc5b538afa89ull, 0xb5e71911d44501fbull, 0x47ab1eff0a417ff6ull, 0x06d89f71cab8351full}}};
static const field FR = {RP, 0xc2e1f593efffffffull,
  {{0xac96341c4ffffffbull, 0x36fc76959f60cd29ull, 0x666ea36f7879462eull, 0x0e0a77c19a07df2full}},
  {{0x1bb8e645ae216da7ull, 0x53fe3ab1e35c59e3ull, 0x8c49833d53bb8085ull, 0x0216d0b17f4e44a5ull}}};

static int fe_is_zero(const fe* a) { return (a->v[0] | a->v[1] | a->v[2] | a->v[3]) == 0; }
static int fe_eq(const fe* a, const fe* b) { return memcmp(a, b, 32) == 0; }
static int geq(const uint64_t* a, const uint64_t* p) {
  for (int i = 3; i >= 0; i--) { if (a[i] > p[i]) return 1; if (a[i] < p[i]) return 0; }
  return 1;
}
static void sub_p(uint64_t* a, const uint64_t* p) {
  u128 b = 0;
  for (int i = 0; i < 4; i++) { u128 d = (u128)a[i] - p[i] - b; a[i] = (uint64_t)d; b = (d >> 64) & 1; }
}
static void fe_add(const field* F, fe* r, const fe* a, const fe* b) {
  u128 c = 0; fe t;
  for (int i = 0; i < 4; i++) { c += (u128)a->v[i] + b->v[i]; t.v[i] = (uint64_t)c; c >>= 64; }
  if (geq(t.v, F->p)) sub_p(t.v, F->p);
  *r = t;
}
static void fe_sub(const field* F, fe* r, const fe* a, const fe* b) {
  u128 bw = 0; fe t;
  for (int i = 0; i < 4; i++) { u128 d = (u128)a->v[i] - b->v[i] - bw; t.v[i] = (uint64_t)d; bw = (d >> 64) & 1; }
  if (bw) { u128 c = 0; for (int i = 0; i < 4; i++) { c += (u128)t.v[i] + F->p[i]; t.v[i] = (uint64_t)c; c >>= 64; } }
  *r = t;
}
static void fe_neg(const field* F, fe* r, const fe* a) { fe z = {{0, 0, 0, 0}}; if (fe_is_zero(a)) *r = *a; else fe_sub(F, r, &z, a); }
/* Montgomery product: full 512-bit schoolbook product, then four reduction rounds (SOS) */
static void fe_mul(const field* F, fe* r, const fe* a, const fe* b) {
  uint64_t t[9] = {0};
  for (int i = 0; i < 4; i++) {
    u128 c = 0;
    for (int j = 0; j < 4; j++) { c += (u128)a->v[i] * b->v[j] + t[i + j]; t[i + j] = (uint64_t)c; c >>= 64; }
    t[i + 4] = (uint64_t)c;
  }
  for (int i = 0; i < 4; i++) {
    uint64_t m = t[i] * F->inv; u128 c = 0;
    for (int j = 0; j < 4; j++) { c += (u128)m * F->p[j] + t[i + j]; t[i + j] = (uint64_t)c; c >>= 64; }
    for (int k = i + 4; k < 9 && c; k++) { c += t[k]; t[k] = (uint64_t)c; c >>= 64; }
  }
  fe o = {{t[4], t[5], t[6], t[7]}};
  if (t[8] || geq(o.v, F->p)) sub_p(o.v, F->p);
  *r = o;
}
static void fe_sqr(const field* F, fe* r, const fe* a) { fe_mul(F, r, a, a); }
static void fe_to_mont(const field* F, fe* r, const fe* a) { fe_mul(F, r, a, &F->r2); }
static void fe_from_mont(const field* F, fe* r, const fe* a) { fe o = {{1, 0, 0, 0}}; fe_mul(F, r, a, &o); }
static void fe_pow(const field* F, fe* r, const fe* a, const uint64_t e[4]) {
  fe res = F->one, base = *a;
  for (int i = 0; i < 4; i++) for (int b = 0; b < 64; b++) {
    if ((e[i] >> b) & 1) fe_mul(F, &res, &res, &base);
    fe_sqr(F, &base, &base);
  }
  *r = res;
}
static void fe_inv(const field* F, fe* r, const fe* a) {
  uint64_t e[4] = {F->p[0] - 2, F->p[1], F->p[2], F->p[3]};
  fe_pow(F, r, a, e);
}
static void fe_from_u64(const field* F, fe* r, uint64_t x) { fe t = {{x, 0, 0, 0}}; fe_to_mont(F, r, &t); }

/* ---- Fq2 = Fq[u]/(u^2+1) ---------------------------------------------------------------------- */
typedef struct { fe c0, c1; } fe2;
static void f2_add(fe2* r, const fe2* a, const fe2* b) { fe_add(&FQ, &r->c0, &a->c0, &b->c0); fe_add(&FQ, &r->c1, &a->c1, &b->c1); }
static void f2_sub(fe2* r, const fe2* a, const fe2* b) { fe_sub(&FQ, &r->c0, &a->c0, &b->c0); fe_sub(&FQ, &r->c1, &a->c1, &b->c1); }
static void f2_mul(fe2* r, const fe2* a, const fe2* b) {   /* schoolbook: 4 multiplications */
  fe t0, t1, t2, t3, o0, o1;
  fe_mul(&FQ, &t0, &a->c0, &b->c0); fe_mul(&FQ, &t1, &a->c1, &b->c1);
  fe_mul(&FQ, &t2, &a->c0, &b->c1); fe_mul(&FQ, &t3, &a->c1, &b->c0);
  fe_sub(&FQ, &o0, &t0, &t1); fe_add(&FQ, &o1, &t2, &t3);
  r->c0 = o0; r->c1 = o1;
}
static int f2_is_zero(const fe2* a) { return fe_is_zero(&a->c0) && fe_is_zero(&a->c1); }
static int f2_eq(const fe2* a, const fe2* b) { return fe_eq(&a->c0, &b->c0) && fe_eq(&a->c1, &b->c1); }
static void f2_inv(fe2* r, const fe2* a) {
  fe n0, n1, d; fe_sqr(&FQ, &n0, &a->c0); fe_sqr(&FQ, &n1, &a->c1); fe_add(&FQ, &d, &n0, &n1); fe_inv(&FQ, &d, &d);
  fe_mul(&FQ, &r->c0, &a->c0, &d); fe_mul(&FQ, &n0, &a->c1, &d); fe_neg(&FQ, &r->c1, &n0);
}

/* ---- generic Jacobian curve arithmetic over "K" = Fq (G1) or Fq2 (G2), via macros ---------------- */
#define DEFINE_CURVE(PFX, K, K_ADD, K_SUB, K_MUL, K_ISZ, K_EQ, K_INV, K_ONE_INIT)                          \
  typedef struct { K x, y; } PFX##_aff;            /* infinity = (0,0) */                                  \
  typedef struct { K x, y, z; } PFX##_jac;         /* infinity: z = 0 */                                   \
  static void PFX##_set_inf(PFX##_jac* p) { memset(p, 0, sizeof(*p)); }                                    \
  static int PFX##_aff_is_inf(const PFX##_aff* p) { return K_ISZ(&p->x) && K_ISZ(&p->y); }                 \
  static void PFX##_dbl(PFX##_jac* r, const PFX##_jac* p) {                                               \
    if (K_ISZ(&p->z) || K_ISZ(&p->y)) { PFX##_set_inf(r); return; }                                        \
    K a, b, c, d, e, f, t, x3, y3, z3;                                                                     \
    K_MUL(&a, &p->x, &p->x); K_MUL(&b, &p->y, &p->y); K_MUL(&c, &b, &b);                                   \
    K_ADD(&t, &p->x, &b); K_MUL(&t, &t, &t); K_SUB(&t, &t, &a); K_SUB(&t, &t, &c); K_ADD(&d, &t, &t);      \
    K_ADD(&e, &a, &a); K_ADD(&e, &e, &a); K_MUL(&f, &e, &e);                                               \
    K_SUB(&x3, &f, &d); K_SUB(&x3, &x3, &d);                                                               \
    K_SUB(&t, &d, &x3); K_MUL(&y3, &e, &t);                                                                \
    K_ADD(&c, &c, &c); K_ADD(&c, &c, &c); K_ADD(&c, &c, &c); K_SUB(&y3, &y3, &c);                          \
    K_MUL(&z3, &p->y, &p->z); K_ADD(&z3, &z3, &z3);                                                        \
    r->x = x3; r->y = y3; r->z = z3;                                                                       \
  }                                                                                                        \
  /* r = p + q (q affine) */                                                                               \
  static void PFX##_madd(PFX##_jac* r, const PFX##_jac* p, const PFX##_aff* q) {                          \
    if (PFX##_aff_is_inf(q)) { *r = *p; return; }                                                          \
    if (K_ISZ(&p->z)) { r->x = q->x; r->y = q->y; K one = K_ONE_INIT; r->z = one; return; }                \
    K z2, u2, s2, h, rr, hh, hhh, v, t, x3, y3, z3;                                                        \
    K_MUL(&z2, &p->z, &p->z); K_MUL(&u2, &q->x, &z2); K_MUL(&s2, &p->z, &z2); K_MUL(&s2, &s2, &q->y);      \
    K_SUB(&h, &u2, &p->x); K_SUB(&rr, &s2, &p->y);                                                         \
    if (K_ISZ(&h)) { if (K_ISZ(&rr)) { PFX##_dbl(r, p); } else { PFX##_set_inf(r); } return; }             \
    K_MUL(&hh, &h, &h); K_MUL(&hhh, &hh, &h); K_MUL(&v, &p->x, &hh);                                       \
    K_MUL(&x3, &rr, &rr); K_SUB(&x3, &x3, &hhh); K_SUB(&x3, &x3, &v); K_SUB(&x3, &x3, &v);                 \
    K_SUB(&t, &v, &x3); K_MUL(&y3, &rr, &t); K_MUL(&t, &p->y, &hhh); K_SUB(&y3, &y3, &t);                  \
    K_MUL(&z3, &p->z, &h);                                                                                 \
    r->x = x3; r->y = y3; r->z = z3;                                                                       \
  }                                                                                                        \
  static void PFX##_add(PFX##_jac* r, const PFX##_jac* p, const PFX##_jac* q) {                           \
    if (K_ISZ(&q->z)) { *r = *p; return; }                                                                 \
    if (K_ISZ(&p->z)) { *r = *q; return; }                                                                 \
    K z1z1, z2z2, u1, u2, s1, s2, h, rr, hh, hhh, v, t, x3, y3, z3;                                        \
    K_MUL(&z1z1, &p->z, &p->z); K_MUL(&z2z2, &q->z, &q->z);                                                \
    K_MUL(&u1, &p->x, &z2z2); K_MUL(&u2, &q->x, &z1z1);                                                    \
    K_MUL(&s1, &q->z, &z2z2); K_MUL(&s1, &s1, &p->y); K_MUL(&s2, &p->z, &z1z1); K_MUL(&s2, &s2, &q->y);    \
    K_SUB(&h, &u2, &u1); K_SUB(&rr, &s2, &s1);                                                             \
    if (K_ISZ(&h)) { if (K_ISZ(&rr)) { PFX##_dbl(r, p); } else { PFX##_set_inf(r); } return; }             \
    K_MUL(&hh, &h, &h); K_MUL(&hhh, &hh, &h); K_MUL(&v, &u1, &hh);                                         \
    K_MUL(&x3, &rr, &rr); K_SUB(&x3, &x3, &hhh); K_SUB(&x3, &x3, &v); K_SUB(&x3, &x3, &v);                 \
    K_SUB(&t, &v, &x3); K_MUL(&y3, &rr, &t); K_MUL(&t, &s1, &hhh); K_SUB(&y3, &y3, &t);                    \
    K_MUL(&z3, &p->z, &q->z); K_MUL(&z3, &z3, &h);                                                         \
    r->x = x3; r->y = y3; r->z = z3;                                                                       \
  }                                                                                                        \
  static void PFX##_to_aff(PFX##_aff* r, const PFX##_jac* p) {                                            \
    if (K_ISZ(&p->z)) { memset(r, 0, sizeof(*r)); return; }                                                \
    K zi, zi2, zi3; K_INV(&zi, &p->z); K_MUL(&zi2, &zi, &zi); K_MUL(&zi3, &zi2, &zi);                      \
    K_MUL(&r->x, &p->x, &zi2); K_MUL(&r->y, &p->y, &zi3);                                                  \
  }                                                                                                        \
  static void PFX##_from_aff(PFX##_jac* r, const PFX##_aff* q) {                                          \
    if (PFX##_aff_is_inf(q)) { PFX##_set_inf(r); return; }                                                 \
    r->x = q->x; r->y = q->y; K one = K_ONE_INIT; r->z = one;                                              \
  }                                                                                                        \
  /* k * p, k = 4 x u64 standard form */                                                                   \
  static void PFX##_mul(PFX##_jac* r, const PFX##_jac* p, const uint64_t k[4]) {                          \
    PFX##_jac acc; PFX##_set_inf(&acc);                                                                    \
    for (int i = 3; i >= 0; i--) for (int b = 63; b >= 0; b--) {                                           \
      PFX##_dbl(&acc, &acc);                                                                               \
      if ((k[i] >> b) & 1) PFX##_add(&acc, &acc, p);                                                       \
    }                                                                                                      \
    *r = acc;                                                                                              \
  }

static void q_add(fe* r, const fe* a, const fe* b) { fe_add(&FQ, r, a, b); }
static void q_sub(fe* r, const fe* a, const fe* b) { fe_sub(&FQ, r, a, b); }
static void q_mul(fe* r, const fe* a, const fe* b) { fe_mul(&FQ, r, a, b); }
static void q_inv(fe* r, const fe* a) { fe_inv(&FQ, r, a); }
#define FQ_ONE_INIT {{0xd35d438dc58f0d9dull, 0x0a78eb28f5c70b3dull, 0x666ea36f7879462cull, 0x0e0a77c19a07df2full}}
#define FQ2_ONE_INIT {FQ_ONE_INIT, {{0, 0, 0, 0}}}
DEFINE_CURVE(g1, fe, q_add, q_sub, q_mul, fe_is_zero, fe_eq, q_inv, FQ_ONE_INIT)
DEFINE_CURVE(g2, fe2, f2_add, f2_sub, f2_mul, f2_is_zero, f2_eq, f2_inv, FQ2_ONE_INIT)

/* ---- Pippenger MSM: unsigned c-bit windows, threaded over (window, point-chunk) tasks ----------------- */
static unsigned scalar_window(const uint64_t* k, unsigned bit, unsigned c) {
  unsigned limb = bit >> 6, off = bit & 63;
  if (limb > 3) return 0;
  uint64_t v = k[limb] >> off;
  if (off + c > 64 && limb < 3) v |= k[limb + 1] << (64 - off);
  return (unsigned)(v & ((1ull << c) - 1));
}
static unsigned pick_c(uint64_t n) {
  unsigned best = 1; double bc = 1e300;
  for (unsigned c = 1; c <= 16; c++) {
    double cost = (double)((254 + c - 1) / c) * ((double)n + 2.0 * (double)(1u << c));
    if (cost < bc) { bc = cost; best = c; }
  }
  return best;
}

/* Work is cut into (window, point-chunk) tasks handed out through an atomic counter, so every host core is busy
 * whatever the window count (rapidsnark's ParallelMultiexp likewise splits the points across threads). */
#define DEFINE_MSM(PFX, AFFSZ)                                                                             \
  typedef struct { const PFX##_aff* bases; const uint64_t* scalars; uint64_t n; unsigned c, W, nchunks;    \
                   volatile unsigned* next; PFX##_jac* part; } PFX##_job;                                  \
  static void* PFX##_worker(void* arg) {                                                                   \
    PFX##_job* j = (PFX##_job*)arg;                                                                        \
    size_t nb = (size_t)1 << j->c;                                                                         \
    PFX##_jac* buckets = (PFX##_jac*)malloc(nb * sizeof(PFX##_jac));                                      \
    for (;;) {                                                                                             \
      unsigned t = __atomic_fetch_add(j->next, 1u, __ATOMIC_RELAXED);                                      \
      if (t >= j->W * j->nchunks) break;                                                                   \
      unsigned w = t / j->nchunks, ch = t % j->nchunks;                                                    \
      uint64_t lo = j->n * ch / j->nchunks, hi = j->n * (ch + 1) / j->nchunks;                             \
      memset(buckets, 0, nb * sizeof(PFX##_jac));                                                          \
      for (uint64_t i = lo; i < hi; i++) {                                                                 \
        unsigned d = scalar_window(j->scalars + 4 * i, w * j->c, j->c);                                    \
        if (d) PFX##_madd(&buckets[d], &buckets[d], &j->bases[i]);                                         \
      }                                                                                                    \
      PFX##_jac run, sum; PFX##_set_inf(&run); PFX##_set_inf(&sum);                                        \
      for (size_t b = nb - 1; b >= 1; b--) { PFX##_add(&run, &run, &buckets[b]); PFX##_add(&sum, &sum, &run); } \
      j->part[t] = sum;                                                                                    \
    }                                                                                                      \
    free(buckets);                                                                                         \
    return 0;                                                                                              \
  }                                                                                                        \
  /* bases: n affine Montgomery points (zkey wire format), scalars: n x 32 B standard form */              \
  int orc_msm_##PFX(const void* bases, const void* scalars, uint64_t n, void* out, int nthreads) {         \
    unsigned c = pick_c(n ? n : 1), W = (254 + c - 1) / c;                                                 \
    if (nthreads < 1) nthreads = 1;                                                                        \
    /* ~2 tasks per thread; a chunk keeps >= 4 * 2^c points so the per-task bucket reduction stays small */\
    unsigned nchunks = (2u * (unsigned)nthreads + W - 1) / W;                                              \
    while (nchunks > 1 && n / nchunks < ((uint64_t)4 << c)) nchunks--;                                     \
    if (nthreads == 1) nchunks = 1;                                                                        \
    if ((unsigned)nthreads > W * nchunks) nthreads = (int)(W * nchunks);                                   \
    PFX##_jac* part = (PFX##_jac*)calloc((size_t)W * nchunks, sizeof(PFX##_jac));                         \
    PFX##_job job = {(const PFX##_aff*)bases, (const uint64_t*)scalars, n, c, W, nchunks, 0, part};        \
    volatile unsigned next = 0; job.next = &next;                                                          \
    pthread_t* th = (pthread_t*)calloc((size_t)nthreads, sizeof(pthread_t));                               \
    for (int t = 1; t < nthreads; t++) pthread_create(&th[t], 0, PFX##_worker, &job);                      \
    PFX##_worker(&job);                                                                                    \
    for (int t = 1; t < nthreads; t++) pthread_join(th[t], 0);                                             \
    PFX##_jac acc; PFX##_set_inf(&acc);                                                                    \
    for (int w = (int)W - 1; w >= 0; w--) {                                                                \
      for (unsigned k = 0; k < c; k++) PFX##_dbl(&acc, &acc);                                              \
      for (unsigned ch = 0; ch < nchunks; ch++) PFX##_add(&acc, &acc, &part[(size_t)w * nchunks + ch]);    \
    }                                                                                                      \
    PFX##_aff r; PFX##_to_aff(&r, &acc); memcpy(out, &r, AFFSZ);                                           \
    free(part); free(th);                                                                                  \
    return 0;                                                                                              \
  }
DEFINE_MSM(g1, 64)
DEFINE_MSM(g2, 128)

/* ---- generators and fixed-base multiplication ------------------------------------------------- */
static void g1_gen(g1_aff* g) { fe_from_u64(&FQ, &g->x, 1); fe_from_u64(&FQ, &g->y, 2); }
static void g2_gen(g2_aff* g) {
  static const uint64_t c[4][4] = {   /* standard form, little-endian limbs */
    {0x46debd5cd992f6edull, 0x674322d4f75edaddull, 0x426a00665e5c4479ull, 0x1800deef121f1e76ull},
    {0x97e485b7aef312c2ull, 0xf1aa493335a9e712ull, 0x7260bfb731fb5d25ull, 0x198e9393920d483aull},
    {0x4ce6cc0166fa7daaull, 0xe3d1e7690c43d37bull, 0x4aab71808dcb408full, 0x12c85ea5db8c6debull},
    {0x55acdadcd122975bull, 0xbc4b313370b38ef3ull, 0xec9e99ad690c3395ull, 0x090689d0585ff075ull}};
  fe t[4];
  for (int i = 0; i < 4; i++) { fe s; memcpy(&s, c[i], 32); fe_to_mont(&FQ, &t[i], &s); }
  g->x.c0 = t[0]; g->x.c1 = t[1]; g->y.c0 = t[2]; g->y.c1 = t[3];
}

/* out[i] = scalars[i] * G (affine, Montgomery): 8-bit fixed-base windows, 32 tables of 255 points */
#define DEFINE_FIXED(PFX, AFFSZ)                                                                           \
  typedef struct { const uint64_t* sc; uint64_t lo, hi; const PFX##_aff* table; char* out; } PFX##_fjob;  \
  static void* PFX##_fworker(void* arg) {                                                                  \
    PFX##_fjob* j = (PFX##_fjob*)arg;                                                                      \
    for (uint64_t i = j->lo; i < j->hi; i++) {                                                             \
      PFX##_jac acc; PFX##_set_inf(&acc);                                                                  \
      const uint8_t* kb = (const uint8_t*)(j->sc + 4 * i);                                                 \
      for (int w = 0; w < 32; w++) if (kb[w]) PFX##_madd(&acc, &acc, &j->table[(size_t)w * 256 + kb[w]]);  \
      PFX##_aff r; PFX##_to_aff(&r, &acc); memcpy(j->out + i * AFFSZ, &r, AFFSZ);                          \
    }                                                                                                      \
    return 0;                                                                                              \
  }                                                                                                        \
  int orc_fixed_base_##PFX(const void* scalars, uint64_t n, void* out, int nthreads) {                     \
    PFX##_aff* table = (PFX##_aff*)calloc(32 * 256, sizeof(PFX##_aff));                                   \
    PFX##_aff g; PFX##_gen(&g);                                                                            \
    PFX##_jac base; PFX##_from_aff(&base, &g);                                                             \
    for (int w = 0; w < 32; w++) {                                                                         \
      PFX##_jac acc; PFX##_set_inf(&acc);                                                                  \
      for (int d = 1; d < 256; d++) { PFX##_add(&acc, &acc, &base); PFX##_to_aff(&table[(size_t)w * 256 + d], &acc); } \
      for (int k = 0; k < 8; k++) PFX##_dbl(&base, &base);                                                 \
    }                                                                                                      \
    if (nthreads < 1) nthreads = 1;                                                                        \
    PFX##_fjob* jobs = (PFX##_fjob*)calloc((size_t)nthreads, sizeof(PFX##_fjob));                         \
    pthread_t* th = (pthread_t*)calloc((size_t)nthreads, sizeof(pthread_t));                               \
    for (int t = 0; t < nthreads; t++) {                                                                   \
      jobs[t] = (PFX##_fjob){(const uint64_t*)scalars, n * t / nthreads, n * (t + 1) / nthreads, table, (char*)out}; \
      if (t) pthread_create(&th[t], 0, PFX##_fworker, &jobs[t]);                                           \
    }                                                                                                      \
    PFX##_fworker(&jobs[0]);                                                                               \
    for (int t = 1; t < nthreads; t++) pthread_join(th[t], 0);                                             \
    free(table); free(jobs); free(th);                                                                     \
    return 0;                                                                                              \
  }
DEFINE_FIXED(g1, 64)
DEFINE_FIXED(g2, 128)

/* ---- NTT over Fr: iterative radix-2, natural order in and out (Fr.fft / Fr.ifft) ---------------- */
static void fr_root(fe* w, unsigned k) {      /* w[k] = 5^((r-1)/2^28) squared 28-k times */
  static const uint64_t e28[4] = {0x9b9709143e1f593full, 0x181585d2833e8487ull, 0x131a029b85045b68ull, 0x000000030644e72eull};
  fe five; fe_from_u64(&FR, &five, 5); fe_pow(&FR, w, &five, e28);
  for (unsigned i = 28; i > k; i--) fe_sqr(&FR, w, w);
}
int orc_ntt(void* data, unsigned k, int inverse) {
  fe* a = (fe*)data; uint64_t n = 1ull << k;
  for (uint64_t i = 1, j = 0; i < n; i++) {
    uint64_t bit = n >> 1;
    for (; j & bit; bit >>= 1) j ^= bit;
    j ^= bit;
    if (i < j) { fe t = a[i]; a[i] = a[j]; a[j] = t; }
  }
  fe wn; fr_root(&wn, k);
  if (inverse) fe_inv(&FR, &wn, &wn);
  for (unsigned s = 1; s <= k; s++) {
    uint64_t len = 1ull << s, half = len >> 1;
    fe wl = wn;
    for (unsigned i = s; i < k; i++) fe_sqr(&FR, &wl, &wl);
    fe* tw = (fe*)malloc(half * sizeof(fe));
    tw[0] = FR.one;
    for (uint64_t t = 1; t < half; t++) fe_mul(&FR, &tw[t], &tw[t - 1], &wl);
    for (uint64_t st = 0; st < n; st += len)
      for (uint64_t t = 0; t < half; t++) {
        fe u = a[st + t], v; fe_mul(&FR, &v, &a[st + t + half], &tw[t]);
        fe_add(&FR, &a[st + t], &u, &v); fe_sub(&FR, &a[st + t + half], &u, &v);
      }
    free(tw);
  }
  if (inverse) {
    fe ninv; fe_from_u64(&FR, &ninv, n); fe_inv(&FR, &ninv, &ninv);
    for (uint64_t i = 0; i < n; i++) fe_mul(&FR, &a[i], &a[i], &ninv);
  }
  return 0;
}

/* ---- buildABC1 -> odd coset -> joinABC --------------------------------------------------------- */
typedef struct __attribute__((packed)) { uint32_t m, c, s; uint8_t val[32]; } coef_rec;
static void to_odd_coset(fe* a, unsigned k) {
  uint64_t n = 1ull << k;
  orc_ntt(a, k, 1);
  fe inc, x = FR.one;
  if (k == 28) fe_from_u64(&FR, &inc, 25); else fr_root(&inc, k + 1);
  for (uint64_t j = 0; j < n; j++) { fe_mul(&FR, &a[j], &a[j], &x); fe_mul(&FR, &x, &x, &inc); }
  orc_ntt(a, k, 0);
}
/* coeffs: payload of zkey section 4; witness: nvars x 32 B standard form; out: 2^k x 32 B standard form */
int orc_h_scalars(const void* coeffs, uint64_t coeffs_size, const void* witness, uint64_t nvars, unsigned k, void* out) {
  const uint8_t* cb = (const uint8_t*)coeffs;
  uint32_t ncoef; memcpy(&ncoef, cb, 4);
  if (coeffs_size != 4 + (uint64_t)ncoef * 44) return 1;
  uint64_t n = 1ull << k;
  fe* A = (fe*)calloc(3 * n, sizeof(fe)); fe* B = A + n; fe* C = B + n;
  const coef_rec* recs = (const coef_rec*)(cb + 4);
  const fe* w = (const fe*)witness;
  for (uint32_t i = 0; i < ncoef; i++) {
    coef_rec r; memcpy(&r, &recs[i], 44);
    if (r.m > 1 || r.c >= n || r.s >= nvars) { free(A); return 2; }
    fe v, p; memcpy(&v, r.val, 32);
    fe_mul(&FR, &p, &v, &w[r.s]);
    fe* dst = (r.m ? B : A) + r.c;
    fe_add(&FR, dst, dst, &p);
  }
  for (uint64_t i = 0; i < n; i++) fe_mul(&FR, &C[i], &A[i], &B[i]);
  to_odd_coset(A, k); to_odd_coset(B, k); to_odd_coset(C, k);
  fe* o = (fe*)out;
  for (uint64_t i = 0; i < n; i++) { fe t; fe_mul(&FR, &t, &A[i], &B[i]); fe_sub(&FR, &t, &t, &C[i]); fe_from_mont(&FR, &o[i], &t); }
  free(A);
  return 0;
}

/* ---- quotient identity: an NTT-free O(n) check of a COMPLETE H-scalar vector ---------------------------
 * groth16_prove.js hands the H MSM the scalars P[i] = (A*B - C)(x_i) on the coset x_i = g*w^i (g = inc =
 * w_{2n}; SURVEY.md 8c "H basis"), where A, B, C (degree < n) interpolate buildABC1's A_T, B_T, C_T = A_T o B_T
 * on the domain {w^c}. A*B - C vanishes on the domain by construction, so A*B - C = Hq * (x^n - 1) with
 * deg Hq <= n - 2 and P[i] = Hq(x_i) * (g^n - 1). Hence, for ANY point z,
 *       A(z) * B(z) - C(z)  ==  Hq(z) * (z^n - 1),
 * with A(z), B(z), C(z) by the barycentric formula over the domain and Hq(z) by the barycentric formula
 * over the coset from the n values P[i] / (g^n - 1). For a random z a wrong vector passes with probability
 * <= 2n / r (Schwartz-Zippel): one field equation checks all n scalars, at full size, with no transform in
 * common with the path under test (buildABC + 4n-element barycentric sums, ~12 n multiplications, threaded).
 * Returns 0 = identity holds, 1 = it does not, other = malformed input. */
typedef struct {
  const coef_rec* recs; uint32_t ncoef; const fe* w; uint64_t nvars; uint64_t n; unsigned k;
  const fe* P; fe z, g; fe* A; fe* B; uint64_t lo, hi; int bad;
  fe sa, sb, sc, sh;
} qc_job;
#define QC_BLK 1024
/* sum_j f[j] * x_j / (z - x_j) over j in [lo, hi), x_j = x0 * w^j: batch inversion per block */
static int qc_bary(const fe* z, const fe* x0, const fe* wn, uint64_t lo, uint64_t hi, const fe* const* f, int nf,
                   int std_form, fe* out) {
  fe xs[QC_BLK], pre[QC_BLK], x, wl; uint64_t e[4] = {lo, 0, 0, 0};
  fe_pow(&FR, &wl, wn, e); fe_mul(&FR, &x, x0, &wl);
  for (int q = 0; q < nf; q++) memset(&out[q], 0, sizeof(fe));
  for (uint64_t s = lo; s < hi; s += QC_BLK) {
    uint64_t cnt = hi - s < QC_BLK ? hi - s : QC_BLK;
    fe acc = FR.one;
    for (uint64_t j = 0; j < cnt; j++) {
      xs[j] = x; fe d; fe_sub(&FR, &d, z, &x);
      if (fe_is_zero(&d)) return 3;
      pre[j] = acc; fe_mul(&FR, &acc, &acc, &d);
      fe_mul(&FR, &x, &x, wn);
    }
    fe inv; fe_inv(&FR, &inv, &acc);
    for (uint64_t j = cnt; j-- > 0;) {
      fe d, wgt; fe_sub(&FR, &d, z, &xs[j]);
      fe_mul(&FR, &wgt, &inv, &pre[j]);            /* 1 / (z - x_j) */
      fe_mul(&FR, &inv, &inv, &d);
      fe_mul(&FR, &wgt, &wgt, &xs[j]);
      for (int q = 0; q < nf; q++) {
        fe v = f[q][s + j], t;
        if (std_form) fe_to_mont(&FR, &v, &v);
        fe_mul(&FR, &t, &v, &wgt); fe_add(&FR, &out[q], &out[q], &t);
      }
    }
  }
  return 0;
}
static void* qc_worker(void* arg) {
  qc_job* j = (qc_job*)arg;
  /* buildABC1 for the constraints [lo, hi): every thread walks the whole list and keeps its own rows */
  for (uint32_t i = 0; i < j->ncoef; i++) {
    coef_rec r; memcpy(&r, &j->recs[i], 44);
    if (r.m > 1 || r.c >= j->n || r.s >= j->nvars) { j->bad = 2; return 0; }
    if (r.c < j->lo || r.c >= j->hi) continue;
    fe v, p; memcpy(&v, r.val, 32);
    fe_mul(&FR, &p, &v, &j->w[r.s]);
    fe* dst = (r.m ? j->B : j->A) + r.c;
    fe_add(&FR, dst, dst, &p);
  }
  fe* C = (fe*)malloc((size_t)(j->hi - j->lo) * sizeof(fe));
  for (uint64_t c = j->lo; c < j->hi; c++) fe_mul(&FR, &C[c - j->lo], &j->A[c], &j->B[c]);
  fe wn; fr_root(&wn, j->k);
  const fe* fs[3] = {j->A, j->B, C - j->lo};
  fe out[3];
  int rc = qc_bary(&j->z, &FR.one, &wn, j->lo, j->hi, fs, 3, 0, out);
  j->sa = out[0]; j->sb = out[1]; j->sc = out[2];
  free(C);
  const fe* fh[1] = {j->P};
  if (!rc) rc = qc_bary(&j->z, &j->g, &wn, j->lo, j->hi, fh, 1, 1, &j->sh);
  if (rc) j->bad = rc;
  return 0;
}
/* coeffs: payload of zkey section 4; witness: nvars x 32 B standard form; h_scalars: 2^k x 32 B standard form
 * (what the H MSM consumes); z_le: the evaluation point, standard form, < r. */
int orc_quotient_check(const void* coeffs, uint64_t coeffs_size, const void* witness, uint64_t nvars, unsigned k,
                       const void* h_scalars, const uint8_t* z_le, int nthreads) {
  const uint8_t* cb = (const uint8_t*)coeffs;
  uint32_t ncoef; memcpy(&ncoef, cb, 4);
  if (coeffs_size != 4 + (uint64_t)ncoef * 44 || k > 28) return 4;
  uint64_t n = 1ull << k;
  if (nthreads < 1) nthreads = 1;
  if ((uint64_t)nthreads > n) nthreads = (int)n;
  fe* A = (fe*)calloc(2 * n, sizeof(fe)); fe* B = A + n;
  fe z, g, t; memcpy(&t, z_le, 32); fe_to_mont(&FR, &z, &t);
  if (k == 28) fe_from_u64(&FR, &g, 25); else fr_root(&g, k + 1);
  qc_job* jobs = (qc_job*)calloc((size_t)nthreads, sizeof(qc_job));
  pthread_t* th = (pthread_t*)calloc((size_t)nthreads, sizeof(pthread_t));
  for (int i = 0; i < nthreads; i++) {
    jobs[i] = (qc_job){(const coef_rec*)(cb + 4), ncoef, (const fe*)witness, nvars, n, k, (const fe*)h_scalars, z, g,
                       A, B, n * (uint64_t)i / (uint64_t)nthreads, n * (uint64_t)(i + 1) / (uint64_t)nthreads, 0};
    if (i) pthread_create(&th[i], 0, qc_worker, &jobs[i]);
  }
  qc_worker(&jobs[0]);
  for (int i = 1; i < nthreads; i++) pthread_join(th[i], 0);
  int rc = 0;
  fe sa, sb, sc, sh; memset(&sa, 0, 32); sb = sc = sh = sa;
  for (int i = 0; i < nthreads; i++) {
    if (jobs[i].bad) rc = jobs[i].bad;
    fe_add(&FR, &sa, &sa, &jobs[i].sa); fe_add(&FR, &sb, &sb, &jobs[i].sb);
    fe_add(&FR, &sc, &sc, &jobs[i].sc); fe_add(&FR, &sh, &sh, &jobs[i].sh);
  }
  free(A); free(jobs); free(th);
  if (rc) return rc;
  /* f(z) = (z^n - 1) / n * S_f;   Hq(z) = (z^n - g^n) / (n g^n (g^n - 1)) * S_h */
  uint64_t en[4] = {n, 0, 0, 0};
  fe zn, gn, zn1, ninv, lhs, rhs, az, bz, cz, hz, den;
  fe_pow(&FR, &zn, &z, en); fe_pow(&FR, &gn, &g, en);
  fe_sub(&FR, &zn1, &zn, &FR.one);
  fe_from_u64(&FR, &ninv, n); fe_inv(&FR, &ninv, &ninv);
  fe_mul(&FR, &t, &zn1, &ninv);
  fe_mul(&FR, &az, &t, &sa); fe_mul(&FR, &bz, &t, &sb); fe_mul(&FR, &cz, &t, &sc);
  fe_sub(&FR, &den, &gn, &FR.one); fe_mul(&FR, &den, &den, &gn);
  if (fe_is_zero(&den)) return 5;
  fe_inv(&FR, &den, &den);
  fe_sub(&FR, &t, &zn, &gn); fe_mul(&FR, &t, &t, &ninv); fe_mul(&FR, &t, &t, &den);
  fe_mul(&FR, &hz, &t, &sh);
  fe_mul(&FR, &lhs, &az, &bz); fe_sub(&FR, &lhs, &lhs, &cz);
  fe_mul(&FR, &rhs, &hz, &zn1);
  return fe_eq(&lhs, &rhs) ? 0 : 1;
}

/* ---- full prove (groth16_prove.js), container parsing included ------------------------------------ */
typedef struct { const uint8_t* p; uint64_t len; } sect;
static int find_sections(const uint8_t* buf, uint64_t size, const char* magic, sect* secs, int maxid) {
  if (size < 12 || memcmp(buf, magic, 4)) return 1;
  uint32_t nsec; memcpy(&nsec, buf + 8, 4);
  uint64_t pos = 12;
  for (int i = 0; i <= maxid; i++) { secs[i].p = 0; secs[i].len = 0; }
  for (uint32_t i = 0; i < nsec; i++) {
    if (pos + 12 > size) return 2;
    uint32_t id; uint64_t len; memcpy(&id, buf + pos, 4); memcpy(&len, buf + pos + 4, 8); pos += 12;
    if (len > size - pos) return 3;
    if ((int)id <= maxid && !secs[id].p) { secs[id].p = buf + pos; secs[id].len = len; }
    pos += len;
  }
  return 0;
}
/* r_le, s_le: 32-byte LE standard-form scalars. proof_points: pi_a(64) pi_b(128) pi_c(64), wire format.
 * public_le: nPublic x 32 B. Returns 0 ok, 3 witness length mismatch, other = error. */
int orc_prove(const void* zkey, uint64_t zkey_size, const void* wtns, uint64_t wtns_size, const uint8_t* r_le,
              const uint8_t* s_le, uint8_t* proof_points, uint8_t* public_le, int nthreads) {
  sect zs[11], ws[3];
  if (find_sections((const uint8_t*)zkey, zkey_size, "zkey", zs, 10)) return 1;
  if (find_sections((const uint8_t*)wtns, wtns_size, "wtns", ws, 2)) return 1;
  for (int i = 1; i <= 9; i++) if (i != 3 && !zs[i].p) return 1;
  if (!ws[1].p || !ws[2].p) return 1;
  const uint8_t* h = zs[2].p;
  uint32_t nvars, npub, domain; memcpy(&nvars, h + 72, 4); memcpy(&npub, h + 76, 4); memcpy(&domain, h + 80, 4);
  if (memcmp(ws[1].p + 4, RP, 32)) return 4;
  uint32_t nwit; memcpy(&nwit, ws[1].p + 36, 4);
  if (nwit != nvars) return 3;
  unsigned k = 0; while ((1u << k) < domain) k++;
  const uint8_t* hp = h + 84;
  g1_aff alpha1, beta1, delta1; g2_aff beta2, delta2;
  memcpy(&alpha1, hp, 64); memcpy(&beta1, hp + 64, 64); memcpy(&beta2, hp + 128, 128);
  memcpy(&delta1, hp + 384, 64); memcpy(&delta2, hp + 448, 128);
  const uint8_t* w = ws[2].p;
  fe* P = (fe*)malloc((size_t)domain * 32);
  if (orc_h_scalars(zs[4].p, zs[4].len, w, nvars, k, P)) { free(P); return 5; }
  g1_aff A, B1, C, H; g2_aff B2;
  orc_msm_g1(zs[5].p, w, nvars, &A, nthreads);
  orc_msm_g1(zs[6].p, w, nvars, &B1, nthreads);
  orc_msm_g2(zs[7].p, w, nvars, &B2, nthreads);
  orc_msm_g1(zs[8].p, w + 32 * (size_t)(npub + 1), nvars - npub - 1, &C, nthreads);
  orc_msm_g1(zs[9].p, P, domain, &H, nthreads);
  free(P);
  uint64_t rk[4], sk[4], nrs[4]; memcpy(rk, r_le, 32); memcpy(sk, s_le, 32);
  fe rm, sm, rs; fe t; memcpy(&t, r_le, 32); fe_to_mont(&FR, &rm, &t); memcpy(&t, s_le, 32); fe_to_mont(&FR, &sm, &t);
  fe_mul(&FR, &rs, &rm, &sm); fe_neg(&FR, &rs, &rs); fe_from_mont(&FR, &t, &rs); memcpy(nrs, &t, 32);
  g1_jac pa, pb1, pc, d1, tmp; g2_jac pb, d2, tmp2;
  g1_from_aff(&d1, &delta1); g2_from_aff(&d2, &delta2);
  g1_from_aff(&pa, &A); g1_madd(&pa, &pa, &alpha1); g1_mul(&tmp, &d1, rk); g1_add(&pa, &pa, &tmp);
  g2_from_aff(&pb, &B2); g2_madd(&pb, &pb, &beta2); g2_mul(&tmp2, &d2, sk); g2_add(&pb, &pb, &tmp2);
  g1_from_aff(&pb1, &B1); g1_madd(&pb1, &pb1, &beta1); g1_mul(&tmp, &d1, sk); g1_add(&pb1, &pb1, &tmp);
  g1_from_aff(&pc, &C); g1_madd(&pc, &pc, &H);
  g1_mul(&tmp, &pa, sk); g1_add(&pc, &pc, &tmp);
  g1_mul(&tmp, &pb1, rk); g1_add(&pc, &pc, &tmp);
  g1_mul(&tmp, &d1, nrs); g1_add(&pc, &pc, &tmp);
  g1_aff oa, oc; g2_aff ob;
  g1_to_aff(&oa, &pa); g2_to_aff(&ob, &pb); g1_to_aff(&oc, &pc);
  memcpy(proof_points, &oa, 64); memcpy(proof_points + 64, &ob, 128); memcpy(proof_points + 192, &oc, 64);
  memcpy(public_le, w + 32, (size_t)npub * 32);
  return 0;
}

/* ---- Poseidon (circomlib parameters, t = 3) and the anonymity-set Merkle tree ------------------------------------
 * Restates what the reference's Rust binary computes (scripts/merkle_tree.rs:138-178 node hash, :206-269 leaves,
 * :411 tree) through light-poseidon 0.2.0 `new_circom(2)` (Cargo.toml:11; not vendored): x^5 S-box, 8 full + 57
 * partial rounds, parameters from the Poseidon reference generator -- an 80-bit Grain LFSR seeded with
 * (field = 1, sbox = 0, n = 254, t, R_F, R_P, thirty 1-bits), 160 warm-up steps, self-shrinking output; constants by
 * rejection sampling, MDS = Cauchy matrix 1 / (x_i + y_j). Pinned by tests/test_poseidon_oracle.py on circomlib's test
 * vector and on the reference's committed Merkle root. */
typedef struct { uint8_t st[80]; } grain;
static int grain_step(grain* g) {
  int nb = g->st[62] ^ g->st[51] ^ g->st[38] ^ g->st[23] ^ g->st[13] ^ g->st[0];
  memmove(g->st, g->st + 1, 79); g->st[79] = (uint8_t)nb;
  return nb;
}
static int grain_bit(grain* g) {
  int nb = grain_step(g);
  while (nb == 0) { grain_step(g); nb = grain_step(g); }
  return grain_step(g);
}
static void grain_init(grain* g, unsigned t, unsigned rf, unsigned rp) {
  unsigned pos = 0;
  unsigned vals[6] = {1, 0, 254, t, rf, rp}, widths[6] = {2, 4, 12, 12, 10, 10};
  for (int f = 0; f < 6; f++) for (int b = (int)widths[f] - 1; b >= 0; b--) g->st[pos++] = (vals[f] >> b) & 1;
  while (pos < 80) g->st[pos++] = 1;
  for (int i = 0; i < 160; i++) grain_step(g);
}
static void grain_254(grain* g, uint64_t out[4]) {   /* 254 bits, most significant first */
  out[0] = out[1] = out[2] = out[3] = 0;
  for (int i = 253; i >= 0; i--) if (grain_bit(g)) out[i >> 6] |= 1ull << (i & 63);
}
#define POS_T 3
#define POS_RF 8
#define POS_RP 57
static fe POS_C[(POS_RF + POS_RP) * POS_T], POS_M[POS_T][POS_T];   /* Montgomery form */
static int pos_ready = 0;
static pthread_mutex_t pos_lock = PTHREAD_MUTEX_INITIALIZER;
static void pos_init(void) {
  pthread_mutex_lock(&pos_lock);
  if (!pos_ready) {
    grain g; grain_init(&g, POS_T, POS_RF, POS_RP);
    for (int i = 0; i < (POS_RF + POS_RP) * POS_T;) {
      fe v; grain_254(&g, v.v);
      if (geq(v.v, RP)) continue;
      fe_to_mont(&FR, &POS_C[i++], &v);
    }
    for (;;) {
      fe xy[2 * POS_T]; int ok = 1;
      for (int i = 0; i < 2 * POS_T; i++) { grain_254(&g, xy[i].v); if (geq(xy[i].v, RP)) sub_p(xy[i].v, RP); }
      for (int i = 0; i < 2 * POS_T && ok; i++) for (int j = 0; j < i; j++) if (fe_eq(&xy[i], &xy[j])) ok = 0;
      for (int i = 0; i < POS_T && ok; i++) for (int j = 0; j < POS_T; j++) {
        fe s, a, b; fe_to_mont(&FR, &a, &xy[i]); fe_to_mont(&FR, &b, &xy[POS_T + j]); fe_add(&FR, &s, &a, &b);
        if (fe_is_zero(&s)) { ok = 0; break; }
        fe_inv(&FR, &POS_M[i][j], &s);
      }
      if (ok) break;
    }
    pos_ready = 1;
  }
  pthread_mutex_unlock(&pos_lock);
}
static void pos_pow5(fe* x) { fe x2, x4; fe_sqr(&FR, &x2, x); fe_sqr(&FR, &x4, &x2); fe_mul(&FR, x, &x4, x); }
/* left, right, out: standard form */
static void pos_hash2(fe* out, const fe* left, const fe* right) {
  fe st[POS_T], nx[POS_T];
  memset(&st[0], 0, sizeof(fe)); fe_to_mont(&FR, &st[1], left); fe_to_mont(&FR, &st[2], right);
  for (int r = 0; r < POS_RF + POS_RP; r++) {
    for (int i = 0; i < POS_T; i++) fe_add(&FR, &st[i], &st[i], &POS_C[r * POS_T + i]);
    if (r < POS_RF / 2 || r >= POS_RF / 2 + POS_RP) for (int i = 0; i < POS_T; i++) pos_pow5(&st[i]);
    else pos_pow5(&st[0]);
    for (int i = 0; i < POS_T; i++) {
      memset(&nx[i], 0, sizeof(fe));
      for (int j = 0; j < POS_T; j++) { fe t; fe_mul(&FR, &t, &POS_M[i][j], &st[j]); fe_add(&FR, &nx[i], &nx[i], &t); }
    }
    memcpy(st, nx, sizeof(st));
  }
  fe_from_mont(&FR, out, &st[0]);
}
typedef struct { const fe* l; const fe* r; fe* o; uint64_t lo, hi, stride; } pos_job;
static void* pos_worker(void* arg) {
  pos_job* j = (pos_job*)arg;
  for (uint64_t i = j->lo; i < j->hi; i++) pos_hash2(&j->o[i], &j->l[i * j->stride], &j->r[i * j->stride]);
  return 0;
}
static void pos_many(const fe* l, const fe* r, uint64_t stride, fe* o, uint64_t n, int nthreads) {
  if (nthreads < 1) nthreads = 1;
  if ((uint64_t)nthreads > n) nthreads = n ? (int)n : 1;
  pos_job* jobs = (pos_job*)calloc((size_t)nthreads, sizeof(pos_job));
  pthread_t* th = (pthread_t*)calloc((size_t)nthreads, sizeof(pthread_t));
  for (int t = 0; t < nthreads; t++) {
    jobs[t] = (pos_job){l, r, o, n * (uint64_t)t / (uint64_t)nthreads, n * (uint64_t)(t + 1) / (uint64_t)nthreads, stride};
    if (t) pthread_create(&th[t], 0, pos_worker, &jobs[t]);
  }
  pos_worker(&jobs[0]);
  for (int t = 1; t < nthreads; t++) pthread_join(th[t], 0);
  free(jobs); free(th);
}
/* n independent hashes: left, right, out = n x 32 B little-endian standard form */
int orc_poseidon2(const void* left, const void* right, uint64_t n, void* out, int nthreads) {
  pos_init();
  pos_many((const fe*)left, (const fe*)right, 1, (fe*)out, n, nthreads);
  return 0;
}
/* Merkle tree as merkle_tree.rs builds it: leaves[i] = H(address[i], balance[i]) for i < n, zero (unhashed) leaves up to
 * 2^log_leaves, node = H(left, right). levels_out: (2^(log_leaves+1) - 1) x 32 B, leaves first, root last. */
int orc_merkle_levels(const void* addresses, const void* balances, uint64_t n, unsigned log_leaves, void* levels_out,
                      int nthreads) {
  pos_init();
  uint64_t N = 1ull << log_leaves;
  if (n > N) return 1;
  fe* lv = (fe*)levels_out;
  memset(lv, 0, (size_t)N * 32);
  pos_many((const fe*)addresses, (const fe*)balances, 1, lv, n, nthreads);
  for (uint64_t w = N; w > 1; w >>= 1) {
    pos_many(lv, lv + 1, 2, lv + w, w / 2, nthreads);
    lv += w;
  }
  return 0;
}

/* ---- element-wise exports used by the oracle-vs-oracle tests ------------------------------------ */
/* field 0 = Fq, 1 = Fr; op 0 = Montgomery mul, 1 = add, 2 = sub, 3 = inverse, 4 = to Montgomery, 5 = from */
int orc_field_op(int fld, int op, const void* a, const void* b, void* out, uint64_t n) {
  const field* F = fld ? &FR : &FQ;
  const fe* x = (const fe*)a; const fe* y = (const fe*)b; fe* o = (fe*)out;
  for (uint64_t i = 0; i < n; i++) {
    switch (op) {
      case 0: fe_mul(F, &o[i], &x[i], &y[i]); break;
      case 1: fe_add(F, &o[i], &x[i], &y[i]); break;
      case 2: fe_sub(F, &o[i], &x[i], &y[i]); break;
      case 3: fe_inv(F, &o[i], &x[i]); break;
      case 4: fe_to_mont(F, &o[i], &x[i]); break;
      default: fe_from_mont(F, &o[i], &x[i]); break;
    }
  }
  return 0;
}
int orc_group_add(int group, const void* a, const void* b, void* out, uint64_t n) {
  for (uint64_t i = 0; i < n; i++) {
    if (group == 1) {
      g1_jac p; g1_aff r; g1_from_aff(&p, (const g1_aff*)a + i); g1_madd(&p, &p, (const g1_aff*)b + i);
      g1_to_aff(&r, &p); memcpy((char*)out + 64 * i, &r, 64);
    } else {
      g2_jac p; g2_aff r; g2_from_aff(&p, (const g2_aff*)a + i); g2_madd(&p, &p, (const g2_aff*)b + i);
      g2_to_aff(&r, &p); memcpy((char*)out + 128 * i, &r, 128);
    }
  }
  return 0;
}
