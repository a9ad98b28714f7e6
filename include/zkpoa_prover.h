/* libzkpoa_prover.so -- C ABI of the MI355X (gfx950) Groth16 prover for zk-proof-of-assets.
 *
 * Drop-in boundary. The reference has an EXEC boundary, not an FFI:
 *   scripts/g16_prove.sh:248-252   "$rapidsnark_path" zkey witness.wtns proof.json public.json
 *   scripts/g16_prove.sh:255-259   npx snarkjs groth16 prove   (same four positional arguments)
 * The `prover` executable built from this library takes exactly that argv. The library
 * entry points below are what an FFI binding for the same step would bind; the first two
 * mirror iden3/rapidsnark's prover.h (the C API of the binary the reference execs), the
 * zkpoa_* ones expose the stages of the path for device-resident use and for parity tests.
 *
 * Conventions: plain C types, caller-allocated buffers, no global state except what hangs
 * off a zkpoa_context. All byte formats are the reference's wire formats:
 *   Fq/Fr element : 32 bytes little-endian
 *   G1 point      : 64 bytes  = x||y, affine, Montgomery form (R = 2^256), infinity = zeros
 *   G2 point      : 128 bytes = x.c0||x.c1||y.c0||y.c1, same conventions
 *   scalar        : 32 bytes little-endian, standard (non-Montgomery) form, < r
 * exactly as in .zkey sections 5-9 and .wtns section 2.
 *
 * The compute path is HIP-only: every function that needs the GPU fails with an error
 * (never falls back to the CPU) when no gfx950 device is usable.
 */
#ifndef ZKPOA_PROVER_H
#define ZKPOA_PROVER_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- return codes: same values as rapidsnark prover.h ---------------------------------- */
#define PROVER_OK 0x0
#define PROVER_ERROR 0x1
#define PROVER_ERROR_SHORT_BUFFER 0x2
#define PROVER_INVALID_WITNESS_LENGTH 0x3
/* Extension (not in rapidsnark): the failure came from the HIP runtime (sticky GPU fault, lost device, out of device or
 * host memory), not from the inputs -- the process's GPU context cannot be trusted any more. Returned by the two
 * groth16_prover* entry points; callers that only test != PROVER_OK are unaffected. */
#define PROVER_ERROR_RUNTIME 0x4

/* ---- one-shot prove: replaces the process exec'd at scripts/g16_prove.sh:248-252 -------- */
/* zkey / wtns are complete file images. proof_buffer / public_buffer receive NUL-terminated
 * JSON text; *proof_size / *public_size are in: capacity, out: bytes needed incl. NUL (also
 * on PROVER_ERROR_SHORT_BUFFER). JSON style is rapidsnark's unless env ZKPOA_JSON=snarkjs.
 * r, s come from /dev/urandom unless env ZKPOA_R / ZKPOA_S (decimal) are set.
 * Self-check: like the reference, which verifies every proof right after proving it (scripts/g16_verify.sh:213-216),
 * the first proof of every key is verified against the verification key the zkey itself carries (sections 2-3)
 * before anything is returned; a failure is PROVER_ERROR with a message. Env ZKPOA_SELFCHECK=0 disables it,
 * =all checks every proof (host only, ~7 ms). */
int groth16_prover(const void* zkey_buffer, unsigned long zkey_size,
                   const void* wtns_buffer, unsigned long wtns_size,
                   char* proof_buffer, unsigned long* proof_size,
                   char* public_buffer, unsigned long* public_size,
                   char* error_msg, unsigned long error_msg_maxsize);

/* Same, reading the zkey from a path (mmap) -- what the `prover` CLI uses. */
int groth16_prover_zkey_file(const char* zkey_file_path,
                             const void* wtns_buffer, unsigned long wtns_size,
                             char* proof_buffer, unsigned long* proof_size,
                             char* public_buffer, unsigned long* public_size,
                             char* error_msg, unsigned long error_msg_maxsize);

/* Same again with BOTH inputs as paths -- exactly the argv of the process exec'd at scripts/g16_prove.sh:248-252
 * (`prover <zkey> <wtns> ...`), and what the `prover` executable calls. Extension (rapidsnark's prover.h has no such
 * entry): the witness is never mapped or copied on the host -- its values go from the page cache straight into the
 * pinned staging buffers of the upload (pread), which at the layer-three size (1.7 GB .wtns) saves the ~50 ms a mapping
 * costs to populate and tear down. Same return codes, buffers and environment as groth16_prover_zkey_file. */
int zkpoa_groth16_prover_files(const char* zkey_file_path, const char* wtns_file_path,
                               char* proof_buffer, unsigned long* proof_size,
                               char* public_buffer, unsigned long* public_size,
                               char* error_msg, unsigned long error_msg_maxsize);

/* Per-thread request options for the three entry points above. They read r, s (ZKPOA_R / ZKPOA_S, decimal; test use),
 * the JSON style (ZKPOA_JSON=snarkjs) and verbosity (ZKPOA_VERBOSE) from the environment; a host that serves several
 * requests from several threads (the `prover` executable's resident server does) cannot change its environment per
 * request. After zkpoa_set_thread_options the CALLING THREAD's following calls use these values instead: a NULL string /
 * verbose 0 means "unset for this thread, whatever the environment says". zkpoa_clear_thread_options returns the thread
 * to the environment. Calls from different threads may overlap: a request on a resident key stages its witness while
 * the previous request is still proving. */
int zkpoa_set_thread_options(const char* r_decimal, const char* s_decimal, const char* json_style, int verbose);
int zkpoa_clear_thread_options(void);

/* "The host has nothing to do right now." groth16_prover_zkey_file / zkpoa_groth16_prover_files keep keys resident
 * (env ZKPOA_KEY_CACHE) and prove ~10 % faster once a key has its fixed-base tables, but building them is seconds of GPU
 * time at the layer-two / -three sizes -- so by default (ZKPOA_PRECOMP unset or "idle") that never happens inside a
 * request: each call of zkpoa_idle_work builds ONE table of the most recently used key that still lacks some and returns
 * 1 (call again while idle), or returns 0 at once when there is nothing to do or a request is in flight. The `prover`
 * server calls it after 300 ms without a request. A host that never calls it gets the whole set built inside the request
 * that follows a key's ZKPOA_PRECOMP_AFTER-th proof (default 64). ZKPOA_PRECOMP=eager: at the second use, inside the
 * request (r03); =0: never. */
int zkpoa_idle_work(void);

/* ---- context ----------------------------------------------------------------------------- */
typedef struct zkpoa_context zkpoa_context;
typedef struct zkpoa_zkey zkpoa_zkey;

/* Creates a context on HIP device `device` (streams + workspace). Fails if no GPU.
 * Threads: a context is not re-entrant -- calls on one context must not overlap, with one exception: the lane-addressed
 * MSM calls (zkpoa_msm_g1_device_lane, zkpoa_msm_table_run_lane) may run concurrently on DISTINCT lanes. Use one
 * context per host thread (or per GPU) otherwise; the one-shot groth16_prover* entry points serialise themselves. */
int zkpoa_context_create(int device, zkpoa_context** ctx, char* error_msg, unsigned long error_msg_maxsize);
void zkpoa_context_destroy(zkpoa_context* ctx);
/* message of the last failing call on this context (valid until the next call) */
const char* zkpoa_last_error(const zkpoa_context* ctx);

/* ---- device-resident proving key: replaces snarkjs readSection(zkey, 4..9) per prove ------ */
/* Parses the binfile container and uploads sections 4-9 to HBM once. */
int zkpoa_zkey_load(zkpoa_context* ctx, const void* zkey_buffer, unsigned long zkey_size, zkpoa_zkey** zkey);
void zkpoa_zkey_free(zkpoa_context* ctx, zkpoa_zkey* zkey);
/* header fields: out[0]=nVars, out[1]=nPublic, out[2]=domainSize, out[3]=nCoefs */
int zkpoa_zkey_info(const zkpoa_zkey* zkey, uint64_t out[4]);

/* Full prove against a resident key: groth16_prove.js steps 2-7.
 * r_le / s_le: 32-byte little-endian standard-form scalars, or NULL for /dev/urandom.
 * proof_points: 64 (pi_a) + 128 (pi_b) + 64 (pi_c) bytes, wire format above.
 * public_le: nPublic * 32 bytes (w[1..nPublic], standard form). */
int zkpoa_prove(zkpoa_context* ctx, const zkpoa_zkey* zkey,
                const void* wtns_buffer, unsigned long wtns_size,
                const uint8_t* r_le, const uint8_t* s_le,
                uint8_t proof_points[256], uint8_t* public_le, unsigned long public_capacity);

/* Same two calls for data that already lives in HBM (what bench.py times for the prove workloads;
 * also the hook for a device-native zkey cache). The five point sections stay owned
 * by the caller and must outlive the handle; d_coef_records = n_coefs 44-byte records of section 4;
 * header_points = alpha1(64) beta1(64) beta2(128) delta1(64) delta2(128), wire format. */
int zkpoa_zkey_load_device(zkpoa_context* ctx, uint64_t n_vars, uint64_t n_public, unsigned log_domain,
                           const void* d_A, const void* d_B1, const void* d_B2, const void* d_C, const void* d_H,
                           const void* d_coef_records, uint64_t n_coefs, const uint8_t header_points[448],
                           zkpoa_zkey** zkey);
/* One rank's shard of a key whose sections are already in HBM (bench.py --gpus N generates only its own ranges of
 * the synthetic key; a device-native zkey cache would hand over what it keeps per GPU). Same arguments, but with
 * world > 1 the buffers hold only what zkpoa_zkey_load_shard / _load_shard_split would upload for this rank:
 * d_A, d_B1, d_B2 the wires [lo, lo + cnt) of sections 5-7 and d_C the same split of section 8, where for n items
 * lo = rank * (n / world) + min(rank, n % world), cnt = n / world + (rank < n % world); d_H that range of section 9,
 * or with split != 0 the cyclic shard H[t * world + rank], t < domain / world (copied: the handle owns its copy);
 * d_coef_records all n_coefs records, or with split != 0 at least those of the constraints c = rank (mod world)
 * (others are ignored). The handle is a shard: zkpoa_prove_partials (+ the split stages) and zkpoa_prove_assemble.
 * `split` is a set of shard flags (below): 1 = ZKPOA_SHARD_SPLIT_CHAIN as before; with ZKPOA_SHARD_BLOCK_CYCLIC(L)
 * d_A, d_B1, d_B2, d_C hold this rank's blocks of 2^L items (block b = rank mod world), concatenated. */
int zkpoa_zkey_load_device_shard(zkpoa_context* ctx, uint64_t n_vars, uint64_t n_public, unsigned log_domain,
                                 uint64_t rank, uint64_t world, int split, const void* d_A, const void* d_B1,
                                 const void* d_B2, const void* d_C, const void* d_H, const void* d_coef_records,
                                 uint64_t n_coefs, const uint8_t header_points[448], zkpoa_zkey** zkey);
/* d_witness: n_vars x 32 B standard form on the device (w[0] = 1). */
int zkpoa_prove_device(zkpoa_context* ctx, const zkpoa_zkey* zkey, const void* d_witness,
                       const uint8_t* r_le, const uint8_t* s_le,
                       uint8_t proof_points[256], uint8_t* public_le, unsigned long public_capacity);

/* ---- one proof sharded over several GPUs (BASELINE.json configs[3]) -------------------
 * The five MSMs shard by contiguous index ranges; one process per GPU owns shard `rank` of `world`:
 *   zkpoa_zkey_load_shard   uploads only that rank's byte range of zkey sections 5-9 (coefficients and
 *                           the H-scalar chain are replicated; see the split chain below for sharding them);
 *   zkpoa_zkey_set_shard    restricts a fully resident key (zkpoa_zkey_load / _load_device) to a shard;
 *   zkpoa_prove_partials    -> partials = A(64) B1(64) B2(128) C(64) H(64): the shard's five MSM results;
 *   (the caller all-gathers the 384 bytes over RCCL and sums component-wise: zkpoa_g1_sum / zkpoa_g2_sum)
 *   zkpoa_prove_assemble    host only: header_points (zkpoa_zkey_header) + summed partials + r, s -> proof.
 * RCCL has no elliptic-curve reduction operator, so "all-reduce of partial sums" = all-gather + local add. */
int zkpoa_zkey_load_shard(zkpoa_context* ctx, const void* zkey_buffer, unsigned long zkey_size,
                          uint64_t rank, uint64_t world, zkpoa_zkey** zkey);
/* Shard flags (zkpoa_zkey_load_shard_ex, and the `split` argument of zkpoa_zkey_load_device_shard):
 *   ZKPOA_SHARD_SPLIT_CHAIN      the H-scalar chain is split over the ranks too (what zkpoa_zkey_load_shard_split does);
 *   ZKPOA_SHARD_BLOCK_CYCLIC(L)  sections 5-8 are dealt out in blocks of 2^L consecutive items (wires for A / B1 / B2,
 *                                section indices for C), block b to rank b mod world, instead of one contiguous range
 *                                per rank: real witnesses cluster (runs of bits, runs of full-width limbs), and equal
 *                                index ranges are then unequal work. Every block is still one contiguous byte range of
 *                                the file. L in 4..24; ZKPOA_SHARD_BLOCK_DEFAULT = blocks of 2^16.
 * The multi-GPU path of groth16_prover_zkey_file (env ZKPOA_DEVICES) loads its shards with both. */
#define ZKPOA_SHARD_SPLIT_CHAIN 0x1
#define ZKPOA_SHARD_BLOCK_CYCLIC(L) (((L) & 0xff) << 8)
#define ZKPOA_SHARD_BLOCK_LOG(flags) ((unsigned)(((flags) >> 8) & 0xff))
#define ZKPOA_SHARD_BLOCK_DEFAULT ZKPOA_SHARD_BLOCK_CYCLIC(16)
int zkpoa_zkey_load_shard_ex(zkpoa_context* ctx, const void* zkey_buffer, unsigned long zkey_size,
                             uint64_t rank, uint64_t world, int flags, zkpoa_zkey** zkey);
int zkpoa_zkey_set_shard(zkpoa_zkey* zkey, uint64_t rank, uint64_t world);
int zkpoa_zkey_header(const zkpoa_zkey* zkey, uint8_t header_points[448]);
int zkpoa_prove_partials(zkpoa_context* ctx, const zkpoa_zkey* zkey, const void* wtns_buffer, unsigned long wtns_size,
                         uint8_t partials[384], uint8_t* public_le, unsigned long public_capacity);
int zkpoa_prove_partials_device(zkpoa_context* ctx, const zkpoa_zkey* zkey, const void* d_witness,
                                uint8_t partials[384]);
int zkpoa_prove_assemble(const uint8_t header_points[448], const uint8_t partial_sums[384],
                         const uint8_t* r_le, const uint8_t* s_le, uint8_t proof_points[256]);

/* ---- the H-scalar chain split over the same ranks (SURVEY.md 8e rows NTT / buildABC / joinABC) -------
 * Replaces, per rank, 1/world of snarkjs groth16_prove.js buildABC1 + 3 x (Fr.ifft, batchApplyKey, Fr.fft)
 * + joinABC. world = G in {2, 4, 8}, G^2 <= domain n, M = n / G, Q = M / G.
 * Rank g owns the constraint rows c = g (mod G) and ends with the H scalars of the odd-coset indices
 * i = g (mod G); its H points are the cyclic shard H[t*G + g], so no scalar ever moves after stage 3.
 * Each transform is a four-step NTT with ONE all-to-all (2 per proof for ifft -> shift -> fft of all three
 * polynomials together):
 *   zkpoa_witness_load       parse a .wtns and upload it into the key's witness buffer (every rank: replicated)
 *   zkpoa_split_stage1       buildABC on the rank's rows + size-M inverse DIF of A, B, C -> d_exchange
 *   [caller: ONE all-to-all with equal splits over the whole buffer (RCCL all_to_all_single)]
 *   zkpoa_split_stage2       twiddle, size-G DFT, coset scale inc^k / n, size-G DFT, twiddle: d_received -> d_exchange
 *   [caller: the same all-to-all again]
 *   zkpoa_split_stage3       size-M forward DIT of A, B, C + joinABC -> H scalars kept on the key handle
 *   zkpoa_prove_partials_device(ctx, key, NULL, partials)   the five MSMs of the shard (H: cyclic shard)
 * d_exchange / d_received: caller-owned device buffers of 3 * M * 32 bytes laid out [G ranks][3 polynomials A, B, C]
 * [Q elements]: the contiguous chunk h (3 * Q * 32 bytes) is what rank h receives, e.g. the tensors handed to RCCL.
 * The three stages ENQUEUE their kernels on the context's lane-0 stream and return (errors of the launch itself are
 * reported; execution errors surface at the next synchronising call): run the collectives on that same stream
 * (zkpoa_context_stream: e.g. torch.cuda.ExternalStream) and no host synchronisation is needed anywhere in the chain,
 * or call zkpoa_context_synchronize before touching the buffers from another stream. The witness MSMs of
 * zkpoa_prove_partials_device run on other lanes and overlap the chain. d_witness NULL = the witness already resident. */
void* zkpoa_context_stream(zkpoa_context* ctx, int lane);   /* hipStream_t of lane 0..5 (NULL on error) */
int zkpoa_context_synchronize(zkpoa_context* ctx);          /* waits for lane 0's stream */
int zkpoa_zkey_load_shard_split(zkpoa_context* ctx, const void* zkey_buffer, unsigned long zkey_size,
                                uint64_t rank, uint64_t world, zkpoa_zkey** zkey);
int zkpoa_zkey_set_shard_split(zkpoa_context* ctx, zkpoa_zkey* zkey, uint64_t rank, uint64_t world);
int zkpoa_witness_load(zkpoa_context* ctx, const zkpoa_zkey* zkey, const void* wtns_buffer, unsigned long wtns_size,
                       uint8_t* public_le, unsigned long public_capacity);
int zkpoa_split_stage1(zkpoa_context* ctx, const zkpoa_zkey* zkey, const void* d_witness, void* d_exchange);
int zkpoa_split_stage2(zkpoa_context* ctx, const zkpoa_zkey* zkey, const void* d_received, void* d_exchange);
int zkpoa_split_stage3(zkpoa_context* ctx, const zkpoa_zkey* zkey, void* d_received);

/* Fixed-base tables for a resident key (zkpoa_msm_table_build applied to sections 9, 8 and the A / B queries, in
 * that order while they fit budget_bytes; 0 = half of the HBM free at the call). Later proves on the handle use
 * them; proofs are bit-identical with and without. groth16_prover_zkey_file's key cache does this by itself the
 * second time a key is used only with env ZKPOA_PRECOMP=eager; by default from zkpoa_idle_work (below: never inside a request). used_bytes (optional) <- HBM taken by the tables.
 * A shard handle (zkpoa_zkey_load_shard*, _load_device_shard) gets the tables of its own ranges -- 1/world of the
 * memory per GPU -- incl. the cyclic H shard of a split handle; a resident key re-pointed with zkpoa_zkey_set_shard*
 * uses its tables only while its range is the whole array they were built from. */
int zkpoa_zkey_precompute(zkpoa_context* ctx, zkpoa_zkey* zkey, uint64_t budget_bytes, uint64_t* used_bytes);

/* The H-MSM scalars of the LAST prove on this key handle (joinABC output, groth16_prove.js; domain x 32 B standard
 * form, or domain / world for a split shard: odd-coset indices i = rank mod world), copied to the host. Parity
 * tests use it to check pi_c at sizes where no CPU transform is affordable (oracle quotient identity). */
int zkpoa_zkey_read_h_scalars(zkpoa_context* ctx, const zkpoa_zkey* zkey, void* out, unsigned long capacity);

/* proof_points / public -> JSON text. style 0 = rapidsnark bytes, 1 = snarkjs bytes
 * (SURVEY.md 8a row a11). Size protocol as groth16_prover. */
int zkpoa_proof_to_json(const uint8_t proof_points[256], int style, char* buffer, unsigned long* size);
int zkpoa_public_to_json(const uint8_t* public_le, unsigned long n_public, int style, char* buffer,
                         unsigned long* size);

/* ---- stages of the path, host buffers in / out (upload + compute + download) -------------- */
/* G1.multiExpAffine / G2.multiExpAffine (snarkjs groth16_prove.js; rapidsnark ParallelMultiexp) */
int zkpoa_msm_g1(zkpoa_context* ctx, const void* bases, const void* scalars, uint64_t n, uint8_t out[64]);
int zkpoa_msm_g2(zkpoa_context* ctx, const void* bases, const void* scalars, uint64_t n, uint8_t out[128]);
/* Fr.fft / Fr.ifft: n = 2^log_n Montgomery-form elements, natural order in and out, in place */
int zkpoa_ntt(zkpoa_context* ctx, void* data, unsigned log_n, int inverse);
/* The H-scalar chain: buildABC1 + 3 x (ifft, batchApplyKey(inc), fft) + joinABC. coeffs is the
 * payload of zkey section 4 (u32 nCoefs, then 44-byte records), witness n_vars x 32 B;
 * out = domain_size x 32 B standard form. */
int zkpoa_h_scalars(zkpoa_context* ctx, const void* coeffs, unsigned long coeffs_size, const void* witness,
                    uint64_t n_vars, unsigned log_domain, void* out);

/* ---- same stages on device-resident data (what bench.py times; pointers are HIP device ptrs) */
int zkpoa_msm_g1_device(zkpoa_context* ctx, const void* d_bases, const void* d_scalars, uint64_t n, uint8_t out[64]);
int zkpoa_msm_g2_device(zkpoa_context* ctx, const void* d_bases, const void* d_scalars, uint64_t n, uint8_t out[128]);
int zkpoa_ntt_device(zkpoa_context* ctx, void* d_data, unsigned log_n, int inverse);
/* G1 MSM on an explicit lane (0..5: an independent HIP stream + workspace each): lets a caller keep
 * several MSMs in flight from several host threads, as zkpoa_prove does internally with its five MSMs.
 * Calls on the same lane must not overlap. zkpoa_last_ms_lane: id 0 = whole MSM (ms), 1 = its accumulation kernel (ms),
 * 2 = that kernel's mixed additions in millions (the non-zero digits of the scalars: the work the VALU roofline counts). */
int zkpoa_msm_g1_device_lane(zkpoa_context* ctx, int lane, const void* d_bases, const void* d_scalars, uint64_t n,
                             uint8_t out[64]);
float zkpoa_last_ms_lane(const zkpoa_context* ctx, int lane, int id);

/* Fixed-base form of the same MSMs for bases that stay resident (a proving key's sections 5-9 never change, and
 * 288 GB of HBM has room): zkpoa_msm_table_build stores 2^(c*j) * P_i for every window j of width c = window_bits
 * (0 = cost model) once, window-major, in the zkey wire format; zkpoa_msm_table_run_lane then computes
 * sum k_i * P_i for n scalars with ALL windows sharing one bucket set (fewer, wider windows: ~12-15 % fewer group
 * additions, 1/W of the bucket-reduction work). Results are bit-identical to zkpoa_msm_g1/g2_device on the
 * original bases. group: 1 = G1 (64-B points, out 64 B), 2 = G2 (128 B). info: out = {n, window_bits, windows, bytes}.
 * The prover builds the same tables for a resident key: zkpoa_zkey_precompute. */
typedef struct zkpoa_msm_table zkpoa_msm_table;
int zkpoa_msm_table_build(zkpoa_context* ctx, int group, const void* d_bases, uint64_t n, int window_bits,
                          zkpoa_msm_table** table);
void zkpoa_msm_table_free(zkpoa_context* ctx, zkpoa_msm_table* table);
int zkpoa_msm_table_info(const zkpoa_msm_table* table, uint64_t out[4]);
int zkpoa_msm_table_run_lane(zkpoa_context* ctx, int lane, const zkpoa_msm_table* table, const void* d_scalars,
                             uint8_t* out);

/* Synthetic bases for benchmarks at sizes where no CPU generator is affordable:
 * P_i = (a + i*b) * G, i in [i0, i0+n), written affine/Montgomery into d_out (n*64 or n*128 B).
 * a_le, b_le: 32-byte LE standard-form scalars. Known discrete logs make an O(n) field-only
 * check of any MSM result possible. */
int zkpoa_gen_bases_g1_device(zkpoa_context* ctx, const uint8_t a_le[32], const uint8_t b_le[32], uint64_t i0,
                              uint64_t n, void* d_out);
int zkpoa_gen_bases_g2_device(zkpoa_context* ctx, const uint8_t a_le[32], const uint8_t b_le[32], uint64_t i0,
                              uint64_t n, void* d_out);

/* Sum of affine points (the "all-reduce of partial sums" step of a sharded MSM, done on the host:
 * SURVEY.md 8e). in: count x 64/128 B, out: 64/128 B. No GPU needed. */
int zkpoa_g1_sum(const void* points, uint64_t count, uint8_t out[64]);
int zkpoa_g2_sum(const void* points, uint64_t count, uint8_t out[128]);
/* k * P on the host (used to check known-dlog MSM results). */
int zkpoa_g1_mul(const uint8_t point[64], const uint8_t scalar_le[32], uint8_t out[64]);
int zkpoa_g2_mul(const uint8_t point[128], const uint8_t scalar_le[32], uint8_t out[128]);

/* ---- phase-2 setup arithmetic: the point sections of `snarkjs zkey new`, scripts/g16_setup.sh:243-246 ----
 * out[s] = sum over the entries e with signal[e] == s of coefs[e] * points[point_index[e]], s < n_signals: one call per
 * zkey section -- A (section 5): the A-matrix coefficients over the Lagrange-form tau*G1 points of the prepared .ptau;
 * B1 / B2 (6, 7): the B matrix over tau*G1 resp. tau*G2; IC + C (3, 8): A over beta*tau*G1, B over alpha*tau*G1 and
 * C over tau*G1 in one call (concatenate the three point arrays and offset the indices). Everything on the device:
 * points in the zkey / ptau wire format (affine Montgomery, 64 B G1, 128 B G2; group = 1 | 2), coefs nnz x 32 B
 * little-endian standard form (< r, as an .r1cs file stores them), indices u32; out n_signals points, wire format,
 * signals without an entry = the point at infinity (all zero). Entries may come in any order. */
int zkpoa_setup_accumulate(zkpoa_context* ctx, int group, const void* d_points, uint64_t n_points, const void* d_coefs,
                           const uint32_t* d_point_index, const uint32_t* d_signal, uint64_t nnz, uint64_t n_signals,
                           void* d_out);
/* `snarkjs zkey new <circuit.r1cs> <pot.ptau> <circuit_0.zkey>` (= `snarkjs groth16 setup`; g16_setup.sh:243-252) on
 * files: reads the R1CS (iden3 binary format) and, from a .ptau prepared for phase 2, only the Lagrange-form point
 * ranges of the circuit's domain; computes sections 3 and 5-8 with zkpoa_setup_accumulate, section 9 (H) as the odd
 * points of the size-2n Lagrange basis, section 4 as snarkjs stores it (A and B terms per constraint ascending by
 * signal, then the nPublic + 1 public rows; values scaled by R^2), header with gamma2 = delta2 = the G2 generator
 * and delta1 = the G1 generator. Section 10 holds a zero circuit hash and no contributions: snarkjs' transcript hash
 * is not restated (nothing in the reference pins it), so the key proves and verifies but `snarkjs zkey verify` would
 * not accept its hash. The `zkpoa-setup` executable takes snarkjs' argument order. Errors: zkpoa_last_error. */
int zkpoa_zkey_new(zkpoa_context* ctx, const char* r1cs_path, const char* ptau_path, const char* zkey_path);
/* For a one-shot command only (zkpoa-setup): on != 0 leaves the large host arrays of zkpoa_zkey_new to the process's
 * exit instead of freeing them before the call returns (giving ~100 GB back page by page takes seconds at the
 * layer-three shape). A long-lived caller leaves this off. */
void zkpoa_setup_defer_host_frees(int on);
/* The arithmetic of `snarkjs zkey contribute <in.zkey> <out.zkey>` (g16_setup.sh:262-266): with a secret d (delta_le:
 * 32 B little-endian in [1, r); NULL = drawn from /dev/urandom), delta1, delta2 <- d * delta1, d * delta2 and every
 * point of sections 8 (C) and 9 (H) <- (1/d) * point (device; one scalar for all points, so no lane diverges). All
 * other sections are copied. NOT produced: the contribution record of section 10 (public key of d, proof of
 * knowledge, transcript hash) -- snarkjs' transcript is not restated, so the result proves and verifies under its new
 * verification key but carries no publicly checkable trail; use snarkjs where that trail is the point. */
int zkpoa_zkey_contribute(zkpoa_context* ctx, const char* zkey_in_path, const char* zkey_out_path, const uint8_t* delta_le);
/* `snarkjs wtns check <circuit.r1cs> <witness.wtns>` (scripts/g16_verify.sh:205-210): every constraint's
 * <A, w> * <B, w> == <C, w> on the device. *violated <- how many constraints fail (0 = "WITNESS IS CORRECT"),
 * *first_violated (optional) <- the smallest failing constraint index. PROVER_ERROR for malformed files. */
int zkpoa_wtns_check(zkpoa_context* ctx, const char* r1cs_path, const char* wtns_path, uint64_t* violated,
                     uint64_t* first_violated);

/* ---- the step after the path; host only, no GPU ----------------------------------------
 * zkpoa_groth16_verify: `npx snarkjs groth16 verify <vkey> <public> <proof>` (scripts/g16_verify.sh:213-216)
 *   on the three JSON texts. Returns PROVER_OK (valid), ZKPOA_VERIFY_INVALID_PROOF, or PROVER_ERROR (malformed).
 * zkpoa_sanitize_proof: `python scripts/sanitize_groth16_proof.py` (sanitize_groth16_proof.py:39-124): the
 *   text of sanitized_proof.json (43-bit x 6 limbs; e(-alpha1, beta2) as negalfa1xbeta2), byte-identical to the
 *   reference's output. Size protocol as groth16_prover. */
#define ZKPOA_VERIFY_INVALID_PROOF 0x10
int zkpoa_groth16_verify(const char* vkey_json, const char* public_json, const char* proof_json,
                         char* error_msg, unsigned long error_msg_maxsize);
int zkpoa_sanitize_proof(const char* vkey_json, const char* public_json, const char* proof_json,
                         char* buffer, unsigned long* size, char* error_msg, unsigned long error_msg_maxsize);
/* The same verification on wire-format points, no JSON. vkey_points = alpha1(64) beta2(128) gamma2(128) delta2(128)
 * then IC[(n_public+1) x 64] -- exactly the verification key a .zkey carries in sections 2 and 3 (what
 * `snarkjs zkey export verificationkey`, scripts/g16_setup.sh:287-293, turns into <circuit>_vkey.json);
 * proof_points as zkpoa_prove returns them; public_le = n_public x 32 B standard form. Same return codes.
 * zkpoa_zkey_vkey copies that key out of a resident proving key (size protocol as groth16_prover; PROVER_ERROR
 * when the handle has none: zkpoa_zkey_load_device keys). */
int zkpoa_groth16_verify_points(const uint8_t* vkey_points, unsigned long vkey_size, const uint8_t proof_points[256],
                                const uint8_t* public_le, unsigned long n_public, char* error_msg,
                                unsigned long error_msg_maxsize);
int zkpoa_zkey_vkey(const zkpoa_zkey* zkey, uint8_t* buffer, unsigned long* size);
/* `snarkjs zkey export verificationkey <zkey> <vkey.json>` (scripts/g16_setup.sh:287-293), host only: the text of
 * <circuit>_vkey.json from the zkey image's sections 1-3, byte for byte what snarkjs writes (1-space indent,
 * vk_alphabeta_12 included); pinned on the reference's committed *_vkey.json. Size protocol as groth16_prover.
 * CLI: `zkpoa-verify --export-vkey <zkey> <vkey.json>`. */
int zkpoa_zkey_export_vkey(const void* zkey_buffer, unsigned long zkey_size, char* buffer, unsigned long* size,
                           char* error_msg, unsigned long error_msg_maxsize);

/* ---- the anonymity-set Merkle tree; GPU ------------------------------------------------------
 * Stands in for the reference's Rust binary `merkle-tree` (scripts/merkle_tree.rs, run at
 * scripts/full_workflow.sh:371-380): circomlib Poseidon(2) over Fr (light-poseidon `new_circom(2)`), leaf =
 * H(address, balance), zero-valued leaves up to the next power of two, node = H(left, right).
 * All field elements: 32 B little-endian standard form (the Rust code's big-endian byte arrays reversed).
 * zkpoa_poseidon2: n independent hashes out[i] = H(left[i], right[i]).
 * zkpoa_merkle_build: the whole tree for n (address, balance) pairs, kept in HBM; info: out = {n, path length
 * (= rs_merkle depth() - 1), nodes}; root; stored leaves (hashes; zero for padding) for locating owned addresses
 * (merkle_tree.rs:329-346); path: the sibling hashes of a leaf from the leaves up + the index bit per level, i.e. one
 * entry of merkle_proofs.json's path_elements / path_indices (merkle_tree.rs:354-376).
 * zkpoa_poseidon_params (host only, no GPU): the hash parameters as this library generates them (Grain LFSR), standard
 * form: 195 round constants, then the 3 x 3 MDS matrix row-major. */
typedef struct zkpoa_merkle zkpoa_merkle;
int zkpoa_poseidon_params(uint8_t out[204 * 32]);
int zkpoa_poseidon2(zkpoa_context* ctx, const void* left, const void* right, uint64_t n, void* out);
int zkpoa_poseidon2_device(zkpoa_context* ctx, const void* d_left, const void* d_right, uint64_t n, void* d_out);
int zkpoa_merkle_build(zkpoa_context* ctx, const void* addresses, const void* balances, uint64_t n, zkpoa_merkle** tree);
int zkpoa_merkle_build_device(zkpoa_context* ctx, const void* d_addresses, const void* d_balances, uint64_t n,
                              zkpoa_merkle** tree);
void zkpoa_merkle_free(zkpoa_context* ctx, zkpoa_merkle* tree);
int zkpoa_merkle_info(const zkpoa_merkle* tree, uint64_t out[3]);
int zkpoa_merkle_root(zkpoa_context* ctx, const zkpoa_merkle* tree, uint8_t root_le[32]);
int zkpoa_merkle_leaves(zkpoa_context* ctx, const zkpoa_merkle* tree, uint64_t first, uint64_t count, void* out);
int zkpoa_merkle_path(zkpoa_context* ctx, const zkpoa_merkle* tree, uint64_t leaf_index, uint8_t* path_le,
                      uint8_t* path_indices);

/* ---- measurement -------------------------------------------------------------------------- */
/* Timings (ms, HIP events on the stream that ran the kernels) of the last call on this context.
 * id: 0 = whole device part of last MSM, 1 = its bucket-accumulation kernel (dominant kernel),
 *     2 = last NTT, 3 = last prove: ABC+NTT chain, 4 = last prove: all MSMs, 5 = last prove: total,
 *     6 = last prove: self-check (host pairing check; 0 when it did not run), 7 = last Merkle tree build.
 * Also: tuning knobs. key "msm_c" forces the Pippenger window (0 = auto); key "msm_max_points" sets the
 * number of points one bucket sort may take (0 = 2^27; larger MSMs run in chunks -- tests force small values);
 * key "prove_serial" = 1 runs the stages of a prove one after the other (solo device times for the roofline);
 * key "lane_workspace_max_mb" caps the workspace of every MSM lane (0 = none).
 * Memory pressure: an MSM's workspace grows with the points it sorts at once (~0.4 GB per million at 2^26). When a
 * lane's workspace does not fit -- HBM full (a 2^27 key, another process on the card) or over the cap above -- the MSM
 * is not failed: it goes through its points in pieces of half the size (again halved if need be, down to 2^16), the
 * context remembers the size that fitted, and the result is the same point. zkpoa_msm_points_limit() returns the
 * number of points one MSM of this context sorts at once right now (2^27 until something did not fit). */
float zkpoa_last_ms(const zkpoa_context* ctx, int id);
int zkpoa_set_option(zkpoa_context* ctx, const char* key, long value);
uint64_t zkpoa_msm_points_limit(const zkpoa_context* ctx);

/* ---- element-wise device hooks used by the parity tests ----------------------------------- */
/* field: 0 = Fq, 1 = Fr. op: 0 = Montgomery mul, 1 = add, 2 = sub, 3 = inverse (b ignored),
 * 4 = to Montgomery, 5 = from Montgomery. a, b, out: n x 32 B host buffers. */
int zkpoa_field_op(zkpoa_context* ctx, int field, int op, const void* a, const void* b, void* out, uint64_t n);
/* group: 1 = G1, 2 = G2. out[i] = a[i] + b[i] (affine in, affine out, all exceptional cases). */
int zkpoa_group_add(zkpoa_context* ctx, int group, const void* a, const void* b, void* out, uint64_t n);

#ifdef __cplusplus
}
#endif
#endif /* ZKPOA_PROVER_H */
