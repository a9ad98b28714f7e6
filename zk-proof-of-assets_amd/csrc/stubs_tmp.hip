// TEMPORARY: entry points not implemented yet (replaced as ntt.hip / prover.hip land).
#include "zkpoa_internal.hpp"
#define NI(ctx) do { if (ctx) (ctx)->last_error = "not implemented yet"; return PROVER_ERROR; } while (0)
extern "C" int zkpoa_ntt(zkpoa_context* ctx, void*, unsigned, int) { NI(ctx); }
extern "C" int zkpoa_ntt_device(zkpoa_context* ctx, void*, unsigned, int) { NI(ctx); }
extern "C" int zkpoa_h_scalars(zkpoa_context* ctx, const void*, unsigned long, const void*, uint64_t, unsigned, void*) { NI(ctx); }
extern "C" int zkpoa_zkey_load(zkpoa_context* ctx, const void*, unsigned long, zkpoa_zkey**) { NI(ctx); }
extern "C" void zkpoa_zkey_free(zkpoa_context*, zkpoa_zkey*) {}
extern "C" int zkpoa_zkey_info(const zkpoa_zkey*, uint64_t*) { return PROVER_ERROR; }
extern "C" int zkpoa_prove(zkpoa_context* ctx, const zkpoa_zkey*, const void*, unsigned long, const uint8_t*, const uint8_t*, uint8_t*, uint8_t*, unsigned long) { NI(ctx); }
extern "C" int zkpoa_proof_to_json(const uint8_t*, int, char*, unsigned long*) { return PROVER_ERROR; }
extern "C" int zkpoa_public_to_json(const uint8_t*, unsigned long, int, char*, unsigned long*) { return PROVER_ERROR; }
extern "C" int groth16_prover(const void*, unsigned long, const void*, unsigned long, char*, unsigned long*, char*, unsigned long*, char*, unsigned long) { return PROVER_ERROR; }
extern "C" int groth16_prover_zkey_file(const char*, const void*, unsigned long, char*, unsigned long*, char*, unsigned long*, char*, unsigned long) { return PROVER_ERROR; }
