// `zkpoa-setup` -- the GPU stand-in for the reference's key-generation command (scripts/g16_setup.sh:243-252):
//     snarkjs zkey new      <circuit.r1cs> <pot.ptau> <circuit_0.zkey>
//     snarkjs groth16 setup <circuit.r1cs> <pot.ptau> <circuit_0.zkey>
//     snarkjs zkey contribute <circuit_0.zkey> <circuit_final.zkey> --name="..." -e="..."     (:262-266; arithmetic only)
//     snarkjs wtns check <circuit.r1cs> <witness.wtns>                                      (scripts/g16_verify.sh:205-210)
// Same three file arguments (the words `zkey new` / `groth16 setup` are accepted and ignored, so the command line
// can be kept as it is with the executable swapped). The .ptau must be prepared for phase 2 (`snarkjs powersoftau
// prepare phase2`), as snarkjs requires too. Exit status 0 / non-zero + message on stderr.
#include "../../include/zkpoa_prover.h"

#include "worker_exit.hpp"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

static bool parse_decimal_or_hex(const char* s, uint8_t out[32]) {   // ZKPOA_DELTA: decimal, or 0x... hex; < 2^256
  memset(out, 0, 32);
  if (!s || !*s) return false;
  const bool hex = s[0] == '0' && (s[1] == 'x' || s[1] == 'X');
  for (const char* p = hex ? s + 2 : s; *p; p++) {
    unsigned d;
    if (*p >= '0' && *p <= '9') d = (unsigned)(*p - '0');
    else if (hex && *p >= 'a' && *p <= 'f') d = (unsigned)(*p - 'a' + 10);
    else if (hex && *p >= 'A' && *p <= 'F') d = (unsigned)(*p - 'A' + 10);
    else return false;
    unsigned carry = d;
    for (int i = 0; i < 32; i++) {
      unsigned v = out[i] * (hex ? 16u : 10u) + carry;
      out[i] = (uint8_t)v;
      carry = v >> 8;
    }
    if (carry) return false;
  }
  return true;
}

int main(int argc, char** argv) {
  int a = 1;
  bool contribute = false, check = false;
  if (argc - a >= 2 && !strcmp(argv[a], "zkey") && !strcmp(argv[a + 1], "contribute")) {
    contribute = true;
    a += 2;
  } else if (argc - a >= 2 && !strcmp(argv[a], "wtns") && !strcmp(argv[a + 1], "check")) {
    check = true;
    a += 2;
  } else if (argc - a >= 2 && ((!strcmp(argv[a], "zkey") && !strcmp(argv[a + 1], "new")) ||
                               (!strcmp(argv[a], "groth16") && !strcmp(argv[a + 1], "setup")))) {
    a += 2;
  }
  // snarkjs' options (--name=..., -e=..., -n=..., -v) are accepted and ignored: the name and the entropy text only feed
  // the contribution record and snarkjs' own random generator; the secret here comes from /dev/urandom (or ZKPOA_DELTA)
  const char* pos[3] = {nullptr, nullptr, nullptr};
  int npos = 0;
  for (int i = a; i < argc; i++) {
    if (argv[i][0] == '-' && argv[i][1]) continue;
    if (npos < 3) pos[npos] = argv[i];
    npos++;
  }
  if (npos != (contribute || check ? 2 : 3)) {
    fprintf(stderr, "usage: zkpoa-setup [zkey new | groth16 setup] <circuit.r1cs> <pot.ptau> <circuit_0.zkey>\n"
                    "       zkpoa-setup zkey contribute <in.zkey> <out.zkey> [--name=...] [-e=...]\n"
                    "       zkpoa-setup wtns check <circuit.r1cs> <witness.wtns>\n");
    return 2;
  }
  uint8_t delta[32];
  const uint8_t* delta_p = nullptr;
  if (contribute && getenv("ZKPOA_DELTA")) {   // tests / reproducible keys only: the secret must not be kept
    if (!parse_decimal_or_hex(getenv("ZKPOA_DELTA"), delta)) {
      fprintf(stderr, "zkpoa-setup: ZKPOA_DELTA is not a number below 2^256\n");
      return 2;
    }
    fprintf(stderr, "zkpoa-setup: WARNING: delta taken from ZKPOA_DELTA -- whoever knows it can forge proofs for this key\n");
    delta_p = delta;
  }
  // `zkey new` / `zkey contribute` run in a worker process and this one leaves as soon as the key is renamed into place
  // (csrc/worker_exit.hpp: a worker that has held ~100 GB of host arrays takes seconds to be dismantled).
  zkpoa::WorkerExit we = zkpoa::WorkerExit::start(!check, "zkpoa-setup");
  if (we.is_worker()) zkpoa_setup_defer_host_frees(1);
  auto leave = [&](int code) -> int {
    if (we.is_worker()) we.leave(code);
    return code;
  };
  struct timespec t0, t1;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  zkpoa_context* ctx = nullptr;
  char err[512] = {0};
  if (zkpoa_context_create(getenv("ZKPOA_DEVICE") ? atoi(getenv("ZKPOA_DEVICE")) : 0, &ctx, err, sizeof err) != PROVER_OK) {
    fprintf(stderr, "zkpoa-setup: %s\n", err);
    return leave(1);
  }
  int rc;
  if (check) {   // snarkjs prints "WITNESS IS CORRECT" and exits 0, or names the failure and exits 1
    uint64_t bad = 0, first = 0;
    rc = zkpoa_wtns_check(ctx, pos[0], pos[1], &bad, &first);
    if (rc == PROVER_OK && bad == 0) printf("[INFO]  zkpoa: WITNESS IS CORRECT\n");
    if (rc == PROVER_OK && bad) {
      fprintf(stderr, "[ERROR] zkpoa: WITNESS CHECK FAILED: %llu constraint(s) do not hold, the first is #%llu\n",
              (unsigned long long)bad, (unsigned long long)first);
      zkpoa_context_destroy(ctx);
      return 1;
    }
  } else {
    rc = contribute ? zkpoa_zkey_contribute(ctx, pos[0], pos[1], delta_p) : zkpoa_zkey_new(ctx, pos[0], pos[1], pos[2]);
  }
  if (rc != PROVER_OK) fprintf(stderr, "zkpoa-setup: %s\n", zkpoa_last_error(ctx));
  if (!we.is_worker()) zkpoa_context_destroy(ctx);
  clock_gettime(CLOCK_MONOTONIC, &t1);
  if (rc == PROVER_OK && getenv("ZKPOA_VERBOSE") && !check)
    fprintf(stderr, "zkpoa-setup: %s written in %.2f s\n", pos[contribute ? 1 : 2],
            (t1.tv_sec - t0.tv_sec) + (t1.tv_nsec - t0.tv_nsec) / 1e9);
  return leave(rc == PROVER_OK ? 0 : 1);
}
