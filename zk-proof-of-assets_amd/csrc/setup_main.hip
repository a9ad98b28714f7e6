// `zkpoa-setup` -- the GPU stand-in for the reference's key-generation command (scripts/g16_setup.sh:243-252):
//     snarkjs zkey new      <circuit.r1cs> <pot.ptau> <circuit_0.zkey>
//     snarkjs groth16 setup <circuit.r1cs> <pot.ptau> <circuit_0.zkey>
// Same three file arguments (the words `zkey new` / `groth16 setup` are accepted and ignored, so the command line
// can be kept as it is with the executable swapped). The .ptau must be prepared for phase 2 (`snarkjs powersoftau
// prepare phase2`), as snarkjs requires too. Exit status 0 / non-zero + message on stderr.
#include "../../include/zkpoa_prover.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

int main(int argc, char** argv) {
  int a = 1;
  if (argc - a >= 2 && ((!strcmp(argv[a], "zkey") && !strcmp(argv[a + 1], "new")) ||
                        (!strcmp(argv[a], "groth16") && !strcmp(argv[a + 1], "setup"))))
    a += 2;
  if (argc - a != 3) {
    fprintf(stderr, "usage: zkpoa-setup [zkey new | groth16 setup] <circuit.r1cs> <pot.ptau> <circuit_0.zkey>\n");
    return 2;
  }
  struct timespec t0, t1;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  zkpoa_context* ctx = nullptr;
  char err[512] = {0};
  if (zkpoa_context_create(getenv("ZKPOA_DEVICE") ? atoi(getenv("ZKPOA_DEVICE")) : 0, &ctx, err, sizeof err) != PROVER_OK) {
    fprintf(stderr, "zkpoa-setup: %s\n", err);
    return 1;
  }
  int rc = zkpoa_zkey_new(ctx, argv[a], argv[a + 1], argv[a + 2]);
  if (rc != PROVER_OK) fprintf(stderr, "zkpoa-setup: %s\n", zkpoa_last_error(ctx));
  zkpoa_context_destroy(ctx);
  clock_gettime(CLOCK_MONOTONIC, &t1);
  if (rc == PROVER_OK && getenv("ZKPOA_VERBOSE"))
    fprintf(stderr, "zkpoa-setup: %s written in %.2f s\n", argv[a + 2],
            (t1.tv_sec - t0.tv_sec) + (t1.tv_nsec - t0.tv_nsec) / 1e9);
  return rc == PROVER_OK ? 0 : 1;
}
