// `prover` -- drop-in for the executable the reference execs at scripts/g16_prove.sh:248-252:
//     prover <circuit.zkey> <witness.wtns> <proof.json> <public.json>
// (rapidsnark's argv; g16_prove.sh:195-199 insists the file is named exactly `prover`).
// Exit status 0 on success; non-zero with a message on stderr otherwise, so the reference's
// `set -eE` / ERR trap (scripts/lib/error_handling.sh:14-41) fires. Outputs are written to a
// temporary name and renamed, so a failed run never leaves a partial proof.json.
#include "../../include/zkpoa_prover.h"

#include <fcntl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <string>
#include <vector>

static bool read_file(const char* path, std::vector<char>& out) {
  FILE* f = fopen(path, "rb");
  if (!f) return false;
  struct stat sb;
  if (fstat(fileno(f), &sb) != 0) {
    fclose(f);
    return false;
  }
  out.resize((size_t)sb.st_size);
  size_t got = out.empty() ? 0 : fread(out.data(), 1, out.size(), f);
  fclose(f);
  return got == out.size();
}

static bool write_atomic(const char* path, const char* text) {
  std::string tmp = std::string(path) + ".tmp." + std::to_string((long)getpid());
  FILE* f = fopen(tmp.c_str(), "wb");
  if (!f) return false;
  size_t len = strlen(text);
  bool ok = fwrite(text, 1, len, f) == len;
  ok = (fclose(f) == 0) && ok;
  if (!ok || rename(tmp.c_str(), path) != 0) {
    unlink(tmp.c_str());
    return false;
  }
  return true;
}

static double now_ms() {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec * 1e3 + ts.tv_nsec / 1e6;
}

int main(int argc, char** argv) {
  const double t_start = now_ms();
  if (argc != 5) {
    fprintf(stderr, "Invalid number of parameters\nUsage: prover <circuit.zkey> <witness.wtns> <proof.json> <public.json>\n");
    return EXIT_FAILURE;
  }
  std::vector<char> wtns;
  if (!read_file(argv[2], wtns)) {
    fprintf(stderr, "Error: cannot read witness file %s\n", argv[2]);
    return EXIT_FAILURE;
  }
  unsigned long proof_size = 1 << 12, public_size = 1 << 16;
  std::vector<char> proof(proof_size), pub(public_size);
  char err[1024] = {0};
  int rc = groth16_prover_zkey_file(argv[1], wtns.data(), wtns.size(), proof.data(), &proof_size, pub.data(),
                                    &public_size, err, sizeof(err));
  if (rc == PROVER_ERROR_SHORT_BUFFER) {
    proof.resize(proof_size);
    pub.resize(public_size);
    rc = groth16_prover_zkey_file(argv[1], wtns.data(), wtns.size(), proof.data(), &proof_size, pub.data(),
                                  &public_size, err, sizeof(err));
  }
  if (rc != PROVER_OK) {
    fprintf(stderr, "Error: %s\n", err);
    return EXIT_FAILURE;
  }
  if (!write_atomic(argv[3], proof.data()) || !write_atomic(argv[4], pub.data())) {
    fprintf(stderr, "Error: cannot write %s / %s\n", argv[3], argv[4]);
    return EXIT_FAILURE;
  }
  if (getenv("ZKPOA_VERBOSE")) fprintf(stderr, "zkpoa: prover process total %.1f ms\n", now_ms() - t_start);
  // Both outputs are complete and renamed into place: leave without running the HIP runtime's teardown
  // (freeing tens of GB of device memory and its queues costs ~0.25 s that a one-shot prover never gets back).
  fflush(stdout);
  fflush(stderr);
  _exit(EXIT_SUCCESS);
}
