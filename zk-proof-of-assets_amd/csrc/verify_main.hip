// `zkpoa-verify <vkey.json> <public.json> <proof.json>` -- same three positional arguments and the same
// verdict lines as `npx snarkjs groth16 verify` at scripts/g16_verify.sh:213-216 ("snarkJS: OK!" /
// "snarkJS: Invalid proof"); exit status 0 only for a valid proof.
#include "../../include/zkpoa_prover.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <fstream>
#include <sstream>
#include <string>

static bool slurp(const char* path, std::string& out) {
  std::ifstream f(path, std::ios::binary);
  if (!f) return false;
  std::ostringstream ss;
  ss << f.rdbuf();
  out = ss.str();
  return true;
}

// `zkpoa-verify --export-vkey <circuit.zkey> <vkey.json>`: snarkjs zkey export verificationkey (g16_setup.sh:287-293)
static int export_vkey(const char* zkey_path, const char* out_path) {
  std::string z;
  if (!slurp(zkey_path, z)) {
    fprintf(stderr, "[ERROR] zkpoa-verify: cannot read %s\n", zkey_path);
    return 2;
  }
  unsigned long size = 0;
  char err[512] = {0};
  int rc = zkpoa_zkey_export_vkey(z.data(), (unsigned long)z.size(), nullptr, &size, err, sizeof(err));
  if (rc != PROVER_ERROR_SHORT_BUFFER) {
    fprintf(stderr, "[ERROR] zkpoa-verify: %s\n", err);
    return 2;
  }
  std::string text(size, '\0');
  rc = zkpoa_zkey_export_vkey(z.data(), (unsigned long)z.size(), &text[0], &size, err, sizeof(err));
  if (rc != PROVER_OK) {
    fprintf(stderr, "[ERROR] zkpoa-verify: %s\n", err);
    return 2;
  }
  std::ofstream f(out_path, std::ios::binary);
  f.write(text.c_str(), (std::streamsize)strlen(text.c_str()));
  return f.good() ? 0 : 2;
}

int main(int argc, char** argv) {
  if (argc == 4 && std::string(argv[1]) == "--export-vkey") return export_vkey(argv[2], argv[3]);
  if (argc != 4) {
    fprintf(stderr, "Usage: zkpoa-verify <verification_key.json> <public.json> <proof.json>\n"
                    "       zkpoa-verify --export-vkey <circuit.zkey> <verification_key.json>\n");
    return 2;
  }
  std::string vk, pub, pr;
  if (!slurp(argv[1], vk) || !slurp(argv[2], pub) || !slurp(argv[3], pr)) {
    fprintf(stderr, "[ERROR] zkpoa-verify: cannot read input files\n");
    return 2;
  }
  char err[512] = {0};
  int rc = zkpoa_groth16_verify(vk.c_str(), pub.c_str(), pr.c_str(), err, sizeof(err));
  if (rc == PROVER_OK) {
    printf("[INFO]  snarkJS: OK!\n");
    return 0;
  }
  if (rc == ZKPOA_VERIFY_INVALID_PROOF) {
    printf("[ERROR] snarkJS: Invalid proof\n");
    return 1;
  }
  fprintf(stderr, "[ERROR] zkpoa-verify: %s\n", err);
  return 2;
}
