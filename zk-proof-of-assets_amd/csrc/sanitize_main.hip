// `zkpoa-sanitize <proof_dir>` -- native stand-in for `python scripts/sanitize_groth16_proof.py <proof_dir>`
// (called at scripts/full_workflow.sh:518,538): reads proof.json + public.json from the directory, finds
// the `*_vkey.json` there or in its parent (sanitize_groth16_proof.py:139-153), writes
// sanitized_proof.json with byte-identical content.
#include "../../include/zkpoa_prover.h"

#include <dirent.h>
#include <stdio.h>
#include <string.h>

#include <fstream>
#include <sstream>
#include <string>
#include <vector>

static bool slurp(const std::string& path, std::string& out) {
  std::ifstream f(path, std::ios::binary);
  if (!f) return false;
  std::ostringstream ss;
  ss << f.rdbuf();
  out = ss.str();
  return true;
}

static std::string find_vkey(const std::string& dir) {
  std::string found;
  DIR* d = opendir(dir.c_str());
  if (!d) return found;
  while (dirent* e = readdir(d)) {
    std::string name = e->d_name;
    const char* suffix = "_vkey.json";
    if (name.size() >= strlen(suffix) && name.compare(name.size() - strlen(suffix), strlen(suffix), suffix) == 0)
      found = dir + "/" + name;
  }
  closedir(d);
  return found;
}

int main(int argc, char** argv) {
  if (argc != 2) {
    fprintf(stderr, "Usage: zkpoa-sanitize <proof_dir>\n");
    return 2;
  }
  std::string dir = argv[1];
  std::string vkey_path = find_vkey(dir);
  if (vkey_path.empty()) vkey_path = find_vkey(dir + "/..");
  if (vkey_path.empty()) {
    fprintf(stderr, "Error: Cannot find vkey file\n");
    return 1;
  }
  std::string vk, pub, pr;
  if (!slurp(vkey_path, vk) || !slurp(dir + "/public.json", pub) || !slurp(dir + "/proof.json", pr)) {
    fprintf(stderr, "Error: cannot read proof.json / public.json / vkey\n");
    return 1;
  }
  unsigned long size = 1 << 16;
  std::vector<char> buf(size);
  char err[512] = {0};
  int rc = zkpoa_sanitize_proof(vk.c_str(), pub.c_str(), pr.c_str(), buf.data(), &size, err, sizeof(err));
  if (rc == PROVER_ERROR_SHORT_BUFFER) {
    buf.resize(size);
    rc = zkpoa_sanitize_proof(vk.c_str(), pub.c_str(), pr.c_str(), buf.data(), &size, err, sizeof(err));
  }
  if (rc != PROVER_OK) {
    fprintf(stderr, "Error: %s\n", err);
    return 1;
  }
  std::ofstream out(dir + "/sanitized_proof.json", std::ios::binary);
  out << buf.data();
  return out.good() ? 0 : 1;
}
