// `merkle-tree` -- drop-in for the reference's Rust binary of the same name (scripts/merkle_tree.rs, Cargo.toml:18-20),
// exec'd at scripts/full_workflow.sh:371-380 as
//     merkle-tree --anon-set <anonymity_set.csv> --poa-input-data <parsed_sigs.json> --output-dir <dir>
// Writes <dir>/merkle_root.json and <dir>/merkle_proofs.json in the shapes merkle_tree.rs serialises (serde_json
// pretty printing: 2-space indent): the Poseidon Merkle root of the anonymity set and, for every owned address of the
// parsed-signatures file (in file order, located by a forward scan like merkle_tree.rs:329-346), its leaf, sibling
// path and index bits. The hashing runs on the GPU (libzkpoa_prover.so: csrc/poseidon.hip); the reference's own note
// says 2.5 hours for a 10 M set on the CPU (merkle_tree.rs:3-5). Exit status 0 / non-zero + message, like the prover.
#include "../../include/zkpoa_prover.h"
#include "mini_json.hpp"

#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <unistd.h>
#include <string.h>

#include <fstream>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

using zkpoa::json::JVal;
typedef unsigned __int128 u128;

static const uint64_t kR[4] = {0x43e1f593f0000001ull, 0x2833e84879b97091ull, 0xb85045b68181585dull, 0x30644e72e131a029ull};

struct U256 {
  uint64_t v[4] = {0, 0, 0, 0};
  bool mul_add(unsigned mul, unsigned add) {   // v = v * mul + add; false on overflow
    u128 carry = add;
    for (int i = 0; i < 4; i++) {
      u128 cur = (u128)v[i] * mul + carry;
      v[i] = (uint64_t)cur;
      carry = cur >> 64;
    }
    return carry == 0;
  }
  bool below_r() const {
    for (int i = 3; i >= 0; i--) {
      if (v[i] < kR[i]) return true;
      if (v[i] > kR[i]) return false;
    }
    return false;
  }
  std::string dec() const {
    uint64_t t[4] = {v[0], v[1], v[2], v[3]};
    std::string rev;
    while (t[0] | t[1] | t[2] | t[3]) {
      u128 rem = 0;
      for (int k = 3; k >= 0; k--) {
        u128 cur = (rem << 64) | t[k];
        t[k] = (uint64_t)(cur / 10);
        rem = cur % 10;
      }
      rev.push_back((char)('0' + (int)rem));
    }
    return rev.empty() ? "0" : std::string(rev.rbegin(), rev.rend());
  }
};

static U256 parse_num(const std::string& s, unsigned base, const char* what) {
  U256 x;
  if (s.empty()) throw std::runtime_error(std::string("empty ") + what);
  for (char ch : s) {
    unsigned d;
    if (ch >= '0' && ch <= '9') d = (unsigned)(ch - '0');
    else if (base == 16 && ch >= 'a' && ch <= 'f') d = (unsigned)(ch - 'a' + 10);
    else if (base == 16 && ch >= 'A' && ch <= 'F') d = (unsigned)(ch - 'A' + 10);
    else throw std::runtime_error(std::string("malformed ") + what + ": " + s);
    if (!x.mul_add(base, d)) throw std::runtime_error(std::string(what) + " does not fit 256 bits: " + s);
  }
  // light-poseidon's hash_bytes_be refuses inputs that are not field elements (the Rust binary would panic)
  if (!x.below_r()) throw std::runtime_error(std::string(what) + " is not below the BN254 scalar modulus: " + s);
  return x;
}

static std::string trim(const std::string& s) {
  size_t a = s.find_first_not_of(" \t\r\n\""), b = s.find_last_not_of(" \t\r\n\"");
  return a == std::string::npos ? "" : s.substr(a, b - a + 1);
}

static std::string read_file(const std::string& path) {
  std::ifstream f(path, std::ios::binary);
  if (!f) throw std::runtime_error("cannot open " + path);
  std::stringstream ss;
  ss << f.rdbuf();
  return ss.str();
}

static void write_file(const std::string& path, const std::string& text) {
  std::string tmp = path + ".tmp." + std::to_string((long)getpid());
  FILE* f = fopen(tmp.c_str(), "wb");
  if (!f || fwrite(text.data(), 1, text.size(), f) != text.size() || fclose(f) != 0 || rename(tmp.c_str(), path.c_str()) != 0)
    throw std::runtime_error("cannot write " + path);
}

#define ZK_CALL(expr, what)                                                                       \
  do {                                                                                            \
    if ((expr) != PROVER_OK) throw std::runtime_error(std::string(what) + ": " + zkpoa_last_error(ctx)); \
  } while (0)

int main(int argc, char** argv) {
  std::string anon, poa, outdir;
  for (int i = 1; i < argc; i++) {
    std::string a = argv[i];
    auto val = [&](std::string& dst) {
      size_t eq = a.find('=');
      if (eq != std::string::npos) dst = a.substr(eq + 1);
      else if (i + 1 < argc) dst = argv[++i];
    };
    if (a == "-a" || a.rfind("--anon-set", 0) == 0) val(anon);
    else if (a == "-p" || a.rfind("--poa-input-data", 0) == 0) val(poa);
    else if (a == "-o" || a.rfind("--output-dir", 0) == 0) val(outdir);
    else if (a == "-h" || a == "--help") {
      printf("Construct a Merkle Tree for the anonymity set of Ethereum addresses & balances.\n\n"
             "Usage: merkle-tree --anon-set <FILE_PATH> --poa-input-data <FILE_PATH> --output-dir <DIR_PATH>\n");
      return 0;
    }
  }
  if (anon.empty() || poa.empty() || outdir.empty()) {
    fprintf(stderr, "error: the following required arguments were not provided: --anon-set <FILE_PATH> "
                    "--poa-input-data <FILE_PATH> --output-dir <DIR_PATH>\n");
    return 2;
  }
  zkpoa_context* ctx = nullptr;
  zkpoa_merkle* tree = nullptr;
  int rc = 0;
  try {
    printf("Initiating Merkle Tree build..\n");
    printf("Trying to read given file '\"%s\"'\n", anon.c_str());
    // csv: heading line "address,eth_balance", then 0x-prefixed hex address, decimal balance (merkle_tree.rs:212-246)
    std::vector<uint64_t> addr, bal;
    {
      std::istringstream in(read_file(anon));
      std::string line;
      bool first = true;
      while (std::getline(in, line)) {
        if (trim(line).empty()) continue;
        if (first) { first = false; continue; }
        size_t comma = line.find(',');
        if (comma == std::string::npos) throw std::runtime_error("Failed to find balance in line in csv file: " + line);
        std::string a = trim(line.substr(0, comma)), b = trim(line.substr(comma + 1));
        if (a.size() < 3 || a[0] != '0' || (a[1] != 'x' && a[1] != 'X')) throw std::runtime_error("address is not 0x-prefixed hex: " + a);
        U256 av = parse_num(a.substr(2), 16, "address"), bv = parse_num(b, 10, "balance");
        addr.insert(addr.end(), av.v, av.v + 4);
        bal.insert(bal.end(), bv.v, bv.v + 4);
      }
    }
    const uint64_t n = addr.size() / 4;
    if (n == 0) throw std::runtime_error("the anonymity set is empty");
    printf("Converting lines in '\"%s\"' into leaf nodes.. (leaf node = hash(address, balance))\n", anon.c_str());
    char err[512] = {0};
    int dev = 0;
    if (const char* e = getenv("ZKPOA_DEVICE")) dev = atoi(e);
    if (zkpoa_context_create(dev, &ctx, err, sizeof(err)) != PROVER_OK) throw std::runtime_error(err);
    ZK_CALL(zkpoa_merkle_build(ctx, addr.data(), bal.data(), n, &tree), "merkle tree build");
    uint64_t info[3];
    zkpoa_merkle_info(tree, info);
    printf("Done creating %llu leaves\n", (unsigned long long)n);
    printf("Number of leaves (after adding padding nodes): %llu\n", 1ull << info[1]);
    printf("Creating Merkle tree..\n");
    printf("Done creating Merkle tree of height %llu\n", (unsigned long long)info[1] + 1);   // rs_merkle depth()
    U256 root;
    ZK_CALL(zkpoa_merkle_root(ctx, tree, reinterpret_cast<uint8_t*>(root.v)), "merkle root");
    const std::string root_path = outdir + "/merkle_root.json", proofs_path = outdir + "/merkle_proofs.json";
    write_file(root_path, "{\n  \"__bigint__\": \"" + root.dec() + "\"\n}");
    printf("Root hash %s written to file \"%s\"\n", root.dec().c_str(), root_path.c_str());

    // owned addresses: accountAttestations[].accountData.{address, balance}.__bigint__ (decimal), merkle_tree.rs:296-327
    JVal doc = zkpoa::json::parse_json(read_file(poa).c_str());
    const JVal& atts = doc.at("accountAttestations");
    if (atts.kind != JVal::ARR) throw std::runtime_error("accountAttestations is not an array");
    std::vector<uint64_t> oaddr, obal;
    std::vector<std::string> oaddr_s, obal_s;
    for (const JVal& a : atts.items) {
      const JVal& d = a.at("accountData");
      oaddr_s.push_back(d.at("address").at("__bigint__").scalar());
      obal_s.push_back(d.at("balance").at("__bigint__").scalar());
      U256 av = parse_num(oaddr_s.back(), 10, "address"), bv = parse_num(obal_s.back(), 10, "balance");
      oaddr.insert(oaddr.end(), av.v, av.v + 4);
      obal.insert(obal.end(), bv.v, bv.v + 4);
    }
    const uint64_t m = oaddr_s.size();
    std::vector<uint64_t> ohash(4 * (m ? m : 1));
    if (m) ZK_CALL(zkpoa_poseidon2(ctx, oaddr.data(), obal.data(), m, ohash.data()), "owned leaf hashes");
    // forward scan of the anonymity set for each owned leaf, in order (merkle_tree.rs:329-346: the owned addresses
    // must appear in the set in the same order, or the Rust binary panics)
    std::vector<uint64_t> leaves(4 * n);
    ZK_CALL(zkpoa_merkle_leaves(ctx, tree, 0, n, leaves.data()), "read leaves");
    std::vector<uint64_t> index(m);
    uint64_t pos = 0;
    for (uint64_t i = 0; i < m; i++) {
      while (pos < n && memcmp(&leaves[4 * pos], &ohash[4 * i], 32) != 0) pos++;
      if (pos >= n)
        throw std::runtime_error("Owned leaf " + oaddr_s[i] + " at index " + std::to_string(i) +
                                 " does not exist in the anonymity set");
      index[i] = pos;
    }
    const unsigned k = (unsigned)info[1];
    std::string js = "{\n  \"leaves\": [";
    for (uint64_t i = 0; i < m; i++) {
      U256 h;
      memcpy(h.v, &ohash[4 * i], 32);
      js += std::string(i ? "," : "") + "\n    {\n      \"address\": {\n        \"__bigint__\": \"" + oaddr_s[i] +
            "\"\n      },\n      \"balance\": {\n        \"__bigint__\": \"" + obal_s[i] +
            "\"\n      },\n      \"hash\": {\n        \"__bigint__\": \"" + h.dec() + "\"\n      }\n    }";
    }
    js += m ? "\n  ],\n  \"path_elements\": [" : "],\n  \"path_elements\": [";
    std::vector<std::vector<uint8_t>> bits(m, std::vector<uint8_t>(k ? k : 1));
    for (uint64_t i = 0; i < m; i++) {
      std::vector<uint64_t> path(4 * (k ? k : 1));
      ZK_CALL(zkpoa_merkle_path(ctx, tree, index[i], reinterpret_cast<uint8_t*>(path.data()), bits[i].data()), "merkle path");
      js += std::string(i ? "," : "") + "\n    [";
      for (unsigned l = 0; l < k; l++) {
        U256 e;
        memcpy(e.v, &path[4 * l], 32);
        js += std::string(l ? "," : "") + "\n      {\n        \"__bigint__\": \"" + e.dec() + "\"\n      }";
      }
      js += k ? "\n    ]" : "]";
    }
    js += m ? "\n  ],\n  \"path_indices\": [" : "],\n  \"path_indices\": [";
    for (uint64_t i = 0; i < m; i++) {
      js += std::string(i ? "," : "") + "\n    [";
      for (unsigned l = 0; l < k; l++) js += std::string(l ? "," : "") + "\n      " + std::to_string((int)bits[i][l]);
      js += k ? "\n    ]" : "]";
    }
    js += m ? "\n  ]\n}" : "]\n}";
    write_file(proofs_path, js);
  } catch (const std::exception& e) {
    fprintf(stderr, "Error: %s\n", e.what());
    rc = 1;
  }
  if (tree) zkpoa_merkle_free(ctx, tree);
  if (ctx) zkpoa_context_destroy(ctx);
  return rc;
}
