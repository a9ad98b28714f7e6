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

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>

#include <exception>
#include <fstream>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <string>
#include <chrono>
#include <thread>
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

static U256 parse_num(const char* b, const char* e, unsigned base, const char* what) {
  U256 x;
  if (b == e) throw std::runtime_error(std::string("empty ") + what);
  for (const char* q = b; q < e; q++) {
    const char ch = *q;
    unsigned d;
    if (ch >= '0' && ch <= '9') d = (unsigned)(ch - '0');
    else if (base == 16 && ch >= 'a' && ch <= 'f') d = (unsigned)(ch - 'a' + 10);
    else if (base == 16 && ch >= 'A' && ch <= 'F') d = (unsigned)(ch - 'A' + 10);
    else throw std::runtime_error(std::string("malformed ") + what + ": " + std::string(b, e));
    if (!x.mul_add(base, d)) throw std::runtime_error(std::string(what) + " does not fit 256 bits: " + std::string(b, e));
  }
  // light-poseidon's hash_bytes_be refuses inputs that are not field elements (the Rust binary would panic)
  if (!x.below_r()) throw std::runtime_error(std::string(what) + " is not below the BN254 scalar modulus: " + std::string(b, e));
  return x;
}

static U256 parse_num(const std::string& s, unsigned base, const char* what) {
  return parse_num(s.data(), s.data() + s.size(), base, what);
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

// ---- the anonymity set: a 10 M-line csv is ~650 MB of text, and a line-at-a-time reader would take longer over it than
// the GPU takes over the whole tree (0.12 s). The file is mapped, cut at line starts into one piece per thread, and every
// piece is parsed twice: lines counted first (so that each piece knows where its rows go), then converted in place.
// Same rules as before: blank lines are skipped, the first non-blank line is the heading, fields are trimmed of blanks
// and double quotes, an address is 0x-prefixed hex, a balance decimal, both below the scalar modulus.
static bool is_trim(char c) { return c == ' ' || c == '\t' || c == '\r' || c == '\n' || c == '"'; }
static void trim_span(const char*& b, const char*& e) {
  while (b < e && is_trim(*b)) b++;
  while (e > b && is_trim(e[-1])) e--;
}
// calls f(line_begin, line_end) for every non-blank line of [b, e) (e at a line start or the end of the file)
template <class Fn>
static void for_lines(const char* b, const char* e, Fn f) {
  while (b < e) {
    const char* nl = static_cast<const char*>(memchr(b, '\n', (size_t)(e - b)));
    const char* le = nl ? nl : e;
    const char *tb = b, *te = le;
    trim_span(tb, te);
    if (tb < te) f(b, le);
    b = nl ? nl + 1 : e;
  }
}
static void parse_row(const char* b, const char* e, uint64_t* addr, uint64_t* bal) {
  const char* comma = static_cast<const char*>(memchr(b, ',', (size_t)(e - b)));
  if (!comma) throw std::runtime_error("Failed to find balance in line in csv file: " + std::string(b, e));
  const char *ab = b, *ae = comma, *bb = comma + 1, *be = e;
  trim_span(ab, ae);
  trim_span(bb, be);
  if (ae - ab < 3 || ab[0] != '0' || (ab[1] != 'x' && ab[1] != 'X'))
    throw std::runtime_error("address is not 0x-prefixed hex: " + std::string(ab, ae));
  const U256 av = parse_num(ab + 2, ae, 16, "address"), bv = parse_num(bb, be, 10, "balance");
  memcpy(addr, av.v, 32);
  memcpy(bal, bv.v, 32);
}
// rows x 4 limbs, NOT zero-filled: at 10 M rows the 640 MB would be touched once by one thread just to be overwritten --
// the parsing threads are the first to touch their own part
struct Limbs {
  std::unique_ptr<uint64_t[]> p;
  uint64_t rows = 0;
  void alloc(uint64_t n) {
    p.reset(new uint64_t[4 * (n ? n : 1)]);
    rows = n;
  }
  uint64_t* data() { return p.get(); }
};
static void read_anonymity_set(const std::string& path, Limbs& addr, Limbs& bal) {
  int fd = open(path.c_str(), O_RDONLY);
  struct stat sb;
  if (fd < 0 || fstat(fd, &sb) != 0) {
    if (fd >= 0) close(fd);
    throw std::runtime_error("cannot open " + path);
  }
  const size_t size = (size_t)sb.st_size;
  if (size == 0) {
    close(fd);
    return;
  }
  void* map = mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
  close(fd);
  if (map == MAP_FAILED) throw std::runtime_error("cannot map " + path);
  const char* base = static_cast<const char*>(map);
  const char* end = base + size;
  try {
    // the heading: the first non-blank line
    const char* body = end;
    {
      const char* b = base;
      while (b < end) {
        const char* nl = static_cast<const char*>(memchr(b, '\n', (size_t)(end - b)));
        const char* le = nl ? nl : end;
        const char *tb = b, *te = le;
        trim_span(tb, te);
        b = nl ? nl + 1 : end;
        if (tb < te) {
          body = b;
          break;
        }
      }
    }
    unsigned T = std::thread::hardware_concurrency();
    if (T > 16) T = 16;
    bool forced = false;
    if (const char* e = getenv("ZKPOA_MERKLE_THREADS")) {
      char* rest = nullptr;
      const long v = strtol(e, &rest, 10);
      if (rest == e || *rest || v < 1 || v > 256)
        throw std::runtime_error(std::string("ZKPOA_MERKLE_THREADS must be 1..256, not '") + e + "'");
      T = (unsigned)v;
      forced = true;
    }
    if (T < 1) T = 1;
    if ((size_t)(end - body) < (1u << 16) && !forced) T = 1;
    std::vector<const char*> cut(T + 1, end);
    cut[0] = body;
    for (unsigned t = 1; t < T; t++) {   // piece boundaries moved forward to the next line start
      const char* p = body + (size_t)(end - body) / T * t;
      if (p < cut[t - 1]) p = cut[t - 1];
      const char* nl = p < end ? static_cast<const char*>(memchr(p, '\n', (size_t)(end - p))) : nullptr;
      cut[t] = nl ? nl + 1 : end;
    }
    std::vector<uint64_t> rows(T, 0);
    std::vector<std::exception_ptr> errs(T);
    auto run = [&](auto fn) {
      std::vector<std::thread> th;
      for (unsigned t = 0; t < T; t++)
        th.emplace_back([&, t] {
          try {
            fn(t);
          } catch (...) {
            errs[t] = std::current_exception();
          }
        });
      for (auto& x : th) x.join();
      for (auto& e : errs)
        if (e) std::rethrow_exception(e);   // the first failing piece in file order
    };
    run([&](unsigned t) { for_lines(cut[t], cut[t + 1], [&](const char*, const char*) { rows[t]++; }); });
    std::vector<uint64_t> first(T + 1, 0);
    for (unsigned t = 0; t < T; t++) first[t + 1] = first[t] + rows[t];
    addr.alloc(first[T]);
    bal.alloc(first[T]);
    run([&](unsigned t) {
      uint64_t r = first[t];
      for_lines(cut[t], cut[t + 1], [&](const char* b, const char* e) {
        parse_row(b, e, addr.data() + 4 * r, bal.data() + 4 * r);
        r++;
      });
    });
  } catch (...) {
    munmap(map, size);
    throw;
  }
  munmap(map, size);
}

#define ZK_CALL(expr, what)                                                                       \
  do {                                                                                            \
    if ((expr) != PROVER_OK) throw std::runtime_error(std::string(what) + ": " + zkpoa_last_error(ctx)); \
  } while (0)

static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv) {
  const bool verbose = getenv("ZKPOA_VERBOSE") && *getenv("ZKPOA_VERBOSE") && strcmp(getenv("ZKPOA_VERBOSE"), "0") != 0;
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
    // the device context (HIP start-up, ~0.4 s) is created on a second thread while this one reads the set
    char err[512] = {0};
    long dev = 0;
    if (const char* e = getenv("ZKPOA_DEVICE")) {
      char* rest = nullptr;
      dev = strtol(e, &rest, 10);
      if (rest == e || *rest || dev < 0 || dev > 1023) throw std::runtime_error(std::string("ZKPOA_DEVICE: not a device index: ") + e);
    }
    int ctx_rc = PROVER_OK;
    std::thread ctx_thread([&] { ctx_rc = zkpoa_context_create((int)dev, &ctx, err, sizeof(err)); });
    Limbs addr, bal;
    const double t0 = now_s();
    try {
      read_anonymity_set(anon, addr, bal);
    } catch (...) {
      ctx_thread.join();
      throw;
    }
    const double t_read = now_s();
    ctx_thread.join();
    const uint64_t n = addr.rows;
    if (n == 0) throw std::runtime_error("the anonymity set is empty");
    printf("Converting lines in '\"%s\"' into leaf nodes.. (leaf node = hash(address, balance))\n", anon.c_str());
    if (ctx_rc != PROVER_OK) throw std::runtime_error(err);
    const double t_ctx = now_s();
    ZK_CALL(zkpoa_merkle_build(ctx, addr.data(), bal.data(), n, &tree), "merkle tree build");
    if (verbose)
      fprintf(stderr, "merkle-tree: read %llu rows %.3f s, waited %.3f s more for the device, leaves + tree (with the upload) %.3f s\n",
              (unsigned long long)n, t_read - t0, t_ctx - t_read, now_s() - t_ctx);
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

    const double t_paths = now_s();
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
    if (verbose) fprintf(stderr, "merkle-tree: %llu paths and both files %.3f s\n", (unsigned long long)m, now_s() - t_paths);
  } catch (const std::exception& e) {
    fprintf(stderr, "Error: %s\n", e.what());
    rc = 1;
  }
  // Both files are complete and renamed into place (or the error is on stderr): leave without the HIP runtime's
  // teardown, as `prover` does -- freeing the tree and the context's queues is ~0.1 s a one-shot command never gets back
  fflush(stdout);
  fflush(stderr);
  _exit(rc);
}
