// Groth16 verification and proof sanitising on the host: what the reference runs
// right after every prove -- `npx snarkjs groth16 verify <vkey> <public> <proof>`
// (scripts/g16_verify.sh:213-216) and `python sanitize_groth16_proof.py <proof_dir>`
// (scripts/sanitize_groth16_proof.py:39-124, 43-bit x 6 limbs per scripts/lib/field_helper.py) --
// as two C-ABI entry points over pairing.hpp. No GPU is needed or used: a verification is three
// Miller loops and one final exponentiation (~15 ms on one core).
#include "pairing.hpp"
#include "../../include/zkpoa_prover.h"
#include "host_curve.hpp"
#include "mini_json.hpp"

#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

using namespace zkpoa;
using namespace zkpoa::pairing;

// (own copy: this translation unit is also linked on its own into zkpoa-verify / zkpoa-sanitize, which must not
// pull in the HIP runtime -- loading it costs 0.1-0.3 s per process, ten times the verification itself)
static void set_err(char* buf, unsigned long cap, const std::string& msg) {
  if (!buf || cap == 0) return;
  size_t k = msg.size() < cap - 1 ? msg.size() : cap - 1;
  memcpy(buf, msg.data(), k);
  buf[k] = 0;
}

namespace {

using zkpoa::json::JVal;
using zkpoa::json::parse_json;

// ---- decimal <-> field ----------------------------------------------------------------------------------
// decimal string -> 256-bit integer (false if not decimal or >= 2^256)
bool parse_u256(const std::string& s, uint64_t out[4]) {
  out[0] = out[1] = out[2] = out[3] = 0;
  if (s.empty()) return false;
  for (char ch : s) {
    if (ch < '0' || ch > '9') return false;
    u128 carry = (unsigned)(ch - '0');
    for (int i = 0; i < 4; i++) {
      u128 cur = (u128)out[i] * 10 + carry;
      out[i] = (uint64_t)cur;
      carry = cur >> 64;
    }
    if (carry) return false;
  }
  return true;
}
bool lt_modulus(const uint64_t v[4], const uint64_t p[4]) {
  for (int i = 3; i >= 0; i--) {
    if (v[i] < p[i]) return true;
    if (v[i] > p[i]) return false;
  }
  return false;
}
HFq fq_from_dec(const std::string& s) {
  uint64_t v[4];
  if (!parse_u256(s, v) || !lt_modulus(v, HFqParams::P)) throw std::runtime_error("coordinate is not a field element: " + s);
  HFq r{{v[0], v[1], v[2], v[3]}};
  return r.to_mont();
}
HFq2 fq2_from_json(const JVal& v) { return HFq2{fq_from_dec(v[0].scalar()), fq_from_dec(v[1].scalar())}; }

// JSON point forms: G1 [x, y, z], G2 [[x0,x1],[y0,y1],[z0,z1]]; z = 0 -> infinity, z = 1 -> affine. For any other
// z the two consumers of these files differ and each entry point follows its own original: the reference's
// sanitizer treats it as homogeneous projective, (x/z, y/z) (scripts/sanitize_groth16_proof.py:44-59), while
// snarkjs' verifier (ffjavascript fromObject, g16_verify.sh:213-216) reads Jacobian coordinates, (x/z^2, y/z^3).
enum class ZRule { Projective, Jacobian };
template <class HF>
Affine<HF> affine_from_xyz(const HF& x, const HF& y, const HF& z, ZRule rule) {
  if (z.is_zero()) return {HF::zero(), HF::zero()};
  HF zi = z.inv();
  if (rule == ZRule::Projective) return {x * zi, y * zi};
  HF zi2 = zi.sqr();
  return {x * zi2, y * zi2 * zi};
}
G1 g1_from_json(const JVal& v, ZRule rule) {
  return affine_from_xyz<HFq>(fq_from_dec(v[0].scalar()), fq_from_dec(v[1].scalar()), fq_from_dec(v[2].scalar()), rule);
}
G2 g2_from_json(const JVal& v, ZRule rule) {
  return affine_from_xyz<HFq2>(fq2_from_json(v[0]), fq2_from_json(v[1]), fq2_from_json(v[2]), rule);
}
G1 g1_neg(const G1& p) { return {p.x, p.y.neg()}; }

struct VKey {
  G1 alpha1;
  G2 beta2, gamma2, delta2;
  std::vector<G1> IC;
  size_t nPublic = 0;
};
VKey vkey_from_json(const JVal& v, ZRule rule) {
  VKey k;
  k.alpha1 = g1_from_json(v.at("vk_alpha_1"), rule);
  k.beta2 = g2_from_json(v.at("vk_beta_2"), rule);
  k.gamma2 = g2_from_json(v.at("vk_gamma_2"), rule);
  k.delta2 = g2_from_json(v.at("vk_delta_2"), rule);
  const JVal& ic = v.at("IC");
  if (ic.kind != JVal::ARR || ic.items.empty()) throw std::runtime_error("vkey: IC is empty");
  for (const auto& p : ic.items) k.IC.push_back(g1_from_json(p, rule));
  // nPublic: a plain non-negative decimal of at most 9 digits (std::stoull would wrap "-1" to 2^64 - 1)
  const std::string& np = v.at("nPublic").scalar();
  if (np.empty() || np.size() > 9) throw std::runtime_error("vkey: nPublic is not a small non-negative integer");
  size_t n = 0;
  for (char ch : np) {
    if (ch < '0' || ch > '9') throw std::runtime_error("vkey: nPublic is not a small non-negative integer");
    n = n * 10 + (size_t)(ch - '0');
  }
  k.nPublic = n;
  if (k.IC.size() != k.nPublic + 1) throw std::runtime_error("vkey: IC length does not match nPublic");
  return k;
}

struct Proof {
  G1 a, c;
  G2 b;
};
Proof proof_from_json(const JVal& v, ZRule rule) {
  return {g1_from_json(v.at("pi_a"), rule), g1_from_json(v.at("pi_c"), rule), g2_from_json(v.at("pi_b"), rule)};
}

G1 g1_add_mul(const G1& acc, const G1& p, const uint64_t k[4]) {
  XYZZ<HFq> r = XYZZ<HFq>::from_affine(acc);
  xyzz_add(r, h_mul(XYZZ<HFq>::from_affine(p), k));
  return h_to_affine(r);
}

// snarkjs groth16 verify: public inputs in the field, proof points on their curves,
// e(-A, B) * e(alpha, beta) * e(vk_x, gamma) * e(C, delta) == 1
// pub: nPublic standard-form scalars (4 x u64 each), already range-checked
bool verify_core(const VKey& vk, const Proof& pr, const uint64_t* pub) {
  if (!g1_on_curve(pr.a) || !g1_on_curve(pr.c) || !g2_on_curve(pr.b)) return false;
  G1 vkx = vk.IC[0];
  for (size_t i = 0; i < vk.nPublic; i++) vkx = g1_add_mul(vkx, vk.IC[i + 1], pub + 4 * i);
  // e(-A, B) e(alpha, beta) e(vk_x, gamma) e(C, delta) == 1: one shared Miller accumulator, one final exponentiation
  const G2 qs[4] = {pr.b, vk.beta2, vk.gamma2, vk.delta2};
  const G1 ps[4] = {g1_neg(pr.a), vk.alpha1, vkx, pr.c};
  return final_exponentiation(multi_miller_loop(qs, ps, 4)).is_one();
}

bool verify_impl(const JVal& vkj, const JVal& pubj, const JVal& prj) {
  VKey vk = vkey_from_json(vkj, ZRule::Jacobian);
  Proof pr = proof_from_json(prj, ZRule::Jacobian);
  if (pubj.kind != JVal::ARR || pubj.items.size() != vk.nPublic) return false;
  std::vector<uint64_t> pub(4 * vk.nPublic + 4);
  for (size_t i = 0; i < vk.nPublic; i++)
    if (!parse_u256(pubj[i].scalar(), &pub[4 * i]) || !lt_modulus(&pub[4 * i], HFrParams::P)) return false;
  return verify_core(vk, pr, pub.data());
}

// ---- sanitizer: 43-bit x 6 limb arrays, Python json.dump formatting ----------------------------------------
std::string u64_dec(uint64_t v) { return std::to_string(v); }
std::string limbs43(const HFq& x) {  // numberToArray(x, 43, 6)
  HFq s = x.from_mont();
  u128 lo = ((u128)s.l[1] << 64) | s.l[0], hi = ((u128)s.l[3] << 64) | s.l[2];
  std::string out = "[";
  for (int i = 0; i < 6; i++) {
    uint64_t limb = (uint64_t)(lo & (((u128)1 << 43) - 1));
    out += u64_dec(limb);
    if (i < 5) out += ", ";
    lo = (lo >> 43) | (hi << (128 - 43));
    hi >>= 43;
  }
  return out + "]";
}
std::string fq2_limbs(const HFq2& x) { return "[" + limbs43(x.c0) + ", " + limbs43(x.c1) + "]"; }
std::string g2_limbs(const G2& p) { return "[" + fq2_limbs(p.x) + ", " + fq2_limbs(p.y) + "]"; }
std::string g1_limbs(const G1& p) { return "[" + limbs43(p.x) + ", " + limbs43(p.y) + "]"; }

std::string sanitize_impl(const JVal& vkj, const JVal& pubj, const JVal& prj) {
  VKey vk = vkey_from_json(vkj, ZRule::Projective);
  Proof pr = proof_from_json(prj, ZRule::Projective);
  Fq12 nab = pairing::pairing(vk.beta2, g1_neg(vk.alpha1));   // pairing(beta, negalpha)
  HFq2 co[6];
  nab.fq2_coeffs(co);
  std::string o = "{\"gamma2\": " + g2_limbs(vk.gamma2) + ", \"delta2\": " + g2_limbs(vk.delta2) + ", \"negalfa1xbeta2\": [";
  for (int i = 0; i < 6; i++) o += fq2_limbs(co[i]) + (i < 5 ? ", " : "");
  o += "], \"IC\": [";
  for (size_t i = 0; i < vk.IC.size(); i++) o += g1_limbs(vk.IC[i]) + (i + 1 < vk.IC.size() ? ", " : "");
  o += "], \"negpa\": " + g1_limbs(g1_neg(pr.a)) + ", \"pb\": " + g2_limbs(pr.b) + ", \"pc\": " + g1_limbs(pr.c);
  o += ", \"pubInput\": [";
  if (pubj.kind != JVal::ARR) throw std::runtime_error("public.json is not an array");
  for (size_t i = 0; i < pubj.items.size(); i++) {
    uint64_t s[4];
    if (!parse_u256(pubj[i].scalar(), s)) throw std::runtime_error("public input is not a decimal integer");
    // int(pubInput): re-emit the canonical decimal (strips leading zeros like Python's int())
    std::string dec;
    {  // decimal of a raw 256-bit integer
      uint64_t v[4] = {s[0], s[1], s[2], s[3]};
      std::string rev;
      while (v[0] | v[1] | v[2] | v[3]) {
        u128 rem = 0;
        for (int k = 3; k >= 0; k--) {
          u128 cur = (rem << 64) | v[k];
          v[k] = (uint64_t)(cur / 10);
          rem = cur % 10;
        }
        rev.push_back((char)('0' + (int)rem));
      }
      dec = rev.empty() ? "0" : std::string(rev.rbegin(), rev.rend());
    }
    o += dec + (i + 1 < pubj.items.size() ? ", " : "");
  }
  o += "]}";
  return o;
}

}  // namespace

extern "C" int zkpoa_groth16_verify(const char* vkey_json, const char* public_json, const char* proof_json,
                                    char* error_msg, unsigned long error_msg_maxsize) {
  try {
    JVal vk = parse_json(vkey_json), pub = parse_json(public_json), pr = parse_json(proof_json);
    return verify_impl(vk, pub, pr) ? PROVER_OK : ZKPOA_VERIFY_INVALID_PROOF;
  } catch (const std::exception& e) {
    set_err(error_msg, error_msg_maxsize, e.what());
    return PROVER_ERROR;
  }
}

// ---- `snarkjs zkey export verificationkey` (scripts/g16_setup.sh:287-293) from zkey sections 1-3 --------------------
// The text snarkjs writes: JSON.stringify(vKey, null, 1) with the keys protocol, curve, nPublic, vk_alpha_1, vk_beta_2,
// vk_gamma_2, vk_delta_2, vk_alphabeta_12, IC (format and values pinned by the reference's committed *_vkey.json).
// vk_alphabeta_12 is wasmcurves' pairing value: e(alpha1, beta2) raised to 2x(6x^2 + 3x + 1), x = 4965661367192848881
// (SURVEY.md 8c), laid out [w^k][v^j] = Fq2 [c0, c1] in the 2-3-2 tower, which is this file's Fq12 as it stands.
namespace {
struct JOut {   // JSON.stringify(., null, 1): one space per nesting level
  std::string s;
  void indent(int d) { s.append((size_t)d, ' '); }
  void str(const std::string& v) { s += "\"" + v + "\""; }
  void g1(const G1& p, int d) {
    const bool inf = p.is_inf();
    std::string c[3] = {inf ? "0" : p.x.to_dec(), inf ? "1" : p.y.to_dec(), inf ? "0" : "1"};
    s += "[\n";
    for (int i = 0; i < 3; i++) {
      indent(d + 1);
      str(c[i]);
      s += i < 2 ? ",\n" : "\n";
    }
    indent(d);
    s += "]";
  }
  void fq2(const HFq2& v, int d) {
    s += "[\n";
    indent(d + 1);
    str(v.c0.to_dec());
    s += ",\n";
    indent(d + 1);
    str(v.c1.to_dec());
    s += "\n";
    indent(d);
    s += "]";
  }
  void g2(const G2& p, int d) {
    const bool inf = p.is_inf();
    HFq2 one = HFq2::one(), zero = HFq2::zero();
    const HFq2 c[3] = {inf ? zero : p.x, inf ? one : p.y, inf ? zero : one};
    s += "[\n";
    for (int i = 0; i < 3; i++) {
      indent(d + 1);
      // the third coordinate is the plain pair ["1", "0"], not a Montgomery-decoded value
      if (i == 2) {
        s += "[\n";
        indent(d + 2);
        str(inf ? "0" : "1");
        s += ",\n";
        indent(d + 2);
        str("0");
        s += "\n";
        indent(d + 1);
        s += "]";
      } else {
        fq2(c[i], d + 1);
      }
      s += i < 2 ? ",\n" : "\n";
    }
    indent(d);
    s += "]";
  }
};

uint32_t le32(const uint8_t* p) {
  uint32_t v;
  memcpy(&v, p, 4);
  return v;
}

std::string export_vkey_impl(const uint8_t* buf, uint64_t size) {
  if (size < 12 || memcmp(buf, "zkey", 4) != 0) throw std::runtime_error("zkey file: invalid file format (bad magic)");
  const uint32_t nsec = le32(buf + 8);
  const uint8_t *s2 = nullptr, *s3 = nullptr;
  uint64_t l2 = 0, l3 = 0, pos = 12;
  for (uint32_t i = 0; i < nsec; i++) {
    if (pos + 12 > size) throw std::runtime_error("zkey file: truncated section table");
    uint32_t id = le32(buf + pos);
    uint64_t len;
    memcpy(&len, buf + pos + 4, 8);
    pos += 12;
    if (len > size - pos) throw std::runtime_error("zkey file: truncated section");
    if (id == 2 && !s2) { s2 = buf + pos; l2 = len; }
    if (id == 3 && !s3) { s3 = buf + pos; l3 = len; }
    pos += len;
  }
  const uint64_t hdr = 4 + 32 + 4 + 32 + 12 + 64 + 64 + 128 + 128 + 64 + 128;
  if (!s2 || l2 < hdr || !s3) throw std::runtime_error("zkey file: missing groth16 header (section 2) or IC (section 3)");
  const uint32_t n_public = le32(s2 + 76);
  if (l3 != ((uint64_t)n_public + 1) * 64) throw std::runtime_error("zkey file: section 3 (IC) has the wrong size");
  const uint8_t* p = s2 + 84;
  G1 alpha1 = h_affine_from_bytes<HFq>(p);
  G2 beta2 = h_affine_from_bytes<HFq2>(p + 128), gamma2 = h_affine_from_bytes<HFq2>(p + 256);
  G2 delta2 = h_affine_from_bytes<HFq2>(p + 448);
  // e(alpha1, beta2)^(2x(6x^2+3x+1)): 190-bit exponent, little-endian limbs
  static const uint64_t kFc[3] = {0x2e5d4e223ddedaf4ull, 0x1ea96b02d9d9e38dull, 0x3bec47df15e307c8ull};
  Fq12 e = pairing::pairing(beta2, alpha1), ab = Fq12::one();
  for (int i = 2; i >= 0; i--)
    for (int b = 63; b >= 0; b--) {
      ab = ab.sqr();
      if ((kFc[i] >> b) & 1) ab = ab * e;
    }
  JOut o;
  o.s = "{\n \"protocol\": \"groth16\",\n \"curve\": \"bn128\",\n \"nPublic\": " + std::to_string(n_public) + ",\n \"vk_alpha_1\": ";
  o.g1(alpha1, 1);
  o.s += ",\n \"vk_beta_2\": ";
  o.g2(beta2, 1);
  o.s += ",\n \"vk_gamma_2\": ";
  o.g2(gamma2, 1);
  o.s += ",\n \"vk_delta_2\": ";
  o.g2(delta2, 1);
  o.s += ",\n \"vk_alphabeta_12\": [\n";
  const Fq6* half[2] = {&ab.c0, &ab.c1};
  for (int k = 0; k < 2; k++) {
    const HFq2 c[3] = {half[k]->c0, half[k]->c1, half[k]->c2};
    o.s += "  [\n";
    for (int j = 0; j < 3; j++) {
      o.s += "   ";
      o.fq2(c[j], 3);
      o.s += j < 2 ? ",\n" : "\n";
    }
    o.s += k < 1 ? "  ],\n" : "  ]\n";
  }
  o.s += " ],\n \"IC\": [\n";
  for (uint32_t i = 0; i <= n_public; i++) {
    o.s += "  ";
    o.g1(h_affine_from_bytes<HFq>(s3 + 64 * (uint64_t)i), 2);
    o.s += i < n_public ? ",\n" : "\n";
  }
  o.s += " ]\n}";
  return o.s;
}
}  // namespace

extern "C" int zkpoa_zkey_export_vkey(const void* zkey_buffer, unsigned long zkey_size, char* buffer, unsigned long* size,
                                      char* error_msg, unsigned long error_msg_maxsize) {
  if (!size || !zkey_buffer) return PROVER_ERROR;
  try {
    std::string s = export_vkey_impl(reinterpret_cast<const uint8_t*>(zkey_buffer), zkey_size);
    unsigned long needed = (unsigned long)s.size() + 1;
    if (!buffer || *size < needed) {
      *size = needed;
      return PROVER_ERROR_SHORT_BUFFER;
    }
    memcpy(buffer, s.c_str(), needed);
    *size = needed;
    return PROVER_OK;
  } catch (const std::exception& e) {
    set_err(error_msg, error_msg_maxsize, e.what());
    return PROVER_ERROR;
  }
}

// Same check on wire-format points (what a zkey and the prover hold): no JSON, no decimal conversion.
extern "C" int zkpoa_groth16_verify_points(const uint8_t* vkey_points, unsigned long vkey_size,
                                           const uint8_t proof_points[256], const uint8_t* public_le,
                                           unsigned long n_public, char* error_msg, unsigned long error_msg_maxsize) {
  try {
    if (!vkey_points || !proof_points || (n_public && !public_le)) throw std::runtime_error("null argument");
    // bound n_public before multiplying: (2^58) * 64 wraps to 0 and would pass the size test
    if (vkey_size < 448 + 64 || n_public != (vkey_size - 448) / 64 - 1 || (vkey_size - 448) % 64 != 0)
      throw std::runtime_error("vkey_points: expected alpha1(64) beta2(128) gamma2(128) delta2(128) + (nPublic+1) IC points");
    VKey vk;
    vk.alpha1 = h_affine_from_bytes<HFq>(vkey_points);
    vk.beta2 = h_affine_from_bytes<HFq2>(vkey_points + 64);
    vk.gamma2 = h_affine_from_bytes<HFq2>(vkey_points + 192);
    vk.delta2 = h_affine_from_bytes<HFq2>(vkey_points + 320);
    vk.nPublic = n_public;
    for (unsigned long i = 0; i <= n_public; i++) vk.IC.push_back(h_affine_from_bytes<HFq>(vkey_points + 448 + 64 * i));
    Proof pr{h_affine_from_bytes<HFq>(proof_points), h_affine_from_bytes<HFq>(proof_points + 192),
             h_affine_from_bytes<HFq2>(proof_points + 64)};
    std::vector<uint64_t> pub(4 * (size_t)n_public + 4);
    for (unsigned long i = 0; i < n_public; i++) {
      memcpy(&pub[4 * i], public_le + 32 * i, 32);
      if (!lt_modulus(&pub[4 * i], HFrParams::P)) return ZKPOA_VERIFY_INVALID_PROOF;
    }
    return verify_core(vk, pr, pub.data()) ? PROVER_OK : ZKPOA_VERIFY_INVALID_PROOF;
  } catch (const std::exception& e) {
    set_err(error_msg, error_msg_maxsize, e.what());
    return PROVER_ERROR;
  }
}

extern "C" int zkpoa_sanitize_proof(const char* vkey_json, const char* public_json, const char* proof_json,
                                    char* buffer, unsigned long* size, char* error_msg,
                                    unsigned long error_msg_maxsize) {
  if (!size) return PROVER_ERROR;
  try {
    JVal vk = parse_json(vkey_json), pub = parse_json(public_json), pr = parse_json(proof_json);
    std::string s = sanitize_impl(vk, pub, pr);
    unsigned long needed = (unsigned long)s.size() + 1;
    if (!buffer || *size < needed) {
      *size = needed;
      return PROVER_ERROR_SHORT_BUFFER;
    }
    memcpy(buffer, s.c_str(), needed);
    *size = needed;
    return PROVER_OK;
  } catch (const std::exception& e) {
    set_err(error_msg, error_msg_maxsize, e.what());
    return PROVER_ERROR;
  }
}
