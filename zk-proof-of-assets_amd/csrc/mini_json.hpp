// A minimal JSON reader (objects, arrays, strings, bare numbers) shared by the host-only tools: the verifier /
// sanitizer (vkey, proof, public files) and the Merkle-tree CLI (the parsed-signatures file). Nesting is capped.
#pragma once
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace zkpoa {
namespace json {

struct JVal {
  enum Kind { OBJ, ARR, STR, NUM } kind = NUM;
  std::string text;                                    // STR / NUM
  std::vector<JVal> items;                             // ARR
  std::vector<std::pair<std::string, JVal>> members;   // OBJ
  const JVal& at(const std::string& key) const {
    for (const auto& m : members)
      if (m.first == key) return m.second;
    throw std::runtime_error("JSON: missing key \"" + key + "\"");
  }
  const JVal& operator[](size_t i) const {
    if (kind != ARR || i >= items.size()) throw std::runtime_error("JSON: array index out of range");
    return items[i];
  }
  const std::string& scalar() const {
    if (kind != STR && kind != NUM) throw std::runtime_error("JSON: expected a number or string");
    return text;
  }
};

constexpr int kJsonMaxDepth = 32;   // proof / vkey / public files nest 5 deep; deeper input is rejected, not recursed into

struct JParser {
  const char* p;
  int depth = 0;
  explicit JParser(const char* s) : p(s) {}
  struct Nest {
    int& d;
    explicit Nest(int& depth) : d(depth) {
      if (++d > kJsonMaxDepth) throw std::runtime_error("JSON: nesting too deep");
    }
    ~Nest() { --d; }
  };
  void ws() {
    while (*p == ' ' || *p == '\n' || *p == '\t' || *p == '\r') p++;
  }
  JVal parse() {
    Nest nest(depth);
    ws();
    JVal v;
    if (*p == '{') {
      v.kind = JVal::OBJ;
      p++;
      ws();
      if (*p == '}') { p++; return v; }
      while (true) {
        ws();
        JVal k = parse();
        if (k.kind != JVal::STR) throw std::runtime_error("JSON: object key is not a string");
        ws();
        if (*p != ':') throw std::runtime_error("JSON: expected ':'");
        p++;
        v.members.emplace_back(k.text, parse());
        ws();
        if (*p == ',') { p++; continue; }
        if (*p == '}') { p++; break; }
        throw std::runtime_error("JSON: expected ',' or '}'");
      }
    } else if (*p == '[') {
      v.kind = JVal::ARR;
      p++;
      ws();
      if (*p == ']') { p++; return v; }
      while (true) {
        v.items.push_back(parse());
        ws();
        if (*p == ',') { p++; continue; }
        if (*p == ']') { p++; break; }
        throw std::runtime_error("JSON: expected ',' or ']'");
      }
    } else if (*p == '"') {
      v.kind = JVal::STR;
      p++;
      while (*p && *p != '"') {
        if (*p == '\\' && p[1]) p++;
        v.text.push_back(*p++);
      }
      if (*p != '"') throw std::runtime_error("JSON: unterminated string");
      p++;
    } else {
      v.kind = JVal::NUM;
      while (*p && *p != ',' && *p != ']' && *p != '}' && *p != ' ' && *p != '\n' && *p != '\t' && *p != '\r')
        v.text.push_back(*p++);
      if (v.text.empty()) throw std::runtime_error("JSON: unexpected character");
    }
    return v;
  }
};

inline JVal parse_json(const char* s) {
  if (!s) throw std::runtime_error("JSON: null input");
  JParser jp(s);
  JVal v = jp.parse();
  jp.ws();
  if (*jp.p) throw std::runtime_error("JSON: trailing characters");
  return v;
}


}  // namespace json
}  // namespace zkpoa
