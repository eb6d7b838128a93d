// Minimal JSON DOM, enough for the `model_config` attribute of a legacy Keras
// .h5 file (read by tf.keras.models.load_model, PyCFD_ML_accelerated.py:831).
#pragma once
#include <cstdlib>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace jsonmin {

struct Value {
  enum Kind { Null, Bool, Num, Str, Arr, Obj } kind = Null;
  bool b = false;
  double num = 0;
  std::string str;
  std::vector<Value> arr;
  std::vector<std::pair<std::string, Value>> obj;

  const Value* get(const std::string& k) const {
    for (auto& kv : obj)
      if (kv.first == k) return &kv.second;
    return nullptr;
  }
  const Value& at(const std::string& k) const {
    const Value* v = get(k);
    if (!v) throw std::runtime_error("json: missing key '" + k + "'");
    return *v;
  }
  int as_int() const { return (int)num; }
};

class Parser {
 public:
  explicit Parser(const std::string& s) : s_(s) {}
  Value parse() {
    Value v = value();
    ws();
    if (i_ != s_.size()) err("trailing characters");
    return v;
  }

 private:
  const std::string& s_;
  size_t i_ = 0;
  [[noreturn]] void err(const char* m) { throw std::runtime_error(std::string("json: ") + m + " at offset " + std::to_string(i_)); }
  void ws() { while (i_ < s_.size() && (s_[i_] == ' ' || s_[i_] == '\n' || s_[i_] == '\t' || s_[i_] == '\r')) ++i_; }
  bool lit(const char* t) {
    size_t n = std::char_traits<char>::length(t);
    if (s_.compare(i_, n, t) == 0) { i_ += n; return true; }
    return false;
  }
  Value value() {
    ws();
    if (i_ >= s_.size()) err("unexpected end");
    Value v;
    char c = s_[i_];
    if (c == '{') {
      v.kind = Value::Obj; ++i_; ws();
      if (i_ < s_.size() && s_[i_] == '}') { ++i_; return v; }
      for (;;) {
        ws();
        std::string k = string();
        ws();
        if (i_ >= s_.size() || s_[i_] != ':') err("expected ':'");
        ++i_;
        v.obj.emplace_back(k, value());
        ws();
        if (i_ < s_.size() && s_[i_] == ',') { ++i_; continue; }
        if (i_ < s_.size() && s_[i_] == '}') { ++i_; break; }
        err("expected ',' or '}'");
      }
    } else if (c == '[') {
      v.kind = Value::Arr; ++i_; ws();
      if (i_ < s_.size() && s_[i_] == ']') { ++i_; return v; }
      for (;;) {
        v.arr.push_back(value());
        ws();
        if (i_ < s_.size() && s_[i_] == ',') { ++i_; continue; }
        if (i_ < s_.size() && s_[i_] == ']') { ++i_; break; }
        err("expected ',' or ']'");
      }
    } else if (c == '"') { v.kind = Value::Str; v.str = string(); }
    else if (lit("true")) { v.kind = Value::Bool; v.b = true; }
    else if (lit("false")) { v.kind = Value::Bool; v.b = false; }
    else if (lit("null")) { v.kind = Value::Null; }
    else {
      const char* st = s_.c_str() + i_;
      char* en = nullptr;
      v.num = std::strtod(st, &en);
      if (en == st) err("bad token");
      v.kind = Value::Num;
      i_ += (size_t)(en - st);
    }
    return v;
  }
  std::string string() {
    if (i_ >= s_.size() || s_[i_] != '"') err("expected string");
    ++i_;
    std::string out;
    while (i_ < s_.size() && s_[i_] != '"') {
      char c = s_[i_++];
      if (c == '\\' && i_ < s_.size()) {
        char e = s_[i_++];
        switch (e) {
          case 'n': out += '\n'; break;
          case 't': out += '\t'; break;
          case 'r': out += '\r'; break;
          case 'b': out += '\b'; break;
          case 'f': out += '\f'; break;
          case 'u': {  // keep BMP code points as UTF-8
            if (i_ + 4 > s_.size()) err("bad \\u escape");
            unsigned cp = (unsigned)std::strtoul(s_.substr(i_, 4).c_str(), nullptr, 16);
            i_ += 4;
            if (cp < 0x80) out += (char)cp;
            else if (cp < 0x800) { out += (char)(0xC0 | (cp >> 6)); out += (char)(0x80 | (cp & 0x3F)); }
            else { out += (char)(0xE0 | (cp >> 12)); out += (char)(0x80 | ((cp >> 6) & 0x3F)); out += (char)(0x80 | (cp & 0x3F)); }
            break;
          }
          default: out += e;
        }
      } else out += c;
    }
    if (i_ >= s_.size()) err("unterminated string");
    ++i_;
    return out;
  }
};

inline Value parse(const std::string& s) { return Parser(s).parse(); }

}  // namespace jsonmin
