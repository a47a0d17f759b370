// ORACLE — TEST INFRASTRUCTURE ONLY.
// nlohmann::json 3.5.0 as vendored by the reference (src/jlib/nlohmann/json.hpp), compiled where it lies.  Two uses:
//  * ref_json_dump_double pins the number formatting of `dicey search` output (silica.h:143,149,160-170);
//  * the object builder below lets the oracle's writers produce every JSON object of `dicey hunt` / `dicey search` the way
//    the reference does — assign the members to a nlohmann::json, dump() it (hunter.h:112-116,122-152, silica.h:113-181) —
//    so key order, string escaping, integer and double formatting are the reference library's own, not a restatement.
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <string>

#include <nlohmann/json.hpp>

extern "C" {
// writes dump() of the double into out (NUL-terminated, cap bytes); returns the length
int ref_json_dump_double(double x, char* out, int cap) {
  nlohmann::json j = x;
  std::string s = j.dump();
  if ((int)s.size() + 1 > cap) return -1;
  std::memcpy(out, s.c_str(), s.size() + 1);
  return (int)s.size();
}

void* ref_json_new() { return new nlohmann::json(); }
void ref_json_free(void* j) { delete (nlohmann::json*)j; }
// the member types are those of the reference's right-hand sides: std::string, uint32_t / std::size_t (unsigned),
// int (std::abs of the score), bool, double
void ref_json_set_str(void* j, const char* key, const char* val, uint64_t len) { (*(nlohmann::json*)j)[key] = std::string(val, (size_t)len); }
void ref_json_set_u64(void* j, const char* key, uint64_t v) { (*(nlohmann::json*)j)[key] = v; }
void ref_json_set_i64(void* j, const char* key, int64_t v) { (*(nlohmann::json*)j)[key] = v; }
void ref_json_set_bool(void* j, const char* key, int v) { (*(nlohmann::json*)j)[key] = (v != 0); }
void ref_json_set_f64(void* j, const char* key, double v) { (*(nlohmann::json*)j)[key] = v; }
// dump() into a malloc'ed, NUL-terminated buffer (free with ref_json_release); NULL if dump() throws (invalid UTF-8:
// json.exception.type_error.316, which the reference does not catch)
char* ref_json_dump(void* j, uint64_t* len) {
  try {
    std::string s = ((nlohmann::json*)j)->dump();
    char* out = (char*)std::malloc(s.size() + 1);
    std::memcpy(out, s.c_str(), s.size() + 1);
    if (len) *len = s.size();
    return out;
  } catch (const std::exception&) {
    return nullptr;
  }
}
void ref_json_release(char* p) { std::free(p); }
}
