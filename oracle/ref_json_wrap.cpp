// ORACLE — TEST INFRASTRUCTURE ONLY.
// nlohmann::json 3.5.0 as vendored by the reference (src/jlib/nlohmann/json.hpp), compiled where it lies; used to pin the
// number formatting of `dicey search` output (silica.h:143,149,160-170 dump doubles through json::dump()).
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
}
