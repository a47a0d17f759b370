// complement() of the reference (src/util.h:54-91): IUPAC table, both cases, anything else -> 'N'
#pragma once
#include <cstring>

namespace dg {
inline char complement_iupac(char n) {
  static const char* from = "AaCcGgTtUuRrYySsWwKkMmBbVvDdHhNn";
  static const char* to = "TtGgCcAaAaYyRrSsWwMmKkVvBbHhDdNn";
  const char* p = n ? std::strchr(from, n) : nullptr;
  return p ? to[p - from] : 'N';
}
}  // namespace dg
