// Test harness (not part of the product): the order-free formulation of neighbors()' size cap (dicey_amd/csrc/cap_enum.hpp) run
// sequentially on the host with the header's own leaf / rank / key functions, so that `pytest -m "not gpu"` can hold the arithmetic
// the device kernel k_cap_enum relies on against the literal restatement of neighbors.h:29-92.
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../dicey_amd/csrc/cap_enum.hpp"

using namespace dg::cap;

extern "C" int ce_enumerate(const char* q, uint32_t m, uint32_t d, uint32_t maxsize, char** out, uint64_t* count, int* fired, uint64_t* nleaves) {
  if (m + d > MAX_KEY_LEN || d < 1 || d > 2 || d >= m) return -1;
  u64 codes = 0;
  for (u32 i = 0; i < m; ++i) {
    const char ch = q[i];
    const u64 c = ch == 'A' ? 0 : ch == 'C' ? 1 : ch == 'G' ? 2 : ch == 'T' ? 3 : 9;
    if (c > 3) return -2;
    codes = (codes << 2) | c;
  }
  std::unordered_map<u64, u64> birth;
  birth[make_key(codes, m)] = 0;
  u64 nl = 0, maxrank = 0;
  for (u32 j1 = 0; j1 < m; ++j1)
    for (u32 op1 = 0; op1 < 8; ++op1)
      for (u32 item = 0; item < items_per_child(m, d); ++item) {
        u64 key, rank;
        if (!leaf_of(codes, m, d, j1, op1, item, key, rank)) continue;
        ++nl;
        maxrank = std::max(maxrank, rank);
        auto it = birth.find(key);
        if (it == birth.end() || rank < it->second) birth[key] = rank;
      }
  if (nleaves) *nleaves = nl;
  // every inserted leaf has its own rank below the size of the trie (the unedited leaf is the one rank without a string)
  if (nl + 1 != total_leaves(m, d) || maxrank > total_leaves(m, d)) return -3;
  const u32 minlen = m - d;
  std::vector<long long> ev(total_leaves(m, d) + 2, 0);
  struct Rec { u64 key, b, dth; };
  std::vector<Rec> recs;
  const u64 INF = ~0ULL;
  for (auto& kv : birth) {
    const u64 key = kv.first;
    const u32 len = key_len(key);
    const u64 bits = key & ((1ULL << (2 * len)) - 1);
    u64 death = INF;
    for (u32 L = minlen ? minlen : 1; L < len; ++L)
      for (u32 a = 0; a + L <= len; ++a) {
        auto it = birth.find(sub_key(bits, len, a, L));
        if (it != birth.end()) death = std::min(death, it->second);
      }
    recs.push_back({key, kv.second, death});
    if (kv.second < death) {
      ev[kv.second] += 1;
      if (death != INF) ev[death] -= 1;
    }
  }
  u64 T = INF;
  long long run = 0;
  for (size_t r = 0; r < ev.size(); ++r) {
    run += ev[r];
    if (run >= (long long)maxsize) {
      T = r;
      break;
    }
  }
  *fired = T != INF;
  std::vector<std::string> set;
  for (const Rec& r : recs)
    if (r.b < r.dth && r.b <= T && (T == INF ? r.dth == INF : T < r.dth)) {
      const u32 len = key_len(r.key);
      std::string s(len, 'A');
      for (u32 i = 0; i < len; ++i) s[i] = "ACGT"[(r.key >> (2 * (len - 1 - i))) & 3];
      set.push_back(s);
    }
  std::sort(set.begin(), set.end());
  size_t bytes = 1;
  for (auto& s : set) bytes += s.size() + 1;
  char* buf = (char*)std::malloc(bytes);
  char* w = buf;
  for (auto& s : set) {
    std::memcpy(w, s.data(), s.size());
    w += s.size();
    *w++ = '\n';
  }
  *w = 0;
  *out = buf;
  *count = set.size();
  return 0;
}
extern "C" void ce_free(void* p) { std::free(p); }
