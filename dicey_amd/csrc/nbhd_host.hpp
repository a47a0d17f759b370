// Host-side neighbourhood enumeration WITH the maxNeighborhood cap (reference src/neighbors.h:29-92).
//
// The search kernel (hunt.hip, k_search) enumerates the occurring strings of the <= d-edit language and keeps the
// substring-minimal ones; that equals what the reference searches as long as neighbors()' early return
// (`strset.size() >= maxsize`, neighbors.h:50) stays silent.  When it fires, the searched set is whatever the working set
// held at that moment, which depends on the order the reference generates strings in.  This file reproduces exactly that:
// the same generation order (at every position: delete, keep, substitute A,C,G,T, insert A,C,G,T before the position;
// neighbors.h:52-78), the same working set (neighbors.h:29-45: a new string is dropped when a member is a substring of it,
// otherwise it replaces every member that contains it), the same stopping rule.  It is used for the (query, strand)
// groups whose neighbourhood could reach the cap; their strings then enter the device pipeline as explicit patterns.
//
// The reference's _insert scans the whole set with std::string::find for every generated string (1.3 s per 20-mer at
// d = 2).  Here the working set is a hash table of strings plus a hash index "substring -> members that contain it":
// every language string has between m-d and m+d characters, so a member can only be contained in / contain strings whose
// length differs by at most 2d, and both tests are a handful of table probes:
//   * a string that was generated before changes nothing (a live member stays, a dropped or replaced one stays
//     dominated: members are only ever replaced by their own substrings) -> one probe of the `seen` table;
//   * a new string s is dominated  <=>  one of its proper substrings of >= m-d characters was generated before;
//   * otherwise s enters and the live members containing s are found under s in the substring index.
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

namespace dg {

class CappedNeighborhood {
 public:
  // q: normalised sequence over A,C,G,T,N.  Returns the final working set in std::set<std::string> order;
  // `fired` <=> the set reached maxsize (the reference then prints its "Neighborhood size exceeds" warning).
  static std::vector<std::string> enumerate(const std::string& q, unsigned dist, bool indel, uint32_t maxsize, bool& fired,
                                            uint64_t* generated = nullptr) {
    CappedNeighborhood w(q, dist, indel, maxsize);
    w.run();
    fired = w.live_ >= (uint64_t)maxsize;
    if (generated) *generated = w.generated_;
    return w.result();
  }

 private:
  struct Entry {  // a generated string
    uint32_t off, len;
    bool live;
  };
  // open-addressing table: 64-bit hash -> value (entry id, or head of a chain); keys are verified by the caller
  struct Table {
    std::vector<uint64_t> key;
    std::vector<uint32_t> val;
    size_t used = 0, mask = 0;
    void init(size_t cap) {
      size_t n = 64;
      while (n < cap * 2) n <<= 1;
      if (key.size() > n) n = key.size();  // a table that grew for an earlier sequence of this thread keeps its size (Work below)
      key.assign(n, 0);
      val.assign(n, ~0u);
      mask = n - 1;
      used = 0;
    }
    void grow() {
      std::vector<uint64_t> k2;
      std::vector<uint32_t> v2;
      k2.swap(key);
      v2.swap(val);
      init(k2.size());
      for (size_t i = 0; i < k2.size(); ++i)
        if (v2[i] != ~0u) put_new(k2[i], v2[i]);
    }
    // slot holding `h` with predicate ok(value), or the empty slot where it would go
    template <class Ok>
    size_t find(uint64_t h, Ok ok) const {
      size_t i = (size_t)(h * 0x9E3779B97F4A7C15ULL >> 17) & mask;
      while (val[i] != ~0u && !(key[i] == h && ok(val[i]))) i = (i + 1) & mask;
      return i;
    }
    void put_new(uint64_t h, uint32_t v) {
      size_t i = (size_t)(h * 0x9E3779B97F4A7C15ULL >> 17) & mask;
      while (val[i] != ~0u) i = (i + 1) & mask;
      key[i] = h;
      val[i] = v;
      ++used;
    }
  };
  struct Link {  // substring index: chain of (member, offset inside the member, length)
    uint32_t entry, at, len, next;
  };

  // The working storage lives once per host thread and is reused from sequence to sequence: an enumeration grows ~5 MB of
  // vectors, and 256 threads each mapping and unmapping that per sequence spend their time in the kernel's address-space lock
  // (r03, 2 x EPYC 9575F: 0.23 ms per 25-mer strand of wall time with 32 threads, 0.58 ms with 256 before this).
  struct Work {
    std::vector<char> arena;
    std::vector<Entry> ent;
    Table seen, sub;
    std::vector<Link> links;
    std::vector<uint64_t> pw, pre;
  };
  static Work& tls() {
    thread_local Work w;
    return w;
  }
  Work& W_;
  std::string cur_;
  const unsigned dist_;
  const bool indel_;
  const uint64_t maxsize_;
  const uint32_t minlen_;  // shortest string of the language
  std::vector<char>& arena_;
  std::vector<Entry>& ent_;
  Table& seen_;
  Table& sub_;
  std::vector<Link>& links_;
  uint64_t live_ = 0, generated_ = 0;
  bool stop_ = false;
  std::vector<uint64_t>& pw_;   // powers of the hash base
  std::vector<uint64_t>& pre_;  // prefix hashes of the candidate

  static constexpr uint64_t BASE = 0x100000001B3ULL * 2 + 1;

  CappedNeighborhood(const std::string& q, unsigned dist, bool indel, uint32_t maxsize)
      : W_(tls()), cur_(q), dist_(dist), indel_(indel), maxsize_(maxsize), minlen_((uint32_t)(q.size() > dist ? q.size() - dist : 0)),
        arena_(W_.arena), ent_(W_.ent), seen_(W_.seen), sub_(W_.sub), links_(W_.links), pw_(W_.pw), pre_(W_.pre) {
    arena_.clear();
    ent_.clear();
    links_.clear();
    seen_.init(1 << 12);
    sub_.init(1 << 14);
    pw_.assign(q.size() + dist + 2, 1);
    for (size_t i = 1; i < pw_.size(); ++i) pw_[i] = pw_[i - 1] * BASE;
    pre_.assign(q.size() + dist + 2, 0);
  }
  uint64_t sub_hash(uint32_t a, uint32_t b) const {  // hash of cand[a,b), length mixed in
    return (pre_[b] - pre_[a] * pw_[b - a]) ^ ((uint64_t)(b - a) << 56);
  }
  void prefix_hashes(const std::string& s) {
    pre_[0] = 0;
    for (size_t i = 0; i < s.size(); ++i) pre_[i + 1] = pre_[i] * BASE + (uint8_t)s[i] + 1;
  }
  bool same(const Entry& e, uint32_t at, const char* p, uint32_t len) const { return std::memcmp(arena_.data() + e.off + at, p, len) == 0; }

  // neighbors.h:29-45 on the candidate `s`
  void put(const std::string& s) {
    ++generated_;
    const uint32_t n = (uint32_t)s.size();
    if (!indel_) {  // plain std::set insert; strings are distinct by construction, but stay exact anyway
      prefix_hashes(s);
      const uint64_t h = sub_hash(0, n);
      size_t slot = seen_.find(h, [&](uint32_t id) { return ent_[id].len == n && same(ent_[id], 0, s.data(), n); });
      if (seen_.val[slot] != ~0u) return;
      add_entry(s, h, slot);
      ++live_;
      return;
    }
    prefix_hashes(s);
    const uint64_t h = sub_hash(0, n);
    {
      size_t slot = seen_.find(h, [&](uint32_t id) { return ent_[id].len == n && same(ent_[id], 0, s.data(), n); });
      if (seen_.val[slot] != ~0u) return;  // generated before: nothing changes
    }
    // dominated by an earlier string?  (proper substrings of at least minlen_ characters)
    for (uint32_t len = minlen_ ? minlen_ : 1; len < n; ++len)
      for (uint32_t a = 0; a + len <= n; ++a) {
        const uint64_t hs = sub_hash(a, a + len);
        size_t slot = seen_.find(hs, [&](uint32_t id) { return ent_[id].len == len && same(ent_[id], 0, s.data() + a, len); });
        if (seen_.val[slot] != ~0u) {
          // remember s as generated (dominated strings never enter, and never need to be looked at again)
          size_t own = seen_.find(h, [](uint32_t) { return false; });
          add_entry(s, h, own, /*live=*/false);
          return;
        }
      }
    // s enters; every live member that contains s leaves (they sit under s in the substring index)
    {
      size_t slot = sub_.find(h, [&](uint32_t head) {
        const Link& l = links_[head];
        return l.len == n && same(ent_[l.entry], l.at, s.data(), n);
      });
      if (sub_.val[slot] != ~0u) {
        for (uint32_t k = sub_.val[slot]; k != ~0u; k = links_[k].next) {
          Entry& e = ent_[links_[k].entry];
          if (e.live) {
            e.live = false;
            --live_;
          }
        }
      }
    }
    size_t own = seen_.find(h, [](uint32_t) { return false; });
    const uint32_t id = add_entry(s, h, own);
    ++live_;
    // index the substrings of s a later, shorter string could be equal to (proper ones of >= minlen_ characters)
    for (uint32_t len = minlen_ ? minlen_ : 1; len < n; ++len)
      for (uint32_t a = 0; a + len <= n; ++a) {
        const uint64_t hs = sub_hash(a, a + len);
        size_t slot = sub_.find(hs, [&](uint32_t head) {
          const Link& l = links_[head];
          return l.len == len && same(ent_[l.entry], l.at, s.data() + a, len);
        });
        Link l{id, a, len, ~0u};
        if (sub_.val[slot] != ~0u) {
          l.next = sub_.val[slot];
          links_.push_back(l);
          sub_.val[slot] = (uint32_t)links_.size() - 1;
        } else {
          links_.push_back(l);
          sub_.key[slot] = hs;
          sub_.val[slot] = (uint32_t)links_.size() - 1;
          if (++sub_.used * 2 > sub_.key.size()) sub_.grow();
        }
      }
  }
  uint32_t add_entry(const std::string& s, uint64_t h, size_t slot, bool live = true) {
    Entry e;
    e.off = (uint32_t)arena_.size();
    e.len = (uint32_t)s.size();
    e.live = live;
    arena_.insert(arena_.end(), s.begin(), s.end());
    ent_.push_back(e);
    seen_.key[slot] = h;
    seen_.val[slot] = (uint32_t)ent_.size() - 1;
    if (++seen_.used * 2 > seen_.key.size()) seen_.grow();
    return (uint32_t)ent_.size() - 1;
  }

  // neighbors.h:47-83.  `s` is edited in place and restored; pos / left as in the reference.
  void walk(std::string& s, unsigned left, size_t pos) {
    if (stop_) return;
    if (live_ >= maxsize_) {  // neighbors.h:50; the set cannot change any more once this holds
      stop_ = true;
      return;
    }
    if (pos >= s.size()) {
      if (left < dist_) put(s);
      return;
    }
    const char orig = s[pos];
    if (left > 0 && indel_) {  // deletion, same position again
      s.erase(pos, 1);
      walk(s, left - 1, pos);
      s.insert(pos, 1, orig);
    }
    walk(s, left, pos + 1);
    if (left > 0) {
      static const char alphabet[4] = {'A', 'C', 'G', 'T'};
      for (char a : alphabet)
        if (a != orig) {
          s[pos] = a;
          walk(s, left - 1, pos + 1);
        }
      s[pos] = orig;
      if (indel_)
        for (char a : alphabet) {  // insertion in front of the position
          s.insert(pos, 1, a);
          walk(s, left - 1, pos + 1);
          s.erase(pos, 1);
        }
    }
  }
  void run() {
    put(cur_);  // neighbors.h:90: the sequence itself, before any size test
    walk(cur_, dist_, 0);
  }
  std::vector<std::string> result() const {
    std::vector<std::string> out;
    out.reserve((size_t)live_);
    for (const Entry& e : ent_)
      if (e.live) out.emplace_back(arena_.data() + e.off, e.len);
    std::sort(out.begin(), out.end());
    return out;
  }
};

}  // namespace dg
