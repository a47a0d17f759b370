// Host-side writer of the sdsl csa_wt<> file layout (what sdsl::store_to_checked_file produces for the type in
// reference src/index.h:80,122).  Used only by the GPU index builder (build.hip).  Field order and the construction
// rules of rank_support_v / select_support_mcl / byte_tree / huff_shape follow sdsl-lite as summarised in SURVEY.md
// Appendix A; no genuine sdsl file exists in this environment to compare against.
#pragma once
#include <cstdio>
#include <deque>
#include <queue>

#include "common.hpp"

namespace dg {

// The genuine field is std::hash<std::string> of sdsl's demangled class name, which cannot be reproduced without sdsl;
// readers in this repository skip it.  Pass the value of a genuine file through DICEY_FM9_HASH to stamp it instead.
static const u64 kPlaceholderClassHash = 0x44494345594F5243ULL;

struct FileOut {
  FILE* f = nullptr;
  bool ok = true;
  u64 written = 0;
  bool open(const char* path) {
    f = std::fopen(path, "wb");
    return f != nullptr;
  }
  void raw(const void* p, size_t n) {
    if (!ok || !n) return;
    if (std::fwrite(p, 1, n, f) != n) ok = false;
    written += n;
  }
  void u64v(u64 v) { raw(&v, 8); }
  void u16v(u16 v) { raw(&v, 2); }
  void u8v(u8 v) { raw(&v, 1); }
  bool close() {
    if (f && std::fclose(f) != 0) ok = false;
    f = nullptr;
    return ok;
  }
};

// int_vector<0> with run-time width
struct PackedVec {
  u64 nelem = 0;
  u8 width = 1;
  std::vector<u64> w;
  void init(u64 n, u8 wd) {
    nelem = n;
    width = wd;
    w.assign((n * wd + 63) / 64, 0);
  }
  void set(u64 i, u64 v) {
    u64 b = i * width, q = b >> 6, o = b & 63;
    u64 m = width == 64 ? ~0ULL : ((1ULL << width) - 1);
    v &= m;
    w[q] |= v << o;
    if (o + width > 64) w[q + 1] |= v >> (64 - o);
  }
  void write(FileOut& o) const {
    o.u64v(nelem * width);
    o.u8v(width);
    o.raw(w.data(), w.size() * 8);
  }
};

static inline u32 bits_hi(u64 x) { return x ? 63u - (u32)__builtin_clzll(x) : 0u; }

// select_support_mcl<b,1>: positions of every 4096th argument; per superblock either all positions ("long", when the
// superblock spans more than log^4 bits) or every 64th relative position ("mini").
inline void write_select_support(FileOut& o, const u64* bv, u64 bv_bits, bool ones) {
  const u64 nwords = (bv_bits + 63) >> 6;
  auto word = [&](u64 k) -> u64 {
    u64 x = ones ? bv[k] : ~bv[k];
    if (k == nwords - 1 && (bv_bits & 63)) x &= (1ULL << (bv_bits & 63)) - 1;
    return x;
  };
  u64 total = 0;
  for (u64 k = 0; k < nwords; ++k) total += (u64)__builtin_popcountll(word(k));
  o.u64v(total);
  if (!total) return;
  const u64 SB = 4096, sb = (total + SB - 1) / SB;
  const u64 cap = nwords << 6;
  const u32 logn = bits_hi(cap) + 1;
  const u64 logn4 = (u64)logn * logn * logn * logn;
  // pass 1: position of every 64th argument and of the last argument of each superblock
  std::vector<u64> p64((total + 63) / 64), last(sb);
  u64 seen = 0;
  for (u64 k = 0; k < nwords; ++k) {
    u64 x = word(k);
    u32 pc = (u32)__builtin_popcountll(x);
    if (!pc) continue;
    u64 lo = seen, hi = seen + pc;  // argument numbers in this word: [lo, hi)
    // only walk the bits of words that hold an argument we sample: every 64th, the last of a superblock, the last
    u64 first64 = (lo + 63) / 64 * 64;
    u64 next_sb_last = lo + (SB - 1 - lo % SB);
    if (first64 < hi || next_sb_last < hi || hi == total) {
      u64 y = x;
      for (u64 a = lo; a < hi; ++a) {
        u32 bit = (u32)__builtin_ctzll(y);
        y &= y - 1;
        if ((a & 63) == 0) p64[a >> 6] = (k << 6) + bit;
        if ((a % SB) == SB - 1 || a == total - 1) last[a / SB] = (k << 6) + bit;
      }
    }
    seen = hi;
  }
  PackedVec superblock;
  superblock.init(sb, (u8)logn);
  std::vector<PackedVec> blk(sb);
  std::vector<u8> is_mini(sb, 1);
  bool any_long = false;
  for (u64 s = 0; s < sb; ++s) {
    u64 first = p64[s * (SB / 64)];
    u64 cnt = (s == sb - 1) ? total - s * SB : SB;
    superblock.set(s, first);
    u64 diff = last[s] - first;
    if (diff > logn4) {
      any_long = true;
      is_mini[s] = 0;
      blk[s].init(SB, (u8)(bits_hi(last[s]) + 1));
      u64 j = 0;
      for (u64 k = first >> 6; j < cnt; ++k) {
        u64 x = word(k);
        if (k == (first >> 6)) x &= ~0ULL << (first & 63);
        while (x && j < cnt) {
          blk[s].set(j++, (k << 6) + (u64)__builtin_ctzll(x));
          x &= x - 1;
        }
      }
    } else {
      blk[s].init(64, (u8)(bits_hi(diff) + 1));
      for (u64 j = 0; j < cnt; j += 64) blk[s].set(j / 64, p64[(s * SB + j) >> 6] - first);
    }
  }
  superblock.write(o);
  // mini_or_long: an empty bit_vector when no long superblock exists
  if (any_long) {
    std::vector<u64> mol((sb + 63) / 64, 0);
    for (u64 s = 0; s < sb; ++s)
      if (is_mini[s]) mol[s >> 6] |= 1ULL << (s & 63);
    o.u64v(sb);
    o.raw(mol.data(), mol.size() * 8);
  } else {
    o.u64v(0);
  }
  for (u64 s = 0; s < sb; ++s) blk[s].write(o);
}

struct HuffNode {
  u64 bv_pos, bv_pos_rank;
  u16 parent, child[2];
};
struct HuffTree {
  std::vector<HuffNode> nodes;  // BFS order, root = 0
  u16 c_to_leaf[256];
  u64 path[256];
  u64 bv_bits = 0;
};
// huff_shape::construct_tree + _byte_tree constructor (BFS renumbering) — sdsl wt_huff<>
inline HuffTree build_huffman(const u64 freq[256]) {
  struct Tmp {
    u64 freq, sym;
    u16 parent, child[2];
  };
  std::vector<Tmp> tn;
  typedef std::pair<u64, u64> P;
  std::priority_queue<P, std::vector<P>, std::greater<P>> pq;
  for (int c = 0; c < 256; ++c)
    if (freq[c]) {
      pq.push(P(freq[c], tn.size()));
      tn.push_back(Tmp{freq[c], (u64)c, 0xFFFF, {0xFFFF, 0xFFFF}});
    }
  while (pq.size() > 1) {
    P a = pq.top();
    pq.pop();
    P b = pq.top();
    pq.pop();
    tn[a.second].parent = tn[b.second].parent = (u16)tn.size();
    pq.push(P(a.first + b.first, tn.size()));
    tn.push_back(Tmp{a.first + b.first, 0, 0xFFFF, {(u16)a.second, (u16)b.second}});
  }
  HuffTree t;
  t.nodes.resize(tn.size());
  auto cp = [&](u32 dst, u32 src) {
    t.nodes[dst] = HuffNode{tn[src].freq, tn[src].sym, tn[src].parent, {tn[src].child[0], tn[src].child[1]}};
  };
  cp(0, (u32)tn.size() - 1);
  t.nodes[0].parent = 0xFFFF;
  u32 cnt = 1;
  std::deque<u32> q;
  q.push_back(0);
  while (!q.empty()) {
    u32 v = q.front();
    q.pop_front();
    u64 frq = t.nodes[v].bv_pos;
    t.nodes[v].bv_pos = t.bv_bits;
    if (t.nodes[v].child[0] != 0xFFFF) {
      t.bv_bits += frq;
      for (u32 k = 0; k < 2; ++k) {
        cp(cnt, t.nodes[v].child[k]);
        t.nodes[cnt].parent = (u16)v;
        q.push_back(cnt);
        t.nodes[v].child[k] = (u16)cnt++;
      }
    }
  }
  for (int c = 0; c < 256; ++c) {
    t.c_to_leaf[c] = 0xFFFF;
    t.path[c] = 0;
  }
  for (u32 v = 0; v < t.nodes.size(); ++v)
    if (t.nodes[v].child[0] == 0xFFFF) t.c_to_leaf[(u8)t.nodes[v].bv_pos_rank] = (u16)v;
  for (int c = 0; c < 256; ++c)
    if (t.c_to_leaf[c] != 0xFFFF) {
      u32 v = t.c_to_leaf[c];
      u64 pw = 0, pl = 0;
      while (v != 0) {
        pw <<= 1;
        if (t.nodes[t.nodes[v].parent].child[1] == v) pw |= 1;
        v = t.nodes[v].parent;
        ++pl;
      }
      t.path[c] = pw | (pl << 56);
    }
  return t;
}

}  // namespace dg
