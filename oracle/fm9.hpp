// ORACLE — TEST INFRASTRUCTURE ONLY.  Nothing under dicey_amd/ may include, link or call this.
//
// CPU restatement of the FM-index the reference uses: sdsl::csa_wt<> with all defaults
// (reference: src/index.h:80 `csa_wt<> fm_index`, src/hunter.h:253, src/silica.h:340).
// The arithmetic lives in the third-party module xxsds/sdsl-lite (.gitmodules:5-8), which is an
// EMPTY submodule in /root/reference and whose pinned commit is unknown.  What follows restates
// sdsl-lite's published algorithms (csa_wt / wt_huff / rank_support_v / select_support_mcl /
// byte_alphabet / sa_order_sa_sampling / isa_sampling) and its serialisation order as summarised in
// SURVEY.md Appendix A.  PARITY UNPINNED for the byte layout: no genuine `.fm9` and no sdsl source
// exist in this environment.  The *semantics* (count = #occurrences, locate = positions,
// extract = substring) are pinned by the brute-force twin orc_bf_locate (oracle_capi.cpp).
//
// Call sites in the reference that this file stands in for:
//   load_from_checked_file  hunter.h:256  silica.h:343
//   sdsl::count             hunter.h:353  silica.h:470
//   sdsl::locate            hunter.h:355  silica.h:472
//   sdsl::extract           hunter.h:371  silica.h:490
//   fm_index.size()         hunter.h:368  silica.h:487
//   sdsl::construct + store_to_checked_file   index.h:121-122
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <deque>
#include <queue>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace orc {

using u8 = uint8_t;
using u16 = uint16_t;
using u32 = uint32_t;
using u64 = uint64_t;

static inline u32 hi_bit(u64 x) { return x ? 63u - (u32)__builtin_clzll(x) : 0u; }  // sdsl bits::hi

// sdsl::int_vector<t_width>; width 0 in the type means "run-time width stored in the file".
struct IntVec {
  u64 bits = 0;   // length in BITS (what sdsl writes first)
  u8 width = 64;  // element width in bits
  std::vector<u64> w;
  u64 size() const { return width ? bits / width : 0; }
  void init(u64 n_elems, u8 wd) {
    width = wd;
    bits = n_elems * wd;
    w.assign((bits + 63) / 64, 0);
  }
  u64 get(u64 i) const {
    u64 b = i * width, q = b >> 6, o = b & 63;
    u64 v = w[q] >> o;
    if (o + width > 64) v |= w[q + 1] << (64 - o);
    return width == 64 ? v : (v & ((1ULL << width) - 1));
  }
  void set(u64 i, u64 v) {
    u64 b = i * width, q = b >> 6, o = b & 63;
    u64 m = width == 64 ? ~0ULL : ((1ULL << width) - 1);
    v &= m;
    w[q] = (w[q] & ~(m << o)) | (v << o);
    if (o + width > 64) {
      u32 sh = 64 - o;
      w[q + 1] = (w[q + 1] & ~(m >> sh)) | (v >> sh);
    }
  }
};

struct WtNode {  // sdsl _byte_tree::_node, 22 bytes on disk
  u64 bv_pos = 0, bv_pos_rank = 0;
  u16 parent = 0xFFFF, child[2] = {0xFFFF, 0xFFFF};
};

struct SelectMcl {  // select_support_mcl<b,1>; the reference never calls select, kept for the writer
  u64 arg_cnt = 0;
  IntVec superblock;
  std::vector<u8> is_mini;  // one per superblock (only meaningful when any_long)
  bool any_long = false;
  std::vector<IntVec> blocks;  // long or mini per superblock
};

struct Csa {
  // wavelet tree (wt_huff<>)
  u64 n = 0;         // = |T| + 1 (sentinel)  == fm_index.size()
  u64 wt_sigma = 0;  // wt_pc::m_sigma
  u64 bv_bits = 0;
  std::vector<u64> bv;
  std::vector<u64> rank_bb;  // rank_support_v::m_basic_block (2 words per 512-bit superblock)
  SelectMcl sel1, sel0;
  std::vector<WtNode> nodes;
  u16 c_to_leaf[256];
  u64 path[256];
  // samples
  IntVec sa_samples;   // SA[32k]
  IntVec isa_samples;  // ISA[64k]
  // byte_alphabet
  u8 char2comp[256];
  std::vector<u8> comp2char;
  std::vector<u64> C;  // sigma+1
  u16 sigma = 0;
  u64 file_hash = 0;

  static constexpr u32 SA_DENS = 32, ISA_DENS = 64;

  // ---- rank_support_v<1,1>::rank ----
  u64 rank1(u64 idx) const {
    const u64* p = rank_bb.data() + ((idx >> 8) & ~1ULL);
    u64 r = p[0] + ((p[1] >> (63 - 9 * ((idx & 0x1FF) >> 6))) & 0x1FF);
    if (idx & 0x3F) r += (u64)__builtin_popcountll(bv[idx >> 6] & ((1ULL << (idx & 63)) - 1));
    return r;
  }
  bool bit(u64 i) const { return (bv[i >> 6] >> (i & 63)) & 1; }

  // ---- wt_pc::rank(i,c): occurrences of c in BWT[0,i) ----
  u64 wt_rank(u64 i, u8 c) const {
    if (c_to_leaf[c] == 0xFFFF) return 0;
    u64 p = path[c];
    u32 len = (u32)(p >> 56);
    u64 res = i;
    u32 v = 0;
    for (u32 l = 0; l < len && res; ++l, p >>= 1) {
      u64 ones = rank1(nodes[v].bv_pos + res) - nodes[v].bv_pos_rank;
      res = (p & 1) ? ones : res - ones;
      v = nodes[v].child[p & 1];
    }
    return res;
  }
  // ---- wt_pc::inverse_select(i): (rank of BWT[i] in BWT[0,i), BWT[i]) ----
  std::pair<u64, u8> inverse_select(u64 i) const {
    u32 v = 0;
    while (nodes[v].child[0] != 0xFFFF) {
      u64 pos = nodes[v].bv_pos + i;
      u64 ones = rank1(pos) - nodes[v].bv_pos_rank;
      if (bit(pos)) {
        i = ones;
        v = nodes[v].child[1];
      } else {
        i -= ones;
        v = nodes[v].child[0];
      }
    }
    return {i, (u8)nodes[v].bv_pos_rank};
  }
  u64 lf(u64 i) const {
    auto rc = inverse_select(i);
    return C[char2comp[rc.second]] + rc.first;
  }

  // ---- backward_search, one symbol (suffix_array_algorithm.hpp) ----
  // closed interval [l,r]; returns size
  u64 bs_step(u64 l, u64 r, u8 c, u64& lo, u64& ro) const {
    u64 cc = char2comp[c];
    if (cc == 0 && c > 0) {
      lo = 1;
      ro = 0;
      return 0;
    }
    u64 cb = C[cc];
    if (l == 0 && r + 1 == n) {
      lo = cb;
      ro = C[cc + 1] - 1;
    } else {
      lo = cb + wt_rank(l, c);
      ro = cb + wt_rank(r + 1, c) - 1;
    }
    return ro + 1 - lo;
  }
  u64 backward_search(const u8* pat, u64 m, u64& l, u64& r) const {
    l = 0;
    r = n - 1;
    u64 e = m;
    while (e > 0 && r + 1 - l > 0) {
      --e;
      bs_step(l, r, pat[e], l, r);
    }
    return r + 1 - l;
  }
  u64 count(const u8* pat, u64 m) const {
    if (m > n) return 0;
    u64 l, r;
    return backward_search(pat, m, l, r);
  }
  // csa_wt::operator[]
  u64 sa(u64 i) const {
    u64 off = 0;
    while (i % SA_DENS) {
      i = lf(i);
      ++off;
    }
    u64 v = sa_samples.get(i / SA_DENS);
    return v + off < n ? v + off : v + off - n;
  }
  std::vector<u64> locate(const u8* pat, u64 m) const {
    std::vector<u64> occ;
    u64 l, r;
    u64 k = (m > n) ? 0 : backward_search(pat, m, l, r);
    occ.resize(k);
    for (u64 i = 0; i < k; ++i) occ[i] = sa(l + i);
    return occ;
  }
  // isa_of_csa_wt::operator[] via sample_qeq
  u64 isa(u64 i) const {
    u64 ci = (i / ISA_DENS + 1) % isa_samples.size();
    u64 res = isa_samples.get(ci), spos = ci * ISA_DENS;
    u64 steps = (spos < i) ? spos + n - i : spos - i;
    while (steps--) res = lf(res);
    return res;
  }
  // extract(csa, b, e) inclusive
  std::string extract(u64 b, u64 e) const {
    std::string out(e - b + 1, '\0');
    u64 steps = e - b + 1;
    u64 order = isa(e);
    // first_row_symbol(order): the comp whose C-range contains order
    u64 cc = (u64)(std::upper_bound(C.begin(), C.end(), order) - C.begin()) - 1;
    out[--steps] = (char)comp2char[cc];
    while (steps) {
      auto rc = inverse_select(order);
      order = C[char2comp[rc.second]] + rc.first;
      out[--steps] = (char)rc.second;
    }
    return out;
  }
};

// ---------------------------------------------------------------------------------------------
// Construction (index.h:97-123 text definition + sdsl::construct semantics)
// ---------------------------------------------------------------------------------------------

// Text of the index from FASTA lines (index.h:105-113): header lines contribute '\n' (except the
// first), sequence lines are upper-cased and concatenated, a final '\n' closes the text.
inline std::string text_from_fasta_lines(const std::vector<std::string>& lines) {
  std::string t;
  bool first = true;
  for (const auto& ln : lines) {
    if (!ln.empty() && ln[0] == '>') {
      if (!first) t.push_back('\n');
      first = false;
    } else {
      for (char ch : ln) t.push_back((char)std::toupper((unsigned char)ch));
    }
  }
  t.push_back('\n');
  return t;
}

// Suffix array of S = T + '\0' by prefix doubling (test-sized inputs only).
inline std::vector<u32> build_sa(const std::string& T) {
  const u64 n = T.size() + 1;
  std::vector<u32> sa(n), rk(n), tmp(n);
  auto at = [&](u64 i) -> u64 { return i < T.size() ? (u8)T[i] : 0; };
  std::vector<u64> key(n);
  for (u64 i = 0; i < n; ++i) {
    u64 k = 0;
    for (u32 j = 0; j < 8; ++j) k = (k << 8) | (i + j < n ? at(i + j) : 0);
    key[i] = k;
    sa[i] = (u32)i;
  }
  // NOTE: padding with 0 beyond the end is safe because the sentinel 0 is unique and smallest.
  std::sort(sa.begin(), sa.end(), [&](u32 a, u32 b) { return key[a] < key[b] || (key[a] == key[b] && a > b); });
  // ties above are resolved arbitrarily; ranks only use key equality
  rk[sa[0]] = 0;
  bool uniq = true;
  for (u64 i = 1; i < n; ++i) {
    bool same = key[sa[i]] == key[sa[i - 1]];
    rk[sa[i]] = same ? rk[sa[i - 1]] : (u32)i;
    uniq &= !same;
  }
  for (u64 h = 8; !uniq; h <<= 1) {
    for (u64 i = 0; i < n; ++i) key[i] = ((u64)rk[i] << 32) | (i + h < n ? (u64)rk[i + h] + 1 : 0);
    std::sort(sa.begin(), sa.end(), [&](u32 a, u32 b) { return key[a] < key[b]; });
    tmp[sa[0]] = 0;
    uniq = true;
    for (u64 i = 1; i < n; ++i) {
      bool same = key[sa[i]] == key[sa[i - 1]];
      tmp[sa[i]] = same ? tmp[sa[i - 1]] : (u32)i;
      uniq &= !same;
    }
    rk.swap(tmp);
  }
  return sa;
}

inline void build_select(const std::vector<u64>& bv, u64 bv_bits, bool want, SelectMcl& s) {
  // select_support_mcl::init_slow restated (SURVEY App. A "select_support_mcl")
  u64 cap = ((bv_bits + 63) >> 6) << 6;
  u32 logn = hi_bit(cap) + 1;
  u64 logn2 = (u64)logn * logn, logn4 = logn2 * logn2;
  u64 cnt = 0;
  for (u64 i = 0; i < bv_bits; ++i) cnt += (((bv[i >> 6] >> (i & 63)) & 1) == (u64)want);
  s.arg_cnt = cnt;
  s.any_long = false;
  s.blocks.clear();
  s.is_mini.clear();
  if (!cnt) return;
  const u64 SB = 4096;
  u64 sb = (cnt + SB - 1) / SB;
  s.superblock.init(sb, (u8)logn);
  s.blocks.resize(sb);
  s.is_mini.assign(sb, 1);
  std::vector<u64> pos(SB);
  u64 seen = 0, sbi = 0;
  for (u64 i = 0; i < bv_bits; ++i) {
    if ((((bv[i >> 6] >> (i & 63)) & 1) == (u64)want)) {
      pos[seen % SB] = i;
      ++seen;
      if (seen % SB == 0 || seen == cnt) {
        u64 last = (seen - 1) % SB;
        s.superblock.set(sbi, pos[0]);
        u64 diff = pos[last] - pos[0];
        if (diff > logn4) {
          s.any_long = true;
          s.is_mini[sbi] = 0;
          s.blocks[sbi].init(SB, (u8)(hi_bit(pos[last]) + 1));
          for (u64 j = 0; j <= last; ++j) s.blocks[sbi].set(j, pos[j]);
        } else {
          s.blocks[sbi].init(64, (u8)(hi_bit(diff) + 1));
          for (u64 j = 0; j <= last; j += 64) s.blocks[sbi].set(j / 64, pos[j] - pos[0]);
        }
        ++sbi;
      }
    }
  }
}

inline void build_rank(const std::vector<u64>& bv, u64 bv_bits, std::vector<u64>& bb) {
  // rank_support_v<1,1> constructor restated
  u64 cap = ((bv_bits + 63) >> 6) << 6;
  u64 words = cap >> 6;
  bb.assign(((cap >> 9) + 1) << 1, 0);
  if (!words) return;
  u64 j = 0, sum = (u64)__builtin_popcountll(bv[0]), second = 0;
  bb[0] = bb[1] = 0;
  u64 i;
  for (i = 1; i < words; ++i) {
    if (!(i & 7)) {
      j += 2;
      bb[j - 1] = second;
      bb[j] = bb[j - 2] + sum;
      second = sum = 0;
    } else {
      second |= sum << (63 - 9 * (i & 7));
    }
    sum += (u64)__builtin_popcountll(bv[i]);
  }
  if (i & 7) {
    second |= sum << (63 - 9 * (i & 7));
    bb[j + 1] = second;
  } else {
    j += 2;
    bb[j - 1] = second;
    bb[j] = bb[j - 2] + sum;
    bb[j + 1] = 0;
  }
}

// Build a csa_wt<> over T (T must not contain '\0').
inline Csa build_csa(const std::string& T) {
  Csa c;
  const u64 n = T.size() + 1;
  c.n = n;
  std::vector<u32> sa = build_sa(T);
  std::vector<u8> bwt(n);
  for (u64 i = 0; i < n; ++i) {
    u64 p = sa[i] ? sa[i] - 1 : n - 1;
    bwt[i] = p < T.size() ? (u8)T[p] : 0;
  }
  // byte_alphabet
  std::vector<u64> freq(256, 0);
  for (u64 i = 0; i < n; ++i) ++freq[bwt[i]];
  std::memset(c.char2comp, 0, sizeof c.char2comp);
  c.sigma = 0;
  c.comp2char.clear();
  c.C.clear();
  c.C.push_back(0);
  for (int ch = 0; ch < 256; ++ch)
    if (freq[ch]) {
      c.char2comp[ch] = (u8)c.sigma;
      c.comp2char.push_back((u8)ch);
      c.C.push_back(c.C.back() + freq[ch]);
      ++c.sigma;
    }
  c.wt_sigma = c.sigma;
  // huff_shape::construct_tree: leaves in byte order, min-heap on (freq, node id)
  struct Tmp {
    u64 freq, sym;
    u16 parent, child[2];
  };
  std::vector<Tmp> tn;
  typedef std::pair<u64, u64> P;
  std::priority_queue<P, std::vector<P>, std::greater<P>> pq;
  for (int ch = 0; ch < 256; ++ch)
    if (freq[ch]) {
      pq.push(P(freq[ch], tn.size()));
      tn.push_back(Tmp{freq[ch], (u64)ch, 0xFFFF, {0xFFFF, 0xFFFF}});
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
  // _byte_tree ctor: BFS renumbering, bv_pos = running bit offset of inner nodes
  c.nodes.assign(tn.size(), WtNode());
  auto cp = [&](u32 dst, u32 src) {
    c.nodes[dst].bv_pos = tn[src].freq;
    c.nodes[dst].bv_pos_rank = tn[src].sym;
    c.nodes[dst].parent = tn[src].parent;
    c.nodes[dst].child[0] = tn[src].child[0];
    c.nodes[dst].child[1] = tn[src].child[1];
  };
  cp(0, (u32)tn.size() - 1);
  c.nodes[0].parent = 0xFFFF;
  u64 bv_size = 0;
  u32 node_cnt = 1;
  std::deque<u32> q;
  q.push_back(0);
  while (!q.empty()) {
    u32 idx = q.front();
    q.pop_front();
    u64 frq = c.nodes[idx].bv_pos;
    c.nodes[idx].bv_pos = bv_size;
    if (c.nodes[idx].child[0] != 0xFFFF) {
      bv_size += frq;
      for (u32 k = 0; k < 2; ++k) {
        cp(node_cnt, c.nodes[idx].child[k]);
        c.nodes[node_cnt].parent = (u16)idx;
        q.push_back(node_cnt);
        c.nodes[idx].child[k] = (u16)node_cnt++;
      }
    }
  }
  for (int i = 0; i < 256; ++i) {
    c.c_to_leaf[i] = 0xFFFF;
    c.path[i] = 0;
  }
  for (u32 v = 0; v < c.nodes.size(); ++v)
    if (c.nodes[v].child[0] == 0xFFFF) c.c_to_leaf[(u8)c.nodes[v].bv_pos_rank] = (u16)v;
  for (int ch = 0; ch < 256; ++ch)
    if (c.c_to_leaf[ch] != 0xFFFF) {
      u32 v = c.c_to_leaf[ch];
      u64 pw = 0, pl = 0;
      while (v != 0) {
        pw <<= 1;
        if (c.nodes[c.nodes[v].parent].child[1] == v) pw |= 1;
        v = c.nodes[v].parent;
        ++pl;
      }
      c.path[ch] = pw | (pl << 56);
    }
  // bit vector: every BWT symbol walks its path, appending one bit per inner node
  c.bv_bits = bv_size;
  c.bv.assign((bv_size + 63) / 64, 0);
  std::vector<u64> fill(c.nodes.size());
  for (u32 v = 0; v < c.nodes.size(); ++v) fill[v] = c.nodes[v].bv_pos;
  for (u64 i = 0; i < n; ++i) {
    u64 p = c.path[bwt[i]];
    u32 len = (u32)(p >> 56), v = 0;
    for (u32 l = 0; l < len; ++l, p >>= 1) {
      u64 at = fill[v]++;
      if (p & 1) c.bv[at >> 6] |= 1ULL << (at & 63);
      v = c.nodes[v].child[p & 1];
    }
  }
  build_rank(c.bv, c.bv_bits, c.rank_bb);
  // init_node_ranks: inner nodes get rank1(bv_pos)
  for (u32 v = 0; v < c.nodes.size(); ++v)
    if (c.nodes[v].child[0] != 0xFFFF) c.nodes[v].bv_pos_rank = c.rank1(c.nodes[v].bv_pos);
  build_select(c.bv, c.bv_bits, true, c.sel1);
  build_select(c.bv, c.bv_bits, false, c.sel0);
  // samples
  u8 wd = (u8)(hi_bit(n) + 1);
  c.sa_samples.init((n + Csa::SA_DENS - 1) / Csa::SA_DENS, wd);
  for (u64 i = 0; i < n; i += Csa::SA_DENS) c.sa_samples.set(i / Csa::SA_DENS, sa[i]);
  c.isa_samples.init((n - 1) / Csa::ISA_DENS + 1, wd);
  for (u64 i = 0; i < n; ++i)
    if (sa[i] % Csa::ISA_DENS == 0) c.isa_samples.set(sa[i] / Csa::ISA_DENS, i);
  return c;
}

// ---------------------------------------------------------------------------------------------
// Serialisation (SURVEY App. A order) — store_to_checked_file: 8-byte class hash, then members.
// ---------------------------------------------------------------------------------------------
struct Out {
  std::vector<u8> b;
  void raw(const void* p, size_t n) {
    const u8* q = (const u8*)p;
    b.insert(b.end(), q, q + n);
  }
  void u64_(u64 v) { raw(&v, 8); }
  void u16_(u16 v) { raw(&v, 2); }
  void u8_(u8 v) { raw(&v, 1); }
  void iv_fixed(const IntVec& v) {  // int_vector<w>, w != 0: bit length + words
    u64_(v.bits);
    raw(v.w.data(), v.w.size() * 8);
  }
  void iv0(const IntVec& v) {  // int_vector<0>: bit length + width byte + words
    u64_(v.bits);
    u8_(v.width);
    raw(v.w.data(), v.w.size() * 8);
  }
};

// The genuine value is std::hash<std::string> of sdsl's demangled class name — not reproducible here.
static const u64 FM9_PLACEHOLDER_HASH = 0x44494345594F5243ULL;  // "DICEYORC"

inline void serialize_select(Out& o, const SelectMcl& s) {
  o.u64_(s.arg_cnt);
  if (!s.arg_cnt) return;
  o.iv0(s.superblock);
  u64 sb = (s.arg_cnt + 4095) >> 12;
  IntVec mol;
  mol.width = 1;
  if (s.any_long) {
    mol.init(sb, 1);
    for (u64 i = 0; i < sb; ++i) mol.set(i, s.is_mini[i]);
  } else {
    mol.bits = 0;
  }
  o.iv_fixed(mol);
  for (u64 i = 0; i < sb; ++i) o.iv0(s.blocks[i]);
}

inline std::vector<u8> serialize_csa(const Csa& c, u64 hash = FM9_PLACEHOLDER_HASH) {
  Out o;
  o.u64_(hash);
  // wt_pc
  o.u64_(c.n);
  o.u64_(c.wt_sigma);
  o.u64_(c.bv_bits);
  o.raw(c.bv.data(), c.bv.size() * 8);
  o.u64_((u64)c.rank_bb.size() * 64);
  o.raw(c.rank_bb.data(), c.rank_bb.size() * 8);
  serialize_select(o, c.sel1);
  serialize_select(o, c.sel0);
  o.u64_((u64)c.nodes.size());
  for (const auto& nd : c.nodes) {
    o.u64_(nd.bv_pos);
    o.u64_(nd.bv_pos_rank);
    o.u16_(nd.parent);
    o.u16_(nd.child[0]);
    o.u16_(nd.child[1]);
  }
  o.raw(c.c_to_leaf, sizeof c.c_to_leaf);
  o.raw(c.path, sizeof c.path);
  // samples
  o.iv0(c.sa_samples);
  o.iv0(c.isa_samples);
  // byte_alphabet
  IntVec c2c;
  c2c.init(256, 8);
  for (int i = 0; i < 256; ++i) c2c.set(i, c.char2comp[i]);
  o.iv_fixed(c2c);
  IntVec cc;
  cc.init(c.comp2char.size(), 8);
  for (size_t i = 0; i < c.comp2char.size(); ++i) cc.set(i, c.comp2char[i]);
  o.iv_fixed(cc);
  IntVec C;
  C.init(c.C.size(), 64);
  for (size_t i = 0; i < c.C.size(); ++i) C.set(i, c.C[i]);
  o.iv_fixed(C);
  o.u16_(c.sigma);
  return o.b;
}

struct In {
  const u8* p;
  size_t n, off = 0;
  void need(size_t k) {
    if (off + k > n) throw std::runtime_error("fm9: truncated");
  }
  u64 u64_() {
    need(8);
    u64 v;
    std::memcpy(&v, p + off, 8);
    off += 8;
    return v;
  }
  u16 u16_() {
    need(2);
    u16 v;
    std::memcpy(&v, p + off, 2);
    off += 2;
    return v;
  }
  u8 u8_() {
    need(1);
    return p[off++];
  }
  void words(std::vector<u64>& w, u64 bits) {
    u64 k = (bits + 63) / 64;
    need(k * 8);
    w.resize(k);
    std::memcpy(w.data(), p + off, k * 8);
    off += k * 8;
  }
  void iv_fixed(IntVec& v, u8 width) {
    v.width = width;
    v.bits = u64_();
    words(v.w, v.bits);
  }
  void iv0(IntVec& v) {
    v.bits = u64_();
    v.width = u8_();
    words(v.w, v.bits);
  }
};

inline void parse_select(In& in, SelectMcl& s) {
  s.arg_cnt = in.u64_();
  s.blocks.clear();
  s.is_mini.clear();
  s.any_long = false;
  if (!s.arg_cnt) return;
  in.iv0(s.superblock);
  u64 sb = (s.arg_cnt + 4095) >> 12;
  IntVec mol;
  in.iv_fixed(mol, 1);
  s.any_long = mol.bits != 0;
  s.is_mini.assign(sb, 1);
  if (s.any_long)
    for (u64 i = 0; i < sb; ++i) s.is_mini[i] = (u8)mol.get(i);
  s.blocks.resize(sb);
  for (u64 i = 0; i < sb; ++i) in.iv0(s.blocks[i]);
}

inline Csa parse_csa(const u8* data, size_t len) {
  Csa c;
  In in{data, len};
  c.file_hash = in.u64_();
  c.n = in.u64_();
  c.wt_sigma = in.u64_();
  c.bv_bits = in.u64_();
  in.words(c.bv, c.bv_bits);
  u64 rb = in.u64_();
  in.words(c.rank_bb, rb);
  parse_select(in, c.sel1);
  parse_select(in, c.sel0);
  u64 nn = in.u64_();
  if (nn > 511) throw std::runtime_error("fm9: node count");
  c.nodes.resize(nn);
  for (auto& nd : c.nodes) {
    nd.bv_pos = in.u64_();
    nd.bv_pos_rank = in.u64_();
    nd.parent = in.u16_();
    nd.child[0] = in.u16_();
    nd.child[1] = in.u16_();
  }
  in.need(sizeof c.c_to_leaf + sizeof c.path);
  std::memcpy(c.c_to_leaf, data + in.off, sizeof c.c_to_leaf);
  in.off += sizeof c.c_to_leaf;
  std::memcpy(c.path, data + in.off, sizeof c.path);
  in.off += sizeof c.path;
  in.iv0(c.sa_samples);
  in.iv0(c.isa_samples);
  IntVec t;
  in.iv_fixed(t, 8);
  if (t.size() != 256) throw std::runtime_error("fm9: char2comp");
  for (int i = 0; i < 256; ++i) c.char2comp[i] = (u8)t.get(i);
  in.iv_fixed(t, 8);
  c.comp2char.resize(t.size());
  for (size_t i = 0; i < c.comp2char.size(); ++i) c.comp2char[i] = (u8)t.get(i);
  in.iv_fixed(t, 64);
  c.C.resize(t.size());
  for (size_t i = 0; i < c.C.size(); ++i) c.C[i] = t.get(i);
  c.sigma = in.u16_();
  if (in.off != len) throw std::runtime_error("fm9: trailing bytes");
  if (c.C.size() != (size_t)c.sigma + 1 || c.C.back() != c.n) throw std::runtime_error("fm9: C[] inconsistent");
  return c;
}

inline std::vector<u8> read_file(const std::string& path) {
  FILE* f = std::fopen(path.c_str(), "rb");
  if (!f) throw std::runtime_error("cannot open " + path);
  std::fseek(f, 0, SEEK_END);
  long sz = std::ftell(f);
  std::fseek(f, 0, SEEK_SET);
  std::vector<u8> b((size_t)sz);
  size_t got = sz ? std::fread(b.data(), 1, (size_t)sz, f) : 0;
  std::fclose(f);
  if (got != (size_t)sz) throw std::runtime_error("short read " + path);
  return b;
}
inline void write_file(const std::string& path, const std::vector<u8>& b) {
  FILE* f = std::fopen(path.c_str(), "wb");
  if (!f) throw std::runtime_error("cannot write " + path);
  if (!b.empty() && std::fwrite(b.data(), 1, b.size(), f) != b.size()) {
    std::fclose(f);
    throw std::runtime_error("short write " + path);
  }
  std::fclose(f);
}

}  // namespace orc
