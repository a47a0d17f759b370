// Reader for the file `dicey index` writes: sdsl::store_to_checked_file(csa_wt<>) (reference src/index.h:122;
// consumed by load_from_checked_file at src/hunter.h:256, src/silica.h:343).
//
// The file is mapped and described in place — nothing is converted: the device gets the very words sdsl wrote.
// Layout (sdsl-lite csa_wt::serialize order):
//   u64 class-name hash | wt_huff{ u64 size, u64 sigma, bit_vector bv, rank_support_v, select_support_mcl x2,
//   byte_tree } | int_vector<0> sa_samples | int_vector<0> isa_samples | byte_alphabet{char2comp, comp2char, C, sigma}
// Every byte must be accounted for; any mismatch is DG_EFORMAT.
#pragma once
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cstdarg>
#include <string>

#include "common.hpp"

namespace dg {

struct Span {  // a run of 64-bit words inside the mapping
  const u64* w = nullptr;
  u64 nwords = 0;
  u64 bits = 0;
  u8 width = 0;  // element width for int_vector<0>
};

struct SdslTreeNode {
  u64 bv_pos, bv_pos_rank;
  u16 parent, child[2];
};

struct SdslSection {  // where a section of the file lies (diagnostics, dg_fm9_check)
  const char* name;
  u64 offset, bytes;
};

struct SdslCsa {
  // mapping
  const u8* base = nullptr;
  size_t len = 0;
  int fd = -1;
  // wavelet tree
  u64 n = 0, wt_sigma = 0;
  Span bv, rank;
  std::vector<SdslTreeNode> nodes;
  u16 c_to_leaf[256];
  u64 path[256];
  // samples
  Span sa_samples, isa_samples;
  // alphabet
  u8 char2comp[256];
  std::vector<u8> comp2char;
  std::vector<u64> C;
  u16 sigma = 0;
  // diagnostics of the last parse: the sections that were accounted for, and — when it failed — the first section whose byte
  // count or invariant is off, with the file offset and what was expected (a foreign file must fail by NAME, not by "bad file")
  std::vector<SdslSection> sections;
  std::string why;
  u64 fail_at = 0;
  bool checked_layout = false;  // the 8-byte class-name hash of store_to_checked_file is in front

  ~SdslCsa() { unmap(); }
  void unmap() {
    if (base) munmap((void*)base, len);
    if (fd >= 0) close(fd);
    base = nullptr;
    fd = -1;
  }
};

namespace sdslio {

struct Cursor {
  const u8* p;
  size_t n, at;
  bool ok = true;
  bool take(void* dst, size_t k) {
    if (!ok || at + k > n) return ok = false;
    std::memcpy(dst, p + at, k);
    at += k;
    return true;
  }
  u64 u64v() {
    u64 v = 0;
    take(&v, 8);
    return v;
  }
  bool words(Span& s, u64 bits) {
    s.bits = bits;
    s.nwords = (bits + 63) >> 6;
    if (!ok || at + s.nwords * 8 > n) return ok = false;
    // not 8-byte aligned in general (1-byte width fields precede it): consumers memcpy / hipMemcpy from it
    s.w = (const u64*)(p + at);
    at += s.nwords * 8;
    return true;
  }
  bool int_vector_fixed(Span& s, u8 width) {
    u64 bits = u64v();
    s.width = width;
    return words(s, bits);
  }
  bool int_vector0(Span& s) {
    u64 bits = u64v();
    u8 wd = 0;
    take(&wd, 1);
    s.width = wd;
    return words(s, bits);
  }
  // select_support_mcl<b,1>: skipped, but walked exactly so that byte accounting stays strict
  bool skip_select() {
    u64 arg_cnt = u64v();
    if (!ok) return false;
    if (!arg_cnt) return true;
    Span tmp;
    if (!int_vector0(tmp)) return false;  // superblock
    u64 sb = (arg_cnt + 4095) >> 12;
    Span mol;
    if (!int_vector_fixed(mol, 1)) return false;  // mini_or_long (may be empty)
    if (mol.bits != 0 && mol.bits != sb) return ok = false;
    for (u64 i = 0; i < sb; ++i)
      if (!int_vector0(tmp)) return false;
    return true;
  }
};

static inline u64 load_u64(const u64* p) {
  u64 v;
  std::memcpy(&v, p, 8);
  return v;
}

}  // namespace sdslio

// Parses csa_wt<>::serialize from `start_off` (8: behind the hash of store_to_checked_file, 0: store_to_file).  On failure c.why
// names the section and c.fail_at is the file offset the parse had reached.
inline int sdsl_parse(SdslCsa& c, size_t start_off) {
  sdslio::Cursor cur{c.base, c.len, start_off};
  c.sections.clear();
  c.why.clear();
  c.nodes.clear();
  const char* sec = "header";
  u64 sec_at = start_off;
  char msg[320];
  auto bad = [&](const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    std::vsnprintf(msg, sizeof msg, fmt, ap);
    va_end(ap);
    char full[480];
    std::snprintf(full, sizeof full, "section '%s' (file offset %llu): %s", sec, (unsigned long long)sec_at, msg);
    c.why = full;
    c.fail_at = cur.at;
    return DG_EFORMAT;
  };
  auto begin = [&](const char* name) {
    sec = name;
    sec_at = cur.at;
  };
  auto end = [&]() { c.sections.push_back(SdslSection{sec, sec_at, cur.at - sec_at}); };
  auto again = [&](const char* name) {  // a section that has been read: its invariants against the others
    sec = name;
    for (const auto& sx : c.sections)
      if (!std::strcmp(sx.name, name)) sec_at = sx.offset;
  };
  auto short_file = [&](const char* what) { return bad("%s needs bytes beyond the end of the file (%llu bytes)", what, (unsigned long long)c.len); };

  begin("wt header (size, sigma)");
  c.n = cur.u64v();
  c.wt_sigma = cur.u64v();
  if (!cur.ok) return short_file("the header");
  if (c.n < 2 || c.n > (1ULL << 40)) return bad("text size %llu is not plausible", (unsigned long long)c.n);
  if (c.wt_sigma == 0 || c.wt_sigma > 256) return bad("alphabet size %llu (a byte alphabet has 1..256 symbols)", (unsigned long long)c.wt_sigma);
  end();
  begin("wt bit_vector");
  if (!cur.int_vector_fixed(c.bv, 1)) return short_file("a bit vector of the announced length");
  if (c.bv.bits > 8 * c.n) return bad("%llu bits for a text of %llu symbols (a Huffman-shaped tree of bytes has at most 8 per symbol)",
                                      (unsigned long long)c.bv.bits, (unsigned long long)c.n);
  end();
  begin("wt rank_support_v");
  if (!cur.int_vector_fixed(c.rank, 64)) return short_file("the rank words");
  // rank_support_v holds two words per 512-bit superblock, ((capacity>>9)+1)*2 of them.  How sdsl rounds the capacity
  // differs between releases, so only what rank() will index is demanded: the superblock of the last bit.
  if (c.rank.nwords < (((c.bv.bits >> 9) + 1) << 1))
    return bad("%llu words, rank() over %llu bits indexes %llu (two per 512-bit superblock: is this rank_support_v<1,1>?)",
               (unsigned long long)c.rank.nwords, (unsigned long long)c.bv.bits, (unsigned long long)(((c.bv.bits >> 9) + 1) << 1));
  if (c.rank.nwords > (((c.bv.bits >> 9) + 3) << 1) + 16)
    return bad("%llu words for %llu bits: far more than two per 512-bit superblock (another rank support?)", (unsigned long long)c.rank.nwords,
               (unsigned long long)c.bv.bits);
  end();
  begin("wt select_support_mcl<1>");
  if (!cur.skip_select()) return bad("the block cannot be walked (arg_cnt, superblock vector, mini_or_long bits, one vector per 4 096 arguments)");
  end();
  begin("wt select_support_mcl<0>");
  if (!cur.skip_select()) return bad("the block cannot be walked (arg_cnt, superblock vector, mini_or_long bits, one vector per 4 096 arguments)");
  end();
  begin("wt byte_tree nodes");
  const u64 nn = cur.u64v();
  if (!cur.ok) return short_file("the node count");
  if (nn != 2 * c.wt_sigma - 1)
    return bad("%llu nodes, a Huffman tree over %llu symbols has %llu (are both select supports present in this file?)", (unsigned long long)nn,
               (unsigned long long)c.wt_sigma, (unsigned long long)(2 * c.wt_sigma - 1));
  c.nodes.resize(nn);
  for (auto& nd : c.nodes) {
    nd.bv_pos = cur.u64v();
    nd.bv_pos_rank = cur.u64v();
    cur.take(&nd.parent, 2);
    cur.take(&nd.child[0], 2);
    cur.take(&nd.child[1], 2);
  }
  if (!cur.ok) return short_file("22 bytes per node");
  end();
  begin("wt byte_tree c_to_leaf / path");
  cur.take(c.c_to_leaf, sizeof c.c_to_leaf);
  cur.take(c.path, sizeof c.path);
  if (!cur.ok) return short_file("the 256 leaf numbers and 256 code words");
  end();
  begin("sa_samples");
  if (!cur.int_vector0(c.sa_samples)) return short_file("the vector");
  end();
  begin("isa_samples");
  if (!cur.int_vector0(c.isa_samples)) return short_file("the vector");
  end();
  Span t;
  begin("alphabet char2comp");
  if (!cur.int_vector_fixed(t, 8)) return short_file("the vector");
  if (t.bits != 2048) return bad("%llu bits, expected 2048 (256 bytes)", (unsigned long long)t.bits);
  std::memcpy(c.char2comp, t.w, 256);
  end();
  begin("alphabet comp2char");
  if (!cur.int_vector_fixed(t, 8)) return short_file("the vector");
  c.comp2char.assign((const u8*)t.w, (const u8*)t.w + t.bits / 8);
  end();
  begin("alphabet C");
  if (!cur.int_vector_fixed(t, 64)) return short_file("the vector");
  c.C.resize(t.nwords);
  for (u64 i = 0; i < t.nwords; ++i) c.C[i] = sdslio::load_u64(t.w + i);
  end();
  begin("alphabet sigma");
  cur.take(&c.sigma, 2);
  if (!cur.ok) return short_file("the 16-bit symbol count");
  end();
  begin("end of file");
  if (cur.at != c.len) return bad("%llu trailing bytes behind the last section", (unsigned long long)(c.len - cur.at));
  // ---- what the sections say about each other
  again("alphabet sigma");
  if (c.sigma != c.wt_sigma) return bad("%u symbols, the wavelet tree says %llu", (unsigned)c.sigma, (unsigned long long)c.wt_sigma);
  again("alphabet comp2char");
  if (c.comp2char.size() != c.sigma) return bad("%zu entries for %u symbols", c.comp2char.size(), (unsigned)c.sigma);
  for (u32 i = 0; i < c.sigma; ++i) {
    if (i && c.comp2char[i] <= c.comp2char[i - 1]) return bad("entry %u (byte %u) is not above entry %u (byte %u): symbols are numbered in byte order", i, (unsigned)c.comp2char[i], i - 1, (unsigned)c.comp2char[i - 1]);
    if (c.char2comp[c.comp2char[i]] != i) return bad("char2comp[%u] = %u, comp2char[%u] = %u: not inverse to each other", (unsigned)c.comp2char[i], (unsigned)c.char2comp[c.comp2char[i]], i, (unsigned)c.comp2char[i]);
  }
  again("alphabet C");
  if (c.C.size() != (size_t)c.sigma + 1) return bad("%zu entries, expected sigma + 1 = %u", c.C.size(), (unsigned)c.sigma + 1);
  if (c.C[0] != 0 || c.C[c.sigma] != c.n) return bad("C[0] = %llu, C[sigma] = %llu; expected 0 and the text size %llu", (unsigned long long)c.C[0], (unsigned long long)c.C[c.sigma], (unsigned long long)c.n);
  for (u32 i = 0; i < c.sigma; ++i)
    if (c.C[i] > c.C[i + 1]) return bad("C[%u] = %llu is above C[%u] = %llu", i, (unsigned long long)c.C[i], i + 1, (unsigned long long)c.C[i + 1]);
  again("sa_samples");
  u8 want_w = (u8)(64 - __builtin_clzll(c.n));
  // sdsl stores the samples with bits::hi(n)+1 bits; a wider (e.g. byte-aligned) vector holds the same values
  if (c.sa_samples.width < want_w || c.sa_samples.width > 64)
    return bad("element width %u, values up to %llu need %u..64 bits", (unsigned)c.sa_samples.width, (unsigned long long)c.n - 1, (unsigned)want_w);
  if (c.sa_samples.bits / c.sa_samples.width != (c.n + 31) / 32)
    return bad("%llu entries of %u bits, expected %llu (every 32nd suffix: is the sampling density 32?)", (unsigned long long)(c.sa_samples.bits / c.sa_samples.width),
               (unsigned)c.sa_samples.width, (unsigned long long)((c.n + 31) / 32));
  again("isa_samples");
  if (c.isa_samples.width != c.sa_samples.width)
    return bad("element width %u, sa_samples has %u", (unsigned)c.isa_samples.width, (unsigned)c.sa_samples.width);
  if (c.isa_samples.bits / c.isa_samples.width != (c.n - 1) / 64 + 1)
    return bad("%llu entries, expected %llu (every 64th text position: is the sampling density 64?)", (unsigned long long)(c.isa_samples.bits / c.isa_samples.width),
               (unsigned long long)((c.n - 1) / 64 + 1));
  again("wt byte_tree nodes");
  for (size_t v = 0; v < c.nodes.size(); ++v) {
    const auto& nd = c.nodes[v];
    const bool leaf = nd.child[0] == 0xFFFF;
    if (leaf && nd.child[1] != 0xFFFF) return bad("node %zu has one child", v);
    if (!leaf && (nd.child[0] >= nn || nd.child[1] >= nn)) return bad("node %zu points at children %u / %u of %llu nodes", v, (unsigned)nd.child[0], (unsigned)nd.child[1], (unsigned long long)nn);
    if (!leaf && nd.bv_pos > c.bv.bits) return bad("node %zu starts at bit %llu of a %llu-bit vector", v, (unsigned long long)nd.bv_pos, (unsigned long long)c.bv.bits);
    if (!leaf && (c.nodes[nd.child[0]].parent != v || c.nodes[nd.child[1]].parent != v)) return bad("the children of node %zu do not name it as their parent", v);
  }
  // every symbol's code word leads from the root to its leaf, and the leaf carries the symbol (wt_pc::rank / inverse_select rely on both)
  again("wt byte_tree c_to_leaf / path");
  u32 nleaf = 0;
  for (u32 ch = 0; ch < 256; ++ch) {
    const u16 lf = c.c_to_leaf[ch];
    if (lf == 0xFFFF) {
      if (c.char2comp[ch] != 0 || (c.sigma && c.comp2char[0] == ch)) return bad("byte %u is in the alphabet but has no leaf", ch);
      continue;
    }
    ++nleaf;
    if (lf >= nn || c.nodes[lf].child[0] != 0xFFFF) return bad("byte %u: c_to_leaf names node %u, which is not a leaf", ch, (unsigned)lf);
    if (c.nodes[lf].bv_pos_rank != ch) return bad("byte %u: its leaf (node %u) carries symbol %llu", ch, (unsigned)lf, (unsigned long long)c.nodes[lf].bv_pos_rank);
    const u64 pw = c.path[ch];
    const u32 plen = (u32)(pw >> 56);
    u32 v = 0;
    bool okp = plen <= 56;
    for (u32 k = 0; okp && k < plen; ++k) {
      if (c.nodes[v].child[0] == 0xFFFF) okp = false;
      else v = c.nodes[v].child[(pw >> k) & 1];
    }
    if (!okp || v != lf)
      return bad("byte %u: its code word (%u bits, read from the lowest bit) ends at node %u, its leaf is node %u (another child order or bit order?)", ch, plen, v, (unsigned)lf);
  }
  if (nleaf != c.sigma) return bad("%u bytes have a leaf, the alphabet has %u symbols", nleaf, (unsigned)c.sigma);
  c.fail_at = 0;
  return DG_OK;
}

// The checks that read the big sections (host only, O(file)): rank words against the bit vector's popcounts, node sizes / starts /
// rank offsets of the tree against them, C[] against the leaf sizes, sample values in range and consistent with each other.
// dg_fm9_check and `dicey index --verify` run it; dg_index_open leaves these sections to the device-side checks (index.hip: decoded
// BWT totals against C[], SA samples against the derived suffix array, k_selfcheck).
inline int sdsl_check_deep(SdslCsa& c) {
  char msg[400];
  auto bad = [&](const char* secname, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    std::vsnprintf(msg, sizeof msg, fmt, ap);
    va_end(ap);
    u64 off = 0;
    for (const auto& sx : c.sections)
      if (!std::strcmp(sx.name, secname)) off = sx.offset;
    c.why = std::string("section '") + secname + "' (file offset " + std::to_string(off) + "): " + msg;
    return DG_EFORMAT;
  };
  const u64 nw = c.bv.nwords;
  auto bvw = [&](u64 k) -> u64 {
    u64 x = sdslio::load_u64(c.bv.w + k);
    if (k == nw - 1 && (c.bv.bits & 63)) x &= (1ULL << (c.bv.bits & 63)) - 1;
    return x;
  };
  auto bvz = [&](u64 k) -> u64 { return k < nw ? bvw(k) : 0ULL; };
  // rank_support_v<1,1>: w[2k] = ones before superblock k, w[2k+1] = seven 9-bit counts (ones before word j of the superblock at shift 63 - 9 j)
  u64 ones = 0;
  for (u64 sb = 0; sb <= (c.bv.bits >> 9); ++sb) {
    const u64 abs_w = sdslio::load_u64(c.rank.w + 2 * sb), rel_w = sdslio::load_u64(c.rank.w + 2 * sb + 1);
    if (abs_w != ones)
      return bad("wt rank_support_v", "superblock %llu: %llu ones before it in the file's word, the bit vector has %llu (is this rank_support_v<1,1> with 512-bit superblocks?)",
                 (unsigned long long)sb, (unsigned long long)abs_w, (unsigned long long)ones);
    u64 inside = 0;
    for (u32 j = 0; j < 8; ++j) {
      if (j && sb * 8 + j < nw) {  // (sdsl fills the fields word by word: behind the vector's last word they stay as they were)
        const u64 f = (rel_w >> (63 - 9 * j)) & 0x1FF;
        if (f != inside)
          return bad("wt rank_support_v", "superblock %llu, word %u: the 9-bit field at shift %u says %llu, the bit vector has %llu", (unsigned long long)sb, j, 63 - 9 * j,
                     (unsigned long long)f, (unsigned long long)inside);
      }
      inside += (u64)__builtin_popcountll(bvz(sb * 8 + j));
    }
    ones += inside;
  }
  auto rank1 = [&](u64 x) -> u64 {  // ones in bits [0, x), x <= bits
    const u64 sb = x >> 9;
    u64 r = sdslio::load_u64(c.rank.w + 2 * sb);
    for (u64 k = sb * 8; k < (x >> 6); ++k) r += (u64)__builtin_popcountll(bvz(k));
    if (x & 63) r += (u64)__builtin_popcountll(bvz(x >> 6) & ((1ULL << (x & 63)) - 1));
    return r;
  };
  // node sizes top-down from the root (all n symbols); an inner node owns bits [bv_pos, bv_pos + size), its children get the zeros / ones
  std::vector<u64> size(c.nodes.size(), 0);
  std::vector<u32> order(1, 0);
  size[0] = c.n;
  u64 inner_bits = 0;
  for (size_t q = 0; q < order.size(); ++q) {
    const u32 v = order[q];
    const auto& nd = c.nodes[v];
    if (nd.child[0] == 0xFFFF) continue;
    if (nd.bv_pos + size[v] > c.bv.bits)
      return bad("wt byte_tree nodes", "node %u owns bits [%llu, %llu) of a %llu-bit vector", v, (unsigned long long)nd.bv_pos, (unsigned long long)(nd.bv_pos + size[v]), (unsigned long long)c.bv.bits);
    const u64 r0 = rank1(nd.bv_pos), r1 = rank1(nd.bv_pos + size[v]);
    if (nd.bv_pos_rank != r0)
      return bad("wt byte_tree nodes", "node %u: bv_pos_rank %llu, the bit vector has %llu ones before bit %llu", v, (unsigned long long)nd.bv_pos_rank, (unsigned long long)r0, (unsigned long long)nd.bv_pos);
    size[nd.child[1]] = r1 - r0;
    size[nd.child[0]] = size[v] - (r1 - r0);
    inner_bits += size[v];
    order.push_back(nd.child[0]);
    order.push_back(nd.child[1]);
    if (order.size() > c.nodes.size()) return bad("wt byte_tree nodes", "the child links do not form a tree");
  }
  if (order.size() != c.nodes.size()) return bad("wt byte_tree nodes", "%zu of %zu nodes hang below the root (node 0)", order.size(), c.nodes.size());
  if (inner_bits != c.bv.bits)
    return bad("wt bit_vector", "%llu bits, the tree's inner nodes own %llu (text size %llu)", (unsigned long long)c.bv.bits, (unsigned long long)inner_bits, (unsigned long long)c.n);
  for (u32 i = 0; i < c.sigma; ++i) {
    const u16 lf = c.c_to_leaf[c.comp2char[i]];
    if (size[lf] != c.C[i + 1] - c.C[i])
      return bad("alphabet C", "symbol %u (byte %u) occurs C[%u] - C[%u] = %llu times, its leaf of the wavelet tree holds %llu", i, (unsigned)c.comp2char[i], i + 1, i,
                 (unsigned long long)(c.C[i + 1] - c.C[i]), (unsigned long long)size[lf]);
  }
  // samples: SA[32 k] and ISA[64 k], bit-packed from the lowest bit
  auto get = [&](const Span& sp, u64 i) -> u64 {
    const u64 b = i * sp.width, q = b >> 6, o = b & 63;
    u64 v = sdslio::load_u64(sp.w + q) >> o;
    if (o + sp.width > 64) v |= sdslio::load_u64(sp.w + q + 1) << (64 - o);
    return sp.width == 64 ? v : v & ((1ULL << sp.width) - 1);
  };
  const u64 nsa = c.sa_samples.bits / c.sa_samples.width, nisa = c.isa_samples.bits / c.isa_samples.width;
  if (get(c.sa_samples, 0) != c.n - 1)
    return bad("sa_samples", "entry 0 is %llu; the smallest suffix is the sentinel at position %llu (another sampling order, or another width?)", (unsigned long long)get(c.sa_samples, 0), (unsigned long long)(c.n - 1));
  for (u64 k = 0; k < nsa; ++k)
    if (get(c.sa_samples, k) >= c.n) return bad("sa_samples", "entry %llu is %llu, the text has %llu positions", (unsigned long long)k, (unsigned long long)get(c.sa_samples, k), (unsigned long long)c.n);
  for (u64 k = 0; k < nisa; ++k) {
    const u64 r = get(c.isa_samples, k);
    if (r >= c.n) return bad("isa_samples", "entry %llu is %llu, the text has %llu suffixes", (unsigned long long)k, (unsigned long long)r, (unsigned long long)c.n);
    if ((r & 31) == 0 && get(c.sa_samples, r >> 5) != 64 * k)
      return bad("isa_samples", "entry %llu says position %llu has rank %llu, but sa_samples[%llu] = %llu", (unsigned long long)k, (unsigned long long)(64 * k), (unsigned long long)r,
                 (unsigned long long)(r >> 5), (unsigned long long)get(c.sa_samples, r >> 5));
  }
  return DG_OK;
}

// Tries the checked layout (8-byte hash first) and, failing full byte accounting, the plain store_to_file layout.  When neither
// parses, the message is the one of the attempt that got further into the file.
inline int sdsl_open(const char* path, SdslCsa& c) {
  c.fd = open(path, O_RDONLY);
  if (c.fd < 0) return fail(DG_EIO, "cannot open %s", path);
  struct stat st;
  if (fstat(c.fd, &st) != 0 || st.st_size < 64) return fail(DG_EIO, "cannot stat %s (or file too small)", path);
  c.len = (size_t)st.st_size;
  void* m = mmap(nullptr, c.len, PROT_READ, MAP_PRIVATE, c.fd, 0);
  if (m == MAP_FAILED) return fail(DG_EIO, "mmap failed for %s", path);
  c.base = (const u8*)m;
  c.checked_layout = true;
  if (sdsl_parse(c, 8) == DG_OK) return DG_OK;   // store_to_checked_file: hash + object
  const std::string why8 = c.why;
  const u64 at8 = c.fail_at;
  c.checked_layout = false;
  if (sdsl_parse(c, 0) == DG_OK) return DG_OK;   // store_to_file: object only
  const bool first = at8 >= c.fail_at;
  return fail(DG_EFORMAT, "%s is not an sdsl csa_wt<wt_huff<>,32,64> file as `dicey index` writes it — %s [read as %s; as %s: %s]", path,
              (first ? why8 : c.why).c_str(), first ? "store_to_checked_file (8-byte class hash first)" : "store_to_file (no hash)",
              first ? "store_to_file" : "store_to_checked_file", (first ? c.why : why8).c_str());
}

}  // namespace dg
