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

// Tries the checked layout (8-byte hash first) and, failing full byte accounting, the plain store_to_file layout.
inline int sdsl_parse(SdslCsa& c, size_t start_off) {
  sdslio::Cursor cur{c.base, c.len, start_off};
  c.n = cur.u64v();
  c.wt_sigma = cur.u64v();
  if (!cur.ok || c.n < 2 || c.wt_sigma == 0 || c.wt_sigma > 256) return DG_EFORMAT;
  if (!cur.int_vector_fixed(c.bv, 1)) return DG_EFORMAT;
  if (!cur.int_vector_fixed(c.rank, 64)) return DG_EFORMAT;
  // rank_support_v holds two words per 512-bit superblock, ((capacity>>9)+1)*2 of them.  How sdsl rounds the capacity
  // differs between releases, so only what rank() will index is demanded: the superblock of the last bit.
  if (c.rank.nwords < (((c.bv.bits >> 9) + 1) << 1)) return DG_EFORMAT;
  if (!cur.skip_select() || !cur.skip_select()) return DG_EFORMAT;
  u64 nn = cur.u64v();
  if (!cur.ok || nn == 0 || nn > 511 || nn != 2 * c.wt_sigma - 1) return DG_EFORMAT;
  c.nodes.resize(nn);
  for (auto& nd : c.nodes) {
    nd.bv_pos = cur.u64v();
    nd.bv_pos_rank = cur.u64v();
    cur.take(&nd.parent, 2);
    cur.take(&nd.child[0], 2);
    cur.take(&nd.child[1], 2);
  }
  cur.take(c.c_to_leaf, sizeof c.c_to_leaf);
  cur.take(c.path, sizeof c.path);
  if (!cur.int_vector0(c.sa_samples) || !cur.int_vector0(c.isa_samples)) return DG_EFORMAT;
  Span t;
  if (!cur.int_vector_fixed(t, 8) || t.bits != 2048) return DG_EFORMAT;
  std::memcpy(c.char2comp, t.w, 256);
  if (!cur.int_vector_fixed(t, 8)) return DG_EFORMAT;
  c.comp2char.assign((const u8*)t.w, (const u8*)t.w + t.bits / 8);
  if (!cur.int_vector_fixed(t, 64)) return DG_EFORMAT;
  c.C.resize(t.nwords);
  for (u64 i = 0; i < t.nwords; ++i) c.C[i] = sdslio::load_u64(t.w + i);
  cur.take(&c.sigma, 2);
  if (!cur.ok || cur.at != c.len) return DG_EFORMAT;
  // semantic cross-checks
  if (c.sigma != c.wt_sigma || c.comp2char.size() != c.sigma || c.C.size() != (size_t)c.sigma + 1) return DG_EFORMAT;
  if (c.C[0] != 0 || c.C[c.sigma] != c.n) return DG_EFORMAT;
  for (u32 i = 0; i < c.sigma; ++i)
    if (c.C[i] > c.C[i + 1]) return DG_EFORMAT;
  u8 want_w = (u8)(64 - __builtin_clzll(c.n));
  // sdsl stores the samples with bits::hi(n)+1 bits; a wider (e.g. byte-aligned) vector holds the same values
  if (c.sa_samples.width < want_w || c.sa_samples.width > 64 || c.isa_samples.width != c.sa_samples.width) return DG_EFORMAT;
  want_w = c.sa_samples.width;
  if (c.sa_samples.bits / want_w != (c.n + 31) / 32) return DG_EFORMAT;
  if (c.isa_samples.bits / want_w != (c.n - 1) / 64 + 1) return DG_EFORMAT;
  for (const auto& nd : c.nodes) {
    bool leaf = nd.child[0] == 0xFFFF;
    if (!leaf && (nd.child[0] >= nn || nd.child[1] >= nn || nd.bv_pos > c.bv.bits)) return DG_EFORMAT;
  }
  return DG_OK;
}

inline int sdsl_open(const char* path, SdslCsa& c) {
  c.fd = open(path, O_RDONLY);
  if (c.fd < 0) return fail(DG_EIO, "cannot open %s", path);
  struct stat st;
  if (fstat(c.fd, &st) != 0 || st.st_size < 64) return fail(DG_EIO, "cannot stat %s (or file too small)", path);
  c.len = (size_t)st.st_size;
  void* m = mmap(nullptr, c.len, PROT_READ, MAP_PRIVATE, c.fd, 0);
  if (m == MAP_FAILED) return fail(DG_EIO, "mmap failed for %s", path);
  c.base = (const u8*)m;
  if (sdsl_parse(c, 8) == DG_OK) return DG_OK;   // store_to_checked_file: hash + object
  if (sdsl_parse(c, 0) == DG_OK) return DG_OK;   // store_to_file: object only
  return fail(DG_EFORMAT, "%s is not an sdsl csa_wt<wt_huff<>,32,64> file (byte accounting failed)", path);
}

}  // namespace dg
