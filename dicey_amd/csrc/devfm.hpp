// Device-side view of the FM-index in HBM and the rank/LF primitives every kernel shares.
//
// Two layers live side by side in HBM:
//  (1) the sdsl sections exactly as `dicey index` wrote them (wavelet-tree bit vector, rank_support_v words,
//      byte_tree, bit-packed SA/ISA samples) — used at load time to derive layer 2, and at query time only for
//      symbols outside {A,C,G,T};
//  (2) derived at load: 64-byte Occ blocks (128 BWT symbols as three bit planes + cumulative A/C/G/T counts),
//      the full suffix array (u32) and a byte copy of the text.  One backward-search step = two 64-byte reads.
//
// sdsl arithmetic restated here (rank_support_v::rank, wt_pc::rank / inverse_select) follows SURVEY.md App. A.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace dg {

using u8 = uint8_t;
using u16 = uint16_t;
using u32 = uint32_t;
using u64 = uint64_t;

#define DG_DEV __device__ __forceinline__

// One 64-byte line: counts of A,C,G,T in BWT[0, 128*blk) and the 3-bit codes of BWT[128*blk, 128*blk+128)
// as bit planes (plane b, word w, bit j  <->  bit b of the code at 128*blk + 64*w + j).
// codes: 0..3 = A,C,G,T   4 = N   5 = '\n'   6 = sentinel 0   7 = any other byte (resolved through the wavelet tree)
struct alignas(64) OccBlock {
  u32 cnt[4];
  u64 pl[3][2];
};
static_assert(sizeof(OccBlock) == 64, "Occ block must be one 64-byte line");

enum : u32 { CODE_N = 4, CODE_NL = 5, CODE_NUL = 6, CODE_OTHER = 7 };

struct WtTables {  // small, read with scalar loads
  u64 node_pos[512];
  u64 node_rank[512];  // leaf: the symbol
  u16 child[512][2];
  u64 path[256];  // path>>56 = code length, low bits consumed LSB first
  u16 c_to_leaf[256];
  u64 C[257];
  u8 char2comp[256];
  u8 comp2char[256];
  u32 sigma;
};

struct KFilter {
  const u32* cp[4];
  u32 s[4];
  u32 nr;    // 0 = no filter
  u32 k;     // order: codes have 2k bits
  u64 pick;  // 2 bits per window position t (0 = right-most character): the copy whose in-line bits cover code bits 2t, 2t+1
};

struct FmView {
  u64 n;  // text length + 1
  // sdsl sections
  const u64* bv;
  const u64* rk;
  const WtTables* wt;
  const u64* sa_samp;
  const u64* isa_samp;
  u32 samp_width;
  u64 n_sa_samp, n_isa_samp;
  // derived
  const OccBlock* occ;
  const u32* sa;   // SA[i], i < n
  const u8* text;  // T[0..n), text[n-1] = 0
  u32 C4[4];       // C[] of A,C,G,T (0 and never matching if the symbol is absent)
  u8 sym_of_code[8];
  // K-mer jump table (derived at load): SA interval [lo,hi) of every A/C/G/T K-mer, (0,0) when it does not occur.
  // code = sum over t of code(kmer[K-1-t]) << 2t, i.e. the LAST character sits in the lowest bits — the order in which
  // backward search meets the characters.
  // Entry format (r05): x = lo; y = the interval's WIDTH hi - lo — below 2^16 in bits 0-15 with, in bits 16-30, the preceding
  // characters of the interval's FIRST suffix (pre5[lo], 15 bits; 0x7FFF when pre5 was not built), or bit 31 set and the width in
  // bits 0-30.  Two thirds of the 16-mers that occur in a 3.1 Gb genome occur once: their strings are settled by the table entry
  // alone, without the line of pre5 (k_search1s / k_search2p; ktab_entry() decodes).
  const uint2* ktab;
  u32 K;  // 0 = no table
  // Presence filters (derived at load): bit = "this k-mer occurs", kept in up to four differently permuted copies.  Copy r
  // stores the 512 codes that differ only in code bits [s[r], s[r]+9) in ONE 64-byte line (line index = the remaining
  // 2k-9 bits), so all k-mers that differ from each other only inside one block of four window positions sit in the same
  // line of the copy whose block that is.  The neighbourhood of a query probes k-mers that all differ from the query's
  // window by an edit or two: lanes pick the copy by the position of their edit, and probes of neighbouring edits fall into
  // the same line.  Any copy answers any code: the choice is locality only.
  //   kf   order K (the table's): in front of the table, about one random 17-mer in six passes on a 3.1 Gb genome;
  //   kf2  order K2 > K (DESIGN.md "long filter"): strings of at least K2 characters are tested here first — a random
  //        19-mer passes with p = 0.011, so nearly every table read and interval extension that remains belongs to a
  //        string that really occurs.  This is what the HBM the table does not need is spent on.
  KFilter kf, kf2;
  // Block minima over the suffix array (derived at load, r03): samin[0] = sa itself, samin[j][b] = min of
  // SA[b * 8^j, min(n, (b+1) * 8^j)), j = 1..nlev-1; every level is padded with 0xFFFFFFFF to a multiple of eight entries
  // (+8), so the eight children of a block are two 16-byte loads.  k_locate_topk walks it to take the smallest positions of a
  // repeat-rich interval without reading the interval (hunter.h:355-357 keeps the first max_locations of the sorted list).
  // Preceding characters (derived at load with the K-mer table, r04): pre5[i] = the five text characters in front of suffix SA[i],
  // three bits each (bits 0-2: T[SA[i]-1], bits 3-5: T[SA[i]-2], ...; 0..3 = A,C,G,T, 7 = anything else or before the text).  A
  // string of K + r characters (r <= 5) whose last K characters have the table interval [lo, hi) occurs exactly at SA[i] - r for
  // the i in [lo, hi) whose entry spells its first r characters: ONE line of this array instead of r dependent Occ lines when the
  // interval is narrow (k_search1s).  nullptr = not built.
  const u16* pre5;
  // The text's shortest run of 'N', capped at 64 (r05; 0 = unknown: no pruning).  k_search: a query with non-A/C/G/T characters whose
  // N's are fewer than this and lie more than d characters from both ends has no neighbourhood string that keeps an N and occurs.
  u32 nrun_min;
  static constexpr u32 MAXLEV = 10;
  const u32* samin[MAXLEV];
  u32 nlev;  // levels present, including level 0; 0 = no hierarchy
  // Suffix array WITH context (derived at load, r06): sax[i] = {SA[i], context word of suffix i}.  The verify stage needs, per
  // located hit, the <= d characters in front of and behind the neighbourhood string (hunter.h:363-378) — one random 64-byte
  // text line per hit, which is what bounded the repeat-rich regime (17 M hits per 100 000 queries: DESIGN "repeats").  The
  // locate job kernels read the suffix-array entries of their survivors anyway; an 8-byte entry brings the context along with
  // the position, in the line that is being fetched.  Context word (sax_ctx_*): bits 0-3 the two characters in front of the
  // suffix (T[p-1], T[p-2]; 2 bits each), bits 4-29 the thirteen characters T[p+16 .. p+28] (2 bits each, nearest first): the
  // characters behind a string of 16..27 characters that starts at p, bit 31 = some character of either window is not
  // A/C/G/T or lies outside the text (the hit then reads the text as before).  nullptr = not built.
  const uint2* sax;
  // Prefix levels (derived at load with sax, r06): for X = 2^16, 2^18, ... (below n) the suffixes whose text position lies below X,
  // in suffix-array order — plv[l].rec[r] = sax[i] for the r-th such i — and a rank directory over the suffix-array indices
  // (plv[l].dir: one 64-byte line per 448 indices = {count before the line, seven 64-bit words of flags}).  hunter.h:355-357 keeps the
  // `take` SMALLEST positions of a string's occurrences: for a repeat family with 100 000 copies spread over the genome those all
  // lie below a small X, and the entries of [lo, hi) with position < X are ONE contiguous run of plv[l].rec (two rank reads) — a few
  // thousand records to select from instead of a walk down the block minima over the whole interval.  k_locate picks the level by
  // the interval's density and checks that the run holds at least `take` records (then the answer is inside it); otherwise the
  // string takes the walk as before.  nplv = 0: not built.
  struct PrefixLevel {
    const u64* dir;
    const uint2* rec;
    u64 x;
  };
  static constexpr u32 MAXPLV = 8, PLV_LINE = 448;
  PrefixLevel plv[MAXPLV];
  u32 nplv;
};
// entries of the suffix array below index i whose text position lies below the level's X
DG_DEV u64 plv_rank(const FmView::PrefixLevel& L, u64 i) {
  const u64 line = i / FmView::PLV_LINE;
  const u32 r = (u32)(i - line * FmView::PLV_LINE), w = r >> 6, b = r & 63u;
  const uint4* p = reinterpret_cast<const uint4*>(L.dir + line * 8);
  const uint4 q0 = p[0], q1 = p[1], q2 = p[2], q3 = p[3];
  const u64 word[8] = {((u64)q0.y << 32) | q0.x, ((u64)q0.w << 32) | q0.z, ((u64)q1.y << 32) | q1.x, ((u64)q1.w << 32) | q1.z,
                       ((u64)q2.y << 32) | q2.x, ((u64)q2.w << 32) | q2.z, ((u64)q3.y << 32) | q3.x, ((u64)q3.w << 32) | q3.z};
  u64 c = word[0];
#pragma unroll
  for (u32 j = 0; j < 7; ++j) {
    const u64 m = j < w ? ~0ULL : (j == w ? ((1ULL << b) - 1ULL) : 0ULL);
    c += (u64)__popcll(word[1 + j] & m);
  }
  return c;
}
static constexpr u32 SAX_POST_OFF = 16, SAX_POST_N = 13, SAX_ESCAPE = 0x80000000u;
// HitSeed::len of a located hit that carries its context (set by the locate job kernels, read by k_verify_memo only): bits 0-19 the
// string's length, bit 31 "context valid", bits 20-21 / 22-23 the codes of T[pos-1] / T[pos-2], bits 24-25 / 26-27 of T[pos+len] /
// T[pos+len+1] (0..3 = A,C,G,T).  Valid only for strings whose two following characters lie inside the context word's window.
static constexpr u32 SEED_LEN_MASK = 0xFFFFFu, SEED_CTX_VALID = 0x80000000u;
static constexpr u32 SEED_KEY_VALID = 0x40000000u;  // r06, set by k_locate: the hit's string lies in FlatSel::key at the seed's slot, its context holds ONE character either side
__host__ __device__ inline u32 seed_len_with_ctx(u32 len, u32 ctx) {
  if ((ctx & SAX_ESCAPE) || len < SAX_POST_OFF || len + 2 > SAX_POST_OFF + SAX_POST_N) return len;
  const u32 post = (ctx >> (4 + 2 * (len - SAX_POST_OFF))) & 15u;
  return len | ((ctx & 15u) << 20) | (post << 24) | SEED_CTX_VALID;
}

struct KtabEntry {
  u32 lo, hi;
  u32 pre_first;  // pre5[lo] when the entry carries it (narrow), else 0xFFFFFFFF
};
DG_DEV KtabEntry ktab_decode(uint2 e) {
  KtabEntry k;
  k.lo = e.x;
  const bool wide = (e.y >> 31) != 0;
  k.hi = e.x + (wide ? (e.y & 0x7FFFFFFFu) : (e.y & 0xFFFFu));
  k.pre_first = wide ? 0xFFFFFFFFu : ((e.y >> 16) & 0x7FFFu);
  return k;
}
DG_DEV KtabEntry ktab_entry(const FmView& f, u64 code) { return ktab_decode(f.ktab[code]); }

// is the k-mer `code` present?  t = window position (0 = right-most character) of the edit the neighbouring lanes vary
// (r05) The lane's copy by SELECTS over scalars.  kf.cp[] / kf.s[] live in the kernel-argument segment; indexed by a lane-varying r the
// compiler turned the choice into vector loads from that segment — four extra memory instructions per lane and a dependent
// round trip in front of every probe phase (ISA of k_search1s, r04).  readfirstlane pins the eight values in scalar registers.
DG_DEV void kf_pick(const KFilter& kf, u32 r, const u32*& base, u32& s) {
  // (the copy's address as copy 0's plus a selected distance: pointer arithmetic on a kernel-argument pointer keeps the loads in the
  // global address space; a pointer rebuilt from two integers becomes a flat one)
  u32 lo[4], hi[4], sh[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const long long d = reinterpret_cast<const char*>(kf.cp[k]) - reinterpret_cast<const char*>(kf.cp[0]);
    lo[k] = (u32)__builtin_amdgcn_readfirstlane((int)(u32)(u64)d);
    hi[k] = (u32)__builtin_amdgcn_readfirstlane((int)(u32)((u64)d >> 32));
    sh[k] = (u32)__builtin_amdgcn_readfirstlane((int)kf.s[k]);
  }
  const u32 l = r == 0 ? lo[0] : r == 1 ? lo[1] : r == 2 ? lo[2] : lo[3];
  const u32 h = r == 0 ? hi[0] : r == 1 ? hi[1] : r == 2 ? hi[2] : hi[3];
  s = r == 0 ? sh[0] : r == 1 ? sh[1] : r == 2 ? sh[2] : sh[3];
  base = reinterpret_cast<const u32*>(reinterpret_cast<const char*>(kf.cp[0]) + (long long)(((u64)h << 32) | l));
}
DG_DEV bool kf_present(const KFilter& kf, u64 code, u32 t) {
  const u32 r = (u32)(kf.pick >> (2 * (t < 31 ? t : 31))) & 3u;
  const u32* base;
  u32 s;
  kf_pick(kf, r, base, s);
  const u32 inl = (u32)(code >> s) & 511u;
  const u64 line = (code & ((1ULL << s) - 1)) | ((code >> (s + 9)) << s);
  return (base[line * 16 + (inl >> 5)] >> (inl & 31)) & 1u;
}

// the same test in two steps for lanes that ask about several codes with the same edit position: pick the copy once
struct KfCopy {
  const u32* base;
  u32 s;
};
DG_DEV KfCopy kf_copy(const KFilter& kf, u32 t) {
  const u32 r = (u32)(kf.pick >> (2 * (t < 31 ? t : 31))) & 3u;
  KfCopy c;
  kf_pick(kf, r, c.base, c.s);
  return c;
}
// address of the word that holds a code's bit (and the bit's number): lanes that ask about several codes compute all the
// addresses first and load them back to back — a test per code with its own branch serialises the loads (r02: every probe of
// the flat kernels waited for the previous one, `s_waitcnt vmcnt(0)` after each)
DG_DEV const u32* kf_word(const KfCopy& c, u64 code, u32& bit) {
  const u32 inl = (u32)(code >> c.s) & 511u;
  const u64 line = (code & ((1ULL << c.s) - 1)) | ((code >> (c.s + 9)) << c.s);
  bit = inl & 31u;
  return c.base + line * 16 + (inl >> 5);
}
DG_DEV bool kf_test(const KfCopy& c, u64 code) {
  const u32 inl = (u32)(code >> c.s) & 511u;
  const u64 line = (code & ((1ULL << c.s) - 1)) | ((code >> (c.s + 9)) << c.s);
  return (c.base[line * 16 + (inl >> 5)] >> (inl & 31)) & 1u;
}

DG_DEV u64 packed_get(const u64* w, u32 width, u64 i) {
  u64 b = i * width, q = b >> 6, o = b & 63;
  u64 v = w[q] >> o;
  if (o + width > 64) v |= w[q + 1] << (64 - o);
  return width == 64 ? v : (v & ((1ULL << width) - 1));
}

// ---- sdsl rank_support_v<1,1>::rank ----
DG_DEV u64 sdsl_rank1(const FmView& f, u64 idx) {
  const u64* p = f.rk + ((idx >> 8) & ~1ULL);
  u64 r = p[0] + ((p[1] >> (63 - 9 * ((idx & 0x1FF) >> 6))) & 0x1FF);
  if (idx & 63) r += (u64)__popcll(f.bv[idx >> 6] & ((1ULL << (idx & 63)) - 1));
  return r;
}
// ---- wt_pc::rank(i, c) ----
DG_DEV u64 wt_rank(const FmView& f, u64 i, u32 c) {
  const WtTables* t = f.wt;
  if (t->c_to_leaf[c] == 0xFFFF) return 0;
  u64 p = t->path[c];
  u32 len = (u32)(p >> 56), v = 0;
  u64 res = i;
  for (u32 l = 0; l < len && res; ++l, p >>= 1) {
    u64 ones = sdsl_rank1(f, t->node_pos[v] + res) - t->node_rank[v];
    res = (p & 1) ? ones : res - ones;
    v = t->child[v][p & 1];
  }
  return res;
}
// ---- wt_pc::inverse_select(i) -> rank of BWT[i] among equal symbols before i; symbol returned through sym ----
DG_DEV u64 wt_inverse_select(const FmView& f, u64 i, u32& sym) {
  const WtTables* t = f.wt;
  u32 v = 0;
  while (t->child[v][0] != 0xFFFF) {
    u64 pos = t->node_pos[v] + i;
    u64 ones = sdsl_rank1(f, pos) - t->node_rank[v];
    bool b = (f.bv[pos >> 6] >> (pos & 63)) & 1;
    i = b ? ones : i - ones;
    v = t->child[v][b];
  }
  sym = (u32)t->node_rank[v];
  return i;
}

// ---- Occ blocks ----
struct OccLine {  // one block held in registers
  u32 cnt[4];
  u64 p0[2], p1[2], p2[2];
};
DG_DEV OccLine occ_load(const OccBlock* occ, u64 blk) {
  const uint4* q = reinterpret_cast<const uint4*>(occ + blk);
  uint4 a = q[0], b = q[1], c = q[2], d = q[3];
  OccLine L;
  L.cnt[0] = a.x; L.cnt[1] = a.y; L.cnt[2] = a.z; L.cnt[3] = a.w;
  L.p0[0] = ((u64)b.y << 32) | b.x; L.p0[1] = ((u64)b.w << 32) | b.z;
  L.p1[0] = ((u64)c.y << 32) | c.x; L.p1[1] = ((u64)c.w << 32) | c.z;
  L.p2[0] = ((u64)d.y << 32) | d.x; L.p2[1] = ((u64)d.w << 32) | d.z;
  return L;
}
// dynamic indexing of small register arrays would send them to scratch: select instead
DG_DEV u32 sel4(u32 c, u32 a0, u32 a1, u32 a2, u32 a3) { return c == 0 ? a0 : c == 1 ? a1 : c == 2 ? a2 : a3; }
// occurrences of code c (0..3) in the first r (0..127) symbols of the block, plus the block's running count
DG_DEV u32 occ_in_line(const OccLine& L, u32 r, u32 c) {
  u64 m0 = r >= 64 ? ~0ULL : ((1ULL << r) - 1);
  u64 m1 = r > 64 ? ((1ULL << (r - 64)) - 1) : 0ULL;
  u64 x0 = (c & 1) ? L.p0[0] : ~L.p0[0], x1 = (c & 1) ? L.p0[1] : ~L.p0[1];
  u64 y0 = (c & 2) ? L.p1[0] : ~L.p1[0], y1 = (c & 2) ? L.p1[1] : ~L.p1[1];
  return sel4(c, L.cnt[0], L.cnt[1], L.cnt[2], L.cnt[3]) + (u32)__popcll(x0 & y0 & ~L.p2[0] & m0) + (u32)__popcll(x1 & y1 & ~L.p2[1] & m1);
}
DG_DEV u32 occ_rank(const FmView& f, u64 i, u32 c) {  // #code c in BWT[0,i), i <= n
  OccLine L = occ_load(f.occ, i >> 7);
  return occ_in_line(L, (u32)(i & 127), c);
}
DG_DEV u32 code_in_line(const OccLine& L, u32 r) {
  u32 j = r & 63;
  u64 a = r < 64 ? L.p0[0] : L.p0[1], b = r < 64 ? L.p1[0] : L.p1[1], c = r < 64 ? L.p2[0] : L.p2[1];
  return (u32)((a >> j) & 1) | ((u32)((b >> j) & 1) << 1) | ((u32)((c >> j) & 1) << 2);
}

// LF(i) and BWT[i] through the Occ blocks; symbols outside A,C,G,T fall back to the wavelet tree.
DG_DEV u64 lf_step(const FmView& f, u64 i, u32& sym) {
  OccLine L = occ_load(f.occ, i >> 7);
  u32 r = (u32)(i & 127);
  u32 code = code_in_line(L, r);
  if (code < 4) {
    sym = code == 0 ? 'A' : code == 1 ? 'C' : code == 2 ? 'G' : 'T';
    return (u64)sel4(code, f.C4[0], f.C4[1], f.C4[2], f.C4[3]) + occ_in_line(L, r, code);
  }
  if (code != CODE_OTHER) {
    sym = code == CODE_N ? 'N' : code == CODE_NL ? '\n' : 0u;
    return f.wt->C[f.wt->char2comp[sym]] + wt_rank(f, i, sym);
  }
  u64 rk = wt_inverse_select(f, i, sym);
  return f.wt->C[f.wt->char2comp[sym]] + rk;
}

// One backward-search step on the half-open SA interval [lo,hi): prepend byte `sym`.
// (sdsl backward_search, suffix_array_algorithm.hpp; call sites hunter.h:353, silica.h:470)
DG_DEV void bs_extend_code(const FmView& f, u32& lo, u32& hi, u32 code) {  // code in 0..3
  OccLine A = occ_load(f.occ, lo >> 7);
  OccLine B = occ_load(f.occ, hi >> 7);
  const u32 c4 = sel4(code, f.C4[0], f.C4[1], f.C4[2], f.C4[3]);
  lo = c4 + occ_in_line(A, lo & 127, code);
  hi = c4 + occ_in_line(B, hi & 127, code);
}
// the same step when both interval ends usually fall into one Occ block (narrow intervals, i.e. after the K-mer table):
// one line read and one pair of plane masks instead of two
DG_DEV void bs_extend_code_narrow(const FmView& f, u32& lo, u32& hi, u32 code) {
  const u32 c4 = sel4(code, f.C4[0], f.C4[1], f.C4[2], f.C4[3]);
  const OccLine A = occ_load(f.occ, lo >> 7);
  if ((lo >> 7) == (hi >> 7)) {
    const u32 a = lo & 127, b = hi & 127;  // a <= b
    const u64 x0 = (code & 1) ? A.p0[0] : ~A.p0[0], x1 = (code & 1) ? A.p0[1] : ~A.p0[1];
    const u64 y0 = (code & 2) ? A.p1[0] : ~A.p1[0], y1 = (code & 2) ? A.p1[1] : ~A.p1[1];
    const u64 w0 = x0 & y0 & ~A.p2[0], w1 = x1 & y1 & ~A.p2[1];  // positions of the block that hold `code`
    const u64 ma0 = a >= 64 ? ~0ULL : ((1ULL << a) - 1), ma1 = a > 64 ? ((1ULL << (a - 64)) - 1) : 0ULL;
    const u64 mb0 = b >= 64 ? ~0ULL : ((1ULL << b) - 1), mb1 = b > 64 ? ((1ULL << (b - 64)) - 1) : 0ULL;
    const u32 below = (u32)__popcll(w0 & ma0) + (u32)__popcll(w1 & ma1);
    const u32 inside = (u32)__popcll(w0 & mb0 & ~ma0) + (u32)__popcll(w1 & mb1 & ~ma1);
    lo = c4 + sel4(code, A.cnt[0], A.cnt[1], A.cnt[2], A.cnt[3]) + below;
    hi = lo + inside;
    return;
  }
  const OccLine B = occ_load(f.occ, hi >> 7);
  lo = c4 + occ_in_line(A, lo & 127, code);
  hi = c4 + occ_in_line(B, hi & 127, code);
}
DG_DEV void bs_extend_sym(const FmView& f, u32& lo, u32& hi, u32 sym, u32 code) {
  if (code < 4) {
    bs_extend_code(f, lo, hi, code);
    return;
  }
  const WtTables* t = f.wt;
  u32 cc = t->char2comp[sym];
  if (cc == 0 && sym > 0) {  // byte not in the alphabet
    lo = hi = 0;
    return;
  }
  u64 cb = t->C[cc];
  lo = (u32)(cb + wt_rank(f, lo, sym));
  hi = (u32)(cb + wt_rank(f, hi, sym));
}

// Add per-lane counters with one atomic per wavefront.  (tools/hostemu pre-defines DG_HAVE_WAVE_ADD with a per-thread
// version because it runs lanes one at a time.)
#ifndef DG_HAVE_WAVE_ADD
DG_DEV void wave_add(unsigned long long* p, u64 v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor((unsigned long long)v, o);
  if ((threadIdx.x & 63) == 0 && v) atomicAdd(p, (unsigned long long)v);
}
#endif

// Sum of a 32-bit value over the wavefront's 64 lanes, the same in every lane (a scalar after the read-lanes).  DPP inside the
// rows of 16, four read-lanes across them: no permute addresses to hold.  (r04's __shfl_xor loops kept six ds_bpermute address
// registers alive across k_search1s and spilled two of them to scratch in EVERY wavefront: 0.87 M write requests and 48 MB of
// WRITE_SIZE per launch for nothing — VERDICT r04 weak #4.)  Call with the whole wavefront active.
DG_DEV u32 wave_sum32(u32 v) {
  v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true);   // quad_perm [1,0,3,2]
  v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true);   // quad_perm [2,3,0,1]
  v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, true);  // row_half_mirror
  v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xF, 0xF, true);  // row_mirror
  return (u32)__builtin_amdgcn_readlane((int)v, 0) + (u32)__builtin_amdgcn_readlane((int)v, 16) + (u32)__builtin_amdgcn_readlane((int)v, 32) +
         (u32)__builtin_amdgcn_readlane((int)v, 48);
}

// Inclusive prefix sum over the wavefront's 64 lanes, data-parallel primitives again: four shifts inside the rows of 16, then the
// rows' totals handed on (row_bcast:15 into rows 1 and 3, row_bcast:31 into rows 2 and 3).  The __shfl_up loop it replaces in
// k_search2p kept seven permute addresses in registers across the whole kernel — which the compiler sent to scratch (r05).  Call with
// the whole wavefront active.
DG_DEV u32 wave_incl_scan32(u32 v) {
  v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, true);   // row_shr:1 (lanes without a source add 0)
  v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, true);   // row_shr:2
  v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, true);   // row_shr:4
  v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, true);   // row_shr:8
  v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false);  // row_bcast:15, rows 1 and 3
  v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false);  // row_bcast:31, rows 2 and 3
  return v;
}

DG_DEV u32 code_of_byte(u32 b) {
  return b == 'A' ? 0u : b == 'C' ? 1u : b == 'G' ? 2u : b == 'T' ? 3u : b == 'N' ? 4u : b == '\n' ? 5u : b == 0 ? 6u : 7u;
}

}  // namespace dg
