// Search stage of the hunt pipeline (included by hunt.hip only): query preparation (hunter.h:299-315), the flat distance-1 kernels
// (k_search1p, k_search1s with the select stage inside), the flat edit-distance-2 kernel (k_search2p), the general walker (k_search)
// and the explicit-pattern search of capped neighbourhoods (k_explicit).  neighbors.h:29-92 + hunter.h:353.
#pragma once
#include <type_traits>

#include "hunt_internal.hpp"
#include "iupac.hpp"

namespace dg {

struct Leaf {
  u32 qs;    // 2*query + strand
  u32 slot;  // running number within its (query,strand) group
  u32 lo, hi;
  u32 nops;
  u32 ops[DMAX];  // pos<<4 | kind<<2 | code, in right-to-left order of application
};

struct Sel {  // a kept neighbourhood string, in search order
  u32 lo, hi;
  u32 len;    // string length; r04 filtered form (bit 31 set): bits 0-7 length, [lo, hi) is the interval of the string's LAST len - pre
              // characters, pre = bits 8-10, and bits 11-26 say which of its <= 16 suffixes are preceded by the first pre characters
  u32 take;   // how many of its occurrences become hits
  u32 hbase;  // first hit slot, relative to the query's first hit
  u32 g;      // 2*query + strand (set for the strings of the flat region, where no leaf record names the group)
};
// r06, filtered form with ONE suffix in its mask and <= 4 characters in front of the table window: bit 30 of `len` is set and bits 27-28 hold
// the code of the text character in front of that occurrence; the string itself (2 bits per character, the first character in the highest
// pair) lies in FlatSel::key at the string's slot.  With the character behind the occurrence, which the locate stage finds in the suffix's
// record (FmView::sax), the verify stage aligns the hit without reading the text.
static constexpr u32 SEL_CTX_SHIFT = 27, SEL_CTX_VALID = 1u << 30;

DG_HD u32 sel_len_filtered(u32 len, u32 pre, u32 mask) { return (len & 255u) | ((pre & 7u) << 8) | ((mask & 0xFFFFu) << 11) | (1u << 31); }
// from a leaf's filter word (mask | characters in front << 16 | 1 << 31, 0 = ordinary interval)
DG_HD u32 sel_len_from(u32 len, u32 fword) { return (fword >> 31) ? sel_len_filtered(len, (fword >> 16) & 7u, fword & 0xFFFFu) : len; }
DG_HD bool sel_filtered(const Sel& s) { return (s.len >> 31) != 0; }
DG_HD u32 sel_strlen(const Sel& s) { return sel_filtered(s) ? (s.len & 255u) : s.len; }  // (an ordinary string may have 30 000 characters)
DG_HD u32 sel_pre(const Sel& s) { return (s.len >> 8) & 7u; }
DG_HD u32 sel_mask(const Sel& s) { return (s.len >> 11) & 0xFFFFu; }
DG_HD u64 sel_occ(const Sel& s) {  // occurrences of the string (sdsl::count)
  if (!sel_filtered(s)) return (u64)s.hi - s.lo;
  u32 m = sel_mask(s), c = 0;
  while (m) {
    m &= m - 1;
    ++c;
  }
  return c;
}

// Largest number of distinct strings neighbors() can hold for a query of length m with nN letters outside A/C/G/T (they
// are 'N' after replaceNonDna and can be substituted by all four bases instead of three); used to prove that the
// maxNeighborhood early return (neighbors.h:50) cannot fire.  Returns ~0 when no such proof is available.
DG_HD u64 neighbourhood_bound(u32 m, u32 d, bool indel, u32 nN = 0) {
  auto binom = [](u64 n, u64 k) {
    u64 r = 1;
    for (u64 i = 1; i <= k; ++i) r = r * (n - k + i) / i;
    return r;
  };
  if (nN > m) nN = m;
  if (!indel) {  // exactly: i substituted positions, j of them at an N (4 letters) and i-j elsewhere (3 letters)
    u64 t = 0;
    for (u32 i = 0; i <= d && i <= m; ++i)
      for (u32 j = 0; j <= i && j <= nN; ++j) {
        if (i - j > m - nN) continue;
        u64 term = binom(nN, j) * binom(m - nN, i - j);
        for (u32 k = 0; k < j; ++k) term *= 4;
        for (u32 k = 0; k < i - j; ++k) term *= 3;
        t += term;
        if (t > (1ULL << 40)) return ~0ULL;
      }
    return t;
  }
  // Edit mode.  With N positions: strings that need the fourth letter at an N position spend one edit on that substitution
  // and reach at most G(d-1) strings with the rest, G(0) = 1, G(1) = 1 + m deletions + 4m substitutions + 4(m+1)
  // insertions; everything else obeys the three-letter count below.
  u64 extra = 0;
  if (nN) {
    if (d == 1) extra = nN;
    else if (d == 2) extra = (u64)nN * (9ULL * m + 5);
    else if (d > 2) return ~0ULL;
  }
  if (d == 0) return 1;
  if (d == 1) return 7ULL * m + 5 + extra;  // 1 + 3m substitutions + m deletions + (3m+4) insertions
  if (d == 2) {
    // Distinct strings within two edits, by length class (DESIGN.md "neighbourhood size bound"); M = m-1 is the query
    // without its last character, which every string of L must still align to (no insertion after the last column).
    const u64 M = m - 1;
    u64 len_m2 = binom(m, 2);                                              // two deletions
    u64 len_m1 = m + 3ULL * m * (m - 1);                                   // D, D+S
    u64 len_0 = 1 + 3ULL * m + 9 * binom(m, 2) + m * (3ULL * (m - 1) + 4) - (3ULL * m + 1);  // q, S, SS, D+I (q and S counted once)
    u64 len_p1 = (3 * M + 4) * (1 + 3 * M) - 6 * M - 3 * M + 3 * (3 * M + 4);  // I, I+S; last column M or S
    u64 len_p2 = 1 + 3 * (m + 1) + 9 * binom(m + 1, 2);                    // supersequences of q[0..m-1) of length m+1, then q[m-1]
    return len_m2 + len_m1 + len_0 + len_p1 + len_p2 + extra;
  }
  return ~0ULL;
}

// ------------------------------------------------------------------------------------------------------------
// (r03: the characters come in as aligned 64-bit words loaded together — the byte loop waited for one load per character, 20 us
// for 100 000 20-mers — and the lane clears its query's group counters, which takes the place of a memset in front of the batch)
// One query: hunter.h:299-315 + util.h:54-114,208-219.  write_bytes: the per-character arrays (fw / rv codes, normalised ASCII) are
// only read by the generic kernels, the full-matrix verify and the classic result fetch; the flat distance-1 path with the
// banded verify and compact results works from the packed records (GidInfo, position masks) alone.  grp_cnt may be null (the
// generic path's group counters).  gi_out: the two strands' records, also stored to b.ginfo.
struct PreparedQuery {
  u32 flags, d, bad;
};
DG_DEV PreparedQuery prepare_query(const Batch& b, u64 q, u32* grp_cnt, u32* nsel, u32* selbase, u32* n_generic, bool write_bytes, GidInfo* gi_out) {
  if (grp_cnt) grp_cnt[2 * q] = grp_cnt[2 * q + 1] = 0;
  nsel[2 * q] = nsel[2 * q + 1] = 0;
  selbase[2 * q] = selbase[2 * q + 1] = 0xFFFFFFFFu;  // "generic path" until k_search1s claims the group
  u64 s = b.qoff[q], e = b.qoff[q + 1];
  // Offsets are checked BEFORE a byte is touched: with dg_hunt_params::max_query_len the host skips its own pass over
  // them, and a decreasing pair would wrap m to ~4e9 and send the byte loop far outside the workspaces.  Such a query counts as
  // "too long" (the batch fails with DG_EINVAL) and is prepared as an empty one.
  const bool bad_len = e < s || e - s > (u64)b.maxlen_bound || e > b.total_qbytes || (q + 1 == b.nq && e != b.total_qbytes);
  if (bad_len) {
    atomicAdd(b.too_long, 1u);
    e = s = 0;
  }
  u32 m = (u32)(e - s), bad = 0, flags = 0, generic = 0;
  u64 pk_fw = 0, pk_rv = 0;  // 2-bit packed strands, q[i] at bits 2(m-1-i) (meaningful for m <= 32 without N)
  u32 pm[4] = {0u, 0u, 0u, 0u};  // position masks of the forward strand: bit i of pm[x] <=> q[i] is base x (m <= 32; an N sets none)
  constexpr u32 NREG = 40;   // queries up to this length travel through registers
  if (m <= NREG) {
    constexpr int NW = NREG / 8 + 1;
    const u64 a0 = s & ~7ULL;
    const u32 sh = (u32)(s & 7) * 8;
    const u64* src = reinterpret_cast<const u64*>(b.qbytes + a0);
    u64 w[NW + 1], x[NW];
#pragma unroll
    for (int i = 0; i <= NW; ++i) w[i] = (u32)(8 * i) < (u32)(s & 7) + m ? src[i] : 0ULL;
#pragma unroll
    for (int i = 0; i < NW; ++i) x[i] = sh ? (w[i] >> sh) | (w[i + 1] << (64 - sh)) : w[i];
#pragma unroll
    for (u32 i = 0; i < NREG; ++i) {
      if (i < m) {
        u32 ch = (u32)(x[i >> 3] >> (8 * (i & 7))) & 255u;
        if (ch >= 'a' && ch <= 'z') ch -= 32;  // boost::to_upper_copy, hunter.h:306
        const u32 code = ch == 'A' ? 0u : ch == 'C' ? 1u : ch == 'G' ? 2u : ch == 'T' ? 3u : 4u;
        bad += (code == 4);  // every replaced character raises one warning (util.h:214); a literal 'N' is replaced too
        if (write_bytes) {
          b.fw[s + i] = (u8)code;
          b.qseq[s + i] = ascii_of(code);
          b.rv[s + (m - 1 - i)] = (u8)(code < 4 ? 3 - code : 4);  // util.h:54-91,110-114
        }
        pk_fw = (pk_fw << 2) | (code & 3u);
        pk_rv |= (u64)((3u - code) & 3u) << (2 * (i & 31u));
        if (i < 32) {
#pragma unroll
          for (u32 x = 0; x < 4; ++x) pm[x] |= (u32)(code == x) << i;
        }
      }
    }
  } else {
    for (u32 i = 0; i < m; ++i) {
      u32 ch = b.qbytes[s + i];
      if (ch >= 'a' && ch <= 'z') ch -= 32;
      u32 code = ch == 'A' ? 0u : ch == 'C' ? 1u : ch == 'G' ? 2u : ch == 'T' ? 3u : 4u;
      bad += (code == 4);
      b.fw[s + i] = (u8)code;
      b.qseq[s + i] = ascii_of(code);
      b.rv[s + (m - 1 - i)] = (u8)(code < 4 ? 3 - code : 4);
    }
  }
  u32 d = b.distance;
  if (m < 10) flags |= DG_Q_TOO_SHORT;  // hunter.h:299
  else if (d >= m) {                    // hunter.h:312-315
    d = m - 1;
    flags |= DG_Q_DIST_ADJUSTED;
  }
  // If the cap could fire for a query, the reference's answer depends on its generation order, which k_search does not
  // reproduce: the host has enumerated such queries beforehand (qmode).  A query that could reach the cap without the host
  // having looked at it is a bookkeeping error of this library and stops the batch.
  const u32 mode = b.qmode ? b.qmode[q] : (u32)QM_KERNEL;
  const bool explicit_set = (mode & 15u) == QM_EXPLICIT;
  if ((mode & 15u) == QM_KERNEL && m >= 10 && neighbourhood_bound(m, d, b.indel != 0, bad) >= b.max_neighborhood) atomicAdd(b.refused, 1u);
  if ((mode & QM_FIRED) && m >= 10) flags |= DG_Q_NBHD_EXCEEDED;  // hunter.h:342-345
  b.qlen[q] = m;
  b.qdist[q] = d;
  b.qflags[q] = flags;
  b.qnondna[q] = bad;
  for (u32 strand = 0; strand < 2; ++strand) {
    GidInfo gi;
    gi.qpk = 0;
    gi.m = ((flags & DG_Q_TOO_SHORT) || explicit_set || (strand && !b.reverse) || bad_len) ? 0u : m;
    gi.d_win = d | (bad == 0 ? 256u : 0u);
    if (b.fastK && gi.m && bad == 0 && d == 1 && m <= 31 && m >= b.fastK + 1) gi.d_win |= 512u;
    if (b.fast2K && gi.m && bad == 0 && d == 2 && m <= 30 && m >= b.fast2K + 2) {
      if (m >= b.fast2_minlen) gi.d_win |= 1024u;
      else atomicAdd(b.short2, 1u);  // k_search2p<., true> leaves it to the walker; the handle switches to the r04 body for the next batches
    }
    if (bad == 0 && m <= 32) gi.qpk = strand ? pk_rv : pk_fw;
    // r05: a query with N's that all lie left of every table window of its strand's strings (string index below m - K - d) walks in
    // window mode like an N-free one — the packed copy (an N as 0) is only read inside windows — and so gets k_search's root split:
    // one lane per first edit instead of one lane per strand with ~1 500 dependent index reads
    // r06: as many N's as the query has edits, none within d characters of either end, fewer than the text's shortest run of N:
    // no neighbourhood string that keeps an N occurs (k_search's census argument), so every N is substituted or deleted and that is
    // the whole budget — 5^d strings per strand (4^d in Hamming mode), searched by k_nres one lane per string; the walker skips
    // the strand.  (r05: the strand whose N sits inside the window zone walked 75-300 dependent reads in one lane.)
    // One N, one edit, the N within a character of an end (bit 13 on top): the strings that substitute or delete it are k_nres's all
    // the same; the strings that KEEP it occur only with the N as their first or last character (a text whose shortest run of N has
    // two or more holds no single N between two bases) — k_nkeep searches those, a lane per edit of the rest.
    bool nres = false;
    if (bad != 0 && bad == d && d <= 2 && m <= 32 && gi.m && b.nrun_min && bad < b.nrun_min && (mode & 15u) == QM_KERNEL) {
      const u32 nm = ~(pm[0] | pm[1] | pm[2] | pm[3]) & (m == 32 ? ~0u : ((1u << m) - 1u));  // bit i: forward character i is an N
      const u32 first = (u32)__builtin_ctz(nm), last = 31u - (u32)__builtin_clz(nm);
      nres = first >= d + 1 && (m - 1 - last) >= d + 1;  // (the same for the reverse strand: its N's are the mirror image)
      if (nres) gi.d_win |= 4096u;
      else if (d == 1 && b.nrun_min >= 2) gi.d_win |= 4096u | 8192u;
      if (gi.d_win & 4096u) gi.qpk = strand ? pk_rv : pk_fw;  // (an N reads as some base: k_nres overwrites those positions)
    }
    if (!(gi.d_win & 4096u) && bad != 0 && m <= 32 && b.tabK && m >= b.tabK + d && gi.m) {  // (a strand of k_nres / k_nkeep does not walk)
      const u32 nm = ~(pm[0] | pm[1] | pm[2] | pm[3]) & (m == 32 ? ~0u : ((1u << m) - 1u));  // bit i: forward character i is an N
      const u32 lim = m - b.tabK - d;  // N's are allowed at string indices below this
      const bool ok = !strand ? (lim < 32 && (nm >> lim) == 0u) : (nm & (m - lim >= 32 ? ~0u : ((1u << (m - lim)) - 1u))) == 0u;
      if (ok) {
        gi.d_win |= 2048u;
        gi.qpk = strand ? pk_rv : pk_fw;
        *b.nwin = 1u;  // (a flag: the handle gives the walker its root split while batches hold such strands)
      }
    }
    b.ginfo[2 * q + strand] = gi;
    if (gi_out) gi_out[strand] = gi;
    // the banded verify takes the query as position masks (band_align_bits); the reverse strand's character j is the complement
    // of the forward strand's character m - 1 - j
    if (m <= 32 && m >= 1) {
      uint4 pq;
      if (!strand) pq = make_uint4(pm[0], pm[1], pm[2], pm[3]);
      else pq = make_uint4(__brev(pm[3]) >> (32 - m), __brev(pm[2]) >> (32 - m), __brev(pm[1]) >> (32 - m), __brev(pm[0]) >> (32 - m));
      b.gpeq[2 * q + strand] = pq;
    }
    generic += (gi.m != 0 && !(gi.d_win & (512u | 1024u)));
  }
  // groups the flat distance-1 kernel does not take: the host launches the generic kernels for them (and repeats a batch it
  // started without, run_batch).  A flag, not a count: every lane that has one stores the same 1.
  if (generic && (b.fastK || b.fast2K)) *n_generic = 1u;
  return PreparedQuery{flags, d, bad};
}
__global__ void k_prepare(Batch b, u32* grp_cnt, u32* nsel, u32* selbase, u32* n_generic, u32 write_bytes) {
  const u64 q = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= b.nq) return;
  (void)prepare_query(b, q, grp_cnt, nsel, selbase, n_generic, write_bytes != 0, nullptr);
}

// ------------------------------------------------------------------------------------------------------------
// Search.  State machine: every loop iteration performs at most one index access (an interval extension = two
// Occ-block reads, or one K-mer table read), whatever trie level the lane is on, so a wavefront stays converged on the
// memory operation.  Frames live in registers (fully unrolled selects over the <= D+1 levels, no scratch).
//
// K-mer table ("window mode"): while fewer than K characters have been emitted the lane only accumulates their 2-bit
// codes; the K-th character turns the code into an SA interval with ONE table read, replacing K extensions.  Once a
// branch has spent its whole budget the rest of the window is copied from the query in O(1).  Lanes whose strings may
// be shorter than K, or whose query holds an N, run the same loop in interval mode from the start.
struct Frame {
  u32 pos;  // query characters still to consume (q[0..pos))
  u32 lo;   // interval mode: SA interval [lo,hi);  window mode: (hi:lo) = accumulated 2-bit codes (up to 34 bits)
  u32 hi;
  u32 st;   // bits 0-3 next edit operation, bits 4-8 emitted count (window mode), bit 9 window mode
};
enum : u32 { ST_WIN = 1u << 9 };

template <int D>
struct FrameStack {
  Frame fr[D + 1];
  DG_DEV Frame get(u32 L) const {
    Frame f = fr[0];
#pragma unroll
    for (int k = 1; k <= D; ++k)
      if (L == (u32)k) f = fr[k];
    return f;
  }
  DG_DEV void set(u32 L, const Frame& f) {
#pragma unroll
    for (int k = 0; k <= D; ++k)
      if (L == (u32)k) fr[k] = f;
  }
};
template <int D>
struct OpStack {
  u32 v[D > 0 ? D : 1] = {0};
  DG_DEV void set(u32 L, u32 x) {
#pragma unroll
    for (int k = 0; k < (D > 0 ? D : 1); ++k)
      if (L == (u32)k) v[k] = x;
  }
};

struct SearchOut {
  Leaf* leaves;   // NSHARD regions of shard_cap entries
  u32 shard_cap;
  Counters* ctr;
  u32* grp_cnt;
};

// emit one character (code 0..3) in front of what the frame stands for; returns false when the branch is dead
// K-mer code -> SA interval: the presence filter first (one bit, FmView::kf), the table entry only for K-mers that occur
DG_DEV uint2 kmer_interval(const FmView& f, u64 code, u32 edit_at, u64& lookups, u64& probes) {
  if (f.kf.nr) {
    ++probes;
    if (!kf_present(f.kf, code, edit_at)) return make_uint2(0u, 0u);
  }
  ++lookups;
  const KtabEntry k = ktab_entry(f, code);
  return make_uint2(k.lo, k.hi);
}
DG_DEV bool frame_emit(const FmView& f, Frame& fr, u32 c, u64& steps, u64& lookups, u64& probes) {
  if (fr.st & ST_WIN) {
    u32 e = (fr.st >> 4) & 31;
    u64 code = ((u64)fr.hi << 32 | fr.lo) | ((u64)c << (2 * e));
    fr.lo = (u32)code;
    fr.hi = (u32)(code >> 32);
    ++e;
    if (e == f.K) {
      uint2 iv = kmer_interval(f, code, e - 1, lookups, probes);
      fr.lo = iv.x;
      fr.hi = iv.y;
      fr.st &= ~(ST_WIN | (31u << 4));
      return iv.x < iv.y;
    }
    fr.st = (fr.st & ~(31u << 4)) | (e << 4);
    return true;
  }
  bs_extend_code(f, fr.lo, fr.hi, c);
  ++steps;
  return fr.lo < fr.hi;
}
// window mode with no budget left: the remaining K-e characters are the query's own; one table read
DG_DEV bool frame_finish_window(const FmView& f, Frame& fr, const u8* seq, u32 m, u64 qpk, u64& lookups, u64& probes) {
  const u32 e = (fr.st >> 4) & 31, need = f.K - e;
  u64 code = (u64)fr.hi << 32 | fr.lo;
  if (m <= 32) {  // qpk holds q[i] at bits 2(m-1-i): the next character to emit is at the bottom after the shift
    u64 w = qpk >> (2 * (m - fr.pos));
    u64 mask = need >= 32 ? ~0ULL : ((1ULL << (2 * need)) - 1);
    code |= (w & mask) << (2 * e);
  } else {
    for (u32 t = 0; t < need; ++t) code |= (u64)seq[fr.pos - 1 - t] << (2 * (e + t));
  }
  fr.pos -= need;
  uint2 iv = kmer_interval(f, code, e ? e - 1 : 0u, lookups, probes);  // the last edit sits just right of the copied characters
  fr.lo = iv.x;
  fr.hi = iv.y;
  fr.st &= ~(ST_WIN | (31u << 4));
  return iv.x < iv.y;
}

// Work split: with the table, the root level of the trie is cut into independent items — one lane per
// (query, strand, window offset j of the first edit, operation), plus one "rest" lane that owns the unedited window and
// every first edit to the left of it.  An item lane jumps straight to its node (the j characters right of the edit are
// the query's own), applies its single operation and explores that subtree only.  ~K*NOPS+1 times more lanes, each with
// a handful of dependent index reads instead of hundreds: the kernel becomes throughput- instead of latency-bound.
// ------------------------------------------------------------------------------------------------------------
// Distance 1, the common case, without the state machine.  Profiling the general kernel (SQ counters, r02) showed it is
// bound by instruction issue, not by memory: ~1400 instructions per wavefront at 45 % lane utilisation, because every lane
// sits in a different state of the walker.  With one edit the work is flat, so it is laid out flat:
//   phase A, one lane per (query, strand, position, operation): build the edited string in a 64-bit register (2 bits per
//            character), take its last K characters as the table code and test the presence filter (or the table itself
//            when there is no filter) — a dozen instructions and one memory access; about four lanes in five stop here;
//   phase B: the survivors of the workgroup are packed through LDS into its first lanes, which read the table entry and
//            extend the interval over the characters left of the window (two Occ lines per step).
// Every lane of phase B has the same few steps ahead of it, so wavefronts stay full and short.  Strings and leaves are
// exactly those of k_search<INDEL,1>: deletions, substitutions by another base and insertions between two characters
// (neighbors.h:51-78; a leading insertion is dominated by the string without it, a trailing one is not generated), and in
// Hamming mode the sequence itself.  Queries with an N, longer than 31 nt or shorter than K+1 stay with k_search.
// (r02's first flat form gave every (position, operation) its own lane — ~200 vector instructions per candidate; it was
// removed in r04.  What follows is the form that replaced it.)
// The same search with one lane per (query, strand, POSITION): the lane builds all eight strings of its position from the
// shared pieces (the characters right of the position, the query shifted by none / one character) in a fully unrolled loop —
// the operation is a compile-time constant in every iteration, so nothing diverges — and issues its eight filter probes
// back to back.  The lane-per-operation form above spends ~200 vector instructions per candidate (every lane runs the code
// of all three operation kinds, the (group, item) decode and the record load for one string) and was bound by instruction
// issue once the long filter had removed most of its memory accesses (r02: 0.39 ms whatever the filter / table orders);
// this form needs ~25 per candidate.  Survivors are queued in LDS as (lane, operation) and rebuilt by the dense phase.
template <bool INDEL>
DG_DEV bool cand1(u64 qpk, u32 m, u32 pos, u32 op, u64& s_pk, u32& mlen, u32& opword) {
  const u32 R = m - pos;  // unchanged characters right of the operation
  const u64 low = qpk & ((1ULL << (2 * R)) - 1);
  const u32 old = (u32)(qpk >> (2 * R)) & 3u;
  if (op == 0) {
    if (INDEL) {
      s_pk = low | ((qpk >> (2 * R + 2)) << (2 * R));
      mlen = m - 1;
      opword = ((pos << 4) | (OP_D << 2)) | (1u << 28);
      // deleting either of two equal neighbours gives the same string: the right-most character of a run does it
      return !(R >= 1 && ((u32)(qpk >> (2 * R - 2)) & 3u) == old);
    }
    s_pk = qpk;  // the sequence itself belongs to the Hamming set
    mlen = m;
    opword = 0;
    return pos == 1;
  }
  if (op < 4) {
    const u32 c = (old + op) & 3u;  // neighbors.h:63: a different base
    s_pk = qpk ^ ((u64)(old ^ c) << (2 * R));
    mlen = m;
    opword = ((pos << 4) | (OP_S << 2) | c) | (1u << 28);
    return true;
  }
  const u32 c = op - 4;
  s_pk = low | ((u64)c << (2 * R)) | ((qpk >> (2 * R)) << (2 * R + 2));
  mlen = m + 1;
  opword = ((pos << 4) | (OP_I << 2) | c) | (1u << 28);
  // neighbors.h:51: nothing after the last character; and a base inserted right of an equal one is the string of the
  // insertion one position further left (which exists from the second position on)
  return pos < m && !(pos >= 2 && c == old);
}
// The place of a window code's bit in a copy of the long filter whose in-line field starts at code bit s: (word offset inside the copy,
// bit number).  A permutation of the code's bits, so place(a | b) = place(a) | place(b): k_search2p<., true> ORs the places of a
// string's parts.  (Word offsets fit 32 bits up to K2 = 18.)
struct FiltPos {
  u32 off, bit;
};
DG_DEV FiltPos filt_pos(u64 w, u32 s) {
  const u32 inl = (u32)(w >> s) & 511u;
  FiltPos r;
  r.off = (u32)(((w >> (s + 9)) << (s + 4)) | ((w & ((1ULL << s) - 1)) << 4)) | (inl >> 5);
  r.bit = inl & 31u;
  return r;
}
// The eight probes of a position when every string of the neighbourhood is at least K2 characters long (the usual primer: m - 1 >= K2),
// without building the eight strings (r05).  The strings of one KIND differ from each other only in the edited character, i.e. in two
// bits of the window code: one base code per kind (the substituted / inserted character's bits cleared; one deletion), one (line,
// in-line bit) pair per base code, and every variant by OR-ing the two bits' images — an in-line bit when the lane's copy of the
// filter holds that code bit in its field, a bit of the 27-bit line index otherwise.  ~10 instead of ~35 instructions per candidate:
// the r04 counters put k_search1s at 64 M vector instructions per launch, two thirds of its run time in issue slots.
// Candidates, validity and order are cand1's (which still rebuilds the survivors in the dense phase).
// The core: word offsets inside the lane's copy of the long filter and the bit numbers for the operations 0 .. NOPS-1 applied at `pos`
// to the packed string (pk, len), all of whose results are at least K2 characters long.  Operation 0 is the deletion (INDEL) or "the
// string itself" (!INDEL: the Hamming ball's member without an edit here); 1-3 substitute the other bases; 4-7 insert A, C, G, T.
template <bool INDEL, u32 NOPS>
DG_DEV void probe8_core(const KfCopy& c, u32 K2, u64 pk, u32 len, u32 pos, u32 (&off)[NOPS], u32 (&bit)[NOPS], u32& old_out) {
  const u32 R = len - pos;
  const bool inwin = R < K2;  // else the last K2 characters of every result are the string's own
  const u32 s = c.s;
  const u64 mask2 = (1ULL << (2 * K2)) - 1;
  const u32 old = (u32)(pk >> (2 * R)) & 3u;
  old_out = old;
  const u64 low = pk & ((1ULL << (2 * R)) - 1);
  const u64 wS = pk & mask2 & ~(inwin ? (3ULL << (2 * R)) : 0ULL);
  auto split = [&](u64 w, u32& line, u32& inl) {
    inl = (u32)(w >> s) & 511u;
    line = (u32)((w & ((1ULL << s) - 1)) | ((w >> (s + 9)) << s));
  };
  // images of the edited character's two code bits
  u32 dl0 = 0, dl1 = 0, di0 = 0, di1 = 0;
  if (inwin) {
    const u32 b0 = 2 * R, b1 = b0 + 1;
    di0 = (b0 >= s && b0 < s + 9) ? 1u << (b0 - s) : 0u;
    dl0 = b0 < s ? 1u << b0 : (b0 >= s + 9 ? 1u << (b0 - 9) : 0u);
    di1 = (b1 >= s && b1 < s + 9) ? 1u << (b1 - s) : 0u;
    dl1 = b1 < s ? 1u << b1 : (b1 >= s + 9 ? 1u << (b1 - 9) : 0u);
  }
  auto place = [&](u32 op, u32 line0, u32 inl0, u32 ch) {
    const u32 line = line0 | ((ch & 1u) ? dl0 : 0u) | ((ch & 2u) ? dl1 : 0u);
    const u32 inl = inl0 | ((ch & 1u) ? di0 : 0u) | ((ch & 2u) ? di1 : 0u);
    off[op] = line * 16u + (inl >> 5);  // word offset inside the copy (< 2^31: a 27-bit line index at K2 = 18, the callers' bound)
    bit[op] = inl & 31u;
  };
  u32 lineS, inlS;
  split(wS, lineS, inlS);
  if (INDEL) {
    u32 lineD, inlD, lineI, inlI;
    split((low | ((pk >> (2 * R + 2)) << (2 * R))) & mask2, lineD, inlD);
    split((low | ((pk >> (2 * R)) << (2 * R + 2))) & mask2, lineI, inlI);  // the inserted character's bits are zero
    place(0, lineD, inlD, 0u);
#pragma unroll
    for (u32 op = 4; op < (NOPS > 4 ? 8u : 4u); ++op) place(op < NOPS ? op : 0u, lineI, inlI, op - 4);
  } else {
    place(0, lineS, inlS, inwin ? old : 0u);
  }
#pragma unroll
  for (u32 op = 1; op < 4; ++op) place(op, lineS, inlS, inwin ? ((old + op) & 3u) : 0u);
}
template <bool INDEL, u32 NOPS>
DG_DEV void probe8_long(const FmView& f, u64 qpk, u32 m, u32 pos, KfCopy& c, u32 (&off)[NOPS], u32 (&bit)[NOPS], u32& valid) {
  const u32 K2 = f.kf2.k, R = m - pos;
  c = kf_copy(f.kf2, R < K2 ? R : K2 - 1);
  u32 old;
  probe8_core<INDEL, NOPS>(c, K2, qpk, m, pos, off, bit, old);
  valid = 14u;  // the three substitutions
  if (INDEL) {
    // deleting either of two equal neighbours gives the same string: the right-most character of a run does it (cand1)
    valid |= (u32)!(R >= 1 && ((u32)(qpk >> (2 * R - 2)) & 3u) == old);
#pragma unroll
    for (u32 op = 4; op < 8; ++op) valid |= (u32)(pos < m && !(pos >= 2 && (op - 4) == old)) << op;  // neighbors.h:51, and cand1's duplicate rule
  } else valid |= (u32)(pos == 1);  // the sequence itself belongs to the Hamming set (lane of position 1)
}

// Second look at a survivor that is longer than the long filter's order: its FIRST K2 characters must occur as well.  The two
// windows overlap in all but (length - K2) characters, yet on a 3.1 Gb genome three of four random survivors end here — for one
// line instead of the table entry and 3-5 Occ lines.  R = characters right of the (last) edit, for the choice of the copy.
DG_DEV bool head_window_occurs(const FmView& f, u64 s_pk, u32 mlen, u32 R) {
  const u32 K2 = f.kf2.k;
  if (!f.kf2.nr || mlen <= K2) return true;
  const u32 cut = mlen - K2;
  const u32 t = R > cut ? R - cut : 0u;
  return kf_present(f.kf2, (s_pk >> (2 * cut)) & ((1ULL << (2 * K2)) - 1), t < K2 ? t : K2 - 1);
}
template <bool INDEL>
__global__ void __launch_bounds__(256) k_search1p(FmView f, Batch b, SearchOut o, u32 ipg, u32 magic) {
  __shared__ u16 q_ent[2048];  // lane | operation << 8
  __shared__ u32 q_n, c_probe;
  constexpr u32 NOPS = INDEL ? 8u : 4u;
  if (threadIdx.x == 0) {
    q_n = 0;
    c_probe = 0;
  }
  __syncthreads();
  // lane -> (group, position): one division per workgroup, a multiplication per lane (exact for the < 512 values it sees)
  const u32 TBX = blockDim.x;  // 64, 128 or 256
  const u32 first = blockIdx.x * TBX;
  const u32 g_first = first / ipg, r_first = first - g_first * ipg;
  const u32 K = f.K, K2 = f.kf2.nr ? f.kf2.k : 0u;
  const u64 kmask = (1ULL << (2 * K)) - 1;
  const u32 lane = threadIdx.x & 63;
  const u32 ngrp2 = (u32)(2 * b.nq);
  u32 mask8 = 0, nprobe = 0;
  {
    const u32 t = r_first + threadIdx.x, qd = (t * magic) >> 16;
    const u32 gid = g_first + qd, pos = t - qd * ipg + 1;
    if (gid < ngrp2) {
      const uint4 raw = *reinterpret_cast<const uint4*>(b.ginfo + gid);
      const u64 qpk = (u64)raw.y << 32 | raw.x;
      const u32 m = raw.z, d_win = raw.w;
      if (m && (d_win & 512u) && pos <= m) {
        const u32 R = m - pos;
        const KfCopy c2 = kf_copy(f.kf2, R < K2 ? R : (K2 ? K2 - 1 : 0u));
        const KfCopy c1 = kf_copy(f.kf, R < K ? R : K - 1);
        const u64 mask2 = K2 ? (1ULL << (2 * K2)) - 1 : 0ULL;
        // all addresses first, then the eight loads back to back, then the bits
        const u32* const idle = reinterpret_cast<const u32*>(f.ktab);  // what a lane without a probe reads
        const u32* addr[NOPS];
        u32 bit[NOPS], word[NOPS], valid = 0, probe = 0;
#pragma unroll
        for (u32 op = 0; op < NOPS; ++op) {
          u64 s_pk;
          u32 mlen, ow;
          const bool ok = cand1<INDEL>(qpk, m, pos, op, s_pk, mlen, ow);
          const bool use2 = K2 && mlen >= K2;
          const bool pr = ok && (use2 || f.kf.nr);
          KfCopy c;
          c.base = use2 ? c2.base : c1.base;
          c.s = use2 ? c2.s : c1.s;
          const u32* a = kf_word(c, use2 ? s_pk & mask2 : s_pk & kmask, bit[op]);
          addr[op] = pr ? a : idle;
          valid |= (u32)ok << op;
          probe |= (u32)pr << op;
        }
#pragma unroll
        for (u32 op = 0; op < NOPS; ++op) word[op] = *addr[op];
#pragma unroll
        for (u32 op = 0; op < NOPS; ++op) {
          const u32 present = ((probe >> op) & 1u) ? (word[op] >> bit[op]) & 1u : 1u;
          mask8 |= (((valid >> op) & 1u) & present) << op;
        }
        nprobe = (u32)__popc(probe);
      }
    }
  }
  nprobe = wave_sum32(nprobe);
  if (lane == 0 && nprobe) atomicAdd(&c_probe, nprobe);
  while (mask8) {
    const u32 op = (u32)__ffs((int)mask8) - 1u;
    mask8 &= mask8 - 1;
    const u32 at = atomicAdd(&q_n, 1u);
    q_ent[at] = (u16)(threadIdx.x | (op << 8));  // at < 2048: eight operations of 256 lanes
  }
  __syncthreads();
  const u32 shard = blockIdx.x & (NSHARD - 1);
  if (threadIdx.x == 0 && c_probe) atomicAdd(&o.ctr->probes[shard], (unsigned long long)c_probe);
  const u32 qn = q_n;
  u32 steps = 0, nlook = 0, nhead = 0;
  for (u32 e0 = 0; e0 < qn; e0 += TBX) {
    if (e0 + (threadIdx.x & ~63u) >= qn) break;  // this wavefront has no survivor to work on
    const u32 e = e0 + threadIdx.x;
    if (e < qn) {
      const u32 ent = q_ent[e], sl = ent & 255u, op = ent >> 8;
      const u32 t = r_first + sl, qd = (t * magic) >> 16;
      const u32 gid = g_first + qd, pos = t - qd * ipg + 1;
      const uint4 raw = *reinterpret_cast<const uint4*>(b.ginfo + gid);
      u64 s_pk;
      u32 mlen, ow;
      (void)cand1<INDEL>((u64)raw.y << 32 | raw.x, raw.z, pos, op, s_pk, mlen, ow);
      u32 lo = 0, hi = 0;
      nhead += (K2 && mlen > K2);
      if (head_window_occurs(f, s_pk, mlen, raw.z - pos)) {
        const KtabEntry iv = ktab_entry(f, s_pk & kmask);
        ++nlook;
        lo = iv.lo;
        hi = iv.hi;
      }
      u64 rs = s_pk >> (2 * K);
      u32 n = mlen - K;
      while (n && lo < hi) {
        bs_extend_code_narrow(f, lo, hi, (u32)rs & 3u);
        rs >>= 2;
        --n;
        ++steps;
      }
      if (lo < hi) {
        const u32 at = atomicAdd(&o.ctr->leaf_cnt[shard], 1u);
        const u32 slot = atomicAdd(o.grp_cnt + gid, 1u);
        if (at < o.shard_cap) {
          Leaf* lf = o.leaves + (u64)shard * o.shard_cap + at;
          lf->qs = gid;
          lf->slot = slot;
          lf->lo = lo;
          lf->hi = hi;
          lf->nops = ow >> 28;
          lf->ops[0] = ow & 0x0FFFFFFFu;
#pragma unroll
          for (int k = 1; k < (int)DMAX; ++k) lf->ops[k] = 0u;
        }
      }
    }
  }
  steps = wave_sum32(steps);
  nlook = wave_sum32(nlook);
  nhead = wave_sum32(nhead);
  if (lane == 0) {
    if (steps) atomicAdd(&o.ctr->steps[shard], (unsigned long long)steps);
    if (nlook) atomicAdd(&o.ctr->lookups[shard], (unsigned long long)nlook);
    if (nhead) atomicAdd(&o.ctr->probes[shard], (unsigned long long)nhead);
  }
}

// k_search1p WITH the select stage (r03).  A workgroup owns whole (query, strand) groups — floor(256 / positions) of them — so
// every string of a group that occurs ends up in this workgroup's LDS (2-bit packed, with its interval), and the group's
// duplicates / substring-minimal filter / std::set order (neighbors.h:29-45, hunter.h:349) are settled right here: leaves never
// travel to HBM, and the scan of the group counts, k_group_pack, k_leaf_alive and k_leaf_rank (62 of the 380 us of a step) have
// nothing left to do.  Kept strings go to the FLAT region of the Sel array — per-shard slices, one atomic per workgroup — and
// selbase[g] / nsel[g] tell the later kernels where a group's strings are.  A workgroup whose strings do not fit the LDS list
// (512; a dozen low-complexity queries side by side) sends its leaves down the generic path like k_search1p does.
struct FlatSel {
  Sel* sel;        // flat region: NSHARD slices of `cap` entries
  u32 cap;
  u32* selbase;    // [2 nq] first Sel slot of a group served here (0xFFFFFFFF: generic path, grp_off based)
  u32* nsel;       // [2 nq]
  u64* key;        // [flat slots] r06: the kept strings themselves (see SEL_CTX_VALID); nullptr = not wanted
};
// r04: (i) the three wavefronts that have nothing to do behind the probe phase END there instead of waiting at the barrier behind
// the dense phase (the usual workgroup has ~40 survivors, one wavefront's worth): the r04a counters showed the kernel resident at
// 6-7 of 8 wavefronts per SIMD, two thirds of the wave cycles waiting — three of four of those slots held by wavefronts parked at
// that barrier; (ii) the LDS list is dynamic (lcap entries, 256 by default: 5.6 KB, 512 when the previous batch's workgroups held
// more than ~64 strings on average) and the survivor queue holds 512 entries, filled in rounds when more survive, so that the
// freed slots can be taken by new workgroups.
static constexpr u32 FUSED_LCAP = 512;   // largest LDS list
static constexpr u32 FUSED_QCAP = 512;   // survivor queue entries per round
static inline u32 fused_lds_bytes(u32 lcap) { return lcap * (8u + 4u + 4u + 2u + 2u + 2u + 2u + 1u); }  // (+ 1: l_ctx, r06)
// TAKE (r04): the workgroup also does k_take's work for its own queries (the occurrences of a query's kept strings in push order:
// take = what hunter.h:349-357 still accepts, a saturating prefix sum) — k_take, 12 us of a 0.34 ms step, is not launched.  Used
// when the whole batch is on the flat path (no generic kernels); gpw is even then, so that both strands of a query sit in one
// workgroup.  (The same round measured k_prepare's work inside this kernel as well: the six lanes that prepare a workgroup's
// queries hold its other 250 up — 0.187 -> 0.247 ms for 17 + 12 us of launches saved; k_prepare stays a launch of its own.)
struct PrepOut {
  u32* qhits;      // [nq] hits per query (k_take's output)
};
template <bool INDEL, bool TAKE>
__global__ void __launch_bounds__(256, 8) k_search1s(FmView f, Batch b, SearchOut o, FlatSel fs, u32 ipg, u32 magic, u32 gpw, u32 lcap, u32 leave, PrepOut po) {  // 8 wavefronts per SIMD: the kernel is bound by requests in flight (r04: 106 SGPRs had left 7)
  __shared__ u16 q_ent[FUSED_QCAP];  // lane | operation << 8
  __shared__ u32 q_n, c_probe, l_n, s_total, s_base;
  __shared__ u32 g_cnt[16], g_start[16], g_alive[16], g_base[16];
  __shared__ unsigned long long g_occ[16];   // TAKE: occurrences of a group's kept strings, each clamped to max_locations
  DG_DYNAMIC_LDS(dyn);  // the list of occurring strings: lcap entries
  unsigned long long* const l_key = reinterpret_cast<unsigned long long*>(dyn);
  u32* const l_lo = reinterpret_cast<u32*>(dyn + (size_t)lcap * 8);
  u32* const l_hi = l_lo + lcap;
  u16* const l_meta = reinterpret_cast<u16*>(l_hi + lcap);  // length | local group << 6 | alive << 15
  u16* const l_pos = l_meta + lcap;
  u16* const l_ord = l_pos + lcap;
  u16* const l_mask = l_ord + lcap;  // filtered intervals (FmView::pre5): which entries of [lo, hi) spell the string's first characters
  u8* const l_ctx = reinterpret_cast<u8*>(l_mask + lcap);  // r06: 8 | code of the character in front of a filtered string's ONE occurrence, 0 = not known
  constexpr u32 NOPS = INDEL ? 8u : 4u;
  if (threadIdx.x == 0) {
    q_n = 0;
    c_probe = 0;
    l_n = 0;
  }
  if (threadIdx.x < 16) {
    g_cnt[threadIdx.x] = g_alive[threadIdx.x] = 0;
    g_occ[threadIdx.x] = 0ULL;
  }
  const u32 ngrp2 = (u32)(2 * b.nq);
  const u32 g_first = blockIdx.x * gpw;
  __syncthreads();
  auto ginfo_of = [&](u32, u32 gid) -> uint4 { return *reinterpret_cast<const uint4*>(b.ginfo + gid); };
  const u32 K = f.K, K2 = f.kf2.nr ? f.kf2.k : 0u;
  const u64 kmask = (1ULL << (2 * K)) - 1;
  const u32 lane = threadIdx.x & 63;
  u32 mask_all = 0, nprobe = 0;
  {
    const u32 lg = (threadIdx.x * magic) >> 16, pos = threadIdx.x - lg * ipg + 1;
    const u32 gid = g_first + lg;
    if (lg < gpw && gid < ngrp2) {
      const uint4 raw = ginfo_of(lg, gid);
      const u64 qpk = (u64)raw.y << 32 | raw.x;
      const u32 m = raw.z, d_win = raw.w;
      if (m && (d_win & 512u) && pos <= m) {
        const u32 R = m - pos;
        const u32* const idle = reinterpret_cast<const u32*>(f.ktab);
        u32 bit[NOPS], word[NOPS], valid = 0, probe = 0;
        if (K2 && K2 <= 18 && m >= K2 + (INDEL ? 1u : 0u)) {  // every string asks the long filter: offsets without building the strings (probe8_long; 32-bit word offsets inside a copy: orders up to 18)
          KfCopy cf;
          u32 off[NOPS];
          probe8_long<INDEL, NOPS>(f, qpk, m, pos, cf, off, bit, valid);
          probe = valid;
          // the eight loads back to back (a lane without a probe reads the copy's first word)
#pragma unroll
          for (u32 op = 0; op < NOPS; ++op) word[op] = cf.base[((valid >> op) & 1u) ? off[op] : 0u];  // (r05: lanes without a probe not loading at all measured the same, 0.148-0.154 ms)
        } else {
          const u32* addr[NOPS];
          const KfCopy c2 = kf_copy(f.kf2, R < K2 ? R : (K2 ? K2 - 1 : 0u));
          const KfCopy c1 = kf_copy(f.kf, R < K ? R : K - 1);
          const u64 mask2 = K2 ? (1ULL << (2 * K2)) - 1 : 0ULL;
#pragma unroll
          for (u32 op = 0; op < NOPS; ++op) {
            u64 s_pk;
            u32 mlen, ow;
            const bool ok = cand1<INDEL>(qpk, m, pos, op, s_pk, mlen, ow);
            const bool use2 = K2 && mlen >= K2;
            const bool pr = ok && (use2 || f.kf.nr);
            KfCopy c;
            c.base = use2 ? c2.base : c1.base;
            c.s = use2 ? c2.s : c1.s;
            const u32* a = kf_word(c, use2 ? s_pk & mask2 : s_pk & kmask, bit[op]);
            addr[op] = pr ? a : idle;
            valid |= (u32)ok << op;
            probe |= (u32)pr << op;
          }
#pragma unroll
          for (u32 op = 0; op < NOPS; ++op) word[op] = *addr[op];
        }
#pragma unroll
        for (u32 op = 0; op < NOPS; ++op) {
          const u32 present = ((probe >> op) & 1u) ? (word[op] >> bit[op]) & 1u : 1u;
          mask_all |= (((valid >> op) & 1u) & present) << op;
        }
        nprobe = (u32)__popc(probe);
      }
    }
  }
  nprobe = wave_sum32(nprobe);
  if (lane == 0 && nprobe) atomicAdd(&c_probe, nprobe);
  const u32 shard = blockIdx.x & (NSHARD - 1);
  u32 steps = 0, nlook = 0, nhead = 0;
  // the dense phase over the queue's first qn entries: survivors rebuilt, table entry, extension; occurring strings to the LDS list
  // (to_lds) or, for a workgroup whose list overflowed, to the generic leaf buffer exactly like k_search1p
  auto dense = [&](const bool to_lds, const u32 qn) {
    for (u32 e0 = 0; e0 < qn; e0 += 256) {
      if (e0 + (threadIdx.x & ~63u) >= qn) break;
      const u32 e = e0 + threadIdx.x;
      if (e < qn) {
        const u32 ent = q_ent[e], sl = ent & 255u, op = ent >> 8;
        const u32 lg = (sl * magic) >> 16, pos = sl - lg * ipg + 1;
        const u32 gid = g_first + lg;
        const uint4 raw = ginfo_of(lg, gid);
        u64 s_pk;
        u32 mlen, ow;
        (void)cand1<INDEL>((u64)raw.y << 32 | raw.x, raw.z, pos, op, s_pk, mlen, ow);
        u32 lo = 0, hi = 0, pre_first = 0xFFFFFFFFu;
        if (to_lds) nhead += (K2 && mlen > K2);
        if ((leave & 2u) || head_window_occurs(f, s_pk, mlen, raw.z - pos)) {  // (leave bit 1: DICEY_EXP=1, no head probe)
          const KtabEntry iv = ktab_entry(f, s_pk & kmask);
          if (to_lds) ++nlook;
          lo = iv.lo;
          hi = iv.hi;
          pre_first = iv.pre_first;
        }
        u64 rs = s_pk >> (2 * K);
        u32 n = mlen - K;
        // r04: a narrow table interval (<= 16 suffixes, the usual case on a genome without repeats) is not extended character by
        // character — mlen - K dependent Occ lines — but FILTERED: the entries of FmView::pre5 say which of its suffixes are
        // preceded by the string's first mlen - K characters (one line, two when the interval straddles); the string then
        // occurs at SA[i] - (mlen - K) for exactly those i, and travels as (interval, mask) instead of its own interval
        u32 fmask = 0, fpre = 0, cx = 0;
        if (to_lds && f.pre5 && n >= 1 && n <= 5 && lo < hi && hi - lo <= 16) {
          const u32 w = hi - lo;
          u32 want = 0;
          for (u32 k2 = 0; k2 < n; ++k2) want |= ((u32)(rs >> (2 * k2)) & 3u) << (3 * k2);
          const u32 wmask = (1u << (3 * n)) - 1u;
          u32 before = 7u;  // the character in front of the string's (single) occurrence: the entry's character n + 1, when it has one
          if (w == 1 && pre_first != 0xFFFFFFFFu) {
            // r05: the interval holds ONE suffix and the table entry carries its preceding characters: no line of pre5
            fmask = (u32)((pre_first & wmask) == want);
            before = (pre_first >> (3 * n)) & 7u;
          } else {
            u32 ent16[16];
#pragma unroll
            for (u32 j = 0; j < 16; ++j) ent16[j] = j < w ? (u32)f.pre5[(u64)lo + j] : 0xFFFFu;
#pragma unroll
            for (u32 j = 0; j < 16; ++j) {
              const bool hit = (ent16[j] & wmask) == want && j < w;
              fmask |= (u32)hit << j;
              before = hit ? (ent16[j] >> (3 * n)) & 7u : before;
            }
            ++nlook;
          }
          if (n <= 4 && before < 4u && (fmask & (fmask - 1u)) == 0u) cx = 8u | before;  // (one occurrence; five characters per entry)
          fpre = n;
          n = 0;
          if (!fmask) lo = hi = 0;
        }
        while (n && lo < hi) {
          bs_extend_code_narrow(f, lo, hi, (u32)rs & 3u);
          rs >>= 2;
          --n;
          if (to_lds) ++steps;
        }
        if (lo < hi) {
          if (to_lds) {
            const u32 at = atomicAdd(&l_n, 1u);
            if (at < lcap) {
              l_key[at] = s_pk;
              l_lo[at] = lo;
              l_hi[at] = hi;
              l_meta[at] = (u16)(mlen | (lg << 6) | (fpre << 10) | (fpre ? 0x2000u : 0u));
              l_mask[at] = (u16)fmask;
              l_ctx[at] = (u8)cx;
            }
          } else {
            const u32 at = atomicAdd(&o.ctr->leaf_cnt[shard], 1u);
            const u32 slot = atomicAdd(o.grp_cnt + gid, 1u);
            if (at < o.shard_cap) {
              Leaf* lf = o.leaves + (u64)shard * o.shard_cap + at;
              lf->qs = gid;
              lf->slot = slot;
              lf->lo = lo;
              lf->hi = hi;
              lf->nops = ow >> 28;
              lf->ops[0] = ow & 0x0FFFFFFFu;
#pragma unroll
              for (int k = 1; k < (int)DMAX; ++k) lf->ops[k] = 0u;
            }
          }
        }
      }
    }
  };
  // survivors enter the queue in rounds of at most FUSED_QCAP; `gone`: this wavefront ended behind the probe phase
  bool gone = false, single_round = false;
  auto rounds = [&](const bool to_lds, const bool may_leave) {
    u32 rem = mask_all;
    for (u32 round = 0;; ++round) {
      while (rem) {
        const u32 op = (u32)__ffs((int)rem) - 1u;
        const u32 at = atomicAdd(&q_n, 1u);
        if (at >= FUSED_QCAP) break;  // next round
        rem &= rem - 1;
        q_ent[at] = (u16)(threadIdx.x | (op << 8));
      }
      __syncthreads();
      const u32 raw_n = q_n, qn = raw_n < FUSED_QCAP ? raw_n : FUSED_QCAP;
      if (round == 0) {
        if (to_lds && threadIdx.x == 0 && c_probe) atomicAdd(&o.ctr->probes[shard], (unsigned long long)c_probe);
        single_round = raw_n <= FUSED_QCAP;
        // one wavefront's worth of survivors and nothing left over: the other wavefronts end here, their slots go to the next workgroup
        if (may_leave && raw_n <= 64 && threadIdx.x >= 64) {
          gone = true;
          return;
        }
      }
      dense(to_lds, qn);
      if (raw_n <= FUSED_QCAP) return;
      __syncthreads();
      if (threadIdx.x == 0) q_n = 0;
      __syncthreads();
    }
  };
  if (leave & 4u) mask_all = 0;  // DICEY_EXP=2: the probe phase alone (no survivor is followed up)
  rounds(true, (leave & 1u) != 0);
  if (gone) return;
  steps = wave_sum32(steps);
  nlook = wave_sum32(nlook);
  nhead = wave_sum32(nhead);
  if (lane == 0) {
    if (steps) atomicAdd(&o.ctr->steps[shard], (unsigned long long)steps);
    if (nlook) atomicAdd(&o.ctr->lookups[shard], (unsigned long long)nlook);
    if (nhead) atomicAdd(&o.ctr->probes[shard], (unsigned long long)nhead);
  }
  __syncthreads();
  const u32 nl = l_n;
  if (nl > lcap) {  // this workgroup's groups take the generic path (selbase stays "generic")
    if (single_round) dense(false, q_n < FUSED_QCAP ? q_n : FUSED_QCAP);  // the queue still holds every survivor
    else {
      __syncthreads();
      if (threadIdx.x == 0) q_n = 0;
      __syncthreads();
      rounds(false, false);
    }
    return;
  }
  // ---- select, per group, in LDS.  Up to 64 strings (the usual workgroup: 12 groups of two or three): by the first wavefront
  // alone.  More strings (repeat families: hundreds per workgroup): all four wavefronts share the pair loops (one wavefront alone
  // took 0.74 instead of 0.51 ms per step on the repeats genome).
  const u32 sstep = nl <= 64 ? 64u : 256u;
  if (threadIdx.x >= sstep) return;
  for (u32 i = threadIdx.x; i < nl; i += sstep) l_pos[i] = (u16)atomicAdd(&g_cnt[(l_meta[i] >> 6) & 15u], 1u);
  __syncthreads();
  if (threadIdx.x == 0) {
    u32 run = 0;
    for (u32 g = 0; g < gpw; ++g) {
      g_start[g] = run;
      run += g_cnt[g];
    }
  }
  __syncthreads();
  for (u32 i = threadIdx.x; i < nl; i += sstep) l_ord[g_start[(l_meta[i] >> 6) & 15u] + l_pos[i]] = (u16)i;
  __syncthreads();
  // alive: no other string of the group is a proper substring, and of equal strings the first of the list stays
  for (u32 i = threadIdx.x; i < nl; i += sstep) {
    const u32 meta = l_meta[i], alen = meta & 63u, lg = (meta >> 6) & 15u;
    bool ok = true;
    if (INDEL) {
      const u64 a = l_key[i];
      const u32 s0 = g_start[lg], s1 = s0 + g_cnt[lg];
      for (u32 j = s0; j < s1 && ok; ++j) {
        const u32 x = l_ord[j];
        if (x == i) continue;
        const u32 xlen = l_meta[x] & 63u;
        if (xlen > alen) continue;
        const u64 xk = l_key[x], xm = xlen >= 32 ? ~0ULL : ((1ULL << (2 * xlen)) - 1);
        bool hit = false;
        for (u32 sh = 0; sh <= alen - xlen; ++sh) hit = hit || (((a >> (2 * sh)) & xm) == xk);
        if (hit) ok = (xlen == alen) && (i < x);
      }
    }
    if (ok) {
      l_meta[i] = (u16)(meta | 0x8000u);
      atomicAdd(&g_alive[lg], 1u);
      if (TAKE) {
        const u64 occ = (meta & 0x2000u) ? (u64)__popc((u32)l_mask[i]) : (u64)l_hi[i] - l_lo[i];
        atomicAdd(&g_occ[lg], (unsigned long long)(occ < b.max_locations ? occ : b.max_locations));
      }
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    u32 total = 0;
    for (u32 g = 0; g < gpw; ++g) {
      g_base[g] = total;
      total += g_alive[g];
    }
    s_total = total;
    s_base = total ? atomicAdd(&o.ctr->sel_cnt[shard], total) : 0u;
  }
  __syncthreads();
  const u32 wbase = s_base;
  const bool room = wbase + s_total <= fs.cap;  // an overflowing slice repeats the batch (Summary::worst_sel) ...
  if (!room && threadIdx.x == 0) atomicOr(&o.ctr->overflow, 1u);  // ... and the kernels behind this one do nothing
  if (threadIdx.x == 0 && nl) atomicAdd(&o.ctr->fused_leaves[shard], (unsigned long long)nl);
  // rank among the group's survivors in std::string order (A < C < G < T = code order; a proper prefix sorts first) -> Sel
  for (u32 i = threadIdx.x; i < nl; i += sstep) {
    const u32 meta = l_meta[i];
    if (!(meta & 0x8000u)) continue;
    const u32 alen = meta & 63u, lg = (meta >> 6) & 15u;
    const u64 ak = l_key[i] << (64 - 2 * alen);
    const u32 s0 = g_start[lg], s1 = s0 + g_cnt[lg];
    u32 r = 0;
    u64 before = (TAKE && (lg & 1u)) ? g_occ[lg - 1] : 0ULL;  // TAKE: occurrences (clamped) of the strings in front of this one in push order
    for (u32 j = s0; j < s1; ++j) {
      const u32 x = l_ord[j], xm = l_meta[x];
      if (x == i || !(xm & 0x8000u)) continue;
      const u32 xlen = xm & 63u;
      const u64 xk = l_key[x] << (64 - 2 * xlen);
      const bool first = (xk < ak) || (xk == ak && xlen < alen);
      r += first;
      if (TAKE && first) {
        const u64 occ = (xm & 0x2000u) ? (u64)__popc((u32)l_mask[x]) : (u64)l_hi[x] - l_lo[x];
        before += occ < b.max_locations ? occ : b.max_locations;
      }
    }
    if (room) {
      Sel sv;
      sv.lo = l_lo[i];
      sv.hi = l_hi[i];
      sv.len = (meta & 0x2000u) ? sel_len_filtered(alen, (meta >> 10) & 7u, (u32)l_mask[i]) | ((u32)l_ctx[i] << SEL_CTX_SHIFT) : alen;
      sv.take = 0;
      sv.hbase = 0;
      if (TAKE) {  // hunter.h:349-357: strings are located in set order, forward strand first, while hits < max_locations
        const u64 M = b.max_locations, occ = sel_occ(sv);
        const u64 h0 = before < M ? before : M, h1 = before + occ < M ? before + occ : M;
        sv.hbase = (u32)h0;
        sv.take = (u32)(h1 - h0);
      }
      sv.g = g_first + lg;
      fs.sel[(u64)shard * fs.cap + wbase + g_base[lg] + r] = sv;
      if (fs.key && (sv.len & SEL_CTX_VALID)) fs.key[(u64)shard * fs.cap + wbase + g_base[lg] + r] = l_key[i];
    }
  }
  if (threadIdx.x < gpw && g_first + threadIdx.x < ngrp2) {
    const u32 gid = g_first + threadIdx.x;
    const uint4 raw = ginfo_of(threadIdx.x, gid);
    if (raw.z && (raw.w & 512u)) {  // groups this kernel searches: their strings are in the flat region, also when there are none
      fs.nsel[gid] = room ? g_alive[threadIdx.x] : 0u;
      fs.selbase[gid] = shard * fs.cap + wbase + g_base[threadIdx.x];
    }
  }
  if (TAKE && threadIdx.x < gpw / 2) {  // k_take's per-query part: the hit count, hunter.h:434, the compact results' word
    const u64 q = (u64)(g_first / 2) + threadIdx.x;
    if (q < b.nq) {
      const u64 M = b.max_locations, tot = g_occ[2 * threadIdx.x] + g_occ[2 * threadIdx.x + 1];
      const u64 hits = room ? (tot < M ? tot : M) : 0ULL;
      po.qhits[q] = (u32)hits;
      u32 fl = b.qflags[q];
      if (hits >= M && !(fl & DG_Q_TOO_SHORT)) {
        fl |= DG_Q_MAX_MATCHES;
        b.qflags[q] = fl;
      }
      if (b.qinfo) b.qinfo[q] = (fl & 255u) | ((b.qdist[q] & 255u) << 8) | (b.qnondna[q] << 16);
    }
  }
}

// (r02 also measured k_search1p cut in two kernels — probe, then finish from a survivor queue in HBM: 0.21 + 0.19 ms against 0.25 ms
// fused; removed in r04.)
// ------------------------------------------------------------------------------------------------------------
// Distance 2 (edit mode) laid out flat as well.  r02 profile of the state machine at d = 2: 124 ms per 100 000 20-mers,
// 1.7 G filter probes + 0.36 G table reads + 0.59 G interval extensions — issue bound like its d = 1 form was, and most of
// its memory accesses belong to strings that do not occur.  Here:
//   * one WORKGROUP per (query, strand); its four wavefronts walk the pairs of edit positions (p2 <= p1, counted as "the
//     operation sits right of q[0..p)"), one wavefront per pair, one LANE per pair of operations (8 x 8: delete, substitute
//     by the three other bases, insert A/C/G/T) — every two-operation path of the trie k_search<true,2> walks
//     (neighbors.h:47-83), so the same strings, duplicates included, reach the select stage;
//   * a lane applies its two operations to the 2-bit packed query with shifts and masks and asks the LONG presence filter
//     (order K2, FmView::kf2) about the last K2 characters; strings shorter than K2 ask the table's filter.  U pairs per
//     wavefront are in flight at once (the probes are independent loads).  Neighbouring pairs differ in p1 only, lanes pick
//     the filter copy by p1's window position, so the probes of a workgroup keep hitting the lines it already has in L1/L2;
//   * survivors (about 1 % of the candidates behind a 19-mer filter on a 3.1 Gb genome) are pushed on an LDS stack; whenever
//     it holds 256 of them the workgroup pops 256 and runs them densely: table entry, interval extension over the
//     characters left of the table window, leaf record.  Leaf slots of the group come from an LDS counter, leaf space from
//     one atomic per wavefront.
// Taken: queries of 2-edit budget without N, up to 30 nt (the edited string fits 64 bits) and at least K + 2 long.
DG_DEV void apply_edit(u64 pk, u32 len, u32 pos, u32 op, u64& out, u32& olen, u32& word) {
  const u32 R = len - pos;  // characters right of the operation
  const u64 low = pk & ((1ULL << (2 * R)) - 1);
  const u32 old = (u32)(pk >> (2 * R)) & 3u;
  if (op == 0) {
    out = low | ((pk >> (2 * R + 2)) << (2 * R));
    olen = len - 1;
    word = (pos << 4) | (OP_D << 2);
  } else if (op < 4) {
    const u32 c = (old + op) & 3u;  // neighbors.h:63: a different base
    out = pk ^ ((u64)(old ^ c) << (2 * R));
    olen = len;
    word = (pos << 4) | (OP_S << 2) | c;
  } else {
    const u32 c = op - 4;
    out = low | ((u64)c << (2 * R)) | ((pk >> (2 * R)) << (2 * R + 2));
    olen = len + 1;
    word = (pos << 4) | (OP_I << 2) | c;
  }
}

// Hamming mode of k_search2p (r05): operation 0 of either loop means "no edit here" instead of a deletion, so that the pairs of
// positions also carry the strings with fewer than two substitutions (the Hamming ball keeps all of them, neighbors.h:57-66): the
// sequence itself and the substitution of its first character on lane (1, 1), a single substitution at p1 >= 2 on lane (1, p1).
static constexpr u32 OPW_NONE = 0xFFFFFFFFu;
DG_DEV void apply_edit_h(bool ham, u64 pk, u32 len, u32 pos, u32 op, u64& out, u32& olen, u32& word) {
  if (ham && op == 0) {
    out = pk;
    olen = len;
    word = OPW_NONE;
  } else apply_edit(pk, len, pos, op, out, olen, word);
}

// The edited string alone (no operation word), the three kinds in one expression: the characters right of the position stay, the ones
// left of it move by the kind's length change, a substituted / inserted character goes in between ("none" = the old character back).
DG_DEV u64 edit_string(bool ham, u64 pk, u32 len, u32 pos, u32 op, u32& olen) {
  const u32 R2 = 2 * (len - pos);
  const bool del = op == 0 && !ham, ins = op >= 4;
  const u32 old = (u32)(pk >> R2) & 3u;
  const u32 c = ins ? op - 4 : (old + op) & 3u;  // (Hamming, op 0: the old character)
  const u64 left = pk >> (ins ? R2 : R2 + 2);
  olen = del ? len - 1 : ins ? len + 1 : len;
  return (pk & ((1ULL << R2) - 1)) | (del ? 0ULL : (u64)c << R2) | (left << (del ? R2 : R2 + 2));
}

// (the lane-per-operation-pair kernel described above — k_search2<U>, r02: 17.4 ms — was removed in r04; k_search2p below is the
// same enumeration with one lane per pair of POSITIONS)

// k_search2 with one lane per PAIR OF POSITIONS: the lane walks the 8 x 8 operations in two fully unrolled loops (every
// operation is a compile-time constant where it is applied, the eight probes of an inner loop are independent loads), skips
// the operations that only repeat another lane's string — deleting the left one of two equal neighbours, inserting a base
// right of an equal one; checked against the reference's minimal set on random and low-complexity queries — and queues
// survivors as (pair, operation, operation) for the dense phase, which rebuilds them.  ~30 vector instructions per candidate
// instead of ~150, and a fifth fewer candidates (12 160 -> ~9 700 for a 20-mer).
DG_DEV void pair_of(u32 w, u32 m, u32& p2, u32& p1) {  // rows p2 = 1, 2, ... hold m, m-1, ... pairs; row a = p2-1 starts at a(2m+1-a)/2
  const float tm = (float)(2 * m + 1);
  int a = (int)((tm - sqrtf(tm * tm - 8.0f * (float)w)) * 0.5f);
  if (a < 0) a = 0;
  if (a > (int)m - 2) a = (int)m - 2;
  while (a > 0 && (u32)a * (2 * m + 1 - (u32)a) / 2 > w) --a;
  while ((u32)(a + 1) * (2 * m - (u32)a) / 2 <= w) ++a;
  p2 = (u32)a + 1;
  p1 = p2 + (w - (u32)a * (2 * m + 1 - (u32)a) / 2);
}
DG_DEV u32 pair_rows4_count(u32 m) {
  u32 n = 0;
  for (u32 p2 = 1; p2 < m; ++p2) n += (m - p2 + 4) & ~3u;
  return n;
}
DG_DEV bool pair_of_rows4(u32 w, u32 m, u32& p2, u32& p1) {
  u32 start = 0;
  for (p2 = 1; p2 < m; ++p2) {
    const u32 pl = (m - p2 + 4) & ~3u;
    if (w < start + pl) break;
    start += pl;
  }
  const u32 j = w - start;
  p1 = m - j;
  return p2 < m && j <= m - p2;
}
static constexpr u32 FUSED2_LCAP = 512;
static constexpr u32 FUSED2_HCAP = 1024;  // hash slots of the select stage (distinct strings of the list)  // strings of ONE (query, strand) group in LDS (k_search2p<true, .>)
// SEL (r04): the select stage inside, like k_search1s — the workgroup owns its group anyway, so the group's occurring strings stay in
// LDS (2-bit packed, with interval and filter word), duplicates / substring-minimal filter / std::set order are settled here and the
// kept strings go to the flat Sel region; leaves do not travel to HBM, and the scans, k_group_pack, k_group_select, k_leaf_alive and
// k_leaf_rank (2.4 of the 12.8 ms of a step) have nothing left to do.  A group with more than FUSED2_LCAP occurring strings is searched
// once more with its leaves written out for the generic select kernels (selbase stays "generic"; the host repeats the batch with
// them when they were not launched).  (A form with one workgroup per QUERY that also did k_take's work — both strands one after the
// other — was measured: 9.68 against 9.28 + 0.27 ms for this kernel and k_take; removed.)
// LONG2 (r05): every group this launch takes has its shortest string (two deletions; none in Hamming mode) at least K2 characters long
// — Batch::fast2_minlen makes k_prepare leave shorter queries to the walker — so every probe asks the long filter and a probe's place in
// the filter is an OR of parts computed once per lane (nine base places, the images of the edited characters' code bits: filt_pos, see
// the body) instead of a string built per probe.  !LONG2 is the r04 body: batches that hold shorter queries (the handle remembers).
// Two instantiations, not a runtime branch: with both bodies in one kernel 13 registers of the common path went to scratch.
template <bool SEL, bool LONG2>
__global__ void __launch_bounds__(256, LONG2 ? 6 : 5) k_search2p(FmView f, Batch b, SearchOut o, u32 filt_ok, FlatSel fs, u32 lcap2, u32 hamming) {
  const u32 expb = hamming >> 8;  // DICEY_EXP (measurement aid, wrong results): 1 = no dense phase, 2 = no head-window probe
  const bool ham = (hamming & 1u) != 0;  // substitutions only; strings with 0, 1 and 2 of them (r05: Hamming distance 2 used to walk k_search<false, 2>, 3.6 x slower per query than the edit form here)
  // Survivors of a pass are kept as one 64-bit mask per lane (bit 8*op1 + op2) instead of a queue of entries: 3 KB of LDS
  // whatever survives (a queue that holds every candidate of a pass needs 32 KB and left four wavefronts per SIMD resident;
  // r02: 8.8 -> 7.6 ms with room for six), and nothing can overflow.  The dense phase numbers the set bits with a prefix sum
  // over the lanes and finds its e-th one by a binary search over the prefix plus a select inside the lane's mask.
  // r05: a pass with at most QS_CAP survivors (the usual case: ~ 925 of a 20-mer's 13 000 candidates) lists them in the same LDS
  // instead — (lane | operation pair << 8), two bytes each, written by the lanes that found them — and stage 1 of the dense phase
  // reads its survivor with one load; the masks and the search over the prefix remain for passes with more.
  constexpr u32 QS_CAP = (256 * 8 + 257 * 4) / 2;
  __shared__ __align__(8) unsigned char q_raw[256 * 8 + 257 * 4 + 4];
  unsigned long long* const q_mask = reinterpret_cast<unsigned long long*>(q_raw);
  u32* const q_ex = reinterpret_cast<u32*>(q_raw + 256 * 8);  // exclusive prefix of the lanes' survivor counts, [256] = total
  u16* const q_surv = reinterpret_cast<u16*>(q_raw);
  __shared__ u32 q_wave[4];
  __shared__ u16 q_pair[256];   // the lanes' (p1 | p2 << 8) of the pass: the dense phase rebuilds other lanes' survivors
  __shared__ u16 q_cand[512];   // survivors whose head window occurs as well (lane | operation pair << 8), for the dense phase's second stage
  __shared__ u32 q_wc[2][4];    // the wavefronts' counts of them, by the parity of the first stage's round
  __shared__ u32 g_slots;
  constexpr u32 LN = SEL ? FUSED2_LCAP : 1u;
  __shared__ u64 l_key[LN];
  __shared__ u32 l_lo[LN], l_hi[LN], l_fw[LN];
  __shared__ u16 l_meta[LN];  // length (6 bits), 0x8000 = kept
  __shared__ u32 l_rank[LN];  // 0xFFFFFFFF = dead, else the rank among the kept strings
  __shared__ u32 l_hash[SEL ? FUSED2_HCAP : 1u];  // list indices of the distinct strings
  __shared__ u16 l_ord[LN];   // the kept strings' list indices
  __shared__ u32 l_n, s_alive, s_base;
  const u32 K = f.K, K2 = f.kf2.nr ? f.kf2.k : 0u;
  const u64 kmask = (1ULL << (2 * K)) - 1, mask2 = K2 ? (1ULL << (2 * K2)) - 1 : 0ULL;
  const u32 lane = threadIdx.x & 63;
  const u32 shard = blockIdx.x & (NSHARD - 1);
  // (the launch's statistics — extension steps, table / pre5 reads, filter probes — go through LDS once per pass: as per-lane sums they
  //  were three registers alive across the whole kernel, which the compiler kept in scratch at five wavefronts per SIMD)
  __shared__ u32 s_cnt[3];
  if (threadIdx.x < 3) s_cnt[threadIdx.x] = 0;
  // one pass over the group's pairs of edit positions; to_lds: occurring strings go to the LDS list, else to leaf records in HBM
  auto search = [&](const u32 gid, const u32 m, const u64 qpk, const bool to_lds, const bool count) {
    const u32 npairs = m * (m + 1) / 2 - 1;  // p2 = 1..m-1, p1 = p2..m
    for (u32 w0 = 0; w0 < npairs; w0 += 256) {
      const u32 w = w0 + threadIdx.x;
      u64 surv = 0;
      u32 steps = 0, nlook = 0, nprobe = 0;
      if (w < npairs) {
        u32 p1, p2;
        pair_of(w, m, p2, p1);
        q_pair[threadIdx.x] = (u16)(p1 | (p2 << 8));
        const u32 R1 = m - p1;
        const KfCopy c2 = kf_copy(f.kf2, R1 < K2 ? R1 : (K2 ? K2 - 1 : 0u));
        const KfCopy c1 = kf_copy(f.kf, R1 < K ? R1 : K - 1);
        const u32 qa = (u32)(qpk >> (2 * R1)) & 3u;                       // q[p1-1]
        const u32 qb = R1 ? (u32)(qpk >> (2 * R1 - 2)) & 3u : 4u;         // q[p1], 4 = none
        const u32 q2a = (u32)(qpk >> (2 * (m - p2))) & 3u;                // q[p2-1]
        const u32 q2b = (u32)(qpk >> (2 * (m - p2) - 2)) & 3u;            // q[p2] (p2 < m)
        if constexpr (LONG2) {
          // r05: the window code of a twice-edited string is (code of the string with both edited characters' bits zero) | first
          // character's bits | second character's bits, and so is its place in the lane's copy of the filter (a permutation of the code's
          // bits, filt_pos).  Nine base places per lane — (deletion, substitution, insertion) x the same for the second edit — and the
          // images of the two code bits of four character slots; a probe's word offset and bit number are then one three-way OR each.
          // ~85 vector instructions per eight probes, loads and bit tests included (the r04 body: ~310); 8 - 16 loads back to back before
          // the first word is looked at (`batch` below).  With the wavefront scan on data-parallel primitives and the statistics in LDS
          // the kernel fits 80 registers: six wavefronts per SIMD (r04: five, r03: four).
          // Measured and not kept (tools/r05_call10.sh, r05_variants.sh): the rounds as a two-deep software pipeline (spills at 96
          // registers), rows of pairs padded to quads of lanes (same number of vector-cache look-ups), seven wavefronts (8-12 B of scratch).
          const u32 s = c2.s;
          const u32 b1 = 2 * R1;
          auto slot_img = [&](u32 bpos, u32 (&io)[2], u32 (&ib)[2]) {  // (a slot left of the window has no image)
            const FiltPos lo = filt_pos((1ULL << bpos) & mask2, s), both = filt_pos((3ULL << bpos) & mask2, s);
            io[0] = lo.off;
            ib[0] = lo.bit;
            io[1] = both.off ^ lo.off;
            ib[1] = both.bit ^ lo.bit;
          };
          auto img_of = [](u32 c, const u32 (&i2)[2]) { return ((c & 1u) ? i2[0] : 0u) | ((c & 2u) ? i2[1] : 0u); };
          u32 e1o[2], e1b[2];
          slot_img(b1, e1o, e1b);
          const u64 low1 = qpk & ((1ULL << b1) - 1);
          // what one kind of first operation (0 deletion — Hamming: none —, 1 substitution, 2 insertion) leaves for the second
          struct Kind {
            u32 boff[3];   // base word offsets: second operation deletion (Hamming: none), substitution, insertion
            u32 so[4];     // word-offset images of the substituted characters (second operation 1..3)
            u32 io[2];     // ... of the slot's two code bits (insertions)
            u32 bits[2];   // bit numbers of the eight second operations, one per byte, without the first character's part
          };
          auto kind = [&](u32 k1) {
            const u64 s1 = k1 == 0 ? (ham ? qpk : low1 | ((qpk >> (b1 + 2)) << b1)) : k1 == 1 ? qpk & ~(3ULL << b1) : low1 | ((qpk >> b1) << (b1 + 2));
            const u32 l1 = k1 == 0 ? (ham ? m : m - 1) : k1 == 1 ? m : m + 1;
            const u32 b2 = 2 * (l1 - p2);
            const u64 low2 = s1 & ((1ULL << b2) - 1);
            Kind kd;
            const FiltPos pd = filt_pos((ham ? s1 : low2 | ((s1 >> (b2 + 2)) << b2)) & mask2, s);
            const FiltPos ps = filt_pos(s1 & ~(3ULL << b2) & mask2, s);
            const FiltPos pi = filt_pos((low2 | ((s1 >> b2) << (b2 + 2))) & mask2, s);
            kd.boff[0] = pd.off;
            kd.boff[1] = ps.off;
            kd.boff[2] = pi.off;
            u32 ib[2];
            slot_img(b2, kd.io, ib);
            kd.bits[0] = pd.bit;
            kd.so[0] = 0;
#pragma unroll
            for (u32 op2 = 1; op2 < 4; ++op2) {
              kd.so[op2] = img_of((q2a + op2) & 3u, kd.io);
              kd.bits[0] |= (ps.bit | img_of((q2a + op2) & 3u, ib)) << (8 * op2);
            }
            kd.bits[1] = (pi.bit * 0x01010101u) | (ib[0] << 8) | (ib[1] << 16) | ((ib[0] | ib[1]) << 24);
            return kd;
          };
          // Five batches of probes, each N1 first operations of ONE shape x NQ second operations: first operations of one shape differ
          // in a character whose code bits lie inside the lane's in-line field, so a batch asks for NQ lines, N1 times each back to
          // back — and no line is asked for in two batches.  (By rounds of one first operation x eight second ones, the lines of the
          // substitution rounds and of the insertion rounds came round two and four times with other wavefronts' lines in between:
          // 120 M of a launch's 277 M requests were L2 hits on lines the vector cache had held a moment before.)  One shape per batch
          // also means one Kind alive at a time.
          auto batch = [&](const Kind& kk, auto G1, auto N1, auto Q0, auto NQ) {
            constexpr u32 g1 = decltype(G1)::value, n1 = decltype(N1)::value, q0 = decltype(Q0)::value, nq = decltype(NQ)::value;
            constexpr u32 k1 = g1 == 0 ? 0u : g1 >= 4 ? 2u : 1u;
            constexpr bool ins1 = k1 == 2;
            u32 off[n1][nq], bitn[n1][2], valid[n1];
            bool any = false;
#pragma unroll
            for (u32 h = 0; h < n1; ++h) {
              const u32 op1 = g1 + h;
              bool v1 = (p1 > p2 || ins1) && !(p1 == m && ins1);
              if (op1 == 0) v1 = v1 && qb != qa;
              if (ins1) v1 = v1 && !(p1 >= 2 && qa == op1 - 4 && p2 + 2 <= p1);
              if (ham) v1 = (op1 >= 1 && op1 <= 3) || (op1 == 0 && p1 == 1);
              const u32 posp = ins1 ? p1 : p1 - 1;
              const u32 c1 = k1 == 1 ? (qa + op1) & 3u : k1 == 2 ? op1 - 4 : 0u;
              const u32 i1o = img_of(c1, e1o), i1b = img_of(c1, e1b) * 0x01010101u;
              u32 vm = 0;
#pragma unroll
              for (u32 q = 0; q < nq; ++q) {
                const u32 op2 = q0 + q;
                bool v2 = true;
                if (op2 == 0) v2 = !(p2 < posp && q2b == q2a);
                if (op2 >= 4) v2 = !(p2 >= 2 && q2a == op2 - 4);
                if (ham) v2 = (op2 >= 1 && op2 <= 3) ? (p2 < p1 && op1 != 0) : (op2 == 0 && p2 == 1);
                v2 = v2 && v1;
                const u32 k2 = op2 == 0 ? 0u : op2 < 4 ? 1u : 2u;
                const u32 i2o = k2 == 1 ? kk.so[op2] : k2 == 2 ? img_of(op2 - 4, kk.io) : 0u;
                off[h][q] = kk.boff[k2] | i1o | i2o;
                vm |= (u32)v2 << q;
              }
              bitn[h][0] = kk.bits[q0 >> 2] | i1b;
              bitn[h][1] = kk.bits[1] | i1b;  // (second half: batches of eight second operations only)
              valid[h] = vm;
              any = any || v1;
            }
            if (any) {
              u32 word[n1][nq];
#pragma unroll
              for (u32 q = 0; q < nq; ++q)  // (a line's probes next to each other)
#pragma unroll
                for (u32 h = 0; h < n1; ++h) {  // lanes without a probe do not load (reading their copy's first word instead: a quarter more look-ups in the vector cache)
                  word[h][q] = 0;
                  if ((valid[h] >> q) & 1u) word[h][q] = c2.base[off[h][q]];
                }
#pragma unroll
              for (u32 h = 0; h < n1; ++h) {
                u32 maskq = 0;
#pragma unroll
                for (u32 q = 0; q < nq; ++q) maskq |= ((word[h][q] >> ((bitn[h][q >> 2] >> (8 * (q & 3u))) & 31u)) & 1u) << q;
                maskq &= valid[h];
                nprobe += (u32)__popc(valid[h]);
                surv |= (u64)maskq << (8 * (g1 + h) + q0);
              }
            }
          };
          using std::integral_constant;
          {
            const Kind kd = kind(0);
            batch(kd, integral_constant<u32, 0>{}, integral_constant<u32, 1>{}, integral_constant<u32, 0>{}, integral_constant<u32, 8>{});
          }
          {
            const Kind kd = kind(1);
            batch(kd, integral_constant<u32, 1>{}, integral_constant<u32, 3>{}, integral_constant<u32, 0>{}, integral_constant<u32, 4>{});
            batch(kd, integral_constant<u32, 1>{}, integral_constant<u32, 3>{}, integral_constant<u32, 4>{}, integral_constant<u32, 4>{});
          }
          if (!ham) {  // (uniform) no insertions in Hamming mode
            const Kind kd = kind(2);
            batch(kd, integral_constant<u32, 4>{}, integral_constant<u32, 4>{}, integral_constant<u32, 0>{}, integral_constant<u32, 4>{});
            batch(kd, integral_constant<u32, 4>{}, integral_constant<u32, 4>{}, integral_constant<u32, 4>{}, integral_constant<u32, 4>{});
          }
        } else {
#pragma unroll
        for (u32 op1 = 0; op1 < 8; ++op1) {
          const bool ins1 = op1 >= 4;
          // the first operation leaves p2 characters to its left; nothing is inserted after the last character
          bool v1 = (p1 > p2 || ins1) && !(p1 == m && ins1);
          if (op1 == 0) v1 = v1 && qb != qa;
          if (ins1) v1 = v1 && !(p1 >= 2 && qa == op1 - 4 && p2 + 2 <= p1);
          if (ham) v1 = (op1 >= 1 && op1 <= 3) || (op1 == 0 && p1 == 1);  // (p1 == 1 implies p2 == 1)
          if (v1) {
            u64 s1;
            u32 l1, w1;
            apply_edit_h(ham, qpk, m, p1, op1, s1, l1, w1);
            const u32 posp = ins1 ? p1 : p1 - 1;  // characters left of the first operation
            // (r04 measured the copy of the filter picked by the SECOND operation's position instead — the inner loop's eight probes then
            //  share three lines: 161 M instead of 175 M fabric reads per launch, but the L2 hits between neighbouring lanes' probes go
            //  (175 M -> 46 M) and the kernel takes 8.93 instead of 8.20 ms; profiles/r04_d2b_pmc_summary.csv.  Not kept.)
            u32 mask8 = 0;
            // all addresses first, then the eight loads back to back, then the bits.  (r04 also measured the rounds as an explicit
            // software pipeline — round k + 1's loads issued before round k's words are consumed, no branches inside a round: 155
            // VGPRs, three wavefronts per SIMD, 10.8 ms; held to 128 VGPRs with 64 B of scratch 9.7 ms; this form 9.3 ms on the same
            // box.  tools/r04_call19.sh.  Not kept: residency is worth more to this kernel than loads in flight per wavefront.)
            u32 bit[8], word[8], valid = 0, probe = 0;
            {
            const u32* const idle = reinterpret_cast<const u32*>(f.ktab);  // what a lane without a probe reads
            const u32* addr[8];
#pragma unroll
            for (u32 op2 = 0; op2 < 8; ++op2) {
              bool v2 = true;
              if (op2 == 0) v2 = !(p2 < posp && q2b == q2a);
              if (op2 >= 4) v2 = !(p2 >= 2 && q2a == op2 - 4);
              // Hamming: a second substitution left of a first one; "none" on the lanes of p2 = 1 only (one lane per first edit)
              if (ham) v2 = (op2 >= 1 && op2 <= 3) ? (p2 < p1 && op1 != 0) : (op2 == 0 && p2 == 1);
              u64 s2;
              u32 l2, w2;
              apply_edit_h(ham, s1, l1, p2, op2, s2, l2, w2);
              const bool use2 = K2 && l2 >= K2;
              const bool pr = v2 && (use2 || f.kf.nr);
              KfCopy c;
              c.base = use2 ? c2.base : c1.base;
              c.s = use2 ? c2.s : c1.s;
              const u32* a = kf_word(c, use2 ? s2 & mask2 : s2 & kmask, bit[op2]);
              addr[op2] = pr ? a : idle;
              valid |= (u32)v2 << op2;
              probe |= (u32)pr << op2;
            }
#pragma unroll
            for (u32 op2 = 0; op2 < 8; ++op2) word[op2] = *addr[op2];
            }
#pragma unroll
            for (u32 op2 = 0; op2 < 8; ++op2) {
              const u32 present = ((probe >> op2) & 1u) ? (word[op2] >> bit[op2]) & 1u : 1u;
              mask8 |= (((valid >> op2) & 1u) & present) << op2;
            }
            nprobe += (u32)__popc(probe);
            surv |= (u64)mask8 << (8 * op1);
          }
        }
        }
      }
      // number the survivors: inclusive scan of the lanes' counts inside the wavefront, wavefront totals through LDS
      const u32 mine = (u32)__popcll(surv);
      const u32 incl = wave_incl_scan32(mine);
      if (lane == 63) q_wave[threadIdx.x >> 6] = incl;
      __syncthreads();
      u32 before = 0;
      for (u32 k = 0; k < (threadIdx.x >> 6); ++k) before += q_wave[k];
      const u32 qn = (expb & 1u) ? 0u : q_wave[0] + q_wave[1] + q_wave[2] + q_wave[3];
      const bool listed = qn <= QS_CAP;  // (uniform)
      if (listed) {
        u32 at = before + incl - mine;
        for (unsigned long long mk = surv; mk; mk &= mk - 1) q_surv[at++] = (u16)(threadIdx.x | (((u32)__ffsll((long long)mk) - 1u) << 8));
      } else {
        q_mask[threadIdx.x] = surv;
        q_ex[threadIdx.x] = before + incl - mine;
      }
      __syncthreads();
      // Dense phase in two stages (r05).  Five of six survivors of the end-window probes die on the first look — the head window
      // (the first K2 characters must occur as well) — and everything after it (table entry, preceding-characters line or Occ steps,
      // the leaf) used to run with every sixth lane alive.  Stage 1 rebuilds a survivor's string and asks the head window only; the ones
      // that pass are queued in LDS (two bytes each) and stage 2 runs over the queue with whole wavefronts, once 256 are waiting or the
      // pass ends.  (Counts per wavefront through q_wc instead of an atomic: the decision to run stage 2 must be the same in every
      // wavefront.)
      u32 have = 0;  // queued survivors (uniform)
      for (u32 e0 = 0, round = 0; e0 < qn; e0 += 256, ++round) {
        const u32 e = e0 + threadIdx.x;
        bool pass = false;
        u32 ent = 0;
        if (e < qn) {
          u32 L = 0, bitno;
          if (listed) {
            const u32 sv = q_surv[e];
            L = sv & 255u;
            bitno = sv >> 8;
          } else {
            // the lane that holds survivor e: the last one whose exclusive prefix is <= e; then its (e - prefix)-th set bit
#pragma unroll
            for (u32 step = 128; step > 0; step >>= 1)
              if (q_ex[L + step] <= e) L += step;
            unsigned long long mk = q_mask[L];
            for (u32 r = e - q_ex[L]; r > 0; --r) mk &= mk - 1;
            bitno = (u32)__ffsll((long long)mk) - 1u;
          }
          const u32 pp = q_pair[L], p1 = pp & 255u, p2 = pp >> 8;
          u32 l1, l2;
          const u64 s1 = edit_string(ham, qpk, m, p1, bitno >> 3, l1);
          const u64 s2 = edit_string(ham, s1, l1, p2, bitno & 7u, l2);
          nprobe += (K2 && l2 > K2);
          pass = (expb & 2u) || head_window_occurs(f, s2, l2, l1 - p2);
          ent = L | (bitno << 8);
        }
        const unsigned long long pm = __ballot(pass);
        if (lane == 0) q_wc[round & 1u][threadIdx.x >> 6] = (u32)__popcll(pm);
        __syncthreads();
        u32 at = have, all = have;
        for (u32 k = 0; k < 4; ++k) {
          const u32 c = q_wc[round & 1u][k];
          if (k < (threadIdx.x >> 6)) at += c;
          all += c;
        }
        if (pass) q_cand[at + (u32)__popcll(pm & ((1ULL << lane) - 1))] = (u16)ent;  // (have < 256 here: at most 512 queued)
        have = all;
        if (have < 256 && e0 + 256 < qn) continue;
        __syncthreads();
        for (u32 t0 = 0; t0 < have; t0 += 256) {
          if (t0 + (threadIdx.x & ~63u) >= have) break;  // wavefront without work
          const u32 t = t0 + threadIdx.x;
          bool leaf = false;
          u32 lo = 0, hi = 0, w1 = 0, w2 = 0, fword = 0, len2 = 0, pre_first = 0xFFFFFFFFu;
          u64 key2 = 0;
          if (t < have) {
            const u32 ce = q_cand[t], L = ce & 255u, bitno = ce >> 8;
            const u32 pp = q_pair[L], p1 = pp & 255u, p2 = pp >> 8;
            u32 l1, l2;
            u64 s1, s2;
            apply_edit_h(ham, qpk, m, p1, bitno >> 3, s1, l1, w1);
            apply_edit_h(ham, s1, l1, p2, bitno & 7u, s2, l2, w2);
            key2 = s2;
            len2 = l2;
            const KtabEntry iv = ktab_entry(f, s2 & kmask);
            ++nlook;
            lo = iv.lo;
            hi = iv.hi;
            pre_first = iv.pre_first;
            u64 rs = s2 >> (2 * K);
            u32 nr = l2 - K;
            // filtered interval (FmView::pre5, see k_search1s): a narrow table interval is settled with one line of the preceding-characters
            // array instead of l2 - K dependent Occ lines; the leaf then carries (interval of its last K characters, mask, l2 - K) in ops[2]
            if (filt_ok && (!ham || to_lds) && f.pre5 && nr >= 1 && nr <= 5 && lo < hi && hi - lo <= 16) {  // (Hamming leaves with fewer than two operations carry no filter word)
              const u32 w = hi - lo;
              u32 want = 0, fm = 0;
              for (u32 k2 = 0; k2 < nr; ++k2) want |= ((u32)(rs >> (2 * k2)) & 3u) << (3 * k2);
              const u32 wmask = (1u << (3 * nr)) - 1u;
              if (w == 1 && pre_first != 0xFFFFFFFFu) fm = (u32)((pre_first & wmask) == want);  // (r05: from the table entry, see k_search1s)
              else {
                u32 ent16[16];
#pragma unroll
                for (u32 j = 0; j < 16; ++j) ent16[j] = j < w ? (u32)f.pre5[(u64)lo + j] : 0xFFFFu;
#pragma unroll
                for (u32 j = 0; j < 16; ++j) fm |= (u32)((ent16[j] & wmask) == want && j < w) << j;
                ++nlook;
              }
              fword = fm | (nr << 16) | (1u << 31);
              nr = 0;
              if (!fm) lo = hi = 0;
            }
            while (nr && lo < hi) {
              bs_extend_code_narrow(f, lo, hi, (u32)rs & 3u);
              rs >>= 2;
              --nr;
              ++steps;
            }
            leaf = lo < hi;
          }
          if (SEL && to_lds) {  // the string itself (2 bits per character: s2 of l2 <= 32 characters), its interval, its filter word
            if (leaf) {
              const u32 la = atomicAdd(&l_n, 1u);
              if (la < FUSED2_LCAP) {
                l_key[la] = key2;
                l_lo[la] = lo;
                l_hi[la] = hi;
                l_fw[la] = fword;
                l_meta[la] = (u16)len2;
                l_rank[la] = 0u;
              }
            }
            continue;
          }
          const unsigned long long lm = __ballot(leaf);
          u32 lbase = 0;
          if (lane == 0 && lm) lbase = atomicAdd(&o.ctr->leaf_cnt[shard], (u32)__popcll(lm));
          lbase = __shfl(lbase, 0);
          if (leaf) {
            const u32 slot = atomicAdd(&g_slots, 1u);
            const u32 la = lbase + (u32)__popcll(lm & ((1ULL << lane) - 1));
            if (la < o.shard_cap) {
              Leaf* lf = o.leaves + (u64)shard * o.shard_cap + la;
              lf->qs = gid;
              lf->slot = slot;
              lf->lo = lo;
              lf->hi = hi;
              // (Hamming: "none" operations are not recorded; the filter word is only read from two-operation leaves, k_group_pack)
              const u32 nops = (u32)(w1 != OPW_NONE) + (u32)(w2 != OPW_NONE);
              lf->nops = nops;
              lf->ops[0] = w1 != OPW_NONE ? w1 : (w2 != OPW_NONE ? w2 : 0u);
              lf->ops[1] = (w1 != OPW_NONE && w2 != OPW_NONE) ? w2 : 0u;
              lf->ops[2] = nops == 2 ? fword : 0u;  // 0, or the filtered form: mask | characters in front << 16 | 1 << 31 (k_group_pack hands it on)
#pragma unroll
              for (int k = 3; k < (int)DMAX; ++k) lf->ops[k] = 0u;
            }
          }
        }
        have = 0;
        __syncthreads();  // (the queue is free again)
      }
      if (count) {  // (the wavefront is whole here)
        steps = wave_sum32(steps);
        nlook = wave_sum32(nlook);
        nprobe = wave_sum32(nprobe);
        if (lane == 0) {
          if (steps) atomicAdd(&s_cnt[0], steps);
          if (nlook) atomicAdd(&s_cnt[1], nlook);
          if (nprobe) atomicAdd(&s_cnt[2], nprobe);
        }
      }
      __syncthreads();  // the masks and prefixes of this pass are not needed any more
    }
  };
  for (u32 once = 0; once < 1u; ++once) {  // (`continue` leaves the group)
    const u32 gid = blockIdx.x;
    const uint4 raw = *reinterpret_cast<const uint4*>(b.ginfo + gid);
    const u32 m = raw.z, d_win = raw.w;
    if (!m || !(d_win & 1024u)) continue;  // uniform for the workgroup
    const u64 qpk = (u64)raw.y << 32 | raw.x;
    __syncthreads();
    if (threadIdx.x == 0) {
      g_slots = 0;
      l_n = 0;
      s_alive = 0;
    }
    __syncthreads();
    // (one call site: the search is 27 KB of code)
    bool over = false;
#pragma unroll 1
    for (u32 pass = 0; pass < (SEL ? 2u : 1u); ++pass) {
      search(gid, m, qpk, SEL && pass == 0, pass == 0);  // (a second pass is not counted again)
      if (!SEL) break;
      if (pass == 0) {  // (search ends with a barrier: the list is complete)
        if (l_n <= lcap2) break;
        over = true;  // (lcap2 <= FUSED2_LCAP) the generic select kernels take this group: once more, leaves to HBM
      }
    }
    if (!SEL || over) {
      if (threadIdx.x == 0 && g_slots) atomicAdd(o.grp_cnt + gid, g_slots);
      continue;
    }
    const u32 nl = l_n;
    // Select without the n^2 pair loop (r04: one lane per pair took as long as the generic select kernels, 2.4 ms per batch — 48
    // occurring strings per group, half of them repeats of another edit path's string):
    //  (1) the list's DISTINCT strings through a hash table of list indices in LDS (compare-and-swap, linear probing; a string that
    //      finds itself there is a duplicate and dies);
    //  (2) a distinct string dies when one of its proper substrings of >= m - 2 characters (the shortest a group can hold) is in the
    //      table: <= 14 look-ups (2 + 3 + 4 + 5 for a string of m + 2 characters), shared by G lanes per string;
    //  (3) the survivors are numbered (l_ord) and ranked among themselves only.
    for (u32 t = threadIdx.x; t < FUSED2_HCAP; t += 256) l_hash[t] = 0xFFFFFFFFu;
    __syncthreads();
    auto hash_of = [](u64 key, u32 len) { return (u32)(((key + len) * 0x9E3779B97F4A7C15ULL) >> 40) & (FUSED2_HCAP - 1u); };
    for (u32 i = threadIdx.x; i < nl; i += 256) {
      const u64 key = l_key[i];
      const u32 len = l_meta[i] & 63u;
      u32 sl = hash_of(key, len);
      for (;;) {
        const u32 old = atomicCAS(&l_hash[sl], 0xFFFFFFFFu, i);
        if (old == 0xFFFFFFFFu) break;  // the string's representative
        if (l_key[old] == key && (l_meta[old] & 63u) == len) {
          l_rank[i] = 0xFFFFFFFFu;  // a repeat
          break;
        }
        sl = (sl + 1u) & (FUSED2_HCAP - 1u);
      }
    }
    __syncthreads();
    const u32 G = (nl && nl < 256u) ? 256u / nl : 1u;
    const u32 ti = threadIdx.x / G, tk = threadIdx.x - ti * G, istep = 256u / G;
    for (u32 i = ti; i < nl && !ham; i += istep) {  // (Hamming mode keeps the whole ball: no string is another's proper substring)
      if (l_rank[i] == 0xFFFFFFFFu) continue;
      const u32 alen = l_meta[i] & 63u, span = alen - (m - 2u);  // 0..4 characters above the shortest
      const u64 a = l_key[i];
      const u32 combos = span * (span + 3u) / 2u;  // sum over k = 1..span of (k + 1) placements
      bool dead = false;
      for (u32 c = tk; c < combos && !dead; c += G) {
        // c -> (k characters shorter, placement sh = 0..k): the blocks start at 0, 2, 5, 9
        const u32 k = c < 2u ? 1u : c < 5u ? 2u : c < 9u ? 3u : 4u;
        const u32 sh = c - (k * (k + 1u) / 2u - 1u);
        const u32 L = alen - k;
        const u64 sub = (a >> (2 * sh)) & (L >= 32 ? ~0ULL : ((1ULL << (2 * L)) - 1));
        u32 sl = hash_of(sub, L);
        for (;;) {
          const u32 v = l_hash[sl];
          if (v == 0xFFFFFFFFu) break;
          if (l_key[v] == sub && (l_meta[v] & 63u) == L) {
            dead = true;
            break;
          }
          sl = (sl + 1u) & (FUSED2_HCAP - 1u);
        }
      }
      if (dead) l_rank[i] = 0xFFFFFFFFu;
    }
    __syncthreads();
    for (u32 i = threadIdx.x; i < nl; i += 256) {
      if (l_rank[i] == 0xFFFFFFFFu) continue;
      l_meta[i] = (u16)(l_meta[i] | 0x8000u);
      l_ord[atomicAdd(&s_alive, 1u)] = (u16)i;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      s_base = s_alive ? atomicAdd(&o.ctr->sel_cnt[shard], s_alive) : 0u;
      if (nl) atomicAdd(&o.ctr->fused_leaves[shard], (unsigned long long)nl);
    }
    // rank among the group's survivors in std::string order (A < C < G < T = code order; a proper prefix sorts first): partial counts
    // per lane, added up in LDS
    const u32 na = s_alive;
    const u32 G2 = (na && na < 256u) ? 256u / na : 1u;
    const u32 ri = threadIdx.x / G2, rk = threadIdx.x - ri * G2, rstep = 256u / G2;
    for (u32 jj = ri; jj < na; jj += rstep) {
      const u32 i = l_ord[jj];
      const u32 alen = l_meta[i] & 63u;
      const u64 ak = l_key[i] << (64 - 2 * alen);
      u32 r = 0;
      for (u32 y = rk; y < na; y += G2) {
        const u32 x = l_ord[y];
        if (x == i) continue;
        const u32 xlen = l_meta[x] & 63u;
        const u64 xk = l_key[x] << (64 - 2 * xlen);
        const bool first = (xk < ak) || (xk == ak && xlen < alen);
        r += first;
      }
      if (r) atomicAdd(&l_rank[i], r);
    }
    __syncthreads();
    const u32 wbase = s_base, total = s_alive;
    const bool room = wbase + total <= fs.cap;  // an overflowing slice repeats the batch (Summary::worst_sel) ...
    if (!room && threadIdx.x == 0) atomicOr(&o.ctr->overflow, 1u);  // ... and the kernels behind this one do nothing
    for (u32 jj = threadIdx.x; jj < na && room; jj += 256) {
      const u32 i = l_ord[jj];
      const u32 meta = l_meta[i];
      Sel sv;
      sv.lo = l_lo[i];
      sv.hi = l_hi[i];
      sv.len = sel_len_from(meta & 63u, l_fw[i]);
      sv.take = 0;  // (k_take's)
      sv.hbase = 0;
      sv.g = gid;
      fs.sel[(u64)shard * fs.cap + wbase + l_rank[i]] = sv;
    }
    if (threadIdx.x == 0) {
      fs.nsel[gid] = room ? total : 0u;
      fs.selbase[gid] = shard * fs.cap + wbase;
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    if (s_cnt[0]) atomicAdd(&o.ctr->steps[shard], (unsigned long long)s_cnt[0]);
    if (s_cnt[1]) atomicAdd(&o.ctr->lookups[shard], (unsigned long long)s_cnt[1]);
    if (s_cnt[2]) atomicAdd(&o.ctr->probes[shard], (unsigned long long)s_cnt[2]);
  }
}


// the groups the walker serves, and the strands of k_nkeep, listed (one atomic per wavefront and list); a list that does not fit
// raises bit 3 of Counters::overflow and the batch is repeated with room
__global__ void __launch_bounds__(256) k_walk_list(Batch b, u32* list, u32* count, u32* klist, u32* kcount, u32 cap, Counters* ctr) {
  const u64 g = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  bool need = false, keep = false;
  if (g < 2 * b.nq) {
    const GidInfo gi = b.ginfo[g];
    need = gi.m != 0 && !(gi.d_win & (512u | 1024u | 4096u));
    keep = gi.m != 0 && (gi.d_win & 8192u);
  }
  const u32 lane = threadIdx.x & 63;
  for (u32 which = 0; which < 2; ++which) {
    const bool mine = which ? keep : need;
    const unsigned long long mk = __ballot(mine);
    if (!mk) continue;
    const u32 leader = (u32)__ffsll((long long)mk) - 1u;
    u32 base = 0;
    if (lane == leader) base = atomicAdd(which ? kcount : count, (u32)__popcll(mk));
    base = __shfl(base, (int)leader);
    if (mine) {
      const u32 at = base + (u32)__popcll(mk & ((1ULL << lane) - 1));
      if (at < cap) (which ? klist : list)[at] = (u32)g;
      else atomicOr(&ctr->overflow, 8u);
    }
  }
}
// r06: wlist / wcount (may be null) — the groups the walker has work for, listed by k_walk_list: with the root split a launch over
// EVERY group is (groups x 145) lanes, 29 M for a batch of 100 000 queries of which a few thousand strands need the walker
// (queries with an N next to 95 % for the flat kernel); the lanes then cover the listed groups only.
struct WalkList {
  const u32* gid;
  const u32* count;
  u32 cap;
};
template <bool INDEL, int D>
__global__ void __launch_bounds__(256) k_search(FmView f, Batch b, SearchOut o, u32 items, WalkList wl) {
  // lane layout: the long-running "rest" lanes come first, packed densely (a rest lane among 63 short item lanes
  // would pin its whole wavefront); item lanes follow, (items-1) consecutive lanes per (query, strand)
  u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  const u64 ngrp = wl.gid ? (u64)(*wl.count < wl.cap ? *wl.count : wl.cap) : b.nq * 2;
  u64 gid;   // 2*query + strand (or its number in the list)
  u32 item;  // items-1 = the rest lane
  if (t < ngrp || items == 1) {
    gid = t;
    item = items - 1;
  } else {
    gid = (t - ngrp) / (items - 1);
    item = (u32)((t - ngrp) % (items - 1));
  }
  u64 steps = 0, lookups = 0, probes = 0;
  constexpr u32 NOPS = INDEL ? 9u : 4u;  // INDEL: D, S(A,C,G,T), I(A,C,G,T);  Hamming: S(A,C,G,T)
  bool active = gid < ngrp;
  if (wl.gid) gid = active ? wl.gid[gid] : 0;
  const u64 q = gid >> 1;
  const u32 strand = (u32)(gid & 1);
  GidInfo gi;
  gi.qpk = 0;
  gi.m = 0;
  gi.d_win = 0;
  if (active) gi = b.ginfo[gid];
  if (gi.m == 0 || (gi.d_win & (512u | 1024u | 4096u))) active = false;  // not searched, or taken by a flat kernel / k_nres + k_nkeep
  if (active) {
    const u8* seq = (strand ? b.rv : b.fw) + b.qoff[q];
    const u32 m = gi.m;
    u32 d = gi.d_win & 255;
    if (d > (u32)D) d = D;  // cannot happen: the host instantiates D >= the largest effective distance
    const bool use_win = f.K != 0 && m >= f.K + d && (gi.d_win & (256u | 2048u));  // (2048: N's, but none inside any window)
    const bool rest = item == items - 1;
    // lanes of a split launch: without the table (or without budget) only the rest lane works, as a full search
    if (!rest && (!use_win || d == 0)) active = false;
    // r05: a query with characters outside A,C,G,T (searched as 'N', util.h:208-219).  A neighbourhood string that keeps an N occurs only
    // where the text has a run of N's of exactly its N-block's length when that block has other characters on both sides inside the
    // string.  If the query has more than d other characters in front of its first and behind its last N (an outer one survives d
    // edits on each side, so every N-block of every string is flanked) and fewer N's than the text's shortest run (FmView::nrun_min),
    // NO string that keeps an N occurs: every N has to be substituted or deleted, and the walk is cut down to that — a node is left
    // as soon as the N's still ahead outnumber the edits left.  (r05 measured 5 % of such queries costing a batch 20 x: one lane per
    // strand walked ~150 strings in interval mode, ~1 500 dependent index reads, to find that none of them occurs.)
    u64 nmask = 0;  // bit i: query character i is an N
    bool nprune = false;
    if (active && !(gi.d_win & 256) && m <= 64 && f.nrun_min) {
      for (u32 i = 0; i < m; ++i) nmask |= (u64)(seq[i] >= 4) << i;
      const u32 nN = (u32)__popcll(nmask);
      if (nN) {
        const u32 first = (u32)__builtin_ctzll(nmask), last = 63u - (u32)__builtin_clzll(nmask);
        nprune = first >= d + 1 && (m - 1 - last) >= d + 1 && nN < f.nrun_min;
        if (nprune && nN > d) active = false;  // more N's than edits: every string keeps one
      }
    }
    if (active) {
      const u64 qpk = gi.qpk;
      FrameStack<D> S;
      OpStack<D> ops;
      u32 L = 0;
      bool single = false;  // an item lane: exactly one root operation
      u32 single_op = 0;
      {
        Frame r;
        r.pos = m;
        r.lo = 0;
        r.hi = use_win ? 0u : (u32)f.n;
        r.st = use_win ? ST_WIN : 0u;
        if (use_win && items > 1) {
          if (rest) {
            // the unedited window in one table read; first edits left of the window follow in interval mode
            if (!frame_finish_window(f, r, seq, m, qpk, lookups, probes)) active = false;
          } else {
            const u32 j = item / NOPS;  // characters right of the edit
            single_op = item % NOPS;
            single = true;
            u64 code = 0;
            if (m <= 32) code = qpk & (j >= 32 ? ~0ULL : ((1ULL << (2 * j)) - 1));
            else
              for (u32 k = 0; k < j; ++k) code |= (u64)seq[m - 1 - k] << (2 * k);
            r.pos = m - j;
            r.lo = (u32)code;
            r.hi = (u32)(code >> 32);
            r.st = ST_WIN | (j << 4) | single_op;
          }
        }
        S.set(0, r);
      }
      while (active) {
        Frame F = S.get(L);
        const u32 budget = d - L;
        if (F.pos == 0) {
          // a complete neighbourhood string whose interval is non-empty (strings are never shorter than K in window mode)
          if (!INDEL || budget == 0) {
            const u32 shard = blockIdx.x & (NSHARD - 1);
            u32 at = atomicAdd(&o.ctr->leaf_cnt[shard], 1u);
            u32 slot = atomicAdd(o.grp_cnt + gid, 1u);
            if (at < o.shard_cap) {
              Leaf* lf = o.leaves + (u64)shard * o.shard_cap + at;
              lf->qs = (u32)gid;
              lf->slot = slot;
              lf->lo = F.lo;
              lf->hi = F.hi;
              lf->nops = L;
#pragma unroll
              for (int k = 0; k < (int)DMAX; ++k) lf->ops[k] = (k < D && (u32)k < L) ? ops.v[k < D ? k : 0] : 0u;
            }
          }
          if (L == 0) break;
          --L;
          continue;
        }
        const u32 pos = F.pos;
        // the query character: from the packed copy when there is one (no memory access on the critical path)
        const u32 here = (m <= 32 && (gi.d_win & 256)) ? (u32)(qpk >> (2 * (m - pos))) & 3u : (u32)seq[pos - 1];
        const u32 op = F.st & 15;
        if (single && L == 0 && op != single_op) break;  // the item's one operation has been explored
        if (budget > 0 && op < NOPS) {
          F.st += 1;  // next operation of this node
          S.set(L, F);
          u32 kind, c;
          if (INDEL) {
            kind = op == 0 ? OP_D : (op <= 4 ? OP_S : OP_I);
            c = op == 0 ? 0u : (op - 1) & 3;
          } else {
            kind = OP_S;
            c = op;
          }
          if (kind == OP_S && c == here) continue;           // a substitution changes the character (neighbors.h:63)
          if (kind == OP_I && L == 0 && pos == m) continue;   // nothing may be inserted after the last character (neighbors.h:51)
          // N's still ahead (left of what this operation consumes) against the edits left behind it
          if (nprune && (u32)__popcll(nmask & ((1ULL << (kind == OP_I ? pos : pos - 1)) - 1)) > budget - 1) continue;
          Frame ch = F;
          ch.st &= ~15u;
          if (kind != OP_D && !frame_emit(f, ch, c, steps, lookups, probes)) continue;
          ch.pos = kind == OP_I ? pos : pos - 1;
          if (budget == 1 && (ch.st & ST_WIN) && !frame_finish_window(f, ch, seq, m, qpk, lookups, probes)) continue;
          ops.set(L, (pos << 4) | (kind << 2) | c);
          ++L;
          S.set(L, ch);
          continue;
        }
        // keep the query character(s)
        bool alive;
        if (budget == 0 && (F.st & ST_WIN)) alive = frame_finish_window(f, F, seq, m, qpk, lookups, probes);  // only a d = 0 root
        else {
          F.st &= ~15u;
          if (nprune && (here >= 4 || (u32)__popcll(nmask & ((1ULL << (pos - 1)) - 1)) > budget)) alive = false;  // a kept N, or more N's ahead than edits
          else if (here < 4) alive = frame_emit(f, F, here, steps, lookups, probes);
          else {  // an N in the query (never in window mode): through the wavelet tree like sdsl
            bs_extend_sym(f, F.lo, F.hi, 'N', here);
            ++steps;
            alive = F.lo < F.hi;
          }
          F.pos = pos - 1;
        }
        if (!alive) {
          if (L == 0) break;
          --L;
          continue;
        }
        S.set(L, F);
      }
    }
  }
  wave_add(&o.ctr->steps[blockIdx.x & (NSHARD - 1)], steps);
  wave_add(&o.ctr->lookups[blockIdx.x & (NSHARD - 1)], lookups);
  wave_add(&o.ctr->probes[blockIdx.x & (NSHARD - 1)], probes);
}

// r06: the strands k_prepare marked with bit 12 — as many N's as edits, every N to be substituted (A, C, G, T) or deleted: one lane
// per resolution (per = 5 or 25 strings per strand in edit mode, 4 or 16 in Hamming mode), searched like an explicit pattern (long
// presence filter on the last and the first K2 characters, table entry, extensions); a string that occurs becomes an ordinary leaf
// of its group whose operations are the resolutions (util.h:208-219 turns non-DNA into N; neighbors.h:57-78 substitutes / deletes
// it like any other character).  Equal strings of two resolutions (neighbouring N's) are dropped by the select stage.
template <bool INDEL>
__global__ void __launch_bounds__(256) k_nres(FmView f, Batch b, SearchOut o, u32 per) {
  const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  const u64 gid = t / per;
  const u32 r = (u32)(t - gid * per);
  constexpr u32 R1 = INDEL ? 5u : 4u;
  u64 steps = 0, lookups = 0, probes = 0;
  GidInfo gi;
  gi.qpk = 0;
  gi.m = 0;
  gi.d_win = 0;
  if (gid < 2 * b.nq) gi = b.ginfo[gid];
  const u32 d = gi.d_win & 255u;
  if (gi.m && (gi.d_win & 4096u) && (d == 2 || r < R1)) {
    const u32 m = gi.m;
    const uint4 pq = b.gpeq[gid];
    const u32 nm = ~(pq.x | pq.y | pq.z | pq.w) & (m == 32 ? ~0u : ((1u << m) - 1u));
    const u32 i0 = (u32)__builtin_ctz(nm), i1 = 31u - (u32)__builtin_clz(nm);  // (d == 1: the same position)
    const u32 c0 = r % R1, c1 = r / R1;  // resolution of the left / right N: 0-3 a base, 4 deleted
    u64 s = 0;
    u32 len = 0;
    for (u32 i = 0; i < m; ++i) {
      u32 c = (u32)(gi.qpk >> (2 * (m - 1 - i))) & 3u;
      if (i == i0) c = c0;
      else if (d == 2 && i == i1) c = c1;
      if (c < 4) {
        s = (s << 2) | c;
        ++len;
      }
    }
    auto at = [&](u32 j) -> u32 { return (u32)(s >> (2 * (len - 1 - j))) & 3u; };
    u32 lo = 0, hi = (u32)f.n, k = len;
    const u32 K = f.K, K2 = f.kf2.nr ? f.kf2.k : 0u, W = K2 > K ? K2 : K;
    if (K && len >= W && W <= 32) {
      const u64 tail = W >= 32 ? s : (s & ((1ULL << (2 * W)) - 1));
      bool alive = true;
      if (K2) {
        ++probes;
        alive = kf_present(f.kf2, tail & ((1ULL << (2 * K2)) - 1), 0u);
        if (alive && len > K2) {
          ++probes;
          alive = kf_present(f.kf2, (s >> (2 * (len - K2))) & ((1ULL << (2 * K2)) - 1), K2 - 1);
        }
      } else if (f.kf.nr) {
        ++probes;
        alive = kf_present(f.kf, tail & ((1ULL << (2 * K)) - 1), 0u);
      }
      if (alive) {
        const KtabEntry iv = ktab_entry(f, tail & ((1ULL << (2 * K)) - 1));
        ++lookups;
        lo = iv.lo;
        hi = iv.hi;
      } else lo = hi = 0;
      k = len - K;
    }
    for (; k > 0 && lo < hi; --k) {
      bs_extend_code(f, lo, hi, at(k - 1));
      ++steps;
    }
    if (lo < hi) {
      const u32 shard = blockIdx.x & (NSHARD - 1);
      const u32 a = atomicAdd(&o.ctr->leaf_cnt[shard], 1u);
      const u32 slot = atomicAdd(o.grp_cnt + gid, 1u);
      if (a < o.shard_cap) {
        Leaf* lf = o.leaves + (u64)shard * o.shard_cap + a;
        lf->qs = (u32)gid;
        lf->slot = slot;
        lf->lo = lo;
        lf->hi = hi;
        lf->nops = d;
#pragma unroll
        for (int x = 0; x < (int)DMAX; ++x) lf->ops[x] = 0u;
        // operations right to left (Leaf): the right N first.  position field: index + 1 for a substitution / deletion (LeafReader)
        const u32 opl = ((i0 + 1u) << 4) | ((c0 == 4 ? (u32)OP_D : (u32)OP_S) << 2) | (c0 & 3u);
        const u32 opr = ((i1 + 1u) << 4) | ((c1 == 4 ? (u32)OP_D : (u32)OP_S) << 2) | (c1 & 3u);
        if (d == 2) {
          lf->ops[0] = opr;
          lf->ops[1] = opl;
        } else lf->ops[0] = opl;
      }
    }
  }
  wave_add(&o.ctr->steps[blockIdx.x & (NSHARD - 1)], steps);
  wave_add(&o.ctr->lookups[blockIdx.x & (NSHARD - 1)], lookups);
  wave_add(&o.ctr->probes[blockIdx.x & (NSHARD - 1)], probes);
}

// r06: the strings of a bit-13 strand that KEEP its N (one N, one edit, the N within a character of an end): the query with one
// edit elsewhere — or none in Hamming mode — such that the N is the string's first or last character; one lane per candidate edit
// (per = 1 + 9 m in edit mode, 1 + 3 m in Hamming mode), over the strands k_walk_list listed.  N last: the search starts from the
// N's own interval and dies within a few steps almost everywhere; N first: the rest is searched like any N-free string and only
// what occurs is extended by the N (through the wavelet tree, like sdsl).  An occurring string becomes a leaf whose operation is
// the edit (neighbors.h:57-78; the N itself stays in the query the leaf reader applies it to).
template <bool INDEL>
__global__ void __launch_bounds__(256) k_nkeep(FmView f, Batch b, SearchOut o, WalkList wl, u32 per) {
  const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  const u64 slot = t / per;
  const u32 e = (u32)(t - slot * per);  // 0: no edit; 1 + 9 i + k (edit mode): k = 0 delete q[i], 1-4 substitute, 5-8 insert in front of q[i]
  const u32 nl = *wl.count < wl.cap ? *wl.count : wl.cap;
  u64 steps = 0, lookups = 0, probes = 0;
  if (slot < nl) {
    const u32 gid = wl.gid[slot];
    const GidInfo gi = b.ginfo[gid];
    const u32 m = gi.m;
    const uint4 pq = b.gpeq[gid];
    const u32 nm = ~(pq.x | pq.y | pq.z | pq.w) & (m == 32 ? ~0u : ((1u << m) - 1u));
    const u32 iN = (u32)__builtin_ctz(nm);
    constexpr u32 PER_POS = INDEL ? 9u : 3u;  // (r06 fix: nine operations per position in edit mode — with 8 the insertion of T was never enumerated; tools/fuzz_n.py)
    bool ok = m != 0 && (gi.d_win & 8192u) != 0;
    u32 kind = 3u /* none */, i = 0, c = 0;
    if (e) {
      i = (e - 1) / PER_POS;
      const u32 k = (e - 1) % PER_POS;
      if (i >= m) ok = false;
      const u32 qi = ok ? (u32)(gi.qpk >> (2 * (m - 1 - i))) & 3u : 0u;
      if (INDEL) {
        if (k == 0) kind = OP_D;
        else if (k <= 4) {
          kind = OP_S;
          c = k - 1;
        } else {
          kind = OP_I;
          c = k - 5;
        }
      } else {  // the three other bases
        kind = OP_S;
        c = (qi + 1 + k) & 3u;
      }
      if (i == iN && kind != OP_I) ok = false;          // the N substituted or deleted: k_nres's strings
      if (kind == OP_S && i != iN && c == qi) ok = false;  // a substitution changes the character (neighbors.h:63)
    } else if (INDEL) ok = false;  // edit mode: the unedited string contains the string without its N, which then occurs too
    // the string, left to right; where the N ends up
    u64 s = 0;  // the characters other than the N, 2 bits each, first character on top
    u32 len = 0, posN = 0;
    if (ok) {
      for (u32 j = 0; j < m; ++j) {
        if (kind == OP_I && j == i) {
          s = (s << 2) | c;
          ++len;
        }
        if (j == iN) {
          posN = len;
          ++len;
          s <<= 2;  // (a placeholder: never read)
          continue;
        }
        if (kind == OP_D && j == i) continue;
        s = (s << 2) | ((kind == OP_S && j == i) ? c : (u32)(gi.qpk >> (2 * (m - 1 - j))) & 3u);
        ++len;
      }
      if (posN != 0 && posN != len - 1) ok = false;  // a single N between two bases: not in this text
      if (len < 2) ok = false;
    }
    if (ok) {
      auto at = [&](u32 j) -> u32 { return (u32)(s >> (2 * (len - 1 - j))) & 3u; };
      u32 lo = 0, hi = (u32)f.n;
      if (posN == len - 1) {  // N last: from the N's interval, then the rest right to left
        bs_extend_sym(f, lo, hi, 'N', 4u);
        ++steps;
        for (u32 k = len - 1; k > 0 && lo < hi; --k) {
          bs_extend_code(f, lo, hi, at(k - 1));
          ++steps;
        }
      } else {  // N first: the N-free rest s[1 .. len) like an explicit pattern, then the N
        const u32 L2 = len - 1;
        const u64 rest = L2 >= 32 ? s : (s & ((1ULL << (2 * L2)) - 1));
        const u32 K = f.K, K2 = f.kf2.nr ? f.kf2.k : 0u, W = K2 > K ? K2 : K;
        u32 k = L2;
        if (K && L2 >= W && W <= 32) {
          bool alive = true;
          const u64 tail = rest & (W >= 32 ? ~0ULL : ((1ULL << (2 * W)) - 1));
          if (K2) {
            ++probes;
            alive = kf_present(f.kf2, tail & ((1ULL << (2 * K2)) - 1), 0u);
            if (alive && L2 > K2) {
              ++probes;
              alive = kf_present(f.kf2, (rest >> (2 * (L2 - K2))) & ((1ULL << (2 * K2)) - 1), K2 - 1);
            }
          } else if (f.kf.nr) {
            ++probes;
            alive = kf_present(f.kf, tail & ((1ULL << (2 * K)) - 1), 0u);
          }
          if (alive) {
            const KtabEntry iv = ktab_entry(f, tail & ((1ULL << (2 * K)) - 1));
            ++lookups;
            lo = iv.lo;
            hi = iv.hi;
          } else lo = hi = 0;
          k = L2 - K;
        }
        for (; k > 0 && lo < hi; --k) {  // characters 1 .. of the string: at(j) with j >= 1
          bs_extend_code(f, lo, hi, at(k));
          ++steps;
        }
        if (lo < hi) {
          bs_extend_sym(f, lo, hi, 'N', 4u);
          ++steps;
        }
      }
      if (lo < hi) {
        const u32 shard = blockIdx.x & (NSHARD - 1);
        const u32 a = atomicAdd(&o.ctr->leaf_cnt[shard], 1u);
        const u32 sl = atomicAdd(o.grp_cnt + gid, 1u);
        if (a < o.shard_cap) {
          Leaf* lf = o.leaves + (u64)shard * o.shard_cap + a;
          lf->qs = gid;
          lf->slot = sl;
          lf->lo = lo;
          lf->hi = hi;
          lf->nops = e ? 1u : 0u;
#pragma unroll
          for (int x = 0; x < (int)DMAX; ++x) lf->ops[x] = 0u;
          if (e) lf->ops[0] = ((kind == OP_I ? i : i + 1u) << 4) | (kind << 2) | (kind == OP_D ? 0u : c);  // (LeafReader: index + 1 for S / D, index for I)
        }
      }
    }
  }
  wave_add(&o.ctr->steps[blockIdx.x & (NSHARD - 1)], steps);
  wave_add(&o.ctr->lookups[blockIdx.x & (NSHARD - 1)], lookups);
  wave_add(&o.ctr->probes[blockIdx.x & (NSHARD - 1)], probes);
}

// Explicit patterns (the host-enumerated capped neighbourhoods): one lane per string, plain backward search right to
// left (sdsl::count, hunter.h:353); occurring strings become leaves of their (query, strand) group like k_search's.
__global__ void __launch_bounds__(256) k_explicit(FmView f, Batch b, SearchOut o) {
  const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  u64 steps = 0, lookups = 0, probes = 0;
  if (i < b.nxs) {
    const u64 s = b.xs_off[i], e = b.xs_off[i + 1];
    u32 lo = 0, hi = (u32)f.n;
    u64 k = e;
    // r04: like every other search kernel, a pattern first asks the long presence filter about its last K2 characters (96 of 100
    // random 18-mers of a 3.1 Gb genome end there, for one line instead of ~13 interval extensions from the full range), then its
    // first K2 characters, then takes the interval of its last K characters from the table.  r03 searched the 40 M patterns of
    // 2 000 capped 25-mers character by character: ~1 G random Occ lines, most of the 51 ms step.
    const u32 len = (u32)(e - s), K = f.K, K2 = f.kf2.nr ? f.kf2.k : 0u, W = K2 > K ? K2 : K;
    if (K && len >= W && W <= 32) {
      u64 code = 0;  // the last W characters, last character in the lowest bits
      bool plain = true;
      for (u32 t = 0; t < W; ++t) {
        const u32 c = b.xs_bytes[e - 1 - t];
        plain = plain && c < 4;
        code |= (u64)(c & 3u) << (2 * t);
      }
      if (plain) {
        bool alive = true;
        if (K2) {
          ++probes;
          alive = kf_present(f.kf2, code & ((1ULL << (2 * K2)) - 1), 0u);
          if (alive && len > K2) {  // the first K2 characters as well
            u64 head = 0;
            bool hp = true;
            for (u32 t = 0; t < K2; ++t) {
              const u32 c = b.xs_bytes[s + K2 - 1 - t];
              hp = hp && c < 4;
              head |= (u64)(c & 3u) << (2 * t);
            }
            if (hp) {
              ++probes;
              alive = kf_present(f.kf2, head, K2 - 1);
            }
          }
        } else if (f.kf.nr) {
          ++probes;
          alive = kf_present(f.kf, code & ((1ULL << (2 * K)) - 1), 0u);
        }
        if (alive) {
          const KtabEntry iv = ktab_entry(f, code & ((1ULL << (2 * K)) - 1));
          ++lookups;
          lo = iv.lo;
          hi = iv.hi;
        } else lo = hi = 0;
        k = e - K;
      }
    }
    for (; k > s && lo < hi; --k) {
      const u32 code = b.xs_bytes[k - 1];
      bs_extend_sym(f, lo, hi, 'N', code);
      ++steps;
    }
    if (lo < hi) {
      const u32 gid = b.xs_gid[i];
      const u32 shard = blockIdx.x & (NSHARD - 1);
      const u32 at = atomicAdd(&o.ctr->leaf_cnt[shard], 1u);
      const u32 slot = atomicAdd(o.grp_cnt + gid, 1u);
      if (at < o.shard_cap) {
        Leaf* lf = o.leaves + (u64)shard * o.shard_cap + at;
        lf->qs = gid;
        lf->slot = slot;
        lf->lo = lo;
        lf->hi = hi;
        lf->nops = LEAF_EXPLICIT;
        lf->ops[0] = (u32)i;
#pragma unroll
        for (int k = 1; k < (int)DMAX; ++k) lf->ops[k] = 0u;
      }
    }
  }
  wave_add(&o.ctr->steps[blockIdx.x & (NSHARD - 1)], steps);
  wave_add(&o.ctr->lookups[blockIdx.x & (NSHARD - 1)], lookups);
  wave_add(&o.ctr->probes[blockIdx.x & (NSHARD - 1)], probes);
}

}  // namespace dg
