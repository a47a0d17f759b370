// dg_hunt / dg_hunt_device: the per-query loop of `dicey hunt` (reference src/hunter.h:291-437) for a whole batch,
// entirely on the GPU.  Five kernels on the index stream:
//
//   k_prepare  upper-case, non-ACGT -> N, reverse complement, distance clamp      hunter.h:299-315, util.h:110,208
//   k_search   neighbourhood enumeration fused with backward search: one lane per (query, strand) walks the trie of
//              edit paths right-to-left, carrying the SA interval; dead branches stop at the first empty interval.
//              Emits every OCCURRING string of the <=d-edit language                neighbors.h:47-83 + hunter.h:353
//   k_select   per query: duplicates out, substring-minimal strings only (= neighbors.h:29-45 restricted to strings
//              that occur), std::set order, max_locations gating                    hunter.h:349-357
//   k_locate   the `take` smallest suffix-array values of each kept interval        hunter.h:355-357
//   k_verify   chromosome lookup, context window, '\n' trimming, Needleman-Wunsch with the reference's tie rules,
//              lead/trail gap stripping -> DnaHit records in reference push order   hunter.h:358-429, needle.h:59-138
//
// Why enumerating only occurring strings is exact: the reference searches the substring-minimal subset M of the
// language L.  A non-minimal string that occurs implies its minimal substring occurs too, hence
// M ∩ Occ = minimal elements of (L ∩ Occ); and every string reachable with fewer than d edits is dominated by the
// same string minus its first character, so only cost-exactly-d leaves can be minimal (DESIGN.md §"neighbourhood").
// This holds while the maxNeighborhood cap cannot fire.  Queries for which it could are enumerated on the host in the
// reference's own generation order (nbhd_host.hpp); when the cap stays silent the kernel's set is the reference's, and
// when it fires the capped set enters the pipeline as explicit patterns (k_explicit) next to k_search's leaves.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <mutex>
#include <set>
#include <string>
#include <thread>
#include <type_traits>

#include "hunt_internal.hpp"
#include "iupac.hpp"
#include "nbhd_host.hpp"
#include "band_bits.hpp"
#include "hunt_cap.hpp"

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
  u32 len;    // string length
  u32 take;   // how many of its occurrences become hits
  u32 hbase;  // first hit slot, relative to the query's first hit
  u32 g;      // 2*query + strand (set for the strings of the flat region, where no leaf record names the group)
};

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
  if (m > b.maxlen_bound) atomicAdd(b.too_long, 1u);
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
    gi.m = ((flags & DG_Q_TOO_SHORT) || explicit_set || (strand && !b.reverse) || m > b.maxlen_bound) ? 0u : m;
    gi.d_win = d | (bad == 0 ? 256u : 0u);
    if (b.fastK && gi.m && bad == 0 && d == 1 && m <= 31 && m >= b.fastK + 1) gi.d_win |= 512u;
    if (b.fast2K && gi.m && bad == 0 && d == 2 && m <= 30 && m >= b.fast2K + 2) gi.d_win |= 1024u;
    if (bad == 0 && m <= 32) gi.qpk = strand ? pk_rv : pk_fw;
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
    generic += (gi.m != 0 && !(gi.d_win & 512u));
  }
  // groups the flat distance-1 kernel does not take: the host launches the generic kernels for them (and repeats a batch it
  // started without, run_batch).  A flag, not a count: every lane that has one stores the same 1.
  if (generic && b.fastK) *n_generic = 1u;
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
  return f.ktab[code];
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
  for (int off = 32; off > 0; off >>= 1) nprobe += __shfl_xor(nprobe, off);
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
        const uint2 iv = f.ktab[s_pk & kmask];
        ++nlook;
        lo = iv.x;
        hi = iv.y;
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
  for (int off = 32; off > 0; off >>= 1) {
    steps += __shfl_xor(steps, off);
    nlook += __shfl_xor(nlook, off);
    nhead += __shfl_xor(nhead, off);
  }
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
};
// r04: (i) the three wavefronts that have nothing to do behind the probe phase END there instead of waiting at the barrier behind
// the dense phase (the usual workgroup has ~40 survivors, one wavefront's worth): the r04a counters showed the kernel resident at
// 6-7 of 8 wavefronts per SIMD, two thirds of the wave cycles waiting — three of four of those slots held by wavefronts parked at
// that barrier; (ii) the LDS list is dynamic (lcap entries, 256 by default: 5.6 KB, 512 when the previous batch's workgroups held
// more than ~64 strings on average) and the survivor queue holds 512 entries, filled in rounds when more survive, so that the
// freed slots can be taken by new workgroups.
static constexpr u32 FUSED_LCAP = 512;   // largest LDS list
static constexpr u32 FUSED_QCAP = 512;   // survivor queue entries per round
static inline u32 fused_lds_bytes(u32 lcap) { return lcap * (8u + 4u + 4u + 2u + 2u + 2u); }
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
        const KfCopy c2 = kf_copy(f.kf2, R < K2 ? R : (K2 ? K2 - 1 : 0u));
        const KfCopy c1 = kf_copy(f.kf, R < K ? R : K - 1);
        const u64 mask2 = K2 ? (1ULL << (2 * K2)) - 1 : 0ULL;
        const u32* const idle = reinterpret_cast<const u32*>(f.ktab);
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
          mask_all |= (((valid >> op) & 1u) & present) << op;
        }
        nprobe = (u32)__popc(probe);
      }
    }
  }
  for (int off = 32; off > 0; off >>= 1) nprobe += __shfl_xor(nprobe, off);
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
        u32 lo = 0, hi = 0;
        if (to_lds) nhead += (K2 && mlen > K2);
        if (head_window_occurs(f, s_pk, mlen, raw.z - pos)) {
          const uint2 iv = f.ktab[s_pk & kmask];
          if (to_lds) ++nlook;
          lo = iv.x;
          hi = iv.y;
        }
        u64 rs = s_pk >> (2 * K);
        u32 n = mlen - K;
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
              l_meta[at] = (u16)(mlen | (lg << 6));
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
  rounds(true, leave != 0);
  if (gone) return;
  for (int off = 32; off > 0; off >>= 1) {
    steps += __shfl_xor(steps, off);
    nlook += __shfl_xor(nlook, off);
    nhead += __shfl_xor(nhead, off);
  }
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
        const u64 occ = (u64)l_hi[i] - l_lo[i];
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
        const u64 occ = (u64)l_hi[x] - l_lo[x];
        before += occ < b.max_locations ? occ : b.max_locations;
      }
    }
    if (room) {
      Sel sv;
      sv.lo = l_lo[i];
      sv.hi = l_hi[i];
      sv.len = alen;
      sv.take = 0;
      sv.hbase = 0;
      if (TAKE) {  // hunter.h:349-357: strings are located in set order, forward strand first, while hits < max_locations
        const u64 M = b.max_locations, occ = (u64)sv.hi - sv.lo;
        const u64 h0 = before < M ? before : M, h1 = before + occ < M ? before + occ : M;
        sv.hbase = (u32)h0;
        sv.take = (u32)(h1 - h0);
      }
      sv.g = g_first + lg;
      fs.sel[(u64)shard * fs.cap + wbase + g_base[lg] + r] = sv;
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
__global__ void __launch_bounds__(256) k_search2p(FmView f, Batch b, SearchOut o) {
  // Survivors of a pass are kept as one 64-bit mask per lane (bit 8*op1 + op2) instead of a queue of entries: 3 KB of LDS
  // whatever survives (a queue that holds every candidate of a pass needs 32 KB and left four wavefronts per SIMD resident;
  // r02: 8.8 -> 7.6 ms with room for six), and nothing can overflow.  The dense phase numbers the set bits with a prefix sum
  // over the lanes and finds its e-th one by a binary search over the prefix plus a select inside the lane's mask.
  __shared__ unsigned long long q_mask[256];
  __shared__ u32 q_ex[256 + 1];  // exclusive prefix of the lanes' survivor counts, [256] = total
  __shared__ u32 q_wave[4];
  __shared__ u32 g_slots;
  const u32 gid = blockIdx.x;
  const uint4 raw = *reinterpret_cast<const uint4*>(b.ginfo + gid);
  const u32 m = raw.z, d_win = raw.w;
  if (!m || !(d_win & 1024u)) return;  // uniform for the workgroup
  const u64 qpk = (u64)raw.y << 32 | raw.x;
  if (threadIdx.x == 0) g_slots = 0;
  __syncthreads();
  const u32 K = f.K, K2 = f.kf2.nr ? f.kf2.k : 0u;
  const u64 kmask = (1ULL << (2 * K)) - 1, mask2 = K2 ? (1ULL << (2 * K2)) - 1 : 0ULL;
  const u32 lane = threadIdx.x & 63;
  const u32 npairs = m * (m + 1) / 2 - 1;  // p2 = 1..m-1, p1 = p2..m
  const u32 shard = blockIdx.x & (NSHARD - 1);
  u32 steps = 0, nlook = 0, nprobe = 0;
  for (u32 w0 = 0; w0 < npairs; w0 += 256) {
    const u32 w = w0 + threadIdx.x;
    u64 surv = 0;
    if (w < npairs) {
      u32 p1, p2;
      pair_of(w, m, p2, p1);
      const u32 R1 = m - p1;
      const KfCopy c2 = kf_copy(f.kf2, R1 < K2 ? R1 : (K2 ? K2 - 1 : 0u));
      const KfCopy c1 = kf_copy(f.kf, R1 < K ? R1 : K - 1);
      const u32 qa = (u32)(qpk >> (2 * R1)) & 3u;                       // q[p1-1]
      const u32 qb = R1 ? (u32)(qpk >> (2 * R1 - 2)) & 3u : 4u;         // q[p1], 4 = none
      const u32 q2a = (u32)(qpk >> (2 * (m - p2))) & 3u;                // q[p2-1]
      const u32 q2b = (u32)(qpk >> (2 * (m - p2) - 2)) & 3u;            // q[p2] (p2 < m)
#pragma unroll
      for (u32 op1 = 0; op1 < 8; ++op1) {
        const bool ins1 = op1 >= 4;
        // the first operation leaves p2 characters to its left; nothing is inserted after the last character
        bool v1 = (p1 > p2 || ins1) && !(p1 == m && ins1);
        if (op1 == 0) v1 = v1 && qb != qa;
        if (ins1) v1 = v1 && !(p1 >= 2 && qa == op1 - 4 && p2 + 2 <= p1);
        if (v1) {
          u64 s1;
          u32 l1, w1;
          apply_edit(qpk, m, p1, op1, s1, l1, w1);
          const u32 posp = ins1 ? p1 : p1 - 1;  // characters left of the first operation
          u32 mask8 = 0;
          // all addresses first, then the eight loads back to back, then the bits
          const u32* const idle = reinterpret_cast<const u32*>(f.ktab);  // what a lane without a probe reads
          const u32* addr[8];
          u32 bit[8], word[8], valid = 0, probe = 0;
#pragma unroll
          for (u32 op2 = 0; op2 < 8; ++op2) {
            bool v2 = true;
            if (op2 == 0) v2 = !(p2 < posp && q2b == q2a);
            if (op2 >= 4) v2 = !(p2 >= 2 && q2a == op2 - 4);
            u64 s2;
            u32 l2, w2;
            apply_edit(s1, l1, p2, op2, s2, l2, w2);
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
    // number the survivors: inclusive scan of the lanes' counts inside the wavefront, wavefront totals through LDS
    const u32 mine = (u32)__popcll(surv);
    u32 incl = mine;
    for (int off = 1; off < 64; off <<= 1) {
      const u32 v = __shfl_up(incl, off);
      if ((int)lane >= off) incl += v;
    }
    if (lane == 63) q_wave[threadIdx.x >> 6] = incl;
    q_mask[threadIdx.x] = surv;
    __syncthreads();
    u32 before = 0;
    for (u32 k = 0; k < (threadIdx.x >> 6); ++k) before += q_wave[k];
    q_ex[threadIdx.x] = before + incl - mine;
    const u32 qn = q_wave[0] + q_wave[1] + q_wave[2] + q_wave[3];
    __syncthreads();
    for (u32 e0 = 0; e0 < qn; e0 += 256) {
      if (e0 + (threadIdx.x & ~63u) >= qn) break;  // wavefront without work
      const u32 e = e0 + threadIdx.x;
      bool leaf = false;
      u32 lo = 0, hi = 0, w1 = 0, w2 = 0;
      if (e < qn) {
        // the lane that holds survivor e: the last one whose exclusive prefix is <= e; then its (e - prefix)-th set bit
        u32 L = 0;
#pragma unroll
        for (u32 step = 128; step > 0; step >>= 1)
          if (q_ex[L + step] <= e) L += step;
        unsigned long long mk = q_mask[L];
        for (u32 r = e - q_ex[L]; r > 0; --r) mk &= mk - 1;
        const u32 bitno = (u32)__ffsll((long long)mk) - 1u;
        u32 p1, p2, l1, l2;
        u64 s1, s2;
        pair_of(w0 + L, m, p2, p1);
        apply_edit(qpk, m, p1, bitno >> 3, s1, l1, w1);
        apply_edit(s1, l1, p2, bitno & 7u, s2, l2, w2);
        nprobe += (K2 && l2 > K2);
        if (head_window_occurs(f, s2, l2, l1 - p2)) {
          const uint2 iv = f.ktab[s2 & kmask];
          ++nlook;
          lo = iv.x;
          hi = iv.y;
        }
        u64 rs = s2 >> (2 * K);
        u32 nr = l2 - K;
        while (nr && lo < hi) {
          bs_extend_code_narrow(f, lo, hi, (u32)rs & 3u);
          rs >>= 2;
          --nr;
          ++steps;
        }
        leaf = lo < hi;
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
          lf->nops = 2;
          lf->ops[0] = w1;
          lf->ops[1] = w2;
#pragma unroll
          for (int k = 2; k < (int)DMAX; ++k) lf->ops[k] = 0u;
        }
      }
    }
    __syncthreads();  // the masks and prefixes of this pass are not needed any more
  }
  for (int off = 32; off > 0; off >>= 1) {
    steps += __shfl_xor(steps, off);
    nlook += __shfl_xor(nlook, off);
    nprobe += __shfl_xor(nprobe, off);
  }
  if (lane == 0) {
    if (steps) atomicAdd(&o.ctr->steps[shard], (unsigned long long)steps);
    if (nlook) atomicAdd(&o.ctr->lookups[shard], (unsigned long long)nlook);
    if (nprobe) atomicAdd(&o.ctr->probes[shard], (unsigned long long)nprobe);
  }
  if (threadIdx.x == 0 && g_slots) atomicAdd(o.grp_cnt + gid, g_slots);
}

template <bool INDEL, int D>
__global__ void __launch_bounds__(256) k_search(FmView f, Batch b, SearchOut o, u32 items) {
  // lane layout: the long-running "rest" lanes come first, packed densely (a rest lane among 63 short item lanes
  // would pin its whole wavefront); item lanes follow, (items-1) consecutive lanes per (query, strand)
  u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  const u64 ngrp = b.nq * 2;
  u64 gid;   // 2*query + strand
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
  const u64 q = gid >> 1;
  const u32 strand = (u32)(gid & 1);
  GidInfo gi;
  gi.qpk = 0;
  gi.m = 0;
  gi.d_win = 0;
  if (active) gi = b.ginfo[gid];
  if (gi.m == 0 || (gi.d_win & (512u | 1024u))) active = false;  // not searched, or taken by a flat kernel
  if (active) {
    const u8* seq = (strand ? b.rv : b.fw) + b.qoff[q];
    const u32 m = gi.m;
    u32 d = gi.d_win & 255;
    if (d > (u32)D) d = D;  // cannot happen: the host instantiates D >= the largest effective distance
    const bool use_win = f.K != 0 && m >= f.K + d && (gi.d_win & 256);
    const bool rest = item == items - 1;
    // lanes of a split launch: without the table (or without budget) only the rest lane works, as a full search
    if (!rest && (!use_win || d == 0)) active = false;
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
          if (here < 4) alive = frame_emit(f, F, here, steps, lookups, probes);
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
          const uint2 iv = f.ktab[code & ((1ULL << (2 * K)) - 1)];
          ++lookups;
          lo = iv.x;
          hi = iv.y;
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

__global__ void k_leaf_overflow(Counters* ctr, u32 shard_cap, u32 surv_cap) {  // NSHARD lanes
  u32 k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < NSHARD && (ctr->leaf_cnt[k] > shard_cap || ctr->surv_cnt[k] > surv_cap)) atomicOr(&ctr->overflow, 1u);
}
// what the host needs at the end of a batch, in 64 bytes instead of the 36 KB of sharded counters
struct Summary {
  unsigned long long nleaf, worst_shard, steps, lookups, sa_reads, win_bytes, nhits, overflow, refused, too_long, probes, worst_surv;
  unsigned long long jobs_big, jobs_small;  // repeat-rich strings queued by k_locate (workgroup / wavefront jobs)
  unsigned long long worst_sel;             // fullest slice of the flat Sel region (k_search1s)
  unsigned long long fused_leaves;          // occurring strings k_search1s kept in LDS (they never became Leaf records)
  unsigned long long n_generic;             // groups searched outside the flat distance-1 kernel (k_prepare)
};
// Totals straight into the pinned host record (no atomics over the bus, no separate copy); the host reads it
// after the batch's single stream synchronisation.  The batch's last kernel: it also leaves the counters ZEROED for the next
// batch (which then needs no memset in front).  One workgroup.
// (r03 tried to run this in "the workgroup of the verify kernel that finishes last": the device-scope fence every workgroup needs
// before it reports in writes back its XCD's L2 — the 8 300 workgroups of a repeat-genome step went from 1.05 to 1.60 ms.  On
// this part workgroups of one launch do not talk to each other cheaply; a 7 us kernel of its own is the better deal.)
DG_DEV void batch_finish(Counters* ctr, const u64* nhits, Summary* host_out) {
  constexpr int NF = 10;  // fields 1, 7 and 8 are maxima, the others sums
  __shared__ unsigned long long acc[NF];
  if (threadIdx.x < NF) acc[threadIdx.x] = 0;
  __syncthreads();
  unsigned long long v[NF] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (u32 k = threadIdx.x; k < NSHARD; k += blockDim.x) {
    const unsigned long long lc = ctr->leaf_cnt[k], sc = ctr->surv_cnt[k], sl = ctr->sel_cnt[k];
    v[8] = sl > v[8] ? sl : v[8];
    v[9] += ctr->fused_leaves[k];
    v[0] += lc;
    v[1] = lc > v[1] ? lc : v[1];
    v[2] += ctr->steps[k];
    v[3] += ctr->lookups[k];
    v[4] += ctr->sa_reads[k];
    v[5] += ctr->win_bytes[k];
    v[6] += ctr->probes[k];
    v[7] = sc > v[7] ? sc : v[7];
  }
#pragma unroll
  for (int f = 0; f < NF; ++f) {
    unsigned long long x = v[f];
    for (int off = 32; off > 0; off >>= 1) {
      const unsigned long long o = __shfl_xor(x, off);
      x = (f == 1 || f == 7 || f == 8) ? (o > x ? o : x) : x + o;
    }
    if ((threadIdx.x & 63) == 0) {
      if (f == 1 || f == 7 || f == 8) atomicMax(&acc[f], x);
      else atomicAdd(&acc[f], x);
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    host_out->worst_surv = acc[7];
    host_out->nleaf = acc[0];
    host_out->worst_shard = acc[1];
    host_out->steps = acc[2];
    host_out->lookups = acc[3];
    host_out->sa_reads = acc[4];
    host_out->win_bytes = acc[5];
    host_out->probes = acc[6];
    host_out->nhits = *nhits;
    host_out->overflow = ctr->overflow;
    host_out->refused = ctr->pad_[1];
    host_out->too_long = ctr->pad_[2];
    host_out->jobs_big = ctr->pad_[0];
    host_out->jobs_small = ctr->pad_[4];
    host_out->worst_sel = acc[8];
    host_out->fused_leaves = acc[9];
    host_out->n_generic = ctr->pad_[6];
    __threadfence_system();
  }
  __syncthreads();
  u32* w = reinterpret_cast<u32*>(ctr);
  for (u32 i = threadIdx.x; i < sizeof(Counters) / 4; i += blockDim.x) w[i] = 0;
}
__global__ void __launch_bounds__(NSHARD) k_summary_block(Counters* ctr, const u64* nhits, Summary* host_out) { batch_finish(ctr, nhits, host_out); }
// group leaves by (query,strand): dst = grp_off[qs] + slot
__global__ void k_group(const Leaf* in, u32 shard_cap, const Counters* ctr, const u64* grp_off, Leaf* out) {
  u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (ctr->overflow || t >= (u64)NSHARD * shard_cap || (u32)(t % shard_cap) >= ctr->leaf_cnt[t / shard_cap]) return;
  Leaf lf = in[t];
  out[grp_off[lf.qs] + lf.slot] = lf;
}

// ------------------------------------------------------------------------------------------------------------
// Left-to-right reader of the string a leaf stands for (query + recorded edits).
struct LeafReader {
  const u8* seq;
  u64 qpk;      // the query 2-bit packed (q[i] at bits 2(m-1-i)) when `packed`: no memory access per character
  bool packed;
  u32 m;
  const u32* ops;
  int k;     // next op (they were recorded right-to-left, so read from the last one)
  u32 qpos;  // next query index to output
  DG_DEV void init(const u8* s, u32 m_, const Leaf& lf) {
    seq = s;
    qpk = 0;
    packed = false;
    m = m_;
    ops = lf.ops;
    k = (int)lf.nops - 1;
    qpos = 0;
  }
  // any leaf of the batch: an explicit pattern reads its own bytes, everything else the query + recorded edits
  DG_DEV void init_any(const Batch& b, const u8* s, u32 m_, const Leaf& lf) {
    if (lf.nops == LEAF_EXPLICIT) {
      const u64 x0 = b.xs_off[lf.ops[0]];
      init(b.xs_bytes + x0, (u32)(b.xs_off[lf.ops[0] + 1] - x0), lf);
      k = -1;
    } else init(s, m_, lf);
  }
  DG_DEV void init_packed(u64 q, u32 m_, const Leaf& lf) {
    init(nullptr, m_, lf);
    qpk = q;
    packed = true;
  }
  DG_DEV int at(u32 i) const { return packed ? (int)((qpk >> (2 * (m - 1 - i))) & 3) : (int)seq[i]; }
  DG_DEV int next() {  // code 0..4, or -1 at the end
    for (;;) {
      if (k < 0) return qpos < m ? at(qpos++) : -1;
      u32 op = ops[k], p = op >> 4, kind = (op >> 2) & 3, c = op & 3;
      u32 upto = kind == OP_I ? p : p - 1;
      if (qpos < upto) return at(qpos++);
      --k;
      qpos = p;
      if (kind != OP_D) return (int)c;
    }
  }
};
DG_DEV u32 leaf_len(const Batch& b, u32 m, const Leaf& lf) {
  if (lf.nops == LEAF_EXPLICIT) return (u32)(b.xs_off[lf.ops[0] + 1] - b.xs_off[lf.ops[0]]);
  u32 len = m;
  for (u32 k = 0; k < lf.nops; ++k) {
    u32 kind = (lf.ops[k] >> 2) & 3;
    len += (kind == OP_I);
    len -= (kind == OP_D);
  }
  return len;
}
// is string(b) found inside string(a)?  (std::string::find, neighbors.h:37,39)
DG_DEV bool leaf_contains(const Batch& bt, const u8* seq, u32 m, const Leaf& a, u32 la, const Leaf& b, u32 lb) {
  if (lb > la) return false;
  for (u32 o = 0; o + lb <= la; ++o) {
    LeafReader ra, rb;
    ra.init_any(bt, seq, m, a);
    rb.init_any(bt, seq, m, b);
    for (u32 i = 0; i < o; ++i) (void)ra.next();
    bool same = true;
    for (u32 i = 0; i < lb; ++i)
      if (ra.next() != rb.next()) {
        same = false;
        break;
      }
    if (same) return true;
  }
  return false;
}
// std::string operator< on the ASCII strings
DG_DEV bool leaf_less(const Batch& bt, const u8* seq, u32 m, const Leaf& a, const Leaf& b) {
  LeafReader ra, rb;
  ra.init_any(bt, seq, m, a);
  rb.init_any(bt, seq, m, b);
  for (;;) {
    int x = ra.next(), y = rb.next();
    if (x < 0 || y < 0) return x < 0 && y >= 0;
    if (x != y) return ascii_rank((u32)x) < ascii_rank((u32)y);
  }
}

// ---- packed strings: the common case (string length <= 42) keeps every neighbourhood string in 128 bits ----
// 3 bits per character, code = ASCII rank + 1 (A1 C2 G3 N4 T5), first character in the top bits, zero padded: unsigned
// 128-bit comparison == std::string operator<, substring tests are shifts and masks.
static constexpr u32 PACK_MAX_LEN = 42;
struct PLeaf {
  u64 hi, lo;  // the 128-bit packed string
  u32 len;
  u32 sa_lo, sa_hi;
  u32 qs;  // 2*query + strand
};
DG_DEV void p128_shl(u64& hi, u64& lo, u32 s) {  // s < 128
  if (s >= 64) {
    hi = lo << (s - 64);
    lo = 0;
  } else if (s) {
    hi = (hi << s) | (lo >> (64 - s));
    lo <<= s;
  }
}
DG_DEV void p128_topmask(u64& hi, u64& lo, u32 nbits) {  // keep the top nbits (<= 128)
  if (nbits >= 128) return;
  if (nbits >= 64) {
    u32 r = nbits - 64;
    lo &= r ? ~0ULL << (64 - r) : 0ULL;
  } else {
    lo = 0;
    hi &= nbits ? ~0ULL << (64 - nbits) : 0ULL;
  }
}
// group leaves by (query,strand) and pack their strings: dst = grp_off[qs] + slot
__global__ void k_group_pack(Batch b, const Leaf* in, u32 shard_cap, const Counters* ctr, const u64* grp_off, PLeaf* out) {
  u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (ctr->overflow || t >= (u64)NSHARD * shard_cap || (u32)(t % shard_cap) >= ctr->leaf_cnt[t / shard_cap]) return;
  Leaf lf = in[t];
  const u64 q = lf.qs >> 1;
  const GidInfo gi = b.ginfo[lf.qs];  // one 16-byte record: length, and the packed query when it has no N and <= 32 nt
  const u32 m = gi.m;
  if (lf.nops != LEAF_EXPLICIT && lf.nops <= DMAX && (gi.d_win & 256) && m + lf.nops <= 32) {
    // the usual leaf: the 2-bit packed query with its recorded operations applied right to left (positions refer to the
    // unchanged part left of the previous operation), then 2 -> 3 bits per character
    u64 x = gi.qpk;
    u32 len = m;
    for (u32 k = 0; k < lf.nops; ++k) {
      const u32 w = lf.ops[k], pos = w >> 4, kind = (w >> 2) & 3u, c = w & 3u;
      const u32 R = len - pos;
      const u64 low = x & ((1ULL << (2 * R)) - 1);
      if (kind == OP_D) {
        x = low | ((x >> (2 * R + 2)) << (2 * R));
        --len;
      } else if (kind == OP_S) {
        x = (x & ~(3ULL << (2 * R))) | ((u64)c << (2 * R));
      } else {
        x = low | ((u64)c << (2 * R)) | ((x >> (2 * R)) << (2 * R + 2));
        ++len;
      }
    }
    u64 hi = 0, lo = 0;
    for (u32 i = 0; i < len; ++i) {
      const u32 c2 = (u32)(x >> (2 * (len - 1 - i))) & 3u;
      hi = (hi << 3) | (lo >> 61);
      lo = (lo << 3) | (u64)(c2 + 1 + (c2 == 3));  // ASCII rank + 1: A1 C2 G3 T5
    }
    p128_shl(hi, lo, 128 - 3 * len);
    PLeaf p;
    p.hi = hi;
    p.lo = lo;
    p.len = len;
    p.sa_lo = lf.lo;
    p.sa_hi = lf.hi;
    p.qs = lf.qs;
    out[grp_off[lf.qs] + lf.slot] = p;
    return;
  }
  LeafReader r;
  if (lf.nops == LEAF_EXPLICIT) r.init_any(b, nullptr, 0, lf);
  else if ((gi.d_win & 256) && m <= 32) r.init_packed(gi.qpk, m, lf);
  else r.init(((lf.qs & 1) ? b.rv : b.fw) + b.qoff[q], m, lf);
  u64 hi = 0, lo = 0;
  u32 len = 0;
  for (int c = r.next(); c >= 0; c = r.next()) {
    u64 code = ascii_rank((u32)c) + 1;
    u32 sh = 125 - 3 * len;  // character i occupies bits [125-3i, 127-3i]
    if (sh >= 64) hi |= code << (sh - 64);
    else if (sh >= 62) {  // straddles the two words (sh = 62 or 63)
      lo |= code << sh;
      hi |= code >> (64 - sh);
    } else lo |= code << sh;
    ++len;
  }
  PLeaf p;
  p.hi = hi;
  p.lo = lo;
  p.len = len;
  p.sa_lo = lf.lo;
  p.sa_hi = lf.hi;
  p.qs = lf.qs;
  out[grp_off[lf.qs] + lf.slot] = p;
}
DG_DEV bool pleaf_contains(const PLeaf& a, const PLeaf& x) {  // is x inside a?  (std::string::find)
  if (x.len > a.len) return false;
  for (u32 o = 0; o + x.len <= a.len; ++o) {
    u64 hi = a.hi, lo = a.lo;
    p128_shl(hi, lo, 3 * o);
    p128_topmask(hi, lo, 3 * x.len);
    if (hi == x.hi && lo == x.lo) return true;
  }
  return false;
}
DG_DEV bool pleaf_less(const PLeaf& a, const PLeaf& x) { return a.hi < x.hi || (a.hi == x.hi && a.lo < x.lo); }

// Minimal-set filter and ordering with one lane per LEAF (any group size stays parallel):
//   k_leaf_alive  leaf survives unless another string of its group is a proper substring, or an equal one has a lower slot
//   k_leaf_rank   rank among the survivors in std::set order -> Sel written at its sorted position
//   k_take        per query: hunter.h:350,357 gating over forward then reverse strings
// above: groups of more than `above` leaves only (the others were served by k_group_select)
__global__ void k_leaf_alive(const PLeaf* G, const u64* grp_off, u64 nq2, u32 indel, u8* alive, const Counters* ctr, u32 above) {
  u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (ctr->overflow || t >= grp_off[nq2]) return;
  bool ok = true;
  if (indel || above) {
    const PLeaf a = G[t];
    const u64 g0 = grp_off[a.qs], g1 = grp_off[a.qs + 1];
    if (g1 - g0 <= above) return;
    if (!indel) {
      alive[t] = true;
      return;
    }
    // "x occurs in a at offset o" = top 3*len(x) bits of (a << 3o) equal x.  The strings of a group differ in length by at
    // most 2d, so the first five shifts of a (enough for d <= 2) are made once and stay in registers.
    constexpr int NSH = 5;
    u64 ah[NSH], al[NSH];
#pragma unroll
    for (int o = 0; o < NSH; ++o) {
      ah[o] = a.hi;
      al[o] = a.lo;
      p128_shl(ah[o], al[o], 3 * o);
    }
    for (u64 j = g0; j < g1 && ok; ++j) {
      if (j == t) continue;
      const PLeaf x = G[j];
      if (x.len > a.len) continue;
      const u32 diff = a.len - x.len, nbits = 3 * x.len;  // nbits <= 126
      const u64 mh = nbits >= 64 ? ~0ULL : (nbits ? ~0ULL << (64 - nbits) : 0ULL);
      const u64 ml = nbits > 64 ? ~0ULL << (128 - nbits) : 0ULL;
      bool hit = false;
#pragma unroll
      for (int o = 0; o < NSH; ++o)
        if ((u32)o <= diff) hit = hit || ((ah[o] & mh) == x.hi && (al[o] & ml) == x.lo);
      for (u32 o = NSH; o <= diff && !hit; ++o) {  // distances above 2
        u64 hi = a.hi, lo = a.lo;
        p128_shl(hi, lo, 3 * o);
        hit = (hi & mh) == x.hi && (lo & ml) == x.lo;
      }
      if (hit) ok = (x.len == a.len) && (t < j);
    }
  }
  alive[t] = ok;
}
__global__ void k_leaf_rank(const PLeaf* G, const u64* grp_off, u64 nq2, const u8* alive, Sel* sel, u32* nsel,
                            const Counters* ctr, u32 above) {
  u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (ctr->overflow || t >= grp_off[nq2]) return;
  const PLeaf a = G[t];
  const u64 g0 = grp_off[a.qs], g1 = grp_off[a.qs + 1];
  if (g1 - g0 <= above) return;
  u32 r = 0, ns = 0;
  for (u64 j = g0; j < g1; ++j) {
    if (!alive[j]) continue;
    ++ns;
    if (j != t && pleaf_less(G[j], a)) ++r;
  }
  if (t == g0) nsel[a.qs] = ns;  // groups without leaves keep the 0 of the memset
  if (!alive[t]) return;
  Sel s;
  s.lo = a.sa_lo;
  s.hi = a.sa_hi;
  s.len = a.len;
  s.take = 0;
  s.hbase = 0;
  sel[g0 + r] = s;
}
// The same selection with one WORKGROUP per (query, strand) group, for batches whose groups are not tiny (distance >= 2:
// ~70 occurring strings per strand of a 20-mer on a 3.1 Gb genome, where the pair loop of k_leaf_alive and the counting loop
// of k_leaf_rank cost group-size^2).  The packed strings are sorted in LDS (bitonic, 128-bit keys: unsigned order ==
// std::string order, i.e. the std::set order the reference walks, hunter.h:349); duplicates are then neighbours, and "some
// other string of the group is a proper substring of this one" (neighbors.h:29-45) becomes one binary search per (length,
// offset) window of the string — at most 14 windows at distance 2.  Survivors leave in sorted order, so the rank comes for
// free.  Groups above SELCAP leaves stay with the lane-per-leaf kernels.
static constexpr u32 SELCAP = 1024;
__global__ void __launch_bounds__(128) k_group_select(const PLeaf* G, const u64* grp_off, u32 indel, Sel* sel, u32* nsel, const Counters* ctr) {
  __shared__ unsigned long long kh[SELCAP], kl[SELCAP];
  __shared__ u16 ix[SELCAP];  // bits 0-9 position in the group, bits 10-15 string length
  __shared__ u32 s_minlen, s_w[2];
  __shared__ u32 bm[128];  // 4096-bit membership sketch of the group's keys: most windows are turned away without a search
  const u64 g = blockIdx.x;
  if (ctr->overflow) return;
  const u64 g0 = grp_off[g];
  const u32 k = (u32)(grp_off[g + 1] - g0);
  if (k == 0 || k > SELCAP) return;  // empty: nsel stays 0;  huge: k_leaf_alive / k_leaf_rank
  if (k == 1) {
    if (threadIdx.x == 0) {
      const PLeaf a = G[g0];
      Sel o;
      o.lo = a.sa_lo;
      o.hi = a.sa_hi;
      o.len = a.len;
      o.take = 0;
      o.hbase = 0;
      sel[g0] = o;
      nsel[g] = 1;
    }
    return;
  }
  u32 n2 = 2;
  while (n2 < k) n2 <<= 1;
  if (threadIdx.x == 0) {
    s_minlen = 0xFFFFFFFFu;
    s_w[1] = 0;  // stays 0 when the workgroup is a single wavefront
  }
  for (u32 i = threadIdx.x; i < 128; i += blockDim.x) bm[i] = 0;
  __syncthreads();
  auto sketch = [](u64 h, u64 l) -> u32 { return (((u32)h ^ (u32)(h >> 32) ^ (u32)l ^ (u32)(l >> 32)) * 0x9E3779B1u) >> 20; };
  for (u32 i = threadIdx.x; i < n2; i += blockDim.x) {
    if (i < k) {
      const PLeaf a = G[g0 + i];
      kh[i] = a.hi;
      kl[i] = a.lo;
      ix[i] = (u16)(i | (a.len << 10));
      atomicMin(&s_minlen, a.len);
      const u32 hb = sketch(a.hi, a.lo);
      atomicOr(&bm[hb >> 5], 1u << (hb & 31));
    } else {
      kh[i] = ~0ULL;
      kl[i] = ~0ULL;
      ix[i] = 0xFFFFu;
    }
  }
  __syncthreads();
  for (u32 kk = 2; kk <= n2; kk <<= 1)
    for (u32 j = kk >> 1; j > 0; j >>= 1) {
      for (u32 i = threadIdx.x; i < n2; i += blockDim.x) {
        const u32 l = i ^ j;
        if (l > i) {
          const u64 ah = kh[i], al = kl[i], bh = kh[l], bl = kl[l];
          const bool gt = ah > bh || (ah == bh && al > bl);
          if (gt == ((i & kk) == 0)) {
            kh[i] = bh;
            kl[i] = bl;
            kh[l] = ah;
            kl[l] = al;
            const u16 t = ix[i];
            ix[i] = ix[l];
            ix[l] = t;
          }
        }
      }
      __syncthreads();
    }
  const u32 minlen = s_minlen;
  const u32 lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  u32 base = 0;
  for (u32 c0 = 0; c0 < k; c0 += blockDim.x) {
    const u32 i = c0 + threadIdx.x;
    bool ok = false;
    u32 len = 0;
    if (i < k) {
      const u64 ah = kh[i], al = kl[i];
      len = ix[i] >> 10;
      ok = !(i > 0 && kh[i - 1] == ah && kl[i - 1] == al);  // equal strings: the first one stays
      if (ok && indel) {
        for (u32 sl = minlen; sl < len && ok; ++sl) {
          const u32 nbits = 3 * sl;  // <= 126
          const u64 mh = nbits >= 64 ? ~0ULL : (nbits ? ~0ULL << (64 - nbits) : 0ULL);
          const u64 ml = nbits > 64 ? ~0ULL << (128 - nbits) : 0ULL;
          for (u32 o = 0; o + sl <= len && ok; ++o) {
            u64 h = ah, l = al;
            p128_shl(h, l, 3 * o);
            h &= mh;
            l &= ml;
            const u32 hb = sketch(h, l);
            if (!((bm[hb >> 5] >> (hb & 31)) & 1u)) continue;
            u32 lo = 0, hi = k;
            while (lo < hi) {
              const u32 mid = (lo + hi) >> 1;
              const u64 xh = kh[mid], xl = kl[mid];
              if (xh < h || (xh == h && xl < l)) lo = mid + 1;
              else hi = mid;
            }
            if (lo < k && kh[lo] == h && kl[lo] == l) ok = false;  // a shorter string of the group occurs in this one
          }
        }
      }
    }
    const unsigned long long mk = __ballot(ok);
    if (lane == 0) s_w[wave] = (u32)__popcll(mk);
    __syncthreads();
    if (ok) {
      const u32 r = base + (wave ? s_w[0] : 0u) + (u32)__popcll(mk & ((1ULL << lane) - 1));
      const PLeaf a = G[g0 + (ix[i] & 1023u)];
      Sel o;
      o.lo = a.sa_lo;
      o.hi = a.sa_hi;
      o.len = a.len;
      o.take = 0;
      o.hbase = 0;
      sel[g0 + r] = o;
    }
    base += s_w[0] + s_w[1];
    __syncthreads();
  }
  if (threadIdx.x == 0) nsel[g] = base;
}
// first Sel slot of group g: the flat region (k_search1s set selbase) or flat_slots + grp_off[g] (generic path)
DG_DEV u64 sel_base_of(const u32* selbase, const u64* grp_off, u64 flat_slots, u64 g) {
  const u32 sb = selbase[g];
  return sb != 0xFFFFFFFFu ? (u64)sb : flat_slots + grp_off[g];
}
__global__ void k_take(Batch b, const u64* grp_off, const u32* selbase, u64 flat_slots, const u32* nsel, Sel* sel, u32* qhits, const Counters* ctr) {
  u64 q = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= b.nq) return;
  if (ctr->overflow) {
    qhits[q] = 0;
    return;
  }
  u64 hits = 0;
  for (u32 strand = 0; strand < 2; ++strand) {
    const u32 ns = nsel[2 * q + strand];
    if (!ns) continue;  // (grp_off is not even computed when the generic kernels were left out)
    Sel* S = sel + sel_base_of(selbase, grp_off, flat_slots, 2 * q + strand);
    for (u32 r = 0; r < ns; ++r) {
      u64 occs = (u64)S[r].hi - S[r].lo;
      u64 take = 0;
      if (hits < b.max_locations) take = occs < b.max_locations - hits ? occs : b.max_locations - hits;
      S[r].take = (u32)take;
      S[r].hbase = (u32)hits;
      hits += take;
    }
  }
  qhits[q] = (u32)hits;
  u32 fl = b.qflags[q];
  if (hits >= b.max_locations && !(fl & DG_Q_TOO_SHORT)) {  // hunter.h:434
    fl |= DG_Q_MAX_MATCHES;
    b.qflags[q] = fl;
  }
  if (b.qinfo) b.qinfo[q] = (fl & 255u) | ((b.qdist[q] & 255u) << 8) | (b.qnondna[q] << 16);  // compact results (dicey_gpu.h DG_QINFO_*)
}

// count mode: occurrences of all kept strings of a (query, strand) group
__global__ void k_group_count(const u64* grp_off, const u32* selbase, u64 flat_slots, const u32* nsel, const Sel* sel, u64 ngrp, u64* out, const Counters* ctr) {
  u64 g = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= ngrp) return;
  u64 sum = 0;
  if (!ctr->overflow) {
    const u32 ns = nsel[g];
    const Sel* S = ns ? sel + sel_base_of(selbase, grp_off, flat_slots, g) : sel;
    for (u32 r = 0; r < ns; ++r) sum += (u64)S[r].hi - S[r].lo;
  }
  out[g] = sum;
}

// Exclusive prefix sum of n 32-bit counts into 64-bit offsets (out[n] = total), lane-independent three-level scheme so
// that no host round trip is needed between the kernels of a batch.
static constexpr u32 SCAN_CHUNK = 64;
__global__ void k_scan_sum(const u32* in, u64 n, u64* part) {  // part[c] = sum of chunk c
  u64 c = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  u64 b0 = c * SCAN_CHUNK;
  if (b0 >= n) return;
  u64 e = b0 + SCAN_CHUNK < n ? b0 + SCAN_CHUNK : n, s = 0;
  for (u64 i = b0; i < e; ++i) s += in[i];
  part[c] = s;
}
__global__ void k_scan_sum64(const u64* in, u64 n, u64* part) {
  u64 c = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  u64 b0 = c * SCAN_CHUNK;
  if (b0 >= n) return;
  u64 e = b0 + SCAN_CHUNK < n ? b0 + SCAN_CHUNK : n, s = 0;
  for (u64 i = b0; i < e; ++i) s += in[i];
  part[c] = s;
}
__global__ void k_scan_top(u64* part, u64 n, u64* total) {  // one lane: exclusive scan of <= a few thousand values
  if (blockIdx.x || threadIdx.x) return;
  u64 run = 0;
  for (u64 i = 0; i < n; ++i) {
    u64 v = part[i];
    part[i] = run;
    run += v;
  }
  *total = run;
}
__global__ void k_scan_apply64(u64* vals, u64 n, const u64* base) {  // vals: chunk sums -> exclusive offsets
  u64 c = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  u64 b0 = c * SCAN_CHUNK;
  if (b0 >= n) return;
  u64 e = b0 + SCAN_CHUNK < n ? b0 + SCAN_CHUNK : n, run = base[c];
  for (u64 i = b0; i < e; ++i) {
    u64 v = vals[i];
    vals[i] = run;
    run += v;
  }
}
__global__ void k_scan_apply(const u32* in, u64 n, const u64* base, u64* out) {
  u64 c = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  u64 b0 = c * SCAN_CHUNK;
  if (b0 >= n) return;
  u64 e = b0 + SCAN_CHUNK < n ? b0 + SCAN_CHUNK : n, run = base[c];
  for (u64 i = b0; i < e; ++i) {
    out[i] = run;
    run += in[i];
  }
}

// One lane per query.  Works in place on the grouped leaf array: `keep` marks survivors, `order` their rank.
__global__ void k_select(Batch b, const Leaf* grouped, const u64* grp_off, Sel* sel, u32* nsel /*[2nq]*/, u32* qhits,
                         u8* scratch_keep, u32* scratch_rank, const Counters* ctr) {
  u64 q = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= b.nq) return;
  if (ctr->overflow) {
    qhits[q] = 0;
    return;
  }
  const u32 m = b.qlen[q];
  u64 hits = 0;
  for (u32 strand = 0; strand < 2; ++strand) {
    const u64 g0 = grp_off[2 * q + strand], g1 = grp_off[2 * q + strand + 1];
    const u32 k = (u32)(g1 - g0);
    nsel[2 * q + strand] = 0;
    if (!k) continue;
    const u8* seq = (strand ? b.rv : b.fw) + b.qoff[q];
    const Leaf* G = grouped + g0;
    u8* keep = scratch_keep + g0;
    u32* rank = scratch_rank + g0;
    // keep[i] <=> no other occurring string is a proper substring of it, and it is the first copy of itself
    for (u32 i = 0; i < k; ++i) {
      u32 li = leaf_len(b, m, G[i]);
      bool alive = true;
      if (b.indel) {
        for (u32 j = 0; j < k && alive; ++j) {
          if (j == i) continue;
          u32 lj = leaf_len(b, m, G[j]);
          if (lj > li) continue;
          if (leaf_contains(b, seq, m, G[i], li, G[j], lj)) alive = (lj == li) && (i < j);  // equal strings: lowest slot stays
        }
      }
      keep[i] = alive;
    }
    // rank among survivors in std::set order
    u32 ns = 0;
    for (u32 i = 0; i < k; ++i) {
      if (!keep[i]) continue;
      u32 r = 0;
      for (u32 j = 0; j < k; ++j)
        if (j != i && keep[j] && leaf_less(b, seq, m, G[j], G[i])) ++r;
      rank[i] = r;
      ++ns;
    }
    Sel* S = sel + g0;
    for (u32 i = 0; i < k; ++i)
      if (keep[i]) {
        Sel s;
        s.lo = G[i].lo;
        s.hi = G[i].hi;
        s.len = leaf_len(b, m, G[i]);
        s.take = 0;
        s.hbase = 0;
        S[rank[i]] = s;
      }
    // hunter.h:350,357: strings in set order while hits < max_locations; per string min(occs, max_locations) positions
    for (u32 r = 0; r < ns; ++r) {
      u64 occs = (u64)S[r].hi - S[r].lo;
      u64 take = 0;
      if (hits < b.max_locations) take = occs < b.max_locations - hits ? occs : b.max_locations - hits;
      S[r].take = (u32)take;
      S[r].hbase = (u32)hits;
      hits += take;
    }
    nsel[2 * q + strand] = ns;
  }
  qhits[q] = (u32)hits;
  u32 fl = b.qflags[q];
  if (hits >= b.max_locations && !(fl & DG_Q_TOO_SHORT)) {  // hunter.h:434
    fl |= DG_Q_MAX_MATCHES;
    b.qflags[q] = fl;
  }
  if (b.qinfo) b.qinfo[q] = (fl & 255u) | ((b.qdist[q] & 255u) << 8) | (b.qnondna[q] << 16);  // compact results (dicey_gpu.h DG_QINFO_*)
}

// ------------------------------------------------------------------------------------------------------------
// Locate: the `take` smallest SA values of [lo,hi), ascending (locate + std::sort + first min(occs,max) entries).
static constexpr u32 TOPK_KMAX = 1024;  // largest `take` k_locate_topk serves (its LDS holds 8 * 1 152 candidate minima)
struct BigJob {  // a repeat-rich string: handled by one workgroup of k_locate_topk / k_locate_big
  u32 lo, occs, take, g, len;
  u32 slot;  // of the kept string (HitSeed::sel)
  u64 out;   // first hit slot
};
// One lane per kept string (r03; r02 walked the strings of a (query, strand) group in one lane — 170 hits per query on the
// repeat-bearing genome made that a chain of several hundred dependent reads).  The lane finds its group through the packed /
// grouped leaf of the same slot (slot_qs: address of that record's `qs` field, slot_stride: record size), serves strings of up to
// 24 occurrences itself and queues the others: up to 256 occurrences for one wavefront (k_locate_small), more for one
// workgroup (k_locate_topk / k_locate_big).
struct LocJobs {
  BigJob* small;
  BigJob* big;
  u32 cap;  // of each list
  u32* n_small;
  u32* n_big;
};
static constexpr u32 LOC_SMALL_MAX = 256;
// A string with up to N occurrences: N loads in flight, a bitonic network on registers (every index is a compile-time constant —
// r02's insertion sort indexed a private array dynamically, i.e. through scratch memory), `take` stores.
template <int N>
DG_DEV void locate_in_registers(const u32* sa, u32 occs, u32 take, HitSeed* out, u32 g, u32 len, u32 slot) {
  u32 v[N];
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] = (u32)i < occs ? sa[i] : 0xFFFFFFFFu;
#pragma unroll
  for (int k = 2; k <= N; k <<= 1)
#pragma unroll
    for (int j = k >> 1; j > 0; j >>= 1)
#pragma unroll
      for (int i = 0; i < N; ++i) {
        const int l = i ^ j;
        if (l > i) {
          const u32 a = v[i], b2 = v[l], mn = a < b2 ? a : b2, mx = a < b2 ? b2 : a;
          v[i] = (i & k) == 0 ? mn : mx;
          v[l] = (i & k) == 0 ? mx : mn;
        }
      }
#pragma unroll
  for (int i = 0; i < N; ++i)
    if ((u32)i < take) out[i] = HitSeed{v[i], g, len, slot};
}
// Slots [0, flat_slots): the flat region (k_search1s; NSHARD slices of flat_cap entries, a slice holds ctr->sel_cnt[shard] strings,
// each naming its group); slots behind it: the generic path's (grp_off based; only when generic_on).
__global__ void __launch_bounds__(256) k_locate(FmView f, const Sel* sel, const u8* slot_qs, u32 slot_stride, const u64* grp_off, const u32* nsel,
                                                u64 ngroups, const u64* hit_off, HitSeed* seeds, Counters* ctr, u64 hit_cap, LocJobs jobs,
                                                u64 flat_slots, u32 flat_cap, u32 generic_on, u32 jobs_on) {
  const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (ctr->overflow || hit_off[ngroups >> 1] > hit_cap) return;
  u64 reads = 0;
  BigJob bj;
  u32 queue = 0;  // 1: wavefront job, 2: workgroup job
  bool have = false;
  u32 g = 0;
  Sel S;
  if (t < flat_slots) {
    const u32 shard = (u32)(t / flat_cap);
    if ((u32)(t - (u64)shard * flat_cap) < ctr->sel_cnt[shard]) {
      S = sel[t];
      g = S.g;
      have = true;
    }
  } else if (generic_on && t - flat_slots < grp_off[ngroups]) {
    const u64 tg = t - flat_slots;
    g = *reinterpret_cast<const u32*>(slot_qs + tg * slot_stride);  // g = 2*query + strand
    S = sel[t];
    have = (tg - grp_off[g]) < nsel[g];  // slots behind the group's kept strings hold nothing
  }
  if (have) {
    const u32 take = S.take;
    if (take) {
      const u32 lo = S.lo, occs = S.hi - S.lo;
      const u64 out0 = hit_off[g >> 1] + S.hbase;
      HitSeed* out = seeds + out0;
      if (occs <= 4) {
        locate_in_registers<4>(f.sa + lo, occs, take, out, g, S.len, (u32)t);
        reads += occs;
      } else if (occs <= 16) {
        locate_in_registers<16>(f.sa + lo, occs, take, out, g, S.len, (u32)t);
        reads += occs;
      } else if (jobs.big && take <= 16384) {
        bj.lo = lo;
        bj.occs = occs;
        bj.take = take;
        bj.g = g;
        bj.len = S.len;
        bj.slot = (u32)t;
        bj.out = out0;
        queue = occs <= LOC_SMALL_MAX ? 1u : 2u;
      } else {
        bj.lo = lo;
        bj.occs = occs;
        bj.take = take;
        bj.g = g;
        bj.len = S.len;
        bj.slot = (u32)t;
        bj.out = out0;
        queue = 3u;  // served by this lane, below
      }
    }
  }
  const u32 lane = threadIdx.x & 63;
  // job slots: one atomic per wavefront and list (every lane on the two list heads was what this kernel waited for)
  for (u32 which = 1; which <= 2; ++which) {
    const unsigned long long mk = __ballot(queue == which);
    if (!mk) continue;
    // the job kernels were left out of this attempt: nobody will write these strings' hits, so the verify kernel must not run
    // (the host sees the job counts and repeats the batch with the job kernels)
    if (!jobs_on && lane == (u32)__ffsll((long long)mk) - 1u) atomicOr(&ctr->overflow, 4u);
    u32 base = 0;
    if (lane == (u32)__ffsll((long long)mk) - 1u) base = atomicAdd(which == 1 ? jobs.n_small : jobs.n_big, (u32)__popcll(mk));
    base = __shfl(base, (int)__ffsll((long long)mk) - 1);
    if (queue == which) {
      const u32 j = base + (u32)__popcll(mk & ((1ULL << lane) - 1));
      if (j < jobs.cap) (which == 1 ? jobs.small : jobs.big)[j] = bj;
      else queue = 3u;  // a full list (more than 2^20 repeat-rich strings in one batch): nothing is dropped, the lane serves it
    }
  }
  if (queue == 3u) {
    // correct for any size, slow: selection by repeated minimum above the previous pick (positions are distinct).  Reached with
    // hunt -m above 16 384, with DICEY_NO_BLOCK_LOCATE, and by the strings a full job list turned away.
    u64 prev = 0;
    bool first = true;
    for (u32 i = 0; i < bj.take; ++i) {
      u32 best = 0xFFFFFFFFu;
      for (u32 j = 0; j < bj.occs; ++j) {
        const u32 x = f.sa[bj.lo + j];
        if ((first || x > prev) && x < best) best = x;
      }
      reads += bj.occs;
      seeds[bj.out + i] = HitSeed{best, bj.g, bj.len, bj.slot};
      prev = best;
      first = false;
    }
  }
  wave_add(&ctr->sa_reads[blockIdx.x & (NSHARD - 1)], reads);
}

// One WAVEFRONT per string of 25..256 occurrences (most of the queued strings on a repeat-bearing genome: 54 k of 72 k per
// 100 000 queries): the interval is read once, sorted in LDS by the wavefront alone (bitonic, no workgroup barrier to wait
// for), the first `take` values are written.  Jobs are taken in grid order: they all cost about the same.
__global__ void __launch_bounds__(64) k_locate_small(FmView f, const BigJob* jobs, const u32* job_count, u32 job_cap, HitSeed* seeds, Counters* ctr) {
  __shared__ u32 buf[LOC_SMALL_MAX];
  const u32 njobs = *job_count < job_cap ? *job_count : job_cap;
  u64 reads = 0;
  for (u32 jb = blockIdx.x; jb < njobs; jb += gridDim.x) {
    const BigJob J = jobs[jb];
    u32 n2 = 32;
    while (n2 < J.occs) n2 <<= 1;
    const u32* sa = f.sa + J.lo;
    for (u32 i = threadIdx.x; i < n2; i += 64) buf[i] = i < J.occs ? sa[i] : 0xFFFFFFFFu;
    reads += J.occs;
    __syncthreads();
    for (u32 kk = 2; kk <= n2; kk <<= 1)
      for (u32 jj = kk >> 1; jj > 0; jj >>= 1) {
        for (u32 i = threadIdx.x; i < n2; i += 64) {
          const u32 l = i ^ jj;
          if (l > i) {
            const u32 a = buf[i], b2 = buf[l];
            if ((a > b2) == ((i & kk) == 0)) {
              buf[i] = b2;
              buf[l] = a;
            }
          }
        }
        __syncthreads();
      }
    for (u32 i = threadIdx.x; i < J.take; i += 64) seeds[J.out + i] = HitSeed{buf[i], J.g, J.len, J.slot};
    __syncthreads();
  }
  if (threadIdx.x == 0 && reads) atomicAdd(&ctr->sa_reads[blockIdx.x & (NSHARD - 1)], (unsigned long long)reads);
}

// One workgroup per repeat-rich string: the `take` smallest suffix-array values of its interval, ascending.
// Radix select, one byte per pass from the top: a 256-bin histogram (LDS atomics) of the values that still match the
// prefix found so far tells which bin holds the take-th smallest value; the passes stop as soon as everything up to the end
// of that bin fits the LDS buffer (on a genome-wide repeat family that is after the first pass: positions spread over the
// whole text, so one top-byte bin holds occs/185 values).  One more pass collects those values, a bitonic sort orders
// them.  2-3 coalesced passes over the interval instead of 33 (r02: 33 ms -> see DESIGN.md on the repeat-rich genome).
__global__ void __launch_bounds__(256) k_locate_big(FmView f, const BigJob* jobs, const u32* job_count, u32 job_cap, HitSeed* seeds,
                                                    Counters* ctr, u32 topk_kmax) {
  constexpr u32 CAP = 16384;
  __shared__ u32 buf[CAP];
  __shared__ u32 hist[256];
  __shared__ u32 fill, s_prefix, s_mask, s_k, s_below, s_done;
  const u32 njobs = *job_count < job_cap ? *job_count : job_cap;
  for (u32 jb = blockIdx.x; jb < njobs; jb += gridDim.x) {
    const BigJob J = jobs[jb];
    if (f.nlev > 1 && J.take <= topk_kmax) continue;  // k_locate_topk's
    const u32* sa = f.sa + J.lo;
    // refine until at most `limit` values are left to sort: sorting costs n log^2 n, another pass over the interval does not
    const u32 limit = 2 * J.take > CAP ? CAP : (2 * J.take < 1024 ? 1024u : 2 * J.take);
    if (threadIdx.x == 0) {
      s_prefix = 0;
      s_mask = 0;
      s_k = J.take - 1;  // rank (among the values matching the prefix) of the largest value we keep
      s_below = 0;       // values smaller than every value matching the prefix
      s_done = J.occs <= limit ? 1u : 0u;  // a short interval is sorted whole
    }
    __syncthreads();
    u32 passes = 0;
    u32 upper = 0xFFFFFFFFu;  // everything <= upper is collected
    if (!s_done) {
      for (int shift = 24; shift >= 0; shift -= 8) {
        hist[threadIdx.x] = 0;
        __syncthreads();
        const u32 prefix = s_prefix, mask = s_mask;
        for (u32 i = threadIdx.x; i < J.occs; i += blockDim.x) {
          const u32 x = sa[i];
          if ((x & mask) == prefix) atomicAdd(&hist[(x >> shift) & 255u], 1u);
        }
        ++passes;
        __syncthreads();
        if (threadIdx.x == 0) {
          u32 k = s_k, cum = 0, bin = 0;
          for (; bin < 256; ++bin) {
            if (k < cum + hist[bin]) break;
            cum += hist[bin];
          }
          s_k = k - cum;
          s_below += cum;
          s_prefix = prefix | (bin << shift);
          s_mask = mask | (255u << shift);
          if (s_below + hist[bin] <= limit || shift == 0) s_done = (u32)shift + 1;  // remember where we stopped
        }
        __syncthreads();
        if (s_done) {
          const u32 sh = s_done - 1;
          upper = s_prefix | (sh ? ((1u << sh) - 1) : 0u);
          break;
        }
      }
    }
    __syncthreads();
    if (threadIdx.x == 0) fill = 0;
    __syncthreads();
    for (u32 i = threadIdx.x; i < J.occs; i += blockDim.x) {
      const u32 x = sa[i];
      if (x <= upper) {
        const u32 at = atomicAdd(&fill, 1u);
        if (at < CAP) buf[at] = x;
      }
    }
    ++passes;
    __syncthreads();
    const u32 have = fill < CAP ? fill : CAP;  // >= take by construction
    u32 n2 = 1;
    while (n2 < have) n2 <<= 1;
    for (u32 i = have + threadIdx.x; i < n2; i += blockDim.x) buf[i] = 0xFFFFFFFFu;
    __syncthreads();
    for (u32 kk = 2; kk <= n2; kk <<= 1)
      for (u32 j = kk >> 1; j > 0; j >>= 1) {
        for (u32 i = threadIdx.x; i < n2; i += blockDim.x) {
          u32 l = i ^ j;
          if (l > i) {
            u32 a = buf[i], b2 = buf[l];
            bool up = (i & kk) == 0;
            if ((a > b2) == up) {
              buf[i] = b2;
              buf[l] = a;
            }
          }
        }
        __syncthreads();
      }
    for (u32 i = threadIdx.x; i < J.take; i += blockDim.x) seeds[J.out + i] = HitSeed{buf[i], J.g, J.len, J.slot};
    if (threadIdx.x == 0) atomicAdd(&ctr->sa_reads[blockIdx.x & (NSHARD - 1)], (unsigned long long)passes * J.occs);
    __syncthreads();
  }
}

// One workgroup per repeat-rich string, without reading its interval (r03).  hunter.h:355-357 keeps the first `take` entries of
// the sorted position list; r02's k_locate_big found them with 2-3 passes over the whole interval (an Alu-like string: 1.1 M
// entries = 13 MB per string, 5.8 of the 9.4 ms of a step on the repeat-bearing genome).  Here the interval is cut into the
// aligned blocks of FmView::samin (fan-out 8) and walked from the coarsest level that fits the LDS buffer down to the entries:
//   invariant   the k smallest values of a set that is partitioned into blocks lie in the k blocks with the smallest minima
//               (a value v in any other block b has the k minima of those blocks below min(b) <= v);
//   per level   the candidates' minima sit in LDS; a radix select over them (256-bin histograms, LDS atomics) gives a threshold
//               T with k <= #(minima <= T) <= KCAP; the blocks under T are expanded into their eight children (two 16-byte
//               loads each), plus the < 8 blocks of the finer level that stick out at either end of the interval;
//   entries     the select is carried on to the exact k-th value, the k survivors are sorted (bitonic) and written.
// Reads: at most the top level's blocks (<= 9 232) and 8 * KCAP + 14 words per level below, whatever the interval holds.
static constexpr u32 TOPK_KCAP = 1152;         // blocks kept per level: k plus slack, so that one histogram pass usually decides
static constexpr u32 TOPK_PAD = 0xFFFFFFFFu;
template <u32 KC>
struct TopkLdsT {
  u32 val[8 * KC + 16];
  u32 cidx[2][KC];
  u32 eidx[16];
  u32 hist[256];
  u32 wsum[4];
  u32 sh[4];
  u32 n_kept, job;
};
// Bitonic sort of n2 keys (n2 a power of two <= 1024; keys behind n2 must be the type's maximum) by a 256-lane workgroup with four
// keys per lane in registers: key i lives in lane i / 4.  Partners at distance 1-2 are in the same lane, at distance 4-128 in the
// same wavefront (one shuffle), only distances 256 and 512 cross wavefronts through LDS (xbuf: 1 024 keys) — 3 barrier rounds for
// 1 024 keys where the compare-exchange-in-LDS form had 55.  v[r] = key 4 tid + r, in and out.
template <class T>
DG_DEV T shfl_xor_key(T x, int m);
template <>
DG_DEV u32 shfl_xor_key<u32>(u32 x, int m) { return (u32)__shfl_xor((int)x, m); }
template <>
DG_DEV u64 shfl_xor_key<u64>(u64 x, int m) { return (u64)__shfl_xor((unsigned long long)x, m); }
template <class T>
DG_DEV void block_sort4(T* xbuf, u32 n2, T (&v)[4]) {
  const u32 i0 = threadIdx.x * 4;
  for (u32 kk = 2; kk <= n2; kk <<= 1) {
    const bool up = (i0 & kk) == 0;  // kk >= 4: the same for the lane's four keys; kk == 2 is handled per pair below
    for (u32 jj = kk >> 1; jj > 0; jj >>= 1) {
      if (jj >= 256) {
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 4; ++r) xbuf[i0 + r] = v[r];
        __syncthreads();
        const bool keep_min = ((i0 & jj) == 0) == up;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const T o = xbuf[(i0 + r) ^ jj];
          v[r] = keep_min ? (v[r] < o ? v[r] : o) : (v[r] < o ? o : v[r]);
        }
      } else if (jj >= 4) {
        const bool keep_min = ((i0 & jj) == 0) == up;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const T o = shfl_xor_key<T>(v[r], (int)(jj >> 2));
          v[r] = keep_min ? (v[r] < o ? v[r] : o) : (v[r] < o ? o : v[r]);
        }
      } else {
#pragma unroll
        for (int a = 0; a < 4; ++a) {
          const int b2 = a ^ (int)jj;
          if (b2 > a && (jj == 1 || jj == 2)) {
            const bool upp = kk == 2 ? ((a & 2) == 0) : up;  // (i & kk) == 0 for i = i0 + a
            const T x = v[a], y = v[b2], mn = x < y ? x : y, mx = x < y ? y : x;
            v[a] = upp ? mn : mx;
            v[b2] = upp ? mx : mn;
          }
        }
      }
    }
  }
}
// threshold T with k <= #(val <= T) <= limit (k <= limit < nv; limit == k: the exact k-th smallest).  All 256 lanes call it.
template <class LDS>
DG_DEV u32 topk_threshold(LDS& S, u32 nv, u32 k, u32 limit) {
  const u32 lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  u32 prefix = 0, mask = 0, kk = k - 1, below = 0;
  for (int shift = 24;; shift -= 8) {
    S.hist[threadIdx.x] = 0;
    __syncthreads();
    for (u32 i = threadIdx.x; i < nv; i += 256) {
      const u32 x = S.val[i];
      if ((x & mask) == prefix) atomicAdd(&S.hist[(x >> shift) & 255u], 1u);
    }
    __syncthreads();
    const u32 h = S.hist[threadIdx.x];
    u32 incl = h;
    for (int off = 1; off < 64; off <<= 1) {
      const u32 v = __shfl_up(incl, off);
      if ((int)lane >= off) incl += v;
    }
    if (lane == 63) S.wsum[wave] = incl;
    __syncthreads();
    for (u32 w = 0; w < wave; ++w) incl += S.wsum[w];
    const u32 excl = incl - h;
    if (excl <= kk && kk < incl) {  // exactly one lane: the bin that holds the k-th smallest value
      S.sh[0] = threadIdx.x;
      S.sh[1] = excl;
      S.sh[2] = h;
    }
    __syncthreads();
    const u32 bin = S.sh[0], ex = S.sh[1], cnt = S.sh[2];
    prefix |= bin << shift;
    mask |= 255u << shift;
    if (below + ex + cnt <= limit || shift == 0) return prefix | (shift ? (1u << shift) - 1u : 0u);
    below += ex;
    kk -= ex;
  }
}
// KC = TOPK_KCAP: any interval (walks the hierarchy).  KC = TOPK_KCAP_MID (r03): intervals that fit the smaller buffer whole (level 0
// only, no expansion) — 24 instead of 46 KB of LDS, six instead of three workgroups per CU; on the repeats genome two thirds of the
// 17 000 jobs of a step are of that kind.  Both walk the same job list and skip what belongs to the other (occ_lo < occs <= occ_hi).
static constexpr u32 TOPK_KCAP_MID = 576;
template <u32 KC>
__global__ void __launch_bounds__(256) k_locate_topk(FmView f, const BigJob* jobs, const u32* job_count, u32 job_cap, u32* next_job,
                                                     HitSeed* seeds, Counters* ctr, u32 occ_lo, u32 occ_hi) {
  constexpr u32 VMAXT = 8 * KC + 16;
  __shared__ TopkLdsT<KC> S;
  const u32 njobs = *job_count < job_cap ? *job_count : job_cap;
  const u32 lane = threadIdx.x & 63;
  u64 reads = 0;
  // jobs: the first gridDim.x by workgroup number, the rest from a counter (an empty list costs no atomic: 768 workgroups on one
  // word were 8 of the 10.7 us this kernel took on a batch without repeat-rich strings)
  for (u32 round = 0;; ++round) {
    __syncthreads();  // the previous job's buffers are free
    if (threadIdx.x == 0) S.job = round == 0 ? blockIdx.x : gridDim.x + atomicAdd(next_job, 1u);
    __syncthreads();
    const u32 jb = S.job;
    if (jb >= njobs) break;
    const BigJob J = jobs[jb];
    if (J.take > TOPK_KMAX || J.occs <= occ_lo || J.occs > occ_hi) continue;  // k_locate_big's / the other buffer size's
    const u32 k = J.take;
    const u64 lo = J.lo, hi = (u64)J.lo + J.occs;
    // full blocks of level j inside [lo, hi): [A(j), B(j))
    auto A = [&](int j) -> u64 { return (lo + ((1ULL << (3 * j)) - 1)) >> (3 * j); };
    auto B = [&](int j) -> u64 { return hi >> (3 * j); };
    auto N = [&](int j) -> u64 { return B(j) > A(j) ? B(j) - A(j) : 0ULL; };
    int L = 0;
    while (L + 1 < (int)f.nlev && N(L) > VMAXT - 16) ++L;  // a level left with more than 9 216 blocks has > 1 000 in the next
    if (N(L) > VMAXT - 16) continue;  // cannot happen: the top level of FmView::samin holds <= 64 blocks
    u32 nv = (u32)N(L);
    {
      const u32* src = f.samin[L] + A(L);
      for (u32 i = threadIdx.x; i < nv; i += 256) S.val[i] = src[i];
      reads += nv;
    }
    u32 nc_prev = 0;
    int cur = 0;
    bool top = true;
    __syncthreads();
    for (int j = L;; --j) {
      // at the entries: up to 96 values more than asked for may survive (the sort drops them) — an exact k-th value costs the
      // radix select all four byte passes, a little slack usually ends it after two
      const u32 limit = j == 0 ? (k + 96 < TOPK_KMAX ? k + 96 : (k > TOPK_KMAX ? k : TOPK_KMAX)) : KC;
      const u32 T = nv > limit ? topk_threshold(S, nv, k, limit) : 0xFFFFFFFEu;
      if (threadIdx.x == 0) S.n_kept = 0;
      __syncthreads();
      if (j == 0) {  // the survivors are the answer: collect, sort, write
        u32* buf = &S.cidx[0][0];
        for (u32 base = 0; base < nv; base += 256) {
          const u32 p = base + threadIdx.x;
          const u32 x = p < nv ? S.val[p] : TOPK_PAD;
          const bool keep = x <= T && x != TOPK_PAD;
          const unsigned long long mk = __ballot(keep);
          u32 at = 0;
          if (lane == 0 && mk) at = atomicAdd(&S.n_kept, (u32)__popcll(mk));
          at = __shfl(at, 0);
          if (keep) buf[at + (u32)__popcll(mk & ((1ULL << lane) - 1))] = x;
        }
        __syncthreads();
        const u32 have = S.n_kept;  // k .. k + 96 values (fewer when the whole interval is shorter); the k smallest are written
        u32 n2 = 4;
        while (n2 < have) n2 <<= 1;
        u32 sv[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) sv[r] = threadIdx.x * 4 + r < have ? buf[threadIdx.x * 4 + r] : TOPK_PAD;
        block_sort4<u32>(buf, n2, sv);
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (threadIdx.x * 4 + r < k) seeds[J.out + threadIdx.x * 4 + r] = HitSeed{sv[r], J.g, J.len, J.slot};
        break;
      }
      // blocks of level j under the threshold -> cidx[cur ^ 1]
      for (u32 base = 0; base < nv; base += 256) {
        const u32 p = base + threadIdx.x;
        const u32 x = p < nv ? S.val[p] : TOPK_PAD;
        const bool keep = x <= T && x != TOPK_PAD;
        u32 idx = 0;
        if (keep) idx = top ? (u32)(A(L) + p) : (p < 8 * nc_prev ? S.cidx[cur][p >> 3] * 8u + (p & 7u) : S.eidx[p - 8 * nc_prev]);
        const unsigned long long mk = __ballot(keep);
        u32 at = 0;
        if (lane == 0 && mk) at = atomicAdd(&S.n_kept, (u32)__popcll(mk));
        at = __shfl(at, 0);
        if (keep) S.cidx[cur ^ 1][at + (u32)__popcll(mk & ((1ULL << lane) - 1))] = idx;
      }
      __syncthreads();
      const u32 nc = S.n_kept;
      cur ^= 1;
      top = false;
      // their children, and the blocks of level j-1 that stick out at either end of the interval
      const u32* lv = f.samin[j - 1];
      const u64 nlow = j - 1 == 0 ? f.n : ~0ULL;  // level 0 is the suffix array itself: nothing beyond n
      for (u32 i = threadIdx.x; i < nc; i += 256) {
        const u64 c8 = (u64)S.cidx[cur][i] * 8;
        const uint4 x = *reinterpret_cast<const uint4*>(lv + c8), y = *reinterpret_cast<const uint4*>(lv + c8 + 4);
        u32 v[8] = {x.x, x.y, x.z, x.w, y.x, y.y, y.z, y.w};
#pragma unroll
        for (int t = 0; t < 8; ++t) S.val[8 * i + t] = c8 + t < nlow ? v[t] : TOPK_PAD;
      }
      const u64 a1 = A(j), b1 = B(j), a0 = A(j - 1), b0 = B(j - 1);
      const u32 nl = (u32)(8 * a1 - a0), nr = (u32)(b0 - 8 * b1);
      if (threadIdx.x < nl + nr) {
        const u64 e = threadIdx.x < nl ? a0 + threadIdx.x : 8 * b1 + (threadIdx.x - nl);
        S.eidx[threadIdx.x] = (u32)e;
        S.val[8 * nc + threadIdx.x] = lv[e];
      }
      reads += 8ULL * nc + nl + nr;
      nv = 8 * nc + nl + nr;
      nc_prev = nc;
      __syncthreads();
    }
  }
  if (threadIdx.x == 0 && reads) atomicAdd(&ctr->sa_reads[blockIdx.x & (NSHARD - 1)], (unsigned long long)reads);
}

// ------------------------------------------------------------------------------------------------------------
// Verify: one lane per hit.
struct VerifyArgs {
  const HitSeed* seeds;
  const u64* nhits;  // on the device: hit_off[nq]
  u64 hit_cap;
  const u64* cum;  // cum[r] = sum of seqlen[0..r)
  u32 nseq;
  dg_hit* hits;
  char* refalign;    // scratch rows of the full-matrix kernels (k_verify, k_verify_long); k_rows_to_ops turns them into ops
  char* queryalign;
  u32 stride;
  u32* ops;          // [nhits * ops_per_hit] compact alignment description (ALN_OP_NONE = unused)
  u32 ops_per_hit;   // the batch's largest effective distance
  u32 debug;              // DICEY_DBG_VERIFY (measurements only: 1 = skip the alignments, 2 = skip the context reads; results are wrong)
  u32* chits;             // != nullptr: compact records (dicey_gpu.h ABI 5: position, meta, ops) instead of dg_hit + ops
};
// the compact record's second word (dicey_gpu.h DG_CHIT_*): delta = DnaHit::start - 1 - (position - start of its sequence)
DG_DEV u32 chit_meta(int score, u32 strand, int delta, u32 aln_len) {
  return ((u32)(-score) & 15u) | ((strand & 1u) << 4) | (((u32)(delta + 32) & 127u) << 5) | (aln_len << 16);
}

// SMALL = true (all queries of the batch <= 32 nt): the DP row lives in registers (columns fully unrolled) and each
// row's trace is one 64-bit word (2 bits per column 1..32; column 0 is implied: vertical below the origin).
// SMALL: queries of at most NCOLS (24 or 32) characters: score row, query and window live in registers.
template <u32 TRACE_WORDS, bool SMALL, int NCOLS = 32>
__global__ void __launch_bounds__(256) k_verify(FmView f, Batch b, VerifyArgs a, Counters* ctr) {
  u64 h = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  const u64 nh = *a.nhits;
  if (ctr->overflow || nh > a.hit_cap || h >= nh) return;
  const HitSeed sd = a.seeds[h];
  const u64 q = sd.qs >> 1;
  const u32 strand = sd.qs & 1;
  const u8* qseq = (strand ? b.rv : b.fw) + b.qoff[q];
  const u32 n = b.qlen[q];  // columns: the query
  const u64 loc = sd.pos;
  const u32 mlen = sd.len;
  // hunter.h:358-362: text position -> (refIndex, chrpos)
  u32 lo_r = 0, hi_r = a.nseq - 1;
  while (lo_r < hi_r) {  // largest r with cum[r] <= loc, capped at nseq-1
    u32 mid = (lo_r + hi_r + 1) >> 1;
    if (a.cum[mid] <= loc) lo_r = mid;
    else hi_r = mid - 1;
  }
  const u32 ref = lo_r;
  u32 chrpos = (u32)(loc - a.cum[ref]);
  // hunter.h:363-378: context, clipped to the text, cut at sequence separators
  u64 pre = b.indel ? b.qdist[q] : 0, post = pre;
  if (pre > loc) pre = loc;
  if (loc + mlen + post > f.n) post = f.n - loc - mlen;
  u32 pre_eff = 0;
  for (u32 i = 1; i <= pre; ++i) {
    if (f.text[loc - i] == '\n') break;
    pre_eff = i;
  }
  u32 post_eff = 0;
  for (u32 i = 0; i < post; ++i) {
    if (f.text[loc + mlen + i] == '\n') break;
    post_eff = i + 1;
  }
  const u8* g = f.text + (loc - pre_eff);  // genomicseq
  const u32 mg = pre_eff + mlen + post_eff;  // rows
  if (pre_eff < chrpos) chrpos -= pre_eff;   // hunter.h:382 (strict <)
  char* ra = a.refalign + h * a.stride;
  char* qa = a.queryalign + h * a.stride;
  dg_hit out;
  out.chr = ref;
  out.query = (u32)q;
  out.strand = strand ? '-' : '+';
  out.reserved = 0;
  atomicAdd(&ctr->win_bytes[blockIdx.x & (NSHARD - 1)], (unsigned long long)(pre + mlen + post));
  if (!b.indel) {
    // hunter.h:79-88,404-405: score = -(mismatches), alignment rows are the raw strings
    int sc = 0;
    u32 k = mg < n ? mg : n;
    for (u32 i = 0; i < k; ++i) sc -= (g[i] != ascii_of(qseq[i]));
    for (u32 i = 0; i < mg; ++i) ra[i] = (char)g[i];
    for (u32 i = 0; i < n; ++i) qa[i] = (char)ascii_of(qseq[i]);
    out.score = sc;
    out.start = chrpos + 1;
    out.aln_len = (u16)(mg > n ? mg : n);  // both rows have the same length here (mg == n)
    a.hits[h] = out;
    return;
  }
  // needle.h:59-138 with AlignConfig<false,true> and DnaScore(0,-1,-1,-1) (hunter.h:383-389):
  // horizontal (gap in the reference row) costs 1 everywhere; vertical (gap in the query row) is free in
  // column 0 and column n; ties prefer horizontal, then vertical, then diagonal.
  u32 tl = 0;
  const u32 S = a.stride;
  if (SMALL) {
    constexpr int NC = NCOLS;
    int s[NC + 1];
    u8 qc[NC];
    u64 tr[NC + 3 * DMAX + 2];
#pragma unroll
    for (int c = 0; c < NC; ++c) qc[c] = (u32)c < n ? ascii_of(qseq[c]) : 0;
    // The window (<= NC + 3*DMAX bytes) and the query are packed into registers once, eight characters per word: the DP
    // rows and the row-writing pass below then take their characters with shifts instead of one dependent load each.
    constexpr int GW = (NC + 3 * DMAX + 7) / 8;
    constexpr int QW = (NC + 7) / 8;
    u64 gw[GW], qw[QW];
#pragma unroll
    for (int w = 0; w < GW; ++w) {
      u64 v = 0;
#pragma unroll
      for (int k = 0; k < 8; ++k)
        if ((u32)(8 * w + k) < mg) v |= (u64)g[8 * w + k] << (8 * k);
      gw[w] = v;
    }
#pragma unroll
    for (int w = 0; w < QW; ++w) {
      u64 v = 0;
#pragma unroll
      for (int k = 0; k < 8; ++k)
        if (8 * w + k < NC) v |= (u64)qc[8 * w + k] << (8 * k);
      qw[w] = v;
    }
    auto g_at = [&](u32 i) -> u32 {
      u64 w = gw[0];
#pragma unroll
      for (int k = 1; k < GW; ++k)
        if ((i >> 3) == (u32)k) w = gw[k];
      return (u32)(w >> (8 * (i & 7))) & 255u;
    };
    auto q_at = [&](u32 i) -> u32 {
      u64 w = qw[0];
#pragma unroll
      for (int k = 1; k < QW; ++k)
        if ((i >> 3) == (u32)k) w = qw[k];
      return (u32)(w >> (8 * (i & 7))) & 255u;
    };
#pragma unroll
    for (int c = 0; c <= NC; ++c) s[c] = -c;
    for (u32 row = 1; row <= mg; ++row) {
      const u8 gc = (u8)g_at(row - 1);
      int diag = 0;  // cell (row-1, 0); s[0] stays 0: vertical gaps are free in column 0
      u64 bits = 0;
#pragma unroll
      for (int c = 1; c <= NC; ++c) {
        if ((u32)c <= n) {
          int up = s[c];
          int dsc = diag + (gc == qc[c - 1] ? 0 : -1);
          int vsc = up + ((u32)c == n ? 0 : -1);
          int hsc = s[c - 1] - 1;
          int best = dsc > vsc ? dsc : vsc;
          best = best > hsc ? best : hsc;
          s[c] = best;
          u64 code = best == hsc ? 1ULL : (best == vsc ? 2ULL : 0ULL);
          bits |= code << (2 * (c - 1));
          diag = up;
        }
      }
      tr[row] = bits;
    }
    int fin = 0;
#pragma unroll
    for (int c = 0; c <= NC; ++c)
      if ((u32)c == n) fin = s[c];
    out.score = fin;
    // Traceback into a move stack held in registers (2 bits per column, <= 76 columns), then ONE forward pass that
    // writes the kept columns: no write-backwards / read-again / compact round trips through global memory.
    u64 mv0 = 0, mv1 = 0, mv2 = 0;  // stack of moves, most recent push at the bottom of mv0
    u32 nmv = 0, trail = 0;
    bool seen_query = false;
    u32 row = mg, col = n;
    while (row > 0 || col > 0) {
      u32 code = col == 0 ? 2u : (row == 0 ? 1u : (u32)(tr[row] >> (2 * (col - 1))) & 3u);
      if (code == 1) --col;
      else if (code == 2) --row;
      else {
        --row;
        --col;
      }
      if (code == 2 && !seen_query) ++trail;  // trailing columns whose query row is a gap (_trailGap, hunter.h:69-77)
      else seen_query = true;
      mv2 = (mv2 << 2) | (mv1 >> 62);
      mv1 = (mv1 << 2) | (mv0 >> 62);
      mv0 = (mv0 << 2) | code;
      ++nmv;
    }
    // forward: pop moves; leading query-gap columns only advance chrpos (hunter.h:391-401)
    u32 r = 0, c = 0, len = 0, lead = 0;
    bool in_lead = true;
    const u32 stop = nmv - trail;
    for (u32 k = 0; k < stop; ++k) {
      const u32 code = (u32)mv0 & 3u;
      mv0 = (mv0 >> 2) | (mv1 << 62);
      mv1 = (mv1 >> 2) | (mv2 << 62);
      mv2 >>= 2;
      char r0, r1;
      if (code == 1) {
        r0 = '-';
        r1 = (char)q_at(c);
        ++c;
      } else if (code == 2) {
        r0 = (char)g_at(r);
        r1 = '-';
        ++r;
      } else {
        r0 = (char)g_at(r);
        r1 = (char)q_at(c);
        ++r;
        ++c;
      }
      if (r1 != '-') in_lead = false;
      if (in_lead) {
        ++lead;
        continue;
      }
      ra[len] = r0;
      qa[len] = r1;
      ++len;
    }
    chrpos += lead;
    out.start = chrpos + 1;
    out.aln_len = (u16)len;
    a.hits[h] = out;
    return;
  } else {
  int s[MAX_QLEN + 1];
  u64 trace[TRACE_WORDS];  // 2 bits per cell: 1 = horizontal, 2 = vertical
  const u32 mf = n + 1;
  for (u32 w = 0; w < TRACE_WORDS; ++w) trace[w] = 0;
  auto set_tr = [&](u32 cell, u64 v) { trace[cell >> 5] |= v << ((cell & 31) * 2); };
  s[0] = 0;
  for (u32 col = 1; col <= n; ++col) {
    s[col] = -(int)col;
    set_tr(col, 1);
  }
  for (u32 row = 1; row <= mg; ++row) {
    int diag = s[0];  // cell (row-1, 0) == 0
    s[0] = 0;
    set_tr(row * mf, 2);
    const u8 gc = g[row - 1];
    for (u32 col = 1; col <= n; ++col) {
      int up = s[col];
      int dsc = diag + (gc == ascii_of(qseq[col - 1]) ? 0 : -1);
      int vsc = up + (col == n ? 0 : -1);
      int hsc = s[col - 1] - 1;
      int best = dsc > vsc ? dsc : vsc;
      best = best > hsc ? best : hsc;
      s[col] = best;
      if (best == hsc) set_tr(row * mf + col, 1);
      else if (best == vsc) set_tr(row * mf + col, 2);
      diag = up;
    }
  }
  out.score = s[n];
  // traceback, columns produced last-to-first; written from the end of the row buffers
  u32 row = mg, col = n;
  while (row > 0 || col > 0) {
    u32 cell = row * mf + col;
    u32 tr = (u32)(trace[cell >> 5] >> ((cell & 31) * 2)) & 3;
    char r0, r1;
    if (tr == 1) {
      --col;
      r0 = '-';
      r1 = (char)ascii_of(qseq[col]);
    } else if (tr == 2) {
      --row;
      r0 = (char)g[row];
      r1 = '-';
    } else {
      --row;
      --col;
      r0 = (char)g[row];
      r1 = (char)ascii_of(qseq[col]);
    }
    ++tl;
    ra[S - tl] = r0;
    qa[S - tl] = r1;
  }
  }
  // hunter.h:391-401 + _trailGap :69-77: drop leading columns whose query row is a gap (each advances chrpos)
  // and the trailing run of such columns
  const u32 base = S - tl;
  u32 lead = 0;
  while (lead < tl && qa[base + lead] == '-') ++lead;
  u32 last = tl - 1;  // _trailGap initialises lastAlignedPos to the last column
  for (u32 j = 0; j < tl; ++j)
    if (qa[base + j] != '-') last = j;
  u32 stop = last + 1;  // exclusive
  u32 len = 0;
  for (u32 j = 0; j < stop; ++j) {
    if (j < lead) continue;
    char x = ra[base + j], y = qa[base + j];
    ra[len] = x;
    qa[len] = y;
    ++len;
  }
  chrpos += lead < stop ? lead : stop;
  out.start = chrpos + 1;
  out.aln_len = (u16)len;
  a.hits[h] = out;
}

// ------------------------------------------------------------------------------------------------------------
// Verify for short queries (<= 32 nt) at distance <= 2 with a BANDED matrix.  The hit stems from a neighbourhood string within
// d operations of the query that occurs at `loc`, so the window (<= d context characters, the string, <= d context characters)
// aligns to the query with score >= -d: leading rows are free in column 0, the string costs at most d, trailing rows are free
// in column n.  On a path of score >= -d the diagonal r - c of an interior cell lies in [-d, (mg - n) + 2d]: the free leading
// rows v0 satisfy v0 + v_end = mg - n + (horizontal - vertical interior moves) <= mg - n + d, and the interior moves shift
// the diagonal by at most d either way.  Every cell of every optimal path is inside that band, its value inside the band is
// the full matrix's value (a better predecessor outside would put that predecessor on an optimal path), and a predecessor
// that ties at such a cell is itself on an optimal path — so the scores AND the reference's tie order (horizontal, then
// vertical, then diagonal; needle.h:105-131) along the traceback are those of the full matrix, with 7 (d <= 1) or 13 (d = 2)
// cells per row instead of the query length.  Storage is by diagonal: k = c - r + dm, dm = mg - n + 2d; diagonal move: same k,
// vertical: k + 1 of the previous row, horizontal: k - 1 of the same row, so one array is updated in place left to right.
// The query slides through a byte window (one character enters per row).  Window and query come in as aligned 64-bit words,
// the alignment rows leave as 64-bit words.
// The alignment is computed ONCE PER DISTINCT WINDOW, not once per hit (r03).  needle()'s result — score, rows, leading gap columns —
// is a function of (query strand, window) = (kept string, context characters left and right of it): hits of the same kept string
// differ only in their <= 2d context characters.  On a repeat-bearing genome a query has 170 hits from a handful of strings
// (1.1 M copies of an Alu-like element): 17 M hits per 100 000 queries, 2.6 of the 5.2 ms of a step in r02's lane-per-hit kernel.
// A workgroup takes 256 * CH consecutive hits (push order: the hits of one kept string are neighbours):
//   1. per hit: seed, context characters (<= 2d byte loads), '\n' trimming, chromosome lookup; the key (kept string's slot,
//      effective context lengths, context bytes) enters a hash table in LDS; the first lane to insert a key owns its class;
//   2. per class: the banded matrix with traceback (band_align below), result (score, leading gap columns, row length, <= d
//      edit columns) into LDS — the table's memory is reused for the trace;
//   3. per hit: the class's result plus the hit's own chromosome coordinate -> dg_hit + ops.
// What leaves is the COMPACT form of the alignment (ABI 4): the kept rows are the query strand's characters with at most
// |score| <= d columns that are not a match, so a hit carries `ops_per_hit` = d 32-bit words {column, kind, reference byte}
// instead of two rows of characters (68 -> 24 bytes per hit at distance 1); dg_hit_rows() rebuilds the rows.
template <int WB, typename TR>
DG_DEV AlnRes band_align(const FmView& f, const Batch& b, const HitSeed sd, TR* tr /* [row * 256] */, u8* lds_g /* 72 bytes */, u32& fault) {
  const u64 q = sd.qs >> 1;
  const u32 strand = sd.qs & 1;
  const u64 qstart = b.qoff[q];
  const u8* qseq = (strand ? b.rv : b.fw) + qstart;
  const u32 n = b.qlen[q];
  const u64 loc = sd.pos;
  const u32 mlen = sd.len;
  const u32 d = b.indel ? b.qdist[q] : 0u;
  AlnRes res;
  res.op[0] = res.op[1] = ALN_OP_NONE;
  // the whole possible window [loc - pre, loc + mlen + post) as aligned words, before its '\n' trimming is known
  u64 pre = d, post = d;
  if (pre > loc) pre = loc;
  if (loc + mlen + post > f.n) post = f.n - loc - mlen;
  constexpr int GW = (32 + 3 * 2 + 7) / 8 + 1;  // 38 bytes at any byte offset
  constexpr int QW = 32 / 8 + 1;
  u64 gw[GW], qw[QW];
  {
    const u64 g0 = loc - pre, a0 = g0 & ~7ULL;
    const u32 sh = (u32)(g0 & 7) * 8;
    const u64* src = reinterpret_cast<const u64*>(f.text + a0);
    u64 w[GW + 1];
#pragma unroll
    for (int i = 0; i <= GW; ++i) w[i] = (u32)(8 * i) < (u32)(g0 & 7) + (u32)(pre + mlen + post) ? src[i] : 0ULL;
#pragma unroll
    for (int i = 0; i < GW; ++i) gw[i] = sh ? (w[i] >> sh) | (w[i + 1] << (64 - sh)) : w[i];
    const u64 b0 = (u64)(uintptr_t)qseq, qa0 = b0 & ~7ULL;
    const u32 qsh = (u32)(b0 & 7) * 8;
    const u64* qsrc = reinterpret_cast<const u64*>((uintptr_t)qa0);
    u64 v[QW + 1];
#pragma unroll
    for (int i = 0; i <= QW; ++i) v[i] = (u32)(8 * i) < (u32)(b0 & 7) + n ? qsrc[i] : 0ULL;
#pragma unroll
    for (int i = 0; i < QW; ++i) qw[i] = qsh ? (v[i] >> qsh) | (v[i + 1] << (64 - qsh)) : v[i];
  }
  auto gw_at = [&](u32 i) -> u32 {  // byte i of the maximal window
    u64 w = gw[0];
#pragma unroll
    for (int k = 1; k < GW; ++k)
      if ((i >> 3) == (u32)k) w = gw[k];
    return (u32)(w >> (8 * (i & 7))) & 255u;
  };
  auto q_at = [&](u32 i) -> u32 {  // ASCII of query character i (i < n)
    u64 w = qw[0];
#pragma unroll
    for (int k = 1; k < QW; ++k)
      if ((i >> 3) == (u32)k) w = qw[k];
    return ascii_of((u32)(w >> (8 * (i & 7))) & 255u);
  };
  // hunter.h:363-378: the context stops at sequence separators
  u32 pre_eff = 0;
  for (u32 i = 1; i <= pre; ++i) {
    if (gw_at((u32)pre - i) == '\n') break;
    pre_eff = i;
  }
  u32 post_eff = 0;
  for (u32 i = 0; i < post; ++i) {
    if (gw_at((u32)pre + mlen + i) == '\n') break;
    post_eff = i + 1;
  }
  res.pre_eff = pre_eff;
  const u32 skip = (u32)pre - pre_eff;         // genomicseq starts at byte `skip` of the maximal window
  const u32 mg = pre_eff + mlen + post_eff;    // rows
  // genomicseq from byte 0 (gsh), the query codes with 7 = "outside" behind the last character (qwm); both also in LDS
  u64 gsh[5], qwm[QW];
#pragma unroll
  for (int i = 0; i < 5; ++i) gsh[i] = skip ? (gw[i] >> (8 * skip)) | (gw[i + 1] << (64 - 8 * skip)) : gw[i];
#pragma unroll
  for (int i = 0; i < QW; ++i) {
    const int keep = (int)n - 8 * i;  // characters of the query in this word
    const u64 km = keep >= 8 ? ~0ULL : (keep <= 0 ? 0ULL : (1ULL << (8 * keep)) - 1);
    qwm[i] = (qw[i] & km) | (0x0707070707070707ULL & ~km);
  }
  u8* const lds_q = lds_g + 40;
#pragma unroll
  for (int i = 0; i < 5; ++i) reinterpret_cast<u64*>(lds_g)[i] = gsh[i];
#pragma unroll
  for (int i = 0; i < 4; ++i) reinterpret_cast<u64*>(lds_q)[i] = qwm[i];
  constexpr u64 ASCII_LUT = 0x4E54474341ULL;  // code 0..4 -> 'A','C','G','T','N'; codes 5..7 -> 0
  auto g_ch = [&](u32 i) -> u32 { return lds_g[i]; };                                     // genomicseq[i]
  auto q_ch = [&](u32 i) -> u32 { return (u32)(ASCII_LUT >> (8 * lds_q[i])) & 255u; };    // ASCII of query character i < n
  u32 nops = 0;
  auto push_op = [&](u32 col, u32 kind, u32 byte) {
    const u32 o = aln_op(col, kind, byte);
    if (nops == 0) res.op[0] = o;
    else if (nops == 1) res.op[1] = o;
    ++nops;
  };
  if (!b.indel) {
    // hunter.h:79-88,404-405: score = -(mismatches), alignment rows are the raw strings (mg == mlen == n here)
    int sc = 0;
    const u32 k = mg < n ? mg : n;
    for (u32 i = 0; i < k; ++i) {
      const u32 gc = g_ch(i);
      if (gc != q_ch(i)) {
        --sc;
        push_op(i, DG_ALN_MISMATCH, gc);
      }
    }
    if (mg != n) fault = 1;  // a Hamming hit's window is the string itself
    res.info = ((u32)sc & 255u) | (n << 16);
    if (nops > 2) fault = 1;
    return res;
  }
  constexpr int NEG = -1000;
  const int dm = (int)mg - (int)n + 2 * (int)d;  // largest diagonal r - c kept; k = c - r + dm
  int s[WB];
#pragma unroll
  for (int k = 0; k < WB; ++k) {
    const int c = k - dm;
    s[k] = (c < 0 || c > (int)n) ? NEG : -c;
  }
  // query window of row r: byte k = q[c - 1] for c = r - dm + k (0 outside the query)
  auto qbyte = [&](int i) -> u64 { return (i >= 0 && i < (int)n) ? (u64)q_at((u32)i) : 0ULL; };
  u64 qlo = 0, qhi = 0;  // bytes 0-7 and 8-15 of the window
#pragma unroll
  for (int k = 0; k < WB; ++k) {  // row 1: c - 1 = k - dm
    const u64 v = qbyte(k - dm);
    if (k < 8) qlo |= v << (8 * k);
    else qhi |= v << (8 * (k - 8));
  }
  // the character that enters the window after row r is q[r + WB - 1 - dm]: the query codes shifted left by WB - dm bytes
  // put it at byte (r - 1), so rows take both their characters from the bottom of two shift registers, a word per 8 rows
  u64 qs[5];
  {
    const u32 off = (u32)((int)WB - dm), ws = off >> 3, bs = (off & 7) * 8;  // 2..11 bytes
    constexpr u64 PAD = 0x0707070707070707ULL;
#pragma unroll
    for (int w = 0; w < 5; ++w) {
      const u64 x0 = w < QW ? qwm[w < QW ? w : 0] : PAD, x1 = w + 1 < QW ? qwm[w + 1 < QW ? w + 1 : 0] : PAD,
                x2 = w + 2 < QW ? qwm[w + 2 < QW ? w + 2 : 0] : PAD;
      const u64 lo = ws ? x1 : x0, hi = ws ? x2 : x1;
      qs[w] = bs ? (lo >> bs) | (hi << (64 - bs)) : lo;
    }
  }
#pragma unroll
  for (int w = 0; w < 5; ++w) {
    u64 gcur = gsh[w], qcur = qs[w];
    const u32 rend = mg < 8u * w + 8u ? mg : 8u * w + 8u;
    for (u32 row = 8u * w + 1; row <= rend; ++row) {
      const u32 gc = (u32)gcur & 255u;
      gcur >>= 8;
      const int c0 = (int)row - dm;
      u32 bits = 0;
      int left = NEG;
#pragma unroll
      for (int k = 0; k < WB; ++k) {
        const int c = c0 + k;
        const u32 qc = (u32)((k < 8 ? qlo >> (8 * k) : qhi >> (8 * (k - 8))) & 255u);
        const int up = k + 1 < WB ? s[k + 1] : NEG;
        const int dsc = s[k] + (gc == qc ? 0 : -1);
        const int vsc = up + (c == (int)n ? 0 : -1);
        const int hsc = left - 1;
        int best = dsc > vsc ? dsc : vsc;
        best = best > hsc ? best : hsc;
        const u32 code = best == hsc ? 1u : (best == vsc ? 2u : 0u);
        const int val = c < 0 ? NEG : (c == 0 ? 0 : (c > (int)n ? NEG : best));
        s[k] = val;
        left = val;
        bits |= code << (2 * k);
      }
      tr[row * 256] = (TR)bits;
      // slide the query window: drop byte 0, the character of column c0 + WB (next row's last diagonal) enters at the top
      const u64 nb = (ASCII_LUT >> (8 * ((u32)qcur & 255u))) & 255u;
      qcur >>= 8;
      qlo = (qlo >> 8) | (qhi << 56);
      qhi >>= 8;
      if (WB <= 8) qlo |= nb << (8 * (WB - 1));
      else qhi |= nb << (8 * (WB - 9));
    }
  }
  int fin = NEG;
#pragma unroll
  for (int k = 0; k < WB; ++k)
    if (k == 2 * (int)d) fin = s[k];  // cell (mg, n)
  // traceback into a move stack held in registers, then one forward pass over the kept columns
  u64 mv0 = 0, mv1 = 0, mv2 = 0;
  u32 nmv = 0, trail = 0;
  bool seen_query = false;
  u32 row = mg, col = n;
  while (row > 0 || col > 0) {
    u32 code;
    if (col == 0) code = 2u;
    else if (row == 0) code = 1u;
    else {
      const int k = (int)col - (int)row + dm;  // inside the band on every optimal path
      code = (k >= 0 && k < WB) ? ((u32)tr[row * 256] >> (2 * k)) & 3u : 1u;
    }
    if (code == 1) --col;
    else if (code == 2) --row;
    else {
      --row;
      --col;
    }
    if (code == 2 && !seen_query) ++trail;  // trailing columns whose query row is a gap (_trailGap, hunter.h:69-77)
    else seen_query = true;
    mv2 = (mv2 << 2) | (mv1 >> 62);
    mv1 = (mv1 << 2) | (mv0 >> 62);
    mv0 = (mv0 << 2) | code;
    ++nmv;
  }
  u32 r = 0, c = 0, len = 0, lead = 0;
  bool in_lead = true;
  const u32 stop = nmv - trail;
  for (u32 k = 0; k < stop; ++k) {
    const u32 code = (u32)mv0 & 3u;
    mv0 = (mv0 >> 2) | (mv1 << 62);
    mv1 = (mv1 >> 2) | (mv2 << 62);
    mv2 >>= 2;
    if (code == 1) {  // gap in the reference row
      in_lead = false;
      push_op(len, DG_ALN_REF_GAP, 0);
      ++c;
      ++len;
    } else if (code == 2) {  // gap in the query row; leading ones only advance chrpos (hunter.h:391-401)
      const u32 gc = g_ch(r);
      ++r;
      if (in_lead) ++lead;
      else {
        push_op(len, DG_ALN_QUERY_GAP, gc);
        ++len;
      }
    } else {
      in_lead = false;
      const u32 gc = g_ch(r), qc = q_ch(c);
      if (gc != qc) push_op(len, DG_ALN_MISMATCH, gc);
      ++r;
      ++c;
      ++len;
    }
  }
  // every kept column that is not a match costs one (free vertical moves exist only in columns 0 and n, and those are the
  // stripped ones): more operations than the score allows, or a score below -d, would contradict the band's premise
  if (nops > 2 || (int)nops != -fin || fin < -(int)d) fault = 1;
  res.info = ((u32)fin & 255u) | (lead << 8) | (len << 16);
  return res;
}

// Dynamic LDS: max(hash table, rows * 256 trace words + 6 * 256 window words), rows = maxlen + 3 d + 2 of the batch.
template <int WB, int CH>
DG_DEV void verify_memo_block(const FmView& f, const Batch& b, const VerifyArgs& a, Counters* ctr, u32 rows) {
  using TR = typename std::conditional<(WB <= 8), u16, u32>::type;
  constexpr u32 NH = 256u * CH;        // hits of a workgroup
  constexpr u32 HCAP = 2 * NH;         // hash slots (a power of two)
  constexpr u32 HSHIFT = 64 - (CH == 1 ? 9 : CH == 2 ? 10 : CH == 4 ? 11 : 12);
  static_assert(CH == 1 || CH == 2 || CH == 4 || CH == 8, "hits per lane");
  constexpr u32 DS = WB <= 8 ? 1 : 2;  // operations per class
  extern __shared__ __align__(16) u8 u_lds[];  // phase 1: hash keys + values; phase 2: trace + window / query bytes
  __shared__ u32 cls_info[NH];
  __shared__ u32 cls_ops[NH * DS];
  __shared__ u16 cls_owner[NH];
  __shared__ u32 s_ncls, s_fault;
  u64* const hkey = reinterpret_cast<u64*>(u_lds);
  u16* const hval = reinterpret_cast<u16*>(u_lds + HCAP * 8);
  constexpr u64 EMPTY = ~0ULL;
  const u64 nh = *a.nhits;
  if (ctr->overflow || nh > a.hit_cap) return;
  const u64 base = (u64)blockIdx.x * NH;
  if (base >= nh) return;
  const u32 tid = threadIdx.x;
  __shared__ u64 s_cum[512];
  const bool cum_in_lds = CH > 1 && a.nseq <= 512;
  if (cum_in_lds)
    for (u32 i = tid; i < a.nseq; i += 256) s_cum[i] = a.cum[i];
  constexpr bool SHARE = CH > 1;  // CH == 1: no table, every hit is aligned by its own lane (batches with a handful of hits per query)
  if (SHARE)
    for (u32 i = tid; i < HCAP; i += 256) hkey[i] = EMPTY;
  if (tid == 0) {
    s_ncls = 0;
    s_fault = 0;
  }
  // ---- phase 1: per hit
  HitSeed sd[CH];
  u32 dq[CH];
#pragma unroll
  for (int j = 0; j < CH; ++j) {
    const u64 h = base + (u32)j * 256u + tid;
    const uint4 v = h < nh ? *reinterpret_cast<const uint4*>(a.seeds + h) : make_uint4(0, 0, 0, 0);
    sd[j] = HitSeed{v.x, v.y, v.z, v.w};
  }
#pragma unroll
  for (int j = 0; j < CH; ++j) dq[j] = b.indel ? b.qdist[sd[j].qs >> 1] : 0u;
  u32 fl[CH];  // context bytes: left of the string at bits 0-15 (nearest first), right of it at bits 16-31
#pragma unroll
  for (int j = 0; j < CH; ++j) {
    const u64 h = base + (u32)j * 256u + tid;
    const u64 loc = sd[j].pos, endp = loc + sd[j].len;
    const u32 d = dq[j];
    u32 x = 0;
    if (SHARE && h < nh && !(a.debug & 2u)) {  // (the lane-per-hit path takes its context from the window band_align loads anyway)
      if (d >= 1 && loc >= 1) x |= (u32)f.text[loc - 1];
      if (DS >= 2 && d >= 2 && loc >= 2) x |= (u32)f.text[loc - 2] << 8;
      if (d >= 1 && endp + 1 <= f.n) x |= (u32)f.text[endp] << 16;
      if (DS >= 2 && d >= 2 && endp + 2 <= f.n) x |= (u32)f.text[endp + 1] << 24;
    }
    fl[j] = x;
  }
  if (SHARE) __syncthreads();  // the table is clear (the lane-per-hit path has no barrier at all: its lanes share nothing)
  u32 ref[CH], cpos[CH], slot[CH];
  bool won[CH];
  u64 wbytes = 0;
#pragma unroll
  for (int j = 0; j < CH; ++j) {
    const u64 h = base + (u32)j * 256u + tid;
    won[j] = false;
    slot[j] = 0;
    ref[j] = cpos[j] = 0;
    if (h >= nh) continue;
    const u64 loc = sd[j].pos, endp = loc + sd[j].len;
    const u32 d = dq[j];
    // hunter.h:358-362: text position -> (refIndex, chrpos); the sequence starts sit in LDS when there are at most 512 of them
    // (GRCh38: 194; the table is filled in front of the barrier the hash table needs anyway)
    u32 lo_r = 0, hi_r = a.nseq - 1;
    if (cum_in_lds) {
      while (lo_r < hi_r) {
        const u32 mid = (lo_r + hi_r + 1) >> 1;
        if (s_cum[mid] <= loc) lo_r = mid;
        else hi_r = mid - 1;
      }
    } else {
      while (lo_r < hi_r) {
        const u32 mid = (lo_r + hi_r + 1) >> 1;
        if (a.cum[mid] <= loc) lo_r = mid;
        else hi_r = mid - 1;
      }
    }
    ref[j] = lo_r;
    u32 chrpos = (u32)(loc - (cum_in_lds ? s_cum[lo_r] : a.cum[lo_r]));
    // hunter.h:363-378: <= d context characters either side, clipped to the text, cut at sequence separators
    u32 pre = d, post = d;
    if (pre > loc) pre = (u32)loc;
    if (endp + post > f.n) post = (u32)(f.n - endp);
    u32 pre_eff = 0, post_eff = 0, pb = 0, qb = 0;
#pragma unroll
    for (u32 i = 0; i < DS; ++i) {
      const u32 ch = (fl[j] >> (8 * i)) & 255u;
      if (i < pre && pre_eff == i && ch != '\n') {
        pre_eff = i + 1;
        pb |= ch << (8 * i);
      }
      const u32 ch2 = (fl[j] >> (16 + 8 * i)) & 255u;
      if (i < post && post_eff == i && ch2 != '\n') {
        post_eff = i + 1;
        qb |= ch2 << (8 * i);
      }
    }
    if (SHARE && pre_eff < chrpos) chrpos -= pre_eff;  // hunter.h:382 (strict <); lane-per-hit path: behind its alignment, below
    cpos[j] = chrpos;
    wbytes += pre + sd[j].len + post;
    const u32 local = (u32)j * 256u + tid;
    if (!SHARE) continue;
    // a class = (kept string, effective context lengths, context bytes); slots beyond 2^28 stay classes of their own
    const u64 key = sd[j].sel < (1u << 28) ? ((u64)(sd[j].sel | (pre_eff << 28) | (post_eff << 30)) << 32) | (pb << 16) | qb
                                           : ((u64)(0xC0000000u | local) << 32);
    u32 sidx = (u32)((key * 0x9E3779B97F4A7C15ULL) >> HSHIFT);
    for (;;) {
      const u64 old = atomicCAS(reinterpret_cast<unsigned long long*>(&hkey[sidx]), (unsigned long long)EMPTY, (unsigned long long)key);
      if (old == EMPTY) {
        won[j] = true;
        break;
      }
      if (old == key) break;
      sidx = (sidx + 1) & (HCAP - 1);
    }
    slot[j] = sidx;
  }
  wave_add(&ctr->win_bytes[blockIdx.x & (NSHARD - 1)], wbytes);
  u32 cls[CH];
  u32 ncls = 0;
  if (SHARE) {
    __syncthreads();
#pragma unroll
    for (int j = 0; j < CH; ++j)
      if (won[j]) {
        const u32 c = atomicAdd(&s_ncls, 1u);
        hval[slot[j]] = (u16)c;
        cls_owner[c] = (u16)((u32)j * 256u + tid);
      }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < CH; ++j) cls[j] = hval[slot[j]];
    ncls = s_ncls;
  }
  if (SHARE) __syncthreads();  // the table's memory becomes the trace
  // ---- phase 2: per class
  TR* const tr = reinterpret_cast<TR*>(u_lds) + tid;
  u64* const win = reinterpret_cast<u64*>(u_lds + ((rows * 256 * sizeof(TR) + 7) & ~(size_t)7)) + tid;  // 6 words per lane, word-major
  u32 fault = 0;
  auto align = [&](const HitSeed& s0) -> AlnRes {
    const u64 q = s0.qs >> 1;
    const uint4 pq = b.gpeq[s0.qs];
    return band_align_bits<WB, TR, 256>(f.text, f.n, b.indel != 0, (u64)s0.pos, s0.len, b.qlen[q], b.indel ? b.qdist[q] : 0u,
                                        PosMasks{pq.x, pq.y, pq.z, pq.w}, tr, win, fault);
  };
  if (!SHARE) {
    cls[0] = tid;
    if (base + tid < nh && !(a.debug & 1u)) {
      const AlnRes r = align(sd[0]);
      if (r.pre_eff < cpos[0]) cpos[0] -= r.pre_eff;  // hunter.h:382 (strict <)
      cls_info[tid] = r.info;
      cls_ops[tid * DS] = r.op[0];
      if (DS > 1) cls_ops[tid * DS + 1] = r.op[1];
    }
  } else {
    for (u32 c = tid; c < ncls && !(a.debug & 1u); c += 256) {
      const u32 own = cls_owner[c];
      const uint4 v = *reinterpret_cast<const uint4*>(a.seeds + base + own);
      const AlnRes r = align(HitSeed{v.x, v.y, v.z, v.w});
      cls_info[c] = r.info;
      cls_ops[c * DS] = r.op[0];
      if (DS > 1) cls_ops[c * DS + 1] = r.op[1];
    }
  }
  if (SHARE) {
    if (fault) s_fault = 1;
    __syncthreads();
    if (s_fault) {  // never observed; fail the batch loudly rather than hand out a wrong alignment
      if (tid == 0) atomicOr(&ctr->overflow, 2u);
      return;
    }
  } else if (fault) {
    atomicOr(&ctr->overflow, 2u);
    return;
  }
  // ---- phase 3: per hit
#pragma unroll
  for (int j = 0; j < CH; ++j) {
    const u64 h = base + (u32)j * 256u + tid;
    if (h >= nh) continue;
    const u32 info = cls_info[cls[j]];
    if (a.chits) {  // compact: the host finds the sequence from the position itself
      const u32 chr0 = (u32)((u64)sd[j].pos - (cum_in_lds ? s_cum[ref[j]] : a.cum[ref[j]]));
      const u32 W = 2u + a.ops_per_hit;
      u32* rec = a.chits + h * W;
      rec[0] = sd[j].pos;
      rec[1] = chit_meta((int)(int8_t)(info & 255u), sd[j].qs & 1u, (int)(cpos[j] + ((info >> 8) & 255u)) - (int)chr0, info >> 16);
      if (a.ops_per_hit >= 1) rec[2] = cls_ops[cls[j] * DS];
      if (a.ops_per_hit >= 2) rec[3] = DS > 1 ? cls_ops[cls[j] * DS + 1] : ALN_OP_NONE;
      for (u32 k = 2; k < a.ops_per_hit; ++k) rec[2 + k] = ALN_OP_NONE;
      continue;
    }
    dg_hit out;
    out.score = (int)(int8_t)(info & 255u);
    out.chr = ref[j];
    out.start = cpos[j] + ((info >> 8) & 255u) + 1;
    out.query = sd[j].qs >> 1;
    out.aln_len = (u16)(info >> 16);
    out.strand = (sd[j].qs & 1) ? '-' : '+';
    out.reserved = 0;
    a.hits[h] = out;
    if (a.ops_per_hit >= 1) a.ops[h * a.ops_per_hit] = cls_ops[cls[j] * DS];
    if (DS > 1 && a.ops_per_hit >= 2) a.ops[h * a.ops_per_hit + 1] = cls_ops[cls[j] * DS + 1];
  }
}

template <int WB, int CH>
__global__ void __launch_bounds__(256) k_verify_memo(FmView f, Batch b, VerifyArgs a, Counters* ctr, u32 rows) {
  verify_memo_block<WB, CH>(f, b, a, ctr, rows);
}

// ------------------------------------------------------------------------------------------------------------
// Verify for queries of any length (the ones above MAX_QLEN, whose full matrix no lane could hold): the same banded matrix
// as k_verify_band (its argument does not depend on the length), 6d + 1 <= 25 diagonals in registers, the trace — one
// 64-bit word per row — in a workspace in HBM, query and window read where they lie.  One lane per hit; this is the rare
// path (a primer is 18-30 nt), built for correctness.
template <int WB>
__global__ void __launch_bounds__(64) k_verify_long(FmView f, Batch b, VerifyArgs a, Counters* ctr, u64* trace, u32 rows_cap) {
  u64 h = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  const u64 nh = *a.nhits;
  if (ctr->overflow || nh > a.hit_cap || h >= nh) return;
  const HitSeed sd = a.seeds[h];
  const u64 q = sd.qs >> 1;
  const u32 strand = sd.qs & 1;
  const u8* qseq = (strand ? b.rv : b.fw) + b.qoff[q];
  const u32 n = b.qlen[q];
  const u64 loc = sd.pos;
  const u32 mlen = sd.len;
  u32 lo_r = 0, hi_r = a.nseq - 1;
  while (lo_r < hi_r) {
    u32 mid = (lo_r + hi_r + 1) >> 1;
    if (a.cum[mid] <= loc) lo_r = mid;
    else hi_r = mid - 1;
  }
  const u32 ref = lo_r;
  u32 chrpos = (u32)(loc - a.cum[ref]);
  const u32 d = b.indel ? b.qdist[q] : 0u;
  u64 pre = d, post = d;
  if (pre > loc) pre = loc;
  if (loc + mlen + post > f.n) post = f.n - loc - mlen;
  u32 pre_eff = 0;
  for (u32 i = 1; i <= pre; ++i) {
    if (f.text[loc - i] == '\n') break;
    pre_eff = i;
  }
  u32 post_eff = 0;
  for (u32 i = 0; i < post; ++i) {
    if (f.text[loc + mlen + i] == '\n') break;
    post_eff = i + 1;
  }
  const u8* g = f.text + (loc - pre_eff);
  const u32 mg = pre_eff + mlen + post_eff;
  if (pre_eff < chrpos) chrpos -= pre_eff;  // hunter.h:382 (strict <)
  char* ra = a.refalign + h * a.stride;
  char* qa = a.queryalign + h * a.stride;
  dg_hit out;
  out.chr = ref;
  out.query = (u32)q;
  out.strand = strand ? '-' : '+';
  out.reserved = 0;
  atomicAdd(&ctr->win_bytes[blockIdx.x & (NSHARD - 1)], (unsigned long long)(pre + mlen + post));
  if (!b.indel) {  // hunter.h:79-88,404-405
    int sc = 0;
    u32 k = mg < n ? mg : n;
    for (u32 i = 0; i < k; ++i) sc -= (g[i] != ascii_of(qseq[i]));
    for (u32 i = 0; i < mg; ++i) ra[i] = (char)g[i];
    for (u32 i = 0; i < n; ++i) qa[i] = (char)ascii_of(qseq[i]);
    out.score = sc;
    out.start = chrpos + 1;
    out.aln_len = (u16)(mg > n ? mg : n);
    a.hits[h] = out;
    return;
  }
  constexpr int NEG = -100000;
  const int dm = (int)mg - (int)n + 2 * (int)d;  // k = c - r + dm, see k_verify_band
  int s[WB];
#pragma unroll
  for (int k = 0; k < WB; ++k) {
    const int c = k - dm;
    s[k] = (c < 0 || c > (int)n) ? NEG : -c;
  }
  u64* tr = trace + h * (u64)rows_cap;
  for (u32 row = 1; row <= mg; ++row) {
    const u32 gc = g[row - 1];
    const int c0 = (int)row - dm;
    u64 bits = 0;
    int left = NEG;
#pragma unroll
    for (int k = 0; k < WB; ++k) {
      const int c = c0 + k;
      const u32 qc = (c >= 1 && c <= (int)n) ? (u32)ascii_of(qseq[c - 1]) : 0u;
      const int up = k + 1 < WB ? s[k + 1] : NEG;
      const int dsc = s[k] + (gc == qc ? 0 : -1);
      const int vsc = up + (c == (int)n ? 0 : -1);
      const int hsc = left - 1;
      int best = dsc > vsc ? dsc : vsc;
      best = best > hsc ? best : hsc;
      const u64 code = best == hsc ? 1ULL : (best == vsc ? 2ULL : 0ULL);
      const int val = c < 0 ? NEG : (c == 0 ? 0 : (c > (int)n ? NEG : best));
      s[k] = val;
      left = val;
      bits |= code << (2 * k);
    }
    if (row < rows_cap) tr[row] = bits;
  }
  int fin = NEG;
#pragma unroll
  for (int k = 0; k < WB; ++k)
    if (k == 2 * (int)d) fin = s[k];
  out.score = fin;
  // traceback, columns produced last-to-first and written from the end of the row buffers
  const u32 S = a.stride;
  u32 tl = 0;
  u32 row = mg, col = n;
  while (row > 0 || col > 0) {
    u32 code;
    if (col == 0) code = 2u;
    else if (row == 0) code = 1u;
    else {
      const int k = (int)col - (int)row + dm;
      code = (k >= 0 && k < WB && row < rows_cap) ? (u32)(tr[row] >> (2 * k)) & 3u : 1u;
    }
    char r0, r1;
    if (code == 1) {
      --col;
      r0 = '-';
      r1 = (char)ascii_of(qseq[col]);
    } else if (code == 2) {
      --row;
      r0 = (char)g[row];
      r1 = '-';
    } else {
      --row;
      --col;
      r0 = (char)g[row];
      r1 = (char)ascii_of(qseq[col]);
    }
    ++tl;
    ra[S - tl] = r0;
    qa[S - tl] = r1;
  }
  // hunter.h:391-401 + _trailGap :69-77: drop leading columns whose query row is a gap (each advances chrpos) and the
  // trailing run of such columns
  const u32 base = S - tl;
  u32 lead = 0;
  while (lead < tl && qa[base + lead] == '-') ++lead;
  u32 last = tl - 1;
  for (u32 j = 0; j < tl; ++j)
    if (qa[base + j] != '-') last = j;
  const u32 stop = last + 1;
  u32 len = 0;
  for (u32 j = 0; j < stop; ++j) {
    if (j < lead) continue;
    char x = ra[base + j], y = qa[base + j];
    ra[len] = x;
    qa[len] = y;
    ++len;
  }
  chrpos += lead < stop ? lead : stop;
  out.start = chrpos + 1;
  out.aln_len = (u16)len;
  a.hits[h] = out;
}

// ------------------------------------------------------------------------------------------------------------
// Host orchestration
// ------------------------------------------------------------------------------------------------------------

// The full-matrix verify kernels (k_verify, k_verify_long: distance 3-4, queries above 32 nt) leave character rows in their scratch
// buffers; this turns a hit's rows into the compact description every consumer reads.  A column whose two rows both hold '-' cannot
// come out of needle(): it is a '-' byte of the genome over a gap of the query.
__global__ void k_rows_to_ops(VerifyArgs a, Counters* ctr) {
  const u64 h = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  const u64 nh = *a.nhits;
  if (ctr->overflow || nh > a.hit_cap || h >= nh) return;
  const char* ra = a.refalign + h * a.stride;
  const char* qa = a.queryalign + h * a.stride;
  const u32 len = a.hits[h].aln_len;
  u32 nops = 0;
  for (u32 i = 0; i < len; ++i) {
    const u32 x = (u8)ra[i], y = (u8)qa[i];
    if (x == y && x != '-') continue;
    const u32 kind = y == '-' ? (u32)DG_ALN_QUERY_GAP : (x == '-' ? (u32)DG_ALN_REF_GAP : (u32)DG_ALN_MISMATCH);
    if (nops < a.ops_per_hit) a.ops[h * a.ops_per_hit + nops] = aln_op(i, kind, kind == DG_ALN_REF_GAP ? 0u : x);
    ++nops;
  }
  for (u32 i = nops; i < a.ops_per_hit; ++i) a.ops[h * a.ops_per_hit + i] = ALN_OP_NONE;
  if (nops > a.ops_per_hit) atomicOr(&ctr->overflow, 2u);  // more edit columns than the distance allows: fail the batch loudly
}

// Fetched results leave the device as ONE block: this kernel lays the pieces (hit offsets, query offsets, the three per-query
// arrays, normalised queries, hits, operation words) out behind each other exactly as the host block holds them, and one copy
// follows (r03: six copies of 0.8-4.7 MB each paid their own start-up, 0.39 ms per 100 000 queries for 10 MB).
// compact records from dg_hit + ops, for the verify kernels that write the classic form (queries above 32 nt, distance above 2)
__global__ void __launch_bounds__(256) k_hits_to_compact(VerifyArgs a, const Counters* ctr) {
  const u64 h = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  const u64 nh = *a.nhits;
  if (ctr->overflow || nh > a.hit_cap || h >= nh) return;
  const dg_hit H = a.hits[h];
  const u32 pos = a.seeds[h].pos;
  const u32 W = 2u + a.ops_per_hit;
  u32* rec = a.chits + h * W;
  rec[0] = pos;
  rec[1] = chit_meta(H.score, H.strand == '-' ? 1u : 0u, (int)(H.start - 1) - (int)(u32)((u64)pos - a.cum[H.chr]), H.aln_len);
  for (u32 k = 0; k < a.ops_per_hit; ++k) rec[2 + k] = a.ops[h * a.ops_per_hit + k];
}

struct PackArgs {
  const u32* src[6];
  u64 dst_word[6];  // offset in the block, in 32-bit words
  u64 nwords[6];    // hits / ops: capacity; the kernel stops at the batch's hit count
  const u64* nhits;
  u32 hit_words, op_words;  // words per hit in segments 4 and 5
};
__global__ void __launch_bounds__(256) k_pack_results(PackArgs a, u32* dst) {
  const u64 nh = *a.nhits;
  u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
#pragma unroll
  for (int s = 0; s < 6; ++s) {
    u64 n = a.nwords[s];
    if (t < n) {
      if (s == 4) n = nh * a.hit_words < n ? nh * a.hit_words : n;
      if (s == 5) n = nh * a.op_words < n ? nh * a.op_words : n;
      if (t < n && a.src[s]) dst[a.dst_word[s] + t] = a.src[s][t];
      return;
    }
    t -= n;
  }
}

static inline double host_us() {
  return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
static double ev_ms(hipEvent_t a, hipEvent_t b) {
  float ms = 0;
  (void)hipEventElapsedTime(&ms, a, b);
  return ms;
}

// Two-launch scan for the production path: tiles of 1024 entries (256 lanes x 4); k_scan_tile_sums writes one sum per
// tile, k_scan_tiles adds up the sums of the tiles before its own (a few hundred values) and scans its tile in place.
constexpr u32 SCAN_TILE = 1024;
DG_DEV u64 block_sum_256(u64 v, u64* lds4) {  // sum over a 256-lane workgroup, result in every lane
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  if ((threadIdx.x & 63) == 0) lds4[threadIdx.x >> 6] = v;
  __syncthreads();
  const u64 r = lds4[0] + lds4[1] + lds4[2] + lds4[3];
  __syncthreads();
  return r;
}
// ctr != nullptr: the first workgroup also checks the search kernels' leaf / survivor regions (what k_leaf_overflow did in a launch
// of its own): the flag is set before any kernel that reads leaves starts
__global__ void __launch_bounds__(256) k_scan_tile_sums(const u32* in, u64 n, u64* sums, Counters* ctr, u32 shard_cap, u32 surv_cap) {
  __shared__ u64 lds4[4];
  if (ctr && blockIdx.x == 0)
    for (u32 k = threadIdx.x; k < NSHARD; k += 256)
      if (ctr->leaf_cnt[k] > shard_cap || ctr->surv_cnt[k] > surv_cap) atomicOr(&ctr->overflow, 1u);
  const u64 base = (u64)blockIdx.x * SCAN_TILE + threadIdx.x * 4;
  u64 s = 0;
  for (u32 k = 0; k < 4; ++k)
    if (base + k < n) s += in[base + k];
  s = block_sum_256(s, lds4);
  if (threadIdx.x == 0) sums[blockIdx.x] = s;
}
__global__ void __launch_bounds__(256) k_scan_tiles(const u32* in, u64 n, const u64* sums, u64* out /*[n+1]*/) {
  __shared__ u64 lds4[4];
  u64 before = 0;  // entries of all earlier tiles
  for (u32 k = threadIdx.x; k < blockIdx.x; k += 256) before += sums[k];
  before = block_sum_256(before, lds4);
  const u64 base = (u64)blockIdx.x * SCAN_TILE + threadIdx.x * 4;
  u32 v[4];
  u64 mine = 0;
  for (u32 k = 0; k < 4; ++k) {
    v[k] = base + k < n ? in[base + k] : 0u;
    mine += v[k];
  }
  u64 incl = mine;  // inclusive scan of the lane totals: inside the wavefront by shuffles, across the four through LDS
  const u32 lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int off = 1; off < 64; off <<= 1) {
    const u64 o = __shfl_up(incl, off);
    if ((int)lane >= off) incl += o;
  }
  if (lane == 63) lds4[wave] = incl;
  __syncthreads();
  u64 run = before + incl - mine;
  for (u32 w = 0; w < wave; ++w) run += lds4[w];
  for (u32 k = 0; k < 4; ++k) {
    if (base + k <= n) out[base + k] = run;  // index n receives the total
    run += v[k];
  }
}

int device_scan(hipStream_t st, const u32* in, u64 n, u64* out /*[n+1]*/, u64* tmp, Counters* ctr, u32 shard_cap, u32 surv_cap) {
  if (n <= ((u64)1 << 24)) {
    const u32 tiles = (u32)((n + SCAN_TILE) / SCAN_TILE);  // covers index n as well
    hipLaunchKernelGGL(k_scan_tile_sums, dim3(tiles), dim3(256), 0, st, in, n, tmp, ctr, shard_cap, surv_cap);
    hipLaunchKernelGGL(k_scan_tiles, dim3(tiles), dim3(256), 0, st, in, n, (const u64*)tmp, out);
    return DG_OK;
  }
  if (ctr) hipLaunchKernelGGL(k_leaf_overflow, dim3(NSHARD / 256), dim3(256), 0, st, ctr, shard_cap, surv_cap);
  const u64 n1 = (n + SCAN_CHUNK - 1) / SCAN_CHUNK, n2 = (n1 + SCAN_CHUNK - 1) / SCAN_CHUNK;
  u64* p1 = tmp;
  u64* p2 = tmp + n1;
  const u32 TB = 128;
  hipLaunchKernelGGL(k_scan_sum, dim3(ceil_div(n1, TB)), dim3(TB), 0, st, in, n, p1);
  hipLaunchKernelGGL(k_scan_sum64, dim3(ceil_div(n2, TB)), dim3(TB), 0, st, (const u64*)p1, n1, p2);
  hipLaunchKernelGGL(k_scan_top, dim3(1), dim3(64), 0, st, p2, n2, out + n);
  hipLaunchKernelGGL(k_scan_apply64, dim3(ceil_div(n2, TB)), dim3(TB), 0, st, p1, n1, (const u64*)p2);
  hipLaunchKernelGGL(k_scan_apply, dim3(ceil_div(n1, TB)), dim3(TB), 0, st, in, n, (const u64*)p1, out);
  return DG_OK;
}

// Host pass over the queries whose neighbourhood could reach the cap (neighbors.h:50): the reference's enumeration is
// run for both strands (nbhd_host.hpp).  mode[q] and the explicit patterns of the queries where the cap fired come back.
struct CapScan {
  std::vector<u8> mode;      // per query, QM_*; empty = no query needed a look
  std::vector<u8> xs_bytes;  // codes 0..4
  std::vector<u64> xs_off;
  std::vector<u32> xs_gid;
  u64 looked_at = 0, fired = 0;
};
// Returns DG_OK, DG_ELIMIT when the explicit patterns of this batch would not fit the host budget (the caller passes fewer
// sequences per call: `dicey hunt` halves its chunk and retries), DG_ENOMEM when an enumeration ran out of memory.
// dev_jobs != nullptr: queries that qualify for the device enumeration (k_cap_enum: edit mode, distance <= 2, A/C/G/T only,
// length + distance <= 31) are listed there instead of being enumerated here.
static int cap_scan(const u8* qbytes, const u64* qoff, size_t nq, const dg_hunt_params* p, bool count_mode, CapScan& cs,
                    std::vector<u32>* dev_jobs = nullptr) {
  const bool indel = !p->hamming;
  struct Job {
    size_t q;
    std::string fw, rv;
    u32 d;
    // the set of each strand, flattened to codes as soon as its enumeration ends (the std::set-ordered strings are freed there)
    std::vector<u8> bytes[2];
    std::vector<u32> lens[2];
    bool fired[2] = {false, false};
  };
  std::vector<Job> jobs;
  u64 worst_bytes = 0;  // what the explicit patterns could need if the cap fired on every strand looked at
  for (size_t q = 0; q < nq; ++q) {
    const u64 s = qoff[q], m = qoff[q + 1] - s;
    if (m < 10 && !count_mode) continue;  // hunter.h:299: not searched at all
    u32 d = p->distance, bad = 0;
    if (d >= m) d = (u32)m - 1;  // hunter.h:312-315
    for (u64 i = 0; i < m; ++i) {
      u8 ch = qbytes[s + i];
      if (ch >= 'a' && ch <= 'z') ch -= 32;
      bad += !(ch == 'A' || ch == 'C' || ch == 'G' || ch == 'T');
    }
    if (neighbourhood_bound((u32)m, d, indel, bad) < p->max_neighborhood) continue;
    if (dev_jobs && indel && d >= 1 && d <= 2 && bad == 0 && m >= 10 && m + d <= cap::MAX_KEY_LEN && q < 0x7FFFFFFFull) {
      dev_jobs->push_back((u32)q);
      continue;
    }
    Job j;
    j.q = q;
    j.d = d;
    j.fw.resize(m);
    j.rv.resize(m);
    for (u64 i = 0; i < m; ++i) {  // boost::to_upper_copy + replaceNonDna + reverseComplement (hunter.h:306-309)
      u8 ch = qbytes[s + i];
      if (ch >= 'a' && ch <= 'z') ch -= 32;
      const bool dna = ch == 'A' || ch == 'C' || ch == 'G' || ch == 'T';
      j.fw[i] = dna ? (char)ch : 'N';
      j.rv[m - 1 - i] = ch == 'A' ? 'T' : ch == 'C' ? 'G' : ch == 'G' ? 'C' : ch == 'T' ? 'A' : 'N';
    }
    // a fired strand holds < max_neighborhood + 10 (one position's worth) strings of <= m + d characters, + offset and group words
    worst_bytes += (p->forward_only ? 1ull : 2ull) * ((u64)p->max_neighborhood + 16) * (m + d + 12);
    jobs.push_back(std::move(j));
  }
  if (jobs.empty()) {
    if (dev_jobs && !dev_jobs->empty()) {  // the device pass fills in its queries' modes and appends its patterns
      cs.mode.assign(nq, (u8)QM_KERNEL);
      cs.xs_off.assign(1, 0);
    }
    return DG_OK;
  }
  u64 budget = 16ull << 30;
  if (const char* e = std::getenv("DICEY_CAP_BUDGET_MB")) budget = (u64)std::max(1, std::atoi(e)) << 20;
  if (worst_bytes > budget)
    return fail(DG_ELIMIT, "%zu of the %zu sequences of this call can reach the maxNeighborhood cap (%u); their explicit neighbourhoods may need "
                "%llu MB of host memory (budget %llu MB): pass fewer sequences per call", jobs.size(), nq, p->max_neighborhood,
                (unsigned long long)(worst_bytes >> 20), (unsigned long long)(budget >> 20));
  const bool reverse = !p->forward_only;
  std::atomic<size_t> next{0};
  std::atomic<int> oom{0};
  auto work = [&]() {
    for (;;) {
      const size_t k = next.fetch_add(1);
      if (k >= jobs.size() * 2 || oom.load()) return;
      Job& j = jobs[k >> 1];
      const int strand = (int)(k & 1);
      if (strand && !reverse) continue;
      try {
        bool fired = false;
        const std::vector<std::string> set = CappedNeighborhood::enumerate(strand ? j.rv : j.fw, j.d, indel, p->max_neighborhood, fired);
        j.fired[strand] = fired;
        if (fired) {  // only a capped set travels; a silent one is what the kernel enumerates itself
          size_t tot = 0;
          for (const std::string& str : set) tot += str.size();
          j.bytes[strand].reserve(tot);
          j.lens[strand].reserve(set.size());
          for (const std::string& str : set) {
            for (char ch : str) j.bytes[strand].push_back(ch == 'A' ? 0 : ch == 'C' ? 1 : ch == 'G' ? 2 : ch == 'T' ? 3 : 4);
            j.lens[strand].push_back((u32)str.size());
          }
        }
      } catch (const std::bad_alloc&) {
        oom.store(1);
        return;
      }
    }
  };
  unsigned nthreads = std::thread::hardware_concurrency();
  if (nthreads == 0) nthreads = 1;
  // 32 threads: the enumeration is a chain of hash probes in ~5 MB per strand, i.e. bound by cache misses — on 2 x EPYC 9575F
  // (128 cores) a 25-mer strand at distance 2 takes 0.24 ms of wall time with 32 threads, 0.39 with 64, 0.49 with 256
  // (tools/nbhd_scaling.py)
  nthreads = (unsigned)std::min<size_t>(std::min<unsigned>(nthreads, 32u), jobs.size() * 2);
  if (const char* e = std::getenv("DICEY_HOST_THREADS")) nthreads = (unsigned)std::max(1, std::atoi(e));
  if (nthreads <= 1) work();
  else {
    std::vector<std::thread> pool;
    for (unsigned t = 0; t < nthreads; ++t) pool.emplace_back(work);
    for (auto& t : pool) t.join();
  }
  if (oom.load()) return fail(DG_ENOMEM, "out of host memory while enumerating capped neighbourhoods; pass fewer sequences per call");
  // Second pass, in parallel as well: a query with ONE capped strand is off the kernel path altogether, so its other (silent)
  // strand travels as explicit patterns too.  (r03: this ran inside the serial collection loop below — a third of the 25-mers at
  // distance 2 fire on one strand only, and 2 000 of them took 3 s where the enumeration itself needs 0.4 s on 128 cores.)
  {
    std::vector<size_t> todo;
    for (size_t k = 0; k < jobs.size(); ++k)
      for (int strand = 0; strand < 2; ++strand)
        if (!(strand && !reverse) && !jobs[k].fired[strand] && jobs[k].fired[strand ^ 1]) todo.push_back(2 * k + (size_t)strand);
    std::atomic<size_t> nx{0};
    auto work2 = [&]() {
      for (;;) {
        const size_t t = nx.fetch_add(1);
        if (t >= todo.size() || oom.load()) return;
        Job& j = jobs[todo[t] >> 1];
        const int strand = (int)(todo[t] & 1);
        try {
          bool f2 = false;
          for (const std::string& str : CappedNeighborhood::enumerate(strand ? j.rv : j.fw, j.d, indel, p->max_neighborhood, f2)) {
            for (char ch : str) j.bytes[strand].push_back(ch == 'A' ? 0 : ch == 'C' ? 1 : ch == 'G' ? 2 : ch == 'T' ? 3 : 4);
            j.lens[strand].push_back((u32)str.size());
          }
        } catch (const std::bad_alloc&) {
          oom.store(1);
          return;
        }
      }
    };
    const unsigned nt2 = (unsigned)std::min<size_t>(nthreads, todo.size());
    if (nt2 <= 1) work2();
    else {
      std::vector<std::thread> pool;
      for (unsigned t = 0; t < nt2; ++t) pool.emplace_back(work2);
      for (auto& t : pool) t.join();
    }
    if (oom.load()) return fail(DG_ENOMEM, "out of host memory while enumerating capped neighbourhoods; pass fewer sequences per call");
  }
  try {
    cs.mode.assign(nq, (u8)QM_KERNEL);
    cs.xs_off.assign(1, 0);
    cs.looked_at = jobs.size();
    for (Job& j : jobs) {
      if (!j.fired[0] && !j.fired[1]) {
        cs.mode[j.q] = QM_SILENT;
        continue;
      }
      ++cs.fired;
      cs.mode[j.q] = QM_EXPLICIT | QM_FIRED;
      for (int strand = 0; strand < 2; ++strand) {
        if (strand && !reverse) continue;
        cs.xs_bytes.insert(cs.xs_bytes.end(), j.bytes[strand].begin(), j.bytes[strand].end());
        for (u32 l : j.lens[strand]) {
          cs.xs_off.push_back(cs.xs_off.back() + l);
          cs.xs_gid.push_back((u32)(2 * j.q + strand));
        }
        std::vector<u8>().swap(j.bytes[strand]);
        std::vector<u32>().swap(j.lens[strand]);
      }
    }
  } catch (const std::bad_alloc&) {
    return fail(DG_ENOMEM, "out of host memory while collecting capped neighbourhoods; pass fewer sequences per call");
  }
  return DG_OK;
}

// Pinned host blocks for fetched results, recycled across batches.  hipHostMalloc of a few MB costs more than copying them, and
// copies into pageable memory ran at a third of the link's speed (r02: 43 % of the un-fetched rate).  A result owns its block
// until dg_hunt_result_free hands it back; blocks are never returned to the driver at exit (the runtime may be gone by then).
struct PinnedBlock {
  void* p;
  size_t cap;
};
struct PinnedPool {
  std::mutex mu;
  std::vector<PinnedBlock*> free_;
  size_t cached = 0;
  static constexpr size_t KEEP = 8ull << 30;
  PinnedBlock* get(size_t bytes) {
    {
      std::lock_guard<std::mutex> g(mu);
      size_t best = (size_t)-1;
      for (size_t i = 0; i < free_.size(); ++i)
        if (free_[i]->cap >= bytes && free_[i]->cap <= 4 * bytes + (1u << 20) && (best == (size_t)-1 || free_[i]->cap < free_[best]->cap)) best = i;
      if (best != (size_t)-1) {
        PinnedBlock* b = free_[best];
        free_.erase(free_.begin() + (long)best);
        cached -= b->cap;
        return b;
      }
    }
    size_t cap = 1u << 16;
    while (cap < bytes + bytes / 4) cap <<= 1;  // head room: the next batch's hit count differs a little
    void* p = nullptr;
    if (hipHostMalloc(&p, cap, hipHostMallocDefault) != hipSuccess) {
      (void)hipGetLastError();
      return nullptr;
    }
    return new PinnedBlock{p, cap};
  }
  void put(PinnedBlock* b) {
    {
      std::lock_guard<std::mutex> g(mu);
      if (cached + b->cap <= KEEP) {
        free_.push_back(b);
        cached += b->cap;
        return;
      }
    }
    (void)hipHostFree(b->p);
    delete b;
  }
};
static PinnedPool& pinned_pool() {
  static PinnedPool* P = new PinnedPool;  // leaked on purpose
  return *P;
}

// One batch through the five kernels.  All sizes that are only known on the device (number of leaves, number of hits)
// are handled with capacity guesses that the kernels check themselves; the host synchronises ONCE at the end, and
// repeats the batch with larger buffers in the rare case a capacity was exceeded.
// group_counts != nullptr: count mode (`dicey padlock`, padlock.h:396-421) — stop after the select stage and return, per
// (query, strand), the occurrences summed over the kept neighbourhood strings; no locate, no verify.
// h_qbytes / h_qoff: the host copy of the queries when the caller has one (needed only when a neighbourhood could reach
// the cap; read back from the device otherwise).
int run_batch(dg_index* ix, const dg_hunt_params* p, const uint32_t* seqlen, uint32_t nseq, const void* d_qbytes,
                     const void* d_qoff, size_t nq, u64 total, u32 maxlen, int fetch, dg_hunt_result** out, SearchExtra* sx,
                     uint64_t* group_counts, const uint8_t* h_qbytes, const uint64_t* h_qoff) {
  if (nseq == 0) return fail(DG_EINVAL, "no reference sequences");
  const bool indel = !p->hamming;
  // queries above MAX_QLEN take the banded long-query verify (k_verify_long); alignment lengths are 16-bit in dg_hit
  static constexpr u32 LONG_QLEN_MAX = 30000;
  if (maxlen > LONG_QLEN_MAX) return fail(DG_ELIMIT, "query of %u nt exceeds the supported maximum of %u", maxlen, LONG_QLEN_MAX);
  u32 dmax_eff = p->distance;
  if (maxlen >= 1 && dmax_eff >= maxlen) dmax_eff = maxlen - 1;
  if (dmax_eff > DMAX) return fail(DG_ELIMIT, "distance %u exceeds the supported maximum of %u", p->distance, DMAX);
  DG_HIP(hipSetDevice(ix->device));
  hipStream_t st = ix->stream;
  const bool host_timing = std::getenv("DICEY_TIMING") && std::atoi(std::getenv("DICEY_TIMING")) >= 2;  // host phases of every batch to stderr
  const double t_enter = host_timing ? host_us() : 0.0;
  double t_launched = 0, t_synced = 0;
  for (int i = 0; i < 9; ++i)
    if (!ix->ev[i]) DG_HIP(hipEventCreate(&ix->ev[i]));
  auto& ws = ix->ws;
  const u64 ngrp = 2 * (u64)nq;
  const u32 TB = 256;
  const bool packed = maxlen + dmax_eff <= PACK_MAX_LEN;  // every neighbourhood string fits 128 bits
  // Alignment rows.  The banded verify writes a row front to back, and a row it keeps has the query's characters plus at most d
  // gap columns inside the query (leading and trailing query-gap columns are stripped, hunter.h:391-401, and an optimal path has
  // score >= -d): maxlen + d bytes, rounded up to the 64-bit words it stores.  The other verify kernels build the row from the
  // end of a buffer twice as long.  (r02: 56 -> 24 bytes per row for 20-mers — what travels to the host and over xGMI.)
  const bool no_band = std::getenv("DICEY_NO_BAND_VERIFY") != nullptr;  // (test switches are read per batch: the GPU suite flips them inside one process)
  const bool band_verify = !no_band && !sx && !group_counts && maxlen <= 32 && dmax_eff <= 2;
  // scratch rows of the full-matrix kernels (built from the end of a buffer twice the row length); the banded kernel needs none
  const u32 stride = band_verify ? 0u : ((maxlen + 3 * dmax_eff) + maxlen + 8 + 7) & ~7u;
  const u32 ops_per_hit = dmax_eff;  // compact alignment description: at most |score| <= d columns are not a match
  // ABI 5: results as compact records (position, meta, ops) + one word per query, in ONE device block [qhits | qinfo | records]
  // that a single copy brings to the host — no pack kernel, 12 instead of 24 bytes per hit at distance 1
  const bool compact = (p->flags & DG_HUNT_COMPACT) && !sx && !group_counts;
  const u32 chit_words = 2u + ops_per_hit;
  const u64 scan_tmp = ngrp / SCAN_CHUNK + ngrp / (SCAN_CHUNK * SCAN_CHUNK) + 64;
  DG_TRY(ws[WS_FW].reserve(total + 8));
  DG_TRY(ws[WS_RV].reserve(total + 8));
  DG_TRY(ws[WS_QSEQ].reserve(total + 8));
  DG_TRY(ws[WS_QMETA].reserve(nq * 16 + 64));
  DG_TRY(ws[WS_GINFO].reserve(ngrp * (sizeof(GidInfo) + sizeof(uint4)) + 64));
  DG_TRY(ws[WS_GRP].reserve((ngrp + 1) * 8 + (nq + 1) * 8 + ngrp * 4 * 3 + nq * 4 + scan_tmp * 8 + sizeof(Counters) + sizeof(Summary) + 512));
  DG_TRY(ws[WS_CUM].reserve((u64)nseq * 8 + 8));
  // Could any query of this batch reach the cap?  (the bound grows with the length and with the number of N's)
  const double t_cap0 = host_us();
  CapScan cs;
  std::vector<u32> dev_jobs;  // queries whose capped neighbourhood is enumerated on the device (k_cap_enum)
  // the caller's host copy of the offsets (nullptr on the dg_hunt_device path).  The read-back below serves cap_scan only and must not
  // outlive it: queue_fetch and the result's qoff distinguish "the caller has the offsets" from "pack them on the device" by this.
  const uint64_t* const caller_qoff = h_qoff;
  if (maxlen >= 1 && neighbourhood_bound(maxlen, dmax_eff, indel, maxlen) >= p->max_neighborhood) {
    std::vector<u8> hb;
    std::vector<u64> ho;
    const uint8_t* sb = h_qbytes;
    const uint64_t* so = h_qoff;
    if (!sb || !so) {
      hb.resize(total + 1);
      ho.resize(nq + 1);
      if (total) DG_HIP(hipMemcpyAsync(hb.data(), d_qbytes, total, hipMemcpyDeviceToHost, st));
      DG_HIP(hipMemcpyAsync(ho.data(), d_qoff, (nq + 1) * 8, hipMemcpyDeviceToHost, st));
      DG_HIP(hipStreamSynchronize(st));
      sb = hb.data();
      so = ho.data();
    }
    // DICEY_CAP_HOST: every capped neighbourhood on the host (nbhd_host.hpp), as before r04 — the GPU suite runs both
    DG_TRY(cap_scan(sb, so, nq, p, group_counts != nullptr, cs, std::getenv("DICEY_CAP_HOST") ? nullptr : &dev_jobs));
  }
  u64 nxs = cs.xs_gid.size();
  if (nxs >= 0x0FFFFFFFull || cs.xs_bytes.size() > (48ull << 30))
    return fail(DG_ELIMIT, "%llu explicit neighbourhood strings in one batch; pass fewer sequences per call", (unsigned long long)nxs);
  // room for what k_cap_enum can append: <= max(maxsize, 1) strings of <= maxlen + d characters per strand of its queries
  const u64 dev_strings = dev_jobs.size() * (p->forward_only ? 1ull : 2ull) * std::max<u64>(p->max_neighborhood, 1);
  const u64 dev_bytes = dev_strings * (maxlen + dmax_eff);
  if (!dev_jobs.empty()) {
    u64 budget = 24ull << 30;
    if (const char* e = std::getenv("DICEY_CAP_BUDGET_MB")) budget = (u64)std::max(1, std::atoi(e)) << 20;
    if (dev_bytes + dev_strings * 12 > budget || nxs + dev_strings >= 0x0FFFFFFFull)
      return fail(DG_ELIMIT, "%zu of the %zu sequences of this call can reach the maxNeighborhood cap (%u); their explicit neighbourhoods may need "
                  "%llu MB of device memory (budget %llu MB): pass fewer sequences per call", dev_jobs.size(), nq, p->max_neighborhood,
                  (unsigned long long)((dev_bytes + dev_strings * 12) >> 20), (unsigned long long)(budget >> 20));
  }
  u8* d_qmode = nullptr;
  u8* d_xs_bytes = nullptr;
  u64* d_xs_off = nullptr;
  u32* d_xs_gid = nullptr;
  if (!cs.mode.empty()) {
    const u64 a0 = (nq + 63) & ~63ull, a1 = a0 + ((cs.xs_bytes.size() + dev_bytes + 63) & ~63ull), a2 = a1 + (nxs + dev_strings + 1) * 8,
              a3 = a2 + (nxs + dev_strings) * 4;
    DG_TRY(ws[WS_XS].reserve(a3 + 64));
    u8* base = ws[WS_XS].as<u8>();
    d_qmode = base;
    d_xs_bytes = base + a0;
    d_xs_off = (u64*)(base + a1);
    d_xs_gid = (u32*)(base + a2);
    DG_HIP(hipMemcpyAsync(d_qmode, cs.mode.data(), nq, hipMemcpyHostToDevice, st));
    if (nxs) {
      DG_HIP(hipMemcpyAsync(d_xs_bytes, cs.xs_bytes.data(), cs.xs_bytes.size(), hipMemcpyHostToDevice, st));
      DG_HIP(hipMemcpyAsync(d_xs_gid, cs.xs_gid.data(), nxs * 4, hipMemcpyHostToDevice, st));
    }
    DG_HIP(hipMemcpyAsync(d_xs_off, cs.xs_off.data(), (nxs + 1) * 8, hipMemcpyHostToDevice, st));
    DG_HIP(hipStreamSynchronize(st));  // the host vectors go out of use only after the copies
  }
  if (!dev_jobs.empty()) {
    // neighbors() with its cap on the device (hunt_cap.hpp): modes and explicit patterns of the listed queries, before k_prepare
    // reads them.  One host round trip for the pattern count (launch sizes below depend on it).
    const u32 njobs = (u32)dev_jobs.size();
    const u32 nwg = std::min<u32>(njobs, 512u);
    const u64 leaves = cap::total_leaves(std::min<u32>(maxlen, cap::MAX_KEY_LEN - dmax_eff), dmax_eff) + 2;
    u32 tcap_log2 = 8;
    while ((1ull << tcap_log2) < leaves * 5 / 2) ++tcap_log2;
    const u32 evcap = (u32)((leaves + 2 + 63) & ~63ull);
    const u64 o_tab = 0, o_ev = o_tab + (u64)nwg * 2 * (16ull << tcap_log2), o_jobs = o_ev + (u64)nwg * 2 * evcap * 4,
              o_alloc = (o_jobs + (u64)njobs * 4 + 63) & ~63ull, o_end = o_alloc + 64;
    DG_TRY(ws[WS_CAP].reserve(o_end));
    u8* cb = ws[WS_CAP].as<u8>();
    struct {
      unsigned long long alloc;
      u32 status, pad;
    } h_al = {((unsigned long long)nxs << 36) | (unsigned long long)cs.xs_bytes.size(), 0u, 0u};
    DG_HIP(hipMemcpyAsync(cb + o_jobs, dev_jobs.data(), (u64)njobs * 4, hipMemcpyHostToDevice, st));
    DG_HIP(hipMemcpyAsync(cb + o_alloc, &h_al, sizeof h_al, hipMemcpyHostToDevice, st));
    Batch qb{};  // k_cap_enum reads the raw queries only
    qb.qbytes = (const u8*)d_qbytes;
    qb.qoff = (const u64*)d_qoff;
    qb.nq = nq;
    CapDevArgs ca;
    ca.jobs = (const u32*)(cb + o_jobs);
    ca.njobs = njobs;
    ca.tab = (u64*)(cb + o_tab);
    ca.tcap_log2 = tcap_log2;
    ca.ev = (int*)(cb + o_ev);
    ca.evcap = evcap;
    ca.qmode = d_qmode;
    ca.xs_bytes = d_xs_bytes;
    ca.xs_off = d_xs_off;
    ca.xs_gid = d_xs_gid;
    ca.alloc = (unsigned long long*)(cb + o_alloc);
    ca.cap_strings = nxs + dev_strings;
    ca.cap_bytes = cs.xs_bytes.size() + dev_bytes;
    ca.status = (u32*)(cb + o_alloc + 8);
    ca.maxsize = p->max_neighborhood;
    ca.distance = p->distance;
    ca.reverse = !p->forward_only;
    hipLaunchKernelGGL(k_cap_enum, dim3(nwg), dim3(256), 0, st, qb, ca);
    DG_HIP(hipMemcpyAsync(&h_al, cb + o_alloc, sizeof h_al, hipMemcpyDeviceToHost, st));
    DG_HIP(hipStreamSynchronize(st));
    DG_HIP(hipGetLastError());
    if (h_al.status & 2u) return fail(DG_EHIP, "internal error: a hash table of the device neighbourhood enumeration overflowed");
    if (h_al.status & 1u) return fail(DG_EHIP, "internal error: the device neighbourhood enumeration produced more strings than the cap allows");
    nxs = h_al.alloc >> 36;
  }
  const double ms_cap = (cs.mode.empty() && dev_jobs.empty()) ? 0.0 : (host_us() - t_cap0) * 1e-3;
  Batch b;
  b.fastK = (dmax_eff == 1 && ix->view.K && maxlen > ix->view.K && ngrp * (u64)std::min(maxlen, 31u) * 9 < 0xFFFFFF00ull && ngrp < (1u << 24)) ? ix->view.K : 0u;
  b.fast2K = (indel && dmax_eff == 2 && ix->view.K && maxlen >= ix->view.K + 2 && ngrp < 0x7FFFFFFFull) ? ix->view.K : 0u;
  b.qmode = d_qmode;
  b.xs_bytes = d_xs_bytes;
  b.xs_off = d_xs_off;
  b.xs_gid = d_xs_gid;
  b.nxs = nxs;
  b.qbytes = (const u8*)d_qbytes;
  b.qoff = (const u64*)d_qoff;
  b.nq = nq;
  b.fw = ws[WS_FW].as<u8>();
  b.rv = ws[WS_RV].as<u8>();
  b.qseq = ws[WS_QSEQ].as<u8>();
  u32* meta = ws[WS_QMETA].as<u32>();
  b.qlen = meta;
  b.qdist = meta + nq;
  b.qflags = meta + 2 * nq;
  b.qnondna = meta + 3 * nq;
  b.qinfo = nullptr;  // set per attempt (the compact block may move when the hit capacity grows)
  b.distance = p->distance;
  b.indel = indel;
  b.reverse = !p->forward_only;
  b.max_locations = p->max_locations;
  b.max_neighborhood = p->max_neighborhood;
  b.ginfo = ws[WS_GINFO].as<GidInfo>();
  b.gpeq = reinterpret_cast<uint4*>(ws[WS_GINFO].as<GidInfo>() + ngrp);  // (GidInfo is 16 bytes: the masks stay 16-byte aligned)
  u8* gp = ws[WS_GRP].as<u8>();
  u64* grp_off = (u64*)gp;
  gp += (ngrp + 1) * 8;
  u64* hit_off = (u64*)gp;
  gp += (nq + 1) * 8;
  u64* scan_buf = (u64*)gp;
  gp += scan_tmp * 8;
  gp = (u8*)(((uintptr_t)gp + 63) & ~(uintptr_t)63);
  // counters, device-side summary, group counts and selection counts sit next to each other: one memset clears them
  u8* const zero_from = gp;
  Counters* ctr = (Counters*)gp;
  gp += (sizeof(Counters) + 63) & ~(size_t)63;
  u32* grp_cnt = (u32*)gp;
  gp += ngrp * 4;
  u32* nsel = (u32*)gp;
  gp += ngrp * 4;
  const size_t zero_bytes = (size_t)(gp - zero_from);
  u32* selbase = (u32*)gp;  // first Sel slot of the groups k_search1s serves (set by k_prepare / k_search1s)
  gp += ngrp * 4;
  b.refused = &ctr->pad_[1];
  b.too_long = &ctr->pad_[2];
  b.maxlen_bound = maxlen;
  u32* qhits = (u32*)gp;  // (compact results: re-pointed into the compact block, per attempt)
  if (!ix->pinned) DG_HIP(hipHostMalloc((void**)&ix->pinned, 4096, 0));
  Summary& hsum = *(Summary*)ix->pinned;
  std::vector<u64> cum(nseq);
  u64 run = 0;
  for (u32 r = 0; r < nseq; ++r) {
    cum[r] = run;
    run += seqlen[r];
  }
  if (cum != ix->cum_cache) {  // sequence lengths rarely change between batches
    DG_HIP(hipMemcpyAsync(ws[WS_CUM].p, cum.data(), (u64)nseq * 8, hipMemcpyHostToDevice, st));
    DG_HIP(hipStreamSynchronize(st));
    ix->cum_cache = cum;
  }

  u32 shard_cap = std::max<u32>(ix->shard_cap_hint, (u32)std::max<u64>(64, (16 * (u64)nq + nxs / 8) / NSHARD));
  u64 hit_cap = std::max<u64>(ix->hit_cap_hint, 4 * (u64)nq + 1024);
  // slices of the flat Sel region (k_search1s): what the previous batch's fullest slice needed plus a quarter — kept strings are a
  // third of the leaf estimate above, and k_locate walks every slot of the region
  u32 flat_req = ix->flat_cap_hint ? ix->flat_cap_hint : shard_cap;
  if (const char* e = std::getenv("DICEY_DEBUG_CAPS")) {  // tests: start from tiny capacities to exercise the retry path
    shard_cap = (u32)std::max(1, std::atoi(e));
    hit_cap = (u64)std::max(1, std::atoi(e));
    flat_req = shard_cap;
  }
  u64 nleaf = 0, nhits = 0;
  bool force_generic = false, force_jobs = false;
  // Fetched results: one pinned block from the pool (pageable copies run at a fraction of the link's speed, and a fresh
  // hipHostMalloc per batch costs more than the copies), laid out for `capn` hits:
  // [hit_off | qoff | qdistance qflags qnondna | qseq | hits | ops].  When the previous fetched batch on this handle tells how many
  // hits to expect, the copies are queued BEHIND the batch's kernels, before its one synchronisation (no second round trip);
  // a batch with more hits than expected copies again after the synchronisation.
  struct FetchLayout {
    u64 o_hit_off, o_qoff, o_meta, o_qseq, o_hits, o_ops, bytes;
  };
  auto fetch_layout = [&](u64 capn) {
    FetchLayout L;
    if (compact) {  // [hit_off (filled on the host) | qhits | qinfo | records]; o_qoff = qhits, o_meta = qinfo, o_hits = records
      L.o_hit_off = 0;
      L.o_qoff = (nq + 1) * 8;
      L.o_meta = L.o_qoff + (u64)nq * 4;
      L.o_qseq = L.o_meta + (u64)nq * 4;
      L.o_hits = L.o_qseq;
      L.o_ops = L.o_hits + capn * (u64)chit_words * 4;
      L.bytes = L.o_ops + 64;
      return L;
    }
    L.o_hit_off = 0;
    L.o_qoff = L.o_hit_off + (nq + 1) * 8;
    L.o_meta = L.o_qoff + (nq + 1) * 8;
    L.o_qseq = L.o_meta + 3 * (u64)nq * 4;
    L.o_hits = (L.o_qseq + total + 15) & ~15ull;
    L.o_ops = (L.o_hits + capn * sizeof(dg_hit) + 15) & ~15ull;
    L.bytes = L.o_ops + capn * (u64)ops_per_hit * 4 + 64;
    return L;
  };
  struct BlockGuard {  // hands an unused block back on every early return
    PinnedBlock* pb = nullptr;
    ~BlockGuard() {
      if (pb) pinned_pool().put(pb);
    }
  } spec;
  u64 spec_cap = 0;
  auto queue_fetch = [&](PinnedBlock* pb, u64 capn) -> int {
    const FetchLayout L = fetch_layout(capn);
    if (compact) {  // the device block already has the host layout behind hit_off
      DG_HIP(hipMemcpyAsync((u8*)pb->p + L.o_qoff, ws[WS_PACK].p, L.o_ops - L.o_qoff, hipMemcpyDeviceToHost, st));
      return DG_OK;
    }
    DG_TRY(ws[WS_PACK].reserve(L.bytes + 64));
    PackArgs pa;
    const u32* srcs[6] = {reinterpret_cast<const u32*>(ws[WS_GRP].as<u8>() + (ngrp + 1) * 8),  // hit_off
                          caller_qoff ? nullptr : reinterpret_cast<const u32*>(d_qoff),          // the host has its own copy
                          ws[WS_QMETA].as<u32>() + nq,                                           // qdist, qflags, qnondna lie in this order
                          ws[WS_QSEQ].as<u32>(), ws[WS_HITS].as<u32>(), ws[WS_OPS].as<u32>()};
    const u64 offs[6] = {L.o_hit_off, L.o_qoff, L.o_meta, L.o_qseq, L.o_hits, L.o_ops};
    const u64 words[6] = {(nq + 1) * 2, (nq + 1) * 2, 3 * (u64)nq, (total + 3) / 4, capn * (sizeof(dg_hit) / 4), capn * (u64)ops_per_hit};
    u64 all = 0;
    for (int k = 0; k < 6; ++k) {
      pa.src[k] = srcs[k];
      pa.dst_word[k] = offs[k] / 4;
      pa.nwords[k] = words[k];
      all += words[k];
    }
    if (!ops_per_hit) pa.src[5] = nullptr;
    pa.nhits = (const u64*)(ws[WS_GRP].as<u8>() + (ngrp + 1) * 8) + nq;  // hit_off[nq]
    pa.hit_words = (u32)(sizeof(dg_hit) / 4);
    pa.op_words = ops_per_hit;
    hipLaunchKernelGGL(k_pack_results, dim3(ceil_div(all, 256)), dim3(256), 0, st, pa, ws[WS_PACK].as<u32>());
    DG_HIP(hipMemcpyAsync(pb->p, ws[WS_PACK].p, L.bytes, hipMemcpyDeviceToHost, st));
    return DG_OK;
  };
  for (int attempt = 0;; ++attempt) {
    if (attempt > 8) return fail(DG_ENOMEM, "buffer overflow persists (%llu leaves, %llu hits)", (unsigned long long)nleaf, (unsigned long long)nhits);
    const u64 leaf_slots = (u64)NSHARD * shard_cap;

    // Distance 1, every string in 128 bits: k_search1s settles the select stage inside the search kernel (flat Sel region of
    // NSHARD slices).  The generic kernels (k_search for N-containing / long queries, k_explicit, scan, pack, alive, rank) run
    // when the previous batch of this handle had work for them; a batch that turns out to need them after all is repeated.
    // DICEY_NO_FUSED_SELECT: the search without the select stage (k_search1p + the generic select kernels) — what batches with
    // strings above 42 characters take anyway; the GPU tests run every distance-1 case both ways
    const bool no_fuse = std::getenv("DICEY_NO_FUSED_SELECT") != nullptr;
    const bool fused = b.fastK && packed && !no_fuse;
    const u32 flat_cap = fused ? flat_req : 0u;
    const u64 flat_slots = (u64)NSHARD * flat_cap;
    const bool generic_on = !fused || ix->generic_hint || nxs > 0 || force_generic;
    bool jobs_on = false;
    DG_TRY(ws[WS_LEAF].reserve(leaf_slots * sizeof(Leaf)));
    DG_TRY(ws[WS_LEAFG].reserve((leaf_slots + 1) * (sizeof(Leaf) > sizeof(PLeaf) ? sizeof(Leaf) : sizeof(PLeaf))));
    DG_TRY(ws[WS_SEL].reserve((flat_slots + leaf_slots + 1) * sizeof(Sel)));
    Sel* const sel_all = ws[WS_SEL].as<Sel>();
    Sel* const sel_gen = sel_all + flat_slots;  // the generic path's slots: grp_off based, behind the flat region
    DG_TRY(ws[WS_SCR].reserve((leaf_slots + 1) * 5 + 64));
    DG_TRY(ws[WS_SEEDS].reserve((hit_cap + 1) * sizeof(HitSeed)));
    const u32 surv_cap = 0xFFFFFFFFu;  // (r02's survivor queue in HBM is gone with k_probe1 / k_finish1; the scan kernel's check keeps its argument)
    DG_TRY(ws[WS_JOBS].reserve(2 * std::min<u64>(leaf_slots, 1u << 20) * sizeof(BigJob)));
    DG_TRY(ws[WS_HITS].reserve((hit_cap + 1) * sizeof(dg_hit)));
    if (stride && !sx && !group_counts) DG_TRY(ws[WS_ALN].reserve((hit_cap + 1) * 2 * (u64)stride));
    if (!sx && !group_counts) DG_TRY(ws[WS_OPS].reserve((hit_cap + 1) * (u64)ops_per_hit * 4 + 64));
    u32* d_chits = nullptr;
    if (compact) {
      DG_TRY(ws[WS_PACK].reserve(2 * (u64)nq * 4 + (hit_cap + 1) * (u64)chit_words * 4 + 64));
      qhits = ws[WS_PACK].as<u32>();
      b.qinfo = qhits + nq;
      d_chits = qhits + 2 * nq;
    }
    // counters: left zeroed by the previous batch's last kernel (batch_finish) unless they moved or that batch did not finish;
    // group counters: cleared by k_prepare
    if (ix->ctr_clean != (const void*)ctr || ix->ctr_clean_gen != ws[WS_GRP].gen) DG_HIP(hipMemsetAsync(zero_from, 0, zero_bytes, st));
    ix->ctr_clean = nullptr;  // dirty until this attempt's last kernel has run
    DG_HIP(hipEventRecord(ix->ev[0], st));
    // the whole batch on the flat distance-1 path: k_search1s settles the `take` values of its own queries (TAKE form);
    // DICEY_NO_PREP_FUSION keeps k_take a launch of its own (the GPU suite runs both)
    const bool prep_in = fused && !generic_on && !group_counts && !std::getenv("DICEY_NO_PREP_FUSION");
    // the per-character arrays (fw / rv codes, normalised ASCII) are read by the generic kernels, the full-matrix verify kernels and
    // the classic result fetch only: 60 byte stores per query that the flat path with compact results does without
    const u32 write_bytes = (prep_in && band_verify && (compact || !fetch)) ? 0u : 1u;
    hipLaunchKernelGGL(k_prepare, dim3(ceil_div(nq, TB)), dim3(TB), 0, st, b, grp_cnt, nsel, selbase, (u32*)&ctr->pad_[6], write_bytes);
    DG_HIP(hipEventRecord(ix->ev[1], st));
    {
      SearchOut so;
      so.leaves = ws[WS_LEAF].as<Leaf>();
      so.shard_cap = shard_cap;
      so.ctr = ctr;
      so.grp_cnt = grp_cnt;
      if (b.fastK) {  // distance 1: the flat kernel takes every query that qualifies, k_search (one lane per strand) the rest
        const u32 ipg = std::min(maxlen, 31u), magic = (65536u + ipg - 1) / ipg;  // longer queries stay with k_search
        if (fused) {
          const u32 gpw = prep_in ? (std::min(16u, 256u / ipg) & ~1u) : std::min(16u, 256u / ipg);  // TAKE: both strands of a query in one workgroup
          FlatSel fs;
          fs.sel = sel_all;
          fs.cap = flat_cap;
          fs.selbase = selbase;
          fs.nsel = nsel;
          const dim3 g1(ceil_div(ngrp, gpw)), b1(256);
          // LDS list of a workgroup: 256 entries unless the previous batch of this handle averaged more than 48 occurring strings
          // per workgroup (repeat-bearing genomes), then 512; a workgroup whose strings do not fit hands its groups to the generic
          // path.  tests: DICEY_FUSED_LCAP lowers the capacity so that ordinary batches exercise that hand-over
          const u32 lcap_env = std::getenv("DICEY_FUSED_LCAP") ? std::max<u32>(1u, std::min<u32>(FUSED_LCAP, (u32)std::atoi(std::getenv("DICEY_FUSED_LCAP")))) : 0u;
          const u32 lcap = lcap_env ? lcap_env : (ix->fused_leaves_hint > 48ull * g1.x ? FUSED_LCAP : FUSED_LCAP / 2);
          const u32 lds1 = fused_lds_bytes(lcap);
          const u32 leave1 = std::getenv("DICEY_EXP_NOLEAVE") ? 0u : 1u;  // (r04 A/B: idle wavefronts end behind the probe phase or wait at the barrier)
          PrepOut po;
          po.qhits = qhits;
          if (prep_in) {
            if (indel) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_search1s<true, true>), g1, b1, lds1, st, ix->view, b, so, fs, ipg, magic, gpw, lcap, leave1, po);
            else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_search1s<false, true>), g1, b1, lds1, st, ix->view, b, so, fs, ipg, magic, gpw, lcap, leave1, po);
          } else {
            if (indel) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_search1s<true, false>), g1, b1, lds1, st, ix->view, b, so, fs, ipg, magic, gpw, lcap, leave1, po);
            else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_search1s<false, false>), g1, b1, lds1, st, ix->view, b, so, fs, ipg, magic, gpw, lcap, leave1, po);
          }
        } else {
          const dim3 g1(ceil_div(ngrp * ipg, TB)), b1(TB);
          if (indel) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_search1p<true>), g1, b1, 0, st, ix->view, b, so, ipg, magic);
          else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_search1p<false>), g1, b1, 0, st, ix->view, b, so, ipg, magic);
        }
        DG_HIP(hipEventRecord(ix->ev[8], st));
      }
      if (b.fast2K) {  // edit distance 2: one workgroup per (query, strand) for every query that qualifies
        hipLaunchKernelGGL(k_search2p, dim3((u32)ngrp), dim3(TB), 0, st, ix->view, b, so);
        DG_HIP(hipEventRecord(ix->ev[8], st));
      }
      if (generic_on) {
      // root-level work split (see k_search): only with the table and with at least one edit to place
      const u32 items = (ix->view.K && dmax_eff >= 1 && !b.fastK) ? ix->view.K * (indel ? 9u : 4u) + 1u : 1u;
      const dim3 grid(ceil_div(ngrp * items, TB)), block(TB);
#define DG_LAUNCH_SEARCH(IND, DD) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_search<IND, DD>), grid, block, 0, st, ix->view, b, so, items)
      if (indel) {
        if (dmax_eff == 0) DG_LAUNCH_SEARCH(true, 0);
        else if (dmax_eff == 1) DG_LAUNCH_SEARCH(true, 1);
        else if (dmax_eff == 2) DG_LAUNCH_SEARCH(true, 2);
        else DG_LAUNCH_SEARCH(true, 4);
      } else {
        if (dmax_eff == 0) DG_LAUNCH_SEARCH(false, 0);
        else if (dmax_eff == 1) DG_LAUNCH_SEARCH(false, 1);
        else if (dmax_eff == 2) DG_LAUNCH_SEARCH(false, 2);
        else DG_LAUNCH_SEARCH(false, 4);
      }
#undef DG_LAUNCH_SEARCH
      if (nxs) hipLaunchKernelGGL(k_explicit, dim3(ceil_div(nxs, TB)), dim3(TB), 0, st, ix->view, b, so);
      }
    }
    DG_HIP(hipEventRecord(ix->ev[2], st));
    // (r03 tried single-launch scans chained by decoupled look-back, and k_take fused with its scan: 17 us against 2 x 4.3 us, and
    //  47 us against 7 + 9 us — descriptor polling with device-scope acquire / release is slow across the XCDs' L2s.  What stays of
    //  that round: the first scan kernel also checks the search kernels' buffers, which was a launch of its own.)
    if (generic_on) DG_TRY(device_scan(st, grp_cnt, ngrp, grp_off, scan_buf, ctr, shard_cap, surv_cap));
    DG_HIP(hipEventRecord(ix->ev[3], st));
    if (packed) {
      if (generic_on) {
      u8* alive = ws[WS_SCR].as<u8>();
      hipLaunchKernelGGL(k_group_pack, dim3(ceil_div(leaf_slots, TB)), dim3(TB), 0, st, b, ws[WS_LEAF].as<Leaf>(), shard_cap, ctr,
                         grp_off, ws[WS_LEAFG].as<PLeaf>());
      // distance >= 2: groups of up to SELCAP strings are sorted by a workgroup each, the lane-per-leaf kernels keep the rest
      const u32 above = (dmax_eff >= 2 && ngrp < 0x7FFFFFFFull) ? SELCAP : 0u;
      if (above)
        // (one wavefront per group was tried: 1.95 -> 2.5 ms, the second wavefront's share of the window searches is worth more than the barriers)
        hipLaunchKernelGGL(k_group_select, dim3((u32)ngrp), dim3(128), 0, st, ws[WS_LEAFG].as<PLeaf>(), grp_off, (u32)indel,
                           sel_gen, nsel, ctr);
      hipLaunchKernelGGL(k_leaf_alive, dim3(ceil_div(leaf_slots, TB)), dim3(TB), 0, st, ws[WS_LEAFG].as<PLeaf>(), grp_off, ngrp,
                         (u32)indel, alive, ctr, above);
      hipLaunchKernelGGL(k_leaf_rank, dim3(ceil_div(leaf_slots, TB)), dim3(TB), 0, st, ws[WS_LEAFG].as<PLeaf>(), grp_off, ngrp,
                         (const u8*)alive, sel_gen, nsel, ctr, above);
      }
      if (!prep_in)
        hipLaunchKernelGGL(k_take, dim3(ceil_div(nq, TB)), dim3(TB), 0, st, b, (const u64*)grp_off, (const u32*)selbase, flat_slots, (const u32*)nsel, sel_all,
                           qhits, ctr);
    } else {
      hipLaunchKernelGGL(k_group, dim3(ceil_div(leaf_slots, TB)), dim3(TB), 0, st, ws[WS_LEAF].as<Leaf>(), shard_cap, ctr, grp_off,
                         ws[WS_LEAFG].as<Leaf>());
      u32* scr_rank = ws[WS_SCR].as<u32>();
      u8* scr_keep = (u8*)(scr_rank + leaf_slots + 1);
      hipLaunchKernelGGL(k_select, dim3(ceil_div(nq, 64)), dim3(64), 0, st, b, ws[WS_LEAFG].as<Leaf>(), grp_off, sel_gen,
                         nsel, qhits, scr_keep, scr_rank, ctr);
    }
    DG_HIP(hipEventRecord(ix->ev[4], st));
    if (group_counts) {
      DG_TRY(ws[WS_HITS].reserve(ngrp * 8 + 64));
      hipLaunchKernelGGL(k_group_count, dim3(ceil_div(ngrp, TB)), dim3(TB), 0, st, (const u64*)grp_off, (const u32*)selbase, flat_slots,
                         (const u32*)nsel, (const Sel*)sel_all, ngrp, ws[WS_HITS].as<u64>(), ctr);
      DG_HIP(hipMemsetAsync(hit_off + nq, 0, 8, st));  // no hits in this mode
      for (int e = 5; e <= 7; ++e) DG_HIP(hipEventRecord(ix->ev[e], st));
    } else {
    DG_TRY(device_scan(st, qhits, nq, hit_off, scan_buf));
    DG_HIP(hipEventRecord(ix->ev[5], st));
    {
      // strings with many occurrences go to two job lists: up to 256 occurrences for a wavefront each, more for a workgroup each
      const u32 job_cap = (u32)std::min<u64>(leaf_slots, 1u << 20);
      LocJobs lj;
      lj.small = ws[WS_JOBS].as<BigJob>();
      lj.big = ws[WS_JOBS].as<BigJob>() + job_cap;
      lj.cap = job_cap;
      lj.n_big = (u32*)&ctr->pad_[0];
      lj.n_small = (u32*)&ctr->pad_[4];
      // the record that shares a kept string's slot names its group: the packed leaf (qs behind 28 bytes) or the grouped leaf (qs first)
      const u8* slot_qs = packed ? (const u8*)ws[WS_LEAFG].p + offsetof(PLeaf, qs) : (const u8*)ws[WS_LEAFG].p + offsetof(Leaf, qs);
      const u32 slot_stride = packed ? (u32)sizeof(PLeaf) : (u32)sizeof(Leaf);
      hipLaunchKernelGGL(k_locate, dim3(ceil_div(flat_slots + (generic_on ? leaf_slots : 0), TB)), dim3(TB), 0, st, ix->view, (const Sel*)sel_all, slot_qs,
                         slot_stride, (const u64*)grp_off, (const u32*)nsel, ngrp, (const u64*)hit_off, ws[WS_SEEDS].as<HitSeed>(), ctr, hit_cap, lj,
                         flat_slots, flat_cap, (u32)generic_on, (u32)(ix->jobs_hint || force_jobs || p->max_locations > TOPK_KMAX));
      // The job kernels (strings of more than 16 occurrences) are launched when the previous batch of this handle queued any job;
      // k_locate queues and counts whether or not they run, and a batch that had jobs after one that had none is repeated with
      // them (the same device as for capacity guesses).  Uniform batches on a genome without repeats never launch them.
      jobs_on = ix->jobs_hint || force_jobs || p->max_locations > TOPK_KMAX;
      if (jobs_on) {
        // repeat-rich strings: up to TOPK_KMAX positions through the block minima, more (hunt -m above 1 024) by radix passes
        const bool topk = ix->view.nlev > 1;
        // A batch with thousands of repeat-rich strings (the previous batch of this handle tells): intervals up to 4 608 entries go
        // to the small-buffer form of k_locate_topk.  (r03 also ran the three job kernels side by side on helper streams: each
        // slowed down by what the others took — 114 / 220 / 228 us alone, 220 / 494 / 268 us together — and the stage gained
        // 0.08 of 0.91 ms; not worth two more streams per handle.)
        const bool rich = topk && ix->jobs_big_hint >= 2048;
        const u32 mid_max = rich ? 8 * TOPK_KCAP_MID : 0u;
        hipLaunchKernelGGL(k_locate_small, dim3(8192), dim3(64), 0, st, ix->view, (const BigJob*)lj.small, (const u32*)lj.n_small, job_cap,
                           ws[WS_SEEDS].as<HitSeed>(), ctr);
        if (topk) {
          if (mid_max)
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_locate_topk<TOPK_KCAP_MID>), dim3(1536), dim3(256), 0, st, ix->view, (const BigJob*)lj.big,
                               (const u32*)lj.n_big, job_cap, (u32*)&ctr->pad_[8], ws[WS_SEEDS].as<HitSeed>(), ctr, 0u, mid_max);
          hipLaunchKernelGGL(HIP_KERNEL_NAME(k_locate_topk<TOPK_KCAP>), dim3(768), dim3(256), 0, st, ix->view, (const BigJob*)lj.big,
                             (const u32*)lj.n_big, job_cap, (u32*)&ctr->pad_[3], ws[WS_SEEDS].as<HitSeed>(), ctr, mid_max, 0xFFFFFFFFu);
        }
        if (!topk || p->max_locations > TOPK_KMAX)
          hipLaunchKernelGGL(k_locate_big, dim3(1024), dim3(256), 0, st, ix->view, (const BigJob*)lj.big, (const u32*)lj.n_big, job_cap,
                             ws[WS_SEEDS].as<HitSeed>(), ctr, topk ? TOPK_KMAX : 0u);
      }
    }
    DG_HIP(hipEventRecord(ix->ev[6], st));
    if (sx) {
      DG_TRY(launch_site_stage(ix, sx, b, ws[WS_SEEDS].as<HitSeed>(), hit_off, hit_cap, ws[WS_CUM].as<u64>(), nseq, dmax_eff, maxlen, ctr));
    } else {
      VerifyArgs va;
      va.seeds = ws[WS_SEEDS].as<HitSeed>();
      va.nhits = hit_off + nq;
      va.hit_cap = hit_cap;
      va.cum = ws[WS_CUM].as<u64>();
      va.nseq = nseq;
      va.hits = ws[WS_HITS].as<dg_hit>();
      va.refalign = ws[WS_ALN].as<char>();
      va.queryalign = ws[WS_ALN].as<char>() + (hit_cap + 1) * (u64)stride;
      va.stride = stride;
      va.ops = ws[WS_OPS].as<u32>();
      va.ops_per_hit = ops_per_hit;
      va.chits = nullptr;
      va.debug = 0;
      const u32 cells = (maxlen + 3 * dmax_eff + 1) * (maxlen + 1);
      const u32 VT = 128;
      const dim3 vgrid(ceil_div(hit_cap, VT)), vblock(VT);
      if (band_verify) {
        va.chits = d_chits;  // the banded kernel writes compact records itself
        // hits per lane: 1 while a query has a handful of hits (every window is its own class: nothing to share, smallest LDS
        // footprint), 8 when the previous batch had dozens of hits per query (repeat families: ~240 hits per kept string)
        const int ch_env = std::getenv("DICEY_VERIFY_CH") ? std::atoi(std::getenv("DICEY_VERIFY_CH")) : 0;
        const u64 per_q = hit_cap / std::max<u64>(nq, 1);
        // (distance 2 on an i.i.d. genome: 59 hits per query from ~50 strings — nothing to share, and the wider trace of 13 diagonals
        //  leaves room for fewer workgroups: r03 measured 2.39 ms with 8 hits per lane against 1.83 ms for the lane-per-hit kernel)
        const u64 share_at = dmax_eff > 1 ? 192 : 24;
        const int ch = ch_env == 1 || ch_env == 4 || ch_env == 8 ? ch_env : (per_q >= share_at ? 8 : (per_q >= share_at / 2 ? 4 : 1));
        const u32 rows = maxlen + 3 * dmax_eff + 2;
        const bool wide = dmax_eff > 1;
        const u32 nw_bytes = ((rows * 256 * (wide ? 4u : 2u) + 7u) & ~7u) + 6u * 256u * 8u, hash_bytes = 2u * 256u * (u32)ch * 10u;
        const u32 lds = std::max(nw_bytes, hash_bytes);
        const dim3 mgrid(ceil_div(hit_cap, (u64)256 * ch)), mblock(256);
#define DG_LAUNCH_MEMO(WBV, CHV) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_verify_memo<WBV, CHV>), mgrid, mblock, lds, st, ix->view, b, va, ctr, rows)
        if (!wide) {
          if (ch == 8) DG_LAUNCH_MEMO(7, 8);
          else if (ch == 4) DG_LAUNCH_MEMO(7, 4);
          else DG_LAUNCH_MEMO(7, 1);
        } else {
          if (ch == 8) DG_LAUNCH_MEMO(13, 8);
          else if (ch == 4) DG_LAUNCH_MEMO(13, 4);
          else DG_LAUNCH_MEMO(13, 1);
        }
#undef DG_LAUNCH_MEMO
      } else {
        if (maxlen > MAX_QLEN) {
          const u32 rows_cap = maxlen + 3 * dmax_eff + 2;
          const u64 tbytes = (hit_cap + 1) * (u64)rows_cap * 8;
          if (tbytes > (64ull << 30))
            return fail(DG_ELIMIT, "a batch with a %u nt query and room for %llu hits needs %llu GB of trace; pass long queries in smaller batches",
                        maxlen, (unsigned long long)hit_cap, (unsigned long long)(tbytes >> 30));
          DG_TRY(ws[WS_DP].reserve(tbytes + 64));
          hipLaunchKernelGGL(HIP_KERNEL_NAME(k_verify_long<6 * DMAX + 1>), dim3(ceil_div(hit_cap, 64)), dim3(64), 0, st, ix->view, b, va, ctr,
                             ws[WS_DP].as<u64>(), rows_cap);
        } else if (maxlen <= 24) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_verify<1, true, 24>), vgrid, vblock, 0, st, ix->view, b, va, ctr);
        else if (maxlen <= 32) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_verify<1, true, 32>), vgrid, vblock, 0, st, ix->view, b, va, ctr);
        else if (cells <= 32 * 160) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_verify<160, false>), vgrid, vblock, 0, st, ix->view, b, va, ctr);
        else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_verify<2200, false>), vgrid, vblock, 0, st, ix->view, b, va, ctr);
        if (ops_per_hit) hipLaunchKernelGGL(k_rows_to_ops, dim3(ceil_div(hit_cap, 256)), dim3(256), 0, st, va, ctr);
        if (compact) {
          va.chits = d_chits;
          hipLaunchKernelGGL(k_hits_to_compact, dim3(ceil_div(hit_cap, 256)), dim3(256), 0, st, va, (const Counters*)ctr);
        }
      }
    }
    DG_HIP(hipEventRecord(ix->ev[7], st));
    }
    hipLaunchKernelGGL(k_summary_block, dim3(1), dim3(NSHARD), 0, st, ctr, (const u64*)(hit_off + nq), &hsum);
    if (fetch && !group_counts && !sx && ix->fetch_hits_hint) {
      const u64 capn = std::min<u64>(hit_cap, ix->fetch_hits_hint);
      if (!spec.pb || spec_cap != capn) {
        if (spec.pb) pinned_pool().put(spec.pb);
        spec.pb = pinned_pool().get(fetch_layout(capn).bytes);
        spec_cap = capn;
      }
      if (spec.pb) DG_TRY(queue_fetch(spec.pb, capn));
    }
    if (host_timing) t_launched = host_us();
    DG_HIP(hipStreamSynchronize(st));  // the only synchronisation of a batch
    DG_HIP(hipGetLastError());
    if (host_timing) t_synced = host_us();
    ix->ctr_clean = ctr;  // batch_finish left the counters zeroed where they are
    ix->ctr_clean_gen = ws[WS_GRP].gen;
    if (hsum.too_long)
    return fail(DG_EINVAL, "%llu queries are longer than the %u nt this batch was sized for%s", hsum.too_long, maxlen,
                p->max_query_len ? " (dg_hunt_params::max_query_len is not an upper bound of this batch, or its offsets decrease)" : "");
    if (hsum.refused)
      return fail(DG_EINVAL, "internal error: %llu quer%s could reach the maxNeighborhood cap (%u) without having been enumerated on the host",
                  hsum.refused, hsum.refused == 1 ? "y" : "ies", p->max_neighborhood);
    nleaf = hsum.nleaf;
    nhits = hsum.nhits;
    // Everything this attempt found wanting is put right before the batch is repeated (one repeat usually serves several causes).
    bool again = false;
    // the generic kernels were left out, and the batch had work for them after all (queries with N, longer than 31 nt, a workgroup
    // of k_search1s whose strings did not fit its LDS list): once more, with them
    if (fused && !generic_on && (hsum.nleaf > 0 || hsum.n_generic > 0)) {
      ix->generic_hint = true;
      force_generic = true;
      again = true;
    }
    if (!jobs_on && !group_counts && (hsum.jobs_small > 0 || hsum.jobs_big > 0)) {  // strings were queued and nobody served them
      ix->jobs_hint = true;
      force_jobs = true;
      again = true;
    }
    const u32 worst = (u32)hsum.worst_shard;
    if (worst > shard_cap) {
      shard_cap = worst + worst / 4 + 64;
      again = true;
    }
    if (fused && hsum.worst_sel > flat_cap) {
      flat_req = (u32)(hsum.worst_sel + hsum.worst_sel / 4 + 64);
      again = true;
    }
    if (!(hsum.overflow & 1) && nhits > hit_cap) {  // (after a buffer overflow the hit count is not this batch's)
      hit_cap = nhits + nhits / 4 + 1024;
      again = true;
    }
    if (again) continue;
    // The hints are sticky: a kernel family that a batch needed stays on for the next eight batches of the handle, so that a stream
    // that alternates (chunks with and without N-containing queries, with and without repeat-rich strings) does not pay a repeated
    // batch at every change (r03 advice; r04 repeats genome: 4 of 14 rotating steps ran twice).
    if (fused) {
      if (hsum.nleaf > 0 || hsum.n_generic > 0 || nxs > 0) ix->generic_sticky = 8;
      else if (ix->generic_sticky) --ix->generic_sticky;
      ix->generic_hint = ix->generic_sticky > 0;
    }
    if (!group_counts) {
      if (hsum.jobs_small > 0 || hsum.jobs_big > 0) ix->jobs_sticky = 8;
      else if (ix->jobs_sticky) --ix->jobs_sticky;
      ix->jobs_hint = ix->jobs_sticky > 0;
    }
    break;
  }
  ix->shard_cap_hint = shard_cap;
  if (b.fastK && hsum.worst_sel) ix->flat_cap_hint = (u32)std::max<unsigned long long>(ix->flat_cap_hint, hsum.worst_sel + hsum.worst_sel / 4 + 64);
  ix->jobs_big_hint = hsum.jobs_big;
  if (b.fastK) ix->fused_leaves_hint = hsum.fused_leaves + nleaf;
  ix->hit_cap_hint = std::max<u64>(ix->hit_cap_hint, nhits + nhits / 4 + 1024);
  if (group_counts) {
    DG_HIP(hipMemcpyAsync(group_counts, ws[WS_HITS].p, ngrp * 8, hipMemcpyDeviceToHost, st));
    DG_HIP(hipStreamSynchronize(st));
  }

  if (hsum.overflow & 2)
    return fail(DG_EHIP, "internal error: an alignment left the band its hit guarantees (or needs more than %u edit columns)", ops_per_hit);
  dg_hunt_result* R = new dg_hunt_result;
  std::memset(R, 0, sizeof *R);
  R->nq = nq;
  R->nhits = nhits;
  R->ops_per_hit = ops_per_hit;
  *out = R;
  if (fetch) {
    PinnedBlock* pb = nullptr;
    u64 capn = 0;
    if (spec.pb && nhits <= spec_cap) {  // everything is on the host already
      pb = spec.pb;
      capn = spec_cap;
      spec.pb = nullptr;
    } else {
      capn = nhits;
      pb = pinned_pool().get(fetch_layout(capn).bytes);
      if (!pb) {
        delete R;
        *out = nullptr;
        return fail(DG_ENOMEM, "cannot allocate %llu bytes of pinned host memory for the results", (unsigned long long)fetch_layout(capn).bytes);
      }
      R->owner_ = pb;  // released with the result on the error paths below
      DG_TRY(queue_fetch(pb, capn));
      DG_HIP(hipStreamSynchronize(st));
      DG_HIP(hipGetLastError());
    }
    R->owner_ = pb;
    const FetchLayout L = fetch_layout(capn);
    u8* hb = (u8*)pb->p;
    if (compact) {
      R->compact = 1;
      R->hit_off = (uint64_t*)(hb + L.o_hit_off);
      const u32* qh = (const u32*)(hb + L.o_qoff);
      u64 run2 = 0;
      for (size_t i = 0; i < nq; ++i) {
        R->hit_off[i] = run2;
        run2 += qh[i];
      }
      R->hit_off[nq] = run2;
      if (run2 != nhits) return fail(DG_EHIP, "internal error: per-query hit counts add up to %llu, the batch has %llu hits", (unsigned long long)run2, (unsigned long long)nhits);
      R->qinfo = (uint32_t*)(hb + L.o_meta);
      R->chits = (uint32_t*)(hb + L.o_hits);
      R->nseq = nseq;
      R->seq_start = (uint64_t*)std::malloc(std::max<size_t>(1, nseq) * 8);
      if (!R->seq_start) return fail(DG_ENOMEM, "out of host memory");
      std::memcpy(R->seq_start, cum.data(), (size_t)nseq * 8);
      ix->fetch_hits_hint = nhits + nhits / 32 + 1024;
    } else {
    if (caller_qoff) std::memcpy(hb + L.o_qoff, caller_qoff, (nq + 1) * 8);
    R->hit_off = (uint64_t*)(hb + L.o_hit_off);
    R->qoff = (uint64_t*)(hb + L.o_qoff);
    R->hits = (dg_hit*)(hb + L.o_hits);
    R->ops = (uint32_t*)(hb + L.o_ops);
    R->qdistance = (uint32_t*)(hb + L.o_meta);
    R->qflags = R->qdistance + nq;
    R->qnondna = R->qflags + nq;
    R->qseq = hb + L.o_qseq;
    ix->fetch_hits_hint = nhits + nhits / 32 + 1024;
    }
  }
  if (compact) {
    R->compact = 1;
    R->d_hits = ws[WS_PACK].as<u32>() + 2 * nq;  // compact records (chit_words each)
    R->d_ops = nullptr;
  } else {
    R->d_hits = ws[WS_HITS].p;
    R->d_ops = ops_per_hit ? ws[WS_OPS].p : nullptr;
  }
  R->ctr_leaves = nleaf + hsum.fused_leaves;
  R->ctr_ext_steps = hsum.steps;
  R->ctr_tab_reads = hsum.lookups;
  R->ctr_filter_probes = hsum.probes;
  R->ctr_sa_reads = hsum.sa_reads;
  R->ctr_win_bytes = hsum.win_bytes;
  R->ms_total = ev_ms(ix->ev[0], ix->ev[7]);
  R->ms_search = ev_ms(ix->ev[1], ix->ev[2]);
  R->ms_search_flat = (b.fastK || b.fast2K) ? ev_ms(ix->ev[1], ix->ev[8]) : 0.0;
  R->ms_select = ev_ms(ix->ev[3], ix->ev[4]);
  R->ms_locate = ev_ms(ix->ev[5], ix->ev[6]);
  R->ms_verify = ev_ms(ix->ev[6], ix->ev[7]);
  R->ms_cap = ms_cap;
  R->cap_queries_device = dev_jobs.size();
  R->cap_queries_host = cs.looked_at;
  R->cap_patterns = nxs;
  if (host_timing)
    std::fprintf(stderr, "dicey timing: batch of %zu on stream %p entered at %.0f us: host %.0f us before the synchronisation (last attempt's "
                 "launches included), %.0f us waiting, %.0f us after; device %.3f ms\n", nq, (void*)st, std::fmod(t_enter, 1e8), t_launched - t_enter,
                 t_synced - t_launched, host_us() - t_synced, R->ms_total);
  return DG_OK;
}

}  // namespace dg

using namespace dg;

extern "C" {

void dg_hunt_result_free(dg_hunt_result* r) {
  if (!r) return;
  if (r->owner_) pinned_pool().put((PinnedBlock*)r->owner_);
  std::free(r->seq_start);
  std::free(r->expanded_);
  std::free(r->refalign);
  std::free(r->queryalign);
  delete r;
}

int dg_hit_rows(const dg_hit* hit, const uint32_t* ops, uint32_t ops_per_hit, const uint8_t* qseq, uint32_t qlen, char* refalign,
                char* queryalign) {
  if (!hit || !qseq || !refalign || !queryalign || (ops_per_hit && !ops)) return fail(DG_EINVAL, "dg_hit_rows: null argument");
  const bool rev = hit->strand == '-';
  auto qch = [&](u32 i) -> char {  // character i of the strand the hit was aligned to (util.h:54-114 for the reverse strand)
    if (!rev) return (char)qseq[i];
    const u8 c = qseq[qlen - 1 - i];
    return c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : c == 'T' ? 'A' : 'N';
  };
  u32 qi = 0, k = 0;
  for (u32 col = 0; col < hit->aln_len; ++col) {
    const uint32_t op = k < ops_per_hit ? ops[k] : DG_ALN_NONE;
    if (op != DG_ALN_NONE && DG_ALN_COL(op) == col) {
      ++k;
      const u32 kind = DG_ALN_KIND(op);
      if (kind == DG_ALN_QUERY_GAP) {
        refalign[col] = (char)DG_ALN_BYTE(op);
        queryalign[col] = '-';
        continue;
      }
      if (qi >= qlen) return fail(DG_EINVAL, "dg_hit_rows: the description consumes more than the query's %u characters", qlen);
      refalign[col] = kind == DG_ALN_REF_GAP ? '-' : (char)DG_ALN_BYTE(op);
      queryalign[col] = qch(qi++);
    } else {
      if (qi >= qlen) return fail(DG_EINVAL, "dg_hit_rows: the description consumes more than the query's %u characters", qlen);
      refalign[col] = queryalign[col] = qch(qi++);
    }
  }
  if (qi != qlen || (k < ops_per_hit && ops[k] != DG_ALN_NONE))
    return fail(DG_EINVAL, "dg_hit_rows: %u of %u query characters and %u operations used by %u columns", qi, qlen, k, (u32)hit->aln_len);
  return DG_OK;
}

int dg_hunt_rows(dg_hunt_result* r) {
  if (!r) return fail(DG_EINVAL, "dg_hunt_rows: null result");
  if (r->refalign) return DG_OK;
  if (r->nhits && (!r->hits || !r->qseq || !r->qoff)) return fail(DG_EINVAL, "dg_hunt_rows: the result was not fetched to the host");
  u32 longest = 0;
  for (u64 h = 0; h < r->nhits; ++h) longest = std::max<u32>(longest, r->hits[h].aln_len);
  const u32 stride = (longest + 7) & ~7u;
  const u64 bytes = std::max<u64>(r->nhits * (u64)stride, 1);
  char* ra = (char*)std::malloc(bytes);
  char* qa = (char*)std::malloc(bytes);
  if (!ra || !qa) {
    std::free(ra);
    std::free(qa);
    return fail(DG_ENOMEM, "dg_hunt_rows: %llu bytes", (unsigned long long)(2 * bytes));
  }
  std::atomic<int> bad{0};
  auto work = [&](u64 h0, u64 h1) {
    for (u64 h = h0; h < h1; ++h) {
      const dg_hit& H = r->hits[h];
      const u64 q0 = r->qoff[H.query];
      if (dg_hit_rows(&H, r->ops ? r->ops + h * r->ops_per_hit : nullptr, r->ops_per_hit, r->qseq + q0, (u32)(r->qoff[H.query + 1] - q0),
                      ra + h * stride, qa + h * stride) != DG_OK)
        bad.store(1);
    }
  };
  unsigned nt = std::min<unsigned>(std::max(1u, std::thread::hardware_concurrency()), 16u);
  if (r->nhits < 65536) nt = 1;
  if (nt <= 1) work(0, r->nhits);
  else {
    std::vector<std::thread> pool;
    const u64 per = (r->nhits + nt - 1) / nt;
    for (unsigned t = 0; t < nt; ++t) pool.emplace_back(work, std::min<u64>(r->nhits, t * per), std::min<u64>(r->nhits, (t + 1) * per));
    for (auto& t : pool) t.join();
  }
  if (bad.load()) {
    std::free(ra);
    std::free(qa);
    return fail(DG_EINVAL, "dg_hunt_rows: a hit's alignment description does not fit its query");
  }
  r->refalign = ra;
  r->queryalign = qa;
  r->aln_stride = stride;
  return DG_OK;
}

int dg_chit_unpack(const dg_hunt_result* r, uint64_t h, uint32_t query, dg_hit* out, const uint32_t** ops) {
  if (!r || !out || !r->compact || !r->chits || !r->seq_start || h >= r->nhits) return fail(DG_EINVAL, "dg_chit_unpack: not a compact result, or hit out of range");
  const u32 W = DG_CHIT_WORDS(r->ops_per_hit);
  const u32* rec = r->chits + h * W;
  const u64 pos = rec[0];
  // hunter.h:358-362: the sequence that holds the position = the last one starting at or before it
  u32 lo = 0, hi = r->nseq - 1;
  while (lo < hi) {
    const u32 mid = (lo + hi + 1) >> 1;
    if (r->seq_start[mid] <= pos) lo = mid;
    else hi = mid - 1;
  }
  const u32 meta = rec[1];
  out->score = -(int32_t)DG_CHIT_NEG_SCORE(meta);
  out->chr = lo;
  out->start = (uint32_t)((int64_t)(pos - r->seq_start[lo]) + DG_CHIT_DELTA(meta) + 1);
  out->query = query;
  out->aln_len = (uint16_t)DG_CHIT_ALN_LEN(meta);
  out->strand = (uint8_t)DG_CHIT_STRAND(meta);
  out->reserved = 0;
  if (ops) *ops = r->ops_per_hit ? rec + 2 : nullptr;
  return DG_OK;
}

int dg_normalize_query(const uint8_t* in, uint32_t len, uint8_t* out, uint32_t* nondna) {
  if ((!in || !out) && len) return fail(DG_EINVAL, "dg_normalize_query: null argument");
  u32 bad = 0;
  for (u32 i = 0; i < len; ++i) {
    u32 ch = in[i];
    if (ch >= 'a' && ch <= 'z') ch -= 32;  // boost::to_upper_copy, hunter.h:306
    const bool dna = ch == 'A' || ch == 'C' || ch == 'G' || ch == 'T';
    bad += !dna;  // util.h:208-219: every character outside A,C,G,T is replaced, a literal 'N' included
    out[i] = (uint8_t)(dna ? ch : 'N');
  }
  if (nondna) *nondna = bad;
  return DG_OK;
}

int dg_hunt_expand(dg_hunt_result* r, const uint8_t* qbytes, const uint64_t* qoff) {
  if (!r) return fail(DG_EINVAL, "dg_hunt_expand: null result");
  if (!r->compact || r->hits || r->expanded_) return DG_OK;
  if (!qoff || (!qbytes && r->nq && qoff[r->nq])) return fail(DG_EINVAL, "dg_hunt_expand: the batch's query bytes and offsets are needed");
  if (!r->chits || !r->qinfo || !r->hit_off) return fail(DG_EINVAL, "dg_hunt_expand: the result was not fetched to the host");
  const size_t nq = r->nq;
  const u64 nh = r->nhits, total = qoff[nq];
  const u32 oph = r->ops_per_hit;
  // one allocation: hits | ops | qflags qdistance qnondna | qoff | qseq
  size_t o_hits = 0, o_ops = (o_hits + nh * sizeof(dg_hit) + 15) & ~(size_t)15, o_meta = (o_ops + nh * oph * 4 + 15) & ~(size_t)15,
         o_qoff = (o_meta + 3 * nq * 4 + 15) & ~(size_t)15, o_qseq = o_qoff + (nq + 1) * 8, bytes = o_qseq + total + 16;
  u8* blk = (u8*)std::malloc(bytes);
  if (!blk) return fail(DG_ENOMEM, "dg_hunt_expand: %zu bytes", bytes);
  dg_hit* hits = (dg_hit*)(blk + o_hits);
  u32* ops = (u32*)(blk + o_ops);
  u32* qflags = (u32*)(blk + o_meta);
  u32 *qdist = qflags + nq, *qnondna = qdist + nq;
  u64* qo = (u64*)(blk + o_qoff);
  u8* qseq = blk + o_qseq;
  std::memcpy(qo, qoff, (nq + 1) * 8);
  std::atomic<int> bad{0};
  auto work = [&](size_t q0, size_t q1) {
    for (size_t q = q0; q < q1; ++q) {
      const u32 w = r->qinfo[q];
      qflags[q] = DG_QINFO_FLAGS(w);
      qdist[q] = DG_QINFO_DISTANCE(w);
      u32 nd = 0;
      (void)dg_normalize_query(qbytes + qoff[q], (u32)(qoff[q + 1] - qoff[q]), qseq + qoff[q], &nd);
      qnondna[q] = DG_QINFO_NONDNA(w);
      if ((nd & 0xFFFFu) != qnondna[q]) bad.store(1);  // the caller's bytes are not the ones the batch was run with
      for (u64 h = r->hit_off[q]; h < r->hit_off[q + 1]; ++h) {
        const u32* po = nullptr;
        if (dg_chit_unpack(r, h, (u32)q, &hits[h], &po) != DG_OK) bad.store(1);
        for (u32 k = 0; k < oph; ++k) ops[h * oph + k] = po[k];
      }
    }
  };
  unsigned nt = std::min<unsigned>(std::max(1u, std::thread::hardware_concurrency()), 16u);
  if (nq < 4096) nt = 1;
  if (nt <= 1) work(0, nq);
  else {
    std::vector<std::thread> pool;
    const size_t per = (nq + nt - 1) / nt;
    for (unsigned t = 0; t < nt; ++t) pool.emplace_back(work, std::min(nq, t * per), std::min(nq, (t + 1) * per));
    for (auto& t : pool) t.join();
  }
  if (bad.load()) {
    std::free(blk);
    return fail(DG_EINVAL, "dg_hunt_expand: the query bytes do not match the batch this result belongs to");
  }
  r->expanded_ = blk;
  r->hits = hits;
  r->ops = oph ? ops : nullptr;
  r->qflags = qflags;
  r->qdistance = qdist;
  r->qnondna = qnondna;
  r->qoff = qo;
  r->qseq = qseq;
  return DG_OK;
}

// dg_hunt's body.  prestaged != nullptr: the queries already lie in that pinned block (bytes at 0, offsets at (total + 63) & ~63 —
// dg_hunt_submit copied them there once, and qbytes / qoff point into it).
static int hunt_host(dg_index* ix, const dg_hunt_params* p, const uint32_t* seqlen, uint32_t nseq, const uint8_t* qbytes,
                     const uint64_t* qoff, size_t nq, dg_hunt_result** out, PinnedBlock* prestaged) {
  if (!ix || !p || !seqlen || !qoff || !out || (!qbytes && nq && qoff[nq])) return fail(DG_EINVAL, "dg_hunt: null argument");
  *out = nullptr;
  if (!nq) return fail(DG_EINVAL, "dg_hunt: empty batch");
  u64 total = qoff[nq];
  u32 maxlen = p->max_query_len;  // the caller's bound: the kernels count the queries above it (k_prepare) and the batch fails if any
  if (!maxlen) {
    for (size_t i = 0; i < nq; ++i) {
      if (qoff[i + 1] < qoff[i]) return fail(DG_EINVAL, "dg_hunt: qoff must be non-decreasing");
      u64 l = qoff[i + 1] - qoff[i];
      if (l > 0xFFFFFFu) return fail(DG_ELIMIT, "query %zu is too long", i);
      maxlen = std::max<u32>(maxlen, (u32)l);
    }
  }
  DG_HIP(hipSetDevice(ix->device));
  DG_TRY(ix->ws[WS_QB].reserve(total + 8));
  DG_TRY(ix->ws[WS_QOFF].reserve((nq + 1) * 8));
  // the queries go up through a pinned block of the pool: copies from pageable memory are staged by the runtime one by one and
  // held up the other handle's stream as well (r03: 3.8 ms per batch with two batches in flight)
  const u64 o_off = (total + 63) & ~63ull;
  PinnedBlock* stage = prestaged ? prestaged : pinned_pool().get(o_off + (nq + 1) * 8 + 64);
  if (stage && !prestaged) {
    if (total) std::memcpy(stage->p, qbytes, total);
    std::memcpy((u8*)stage->p + o_off, qoff, (nq + 1) * 8);
  }
  const void* src_q = stage ? stage->p : (const void*)qbytes;
  const void* src_o = stage ? (const void*)((u8*)stage->p + o_off) : (const void*)qoff;
  int rc = DG_OK;
  if (total && hipMemcpyAsync(ix->ws[WS_QB].p, src_q, total, hipMemcpyHostToDevice, ix->stream) != hipSuccess) rc = fail(DG_EHIP, "dg_hunt: query upload failed");
  if (rc == DG_OK && hipMemcpyAsync(ix->ws[WS_QOFF].p, src_o, (nq + 1) * 8, hipMemcpyHostToDevice, ix->stream) != hipSuccess)
    rc = fail(DG_EHIP, "dg_hunt: offset upload failed");
  if (rc == DG_OK) rc = run_batch(ix, p, seqlen, nseq, ix->ws[WS_QB].p, ix->ws[WS_QOFF].p, nq, total, maxlen, 1, out, nullptr, nullptr, qbytes, qoff);
  if (stage) {
    (void)hipStreamSynchronize(ix->stream);  // (run_batch has synchronised already unless it failed early)
    if (!prestaged) pinned_pool().put(stage);
  }
  if (rc != DG_OK && *out) {
    dg_hunt_result_free(*out);
    *out = nullptr;
  }
  return rc;
}

int dg_hunt(dg_index* ix, const dg_hunt_params* p, const uint32_t* seqlen, uint32_t nseq, const uint8_t* qbytes,
            const uint64_t* qoff, size_t nq, dg_hunt_result** out) {
  // a submitted batch owns the lane's stream, workspaces and hints until its ticket has been waited for
  if (ix && ix->busy.load()) return fail(DG_EINVAL, "dg_hunt: a dg_hunt_submit batch is in flight on this handle (dg_hunt_wait first)");
  return hunt_host(ix, p, seqlen, nseq, qbytes, qoff, nq, out, nullptr);
}

// Asynchronous form of dg_hunt (ABI 4).  The batch is driven by the handle's helper thread (created at the first submit, alive
// until the handle closes — a fresh thread per batch paid HIP's per-thread start-up every time: 2.5 ms per 100 000-query batch in
// the first r03 measurement): copies, kernels, the batch's one synchronisation, the results into a pinned block.  The caller goes
// on — typically formatting the previous batch — and collects with dg_hunt_wait.  One batch per handle at a time; a second handle
// from dg_index_share runs its batch concurrently on its own stream, which is how a single-threaded host keeps two in flight.
struct dg_hunt_ticket {
  dg_index* ix = nullptr;
  dg_hunt_params p;
  std::vector<uint32_t> seqlen;
  PinnedBlock* stage = nullptr;  // the queries, copied once at submit: bytes at 0, offsets at off_at
  uint64_t off_at = 0;
  size_t nq = 0;
  int rc = DG_OK;
  std::string err;
  dg_hunt_result* res = nullptr;
  bool done = false;
};
struct dg_index::Worker {
  std::thread th;
  std::mutex mu;
  std::condition_variable cv;
  dg_hunt_ticket* job = nullptr;  // handed over, not yet taken
  bool quit = false;
  void loop() {
    for (;;) {
      dg_hunt_ticket* t = nullptr;
      {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return job || quit; });
        if (!job) return;
        t = job;
        job = nullptr;
      }
      const bool tm = std::getenv("DICEY_TIMING") && std::atoi(std::getenv("DICEY_TIMING")) >= 2;
      const double t_take = tm ? host_us() : 0.0;
      t->rc = hunt_host(t->ix, &t->p, t->seqlen.data(), (uint32_t)t->seqlen.size(), (const uint8_t*)t->stage->p,
                        (const uint64_t*)((const uint8_t*)t->stage->p + t->off_at), t->nq, &t->res, t->stage);
      if (t->rc != DG_OK) t->err = dg_last_error();  // the message is thread-local: carried over to the waiting thread
      if (tm) std::fprintf(stderr, "dicey timing: worker %p took its batch at %.0f us, done at %.0f us\n", (void*)t->ix, std::fmod(t_take, 1e8), std::fmod(host_us(), 1e8));
      {
        std::lock_guard<std::mutex> lk(mu);
        t->done = true;
      }
      cv.notify_all();
    }
  }
};
void dg_index::stop_worker() {
  if (!worker) return;
  {
    std::lock_guard<std::mutex> lk(worker->mu);
    worker->quit = true;
  }
  worker->cv.notify_all();
  if (worker->th.joinable()) worker->th.join();
  delete worker;
  worker = nullptr;
}

int dg_hunt_submit(dg_index* ix, const dg_hunt_params* p, const uint32_t* seqlen, uint32_t nseq, const uint8_t* qbytes,
                   const uint64_t* qoff, size_t nq, dg_hunt_ticket** out) {
  if (!ix || !p || !seqlen || !qoff || !out || (!qbytes && nq && qoff[nq])) return fail(DG_EINVAL, "dg_hunt_submit: null argument");
  *out = nullptr;
  if (!nq) return fail(DG_EINVAL, "dg_hunt_submit: empty batch");
  const bool tm = std::getenv("DICEY_TIMING") && std::atoi(std::getenv("DICEY_TIMING")) >= 2;
  const double t_sub = tm ? host_us() : 0.0;
  // two lanes per handle (ABI 5): the handle itself while it is idle, else its internal second lane (a shared handle: own stream,
  // workspaces and helper thread on the same resident index; created at the first need).  A caller that keeps one batch in flight
  // never leaves the first lane.
  dg_index* const owner = ix;
  if (ix->busy.exchange(true)) {
    if (!owner->lane2 && dg_index_share(owner, &owner->lane2) != DG_OK) owner->lane2 = nullptr;
    ix = owner->lane2;
    if (!ix || ix->busy.exchange(true))
      return fail(DG_EINVAL, "dg_hunt_submit: two batches are already in flight on this handle (wait for the older ticket first)");
  }
  dg_hunt_ticket* t = nullptr;
  try {
    t = new dg_hunt_ticket;
    t->ix = ix;
    t->p = *p;
    t->seqlen.assign(seqlen, seqlen + nseq);
    t->nq = nq;
    t->off_at = (qoff[nq] + 63) & ~63ull;
    t->stage = pinned_pool().get(t->off_at + (nq + 1) * 8 + 64);  // the caller's buffers are free again when this call returns
    if (!t->stage) {
      delete t;
      ix->busy.store(false);
      return fail(DG_ENOMEM, "dg_hunt_submit: no pinned memory for %llu query bytes", (unsigned long long)qoff[nq]);
    }
    if (qoff[nq]) std::memcpy(t->stage->p, qbytes, qoff[nq]);
    std::memcpy((uint8_t*)t->stage->p + t->off_at, qoff, (nq + 1) * 8);
    if (!ix->worker) {
      dg_index::Worker* w = new dg_index::Worker;
      try {
        w->th = std::thread([w] { w->loop(); });
      } catch (...) {  // no thread: no worker either, or later submissions would wait for one that never runs
        delete w;
        throw;
      }
      ix->worker = w;
    }
    {
      std::lock_guard<std::mutex> lk(ix->worker->mu);
      ix->worker->job = t;
    }
    ix->worker->cv.notify_all();
  } catch (const std::exception& e) {
    if (t && t->stage) pinned_pool().put(t->stage);
    delete t;
    ix->busy.store(false);
    return fail(DG_ENOMEM, "dg_hunt_submit: %s", e.what());
  }
  *out = t;
  if (tm) std::fprintf(stderr, "dicey timing: submit to %p from %.0f to %.0f us\n", (void*)ix, std::fmod(t_sub, 1e8), std::fmod(host_us(), 1e8));
  return DG_OK;
}

int dg_hunt_wait(dg_hunt_ticket* t, dg_hunt_result** out) {
  if (!t || !out) return fail(DG_EINVAL, "dg_hunt_wait: null argument");
  dg_index* ix = t->ix;
  const bool tm = std::getenv("DICEY_TIMING") && std::atoi(std::getenv("DICEY_TIMING")) >= 2;
  const double t_w = tm ? host_us() : 0.0;
  {
    std::unique_lock<std::mutex> lk(ix->worker->mu);
    ix->worker->cv.wait(lk, [&] { return t->done; });
  }
  ix->busy.store(false);
  if (tm) std::fprintf(stderr, "dicey timing: wait on %p from %.0f to %.0f us\n", (void*)ix, std::fmod(t_w, 1e8), std::fmod(host_us(), 1e8));
  *out = t->res;
  const int rc = t->rc;
  const std::string err = t->err;
  if (t->stage) pinned_pool().put(t->stage);
  delete t;
  if (rc != DG_OK) return fail(rc, "%s", err.c_str());
  return DG_OK;
}

int dg_neighborhood_count(dg_index* ix, uint32_t distance, int hamming, uint32_t max_neighborhood, const uint8_t* qbytes,
                           const uint64_t* qoff, size_t nq, uint64_t* fw_count, uint64_t* rv_count) {
  if (!ix || !qoff || !fw_count || !rv_count || (!qbytes && nq && qoff[nq])) return fail(DG_EINVAL, "dg_neighborhood_count: null argument");
  if (!nq) return DG_OK;
  // Sequences of A/C/G/T go through the search kernel (count mode of run_batch).  Anything else in a sequence (N, IUPAC
  // letters: the reference matches them byte for byte) is rare enough for the host: enumerate the neighbourhood as
  // neighbors.h defines it and count every string with dg_count.
  std::vector<size_t> fast, slow;
  u32 maxlen = 0;
  for (size_t i = 0; i < nq; ++i) {
    if (qoff[i + 1] < qoff[i]) return fail(DG_EINVAL, "dg_neighborhood_count: qoff must be non-decreasing");
    const u64 l = qoff[i + 1] - qoff[i];
    // hunt skips queries under 10 nt and clamps the distance to the length (hunter.h:299-315); padlock.h does neither
    if (l < 10 || l <= distance) return fail(DG_ELIMIT, "sequence %zu has %llu nt; this entry point takes >= 10 nt and more than `distance`", i, (unsigned long long)l);
    if (l > MAX_QLEN) return fail(DG_ELIMIT, "sequence %zu exceeds %u nt", i, MAX_QLEN);
    bool plain = true;
    for (u64 k = qoff[i]; k < qoff[i + 1]; ++k) {
      const u8 ch = qbytes[k];
      plain = plain && (ch == 'A' || ch == 'C' || ch == 'G' || ch == 'T');
    }
    (plain ? fast : slow).push_back(i);
    maxlen = std::max<u32>(maxlen, (u32)l);
  }
  if (distance > DMAX) return fail(DG_ELIMIT, "distance %u exceeds the supported maximum of %u", distance, DMAX);
  DG_HIP(hipSetDevice(ix->device));
  if (!fast.empty()) {
    std::string buf;
    std::vector<u64> off(1, 0);
    for (size_t i : fast) {
      buf.append((const char*)qbytes + qoff[i], qoff[i + 1] - qoff[i]);
      off.push_back(buf.size());
    }
    DG_TRY(ix->ws[WS_QB].reserve(buf.size() + 8));
    DG_TRY(ix->ws[WS_QOFF].reserve(off.size() * 8));
    DG_HIP(hipMemcpyAsync(ix->ws[WS_QB].p, buf.data(), buf.size(), hipMemcpyHostToDevice, ix->stream));
    DG_HIP(hipMemcpyAsync(ix->ws[WS_QOFF].p, off.data(), off.size() * 8, hipMemcpyHostToDevice, ix->stream));
    dg_hunt_params hp{};
    hp.distance = distance;
    hp.hamming = hamming;
    hp.forward_only = 0;
    hp.max_locations = 1;
    hp.max_neighborhood = max_neighborhood;
    const uint32_t one_seq = 1;  // chromosome lookup is not used in count mode
    std::vector<u64> counts(2 * fast.size());
    dg_hunt_result* hr = nullptr;
    int rc = run_batch(ix, &hp, &one_seq, 1, ix->ws[WS_QB].p, ix->ws[WS_QOFF].p, fast.size(), buf.size(), maxlen, 0, &hr, nullptr, counts.data(),
                       (const uint8_t*)buf.data(), off.data());
    if (hr) dg_hunt_result_free(hr);
    if (rc != DG_OK) return rc;
    for (size_t k = 0; k < fast.size(); ++k) {
      fw_count[fast[k]] = counts[2 * k];
      rv_count[fast[k]] = counts[2 * k + 1];
    }
  }
  if (!slow.empty()) {
    std::string buf;
    std::vector<u64> off(1, 0);
    std::vector<std::pair<size_t, size_t>> owner;  // (sequence, strand) of every string
    for (size_t i : slow) {
      std::string fw((const char*)qbytes + qoff[i], qoff[i + 1] - qoff[i]), rv(fw.rbegin(), fw.rend());
      for (char& ch : rv) ch = complement_iupac(ch);
      for (size_t strand = 0; strand < 2; ++strand) {
        bool fired = false;  // the sums are taken over whatever the (possibly capped) reference set holds
        for (const std::string& s : CappedNeighborhood::enumerate(strand ? rv : fw, distance, !hamming, max_neighborhood, fired)) {
          buf += s;
          off.push_back(buf.size());
          owner.emplace_back(i, strand);
        }
      }
      fw_count[i] = rv_count[i] = 0;
    }
    std::vector<u64> cnt(owner.size());
    int rc = dg_count(ix, (const uint8_t*)buf.data(), off.data(), owner.size(), cnt.data());
    if (rc != DG_OK) return rc;
    for (size_t k = 0; k < owner.size(); ++k) (owner[k].second ? rv_count : fw_count)[owner[k].first] += cnt[k];
  }
  return DG_OK;
}

int dg_neighbors(const uint8_t* seq, uint32_t len, uint32_t distance, int hamming, uint32_t max_neighborhood, char** out,
                 uint64_t* count, int* cap_fired) {
  if (!seq || !out) return fail(DG_EINVAL, "dg_neighbors: null argument");
  *out = nullptr;
  if (len == 0 || len > 0xFFFFFFu) return fail(DG_EINVAL, "dg_neighbors: sequence length %u", len);
  // the enumeration is exponential in the distance and bounded only by the cap: keep to what the hunt path itself supports
  if (distance > DMAX) return fail(DG_ELIMIT, "dg_neighbors: distance %u exceeds the supported maximum of %u", distance, DMAX);
  if (distance >= len) return fail(DG_EINVAL, "dg_neighbors: distance %u must be smaller than the sequence length %u (hunter.h:312-315 clamps it first)", distance, len);
  if (max_neighborhood > (1u << 24)) return fail(DG_ELIMIT, "dg_neighbors: max_neighborhood %u exceeds 2^24", max_neighborhood);
  try {
    bool fired = false;
    const std::vector<std::string> set = CappedNeighborhood::enumerate(std::string((const char*)seq, len), distance, !hamming, max_neighborhood, fired);
    size_t bytes = 1;
    for (const std::string& s : set) bytes += s.size() + 1;
    char* buf = (char*)std::malloc(bytes);
    if (!buf) return fail(DG_ENOMEM, "dg_neighbors: out of memory");
    char* w = buf;
    for (const std::string& s : set) {
      std::memcpy(w, s.data(), s.size());
      w += s.size();
      *w++ = '\n';
    }
    *w = 0;
    *out = buf;
    if (count) *count = set.size();
    if (cap_fired) *cap_fired = fired;
  } catch (const std::bad_alloc&) {
    return fail(DG_ENOMEM, "dg_neighbors: out of memory");
  } catch (const std::exception& e) {
    return fail(DG_EINVAL, "dg_neighbors: %s", e.what());
  }
  return DG_OK;
}
void dg_buffer_free(void* p) { std::free(p); }

int dg_hunt_device(dg_index* ix, const dg_hunt_params* p, const uint32_t* seqlen, uint32_t nseq, const void* d_qbytes,
                   const void* d_qoff, size_t nq, uint64_t total_qbytes, int fetch, dg_hunt_result** out) {
  if (!ix || !p || !seqlen || !d_qbytes || !d_qoff || !out) return fail(DG_EINVAL, "dg_hunt_device: null argument");
  *out = nullptr;
  if (!nq) return fail(DG_EINVAL, "dg_hunt_device: empty batch");
  if (ix->busy.load()) return fail(DG_EINVAL, "dg_hunt_device: a dg_hunt_submit batch is in flight on this handle (dg_hunt_wait first)");
  DG_HIP(hipSetDevice(ix->device));
  // Query lengths are needed on the host only to size buffers and to check the supported envelope.  A repeated call with the
  // same offsets buffer, count and byte total reuses the previous maximum as an upper bound instead of reading the offsets
  // back (a 0.1 ms host round trip per batch); k_prepare counts queries that exceed the bound, in which case the batch is
  // repeated with the offsets read afresh.
  u32 maxlen = 0;
  dg_index::QoffSeen* hit = nullptr;
  for (auto& e : ix->seen)
    if (e.qoff == d_qoff && e.nq == nq && e.total == total_qbytes && e.maxlen > 0) hit = &e;
  const bool cached = hit != nullptr && !p->max_query_len;
  if (p->max_query_len) {
    maxlen = p->max_query_len;  // no read-back at all: k_prepare counts queries above the caller's bound and the batch fails if any
    hit = nullptr;
  } else if (cached) maxlen = hit->maxlen;
  else {
    std::vector<u64> hoff(nq + 1);
    DG_HIP(hipMemcpyAsync(hoff.data(), d_qoff, (nq + 1) * 8, hipMemcpyDeviceToHost, ix->stream));
    DG_HIP(hipStreamSynchronize(ix->stream));
    if (hoff[nq] != total_qbytes) return fail(DG_EINVAL, "dg_hunt_device: total_qbytes does not match qoff[nq]");
    for (size_t i = 0; i < nq; ++i) {
      if (hoff[i + 1] < hoff[i]) return fail(DG_EINVAL, "dg_hunt_device: qoff must be non-decreasing");
      u64 l = hoff[i + 1] - hoff[i];
      if (l > 0xFFFFFFu) return fail(DG_ELIMIT, "query %zu is too long", i);
      maxlen = std::max<u32>(maxlen, (u32)l);
    }
    hit = &ix->seen[ix->seen_next++ % dg_index::NSEEN];
    hit->qoff = d_qoff;
    hit->nq = nq;
    hit->total = total_qbytes;
    hit->maxlen = maxlen;
  }
  int rc = run_batch(ix, p, seqlen, nseq, d_qbytes, d_qoff, nq, total_qbytes, maxlen, fetch, out);
  if (rc != DG_OK && *out) {
    dg_hunt_result_free(*out);
    *out = nullptr;
  }
  if (rc != DG_OK && hit) hit->qoff = nullptr;  // whatever went wrong, the next call reads the offsets again
  if (rc == DG_EINVAL && cached) return dg_hunt_device(ix, p, seqlen, nseq, d_qbytes, d_qoff, nq, total_qbytes, fetch, out);
  return rc;
}

}  // extern "C"

