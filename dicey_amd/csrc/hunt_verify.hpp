// Verify stage of the hunt pipeline (included by hunt.hip only): chromosome lookup, context window, '\n' trimming, needle() with the
// reference's tie rules, gap stripping -> hits in reference push order (hunter.h:358-429, needle.h:59-138).
#pragma once
#include "hunt_locate.hpp"
#include "band_bits.hpp"

namespace dg {

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
  const u64* selkey;      // != nullptr (r06, batches at distance <= 1): a hit whose seed says so (SEED_KEY_VALID) is aligned from its kept string's
                          // codes (FlatSel::key at the seed's slot) and the seed's context codes, without a line of the text
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
  // (r04: also on the lane-per-hit path — five dependent global loads per hit for the binary search were a fifth of that kernel's
  //  chain; one barrier is cheaper)
  const bool cum_in_lds = a.nseq <= 512;
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
  u32 sctx[CH];  // the seed's length word as the locate stage left it: bit 31 = the hit brought its context along (devfm.hpp, FmView::sax)
#pragma unroll
  for (int j = 0; j < CH; ++j) {
    const u64 h = base + (u32)j * 256u + tid;
    const uint4 v = h < nh ? *reinterpret_cast<const uint4*>(a.seeds + h) : make_uint4(0, 0, 0, 0);
    sctx[j] = v.z;
    sd[j] = HitSeed{v.x, v.y, v.z & SEED_LEN_MASK, v.w};
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
    if (SHARE && h < nh && !(a.debug & 2u) && (sctx[j] & SEED_CTX_VALID)) {
      // r06: the characters came with the suffix-array entry the locate kernel read (all four exist and are A/C/G/T, or the bit
      // would not be set): no text line for this hit
      const u32 c = sctx[j] >> 20;
      auto asc = [](u32 code) -> u32 { return (0x54474341u >> (8u * (code & 3u))) & 255u; };  // A C G T
      if (d >= 1) x |= asc(c) | (asc(c >> 4) << 16);
      if (DS >= 2 && d >= 2) x |= (asc(c >> 2) << 8) | (asc(c >> 6) << 24);
    } else if (SHARE && h < nh && !(a.debug & 2u)) {  // (the lane-per-hit path takes its context from the window band_align loads anyway)
      if (d >= 1 && loc >= 1) x |= (u32)f.text[loc - 1];
      if (DS >= 2 && d >= 2 && loc >= 2) x |= (u32)f.text[loc - 2] << 8;
      if (d >= 1 && endp + 1 <= f.n) x |= (u32)f.text[endp] << 16;
      if (DS >= 2 && d >= 2 && endp + 2 <= f.n) x |= (u32)f.text[endp + 1] << 24;
    }
    fl[j] = x;
  }
  if (SHARE || cum_in_lds) __syncthreads();  // the table is clear, the sequence starts are in LDS
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
  auto align = [&](const HitSeed& s0, const u32 lenword) -> AlnRes {
    const u64 q = s0.qs >> 1;
    const uint4 pq = b.gpeq[s0.qs];
    const u32 d = b.indel ? b.qdist[q] : 0u;
    KeyWindow kw{0ULL, 0u, 0u};
    bool from_key = false;
    if (a.selkey && (lenword & SEED_KEY_VALID) && b.indel && d <= 1u && !(a.debug & 2u)) {
      kw.key = a.selkey[s0.sel];
      kw.pre = (lenword >> 20) & 3u;
      kw.post = (lenword >> 24) & 3u;
      from_key = true;
    }
    return band_align_bits<WB, TR, 256>(f.text, f.n, b.indel != 0, (u64)s0.pos, s0.len, b.qlen[q], d, PosMasks{pq.x, pq.y, pq.z, pq.w}, tr, win, fault,
                                        from_key, kw);
  };
  if (!SHARE) {
    cls[0] = tid;
    if (base + tid < nh && !(a.debug & 1u)) {
      const AlnRes r = align(sd[0], sctx[0]);
      if (r.pre_eff < cpos[0]) cpos[0] -= r.pre_eff;  // hunter.h:382 (strict <)
      cls_info[tid] = r.info;
      cls_ops[tid * DS] = r.op[0];
      if (DS > 1) cls_ops[tid * DS + 1] = r.op[1];
    }
  } else {
    for (u32 c = tid; c < ncls && !(a.debug & 1u); c += 256) {
      const u32 own = cls_owner[c];
      const uint4 v = *reinterpret_cast<const uint4*>(a.seeds + base + own);
      const AlnRes r = align(HitSeed{v.x, v.y, v.z & SEED_LEN_MASK, v.w}, v.z);
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

}  // namespace dg
