// `dicey search` on the GPU: the per-hit block of reference src/silica.h:474-532 as a kernel (k_site) that replaces the
// verify stage of the hunt pipeline, and dg_search_sites (silica.h:429-573 for a batch of primers).
#include <algorithm>
#include <set>

#include "hunt_internal.hpp"
#include "thal_wave.hpp"

namespace dg {

// `dicey search` per-hit block (reference src/silica.h:474-532): context window with the primer's 5' overhang
// (koffset), '\n' trimming, thal(primer, window) -> Tm, and — when Tm passes the cut — the Needleman-Wunsch of the
// window against the searched k-mer to find the position that de-duplicates hits.  One lane per located hit.
struct SiteArgs {
  const HitSeed* seeds;
  const u64* nhits;
  u64 hit_cap, first, count;  // this launch handles hits [first, first+count)
  const u64* cum;
  u32 nseq;
  const thal::Tables* tables;
  thal::Env env;
  const u8* pfw;   // full primers as codes 0..4, concatenated; prv = reverse complements
  const u8* prv;
  const u64* poff;
  const u32* koff;  // primer length - k
  double cut_temp;
  SiteRaw* out;
  u8* windows;
  u32 win_stride;
  double* dp;      // per launch lane: 2 * dp_stride doubles, interleaved per wavefront
  u64 dp_stride;   // cells per plane = max primer length * dp_row
  u32 dp_row;      // common row length (largest window of the batch)
  u32 only_flagged;  // k_site: recompute only the hits the wave kernel handed back (SiteRaw.pad == 1)
  u32 wave_len1, wave_stride, wave_bytes;  // k_site_wave: LDS table shape per wavefront
  u32 force_redo;    // debugging aid: hand every hit back to k_site
};
DG_DEV bool codes_self_complementary(const u8* s, u32 n, bool ascii) {  // symmetry_thermo, thal.h:1976-2010
  if (n & 1) return false;
  for (u32 i = 0; i < n / 2; ++i) {
    u32 a = ascii ? code_of_byte(s[i]) : s[i], c = ascii ? code_of_byte(s[n - 1 - i]) : s[n - 1 - i];
    if (a < 4 || c < 4) {
      if (a > 3 || c > 3 || a + c != 3) return false;
    }
  }
  return true;
}
// silica.h:519-532: needle(window, searched k-mer); leading columns whose k-mer row is a gap shift the position.
// Those columns are exactly the vertical moves taken in column 0, so the traceback stops when it gets there and
// returns the row it arrives at.
template <u32 TRACE_WORDS>
DG_DEV u32 site_lead_rows(const u8* g, u32 mg, const u8* qseq, u32 n) {
  int s[MAX_QLEN + 1];
  u64 trace[TRACE_WORDS];
  const u32 mf = n + 1;
  for (u32 w = 0; w < TRACE_WORDS; ++w) trace[w] = 0;
  s[0] = 0;
  for (u32 col = 1; col <= n; ++col) s[col] = -(int)col;
  for (u32 row = 1; row <= mg; ++row) {
    int diag = 0;
    const u8 gc = g[row - 1];
    for (u32 col = 1; col <= n; ++col) {
      int up = s[col];
      int dsc = diag + (gc == ascii_of(qseq[col - 1]) ? 0 : -1);
      int vsc = up + (col == n ? 0 : -1);
      int hsc = s[col - 1] - 1;
      int best = dsc > vsc ? dsc : vsc;
      best = best > hsc ? best : hsc;
      s[col] = best;
      u32 cell = row * mf + col;
      if (best == hsc) trace[cell >> 5] |= 1ULL << ((cell & 31) * 2);
      else if (best == vsc) trace[cell >> 5] |= 2ULL << ((cell & 31) * 2);
      diag = up;
    }
  }
  u32 row = mg, col = n;
  while (col > 0) {
    u32 tr = 1;  // row 0: horizontal
    if (row > 0) {
      u32 cell = row * mf + col;
      tr = (u32)(trace[cell >> 5] >> ((cell & 31) * 2)) & 3;
    }
    if (tr == 1) --col;
    else if (tr == 2) --row;
    else {
      --row;
      --col;
    }
  }
  return row;
}

// LDS_TABLES: the 45 KB of nearest-neighbour tables are staged in LDS once per workgroup (every loop step of thal reads
// about ten of them).  DP planes are interleaved across the 64 lanes of a wavefront with a common row length, so lanes
// working on the same (i,j) cell read one contiguous 512-byte run instead of 64 scattered lines.
template <u32 TRACE_WORDS, bool LDS_TABLES>
__global__ void __launch_bounds__(256) k_site(FmView f, Batch b, SiteArgs a, Counters* ctr) {
  __shared__ thal::Tables lds_tables;
  const thal::Tables* tabs = a.tables;
  if (LDS_TABLES && a.only_flagged) {  // nothing handed back to this workgroup: skip staging the tables
    const u64 hh = a.first + (u64)blockIdx.x * blockDim.x + threadIdx.x;
    const bool want = !ctr->overflow && *a.nhits <= a.hit_cap && hh < a.first + a.count && hh < *a.nhits && a.out[hh].pad == 1;
    if (!__syncthreads_or(want)) return;
  }
  if (LDS_TABLES) {
    const u64* src = reinterpret_cast<const u64*>(a.tables);
    u64* dst = reinterpret_cast<u64*>(&lds_tables);
    for (u32 k = threadIdx.x; k < sizeof(thal::Tables) / 8; k += blockDim.x) dst[k] = src[k];
    __syncthreads();
    tabs = &lds_tables;
  }
  const u64 lane = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  const u64 h = a.first + lane;
  const u64 nh = *a.nhits;
  if (ctr->overflow || nh > a.hit_cap || lane >= a.count || h >= nh) return;
  if (a.only_flagged && a.out[h].pad != 1) return;
  const HitSeed sd = a.seeds[h];
  const u64 q = sd.qs >> 1;
  const u32 strand = sd.qs & 1;
  const u64 loc = sd.pos;
  const u32 mlen = sd.len, koff = a.koff[q];
  u32 lo_r = 0, hi_r = a.nseq - 1;
  while (lo_r < hi_r) {
    u32 mid = (lo_r + hi_r + 1) >> 1;
    if (a.cum[mid] <= loc) lo_r = mid;
    else hi_r = mid - 1;
  }
  const u32 ref = lo_r;
  u32 chrpos = (u32)(loc - a.cum[ref]);
  u64 pre = b.indel ? b.qdist[q] : 0, post = pre;  // silica.h:480-483: the overhang is on the 5' side of the primer
  if (strand) post += koff;
  else pre += koff;
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
  if (pre_eff <= chrpos) chrpos -= pre_eff;  // silica.h:501 (non-strict)
  u8* win = a.windows + h * a.win_stride;
  for (u32 i = 0; i < mg && i < a.win_stride; ++i) win[i] = g[i];
  // thal(oligo1 = reverse complement of the primer for forward hits / the primer for reverse hits, oligo2 = window)
  const u64 p0 = a.poff[q];
  const u32 plen = (u32)(a.poff[q + 1] - p0);
  const u8* prim = (strand ? a.pfw : a.prv) + p0;
  SiteRaw r;
  r.ref = ref;
  r.glen = mg;
  r.qs = sd.qs;
  r.pad = 0;
  r.temp = -thal::kInf;
  if (!(plen > (u32)thal::kMaxAlign && mg > (u32)thal::kMaxAlign) && plen <= 64 && mg <= 320) {
    const bool sym = codes_self_complementary(prim, plen, false) && codes_self_complementary(g, mg, true);
    // wave w of this launch owns 64 * dp_stride cells; inside, cell c of lane l sits at c*64 + l
    thal::Cell* cells = reinterpret_cast<thal::Cell*>(a.dp) + (lane >> 6) * (64 * a.dp_stride) + (lane & 63);
    thal::Result tr;
    if (plen <= (u32)thal::kPackedMax && mg <= (u32)thal::kPackedMax) {  // sequences in registers
      thal::PackedSeq fa, fb;
      fa.set(0, 4);
      fa.set((int)plen + 1, 4);
      for (u32 i = 0; i < plen; ++i) fa.set((int)i + 1, prim[i]);
      fb.set(0, 4);
      fb.set((int)mg + 1, 4);
      for (u32 j = 0; j < mg; ++j) {
        u32 c = code_of_byte(g[mg - 1 - j]);
        fb.set((int)j + 1, c < 4 ? c : 4);
      }
      tr = thal::end1_tm<thal::PackedSeq>(*tabs, a.env, fa, (int)plen, fb, (int)mg, sym, cells, (int)a.dp_row, 64);
    } else {
      u8 fa[66], fb[322];
      fa[0] = fa[plen + 1] = 4;
      for (u32 i = 0; i < plen; ++i) fa[i + 1] = prim[i];
      fb[0] = fb[mg + 1] = 4;
      for (u32 j = 0; j < mg; ++j) {
        u32 c = code_of_byte(g[mg - 1 - j]);
        fb[j + 1] = (u8)(c < 4 ? c : 4);
      }
      const u8* pa = fa;
      const u8* pb = fb;
      tr = thal::end1_tm<const u8*>(*tabs, a.env, pa, (int)plen, pb, (int)mg, sym, cells, (int)a.dp_row, 64);
    }
    r.temp = tr.temp;
  }
  u32 alignpos = chrpos;
  if (r.temp > a.cut_temp) {
    const u8* qseq = (strand ? b.rv : b.fw) + b.qoff[q];
    alignpos = chrpos + site_lead_rows<TRACE_WORDS>(g, mg, qseq, b.qlen[q]);
  }
  r.chrpos = chrpos;
  r.alignpos = alignpos;
  a.out[h] = r;
}

// The same alignment by one wavefront: lane c owns column c of the score row.  The horizontal-gap dependency along a
// row, best[c] = max(cand[c], best[c-1] - 1), is a prefix maximum of cand[c] + c; every lane keeps the 2-bit moves of its
// own column (row r at bits 2(r-1)), and the traceback reads them back lane by lane.  n <= 63, mg <= 64.
DG_DEV u32 wave_lead_rows(u32 mybyte, u32 g0, u32 mg, const u8* qseq, u32 n) {
  const u32 lane = threadIdx.x & 63;
  const u32 qc = (lane >= 1 && lane <= n) ? (u32)ascii_of(qseq[lane - 1]) : 0u;
  int sc = -(int)lane;  // row 0
  u64 tr0 = 0, tr1 = 0;
  for (u32 row = 1; row <= mg; ++row) {
    const u32 gc = (u32)__shfl((int)mybyte, (int)(g0 + row - 1));
    const int up = sc, diagv = __shfl_up(sc, 1);
    const int dsc = diagv + (gc == qc ? 0 : -1);
    const int vsc = up + (lane == n ? 0 : -1);
    const int cand = lane == 0 ? 0 : (dsc > vsc ? dsc : vsc);
    int t = cand + (int)lane;
    for (int off = 1; off < 64; off <<= 1) {
      const int o = __shfl_up(t, off);
      if ((int)lane >= off && o > t) t = o;
    }
    const int best = t - (int)lane;
    const int hsc = __shfl_up(best, 1) - 1;
    u64 mv = 0;
    if (lane >= 1) {
      if (best == hsc) mv = 1;
      else if (best == vsc) mv = 2;
    }
    if (row <= 32) tr0 |= mv << (2 * (row - 1));
    else tr1 |= mv << (2 * (row - 33));
    sc = lane == 0 ? 0 : best;
  }
  u32 row = mg, col = n;
  while (col > 0) {
    u32 tr = 1;  // row 0: horizontal
    if (row > 0) {
      const u64 w = row <= 32 ? tr0 : tr1;
      const u32 lo = (u32)__shfl((int)(u32)w, (int)col), hi = (u32)__shfl((int)(u32)(w >> 32), (int)col);
      const u64 wc = ((u64)hi << 32) | lo;
      tr = (u32)(wc >> (2 * ((row - 1) & 31))) & 3;
    }
    if (tr == 1) --col;
    else if (tr == 2) --row;
    else {
      --row;
      --col;
    }
  }
  return row;
}

// One WAVEFRONT per located hit: context window and thal() with the DP table in LDS (thal_wave.hpp).  Workgroups are
// persistent (the tables are staged once), wave w of the grid takes hits w, w + #waves, ...  Everything that depends
// only on the hit is wave-uniform; hits whose Tm passes the cut are aligned by the same wavefront (silica.h:519-532).  Pairs the wave formulation cannot take (an oligo longer than the LDS table) or
// hands back as ambiguous are marked pad = 1 and recomputed by k_site.
__global__ void __launch_bounds__(1024) k_site_wave(FmView f, Batch b, SiteArgs a, Counters* ctr) {
  DG_DYNAMIC_LDS(lds_raw);
  thal::Tables* tabs = reinterpret_cast<thal::Tables*>(lds_raw);
  thal::wave_header_init(lds_raw, a.tables, a.env);
  const thal::EndTables& ends = *thal::wave_end_tables(lds_raw);
  const u32 wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), wpb = blockDim.x >> 6;
  const u32 lane = threadIdx.x & 63;
  const thal::WaveMem wm =
      thal::wave_mem_at(lds_raw + thal::wave_header_bytes() + (size_t)wave * a.wave_bytes, a.wave_len1, a.wave_stride);
  const u64 nh = *a.nhits;
  if (ctr->overflow || nh > a.hit_cap) return;
  for (u64 h = (u64)blockIdx.x * wpb + wave; h < nh; h += (u64)gridDim.x * wpb) {
#ifdef DG_WAVE_PROFILE
    DG_PROF_T(tH0);
#endif
    const HitSeed sd = a.seeds[h];
    const u64 q = sd.qs >> 1;
    const u32 strand = sd.qs & 1;
    const u64 loc = sd.pos;
    const u32 mlen = sd.len, koff = a.koff[q];
    u32 lo_r = 0, hi_r = a.nseq - 1;
    while (lo_r < hi_r) {
      u32 mid = (lo_r + hi_r + 1) >> 1;
      if (a.cum[mid] <= loc) lo_r = mid;
      else hi_r = mid - 1;
    }
    const u32 ref = lo_r;
    u32 chrpos = (u32)(loc - a.cum[ref]);
    u64 pre = b.indel ? b.qdist[q] : 0, post = pre;  // silica.h:480-483
    if (strand) post += koff;
    else pre += koff;
    if (pre > loc) pre = loc;
    if (loc + mlen + post > f.n) post = f.n - loc - mlen;
    // the window stops at a '\n' on either side: lanes look at one byte each
    const u64 wlo = loc - pre, wlen = pre + mlen + post;
    const u8 mybyte = lane < wlen ? f.text[wlo + lane] : 0;
    const u64 nlmask = __ballot(lane < wlen && mybyte == '\n');
    const u64 before = nlmask & thal::low_bits((int)pre);             // newlines in the left context
    const u64 after = wlen <= 64 ? nlmask >> ((pre + mlen) & 63) : 0;  // newlines in the right context
    const u32 pre_eff = before ? (u32)pre - (64 - (u32)__builtin_clzll(before)) : (u32)pre;
    const u32 post_eff = after ? (u32)__builtin_ctzll(after) : (u32)post;
    const u32 mg = pre_eff + mlen + post_eff;
    const u32 g0 = (u32)pre - pre_eff;  // window start inside the 64 bytes the lanes hold
    if (pre_eff <= chrpos) chrpos -= pre_eff;  // silica.h:501 (non-strict)
    u8* win = a.windows + h * a.win_stride;
    if (lane >= g0 && lane < g0 + mg && lane - g0 < a.win_stride) win[lane - g0] = mybyte;
    const u64 p0 = a.poff[q];
    const u32 plen = (u32)(a.poff[q + 1] - p0);
    const u8* prim = (strand ? a.pfw : a.prv) + p0;
    SiteRaw r;
    r.ref = ref;
    r.glen = mg;
    r.qs = sd.qs;
    r.pad = 0;
    r.temp = -thal::kInf;
    r.chrpos = chrpos;
    r.alignpos = chrpos;
    if (wlen > 64 || plen > a.wave_len1 || mg > a.wave_stride || plen == 0 || mg == 0 || pre + mlen >= 64) {
      r.pad = 1;
    } else {
      // framed code sequences: lane L holds position L; words are OR-reduced over the wave
      const u32 mycode = code_of_byte(mybyte);
      const u32 gcode = mycode < 4 ? mycode : 4;
      // position L of oligo 2 (the reversed window) is window byte mg-L, held by lane g0+mg-L
      const int srcl = (int)(g0 + mg) - (int)lane;
      const u32 got = (u32)__shfl((int)gcode, srcl >= 0 && srcl < 64 ? srcl : 0);
      const u32 cb = (lane >= 1 && lane <= mg) ? got : 4u;
      const u32 ca = (lane >= 1 && lane <= plen) ? (u32)prim[lane - 1] : 4u;
      const bool ina = lane <= plen + 1, inb = lane <= mg + 1;
      thal::PlaneSeq fa, fb;
      fa.b0 = __ballot(ina && (ca & 1));
      fa.b1 = __ballot(ina && (ca & 2));
      fa.b2 = __ballot(ina && (ca & 4));
      fb.b0 = __ballot(inb && (cb & 1));
      fb.b1 = __ballot(inb && (cb & 2));
      fb.b2 = __ballot(inb && (cb & 4));
      // symmetry_thermo (thal.h:1976-2010) on both oligos
      bool sym = !(plen & 1) && !(mg & 1);
      if (sym) {
        bool bad = false;
        if (lane < plen / 2) {
          const u32 x = prim[lane], y = prim[plen - 1 - lane];
          if ((x < 4 || y < 4) && (x > 3 || y > 3 || x + y != 3)) bad = true;
        }
        if (lane < mg / 2) {
          const u32 x = (u32)fb[(int)mg - (int)lane], y = (u32)fb[(int)lane + 1];  // window[lane], window[mg-1-lane]
          if ((x < 4 || y < 4) && (x > 3 || y > 3 || x + y != 3)) bad = true;
        }
        sym = __ballot(bad) == 0;
      }
      bool amb = false;
      const thal::Result tr = thal::wave_end1_tm(*tabs, ends, a.env, fa, (int)plen, fb, (int)mg, sym, wm, (int)a.wave_stride, amb);
      r.temp = tr.temp;
      if (amb || a.force_redo) r.pad = 1;
      else if (r.temp > a.cut_temp)  // silica.h:519-532
        r.alignpos = chrpos + wave_lead_rows(mybyte, g0, mg, (strand ? b.rv : b.fw) + b.qoff[q], b.qlen[q]);
    }
    if (lane == 0) a.out[h] = r;
#ifdef DG_WAVE_PROFILE
    DG_PROF_T(tH1);
    if (lane == 0) atomicAdd(&::dg::thal::g_wave_prof[6], tH1 - tH0);
#endif
  }
}

// Only hits whose Tm passes the cut (or whose thal() was refused: the reference's error flag) matter to the host: keep flag,
// prefix sum, stable compaction of record + window, so that thousands instead of millions of records cross the bus.
__global__ void k_site_keep(const SiteRaw* raw, u64 n, double cut, u32* keep) {
  const u64 h = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (h >= n) return;
  const double t = raw[h].temp;
  keep[h] = (t > cut || t == -thal::kInf) ? 1u : 0u;
}
__global__ void k_site_compact(const SiteRaw* raw, const u8* win, u32 stride, const u32* keep, const u64* pos, u64 n, SiteRaw* oraw,
                               u8* owin) {
  const u64 h = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (h >= n || !keep[h]) return;
  const u64 at = pos[h];
  oraw[at] = raw[h];
  const uint4* src = reinterpret_cast<const uint4*>(win + h * stride);  // stride is a multiple of 16
  uint4* dst = reinterpret_cast<uint4*>(owin + at * stride);
  for (u32 k = 0; k < stride / 16; ++k) dst[k] = src[k];
}

int launch_site_stage(dg_index* ix, SearchExtra* sx, const Batch& b, const HitSeed* seeds, const u64* hit_off, u64 hit_cap,
                      const u64* cum, u32 nseq, u32 dmax_eff, u32 maxlen, Counters* ctr) {
  auto& ws = ix->ws;
  hipStream_t st = ix->stream;
  const u64 nq = b.nq;
      const u32 wmax = sx->max_primer_len + 3 * dmax_eff + 2;  // k + overhang + context on both sides + edits
      const u32 wstride = (wmax + 15) & ~15u;
      const u64 dp_stride = (u64)sx->max_primer_len * wmax;
      static const bool no_lds = exp_env("DICEY_NO_LDS_TABLES") != nullptr;  // debugging aid (no barrier in the kernel)
      static const bool no_wave = exp_env("DICEY_NO_WAVE_THAL") != nullptr;  // debugging aid: sequential thal only
      // wave-per-hit path: LDS holds the tables once per workgroup and one DP table per wavefront
      const u32 tab_bytes = thal::wave_header_bytes();
      const u32 per_wave = thal::wave_mem_bytes(sx->max_primer_len, wmax);
      const u32 lds_cap = 160 * 1024;
      u32 wpb = std::min<u32>(16, (lds_cap - tab_bytes) / per_wave);
      const bool wave_path = !no_wave && !no_lds && sx->max_primer_len <= (u32)thal::kWaveMaxLen && wmax <= (u32)thal::kWaveMaxLen && wpb >= 2 && maxlen <= 63;
      // DP tables of the sequential kernel live in HBM, one launch per chunk of hits: 6 GB when it does all the work,
      // 1 GB when it only sees what the wave kernel hands back
      const u64 dp_budget = wave_path ? ((u64)1 << 30) : ((u64)6 << 30);
      const u64 chunk = std::max<u64>(4096, std::min<u64>(hit_cap, dp_budget / (2 * dp_stride * 8 + 1)));
      DG_TRY(ws[WS_HITS].reserve((hit_cap + 1) * sizeof(SiteRaw)));
      DG_TRY(ws[WS_ALN].reserve((hit_cap + 1) * (u64)wstride));
      DG_TRY(ws[WS_DP].reserve(((chunk + 63) & ~(u64)63) * 2 * dp_stride * 8 + 64));
      SiteArgs sa;
      sa.seeds = seeds;
      sa.nhits = hit_off + nq;
      sa.hit_cap = hit_cap;
      sa.cum = cum;
      sa.nseq = nseq;
      sa.tables = sx->th->d_tables;
      sa.env = sx->th->env;
      sa.pfw = sx->d_pfw;
      sa.prv = sx->d_prv;
      sa.poff = sx->d_poff;
      sa.koff = sx->d_koff;
      sa.cut_temp = sx->cut_temp;
      sa.out = ws[WS_HITS].as<SiteRaw>();
      sa.windows = ws[WS_ALN].as<u8>();
      sa.win_stride = wstride;
      sa.dp = ws[WS_DP].as<double>();
      sa.dp_stride = dp_stride;
      sa.dp_row = wmax;
      sa.only_flagged = 0;
      sa.wave_len1 = sa.wave_stride = sa.wave_bytes = 0;
      static const bool force_redo = exp_env("DICEY_DEBUG_THAL_REDO") != nullptr;
      sa.force_redo = force_redo ? 1 : 0;
      const u32 cells = (wmax + 1) * (maxlen + 1);
      if (wave_path) {
        const u32 lds_total = thal::wave_lds_total(wpb, per_wave);
        static int cus = 0;
        if (!cus) DG_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, ix->device));
        DG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_site_wave), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_total));
        sa.first = 0;
        sa.count = hit_cap;
        sa.wave_len1 = sx->max_primer_len;
        sa.wave_stride = wmax;
        sa.wave_bytes = per_wave;
        const u32 blocks_per_cu = std::max<u32>(1, lds_cap / lds_total);
        const u64 want_blocks = ceil_div(hit_cap, wpb);
        const dim3 wgrid((u32)std::min<u64>(want_blocks, (u64)cus * blocks_per_cu)), wblock(wpb * 64);
        hipLaunchKernelGGL(k_site_wave, wgrid, wblock, lds_total, st, ix->view, b, sa, ctr);
        sa.only_flagged = 1;  // whatever the wave kernel handed back goes through the sequential kernel below
      }
      for (u64 first = 0; first < hit_cap; first += chunk) {  // launches beyond the real hit count exit at once
        sa.first = first;
        sa.count = std::min<u64>(chunk, hit_cap - first);
        const dim3 sgrid(ceil_div(sa.count, 256)), sblock(256);
        if (cells <= 32 * 128) {
          if (no_lds) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_site<128, false>), sgrid, sblock, 0, st, ix->view, b, sa, ctr);
          else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_site<128, true>), sgrid, sblock, 0, st, ix->view, b, sa, ctr);
        } else {
          if (no_lds) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_site<2600, false>), sgrid, sblock, 0, st, ix->view, b, sa, ctr);
          else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_site<2600, true>), sgrid, sblock, 0, st, ix->view, b, sa, ctr);
        }
      }
      sx->d_sites = sa.out;
      sx->d_windows = sa.windows;
      sx->win_stride = wstride;
      sx->d_hit_off = hit_off;
      sx->d_qflags = b.qflags;
  return DG_OK;
}

}  // namespace dg

using namespace dg;

// ------------------------------------------------------------------------------------------------------------
// dg_search_sites
// ------------------------------------------------------------------------------------------------------------
#ifdef DG_WAVE_PROFILE
extern "C" void dg_debug_wave_profile(unsigned long long* out) {
  (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(dg::thal::g_wave_prof), 64);
  unsigned long long z[8] = {0};
  (void)hipMemcpyToSymbol(HIP_SYMBOL(dg::thal::g_wave_prof), z, 64);
}
#endif

extern "C" {

void dg_search_result_free(dg_search_result* r) {
  if (!r) return;
  delete[] r->sites;
  delete[] r->genome_pool;
  delete[] r->pflags;
  delete[] r->match_temp;
  delete r;
}

int dg_search_sites(dg_index* ix, dg_thal* th, const dg_search_params* p, const uint32_t* seqlen, uint32_t nseq,
                    const uint8_t* pbytes, const uint64_t* poff, size_t np, dg_search_result** out) {
  if (!ix || !th || !p || !seqlen || !pbytes || !poff || !out) return fail(DG_EINVAL, "dg_search_sites: null argument");
  *out = nullptr;
  if (!np) return fail(DG_EINVAL, "dg_search_sites: no primers");
  if (th->device != ix->device) return fail(DG_EINVAL, "dg_search_sites: index and thal handles live on different devices");
  if (p->kmer < 10 || p->kmer > MAX_QLEN) return fail(DG_ELIMIT, "k-mer size %u outside [10,%u]", p->kmer, MAX_QLEN);
  // k-mer queries (last k nucleotides), primer codes and their reverse complements
  const u64 ptotal = poff[np];
  std::vector<u8> kq(np * (u64)p->kmer), pfw(ptotal), prv(ptotal);
  std::vector<u64> koffs(np + 1);
  std::vector<u32> koff(np);
  u32 maxp = 0, maxk = 0;
  for (size_t i = 0; i < np; ++i) {
    const u64 b0 = poff[i], len = poff[i + 1] - b0;
    if (len < p->kmer) return fail(DG_EINVAL, "primer %zu is shorter than k", i);
    if (len > 64) return fail(DG_ELIMIT, "primer %zu has %llu nt; thal() accepts at most 60 and this build stores 64", i, (unsigned long long)len);
    maxp = std::max<u32>(maxp, (u32)len);
    koff[i] = (u32)(len - p->kmer);
    maxk = std::max(maxk, koff[i]);
    koffs[i] = (u64)i * p->kmer;
    for (u64 k = 0; k < len; ++k) {
      u8 ch = pbytes[b0 + k];
      if (ch >= 'a' && ch <= 'z') ch -= 32;
      u8 c = ch == 'A' ? 0 : ch == 'C' ? 1 : ch == 'G' ? 2 : ch == 'T' ? 3 : 4;
      pfw[b0 + k] = c;
      prv[b0 + (len - 1 - k)] = c < 4 ? (u8)(3 - c) : (u8)4;
    }
    std::memcpy(kq.data() + koffs[i], pbytes + b0 + koff[i], p->kmer);
  }
  koffs[np] = np * (u64)p->kmer;
  DG_HIP(hipSetDevice(ix->device));
  hipStream_t st = ix->stream;
  auto& ws = ix->ws;
  DG_TRY(ws[WS_QB].reserve(kq.size() + 8));
  DG_TRY(ws[WS_QOFF].reserve((np + 1) * 8));
  DG_TRY(ws[WS_PRIM].reserve(2 * ptotal + (np + 1) * 8 + np * 4 + 64));
  u8* d_pfw = ws[WS_PRIM].as<u8>();
  u8* d_prv = d_pfw + ptotal;
  u64* d_poff = (u64*)(d_pfw + ((2 * ptotal + 7) & ~7ULL));
  u32* d_koff = (u32*)(d_poff + np + 1);
  DG_HIP(hipMemcpyAsync(ws[WS_QB].p, kq.data(), kq.size(), hipMemcpyHostToDevice, st));
  DG_HIP(hipMemcpyAsync(ws[WS_QOFF].p, koffs.data(), (np + 1) * 8, hipMemcpyHostToDevice, st));
  DG_HIP(hipMemcpyAsync(d_pfw, pfw.data(), ptotal, hipMemcpyHostToDevice, st));
  DG_HIP(hipMemcpyAsync(d_prv, prv.data(), ptotal, hipMemcpyHostToDevice, st));
  DG_HIP(hipMemcpyAsync(d_poff, poff, (np + 1) * 8, hipMemcpyHostToDevice, st));
  DG_HIP(hipMemcpyAsync(d_koff, koff.data(), np * 4, hipMemcpyHostToDevice, st));
  dg_hunt_params hp{};
  hp.distance = p->distance;
  hp.hamming = p->hamming;
  hp.forward_only = 0;
  hp.max_locations = p->max_locations;
  hp.max_neighborhood = p->max_neighborhood;
  SearchExtra sx;
  sx.th = th;
  sx.d_pfw = d_pfw;
  sx.d_prv = d_prv;
  sx.d_poff = d_poff;
  sx.d_koff = d_koff;
  sx.cut_temp = p->cut_temp;
  sx.max_primer_len = maxp;
  sx.max_koff = maxk;
  dg_hunt_result* hr = nullptr;
  ix->sw = dg_switches::read();
  int rc = run_batch(ix, &hp, seqlen, nseq, ws[WS_QB].p, ws[WS_QOFF].p, np, kq.size(), p->kmer, 0, &hr, &sx, nullptr, kq.data(), koffs.data());
  if (rc != DG_OK) {
    if (hr) dg_hunt_result_free(hr);
    return rc;
  }
  const u64 nhits = hr->nhits;
  const double ms_dev = hr->ms_total;
  const dg_hunt_result hcopy = *hr;  // counters and phase timings (the pointers inside are not used)
  dg_hunt_result_free(hr);
  std::vector<u32> qfl(np);
  DG_HIP(hipMemcpyAsync(qfl.data(), sx.d_qflags, np * 4, hipMemcpyDeviceToHost, st));
  // stable compaction on the device (debugging aid DICEY_DEBUG_DUMP_RAW keeps every record)
  const char* dump = exp_env("DICEY_DEBUG_DUMP_RAW");
  u64 nkeep = 0;
  std::vector<SiteRaw> raw;
  std::vector<u8> win;
  if (nhits) {
    const u64 scan_words = nhits / 64 + nhits / 4096 + 64;
    DG_TRY(ws[WS_LEAF].reserve(nhits * 4 + (nhits + 1 + scan_words) * 8 + 64));
    DG_TRY(ws[WS_LEAFG].reserve(nhits * sizeof(SiteRaw)));
    DG_TRY(ws[WS_SEL].reserve(nhits * (u64)sx.win_stride));
    u64* pos = ws[WS_LEAF].as<u64>();
    u64* tmp = pos + nhits + 1;
    u32* keep = (u32*)(tmp + scan_words);
    const dim3 g1(ceil_div(nhits, 256)), b1(256);
    hipLaunchKernelGGL(k_site_keep, g1, b1, 0, st, (const SiteRaw*)sx.d_sites, nhits, dump ? -1e300 : p->cut_temp, keep);
    DG_TRY(device_scan(st, keep, nhits, pos, tmp));
    hipLaunchKernelGGL(k_site_compact, g1, b1, 0, st, (const SiteRaw*)sx.d_sites, (const u8*)sx.d_windows, sx.win_stride, (const u32*)keep,
                       (const u64*)pos, nhits, ws[WS_LEAFG].as<SiteRaw>(), ws[WS_SEL].as<u8>());
    DG_HIP(hipMemcpyAsync(&nkeep, pos + nhits, 8, hipMemcpyDeviceToHost, st));
    DG_HIP(hipStreamSynchronize(st));
    raw.resize(nkeep);
    win.resize(nkeep * (u64)sx.win_stride);
    if (nkeep) {
      DG_HIP(hipMemcpyAsync(raw.data(), ws[WS_LEAFG].p, nkeep * sizeof(SiteRaw), hipMemcpyDeviceToHost, st));
      DG_HIP(hipMemcpyAsync(win.data(), ws[WS_SEL].p, nkeep * (u64)sx.win_stride, hipMemcpyDeviceToHost, st));
    }
  }
  DG_HIP(hipStreamSynchronize(st));
  DG_HIP(hipGetLastError());
  if (dump) {  // per-hit records for tools/diff_raw.py
    if (FILE* fp = std::fopen(dump, "wb")) {
      std::fwrite(raw.data(), sizeof(SiteRaw), raw.size(), fp);
      std::fclose(fp);
    }
    const std::string wp = std::string(dump) + ".win";
    if (FILE* fp = std::fopen(wp.c_str(), "wb")) {
      std::fwrite(win.data(), sx.win_stride, raw.size(), fp);
      std::fclose(fp);
    }
  }
  // Tm of every primer against its perfect complement (silica.h:431-443)
  std::vector<u8> pairs;
  std::vector<u64> pairoff(2 * np + 1, 0);
  for (size_t i = 0; i < np; ++i) {
    const u64 b0 = poff[i], len = poff[i + 1] - b0;
    static const char asc[5] = {'A', 'C', 'G', 'T', 'N'};
    for (u64 k = 0; k < len; ++k) pairs.push_back((u8)asc[pfw[b0 + k]]);
    pairoff[2 * i + 1] = pairs.size();
    for (u64 k = 0; k < len; ++k) pairs.push_back((u8)asc[prv[b0 + k]]);
    pairoff[2 * i + 2] = pairs.size();
  }
  dg_search_result* R = new dg_search_result;
  std::memset(R, 0, sizeof *R);
  R->nprimers = np;
  R->nhits = nhits;
  R->ms_device = ms_dev;
  R->ms_fm_search = hcopy.ms_search;
  R->ms_site_stage = hcopy.ms_verify;
  R->ctr_ext_steps = hcopy.ctr_ext_steps;
  R->ctr_tab_reads = hcopy.ctr_tab_reads;
  R->ctr_filter_probes = hcopy.ctr_filter_probes;
  R->ctr_sa_reads = hcopy.ctr_sa_reads;
  R->pflags = new uint32_t[np];
  R->match_temp = new double[np];
  *out = R;
  rc = dg_thal_batch(th, pairs.data(), pairoff.data(), np, R->match_temp, nullptr, nullptr);
  if (rc != DG_OK) {
    dg_search_result_free(R);
    *out = nullptr;
    return rc;
  }
  // silica.h:533-566 on the host: de-duplicate on (refIndex, alignpos) per primer and strand, trim the site
  std::vector<dg_site> sites;
  std::string pool;
  for (size_t q = 0; q < np; ++q) {
    u32 fl = qfl[q] & (DG_Q_MAX_MATCHES | DG_Q_NBHD_EXCEEDED);
    if (R->match_temp[q] == -thal::kInf) fl |= DG_P_THAL_FAILED;
    R->pflags[q] = fl;
  }
  std::set<std::pair<u32, u32>> seen[2];  // silica.h:465 TUniquePrimerHits, one per strand, per primer
  size_t cur = (size_t)-1;
  for (u64 h = 0; h < raw.size(); ++h) {  // records arrive primer by primer, in hit order
    const SiteRaw& r = raw[h];
    const size_t q = r.qs >> 1;
    if (q != cur) {
      seen[0].clear();
      seen[1].clear();
      cur = q;
    }
    const u32 plen = (u32)(poff[q + 1] - poff[q]);
    const u32 fr = r.qs & 1;
    if (r.temp == -thal::kInf) {
      R->pflags[q] |= DG_P_THAL_FAILED;
      continue;
    }
    if (!(r.temp > p->cut_temp)) continue;
    std::pair<u32, u32> key(r.ref, r.alignpos);
    if (!seen[fr].insert(key).second) continue;
    const char* g = (const char*)win.data() + h * (u64)sx.win_stride;
    u32 glen = std::min<u32>(r.glen, sx.win_stride);
    u32 chrpos, goff = 0, gl = glen;
    u32 alignshift = r.alignpos - r.chrpos;
    if (fr) {
      chrpos = r.alignpos;
      goff = std::min(alignshift, glen);
      gl = std::min<u32>(plen, glen - goff);
    } else {
      chrpos = r.alignpos - koff[q];
      if (alignshift >= koff[q]) {
        alignshift -= koff[q];
        goff = std::min(alignshift, glen);
        gl = std::min<u32>(plen, glen - goff);
      }
    }
    dg_site sgl;
    std::memset(&sgl, 0, sizeof sgl);
    sgl.ref = r.ref;
    sgl.pos = chrpos;
    sgl.primer = (u32)q;
    sgl.on_for = fr ? 0 : 1;
    sgl.temp = r.temp;
    sgl.perf_temp = R->match_temp[q];
    sgl.genome_off = pool.size();
    sgl.genome_len = gl;
    pool.append(g + goff, gl);
    sites.push_back(sgl);
  }
  R->nsites = sites.size();
  R->sites = new dg_site[sites.size() ? sites.size() : 1];
  std::copy(sites.begin(), sites.end(), R->sites);
  R->genome_pool = new char[pool.size() + 1];
  std::memcpy(R->genome_pool, pool.data(), pool.size());
  R->genome_pool[pool.size()] = 0;
  return DG_OK;
}

}  // extern "C"
