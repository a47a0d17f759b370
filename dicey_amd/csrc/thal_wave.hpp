// thal() END1 melting temperature with ONE WAVEFRONT per oligo pair, DP table in LDS (device only).
//
// thal.hpp states the reference's table fill (src/thal.h:1503-1551) one cell after the other.  The recurrence only
// looks at cells (ii,jj) with ii < i AND jj < j, so the cells of one row are independent of each other: the 64 lanes
// first settle the sequence-only terms of a row (left/right ends, stack extension: one lane per column), then share
// out all (target column, opening row) pairs of that row and scan the pairing opening cells of "their" row with a
// bit mask.  The reference walks the openings in a fixed order (loop size d ascending, ii descending) and replaces
// the running value on strict improvement of the free energy, i.e. it keeps the FIRST minimum in visiting order;
// lanes therefore carry (G, visiting key) and the group reduces lexicographically — the same cell value, bit for bit.
//
// The one place where the walk is not a plain minimum is its entropy cutoff (thal.h:1322-1330: a candidate with
// S < -2500 that "wins" resets the value to (H=0, S=-3224)).  Such candidates only ever arise from openings that hold
// that same placeholder, their free energy is ~1e6 and a real structure's is below 3e5, so they can never displace a
// real value and re-writing the placeholder over itself changes nothing.  The kernel CHECKS this separation for every
// value it touches (g < 8e5 for real ones, g > 9e5 for cutoff ones, placeholder exactly (0,-3224)) and reports
// `ambiguous` otherwise; the caller then recomputes that pair with the sequential code of thal.hpp.
//
// Only cells whose two bases pair ever hold a value (every other cell is the constant (H=inf, S=-1) of initMatrix), and
// pairing is a property of the sequences alone: the table in LDS stores just those cells, row after row; cell (i,j) sits
// at rowstart[i] + (number of pairing columns of row i left of j), one popcount on the row's column mask.  That is about
// a quarter of len1*len2 for mixed sequences; a pair with more pairing cells than the table holds is handed back.
//
// The traceback (thal.h:2133-2179) needs no second search: every cell records which of the reference's three tests
// (left end, stacked pair, first matching opening) identifies its value, in the reference's order of testing.
#pragma once
#include <cstddef>

#include "thal.hpp"

namespace dg {
namespace thal {

constexpr int kWaveMaxLen = kPlaneMax;  // longest oligo the wave kernel takes (lane = position, 64-bit column masks)

struct RowInfo {   // per column of the row being filled
  double rS, rH;   // right end at (i,j)
  double G2;       // free energy of the value the cell holds before the openings are tried
  double eS, eH;   // left end at (i,j): the traceback's first test (thal.h:2143)
  double kS, kH;   // stacked continuation of (i-1,j-1): its second test (thal.h:2150)
  double pad;      // (r05: without it — 56 bytes — `search` ran 7 % slower: the record is read with 16-byte LDS loads)
};
struct WaveMem {  // LDS owned by one wavefront
  RowInfo* row;              // [stride]
  Cell* cells;               // [cap] pairing cells, row after row
  unsigned short* from;      // same indexing: kFromLeft / kFromStack / visiting key of the opening / kFromNone
  unsigned short* rowstart;  // [len1 + 2] first slot of each row
  unsigned char* tlist;      // [stride] columns of the current row that take openings
  unsigned char* bcol;       // [stride + 2] r05: bcol[jj] = b[jj] (jj = 0 .. len2 + 1, the frame's sentinels included)
  unsigned char* qcol;       // [stride + 2] r05: qcol[jj] = b[jj] * 5 + b[jj + 1]: the opening pair's column half of a four-index table entry
  unsigned cap;
};
constexpr unsigned short kFromNone = 0, kFromLeft = 1, kFromStack = 2;  // opening keys are >= 3*64

// slots for pairing cells: 7/16 of the full table (mixed sequences need about 1/4) plus one row
DG_HD constexpr unsigned wave_cell_cap(unsigned len1, unsigned stride) { return (len1 * stride * 7u) / 16u + stride + 16u; }
DG_HD constexpr unsigned wave_mem_bytes(unsigned len1, unsigned stride) {
  return (stride * (unsigned)sizeof(RowInfo) + wave_cell_cap(len1, stride) * ((unsigned)sizeof(Cell) + 2u) + (len1 + 2u) * 2u + stride + 2u * (stride + 2u) + 15u) & ~15u;
}
__device__ inline WaveMem wave_mem_at(unsigned char* base, unsigned len1, unsigned stride) {
  WaveMem m;
  m.cap = wave_cell_cap(len1, stride);
  m.row = reinterpret_cast<RowInfo*>(base);
  m.cells = reinterpret_cast<Cell*>(base + stride * sizeof(RowInfo));
  m.from = reinterpret_cast<unsigned short*>(m.cells + m.cap);
  m.rowstart = m.from + m.cap;
  m.tlist = reinterpret_cast<unsigned char*>(m.rowstart + len1 + 2);
  m.bcol = m.tlist + stride;
  m.qcol = m.bcol + stride + 2;
  return m;
}

// The parameter tables as one array of doubles: entry e of an ...S table sits at offS + e, of its ...H twin at offH + e.
// kOff* name a table PAIR by the position of its S member, kTwin* the distance to the H twin.
#define DG_TOFF(member) ((int)(offsetof(Tables, member) / sizeof(double)))
constexpr int kOffStack = DG_TOFF(stackS), kOffStackmm = DG_TOFF(stackmmS), kOffTstack = DG_TOFF(tstackS);
constexpr int kOffInterior = DG_TOFF(interiorS), kOffBulge = DG_TOFF(bulgeS), kOffAtp = DG_TOFF(atpS);
static_assert(DG_TOFF(stackH) - DG_TOFF(stackS) == 625 && DG_TOFF(stackmmH) - DG_TOFF(stackmmS) == 625 &&
                  DG_TOFF(tstackH) - DG_TOFF(tstackS) == 625,
              "four-index tables: H follows S");
static_assert(DG_TOFF(interiorH) - DG_TOFF(interiorS) == 30 && DG_TOFF(bulgeH) - DG_TOFF(bulgeS) == 30, "loop tables: H follows S");
static_assert(DG_TOFF(atpH) - DG_TOFF(atpS) == 25, "atp: H follows S");
constexpr int kTwin4 = 625, kTwinLoop = 30, kTwinAtp = 25;

// Left and right duplex ends (thal.h:853-1076) only look at the closing pair and its two outer neighbours: with x pairing
// y = 3-x that is 4 x 5 x 5 cases per end and per RC value (both oligos self-complementary or not).  A workgroup works
// them out once with the functions of thal.hpp — same arithmetic, same bits — and the row pass reads them back.
struct EndTables {
  Cell L[2][4][5][5];  // [symmetric][x][a[i-1]][b[j-1]]  -> (eS, eH) of left_end
  Cell R[2][4][5][5];  // [symmetric][x][a[i+1]][b[j+1]]  -> (rS, rH) of right_end
};
// r05: the end tables can live INSIDE the LDS copy of the parameter tables, over tstack2 — which, like the dangles, only the two end
// functions read; once the 400 cases are worked out nothing looks at it again (the END1 step below reads the end tables too).
// INSIDE = true saves 6 400 bytes of header: a seventh wavefront of 40-mers per CU in k_thal_self_wave (padlock probes: 148 -> 126 ms
// per 268 000).  The kernels that hold 16 wavefronts anyway keep the tables BEHIND the parameter tables: with them inside
// `search` (k_site_wave) ran 7 % slower, same code, same occupancy (1 267 against 1 360 ms per step, tools/r05_call10.sh with
// DG_E_VARIANT builds) — an LDS placement effect that was measured, not explained.
static_assert(sizeof(EndTables) <= 2 * sizeof(double[5][5][5][5]) && offsetof(Tables, tstack2H) == offsetof(Tables, tstack2S) + sizeof(double[5][5][5][5]),
              "the end tables fit where tstack2S / tstack2H sit");
template <bool INSIDE>
DG_HD constexpr unsigned wave_e_offset() {
  return INSIDE ? (unsigned)offsetof(Tables, tstack2S) : (((unsigned)sizeof(Tables) + 15u) & ~15u);
}
template <bool INSIDE = false>
DG_HD constexpr unsigned wave_header_bytes() {  // parameter tables and end tables at the start of a workgroup's LDS
  return (((unsigned)sizeof(Tables) + 15u) & ~15u) + (INSIDE ? 0u : (((unsigned)sizeof(EndTables) + 15u) & ~15u));
}
template <bool INSIDE = false>
__device__ inline const EndTables* wave_end_tables(const unsigned char* lds) {
  return reinterpret_cast<const EndTables*>(lds + wave_e_offset<INSIDE>());
}
// dynamic LDS of a workgroup of `waves` wavefronts (wave_header_init stages the end tables behind the header before it moves them in)
template <bool INSIDE = false>
inline unsigned wave_lds_total(unsigned waves, unsigned per_wave) {
  const unsigned body = waves * per_wave, stage = ((unsigned)sizeof(EndTables) + 15u) & ~15u;
  return wave_header_bytes<INSIDE>() + (body > stage ? body : stage);
}
// every thread of the workgroup calls this once; ends with a barrier
template <bool INSIDE = false>
__device__ inline void wave_header_init(unsigned char* lds, const Tables* global_tables, const Env& env) {
  {
    const uint64_t* src = reinterpret_cast<const uint64_t*>(global_tables);
    uint64_t* dst = reinterpret_cast<uint64_t*>(lds);
    for (unsigned k = threadIdx.x; k < sizeof(Tables) / 8; k += blockDim.x) dst[k] = src[k];
  }
  __syncthreads();
  const Tables& T = *reinterpret_cast<const Tables*>(lds);
  EndTables* E = reinterpret_cast<EndTables*>(lds + wave_e_offset<INSIDE>());
  // every case first into the LDS behind the header (the wavefronts' DP memory, not in use yet: wave_lds_total keeps it large enough),
  // then, behind a barrier, over the table the cases were read from.  (r05: the cases held in registers instead — the loop unrolled
  // seven times — doubled the kernel's code and `search` ran 7 % slower.)
  EndTables* tmp = reinterpret_cast<EndTables*>(lds + wave_header_bytes<INSIDE>());
  for (unsigned k = threadIdx.x; k < 400; k += blockDim.x) {
    const unsigned right = k / 200, sym = (k / 100) & 1, x = (k / 25) & 3, n1 = (k / 5) % 5, n2 = k % 5;
    uint8_t aa[2], bb[2];
    ProblemT<const uint8_t*> p;
    p.T = &T;
    p.len1 = p.len2 = 1;
    p.rc = sym ? env.rc_sym : env.rc_asym;
    p.C = nullptr;
    p.row = 1;
    p.cs = 1;
    double S = -1.0, H = kInf;
    if (!right) {
      aa[0] = (uint8_t)n1;
      aa[1] = (uint8_t)x;
      bb[0] = (uint8_t)n2;
      bb[1] = (uint8_t)(3 - x);
      p.a = aa;
      p.b = bb;
      left_end(p, 1, 1, S, H);
      tmp->L[sym][x][n1][n2].s = S;
      tmp->L[sym][x][n1][n2].h = H;
    } else {
      aa[0] = (uint8_t)x;
      aa[1] = (uint8_t)n1;
      bb[0] = (uint8_t)(3 - x);
      bb[1] = (uint8_t)n2;
      p.a = aa;
      p.b = bb;
      right_end(p, 0, 0, S, H);
      tmp->R[sym][x][n1][n2].s = S;
      tmp->R[sym][x][n1][n2].h = H;
    }
  }
  __syncthreads();
  {
    const uint64_t* src = reinterpret_cast<const uint64_t*>(tmp);
    uint64_t* dst = reinterpret_cast<uint64_t*>(E);
    for (unsigned k = threadIdx.x; k < sizeof(EndTables) / 8; k += blockDim.x) dst[k] = src[k];
  }
  __syncthreads();
}

#ifdef DG_WAVE_PROFILE
__device__ unsigned long long g_wave_prof[8];
#define DG_PROF_T(var) const unsigned long long var = __builtin_readcyclecounter()
#define DG_PROF_ADD(slot, t0, t1) prof_acc[slot] += (t1) - (t0)
#define DG_PROF_DECL unsigned long long prof_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define DG_PROF_FLUSH do { if ((threadIdx.x & 63) == 0) for (int k_ = 0; k_ < 8; ++k_) if (prof_acc[k_]) atomicAdd(&::dg::thal::g_wave_prof[k_], prof_acc[k_]); } while (0)
#else
#define DG_PROF_T(var)
#define DG_PROF_ADD(slot, t0, t1)
#define DG_PROF_DECL
#define DG_PROF_FLUSH
#endif

__device__ inline void wave_sync() {  // lanes of a wave run in lockstep; this only pins the order of LDS traffic
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ inline uint64_t wave_uniform(uint64_t x) {
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)x), hi = __builtin_amdgcn_readfirstlane((unsigned)(x >> 32));
  return ((uint64_t)hi << 32) | lo;
}
__device__ inline uint64_t low_bits(int n) { return n >= 64 ? ~0ULL : ((1ULL << n) - 1); }

// a, b: framed code sequences (wave-uniform), oligo2 reversed; 1 <= len1, len2 <= kWaveMaxLen; m from
// wave_mem_at(.., >= len1, stride >= len2).  Every lane returns the same Result.
template <class SeqT>
__device__ inline Result wave_end1_tm(const Tables& T, const EndTables& E, const Env& env, const SeqT& a, int len1, const SeqT& b, int len2,
                                      bool both_symmetric, const WaveMem& m, int stride, bool& ambiguous) {
  const int lane = (int)(threadIdx.x & 63);
  Result r;
  r.temp = -kInf;
  r.end1 = r.end2 = -1;
  r.ok = true;
  ProblemT<SeqT> p;
  p.T = &T;
  p.a = a;
  p.b = b;
  p.len1 = len1;
  p.len2 = len2;
  p.rc = both_symmetric ? env.rc_sym : env.rc_asym;
  p.C = nullptr;  // cells are addressed through slot() below
  p.row = stride;
  p.cs = 1;
  bool amb = false;
  DG_PROF_DECL;
  const double* TD = reinterpret_cast<const double*>(&T);
  // pm[c]: columns jj (bit jj-1) whose base pairs with code c
  const int myb = lane < len2 ? b[lane + 1] : 4;
  uint64_t pm[4];
  for (int c = 0; c < 4; ++c) pm[c] = __ballot(myb == 3 - c);
  const int j = lane + 1;  // the column this lane owns in the per-row pass
  // r05: the second oligo as bytes in LDS for the openings loop — its lanes ask for b[jj], b[jj + 1] at lane-varying jj, which on the
  // wave-uniform bit planes is a dozen 64-bit shifts per candidate
  {
    const int nb = lane + 2 <= len2 + 1 ? b[lane + 2] : 4;
    if (lane <= len2) {
      m.bcol[j] = (unsigned char)(lane < len2 ? myb : 4);           // bcol[len2 + 1] = the frame's sentinel
      m.qcol[j] = (unsigned char)((lane < len2 ? myb : 4) * 5 + (lane < len2 ? nb : 4));
    }
    if (lane == 0) {
      m.bcol[0] = 4;
      m.qcol[0] = (unsigned char)(4 * 5 + b[1]);
    }
  }
  wave_sync();
  const Cell kNoPair = {kInf, -1.0};  // what initMatrix leaves in a cell whose bases do not pair (thal.h:820-835)
  auto row_mask = [&](int code) -> uint64_t { return code == 0 ? pm[0] : code == 1 ? pm[1] : code == 2 ? pm[2] : code == 3 ? pm[3] : 0; };
  unsigned rowbase = 0, prevbase = 0;  // first slot of row i / of row i-1
  uint64_t prevmask = 0;
  for (int i = 1; i <= len1; ++i) {
    DG_PROF_T(tA0);
    const int ai = a[i];
    const uint64_t imask = row_mask(ai);
    if (rowbase + (unsigned)__popcll(imask) > m.cap) {  // more pairing cells than the table holds: not for this kernel
      ambiguous = true;
      r.temp = 0.0;
      return r;
    }
    if (lane == 0) m.rowstart[i] = (unsigned short)rowbase;
    const unsigned myslot = rowbase + (unsigned)__popcll(imask & low_bits(lane));  // slot of (i, j) when it pairs
    const bool mine = lane < len2 && pairs(ai, myb);
    bool takes = false;
    // ---- per-row pass: left end, stack extension, right end (thal.h:1507-1519) ----
    if (lane < len2) {
      double curS = -1.0, curH = kInf;
      unsigned short fr = kFromNone;
      if (mine) {
        curS = kMinEntropy;
        curH = 0.0;
        const int esym = both_symmetric ? 1 : 0;
        const Cell le = E.L[esym][ai][a[i - 1]][b[j - 1]];  // left_end(p, i, j)
        const double eS = le.s, eH = le.h;
        if (fin(eH)) {
          curS = eS;
          curH = eH;
        }
        double kS = 0, kH = 0;
        if (i > 1 && j > 1) {
          const Cell re = E.R[esym][ai][a[i + 1]][b[j + 1]];  // right_end(p, i, j)
          const double rS = re.s, rH = re.h;
          const bool dpair = (prevmask >> (j - 2)) & 1;
          const Cell diag = dpair ? m.cells[prevbase + (unsigned)__popcll(prevmask & low_bits(j - 2))] : kNoPair;
          double nS, nH;
          stack_pick(p, i, j, curS, curH, diag, rS, rH, nS, nH);
          curS = nS;
          curH = nH;
          kS = T.stackS[p.a[i - 1]][ai][p.b[j - 1]][myb] + diag.s;  // what the traceback compares (thal.h:2150)
          kH = T.stackH[p.a[i - 1]][ai][p.b[j - 1]][myb] + diag.h;
          RowInfo ri;
          ri.rS = rS;
          ri.rH = rH;
          ri.G2 = curH + rH - kT * (curS + rS);
          ri.eS = eS;
          ri.eH = eH;
          ri.kS = kS;
          ri.kH = kH;
          ri.pad = 0;
          m.row[lane] = ri;
          takes = true;
          const double g = curH - kT * curS;  // real value, or exactly the placeholder
          if (curS < kMinEntropyCutoff ? !(curS == kMinEntropy && curH == 0.0) : !(g < 800000.0)) amb = true;
        }
        if (curS == eS && curH == eH) fr = kFromLeft;
        else if (i > 1 && j > 1 && curS == kS && curH == kH) fr = kFromStack;
      }
      if (mine) {
        Cell c;
        c.h = curH;
        c.s = curS;
        m.cells[myslot] = c;
        m.from[myslot] = fr;
      }
    }
    const uint64_t tm = __ballot(takes);
    if (takes) m.tlist[__popcll(tm & low_bits(lane))] = (unsigned char)j;
    wave_sync();
    DG_PROF_T(tA1);
    DG_PROF_ADD(0, tA0, tA1);
    const int P = __popcll(tm);
    const unsigned thisbase = rowbase;
    prevbase = rowbase;
    prevmask = imask;
    rowbase += (unsigned)__popcll(imask);
    if (P == 0) continue;
    // ---- openings: groups of R lanes (one per opening row ii) serve one target column each ----
    const int n1 = i - 1 < kMaxLoop + 1 ? i - 1 : kMaxLoop + 1;  // l1 = i-ii-1 <= 30
    int lg = 0;
    while ((1 << lg) < n1) ++lg;
    const int R = 1 << lg, G = 64 >> lg;
    const int rr = lane & (R - 1), ii = i - 1 - rr;
    const int ca = rr < n1 ? a[ii] : 4;
    const int ao1 = rr < n1 ? a[ii + 1] : 4;  // the opening row's inner neighbour: the same for every target of this row
    const uint64_t rowmask = row_mask(ca);
    const unsigned obase = rr < n1 ? m.rowstart[ii] : 0;  // first slot of the opening row this lane serves
    const int l1 = rr;
    for (int t0 = 0; t0 < P; t0 += G) {
      DG_PROF_T(tB0);
      const int t = t0 + (lane >> lg);
      double bestG = 1e300, bestS = 0, bestH = 0, G2 = 0;
      unsigned bestKey = ~0u;
      int tj = 0;
      if (t < P && rr < n1) {
        tj = m.tlist[t];
        const RowInfo ri = m.row[tj - 1];
        G2 = ri.G2;
        // jj < tj, l2 = tj-jj-1 <= 30-l1, and not the stacked pair itself
        uint64_t mask = rowmask & low_bits(tj - 1);
        const int lo = tj - 2 - (kMaxLoop - l1);
        if (lo > 0) mask &= ~low_bits(lo);
        if (l1 == 0) mask &= ~(1ULL << (tj - 2));
        // thal.h:1200-1333 without branches: the four loop shapes differ in which tables feed the sum
        //   single bulge   bulge[ls] + stack[ao][ac][bo][bc]                    (sign test BEFORE the opening is added)
        //   longer bulge   bulge[ls] + atp[ao][bo] + atp[ac][bc]
        //   1x1 loop       stackmm[ao][ao'][bo][bo'] + stackmm[bc][bc'][ac][ac']
        //   interior loop  interior[ls] + tstack[ao][ao'][bo][bo'] + tstack[bc][bc'][ac][ac'] + ILA*|l1-l2|
        // (o = opening pair (ii,jj), c = closing pair (i,tj), ' = the neighbour inside the loop); operand order as there.
        const int ao = ca, ac = ai, ac1 = a[i - 1], bc = m.bcol[tj], bc1 = m.bcol[tj - 1];
        const int q2 = ((bc * 5 + bc1) * 5 + ac) * 5 + ac1, q1hi = (ao * 5 + ao1) * 25, sthi = (ao * 5 + ac) * 25, atpc = ac * 5 + bc;
        // r05: candidates are visited from the highest column down.  The table slot of the first one is a popcount; every further
        // candidate is the next lower pairing column of the opening row, i.e. the slot before (the columns the mask leaves out lie
        // above the first or below the last candidate).
        unsigned slot = 0;
        if (mask) slot = obase + (unsigned)__popcll(rowmask & low_bits(63 - __builtin_clzll(mask)));
        auto consider = [&](double S, double H, const Cell& open, int l2, bool single) {
          H += open.h;
          S += open.s;
          if (!fin(H)) {
            H = kInf;
            S = -1.0;
          }
          if (!single && H > 0 && S > 0) {
            H = kInf;
            S = -1.0;
          }
          // (-1, inf): its free energy is above the placeholder's, never taken
          const bool valid = fin(open.h) && fin(H);
          const bool cut = S < kMinEntropyCutoff;
          const double g = H - kT * S;
          if (valid && (cut ? !(g > 900000.0) : !(g < 800000.0))) amb = true;
          const double G1 = H + ri.rH - kT * (S + ri.rS);
          const unsigned key = (unsigned)((l1 + l2 + 2) * 64 + (i - ii));
          if (valid && !cut && (G1 < bestG || (G1 == bestG && key < bestKey))) {
            bestG = G1;
            bestKey = key;
            bestS = S;
            bestH = H;
          }
        };
        // (1) the two highest candidates with the body that knows every loop shape: a single bulge (l1 + l2 = 1) and the 1x1 loop
        //     can only be the opening row's columns tj - 1 and tj - 2, i.e. among these two
#pragma unroll 1
        for (int rep = 0; rep < 2 && mask; ++rep) {
          const int top = 63 - __builtin_clzll(mask);
          mask &= ~(1ULL << top);
          const int jj = top + 1;
          const Cell open = m.cells[slot];
          --slot;
          const int bo = m.bcol[jj], bo1 = m.bcol[jj + 1];
          const int l2 = tj - jj - 1, ls = l1 + l2 - 1;
          const bool bulge = (l1 == 0) != (l2 == 0);
          const bool single = bulge && l1 + l2 == 1, longb = bulge && !single, one = l1 == 1 && l2 == 1;
          const bool inter = !bulge && !one;
          const int q1 = q1hi + bo * 5 + bo1;
          const int e0 = one ? kOffStackmm + q1 : (inter ? kOffInterior : kOffBulge) + ls;
          const int e1 = single ? kOffStack + sthi + bo * 5 + bc : longb ? kOffAtp + ao * 5 + bo : one ? kOffStackmm + q2 : kOffTstack + q1;
          const int e2 = longb ? kOffAtp + atpc : kOffTstack + q2;
          const bool three = longb || inter;
          const int h0 = one ? kTwin4 : kTwinLoop, h12 = longb ? kTwinAtp : kTwin4;  // distance to the H twin
          double S = TD[e0] + TD[e1];
          double H = TD[e0 + h0] + TD[e1 + h12];
          const double S3 = S + TD[e2], H3 = H + TD[e2 + h12];
          if (three) {
            S = S3;
            H = H3;
          }
          const int asym = l1 > l2 ? l1 - l2 : l2 - l1;
          const double S4 = S + (kILAS * asym), H4 = H + (kILAH * asym);
          if (inter) {
            S = S4;
            H = H4;
          }
          if (single && (H > 0 || S > 0)) {
            H = kInf;
            S = -1.0;
          }
          consider(S, H, open, l2, single);
        }
        // (2) everything below them has l2 >= 2: interior loops for this lane's opening row when l1 >= 1, longer bulges when l1 = 0 — one
        //     shape per LANE, so the tables, the twin distance and the closing pair's term are settled outside the loop and a
        //     candidate costs its cell, one column byte, two table pairs and the sums (same operands, same order as above)
        if (mask) {
          const bool isb = l1 == 0;
          const int base0 = isb ? kOffBulge : kOffInterior;
          const int base1 = isb ? kOffAtp + ao * 5 : kOffTstack + q1hi;
          const int tw12 = isb ? kTwinAtp : kTwin4;
          const unsigned char* const col = isb ? m.bcol : m.qcol;
          const int e2 = isb ? kOffAtp + atpc : kOffTstack + q2;
          const double c2S = TD[e2], c2H = TD[e2 + tw12];
          const int lsbase = l1 - 1 + tj - 1;  // ls = l1 + l2 - 1 = lsbase - jj
          do {
            const int top = 63 - __builtin_clzll(mask);
            mask &= ~(1ULL << top);
            const int jj = top + 1;
            const Cell open = m.cells[slot];
            --slot;
            const int cix = col[jj];
            const int l2 = tj - jj - 1, ls = lsbase - jj;
            double S = TD[base0 + ls] + TD[base1 + cix];
            double H = TD[base0 + ls + kTwinLoop] + TD[base1 + cix + tw12];
            S = S + c2S;
            H = H + c2H;
            if (!isb) {
              const int asym = l1 > l2 ? l1 - l2 : l2 - l1;
              S = S + (kILAS * asym);
              H = H + (kILAH * asym);
            }
            consider(S, H, open, l2, false);
          } while (mask);
        }
      }
      DG_PROF_T(tB1);
      DG_PROF_ADD(1, tB0, tB1);
      DG_PROF_ADD(4, 0ULL, 1ULL);
      double grpG = bestG;
      unsigned grpKey = bestKey;
      for (int off = R >> 1; off > 0; off >>= 1) {
        const double oG = __shfl_xor(grpG, off);
        const unsigned oK = (unsigned)__shfl_xor((int)grpKey, off);
        if (oG < grpG || (oG == grpG && oK < grpKey)) {
          grpG = oG;
          grpKey = oK;
        }
      }
      if (bestKey != ~0u && bestKey == grpKey && grpG < G2) {  // the walk's final value for (i,tj)
        const unsigned at = thisbase + (unsigned)__popcll(imask & low_bits(tj - 1));
        const RowInfo ri = m.row[tj - 1];
        unsigned short fr = (unsigned short)bestKey;  // the traceback's tests, in its order, applied to the new value
        if (bestS == ri.eS && bestH == ri.eH) fr = kFromLeft;
        else if (bestS == ri.kS && bestH == ri.kH) fr = kFromStack;
        Cell c;
        c.h = bestH;
        c.s = bestS;
        m.cells[at] = c;
        m.from[at] = fr;
      }
      DG_PROF_T(tB2);
      DG_PROF_ADD(2, tB1, tB2);
    }
    wave_sync();
  }
  wave_sync();
  DG_PROF_T(tE0);
  ambiguous = __ballot(amb) != 0;
  // ---- END1: the first oligo's last base takes part (thal.h:2608-2626), first minimum over j ----
  double myG = kInf;
  if (lane < len2) {
    // right_end(p, len1, j) from the end tables when (len1, j) pairs; a cell that does not pair holds (inf, -1) and its free energy
    // is inf whatever stands here
    const int al = a[len1];
    const bool pr = pairs(al, myb);
    const Cell re = pr ? E.R[both_symmetric ? 1 : 0][al][a[len1 + 1]][b[j + 1]] : kNoPair;
    double rS = pr ? re.s : -1.0, rH = pr ? re.h : kInf;
    rS = rS + 0.000001;
    rH = rH + 0.000001;
    const Cell c = ((prevmask >> lane) & 1) ? m.cells[prevbase + (unsigned)__popcll(prevmask & low_bits(lane))] : kNoPair;  // row len1
    myG = (c.h + rH + kInitH) - kT * (c.s + rS + kInitS);
  }
  double bestG = myG;
  int bestJ = myG < kInf ? j : 0x7fffffff;
  if (!(myG < kInf)) bestG = kInf;
  for (int off = 32; off > 0; off >>= 1) {
    const double oG = __shfl_xor(bestG, off);
    const int oJ = __shfl_xor(bestJ, off);
    if (oG < bestG || (oG == bestG && oJ < bestJ)) {
      bestG = oG;
      bestJ = oJ;
    }
  }
  int bestI = len1;
  if (bestJ == 0x7fffffff) bestJ = 0;
  if (!fin(bestG)) bestI = bestJ = 1;
  double rS = -1.0, rH = kInf;  // right_end(p, bestI, bestJ): (bestI, bestJ) pairs whenever its value is used
  {
    const int x = a[bestI], y = b[bestJ];
    if (pairs(x, y)) {
      const Cell re = E.R[both_symmetric ? 1 : 0][x][a[bestI + 1]][b[bestJ + 1]];
      rS = re.s;
      rH = re.h;
    }
  }
  auto slot = [&](int ci, int cj, bool& there) -> unsigned {  // wave-uniform lookup of an arbitrary cell
    const uint64_t mk = row_mask(a[ci]);
    there = (mk >> (cj - 1)) & 1;
    return (unsigned)m.rowstart[ci] + (unsigned)__popcll(mk & low_bits(cj - 1));
  };
  bool top_there;
  const unsigned top_at = slot(bestI, bestJ, top_there);
  const Cell top = top_there ? m.cells[top_at] : kNoPair;
  const double dH = top.h + rH + kInitH;
  const double dS = top.s + rS + kInitS;
  if (!fin(top.h)) {
    r.temp = 0.0;
    return r;
  }
  // traceback: follow the recorded tests, counting base pairs
  int ti = bestI, tj = bestJ, npairs = 1;
  for (;;) {
    bool there;
    const unsigned at = slot(ti, tj, there);
    const unsigned fr = there ? m.from[at] : kFromNone;
    if (fr == kFromLeft || fr == kFromNone) break;
    if (fr == kFromStack) {
      --ti;
      --tj;
    } else {
      const int d = (int)(fr >> 6), di = (int)(fr & 63);
      const int l1 = di - 1, l2 = d - 2 - l1;
      ti -= di;
      tj = tj - 1 - l2;
    }
    ++npairs;
  }
  DG_PROF_T(tE1);
  DG_PROF_ADD(3, tE0, tE1);
  DG_PROF_ADD(5, 0ULL, 1ULL);
  DG_PROF_FLUSH;
  const int N = npairs - 1;
  r.temp = (dH / (dS + (N * env.salt_correction) + p.rc)) - kZeroC;
  r.end1 = bestI;
  r.end2 = bestJ;
  return r;
}

}  // namespace thal
}  // namespace dg
