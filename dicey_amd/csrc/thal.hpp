// Thermodynamic alignment of two oligos (nearest-neighbour model, dimer mode "END1"): the melting temperature `dicey
// search` asks primer3's thal() for — reference src/thal.h:2409-2655 (entry), :853-1076 (terminal stacks/dangles),
// :1123-1163 (stack extension), :1200-1333 (bulges / internal loops), :1503-1551 (table fill), :2133-2179 (traceback),
// :2204 (Tm); called at src/silica.h:437 and :511 with temponly = 1, type = thal_end1.
//
// Re-expressed for one-lane-per-pair execution on the GPU: no globals, no allocation, caller-provided DP planes, the
// traceback only counts base pairs (that is all the Tm formula needs).  Every floating-point expression keeps the
// reference's operand order so that results are bit-identical (compile with -ffp-contract=off; log/sqrt appear only
// in the two host-side constants salt_correction and RC).
#pragma once
#include <cstdint>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define DG_HD __host__ __device__ inline
#else
#define DG_HD inline
#endif

namespace dg {
namespace thal {

constexpr double kInf = 999999.0;        // thal.h:117 _INFINITY
constexpr double kT = 310.15;            // thal.h:128 TEMP_KELVIN
constexpr double kZeroC = 273.15;        // thal.h:127 ABSOLUTE_ZERO
constexpr double kMinEntropyCutoff = -2500.0;
constexpr double kMinEntropy = -3224.0;
constexpr double kInitH = 200.0, kInitS = -5.7;  // duplex initiation (thal.h:2502-2503)
constexpr double kILAS = (-300 / 310.15), kILAH = 0.0;
constexpr int kMaxLoop = 30;
constexpr int kMaxAlign = 60;  // THAL_MAX_ALIGN

struct Tables {  // primer3_config/*.ds|*.dh as thal.h:497-765 loads them
  double stackS[5][5][5][5], stackH[5][5][5][5];
  double stackmmS[5][5][5][5], stackmmH[5][5][5][5];
  double dangle3S[5][5][5], dangle3H[5][5][5], dangle5S[5][5][5], dangle5H[5][5][5];
  double tstackS[5][5][5][5], tstackH[5][5][5][5];
  double tstack2S[5][5][5][5], tstack2H[5][5][5][5];
  double interiorS[30], interiorH[30], bulgeS[30], bulgeH[30];
  double atpS[5][5], atpH[5][5];
};
struct Env {
  double salt_correction;  // saltCorrectS(mv, dv, dntp)
  double rc_sym, rc_asym;  // R*log(dna/1e9), R*log(dna/4e9)
};

DG_HD bool fin(double x) { return x < kInf / 2; }
DG_HD int pairs(int a, int b) { return a + b == 3 && a < 4 && b < 4; }  // A-T, C-G (BPI, thal.h:133-138)

// A framed code sequence (sentinel 4, codes, sentinel 4) held in registers: 3 bits per code, 21 codes per word.
// Indexing is shifts and selects — no memory access on the DP's critical path (up to 84 codes).
struct PackedSeq {
  uint64_t w[4];
  DG_HD PackedSeq() : w{0, 0, 0, 0} {}
  DG_HD void set(int i, unsigned c) {
    const int k = i / 21, sh = 3 * (i % 21);
    const uint64_t v = (uint64_t)c << sh;
    if (k == 0) w[0] |= v;
    else if (k == 1) w[1] |= v;
    else if (k == 2) w[2] |= v;
    else w[3] |= v;
  }
  DG_HD int operator[](int i) const {
    const int k = i / 21, sh = 3 * (i % 21);
    const uint64_t x = k == 0 ? w[0] : k == 1 ? w[1] : k == 2 ? w[2] : w[3];
    return (int)((x >> sh) & 7);
  }
};
constexpr int kPackedMax = 82;  // longest sequence a PackedSeq can frame

// The same as three bit planes, position i <-> bit i: what a wavefront gets from three ballots when lane i holds code i.
struct PlaneSeq {
  uint64_t b0, b1, b2;
  DG_HD int operator[](int i) const { return (int)(((b0 >> i) & 1) | (((b1 >> i) & 1) << 1) | (((b2 >> i) & 1) << 2)); }
};
constexpr int kPlaneMax = 62;  // positions 0..len+1 must fit 64 bits

struct alignas(16) Cell {  // enthalpy and entropy of the best structure ending with pair (i,j): one 16-byte access
  double h, s;
};

template <class SeqT>
struct ProblemT {
  const Tables* T;
  SeqT a;  // numSeq1[0..len1+1], sentinels 4 at both ends
  SeqT b;  // numSeq2 (second oligo REVERSED), same framing
  int len1, len2;
  double rc;
  Cell* C;  // DP table, (i,j) 1-based; cell (i,j) lives at ((j-1) + (i-1)*row) * cs
  int row;  // cells per row (>= len2; a common value lets the lanes of a wave touch the same cell index together)
  int cs;   // distance between consecutive cells (1, or 64 when tables are interleaved across a wavefront)
  DG_HD Cell& cell(int i, int j) const { return C[(size_t)((j - 1) + (i - 1) * row) * cs]; }
  DG_HD double& h(int i, int j) const { return cell(i, j).h; }
  DG_HD double& s(int i, int j) const { return cell(i, j).s; }
};
typedef ProblemT<const uint8_t*> Problem;

// The three dangling-end alternatives of an end compete with the terminal-mismatch stack in the same way.
template <class PROB>
DG_HD void end_compete(const PROB& p, double S2, double H2, double& S1, double& H1, double& T1, double G1) {
  double G2 = H2 - kT * S2;
  if (!fin(H2) || G2 > 0) {
    H2 = kInf;
    S2 = -1.0;
    G2 = 1.0;
  }
  double T2 = (H2 + kInitH) / (S2 + kInitS + p.rc);
  if (fin(H1) && G1 < 0) {
    T1 = (H1 + kInitH) / (S1 + kInitS + p.rc);
    if (T1 < T2 && G2 < 0) {
      S1 = S2;
      H1 = H2;
      T1 = T2;
    }
  } else if (G2 < 0) {
    S1 = S2;
    H1 = H2;
    T1 = T2;
  }
}
template <class PROB>
DG_HD void end_finish(const PROB& p, int x, int y, double S1, double H1, double T1, double& outS, double& outH) {
  const Tables& t = *p.T;
  double S2 = t.atpS[x][y], H2 = t.atpH[x][y];
  double T2 = (H2 + kInitH) / (S2 + kInitS + p.rc);
  if (fin(H1) && !(T1 < T2)) {
    outS = S1;
    outH = H1;
  } else {
    outS = S2;
    outH = H2;
  }
}

// Left end of the duplex at (i,j): thal.h:853-959.  Leaves (outS,outH) untouched when i/j cannot pair.
template <class PROB>
DG_HD void left_end(const PROB& p, int i, int j, double& outS, double& outH) {
  const Tables& t = *p.T;
  const int x = p.a[i], y = p.b[j], xm = p.a[i - 1], ym = p.b[j - 1];
  if (!pairs(x, y)) {
    p.s(i, j) = -1.0;
    p.h(i, j) = kInf;
    return;
  }
  double S1 = t.atpS[x][y] + t.tstack2S[y][ym][x][xm];
  double H1 = t.atpH[x][y] + t.tstack2H[y][ym][x][xm];
  double G1 = H1 - kT * S1, T1 = -kInf;
  if (!fin(H1) || G1 > 0) {
    H1 = kInf;
    S1 = -1.0;
    G1 = 1.0;
  }
  if (!pairs(xm, ym)) {
    const bool d3 = fin(t.dangle3H[y][ym][x]), d5 = fin(t.dangle5H[y][x][xm]);
    if (d3 && d5)
      end_compete(p, t.atpS[x][y] + t.dangle3S[y][ym][x] + t.dangle5S[y][x][xm], t.atpH[x][y] + t.dangle3H[y][ym][x] + t.dangle5H[y][x][xm],
                  S1, H1, T1, G1);
    else if (d3)
      end_compete(p, t.atpS[x][y] + t.dangle3S[y][ym][x], t.atpH[x][y] + t.dangle3H[y][ym][x], S1, H1, T1, G1);
    else if (d5)
      end_compete(p, t.atpS[x][y] + t.dangle5S[y][x][xm], t.atpH[x][y] + t.dangle5H[y][x][xm], S1, H1, T1, G1);
  }
  end_finish(p, x, y, S1, H1, T1, outS, outH);
}
// Right end at (i,j): thal.h:962-1076
template <class PROB>
DG_HD void right_end(const PROB& p, int i, int j, double& outS, double& outH) {
  const Tables& t = *p.T;
  const int x = p.a[i], y = p.b[j], xp = p.a[i + 1], yp = p.b[j + 1];
  if (!pairs(x, y)) {
    outS = -1.0;
    outH = kInf;
    return;
  }
  double S1 = t.atpS[x][y] + t.tstack2S[x][xp][y][yp];
  double H1 = t.atpH[x][y] + t.tstack2H[x][xp][y][yp];
  double G1 = H1 - kT * S1, T1 = -kInf;
  if (!fin(H1) || G1 > 0) {
    H1 = kInf;
    S1 = -1.0;
    G1 = 1.0;
  }
  if (!pairs(xp, yp)) {
    const bool d3 = fin(t.dangle3H[x][xp][y]), d5 = fin(t.dangle5H[x][y][yp]);
    if (d3 && d5)
      end_compete(p, t.atpS[x][y] + t.dangle3S[x][xp][y] + t.dangle5S[x][y][yp], t.atpH[x][y] + t.dangle3H[x][xp][y] + t.dangle5H[x][y][yp],
                  S1, H1, T1, G1);
    else if (d3)
      end_compete(p, t.atpS[x][y] + t.dangle3S[x][xp][y], t.atpH[x][y] + t.dangle3H[x][xp][y], S1, H1, T1, G1);
    else if (d5)
      end_compete(p, t.atpS[x][y] + t.dangle5S[x][y][yp], t.atpH[x][y] + t.dangle5H[x][y][yp], S1, H1, T1, G1);
  }
  end_finish(p, x, y, S1, H1, T1, outS, outH);
}

// Extend the duplex ending at (i-1,j-1) by the stacked pair (i,j) if that melts higher: thal.h:1123-1163, on values.
// (S0,H0) is what cell (i,j) holds, `diag` is cell (i-1,j-1), (rS,rH) the right end at (i,j).  When neither comparison
// holds (a NaN) the reference leaves the cell as it was, before the cutoff normalisation.
template <class PROB>
DG_HD void stack_pick(const PROB& p, int i, int j, double S0, double H0, const Cell& diag, double rS, double rH, double& outS,
                      double& outH) {
  const Tables& t = *p.T;
  outS = S0;
  outH = H0;
  const double T0 = (H0 + kInitH + rH) / (S0 + kInitS + rS + p.rc);
  const double stH = t.stackH[p.a[i - 1]][p.a[i]][p.b[j - 1]][p.b[j]];
  double S1, H1, T1;
  if (fin(diag.h) && fin(stH)) {
    S1 = diag.s + t.stackS[p.a[i - 1]][p.a[i]][p.b[j - 1]][p.b[j]];
    H1 = diag.h + stH;
    T1 = (H1 + kInitH + rH) / (S1 + kInitS + rS + p.rc);
  } else {
    S1 = -1.0;
    H1 = kInf;
    T1 = (H1 + kInitH) / (S1 + kInitS + p.rc);
  }
  if (S1 < kMinEntropyCutoff) {
    S1 = kMinEntropy;
    H1 = 0.0;
  }
  if (S0 < kMinEntropyCutoff) {
    S0 = kMinEntropy;
    H0 = 0.0;
  }
  if (T1 > T0) {
    outS = S1;
    outH = H1;
  } else if (T0 >= T1) {
    outS = S0;
    outH = H0;
  }
}

// Candidate (S,H) for closing pair (ii,jj) after an opening pair (i,j) with unpaired bases in between: the table part of
// thal.h:1200-1333 (everything up to the G1/G2 comparison).
template <class PROB>
DG_HD void loop_candidate(const PROB& p, const Cell& open, int i, int j, int ii, int jj, double& S, double& H) {
  const Tables& t = *p.T;
  const int l1 = ii - i - 1, l2 = jj - j - 1, ls = l1 + l2 - 1;
  S = -1.0;
  H = kInf;
  if ((l1 == 0 && l2 > 0) || (l2 == 0 && l1 > 0)) {
    if (l2 == 1 || l1 == 1) {  // single-base bulge: the flanking pairs still stack
      H = t.bulgeH[ls] + t.stackH[p.a[i]][p.a[ii]][p.b[j]][p.b[jj]];
      S = t.bulgeS[ls] + t.stackS[p.a[i]][p.a[ii]][p.b[j]][p.b[jj]];
      if (H > 0 || S > 0) {
        H = kInf;
        S = -1.0;
      }
      H += open.h;
      S += open.s;
      if (!fin(H)) {
        H = kInf;
        S = -1.0;
      }
    } else {
      H = t.bulgeH[ls] + t.atpH[p.a[i]][p.b[j]] + t.atpH[p.a[ii]][p.b[jj]];
      H += open.h;
      S = t.bulgeS[ls] + t.atpS[p.a[i]][p.b[j]] + t.atpS[p.a[ii]][p.b[jj]];
      S += open.s;
      if (!fin(H)) {
        H = kInf;
        S = -1.0;
      }
      if (H > 0 && S > 0) {
        H = kInf;
        S = -1.0;
      }
    }
  } else if (l1 == 1 && l2 == 1) {
    S = t.stackmmS[p.a[i]][p.a[i + 1]][p.b[j]][p.b[j + 1]] + t.stackmmS[p.b[jj]][p.b[jj - 1]][p.a[ii]][p.a[ii - 1]];
    S += open.s;
    H = t.stackmmH[p.a[i]][p.a[i + 1]][p.b[j]][p.b[j + 1]] + t.stackmmH[p.b[jj]][p.b[jj - 1]][p.a[ii]][p.a[ii - 1]];
    H += open.h;
    if (!fin(H)) {
      H = kInf;
      S = -1.0;
    }
    if (H > 0 && S > 0) {
      H = kInf;
      S = -1.0;
    }
  } else {
    const int asym = l1 > l2 ? l1 - l2 : l2 - l1;
    H = t.interiorH[ls] + t.tstackH[p.a[i]][p.a[i + 1]][p.b[j]][p.b[j + 1]] + t.tstackH[p.b[jj]][p.b[jj - 1]][p.a[ii]][p.a[ii - 1]] +
        (kILAH * asym);
    H += open.h;
    S = t.interiorS[ls] + t.tstackS[p.a[i]][p.a[i + 1]][p.b[j]][p.b[j + 1]] + t.tstackS[p.b[jj]][p.b[jj - 1]][p.a[ii]][p.a[ii - 1]] +
        (kILAS * asym);
    S += open.s;
    if (!fin(H)) {
      H = kInf;
      S = -1.0;
    }
    if (H > 0 && S > 0) {
      H = kInf;
      S = -1.0;
    }
  }
}

// the diagonal walk over opening pairs (ii,jj) whose loop to (i,j) has d-2 unpaired bases: thal.h:1521-1527
DG_HD void loop_start(int i, int j, int d, int& ii, int& jj) {
  ii = i - 1;
  jj = -ii - d + (j + i);
  if (jj < 1) {
    ii -= (jj - 1 < 0 ? 1 - jj : jj - 1);
    jj = 1;
  }
}

struct Result {
  double temp;
  int end1, end2;
  bool ok;
};

// a/b: framed code sequences (see ProblemT); oligo2 must already be reversed.  cells: len1*row Cells (stride cs).
template <class SeqT>
DG_HD Result end1_tm(const Tables& T, const Env& env, const SeqT& a, int len1, const SeqT& b, int len2, bool both_symmetric,
                     Cell* cells, int row = 0, int cs = 1) {
  Result r;
  r.temp = -kInf;  // THAL_ERROR_SCORE
  r.end1 = r.end2 = -1;
  r.ok = false;
  if (len1 <= 0 || len2 <= 0) {
    r.temp = 0.0;
    return r;
  }
  if (len1 > kMaxAlign && len2 > kMaxAlign) return r;
  ProblemT<SeqT> p;
  p.T = &T;
  p.a = a;
  p.b = b;
  p.len1 = len1;
  p.len2 = len2;
  p.rc = both_symmetric ? env.rc_sym : env.rc_asym;
  p.C = cells;
  p.row = row > 0 ? row : len2;
  p.cs = cs;
  r.ok = true;
  // initMatrix (thal.h:820-835) + fillMatrix (thal.h:1503-1551)
  for (int i = 1; i <= len1; ++i)
    for (int j = 1; j <= len2; ++j) {
      if (pairs(a[i], b[j])) {
        p.h(i, j) = 0.0;
        p.s(i, j) = kMinEntropy;
      } else {
        p.h(i, j) = kInf;
        p.s(i, j) = -1.0;
      }
    }
  for (int i = 1; i <= len1; ++i)
    for (int j = 1; j <= len2; ++j) {
      if (!fin(p.h(i, j))) continue;
      double eS = -1.0, eH = kInf;
      left_end(p, i, j, eS, eH);
      if (fin(eH)) {
        p.s(i, j) = eS;
        p.h(i, j) = eH;
      }
      if (i > 1 && j > 1) {
        // The reference evaluates RSH(i,j) and re-reads cell (i,j) for every candidate; both only depend on the
        // sequences / on the value we hold, so they are computed once and the cell stays in registers.
        double rS, rH;
        right_end(p, i, j, rS, rH);
        const Cell here = p.cell(i, j), diag = p.cell(i - 1, j - 1);
        double curS, curH;
        stack_pick(p, i, j, here.s, here.h, diag, rS, rH, curS, curH);
        bool changed = curS != here.s || curH != here.h;
        for (int d = 3; d <= kMaxLoop + 2; ++d) {
          int ii, jj;
          loop_start(i, j, d, ii, jj);
          for (; ii > 0 && jj < j; --ii, ++jj) {
            const Cell open = p.cell(ii, jj);
            if (!fin(open.h)) continue;
            double S, H;
            loop_candidate(p, open, ii, jj, i, j, S, H);
            const double G1 = H + rH - kT * (S + rS);
            const double G2 = curH + rH - kT * (curS + rS);
            double lS = -1.0, lH = kInf;
            if (G1 < G2) {
              lS = S;
              lH = H;
            }
            if (lS < kMinEntropyCutoff) {
              lS = kMinEntropy;
              lH = 0.0;
            }
            if (fin(lH)) {
              curH = lH;
              curS = lS;
              changed = true;
            }
          }
        }
        if (changed) {
          p.h(i, j) = curH;
          p.s(i, j) = curS;
        }
      }
    }
  // END1: the first oligo's last base must take part (thal.h:2608-2626)
  int bestI = len1, bestJ = 0;
  double bestG = kInf;
  for (int j = 1; j <= len2; ++j) {
    double rS, rH;
    right_end(p, len1, j, rS, rH);
    rS = rS + 0.000001;
    rH = rH + 0.000001;
    double G1 = (p.h(len1, j) + rH + kInitH) - kT * (p.s(len1, j) + rS + kInitS);
    if (G1 < bestG) {
      bestG = G1;
      bestJ = j;
    }
  }
  if (!fin(bestG)) bestI = bestJ = 1;
  double rS, rH;
  right_end(p, bestI, bestJ, rS, rH);
  const double dH = p.h(bestI, bestJ) + rH + kInitH;
  const double dS = p.s(bestI, bestJ) + rS + kInitS;
  if (!fin(p.h(bestI, bestJ))) {
    r.temp = 0.0;
    return r;
  }
  // traceback (thal.h:2133-2179): only the number of base pairs matters for Tm
  int i = bestI, j = bestJ, npairs = 1;
  for (;;) {
    double eS = -1.0, eH = kInf;
    left_end(p, i, j, eS, eH);
    if (p.s(i, j) == eS && p.h(i, j) == eH) break;
    bool done = false;
    if (i > 1 && j > 1) {
      const double stS = T.stackS[a[i - 1]][a[i]][b[j - 1]][b[j]], stH = T.stackH[a[i - 1]][a[i]][b[j - 1]][b[j]];
      if (p.s(i, j) == stS + p.s(i - 1, j - 1) && p.h(i, j) == stH + p.h(i - 1, j - 1)) {
        --i;
        --j;
        ++npairs;
        done = true;
      }
    }
    for (int d = 3; !done && d <= kMaxLoop + 2; ++d) {
      int ii, jj;
      loop_start(i, j, d, ii, jj);
      for (; !done && ii > 0 && jj < j; --ii, ++jj) {
        double lS, lH;
        const Cell open = p.cell(ii, jj);
        loop_candidate(p, open, ii, jj, i, j, lS, lH);
        if (p.s(i, j) == lS && p.h(i, j) == lH) {
          i = ii;
          j = jj;
          ++npairs;
          done = true;
          break;
        }
      }
    }
    if (!done) break;  // the reference would spin here; cannot happen on a consistent table
  }
  const int N = npairs - 1;  // (ps1 + ps2 entries)/2 - 1, thal.h:2197-2203
  r.temp = (dH / (dS + (N * env.salt_correction) + p.rc)) - kZeroC;
  r.end1 = bestI;
  r.end2 = bestJ;
  return r;
}

}  // namespace thal
}  // namespace dg
