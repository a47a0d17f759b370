// neighbors() WITH its size cap (reference src/neighbors.h:29-92) without its order dependence (r04), shared by the device kernel
// k_cap_enum (hunt.hip) and by a host harness (tests/host) that holds it against the literal restatement of the reference.
//
// The reference walks the trie of edit paths depth first, hands every leaf to _insert (neighbors.h:29-45) and gives up as soon as
// the working set holds `maxsize` strings (:50).  Two observations replace the walk by arithmetic over ALL leaves at once:
//  (1) After t leaves the working set is the set of substring-minimal strings among the first t leaves (and the sequence itself,
//      inserted first, :90): _insert drops a string that contains a member and replaces the members that contain it, which keeps
//      exactly the minimal elements, whatever the order.  So a distinct string x is a member from its BIRTH — the first leaf that
//      spells it — until its DEATH — the first leaf that spells a proper substring of it (never a member when that comes first).
//  (2) The rank of a leaf in the walk is a closed form of its edit path, because the size of a subtree only depends on the
//      characters left and the edits left: F(L, 0) = 1, F(L, 1) = 8 L + 1, F(L, 2) = 32 L^2 + 8 L + 1 for sequences over A,C,G,T
//      (children in the order of :52-78: deletion, no change, three substitutions in alphabet order, four insertions).
// Then |working set after leaf t| = #births <= t - #deaths <= t: +1 / -1 events on a rank-indexed array, a prefix sum, the first
// rank T where it reaches maxsize; the reference's answer is the set of strings alive at T (everything alive at the end when the
// cap stays silent).  Births are a hash-table minimum, deaths <= 14 probes per string (its proper substrings of >= m - d
// characters: language strings have m - d .. m + d characters).
//
// Strings are 2-bit packed with a sentinel bit above the first character: key = 1 << 2 len | codes, first character on top —
// up to 31 characters; sequences with an N, longer ones, Hamming mode and distances above 2 stay on the host (nbhd_host.hpp).
#pragma once
#include <cstdint>
#ifdef __HIPCC__
#include <hip/hip_runtime.h>
#define DG_CE __host__ __device__ __forceinline__
#else
#define DG_CE inline
#endif

namespace dg {
namespace cap {
using u32 = uint32_t;
using u64 = uint64_t;

static constexpr u32 MAX_KEY_LEN = 31;

DG_CE u64 leaves1(u64 L) { return 8 * L + 1; }               // leaves under a node with L characters and one edit left
DG_CE u64 leaves2(u64 L) { return 32 * L * L + 8 * L + 1; }  // ... two edits left
DG_CE u64 total_leaves(u32 m, u32 d) { return d == 0 ? 1 : d == 1 ? leaves1(m) : leaves2(m); }

DG_CE u32 key_len(u64 key) {  // position of the sentinel bit / 2
  u32 hb = 63;
  while (!((key >> hb) & 1ULL)) --hb;
  return hb >> 1;
}
DG_CE u64 make_key(u64 codes, u32 len) { return (1ULL << (2 * len)) | (codes & ((1ULL << (2 * len)) - 1)); }
// substring [a, a + L) of a string of `len` characters (first character on top)
DG_CE u64 sub_key(u64 codes, u32 len, u32 a, u32 L) { return make_key(codes >> (2 * (len - a - L)), L); }

// One edit at position p of a string (codes, len): op 0 = delete, 1..3 = substitute by the idx-th other base (alphabet order,
// neighbors.h:61-66), 4..7 = insert base op - 4 in front of position p (:70-75).  Returns the new string and the position the walk
// continues at.
DG_CE void apply(u64 codes, u32 len, u32 p, u32 op, u64& out, u32& olen, u32& next_pos) {
  const u32 R = len - p;  // characters from position p on
  const u64 low = codes & ((1ULL << (2 * R)) - 1), high = codes >> (2 * R);
  if (op == 0) {
    out = (high << (2 * (R - 1))) | (low & ((1ULL << (2 * (R - 1))) - 1));
    olen = len - 1;
    next_pos = p;
  } else if (op < 4) {
    const u32 orig = (u32)(codes >> (2 * (R - 1))) & 3u;
    const u32 idx = op - 1, c = idx + (idx >= orig);  // the idx-th base that is not the original
    out = codes ^ ((u64)(orig ^ c) << (2 * (R - 1)));
    olen = len;
    next_pos = p + 1;
  } else {
    out = (high << (2 * (R + 1))) | ((u64)(op - 4) << (2 * R)) | low;
    olen = len + 1;
    next_pos = p + 1;
  }
}

// Rank (number of leaves the walk meets before it) of a leaf inside a node with `L` characters left and ONE edit left: the edit
// at level j (j characters passed unchanged) with operation op, or no edit at all (j = L).
DG_CE u64 rank1(u32 L, u32 j, u32 op) {
  if (j >= L) return L;                                  // every level's deletion leaf lies before the unedited leaf
  if (op == 0) return j;                                 // deletions of the levels above
  return (u64)j + 1 + leaves1(L - j - 1) + (op - 1);     // this level's deletion, its "no change" subtree, then S, S, S, I, I, I, I
}
// the same for the first edit of a node with TWO edits left: rank of the first leaf of the child's subtree
DG_CE u64 base2(u32 L, u32 j, u32 op) {
  // levels above: their deletion subtrees (one edit left, L - i - 1 characters each)
  const u64 above = 8ULL * ((u64)j * (L - 1) - (u64)j * (j - 1) / 2) + j;
  if (j >= L) return above;  // the unedited leaf
  const u64 Lj = L - j;
  if (op == 0) return above;
  const u64 after_keep = above + leaves1(Lj - 1) + leaves2(Lj - 1);
  if (op < 4) return after_keep + (u64)(op - 1) * leaves1(Lj - 1);
  return after_keep + 3 * leaves1(Lj - 1) + (u64)(op - 4) * leaves1(Lj);
}

// Leaf number `item` of the first-edit child (j1, op1) of a sequence of m characters at distance d (1 or 2):
//   d = 1: item is ignored, the leaf is the edited string itself;
//   d = 2: item in [0, 8 L' + 1): item < 8 L' = second edit at level item / 8 with operation item % 8, item = 8 L' = no second edit.
// Returns false for items beyond the child's leaves.  rank counts from 1 (rank 0 is the sequence itself, neighbors.h:90).
DG_CE bool leaf_of(u64 q, u32 m, u32 d, u32 j1, u32 op1, u32 item, u64& key, u64& rank) {
  u64 s1;
  u32 l1, p1;
  apply(q, m, j1, op1, s1, l1, p1);
  if (d == 1) {
    key = make_key(s1, l1);
    rank = 1 + rank1(m, j1, op1);
    return item == 0;
  }
  const u32 L1 = l1 - p1;  // characters left after the first edit
  if (item > 8 * L1) return false;
  const u64 base = 1 + base2(m, j1, op1);
  if (item == 8 * L1) {
    key = make_key(s1, l1);
    rank = base + rank1(L1, L1, 0);
    return true;
  }
  const u32 j2 = item >> 3, op2 = item & 7u;
  u64 s2;
  u32 l2, p2;
  apply(s1, l1, p1 + j2, op2, s2, l2, p2);
  key = make_key(s2, l2);
  rank = base + rank1(L1, j2, op2);
  return true;
}
DG_CE u32 items_per_child(u32 m, u32 d) { return d == 1 ? 1u : 8u * (m + 1) + 1u; }  // upper bound of leaf_of's item range

DG_CE u64 hash_key(u64 key) { return key * 0x9E3779B97F4A7C15ULL; }

}  // namespace cap
}  // namespace dg
