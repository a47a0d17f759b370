// Select stage of the hunt pipeline (included by hunt.hip only): batch summary, grouping of leaves by (query, strand), duplicate /
// substring-minimal filter and std::set order (neighbors.h:29-45), max_locations gating (hunter.h:349-357), device prefix sums.
#pragma once
#include "hunt_search.hpp"

namespace dg {

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
  unsigned long long n_nwin;                // flag: N-bearing strands that walk in window mode (k_prepare)
  unsigned long long n_short2;              // distance 2: queries k_search2p<., true> left to the walker for their length (k_prepare)
  unsigned long long n_walk;                // groups listed for the walker (k_walk_list)
#ifdef DG_TOPK_PROFILE
  unsigned long long prof[24];
#endif
};
// Totals straight into the pinned host record (no atomics over the bus, no separate copy); the host reads it
// after the batch's single stream synchronisation.  The batch's last kernel: it also leaves the counters ZEROED for the next
// batch (which then needs no memset in front).  One workgroup.
// (r03 tried to run this in "the workgroup of the verify kernel that finishes last": the device-scope fence every workgroup needs
// before it reports in writes back its XCD's L2 — the 8 300 workgroups of a repeat-genome step went from 1.05 to 1.60 ms.  On
// this part workgroups of one launch do not talk to each other cheaply; a 7 us kernel of its own is the better deal.)
DG_DEV void batch_finish(Counters* ctr, const u64* nhits, Summary* host_out) {
  constexpr int NF = 12;  // fields 1, 7 and 8 are maxima, the others sums
  __shared__ unsigned long long acc[NF];
  if (threadIdx.x < NF) acc[threadIdx.x] = 0;
  __syncthreads();
  unsigned long long v[NF] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  if (threadIdx.x < JOB_SHARDS) {  // locate jobs queued by k_locate: wavefront jobs, workgroup jobs (both lists)
    v[10] = ctr->job_cnt[0][threadIdx.x];
    v[11] = (unsigned long long)ctr->job_cnt[1][threadIdx.x] + ctr->job_cnt[2][threadIdx.x];
  }
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
    host_out->jobs_big = acc[11];
    host_out->jobs_small = acc[10];
    host_out->worst_sel = acc[8];
    host_out->fused_leaves = acc[9];
    host_out->n_generic = ctr->pad_[6];
    host_out->n_nwin = ctr->pad_[7];
    host_out->n_short2 = ctr->pad_[9];
    host_out->n_walk = ctr->pad_[10] > ctr->pad_[11] ? ctr->pad_[10] : ctr->pad_[11];  // (the longer of the two lists: they share a capacity)
#ifdef DG_TOPK_PROFILE
    for (int i = 0; i < 24; ++i) host_out->prof[i] = ctr->prof[i];
#endif
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
// filt_out[slot of the packed leaf]: the filtered form of the leaf's interval (k_search2p: Leaf::ops[2] of a two-operation leaf), 0 = none
__global__ void k_group_pack(Batch b, const Leaf* in, u32 shard_cap, const Counters* ctr, const u64* grp_off, PLeaf* out, u32* filt_out) {
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
    filt_out[grp_off[lf.qs] + lf.slot] = (lf.nops == 2 && (lf.ops[2] >> 31)) ? lf.ops[2] : 0u;
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
  filt_out[grp_off[lf.qs] + lf.slot] = (lf.nops == 2 && (lf.ops[2] >> 31)) ? lf.ops[2] : 0u;
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
__global__ void k_leaf_rank(const PLeaf* G, const u32* filt, const u64* grp_off, u64 nq2, const u8* alive, Sel* sel, u32* nsel,
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
  s.len = sel_len_from(a.len, filt[t]);
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
__global__ void __launch_bounds__(128) k_group_select(const PLeaf* G, const u32* filt, const u64* grp_off, u32 indel, Sel* sel, u32* nsel, const Counters* ctr) {
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
      o.len = sel_len_from(a.len, filt[g0]);
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
      o.len = sel_len_from(a.len, filt[g0 + (ix[i] & 1023u)]);
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
// hunter.h:349-357: the kept strings of a query in push order (forward strand, then reverse) report occurrences while fewer than
// max_locations are in: take = min(occurrences, what is left), the first hit slot = what the strings before took.  Sixteen lanes per
// query (r06): the occurrence counts of sixteen strings are read together and their running total is a prefix sum over the lanes — one
// lane per query walked its strings one memory latency at a time, and a query of a repeat family keeps a few hundred of them (0.1 ms
// of a repeat-rich batch was this kernel waiting for its slowest lanes).
__global__ void __launch_bounds__(256) k_take(Batch b, const u64* grp_off, const u32* selbase, u64 flat_slots, const u32* nsel, Sel* sel, u32* qhits, const Counters* ctr) {
  const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  const u64 q = t >> 4;
  const u32 sub = (u32)(t & 15);
  if (q >= b.nq) return;  // (whole sixteen-lane groups leave together)
  if (ctr->overflow) {
    if (sub == 0) qhits[q] = 0;
    return;
  }
  const u64 M = b.max_locations;
  u64 seen = 0;  // occurrences of the strings so far (not capped)
  for (u32 strand = 0; strand < 2; ++strand) {
    const u32 ns = nsel[2 * q + strand];
    if (!ns) continue;  // (grp_off is not even computed when the generic kernels were left out)
    Sel* S = sel + sel_base_of(selbase, grp_off, flat_slots, 2 * q + strand);
    for (u32 r0 = 0; r0 < ns; r0 += 16) {
      const u32 r = r0 + sub;
      const u64 occ = r < ns ? sel_occ(S[r]) : 0;
      u64 incl = occ;
#pragma unroll
      for (int off = 1; off < 16; off <<= 1) {
        const u64 v = __shfl_up((unsigned long long)incl, off, 16);
        if ((int)sub >= off) incl += v;
      }
      const u64 before = seen + incl - occ, hb = before < M ? before : M;
      if (r < ns) {
        S[r].take = (u32)(occ < M - hb ? occ : M - hb);
        S[r].hbase = (u32)hb;
      }
      seen += __shfl((unsigned long long)incl, 15, 16);
    }
  }
  if (sub) return;
  const u64 hits = seen < M ? seen : M;
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
    for (u32 r = 0; r < ns; ++r) sum += sel_occ(S[r]);
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

}  // namespace dg
