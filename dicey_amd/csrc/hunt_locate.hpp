// Locate stage of the hunt pipeline (included by hunt.hip only): the `take` smallest suffix-array values of every kept interval
// (sdsl::locate + std::sort + the first max_locations entries, hunter.h:355-357).
#pragma once
#include "hunt_select.hpp"

namespace dg {

// ------------------------------------------------------------------------------------------------------------
// Locate: the `take` smallest SA values of [lo,hi), ascending (locate + std::sort + first min(occs,max) entries).
static constexpr u32 TOPK_KMAX = 1024;  // largest `take` k_locate_topk serves (its LDS holds 8 * 1 152 candidate minima)
static constexpr u32 TOPK_KCAP = 1152;      // blocks kept per level: k plus slack, so that one histogram pass usually decides
static constexpr u32 TOPK_KCAP_MID = 576;   // the small-buffer form: intervals of up to 8 * 576 entries, read whole
struct BigJob {  // a repeat-rich string: handled by one workgroup of k_locate_topk / k_locate_big
  u32 lo, occs, take, g;
  u32 len;   // bits 0-23 the string's length; bits 24-31: 0 = [lo, lo + occs) is its suffix-array interval, l + 1 = a run of the records
             // of prefix level l (FmView::plv[l].rec) that holds the interval's `take` smallest positions (k_locate found it)
  u32 slot;  // of the kept string (HitSeed::sel)
  u64 out;   // first hit slot
};
// One lane per kept string (r03; r02 walked the strings of a (query, strand) group in one lane — 170 hits per query on the
// repeat-bearing genome made that a chain of several hundred dependent reads).  The lane finds its group through the packed /
// grouped leaf of the same slot (slot_qs: address of that record's `qs` field, slot_stride: record size), serves strings of up to
// 24 occurrences itself and queues the others: up to 256 occurrences for one wavefront (k_locate_small), more for one
// workgroup (k_locate_topk / k_locate_big).
// r06: three lists (wavefront jobs, workgroup jobs that fit the small buffer whole, the rest), each in JOB_SHARDS regions with a
// producer counter of their own, and consumers that take jobs by their number (static stride) — r05 had ONE word per list for the
// producers and ONE `next job` word per consumer kernel: ~40 000 + 2 x 16 000 returning atomics on hot words per repeat-rich batch at
// ~11 ns each, i.e. most of the stage's 0.9 ms (r05 counters: 1 800 vector instructions per job and 30 us per job).
enum : u32 { JL_SMALL = 0, JL_MID = 1, JL_BIG = 2 };
struct LocJobs {
  BigJob* list[3];  // JOB_SHARDS regions of shard_cap jobs each
  u32 shard_cap;
  u32 (*cnt)[JOB_SHARDS];  // Counters::job_cnt (a region that overflows keeps counting: readers clamp to shard_cap)
  u32 mid_max;             // workgroup jobs of up to this many occurrences (and take <= TOPK_KMAX) go to JL_MID; 0 = no such list
};
// flat job number -> slot of the list: the regions' fill counts as an exclusive prefix in LDS (64 counters, once per workgroup)
struct JobIndex {
  u32 pre[JOB_SHARDS + 1];
};
DG_DEV void job_index_init(JobIndex& J, const u32* cnt, u32 shard_cap) {  // every lane of the workgroup (>= 64 lanes) calls it
  if (threadIdx.x < JOB_SHARDS) {
    const u32 c = cnt[threadIdx.x] < shard_cap ? cnt[threadIdx.x] : shard_cap;
    u32 incl = c;
    for (int off = 1; off < 64; off <<= 1) {
      const u32 v = __shfl_up(incl, off);
      if ((int)threadIdx.x >= off) incl += v;
    }
    J.pre[threadIdx.x + 1] = incl;
    if (threadIdx.x == 0) J.pre[0] = 0;
  }
  __syncthreads();
}
DG_DEV u32 job_slot(const JobIndex& J, u32 shard_cap, u32 j) {
  u32 lo = 0, hi = JOB_SHARDS;
  while (hi - lo > 1) {
    const u32 mid = (lo + hi) >> 1;
    if (J.pre[mid] <= j) lo = mid;
    else hi = mid;
  }
  return lo * shard_cap + (j - J.pre[lo]);
}
static constexpr u32 LOC_SMALL_MAX = 256;
// A string with up to N occurrences: N loads in flight, a bitonic network on registers (every index is a compile-time constant —
// r02's insertion sort indexed a private array dynamically, i.e. through scratch memory), `take` stores.
// mask / back: the filtered form of a kept string (Sel, r04) — only the suffixes whose bit is set count, and the string starts
// `back` characters in front of them; mask = all ones, back = 0 for an ordinary interval
template <int N>
DG_DEV void locate_in_registers(const u32* sa, u32 occs, u32 take, HitSeed* out, u32 g, u32 len, u32 slot, u32 mask, u32 back) {
  u32 v[N];
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] = ((u32)i < occs && ((mask >> i) & 1u)) ? sa[i] - back : 0xFFFFFFFFu;
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
                                                u64 flat_slots, u32 flat_cap, u32 generic_on, u32 jobs_on, u32 direct_ctx) {
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
      const u32 lo = S.lo, occs = S.hi - S.lo;  // suffixes to read (a filtered interval holds <= 16 and keeps those in its mask)
      const u32 fmask = sel_filtered(S) ? sel_mask(S) : 0xFFFFFFFFu, back = sel_filtered(S) ? sel_pre(S) : 0u, slen = sel_strlen(S);
      const u64 out0 = hit_off[g >> 1] + S.hbase;
      HitSeed* out = seeds + out0;
      if (direct_ctx && (S.len & SEL_CTX_VALID) && sel_filtered(S)) {
        // r06: ONE occurrence, and the search kernel knew the character in front of it (Sel): its record brings the position and the
        // character behind the string (the table window's suffix starts `back` characters into the string and ends with it, so that
        // character is the record's T[q + K]) — the verify stage then aligns the hit without a line of the text
        const u32 j = (u32)__ffs((int)fmask) - 1u;
        const uint2 rec = f.sax[(u64)lo + j];
        u32 lw = slen;
        if (!(rec.y & SAX_ESCAPE)) lw |= (((S.len >> SEL_CTX_SHIFT) & 3u) << 20) | (((rec.y >> (4u + 2u * (f.K - SAX_POST_OFF))) & 3u) << 24) | SEED_CTX_VALID | SEED_KEY_VALID;
        out[0] = HitSeed{rec.x - back, g, lw, (u32)t};
        reads += 2;
      } else if (occs <= 4) {
        locate_in_registers<4>(f.sa + lo, occs, take, out, g, slen, (u32)t, fmask, back);
        reads += occs;
      } else if (occs <= 16) {
        locate_in_registers<16>(f.sa + lo, occs, take, out, g, slen, (u32)t, fmask, back);
        reads += occs;
      } else {
        bj.lo = lo;
        bj.occs = occs;
        bj.take = take;
        bj.g = g;
        bj.len = slen;
        bj.slot = (u32)t;
        bj.out = out0;
        // (hunt -m above 16 384: served by this lane, below)
        queue = take > 16384 ? 4u : occs <= LOC_SMALL_MAX ? 1u : (occs <= jobs.mid_max && take <= TOPK_KMAX) ? 2u : 3u;
        // r06: a string with far more occurrences than it may report (a repeat family): the occurrences below text position X are one
        // run of level l's records; the coarsest-grained level whose run should hold about 2 take .. 8 take of them by the interval's
        // density is asked (two rank reads), then — while the run is too short — the next ones; a run of take .. 9 216 records
        // replaces the interval (and the walk down the block minima over it), anything else keeps it
        if (queue == 3u && take <= TOPK_KMAX && f.nplv && occs > 2 * TOPK_KCAP_MID * 8u) {
          u32 lv = 0;
          while (lv < f.nplv && (u64)occs * f.plv[lv].x < 2ULL * take * f.n) ++lv;
          for (; lv < f.nplv; ++lv) {
            const u64 r0 = plv_rank(f.plv[lv], lo), r1 = plv_rank(f.plv[lv], (u64)lo + occs);
            reads += 16;
            if (r1 - r0 > 8ULL * TOPK_KCAP) break;  // denser here than on average: the walk serves it
            if (r1 - r0 >= take) {
              bj.lo = (u32)r0;
              bj.occs = (u32)(r1 - r0);
              bj.len = slen | ((lv + 1u) << 24);
              queue = bj.occs <= jobs.mid_max ? 2u : 3u;
              break;
            }
          }
        }
      }
    }
  }
  const u32 lane = threadIdx.x & 63;
  // job slots: one atomic per wavefront and list, on the counter of this wavefront's region
  const u32 shard = (blockIdx.x * 4u + (threadIdx.x >> 6)) & (JOB_SHARDS - 1);
  for (u32 which = 1; which <= 3; ++which) {
    const unsigned long long mk = __ballot(queue == which);
    if (!mk) continue;
    const u32 leader = (u32)__ffsll((long long)mk) - 1u;
    // the job kernels were left out of this attempt: nobody will write these strings' hits, so the verify kernel must not run
    // (the host sees the job counts and repeats the batch with the job kernels)
    if (!jobs_on && lane == leader) atomicOr(&ctr->overflow, 4u);
    u32 base = 0;
    if (lane == leader) base = atomicAdd(&jobs.cnt[which - 1][shard], (u32)__popcll(mk));
    base = __shfl(base, (int)leader);
    if (queue == which) {
      const u32 j = base + (u32)__popcll(mk & ((1ULL << lane) - 1));
      if (j < jobs.shard_cap) jobs.list[which - 1][(u64)shard * jobs.shard_cap + j] = bj;
      else queue = 4u;  // a full region: nothing is dropped, the lane serves it
    }
  }
  if (queue == 4u) {
    // correct for any size, slow: selection by repeated minimum above the previous pick (positions are distinct).  Reached with
    // hunt -m above 16 384 and by the strings a full job region turned away.
    u64 prev = 0;
    bool first = true;
    for (u32 i = 0; i < bj.take; ++i) {
      u32 best = 0xFFFFFFFFu;
      const u32 lvq = bj.len >> 24;
      for (u32 j = 0; j < bj.occs; ++j) {
        const u32 x = lvq ? f.plv[lvq - 1].rec[bj.lo + j].x : f.sa[bj.lo + j];
        if ((first || x > prev) && x < best) best = x;
      }
      reads += bj.occs;
      seeds[bj.out + i] = HitSeed{best, bj.g, bj.len & 0xFFFFFFu, bj.slot};
      prev = best;
      first = false;
    }
  }
  wave_add(&ctr->sa_reads[blockIdx.x & (NSHARD - 1)], reads);
}

// Sort of n <= 4 * LANES 64-bit keys in LDS whose high 32 bits (text positions: distinct) decide the order, by ONE distribution pass
// instead of a comparison network (r06).  The bitonic network of 1 024 keys is 55 compare-exchange stages — ~1 300 vector
// instructions per lane for 32-bit keys, twice that for 64-bit ones, and with 13 000 workgroup jobs per repeat-rich batch the
// locate kernels were bound by instruction issue, not by memory (r05 counters: 78 M vector + 75 M scalar wave instructions per
// launch of k_locate_topk<576>, 60 % of the issue slots of its duration).  Positions of a repeat family's copies are spread about
// evenly over the range they span, so: 4 * LANES buckets of equal (power-of-two) width over [min, max], a count per bucket (LDS
// atomics), an exclusive scan, every key to its bucket's place, and the lane that owns four neighbouring buckets orders the few
// keys that landed there by insertion.  ~250 instructions per lane.  Returns false — keys untouched — when some lane's buckets hold
// more than BUCKET_SKEW keys (clustered positions: tandem arrays); the caller then runs the network.
// keys: LDS, n entries (in and out); cnt: 4 * LANES words of LDS; red: 16 words of LDS.  Every lane of the workgroup calls it.
static constexpr u32 BUCKET_SKEW = 32;
DG_DEV u32 wave_min32(u32 x) {
  for (int off = 32; off > 0; off >>= 1) {
    const u32 o = (u32)__shfl_xor((int)x, off);
    x = o < x ? o : x;
  }
  return x;
}
DG_DEV u32 wave_max32(u32 x) {
  for (int off = 32; off > 0; off >>= 1) {
    const u32 o = (u32)__shfl_xor((int)x, off);
    x = o > x ? o : x;
  }
  return x;
}
template <u32 LANES>
DG_DEV bool bucket_sort_keys(u64* keys, u32* cnt, u32* red, u32 n) {
  constexpr u32 NBK = 4 * LANES, NW = LANES / 64;
  constexpr u32 LOGB = LANES == 64 ? 8 : 10;
  static_assert(LANES == 64 || LANES == 256, "one wavefront or four");
  const u32 tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  u64 kv[4];
  u32 mn = 0xFFFFFFFFu, mx = 0;
#pragma unroll
  for (u32 r = 0; r < 4; ++r) {
    const u32 i = tid + r * LANES;
    kv[r] = i < n ? keys[i] : ~0ULL;
    if (i < n) {
      const u32 pos = (u32)(kv[r] >> 32);
      mn = pos < mn ? pos : mn;
      mx = pos > mx ? pos : mx;
    }
    cnt[i] = 0;
  }
  mn = wave_min32(mn);
  mx = wave_max32(mx);
  if (lane == 0) {
    red[wave] = mn;
    red[4 + wave] = mx;
  }
  __syncthreads();
#pragma unroll
  for (u32 w = 0; w < NW; ++w) {
    mn = red[w] < mn ? red[w] : mn;
    mx = red[4 + w] > mx ? red[4 + w] : mx;
  }
  const u32 span = mx - mn;  // (n >= 1: mx >= mn)
  const u32 sh = span < NBK ? 0u : (32u - (u32)__clz((int)span)) - LOGB;
  u32 bk[4];
#pragma unroll
  for (u32 r = 0; r < 4; ++r) {
    bk[r] = ((u32)(kv[r] >> 32) - mn) >> sh;
    if (tid + r * LANES < n) atomicAdd(&cnt[bk[r]], 1u);
  }
  __syncthreads();
  const uint4 c4 = *reinterpret_cast<const uint4*>(cnt + 4 * tid);
  const u32 local = c4.x + c4.y + c4.z + c4.w;
  const u32 incl = wave_incl_scan32(local);
  const u32 wmax = wave_max32(local);
  if (lane == 63) red[8 + wave] = incl;
  if (lane == 0) red[12 + wave] = wmax;
  __syncthreads();
  u32 before = 0, worst = 0;
#pragma unroll
  for (u32 w = 0; w < NW; ++w) {
    if (w < wave) before += red[8 + w];
    worst = red[12 + w] > worst ? red[12 + w] : worst;
  }
  if (worst > BUCKET_SKEW) return false;  // (the same answer in every lane; keys[] has not been written)
  const u32 e0 = before + incl - local;
  *reinterpret_cast<uint4*>(cnt + 4 * tid) = make_uint4(e0, e0 + c4.x, e0 + c4.x + c4.y, e0 + c4.x + c4.y + c4.z);
  __syncthreads();
#pragma unroll
  for (u32 r = 0; r < 4; ++r)
    if (tid + r * LANES < n) keys[atomicAdd(&cnt[bk[r]], 1u)] = kv[r];
  __syncthreads();
  // cnt[b] is now the END of bucket b: this lane's four buckets hold keys[e0 .. cnt[4 tid + 3])
  const u32 s1 = cnt[4 * tid + 3];
  for (u32 i = e0 + 1; i < s1; ++i) {
    const u64 key = keys[i];
    u32 j = i;
    while (j > e0 && keys[j - 1] > key) {
      keys[j] = keys[j - 1];
      --j;
    }
    keys[j] = key;
  }
  __syncthreads();
  return true;
}

// One WAVEFRONT per string of 17..256 occurrences (most of the queued strings on a repeat-bearing genome: 54 k of 72 k per
// 100 000 queries): the interval is read once, sorted in LDS by the wavefront alone (bitonic, no workgroup barrier to wait
// for), the first `take` values are written.  Jobs by number, static stride: they all cost about the same.
// r06: keys are (position << 32 | context word) — read from FmView::sax when the verify kernel that follows takes the hits' context
// from their seeds (with_ctx), else position << 32 | SAX_ESCAPE ("no context": the seed's length stays plain).
__global__ void __launch_bounds__(64) k_locate_small(FmView f, LocJobs jobs, HitSeed* seeds, Counters* ctr, u32 with_ctx) {
  __shared__ u64 buf[LOC_SMALL_MAX];
  __shared__ u32 cnt[LOC_SMALL_MAX], red[16];
  __shared__ JobIndex JI;
  job_index_init(JI, jobs.cnt[JL_SMALL], jobs.shard_cap);
  const u32 njobs = JI.pre[JOB_SHARDS];
  const bool sax_on = with_ctx && f.sax;
  u64 reads = 0;
  for (u32 jb = blockIdx.x; jb < njobs; jb += gridDim.x) {
    const unsigned long long p0 = DG_LPROF_NOW();
    const BigJob J = jobs.list[JL_SMALL][job_slot(JI, jobs.shard_cap, jb)];
    u32 n2 = 32;
    while (n2 < J.occs) n2 <<= 1;
    {  // (the up to four entries of a lane are read together)
      u64 key[4];
#pragma unroll
      for (u32 r = 0; r < 4; ++r) {
        const u32 i = threadIdx.x + r * 64;
        key[r] = ~0ULL;
        if (i < J.occs) {
          if (sax_on) {
            const uint2 rc = f.sax[(u64)J.lo + i];
            key[r] = ((u64)rc.x << 32) | rc.y;
          } else key[r] = ((u64)f.sa[(u64)J.lo + i] << 32) | SAX_ESCAPE;
        }
      }
#pragma unroll
      for (u32 r = 0; r < 4; ++r)
        if (threadIdx.x + r * 64 < n2) buf[threadIdx.x + r * 64] = key[r];
    }
    reads += J.occs;
    __syncthreads();
    const unsigned long long p1 = DG_LPROF_NOW();
    if (!bucket_sort_keys<64>(buf, cnt, red, J.occs))
    for (u32 kk = 2; kk <= n2; kk <<= 1)
      for (u32 jj = kk >> 1; jj > 0; jj >>= 1) {
        for (u32 i = threadIdx.x; i < n2; i += 64) {
          const u32 l = i ^ jj;
          if (l > i) {
            const u64 a = buf[i], b2 = buf[l];
            if ((a > b2) == ((i & kk) == 0)) {
              buf[i] = b2;
              buf[l] = a;
            }
          }
        }
        __syncthreads();
      }
    const unsigned long long p2 = DG_LPROF_NOW();
    for (u32 i = threadIdx.x; i < J.take; i += 64)
      seeds[J.out + i] = HitSeed{(u32)(buf[i] >> 32), J.g, seed_len_with_ctx(J.len, (u32)buf[i]), J.slot};
    __syncthreads();
    const unsigned long long p3 = DG_LPROF_NOW();
    DG_LPROF_ADD(16, 1);
    DG_LPROF_ADD(17, p1 - p0);
    DG_LPROF_ADD(18, p2 - p1);
    DG_LPROF_ADD(19, p3 - p2);
    DG_LPROF_ADD(20, p3 - p0);
  }
  if (threadIdx.x == 0 && reads) atomicAdd(&ctr->sa_reads[blockIdx.x & (NSHARD - 1)], (unsigned long long)reads);
}

// One workgroup per repeat-rich string: the `take` smallest suffix-array values of its interval, ascending.
// Radix select, one byte per pass from the top: a 256-bin histogram (LDS atomics) of the values that still match the
// prefix found so far tells which bin holds the take-th smallest value; the passes stop as soon as everything up to the end
// of that bin fits the LDS buffer (on a genome-wide repeat family that is after the first pass: positions spread over the
// whole text, so one top-byte bin holds occs/185 values).  One more pass collects those values, a bitonic sort orders
// them.  2-3 coalesced passes over the interval instead of 33 (r02: 33 ms -> see DESIGN.md on the repeat-rich genome).
__global__ void __launch_bounds__(256) k_locate_big(FmView f, LocJobs jobs, HitSeed* seeds, Counters* ctr, u32 topk_kmax) {
  constexpr u32 CAP = 16384;
  __shared__ u32 buf[CAP];
  __shared__ u32 hist[256];
  __shared__ u32 fill, s_prefix, s_mask, s_k, s_below, s_done;
  __shared__ JobIndex JI;
  job_index_init(JI, jobs.cnt[JL_BIG], jobs.shard_cap);
  const u32 njobs = JI.pre[JOB_SHARDS];
  for (u32 jb = blockIdx.x; jb < njobs; jb += gridDim.x) {
    const BigJob J = jobs.list[JL_BIG][job_slot(JI, jobs.shard_cap, jb)];
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
    for (u32 i = threadIdx.x; i < J.take; i += blockDim.x) seeds[J.out + i] = HitSeed{buf[i], J.g, J.len & 0xFFFFFFu, J.slot};
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
static constexpr u32 TOPK_PAD = 0xFFFFFFFFu;
template <u32 KC>
struct TopkLdsT {
  static constexpr u32 CW = 2 * KC > 2 * TOPK_KMAX ? 2 * KC : 2 * TOPK_KMAX;  // words: both index lists, or the final TOPK_KMAX 64-bit keys
  u32 val[8 * KC + 16];
  alignas(8) u32 cidx[CW];  // [cur * KC + i]; at the entries: the survivors' (position << 32 | context word) keys
  u32 eidx[16];
  u32 hist[256];
  u32 wsum[4];
  u32 sh[4];
  u32 n_kept;
  JobIndex ji;
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
// KC = TOPK_KCAP: any interval (walks the hierarchy), list JL_BIG.  KC = TOPK_KCAP_MID (r03): intervals that fit the smaller buffer
// whole (level 0 only, no expansion), list JL_MID — 26 instead of 46 KB of LDS; on the repeats genome two thirds of the 17 000
// workgroup jobs of a step are of that kind.
// r06: (i) jobs by number with a static stride (no `next job` word: see LocJobs); (ii) with_ctx: the entries are read from
// FmView::sax, and the survivors leave with their context word — at the entries a survivor's slot in `val` is overwritten by its
// index inside the interval, and once the index lists are dead the record {position, context} is read again (an L2 hit: this
// workgroup fetched the line a moment ago) into a 64-bit key, so the sort carries the context along.
template <u32 KC>
__global__ void __launch_bounds__(256) k_locate_topk(FmView f, LocJobs jobs, u32 which, HitSeed* seeds, Counters* ctr, u32 with_ctx) {
  constexpr u32 VMAXT = 8 * KC + 16;
  __shared__ TopkLdsT<KC> S;
  job_index_init(S.ji, jobs.cnt[which], jobs.shard_cap);
  const u32 njobs = S.ji.pre[JOB_SHARDS];
  const u32 lane = threadIdx.x & 63;
  const bool sax_on = with_ctx && f.sax;
  u64 reads = 0;
  for (u32 jb = blockIdx.x; jb < njobs; jb += gridDim.x) {
    __syncthreads();  // the previous job's buffers are free
    const u32 pb = which == JL_MID ? 0u : 8u;
    const unsigned long long p0 = DG_LPROF_NOW();
    unsigned long long p_thr = 0, p_lev = 0;
    const BigJob J = jobs.list[which][job_slot(S.ji, jobs.shard_cap, jb)];
    if (J.take > TOPK_KMAX) continue;  // k_locate_big's (hunt -m above 1 024)
    const u32 k = J.take;
    const u32 lvj = J.len >> 24, jlen = J.len & 0xFFFFFFu;  // a run of a prefix level's records, or the suffix-array interval itself
    const uint2* const recs = lvj ? f.plv[lvj - 1].rec : f.sax;
    const bool use_recs = sax_on || lvj;
    const u64 lo = J.lo, hi = (u64)J.lo + J.occs;
    // full blocks of level j inside [lo, hi): [A(j), B(j))
    auto A = [&](int j) -> u64 { return (lo + ((1ULL << (3 * j)) - 1)) >> (3 * j); };
    auto B = [&](int j) -> u64 { return hi >> (3 * j); };
    auto N = [&](int j) -> u64 { return B(j) > A(j) ? B(j) - A(j) : 0ULL; };
    int L = 0;
    while (!lvj && L + 1 < (int)f.nlev && N(L) > VMAXT - 16) ++L;  // a level left with more than 9 216 blocks has > 1 000 in the next
    if (N(L) > VMAXT - 16) continue;  // cannot happen: the top level of FmView::samin holds <= 64 blocks (JL_MID: occs <= 8 KC)
    u32 nv = (u32)N(L);
    // (eight loads in flight per lane: one load per trip of a plain loop is one memory latency per trip — the phase clocks of a
    //  development build showed half of a job's 60 us in this loop and in the survivors' record loads below)
    if (L == 0 && use_recs) {
      const uint2* src = recs + A(0);
      for (u32 base = 0; base < nv; base += 8 * 256) {
        u32 r[8];
#pragma unroll
        for (u32 u = 0; u < 8; ++u) {
          const u32 i = base + u * 256 + threadIdx.x;
          r[u] = i < nv ? src[i].x : TOPK_PAD;
        }
#pragma unroll
        for (u32 u = 0; u < 8; ++u) {
          const u32 i = base + u * 256 + threadIdx.x;
          if (i < nv) S.val[i] = r[u];
        }
      }
    } else {
      const u32* src = f.samin[L] + A(L);
      for (u32 base = 0; base < nv; base += 8 * 256) {
        u32 r[8];
#pragma unroll
        for (u32 u = 0; u < 8; ++u) {
          const u32 i = base + u * 256 + threadIdx.x;
          r[u] = i < nv ? src[i] : TOPK_PAD;
        }
#pragma unroll
        for (u32 u = 0; u < 8; ++u) {
          const u32 i = base + u * 256 + threadIdx.x;
          if (i < nv) S.val[i] = r[u];
        }
      }
    }
    reads += nv;
    u32 nc_prev = 0;
    u32 cur = 0;
    bool top = true;
    __syncthreads();
    const unsigned long long p1 = DG_LPROF_NOW();
    for (int j = L;; --j) {
      // at the entries: up to 96 values more than asked for may survive (the sort drops them) — an exact k-th value costs the
      // radix select all four byte passes, a little slack usually ends it after two
      const u32 limit = j == 0 ? (k + 96 < TOPK_KMAX ? k + 96 : TOPK_KMAX) : KC;
      const unsigned long long pt0 = DG_LPROF_NOW();
      const u32 T = nv > limit ? topk_threshold(S, nv, k, limit) : 0xFFFFFFFEu;
      if (threadIdx.x == 0) S.n_kept = 0;
      __syncthreads();
      const unsigned long long pt1 = DG_LPROF_NOW();
      p_thr += pt1 - pt0;
      if (j == 0) {  // the survivors are the answer: collect, sort, write
        // the survivors, listed: their index inside the interval (records) or their position (plain suffix array).  The list lies in
        // the half of the index lists that is free at this level; the other half is still read here (a survivor's index)
        u32* const slist = S.cidx + (top ? TOPK_KMAX : (cur ^ 1u) * KC);
        for (u32 base = 0; base < nv; base += 256) {
          const u32 p = base + threadIdx.x;
          const u32 x = p < nv ? S.val[p] : TOPK_PAD;
          const bool keep = x <= T && x != TOPK_PAD;
          u32 e = x;
          if (keep && use_recs) e = (u32)((top ? A(0) + p : (p < 8 * nc_prev ? (u64)S.cidx[cur * KC + (p >> 3)] * 8u + (p & 7u) : (u64)S.eidx[p - 8 * nc_prev])) - lo);
          const unsigned long long mk = __ballot(keep);
          u32 at = 0;
          if (lane == 0 && mk) at = atomicAdd(&S.n_kept, (u32)__popcll(mk));
          at = __shfl(at, 0);
          if (keep) slist[at + (u32)__popcll(mk & ((1ULL << lane) - 1))] = e;
        }
        __syncthreads();
        const u32 have = S.n_kept;  // k .. k + 96 values (fewer when the whole interval is shorter); the k smallest are written
        // every lane takes up to four listed survivors, reads their records together (L2 hits: this workgroup fetched those lines a
        // moment ago) and lays the 64-bit keys over the index lists, which are dead from here on
        u64* keys = reinterpret_cast<u64*>(S.cidx);
        {
          u32 mine[4];
#pragma unroll
          for (u32 r = 0; r < 4; ++r) mine[r] = threadIdx.x + r * 256 < have ? slist[threadIdx.x + r * 256] : TOPK_PAD;
          __syncthreads();
          u64 kv[4];
          if (use_recs) {
            uint2 rec[4];
#pragma unroll
            for (u32 r = 0; r < 4; ++r) rec[r] = threadIdx.x + r * 256 < have ? recs[lo + mine[r]] : make_uint2(0, 0);
#pragma unroll
            for (u32 r = 0; r < 4; ++r) kv[r] = ((u64)rec[r].x << 32) | (sax_on ? rec[r].y : SAX_ESCAPE);
          } else {
#pragma unroll
            for (u32 r = 0; r < 4; ++r) kv[r] = ((u64)mine[r] << 32) | SAX_ESCAPE;
          }
#pragma unroll
          for (u32 r = 0; r < 4; ++r)
            if (threadIdx.x + r * 256 < have) keys[threadIdx.x + r * 256] = kv[r];
          __syncthreads();
        }
        const unsigned long long p3 = DG_LPROF_NOW();
        u64 sv[4];
        if (bucket_sort_keys<256>(keys, S.val, S.hist, have)) {  // (the candidates in val are dead: its first 1 024 words count)
#pragma unroll
          for (int r = 0; r < 4; ++r) sv[r] = threadIdx.x * 4 + r < have ? keys[threadIdx.x * 4 + r] : ~0ULL;
        } else {  // clustered positions: the comparison network
          u32 n2 = 4;
          while (n2 < have) n2 <<= 1;
#pragma unroll
          for (int r = 0; r < 4; ++r) sv[r] = threadIdx.x * 4 + r < have ? keys[threadIdx.x * 4 + r] : ~0ULL;
          block_sort4<u64>(keys, n2, sv);
        }
        const unsigned long long p4 = DG_LPROF_NOW();
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (threadIdx.x * 4 + r < k)
            seeds[J.out + threadIdx.x * 4 + r] = HitSeed{(u32)(sv[r] >> 32), J.g, seed_len_with_ctx(jlen, (u32)sv[r]), J.slot};
        const unsigned long long p5 = DG_LPROF_NOW();
        DG_LPROF_ADD(pb + 0, 1);
        DG_LPROF_ADD(pb + 1, p1 - p0);
        DG_LPROF_ADD(pb + 2, p_thr);
        DG_LPROF_ADD(pb + 3, p3 - pt1);
        DG_LPROF_ADD(pb + 4, p4 - p3);
        DG_LPROF_ADD(pb + 5, p5 - p4);
        DG_LPROF_ADD(pb + 6, p_lev);
        DG_LPROF_ADD(pb + 7, p5 - p0);
        break;
      }
      if constexpr (KC != TOPK_KCAP) break;  // (the small-buffer form never leaves the entries: no code, no registers for the walk)
      else {
      // blocks of level j under the threshold -> cidx[cur ^ 1]
      for (u32 base = 0; base < nv; base += 256) {
        const u32 p = base + threadIdx.x;
        const u32 x = p < nv ? S.val[p] : TOPK_PAD;
        const bool keep = x <= T && x != TOPK_PAD;
        u32 idx = 0;
        if (keep) idx = top ? (u32)(A(L) + p) : (p < 8 * nc_prev ? S.cidx[cur * KC + (p >> 3)] * 8u + (p & 7u) : S.eidx[p - 8 * nc_prev]);
        const unsigned long long mk = __ballot(keep);
        u32 at = 0;
        if (lane == 0 && mk) at = atomicAdd(&S.n_kept, (u32)__popcll(mk));
        at = __shfl(at, 0);
        if (keep) S.cidx[(cur ^ 1) * KC + at + (u32)__popcll(mk & ((1ULL << lane) - 1))] = idx;
      }
      __syncthreads();
      const u32 nc = S.n_kept;
      cur ^= 1;
      top = false;
      // their children, and the blocks of level j-1 that stick out at either end of the interval
      const u32* lv = f.samin[j - 1];
      const bool from_recs = j - 1 == 0 && sax_on;  // the entries themselves: from the records, whose lines the survivors read again
      const u64 nlow = j - 1 == 0 ? f.n : ~0ULL;  // level 0 is the suffix array itself: nothing beyond n
      // (the blocks of two trips are read before either is stored: the loads of a lane's blocks are in flight together)
      for (u32 base = 0; base < nc; base += 2 * 256) {
        u32 v[2][8];
        u64 c8[2];
#pragma unroll
        for (u32 u = 0; u < 2; ++u) {
          const u32 i = base + u * 256 + threadIdx.x;
          c8[u] = i < nc ? (u64)S.cidx[cur * KC + i] * 8 : 0;
          if (i < nc) {
            if (from_recs) {
              const uint4* r4 = reinterpret_cast<const uint4*>(f.sax + c8[u]);
              const uint4 a = r4[0], b = r4[1], c = r4[2], d = r4[3];
              v[u][0] = a.x, v[u][1] = a.z, v[u][2] = b.x, v[u][3] = b.z, v[u][4] = c.x, v[u][5] = c.z, v[u][6] = d.x, v[u][7] = d.z;
            } else {
              const uint4 x = *reinterpret_cast<const uint4*>(lv + c8[u]), y = *reinterpret_cast<const uint4*>(lv + c8[u] + 4);
              v[u][0] = x.x, v[u][1] = x.y, v[u][2] = x.z, v[u][3] = x.w, v[u][4] = y.x, v[u][5] = y.y, v[u][6] = y.z, v[u][7] = y.w;
            }
          }
        }
#pragma unroll
        for (u32 u = 0; u < 2; ++u) {
          const u32 i = base + u * 256 + threadIdx.x;
          if (i < nc) {
#pragma unroll
            for (int t = 0; t < 8; ++t) S.val[8 * i + t] = c8[u] + t < nlow ? v[u][t] : TOPK_PAD;
          }
        }
      }
      const u64 a1 = A(j), b1 = B(j), a0 = A(j - 1), b0 = B(j - 1);
      const u32 nl = (u32)(8 * a1 - a0), nr = (u32)(b0 - 8 * b1);
      if (threadIdx.x < nl + nr) {
        const u64 e = threadIdx.x < nl ? a0 + threadIdx.x : 8 * b1 + (threadIdx.x - nl);
        S.eidx[threadIdx.x] = (u32)e;
        S.val[8 * nc + threadIdx.x] = from_recs ? f.sax[e].x : lv[e];
      }
      reads += 8ULL * nc + nl + nr;
      nv = 8 * nc + nl + nr;
      nc_prev = nc;
      __syncthreads();
      p_lev += DG_LPROF_NOW() - pt1;
      }
    }
  }
  if (threadIdx.x == 0 && reads) atomicAdd(&ctr->sa_reads[blockIdx.x & (NSHARD - 1)], (unsigned long long)reads);
}

}  // namespace dg
