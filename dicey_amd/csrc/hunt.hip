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

#include "hunt_search.hpp"
#include "hunt_select.hpp"
#include "hunt_locate.hpp"
#include "hunt_verify.hpp"

namespace dg {

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
                    const dg_switches& sw, std::vector<u32>* dev_jobs = nullptr) {
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
  if (sw.cap_budget_mb) budget = sw.cap_budget_mb << 20;
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
  if (sw.host_threads) nthreads = sw.host_threads;
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
  const dg_switches& sw = ix->sw;  // the environment switches, read by the entry point on the caller's thread
  const bool host_timing = sw.host_timing;  // host phases of every batch to stderr
  // HIP events between the stages (ms_select / ms_locate / ms_verify of the result) only on request: every record is a marker packet
  // the next kernel waits behind; the batch's total and the flat search kernel's time are always measured (four events)
  const bool phase_events = (p->flags & DG_HUNT_PHASE_TIMES) != 0 || sx || group_counts;
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
  const bool no_band = sw.no_band;  // (test switches are read per batch: the GPU suite flips them inside one process)
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
      // read back from the device, so never seen by a host pass: cap_scan indexes with them
      if (ho[nq] != total) return fail(DG_EINVAL, "dg_hunt_device: total_qbytes does not match qoff[nq]");
      for (size_t i = 0; i < nq; ++i)
        if (ho[i + 1] < ho[i] || ho[i + 1] - ho[i] > maxlen)
          return fail(DG_EINVAL, "dg_hunt_device: qoff decreases at query %zu, or the query exceeds the %u nt this batch was sized for", i, maxlen);
      sb = hb.data();
      so = ho.data();
    }
    // DICEY_CAP_HOST: every capped neighbourhood on the host (nbhd_host.hpp), as before r04 — the GPU suite runs both
    DG_TRY(cap_scan(sb, so, nq, p, group_counts != nullptr, cs, sw, sw.cap_host ? nullptr : &dev_jobs));
  }
  u64 nxs = cs.xs_gid.size();
  if (nxs >= 0x0FFFFFFFull || cs.xs_bytes.size() > (48ull << 30))
    return fail(DG_ELIMIT, "%llu explicit neighbourhood strings in one batch; pass fewer sequences per call", (unsigned long long)nxs);
  // room for what k_cap_enum can append: <= max(maxsize, 1) strings of <= maxlen + d characters per strand of its queries
  const u64 dev_strings = dev_jobs.size() * (p->forward_only ? 1ull : 2ull) * std::max<u64>(p->max_neighborhood, 1);
  const u64 dev_bytes = dev_strings * (maxlen + dmax_eff);
  if (!dev_jobs.empty()) {
    u64 budget = 24ull << 30;
    if (sw.cap_budget_mb) budget = sw.cap_budget_mb << 20;
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
    // persistent workgroups; the kernel is chains of atomics and probes per workgroup, so its time falls with the number of
    // workgroups in flight and the independent accesses each keeps in flight (r04 first form: 512 workgroups, one access at a
    // time per lane: 46.7 ms per 2 000 25-mers)
    const u32 nwg = std::min<u32>(njobs, 1024u);  // four per CU fit (113 VGPRs)
    const u64 leaves = cap::total_leaves(std::min<u32>(maxlen, cap::MAX_KEY_LEN - dmax_eff), dmax_eff) + 2;
    // table slots: twice the number of DISTINCT strings a strand can hold (neighbourhood_bound; 15 705 for a 25-mer at distance 2
    // against 20 201 leaves) — 512 KB per strand instead of 1 MB for 25-mers
    const u64 distinct = std::min<u64>(leaves, neighbourhood_bound(std::min<u32>(maxlen, cap::MAX_KEY_LEN - dmax_eff), dmax_eff, true, 0));
    u32 tcap_log2 = 8;
    while ((1ull << tcap_log2) < 2 * distinct) ++tcap_log2;
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
  // (r05: Hamming distance 2 as well — the same kernel with "no edit" in place of the deletions)
  b.fast2K = (dmax_eff == 2 && (indel || !sw.no_flat_ham2) && ix->view.K && maxlen >= ix->view.K + 2 && ngrp < 0x7FFFFFFFull) ? ix->view.K : 0u;
  b.tabK = sw.no_nwin ? 0u : ix->view.K;
  b.nrun_min = ((dmax_eff == 1 || dmax_eff == 2) && ngrp < 0x7FFFFFFFull) ? ix->view.nrun_min : 0u;  // (non-zero: k_nres is launched with the generic kernels)
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
  b.nwin = &ctr->pad_[7];
  b.short2 = &ctr->pad_[9];
  b.fast2_minlen = 0;
  b.maxlen_bound = maxlen;
  b.total_qbytes = total;
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

  u32 base_gen = 0;
  hipEvent_t base_ev = nullptr;
  if (dg_index::SharedHints* sh = ix->shared_hints.load()) {  // what the handle's other lanes have learnt since this lane's previous batch
    std::lock_guard<std::mutex> lk(sh->mu);
    // the lanes' common timeline: this lane's stream is idle here (its previous batch has been synchronised), so a base recorded on it
    // completes at once; the other base stays valid for the other lanes' batches in flight
    const double now = host_us();
    if (!sh->base_gen || now - sh->base_host_us > 4e6) {
      const u32 g = sh->base_gen + 1;
      if (!sh->ev_base[g & 1] && hipEventCreate(&sh->ev_base[g & 1]) != hipSuccess) sh->ev_base[g & 1] = nullptr;
      if (sh->ev_base[g & 1] && hipEventRecord(sh->ev_base[g & 1], st) == hipSuccess) {
        sh->base_gen = g;
        sh->base_host_us = now;
      }
    }
    base_gen = sh->base_gen;
    base_ev = base_gen ? sh->ev_base[base_gen & 1] : nullptr;
    if (sh->valid) {
      ix->shard_cap_hint = std::max(ix->shard_cap_hint, sh->shard_cap);
      ix->flat_cap_hint = std::max(ix->flat_cap_hint, sh->flat_cap);
      ix->hit_cap_hint = std::max(ix->hit_cap_hint, sh->hit_cap);
      if (sh->fetch_hits) ix->fetch_hits_hint = sh->fetch_hits;  // (the most recent batch's, like the two below)
      ix->generic_sticky = std::max(ix->generic_sticky, sh->generic_sticky);
      ix->jobs_sticky = std::max(ix->jobs_sticky, sh->jobs_sticky);
      ix->generic_hint = ix->generic_sticky > 0;
      ix->jobs_hint = ix->jobs_sticky > 0;
      ix->jobs_big_hint = sh->jobs_big;
      ix->fused_leaves_hint = sh->fused_leaves;
    }
  }
  u32 shard_cap = std::max<u32>(ix->shard_cap_hint, (u32)std::max<u64>(64, (16 * (u64)nq + nxs / 8) / NSHARD));
  u64 hit_cap = std::max<u64>(ix->hit_cap_hint, 4 * (u64)nq + 1024);
  // slices of the flat Sel region (k_search1s): what the previous batch's fullest slice needed plus a quarter — kept strings are a
  // third of the leaf estimate above, and k_locate walks every slot of the region
  u32 flat_req = ix->flat_cap_hint ? ix->flat_cap_hint : shard_cap;
  if (sw.debug_caps) {  // tests: start from tiny capacities to exercise the retry path
    const char* e = sw.debug_caps_s.c_str();
    shard_cap = (u32)std::max(1, std::atoi(e));
    hit_cap = (u64)std::max(1, std::atoi(e));
    flat_req = shard_cap;
  }
  u64 nleaf = 0, nhits = 0;
  bool force_generic = false, force_jobs = false, force_short2 = false;
  u64 walk_cap = 0;  // room of the walker's group list (k_walk_list)
  u32 flat_form = 0;  // which flat search kernel the (last) attempt launched: dg_hunt_result::flat_kernel_form
  u32 verify_form = 0;  // ... and which k_verify_memo instantiation: dg_hunt_result::verify_kernel_form
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
  std::vector<u32> dump_cnt;
  for (int attempt = 0;; ++attempt) {
    if (attempt > 8) return fail(DG_ENOMEM, "buffer overflow persists (%llu leaves, %llu hits)", (unsigned long long)nleaf, (unsigned long long)nhits);
    const u64 leaf_slots = (u64)NSHARD * shard_cap;

    // Distance 1, every string in 128 bits: k_search1s settles the select stage inside the search kernel (flat Sel region of
    // NSHARD slices).  The generic kernels (k_search for N-containing / long queries, k_explicit, scan, pack, alive, rank) run
    // when the previous batch of this handle had work for them; a batch that turns out to need them after all is repeated.
    // DICEY_NO_FUSED_SELECT: the search without the select stage (k_search1p + the generic select kernels) — what batches with
    // strings above 42 characters take anyway; the GPU tests run every distance-1 case both ways
    const bool no_fuse = sw.no_fuse;
    // r04: the same at edit distance 2 (k_search2p<true, .>: one workgroup per group or per query); DICEY_NO_FUSED_SELECT2 keeps the
    // generic select kernels for that distance only
    const bool fused = (b.fastK || (b.fast2K && !sw.no_fuse2)) && packed && !no_fuse;
    const u32 flat_cap = fused ? flat_req : 0u;
    const u64 flat_slots = (u64)NSHARD * flat_cap;
    const bool generic_on = !fused || ix->generic_hint || nxs > 0 || force_generic;
    bool jobs_on = false, walker_on = true, with_ctx_hits = false;
    u32 direct_ctx = 0;
    bool keys_on = false;  // this attempt's k_search1s wrote the kept strings' keys (FlatSel::key)
    DG_TRY(ws[WS_LEAF].reserve(leaf_slots * sizeof(Leaf)));
    DG_TRY(ws[WS_LEAFG].reserve((leaf_slots + 1) * (sizeof(Leaf) > sizeof(PLeaf) ? sizeof(Leaf) : sizeof(PLeaf))));
    DG_TRY(ws[WS_SEL].reserve((flat_slots + leaf_slots + 1) * sizeof(Sel)));
    Sel* const sel_all = ws[WS_SEL].as<Sel>();
    Sel* const sel_gen = sel_all + flat_slots;  // the generic path's slots: grp_off based, behind the flat region
    DG_TRY(ws[WS_SCR].reserve((leaf_slots + 1) * 5 + 64));
    DG_TRY(ws[WS_SEEDS].reserve((hit_cap + 1) * sizeof(HitSeed)));
    const u32 surv_cap = 0xFFFFFFFFu;  // (r02's survivor queue in HBM is gone with k_probe1 / k_finish1; the scan kernel's check keeps its argument)
    // locate jobs: three lists of JOB_SHARDS regions each (hunt_locate.hpp); a region that fills up turns its strings away to the
    // queueing lane, so the size is a matter of speed only
    const u32 job_shard_cap = (u32)(std::min<u64>(leaf_slots, 1u << 20) / JOB_SHARDS * 3 / 2 + 32);
    DG_TRY(ws[WS_JOBS].reserve(3 * (u64)JOB_SHARDS * job_shard_cap * sizeof(BigJob)));
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
    // distance 2: the LONG2 body of k_search2p takes queries whose shortest string still asks the long filter; the handle falls back to
    // the r04 body while its batches hold shorter ones (short2_sticky), and a batch that turns out to hold some is repeated with it
    const bool long2 = b.fast2K && ix->view.kf2.nr && ix->view.kf2.k <= 18 /* word offsets inside a copy stay below 2^32 */ && !ix->short2_sticky && !force_short2 && !sw.no_long2;
    b.fast2_minlen = long2 ? ix->view.kf2.k + (indel ? 2u : 0u) : 0u;
    const bool prep_in = fused && b.fastK && !generic_on && !group_counts && !sw.no_prep_fusion;
    flat_form = b.fastK ? (fused ? (prep_in ? 3u : 2u) : 1u) : b.fast2K ? (fused ? 5u : 4u) + (long2 ? 2u : 0u) : 0u;
    // the per-character arrays (fw / rv codes, normalised ASCII) are read by the generic kernels, the full-matrix verify kernels and
    // the classic result fetch only: 60 byte stores per query that the flat path with compact results does without
    const u32 write_bytes = (prep_in && band_verify && (compact || !fetch)) ? 0u : 1u;
    hipLaunchKernelGGL(k_prepare, dim3(ceil_div(nq, TB)), dim3(TB), 0, st, b, grp_cnt, nsel, selbase, (u32*)&ctr->pad_[6], write_bytes);
    DG_HIP(hipEventRecord(ix->ev[1], st));
    {
      SearchOut so;
      WalkList kl{nullptr, nullptr, 0u};  // the strands of k_nkeep (k_walk_list)
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
          // r06: strings with one occurrence leave with their characters when the locate / verify stages of this batch can use them
          // (direct_ctx below: the records are resident, distance <= 1, banded verify)
          fs.key = nullptr;
          if (!sx && !group_counts && band_verify && ix->view.sax && indel && dmax_eff <= 1 && !sw.no_direct_ctx) {
            DG_TRY(ws[WS_SELKEY].reserve((flat_slots + 1) * 8));
            fs.key = ws[WS_SELKEY].as<u64>();
            keys_on = true;
          }
          const dim3 g1(ceil_div(ngrp, gpw)), b1(256);
          // LDS list of a workgroup: 256 entries unless the previous batch of this handle averaged more than 48 occurring strings
          // per workgroup (repeat-bearing genomes), then 512; a workgroup whose strings do not fit hands its groups to the generic
          // path.  tests: DICEY_FUSED_LCAP lowers the capacity so that ordinary batches exercise that hand-over
          const u32 lcap_env = sw.fused_lcap ? std::max<u32>(1u, std::min<u32>(FUSED_LCAP, sw.fused_lcap)) : 0u;
          const u32 lcap = lcap_env ? lcap_env : (ix->fused_leaves_hint > 48ull * g1.x ? FUSED_LCAP : FUSED_LCAP / 2);
          const u32 lds1 = fused_lds_bytes(lcap);
          const u32 leave1 = 1u | (sw.exp_bits << 1);  // bit 0: idle wavefronts end behind the probe phase (r04 A/B on one box: 0.2188 ms with, 0.2194 without — harmless, kept)
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
        // filtered leaves only where the packed select path (k_group_pack) hands the filter word on
        const u32 filt_ok = (u32)(packed && !sw.no_pre5_d2);
        FlatSel fs;
        fs.sel = sel_all;
        fs.cap = flat_cap;
        fs.selbase = selbase;
        fs.nsel = nsel;
        fs.key = nullptr;
        // (tests: DICEY_FUSED_LCAP lowers the list's capacity so that ordinary groups exercise the hand-over to the generic select kernels)
        const u32 lcap2 = sw.fused_lcap ? std::max<u32>(1u, std::min<u32>(FUSED2_LCAP, sw.fused_lcap)) : FUSED2_LCAP;
        const u32 ham2 = (indel ? 0u : 1u) | (sw.exp_bits << 8);  // (DICEY_EXP: k_search2p's measurement switches)
        if (fused && long2) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_search2p<true, true>), dim3((u32)ngrp), dim3(TB), 0, st, ix->view, b, so, filt_ok, fs, lcap2, ham2);
        else if (fused) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_search2p<true, false>), dim3((u32)ngrp), dim3(TB), 0, st, ix->view, b, so, filt_ok, fs, lcap2, ham2);
        else if (long2) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_search2p<false, true>), dim3((u32)ngrp), dim3(TB), 0, st, ix->view, b, so, filt_ok, fs, lcap2, ham2);
        else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_search2p<false, false>), dim3((u32)ngrp), dim3(TB), 0, st, ix->view, b, so, filt_ok, fs, lcap2, ham2);
        DG_HIP(hipEventRecord(ix->ev[8], st));
      }
      // edit distance 2: the walker (k_search) only serves the groups k_search2p does not take (N in the query, above 30 nt); it is
      // launched when the previous batch of the handle had such groups — 0.17 ms of the 13 ms step on batches that have none — and
      // a batch that has some after all is repeated with it, like the generic kernels at distance 1
      walker_on = !(b.fast2K && !ix->generic_hint && !force_generic && nxs == 0);
      if (generic_on && walker_on) {
      // root-level work split (see k_search): only with the table and with at least one edit to place
      // (r05: also beside the flat distance-1 kernel — the walker is only launched when a batch has groups for it, and those are the
      // queries with N that the split turns from one 1 500-read chain per strand into lanes of a handful of reads)
      // — while the handle's batches hold such strands (nwin_sticky): 29 M lanes that leave at once cost a repeats-genome step 0.15 ms
      const u32 items = (ix->view.K && dmax_eff >= 1 && (!b.fastK || ix->nwin_sticky)) ? ix->view.K * (indel ? 9u : 4u) + 1u : 1u;
      // beside a flat kernel the walker serves a few strands of the batch: with the root split its lanes cover a LIST of them
      // (k_walk_list; room for what the previous batch listed plus a quarter — a list that overflows repeats the batch)
      WalkList wl{nullptr, nullptr, 0u};
      u64 wgroups = ngrp;
      // (r06 fix: also for a batch WITHOUT a flat kernel — queries shorter than the table order — when k_prepare may mark strands for
      //  k_nkeep (b.nrun_min): they are found through this list only; the list then simply holds every group, no hint involved.
      //  tools/fuzz_hunt.py in the full-size layout found a 12-mer ending in N that lost its two exact hits in Hamming mode)
      const bool flat_on = b.fastK || b.fast2K;
      if ((flat_on || b.nrun_min) && ngrp < 0x7FFFFFFFull) {
        walk_cap = std::max<u64>(walk_cap, flat_on ? std::min<u64>(ngrp, (u64)ix->walk_hint + ix->walk_hint / 4 + 4096) : ngrp);
        DG_TRY(ws[WS_WALK].reserve(2 * (walk_cap + 1) * 4));
        wl.gid = ws[WS_WALK].as<u32>();
        wl.count = &ctr->pad_[10];
        wl.cap = (u32)walk_cap;
        wgroups = walk_cap;
        kl.gid = ws[WS_WALK].as<u32>() + walk_cap + 1;  // the strands of k_nkeep, behind the walker's groups
        kl.count = &ctr->pad_[11];
        kl.cap = (u32)walk_cap;
        hipLaunchKernelGGL(k_walk_list, dim3(ceil_div(ngrp, TB)), dim3(TB), 0, st, b, ws[WS_WALK].as<u32>(), &ctr->pad_[10],
                           ws[WS_WALK].as<u32>() + walk_cap + 1, &ctr->pad_[11], wl.cap, ctr);
      }
      const dim3 grid(ceil_div(wgroups * items, TB)), block(TB);
#define DG_LAUNCH_SEARCH(IND, DD) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_search<IND, DD>), grid, block, 0, st, ix->view, b, so, items, wl)
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
      // strands whose N's can only be substituted or deleted (k_prepare, bit 12 of GidInfo::d_win): a lane per resolved string
      if (generic_on && b.nrun_min) {
        const u32 per = indel ? (dmax_eff == 1 ? 5u : 25u) : (dmax_eff == 1 ? 4u : 16u);
        if (indel) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_nres<true>), dim3(ceil_div(ngrp * per, TB)), dim3(TB), 0, st, ix->view, b, so, per);
        else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_nres<false>), dim3(ceil_div(ngrp * per, TB)), dim3(TB), 0, st, ix->view, b, so, per);
        // ... and the strings of the listed bit-13 strands that keep their N: a lane per candidate edit
        if (kl.gid) {
          const u32 kper = 1u + (indel ? 9u : 3u) * std::min(maxlen, 32u);
          if (indel) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_nkeep<true>), dim3(ceil_div((u64)kl.cap * kper, TB)), dim3(TB), 0, st, ix->view, b, so, kl, kper);
          else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_nkeep<false>), dim3(ceil_div((u64)kl.cap * kper, TB)), dim3(TB), 0, st, ix->view, b, so, kl, kper);
        }
      }
    }
    if (phase_events) DG_HIP(hipEventRecord(ix->ev[2], st));
    // (r03 tried single-launch scans chained by decoupled look-back, and k_take fused with its scan: 17 us against 2 x 4.3 us, and
    //  47 us against 7 + 9 us — descriptor polling with device-scope acquire / release is slow across the XCDs' L2s.  What stays of
    //  that round: the first scan kernel also checks the search kernels' buffers, which was a launch of its own.)
    if (generic_on) DG_TRY(device_scan(st, grp_cnt, ngrp, grp_off, scan_buf, ctr, shard_cap, surv_cap));
    if (phase_events) DG_HIP(hipEventRecord(ix->ev[3], st));
    if (packed) {
      if (generic_on) {
      u8* alive = ws[WS_SCR].as<u8>();
      // the leaves' filter words sit behind the packed leaves: the buffer is sized for 36-byte Leaf records, PLeaf takes 32
      static_assert(sizeof(Leaf) >= sizeof(PLeaf) + sizeof(u32), "room for the filter words");
      u32* filt = (u32*)((u8*)ws[WS_LEAFG].p + (leaf_slots + 1) * sizeof(PLeaf));
      hipLaunchKernelGGL(k_group_pack, dim3(ceil_div(leaf_slots, TB)), dim3(TB), 0, st, b, ws[WS_LEAF].as<Leaf>(), shard_cap, ctr,
                         grp_off, ws[WS_LEAFG].as<PLeaf>(), filt);
      // distance >= 2: groups of up to SELCAP strings are sorted by a workgroup each, the lane-per-leaf kernels keep the rest
      const u32 above = (dmax_eff >= 2 && ngrp < 0x7FFFFFFFull) ? SELCAP : 0u;
      if (above)
        // (one wavefront per group was tried: 1.95 -> 2.5 ms, the second wavefront's share of the window searches is worth more than the barriers)
        hipLaunchKernelGGL(k_group_select, dim3((u32)ngrp), dim3(128), 0, st, ws[WS_LEAFG].as<PLeaf>(), (const u32*)filt, grp_off, (u32)indel,
                           sel_gen, nsel, ctr);
      hipLaunchKernelGGL(k_leaf_alive, dim3(ceil_div(leaf_slots, TB)), dim3(TB), 0, st, ws[WS_LEAFG].as<PLeaf>(), grp_off, ngrp,
                         (u32)indel, alive, ctr, above);
      hipLaunchKernelGGL(k_leaf_rank, dim3(ceil_div(leaf_slots, TB)), dim3(TB), 0, st, ws[WS_LEAFG].as<PLeaf>(), (const u32*)filt, grp_off, ngrp,
                         (const u8*)alive, sel_gen, nsel, ctr, above);
      }
      if (!prep_in)
        hipLaunchKernelGGL(k_take, dim3(ceil_div(nq * 16, 256)), dim3(256), 0, st, b, (const u64*)grp_off, (const u32*)selbase, flat_slots, (const u32*)nsel, sel_all,
                           qhits, ctr);
    } else {
      hipLaunchKernelGGL(k_group, dim3(ceil_div(leaf_slots, TB)), dim3(TB), 0, st, ws[WS_LEAF].as<Leaf>(), shard_cap, ctr, grp_off,
                         ws[WS_LEAFG].as<Leaf>());
      u32* scr_rank = ws[WS_SCR].as<u32>();
      u8* scr_keep = (u8*)(scr_rank + leaf_slots + 1);
      hipLaunchKernelGGL(k_select, dim3(ceil_div(nq, 64)), dim3(64), 0, st, b, ws[WS_LEAFG].as<Leaf>(), grp_off, sel_gen,
                         nsel, qhits, scr_keep, scr_rank, ctr);
    }
    if (phase_events) DG_HIP(hipEventRecord(ix->ev[4], st));
    if (group_counts) {
      DG_TRY(ws[WS_HITS].reserve(ngrp * 8 + 64));
      hipLaunchKernelGGL(k_group_count, dim3(ceil_div(ngrp, TB)), dim3(TB), 0, st, (const u64*)grp_off, (const u32*)selbase, flat_slots,
                         (const u32*)nsel, (const Sel*)sel_all, ngrp, ws[WS_HITS].as<u64>(), ctr);
      DG_HIP(hipMemsetAsync(hit_off + nq, 0, 8, st));  // no hits in this mode
      for (int e = 5; e <= 7; ++e)
        if (phase_events || e == 7) DG_HIP(hipEventRecord(ix->ev[e], st));
    } else {
    DG_TRY(device_scan(st, qhits, nq, hit_off, scan_buf));
    if (phase_events) DG_HIP(hipEventRecord(ix->ev[5], st));
    {
      // strings with many occurrences go to three job lists: up to 256 occurrences for a wavefront each, more for a workgroup each
      // (those that fit the small buffer of k_locate_topk whole, and the rest)
      jobs_on = ix->jobs_hint || force_jobs || p->max_locations > TOPK_KMAX;
      // repeat-rich strings: up to TOPK_KMAX positions through the block minima, more (hunt -m above 1 024) by radix passes
      const bool topk = ix->view.nlev > 1;
      // A batch with thousands of repeat-rich strings (the previous batch of this handle tells): intervals up to 4 608 entries go
      // to the small-buffer form of k_locate_topk.  (r03 also ran the three job kernels side by side on helper streams: each
      // slowed down by what the others took — 114 / 220 / 228 us alone, 220 / 494 / 268 us together — and the stage gained
      // 0.08 of 0.91 ms; not worth two more streams per handle.)
      const bool rich = topk && ix->jobs_big_hint >= 2048;
      LocJobs lj;
      for (int l = 0; l < 3; ++l) lj.list[l] = ws[WS_JOBS].as<BigJob>() + (u64)l * JOB_SHARDS * job_shard_cap;
      lj.shard_cap = job_shard_cap;
      lj.cnt = ctr->job_cnt;
      lj.mid_max = rich ? 8 * TOPK_KCAP_MID : 0u;
      // hits of the job kernels leave with their context characters (FmView::sax) when the kernel that reads the seeds is
      // k_verify_memo — the only reader that knows the encoding (HitSeed::len, devfm.hpp)
      const u32 with_ctx = (!sx && !group_counts && band_verify && ix->view.sax) ? 1u : 0u;
      with_ctx_hits = with_ctx != 0;
      // r06: the same for a string with ONE occurrence that k_locate serves itself, when the search kernel knew the character in front
      // of it (Sel::key / SEL_CTX_VALID): one context character either side, i.e. batches at distance <= 1 (DICEY_NO_DIRECT_CTX: tests)
      direct_ctx = (with_ctx && keys_on && ix->view.K >= SAX_POST_OFF && ix->view.K + 1 <= SAX_POST_OFF + SAX_POST_N) ? 1u : 0u;
      // the record that shares a kept string's slot names its group: the packed leaf (qs behind 28 bytes) or the grouped leaf (qs first)
      const u8* slot_qs = packed ? (const u8*)ws[WS_LEAFG].p + offsetof(PLeaf, qs) : (const u8*)ws[WS_LEAFG].p + offsetof(Leaf, qs);
      const u32 slot_stride = packed ? (u32)sizeof(PLeaf) : (u32)sizeof(Leaf);
      hipLaunchKernelGGL(k_locate, dim3(ceil_div(flat_slots + (generic_on ? leaf_slots : 0), TB)), dim3(TB), 0, st, ix->view, (const Sel*)sel_all, slot_qs,
                         slot_stride, (const u64*)grp_off, (const u32*)nsel, ngrp, (const u64*)hit_off, ws[WS_SEEDS].as<HitSeed>(), ctr, hit_cap, lj,
                         flat_slots, flat_cap, (u32)generic_on, (u32)jobs_on, direct_ctx);
      // The job kernels (strings of more than 16 occurrences) are launched when the previous batch of this handle queued any job;
      // k_locate queues and counts whether or not they run, and a batch that had jobs after one that had none is repeated with
      // them (the same device as for capacity guesses).  Uniform batches on a genome without repeats never launch them.
      if (jobs_on) {
        hipLaunchKernelGGL(k_locate_small, dim3(8192), dim3(64), 0, st, ix->view, lj, ws[WS_SEEDS].as<HitSeed>(), ctr, with_ctx);
        if (topk) {
          if (lj.mid_max)
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_locate_topk<TOPK_KCAP_MID>), dim3(1280), dim3(256), 0, st, ix->view, lj, (u32)JL_MID,
                               ws[WS_SEEDS].as<HitSeed>(), ctr, with_ctx);
          hipLaunchKernelGGL(HIP_KERNEL_NAME(k_locate_topk<TOPK_KCAP>), dim3(768), dim3(256), 0, st, ix->view, lj, (u32)JL_BIG,
                             ws[WS_SEEDS].as<HitSeed>(), ctr, with_ctx);
        }
        if (!topk || p->max_locations > TOPK_KMAX)
          hipLaunchKernelGGL(k_locate_big, dim3(1024), dim3(256), 0, st, ix->view, lj, ws[WS_SEEDS].as<HitSeed>(), ctr, topk ? TOPK_KMAX : 0u);
      }
      if (!sw.dump_jobs.empty()) {  // development aid: the regions' fill counts, before the summary kernel clears them
        dump_cnt.assign(3 * JOB_SHARDS, 0);
        DG_HIP(hipMemcpyAsync(dump_cnt.data(), ctr->job_cnt, sizeof(u32) * 3 * JOB_SHARDS, hipMemcpyDeviceToHost, st));
      }
    }
    if (phase_events) DG_HIP(hipEventRecord(ix->ev[6], st));
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
      va.selkey = direct_ctx ? (const u64*)ws[WS_SELKEY].p : nullptr;
      va.debug = (sw.exp_bits >> 4) & 3u;  // (DICEY_EXP bits 4-5, measurement aid: verify without its alignments / without the hits' context)
      const u32 cells = (maxlen + 3 * dmax_eff + 1) * (maxlen + 1);
      const u32 VT = 128;
      const dim3 vgrid(ceil_div(hit_cap, VT)), vblock(VT);
      if (band_verify) {
        va.chits = d_chits;  // the banded kernel writes compact records itself
        // hits per lane: 1 while a query has a handful of hits (every window is its own class: nothing to share, smallest LDS
        // footprint), 8 when the previous batch had dozens of hits per query (repeat families: ~240 hits per kept string)
        const int ch_env = sw.verify_ch;
        const u64 per_q = hit_cap / std::max<u64>(nq, 1);
        // (distance 2 on an i.i.d. genome: 59 hits per query from ~50 strings — nothing to share, and the wider trace of 13 diagonals
        //  leaves room for fewer workgroups: r03 measured 2.39 ms with 8 hits per lane against 1.83 ms for the lane-per-hit kernel)
        const u64 share_at = dmax_eff > 1 ? 192 : 24;
        // r06: with the hits' context in their seeds (FmView::sax) the kernel no longer waits for a text line per hit, and what it
        // answers to is residency: 4 hits per lane leave room for four workgroups per CU, 8 for two (one batch at a time on the
        // repeats genome, verify: 0.61 ms with 8, 0.48 ms with 4)
        const int ch_rich = (with_ctx_hits && dmax_eff <= 1) ? 4 : 8;
        const int ch = ch_env == 1 || ch_env == 4 || ch_env == 8 ? ch_env : (per_q >= share_at ? ch_rich : (per_q >= share_at / 2 ? 4 : 1));
        const u32 rows = maxlen + 3 * dmax_eff + 2;
        const bool wide = dmax_eff > 1;
        const u32 nw_bytes = ((rows * 256 * (wide ? 4u : 2u) + 7u) & ~7u) + 6u * 256u * 8u, hash_bytes = 2u * 256u * (u32)ch * 10u;
        const u32 lds = std::max(nw_bytes, hash_bytes);
        const dim3 mgrid(ceil_div(hit_cap, (u64)256 * ch)), mblock(256);
        verify_form = ((wide ? 13u : 7u) << 8) | (u32)ch;
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
    if (const char* dj = (sw.dump_jobs.empty() || dump_cnt.empty()) ? nullptr : sw.dump_jobs.c_str()) {  // development aid: (lo, occs, take, len) of the batch's locate jobs, list after list
      if (FILE* fj = std::fopen(dj, "wb")) {
        std::vector<BigJob> hj;
        for (u32 l = 0; l < 3; ++l)
          for (u32 sh = 0; sh < JOB_SHARDS; ++sh) {
            const u32 c = std::min(dump_cnt[l * JOB_SHARDS + sh], job_shard_cap);
            if (!c) continue;
            hj.resize(c);
            DG_HIP(hipMemcpy(hj.data(), ws[WS_JOBS].as<BigJob>() + ((u64)l * JOB_SHARDS + sh) * job_shard_cap, (size_t)c * sizeof(BigJob), hipMemcpyDeviceToHost));
            for (const BigJob& j : hj) {
              const u32 rec[4] = {j.lo, j.occs, j.take, j.len | (l << 28)};
              std::fwrite(rec, 4, 4, fj);
            }
          }
        std::fclose(fj);
      }
    }
#ifdef DG_TOPK_PROFILE
    {
      static const char* nm[24] = {"mid.jobs", "mid.load", "mid.threshold", "mid.keys", "mid.sort", "mid.write", "mid.levels", "mid.total",
                                   "big.jobs", "big.load", "big.threshold", "big.keys", "big.sort", "big.write", "big.levels", "big.total",
                                   "small.jobs", "small.load", "small.sort", "small.write", "small.total", "-", "-", "-"};
      std::fprintf(stderr, "topk profile (us summed over workgroups):");
      for (int i = 0; i < 21; ++i) std::fprintf(stderr, " %s=%.1f", nm[i], (i % 8 == 0 && i < 17) ? (double)hsum.prof[i] : hsum.prof[i] * 0.01);
      std::fprintf(stderr, "\n");
    }
#endif
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
    if (!walker_on && hsum.n_generic > 0) {  // edit distance 2: groups for the walker after all
      ix->generic_hint = true;
      ix->generic_sticky = 8;
      force_generic = true;
      again = true;
    }
    if (long2 && hsum.n_short2 > 0) {  // queries too short for the LONG2 body went to the walker (or to nobody): once more with the r04 body
      ix->short2_sticky = 8;
      force_short2 = true;
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
    if (hsum.overflow & 8u) {  // the walker's group list was too short
      walk_cap = std::min<u64>(ngrp, hsum.n_walk + hsum.n_walk / 4 + 4096);
      again = true;
    }
    ix->walk_hint = (u32)std::min<u64>(hsum.n_walk, 0xFFFFFFFFull);
    if (again) continue;
    // The hints are sticky: a kernel family that a batch needed stays on for the next eight batches of the handle, so that a stream
    // that alternates (chunks with and without N-containing queries, with and without repeat-rich strings) does not pay a repeated
    // batch at every change (r03 advice; r04 repeats genome: 4 of 14 rotating steps ran twice).
    if (b.fast2K && !fused) {
      if (hsum.n_generic > 0 || nxs > 0) ix->generic_sticky = 8;
      else if (ix->generic_sticky) --ix->generic_sticky;
      ix->generic_hint = ix->generic_sticky > 0;
    }
    if (fused) {
      if (hsum.nleaf > 0 || hsum.n_generic > 0 || nxs > 0) ix->generic_sticky = 8;
      else if (ix->generic_sticky) --ix->generic_sticky;
      ix->generic_hint = ix->generic_sticky > 0;
    }
    if (hsum.n_nwin) ix->nwin_sticky = 8;
    else if (ix->nwin_sticky) --ix->nwin_sticky;
    if (b.fast2K && !long2 && ix->short2_sticky && !force_short2) --ix->short2_sticky;  // (eight batches on the r04 body, then LONG2 is tried again)
    if (!group_counts) {
      if (hsum.jobs_small > 0 || hsum.jobs_big > 0) ix->jobs_sticky = 8;
      else if (ix->jobs_sticky) --ix->jobs_sticky;
      ix->jobs_hint = ix->jobs_sticky > 0;
    }
    break;
  }
  ix->shard_cap_hint = shard_cap;
  if ((b.fastK || b.fast2K) && hsum.worst_sel) ix->flat_cap_hint = (u32)std::max<unsigned long long>(ix->flat_cap_hint, hsum.worst_sel + hsum.worst_sel / 4 + 64);
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
    R->d_block = ws[WS_PACK].p;
    R->d_block_bytes = 8ull * nq + nhits * (u64)chit_words * 4;
  } else {
    R->d_hits = ws[WS_HITS].p;
    R->d_ops = ops_per_hit ? ws[WS_OPS].p : nullptr;
  }
  R->stream = (void*)st;
  R->ctr_leaves = nleaf + hsum.fused_leaves;
  R->ctr_ext_steps = hsum.steps;
  R->ctr_tab_reads = hsum.lookups;
  R->ctr_filter_probes = hsum.probes;
  R->ctr_sa_reads = hsum.sa_reads;
  R->ctr_win_bytes = hsum.win_bytes;
  R->ms_total = ev_ms(ix->ev[0], ix->ev[7]);
  R->ms_search_flat = (b.fastK || b.fast2K) ? ev_ms(ix->ev[1], ix->ev[8]) : 0.0;
  if (phase_events) {
    R->ms_search = ev_ms(ix->ev[1], ix->ev[2]);
    R->ms_select = ev_ms(ix->ev[3], ix->ev[4]);
    R->ms_locate = ev_ms(ix->ev[5], ix->ev[6]);
    R->ms_verify = ev_ms(ix->ev[6], ix->ev[7]);
  } else R->ms_search = R->ms_search_flat;
  R->ms_cap = ms_cap;
  R->cap_queries_device = dev_jobs.size();
  R->cap_queries_host = cs.looked_at;
  R->cap_patterns = nxs;
  R->flat_kernel_form = flat_form;
  R->verify_kernel_form = verify_form;
  R->t_base_gen = 0;
  R->t_search_begin_ms = R->t_search_end_ms = 0.0;
  if (base_ev && (b.fastK || b.fast2K)) {
    // the two base events are used in turn and re-recorded under the shared record's mutex: a batch that outlived two generations
    // (cap-prone or very long batches) finds its base re-recorded and reports no interval rather than a wrong one
    dg_index::SharedHints* shb = ix->shared_hints.load();
    std::unique_lock<std::mutex> lkb(shb->mu);
    float t0 = 0.f, t1 = 0.f;
    if (shb->base_gen - base_gen < 2 && hipEventElapsedTime(&t0, base_ev, ix->ev[1]) == hipSuccess && hipEventElapsedTime(&t1, base_ev, ix->ev[8]) == hipSuccess && t0 >= 0.f) {
      R->t_search_begin_ms = t0;
      R->t_search_end_ms = t1;
      R->t_base_gen = base_gen;
    }
  }
  if (dg_index::SharedHints* sh = ix->shared_hints.load()) {  // for the other lanes' next batches
    std::lock_guard<std::mutex> lk(sh->mu);
    sh->shard_cap = std::max(sh->shard_cap, ix->shard_cap_hint);
    sh->flat_cap = std::max(sh->flat_cap, ix->flat_cap_hint);
    sh->hit_cap = std::max(sh->hit_cap, ix->hit_cap_hint);
    sh->fetch_hits = ix->fetch_hits_hint;
    sh->generic_sticky = ix->generic_sticky;
    sh->jobs_sticky = ix->jobs_sticky;
    sh->jobs_big = ix->jobs_big_hint;
    sh->fused_leaves = ix->fused_leaves_hint;
    sh->valid = true;
  }
  if (host_timing)
    std::fprintf(stderr, "dicey timing: batch of %zu on stream %p entered at %.0f us: host %.0f us before the synchronisation (last attempt's "
                 "launches included), %.0f us waiting, %.0f us after; device %.3f ms\n", nq, (void*)st, std::fmod(t_enter, 1e8), t_launched - t_enter,
                 t_synced - t_launched, host_us() - t_synced, R->ms_total);
  return DG_OK;
}

}  // namespace dg

using namespace dg;

dg_switches dg_switches::read() {
  dg_switches w;
  auto num = [](const char* e) -> long { return e ? std::atol(e) : 0; };
  // deployment (every build)
  w.host_timing = num(std::getenv("DICEY_TIMING")) >= 2;
  if (const char* e = std::getenv("DICEY_CAP_BUDGET_MB")) w.cap_budget_mb = (uint64_t)std::max(1L, num(e));
  if (const char* e = std::getenv("DICEY_HOST_THREADS")) w.host_threads = (unsigned)std::max(1L, num(e));
  // development builds only (experiments.hpp: nullptr in the product library)
  w.no_band = exp_env("DICEY_NO_BAND_VERIFY") != nullptr;
  w.cap_host = exp_env("DICEY_CAP_HOST") != nullptr;
  w.no_fuse = exp_env("DICEY_NO_FUSED_SELECT") != nullptr;
  w.no_fuse2 = exp_env("DICEY_NO_FUSED_SELECT2") != nullptr;
  w.no_prep_fusion = exp_env("DICEY_NO_PREP_FUSION") != nullptr;
  w.no_pre5_d2 = exp_env("DICEY_NO_PRE5_D2") != nullptr;
  w.no_flat_ham2 = exp_env("DICEY_NO_FLAT_HAMMING2") != nullptr;
  w.no_nwin = exp_env("DICEY_NO_N_WINDOW") != nullptr;
  w.no_long2 = exp_env("DICEY_NO_LONG2") != nullptr;
  w.no_direct_ctx = exp_env("DICEY_NO_DIRECT_CTX") != nullptr;
  if (const char* e = exp_env("DICEY_DEBUG_CAPS")) {
    w.debug_caps = true;
    w.debug_caps_s = e;
  }
  if (const char* e = exp_env("DICEY_FUSED_LCAP")) w.fused_lcap = (uint32_t)std::max(1L, num(e));
  w.verify_ch = (int)num(exp_env("DICEY_VERIFY_CH"));
  if (const char* e = exp_env("DICEY_DUMP_JOBS")) w.dump_jobs = e;
  w.exp_bits = (uint32_t)num(exp_env("DICEY_EXP"));  // measurement aid (wrong results): phases of the search / verify kernels switched off
  return w;
}
static bool any_lane_busy(dg_index* ix) {
  if (ix->busy.load()) return true;
  std::lock_guard<std::mutex> lk(ix->lanes_mu);
  for (dg_index* l : ix->lanes)
    if (l && l->busy.load()) return true;
  return false;
}

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
  // The offsets are in host memory here: one pass over them (cheap beside the memcpy below) settles monotonicity and the
  // maximum, whether or not the caller named a bound — cap_scan and the staging copy below index with them.
  u32 maxlen = 0;
  for (size_t i = 0; i < nq; ++i) {
    if (qoff[i + 1] < qoff[i]) return fail(DG_EINVAL, "dg_hunt: qoff must be non-decreasing");
    u64 l = qoff[i + 1] - qoff[i];
    if (l > 0xFFFFFFu) return fail(DG_ELIMIT, "query %zu is too long", i);
    maxlen = std::max<u32>(maxlen, (u32)l);
  }
  if (p->max_query_len) {
    if (maxlen > p->max_query_len)
      return fail(DG_EINVAL, "dg_hunt: a query of %u nt exceeds dg_hunt_params::max_query_len = %u", maxlen, p->max_query_len);
    maxlen = p->max_query_len;  // the caller's bound sizes the batch (stable workspace sizes over a stream of batches)
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
  // a submitted batch owns its lane's stream, workspaces and hints until its ticket has been waited for; the blocking call runs on
  // the handle itself and is refused while ANY ticket of the handle is open (dicey_gpu.h)
  if (ix && any_lane_busy(ix)) return fail(DG_EINVAL, "dg_hunt: a dg_hunt_submit batch is in flight on this handle (dg_hunt_wait first)");
  if (ix) ix->sw = dg_switches::read();
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
  // dg_hunt_device_submit: the queries stay where the caller has them in HBM
  const void* d_qbytes = nullptr;
  const void* d_qoff = nullptr;
  uint64_t total_qbytes = 0;
  int fetch = 0;
  int rc = DG_OK;
  std::string err;
  dg_hunt_result* res = nullptr;
  bool done = false;
};
static int hunt_device(dg_index* ix, const dg_hunt_params* p, const uint32_t* seqlen, uint32_t nseq, const void* d_qbytes,
                       const void* d_qoff, size_t nq, uint64_t total_qbytes, int fetch, dg_hunt_result** out);
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
      const bool tm = t->ix->sw.host_timing;  // (read by the submitting thread)
      const double t_take = tm ? host_us() : 0.0;
      if (t->d_qoff)
        t->rc = hunt_device(t->ix, &t->p, t->seqlen.data(), (uint32_t)t->seqlen.size(), t->d_qbytes, t->d_qoff, t->nq, t->total_qbytes,
                            t->fetch, &t->res);
      else
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

// dg_hunt_submit / dg_hunt_device_submit: qoff != nullptr = host queries (copied into a pinned block here), else the device form
static int submit_batch(const char* who, dg_index* ix, const dg_hunt_params* p, const uint32_t* seqlen, uint32_t nseq, const uint8_t* qbytes,
                        const uint64_t* qoff, const void* d_qbytes, const void* d_qoff, uint64_t total_qbytes, int fetch, size_t nq,
                        dg_hunt_ticket** out) {
  const dg_switches sw_now = dg_switches::read();  // on the submitting thread; the lane's helper thread never touches the environment
  const bool tm = sw_now.host_timing;
  const double t_sub = tm ? host_us() : 0.0;
  // Lanes of a handle (ABI 5: two, r04: three): the handle itself while it is idle, else the first idle internal lane (a shared
  // handle: own stream, workspaces and helper thread on the same resident index; created at the first need).  A caller that keeps
  // one batch in flight never leaves the first lane.  Two lanes leave the search kernel idle a quarter of the time (both batches in
  // their launch-bound tails at once: profiles/r04c_k_search_launches.json); the third fills that.
  dg_index* const owner = ix;
  std::unique_lock<std::mutex> lanes_lk(owner->lanes_mu);  // lanes, their shared record and their helper threads are created lazily: one submitter at a time
  if (ix->busy.exchange(true)) {
    ix = nullptr;
    for (dg_index*& l : owner->lanes) {
      if (!l) {
        if (dg_index_share(owner, &l) != DG_OK) {
          l = nullptr;
          break;
        }
        dg_index::SharedHints* sh = owner->shared_hints.load();
        if (!sh) {
          sh = new dg_index::SharedHints;
          owner->shared_hints = sh;
        }
        l->shared_hints = sh;
      }
      if (!l->busy.exchange(true)) {
        ix = l;
        break;
      }
    }
    if (!ix)
      return fail(DG_EINVAL, "%s: %d batches are already in flight on this handle (wait for the oldest ticket first)", who, 1 + dg_index::NEXTRA);
  }
  ix->sw = sw_now;
  dg_hunt_ticket* t = nullptr;
  try {
    t = new dg_hunt_ticket;
    t->ix = ix;
    t->p = *p;
    t->seqlen.assign(seqlen, seqlen + nseq);
    t->nq = nq;
    if (qoff) {
      t->off_at = (qoff[nq] + 63) & ~63ull;
      t->stage = pinned_pool().get(t->off_at + (nq + 1) * 8 + 64);  // the caller's buffers are free again when this call returns
      if (!t->stage) {
        delete t;
        ix->busy.store(false);
        return fail(DG_ENOMEM, "%s: no pinned memory for %llu query bytes", who, (unsigned long long)qoff[nq]);
      }
      if (qoff[nq]) std::memcpy(t->stage->p, qbytes, qoff[nq]);
      std::memcpy((uint8_t*)t->stage->p + t->off_at, qoff, (nq + 1) * 8);
    } else {
      t->d_qbytes = d_qbytes;
      t->d_qoff = d_qoff;
      t->total_qbytes = total_qbytes;
      t->fetch = fetch;
    }
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
    lanes_lk.unlock();
  } catch (const std::exception& e) {
    if (t && t->stage) pinned_pool().put(t->stage);
    delete t;
    ix->busy.store(false);
    return fail(DG_ENOMEM, "%s: %s", who, e.what());
  }
  *out = t;
  if (tm) std::fprintf(stderr, "dicey timing: submit to %p from %.0f to %.0f us\n", (void*)ix, std::fmod(t_sub, 1e8), std::fmod(host_us(), 1e8));
  return DG_OK;
}

int dg_hunt_submit(dg_index* ix, const dg_hunt_params* p, const uint32_t* seqlen, uint32_t nseq, const uint8_t* qbytes,
                   const uint64_t* qoff, size_t nq, dg_hunt_ticket** out) {
  if (!ix || !p || !seqlen || !qoff || !out || (!qbytes && nq && qoff[nq])) return fail(DG_EINVAL, "dg_hunt_submit: null argument");
  *out = nullptr;
  if (!nq) return fail(DG_EINVAL, "dg_hunt_submit: empty batch");
  return submit_batch("dg_hunt_submit", ix, p, seqlen, nseq, qbytes, qoff, nullptr, nullptr, 0, 0, nq, out);
}

int dg_hunt_device_submit(dg_index* ix, const dg_hunt_params* p, const uint32_t* seqlen, uint32_t nseq, const void* d_qbytes,
                          const void* d_qoff, size_t nq, uint64_t total_qbytes, int fetch, dg_hunt_ticket** out) {
  if (!ix || !p || !seqlen || !d_qbytes || !d_qoff || !out) return fail(DG_EINVAL, "dg_hunt_device_submit: null argument");
  *out = nullptr;
  if (!nq) return fail(DG_EINVAL, "dg_hunt_device_submit: empty batch");
  return submit_batch("dg_hunt_device_submit", ix, p, seqlen, nseq, nullptr, nullptr, d_qbytes, d_qoff, total_qbytes, fetch, nq, out);
}

int dg_hunt_wait(dg_hunt_ticket* t, dg_hunt_result** out) {
  if (!t || !out) return fail(DG_EINVAL, "dg_hunt_wait: null argument");
  dg_index* ix = t->ix;
  const bool tm = ix->sw.host_timing;
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
    ix->sw = dg_switches::read();
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
  if (any_lane_busy(ix)) return fail(DG_EINVAL, "dg_hunt_device: a dg_hunt_submit batch is in flight on this handle (dg_hunt_wait first)");
  ix->sw = dg_switches::read();
  return hunt_device(ix, p, seqlen, nseq, d_qbytes, d_qoff, nq, total_qbytes, fetch, out);
}

// dg_hunt_device's body (also what a lane's helper thread runs for dg_hunt_device_submit)
static int hunt_device(dg_index* ix, const dg_hunt_params* p, const uint32_t* seqlen, uint32_t nseq, const void* d_qbytes,
                       const void* d_qoff, size_t nq, uint64_t total_qbytes, int fetch, dg_hunt_result** out) {
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
  if (rc == DG_EINVAL && cached) return hunt_device(ix, p, seqlen, nseq, d_qbytes, d_qoff, nq, total_qbytes, fetch, out);
  return rc;
}

}  // extern "C"

