// The object behind the opaque dg_index handle.
#pragma once
#include <mutex>
#include <string>
#include <atomic>
#include "common.hpp"
#include "devfm.hpp"
#include "experiments.hpp"

// Test / development switches of the hunt pipeline (environment variables).  Read ONCE per batch, on the thread that calls the
// library (the submitting thread for dg_hunt_submit batches) — never on the lanes' helper threads, where getenv would race with a
// host process that writes its environment — and carried to run_batch in the handle.
struct dg_switches {
  bool host_timing = false, no_band = false, cap_host = false, no_fuse = false, no_fuse2 = false, no_prep_fusion = false, no_pre5_d2 = false;
  bool debug_caps = false, no_flat_ham2 = false, no_nwin = false, no_long2 = false, no_direct_ctx = false;
  uint32_t fused_lcap = 0;       // DICEY_FUSED_LCAP (0 = unset)
  int verify_ch = 0;             // DICEY_VERIFY_CH
  uint64_t cap_budget_mb = 0;    // DICEY_CAP_BUDGET_MB (0 = unset)
  unsigned host_threads = 0;     // DICEY_HOST_THREADS (0 = unset)
  uint32_t exp_bits = 0;         // DICEY_EXP: measurement aid, phases of k_search1s switched off (results are then wrong)
  std::string dump_jobs, debug_caps_s;
  static dg_switches read();     // hunt.hip
};

struct dg_index {
  int device = 0;
  hipStream_t stream = nullptr;
  dg::FmView view;
  std::vector<void*> owned;  // device allocations that live as long as the index
  uint32_t sigma = 0;
  uint32_t code_len[256] = {0};
  uint64_t file_bytes = 0, hbm_bytes = 0;
  double load_seconds = 0, derive_seconds = 0;
  // grow-only batch workspaces (see hunt.hip / seam.hip for the slot meaning)
  static constexpr int NWS = 26;
  dg::DevBuf ws[NWS];
  hipEvent_t ev[9] = {nullptr};  // [8]: end of the flat distance-1 kernel
  uint32_t flat_cap_hint = 0;   // slice capacity of the flat Sel region that was enough so far (hunt.hip)
  uint32_t shard_cap_hint = 0;  // capacities that were enough for the previous batch (hunt.hip)
  uint64_t hit_cap_hint = 0;
  bool generic_hint = true;      // the previous distance-1 batch had work for the kernels outside k_search1s (hunt.hip run_batch)
  bool jobs_hint = true;         // the previous batch queued strings for the locate job kernels
  uint32_t walk_hint = 0;        // groups the previous batch listed for the walker (k_walk_list)
  uint32_t nwin_sticky = 0;      // batches the walker keeps its root split beside the flat distance-1 kernel (N-bearing strands in window mode)
  uint32_t short2_sticky = 0;    // distance 2: batches left on k_search2p's r04 body after one that held queries too short for LONG2
  uint32_t generic_sticky = 1, jobs_sticky = 1;  // batches the two hints stay on after the last batch that needed them
  uint64_t jobs_big_hint = 0;    // repeat-rich strings (workgroup locate jobs) of the previous batch
  uint64_t fused_leaves_hint = 0;  // occurring strings the previous distance-1 batch held in k_search1s' LDS lists (+ generic leaves)
  uint64_t fetch_hits_hint = 0;  // hits of the previous fetched batch (+3 %): this many are copied to the host before the batch's synchronisation
  // dg_hunt_device: the (offsets pointer, count, bytes) of the previous call and the longest query it held; a repeated
  // call skips reading the offsets back, and k_prepare reports any query longer than this bound (hunt.hip)
  // (r04: a small table, not one entry — a caller that cycles through a ring of resident batches finds each of them again)
  struct QoffSeen {
    const void* qoff = nullptr;
    uint64_t nq = 0, total = 0;
    uint32_t maxlen = 0;
  };
  static constexpr int NSEEN = 32;
  QoffSeen seen[NSEEN];
  uint32_t seen_next = 0;
  void* pinned = nullptr;             // 4 KB of pinned host memory for the end-of-batch summary
  // the batch counters are left zeroed by the last kernel of a batch (hunt.hip batch_finish): the next batch skips its memset
  // when they still sit where that kernel cleaned them
  std::atomic<bool> busy{false};      // a dg_hunt_submit batch is in flight on this handle (lane)
  dg_switches sw;                     // the switches of the batch this handle (lane) runs; set by the entry point on the caller's thread
  std::mutex lanes_mu;                // guards the lazy creation of lanes[] / shared_hints / a lane's worker (concurrent submitters)
  // ABI 5: dg_hunt_submit keeps several batches in flight on one handle; a submission that finds the handle busy runs on an internal
  // lane (a shared handle: own stream, workspaces, helper thread; created at the first need, closed with the handle)
#ifndef DG_NEXTRA
#define DG_NEXTRA 2
#endif
  static constexpr int NEXTRA = DG_NEXTRA;    // internal lanes beside the handle itself: 1 + NEXTRA batches in flight.  r05, same box (tools/r05_call13.sh): 3 in flight
                                              // 0.178-0.184 ms per step at distance 1, 4: 0.193-0.195, 5: 0.186, 6: 0.181; distance 2: 6.14-6.61 / 6.34-6.68 / - / 6.05-6.31
  dg_index* lanes[NEXTRA] = {};
  // The lanes learn together (r04): capacities and kernel-family hints live per lane (each lane's batches read and write its own
  // without locks), and a lane merges the pair's common record in when a batch starts and writes its own back when it ends — a
  // lane that runs its first batch does not repeat it for a capacity another lane has already learnt.
  struct SharedHints {
    std::mutex mu;
    uint32_t flat_cap = 0, shard_cap = 0, generic_sticky = 0, jobs_sticky = 0;
    uint64_t hit_cap = 0, jobs_big = 0, fused_leaves = 0, fetch_hits = 0;
    bool valid = false;
    // the lanes' common timeline (dg_hunt_result::t_search_*): two base events used in turn, a new one every few seconds
    hipEvent_t ev_base[2] = {nullptr, nullptr};
    uint32_t base_gen = 0;
    double base_host_us = 0;
  };
  // owned by the handle that owns the internal lanes, which point at the same record; atomic: created by the submitting thread while the first
  // lane's helper thread may be finishing a batch
  std::atomic<SharedHints*> shared_hints{nullptr};
  struct Worker;                      // the helper thread that drives dg_hunt_submit batches (hunt.hip)
  Worker* worker = nullptr;
  void stop_worker();
  const void* ctr_clean = nullptr;
  unsigned long long ctr_clean_gen = 0;
  std::vector<uint64_t> cum_cache;    // cumulative sequence starts currently resident in WS_CUM
  ~dg_index();
};

namespace dg {
// build.hip: sa = inverse permutation of isa by a radix sort of (isa[p], p); isa is clobbered
int derive_sa_by_sort(hipStream_t st, u32* isa, u64 n, u32* sa);
}
