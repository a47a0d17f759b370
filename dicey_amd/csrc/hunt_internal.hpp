// Shared between hunt.hip (the hunt pipeline, run_batch) and search.hip (the `dicey search` stage that replaces verify).
#pragma once
#include "devfm.hpp"
#include "index_internal.hpp"
#include "thal_internal.hpp"

namespace dg {

static constexpr u32 DMAX = 4;          // largest supported distance
static constexpr u32 MAX_QLEN = 255;    // longest supported query
enum : u32 { OP_S = 0, OP_I = 1, OP_D = 2 };


struct Batch {  // device pointers of one batch
  const u8* qbytes;
  const u64* qoff;
  u64 nq;
  u64 total_qbytes;  // bytes behind qbytes: k_prepare refuses offsets that decrease, leave this range or do not end at it
  u8* fw;    // codes 0..4 (A,C,G,T,N), same offsets as qbytes
  u8* rv;    // reverse complement
  u8* qseq;  // normalised ASCII
  u32* qlen;
  u32* qdist;
  u32* qflags;
  u32* qnondna;
  u32* qinfo;  // != nullptr: flags | distance << 8 | nondna << 16 per query, for compact results (written where the flags become final)
  u32 distance;
  u32 indel, reverse;
  u64 max_locations;
  u32 max_neighborhood;  // neighbors()' cap (hunter.h:334)
  u32* refused;          // counter of queries whose neighbourhood could reach the cap (see k_prepare)
  u32 maxlen_bound;      // the host sized the batch for queries up to this length ...
  u32* too_long;         // ... k_prepare counts the ones that are longer (only possible when the host trusted a cached bound)
  struct GidInfo* ginfo;  // [2*nq] what every lane of a (query, strand) needs, in one 16-byte record
  uint4* gpeq;            // [2*nq] position masks of the strand (x, y, z, w: bit i <=> character i is A, C, G, T), queries up to 32 nt
  u32 nrun_min;           // FmView::nrun_min (0 = unknown): k_prepare marks the strands whose N's can only be substituted or deleted (k_nres)
  u32 tabK;               // order of the K-mer table (0 = none): k_prepare marks N-bearing strands whose N's stay left of every window
  u32* nwin;              // ... and raises this flag when it marked one
  u32 fastK;              // != 0: the batch runs k_search1 (distance 1, table order fastK) for the queries that qualify
  u32 fast2K;             // != 0: the batch runs k_search2 (edit distance 2) for the queries that qualify
  u32 fast2_minlen;       // ... of at least this length (k_search2p<., LONG2 = true>: K2 + 2, K2 in Hamming mode; 0 otherwise)
  u32* short2;            // counter of the qualifying queries below it (they go to the walker)
  // Queries whose neighbourhood could reach the cap are enumerated on the host (nbhd_host.hpp) before the batch starts:
  // qmode[q] (nullptr = no such query in this batch): QM_KERNEL = not looked at (k_prepare's own bound must hold),
  // QM_EXPLICIT = the strings of both strands arrive as explicit patterns (xs_*) and k_search skips the query,
  // QM_SILENT = the host ran the reference's enumeration and the cap stayed silent on both strands, so k_search's set is
  // the reference's; | QM_FIRED = a strand reached the cap (hunter.h:342-345 warning).
  const u8* qmode;
  const u8* xs_bytes;  // explicit patterns: codes 0..4, pattern i = xs_bytes[xs_off[i] .. xs_off[i+1])
  const u64* xs_off;
  const u32* xs_gid;   // 2*query + strand of pattern i; patterns of a group are in std::set order
  u64 nxs;
};
enum : u8 { QM_KERNEL = 0, QM_EXPLICIT = 1, QM_SILENT = 2, QM_FIRED = 16 };
static constexpr u32 LEAF_EXPLICIT = 0x80000000u;  // Leaf::nops marker: ops[0] is an index into Batch::xs_*
struct GidInfo {
  u64 qpk;    // the sequence 2-bit packed, q[i] at bits 2(m-1-i) (only for m <= 32 without N)
  u32 m;      // length; 0 = this (query, strand) is not searched
  u32 d_win;  // bits 0-7 effective distance, bit 8: window mode allowed by the query (no N), bit 9: taken by k_search1, bit 10: taken by k_search2,
              // bit 11: N's left of every table window (walker in window mode), bit 12: as many N's as edits, all of them to be resolved (k_nres)
};

DG_DEV u32 ascii_rank(u32 code) { return code == 3 ? 4u : code == 4 ? 3u : code; }  // 'A'<'C'<'G'<'N'<'T'
DG_DEV u8 ascii_of(u32 code) { return code == 0 ? 'A' : code == 1 ? 'C' : code == 2 ? 'G' : code == 3 ? 'T' : 'N'; }



// Counters and the leaf allocator are sharded by workgroup: a single hot word serialises at ~11 ns per atomic, which
// is milliseconds once hundreds of thousands of wavefronts report.
static constexpr u32 NSHARD = 1024;
struct Counters {
  u32 overflow;          // set by k_leaf_overflow when a shard's region was too small: later kernels do nothing
  u32 pad_[15];
  u32 leaf_cnt[NSHARD];  // leaves allocated in each shard's region of the leaf buffer
  unsigned long long steps[NSHARD], lookups[NSHARD], sa_reads[NSHARD], win_bytes[NSHARD], probes[NSHARD];
  u32 surv_cnt[NSHARD];  // filter survivors queued in each shard's region of the survivor buffer (k_probe1 -> k_finish1)
  u32 sel_cnt[NSHARD];   // kept strings in each shard's slice of the flat Sel region (k_search1s)
  unsigned long long fused_leaves[NSHARD];  // occurring strings k_search1s settled in LDS
  // locate jobs queued by k_locate (hunt_locate.hpp): per list JOB_SHARDS producer counters — ONE word per list cost ~11 ns per
  // wavefront that queued a job (40 000 of them on a repeat-rich batch: most of that kernel's 0.26 ms, r05 counters)
  u32 job_cnt[3][64];
#ifdef DG_TOPK_PROFILE  // development builds only (tools/r06_prof.sh): phase clocks of the locate job kernels, 10 ns units
  unsigned long long prof[24];
#endif
};
#ifdef DG_TOPK_PROFILE
#define DG_LPROF_NOW() wall_clock64()
#define DG_LPROF_ADD(slot, dt)                                                        \
  do {                                                                               \
    if (threadIdx.x == 0) atomicAdd(&ctr->prof[slot], (unsigned long long)(dt));     \
  } while (0)
#else
#define DG_LPROF_NOW() 0ULL
#define DG_LPROF_ADD(slot, dt) \
  do {                         \
    (void)(slot);              \
    (void)(dt);                \
  } while (0)
#endif
static constexpr u32 JOB_SHARDS = 64;

struct HitSeed {  // 16 bytes, read as one uint4
  u32 pos;  // text position of the neighbourhood string
  u32 qs;
  u32 len;  // its length
  u32 sel;  // slot of the kept string (Sel) the hit stems from: hits of one slot share query, strand and string
};

enum WsSlot { WS_QB = 0, WS_QOFF, WS_FW, WS_RV, WS_QSEQ, WS_QMETA, WS_LEAF, WS_LEAFG, WS_SEL, WS_GRP, WS_MISC, WS_SEEDS, WS_HITS, WS_ALN, WS_CUM, WS_SCR, WS_JOBS, WS_DP, WS_PRIM, WS_GINFO, WS_XS, WS_OPS, WS_PACK, WS_CAP, WS_WALK, WS_SELKEY };

// one located hit after the `dicey search` stage
struct SiteRaw {
  double temp;   // o.temp of thal(); -999999 when thal() refuses (both sequences > 60 nt)
  u32 ref, chrpos, alignpos, glen;
  u32 qs;        // 2*primer + strand
  u32 pad;
};

struct SearchExtra {  // set by dg_search_sites: replace the verify stage by k_site
  const dg_thal* th;
  const u8* d_pfw;
  const u8* d_prv;
  const u64* d_poff;
  const u32* d_koff;
  double cut_temp;
  u32 max_primer_len, max_koff;
  // results (device buffers of the index workspaces, valid until the next call)
  const SiteRaw* d_sites = nullptr;
  const u8* d_windows = nullptr;
  u32 win_stride = 0;
  const u64* d_hit_off = nullptr;
  const u32* d_qflags = nullptr;
};


// hunt.hip
int run_batch(dg_index* ix, const dg_hunt_params* p, const uint32_t* seqlen, uint32_t nseq, const void* d_qbytes, const void* d_qoff,
              size_t nq, u64 total, u32 maxlen, int fetch, dg_hunt_result** out, SearchExtra* sx = nullptr,
              uint64_t* group_counts = nullptr, const uint8_t* h_qbytes = nullptr, const uint64_t* h_qoff = nullptr);
// hunt.hip: exclusive prefix sums of n 32-bit counts into 64-bit offsets (out[n] = total); tmp: n/64 + 64 words
int device_scan(hipStream_t st, const u32* in, u64 n, u64* out, u64* tmp, Counters* ctr = nullptr, u32 shard_cap = 0, u32 surv_cap = 0);
// search.hip: launches k_site over the located hits (capacity hit_cap) and fills sx's result pointers
int launch_site_stage(dg_index* ix, SearchExtra* sx, const Batch& b, const HitSeed* seeds, const u64* hit_off, u64 hit_cap,
                      const u64* cum, u32 nseq, u32 dmax_eff, u32 maxlen, Counters* ctr);

}  // namespace dg
