/* dicey_gpu.h — C ABI of libdiceygpu.so: the MI355X (gfx950) in-silico-PCR search path.
 *
 * gear-genomics/dicey has no plugin/FFI interface.  The seam this library replaces is the set of
 * calls `dicey hunt` makes into sdsl-lite and into its own helper headers:
 *
 *   reference call (file:line)                              entry point here
 *   ------------------------------------------------------  ---------------------------------
 *   load_from_checked_file(csa_wt<>&, path)  hunter.h:253-256, silica.h:340-343   dg_index_open
 *   fm_index.size()                          hunter.h:368-369                     dg_index_stats
 *   sdsl::count(fm_index, b, e)              hunter.h:353, silica.h:470           dg_count
 *   sdsl::locate(...) + std::sort            hunter.h:355-356, silica.h:472-473   dg_locate
 *   sdsl::extract(fm_index, lo, hi)          hunter.h:371, silica.h:490           dg_extract
 *   neighbors() -> count -> locate -> extract -> needle()/needleScore -> DnaHit push
 *                                            hunter.h:291-437 (whole per-query loop)  dg_hunt
 *   sdsl::construct + store_to_checked_file  index.h:121-122                      dg_index_build
 *   neighbors(q, alphabet, d, indel, maxsize, set)  neighbors.h:86-92 (hunter.h:334,339)  dg_neighbors
 *
 * Conventions: plain C types only; every function returns 0 on success or a negative DG_E* code and
 * leaves a message retrievable with dg_last_error() (thread-local); objects returned through `out`
 * parameters are released with the paired *_free / *_close; no exceptions cross the boundary; one
 * HIP stream per dg_index; a dg_index may be used by one thread at a time (distinct handles are
 * independent — one handle per GPU for multi-GPU).  There is NO CPU fallback: without a HIP device
 * every compute entry point fails with DG_ENODEV.
 */
#ifndef DICEY_GPU_H
#define DICEY_GPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 1: hunt/count/locate/extract/index build; 2: + thal, search sites, neighbourhood counts, padlock scan, shared handles;
 * 3: + dg_neighbors, capped neighbourhoods answered instead of refused, max_locations 0
 * 4: hits carry their alignment in compact form (dg_hunt_result::ops; dg_hunt_rows / dg_hit_rows rebuild the two rows),
 *    result buffers come from a pinned pool, dg_hunt_submit / dg_hunt_wait
 * 5: dg_hunt_params grows by max_query_len and flags; DG_HUNT_COMPACT: 8 + 4 d bytes per hit and 8 bytes per query cross PCIe / xGMI
 *    (dg_chit_unpack, dg_hunt_expand, dg_normalize_query turn them back); dg_hunt_submit keeps up to three batches in flight on ONE handle
 * 6: dg_hunt_result grows by stream / d_block / d_block_bytes (the gather over RCCL lives in libdiceygather.so, include/dicey_gather.h) */
#define DG_ABI_VERSION 7

enum {
  DG_OK = 0,
  DG_EINVAL = -1,   /* bad argument */
  DG_EIO = -2,      /* file cannot be read / written */
  DG_EFORMAT = -3,  /* not an sdsl csa_wt<> file (byte accounting failed) */
  DG_ENODEV = -4,   /* no usable HIP device */
  DG_EHIP = -5,     /* HIP runtime error */
  DG_ENOMEM = -6,
  DG_ELIMIT = -7    /* input outside the supported envelope (see DESIGN.md "limits") */
};

typedef struct dg_index dg_index;

/* flags for dg_index_open */
#define DG_OPEN_DEFAULT 0u
#define DG_OPEN_NO_SELFCHECK 1u /* skip the load-time self validation (C[] vs Occ totals, SA permutation spot checks) */
#define DG_OPEN_NO_KMER_TABLE 2u /* do not derive the K-mer jump table (saves up to 34 GB of HBM; search is slower) */
#define DG_OPEN_COMPACT 4u       /* ABI 7: a process that opens the index for ONE input — skip the layouts that only pay for a resident index: the suffix
                                  * array with context records and its prefix levels (40 GB of HBM on a 3.1 Gb genome, 0.36 s to derive, and as much again for
                                  * the driver to wipe at exit); hits of repeat-rich strings then read their context from the text and walk the block minima,
                                  * results are the same.  (ABI 4-6: accepted, no effect.)  `dicey hunt|search|padlock` pass it. */
#define DG_OPEN_NO_PRE5 16u      /* ABI 7: do not derive the preceding-characters array (6.2 GB, 0.19 s on a 3.1 Gb genome): narrow table intervals are then
                                  * extended character by character through the Occ blocks — the distance-1 search kernel takes a quarter longer per
                                  * batch (0.146 -> 0.183 ms per 100 000 20-mers; distance 2: + 5 %), which a process that answers a few million
                                  * queries and exits never earns back.  Same results.  `dicey hunt` passes it unless the input is very large. */
#define DG_OPEN_BIG_TABLE 8u     /* table of order ceil(log4 n) + 1 when the device has room (137 GB instead of 34 GB on a 3.1 Gb genome):
                                    the distance-1 search kernel gains ~5 % (0.175 -> 0.166 ms per 100 000 20-mers), the open takes longer
                                    and the process holds 199 GB instead of 90 GB — for resident servers with HBM to spare */

/* Parses the file written by `dicey index` (sdsl store_to_checked_file of csa_wt<>) unchanged, uploads it to
 * HBM on `device`, and derives the search layouts there (Occ blocks, full suffix array, text copy). */
int dg_index_open(const char* fm9_path, int device, uint32_t flags, dg_index** out);
void dg_index_close(dg_index* ix);
/* A second handle on the SAME device-resident index with its own HIP stream and batch workspaces, so that another host
 * thread can run batches concurrently (the small tail kernels of one batch overlap with the search kernel of the other).
 * No index data is copied.  Close every shared handle before the handle it was taken from. */
int dg_index_share(dg_index* src, dg_index** out);
/* The handle's HIP stream (a hipStream_t).  Work a caller enqueues on it is ordered with the handle's batches: a device-side copy
 * out of dg_hunt_result::d_hits issued on this stream is complete before the next batch overwrites the buffer, with no host
 * synchronisation (bench.py stages its RCCL gather this way). */
void* dg_index_stream(dg_index* ix);

typedef struct {
  uint64_t n;              /* fm_index.size(): text length + 1 (sentinel) */
  uint32_t sigma;          /* alphabet size incl. sentinel */
  uint32_t code_len[256];  /* Huffman code length of each byte in the loaded wavelet tree (0 = absent) */
  uint64_t file_bytes;     /* size of the .fm9 */
  uint64_t hbm_bytes;      /* device memory held by this index (raw sections + derived layouts) */
  double load_seconds;     /* file read + upload + derivation */
  double derive_seconds;   /* derivation kernels only */
} dg_index_stats_t;
int dg_index_stats(const dg_index* ix, dg_index_stats_t* out);

/* ---- sdsl seam, batched.  Patterns are raw bytes, concatenated; pattern i = pat[off[i] .. off[i+1]). ---- */
int dg_count(dg_index* ix, const uint8_t* pat, const uint64_t* off, size_t npat, uint64_t* counts /* [npat] */);

typedef struct {
  size_t npat;
  uint64_t* off; /* [npat+1] */
  uint64_t* pos; /* positions of pattern i, ascending: pos[off[i] .. off[i+1]) */
} dg_locations;
int dg_locate(dg_index* ix, const uint8_t* pat, const uint64_t* off, size_t npat, dg_locations** out);
void dg_locations_free(dg_locations* l);

/* text[lo[i] .. hi[i]] inclusive (hi < n), written to out + out_off[i] */
int dg_extract(dg_index* ix, const uint64_t* lo, const uint64_t* hi, size_t nrange, uint8_t* out, const uint64_t* out_off);

/* ---- hunt ---- */
typedef struct {
  uint32_t distance;         /* -d  (hunter.h:188, default 1) */
  int32_t hamming;           /* -n  (hunter.h:189): substitutions only, no context, no alignment */
  int32_t forward_only;      /* -f  (hunter.h:190) */
  uint64_t max_locations;    /* -m  (hunter.h:186, default 1000) */
  uint32_t max_neighborhood; /* -x  (hunter.h:187, default 10000) */
  uint32_t max_query_len;    /* ABI 5: an upper bound of the batch's query lengths when the caller knows one (a primer design tool does);
                              * 0 = the library finds out (the host entry points scan their offsets, dg_hunt_device reads its offsets
                              * back: a host round trip per new buffer).  The kernels count queries above the bound and the call fails
                              * with DG_EINVAL rather than answer them wrongly. */
  uint32_t flags;            /* ABI 5: DG_HUNT_* */
} dg_hunt_params;
#define DG_HUNT_COMPACT 1u   /* results in compact form (below): what the host needs to rebuild every DnaHit, nothing it already has */
#define DG_HUNT_PHASE_TIMES 2u /* measure ms_select / ms_locate / ms_verify too (HIP events between the stages: a few microseconds of
                                * stream markers per batch); ms_total and ms_search_flat are always measured */

/* per-query flag bits */
#define DG_Q_TOO_SHORT 1u     /* < 10 nt: "Error: Input sequence is shorter than 10 nucleotides!" (hunter.h:299-303) */
#define DG_Q_DIST_ADJUSTED 2u /* hunter.h:312-315 */
#define DG_Q_MAX_MATCHES 4u   /* hits >= max_locations (hunter.h:434-437) */
#define DG_Q_NBHD_EXCEEDED 8u /* hunter.h:342-345: a strand's neighbourhood reached max_neighborhood; the strings searched are
                               * the ones the reference's capped enumeration holds (see dg_neighbors) */

typedef struct {
  int32_t score;     /* DnaHit::score (0, -1, ...) */
  uint32_t chr;      /* refIndex */
  uint32_t start;    /* 1-based (chrpos+1, hunter.h:402) */
  uint32_t query;    /* index into the batch */
  uint16_t aln_len;  /* columns in refalign/queryalign */
  uint8_t strand;    /* '+' or '-' */
  uint8_t reserved;
} dg_hit;

/* Compact alignment (ABI 4).  The rows the reference pushes into DnaHit (hunter.h:391-405, 411-426) are the characters of the
 * query strand the hit was found on, with at most |score| <= distance columns that are not a match: leading and trailing
 * columns whose query row is a gap are stripped, and every other non-matching column costs one.  A hit therefore travels as
 * dg_hit + `ops_per_hit` 32-bit words, one per non-matching column in column order, DG_ALN_NONE behind the last:
 *   bits 0-15 column, bits 16-17 kind, bits 24-31 the reference byte of that column (kinds MISMATCH and QUERY_GAP)
 * 24 bytes per hit at distance 1 instead of 68 with two character rows — what crosses PCIe and xGMI.  dg_hit_rows() /
 * dg_hunt_rows() rebuild refalign / queryalign byte for byte. */
#define DG_ALN_MISMATCH 0u  /* reference byte over a different query character */
#define DG_ALN_REF_GAP 1u   /* '-' in refalign over a query character */
#define DG_ALN_QUERY_GAP 2u /* reference byte over '-' in queryalign */
#define DG_ALN_NONE 0xFFFFFFFFu
#define DG_ALN_COL(op) ((op) & 0xFFFFu)
#define DG_ALN_KIND(op) (((op) >> 16) & 3u)
#define DG_ALN_BYTE(op) ((op) >> 24)

/* Compact results (ABI 5, DG_HUNT_COMPACT).  Per hit 2 + ops_per_hit words, in push order, the hits of query i at
 * chits + hit_off[i] * (2 + ops_per_hit):
 *   word 0   text position of the neighbourhood string the hit stems from (what sdsl::locate returned, hunter.h:355): the
 *            sequence is the one that holds this position (hunter.h:358-362, seq_start[] below), chrpos = position - its start
 *   word 1   bits 0-3 -score, bit 4 strand (1 = '-'), bits 5-11 delta + 32 where DnaHit::start = chrpos + delta + 1 (the
 *            context characters in front of the string, hunter.h:382 with its strict '<', and the leading gap columns of
 *            hunter.h:391-401 are in delta), bits 16-31 aln_len
 *   words 2.. the alignment description of ABI 4 (DG_ALN_*)
 * Per query one word qinfo: bits 0-7 DG_Q_* flags, bits 8-15 effective distance, bits 16-31 characters replaced by 'N'.
 * The normalised queries are NOT returned: dg_normalize_query applies util.h:208-219 + to_upper to the caller's own bytes.
 * 12 bytes per hit at distance 1 (ABI 4: 24), 8 bytes per query (ABI 4: 12 + the sequence + 8 of offsets). */
#define DG_CHIT_WORDS(ops_per_hit) (2u + (ops_per_hit))
#define DG_CHIT_NEG_SCORE(meta) ((meta) & 15u)
#define DG_CHIT_STRAND(meta) ((((meta) >> 4) & 1u) ? '-' : '+')
#define DG_CHIT_DELTA(meta) ((int32_t)(((meta) >> 5) & 127u) - 32)
#define DG_CHIT_ALN_LEN(meta) ((meta) >> 16)
#define DG_QINFO_FLAGS(w) ((w) & 255u)
#define DG_QINFO_DISTANCE(w) (((w) >> 8) & 255u)
#define DG_QINFO_NONDNA(w) ((w) >> 16)

typedef struct {
  size_t nq;
  uint64_t nhits;
  uint64_t* hit_off;      /* [nq+1]: hits of query i = hits[hit_off[i] .. hit_off[i+1]) in REFERENCE PUSH ORDER (pre-sort) */
  dg_hit* hits;           /* [nhits] */
  uint32_t ops_per_hit;   /* words of alignment description per hit: the batch's largest effective distance (0: every row is the query) */
  uint32_t aln_stride;    /* bytes per row in refalign / queryalign (0 until dg_hunt_rows has run) */
  uint32_t* ops;          /* [nhits * ops_per_hit] */
  char* refalign;         /* NULL until dg_hunt_rows(): row of hit h = refalign + h*aln_stride, aln_len bytes */
  char* queryalign;       /* likewise */
  uint32_t* qflags;       /* [nq] DG_Q_* */
  uint32_t* qdistance;    /* [nq] effective (clamped) distance */
  uint32_t* qnondna;      /* [nq] number of characters replaced by 'N' (one warning each, util.h:214) */
  uint8_t* qseq;          /* normalised (upper-cased, non-ACGT -> N) queries, concatenated like the input */
  uint64_t* qoff;         /* [nq+1] */
  /* measurement: exact op counters of the executed device algorithm (DESIGN.md "algorithmic bytes") */
  uint64_t ctr_ext_steps; /* interval extensions (2 Occ-block reads each) */
  uint64_t ctr_leaves;    /* occurring neighbourhood strings emitted by the search kernel */
  uint64_t ctr_sa_reads;  /* suffix-array entries (and block minima) read by locate */
  uint64_t ctr_win_bytes; /* text window bytes the reference would extract (hunter.h:371), one window per hit */
  uint64_t ctr_tab_reads; /* K-mer jump-table entries read (8 B each) */
  double ms_total;        /* device time of the whole batch (HIP events on the index stream) */
  double ms_search;       /* of which: neighbourhood/backward-search kernel */
  double ms_select;       /* minimal-set + ordering kernel */
  double ms_locate;
  double ms_verify;       /* window fetch + Needleman-Wunsch */
  /* device-resident copies (HIP pointers on the index's device), valid until the next call on this index:
   * what a multi-GPU driver hands to RCCL to gather hit lists without a host round trip */
  const void* d_hits;     /* dg_hit[nhits] */
  const void* d_ops;      /* uint32_t[nhits * ops_per_hit] */
  uint64_t ctr_filter_probes; /* K-mer presence-filter bits tested (one 4-byte word each); ctr_tab_reads counts the table
                               * entries actually read, i.e. the probes that found their K-mer present */
  double ms_search_flat;      /* part of ms_search spent in the flat kernel (k_search1p at distance 1, k_search2p at edit distance 2); 0 when none ran */
  void* owner_;               /* library internal (pinned-pool bookkeeping) */
  /* ABI 5.  compact != 0: chits / qinfo / seq_start are set, hit_off as always; hits, ops, qflags, qdistance, qnondna, qseq, qoff are
   * NULL until dg_hunt_expand() builds them on the host; d_hits then points at the compact records in HBM and d_ops is NULL. */
  uint32_t compact;
  uint32_t nseq;
  uint32_t* chits;            /* [nhits * DG_CHIT_WORDS(ops_per_hit)] */
  uint32_t* qinfo;            /* [nq] */
  uint64_t* seq_start;        /* [nseq] text position of every sequence's first character (prefix sums of seqlen) */
  void* expanded_;            /* library internal (dg_hunt_expand's allocations) */
  /* measurement: the capped-neighbourhood stage in front of the batch (hunt_cap.hpp / nbhd_host.hpp), host wall clock, and what it did */
  double ms_cap;              /* 0 when no query of the batch could reach the cap */
  uint64_t cap_queries_device, cap_queries_host, cap_patterns; /* queries enumerated on the device / on the host, explicit patterns searched */
  /* Where the batch's flat search kernel ran on the handle's timeline (r04): begin / end in ms since a base event the handle's lanes
   * share.  With several batches in flight the launches of neighbouring batches overlap, and the time the kernel RAN is the union of these
   * intervals, not the sum of their lengths (bench.py's roofline).  t_base_gen: intervals of equal generation share a base (the
   * library takes a new base every few seconds to keep float precision); 0 = no timeline (one lane only so far, or no flat kernel). */
  double t_search_begin_ms, t_search_end_ms;
  uint32_t t_base_gen;
  uint32_t flat_kernel_form; /* measurement: the flat search kernel this batch ran — 0 none (general k_search only), 1 k_search1p, 2 k_search1s<.,
                              * false> (select stage inside), 3 k_search1s<., true> (select and take inside), 4 k_search2p<false, false>, 5 k_search2p<true, false>,
                              * 6 k_search2p<false, true>, 7 k_search2p<true, true> (second template argument: the r05 body for batches
                              * whose shortest strings still ask the long filter) */
  /* ABI 6: what a gather over RCCL needs (include/dicey_gather.h).  stream: the hipStream_t the batch ran on — the handle's own, or
   * the internal lane's for a dg_hunt_submit / dg_hunt_device_submit batch; a device-side copy out of the result's device buffers
   * queued on it is ordered before that lane's next batch, with no host synchronisation.  Compact results: d_block is the batch's
   * whole answer in HBM as ONE block [hit counts u32[nq] | qinfo u32[nq] | records (nhits * DG_CHIT_WORDS words)] = d_block_bytes
   * bytes (d_hits points at its records; hit_off is the prefix sum of the counts); after a fetch the host has the same bytes at
   * (uint8_t*)qinfo - 4 nq.  NULL / 0 for a classic result. */
  void* stream;
  const void* d_block;
  uint64_t d_block_bytes;
  /* ABI 7, measurement: which verify kernel served the batch — band width << 8 | hits per lane of k_verify_memo (7 << 8 | 4 is
   * k_verify_memo<7, 4>), 0 for the full-matrix / long-query kernels and for `search` (k_site); and how many of the batch's locate
   * jobs (strings with more than 16 occurrences) took a run of a prefix level's records instead of a walk down the block minima */
  uint32_t verify_kernel_form;
  uint32_t reserved7;
} dg_hunt_result;

/* One compact hit as a dg_hit (query = the query it belongs to, from hit_off) and a pointer to its ops words. */
int dg_chit_unpack(const dg_hunt_result* r, uint64_t h, uint32_t query, dg_hit* out, const uint32_t** ops);
/* hunter.h:306 + util.h:208-219: upper-case, every character outside A,C,G,T becomes 'N'; *nondna = how many were replaced
 * (a literal 'N' counts: the reference warns once per replaced character). */
int dg_normalize_query(const uint8_t* in, uint32_t len, uint8_t* out, uint32_t* nondna);
/* Builds the ABI-4 arrays (hits, ops, qflags, qdistance, qnondna, qseq, qoff) of a compact result on the host, from the caller's
 * own query bytes (the ones the batch was submitted with); several host threads.  Afterwards dg_hunt_rows / dg_hit_rows work as
 * on a classic result.  A no-op on a classic result. */
int dg_hunt_expand(dg_hunt_result* r, const uint8_t* qbytes, const uint64_t* qoff);

/* The two alignment rows of one hit from its compact description.  qseq / qlen: the NORMALISED query (dg_hunt_result::qseq:
 * A,C,G,T,N), forward strand — the reverse complement a '-' hit was aligned to is formed here (util.h:54-114).  ops: the hit's
 * ops_per_hit words (NULL when ops_per_hit == 0).  Writes hit->aln_len bytes to each row (no terminator).  DG_EINVAL when the
 * description does not fit the query (a corrupted record). */
int dg_hit_rows(const dg_hit* hit, const uint32_t* ops, uint32_t ops_per_hit, const uint8_t* qseq, uint32_t qlen, char* refalign,
                char* queryalign);
/* Fills r->refalign / r->queryalign / r->aln_stride for every hit of a fetched result (several host threads). */
int dg_hunt_rows(dg_hunt_result* r);

/* Host-buffer entry point: queries are raw bytes as read from the FASTA/argv (any case), concatenated.
 * seqlen[i] = faidx length of sequence i + 1 (util.h:201), nseq sequences. */
int dg_hunt(dg_index* ix, const dg_hunt_params* p, const uint32_t* seqlen, uint32_t nseq, const uint8_t* qbytes,
            const uint64_t* qoff, size_t nq, dg_hunt_result** out);
void dg_hunt_result_free(dg_hunt_result* r);

/* Asynchronous form: dg_hunt_submit returns as soon as the batch is handed to the library (the query buffers are copied, the
 * caller may reuse them at once); dg_hunt_wait blocks until the result is on the host (a pinned block, like dg_hunt's) and releases
 * the ticket.  ABI 5: up to THREE batches may be in flight on one handle (the library runs them on internal lanes — own stream,
 * own workspaces, own helper thread each, the same resident index — so uploads, kernels and downloads of neighbouring batches
 * overlap): submit A, submit B, wait A, submit C, wait B, ...  Tickets are waited for in the order they were submitted; a fourth
 * submit before the first wait fails with DG_EINVAL.  Every ticket must be waited for before its handle is closed. */
typedef struct dg_hunt_ticket dg_hunt_ticket;
int dg_hunt_submit(dg_index* ix, const dg_hunt_params* p, const uint32_t* seqlen, uint32_t nseq, const uint8_t* qbytes,
                   const uint64_t* qoff, size_t nq, dg_hunt_ticket** out);
int dg_hunt_wait(dg_hunt_ticket* t, dg_hunt_result** out);

/* Device-resident entry point used by bench.py: d_qbytes / d_qoff are HIP device pointers on the index's device.
 * The result stays in HBM; only the counters/timings and nhits are copied back.  `fetch` != 0 additionally
 * copies hits to host like dg_hunt.  The offsets are read back once to size the batch; a repeated call with the same
 * d_qoff, nq and total_qbytes reuses that bound, and the kernels report any query that exceeds it (the call then reads
 * the offsets again by itself), so the buffers may be refilled in place between calls. */
int dg_hunt_device(dg_index* ix, const dg_hunt_params* p, const uint32_t* seqlen, uint32_t nseq, const void* d_qbytes,
                   const void* d_qoff, size_t nq, uint64_t total_qbytes, int fetch, dg_hunt_result** out);
/* Asynchronous form of dg_hunt_device (r04), on the same lanes as dg_hunt_submit and collected with dg_hunt_wait: a caller whose
 * batches are resident in HBM keeps two or three in flight — submit A, submit B, wait A, submit C, wait B, ... — so that one batch's
 * launch-bound tail (locate, verify, the summary's read-back) runs beside another batch's search kernel.  The device buffers must
 * stay untouched until the ticket has been waited for; a result left in HBM (fetch = 0) is valid until the next submit after its
 * own wait (the lane that ran it is idle until then).  Replaces nothing in the reference: hunter.h:291 walks its queries one by one. */
int dg_hunt_device_submit(dg_index* ix, const dg_hunt_params* p, const uint32_t* seqlen, uint32_t nseq, const void* d_qbytes,
                          const void* d_qoff, size_t nq, uint64_t total_qbytes, int fetch, dg_hunt_ticket** out);

/* ---- `dicey padlock`: how often the neighbourhood of a probe arm occurs (reference src/padlock.h:392-421) ----
 * For sequence i (>= 10 nt; A/C/G/T sequences run on the search kernel, sequences with other letters are enumerated on
 * the host and counted with dg_count): fw_count[i] = sum over s in neighbors(seq_i, distance, indel, maxsize) of
 * sdsl::count(fm_index, s) and rv_count[i] the same for its reverse complement — the totals the reference accumulates in
 * hits[0] / hits[1] (its loops only stop early once the running total already exceeds the threshold it is compared with,
 * so every comparison it makes has the same outcome on the full totals).  The original sequence is part of its own
 * neighbourhood.  max_neighborhood is neighbors()' cap (10000 in padlock.h:396); when it can fire the reference's capped
 * enumeration is reproduced on the host and its strings are counted (DESIGN.md "the cap").  Sequences above 255 nt are
 * refused here with DG_ELIMIT (dg_hunt takes them). */
int dg_neighborhood_count(dg_index* ix, uint32_t distance, int hamming, uint32_t max_neighborhood, const uint8_t* qbytes,
                           const uint64_t* qoff, size_t nq, uint64_t* fw_count, uint64_t* rv_count);

/* ---- neighbors() with its size cap (reference src/neighbors.h:29-92), host side ----
 * The strings `dicey hunt` / `search` / `padlock` search for one sequence and one strand: seq is the normalised sequence
 * (A,C,G,T,N), alphabet {A,C,G,T}.  Edit mode keeps the substring-minimal strings; when the working set reaches
 * max_neighborhood the reference stops generating (neighbors.h:50) and what it holds then depends on its generation order,
 * which this function follows.  *out = the strings in std::set order, '\n'-terminated each, NUL at the end (release with
 * dg_buffer_free); *count their number; *cap_fired != 0 <=> count >= max_neighborhood (hunter.h:342-345 warns).
 * dg_hunt, dg_search_sites and dg_neighborhood_count call this for the sequences whose neighbourhood could reach the cap
 * and search exactly these strings; all other sequences are enumerated inside the search kernel.  Needs no device. */
int dg_neighbors(const uint8_t* seq, uint32_t len, uint32_t distance, int hamming, uint32_t max_neighborhood, char** out,
                 uint64_t* count, int* cap_fired);
void dg_buffer_free(void* p);

/* ---- index construction on the GPU (what `dicey index` does with sdsl::construct, index.h:97-123) ---- */
/* text = SEQ1 '\n' SEQ2 '\n' ... SEQk '\n' (upper-cased), no NUL inside.  Writes sdsl csa_wt<> layout. */
int dg_index_build(const uint8_t* text, uint64_t len, int device, const char* out_fm9_path);
/* same, text already in HBM (bench: synthetic genome generated on device) */
int dg_index_build_device(const void* d_text, uint64_t len, int device, const char* out_fm9_path);

/* ---- thermodynamic alignment for `dicey search` (primer3 thal(), type END1, temponly) ----
 * replaces primer3thal::get_thermodynamic_values + set_thal_default_args (silica.h:316-329) and primer3thal::thal()
 * (silica.h:437, 511; thal.h:2409-2655).  config_dir holds primer3's stack.ds ... tstack2.dh tables. */
typedef struct dg_thal dg_thal;
int dg_thal_open(const char* config_dir, double mv, double dv, double dntp, double dna_conc, int device, dg_thal** out);
void dg_thal_close(dg_thal* th);
/* pair k = (oligo1, oligo2) = seqs[off[2k]..off[2k+1]), seqs[off[2k+1]..off[2k+2]); temp[k] = o.temp (-999999 = THAL_ERROR_SCORE
 * when both sequences exceed 60 nt), end1/end2 = align_end_1/2 (may be NULL) */
int dg_thal_batch(dg_thal* th, const uint8_t* seqs, const uint64_t* off, size_t npairs, double* temp, int32_t* end1, int32_t* end2);

/* ---- `dicey padlock`: the per-position values of a batch of exons (reference src/padlock.h:321-428) ----
 * exons: strand-corrected, upper-case exon sequences (exonseq of padlock.h:313-315), concatenated; exon e =
 * exons[exon_off[e] .. exon_off[e+1]).  Exons shorter than 2*armlen carry no positions.  For exon e, arm window q
 * (0 <= q <= len-armlen) has slot pos_off[e] + q:
 *   arm_gc      gccontent(arm)                          (padlock.h:324, 342; -1 when the window holds an N)
 *   arm_tm      thal(arm, reverse complement).temp      (:331-336; DG_PADLOCK_NOT_COMPUTED when arm_gc fails the filter)
 *   probe_gc / probe_tm  the same for the 2*armlen probe starting at q (:356-371; q <= len-2*armlen; probe_tm is computed
 *                only when both arms pass GC, the Tm ceiling 93 + GC - 675/armlen and the Tm difference)
 *   arm_count   sdsl::count(arm) + sdsl::count(reverse complement)   (:381-385)
 *   arm_nbcount occurrences summed over neighbors(arm) and neighbors(reverse complement) (:396-405; see
 *                dg_neighborhood_count); both -1 unless the arm belongs to a probe inside the Tm window (:372-374)
 * The caller replays the reference's decisions (including `k += targetlen - 1` after an accepted probe) on these arrays;
 * a decision that would need a value not computed here cannot be reached.  thal refusing a pair (both oligos > 60 nt)
 * shows as -999999, the reference's error path (:337-340). */
#define DG_PADLOCK_NOT_COMPUTED (-1e300)
typedef struct {
  uint32_t armlen;    /* -m */
  uint32_t distance;  /* -d */
  int32_t hamming;    /* -n */
  uint32_t tmdiff;    /* -z */
  double gc_min, gc_max; /* --gcmin / --gcmax */
} dg_padlock_params;
typedef struct {
  uint64_t nexons, npos;
  uint64_t* pos_off; /* nexons+1 */
  double *arm_gc, *arm_tm, *probe_gc, *probe_tm;
  int64_t *arm_count, *arm_nbcount;
  uint64_t n_arm_thal, n_probe_thal, n_arms_counted; /* work done, for measurements */
} dg_padlock_result;
int dg_padlock_scan(dg_index* ix, dg_thal* th, const dg_padlock_params* p, const uint8_t* exons, const uint64_t* exon_off, size_t nexons,
                    dg_padlock_result** out);
void dg_padlock_result_free(dg_padlock_result* r);

/* ---- `dicey search`: binding sites of a batch of primers (reference src/silica.h:429-573) ----
 * Per primer: thal(primer, revcomp) (silica.h:437); neighbourhood of its last k nucleotides on both strands through the
 * FM-index; per located hit the context window with the 5' overhang, thal(primer, window), the Tm cut, the alignment
 * position that de-duplicates hits, and the trimmed genomic site (silica.h:474-566). */
typedef struct {
  uint32_t distance;         /* -d */
  int32_t hamming;           /* -n */
  uint64_t max_locations;    /* -m (default 10000) */
  uint32_t max_neighborhood; /* -x */
  uint32_t kmer;             /* -k (default 15) */
  double cut_temp;           /* -c (default 45.0) */
} dg_search_params;
#define DG_P_THAL_FAILED 16u /* "Error: Thermodynamical calculation failed!" (silica.h:438-442, 512-516) */
typedef struct {
  uint32_t ref;      /* refIndex */
  uint32_t pos;      /* PrimerBind::pos (0-based) */
  uint32_t primer;   /* primerId */
  uint8_t on_for;    /* forward-strand site */
  uint8_t reserved[3];
  double temp;       /* Tm of primer vs site */
  double perf_temp;  /* Tm of primer vs its perfect complement */
  uint64_t genome_off;
  uint32_t genome_len;
  uint32_t pad;
} dg_site;
typedef struct {
  size_t nprimers;
  uint64_t nsites;
  dg_site* sites;     /* reference push order: primer-major, forward-strand hits then reverse-strand hits */
  char* genome_pool;  /* PrimerBind::genome strings */
  uint32_t* pflags;   /* [nprimers] DG_Q_MAX_MATCHES | DG_Q_NBHD_EXCEEDED | DG_P_THAL_FAILED */
  double* match_temp; /* [nprimers] */
  uint64_t nhits;     /* located hits, each of which went through thal() */
  double ms_device;   /* device time of the search + site kernels (HIP events) */
  /* measurement, as in dg_hunt_result: the FM-index part and the per-hit thal()/alignment part of ms_device */
  double ms_fm_search, ms_site_stage;
  uint64_t ctr_ext_steps, ctr_tab_reads, ctr_filter_probes, ctr_sa_reads;
} dg_search_result;
/* primers: already upper-cased / N-replaced sequences (silica.h:368), each at least `kmer` long */
int dg_search_sites(dg_index* ix, dg_thal* th, const dg_search_params* p, const uint32_t* seqlen, uint32_t nseq,
                    const uint8_t* pbytes, const uint64_t* poff, size_t nprimers, dg_search_result** out);
void dg_search_result_free(dg_search_result* r);

/* ABI 7.  Acceptance check of an index file WITHOUT a device (reference: the file sdsl::store_to_checked_file writes at
 * src/index.h:121-122 and load_from_checked_file reads at src/hunter.h:253-256).  The sections of csa_wt<wt_huff<>, 32, 64> are
 * walked byte by byte and held against each other; with DG_FM9_CHECK_DEEP the large sections are read through as well (rank words
 * against the bit vector's popcounts, node sizes / offsets of the Huffman tree, C[] against the leaf sizes, SA / ISA samples).
 * `report` (may be NULL) receives one JSON object: {"ok", "layout", "n", "sigma", "sections":[{"name","offset","bytes"}...],
 * "error"}.  Returns DG_OK, DG_EIO, or DG_EFORMAT with a message (also dg_last_error()) that NAMES the first section whose byte
 * count or invariant is off — what a maintainer needs when the first genuine `dicey index` file does not load. */
#define DG_FM9_CHECK_DEEP 1u
int dg_fm9_check(const char* fm9_path, uint32_t flags, char* report, size_t report_cap);

const char* dg_last_error(void);
int dg_abi_version(void);
int dg_device_count(void);

#ifdef __cplusplus
}
#endif
#endif
