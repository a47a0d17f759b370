/* dicey_gather.h — C ABI of libdiceygather.so: the gather of per-GPU hit lists to one rank over RCCL / xGMI.
 *
 * BASELINE.json north_star: "the primer batch shards embarrassingly across the 8 GPUs of one node with RCCL over xGMI only to
 * gather hit lists" — one process per GPU, every rank searches its contiguous slice of the batch (hunter.h:291 treats queries
 * independently) against its own index replica, and only the variable-length hit lists travel.  The reference has
 * no counterpart (it is one process, one thread); what this replaces in a sharded run of hunter.h:291-444 is the concatenation
 * of the shards' DnaHit lists in query order.
 *
 * The library links RCCL and the HIP runtime libdiceygpu.so links: the staging copy of a batch's records is queued on the stream
 * the batch ran on (dg_hunt_result::stream), so it is ordered before the lane's next batch WITHOUT a host synchronisation, and
 * the transfers run on the communicator's own stream beside the next batches' kernels.
 *
 * Protocol (the same on every rank, every step): submit(k) stages payload k, starts the exchange of the ranks' byte counts for
 * step k (ncclAllGather of one uint64 per rank, read back into pinned memory), and launches the transfer of step k - 1 — whose
 * counts were exchanged one submit ago — as ONE group of exact-size ncclSend / ncclRecv: every rank sends the bytes it has, the
 * root receives exactly those (no padding, no capacity travels).  finish() launches what is left and waits.
 *
 * Host language: C++ above this C ABI (dicey_amd/cli: `dicey hunt --ranks N --rank R`), ctypes in bench.py.
 */
#ifndef DICEY_GATHER_H
#define DICEY_GATHER_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define DG_COMM_ID_BYTES 128 /* sizeof(ncclUniqueId) */
typedef struct dg_comm dg_comm;

/* rank 0: a fresh communicator id (ncclGetUniqueId); it reaches the other ranks through whatever started them (a file, an
 * environment variable, torch.distributed's store) */
int dg_comm_unique_id(uint8_t id[DG_COMM_ID_BYTES]);
/* RCCL communicator of `nranks` processes, this one being `rank` on HIP device `device`.  capacity: upper bound of one step's
 * payload on any rank (staging ring of 3 slots here; on the root additionally 3 x nranks receive buffers); root: the gathering rank. */
int dg_comm_open(const uint8_t id[DG_COMM_ID_BYTES], int nranks, int rank, int device, uint64_t capacity, int root, dg_comm** out);
/* The same protocol over TCP sockets between processes on 127.0.0.1, payloads in HOST memory: the transport of the CPU test suite
 * (world size 2-8 without a GPU) — the size agreement, the one-step-behind pipeline, ordering and the ring of slots are the code
 * the RCCL form runs; only the three transport calls differ.  `dicey hunt --ranks` and bench.py never open this form. */
int dg_comm_open_tcp(int port, int nranks, int rank, uint64_t capacity, int root, dg_comm** out);
int dg_comm_close(dg_comm* c);

/* Hands in one step's payload: d_payload[0, nbytes) on this rank's device (host memory for the TCP form).  producer_stream: the
 * hipStream_t the payload's producer ran on (dg_hunt_result::stream, or dg_index_stream) — the staging copy is queued there, the call
 * returns without waiting for it (NULL: the payload is complete already; the copy goes to the communicator's stream).
 * nbytes <= capacity; 0 is a legal payload (a rank behind the end of the batch still takes part in every step). */
int dg_gather_submit(dg_comm* c, void* producer_stream, const void* d_payload, uint64_t nbytes);
/* Launches the transfers not yet launched and waits for all of them.  Root: *payload_bytes = bytes of payload received (own
 * included) since the previous finish, *steps = steps completed since then.  Either pointer may be NULL. */
int dg_gather_finish(dg_comm* c, uint64_t* payload_bytes, uint64_t* steps);
/* Root, after dg_gather_finish: what rank `r` sent in the most recent step.  *ptr is a device pointer (host pointer for the TCP
 * form) valid until the next submit. */
int dg_gather_last(dg_comm* c, int r, const void** ptr, uint64_t* nbytes);
/* the same, copied into host memory (out_capacity >= *nbytes) */
int dg_gather_last_to_host(dg_comm* c, int r, void* out, uint64_t out_capacity, uint64_t* nbytes);
/* MAX over the ranks of a host value (a collective: every rank calls it; used once to agree on the capacity and by bench.py for
 * the max-over-ranks time) */
int dg_comm_max_u64(dg_comm* c, uint64_t mine, uint64_t* out);
/* barrier of the communicator's ranks (a collective + host wait) */
int dg_comm_barrier(dg_comm* c);
const char* dg_gather_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
