// libdiceygather.so — the gather of per-GPU hit lists to one rank (include/dicey_gather.h).
//
// One pipeline (Gather: ring of slots, size agreement one step ahead of the transfer, exact-size transfers) over two transports:
//   RcclTransport  RCCL over xGMI: ncclAllGather of the byte counts, one ncclGroup of ncclSend / ncclRecv per step, on the
//                  communicator's own HIP stream; staging copies on the producer's stream.  The product.
//   TcpTransport   sockets on 127.0.0.1 and host memory: the CPU suite's stand-in for a node of GPUs (tests/test_gather_tcp.py).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <arpa/inet.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <sys/socket.h>
#include <sys/time.h>
#include <unistd.h>

#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/dicey_gather.h"

namespace {

using u8 = uint8_t;
using u64 = uint64_t;
enum { G_OK = 0, G_EINVAL = 1, G_EHIP = 4, G_ENOMEM = 5, G_ELIMIT = 6, G_ECOMM = 9 };

thread_local std::string g_err;
int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  std::vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}
#define G_HIP(x)                                                                                  \
  do {                                                                                            \
    hipError_t e_ = (x);                                                                          \
    if (e_ != hipSuccess) return fail(G_EHIP, "%s: %s", #x, hipGetErrorString(e_));               \
  } while (0)
#define G_NCCL(x)                                                                                 \
  do {                                                                                            \
    ncclResult_t r_ = (x);                                                                        \
    if (r_ != ncclSuccess) return fail(G_ECOMM, "%s: %s", #x, ncclGetErrorString(r_));            \
  } while (0)
#define G_TRY(x)                \
  do {                          \
    int rc_ = (x);              \
    if (rc_ != G_OK) return rc_; \
  } while (0)

constexpr int DEPTH = 3;  // slots: one being staged, one whose sizes travel, one whose payload travels

struct Transport {
  int nranks = 1, rank = 0, root = 0;
  virtual ~Transport() {}
  virtual int alloc(void** p, u64 bytes) = 0;
  virtual void release(void* p) = 0;
  // copy of a payload into a staging slot, ordered behind the producer's queued work
  virtual int stage(int slot, void* producer_stream, void* dst, const void* src, u64 n) = 0;
  virtual int sizes_begin(int slot, u64 mine) = 0;          // start the exchange of this step's byte counts
  virtual int sizes_end(int slot, u64* all) = 0;            // ... and its result on the host (waits if it has to)
  // one step's transfer: everybody's `send[0, sizes[rank])` to the root's recv[r]; queued behind the slot's staging copy
  virtual int transfer(int slot, const void* send, void* const* recv, const u64* sizes) = 0;
  virtual int transfer_wait(int slot) = 0;
  virtual int to_host(void* dst, const void* src, u64 n) = 0;
  virtual int max_u64(u64 mine, u64* out) = 0;
};

// ---------------------------------------------------------------------------------------------------------------- RCCL
struct RcclTransport : Transport {
  ncclComm_t comm = nullptr;
  hipStream_t cs = nullptr;  // the communicator's stream: size exchanges and transfers, in the same order on every rank
  int device = 0;
  hipEvent_t ev_staged[DEPTH] = {}, ev_sizes[DEPTH] = {}, ev_done[DEPTH] = {};
  u64* h_mine = nullptr;   // pinned [DEPTH]
  u64* h_sizes = nullptr;  // pinned [DEPTH][nranks]
  u64* d_mine = nullptr;   // [DEPTH]
  u64* d_sizes = nullptr;  // [DEPTH][nranks]
  u64* d_red = nullptr;    // scratch of max_u64
  u64* h_red = nullptr;

  int open(const u8* id, int n, int r, int dev, int root_) {
    nranks = n;
    rank = r;
    root = root_;
    device = dev;
    G_HIP(hipSetDevice(dev));
    ncclUniqueId uid;
    static_assert(sizeof(ncclUniqueId) == DG_COMM_ID_BYTES, "ncclUniqueId size");
    std::memcpy(&uid, id, sizeof uid);
    G_NCCL(ncclCommInitRank(&comm, n, uid, r));
    G_HIP(hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
    for (int i = 0; i < DEPTH; ++i) {
      G_HIP(hipEventCreateWithFlags(&ev_staged[i], hipEventDisableTiming));
      G_HIP(hipEventCreateWithFlags(&ev_sizes[i], hipEventDisableTiming));
      G_HIP(hipEventCreateWithFlags(&ev_done[i], hipEventDisableTiming));
    }
    G_HIP(hipHostMalloc((void**)&h_mine, DEPTH * 8, 0));
    G_HIP(hipHostMalloc((void**)&h_sizes, (size_t)DEPTH * n * 8, 0));
    G_HIP(hipHostMalloc((void**)&h_red, 16, 0));
    G_HIP(hipMalloc((void**)&d_mine, DEPTH * 8));
    G_HIP(hipMalloc((void**)&d_sizes, (size_t)DEPTH * n * 8));
    G_HIP(hipMalloc((void**)&d_red, 16));
    return G_OK;
  }
  ~RcclTransport() override {
    (void)hipSetDevice(device);
    if (cs) (void)hipStreamSynchronize(cs);
    if (comm) (void)ncclCommDestroy(comm);
    for (int i = 0; i < DEPTH; ++i) {
      if (ev_staged[i]) (void)hipEventDestroy(ev_staged[i]);
      if (ev_sizes[i]) (void)hipEventDestroy(ev_sizes[i]);
      if (ev_done[i]) (void)hipEventDestroy(ev_done[i]);
    }
    if (h_mine) (void)hipHostFree(h_mine);
    if (h_sizes) (void)hipHostFree(h_sizes);
    if (h_red) (void)hipHostFree(h_red);
    if (d_mine) (void)hipFree(d_mine);
    if (d_sizes) (void)hipFree(d_sizes);
    if (d_red) (void)hipFree(d_red);
    if (cs) (void)hipStreamDestroy(cs);
  }
  int alloc(void** p, u64 bytes) override {
    G_HIP(hipSetDevice(device));
    G_HIP(hipMalloc(p, bytes ? bytes : 8));
    return G_OK;
  }
  void release(void* p) override {
    if (p) (void)hipFree(p);
  }
  int stage(int slot, void* producer_stream, void* dst, const void* src, u64 n) override {
    hipStream_t ps = producer_stream ? (hipStream_t)producer_stream : cs;
    if (n) G_HIP(hipMemcpyAsync(dst, src, n, hipMemcpyDeviceToDevice, ps));
    G_HIP(hipEventRecord(ev_staged[slot], ps));  // the transfer of this slot waits for it on the communicator's stream
    return G_OK;
  }
  int sizes_begin(int slot, u64 mine) override {
    h_mine[slot] = mine;
    G_HIP(hipMemcpyAsync(d_mine + slot, h_mine + slot, 8, hipMemcpyHostToDevice, cs));
    G_NCCL(ncclAllGather(d_mine + slot, d_sizes + (size_t)slot * nranks, 1, ncclUint64, comm, cs));
    G_HIP(hipMemcpyAsync(h_sizes + (size_t)slot * nranks, d_sizes + (size_t)slot * nranks, (size_t)nranks * 8, hipMemcpyDeviceToHost, cs));
    G_HIP(hipEventRecord(ev_sizes[slot], cs));
    return G_OK;
  }
  int sizes_end(int slot, u64* all) override {
    G_HIP(hipEventSynchronize(ev_sizes[slot]));  // queued one submit ago: a step's kernels have run since
    std::memcpy(all, h_sizes + (size_t)slot * nranks, (size_t)nranks * 8);
    return G_OK;
  }
  int transfer(int slot, const void* send, void* const* recv, const u64* sizes) override {
    G_HIP(hipStreamWaitEvent(cs, ev_staged[slot], 0));
    if (rank == root && sizes[rank]) G_HIP(hipMemcpyAsync(recv[rank], send, sizes[rank], hipMemcpyDeviceToDevice, cs));
    bool any = false;
    for (int r = 0; r < nranks; ++r) any = any || (r != root && sizes[r]);
    if (any) {
      G_NCCL(ncclGroupStart());
      if (rank != root) {
        if (sizes[rank]) G_NCCL(ncclSend(send, sizes[rank], ncclUint8, root, comm, cs));
      } else {
        for (int r = 0; r < nranks; ++r)
          if (r != root && sizes[r]) G_NCCL(ncclRecv(recv[r], sizes[r], ncclUint8, r, comm, cs));
      }
      G_NCCL(ncclGroupEnd());
    }
    G_HIP(hipEventRecord(ev_done[slot], cs));
    return G_OK;
  }
  int transfer_wait(int slot) override {
    G_HIP(hipEventSynchronize(ev_done[slot]));
    return G_OK;
  }
  int to_host(void* dst, const void* src, u64 n) override {
    if (n) G_HIP(hipMemcpy(dst, src, n, hipMemcpyDeviceToHost));
    return G_OK;
  }
  int max_u64(u64 mine, u64* out) override {
    h_red[0] = mine;
    G_HIP(hipMemcpyAsync(d_red, h_red, 8, hipMemcpyHostToDevice, cs));
    G_NCCL(ncclAllReduce(d_red, d_red + 1, 1, ncclUint64, ncclMax, comm, cs));
    G_HIP(hipMemcpyAsync(h_red + 1, d_red + 1, 8, hipMemcpyDeviceToHost, cs));
    G_HIP(hipStreamSynchronize(cs));
    *out = h_red[1];
    return G_OK;
  }
};

// ----------------------------------------------------------------------------------------------------------------- TCP
// A star around the root: rank r != root holds one connection to it.  Everything is blocking and happens inside the call that
// asks for it, in the order the pipeline asks — the order the RCCL form queues on its stream.
struct TcpTransport : Transport {
  std::vector<int> fd;  // root: fd[r] per rank (-1 for itself); others: fd[0] = the connection to the root
  int lfd = -1;
  std::vector<u64> pending[DEPTH];
  static bool xsend(int s, const void* p, u64 n) {
    const u8* b = (const u8*)p;
    while (n) {
      ssize_t k = ::send(s, b, n > (1u << 30) ? (1u << 30) : n, MSG_NOSIGNAL);
      if (k <= 0) return false;
      b += k;
      n -= (u64)k;
    }
    return true;
  }
  static bool xrecv(int s, void* p, u64 n) {
    u8* b = (u8*)p;
    while (n) {
      ssize_t k = ::recv(s, b, n > (1u << 30) ? (1u << 30) : n, 0);
      if (k <= 0) return false;
      b += k;
      n -= (u64)k;
    }
    return true;
  }
  int open(int port, int n, int r, int root_) {
    nranks = n;
    rank = r;
    root = root_;
    sockaddr_in a{};
    a.sin_family = AF_INET;
    a.sin_port = htons((uint16_t)port);
    a.sin_addr.s_addr = htonl(INADDR_LOOPBACK);
    const int one = 1;
    if (r == root) {
      fd.assign(n, -1);
      lfd = ::socket(AF_INET, SOCK_STREAM, 0);
      if (lfd < 0) return fail(G_ECOMM, "socket failed");
      ::setsockopt(lfd, SOL_SOCKET, SO_REUSEADDR, &one, sizeof one);
      if (::bind(lfd, (sockaddr*)&a, sizeof a) != 0 || ::listen(lfd, n) != 0) return fail(G_ECOMM, "cannot listen on 127.0.0.1:%d", port);
      // a peer that dies at start-up must not leave the root in accept() / recv() for ever: 60 s for a peer to
      // connect, 60 s for its hello
      timeval tmo{};
      tmo.tv_sec = 60;
      ::setsockopt(lfd, SOL_SOCKET, SO_RCVTIMEO, &tmo, sizeof tmo);
      for (int k = 0; k < n - 1; ++k) {
        int s = ::accept(lfd, nullptr, nullptr);
        if (s < 0) return fail(G_ECOMM, "no peer connected within 60 s (%d of %d are here)", k, n - 1);
        ::setsockopt(s, IPPROTO_TCP, TCP_NODELAY, &one, sizeof one);
        ::setsockopt(s, SOL_SOCKET, SO_RCVTIMEO, &tmo, sizeof tmo);
        int32_t who = -1;
        if (!xrecv(s, &who, 4) || who < 0 || who >= n || who == root || fd[who] != -1) {
          ::close(s);
          return fail(G_ECOMM, "bad hello from a peer");
        }
        timeval none{};
        ::setsockopt(s, SOL_SOCKET, SO_RCVTIMEO, &none, sizeof none);  // (the payload transfers wait as long as the peers compute)
        fd[who] = s;
      }
    } else {
      fd.assign(1, -1);
      for (int attempt = 0; attempt < 600; ++attempt) {  // the root may not be listening yet
        int s = ::socket(AF_INET, SOCK_STREAM, 0);
        if (s < 0) return fail(G_ECOMM, "socket failed");
        if (::connect(s, (sockaddr*)&a, sizeof a) == 0) {
          ::setsockopt(s, IPPROTO_TCP, TCP_NODELAY, &one, sizeof one);
          fd[0] = s;
          break;
        }
        ::close(s);
        std::this_thread::sleep_for(std::chrono::milliseconds(50));
      }
      if (fd[0] < 0) return fail(G_ECOMM, "cannot reach the root on 127.0.0.1:%d", port);
      const int32_t who = r;
      if (!xsend(fd[0], &who, 4)) return fail(G_ECOMM, "hello failed");
    }
    return G_OK;
  }
  ~TcpTransport() override {
    for (int s : fd)
      if (s >= 0) ::close(s);
    if (lfd >= 0) ::close(lfd);
  }
  int alloc(void** p, u64 bytes) override {
    *p = std::malloc(bytes ? bytes : 8);
    return *p ? G_OK : fail(G_ENOMEM, "out of host memory (%llu bytes)", (unsigned long long)bytes);
  }
  void release(void* p) override { std::free(p); }
  int stage(int, void*, void* dst, const void* src, u64 n) override {
    if (n) std::memcpy(dst, src, n);
    return G_OK;
  }
  int allgather(u64 mine, u64* all) {
    if (rank == root) {
      all[root] = mine;
      for (int r = 0; r < nranks; ++r)
        if (r != root && !xrecv(fd[r], &all[r], 8)) return fail(G_ECOMM, "size of rank %d did not arrive", r);
      for (int r = 0; r < nranks; ++r)
        if (r != root && !xsend(fd[r], all, (u64)nranks * 8)) return fail(G_ECOMM, "sizes to rank %d failed", r);
    } else {
      if (!xsend(fd[0], &mine, 8) || !xrecv(fd[0], all, (u64)nranks * 8)) return fail(G_ECOMM, "size exchange failed");
    }
    return G_OK;
  }
  int sizes_begin(int slot, u64 mine) override {
    pending[slot].assign(nranks, 0);
    return allgather(mine, pending[slot].data());
  }
  int sizes_end(int slot, u64* all) override {
    std::memcpy(all, pending[slot].data(), (size_t)nranks * 8);
    return G_OK;
  }
  int transfer(int, const void* send, void* const* recv, const u64* sizes) override {
    if (rank == root) {
      if (sizes[rank]) std::memcpy(recv[rank], send, sizes[rank]);
      for (int r = 0; r < nranks; ++r)
        if (r != root && sizes[r] && !xrecv(fd[r], recv[r], sizes[r])) return fail(G_ECOMM, "payload of rank %d did not arrive", r);
    } else if (sizes[rank] && !xsend(fd[0], send, sizes[rank])) return fail(G_ECOMM, "payload send failed");
    return G_OK;
  }
  int transfer_wait(int) override { return G_OK; }
  int to_host(void* dst, const void* src, u64 n) override {
    if (n) std::memcpy(dst, src, n);
    return G_OK;
  }
  int max_u64(u64 mine, u64* out) override {
    std::vector<u64> all(nranks);
    G_TRY(allgather(mine, all.data()));
    u64 m = 0;
    for (u64 v : all) m = v > m ? v : m;
    *out = m;
    return G_OK;
  }
};

}  // namespace

// ------------------------------------------------------------------------------------------------------------ pipeline
struct dg_comm {
  Transport* tr = nullptr;
  u64 capacity = 0;
  void* send[DEPTH] = {};
  std::vector<void*> recv[DEPTH];  // root: [nranks] each
  u64 mine[DEPTH] = {};
  std::vector<u64> sizes[DEPTH];
  bool in_flight[DEPTH] = {};      // the slot's transfer has been launched and not yet waited for
  u64 staged = 0, launched = 0;    // steps handed in / steps whose transfer has been launched
  u64 payload_bytes = 0, steps_done = 0;
  int last_slot = -1;
  ~dg_comm() {
    if (tr) {
      for (int s = 0; s < DEPTH; ++s) {
        tr->release(send[s]);
        for (void* p : recv[s]) tr->release(p);
      }
      delete tr;
    }
  }
  int drain(int slot) {
    if (!in_flight[slot]) return G_OK;
    G_TRY(tr->transfer_wait(slot));
    in_flight[slot] = false;
    if (tr->rank == tr->root)
      for (u64 v : sizes[slot]) payload_bytes += v;
    ++steps_done;
    last_slot = slot;
    return G_OK;
  }
  int launch(u64 step) {
    const int slot = (int)(step % DEPTH);
    sizes[slot].assign(tr->nranks, 0);
    G_TRY(tr->sizes_end(slot, sizes[slot].data()));
    for (int r = 0; r < tr->nranks; ++r)
      if (sizes[slot][r] > capacity)
        return fail(G_ELIMIT, "rank %d sends %llu bytes, the communicator was opened for %llu", r, (unsigned long long)sizes[slot][r],
                    (unsigned long long)capacity);
    if (sizes[slot][tr->rank] != mine[slot]) return fail(G_ECOMM, "size exchange out of step (slot %d)", slot);
    G_TRY(tr->transfer(slot, send[slot], recv[slot].empty() ? nullptr : recv[slot].data(), sizes[slot].data()));
    in_flight[slot] = true;
    launched = step + 1;
    return G_OK;
  }
};

static int open_common(Transport* tr, u64 capacity, dg_comm** out) {
  dg_comm* c = new dg_comm;
  c->tr = tr;
  // the capacity every rank agreed on: the largest any of them asked for
  int rc = tr->max_u64(capacity, &c->capacity);
  for (int s = 0; rc == G_OK && s < DEPTH; ++s) {
    rc = tr->alloc(&c->send[s], c->capacity);
    if (rc == G_OK && tr->rank == tr->root) {
      c->recv[s].assign(tr->nranks, nullptr);
      for (int r = 0; rc == G_OK && r < tr->nranks; ++r) rc = tr->alloc(&c->recv[s][r], c->capacity);
    }
  }
  if (rc != G_OK) {
    delete c;
    return rc;
  }
  *out = c;
  return G_OK;
}

extern "C" {

const char* dg_gather_last_error(void) { return g_err.c_str(); }

int dg_comm_unique_id(uint8_t id[DG_COMM_ID_BYTES]) {
  if (!id) return fail(G_EINVAL, "dg_comm_unique_id: null argument");
  ncclUniqueId uid;
  G_NCCL(ncclGetUniqueId(&uid));
  std::memcpy(id, &uid, sizeof uid);
  return G_OK;
}

int dg_comm_open(const uint8_t id[DG_COMM_ID_BYTES], int nranks, int rank, int device, uint64_t capacity, int root, dg_comm** out) {
  if (!id || !out || nranks < 1 || rank < 0 || rank >= nranks || root < 0 || root >= nranks) return fail(G_EINVAL, "dg_comm_open: bad argument");
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return fail(G_EHIP, "dg_comm_open: no HIP device (the RCCL gather has no CPU form)");
  RcclTransport* tr = new RcclTransport;
  int rc = tr->open(id, nranks, rank, device, root);
  if (rc != G_OK) {
    delete tr;
    return rc;
  }
  return open_common(tr, capacity, out);
}

int dg_comm_open_tcp(int port, int nranks, int rank, uint64_t capacity, int root, dg_comm** out) {
  if (!out || nranks < 1 || rank < 0 || rank >= nranks || root < 0 || root >= nranks || port <= 0) return fail(G_EINVAL, "dg_comm_open_tcp: bad argument");
  *out = nullptr;
  TcpTransport* tr = new TcpTransport;
  int rc = tr->open(port, nranks, rank, root);
  if (rc != G_OK) {
    delete tr;
    return rc;
  }
  return open_common(tr, capacity, out);
}

int dg_comm_close(dg_comm* c) {
  if (!c) return G_OK;
  int rc = dg_gather_finish(c, nullptr, nullptr);
  delete c;
  return rc;
}

int dg_gather_submit(dg_comm* c, void* producer_stream, const void* d_payload, uint64_t nbytes) {
  if (!c || (!d_payload && nbytes)) return fail(G_EINVAL, "dg_gather_submit: null argument");
  if (nbytes > c->capacity)
    return fail(G_ELIMIT, "dg_gather_submit: %llu bytes exceed the communicator's capacity of %llu", (unsigned long long)nbytes,
                (unsigned long long)c->capacity);
  const u64 step = c->staged;
  const int slot = (int)(step % DEPTH);
  G_TRY(c->drain(slot));  // the transfer that used this slot DEPTH steps ago
  c->mine[slot] = nbytes;
  G_TRY(c->tr->stage(slot, producer_stream, c->send[slot], d_payload, nbytes));
  G_TRY(c->tr->sizes_begin(slot, nbytes));
  c->staged = step + 1;
  while (c->launched + 1 < c->staged) G_TRY(c->launch(c->launched));  // the step before this one: its sizes travelled while this step computed
  return G_OK;
}

int dg_gather_finish(dg_comm* c, uint64_t* payload_bytes, uint64_t* steps) {
  if (!c) return fail(G_EINVAL, "dg_gather_finish: null argument");
  while (c->launched < c->staged) G_TRY(c->launch(c->launched));
  for (u64 k = 0; k < DEPTH; ++k) G_TRY(c->drain((int)((c->staged + k) % DEPTH)));  // oldest first
  if (payload_bytes) *payload_bytes = c->payload_bytes;
  if (steps) *steps = c->steps_done;
  c->payload_bytes = 0;
  c->steps_done = 0;
  return G_OK;
}

int dg_gather_last(dg_comm* c, int r, const void** ptr, uint64_t* nbytes) {
  if (!c || !ptr || !nbytes || r < 0 || r >= c->tr->nranks) return fail(G_EINVAL, "dg_gather_last: bad argument");
  if (c->tr->rank != c->tr->root) return fail(G_EINVAL, "dg_gather_last: only the root holds the gathered payloads");
  if (c->last_slot < 0 || c->launched != c->staged) return fail(G_EINVAL, "dg_gather_last: call dg_gather_finish first");
  *ptr = c->recv[c->last_slot][r];
  *nbytes = c->sizes[c->last_slot][r];
  return G_OK;
}

int dg_gather_last_to_host(dg_comm* c, int r, void* out, uint64_t out_capacity, uint64_t* nbytes) {
  const void* p = nullptr;
  u64 n = 0;
  G_TRY(dg_gather_last(c, r, &p, &n));
  if (nbytes) *nbytes = n;
  if (n > out_capacity || (!out && n)) return fail(G_EINVAL, "dg_gather_last_to_host: %llu bytes do not fit", (unsigned long long)n);
  return c->tr->to_host(out, p, n);
}

int dg_comm_max_u64(dg_comm* c, uint64_t mine, uint64_t* out) {
  if (!c || !out) return fail(G_EINVAL, "dg_comm_max_u64: null argument");
  if (c->launched != c->staged) return fail(G_EINVAL, "dg_comm_max_u64: call dg_gather_finish first (collectives must stay in step)");
  return c->tr->max_u64(mine, out);
}

int dg_comm_barrier(dg_comm* c) {
  u64 x = 0;
  return dg_comm_max_u64(c, 1, &x);
}

}  // extern "C"
