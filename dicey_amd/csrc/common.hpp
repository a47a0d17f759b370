// Shared host-side plumbing of libdiceygpu: error reporting, HIP call checking, device buffers.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/dicey_gpu.h"

// dynamic LDS of a kernel (size given at launch)
#ifndef DG_DYNAMIC_LDS
#define DG_DYNAMIC_LDS(name) extern __shared__ __align__(16) unsigned char name[]
#endif

namespace dg {

using u8 = uint8_t;
using u16 = uint16_t;
using u32 = uint32_t;
using u64 = uint64_t;
using i32 = int32_t;

std::string& last_error();  // thread-local, defined in index.hip
int fail(int code, const char* fmt, ...);

#define DG_HIP(expr)                                                                                 \
  do {                                                                                               \
    hipError_t e__ = (expr);                                                                         \
    if (e__ != hipSuccess)                                                                           \
      return ::dg::fail(e__ == hipErrorOutOfMemory ? DG_ENOMEM : DG_EHIP, "%s failed: %s (%s:%d)", #expr, \
                        hipGetErrorString(e__), __FILE__, __LINE__);                                 \
  } while (0)

#define DG_TRY(expr)          \
  do {                        \
    int rc__ = (expr);        \
    if (rc__ != DG_OK) return rc__; \
  } while (0)

// Grow-only device buffer (batch workspaces are reused across calls; hipMalloc is not on the hot path).
struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
  unsigned long long gen = 0;  // bumped whenever the block is replaced: what it held is gone
  int reserve(size_t bytes) {
    if (bytes <= cap) return DG_OK;
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
    size_t want = bytes + bytes / 4 + 256;
    hipError_t e = hipMalloc(&p, want);
    if (e != hipSuccess) {
      p = nullptr;
      return fail(DG_ENOMEM, "hipMalloc(%zu) failed: %s", want, hipGetErrorString(e));
    }
    cap = want;
    ++gen;
    return DG_OK;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
  }
  template <class T>
  T* as() const {
    return (T*)p;
  }
};

static inline u32 ceil_div(u64 a, u64 b) { return (u32)((a + b - 1) / b); }

// Large device allocations go through the stream-ordered pool: hipMalloc costs ~40 ms per GiB on this stack (5.8 s for the
// 137 GB K-mer table, tools/microbench/malloc_bench.hip), hipMallocAsync 0.1-0.3 s for the same block.  Memory obtained
// here is released with big_free on the same stream; the caller synchronises the stream before other streams use it.
static inline hipError_t big_alloc(void** p, size_t bytes, hipStream_t st) {
  if (bytes >= ((size_t)64 << 20)) {  // (r05 A/B of the pool against hipMalloc: no difference in kernel time, 0.157 ms either way)
    const hipError_t e = hipMallocAsync(p, bytes, st);
    if (e == hipSuccess) return hipSuccess;
    if (std::getenv("DICEY_TIMING")) std::fprintf(stderr, "dicey timing: hipMallocAsync(%zu) failed: %s\n", bytes, hipGetErrorString(e));
    (void)hipGetLastError();
  }
  return hipMalloc(p, bytes);
}
static inline void big_free(void* p, hipStream_t st) {
  if (!p) return;
  if (hipFreeAsync(p, st) != hipSuccess) {
    (void)hipGetLastError();
    (void)hipFree(p);
  }
}

}  // namespace dg
