// Shared host-side plumbing of libdiceygpu: error reporting, HIP call checking, device buffers.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
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

}  // namespace dg
