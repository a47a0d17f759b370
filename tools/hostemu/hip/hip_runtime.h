// DEVELOPMENT TOOL — NOT PART OF THE PRODUCT, NEVER LOADED BY dicey_amd.
//
// A tiny stand-in for <hip/hip_runtime.h> that lets the UNMODIFIED .hip sources under dicey_amd/csrc be
// compiled with g++ and executed one thread at a time on the host.  Its only purpose is to debug kernel logic
// in the build container, which has no GPU (each real GPU run costs minutes of a limited budget).
// tools/hostemu/Makefile builds libdiceygpu_hostemu.so from it; nothing in dicey_amd/, bench.py,
// __graft_entry__.py or the `-m gpu` tests loads that library, and the package refuses to fall back to it.
//
// Supported: lane-independent kernels (no __syncthreads, no wave collectives), the HIP memory/stream/event calls
// the sources use, and atomics (made real so blocks can be spread over OpenMP threads).
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __shared__ static thread_local
#ifndef __align__
#define __align__(n) alignas(n)
#endif
#define __launch_bounds__(...)
#define __restrict__

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct uint2 { unsigned x, y; };
struct uint3 { unsigned x, y, z; };
struct uint4 { unsigned x, y, z, w; };
static inline uint4 make_uint4(unsigned a, unsigned b, unsigned c, unsigned d) { return uint4{a, b, c, d}; }
static inline uint2 make_uint2(unsigned a, unsigned b) { return uint2{a, b}; }

extern thread_local uint3 threadIdx, blockIdx;
extern thread_local dim3 blockDim, gridDim;

inline void __syncthreads() {
  std::fprintf(stderr, "hostemu: __syncthreads is not emulated\n");
  std::abort();
}

// Wave collectives are not emulated (lanes run one after the other): kernels that use them compile, and abort if a
// launch reaches them.  emu_globals.cpp sets DICEY_NO_WAVE_THAL so the library takes its lane-per-item kernels.
[[noreturn]] inline void emu_no_collectives(const char* what) {
  std::fprintf(stderr, "hostemu: %s is not emulated\n", what);
  std::abort();
}
inline unsigned long long __ballot(int) { emu_no_collectives("__ballot"); }
template <class T> inline T __shfl(T, int) { emu_no_collectives("__shfl"); }
template <class T> inline T __shfl_xor(T, int) { emu_no_collectives("__shfl_xor"); }
template <class T> inline T __shfl_up(T, int) { emu_no_collectives("__shfl_up"); }
template <class T> inline T __shfl_down(T, int) { emu_no_collectives("__shfl_down"); }
inline int __syncthreads_or(int) { emu_no_collectives("__syncthreads_or"); }
inline unsigned __builtin_amdgcn_readfirstlane(unsigned x) { return x; }
inline int __builtin_amdgcn_readlane(int x, int) { return x; }
inline int __builtin_amdgcn_mov_dpp(int, int, int, int, bool) { emu_no_collectives("mov_dpp"); }
inline void __builtin_amdgcn_wave_barrier() {}
inline void __threadfence_system() {}
#define __builtin_amdgcn_fence(order, scope) ((void)0)

typedef int hipError_t;
enum { hipSuccess = 0, hipErrorOutOfMemory = 2, hipErrorInvalidValue = 1 };
enum hipMemcpyKind { hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyHostToHost, hipMemcpyDefault };
typedef struct hipStreamEmu* hipStream_t;
struct hipEventEmu { std::chrono::steady_clock::time_point t; };
typedef hipEventEmu* hipEvent_t;

inline const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "success" : "hostemu error"; }
inline hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
enum { hipDeviceAttributeMultiprocessorCount = 1, hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
inline hipError_t hipDeviceGetAttribute(int* v, int, int) { *v = 8; return hipSuccess; }
inline hipError_t hipFuncSetAttribute(const void*, int, int) { return hipSuccess; }
// kernels with launch-sized LDS only exist in wave-collective form; give them a dummy buffer so they compile
#define DG_DYNAMIC_LDS(name) static thread_local unsigned char name[16]
inline hipError_t hipSetDevice(int) { return hipSuccess; }
inline hipError_t hipGetLastError() { return hipSuccess; }
inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
inline hipError_t hipMalloc(void** p, size_t n) {
  size_t r = (n + 63) / 64 * 64;
  *p = std::aligned_alloc(64, r ? r : 64);
  return *p ? hipSuccess : hipErrorOutOfMemory;
}
inline hipError_t hipFree(void* p) { std::free(p); return hipSuccess; }
inline hipError_t hipMallocAsync(void** p, size_t n, hipStream_t) { return hipMalloc(p, n); }
inline hipError_t hipFreeAsync(void* p, hipStream_t) { std::free(p); return hipSuccess; }
inline hipError_t hipHostMalloc(void** p, size_t n, unsigned = 0) { return hipMalloc(p, n); }
inline hipError_t hipHostFree(void* p) { std::free(p); return hipSuccess; }
inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { std::memcpy(d, s, n); return hipSuccess; }
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { std::memcpy(d, s, n); return hipSuccess; }
inline hipError_t hipMemset(void* d, int v, size_t n) { std::memset(d, v, n); return hipSuccess; }
inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { std::memset(d, v, n); return hipSuccess; }
inline hipError_t hipStreamCreate(hipStream_t* s) { *s = (hipStream_t)std::malloc(8); return hipSuccess; }
inline hipError_t hipStreamDestroy(hipStream_t s) { std::free(s); return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
inline hipError_t hipEventCreate(hipEvent_t* e) { *e = new hipEventEmu; return hipSuccess; }
inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { e->t = std::chrono::steady_clock::now(); return hipSuccess; }
inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) {
  *ms = std::chrono::duration<float, std::milli>(b->t - a->t).count();
  return hipSuccess;
}
inline hipError_t hipMemGetInfo(size_t* f, size_t* t) { *f = *t = (size_t)32 << 30; return hipSuccess; }

inline int __popc(unsigned x) { return __builtin_popcount(x); }
inline int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
inline int __clz(int x) { return x ? __builtin_clz((unsigned)x) : 32; }
inline int __clzll(long long x) { return x ? __builtin_clzll((unsigned long long)x) : 64; }
inline int __ffs(int x) { return __builtin_ffs(x); }
inline int __ffsll(long long x) { return __builtin_ffsll(x); }

template <class T> inline T atomicAdd(T* p, T v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
template <class T> inline T atomicSub(T* p, T v) { return __atomic_fetch_sub(p, v, __ATOMIC_RELAXED); }
template <class T> inline T atomicOr(T* p, T v) { return __atomic_fetch_or(p, v, __ATOMIC_RELAXED); }
template <class T> inline T atomicExch(T* p, T v) { return __atomic_exchange_n(p, v, __ATOMIC_RELAXED); }
template <class T> inline T atomicMax(T* p, T v) {
  T o = __atomic_load_n(p, __ATOMIC_RELAXED);
  while (o < v && !__atomic_compare_exchange_n(p, &o, v, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
  return o;
}
template <class T> inline T atomicMin(T* p, T v) {
  T o = __atomic_load_n(p, __ATOMIC_RELAXED);
  while (o > v && !__atomic_compare_exchange_n(p, &o, v, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
  return o;
}
template <class T> inline T atomicCAS(T* p, T cmp, T v) {
  __atomic_compare_exchange_n(p, &cmp, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED);
  return cmp;
}

template <class K, class... A>
inline void hostemu_launch(K kernel, dim3 grid, dim3 block, A... args) {
#pragma omp parallel for schedule(dynamic, 4)
  for (long long b = 0; b < (long long)grid.x * grid.y; ++b) {
    gridDim = grid;
    blockDim = block;
    blockIdx.x = (unsigned)(b % grid.x);
    blockIdx.y = (unsigned)(b / grid.x);
    blockIdx.z = 0;
    for (unsigned ty = 0; ty < block.y; ++ty)
      for (unsigned tx = 0; tx < block.x; ++tx) {
        threadIdx.x = tx;
        threadIdx.y = ty;
        threadIdx.z = 0;
        kernel(args...);
      }
  }
}
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) hostemu_launch(kernel, grid, block, __VA_ARGS__)
#define HIP_KERNEL_NAME(...) __VA_ARGS__

// lanes run one at a time here: every thread adds its own value
#define DG_HAVE_WAVE_ADD 1
namespace dg { inline void wave_add(unsigned long long* p, uint64_t v) { if (v) __atomic_fetch_add(p, (unsigned long long)v, __ATOMIC_RELAXED); } }
