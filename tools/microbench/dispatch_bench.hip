// How does a chain of small kernels fare BESIDE a chip-filling gather kernel on another stream?  (measurement tool, r06)
// A = the search kernel's shape: 16 384 work items of 256 lanes, each a few dependent random 64-byte-line reads of an 8 GiB buffer,
//     64 VGPRs (eight workgroups per CU).  flat: one workgroup per item.  persistent: G workgroups, each takes items by stride —
//     with G below the chip's 2 048 workgroup slots nothing of A is ever PENDING, the slots it leaves are free for other queues.
// B = five dependent small kernels (400 workgroups each, a few reads) on a second stream, launched while A runs.
// usage: dispatch_bench [items] [iters_per_item]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <string>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t x) {
  x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return x;
}
__device__ __forceinline__ uint64_t item_work(const uint4* buf, uint64_t nlines, uint64_t item, int iters) {
  uint64_t s = mix(item * 256 + threadIdx.x + 1), acc = 0;
  for (int i = 0; i < iters; ++i) {
    const uint4 v = buf[(s & (nlines - 1)) * 4 + (threadIdx.x & 3)];
    acc += v.x + v.w;
    s = mix(s + v.y);
  }
  return acc;
}
__global__ void __launch_bounds__(256, 8) k_a_flat(const uint4* buf, uint64_t nlines, int iters, uint64_t* out) {
  const uint64_t acc = item_work(buf, nlines, blockIdx.x, iters);
  if (acc == 0x1234567ULL) out[0] = acc;
}
__global__ void __launch_bounds__(256, 8) k_a_persist(const uint4* buf, uint64_t nlines, int iters, uint32_t items, uint64_t* out) {
  uint64_t acc = 0;
  for (uint32_t it = blockIdx.x; it < items; it += gridDim.x) acc += item_work(buf, nlines, it, iters);
  if (acc == 0x1234567ULL) out[0] = acc;
}
__global__ void __launch_bounds__(256) k_b(const uint4* buf, uint64_t nlines, int iters, uint64_t* out, uint64_t salt) {
  const uint64_t acc = item_work(buf, nlines, blockIdx.x + salt, iters);
  if (acc == 0x1234567ULL) out[1] = acc;
}

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv) {
  const uint32_t items = argc > 1 ? (uint32_t)atoi(argv[1]) : 16384;
  const int iters = argc > 2 ? atoi(argv[2]) : 12;
  const uint64_t bytes = 8ULL << 30, nlines = bytes / 64;
  uint4* buf;
  uint64_t* out;
  CK(hipMalloc(&buf, bytes));
  CK(hipMemset(buf, 1, bytes));
  CK(hipMalloc(&out, 64));
  hipStream_t sa, sa2, sb;
  CK(hipStreamCreate(&sa));
  CK(hipStreamCreate(&sa2));
  CK(hipStreamCreate(&sb));
  hipEvent_t a0, a1, b0, b1, c0, c1;
  for (hipEvent_t* evp : {&a0, &a1, &b0, &b1, &c0, &c1}) CK(hipEventCreate(evp));
  auto launch_a = [&](hipStream_t s, uint32_t grid) {  // grid 0 = flat
    if (grid == 0) hipLaunchKernelGGL(k_a_flat, dim3(items), dim3(256), 0, s, buf, nlines, iters, out);
    else hipLaunchKernelGGL(k_a_persist, dim3(grid), dim3(256), 0, s, buf, nlines, iters, items, out);
  };
  auto chain_b = [&](hipStream_t s) {
    for (int k = 0; k < 5; ++k) hipLaunchKernelGGL(k_b, dim3(400), dim3(256), 0, s, buf, nlines, 4, out, (uint64_t)(1000000 + k * 1000));
  };
  // warm up
  launch_a(sa, 0);
  launch_a(sa, 2048);
  chain_b(sb);
  CK(hipDeviceSynchronize());
  auto wait_us = [](double us) { const double t = now_us(); while (now_us() - t < us) {} };
  float ms = 0;
  {  // B alone
    CK(hipEventRecord(b0, sb));
    chain_b(sb);
    CK(hipEventRecord(b1, sb));
    CK(hipDeviceSynchronize());
    CK(hipEventElapsedTime(&ms, b0, b1));
    printf("B alone (5 dependent kernels of 400 workgroups)        %8.1f us\n", ms * 1e3);
  }
  for (uint32_t grid : {0u, 2048u, 1920u, 1792u, 1536u, 1024u}) {
    for (int two = 0; two < 2; ++two) {
      for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(a0, sa));
        launch_a(sa, grid);
        CK(hipEventRecord(a1, sa));
        if (two) {
          CK(hipEventRecord(c0, sa2));
          launch_a(sa2, grid);
          CK(hipEventRecord(c1, sa2));
        }
        wait_us(40);
        CK(hipEventRecord(b0, sb));
        chain_b(sb);
        CK(hipEventRecord(b1, sb));
        CK(hipDeviceSynchronize());
        float ta = 0, tb = 0, tc = 0, lag = 0;
        CK(hipEventElapsedTime(&ta, a0, a1));
        CK(hipEventElapsedTime(&tb, b0, b1));
        CK(hipEventElapsedTime(&lag, a0, b1));
        if (two) { CK(hipEventElapsedTime(&tc, c0, c1)); }
        if (rep) printf("A %-15s%s  A %7.1f us%s   B chain beside it %7.1f us (ends %7.1f us after A's start)\n", grid ? ("persistent " + std::to_string(grid)).c_str() : "flat",
                        two ? " x2 streams" : "           ", ta * 1e3, two ? (" / " + std::to_string((int)(tc * 1e3)) + " us").c_str() : "          ", tb * 1e3, lag * 1e3);
      }
    }
  }
  {  // A alone, both forms
    for (uint32_t grid : {0u, 2048u, 1792u, 1536u}) {
      CK(hipEventRecord(a0, sa));
      launch_a(sa, grid);
      CK(hipEventRecord(a1, sa));
      CK(hipDeviceSynchronize());
      CK(hipEventElapsedTime(&ms, a0, a1));
      printf("A alone %-16s %7.1f us\n", grid ? ("persistent " + std::to_string(grid)).c_str() : "flat", ms * 1e3);
    }
  }
  return 0;
}
