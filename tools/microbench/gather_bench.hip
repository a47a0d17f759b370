// Random 64-byte-line gather microbenchmark (measurement tool, not part of the product).
// Same access shape as the Occ-block reads of k_search: every lane reads whole 64-byte lines (4 x dwordx4) at
// pseudo-random line addresses of a buffer much larger than the 256 MiB Infinity Cache.  Purposes:
//   (1) the random-gather ceiling that puts roofline.frac into context;
//   (2) a known byte count to calibrate rocprofv3 FETCH_SIZE for this access pattern (MI355X_MICROARCH.md §HBM).
// usage: gather_bench <buffer_GiB> <lines_per_lane> <dependent:0|1> <lanes>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t x) {
  x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return x;
}

template <bool DEP, int PAIR>
__global__ void __launch_bounds__(256) k_gather(const uint4* buf, uint64_t nlines, int iters, uint64_t* out) {
  uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint64_t s = mix(t + 1);
  uint64_t acc = 0;
  for (int i = 0; i < iters; ++i) {
    // PAIR lines in flight per iteration (k_search reads the blocks of both interval ends)
    uint4 v[PAIR][4];
#pragma unroll
    for (int p = 0; p < PAIR; ++p) {
      uint64_t line = mix(s + p) % nlines;
      const uint4* q = buf + line * 4;
      v[p][0] = q[0]; v[p][1] = q[1]; v[p][2] = q[2]; v[p][3] = q[3];
    }
    uint64_t h = 0;
#pragma unroll
    for (int p = 0; p < PAIR; ++p) h += v[p][0].x + v[p][1].y + v[p][2].z + v[p][3].w;
    acc += h;
    s = DEP ? mix(s ^ h) : mix(s + 0x9E3779B97F4A7C15ULL);  // DEP: next address depends on loaded data (a search chain)
  }
  out[t] = acc;
}

int main(int argc, char** argv) {
  double gib = argc > 1 ? atof(argv[1]) : 2.0;
  int iters = argc > 2 ? atoi(argv[2]) : 256;
  int dep = argc > 3 ? atoi(argv[3]) : 1;
  uint64_t lanes = argc > 4 ? strtoull(argv[4], 0, 10) : 200000;
  lanes = (lanes + 255) / 256 * 256;
  uint64_t bytes = (uint64_t)(gib * (1ull << 30)) / 64 * 64, nlines = bytes / 64;
  uint4* buf; uint64_t* out;
  CK(hipMalloc(&buf, bytes)); CK(hipMalloc(&out, lanes * 8));
  CK(hipMemset(buf, 1, bytes));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(a, 0));
    if (dep) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_gather<true, 2>), dim3(lanes / 256), dim3(256), 0, 0, buf, nlines, iters, out);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_gather<false, 2>), dim3(lanes / 256), dim3(256), 0, 0, buf, nlines, iters, out);
    CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    double total = (double)lanes * iters * 2 * 64;
    printf("{\"tool\":\"gather_bench\",\"buffer_GiB\":%.2f,\"lanes\":%llu,\"iters\":%d,\"dependent\":%d,\"bytes\":%.0f,\"ms\":%.3f,\"GBps\":%.1f,\"Glines_per_s\":%.2f}\n",
           gib, (unsigned long long)lanes, iters, dep, total, ms, total / ms / 1e6, total / 64 / ms / 1e6);
  }
  return 0;
}
