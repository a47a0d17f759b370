// Random 64-byte-line gather microbenchmark (measurement tool, not part of the product).
// Same access shape as the Occ-block reads of k_search: every lane reads whole 64-byte lines (4 x dwordx4) at
// pseudo-random line addresses of a buffer much larger than the 256 MiB Infinity Cache.  Purposes:
//   (1) the random-gather ceiling that puts roofline.frac into context;
//   (2) a known byte count to calibrate rocprofv3 FETCH_SIZE for this access pattern (MI355X_MICROARCH.md §HBM).
// usage: gather_bench <buffer_GiB> <lines_per_lane> <dependent:0|1> <lanes>
//        gather_bench filter <copy_GiB> <strands> <lines_per_strand> [rotate:0|1] [reps]   (r03; rotate r04)
//        gather_bench filter2 <log2 lines per copy> <strands> <lines_per_strand> [reps]     (r05: cheap addresses, 8 wavefronts per SIMD)
// `filter`: the access shape of k_search1p's probe phase — four copies of a presence filter (copy_GiB each), one lane per
// (strand, position) with 20 positions per strand, eight independent 4-byte loads per lane; the 160 probes of a strand fall
// into <lines_per_strand> distinct random 64-byte lines (neighbouring positions share lines, like the kernel's choice of copy
// by edit position).  Reports lines/s and probes/s.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cstring>

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

// one lane per (strand, position): 8 probes, spread over the strand's `lps` lines so that neighbouring positions share lines
__global__ void __launch_bounds__(256) k_filter(const uint32_t* buf, uint64_t lines_per_copy, uint32_t lps, uint64_t nstrands, uint64_t* out, uint64_t salt) {
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t strand = t / 20;
  const uint32_t pos = (uint32_t)(t - strand * 20);
  if (strand >= nstrands) return;
  const uint32_t* addr[8];
#pragma unroll
  for (int op = 0; op < 8; ++op) {
    const uint32_t slot = (pos * lps / 20 + (op >= 4 ? 1u : 0u)) % lps;  // two neighbouring line slots per position
    const uint64_t h = mix((strand + salt) * 64 + slot + 1);
    const uint64_t copy = slot * 4 / lps;  // line slots map to the four copies in turn, like edit positions do
    const uint64_t line = copy * lines_per_copy + h % lines_per_copy;
    addr[op] = buf + line * 16 + (mix(t * 8 + op) & 15);
  }
  uint32_t w[8];
#pragma unroll
  for (int op = 0; op < 8; ++op) w[op] = *addr[op];
  uint32_t acc = 0;
#pragma unroll
  for (int op = 0; op < 8; ++op) acc += w[op];
  if (acc == 0x12345678u) out[0] = acc;
}

// r05: the same access shape with address arithmetic as cheap as the kernel's own (~10 instructions per probe).  k_filter above spends
// two 64-bit mixes and a 64-bit modulo per probe — ~400 instructions per lane — and is bound by THAT: 10 G lines/s where the kernel it
// is meant to bound runs at 28 G ("the microbenchmark is slower than the kernel it is supposed to bound").  Here a lane
// hashes its two line slots with one 32-bit multiply each (copy size a power of two: the 18-mer filter's 2^27 lines), issues its eight
// loads back to back like hunt_search.hpp's probe phase, and the kernel is compiled for eight wavefronts per SIMD.
__device__ __forceinline__ uint32_t hash32(uint32_t x) {
  x *= 0x9E3779B1u;
  x ^= x >> 15;
  x *= 0x85EBCA77u;
  return x ^ (x >> 13);
}
__global__ void __launch_bounds__(256, 8) k_filter2(const uint32_t* buf, uint32_t line_mask, uint32_t lines_log2, uint32_t lps, uint32_t nstrands, uint64_t* out, uint32_t salt) {
  // The probe phase's shape (hunt_search.hpp k_search1s): a lane = (strand, position); its eight loads are one deletion, three
  // substitutions and four insertions.  The strings of one operation KIND whose edit lies in the same filter field share a line, and
  // the field's copy of the filter holds it: lps / 3 fields per strand x 3 kinds = lps lines per strand, and ONE wave instruction (one
  // operation of every lane) touches only the lines of its kind — lps / 3 per strand.  (The first r05 form gave every position its own
  // pair of line slots: a wave instruction then touched all lps lines of its strands, 38 lines per instruction instead of 13, and ran
  // at 10 G lines/s where this shape runs at the kernel's rate.)
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t strand = t / 20u, pos = t - strand * 20u;
  if (strand >= nstrands) return;
  const uint32_t nfield = lps / 3u ? lps / 3u : 1u;
  const uint32_t field = pos * nfield / 20u;
  const uint32_t* line[3];
#pragma unroll
  for (uint32_t kind = 0; kind < 3; ++kind) {
    const uint32_t slot = field * 3u + kind;
    const uint32_t h = hash32((strand + salt) * 64u + slot + 1u);
    line[kind] = buf + ((uint64_t)(field * 4u / nfield) << (lines_log2 + 4)) + (uint64_t)(h & line_mask) * 16u;
  }
  const uint32_t* addr[8];
#pragma unroll
  for (int op = 0; op < 8; ++op) addr[op] = line[op == 0 ? 0 : op < 4 ? 1 : 2] + ((pos * 3u + (uint32_t)op * 5u) & 15u);
  uint32_t w[8];
#pragma unroll
  for (int op = 0; op < 8; ++op) w[op] = *addr[op];
  uint32_t acc = 0;
#pragma unroll
  for (int op = 0; op < 8; ++op) acc += w[op];
  if (acc == 0x12345678u) out[0] = acc;
}
// filter2 <copy_log2_lines> <strands> <lines_per_strand> [reps]: four copies of 2^copy_log2_lines lines each
static int filter2_main(int argc, char** argv) {
  const uint32_t lg = argc > 2 ? (uint32_t)atoi(argv[2]) : 27u;
  const uint32_t strands = argc > 3 ? (uint32_t)strtoul(argv[3], 0, 10) : 200000u;
  const uint32_t lps = argc > 4 ? (uint32_t)atoi(argv[4]) : 12u;
  const int reps = argc > 5 ? atoi(argv[5]) : 6;
  const uint64_t bytes = (64ull << lg) * 4;
  uint32_t* buf; uint64_t* out;
  CK(hipMalloc(&buf, bytes)); CK(hipMalloc(&out, 64));
  CK(hipMemset(buf, 1, bytes));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  const uint32_t lanes = (strands * 20u + 255u) / 256u * 256u;
  for (int rep = 0; rep < reps; ++rep) {
    CK(hipEventRecord(a, 0));
    hipLaunchKernelGGL(k_filter2, dim3(lanes / 256), dim3(256), 0, 0, buf, (1u << lg) - 1u, lg, lps, strands, out, (uint32_t)rep * strands);  // fresh lines every repetition
    CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    printf("{\"tool\":\"gather_bench filter2\",\"copies\":4,\"copy_GiB\":%.2f,\"strands\":%u,\"lines_per_strand\":%u,\"probes\":%llu,\"rep\":%d,\"ms\":%.4f,"
           "\"Glines_per_s\":%.2f,\"Gprobes_per_s\":%.2f}\n", (double)(64ull << lg) / (1ull << 30), strands, lps, (unsigned long long)strands * 160ull, rep, ms,
           (double)strands * lps / ms / 1e6, (double)strands * 160 / ms / 1e6);
  }
  return 0;
}

static int filter_main(int argc, char** argv) {
  const double gib = argc > 2 ? atof(argv[2]) : 8.0;
  const uint64_t strands = argc > 3 ? strtoull(argv[3], 0, 10) : 200000;
  const uint32_t lps = argc > 4 ? (uint32_t)atoi(argv[4]) : 12;
  // r04: rotate = 1 gives every repetition its own addresses (a stream of distinct batches); 0 replays the same lines, which the
  // 256 MiB Infinity Cache then serves (r03's numbers: first repetition 10 G lines/s, replays 21 G at 6 lines per strand)
  const int rotate = argc > 5 ? atoi(argv[5]) : 0;
  const int reps = argc > 6 ? atoi(argv[6]) : 4;
  const uint64_t lines_per_copy = (uint64_t)(gib * (1ull << 30)) / 64, bytes = lines_per_copy * 64 * 4;
  uint32_t* buf; uint64_t* out;
  CK(hipMalloc(&buf, bytes)); CK(hipMalloc(&out, 64));
  CK(hipMemset(buf, 1, bytes));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  const uint64_t lanes = (strands * 20 + 255) / 256 * 256;
  for (int rep = 0; rep < reps; ++rep) {
    CK(hipEventRecord(a, 0));
    hipLaunchKernelGGL(k_filter, dim3(lanes / 256), dim3(256), 0, 0, buf, lines_per_copy, lps, strands, out, rotate ? (uint64_t)rep * strands : 0ull);
    CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    printf("{\"tool\":\"gather_bench filter\",\"copies\":4,\"copy_GiB\":%.2f,\"strands\":%llu,\"lines_per_strand\":%u,\"probes\":%llu,\"rotate\":%d,\"rep\":%d,\"ms\":%.4f,"
           "\"Glines_per_s\":%.2f,\"Gprobes_per_s\":%.2f}\n", gib, (unsigned long long)strands, lps, (unsigned long long)(strands * 160), rotate, rep, ms,
           (double)strands * lps / ms / 1e6, (double)strands * 160 / ms / 1e6);
  }
  return 0;
}

int main(int argc, char** argv) {
  if (argc > 1 && !strcmp(argv[1], "filter")) return filter_main(argc, argv);
  if (argc > 1 && !strcmp(argv[1], "filter2")) return filter2_main(argc, argv);
  double gib = argc > 1 ? atof(argv[1]) : 2.0;
  int iters = argc > 2 ? atoi(argv[2]) : 256;
  int dep_arg = argc > 3 ? atoi(argv[3]) : 1;  // 2 = both forms
  // lanes: one value or a comma-separated list (one allocation serves all of them)
  uint64_t lane_list[16];
  int nl = 0;
  {
    const char* p = argc > 4 ? argv[4] : "200000";
    while (*p && nl < 16) {
      lane_list[nl++] = (strtoull(p, (char**)&p, 10) + 255) / 256 * 256;
      if (*p == ',') ++p;
    }
  }
  uint64_t max_lanes = 0;
  for (int i = 0; i < nl; ++i) max_lanes = lane_list[i] > max_lanes ? lane_list[i] : max_lanes;
  uint64_t bytes = (uint64_t)(gib * (1ull << 30)) / 64 * 64, nlines = bytes / 64;
  uint4* buf; uint64_t* out;
  CK(hipMalloc(&buf, bytes)); CK(hipMalloc(&out, max_lanes * 8));
  CK(hipMemset(buf, 1, bytes));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int li = 0; li < nl; ++li)
    for (int dep = (dep_arg == 2 ? 0 : dep_arg); dep <= (dep_arg == 2 ? 1 : dep_arg); ++dep) {
      const uint64_t lanes = lane_list[li];
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
    }
  return 0;
}
