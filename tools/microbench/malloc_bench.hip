// How long does hipMalloc take as a function of the size?  (the 137 GB K-mer table showed 2-3 s)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  (void)hipFree(nullptr);
  // stream-ordered pool, fresh process state: five 12 GiB blocks, then the big one
  if (std::getenv("ASYNC_FIRST")) {
    hipStream_t st;
    (void)hipStreamCreate(&st);
    void* q[5];
    for (int i = 0; i < 5; ++i) {
      double t0 = now();
      hipError_t e = hipMallocAsync(&q[i], 12ull << 30, st);
      (void)hipStreamSynchronize(st);
      printf("hipMallocAsync 12 GiB #%d: %s %8.1f ms\n", i, e == hipSuccess ? "ok" : "failed", (now() - t0) * 1e3);
    }
    if (std::getenv("FREE_SOME")) {
      double t0 = now();
      for (int i = 0; i < 3; ++i) (void)hipFreeAsync(q[i], st);
      (void)hipStreamSynchronize(st);
      printf("hipFreeAsync 3 x 12 GiB: %8.1f ms\n", (now() - t0) * 1e3);
    }
    if (std::getenv("PLAIN_TOO")) {
      void* r = nullptr;
      double t0 = now();
      (void)hipMalloc(&r, 2ull << 30);
      printf("hipMalloc 2 GiB: %8.1f ms\n", (now() - t0) * 1e3);
      t0 = now();
      (void)hipFree(r);
      printf("hipFree 2 GiB: %8.1f ms\n", (now() - t0) * 1e3);
    }
    void* p = nullptr;
    double t0 = now();
    hipError_t e = hipMallocAsync(&p, 128ull << 30, st);
    (void)hipStreamSynchronize(st);
    printf("hipMallocAsync 128 GiB: %s %8.1f ms\n", e == hipSuccess ? "ok" : "failed", (now() - t0) * 1e3);
    t0 = now();
    (void)hipMemsetAsync(p, 0, 128ull << 30, st);
    (void)hipStreamSynchronize(st);
    printf("memset 128 GiB: %8.1f ms\n", (now() - t0) * 1e3);
    return 0;
  }
  const double sizes[] = {1, 8, 32, 64, 137, 137};
  for (double gb : sizes) {
    size_t b = (size_t)(gb * (1ull << 30));
    void* p = nullptr;
    double t0 = now();
    hipError_t e = hipMalloc(&p, b);
    double t1 = now();
    if (e != hipSuccess) { printf("hipMalloc %.0f GiB failed\n", gb); continue; }
    (void)hipMemset(p, 0, b);
    (void)hipDeviceSynchronize();
    double t2 = now();
    (void)hipFree(p);
    double t3 = now();
    printf("hipMalloc %5.0f GiB: malloc %8.1f ms  memset %8.1f ms  free %8.1f ms\n", gb, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3);
  }
  // many chunks instead of one block
  {
    const int N = 137;
    void* ps[N];
    double t0 = now();
    for (int i = 0; i < N; ++i) (void)hipMalloc(&ps[i], 1ull << 30);
    double t1 = now();
    for (int i = 0; i < N; ++i) (void)hipFree(ps[i]);
    printf("137 x 1 GiB: malloc %8.1f ms free %8.1f ms\n", (t1 - t0) * 1e3, (now() - t1) * 1e3);
  }
  // stream-ordered pool
  {
    void* p = nullptr;
    hipStream_t st;
    (void)hipStreamCreate(&st);
    double t0 = now();
    hipError_t e = hipMallocAsync(&p, 137ull << 30, st);
    (void)hipStreamSynchronize(st);
    double t1 = now();
    printf("hipMallocAsync 137 GiB: %s %8.1f ms\n", e == hipSuccess ? "ok" : "failed", (t1 - t0) * 1e3);
    if (e == hipSuccess) { (void)hipFreeAsync(p, st); (void)hipStreamSynchronize(st); }
    t0 = now();
    e = hipMallocAsync(&p, 137ull << 30, st);
    (void)hipStreamSynchronize(st);
    printf("hipMallocAsync 137 GiB again: %s %8.1f ms\n", e == hipSuccess ? "ok" : "failed", (now() - t0) * 1e3);
  }
  return 0;
}
