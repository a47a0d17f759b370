// How fast can ONE output file in /dev/shm take 5 GB?  (r06: sizing the writer of `dicey hunt` on the GPU box's host.)
//   g++ -O2 -std=c++17 -pthread tools/write_bench.cpp -o /tmp/write_bench && /tmp/write_bench /dev/shm/wb.out 5000 64
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv) {
  const char* path = argc > 1 ? argv[1] : "/dev/shm/wb.out";
  const size_t mb = argc > 2 ? std::strtoull(argv[2], nullptr, 10) : 2000;
  const unsigned maxthr = argc > 3 ? std::atoi(argv[3]) : 8;
  const size_t total = mb << 20, blob = 64u << 20;
  std::string src(blob, 'x');
  for (size_t i = 0; i < blob; i += 61) src[i] = '\n';
  {  // (a) one write() stream
    unlink(path);
    int fd = open(path, O_WRONLY | O_CREAT | O_TRUNC, 0644);
    const double t0 = now();
    for (size_t o = 0; o < total; o += blob)
      if (write(fd, src.data(), blob) != (ssize_t)blob) return 1;
    close(fd);
    std::printf("one write stream          %6.2f GB/s\n", total / (now() - t0) / 1e9);
  }
  for (unsigned T = 2; T <= maxthr; T *= 2) {  // (b) parallel pwrite
    unlink(path);
    int fd = open(path, O_WRONLY | O_CREAT | O_TRUNC, 0644);
    const double t0 = now();
    std::vector<std::thread> th;
    const size_t nb = total / blob;
    for (unsigned t = 0; t < T; ++t)
      th.emplace_back([&, t]() {
        for (size_t b = t; b < nb; b += T)
          if (pwrite(fd, src.data(), blob, (off_t)(b * blob)) != (ssize_t)blob) std::abort();
      });
    for (auto& x : th) x.join();
    close(fd);
    std::printf("pwrite x%-3u               %6.2f GB/s\n", T, total / (now() - t0) / 1e9);
  }
  for (int populate = 0; populate < 2; ++populate)
    for (unsigned T = 1; T <= maxthr; T *= 2) {  // (c) shared mapping, threads copy (optionally MADV_POPULATE_WRITE first)
      unlink(path);
      int fd = open(path, O_RDWR | O_CREAT | O_TRUNC, 0644);
      const double t0 = now();
      if (ftruncate(fd, (off_t)total)) return 1;
      char* m = (char*)mmap(nullptr, total, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
      if (m == MAP_FAILED) return 1;
      std::vector<std::thread> th;
      const size_t nb = total / blob;
      for (unsigned t = 0; t < T; ++t)
        th.emplace_back([&, t]() {
          for (size_t b = t; b < nb; b += T) {
#ifdef MADV_POPULATE_WRITE
            if (populate) madvise(m + b * blob, blob, MADV_POPULATE_WRITE);
#endif
            std::memcpy(m + b * blob, src.data(), blob);
          }
        });
      for (auto& x : th) x.join();
      const double t1 = now();
      munmap(m, total);
      close(fd);
      std::printf("mmap copy x%-3u%s        %6.2f GB/s  (+ %.3f s munmap/close)\n", T, populate ? " populate" : "         ", total / (t1 - t0) / 1e9, now() - t1);
    }
  {  // (d) splice-free alternative: several files? no: ONE output. vmsplice from user pages into a pipe does not apply to a file.
  }
  unlink(path);
  return 0;
}
