// Batched equivalents of the three sdsl calls `dicey hunt`/`search` make into the FM-index:
//   sdsl::count   (reference src/hunter.h:353, src/silica.h:365,367,392,394,470)
//   sdsl::locate  (src/hunter.h:355 + the std::sort at :356; src/silica.h:472-473)
//   sdsl::extract (src/hunter.h:371, src/silica.h:490)
// One lane per pattern / occurrence / range; arbitrary byte patterns (symbols outside A,C,G,T go through the
// wavelet tree exactly as sdsl would).
#include <algorithm>

#include "devfm.hpp"
#include "index_internal.hpp"

namespace dg {

__global__ void k_count(FmView f, const u8* pat, const u64* off, u64 npat, u32* lo_out, u32* hi_out) {
  u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= npat) return;
  u64 b = off[t], e = off[t + 1];
  u32 lo = 0, hi = (u32)f.n;
  if (e - b > f.n) hi = 0;  // sdsl::count: pattern longer than the text
  while (e > b && lo < hi) {
    u32 sym = pat[--e];
    bs_extend_sym(f, lo, hi, sym, code_of_byte(sym));
  }
  if (hi < lo) hi = lo;
  lo_out[t] = lo;
  hi_out[t] = hi;
}

__global__ void k_copy_sa(FmView f, const u32* lo, const u64* out_off, u64 npat, u64* out) {
  u64 t = blockIdx.x;  // one workgroup per pattern, threads stride over its occurrences (coalesced SA reads)
  if (t >= npat) return;
  u64 k = out_off[t + 1] - out_off[t];
  for (u64 j = threadIdx.x; j < k; j += blockDim.x) out[out_off[t] + j] = f.sa[(u64)lo[t] + j];
}

__global__ void k_extract(FmView f, const u64* lo, const u64* hi, const u64* out_off, u64 nr, u8* out) {
  u64 t = blockIdx.x;
  if (t >= nr) return;
  u64 k = hi[t] - lo[t] + 1;
  for (u64 j = threadIdx.x; j < k; j += blockDim.x) out[out_off[t] + j] = f.text[lo[t] + j];
}

static int count_intervals(dg_index* ix, const u8* pat, const u64* off, size_t npat, std::vector<u32>& lo, std::vector<u32>& hi) {
  DG_HIP(hipSetDevice(ix->device));
  u64 total = off[npat];
  DG_TRY(ix->ws[0].reserve(total + 8));
  DG_TRY(ix->ws[1].reserve((npat + 1) * 8));
  DG_TRY(ix->ws[2].reserve(npat * 4 + 4));
  DG_TRY(ix->ws[3].reserve(npat * 4 + 4));
  DG_HIP(hipMemcpyAsync(ix->ws[0].p, pat, total, hipMemcpyHostToDevice, ix->stream));
  DG_HIP(hipMemcpyAsync(ix->ws[1].p, off, (npat + 1) * 8, hipMemcpyHostToDevice, ix->stream));
  hipLaunchKernelGGL(k_count, dim3(ceil_div(npat, 256)), dim3(256), 0, ix->stream, ix->view, ix->ws[0].as<u8>(),
                     ix->ws[1].as<u64>(), (u64)npat, ix->ws[2].as<u32>(), ix->ws[3].as<u32>());
  lo.resize(npat);
  hi.resize(npat);
  DG_HIP(hipMemcpyAsync(lo.data(), ix->ws[2].p, npat * 4, hipMemcpyDeviceToHost, ix->stream));
  DG_HIP(hipMemcpyAsync(hi.data(), ix->ws[3].p, npat * 4, hipMemcpyDeviceToHost, ix->stream));
  DG_HIP(hipStreamSynchronize(ix->stream));
  DG_HIP(hipGetLastError());
  return DG_OK;
}

}  // namespace dg

using namespace dg;

extern "C" {

int dg_count(dg_index* ix, const uint8_t* pat, const uint64_t* off, size_t npat, uint64_t* counts) {
  if (!ix || !off || !counts || (!pat && off[npat])) return fail(DG_EINVAL, "dg_count: null argument");
  if (!npat) return DG_OK;
  std::vector<u32> lo, hi;
  DG_TRY(count_intervals(ix, pat, off, npat, lo, hi));
  for (size_t i = 0; i < npat; ++i) counts[i] = hi[i] - lo[i];
  return DG_OK;
}

int dg_locate(dg_index* ix, const uint8_t* pat, const uint64_t* off, size_t npat, dg_locations** out) {
  if (!ix || !off || !out || (!pat && off[npat])) return fail(DG_EINVAL, "dg_locate: null argument");
  dg_locations* L = new dg_locations;
  L->npat = npat;
  L->off = new uint64_t[npat + 1];
  L->off[0] = 0;
  L->pos = nullptr;
  *out = L;
  if (!npat) return DG_OK;
  std::vector<u32> lo, hi;
  int rc = count_intervals(ix, pat, off, npat, lo, hi);
  if (rc != DG_OK) {
    dg_locations_free(L);
    *out = nullptr;
    return rc;
  }
  for (size_t i = 0; i < npat; ++i) L->off[i + 1] = L->off[i] + (hi[i] - lo[i]);
  u64 total = L->off[npat];
  L->pos = new uint64_t[total ? total : 1];
  if (total) {
    auto body = [&]() -> int {
      DG_TRY(ix->ws[4].reserve((npat + 1) * 8));
      DG_TRY(ix->ws[5].reserve(total * 8));
      DG_HIP(hipMemcpyAsync(ix->ws[4].p, L->off, (npat + 1) * 8, hipMemcpyHostToDevice, ix->stream));
      hipLaunchKernelGGL(k_copy_sa, dim3((u32)npat), dim3(64), 0, ix->stream, ix->view, ix->ws[2].as<u32>(), ix->ws[4].as<u64>(),
                         (u64)npat, ix->ws[5].as<u64>());
      DG_HIP(hipMemcpyAsync(L->pos, ix->ws[5].p, total * 8, hipMemcpyDeviceToHost, ix->stream));
      DG_HIP(hipStreamSynchronize(ix->stream));
      DG_HIP(hipGetLastError());
      return DG_OK;
    };
    rc = body();
    if (rc != DG_OK) {
      dg_locations_free(L);
      *out = nullptr;
      return rc;
    }
    for (size_t i = 0; i < npat; ++i) std::sort(L->pos + L->off[i], L->pos + L->off[i + 1]);  // hunter.h:356
  }
  return DG_OK;
}

void dg_locations_free(dg_locations* l) {
  if (!l) return;
  delete[] l->off;
  delete[] l->pos;
  delete l;
}

int dg_extract(dg_index* ix, const uint64_t* lo, const uint64_t* hi, size_t nr, uint8_t* out, const uint64_t* out_off) {
  if (!ix || !lo || !hi || !out || !out_off) return fail(DG_EINVAL, "dg_extract: null argument");
  if (!nr) return DG_OK;
  u64 total = 0;
  for (size_t i = 0; i < nr; ++i) {
    if (hi[i] < lo[i] || hi[i] >= ix->view.n) return fail(DG_EINVAL, "dg_extract: range %zu out of bounds", i);
    if (out_off[i] != total) return fail(DG_EINVAL, "dg_extract: out_off must be the running sum of range lengths");
    total += hi[i] - lo[i] + 1;
  }
  DG_HIP(hipSetDevice(ix->device));
  DG_TRY(ix->ws[0].reserve(nr * 8));
  DG_TRY(ix->ws[1].reserve(nr * 8));
  DG_TRY(ix->ws[4].reserve(nr * 8));
  DG_TRY(ix->ws[5].reserve(total));
  DG_HIP(hipMemcpyAsync(ix->ws[0].p, lo, nr * 8, hipMemcpyHostToDevice, ix->stream));
  DG_HIP(hipMemcpyAsync(ix->ws[1].p, hi, nr * 8, hipMemcpyHostToDevice, ix->stream));
  DG_HIP(hipMemcpyAsync(ix->ws[4].p, out_off, nr * 8, hipMemcpyHostToDevice, ix->stream));
  hipLaunchKernelGGL(k_extract, dim3((u32)nr), dim3(64), 0, ix->stream, ix->view, ix->ws[0].as<u64>(), ix->ws[1].as<u64>(),
                     ix->ws[4].as<u64>(), (u64)nr, ix->ws[5].as<u8>());
  DG_HIP(hipMemcpyAsync(out, ix->ws[5].p, total, hipMemcpyDeviceToHost, ix->stream));
  DG_HIP(hipStreamSynchronize(ix->stream));
  DG_HIP(hipGetLastError());
  return DG_OK;
}

}  // extern "C"
