// The object behind the opaque dg_thal handle (thal_api.hip owns it; hunt.hip reads the tables for `dicey search`).
#pragma once
#include "common.hpp"
#include "thal.hpp"

struct dg_thal {
  int device = 0;
  hipStream_t stream = nullptr;
  dg::thal::Tables host_tables;
  dg::thal::Tables* d_tables = nullptr;
  dg::thal::Env env;
  dg::DevBuf ws[6];
  ~dg_thal() {
    if (d_tables) (void)hipFree(d_tables);
    for (auto& w : ws) w.release();
    if (stream) (void)hipStreamDestroy(stream);
  }
};


namespace dg {
constexpr uint32_t kSelfWindowMax = 48;  // longest window of thal_self_windows (two tables per workgroup still fit LDS)
// thal_api.hip: thal(window, reverse complement of the window) for many windows of one byte buffer
int thal_self_windows(dg_thal* th, const uint8_t* bytes, uint64_t nbytes, const uint64_t* win_off, const uint32_t* win_len, size_t n,
                      double* temp, uint32_t uniform_len = 0);
}  // namespace dg
