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

