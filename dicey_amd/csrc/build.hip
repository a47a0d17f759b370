// placeholder until the GPU builder lands (next commit)
#include "index_internal.hpp"
extern "C" {
int dg_index_build(const uint8_t*, uint64_t, int, const char*) { return dg::fail(DG_ELIMIT, "dg_index_build: not implemented yet"); }
int dg_index_build_device(const void*, uint64_t, int, const char*) { return dg::fail(DG_ELIMIT, "dg_index_build_device: not implemented yet"); }
}
