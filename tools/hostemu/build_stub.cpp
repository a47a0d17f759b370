// DEVELOPMENT TOOL: the GPU index builder is not emulated (rocPRIM, wave collectives).
#include <cstdint>
extern "C" {
int dg_index_build(const uint8_t*, uint64_t, int, const char*) { return -7; }
int dg_index_build_device(const void*, uint64_t, int, const char*) { return -7; }
}
struct hipStreamEmu;
namespace dg {
int derive_sa_by_sort(hipStreamEmu*, uint32_t*, uint64_t, uint32_t*) { return -4; }  // index.hip then walks the SA samples
}
