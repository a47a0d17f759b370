// DEVELOPMENT TOOL — see hip/hip_runtime.h in this directory.
#include <hip/hip_runtime.h>
thread_local uint3 threadIdx, blockIdx;
thread_local dim3 blockDim, gridDim;

#include <cstdlib>
// Wave-cooperative kernels cannot run with serial lanes: make the library choose its lane-per-item kernels.
namespace {
struct EmuEnv {
  EmuEnv() {
    setenv("DICEY_NO_WAVE_THAL", "1", 0);
    setenv("DICEY_NO_BLOCK_SCAN", "1", 0);
  }
} emu_env;
}  // namespace
