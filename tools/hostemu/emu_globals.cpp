// DEVELOPMENT TOOL — see hip/hip_runtime.h in this directory.
#include <hip/hip_runtime.h>
thread_local uint3 threadIdx, blockIdx;
thread_local dim3 blockDim, gridDim;
