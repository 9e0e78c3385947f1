// largest dynamic LDS size a kernel launch accepts (gfx950): tools/dbg/ldsprobe
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(int* out) { extern __shared__ char s[]; s[threadIdx.x] = 1; __syncthreads(); if (threadIdx.x == 0) *out = s[1]; }
int main() {
  int* d; hipMalloc(&d, 4);
  for (int kb = 160; kb >= 120; kb -= 1) {
    for (int sub = 1023; sub >= 0; sub -= 256) {
      size_t bytes = (size_t)kb * 1024 - (1023 - sub);
      (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
      (void)hipGetLastError();
      hipLaunchKernelGGL(k, dim3(1), dim3(64), bytes, 0, d);
      hipError_t e = hipGetLastError();
      hipDeviceSynchronize();
      if (e == hipSuccess) { printf("largest accepted dynamic LDS: %zu bytes\n", bytes); return 0; }
    }
  }
  printf("none accepted\n");
  return 1;
}
