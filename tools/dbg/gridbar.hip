// Micro-benchmark: latency and correctness of a software grid barrier on MI355X (8 XCDs), 256 x 1024-thread workgroups.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/dbg/gridbar tools/dbg/gridbar.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ bool spin_until(unsigned* cnt, unsigned target) {
  // bounded spin: a lost workgroup must not hang the GPU
  for (int i = 0; i < (1 << 22); ++i) {
    const unsigned v = __hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if ((int)(v - target) >= 0) return true;
    __builtin_amdgcn_s_sleep(1);
  }
  return false;
}

// mode 0: flat counter; mode 1: per-XCD counter (blockIdx % 8) + top-level counter
template <int MODE>
__global__ __launch_bounds__(1024) void bar_kernel(unsigned* cnt, unsigned* slots, int nbar, int* errors, int check) {
  const int tid = threadIdx.x, b = blockIdx.x, G = gridDim.x;
  int err = 0;
  for (int it = 0; it < nbar; ++it) {
    if (check && tid == 0) slots[b] = (unsigned)(it + 1);
    // ---- barrier
    if (tid < 64) {   // control wave
      if (check) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      if (tid == 0) {
        bool ok;
        if (MODE == 0) {
          __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
          ok = spin_until(cnt, (unsigned)(it + 1) * G);
        } else {
          const int x = b & 7, nx = (G + 7 - x) / 8;   // workgroups with this residue
          const unsigned old = __hip_atomic_fetch_add(cnt + 16 * (1 + x), 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
          if (old + 1 == (unsigned)(it + 1) * nx) __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
          ok = spin_until(cnt, (unsigned)(it + 1) * 8);
        }
        if (!ok) err |= 2;
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    if (check && tid == 0) {
      const unsigned v = slots[(b + 37) % G];
      if (v != (unsigned)(it + 1) && v != (unsigned)(it + 2)) err |= 1;
    }
  }
  if (err && tid == 0) atomicOr(errors, err);
}

int main(int argc, char** argv) {
  const int G = argc > 1 ? atoi(argv[1]) : 256, nbar = argc > 2 ? atoi(argv[2]) : 2000;
  unsigned *cnt, *slots; int* errors;
  CK(hipMalloc(&cnt, 4096)); CK(hipMalloc(&slots, G * 4)); CK(hipMalloc(&errors, 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int mode = 0; mode < 2; ++mode)
    for (int check = 0; check < 2; ++check) {
      CK(hipMemset(cnt, 0, 4096)); CK(hipMemset(slots, 0, G * 4)); CK(hipMemset(errors, 0, 4));
      CK(hipEventRecord(e0));
      if (mode == 0) hipLaunchKernelGGL(bar_kernel<0>, dim3(G), dim3(1024), 0, 0, cnt, slots, nbar, errors, check);
      else hipLaunchKernelGGL(bar_kernel<1>, dim3(G), dim3(1024), 0, 0, cnt, slots, nbar, errors, check);
      CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      int h; CK(hipMemcpy(&h, errors, 4, hipMemcpyDeviceToHost));
      printf("mode %d check %d: %d barriers, %.3f us each, errors=%d\n", mode, check, nbar, ms * 1e3 / nbar, h);
    }
  return 0;
}
