#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) float f32x4;
__global__ void k(int* out) {  // out[LA][LB] = encoded (lane, reg) position or -1
  int lane = threadIdx.x;
  for (int LA = 0; LA < 64; ++LA) for (int LB = 0; LB < 64; ++LB) {
    float a = lane == LA ? 1.f : 0.f, b = lane == LB ? 1.f : 0.f;
    f32x4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) if (c[r] != 0.f) out[LA * 64 + LB] = lane * 4 + r;
  }
}
int main() {
  int* d; hipMalloc(&d, 4096 * 4); hipMemset(d, 0xff, 4096 * 4);
  k<<<1, 64>>>(d);
  int h[4096]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  // print for LA=0..63: which LBs hit
  for (int LA = 0; LA < 64; LA += 1) {
    printf("LA=%2d:", LA);
    for (int LB = 0; LB < 64; ++LB) if (h[LA * 64 + LB] >= 0) { int p = h[LA*64+LB]; int ln = p / 4, r = p % 4; printf(" LB%d->(row%d,col%d)", LB, (ln >> 4) * 4 + r, ln & 15); if (LB > 20 && LA > 2) break; }
    printf("\n");
    if (LA == 3) LA = 14; if (LA == 17) LA = 30; if (LA==33) LA = 46; if (LA == 49) break;
  }
  return 0;
}
