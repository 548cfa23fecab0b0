// Diagnostic (GPU box): which XCD does block b of consecutive launches land on?  Launches grids of the step's sizes in
// the step's order and prints HW_REG_XCC_ID of the first 16 blocks of each, three rounds.
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(int* out, int n) {
  if (threadIdx.x == 0 && (int)blockIdx.x < n) out[blockIdx.x] = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 15;
  // a little work so that launches do not all finish instantly
  for (volatile int i = 0; i < 2000; ++i) {}
}
int main() {
  const int grids[6] = {252, 619, 230, 168, 634, 329};   // phase A (PAIR), wgrad[D]+NDiv, reduce[D], phase B, wgrad[G], reduce[G]
  int* d; hipMalloc(&d, 6 * 1024 * sizeof(int));
  static int h[6 * 1024];
  for (int round = 0; round < 3; ++round) {
    for (int g = 0; g < 6; ++g) hipLaunchKernelGGL(k, dim3(grids[g]), dim3(256), 0, 0, d + g * 1024, 1024);
    hipDeviceSynchronize();
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int g = 0; g < 6; ++g) {
      printf("round %d grid %3d: ", round, grids[g]);
      for (int b = 0; b < 16; ++b) printf("%d ", h[g * 1024 + b]);
      int ok = 1;
      for (int b = 8; b < grids[g]; ++b) ok &= (h[g * 1024 + b] == h[g * 1024 + b - 8]);
      printf(" | b and b+8 always share: %s\n", ok ? "yes" : "NO");
    }
  }
  return 0;
}
