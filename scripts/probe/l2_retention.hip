// Diagnostic (GPU box): does data WRITTEN by a kernel's blocks stay readable from the writing XCD's L2 by the next
// kernel?  Kernel W: block b writes its 96 KB region.  Kernel R: block b reads region (b + shift) % nblocks.
// shift 0 = same XCD as the writer (blocks b of both launches run on XCD b % 8), shift 1 = another XCD.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int kRegion = 96 * 1024 / 16;   // float4 per region
__global__ __launch_bounds__(256) void kw(f32x4* buf, float v) {
  f32x4* p = buf + (size_t)blockIdx.x * kRegion;
  for (int i = threadIdx.x; i < kRegion; i += 256) p[i] = f32x4{v, v, v, v};
}
__global__ __launch_bounds__(256) void kr(const f32x4* buf, int shift, float* out) {
  const f32x4* p = buf + (size_t)((blockIdx.x + shift) % gridDim.x) * kRegion;
  f32x4 acc = {0, 0, 0, 0};
  f32x4 r[24];
#pragma unroll
  for (int u = 0; u < 24; ++u) r[u] = p[threadIdx.x + 256 * u];     // 24 * 4 KB = 96 KB, all loads in flight
#pragma unroll
  for (int u = 0; u < 24; ++u) acc += r[u];
  if (acc[0] == 123.f) out[blockIdx.x] = acc[1];
}
int main() {
  const int nb = 256;                       // 24 MB in all, like the 25 MB of activations of the config-2 step
  f32x4* buf; float* out;
  hipMalloc(&buf, (size_t)nb * kRegion * 16); hipMalloc(&out, 4096);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int shift : {0, 1, 0, 1, 8, 3}) {
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
      hipLaunchKernelGGL(kw, dim3(nb), dim3(256), 0, 0, buf, (float)rep);
      hipEventRecord(a);
      hipLaunchKernelGGL(kr, dim3(nb), dim3(256), 0, 0, buf, shift, out);
      hipEventRecord(b); hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b);
      best = ms < best ? ms : best;
    }
    printf("reader shift %d (%s XCD as the writer): read kernel %.2f us  = %.2f TB/s\n", shift,
           shift % 8 == 0 ? "same" : "another", best * 1e3, nb * 96.0 * 1024 / (best * 1e-3) / 1e12);
  }
  return 0;
}
