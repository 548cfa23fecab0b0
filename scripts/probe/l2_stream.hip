// Diagnostic (GPU box): how fast can ONE workgroup per CU stream an L2-resident weight set (the packed copies of the
// step kernels: 1 KiB contiguous per wave-load) into registers or LDS, and does it matter that all CUs sweep the same
// addresses in the same order?   hipcc -O3 --offload-arch=gfx950 l2_stream.hip -o l2_stream && ./l2_stream
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

// MODE 0: registers, same order on every workgroup; 1: start offset rotated per workgroup; 2: LDS-DMA ring
template <int INFLIGHT, int MODE>
__global__ __launch_bounds__(1024) void k_stream(const f32x4* __restrict__ w, int frags_per_wave, int iters, float* out) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const f32x4* base = w + (size_t)wave * frags_per_wave * 64 + lane;
  const int off = MODE == 1 ? (int)((blockIdx.x * 37u) % (unsigned)frags_per_wave) : 0;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  if (MODE == 2) {
    float* ring = smem + wave * INFLIGHT * 256;    // INFLIGHT x 1 KiB per wave
    for (int it = 0; it < iters; ++it) {
      for (int f0 = 0; f0 < frags_per_wave; f0 += INFLIGHT) {
#pragma unroll
        for (int p = 0; p < INFLIGHT; ++p) {
          int f = f0 + p; f = f < frags_per_wave ? f : frags_per_wave - 1;
          __builtin_amdgcn_global_load_lds((const void*)(base + (size_t)f * 64 - lane + lane), (__attribute__((address_space(3))) void*)(ring + p * 256), 16, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
    }
    acc[0] = smem[threadIdx.x];
  } else {
    f32x4 ring[INFLIGHT];
#pragma unroll
    for (int p = 0; p < INFLIGHT; ++p) ring[p] = base[(size_t)((p + off) % frags_per_wave) * 64];
    const int total = frags_per_wave * iters;
    int f = INFLIGHT;
    for (; f + INFLIGHT <= total + INFLIGHT; f += INFLIGHT) {
#pragma unroll
      for (int p = 0; p < INFLIGHT; ++p) {
        const f32x4 v = ring[p];
        int nf = (f + p + off) % frags_per_wave;
        ring[p] = base[(size_t)nf * 64];
        __builtin_amdgcn_sched_barrier(0x0786);
        acc += v;
      }
    }
#pragma unroll
    for (int p = 0; p < INFLIGHT; ++p) acc += ring[p];
  }
  if (acc[0] + acc[1] + acc[2] + acc[3] == 123.456f) out[blockIdx.x * blockDim.x + threadIdx.x] = acc[0];
  (void)nw;
}

template <int INFLIGHT, int MODE>
static void run(const char* name, const f32x4* w, size_t bytes, int grid, int waves, float* out) {
  const int frags_per_wave = (int)(bytes / 1024 / waves);
  const int iters = 20;
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  const size_t lds = MODE == 2 ? (size_t)waves * INFLIGHT * 1024 : 0;
  if (lds > 64 * 1024) hipFuncSetAttribute((const void*)k_stream<INFLIGHT, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(a);
    hipLaunchKernelGGL((k_stream<INFLIGHT, MODE>), dim3(grid), dim3(64 * waves), lds, 0, w, frags_per_wave, iters, out);
    hipEventRecord(b);
    hipEventSynchronize(b);
  }
  float ms = 0.f;
  hipEventElapsedTime(&ms, a, b);
  const double per_cu = (double)frags_per_wave * waves * 1024.0 * iters / (ms * 1e-3);
  printf("%-34s grid %3d waves %2d inflight %2d: %7.3f ms  %6.1f GB/s per workgroup = %5.1f B/clk @2.4GHz, chip %6.2f TB/s\n",
         name, grid, waves, INFLIGHT, ms, per_cu / 1e9, per_cu / 2.4e9, per_cu * grid / 1e12);
}

int main() {
  const size_t bytes = 768 * 1024;           // ~ the weight bytes one phase-A tile pulls (G + D fwd + D dgrad packed)
  f32x4* w; float* out;
  hipMalloc(&w, bytes); hipMalloc(&out, 1 << 22);
  hipMemset(w, 0, bytes);
  for (int grid : {168, 256}) {
    for (int waves : {4, 8, 16}) {
      run<8, 0>("registers, same order", w, bytes, grid, waves, out);
      run<24, 0>("registers, same order", w, bytes, grid, waves, out);
      run<24, 1>("registers, rotated per workgroup", w, bytes, grid, waves, out);
      run<8, 2>("LDS-DMA ring", w, bytes, grid, waves, out);
    }
  }
  // a 16x larger buffer (beyond one XCD's L2 share but inside the Infinity Cache)
  f32x4* big; hipMalloc(&big, 16 * bytes); hipMemset(big, 0, 16 * bytes);
  run<24, 0>("registers, 12 MB buffer", big, 16 * bytes, 256, 4, out);
  return 0;
}
