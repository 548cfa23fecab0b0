// Probe: two processes on ONE GPU share an uncached allocation through hipIpc and hand-shake
// from inside concurrently running kernels (bounded spins). Prints what works.
#include <hip/hip_runtime.h>
#include <unistd.h>
#include <sys/wait.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cstdint>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("[%d] %s -> %s\n", getpid(), #x, hipGetErrorString(e_)); fflush(stdout); _exit(3); } } while (0)

__global__ void k_pingpong(uint32_t* mine, uint32_t* peer, int rounds, int* status, float* peer_data, float* my_data, float* out) {
  // each round: write data to the peer, release a flag there, wait for the peer's flag here, read its data
  const long long t0 = wall_clock64();
  float acc = 0.f;
  for (int r = 1; r <= rounds; ++r) {
    __hip_atomic_store(&peer_data[threadIdx.x + 64 * (r & 1)], (float)(r * 1000 + (int)threadIdx.x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(peer, (uint32_t)r, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    if (threadIdx.x == 0) {
      while ((int32_t)(__hip_atomic_load(mine, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - (uint32_t)r) < 0) {
        if (wall_clock64() - t0 > 300000000LL) { *status = r; break; }   // 3 s at 100 MHz
      }
    }
    __syncthreads();
    __threadfence_system();
    if (*status) return;
    const float v = __hip_atomic_load(&my_data[threadIdx.x + 64 * (r & 1)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (v != (float)(r * 1000 + (int)threadIdx.x)) { if (threadIdx.x == 0) *status = -r; }
    acc += v;
  }
  out[threadIdx.x] = acc;
  if (threadIdx.x == 0) out[64] = (float)(wall_clock64() - t0);
}

int main() {
  int p2c[2], c2p[2];
  if (pipe(p2c) || pipe(c2p)) return 1;
  const pid_t child = fork();          // before any HIP call
  const int me = child == 0 ? 1 : 0;
  const int rd = me ? p2c[0] : c2p[0], wr = me ? c2p[1] : p2c[1];
  CK(hipSetDevice(0));
  const size_t bytes = 1 << 20;
  void* region = nullptr;
  CK(hipExtMallocWithFlags(&region, bytes, hipDeviceMallocUncached));
  CK(hipMemset(region, 0, bytes));
  CK(hipDeviceSynchronize());
  hipIpcMemHandle_t h, hp;
  CK(hipIpcGetMemHandle(&h, region));
  if (write(wr, &h, sizeof(h)) != (ssize_t)sizeof(h)) return 2;
  if (read(rd, &hp, sizeof(hp)) != (ssize_t)sizeof(hp)) return 2;
  void* peer = nullptr;
  CK(hipIpcOpenMemHandle(&peer, hp, hipIpcMemLazyEnablePeerAccess));
  printf("[rank %d] uncached alloc + ipc export/open ok (local %p, peer %p)\n", me, region, peer); fflush(stdout);
  int* status; float* out;
  CK(hipMalloc(&status, 4)); CK(hipMemset(status, 0, 4));
  CK(hipMalloc(&out, 65 * 4));
  // handshake so both have opened before kernels start
  char c = 'x';
  if (write(wr, &c, 1) != 1 || read(rd, &c, 1) != 1) return 2;
  uint32_t* myflag = (uint32_t*)region;            float* mydata = (float*)region + 1024;
  uint32_t* peerflag = (uint32_t*)peer;            float* peerdata = (float*)peer + 1024;
  const int rounds = 2000;
  hipLaunchKernelGGL(k_pingpong, dim3(1), dim3(64), 0, 0, myflag, peerflag, rounds, status, peerdata, mydata, out);
  CK(hipDeviceSynchronize());
  int st; float o[65];
  CK(hipMemcpy(&st, status, 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(o, out, 65 * 4, hipMemcpyDeviceToHost));
  printf("[rank %d] ping-pong status %d (0 = ok, >0 timeout round, <0 stale data round); %d rounds in %.1f us -> %.2f us/round\n",
         me, st, rounds, o[64] / 100.0, o[64] / 100.0 / rounds);
  fflush(stdout);
  if (write(wr, &c, 1) != 1 || read(rd, &c, 1) != 1) return 2;
  CK(hipIpcCloseMemHandle(peer));
  CK(hipFree(region));
  if (me == 0) { int ws; waitpid(child, &ws, 0); }
  return st != 0;
}
