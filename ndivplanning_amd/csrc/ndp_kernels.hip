// ndp_kernels.hip -- gfx950 kernels of the GAN train step (see include/ndp.h).
//
//   k_g_fwd      Decoder.forward, one workgroup per 16*RT-row tile, all five layers fused
//   k_d          Discriminator forward (+ BCE + backward data path), fused per row tile
//   k_g_bwd      backward data path of the Decoder per row tile
//   k_wgrad      all weight/bias gradients of one network: dW = dY^T X as MFMA blocks,
//                rows split into chunks (split-K) -> partial slabs
//   k_reduce_adam  sum the slabs (fixed order: bitwise reproducible), Adam, loss scalars
//   k_ndiv       normalized-diversification loss + gradient (diversity.py)
//   k_philox     uniform noise
#include <hip/hip_runtime.h>
#include <math.h>
#include <stddef.h>
#include <stdint.h>

#include "ndp_device.h"

namespace ndp {

// Diagnostic build only (-DNDP_STAMPS, scripts/diag_stamps.py): thread 0 of every workgroup
// records {shader clock, 100 MHz wall clock} at phase boundaries into a side buffer that no
// kernel reads.  The shipped library contains no stamp.
#ifdef NDP_STAMPS
__device__ unsigned long long* g_stamps = nullptr;
__device__ int g_stamp_kernel = 0;   // which kernel flushes: 1 k_g_fwd, 2 k_d, 3 k_g_bwd, 4 k_wgrad (0 = any)
#define NDP_STAMP_ON(id) (g_stamps != nullptr && (g_stamp_kernel == 0 || g_stamp_kernel == (id)))
// stamps are kept in LDS (a global store per stamp would sit in the wave's vmcnt queue and
// delay the next counted wait) and flushed by NDP_STAMP_FLUSH at the end of the kernel
#define NDP_STAMP_DECL __shared__ unsigned long long stamp_store_[64]; unsigned long long* stamp_lds_ = stamp_store_
#define NDP_STAMP_PTR stamp_lds_
#define NDP_STAMP(i)                                             \
  do {                                                           \
    if (threadIdx.x == 0) {                                      \
      stamp_lds_[2 * (i)] = clock64();                           \
      stamp_lds_[2 * (i) + 1] = wall_clock64();                  \
    }                                                            \
  } while (0)
#define NDP_STAMP_FLUSH(n, id)                                                              \
  do {                                                                                      \
    if (threadIdx.x == 0 && NDP_STAMP_ON(id))                                               \
      for (int i_ = 0; i_ < 2 * (n); ++i_)                                                  \
        g_stamps[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 64 + i_] = stamp_lds_[i_]; \
  } while (0)
#else
#define NDP_STAMP_DECL
#define NDP_STAMP_PTR nullptr
#define NDP_STAMP(i) do { } while (0)
#define NDP_STAMP_FLUSH(n, id) do { } while (0)
#endif

// -DNDP_PLAIN_KERNARGS: ablation of load_kernargs (ndp_device.h) in the phase kernels (27.5 vs 27.9 us for phase A)
#ifdef NDP_PLAIN_KERNARGS
constexpr bool kFastKernargs = false;
#else
constexpr bool kFastKernargs = true;
#endif
constexpr int CODE = 256;
constexpr int ADIM = 4;
constexpr int TAILLD = 16;   // LDS row stride of the narrow "tail" inputs (noise / action)

struct GNet {
  const float *w1, *b1, *w2, *b2, *w3, *b3, *w4, *b4, *w5, *b5;
  int ld1, nz;
  // lane-ordered copies (PACKED kernels): forward fc1(main 256 cols)..fc4, data-gradient fc2..fc4
  const float *pf1, *pf2, *pf3, *pf4, *pg2, *pg3, *pg4;
};
struct DNet {
  const float *w1, *b1, *w2, *b2, *w3, *b3, *w4, *b4;
  const float *pf1, *pf2, *pf3, *pg2, *pg3;   // forward fc1(main)..fc3, data-gradient fc2, fc3
};

// Where a network's packed copies live and how canonical parameter indices map into them.
struct PackLayer {
  int w_off;      // canonical offset of the weight [out][ld]
  int ld, out;    // row stride (= in features) and rows
  int main0;      // first "main" column (0; 4 for D.fc1 whose columns 0..3 are the action tail)
  int main_in;    // number of main columns (256 for the fc1 layers, else ld)
  int fwd_off;    // offset of the forward-packed copy in the packed buffer, or -1
  int dg_off;     // offset of the data-gradient-packed copy, or -1
};
struct PackSpec {
  PackLayer L[4];
  int nl;
  float* packed;  // null: no packed copy is maintained
};

__device__ __forceinline__ void pack_store(const PackSpec& ps, int p, float v) {
  // (no early exits: the loop unrolls completely and every index into ps.L is static -- a by-value copy of the
  // arguments then stays in registers)
#pragma unroll
  for (int l = 0; l < 4; ++l) {
    const PackLayer L = ps.L[l];
    const int rel = p - L.w_off;
    if (l < ps.nl && rel >= 0 && rel < L.out * L.ld) {      // the layers' ranges are disjoint: at most one matches
      const int j = rel / L.ld, k = rel % L.ld;
      const int km = k - L.main0;
      if (L.fwd_off >= 0 && km >= 0 && km < L.main_in)
        ps.packed[L.fwd_off + fwd_pack_offset(j, km, L.main_in, L.out)] = v;
      if (L.dg_off >= 0) ps.packed[L.dg_off + dgrad_pack_offset(j, k, L.ld, L.out)] = v;
    }
  }
}

struct PackArgs { const float* params; int n; PackSpec ps; };
__global__ __launch_bounds__(kThreads) void k_pack(PackArgs a) {
  const int p = blockIdx.x * kThreads + threadIdx.x;
  if (p < a.n) pack_store(a.ps, p, a.params[p]);
}

// ================================================================ uniform noise (Philox-4x32-10)
__device__ __forceinline__ void philox_round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
  const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
  const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
  const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
  const uint32_t n1 = (uint32_t)p1;
  const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
  const uint32_t n3 = (uint32_t)p0;
  c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}


// element `e` of the noise stream (seed, offset): Philox-4x32-10 block e/4, word e%4 -> U[0,1)
__device__ __forceinline__ float philox_uniform(uint64_t e, uint64_t seed, uint32_t off) {
  const uint64_t blk = e >> 2;
  uint32_t c[4] = {(uint32_t)blk, (uint32_t)(blk >> 32), off, 0x6e647021u};
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    philox_round(c, k0, k1);
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  const uint32_t w = (e & 3) == 0 ? c[0] : (e & 3) == 1 ? c[1] : (e & 3) == 2 ? c[2] : c[3];
  return (float)(w >> 8) * (1.0f / 16777216.0f);
}

// ================================================================ G forward
struct GFwdArgs {
  GNet net;
  const float* code; int64_t ld_code; int code_rep; int code_vec4;
  const float* noise; int64_t ld_noise;
  int64_t m;
  float *h1, *h2, *h3, *h4;   // [mpad x 128/64/128/256] or all null
  float* action_hat;          // [m x 4]
  // device noise: when noise_out != null the kernel draws U[0,1) itself (stream = seed,
  // offset *noise_step), uses it and writes it to noise_out[m x nz] for the later kernels
  float* noise_out; uint64_t noise_seed; const int32_t* noise_step;
};

template <int RT>
constexpr int g_fwd_lds_floats() { return 16 * RT * (260 + TAILLD + 132 + 68 + 132 + 260 + 4); }

template <int RT, int W1ALIGN, bool PK>
__global__ __launch_bounds__(kThreads) void k_g_fwd(GFwdArgs a) {
  constexpr int R = 16 * RT;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Xc = smem;                 // R x 260
  float* Xt = Xc + R * 260;         // R x 16   noise
  float* H1 = Xt + R * TAILLD;      // R x 132
  float* H2 = H1 + R * 132;         // R x 68
  float* H3 = H2 + R * 68;          // R x 132
  float* H4 = H3 + R * 132;         // R x 260
  float* A = H4 + R * 260;          // R x 4
  const int64_t row0 = (int64_t)blockIdx.x * R;
  const GNet& n = a.net;
  NDP_STAMP_DECL;
  NDP_STAMP(0);

  // each layer's first weight loads are issued one stage early (they fly across the barrier)
  FwdW<256, 128, W1ALIGN, PK> w1;
  w1.preload(PK ? n.pf1 : n.w1, n.ld1, n.b1, n.w1 + CODE, n.nz);
  load_code_tile<RT>(Xc, 260, a.code, a.ld_code, a.code_rep, row0, a.m, a.code_vec4 != 0);
  for (int idx = threadIdx.x; idx < R * TAILLD; idx += kThreads) {
    const int i = idx / TAILLD, t = idx % TAILLD;
    const int64_t row = row0 + i;
    float v = 0.f;
    if (row < a.m && t < n.nz) {
      if (a.noise_out != nullptr) {
        v = philox_uniform((uint64_t)(row * n.nz + t), a.noise_seed, (uint32_t)*a.noise_step);
        a.noise_out[row * n.nz + t] = v;
      } else {
        v = a.noise[row * a.ld_noise + t];
      }
    }
    Xt[idx] = v;
  }
  __syncthreads();
  NDP_STAMP(1);
  // every layer's first weight fragments are issued inside the previous layer's k-loop (FwdW::preload_slice)
  FwdW<128, 64, 4, PK> w2;
  w2.bind(PK ? n.pf2 : n.w2, 128, n.b2, nullptr, 0);
  layer_fwd_run<RT, 256, 128, ACT_RELU, W1ALIGN, PK>(w1, Xc, 260, H1, 132, Xt, TAILLD, w2);
  __syncthreads();
  NDP_STAMP(2);
  FwdW<64, 128, 4, PK> w3;
  w3.bind(PK ? n.pf3 : n.w3, 64, n.b3, nullptr, 0);
  layer_fwd_run<RT, 128, 64, ACT_RELU, 4, PK>(w2, H1, 132, H2, 68, nullptr, 0, w3);
  __syncthreads();
  NDP_STAMP(3);
  FwdW<128, 256, 4, PK> w4;
  w4.bind(PK ? n.pf4 : n.w4, 128, n.b4, nullptr, 0);
  layer_fwd_run<RT, 64, 128, ACT_RELU, 4, PK>(w3, H2, 68, H3, 132, nullptr, 0, w4);
  __syncthreads();
  NDP_STAMP(4);
  layer_fwd_run<RT, 128, 256, ACT_RELU, 4, PK>(w4, H3, 132, H4, 260, nullptr, 0);
  __syncthreads();
  NDP_STAMP(5);
  layer_fwd_narrow<RT, 256, 4>(H4, 260, n.w5, n.b5, A, 4);
  __syncthreads();
  NDP_STAMP(6);
  if (a.h1 != nullptr) {
    store_tile<RT, 128>(a.h1 + row0 * 128, 128, H1, 132);
    store_tile<RT, 64>(a.h2 + row0 * 64, 64, H2, 68);
    store_tile<RT, 128>(a.h3 + row0 * 128, 128, H3, 132);
    store_tile<RT, 256>(a.h4 + row0 * 256, 256, H4, 260);
  }
  if (threadIdx.x < R) {
    const int64_t row = row0 + threadIdx.x;
    if (row < a.m)
      *reinterpret_cast<f32x4*>(a.action_hat + row * 4) = *reinterpret_cast<const f32x4*>(A + threadIdx.x * 4);
  }
  NDP_STAMP(7);
  NDP_STAMP_FLUSH(8, 1);
}

// ================================================================ NDiv (diversity.py)
// One thread per (row n, sample i); a workgroup holds G = 256/K rows in LDS.
// Pass 1: row sums s_i = sum_j d_ij for x and z.  Pass 2: hinge terms and the gradient
//   dL/dx_i = sum_j -(m_ij/s_i + m_ji/s_j) (x_i - x_j)/d_ij   (0 where d_ij == 0).
struct NdivArgs {
  const float* x; int cx; const float* z; int cz;
  int64_t n; int k; int rows_per_block;
  float grad_scale;
  float* grad;         // [n*k x cx] or null
  float* partials;     // [gridDim.x]
};
constexpr int kNdivMaxC = 16;

// MX / MZ: compile-time bounds of the channel loops (4 / 2 for the training path: actions
// and <= 2-d noise; 16 / 16 generic).  Stand-alone launches use blockDim.x = 64 for k <= 64
// (one wave holds 64/k rows) else 256; fused into k_d's launch the block has 256 threads.
template <int MX, int MZ>
__device__ __forceinline__ void ndiv_block(const NdivArgs& a, int bidx, float* smem) {
  const int k = a.k, cx = a.cx, cz = a.cz, G = a.rows_per_block;
  float* xs = smem;                     // G*k*cx
  float* zs = xs + G * k * cx;          // G*k*cz
  float* sx = zs + G * k * cz;          // G*k
  float* sz = sx + G * k;               // G*k
  float* red = sz + G * k;              // 4
  const int nthreads = blockDim.x;
  const int64_t n0 = (int64_t)bidx * G;
  const int64_t nrows = (a.n - n0) < G ? (a.n - n0) : G;
  const int nact = (int)nrows * k;
  for (int idx = threadIdx.x; idx < nact * cx; idx += nthreads) xs[idx] = a.x[n0 * k * cx + idx];
  for (int idx = threadIdx.x; idx < nact * cz; idx += nthreads) zs[idx] = a.z[n0 * k * cz + idx];
  __syncthreads();
  const int t = threadIdx.x;
  const bool on = t < nact;
  const int g = on ? t / k : 0, i = on ? t % k : 0;
  float xi[MX], zi[MZ];
#pragma unroll
  for (int d = 0; d < MX; ++d) xi[d] = (on && d < cx) ? xs[(g * k + i) * cx + d] : 0.f;
#pragma unroll
  for (int d = 0; d < MZ; ++d) zi[d] = (on && d < cz) ? zs[(g * k + i) * cz + d] : 0.f;
  if (on) {
    float ssx = 0.f, ssz = 0.f;
    for (int j = 0; j < k; ++j) {
      float dx2 = 0.f, dz2 = 0.f;
#pragma unroll
      for (int d = 0; d < MX; ++d)
        if (d < cx) { const float e = xi[d] - xs[(g * k + j) * cx + d]; dx2 = fmaf(e, e, dx2); }
#pragma unroll
      for (int d = 0; d < MZ; ++d)
        if (d < cz) { const float e = zi[d] - zs[(g * k + j) * cz + d]; dz2 = fmaf(e, e, dz2); }
      ssx += sqrtf(dx2);
      ssz += sqrtf(dz2);
    }
    sx[g * k + i] = ssx;
    sz[g * k + i] = ssz;
  }
  __syncthreads();
  float loss = 0.f;
  if (on) {
    const float sxi = sx[g * k + i], szi = sz[g * k + i];
    float gr[MX];
#pragma unroll
    for (int d = 0; d < MX; ++d) gr[d] = 0.f;
    for (int j = 0; j < k; ++j) {
      float dx2 = 0.f, dz2 = 0.f;
      float e[MX];
#pragma unroll
      for (int d = 0; d < MX; ++d) {
        e[d] = 0.f;
        if (d < cx) { e[d] = xi[d] - xs[(g * k + j) * cx + d]; dx2 = fmaf(e[d], e[d], dx2); }
      }
#pragma unroll
      for (int d = 0; d < MZ; ++d)
        if (d < cz) { const float f = zi[d] - zs[(g * k + j) * cz + d]; dz2 = fmaf(f, f, dz2); }
      const float dx = sqrtf(dx2), dz = sqrtf(dz2);
      const float sxj = sx[g * k + j], szj = sz[g * k + j];
      // z_delta * 0.8 - x_delta (diversity.py:40); relu propagates NaN like torch
      const float hij = __fsub_rn(__fmul_rn(dz / szi, 0.8f), dx / sxi);
      const float hji = __fsub_rn(__fmul_rn(dz / szj, 0.8f), dx / sxj);
      loss += (hij > 0.f || hij != hij) ? hij : 0.f;
      float w = (hij > 0.f ? 1.f / sxi : 0.f) + (hji > 0.f ? 1.f / sxj : 0.f);
      if (hij != hij || hji != hji) w = hij + hji;          // NaN propagates into the gradient
      if (dx > 0.f) {
        const float f = w / dx;
#pragma unroll
        for (int d = 0; d < MX; ++d)
          if (d < cx) gr[d] = fmaf(-f, e[d], gr[d]);
      } else if (w != w) {
#pragma unroll
        for (int d = 0; d < MX; ++d)
          if (d < cx) gr[d] = w;
      }
    }
    if (a.grad != nullptr) {
#pragma unroll
      for (int d = 0; d < MX; ++d)
        if (d < cx) a.grad[(n0 * k + t) * cx + d] = a.grad_scale * gr[d];
    }
  }
  const float tot = block_sum(loss, red);
  if (threadIdx.x == 0) a.partials[bidx] = tot;
}

template <int MX, int MZ>
__global__ __launch_bounds__(kThreads) void k_ndiv(NdivArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  ndiv_block<MX, MZ>(a, blockIdx.x, smem);
}

// ================================================================ D forward / backward
// One workgroup = the same 16*RT rows of NP passes (NP = 2: the real and the fake batch of the
// D step), stacked in LDS as 16*RT*NP rows, so both passes share ONE stream of D's weights
// (the per-CU L2->L1 rate, not the MFMA rate, bounds these kernels at small M).
// Workgroups with blockIdx.x >= ntiles (if any) are NDiv blocks riding in the same launch:
// the NDiv loss/gradient needs only action_hat and the noise, so it runs on otherwise idle
// CUs during the D step instead of as a kernel of its own.
struct DPass {
  const float* action; int action_rep; float target;
};
struct DArgs {
  DNet net;
  DPass pass[2];
  const float* code; int64_t ld_code; int code_rep; int code_vec4;
  int64_t m, mpad;
  int ntiles;
  const float* ext_dlogit;    // upstream dLoss/dlogit [m] (module backward) or null -> BCE
  float inv_m;                // BCE mean scale (1 / global M)
  int do_backward;
  float* logits;              // [npass][mpad] or null
  float *h1, *h2, *h3;        // [npass*mpad x 64/128/256] or null: saved for k_wgrad
  float *dy1, *dy2, *dy3, *dl;
  float* xa;                  // [npass*mpad x 4] action inputs, for k_wgrad
  float* d_action;            // [m x 4] or null (pass 0)
  float* loss_partials;       // [ntiles] raw BCE sums over all passes, or null
  NdivArgs nd;                // blocks ntiles.. : NDiv (cx <= 4, cz <= 2), nd.n == 0 -> none
};

template <int RT, int NP>
constexpr int d_lds_floats() { return 16 * RT * NP * (260 + TAILLD + 68 + 132 + 260 + 2) + 8; }

template <int RT, int NP, bool PK>
__global__ __launch_bounds__(kThreads) void k_d(DArgs a) {
  constexpr int R = 16 * RT;        // rows per pass
  constexpr int RR = R * NP;        // LDS rows
  constexpr int RTT = RT * NP;      // 16-row tiles the layer functions see
  extern __shared__ __attribute__((aligned(16))) float smem[];
  if ((int)blockIdx.x >= a.ntiles) {
    ndiv_block<4, 2>(a.nd, (int)blockIdx.x - a.ntiles, smem);
    return;
  }
  float* Xc = smem;               // RR x 260
  float* Xt = Xc + RR * 260;      // RR x 16   action
  float* H1 = Xt + RR * TAILLD;   // RR x 68
  float* H2 = H1 + RR * 68;       // RR x 132
  float* H3 = H2 + RR * 132;      // RR x 260
  float* L = H3 + RR * 260;       // RR        logits
  float* DL = L + RR;             // RR        dLoss/dlogit
  float* red = DL + RR;           // 4 (+4 pad)
  const int64_t row0 = (int64_t)blockIdx.x * R;
  const DNet& n = a.net;
  NDP_STAMP_DECL;
  NDP_STAMP(0);

  // cat([action, code]) (models/gan.py:105): weight columns 0..3 = action (tail), 4..259 = code.
  // Each layer's first weight loads are issued one stage early (they fly across the barrier).
  FwdW<256, 64, 4, PK> w1;
  w1.preload(PK ? n.pf1 : n.w1 + ADIM, 260, n.b1, n.w1, ADIM);
  // the code part of the input is the same for every pass: fetch once, write NP copies
  for (int idx = threadIdx.x; idx < R * 64; idx += kThreads) {
    const int i = idx >> 6, k = 4 * (idx & 63);
    const int64_t row = row0 + i;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (row < a.m) {
      const float* p = a.code + (row / a.code_rep) * a.ld_code + k;
      if (a.code_vec4) {
        v = *reinterpret_cast<const f32x4*>(p);
      } else {
        v[0] = p[0]; v[1] = p[1]; v[2] = p[2]; v[3] = p[3];
      }
    }
#pragma unroll
    for (int ps = 0; ps < NP; ++ps) *reinterpret_cast<f32x4*>(Xc + (ps * R + i) * 260 + k) = v;
  }
  for (int idx = threadIdx.x; idx < RR * TAILLD; idx += kThreads) {
    const int lr = idx / TAILLD, t = idx % TAILLD;
    const int ps = lr / R;
    const int64_t row = row0 + (lr - ps * R);
    const DPass& pp = a.pass[ps];
    Xt[idx] = (row < a.m && t < ADIM) ? pp.action[(row / pp.action_rep) * ADIM + t] : 0.f;
  }
  __syncthreads();
  NDP_STAMP(1);
  FwdW<64, 128, 4, PK> w2;
  w2.bind(PK ? n.pf2 : n.w2, 64, n.b2, nullptr, 0);
  layer_fwd_run<RTT, 256, 64, ACT_LRELU, 4, PK>(w1, Xc, 260, H1, 68, Xt, TAILLD, w2);
  __syncthreads();
  NDP_STAMP(2);
  FwdW<128, 256, 4, PK> w3;
  w3.bind(PK ? n.pf3 : n.w3, 128, n.b3, nullptr, 0);
  layer_fwd_run<RTT, 64, 128, ACT_LRELU, 4, PK>(w2, H1, 68, H2, 132, nullptr, 0, w3);
  __syncthreads();
  NDP_STAMP(3);
  layer_fwd_run<RTT, 128, 256, ACT_LRELU, 4, PK>(w3, H2, 132, H3, 260, nullptr, 0);
  __syncthreads();
  NDP_STAMP(4);
  layer_fwd_narrow<RTT, 256, 1>(H3, 260, n.w4, n.b4, L, 1);
  __syncthreads();
  NDP_STAMP(5);

  float lsum = 0.f;
  if (threadIdx.x < RR) {
    const int ps = threadIdx.x / R;
    const int i = threadIdx.x - ps * R;
    const int64_t row = row0 + i;
    const float target = a.pass[ps].target;
    const float x = L[threadIdx.x];
    float dl = 0.f;
    if (row < a.m) {
      // BCEWithLogits: max(x,0) - x*y + log1p(exp(-|x|));  d/dx = sigmoid(x) - y
      lsum = fmaxf(x, 0.f) - x * target + log1pf(expf(-fabsf(x)));
      dl = a.ext_dlogit != nullptr ? a.ext_dlogit[row]
                                   : (1.f / (1.f + expf(-x)) - target) * a.inv_m;
      if (a.logits != nullptr) a.logits[(int64_t)ps * a.mpad + row] = x;
    }
    DL[threadIdx.x] = dl;
    if (a.dl != nullptr) a.dl[(int64_t)ps * a.mpad + row] = dl;
  }
  if (a.loss_partials != nullptr) {
    const float tot = block_sum(lsum, red);
    if (threadIdx.x == 0) a.loss_partials[blockIdx.x] = tot;
  }
  NDP_STAMP(6);
  if (!a.do_backward) {
    NDP_STAMP_FLUSH(7, 2);
    return;
  }
  if (a.h1 != nullptr) {
#pragma unroll
    for (int ps = 0; ps < NP; ++ps) {
      const int64_t g0 = (int64_t)ps * a.mpad + row0;
      store_tile<RT, 64>(a.h1 + g0 * 64, 64, H1 + ps * R * 68, 68);
      store_tile<RT, 128>(a.h2 + g0 * 128, 128, H2 + ps * R * 132, 132);
      store_tile<RT, 256>(a.h3 + g0 * 256, 256, H3 + ps * R * 260, 260);
    }
    if (threadIdx.x < RR) {
      const int ps = threadIdx.x / R;
      const int64_t g = (int64_t)ps * a.mpad + row0 + (threadIdx.x - ps * R);
      *reinterpret_cast<f32x4*>(a.xa + g * 4) = *reinterpret_cast<const f32x4*>(Xt + threadIdx.x * TAILLD);
    }
  }
  __syncthreads();
  NDP_STAMP(7);
  DgW<128, 256, PK> g3;
  g3.preload(PK ? n.pg3 : n.w3, 128);
  layer_dgrad_narrow<RTT, 256, 1, ACT_LRELU>(DL, 1, n.w4, H3, 260);          // H3 := dY3
  __syncthreads();
  NDP_STAMP(8);
  DgW<64, 128, PK> g2;
  g2.bind(PK ? n.pg2 : n.w2, 64);
  layer_dgrad_run<RTT, 128, 256, ACT_LRELU, PK>(g3, H3, 260, H2, 132, g2);   // H2 := dY2
  __syncthreads();
  NDP_STAMP(9);
  layer_dgrad_run<RTT, 64, 128, ACT_LRELU, PK>(g2, H2, 132, H1, 68);         // H1 := dY1
  __syncthreads();
  NDP_STAMP(10);
  if (a.dy1 != nullptr) {
#pragma unroll
    for (int ps = 0; ps < NP; ++ps) {
      const int64_t g0 = (int64_t)ps * a.mpad + row0;
      store_tile<RT, 64>(a.dy1 + g0 * 64, 64, H1 + ps * R * 68, 68);
      store_tile<RT, 128>(a.dy2 + g0 * 128, 128, H2 + ps * R * 132, 132);
      store_tile<RT, 256>(a.dy3 + g0 * 256, 256, H3 + ps * R * 260, 260);
    }
  }
  if (a.d_action != nullptr && threadIdx.x < R * ADIM) {
    // dLoss/d action = dY1 . W1[:, 0:4]   (pass 0)
    const int i = threadIdx.x >> 2, j = threadIdx.x & 3;
    const int64_t row = row0 + i;
    float s = 0.f;
#pragma unroll 8
    for (int o = 0; o < 64; ++o) s = fmaf(H1[i * 68 + o], n.w1[o * 260 + j], s);
    if (row < a.m) a.d_action[row * ADIM + j] = s;
  }
  NDP_STAMP(11);
  NDP_STAMP_FLUSH(12, 2);
}

// ================================================================ G backward (data path)
struct GBwdArgs {
  GNet net;
  int64_t m;
  const float *h1, *h2, *h3, *h4;   // saved by k_g_fwd [mpad x .]
  const float* d_action;            // [m x 4]
  const float* d_action2;           // [m x 4] added to d_action, or null (NDiv gradient)
  float *dy1, *dy2, *dy3, *dy4, *dy5;   // [mpad x 128/64/128/256/4]
};

template <int RT>
constexpr int g_bwd_lds_floats() { return 16 * RT * (132 + 68 + 132 + 260 + 4); }

template <int RT, bool PK>
__global__ __launch_bounds__(kThreads) void k_g_bwd(GBwdArgs a) {
  constexpr int R = 16 * RT;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* H1 = smem;               // R x 132
  float* H2 = H1 + R * 132;       // R x 68
  float* H3 = H2 + R * 68;        // R x 132
  float* H4 = H3 + R * 132;       // R x 260
  float* DA = H4 + R * 260;       // R x 4
  const int64_t row0 = (int64_t)blockIdx.x * R;
  const GNet& n = a.net;

  DgW<128, 256, PK> g4;
  g4.preload(PK ? n.pg4 : n.w4, 128);
  load_tile<RT, 128>(H1, 132, a.h1 + row0 * 128, 128);
  load_tile<RT, 64>(H2, 68, a.h2 + row0 * 64, 64);
  load_tile<RT, 128>(H3, 132, a.h3 + row0 * 128, 128);
  load_tile<RT, 256>(H4, 260, a.h4 + row0 * 256, 256);
  if (threadIdx.x < R) {
    const int64_t row = row0 + threadIdx.x;
    f32x4 g = {0.f, 0.f, 0.f, 0.f};
    if (row < a.m) {
      g = *reinterpret_cast<const f32x4*>(a.d_action + row * 4);
      if (a.d_action2 != nullptr) g += *reinterpret_cast<const f32x4*>(a.d_action2 + row * 4);
    }
    *reinterpret_cast<f32x4*>(DA + threadIdx.x * 4) = g;
    *reinterpret_cast<f32x4*>(a.dy5 + row * 4) = g;
  }
  __syncthreads();
  layer_dgrad_narrow<RT, 256, 4, ACT_RELU>(DA, 4, n.w5, H4, 260);                // H4 := dY4
  __syncthreads();
  DgW<64, 128, PK> g3;
  g3.bind(PK ? n.pg3 : n.w3, 64);
  layer_dgrad_run<RT, 128, 256, ACT_RELU, PK>(g4, H4, 260, H3, 132, g3);         // H3 := dY3
  __syncthreads();
  DgW<128, 64, PK> g2;
  g2.bind(PK ? n.pg2 : n.w2, 128);
  layer_dgrad_run<RT, 64, 128, ACT_RELU, PK>(g3, H3, 132, H2, 68, g2);           // H2 := dY2
  __syncthreads();
  layer_dgrad_run<RT, 128, 64, ACT_RELU, PK>(g2, H2, 68, H1, 132);               // H1 := dY1
  __syncthreads();
  store_tile<RT, 128>(a.dy1 + row0 * 128, 128, H1, 132);
  store_tile<RT, 64>(a.dy2 + row0 * 64, 64, H2, 68);
  store_tile<RT, 128>(a.dy3 + row0 * 128, 128, H3, 132);
  store_tile<RT, 256>(a.dy4 + row0 * 256, 256, H4, 260);
}

// Per-tile segment sums of a pre-activation gradient tile for the K-deduplicated fc1 weight
// gradient: entry (tile*S + s) = sum over the tile's rows r with floor(r / K) == floor(16*tile / K) + s
// of (dY[r] + dY[r + pass stride]) -- all rows of one entry multiply the SAME code row.
// dY: LDS tile [NP*16][ld]; out: global [entries_padded][W].  The last tile also zeroes the padding.
template <int W, int NP>
__device__ __forceinline__ void store_segment_sums(const float* dY, int ld, float* __restrict__ out,
                                                   int64_t row0, int K, int S, int entries_padded,
                                                   int tile, int ntiles) {
  const int f0 = (int)(row0 / K);
  for (int idx = threadIdx.x; idx < S * W; idx += kThreads) {
    const int sgm = idx / W, col = idx - sgm * W;
    const int64_t lo64 = (int64_t)(f0 + sgm) * K, hi64 = lo64 + K;
    const int lo = (int)(lo64 > row0 ? lo64 - row0 : 0);
    const int hi = (int)(hi64 < row0 + 16 ? hi64 - row0 : 16);
    float acc = 0.f;
    for (int r = lo; r < hi; ++r) {
#pragma unroll
      for (int ps = 0; ps < NP; ++ps) acc += dY[(ps * 16 + r) * ld + col];
    }
    out[((size_t)tile * S + sgm) * W + col] = acc;
  }
  if (tile == ntiles - 1) {
    const int first = ntiles * S;
    for (int idx = threadIdx.x; idx < (entries_padded - first) * W; idx += kThreads) out[(size_t)first * W + idx] = 0.f;
  }
}

// ================================================================ fused row-tile kernels of the step
// Phase A (train_gan.py:165-183): G forward, D on the fake and on the real rows, BCE, D's backward data path, in one
// launch of two kinds of workgroups, each on one 16-row tile whose activations never leave LDS:
//   role 0 (blocks 0 .. ntiles-1)   tile t of the M = FLAT*K generated rows: G fc1..fc5 (draws the noise), then D on
//                                   the 16 fake rows (action_hat goes from G to D through LDS), loss, D backward;
//   role 1 (blocks ntiles ..)       D on the REAL rows.  The reference feeds D K copies of every (action, code) pair
//                                   (repeat_interleave, train_gan.py:140-156: action_unsqueeze / codes_unsqueeze), so the
//                                   K rows of a FLAT row have identical logits and identical gradients.  They are
//                                   computed once: role 1 runs over the FLAT distinct rows and weights each with K --
//                                   dLoss/dlogit and the BCE term are multiplied by K, everything downstream (deltas,
//                                   weight-gradient sums) is linear in them.  Same sums as the reference's, K equal
//                                   terms added as one product; 1/K of the real pass' rows through D, its activations
//                                   and deltas for k_wgrad 1/K of the bytes.
// The real pass does not depend on G, so its workgroups run beside the others from the start.
// LDS regions are reused as the data dies:  XC code tile -> D.h3 / dY3 | B1 G.h1, G.h3 -> D.h2 / dY2 | B2 G.h2 ->
// D.h1 / dY1 | B3 G.h4.
struct PhaseAArgs {
  GNet g; DNet d;
  const float* code; int code_rep;          // [flat x 256], generated row r uses code[r / code_rep] (code_rep = K)
  const float* noise;                        // [m x nz] input, or null when noise_out != null
  float* noise_out; uint64_t noise_seed; const int32_t* noise_step;
  const float* actions;                      // ground-truth actions [flat x 4], one per FLAT row
  int64_t m, mpad;                           // generated (fake) rows, padded
  int64_t flat, rpad;                        // distinct real rows (m / K), padded
  float inv_m;                               // BCE mean scale: 1 / global M
  float real_scale;                          // K: what one real row stands for
  float *gh1, *gh2, *gh3, *gh4;              // G activations out [mpad x .]
  float* action_hat;                         // [m x 4] out
  float *h1, *h2, *h3, *dy1, *dy2, *dy3, *dl, *xa;   // D buffers for k_wgrad [(rpad + mpad) x .]: real rows, then fake rows
  float* loss_partials;                      // [gridDim.x]
  float* dy1seg; int seg_s, seg_entries;     // fc1 code-column operand of k_wgrad [(seg_entries + rpad) x 64]: per-tile
                                             // segment sums of the fake pass' dY1, then the real rows' dY1 as they are
};

// Small weights every tile needs -- G.fc5 [4 x 256] + bias, D.fc4 [256] + bias -- are staged in LDS once per workgroup:
// read straight from global inside the narrow (VALU) layers they were a serial chain of L2 round trips (stamps: fc5
// 0.9 us, D fc4 + loss 1.2 us, 4.9 us for the 64-step dA loop at large M).
constexpr int kSW5 = 1024 + 8, kSW4 = 256 + 8;
constexpr int phase_a_lds_floats() {
  return 16 * (260 + TAILLD + 132 + 68 + 2) + 16 * 260 + 16 * 4 + 8 + kSW5 + kSW4;
}

// ---- The K rows of a flat row share its code (the reference repeats it K times, train_gan.py:140-156), so a 16-row tile
// holds few DISTINCT code rows: 1 when K % 16 == 0, at most 4 when K >= 6 (floor(15 / K) + 2).  The 256 code columns of
// G.fc1 and of D.fc1 then need one dot product per output column and DISTINCT code row -- on the VALU, 128 FMAs per
// thread and code row, the weights read once for all rows -- instead of 16 k-steps of the matrix pipe over 16 rows, and
// the layer's output is the row's product plus the narrow tail (noise / action columns): no MFMA at all for fc1.
// (NR variants of the phase kernels: NR = number of code rows carried per tile, 0 = the MFMA path.  27 % of phase A's
// MFMAs, 10 % of phase B's.)
// part[(j * PARTS + p) * OUT + o] = sum over segment p of code row j of W[o][.] * code_j[.]   (PARTS = 256 threads / OUT).
// W is read from the layer's forward-packed copy (fwd_pack_offset): the 4 consecutive k of one (t, q) are 16 bytes, and
// the 16 output columns of a tile sit 16 bytes apart -- a wave's load covers four 256-byte spans (row-major W would be
// 64 different cache lines per load: measured 2.8 % instead of 4.9 % at config 5).  `code`: NR rows of stride 260 in LDS.
template <int OUT, int NR>
__device__ __forceinline__ void code_rows_dot(const float* __restrict__ Wpacked, const float* code, float* part) {
  constexpr int PARTS = kThreads / OUT, SEG = CODE / PARTS;
  const int o = threadIdx.x % OUT, p = threadIdx.x / OUT;
  const float* w = Wpacked + fwd_pack_offset(o, p * SEG, CODE, OUT);      // + 256 floats per k-step t, + 64 per q
  const float* cr = code + p * SEG;
  float s0[NR], s1[NR];
#pragma unroll
  for (int j = 0; j < NR; ++j) s0[j] = s1[j] = 0.f;
#pragma unroll 2
  for (int t = 0; t < SEG / 16; ++t) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 wv = *reinterpret_cast<const f32x4*>(w + t * 256 + q * 64);
#pragma unroll
      for (int j = 0; j < NR; ++j) {
        const f32x4 cv = *reinterpret_cast<const f32x4*>(cr + j * 260 + 16 * t + 4 * q);
        s0[j] = fmaf(wv[0], cv[0], s0[j]); s1[j] = fmaf(wv[1], cv[1], s1[j]);
        s0[j] = fmaf(wv[2], cv[2], s0[j]); s1[j] = fmaf(wv[3], cv[3], s1[j]);
      }
    }
  }
#pragma unroll
  for (int j = 0; j < NR; ++j) part[(j * PARTS + p) * OUT + o] = s0[j] + s1[j];
}
// the tile's NR code rows (flat rows f0 .. f0 + NR - 1, zeros past the last one) -> LDS rows of stride 260
template <int NR>
__device__ __forceinline__ void load_code_rows(float* XC, const float* __restrict__ code, int64_t f0, int64_t nflat) {
  const int j = threadIdx.x >> 6, c4 = threadIdx.x & 63;
  if (j < NR) {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (f0 + j < nflat) v = *reinterpret_cast<const f32x4*>(code + (size_t)(f0 + j) * CODE + 4 * c4);
    *reinterpret_cast<f32x4*>(XC + j * 260 + 4 * c4) = v;
  }
}
// Y[r][o] = act(bias[o] + the PARTS partial sums of row r's code row + sum_j XT[r][j] * Wt[o][j]) for the tile's 16 rows;
// row r's code row = (rem0 + r) / K, rem0 = row0 % K, clamped to NR - 1 (rows past M: masked later, only finite here)
template <int OUT, int ACT, int NR>
__device__ __forceinline__ void fc1_from_code_rows(const float* part, const float* __restrict__ bias,
                                                   const float* __restrict__ Wt, int ldw, int ntail,
                                                   const float* XT, float* Y, int ldy, int rem0, int K) {
  constexpr int PARTS = kThreads / OUT, RSTEP = kThreads / OUT;
  const int o = threadIdx.x % OUT, r0 = threadIdx.x / OUT;
  float wt[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) wt[j] = j < ntail ? Wt[(size_t)o * ldw + j] : 0.f;
  float base[NR];
#pragma unroll
  for (int j = 0; j < NR; ++j) {
    base[j] = bias[o];
#pragma unroll
    for (int p = 0; p < PARTS; ++p) base[j] += part[(j * PARTS + p) * OUT + o];
  }
  for (int r = r0; r < 16; r += RSTEP) {
    float s = base[0];
    if (NR > 1) {
      const int x = rem0 + r;
      const int idx = (x >= K) + (x >= 2 * K) + (x >= 3 * K);
#pragma unroll
      for (int j = 1; j < NR; ++j) s = idx >= j ? base[j] : s;
    }
#pragma unroll
    for (int j = 0; j < 16; ++j)
      if (j < ntail) s = fmaf(XT[r * TAILLD + j], wt[j], s);
    Y[r * ldy + o] = act_fwd<ACT>(s);
  }
}

// D forward + BCE + D backward data path on one 16-row tile whose inputs are in XC (codes) and XT (actions) and whose
// fc1 weights `dw1` are already in flight.  real: the tile holds distinct real rows (weight real_scale each, stored
// at row offset 0); otherwise fake rows (stored behind the rpad real rows).
// NR > 0 (fake rows only): fc1 from the tile's code-row partial products `cpd` (see code_rows_dot); `dw2` is then already
// in flight.
template <bool PK, int RG, int NR>
__device__ __forceinline__ void phase_a_d_part(const PhaseAArgs& a, bool real, int tile, int ntiles,
                                               FwdW<256, 64, 4, PK, RG>& dw1, FwdW<64, 128, 4, PK, RG>& dw2, const float* cpd,
                                               float* XC, float* XT, float* B1,
                                               float* B2, float* L, float* DL, const float* W4S,
                                               unsigned long long* stamp_lds_) {
  constexpr int R = 16;
  const DNet& d = a.d;
  (void)stamp_lds_;
  const int64_t row0 = (int64_t)tile * R;
  const int64_t nvalid = real ? a.flat : a.m;
  const int64_t g0 = (real ? 0 : a.rpad) + row0;                  // row of this tile in the buffers k_wgrad reads
  // each layer's first weight fragments are issued inside the previous layer's k-loop (FwdW::preload_slice)
  NDP_STAMP(10);
  if (NR > 0) {
    fc1_from_code_rows<64, ACT_LRELU, (NR > 0 ? NR : 1)>(cpd, d.b1, d.w1, 260, ADIM, XT, B2, 68,
                                                         (int)(row0 % a.code_rep), a.code_rep);   // D.h1 -> B2
  } else {
    dw2.bind(PK ? d.pf2 : d.w2, 64, d.b2, nullptr, 0);
    layer_fwd_run<1, 256, 64, ACT_LRELU, 4, PK>(dw1, XC, 260, B2, 68, XT, TAILLD, dw2);  // D.h1 -> B2
  }
  NDP_STAMP(11);
  __syncthreads();
  NDP_STAMP(3);
  FwdW<128, 256, 4, PK, RG> dw3;
  dw3.bind(PK ? d.pf3 : d.w3, 128, d.b3, nullptr, 0);
  layer_fwd_run<1, 64, 128, ACT_LRELU, 4, PK>(dw2, B2, 68, B1, 132, nullptr, 0, dw3);  // D.h2 -> B1
  __syncthreads();
  // (the backward's first layer too: its fragments wait in registers through the narrow fc4 / loss stages)
  DgW<128, 256, PK, RG> dg3;
  dg3.bind(PK ? d.pg3 : d.w3, 128);
  layer_fwd_run<1, 128, 256, ACT_LRELU, 4, PK>(dw3, B1, 132, XC, 260, nullptr, 0, dg3); // D.h3 -> XC (code tile is dead)
  __syncthreads();
  NDP_STAMP(4);
  layer_fwd_narrow<1, 256, 1>(XC, 260, W4S, W4S + 256, L, 1);
  __syncthreads();
  float lsum = 0.f;
  if (threadIdx.x < R) {
    const int64_t row = row0 + threadIdx.x;
    const float target = real ? 1.f : 0.f;                        // real: ones, fake: zeros (train_gan.py:174-181)
    const float weight = real ? a.real_scale : 1.f;               // a real row stands for K identical rows
    const float x = L[threadIdx.x];
    float dl = 0.f;
    if (row < nvalid) {
      lsum = weight * (fmaxf(x, 0.f) - x * target + log1pf(expf(-fabsf(x))));
      dl = weight * ((1.f / (1.f + expf(-x)) - target) * a.inv_m);
    }
    DL[threadIdx.x] = dl;
    a.dl[g0 + threadIdx.x] = dl;
  }
  if (threadIdx.x < 64) {               // the rows' losses all sit in wave 0: no block-wide reduction
    const float tot = wave_sum(lsum);
    if (threadIdx.x == 0) a.loss_partials[blockIdx.x] = tot;
  }
  NDP_STAMP(5);
  store_tile<1, 64>(a.h1 + g0 * 64, 64, B2, 68);
  store_tile<1, 128>(a.h2 + g0 * 128, 128, B1, 132);
  store_tile<1, 256>(a.h3 + g0 * 256, 256, XC, 260);
  if (threadIdx.x < R)
    *reinterpret_cast<f32x4*>(a.xa + (g0 + threadIdx.x) * 4) = *reinterpret_cast<const f32x4*>(XT + threadIdx.x * TAILLD);
  __syncthreads();
  NDP_STAMP(6);
  layer_dgrad_narrow<1, 256, 1, ACT_LRELU>(DL, 1, W4S, XC, 260);                       // XC := dY3
  __syncthreads();
  DgW<64, 128, PK, RG> dg2;
  dg2.bind(PK ? d.pg2 : d.w2, 64);
  layer_dgrad_run<1, 128, 256, ACT_LRELU, PK>(dg3, XC, 260, B1, 132, dg2);             // B1 := dY2
  __syncthreads();
  NDP_STAMP(7);
  layer_dgrad_run<1, 64, 128, ACT_LRELU, PK>(dg2, B1, 132, B2, 68);                    // B2 := dY1
  __syncthreads();
  // fc1's code-column weight gradient runs K-deduplicated: the fake rows of a tile that share a FLAT row (same code)
  // are pre-summed here; a real row is its FLAT row
  if (real) store_tile<1, 64>(a.dy1seg + ((size_t)a.seg_entries + row0) * 64, 64, B2, 68);
  else store_segment_sums<64, 1>(B2, 68, a.dy1seg, row0, a.code_rep, a.seg_s, a.seg_entries, tile, ntiles);
  store_tile<1, 64>(a.dy1 + g0 * 64, 64, B2, 68);
  store_tile<1, 128>(a.dy2 + g0 * 128, 128, B1, 132);
  store_tile<1, 256>(a.dy3 + g0 * 256, 256, XC, 260);
}

#ifndef NDP_PHASE_A_SMALL_RING
#define NDP_PHASE_A_SMALL_RING 24     // 32 needs 36 bytes of scratch per lane under the 168-VGPR cap of three workgroups per CU;
#endif                                // 24: none, and 0.7 - 1 % of the large-M step (B = 1024 / K = 6: 0.5325 -> 0.5288 ms)
// RG = VGPR budget of each weight prefetch ring: 96 keeps a lone workgroup per CU streaming (grids of up to 256
// workgroups); a small one (phase A 24, phase B 32) with a tighter register cap lets several workgroups share a CU at large M.
template <bool PK, int RG, int NR>
__global__ __launch_bounds__(kThreads, (RG >= 96 ? 1 : 3)) void k_phase_a(PhaseAArgs a_segment) {
  constexpr bool DD = NR > 0;
  constexpr int NRR = NR > 0 ? NR : 1;
  // one workgroup per CU (RG = 96): nothing hides the argument loads, read them in one round trip (load_kernargs);
  // with several workgroups per CU the registers that costs are worth more
  const PhaseAArgs a = (RG >= 96 && kFastKernargs) ? load_kernargs<PhaseAArgs>() : a_segment;
  constexpr int R = 16;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* XC = smem;                  // 16 x 260
  float* XT = XC + R * 260;          // 16 x 16
  float* B1 = XT + R * TAILLD;       // 16 x 132
  float* B2 = B1 + R * 132;          // 16 x 68
  float* B3 = B2 + R * 68;           // 16 x 260
  float* A = B3 + R * 260;           // 16 x 4
  float* L = A + R * 4;              // 16
  float* DL = L + R;                 // 16
  float* red = DL + R;               // 8
  float* W5S = red + 8;              // G.fc5 weights [4][256] + bias (kSW5 floats)
  float* W4S = W5S + kSW5;           // D.fc4 weights [256] + bias    (kSW4 floats)
  const int ntiles = (int)(a.mpad / R);
  const bool real = (int)blockIdx.x >= ntiles;                    // role 1: a tile of distinct real rows
  const GNet& g = a.g;
  const DNet& d = a.d;
  NDP_STAMP_DECL;
  NDP_STAMP(0);

  FwdW<256, 64, 4, PK, RG> dw1;
  FwdW<64, 128, 4, PK, RG> dw2;
  if (!real) {
    const int tile = (int)blockIdx.x;
    const int64_t row0 = (int64_t)tile * R;
    // ---------------- G forward
    FwdW<256, 128, 2, PK, RG> gw1;
    FwdW<128, 64, 4, PK, RG> gw2;
    // The input tile's loads go out BEFORE the weight ring's: loads return in order, so behind the ring's 24 fragments
    // per lane (1.4 us to issue at kernel start, stamps) the tile's LDS writes waited for all of them.
    f32x4 ct[R * 64 / kThreads];                          // the code tile, 4 float4 per thread
    if (DD) {                                            // the tile's distinct code rows -> XC rows 0 .. NR-1
      load_code_rows<NRR>(XC, a.code, (int64_t)((uint32_t)row0 / (uint32_t)a.code_rep), a.flat);
    } else {
#pragma unroll
      for (int u = 0; u < R * 64 / kThreads; ++u) {
        const int idx = threadIdx.x + u * kThreads;
        const int i = idx >> 6, k = 4 * (idx & 63);
        const int64_t row = row0 + i;
        const bool ok = row < a.m;                       // unconditional load from a valid row, zeroed by a select
        const f32x4 v = *reinterpret_cast<const f32x4*>(a.code + (size_t)((uint32_t)(ok ? row : 0) / (uint32_t)a.code_rep) * CODE + k);
        ct[u] = ok ? v : f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
    const f32x4 w5r = *reinterpret_cast<const f32x4*>(g.w5 + 4 * threadIdx.x);
    const float b5r = g.b5[threadIdx.x & 3], w4r = d.w4[threadIdx.x], b4r = d.b4[0];
    float nv = 0.f;                                       // the noise tile: R x TAILLD = one element per thread
    {
      static_assert(R * TAILLD == kThreads, "one noise element per thread");
      const int i = threadIdx.x / TAILLD, t = threadIdx.x % TAILLD;
      const int64_t row = row0 + i;
      if (row < a.m && t < g.nz) {
        if (a.noise_out != nullptr) {
          nv = philox_uniform((uint64_t)(row * g.nz + t), a.noise_seed, (uint32_t)*a.noise_step);
          a.noise_out[row * g.nz + t] = nv;
        } else {
          nv = a.noise[row * g.nz + t];
        }
      }
    }
    pin_vmem();
    if (DD) gw2.preload(PK ? g.pf2 : g.w2, 128, g.b2, nullptr, 0);
    else gw1.preload(PK ? g.pf1 : g.w1, g.ld1, g.b1, g.w1 + CODE, g.nz);
    NDP_STAMP(24);
    if (!DD) {
#pragma unroll
      for (int u = 0; u < R * 64 / kThreads; ++u) {
        const int idx = threadIdx.x + u * kThreads;
        *reinterpret_cast<f32x4*>(XC + (idx >> 6) * 260 + 4 * (idx & 63)) = ct[u];
      }
    }
    NDP_STAMP(25);
    XT[threadIdx.x] = nv;
    *reinterpret_cast<f32x4*>(W5S + 4 * threadIdx.x) = w5r;
    if (threadIdx.x < 4) W5S[1024 + threadIdx.x] = b5r;
    W4S[threadIdx.x] = w4r;
    if (threadIdx.x == 0) W4S[256] = b4r;
    NDP_STAMP(26);
    __syncthreads();
    NDP_STAMP(1);
    // every layer's first weight fragments are issued inside the previous layer's k-loop (FwdW::preload_slice)
    float* CPG = XC + NRR * 260;                         // NR > 0: partial code-row products of G.fc1 [NR][2][128] ...
    float* CPD = CPG + NRR * 256;                        // ... and of D.fc1 [NR][4][64], behind the code rows
    if (DD) {
      code_rows_dot<128, NRR>(g.pf1, XC, CPG);
      code_rows_dot<64, NRR>(d.pf1, XC, CPD);
      __syncthreads();
      fc1_from_code_rows<128, ACT_RELU, NRR>(CPG, g.b1, g.w1 + CODE, g.ld1, g.nz, XT, B1, 132,
                                             (int)(row0 % a.code_rep), a.code_rep);       // h1 -> B1
    } else {
      gw2.bind(PK ? g.pf2 : g.w2, 128, g.b2, nullptr, 0);
      layer_fwd_run<1, 256, 128, ACT_RELU, 2, PK>(gw1, XC, 260, B1, 132, XT, TAILLD, gw2); // h1 -> B1
    }
    __syncthreads();
    NDP_STAMP(12);
    FwdW<64, 128, 4, PK, RG> gw3;
    gw3.bind(PK ? g.pf3 : g.w3, 64, g.b3, nullptr, 0);
    store_tile<1, 128>(a.gh1 + row0 * 128, 128, B1, 132);
    layer_fwd_run<1, 128, 64, ACT_RELU, 4, PK>(gw2, B1, 132, B2, 68, nullptr, 0, gw3);   // h2 -> B2
    __syncthreads();
    NDP_STAMP(13);
    FwdW<128, 256, 4, PK, RG> gw4;
    gw4.bind(PK ? g.pf4 : g.w4, 128, g.b4, nullptr, 0);
    NDP_STAMP(16);
    store_tile<1, 64>(a.gh2 + row0 * 64, 64, B2, 68);
    NDP_STAMP(17);
    layer_fwd_run<1, 64, 128, ACT_RELU, 4, PK>(gw3, B2, 68, B1, 132, nullptr, 0, gw4);   // h3 -> B1 (h1 is stored)
    NDP_STAMP(18);
    __syncthreads();
    NDP_STAMP(14);
    if (DD) dw2.bind(PK ? d.pf2 : d.w2, 64, d.b2, nullptr, 0);                            // D's first MFMA layer flies during G fc4
    else dw1.bind(PK ? d.pf1 : d.w1 + ADIM, 260, d.b1, d.w1, ADIM);
    NDP_STAMP(19);
    store_tile<1, 128>(a.gh3 + row0 * 128, 128, B1, 132);
    NDP_STAMP(20);
    if (DD) layer_fwd_run<1, 128, 256, ACT_RELU, 4, PK>(gw4, B1, 132, B3, 260, nullptr, 0, dw2);
    else layer_fwd_run<1, 128, 256, ACT_RELU, 4, PK>(gw4, B1, 132, B3, 260, nullptr, 0, dw1); // h4 -> B3
    NDP_STAMP(21);
    __syncthreads();
    NDP_STAMP(15);
    store_tile<1, 256>(a.gh4 + row0 * 256, 256, B3, 260);
    NDP_STAMP(22);
    layer_fwd_narrow<1, 256, 4>(B3, 260, W5S, W5S + 1024, A, 4);                          // action_hat -> A
    NDP_STAMP(23);
    __syncthreads();
    NDP_STAMP(2);
    for (int idx = threadIdx.x; idx < R * TAILLD; idx += kThreads) {                      // fake actions -> XT (the noise is dead)
      const int i = idx / TAILLD, t = idx % TAILLD;
      const int64_t row = row0 + i;
      XT[idx] = (row < a.m && t < ADIM) ? A[i * 4 + t] : 0.f;
    }
    if (threadIdx.x < R) {
      const int64_t row = row0 + threadIdx.x;
      if (row < a.m)
        *reinterpret_cast<f32x4*>(a.action_hat + row * 4) = *reinterpret_cast<const f32x4*>(A + threadIdx.x * 4);
    }
    __syncthreads();
    NDP_STAMP(9);
    // ---------------- D on the fake rows
    phase_a_d_part<PK, RG, NR>(a, false, tile, ntiles, dw1, dw2, CPD, XC, XT, B1, B2, L, DL, W4S, NDP_STAMP_PTR);
  } else {
    // ---------------- role 1: D on one tile of the distinct real rows
    const int tile = (int)blockIdx.x - ntiles;
    const int64_t row0 = (int64_t)tile * R;
    dw1.preload(PK ? d.pf1 : d.w1 + ADIM, 260, d.b1, d.w1, ADIM);
    W4S[threadIdx.x] = d.w4[threadIdx.x];
    if (threadIdx.x == 0) W4S[256] = d.b4[0];
    for (int idx = threadIdx.x; idx < R * 64; idx += kThreads) {
      const int i = idx >> 6, k = 4 * (idx & 63);
      const int64_t row = row0 + i;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (row < a.flat) v = *reinterpret_cast<const f32x4*>(a.code + row * CODE + k);
      *reinterpret_cast<f32x4*>(XC + i * 260 + k) = v;
    }
    for (int idx = threadIdx.x; idx < R * TAILLD; idx += kThreads) {
      const int i = idx / TAILLD, t = idx % TAILLD;
      const int64_t row = row0 + i;
      XT[idx] = (row < a.flat && t < ADIM) ? a.actions[row * ADIM + t] : 0.f;
    }
    __syncthreads();
    phase_a_d_part<PK, RG, 0>(a, true, tile, ntiles, dw1, dw2, nullptr, XC, XT, B1, B2, L, DL, W4S, NDP_STAMP_PTR);
  }
  (void)red;
  NDP_STAMP(8);
  NDP_STAMP_FLUSH(32, 5);
}

// Phase B (train_gan.py:187-202) for one 16-row tile: D' forward with the updated D, G loss,
// dLoss/d action_hat, + the NDiv gradient, then G's backward data path.  Versus k_d<1,1> +
// k_g_bwd: one launch less, dLoss/d action_hat never leaves LDS.  39 KB LDS -> 4 per CU.
struct PhaseBArgs {
  GNet g; DNet d;
  const float* code; int code_rep;
  const float* action_hat;                   // [m x 4]
  const float* nd_grad;                      // [m x 4] factor * dNDiv/d action_hat, or null
  int64_t m;
  float inv_m;
  const float *gh1, *gh2, *gh3, *gh4;        // G activations saved by phase A
  float *dy1, *dy2, *dy3, *dy4, *dy5;        // G pre-activation gradients out [mpad x 128/64/128/256/4]
  float* loss_partials;                      // [ntiles] raw BCE sums (G loss)
  float* dy1seg; int seg_s, seg_entries;     // segment sums of G's dY1 [seg_entries x 128]
};

// pre: G's saved activations h2..h4 and the action columns of D.fc1 are fetched into LDS regions of their own at
// the start of the kernel (+30 KB) instead of into the dead D regions in the middle of it
constexpr int phase_b_lds_floats(bool pre) {
  return 16 * (260 + TAILLD + 132 + 68 + 132 + 4 + 2) + 8 + 256 + kSW4 + kSW5 + (pre ? 16 * (260 + 132 + 68) : 0);
}

// PRE (one workgroup per CU, small M): the loads of G's activations for the backward half are issued with the
// input tile, so the ~2.6 us round trip in the middle of the kernel (global -> LDS -> barrier, measured with
// stamps) overlaps the D' half; with several workgroups per CU (large M) other workgroups hide it and the LDS
// is better spent on residency.
template <bool PK, int RG, bool PRE, int NR>
__global__ __launch_bounds__(kThreads, (RG >= 96 ? 2 : 3)) void k_phase_b(PhaseBArgs a_segment) {
  constexpr bool DD = NR > 0;
  constexpr int NRR = NR > 0 ? NR : 1;
  const PhaseBArgs a = (RG >= 96 && kFastKernargs) ? load_kernargs<PhaseBArgs>() : a_segment;   // see k_phase_a
  constexpr int R = 16;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* XC = smem;                  // 16 x 260: code tile -> D.h3 / dY3 -> G.h4 / dY4
  float* XT = XC + R * 260;          // 16 x 16 : action_hat
  float* B1 = XT + R * TAILLD;       // 16 x 132: D.h2 / dY2 -> G.h3 / dY3
  float* B2 = B1 + R * 132;          // 16 x 68 : D.h1 / dY1 -> G.h2 / dY2
  float* H1 = B2 + R * 68;           // 16 x 132: G.h1 / dY1
  float* DA = H1 + R * 132;          // 16 x 4
  float* L = DA + R * 4;             // 16
  float* DL = L + R;                 // 16
  float* red = DL + R;               // 8
  float* W1A = red + 8;              // 64 x 4 action columns of D.fc1   } small weights staged once per workgroup
  float* W4S = W1A + 256;            // D.fc4 [256] + bias                } (see phase_a_lds_floats)
  float* W5S = W4S + kSW4;           // G.fc5 [4][256]                    }
  float* G4 = PRE ? W5S + kSW5 : XC; // G.h4 / dY4   (!PRE: the D regions, loaded when they are dead)
  float* G3 = PRE ? G4 + R * 260 : B1;
  float* G2 = PRE ? G3 + R * 132 : B2;
  const int64_t row0 = (int64_t)blockIdx.x * R;
  const GNet& g = a.g;
  const DNet& d = a.d;
  NDP_STAMP_DECL;
  NDP_STAMP(0);

  FwdW<256, 64, 4, PK, RG> dw1;
  FwdW<64, 128, 4, PK, RG> dw2;
  // The input tile's and the small weights' loads go out BEFORE the weight ring's (see k_phase_a).
  f32x4 ct[R * 64 / kThreads];
  float av = 0.f;                                        // the action tile: one element per thread
  {
    if (DD) {                                            // the tile's distinct code rows (see code_rows_dot) -> XC rows 0 .. NR-1
      load_code_rows<NRR>(XC, a.code, row0 / a.code_rep, (a.m + a.code_rep - 1) / a.code_rep);
    } else {
#pragma unroll
      for (int u = 0; u < R * 64 / kThreads; ++u) {
        const int idx = threadIdx.x + u * kThreads;
        const int i = idx >> 6, k = 4 * (idx & 63);
        const int64_t row = row0 + i;
        const bool ok = row < a.m;
        const f32x4 v = *reinterpret_cast<const f32x4*>(a.code + ((ok ? row : 0) / a.code_rep) * CODE + k);
        ct[u] = ok ? v : f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
    const int i = threadIdx.x / TAILLD, t = threadIdx.x % TAILLD;
    const int64_t row = row0 + i;
    if (row < a.m && t < ADIM) av = a.action_hat[row * ADIM + t];
  }
  const float w1a = d.w1[(threadIdx.x >> 2) * 260 + (threadIdx.x & 3)], w4r = d.w4[threadIdx.x], b4r = d.b4[0];
  const f32x4 w5r = *reinterpret_cast<const f32x4*>(g.w5 + 4 * threadIdx.x);
  pin_vmem();
  if (DD) dw2.preload(PK ? d.pf2 : d.w2, 64, d.b2, nullptr, 0);
  else dw1.preload(PK ? d.pf1 : d.w1 + ADIM, 260, d.b1, d.w1, ADIM);
  if (!DD) {
#pragma unroll
    for (int u = 0; u < R * 64 / kThreads; ++u) {
      const int idx = threadIdx.x + u * kThreads;
      *reinterpret_cast<f32x4*>(XC + (idx >> 6) * 260 + 4 * (idx & 63)) = ct[u];
    }
  }
  XT[threadIdx.x] = av;
  // G's saved activations: PRE keeps them in registers until D'.fc1 is done (their loads are issued now, behind
  // the input tile, and nothing waits for them here)
  TileRegs<1, 128> t1;
  TileRegs<1, 256> t4;
  TileRegs<1, 128> t3;
  TileRegs<1, 64> t2;
  W1A[threadIdx.x] = w1a;
  W4S[threadIdx.x] = w4r;
  if (threadIdx.x == 0) W4S[256] = b4r;
  *reinterpret_cast<f32x4*>(W5S + 4 * threadIdx.x) = w5r;
  if (PRE) {
    t1.load(a.gh1 + row0 * 128, 128);
    t4.load(a.gh4 + row0 * 256, 256);
    t3.load(a.gh3 + row0 * 128, 128);
    t2.load(a.gh2 + row0 * 64, 64);
    pin_vmem();
  } else {
    load_tile<1, 128>(H1, 132, a.gh1 + row0 * 128, 128);                                 // needed last; region is free
  }
  __syncthreads();
  NDP_STAMP(1);
  if (DD) {
    code_rows_dot<64, NRR>(d.pf1, XC, XC + NRR * 260);
    __syncthreads();
    fc1_from_code_rows<64, ACT_LRELU, NRR>(XC + NRR * 260, d.b1, d.w1, 260, ADIM, XT, B2, 68,
                                           (int)(row0 % a.code_rep), a.code_rep);        // D.h1 -> B2
  } else {
    dw2.bind(PK ? d.pf2 : d.w2, 64, d.b2, nullptr, 0);
    layer_fwd_run<1, 256, 64, ACT_LRELU, 4, PK>(dw1, XC, 260, B2, 68, XT, TAILLD, dw2);   // D.h1 -> B2
  }
  if (PRE) {                                                                             // regions of their own: no hazard
    t1.store(H1, 132);
    t4.store(G4, 260);
    t3.store(G3, 132);
    t2.store(G2, 68);
  }
  __syncthreads();
  FwdW<128, 256, 4, PK, RG> dw3;
  dw3.bind(PK ? d.pf3 : d.w3, 128, d.b3, nullptr, 0);
  layer_fwd_run<1, 64, 128, ACT_LRELU, 4, PK>(dw2, B2, 68, B1, 132, nullptr, 0, dw3);   // D.h2 -> B1
  __syncthreads();
  DgW<128, 256, PK, RG> dg3;
  dg3.bind(PK ? d.pg3 : d.w3, 128);
  layer_fwd_run<1, 128, 256, ACT_LRELU, 4, PK>(dw3, B1, 132, XC, 260, nullptr, 0, dg3); // D.h3 -> XC
  __syncthreads();
  NDP_STAMP(2);
  layer_fwd_narrow<1, 256, 1>(XC, 260, W4S, W4S + 256, L, 1);
  __syncthreads();
  float lsum = 0.f;
  if (threadIdx.x < R) {
    const int64_t row = row0 + threadIdx.x;
    const float x = L[threadIdx.x];
    float dl = 0.f;
    if (row < a.m) {                                             // target 1: G wants D fooled (train_gan.py:187-190)
      lsum = fmaxf(x, 0.f) - x + log1pf(expf(-fabsf(x)));
      dl = (1.f / (1.f + expf(-x)) - 1.f) * a.inv_m;
    }
    DL[threadIdx.x] = dl;
  }
  if (threadIdx.x < 64) {               // the 16 rows' losses all sit in wave 0: no block-wide reduction
    const float tot = wave_sum(lsum);
    if (threadIdx.x == 0) a.loss_partials[blockIdx.x] = tot;
  }
  __syncthreads();                      // DL
  NDP_STAMP(3);
  layer_dgrad_narrow<1, 256, 1, ACT_LRELU>(DL, 1, W4S, XC, 260);                        // XC := D.dY3
  __syncthreads();
  DgW<64, 128, PK, RG> dg2;
  dg2.bind(PK ? d.pg2 : d.w2, 64);
  layer_dgrad_run<1, 128, 256, ACT_LRELU, PK>(dg3, XC, 260, B1, 132, dg2);              // B1 := D.dY2
  __syncthreads();
  DgW<128, 256, PK, RG> gg4;                                                             // G's backward starts streaming
  gg4.bind(PK ? g.pg4 : g.w4, 128);
  layer_dgrad_run<1, 64, 128, ACT_LRELU, PK>(dg2, B1, 132, B2, 68, gg4);                // B2 := D.dY1
  __syncthreads();
  NDP_STAMP(4);
  // dLoss/d action_hat = D.dY1 . W1[:, 0:4] (+ NDiv gradient) -> DA and dy5
  {
    // all 256 threads: (row i, action column j, quarter `part` of the 64 terms), quarters added by two shuffles
    // (one wave of 64 threads doing 64-term dot products was 0.9 us of a 21 us kernel)
    const int i = threadIdx.x >> 4, j = (threadIdx.x >> 2) & 3, part = threadIdx.x & 3;
    const int64_t row = row0 + i;
    const float ng = (part == 0 && row < a.m && a.nd_grad != nullptr) ? a.nd_grad[row * ADIM + j] : 0.f;
    float s = 0.f;
#pragma unroll
    for (int o = 16 * part; o < 16 * part + 16; ++o) s = fmaf(B2[i * 68 + o], W1A[o * 4 + j], s);
    s += __shfl_xor(s, 1, 64);
    s += __shfl_xor(s, 2, 64);
    if (part == 0) {
      s = row < a.m ? s + ng : 0.f;
      DA[i * ADIM + j] = s;
      a.dy5[row * ADIM + j] = s;
    }
  }
  __syncthreads();
  NDP_STAMP(5);

  // ---------------- G backward data path (!PRE: regions XC, B1, B2 are free again and take G's activations)
  if (!PRE) {
    load_tile<1, 256>(G4, 260, a.gh4 + row0 * 256, 256);
    load_tile<1, 128>(G3, 132, a.gh3 + row0 * 128, 128);
    load_tile<1, 64>(G2, 68, a.gh2 + row0 * 64, 64);
    __syncthreads();
  }
  NDP_STAMP(6);
  layer_dgrad_narrow<1, 256, 4, ACT_RELU>(DA, 4, W5S, G4, 260);                         // G4 := G.dY4
  __syncthreads();
  NDP_STAMP(7);
  DgW<64, 128, PK, RG> gg3;
  gg3.bind(PK ? g.pg3 : g.w3, 64);
  layer_dgrad_run<1, 128, 256, ACT_RELU, PK>(gg4, G4, 260, G3, 132, gg3);               // G3 := G.dY3
  __syncthreads();
  NDP_STAMP(8);
  DgW<128, 64, PK, RG> gg2;
  gg2.bind(PK ? g.pg2 : g.w2, 128);
  store_tile<1, 256>(a.dy4 + row0 * 256, 256, G4, 260);
  layer_dgrad_run<1, 64, 128, ACT_RELU, PK>(gg3, G3, 132, G2, 68, gg2);                 // G2 := G.dY2
  __syncthreads();
  NDP_STAMP(9);
  store_tile<1, 128>(a.dy3 + row0 * 128, 128, G3, 132);
  layer_dgrad_run<1, 128, 64, ACT_RELU, PK>(gg2, G2, 68, H1, 132);                      // H1 := G.dY1
  __syncthreads();
  NDP_STAMP(10);
  store_tile<1, 64>(a.dy2 + row0 * 64, 64, G2, 68);
  store_tile<1, 128>(a.dy1 + row0 * 128, 128, H1, 132);
  store_segment_sums<128, 1>(H1, 132, a.dy1seg, row0, a.code_rep, a.seg_s, a.seg_entries, (int)blockIdx.x, (int)gridDim.x);
  NDP_STAMP(11);
  NDP_STAMP_FLUSH(12, 6);
}

// Adam state word: {int32 step, float lr/(1-b1^t), float sqrt(1-b2^t), pad}.  One thread
// advances it ahead of the kernel that applies the update, so that the fp64 pow() runs once
// per step instead of once per workgroup.
__device__ __forceinline__ void adam_advance(int32_t* state, float lr, float b1, float b2) {
  const int32_t t = state[0] + 1;
  state[0] = t;
  const double bc1 = 1.0 - pow((double)b1, (double)t);
  const double bc2 = 1.0 - pow((double)b2, (double)t);
  reinterpret_cast<float*>(state)[1] = (float)((double)lr / bc1);
  reinterpret_cast<float*>(state)[2] = (float)sqrt(bc2);
}

// ================================================================ weight gradients
// One job = one block of one layer's dW = sum_rows dY[row][j] * X[row][k]:
//   FULL      64 j x 64 k, both operands read as float4 (j = 4c+u, k = 4c+v permuted tiles)
//   SKINNY_B  64 j x (<=16) k   (the action / noise input columns)
//   SKINNY_A  (<=16) j x 64 k   (the 4- and 1-wide output layers)
// Workgroup (job, chunk): the 4 waves take interleaved 4-row steps of the chunk's rows,
// accumulate in registers, then add their four results through LDS and store one slab.
//   WIDE      256 j x 128 k: a whole layer's dW (D.fc3, G.fc4) per workgroup, operands staged through LDS (large M)
enum { WG_FULL = 0, WG_SKINNY_B = 1, WG_SKINNY_A = 2, WG_WIDE = 3 };
#ifndef NDP_WGRAD_PFG
#define NDP_WGRAD_PFG 8     // 4-row steps in the operand prefetch ring
#endif

struct WgradJob {
  const float* A;      // dY block: A[row*lda + j]
  const float* B;      // X block:  B[brow*ldb + k], brow = min((row % b_rowmod) / b_rowdiv, b_rowmax)
  int lda, ldb;
  int a_cols, b_cols;
  int b_rowmod, b_rowdiv, b_rowmax;
  int b_vec;           // B rows 16-byte aligned
  int dst_off, dst_ld; // slab[dst_off + j*dst_ld + k]
  int bias_off;        // slab[bias_off + j] = sum_rows dY[row][j], or -1
  int kind;
  // K-deduplicated jobs (the code columns of fc1): A holds per-tile SEGMENT sums of dY1 -- the rows
  // of a 16-row tile that share a FLAT row (same code) are pre-summed by the phase kernel, real and
  // fake pass together -- so the job runs over rows = ntiles*seg_s entries instead of all M rows;
  // entry e = tile*seg_s + s pairs with code row (16*tile)/seg_k + s.  rows == 0: plain job.
  int rows, seg, seg_s, seg_k;   // rows != 0: the job's own row space; seg: rows are segment-sum entries
  int seg_wrap;        // seg jobs: entries from seg_wrap on are plain rows -- entry seg_wrap + f pairs with code row f
                       // (the deduplicated real pass of the D step, k_phase_a); 0: none
  int nchunks;         // row chunks (= slabs) of THIS job, <= WgradArgs::nchunks: the full 64 x 64 jobs take more
};
constexpr int kMaxJobs = 28;
struct WgradHead {     // what every workgroup reads: first in the argument struct, a few cache lines (see k_wgrad)
  int njobs;           // 64 x 64 (and skinny) jobs: job[0 .. njobs)
  int nwide;           // WIDE jobs: job[njobs .. njobs + nwide), each over wide_chunks row chunks; their workgroups come
  int wide_chunks;     // FIRST in the grid (the longest); nwide * wide_chunks is a multiple of 8
  int rows;            // total rows (multiple of 16)
  float* slabs;        // [nchunks][slab_stride]
  int64_t slab_stride;
  int32_t* bump;       // Adam state {int32 step; float step_size; float bc2_sqrt; pad} to advance, or null
  float lr, beta1, beta2;
  int nchunks;         // maximum over the jobs: grid and slab count
  NdivArgs nd;         // blocks njobs*nchunks.. : NDiv (cx <= 4, cz <= 2) riding in this launch; nd.n == 0: none
};
struct WgradArgs : WgradHead {
  WgradJob job[kMaxJobs];
  int net_is_g;        // host side only: kernel timing label
  int nreg, reg_begin[4], reg_end[4];   // host side only: layers whose jobs are "light" (fewer chunks)
  int wide_begin, wide_end;             // host side only: the parameter range the WIDE job writes
};

#ifdef NDP_STAMPS
#define NDP_WSTAMP(i) do { if (threadIdx.x == 0) { wst[2 * (i)] = clock64(); wst[2 * (i) + 1] = wall_clock64(); } } while (0)
#else
#define NDP_WSTAMP(i) do { } while (0)
#endif

template <int MT, int NT, int KIND>
__device__ __forceinline__ void wgrad_block(const WgradJob& jb, int rbeg, int rend, int row_last,
                                            float* slab, float* smem, unsigned long long* wst) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = lane & 15, q = lane >> 4;
  f32x4 acc[MT][NT];
  float bs[MT];
#pragma unroll
  for (int u = 0; u < MT; ++u) {
    bs[u] = 0.f;
#pragma unroll
    for (int v = 0; v < NT; ++v) acc[u][v] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  // Main loop.  A wave takes rows rbeg + 16 t + 4 wave + q of step t (nsteps steps).  Operands
  // go through a register ring of PF steps: slot p is consumed by step t (t % PF == p) and
  // refilled at once with step t + PF, so every load is issued PF*16 MFMAs before its use.
  // Every load is UNconditional (rows past the end are clamped to a valid row and simply never
  // consumed; narrow operands are loaded from a clamped column and zeroed by a select): a
  // branch around a load makes hipcc fall back to s_waitcnt vmcnt(0) inside the loop.
  constexpr int PF = NDP_WGRAD_PFG;
  const bool seg_b = jb.seg != 0;
  const bool plain_b = !seg_b && jb.b_rowdiv == 1 && jb.b_rowmod == 0x7fffffff;         // uniform per workgroup
  const float b_inv = 1.0f / (float)(seg_b ? jb.seg_k : jb.b_rowdiv);
  const float s_inv = 1.0f / (float)(seg_b ? jb.seg_s : 1);
  const int nsteps = rend > rbeg ? (rend - rbeg) >> 4 : 0;
  const int row_first = rbeg + 4 * wave + q;
  const int ca = KIND == WG_SKINNY_A ? (c < jb.a_cols ? c : jb.a_cols - 1) : 4 * c;
  const int cb = KIND == WG_SKINNY_B ? (c < jb.b_cols ? c : jb.b_cols - 1) : 4 * c;
  const bool a_on = KIND != WG_SKINNY_A || c < jb.a_cols;
  const bool b_on = KIND != WG_SKINNY_B || c < jb.b_cols;
  f32x4 ra[PF], rb[PF];
  auto load_a = [&](int t) -> f32x4 {
    int row = row_first + 16 * t;
    row = row < row_last ? row : row_last;
    const float* p = jb.A + (size_t)row * jb.lda + ca;
    if (KIND == WG_SKINNY_A) {
      const float v = *p;
      return f32x4{a_on ? v : 0.f, 0.f, 0.f, 0.f};
    }
    return *reinterpret_cast<const f32x4*>(p);
  };
  auto load_b = [&](int t) -> f32x4 {
    int row = row_first + 16 * t;
    row = row < row_last ? row : row_last;
    // B row: min(((row mod b_rowmod) / b_rowdiv), b_rowmax).  Most jobs read B by the plain row;
    // the others (codes, broadcast over the K samples and both passes) wrap at most once and
    // divide through a float reciprocal + fix-up (rows < 2^24): an integer division costs ~40
    // VALU issues per step and un-hides the MFMAs.
    int brow = row;
    if (seg_b) {
      // entry -> (tile, s) -> code row (16*tile)/K + s; exact divisions by reciprocal + fix-up
      const bool ident = jb.seg_wrap != 0 && row >= jb.seg_wrap;
      const int e = ident ? 0 : row;
      int tile = (int)((float)e * s_inv);
      tile = tile * jb.seg_s > e ? tile - 1 : tile;
      tile = (tile + 1) * jb.seg_s <= e ? tile + 1 : tile;
      const int x = 16 * tile;
      int f0 = (int)((float)x * b_inv);
      f0 = f0 * jb.seg_k > x ? f0 - 1 : f0;
      f0 = (f0 + 1) * jb.seg_k <= x ? f0 + 1 : f0;
      brow = ident ? row - jb.seg_wrap : f0 + (e - tile * jb.seg_s);
    } else if (!plain_b) {
      const int x = row >= jb.b_rowmod ? row - jb.b_rowmod : row;
      int qd = (int)((float)x * b_inv);
      qd = qd * jb.b_rowdiv > x ? qd - 1 : qd;
      qd = (qd + 1) * jb.b_rowdiv <= x ? qd + 1 : qd;
      brow = qd;
    }
    brow = brow < jb.b_rowmax ? brow : jb.b_rowmax;
    const float* p = jb.B + (size_t)brow * jb.ldb + cb;
    if (KIND == WG_SKINNY_B) {
      const float v = *p;
      return f32x4{b_on ? v : 0.f, 0.f, 0.f, 0.f};
    }
    if (jb.b_vec) return *reinterpret_cast<const f32x4*>(p);
    return f32x4{p[0], p[1], p[2], p[3]};
  };
  auto mma_step = [&](const f32x4& va, const f32x4& vb) {
#pragma unroll
    for (int u = 0; u < MT; ++u) {
      bs[u] += va[u];
#pragma unroll
      for (int v = 0; v < NT; ++v) acc[u][v] = mfma16(va[u], vb[v], acc[u][v]);
    }
  };
  NDP_WSTAMP(1);
#pragma unroll
  for (int p = 0; p < PF; ++p) {
    ra[p] = load_a(p);
    rb[p] = load_b(p);
  }
  pin_vmem();
  int t = 0;
  for (; t + PF <= nsteps; t += PF) {
#pragma unroll
    for (int p = 0; p < PF; ++p) {
      const f32x4 va = ra[p], vb = rb[p];
      ra[p] = load_a(t + p + PF);
      rb[p] = load_b(t + p + PF);
      pin_vmem();
      mma_step(va, vb);
    }
  }
  const int rem = nsteps - t;                    // < PF; slots 0..rem-1 hold those steps
#pragma unroll
  for (int p = 0; p < PF; ++p)
    if (p < rem) mma_step(ra[p], rb[p]);
  NDP_WSTAMP(2);
  // ---- cross-wave reduction through LDS: tile image [JR][KC] per wave
  constexpr int JR = KIND == WG_SKINNY_A ? 16 : 64;
  constexpr int KC = KIND == WG_SKINNY_B ? 16 : 64;
  float* mine = smem + wave * (JR * KC);
  float* bsh = smem + kWaves * (JR * KC);   // [4][64] bias sums
#pragma unroll
  for (int u = 0; u < MT; ++u)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      // C tile (u,v): lane holds M-row m' = 4q+i, N-col n' = c
      const int j = KIND == WG_SKINNY_A ? (4 * q + i) : (4 * (4 * q + i) + u);
      if (KIND == WG_SKINNY_B) {
        mine[j * KC + c] = acc[u][0][i];
      } else {
        f32x4 t;
#pragma unroll
        for (int v = 0; v < NT; ++v) t[v] = acc[u][v][i];
        *reinterpret_cast<f32x4*>(mine + j * KC + 4 * c) = t;
      }
    }
#pragma unroll
  for (int u = 0; u < MT; ++u) {
    float s = bs[u];
    s += __shfl_xor(s, 16, 64);
    s += __shfl_xor(s, 32, 64);
    if (q == 0) bsh[wave * 64 + (KIND == WG_SKINNY_A ? c : 4 * c + u)] = s;
  }
  __syncthreads();
  NDP_WSTAMP(3);
  const int jn = jb.a_cols, kn = jb.b_cols;
  for (int e = threadIdx.x; e < JR * KC / 4; e += kThreads) {
    const int j = e / (KC / 4), k = 4 * (e % (KC / 4));
    f32x4 s = *reinterpret_cast<const f32x4*>(smem + j * KC + k);
#pragma unroll
    for (int w = 1; w < kWaves; ++w) s += *reinterpret_cast<const f32x4*>(smem + w * (JR * KC) + j * KC + k);
    if (j < jn) {
      float* d = slab + jb.dst_off + (size_t)j * jb.dst_ld + k;
      if (KIND != WG_SKINNY_B && ((jb.dst_ld | jb.dst_off) & 3) == 0) {
        *reinterpret_cast<f32x4*>(d) = s;
      } else {
#pragma unroll
        for (int t = 0; t < 4; ++t)
          if (k + t < kn) d[t] = s[t];
      }
    }
  }
  if (jb.bias_off >= 0 && threadIdx.x < 64 && (int)threadIdx.x < jn) {
    const int j = threadIdx.x;
    slab[jb.bias_off + j] = bsh[j] + bsh[64 + j] + bsh[128 + j] + bsh[192 + j];
  }
}

// WIDE job: dW [256 x 128] = sum over the chunk's rows of dY[row][0..256) x X[row][0..128), one workgroup, every wave
// over ALL rows: wave w owns the 64 j of tile w and all 128 k (32 accumulator tiles = 128 registers).  The register-fed
// 64 x 64 blocks above fetch 2 KB per 16 MFMAs and wave -- 16 B/clk per CU from L2, which is what the memory pipe gives
// beyond L1 (counters at B = 128 / K = 32: matrix pipe 28 % busy, L2 hit rate 48 %, the eight jobs of this layer read
// every activation byte 2.7 times).  Here slabs of 16 rows (24 KB: 16 x 256 floats of dY, 16 x 128 of X) go global ->
// LDS by LDS-DMA (global_load_lds_dwordx4: no register round trip), three buffers, TWO slabs in flight while one is
// multiplied (a slab is 128 MFMAs per wave, ~4,100 cycles; an HBM miss under load is longer than that): a counted
// s_waitcnt leaves the next slab's loads pending across the one raw barrier per slab.  6 B/clk per CU from L2.
// Chunks are whole slabs (rows are padded to 32 and chunk bounds to 16), so no row needs masking.
// JW = 64-row tiles of dW per workgroup: 4 -- the whole 256 x 128 layer, wave w owns tile w and both 64-column halves (32
// accumulator tiles); 2 -- half of it (128 x 128: two jobs per layer), wave w owns tile w >> 1 and column half w & 1 (16
// accumulator tiles, three workgroups per CU): at the same workgroup duration the layer then needs half the row chunks,
// i.e. half the slabs for k_reduce_adam to add.
constexpr int kWideR = 16;
template <int JW> constexpr int wide_slab_floats() { return kWideR * (64 * JW + 128); }
constexpr int wgrad_wide_lds_floats() { return 3 * wide_slab_floats<4>(); }
template <int JW>
__device__ __forceinline__ void wgrad_wide(const WgradJob& jb, int rbeg, int rend, float* slab, float* smem) {
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  typedef const __attribute__((address_space(1))) void* glb_ptr_t;
  constexpr int AW = 64 * JW;                       // dY columns of this job
  constexpr int NK = JW == 4 ? 2 : 1;               // 64-column halves of X per wave
  constexpr int SLAB = wide_slab_floats<JW>();
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int c = lane & 15, q = lane >> 4;
  const int jt = JW == 4 ? wave : (wave >> 1), kt0 = JW == 4 ? 0 : (wave & 1);
  f32x4 acc[4][4 * NK];
  float bs[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    bs[u] = 0.f;
#pragma unroll
    for (int v = 0; v < 4 * NK; ++v) acc[u][v] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const int nslabs = rend > rbeg ? (rend - rbeg) / kWideR : 0;
  // one wave-instruction writes 1 KB of LDS, lane i at base + 16 i: a whole 256-float dY row (JW = 4; wave w: rows w, w +
  // 4, w + 8, w + 12) or two 128-float rows (dY at JW = 2, X always; wave w: row pairs w and w + 4, lanes 0..31 the first
  // row of the pair)
  auto fetch = [&](int s) {
    float* buf = smem + (s % 3) * SLAB;
    const size_t r0 = (size_t)rbeg + (size_t)s * kWideR;
    if (JW == 4) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = wave + 4 * i;
        __builtin_amdgcn_global_load_lds((glb_ptr_t)(jb.A + (r0 + row) * jb.lda + 4 * lane), (lds_ptr_t)(buf + row * AW), 16, 0, 0);
      }
    } else {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int pair = wave + 4 * i;
        __builtin_amdgcn_global_load_lds((glb_ptr_t)(jb.A + (r0 + 2 * pair + (lane >> 5)) * jb.lda + 4 * (lane & 31)),
                                         (lds_ptr_t)(buf + pair * 256), 16, 0, 0);
      }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int pair = wave + 4 * i;
      __builtin_amdgcn_global_load_lds((glb_ptr_t)(jb.B + (r0 + 2 * pair + (lane >> 5)) * jb.ldb + 4 * (lane & 31)),
                                       (lds_ptr_t)(buf + kWideR * AW + pair * 256), 16, 0, 0);
    }
  };
  if (nslabs > 0) fetch(0);
  if (nslabs > 1) fetch(1);
  for (int s = 0; s < nslabs; ++s) {
    // slab s has landed (this wave's share: all but the loads of slab s + 1, if any), for every wave: barrier.  The
    // barrier also says every wave is done reading the buffer of slab s - 1, which slab s + 2 then overwrites.
    if (s + 1 < nslabs) {
      if (JW == 4) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    if (s + 2 < nslabs) fetch(s + 2);
    const float* As = smem + (s % 3) * SLAB + 64 * jt + 4 * c;
    const float* Bs = smem + (s % 3) * SLAB + kWideR * AW + 64 * kt0 + 4 * c;
#pragma unroll
    for (int st = 0; st < kWideR / 4; ++st) {
      const f32x4 va = *reinterpret_cast<const f32x4*>(As + (4 * st + q) * AW);
      f32x4 vb[NK];
#pragma unroll
      for (int k = 0; k < NK; ++k) vb[k] = *reinterpret_cast<const f32x4*>(Bs + (4 * st + q) * 128 + 64 * k);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (kt0 == 0) bs[u] += va[u];
#pragma unroll
        for (int k = 0; k < NK; ++k)
#pragma unroll
          for (int v = 0; v < 4; ++v) acc[u][4 * k + v] = mfma16(va[u], vb[k][v], acc[u][4 * k + v]);
      }
    }
  }
  // C tile (u, v) of column half k: the lane holds dW row j = 64 jt + 4 (4 q + i) + u, columns 64 (kt0 + k) + 4 c + v
#pragma unroll
  for (int u = 0; u < 4; ++u)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float* d = slab + jb.dst_off + (size_t)(64 * jt + 4 * (4 * q + i) + u) * jb.dst_ld + 64 * kt0 + 4 * c;
#pragma unroll
      for (int k = 0; k < NK; ++k)
        *reinterpret_cast<f32x4*>(d + 64 * k) = f32x4{acc[u][4 * k][i], acc[u][4 * k + 1][i], acc[u][4 * k + 2][i], acc[u][4 * k + 3][i]};
    }
  if (jb.bias_off >= 0 && kt0 == 0) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      float sum = bs[u];
      sum += __shfl_xor(sum, 16, 64);
      sum += __shfl_xor(sum, 32, 64);
      if (q == 0) slab[jb.bias_off + 64 * jt + 4 * c + u] = sum;
    }
  }
}

constexpr int wgrad_lds_floats() { return kWaves * 64 * 64 + kWaves * 64; }
// dynamic LDS of k_wgrad_wide: the 64 x 64 jobs' cross-wave sum image or the WIDE job's three slabs
constexpr int wgrad_wide_launch_lds_floats() { return wgrad_wide_lds_floats() > wgrad_lds_floats() ? wgrad_wide_lds_floats() : wgrad_lds_floats(); }

// Arguments of k_wgrad (load_kernargs explains why): the head's cache lines and the two lines of this workgroup's 88-byte
// job entry are requested together, one wait; head and entry are then copied out through a pointer the compiler cannot
// trace to the kernarg segment.  The 2.5 KB table itself is never read as a whole.
__device__ __forceinline__ void wgrad_fetch_args(WgradHead& a, WgradJob& jb, int& job_id, int& chunk) {
  typedef const __attribute__((address_space(4))) char* kbytes_t;
  static_assert(sizeof(WgradHead) <= 192 && offsetof(WgradArgs, job) % 8 == 0, "k_wgrad argument prefetch");
    kbytes_t kp = (kbytes_t)__builtin_amdgcn_kernarg_segment_ptr();
    uint32_t h0, h1, h2, hw0, hw1;
    asm volatile("s_load_dword %0, %5, 0x0\n\ts_load_dword %1, %5, 0x40\n\ts_load_dword %2, %5, 0x80\n\t"
                 "s_load_dword %3, %5, 0x4\n\ts_load_dword %4, %5, 0x8\n\ts_waitcnt lgkmcnt(0)"
                 : "=&s"(h0), "=&s"(h1), "=&s"(h2), "=&s"(hw0), "=&s"(hw1) : "s"(kp) : "memory");
    const int njobs = (int)h0;                        // WgradHead::njobs, nwide, wide_chunks are the first three words
    const int wide_blocks = (int)hw0 * (int)hw1;
    if ((int)blockIdx.x < wide_blocks) {
      job_id = njobs + (int)blockIdx.x / (int)hw1;
      chunk = (int)blockIdx.x % (int)hw1;
    } else {
      const int b = (int)blockIdx.x - wide_blocks;
      const int idx = b >> 3;
      job_id = idx % njobs;
      chunk = (b & 7) + 8 * (idx / njobs);
    }
    const uint32_t joff = (uint32_t)(offsetof(WgradArgs, job) + (size_t)job_id * sizeof(WgradJob));
    uint32_t j0, j1;
    asm volatile("s_load_dword %0, %2, %3\n\ts_load_dword %1, %2, %4\n\ts_waitcnt lgkmcnt(0)"
                 : "=&s"(j0), "=&s"(j1) : "s"(kp), "s"(joff), "s"(joff + (uint32_t)sizeof(WgradJob) - 4u) : "memory");
    uint64_t v = (uint64_t)kp;
    asm volatile("" : "+s"(v) : "s"(h0 | h1 | h2 | j0 | j1));
    __builtin_memcpy(&a, (kbytes_t)v, sizeof(WgradHead));
    __builtin_memcpy(&jb, (kbytes_t)v + joff, sizeof(WgradJob));
}

// WIDE: the instantiation that can run WG_WIDE jobs (large M: its 128 accumulator registers leave one workgroup per CU;
// the other keeps two)
template <bool WIDE>
__device__ __forceinline__ void wgrad_body(float* smem) {
  WgradHead a;
  WgradJob jb;
  int job_id, chunk;
  wgrad_fetch_args(a, jb, job_id, chunk);
  // Linear grid with an XCD-aware order: workgroups are dealt round-robin over the 8 XCDs
  // (b % 8 labels the XCD; speed only, never correctness), and every job of a row chunk reads
  // the same activation rows, so all jobs of chunk c are given to XCD c % 8: the rows then cross
  // the fabric once per chunk and the other 18..25 jobs hit that XCD's L2 (PMC: FETCH_SIZE of
  // k_wgrad[D] 55 MB with the plain order, where each XCD fetched every chunk).
  // Blocks past the wgrad slots are NDiv blocks (the NDiv loss/gradient needs only action_hat
  // and the noise: it runs beside the D weight gradients instead of as a kernel of its own).
  const int slots = a.nwide * a.wide_chunks + 8 * ((a.nchunks + 7) / 8) * a.njobs;
  if ((int)blockIdx.x >= slots) {
    ndiv_block<4, 2>(a.nd, (int)blockIdx.x - slots, smem);
    return;
  }
  if (chunk >= jb.nchunks) return;
  // a job's row space: all rows of the step (a.rows), or its own (segment-sum jobs)
  const int jrows = jb.rows != 0 ? jb.rows : a.rows;
  const int jrpc = ((jrows + jb.nchunks - 1) / jb.nchunks + 15) & ~15;
  const int rbeg = chunk * jrpc;
  int rend = rbeg + jrpc;
  rend = rend < jrows ? rend : jrows;
  float* slab = a.slabs + (size_t)chunk * a.slab_stride;
  if (a.bump != nullptr && blockIdx.x == 0 && threadIdx.x == 0)   // block 0 = (chunk 0, job 0): always live (every job has >= 1 chunk)
    adam_advance(a.bump, a.lr, a.beta1, a.beta2);
  const int kind = jb.kind;   // uniform per workgroup
#ifdef NDP_STAMPS
  __shared__ unsigned long long wst[32];
  NDP_WSTAMP(0);
#else
  unsigned long long* wst = nullptr;
#endif
  if (WIDE && kind == WG_WIDE && jb.a_cols == 256) wgrad_wide<4>(jb, rbeg, rend, slab, smem);
  else if (WIDE && kind == WG_WIDE) wgrad_wide<2>(jb, rbeg, rend, slab, smem);
  else if (kind == WG_FULL) wgrad_block<4, 4, WG_FULL>(jb, rbeg, rend, jrows - 1, slab, smem, wst);
  else if (kind == WG_SKINNY_B) wgrad_block<4, 1, WG_SKINNY_B>(jb, rbeg, rend, jrows - 1, slab, smem, wst);
  else wgrad_block<1, 4, WG_SKINNY_A>(jb, rbeg, rend, jrows - 1, slab, smem, wst);
#ifdef NDP_STAMPS
  NDP_WSTAMP(4);
  if (threadIdx.x == 0 && NDP_STAMP_ON(4))
    for (int i_ = 0; i_ < 10; ++i_) g_stamps[(size_t)blockIdx.x * 64 + i_] = wst[i_];
#endif
}

__global__ __launch_bounds__(kThreads) void k_wgrad(WgradArgs a_segment) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  (void)a_segment;
  wgrad_body<false>(smem);
}
// (two workgroups per CU: at most 256 registers per lane, 128 of them the WIDE job's accumulators)
__global__ __launch_bounds__(kThreads, 2) void k_wgrad_wide(WgradArgs a_segment) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  (void)a_segment;
  wgrad_body<true>(smem);
}

// ================================================================ slab reduce + Adam + losses
struct LossTerm {
  const float* partials; int count; float scale; int slot;   // losses[slot] = scale * sum
};
// Peer-to-peer SUM over ranks (include/ndp.h, "peer-to-peer gradient exchange"). Region layout in
// 4-byte words: status[64] | flags[net 2][src 8][workgroup kP2PBlocks] | inbox D [src 8][parity 2][kP2PCapD]
// | inbox G [src 8][parity 2][kP2PCapG].
constexpr int kP2PRanks = 8;
constexpr int kP2PCapD = 58368;                 // >= 58,305, multiple of kThreads
constexpr int kP2PCapG = 86016;                 // >= Decoder(16) = 85,572
constexpr int kP2PBlocks = kP2PCapG / kThreads; // 336 workgroups at most
constexpr int64_t kP2PFlagOff = 64;
constexpr int64_t kP2PInboxD = kP2PFlagOff + 2ll * kP2PRanks * kP2PBlocks;
constexpr int64_t kP2PInboxG = kP2PInboxD + (int64_t)kP2PRanks * 2 * kP2PCapD;
constexpr int64_t kP2PWords = kP2PInboxG + (int64_t)kP2PRanks * 2 * kP2PCapG;

struct P2PArgs {
  int world, rank, net;                 // world <= 1: no exchange
  long long timeout_ticks;              // wall_clock64 ticks (100 MHz)
  uint32_t* region[kP2PRanks];
};

// one value per thread; every thread of the workgroup must call (barriers inside)
__device__ __forceinline__ float p2p_sum(const P2PArgs& x, float g, int64_t p, bool valid, uint32_t step) {
  const int W = x.world, me = x.rank;
  const int64_t cap = x.net == 0 ? kP2PCapD : kP2PCapG;
  const int64_t inbox = x.net == 0 ? kP2PInboxD : kP2PInboxG;
  const int64_t par = (int64_t)(step & 1u) * cap;
  if (valid) {
    const int64_t at = inbox + (int64_t)me * 2 * cap + par + p;
#pragma unroll
    for (int r = 0; r < kP2PRanks; ++r)
      if (r < W && r != me)
        __hip_atomic_store(reinterpret_cast<float*>(x.region[r]) + at, g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  // No system-scope fence: on this part it writes back / invalidates the whole L2 (measured ~27 ns per wave
  // and fence, x 8 per workgroup x 228-328 workgroups), and nothing here is cached -- the regions are
  // uncached memory and every access below is a system-scope atomic (sc0 sc1: bypasses L1 and L2).  What
  // the protocol needs is that the workgroup's pushes are ACKNOWLEDGED before a flag goes out: the
  // workgroup-scope release is the s_waitcnt vmcnt(0), the barrier collects all four waves.
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (a workgroup-scope release alone does not wait for the acks)
  __syncthreads();
  const int t = threadIdx.x;
  if (t < W && t != me) {
    const int64_t fbase = kP2PFlagOff + (int64_t)x.net * kP2PRanks * kP2PBlocks + blockIdx.x;
    __hip_atomic_store(x.region[t] + fbase + (int64_t)me * kP2PBlocks, step, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    uint32_t* status = x.region[me];
    const uint32_t* flag = x.region[me] + fbase + (int64_t)t * kP2PBlocks;
    if (__hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == 0u) {
      const long long t0 = wall_clock64();
      while ((int32_t)(__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - step) < 0) {
        if (wall_clock64() - t0 > x.timeout_ticks) {     // exit condition every wave reaches
          // the first waiter to give up leaves a record of what it was waiting for in the spare status words
          // (ndp_p2p_diagnostics): {code, workgroup, net, expected step, flag value seen, peer, ticks waited}
          const uint32_t seen = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          uint32_t expected = 0u;
          if (__hip_atomic_compare_exchange_strong(status, &expected, (uint32_t)(1 + t), __ATOMIC_RELAXED,
                                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) {
            __hip_atomic_store(status + 1, (uint32_t)blockIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(status + 2, (uint32_t)x.net, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(status + 3, step, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(status + 4, seen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(status + 5, (uint32_t)t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(status + 6, (uint32_t)(wall_clock64() - t0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          }
          break;
        }
        __builtin_amdgcn_s_sleep(2);
      }
    }
  }
  __syncthreads();                      // the inbox loads below are issued after the flags were seen
  // unconditional loads (a branch around a load makes hipcc wait for each one in turn): slots of ranks
  // that do not exist read this rank's own, unused, slot; p < cap always
  float v[kP2PRanks];
  const float* own = reinterpret_cast<const float*>(x.region[me]) + inbox + par + p;
#pragma unroll
  for (int r = 0; r < kP2PRanks; ++r)
    v[r] = __hip_atomic_load(own + (int64_t)(r < W ? r : me) * 2 * cap, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  float s = 0.f;                        // rank order 0..W-1 on every rank: bit-identical replicas
#pragma unroll
  for (int r = 0; r < kP2PRanks; ++r)
    if (r < W) s += (r == me) ? g : v[r];
  (void)valid;
  return s;
}

struct ReduceArgs {
  const float* slabs; int nchunks; int64_t slab_stride; int64_t n;
  // parameter ranges (whole layers) whose weight-gradient jobs wrote fewer slabs than nchunks
  int nregions; int reg_begin[4], reg_end[4], reg_slabs[4];
  P2PArgs p2p;
  float* grad;                         // [n] or null
  float *params, *exp_avg, *exp_avg_sq;   // Adam (params null -> no update)
  const int32_t* step;                 // Adam state word (already advanced for this step)
  float lr, beta1, beta2, eps;
  LossTerm loss[3]; int nloss;
  float* losses; float* loss_sums;
  PackSpec pack;                       // lane-ordered copies to refresh with the new parameters
};

__device__ __forceinline__ void adam_update(float& p, float g, float& m, float& v,
                                            float step_size, float bc2_sqrt, float b1, float b2, float eps) {
  m = m + (g - m) * (1.f - b1);                 // exp_avg.lerp_(grad, 1 - beta1)
  v = v * b2 + (1.f - b2) * g * g;              // exp_avg_sq.mul_(b2).addcmul_(g, g, 1 - b2)
  const float denom = sqrtf(v) / bc2_sqrt + eps;
  p = p - step_size * (m / denom);              // param.addcdiv_(exp_avg, denom, -step_size)
}

template <bool P2P>
__global__ __launch_bounds__(kThreads) void k_reduce_adam(ReduceArgs a) {
  // (load_kernargs was tried here too: rocprofv3 averages 6.14 / 5.23 us against 5.51 / 5.45 us plain for the G / D
  // launch -- every thread of 330 short-lived workgroups copying the struct costs more than the scalar misses do)
  __shared__ float sh[8];
  const int64_t p = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  const bool live = p < a.n;
  const bool upd = live && a.params != nullptr;
  float g = 0.f;
  // Everything this thread needs is requested before anything is consumed: the Adam operands, and up to 32 slab
  // values at a time (one memory round trip for the 16..24 slabs of the small configurations).  The slabs are
  // added in chunk order: bitwise reproducible.
  float pv = 0.f, m = 0.f, v = 0.f, step_size = 0.f, bc2_sqrt = 1.f;
  if (upd) {
    pv = a.params[p]; m = a.exp_avg[p]; v = a.exp_avg_sq[p];
    step_size = reinterpret_cast<const float*>(a.step)[1];
    bc2_sqrt = reinterpret_cast<const float*>(a.step)[2];
  }
  if (live) {
    const float* sp = a.slabs + p;
    int nch = a.nchunks;
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (r < a.nregions && p >= a.reg_begin[r] && p < a.reg_end[r]) nch = a.reg_slabs[r];
    // (the exchanging variant keeps 16 in flight: its waves hold their registers while they wait for the peers,
    // and with several ranks sharing one GPU in the tests a fat waiting kernel can keep a peer's kernels off the CUs)
    constexpr int NB = P2P ? 16 : 32;
    for (int ch = 0; ch < nch; ch += NB) {
      float t[NB];
#pragma unroll
      for (int u = 0; u < NB; ++u) t[u] = (ch + u < nch) ? sp[(size_t)(ch + u) * a.slab_stride] : 0.f;
#pragma unroll
      for (int u = 0; u < NB; ++u) g += t[u];
    }
  }
  if constexpr (P2P) g = p2p_sum(a.p2p, g, p, live, (uint32_t)a.step[0]);     // sum over ranks
  if (live && a.grad != nullptr) a.grad[p] = g;
  if (upd) {
    adam_update(pv, g, m, v, step_size, bc2_sqrt, a.beta1, a.beta2, a.eps);
    a.params[p] = pv;
    a.exp_avg[p] = m;
    a.exp_avg_sq[p] = v;
    if (a.pack.packed != nullptr) pack_store(a.pack, (int)p, pv);
  }
  // the loss scalars are the LAST block's job: the launch has one block more than the parameters need, so
  // this runs beside the parameter blocks instead of after one of them
  if (blockIdx.x == gridDim.x - 1) {
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      if (t >= a.nloss) break;
      const LossTerm lt = a.loss[t];
      float s = 0.f;
      for (int i = threadIdx.x; i < lt.count; i += kThreads) s += lt.partials[i];
      s = block_sum(s, sh + 4);
      if (threadIdx.x == 0) {
        s *= lt.scale;
        a.losses[lt.slot] = s;
        if (a.loss_sums != nullptr) a.loss_sums[lt.slot] += s;
      }
    }
  }
}

__global__ void k_adam_advance(int32_t* state, float lr, float b1, float b2) { adam_advance(state, lr, b1, b2); }

struct AdamArgs {
  float *params, *exp_avg, *exp_avg_sq; const float* grad; int64_t n;
  const int32_t* step; float lr, beta1, beta2, eps;
  PackSpec pack;
};
__global__ __launch_bounds__(kThreads) void k_adam(AdamArgs a) {
  const float step_size = reinterpret_cast<const float*>(a.step)[1];
  const float bc2_sqrt = reinterpret_cast<const float*>(a.step)[2];
  const int64_t p = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  if (p < a.n) {
    float pv = a.params[p], m = a.exp_avg[p], v = a.exp_avg_sq[p];
    adam_update(pv, a.grad[p], m, v, step_size, bc2_sqrt, a.beta1, a.beta2, a.eps);
    a.params[p] = pv;
    a.exp_avg[p] = m;
    a.exp_avg_sq[p] = v;
    if (a.pack.packed != nullptr) pack_store(a.pack, (int)p, pv);
  }
}

__global__ __launch_bounds__(kThreads) void k_sum_partials(const float* partials, int count,
                                                           float scale, float* out) {
  __shared__ float red[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < count; i += kThreads) s += partials[i];
  s = block_sum(s, red);
  if (threadIdx.x == 0) *out = s * scale;
}

// ================================================================ uniform noise kernel
__global__ __launch_bounds__(kThreads) void k_philox(float* out, int64_t n, uint64_t seed,
                                                     const int32_t* offset_dev) {
  const int64_t idx = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  if (idx >= n) return;
  const uint32_t off = offset_dev != nullptr ? (uint32_t)*offset_dev : 0u;
  out[idx] = philox_uniform((uint64_t)idx, seed, off);
}

}  // namespace ndp

#include "ndp_capi.inc"
#include "ndp_encoder.inc"
#include "ndp_forward_model.inc"
