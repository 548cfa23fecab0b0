// ndp_device.h -- gfx950 (CDNA4) device building blocks for the fused MLP kernels.
//
// Everything is exact fp32: the matrix work runs on v_mfma_f32_16x16x4_f32 (f32 in,
// f32 accumulate, bitwise an fmaf chain -- cdna_hip_programming.md section 3), there is
// no reduced-precision path.
//
// Tiling model (one workgroup = 4 waves = 256 threads = one tile of R = 16*RT rows):
//   * the row tile's activations live in LDS, row-major with stride ld = width + 4
//     floats (ld = 4 mod 32: conflict-free C-layout ds_write_b32, one 2-way slot per
//     ds_read_b128 group);
//   * a layer's weights are read ONCE per workgroup, straight from global/L2 into the
//     MFMA B operand registers: every weight element is needed by exactly one wave
//     (wave w owns output columns), so staging them through LDS would only add a round
//     trip.  Both the forward (reduce over W's row = contiguous dim) and the data
//     gradient (reduce over W's column) read W in its native nn.Linear [out][in]
//     layout with 8/16-byte vector loads, by permuting the reduction index (forward)
//     or the output column index (dgrad) among lanes -- a sum does not care about order.
//
// MFMA 16x16x4 f32 operand maps (lane l: c = l & 15, q = l >> 4):
//   A[i = c][k = q], B[k = q][j = c], C/D: col j = c, rows i = 4q + reg (reg 0..3).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ndp {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int kThreads = 256;
constexpr int kWaves = 4;
constexpr float kLreluSlope = 0.01f;   // F.leaky_relu default (models/gan.py:106-108)

enum Act { ACT_NONE = 0, ACT_RELU = 1, ACT_LRELU = 2 };

__device__ __forceinline__ int lds_ld(int width) { return width + 4; }

template <int ACT>
__device__ __forceinline__ float act_fwd(float v) {
  if (ACT == ACT_RELU) return v > 0.f ? v : 0.f;
  if (ACT == ACT_LRELU) return v > 0.f ? v : v * kLreluSlope;
  return v;
}

// derivative taken from the POST-activation value h (sign-preserving activations):
// relu: threshold_backward passes g where x > 0; lrelu: g where x > 0 else slope*g.
template <int ACT>
__device__ __forceinline__ float act_bwd(float h, float g) {
  if (ACT == ACT_RELU) return h > 0.f ? g : 0.f;
  if (ACT == ACT_LRELU) return h > 0.f ? g : g * kLreluSlope;
  return g;
}

// Scheduling fence that global loads may not cross (everything else may): hipcc otherwise
// sinks prefetch loads back down to just ahead of their first use, which re-exposes the
// L2 latency the prefetch ring is there to hide.  MFMAs may not cross either (or the compiler
// hoists the consumers up between the prefetch loads instead).  Mask = all but VMEM and MFMA
// (LLVM SchedGroupMask: ALU 1, VALU 2, SALU 4, MFMA 8, VMEM 0x10/0x20/0x40, DS 0x80/0x100/0x200, TRANS 0x400).
__device__ __forceinline__ void pin_vmem() { __builtin_amdgcn_sched_barrier(0x0786); }
// ... and one that LDS reads may not cross either: the A operand of k-step t+1 is read from LDS BEFORE the MFMAs of
// step t are issued (two register sets), so its ~100-cycle latency runs under 4..16 MFMAs instead of stalling the
// matrix pipe once per k-step -- without the fence hipcc sinks the read to just ahead of its first use again
// (ISA of round 1: ds_read_b128, s_waitcnt lgkmcnt, MFMAs, ds_read_b128, ... in every layer loop).
__device__ __forceinline__ void pin_vmem_lds() { __builtin_amdgcn_sched_barrier(0x0606); }

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
#ifdef NDP_EXP_NOMFMA     // diagnostic ablation: keep operands live, drop the matrix op
  c[0] += a * b;
  return c;
#else
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
#endif
}

// Position of weight element (row j, main column k) of a layer with IN main inputs and OUT
// outputs in the forward-packed copy: [wave][tile n][k-step t][lane = 16q + c][e].
__device__ __forceinline__ int fwd_pack_offset(int j, int k, int in, int out) {
  const int per_wave = out >> 2, nt = out >> 6, nit = in >> 4;
  const int wave = j / per_wave, n = (j % per_wave) >> 4, c = j & 15;
  const int t = k >> 4, q = (k & 15) >> 2, e = k & 3;
  return (((wave * nt + n) * nit + t) * 64 + 16 * q + c) * 4 + e;
}
// ... and in the data-gradient-packed copy (dX = dY . W: reduce over row j, output column k;
// in = columns of W, out = rows): [wave][step t][lane = 16q + c][s][v], V = in/64.
__device__ __forceinline__ int dgrad_pack_offset(int j, int k, int in, int out) {
  const int v_ = in >> 6, nit = out >> 4;
  const int wave = k / (16 * v_), c = (k % (16 * v_)) / v_, v = k % v_;
  const int t = j >> 4, q = (j & 15) >> 2, s = j & 3;
  return (((wave * nit + t) * 64 + 16 * q + c) * 4 + s) * v_ + v;
}

// Kernel arguments arrive through scalar loads from the kernarg segment.  hipcc places those loads at the kernel's
// entry and, short of scalar registers for a 0.5 KB argument struct, serialises them: load 16 dwords, wait, park
// them in VGPR lanes, load the next 16, wait ... -- five to six DEPENDENT scalar-cache misses (~0.3-0.4 us each at
// kernel start) before the first weight load is issued: 2.3 us of a 26 us kernel (stamps + ISA, round 2).
// load_kernargs<T>() reads the struct through the kernarg pointer instead: first one dword of each of its 64-byte
// lines (independent loads, ONE wait; up to 8 lines = 512 bytes), then the fields -- which now hit the scalar
// cache.  The kernel's by-value parameter only sizes and fills the segment; the body reads the copy.
template <class T>
__device__ __forceinline__ T load_kernargs() {
  typedef const __attribute__((address_space(4))) void* kptr_t;
  kptr_t kp = (kptr_t)__builtin_amdgcn_kernarg_segment_ptr();
  constexpr int LAST = (((int)sizeof(T) - 4) / 64) * 64;          // offset of the last line that holds a field
  constexpr int O1 = 64 < LAST ? 64 : LAST, O2 = 128 < LAST ? 128 : LAST, O3 = 192 < LAST ? 192 : LAST,
                O4 = 256 < LAST ? 256 : LAST, O5 = 320 < LAST ? 320 : LAST, O6 = 384 < LAST ? 384 : LAST,
                O7 = 448 < LAST ? 448 : LAST;
  static_assert(sizeof(T) <= 576, "argument struct too large for the 8-line prefetch");
  uint32_t t0, t1, t2, t3, t4, t5, t6, t7;
  asm volatile(
      "s_load_dword %0, %8, 0x0\n\ts_load_dword %1, %8, %9\n\ts_load_dword %2, %8, %10\n\t"
      "s_load_dword %3, %8, %11\n\ts_load_dword %4, %8, %12\n\ts_load_dword %5, %8, %13\n\t"
      "s_load_dword %6, %8, %14\n\ts_load_dword %7, %8, %15\n\ts_waitcnt lgkmcnt(0)"
      : "=&s"(t0), "=&s"(t1), "=&s"(t2), "=&s"(t3), "=&s"(t4), "=&s"(t5), "=&s"(t6), "=&s"(t7)
      : "s"(kp), "i"(O1), "i"(O2), "i"(O3), "i"(O4), "i"(O5), "i"(O6), "i"(O7)
      : "memory");
  // the struct is read through a pointer the compiler cannot trace back to the kernarg segment, so that none of its
  // loads can be placed ahead of the line prefetch
  uint64_t v = (uint64_t)kp;
  asm volatile("" : "+s"(v) : "s"(t0 | t1 | t2 | t3 | t4 | t5 | t6 | t7));
  T out;
  __builtin_memcpy(&out, (kptr_t)v, sizeof(T));
  return out;
}

template <int WALIGN>
__device__ __forceinline__ f32x4 ldg4(const float* p) {
#ifdef NDP_EXP_NOLOAD     // diagnostic ablation: no weight traffic
  const float v = (float)(reinterpret_cast<uintptr_t>(p) & 1023) * 1e-3f;
  return f32x4{v, v + 1.f, v + 2.f, v + 3.f};
#endif
  if (WALIGN >= 4) {
    return *reinterpret_cast<const f32x4*>(p);
  } else {
    f32x2 lo = *reinterpret_cast<const f32x2*>(p);
    f32x2 hi = *reinterpret_cast<const f32x2*>(p + 2);
    f32x4 r = {lo[0], lo[1], hi[0], hi[1]};
    return r;
  }
}

// ----------------------------------------------------------------------------------
// Y[R x OUT] = act( X[R x IN] . Wm^T  (+ Xt[R x tail_n] . Wt^T)  + bias )
//   X, Xt, Y in LDS; Wm/Wt rows are rows of one nn.Linear weight (row stride ldw).
//   IN % 16 == 0, OUT % 64 == 0.  Wave w owns columns [w*OUT/4, (w+1)*OUT/4).
//   Lane (q,c) loads 4 consecutive reduction indices k = 16t + 4q .. +3 of its weight
//   row and of its activation row; MFMA step s pairs element s of both, so the four
//   lane groups cover k = 16t + {s, 4+s, 8+s, 12+s}: a permutation of the k order.
//   WALIGN: 4 if (Wm + j*ldw) is 16-byte aligned for every j, else 2 (8-byte), else 1.
//   PACKED: Wm is the lane-ordered copy of the main weights (see fwd_pack_offset): the wave's
//   fragment for column tile n, k-step t is ONE contiguous 1 KiB block, lane l at +16 l bytes.
//   With the native [out][in] layout every quad of lanes touches 4 different cache lines
//   (lane = weight row) and the per-CU L1 tag rate, not L2 or the MFMA pipe, sets the pace:
//   measured 12 B/clk/CU native vs ~20 B/clk/CU packed on the 131 KB layers.
//
// The layer is split in two so that a kernel can issue layer l+1's first weight loads BEFORE
// it computes layer l (they fly across the barrier between the layers):
//   FwdW::preload   bias, tail weights and the first PF k-steps of the weight ring -> registers
//   layer_fwd_run   the k-loop (keeps the ring PF steps ahead), tail step, epilogue to LDS
// The loop is fully unrolled, so ring slots are static registers and each load is issued
// ~PF*4*RT*NT MFMAs (>= 1,000 cycles) before its use; an L2 / Infinity-Cache hit costs 500-900.
template <int IN, int OUT, int WALIGN, bool PACKED, int RING = 96>
struct FwdW {
  static constexpr int NT = OUT / 64;
  static constexpr int NIT = IN / 16;
  static constexpr int PF0 = RING / (4 * NT);      // RING = VGPR budget of the prefetch ring
  static constexpr int PF = PF0 < NIT ? PF0 : NIT;
  f32x4 ring[PF][NT];
  f32x4 wtail[NT];
  float bias_r[NT];
  const float* wbase;   // lane's base address (native: its weight row + 4q; packed: + 4*lane)
  int ldw;

  // (Tried, round 2: starting every workgroup's k sweep at a different step, because a microbenchmark of 168-256
  // workgroups streaming the same packed weights in lockstep, 24 fragments deep, gets 20 B/clk/CU against 37-47
  // when they are desynchronised -- scripts/probe/l2_stream.hip.  In the real kernels it changed nothing: with
  // weight loads removed altogether (-DNDP_EXP_NOLOAD) they are no faster either.  The weight stream is not what
  // these kernels wait for.)
  __device__ __forceinline__ f32x4 frag(int n, int t) const {
    if (PACKED) {
      const int wave = threadIdx.x >> 6;
      return ldg4<4>(wbase + (size_t)((wave * NT + n) * NIT + t) * 256);
    }
    return ldg4<WALIGN>(wbase + (size_t)(n * 16) * ldw + 16 * t);
  }

  // The layer's operands, no loads yet.  The loads are issued either at once (preload) or a slice per k-step of the
  // PREVIOUS layer's loop (preload_slice, called by layer_*_run for its `next` argument): a burst of 24+ wave-loads
  // holds the wave at the load instructions for 0.8 us -- the memory pipe accepts them only as fast as the data
  // returns (stamps, round 2) -- during which it issues no MFMA; spread between the MFMAs of a loop they cost nothing.
  const float* bias_p; const float* wt_p; int tail_n_;
  __device__ __forceinline__ void bind(const float* __restrict__ Wm, int ldw_, const float* __restrict__ bias,
                                       const float* __restrict__ Wt, int tail_n) {
    static_assert(IN % 16 == 0 && OUT % 64 == 0, "layer_fwd shape");
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 15, q = lane >> 4;
    const int col0 = wave * (OUT / 4);
    ldw = ldw_;
    wbase = PACKED ? Wm + 4 * lane : Wm + (size_t)(col0 + c) * ldw_ + 4 * q;
    bias_p = bias + col0 + c;
    wt_p = Wt != nullptr ? Wt + (size_t)(col0 + c) * ldw_ + 4 * q : nullptr;
    tail_n_ = tail_n;
  }
  static constexpr int kItems = PF * NT;
  // slice i of nslices: bias and tail weights with slice 0, ring fragment j with slice j * nslices / kItems
  __device__ __forceinline__ void preload_slice(int i, int nslices) {
    const int q = (threadIdx.x & 63) >> 4;
    if (i == 0) {
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        bias_r[n] = bias_p[n * 16];
        wtail[n] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (wt_p != nullptr) {
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (4 * q + j < tail_n_) wtail[n][j] = wt_p[(size_t)(n * 16) * ldw + j];
        }
      }
    }
#pragma unroll
    for (int p = 0; p < PF; ++p)
#pragma unroll
      for (int n = 0; n < NT; ++n)
        if ((p * NT + n) * nslices / kItems == i) ring[p][n] = frag(n, p);
  }
  __device__ __forceinline__ void preload(const float* __restrict__ Wm, int ldw_,
                                          const float* __restrict__ bias,
                                          const float* __restrict__ Wt, int tail_n) {
    bind(Wm, ldw_, bias, Wt, tail_n);
    preload_slice(0, 1);
    pin_vmem();
  }
};

// `next` of a layer that has nothing to prefetch for
struct NoPrefetch {
  __device__ __forceinline__ void preload_slice(int, int) {}
};

template <int RT, int IN, int OUT, int ACT, int WALIGN, bool PACKED, int RING, class NEXT>
__device__ __forceinline__ void layer_fwd_run(FwdW<IN, OUT, WALIGN, PACKED, RING>& w,
                                              const float* X, int ldx, float* Y, int ldy,
                                              const float* Xt, int ldt, NEXT& next) {
  constexpr int NT = OUT / 64, NIT = IN / 16;
  constexpr int PF = FwdW<IN, OUT, WALIGN, PACKED, RING>::PF;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = lane & 15, q = lane >> 4;
  const int col0 = wave * (OUT / 4);
  const float* xp = X + c * ldx + 4 * q;

  // A single accumulator would make every MFMA wait for the previous one (40-cycle dependent latency against a
  // 32-cycle issue): the narrowest layers (one 16 x 16 output tile per wave) alternate between two and add them.
  constexpr int NA = (RT * NT == 1) ? 2 : 1;
  f32x4 acc[NA][RT][NT];
#pragma unroll
  for (int a = 0; a < NA; ++a)
#pragma unroll
    for (int r = 0; r < RT; ++r)
#pragma unroll
      for (int n = 0; n < NT; ++n) acc[a][r][n] = f32x4{0.f, 0.f, 0.f, 0.f};

  // A operand: two register sets, step t+1 (or the tail) is read while step t computes
  f32x4 av[2][RT];
#pragma unroll
  for (int r = 0; r < RT; ++r) av[0][r] = *reinterpret_cast<const f32x4*>(xp + r * 16 * ldx);
#pragma unroll
  for (int t = 0; t < NIT; ++t) {
    f32x4 bv[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) bv[n] = w.ring[t % PF][n];
    if (t + PF < NIT) {
#pragma unroll
      for (int n = 0; n < NT; ++n) w.ring[t % PF][n] = w.frag(n, t + PF);
    }
    next.preload_slice(t, NIT);          // the following layer's first fragments, a slice per k-step
    if (t + 1 < NIT) {
#pragma unroll
      for (int r = 0; r < RT; ++r) av[(t + 1) & 1][r] = *reinterpret_cast<const f32x4*>(xp + r * 16 * ldx + 16 * (t + 1));
    } else if (Xt != nullptr) {
      // the "tail" (<= 16 extra inputs: noise / action) is one more 16-wide k-step with masked
      // weights; Xt rows are zero-padded to 16 in LDS
#pragma unroll
      for (int r = 0; r < RT; ++r) av[(t + 1) & 1][r] = *reinterpret_cast<const f32x4*>(Xt + (r * 16 + c) * ldt + 4 * q);
    }
    pin_vmem_lds();
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int r = 0; r < RT; ++r)
#pragma unroll
        for (int n = 0; n < NT; ++n)
          acc[s % NA][r][n] = mfma16(av[t & 1][r][s], bv[n][s], acc[s % NA][r][n]);
  }
  if (Xt != nullptr) {
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int r = 0; r < RT; ++r)
#pragma unroll
        for (int n = 0; n < NT; ++n)
          acc[s % NA][r][n] = mfma16(av[NIT & 1][r][s], w.wtail[n][s], acc[s % NA][r][n]);
  }
  if (NA == 2) {
#pragma unroll
    for (int r = 0; r < RT; ++r)
#pragma unroll
      for (int n = 0; n < NT; ++n) acc[0][r][n] += acc[NA - 1][r][n];
  }

#pragma unroll
  for (int n = 0; n < NT; ++n) {
    const int col = col0 + n * 16 + c;
#pragma unroll
    for (int r = 0; r < RT; ++r)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = r * 16 + 4 * q + i;
        Y[row * ldy + col] = act_fwd<ACT>(acc[0][r][n][i] + w.bias_r[n]);
      }
  }
}
template <int RT, int IN, int OUT, int ACT, int WALIGN, bool PACKED, int RING>
__device__ __forceinline__ void layer_fwd_run(FwdW<IN, OUT, WALIGN, PACKED, RING>& w,
                                              const float* X, int ldx, float* Y, int ldy,
                                              const float* Xt, int ldt) {
  NoPrefetch none;
  layer_fwd_run<RT, IN, OUT, ACT, WALIGN, PACKED, RING>(w, X, ldx, Y, ldy, Xt, ldt, none);
}

// ----------------------------------------------------------------------------------
// In place: H[R x IN] <- act'(H) * ( dY[R x OUT] . W[OUT x IN] )
//   dY, H in LDS; W = nn.Linear weight [OUT][ldw] whose layer maps IN -> OUT.
//   IN in {64, 128}: wave w owns the 16*V columns [w*16V, (w+1)*16V), V = IN/64; lane c
//   holds columns V*c .. V*c+V-1 of that group (one V-wide vector load per weight row),
//   i.e. MFMA column tile v consists of columns {V*c + v}.  Reduction index j = 16t+4q+s
//   pairs element s of the lane's dY vector with weight row j.
//   PACKED: W is the dgrad_pack_offset copy: lane l of wave w reads its 4V floats of step t
//   at ((w*NIT + t)*64 + l)*4V.
// Split like the forward layer: DgW::preload issues the first PF steps, layer_dgrad_run computes.
template <int IN, int OUT, bool PACKED, int RING = 96>
struct DgW {
  static constexpr int V = IN / 64;
  static constexpr int NIT = OUT / 16;
  static constexpr int PF0 = RING / (4 * V);
  static constexpr int PF = PF0 < NIT ? PF0 : NIT;
  float ring[PF][4][V];
  const float* wbase;
  int ldw;

  __device__ __forceinline__ void load_step(int t, float (&dst)[4][V]) const {
    if (PACKED) {
      const float* p = wbase + (size_t)t * 64 * (4 * V);
#pragma unroll
      for (int h = 0; h < V; ++h) {
        const f32x4 x = *reinterpret_cast<const f32x4*>(p + 4 * h);
#pragma unroll
        for (int e = 0; e < 4; ++e) dst[(4 * h + e) / V][(4 * h + e) % V] = x[e];
      }
      return;
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const float* p = wbase + (size_t)(16 * t + s) * ldw;
      if (V == 2) {
        f32x2 v2 = *reinterpret_cast<const f32x2*>(p);
        dst[s][0] = v2[0];
        dst[s][V - 1] = v2[1];
      } else {
        dst[s][0] = *p;
      }
    }
  }

  __device__ __forceinline__ void bind(const float* __restrict__ W, int ldw_) {
    static_assert((IN == 64 || IN == 128) && OUT % 16 == 0, "layer_dgrad shape");
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 15, q = lane >> 4;
    ldw = ldw_;
    wbase = PACKED ? W + ((size_t)wave * NIT * 64 + lane) * (4 * V)
                   : W + (size_t)(4 * q) * ldw_ + wave * 16 * V + V * c;
  }
  // ring step p with slice p * nslices / PF (FwdW::preload_slice)
  __device__ __forceinline__ void preload_slice(int i, int nslices) {
#pragma unroll
    for (int p = 0; p < PF; ++p)
      if (p * nslices / PF == i) load_step(p, ring[p]);
  }
  __device__ __forceinline__ void preload(const float* __restrict__ W, int ldw_) {
    bind(W, ldw_);
    preload_slice(0, 1);
    pin_vmem();
  }
};

template <int RT, int IN, int OUT, int ACT, bool PACKED, int RING, class NEXT>
__device__ __forceinline__ void layer_dgrad_run(DgW<IN, OUT, PACKED, RING>& w, const float* dY, int ldd,
                                                float* H, int ldh, NEXT& next) {
  constexpr int V = IN / 64, NIT = OUT / 16;
  constexpr int PF = DgW<IN, OUT, PACKED, RING>::PF;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = lane & 15, q = lane >> 4;
  const int colbase = wave * 16 * V + V * c;
  const float* dp = dY + c * ldd + 4 * q;

  constexpr int NA = (RT * V == 1) ? 2 : 1;        // two accumulators where one would serialise the MFMAs (layer_fwd_run)
  f32x4 acc[NA][RT][V];
#pragma unroll
  for (int a = 0; a < NA; ++a)
#pragma unroll
    for (int r = 0; r < RT; ++r)
#pragma unroll
      for (int v = 0; v < V; ++v) acc[a][r][v] = f32x4{0.f, 0.f, 0.f, 0.f};

  f32x4 av[2][RT];                                 // dY operand of step t+1 is read while step t computes
#pragma unroll
  for (int r = 0; r < RT; ++r) av[0][r] = *reinterpret_cast<const f32x4*>(dp + r * 16 * ldd);
#pragma unroll
  for (int t = 0; t < NIT; ++t) {
    float bv[4][V];
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int v = 0; v < V; ++v) bv[s][v] = w.ring[t % PF][s][v];
    if (t + PF < NIT) w.load_step(t + PF, w.ring[t % PF]);
    next.preload_slice(t, NIT);
    if (t + 1 < NIT) {
#pragma unroll
      for (int r = 0; r < RT; ++r) av[(t + 1) & 1][r] = *reinterpret_cast<const f32x4*>(dp + r * 16 * ldd + 16 * (t + 1));
    }
    pin_vmem_lds();
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int r = 0; r < RT; ++r)
#pragma unroll
        for (int v = 0; v < V; ++v) acc[s % NA][r][v] = mfma16(av[t & 1][r][s], bv[s][v], acc[s % NA][r][v]);
  }
  if (NA == 2) {
#pragma unroll
    for (int r = 0; r < RT; ++r)
#pragma unroll
      for (int v = 0; v < V; ++v) acc[0][r][v] += acc[NA - 1][r][v];
  }

#pragma unroll
  for (int r = 0; r < RT; ++r)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float* hp = H + (r * 16 + 4 * q + i) * ldh + colbase;
#pragma unroll
      for (int v = 0; v < V; ++v) hp[v] = act_bwd<ACT>(hp[v], acc[0][r][v][i]);
    }
}
template <int RT, int IN, int OUT, int ACT, bool PACKED, int RING>
__device__ __forceinline__ void layer_dgrad_run(DgW<IN, OUT, PACKED, RING>& w, const float* dY, int ldd,
                                                float* H, int ldh) {
  NoPrefetch none;
  layer_dgrad_run<RT, IN, OUT, ACT, PACKED, RING>(w, dY, ldd, H, ldh, none);
}

// ----------------------------------------------------------------------------------
// Narrow output layers on the VALU: Y[R x OUT] = X[R x IN] . W^T + bias, OUT in {1, 4}.
//   256 threads = R rows x PARTS parts; a part takes every PARTS-th float4 of the row.
//   Result written to Yout (LDS, row stride ldy) by the part-0 lane of each row.
template <int RT, int IN, int OUT>
__device__ __forceinline__ void layer_fwd_narrow(const float* X, int ldx,
                                                 const float* __restrict__ W,
                                                 const float* __restrict__ bias,
                                                 float* Y, int ldy) {
  constexpr int R = 16 * RT;
  constexpr int PARTS = kThreads / R;           // 16 (RT=1) or 8 (RT=2)
  constexpr int NV = IN / 4 / PARTS;            // float4s per part
  static_assert(IN % (4 * PARTS) == 0, "layer_fwd_narrow shape");
  const int row = threadIdx.x / PARTS, part = threadIdx.x % PARTS;
  float sum[OUT];
#pragma unroll
  for (int o = 0; o < OUT; ++o) sum[o] = 0.f;
#pragma unroll
  for (int m = 0; m < NV; ++m) {
    const int k = 4 * (part + PARTS * m);
    const f32x4 x = *reinterpret_cast<const f32x4*>(X + row * ldx + k);
#pragma unroll
    for (int o = 0; o < OUT; ++o) {
      const f32x4 w = *reinterpret_cast<const f32x4*>(W + (size_t)o * IN + k);
      sum[o] = fmaf(x[0], w[0], sum[o]);
      sum[o] = fmaf(x[1], w[1], sum[o]);
      sum[o] = fmaf(x[2], w[2], sum[o]);
      sum[o] = fmaf(x[3], w[3], sum[o]);
    }
  }
#pragma unroll
  for (int o = 0; o < OUT; ++o) {
#pragma unroll
    for (int d = PARTS / 2; d >= 1; d >>= 1) sum[o] += __shfl_xor(sum[o], d, 64);
  }
  if (part == 0) {
#pragma unroll
    for (int o = 0; o < OUT; ++o) Y[row * ldy + o] = sum[o] + bias[o];
  }
}

// In place: H[R x IN] <- act'(H) * ( dY[R x OUT] . W[OUT x IN] ), OUT in {1, 4} (VALU).
template <int RT, int IN, int OUT, int ACT>
__device__ __forceinline__ void layer_dgrad_narrow(const float* dY, int ldd,
                                                   const float* __restrict__ W,
                                                   float* H, int ldh) {
  constexpr int R = 16 * RT;
  constexpr int NV = R * IN / 4;
  for (int idx = threadIdx.x; idx < NV; idx += kThreads) {
    const int row = idx / (IN / 4), k = 4 * (idx % (IN / 4));
    f32x4 g = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int o = 0; o < OUT; ++o) {
      const float d = dY[row * ldd + o];
      const f32x4 w = *reinterpret_cast<const f32x4*>(W + (size_t)o * IN + k);
      g[0] = fmaf(d, w[0], g[0]);
      g[1] = fmaf(d, w[1], g[1]);
      g[2] = fmaf(d, w[2], g[2]);
      g[3] = fmaf(d, w[3], g[3]);
    }
    f32x4* hp = reinterpret_cast<f32x4*>(H + row * ldh + k);
    f32x4 h = *hp;
    h[0] = act_bwd<ACT>(h[0], g[0]);
    h[1] = act_bwd<ACT>(h[1], g[1]);
    h[2] = act_bwd<ACT>(h[2], g[2]);
    h[3] = act_bwd<ACT>(h[3], g[3]);
    *hp = h;
  }
}

// Copy an LDS tile [R x W] (stride ld) to global rows (stride gld), float4 coalesced.
template <int RT, int W>
__device__ __forceinline__ void store_tile(float* __restrict__ dst, size_t gld,
                                           const float* src, int ld) {
  constexpr int R = 16 * RT;
  constexpr int NV = R * W / 4;
#ifdef NDP_EXP_NOSTORE     // diagnostic ablation: what the activation stores cost on the critical path
  if (gld == 0x7fffffff)
#endif
  for (int idx = threadIdx.x; idx < NV; idx += kThreads) {
    const int row = idx / (W / 4), k = 4 * (idx % (W / 4));
    *reinterpret_cast<f32x4*>(dst + (size_t)row * gld + k) =
        *reinterpret_cast<const f32x4*>(src + row * ld + k);
  }
}

// A global tile [16 RT x W] held in registers between its loads (issued early) and its LDS writes (done where
// the data is needed, when the loads have long returned): the early loads do not hold up an earlier barrier.
template <int RT, int W>
struct TileRegs {
  static constexpr int NV = 16 * RT * W / 4;
  static constexpr int N = (NV + kThreads - 1) / kThreads;
  f32x4 v[N];
  __device__ __forceinline__ void load(const float* __restrict__ src, size_t gld) {
#pragma unroll
    for (int u = 0; u < N; ++u) {
      int idx = threadIdx.x + u * kThreads;
      idx = idx < NV ? idx : NV - 1;                      // unconditional load (clamped), see wgrad
      const int row = idx / (W / 4), k = 4 * (idx % (W / 4));
      v[u] = ldg4<4>(src + (size_t)row * gld + k);
    }
  }
  __device__ __forceinline__ void store(float* dst, int ld) const {
#pragma unroll
    for (int u = 0; u < N; ++u) {
      const int idx = threadIdx.x + u * kThreads;
      if (idx < NV) {
        const int row = idx / (W / 4), k = 4 * (idx % (W / 4));
        *reinterpret_cast<f32x4*>(dst + row * ld + k) = v[u];
      }
    }
  }
};

template <int RT, int W>
__device__ __forceinline__ void load_tile(float* dst, int ld,
                                          const float* __restrict__ src, size_t gld) {
  constexpr int R = 16 * RT;
  constexpr int NV = R * W / 4;
  for (int idx = threadIdx.x; idx < NV; idx += kThreads) {
    const int row = idx / (W / 4), k = 4 * (idx % (W / 4));
    *reinterpret_cast<f32x4*>(dst + row * ld + k) =
        *reinterpret_cast<const f32x4*>(src + (size_t)row * gld + k);
  }
}

// Stage the 256-wide code part of the network input for a row tile: LDS row i holds
// code[min((row0+i)/rep, nrows_code-1)] (rows >= m are zero-filled).
template <int RT>
__device__ __forceinline__ void load_code_tile(float* dst, int ld,
                                               const float* __restrict__ code, int64_t ld_code,
                                               int rep, int64_t row0, int64_t m, bool vec4) {
  constexpr int R = 16 * RT;
  constexpr int NV = R * 64;
  for (int idx = threadIdx.x; idx < NV; idx += kThreads) {
    const int i = idx >> 6, k = 4 * (idx & 63);
    const int64_t row = row0 + i;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (row < m) {
      const float* p = code + (row / rep) * ld_code + k;
      if (vec4) {
        v = *reinterpret_cast<const f32x4*>(p);
      } else {
        v[0] = p[0]; v[1] = p[1]; v[2] = p[2]; v[3] = p[3];
      }
    }
    *reinterpret_cast<f32x4*>(dst + i * ld + k) = v;
  }
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
  return v;
}

// Sum over the threads of a workgroup (blockDim.x = 64..256); result valid in thread 0.
// `red` = 4 LDS floats.
__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  float r = 0.f;
  if (threadIdx.x == 0) {
    const int nw = blockDim.x >> 6;
    for (int w = 0; w < nw; ++w) r += red[w];
  }
  __syncthreads();
  return r;
}

}  // namespace ndp
