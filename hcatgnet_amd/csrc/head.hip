// Regression head, forward + loss + backward in ONE launch (SURVEY rows a10, a12 and their part of a11; f2):
//     z    = LeakyReLU(emb W0^T + b0)            [B, 2D] -> [B, D]        reference model/gcn.py:36-45, 70-71
//     out  = z W1^T + b1                         [B, D]  -> [B, C]
//     mse  = mean((out - y)^2) ; loss = sqrt(mse) reference utils/utils_model.py:64 (`torch.sqrt(model.loss(out, y))`)
//     dout = dloss/dout ; dz, demb, dW0, db0, dW1, db1   (`loss.backward()`, utils/utils_model.py:65, upstream grad 1)
// As separate launches (readout fwd, mse fwd, sqrt, three torch kernels of sqrt's backward, mse bwd, readout bwd)
// this dependent chain of eight tiny kernels cost ~39 us of a 132 us training step: pure launch latency around
// 67 MFLOP.  Here one kernel walks the chain; the only global dependence (every graph's gradient needs the batch
// loss) is a grid barrier in the middle: per-workgroup partial sums of squared errors -> sense-reversing barrier
// on two device words -> every workgroup adds the partials in the same fixed order (bitwise reproducible loss).
// One workgroup (4 waves) per 32-graph tile; the tile's emb / z / (out - y) stay in LDS across the barrier.
// The grid never exceeds the CU count, so all workgroups are co-resident (the barrier cannot starve).
// Contractions on v_mfma_f32_32x32x2_f32 (exact f32): the head is latency-bound, not MFMA-bound.
#include "common.h"

#ifdef HCG_HEAD_STAMP
__device__ unsigned long long g_head_stamp[16 * 16];
#define HSTAMP(i) do { if (threadIdx.x == 0 && blockIdx.x < 16) { unsigned long long _t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(_t)::"memory"); g_head_stamp[blockIdx.x * 16 + (i)] = _t; } } while (0)
#else
#define HSTAMP(i) do { } while (0)
#endif

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
// see fused.hip (mfma_results_fence): keep VALU reads of an accumulator a whole foreign MFMA away from the chain's
// last MFMA when several waves share the SIMD's matrix pipe (f32 32x32x2: 16 passes = 64 cycles)
__device__ __forceinline__ void mfma_results_fence(f32x16& a) { asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" : "+v"(a)); }

constexpr int RT = 32;            // graphs per tile
constexpr int RCMAX = 8;
constexpr int MAXGRID = 256;
static_assert(2 + 2 * MAXGRID <= HCG_HEAD_SYNC_WORDS, "sync words");

// Shapes of the head for hidden width RD (= embedding_dim: 64, the reference's default, options/base_options.py:199-204;
// 128 = BASELINE configs[4]).  Wave roles: forward = RD/32 output column blocks x 2 K halves -> HW = RD/16 waves (4 / 8);
// backward = one 32-column block of the 2RD-wide embedding per wave (2RD/32 = HW blocks).
template <int RD_>
struct HC {
  static constexpr int RD = RD_;          // hidden width
  static constexpr int RK = 2 * RD;       // pooled embedding width
  static constexpr int NB = RD / 32;      // output column blocks of the forward GEMM
  static constexpr int HW = 2 * NB;       // waves per workgroup
  static constexpr int NT = HW * 64;
  static constexpr int ES = RK + 4;       // LDS stride of the emb tile
  static constexpr int ZS = RD + 4;       // LDS stride of the z / dz tile
  static constexpr int WS0 = RK + 1;      // LDS stride of the W0 image [RD][RK]
  static constexpr bool W0_LDS = RD <= 64;   // RD = 128: the image would be 131 KB -- the W0 fragments come from global memory / L2
  static constexpr int OJ = RD / 8;       // out projection: threads per graph row (8 hidden units each)
  static constexpr int QN = RD / 4;       // backward step 1: float4 column groups of a dz row
  // slab layout per WORKGROUP (same as readout.hip): dW0 [RD][RK] | db0 [RD] | dW1 [C][RD] | db1 [C]  (C padded to RCMAX)
  static constexpr int SMALL = RD + RCMAX * RD + RCMAX;       // db0 | dW1 | db1
  static constexpr int SLAB = RD * RK + SMALL;
  static constexpr int ESZ = RT * ES > HW * (SMALL + 8) ? RT * ES : HW * (SMALL + 8);    // emb tile, later the combine scratch
};

__device__ __forceinline__ constexpr int krow(int i, int h) { return (i & 3) + 8 * (i >> 2) + 4 * h; }

template <int RD>
struct HeadLds {
  using C_ = HC<RD>;
  float w0[C_::W0_LDS ? RD * C_::WS0 : 4];   // W0 [d][k]
  float e[C_::ESZ];              // emb tile
  float z[RT * C_::ZS];          // z tile, later dz
  float part[C_::NB][RT * 33];   // K-half partial sums of the forward GEMM (per column block)
  float diff[RT][RCMAX];         // out - y, later dout
  float w1[RCMAX * RD];
  float red[C_::HW * 64];
  float bcast[4];
};

// Grid-wide exchange of the per-workgroup squared-error partials WITHOUT read-modify-write atomics (128 workgroups
// taking turns on one counter word cost ~10 us on this 8-XCD part: device-scope atomics are resolved at the memory
// side).  `sync` = HCG_HEAD_SYNC_WORDS int32 words, all zero before the first launch ever:
//   sync[0]              generation = number of launches completed so far (every workgroup reads it on entry)
//   sync[2 + 2b .. +1]   slot of workgroup b: {partial sum bits, stamp}, written as ONE 8-byte store, stamp = generation + 1
// A workgroup publishes its slot, then wave 0 polls all slots until every stamp is current -- nobody can get past that
// before every workgroup has read the generation, so workgroup 0 may advance it right afterwards.  Every workgroup
// adds the same partials in the same lane order: the loss is bitwise reproducible.
// Split in two so that work which does not need the sum can sit between them: publish_partial() right after the
// forward, collect_partials() only where the sum is first needed.
__device__ __forceinline__ void publish_partial(int* sync, int gen, float my_partial) {
  unsigned long long* slots = reinterpret_cast<unsigned long long*>(sync + 2);
  if (threadIdx.x == 0) {
    const unsigned long long v = ((unsigned long long)(unsigned)(gen + 1) << 32) | (unsigned long long)__float_as_uint(my_partial);
    __hip_atomic_store(&slots[blockIdx.x], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// The poll is BOUNDED (MI355X_MICROARCH.md, Correctness boundaries: "bound every spin"): a workgroup that has waited
// HEAD_SPIN_TICKS of the constant 100 MHz clock (s_memrealtime; 2 s -- an exchange takes microseconds) gives up, ORs
// HCG_HEAD_ERR_TIMEOUT into sync[1] and returns NaN, so the launch ends and the loss it leaves is NaN instead of the
// GPU hanging (a grid that is not co-resident, or two launches interleaving on ONE set of sync words from two streams).
constexpr unsigned long long HEAD_SPIN_TICKS = 200000000ull;
__device__ __forceinline__ float collect_partials(int* sync, int nblk, int gen, float* bcast) {
  unsigned long long* slots = reinterpret_cast<unsigned long long*>(sync + 2);
  const int lane = threadIdx.x & 63;
  if (threadIdx.x < 64) {
    float s = 0.f;
    bool done;
    const unsigned long long t_start = __builtin_amdgcn_s_memrealtime();
    do {
      done = true;
      s = 0.f;
#pragma unroll
      for (int k = 0; k < MAXGRID / 64; ++k) {
        const int b = lane + 64 * k;
        const unsigned long long v = __hip_atomic_load(&slots[b < nblk ? b : nblk - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (b < nblk) {
          done = done && (int)(v >> 32) == gen + 1;
          s += __uint_as_float((unsigned)v);
        }
      }
      done = __all(done);
      if (!done) {
        __builtin_amdgcn_s_sleep(2);
        if (__builtin_amdgcn_s_memrealtime() - t_start > HEAD_SPIN_TICKS) {   // wave-uniform (scalar clock)
          if (lane == 0) atomicOr(&sync[1], HCG_HEAD_ERR_TIMEOUT);
          s = __builtin_nanf("");
          break;
        }
      }
    } while (!done);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    if (lane == 0) {
      bcast[1] = s;
      if (blockIdx.x == 0) __hip_atomic_store(&sync[0], gen + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  __syncthreads();
  return bcast[1];
}

// stage rows [g0, g0 + n) of a [B, W] matrix into dst[row * ld + c], rows >= n zero (all 256 threads, float4)
template <int W, int NT>
__device__ __forceinline__ void stage_rows(float* dst, int ld, const float* __restrict__ src, int g0, int n, int B) {
  constexpr int PER_ROW = W / 4, ITER = RT * PER_ROW / NT;
  static_assert(RT * PER_ROW % NT == 0, "rows per thread");
  float4 v[ITER];
#pragma unroll
  for (int it = 0; it < ITER; ++it) {
    const int idx = threadIdx.x + it * NT, row = idx / PER_ROW, c4 = idx - row * PER_ROW;
    int g = g0 + row;
    if (g > B - 1) g = B - 1;
    v[it] = *reinterpret_cast<const float4*>(src + (size_t)g * W + 4 * c4);
  }
#pragma unroll
  for (int it = 0; it < ITER; ++it) {
    const int idx = threadIdx.x + it * NT, row = idx / PER_ROW, c4 = idx - row * PER_ROW;
    if (row >= n) v[it] = make_float4(0.f, 0.f, 0.f, 0.f);
    *reinterpret_cast<float4*>(dst + row * ld + 4 * c4) = v[it];
  }
}

template <int RD>
__global__ __launch_bounds__(HC<RD>::NT, 1) void k_head(const float* __restrict__ emb, const float* __restrict__ y,
                                                       const float* __restrict__ W0, const float* __restrict__ b0,
                                                       const float* __restrict__ W1, const float* __restrict__ b1, int B, int C,
                                                       float slope, int rmse, float* __restrict__ z, float* __restrict__ out,
                                                       float* __restrict__ loss, float* __restrict__ demb,
                                                       float* __restrict__ slabs, int* __restrict__ sync,
                                                       int* __restrict__ step_counter, float* __restrict__ sse_tail) {
  using K = HC<RD>;
  constexpr int RK = K::RK, NB = K::NB, HW = K::HW, NT = K::NT, ES = K::ES, ZS = K::ZS, WS0 = K::WS0, OJ = K::OJ, QN = K::QN;
  constexpr int SMALL = K::SMALL, SLAB = K::SLAB;
  constexpr bool W0_LDS = K::W0_LDS;
  __shared__ HeadLds<RD> L;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int tiles = (B + RT - 1) / RT;
  const int nblk = gridDim.x;
  const int gen = __hip_atomic_load(&sync[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // launches completed so far
  HSTAMP(0);

  // weights -> LDS once.  Every global load of the prologue is issued before the first LDS write (a load-store
  // loop would serialise 32 HBM round trips per thread: that alone cost ~15 us of this launch-latency-bound kernel)
  {
    constexpr int W4 = W0_LDS ? RD * RK / 4 / NT : 1;          // 8 float4 of W0 per thread
    constexpr int W1N = RCMAX * RD / NT;
    float4 wv[W4];
    if (W0_LDS) {
#pragma unroll
      for (int it = 0; it < W4; ++it) wv[it] = *reinterpret_cast<const float4*>(W0 + 4 * (threadIdx.x + it * NT));
    }
    float w1v[W1N];
#pragma unroll
    for (int it = 0; it < W1N; ++it) {
      const int idx = threadIdx.x + it * NT;
      w1v[it] = W1[idx < C * RD ? idx : 0];
    }
    if (W0_LDS) {
#pragma unroll
      for (int it = 0; it < W4; ++it) {
        const int f4 = threadIdx.x + it * NT, row = f4 / (RK / 4), c4 = f4 - row * (RK / 4);
        float* dst = L.w0 + row * WS0 + 4 * c4;
        dst[0] = wv[it].x; dst[1] = wv[it].y; dst[2] = wv[it].z; dst[3] = wv[it].w;
      }
    }
#pragma unroll
    for (int it = 0; it < W1N; ++it) {
      const int idx = threadIdx.x + it * NT;
      L.w1[idx] = idx < C * RD ? w1v[it] : 0.f;
    }
  }
  const int nb = wave % NB, kh = wave / NB;             // forward: output column block / K half of this wave
  const float bz = b0[nb * 32 + r];
  const int orow = threadIdx.x / OJ, oj = threadIdx.x % OJ;   // out projection: OJ threads per graph row, 8 hidden units each
  float b1v[RCMAX];
#pragma unroll
  for (int c = 0; c < RCMAX; ++c) b1v[c] = b1[c < C ? c : 0];

  // ---------------------------------------------------------------- phase 1: forward + squared error
  float sse = 0.f;                                      // thread-private partial, fixed tile order
  int staged = -1;
  for (int t = blockIdx.x; t < tiles; t += nblk) {
    const int g0 = t * RT, n = B - g0 < RT ? B - g0 : RT;
    // the targets of this tile are requested here, a whole GEMM ahead of their use (unconditional, clamped)
    float yv[RCMAX];
#pragma unroll
    for (int c = 0; c < RCMAX; ++c)
      yv[c] = y[(size_t)(g0 + orow < B ? g0 + orow : B - 1) * C + (c < C ? c : C - 1)];
    __syncthreads();                                    // previous tile's readers are done (also orders the weight staging)
    stage_rows<RK, NT>(L.e, ES, emb, g0, n, B);
    __syncthreads();
    HSTAMP(1);
    staged = t;
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    if (W0_LDS) {
#pragma unroll
      for (int t8 = 0; t8 < RK / 16; ++t8) {              // this wave's K half: k = RD kh + 8 t8 + 4h + u
        const int k0 = RD * kh + 8 * t8 + 4 * h;
        const float4 a = *reinterpret_cast<const float4*>(L.e + r * ES + k0);
        const float* wrow = L.w0 + (nb * 32 + r) * WS0 + k0;
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, wrow[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, wrow[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, wrow[2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, wrow[3], acc, 0, 0, 0);
      }
    } else {
      // W0 fragments straight from global memory / L2 (every workgroup reads the same 131 KB): eight float4 in flight
      constexpr int WB = 8;
      const float* wg = W0 + (size_t)(nb * 32 + r) * RK + RD * kh + 4 * h;
#pragma unroll
      for (int t0 = 0; t0 < RK / 16; t0 += WB) {
        float4 wv[WB];
#pragma unroll
        for (int u = 0; u < WB; ++u) wv[u] = *reinterpret_cast<const float4*>(wg + 8 * (t0 + u));
#pragma unroll
        for (int u = 0; u < WB; ++u) {
          const float4 a = *reinterpret_cast<const float4*>(L.e + r * ES + RD * kh + 8 * (t0 + u) + 4 * h);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, wv[u].x, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, wv[u].y, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, wv[u].z, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, wv[u].w, acc, 0, 0, 0);
        }
      }
    }
    mfma_results_fence(acc);
    HSTAMP(2);
    if (kh == 1) {
#pragma unroll
      for (int i = 0; i < 16; ++i) L.part[nb][krow(i, h) * 33 + r] = acc[i];
    }
    __syncthreads();
    if (kh == 0) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int row = krow(i, h);
        const float v = hcg_leaky((acc[i] + L.part[nb][row * 33 + r]) + bz, slope);
        L.z[row * ZS + nb * 32 + r] = v;
        if (row < n) z[(size_t)(g0 + row) * RD + nb * 32 + r] = v;
      }
    }
    __syncthreads();
    {   // out[row][c] = z[row] . W1[c] + b1[c] ; diff = out - y.  Thread (row, j): hidden units 8j..8j+7 of every class,
        // the OJ partial sums of a row meet by xor-shuffles (fixed order); the target is loaded ahead of the arithmetic
      const float4 za = *reinterpret_cast<const float4*>(L.z + orow * ZS + 8 * oj);
      const float4 zb = *reinterpret_cast<const float4*>(L.z + orow * ZS + 8 * oj + 4);
#pragma unroll
      for (int c = 0; c < RCMAX; ++c) {
        if (c < C) {                                     // block-uniform
          const float4 wa = *reinterpret_cast<const float4*>(L.w1 + c * RD + 8 * oj);
          const float4 wb = *reinterpret_cast<const float4*>(L.w1 + c * RD + 8 * oj + 4);
          float s = ((za.x * wa.x + za.y * wa.y) + (za.z * wa.z + za.w * wa.w)) + ((zb.x * wb.x + zb.y * wb.y) + (zb.z * wb.z + zb.w * wb.w));
#pragma unroll
          for (int off = 1; off < OJ; off <<= 1) s += __shfl_xor(s, off, 64);
          if (oj == 0) {
            float d = 0.f;
            if (orow < n) {
              s += b1v[c];
              out[(size_t)(g0 + orow) * C + c] = s;
              d = s - yv[c];
              sse += d * d;
            }
            L.diff[orow][c] = d;
          }
        } else if (oj == 0) {
          L.diff[orow][c] = 0.f;
        }
      }
    }
  }
  // block partial of the squared error: lanes -> wave (fixed xor tree) -> block (fixed order)
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) sse += __shfl_xor(sse, off, 64);
  if (lane == 0) L.red[wave] = sse;
  __syncthreads();
  float block_sse = L.red[0];
#pragma unroll
  for (int w = 1; w < HW; ++w) block_sse += L.red[w];
  HSTAMP(3);

  // ---------------------------------------------------------------- grid-wide: every workgroup needs the batch's squared error
  // -- but only as ONE scalar factor of a backward that is linear in it (dloss/dout = gscale * diff).  The partial is
  // published here, the backward below runs on the unscaled diff while the other workgroups' partials arrive, and the
  // sum is collected where the first result leaves the chip (demb); the weight-gradient sums are scaled once at the end.
  publish_partial(sync, gen, block_sse);
  float gscale = 0.f;
  bool have_scale = false;
  auto collect = [&]() {                               // block-uniform; every workgroup runs it exactly once
    const float total_sse = collect_partials(sync, nblk, gen, L.bcast);
    HSTAMP(4);
    if (threadIdx.x == 0) {
      const float mse = total_sse / ((float)B * (float)C);
      const float lv = rmse ? sqrtf(mse) : mse;
      // dloss/dout = scale * diff.  HCG_HEAD_SSE: scale 1 -- the gradients leave as those of SSE / 2 and the batch's SSE and
      // element count go to `sse_tail`: ranks of a data-parallel job sum both and scale once, which reproduces the gradient
      // of sqrt(MSE) over the CONCATENATED batch exactly (hcg_sse_finalize / hcg_adam_step_dev_sse)
      L.bcast[0] = rmse == HCG_HEAD_SSE ? 1.0f : rmse ? 1.0f / ((float)B * (float)C * lv) : 2.0f / ((float)B * (float)C);
      if (blockIdx.x == 0) {
        loss[0] = lv;
        loss[1] = mse;
        if (sse_tail) { sse_tail[0] = total_sse; sse_tail[1] = (float)B * (float)C; }
        if (step_counter) step_counter[0] += 1;        // this training step's number, for the update launched later
      }
    }
    __syncthreads();
    gscale = L.bcast[0];
    have_scale = true;
  };

  // ---------------------------------------------------------------- phase 2: backward (unscaled), loss, scale
  const int q = threadIdx.x % QN, rgrp = threadIdx.x / QN;     // step 1: float4 column group / row (16 rows per pass)
  const int cb = wave;                                 // backward: this wave's 32-column block of the 2RD-wide embedding
  f32x16 dw0[NB];
#pragma unroll
  for (int mb = 0; mb < NB; ++mb)
#pragma unroll
    for (int i = 0; i < 16; ++i) dw0[mb][i] = 0.f;
  float4 db0 = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 dw1[RCMAX];
  float db1[RCMAX];
#pragma unroll
  for (int c = 0; c < RCMAX; ++c) { dw1[c] = make_float4(0.f, 0.f, 0.f, 0.f); db1[c] = 0.f; }

  for (int t = blockIdx.x; t < tiles; t += nblk) {
    const int g0 = t * RT, n = B - g0 < RT ? B - g0 : RT;
    if (t != staged) {                                 // more tiles than workgroups: bring the tile back (block-uniform)
      __syncthreads();
      stage_rows<RK, NT>(L.e, ES, emb, g0, n, B);
      stage_rows<RD, NT>(L.z, ZS, z, g0, n, B);
      if (threadIdx.x < RT * RCMAX) {
        const int row = threadIdx.x >> 3, c = threadIdx.x & 7;
        float d = 0.f;
        if (c < C && row < n) d = out[(size_t)(g0 + row) * C + c] - y[(size_t)(g0 + row) * C + c];
        L.diff[row][c] = d;
      }
      __syncthreads();
      staged = t;
    }
    // 1. dz = (dout W1) * leaky'(z) -> L.z ; db0, dW1, db1 partial sums.  Two (row, 4-column) slots per thread.
    float4 dzv[2];
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int row = it * 16 + rgrp;
      float4 d = make_float4(0.f, 0.f, 0.f, 0.f);
      if (row < n) {
        const float4 zz = *reinterpret_cast<const float4*>(L.z + row * ZS + 4 * q);
#pragma unroll
        for (int c = 0; c < RCMAX; ++c) {
          if (c < C) {
            const float go = L.diff[row][c];                  // (unscaled: see above)
            const float4 w = *reinterpret_cast<const float4*>(L.w1 + c * RD + 4 * q);
            d.x += go * w.x; d.y += go * w.y; d.z += go * w.z; d.w += go * w.w;
            dw1[c].x += go * zz.x; dw1[c].y += go * zz.y; dw1[c].z += go * zz.z; dw1[c].w += go * zz.w;
            if (q == 0) db1[c] += go;
          }
        }
        d.x *= hcg_leaky_grad(zz.x, slope); d.y *= hcg_leaky_grad(zz.y, slope);
        d.z *= hcg_leaky_grad(zz.z, slope); d.w *= hcg_leaky_grad(zz.w, slope);
        db0.x += d.x; db0.y += d.y; db0.z += d.z; db0.w += d.w;
      }
      dzv[it] = d;
    }
    __syncthreads();                                   // every read of z is done
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int row = it * 16 + rgrp;
      *reinterpret_cast<float4*>(L.z + row * ZS + 4 * q) = dzv[it];
    }
    __syncthreads();
    HSTAMP(5);
    // 2. dW0[:, cb] += dz^T emb[:, cb]   (K = graph rows)
#pragma unroll
    for (int s = 0; s < RT / 2; ++s) {
      const int row = 2 * s + h;
      const float bv = L.e[row * ES + cb * 32 + r];
#pragma unroll
      for (int mb = 0; mb < NB; ++mb)
        dw0[mb] = __builtin_amdgcn_mfma_f32_32x32x2f32(L.z[row * ZS + mb * 32 + r], bv, dw0[mb], 0, 0, 0);
    }
    // 3. demb[:, cb] = dz W0[:, cb]
    f32x16 de;
#pragma unroll
    for (int i = 0; i < 16; ++i) de[i] = 0.f;
    if (W0_LDS) {
#pragma unroll
      for (int t8 = 0; t8 < RD / 8; ++t8) {
        const float4 a = *reinterpret_cast<const float4*>(L.z + r * ZS + 8 * t8 + 4 * h);
        const int d0 = 8 * t8 + 4 * h;
        de = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, L.w0[(d0 + 0) * WS0 + cb * 32 + r], de, 0, 0, 0);
        de = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, L.w0[(d0 + 1) * WS0 + cb * 32 + r], de, 0, 0, 0);
        de = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, L.w0[(d0 + 2) * WS0 + cb * 32 + r], de, 0, 0, 0);
        de = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, L.w0[(d0 + 3) * WS0 + cb * 32 + r], de, 0, 0, 0);
      }
    } else {
      constexpr int WB = 4;                            // 16 dword loads of W0 in flight (coalesced across r)
#pragma unroll
      for (int t0 = 0; t0 < RD / 8; t0 += WB) {
        float wv[WB][4];
#pragma unroll
        for (int u = 0; u < WB; ++u)
#pragma unroll
          for (int j = 0; j < 4; ++j) wv[u][j] = W0[(size_t)(8 * (t0 + u) + 4 * h + j) * RK + cb * 32 + r];
#pragma unroll
        for (int u = 0; u < WB; ++u) {
          const float4 a = *reinterpret_cast<const float4*>(L.z + r * ZS + 8 * (t0 + u) + 4 * h);
          de = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, wv[u][0], de, 0, 0, 0);
          de = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, wv[u][1], de, 0, 0, 0);
          de = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, wv[u][2], de, 0, 0, 0);
          de = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, wv[u][3], de, 0, 0, 0);
        }
      }
    }
    mfma_results_fence(de);
    HSTAMP(6);
    if (!have_scale) collect();
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int row = krow(i, h);
      if (row < n) demb[(size_t)(g0 + row) * RK + cb * 32 + r] = gscale * de[i];
    }
  }
  if (!have_scale) collect();                          // (a workgroup without a tile still takes part in the exchange)

  // ---------------------------------------------------------------- one slab per workgroup
  float* slab = slabs + (size_t)blockIdx.x * SLAB;
#pragma unroll
  for (int mb = 0; mb < NB; ++mb) mfma_results_fence(dw0[mb]);
#pragma unroll
  for (int mb = 0; mb < NB; ++mb)
#pragma unroll
    for (int i = 0; i < 16; ++i) slab[(mb * 32 + krow(i, h)) * RK + cb * 32 + r] = gscale * dw0[mb][i];
  // db0 / dW1 / db1: thread (row group, q) holds partial sums for columns 4q..4q+3; combine the row groups of a wave
  // (64 / QN of them) by shuffles, the waves through LDS -- all in a fixed order
  auto fold = [](float4 v) {
#pragma unroll
    for (int off = QN; off < 64; off <<= 1) {
      v.x += __shfl_xor(v.x, off, 64); v.y += __shfl_xor(v.y, off, 64); v.z += __shfl_xor(v.z, off, 64); v.w += __shfl_xor(v.w, off, 64);
    }
    return v;
  };
  __syncthreads();
  float* scratch = L.e;                                // [HW][SMALL + 8]: the emb tile is dead
  {
    float* mine = scratch + wave * (SMALL + 8);
    db0 = fold(db0);
    if (lane < QN) *reinterpret_cast<float4*>(mine + 4 * q) = db0;
#pragma unroll
    for (int c = 0; c < RCMAX; ++c) {
      if (c < C) {                                     // block-uniform: classes the model does not have cost no shuffles
        const float4 v = fold(dw1[c]);
        if (lane < QN) *reinterpret_cast<float4*>(mine + RD + c * RD + 4 * q) = v;
        float sc = db1[c];                             // lanes with q == 0 hold the partial sums
#pragma unroll
        for (int off = QN; off < 64; off <<= 1) sc += __shfl_xor(sc, off, 64);
        if (lane == 0) mine[RD + RCMAX * RD + c] = sc;
      } else {
        if (lane < QN) *reinterpret_cast<float4*>(mine + RD + c * RD + 4 * q) = make_float4(0.f, 0.f, 0.f, 0.f);
        if (lane == 0) mine[RD + RCMAX * RD + c] = 0.f;
      }
    }
  }
  __syncthreads();
  HSTAMP(7);
  for (int idx = threadIdx.x; idx < SMALL; idx += NT) {
    float sm = scratch[idx];
#pragma unroll
    for (int w = 1; w < HW; ++w) sm += scratch[w * (SMALL + 8) + idx];
    slab[RD * RK + idx] = gscale * sm;
  }
}

// Workgroups that are certainly co-resident: the grid-wide exchange needs every workgroup of the launch on a CU at the
// same time.  One per CU at most (70-80 KB of LDS each), and only if the occupancy query admits one at all; the guide's
// SGPR cap min(API, 8, 800 / (ceil(sgpr / 16) * 16 + 16)) is >= 7 for any kernel, far above the 1 used here.
template <int RD>
int head_resident_cap_t() {
  static int cap = -1;      // queried once per process (also keeps the query out of a stream capture)
  if (cap >= 0) return cap;
  int dev = 0, cus = MAXGRID, per_cu = 0;
  if (hipGetDevice(&dev) == hipSuccess) {
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
  }
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_head<RD>, HC<RD>::NT, 0) != hipSuccess) per_cu = 0;
  if (cus > MAXGRID) cus = MAXGRID;
  cap = per_cu >= 1 ? cus : 0;
  return cap;
}
int head_resident_cap(int64_t D) { return D == 128 ? head_resident_cap_t<128>() : head_resident_cap_t<64>(); }

int head_grid(int64_t B, int64_t D) {
  int cus = head_resident_cap(D);
  if (cus < 1) cus = 1;     // (hcg_head_fwd_bwd refuses to launch when the cap is 0)
  int grid = (int)((B + RT - 1) / RT);
  if (grid > cus) grid = cus;
  return grid < 1 ? 1 : grid;
}
size_t head_slab(int64_t D) { return D == 128 ? (size_t)HC<128>::SLAB : (size_t)HC<64>::SLAB; }

}  // namespace

extern "C" int hcg_head_supported(int64_t D, int64_t C) { return ((D == 64 || D == 128) && C >= 1 && C <= RCMAX) ? 1 : 0; }

// workspace: [grid][SLAB] gradient slabs
extern "C" size_t hcg_head_workspace_bytes_d(int64_t B, int64_t D) {
  if (D != 64 && D != 128) return 0;
  return hcg_align_up((size_t)head_grid(B, D) * head_slab(D) * sizeof(float), 256) + 256;
}
extern "C" size_t hcg_head_workspace_bytes(int64_t B) { return hcg_head_workspace_bytes_d(B, 64); }

extern "C" int hcg_head_fwd_bwd(const float* emb, const float* y, const float* W0, const float* b0, const float* W1,
                                const float* b1, int64_t B, int64_t D, int64_t C, float slope, int rmse, float* z,
                                float* out, float* loss, float* demb, void* workspace, size_t workspace_bytes,
                                int32_t* sync, int32_t* step_counter, hcg_stream_t stream) {
  if (rmse != 0 && rmse != 1) return HCG_ERR_INVALID_ARG;
  return hcg_head_fwd_bwd_ex(emb, y, W0, b0, W1, b1, B, D, C, slope, rmse, z, out, loss, demb, workspace, workspace_bytes, sync,
                             step_counter, nullptr, stream);
}

extern "C" int hcg_head_fwd_bwd_ex(const float* emb, const float* y, const float* W0, const float* b0, const float* W1,
                                   const float* b1, int64_t B, int64_t D, int64_t C, float slope, int rmse, float* z,
                                   float* out, float* loss, float* demb, void* workspace, size_t workspace_bytes,
                                   int32_t* sync, int32_t* step_counter, float* sse_tail, hcg_stream_t stream) {
  if (!hcg_head_supported(D, C)) return HCG_ERR_UNSUPPORTED;
  if (rmse < 0 || rmse > HCG_HEAD_SSE || (rmse == HCG_HEAD_SSE && !sse_tail)) return HCG_ERR_INVALID_ARG;
  if (head_resident_cap(D) < 1) return HCG_ERR_UNSUPPORTED;   // not even one workgroup per CU: the exchange cannot run
  if (B <= 0 || !emb || !y || !W0 || !b0 || !W1 || !b1 || !z || !out || !loss || !demb || !workspace || !sync)
    return HCG_ERR_INVALID_ARG;
  if (workspace_bytes < hcg_head_workspace_bytes_d(B, D)) return HCG_ERR_WORKSPACE;
  const int grid = head_grid(B, D);
  float* slabs = (float*)workspace;
  if (D == 128)
    hipLaunchKernelGGL(k_head<128>, dim3(grid), dim3(HC<128>::NT), 0, (hipStream_t)stream, emb, y, W0, b0, W1, b1, (int)B, (int)C,
                       slope, rmse, z, out, loss, demb, slabs, (int*)sync, (int*)step_counter, sse_tail);
  else
    hipLaunchKernelGGL(k_head<64>, dim3(grid), dim3(HC<64>::NT), 0, (hipStream_t)stream, emb, y, W0, b0, W1, b1, (int)B, (int)C,
                       slope, rmse, z, out, loss, demb, slabs, (int*)sync, (int*)step_counter, sse_tail);
  HCG_CHECK_LAUNCH();
  return HCG_OK;
}

extern "C" int hcg_head_reduce_job_d(const void* workspace, size_t workspace_bytes, int64_t B, int64_t D, int64_t C, float* dW0,
                                     float* db0, float* dW1, float* db1, hcg_reduce_job* job) {
  if (B <= 0 || (D != 64 && D != 128) || C < 1 || C > RCMAX || !dW0 || !db0 || !dW1 || !db1 || !job || !workspace)
    return HCG_ERR_INVALID_ARG;
  if (workspace_bytes < hcg_head_workspace_bytes_d(B, D)) return HCG_ERR_WORKSPACE;
  const int32_t RD = (int32_t)D, RK = 2 * RD;
  job->slabs = (const float*)workspace;
  job->nslabs = head_grid(B, D);
  job->slab_floats = (int32_t)head_slab(D);
  job->nseg = 4;
  job->reserved = 0;
  job->seg[0] = hcg_reduce_seg{0, RD * RK, RK, RK, dW0};
  job->seg[1] = hcg_reduce_seg{RD * RK, RD, 1, 1, db0};
  job->seg[2] = hcg_reduce_seg{RD * RK + RD, (int32_t)C * RD, RD, RD, dW1};
  job->seg[3] = hcg_reduce_seg{RD * RK + RD + RCMAX * RD, (int32_t)C, 1, 1, db1};
  return HCG_OK;
}
extern "C" int hcg_head_reduce_job(const void* workspace, size_t workspace_bytes, int64_t B, int64_t C, float* dW0,
                                   float* db0, float* dW1, float* db1, hcg_reduce_job* job) {
  return hcg_head_reduce_job_d(workspace, workspace_bytes, B, 64, C, dW0, db0, dW1, db1, job);
}
