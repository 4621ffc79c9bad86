// The regression head on a tile of <= 32 graphs (SURVEY rows a10, a12 and their part of a11; f2), shared by the stand-alone
// head kernel (head.hip) and by the tail phase of the small-graph forward (fused.hip: the head rides in the conv stack's
// launch):
//     z    = LeakyReLU(emb W0^T + b0)            [B, 2D] -> [B, D]        reference model/gcn.py:36-45, 70-71
//     out  = z W1^T + b1                         [B, D]  -> [B, C]
//     diff = out - y ; sse += diff^2             reference utils/utils_model.py:64 (`torch.sqrt(model.loss(out, y))`)
//     backward of SSE / 2 (dloss/dout = diff, UNSCALED): dz, demb, dW0, db0, dW1, db1     (utils/utils_model.py:65)
// The batch-dependent factor of the real loss gradient -- 1 / (B C sqrt(MSE)) -- is ONE scalar and the backward is linear
// in it, so nothing here waits for the other workgroups: each leaves its partial sum of squared errors in its slab and the
// step's last launch (reduce.hip: k_step_tail) applies the scale while it reduces.  Round 2 exchanged that scalar across
// the grid inside the head kernel (8-byte stamped slots, bounded polls, co-residency): all of it is gone.
// Contractions on v_mfma_f32_32x32x2_f32 (exact f32): the head is latency-bound, not MFMA-bound.
#pragma once
#include "common.h"

namespace hcg_head {

typedef float f32x16 __attribute__((ext_vector_type(16)));
// see split_mfma.h (mfma_results_fence): keep VALU reads of an accumulator a whole foreign MFMA away from the chain's
// last MFMA when several waves share the SIMD's matrix pipe (f32 32x32x2: 16 passes = 64 cycles)
__device__ __forceinline__ void results_fence(f32x16& a) { asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" : "+v"(a)); }

constexpr int RT = 32;            // graphs per tile
constexpr int RCMAX = 8;

// Shapes of the head for hidden width RD (= embedding_dim: 64, the reference's default, options/base_options.py:199-204;
// 128 = BASELINE configs[4]).  Wave roles: forward = RD/32 output column blocks x 2 K halves -> HW = RD/16 waves (4 / 8);
// backward = one 32-column block of the 2RD-wide embedding per wave (2RD/32 = HW blocks).
template <int RD_>
struct HC {
  static constexpr int RD = RD_;          // hidden width
  static constexpr int RK = 2 * RD;       // pooled embedding width
  static constexpr int NB = RD / 32;      // output column blocks of the forward GEMM
  static constexpr int HW = 2 * NB;       // working waves
  static constexpr int NT = HW * 64;      // working threads
  static constexpr int ES = RK + 4;       // LDS stride of the emb tile
  static constexpr int ZS = RD + 4;       // LDS stride of the z / dz tile
  static constexpr int WS0 = RK + 1;      // LDS stride of the W0 image [RD][RK]
  static constexpr bool W0_LDS = RD <= 64;   // RD = 128: the image would be 131 KB -- the W0 fragments come from global memory / L2
  static constexpr int OJ = RD / 8;       // out projection: threads per graph row (8 hidden units each)
  static constexpr int QN = RD / 4;       // backward step 1: float4 column groups of a dz row
  // slab layout per WORKGROUP: dW0 [RD][RK] | db0 [RD] | dW1 [RCMAX][RD] | db1 [RCMAX]; the workgroups' partial sums of
  // squared errors sit behind the last slab, contiguous ([nslabs] floats: every block of the step's last launch adds all of
  // them -- strided inside the slabs that was one 64-byte sector per partial and block)
  static constexpr int SMALL = RD + RCMAX * RD + RCMAX;       // db0 | dW1 | db1
  static constexpr int SLAB = RD * RK + SMALL;                // (a multiple of 8 floats: slab rows stay 32-byte aligned)
  static_assert(SLAB % 8 == 0, "slab alignment");
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
  float diff[RT][RCMAX];         // out - y
  float w1[RCMAX * RD];
  float red[16];
};

// per-thread state that lives across the tiles of a workgroup
// RC = compile-time bound of the class count handled in registers: 1 (the reference's regression, n_classes = 1,
// options/base_options.py:178-183) or RCMAX -- with eight classes' accumulators live across the tile loop the 128-wide
// head spilled 82 registers
template <int RD, int RC>
struct HeadState {
  using K = HC<RD>;
  float bz;                      // b0 of this wave's forward column
  float b1v[RC];
  float sse;                     // thread-private partial of the squared error, fixed tile order
  f32x16 dw0[K::NB];             // dW0[:, cb] row blocks, accumulated over the tiles
  float4 db0;
  float4 dw1[RC];
  float db1[RC];
};

// weights -> LDS once per workgroup; biases -> registers.  Every global load is issued before the first LDS write (a
// load-store loop would serialise 32 HBM round trips per thread).  All threads of the block call it; threads >= NT idle.
// Ends WITHOUT a barrier: the first tile's leading __syncthreads orders the staging.
template <int RD, int RC>
__device__ __forceinline__ void head_begin(HeadLds<RD>& L, HeadState<RD, RC>& S, const float* __restrict__ W0,
                                           const float* __restrict__ b0, const float* __restrict__ W1,
                                           const float* __restrict__ b1, int C) {
  using K = HC<RD>;
  constexpr int RK = K::RK, NT = K::NT, WS0 = K::WS0, NB = K::NB;
  constexpr bool W0_LDS = K::W0_LDS;
  const bool active = threadIdx.x < NT;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31;
  S.sse = 0.f;
#pragma unroll
  for (int mb = 0; mb < NB; ++mb)
#pragma unroll
    for (int i = 0; i < 16; ++i) S.dw0[mb][i] = 0.f;
  S.db0 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int c = 0; c < RC; ++c) { S.dw1[c] = make_float4(0.f, 0.f, 0.f, 0.f); S.db1[c] = 0.f; S.b1v[c] = 0.f; }
  S.bz = 0.f;
  if (!active) return;
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
  S.bz = b0[(wave % NB) * 32 + r];
#pragma unroll
  for (int c = 0; c < RC; ++c) S.b1v[c] = b1[c < C ? c : 0];
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

// One tile: rows [0, n) are graphs gmap(row) (ascending), rows >= n are padding.  `BACKWARD` false: forward + squared
// error only.  Every thread of the block must call it (workgroup barriers inside); threads >= NT only take the barriers.
template <int RD, int RC, bool BACKWARD, class GMap>
__device__ __forceinline__ void head_tile(HeadLds<RD>& L, HeadState<RD, RC>& S, GMap gmap, int n, int C, float slope,
                                          const float* __restrict__ emb, const float* __restrict__ y,
                                          const float* __restrict__ W0, float* __restrict__ z, float* __restrict__ out,
                                          float* __restrict__ demb) {
  using K = HC<RD>;
  constexpr int RK = K::RK, NB = K::NB, NT = K::NT, ES = K::ES, ZS = K::ZS, WS0 = K::WS0, OJ = K::OJ, QN = K::QN;
  constexpr bool W0_LDS = K::W0_LDS;
  const bool active = threadIdx.x < NT;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int nb = wave % NB, kh = (wave / NB) & 1;              // forward: output column block / K half of this wave
  const int orow = (threadIdx.x / OJ) & (RT - 1), oj = threadIdx.x % OJ;   // out projection: OJ threads per graph row
  const int last = n > 0 ? n - 1 : 0;
  // ---- targets + emb rows: every load issued before the first LDS write
  float yv[RC];
  constexpr int PER_ROW = RK / 4, ITER = RT * PER_ROW / NT;
  static_assert(RT * PER_ROW % NT == 0, "rows per thread");
  float4 ev[ITER];
  if (active) {
    const int gy = gmap(orow < n ? orow : last);
#pragma unroll
    for (int c = 0; c < RC; ++c) yv[c] = y[(size_t)gy * C + (c < C ? c : C - 1)];
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
      const int idx = threadIdx.x + it * NT, row = idx / PER_ROW, c4 = idx - row * PER_ROW;
      ev[it] = *reinterpret_cast<const float4*>(emb + (size_t)gmap(row < n ? row : last) * RK + 4 * c4);
    }
  }
  __syncthreads();                                      // previous tile's readers are done (also orders the weight staging)
  if (active) {
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
      const int idx = threadIdx.x + it * NT, row = idx / PER_ROW, c4 = idx - row * PER_ROW;
      if (row >= n) ev[it] = make_float4(0.f, 0.f, 0.f, 0.f);
      *reinterpret_cast<float4*>(L.e + row * ES + 4 * c4) = ev[it];
    }
  }
  __syncthreads();
  // ---- z = LeakyReLU(emb W0^T + b0): wave (nb, kh) = output column block x K half
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  if (active) {
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
      // W0 fragments straight from global memory / L2 (every workgroup reads the same 131 KB): four float4 in flight
      // (eight, beside the backward's accumulators that live across the tile loop, spilled 82 registers)
      constexpr int WB = 4;
      const float* wg = W0 + (size_t)(nb * 32 + r) * RK + RD * kh + 4 * h;
#pragma unroll 1
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
    results_fence(acc);
    if (kh == 1) {
#pragma unroll
      for (int i = 0; i < 16; ++i) L.part[nb][krow(i, h) * 33 + r] = acc[i];
    }
  }
  __syncthreads();
  if (active && kh == 0) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int row = krow(i, h);
      const float v = hcg_leaky((acc[i] + L.part[nb][row * 33 + r]) + S.bz, slope);
      L.z[row * ZS + nb * 32 + r] = v;
      if (row < n) z[(size_t)gmap(row) * RD + nb * 32 + r] = v;
    }
  }
  __syncthreads();
  if (active) {
    // out[row][c] = z[row] . W1[c] + b1[c] ; diff = out - y.  Thread (row, j): hidden units 8j..8j+7 of every class,
    // the OJ partial sums of a row meet by xor-shuffles (fixed order)
    const float4 za = *reinterpret_cast<const float4*>(L.z + orow * ZS + 8 * oj);
    const float4 zb = *reinterpret_cast<const float4*>(L.z + orow * ZS + 8 * oj + 4);
    const int go = gmap(orow < n ? orow : last);
#pragma unroll
    for (int c = 0; c < RC; ++c) {
      if (c < C) {                                     // block-uniform
        const float4 wa = *reinterpret_cast<const float4*>(L.w1 + c * RD + 8 * oj);
        const float4 wb = *reinterpret_cast<const float4*>(L.w1 + c * RD + 8 * oj + 4);
        float s = __fmaf_rn(zb.w, wb.w, __fmaf_rn(zb.z, wb.z, __fmaf_rn(zb.y, wb.y, __fmaf_rn(zb.x, wb.x,
                  __fmaf_rn(za.w, wa.w, __fmaf_rn(za.z, wa.z, __fmaf_rn(za.y, wa.y, za.x * wa.x)))))));
#pragma unroll
        for (int off = 1; off < OJ; off <<= 1) s += __shfl_xor(s, off, 64);
        if (oj == 0) {
          float d = 0.f;
          if (orow < n) {
            s += S.b1v[c];
            out[(size_t)go * C + c] = s;
            d = s - yv[c];
            S.sse += d * d;
          }
          L.diff[orow][c] = d;
        }
      } else if (oj == 0) {
        L.diff[orow][c] = 0.f;
      }
    }
  }
  if (!BACKWARD) return;
  __syncthreads();
  // ---- backward of SSE / 2 on the unscaled diff
  const int q = threadIdx.x % QN, rgrp = (threadIdx.x / QN) & 15;     // step 1: float4 column group / row (16 rows per pass)
  const int cb = wave & (K::HW - 1);                   // this wave's 32-column block of the 2RD-wide embedding
  // 1. dz = (dout W1) * leaky'(z) -> L.z ; db0, dW1, db1 partial sums.  Two (row, 4-column) slots per thread.
  float4 dzv[2];
  if (active) {
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int row = it * 16 + rgrp;
      float4 d = make_float4(0.f, 0.f, 0.f, 0.f);
      if (row < n) {
        const float4 zz = *reinterpret_cast<const float4*>(L.z + row * ZS + 4 * q);
#pragma unroll
        for (int c = 0; c < RC; ++c) {
          if (c < C) {
            const float go = L.diff[row][c];
            const float4 w = *reinterpret_cast<const float4*>(L.w1 + c * RD + 4 * q);
            d.x += go * w.x; d.y += go * w.y; d.z += go * w.z; d.w += go * w.w;
            S.dw1[c].x += go * zz.x; S.dw1[c].y += go * zz.y; S.dw1[c].z += go * zz.z; S.dw1[c].w += go * zz.w;
            if (q == 0) S.db1[c] += go;
          }
        }
        d.x *= hcg_leaky_grad(zz.x, slope); d.y *= hcg_leaky_grad(zz.y, slope);
        d.z *= hcg_leaky_grad(zz.z, slope); d.w *= hcg_leaky_grad(zz.w, slope);
        S.db0.x += d.x; S.db0.y += d.y; S.db0.z += d.z; S.db0.w += d.w;
      }
      dzv[it] = d;
    }
  }
  __syncthreads();                                   // every read of z is done
  if (active) {
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int row = it * 16 + rgrp;
      *reinterpret_cast<float4*>(L.z + row * ZS + 4 * q) = dzv[it];
    }
  }
  __syncthreads();
  if (active) {
    // 2. dW0[:, cb] += dz^T emb[:, cb]   (K = graph rows)
#pragma unroll
    for (int s = 0; s < RT / 2; ++s) {
      const int row = 2 * s + h;
      const float bv = L.e[row * ES + cb * 32 + r];
#pragma unroll
      for (int mb = 0; mb < NB; ++mb)
        S.dw0[mb] = __builtin_amdgcn_mfma_f32_32x32x2f32(L.z[row * ZS + mb * 32 + r], bv, S.dw0[mb], 0, 0, 0);
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
#pragma unroll 1
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
    results_fence(de);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int row = krow(i, h);
      if (row < n) demb[(size_t)gmap(row) * RK + cb * 32 + r] = de[i];
    }
  }
}

// One slab per workgroup: gradient partial sums (BACKWARD) and the partial sum of squared errors.  Every thread of the
// block calls it.
template <int RD, int RC, bool BACKWARD>
__device__ __forceinline__ void head_end(HeadLds<RD>& L, HeadState<RD, RC>& S, int C, float* __restrict__ slab,
                                         float* __restrict__ sse_out) {
  using K = HC<RD>;
  constexpr int RK = K::RK, NB = K::NB, HW = K::HW, NT = K::NT, QN = K::QN, SMALL = K::SMALL;
  const bool active = threadIdx.x < NT;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5, q = threadIdx.x % QN, cb = wave & (HW - 1);
  // block partial of the squared error: lanes -> wave (fixed xor tree) -> block (fixed order)
  float sse = S.sse;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) sse += __shfl_xor(sse, off, 64);
  __syncthreads();                                     // the last tile's readers of e / z are done
  if (active && lane == 0) L.red[wave] = sse;
  float* scratch = L.e;                                // [HW][SMALL + 8]: the emb tile is dead
  if (BACKWARD && active) {
#pragma unroll
    for (int mb = 0; mb < NB; ++mb) results_fence(S.dw0[mb]);
#pragma unroll
    for (int mb = 0; mb < NB; ++mb)
#pragma unroll
      for (int i = 0; i < 16; ++i) slab[(mb * 32 + krow(i, h)) * RK + cb * 32 + r] = S.dw0[mb][i];
    // db0 / dW1 / db1: thread (row group, q) holds partial sums for columns 4q..4q+3; combine the row groups of a wave
    // (64 / QN of them) by shuffles, the waves through LDS -- all in a fixed order
    auto fold = [](float4 v) {
#pragma unroll
      for (int off = QN; off < 64; off <<= 1) {
        v.x += __shfl_xor(v.x, off, 64); v.y += __shfl_xor(v.y, off, 64); v.z += __shfl_xor(v.z, off, 64); v.w += __shfl_xor(v.w, off, 64);
      }
      return v;
    };
    float* mine = scratch + wave * (SMALL + 8);
    const float4 db0 = fold(S.db0);
    if (lane < QN) *reinterpret_cast<float4*>(mine + 4 * q) = db0;
#pragma unroll
    for (int c = 0; c < RCMAX; ++c) {
      if (c < RC && c < C) {                           // block-uniform: classes the model does not have cost no shuffles
        const float4 v = fold(S.dw1[c < RC ? c : 0]);
        if (lane < QN) *reinterpret_cast<float4*>(mine + RD + c * RD + 4 * q) = v;
        float sc = S.db1[c < RC ? c : 0];              // lanes with q == 0 hold the partial sums
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
  if (threadIdx.x == 0) {
    float block_sse = L.red[0];
#pragma unroll
    for (int w = 1; w < HW; ++w) block_sse += L.red[w];
    sse_out[0] = block_sse;
  }
  if (BACKWARD && active) {
    for (int idx = threadIdx.x; idx < SMALL; idx += NT) {
      float sm = scratch[idx];
#pragma unroll
      for (int w = 1; w < HW; ++w) sm += scratch[w * (SMALL + 8) + idx];
      slab[RD * RK + idx] = sm;
    }
  }
}

// host side: the reduce job of `nslabs` head slabs (dW0 == nullptr: partials only)
template <int RD>
inline void head_fill_job(const float* slabs, int nslabs, int C, float* dW0, float* db0, float* dW1, float* db1,
                          hcg_reduce_job* job) {
  using K = HC<RD>;
  constexpr int RK = K::RK;
  job->slabs = slabs;
  job->sse_part = slabs + (size_t)nslabs * K::SLAB;
  job->nslabs = nslabs;
  job->slab_floats = K::SLAB;
  job->nseg = dW0 ? 4 : 0;
  job->reserved = 0;
  for (int g = 0; g < HCG_REDUCE_MAX_SEGS; ++g) job->seg[g] = hcg_reduce_seg{0, 0, 1, 1, nullptr};
  if (dW0) {
    job->seg[0] = hcg_reduce_seg{0, RD * RK, RK, RK, dW0};
    job->seg[1] = hcg_reduce_seg{RD * RK, RD, 1, 1, db0};
    job->seg[2] = hcg_reduce_seg{RD * RK + RD, (int32_t)C * RD, RD, RD, dW1};
    job->seg[3] = hcg_reduce_seg{RD * RK + RD + RCMAX * RD, (int32_t)C, 1, 1, db1};
  }
}

}  // namespace hcg_head

// ======================================================================================================================
// The same head for the TAIL of the small-graph forward launch (fused.hip), cut for latency: a workgroup of 8 waves has
// just finished its tiles and owns ~16 graphs (C3: 4096 graphs on 256 workgroups), nothing else runs on the CU while it
// finishes them, so the tail is a pure dependent chain.  Against the 32-row tile code above (measured as the tail: 11 us):
//   * 16-row tiles on v_mfma_f32_16x16x4_f32 (exact f32, same FLOP/clk as 32x32x2): no half-empty 32-row blocks;
//   * all 8 waves work (forward: 4 column blocks x 2 K halves; backward: one 16-column block of emb per wave, 4 dW0 blocks
//     + demb each), two or four independent accumulators per wave (40-cycle dependent latency vs 32-cycle issue);
//   * the weight loads are issued BEFORE the barrier that ends the tile loop, the pooled rows are taken from LDS where the
//     pooling epilogue left them (no L2 round trip), out / error / dz are ONE phase (a 16-lane butterfly gives every lane of
//     a row the error, so dz needs no barrier), five barriers per tile instead of eight.
// Same slab layout as hcg_head::HC<64> (hcg_head::head_fill_job<64> describes it).
// ======================================================================================================================
// Diagnostic builds only (tools/probe_head_tail.hip defines HCG_HEAD_STAMP): s_memtime stamps of the first workgroups' waves
#ifdef HCG_HEAD_STAMP
__device__ unsigned long long g_h16_stamp[4][8][16];
#define H16STAMP(i)                                                                                        \
  do {                                                                                                     \
    __builtin_amdgcn_sched_barrier(0);                                                                     \
    unsigned long long _t;                                                                                 \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(_t)::"memory");                              \
    __builtin_amdgcn_sched_barrier(0);                                                                     \
    if ((threadIdx.x & 63) == 0 && blockIdx.x < 4) g_h16_stamp[blockIdx.x][threadIdx.x >> 6][(i)] = _t;    \
  } while (0)
#else
#define H16STAMP(i) do { } while (0)
#endif

namespace hcg_head16 {

using hcg_head::RCMAX;
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void results_fence(f32x4& a) { asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" : "+v"(a)); }

// c += A B, IN PLACE.  The builtin lets hipcc write the first result of a chain next to a shared zero quad and free the
// intermediate registers of the chain at once; a VALU instruction may then reuse one of them while the chain's next MFMA --
// queued behind the other wave of the SIMD in the shared matrix pipe -- has not read it as srcC yet (the window the ISA
// lint, tools/isa_lint.py, closes at 19 wait states).  Tied operands keep every chain in its own four registers; reads of a
// finished chain go through results_fence.
__device__ __forceinline__ void mfma16(f32x4& c, float a, float b) {
  asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}

constexpr int RD = 64, RK = 128, RT = 16, NT = 512;
constexpr int ES = RK + 4, ZS = RD + 4, WS0 = RK + 1;
constexpr int SMALL = hcg_head::HC<RD>::SMALL, SLAB = hcg_head::HC<RD>::SLAB;

struct Lds {
  float e[RT * ES];              // pooled rows of a tile when they do not sit in the caller's LDS rows
  float z[RT * ZS];              // z, then dz
  float part[4][RT * 17];        // K-half partial sums of the forward GEMM, per 16-column block
  float small[4][SMALL + 8];     // db0 | dW1 | db1 of waves 0..3
  float red[8];
};

// Registers that cross the barrier at the end of the tile loop: the head's weights never pass through LDS.  An MFMA sums over
// k in whatever order A and B agree on, so lane (r16, kq) takes the CONTIGUOUS k range 16 kq .. 16 kq + 15 of its half:
// the forward's B fragment is four float4 of one W0 row straight from L2, its A fragment four ds_read_b128 of one pooled
// row; the backward's B fragment (W0 columns) is 16 dwords, coalesced over r16.  (First form: a [64][129] image in LDS,
// staged behind the barrier, read with 4-way bank conflicts: 1500 + 1600 cycles of staging and 2500 / 4300 cycles in the
// two MFMA phases -- tools/probe_head_tail.hip.)
template <int RC>
struct Prefetch {
  float4 w0f[4];                 // forward:  W0[16 cb + r16][64 kh + 16 kq + 0..15]
  float w0b[16];                 // backward: W0[16 kq + s][16 nb + r16], s = 0..15
  float4 w1r[RC];                // W1[c][4 q .. 4 q + 3]
  float bz;
};

template <int RC>
struct State {
  float b1v[RC];
  float sse;
  f32x4 dw0[4];                  // dW0[16 mb .. 16 mb + 15][16 nb .. 16 nb + 15], nb = wave
  float4 db0;
  float4 dw1[RC];
  float db1[RC];
};

// loads only (no LDS access): safe while other waves are still inside their tiles
template <int RC>
__device__ __forceinline__ void prefetch(Prefetch<RC>& P, const float* __restrict__ W0, const float* __restrict__ b0,
                                         const float* __restrict__ W1, int C) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r16 = lane & 15, kq = lane >> 4, cb = wave & 3, kh = wave >> 2, q = threadIdx.x & 15;
  const float* wf = W0 + (size_t)(cb * 16 + r16) * RK + 64 * kh + 16 * kq;
#pragma unroll
  for (int i = 0; i < 4; ++i) P.w0f[i] = *reinterpret_cast<const float4*>(wf + 4 * i);
  const float* wb = W0 + (size_t)(16 * kq) * RK + wave * 16 + r16;
#pragma unroll
  for (int s = 0; s < 16; ++s) P.w0b[s] = wb[(size_t)s * RK];
#pragma unroll
  for (int c = 0; c < RC; ++c) P.w1r[c] = *reinterpret_cast<const float4*>(W1 + (size_t)(c < C ? c : 0) * RD + 4 * q);
  P.bz = b0[cb * 16 + r16];
}

template <int RC>
__device__ __forceinline__ void begin(State<RC>& S, const float* __restrict__ b1, int C) {
  S.sse = 0.f;
#pragma unroll
  for (int mb = 0; mb < 4; ++mb) S.dw0[mb] = f32x4{0.f, 0.f, 0.f, 0.f};
  S.db0 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int c = 0; c < RC; ++c) { S.dw1[c] = make_float4(0.f, 0.f, 0.f, 0.f); S.db1[c] = 0.f; S.b1v[c] = b1[c < C ? c : 0]; }
}

// One tile of n <= 16 graphs.  `erows`: their pooled rows in LDS (stride ES), or nullptr -> copied from global `emb`.
// The caller has put a workgroup barrier between the writers of `erows` and this call.
template <int RC, bool BACKWARD, class GMap>
__device__ __forceinline__ void tile(Lds& L, State<RC>& S, const Prefetch<RC>& P, GMap gmap, int n, int C, float slope,
                                     const float* erows, const float* __restrict__ emb, const float* __restrict__ y,
                                     float* __restrict__ z, float* __restrict__ out, float* __restrict__ demb, bool first) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r16 = lane & 15, kq = lane >> 4;
  const int cb = wave & 3, kh = wave >> 2;
  const int prow = threadIdx.x >> 4, q = threadIdx.x & 15;      // out / dz phase (threads < 256): row, 4 hidden units
  const int last = n - 1;
  float yv[RC];
  if (threadIdx.x < 256) {
    const int gy = gmap(prow < n ? prow : last);
#pragma unroll
    for (int c = 0; c < RC; ++c) yv[c] = y[(size_t)gy * C + (c < C ? c : C - 1)];
  }
  if (erows == nullptr) {       // block-uniform: more graphs per workgroup than the caller keeps in LDS
    const int row = threadIdx.x >> 5, c4 = threadIdx.x & 31;     // 16 rows x 32 float4
    const float4 v = *reinterpret_cast<const float4*>(emb + (size_t)gmap(row < n ? row : last) * RK + 4 * c4);
    __syncthreads();                                             // the previous tile's readers of e / z / part are done
    *reinterpret_cast<float4*>(L.e + row * ES + 4 * c4) = v;
    erows = L.e;
    __syncthreads();
  } else if (!first) {
    __syncthreads();                                             // the previous tile's readers of z / part are done
  }
  H16STAMP(3);
  // ---- z = LeakyReLU(emb W0^T + b0): wave = (16-column block cb, K half kh); rows >= n enter as zeros
  f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
  asm volatile("" : "+v"(a0), "+v"(a1));               // (two accumulators of their own, not one shared zero quad)
  {
    const float* er = erows + r16 * ES + 64 * kh + 16 * kq;
    float4 ev[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) ev[i] = *reinterpret_cast<const float4*>(er + 4 * i);
    if (!(r16 < n)) {
#pragma unroll
      for (int i = 0; i < 4; ++i) ev[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      mfma16(a0, ev[i].x, P.w0f[i].x);
      mfma16(a1, ev[i].y, P.w0f[i].y);
      mfma16(a0, ev[i].z, P.w0f[i].z);
      mfma16(a1, ev[i].w, P.w0f[i].w);
    }
    results_fence(a0);
    results_fence(a1);
#pragma unroll
    for (int j = 0; j < 4; ++j) a0[j] += a1[j];
    if (kh == 1) {
#pragma unroll
      for (int j = 0; j < 4; ++j) L.part[cb][(4 * kq + j) * 17 + r16] = a0[j];
    }
  }
  H16STAMP(4);
  __syncthreads();
  if (kh == 0) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int row = 4 * kq + j;
      const float v = hcg_leaky((a0[j] + L.part[cb][row * 17 + r16]) + P.bz, slope);
      L.z[row * ZS + cb * 16 + r16] = v;
      if (row < n) z[(size_t)gmap(row) * RD + cb * 16 + r16] = v;
    }
  }
  H16STAMP(5);
  __syncthreads();
  // ---- out, error, dz in one phase: thread (row, q) owns hidden units 4q .. 4q+3 of its row; a 16-lane butterfly leaves
  //      the row's output in every lane, so the error -- and with it dz -- needs no further exchange
  if (threadIdx.x < 256) {
    const float4 zz = *reinterpret_cast<const float4*>(L.z + prow * ZS + 4 * q);
    const bool live = prow < n;
    float4 dz = make_float4(0.f, 0.f, 0.f, 0.f);
    const int go = gmap(live ? prow : last);
#pragma unroll
    for (int c = 0; c < RC; ++c) {
      if (c < C) {                                     // block-uniform
        const float4 w = P.w1r[c];
        // (explicit fma chain: left to the compiler, the forward-only and the training instantiation contracted this
        //  expression differently and their outputs differed in the last bit)
        float s = __fmaf_rn(zz.w, w.w, __fmaf_rn(zz.z, w.z, __fmaf_rn(zz.y, w.y, zz.x * w.x)));
#pragma unroll
        for (int off = 1; off < 16; off <<= 1) s += __shfl_xor(s, off, 64);
        s += S.b1v[c];
        const float d = live ? s - yv[c] : 0.f;
        if (q == 0 && live) {
          out[(size_t)go * C + c] = s;
          S.sse += d * d;
        }
        if (BACKWARD) {
          dz.x += d * w.x; dz.y += d * w.y; dz.z += d * w.z; dz.w += d * w.w;
          S.dw1[c].x += d * zz.x; S.dw1[c].y += d * zz.y; S.dw1[c].z += d * zz.z; S.dw1[c].w += d * zz.w;
          if (q == 0) S.db1[c] += d;
        }
      }
    }
    if (BACKWARD) {
      dz.x *= hcg_leaky_grad(zz.x, slope); dz.y *= hcg_leaky_grad(zz.y, slope);
      dz.z *= hcg_leaky_grad(zz.z, slope); dz.w *= hcg_leaky_grad(zz.w, slope);
      S.db0.x += dz.x; S.db0.y += dz.y; S.db0.z += dz.z; S.db0.w += dz.w;
      *reinterpret_cast<float4*>(L.z + prow * ZS + 4 * q) = dz;       // (own entries only: nobody else reads this row's z)
    }
  }
  if (!BACKWARD) return;
  H16STAMP(6);
  __syncthreads();
  H16STAMP(7);
  // ---- backward on the matrix cores: wave nb owns columns 16 nb .. 16 nb + 15 of the 128-wide embedding
  {
    const int nb = wave;
    // dW0[:, nb] += dz^T emb[:, nb]   (K = the tile's 16 graph rows, lane kq: rows 4 kq .. 4 kq + 3; four independent accumulators)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int row = 4 * kq + ks;
      const float bv = row < n ? erows[row * ES + nb * 16 + r16] : 0.f;
#pragma unroll
      for (int mb = 0; mb < 4; ++mb) mfma16(S.dw0[mb], L.z[row * ZS + mb * 16 + r16], bv);
    }
    // demb[:, nb] = dz W0[:, nb]   (K = 64 hidden units, lane kq: units 16 kq .. 16 kq + 15)
    f32x4 d0 = {0.f, 0.f, 0.f, 0.f}, d1 = {0.f, 0.f, 0.f, 0.f};
    asm volatile("" : "+v"(d0), "+v"(d1));             // (two accumulators of their own, not one shared zero quad)
    const float* zr = L.z + r16 * ZS + 16 * kq;
    float4 zv[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) zv[i] = *reinterpret_cast<const float4*>(zr + 4 * i);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      mfma16(d0, zv[i].x, P.w0b[4 * i]);
      mfma16(d1, zv[i].y, P.w0b[4 * i + 1]);
      mfma16(d0, zv[i].z, P.w0b[4 * i + 2]);
      mfma16(d1, zv[i].w, P.w0b[4 * i + 3]);
    }
    results_fence(d0);
    results_fence(d1);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int row = 4 * kq + j;
      if (row < n) demb[(size_t)gmap(row) * RK + nb * 16 + r16] = d0[j] + d1[j];
    }
  }
  H16STAMP(8);
}

template <int RC, bool BACKWARD>
__device__ __forceinline__ void end(Lds& L, State<RC>& S, int C, float* __restrict__ slab, float* __restrict__ sse_out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r16 = lane & 15, kq = lane >> 4, q = threadIdx.x & 15;
  if (BACKWARD) {
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) results_fence(S.dw0[mb]);
#pragma unroll
    for (int mb = 0; mb < 4; ++mb)
#pragma unroll
      for (int j = 0; j < 4; ++j) slab[(mb * 16 + 4 * kq + j) * RK + wave * 16 + r16] = S.dw0[mb][j];
  }
  H16STAMP(9);
  // db0 / dW1 / db1 / SSE: thread (row, q) of waves 0..3 holds partial sums for hidden units 4q..4q+3 of its row; the
  // four rows of a wave meet by shuffles, the four waves through LDS -- a fixed order
  auto fold = [](float4 v) {
#pragma unroll
    for (int off = 16; off < 64; off <<= 1) {
      v.x += __shfl_xor(v.x, off, 64); v.y += __shfl_xor(v.y, off, 64); v.z += __shfl_xor(v.z, off, 64); v.w += __shfl_xor(v.w, off, 64);
    }
    return v;
  };
  float sse = S.sse;                                   // lives in the q == 0 lanes (0, 16, 32, 48)
  sse += __shfl_xor(sse, 16, 64);
  sse += __shfl_xor(sse, 32, 64);
  if (wave < 4) {
    if (lane == 0) L.red[wave] = sse;
    if (BACKWARD) {
      float* mine = L.small[wave];
      const float4 db0 = fold(S.db0);
      if (lane < 16) *reinterpret_cast<float4*>(mine + 4 * q) = db0;
#pragma unroll
      for (int c = 0; c < RCMAX; ++c) {
        if (c < RC && c < C) {                         // block-uniform
          const float4 v = fold(S.dw1[c < RC ? c : 0]);
          if (lane < 16) *reinterpret_cast<float4*>(mine + RD + c * RD + 4 * q) = v;
          float sc = S.db1[c < RC ? c : 0];            // lanes with q == 0 hold the partial sums
          sc += __shfl_xor(sc, 16, 64);
          sc += __shfl_xor(sc, 32, 64);
          if (lane == 0) mine[RD + RCMAX * RD + c] = sc;
        } else {
          if (lane < 16) *reinterpret_cast<float4*>(mine + RD + c * RD + 4 * q) = make_float4(0.f, 0.f, 0.f, 0.f);
          if (lane == 0) mine[RD + RCMAX * RD + c] = 0.f;
        }
      }
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) sse_out[0] = ((L.red[0] + L.red[1]) + L.red[2]) + L.red[3];
  if (BACKWARD) {
    for (int idx = threadIdx.x; idx < SMALL; idx += NT)
      slab[RD * RK + idx] = ((L.small[0][idx] + L.small[1][idx]) + L.small[2][idx]) + L.small[3][idx];
  }
  H16STAMP(10);
}

}  // namespace hcg_head16
