// Fused readout MLP for the reference's default head (model/gcn.py:36-45, 70-71):
//     z = LeakyReLU(emb W0^T + b0)   [B, 2D] -> [B, D]
//     out = z W1^T + b1              [B, D]  -> [B, C]        (D = 64, C = n_classes <= 8)
// forward = ONE launch, backward = ONE launch + a fixed-order slab reduction (a11).  The path is
// tiny (67 MFLOP at B = 4096) and purely launch/latency bound, which is why it is fused: through the
// generic GEMMs it cost 6 GEMM launches + 6 helper launches per step.
// Wave-autonomous 32-graph tiles; emb W0^T and dz^T emb / dz W0 on v_mfma_f32_32x32x2_f32.
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
// see fused.hip (mfma_results_fence): keep VALU reads of an accumulator a whole foreign MFMA away from the chain's
// last MFMA when several waves share the SIMD's matrix pipe (f32 32x32x2: 16 passes = 64 cycles)
__device__ __forceinline__ void mfma_results_fence(f32x16& a) { asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" : "+v"(a)); }

constexpr int RD = 64;            // hidden width
constexpr int RK = 2 * RD;        // pooled embedding width
constexpr int RT = 32;            // graphs per wave tile
constexpr int ES = RK + 4;        // LDS stride of the emb tile
constexpr int ZS = RD + 4;        // LDS stride of the z / dz tile
constexpr int RWAVES = 4;
constexpr int RCMAX = 8;

struct RLds {
  float e[RT * ES];
  float z[RT * ZS];
};

// (loads are unconditional with a clamped row: a per-lane guard would serialise them, see fused.hip)
__device__ __forceinline__ void stage_emb(float* buf, const float* __restrict__ emb, int g0, int n, int B, int lane) {
  const int q = lane & 31, r2 = lane >> 5;  // 32 lanes x 16 B = one 512-byte row
  float4 v[RT / 2];
#pragma unroll
  for (int it = 0; it < RT / 2; ++it) {
    int row = g0 + it * 2 + r2;
    if (row > B - 1) row = B - 1;
    v[it] = *reinterpret_cast<const float4*>(emb + (size_t)row * RK + 4 * q);
  }
#pragma unroll
  for (int it = 0; it < RT / 2; ++it) {
    const int row = it * 2 + r2;
    if (row >= n) v[it] = make_float4(0.f, 0.f, 0.f, 0.f);
    *reinterpret_cast<float4*>(buf + row * ES + 4 * q) = v[it];
  }
}

// forward: a workgroup = 4 waves = 2 tiles; wave (pair p = wave >> 1, nb = wave & 1) computes the
// 32 output columns [32 nb, 32 nb + 32) of tile p's z (64 MFMAs), the two halves of out = z W1^T
// meet in LDS.  Splitting columns over waves spreads this launch-latency-bound head over 4x more SIMDs.
__global__ __launch_bounds__(RWAVES * 64, 1) void k_readout_fwd(const float* __restrict__ emb, const float* __restrict__ W0,
                                                                const float* __restrict__ b0, const float* __restrict__ W1,
                                                                const float* __restrict__ b1, int B, int C, float slope,
                                                                float* __restrict__ z, float* __restrict__ out) {
  __shared__ RLds lds[RWAVES];
  __shared__ float opart[2][RT][RCMAX];      // nb == 1 half of the out dot products, per tile pair
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  RLds& L = lds[wave];
  const int r = lane & 31, h = lane >> 5;
  const int pair = wave >> 1, nb = wave & 1;
  const int tiles = (B + RT - 1) / RT;

  // this wave's weight columns: B[k][j] = W0[nb*32 + j][k]; k-step s = 4t+u <-> k = 8t + 4h + u
  float wreg[RK / 2];
  {
    float* wl = reinterpret_cast<float*>(&lds[0]);   // W0 [64][128] -> LDS image [n][RK + 1], coalesced, once
    for (int idx = threadIdx.x; idx < RD * RK; idx += RWAVES * 64) wl[(idx / RK) * (RK + 1) + (idx % RK)] = W0[idx];
    __syncthreads();
#pragma unroll
    for (int s = 0; s < RK / 2; ++s) wreg[s] = wl[(nb * 32 + r) * (RK + 1) + 8 * (s >> 2) + 4 * h + (s & 3)];
    __syncthreads();
  }
  const float bz = b0[nb * 32 + r];

  for (int t0 = blockIdx.x * 2; t0 < tiles; t0 += gridDim.x * 2) {   // block-uniform trip count (barriers inside)
    const int t = t0 + pair;
    const bool live = t < tiles;
    const int g0 = live ? t * RT : 0;
    const int n = live ? (B - g0 < RT ? B - g0 : RT) : 0;
    stage_emb(L.e, emb, g0, n, B, lane);
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
    for (int t8 = 0; t8 < RK / 8; ++t8) {
      const float4 a = *reinterpret_cast<const float4*>(L.e + r * ES + 8 * t8 + 4 * h);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, wreg[4 * t8 + 0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, wreg[4 * t8 + 1], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, wreg[4 * t8 + 2], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, wreg[4 * t8 + 3], acc, 0, 0, 0);
    }
    mfma_results_fence(acc);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int row = (i & 3) + 8 * (i >> 2) + 4 * h;
      const float v = hcg_leaky(acc[i] + bz, slope);
      L.z[row * ZS + r] = v;                       // this wave's 32 columns, [row][0..32)
      if (row < n) z[(size_t)(g0 + row) * RD + nb * 32 + r] = v;
    }
    // out[row][c]: lane (row = r, half h) sums 16 of this wave's 32 columns; halves and the two waves combine
    float sc[RCMAX];
#pragma unroll
    for (int c = 0; c < RCMAX; ++c) {
      float s = 0.f;
      if (c < C) {
#pragma unroll
        for (int j = 0; j < 16; ++j) s += L.z[r * ZS + 16 * h + j] * W1[c * RD + nb * 32 + 16 * h + j];
      }
      s += __shfl_xor(s, 32, 64);
      sc[c] = s;
    }
    if (nb == 1 && h == 0) {
#pragma unroll
      for (int c = 0; c < RCMAX; ++c) opart[pair][r][c] = sc[c];
    }
    __syncthreads();
    if (nb == 0 && h == 0 && r < n) {
#pragma unroll
      for (int c = 0; c < RCMAX; ++c)
        if (c < C) out[(size_t)(g0 + r) * C + c] = (sc[c] + opart[pair][r][c]) + b1[c];
    }
    __syncthreads();
  }
}

// slab layout per WORKGROUP: dW0 [RD][RK] | db0 [RD] | dW1 [C][RD] | db1 [C]   (C padded to RCMAX)
constexpr int SLAB = RD * RK + RD + RCMAX * RD + RCMAX;

// backward: the 4 waves of a workgroup share each tile by OUTPUT COLUMN BLOCK: wave nb computes
// dW0[:, 32nb..] (32 MFMAs) and demb[:, 32nb..] (32 MFMAs); dz (cheap, VALU) is recomputed by every
// wave so no wave waits for another; wave 0 alone accumulates db0 / dW1 / db1.  Column blocks are
// disjoint, so the workgroup's slab needs no cross-wave reduction.
__global__ __launch_bounds__(RWAVES * 64, 1) void k_readout_bwd(const float* __restrict__ dout, const float* __restrict__ emb,
                                                                const float* __restrict__ z, const float* __restrict__ W0,
                                                                const float* __restrict__ W1, int B, int C, float slope,
                                                                float* __restrict__ demb, float* __restrict__ slabs) {
  __shared__ RLds lds[RWAVES];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  RLds& L = lds[wave];
  const int r = lane & 31, h = lane >> 5, q = lane & 15, r4 = lane >> 4;
  const int nb = wave;                       // this wave's 32-column block of the 128-wide embedding
  const int tiles = (B + RT - 1) / RT;

  // demb operand: B[k = d][j] = W0[d][nb*32 + j]; k-step s <-> d = 8t + 4h + u  (128-byte row pieces)
  float wreg[RD / 2];
#pragma unroll
  for (int s = 0; s < RD / 2; ++s) wreg[s] = W0[(size_t)(8 * (s >> 2) + 4 * h + (s & 3)) * RK + nb * 32 + r];

  f32x16 dw0[2];
#pragma unroll
  for (int mb = 0; mb < 2; ++mb)
#pragma unroll
    for (int i = 0; i < 16; ++i) dw0[mb][i] = 0.f;
  float4 db0 = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 dw1[RCMAX];
  float db1[RCMAX];
#pragma unroll
  for (int c = 0; c < RCMAX; ++c) { dw1[c] = make_float4(0.f, 0.f, 0.f, 0.f); db1[c] = 0.f; }

  for (int t = blockIdx.x; t < tiles; t += gridDim.x) {
    const int g0 = t * RT;
    const int n = B - g0 < RT ? B - g0 : RT;
    // 1. dz = (dout W1) * leaky'(z)  -> L.z  ; db0, dW1, db1 partial sums
    //    (all loads first, unconditional with clamped rows; W1 rows are loop invariants)
    float4 zr[RT / 4];
    float gor[RT / 4][RCMAX];
#pragma unroll
    for (int it = 0; it < RT / 4; ++it) {
      int row = g0 + it * 4 + r4;
      if (row > B - 1) row = B - 1;
      zr[it] = *reinterpret_cast<const float4*>(z + (size_t)row * RD + 4 * q);
#pragma unroll
      for (int c = 0; c < RCMAX; ++c) gor[it][c] = dout[(size_t)row * C + (c < C ? c : C - 1)];
    }
    float4 w1r[RCMAX];
#pragma unroll
    for (int c = 0; c < RCMAX; ++c) w1r[c] = *reinterpret_cast<const float4*>(W1 + (c < C ? c : C - 1) * RD + 4 * q);
#pragma unroll
    for (int it = 0; it < RT / 4; ++it) {
      const int row = it * 4 + r4;
      float4 d = make_float4(0.f, 0.f, 0.f, 0.f);
      if (row < n) {
        const float4 zz = zr[it];
#pragma unroll
        for (int c = 0; c < RCMAX; ++c) {
          if (c < C) {
            const float go = gor[it][c];
            const float4 w = w1r[c];
            d.x += go * w.x; d.y += go * w.y; d.z += go * w.z; d.w += go * w.w;
            if (wave == 0) {
              dw1[c].x += go * zz.x; dw1[c].y += go * zz.y; dw1[c].z += go * zz.z; dw1[c].w += go * zz.w;
              if (q == 0) db1[c] += go;
            }
          }
        }
        d.x *= hcg_leaky_grad(zz.x, slope); d.y *= hcg_leaky_grad(zz.y, slope);
        d.z *= hcg_leaky_grad(zz.z, slope); d.w *= hcg_leaky_grad(zz.w, slope);
        if (wave == 0) { db0.x += d.x; db0.y += d.y; db0.z += d.z; db0.w += d.w; }
      }
      *reinterpret_cast<float4*>(L.z + row * ZS + 4 * q) = d;
    }
    // 2. emb tile (rows >= n zero)
    stage_emb(L.e, emb, g0, n, B, lane);
    // 3. dW0[:, nb] += dz^T emb[:, nb]   (K = graph rows)
#pragma unroll
    for (int s = 0; s < RT / 2; ++s) {
      const int row = 2 * s + h;
      const float bv = L.e[row * ES + nb * 32 + r];
      dw0[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(L.z[row * ZS + r], bv, dw0[0], 0, 0, 0);
      dw0[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(L.z[row * ZS + 32 + r], bv, dw0[1], 0, 0, 0);
    }
    // 4. demb[:, nb] = dz W0[:, nb]
    f32x16 de;
#pragma unroll
    for (int i = 0; i < 16; ++i) de[i] = 0.f;
#pragma unroll
    for (int t8 = 0; t8 < RD / 8; ++t8) {
      const float4 a = *reinterpret_cast<const float4*>(L.z + r * ZS + 8 * t8 + 4 * h);
      de = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, wreg[4 * t8 + 0], de, 0, 0, 0);
      de = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, wreg[4 * t8 + 1], de, 0, 0, 0);
      de = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, wreg[4 * t8 + 2], de, 0, 0, 0);
      de = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, wreg[4 * t8 + 3], de, 0, 0, 0);
    }
    mfma_results_fence(de);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int row = (i & 3) + 8 * (i >> 2) + 4 * h;
      if (row < n) demb[(size_t)(g0 + row) * RK + nb * 32 + r] = de[i];
    }
  }

  // one slab per workgroup; each wave owns its dW0 column block, wave 0 the small vectors
  float* slab = slabs + (size_t)blockIdx.x * SLAB;
  mfma_results_fence(dw0[0]);
  mfma_results_fence(dw0[1]);
#pragma unroll
  for (int mb = 0; mb < 2; ++mb)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int d = mb * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
      slab[d * RK + nb * 32 + r] = dw0[mb][i];
    }
  if (wave == 0) {
    // combine the 4 row lanes of the float4 accumulators
    db0.x += __shfl_xor(db0.x, 16, 64); db0.y += __shfl_xor(db0.y, 16, 64); db0.z += __shfl_xor(db0.z, 16, 64); db0.w += __shfl_xor(db0.w, 16, 64);
    db0.x += __shfl_xor(db0.x, 32, 64); db0.y += __shfl_xor(db0.y, 32, 64); db0.z += __shfl_xor(db0.z, 32, 64); db0.w += __shfl_xor(db0.w, 32, 64);
    if (r4 == 0) *reinterpret_cast<float4*>(slab + RD * RK + 4 * q) = db0;
#pragma unroll
    for (int c = 0; c < RCMAX; ++c) {
      float4 v = dw1[c];
      v.x += __shfl_xor(v.x, 16, 64); v.y += __shfl_xor(v.y, 16, 64); v.z += __shfl_xor(v.z, 16, 64); v.w += __shfl_xor(v.w, 16, 64);
      v.x += __shfl_xor(v.x, 32, 64); v.y += __shfl_xor(v.y, 32, 64); v.z += __shfl_xor(v.z, 32, 64); v.w += __shfl_xor(v.w, 32, 64);
      if (r4 == 0) *reinterpret_cast<float4*>(slab + RD * RK + RD + c * RD + 4 * q) = v;
      float s = db1[c];  // lanes with q == 0 hold the partial sums (4 of them)
      s += __shfl_xor(s, 16, 64);
      s += __shfl_xor(s, 32, 64);
      if (lane == 0) slab[RD * RK + RD + RCMAX * RD + c] = s;
    }
  }
}

// backward: one workgroup per tile (4 waves = 4 column blocks); forward: one workgroup per 2 tiles
int readout_grid(int64_t B) {
  int grid = (int)((B + RT - 1) / RT);
  if (grid > 512) grid = 512;
  return grid < 1 ? 1 : grid;
}
int readout_fwd_grid(int64_t B) {
  int grid = (int)(((B + RT - 1) / RT + 1) / 2);
  if (grid > 512) grid = 512;
  return grid < 1 ? 1 : grid;
}

}  // namespace

// 1 when the fused readout applies: two readout layers [2D -> D -> C], D = 64, C <= 8
static int hcg_readout2_supported(int64_t D, int64_t C) { return (D == RD && C >= 1 && C <= RCMAX) ? 1 : 0; }

size_t hcg_readout2_workspace_bytes_impl(int64_t B) { return (size_t)readout_grid(B) * SLAB * sizeof(float) + 256; }

extern "C" int hcg_readout2_fwd(const float* emb, const float* W0, const float* b0, const float* W1, const float* b1,
                                int64_t B, int64_t D, int64_t C, float slope, float* z, float* out, hcg_stream_t stream) {
  if (!hcg_readout2_supported(D, C)) return HCG_ERR_UNSUPPORTED;
  if (B < 0) return HCG_ERR_INVALID_ARG;
  if (B == 0) return HCG_OK;
  if (!emb || !W0 || !b0 || !W1 || !b1 || !z || !out) return HCG_ERR_INVALID_ARG;
  hipLaunchKernelGGL(k_readout_fwd, dim3(readout_fwd_grid(B)), dim3(RWAVES * 64), 0, (hipStream_t)stream, emb, W0, b0, W1, b1,
                     (int)B, (int)C, slope, z, out);
  HCG_CHECK_LAUNCH();
  return HCG_OK;
}

// backward without the slab reduction (pair with hcg_readout2_reduce_job + hcg_step_tail)
extern "C" int hcg_readout2_bwd_partial(const float* dout, const float* emb, const float* z, const float* W0,
                                        const float* W1, int64_t B, int64_t D, int64_t C, float slope, float* demb,
                                        void* workspace, size_t workspace_bytes, hcg_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!hcg_readout2_supported(D, C)) return HCG_ERR_UNSUPPORTED;
  if (B <= 0 || !W0 || !W1 || !dout || !emb || !z || !demb || !workspace) return HCG_ERR_INVALID_ARG;
  const int grid = readout_grid(B);
  if (workspace_bytes < (size_t)grid * SLAB * sizeof(float)) return HCG_ERR_WORKSPACE;
  hipLaunchKernelGGL(k_readout_bwd, dim3(grid), dim3(RWAVES * 64), 0, stream, dout, emb, z, W0, W1, (int)B, (int)C, slope,
                     demb, (float*)workspace);
  HCG_CHECK_LAUNCH();
  return HCG_OK;
}

extern "C" int hcg_readout2_reduce_job(const void* workspace, size_t workspace_bytes, int64_t B, int64_t C, float* dW0,
                                       float* db0, float* dW1, float* db1, hcg_reduce_job* job) {
  if (B <= 0 || C < 1 || C > RCMAX || !dW0 || !db0 || !dW1 || !db1 || !job || !workspace) return HCG_ERR_INVALID_ARG;
  const int grid = readout_grid(B);
  if (workspace_bytes < (size_t)grid * SLAB * sizeof(float)) return HCG_ERR_WORKSPACE;
  job->slabs = (const float*)workspace;
  job->nslabs = grid;
  job->slab_floats = SLAB;
  job->nseg = 4;
  job->sse_part = nullptr;
  job->reserved = 0;
  job->seg[0] = hcg_reduce_seg{0, RD * RK, RK, RK, dW0};
  job->seg[1] = hcg_reduce_seg{RD * RK, RD, 1, 1, db0};
  job->seg[2] = hcg_reduce_seg{RD * RK + RD, (int32_t)C * RD, RD, RD, dW1};
  job->seg[3] = hcg_reduce_seg{RD * RK + RD + RCMAX * RD, (int32_t)C, 1, 1, db1};
  return HCG_OK;
}
