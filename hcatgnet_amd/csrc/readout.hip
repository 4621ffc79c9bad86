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

__global__ __launch_bounds__(RWAVES * 64, 1) void k_readout_fwd(const float* __restrict__ emb, const float* __restrict__ W0,
                                                                const float* __restrict__ b0, const float* __restrict__ W1,
                                                                const float* __restrict__ b1, int B, int C, float slope,
                                                                float* __restrict__ z, float* __restrict__ out) {
  __shared__ RLds lds[RWAVES];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  RLds& L = lds[wave];
  const int r = lane & 31, h = lane >> 5;
  const int tiles = (B + RT - 1) / RT;

  // W0 [64][128] -> LDS image [n][RK + 1] (coalesced, once per workgroup) -> 128 VGPRs per lane.
  // B[k][j] = W0[j][k]; k-step s = 4t+u <-> k = 8t + 4h + u.  (As 128 strided global loads per lane this
  // prologue was most of the kernel's 15 us.)
  float wreg[2][RK / 2];
  {
    float* wl = reinterpret_cast<float*>(&lds[0]);   // 64 * 129 floats = 33 KB of the 100 KB block
    for (int idx = threadIdx.x; idx < RD * RK; idx += RWAVES * 64) wl[(idx / RK) * (RK + 1) + (idx % RK)] = W0[idx];
    __syncthreads();
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int s = 0; s < RK / 2; ++s) wreg[nb][s] = wl[(nb * 32 + r) * (RK + 1) + 8 * (s >> 2) + 4 * h + (s & 3)];
    __syncthreads();
  }
  const float bz0 = b0[r], bz1 = b0[32 + r];

  for (int t = blockIdx.x * RWAVES + wave; t < tiles; t += gridDim.x * RWAVES) {
    const int g0 = t * RT;
    const int n = B - g0 < RT ? B - g0 : RT;
    stage_emb(L.e, emb, g0, n, B, lane);
    f32x16 acc0, acc1;
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }
#pragma unroll
    for (int t8 = 0; t8 < RK / 8; ++t8) {
      const float4 a = *reinterpret_cast<const float4*>(L.e + r * ES + 8 * t8 + 4 * h);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, wreg[0][4 * t8 + 0], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, wreg[1][4 * t8 + 0], acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, wreg[0][4 * t8 + 1], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, wreg[1][4 * t8 + 1], acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, wreg[0][4 * t8 + 2], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, wreg[1][4 * t8 + 2], acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, wreg[0][4 * t8 + 3], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, wreg[1][4 * t8 + 3], acc1, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int row = (i & 3) + 8 * (i >> 2) + 4 * h;
      const float v0 = hcg_leaky(acc0[i] + bz0, slope), v1 = hcg_leaky(acc1[i] + bz1, slope);
      L.z[row * ZS + r] = v0;
      L.z[row * ZS + 32 + r] = v1;
      if (row < n) {
        z[(size_t)(g0 + row) * RD + r] = v0;
        z[(size_t)(g0 + row) * RD + 32 + r] = v1;
      }
    }
    // out[row][c] = sum_j z[row][j] W1[c][j] + b1[c]: lane (row = r, half h) sums 32 columns
    for (int c = 0; c < C; ++c) {
      float s = 0.f;
#pragma unroll 8
      for (int j = 0; j < 32; ++j) s += L.z[r * ZS + 32 * h + j] * W1[c * RD + 32 * h + j];
      const float o = __shfl_xor(s, 32, 64);
      if (h == 0 && r < n) out[(size_t)(g0 + r) * C + c] = (s + o) + b1[c];
    }
  }
}

// slab layout per wave: dW0 [RD][RK] | db0 [RD] | dW1 [C][RD] | db1 [C]   (C padded to RCMAX)
constexpr int SLAB = RD * RK + RD + RCMAX * RD + RCMAX;

__global__ __launch_bounds__(RWAVES * 64, 1) void k_readout_bwd(const float* __restrict__ dout, const float* __restrict__ emb,
                                                                const float* __restrict__ z, const float* __restrict__ W0,
                                                                const float* __restrict__ W1, int B, int C, float slope,
                                                                float* __restrict__ demb, float* __restrict__ slabs) {
  __shared__ RLds lds[RWAVES];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  RLds& L = lds[wave];
  const int r = lane & 31, h = lane >> 5, q = lane & 15, r4 = lane >> 4;
  const int tiles = (B + RT - 1) / RT;

  // demb operand: B[k = d][j] = W0[d][nb*32 + j] read from a workgroup-shared LDS copy of W0 (row-major,
  // lanes read consecutive j: conflict-free); k-step s <-> d = 8t + 4h + u
  __shared__ float w0s[RD * RK];
  for (int idx = threadIdx.x; idx < RD * RK; idx += RWAVES * 64) w0s[idx] = W0[idx];
  __syncthreads();

  f32x16 dw0[2][4];
#pragma unroll
  for (int mb = 0; mb < 2; ++mb)
#pragma unroll
    for (int nb = 0; nb < 4; ++nb)
#pragma unroll
      for (int i = 0; i < 16; ++i) dw0[mb][nb][i] = 0.f;
  float4 db0 = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 dw1[RCMAX];
  float db1[RCMAX];
#pragma unroll
  for (int c = 0; c < RCMAX; ++c) { dw1[c] = make_float4(0.f, 0.f, 0.f, 0.f); db1[c] = 0.f; }

  for (int t = blockIdx.x * RWAVES + wave; t < tiles; t += gridDim.x * RWAVES) {
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
            dw1[c].x += go * zz.x; dw1[c].y += go * zz.y; dw1[c].z += go * zz.z; dw1[c].w += go * zz.w;
            if (q == 0) db1[c] += go;
          }
        }
        d.x *= hcg_leaky_grad(zz.x, slope); d.y *= hcg_leaky_grad(zz.y, slope);
        d.z *= hcg_leaky_grad(zz.z, slope); d.w *= hcg_leaky_grad(zz.w, slope);
        db0.x += d.x; db0.y += d.y; db0.z += d.z; db0.w += d.w;
      }
      *reinterpret_cast<float4*>(L.z + row * ZS + 4 * q) = d;
    }
    // 2. emb tile (rows >= n zero)
    stage_emb(L.e, emb, g0, n, B, lane);
    // 3. dW0 += dz^T emb   (K = graph rows)
#pragma unroll
    for (int s = 0; s < RT / 2; ++s) {
      const int row = 2 * s + h;
      const float a0 = L.z[row * ZS + r], a1 = L.z[row * ZS + 32 + r];
#pragma unroll
      for (int nb = 0; nb < 4; ++nb) {
        const float b = L.e[row * ES + nb * 32 + r];
        dw0[0][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b, dw0[0][nb], 0, 0, 0);
        dw0[1][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b, dw0[1][nb], 0, 0, 0);
      }
    }
    // 4. demb = dz W0
    f32x16 de[4];
#pragma unroll
    for (int nb = 0; nb < 4; ++nb)
#pragma unroll
      for (int i = 0; i < 16; ++i) de[nb][i] = 0.f;
#pragma unroll
    for (int t8 = 0; t8 < RD / 8; ++t8) {
      const float4 a = *reinterpret_cast<const float4*>(L.z + r * ZS + 8 * t8 + 4 * h);
      const float av[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int d = 8 * t8 + 4 * h + u;
#pragma unroll
        for (int nb = 0; nb < 4; ++nb)
          de[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], w0s[d * RK + nb * 32 + r], de[nb], 0, 0, 0);
      }
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int row = (i & 3) + 8 * (i >> 2) + 4 * h;
      if (row < n) {
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) demb[(size_t)(g0 + row) * RK + nb * 32 + r] = de[nb][i];
      }
    }
  }

  // one slab per wave (fixed-order reduction happens in k_readout_reduce)
  float* slab = slabs + (size_t)(blockIdx.x * RWAVES + wave) * SLAB;
#pragma unroll
  for (int mb = 0; mb < 2; ++mb)
#pragma unroll
    for (int nb = 0; nb < 4; ++nb)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int d = mb * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
        slab[d * RK + nb * 32 + r] = dw0[mb][nb][i];
      }
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

constexpr int RR_SLICES = 16;
__global__ __launch_bounds__(256) void k_readout_reduce(const float* __restrict__ slabs, int nslabs, int C,
                                                        float* __restrict__ dW0, float* __restrict__ db0,
                                                        float* __restrict__ dW1, float* __restrict__ db1) {
  __shared__ float part[RR_SLICES][16];
  const int o = threadIdx.x & 15, sl = threadIdx.x >> 4;
  const int idx = blockIdx.x * 16 + o;
  float s = 0.f;
  if (idx < SLAB) {
#pragma unroll 8
    for (int b = sl; b < nslabs; b += RR_SLICES) s += slabs[(size_t)b * SLAB + idx];
  }
  part[sl][o] = s;
  __syncthreads();
  if (sl == 0 && idx < SLAB) {
    float tot = 0.f;
#pragma unroll
    for (int k = 0; k < RR_SLICES; ++k) tot += part[k][o];
    if (idx < RD * RK) dW0[idx] = tot;
    else if (idx < RD * RK + RD) db0[idx - RD * RK] = tot;
    else if (idx < RD * RK + RD + RCMAX * RD) { const int j = idx - RD * RK - RD; if (j < C * RD) dW1[j] = tot; }
    else { const int c = idx - RD * RK - RD - RCMAX * RD; if (c < C) db1[c] = tot; }
  }
}

int readout_grid(int64_t B) {
  const int tiles = (int)((B + RT - 1) / RT);
  int grid = (tiles + RWAVES - 1) / RWAVES;
  if (grid > 256) grid = 256;
  return grid < 1 ? 1 : grid;
}

}  // namespace

// 1 when the fused readout applies: two readout layers [2D -> D -> C], D = 64, C <= 8
extern "C" int hcg_readout2_supported(int64_t D, int64_t C) { return (D == RD && C >= 1 && C <= RCMAX) ? 1 : 0; }

extern "C" size_t hcg_readout2_workspace_bytes(int64_t B) { return (size_t)readout_grid(B) * RWAVES * SLAB * sizeof(float) + 256; }

extern "C" int hcg_readout2_fwd(const float* emb, const float* W0, const float* b0, const float* W1, const float* b1,
                                int64_t B, int64_t D, int64_t C, float slope, float* z, float* out, hcg_stream_t stream) {
  if (!hcg_readout2_supported(D, C)) return HCG_ERR_UNSUPPORTED;
  if (B < 0) return HCG_ERR_INVALID_ARG;
  if (B == 0) return HCG_OK;
  if (!emb || !W0 || !b0 || !W1 || !b1 || !z || !out) return HCG_ERR_INVALID_ARG;
  hipLaunchKernelGGL(k_readout_fwd, dim3(readout_grid(B)), dim3(RWAVES * 64), 0, (hipStream_t)stream, emb, W0, b0, W1, b1,
                     (int)B, (int)C, slope, z, out);
  HCG_CHECK_LAUNCH();
  return HCG_OK;
}

extern "C" int hcg_readout2_bwd(const float* dout, const float* emb, const float* z, const float* W0, const float* W1,
                                int64_t B, int64_t D, int64_t C, float slope, float* demb, float* dW0, float* db0,
                                float* dW1, float* db1, void* workspace, size_t workspace_bytes, hcg_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!hcg_readout2_supported(D, C)) return HCG_ERR_UNSUPPORTED;
  if (B < 0 || !dW0 || !db0 || !dW1 || !db1 || !W0 || !W1) return HCG_ERR_INVALID_ARG;
  if (B > 0 && (!dout || !emb || !z || !demb)) return HCG_ERR_INVALID_ARG;
  const int grid = B > 0 ? readout_grid(B) : 0;
  if (workspace_bytes < (size_t)grid * RWAVES * SLAB * sizeof(float)) return HCG_ERR_WORKSPACE;
  if (grid > 0) {
    hipLaunchKernelGGL(k_readout_bwd, dim3(grid), dim3(RWAVES * 64), 0, stream, dout, emb, z, W0, W1, (int)B, (int)C,
                       slope, demb, (float*)workspace);
    HCG_CHECK_LAUNCH();
  }
  hipLaunchKernelGGL(k_readout_reduce, dim3((SLAB + 15) / 16), dim3(256), 0, stream, (const float*)workspace,
                     grid * RWAVES, (int)C, dW0, db0, dW1, db1);
  HCG_CHECK_LAUNCH();
  return HCG_OK;
}
