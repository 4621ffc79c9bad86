// General-shape fp32 GEMM on the f32-input matrix cores of gfx950 (v_mfma_f32_32x32x2_f32: exact
// f32, a k-ordered fmaf chain), used by the any-shape path for
//   a4 / a10  y  = x W^T (+b, LeakyReLU)      reference: GCNConv.lin, readout nn.Linear (model/gcn.py:18-45)
//   a11       dW = dz^T x   (split-K over the node dimension, deterministic 2-stage reduce)
//             dx = dz W
// The fused per-graph kernels (fused.hip) carry their own MFMA loops; this file is the fallback
// that accepts any M, N, K and any of the three operand orientations through element strides.
#include "common.h"

namespace {

constexpr int BM = 64, BN = 64, BK = 16, LDS_STRIDE = 65;
typedef float f32x16 __attribute__((ext_vector_type(16)));

// Stage a [BK x 64] operand tile into LDS as tile[k][i] (i = m for A, n for B).
// elem(i, k) lives at base[i*si + k*sk]; the thread->element map follows the unit stride so the
// global reads are coalesced for every orientation.
__device__ __forceinline__ void stage_tile(const float* __restrict__ base, int64_t si, int64_t sk,
                                           int64_t i0, int64_t imax, int64_t k0, int64_t kmax,
                                           float (*tile)[LDS_STRIDE], int tid) {
  if (sk == 1) {  // k contiguous: 16 consecutive threads walk one row's k-range
    const int k = tid & (BK - 1);
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int i = (tid >> 4) + it * 16;
      const int64_t gi = i0 + i, gk = k0 + k;
      // unconditional, clamped load + select: a load behind a per-lane guard is a serialised memory round trip
      const float v = base[(gi < imax ? gi : imax - 1) * si + (gk < kmax ? gk : kmax - 1)];
      tile[k][i] = (gi < imax && gk < kmax) ? v : 0.f;
    }
  } else {  // i contiguous (or generic): 64 consecutive threads walk the i-range
    const int i = tid & 63;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int k = (tid >> 6) + it * 4;
      const int64_t gi = i0 + i, gk = k0 + k;
      const float v = base[(gi < imax ? gi : imax - 1) * si + (gk < kmax ? gk : kmax - 1) * sk];
      tile[k][i] = (gi < imax && gk < kmax) ? v : 0.f;
    }
  }
}

__global__ __launch_bounds__(256) void k_gemm(const float* __restrict__ A, int64_t sam, int64_t sak,
                                              const float* __restrict__ B, int64_t sbk, int64_t sbn,
                                              float* __restrict__ C, int64_t M, int64_t N, int64_t K,
                                              int64_t k_per_split, const float* __restrict__ bias, int act,
                                              float slope, float* __restrict__ partials) {
  __shared__ float As[BK][LDS_STRIDE];
  __shared__ float Bs[BK][LDS_STRIDE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int64_t m0 = (int64_t)blockIdx.y * BM, n0 = (int64_t)blockIdx.x * BN;
  const int64_t kbeg = (int64_t)blockIdx.z * k_per_split;
  const int64_t kend = (kbeg + k_per_split < K) ? kbeg + k_per_split : K;

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;

  for (int64_t k0 = kbeg; k0 < kend; k0 += BK) {
    stage_tile(A, sam, sak, m0, M, k0, kend, As, tid);
    stage_tile(B, sbn, sbk, n0, N, k0, kend, Bs, tid);
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < BK; kk += 2) {
      const float a = As[kk + (lane >> 5)][wr * 32 + (lane & 31)];
      const float b = Bs[kk + (lane >> 5)][wc * 32 + (lane & 31)];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    __syncthreads();
  }

  // C/D map of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
  const int64_t n = n0 + wc * 32 + (lane & 31);
  if (n >= N) return;
  const bool direct = (partials == nullptr);
  const float bv = (direct && bias) ? bias[n] : 0.f;
  float* dst = direct ? C : partials + (size_t)blockIdx.z * M * N;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int64_t m = m0 + wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    if (m < M) {
      float v = acc[r];
      if (direct) {
        v += bv;
        if (act) v = hcg_leaky(v, slope);
      }
      dst[m * N + n] = v;
    }
  }
}

// C[i] = sum_z partials[z][i]: 64 outputs x 4 split-lanes per block, lanes combined in a fixed order
__global__ __launch_bounds__(256) void k_splitk_reduce(const float* __restrict__ partials, float* __restrict__ C,
                                                       int64_t MN, int64_t N, int splits,
                                                       const float* __restrict__ bias, int act, float slope) {
  __shared__ float part[4][64];
  const int o = threadIdx.x & 63, sl = threadIdx.x >> 6;
  const int64_t i = (int64_t)blockIdx.x * 64 + o;
  float s = 0.f;
  if (i < MN) {
#pragma unroll 4
    for (int z = sl; z < splits; z += 4) s += partials[(size_t)z * MN + i];
  }
  part[sl][o] = s;
  __syncthreads();
  if (sl == 0 && i < MN) {
    float t = ((part[0][o] + part[1][o]) + part[2][o]) + part[3][o];
    if (bias) t += bias[i % N];
    if (act) t = hcg_leaky(t, slope);
    C[i] = t;
  }
}

constexpr int CS_MAX_PARTS = 1024;  // partial rows after stage 1 (enough workgroups to fill 256 CUs several times over)

// partial[b][d] = sum over the row chunk of block b of src[m][d] * (mask ? leaky'(mask[m][d]) : 1);
// 64 feature lanes x 4 row lanes per block, the 4 row lanes combined in a fixed order
__global__ __launch_bounds__(256) void k_colsum_stage1(const float* __restrict__ src, const float* __restrict__ mask,
                                                       float slope, float* __restrict__ partial, int64_t M, int64_t D,
                                                       int64_t rows_per_block) {
  __shared__ float part[4][64];
  const int fl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int64_t d = (int64_t)blockIdx.x * 64 + fl;
  const int64_t mbeg = (int64_t)blockIdx.y * rows_per_block;
  const int64_t mend = mbeg + rows_per_block < M ? mbeg + rows_per_block : M;
  float s = 0.f;
  if (d < D) {
    // four rows in flight per thread (independent loads), added in row order: the result does not depend on the unroll
    int64_t m = mbeg + rl;
    for (; m + 12 < mend; m += 16) {
      float v0 = src[m * D + d], v1 = src[(m + 4) * D + d], v2 = src[(m + 8) * D + d], v3 = src[(m + 12) * D + d];
      if (mask) {
        v0 *= hcg_leaky_grad(mask[m * D + d], slope); v1 *= hcg_leaky_grad(mask[(m + 4) * D + d], slope);
        v2 *= hcg_leaky_grad(mask[(m + 8) * D + d], slope); v3 *= hcg_leaky_grad(mask[(m + 12) * D + d], slope);
      }
      s = (((s + v0) + v1) + v2) + v3;
    }
    for (; m < mend; m += 4) {
      float v = src[m * D + d];
      if (mask) v *= hcg_leaky_grad(mask[m * D + d], slope);
      s += v;
    }
  }
  part[rl][fl] = s;
  __syncthreads();
  if (rl == 0 && d < D) partial[(size_t)blockIdx.y * D + d] = ((part[0][fl] + part[1][fl]) + part[2][fl]) + part[3][fl];
}

__global__ __launch_bounds__(256) void k_colsum_stage2(const float* __restrict__ partial, float* __restrict__ out,
                                                       int64_t nb, int64_t D) {
  const int64_t d = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (d >= D) return;
  float s = 0.f;
#pragma unroll 8
  for (int64_t b = 0; b < nb; ++b) s += partial[(size_t)b * D + d];
  out[d] = s;
}

}  // namespace

size_t hcg_gemm_partial_floats(int64_t M, int64_t N, int64_t K, int* splits_out) {
  const int64_t tiles = hcg_cdiv(M, BM) * hcg_cdiv(N, BN);
  int64_t splits = 1;
  if (tiles < 256 && K >= 512) {
    // few output tiles and a long contraction (weight gradients: K = rows of the batch): each tile's K loop is a chain of
    // dependent stage-and-multiply rounds, so spread it -- ~2 workgroups per CU, at least 128 of K per split
    splits = 512 / tiles;
    const int64_t max_splits = hcg_cdiv(K, 128);
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
  }
  if (splits_out) *splits_out = (int)splits;
  return splits > 1 ? (size_t)splits * M * N : 0;
}

int hcg_gemm(const float* A, int64_t sam, int64_t sak, const float* B, int64_t sbk, int64_t sbn, float* C,
             int64_t M, int64_t N, int64_t K, const float* bias, int act, float slope, float* partials,
             size_t partial_floats, hipStream_t stream) {
  if (M <= 0 || N <= 0) return HCG_OK;
  if (K <= 0) {  // empty contraction: C = act(bias)
    hipLaunchKernelGGL(k_splitk_reduce, dim3((unsigned)hcg_cdiv(M * N, 64)), dim3(256), 0, stream,
                       (const float*)nullptr, C, M * N, N, 0, bias, act, slope);
    HCG_CHECK_LAUNCH();
    return HCG_OK;
  }
  int splits = 1;
  const size_t need = hcg_gemm_partial_floats(M, N, K, &splits);
  if (splits > 1 && (partials == nullptr || partial_floats < need)) splits = 1;  // degrade, never fail
  int64_t kper = hcg_cdiv(hcg_cdiv(K, splits), BK) * BK;
  splits = (int)hcg_cdiv(K, kper);
  dim3 grid((unsigned)hcg_cdiv(N, BN), (unsigned)hcg_cdiv(M, BM), (unsigned)splits);
  if (grid.y > 65535u) return HCG_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(k_gemm, grid, dim3(256), 0, stream, A, sam, sak, B, sbk, sbn, C, M, N, K, kper, bias, act,
                     slope, splits > 1 ? partials : (float*)nullptr);
  HCG_CHECK_LAUNCH();
  if (splits > 1) {
    hipLaunchKernelGGL(k_splitk_reduce, dim3((unsigned)hcg_cdiv(M * N, 64)), dim3(256), 0, stream,
                       (const float*)partials, C, M * N, N, splits, bias, act, slope);
    HCG_CHECK_LAUNCH();
  }
  return HCG_OK;
}

static int64_t colsum_parts(int64_t M) {
  int64_t nb = hcg_cdiv(M > 0 ? M : 1, 256);
  return nb > CS_MAX_PARTS ? CS_MAX_PARTS : nb;
}

size_t hcg_colsum_partial_floats(int64_t M, int64_t D) { return (size_t)colsum_parts(M) * D; }

// masked variant is reached through hcg_colsum_masked (declared in layer.hip)
int hcg_colsum_masked(const float* src, const float* mask, float slope, float* out, int64_t M, int64_t D,
                      float* partials, hipStream_t stream) {
  if (D <= 0) return HCG_OK;
  const int64_t nb = colsum_parts(M);
  const int64_t rows_per_block = hcg_cdiv(M > 0 ? M : 1, nb);
  dim3 g1((unsigned)hcg_cdiv(D, 64), (unsigned)nb);
  hipLaunchKernelGGL(k_colsum_stage1, g1, dim3(256), 0, stream, src, mask, slope, partials, M, D, rows_per_block);
  HCG_CHECK_LAUNCH();
  hipLaunchKernelGGL(k_colsum_stage2, dim3((unsigned)hcg_cdiv(D, 256)), dim3(256), 0, stream,
                     (const float*)partials, out, nb, D);
  HCG_CHECK_LAUNCH();
  return HCG_OK;
}

int hcg_colsum(const float* src, float* out, int64_t M, int64_t D, float* partials, hipStream_t stream) {
  return hcg_colsum_masked(src, nullptr, 0.f, out, M, D, partials, stream);
}
