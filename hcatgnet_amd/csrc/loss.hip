// Mean-squared-error loss on the device (the step right after the path).
// The reference computes `torch.sqrt(nn.MSELoss()(out, y.unsqueeze(1)))` (utils/utils_model.py:64,
// model/networks.py:32); through stock torch ops that is a subtract/square, a reduction and, in the
// backward, a fill + two elementwise kernels -- six launches of ~3-5 us around a 16 KB tensor.  Here:
// one launch forward (fixed-order block reduction: deterministic), one launch backward.
#include "common.h"

namespace {

constexpr int LT = 1024;

__global__ __launch_bounds__(LT) void k_mse_fwd(const float* __restrict__ a, const float* __restrict__ b, int64_t n,
                                                float* __restrict__ loss) {
  __shared__ float part[LT / 64];
  float s = 0.f;
  for (int64_t i = threadIdx.x; i < n; i += LT) {
    const float d = a[i] - b[i];
    s += d * d;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    float tot = 0.f;
#pragma unroll
    for (int w = 0; w < LT / 64; ++w) tot += part[w];
    loss[0] = tot / (float)n;
  }
}

// da = g * 2 (a - b) / n ; db = -da (nullable)
__global__ __launch_bounds__(256) void k_mse_bwd(const float* __restrict__ a, const float* __restrict__ b,
                                                 const float* __restrict__ g, int64_t n, float* __restrict__ da,
                                                 float* __restrict__ db) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float v = g[0] * (2.0f * (a[i] - b[i]) / (float)n);
  if (da) da[i] = v;
  if (db) db[i] = -v;
}

// loss AND its gradient in one launch (one workgroup: n = B * C is a few thousand): mode 0 = MSE, 1 = sqrt(MSE),
// HCG_LOSS_SSE = the data-parallel form with a collective (dout = a - b unscaled, [SSE, n] -> sse_tail).
__global__ __launch_bounds__(LT) void k_loss_fwd_bwd(const float* __restrict__ a, const float* __restrict__ b, int64_t n, int mode,
                                                     float* __restrict__ loss, float* __restrict__ da,
                                                     float* __restrict__ sse_tail) {
  __shared__ float part[LT / 64];
  __shared__ float bc;
  float s = 0.f;
  for (int64_t i = threadIdx.x; i < n; i += LT) {
    const float d = a[i] - b[i];
    s += d * d;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    float tot = 0.f;
#pragma unroll
    for (int w = 0; w < LT / 64; ++w) tot += part[w];
    const float mse = tot / (float)n, lv = mode ? sqrtf(mse) : mse;
    loss[0] = lv;
    loss[1] = mse;
    if (sse_tail) { sse_tail[0] = tot; sse_tail[1] = (float)n; }
    bc = mode == HCG_LOSS_SSE ? 1.0f : mode ? 1.0f / ((float)n * lv) : 2.0f / (float)n;
  }
  __syncthreads();
  const float scale = bc;
  for (int64_t i = threadIdx.x; i < n; i += LT) da[i] = scale * (a[i] - b[i]);
}

}  // namespace

extern "C" int hcg_loss_fwd_bwd(const float* a, const float* b, int64_t n, int mode, float* loss, float* da, float* sse_tail,
                                hcg_stream_t stream) {
  if (n <= 0 || !a || !b || !loss || !da || mode < 0 || mode > HCG_LOSS_SSE || (mode == HCG_LOSS_SSE && !sse_tail))
    return HCG_ERR_INVALID_ARG;
  hipLaunchKernelGGL(k_loss_fwd_bwd, dim3(1), dim3(LT), 0, (hipStream_t)stream, a, b, n, mode, loss, da, sse_tail);
  HCG_CHECK_LAUNCH();
  return HCG_OK;
}

extern "C" int hcg_mse_fwd(const float* a, const float* b, int64_t n, float* loss, hcg_stream_t stream) {
  if (n <= 0 || !a || !b || !loss) return HCG_ERR_INVALID_ARG;
  hipLaunchKernelGGL(k_mse_fwd, dim3(1), dim3(LT), 0, (hipStream_t)stream, a, b, n, loss);
  HCG_CHECK_LAUNCH();
  return HCG_OK;
}

extern "C" int hcg_mse_bwd(const float* a, const float* b, const float* grad_loss, int64_t n, float* da, float* db,
                           hcg_stream_t stream) {
  if (n <= 0 || !a || !b || !grad_loss || (!da && !db)) return HCG_ERR_INVALID_ARG;
  hipLaunchKernelGGL(k_mse_bwd, dim3((unsigned)hcg_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, a, b, grad_loss, n,
                     da, db);
  HCG_CHECK_LAUNCH();
  return HCG_OK;
}
