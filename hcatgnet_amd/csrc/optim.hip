// Adam step for the model's parameters in ONE launch per contiguous segment (the step right
// after the path).  The reference uses torch.optim.Adam(lr, eps=1e-9) (model/networks.py:38) on 8 small
// tensors (16 641 floats): through torch's foreach implementation that is ~10 multi-tensor launches and
// ~0.5 ms of host time per step -- more than the whole fwd+bwd here.  Same update rule as torch
// (amsgrad=False, weight_decay=0, maximize=False):
//   m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g^2
//   p -= (lr / (1 - b1^t)) * m / ( sqrt(v) / sqrt(1 - b2^t) + eps )
#include "common.h"

namespace {

// b^t for an integer t >= 0 by squaring, in double: a few ulp of double, far inside the float the caller rounds to
// (torch computes `1 - beta ** step` in Python doubles); ~20 multiplications instead of a library pow().
__device__ __forceinline__ double hcg_powi(double b, int t) {
  double r = 1.0;
  while (t > 0) {
    if (t & 1) r *= b;
    b *= b;
    t >>= 1;
  }
  return r;
}

__global__ __launch_bounds__(256) void k_adam(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                              float* __restrict__ v, int64_t n, float lr, float b1, float b2, float eps,
                                              float bc1, float bc2_sqrt) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float mi = m[i], vi = v[i];
  p[i] = hcg_adam_update(p[i], g[i], mi, vi, b1, b2, eps, lr / bc1, bc2_sqrt);
  m[i] = mi;
  v[i] = vi;
}

}  // namespace

// step = 1-based step count of this update; bias corrections are computed on the host in double like torch
extern "C" int hcg_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr,
                             float beta1, float beta2, float eps, int64_t step, hcg_stream_t stream) {
  if (n < 0 || step < 1 || (n > 0 && (!param || !grad || !exp_avg || !exp_avg_sq))) return HCG_ERR_INVALID_ARG;
  if (n == 0) return HCG_OK;
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  hipLaunchKernelGGL(k_adam, dim3((unsigned)hcg_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg,
                     exp_avg_sq, n, lr, beta1, beta2, eps, (float)bc1, (float)sqrt(bc2));
  HCG_CHECK_LAUNCH();
  return HCG_OK;
}

namespace {

// step count and learning rate read from device memory (hipGraph-capturable).  Bias corrections in double
// like torch's host computation.  The last workgroup to take a ticket publishes step + 1.
// SSE: `g` = [n summed SSE/2-gradients | SSE | count] (data-parallel form HCG_LOSS_SSE): the
// gradient of sqrt(MSE) over all ranks' graphs is g * 1 / (count * sqrt(SSE / count)); written back in place.
template <bool SSE>
__global__ __launch_bounds__(256) void k_adam_dev(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                  float* __restrict__ v, int64_t n, const float* __restrict__ lr_dev, float b1,
                                                  float b2, float eps, int* __restrict__ step_dev, float* __restrict__ loss) {
  const int t = step_dev[0] + 1;
  const float lr = lr_dev[0];
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  float gs = 1.0f;
  if (SSE) {
    const float sse = g[n], cnt = g[n + 1], mse = sse / cnt, lv = sqrtf(mse);
    gs = 1.0f / (cnt * lv);
    if (i == 0) { loss[0] = lv; loss[1] = mse; }
  }
  if (i < n) {
    const float bc1 = (float)(1.0 - hcg_powi((double)b1, t));
    const float bc2_sqrt = (float)sqrt(1.0 - hcg_powi((double)b2, t));
    const float gi = g[i] * gs;
    if (SSE) g[i] = gi;
    float mi = m[i], vi = v[i];
    p[i] = hcg_adam_update(p[i], gi, mi, vi, b1, b2, eps, lr / bc1, bc2_sqrt);
    m[i] = mi;
    v[i] = vi;
  }
  __syncthreads();                                     // every thread of this block has read the step word
  if (threadIdx.x == 0) {
    const int ticket = __hip_atomic_fetch_add(&step_dev[1], 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if (ticket == (int)gridDim.x - 1) {
      __hip_atomic_store(&step_dev[1], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&step_dev[0], t, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

}  // namespace

extern "C" int hcg_adam_step_dev(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                                 const float* lr_dev, float beta1, float beta2, float eps, int32_t* step_dev,
                                 hcg_stream_t stream) {
  if (n <= 0 || !param || !grad || !exp_avg || !exp_avg_sq || !lr_dev || !step_dev) return HCG_ERR_INVALID_ARG;
  hipLaunchKernelGGL(k_adam_dev<false>, dim3((unsigned)hcg_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, param,
                     const_cast<float*>(grad), exp_avg, exp_avg_sq, n, lr_dev, beta1, beta2, eps, (int*)step_dev, nullptr);
  HCG_CHECK_LAUNCH();
  return HCG_OK;
}

extern "C" int hcg_adam_step_dev_sse(float* param, float* flat, float* exp_avg, float* exp_avg_sq, int64_t n,
                                     const float* lr_dev, float beta1, float beta2, float eps, int32_t* step_dev, float* loss,
                                     hcg_stream_t stream) {
  if (n <= 0 || !param || !flat || !exp_avg || !exp_avg_sq || !lr_dev || !step_dev || !loss) return HCG_ERR_INVALID_ARG;
  hipLaunchKernelGGL(k_adam_dev<true>, dim3((unsigned)hcg_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, param, flat,
                     exp_avg, exp_avg_sq, n, lr_dev, beta1, beta2, eps, (int*)step_dev, loss);
  HCG_CHECK_LAUNCH();
  return HCG_OK;
}

namespace {
__global__ __launch_bounds__(256) void k_sse_finalize(float* __restrict__ g, int64_t n, float* __restrict__ loss) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const float sse = g[n], cnt = g[n + 1], mse = sse / cnt, lv = sqrtf(mse);
  if (i == 0) { loss[0] = lv; loss[1] = mse; }
  if (i < n) g[i] *= 1.0f / (cnt * lv);
}
}  // namespace

extern "C" int hcg_sse_finalize(float* flat, int64_t n, float* loss, hcg_stream_t stream) {
  if (n <= 0 || !flat || !loss) return HCG_ERR_INVALID_ARG;
  hipLaunchKernelGGL(k_sse_finalize, dim3((unsigned)hcg_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, flat, n, loss);
  HCG_CHECK_LAUNCH();
  return HCG_OK;
}
