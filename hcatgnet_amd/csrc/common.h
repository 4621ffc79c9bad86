// Internal helpers shared by the HIP translation units of libhcatgnet_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "../../include/hcatgnet_hip.h"

#define HCG_WAVE 64

static inline int hcg_hip_err(hipError_t e) { return e == hipSuccess ? HCG_OK : (HCG_ERR_HIP_BASE - (int)e); }

// launch + return on error (kernel launch errors are sticky-free: read with hipGetLastError)
#define HCG_CHECK_LAUNCH()                                  \
  do {                                                      \
    hipError_t _e = hipGetLastError();                      \
    if (_e != hipSuccess) return hcg_hip_err(_e);           \
  } while (0)

#define HCG_TRY(expr)                 \
  do {                                \
    int _rc = (expr);                 \
    if (_rc != HCG_OK) return _rc;    \
  } while (0)

static inline size_t hcg_align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
static inline int64_t hcg_cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// bump allocator over the caller's workspace
struct HcgArena {
  char* base;
  size_t size, off;
  HcgArena(void* p, size_t n) : base((char*)p), size(n), off(0) {}
  template <typename T>
  T* take(size_t count) {
    size_t bytes = hcg_align_up(count * sizeof(T), 256);
    if (off + bytes > size) return nullptr;
    T* r = (T*)(base + off);
    off += bytes;
    return r;
  }
};

// torch.optim.Adam's update of one element (amsgrad / weight_decay / maximize off), shared by every kernel that applies it
// (optim.hip, reduce.hip) and written with explicit roundings: left to the compiler, the fused and the stand-alone update
// contracted these expressions into different fma's, and with eps = 1e-9 (model/networks.py:38) a last-bit difference in a
// moment of a near-zero gradient becomes an lr-sized difference of the parameter.
//   step_size = lr / (1 - b1^t), bc2_sqrt = sqrt(1 - b2^t)
__device__ __forceinline__ float hcg_adam_update(float p, float g, float& m, float& v, float b1, float b2, float eps,
                                                 float step_size, float bc2_sqrt) {
  m = __fmaf_rn(b1, m, __fmul_rn(1.0f - b1, g));
  v = __fmaf_rn(b2, v, __fmul_rn(__fmul_rn(1.0f - b2, g), g));
  const float denom = __fadd_rn(__fdiv_rn(__fsqrt_rn(v), bc2_sqrt), eps);
  return __fsub_rn(p, __fmul_rn(step_size, __fdiv_rn(m, denom)));
}

__device__ __forceinline__ float hcg_leaky(float v, float slope) { return v > 0.f ? v : v * slope; }
__device__ __forceinline__ float hcg_leaky_grad(float y_out, float slope) { return y_out > 0.f ? 1.f : slope; }

// ---- internal launchers implemented in gemm.hip -------------------------------------------
// C[M,N] = sum_k A(m,k) B(k,n), A(m,k) at A[m*sam + k*sak], B(k,n) at B[k*sbk + n*sbn];
// epilogue: + bias[n] (nullable), LeakyReLU when act.  splits > 1: deterministic split-K through
// `partials` ([splits, M, N] floats).
size_t hcg_gemm_partial_floats(int64_t M, int64_t N, int64_t K, int* splits_out);
int hcg_gemm(const float* A, int64_t sam, int64_t sak, const float* B, int64_t sbk, int64_t sbn,
             float* C, int64_t M, int64_t N, int64_t K, const float* bias, int act, float slope,
             float* partials, size_t partial_floats, hipStream_t stream);
// out[d] = sum_m src[m, d]  (deterministic two-stage); partials >= colsum_partial_floats
size_t hcg_colsum_partial_floats(int64_t M, int64_t D);
int hcg_colsum(const float* src, float* out, int64_t M, int64_t D, float* partials, hipStream_t stream);

// ---- one-graph-per-wave kernels (wave.hip), selected inside the hcg_mid_* entry points (mid.hip) ----------
int hcg_w64_applicable(int64_t F, int64_t D, int64_t max_nodes, int64_t max_edges);
int hcg_w64_bwd_grid(int64_t B);
int hcg_w64_fwd_launch(const float* x, const float* W, const float* b, const int64_t* edge_index, int64_t E,
                       const int32_t* graph_ptr, const int32_t* edge_ptr, int64_t B, int64_t F, float slope, int apply_act,
                       float* out, float* emb, int32_t* status, hipStream_t stream);
int hcg_w64_bwd_launch(const float* dout, const float* demb, const float* emb, const float* out, const float* x,
                       const float* W, const int64_t* edge_index, int64_t E, const int32_t* graph_ptr,
                       const int32_t* edge_ptr, int64_t B, int64_t F, float slope, int apply_act, float* dx,
                       float* partials, int32_t* status, hipStream_t stream);
