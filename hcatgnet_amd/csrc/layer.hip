// Any-shape GCN layer: dense transform on the matrix cores (gemm.hip) + CSR segmented sum.
// Replaces, per layer, PyG's gcn_norm + Linear + index_select + message + scatter_add_ + bias and
// the LeakyReLU that follows (reference call sites model/gcn.py:58-63), and
// their autograd (row a11).  The segmented sum walks each node's incoming row in the plan's fixed
// order: no atomics, bitwise reproducible.
#include "common.h"

int hcg_colsum_masked(const float* src, const float* mask, float slope, float* out, int64_t M, int64_t D,
                      float* partials, hipStream_t stream);

namespace {

// out_i = epi( dinv_i * sum_k w_k * dinv_{c_k} * T(src)_{c_k}  +  fill * dinv_i^2 * T(src)_i )
//   forward : T = identity on h,           epi = LeakyReLU(. + bias)
//   backward: T = dA * leaky'(A) (dY),     epi = identity            (rows = the transpose CSC)
// VEC = 4: D % 4 == 0, each lane owns float4 feature groups; VEC = 1: any D.
template <int VEC, bool BWD>
__global__ __launch_bounds__(256) void k_aggregate(const float* __restrict__ src, const float* __restrict__ mask_src,
                                                   const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                   const float* __restrict__ ew, const float* __restrict__ dinv,
                                                   const float* __restrict__ bias, float fill, float slope, int apply_act,
                                                   float* __restrict__ out, int64_t N, int D, int lanes_per_row) {
  const int tid = threadIdx.x;
  const int rows_per_block = 256 / lanes_per_row;
  const int64_t i = (int64_t)blockIdx.x * rows_per_block + tid / lanes_per_row;
  if (i >= N) return;
  const int lr = tid % lanes_per_row;
  const int32_t kb = rowptr[i], ke = rowptr[i + 1];
  const float di = dinv[i];
  const int groups = D / VEC;
  for (int gidx = lr; gidx < groups; gidx += lanes_per_row) {
    const int f = gidx * VEC;
    float acc[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) acc[v] = 0.f;
    for (int32_t k = kb; k < ke; ++k) {
      const int32_t c = col[k];
      if (c < 0) continue;   // explicit self loop: collapsed into the unit self loop below
      float w = dinv[c];
      if (ew) w *= ew[k];
      float val[VEC];
      if (VEC == 4) {
        const float4 t = *reinterpret_cast<const float4*>(src + (size_t)c * D + f);
        val[0] = t.x; val[1 % VEC] = t.y; val[2 % VEC] = t.z; val[3 % VEC] = t.w;
        if (BWD && apply_act) {
          const float4 m = *reinterpret_cast<const float4*>(mask_src + (size_t)c * D + f);
          val[0] *= hcg_leaky_grad(m.x, slope); val[1 % VEC] *= hcg_leaky_grad(m.y, slope);
          val[2 % VEC] *= hcg_leaky_grad(m.z, slope); val[3 % VEC] *= hcg_leaky_grad(m.w, slope);
        }
      } else {
        val[0] = src[(size_t)c * D + f];
        if (BWD && apply_act) val[0] *= hcg_leaky_grad(mask_src[(size_t)c * D + f], slope);
      }
#pragma unroll
      for (int v = 0; v < VEC; ++v) acc[v] += w * val[v];
    }
    // self loop, weight `fill`
    float self[VEC];
    if (VEC == 4) {
      const float4 t = *reinterpret_cast<const float4*>(src + (size_t)i * D + f);
      self[0] = t.x; self[1 % VEC] = t.y; self[2 % VEC] = t.z; self[3 % VEC] = t.w;
      if (BWD && apply_act) {
        const float4 m = *reinterpret_cast<const float4*>(mask_src + (size_t)i * D + f);
        self[0] *= hcg_leaky_grad(m.x, slope); self[1 % VEC] *= hcg_leaky_grad(m.y, slope);
        self[2 % VEC] *= hcg_leaky_grad(m.z, slope); self[3 % VEC] *= hcg_leaky_grad(m.w, slope);
      }
    } else {
      self[0] = src[(size_t)i * D + f];
      if (BWD && apply_act) self[0] *= hcg_leaky_grad(mask_src[(size_t)i * D + f], slope);
    }
    float res[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      float y = di * acc[v] + (fill * di * di) * self[v];
      if (!BWD) {
        y += bias[f + v];
        if (apply_act) y = hcg_leaky(y, slope);
      }
      res[v] = y;
    }
    if (VEC == 4) {
      *reinterpret_cast<float4*>(out + (size_t)i * D + f) = make_float4(res[0], res[1 % VEC], res[2 % VEC], res[3 % VEC]);
    } else {
      out[(size_t)i * D + f] = res[0];
    }
  }
}

template <bool BWD>
int launch_aggregate(const float* src, const float* mask_src, const int32_t* rowptr, const int32_t* col,
                     const float* ew, const float* dinv, const float* bias, float fill, float slope, int apply_act,
                     float* out, int64_t N, int64_t D, hipStream_t stream) {
  if (N <= 0 || D <= 0) return HCG_OK;
  const bool vec4 = (D % 4 == 0) && (((uintptr_t)src | (uintptr_t)out | (uintptr_t)mask_src) % 16 == 0);
  const int groups = (int)(vec4 ? D / 4 : D);
  int lpr = 1;
  while (lpr < groups && lpr < 64) lpr <<= 1;
  const int rows_per_block = 256 / lpr;
  const unsigned grid = (unsigned)hcg_cdiv(N, rows_per_block);
  if (vec4)
    hipLaunchKernelGGL((k_aggregate<4, BWD>), dim3(grid), dim3(256), 0, stream, src, mask_src, rowptr, col, ew, dinv,
                       bias, fill, slope, apply_act, out, N, (int)D, lpr);
  else
    hipLaunchKernelGGL((k_aggregate<1, BWD>), dim3(grid), dim3(256), 0, stream, src, mask_src, rowptr, col, ew, dinv,
                       bias, fill, slope, apply_act, out, N, (int)D, lpr);
  HCG_CHECK_LAUNCH();
  return HCG_OK;
}

// Gradient of a layer's output w.r.t. the per-edge multipliers `ew` of k_aggregate (explain mode):
//   dew[k] = dinv_i * dinv_{col k} * < dY_i , h_{col k} >,   dY = dout * leaky'(out),   i = row of entry k.
// 16 lanes per destination row walk its entries; each entry's D-term dot product is reduced inside the 16-lane
// group in a fixed order.
__global__ __launch_bounds__(256) void k_edge_weight_grad(const float* __restrict__ dout, const float* __restrict__ out,
                                                          const float* __restrict__ h, const int32_t* __restrict__ rowptr,
                                                          const int32_t* __restrict__ col, const float* __restrict__ dinv,
                                                          float slope, int apply_act, float* __restrict__ dew, int64_t N,
                                                          int D) {
  const int64_t i = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
  if (i >= N) return;                       // whole 16-lane groups leave together
  const int lr = threadIdx.x & 15;
  const int32_t kb = rowptr[i], ke = rowptr[i + 1];
  const float di = dinv[i];
  for (int32_t k = kb; k < ke; ++k) {
    const int32_t c = col[k];
    float s = 0.f;
    if (c >= 0) {
      for (int f = lr; f < D; f += 16) {
        float g = dout[(size_t)i * D + f];
        if (apply_act) g *= hcg_leaky_grad(out[(size_t)i * D + f], slope);
        s += g * h[(size_t)c * D + f];
      }
    }
    s += __shfl_xor(s, 1, 16);
    s += __shfl_xor(s, 2, 16);
    s += __shfl_xor(s, 4, 16);
    s += __shfl_xor(s, 8, 16);
    if (lr == 0) dew[k] = c >= 0 ? di * dinv[c] * s : 0.f;
  }
}

__global__ __launch_bounds__(256) void k_act_bwd(const float* __restrict__ dy, const float* __restrict__ y,
                                                 float* __restrict__ dz, int64_t n, int act, float slope) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float g = dy[i];
  if (act) g *= hcg_leaky_grad(y[i], slope);
  dz[i] = g;
}

}  // namespace

// ------------------------------------------------------------------------------ dense linear
static size_t linear_workspace_bytes(int64_t M, int64_t D_in, int64_t D_out) {
  size_t f = hcg_gemm_partial_floats(D_out, D_in, M, nullptr) + hcg_colsum_partial_floats(M, D_out);
  return hcg_align_up(f * sizeof(float), 256) + 512;
}

extern "C" int hcg_linear_fwd(const float* x, const float* W, const float* b, float* y, int64_t M, int64_t D_in,
                              int64_t D_out, int act, float slope, hcg_stream_t stream) {
  if (M < 0 || D_in < 0 || D_out <= 0 || !W || !y || (M > 0 && D_in > 0 && !x)) return HCG_ERR_INVALID_ARG;
  // y[m, o] = sum_k x[m, k] * W[o, k]
  return hcg_gemm(x, D_in, 1, W, 1, D_in, y, M, D_out, D_in, b, act, slope, nullptr, 0, (hipStream_t)stream);
}

extern "C" int hcg_linear_bwd(const float* dy, const float* y, const float* x, const float* W, float* dx, float* dW,
                              float* db, float* dz_ws, int64_t M, int64_t D_in, int64_t D_out, int act, float slope,
                              void* workspace, size_t workspace_bytes, hcg_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (M < 0 || D_in <= 0 || D_out <= 0 || !W || !dW) return HCG_ERR_INVALID_ARG;
  if (M > 0 && (!dy || !y || !x || !dz_ws)) return HCG_ERR_INVALID_ARG;
  HcgArena arena(workspace, workspace_bytes);
  int splits = 1;
  const size_t pf = hcg_gemm_partial_floats(D_out, D_in, M, &splits);
  float* partials = pf ? arena.take<float>(pf) : nullptr;
  float* cs = db ? arena.take<float>(hcg_colsum_partial_floats(M, D_out)) : nullptr;
  if ((pf && !partials) || (db && !cs)) return HCG_ERR_WORKSPACE;
  if (M > 0) {
    hipLaunchKernelGGL(k_act_bwd, dim3((unsigned)hcg_cdiv(M * D_out, 256)), dim3(256), 0, stream, dy, y, dz_ws,
                       M * D_out, act, slope);
    HCG_CHECK_LAUNCH();
  }
  // dW[o, k] = sum_m dz[m, o] x[m, k]
  HCG_TRY(hcg_gemm(dz_ws, 1, D_out, x, D_in, 1, dW, D_out, D_in, M, nullptr, 0, 0.f, partials, pf, stream));
  if (db) HCG_TRY(hcg_colsum(dz_ws, db, M, D_out, cs, stream));
  // dx[m, k] = sum_o dz[m, o] W[o, k]
  if (dx) HCG_TRY(hcg_gemm(dz_ws, D_out, 1, W, D_in, 1, dx, M, D_in, D_out, nullptr, 0, 0.f, nullptr, 0, stream));
  return HCG_OK;
}

// ------------------------------------------------------------------------------ GCN layer
extern "C" int hcg_gcn_layer_fwd(const float* x, const float* W, const float* b, const int32_t* rowptr,
                                 const int32_t* col, const float* ew_csr, const float* dinv, float fill, float slope,
                                 int apply_act, float* h_ws, float* out, int64_t N, int64_t E, int64_t F, int64_t D,
                                 hcg_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (N < 0 || E < 0 || F <= 0 || D <= 0 || !W || !b) return HCG_ERR_INVALID_ARG;
  if (N == 0) return HCG_OK;
  if (!x || !rowptr || !dinv || !h_ws || !out || (E > 0 && !col)) return HCG_ERR_INVALID_ARG;
  HCG_TRY(hcg_gemm(x, F, 1, W, 1, F, h_ws, N, D, F, nullptr, 0, 0.f, nullptr, 0, stream));  // a4: h = x W^T
  return launch_aggregate<false>(h_ws, nullptr, rowptr, col, ew_csr, dinv, b, fill, slope, apply_act, out, N, D,
                                 stream);                                                      // a5-a8
}

// workspace sizes of the any-shape entry points, one query (a, b, c as the kind names them)
size_t hcg_plan_workspace_bytes_impl(int64_t N, int64_t E, int64_t B, int mode);   // plan.hip
size_t hcg_readout2_workspace_bytes_impl(int64_t B);                               // readout.hip
extern "C" size_t hcg_general_workspace_bytes(int kind, int64_t a, int64_t b, int64_t c, int mode) {
  switch (kind) {
    case HCG_WS_PLAN: return hcg_plan_workspace_bytes_impl(a, b, c, mode);
    case HCG_WS_LINEAR: return linear_workspace_bytes(a, b, c);           // M, D_in, D_out
    case HCG_WS_GCN_LAYER_BWD: return linear_workspace_bytes(a, b, c);    // N, F, D: dW = dH^T X split-K + the bias column sum
    case HCG_WS_READOUT2: return hcg_readout2_workspace_bytes_impl(a);    // B
    default: return 0;
  }
}

extern "C" int hcg_gcn_layer_bwd(const float* dout, const float* out, const float* x, const float* W,
                                 const int32_t* rowptr_t, const int32_t* col_t, const float* ew_csc, const float* dinv,
                                 float fill, float slope, int apply_act, float* dh_ws, float* dx, float* dW, float* db,
                                 int64_t N, int64_t E, int64_t F, int64_t D, void* workspace, size_t workspace_bytes,
                                 hcg_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (N < 0 || E < 0 || F <= 0 || D <= 0 || !W || !dW || !db) return HCG_ERR_INVALID_ARG;
  if (N > 0 && (!dout || !out || !x || !rowptr_t || !dinv || !dh_ws || (E > 0 && !col_t))) return HCG_ERR_INVALID_ARG;
  HcgArena arena(workspace, workspace_bytes);
  const size_t pf = hcg_gemm_partial_floats(D, F, N, nullptr);
  float* partials = pf ? arena.take<float>(pf) : nullptr;
  float* cs = arena.take<float>(hcg_colsum_partial_floats(N, D));
  if ((pf && !partials) || !cs) return HCG_ERR_WORKSPACE;
  // db = sum_i dY_i, dY = dout * leaky'(out)
  HCG_TRY(hcg_colsum_masked(dout, apply_act ? out : nullptr, slope, db, N, D, cs, stream));
  // dH = Ahat^T dY  (transpose rows, same fixed-order segmented sum)
  HCG_TRY(launch_aggregate<true>(dout, out, rowptr_t, col_t, ew_csc, dinv, nullptr, fill, slope, apply_act, dh_ws, N,
                                 D, stream));
  // dW[d, f] = sum_i dH[i, d] x[i, f]
  HCG_TRY(hcg_gemm(dh_ws, 1, D, x, F, 1, dW, D, F, N, nullptr, 0, 0.f, partials, pf, stream));
  // dx[i, f] = sum_d dH[i, d] W[d, f]
  if (dx && N > 0) HCG_TRY(hcg_gemm(dh_ws, D, 1, W, F, 1, dx, N, F, D, nullptr, 0, 0.f, nullptr, 0, stream));
  return HCG_OK;
}

// d loss / d ew_csr of hcg_gcn_layer_fwd (explain mode): `h` = x W^T (recompute it with hcg_linear_fwd), `out` = the
// layer's saved output, `dout` the upstream gradient; entries of explicit self-loop edges (col < 0) get 0.
extern "C" int hcg_gcn_edge_weight_grad(const float* dout, const float* out, const float* h, const int32_t* rowptr,
                                        const int32_t* col, const float* dinv, float slope, int apply_act, float* dew_csr,
                                        int64_t N, int64_t E, int64_t D, hcg_stream_t stream) {
  if (N < 0 || E < 0 || D <= 0) return HCG_ERR_INVALID_ARG;
  if (N == 0 || E == 0) return HCG_OK;
  if (!dout || !out || !h || !rowptr || !col || !dinv || !dew_csr) return HCG_ERR_INVALID_ARG;
  hipLaunchKernelGGL(k_edge_weight_grad, dim3((unsigned)hcg_cdiv(N, 16)), dim3(256), 0, (hipStream_t)stream, dout, out, h,
                     rowptr, col, dinv, slope, apply_act, dew_csr, N, (int)D);
  HCG_CHECK_LAUNCH();
  return HCG_OK;
}
