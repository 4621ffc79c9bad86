// Graph pooling: emb[g] = [ max_{i in g} a_i , mean_{i in g} a_i ]  -- max FIRST,
// as `torch.cat([gmp(x, batch), gap(x, batch)], dim=1)` in the reference (model/gcn.py:65-66).
// Replaces PyG global_max_pool / global_mean_pool (two scatter-reduces + count + div + cat) with
// one segmented pass over graph_ptr; backward follows torch's `amax` rule: the max-branch gradient
// is split EVENLY among tied rows (chemically equivalent atoms give bit-identical rows, so ties
// are common -- SURVEY 7, hard parts).
#include "common.h"

namespace {

// One 256-thread block per graph: lane (tid & 63) walks features, (tid >> 6) walks rows mod 4.
__global__ __launch_bounds__(256) void k_pool_fwd(const float* __restrict__ a, const int32_t* __restrict__ graph_ptr,
                                                  float* __restrict__ emb, int D) {
  __shared__ float smax[4][64];
  __shared__ float ssum[4][64];
  const int g = blockIdx.x, fw = threadIdx.x & 63, rw = threadIdx.x >> 6;
  const int beg = graph_ptr[g], end = graph_ptr[g + 1];
  const int n = end - beg;
  const float cnt_n = (float)(n > 0 ? n : 1);
  for (int f0 = 0; f0 < D; f0 += 64) {
    const int f = f0 + fw;
    float mx = -INFINITY, sm = 0.f;
    if (f < D) {
      for (int r = beg + rw; r < end; r += 4) {
        const float v = a[(size_t)r * D + f];
        mx = fmaxf(mx, v);
        sm += v;
      }
    }
    smax[rw][fw] = mx;
    ssum[rw][fw] = sm;
    __syncthreads();
    if (rw == 0 && f < D) {
      float m = fmaxf(fmaxf(smax[0][fw], smax[1][fw]), fmaxf(smax[2][fw], smax[3][fw]));
      float s = ((ssum[0][fw] + ssum[1][fw]) + ssum[2][fw]) + ssum[3][fw];
      if (n <= 0) m = 0.f;  // empty graph slot: PyG/torch scatter leaves the zero initialiser
      emb[(size_t)g * 2 * D + f] = m;
      emb[(size_t)g * 2 * D + D + f] = s / cnt_n;  // true division: bit-equal to torch's sum / count
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void k_pool_bwd(const float* __restrict__ demb, const float* __restrict__ a,
                                                  const float* __restrict__ emb, const int32_t* __restrict__ graph_ptr,
                                                  float* __restrict__ da, int D) {
  __shared__ int scnt[4][64];
  const int g = blockIdx.x, fw = threadIdx.x & 63, rw = threadIdx.x >> 6;
  const int beg = graph_ptr[g], end = graph_ptr[g + 1];
  const int n = end - beg;
  if (n <= 0) return;
  const float cnt_n = (float)n;
  for (int f0 = 0; f0 < D; f0 += 64) {
    const int f = f0 + fw;
    float mx = 0.f, gmax = 0.f, gmean = 0.f;
    int cnt = 0;
    if (f < D) {
      mx = emb[(size_t)g * 2 * D + f];
      gmax = demb[(size_t)g * 2 * D + f];
      gmean = demb[(size_t)g * 2 * D + D + f] / cnt_n;
      for (int r = beg + rw; r < end; r += 4) cnt += (a[(size_t)r * D + f] == mx);
    }
    scnt[rw][fw] = cnt;
    __syncthreads();
    if (f < D) {
      const int ties = scnt[0][fw] + scnt[1][fw] + scnt[2][fw] + scnt[3][fw];
      const float share = gmax / (float)(ties > 0 ? ties : 1);
      for (int r = beg + rw; r < end; r += 4) {
        const float v = a[(size_t)r * D + f];
        da[(size_t)r * D + f] = gmean + (v == mx ? share : 0.f);
      }
    }
    __syncthreads();
  }
}

}  // namespace

extern "C" int hcg_pool_fwd(const float* a, const int32_t* graph_ptr, float* emb, int64_t N, int64_t B, int64_t D,
                            hcg_stream_t stream) {
  if (N < 0 || B < 0 || D <= 0 || D > (1 << 20)) return HCG_ERR_INVALID_ARG;
  if (B == 0) return HCG_OK;
  if (!graph_ptr || !emb || (N > 0 && !a)) return HCG_ERR_INVALID_ARG;
  hipLaunchKernelGGL(k_pool_fwd, dim3((unsigned)B), dim3(256), 0, (hipStream_t)stream, a, graph_ptr, emb, (int)D);
  HCG_CHECK_LAUNCH();
  return HCG_OK;
}

extern "C" int hcg_pool_bwd(const float* demb, const float* a, const float* emb, const int32_t* graph_ptr, float* da,
                            int64_t N, int64_t B, int64_t D, hcg_stream_t stream) {
  if (N < 0 || B < 0 || D <= 0 || D > (1 << 20)) return HCG_ERR_INVALID_ARG;
  if (B == 0 || N == 0) return HCG_OK;
  if (!graph_ptr || !emb || !demb || !a || !da) return HCG_ERR_INVALID_ARG;
  hipLaunchKernelGGL(k_pool_bwd, dim3((unsigned)B), dim3(256), 0, (hipStream_t)stream, demb, a, emb, graph_ptr, da,
                     (int)D);
  HCG_CHECK_LAUNCH();
  return HCG_OK;
}
