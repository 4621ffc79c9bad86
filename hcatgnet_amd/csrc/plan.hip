// Batch plan: everything `gcn_norm` + the scatter indices of one batch need, built ONCE per batch
// (the reference recomputes gcn_norm in every layer of every step: PyG GCNConv(cached=False),
// call sites model/gcn.py:58,62):
//   graph_ptr  <- batch               (node range per graph; replaces the `batch` vector of a9)
//   CSR by target (rowptr/col/eid) and its transpose (rowptr_t/col_t/eid_t), both STABLE in the
//   input edge order, so every per-node sum has one fixed order -> bitwise reproducible results;
//   dinv = (fill + in-degree)^-1/2.
// Two modes:
//   GENERAL  any edge order / any graph size: 64-bit (node<<32 | edge) keys through hipCUB's
//            device radix sort (stable by construction), rowptr by binary search.
//   BLOCKED  edges grouped by graph as PyG-style collation emits them (reference data/rhcaa.py:66-67
//            + Batch collate): one wavefront per graph, edges staged in LDS, lane-per-node stable
//            counting sort -- no device-wide sort, no atomics.
#include "common.h"
#include "ptrs.h"
#include <hipcub/hipcub.hpp>

namespace {

// ------------------------------------------------------------------ shared: graph_ptr from batch
__global__ __launch_bounds__(256) void k_graph_ptr(const int64_t* __restrict__ batch, int64_t N, int64_t B,
                                                   int32_t* __restrict__ graph_ptr, int32_t* __restrict__ status) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i > N) return;
  int64_t prev = (i > 0) ? batch[i - 1] : -1;
  int64_t cur = (i < N) ? batch[i] : B;
  int st = 0;
  if (i < N && (cur < 0 || cur >= B)) { st |= HCG_STATUS_BATCH_RANGE; }
  if (i > 0 && i < N && cur < prev) { st |= HCG_STATUS_BATCH_UNSORTED; }
  if (st) atomicOr(status, st);
  if (prev < -1) prev = -1;
  if (prev > B) prev = B;
  if (cur < 0) cur = 0;
  if (cur > B) cur = B;
  for (int64_t g = prev + 1; g <= cur; ++g) graph_ptr[g] = (int32_t)i;  // first node with batch >= g
}

// ------------------------------------------------------------------ GENERAL mode
__global__ __launch_bounds__(256) void k_make_keys(const int64_t* __restrict__ ei, int64_t N, int64_t E,
                                                   uint64_t* __restrict__ key_dst, uint64_t* __restrict__ key_src,
                                                   int32_t* __restrict__ status) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  int64_t s = ei[e], d = ei[E + e];
  if (s < 0 || s >= N || d < 0 || d >= N) {
    atomicOr(status, HCG_STATUS_INDEX_RANGE);
    s = 0; d = 0;  // keep every later access in bounds; the host raises on the status bit
  }
  key_dst[e] = ((uint64_t)d << 32) | (uint64_t)e;
  key_src[e] = ((uint64_t)s << 32) | (uint64_t)e;
}

// sorted keys -> col (the OTHER endpoint) + eid; `other_row` = row of edge_index to read
__global__ __launch_bounds__(256) void k_extract(const uint64_t* __restrict__ keys, const int64_t* __restrict__ other_row,
                                                 int64_t N, int64_t E, int32_t* __restrict__ col, int32_t* __restrict__ eid,
                                                 int32_t* __restrict__ status) {
  const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= E) return;
  const uint32_t e = (uint32_t)(keys[k] & 0xffffffffu);
  int64_t o = other_row[e];
  if (o < 0 || o >= N) o = 0;
  // explicit self loop (i, i): kept in its slot as -1 = "skip" (PyG add_remaining_self_loops collapses it
  // into the unit self loop every node gets)
  if ((int64_t)(keys[k] >> 32) == o) {
    o = -1;
    // with explicit edge weights PyG would use this edge's weight as the node's loop weight: not implemented
    if (eid) atomicOr(status, HCG_STATUS_WEIGHTED_SELF_LOOP);
  }
  col[k] = (int32_t)o;
  if (eid) eid[k] = (int32_t)e;
}

__global__ __launch_bounds__(256) void k_rowptr_search(const uint64_t* __restrict__ keys, int64_t N, int64_t E,
                                                       int32_t* __restrict__ rowptr) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i > N) return;
  const uint64_t target = (uint64_t)i << 32;  // first key with node >= i
  int64_t lo = 0, hi = E;
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (keys[mid] < target) lo = mid + 1; else hi = mid;
  }
  rowptr[i] = (int32_t)lo;
}

__global__ __launch_bounds__(256) void k_permute_weights(const float* __restrict__ ew, const int32_t* __restrict__ eid,
                                                         int64_t E, float* __restrict__ out) {
  const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k < E) out[k] = ew[eid[k]];
}

// dinv[i] = (fill + sum_k w_k)^-1/2 over the incoming row (fixed CSR order), 0 where degree == 0
__global__ __launch_bounds__(256) void k_dinv(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                              const float* __restrict__ ew_csr, int64_t N, float fill,
                                              float* __restrict__ dinv) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const int32_t b = rowptr[i], e = rowptr[i + 1];
  float deg = 0.f;
  for (int32_t k = b; k < e; ++k)
    if (col[k] >= 0) deg += ew_csr ? ew_csr[k] : 1.0f;   // col < 0: explicit self loop, not an edge
  deg += fill;
  dinv[i] = deg > 0.f ? 1.0f / sqrtf(deg) : 0.f;
}

// ------------------------------------------------------------------ BLOCKED mode
// blocked mode: graph_ptr and edge_ptr in ONE launch (threads [0, N] walk nodes, [N+1, N+E+1] edges)
__global__ __launch_bounds__(256) void k_ptrs(const int64_t* __restrict__ ei, const int64_t* __restrict__ batch,
                                              int64_t N, int64_t E, int64_t B, int32_t* __restrict__ graph_ptr,
                                              int32_t* __restrict__ edge_ptr, int32_t* __restrict__ status) {
  hcg_ptrs_thread((int64_t)blockIdx.x * blockDim.x + threadIdx.x, ei, batch, N, E, B, graph_ptr, edge_ptr, status);
}

constexpr int GE_LDS_EDGES = 2048;  // edges of one graph staged in LDS (16 KiB); larger graphs re-read global

// One wavefront per graph.  Lane i of a 64-node chunk owns node (chunk + i): it walks the graph's
// edge list IN INPUT ORDER and appends the edges that end (CSR) / start (CSC) at its node, so rows
// come out stable without any sort or atomic.  Row starts come from a wave-wide exclusive scan.
__global__ __launch_bounds__(64) void k_graph_csr(const int64_t* __restrict__ ei, int64_t E, int64_t B,
                                                  const int32_t* __restrict__ graph_ptr, const int32_t* __restrict__ edge_ptr,
                                                  float fill, int32_t* __restrict__ rowptr, int32_t* __restrict__ col,
                                                  int32_t* __restrict__ eid, int32_t* __restrict__ rowptr_t,
                                                  int32_t* __restrict__ col_t, int32_t* __restrict__ eid_t,
                                                  float* __restrict__ dinv, int32_t* __restrict__ status) {
  __shared__ int2 sedge[GE_LDS_EDGES];
  const int g = blockIdx.x, lane = threadIdx.x;
  const int nbeg = graph_ptr[g], nend = graph_ptr[g + 1];
  const int ebeg = edge_ptr[g], eend = edge_ptr[g + 1];
  const int n = nend - nbeg, ne = eend - ebeg;
  if (g == 0 && lane == 0) { rowptr[graph_ptr[B]] = (int32_t)E; rowptr_t[graph_ptr[B]] = (int32_t)E; }
  if (n <= 0) return;
  const bool in_lds = ne <= GE_LDS_EDGES;
  int bad = 0;
  if (in_lds) {
    for (int k = lane; k < ne; k += 64) {
      int64_t s = ei[ebeg + k] - nbeg, d = ei[E + ebeg + k] - nbeg;
      if (s < 0 || s >= n || d < 0 || d >= n) { bad = 1; s = 0; d = 0; }
      sedge[k] = make_int2((int)s, (int)d);
    }
    __syncthreads();
  }
  if (bad) atomicOr(status, HCG_STATUS_EDGE_UNGROUPED);

  int base_in = ebeg, base_out = ebeg;  // running row starts (wave-uniform)
  for (int c0 = 0; c0 < n; c0 += 64) {
    const int i = c0 + lane;
    const bool act = i < n;
    int cin = 0, cout = 0, nself = 0;   // nself: explicit (i, i) edges -- slots kept (col = -1), not counted in the degree
    if (in_lds) {
      for (int k = 0; k < ne; ++k) { const int2 sd = sedge[k]; cin += (sd.y == i); cout += (sd.x == i); nself += (sd.y == i && sd.x == i); }
    } else {
      for (int k = 0; k < ne; ++k) {
        const int s = (int)(ei[ebeg + k] - nbeg), d = (int)(ei[E + ebeg + k] - nbeg);
        cin += (d == i); cout += (s == i); nself += (d == i && s == i);
      }
    }
    if (!act) { cin = 0; cout = 0; nself = 0; }
    // wave-wide inclusive scan of (cin, cout)
    int sin = cin, sout = cout;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int a = __shfl_up(sin, off, 64), b = __shfl_up(sout, off, 64);
      if (lane >= off) { sin += a; sout += b; }
    }
    int pin = base_in + sin - cin, pout = base_out + sout - cout;
    if (act) {
      rowptr[nbeg + i] = pin;
      rowptr_t[nbeg + i] = pout;
      const float deg = fill + (float)(cin - nself);
      dinv[nbeg + i] = deg > 0.f ? 1.0f / sqrtf(deg) : 0.f;
      if (in_lds) {
        for (int k = 0; k < ne; ++k) {
          const int2 sd = sedge[k];
          if (sd.y == i) {
            col[pin] = sd.x == i ? -1 : nbeg + sd.x;
            if (eid) { eid[pin] = ebeg + k; if (sd.x == i) atomicOr(status, HCG_STATUS_WEIGHTED_SELF_LOOP); }
            ++pin;
          }
          if (sd.x == i) { col_t[pout] = sd.y == i ? -1 : nbeg + sd.y; if (eid_t) eid_t[pout] = ebeg + k; ++pout; }
        }
      } else {
        for (int k = 0; k < ne; ++k) {
          int s = (int)(ei[ebeg + k] - nbeg), d = (int)(ei[E + ebeg + k] - nbeg);
          if (s < 0 || s >= n || d < 0 || d >= n) { s = 0; d = 0; atomicOr(status, HCG_STATUS_EDGE_UNGROUPED); }
          if (d == i) {
            col[pin] = s == i ? -1 : nbeg + s;
            if (eid) { eid[pin] = ebeg + k; if (s == i) atomicOr(status, HCG_STATUS_WEIGHTED_SELF_LOOP); }
            ++pin;
          }
          if (s == i) { col_t[pout] = d == i ? -1 : nbeg + d; if (eid_t) eid_t[pout] = ebeg + k; ++pout; }
        }
      }
    }
    base_in += __shfl(sin, 63, 64);
    base_out += __shfl(sout, 63, 64);
  }
}

// weighted degree needs the permuted weights first -> separate pass (k_dinv) in blocked mode too

size_t sort_temp_bytes(int64_t E) {
  size_t bytes = 0;
  (void)hipcub::DeviceRadixSort::SortKeys(nullptr, bytes, (const uint64_t*)nullptr, (uint64_t*)nullptr, (int)E, 0, 64,
                                    (hipStream_t)0);
  return bytes;
}

int bits_for(int64_t v) { int b = 1; while (((int64_t)1 << b) <= v && b < 31) ++b; return b; }

}  // namespace

size_t hcg_plan_workspace_bytes_impl(int64_t N, int64_t E, int64_t B, int mode) {
  (void)N; (void)B;
  if ((mode & ~(HCG_PLAN_PTRS_ONLY | HCG_PLAN_KEEP_STATUS)) == HCG_PLAN_BLOCKED || E <= 0) return 256;
  const size_t keys = hcg_align_up((size_t)E * sizeof(uint64_t), 256);
  return 3 * keys + hcg_align_up(sort_temp_bytes(E), 256) + 1024;
}

extern "C" int hcg_plan_build(const int64_t* edge_index, const int64_t* batch, const float* edge_weight, int64_t N,
                              int64_t E, int64_t B, float fill, int mode, int32_t* graph_ptr, int32_t* edge_ptr,
                              int32_t* rowptr, int32_t* col, int32_t* eid, int32_t* rowptr_t, int32_t* col_t,
                              int32_t* eid_t, float* dinv, float* ew_csr, float* ew_csc, float* dinv_unw,
                              int32_t* status, void* workspace, size_t workspace_bytes, hcg_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (N < 0 || E < 0 || B < 0 || N >= ((int64_t)1 << 31) || E >= ((int64_t)1 << 31) || B >= ((int64_t)1 << 31))
    return HCG_ERR_INVALID_ARG;
  const bool want_csr = (mode & HCG_PLAN_PTRS_ONLY) == 0;
  if (!graph_ptr || !status) return HCG_ERR_INVALID_ARG;
  if (want_csr && (!rowptr || !rowptr_t || !dinv)) return HCG_ERR_INVALID_ARG;
  if (E > 0 && (!edge_index || (want_csr && (!col || !col_t)))) return HCG_ERR_INVALID_ARG;
  if (N > 0 && !batch) return HCG_ERR_INVALID_ARG;
  if (edge_weight && (!eid || !eid_t || !ew_csr || !ew_csc || !dinv_unw)) return HCG_ERR_INVALID_ARG;
  const bool ptrs_only = (mode & HCG_PLAN_PTRS_ONLY) != 0;
  const bool keep_status = (mode & HCG_PLAN_KEEP_STATUS) != 0;
  mode &= ~(HCG_PLAN_PTRS_ONLY | HCG_PLAN_KEEP_STATUS);
  if (mode != HCG_PLAN_GENERAL && mode != HCG_PLAN_BLOCKED) return HCG_ERR_INVALID_ARG;
  if (mode == HCG_PLAN_BLOCKED && !edge_ptr) return HCG_ERR_INVALID_ARG;
  if (ptrs_only && mode != HCG_PLAN_BLOCKED) return HCG_ERR_INVALID_ARG;

  if (!keep_status) HCG_TRY(hcg_hip_err(hipMemsetAsync(status, 0, 4 * sizeof(int32_t), stream)));
  if (mode == HCG_PLAN_BLOCKED) {
    hipLaunchKernelGGL(k_ptrs, dim3((unsigned)hcg_cdiv(N + E + 2, 256)), dim3(256), 0, stream, edge_index, batch, N, E,
                       B, graph_ptr, edge_ptr, status);
    HCG_CHECK_LAUNCH();
    if (ptrs_only) return HCG_OK;   // the fused kernels rebuild gcn_norm on chip from the raw edges
  } else {
    hipLaunchKernelGGL(k_graph_ptr, dim3((unsigned)hcg_cdiv(N + 1, 256)), dim3(256), 0, stream, batch, N, B, graph_ptr,
                       status);
    HCG_CHECK_LAUNCH();
  }

  if (mode == HCG_PLAN_BLOCKED) {
    if (B > 0) {
      hipLaunchKernelGGL(k_graph_csr, dim3((unsigned)B), dim3(64), 0, stream, edge_index, E, B,
                         (const int32_t*)graph_ptr, (const int32_t*)edge_ptr, fill, rowptr, col, eid, rowptr_t, col_t,
                         eid_t, dinv, status);
      HCG_CHECK_LAUNCH();
    } else {
      HCG_TRY(hcg_hip_err(hipMemsetAsync(rowptr, 0, sizeof(int32_t), stream)));
      HCG_TRY(hcg_hip_err(hipMemsetAsync(rowptr_t, 0, sizeof(int32_t), stream)));
    }
  } else {
    if (E > 0) {
      HcgArena arena(workspace, workspace_bytes);
      uint64_t* k0 = arena.take<uint64_t>((size_t)E);
      uint64_t* k1 = arena.take<uint64_t>((size_t)E);
      uint64_t* k2 = arena.take<uint64_t>((size_t)E);
      size_t temp_bytes = sort_temp_bytes(E);
      char* temp = arena.take<char>(temp_bytes);
      if (!k0 || !k1 || !k2 || !temp) return HCG_ERR_WORKSPACE;
      const unsigned gE = (unsigned)hcg_cdiv(E, 256), gN = (unsigned)hcg_cdiv(N + 1, 256);
      hipLaunchKernelGGL(k_make_keys, dim3(gE), dim3(256), 0, stream, edge_index, N, E, k0, k1, status);
      HCG_CHECK_LAUNCH();
      const int end_bit = 32 + bits_for(N);
      // CSR by target
      HCG_TRY(hcg_hip_err(hipcub::DeviceRadixSort::SortKeys(temp, temp_bytes, (const uint64_t*)k0, k2, (int)E, 0,
                                                             end_bit, stream)));
      hipLaunchKernelGGL(k_extract, dim3(gE), dim3(256), 0, stream, (const uint64_t*)k2, edge_index, N, E, col, eid, status);
      HCG_CHECK_LAUNCH();
      hipLaunchKernelGGL(k_rowptr_search, dim3(gN), dim3(256), 0, stream, (const uint64_t*)k2, N, E, rowptr);
      HCG_CHECK_LAUNCH();
      // CSC (by source)
      HCG_TRY(hcg_hip_err(hipcub::DeviceRadixSort::SortKeys(temp, temp_bytes, (const uint64_t*)k1, k2, (int)E, 0,
                                                             end_bit, stream)));
      hipLaunchKernelGGL(k_extract, dim3(gE), dim3(256), 0, stream, (const uint64_t*)k2, edge_index + E, N, E, col_t,
                         eid_t, status);
      HCG_CHECK_LAUNCH();
      hipLaunchKernelGGL(k_rowptr_search, dim3(gN), dim3(256), 0, stream, (const uint64_t*)k2, N, E, rowptr_t);
      HCG_CHECK_LAUNCH();
    } else {
      HCG_TRY(hcg_hip_err(hipMemsetAsync(rowptr, 0, (size_t)(N + 1) * sizeof(int32_t), stream)));
      HCG_TRY(hcg_hip_err(hipMemsetAsync(rowptr_t, 0, (size_t)(N + 1) * sizeof(int32_t), stream)));
    }
  }

  if (edge_weight && E > 0) {
    const unsigned gE = (unsigned)hcg_cdiv(E, 256);
    hipLaunchKernelGGL(k_permute_weights, dim3(gE), dim3(256), 0, stream, edge_weight, (const int32_t*)eid, E, ew_csr);
    HCG_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_permute_weights, dim3(gE), dim3(256), 0, stream, edge_weight, (const int32_t*)eid_t, E,
                       ew_csc);
    HCG_CHECK_LAUNCH();
  }
  if (N > 0 && (mode == HCG_PLAN_GENERAL || edge_weight)) {
    hipLaunchKernelGGL(k_dinv, dim3((unsigned)hcg_cdiv(N, 256)), dim3(256), 0, stream, (const int32_t*)rowptr,
                       (const int32_t*)col, (const float*)(edge_weight ? ew_csr : nullptr), N, fill, dinv);
    HCG_CHECK_LAUNCH();
  }
  if (N > 0 && edge_weight) {
    hipLaunchKernelGGL(k_dinv, dim3((unsigned)hcg_cdiv(N, 256)), dim3(256), 0, stream, (const int32_t*)rowptr,
                       (const int32_t*)col, (const float*)nullptr, N, 1.0f, dinv_unw);
    HCG_CHECK_LAUNCH();
  }
  return HCG_OK;
}
