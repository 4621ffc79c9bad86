// On-device collation (the step immediately BEFORE the path).
// The reference builds every batch on the host: per-graph `torch.load` (data/datasets.py:74-78) and PyG's
// DataLoader collate (call_methods.py:41-46).  At >1e7 graphs/s that is three orders of magnitude too
// slow, and 288 GB of HBM hold any dataset this model family sees -- so the whole dataset lives in HBM
// once (features, local edge lists, per-graph offsets) and a batch is ONE gather launch:
//   x_out      rows of the selected graphs, concatenated in selection order
//   edge_index local ids + the graph's node offset in the batch (int64 [2, E], PyG collation rule)
//   batch      graph id in the batch per node (int64)
//   y / idx    per-graph scalars
// The batch's graph_ptr / edge_ptr (= the whole plan of the fused path) are prefix sums of per-graph sizes
// the host already knows, so they are uploaded, not recomputed.
// Every batch of an epoch in ONE launch (round 3): a shuffled epoch's batches are known once its permutation is drawn, and at
// the reference's batch size 40 a per-batch collate launch is a tenth of the step it feeds (40 workgroups, one dependent
// hop): train.EpochWindow hands the launch a device array of slot descriptors, grid = (largest batch, slots).
#include "common.h"

namespace {

struct Dataset {
  const float* x_all;
  const int32_t* src_all;
  const int32_t* dst_all;
  const int64_t* node_ptr_all;
  const int64_t* edge_ptr_all;
  const float* y_all;
  const int64_t* idx_all;
  int F;
};

// one 256-thread workgroup per selected graph (blockIdx.x) of a slot (MULTI: blockIdx.y, descriptor read from the device
// array -- a uniform address, scalar loads)
template <bool MULTI>
__global__ __launch_bounds__(256) void k_collate(Dataset ds, hcg_collate_slot one, const hcg_collate_slot* __restrict__ slots) {
  const hcg_collate_slot s = MULTI ? slots[blockIdx.y] : one;
  const int b = blockIdx.x, tid = threadIdx.x;
  if (MULTI && b >= (int)s.B) return;
  // (pointers that arrive through memory are generic to the compiler: without the address-space casts every access of the
  //  multi-slot form is a flat_* instruction -- tools/isa_lint.py)
#define GLOBAL_PTR(T, p) reinterpret_cast<__attribute__((address_space(1))) T*>(reinterpret_cast<uintptr_t>(p))
  auto* ids = GLOBAL_PTR(const int64_t, s.ids);
  auto* gp = GLOBAL_PTR(const int32_t, s.graph_ptr);
  auto* ep = GLOBAL_PTR(const int32_t, s.edge_ptr);
  auto* x_out = GLOBAL_PTR(float, s.x_out);
  auto* ei_out = GLOBAL_PTR(int64_t, s.edge_index_out);
  auto* batch_out = GLOBAL_PTR(int64_t, s.batch_out);
  auto* y_out = GLOBAL_PTR(float, s.y_out);
  auto* idx_out = GLOBAL_PTR(int64_t, s.idx_out);
#undef GLOBAL_PTR
  const int F = ds.F;
  const int64_t g = ids[b];
  const int64_t nin = ds.node_ptr_all[g], ein = ds.edge_ptr_all[g];
  const int nout = gp[b], n = gp[b + 1] - nout;
  const int eout = ep[b], ne = ep[b + 1] - eout;
  // features: the graph's rows are one contiguous block of n*F floats on both sides
  const float* xs = ds.x_all + (size_t)nin * F;
  auto* xd = x_out + (size_t)nout * F;
  const int total = n * F;
  if ((F & 3) == 0 && (((uintptr_t)xs | (uintptr_t)xd) & 15) == 0) {
    typedef float vec4 __attribute__((ext_vector_type(4)));
    auto* xd4 = reinterpret_cast<__attribute__((address_space(1))) vec4*>(xd);
    for (int i = tid; i < total / 4; i += 256) xd4[i] = reinterpret_cast<const vec4*>(xs)[i];
  } else {
    for (int i = tid; i < total; i += 256) xd[i] = xs[i];
  }
  for (int i = tid; i < n; i += 256) batch_out[nout + i] = b;
  for (int k = tid; k < ne; k += 256) {
    ei_out[eout + k] = (int64_t)ds.src_all[ein + k] + nout;
    ei_out[s.E_out + eout + k] = (int64_t)ds.dst_all[ein + k] + nout;
  }
  if (tid == 0) {
    if (s.y_out) y_out[b] = ds.y_all[g];
    if (s.idx_out) idx_out[b] = ds.idx_all[g];
  }
}

}  // namespace

extern "C" int hcg_collate(const hcg_collate_args* a, hcg_stream_t stream) {
  if (!a || a->nslots < 1 || a->F <= 0 || a->F >= (1 << 20)) return HCG_ERR_INVALID_ARG;
  if (!a->x_all || !a->node_ptr_all || !a->edge_ptr_all) return HCG_ERR_INVALID_ARG;
  const Dataset ds{a->x_all, a->src_all, a->dst_all, a->node_ptr_all, a->edge_ptr_all, a->y_all, a->idx_all, (int)a->F};
  if (a->nslots == 1) {
    const hcg_collate_slot& s = a->slot;
    if (s.B < 0 || s.N_out < 0 || s.E_out < 0) return HCG_ERR_INVALID_ARG;
    if (s.B == 0) return HCG_OK;
    if (!s.ids || !s.graph_ptr || !s.edge_ptr || !s.x_out || !s.batch_out) return HCG_ERR_INVALID_ARG;
    if (s.E_out > 0 && (!a->src_all || !a->dst_all || !s.edge_index_out)) return HCG_ERR_INVALID_ARG;
    if ((s.y_out && !a->y_all) || (s.idx_out && !a->idx_all)) return HCG_ERR_INVALID_ARG;
    hipLaunchKernelGGL(k_collate<false>, dim3((unsigned)s.B), dim3(256), 0, (hipStream_t)stream, ds, s, nullptr);
  } else {
    if (!a->slots_dev || a->max_B < 1 || a->nslots > 65535 || !a->src_all || !a->dst_all) return HCG_ERR_INVALID_ARG;
    hipLaunchKernelGGL(k_collate<true>, dim3((unsigned)a->max_B, (unsigned)a->nslots), dim3(256), 0, (hipStream_t)stream, ds,
                       hcg_collate_slot{}, a->slots_dev);
  }
  HCG_CHECK_LAUNCH();
  return HCG_OK;
}
