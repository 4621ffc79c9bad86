// graph_ptr / edge_ptr (+ device-side validation) of a blocked batch: the body shared by k_ptrs (plan.hip) and by the slab
// reduction that builds the NEXT batch's plan beside its own work (reduce.hip).  Thread `tid` of N + E + 2: threads [0, N]
// walk nodes, [N + 1, N + E + 1] edges.
#pragma once
#include "common.h"

__device__ __forceinline__ void hcg_ptrs_thread(int64_t tid, const int64_t* __restrict__ ei, const int64_t* __restrict__ batch,
                                                int64_t N, int64_t E, int64_t B, int32_t* __restrict__ graph_ptr,
                                                int32_t* __restrict__ edge_ptr, int32_t* __restrict__ status) {
  int st = 0;
  int64_t prev = -1, cur = B, pos;
  int32_t* dst;
  if (tid <= N) {
    const int64_t i = tid;
    pos = i; dst = graph_ptr;
    if (i > 0) prev = batch[i - 1];
    if (i < N) {
      cur = batch[i];
      if (cur < 0 || cur >= B) st |= HCG_STATUS_BATCH_RANGE;
      if (i > 0 && cur < prev) st |= HCG_STATUS_BATCH_UNSORTED;
    }
  } else if (tid <= N + 1 + E) {
    const int64_t e = tid - (N + 1);
    pos = e; dst = edge_ptr;
    if (e > 0) {
      int64_t s = ei[e - 1];
      if (s < 0 || s >= N) s = 0;  // range errors are flagged by the thread that owns the edge
      prev = batch[s];
    }
    if (e < E) {
      int64_t s = ei[e], d = ei[E + e];
      if (s < 0 || s >= N || d < 0 || d >= N) { st |= HCG_STATUS_INDEX_RANGE; s = 0; d = 0; }
      cur = batch[s];
      if (batch[d] != cur) st |= HCG_STATUS_EDGE_UNGROUPED;
      if (e > 0 && cur < prev) st |= HCG_STATUS_EDGE_UNGROUPED;
    }
  } else {
    return;
  }
  if (st) atomicOr(status, st);
  if (prev < -1) prev = -1;
  if (prev > B) prev = B;
  if (cur < 0) cur = 0;
  if (cur > B) cur = B;
  for (int64_t g = prev + 1; g <= cur; ++g) dst[g] = (int32_t)pos;
}
