// One launch for every pending gradient-slab reduction of a backward pass.
// The backward kernels (fused.hip, readout.hip) leave per-workgroup partial slabs; each of the three
// reductions of a step used to be its own ~4.4 us launch.  A job = one slab set, split into up to four
// segments that are written (with optional row compaction KPAD -> F) to their gradient tensors.
// Every output element is summed over the slabs in a fixed order: bitwise reproducible.
#include "common.h"

namespace {

struct Jobs {
  int njobs;
  hcg_reduce_job job[HCG_REDUCE_MAX_JOBS];
};

constexpr int RS = 16;  // slab slices per output element

__global__ __launch_bounds__(256) void k_reduce_jobs(Jobs jobs) {
  __shared__ float part[RS][16];
  const hcg_reduce_job& J = jobs.job[blockIdx.y];
  const int o = threadIdx.x & 15, sl = threadIdx.x >> 4;
  const int idx = blockIdx.x * 16 + o;
  if (blockIdx.x * 16 >= J.slab_floats) return;   // block-uniform
  float s = 0.f;
  if (idx < J.slab_floats) {
#pragma unroll 8
    for (int b = sl; b < J.nslabs; b += RS) s += J.slabs[(size_t)b * J.slab_floats + idx];
  }
  part[sl][o] = s;
  __syncthreads();
  if (sl == 0 && idx < J.slab_floats) {
    float tot = 0.f;
#pragma unroll
    for (int k = 0; k < RS; ++k) tot += part[k][o];
    for (int g = 0; g < J.nseg; ++g) {
      const hcg_reduce_seg& S = J.seg[g];
      const int rel = idx - S.begin;
      if (rel >= 0 && rel < S.count) {
        const int rr = rel / S.row_in, cc = rel - rr * S.row_in;
        if (cc < S.row_out) S.dst[(size_t)rr * S.row_out + cc] = tot;
      }
    }
  }
}

}  // namespace

extern "C" int hcg_reduce_slabs(const hcg_reduce_job* jobs_host, int njobs, hcg_stream_t stream) {
  if (njobs < 0 || njobs > HCG_REDUCE_MAX_JOBS || (njobs > 0 && !jobs_host)) return HCG_ERR_INVALID_ARG;
  if (njobs == 0) return HCG_OK;
  Jobs jobs;
  jobs.njobs = njobs;
  int max_floats = 0;
  for (int j = 0; j < njobs; ++j) {
    const hcg_reduce_job& J = jobs_host[j];
    if (!J.slabs || J.nslabs < 0 || J.slab_floats <= 0 || J.nseg < 0 || J.nseg > HCG_REDUCE_MAX_SEGS) return HCG_ERR_INVALID_ARG;
    for (int g = 0; g < J.nseg; ++g)
      if (!J.seg[g].dst || J.seg[g].row_in <= 0 || J.seg[g].row_out <= 0 || J.seg[g].row_out > J.seg[g].row_in)
        return HCG_ERR_INVALID_ARG;
    jobs.job[j] = J;
    if (J.slab_floats > max_floats) max_floats = J.slab_floats;
  }
  for (int j = njobs; j < HCG_REDUCE_MAX_JOBS; ++j) jobs.job[j] = jobs.job[0];
  hipLaunchKernelGGL(k_reduce_jobs, dim3((max_floats + 15) / 16, njobs), dim3(256), 0, (hipStream_t)stream, jobs);
  HCG_CHECK_LAUNCH();
  return HCG_OK;
}

extern "C" size_t hcg_reduce_job_bytes(void) { return sizeof(hcg_reduce_job); }
