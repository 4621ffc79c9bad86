// One launch for every pending gradient-slab reduction of a backward pass.
// The backward kernels (fused.hip, readout.hip) leave per-workgroup partial slabs; each of the three
// reductions of a step used to be its own ~4.4 us launch.  A job = one slab set, split into up to four
// segments that are written (with optional row compaction KPAD -> F) to their gradient tensors.
// Every output element is summed over the slabs in a fixed order: bitwise reproducible.
#include "common.h"
#include <string.h>
#include "ptrs.h"

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

struct Jobs {
  int njobs;
  hcg_reduce_job job[HCG_REDUCE_MAX_JOBS];
};

// slab slices per output element.  8 slices x 32 outputs per block: every slab row a wave touches is a full 128-byte line
// (measured per launch at C3: 32 x 8 = 9.5 us, 16 x 16 = 6.2, 8 x 32 = 4.8, 4 x 64 = 6.0; 16 loads in flight per thread: 4.5)
constexpr int RS = 8;
constexpr int RO = 256 / RS;   // output elements per 256-thread block

// Adam state for the fused "reduce, then update" variant: every reduced gradient element is written to its place in
// the flat gradient buffer AND immediately used for torch.optim.Adam's update of the parameter at the same offset
// (one launch fewer per step; the reference's optimiser: model/networks.py:38).
struct AdamArgs {
  const float* grad_flat;   // base of the flat gradient buffer every segment's dst points into
  float* p;
  float* m;
  float* v;
  const float* lr_dev;
  const int* step_dev;      // [0] = 1-based number of this update
  float b1, b2, eps;
};

// PLAN: row `njobs` of the grid derives the pointers of the NEXT batch's plan (hcg_ptrs_thread) beside the reduction -- the
// one launch of the next step that depends on nothing of this step, folded into this step's last launch.
struct PlanArgs {
  const int64_t* ei;
  const int64_t* batch;
  int64_t N, E, B;
  int32_t* graph_ptr;
  int32_t* edge_ptr;
  int32_t* status;
};

// XCHG: data-parallel one-shot exchange fused between the reduction and the update.  Every rank owns an
// INBOX of 2 x world x (n + 2) eight-byte granules {fp32 value, step stamp} in fine-grained (uncached) device memory that
// its peers map through IPC.  The thread that finishes a gradient element stores it as ONE 8-byte system-scope granule into
// every peer's inbox over the direct xGMI link (a granule is written by one store: no flag, no ordering needed), then polls
// the `world - 1` granules its peers wrote into ITS inbox until their stamp is this step's, adds all contributions in
// RANK ORDER (bitwise the same sum on every rank: replicas cannot drift) and applies the update.  Inboxes are double-buffered
// by step parity: a peer can only overwrite the slots of step s at step s + 2, which it reaches after it has received this
// rank's step-(s + 1) granules, i.e. after this rank has finished reading step s.  Every poll is bounded (2 s of the
// 100 MHz clock): on expiry the error word gets HCG_XCHG_ERR_TIMEOUT and the element becomes NaN instead of a hang.
struct XchgArgs {
  unsigned long long* inbox;                     // this rank's inbox (local, fine-grained)
  unsigned long long* peer[HCG_XCHG_MAX_WORLD];  // every rank's inbox as mapped here (peer[rank] == inbox)
  int rank, world;
  int64_t n_ext;                                 // n + 2: gradients | SSE | count
  int mode;                                      // HCG_XCHG_MEAN: (sum over ranks) / world; HCG_XCHG_SSE: the SSE form's scale
  float* flat_ext;                               // [n + 2] local: gradients are written back here, [n], [n + 1] = this rank's SSE, count
  float* loss;                                   // [2]: SSE mode: global sqrt(MSE), MSE
  int32_t* err;
};

constexpr unsigned long long XCHG_SPIN_TICKS = 200000000ull;

__device__ __forceinline__ void xchg_publish(const XchgArgs& X, int parity, int64_t elem, float v, unsigned step) {
  const unsigned long long g = ((unsigned long long)step << 32) | (unsigned long long)__float_as_uint(v);
  const size_t slot = ((size_t)parity * X.world + X.rank) * X.n_ext + elem;
  for (int p = 0; p < X.world; ++p)
    if (p != X.rank) __hip_atomic_store(X.peer[p] + slot, g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// sum over the ranks in rank order; `mine` = this rank's own contribution
__device__ __forceinline__ float xchg_gather(const XchgArgs& X, int parity, int64_t elem, float mine, unsigned step) {
  float sum = 0.f;
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  for (int q = 0; q < X.world; ++q) {
    float v = mine;
    if (q != X.rank) {
      const unsigned long long* src = X.inbox + ((size_t)parity * X.world + q) * X.n_ext + elem;
      unsigned long long g;
      for (;;) {
        g = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if ((unsigned)(g >> 32) == step) break;
        if (__builtin_amdgcn_s_memrealtime() - t0 > XCHG_SPIN_TICKS) {
          atomicOr(X.err, HCG_XCHG_ERR_TIMEOUT);
          g = 0x7fc00000ull;                         // NaN
          break;
        }
        __builtin_amdgcn_s_sleep(1);
      }
      v = __uint_as_float((unsigned)g);
    }
    sum += v;
  }
  return sum;
}

template <bool ADAM, bool PLAN = false, bool XCHG = false>
__global__ __launch_bounds__(256) void k_reduce_jobs(Jobs jobs, AdamArgs A, PlanArgs P = PlanArgs{}, XchgArgs X = XchgArgs{}) {
  if (PLAN && (int)blockIdx.y == jobs.njobs) {        // block-uniform
    const int64_t total = P.N + P.E + 2, stride = (int64_t)gridDim.x * 256;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += stride)
      hcg_ptrs_thread(t, P.ei, P.batch, P.N, P.E, P.B, P.graph_ptr, P.edge_ptr, P.status);
    return;
  }
  __shared__ float part[RS][RO];
  const hcg_reduce_job& J = jobs.job[blockIdx.y];
  const int o = threadIdx.x % RO, sl = threadIdx.x / RO;
  const int idx = blockIdx.x * RO + o;
  if (blockIdx.x * RO >= J.slab_floats) return;   // block-uniform
  __shared__ float adam_c[3];                        // lr / bias-correction-1, sqrt(bias-correction-2), [XCHG] gradient scale
  const unsigned xstep = XCHG ? (unsigned)A.step_dev[0] : 0u;
  const int parity = (int)(xstep & 1u);
  if (ADAM && threadIdx.x == 0) {                    // bias corrections in double like torch's host computation
    const int t = A.step_dev[0];                     // number of THIS update (advanced earlier in the step)
    adam_c[0] = A.lr_dev[0] / (float)(1.0 - hcg_powi((double)A.b1, t));
    adam_c[1] = (float)sqrt(1.0 - hcg_powi((double)A.b2, t));
  }
  if (XCHG && threadIdx.x == 255) {                  // the two tail elements [SSE, count]: published once per rank, gathered by every block
    const int64_t n = X.n_ext - 2;
    const float sse = X.flat_ext[n], cnt = X.flat_ext[n + 1];
    if (blockIdx.x == 0 && blockIdx.y == 0) {
      xchg_publish(X, parity, n, sse, xstep);
      xchg_publish(X, parity, n + 1, cnt, xstep);
    }
    float scale = 1.0f / (float)X.world;
    if (X.mode == HCG_XCHG_SSE) {
      const float sse_t = xchg_gather(X, parity, n, sse, xstep), cnt_t = xchg_gather(X, parity, n + 1, cnt, xstep);
      const float mse = sse_t / cnt_t, lv = sqrtf(mse);
      scale = 1.0f / (cnt_t * lv);
      if (blockIdx.x == 0 && blockIdx.y == 0) { X.loss[0] = lv; X.loss[1] = mse; }
    }
    adam_c[2] = scale;
  }
  // where this output element goes (threads of slice 0 only), and -- fused update -- its parameter and moments,
  // requested BEFORE the slab loop: the update then waits on nothing but the sum
  float* gdst = nullptr;
  size_t off = 0;
  float m0 = 0.f, v0 = 0.f, p0 = 0.f;
  if (sl == 0 && idx < J.slab_floats) {
    for (int g = 0; g < J.nseg; ++g) {
      const hcg_reduce_seg& S = J.seg[g];
      const int rel = idx - S.begin;
      if (rel >= 0 && rel < S.count) {
        const int rr = rel / S.row_in, cc = rel - rr * S.row_in;
        if (cc < S.row_out) gdst = S.dst + (size_t)rr * S.row_out + cc;
      }
    }
    if (ADAM && gdst) {
      off = (size_t)(gdst - A.grad_flat);
      m0 = A.m[off];
      v0 = A.v[off];
      p0 = A.p[off];
    }
  }
  float s = 0.f;
  if (idx < J.slab_floats) {
#pragma unroll 16   // (the head's 128 slabs = exactly one group of 16 per thread, the conv layers' 256 two)
    for (int b = sl; b < J.nslabs; b += RS) s += J.slabs[(size_t)b * J.slab_floats + idx];
  }
  part[sl][o] = s;
  __syncthreads();
  if (gdst) {
    float tot = 0.f;
#pragma unroll
    for (int k = 0; k < RS; ++k) tot += part[k][o];
    if (XCHG) {                                        // this rank's partial -> every peer; all ranks' partials -> the gradient
      xchg_publish(X, parity, (int64_t)off, tot, xstep);
      tot = xchg_gather(X, parity, (int64_t)off, tot, xstep) * adam_c[2];
    }
    *gdst = tot;
    if (ADAM) {
      const float mi = A.b1 * m0 + (1.0f - A.b1) * tot;
      const float vi = A.b2 * v0 + (1.0f - A.b2) * tot * tot;
      A.m[off] = mi;
      A.v[off] = vi;
      A.p[off] = p0 - adam_c[0] * (mi / (sqrtf(vi) / adam_c[1] + A.eps));
    }
  }
}

}  // namespace

static int launch_reduce(const hcg_reduce_job* jobs_host, int njobs, const AdamArgs* adam, int64_t n_flat, hipStream_t stream,
                         const PlanArgs* plan = nullptr, const XchgArgs* xchg = nullptr) {
  if (njobs < 0 || njobs > HCG_REDUCE_MAX_JOBS || (njobs > 0 && !jobs_host)) return HCG_ERR_INVALID_ARG;
  if (njobs == 0) return adam ? HCG_ERR_INVALID_ARG : HCG_OK;
  Jobs jobs;
  jobs.njobs = njobs;
  int max_floats = 0;
  for (int j = 0; j < njobs; ++j) {
    const hcg_reduce_job& J = jobs_host[j];
    if (!J.slabs || J.nslabs < 0 || J.slab_floats <= 0 || J.nseg < 0 || J.nseg > HCG_REDUCE_MAX_SEGS) return HCG_ERR_INVALID_ARG;
    for (int g = 0; g < J.nseg; ++g) {
      const hcg_reduce_seg& S = J.seg[g];
      if (!S.dst || S.row_in <= 0 || S.row_out <= 0 || S.row_out > S.row_in || S.count < 0) return HCG_ERR_INVALID_ARG;
      if (adam) {   // the update addresses param / moments by the gradient's offset: every dst must lie in the flat buffer
        const int64_t rows = (S.count + S.row_in - 1) / S.row_in;
        if (S.dst < adam->grad_flat || S.dst + rows * S.row_out > adam->grad_flat + n_flat) return HCG_ERR_INVALID_ARG;
      }
    }
    jobs.job[j] = J;
    if (J.slab_floats > max_floats) max_floats = J.slab_floats;
  }
  for (int j = njobs; j < HCG_REDUCE_MAX_JOBS; ++j) jobs.job[j] = jobs.job[0];
  // (plan row: one thread per node / edge as k_ptrs runs it -- fewer, looping blocks serialise its dependent loads: +3 us)
  unsigned gx = (max_floats + RO - 1) / RO;
  if (plan) { const unsigned px = (unsigned)hcg_cdiv(plan->N + plan->E + 2, 256); if (px > gx) gx = px; }
  const dim3 grid(gx, njobs + (plan ? 1 : 0));
  if (xchg && adam && plan) hipLaunchKernelGGL((k_reduce_jobs<true, true, true>), grid, dim3(256), 0, stream, jobs, *adam, *plan, *xchg);
  else if (xchg && adam) hipLaunchKernelGGL((k_reduce_jobs<true, false, true>), grid, dim3(256), 0, stream, jobs, *adam, PlanArgs{}, *xchg);
  else if (plan && adam) hipLaunchKernelGGL((k_reduce_jobs<true, true, false>), grid, dim3(256), 0, stream, jobs, *adam, *plan, XchgArgs{});
  else if (adam) hipLaunchKernelGGL((k_reduce_jobs<true, false, false>), grid, dim3(256), 0, stream, jobs, *adam, PlanArgs{}, XchgArgs{});
  else hipLaunchKernelGGL((k_reduce_jobs<false, false, false>), grid, dim3(256), 0, stream, jobs, AdamArgs{}, PlanArgs{}, XchgArgs{});
  HCG_CHECK_LAUNCH();
  return HCG_OK;
}

extern "C" int hcg_reduce_slabs(const hcg_reduce_job* jobs_host, int njobs, hcg_stream_t stream) {
  return launch_reduce(jobs_host, njobs, nullptr, 0, (hipStream_t)stream);
}

extern "C" int hcg_reduce_slabs_adam(const hcg_reduce_job* jobs_host, int njobs, const float* grad_flat, float* param_flat,
                                     float* exp_avg, float* exp_avg_sq, int64_t n, const float* lr_dev, float beta1,
                                     float beta2, float eps, const int32_t* step_dev, hcg_stream_t stream) {
  if (n <= 0 || !grad_flat || !param_flat || !exp_avg || !exp_avg_sq || !lr_dev || !step_dev) return HCG_ERR_INVALID_ARG;
  AdamArgs a{grad_flat, param_flat, exp_avg, exp_avg_sq, lr_dev, (const int*)step_dev, beta1, beta2, eps};
  return launch_reduce(jobs_host, njobs, &a, n, (hipStream_t)stream);
}

// hcg_reduce_slabs_adam + the pointers-only plan (HCG_PLAN_BLOCKED | HCG_PLAN_PTRS_ONLY | HCG_PLAN_KEEP_STATUS) of the NEXT
// batch in the same launch
extern "C" int hcg_reduce_slabs_adam_plan(const hcg_reduce_job* jobs_host, int njobs, const float* grad_flat, float* param_flat,
                                          float* exp_avg, float* exp_avg_sq, int64_t n, const float* lr_dev, float beta1,
                                          float beta2, float eps, const int32_t* step_dev, const int64_t* next_edge_index,
                                          const int64_t* next_batch, int64_t N, int64_t E, int64_t B, int32_t* next_graph_ptr,
                                          int32_t* next_edge_ptr, int32_t* next_status, hcg_stream_t stream) {
  if (n <= 0 || !grad_flat || !param_flat || !exp_avg || !exp_avg_sq || !lr_dev || !step_dev) return HCG_ERR_INVALID_ARG;
  if (N < 0 || E < 0 || B < 0 || !next_batch || !next_graph_ptr || !next_edge_ptr || !next_status || (E > 0 && !next_edge_index))
    return HCG_ERR_INVALID_ARG;
  AdamArgs a{grad_flat, param_flat, exp_avg, exp_avg_sq, lr_dev, (const int*)step_dev, beta1, beta2, eps};
  PlanArgs pl{next_edge_index, next_batch, N, E, B, next_graph_ptr, next_edge_ptr, next_status};
  return launch_reduce(jobs_host, njobs, &a, n, (hipStream_t)stream, &pl);
}

// ---- one-shot gradient exchange over xGMI (data parallel): buffers + the fused launch ------------------------------
extern "C" size_t hcg_xchg_inbox_bytes(int64_t n, int world) {
  if (n <= 0 || world < 1 || world > HCG_XCHG_MAX_WORLD) return 0;
  return (size_t)2 * world * (n + 2) * sizeof(unsigned long long);
}
// fine-grained (uncached) device memory for an inbox: the ONE allocation this library makes (a peer's kernel writes into
// it while this rank's kernel polls it, which ordinary coarse-grained device memory does not make visible); zero-filled
extern "C" int hcg_xchg_alloc(size_t bytes, void** ptr) {
  if (!ptr || bytes == 0) return HCG_ERR_INVALID_ARG;
  hipError_t e = hipExtMallocWithFlags(ptr, bytes, hipDeviceMallocUncached);
  if (e != hipSuccess) return hcg_hip_err(e);
  e = hipMemset(*ptr, 0, bytes);
  if (e != hipSuccess) return hcg_hip_err(e);
  return hcg_hip_err(hipDeviceSynchronize());
}
extern "C" int hcg_xchg_free(void* ptr) { return ptr ? hcg_hip_err(hipFree(ptr)) : HCG_OK; }
// zero an inbox (between optimisers: the stamps are step numbers); synchronises the device
extern "C" int hcg_xchg_zero(void* ptr, size_t bytes) {
  if (!ptr) return HCG_ERR_INVALID_ARG;
  hipError_t e = hipDeviceSynchronize();
  if (e == hipSuccess) e = hipMemset(ptr, 0, bytes);
  if (e == hipSuccess) e = hipDeviceSynchronize();
  return hcg_hip_err(e);
}
extern "C" int hcg_xchg_ipc_export(void* ptr, void* handle64) {
  if (!ptr || !handle64) return HCG_ERR_INVALID_ARG;
  static_assert(sizeof(hipIpcMemHandle_t) == HCG_XCHG_HANDLE_BYTES, "IPC handle size");
  return hcg_hip_err(hipIpcGetMemHandle(reinterpret_cast<hipIpcMemHandle_t*>(handle64), ptr));
}
extern "C" int hcg_xchg_ipc_open(const void* handle64, void** ptr) {
  if (!ptr || !handle64) return HCG_ERR_INVALID_ARG;
  hipIpcMemHandle_t h;
  memcpy(&h, handle64, sizeof(h));
  return hcg_hip_err(hipIpcOpenMemHandle(ptr, h, hipIpcMemLazyEnablePeerAccess));
}
extern "C" int hcg_xchg_ipc_close(void* ptr) { return ptr ? hcg_hip_err(hipIpcCloseMemHandle(ptr)) : HCG_OK; }

extern "C" int hcg_reduce_slabs_xchg_adam(const hcg_reduce_job* jobs_host, int njobs, float* flat_ext, float* param_flat,
                                          float* exp_avg, float* exp_avg_sq, int64_t n, const float* lr_dev, float beta1,
                                          float beta2, float eps, const int32_t* step_dev, void* inbox, void* const* peers,
                                          int rank, int world, int mode, float* loss, int32_t* err,
                                          const int64_t* next_edge_index, const int64_t* next_batch, int64_t N, int64_t E,
                                          int64_t B, int32_t* next_graph_ptr, int32_t* next_edge_ptr, int32_t* next_status,
                                          hcg_stream_t stream) {
  if (n <= 0 || !flat_ext || !param_flat || !exp_avg || !exp_avg_sq || !lr_dev || !step_dev || !inbox || !peers || !loss || !err)
    return HCG_ERR_INVALID_ARG;
  if (world < 1 || world > HCG_XCHG_MAX_WORLD || rank < 0 || rank >= world || (mode != HCG_XCHG_MEAN && mode != HCG_XCHG_SSE))
    return HCG_ERR_INVALID_ARG;
  AdamArgs a{flat_ext, param_flat, exp_avg, exp_avg_sq, lr_dev, (const int*)step_dev, beta1, beta2, eps};
  XchgArgs x{};
  x.inbox = (unsigned long long*)inbox;
  for (int p = 0; p < world; ++p) {
    if (!peers[p]) return HCG_ERR_INVALID_ARG;
    x.peer[p] = (unsigned long long*)peers[p];
  }
  x.rank = rank; x.world = world; x.n_ext = n + 2; x.mode = mode; x.flat_ext = flat_ext; x.loss = loss; x.err = err;
  if (next_batch) {
    if (N < 0 || E < 0 || B < 0 || !next_graph_ptr || !next_edge_ptr || !next_status || (E > 0 && !next_edge_index)) return HCG_ERR_INVALID_ARG;
    PlanArgs pl{next_edge_index, next_batch, N, E, B, next_graph_ptr, next_edge_ptr, next_status};
    return launch_reduce(jobs_host, njobs, &a, n, (hipStream_t)stream, &pl, &x);
  }
  return launch_reduce(jobs_host, njobs, &a, n, (hipStream_t)stream, nullptr, &x);
}

extern "C" size_t hcg_reduce_job_bytes(void) { return sizeof(hcg_reduce_job); }

// Two backward launches of ONE layer over two groups of graphs (size-grouped batches: small-graph tiles + one graph per
// wave) leave their slabs back to back in one workspace: `more` is appended to `job` -- same slab geometry, same
// destinations, slabs contiguous -- so that the layer's gradient is ONE fixed-order sum over all of them.
extern "C" int hcg_reduce_job_append(hcg_reduce_job* job, const hcg_reduce_job* more) {
  if (!job || !more || job->slab_floats != more->slab_floats || job->nseg != more->nseg) return HCG_ERR_INVALID_ARG;
  if (more->slabs != job->slabs + (size_t)job->nslabs * job->slab_floats) return HCG_ERR_INVALID_ARG;
  for (int g = 0; g < job->nseg; ++g) {
    const hcg_reduce_seg &a = job->seg[g], &b = more->seg[g];
    if (a.begin != b.begin || a.count != b.count || a.row_in != b.row_in || a.row_out != b.row_out || a.dst != b.dst)
      return HCG_ERR_INVALID_ARG;
  }
  job->nslabs += more->nslabs;
  return HCG_OK;
}
