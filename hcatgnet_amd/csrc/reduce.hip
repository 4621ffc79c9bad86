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
  int nblk[HCG_REDUCE_MAX_JOBS];     // workgroups of job j: ceil(slab_floats / RO) -- the grid is their sum (+ the plan's)
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
  const int* step_dev;      // [0] = 1-based number of this update, [1] = exchange stamp (monotonic)
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
  const float* flat_ext;                         // [n + 2] local; without SSE partials in a job, [n], [n + 1] = this rank's SSE, count
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

// LOSS: the deferred loss scale.  Every backward launch of the step ran on the UNSCALED error (out - y): the whole
// backward is linear in dloss/dout = scale * (out - y), scale being the one number that depends on the whole batch
// (1 / (count * sqrt(MSE)) for the reference's sqrt(MSE), utils/utils_model.py:64).  The readout head leaves one partial sum
// of squared errors per workgroup behind its slabs (hcg_reduce_job.sse_part); every block of THIS launch adds them in the same
// fixed order (bitwise the same scale everywhere), and the scale multiplies each gradient element as it is reduced --
// no grid-wide exchange, no launch of its own (round 2's head kernel spent a grid barrier on this scalar).
struct LossArgs {
  const float* sse_part;    // [nparts] partials; nullptr = the slabs hold final gradients (scale 1)
  int nparts;
  float count;              // elements of the squared-error sum on this rank (B * C)
  int mode;                 // HCG_LOSS_MSE / HCG_LOSS_RMSE / HCG_LOSS_SSE
  float* loss;              // [2] nullable: the loss, the MSE
  float* sse_tail;          // [2] nullable: this rank's SSE and count (data-parallel "sse" form with a collective)
};

// sum of the SSE partials: lanes take partials l, l + 64, ... in ascending order, then a fixed xor tree
__device__ __forceinline__ float loss_sse_sum(const LossArgs& L, int lane) {
  float s = 0.f;
  for (int b = lane; b < L.nparts; b += 64) s += L.sse_part[b];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
  return s;
}

template <bool ADAM, bool XCHG = false>
__global__ __launch_bounds__(256) void k_step_tail(Jobs jobs, AdamArgs A, PlanArgs P, LossArgs L, XchgArgs X = XchgArgs{}) {
  // a 1-D grid without idle workgroups: [blocks of job 0 | blocks of job 1 | ... | the plan's blocks] (as a 2-D grid of
  // njobs + 1 rows as wide as the widest -- the plan's -- two thirds of the 6 000 workgroups at C3 returned at once)
  int blk = blockIdx.x, jsel = 0;
  while (jsel < jobs.njobs && blk >= jobs.nblk[jsel]) { blk -= jobs.nblk[jsel]; ++jsel; }
  if (jsel == jobs.njobs) {                          // block-uniform: the NEXT batch's plan
    const int64_t t = (int64_t)blk * 256 + threadIdx.x;
    if (t < P.N + P.E + 2) hcg_ptrs_thread(t, P.ei, P.batch, P.N, P.E, P.B, P.graph_ptr, P.edge_ptr, P.status);
    return;
  }
  __shared__ float part[RS][RO];
  const hcg_reduce_job& J = jobs.job[jsel];
  const int o = threadIdx.x % RO, sl = threadIdx.x / RO;
  const int idx = blk * RO + o;
  // [0] lr / bias-correction-1, [1] sqrt(bias-correction-2), [2] scale applied to this rank's sum (before an exchange),
  // [3] scale applied to the exchanged total
  __shared__ float adam_c[4];
  // the exchange stamp is step_dev[1]: advanced with step_dev[0] by the step's first launch, but owned by no optimiser state --
  // reloading a checkpoint re-bases the Adam step count, never the stamp (a stamp an inbox has already seen would let a stale
  // granule pass as this step's)
  const unsigned xstep = XCHG ? (unsigned)A.step_dev[1] : 0u;
  const int parity = (int)(xstep & 1u);
  const bool first = blockIdx.x == 0;
  if (ADAM && threadIdx.x == 0) {                    // bias corrections in double like torch's host computation
    const int t = A.step_dev[0];                     // number of THIS update (advanced earlier in the step)
    adam_c[0] = A.lr_dev[0] / (float)(1.0 - hcg_powi((double)A.b1, t));
    adam_c[1] = (float)sqrt(1.0 - hcg_powi((double)A.b2, t));
  }
  if (threadIdx.x >= 192) {                          // wave 3: the loss, its scale, the exchange of [SSE, count]
    const int lane = threadIdx.x - 192;
    float sse = 0.f, cnt = 1.f, pre = 1.f, post = 1.f;
    const bool deferred = L.sse_part != nullptr;
    if (deferred) {
      sse = loss_sse_sum(L, lane);
      cnt = L.count;
      const float mse = sse / cnt, lv = sqrtf(mse);
      if (L.mode == HCG_LOSS_RMSE) pre = 1.0f / (cnt * lv);
      else if (L.mode == HCG_LOSS_MSE) pre = 2.0f / cnt;
      if (first && lane == 0) {
        if (L.loss && !(XCHG && X.mode == HCG_XCHG_SSE)) { L.loss[0] = L.mode == HCG_LOSS_MSE ? mse : lv; L.loss[1] = mse; }
        if (L.sse_tail) { L.sse_tail[0] = sse; L.sse_tail[1] = cnt; }
      }
    } else if (XCHG) {
      const int64_t n = X.n_ext - 2;
      sse = X.flat_ext[n];
      cnt = X.flat_ext[n + 1];
    }
    if (XCHG && lane == 63) {                        // the two tail elements: published once per rank, gathered by every block
      const int64_t n = X.n_ext - 2;
      if (first) {
        xchg_publish(X, parity, n, sse, xstep);
        xchg_publish(X, parity, n + 1, cnt, xstep);
      }
      post = 1.0f / (float)X.world;
      if (X.mode == HCG_XCHG_SSE) {
        const float sse_t = xchg_gather(X, parity, n, sse, xstep), cnt_t = xchg_gather(X, parity, n + 1, cnt, xstep);
        const float mse = sse_t / cnt_t, lv = sqrtf(mse);
        pre = 1.0f;
        post = 1.0f / (cnt_t * lv);
        if (first) { X.loss[0] = lv; X.loss[1] = mse; }
      }
    }
    if (lane == 63) { adam_c[2] = pre; adam_c[3] = post; }
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
    tot *= adam_c[2];
    if (XCHG) {                                        // this rank's partial -> every peer; all ranks' partials -> the gradient
      xchg_publish(X, parity, (int64_t)off, tot, xstep);
      tot = xchg_gather(X, parity, (int64_t)off, tot, xstep) * adam_c[3];
    }
    *gdst = tot;
    if (ADAM) {
      float mi = m0, vi = v0;
      A.p[off] = hcg_adam_update(p0, tot, mi, vi, A.b1, A.b2, A.eps, adam_c[0], adam_c[1]);
      A.m[off] = mi;
      A.v[off] = vi;
    }
  }
}

// the loss alone from a head's SSE partials (forward-only steps: the reference's eval_network body, utils/utils_model.py:75-78)
__global__ __launch_bounds__(64) void k_loss_finalize(LossArgs L) {
  const float sse = loss_sse_sum(L, threadIdx.x);
  if (threadIdx.x == 0) {
    const float mse = sse / L.count;
    L.loss[0] = L.mode == HCG_LOSS_MSE ? mse : sqrtf(mse);
    L.loss[1] = mse;
    if (L.sse_tail) { L.sse_tail[0] = sse; L.sse_tail[1] = L.count; }
  }
}

}  // namespace

// the job that carries the head's SSE partials -> LossArgs (at most one per step)
static int loss_args_from_jobs(const hcg_reduce_job* jobs_host, int njobs, float count, int mode, float* loss, float* sse_tail,
                               LossArgs* out) {
  LossArgs L{};
  for (int j = 0; j < njobs; ++j) {
    const hcg_reduce_job& J = jobs_host[j];
    if (!J.sse_part) continue;
    if (L.sse_part || J.nslabs < 1) return HCG_ERR_INVALID_ARG;
    L.sse_part = J.sse_part;
    L.nparts = J.nslabs;
  }
  if (L.sse_part) {
    if (!(count > 0.f) || (mode != HCG_LOSS_MSE && mode != HCG_LOSS_RMSE && mode != HCG_LOSS_SSE)) return HCG_ERR_INVALID_ARG;
    L.count = count; L.mode = mode; L.loss = loss; L.sse_tail = sse_tail;
  }
  *out = L;
  return HCG_OK;
}

// Workgroups of the exchanging tail that can be resident at once (occupancy query x CUs, cached).  A polling workgroup waits
// for granules that the peer publishes from ITS workgroups; if more workgroups poll than fit on the chip, a rank's resident
// ones could wait for elements whose publishers are still queued behind the peer's resident pollers -- a circular wait that
// only the bounded polls would end.  hcg_step_tail refuses such a launch instead (HCG_ERR_UNSUPPORTED); blocks without an
// output element return at once and do not count.
extern "C" int hcg_xchg_resident_blocks(void) {
  static int cap = -1;      // queried once per process (also keeps the query out of a stream capture)
  if (cap >= 0) return cap;
  int dev = 0, cus = 0, per_cu = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_step_tail<true, true>, 256, 0) != hipSuccess) per_cu = 0;
  cap = cus * per_cu;
  return cap;
}

// sizeof of the ABI's HOST structs, for bindings to check their mirrors against
extern "C" size_t hcg_struct_bytes(int which) {
  switch (which) {
    case HCG_STRUCT_REDUCE_JOB: return sizeof(hcg_reduce_job);
    case HCG_STRUCT_TAIL_ARGS: return sizeof(hcg_tail_args);
    case HCG_STRUCT_FUSED_FWD_ARGS: return sizeof(hcg_fused_fwd_args);
    case HCG_STRUCT_COLLATE_ARGS: return sizeof(hcg_collate_args);
    case HCG_STRUCT_COLLATE_SLOT: return sizeof(hcg_collate_slot);
    default: return 0;
  }
}

extern "C" int hcg_step_tail(const hcg_tail_args* a, hcg_stream_t stream_) {
  if (!a) return HCG_ERR_INVALID_ARG;
  hipStream_t stream = (hipStream_t)stream_;
  const int njobs = a->njobs;
  const hcg_reduce_job* jobs_host = a->jobs_host;
  const bool adam = a->param != nullptr, plan = a->next_batch != nullptr, xchg = a->inbox != nullptr;
  if (njobs < 0 || njobs > HCG_REDUCE_MAX_JOBS || (njobs > 0 && !jobs_host)) return HCG_ERR_INVALID_ARG;
  if (njobs == 0) return (adam || plan || xchg) ? HCG_ERR_INVALID_ARG : HCG_OK;
  AdamArgs A{};
  if (adam) {
    if (a->n <= 0 || !a->grad_flat || !a->exp_avg || !a->exp_avg_sq || !a->lr_dev || !a->step_dev) return HCG_ERR_INVALID_ARG;
    A = AdamArgs{a->grad_flat, a->param, a->exp_avg, a->exp_avg_sq, a->lr_dev, (const int*)a->step_dev, a->beta1, a->beta2, a->eps};
  }
  PlanArgs P{};
  if (plan) {
    if (a->next_N < 0 || a->next_E < 0 || a->next_B < 0 || !a->next_graph_ptr || !a->next_edge_ptr || !a->next_status ||
        (a->next_E > 0 && !a->next_edge_index))
      return HCG_ERR_INVALID_ARG;
    P = PlanArgs{a->next_edge_index, a->next_batch, a->next_N, a->next_E, a->next_B, a->next_graph_ptr, a->next_edge_ptr, a->next_status};
  }
  XchgArgs X{};
  if (xchg) {
    if (!adam || !a->peers_host || !a->loss || !a->xchg_err) return HCG_ERR_INVALID_ARG;
    if (a->world < 1 || a->world > HCG_XCHG_MAX_WORLD || a->rank < 0 || a->rank >= a->world ||
        (a->xchg_mode != HCG_XCHG_MEAN && a->xchg_mode != HCG_XCHG_SSE))
      return HCG_ERR_INVALID_ARG;
    X.inbox = (unsigned long long*)a->inbox;
    for (int p = 0; p < a->world; ++p) {
      if (!a->peers_host[p]) return HCG_ERR_INVALID_ARG;
      X.peer[p] = (unsigned long long*)a->peers_host[p];
    }
    X.rank = a->rank; X.world = a->world; X.n_ext = a->n + 2; X.mode = a->xchg_mode; X.flat_ext = a->grad_flat; X.loss = a->loss;
    X.err = a->xchg_err;
  }
  LossArgs L{};
  HCG_TRY(loss_args_from_jobs(jobs_host, njobs, a->loss_count, a->loss_mode, a->loss, a->sse_tail, &L));
  Jobs jobs;
  jobs.njobs = njobs;
  unsigned total_blocks = 0;
  for (int j = 0; j < njobs; ++j) {
    const hcg_reduce_job& J = jobs_host[j];
    if (!J.slabs || J.nslabs < 0 || J.slab_floats <= 0 || J.nseg < 0 || J.nseg > HCG_REDUCE_MAX_SEGS) return HCG_ERR_INVALID_ARG;
    for (int g = 0; g < J.nseg; ++g) {
      const hcg_reduce_seg& S = J.seg[g];
      if (!S.dst || S.row_in <= 0 || S.row_out <= 0 || S.row_out > S.row_in || S.count < 0) return HCG_ERR_INVALID_ARG;
      if (adam) {   // the update addresses param / moments by the gradient's offset: every dst must lie in the flat buffer
        const int64_t rows = (S.count + S.row_in - 1) / S.row_in;
        if (S.dst < a->grad_flat || S.dst + rows * S.row_out > a->grad_flat + a->n) return HCG_ERR_INVALID_ARG;
      }
    }
    jobs.job[j] = J;
    jobs.nblk[j] = J.nseg > 0 ? (J.slab_floats + RO - 1) / RO : 0;      // (a job without segments only carries partials)
    total_blocks += (unsigned)jobs.nblk[j];
  }
  for (int j = njobs; j < HCG_REDUCE_MAX_JOBS; ++j) { jobs.job[j] = jobs.job[0]; jobs.nblk[j] = 0; }
  if (xchg && (int)total_blocks > hcg_xchg_resident_blocks()) return HCG_ERR_UNSUPPORTED;   // (see hcg_xchg_resident_blocks)
  // (plan blocks: one thread per node / edge as k_ptrs runs it -- fewer, looping blocks serialise its dependent loads: +3 us)
  if (plan) total_blocks += (unsigned)hcg_cdiv(P.N + P.E + 2, 256);
  if (total_blocks == 0) return HCG_OK;
  const dim3 grid(total_blocks);
  if (xchg) hipLaunchKernelGGL((k_step_tail<true, true>), grid, dim3(256), 0, stream, jobs, A, P, L, X);
  else if (adam) hipLaunchKernelGGL((k_step_tail<true, false>), grid, dim3(256), 0, stream, jobs, A, P, L, XchgArgs{});
  else hipLaunchKernelGGL((k_step_tail<false, false>), grid, dim3(256), 0, stream, jobs, A, P, L, XchgArgs{});
  HCG_CHECK_LAUNCH();
  return HCG_OK;
}

extern "C" int hcg_loss_finalize(const hcg_reduce_job* head_job_host, float count, int mode, float* loss, float* sse_tail,
                                 hcg_stream_t stream) {
  if (!head_job_host || !loss) return HCG_ERR_INVALID_ARG;
  LossArgs L{};
  HCG_TRY(loss_args_from_jobs(head_job_host, 1, count, mode, loss, sse_tail, &L));
  if (!L.sse_part) return HCG_ERR_INVALID_ARG;
  hipLaunchKernelGGL(k_loss_finalize, dim3(1), dim3(64), 0, (hipStream_t)stream, L);
  HCG_CHECK_LAUNCH();
  return HCG_OK;
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
