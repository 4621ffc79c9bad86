// Regression head as ONE stand-alone launch (SURVEY rows a10, a12 and their part of a11; f2): the form used behind conv
// stacks whose forward is more than one launch (one graph per workgroup / wave, wide layers, size-grouped batches).  The
// small-graph tiles carry the same tile code in the tail of their forward launch instead (fused.hip).
// As separate launches (readout fwd, mse fwd, sqrt, three torch kernels of sqrt's backward, mse bwd, readout bwd) this
// dependent chain of eight tiny kernels cost ~39 us of a 132 us training step: pure launch latency around 67 MFLOP.
// One workgroup per 32-graph tile, forward and backward of a tile back to back; nothing waits for another workgroup
// (head_tile.h: the loss scale is deferred to the step's last launch), so the grid is simply min(tiles, CUs).
#include "common.h"
#include "head_tile.h"

namespace {
using namespace hcg_head;

template <int RD, int RC, bool BACKWARD>
__global__ __launch_bounds__(HC<RD>::NT, 1) void k_head(const float* __restrict__ emb, const float* __restrict__ y,
                                                       const float* __restrict__ W0, const float* __restrict__ b0,
                                                       const float* __restrict__ W1, const float* __restrict__ b1, int B, int C,
                                                       float slope, float* __restrict__ z, float* __restrict__ out,
                                                       float* __restrict__ demb, float* __restrict__ slabs,
                                                       int* __restrict__ step_counter) {
  __shared__ HeadLds<RD> L;
  HeadState<RD, RC> S;
  const int tiles = (B + RT - 1) / RT;
  if (step_counter && blockIdx.x == 0 && threadIdx.x == 0) { step_counter[0] += 1; step_counter[1] += 1; }   // this training step's number, for the update launched later | exchange stamp
  head_begin<RD, RC>(L, S, W0, b0, W1, b1, C);
  for (int t = blockIdx.x; t < tiles; t += gridDim.x) {
    const int g0 = t * RT, n = B - g0 < RT ? B - g0 : RT;
    head_tile<RD, RC, BACKWARD>(L, S, [g0](int row) { return g0 + row; }, n, C, slope, emb, y, W0, z, out, demb);
  }
  head_end<RD, RC, BACKWARD>(L, S, C, slabs + (size_t)blockIdx.x * HC<RD>::SLAB, slabs + (size_t)gridDim.x * HC<RD>::SLAB + blockIdx.x);
}

// D = 64: the stand-alone launch runs the latency-cut 16-row tile code of the forward's tail (head_tile.h: hcg_head16; 8
// waves, weight fragments in registers straight from L2): its per-tile chain is about half the 32-row code's, and without a
// grid-wide exchange nothing ties the grid to one workgroup per 32 graphs -- 16 graphs per tile fill twice the CUs
// (B = 4096: 256 workgroups instead of 128).
template <int RC, bool BACKWARD>
__global__ __launch_bounds__(hcg_head16::NT, 1) void k_head16(const float* __restrict__ emb, const float* __restrict__ y,
                                                             const float* __restrict__ W0, const float* __restrict__ b0,
                                                             const float* __restrict__ W1, const float* __restrict__ b1, int B,
                                                             int C, float slope, float* __restrict__ z, float* __restrict__ out,
                                                             float* __restrict__ demb, float* __restrict__ slabs,
                                                             int* __restrict__ step_counter) {
  namespace h16 = hcg_head16;
  __shared__ h16::Lds L;
  if (step_counter && blockIdx.x == 0 && threadIdx.x == 0) { step_counter[0] += 1; step_counter[1] += 1; }
  h16::Prefetch<RC> P;
  h16::prefetch<RC>(P, W0, b0, W1, C);
  h16::State<RC> S;
  h16::begin<RC>(S, b1, C);
  constexpr int T16 = h16::RT;
  const int tiles = (B + T16 - 1) / T16;
  bool first = true;
  for (int t = blockIdx.x; t < tiles; t += gridDim.x) {
    const int g0 = t * T16, n = B - g0 < T16 ? B - g0 : T16;
    h16::tile<RC, BACKWARD>(L, S, P, [g0](int row) { return g0 + row; }, n, C, slope, nullptr, emb, y, z, out, demb, first);
    first = false;
  }
  __syncthreads();
  h16::end<RC, BACKWARD>(L, S, C, slabs + (size_t)blockIdx.x * h16::SLAB, slabs + (size_t)gridDim.x * h16::SLAB + blockIdx.x);
}

int head_grid(int64_t B, int64_t D = 128) {
  const int rt = D == 64 ? hcg_head16::RT : RT;
  static int cus = 0;       // queried once per process (also keeps the query out of a stream capture)
  if (cus == 0) {
    int dev = 0, v = 0;
    cus = 256;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0)
      cus = v;
  }
  int grid = (int)((B + rt - 1) / rt);
  if (grid > cus) grid = cus;
  return grid < 1 ? 1 : grid;
}
size_t head_slab(int64_t D) { return D == 128 ? (size_t)HC<128>::SLAB : (size_t)HC<64>::SLAB; }

}  // namespace

extern "C" int hcg_head_supported(int64_t D, int64_t C) { return ((D == 64 || D == 128) && C >= 1 && C <= RCMAX) ? 1 : 0; }

// workspace: [grid][SLAB] gradient slabs | [grid] SSE partials
extern "C" size_t hcg_head_workspace_bytes(int64_t B, int64_t D) {
  if (D != 64 && D != 128) return 0;
  return hcg_align_up((size_t)head_grid(B, D) * (head_slab(D) + 1) * sizeof(float), 256) + 256;
}

extern "C" int hcg_head_fwd_bwd(const float* emb, const float* y, const float* W0, const float* b0, const float* W1,
                                const float* b1, int64_t B, int64_t D, int64_t C, float slope, int flags, float* z,
                                float* out, float* demb, void* workspace, size_t workspace_bytes, int32_t* step_counter,
                                hcg_stream_t stream) {
  if (!hcg_head_supported(D, C)) return HCG_ERR_UNSUPPORTED;
  if (flags & ~HCG_HEAD_FORWARD_ONLY) return HCG_ERR_INVALID_ARG;
  const bool bwd = !(flags & HCG_HEAD_FORWARD_ONLY);
  if (B <= 0 || !emb || !y || !W0 || !b0 || !W1 || !b1 || !z || !out || (bwd && !demb) || !workspace) return HCG_ERR_INVALID_ARG;
  if (workspace_bytes < hcg_head_workspace_bytes(B, D)) return HCG_ERR_WORKSPACE;
  const int grid = head_grid(B, D);
  float* slabs = (float*)workspace;
#define LAUNCH_HEAD(RD_, RC_, BW)                                                                                         \
  hipLaunchKernelGGL((k_head<RD_, RC_, BW>), dim3(grid), dim3(HC<RD_>::NT), 0, (hipStream_t)stream, emb, y, W0, b0, W1, b1,   \
                     (int)B, (int)C, slope, z, out, demb, slabs, (int*)step_counter)
#define DISPATCH_HEAD(RD_)                                                                            \
  do {                                                                                                \
    if (C == 1) { if (bwd) LAUNCH_HEAD(RD_, 1, true); else LAUNCH_HEAD(RD_, 1, false); }              \
    else        { if (bwd) LAUNCH_HEAD(RD_, RCMAX, true); else LAUNCH_HEAD(RD_, RCMAX, false); }      \
  } while (0)
#define LAUNCH_HEAD16(RC_, BW)                                                                                            \
  hipLaunchKernelGGL((k_head16<RC_, BW>), dim3(grid), dim3(hcg_head16::NT), 0, (hipStream_t)stream, emb, y, W0, b0, W1, b1,    \
                     (int)B, (int)C, slope, z, out, demb, slabs, (int*)step_counter)
  if (D == 128) {
    DISPATCH_HEAD(128);
  } else if (C == 1) {
    if (bwd) LAUNCH_HEAD16(1, true); else LAUNCH_HEAD16(1, false);
  } else {
    if (bwd) LAUNCH_HEAD16(RCMAX, true); else LAUNCH_HEAD16(RCMAX, false);
  }
#undef LAUNCH_HEAD16
#undef DISPATCH_HEAD
#undef LAUNCH_HEAD
  HCG_CHECK_LAUNCH();
  return HCG_OK;
}

extern "C" int hcg_head_reduce_job(const void* workspace, size_t workspace_bytes, int64_t B, int64_t D, int64_t C, float* dW0,
                                   float* db0, float* dW1, float* db1, hcg_reduce_job* job) {
  if (B <= 0 || (D != 64 && D != 128) || C < 1 || C > RCMAX || !job || !workspace) return HCG_ERR_INVALID_ARG;
  if (dW0 && (!db0 || !dW1 || !db1)) return HCG_ERR_INVALID_ARG;
  if (workspace_bytes < hcg_head_workspace_bytes(B, D)) return HCG_ERR_WORKSPACE;
  if (D == 128) head_fill_job<128>((const float*)workspace, head_grid(B, D), (int)C, dW0, db0, dW1, db1, job);
  else head_fill_job<64>((const float*)workspace, head_grid(B, D), (int)C, dW0, db0, dW1, db1, job);
  return HCG_OK;
}
