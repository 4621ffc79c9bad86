// Dev experiment (not product, kept for the record of DESIGN 7): a DENSE-adjacency forward for graphs up to 128 nodes --
// Y = (C + I) H' on the matrix cores with the count matrix as bytes in LDS -- measured against the CSR route of
// csrc/mid.hip (k_mid_layer_fwd) and rejected: 100.9 / 116.7 us per launch (layer 1 / layer 2 of the REAL batch, 4096
// graphs of 57..117 atoms) against 79.8 / 80.1 us.  288 MFMAs per 128-node graph (4.6 k matrix-pipe cycles per CU) and five
// workgroup barriers per graph cost more than the counting sort + LDS gather they replace.  Per-phase s_memtime stamps:
// -DHCG_DENSE_STAMP.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -fno-slp-vectorize -DHCG_DENSE_STAMP -o tools/probe_dense tools/probe_dense.hip \
//         -L hcatgnet_amd/csrc -lhcatgnet_hip '-Wl,-rpath,$ORIGIN/../hcatgnet_amd/csrc'
#include "../hcatgnet_amd/csrc/mid.hip"
#include <cstdio>
#include <vector>
#include <random>

// =====================================================================================================
// forward of one layer for graphs up to 128 nodes: DENSE adjacency on the matrix cores (round 3)
// =====================================================================================================
// The CSR route above spends most of a graph's ~10 us in dependent LDS round trips: the counting sort with its seven
// barriers (20 %) and the wavefront segmented sum (33 %), with three of eight waves busy in the GEMM in between.  The
// small-graph tiles never had that chain: their neighbourhood sum is a product with the tile's adjacency COUNT matrix on
// the matrix cores.  The same works for a whole 57..128-node graph: C[dst][src] as BYTES (128 x 136 B = 17 KB of LDS; LDS
// integer atomics on the containing dword, order-independent -> deterministic), C += I, in-degree counters beside it; then
//   Y[rb] = sum_kb C[rb][kb] H'[kb]        one 32 x 32 output block (rb, cb) per wave, 2 x nblk k-steps, the count fragment
//                                          exact in ONE bf16 piece (8 bytes = one ds_read_b64), H' split in registers
// -- 12 nblk^2 MFMAs per graph (192 at 128 nodes: 1.5 k cycles over the four SIMDs) instead of the CSR build + the gather.
// Counts stay exact while a node's in-degree is <= 254 (checked on the device: HCG_STATUS_SHAPE_LIMIT otherwise; molecular
// graphs have degree <= 6).  Summation order = the matrix pipe's (fixed), not "ascending neighbour id": results agree with
// the CSR route to f32 rounding, run to run bitwise.
constexpr int DENSE_MAX_NODES = 128;
__host__ __device__ inline int dense_cb(int npad) { return npad + 8; }     // bytes per count row: 8-byte aligned fragments

#ifdef HCG_DENSE_STAMP      // tools/probe_dense.hip: s_memtime stamps of the per-graph phases of the first workgroups
__device__ unsigned long long g_dense_stamp[4][MW][4][12];
#define DSTAMP(i)                                                                                           \
  do {                                                                                                      \
    __builtin_amdgcn_sched_barrier(0);                                                                      \
    unsigned long long _t;                                                                                  \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(_t)::"memory");                               \
    __builtin_amdgcn_sched_barrier(0);                                                                      \
    if ((threadIdx.x & 63) == 0 && blockIdx.x < 4 && dstamp_it < 4) g_dense_stamp[blockIdx.x][threadIdx.x >> 6][dstamp_it][(i)] = _t; \
  } while (0)
#else
#define DSTAMP(i) do { } while (0)
#endif

struct DenseLds {
  float* t0;              // [npad][HS]   X -> H' -> activated output
  short* wl;              // 3 planes of the pre-split weight image
  unsigned* cnt;          // [npad][dense_cb / 4]   C + I, one byte per (dst, src)
  int* deg;               // [npad]   in-degree (without the self loop)
  float* dinv;            // [npad]   (1 + in-degree)^-1/2, one precise division + square root per ROW (in the accumulator
                          //          epilogues they would be 32 per lane: ~1 300 VALU instructions per graph and wave)
  float* red;             // [MW * 2 * DD]   pooling combine: in the count matrix's bytes (dead by then), so that a
                          //          128-node graph with a 64-wide input still leaves room for a second workgroup per CU
};

__host__ __device__ inline size_t dense_lds_bytes(int npad, int wl_k) {
  size_t b = (size_t)npad * HS * 4;
  b += (size_t)3 * DD * (wl_k + WPAD) * 2;
  b = (b + 15) / 16 * 16;
  size_t c = (size_t)npad * dense_cb(npad);
  if (c < (size_t)MW * 2 * DD * 4) c = (size_t)MW * 2 * DD * 4;
  b += (c + 15) / 16 * 16;
  b += (size_t)(npad + 3) / 4 * 16 * 2;
  return b + 64;
}

__device__ __forceinline__ DenseLds dense_carve(char* base, int npad, int wl_k) {
  DenseLds L;
  unsigned off = 0;                       // (integer offsets: see carve())
  L.t0 = reinterpret_cast<float*>(base);
  off += (unsigned)npad * HS * 4;
  L.wl = reinterpret_cast<short*>(base + off);
  off += 3u * DD * (wl_k + WPAD) * 2;
  off = (off + 15u) / 16u * 16u;
  L.cnt = reinterpret_cast<unsigned*>(base + off);
  L.red = reinterpret_cast<float*>(base + off);
  unsigned c = (unsigned)npad * dense_cb(npad);
  if (c < (unsigned)MW * 2 * DD * 4) c = (unsigned)MW * 2 * DD * 4;
  off += (c + 15u) / 16u * 16u;
  L.deg = reinterpret_cast<int*>(base + off);
  off += (unsigned)(npad + 3) / 4 * 16;
  L.dinv = reinterpret_cast<float*>(base + off);
  return L;
}

// A graph's x rows in registers, requested a whole graph AHEAD (loads only: unconditional, clamped addresses) and written
// to the LDS tile at the start of the graph's own iteration.  Element (row, k) of the padded [128][KPAD] tile per slot:
// VEC (F == KPAD, 16-byte aligned rows): one float4 per slot; otherwise one dword (F = 25: 78 % of the slots are real).
template <int KPAD, bool VEC>
struct DenseRows {
  static constexpr int NV = DENSE_MAX_NODES * KPAD / 4 / MT;     // float4 slots per thread (VEC)
  static constexpr int NS = DENSE_MAX_NODES * KPAD / MT;         // dword slots per thread
  float4 v4[VEC ? NV : 1];
  float v1[VEC ? 1 : NS];
  // (no branch around a load: hipcc ends every guarded load with its own s_waitcnt vmcnt(0), which would turn this
  //  prefetch into an exposed HBM round trip)
  __device__ __forceinline__ void load(const float* __restrict__ g, int F, const GraphInfo& gi) {
    const int nlast = gi.n > 0 ? gi.n - 1 : 0;
    const float* base = g + (size_t)(gi.n > 0 ? gi.nbase : 0) * F;       // (an empty graph at the end of the batch has nbase == N)
    if constexpr (VEC) {
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        const int e = threadIdx.x + j * MT, row = e / (KPAD / 4), c4 = e % (KPAD / 4);
        v4[j] = *reinterpret_cast<const float4*>(base + (size_t)(row < gi.n ? row : nlast) * F + 4 * c4);
      }
    } else {
#pragma unroll
      for (int j = 0; j < NS; ++j) {
        const int e = threadIdx.x + j * MT, row = e / KPAD, k = e % KPAD;
        v1[j] = base[(size_t)(row < gi.n ? row : nlast) * F + (k < F ? k : F - 1)];
      }
    }
  }
  __device__ __forceinline__ void write(float* t, int F, const GraphInfo& gi) const {
    const int nrows = gi.nblk * 32;
    if constexpr (VEC) {
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        const int e = threadIdx.x + j * MT, row = e / (KPAD / 4), c4 = e % (KPAD / 4);
        if (row < nrows) *reinterpret_cast<float4*>(t + row * HS + 4 * c4) = row < gi.n ? v4[j] : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    } else {
#pragma unroll
      for (int j = 0; j < NS; ++j) {
        const int e = threadIdx.x + j * MT, row = e / KPAD, k = e % KPAD;
        if (row < nrows) t[row * HS + k] = (row < gi.n && k < F) ? v1[j] : 0.f;
      }
    }
  }
};

template <int KPAD, bool POOL, bool VEC>
__global__ __launch_bounds__(MT, 2) void k_mid_dense_fwd(const float* __restrict__ x, int F, const float* __restrict__ W,
                                                         const float* __restrict__ bias, const int64_t* __restrict__ ei,
                                                         int64_t E, const int32_t* __restrict__ graph_ptr,
                                                         const int32_t* __restrict__ edge_ptr, int B, int npad, int emax,
                                                         float slope, int apply_act, float* __restrict__ out, int ldo, int coff,
                                                         float* __restrict__ emb, int32_t* __restrict__ status) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const DenseLds L = dense_carve(smem, npad, KPAD);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int CB = dense_cb(npad);
  const int rb = wave >> 1, cb = wave & 1;             // this wave's 32 x 32 block of H' and of the output
  constexpr int ld = KPAD + WPAD, plane = DD * ld;
  stage_weight_split<false, MT, DD, KPAD>(L.wl, W, DD, F);
  const float bcol = bias[cb * 32 + r];
  const float slope_eff = apply_act ? slope : 1.0f;
  int dstamp_it = 0;
  (void)dstamp_it;

  GraphInfo gi;
  EdgeRegs er;
  DenseRows<KPAD, VEC> xr;
  {   // (the grid never exceeds B: every workgroup has a first graph)
    gi = graph_info(blockIdx.x < (unsigned)B ? blockIdx.x : B - 1, graph_ptr, edge_ptr, npad, emax, status);
    er.load(gi, ei, E);
    xr.load(x, F, gi);
  }
  __syncthreads();                                     // the weight image is staged
  for (int g = blockIdx.x; g < B; g += gridDim.x) {
    DSTAMP(0);
    const GraphInfo gcur = gi;
    const int nrows = gcur.nblk * 32;
    // ---- this graph's x rows (requested a graph ago) -> tile; count matrix and in-degree counters cleared
    xr.write(L.t0, F, gcur);
    {
      uint4* c4 = reinterpret_cast<uint4*>(L.cnt);
      const int n16 = nrows * CB / 16;                  // (CB is a multiple of 8, nrows of 32: whole uint4)
      for (int i = tid; i < n16; i += MT) c4[i] = make_uint4(0u, 0u, 0u, 0u);
      for (int i = tid; i < nrows; i += MT) L.deg[i] = 0;
    }
    __syncthreads();
    DSTAMP(1);
    // ---- edges (requested a graph ago) -> C[dst][src] += 1 (a byte of its dword), in-degree; the unit self loop
    {
      bool bad = false;
#pragma unroll
      for (int j = 0; j < EPT; ++j) {
        const int e = tid + j * MT;
        if (e < gcur.ne) {
          const long long sv = er.s[j], dv = er.d[j];
          const unsigned sl = (unsigned)((int)sv - gcur.nbase), dl = (unsigned)((int)dv - gcur.nbase);
          const bool ok = sl < (unsigned)gcur.n && dl < (unsigned)gcur.n && (sv >> 31) == 0 && (dv >> 31) == 0;
          bad |= !ok;
          if (ok && sl != dl) {          // an explicit (i, i) edge collapses into the unit self loop (PyG add_remaining_self_loops)
            const unsigned at = dl * CB + sl;
            atomicAdd(&L.cnt[at >> 2], 1u << (8 * (at & 3)));
            atomicAdd(&L.deg[dl], 1);
          }
        }
      }
      if (tid < gcur.n) {
        const unsigned at = (unsigned)tid * CB + tid;
        atomicAdd(&L.cnt[at >> 2], 1u << (8 * (at & 3)));
      }
      if (__ballot(bad) != 0ull && lane == 0) atomicOr(status, HCG_STATUS_EDGE_UNGROUPED);   // edge leaves its graph: ignored
    }
    __syncthreads();
    DSTAMP(2);

    // ---- H block (rb, cb) = X[rb] W[cb]^T on the matrix cores: all eight waves (one wave per 32-row block left five of
    //      eight idle on an 87-node graph); dinv of the block's rows by its cb == 0 wave, one lane per row
    const bool have_blk = rb < gcur.nblk;
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    if (have_blk) {
      if (cb == 0 && lane < 32) {
        const int row = rb * 32 + lane, dg = L.deg[row];
        L.dinv[row] = row < gcur.n ? 1.0f / sqrtf(1.0f + (float)dg) : 0.f;
        if (dg > 254) atomicOr(status, HCG_STATUS_SHAPE_LIMIT);      // a count byte may have wrapped
      }
      const float* blk = L.t0 + rb * 32 * HS;
#pragma unroll
      for (int s2 = 0; s2 < KPAD / 16; ++s2) {
        const float4 a0 = *reinterpret_cast<const float4*>(blk + r * HS + 16 * s2 + 8 * h);
        const float4 a1 = *reinterpret_cast<const float4*>(blk + r * HS + 16 * s2 + 8 * h + 4);
        const float xa[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
        const Split3 A = split3(xa);
        const short* w0 = L.wl + (cb * 32 + r) * ld + 16 * s2 + 8 * h;
        mfma_split(acc, A, *reinterpret_cast<const bf16x8*>(w0), *reinterpret_cast<const bf16x8*>(w0 + plane),
                   *reinterpret_cast<const bf16x8*>(w0 + 2 * plane));
      }
      mfma_results_fence(acc);
    }
    DSTAMP(3);
    __syncthreads();                                   // every wave has read its x block: H' may overwrite the tile
    if (have_blk) {
#pragma unroll
      for (int i = 0; i < 16; ++i) L.t0[(rb * 32 + krow(i, h)) * HS + cb * 32 + r] = acc[i] * L.dinv[rb * 32 + krow(i, h)];
    }
    __syncthreads();
    DSTAMP(4);

    // the NEXT graph's scalars, edges and x rows are requested here: they land while this graph is aggregated and stored
    // (unconditional: past the last graph of this workgroup the last graph of the batch is requested again and dropped)
    {
      const int gn = g + (int)gridDim.x < B ? g + (int)gridDim.x : B - 1;
      gi = graph_info(gn, graph_ptr, edge_ptr, npad, emax, status);
      er.load(gi, ei, E);
      xr.load(x, F, gi);
    }
    DSTAMP(9);

    // ---- Y block (rb, cb) = sum_kb C[rb][kb] H'[kb][cb]; out = LeakyReLU(dinv_i Y_i + b) in the accumulators
    f32x16 y;
#pragma unroll
    for (int i = 0; i < 16; ++i) y[i] = 0.f;
    float pm = -INFINITY, ps = 0.f;
    if (have_blk) {
      const unsigned char* crow = reinterpret_cast<const unsigned char*>(L.cnt) + (size_t)(rb * 32 + r) * CB + 8 * h;
      const float* hcol = L.t0 + (8 * h) * HS + cb * 32 + r;
      for (int ks = 0; ks < 2 * gcur.nblk; ++ks) {
        const uint2 cw = *reinterpret_cast<const uint2*>(crow + 16 * ks);          // counts of sources 16 ks + 8 h + 0..7
        float hb[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) hb[j] = hcol[(16 * ks + j) * HS];
        u32x4 u;
        u[0] = pk_bf16((float)(cw.x & 0xffu), (float)((cw.x >> 8) & 0xffu));
        u[1] = pk_bf16((float)((cw.x >> 16) & 0xffu), (float)(cw.x >> 24));
        u[2] = pk_bf16((float)(cw.y & 0xffu), (float)((cw.y >> 8) & 0xffu));
        u[3] = pk_bf16((float)((cw.y >> 16) & 0xffu), (float)(cw.y >> 24));
        mfma_exact_a(y, __builtin_bit_cast(bf16x8, u), split3(hb));
      }
      DSTAMP(10);
      mfma_results_fence(y);
      // epilogue straight out of the accumulators: lane = output column, 16 rows in registers (a 128-byte row segment per
      // half-wave and store); the pooling partials of the block fall out of the same registers
      float* orow = out + (size_t)gcur.nbase * ldo + coff + cb * 32 + r;
      float dv[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) dv[i] = L.dinv[rb * 32 + krow(i, h)];      // (all 16 LDS reads in front of the guarded stores)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int row = rb * 32 + krow(i, h);
        float v = fmaf(dv[i], y[i], bcol);
        v = fmaxf(v, slope_eff * v);             // LeakyReLU as max(v, slope*v): exact for 0 <= slope <= 1; no activation: slope_eff = 1
        if (row < gcur.n) {
          orow[(size_t)row * ldo] = v;
          if (POOL) { pm = fmaxf(pm, v); ps += v; }
        }
      }
    }
    DSTAMP(5);
    if (POOL) {   // block partials: lane halves (xor 32) -> LDS [rb][64 columns] -> fixed order over the row blocks
      pm = fmaxf(pm, __shfl_xor(pm, 32, 64));
      ps += __shfl_xor(ps, 32, 64);
      __syncthreads();                           // (the count matrix, whose bytes `red` shares, has been read by every wave)
      if (have_blk && h == 0) {
        L.red[rb * 2 * DD + cb * 32 + r] = pm;
        L.red[rb * 2 * DD + DD + cb * 32 + r] = ps;
      }
      __syncthreads();
      if (tid < DD) {
        float m = -INFINITY, sm = 0.f;
        for (int b2 = 0; b2 < gcur.nblk; ++b2) {          // fixed order over the row blocks
          m = fmaxf(m, L.red[b2 * 2 * DD + tid]);
          sm += L.red[b2 * 2 * DD + DD + tid];
        }
        if (gcur.n <= 0) m = 0.f;
        emb[(size_t)g * 2 * ldo + coff + tid] = m;                                   // [max | mean], each ldo wide
        emb[(size_t)g * 2 * ldo + ldo + coff + tid] = sm / (float)(gcur.n > 0 ? gcur.n : 1);
      }
    }
    DSTAMP(7);
    __syncthreads();   // the tile, the counts and the combine scratch are free for the next graph
    DSTAMP(8);
#ifdef HCG_DENSE_STAMP
    ++dstamp_it;
#endif
  }
}


static int launch_dense_fwd(const float* x, const float* W, const float* b, const int64_t* edge_index, int64_t E,
                            const int32_t* graph_ptr, const int32_t* edge_ptr, int B, int F, int D, int max_nodes, int max_edges,
                            float slope, int apply_act, float* out, float* emb, int32_t* status, hipStream_t stream) {
  const int npad = pad32(max_nodes), emax = pad8(max_edges);
  const int kpad = F <= 32 ? 32 : 64;
  if (npad > DENSE_MAX_NODES || F > 64 || D != DD) return HCG_ERR_UNSUPPORTED;
  const size_t lds = dense_lds_bytes(npad, kpad);
  const dim3 grid(mid_grid(B, wgs_per_cu(lds))), blk(MT);
  const int coff = 0;
#define LAUNCH_DENSE_FWD(KP, PL, VC)                                                                                       \
  do {                                                                                                                     \
    auto kfn = k_mid_dense_fwd<KP, PL, VC>;                                                                                \
    hipError_t e = allow_big_lds<k_mid_dense_fwd<KP, PL, VC>>();                                                           \
    if (e != hipSuccess) return hcg_hip_err(e);                                                                            \
    hipLaunchKernelGGL(kfn, grid, blk, lds, stream, x, (int)F, W, b, edge_index, E, graph_ptr, edge_ptr, (int)B, npad,      \
                       emax, slope, apply_act, out, (int)D, coff, emb, status);                                            \
  } while (0)
  const bool vec = F == kpad && ((uintptr_t)x % 16 == 0);
  if (kpad == 32) {
    if (vec) { if (emb) LAUNCH_DENSE_FWD(32, true, true); else LAUNCH_DENSE_FWD(32, false, true); }
    else     { if (emb) LAUNCH_DENSE_FWD(32, true, false); else LAUNCH_DENSE_FWD(32, false, false); }
  } else {
    if (vec) { if (emb) LAUNCH_DENSE_FWD(64, true, true); else LAUNCH_DENSE_FWD(64, false, true); }
    else     { if (emb) LAUNCH_DENSE_FWD(64, true, false); else LAUNCH_DENSE_FWD(64, false, false); }
  }
#undef LAUNCH_DENSE_FWD
  return hcg_hip_err(hipGetLastError());
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

int main() {
  const int B = 4096, F = 25, D = 64;
  std::mt19937 rng(1);
  std::vector<int> gp(B + 1), ep(B + 1);
  std::vector<long long> src, dst;
  int N = 0;
  for (int g = 0; g < B; ++g) {
    const int n = 57 + (int)(rng() % 61);           // 57 .. 117 atoms
    gp[g] = N; ep[g] = (int)src.size();
    auto bond = [&](int i, int j) { src.push_back(N + i); dst.push_back(N + j); src.push_back(N + j); dst.push_back(N + i); };
    for (int i = 1; i < n; ++i) bond(i, (int)(rng() % i) < i - 3 ? i - 1 : (int)(rng() % i));
    for (int c = 0; c < 4; ++c) bond((int)(rng() % (n / 2)), n / 2 + (int)(rng() % (n / 2)));
    N += n;
  }
  gp[B] = N; ep[B] = (int)src.size();
  const int E = (int)src.size();
  std::vector<long long> ei(2 * (size_t)E);
  for (int e = 0; e < E; ++e) { ei[e] = src[e]; ei[E + e] = dst[e]; }
  std::vector<float> x((size_t)N * F), W(D * F), bias(D, 0.1f);
  for (auto& v : x) v = (float)(rng() % 2000) / 1000.f - 1.f;
  for (auto& v : W) v = (float)(rng() % 2000) / 8000.f - 0.125f;
  printf("N %d E %d\n", N, E);
  float *dx, *dW, *db, *dout, *demb; long long* dei; int *dgp, *dep, *dstatus;
  CK(hipMalloc(&dx, x.size() * 4)); CK(hipMalloc(&dW, W.size() * 4)); CK(hipMalloc(&db, D * 4));
  CK(hipMalloc(&dout, (size_t)N * D * 4)); CK(hipMalloc(&demb, (size_t)B * 2 * D * 4));
  CK(hipMalloc(&dei, ei.size() * 8)); CK(hipMalloc(&dgp, (B + 1) * 4)); CK(hipMalloc(&dep, (B + 1) * 4)); CK(hipMalloc(&dstatus, 16));
  CK(hipMemcpy(dx, x.data(), x.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dW, W.data(), W.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(db, bias.data(), D * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dei, ei.data(), ei.size() * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(dgp, gp.data(), (B + 1) * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dep, ep.data(), (B + 1) * 4, hipMemcpyHostToDevice));
  CK(hipMemset(dstatus, 0, 16));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int it = 0; it < 25; ++it) {
    if (it == 5) { CK(hipDeviceSynchronize()); CK(hipEventRecord(e0, 0)); }
    int rc = launch_dense_fwd(dx, dW, db, (const int64_t*)dei, E, dgp, dep, B, F, D, 117, 300, 0.01f, 1, dout, nullptr, dstatus, 0);
    if (rc) { printf("rc %d\n", rc); return 1; }
  }
  CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  printf("dense forward (F = 25, no pooling): %.2f us per launch\n", ms * 1000.f / 20);
#ifdef HCG_DENSE_STAMP
  static unsigned long long st[4][MW][4][12];
  CK(hipMemcpyFromSymbol(st, HIP_SYMBOL(g_dense_stamp), sizeof(st)));
  const int idx[] = {0, 1, 2, 3, 4, 9, 10, 5, 7, 8};
  const char* nm[] = {"", "rows -> LDS, zero, barrier", "counts + barrier", "GEMM (8 waves)", "barrier, H' write, barrier", "prefetch issue",
                      "aggregate loop", "fence + epilogue stores", "pool", "end barrier"};
  for (int w : {0, 2, 5, 7})
    for (int it = 0; it < 3; ++it) {
      printf("block 0 wave %d graph %d (core clocks):", w, it);
      for (int i = 1; i < 10; ++i) printf(" %s %llu |", nm[i], st[0][w][it][idx[i]] - st[0][w][it][idx[i - 1]]);
      printf(" total %llu\n", st[0][w][it][8] - st[0][w][it][0]);
    }
#endif
  return 0;
}
