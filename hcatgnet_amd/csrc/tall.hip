// Wide layers over LARGE graphs (embedding_dim 128, ~200-atom ligands: BASELINE configs[4]): the layer as a DENSE
// row-streaming transform over all nodes of the batch + a per-graph segmented sum, instead of one graph per workgroup.
//
// Why a second family beside mid.hip.  A 200-node x 128-d graph does not fit a workgroup's LDS with its weights (x tile
// 102 KB + H' tile 102 KB + the pre-split weight image 98 KB of 160 KB), so mid.hip runs such a layer as two 64-column
// launches of ONE workgroup per CU whose phases (stage x, CSR, MFMA, segmented sum, store) run back to back: ~10 us per
// graph and half, 16-18 % of the step's HBM roofline (profiles/r02_c_*).  Here the dense part has no graph structure at
// all, so it streams: every wave owns 32 consecutive node rows, keeps the whole 128 x 128 weight image in LDS, reads its
// A fragments straight from global memory (the next chunk's loads in flight under the current chunk's MFMAs) and never
// meets a barrier after the prologue.  The graph part keeps only the CSR in LDS (~10 KB: 8+ workgroups per CU), gathers
// neighbour rows from L2 (a graph's rows are touched ~3x within microseconds) and carries the fused epilogues:
//
//   forward   H  = X W^T                              k_tall_mm   (split-bf16 MFMAs, f32 accuracy: split_mfma.h)
//             out = LeakyReLU(Ahat H + b), [max, mean] pool       k_seg_fwd
//   backward  dH = Ahat^T (dA (.) leaky'(A)),  db slabs           k_seg_bwd  (dA = dout, or the pooled gradient expanded on chip)
//             dX = dH W  (optionally premasked with leaky'(X))    k_tall_mm  (transposed image)
//             dW slabs = dH^T X                                   k_tall_dw
//
// Arithmetic per element is the one of mid.hip (H' = dinv . H, self term first, neighbours by ascending id, one scale by
// dinv_i at the end), so both families agree to f32 rounding of the GEMM only.  The price is HBM traffic: H and dH make a
// round trip (they mostly hit the 256 MB Infinity Cache: written and read back within ~100 us).
// Reference: the same PyG GCNConv call sites as mid.hip (model/gcn.py:58-63), `loss.backward()` utils/utils_model.py:65.
#include "common.h"
#include "split_mfma.h"

namespace {

// =====================================================================================================
// dense part 1: out[N x NO] = A[N x K] * B   (B's pre-split image resident in LDS; one 32-row block per wave iteration)
// =====================================================================================================
constexpr int TW = 8;                 // waves per workgroup: two per SIMD, 256 VGPRs each
constexpr int TT = TW * 64;

// TRANS = false: B[k][n] = W[n][k]  (H = X W^T;  W is [NO x K] row-major = the layer's weight)
// TRANS = true : B[k][n] = W[k][n]  (dX = dH W;  W is [K x NO] row-major = the same weight)
// KP / NOB * 32: K and NO padded to the image extent; lda / ldo: the real row lengths of A / out (multiples of 4).
template <int KP, int NOB, bool TRANS, bool PREMASK>
__global__ __launch_bounds__(TT, 1) void k_tall_mm(const float* __restrict__ A, int lda, const float* __restrict__ W, int wrows,
                                                   int wcols, float* __restrict__ out, int ldo, const float* __restrict__ xmask,
                                                   float slope, int N) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  short* wl = reinterpret_cast<short*>(smem);
  constexpr int ROWS = NOB * 32, ld = KP + WPAD, plane = ROWS * ld;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  stage_weight_split<TRANS, TT, ROWS, KP>(wl, W, wrows, wcols);
  __syncthreads();

  constexpr int KS = KP / 16;                    // k-steps of a row block
  constexpr int CK = KS >= 4 ? 4 : KS;           // k-steps per chunk: 32 A values per lane in flight
  constexpr int NCH = KS / CK;
  const int nrb = (N + 31) / 32;
  const int stride = gridDim.x * TW;

  // chunk c of row block rb: this lane's 8 consecutive k of every k-step (two float4; clamped, never guarded: columns
  // past lda meet zero image columns, rows past N are not stored)
  auto load_chunk = [&](float (&a)[CK][8], int rb, int c) {
    int row = rb * 32 + r;
    if (row > N - 1) row = N - 1;
    const float* base = A + (size_t)row * lda;
#pragma unroll
    for (int s = 0; s < CK; ++s) {
      int k0 = (c * CK + s) * 16 + 8 * h, k1 = k0 + 4;
      if (k0 > lda - 4) k0 = lda - 4;
      if (k1 > lda - 4) k1 = lda - 4;
      const float4 v0 = *reinterpret_cast<const float4*>(base + k0);
      const float4 v1 = *reinterpret_cast<const float4*>(base + k1);
      a[s][0] = v0.x; a[s][1] = v0.y; a[s][2] = v0.z; a[s][3] = v0.w;
      a[s][4] = v1.x; a[s][5] = v1.y; a[s][6] = v1.z; a[s][7] = v1.w;
    }
  };

  int rb = blockIdx.x * TW + wave;
  float cur[CK][8], nxt[CK][8];
  if (rb < nrb) load_chunk(cur, rb, 0);
  for (; rb < nrb; rb += stride) {
    f32x16 acc[NOB];
#pragma unroll
    for (int nb = 0; nb < NOB; ++nb)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[nb][i] = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      // the next chunk (of this row block, or the first of the wave's next one) is requested before this one is used
      if (c + 1 < NCH) load_chunk(nxt, rb, c + 1);
      else if (rb + stride < nrb) load_chunk(nxt, rb + stride, 0);
#pragma unroll
      for (int s = 0; s < CK; ++s) {
        const Split3 As = split3(cur[s]);
#pragma unroll
        for (int nb = 0; nb < NOB; ++nb) {
          const short* w0 = wl + (nb * 32 + r) * ld + (c * CK + s) * 16 + 8 * h;
          mfma_split(acc[nb], As, *reinterpret_cast<const bf16x8*>(w0), *reinterpret_cast<const bf16x8*>(w0 + plane),
                     *reinterpret_cast<const bf16x8*>(w0 + 2 * plane));
        }
      }
#pragma unroll
      for (int s = 0; s < CK; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) cur[s][j] = nxt[s][j];
    }
#pragma unroll
    for (int nb = 0; nb < NOB; ++nb) mfma_results_fence(acc[nb]);
    const int row0 = rb * 32;
#pragma unroll
    for (int nb = 0; nb < NOB; ++nb) {
      const int col = nb * 32 + r;
      const int colc = col < ldo ? col : ldo - 1;
      if (PREMASK) {      // dx handed down already multiplied by leaky'(x) of the layer below (16 loads together, clamped)
        float xm[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          int row = row0 + krow(i, h);
          if (row > N - 1) row = N - 1;
          xm[i] = xmask[(size_t)row * ldo + colc];
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[nb][i] *= hcg_leaky_grad(xm[i], slope);
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int row = row0 + krow(i, h);
        if (row < N && col < ldo) out[(size_t)row * ldo + col] = acc[nb][i];
      }
    }
  }
}

// =====================================================================================================
// dense part 2: dW[D x F] = dH^T X over all nodes, split over workgroups by node rows -> one slab per workgroup
// =====================================================================================================
// A workgroup walks 64-row tiles of dH [N x D] and X [N x F] (both staged row-contiguous into LDS, the next tile's loads in
// flight under this tile's MFMAs); the D/32 x FP/32 output blocks are spread over the 8 waves (16 blocks: two per wave
// sharing the dH fragment; fewer blocks than waves: the waves of a block take alternate k-steps and meet in LDS at the
// end, fixed order).  Both operands are read down LDS columns (K = nodes).
template <int DB, int NBF>
__global__ __launch_bounds__(TT, 4) void k_tall_dw(const float* __restrict__ Z, const float* __restrict__ X, int F,
                                                   float* __restrict__ slabs, int N) {
  constexpr int D = DB * 32, FP = NBF * 32, TR = 64;
  constexpr int ZS = D + 4, XS = FP + 4;
  constexpr int NTILE = DB * NBF, TPW = NTILE >= TW ? NTILE / TW : 1, KPARTS = NTILE >= TW ? 1 : TW / NTILE;
  static_assert(NTILE * KPARTS == TW * TPW, "block -> wave map");
  static_assert(TPW == 1 || NBF % TPW == 0, "a wave's blocks share the dH fragment");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* zt = reinterpret_cast<float*>(smem);          // [TR][ZS]
  float* xt = zt + TR * ZS;                            // [TR][XS]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int blk0 = (wave % (NTILE / TPW)) * TPW, part = wave / (NTILE / TPW);
  const int mbd = blk0 / NBF, nbf0 = blk0 % NBF;
  constexpr int ZPT = TR * D / 4 / TT, XPT = (TR * FP / 4 + TT - 1) / TT;     // float4 per thread and tile
  const int ntiles = (N + TR - 1) / TR;

  f32x16 dw[TPW];
#pragma unroll
  for (int j = 0; j < TPW; ++j)
#pragma unroll
    for (int i = 0; i < 16; ++i) dw[j][i] = 0.f;

  float4 zv[ZPT], xv[XPT];
  auto load_tile = [&](int t) {
    const int row0 = t * TR;
#pragma unroll
    for (int j = 0; j < ZPT; ++j) {
      const int idx = tid + j * TT, row = idx / (D / 4), c4 = idx - row * (D / 4);
      int gr = row0 + row;
      if (gr > N - 1) gr = N - 1;
      zv[j] = *reinterpret_cast<const float4*>(Z + (size_t)gr * D + 4 * c4);
    }
#pragma unroll
    for (int j = 0; j < XPT; ++j) {
      const int idx = tid + j * TT, row = (idx / (FP / 4)) % TR, c4 = idx % (FP / 4);
      int gr = row0 + row, c = 4 * c4;
      if (gr > N - 1) gr = N - 1;
      if (c > F - 4) c = F - 4;
      xv[j] = *reinterpret_cast<const float4*>(X + (size_t)gr * F + c);
    }
  };
  auto store_tile = [&](int t) {                       // rows past N and columns past F become zeros (they are summed)
    const int row0 = t * TR;
#pragma unroll
    for (int j = 0; j < ZPT; ++j) {
      const int idx = tid + j * TT, row = idx / (D / 4), c4 = idx - row * (D / 4);
      *reinterpret_cast<float4*>(zt + row * ZS + 4 * c4) = row0 + row < N ? zv[j] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int j = 0; j < XPT; ++j) {
      const int idx = tid + j * TT, row = idx / (FP / 4), c4 = idx - row * (FP / 4);
      if (row < TR)
        *reinterpret_cast<float4*>(xt + row * XS + 4 * c4) = (row0 + row < N && 4 * c4 < F) ? xv[j] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };

  int t = blockIdx.x;
  if (t < ntiles) load_tile(t);
  for (; t < ntiles; t += gridDim.x) {
    __syncthreads();                                   // the previous tile's readers are done
    store_tile(t);
    __syncthreads();
    if (t + (int)gridDim.x < ntiles) load_tile(t + gridDim.x);
#pragma unroll
    for (int ks = 0; ks < TR / 16; ++ks) {
      if (KPARTS > 1 && (ks % KPARTS) != part) continue;      // (compile-time unrolled; wave-uniform)
      float av[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) av[j] = zt[(16 * ks + 8 * h + j) * ZS + mbd * 32 + r];
      const Split3 As = split3(av);
#pragma unroll
      for (int b = 0; b < TPW; ++b) {
        float bv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) bv[j] = xt[(16 * ks + 8 * h + j) * XS + (nbf0 + b) * 32 + r];
        const Split3 Bs = split3(bv);
        mfma_split(dw[b], As, Bs.p1, Bs.p2, Bs.p3);
      }
    }
  }

  // ---- this workgroup's slab: dW [D][FP]
  float* slab = slabs + (size_t)blockIdx.x * (D * FP);
#pragma unroll
  for (int b = 0; b < TPW; ++b) mfma_results_fence(dw[b]);
  if (KPARTS == 1) {
#pragma unroll
    for (int b = 0; b < TPW; ++b)
#pragma unroll
      for (int i = 0; i < 16; ++i) slab[(mbd * 32 + krow(i, h)) * FP + (nbf0 + b) * 32 + r] = dw[b][i];
  } else {
    __syncthreads();                                   // the tiles are dead: [NTILE][32 * 32] combine scratch over them
    float* comb = zt;
    static_assert(NTILE * 1024 <= TR * ZS + TR * XS || KPARTS == 1, "combine scratch");
    for (int round = 0; round < KPARTS; ++round) {
      if (part == round) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          float* c = comb + blk0 * 1024 + krow(i, h) * 32 + r;
          *c = round == 0 ? dw[0][i] : *c + dw[0][i];
        }
      }
      __syncthreads();
    }
    for (int idx = tid; idx < NTILE * 1024; idx += TT) {
      const int b = idx >> 10, rr = (idx >> 5) & 31, cc = idx & 31;
      slab[((b / NBF) * 32 + rr) * FP + (b % NBF) * 32 + cc] = comb[idx];
    }
  }
}

// =====================================================================================================
// graph part: one graph per workgroup of 4 waves; only the CSR lives in LDS, rows are gathered from global memory / L2
// =====================================================================================================
constexpr int SN = 256, SW = SN / 64;
constexpr int SEG_MAX_NODES = 224;    // (same limits as mid.hip: every batch one family takes, the other takes too)
constexpr int SEG_MAX_EDGES = 1024;
constexpr int SEPT = SEG_MAX_EDGES / SN;
constexpr int SEG_MAX_D = 128;

struct SegLds {
  int rowptr[SEG_MAX_NODES + 1 + 3];
  int cursor[SEG_MAX_NODES];
  int degin[SEG_MAX_NODES];
  float dinv[SEG_MAX_NODES];
  unsigned short col[SEG_MAX_EDGES];
  float red[SW * 2 * SEG_MAX_D];
  float bc[SEG_MAX_D];
};

struct SegGraph { int nbase, n, ebase, ne, nld; };

__device__ __forceinline__ SegGraph seg_graph(int g, const int32_t* __restrict__ graph_ptr, const int32_t* __restrict__ edge_ptr,
                                              int32_t* status) {
  SegGraph gi;
  gi.nbase = graph_ptr[g];
  gi.n = graph_ptr[g + 1] - gi.nbase;
  gi.ebase = edge_ptr[g];
  gi.ne = edge_ptr[g + 1] - gi.ebase;
  if (gi.n < 0 || gi.n > SEG_MAX_NODES || gi.ne < 0 || gi.ne > SEG_MAX_EDGES) {     // host metadata was wrong: refuse the graph
    if (threadIdx.x == 0) atomicOr(status, HCG_STATUS_SHAPE_LIMIT);
    gi.n = 0;
    gi.ne = 0;
  }
  gi.nld = gi.n > 0 ? gi.nbase : 0;     // base of clamped loads (an empty graph at the end of the batch has nbase == N)
  return gi;
}

struct SegEdges {
  long long s[SEPT], d[SEPT];
  __device__ __forceinline__ void load(const SegGraph& gi, const int64_t* __restrict__ ei, int64_t E) {
#pragma unroll
    for (int j = 0; j < SEPT; ++j) {
      const int e = threadIdx.x + j * SN;
      int64_t k = (int64_t)gi.ebase + (e < gi.ne ? e : (gi.ne > 0 ? gi.ne - 1 : 0));
      if (k > E - 1) k = E - 1;
      s[j] = ei[k];
      d[j] = ei[E + k];
    }
  }
};

// The algorithm of mid.hip's build_csr for 256 threads: in-degree -> dinv = (1 + deg_in)^-1/2, counting sort into rows
// (BY_SRC: rows = sources = the transpose), explicit (i, i) edges collapse into the unit self loop, every row sorted by id.
template <bool BY_SRC>
__device__ __forceinline__ void seg_build_csr(SegLds& L, const SegGraph& gi, const SegEdges& er, int32_t* status) {
  const int tid = threadIdx.x;
  const int n = gi.n;
  for (int i = tid; i < n; i += SN) { L.cursor[i] = 0; if (BY_SRC) L.degin[i] = 0; }
  __syncthreads();
  unsigned short es[SEPT], ed[SEPT];
  bool bad = false;
#pragma unroll
  for (int j = 0; j < SEPT; ++j) {
    const int e = tid + j * SN;
    es[j] = 0xffff;
    ed[j] = 0xffff;
    if (e < gi.ne) {
      const long long s = er.s[j], d = er.d[j];
      const unsigned sl = (unsigned)((int)s - gi.nbase), dl = (unsigned)((int)d - gi.nbase);
      const bool ok = sl < (unsigned)n && dl < (unsigned)n && (s >> 31) == 0 && (d >> 31) == 0;
      bad |= !ok;
      if (ok && sl != dl) {
        es[j] = (unsigned short)sl;
        ed[j] = (unsigned short)dl;
        atomicAdd(&L.cursor[BY_SRC ? sl : dl], 1);
        if (BY_SRC) atomicAdd(&L.degin[dl], 1);
      }
    }
  }
  if (__ballot(bad) != 0ull && (tid & 63) == 0) atomicOr(status, HCG_STATUS_EDGE_UNGROUPED);
  __syncthreads();
  if (tid < 64) {     // exclusive scan of the row sizes (<= 224 rows: 4 per lane)
    constexpr int RPL = (SEG_MAX_NODES + 63) / 64;
    int v[RPL], tot = 0;
#pragma unroll
    for (int j = 0; j < RPL; ++j) {
      const int i = tid * RPL + j;
      v[j] = i < n ? L.cursor[i] : 0;
      tot += v[j];
    }
    int incl = tot;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int t = __shfl_up(incl, off, 64);
      if (tid >= off) incl += t;
    }
    int run = incl - tot;
#pragma unroll
    for (int j = 0; j < RPL; ++j) {
      const int i = tid * RPL + j;
      if (i < n) L.rowptr[i] = run;
      run += v[j];
    }
    if (tid == 63) L.rowptr[n] = incl;
  }
  for (int i = tid; i < n; i += SN) L.dinv[i] = 1.0f / sqrtf(1.0f + (float)(BY_SRC ? L.degin[i] : L.cursor[i]));
  __syncthreads();
  for (int i = tid; i < n; i += SN) L.cursor[i] = L.rowptr[i];
  __syncthreads();
#pragma unroll
  for (int j = 0; j < SEPT; ++j) {
    if (es[j] != 0xffff) {
      const int p = atomicAdd(&L.cursor[BY_SRC ? es[j] : ed[j]], 1);
      L.col[p] = BY_SRC ? ed[j] : es[j];
    }
  }
  __syncthreads();
  for (int i = tid; i < n; i += SN) {
    const int kb = L.rowptr[i], ke = L.rowptr[i + 1], len = ke - kb;
    if (len > 1 && len <= 4) {
      unsigned a0 = L.col[kb], a1 = L.col[kb + 1], a2 = len > 2 ? L.col[kb + 2] : 0xffffu, a3 = len > 3 ? L.col[kb + 3] : 0xffffu;
      unsigned t;
      t = min(a0, a1); a1 = max(a0, a1); a0 = t;
      t = min(a2, a3); a3 = max(a2, a3); a2 = t;
      t = min(a0, a2); a2 = max(a0, a2); a0 = t;
      t = min(a1, a3); a3 = max(a1, a3); a1 = t;
      t = min(a1, a2); a2 = max(a1, a2); a1 = t;
      L.col[kb] = (unsigned short)a0;
      L.col[kb + 1] = (unsigned short)a1;
      if (len > 2) L.col[kb + 2] = (unsigned short)a2;
      if (len > 3) L.col[kb + 3] = (unsigned short)a3;
    } else if (len > 4) {
      for (int a = kb + 1; a < ke; ++a) {
        const unsigned short key = L.col[a];
        int b = a - 1;
        while (b >= kb && L.col[b] > key) { L.col[b + 1] = L.col[b]; --b; }
        L.col[b + 1] = key;
      }
    }
  }
  __syncthreads();
}

__device__ __forceinline__ float4 f4_zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ float4 f4_scale(float s, float4 v) { return make_float4(s * v.x, s * v.y, s * v.z, s * v.w); }
__device__ __forceinline__ void f4_add(float4& a, const float4 v) { a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w; }

// One row's slot of the walk: the row, its CSR range and its first four neighbours (absent ones point at the row itself)
struct SegRow {
  int row, kb, ke, c[4];
  bool valid;
  __device__ __forceinline__ void set(const SegLds& L, int row_, int n) {
    valid = row_ < n;
    row = valid ? row_ : (n > 0 ? n - 1 : 0);
    kb = valid ? L.rowptr[row] : 0;
    ke = valid ? L.rowptr[row + 1] : 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) c[j] = kb + j < ke ? L.col[kb + j] : row;
  }
};

// ---- forward: out_i = LeakyReLU(dinv_i (dinv_i H_i + sum_k dinv_k H_k) + b), [max | mean] pooling
template <int D, bool POOL>
__global__ __launch_bounds__(SN, 4) void k_seg_fwd(const float* __restrict__ H, const float* __restrict__ bias,
                                                const int64_t* __restrict__ ei, int64_t E, const int32_t* __restrict__ graph_ptr,
                                                const int32_t* __restrict__ edge_ptr, int B, float slope, int apply_act,
                                                float* __restrict__ out, float* __restrict__ emb, int32_t* __restrict__ status) {
  __shared__ SegLds L;
  constexpr int LPR = D / 4, RPW = 64 / LPR, RPP = RPW * SW, INF = 2;      // INF rows of a lane in flight
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int q = lane % LPR, rsub = lane / LPR;
  const float4 bq = *reinterpret_cast<const float4*>(bias + 4 * q);

  SegGraph gnext;
  SegEdges er;
  if ((int)blockIdx.x < B) {
    gnext = seg_graph(blockIdx.x, graph_ptr, edge_ptr, status);
    er.load(gnext, ei, E);
  }
  for (int g = blockIdx.x; g < B; g += gridDim.x) {
    const SegGraph gi = gnext;
    seg_build_csr<false>(L, gi, er, status);
    if (g + (int)gridDim.x < B) {                     // the NEXT graph's scalars and edges: in flight under this graph's rows
      gnext = seg_graph(g + gridDim.x, graph_ptr, edge_ptr, status);
      er.load(gnext, ei, E);
    }
    float4 pmax = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY), psum = f4_zero();
    const float* Hg = H + (size_t)gi.nld * D + 4 * q;
    for (int pb = 0; pb < gi.n; pb += RPP * INF) {     // (block-uniform trip count: the tail loop below votes per wave)
      const int row0 = pb + wave * RPW + rsub;
      SegRow rw[INF];
      float4 hs[INF], hv[INF][4];
#pragma unroll
      for (int u = 0; u < INF; ++u) {
        rw[u].set(L, row0 + u * RPP, gi.n);
        hs[u] = *reinterpret_cast<const float4*>(Hg + (size_t)rw[u].row * D);
#pragma unroll
        for (int j = 0; j < 4; ++j) hv[u][j] = *reinterpret_cast<const float4*>(Hg + (size_t)rw[u].c[j] * D);
      }
#pragma unroll
      for (int u = 0; u < INF; ++u) {
        const SegRow& w = rw[u];
        const float di = L.dinv[w.row];
        float4 acc = f4_scale(di, hs[u]);
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (w.kb + j < w.ke) f4_add(acc, f4_scale(L.dinv[w.c[j]], hv[u][j]));
        for (int k = w.kb + 4; __any(k < w.ke); ++k) {
          const int c = k < w.ke ? L.col[k] : w.row;
          const float4 v = *reinterpret_cast<const float4*>(Hg + (size_t)c * D);
          if (k < w.ke) f4_add(acc, f4_scale(L.dinv[c], v));
        }
        float4 y = make_float4(fmaf(di, acc.x, bq.x), fmaf(di, acc.y, bq.y), fmaf(di, acc.z, bq.z), fmaf(di, acc.w, bq.w));
        if (apply_act) { y.x = fmaxf(y.x, slope * y.x); y.y = fmaxf(y.y, slope * y.y); y.z = fmaxf(y.z, slope * y.z); y.w = fmaxf(y.w, slope * y.w); }
        if (w.valid) {
          *reinterpret_cast<float4*>(out + (size_t)(gi.nbase + w.row) * D + 4 * q) = y;
          if (POOL) {
            pmax = make_float4(fmaxf(pmax.x, y.x), fmaxf(pmax.y, y.y), fmaxf(pmax.z, y.z), fmaxf(pmax.w, y.w));
            f4_add(psum, y);
          }
        }
      }
    }
    if (POOL) {
#pragma unroll
      for (int off = LPR; off < 64; off <<= 1) {
        pmax = make_float4(fmaxf(pmax.x, __shfl_xor(pmax.x, off, 64)), fmaxf(pmax.y, __shfl_xor(pmax.y, off, 64)),
                           fmaxf(pmax.z, __shfl_xor(pmax.z, off, 64)), fmaxf(pmax.w, __shfl_xor(pmax.w, off, 64)));
        psum.x += __shfl_xor(psum.x, off, 64); psum.y += __shfl_xor(psum.y, off, 64);
        psum.z += __shfl_xor(psum.z, off, 64); psum.w += __shfl_xor(psum.w, off, 64);
      }
      if (rsub == 0) {
        *reinterpret_cast<float4*>(L.red + wave * 2 * D + 4 * q) = pmax;
        *reinterpret_cast<float4*>(L.red + wave * 2 * D + D + 4 * q) = psum;
      }
      __syncthreads();
      if (tid < D) {
        float m = -INFINITY, s = 0.f;
#pragma unroll
        for (int w = 0; w < SW; ++w) {                  // fixed order over the waves
          m = fmaxf(m, L.red[w * 2 * D + tid]);
          s += L.red[w * 2 * D + D + tid];
        }
        if (gi.n <= 0) m = 0.f;
        emb[(size_t)g * 2 * D + tid] = m;
        emb[(size_t)g * 2 * D + D + tid] = s / (float)(gi.n > 0 ? gi.n : 1);
      }
    }
    __syncthreads();   // the CSR and the combine scratch are free for the next graph
  }
}

// ---- backward: G = dA (.) leaky'(A);  db += colsum G;  dH_j = dinv_j (dinv_j G_j + sum_{k in row j of the transpose} dinv_k G_k)
// POOLG: dA is the pooled gradient expanded on chip (mean share + the max's share split evenly over ties, as
// global_max_pool's backward does through torch.max).  TWO: both dout and a_out are read per row (activation derivative
// applied here); otherwise exactly one tensor is gathered.
template <int D, bool POOLG, bool TWO>
__global__ __launch_bounds__(SN, 4) void k_seg_bwd(const float* __restrict__ dout, const float* __restrict__ demb,
                                                const float* __restrict__ emb, const float* __restrict__ a_out,
                                                const int64_t* __restrict__ ei, int64_t E, const int32_t* __restrict__ graph_ptr,
                                                const int32_t* __restrict__ edge_ptr, int B, float slope, int act_here,
                                                float* __restrict__ Z, float* __restrict__ db_slabs, int32_t* __restrict__ status) {
  __shared__ SegLds L;
  constexpr int LPR = D / 4, RPW = 64 / LPR, RPP = RPW * SW, INF = TWO ? 1 : 2;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int q = lane % LPR, rsub = lane / LPR;
  float4 dbacc = f4_zero();

  SegGraph gnext;
  SegEdges er;
  if ((int)blockIdx.x < B) {
    gnext = seg_graph(blockIdx.x, graph_ptr, edge_ptr, status);
    er.load(gnext, ei, E);
  }
  for (int g = blockIdx.x; g < B; g += gridDim.x) {
    const SegGraph gi = gnext;
    // the pooled pieces of this graph: requested before the CSR build
    float4 gmx = f4_zero(), dmx = f4_zero(), dmean = f4_zero(), share = f4_zero();
    if (POOLG) {
      const size_t eb = (size_t)g * 2 * D + 4 * q;
      gmx = *reinterpret_cast<const float4*>(emb + eb);
      dmx = *reinterpret_cast<const float4*>(demb + eb);
      dmean = *reinterpret_cast<const float4*>(demb + eb + D);
    }
    seg_build_csr<true>(L, gi, er, status);
    if (g + (int)gridDim.x < B) {
      gnext = seg_graph(g + gridDim.x, graph_ptr, edge_ptr, status);
      er.load(gnext, ei, E);
    }
    const float* Ag = a_out ? a_out + (size_t)gi.nld * D + 4 * q : nullptr;
    const float* Dg = dout ? dout + (size_t)gi.nld * D + 4 * q : nullptr;
    if (POOLG) {
      const float cntf = (float)(gi.n > 0 ? gi.n : 1);
      dmean = make_float4(dmean.x / cntf, dmean.y / cntf, dmean.z / cntf, dmean.w / cntf);
      // ties of the column maxima: one pass over the graph's rows (they stay in L2 for the gather below)
      float4 ties = f4_zero();
      constexpr int TB = 4;
      for (int pb = 0; pb < gi.n; pb += RPP * TB) {
        const int row0 = pb + wave * RPW + rsub;
        float4 a[TB];
#pragma unroll
        for (int u = 0; u < TB; ++u) {
          int row = row0 + u * RPP;
          if (row > gi.n - 1) row = gi.n > 0 ? gi.n - 1 : 0;
          a[u] = *reinterpret_cast<const float4*>(Ag + (size_t)row * D);
        }
#pragma unroll
        for (int u = 0; u < TB; ++u) {
          if (row0 + u * RPP < gi.n) {
            ties.x += (a[u].x == gmx.x); ties.y += (a[u].y == gmx.y); ties.z += (a[u].z == gmx.z); ties.w += (a[u].w == gmx.w);
          }
        }
      }
#pragma unroll
      for (int off = LPR; off < 64; off <<= 1) {
        ties.x += __shfl_xor(ties.x, off, 64); ties.y += __shfl_xor(ties.y, off, 64);
        ties.z += __shfl_xor(ties.z, off, 64); ties.w += __shfl_xor(ties.w, off, 64);
      }
      if (rsub == 0) *reinterpret_cast<float4*>(L.red + wave * D + 4 * q) = ties;
      __syncthreads();
      float4 tot = f4_zero();
#pragma unroll
      for (int w = 0; w < SW; ++w) f4_add(tot, *reinterpret_cast<const float4*>(L.red + w * D + 4 * q));   // (counts: exact)
      share = make_float4(dmx.x / fmaxf(tot.x, 1.f), dmx.y / fmaxf(tot.y, 1.f), dmx.z / fmaxf(tot.z, 1.f), dmx.w / fmaxf(tot.w, 1.f));
    }
    // G of one row from what was loaded for it
    auto grad_of = [&](const float4 d, const float4 a) {
      float4 gq;
      if (POOLG) {
        gq = make_float4(dmean.x + (a.x == gmx.x ? share.x : 0.f), dmean.y + (a.y == gmx.y ? share.y : 0.f),
                         dmean.z + (a.z == gmx.z ? share.z : 0.f), dmean.w + (a.w == gmx.w ? share.w : 0.f));
      } else {
        gq = d;
      }
      if (POOLG || TWO) {
        if (act_here) {
          gq.x *= hcg_leaky_grad(a.x, slope); gq.y *= hcg_leaky_grad(a.y, slope);
          gq.z *= hcg_leaky_grad(a.z, slope); gq.w *= hcg_leaky_grad(a.w, slope);
        }
      }
      return gq;
    };
    for (int pb = 0; pb < gi.n; pb += RPP * INF) {
      const int row0 = pb + wave * RPW + rsub;
      SegRow rw[INF];
      float4 ds[INF], as[INF], dv[INF][4], av[INF][4];
#pragma unroll
      for (int u = 0; u < INF; ++u) {
        rw[u].set(L, row0 + u * RPP, gi.n);
        if (!POOLG) ds[u] = *reinterpret_cast<const float4*>(Dg + (size_t)rw[u].row * D);
        if (POOLG || TWO) as[u] = *reinterpret_cast<const float4*>(Ag + (size_t)rw[u].row * D);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (!POOLG) dv[u][j] = *reinterpret_cast<const float4*>(Dg + (size_t)rw[u].c[j] * D);
          if (POOLG || TWO) av[u][j] = *reinterpret_cast<const float4*>(Ag + (size_t)rw[u].c[j] * D);
        }
      }
#pragma unroll
      for (int u = 0; u < INF; ++u) {
        const SegRow& w = rw[u];
        const float di = L.dinv[w.row];
        const float4 gs = grad_of(POOLG ? f4_zero() : ds[u], (POOLG || TWO) ? as[u] : f4_zero());
        if (w.valid) f4_add(dbacc, gs);
        float4 acc = f4_scale(di, gs);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (w.kb + j < w.ke) {
            const float4 gj = grad_of(POOLG ? f4_zero() : dv[u][j], (POOLG || TWO) ? av[u][j] : f4_zero());
            f4_add(acc, f4_scale(L.dinv[w.c[j]], gj));
          }
        }
        for (int k = w.kb + 4; __any(k < w.ke); ++k) {
          const int c = k < w.ke ? L.col[k] : w.row;
          float4 d2 = f4_zero(), a2 = f4_zero();
          if (!POOLG) d2 = *reinterpret_cast<const float4*>(Dg + (size_t)c * D);
          if (POOLG || TWO) a2 = *reinterpret_cast<const float4*>(Ag + (size_t)c * D);
          if (k < w.ke) f4_add(acc, f4_scale(L.dinv[c], grad_of(d2, a2)));
        }
        if (w.valid) *reinterpret_cast<float4*>(Z + (size_t)(gi.nbase + w.row) * D + 4 * q) = f4_scale(di, acc);
      }
    }
    __syncthreads();   // the CSR (and the tie scratch) are free for the next graph
  }
  // ---- this workgroup's bias-gradient slab [D]: lanes of a column -> wave (xor shuffles) -> workgroup (LDS), fixed order
#pragma unroll
  for (int off = LPR; off < 64; off <<= 1) {
    dbacc.x += __shfl_xor(dbacc.x, off, 64); dbacc.y += __shfl_xor(dbacc.y, off, 64);
    dbacc.z += __shfl_xor(dbacc.z, off, 64); dbacc.w += __shfl_xor(dbacc.w, off, 64);
  }
  if (rsub == 0) *reinterpret_cast<float4*>(L.red + wave * D + 4 * q) = dbacc;
  __syncthreads();
  if (tid < D) {
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < SW; ++w) s += L.red[w * D + tid];
    db_slabs[(size_t)blockIdx.x * D + tid] = s;
  }
}

// ---------------------------------------------------------------------------------------------------- host side
int cu_count() {
  static int cus = 0;
  if (cus > 0) return cus;
  int dev = 0, v = 0;
  cus = 256;
  if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0)
    cus = v;
  return cus;
}
int seg_grid(int64_t B) {            // 8 workgroups of 4 waves per CU at most (32 waves: what <= 64 VGPRs admit)
  int64_t g = (int64_t)cu_count() * 8;
  if (g > B) g = B;
  return g < 1 ? 1 : (int)g;
}
int mm_grid(int64_t N) {
  int64_t g = hcg_cdiv(hcg_cdiv(N, 32), TW);
  if (g > cu_count()) g = cu_count();
  return g < 1 ? 1 : (int)g;
}
constexpr int DW_TILE = 64;
int dw_grid(int64_t N) {             // two workgroups per CU; every workgroup leaves a slab, so no more than there are tiles
  int64_t g = hcg_cdiv(N, DW_TILE);
  if (g > 2 * (int64_t)cu_count()) g = 2 * (int64_t)cu_count();
  return g < 1 ? 1 : (int)g;
}
int tall_fpad(int64_t F) { return F <= 32 ? 32 : (F <= 64 ? 64 : 128); }

template <auto KFN>
hipError_t allow_lds(size_t bytes) {
  static hipError_t st = hipFuncSetAttribute((const void*)KFN, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  return st;
}

struct TallWs { float* inter; float* dw_slabs; float* db_slabs; size_t total; };
TallWs tall_carve(void* ws, int64_t N, int64_t B, int64_t F, int64_t D) {
  TallWs t;
  const size_t o1 = hcg_align_up((size_t)N * D * sizeof(float), 256);
  const size_t o2 = o1 + hcg_align_up((size_t)dw_grid(N) * D * tall_fpad(F) * sizeof(float), 256);
  const size_t o3 = o2 + hcg_align_up((size_t)seg_grid(B) * D * sizeof(float), 256);
  char* p = (char*)ws;
  t.inter = (float*)p;
  t.dw_slabs = p ? (float*)(p + o1) : nullptr;
  t.db_slabs = p ? (float*)(p + o2) : nullptr;
  t.total = o3 + 256;
  return t;
}

}  // namespace

// 1 when these kernels take the layer: D = 128 output columns (the width they were built for; D = 64 layers stay with
// the one-graph-per-workgroup kernels, which hold the whole layer in LDS), F a multiple of 4 up to 128, the graph limits
// of hcg_mid_supported
extern "C" int hcg_tall_supported(int64_t F, int64_t D, int64_t max_nodes_per_graph, int64_t max_edges_per_graph) {
  return (D == 128 && F >= 4 && F <= 128 && (F % 4) == 0 && max_nodes_per_graph >= 1 && max_nodes_per_graph <= SEG_MAX_NODES &&
          max_edges_per_graph >= 0 && max_edges_per_graph <= SEG_MAX_EDGES) ? 1 : 0;
}

extern "C" size_t hcg_tall_workspace_bytes(int64_t N, int64_t B, int64_t F, int64_t D) {
  if (N <= 0 || B <= 0 || D != 128 || F < 4 || F > 128) return 0;
  return tall_carve(nullptr, N, B, F, D).total;
}

extern "C" int hcg_tall_layer_fwd(const float* x, const float* W, const float* b, const int64_t* edge_index, int64_t E,
                                  const int32_t* graph_ptr, const int32_t* edge_ptr, int64_t N, int64_t B, int64_t F, int64_t D,
                                  int64_t max_nodes, int64_t max_edges, float slope, int apply_act, float* out, float* emb,
                                  int32_t* status, void* workspace, size_t workspace_bytes, hcg_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!hcg_tall_supported(F, D, max_nodes, max_edges)) return HCG_ERR_UNSUPPORTED;
  if (apply_act && !(slope >= 0.f && slope <= 1.f)) return HCG_ERR_UNSUPPORTED;   // LeakyReLU is evaluated as max(v, slope*v)
  if (N < 0 || B < 0 || E < 0) return HCG_ERR_INVALID_ARG;
  if (B == 0 || N == 0) return HCG_OK;
  if (!x || !W || !b || !graph_ptr || !edge_ptr || !out || !status || !workspace || (E > 0 && !edge_index)) return HCG_ERR_INVALID_ARG;
  if (workspace_bytes < hcg_tall_workspace_bytes(N, B, F, D)) return HCG_ERR_WORKSPACE;
  if (N > (int64_t)INT32_MAX / 2) return HCG_ERR_UNSUPPORTED;
  if (E == 0) { edge_index = reinterpret_cast<const int64_t*>(graph_ptr); E = 1; }
  const TallWs ws = tall_carve(workspace, N, B, F, D);
  const int fp = tall_fpad(F);
  const dim3 grid(mm_grid(N)), blk(TT);
#define LAUNCH_MM_FWD(KP)                                                                                                  \
  do {                                                                                                                     \
    const size_t lds = (size_t)3 * 128 * (KP + WPAD) * 2;                                                                  \
    hipError_t e = allow_lds<k_tall_mm<KP, 4, false, false>>(lds);                                                         \
    if (e != hipSuccess) return hcg_hip_err(e);                                                                            \
    hipLaunchKernelGGL((k_tall_mm<KP, 4, false, false>), grid, blk, lds, stream, x, (int)F, W, (int)D, (int)F, ws.inter,   \
                       (int)D, (const float*)nullptr, slope, (int)N);                                                      \
  } while (0)
  if (fp == 32) LAUNCH_MM_FWD(32); else if (fp == 64) LAUNCH_MM_FWD(64); else LAUNCH_MM_FWD(128);
#undef LAUNCH_MM_FWD
  HCG_CHECK_LAUNCH();
  const dim3 sgrid(seg_grid(B)), sblk(SN);
  if (emb)
    hipLaunchKernelGGL((k_seg_fwd<128, true>), sgrid, sblk, 0, stream, ws.inter, b, edge_index, E, graph_ptr, edge_ptr, (int)B,
                       slope, apply_act, out, emb, status);
  else
    hipLaunchKernelGGL((k_seg_fwd<128, false>), sgrid, sblk, 0, stream, ws.inter, b, edge_index, E, graph_ptr, edge_ptr, (int)B,
                       slope, apply_act, out, emb, status);
  HCG_CHECK_LAUNCH();
  return HCG_OK;
}

// dout == NULL selects the pooled form (upstream gradient = demb [B, 2D], expanded on chip with `emb` and `out`).
// apply_act: bit 0 = multiply the upstream gradient by leaky'(out); bit 1 = hand dx down already multiplied by leaky'(x).
// Leaves dW / db slabs in `workspace`: describe them with hcg_tall_reduce_jobs (two jobs) and sum with hcg_reduce_slabs.
extern "C" int hcg_tall_layer_bwd(const float* dout, const float* demb, const float* emb, const float* out, const float* x,
                                  const float* W, const int64_t* edge_index, int64_t E, const int32_t* graph_ptr,
                                  const int32_t* edge_ptr, int64_t N, int64_t B, int64_t F, int64_t D, int64_t max_nodes,
                                  int64_t max_edges, float slope, int apply_act, float* dx, int32_t* status, void* workspace,
                                  size_t workspace_bytes, hcg_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!hcg_tall_supported(F, D, max_nodes, max_edges)) return HCG_ERR_UNSUPPORTED;
  if (N <= 0 || B <= 0 || E < 0 || !W || !workspace || !x || !graph_ptr || !edge_ptr || !status || (E > 0 && !edge_index))
    return HCG_ERR_INVALID_ARG;
  if (N > (int64_t)INT32_MAX / 2) return HCG_ERR_UNSUPPORTED;
  const bool poolg = (dout == nullptr);
  if (poolg && (!demb || !emb)) return HCG_ERR_INVALID_ARG;
  if ((apply_act & ~3) || ((apply_act & 2) && !dx)) return HCG_ERR_INVALID_ARG;
  if ((poolg || (apply_act & 1)) && !out) return HCG_ERR_INVALID_ARG;
  if (workspace_bytes < hcg_tall_workspace_bytes(N, B, F, D)) return HCG_ERR_WORKSPACE;
  if (E == 0) { edge_index = reinterpret_cast<const int64_t*>(graph_ptr); E = 1; }
  const TallWs ws = tall_carve(workspace, N, B, F, D);
  const int act_here = apply_act & 1;
  const dim3 sgrid(seg_grid(B)), sblk(SN);
  if (poolg)
    hipLaunchKernelGGL((k_seg_bwd<128, true, false>), sgrid, sblk, 0, stream, dout, demb, emb, out, edge_index, E, graph_ptr,
                       edge_ptr, (int)B, slope, act_here, ws.inter, ws.db_slabs, status);
  else if (act_here)
    hipLaunchKernelGGL((k_seg_bwd<128, false, true>), sgrid, sblk, 0, stream, dout, demb, emb, out, edge_index, E, graph_ptr,
                       edge_ptr, (int)B, slope, act_here, ws.inter, ws.db_slabs, status);
  else
    hipLaunchKernelGGL((k_seg_bwd<128, false, false>), sgrid, sblk, 0, stream, dout, demb, emb, (const float*)nullptr, edge_index,
                       E, graph_ptr, edge_ptr, (int)B, slope, 0, ws.inter, ws.db_slabs, status);
  HCG_CHECK_LAUNCH();
  const int fp = tall_fpad(F);
  // dW slabs = dH^T x
  {
    const dim3 grid(dw_grid(N)), blk(TT);
#define LAUNCH_DW(NBF)                                                                                               \
  do {                                                                                                               \
    const size_t lds = (size_t)DW_TILE * ((128 + 4) + (NBF * 32 + 4)) * sizeof(float);                               \
    hipError_t e = allow_lds<k_tall_dw<4, NBF>>(lds);                                                                \
    if (e != hipSuccess) return hcg_hip_err(e);                                                                      \
    hipLaunchKernelGGL((k_tall_dw<4, NBF>), grid, blk, lds, stream, ws.inter, x, (int)F, ws.dw_slabs, (int)N);       \
  } while (0)
    if (fp == 32) LAUNCH_DW(1); else if (fp == 64) LAUNCH_DW(2); else LAUNCH_DW(4);
#undef LAUNCH_DW
    HCG_CHECK_LAUNCH();
  }
  // dx = dH W
  if (dx) {
    const dim3 grid(mm_grid(N)), blk(TT);
    const size_t lds = (size_t)3 * fp * (128 + WPAD) * 2;
    const bool pm = (apply_act & 2) != 0;
#define LAUNCH_MM_BWD(NOB, PM)                                                                                             \
  do {                                                                                                                     \
    hipError_t e = allow_lds<k_tall_mm<128, NOB, true, PM>>(lds);                                                          \
    if (e != hipSuccess) return hcg_hip_err(e);                                                                            \
    hipLaunchKernelGGL((k_tall_mm<128, NOB, true, PM>), grid, blk, lds, stream, ws.inter, (int)D, W, (int)D, (int)F, dx,   \
                       (int)F, x, slope, (int)N);                                                                          \
  } while (0)
    if (fp == 32)      { if (pm) LAUNCH_MM_BWD(1, true); else LAUNCH_MM_BWD(1, false); }
    else if (fp == 64) { if (pm) LAUNCH_MM_BWD(2, true); else LAUNCH_MM_BWD(2, false); }
    else               { if (pm) LAUNCH_MM_BWD(4, true); else LAUNCH_MM_BWD(4, false); }
#undef LAUNCH_MM_BWD
    HCG_CHECK_LAUNCH();
  }
  return HCG_OK;
}

// two jobs: job_host[0] = dW [D, F] from the k_tall_dw slabs, job_host[1] = db [D] from the k_seg_bwd slabs
extern "C" int hcg_tall_reduce_jobs(const void* workspace, size_t workspace_bytes, int64_t N, int64_t B, int64_t F, int64_t D,
                                    float* dW, float* db, hcg_reduce_job* job_host) {
  if (D != 128 || F < 4 || F > 128 || N <= 0 || B <= 0 || !dW || !db || !job_host || !workspace) return HCG_ERR_INVALID_ARG;
  if (workspace_bytes < hcg_tall_workspace_bytes(N, B, F, D)) return HCG_ERR_WORKSPACE;
  const TallWs ws = tall_carve(const_cast<void*>(workspace), N, B, F, D);
  const int fp = tall_fpad(F);
  hcg_reduce_job* j = job_host;
  j->slabs = ws.dw_slabs;
  j->nslabs = dw_grid(N);
  j->slab_floats = (int32_t)(D * fp);
  j->nseg = 1;
  j->reserved = 0;
  j->seg[0] = hcg_reduce_seg{0, (int32_t)(D * fp), fp, (int32_t)F, dW};
  j = job_host + 1;
  j->slabs = ws.db_slabs;
  j->nslabs = seg_grid(B);
  j->slab_floats = (int32_t)D;
  j->nseg = 1;
  j->reserved = 0;
  j->seg[0] = hcg_reduce_seg{0, (int32_t)D, 1, 1, db};
  return HCG_OK;
}
