// Layers cut by what needs the graph and what does not: dense row-streaming parts over ALL nodes of the batch + per-graph
// parts that keep a whole tile in LDS.  Two uses:
//   * 128-wide layers over large graphs (embedding_dim 128, ~200-atom ligands: BASELINE configs[4]), forward and backward;
//   * the BACKWARD of 64-wide layers over graphs of 65 .. 224 nodes (the reference's own regime: 56-184 atoms, F = 25 / 32).
//
// Why a second family beside mid.hip.  A 200-node x 128-d graph does not fit a workgroup's LDS with its weights (x tile
// 102 KB + H' tile 102 KB + the pre-split weight image 98 KB of 160 KB), so mid.hip runs such a layer as two 64-column
// launches of ONE workgroup per CU whose phases (stage x, CSR, MFMA, segmented sum, store) run back to back: ~10 us per
// graph and half, 16-18 % of the step's HBM roofline on C5 (profiles/r02_c_*); its backward (dY' tile, dH tile, dW and dX
// on the matrix cores, all per graph) is the same chain at 23-24 % on the reference's graph sizes.
//
//   forward   H' = dinv . (x W^T) into an LDS tile, out = LeakyReLU(Ahat H' + b), [max, mean] pool   k_split_weight + k_seg_fwd
//   backward  dH = Ahat^T (dA (.) leaky'(A)),  db slabs     k_gseg_bwd  (dA = dout, or the pooled gradient expanded on chip;
//                                                                        independent 64-column groups, 2-4 workgroups per CU)
//             dW slabs = dH^T X                             k_tall_dw   (no graph structure: 64-row tiles of all nodes)
//             dX = dH W  (optionally premasked)             k_tall_mm   (no graph structure: 32-row blocks, image resident)
//
// All contractions are split-bf16 MFMAs at f32 accuracy (split_mfma.h).  Arithmetic per element is the one of mid.hip
// (H' = dinv . H, self term first, neighbours by ascending id, one scale by dinv_i at the end), so both families agree to
// f32 rounding of the GEMM only; every reduction has a fixed order (run-to-run bitwise).  The price of the cut is HBM
// traffic in the backward: dH makes a round trip (written once, read by both dense kernels).  What bounds each kernel and
// what was tried: DESIGN.md 4.2c.
// Reference: the same PyG GCNConv call sites as mid.hip (model/gcn.py:58-63), `loss.backward()` utils/utils_model.py:65.
#include <type_traits>

#include "common.h"
#include "split_mfma.h"

namespace {

// =====================================================================================================
// dense part 1: out[N x NO] = A[N x K] * B   (B's pre-split image resident in LDS)
// =====================================================================================================
constexpr int TW = 16;                // waves per workgroup: the whole CU, four per SIMD, 128 VGPRs each
constexpr int TT = TW * 64;

// TRANS = false: B[k][n] = W[n][k]  (H = X W^T;  W is [NO x K] row-major = the layer's weight; kept for tools/probe_tall.hip:
//                the forward fuses this product into k_seg_fwd)
// TRANS = true : B[k][n] = W[k][n]  (dX = dH W;  W is [K x NO] row-major = the same weight)
// KP / NOB * 32: K and NO padded to the image extent; lda / ldo: the real row lengths of A / out (multiples of 4).
// A wave owns one 32-row block x NBW 32-column blocks (NOB = 4: two waves share a row block, each re-reading its A rows
// -- the second read hits L1 / L2); its A fragments come straight from global memory, a chunk of CK k-steps ahead.
// (First version: 8 waves x 4 column blocks at 244 VGPRs: 68.6 us for C5's layer, three times its MFMA time -- two waves
// per SIMD do not cover the row loads and the stores.)
template <int KP, int NOB, bool TRANS, bool PREMASK>
__global__ __launch_bounds__(TT, 4) void k_tall_mm(const float* __restrict__ A, int lda, const float* __restrict__ W, int wrows,
                                                   int wcols, float* __restrict__ out, int ldo, const float* __restrict__ xmask,
                                                   float slope, int N) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  short* wl = reinterpret_cast<short*>(smem);
  constexpr int ROWS = NOB * 32, ld = KP + WPAD, plane = ROWS * ld;
  constexpr int CS = NOB >= 2 ? 2 : 1, NBW = NOB / CS;          // waves per row block / column blocks per wave
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  stage_weight_split<TRANS, TT, ROWS, KP>(wl, W, wrows, wcols);
  __syncthreads();

  constexpr int KS = KP / 16;                    // k-steps of a row block
  constexpr int CK = KS >= 2 ? 2 : KS;           // k-steps per chunk: 16 A values per lane in flight (4: 20-84 spilled registers)
  constexpr int NCH = KS / CK;
  const int nrb = (N + 31) / 32;
  const int stride = gridDim.x * (TW / CS);
  const int nb0 = (wave % CS) * NBW;

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

  int rb = blockIdx.x * (TW / CS) + wave / CS;
  float cur[CK][8], nxt[CK][8];
  if (rb < nrb) load_chunk(cur, rb, 0);
  for (; rb < nrb; rb += stride) {
    f32x16 acc[NBW];
#pragma unroll
    for (int nb = 0; nb < NBW; ++nb)
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
        for (int nb = 0; nb < NBW; ++nb) {
          const short* w0 = wl + ((nb0 + nb) * 32 + r) * ld + (c * CK + s) * 16 + 8 * h;
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
    for (int nb = 0; nb < NBW; ++nb) mfma_results_fence(acc[nb]);
    const int row0 = rb * 32;
#pragma unroll
    for (int nb = 0; nb < NBW; ++nb) {
      const int col = (nb0 + nb) * 32 + r;
      const int colc = col < ldo ? col : ldo - 1;
      if (PREMASK) {      // dx handed down already multiplied by leaky'(x) of the layer below (4 loads together, clamped: the kernel is at its register limit)
#pragma unroll
        for (int i0 = 0; i0 < 16; i0 += 4) {
          float xm[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            int row = row0 + krow(i0 + i, h);
            if (row > N - 1) row = N - 1;
            xm[i] = xmask[(size_t)row * ldo + colc];
          }
#pragma unroll
          for (int i = 0; i < 4; ++i) acc[nb][i0 + i] *= hcg_leaky_grad(xm[i], slope);
        }
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
// A workgroup of 8 waves walks 64-row tiles of dH [N x D] and X [N x F].  K = nodes: both MFMA operands run DOWN the
// columns of the row-major tensors, so every (column, 8 nodes) fragment is loaded once (8 dword loads, a wave's lanes on
// consecutive columns: 256-byte segments; the next tile's loads in flight under this tile's MFMAs), split ONCE into its
// three bf16 pieces and stored as three 16-byte LDS writes into a [column][node] image -- the waves then fetch finished
// operands with ds_read_b128 and spend no VALU on them.  (First version: f32 tiles in LDS, 24 ds_read_b32 + three
// 44-instruction splits per k-step and wave: 66.9 us for C5's layer, VALU-bound.)  The D/32 x FP/32 output blocks are
// spread over the waves (16 blocks: two per wave sharing the dH fragment; fewer blocks than waves: the waves of a block
// take alternate k-steps and meet in LDS at the end, fixed order).
constexpr int DWW = 8, DWT = DWW * 64;
// position of node octet `o` inside column c's image row: XOR-swizzled by the column's 16-block, so that the 16 lanes of a
// ds_write_b128 phase (columns 4 apart: 144-dword stride = 16 banks) spread over all 64 banks; the readers (16 consecutive
// columns of one 16-block: 36-dword stride) are conflict-free either way
__device__ __forceinline__ constexpr int dw_oct(int c, int o) { return o ^ ((c >> 4) & 3); }

// XVEC = false: X rows are not 16-byte aligned (F no multiple of 4, e.g. the reference's 25 node features): its stagers use
// dword loads.  D = 64 (DB = 2): 55 KB of images, two workgroups per CU.
// FIRST (the first layer's whole backward as ONE dense launch; D = 64): Z = the upstream gradient dA, multiplied on the way
// in by leaky'(A) from the forward's sign pieces (`signs` [N][4] x 16 bits: piece j bit q = column 4 q + j is positive);
// X = Ahat x, written by the forward's training form.  dW = (dA (.) leaky')^T (Ahat x) is the same sum as (Ahat^T G)^T x
// in another order, so the layer needs no transpose sum, no dH round trip and no second launch; the bias gradient -- the
// column sums of G -- leaves in `db_slabs` [grid][D].
template <int DB, int NBF, bool XVEC = true, bool FIRST = false>
__global__ __launch_bounds__(DWT, DB == 2 ? 4 : 2) void k_tall_dw(const float* __restrict__ Z, const float* __restrict__ X, int F,
                                                                 float* __restrict__ slabs, int N_arg,
                                                                 const unsigned short* __restrict__ signs = nullptr,
                                                                 float slope = 1.f, float* __restrict__ db_slabs = nullptr,
                                                                 const int32_t* __restrict__ n_dev = nullptr) {
  static_assert(!FIRST || ((DB == 2 || DB == 4) && XVEC), "first-layer form: padded Ahat x rows");
  // `n_dev` (FIRST): the batch's node count in device memory (graph_ptr[B]) -- a captured epoch's batch slots have a fixed
  // CAPACITY of rows (train.EpochWindow), the launch is sized for it and the rows past the batch's own stay out of the sums
  int N = N_arg;
  if constexpr (FIRST) {
    if (n_dev) {
      const int nd = __builtin_amdgcn_readfirstlane(n_dev[0]);
      N = nd < N_arg ? (nd > 0 ? nd : 1) : N_arg;
    }
  }
  constexpr int D = DB * 32, FP = NBF * 32, TR = 64, LDT = TR + 8;
  constexpr int NTILE = DB * NBF, TPW = NTILE >= DWW ? NTILE / DWW : 1, KPARTS = NTILE >= DWW ? 1 : DWW / NTILE;
  static_assert(NTILE * KPARTS == DWW * TPW, "block -> wave map");
  static_assert(TPW == 1 || NBF % TPW == 0, "a wave's blocks share the dH fragment");
  static_assert(KPARTS <= TR / 16, "k-steps per tile");
  constexpr int ZQ = D / 4, XQ = FP / 4, ZTH = ZQ * 8, XTH = XQ * 8;     // column quads / stager threads per tensor
  static_assert(ZTH + XTH <= DWT && ZTH % 64 == 0, "staging map: a (column quad, node octet) per thread");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  short* zp = reinterpret_cast<short*>(smem);          // 3 planes [D][LDT]
  short* xp = zp + 3 * D * LDT;                        // 3 planes [FP][LDT]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int blk0 = (wave % (NTILE / TPW)) * TPW, part = wave / (NTILE / TPW);
  const int mbd = blk0 / NBF, nbf0 = blk0 % NBF;
  const int ntiles = (N + TR - 1) / TR;

  f32x16 dw[TPW];
#pragma unroll
  for (int j = 0; j < TPW; ++j)
#pragma unroll
    for (int i = 0; i < 16; ++i) dw[j][i] = 0.f;

  // staging: the first ZTH threads take dH, the next XTH take X (the rest idle); thread (cq, oct): columns 4 cq .. 4 cq + 3 of
  // nodes 8 oct .. 8 oct + 7: eight float4 loads (a wave covers whole rows per instruction), four splits, twelve 16-byte LDS writes
  // XVEC = false (rows of X not 16-byte aligned): X is staged per (column, node octet) instead -- thread t takes column
  // t % FP of octet t / FP with eight dword loads (a wave's lanes on consecutive columns), one split, three writes; the dH
  // stagers do that on top of their own share.
  const bool is_x = XVEC && tid >= ZTH;                // (wave-uniform)
  const bool live = XVEC ? tid < ZTH + XTH : tid < ZTH;
  const int sidx = is_x ? tid - ZTH : tid;
  const int cq = is_x ? sidx % XQ : sidx % ZQ, oct = (is_x ? sidx / XQ : sidx / ZQ) & 7;
  const int xc = tid % FP, xo = tid / FP;              // XVEC = false: this thread's X item
  const bool x_item = !XVEC && xo < 8;
  static_assert(XVEC || FP * 8 <= DWT, "one X item per thread");
  float4 sv[8];
  float xs[XVEC ? 1 : 8];
  // FIRST: the sign pieces of this thread's eight nodes (D = 64: four 16-bit pieces per row, D = 128: four 32-bit pieces)
  typedef typename std::conditional<DB == 2, uint2, uint4>::type SignRow;
  SignRow sg[FIRST ? 8 : 1];
  float dbacc[FIRST ? 4 : 1] = {};                     // FIRST: column sums of G over this thread's nodes
  auto load_tile = [&](int t) {
    if (live) {
      const int row0 = t * TR + oct * 8;
      const float* src = is_x ? X : Z;
      const int ldm = is_x ? F : D;
      int c = 4 * cq;
      if (c > ldm - 4) c = ldm - 4;
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        int node = row0 + u;
        if (node > N - 1) node = N - 1;
        sv[u] = *reinterpret_cast<const float4*>(src + (size_t)node * ldm + c);
        if constexpr (FIRST) {
          if (!is_x) sg[u] = *reinterpret_cast<const SignRow*>(reinterpret_cast<const char*>(signs) + (size_t)node * sizeof(SignRow));
        }
      }
    }
    if (x_item) {                                       // (wave-uniform: FP * 8 is a multiple of 64)
      const int row0 = t * TR + xo * 8, cc = xc < F ? xc : F - 1;
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        int node = row0 + u;
        if (node > N - 1) node = N - 1;
        xs[u] = X[(size_t)node * F + cc];
      }
    }
  };
  auto store_tile = [&](int t) {                       // rows past N and columns past F become zeros (they are summed)
    if (live) {
      const int row0 = t * TR + oct * 8;
      short* planes = is_x ? xp : zp;
      const int rows = is_x ? FP : D, lim = is_x ? F : D;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int c = 4 * cq + i;
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          float e = i == 0 ? sv[u].x : (i == 1 ? sv[u].y : (i == 2 ? sv[u].z : sv[u].w));
          if constexpr (FIRST) {
            if (!is_x) {                               // G = dA (.) leaky'(A): piece i of the node, bit cq
              unsigned bit;
              if constexpr (DB == 2) {
                const unsigned w = i < 2 ? sg[u].x : sg[u].y;
                bit = (w >> (16 * (i & 1) + cq)) & 1u;
              } else {
                const uint4 w4 = *reinterpret_cast<const uint4*>(&sg[u]);
                const unsigned w = i == 0 ? w4.x : (i == 1 ? w4.y : (i == 2 ? w4.z : w4.w));
                bit = (w >> cq) & 1u;
              }
              e *= bit ? 1.f : slope;
            }
          }
          v[u] = (row0 + u < N && c < lim) ? e : 0.f;
        }
        if constexpr (FIRST) {
          if (!is_x) dbacc[i] += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
        }
        const Split3 sp = split3(v);
        short* dst = planes + c * LDT + dw_oct(c, oct) * 8;
        *reinterpret_cast<bf16x8*>(dst) = sp.p1;
        *reinterpret_cast<bf16x8*>(dst + rows * LDT) = sp.p2;
        *reinterpret_cast<bf16x8*>(dst + 2 * rows * LDT) = sp.p3;
      }
    }
    if (x_item) {
      const int row0 = t * TR + xo * 8;
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = (row0 + u < N && xc < F) ? xs[u] : 0.f;
      const Split3 sp = split3(v);
      short* dst = xp + xc * LDT + dw_oct(xc, xo) * 8;
      *reinterpret_cast<bf16x8*>(dst) = sp.p1;
      *reinterpret_cast<bf16x8*>(dst + FP * LDT) = sp.p2;
      *reinterpret_cast<bf16x8*>(dst + 2 * FP * LDT) = sp.p3;
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
      if (KPARTS > 1 && (ks % KPARTS) != part) continue;      // (wave-uniform)
      const int ca = mbd * 32 + r;
      const short* za = zp + ca * LDT + dw_oct(ca, 2 * ks + h) * 8;
      Split3 As;
      As.p1 = *reinterpret_cast<const bf16x8*>(za);
      As.p2 = *reinterpret_cast<const bf16x8*>(za + D * LDT);
      As.p3 = *reinterpret_cast<const bf16x8*>(za + 2 * D * LDT);
#pragma unroll
      for (int b = 0; b < TPW; ++b) {
        const int cb = (nbf0 + b) * 32 + r;
        const short* xb = xp + cb * LDT + dw_oct(cb, 2 * ks + h) * 8;
        mfma_split(dw[b], As, *reinterpret_cast<const bf16x8*>(xb), *reinterpret_cast<const bf16x8*>(xb + FP * LDT),
                   *reinterpret_cast<const bf16x8*>(xb + 2 * FP * LDT));
      }
    }
  }

  if constexpr (FIRST) {
    // bias gradient: the eight node octets of a column quad meet in LDS (the images are dead), fixed order
    __syncthreads();
    float* comb = reinterpret_cast<float*>(smem);
    if (tid < ZTH) *reinterpret_cast<float4*>(comb + oct * D + 4 * cq) = make_float4(dbacc[0], dbacc[1], dbacc[2], dbacc[3]);
    __syncthreads();
    if (tid < D) {
      float sum = 0.f;
#pragma unroll
      for (int o = 0; o < 8; ++o) sum += comb[o * D + tid];
      db_slabs[(size_t)blockIdx.x * D + tid] = sum;
    }
    __syncthreads();                                   // (the combine below reuses the scratch)
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
    __syncthreads();                                   // the images are dead: [NTILE][32 * 32] combine scratch over them
    float* comb = reinterpret_cast<float*>(smem);
    static_assert(KPARTS == 1 || NTILE * 1024 * 4 <= 3 * (D + FP) * LDT * 2, "combine scratch");
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
    for (int idx = tid; idx < NTILE * 1024; idx += DWT) {
      const int b = idx >> 10, rr = (idx >> 5) & 31, cc = idx & 31;
      slab[((b / NBF) * 32 + rr) * FP + (b % NBF) * 32 + cc] = comb[idx];
    }
  }
}

// =====================================================================================================
// graph part of the 128-wide forward: one graph per workgroup iteration, 16 waves (the whole CU); the graph's rows sit in ONE
// LDS tile
// =====================================================================================================
// A 200-node x 128-d graph is 102 KB: it fits LDS once (not twice, and not beside a weight image).  Per graph: x rows ->
// registers (requested one graph AHEAD, under the previous graph's sums), CSR build, x rows -> tile, H' on the matrix cores
// back into the tile, wavefront segmented sum out of LDS (32 lanes x float4 per row, 32 rows per pass), epilogue,
// row-contiguous 512-byte stores.  (Gathering neighbour rows of a dense H = x W^T from L2 instead -- one workgroup of 4
// waves per graph, 4 per CU -- was the first form: 59-93 us per launch on C5, every pass a dependent L2 round trip.)
#ifdef HCG_SEG_STAMP      // tools/probe_seg.hip: s_memtime stamps of the per-graph phases
__device__ unsigned long long g_seg_stamp[8 * 8 * 8];      // [block < 8][graph iteration < 8][phase < 8]
#define SSTAMP(it, ph) do { if (threadIdx.x == 0 && blockIdx.x < 8 && (it) < 8) { unsigned long long _t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(_t)::"memory"); g_seg_stamp[(blockIdx.x * 8 + (it)) * 8 + (ph)] = _t; } } while (0)
#else
#define SSTAMP(it, ph) do { } while (0)
#endif
#ifdef HCG_SEG_KSTAMP     // tools/probe_seg.hip: stamps inside the k-loop of the first workgroups (wave 0)
__device__ unsigned long long g_seg_kstamp[4 * 2 * 8 * 6];   // [block < 4][graph iteration < 2][k-step < 8][point < 6]
#define KSTAMP(it, ks, pt) do { if (threadIdx.x == 0 && blockIdx.x < 4 && (it) < 2 && (ks) < 8) { unsigned long long _t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(_t)::"memory"); g_seg_kstamp[((blockIdx.x * 2 + (it)) * 8 + (ks)) * 6 + (pt)] = _t; } } while (0)
#else
#define KSTAMP(it, ks, pt) do { } while (0)
#endif
constexpr int SN = 1024, SW = SN / 64;
constexpr int SEG_MAX_NODES = 224;    // (same limits as mid.hip: every batch one family takes, the other takes too)
constexpr int SEG_MAX_EDGES = 1024;   // one edge per thread
constexpr int SEG_D = 128, SEG_TS = SEG_D + 4;
constexpr int SEG_RPT = SEG_MAX_NODES * (SEG_D / 4) / SN;      // float4 per thread to hold one graph's rows (7)
static_assert(SEG_MAX_EDGES == SN && SEG_MAX_NODES * (SEG_D / 4) % SN == 0, "thread maps");

template <bool WITH_RED>      // (k_seg_fwd keeps its combine scratch in the weight-chunk buffers: WITH_RED = false)
struct SegLdsT {
  int rowptr[SEG_MAX_NODES + 4];
  int cursor[SEG_MAX_NODES];
  int degin[SEG_MAX_NODES];
  float dinv[SEG_MAX_NODES];
  unsigned short col[SEG_MAX_EDGES];
  float red[WITH_RED ? SW * 2 * SEG_D : 4];       // combine scratch (the fused forward keeps it in its weight-chunk buffers)
};

struct SegGraph { int nbase, n, ebase, ne, nld; };

__device__ __forceinline__ SegGraph seg_graph(int g, const int32_t* __restrict__ graph_ptr, const int32_t* __restrict__ edge_ptr,
                                              int npad, int32_t* status) {
  SegGraph gi;
  gi.nbase = graph_ptr[g];
  gi.n = graph_ptr[g + 1] - gi.nbase;
  gi.ebase = edge_ptr[g];
  gi.ne = edge_ptr[g + 1] - gi.ebase;
  // host metadata that does not fit: the graph is refused and reported.  The selects stay OUTSIDE the reporting thread's
  // branch (assigned inside it, n and ne became per-lane registers and every address derived from them a 64-bit vector
  // computation -- mid.hip, round 3)
  const bool bad = gi.n < 0 || gi.n > npad || gi.ne < 0 || gi.ne > SEG_MAX_EDGES;
  gi.n = __builtin_amdgcn_readfirstlane(bad ? 0 : gi.n);
  gi.ne = __builtin_amdgcn_readfirstlane(bad ? 0 : gi.ne);
  if (bad && threadIdx.x == 0) atomicOr(status, HCG_STATUS_SHAPE_LIMIT);
  gi.nld = gi.n > 0 ? gi.nbase : 0;     // base of clamped loads (an empty graph at the end of the batch has nbase == N)
  return gi;
}

struct SegEdge {       // this thread's edge of the graph (loads only: unconditional, clamped)
  long long s, d;
  __device__ __forceinline__ void load(const SegGraph& gi, const int64_t* __restrict__ ei, int64_t E) {
    // (workgroup-uniform bases + one unsigned 32-bit byte offset: the scalar-base form of global_load; clamps are scalar work)
    long long eb = gi.ebase;
    eb = eb < 0 ? 0 : (eb > E - 1 ? E - 1 : eb);
    const long long room = E - eb;
    const int nec = (long long)gi.ne < room ? gi.ne : (int)room;
    const int last = nec > 0 ? nec - 1 : 0;
    const int e = threadIdx.x;
    const unsigned off = 8u * (unsigned)(e < last ? e : last);
    s = *reinterpret_cast<const long long*>(reinterpret_cast<const char*>(ei + eb) + off);
    d = *reinterpret_cast<const long long*>(reinterpret_cast<const char*>(ei + E + eb) + off);
  }
};

// this thread's share of a graph's rows of a [N, F] tensor, F <= 128 a multiple of 4: rows rg, rg + 32, ... (rg = tid / 32),
// columns 4 c4 .. 4 c4 + 3 (c4 = tid % 32; columns past F: clamped loads, zeroed by the user)
struct SegRowsF {
  float4 v[SEG_RPT];
  __device__ __forceinline__ void load(const float* __restrict__ src, int F, const SegGraph& gi) {
    const int rg = threadIdx.x >> 5, c4 = threadIdx.x & 31;
    const char* base = reinterpret_cast<const char*>(src + (size_t)gi.nld * F);     // (uniform base + 32-bit byte offsets)
    const unsigned F4 = 4u * (unsigned)F, coff = 4u * (unsigned)(4 * c4 < F ? 4 * c4 : F - 4);
#pragma unroll
    for (int j = 0; j < SEG_RPT; ++j) {
      if (j * 32 < gi.n) {
        const int row = rg + 32 * j;
        v[j] = *reinterpret_cast<const float4*>(base + __umul24((unsigned)(row < gi.n ? row : gi.n - 1), F4) + coff);
      }
    }
  }
};

// The algorithm of mid.hip's build_csr for 1024 threads (one edge each): in-degree -> dinv = (1 + deg_in)^-1/2, counting
// sort into rows (BY_SRC: rows = sources = the transpose), explicit (i, i) edges collapse into the unit self loop, every
// row sorted by id.  Ends with a barrier.
template <bool BY_SRC, class LDS>
__device__ __forceinline__ void seg_build_csr(LDS& L, const SegGraph& gi, const SegEdge& er, int32_t* status) {
  const int tid = threadIdx.x;
  const int n = gi.n;
  if (tid < n) { L.cursor[tid] = 0; if (BY_SRC) L.degin[tid] = 0; }
  __syncthreads();
  unsigned short es = 0xffff, ed = 0xffff;
  bool bad = false;
  if (tid < gi.ne) {
    const unsigned sl = (unsigned)((int)er.s - gi.nbase), dl = (unsigned)((int)er.d - gi.nbase);
    const bool ok = sl < (unsigned)n && dl < (unsigned)n && (er.s >> 31) == 0 && (er.d >> 31) == 0;
    bad = !ok;
    if (ok && sl != dl) {
      es = (unsigned short)sl;
      ed = (unsigned short)dl;
      atomicAdd(&L.cursor[BY_SRC ? sl : dl], 1);
      if (BY_SRC) atomicAdd(&L.degin[dl], 1);
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
  } else if (tid >= SN - SEG_MAX_NODES) {      // (other waves, meanwhile) dinv of every row
    const int i = tid - (SN - SEG_MAX_NODES);
    if (i < n) L.dinv[i] = 1.0f / sqrtf(1.0f + (float)(BY_SRC ? L.degin[i] : L.cursor[i]));
  }
  __syncthreads();
  if (tid < n) L.cursor[tid] = L.rowptr[tid];
  __syncthreads();
  if (es != 0xffff) {
    const int p = atomicAdd(&L.cursor[BY_SRC ? es : ed], 1);
    L.col[p] = BY_SRC ? ed : es;
  }
  __syncthreads();
  if (tid < n) {
    const int i = tid;
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

// acc = t[row] + sum_{k in [kb, ke)} t[col[k]] for this lane's (row, 4 c4 ..) slot: the first four neighbours' rows are
// requested together (independent LDS reads instead of a chain of dependent ones), longer rows loop on
__device__ __forceinline__ float4 seg_row_sum(const float* t, const unsigned short* col, int row, int kb, int ke, int c4) {
  float4 acc = *reinterpret_cast<const float4*>(t + row * SEG_TS + 4 * c4);
  int c[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) c[j] = kb + j < ke ? col[kb + j] : row;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float4 v = *reinterpret_cast<const float4*>(t + c[j] * SEG_TS + 4 * c4);
    if (kb + j < ke) f4_add(acc, v);
  }
  for (int k = kb + 4; __any(k < ke); ++k) {
    if (k < ke) f4_add(acc, *reinterpret_cast<const float4*>(t + col[k] * SEG_TS + 4 * c4));
  }
  return acc;
}

// ---- forward: H' = dinv . H;  out_i = LeakyReLU(dinv_i (H'_i + sum_k H'_k) + b), [max | mean] pooling
// Pre-split image of a layer's weight in global memory for the fused forward: three bf16 planes of [128][KP] (zero past F),
// W = p1 + p2 + p3.  One tiny launch per layer forward; every workgroup then streams 16-column chunks of it through LDS.
__global__ __launch_bounds__(256) void k_split_weight(const float* __restrict__ W, int D, int F, int KP, short* __restrict__ img) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= 128 * KP) return;
  const int n = idx / KP, k = idx - n * KP;
  const float x = (n < D && k < F) ? W[n * F + k] : 0.f;
  const unsigned u1 = pk_bf16(x, 0.f) & 0xffffu;
  const float r1 = x - __uint_as_float(u1 << 16);
  const unsigned u2 = pk_bf16(r1, 0.f) & 0xffffu;
  const float r2 = r1 - __uint_as_float(u2 << 16);
  const unsigned u3 = pk_bf16(r2, 0.f) & 0xffffu;
  // chunk-major: [k / 16][plane][row n][k % 16] -- a workgroup's per-k-step copy is then ONE contiguous 12 KB block (row-major
  // planes made it 384 pieces of 32 bytes, a 128-byte L2 line each: 256 CUs pulling 4x the bytes through the same L2 channels
  // in step was ~1500 of the k-step's 4200 cycles)
  short* dst = img + (size_t)(k >> 4) * (3 * 128 * 16) + n * 16 + (k & 15);
  dst[0] = (short)u1;
  dst[128 * 16] = (short)u2;
  dst[2 * 128 * 16] = (short)u3;
}

constexpr int WCH = 3 * 128 * 16;     // shorts of one 16-column weight chunk (three planes, unpadded rows, XOR-swizzled halves)

// `src` = x [N, F], `gW` = k_split_weight's image: the graph's H tile is produced here, on the matrix cores, straight into LDS --
// H never exists in global memory (420 MB less traffic per C5 layer pair than a dense H = x W^T launch in front of the sums,
// which measured 63 + 55 = 119 us per layer).  The 98 KB weight image cannot sit in LDS beside the 106-118 KB tile, so it
// streams: one
// 16-column chunk (12 KB, double buffered) per k-step, requested from L2 two k-steps ahead; wave (row block, column half)
// keeps 2 accumulator blocks; the A fragments come from the graph's x rows staged through the same LDS tile.
// Measured on C5 (tools/probe_seg.hip, per 200-node graph): CSR 2.3 us, MFMA phase 14 us (its MFMAs alone: 5.9 us -- the
// barrier per k-step keeps all 16 waves in the same phase, so splitting, LDS traffic and MFMAs add up instead of
// overlapping), sums + stores 9 us: 110 us per layer against 119 us for k_tall_mm + the unfused form, with half the HBM
// traffic.  The MFMA phase costs ~3.5 k cycles per k-step against 1.3 k of matrix-pipe time, whatever was tried:
// 32-column chunks (half the barriers); one wave per row block with all four column blocks (one split per 24 MFMAs) -- 145
// spilled registers at the 128 the 16-wave workgroup allows (1.5x slower), and as an 8-wave kernel with 256 registers
// 120 us per layer (phase: 16 us per graph, 9.5 with the MFMAs compiled out, 14.5 with the weight loads compiled out); two
// 64-column groups per graph in two 8-wave workgroups per CU with the A fragments from global memory: 109 / 115 us.
// The chain barrier -> LDS turn-around of the next weight chunk -> dependent MFMA chains per k-step is what they share; a
// resident weight image would remove it and does not fit beside the tile.  PMC (profiles/r02_e_traffic_C5.txt): ~300 MB
// per launch for 210 MB of operands -- the 9-33 spilled registers of this kernel and weight chunks that miss L2.
// BITS (needs POOL; the pooled layer of a training step): `out` is not written -- one byte per (row, 4 columns) leaves
// instead (low nibble: the value is positive, high nibble: it is its graph's column maximum; mid.hip: mid_agg_unit has the
// scheme), all the pooled backward (k_gseg_bwd<.., BITS>) needs of the layer's output.
// ZS (training form of the FIRST layer, not POOL): two more outputs for the dense first-layer backward (k_tall_dw<FIRST>):
// `zagg` [N][zld] = Ahat x and four 32-bit sign pieces per output row (piece j bit q = column 4 q + j is positive).  The x
// rows go into the tile ALREADY scaled by their dinv (the CSR build has finished): Ahat x is then the plain row sum of the
// sums phase over the x tile, and the GEMM yields H' = (dinv x) W^T directly.
template <bool POOL, bool BITS = false, bool ZS = false>
__global__ __launch_bounds__(SN, 4) void k_seg_fwd(const float* __restrict__ src, int F, int KP, const short* __restrict__ gW,
                                                   const float* __restrict__ bias,
                                                   const int64_t* __restrict__ ei, int64_t E, const int32_t* __restrict__ graph_ptr,
                                                   const int32_t* __restrict__ edge_ptr, int B, int npad, float slope, int apply_act,
                                                   float* __restrict__ out, float* __restrict__ emb, int32_t* __restrict__ status,
                                                   unsigned char* __restrict__ poolbits = nullptr,
                                                   float* __restrict__ zagg = nullptr, int zld = 0,
                                                   uint32_t* __restrict__ signs = nullptr) {
  static_assert(!BITS || POOL, "the bit form belongs to the pooled layer");
  static_assert(!ZS || !POOL, "Ahat x / sign pieces: the first (not pooled) layer");
  constexpr int D = SEG_D;
  __shared__ SegLdsT<false> L;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* tile = reinterpret_cast<float*>(smem);          // [npad][SEG_TS]: the graph's x rows, then its H' rows
  short* wl = reinterpret_cast<short*>(tile + (size_t)npad * SEG_TS);     // two weight chunks (2 x 12 KB), later the combine scratch
  float* red = reinterpret_cast<float*>(wl);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c4 = tid & 31, rg = tid >> 5;
  const int r = lane & 31, h = lane >> 5;
  const float4 bq = *reinterpret_cast<const float4*>(bias + 4 * c4);

  SegGraph gnext;
  SegEdge er;
  SegRowsF xrows;
  if ((int)blockIdx.x < B) {
    gnext = seg_graph(blockIdx.x, graph_ptr, edge_ptr, npad, status);
    er.load(gnext, ei, E);
    xrows.load(src, F, gnext);
  }
  // (Per graph on C5, s_memtime stamps of tools/probe_seg.hip: CSR build 1.3 us, tile write 0.85 us, sums + stores 10 us -- the
  //  last phase is the memory system draining 102 KB of loads and 102 KB of stores per CU at ~20 KB/us, i.e. ~5 TB/s over the
  //  chip.  Requesting the next graph's rows one phase earlier, from a second register slot, moved that wait into the CSR
  //  build -- the load issue blocks behind the previous graph's stores -- and was 7 % slower overall.)
  [[maybe_unused]] int sit = 0;
  for (int g = blockIdx.x; g < B; g += gridDim.x, ++sit) {
    const SegGraph gi = gnext;
    SSTAMP(sit, 0);
    seg_build_csr<false>(L, gi, er, status);
    SSTAMP(sit, 1);
    {
      // ---- H' = dinv . (x W^T) -> tile.  The graph's x rows (requested a graph ahead, row-contiguous) go through the SAME
      // LDS tile first: the A fragments are then conflict-free ds_read_b128 instead of 16-byte global loads of 32 different
      // rows per instruction (measured: those loads were ~10 k of the phase's 33 k cycles on C5, and both waves of a row
      // block issued them).  wave = (row block, column half); 14 of 16 waves busy on a 200-node graph.
      const int rbw = wave >> 1, ch = wave & 1;
      const bool active = rbw * 32 < gi.n;                          // wave-uniform
      const int KS = KP / 16;
#pragma unroll
      for (int j = 0; j < SEG_RPT; ++j) {
        const int row = rg + 32 * j;
        if (j * 32 < gi.n && row < gi.n) {
          float4 xv = 4 * c4 < F ? xrows.v[j] : f4_zero();
          if constexpr (ZS) xv = f4_scale(L.dinv[row], xv);
          *reinterpret_cast<float4*>(tile + row * SEG_TS + 4 * c4) = xv;
        }
      }
      f32x16 acc[2];
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[nb][i] = 0.f;
      // the chunk copy: threads 0 .. 767 move 16 bytes each: plane wp, image row wn, half whalf of the 16 columns
      const int wp = tid >> 8, wn = (tid & 255) >> 1, whalf = tid & 1;
      const bool wcopy = tid < 768;
      const short* wsrc = gW + (wcopy ? tid : 0) * 8;                // (chunk-major image: thread t moves bytes [16 t, 16 t + 16) of a chunk)
      short* wdst = wl + (wp * 128 + wn) * 16 + ((whalf ^ ((wn >> 3) & 1)) * 8);
      // weight chunks are requested TWO k-steps ahead (one k-step of MFMAs is ~0.6 us, an L2 round trip under load ~1 us)
      u32x4 wreg = {0u, 0u, 0u, 0u}, wreg2 = {0u, 0u, 0u, 0u};
      if (wcopy) wreg = *reinterpret_cast<const u32x4*>(wsrc);
      if (KS > 1 && wcopy) wreg2 = *reinterpret_cast<const u32x4*>(wsrc + WCH);
      if (wcopy) *reinterpret_cast<u32x4*>(wdst) = wreg;           // chunk 0 -> buffer 0 (free since the previous graph's last barrier)
      wreg = wreg2;
      __syncthreads();                                             // x tile + chunk 0 visible
      if constexpr (ZS) {
        // Ahat x = dinv_i (x'_i + sum of the neighbours' x'_j): the tile is only READ until the k-loop's last barrier
        if (4 * c4 < zld) {
#pragma unroll
          for (int j = 0; j < SEG_RPT; ++j) {
            if (j * 32 < gi.n) {
              const int row = rg + 32 * j;
              const bool valid = row < gi.n;
              const int rr = valid ? row : gi.n - 1;
              const int kb = valid ? L.rowptr[rr] : 0, ke = valid ? L.rowptr[rr + 1] : 0;
              const float4 zs = seg_row_sum(tile, L.col, rr, kb, ke, c4);
              if (valid) *reinterpret_cast<float4*>(zagg + (size_t)(gi.nbase + row) * zld + 4 * c4) = f4_scale(L.dinv[rr], zs);
            }
          }
        }
      }
      const float* arow = tile + (rbw * 32 + r) * SEG_TS + 8 * h;   // (rows past the graph: stale LDS, they only feed unused output rows)
      auto split_a = [&](int ks) {
        const float4 a0 = *reinterpret_cast<const float4*>(arow + 16 * ks);
        const float4 a1 = *reinterpret_cast<const float4*>(arow + 16 * ks + 4);
        const float av[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
        return split3(av);
      };
      // the A fragment of k-step ks + 1 is read and split in the shadow of k-step ks's MFMAs: it does not depend on the
      // barrier (the x tile is complete), and right behind a barrier all 16 waves would otherwise run their ~50 VALU
      // instructions of splitting at the same time, with the matrix pipes idle
      Split3 As = split_a(0);
      for (int ks = 0; ks < KS; ++ks) {
        KSTAMP(sit, ks, 0);
        if (ks + 2 < KS && wcopy) wreg2 = *reinterpret_cast<const u32x4*>(wsrc + (size_t)(ks + 2) * WCH);
        if (active) {
          const short* wb = wl + (ks & 1) * WCH + (ch * 64 + r) * 16 + ((h ^ ((r >> 3) & 1)) * 8);
#pragma unroll
          for (int nb = 0; nb < 2; ++nb) {
            const short* w0 = wb + nb * 32 * 16;
            mfma_split(acc[nb], As, *reinterpret_cast<const bf16x8*>(w0), *reinterpret_cast<const bf16x8*>(w0 + 128 * 16),
                       *reinterpret_cast<const bf16x8*>(w0 + 2 * 128 * 16));
          }
          KSTAMP(sit, ks, 1);
          if (ks + 1 < KS) As = split_a(ks + 1);
          KSTAMP(sit, ks, 2);
        }
        if (ks + 1 < KS && wcopy) *reinterpret_cast<u32x4*>(wdst + ((ks + 1) & 1) * WCH) = wreg;   // (that buffer's readers passed the last barrier)
        KSTAMP(sit, ks, 3);
        __syncthreads();
        KSTAMP(sit, ks, 4);
        wreg = wreg2;
      }
      // (the loop's last barrier: every A fragment has been read -- the accumulators may now overwrite the tile)
      mfma_results_fence(acc[0], acc[1]);
      if (active) {
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const int row = rbw * 32 + krow(i, h);
            if (row < gi.n) tile[row * SEG_TS + (ch * 2 + nb) * 32 + r] = ZS ? acc[nb][i] : acc[nb][i] * L.dinv[row];
          }
      }
    }
    SSTAMP(sit, 2);
    __syncthreads();
    SSTAMP(sit, 3);
    if (g + (int)gridDim.x < B) {                     // the NEXT graph's scalars, edge and rows: in flight under this graph's sums
      gnext = seg_graph(g + gridDim.x, graph_ptr, edge_ptr, npad, status);
      er.load(gnext, ei, E);
      xrows.load(src, F, gnext);                      // (requested before the MFMA loop instead: the same ~5 us of exposed load
    }                                                 //  time per graph moves into that phase -- measured equal, more spills)
    float4 pmax = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY), psum = f4_zero();
    unsigned sgn = 0, mxb = 0;                         // BITS: nibble j = this thread's row rg + 32 j, bit c = column 4 c4 + c
#pragma unroll
    for (int j = 0; j < SEG_RPT; ++j) {
      if (j * 32 < gi.n) {                             // (block-uniform: the tail loop of seg_row_sum votes per wave)
        const int row = rg + 32 * j;
        const bool valid = row < gi.n;
        const int rr = valid ? row : gi.n - 1;
        const int kb = valid ? L.rowptr[rr] : 0, ke = valid ? L.rowptr[rr + 1] : 0;
        const float4 acc = seg_row_sum(tile, L.col, rr, kb, ke, c4);
        const float di = L.dinv[rr];
        float4 y = make_float4(fmaf(di, acc.x, bq.x), fmaf(di, acc.y, bq.y), fmaf(di, acc.z, bq.z), fmaf(di, acc.w, bq.w));
        if (apply_act) { y.x = fmaxf(y.x, slope * y.x); y.y = fmaxf(y.y, slope * y.y); y.z = fmaxf(y.z, slope * y.z); y.w = fmaxf(y.w, slope * y.w); }
        if (valid) {
          if constexpr (BITS) {
            const float yv[4] = {y.x, y.y, y.z, y.w}, mv[4] = {pmax.x, pmax.y, pmax.z, pmax.w};
            unsigned sn = 0;
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) {
              sn |= (unsigned)(yv[cc] > 0.f) << cc;
              const unsigned bit = 1u << (4 * j + cc), colmask = 0x11111111u << cc;
              mxb = yv[cc] > mv[cc] ? ((mxb & ~colmask) | bit) : (yv[cc] == mv[cc] ? (mxb | bit) : mxb);
            }
            sgn |= sn << (4 * j);
          } else {
            *reinterpret_cast<float4*>(out + (size_t)(gi.nbase + row) * D + 4 * c4) = y;
          }
          if constexpr (ZS) {
            const unsigned long long b0 = __builtin_amdgcn_ballot_w64(y.x > 0.f), b1 = __builtin_amdgcn_ballot_w64(y.y > 0.f);
            const unsigned long long b2 = __builtin_amdgcn_ballot_w64(y.z > 0.f), b3 = __builtin_amdgcn_ballot_w64(y.w > 0.f);
            if (c4 < 4) {
              const unsigned long long bsel = c4 == 0 ? b0 : (c4 == 1 ? b1 : (c4 == 2 ? b2 : b3));
              signs[(size_t)(gi.nbase + row) * 4 + c4] = (uint32_t)(bsel >> (32 * (lane >> 5)));
            }
          }
          if (POOL) {
            pmax = make_float4(fmaxf(pmax.x, y.x), fmaxf(pmax.y, y.y), fmaxf(pmax.z, y.z), fmaxf(pmax.w, y.w));
            f4_add(psum, y);
          }
        }
      }
    }
    const float4 own_max = pmax;                       // (BITS: this thread's own maxima, before the combine)
    if (POOL) {      // the two row slots of a wave (xor 32) -> workgroup (LDS, fixed order over the waves)
      pmax = make_float4(fmaxf(pmax.x, __shfl_xor(pmax.x, 32, 64)), fmaxf(pmax.y, __shfl_xor(pmax.y, 32, 64)),
                         fmaxf(pmax.z, __shfl_xor(pmax.z, 32, 64)), fmaxf(pmax.w, __shfl_xor(pmax.w, 32, 64)));
      psum.x += __shfl_xor(psum.x, 32, 64); psum.y += __shfl_xor(psum.y, 32, 64);
      psum.z += __shfl_xor(psum.z, 32, 64); psum.w += __shfl_xor(psum.w, 32, 64);
      if (lane < 32) {
        *reinterpret_cast<float4*>(red + wave * 2 * D + 4 * c4) = pmax;
        *reinterpret_cast<float4*>(red + wave * 2 * D + D + 4 * c4) = psum;
      }
      __syncthreads();
      if (tid < D) {
        float m = -INFINITY, s = 0.f;
#pragma unroll
        for (int w = 0; w < SW; ++w) {
          m = fmaxf(m, red[w * 2 * D + tid]);
          s += red[w * 2 * D + D + tid];
        }
        if (gi.n <= 0) m = 0.f;
        emb[(size_t)g * 2 * D + tid] = m;
        emb[(size_t)g * 2 * D + D + tid] = s / (float)(gi.n > 0 ? gi.n : 1);
        if (BITS) red[tid] = m;        // (wave 0's slot of this column: this thread was its only reader)
      }
    }
    SSTAMP(sit, 4);
    __syncthreads();   // the tile, the CSR and the combine scratch are free for the next graph
    SSTAMP(sit, 5);
    if constexpr (BITS) {
      // (`red` aliases the weight-chunk buffers: their next write sits behind the barriers of the next graph's CSR build)
      const float4 gm = *reinterpret_cast<const float4*>(red + 4 * c4);
      const unsigned keep = (own_max.x == gm.x ? 0x11111111u : 0u) | (own_max.y == gm.y ? 0x22222222u : 0u) |
                            (own_max.z == gm.z ? 0x44444444u : 0u) | (own_max.w == gm.w ? 0x88888888u : 0u);
      mxb &= keep;
      unsigned char* bits_graph = poolbits + (size_t)gi.nbase * (D / 4) + c4;
#pragma unroll
      for (int j = 0; j < SEG_RPT; ++j) {
        const int row = rg + 32 * j;
        if (j * 32 < gi.n && row < gi.n)
          bits_graph[(size_t)row * (D / 4)] = (unsigned char)(((sgn >> (4 * j)) & 0xfu) | (((mxb >> (4 * j)) & 0xfu) << 4));
      }
    }
  }
}

// =====================================================================================================
// graph part of the backward: G = dA (.) leaky'(A);  db += colsum G;  dY' = dinv . G;
//                             dH_j = dinv_j (dY'_j + sum_{k in row j of the transpose} dY'_k)
// =====================================================================================================
// POOLG: dA is the pooled gradient expanded on chip (mean share + the max's share split evenly over ties, as
// global_max_pool's backward does through torch.max).  TWO: both dout and a_out are read (activation derivative applied
// here); otherwise exactly one tensor is read.  Per graph: rows -> registers (requested one graph ahead), CSR build of the
// transpose, G rows scaled by dinv into an LDS tile, wavefront segmented sum out of LDS, row-contiguous stores.
// The sums never mix columns, so a layer is handled in independent 64-column groups (blockIdx.y): a 64-wide tile of a
// <= 128-node graph is 35 KB -- four workgroups of 4 waves share a CU and the per-graph phases of four graphs overlap
// (<= 224 nodes: 8 waves, two per CU).  (The first form kept a 128-wide tile, 106 KB = ONE 16-wave workgroup per CU whose
// phases ran back to back: 70 us per launch on C5 against ~55 now; a second register slot for the rows of graph g + 1,
// requested before graph g's CSR build, only moved its wait: the load issue blocks behind the previous graph's stores.)
template <int D, int NT, int NMAX>
struct GS {
  static constexpr int LPR = D / 4, RPP = NT / LPR, RPT = (NMAX + RPP - 1) / RPP, TS = D + 4, NW = NT / 64;
  static constexpr int EPT = SEG_MAX_EDGES / NT;
  static_assert(NT % LPR == 0 && SEG_MAX_EDGES % NT == 0 && NT >= 128, "thread maps");
  struct Lds {
    int rowptr[NMAX + 4];
    int cursor[NMAX];
    int degin[NMAX];
    float dinv[NMAX];
    unsigned short col[SEG_MAX_EDGES];
    float red[NW * D];
  };
  struct Edges {       // this thread's edges of the graph (loads only: unconditional, clamped)
    long long s[EPT], d[EPT];
    __device__ __forceinline__ void load(const SegGraph& gi, const int64_t* __restrict__ ei, int64_t E) {
      long long eb = gi.ebase;
      eb = eb < 0 ? 0 : (eb > E - 1 ? E - 1 : eb);
      const long long room = E - eb;
      const int nec = (long long)gi.ne < room ? gi.ne : (int)room;
      const int last = nec > 0 ? nec - 1 : 0;
      const char* sb = reinterpret_cast<const char*>(ei + eb);
      const char* db = reinterpret_cast<const char*>(ei + E + eb);
#pragma unroll
      for (int j = 0; j < EPT; ++j) {
        const int e = threadIdx.x + j * NT;
        const unsigned off = 8u * (unsigned)(e < last ? e : last);
        s[j] = *reinterpret_cast<const long long*>(sb + off);
        d[j] = *reinterpret_cast<const long long*>(db + off);
      }
    }
  };
  // this thread's share of a graph's rows: rows rg, rg + RPP, ... (rg = tid / LPR), columns 4 c4 .. 4 c4 + 3 (c4 = tid % LPR)
  struct Rows {
    float4 v[RPT];
    __device__ __forceinline__ void load(const float* __restrict__ src, const SegGraph& gi, int ld) {
      const int rg = threadIdx.x / LPR, c4 = threadIdx.x % LPR;
      const char* base = reinterpret_cast<const char*>(src + (size_t)gi.nld * ld);     // (uniform base + 32-bit byte offsets)
      const unsigned ld4 = 4u * (unsigned)ld;
#pragma unroll
      for (int j = 0; j < RPT; ++j) {
        if (j * RPP < gi.n) {                     // block-uniform guard, clamped address: no per-lane branch around the load
          const int row = rg + RPP * j;
          v[j] = *reinterpret_cast<const float4*>(base + __umul24((unsigned)(row < gi.n ? row : gi.n - 1), ld4) + 16u * (unsigned)c4);
        }
      }
    }
  };
  // mid.hip's build_csr (see seg_build_csr above), any thread count
  template <bool BY_SRC>
  static __device__ __forceinline__ void build_csr(Lds& L, const SegGraph& gi, const Edges& er, int32_t* status) {
    const int tid = threadIdx.x;
    const int n = gi.n;
    for (int i = tid; i < n; i += NT) { L.cursor[i] = 0; if (BY_SRC) L.degin[i] = 0; }
    __syncthreads();
    unsigned short es[EPT], ed[EPT];
    bool bad = false;
#pragma unroll
    for (int j = 0; j < EPT; ++j) {
      es[j] = 0xffff;
      ed[j] = 0xffff;
      if (tid + j * NT < gi.ne) {
        const unsigned sl = (unsigned)((int)er.s[j] - gi.nbase), dl = (unsigned)((int)er.d[j] - gi.nbase);
        const bool ok = sl < (unsigned)n && dl < (unsigned)n && (er.s[j] >> 31) == 0 && (er.d[j] >> 31) == 0;
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
    if (tid < 64) {     // exclusive scan of the row sizes (<= 256 rows: 4 per lane)
      constexpr int RPL = (NMAX + 63) / 64;
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
    } else {                                     // (the other waves, meanwhile) dinv of every row
      for (int i = tid - 64; i < n; i += NT - 64) L.dinv[i] = 1.0f / sqrtf(1.0f + (float)(BY_SRC ? L.degin[i] : L.cursor[i]));
    }
    __syncthreads();
    for (int i = tid; i < n; i += NT) L.cursor[i] = L.rowptr[i];
    __syncthreads();
#pragma unroll
    for (int j = 0; j < EPT; ++j) {
      if (es[j] != 0xffff) {
        const int p = atomicAdd(&L.cursor[BY_SRC ? es[j] : ed[j]], 1);
        L.col[p] = BY_SRC ? ed[j] : es[j];
      }
    }
    __syncthreads();
    for (int i = tid; i < n; i += NT) {
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
  static __device__ __forceinline__ float4 row_sum(const float* t, const unsigned short* col, int row, int kb, int ke, int c4) {
    float4 acc = *reinterpret_cast<const float4*>(t + row * TS + 4 * c4);
    int c[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) c[j] = kb + j < ke ? col[kb + j] : row;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float4 v = *reinterpret_cast<const float4*>(t + c[j] * TS + 4 * c4);
      if (kb + j < ke) f4_add(acc, v);
    }
    for (int k = kb + 4; __any(k < ke); ++k) {
      if (k < ke) f4_add(acc, *reinterpret_cast<const float4*>(t + col[k] * TS + 4 * c4));
    }
    return acc;
  }
  // the lanes of a wave that hold the same columns (64 / LPR of them) -> one value per column, fixed order
  static __device__ __forceinline__ float4 fold(float4 v) {
#pragma unroll
    for (int off = LPR; off < 64; off <<= 1) {
      v.x += __shfl_xor(v.x, off, 64); v.y += __shfl_xor(v.y, off, 64); v.z += __shfl_xor(v.z, off, 64); v.w += __shfl_xor(v.w, off, 64);
    }
    return v;
  }
};

// BITS (needs POOLG): the pooled layer's output was never stored; one byte per (row, 4 columns) -- low nibble: the value is
// positive, high nibble: it is its graph's column maximum -- written by the forward's bit form (mid.hip: mid_agg_unit,
// k_seg_fwd<POOL, BITS>) stands in for `a_out` AND `emb`: a sixteenth of the bytes, and no equality tests against the maxima.
template <int D, int NT, int NMAX, bool POOLG, bool TWO, bool BITS = false>
__global__ __launch_bounds__(NT, NT == 256 ? 4 : 2) void k_gseg_bwd(
    const float* __restrict__ dout, const float* __restrict__ demb, const float* __restrict__ emb, const float* __restrict__ a_out,
    const int64_t* __restrict__ ei, int64_t E, const int32_t* __restrict__ graph_ptr, const int32_t* __restrict__ edge_ptr, int B,
    int npad, float slope, int act_here, float* __restrict__ Z, float* __restrict__ db_slabs, int32_t* __restrict__ status, int ld,
    const unsigned char* __restrict__ poolbits = nullptr) {
  static_assert(!BITS || (POOLG && !TWO), "the bit form is the pooled backward");
  // `ld` = row length of the tensors (a layer `ld` columns wide is handled as ld / D independent column groups, blockIdx.y:
  // the sums never mix columns, and D = 64 column groups of a 128-wide layer leave room for two workgroups per CU)
  const int coff = blockIdx.y * D;
  using G = GS<D, NT, NMAX>;
  constexpr int LPR = G::LPR, RPP = G::RPP, RPT = G::RPT, TS = G::TS, NW = G::NW;
  constexpr bool NEED_A = (POOLG || TWO) && !BITS;
  __shared__ typename G::Lds L;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* tile = reinterpret_cast<float*>(smem);          // [npad][TS]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c4 = tid % LPR, rg = tid / LPR;
  float4 dbacc = f4_zero();

  SegGraph gnext;
  typename G::Edges er;
  typename G::Rows drows, arows;
  unsigned char brows[BITS ? RPT : 1];                 // BITS: this thread's byte of each of its rows
  float4 gmx = f4_zero(), dmx = f4_zero(), dmean = f4_zero();
  auto request = [&](int g) {                          // everything of graph g this thread will need: loads only
    gnext = seg_graph(g, graph_ptr, edge_ptr, npad, status);
    er.load(gnext, ei, E);
    if (!POOLG) drows.load(dout + coff, gnext, ld);
    if (NEED_A) arows.load(a_out + coff, gnext, ld);
    if constexpr (BITS) {
      const unsigned ldb = (unsigned)ld >> 2;
      const unsigned char* base = poolbits + (size_t)gnext.nld * ldb + (coff >> 2) + c4;
#pragma unroll
      for (int j = 0; j < RPT; ++j) {
        if (j * RPP < gnext.n) {
          const int row = rg + RPP * j;
          brows[j] = base[__umul24((unsigned)(row < gnext.n ? row : gnext.n - 1), ldb)];
        }
      }
    }
    if (POOLG) {
      const size_t eb = (size_t)g * 2 * ld + coff + 4 * c4;      // [max | mean], each ld wide
      if (!BITS) gmx = *reinterpret_cast<const float4*>(emb + eb);
      dmx = *reinterpret_cast<const float4*>(demb + eb);
      dmean = *reinterpret_cast<const float4*>(demb + eb + ld);
    }
  };
  if ((int)blockIdx.x < B) request(blockIdx.x);
  for (int g = blockIdx.x; g < B; g += gridDim.x) {
    const SegGraph gi = gnext;
    G::template build_csr<true>(L, gi, er, status);
    float4 share = f4_zero(), dmn = f4_zero();
    const float4 gm = gmx;
    if (POOLG) {
      const float cntf = (float)(gi.n > 0 ? gi.n : 1);
      dmn = make_float4(dmean.x / cntf, dmean.y / cntf, dmean.z / cntf, dmean.w / cntf);
      float4 ties = f4_zero();                           // ties of the column maxima among this thread's rows
#pragma unroll
      for (int j = 0; j < RPT; ++j) {
        if (j * RPP < gi.n && rg + RPP * j < gi.n) {
          if constexpr (BITS) {
            const unsigned bv = brows[j];
            ties.x += (float)(bv >> 4 & 1u); ties.y += (float)(bv >> 5 & 1u); ties.z += (float)(bv >> 6 & 1u); ties.w += (float)(bv >> 7 & 1u);
          } else {
            const float4 a = arows.v[j];
            ties.x += (a.x == gm.x); ties.y += (a.y == gm.y); ties.z += (a.z == gm.z); ties.w += (a.w == gm.w);
          }
        }
      }
      ties = G::fold(ties);
      if (lane < LPR) *reinterpret_cast<float4*>(L.red + wave * D + 4 * c4) = ties;
      __syncthreads();
      float4 tot = f4_zero();
#pragma unroll
      for (int w = 0; w < NW; ++w) f4_add(tot, *reinterpret_cast<const float4*>(L.red + w * D + 4 * c4));   // (counts: exact)
      share = make_float4(dmx.x / fmaxf(tot.x, 1.f), dmx.y / fmaxf(tot.y, 1.f), dmx.z / fmaxf(tot.z, 1.f), dmx.w / fmaxf(tot.w, 1.f));
    }
#pragma unroll
    for (int j = 0; j < RPT; ++j) {
      const int row = rg + RPP * j;
      if (j * RPP < gi.n && row < gi.n) {
        float4 a = f4_zero(), gq;
        if (NEED_A) a = arows.v[j];
        if constexpr (BITS) {
          const unsigned bv = brows[j];
          gq = make_float4(dmn.x + ((bv & 0x10u) ? share.x : 0.f), dmn.y + ((bv & 0x20u) ? share.y : 0.f),
                           dmn.z + ((bv & 0x40u) ? share.z : 0.f), dmn.w + ((bv & 0x80u) ? share.w : 0.f));
          if (act_here) {
            gq.x *= (bv & 1u) ? 1.f : slope; gq.y *= (bv & 2u) ? 1.f : slope;
            gq.z *= (bv & 4u) ? 1.f : slope; gq.w *= (bv & 8u) ? 1.f : slope;
          }
        } else if (POOLG) {
          gq = make_float4(dmn.x + (a.x == gm.x ? share.x : 0.f), dmn.y + (a.y == gm.y ? share.y : 0.f),
                           dmn.z + (a.z == gm.z ? share.z : 0.f), dmn.w + (a.w == gm.w ? share.w : 0.f));
        } else {
          gq = drows.v[j];
        }
        if (NEED_A && act_here) {
          gq.x *= hcg_leaky_grad(a.x, slope); gq.y *= hcg_leaky_grad(a.y, slope);
          gq.z *= hcg_leaky_grad(a.z, slope); gq.w *= hcg_leaky_grad(a.w, slope);
        }
        f4_add(dbacc, gq);
        *reinterpret_cast<float4*>(tile + row * TS + 4 * c4) = f4_scale(L.dinv[row], gq);
      }
    }
    __syncthreads();
    if (g + (int)gridDim.x < B) request(g + gridDim.x);   // the NEXT graph: in flight under this graph's sums
#pragma unroll
    for (int j = 0; j < RPT; ++j) {
      if (j * RPP < gi.n) {
        const int row = rg + RPP * j;
        const bool valid = row < gi.n;
        const int rr = valid ? row : gi.n - 1;
        const int kb = valid ? L.rowptr[rr] : 0, ke = valid ? L.rowptr[rr + 1] : 0;
        const float4 acc = G::row_sum(tile, L.col, rr, kb, ke, c4);
        if (valid) *reinterpret_cast<float4*>(Z + (size_t)(gi.nbase + row) * ld + coff + 4 * c4) = f4_scale(L.dinv[rr], acc);
      }
    }
    __syncthreads();   // the tile, the CSR and the tie scratch are free for the next graph
  }
  // ---- this workgroup's bias-gradient slab [D]: lanes of a column -> wave -> workgroup (LDS), fixed order
  dbacc = G::fold(dbacc);
  if (lane < LPR) *reinterpret_cast<float4*>(L.red + wave * D + 4 * c4) = dbacc;
  __syncthreads();
  if (tid < D) {
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) s += L.red[w * D + tid];
    db_slabs[(size_t)blockIdx.x * ld + coff + tid] = s;
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
int seg_grid(int64_t B) {            // one workgroup of 16 waves per CU
  int64_t g = (int64_t)cu_count();
  if (g > B) g = B;
  return g < 1 ? 1 : (int)g;
}
int mm_grid(int64_t N) {
  int64_t g = hcg_cdiv(hcg_cdiv(N, 32), TW / 2);
  if (g > cu_count()) g = cu_count();
  return g < 1 ? 1 : (int)g;
}
constexpr int DW_TILE = 64;
int dw_grid(int64_t N, int64_t D = 128) {   // one workgroup per CU (110 KB of operand images; D = 64: two); every workgroup leaves a slab
  int64_t g = hcg_cdiv(N, DW_TILE);
  const int64_t cap = (int64_t)cu_count() * (D == 64 ? 2 : 1);
  if (g > cap) g = cap;
  return g < 1 ? 1 : (int)g;
}
int seg_grid64(int64_t B) {          // 64-wide layers: up to four 4-wave workgroups per CU (two 8-wave ones for graphs > 128 nodes)
  int64_t g = (int64_t)cu_count() * 4;
  if (g > B) g = B;
  return g < 1 ? 1 : (int)g;
}
int seg_npad(int64_t max_nodes) { return (int)((max_nodes + 3) / 4 * 4); }
size_t seg_tile_bytes(int npad) { return (size_t)npad * SEG_TS * sizeof(float); }
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
  const size_t o2 = o1 + hcg_align_up((size_t)dw_grid(N, D) * D * tall_fpad(F) * sizeof(float), 256);
  const int dbs = seg_grid64(B) > dw_grid(N, D) ? seg_grid64(B) : dw_grid(N, D);      // (first-layer form: one db slab per dW slab)
  const size_t o3 = o2 + hcg_align_up((size_t)dbs * D * sizeof(float), 256);
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
  if (max_nodes_per_graph < 1 || max_nodes_per_graph > SEG_MAX_NODES || max_edges_per_graph < 0 || max_edges_per_graph > SEG_MAX_EDGES)
    return 0;
  if (D == 128) return (F >= 4 && F <= 128 && (F % 4) == 0) ? 1 : 0;
  // 64-wide layers over graphs of 65 .. 224 nodes (the reference's own sizes; up to 64 nodes the one-graph-per-wave kernels
  // take the layer): the BACKWARD runs here, the forward stays with the one-graph-per-workgroup kernel (hcg_mid_layer_fwd)
  if (D == 64) return (F >= 1 && F <= 64 && max_nodes_per_graph > 64 && hcg_mid_supported(F, D, max_nodes_per_graph, max_edges_per_graph)) ? 1 : 0;
  return 0;
}

extern "C" size_t hcg_tall_workspace_bytes(int64_t N, int64_t B, int64_t F, int64_t D) {
  if (N <= 0 || B <= 0 || (D != 128 && D != 64) || F < 1 || F > 128) return 0;
  return tall_carve(nullptr, N, B, F, D).total;
}

extern "C" int hcg_tall_layer_fwd(const float* x, const float* W, const float* b, const int64_t* edge_index, int64_t E,
                                  const int32_t* graph_ptr, const int32_t* edge_ptr, int64_t N, int64_t B, int64_t F, int64_t D,
                                  int64_t max_nodes, int64_t max_edges, float slope, int apply_act, float* out, float* emb,
                                  uint8_t* poolbits, float* xagg, uint8_t* signbits, int32_t* status, void* workspace,
                                  size_t workspace_bytes, hcg_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!hcg_tall_supported(F, D, max_nodes, max_edges)) return HCG_ERR_UNSUPPORTED;
  if (D == 64)     // (64-wide layers: only the backward is cut this way)
    return hcg_mid_layer_fwd(x, W, b, edge_index, E, graph_ptr, edge_ptr, N, B, F, D, max_nodes, max_edges, slope, apply_act, out, emb,
                             poolbits, xagg, signbits, status, stream_);
  if ((xagg == nullptr) != (signbits == nullptr)) return HCG_ERR_INVALID_ARG;
  if (xagg && (emb || poolbits || !out)) return HCG_ERR_UNSUPPORTED;     // (first-layer form: a layer that is not pooled)
  if (poolbits && !emb) return HCG_ERR_INVALID_ARG;
  if (apply_act && !(slope >= 0.f && slope <= 1.f)) return HCG_ERR_UNSUPPORTED;   // LeakyReLU is evaluated as max(v, slope*v)
  if (N < 0 || B < 0 || E < 0) return HCG_ERR_INVALID_ARG;
  if (B == 0 || N == 0) return HCG_OK;
  if (!x || !W || !b || !graph_ptr || !edge_ptr || (!out && !poolbits) || !status || !workspace || (E > 0 && !edge_index)) return HCG_ERR_INVALID_ARG;
  if (workspace_bytes < hcg_tall_workspace_bytes(N, B, F, D)) return HCG_ERR_WORKSPACE;
  if (N > (int64_t)INT32_MAX / 2) return HCG_ERR_UNSUPPORTED;
  if (E == 0) { edge_index = reinterpret_cast<const int64_t*>(graph_ptr); E = 1; }
  const TallWs ws = tall_carve(workspace, N, B, F, D);
  const dim3 sgrid(seg_grid(B)), sblk(SN);
  const int npad = seg_npad(max_nodes);
  const size_t slds = seg_tile_bytes(npad);
  const size_t wbuf = (size_t)2 * WCH * sizeof(short);
  // x -> (MFMA) -> H tile in LDS -> sums: the tile (<= 224 rows), two weight chunks and the CSR fit 160 KB of LDS
  {
    const int KP = (int)((F + 15) / 16 * 16);
    short* img = reinterpret_cast<short*>(ws.inter);                 // 3 x 128 x KP bf16
    hipLaunchKernelGGL(k_split_weight, dim3((128 * KP + 255) / 256), dim3(256), 0, stream, W, (int)D, (int)F, KP, img);
    HCG_CHECK_LAUNCH();
    const size_t lds = slds + wbuf, lds_max = 160 * 1024 - 512 - sizeof(SegLdsT<false>);
    if (lds > lds_max) return HCG_ERR_UNSUPPORTED;
    hipError_t e = xagg ? allow_lds<k_seg_fwd<false, false, true>>(lds_max)
                        : (poolbits ? allow_lds<k_seg_fwd<true, true>>(lds_max)
                                    : (emb ? allow_lds<k_seg_fwd<true>>(lds_max) : allow_lds<k_seg_fwd<false>>(lds_max)));
    if (e != hipSuccess) return hcg_hip_err(e);
    if (xagg)
      hipLaunchKernelGGL((k_seg_fwd<false, false, true>), sgrid, sblk, lds, stream, x, (int)F, KP, (const short*)img, b, edge_index,
                         E, graph_ptr, edge_ptr, (int)B, npad, slope, apply_act, out, emb, status, (unsigned char*)nullptr, xagg,
                         tall_fpad(F), reinterpret_cast<uint32_t*>(signbits));
    else if (poolbits)
      hipLaunchKernelGGL((k_seg_fwd<true, true>), sgrid, sblk, lds, stream, x, (int)F, KP, (const short*)img, b, edge_index, E,
                         graph_ptr, edge_ptr, (int)B, npad, slope, apply_act, out, emb, status, poolbits);
    else if (emb)
      hipLaunchKernelGGL((k_seg_fwd<true>), sgrid, sblk, lds, stream, x, (int)F, KP, (const short*)img, b, edge_index, E,
                         graph_ptr, edge_ptr, (int)B, npad, slope, apply_act, out, emb, status);
    else
      hipLaunchKernelGGL((k_seg_fwd<false>), sgrid, sblk, lds, stream, x, (int)F, KP, (const short*)img, b, edge_index, E,
                         graph_ptr, edge_ptr, (int)B, npad, slope, apply_act, out, emb, status);
    HCG_CHECK_LAUNCH();
  }
  return HCG_OK;
}

// dout == NULL selects the pooled form (upstream gradient = demb [B, 2D], expanded on chip with `emb` and `out`).
// apply_act: bit 0 = multiply the upstream gradient by leaky'(out); bit 1 = hand dx down already multiplied by leaky'(x).
// Leaves dW / db slabs in `workspace`: describe them with hcg_tall_reduce_jobs (two jobs) and sum with hcg_step_tail.
extern "C" int hcg_tall_layer_bwd(const float* dout, const float* demb, const float* emb, const float* out,
                                  const uint8_t* poolbits, const float* xagg, const uint8_t* signbits, const int32_t* n_dev,
                                  const float* x,
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
  const bool bits = poolbits != nullptr;      // the forward's bit form stands in for `out` and `emb`
  if (bits && !poolg) return HCG_ERR_INVALID_ARG;
  if ((xagg == nullptr) != (signbits == nullptr)) return HCG_ERR_INVALID_ARG;
  if (xagg) {
    // ---- first-layer form: dW = (dout (.) leaky'(out))^T (Ahat x) and db = its column sums, ONE dense launch (k_tall_dw<FIRST>)
    if (dx || poolg || (D == 64 && F > 64)) return HCG_ERR_UNSUPPORTED;
    if (workspace_bytes < hcg_tall_workspace_bytes(N, B, F, D)) return HCG_ERR_WORKSPACE;
    const TallWs ws = tall_carve(workspace, N, B, F, D);
    const int fp = tall_fpad(F);
    const dim3 grid(dw_grid(N, D)), blk(DWT);
    const float slope_eff = (apply_act & 1) ? slope : 1.f;
#define LAUNCH_DW_FIRST(DBV, NBF)                                                                                        \
  do {                                                                                                                   \
    const size_t lds = (size_t)3 * (DBV * 32 + NBF * 32) * (DW_TILE + 8) * sizeof(short);                                \
    hipError_t e = allow_lds<k_tall_dw<DBV, NBF, true, true>>(lds);                                                      \
    if (e != hipSuccess) return hcg_hip_err(e);                                                                          \
    hipLaunchKernelGGL((k_tall_dw<DBV, NBF, true, true>), grid, blk, lds, stream, dout, xagg, fp, ws.dw_slabs, (int)N,   \
                       reinterpret_cast<const unsigned short*>(signbits), slope_eff, ws.db_slabs, n_dev);                \
  } while (0)
    if (D == 64) { if (fp == 32) LAUNCH_DW_FIRST(2, 1); else LAUNCH_DW_FIRST(2, 2); }
    else         { if (fp == 32) LAUNCH_DW_FIRST(4, 1); else if (fp == 64) LAUNCH_DW_FIRST(4, 2); else LAUNCH_DW_FIRST(4, 4); }
#undef LAUNCH_DW_FIRST
    HCG_CHECK_LAUNCH();
    return HCG_OK;
  }
  if (poolg && (!demb || (!emb && !bits))) return HCG_ERR_INVALID_ARG;
  if ((apply_act & ~3) || ((apply_act & 2) && !dx)) return HCG_ERR_INVALID_ARG;
  if ((poolg || (apply_act & 1)) && !out && !bits) return HCG_ERR_INVALID_ARG;
  if (workspace_bytes < hcg_tall_workspace_bytes(N, B, F, D)) return HCG_ERR_WORKSPACE;
  if (E == 0) { edge_index = reinterpret_cast<const int64_t*>(graph_ptr); E = 1; }
  const TallWs ws = tall_carve(workspace, N, B, F, D);
  const int act_here = apply_act & 1;
  {   // ---- dH = Ahat^T (dA . leaky'), db slabs: 64-column groups (one for D = 64, two for D = 128: blockIdx.y)
    const int npad = seg_npad(max_nodes);
    const dim3 sgrid(seg_grid64(B), (unsigned)(D / 64));
    const size_t slds = (size_t)npad * (64 + 4) * sizeof(float);
#define LAUNCH_GSEG(NTV, NMAXV, PG, TW2, AOUT, BT)                                                                         \
  do {                                                                                                                     \
    hipError_t e = allow_lds<k_gseg_bwd<64, NTV, NMAXV, PG, TW2, BT>>((size_t)NMAXV * 68 * sizeof(float));                \
    if (e != hipSuccess) return hcg_hip_err(e);                                                                            \
    hipLaunchKernelGGL((k_gseg_bwd<64, NTV, NMAXV, PG, TW2, BT>), sgrid, dim3(NTV), slds, stream, dout, demb, emb, AOUT,   \
                       edge_index, E, graph_ptr, edge_ptr, (int)B, npad, slope, act_here, ws.inter, ws.db_slabs, status,   \
                       (int)D, poolbits);                                                                                  \
  } while (0)
#define DISPATCH_GSEG(NTV, NMAXV)                                                                                          \
  do {                                                                                                                     \
    if (bits) LAUNCH_GSEG(NTV, NMAXV, true, false, (const float*)nullptr, true);                                           \
    else if (poolg) LAUNCH_GSEG(NTV, NMAXV, true, false, out, false);                                                      \
    else if (act_here) LAUNCH_GSEG(NTV, NMAXV, false, true, out, false);                                                   \
    else LAUNCH_GSEG(NTV, NMAXV, false, false, (const float*)nullptr, false);                                              \
  } while (0)
    if (npad <= 128) DISPATCH_GSEG(256, 128); else DISPATCH_GSEG(512, 224);
#undef DISPATCH_GSEG
#undef LAUNCH_GSEG
    HCG_CHECK_LAUNCH();
  }
  if (D == 64) {
    const int fp = tall_fpad(F);
    {   // dW slabs = dH^T x
      const dim3 grid(dw_grid(N, 64)), blk(DWT);
      const bool xvec = (F % 4) == 0 && ((uintptr_t)x % 16) == 0;
#define LAUNCH_DW64(NBF, XV)                                                                                          \
  do {                                                                                                               \
    const size_t lds = (size_t)3 * (64 + NBF * 32) * (DW_TILE + 8) * sizeof(short);                                  \
    hipError_t e = allow_lds<k_tall_dw<2, NBF, XV>>(lds);                                                            \
    if (e != hipSuccess) return hcg_hip_err(e);                                                                      \
    hipLaunchKernelGGL((k_tall_dw<2, NBF, XV>), grid, blk, lds, stream, ws.inter, x, (int)F, ws.dw_slabs, (int)N);   \
  } while (0)
      if (fp == 32) { if (xvec) LAUNCH_DW64(1, true); else LAUNCH_DW64(1, false); }
      else          { if (xvec) LAUNCH_DW64(2, true); else LAUNCH_DW64(2, false); }
#undef LAUNCH_DW64
      HCG_CHECK_LAUNCH();
    }
    if (dx) {   // dx = dH W
      const dim3 grid(mm_grid(N)), blk(TT);
      const size_t lds = (size_t)3 * fp * (64 + WPAD) * 2;
      const bool pm = (apply_act & 2) != 0;
#define LAUNCH_MM64(NOB, PM)                                                                                               \
  do {                                                                                                                     \
    hipError_t e = allow_lds<k_tall_mm<64, NOB, true, PM>>(lds);                                                           \
    if (e != hipSuccess) return hcg_hip_err(e);                                                                            \
    hipLaunchKernelGGL((k_tall_mm<64, NOB, true, PM>), grid, blk, lds, stream, ws.inter, (int)D, W, (int)D, (int)F, dx,    \
                       (int)F, x, slope, (int)N);                                                                          \
  } while (0)
      if (fp == 32) { if (pm) LAUNCH_MM64(1, true); else LAUNCH_MM64(1, false); }
      else          { if (pm) LAUNCH_MM64(2, true); else LAUNCH_MM64(2, false); }
#undef LAUNCH_MM64
      HCG_CHECK_LAUNCH();
    }
    return HCG_OK;
  }
  const int fp = tall_fpad(F);
  // dW slabs = dH^T x
  {
    const dim3 grid(dw_grid(N)), blk(DWT);
#define LAUNCH_DW(NBF)                                                                                               \
  do {                                                                                                               \
    const size_t lds = (size_t)3 * (128 + NBF * 32) * (DW_TILE + 8) * sizeof(short);                                 \
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
                                    int first_layer_form, float* dW, float* db, hcg_reduce_job* job_host) {
  if ((D != 128 && D != 64) || F < 1 || F > 128 || N <= 0 || B <= 0 || !dW || !db || !job_host || !workspace) return HCG_ERR_INVALID_ARG;
  if (workspace_bytes < hcg_tall_workspace_bytes(N, B, F, D)) return HCG_ERR_WORKSPACE;
  const TallWs ws = tall_carve(const_cast<void*>(workspace), N, B, F, D);
  const int fp = tall_fpad(F);
  hcg_reduce_job* j = job_host;
  j->slabs = ws.dw_slabs;
  j->nslabs = dw_grid(N, D);
  j->slab_floats = (int32_t)(D * fp);
  j->nseg = 1;
  j->sse_part = nullptr;
  j->reserved = 0;
  j->seg[0] = hcg_reduce_seg{0, (int32_t)(D * fp), fp, (int32_t)F, dW};
  j = job_host + 1;
  j->slabs = ws.db_slabs;
  j->nslabs = first_layer_form ? dw_grid(N, D) : seg_grid64(B);      // (first-layer form: k_tall_dw<FIRST> leaves the db slabs)
  j->slab_floats = (int32_t)D;
  j->nseg = 1;
  j->sse_part = nullptr;
  j->reserved = 0;
  j->seg[0] = hcg_reduce_seg{0, (int32_t)D, 1, 1, db};
  return HCG_OK;
}
