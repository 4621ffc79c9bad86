// Fused per-layer kernels for batches of SMALL graphs (the BASELINE workload: ~30-atom graphs, 64-d).
//
// One GCN layer (reference: PyG GCNConv + nn.LeakyReLU, call sites model/gcn.py:58-63; SURVEY rows
// a3-a8, and a9 for the last layer) is ONE kernel: a tile of whole graphs (<= 32 node rows) is
// brought on chip once, transformed AND aggregated on the matrix cores, biased/activated,
// (pooled) and written once -- the "layer-fused minimum" HBM traffic of SURVEY 8(d).
//
// CDNA4 mapping
//   * wavefront-autonomous tiles: each 64-lane wave owns a stream of tiles and a private LDS region
//     (one [32][68] fp32 tile; the [32][33] adjacency-count matrix of the forward lives in the same
//     bytes until the tile is staged); no workgroup barrier in the steady state.  512-thread
//     workgroup (8 waves = 2 per SIMD) per CU.
//   * gcn_norm without any index structure: the tile's raw COO edges (int64 `edge_index`, grouped by
//     graph as PyG collation emits them) are scattered into C[dst][src] += 1 with LDS integer atomics
//     (order-independent -> deterministic), C += I, deg = row sums, dinv = deg^-1/2.  No CSR, no
//     sort: the fused path needs only graph_ptr / edge_ptr from the batch plan.
//   * every contraction runs on v_mfma_f32_32x32x16_bf16 with SPLIT operands: an f32 value is the sum
//     of three bf16 pieces (round-to-nearest residuals: x = p1 + p2 + p3 to 2^-24 |x|), a product
//     keeps the six cross terms down to 2^-24 (p1q1, p1q2, p2q1, p1q3, p2q2, p3q1), products of bf16
//     pairs are exact in the f32 accumulator.  Measured against fp64 the result is as accurate as the
//     exact-f32 MFMA chain (v_mfma_f32_32x32x2_f32: 1.5e-7 vs 2.1e-7 |err|inf/|ref|inf on a 64-term
//     dot, tools/probe_bf16x.hip) at 1/16 of its cycles per product: 6 bf16 MFMAs replace 8 f32 ones
//     of 2x the cycles each, the f32 matrix rate (64 FLOP/clk/SIMD) having been the compute floor of
//     these kernels.  Small integer operands (the adjacency counts) are exact in ONE piece.
//       H   = X W^T                      X pieces split on the fly from the fp32 LDS tile,
//                                        W pieces pre-split once per workgroup into LDS
//       Y   = (C + I) (dinv . H)         B operand = the H ACCUMULATORS themselves (split in registers):
//                                        an MFMA sums over k, so slot j of k-step s on lane-half h may
//                                        stand for k = krow(8s + j, h) -- exactly the rows accumulator
//                                        register 8s + j holds there
//       out = LeakyReLU(dinv . Y + b)    epilogue / pooling in accumulator layout
//     backward: dH = (C + I)^T (dinv . dY), dW += dH^T X (A operand = dH accumulators, same trick),
//     dX = dH W (one LDS transpose).
//   * backward: dW accumulates in MFMA accumulators ACROSS all tiles of a wave; waves combine
//     through LDS, workgroups through a [grid][D*KPAD+D] slab reduced in a fixed order by a second
//     kernel -> bitwise reproducible gradients.
#include "common.h"
#include "split_mfma.h"
#include "head_tile.h"

// Diagnostic builds only (tools/probe_fused.hip defines HCG_STAMP): s_memtime stamps of a few waves go to
// a buffer of their own; the product build compiles STAMP() to nothing and executes no stamp.
#ifdef HCG_STAMP
__device__ unsigned long long* g_stamp_buf = nullptr;
// stamps are parked in LDS (no vector-memory op in the timed stream) and flushed by STAMP_FLUSH()
#define STAMP_DECL __shared__ unsigned long long s_stamp[8][64];
#define STAMP(idx)                                                                                          \
  do {                                                                                                      \
    __builtin_amdgcn_sched_barrier(0);                                                                      \
    unsigned long long _t;                                                                                  \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(_t)::"memory");                              \
    __builtin_amdgcn_sched_barrier(0);                                                                      \
    if ((threadIdx.x & 63) == 0) s_stamp[threadIdx.x >> 6][(idx)] = _t;                                      \
  } while (0)
#define STAMP_FLUSH()                                                                                       \
  do {                                                                                                      \
    if (g_stamp_buf && blockIdx.x < 4 && (threadIdx.x & 63) == 0)                                           \
      for (int _i = 0; _i < 64; ++_i)                                                                       \
        g_stamp_buf[((size_t)blockIdx.x * 8 + (threadIdx.x >> 6)) * 64 + _i] = s_stamp[threadIdx.x >> 6][_i]; \
  } while (0)
#else
#define STAMP_DECL
#define STAMP_FLUSH() do { } while (0)
#define STAMP(idx) do { } while (0)
#endif
// the backward's stamps (same buffer): only with -DHCG_STAMP -DHCG_STAMP_BWD, and the probe then runs a backward last
#if defined(HCG_STAMP) && defined(HCG_STAMP_BWD)
#define BSTAMP_DECL STAMP_DECL
#define BSTAMP(idx) STAMP(idx)
#define BSTAMP_FLUSH() STAMP_FLUSH()
#else
#define BSTAMP_DECL
#define BSTAMP(idx) do { } while (0)
#define BSTAMP_FLUSH() do { } while (0)
#endif

namespace {

constexpr int TM = 32;          // node rows per wave tile
constexpr int CS = TM + 1;      // row stride of the adjacency-count matrix (conflict-free rows AND columns)
constexpr int CNT_WORDS = 1280; // >= TM * CS, multiple of 256 (zero-filled with 5 int4 stores per lane)
constexpr int WAVES = 8;        // waves per workgroup
constexpr int BUF_FLOATS = TM * HS;
constexpr int CNT_EXACT = 256;  // adjacency counts up to here are exact in one bf16 piece
static_assert(CNT_WORDS * 4 <= BUF_FLOATS * 4, "the count matrix must fit inside the tile buffer it aliases");

// per-wave LDS.  Forward: the count matrix occupies the first CNT_WORDS words of `buf` while the tile's
// adjacency fragments are being built, then the x tile is staged over it.  Backward: `cnt` is separate.
struct WaveLdsF {
  float buf[BUF_FLOATS];   // adjacency counts, then the x tile / activated output tile
  float ldinv[TM];         // in-degree counter while the edges are scattered, then (1 + deg)^-1/2
  int lgp[TM + 4];         // node offset of every graph of the tile (a tile holds <= TM graphs)
};
struct WaveLdsB {
  float buf[BUF_FLOATS];   // dY', then x tile, then dH (for the dX transpose), then dX (wide stores)
  int cnt[CNT_WORDS];      // C + I of the tile: cnt[dst * CS + src]
  float ldinv[TM];
  int lgp[TM + 4];
};

// All four tile scalars come from ONE round of (scalar) loads of the blocked plan's graph_ptr / edge_ptr.
// Rule for every global load in this file: never guard a load with a per-lane branch (hipcc then
// serialises it behind its own s_waitcnt vmcnt(0)); clamp the index into range and select afterwards.
struct TileInfo {
  int g0, g1, nbase, n, ebase, ne;
};

struct TileRaw { int g0, g1, p0, p1, e0, e1; };   // the four loaded words, before any arithmetic

// loads only (scalar: `t` is wave-uniform) -- kept apart from tile_finish() so that the scalars of the tile
// AFTER the next one can be in flight for a whole tile: as a load-then-use sequence in front of the
// prefetch they cost ~4.5k cycles per tile (scalar-cache misses while every wave of the chip asks at once)
__device__ __forceinline__ TileRaw tile_raw(int t, int num_tiles, int gpt, int B, const int32_t* __restrict__ graph_ptr,
                                            const int32_t* __restrict__ edge_ptr) {
  TileRaw w;
  t = __builtin_amdgcn_readfirstlane(t < num_tiles ? t : num_tiles - 1);
  w.g0 = t * gpt;
  w.g1 = w.g0 + gpt < B ? w.g0 + gpt : B;
  w.p0 = graph_ptr[w.g0];
  w.p1 = graph_ptr[w.g1];
  w.e0 = edge_ptr[w.g0];
  w.e1 = edge_ptr[w.g1];
  return w;
}

// The tiles of one wave, round by round.  Full rounds: tile k * stride + 8 b + w (a workgroup's tiles of a round are
// consecutive).  The LAST, partial round is dealt round-robin over the workgroups instead (position p = w * G + b): with the
// plain formula its tiles went to the first workgroups' eight waves each while the other CUs idled -- 2 826 tiles on 2 048
// waves (the ragged batch's small graphs) took two full rounds; dealt evenly every CU runs three of its waves a second time,
// alone on their SIMDs.  `num_tiles` = no tile.
// The host picks DEAL only when the last round IS partial (4096 tiles on 2048 waves: the plain form, whose code is the
// one the headline configuration was tuned with -- the dealt form's extra scalars cost its kernels 1-1.7 us).
template <bool DEAL>
struct TileSeq {
  int kf, rem, stride, first, last_p;
  __device__ __forceinline__ TileSeq(int num_tiles, int wave) {
    const int G = (int)gridDim.x;
    stride = G * WAVES;
    first = (int)blockIdx.x * WAVES + wave;
    if constexpr (DEAL) {
      kf = num_tiles / stride;
      rem = num_tiles - kf * stride;
      last_p = wave * G + (int)blockIdx.x;
    } else {
      kf = rem = last_p = 0;
    }
  }
  // tile of round k; `prev` = the tile of round k - 1 (the plain form only adds the stride, as it always did)
  __device__ __forceinline__ int at(int k, int prev, int num_tiles) const {
    if constexpr (!DEAL) return prev + stride;
    if (k < kf) return k * stride + first;
    return (k == kf && last_p < rem) ? kf * stride + last_p : num_tiles;
  }
};

__device__ __forceinline__ TileInfo tile_finish(const TileRaw& w, int gpt, int lane, int32_t* status) {
  TileInfo ti;
  ti.g0 = w.g0;
  ti.g1 = w.g1;
  ti.nbase = w.p0;
  ti.n = w.p1 - w.p0;
  ti.ebase = w.e0;
  ti.ne = w.e1 - w.e0;
  if (ti.n > TM || ti.n < 0 || ti.ne < 0 || gpt > TM) {  // host metadata was wrong: refuse the tile
    if (lane == 0) atomicOr(status, HCG_STATUS_SHAPE_LIMIT);
    ti.n = 0;
    ti.ne = 0;
    ti.g1 = ti.g0;
  }
  return ti;
}

// per-lane index data of a tile (first 64 edges + graph offsets), loaded into registers by load() and
// turned into the LDS count matrix by build() -- split so the NEXT tile can be in flight while the
// current one computes.
struct TileEdges {
  long long es, ed;
  int gp_raw;

  // Loads ONLY: no arithmetic on a loaded value and no branch in here -- either makes hipcc place an
  // s_waitcnt vmcnt(0) right behind the loads, which turns the next-tile prefetch into a stall.
  // (E >= 1 and `ei` readable are guaranteed by the host wrappers, also for edge-less batches.)
  __device__ __forceinline__ void load(const TileInfo& ti, const int32_t* __restrict__ graph_ptr,
                                       const int64_t* __restrict__ ei, int64_t E, int lane) {
    const int ng = ti.g1 - ti.g0;
    gp_raw = graph_ptr[ti.g0 + (lane <= ng ? lane : ng)];
    int64_t k = (int64_t)ti.ebase + (lane < ti.ne ? lane : (ti.ne > 0 ? ti.ne - 1 : 0));
    if (k > E - 1) k = E - 1;
    es = ei[k];
    ed = ei[E + k];
  }

  // C = I + sum_e [dst_e][src_e],  ldinv = (row sum)^-1/2 ; edges beyond the first 64 are read here
  __device__ __forceinline__ void build(int* cnt, float* ldinv, int* lgp, const TileInfo& ti,
                                        const int64_t* __restrict__ ei, int64_t E, int lane, int32_t* status) const {
#pragma unroll
    for (int j = 0; j < CNT_WORDS / 256; ++j)
      *reinterpret_cast<int4*>(&cnt[(lane + 64 * j) * 4]) = make_int4(0, 0, 0, 0);
    if (lane < TM) ldinv[lane] = 0.f;
    int* degc = reinterpret_cast<int*>(ldinv);
    if (lane < ti.n) cnt[lane * CS + lane] = 1;                         // self loop, weight 1
    const int ng = ti.g1 - ti.g0;
    if (lane <= ng) lgp[lane] = gp_raw - ti.nbase;
    for (int k0 = 0; k0 < ti.ne; k0 += 64) {
      long long s = es, d = ed;
      if (k0 > 0) {   // wave-uniform: only tiles with more than 64 edges come here
        int64_t k = (int64_t)ti.ebase + (k0 + lane < ti.ne ? k0 + lane : ti.ne - 1);
        if (k > E - 1) k = E - 1;
        s = ei[k];
        d = ei[E + k];
      }
      // local ids in 32 bits; one unsigned compare each covers "< 0" and ">= n" (ids are < 2^31)
      const unsigned sl = (unsigned)((int)s - ti.nbase), dl = (unsigned)((int)d - ti.nbase);
      const bool live = k0 + lane < ti.ne;
      const bool ok = sl < (unsigned)ti.n && dl < (unsigned)ti.n && (s >> 31) == 0 && (d >> 31) == 0;
      // an explicit (i, i) edge collapses into the unit self loop every node already has (PyG
      // add_remaining_self_loops): it is neither counted nor added
      if (live && ok && sl != dl) {
        atomicAdd(&cnt[dl * CS + sl], 1);                               // ds_add_u32: exact, order-independent
        atomicAdd(&degc[dl], 1);
      }
      if (__ballot(live && !ok) != 0ull) {                              // edge leaves its graph: ignored, flagged
        if (lane == 0) atomicOr(status, HCG_STATUS_EDGE_UNGROUPED);
      }
    }
    if (lane < TM) {
      const float deg = 1.0f + (float)degc[lane];
      ldinv[lane] = lane < ti.n ? 1.0f / sqrtf(deg) : 0.f;
    }
  }
};

// The tile adjacency as the (exact, one-piece) A operand of the aggregation MFMAs, k-steps s = 0, 1.
//   forward  (Y  = (C + I) H'):    A[m = r][slot j] = C[r][krow(8s + j, h)]   (k order of the H accumulators)
//   backward (dH = (C + I)^T dY'): A[m = r][slot j] = C[16s + 8h + j][r]       (natural k order, B read from LDS)
// Counts above CNT_EXACT (that many parallel edges between one pair of atoms) are not exact in bf16: flagged.
struct AdjFrags {
  bf16x8 f[2];
  template <bool TRANSPOSED>
  __device__ __forceinline__ void read(const int* cnt, int r, int h, int lane, int32_t* status) {
    int worst = 0;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      u32x4 u;
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        const int k0 = TRANSPOSED ? 16 * s + 8 * h + 2 * jj : krow(8 * s + 2 * jj, h);
        const int k1 = TRANSPOSED ? k0 + 1 : krow(8 * s + 2 * jj + 1, h);
        const int c0 = TRANSPOSED ? cnt[k0 * CS + r] : cnt[r * CS + k0];
        const int c1 = TRANSPOSED ? cnt[k1 * CS + r] : cnt[r * CS + k1];
        worst = worst > c0 ? worst : c0;
        worst = worst > c1 ? worst : c1;
        u[jj] = pk_bf16((float)c0, (float)c1);
      }
      f[s] = __builtin_bit_cast(bf16x8, u);
    }
    if (__ballot(worst > CNT_EXACT) != 0ull) {
      if (lane == 0) atomicOr(status, HCG_STATUS_SHAPE_LIMIT);
    }
  }
};

__device__ __forceinline__ float4 f4_add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 f4_shfl_xor(float4 v, int m) {
  return make_float4(__shfl_xor(v.x, m, 64), __shfl_xor(v.y, m, 64), __shfl_xor(v.z, m, 64), __shfl_xor(v.w, m, 64));
}

// Stage n rows of a row-major [Nrows, F] global matrix into buf[row][0..KPAD), rows >= n and columns
// >= F zero-filled, in two phases so a caller can put every load of a tile in flight before the
// first LDS write: load() only issues global loads into registers, write() only touches LDS.
// VEC: F == KPAD and 16-byte aligned rows -> one float4 per lane, 8 loads in flight; else the tile's
// contiguous element block [nbase*F, (nbase+n)*F) is read with coalesced scalar loads.
template <int KPAD, bool VEC>
struct Stager {
  static constexpr int NIT = VEC ? TM / 4 : TM * KPAD / 64;
  float4 v4[VEC ? TM / 4 : 1];
  float v1[VEC ? 1 : TM * KPAD / 64];

  __device__ __forceinline__ void load(const float* __restrict__ g, int F, int64_t Nrows, int nbase, int n, int lane) {
    if constexpr (VEC) {
      const int q = lane & 15, r4 = lane >> 4;
#pragma unroll
      for (int it = 0; it < TM / 4; ++it) {
        int64_t row = (int64_t)nbase + it * 4 + r4;
        if (row > Nrows - 1) row = Nrows - 1;
        v4[it] = *reinterpret_cast<const float4*>(g + (size_t)row * F + (4 * q < KPAD ? 4 * q : 0));
      }
    } else {
      const int total = n * F;
      const float* base = g + (size_t)nbase * F;
#pragma unroll
      for (int j = 0; j < NIT; ++j) {
        const int idx = lane + 64 * j;
        v1[j] = base[idx < total ? idx : (total > 0 ? total - 1 : 0)];
      }
    }
  }

  __device__ __forceinline__ void write(float* buf, int F, int n, int lane) {
    if constexpr (VEC) {
      const int q = lane & 15, r4 = lane >> 4;
#pragma unroll
      for (int it = 0; it < TM / 4; ++it) {
        const int row = it * 4 + r4;
        float4 v = v4[it];
        if (row >= n) v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (4 * q < KPAD) *reinterpret_cast<float4*>(buf + row * HS + 4 * q) = v;
      }
    } else {
      const int total = n * F;
      for (int idx = lane; idx < TM * KPAD; idx += 64) buf[(idx / KPAD) * HS + (idx % KPAD)] = 0.f;
#pragma unroll
      for (int j = 0; j < NIT; ++j) {
        const int idx = lane + 64 * j;
        if (idx < total) {
          const int row = idx / F, k = idx - row * F;
          buf[row * HS + k] = v1[j];
        }
      }
    }
  }
};

// =====================================================================================================
// forward:  out = LeakyReLU( Ahat (x W^T) + b ),  optional pooled epilogue emb[g] = [max, mean]
// =====================================================================================================
// STACK2: TWO conv layers per launch -- the first layer's output tile is already in LDS in exactly the
// layout the next GEMM reads (the epilogue puts it there for the wide stores), the adjacency fragments and
// dinv are shared, and the second launch's prologue / tile staging / count build disappear.  POOL then
// refers to layer 2.
// BITS (training, needs POOL): the pooled layer's activations do NOT go to HBM; what the pooled backward needs of
// them leaves as two bits per element, in the accumulator layout: poolbits[tile][0][lane] bit 16 b + i <-> value at
// (row krow(i, h), column 32 b + r) is > 0, poolbits[tile][1][lane] the same positions: value == its graph's column max.
// HEAD (needs POOL): the regression head rides in the TAIL of this launch (head_tile.h).  A workgroup's tiles are
// TileSeq's: {8 b + w + k * 8 grid} in the full rounds -- runs of 8 gpt consecutive graphs per round k -- and the dealt ones
// of the last round; once its waves have left the tile loop their LDS is
// free, the pooled rows they wrote are visible to the whole workgroup (workgroup-scope barrier), and the workgroup runs
// readout forward, squared error and -- HEAD == 2 -- the UNSCALED readout backward over its own graphs (head_tile.h:
// hcg_head16, 16-row tiles on all 8 waves): z, out, demb and one gradient slab + SSE partial per workgroup.  Round 2 spent a launch of its own (15 us at C3: a grid-wide exchange of
// one scalar on 128 of 256 CUs) on this; the scalar is applied by the step's last launch now (reduce.hip).
struct FwdHead {
  const float* y;
  const float* W0;
  const float* b0;
  const float* W1;
  const float* b1;
  int C;
  float* z;
  float* out;
  float* demb;
  float* slabs;
  int* step_counter;
};

constexpr int SEMB_ROWS = 32;    // pooled rows a workgroup keeps in LDS for its head tail (C3: 16 graphs per workgroup)

template <int RC, bool BACKWARD, bool DEAL>
__device__ __forceinline__ void fwd_head_tail(void* lds_base, const float* semb, const FwdHead& H, const float* __restrict__ emb,
                                              int gpt, int B, int num_tiles, float slope) {
  using namespace hcg_head16;
  H16STAMP(0);
  Prefetch<RC> P;              // the head's weights are requested while the slower waves finish their tiles
  prefetch<RC>(P, H.W0, H.b0, H.W1, H.C);
  State<RC> S;
  begin<RC>(S, H.b1, H.C);
  __syncthreads();             // every wave has left its tiles: LDS free, this workgroup's pooled rows visible to all of it
  H16STAMP(1);
  Lds& HL = *reinterpret_cast<Lds*>(lds_base);
  const int R = WAVES * gpt;                               // graphs of this workgroup per FULL round of its waves
  const int G = (int)gridDim.x, stride = G * WAVES;        // tiles between two rounds
  // TileSeq: full rounds, tiles of the dealt last round (not DEAL: every round, the last one included, is a "full" one whose
  // run of tiles may end early)
  const int kf = DEAL ? num_tiles / stride : 0;
  const int rem = DEAL ? num_tiles - kf * stride : 0;
  int rows_total = 0;
  if constexpr (DEAL) {
    for (int k = 0; k < kf; ++k) {
      const int left = B - (k * stride + (int)blockIdx.x * WAVES) * gpt;
      rows_total += left < R ? left : R;
    }
  } else {
    for (int tb = blockIdx.x * WAVES; tb < num_tiles; tb += stride) {
      const int left = B - tb * gpt;
      rows_total += left < R ? left : R;
    }
  }
  const int rows_full = rows_total;
  for (int p = (int)blockIdx.x; p < rem; p += G) {         // the last round's tiles of this workgroup: waves 0, 1, ...
    const int left = B - (kf * stride + p) * gpt;
    rows_total += left < gpt ? left : gpt;
  }
  const int base = blockIdx.x * WAVES * gpt, step_g = stride * gpt, base_last = (kf * stride + (int)blockIdx.x) * gpt;
  const bool in_lds = rows_total <= SEMB_ROWS;             // block-uniform
  H16STAMP(2);
  for (int j0 = 0; j0 < rows_total; j0 += RT) {
    const int n = rows_total - j0 < RT ? rows_total - j0 : RT;
    tile<RC, BACKWARD>(HL, S, P,
                       [=](int row) {
                         const int j = j0 + row;
                         if (j < rows_full) { const int k = j / R; return base + k * step_g + (j - k * R); }
                         const int jj = j - rows_full, w = jj / gpt;        // (only the batch's very last tile can be short: it is
                         return base_last + w * G * gpt + (jj - w * gpt);   //  the last slot of its workgroup)
                       },
                       n, H.C, slope, in_lds ? semb + j0 * ES : nullptr, emb, H.y, H.z, H.out, H.demb, j0 == 0);
  }
  end<RC, BACKWARD>(HL, S, H.C, H.slabs + (size_t)blockIdx.x * SLAB, H.slabs + (size_t)gridDim.x * SLAB + blockIdx.x);
}

template <int KPAD, bool VEC, bool POOL, bool STACK2, bool BITS = false, int HEAD = 0, bool DEAL = false>
__global__ __launch_bounds__(WAVES * 64, 2) void k_fused_layer_fwd(
    const float* __restrict__ x, int F, const float* __restrict__ W, const float* __restrict__ bias,
    const float* __restrict__ W2, const float* __restrict__ bias2,
    const int64_t* __restrict__ ei, int64_t E, const int32_t* __restrict__ graph_ptr,
    const int32_t* __restrict__ edge_ptr, int64_t N, int gpt, int B, int num_tiles, float slope, int apply_act,
    float* __restrict__ out, float* __restrict__ out2, float* __restrict__ emb, uint32_t* __restrict__ poolbits,
    int32_t* __restrict__ status, FwdHead HA = FwdHead{}) {
  static_assert(!BITS || POOL, "the bit form belongs to the pooled layer");
  static_assert(HEAD == 0 || POOL, "the head reads the pooled embedding");
  __shared__ WaveLdsF lds[WAVES];
  static_assert(HEAD == 0 || sizeof(hcg_head16::Lds) <= sizeof(WaveLdsF) * WAVES, "the head's LDS aliases the tile buffers");
  __shared__ float semb[HEAD != 0 ? SEMB_ROWS * hcg_head16::ES : 4];   // this workgroup's pooled rows, for its head tail
  if (HEAD != 0 && HA.step_counter && blockIdx.x == 0 && threadIdx.x == 0) { HA.step_counter[0] += 1; HA.step_counter[1] += 1; }   // this step's number | the exchange stamp (never re-based)
  __shared__ __attribute__((aligned(16))) short w1l[3 * DD * (KPAD + WPAD)];
  __shared__ __attribute__((aligned(16))) short w2l[STACK2 ? 3 * DD * (DD + WPAD) : 8];
  STAMP_DECL
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  WaveLdsF& L = lds[wave];
  int* cnt = reinterpret_cast<int*>(L.buf);
  const int r = lane & 31, h = lane >> 5;
  const TileSeq<DEAL> seq(num_tiles, wave);
  int t = DEAL ? seq.at(0, 0, num_tiles) : seq.first;
  bool have = t < num_tiles;

  // first tile's loads go out before anything else
  TileInfo ti;
  TileEdges te;
  Stager<KPAD, VEC> sx;
  TileRaw raw_next;   // scalars of the tile after the current one, always one tile ahead
  if (have) {
    const TileRaw raw0 = tile_raw(t, num_tiles, gpt, B, graph_ptr, edge_ptr);
    raw_next = tile_raw(seq.at(1, t, num_tiles), num_tiles, gpt, B, graph_ptr, edge_ptr);
    ti = tile_finish(raw0, gpt, lane, status);
    sx.load(x, F, N, ti.nbase, ti.n, lane);
    te.load(ti, graph_ptr, ei, E, lane);
  }

  // weight operands: W [64][F] -> three bf16 planes in LDS, once per workgroup
  // (bias loads first: behind the staging they were one more memory round trip in front of the first tile)
  float b0 = bias[r], b1 = bias[32 + r];
  float c0 = STACK2 ? bias2[r] : 0.f, c1 = STACK2 ? bias2[32 + r] : 0.f;
  stage_weight_split<false, WAVES * 64, DD, KPAD>(w1l, W, DD, F);
  if (STACK2) stage_weight_split<false, WAVES * 64, DD, DD>(w2l, W2, DD, DD);
  const float slope_eff = apply_act ? slope : 1.0f;
  // retire the bias loads HERE: left pending, their first use (in the epilogue of the tile loop) makes
  // hipcc wait on the vector-memory counter there, which also drains the next-tile prefetch
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(b0), "+v"(b1), "+v"(c0), "+v"(c1));
  __syncthreads();
  STAMP(0);
  int stamp_it = 0;
  // adjacency fragments + dinv of this lane's 16 accumulator rows: read ONCE per tile into registers, shared by
  // both layers of a stacked launch.  Order per tile: count matrix (in the tile buffer's bytes) -> fragments
  // -> x tile staged over it.
  AdjFrags adj;
  float dvr[16];
  if (have) {
    te.build(cnt, L.ldinv, L.lgp, ti, ei, E, lane, status);
    adj.read<false>(cnt, r, h, lane, status);
#pragma unroll
    for (int i = 0; i < 16; ++i) dvr[i] = L.ldinv[krow(i, h)];
    sx.write(L.buf, F, ti.n, lane);
  }
  int round_it = 0;      // rounds of this wave so far (tile t = seq.at(round_it))
  while (have) {
    STAMP(1 + 8 * stamp_it);
    // prefetch the next tile of this wave (registers only) while this one computes
    const int tn = seq.at(round_it + 1, t, num_tiles);
    const bool have_next = tn < num_tiles;
    TileInfo tin;
    TileEdges ten;
    Stager<KPAD, VEC> sxn;
    const TileRaw raw_cur = raw_next;                                    // loaded one tile ago
    raw_next = tile_raw(seq.at(round_it + 2, tn, num_tiles), num_tiles, gpt, B, graph_ptr, edge_ptr);   // consumed one tile from now
    if (VEC && have_next) {   // (the scalar-staging variants are short of registers: they load after the compute)
      tin = tile_finish(raw_cur, gpt, lane, status);
      sxn.load(x, F, N, tin.nbase, tin.n, lane);
      ten.load(tin, graph_ptr, ei, E, lane);
    }
    STAMP(2 + 8 * stamp_it);

    f32x16 y0, y1;
#pragma unroll
    for (int layer = 0; layer < (STACK2 ? 2 : 1); ++layer) {
      // ---- H = X W^T
      f32x16 acc0, acc1;
#pragma unroll
      for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }
      if constexpr (STACK2) {
        if (layer == 0) tile_gemm_split<KPAD>(L.buf, w1l, acc0, acc1, lane);
        else tile_gemm_split<DD>(L.buf, w2l, acc0, acc1, lane);
      } else {
        tile_gemm_split<KPAD>(L.buf, w1l, acc0, acc1, lane);
      }
      if (layer == 0) STAMP(3 + 8 * stamp_it);
      if (layer == 1 && stamp_it == 0) STAMP(41);
      mfma_results_fence(acc0, acc1);

      // ---- H' = dinv (.) H  (in the accumulators), Y = (C + I) H': B operand = the H' accumulators, split in
      //      registers; slot j of k-step s <-> accumulator register 8s + j
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        acc0[i] *= dvr[i];
        acc1[i] *= dvr[i];
        y0[i] = 0.f;
        y1[i] = 0.f;
      }
      if (layer == 0) STAMP(4 + 8 * stamp_it);
      if (layer == 1 && stamp_it == 0) STAMP(42);
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const float h0[8] = {acc0[8 * s], acc0[8 * s + 1], acc0[8 * s + 2], acc0[8 * s + 3],
                             acc0[8 * s + 4], acc0[8 * s + 5], acc0[8 * s + 6], acc0[8 * s + 7]};
        const float h1[8] = {acc1[8 * s], acc1[8 * s + 1], acc1[8 * s + 2], acc1[8 * s + 3],
                             acc1[8 * s + 4], acc1[8 * s + 5], acc1[8 * s + 6], acc1[8 * s + 7]};
        mfma_exact_a(y0, adj.f[s], split3(h0));
        mfma_exact_a(y1, adj.f[s], split3(h1));
      }
      if (layer == 0) STAMP(6 + 8 * stamp_it);
      if (layer == 1 && stamp_it == 0) STAMP(43);
      mfma_results_fence(y0, y1);

      // ---- out = LeakyReLU(dinv (.) Y + b): accumulator layout, column = lane (feature), rows in registers.
      //      The values go back through the (now dead) input tile so that the HBM stores are 8 row-contiguous
      //      dwordx4 per lane instead of 64 exec-masked dword stores -- and, stacked, so that they ARE the
      //      next layer's input tile.
      const float bb0 = layer == 0 ? b0 : c0, bb1 = layer == 0 ? b1 : c1;
      const bool to_hbm = !(BITS && layer == (STACK2 ? 1 : 0));   // (folds: the layer loop is unrolled)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int row = krow(i, h);
        float v0 = fmaf(y0[i], dvr[i], bb0), v1 = fmaf(y1[i], dvr[i], bb1);
        // LeakyReLU as max(v, slope*v): exact for 0 <= slope <= 1 (the host rejects other slopes)
        v0 = fmaxf(v0, slope_eff * v0);          // no activation: slope_eff = 1 -> max(v, v) = v (no per-element select)
        v1 = fmaxf(v1, slope_eff * v1);
        y0[i] = v0;
        y1[i] = v1;
        if (to_hbm) {
          L.buf[row * HS + r] = v0;
          L.buf[row * HS + 32 + r] = v1;
        }
      }
      if (layer == 0) STAMP(7 + 8 * stamp_it);
      if (layer == 1 && stamp_it == 0) STAMP(44);
      if (to_hbm && ti.n > 0) {   // wave-uniform
        // rows >= n are redirected to row n-1 (read AND write): duplicate identical stores instead of a
        // per-lane branch around every store -> all 8 LDS reads and 8 stores stay in one basic block
        const int q = lane & 15, r4 = lane >> 4;
        float* dst = layer == 0 ? out : out2;
        float4 ov[TM / 4];
#pragma unroll
        for (int it = 0; it < TM / 4; ++it) {
          const int row = it * 4 + r4 < ti.n ? it * 4 + r4 : ti.n - 1;
          ov[it] = *reinterpret_cast<const float4*>(L.buf + row * HS + 4 * q);
        }
#pragma unroll
        for (int it = 0; it < TM / 4; ++it) {
          const int row = it * 4 + r4 < ti.n ? it * 4 + r4 : ti.n - 1;
          *reinterpret_cast<float4*>(dst + (size_t)(ti.nbase + row) * DD + 4 * q) = ov[it];
        }
      }
      if (stamp_it == 0) STAMP(layer == 0 ? 40 : 45);
    }
    if (POOL) {
      uint32_t pos = 0, ismax = 0;
      if (BITS) {
#pragma unroll
        for (int i = 0; i < 16; ++i) pos |= (uint32_t)(y0[i] > 0.f) << i | (uint32_t)(y1[i] > 0.f) << (16 + i);
      }
      for (int g = ti.g0; g < ti.g1; ++g) {
        const int gb = L.lgp[g - ti.g0], ge = L.lgp[g - ti.g0 + 1];
        float m0 = -INFINITY, m1 = -INFINITY, s0 = 0.f, s1 = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int row = krow(i, h);
          if (row >= gb && row < ge) { m0 = fmaxf(m0, y0[i]); m1 = fmaxf(m1, y1[i]); s0 += y0[i]; s1 += y1[i]; }
        }
        m0 = fmaxf(m0, __shfl_xor(m0, 32, 64));
        m1 = fmaxf(m1, __shfl_xor(m1, 32, 64));
        s0 += __shfl_xor(s0, 32, 64);
        s1 += __shfl_xor(s1, 32, 64);
        if (BITS) {
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const int row = krow(i, h);
            if (row >= gb && row < ge) ismax |= (uint32_t)(y0[i] == m0) << i | (uint32_t)(y1[i] == m1) << (16 + i);
          }
        }
        if (h == 0) {
          const int n = ge - gb;
          const float cntf = (float)(n > 0 ? n : 1);
          if (n <= 0) { m0 = 0.f; m1 = 0.f; }
          float* e = emb + (size_t)g * 2 * DD;
          e[r] = m0;
          e[32 + r] = m1;
          e[DD + r] = s0 / cntf;
          e[DD + 32 + r] = s1 / cntf;
          if constexpr (HEAD != 0) {      // row of this graph among the workgroup's graphs (round, wave, graph of the tile)
            const int j = round_it * WAVES * gpt + wave * gpt + (g - ti.g0);
            if (j < SEMB_ROWS) {
              float* se = semb + j * hcg_head16::ES;
              se[r] = m0;
              se[32 + r] = m1;
              se[DD + r] = s0 / cntf;
              se[DD + 32 + r] = s1 / cntf;
            }
          }
        }
      }
      if (BITS) {
        poolbits[(size_t)t * 128 + lane] = pos;
        poolbits[(size_t)t * 128 + 64 + lane] = ismax;
      }
    }
    STAMP(5 + 8 * stamp_it);
    if (stamp_it < 6) ++stamp_it;

    have = have_next;
    ++round_it;
    if (have_next) {
      if (!VEC) {
        tin = tile_finish(raw_cur, gpt, lane, status);
        sxn.load(x, F, N, tin.nbase, tin.n, lane);
        ten.load(tin, graph_ptr, ei, E, lane);
      }
      t = tn;
      ti = tin;
      STAMP(56);
      ten.build(cnt, L.ldinv, L.lgp, ti, ei, E, lane, status);
      adj.read<false>(cnt, r, h, lane, status);
#pragma unroll
      for (int i = 0; i < 16; ++i) dvr[i] = L.ldinv[krow(i, h)];
      STAMP(57);
      sxn.write(L.buf, F, ti.n, lane);
      STAMP(58);
    }
  }
  STAMP(63);
  STAMP_FLUSH();
  if constexpr (HEAD != 0) {
    if (HA.C == 1) fwd_head_tail<1, HEAD == 2, DEAL>(&lds[0], semb, HA, emb, gpt, B, num_tiles, slope);
    else fwd_head_tail<hcg_head::RCMAX, HEAD == 2, DEAL>(&lds[0], semb, HA, emb, gpt, B, num_tiles, slope);
  }
}

// =====================================================================================================
// backward of one layer.
//   dY = dA (.) leaky'(A)            dA = dout, or (POOLG) the pooled-gradient expansion
//   db += colsum dY ;  dH = Ahat^T dY ;  dW += dH^T x ;  dx = dH W  (NEEDS_DX)
// per-workgroup partial sums go to `partials[blockIdx][64*KPAD + 64]`.
// =====================================================================================================
// BITS (needs POOLG): the layer's output was never stored; its sign / is-the-column-max bits (`poolbits`, written by the
// BITS forward over the same tiles) stand in for a_out and emb.
template <int KPAD, bool VEC, bool NEEDS_DX, bool POOLG, bool BITS = false, bool DEAL = false>
__global__ __launch_bounds__(WAVES * 64, 2) void k_fused_layer_bwd(
    const float* __restrict__ dout, const float* __restrict__ demb, const float* __restrict__ emb,
    const float* __restrict__ a_out, const uint32_t* __restrict__ poolbits, const float* __restrict__ x, int F,
    const float* __restrict__ W,
    const int64_t* __restrict__ ei, int64_t E, const int32_t* __restrict__ graph_ptr,
    const int32_t* __restrict__ edge_ptr, int64_t N, int gpt, int B, int num_tiles, float slope, int apply_act,
    float* __restrict__ dx, float* __restrict__ partials, int32_t* __restrict__ status) {
  __shared__ WaveLdsB lds[WAVES];
  // dx operand: image row f, column d <- W[d][f], three bf16 planes, shared by the 8 waves
  __shared__ __attribute__((aligned(16))) short wtl[NEEDS_DX ? 3 * KPAD * (DD + WPAD) : 8];
  BSTAMP_DECL
  int bstamp_it = 0;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  WaveLdsB& L = lds[wave];
  const int r = lane & 31, h = lane >> 5, q = lane & 15, r4 = lane >> 4;
  constexpr int NBF = KPAD / 32;  // column blocks of dW (input-feature dimension)
  const float slope_eff = (apply_act & 1) ? slope : 1.0f;
  // bit 1 of apply_act: dx leaves this kernel already multiplied by leaky'(x) -- x is the previous layer's activated
  // output, so that layer's backward runs with bit 0 clear and never reads its own output (a_out == nullptr there)
  static_assert(!BITS || POOLG, "the bit form is the pooled backward");
  const bool premask = NEEDS_DX && (apply_act & 2);
  const bool use_out = !BITS && a_out != nullptr;
  const TileSeq<DEAL> seq(num_tiles, wave);
  int round_it = 0;      // rounds of this wave so far (tile t = seq.at(round_it))

  // first tile's loads go out before anything else
  int t = DEAL ? seq.at(0, 0, num_tiles) : seq.first;
  bool have = t < num_tiles;
  TileInfo ti;
  TileRaw raw_next;
  Stager<DD, true> sa{}, sd;
  uint32_t pos = 0, ismax = 0;
  TileEdges te;
  if (have) {
    const TileRaw raw0 = tile_raw(t, num_tiles, gpt, B, graph_ptr, edge_ptr);
    raw_next = tile_raw(seq.at(1, t, num_tiles), num_tiles, gpt, B, graph_ptr, edge_ptr);
    ti = tile_finish(raw0, gpt, lane, status);
    if (BITS) {
      pos = poolbits[(size_t)t * 128 + lane];
      ismax = poolbits[(size_t)t * 128 + 64 + lane];
    } else if (POOLG || use_out) {
      sa.load(a_out, DD, N, ti.nbase, ti.n, lane);
    }
    if (!POOLG) sd.load(dout, DD, N, ti.nbase, ti.n, lane);
    te.load(ti, graph_ptr, ei, E, lane);
  }

  if (NEEDS_DX) {
    stage_weight_split<true, WAVES * 64, KPAD, DD>(wtl, W, DD, F);
    __syncthreads();
  }

  f32x16 dw[2][NBF];  // dW[d-block][f-block], accumulated over every tile of this wave
#pragma unroll
  for (int mb = 0; mb < 2; ++mb)
#pragma unroll
    for (int nb = 0; nb < NBF; ++nb)
#pragma unroll
      for (int i = 0; i < 16; ++i) dw[mb][nb][i] = 0.f;
  float4 dbacc = make_float4(0.f, 0.f, 0.f, 0.f);
  float dbl0 = 0.f, dbl1 = 0.f;   // BITS: column r / 32 + r of db (this lane's half of the rows)
  BSTAMP(0);

  while (have) {
    te.build(L.cnt, L.ldinv, L.lgp, ti, ei, E, lane, status);
    BSTAMP(1 + 8 * bstamp_it);

    // ---- 1. dY' = dinv (.) dA (.) leaky'(A) -> buf (rows >= n zero); A / dA stay in registers
    float4 dy[TM / 4];
    if constexpr (BITS) {
      // accumulator layout (lane = column, 16 rows in registers): the pooled gradient is two scalars per lane and graph
      // every row < n belongs to exactly one graph of the tile: each graph writes its own rows of dY', the rest are zero
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int row = krow(i, h);
        if (row >= ti.n) { L.buf[row * HS + r] = 0.f; L.buf[row * HS + 32 + r] = 0.f; }
      }
      for (int g = ti.g0; g < ti.g1; ++g) {
        const int gb = L.lgp[g - ti.g0], ge = L.lgp[g - ti.g0 + 1];
        if (ge <= gb) continue;
        const float* de = demb + (size_t)g * 2 * DD;
        const float cntf = (float)(ge - gb);
        // (requesting these four ahead of the count build was measured: 12 spilled registers, +1.8 us)
        const float dmx0 = de[r], dmx1 = de[32 + r], dme0 = de[DD + r] / cntf, dme1 = de[DD + 32 + r] / cntf;
        uint32_t rows = 0;
#pragma unroll
        for (int i = 0; i < 16; ++i) rows |= (uint32_t)(krow(i, h) >= gb && krow(i, h) < ge) << i;
        float t0 = (float)__builtin_popcount(ismax & rows), t1 = (float)__builtin_popcount((ismax >> 16) & rows);
        t0 += __shfl_xor(t0, 32, 64);
        t1 += __shfl_xor(t1, 32, 64);
        const float share0 = dmx0 / fmaxf(t0, 1.f), share1 = dmx1 / fmaxf(t1, 1.f);   // ties of the max split evenly
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          if (rows >> i & 1) {
            const int row = krow(i, h);
            const float g0 = (dme0 + ((ismax >> i & 1) ? share0 : 0.f)) * ((pos >> i & 1) ? 1.f : slope_eff);
            const float g1 = (dme1 + ((ismax >> (16 + i) & 1) ? share1 : 0.f)) * ((pos >> (16 + i) & 1) ? 1.f : slope_eff);
            dbl0 += g0;
            dbl1 += g1;
            const float di = L.ldinv[row];
            L.buf[row * HS + r] = di * g0;
            L.buf[row * HS + 32 + r] = di * g1;
          }
        }
      }
    } else {
    if (POOLG) {
#pragma unroll
      for (int it = 0; it < TM / 4; ++it) dy[it] = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int g = ti.g0; g < ti.g1; ++g) {
        const int gb = L.lgp[g - ti.g0], ge = L.lgp[g - ti.g0 + 1];
        if (ge <= gb) continue;
        const float4 gmx = *reinterpret_cast<const float4*>(emb + (size_t)g * 2 * DD + 4 * q);
        const float4 dmx = *reinterpret_cast<const float4*>(demb + (size_t)g * 2 * DD + 4 * q);
        float4 dmean = *reinterpret_cast<const float4*>(demb + (size_t)g * 2 * DD + DD + 4 * q);
        const float cntf = (float)(ge - gb);
        dmean = make_float4(dmean.x / cntf, dmean.y / cntf, dmean.z / cntf, dmean.w / cntf);
        float4 ties = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int it = 0; it < TM / 4; ++it) {
          const int i = it * 4 + r4;
          if (i >= gb && i < ge) {
            const float4 a = sa.v4[it];
            ties.x += (a.x == gmx.x); ties.y += (a.y == gmx.y); ties.z += (a.z == gmx.z); ties.w += (a.w == gmx.w);
          }
        }
        ties = f4_add(ties, f4_shfl_xor(ties, 16));
        ties = f4_add(ties, f4_shfl_xor(ties, 32));
        const float4 share = make_float4(dmx.x / fmaxf(ties.x, 1.f), dmx.y / fmaxf(ties.y, 1.f), dmx.z / fmaxf(ties.z, 1.f),
                                         dmx.w / fmaxf(ties.w, 1.f));
#pragma unroll
        for (int it = 0; it < TM / 4; ++it) {
          const int i = it * 4 + r4;
          if (i >= gb && i < ge) {
            const float4 a = sa.v4[it];
            dy[it] = make_float4(dmean.x + (a.x == gmx.x ? share.x : 0.f), dmean.y + (a.y == gmx.y ? share.y : 0.f),
                                 dmean.z + (a.z == gmx.z ? share.z : 0.f), dmean.w + (a.w == gmx.w ? share.w : 0.f));
          }
        }
      }
    } else {
#pragma unroll
      for (int it = 0; it < TM / 4; ++it) dy[it] = sd.v4[it];
    }
#pragma unroll
    for (int it = 0; it < TM / 4; ++it) {
      const int i = it * 4 + r4;
      float4 d = make_float4(0.f, 0.f, 0.f, 0.f);
      if (i < ti.n) {
        d = dy[it];
        if (POOLG || use_out) {
          const float4 a = sa.v4[it];            // (no activation: slope_eff = 1 -> factor 1 either way, no branch)
          d.x *= hcg_leaky_grad(a.x, slope_eff); d.y *= hcg_leaky_grad(a.y, slope_eff);
          d.z *= hcg_leaky_grad(a.z, slope_eff); d.w *= hcg_leaky_grad(a.w, slope_eff);
        }
        dbacc = f4_add(dbacc, d);
        const float di = L.ldinv[i];
        d = make_float4(di * d.x, di * d.y, di * d.z, di * d.w);
      }
      *reinterpret_cast<float4*>(L.buf + i * HS + 4 * q) = d;
    }
    }

    BSTAMP(2 + 8 * bstamp_it);
    // next tile of this wave: scalars one tile further ahead (the row loads wait until the accumulators leave room:
    // measured with the loads here, the dx / pooled variants spill 24-89 VGPRs and run 20-30 % SLOWER)
    const int tn = seq.at(round_it + 1, t, num_tiles);
    const bool have_next = tn < num_tiles;
    const TileRaw raw_cur = raw_next;
    raw_next = tile_raw(seq.at(round_it + 2, tn, num_tiles), num_tiles, gpt, B, graph_ptr, edge_ptr);
    TileInfo tin;
    Stager<DD, true> san{}, sdn;
    TileEdges ten;
    // x rows of THIS tile: in the variant with registers to spare (no dx, no pooled prologue) they are requested here,
    // one aggregation ahead of their use, instead of waiting out an HBM round trip in front of the dW MFMAs
#ifndef HCG_EARLY_X
#define HCG_EARLY_X 1
#endif
    constexpr bool EARLY_X = HCG_EARLY_X && !NEEDS_DX && !POOLG;
    Stager<KPAD, VEC> sx;
    if (EARLY_X) sx.load(x, F, N, ti.nbase, ti.n, lane);

    // ---- 2. dH = dinv (.) ( (C + I)^T dY' ):  A[m = j][k = i] = cnt[i][j] (exact),  B[k = i][col] = dY'[i][col] (split)
    AdjFrags adj;
    adj.read<true>(L.cnt, r, h, lane, status);
    f32x16 dh0, dh1;
#pragma unroll
    for (int i = 0; i < 16; ++i) { dh0[i] = 0.f; dh1[i] = 0.f; }
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      float g0[8], g1[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        g0[j] = L.buf[(16 * s + 8 * h + j) * HS + r];
        g1[j] = L.buf[(16 * s + 8 * h + j) * HS + 32 + r];
      }
      mfma_exact_a(dh0, adj.f[s], split3(g0));
      mfma_exact_a(dh1, adj.f[s], split3(g1));
    }
    mfma_results_fence(dh0, dh1);
    {
      float dvr[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) dvr[i] = L.ldinv[krow(i, h)];
#pragma unroll
      for (int i = 0; i < 16; ++i) { dh0[i] *= dvr[i]; dh1[i] *= dvr[i]; }
    }

    BSTAMP(3 + 8 * bstamp_it);
    // ---- 3. x tile -> buf ; dW += dH^T x.  A operand = the dH accumulators (slot j of k-step s <-> node krow(8s + j, h)),
    //         B[k = node][f] read down the columns of the x tile
    if (!EARLY_X) sx.load(x, F, N, ti.nbase, ti.n, lane);
    sx.write(L.buf, F, ti.n, lane);
    BSTAMP(4 + 8 * bstamp_it);
    // premask, wide rows: bit 4 it + c <-> x[row it*4 + r4][4 q + c] > 0 -- one register carried to the dx stores
    // instead of the rows (measured against bits taken in the accumulator layout inside the dW loop: those spill)
    uint32_t xpos = 0;
    if (NEEDS_DX && VEC) {
      if (premask) {
#pragma unroll
        for (int it = 0; it < TM / 4; ++it) {
          const float4 v = sx.v4[it];
          xpos |= (uint32_t)(v.x > 0.f) << (4 * it) | (uint32_t)(v.y > 0.f) << (4 * it + 1) |
                  (uint32_t)(v.z > 0.f) << (4 * it + 2) | (uint32_t)(v.w > 0.f) << (4 * it + 3);
        }
      }
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const float a0[8] = {dh0[8 * s], dh0[8 * s + 1], dh0[8 * s + 2], dh0[8 * s + 3],
                           dh0[8 * s + 4], dh0[8 * s + 5], dh0[8 * s + 6], dh0[8 * s + 7]};
      const float a1[8] = {dh1[8 * s], dh1[8 * s + 1], dh1[8 * s + 2], dh1[8 * s + 3],
                           dh1[8 * s + 4], dh1[8 * s + 5], dh1[8 * s + 6], dh1[8 * s + 7]};
      const Split3 A0 = split3(a0), A1 = split3(a1);
#pragma unroll
      for (int nb = 0; nb < NBF; ++nb) {
        float xb[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) xb[j] = L.buf[krow(8 * s + j, h) * HS + nb * 32 + r];
        const Split3 Bx = split3(xb);
        mfma_split(dw[0][nb], A0, Bx.p1, Bx.p2, Bx.p3);
        mfma_split(dw[1][nb], A1, Bx.p1, Bx.p2, Bx.p3);
      }
    }

    BSTAMP(5 + 8 * bstamp_it);
    // ---- 4. dx = dH W: dH -> buf as [node][d] (the contraction runs over the accumulator's LANE
    //         dimension, so this one needs the LDS transpose), B[k = d][j = f] = W[d][f] from the pre-split image
    if (NEEDS_DX) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int node = krow(i, h);
        L.buf[node * HS + r] = dh0[i];
        L.buf[node * HS + 32 + r] = dh1[i];
      }
      f32x16 dxa[NBF];
#pragma unroll
      for (int nb = 0; nb < NBF; ++nb)
#pragma unroll
        for (int i = 0; i < 16; ++i) dxa[nb][i] = 0.f;
      constexpr int ld = DD + WPAD, plane = KPAD * ld;
#pragma unroll
      for (int s = 0; s < DD / 16; ++s) {
        const float4 a0 = *reinterpret_cast<const float4*>(L.buf + r * HS + 16 * s + 8 * h);
        const float4 a1 = *reinterpret_cast<const float4*>(L.buf + r * HS + 16 * s + 8 * h + 4);
        const float xa[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
        const Split3 A = split3(xa);
#pragma unroll
        for (int nb = 0; nb < NBF; ++nb) {
          const short* w0 = wtl + (nb * 32 + r) * ld + 16 * s + 8 * h;
          mfma_split(dxa[nb], A, *reinterpret_cast<const bf16x8*>(w0), *reinterpret_cast<const bf16x8*>(w0 + plane),
                     *reinterpret_cast<const bf16x8*>(w0 + 2 * plane));
        }
      }
#pragma unroll
      for (int nb = 0; nb < NBF; ++nb) mfma_results_fence(dxa[nb]);
      BSTAMP(6 + 8 * bstamp_it);
      if (VEC) {   // F == KPAD: rows are whole float4 groups -> transpose through LDS, dwordx4 stores
#pragma unroll
        for (int i = 0; i < 16; ++i)
#pragma unroll
          for (int nb = 0; nb < NBF; ++nb) L.buf[krow(i, h) * HS + nb * 32 + r] = dxa[nb][i];
        if (premask) {    // own rows only (the mask bits are this lane's rows), rows >= n not stored
          if (4 * q < KPAD) {
            float4 ov[TM / 4];
#pragma unroll
            for (int it = 0; it < TM / 4; ++it) {
              ov[it] = *reinterpret_cast<const float4*>(L.buf + (it * 4 + r4) * HS + 4 * q);
              ov[it].x *= (xpos >> (4 * it) & 1) ? 1.f : slope;
              ov[it].y *= (xpos >> (4 * it + 1) & 1) ? 1.f : slope;
              ov[it].z *= (xpos >> (4 * it + 2) & 1) ? 1.f : slope;
              ov[it].w *= (xpos >> (4 * it + 3) & 1) ? 1.f : slope;
            }
#pragma unroll
            for (int it = 0; it < TM / 4; ++it)
              if (it * 4 + r4 < ti.n)
                *reinterpret_cast<float4*>(dx + (size_t)(ti.nbase + it * 4 + r4) * F + 4 * q) = ov[it];
          }
        } else if (ti.n > 0) {   // rows >= n redirected to row n-1 (duplicate identical stores, no per-lane branch)
          const int qc = 4 * q < KPAD ? q : 0;
          float4 ov[TM / 4];
#pragma unroll
          for (int it = 0; it < TM / 4; ++it) {
            const int row = it * 4 + r4 < ti.n ? it * 4 + r4 : ti.n - 1;
            ov[it] = *reinterpret_cast<const float4*>(L.buf + row * HS + 4 * qc);
          }
#pragma unroll
          for (int it = 0; it < TM / 4; ++it) {
            const int row = it * 4 + r4 < ti.n ? it * 4 + r4 : ti.n - 1;
            *reinterpret_cast<float4*>(dx + (size_t)(ti.nbase + row) * F + 4 * qc) = ov[it];
          }
        }
      } else {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int row = krow(i, h);
          if (row < ti.n) {
#pragma unroll
            for (int nb = 0; nb < NBF; ++nb) {
              const int f = nb * 32 + r;
              if (f < F) {
                const size_t at = (size_t)(ti.nbase + row) * F + f;
                dx[at] = premask ? dxa[nb][i] * hcg_leaky_grad(x[at], slope) : dxa[nb][i];
              }
            }
          }
        }
      }
    }

    BSTAMP(7 + 8 * bstamp_it);
    have = have_next;
    ++round_it;
    if (have_next) {
      tin = tile_finish(raw_cur, gpt, lane, status);
      if (BITS) {
        pos = poolbits[(size_t)tn * 128 + lane];
        ismax = poolbits[(size_t)tn * 128 + 64 + lane];
      } else if (POOLG || use_out) {
        san.load(a_out, DD, N, tin.nbase, tin.n, lane);
      }
      if (!POOLG) sdn.load(dout, DD, N, tin.nbase, tin.n, lane);
      ten.load(tin, graph_ptr, ei, E, lane);
      t = tn;
      ti = tin;
      sa = san;
      if (!POOLG) sd = sdn;
      te = ten;
    }
    BSTAMP(8 + 8 * bstamp_it);
    if (bstamp_it < 5) ++bstamp_it;
  }
  BSTAMP(60);

  // ---- combine the waves of this workgroup (fixed order, two rounds of four waves through a flat
  //      view of the LDS) and publish one partial slab
  constexpr int SLABF = DD * KPAD + DD;
  constexpr int PER_T = (SLABF + WAVES * 64 - 1) / (WAVES * 64);
  static_assert(sizeof(WaveLdsB) * WAVES >= 4 * SLABF * sizeof(float), "wave-combine scratch must fit in the tile buffers");
  float* flat = reinterpret_cast<float*>(&lds[0]);
#pragma unroll
  for (int mb = 0; mb < 2; ++mb)
#pragma unroll
    for (int nb = 0; nb < NBF; ++nb) mfma_results_fence(dw[mb][nb]);
  dbacc = f4_add(dbacc, f4_shfl_xor(dbacc, 16));
  dbacc = f4_add(dbacc, f4_shfl_xor(dbacc, 32));
  dbl0 += __shfl_xor(dbl0, 32, 64);
  dbl1 += __shfl_xor(dbl1, 32, 64);
  float tot[PER_T];
#pragma unroll
  for (int j = 0; j < PER_T; ++j) tot[j] = 0.f;
#pragma unroll
  for (int round = 0; round < 2; ++round) {
    __syncthreads();   // tile buffers (round 0) / the previous round's regions (round 1) are free
    if ((wave >> 2) == round) {
      float* mine = flat + (size_t)(wave & 3) * SLABF;
#pragma unroll
      for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int nb = 0; nb < NBF; ++nb)
#pragma unroll
          for (int i = 0; i < 16; ++i) mine[(mb * 32 + krow(i, h)) * KPAD + nb * 32 + r] = dw[mb][nb][i];
      if (BITS) {
        if (h == 0) { mine[DD * KPAD + r] = dbl0; mine[DD * KPAD + 32 + r] = dbl1; }
      } else if (r4 == 0) {
        *reinterpret_cast<float4*>(mine + DD * KPAD + 4 * q) = dbacc;
      }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < PER_T; ++j) {
      const int idx = threadIdx.x + j * WAVES * 64;
      if (idx < SLABF) tot[j] += ((flat[idx] + flat[SLABF + idx]) + flat[2 * SLABF + idx]) + flat[3 * SLABF + idx];
    }
  }
  float* slab = partials + (size_t)blockIdx.x * SLABF;
#pragma unroll
  for (int j = 0; j < PER_T; ++j) {
    const int idx = threadIdx.x + j * WAVES * 64;
    if (idx < SLABF) slab[idx] = tot[j];
  }
  BSTAMP(63);
  BSTAMP_FLUSH();
}


int pick_grid(int num_tiles) {
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) == hipSuccess) {
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
  }
  int grid = (num_tiles + WAVES - 1) / WAVES;
  if (grid > cus) grid = cus;
  return grid < 1 ? 1 : grid;
}

}  // namespace

// graphs per 32-row tile for this (F, D, max_nodes), or 0 when the fused kernels do not apply
extern "C" int hcg_fused_graphs_per_tile(int64_t F, int64_t D, int64_t max_nodes_per_graph) {
  if (D != DD || F < 1 || F > 64 || max_nodes_per_graph < 1 || max_nodes_per_graph > TM) return 0;
  return (int)(TM / max_nodes_per_graph);
}

extern "C" size_t hcg_fused_workspace_bytes(int64_t B, int64_t F, int64_t D, int graphs_per_tile) {
  if (graphs_per_tile <= 0) return 0;
  const int kpad = F <= 32 ? 32 : 64;
  const int tiles = (int)((B + graphs_per_tile - 1) / graphs_per_tile);
  return (size_t)pick_grid(tiles) * (DD * kpad + DD) * sizeof(float) + 256;
}

// workspace of the head in the forward's tail: one slab (gradient partial sums + SSE partial) per workgroup of the launch
static size_t fused_head_workspace_bytes(int64_t B, int graphs_per_tile) {
  if (graphs_per_tile <= 0 || B <= 0) return 0;
  const int tiles = (int)((B + graphs_per_tile - 1) / graphs_per_tile);
  return hcg_align_up((size_t)pick_grid(tiles) * (hcg_head::HC<DD>::SLAB + 1) * sizeof(float), 256) + 256;
}

extern "C" int hcg_fused_head_reduce_job(const void* workspace, size_t workspace_bytes, int64_t B, int graphs_per_tile, int64_t C,
                                         float* dW0, float* db0, float* dW1, float* db1, hcg_reduce_job* job) {
  if (B <= 0 || graphs_per_tile < 1 || C < 1 || C > hcg_head::RCMAX || !job || !workspace) return HCG_ERR_INVALID_ARG;
  if (dW0 && (!db0 || !dW1 || !db1)) return HCG_ERR_INVALID_ARG;
  if (workspace_bytes < fused_head_workspace_bytes(B, graphs_per_tile)) return HCG_ERR_WORKSPACE;
  const int tiles = (int)((B + graphs_per_tile - 1) / graphs_per_tile);
  hcg_head::head_fill_job<DD>((const float*)workspace, pick_grid(tiles), (int)C, dW0, db0, dW1, db1, job);
  return HCG_OK;
}

// Every forward form of the small-graph tiles: one or two stacked conv layers, optional [max, mean] pooling of the last one,
// optional training form (the pooled layer's activations stay on chip, 2 bits per element leave), optional readout head in
// the tail of the launch.
extern "C" int hcg_fused_forward(const hcg_fused_fwd_args* a, hcg_stream_t stream_) {
  if (!a) return HCG_ERR_INVALID_ARG;
  hipStream_t stream = (hipStream_t)stream_;
  const float *x = a->x, *W = a->W1, *b = a->b1, *W2 = a->W2, *b2 = a->b2;
  const int64_t* edge_index = a->edge_index;
  int64_t E = a->E;
  const int64_t N = a->N, B = a->B, F = a->F, D = a->D;
  const int graphs_per_tile = a->graphs_per_tile;
  float *out = a->out1, *out2 = a->out2, *emb = a->emb;
  uint32_t* poolbits = a->poolbits;
  const bool stack2 = W2 != nullptr;
  const bool bits = poolbits != nullptr;   // the pooled layer's activations stay on chip (training form)
  const bool head = a->head_W0 != nullptr;
  if (D != DD || F < 1 || F > 64 || graphs_per_tile < 1) return HCG_ERR_UNSUPPORTED;
  if (a->apply_act && !(a->slope >= 0.f && a->slope <= 1.f)) return HCG_ERR_UNSUPPORTED;  // LeakyReLU is evaluated as max(v, slope*v)
  if (N < 0 || B < 0 || E < 0) return HCG_ERR_INVALID_ARG;
  if (B == 0 || N == 0) return head ? HCG_ERR_INVALID_ARG : HCG_OK;
  if (!x || !W || !b || !a->graph_ptr || !a->edge_ptr || !a->status || (E > 0 && !edge_index)) return HCG_ERR_INVALID_ARG;
  if (bits && !emb) return HCG_ERR_INVALID_ARG;
  if (stack2) {
    if (!b2 || !out || (!out2 && !bits)) return HCG_ERR_INVALID_ARG;
  } else {
    if (out2 || (!out && !bits)) return HCG_ERR_INVALID_ARG;
  }
  const bool head_bwd = head && !(a->head_flags & HCG_HEAD_FORWARD_ONLY);
  FwdHead H{};
  if (head) {
    // the head needs the training form of a stacked pair (the one form the training step of a two-layer model issues)
    if (!stack2 || !bits || (a->head_flags & ~HCG_HEAD_FORWARD_ONLY)) return HCG_ERR_UNSUPPORTED;
    if (a->C < 1 || a->C > hcg_head::RCMAX) return HCG_ERR_UNSUPPORTED;
    if (!a->y || !a->head_b0 || !a->head_W1 || !a->head_b1 || !a->z || !a->out || (head_bwd && !a->demb) || !a->head_workspace)
      return HCG_ERR_INVALID_ARG;
    if (a->head_workspace_bytes < fused_head_workspace_bytes(B, graphs_per_tile)) return HCG_ERR_WORKSPACE;
    H = FwdHead{a->y, a->head_W0, a->head_b0, a->head_W1, a->head_b1, (int)a->C, a->z, a->out, a->demb, (float*)a->head_workspace,
                (int*)a->step_counter};
  }
  const int tiles = (int)((B + graphs_per_tile - 1) / graphs_per_tile);
  const int grid = pick_grid(tiles);
  const bool vec = (F == 64 || F == 32) && ((uintptr_t)x % 16 == 0);
  const dim3 g(grid), blk(WAVES * 64);
  if (E == 0) { edge_index = reinterpret_cast<const int64_t*>(a->graph_ptr); E = 1; }  // readable dummy; no tile has edges
  // a partial last round is dealt evenly over the workgroups (TileSeq); whole rounds keep the plain form
  const bool deal = tiles % (grid * WAVES) != 0 && tiles > grid * WAVES;
#define LAUNCH_FWD_D(KP, VC, PL, ST, BT, HD, DL)                                                                           \
  hipLaunchKernelGGL((k_fused_layer_fwd<KP, VC, PL, ST, BT, HD, DL>), g, blk, 0, stream, x, (int)F, W, b, W2, b2, edge_index, \
                     E, a->graph_ptr, a->edge_ptr, N, graphs_per_tile, (int)B, tiles, a->slope, a->apply_act, out, out2, \
                     emb, poolbits, a->status, H)
#define LAUNCH_FWD(KP, VC, PL, ST, BT, HD)                                                                               \
  do { if (deal) LAUNCH_FWD_D(KP, VC, PL, ST, BT, HD, true); else LAUNCH_FWD_D(KP, VC, PL, ST, BT, HD, false); } while (0)
#define DISPATCH_FWD(KP, VC)                                                                                     \
  do {                                                                                                           \
    if (head)        { if (head_bwd) LAUNCH_FWD(KP, VC, true, true, true, 2); else LAUNCH_FWD(KP, VC, true, true, true, 1); } \
    else if (bits)   { if (stack2) LAUNCH_FWD(KP, VC, true, true, true, 0); else LAUNCH_FWD(KP, VC, true, false, true, 0); } \
    else if (stack2) { if (emb) LAUNCH_FWD(KP, VC, true, true, false, 0); else LAUNCH_FWD(KP, VC, false, true, false, 0); }  \
    else             { if (emb) LAUNCH_FWD(KP, VC, true, false, false, 0); else LAUNCH_FWD(KP, VC, false, false, false, 0); } \
  } while (0)
  if (F <= 32) { if (vec) DISPATCH_FWD(32, true); else DISPATCH_FWD(32, false); }
  else         { if (vec) DISPATCH_FWD(64, true); else DISPATCH_FWD(64, false); }
#undef DISPATCH_FWD
#undef LAUNCH_FWD
#undef LAUNCH_FWD_D
  HCG_CHECK_LAUNCH();
  return HCG_OK;
}

// kind HCG_FUSED_POOLBITS: 2 bits per element of the pooled layer (training forms, see k_fused_layer_fwd<BITS>);
// kind HCG_FUSED_HEAD_WS: the head-in-the-forward's slabs + SSE partials
extern "C" size_t hcg_fused_aux_bytes(int kind, int64_t B, int graphs_per_tile) {
  if (graphs_per_tile <= 0 || B <= 0) return 0;
  if (kind == HCG_FUSED_HEAD_WS) return fused_head_workspace_bytes(B, graphs_per_tile);
  if (kind != HCG_FUSED_POOLBITS) return 0;
  return (size_t)((B + graphs_per_tile - 1) / graphs_per_tile) * 128 * sizeof(uint32_t);
}

static int launch_fused_bwd(const float* dout, const float* demb, const float* emb, const float* out,
                            const uint32_t* poolbits, const float* x, const float* W, const int64_t* edge_index, int64_t E,
                            const int32_t* graph_ptr, const int32_t* edge_ptr, int64_t N, int64_t B, int64_t F,
                            int64_t D, int graphs_per_tile, float slope, int apply_act, float* dx, int32_t* status,
                            void* workspace, size_t workspace_bytes, hipStream_t stream) {
  if (D != DD || F < 1 || F > 64 || graphs_per_tile < 1) return HCG_ERR_UNSUPPORTED;
  if (N < 0 || B < 0 || E < 0 || !W || !workspace) return HCG_ERR_INVALID_ARG;
  const bool poolg = (dout == nullptr);
  const bool bits = poolbits != nullptr;
  if (bits) {   // stands in for `out` and `emb`
    if (!poolg || !demb) return HCG_ERR_INVALID_ARG;
    out = x;    // (never read; keeps the pointer checks below uniform)
    emb = demb;
  }
  if (poolg && (!demb || !emb)) return HCG_ERR_INVALID_ARG;
  if (apply_act & ~3) return HCG_ERR_INVALID_ARG;
  if ((apply_act & 2) && !dx) return HCG_ERR_INVALID_ARG;
  // `out` is only read for the activation derivative and the pooled-gradient routing
  if (N > 0 && B > 0 && (((poolg || (apply_act & 1)) && !out) || !x || !graph_ptr || !edge_ptr || !status || (E > 0 && !edge_index)))
    return HCG_ERR_INVALID_ARG;
  if (!poolg && !(apply_act & 1)) out = nullptr;
  const int kpad = F <= 32 ? 32 : 64;
  const int tiles = (int)((B + graphs_per_tile - 1) / graphs_per_tile);
  const int grid = (N > 0 && B > 0) ? pick_grid(tiles) : 0;
  const size_t slab = (size_t)(DD * kpad + DD);
  if (workspace_bytes < (size_t)grid * slab * sizeof(float)) return HCG_ERR_WORKSPACE;
  float* partials = (float*)workspace;
  if (grid > 0) {
    const bool vec = (F == 64 || F == 32) && ((uintptr_t)x % 16 == 0);
    const bool ndx = dx != nullptr;
    const dim3 g(grid), blk(WAVES * 64);
    if (E == 0) { edge_index = reinterpret_cast<const int64_t*>(graph_ptr); E = 1; }  // readable dummy
    const bool deal = tiles % (grid * WAVES) != 0 && tiles > grid * WAVES;     // (as the forward: TileSeq)
#define LAUNCH_BWD_D(KP, VC, DX, PG, BT, DL)                                                                          \
  hipLaunchKernelGGL((k_fused_layer_bwd<KP, VC, DX, PG, BT, DL>), g, blk, 0, stream, dout, demb, emb, out, poolbits, x, \
                     (int)F, W, edge_index, E, graph_ptr, edge_ptr, N, graphs_per_tile, (int)B, tiles, slope, apply_act, \
                     dx, partials, status)
#define LAUNCH_BWD(KP, VC, DX, PG, BT)                                                                                \
  do { if (deal) LAUNCH_BWD_D(KP, VC, DX, PG, BT, true); else LAUNCH_BWD_D(KP, VC, DX, PG, BT, false); } while (0)
#define DISPATCH_BWD(KP, VC)                                                                                   \
  do {                                                                                                         \
    if (bits)     { if (ndx) LAUNCH_BWD(KP, VC, true, true, true); else LAUNCH_BWD(KP, VC, false, true, true); }   \
    else if (ndx) { if (poolg) LAUNCH_BWD(KP, VC, true, true, false); else LAUNCH_BWD(KP, VC, true, false, false); } \
    else          { if (poolg) LAUNCH_BWD(KP, VC, false, true, false); else LAUNCH_BWD(KP, VC, false, false, false); } \
  } while (0)
    if (kpad == 32) { if (vec) DISPATCH_BWD(32, true); else DISPATCH_BWD(32, false); }
    else            { if (vec) DISPATCH_BWD(64, true); else DISPATCH_BWD(64, false); }
#undef DISPATCH_BWD
#undef LAUNCH_BWD
#undef LAUNCH_BWD_D
    HCG_CHECK_LAUNCH();
  }
  return HCG_OK;
}

// `poolbits` != NULL: pooled backward of a layer whose forward ran in the training form (poolbits stand in for out / emb;
// dout, emb, out must then be NULL)
extern "C" int hcg_fused_layer_bwd(const float* dout, const float* demb, const float* emb, const float* out,
                                   const uint32_t* poolbits, const float* x, const float* W, const int64_t* edge_index, int64_t E,
                                   const int32_t* graph_ptr, const int32_t* edge_ptr, int64_t N, int64_t B, int64_t F,
                                   int64_t D, int graphs_per_tile, float slope, int apply_act, float* dx, int32_t* status,
                                   void* workspace, size_t workspace_bytes, hcg_stream_t stream) {
  if (poolbits && (dout || emb || out || !demb)) return HCG_ERR_INVALID_ARG;
  return launch_fused_bwd(dout, demb, emb, out, poolbits, x, W, edge_index, E, graph_ptr, edge_ptr, N, B, F, D,
                          graphs_per_tile, slope, apply_act, dx, status, workspace, workspace_bytes, (hipStream_t)stream);
}

// host-side description of this layer's slab set for hcg_step_tail (no launch)
extern "C" int hcg_fused_reduce_job(const void* workspace, size_t workspace_bytes, int64_t N, int64_t B, int64_t F, int64_t D,
                                    int graphs_per_tile, float* dW, float* db, hcg_reduce_job* job) {
  if (D != DD || F < 1 || F > 64 || graphs_per_tile < 1 || !dW || !db || !job || !workspace) return HCG_ERR_INVALID_ARG;
  const int kpad = F <= 32 ? 32 : 64;
  const int tiles = (int)((B + graphs_per_tile - 1) / graphs_per_tile);
  const int grid = (N > 0 && B > 0) ? pick_grid(tiles) : 0;
  const int slab_floats = DD * kpad + DD;
  if (workspace_bytes < (size_t)grid * slab_floats * sizeof(float)) return HCG_ERR_WORKSPACE;
  job->slabs = (const float*)workspace;
  job->nslabs = grid;
  job->slab_floats = slab_floats;
  job->nseg = 2;
  job->sse_part = nullptr;
  job->reserved = 0;
  job->seg[0] = hcg_reduce_seg{0, DD * kpad, kpad, (int32_t)F, dW};
  job->seg[1] = hcg_reduce_seg{DD * kpad, DD, 1, 1, db};
  return HCG_OK;
}
