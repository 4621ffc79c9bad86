// Fused per-layer kernels for batches of SMALL graphs (the BASELINE workload: ~30-atom graphs, 64-d).
//
// One GCN layer (reference: PyG GCNConv + nn.LeakyReLU, call sites model/gcn.py:58-63; SURVEY rows
// a4-a8, and a9 for the last layer) is ONE kernel: a tile of whole graphs (<= 32 node rows) is
// brought on chip once, transformed on the f32 matrix cores, aggregated out of LDS with a fixed-order
// segmented sum (no atomics), biased/activated, (pooled) and written once -- the "layer-fused
// minimum" HBM traffic of SURVEY 8(d): read the layer input once, write its output once.
//
// CDNA4 mapping
//   * wavefront-autonomous tiles: each 64-lane wave owns a stream of tiles and a private LDS region
//     (two [32][D+4] fp32 buffers); no workgroup barrier in the steady state.  A 512-thread
//     workgroup (8 waves = 2 per SIMD) per CU keeps the matrix pipe of every SIMD fed by one wave
//     while its partner stages / aggregates.
//   * v_mfma_f32_32x32x2_f32 (exact f32): the weight operand lives in 64 VGPRs for the whole
//     kernel; the activation operand is read from LDS with ds_read_b128 -- legal because the MFMA
//     sums over k, so the (k-step, lane-half) -> k assignment is free: lane (r, h) takes
//     k = 8t + 4h + u for u = 0..3 of its 16-byte read.  Row stride D+4 floats makes those reads
//     conflict-free (16 lanes x 16 B = 64 banks).
//   * backward: dW accumulates in MFMA accumulators ACROSS all tiles of a wave (K = node rows);
//     waves combine through LDS, workgroups through a [grid][D*KPAD+D] slab reduced in a fixed order
//     by a second kernel -> bitwise reproducible gradients.
#include "common.h"

// Diagnostic builds only (tools/probe_fused.hip defines HCG_STAMP): s_memtime stamps of a few waves go to
// a buffer of their own; the product build compiles STAMP() to nothing and executes no stamp.
#ifdef HCG_STAMP
__device__ unsigned long long* g_stamp_buf = nullptr;
#define STAMP(idx)                                                                                          \
  do {                                                                                                      \
    __builtin_amdgcn_sched_barrier(0);                                                                      \
    unsigned long long _t;                                                                                  \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(_t)::"memory");                              \
    __builtin_amdgcn_sched_barrier(0);                                                                      \
    if (g_stamp_buf && (threadIdx.x & 63) == 0 && blockIdx.x < 4)                                           \
      g_stamp_buf[((size_t)blockIdx.x * WAVES + (threadIdx.x >> 6)) * 64 + (idx)] = _t;                     \
  } while (0)
#else
#define STAMP(idx) do { } while (0)
#endif
// tools/probe_fused.hip -DHCG_ABLATE=<bits> (timing-only diagnostic builds; results are wrong on purpose):
//   1 = no MFMA in the forward GEMM, 2 = no neighbour gather, 4 = no output stores, 8 = no x loads
#ifndef HCG_ABLATE
#define HCG_ABLATE 0
#endif

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int TM = 32;          // node rows per wave tile
constexpr int DD = 64;          // layer width handled by this file (embedding_dim = 64)
constexpr int HS = DD + 4;      // LDS row stride (floats) of a [TM][DD] buffer
constexpr int WAVES = 8;        // waves per workgroup
constexpr int LCOL_CAP = 128;   // edges of one tile whose neighbour row offsets are cached in LDS
constexpr int BUF_FLOATS = TM * HS;
constexpr unsigned ZERO_ROW_OFF = TM * HS * 4;  // byte offset of the all-zero row that follows a gather buffer

// bufA is the buffer the segmented sum gathers from (forward: h', backward: dY'); it that is gathered from is followed by one
// all-zero row so that a missing neighbour slot is an unconditional add of zeros (no select per value).
// `lofs` holds, per edge of the tile, the BYTE offset of the neighbour's row inside that buffer.
struct WaveLds {
  float bufA[BUF_FLOATS];
  float zeroA[HS];
  float bufB[BUF_FLOATS];
  int lrow[TM + 4];
  float ldinv[TM];
  int lgp[TM + 4];  // node offset of every graph of the tile (a tile holds <= TM graphs)
  unsigned short lofs[LCOL_CAP];
};

__device__ __forceinline__ float4 f4_add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 f4_max(float4 a, float4 b) {
  return make_float4(fmaxf(a.x, b.x), fmaxf(a.y, b.y), fmaxf(a.z, b.z), fmaxf(a.w, b.w));
}
__device__ __forceinline__ float4 f4_shfl_xor(float4 v, int m) {
  return make_float4(__shfl_xor(v.x, m, 64), __shfl_xor(v.y, m, 64), __shfl_xor(v.z, m, 64), __shfl_xor(v.w, m, 64));
}

// tile bookkeeping shared by forward and backward.  `rp`/`ci` are the CSR (forward) or its transpose
// (backward); `edge_ptr[g]` == rp[graph_ptr[g]] because a blocked plan keeps every graph's edges
// contiguous, so all four tile scalars come from ONE round of (scalar) loads.
// Rule for every global load in this file: never guard a load with a per-lane branch (hipcc then
// serialises it behind its own s_waitcnt vmcnt(0)); clamp the index into range and select afterwards.
struct TileInfo {
  int g0, g1, nbase, n, ebase, ne;
  bool cols_in_lds;
};

__device__ __forceinline__ TileInfo tile_scalars(int t, int gpt, int B, const int32_t* __restrict__ graph_ptr,
                                                 const int32_t* __restrict__ edge_ptr, int lane, int32_t* status) {
  TileInfo ti;
  t = __builtin_amdgcn_readfirstlane(t);   // wave-uniform by construction: lets the four loads be scalar (s_load)
  ti.g0 = t * gpt;
  ti.g1 = ti.g0 + gpt < B ? ti.g0 + gpt : B;
  ti.nbase = graph_ptr[ti.g0];
  ti.n = graph_ptr[ti.g1] - ti.nbase;
  ti.ebase = edge_ptr[ti.g0];
  ti.ne = edge_ptr[ti.g1] - ti.ebase;
  if (ti.n > TM || ti.n < 0 || ti.ne < 0 || gpt > TM) {  // host metadata was wrong: refuse the tile
    if (lane == 0) atomicOr(status, HCG_STATUS_SHAPE_LIMIT);
    ti.n = 0;
    ti.ne = 0;
    ti.g1 = ti.g0;
  }
  ti.cols_in_lds = ti.ne <= LCOL_CAP;
  return ti;
}

// per-lane index data of a tile: loaded into registers by load(), published to the wave's LDS by
// write() -- split so the NEXT tile's index can be in flight while the current tile computes.
struct TileIndex {
  int my_row, my_gp;
  float my_dinv;
  int my_col[LCOL_CAP / 64];

  __device__ __forceinline__ void load(const TileInfo& ti, const int32_t* __restrict__ graph_ptr,
                                       const int32_t* __restrict__ rp, const int32_t* __restrict__ ci,
                                       const float* __restrict__ dinv, int64_t N, int lane) {
    const int lr = lane <= ti.n ? lane : ti.n;               // rp has N + 1 entries: nbase + n is valid
    my_row = rp[ti.nbase + lr] - ti.ebase;
    int64_t dn = (int64_t)ti.nbase + (lane < TM ? lane : 0);
    if (dn > N - 1) dn = N - 1;
    my_dinv = dinv[dn];
    const int ng = ti.g1 - ti.g0;
    my_gp = graph_ptr[ti.g0 + (lane <= ng ? lane : ng)] - ti.nbase;
    const bool want = ti.cols_in_lds && ti.ne > 0;          // wave-uniform
#pragma unroll
    for (int j = 0; j < LCOL_CAP / 64; ++j) {
      const int k = lane + 64 * j;
      my_col[j] = want ? ci[ti.ebase + (k < ti.ne ? k : ti.ne - 1)] - ti.nbase : 0;
    }
  }

  __device__ __forceinline__ void write(WaveLds& L, const TileInfo& ti, int lane, int32_t* status) const {
    const int ng = ti.g1 - ti.g0;
    if (ti.cols_in_lds) {
#pragma unroll
      for (int j = 0; j < LCOL_CAP / 64; ++j) {
        const int k = lane + 64 * j;
        if (k < ti.ne) {
          int c = my_col[j];
          if (c < 0 || c >= ti.n) { c = 0; atomicOr(status, HCG_STATUS_EDGE_UNGROUPED); }
          L.lofs[k] = (unsigned short)(c * HS * 4);
        }
      }
    }
    if (lane <= ti.n) L.lrow[lane] = my_row;
    if (lane < TM) L.ldinv[lane] = lane < ti.n ? my_dinv : 0.f;
    if (lane <= ng) L.lgp[lane] = my_gp;
  }
};

constexpr int GJ = 4;  // neighbours gathered with all reads in flight; longer rows continue in a loop

// res[it] = sum_{k in row i} src[c_k] + src[i]   for this lane's rows i = 4*it + r4 (it = 0..TM/4-1),
// float4 slot q.  Neighbours first in CSR order, self loop last (the order the reference's
// scatter_add_ over [edges ; self loops] adds them).  Written for ILP: the row offsets, then the
// neighbour indices, then the neighbour rows are each read as one batch of independent LDS loads --
// as a dependent per-neighbour loop this phase took 17k cycles per tile (4x the MFMA time).
__device__ __forceinline__ void gather_tile(const WaveLds& L, const float* src, const TileInfo& ti,
                                            const int32_t* __restrict__ ci, int q, int r4, int32_t* status,
                                            float4 (&res)[TM / 4]) {
  constexpr int HALF = TM / 8;  // two batches of 4 row groups: bounds the live index registers
  const char* srcb = reinterpret_cast<const char*>(src) + 16 * q;
#pragma unroll
  for (int hb = 0; hb < 2; ++hb) {
    int kb[HALF], dg[HALF];
#pragma unroll
    for (int u = 0; u < HALF; ++u) {
      const int i = (hb * HALF + u) * 4 + r4;
      const int ic = i < ti.n ? i : (ti.n > 0 ? ti.n - 1 : 0);
      kb[u] = L.lrow[ic];
      dg[u] = i < ti.n ? L.lrow[ic + 1] - kb[u] : 0;
    }
    if (ti.cols_in_lds) {
      unsigned ofs[HALF][GJ];
#pragma unroll
      for (int u = 0; u < HALF; ++u)
#pragma unroll
        for (int j = 0; j < GJ; ++j) {
          const int k = kb[u] + j;
          const unsigned o = L.lofs[k < LCOL_CAP ? k : LCOL_CAP - 1];
          ofs[u][j] = j < dg[u] ? o : ZERO_ROW_OFF;        // missing slot -> the zero row
        }
#pragma unroll
      for (int u = 0; u < HALF; ++u) {
        float4 acc = *reinterpret_cast<const float4*>(srcb + ofs[u][0]);
#pragma unroll
        for (int j = 1; j < GJ; ++j) acc = f4_add(acc, *reinterpret_cast<const float4*>(srcb + ofs[u][j]));
        res[hb * HALF + u] = acc;
      }
#pragma unroll
      for (int u = 0; u < HALF; ++u) {
        if (__ballot(dg[u] > GJ) != 0ull) {                  // rare: a node with more than GJ neighbours
          const int ke = kb[u] + dg[u];
          for (int k = kb[u] + GJ; __ballot(k < ke) != 0ull; ++k)
            if (k < ke) res[hb * HALF + u] = f4_add(res[hb * HALF + u], *reinterpret_cast<const float4*>(srcb + L.lofs[k]));
        }
      }
    } else {                                                  // tile with > LCOL_CAP edges: indices from global
#pragma unroll
      for (int u = 0; u < HALF; ++u) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        const int ke = kb[u] + dg[u];
        for (int k = kb[u]; __ballot(k < ke) != 0ull; ++k) {
          if (k < ke) {
            int c = ci[ti.ebase + k] - ti.nbase;
            if (c < 0 || c >= ti.n) { c = 0; atomicOr(status, HCG_STATUS_EDGE_UNGROUPED); }
            acc = f4_add(acc, *reinterpret_cast<const float4*>(srcb + c * HS * 4));
          }
        }
        res[hb * HALF + u] = acc;
      }
    }
  }
#pragma unroll
  for (int it = 0; it < TM / 4; ++it)
    res[it] = f4_add(res[it], *reinterpret_cast<const float4*>(srcb + (it * 4 + r4) * HS * 4));
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
        if (HCG_ABLATE & 8) { v4[it] = make_float4((float)row, 1.f, 2.f, 3.f); continue; }
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

// acc{0,1}[TM x 64] = buf[TM x KPAD] * Wreg  (Wreg[nb][s]: B operand of k-step s, column block nb)
template <int KPAD>
__device__ __forceinline__ void tile_gemm(const float* buf, const float (&wreg)[2][KPAD / 2], f32x16& acc0,
                                          f32x16& acc1, int lane) {
  const int r = lane & 31, h = lane >> 5;
#pragma unroll
  for (int t = 0; t < KPAD / 8; ++t) {
    const float4 a = *reinterpret_cast<const float4*>(buf + r * HS + 8 * t + 4 * h);
    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, wreg[0][4 * t + 0], acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, wreg[1][4 * t + 0], acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, wreg[0][4 * t + 1], acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, wreg[1][4 * t + 1], acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, wreg[0][4 * t + 2], acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, wreg[1][4 * t + 2], acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, wreg[0][4 * t + 3], acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, wreg[1][4 * t + 3], acc1, 0, 0, 0);
  }
}

// Block-cooperative staging of a [rows x cols] row-major global matrix into LDS as dst[r * ld + c]
// (zero padded to cols_pad); every thread of the 512-thread workgroup takes part.
__device__ __forceinline__ void stage_matrix(float* dst, int ld, const float* __restrict__ g, int rows, int cols,
                                             int cols_pad) {
  for (int idx = threadIdx.x; idx < rows * cols_pad; idx += WAVES * 64) {
    const int r = idx / cols_pad, c = idx - r * cols_pad;
    const float v = g[(size_t)r * cols + (c < cols ? c : cols - 1)];
    dst[r * ld + c] = c < cols ? v : 0.f;
  }
}

// =====================================================================================================
// forward:  out = LeakyReLU( Ahat (x W^T) + b ),  optional pooled epilogue emb[g] = [max, mean]
// =====================================================================================================
template <int KPAD, bool VEC, bool POOL>
__global__ __launch_bounds__(WAVES * 64, 2) void k_fused_layer_fwd(
    const float* __restrict__ x, int F, const float* __restrict__ W, const float* __restrict__ bias,
    const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col, const float* __restrict__ dinv,
    const int32_t* __restrict__ graph_ptr, const int32_t* __restrict__ edge_ptr, int64_t N, int gpt, int B,
    int num_tiles, float slope, int apply_act, float* __restrict__ out, float* __restrict__ emb,
    int32_t* __restrict__ status) {
  __shared__ WaveLds lds[WAVES];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform -> scalar tile loads
  WaveLds& L = lds[wave];
  const int r = lane & 31, h = lane >> 5, q = lane & 15, r4 = lane >> 4;
  const int stride = gridDim.x * WAVES;
  int t = blockIdx.x * WAVES + wave;
  bool have = t < num_tiles;

  // first tile's loads go out before anything else
  TileInfo ti;
  TileIndex tix;
  Stager<KPAD, VEC> sx;
  if (have) {
    ti = tile_scalars(t, gpt, B, graph_ptr, edge_ptr, lane, status);
    sx.load(x, F, N, ti.nbase, ti.n, lane);
    tix.load(ti, graph_ptr, rowptr, col, dinv, N, lane);
  }

  // weight operand: W [64][F] -> LDS (coalesced, once per workgroup) -> 64 VGPRs per lane.
  // B[k][j] = W[j][k]; k-step s = 4t+u <-> k = 8t + 4h + u.  LDS image [n][KPAD + 1]: lane r reads
  // row nb*32 + r at a fixed k -> stride KPAD + 1 floats -> conflict-free.
  float wreg[2][KPAD / 2];
  {
    float* wl = reinterpret_cast<float*>(&lds[0]);   // 64 * (KPAD + 1) floats fit in one wave region
    stage_matrix(wl, KPAD + 1, W, DD, F, KPAD);
    __syncthreads();
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int s = 0; s < KPAD / 2; ++s) wreg[nb][s] = wl[(nb * 32 + r) * (KPAD + 1) + 8 * (s >> 2) + 4 * h + (s & 3)];
    __syncthreads();
  }
  for (int k = lane; k < HS; k += 64) L.zeroA[k] = 0.f;
  const float4 bia = *reinterpret_cast<const float4*>(bias + 4 * q);
  STAMP(0);
  int stamp_it = 0;
  if (have) {
    tix.write(L, ti, lane, status);
    sx.write(L.bufB, F, ti.n, lane);
  }

  while (have) {
    STAMP(1 + 8 * stamp_it);
    // prefetch the next tile of this wave (registers only) while this one computes
    const int tn = t + stride;
    const bool have_next = tn < num_tiles;
    TileInfo tin;
    TileIndex tixn;
    Stager<KPAD, VEC> sxn;
    if (VEC && have_next) {   // (the scalar-staging variants are short of registers: they load after the compute)
      tin = tile_scalars(tn, gpt, B, graph_ptr, edge_ptr, lane, status);
      sxn.load(x, F, N, tin.nbase, tin.n, lane);
      tixn.load(tin, graph_ptr, rowptr, col, dinv, N, lane);
    }
    STAMP(2 + 8 * stamp_it);

    f32x16 acc0, acc1;
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }
    if (!(HCG_ABLATE & 1)) tile_gemm<KPAD>(L.bufB, wreg, acc0, acc1, lane);
    STAMP(3 + 8 * stamp_it);

    // h' = dinv (.) h  -> bufA   (C/D map: col = lane&31, row = (i&3) + 8*(i>>2) + 4*h)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int row = (i & 3) + 8 * (i >> 2) + 4 * h;
      const float dv = L.ldinv[row];
      L.bufA[row * HS + r] = dv * acc0[i];
      L.bufA[row * HS + 32 + r] = dv * acc1[i];
    }
    STAMP(4 + 8 * stamp_it);

    // segmented sum + bias + LeakyReLU, 4 node rows per wave step, all LDS reads batched
    float4 y[TM / 4];
    if (!(HCG_ABLATE & 2)) {
      gather_tile(L, L.bufA, ti, col, q, r4, status, y);
    } else {
#pragma unroll
      for (int it = 0; it < TM / 4; ++it) y[it] = *reinterpret_cast<const float4*>(L.bufA + (it * 4 + r4) * HS + 4 * q);
    }
#pragma unroll
    for (int it = 0; it < TM / 4; ++it) {
      const int i = it * 4 + r4;
      const float di = L.ldinv[i];
      float4 v = make_float4(di * y[it].x + bia.x, di * y[it].y + bia.y, di * y[it].z + bia.z, di * y[it].w + bia.w);
      // LeakyReLU as max(v, slope*v): exact for 0 <= slope <= 1 (the host rejects other slopes), 2 ops not 3
      if (apply_act) v = make_float4(fmaxf(v.x, slope * v.x), fmaxf(v.y, slope * v.y), fmaxf(v.z, slope * v.z), fmaxf(v.w, slope * v.w));
      y[it] = v;
      if (!(HCG_ABLATE & 4)) {
        if (i < ti.n) *reinterpret_cast<float4*>(out + (size_t)(ti.nbase + i) * DD + 4 * q) = v;
      } else if (v.x == 12345.678f) {
        out[0] = v.y;   // keeps the values live without storing them
      }
    }
    if (POOL) {
      for (int g = ti.g0; g < ti.g1; ++g) {
        const int gb = L.lgp[g - ti.g0], ge = L.lgp[g - ti.g0 + 1];
        float4 pmax = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
        float4 psum = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int it = 0; it < TM / 4; ++it) {
          const int i = it * 4 + r4;
          if (i >= gb && i < ge) { pmax = f4_max(pmax, y[it]); psum = f4_add(psum, y[it]); }
        }
        pmax = f4_max(pmax, f4_shfl_xor(pmax, 16));
        pmax = f4_max(pmax, f4_shfl_xor(pmax, 32));
        psum = f4_add(psum, f4_shfl_xor(psum, 16));
        psum = f4_add(psum, f4_shfl_xor(psum, 32));
        if (r4 == 0) {
          const int n = ge - gb;
          const float cnt = (float)(n > 0 ? n : 1);
          if (n <= 0) pmax = make_float4(0.f, 0.f, 0.f, 0.f);
          *reinterpret_cast<float4*>(emb + (size_t)g * 2 * DD + 4 * q) = pmax;
          *reinterpret_cast<float4*>(emb + (size_t)g * 2 * DD + DD + 4 * q) =
              make_float4(psum.x / cnt, psum.y / cnt, psum.z / cnt, psum.w / cnt);
        }
      }
    }
    STAMP(5 + 8 * stamp_it);
    if (stamp_it < 6) ++stamp_it;

    have = have_next;
    if (have_next) {
      if (!VEC) {
        tin = tile_scalars(tn, gpt, B, graph_ptr, edge_ptr, lane, status);
        sxn.load(x, F, N, tin.nbase, tin.n, lane);
        tixn.load(tin, graph_ptr, rowptr, col, dinv, N, lane);
      }
      t = tn;
      ti = tin;
      tixn.write(L, ti, lane, status);
      sxn.write(L.bufB, F, ti.n, lane);
    }
  }
  STAMP(63);
}

// =====================================================================================================
// backward of one layer.
//   dY = dA (.) leaky'(A)            dA = dout, or (POOLG) the pooled-gradient expansion
//   db += colsum dY ;  dH = Ahat^T dY ;  dW += dH^T x ;  dx = dH W  (NEEDS_DX)
// per-workgroup partial sums go to `partials[blockIdx][64*KPAD + 64]`.
// =====================================================================================================
template <int KPAD, bool VEC, bool NEEDS_DX, bool POOLG>
__global__ __launch_bounds__(WAVES * 64, 2) void k_fused_layer_bwd(
    const float* __restrict__ dout, const float* __restrict__ demb, const float* __restrict__ emb,
    const float* __restrict__ a_out, const float* __restrict__ x, int F, const float* __restrict__ W,
    const int32_t* __restrict__ rowptr_t, const int32_t* __restrict__ col_t, const float* __restrict__ dinv,
    const int32_t* __restrict__ graph_ptr, const int32_t* __restrict__ edge_ptr, int64_t N, int gpt, int B,
    int num_tiles, float slope, int apply_act, float* __restrict__ dx, float* __restrict__ partials,
    int32_t* __restrict__ status) {
  __shared__ WaveLds lds[WAVES];
  __shared__ float wlds[NEEDS_DX ? DD * KPAD : 4];   // dx operand W [d][f], shared by the 8 waves
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  WaveLds& L = lds[wave];
  const int r = lane & 31, h = lane >> 5, q = lane & 15, r4 = lane >> 4;
  constexpr int NBF = KPAD / 32;  // column blocks of dW (input-feature dimension)
  const int stride = gridDim.x * WAVES;

  if (NEEDS_DX) {
    stage_matrix(wlds, KPAD, W, DD, F, KPAD);
    __syncthreads();
  }
  for (int k = lane; k < HS; k += 64) L.zeroA[k] = 0.f;

  f32x16 dw[2][NBF];  // dW[d-block][f-block], accumulated over every tile of this wave
#pragma unroll
  for (int mb = 0; mb < 2; ++mb)
#pragma unroll
    for (int nb = 0; nb < NBF; ++nb)
#pragma unroll
      for (int i = 0; i < 16; ++i) dw[mb][nb][i] = 0.f;
  float4 dbacc = make_float4(0.f, 0.f, 0.f, 0.f);

  for (int t = blockIdx.x * WAVES + wave; t < num_tiles; t += stride) {
    const TileInfo ti = tile_scalars(t, gpt, B, graph_ptr, edge_ptr, lane, status);
    // the global reads of step 1 go in flight together: A rows, dA rows, index data
    Stager<DD, true> sa;
    sa.load(a_out, DD, N, ti.nbase, ti.n, lane);
    Stager<DD, true> sd;
    if (!POOLG) sd.load(dout, DD, N, ti.nbase, ti.n, lane);
    TileIndex tix;
    tix.load(ti, graph_ptr, rowptr_t, col_t, dinv, N, lane);
    tix.write(L, ti, lane, status);

    // ---- 1. dY' = dinv (.) dA (.) leaky'(A) -> bufA (rows >= n zero); A / dA stay in registers
    float4 dy[TM / 4];
    if (POOLG) {
#pragma unroll
      for (int it = 0; it < TM / 4; ++it) dy[it] = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int g = ti.g0; g < ti.g1; ++g) {
        const int gb = L.lgp[g - ti.g0], ge = L.lgp[g - ti.g0 + 1];
        if (ge <= gb) continue;
        const float4 gmx = *reinterpret_cast<const float4*>(emb + (size_t)g * 2 * DD + 4 * q);
        const float4 dmx = *reinterpret_cast<const float4*>(demb + (size_t)g * 2 * DD + 4 * q);
        float4 dmean = *reinterpret_cast<const float4*>(demb + (size_t)g * 2 * DD + DD + 4 * q);
        const float cnt = (float)(ge - gb);
        dmean = make_float4(dmean.x / cnt, dmean.y / cnt, dmean.z / cnt, dmean.w / cnt);
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
        if (apply_act) {
          const float4 a = sa.v4[it];
          d.x *= hcg_leaky_grad(a.x, slope); d.y *= hcg_leaky_grad(a.y, slope);
          d.z *= hcg_leaky_grad(a.z, slope); d.w *= hcg_leaky_grad(a.w, slope);
        }
        dbacc = f4_add(dbacc, d);
        const float di = L.ldinv[i];
        d = make_float4(di * d.x, di * d.y, di * d.z, di * d.w);
      }
      *reinterpret_cast<float4*>(L.bufA + i * HS + 4 * q) = d;
    }

    // x rows: issued now (the A / dA registers are free again), consumed after the segmented sum
    Stager<KPAD, VEC> sx;
    if (VEC) sx.load(x, F, N, ti.nbase, ti.n, lane);

    // ---- 2. dH_j = dinv_j * ( sum_{k in out(j)} dY'_k + dY'_j ) -> bufB (rows >= n zero)
    {
      float4 res[TM / 4];
      gather_tile(L, L.bufA, ti, col_t, q, r4, status, res);
#pragma unroll
      for (int it = 0; it < TM / 4; ++it) {
        const int i = it * 4 + r4;
        const float di = i < ti.n ? L.ldinv[i] : 0.f;
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i < ti.n) s = make_float4(di * res[it].x, di * res[it].y, di * res[it].z, di * res[it].w);
        *reinterpret_cast<float4*>(L.bufB + i * HS + 4 * q) = s;
      }
    }

    // ---- 3. x tile -> bufA ; dW += dH^T x   (K = node rows: k-step s <-> node 2s + h)
    if (!VEC) sx.load(x, F, N, ti.nbase, ti.n, lane);   // scalar-staging variants: short of registers, load late
    sx.write(L.bufA, F, ti.n, lane);
#pragma unroll
    for (int s = 0; s < TM / 2; ++s) {
      const int node = 2 * s + h;
      const float a0 = L.bufB[node * HS + r], a1 = L.bufB[node * HS + 32 + r];
#pragma unroll
      for (int nb = 0; nb < NBF; ++nb) {
        const float b = L.bufA[node * HS + nb * 32 + r];
        dw[0][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b, dw[0][nb], 0, 0, 0);
        dw[1][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b, dw[1][nb], 0, 0, 0);
      }
    }

    // ---- 4. dx = dH W   (B[k = d][j = f] = W[d][f] from the workgroup's LDS copy; k-step s <-> d = 8t+4h+u)
    if (NEEDS_DX) {
      f32x16 dxa[NBF];
#pragma unroll
      for (int nb = 0; nb < NBF; ++nb)
#pragma unroll
        for (int i = 0; i < 16; ++i) dxa[nb][i] = 0.f;
#pragma unroll
      for (int t8 = 0; t8 < DD / 8; ++t8) {
        const float4 a = *reinterpret_cast<const float4*>(L.bufB + r * HS + 8 * t8 + 4 * h);
        const float av[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int d = 8 * t8 + 4 * h + u;
#pragma unroll
          for (int nb = 0; nb < NBF; ++nb)
            dxa[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], wlds[d * KPAD + nb * 32 + r], dxa[nb], 0, 0, 0);
        }
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int row = (i & 3) + 8 * (i >> 2) + 4 * h;
        if (row < ti.n) {
#pragma unroll
          for (int nb = 0; nb < NBF; ++nb) {
            const int f = nb * 32 + r;
            if (f < F) dx[(size_t)(ti.nbase + row) * F + f] = dxa[nb][i];
          }
        }
      }
    }
  }

  // ---- combine the waves of this workgroup (fixed order) and publish one partial slab
  __syncthreads();  // every wave is done with its tile buffers
  float* mine = reinterpret_cast<float*>(&lds[wave]);  // >= 64*KPAD + 64 floats (two tile buffers)
#pragma unroll
  for (int mb = 0; mb < 2; ++mb)
#pragma unroll
    for (int nb = 0; nb < NBF; ++nb)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int d = mb * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
        mine[d * KPAD + nb * 32 + r] = dw[mb][nb][i];
      }
  dbacc = f4_add(dbacc, f4_shfl_xor(dbacc, 16));
  dbacc = f4_add(dbacc, f4_shfl_xor(dbacc, 32));
  if (r4 == 0) *reinterpret_cast<float4*>(mine + DD * KPAD + 4 * q) = dbacc;
  __syncthreads();
  float* slab = partials + (size_t)blockIdx.x * (DD * KPAD + DD);
  for (int idx = threadIdx.x; idx < DD * KPAD + DD; idx += WAVES * 64) {
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < WAVES; ++w) s += reinterpret_cast<const float*>(&lds[w])[idx];
    slab[idx] = s;
  }
}

// dW[d][f] = sum_b slab[b][d*KPAD + f],  db[d] = sum_b slab[b][64*KPAD + d]   (fixed order over b)
constexpr int RED_SLICES = 16;
__global__ __launch_bounds__(256) void k_fused_reduce(const float* __restrict__ partials, int nslabs, int KPAD, int F,
                                                      float* __restrict__ dW, float* __restrict__ db) {
  __shared__ float part[RED_SLICES][16];
  const int slab_floats = DD * KPAD + DD;
  const int o = threadIdx.x & 15, sl = threadIdx.x >> 4;
  const int idx = blockIdx.x * 16 + o;
  float s = 0.f;
  if (idx < slab_floats) {
    for (int b = sl; b < nslabs; b += RED_SLICES) s += partials[(size_t)b * slab_floats + idx];
  }
  part[sl][o] = s;
  __syncthreads();
  if (sl == 0 && idx < slab_floats) {
    float tot = 0.f;
#pragma unroll
    for (int k = 0; k < RED_SLICES; ++k) tot += part[k][o];
    if (idx < DD * KPAD) {
      const int d = idx / KPAD, f = idx % KPAD;
      if (f < F) dW[(size_t)d * F + f] = tot;
    } else {
      db[idx - DD * KPAD] = tot;
    }
  }
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

extern "C" int hcg_fused_layer_fwd(const float* x, const float* W, const float* b, const int32_t* rowptr,
                                   const int32_t* col, const float* dinv, const int32_t* graph_ptr,
                                   const int32_t* edge_ptr, int64_t N, int64_t B,
                                   int64_t F, int64_t D, int graphs_per_tile, float slope, int apply_act, float* out,
                                   float* emb, int32_t* status, hcg_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (D != DD || F < 1 || F > 64 || graphs_per_tile < 1) return HCG_ERR_UNSUPPORTED;
  if (apply_act && !(slope >= 0.f && slope <= 1.f)) return HCG_ERR_UNSUPPORTED;  // LeakyReLU is evaluated as max(v, slope*v)
  if (N < 0 || B < 0) return HCG_ERR_INVALID_ARG;
  if (B == 0 || N == 0) return HCG_OK;
  if (!x || !W || !b || !rowptr || !dinv || !graph_ptr || !edge_ptr || !out || !status) return HCG_ERR_INVALID_ARG;
  const int tiles = (int)((B + graphs_per_tile - 1) / graphs_per_tile);
  const int grid = pick_grid(tiles);
  const bool vec = (F == 64 || F == 32) && ((uintptr_t)x % 16 == 0);
  const dim3 g(grid), blk(WAVES * 64);
#define LAUNCH_FWD(KP, VC, PL)                                                                                       \
  hipLaunchKernelGGL((k_fused_layer_fwd<KP, VC, PL>), g, blk, 0, stream, x, (int)F, W, b, rowptr, col, dinv, graph_ptr, \
                     edge_ptr, N, graphs_per_tile, (int)B, tiles, slope, apply_act, out, emb, status)
  if (F <= 32) {
    if (vec) { if (emb) LAUNCH_FWD(32, true, true); else LAUNCH_FWD(32, true, false); }
    else     { if (emb) LAUNCH_FWD(32, false, true); else LAUNCH_FWD(32, false, false); }
  } else {
    if (vec) { if (emb) LAUNCH_FWD(64, true, true); else LAUNCH_FWD(64, true, false); }
    else     { if (emb) LAUNCH_FWD(64, false, true); else LAUNCH_FWD(64, false, false); }
  }
#undef LAUNCH_FWD
  HCG_CHECK_LAUNCH();
  return HCG_OK;
}

extern "C" int hcg_fused_layer_bwd(const float* dout, const float* demb, const float* emb, const float* out,
                                   const float* x, const float* W, const int32_t* rowptr_t, const int32_t* col_t,
                                   const float* dinv, const int32_t* graph_ptr, const int32_t* edge_ptr, int64_t N,
                                   int64_t B, int64_t F, int64_t D,
                                   int graphs_per_tile, float slope, int apply_act, float* dx, int32_t* status,
                                   void* workspace, size_t workspace_bytes, hcg_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (D != DD || F < 1 || F > 64 || graphs_per_tile < 1) return HCG_ERR_UNSUPPORTED;
  if (N < 0 || B < 0 || !W || !workspace) return HCG_ERR_INVALID_ARG;
  const bool poolg = (dout == nullptr);
  if (poolg && (!demb || !emb)) return HCG_ERR_INVALID_ARG;
  if (N > 0 && B > 0 && (!out || !x || !rowptr_t || !dinv || !graph_ptr || !edge_ptr || !status)) return HCG_ERR_INVALID_ARG;
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
#define LAUNCH_BWD(KP, VC, DX, PG)                                                                                  \
  hipLaunchKernelGGL((k_fused_layer_bwd<KP, VC, DX, PG>), g, blk, 0, stream, dout, demb, emb, out, x, (int)F, W,    \
                     rowptr_t, col_t, dinv, graph_ptr, edge_ptr, N, graphs_per_tile, (int)B, tiles, slope, apply_act, dx,    \
                     partials, status)
#define DISPATCH_BWD(KP, VC)                                                             \
  do {                                                                                   \
    if (ndx) { if (poolg) LAUNCH_BWD(KP, VC, true, true); else LAUNCH_BWD(KP, VC, true, false); } \
    else     { if (poolg) LAUNCH_BWD(KP, VC, false, true); else LAUNCH_BWD(KP, VC, false, false); } \
  } while (0)
    if (kpad == 32) { if (vec) DISPATCH_BWD(32, true); else DISPATCH_BWD(32, false); }
    else            { if (vec) DISPATCH_BWD(64, true); else DISPATCH_BWD(64, false); }
#undef DISPATCH_BWD
#undef LAUNCH_BWD
    HCG_CHECK_LAUNCH();
  }
  return HCG_OK;
}

// second stage of the backward: dW[D, F], db[D] <- the per-workgroup slabs left in `workspace` by
// hcg_fused_layer_bwd (same B, F, D, graphs_per_tile), summed in a fixed order.
extern "C" int hcg_fused_reduce_grads(const void* workspace, size_t workspace_bytes, int64_t N, int64_t B, int64_t F,
                                      int64_t D, int graphs_per_tile, float* dW, float* db, hcg_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (D != DD || F < 1 || F > 64 || graphs_per_tile < 1 || !dW || !db) return HCG_ERR_INVALID_ARG;
  const int kpad = F <= 32 ? 32 : 64;
  const int tiles = (int)((B + graphs_per_tile - 1) / graphs_per_tile);
  const int grid = (N > 0 && B > 0) ? pick_grid(tiles) : 0;
  const int slab_floats = DD * kpad + DD;
  if (workspace_bytes < (size_t)grid * slab_floats * sizeof(float)) return HCG_ERR_WORKSPACE;
  hipLaunchKernelGGL(k_fused_reduce, dim3((slab_floats + 15) / 16), dim3(256), 0, stream, (const float*)workspace, grid,
                     kpad, (int)F, dW, db);
  HCG_CHECK_LAUNCH();
  return HCG_OK;
}
