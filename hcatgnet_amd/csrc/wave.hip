// One graph per WAVE: graphs of 33 .. 64 nodes (<= 256 directed edges), D = 64, F <= 64.
//
// Between the small-graph tiles (fused.hip: a 32-row tile per wave, dense tile adjacency on the matrix cores) and the
// one-graph-per-workgroup kernels (mid.hip: 8 waves and ~7 workgroup barriers per graph) sit the ragged ~30-atom batches
// of SURVEY 8(d) (n_g ~ U{24..36}) and the smaller half of the reference's own graphs.  On mid.hip they pay a whole
// workgroup's latency chain per graph (2 of 8 waves have a row block to transform, the rest wait at barriers).  Here a
// graph is owned by ONE wave, like a tile in fused.hip: no workgroup barrier in the steady state, every wave streams its
// own graphs, LDS is wave-private:
//   * gcn_norm on chip: the graph's raw COO edges (4 per lane, requested one graph ahead) -> in-degree, dinv and a CSR
//     in LDS (integer LDS atomics for the counting sort, every row then sorted by id: fixed summation order);
//   * H' = dinv . (X W^T): X rows -> one [64][68] fp32 LDS tile, the two 32-row blocks transformed in turn on the
//     matrix cores (split-bf16 MFMAs, split_mfma.h; a graph of <= 32 nodes costs ONE block), H' written back in place;
//   * Y_i = H'_i + sum_k H'_{col k}: wavefront segmented sum out of LDS (16 lanes x float4 per row, 4 rows per pass);
//   * out = LeakyReLU(dinv . Y + b): one 256-byte store per row; optional [max, mean] epilogue.
// backward mirrors mid.hip's (dY' tile -> transpose segmented sum -> dH tile -> dW on the matrix cores with K = nodes,
// accumulated in registers over ALL graphs of the wave; dX = dH W), the waves of a workgroup combine in a fixed order.
// Same slab layout and reduction as the other families; selected inside hcg_mid_* (mid.hip) -- no API of its own.
#include "common.h"
#include "split_mfma.h"

namespace {

constexpr int WN = 64;             // nodes per graph (two MFMA row blocks)
constexpr int WE = 256;            // directed edges per graph
constexpr int WEPT = WE / 64;      // edges per lane, kept in registers from the load to the CSR fill
constexpr int W_SMALL = 65 * 4 + 12 + 3 * 64 * 4 + WE * 2;   // rowptr (padded to 272) | cursor | degin | dinv | col (u16)

__host__ __device__ constexpr size_t w_wave_bytes(int tiles) { return (size_t)tiles * WN * HS * 4 + W_SMALL; }

struct WLds {
  float* t0;              // [64][HS]  forward: X -> H' ; backward: dY' -> dH
  float* t1;              // (unused: one tile per wave)
  int* rowptr;            // [65]
  int* cursor;            // [64]  row sizes, then fill cursors
  int* degin;             // [64]  in-degree (transpose CSR: rows are sources)
  float* dinv;            // [64]
  unsigned short* col;    // [WE]
};

__device__ __forceinline__ WLds w_carve(char* base, int tiles) {
  WLds L;
  L.t0 = reinterpret_cast<float*>(base);
  L.t1 = tiles > 1 ? L.t0 + WN * HS : nullptr;
  char* s = base + (size_t)tiles * WN * HS * 4;
  L.rowptr = reinterpret_cast<int*>(s);
  L.cursor = reinterpret_cast<int*>(s + 272);
  L.degin = L.cursor + 64;
  L.dinv = reinterpret_cast<float*>(L.degin + 64);
  L.col = reinterpret_cast<unsigned short*>(L.dinv + 64);
  return L;
}

// LDS is wave-private and a wave's LDS operations execute in order: nothing to wait for across lanes, the compiler only
// has to keep the program order of the accesses
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

struct WGraph { int nbase, n, ebase, ne, nblk, nld; };   // nld: row base for (clamped) LOADS: an empty graph at the very end
                                                         // of the batch has nbase == N, one row past the matrices

__device__ __forceinline__ WGraph w_graph(int g, int B, const int32_t* __restrict__ graph_ptr, const int32_t* __restrict__ edge_ptr,
                                          int lane, int32_t* status) {
  WGraph gi;
  g = __builtin_amdgcn_readfirstlane(g < B ? g : B - 1);
  gi.nbase = graph_ptr[g];
  gi.n = graph_ptr[g + 1] - gi.nbase;
  gi.ebase = edge_ptr[g];
  gi.ne = edge_ptr[g + 1] - gi.ebase;
  // host metadata was wrong: the graph is refused and reported.  (The selects stay OUTSIDE the reporting lane's branch:
  // assigned inside it, n and ne became per-lane registers and every address derived from them a 64-bit vector computation.)
  const bool bad = gi.n < 0 || gi.n > WN || gi.ne < 0 || gi.ne > WE;
  gi.n = __builtin_amdgcn_readfirstlane(bad ? 0 : gi.n);
  gi.ne = __builtin_amdgcn_readfirstlane(bad ? 0 : gi.ne);
  if (bad && lane == 0) atomicOr(status, HCG_STATUS_SHAPE_LIMIT);
  gi.nblk = (gi.n + 31) / 32;
  gi.nld = gi.n > 0 ? gi.nbase : 0;
  return gi;
}

// the graph's scalars pinned into SGPRs: every guard built from them is then a scalar branch (carried through the graph
// loop as a struct the compiler otherwise keeps them in VGPRs and turns `if (4 * j < rows)` around a load into a
// per-lane branch with its own s_waitcnt vmcnt(0))
__device__ __forceinline__ WGraph w_uniform(const WGraph& a) {
  WGraph u;
  u.nbase = __builtin_amdgcn_readfirstlane(a.nbase);
  u.n = __builtin_amdgcn_readfirstlane(a.n);
  u.ebase = __builtin_amdgcn_readfirstlane(a.ebase);
  u.ne = __builtin_amdgcn_readfirstlane(a.ne);
  u.nblk = __builtin_amdgcn_readfirstlane(a.nblk);
  u.nld = __builtin_amdgcn_readfirstlane(a.nld);
  return u;
}

struct WEdges {   // loads only (unconditional, clamped): consumed one graph later
  long long s[WEPT], d[WEPT];
  __device__ __forceinline__ void load(const WGraph& gi, const int64_t* __restrict__ ei, int64_t E, int lane) {
    // (wave-uniform bases + one unsigned 32-bit byte offset per slot: the scalar-base form of global_load; clamps are scalar)
    long long eb = gi.ebase;
    eb = eb < 0 ? 0 : (eb > E - 1 ? E - 1 : eb);
    const long long room = E - eb;
    const int nec = (long long)gi.ne < room ? gi.ne : (int)room;
    const int last = nec > 0 ? nec - 1 : 0;
    const char* sb = reinterpret_cast<const char*>(ei + eb);
    const char* db = reinterpret_cast<const char*>(ei + E + eb);
#pragma unroll
    for (int j = 0; j < WEPT; ++j) {
      const int e = lane + 64 * j;
      const unsigned off = 8u * (unsigned)(e < last ? e : last);
      s[j] = *reinterpret_cast<const long long*>(sb + off);
      d[j] = *reinterpret_cast<const long long*>(db + off);
    }
  }
};

// dinv = (1 + in-degree)^-1/2 and a CSR of the graph in LDS; BY_SRC = false: rows = targets (forward aggregation),
// true: rows = sources (the transpose).  Explicit (i, i) edges collapse into the unit self loop.  One wave.
template <bool BY_SRC>
__device__ __forceinline__ void w_build_csr(const WLds& L, const WGraph& gi, const WEdges& er, int lane, int32_t* status) {
  L.cursor[lane] = 0;
  L.degin[lane] = 0;
  wave_sync();
  unsigned short es[WEPT], ed[WEPT];
  bool bad = false;
#pragma unroll
  for (int j = 0; j < WEPT; ++j) {
    const int e = lane + 64 * j;
    es[j] = 0xffff;
    ed[j] = 0xffff;
    if (e < gi.ne) {
      const long long s = er.s[j], d = er.d[j];
      const unsigned sl = (unsigned)((int)s - gi.nbase), dl = (unsigned)((int)d - gi.nbase);
      const bool ok = sl < (unsigned)gi.n && dl < (unsigned)gi.n && (s >> 31) == 0 && (d >> 31) == 0;
      bad |= !ok;
      if (ok && sl != dl) {
        es[j] = (unsigned short)sl;
        ed[j] = (unsigned short)dl;
        atomicAdd(&L.cursor[BY_SRC ? sl : dl], 1);
        if (BY_SRC) atomicAdd(&L.degin[dl], 1);
      }
    }
  }
  if (__ballot(bad) != 0ull && lane == 0) atomicOr(status, HCG_STATUS_EDGE_UNGROUPED);   // edge leaves its graph: ignored
  wave_sync();
  const int cnt = L.cursor[lane];
  int incl = cnt;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int t = __shfl_up(incl, off, 64);
    if (lane >= off) incl += t;
  }
  L.rowptr[lane] = incl - cnt;
  if (lane == 63) L.rowptr[64] = incl;
  const int degin = BY_SRC ? L.degin[lane] : cnt;
  L.dinv[lane] = lane < gi.n ? 1.0f / sqrtf(1.0f + (float)degin) : 0.f;
  wave_sync();
  L.cursor[lane] = incl - cnt;
  wave_sync();
#pragma unroll
  for (int j = 0; j < WEPT; ++j) {
    if (es[j] != 0xffff) {
      const int p = atomicAdd(&L.cursor[BY_SRC ? es[j] : ed[j]], 1);
      L.col[p] = BY_SRC ? ed[j] : es[j];
    }
  }
  wave_sync();
  {   // every row sorted by id: fixed summation order whatever order the LDS atomics ran in.  Rows of <= 4 entries (all
      // of them in molecular graphs) through a register network, longer ones by insertion.
    const int kb = lane < gi.n ? L.rowptr[lane] : 0, ke = lane < gi.n ? L.rowptr[lane + 1] : 0;
    const int len = ke - kb;
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
  wave_sync();
}

// acc = t[row] + sum_{k in [kb, ke)} t[col[k]] for this lane's (row, 4q..4q+3) slot.  Head: the row itself and its first
// four neighbours, requested together (independent LDS reads, straight-line code: several rows' chains interleave when
// the caller unrolls).  Tail: rows with more than four entries (none in molecular graphs) loop on, behind ONE wave-uniform
// branch per group of rows.
__device__ __forceinline__ float4 w_row_sum_head(const float* t, const unsigned short* col, int row, int kb, int ke, int q) {
  float4 acc = *reinterpret_cast<const float4*>(t + row * HS + 4 * q);
  int c[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) c[j] = col[min(kb + j, WE - 1)];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float4 v = *reinterpret_cast<const float4*>(t + (kb + j < ke ? c[j] : row) * HS + 4 * q);
    const float m = kb + j < ke ? 1.f : 0.f;
    acc.x = fmaf(m, v.x, acc.x); acc.y = fmaf(m, v.y, acc.y); acc.z = fmaf(m, v.z, acc.z); acc.w = fmaf(m, v.w, acc.w);
  }
  return acc;
}
__device__ __forceinline__ void w_row_sum_tail(float4& acc, const float* t, const unsigned short* col, int kb, int ke, int q) {
  for (int k = kb + 4; __any(k < ke); ++k) {
    if (k < ke) {
      const float4 v = *reinterpret_cast<const float4*>(t + col[k] * HS + 4 * q);
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
  }
}

// rows [nbase, nbase + n) of a row-major [*, F] matrix -> t[row][0 .. KPAD), zero padded past F and up to `rows` rows.
// Wide form (F == KPAD, 16-byte rows): 8 float4 per lane in flight per half.
template <int KPAD>
__device__ __forceinline__ void w_stage_rows(float* t, const float* __restrict__ g, int F, int nbase, int n, int rows, int lane) {
  if (F == KPAD && ((uintptr_t)g % 16 == 0)) {
    constexpr int PER_ROW = KPAD / 4;                 // float4 per row: 16 (8)
    constexpr int RPP = 64 / PER_ROW;                 // rows per pass: 4 (8)
    const int c4 = lane % PER_ROW, r0 = lane / PER_ROW;
    const char* gb = reinterpret_cast<const char*>(g + (size_t)(n > 0 ? nbase : 0) * F);   // (wave-uniform base + 32-bit byte offsets)
    const unsigned F4 = 4u * (unsigned)F;
    const int nlast = n > 0 ? n - 1 : 0;
    for (int base = 0; base < rows; base += 8 * RPP) {     // 32 (64) rows per batch of 8 loads
      float4 v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int row = base + j * RPP + r0;
        v[j] = *reinterpret_cast<const float4*>(gb + __umul24((unsigned)(row < nlast ? row : nlast), F4) + 16u * (unsigned)c4);
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int row = base + j * RPP + r0;
        if (row < rows) *reinterpret_cast<float4*>(t + row * HS + 4 * c4) = row < n ? v[j] : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
  } else {
    for (int idx = lane; idx < rows * KPAD; idx += 64) {
      const int row = idx / KPAD, c = idx - row * KPAD;
      t[row * HS + c] = (row < n && c < F) ? g[(size_t)(nbase + row) * F + c] : 0.f;
    }
  }
}

// the wide form in two phases, so that a graph's first rows can be requested a graph ahead: load() only issues global
// loads (8 float4 per lane = rows [base, base + 512 / KPAD * 4)), write() only touches LDS
template <int KPAD>
struct WRowsAhead {
  static constexpr int PER_ROW = KPAD / 4, RPP = 64 / PER_ROW, ROWS = 8 * RPP;
  float4 v[8];
  __device__ __forceinline__ void load(const float* __restrict__ g, int F, int nbase, int n, int base, int lane) {
    const int c4 = lane % PER_ROW, r0 = lane / PER_ROW;
    const char* gb = reinterpret_cast<const char*>(g + (size_t)(n > 0 ? nbase : 0) * F);   // (wave-uniform base + 32-bit byte offsets)
    const unsigned F4 = 4u * (unsigned)F;
    const int nlast = n > 0 ? n - 1 : 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int row = base + j * RPP + r0;
      v[j] = *reinterpret_cast<const float4*>(gb + __umul24((unsigned)(row < nlast ? row : nlast), F4) + 16u * (unsigned)c4);
    }
  }
  __device__ __forceinline__ void write(float* t, int n, int rows, int base, int lane) const {
    const int c4 = lane % PER_ROW, r0 = lane / PER_ROW;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int row = base + j * RPP + r0;
      if (row < rows) *reinterpret_cast<float4*>(t + row * HS + 4 * c4) = row < n ? v[j] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
};

// like stage_weight_split (split_mfma.h) for a thread count that need not divide the image
template <bool TRANS, int NT, int ROWS, int K>
__device__ __forceinline__ void w_stage_weight(short* wl, const float* __restrict__ g, int grows, int cols) {
  constexpr int ld = K + WPAD, plane = ROWS * ld, total = ROWS * K, PER = (total + NT - 1) / NT;
  float v[PER];
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    int idx = threadIdx.x + j * NT;
    idx = idx < total ? idx : total - 1;
    const int a = TRANS ? idx / ROWS : idx / K, b = TRANS ? idx - a * ROWS : idx - a * K;   // TRANS: (d, f) else (n, k)
    v[j] = TRANS ? g[(a < grows ? a : grows - 1) * cols + (b < cols ? b : cols - 1)]
                 : g[(a < grows ? a : grows - 1) * cols + (b < cols ? b : cols - 1)];
  }
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    const int idx = threadIdx.x + j * NT;
    if (idx < total) {
      const int a = TRANS ? idx / ROWS : idx / K, b = TRANS ? idx - a * ROWS : idx - a * K;
      float x = v[j];
      if (b >= cols || a >= grows) x = 0.f;
      const int ir = TRANS ? b : a, ic = TRANS ? a : b;
      const unsigned u1 = pk_bf16(x, 0.f) & 0xffffu;
      const float r1 = x - __uint_as_float(u1 << 16);
      const unsigned u2 = pk_bf16(r1, 0.f) & 0xffffu;
      const float r2 = r1 - __uint_as_float(u2 << 16);
      const unsigned u3 = pk_bf16(r2, 0.f) & 0xffffu;
      wl[ir * ld + ic] = (short)u1;
      wl[plane + ir * ld + ic] = (short)u2;
      wl[2 * plane + ir * ld + ic] = (short)u3;
    }
  }
}

// =====================================================================================================
// forward of one layer
// =====================================================================================================
template <int KPAD, bool POOL, int NW>
__global__ __launch_bounds__(NW * 64) void k_w64_layer_fwd(const float* __restrict__ x, int F, const float* __restrict__ W,
                                                           const float* __restrict__ bias, const int64_t* __restrict__ ei,
                                                           int64_t E, const int32_t* __restrict__ graph_ptr,
                                                           const int32_t* __restrict__ edge_ptr, int B, float slope,
                                                           int apply_act, float* __restrict__ out, float* __restrict__ emb,
                                                           int32_t* __restrict__ status) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const WLds L = w_carve(smem + (size_t)wave * w_wave_bytes(1), 1);
  short* wl = reinterpret_cast<short*>(smem + (size_t)NW * w_wave_bytes(1));
  const int r = lane & 31, h = lane >> 5, q = lane & 15, r4 = lane >> 4;
  const int stride = gridDim.x * NW;
  int g = blockIdx.x * NW + wave;

  WGraph gi;
  WEdges er;
  if (g < B) {
    gi = w_graph(g, B, graph_ptr, edge_ptr, lane, status);
    er.load(gi, ei, E, lane);
  }
  const float4 bq = *reinterpret_cast<const float4*>(bias + 4 * q);
  w_stage_weight<false, NW * 64, DD, KPAD>(wl, W, DD, F);
  const float slope_eff = apply_act ? slope : 1.0f;
  __syncthreads();

  const bool wide = F == KPAD && ((uintptr_t)x % 16 == 0);   // whole float4 rows: the first rows of a graph are requested a graph ahead
  WRowsAhead<KPAD> ahead;
  if (wide && g < B) ahead.load(x, F, gi.nld, gi.n, 0, lane);

  for (; g < B; g += stride) {
    const WGraph gc = w_uniform(gi);
    const int rows = gc.nblk * 32;
    if (wide) {
      ahead.write(L.t0, gc.n, rows, 0, lane);
      for (int base = WRowsAhead<KPAD>::ROWS; base < rows; base += WRowsAhead<KPAD>::ROWS) {
        WRowsAhead<KPAD> more;
        more.load(x, F, gc.nld, gc.n, base, lane);
        more.write(L.t0, gc.n, rows, base, lane);
      }
    } else {
      w_stage_rows<KPAD>(L.t0, x, F, gc.nbase, gc.n, rows, lane);
    }
    w_build_csr<false>(L, gc, er, lane, status);
    const bool have_next = g + stride < B;
    if (have_next) {                                     // the NEXT graph's scalars and edges: in flight for the whole graph
      gi = w_graph(g + stride, B, graph_ptr, edge_ptr, lane, status);
      er.load(gi, ei, E, lane);
    }

    // ---- H' = dinv (.) (X W^T), in place, one 32-row block at a time
    for (int mb = 0; mb < gc.nblk; ++mb) {
      float* blk = L.t0 + mb * 32 * HS;
      f32x16 acc0, acc1;
#pragma unroll
      for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }
      tile_gemm_split<KPAD>(blk, wl, acc0, acc1, lane);
      mfma_results_fence(acc0, acc1);
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int row = krow(i, h);
        const float dv = L.dinv[mb * 32 + row];
        blk[row * HS + r] = acc0[i] * dv;
        blk[row * HS + 32 + r] = acc1[i] * dv;
      }
    }
    wave_sync();
    if (wide && have_next) ahead.load(x, F, gi.nld, gi.n, 0, lane);   // lands while this graph is aggregated and stored

    // ---- Y_i = H'_i + sum_k H'_{col k};  out = LeakyReLU(dinv_i Y_i + b).  16 lanes x float4 per row, 4 rows per pass.
    float4 pmax = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY), psum = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int pass0 = 0; pass0 < rows / 4; pass0 += 4) {   // four independent rows per lane slot at a time (rows is 32 or 64)
      int kb[4], ke[4];
      float4 acc[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int row = (pass0 + u) * 4 + r4;
        kb[u] = L.rowptr[row];
        ke[u] = row < gc.n ? L.rowptr[row + 1] : kb[u];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) acc[u] = w_row_sum_head(L.t0, L.col, (pass0 + u) * 4 + r4, kb[u], ke[u], q);
      if (__any(ke[0] - kb[0] > 4 || ke[1] - kb[1] > 4 || ke[2] - kb[2] > 4 || ke[3] - kb[3] > 4)) {
#pragma unroll
        for (int u = 0; u < 4; ++u) w_row_sum_tail(acc[u], L.t0, L.col, kb[u], ke[u], q);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int row = (pass0 + u) * 4 + r4;
        const bool valid = row < gc.n;
        const float di = L.dinv[row];
        float4 y = make_float4(fmaf(di, acc[u].x, bq.x), fmaf(di, acc[u].y, bq.y), fmaf(di, acc[u].z, bq.z), fmaf(di, acc[u].w, bq.w));
        y.x = fmaxf(y.x, slope_eff * y.x); y.y = fmaxf(y.y, slope_eff * y.y);
        y.z = fmaxf(y.z, slope_eff * y.z); y.w = fmaxf(y.w, slope_eff * y.w);
        if (valid) {
          *reinterpret_cast<float4*>(out + (size_t)(gc.nbase + row) * DD + 4 * q) = y;
          if (POOL) {
            pmax = make_float4(fmaxf(pmax.x, y.x), fmaxf(pmax.y, y.y), fmaxf(pmax.z, y.z), fmaxf(pmax.w, y.w));
            psum.x += y.x; psum.y += y.y; psum.z += y.z; psum.w += y.w;
          }
        }
      }
    }
    if (POOL) {   // this lane's row slots -> the four row groups of the wave (xor 16, 32): fixed order
      pmax = make_float4(fmaxf(pmax.x, __shfl_xor(pmax.x, 16, 64)), fmaxf(pmax.y, __shfl_xor(pmax.y, 16, 64)),
                         fmaxf(pmax.z, __shfl_xor(pmax.z, 16, 64)), fmaxf(pmax.w, __shfl_xor(pmax.w, 16, 64)));
      pmax = make_float4(fmaxf(pmax.x, __shfl_xor(pmax.x, 32, 64)), fmaxf(pmax.y, __shfl_xor(pmax.y, 32, 64)),
                         fmaxf(pmax.z, __shfl_xor(pmax.z, 32, 64)), fmaxf(pmax.w, __shfl_xor(pmax.w, 32, 64)));
      psum.x += __shfl_xor(psum.x, 16, 64); psum.y += __shfl_xor(psum.y, 16, 64); psum.z += __shfl_xor(psum.z, 16, 64); psum.w += __shfl_xor(psum.w, 16, 64);
      psum.x += __shfl_xor(psum.x, 32, 64); psum.y += __shfl_xor(psum.y, 32, 64); psum.z += __shfl_xor(psum.z, 32, 64); psum.w += __shfl_xor(psum.w, 32, 64);
      if (r4 == 0) {
        const float cntf = (float)(gc.n > 0 ? gc.n : 1);
        if (gc.n <= 0) pmax = make_float4(0.f, 0.f, 0.f, 0.f);
        *reinterpret_cast<float4*>(emb + (size_t)g * 2 * DD + 4 * q) = pmax;
        *reinterpret_cast<float4*>(emb + (size_t)g * 2 * DD + DD + 4 * q) =
            make_float4(psum.x / cntf, psum.y / cntf, psum.z / cntf, psum.w / cntf);
      }
    }
    wave_sync();   // the tile and the CSR are free for the next graph
  }
}

// =====================================================================================================
// backward of one layer (same contract as k_mid_layer_bwd, D = 64):
//   dY = dA (.) leaky'(A)            dA = dout, or (POOLG) the pooled-gradient expansion (ties of the max split evenly)
//   db += colsum dY ;  dH = Ahat^T dY ;  dW += dH^T x ;  dx = dH W  (NEEDS_DX)
// ONE LDS tile per wave (as many waves per CU as the forward): dY' -> tile; the transpose segmented sum of EVERY row is
// taken into registers before the first dH row is written back over the tile; x never goes through LDS -- the dW
// contraction reads it as the B operand straight from global memory (per k-step 8 coalesced 128-byte row segments per
// lane-half), and so does the premask of dx.
// =====================================================================================================
template <int KPAD, bool NEEDS_DX, bool POOLG, int NW>
__global__ __launch_bounds__(NW * 64) void k_w64_layer_bwd(
    const float* __restrict__ dout, const float* __restrict__ demb, const float* __restrict__ emb,
    const float* __restrict__ a_out, const float* __restrict__ x, int F, const float* __restrict__ W,
    const int64_t* __restrict__ ei, int64_t E, const int32_t* __restrict__ graph_ptr, const int32_t* __restrict__ edge_ptr,
    int B, float slope, int apply_act, float* __restrict__ dx, float* __restrict__ partials, int32_t* __restrict__ status) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const WLds L = w_carve(smem + (size_t)wave * w_wave_bytes(1), 1);
  short* wtl = reinterpret_cast<short*>(smem + (size_t)NW * w_wave_bytes(1));
  const int r = lane & 31, h = lane >> 5, q = lane & 15, r4 = lane >> 4;
  constexpr int NBF = KPAD / 32;
  constexpr int ld = DD + WPAD, plane = KPAD * ld;
  const bool act_here = apply_act & 1, premask = NEEDS_DX && (apply_act & 2);
  const bool need_a = POOLG || act_here;
  const int stride = gridDim.x * NW;
  int g = blockIdx.x * NW + wave;

  WGraph gi;
  WEdges er;
  if (g < B) {
    gi = w_graph(g, B, graph_ptr, edge_ptr, lane, status);
    er.load(gi, ei, E, lane);
  }
  if (NEEDS_DX) {
    w_stage_weight<true, NW * 64, KPAD, DD>(wtl, W, DD, F);     // image row f, column d <- W[d][f]
    __syncthreads();
  }

  f32x16 dw[2][NBF];   // dW[d-block][f-block], accumulated over every graph of this wave
#pragma unroll
  for (int mb = 0; mb < 2; ++mb)
#pragma unroll
    for (int nb = 0; nb < NBF; ++nb)
#pragma unroll
      for (int i = 0; i < 16; ++i) dw[mb][nb][i] = 0.f;
  float4 dbacc = make_float4(0.f, 0.f, 0.f, 0.f);
  const int fcol0 = r < F ? r : F - 1, fcol1 = 32 + r < F ? 32 + r : F - 1;   // this lane's x columns (clamped: dW columns >= F are dropped)

  for (; g < B; g += stride) {
    const WGraph gc = w_uniform(gi);
    const int rows = gc.nblk * 32;
    const int nlast = gc.n > 0 ? gc.n - 1 : 0;
    // this graph's rows of the layer output / upstream gradient are requested FIRST: they land while the CSR is built
    float4 gmx = make_float4(0.f, 0.f, 0.f, 0.f), share = gmx, dmean = gmx, dmx = gmx;
    if (POOLG) {
      const size_t eb = (size_t)g * 2 * DD + 4 * q;
      gmx = *reinterpret_cast<const float4*>(emb + eb);
      dmx = *reinterpret_cast<const float4*>(demb + eb);
      dmean = *reinterpret_cast<const float4*>(demb + eb + DD);
    }
    float4 av[16];                                       // this lane slot's rows (<= 64 rows / 4): layer output, or (no
#pragma unroll                                           // activation derivative needed) the upstream gradient itself
    for (int j = 0; j < 16; ++j) av[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    {
      const float* src = (need_a ? a_out : dout) + (size_t)gc.nld * DD + 4 * q;
#pragma unroll
      for (int j = 0; j < 8; ++j)                        // rows 0..31: unconditional, clamped (min, not a branch)
        av[j] = *reinterpret_cast<const float4*>(src + (size_t)min(r4 + 4 * j, nlast) * DD);
      if (rows > 32) {                                   // ONE scalar branch around the second row block's loads
#pragma unroll
        for (int j = 8; j < 16; ++j)
          av[j] = *reinterpret_cast<const float4*>(src + (size_t)min(r4 + 4 * j, nlast) * DD);
      }
    }
    w_build_csr<true>(L, gc, er, lane, status);
    if (g + stride < B) {
      gi = w_graph(g + stride, B, graph_ptr, edge_ptr, lane, status);
      er.load(gi, ei, E, lane);
    }

    // ---- 1. dY' = dinv (.) dA (.) leaky'(A) -> tile (rows >= n zero).  Lane slot: row group r4 (rows r4 + 4 j), columns 4q..
    if (POOLG) {
      const float cntf = (float)(gc.n > 0 ? gc.n : 1);
      dmean = make_float4(dmean.x / cntf, dmean.y / cntf, dmean.z / cntf, dmean.w / cntf);
    }
    if (POOLG) {
      float4 ties = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        if (r4 + 4 * j < gc.n) {
          const float4 a = av[j];
          ties.x += (a.x == gmx.x); ties.y += (a.y == gmx.y); ties.z += (a.z == gmx.z); ties.w += (a.w == gmx.w);
        }
      }
      ties = make_float4(ties.x + __shfl_xor(ties.x, 16, 64), ties.y + __shfl_xor(ties.y, 16, 64),
                         ties.z + __shfl_xor(ties.z, 16, 64), ties.w + __shfl_xor(ties.w, 16, 64));
      ties = make_float4(ties.x + __shfl_xor(ties.x, 32, 64), ties.y + __shfl_xor(ties.y, 32, 64),
                         ties.z + __shfl_xor(ties.z, 32, 64), ties.w + __shfl_xor(ties.w, 32, 64));
      share = make_float4(dmx.x / fmaxf(ties.x, 1.f), dmx.y / fmaxf(ties.y, 1.f), dmx.z / fmaxf(ties.z, 1.f), dmx.w / fmaxf(ties.w, 1.f));
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int row = r4 + 4 * j;
      if (4 * j < rows) {                                // wave-uniform
        float4 d = make_float4(0.f, 0.f, 0.f, 0.f);
        if (!POOLG) {
          d = av[j];                                     // (no activation derivative: av IS the upstream gradient)
          if (need_a) d = *reinterpret_cast<const float4*>(dout + (size_t)(gc.nld + min(row, nlast)) * DD + 4 * q);
        }
        if (row < gc.n) {
          const float4 a = av[j];
          if (POOLG)
            d = make_float4(dmean.x + (a.x == gmx.x ? share.x : 0.f), dmean.y + (a.y == gmx.y ? share.y : 0.f),
                            dmean.z + (a.z == gmx.z ? share.z : 0.f), dmean.w + (a.w == gmx.w ? share.w : 0.f));
          if (act_here) {
            d.x *= hcg_leaky_grad(a.x, slope); d.y *= hcg_leaky_grad(a.y, slope);
            d.z *= hcg_leaky_grad(a.z, slope); d.w *= hcg_leaky_grad(a.w, slope);
          }
          dbacc.x += d.x; dbacc.y += d.y; dbacc.z += d.z; dbacc.w += d.w;
          const float di = L.dinv[row];
          d = make_float4(di * d.x, di * d.y, di * d.z, di * d.w);
        } else {
          d = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        *reinterpret_cast<float4*>(L.t0 + row * HS + 4 * q) = d;
      }
    }
    wave_sync();

    // x operand of the dW contraction, k-step 0: requested here, a whole segmented-sum phase ahead of its use
    const float* xg = x + (size_t)gc.nld * F;
    float bv0[8], bv1[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int node = 8 * h + j;
      const float* xr = xg + (size_t)(node < gc.n ? node : nlast) * F;
      bv0[j] = xr[fcol0];
      bv1[j] = NBF > 1 ? xr[fcol1] : 0.f;
    }

    // ---- 2. dH_j = dinv_j (dY'_j + sum_{k in row j of the transpose} dY'_{col k}): every row into registers (av is dead),
    //         then back over the tile
#pragma unroll
    for (int j = 0; j < 16; ++j) {                       // (one row per lane slot at a time: interleaving two or four rows'
      if (4 * j < rows) {                                //  chains here spills 30-145 registers next to the dW accumulators)
        const int row = r4 + 4 * j;
        const int kb = L.rowptr[row], ke = row < gc.n ? L.rowptr[row + 1] : kb;
        float4 acc = w_row_sum_head(L.t0, L.col, row, kb, ke, q);
        if (__any(ke - kb > 4)) w_row_sum_tail(acc, L.t0, L.col, kb, ke, q);
        const float di = L.dinv[row];
        av[j] = make_float4(di * acc.x, di * acc.y, di * acc.z, di * acc.w);
      }
    }
    wave_sync();
#pragma unroll
    for (int j = 0; j < 16; ++j)
      if (4 * j < rows) *reinterpret_cast<float4*>(L.t0 + (r4 + 4 * j) * HS + 4 * q) = av[j];
    wave_sync();

    // ---- 3. dW[mb][nb] += dH^T x over the graph's nodes (K = nodes, 16 per step): A read down the columns of the dH tile,
    //         B = x straight from global memory (rows >= n: any finite value, their dH rows are zero)
    for (int ks = 0; ks < gc.nblk * 2; ++ks) {
      const Split3 B0 = split3(bv0), B1 = split3(bv1);
      if (ks + 1 < gc.nblk * 2) {                        // next k-step's x rows: in flight during this k-step's MFMAs
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int node = 16 * (ks + 1) + 8 * h + j;
          const float* xr = xg + (size_t)(node < gc.n ? node : nlast) * F;
          bv0[j] = xr[fcol0];
          bv1[j] = NBF > 1 ? xr[fcol1] : 0.f;
        }
      }
      Split3 A[2];
#pragma unroll
      for (int mb = 0; mb < 2; ++mb) {
        float avv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) avv[j] = L.t0[(16 * ks + 8 * h + j) * HS + mb * 32 + r];
        A[mb] = split3(avv);
      }
      mfma_split(dw[0][0], A[0], B0.p1, B0.p2, B0.p3);
      mfma_split(dw[1][0], A[1], B0.p1, B0.p2, B0.p3);
      if (NBF > 1) {
        mfma_split(dw[0][NBF - 1], A[0], B1.p1, B1.p2, B1.p3);
        mfma_split(dw[1][NBF - 1], A[1], B1.p1, B1.p2, B1.p3);
      }
    }

    // ---- 4. dx = dH W, one 32-row block at a time (premask: times leaky'(x), x read where the result is stored)
    if (NEEDS_DX) {
      for (int mb = 0; mb < gc.nblk; ++mb) {
        const float* blk = L.t0 + mb * 32 * HS;
        f32x16 dxa[NBF];
#pragma unroll
        for (int nb = 0; nb < NBF; ++nb)
#pragma unroll
          for (int i = 0; i < 16; ++i) dxa[nb][i] = 0.f;
#pragma unroll
        for (int s = 0; s < DD / 16; ++s) {
          const float4 a0 = *reinterpret_cast<const float4*>(blk + r * HS + 16 * s + 8 * h);
          const float4 a1 = *reinterpret_cast<const float4*>(blk + r * HS + 16 * s + 8 * h + 4);
          const float xa[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
          const Split3 A = split3(xa);
#pragma unroll
          for (int nb = 0; nb < NBF; ++nb) {
            const short* w0 = wtl + (nb * 32 + r) * ld + 16 * s + 8 * h;
            mfma_split(dxa[nb], A, *reinterpret_cast<const bf16x8*>(w0), *reinterpret_cast<const bf16x8*>(w0 + plane),
                       *reinterpret_cast<const bf16x8*>(w0 + 2 * plane));
          }
        }
        // premask: times leaky'(x).  The x values of a 32-column block are requested TOGETHER, unconditionally (clamped),
        // behind one kernel-uniform branch: a load inside the per-lane `row < n` guard is a serialised memory round trip
#pragma unroll
        for (int nb = 0; nb < NBF; ++nb) mfma_results_fence(dxa[nb]);
#pragma unroll
        for (int nb = 0; nb < NBF; ++nb) {
          const int f = nb * 32 + r, fc = nb == 0 ? fcol0 : fcol1;
          if (premask) {
            float xm[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) xm[i] = xg[(size_t)min(mb * 32 + krow(i, h), nlast) * F + fc];
#pragma unroll
            for (int i = 0; i < 16; ++i) dxa[nb][i] *= hcg_leaky_grad(xm[i], slope);
          }
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const int row = mb * 32 + krow(i, h);
            if (row < gc.n && f < F) dx[(size_t)(gc.nbase + row) * F + f] = dxa[nb][i];
          }
        }
      }
    }
    wave_sync();   // the tile / the CSR are free for the next graph
  }

  // ---- combine the waves of this workgroup in a fixed order and publish one slab: dW [64][KPAD] | db [64]
  constexpr int SLABF = DD * KPAD + DD;
  static_assert(w_wave_bytes(1) >= SLABF * sizeof(float), "a wave's slab must fit in its own tile");
#pragma unroll
  for (int mb = 0; mb < 2; ++mb)
#pragma unroll
    for (int nb = 0; nb < NBF; ++nb) mfma_results_fence(dw[mb][nb]);
  dbacc.x += __shfl_xor(dbacc.x, 16, 64); dbacc.y += __shfl_xor(dbacc.y, 16, 64); dbacc.z += __shfl_xor(dbacc.z, 16, 64); dbacc.w += __shfl_xor(dbacc.w, 16, 64);
  dbacc.x += __shfl_xor(dbacc.x, 32, 64); dbacc.y += __shfl_xor(dbacc.y, 32, 64); dbacc.z += __shfl_xor(dbacc.z, 32, 64); dbacc.w += __shfl_xor(dbacc.w, 32, 64);
  __syncthreads();                                     // every wave is done with its tile
  {
    float* mine = reinterpret_cast<float*>(smem + (size_t)wave * w_wave_bytes(1));
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
      for (int nb = 0; nb < NBF; ++nb)
#pragma unroll
        for (int i = 0; i < 16; ++i) mine[(mb * 32 + krow(i, h)) * KPAD + nb * 32 + r] = dw[mb][nb][i];
    if (r4 == 0) *reinterpret_cast<float4*>(mine + DD * KPAD + 4 * q) = dbacc;
  }
  __syncthreads();
  float* slab = partials + (size_t)blockIdx.x * SLABF;
  for (int idx = threadIdx.x; idx < SLABF; idx += NW * 64) {
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) s += reinterpret_cast<const float*>(smem + (size_t)w * w_wave_bytes(1))[idx];
    slab[idx] = s;
  }
}

int w_cus() {
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) == hipSuccess) {
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
  }
  return cus;
}

int w_grid(int64_t B, int nw) {
  int64_t grid = (B + nw - 1) / nw;
  const int cus = w_cus();
  if (grid > cus) grid = cus;
  return grid < 1 ? 1 : (int)grid;
}

template <auto KFN>
hipError_t w_allow_big_lds() {
  static hipError_t st = hipFuncSetAttribute((const void*)KFN, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  return st;
}

constexpr int NW_FWD = 7, NW_BWD = 8, NW_BWD_DX = 7;   // waves per workgroup: 160 KB of LDS / (one tile per wave + the weight image)

}  // namespace

// ---- internal interface used by the hcg_mid_* entry points (mid.hip) ----------------------------------------
int hcg_w64_applicable(int64_t F, int64_t D, int64_t max_nodes, int64_t max_edges) {
  return (D == DD && F >= 1 && F <= 64 && max_nodes >= 1 && max_nodes <= WN && max_edges >= 0 && max_edges <= WE) ? 1 : 0;
}

// one grid for both backward variants of a step keeps the slab count a function of the batch only
int hcg_w64_bwd_grid(int64_t B) { return w_grid(B, NW_BWD_DX); }

int hcg_w64_fwd_launch(const float* x, const float* W, const float* b, const int64_t* edge_index, int64_t E,
                       const int32_t* graph_ptr, const int32_t* edge_ptr, int64_t B, int64_t F, float slope, int apply_act,
                       float* out, float* emb, int32_t* status, hipStream_t stream) {
  const size_t lds = (size_t)NW_FWD * w_wave_bytes(1) + (size_t)3 * DD * ((F <= 32 ? 32 : 64) + WPAD) * 2;
  const dim3 grid(w_grid(B, NW_FWD)), blk(NW_FWD * 64);
#define LAUNCH_W_FWD(KP, PL)                                                                                       \
  do {                                                                                                             \
    hipError_t e = w_allow_big_lds<k_w64_layer_fwd<KP, PL, NW_FWD>>();                                             \
    if (e != hipSuccess) return hcg_hip_err(e);                                                                    \
    hipLaunchKernelGGL((k_w64_layer_fwd<KP, PL, NW_FWD>), grid, blk, lds, stream, x, (int)F, W, b, edge_index, E,  \
                       graph_ptr, edge_ptr, (int)B, slope, apply_act, out, emb, status);                           \
  } while (0)
  if (F <= 32) { if (emb) LAUNCH_W_FWD(32, true); else LAUNCH_W_FWD(32, false); }
  else         { if (emb) LAUNCH_W_FWD(64, true); else LAUNCH_W_FWD(64, false); }
#undef LAUNCH_W_FWD
  HCG_CHECK_LAUNCH();
  return HCG_OK;
}

int hcg_w64_bwd_launch(const float* dout, const float* demb, const float* emb, const float* out, const float* x,
                       const float* W, const int64_t* edge_index, int64_t E, const int32_t* graph_ptr,
                       const int32_t* edge_ptr, int64_t B, int64_t F, float slope, int apply_act, float* dx,
                       float* partials, int32_t* status, hipStream_t stream) {
  const bool poolg = dout == nullptr, ndx = dx != nullptr;
  const int kpad = F <= 32 ? 32 : 64;
  const int gsz = hcg_w64_bwd_grid(B);
  const dim3 grid(gsz);
#define LAUNCH_W_BWD(KP, DX, PG, NWV)                                                                                   \
  do {                                                                                                                  \
    const size_t lds = (size_t)NWV * w_wave_bytes(1) + (DX ? (size_t)3 * KP * (DD + WPAD) * 2 : 0);                     \
    hipError_t e = w_allow_big_lds<k_w64_layer_bwd<KP, DX, PG, NWV>>();                                                 \
    if (e != hipSuccess) return hcg_hip_err(e);                                                                         \
    hipLaunchKernelGGL((k_w64_layer_bwd<KP, DX, PG, NWV>), grid, dim3(NWV * 64), lds, stream, dout, demb, emb, out, x,  \
                       (int)F, W, edge_index, E, graph_ptr, edge_ptr, (int)B, slope, apply_act, dx, partials, status);  \
  } while (0)
#define DISPATCH_W_BWD(KP)                                                                                        \
  do {                                                                                                            \
    if (ndx) { if (poolg) LAUNCH_W_BWD(KP, true, true, NW_BWD_DX); else LAUNCH_W_BWD(KP, true, false, NW_BWD_DX); } \
    else     { if (poolg) LAUNCH_W_BWD(KP, false, true, NW_BWD); else LAUNCH_W_BWD(KP, false, false, NW_BWD); }   \
  } while (0)
  if (kpad == 32) DISPATCH_W_BWD(32); else DISPATCH_W_BWD(64);
#undef DISPATCH_W_BWD
#undef LAUNCH_W_BWD
  HCG_CHECK_LAUNCH();
  return HCG_OK;
}
