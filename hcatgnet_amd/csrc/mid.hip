// Fused per-layer kernels for batches of MID-SIZE graphs: one graph (33 .. 192 nodes) per workgroup.
//
// This is the reference's real regime: hcatgnet's reaction graphs have 56-184 atoms (3 molecules + explicit H),
// F = 25 / 32 input features, embedding_dim 64 ("Real data").  The small-graph kernels (fused.hip)
// stop at 32 nodes per tile; the any-shape kernels (layer.hip) round-trip every intermediate through HBM.  Here a
// workgroup of 8 waves owns a whole graph:
//   * gcn_norm on chip: the graph's raw COO edges -> in-degree + a CSR in LDS (integer atomics for the counting
//     sort, then every row sorted by source id: the per-node sum order is fixed -> bitwise reproducible);
//   * H' = dinv . (X W^T): the graph's X rows are staged into ONE LDS tile, each wave transforms its 32-row blocks
//     on the matrix cores (split-bf16 MFMAs, split_mfma.h) and writes H' back over its own rows;
//   * Y_i = H'_i + sum_{k in row i} H'_{col k}: wavefront segmented sum out of LDS (16 lanes x float4 per row,
//     four rows per pass) -- the neighbour gather never touches HBM;
//   * out = LeakyReLU(dinv . Y + b) stored once (row-contiguous 256 B), optional [max, mean] pooling epilogue.
// backward mirrors it (dY' tile -> transpose segmented sum -> dH tile -> dW on the matrix cores with K = nodes,
// dX = dH W), per-workgroup gradient slabs reduced in a fixed order by hcg_step_tail.
// LDS is sized at launch from the batch's largest graph (dynamic shared memory): 2 workgroups per CU up to 96 nodes.
#include "common.h"
#include "split_mfma.h"

namespace {

constexpr int MW = 8;                 // waves per workgroup (two per SIMD: the per-graph phases are latency chains)
constexpr int MT = MW * 64;           // threads
constexpr int MID_MAX_NODES = 224;    // 7 row blocks of 32 (two fp32 tiles + one weight image still fit 160 KB of LDS)
constexpr int MID_MAX_EDGES = 1024;
                                       // directed edges of one graph (2 per thread kept in registers; LDS col array, 16-bit ids)

#ifdef HCG_MID_STAMP        // tools/probe_mid.hip: s_memtime stamps of the per-graph phases of the first workgroup's first graphs
__device__ unsigned long long g_mid_stamp[MW][4][16];
#define MSTAMP(i)                                                                                           \
  do {                                                                                                      \
    __builtin_amdgcn_sched_barrier(0);                                                                      \
    unsigned long long _t;                                                                                  \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(_t)::"memory");                               \
    __builtin_amdgcn_sched_barrier(0);                                                                      \
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0 && mstamp_it < 4) g_mid_stamp[threadIdx.x >> 6][mstamp_it][(i)] = _t; \
  } while (0)
#else
#define MSTAMP(i) do { } while (0)
#endif

constexpr int NSLOT = 4;              // neighbour slots per row of the forward's slot table (rows of higher in-degree: CSR route)

struct MidLds {   // carved out of dynamic shared memory by carve()
  float* t0;              // [npad][HS]   forward: X -> H' (+ one all-zero row behind it);   backward: dY' -> X
  float* t1;              // [npad][HS]   backward only: dH
  short* wl;              // 3 planes of the pre-split weight image
  int* rowptr;            // [npad + 1]
  int* cursor;            // [npad]   degree counter (the fill pass counts it back down to zero)
  float* dinv;            // [npad]
  unsigned short* col;    // [emax]
  unsigned short* nbr;    // [npad][NSLOT]   forward: the first NSLOT sources of every target, 0xffff = empty
  int* flag;              // [4]      forward: [0] = some row of this graph overflowed its slots
  float* red;             // [MW * 2 * DD]   pooling / bias-gradient combine
};

__host__ __device__ inline size_t mid_lds_bytes(int npad, int emax, int wl_rows, int wl_k, bool two_tiles) {
  size_t b = (size_t)npad * HS * 4 * (two_tiles ? 2 : 1);
  b += (size_t)HS * 4;                         // the all-zero row behind the tile(s)
  b += (size_t)3 * wl_rows * (wl_k + WPAD) * 2;
  b = (b + 15) / 16 * 16;
  b += (size_t)(npad + 1 + 3) / 4 * 16;       // rowptr
  b += (size_t)(npad + 3) / 4 * 16 * 2;       // cursor, dinv
  b += (size_t)(emax + 7) / 8 * 16;           // col (u16)
  b += (size_t)npad * NSLOT * 2 + 16;         // nbr, flag
  b += (size_t)MW * 2 * DD * 4;
  return b + 64;
}

// (offsets are rounded as INTEGERS: a pointer -> integer -> pointer round trip makes the compiler forget that the carved
//  arrays are LDS -- every access to them became a flat load, whose s_waitcnt vmcnt(0) also waits for the prefetched HBM loads)
__device__ __forceinline__ MidLds carve(char* base, int npad, int emax, int wl_rows, int wl_k, bool two_tiles) {
  MidLds L;
  unsigned off = 0;
  L.t0 = reinterpret_cast<float*>(base);
  off += (unsigned)npad * HS * 4;
  L.t1 = two_tiles ? reinterpret_cast<float*>(base + off) : nullptr;
  off += (two_tiles ? (unsigned)npad * HS * 4 : 0u) + (unsigned)HS * 4;   // + the zero row: row npad (one tile) / 2 npad (two) of t0
  L.wl = reinterpret_cast<short*>(base + off);
  off += 3u * wl_rows * (wl_k + WPAD) * 2;
  off = (off + 15u) / 16u * 16u;
  L.rowptr = reinterpret_cast<int*>(base + off);
  off += (unsigned)(npad + 1 + 3) / 4 * 16;
  L.cursor = reinterpret_cast<int*>(base + off);
  off += (unsigned)(npad + 3) / 4 * 16;
  L.dinv = reinterpret_cast<float*>(base + off);
  off += (unsigned)(npad + 3) / 4 * 16;
  L.col = reinterpret_cast<unsigned short*>(base + off);
  off += (unsigned)(emax + 7) / 8 * 16;
  L.nbr = reinterpret_cast<unsigned short*>(base + off);
  off += (unsigned)npad * NSLOT * 2;
  L.flag = reinterpret_cast<int*>(base + off);
  off += 16;
  L.red = reinterpret_cast<float*>(base + off);
  return L;
}

struct GraphInfo { int nbase, n, ebase, ne, nblk; };

// host metadata that does not fit the kernel's tile: the graph is refused (n = ne = 0) and reported.  The selects stay
// OUTSIDE the branch of the reporting thread: assigned inside it, n and ne became per-lane registers and every address
// derived from them a 64-bit vector computation.
__device__ __forceinline__ void graph_validate(GraphInfo& gi, int npad, int emax, int32_t* status) {
  const bool bad = gi.n < 0 || gi.n > npad || gi.ne < 0 || gi.ne > emax;
  gi.n = __builtin_amdgcn_readfirstlane(bad ? 0 : gi.n);
  gi.ne = __builtin_amdgcn_readfirstlane(bad ? 0 : gi.ne);
  gi.nblk = (gi.n + 31) / 32;
  if (bad && threadIdx.x == 0) atomicOr(status, HCG_STATUS_SHAPE_LIMIT);
}

__device__ __forceinline__ GraphInfo graph_info(int g, const int32_t* __restrict__ graph_ptr, const int32_t* __restrict__ edge_ptr,
                                                int npad, int emax, int32_t* status) {
  GraphInfo gi;
  gi.nbase = graph_ptr[g];
  gi.n = graph_ptr[g + 1] - gi.nbase;
  gi.ebase = edge_ptr[g];
  gi.ne = edge_ptr[g + 1] - gi.ebase;
  graph_validate(gi, npad, emax, status);
  return gi;
}

// The four scalars of graph g, requested TWO graphs ahead.  A workgroup-uniform index would compile to s_load (counted on
// lgkmcnt, which every LDS read waits on: ~1 000 cycles exposed per graph in front of the edge prefetch, measured); a
// per-lane index makes them ONE vector load per lane (lanes 4k + 0..3: graph_ptr[g], graph_ptr[g + 1], edge_ptr[g],
// edge_ptr[g + 1]), counted on vmcnt, long complete when graph_finish reads them a whole graph later.
__device__ __forceinline__ int graph_raw(int g, const int32_t* __restrict__ graph_ptr, const int32_t* __restrict__ edge_ptr) {
  const int l = threadIdx.x & 3;
  const int32_t* p = (l & 2) ? edge_ptr : graph_ptr;
  return p[g + (l & 1)];
}
__device__ __forceinline__ GraphInfo graph_finish(int raw, int npad, int emax, int32_t* status) {
  GraphInfo gi;
  gi.nbase = __builtin_amdgcn_readlane(raw, 0);
  gi.n = __builtin_amdgcn_readlane(raw, 1) - gi.nbase;
  gi.ebase = __builtin_amdgcn_readlane(raw, 2);
  gi.ne = __builtin_amdgcn_readlane(raw, 3) - gi.ebase;
  graph_validate(gi, npad, emax, status);
  return gi;
}

// A graph's raw edges, two per thread, requested one graph AHEAD (loads only: unconditional, clamped index -- E >= 1 and
// `ei` readable are guaranteed by the host wrappers) and consumed by build_csr of the next iteration.
constexpr int EPT = MID_MAX_EDGES / MT;
struct EdgeRegs {
  long long s[EPT], d[EPT];
  // (workgroup-uniform bases + one unsigned 32-bit byte offset per slot = the scalar-base form of global_load; the clamps of
  //  the graph's edge range are scalar work.  The kernels are bound by VALU issue: per-slot 64-bit index arithmetic counts.)
  __device__ __forceinline__ void load(const GraphInfo& gi, const int64_t* __restrict__ ei, int64_t E) {
    long long eb = gi.ebase;
    eb = eb < 0 ? 0 : (eb > E - 1 ? E - 1 : eb);
    const long long room = E - eb;
    const int nec = (long long)gi.ne < room ? gi.ne : (int)room;
    const int last = nec > 0 ? nec - 1 : 0;
    const char* sb = reinterpret_cast<const char*>(ei + eb);
    const char* db = reinterpret_cast<const char*>(ei + E + eb);
#pragma unroll
    for (int j = 0; j < EPT; ++j) {
      const int e = threadIdx.x + j * MT;
      const unsigned off = 8u * (unsigned)(e < last ? e : last);
      s[j] = *reinterpret_cast<const long long*>(sb + off);
      d[j] = *reinterpret_cast<const long long*>(db + off);
    }
  }
};

// exclusive scan of the row sizes (cursor) into rowptr by ONE wave (<= 256 rows: 4 per lane); tid = lane of wave 0
__device__ __forceinline__ void csr_scan_rows(const MidLds& L, int nrows) {
  const int tid = threadIdx.x;
  constexpr int RPL = (MID_MAX_NODES + 63) / 64;
  int v[RPL], tot = 0;
#pragma unroll
  for (int j = 0; j < RPL; ++j) {
    const int i = tid * RPL + j;
    v[j] = i < nrows ? L.cursor[i] : 0;
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
    if (i < nrows) L.rowptr[i] = run;
    run += v[j];
  }
  if (tid == 63) L.rowptr[nrows] = incl;
}

// In-degree -> dinv, and a CSR of the graph in LDS.  BY_SRC = false: rows = targets, col = sources (forward
// aggregation); BY_SRC = true: rows = sources, col = targets (the transpose, for the backward).  dinv is always
// (1 + in-degree)^-1/2.  Explicit (i, i) edges collapse into the unit self loop (PyG add_remaining_self_loops).
// Every row ends up sorted by id, whatever order the LDS atomics ran in.  All MT threads; ends with a barrier.
// PRECONDITION: cursor[0 .. npad) (and degin_scratch) are ZERO -- csr_counters_clear() before the graph loop; the fill
// pass counts every row's cursor back down to zero, so the invariant holds from graph to graph with no clearing pass.
// Four barriers: count | scan (wave 0) beside dinv (the other waves) | fill (csr_count_scan_fill) | sort (csr_sort_rows).
__device__ __forceinline__ void csr_counters_clear(const MidLds& L, int npad, int* degin_scratch) {
  for (int i = threadIdx.x; i < npad; i += MT) { L.cursor[i] = 0; if (degin_scratch) degin_scratch[i] = 0; }
}

template <bool BY_SRC>
__device__ __forceinline__ void csr_count_scan_fill(const MidLds& L, const GraphInfo& gi, const EdgeRegs& er, int32_t* status,
                                                    int* degin_scratch, int mstamp_it = 4) {
  static_assert(MT - 64 >= MID_MAX_NODES, "one dinv row per thread beside the scan wave");
  const int tid = threadIdx.x;
  const int nrows = gi.nblk * 32;
  // this thread's edges (already in registers) -> local ids, kept from the counting pass to the fill pass
  unsigned short es[EPT], ed[EPT];
  bool bad = false;
#pragma unroll
  for (int j = 0; j < EPT; ++j) {
    const int e = tid + j * MT;
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
        if (BY_SRC) atomicAdd(&degin_scratch[dl], 1);
      }
    }
  }
  if (__ballot(bad) != 0ull && (tid & 63) == 0) atomicOr(status, HCG_STATUS_EDGE_UNGROUPED);   // edge leaves its graph: ignored
  __syncthreads();
  MSTAMP(2);
  if (tid < 64) {     // exclusive scan of the row sizes by wave 0 ...
    csr_scan_rows(L, nrows);
  } else if (tid - 64 < nrows) {     // ... while the other waves turn the in-degrees into dinv (MT - 64 >= MID_MAX_NODES rows)
    const int i = tid - 64;
    const int degin = BY_SRC ? degin_scratch[i] : L.cursor[i];
    L.dinv[i] = i < gi.n ? 1.0f / sqrtf(1.0f + (float)degin) : 0.f;
    if (BY_SRC) degin_scratch[i] = 0;
  }
  __syncthreads();
  MSTAMP(3);
#pragma unroll
  for (int j = 0; j < EPT; ++j) {
    if (es[j] != 0xffff) {
      const int rowi = BY_SRC ? es[j] : ed[j];
      const int left = atomicSub(&L.cursor[rowi], 1);          // counts the row back down to zero
      L.col[L.rowptr[rowi] + left - 1] = BY_SRC ? ed[j] : es[j];
    }
  }
  __syncthreads();
  MSTAMP(4);
}

// second half of the CSR build: every row sorted by id (fixed summation order).  Ends with a barrier.
__device__ __forceinline__ void csr_sort_rows(const MidLds& L, const GraphInfo& gi) {
  const int tid = threadIdx.x;
  for (int i = tid; i < gi.n; i += MT) {
    const int kb = L.rowptr[i], ke = L.rowptr[i + 1], len = ke - kb;
    if (len > 1 && len <= 4) {                      // (every row of a molecular graph) a register network: no dependent LDS chain
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

// acc = t[row] + sum_{k in [kb, ke)} t[col[k]] for this lane's (row, 4q..4q+3) slot: the first four neighbours' indices and
// rows are requested together (independent LDS reads instead of a chain of dependent ones), longer rows loop on
__device__ __forceinline__ float4 mid_row_sum(const float* t, const unsigned short* col, int row, int kb, int ke, int q) {
  float4 acc = *reinterpret_cast<const float4*>(t + row * HS + 4 * q);
  int c[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) c[j] = kb + j < ke ? col[kb + j] : row;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float4 v = *reinterpret_cast<const float4*>(t + c[j] * HS + 4 * q);
    if (kb + j < ke) { acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
  }
  for (int k = kb + 4; __any(k < ke); ++k) {
    if (k < ke) {
      const float4 v = *reinterpret_cast<const float4*>(t + col[k] * HS + 4 * q);
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
  }
  return acc;
}

// Stage columns [c0, c0 + KPAD) of rows [nbase, nbase + n) of a row-major [Nrows, F] matrix into t[row][0..KPAD) (zero
// padded past column F and to whole 32-row blocks).  All MT threads; no trailing barrier.  (Batching every load of the
// graph into registers first was measured: it costs the forward its second workgroup per CU -- 218 VGPRs -- and ran
// 1.7x slower.)
// BATCH: request every row of the graph before the first LDS write (one exposed HBM round trip instead of one per loop
// iteration) -- for the kernels whose register budget has room (they run one workgroup per CU anyway).
template <int KPAD, bool BATCH = false>
__device__ __forceinline__ void stage_graph_rows(float* t, const float* __restrict__ g, int F, int c0, int nbase, int n, int nblk) {
  const int tid = threadIdx.x;
  const int rows = nblk * 32;
  if (c0 + KPAD <= F && (F & 3) == 0 && ((uintptr_t)g % 16 == 0)) {
    constexpr int PER_ROW = KPAD / 4;
    if constexpr (BATCH) {
      constexpr int NIT = (MID_MAX_NODES * PER_ROW + MT - 1) / MT;
      float4 v[NIT];
#pragma unroll
      for (int j = 0; j < NIT; ++j) {
        const int idx = tid + j * MT, row = idx / PER_ROW, c4 = idx - row * PER_ROW;
        if (j * MT < rows * PER_ROW)           // block-uniform guard, clamped address: no per-lane branch around the load
          v[j] = *reinterpret_cast<const float4*>(g + (size_t)((n > 0 ? nbase : 0) + (row < n ? row : (n > 0 ? n - 1 : 0))) * F + c0 + 4 * c4);   // (an empty graph at the end of the batch has nbase == N)
      }
#pragma unroll
      for (int j = 0; j < NIT; ++j) {
        const int idx = tid + j * MT, row = idx / PER_ROW, c4 = idx - row * PER_ROW;
        if (j * MT < rows * PER_ROW && row < rows)
          *reinterpret_cast<float4*>(t + row * HS + 4 * c4) = row < n ? v[j] : make_float4(0.f, 0.f, 0.f, 0.f);
      }
      return;
    }
    for (int idx = tid; idx < rows * PER_ROW; idx += MT) {
      const int row = idx / PER_ROW, c4 = idx - row * PER_ROW;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (row < n) v = *reinterpret_cast<const float4*>(g + (size_t)(nbase + row) * F + c0 + 4 * c4);
      *reinterpret_cast<float4*>(t + row * HS + 4 * c4) = v;
    }
  } else {
    for (int idx = tid; idx < rows * KPAD; idx += MT) {
      const int row = idx / KPAD, c = idx - row * KPAD;
      t[row * HS + c] = (row < n && c0 + c < F) ? g[(size_t)(nbase + row) * F + c0 + c] : 0.f;
    }
  }
}

// A graph's x rows in registers, requested a whole graph AHEAD (loads only: unconditional, clamped addresses -- hipcc ends
// every guarded load with its own s_waitcnt vmcnt(0), which would turn the prefetch into an exposed HBM round trip) and
// written to the LDS tile at the start of the graph's own iteration.  Staging the rows inside the iteration cost 4 400 ..
// 5 900 of a graph's ~18 600 cycles (F = 25: eight dependent round trips), measured with tools/probe_mid.hip.
// Element (row, k) of the padded [NR][KPAD] tile per slot: VEC (F == KPAD, 16-byte aligned rows): one float4 per slot;
// otherwise one dword.  NR = the row capacity the kernel is compiled for (128 or MID_MAX_NODES): 8 .. 28 registers.
template <int KPAD, bool VEC, int NR>
struct XRows {
  static constexpr int PER_ROW = VEC ? KPAD / 4 : KPAD;        // slots per tile row
  static constexpr int RSTEP = MT / PER_ROW;                   // rows between a thread's consecutive slots
  static constexpr int NJ = (NR + RSTEP - 1) / RSTEP;          // slots per thread
  static_assert(MT % PER_ROW == 0, "a thread keeps its column over all its slots");
  float4 v4[VEC ? NJ : 1];
  float v1[VEC ? 1 : NJ];
  // slot j of thread t: row = t / PER_ROW + j * RSTEP, column (group) = t % PER_ROW -- the column is the thread's own, so the
  // per-slot work is one min and one 24-bit multiply-add (global offset) or a compile-time LDS offset (write)
  __device__ __forceinline__ void load(const float* __restrict__ g, int F, const GraphInfo& gi) {
    const int nlast = gi.n > 0 ? gi.n - 1 : 0;
    const char* base = reinterpret_cast<const char*>(g + (size_t)(gi.n > 0 ? gi.nbase : 0) * F);   // (an empty graph at the end of the batch has nbase == N)
    int t0 = threadIdx.x;
    asm volatile("" : "+v"(t0));      // (opaque: the slot indices are recomputed here, not kept live across the graph loop)
    const int row0 = t0 / PER_ROW, c = t0 % PER_ROW;
    // workgroup-uniform base + unsigned 32-bit BYTE offset = the scalar-base form of global_load (one offset register per
    // load; with a 64-bit address pair per slot the kernel lost its second workgroup per CU).  A graph's rows span < 16 MB.
    const unsigned F4 = 4u * (unsigned)F;
    const unsigned c4 = VEC ? 16u * (unsigned)c : 4u * (unsigned)(c < F ? c : F - 1);
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int row = row0 + j * RSTEP;
      const unsigned off = __umul24((unsigned)(row < nlast ? row : nlast), F4) + c4;
      if constexpr (VEC) v4[j] = *reinterpret_cast<const float4*>(base + off);
      else v1[j] = *reinterpret_cast<const float*>(base + off);
    }
  }
  __device__ __forceinline__ void write(float* t, int F, const GraphInfo& gi) const {
    const int nrows = gi.nblk * 32;
    int t0 = threadIdx.x;
    asm volatile("" : "+v"(t0));
    const int row0 = t0 / PER_ROW, c = t0 % PER_ROW;
    float* at = t + row0 * HS + (VEC ? 4 * c : c);
    const int nk = (VEC || c < F) ? gi.n : 0;          // rows below nk carry data in this thread's column, the rest are zero
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int row = row0 + j * RSTEP;
      // nrows is a multiple of 32: for RSTEP <= 32 a slot is inside the tile for EVERY thread or for none (scalar branch)
      const bool inside = RSTEP <= 32 ? j * RSTEP < nrows : row < nrows;
      if (inside) {
        if constexpr (VEC) *reinterpret_cast<float4*>(at + j * RSTEP * HS) = row < nk ? v4[j] : make_float4(0.f, 0.f, 0.f, 0.f);
        else at[j * RSTEP * HS] = row < nk ? v1[j] : 0.f;
      }
    }
  }
};


// The backward's x rows, requested while the transpose sum runs (its own iteration: the x tile shares t0 with dY', so the
// rows wait in registers) -- float4 slots for every width: F == KPAD and 16-byte aligned rows: one float4 load per slot,
// otherwise four dword loads (clamped columns, zeroed on the way into the tile).  Loads only, unconditional, on a
// workgroup-uniform base; the scalar staging loop it replaces for F = 25 was eight dependent HBM round trips per graph (5 of
// the layer-1 backward's 9.5 us at the reference's batch size 40).
template <int KPAD, int NR>
struct XRows4 {
  static constexpr int PER_ROW = KPAD / 4, RSTEP = MT / PER_ROW, NJ = (NR + RSTEP - 1) / RSTEP;
  float4 v[NJ];
  __device__ __forceinline__ void load(const float* __restrict__ g, int F, bool vec, const GraphInfo& gi) {
    const int nlast = gi.n > 0 ? gi.n - 1 : 0;
    const char* base = reinterpret_cast<const char*>(g + (size_t)(gi.n > 0 ? gi.nbase : 0) * F);
    int t0 = threadIdx.x;
    asm volatile("" : "+v"(t0));
    const int row0 = t0 / PER_ROW, c = t0 % PER_ROW;
    const unsigned F4 = 4u * (unsigned)F;
    if (vec) {
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int row = row0 + j * RSTEP;
        if (j * RSTEP < gi.nblk * 32)
          v[j] = *reinterpret_cast<const float4*>(base + __umul24((unsigned)(row < nlast ? row : nlast), F4) + 16u * (unsigned)c);
      }
    } else {
      unsigned co[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) co[i] = 4u * (unsigned)(4 * c + i < F ? 4 * c + i : F - 1);
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int row = row0 + j * RSTEP;
        if (j * RSTEP < gi.nblk * 32) {            // (workgroup-uniform: no per-lane branch around the loads)
          const char* rp = base + __umul24((unsigned)(row < nlast ? row : nlast), F4);
          v[j].x = *reinterpret_cast<const float*>(rp + co[0]);
          v[j].y = *reinterpret_cast<const float*>(rp + co[1]);
          v[j].z = *reinterpret_cast<const float*>(rp + co[2]);
          v[j].w = *reinterpret_cast<const float*>(rp + co[3]);
        }
      }
    }
  }
  __device__ __forceinline__ void write(float* t, int F, const GraphInfo& gi) const {
    const int nrows = gi.nblk * 32;
    int t0 = threadIdx.x;
    asm volatile("" : "+v"(t0));
    const int row0 = t0 / PER_ROW, c = t0 % PER_ROW;
    float* at = t + row0 * HS + 4 * c;
    const bool k0 = 4 * c < F, k1 = 4 * c + 1 < F, k2 = 4 * c + 2 < F, k3 = 4 * c + 3 < F;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int row = row0 + j * RSTEP;
      const bool inside = RSTEP <= 32 ? j * RSTEP < nrows : row < nrows;
      if (inside) {
        const bool live = row < gi.n;
        *reinterpret_cast<float4*>(at + j * RSTEP * HS) = make_float4(live && k0 ? v[j].x : 0.f, live && k1 ? v[j].y : 0.f,
                                                                      live && k2 ? v[j].z : 0.f, live && k3 ? v[j].w : 0.f);
      }
    }
  }
};

// Aggregation + epilogue of one unit of 16 rows by one wave: 16 lanes x float4 per row, 4 rows per pass (packed f32 math:
// v_pk_add / v_pk_fma / v_pk_mul -- these kernels are bound by VALU issue).  The four passes' index words and dinv are read
// up front in ONE LDS round trip; each pass then issues its five row reads together.
//   slot route: nbr[row] = up to NSLOT source ids, ascending (sorted by the lane that computed the row's dinv), empty =
//               `empty_id` (the zero row, larger than any id): the sum runs over ascending ids, bitwise run to run
//   CSR route : rows already sorted in col; rows longer than NSLOT continue in a per-lane loop
// Combine a value over the four 16-lane rows of a wave with gfx950's v_permlane16_swap / v_permlane32_swap (one VALU
// instruction per exchange; __shfl_xor compiles to ds_bpermute: 16 dependent LDS round trips per graph in the pooling tail).
// permlane16_swap(v, v) -> { rows (0, 0, 2, 2), rows (1, 1, 3, 3) };  permlane32_swap(v, v) -> { lower half twice, upper half twice }
// (the results are copied to scalars before they are reinterpreted: __builtin_bit_cast applied to the vector ELEMENT p[1]
//  reads element 0 with hipcc 7.2 -- it produced max(a, a) / a + a, measured as wrong pooled outputs)
__device__ __forceinline__ void rows_pair(float v, float& even, float& odd) {
  const auto p = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  const unsigned p0 = p[0], p1 = p[1];
  even = __uint_as_float(p0);
  odd = __uint_as_float(p1);
}
__device__ __forceinline__ void halves_pair(float v, float& lower, float& upper) {
  const auto p = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  const unsigned p0 = p[0], p1 = p[1];
  lower = __uint_as_float(p0);
  upper = __uint_as_float(p1);
}
__device__ __forceinline__ float rows_max(float v) {
  float a, b;
  rows_pair(v, a, b);
  v = fmaxf(a, b);
  halves_pair(v, a, b);
  return fmaxf(a, b);
}
__device__ __forceinline__ float rows_sum(float v) {        // (row 0 + row 1) + (row 2 + row 3), the same in every lane
  float a, b;
  rows_pair(v, a, b);
  v = a + b;
  halves_pair(v, a, b);
  return a + b;
}

struct Quad { f32x2 lo, hi; };
__device__ __forceinline__ Quad ldq(const char* p) {
  const float4 v = *reinterpret_cast<const float4*>(p);
  Quad q;
  q.lo = f32x2{v.x, v.y};
  q.hi = f32x2{v.z, v.w};
  return q;
}

// Z = Ahat x of a unit of 16 rows, from the x tile BEFORE the GEMM overwrites it (training form of the FIRST layer: with Z
// its weight gradient is the dense product G^T Z, G = dA (.) leaky'(A) -- no transpose sum, no dH round trip: csrc/tall.hip).
// KPAD / 4 lanes per row; a barrier in front makes every dinv and the sorted slots readable (first form: no barrier, every
// lane sorted its own copy of the slots and re-derived five dinv per row from the counters -- 1 / sqrtf is ~25 VALU
// instructions without fast-math: the extra aggregation cost 18 us of a 45 us launch, tools/ab_kernels.sh).
template <int KPAD, bool CSR>
__device__ __forceinline__ void mid_zagg_unit(const MidLds& L, int u, int n, int npad, unsigned empty_id, float* __restrict__ z_graph) {
  constexpr int LPRQ = KPAD / 4, RPP = 64 / LPRQ, NP = 16 / RPP;
  const int lane = threadIdx.x & 63, qq = lane % LPRQ, rr = lane / LPRQ;
  const char* tq = reinterpret_cast<const char*>(L.t0 + 4 * qq);
#pragma unroll
  for (int pass = 0; pass < NP; ++pass) {
    const int row = u * 16 + pass * RPP + rr;
    unsigned a0, a1, a2, a3;
    int kb = 0, ke = 0;
    float di;
    if constexpr (!CSR) {
      const uint2 nb = *reinterpret_cast<const uint2*>(L.nbr + row * NSLOT);       // (sorted by the row's dinv lane)
      a0 = nb.x & 0xffffu; a1 = nb.x >> 16; a2 = nb.y & 0xffffu; a3 = nb.y >> 16;
      di = L.dinv[row];
    } else {
      kb = L.rowptr[row];
      ke = L.rowptr[row + 1];
      a0 = kb + 0 < ke ? L.col[kb + 0] : empty_id;
      a1 = kb + 1 < ke ? L.col[kb + 1] : empty_id;
      a2 = kb + 2 < ke ? L.col[kb + 2] : empty_id;
      a3 = kb + 3 < ke ? L.col[kb + 3] : empty_id;
      di = L.dinv[row];
    }
    // an empty slot names the all-zero row; its factor only has to be finite: the last real row's
    const unsigned last = (unsigned)npad - 1u;
    auto dof = [&](unsigned a) -> float { return L.dinv[a < last ? a : last]; };
    const float d0 = dof(a0), d1 = dof(a1), d2 = dof(a2), d3 = dof(a3);
    Quad acc = ldq(tq + __umul24((unsigned)row, HS * 4u));
    const Quad n0 = ldq(tq + __umul24(a0, HS * 4u)), n1 = ldq(tq + __umul24(a1, HS * 4u));
    const Quad n2 = ldq(tq + __umul24(a2, HS * 4u)), n3 = ldq(tq + __umul24(a3, HS * 4u));
    acc.lo = f32x2{di, di} * acc.lo; acc.hi = f32x2{di, di} * acc.hi;
    acc.lo = __builtin_elementwise_fma(f32x2{d0, d0}, n0.lo, acc.lo); acc.hi = __builtin_elementwise_fma(f32x2{d0, d0}, n0.hi, acc.hi);
    acc.lo = __builtin_elementwise_fma(f32x2{d1, d1}, n1.lo, acc.lo); acc.hi = __builtin_elementwise_fma(f32x2{d1, d1}, n1.hi, acc.hi);
    acc.lo = __builtin_elementwise_fma(f32x2{d2, d2}, n2.lo, acc.lo); acc.hi = __builtin_elementwise_fma(f32x2{d2, d2}, n2.hi, acc.hi);
    acc.lo = __builtin_elementwise_fma(f32x2{d3, d3}, n3.lo, acc.lo); acc.hi = __builtin_elementwise_fma(f32x2{d3, d3}, n3.hi, acc.hi);
    if constexpr (CSR) {
      for (int k = kb + NSLOT; __any(k < ke); ++k) {
        if (k < ke) {
          const unsigned c = L.col[k];
          const float dc = L.dinv[c];
          const Quad v = ldq(tq + __umul24(c, HS * 4u));
          acc.lo = __builtin_elementwise_fma(f32x2{dc, dc}, v.lo, acc.lo); acc.hi = __builtin_elementwise_fma(f32x2{dc, dc}, v.hi, acc.hi);
        }
      }
    }
    if (row < n) {
      const f32x2 lo = f32x2{di, di} * acc.lo, hi = f32x2{di, di} * acc.hi;
      *reinterpret_cast<float4*>(z_graph + (size_t)row * KPAD + 4 * qq) = make_float4(lo.x, lo.y, hi.x, hi.y);
    }
  }
}

// TO_TILE (the backward's transpose sum): dH_row = dinv_row * sum goes to the second tile `t1` for EVERY row of the unit
// (rows past the graph are zero), no epilogue.
// BITS (needs POOL; the pooled layer of a training step whose backward is csrc/tall.hip's): the unit's outputs are NOT
// stored.  What the pooled backward needs of them -- is the value positive (LeakyReLU'), is it its graph's column maximum
// (where the max pool's gradient goes, ties included) -- leaves as one byte per (row, 4 columns) once the graph's maxima are
// known.  Until then a lane keeps two 32-bit masks for its <= 8 rows (nibble k = its k-th row, bit c = column 4 q + c):
// `sgn`, and `mxb` = the rows that attain the lane's OWN running maximum `pmax` (reset when a larger value arrives, joined on
// equality) -- a row is a maximum of the graph iff it is one of the lane's and the lane's maximum equals the graph's.
// SIGNS (training form of the FIRST layer, 64 columns): besides its output a row leaves four 16-bit pieces, piece j bit q =
// (column 4 q + j is positive) -- exactly what one v_cmp per register of the quad layout produces wave-wide -- for the
// dense first-layer backward of csrc/tall.hip, which then never reads this output.
template <bool CSR, bool POOL, bool TO_TILE = false, bool BITS = false, bool SIGNS = false>
__device__ __forceinline__ void mid_agg_unit(const MidLds& L, int u, int n, unsigned empty_id, const Quad& bq, float slope_eff,
                                             float* __restrict__ out_graph, int ldo, Quad& pmax, Quad& psum,
                                             unsigned* sgn = nullptr, unsigned* mxb = nullptr, int kbase = 0,
                                             unsigned short* __restrict__ sign_graph = nullptr) {
  const int lane = threadIdx.x & 63, q = lane & 15, r4 = lane >> 4;
  const char* tq = reinterpret_cast<const char*>(L.t0 + 4 * q);
  uint2 nb[4];
  float di[4];
  int kb[4], ke[4];
#pragma unroll
  for (int pass = 0; pass < 4; ++pass) {
    const int row = u * 16 + pass * 4 + r4;
    if constexpr (!CSR) {
      nb[pass] = *reinterpret_cast<const uint2*>(L.nbr + row * NSLOT);
    } else {
      kb[pass] = L.rowptr[row];
      ke[pass] = L.rowptr[row + 1];
    }
    di[pass] = L.dinv[row];
  }
  if (q == 0) {                                        // the rows' slots and counters are clean for the next graph
    const unsigned e2 = empty_id | (empty_id << 16);
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
      const int row = u * 16 + pass * 4 + r4;
      *reinterpret_cast<uint2*>(L.nbr + row * NSLOT) = make_uint2(e2, e2);
      L.cursor[row] = 0;
    }
  }
#pragma unroll
  for (int pass = 0; pass < 4; ++pass) {
    const int row = u * 16 + pass * 4 + r4;
    unsigned a0, a1, a2, a3;
    if constexpr (!CSR) {
      a0 = nb[pass].x & 0xffffu; a1 = nb[pass].x >> 16; a2 = nb[pass].y & 0xffffu; a3 = nb[pass].y >> 16;   // (sorted by the row's dinv lane)
    } else {
      a0 = kb[pass] + 0 < ke[pass] ? L.col[kb[pass] + 0] : empty_id;
      a1 = kb[pass] + 1 < ke[pass] ? L.col[kb[pass] + 1] : empty_id;
      a2 = kb[pass] + 2 < ke[pass] ? L.col[kb[pass] + 2] : empty_id;
      a3 = kb[pass] + 3 < ke[pass] ? L.col[kb[pass] + 3] : empty_id;
    }
    Quad acc = ldq(tq + __umul24((unsigned)row, HS * 4u));
    const Quad n0 = ldq(tq + __umul24(a0, HS * 4u)), n1 = ldq(tq + __umul24(a1, HS * 4u));
    const Quad n2 = ldq(tq + __umul24(a2, HS * 4u)), n3 = ldq(tq + __umul24(a3, HS * 4u));
    acc.lo += n0.lo; acc.hi += n0.hi;
    acc.lo += n1.lo; acc.hi += n1.hi;
    acc.lo += n2.lo; acc.hi += n2.hi;
    acc.lo += n3.lo; acc.hi += n3.hi;
    if constexpr (CSR) {
      for (int k = kb[pass] + NSLOT; __any(k < ke[pass]); ++k) {
        if (k < ke[pass]) {
          const Quad v = ldq(tq + __umul24((unsigned)L.col[k], HS * 4u));
          acc.lo += v.lo; acc.hi += v.hi;
        }
      }
    }
    const f32x2 d2 = f32x2{di[pass], di[pass]}, s2 = f32x2{slope_eff, slope_eff};
    if constexpr (TO_TILE) {
      const f32x2 lo = d2 * acc.lo, hi = d2 * acc.hi;
      *reinterpret_cast<float4*>(L.t1 + row * HS + 4 * q) = make_float4(lo.x, lo.y, hi.x, hi.y);
      continue;
    }
    Quad y;
    y.lo = __builtin_elementwise_fma(d2, acc.lo, bq.lo);
    y.hi = __builtin_elementwise_fma(d2, acc.hi, bq.hi);
    y.lo = __builtin_elementwise_max(y.lo, s2 * y.lo);   // LeakyReLU as max(v, slope v): exact for 0 <= slope <= 1; none: slope 1
    y.hi = __builtin_elementwise_max(y.hi, s2 * y.hi);
    if (row < n) {
      if constexpr (BITS) {
        const unsigned k4 = 4u * (unsigned)(kbase + pass);
        const float yv[4] = {y.lo.x, y.lo.y, y.hi.x, y.hi.y}, mv[4] = {pmax.lo.x, pmax.lo.y, pmax.hi.x, pmax.hi.y};
        unsigned sn = 0, m = *mxb;
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) {
          sn |= (unsigned)(yv[cc] > 0.f) << cc;
          const unsigned bit = 1u << (k4 + cc), colmask = 0x11111111u << cc;
          m = yv[cc] > mv[cc] ? ((m & ~colmask) | bit) : (yv[cc] == mv[cc] ? (m | bit) : m);
        }
        *sgn |= sn << k4;
        *mxb = m;
      } else {
        *reinterpret_cast<float4*>(out_graph + (size_t)row * ldo + 4 * q) = make_float4(y.lo.x, y.lo.y, y.hi.x, y.hi.y);
      }
      if constexpr (SIGNS) {
        const unsigned long long b0 = __builtin_amdgcn_ballot_w64(y.lo.x > 0.f), b1 = __builtin_amdgcn_ballot_w64(y.lo.y > 0.f);
        const unsigned long long b2 = __builtin_amdgcn_ballot_w64(y.hi.x > 0.f), b3 = __builtin_amdgcn_ballot_w64(y.hi.y > 0.f);
        if (q < 4) {
          const unsigned long long bsel = q == 0 ? b0 : (q == 1 ? b1 : (q == 2 ? b2 : b3));
          sign_graph[row * 4 + q] = (unsigned short)(bsel >> (16 * r4));
        }
      }
      if (POOL) {
        pmax.lo = __builtin_elementwise_max(pmax.lo, y.lo);
        pmax.hi = __builtin_elementwise_max(pmax.hi, y.hi);
        psum.lo += y.lo;
        psum.hi += y.hi;
      }
    }
  }
}

// =====================================================================================================
// forward of one layer, one graph per workgroup iteration
// =====================================================================================================
// One launch produces 64 output columns [coff, coff + 64) of a layer `ldo` columns wide (ldo = 64: the whole layer; ldo =
// 128: one of two independent column halves -- W / bias already point at the half's rows).  Inputs wider than 64
// features are contracted in K-chunks of 64 through the same LDS tile (MULTIK: accumulators stay in registers, a wave owns
// ONE 32-row block, the host guarantees nblk <= 8).
//
// A graph's iteration is a chain of dependent LDS / barrier round trips (tools/probe_mid.hip stamps them), so the kernel is
// built to keep that chain short -- round 3, 18 600 -> ~? cycles per graph on the reference-sized batch:
//   * neighbour SLOT TABLE instead of a CSR: one pass over the edges (slot = atomic in-degree counter; the first NSLOT
//     sources of a target go to nbr[target][slot]) replaces count + scan + fill + sort and three of their four barriers;
//     the slots' order depends on the atomics, so the lane that computes a row's dinv SORTS its four ids in registers
//     (summation stays "ascending neighbour id", bitwise run to run).  A graph with an in-degree above NSLOT (no molecule of the reference's data) takes
//     the CSR route for that graph -- the counters already hold the in-degrees.
//   * aggregation (mid_agg_unit): the index words of a unit's four passes are read in ONE round trip (the CSR went through
//     rowptr -> col -> rows once per pass), empty slots name an all-zero row (no branches), packed f32 math.
//   * the kernel is bound by VALU ISSUE, not by the length of the chain (two workgroups = four waves per SIMD; per graph and
//     CU ~8 x the per-wave VALU count in cycles): index arithmetic is 24-bit multiply-adds on workgroup-uniform bases,
//     LDS offsets are compile-time immediates, guards are scalar where the geometry allows it.
//   * the next graph's edges and x rows are requested a graph ahead, its four scalars two graphs ahead.
// VEC / NR: how the x rows are prefetched (XRows; unused by MULTIK, which stages chunk by chunk inside the iteration).
// (launch bounds: four waves per SIMD = two workgroups per CU = at most 128 VGPRs -- the scheduler trades load batching for
//  registers instead of silently dropping to one workgroup per CU; MULTIK needs more and runs one workgroup per CU)
// BITS (needs POOL): `out` is not written; `poolbits` [N][ldo / 4] bytes take its place for the pooled backward of
// csrc/tall.hip (mid_agg_unit): 91 MB less written here and 91 MB less read there on a 356 k-node batch.
// ZS (training form of the FIRST layer; not POOL, F <= 64, 64 output columns): two more outputs -- `zagg` [N][KPAD] = Ahat x
// (mid_zagg_unit) and `signs` [N][4] 16-bit sign pieces of the output (mid_agg_unit) -- for csrc/tall.hip's dense first-layer
// backward.
template <int KPAD, bool POOL, bool MULTIK, bool VEC, int NR, bool BITS = false, bool ZS = false>
__global__ __launch_bounds__(MT, MULTIK ? 2 : 4) void k_mid_layer_fwd(const float* __restrict__ x, int F, const float* __restrict__ W,
                                                         const float* __restrict__ bias, const int64_t* __restrict__ ei,
                                                         int64_t E, const int32_t* __restrict__ graph_ptr,
                                                         const int32_t* __restrict__ edge_ptr, int B, int npad, int emax,
                                                         float slope, int apply_act, float* __restrict__ out, int ldo, int coff,
                                                         float* __restrict__ emb, int32_t* __restrict__ status,
                                                         unsigned char* __restrict__ poolbits = nullptr,
                                                         float* __restrict__ zagg = nullptr,
                                                         unsigned short* __restrict__ signs = nullptr) {
  static_assert(!BITS || (POOL && !MULTIK), "the bit form belongs to the pooled layer (F <= 64)");
  static_assert(!ZS || (!POOL && !MULTIK), "Ahat x / sign pieces: the first (not pooled) layer, F <= 64");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int nkc = MULTIK ? (F + KPAD - 1) / KPAD : 1;  // K-chunks (MULTIK: F > 64; compiled apart, it costs registers)
  const MidLds L = carve(smem, npad, emax, DD * nkc, KPAD, false);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;

  // the FIRST graph's scalars, edges and x rows are requested before anything else: they are in flight while the weight
  // image is staged (at the reference's batch size 40 a launch is one graph long: serialised, weights -> scalars -> rows
  // were three exposed round trips in front of the first instruction of the graph)
  // (the grid never exceeds B: every workgroup has a first graph; past its last graph a workgroup requests the batch's
  //  last graph again and drops it -- the prefetches are unconditional)
  const int G = (int)gridDim.x;
  GraphInfo gi = graph_info(blockIdx.x, graph_ptr, edge_ptr, npad, emax, status);
  EdgeRegs er;
  XRows<KPAD, VEC, MULTIK ? 32 : NR> xr;
  er.load(gi, ei, E);
  if constexpr (!MULTIK) xr.load(x, F, gi);
  int raw_next = graph_raw(min((int)blockIdx.x + G, B - 1), graph_ptr, edge_ptr);

  // the weight image(s) are loop invariant: staged ONCE per workgroup, every K-chunk's image resident (MULTIK: two images,
  // 55 KB -- these kernels run one workgroup per CU anyway; re-staging a chunk's image per graph cost 8 global loads, 24
  // splits and 24 ds_write_b16 per thread and graph)
  constexpr int IMG = 3 * DD * (KPAD + WPAD);
  if (!MULTIK) {
    stage_weight_split<false, MT, DD, KPAD>(L.wl, W, DD, F);
  } else {
    for (int kc = 0; kc < nkc; ++kc) stage_weight_split<false, MT, DD, KPAD>(L.wl + kc * IMG, W, DD, F, kc * KPAD);
  }
  Quad bq;
  {
    const float4 b4 = *reinterpret_cast<const float4*>(bias + 4 * (lane & 15));
    bq.lo = f32x2{b4.x, b4.y};
    bq.hi = f32x2{b4.z, b4.w};
  }
  const float slope_eff = apply_act ? slope : 1.0f;    // LeakyReLU as max(v, slope v): exact for 0 <= slope <= 1; none: slope 1
  const unsigned empty_id = (unsigned)npad;            // an empty slot names the all-zero row behind the tile: no branches
  csr_counters_clear(L, npad, nullptr);
  for (int i = tid; i < npad * NSLOT / 2; i += MT) reinterpret_cast<unsigned*>(L.nbr)[i] = empty_id | (empty_id << 16);
  if (tid < HS) L.t0[npad * HS + tid] = 0.f;
  if (tid == 0) L.flag[0] = 0;
  __syncthreads();

  int mstamp_it = 0, rot = 0;
  (void)mstamp_it;
  for (int g = blockIdx.x; g < B; g += G, rot += 3) {
    MSTAMP(0);
    const GraphInfo gcur = gi;
    const int nrows = gcur.nblk * 32;
    // ---- the graph's edges (in registers since the previous graph) -> in-degree counters + slot table; its x rows -> tile
    unsigned short es[EPT], ed[EPT];
    {
      bool bad = false, over = false;
#pragma unroll
      for (int j = 0; j < EPT; ++j) {
        const int e = tid + j * MT;
        es[j] = 0xffff;
        ed[j] = 0xffff;
        if (e < gcur.ne) {
          const long long sv = er.s[j], dv = er.d[j];
          const unsigned sl = (unsigned)((int)sv - gcur.nbase), dl = (unsigned)((int)dv - gcur.nbase);
          const bool ok = sl < (unsigned)gcur.n && dl < (unsigned)gcur.n && (sv >> 31) == 0 && (dv >> 31) == 0;
          bad |= !ok;
          if (ok && sl != dl) {          // an explicit (i, i) edge collapses into the unit self loop (PyG add_remaining_self_loops)
            es[j] = (unsigned short)sl;
            ed[j] = (unsigned short)dl;
            const int slot = atomicAdd(&L.cursor[dl], 1);
            if (slot < NSLOT) L.nbr[dl * NSLOT + slot] = (unsigned short)sl;
            else over = true;
          }
        }
      }
      if (__ballot(bad) != 0ull && lane == 0) atomicOr(status, HCG_STATUS_EDGE_UNGROUPED);   // edge leaves its graph: ignored
      if (__ballot(over) != 0ull && lane == 0) L.flag[0] = 1;
    }
    if constexpr (MULTIK) stage_graph_rows<KPAD, true>(L.t0, x, F, 0, gcur.nbase, gcur.n, gcur.nblk);
    else xr.write(L.t0, F, gcur);
    MSTAMP(1);
    __syncthreads();
    MSTAMP(2);
    const bool csr_route = __builtin_amdgcn_readfirstlane(L.flag[0]) != 0;
    // dinv = (1 + in-degree)^-1/2 of a row block by the wave that owns the block in the GEMM below (its H' write reads them
    // back: same wave, LDS runs in order -- no barrier in between)
    // (row block -> wave rotates from graph to graph: the two or three GEMM waves of an 87-node graph would otherwise load
    //  the same SIMDs every time, in both workgroups of the CU)
    const int wblk = (wave + rot) & (MW - 1);
    if (wblk < gcur.nblk && lane < 32) {
      const int i = wblk * 32 + lane;
      L.dinv[i] = i < gcur.n ? 1.0f / sqrtf(1.0f + (float)L.cursor[i]) : 0.f;
      // the row's slots, filled in the order the atomics ran, sorted ONCE here by the row's own lane (in the aggregation the
      // 16 lanes of a row would each repeat the network: 10 VALU per pass): ascending ids, empty (= npad) last
      const uint2 nb = *reinterpret_cast<const uint2*>(L.nbr + i * NSLOT);
      unsigned a0 = nb.x & 0xffffu, a1 = nb.x >> 16, a2 = nb.y & 0xffffu, a3 = nb.y >> 16, t;
      t = min(a0, a1); a1 = max(a0, a1); a0 = t;
      t = min(a2, a3); a3 = max(a2, a3); a2 = t;
      t = min(a0, a2); a2 = max(a0, a2); a0 = t;
      t = min(a1, a3); a3 = max(a1, a3); a1 = t;
      t = min(a1, a2); a2 = max(a1, a2); a1 = t;
      *reinterpret_cast<uint2*>(L.nbr + i * NSLOT) = make_uint2(a0 | (a1 << 16), a2 | (a3 << 16));
    }
    if (csr_route) {        // some in-degree > NSLOT: CSR of the graph (the counters hold the row sizes), rows sorted by id
      if (tid < 64) csr_scan_rows(L, nrows);
      __syncthreads();      // (also: every dinv above has read its counter before the fill pass counts them down)
#pragma unroll
      for (int j = 0; j < EPT; ++j) {
        if (es[j] != 0xffff) {
          const int left = atomicSub(&L.cursor[ed[j]], 1);          // counts the row back down to zero
          L.col[L.rowptr[ed[j]] + left - 1] = es[j];
        }
      }
      __syncthreads();
      csr_sort_rows(L, gcur);
    }
    MSTAMP(3);
    if constexpr (ZS) {
      __syncthreads();                                   // every dinv and the sorted slots (the sorted CSR) are complete
      float* z_graph = zagg + (size_t)gcur.nbase * KPAD;
      for (int u = wave; u < gcur.nblk * 2; u += MW) {
        if (!csr_route) mid_zagg_unit<KPAD, false>(L, u, gcur.n, npad, empty_id, z_graph);
        else mid_zagg_unit<KPAD, true>(L, u, gcur.n, npad, empty_id, z_graph);
      }
      __syncthreads();                                   // every x row has been read: the GEMM may overwrite the tile
    }

    // ---- H' = dinv (.) (X W^T), in place: wave -> its own 32-row block, both column halves (eight waves on (row block,
    //      column half) blocks were measured: every A fragment is then split twice, and the kernel is bound by VALU issue)
    if (wblk < gcur.nblk) {       // (nblk <= 7; MULTIK: the host guarantees nblk <= 8)
      float* blk = L.t0 + wblk * 32 * HS;
      f32x16 acc0, acc1;
#pragma unroll
      for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }
      if constexpr (!MULTIK) {
        // transposed product: lane = tile row, register quads = 4 consecutive columns -> H' goes back as 8 x ds_write_b128
        // with ONE dinv per lane (row-per-register it was 32 x ds_write_b32 + 16 dinv reads: 1 500 cycles per graph)
        tile_gemm_split_t<KPAD>(blk, L.wl, acc0, acc1, lane);
        mfma_results_fence(acc0, acc1);
        MSTAMP(4);
        const float dv = L.dinv[wblk * 32 + r];
        float* hrow = blk + r * HS + 4 * h;
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          *reinterpret_cast<float4*>(hrow + 8 * g4) =
              make_float4(acc0[4 * g4] * dv, acc0[4 * g4 + 1] * dv, acc0[4 * g4 + 2] * dv, acc0[4 * g4 + 3] * dv);
          *reinterpret_cast<float4*>(hrow + 32 + 8 * g4) =
              make_float4(acc1[4 * g4] * dv, acc1[4 * g4 + 1] * dv, acc1[4 * g4 + 2] * dv, acc1[4 * g4 + 3] * dv);
        }
      }
    }
    if constexpr (MULTIK) {
      // inputs wider than 64 features: K-chunk by K-chunk through the same tile; the accumulators stay in registers
      const int mb = wblk;
      float* blk = L.t0 + mb * 32 * HS;
      f32x16 acc0, acc1;
#pragma unroll
      for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }
      for (int kc = 0; kc < nkc; ++kc) {
        if (kc > 0) {
          __syncthreads();                                           // every wave is done with the previous chunk
          stage_graph_rows<KPAD, true>(L.t0, x, F, kc * KPAD, gcur.nbase, gcur.n, gcur.nblk);
          __syncthreads();
        }
        if (mb < gcur.nblk) tile_gemm_split_t<KPAD>(blk, L.wl + kc * IMG, acc0, acc1, lane);
      }
      MSTAMP(4);
      __syncthreads();                                               // the last x chunk is dead: H' may overwrite it
      if (mb < gcur.nblk) {
        mfma_results_fence(acc0, acc1);
        const float dv = L.dinv[mb * 32 + r];
        float* hrow = blk + r * HS + 4 * h;
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          *reinterpret_cast<float4*>(hrow + 8 * g4) =
              make_float4(acc0[4 * g4] * dv, acc0[4 * g4 + 1] * dv, acc0[4 * g4 + 2] * dv, acc0[4 * g4 + 3] * dv);
          *reinterpret_cast<float4*>(hrow + 32 + 8 * g4) =
              make_float4(acc1[4 * g4] * dv, acc1[4 * g4 + 1] * dv, acc1[4 * g4 + 2] * dv, acc1[4 * g4 + 3] * dv);
        }
      }
    }
    MSTAMP(6);
    __syncthreads();
    MSTAMP(7);
    if (tid == 0) L.flag[0] = 0;                         // (every thread read it before the barrier above)

    // the NEXT graph's edges and x rows are requested here (its scalars were requested a graph ago): they land while this
    // graph is aggregated and stored; the scalars of the graph after next follow
    gi = graph_finish(raw_next, npad, emax, status);
    er.load(gi, ei, E);
    if constexpr (!MULTIK) xr.load(x, F, gi);
    raw_next = graph_raw(min(g + 2 * G, B - 1), graph_ptr, edge_ptr);
    MSTAMP(8);

    // ---- Y_i = H'_i + sum_k H'_{nbr k} (ascending ids);  out = LeakyReLU(dinv_i Y_i + b)
    Quad pmax, psum;
    pmax.lo = pmax.hi = f32x2{-INFINITY, -INFINITY};
    psum.lo = psum.hi = f32x2{0.f, 0.f};
    float* out_graph = out + (size_t)gcur.nbase * ldo + coff;
    unsigned sgn = 0, mxb = 0;                             // BITS: this lane's rows, see mid_agg_unit
    int kb4 = 0;
    for (int u = wave; u < gcur.nblk * 2; u += MW, kb4 += 4) {      // units of 16 rows (at most two per wave: nblk <= 7)
      unsigned short* sign_graph = ZS ? signs + (size_t)gcur.nbase * 4 : nullptr;
      if (!csr_route) mid_agg_unit<false, POOL, false, BITS, ZS>(L, u, gcur.n, empty_id, bq, slope_eff, out_graph, ldo, pmax, psum, &sgn, &mxb, kb4, sign_graph);
      else mid_agg_unit<true, POOL, false, BITS, ZS>(L, u, gcur.n, empty_id, bq, slope_eff, out_graph, ldo, pmax, psum, &sgn, &mxb, kb4, sign_graph);
    }
    MSTAMP(9);
    if (POOL) {   // rows of this lane's (r4, q) slot -> wave (the four 16-lane rows) -> workgroup (LDS, fixed order)
      const int q = lane & 15, r4 = lane >> 4;
      const float4 pm = make_float4(rows_max(pmax.lo.x), rows_max(pmax.lo.y), rows_max(pmax.hi.x), rows_max(pmax.hi.y));
      const float4 sm = make_float4(rows_sum(psum.lo.x), rows_sum(psum.lo.y), rows_sum(psum.hi.x), rows_sum(psum.hi.y));
      if (r4 == 0) {
        *reinterpret_cast<float4*>(L.red + wave * 2 * DD + 4 * q) = pm;
        *reinterpret_cast<float4*>(L.red + wave * 2 * DD + DD + 4 * q) = sm;
      }
      __syncthreads();
      if (tid < DD) {
        float m = -INFINITY, sum = 0.f;
#pragma unroll
        for (int w = 0; w < MW; ++w) {                    // fixed order over the waves
          m = fmaxf(m, L.red[w * 2 * DD + tid]);
          sum += L.red[w * 2 * DD + DD + tid];
        }
        if (gcur.n <= 0) m = 0.f;
        emb[(size_t)g * 2 * ldo + coff + tid] = m;                                   // [max | mean], each ldo wide
        emb[(size_t)g * 2 * ldo + ldo + coff + tid] = sum / (float)(gcur.n > 0 ? gcur.n : 1);
        if (BITS) L.red[tid] = m;      // (wave 0's slot of this column: this thread was its only reader)
      }
    }
    MSTAMP(10);
    __syncthreads();   // the tile, the slot table and the combine scratch are free for the next graph
    MSTAMP(11);
    if constexpr (BITS) {
      // the graph's column maxima are known: a row attains one iff it attains the lane's own maximum and that IS the graph's
      // (L.red is next written two barriers into the next graph)
      const int q = lane & 15, r4 = lane >> 4;
      const float4 gm = *reinterpret_cast<const float4*>(L.red + 4 * q);
      const unsigned keep = (pmax.lo.x == gm.x ? 0x11111111u : 0u) | (pmax.lo.y == gm.y ? 0x22222222u : 0u) |
                            (pmax.hi.x == gm.z ? 0x44444444u : 0u) | (pmax.hi.y == gm.w ? 0x88888888u : 0u);
      mxb &= keep;
      unsigned char* bits_graph = poolbits + (size_t)gcur.nbase * (ldo >> 2) + (coff >> 2) + q;
      unsigned k4 = 0;
      for (int u = wave; u < gcur.nblk * 2; u += MW) {
#pragma unroll
        for (int pass = 0; pass < 4; ++pass, k4 += 4) {
          const int row = u * 16 + pass * 4 + r4;
          if (row < gcur.n) bits_graph[(size_t)row * (ldo >> 2)] = (unsigned char)(((sgn >> k4) & 0xfu) | (((mxb >> k4) & 0xfu) << 4));
        }
      }
    }
#ifdef HCG_MID_STAMP
    ++mstamp_it;
#endif
  }
}

// =====================================================================================================
// backward of one layer, one graph per workgroup iteration
//   dY = dA (.) leaky'(A)            dA = dout, or (POOLG) the pooled-gradient expansion (ties of the max split evenly)
//   db += colsum dY ;  dH = Ahat^T dY ;  dW += dH^T x ;  dx = dH W  (NEEDS_DX)
// per-workgroup partial sums go to `partials[blockIdx][64*KPAD + 64]` (same slab layout as fused.hip).
// =====================================================================================================
// One launch handles 64 of the layer's `ldo` output columns, [coff, coff + 64) (ldo = 128: one of two column halves; W
// already points at the half's rows, dW / db of the half go to this launch's own slabs).  Inputs wider than 64 features
// are handled in f-chunks of 64 through the same tiles.  ACC_DX: add to dx instead of storing (second column half).
template <int KPAD, int NFC, bool NEEDS_DX, bool POOLG>
__global__ __launch_bounds__(MT, 2) void k_mid_layer_bwd(
    const float* __restrict__ dout, const float* __restrict__ demb, const float* __restrict__ emb,
    const float* __restrict__ a_out, int ldo, int coff, const float* __restrict__ x, int F, const float* __restrict__ W,
    const int64_t* __restrict__ ei, int64_t E, const int32_t* __restrict__ graph_ptr, const int32_t* __restrict__ edge_ptr,
    int B, int npad, int emax, float slope, int apply_act, float* __restrict__ dx, int acc_dx, float* __restrict__ partials,
    int32_t* __restrict__ status) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const MidLds L = carve(smem, npad, emax, NEEDS_DX ? KPAD : 0, DD, true);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5, q = lane & 15, r4 = lane >> 4;
  constexpr int NBF = KPAD / 32;
  constexpr int ld = DD + WPAD, plane = KPAD * ld;
  constexpr int FPAD = NFC * KPAD;                    // padded input width of the slab rows
  const int c4 = tid & 15, rg = tid >> 4;             // step 1: this thread's float4 column group / row group
  // apply_act bits as in hcg_fused_layer_bwd: bit 0 = multiply the upstream gradient by leaky'(a_out); bit 1 = hand dx
  // down already multiplied by leaky'(x) (the x chunk sits in t0 when dx is stored), so the layer below runs with bit 0
  // clear and never reads its own output
  const bool act_here = apply_act & 1, premask = NEEDS_DX && (apply_act & 2);
  const bool need_a = POOLG || act_here;

  // the first graph's scalars and edges are requested before the weight image is staged (one exposed round trip less at the
  // reference's batch size, where a launch is one graph long)
  GraphInfo gnext;
  EdgeRegs er;
  if ((int)blockIdx.x < B) {
    gnext = graph_info(blockIdx.x, graph_ptr, edge_ptr, npad, emax, status);
    er.load(gnext, ei, E);
  }
  if (NEEDS_DX && NFC == 1) stage_weight_split<true, MT, KPAD, DD>(L.wl, W, DD, F);   // image row f, column d <- W[d][f]
  const bool xvec = F == KPAD && ((uintptr_t)x % 16 == 0);
  const unsigned empty_id = 2u * (unsigned)npad;       // an empty slot names the all-zero row behind the two tiles
  csr_counters_clear(L, npad, reinterpret_cast<int*>(L.red));
  for (int i = tid; i < npad * NSLOT / 2; i += MT) reinterpret_cast<unsigned*>(L.nbr)[i] = empty_id | (empty_id << 16);
  if (tid < HS) L.t0[2 * npad * HS + tid] = 0.f;
  if (tid == 0) L.flag[0] = 0;
  __syncthreads();

  // dW: per f-chunk 2 x NBF output blocks (d-block mbw x f-block nbw), each shared by NPART waves that take every
  // NPART-th k-step
  constexpr int NBLOCKS = 2 * NBF, NPART = MW / NBLOCKS;
  const int blk_id = wave % NBLOCKS, part = wave / NBLOCKS;
  const int mbw = blk_id / NBF, nbw = blk_id % NBF;
  f32x16 dw[NFC];
#pragma unroll
  for (int fc = 0; fc < NFC; ++fc)
#pragma unroll
    for (int i = 0; i < 16; ++i) dw[fc][i] = 0.f;
  float4 dbacc = make_float4(0.f, 0.f, 0.f, 0.f);

  XRows4<KPAD, MID_MAX_NODES> xr;
  int mstamp_it = 0;
  (void)mstamp_it;
  for (int g = blockIdx.x; g < B; g += gridDim.x) {
    MSTAMP(0);
    const GraphInfo gi = gnext;
    const int rows = gi.nblk * 32;
    // ---- 1. dY' = dinv (.) dA (.) leaky'(A) -> t0 (rows >= n zero)
    //         this thread's rows (row group rg, float4 column group c4): every global load of the graph is requested HERE, in
    //         front of the CSR build (they depend on the graph's scalars only; behind the build they were an exposed round trip
    //         per graph), workgroup-uniform base + 32-bit byte offsets
    constexpr int NR = MID_MAX_NODES / (MT / 16);
    float4 av[NR] = {}, dv[NR];
    {
      const size_t gbase = (size_t)(gi.n > 0 ? gi.nbase : 0) * ldo + coff;
      const char* ab = reinterpret_cast<const char*>(a_out + gbase);
      const char* db = reinterpret_cast<const char*>(dout + gbase);
      const unsigned ld4 = 4u * (unsigned)ldo;
      const int nlast = gi.n > 0 ? gi.n - 1 : 0;
#pragma unroll
      for (int j = 0; j < NR; ++j) {
        const int row = rg + j * (MT / 16);
        if (j * (MT / 16) < rows) {                          // block-uniform
          const unsigned at = __umul24((unsigned)(row < nlast ? row : nlast), ld4) + 16u * (unsigned)c4;
          if (need_a) av[j] = *reinterpret_cast<const float4*>(ab + at);   // (kernel-uniform)
          if (!POOLG) dv[j] = *reinterpret_cast<const float4*>(db + at);
        }
      }
    }
    float4 gmx = make_float4(0.f, 0.f, 0.f, 0.f), share = gmx, dmean = gmx, dmx = gmx;
    if (POOLG) {
      const size_t eb = (size_t)g * 2 * ldo + coff + 4 * c4;       // [max | mean], each ldo wide
      gmx = *reinterpret_cast<const float4*>(emb + eb);
      dmx = *reinterpret_cast<const float4*>(demb + eb);
      dmean = *reinterpret_cast<const float4*>(demb + eb + ldo);
    }
    MSTAMP(1);
    // ---- the graph's edges -> out-degree counters + slot table of the TRANSPOSE (row = source, slots = its targets) and
    //      in-degree counters (dinv); one pass, as in the forward.  A graph with an out-degree above NSLOT takes the CSR route.
    int* degin = reinterpret_cast<int*>(L.red);
    unsigned short es[EPT], ed[EPT];
    {
      bool bad = false, over = false;
#pragma unroll
      for (int j = 0; j < EPT; ++j) {
        const int e = tid + j * MT;
        es[j] = 0xffff;
        ed[j] = 0xffff;
        if (e < gi.ne) {
          const long long sv = er.s[j], dv2 = er.d[j];
          const unsigned sl = (unsigned)((int)sv - gi.nbase), dl = (unsigned)((int)dv2 - gi.nbase);
          const bool ok = sl < (unsigned)gi.n && dl < (unsigned)gi.n && (sv >> 31) == 0 && (dv2 >> 31) == 0;
          bad |= !ok;
          if (ok && sl != dl) {
            es[j] = (unsigned short)sl;
            ed[j] = (unsigned short)dl;
            const int slot = atomicAdd(&L.cursor[sl], 1);
            if (slot < NSLOT) L.nbr[sl * NSLOT + slot] = (unsigned short)dl;
            else over = true;
            atomicAdd(&degin[dl], 1);
          }
        }
      }
      if (__ballot(bad) != 0ull && lane == 0) atomicOr(status, HCG_STATUS_EDGE_UNGROUPED);
      if (__ballot(over) != 0ull && lane == 0) L.flag[0] = 1;
    }
    // (POOLG) this thread's tie counts over its rows -> the wave's four row groups -> one partial per wave (the column maxima
    // and the rows were requested above: they have been in flight for the whole pass over the edges)
    float* tiesc = L.red + 256;                           // [MW][64] behind the in-degree counters
    if (POOLG) {
      float4 ties = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int j = 0; j < NR; ++j) {
        const int row = rg + j * (MT / 16);
        if (j * (MT / 16) < rows && row < gi.n) {
          const float4 a = av[j];
          ties.x += (a.x == gmx.x); ties.y += (a.y == gmx.y); ties.z += (a.z == gmx.z); ties.w += (a.w == gmx.w);
        }
      }
      ties = make_float4(rows_sum(ties.x), rows_sum(ties.y), rows_sum(ties.z), rows_sum(ties.w));   // (counts: exact in any order)
      if (r4 == 0) *reinterpret_cast<float4*>(tiesc + wave * DD + 4 * q) = ties;
    }
    __syncthreads();
    MSTAMP(2);
    const bool csr_route = __builtin_amdgcn_readfirstlane(L.flag[0]) != 0;
    // one thread per row (waves 1..: wave 0 is the scan wave of the CSR route): dinv, the counter back to zero, slots sorted
    if (tid >= 64 && tid - 64 < rows) {
      const int i = tid - 64;
      L.dinv[i] = i < gi.n ? 1.0f / sqrtf(1.0f + (float)degin[i]) : 0.f;
      degin[i] = 0;
      const uint2 nb = *reinterpret_cast<const uint2*>(L.nbr + i * NSLOT);
      unsigned a0 = nb.x & 0xffffu, a1 = nb.x >> 16, a2 = nb.y & 0xffffu, a3 = nb.y >> 16, t;
      t = min(a0, a1); a1 = max(a0, a1); a0 = t;
      t = min(a2, a3); a3 = max(a2, a3); a2 = t;
      t = min(a0, a2); a2 = max(a0, a2); a0 = t;
      t = min(a1, a3); a3 = max(a1, a3); a1 = t;
      t = min(a1, a2); a2 = max(a1, a2); a1 = t;
      *reinterpret_cast<uint2*>(L.nbr + i * NSLOT) = make_uint2(a0 | (a1 << 16), a2 | (a3 << 16));
    }
    if (csr_route) {
      if (tid < 64) csr_scan_rows(L, rows);
      __syncthreads();
#pragma unroll
      for (int j = 0; j < EPT; ++j) {
        if (es[j] != 0xffff) {
          const int left = atomicSub(&L.cursor[es[j]], 1);           // counts the row back down to zero
          L.col[L.rowptr[es[j]] + left - 1] = ed[j];
        }
      }
      __syncthreads();
      csr_sort_rows(L, gi);                                          // (ends with a barrier)
    } else {
      __syncthreads();                                               // dinv and the wave partials of the tie counts are visible
    }
    MSTAMP(3);
    if (tid == 0) L.flag[0] = 0;                           // (every thread read it before the barrier above)
    if (g + (int)gridDim.x < B) {                          // the NEXT graph's scalars and edges: in flight for the whole graph
      gnext = graph_info(g + gridDim.x, graph_ptr, edge_ptr, npad, emax, status);
      er.load(gnext, ei, E);
    }
    if (POOLG) {
      const float cntf = (float)(gi.n > 0 ? gi.n : 1);
      dmean = make_float4(dmean.x / cntf, dmean.y / cntf, dmean.z / cntf, dmean.w / cntf);
      float4 tot = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int w = 0; w < MW; ++w) {
        const float4 t = *reinterpret_cast<const float4*>(tiesc + w * DD + 4 * c4);
        tot.x += t.x; tot.y += t.y; tot.z += t.z; tot.w += t.w;
      }
      share = make_float4(dmx.x / fmaxf(tot.x, 1.f), dmx.y / fmaxf(tot.y, 1.f), dmx.z / fmaxf(tot.z, 1.f), dmx.w / fmaxf(tot.w, 1.f));
    }
    MSTAMP(4);
#pragma unroll
    for (int j = 0; j < NR; ++j) {
      const int row = rg + j * (MT / 16);
      if (j * (MT / 16) < rows && row < rows) {
        float4 d = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row < gi.n) {
          const float4 a = av[j];
          if (POOLG) {
            d = make_float4(dmean.x + (a.x == gmx.x ? share.x : 0.f), dmean.y + (a.y == gmx.y ? share.y : 0.f),
                            dmean.z + (a.z == gmx.z ? share.z : 0.f), dmean.w + (a.w == gmx.w ? share.w : 0.f));
          } else {
            d = dv[j];
          }
          if (act_here) {
            d.x *= hcg_leaky_grad(a.x, slope); d.y *= hcg_leaky_grad(a.y, slope);
            d.z *= hcg_leaky_grad(a.z, slope); d.w *= hcg_leaky_grad(a.w, slope);
          }
          dbacc.x += d.x; dbacc.y += d.y; dbacc.z += d.z; dbacc.w += d.w;
          const float di = L.dinv[row];
          d = make_float4(di * d.x, di * d.y, di * d.z, di * d.w);
        }
        *reinterpret_cast<float4*>(L.t0 + row * HS + 4 * c4) = d;
      }
    }
    __syncthreads();
    MSTAMP(5);
    if constexpr (NFC == 1) xr.load(x, F, xvec, gi);      // the x rows (step 3) land while the transpose sum runs

    // ---- 2. dH_j = dinv_j (dY'_j + sum_{k in row j of the transpose} dY'_{col k}) -> t1 (mid_agg_unit: slot table / CSR route;
    //         its lanes also reset the rows' slots and counters for the next graph)
    {
      Quad z0, z1, z2;
      z0.lo = z0.hi = z1.lo = z1.hi = z2.lo = z2.hi = f32x2{0.f, 0.f};
      for (int u = wave; u < gi.nblk * 2; u += MW) {      // units of 16 rows
        if (!csr_route) mid_agg_unit<false, false, true>(L, u, gi.n, empty_id, z0, 1.0f, nullptr, 0, z1, z2);
        else mid_agg_unit<true, false, true>(L, u, gi.n, empty_id, z0, 1.0f, nullptr, 0, z1, z2);
      }
    }
    MSTAMP(6);
    __syncthreads();
    MSTAMP(7);

#pragma unroll
    for (int fc = 0; fc < NFC; ++fc) {
      // ---- 3. x chunk -> t0 (and, chunked, the matching rows of the dx operand image)
      if constexpr (NFC == 1) xr.write(L.t0, F, gi);
      else stage_graph_rows<KPAD, true>(L.t0, x, F, fc * KPAD, gi.nbase, gi.n, gi.nblk);
      if (NEEDS_DX && NFC > 1) stage_weight_split<true, MT, KPAD, DD>(L.wl, W, DD, F, fc * KPAD);
      __syncthreads();
      MSTAMP(8);

      // ---- 4. dW[mbw][fc, nbw] += dH^T x over the graph's nodes (K = nodes, 16 per step); both operands read down columns
      for (int ks = part; ks < gi.nblk * 2; ks += NPART) {
        float avv[8], bvv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int node = 16 * ks + 8 * h + j;
          avv[j] = L.t1[node * HS + mbw * 32 + r];
          bvv[j] = L.t0[node * HS + nbw * 32 + r];
        }
        const Split3 A = split3(avv), Bx = split3(bvv);
        mfma_split(dw[fc], A, Bx.p1, Bx.p2, Bx.p3);
      }

      MSTAMP(9);
      // ---- 5. dx[:, chunk] (+)= dH W[:, chunk], each wave on its own row blocks.  TRANSPOSED product (the operand image is
      //         the MFMA's A operand, the dH rows its B operand): a lane holds one ROW of the block and a register quad four
      //         consecutive columns, so premask reads, read-back and stores are 128-bit per quad (row-per-register they were 32
      //         guarded dword stores with 64-bit addresses per block: 8 000 of the layer-2 backward's 25 000 cycles per graph)
      if (NEEDS_DX && NFC == 1) {
        const bool f4ok = (F & 3) == 0 && ((uintptr_t)dx % 16 == 0);
        const int nlast = gi.n > 0 ? gi.n - 1 : 0;
        // one 32 x 32 output block (row block mb, column block nb) per wave and round: all eight waves busy on a graph of up
        // to 128 nodes (one wave per ROW block left five of eight idle for 5 900 cycles; the dH fragment is then split once per
        // column block -- this kernel runs one workgroup per CU and has VALU slots to spare)
        for (int b8 = wave; b8 < gi.nblk * NBF; b8 += MW) {
          const int mb = b8 / NBF, nb = b8 % NBF;
          const float* blk = L.t1 + mb * 32 * HS;
          f32x16 v;
#pragma unroll
          for (int i = 0; i < 16; ++i) v[i] = 0.f;
#pragma unroll
          for (int s = 0; s < DD / 16; ++s) {
            const float4 a0 = *reinterpret_cast<const float4*>(blk + r * HS + 16 * s + 8 * h);
            const float4 a1 = *reinterpret_cast<const float4*>(blk + r * HS + 16 * s + 8 * h + 4);
            const float xa[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
            const Split3 X = split3(xa);
            const short* w0 = L.wl + (nb * 32 + r) * ld + 16 * s + 8 * h;
            const bf16x8 b1 = *reinterpret_cast<const bf16x8*>(w0), b2 = *reinterpret_cast<const bf16x8*>(w0 + plane),
                         b3 = *reinterpret_cast<const bf16x8*>(w0 + 2 * plane);
            v = HCG_MFMA(b1, X.p3, v);       // the six cross terms of mfma_split, operands swapped
            v = HCG_MFMA(b3, X.p1, v);
            v = HCG_MFMA(b2, X.p2, v);
            v = HCG_MFMA(b1, X.p2, v);
            v = HCG_MFMA(b2, X.p1, v);
            v = HCG_MFMA(b1, X.p1, v);
          }
          mfma_results_fence(v);
          const int row = mb * 32 + r;                      // this lane's row; quad g4: columns f0 .. f0 + 3
          const bool live = row < gi.n;
          float* drow = dx + (size_t)((gi.n > 0 ? gi.nbase : 0) + (row < nlast ? row : nlast)) * F;
          if (premask) {
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
              const float4 xv = *reinterpret_cast<const float4*>(L.t0 + row * HS + nb * 32 + 8 * g4 + 4 * h);
              v[4 * g4] *= hcg_leaky_grad(xv.x, slope); v[4 * g4 + 1] *= hcg_leaky_grad(xv.y, slope);
              v[4 * g4 + 2] *= hcg_leaky_grad(xv.z, slope); v[4 * g4 + 3] *= hcg_leaky_grad(xv.w, slope);
            }
          }
          // second column half of a 128-wide layer (acc_dx): the first half's dx is read back -- the block's four quads
          // requested TOGETHER, unconditionally (clamped row and columns), behind one kernel-uniform branch
          if (acc_dx) {
            float4 old[4];
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
              const int f0 = nb * 32 + 8 * g4 + 4 * h;
              if (f4ok) {
                old[g4] = *reinterpret_cast<const float4*>(drow + (f0 + 3 < F ? f0 : F - 4));
              } else {
                old[g4] = make_float4(drow[f0 < F ? f0 : F - 1], drow[f0 + 1 < F ? f0 + 1 : F - 1], drow[f0 + 2 < F ? f0 + 2 : F - 1],
                                      drow[f0 + 3 < F ? f0 + 3 : F - 1]);
              }
            }
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) { v[4 * g4] += old[g4].x; v[4 * g4 + 1] += old[g4].y; v[4 * g4 + 2] += old[g4].z; v[4 * g4 + 3] += old[g4].w; }
          }
#pragma unroll
          for (int g4 = 0; g4 < 4; ++g4) {
            const int f0 = nb * 32 + 8 * g4 + 4 * h;
            if (live) {
              if (f4ok) {
                if (f0 + 3 < F) *reinterpret_cast<float4*>(drow + f0) = make_float4(v[4 * g4], v[4 * g4 + 1], v[4 * g4 + 2], v[4 * g4 + 3]);
              } else {
                if (f0 < F) drow[f0] = v[4 * g4];
                if (f0 + 1 < F) drow[f0 + 1] = v[4 * g4 + 1];
                if (f0 + 2 < F) drow[f0 + 2] = v[4 * g4 + 2];
                if (f0 + 3 < F) drow[f0 + 3] = v[4 * g4 + 3];
              }
            }
          }
        }
      }
      if (NEEDS_DX && NFC > 1) {   // (wide inputs, f-chunked: row-per-register form -- the transposed one spills 150+ registers beside two dW chunks)
        for (int mb = wave; mb < gi.nblk; mb += MW) {
          const float* blk = L.t1 + mb * 32 * HS;
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
              const short* w0 = L.wl + (nb * 32 + r) * ld + 16 * s + 8 * h;
              mfma_split(dxa[nb], A, *reinterpret_cast<const bf16x8*>(w0), *reinterpret_cast<const bf16x8*>(w0 + plane),
                         *reinterpret_cast<const bf16x8*>(w0 + 2 * plane));
            }
          }
#pragma unroll
          for (int nb = 0; nb < NBF; ++nb) mfma_results_fence(dxa[nb]);
          // second column half of a 128-wide layer (acc_dx): the first half's dx is read back -- all 16 values of a
          // 32-column block requested TOGETHER, unconditionally (clamped), behind one kernel-uniform branch; inside the
          // per-lane guards below every one of them was a serialised memory round trip
          const int nlast = gi.n > 0 ? gi.n - 1 : 0;
#pragma unroll
          for (int nb = 0; nb < NBF; ++nb) {
            const int f = fc * KPAD + nb * 32 + r;
            const int fcl = f < F ? f : F - 1;
            if (premask) {
#pragma unroll
              for (int i = 0; i < 16; ++i)
                dxa[nb][i] *= hcg_leaky_grad(L.t0[(mb * 32 + krow(i, h)) * HS + nb * 32 + r], slope);
            }
            if (acc_dx) {                                  // eight at a time: the kernel is at its register limit
#pragma unroll
              for (int i0 = 0; i0 < 16; i0 += 8) {
                float old[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) old[i] = dx[(size_t)((gi.n > 0 ? gi.nbase : 0) + min(mb * 32 + krow(i0 + i, h), nlast)) * F + fcl];
#pragma unroll
                for (int i = 0; i < 8; ++i) dxa[nb][i0 + i] += old[i];
              }
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) {
              const int row = mb * 32 + krow(i, h);
              if (row < gi.n && f < F) dx[(size_t)(gi.nbase + row) * F + f] = dxa[nb][i];
            }
          }
        }
      }
      MSTAMP(10);
      __syncthreads();   // t0 / the operand image are free for the next chunk (or the next graph)
      MSTAMP(11);
    }
#ifdef HCG_MID_STAMP
    ++mstamp_it;
#endif
  }

  // ---- publish this workgroup's slab: dW [64][FPAD] | db [64]
  constexpr int SLABF = DD * FPAD + DD;
  float* slab = partials + (size_t)blockIdx.x * SLABF;
  float* comb = L.t0;                                    // [NBLOCKS][32 * 32]: the K parts of a block meet here, fixed order
  static_assert(NBLOCKS * 1024 * 4 <= 2 * 32 * HS * 4, "combine scratch must fit in the two smallest tiles");
#pragma unroll
  for (int fc = 0; fc < NFC; ++fc) {
    mfma_results_fence(dw[fc]);
    for (int round = 0; round < NPART; ++round) {
      if (part == round) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          float* c = comb + blk_id * 1024 + krow(i, h) * 32 + r;
          *c = round == 0 ? dw[fc][i] : *c + dw[fc][i];
        }
      }
      __syncthreads();
    }
    for (int idx = tid; idx < NBLOCKS * 1024; idx += MT) {
      const int b = idx >> 10, rr = (idx >> 5) & 31, cc = idx & 31;
      slab[((b / NBF) * 32 + rr) * FPAD + fc * KPAD + (b % NBF) * 32 + cc] = comb[idx];
    }
    __syncthreads();
  }
  float* sc = L.t0;                                      // [MT / 16 row groups][64]
  *reinterpret_cast<float4*>(sc + rg * DD + 4 * c4) = dbacc;
  __syncthreads();
  if (tid < DD) {
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < MT / 16; ++k) s += sc[k * DD + tid];
    slab[DD * FPAD + tid] = s;
  }
}

int mid_grid(int64_t B, int wgs_per_cu) {
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) == hipSuccess) {
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
  }
  int64_t grid = (int64_t)cus * wgs_per_cu;
  if (grid > B) grid = B;
  return grid < 1 ? 1 : (int)grid;
}

// workgroups of 8 waves that fit a CU: 160 KB of LDS, at most 2 (16 waves: the kernels use up to 256 VGPRs per lane)
int wgs_per_cu(size_t lds) { return lds * 2 <= 160 * 1024 ? 2 : 1; }

int pad32(int64_t v) { return (int)((v + 31) / 32 * 32); }
int pad8(int64_t v) { return (int)((v + 7) / 8 * 8 > 8 ? (v + 7) / 8 * 8 : 8); }

// dynamic LDS above 64 KB has to be allowed per kernel: once per process (not per launch: the step may be under capture)
template <auto KFN>
hipError_t allow_big_lds() {   // (the kernel is a template VALUE parameter: one static per instantiation, not per signature)
  static hipError_t st = hipFuncSetAttribute((const void*)KFN, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  return st;
}

}  // namespace

// 1 when the one-graph-per-workgroup kernels apply: D = 64 or 128 (two independent 64-column halves), F <= 128
// (contracted in chunks of 64), every graph of the batch within 224 nodes / 1024 directed edges
extern "C" int hcg_mid_supported(int64_t F, int64_t D, int64_t max_nodes_per_graph, int64_t max_edges_per_graph) {
  return ((D == DD || D == 2 * DD) && F >= 1 && F <= 2 * DD && max_nodes_per_graph >= 1 &&
          max_nodes_per_graph <= MID_MAX_NODES && max_edges_per_graph >= 0 && max_edges_per_graph <= MID_MAX_EDGES) ? 1 : 0;
}

extern "C" int hcg_mid_layer_fwd(const float* x, const float* W, const float* b, const int64_t* edge_index, int64_t E,
                                 const int32_t* graph_ptr, const int32_t* edge_ptr, int64_t N, int64_t B, int64_t F, int64_t D,
                                 int64_t max_nodes, int64_t max_edges, float slope, int apply_act, float* out, float* emb,
                                 uint8_t* poolbits, float* xagg, uint8_t* signbits, int32_t* status, hcg_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!hcg_mid_supported(F, D, max_nodes, max_edges)) return HCG_ERR_UNSUPPORTED;
  // Ahat x + sign pieces (training form of a first, not pooled, 64-wide layer over graphs of more than 64 nodes)
  if ((xagg == nullptr) != (signbits == nullptr)) return HCG_ERR_INVALID_ARG;
  if (xagg && (emb || poolbits || D != DD || F > 64 || !out || hcg_w64_applicable(F, D, max_nodes, max_edges))) return HCG_ERR_UNSUPPORTED;
  if (apply_act && !(slope >= 0.f && slope <= 1.f)) return HCG_ERR_UNSUPPORTED;   // LeakyReLU is evaluated as max(v, slope*v)
  if (N < 0 || B < 0 || E < 0) return HCG_ERR_INVALID_ARG;
  if (B == 0 || N == 0) return HCG_OK;
  if (!x || !W || !b || !graph_ptr || !edge_ptr || (!out && !poolbits) || !status || (E > 0 && !edge_index)) return HCG_ERR_INVALID_ARG;
  // the bit form (training, pooled layer: `out` is not written) exists for F <= 64 on graphs of more than 64 nodes
  if (poolbits && (!emb || F > 64 || hcg_w64_applicable(F, D, max_nodes, max_edges))) return HCG_ERR_UNSUPPORTED;
  if (E == 0) { edge_index = reinterpret_cast<const int64_t*>(graph_ptr); E = 1; }  // readable dummy; no graph has edges
  // graphs up to 64 nodes: one graph per WAVE (wave.hip).  (Round 3 measured this kernel's rebuilt forward on them instead:
  // equal at 4096 graphs of 37-63 atoms -- 49.5 / 47.2 against 49.5 / 48.0 us -- and 18.1 / 17.6 against 19.2 / 19.2 us on the
  // 1 270 graphs of 33-36 atoms of the ragged batch, whose launches are one graph chain long either way.)
  if (hcg_w64_applicable(F, D, max_nodes, max_edges))
    return hcg_w64_fwd_launch(x, W, b, edge_index, E, graph_ptr, edge_ptr, B, F, slope, apply_act, out, emb, status, stream);
  const int npad = pad32(max_nodes), emax = pad8(max_edges);
  const int kpad = F <= 32 ? 32 : 64;
  const int nimg = F > 64 ? (int)((F + 63) / 64) : 1;                 // K-chunk weight images kept resident (MULTIK)
  const size_t lds = mid_lds_bytes(npad, emax, DD * nimg, kpad, false);
  const dim3 grid(mid_grid(B, wgs_per_cu(lds))), blk(MT);
#define LAUNCH_MID_FWD(KP, PL, MK, VC, NRC, BT, ZSV)                                                                       \
  do {                                                                                                                     \
    auto kfn = k_mid_layer_fwd<KP, PL, MK, VC, NRC, BT, ZSV>;                                                              \
    hipError_t e = allow_big_lds<k_mid_layer_fwd<KP, PL, MK, VC, NRC, BT, ZSV>>();                                         \
    if (e != hipSuccess) return hcg_hip_err(e);                                                                            \
    hipLaunchKernelGGL(kfn, grid, blk, lds, stream, x, (int)F, Wh, bh, edge_index, E, graph_ptr, edge_ptr, (int)B, npad,    \
                       emax, slope, apply_act, out, (int)D, coff, emb, status, poolbits, xagg,                             \
                       reinterpret_cast<unsigned short*>(signbits));                                                       \
  } while (0)
#define LAUNCH_MID_FWD_V(KP, PL, NRC, BT, ZSV)                                                                             \
  do { if (vec) LAUNCH_MID_FWD(KP, PL, false, true, NRC, BT, ZSV); else LAUNCH_MID_FWD(KP, PL, false, false, NRC, BT, ZSV); } while (0)
#define LAUNCH_MID_FWD_X(KP, PL)                                                                                           \
  do {                                                                                                                     \
    if (PL && poolbits) { if (npad <= 128) LAUNCH_MID_FWD_V(KP, true, 128, true, false); else LAUNCH_MID_FWD_V(KP, true, MID_MAX_NODES, true, false); } \
    else if (!PL && xagg) { if (npad <= 128) LAUNCH_MID_FWD_V(KP, false, 128, false, true); else LAUNCH_MID_FWD_V(KP, false, MID_MAX_NODES, false, true); } \
    else { if (npad <= 128) LAUNCH_MID_FWD_V(KP, PL, 128, false, false); else LAUNCH_MID_FWD_V(KP, PL, MID_MAX_NODES, false, false); } \
  } while (0)
  const bool vec = F == kpad && ((uintptr_t)x % 16 == 0);     // whole float4 rows in the x prefetch
  for (int half = 0; half < (int)(D / DD); ++half) {     // 64 output columns per launch
    const float* Wh = W + (size_t)half * DD * F;
    const float* bh = b + half * DD;
    const int coff = half * DD;
    if (kpad == 32)   { if (emb) LAUNCH_MID_FWD_X(32, true); else LAUNCH_MID_FWD_X(32, false); }
    else if (F <= 64) { if (emb) LAUNCH_MID_FWD_X(64, true); else LAUNCH_MID_FWD_X(64, false); }
    else              { if (emb) LAUNCH_MID_FWD(64, true, true, false, 32, false, false); else LAUNCH_MID_FWD(64, false, true, false, 32, false, false); }
    HCG_CHECK_LAUNCH();
  }
#undef LAUNCH_MID_FWD_X
#undef LAUNCH_MID_FWD_V
#undef LAUNCH_MID_FWD
  return HCG_OK;
}

// slab geometry of one column half: dW [64][fpad] | db [64]
static int mid_fpad(int64_t F) { return F <= 32 ? 32 : (F <= 64 ? 64 : 128); }

static int mid_bwd_grid(int64_t B, int64_t F, int64_t max_nodes, int64_t max_edges, size_t* lds_out, bool needs_dx) {
  const int kpad = F <= 32 ? 32 : 64;
  const size_t lds = mid_lds_bytes(pad32(max_nodes), pad8(max_edges), needs_dx ? kpad : 0, DD, true);
  if (lds_out) *lds_out = lds;
  // one grid size for both variants of a step (with / without dx) keeps the slab count a function of the batch only
  const size_t lds_worst = mid_lds_bytes(pad32(max_nodes), pad8(max_edges), kpad, DD, true);
  return mid_grid(B, wgs_per_cu(lds_worst));
}

extern "C" size_t hcg_mid_workspace_bytes(int64_t B, int64_t F, int64_t D, int64_t max_nodes, int64_t max_edges) {
  if (!hcg_mid_supported(F, D, max_nodes, max_edges)) return 0;
  if (hcg_w64_applicable(F, D, max_nodes, max_edges))
    return (size_t)hcg_w64_bwd_grid(B) * (DD * mid_fpad(F) + DD) * sizeof(float) + 256;
  const size_t half = (size_t)mid_bwd_grid(B, F, max_nodes, max_edges, nullptr, true) * (DD * mid_fpad(F) + DD) * sizeof(float);
  return (size_t)(D / DD) * half + 256;
}

// backward, stage 1 (one launch per 64-column half).  dout == NULL selects the pooled form (upstream gradient = demb
// [B, 2D], expanded on chip with `emb`).  dx nullable (first layer).  Leaves one slab per workgroup and half in
// `workspace`; describe them with hcg_mid_reduce_job (one job per half) and sum with hcg_step_tail.
extern "C" int hcg_mid_layer_bwd(const float* dout, const float* demb, const float* emb, const float* out, const float* x,
                                 const float* W, const int64_t* edge_index, int64_t E, const int32_t* graph_ptr,
                                 const int32_t* edge_ptr, int64_t N, int64_t B, int64_t F, int64_t D, int64_t max_nodes,
                                 int64_t max_edges, float slope, int apply_act, float* dx, int32_t* status, void* workspace,
                                 size_t workspace_bytes, hcg_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!hcg_mid_supported(F, D, max_nodes, max_edges)) return HCG_ERR_UNSUPPORTED;
  if (N <= 0 || B <= 0 || E < 0 || !W || !workspace || !x || !graph_ptr || !edge_ptr || !status || (E > 0 && !edge_index))
    return HCG_ERR_INVALID_ARG;
  const bool poolg = (dout == nullptr);
  if (poolg && (!demb || !emb)) return HCG_ERR_INVALID_ARG;
  if ((apply_act & ~3) || ((apply_act & 2) && !dx)) return HCG_ERR_INVALID_ARG;
  if ((poolg || (apply_act & 1)) && !out) return HCG_ERR_INVALID_ARG;   // `out` is only read for leaky' and the pooled routing
  if (workspace_bytes < hcg_mid_workspace_bytes(B, F, D, max_nodes, max_edges)) return HCG_ERR_WORKSPACE;
  if (E == 0) { edge_index = reinterpret_cast<const int64_t*>(graph_ptr); E = 1; }
  if (hcg_w64_applicable(F, D, max_nodes, max_edges))
    return hcg_w64_bwd_launch(dout, demb, emb, out, x, W, edge_index, E, graph_ptr, edge_ptr, B, F, slope, apply_act, dx,
                              (float*)workspace, status, stream);
  const int npad = pad32(max_nodes), emax = pad8(max_edges), fpad = mid_fpad(F);
  const bool ndx = dx != nullptr;
  size_t lds = 0;
  const int gsz = mid_bwd_grid(B, F, max_nodes, max_edges, &lds, ndx);
  const dim3 grid(gsz), blk(MT);
  const size_t slab_half = (size_t)gsz * (DD * fpad + DD);
#define LAUNCH_MID_BWD(KP, NF, DX, PG)                                                                                      \
  do {                                                                                                                      \
    auto kfn = k_mid_layer_bwd<KP, NF, DX, PG>;                                                                             \
    hipError_t e = allow_big_lds<k_mid_layer_bwd<KP, NF, DX, PG>>();                                                        \
    if (e != hipSuccess) return hcg_hip_err(e);                                                                             \
    hipLaunchKernelGGL(kfn, grid, blk, lds, stream, dout, demb, emb, out, (int)D, coff, x, (int)F, Wh, edge_index, E,        \
                       graph_ptr, edge_ptr, (int)B, npad, emax, slope, apply_act, dx, half > 0 ? 1 : 0, partials, status);   \
  } while (0)
#define DISPATCH_MID_BWD(KP, NF)                                                                              \
  do {                                                                                                        \
    if (ndx) { if (poolg) LAUNCH_MID_BWD(KP, NF, true, true); else LAUNCH_MID_BWD(KP, NF, true, false); }     \
    else     { if (poolg) LAUNCH_MID_BWD(KP, NF, false, true); else LAUNCH_MID_BWD(KP, NF, false, false); }   \
  } while (0)
  for (int half = 0; half < (int)(D / DD); ++half) {
    const float* Wh = W + (size_t)half * DD * F;
    const int coff = half * DD;
    float* partials = (float*)workspace + (size_t)half * slab_half;
    if (fpad == 32) DISPATCH_MID_BWD(32, 1);
    else if (fpad == 64) DISPATCH_MID_BWD(64, 1);
    else DISPATCH_MID_BWD(64, 2);
    HCG_CHECK_LAUNCH();
  }
#undef DISPATCH_MID_BWD
#undef LAUNCH_MID_BWD
  return HCG_OK;
}

// job `half` (0 .. D/64 - 1): rows [64 half, 64 half + 64) of dW [D, F] and of db [D]
extern "C" int hcg_mid_reduce_job(const void* workspace, size_t workspace_bytes, int64_t B, int64_t F, int64_t D,
                                  int64_t max_nodes, int64_t max_edges, int half, float* dW, float* db, hcg_reduce_job* job) {
  if (!hcg_mid_supported(F, D, max_nodes, max_edges) || !dW || !db || !job || !workspace || B <= 0 || half < 0 || half >= D / DD)
    return HCG_ERR_INVALID_ARG;
  if (workspace_bytes < hcg_mid_workspace_bytes(B, F, D, max_nodes, max_edges)) return HCG_ERR_WORKSPACE;
  const int fpad = mid_fpad(F);
  const int gsz = hcg_w64_applicable(F, D, max_nodes, max_edges) ? hcg_w64_bwd_grid(B) : mid_bwd_grid(B, F, max_nodes, max_edges, nullptr, true);
  job->slabs = (const float*)workspace + (size_t)half * gsz * (DD * fpad + DD);
  job->nslabs = gsz;
  job->slab_floats = DD * fpad + DD;
  job->nseg = 2;
  job->sse_part = nullptr;
  job->reserved = 0;
  job->seg[0] = hcg_reduce_seg{0, DD * fpad, fpad, (int32_t)F, dW + (size_t)half * DD * F};
  job->seg[1] = hcg_reduce_seg{DD * fpad, DD, 1, 1, db + half * DD};
  return HCG_OK;
}
