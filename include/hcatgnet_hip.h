/*
 * hcatgnet_hip.h -- C ABI of libhcatgnet_hip.so (MI355X / gfx950 only).
 *
 * Drop-in boundary for ONE path of EdAguilarB/hcatgnet: the GCN message-passing
 * forward+backward that `GCN.forward` (reference model/gcn.py:54-76) runs through
 * torch_geometric.  The reference has no FFI layer of its own (pure Python on PyG); this ABI is
 * what a Python `nn.Module` binds with ctypes in place of
 *   - torch_geometric.nn.GCNConv            (call sites model/gcn.py:18-20, 27-29, 58, 62, 127, 131)
 *   - torch_geometric.nn.global_max_pool /
 *     global_mean_pool + torch.cat           (call sites model/gcn.py:65-66, 134-135)
 *   - the readout nn.Linear/LeakyReLU stack  (model/gcn.py:36-45, 70-71)
 *   - torch autograd through all of the above (utils/utils_model.py:65 `loss.backward()`)
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the name ends in `_host`;
 *   - fp32 tensors are row-major contiguous; `edge_index` is int64 [2,E] contiguous
 *     (row 0 = source, row 1 = target: flow source->target, as in the reference data/rhcaa.py:160);
 *     `batch` is int64 [N], non-decreasing graph id; derived indices are int32;
 *   - the caller (PyTorch) owns every buffer; the library never allocates or frees device
 *     memory, keeps no global state, and only ENQUEUES work on the `stream` it is given
 *     (no internal synchronisation) -- safe under hipGraph capture and one-process-per-GPU DP;
 *   - every function returns an int: 0 = HCG_OK, negative = error (see below);
 *     nothing throws or exits across the ABI;
 *   - data-dependent violations that only the device can see (index out of range, unsorted
 *     `batch`, an edge that crosses graphs in blocked mode) are reported through the 4-word
 *     `status` array written by hcg_plan_build: the host wrapper reads it when it chooses to sync.
 */
#ifndef HCATGNET_HIP_H
#define HCATGNET_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HCG_ABI_VERSION 1

#define HCG_OK 0
#define HCG_ERR_INVALID_ARG (-1)
#define HCG_ERR_WORKSPACE (-2)   /* caller-provided workspace too small */
#define HCG_ERR_UNSUPPORTED (-3) /* shape not supported by this entry point */
#define HCG_ERR_HIP_BASE (-1000) /* -1000 - hipError_t */

/* hcg_plan_build flags */
#define HCG_PLAN_GENERAL 0 /* device radix sort; any edge order, any graph size           */
#define HCG_PLAN_BLOCKED 1 /* edges grouped by graph (PyG collate order); per-graph waves */
#define HCG_PLAN_KEEP_STATUS 4 /* OR: do not zero `status` first (flags accumulate in a caller-owned,
                                  once-zeroed buffer until the caller reads and clears it) */
#define HCG_PLAN_PTRS_ONLY 2 /* OR with BLOCKED: only graph_ptr / edge_ptr / status (one launch);
                                all CSR outputs may be NULL -- enough for the hcg_fused_* kernels */

/* bits of status[0] written by hcg_plan_build */
#define HCG_STATUS_INDEX_RANGE 1   /* an edge_index entry outside [0,N)                  */
#define HCG_STATUS_BATCH_UNSORTED 2 /* batch not non-decreasing                           */
#define HCG_STATUS_BATCH_RANGE 4   /* a batch id outside [0,B)                           */
#define HCG_STATUS_EDGE_UNGROUPED 8 /* blocked mode: edges not grouped by graph / cross   */
#define HCG_STATUS_SHAPE_LIMIT 16  /* fused mode: a graph exceeds the tile the host chose */
#define HCG_STATUS_WEIGHTED_SELF_LOOP 32 /* explicit (i,i) edge together with edge_weight: PyG would take its
                                            weight as the node's loop weight; not implemented (loop weight = fill) */

/* activation codes of hcg_linear_* */
#define HCG_ACT_NONE 0
#define HCG_ACT_LEAKY 1

typedef void* hcg_stream_t; /* hipStream_t */
typedef struct hcg_reduce_job hcg_reduce_job; /* defined with the slab reduction below */

int hcg_version(void);
const char* hcg_error_string(int code);

/* ---- batch plan: gcn_norm + CSR/CSC + graph_ptr, ONCE per batch (reference recomputes
 *      gcn_norm every layer, every step: PyG GCNConv(cached=False)) ---------- */
/* workspace sizes of the any-shape entry points, ONE query: kind HCG_WS_PLAN (a, b, c = N, E, B; mode = the plan flags),
 * HCG_WS_LINEAR (M, D_in, D_out), HCG_WS_GCN_LAYER_BWD (N, F, D), HCG_WS_READOUT2 (B) */
#define HCG_WS_PLAN 0
#define HCG_WS_LINEAR 1
#define HCG_WS_GCN_LAYER_BWD 2
#define HCG_WS_READOUT2 3
size_t hcg_general_workspace_bytes(int kind, int64_t a, int64_t b, int64_t c, int mode);

/* Outputs (all caller-allocated):
 *   graph_ptr [B+1]  node range of each graph                    (a9's `batch`)
 *   edge_ptr  [B+1]  edge range of each graph in CSR order (blocked mode; may be NULL in general)
 *   rowptr [N+1], col [E], eid [E]      incoming edges of each node, stable in input order:
 *                                       col = source id, eid = position in edge_index;
 *                                       col = -1 marks an explicit self-loop edge (i, i): as in PyG's
 *                                       add_remaining_self_loops it is not an edge of its own (every node
 *                                       gets exactly one self loop of weight `fill`) and is not counted in dinv
 *   rowptr_t [N+1], col_t [E], eid_t [E] outgoing edges (the transpose, for the backward)
 *   dinv [N]         (fill + sum of incoming weights)^-1/2, 0 where the degree is 0
 *   ew_csr, ew_csc [E]  edge weights permuted to CSR / CSC order (only when edge_weight != NULL)
 *   dinv_unw [N]     (1 + in-degree)^-1/2, only when edge_weight != NULL: the reference hands the
 *                    weights to conv1 only (model/gcn.py:58 vs :62), later layers run unweighted
 *   status [4]       status[0] = OR of HCG_STATUS_* bits (0 = clean)
 */
int hcg_plan_build(const int64_t* edge_index, const int64_t* batch, const float* edge_weight,
                   int64_t N, int64_t E, int64_t B, float fill, int mode,
                   int32_t* graph_ptr, int32_t* edge_ptr,
                   int32_t* rowptr, int32_t* col, int32_t* eid,
                   int32_t* rowptr_t, int32_t* col_t, int32_t* eid_t,
                   float* dinv, float* ew_csr, float* ew_csc, float* dinv_unw, int32_t* status,
                   void* workspace, size_t workspace_bytes, hcg_stream_t stream);

/* ---- dense linear (a4, a10):  y = act(x W^T + b),  W is [D_out, D_in] like nn.Linear ------ */
int hcg_linear_fwd(const float* x, const float* W, const float* b /*nullable*/, float* y,
                   int64_t M, int64_t D_in, int64_t D_out, int act, float slope, hcg_stream_t stream);
/* dz = dy * act'(y) is formed internally (y = saved OUTPUT); then dW = dz^T x, db = colsum dz,
 * dx = dz W (dx nullable).  dz_ws: caller buffer [M, D_out] (may alias nothing). */
int hcg_linear_bwd(const float* dy, const float* y, const float* x, const float* W,
                   float* dx /*nullable*/, float* dW, float* db /*nullable*/, float* dz_ws,
                   int64_t M, int64_t D_in, int64_t D_out, int act, float slope,
                   void* workspace, size_t workspace_bytes, hcg_stream_t stream);

/* ---- one GCNConv + bias + LeakyReLU (a4-a8), general shapes ------------------------------
 *   h = x W^T ; out_i = leaky( dinv_i * sum_{k in row i} w_k dinv_col[k] h_col[k]
 *                               + fill * dinv_i^2 h_i + b )
 *   h_ws: caller buffer [N, D] (the pre-aggregation features; not needed afterwards). */
int hcg_gcn_layer_fwd(const float* x, const float* W, const float* b,
                      const int32_t* rowptr, const int32_t* col, const float* ew_csr /*nullable*/,
                      const float* dinv, float fill, float slope, int apply_act,
                      float* h_ws, float* out, int64_t N, int64_t E, int64_t F, int64_t D,
                      hcg_stream_t stream);
/* backward of the above.  `out` = saved output (gives the LeakyReLU mask), `x` = saved input.
 *   dh_ws: caller buffer [N, D].  dx nullable (first layer: x has no grad). */
int hcg_gcn_layer_bwd(const float* dout, const float* out, const float* x, const float* W,
                      const int32_t* rowptr_t, const int32_t* col_t, const float* ew_csc /*nullable*/,
                      const float* dinv, float fill, float slope, int apply_act,
                      float* dh_ws, float* dx /*nullable*/, float* dW, float* db,
                      int64_t N, int64_t E, int64_t F, int64_t D,
                      void* workspace, size_t workspace_bytes, hcg_stream_t stream);

/* explain mode (f4): gradient of the layer w.r.t. the per-edge multipliers ew_csr handed to hcg_gcn_layer_fwd --
 * PyG's Explainer multiplies every message by an edge mask inside each MessagePassing layer (reference
 * scripts_experiments/explain_gnn.py:39-50: edge_mask_type='object').  dew_csr[k] = dinv_i dinv_{col k} <dY_i, h_{col k}>
 * with dY = dout * act'(out) and h = x W^T (recompute with hcg_linear_fwd); 0 for explicit self-loop entries. */
int hcg_gcn_edge_weight_grad(const float* dout, const float* out, const float* h,
                             const int32_t* rowptr, const int32_t* col, const float* dinv,
                             float slope, int apply_act, float* dew_csr,
                             int64_t N, int64_t E, int64_t D, hcg_stream_t stream);

/* ---- graph pooling (a9): emb[g] = [ max_i a_i , mean_i a_i ]  (max FIRST, model/gcn.py:65-66) */
int hcg_pool_fwd(const float* a, const int32_t* graph_ptr, float* emb /*[B,2D]*/,
                 int64_t N, int64_t B, int64_t D, hcg_stream_t stream);
/* da_i = demb_max[g] * [a_i == max_g] / #ties  +  demb_mean[g] / n_g   (torch amax semantics) */
int hcg_pool_bwd(const float* demb, const float* a, const float* emb, const int32_t* graph_ptr,
                 float* da, int64_t N, int64_t B, int64_t D, hcg_stream_t stream);

/* ---- fused per-layer kernels for batches of small graphs (<= 32 nodes per tile, D = 64, F <= 64) ----
 * One launch per layer: tile of whole graphs -> LDS, gcn_norm rebuilt on chip from the tile's raw COO
 * edges (LDS integer atomics into a [32][32] count matrix: no CSR), x W^T AND the neighbourhood sum
 * on the f32 matrix cores, bias + LeakyReLU, optional [max, mean] pooling epilogue (a3-a9).
 * Needs only graph_ptr / edge_ptr of a BLOCKED plan (edges grouped by graph) and the original int64
 * edge_index [2, E]; unweighted edges (self-loop weight 1).
 *   hcg_fused_graphs_per_tile: graphs packed into one 32-row tile, 0 = shape not supported
 *   emb != NULL  : also write emb[B, 2D] = [max, mean] of `out` per graph (last conv layer)
 *   status       : the plan's status words; HCG_STATUS_SHAPE_LIMIT is raised if a tile exceeds 32 rows,
 *                  HCG_STATUS_EDGE_UNGROUPED if an edge leaves its tile (such edges are ignored)
 */
int hcg_fused_graphs_per_tile(int64_t F, int64_t D, int64_t max_nodes_per_graph);
size_t hcg_fused_workspace_bytes(int64_t B, int64_t F, int64_t D, int graphs_per_tile);
/* EVERY forward form of the small-graph tiles is one entry point with one argument block (HOST struct; zero it first):
 *   W2 != NULL       two stacked conv layers F -> D -> D in ONE launch (the reference's default n_convolutions = 2): the
 *                    first layer's output tile never leaves the chip before it is consumed; out1 (and out2) are written
 *                    once each (the backward needs them)
 *   emb != NULL      also emb[B, 2D] = [max, mean] of the last layer per graph
 *   poolbits != NULL training form of the POOLED (last) layer: its node activations never reach HBM (out2 / out1 of a single
 *                    layer = NULL).  All the pooled backward needs of them is, per element, the sign (LeakyReLU') and
 *                    whether it is its graph's column maximum (torch amax backward: ties share the gradient evenly) -- they
 *                    leave as two bits per element (hcg_fused_aux_bytes: 512 B per 32-row tile, in the matrix-core
 *                    accumulator layout) and hcg_fused_layer_bwd over the SAME plan and graphs_per_tile reads them in
 *                    place of `out` and `emb`
 *   head_W0 != NULL  (stacked training form only) the regression head in the TAIL of the same launch: every workgroup runs
 *                    readout forward, squared error and (unless head_flags = HCG_HEAD_FORWARD_ONLY) the unscaled readout
 *                    backward over its own graphs -- see hcg_head_fwd_bwd for the contract of y / z / out / demb /
 *                    step_counter; head_workspace (hcg_fused_aux_bytes) gets one gradient slab + SSE partial per
 *                    workgroup, described by hcg_fused_head_reduce_job.  A training step is then FOUR launches: this one,
 *                    two backward launches, hcg_step_tail.
 * Replaces reference model/gcn.py:58-66 (+ :70-71 with the head). */
typedef struct hcg_fused_fwd_args {
  const float* x;
  const float* W1;
  const float* b1;
  const float* W2;            /* nullable */
  const float* b2;
  const int64_t* edge_index;
  int64_t E;
  const int32_t* graph_ptr;
  const int32_t* edge_ptr;
  int64_t N, B, F, D;
  int32_t graphs_per_tile;
  int32_t apply_act;
  float slope;
  int32_t head_flags;
  float* out1;                /* layer-1 node embeddings */
  float* out2;                /* layer-2 node embeddings (stacked, no poolbits) */
  float* emb;                 /* nullable */
  uint32_t* poolbits;         /* nullable */
  int32_t* status;
  const float* y;             /* head: targets [B, C] */
  const float* head_W0;       /* NULL = no head */
  const float* head_b0;
  const float* head_W1;
  const float* head_b1;
  int64_t C;
  float* z;
  float* out;
  float* demb;
  void* head_workspace;
  size_t head_workspace_bytes;
  int32_t* step_counter;      /* nullable */
} hcg_fused_fwd_args;
int hcg_fused_forward(const hcg_fused_fwd_args* args_host, hcg_stream_t stream);
#define HCG_FUSED_POOLBITS 0 /* bytes of `poolbits` */
#define HCG_FUSED_HEAD_WS 1  /* bytes of `head_workspace` */
size_t hcg_fused_aux_bytes(int kind, int64_t B, int graphs_per_tile);
int hcg_fused_head_reduce_job(const void* workspace, size_t workspace_bytes, int64_t B, int graphs_per_tile, int64_t C,
                              float* dW0 /*NULL: partials only*/, float* db0, float* dW1, float* db1, hcg_reduce_job* job_host);
/* backward, stage 1 (ONE launch).  dout == NULL selects the pooled form: the upstream gradient is
 * demb[B, 2D] and is expanded on chip with `emb` (ties of the max split evenly) -- or, poolbits != NULL, with the
 * training form's bits (dout, emb, out = NULL then).  dx nullable (first layer).  Leaves one partial slab
 * [D*KPAD + D] per workgroup in `workspace` (hcg_fused_workspace_bytes); stage 2 = hcg_fused_reduce_job + hcg_step_tail
 * sums them in a fixed order into dW [D, F], db [D]: bitwise reproducible.
 * apply_act here is a bit set: bit 0 = multiply the upstream gradient by LeakyReLU'(out) (as in the forward);
 * bit 1 = hand dx down ALREADY multiplied by LeakyReLU'(x) -- x being the previous layer's activated output, whose
 * rows this kernel holds anyway -- so that layer's backward is called with bit 0 clear and `out` = NULL and never
 * reads its own output.  `out` is only required with bit 0 set or in the pooled form without poolbits. */
int hcg_fused_layer_bwd(const float* dout /*nullable*/, const float* demb, const float* emb,
                        const float* out, const uint32_t* poolbits /*nullable*/, const float* x, const float* W,
                        const int64_t* edge_index, int64_t E,
                        const int32_t* graph_ptr, const int32_t* edge_ptr,
                        int64_t N, int64_t B, int64_t F, int64_t D,
                        int graphs_per_tile, float slope, int apply_act,
                        float* dx /*nullable*/, int32_t* status,
                        void* workspace, size_t workspace_bytes, hcg_stream_t stream);

/* ---- fused per-layer kernels for batches of MID-SIZE graphs: one graph per workgroup, <= 224 nodes and <= 1024
 * directed edges per graph, D = 64 or 128 (two 64-column halves, one launch each), F <= 128 (contracted in chunks of 64)
 * -- the size range of the reference's own reaction graphs (56-184 atoms, F = 25 / 32) and of BASELINE's large-ligand
 * configuration (200 nodes, 128-d).  Same contract as hcg_fused_layer_*: raw int64 edge_index grouped by graph + graph_ptr / edge_ptr of
 * a BLOCKED plan, gcn_norm and the CSR rebuilt on chip per graph, unweighted edges, optional pooled epilogue /
 * pooled-gradient prologue, per-workgroup gradient slabs (hcg_mid_reduce_job + hcg_step_tail).
 * `max_nodes` / `max_edges` = largest graph of the batch (host metadata; sizes the dynamic LDS); a graph that exceeds
 * them is skipped and flagged HCG_STATUS_SHAPE_LIMIT.  apply_act of hcg_mid_layer_bwd is the same bit set as in
 * hcg_fused_layer_bwd (bit 1 = premasked dx; `out` nullable when bit 0 is clear and dout is given). */
int hcg_mid_supported(int64_t F, int64_t D, int64_t max_nodes_per_graph, int64_t max_edges_per_graph);
size_t hcg_mid_workspace_bytes(int64_t B, int64_t F, int64_t D, int64_t max_nodes, int64_t max_edges);
int hcg_mid_layer_fwd(const float* x, const float* W, const float* b,
                      const int64_t* edge_index, int64_t E, const int32_t* graph_ptr, const int32_t* edge_ptr,
                      int64_t N, int64_t B, int64_t F, int64_t D, int64_t max_nodes, int64_t max_edges,
                      float slope, int apply_act, float* out, float* emb /*nullable*/,
                      uint8_t* poolbits /*nullable: see hcg_tall_layer_fwd*/,
                      float* xagg /*nullable*/, uint8_t* signbits /*nullable: both, see hcg_tall_layer_fwd*/, int32_t* status,
                      hcg_stream_t stream);
int hcg_mid_layer_bwd(const float* dout /*nullable*/, const float* demb, const float* emb,
                      const float* out, const float* x, const float* W,
                      const int64_t* edge_index, int64_t E, const int32_t* graph_ptr, const int32_t* edge_ptr,
                      int64_t N, int64_t B, int64_t F, int64_t D, int64_t max_nodes, int64_t max_edges,
                      float slope, int apply_act, float* dx /*nullable*/, int32_t* status,
                      void* workspace, size_t workspace_bytes, hcg_stream_t stream);

/* ---- wide layers over large graphs (D = 128; BASELINE configs[4]: 200-atom graphs x 128-d): the layer as a dense
 * row-streaming transform over all nodes (weight image resident in LDS, A fragments straight from global memory, no
 * barrier in the steady state) + a per-graph segmented sum that keeps only the CSR in LDS and gathers rows through L2
 * (csrc/tall.hip).  Same contract and graph limits as hcg_mid_layer_* (raw grouped edge_index + graph_ptr / edge_ptr of a
 * BLOCKED plan, gcn_norm rebuilt on chip per graph, pooled epilogue / pooled-gradient prologue, apply_act bits of
 * hcg_fused_layer_bwd); F a multiple of 4, <= 128.  The forward needs the workspace too (H = x W^T makes a round trip).
 * hcg_tall_layer_bwd leaves dW / db slabs in the workspace: hcg_tall_reduce_jobs fills TWO jobs (dW, db) for
 * hcg_step_tail.  Replaces the same PyG GCNConv call sites (model/gcn.py:58-63) and their autograd. */
int hcg_tall_supported(int64_t F, int64_t D, int64_t max_nodes_per_graph, int64_t max_edges_per_graph);
size_t hcg_tall_workspace_bytes(int64_t N, int64_t B, int64_t F, int64_t D);
int hcg_tall_layer_fwd(const float* x, const float* W, const float* b,
                       const int64_t* edge_index, int64_t E, const int32_t* graph_ptr, const int32_t* edge_ptr,
                       int64_t N, int64_t B, int64_t F, int64_t D, int64_t max_nodes, int64_t max_edges,
                       float slope, int apply_act, float* out, float* emb /*nullable*/,
                       uint8_t* poolbits /*nullable*/, float* xagg /*nullable*/, uint8_t* signbits /*nullable*/,
                       int32_t* status, void* workspace, size_t workspace_bytes, hcg_stream_t stream);
/* `poolbits` [N, D / 4] bytes (training form of the POOLED layer, needs emb; F <= 64 for D = 64): the layer's output is
 * NOT written (`out` may be NULL) -- one byte per (row, 4 columns) leaves instead: bits 0-3 = the value is positive
 * (LeakyReLU'), bits 4-7 = it is its graph's column maximum (where global_max_pool's gradient goes, ties included).  Given
 * to hcg_tall_layer_bwd (pooled form: dout = NULL) they stand in for `out` AND `emb` (both may be NULL): a sixteenth of the
 * bytes on both sides of the step. */
/* `xagg` [N, 32 | 64 (F <= 32 | F <= 64)] + `signbits` [N, 8] bytes (training form of a FIRST, not pooled, 64-wide layer; given
 * together): besides `out` the forward leaves xagg = Ahat x (zero-padded columns) and, per row, four 16-bit pieces (piece j bit
 * q = column 4 q + j of `out` is positive).  Given to hcg_tall_layer_bwd with dout and dx = NULL they select the first-layer
 * form: dW = (dout (.) leaky'(out))^T xagg and db = the column sums of that product's left operand in ONE dense launch (the
 * same sums as (Ahat^T G)^T x in another order) -- no transpose sum, no dH round trip; `out`, `x` are not read.
 * hcg_tall_reduce_jobs(first_layer_form = 1) describes its slabs. */
int hcg_tall_layer_bwd(const float* dout /*nullable*/, const float* demb, const float* emb,
                       const float* out, const uint8_t* poolbits /*nullable*/,
                       const float* xagg /*nullable*/, const uint8_t* signbits /*nullable*/,
                       const int32_t* n_dev /*nullable; first-layer form: the batch's node count on the device (graph_ptr + B) when
                                              N is a CAPACITY (a captured epoch's batch slot): rows past it stay out of the sums*/,
                       const float* x, const float* W,
                       const int64_t* edge_index, int64_t E, const int32_t* graph_ptr, const int32_t* edge_ptr,
                       int64_t N, int64_t B, int64_t F, int64_t D, int64_t max_nodes, int64_t max_edges,
                       float slope, int apply_act, float* dx /*nullable*/, int32_t* status,
                       void* workspace, size_t workspace_bytes, hcg_stream_t stream);

/* ---- fused readout head (a10 + its backward) for the reference's default shape (the autograd path's form):
 *      z = LeakyReLU(emb W0^T + b0) [B,2D]->[B,D];  out = z W1^T + b1 [B,D]->[B,C];  D = 64, C <= 8 (= hcg_head_supported
 *      with D = 64).  forward: one launch (z is kept for the backward).  backward: hcg_readout2_bwd_partial (below: demb
 *      + per-workgroup slabs, workspace HCG_WS_READOUT2) + hcg_readout2_reduce_job + hcg_step_tail. */
int hcg_readout2_fwd(const float* emb, const float* W0, const float* b0, const float* W1, const float* b1,
                     int64_t B, int64_t D, int64_t C, float slope, float* z, float* out, hcg_stream_t stream);

/* ---- regression head: readout forward, squared error, readout backward in ONE launch (a10 + a12 + their backward; f2)
 * The reference's step runs  out = readout(emb); loss = torch.sqrt(MSELoss()(out, y.unsqueeze(1)));
 * loss.backward()  (model/gcn.py:70-71, utils/utils_model.py:64-65).  The whole backward is LINEAR in the one number
 * that needs every graph of the batch, dloss/dout = scale * (out - y) with scale = 1 / (B C sqrt(MSE)): so this kernel --
 * and every conv backward launch behind it -- runs on the UNSCALED error (the gradients of SSE / 2), every workgroup
 * leaves ONE partial sum of squared errors in its slab, and the step's last launch (hcg_step_tail) adds the partials in
 * a fixed order, derives loss and scale and multiplies each gradient element as it reduces it.  No grid-wide exchange
 * (round 2's head kernel spent a grid barrier, 520 sync words and a time-out path on that scalar), any grid size.
 *   z [B,D], out [B,C] : forward results
 *   demb [B,2D]        : d (SSE / 2) / d emb  -- unscaled
 *   workspace          : gradient slabs + SSE partials (describe them with hcg_head_reduce_job)
 *   step_counter       : nullable; one int32 device word incremented by 1 per launch -- the number of the training
 *                        step, read later in the same step by hcg_step_tail's update
 *   flags              : HCG_HEAD_FORWARD_ONLY = no backward (demb / gradient slabs untouched; the partials are written)
 * y is [B,C] like out.  D = 64 or 128 (8 waves per workgroup, W0 fragments from L2 instead of LDS), C <= 8
 * (hcg_head_supported). */
#define HCG_HEAD_FORWARD_ONLY 1
int hcg_head_supported(int64_t D, int64_t C);
size_t hcg_head_workspace_bytes(int64_t B, int64_t D);
int hcg_head_fwd_bwd(const float* emb, const float* y, const float* W0, const float* b0, const float* W1,
                     const float* b1, int64_t B, int64_t D, int64_t C, float slope, int flags,
                     float* z, float* out, float* demb, void* workspace, size_t workspace_bytes,
                     int32_t* step_counter /*nullable*/, hcg_stream_t stream);

/* loss modes of hcg_step_tail / hcg_loss_finalize / hcg_loss_fwd_bwd */
#define HCG_LOSS_MSE 0  /* nn.MSELoss                                  : scale 2 / count            */
#define HCG_LOSS_RMSE 1 /* torch.sqrt(nn.MSELoss) (the reference's step): scale 1 / (count sqrt(MSE)) */
#define HCG_LOSS_SSE 2  /* data parallel with a collective: gradients stay those of SSE / 2, [SSE, count] go to sse_tail;
                           ranks sum gradients, SSE and count (ONE all-reduce) and hcg_sse_finalize / hcg_adam_step_dev_sse
                           scale by 1 / (count sqrt(SSE / count)): the gradient of sqrt(MSE) over the concatenated batch of
                           all ranks, which is what the reference's step computes on one device */

/* ---- MSE loss (a12 / f2): loss[0] = mean((a - b)^2) over n elements, fixed-order reduction;
 *      backward: da = grad_loss[0] * 2 (a - b) / n, db = -da (either may be NULL). */
int hcg_mse_fwd(const float* a, const float* b, int64_t n, float* loss, hcg_stream_t stream);
int hcg_mse_bwd(const float* a, const float* b, const float* grad_loss, int64_t n,
                float* da /*nullable*/, float* db /*nullable*/, hcg_stream_t stream);

/* Loss AND its gradient in one launch, for heads the fused kernel does not cover (other widths): mode 0 = MSE,
 * 1 = sqrt(MSE) (the reference's step, utils/utils_model.py:64), HCG_LOSS_SSE = unscaled da = a - b with [SSE, n] stored
 * in sse_tail.  loss[0] = the loss, loss[1] = MSE; da [n] = d loss / d a. */
int hcg_loss_fwd_bwd(const float* a, const float* b, int64_t n, int mode, float* loss, float* da,
                     float* sse_tail /*nullable unless mode = HCG_LOSS_SSE*/, hcg_stream_t stream);

/* ---- batched slab reduction: ONE launch for all pending gradient reductions of a backward pass.
 * hcg_fused_layer_bwd and hcg_readout2_bwd_partial leave per-workgroup slabs in their workspaces;
 * hcg_fused_reduce_job / hcg_readout2_reduce_job describe them (host-side, no launch), hcg_step_tail
 * sums up to HCG_REDUCE_MAX_JOBS of them in a fixed order. */
#define HCG_REDUCE_MAX_JOBS 8
#define HCG_REDUCE_MAX_SEGS 4
typedef struct hcg_reduce_seg {
  int32_t begin, count;      /* element range of the slab */
  int32_t row_in, row_out;   /* rows of row_in elements are written as rows of row_out (<= row_in) elements */
  float* dst;
} hcg_reduce_seg;
typedef struct hcg_reduce_job {
  const float* slabs;        /* [nslabs][slab_floats] */
  const float* sse_part;     /* != NULL: [nslabs] partial sums of squared errors, one per workgroup, contiguous (the head's
                                job: hcg_head_reduce_job / hcg_fused_head_reduce_job fill it in); NULL = none */
  int32_t nslabs, slab_floats, nseg, reserved;
  hcg_reduce_seg seg[HCG_REDUCE_MAX_SEGS];
} hcg_reduce_job;
/* sizeof of the ABI's HOST structs (a binding checks its mirrors against them) */
#define HCG_STRUCT_REDUCE_JOB 0
#define HCG_STRUCT_TAIL_ARGS 1
#define HCG_STRUCT_FUSED_FWD_ARGS 2
#define HCG_STRUCT_COLLATE_ARGS 3
#define HCG_STRUCT_COLLATE_SLOT 4
size_t hcg_struct_bytes(int which);
int hcg_fused_reduce_job(const void* workspace, size_t workspace_bytes, int64_t N, int64_t B, int64_t F,
                         int64_t D, int graphs_per_tile, float* dW, float* db, hcg_reduce_job* job_host);
int hcg_readout2_bwd_partial(const float* dout, const float* emb, const float* z, const float* W0,
                             const float* W1, int64_t B, int64_t D, int64_t C, float slope, float* demb,
                             void* workspace, size_t workspace_bytes, hcg_stream_t stream);
int hcg_readout2_reduce_job(const void* workspace, size_t workspace_bytes, int64_t B, int64_t C,
                            float* dW0, float* db0, float* dW1, float* db1, hcg_reduce_job* job_host);
/* one job per 64-column half (half = 0 .. D/64 - 1): rows [64 half, 64 half + 64) of dW [D, F] / db [D] */
int hcg_mid_reduce_job(const void* workspace, size_t workspace_bytes, int64_t B, int64_t F, int64_t D,
                       int64_t max_nodes, int64_t max_edges, int half, float* dW, float* db, hcg_reduce_job* job_host);
/* job_host[0] = dW [D, F], job_host[1] = db [D] of hcg_tall_layer_bwd */
int hcg_tall_reduce_jobs(const void* workspace, size_t workspace_bytes, int64_t N, int64_t B, int64_t F, int64_t D,
                         int first_layer_form, float* dW, float* db, hcg_reduce_job* job_host /*[2]*/);
/* the head's slabs; job_host->sse_part = the workgroups' SSE partials.  dW0 == NULL (forward-only head): no segments,
 * the job then only carries the partials (hcg_loss_finalize) */
int hcg_head_reduce_job(const void* workspace, size_t workspace_bytes, int64_t B, int64_t D, int64_t C,
                        float* dW0, float* db0, float* dW1, float* db1, hcg_reduce_job* job_host);
/* `more` (same slab geometry and destinations, slabs directly behind `job`'s) becomes part of `job`: one fixed-order sum */
int hcg_reduce_job_append(hcg_reduce_job* job_host, const hcg_reduce_job* more_host);

/* ---- the step tail: ONE launch behind the backward launches of a training step (f2) --------------------------------
 *   (1) every pending slab reduction (jobs_host[0 .. njobs)), each output element summed in a fixed order;
 *   (2) the loss and its deferred scale: when a job carries SSE partials (sse_part), loss[0] = the loss (`loss_mode`),
 *       loss[1] = MSE, and every reduced element is multiplied by the scale (see hcg_head_fwd_bwd); HCG_LOSS_SSE leaves the
 *       gradients unscaled and stores this rank's [SSE, count] in sse_tail;
 *   (3) inbox != NULL: the data-parallel one-shot exchange over xGMI (below) between reduction and update;
 *   (4) param != NULL: torch.optim.Adam's update of the parameter / moments at the offset of each gradient element in the
 *       flat buffer [grad_flat, grad_flat + n) -- every segment's dst must point into it and every element of it must be
 *       covered by exactly one segment; `step_dev[0]` = 1-based number of THIS update, already advanced when the kernel
 *       runs (the head launch's step_counter does that earlier in the step; this launch only reads it), lr_dev[0] = lr;
 *   (5) next_batch != NULL: the pointers-only plan of the NEXT batch (what hcg_plan_build does with HCG_PLAN_BLOCKED |
 *       HCG_PLAN_PTRS_ONLY | HCG_PLAN_KEEP_STATUS): the one launch of the next step that depends on nothing of this one.
 * Replaces the four hcg_reduce_slabs* entry points of round 2 (one symbol per fusion combination).  HOST struct; zero it
 * first, unused parts stay NULL. */
typedef struct hcg_tail_args {
  const hcg_reduce_job* jobs_host;
  int32_t njobs;
  int32_t loss_mode;           /* HCG_LOSS_*; only read when a job has sse_part */
  float loss_count;            /* elements of the squared-error sum on this rank: B * C */
  float beta1, beta2, eps;
  float* loss;                 /* [2], nullable without exchange */
  float* sse_tail;             /* [2], nullable */
  float* grad_flat;            /* [n] ([n + 2] with an exchange: gradients | SSE | count) */
  float* param;                /* NULL = no update */
  float* exp_avg;
  float* exp_avg_sq;
  int64_t n;
  const float* lr_dev;
  const int32_t* step_dev;
  const int64_t* next_edge_index;
  const int64_t* next_batch;   /* NULL = no plan */
  int64_t next_N, next_E, next_B;
  int32_t* next_graph_ptr;
  int32_t* next_edge_ptr;
  int32_t* next_status;
  void* inbox;                 /* NULL = no exchange */
  void* const* peers_host;     /* HOST array of `world` device pointers (peers_host[rank] = inbox) */
  int32_t rank, world, xchg_mode;
  int32_t reserved;
  int32_t* xchg_err;
} hcg_tail_args;
int hcg_step_tail(const hcg_tail_args* args_host, hcg_stream_t stream);
/* forward-only steps (the reference's eval_network body, utils/utils_model.py:75-78): the loss alone from a head job's partials */
int hcg_loss_finalize(const hcg_reduce_job* head_job_host, float count, int loss_mode, float* loss,
                      float* sse_tail /*nullable*/, hcg_stream_t stream);

/* ---- Adam update (f2) over one contiguous fp32 segment: torch.optim.Adam's rule (amsgrad / weight_decay /
 *      maximize off).  `step` = 1-based count of this update.  One launch. */
int hcg_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                  float lr, float beta1, float beta2, float eps, int64_t step, hcg_stream_t stream);

/* Same update with the step count and the learning rate in DEVICE memory, so that the launch can sit inside a
 * captured hipGraph: `step_dev[0]` = number of updates done so far (the kernel uses step_dev[0] + 1 and the last
 * workgroup to finish stores the incremented count; `step_dev[1]` is its ticket word, zero between launches),
 * `lr_dev[0]` = learning rate (the host rewrites it when a scheduler changes it). */
int hcg_adam_step_dev(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                      const float* lr_dev, float beta1, float beta2, float eps, int32_t* step_dev, hcg_stream_t stream);

/* Data-parallel SSE form (HCG_LOSS_SSE): `flat` = [n summed SSE/2-gradients | SSE | count].  Scales the n
 * gradients in place by 1 / (count * L), L = sqrt(SSE / count), and stores loss[0] = L, loss[1] = SSE / count.
 * hcg_adam_step_dev_sse does the same and applies hcg_adam_step_dev's update with the scaled gradient: one launch. */
int hcg_sse_finalize(float* flat, int64_t n, float* loss, hcg_stream_t stream);
int hcg_adam_step_dev_sse(float* param, float* flat, float* exp_avg, float* exp_avg_sq, int64_t n,
                          const float* lr_dev, float beta1, float beta2, float eps, int32_t* step_dev, float* loss,
                          hcg_stream_t stream);

/* ---- data parallel: one-shot gradient exchange over xGMI, fused between the slab reduction and the update ------------
 * Every rank owns an inbox (hcg_xchg_inbox_bytes: 2 x world x (n + 2) eight-byte {value, step} granules) in fine-grained
 * device memory (hcg_xchg_alloc: the ONE allocation this library makes -- ordinary device memory does not make a peer's
 * stores visible to a running kernel), exports it (hcg_xchg_ipc_export -> 64-byte handle, exchanged by the host through
 * torch.distributed) and maps its peers' (hcg_xchg_ipc_open).  hcg_step_tail with `inbox` set: every reduced element is
 * first written into all peers' inboxes (one 8-byte system-scope store per peer, over the direct links), the peers'
 * contributions are polled out of the own inbox and added in rank order, and the update runs on the total: xchg_mode
 * HCG_XCHG_MEAN = mean over the ranks of each rank's own loss gradient; HCG_XCHG_SSE = the gradient of sqrt(MSE) over the
 * concatenated batch (SSE and count travel as elements n, n + 1: from the head's partials, or, without a job that carries
 * them, from grad_flat[n], [n + 1]; loss[0..1] = the global sqrt(MSE), MSE).  Polls are bounded (2 s): HCG_XCHG_ERR_TIMEOUT
 * is ORed into xchg_err[0] and the element becomes NaN.  `step_dev[0]` stamps the granules: it must advance by one per
 * exchange on every rank, and an inbox must be re-zeroed before it serves another optimiser. */
#define HCG_XCHG_MAX_WORLD 8
#define HCG_XCHG_HANDLE_BYTES 64
#define HCG_XCHG_MEAN 0
#define HCG_XCHG_SSE 1
#define HCG_XCHG_ERR_TIMEOUT 1
size_t hcg_xchg_inbox_bytes(int64_t n, int world);
/* workgroups of the exchanging hcg_step_tail that are resident at once on this device (occupancy x CUs).  A launch whose
 * jobs need more polling workgroups (sum over the jobs of ceil(slab_floats / 32)) is refused with HCG_ERR_UNSUPPORTED: two
 * ranks' resident pollers could otherwise wait on each other's queued publishers until the bounded polls expire */
int hcg_xchg_resident_blocks(void);
int hcg_xchg_alloc(size_t bytes, void** ptr);
int hcg_xchg_free(void* ptr);
int hcg_xchg_zero(void* ptr, size_t bytes);
int hcg_xchg_ipc_export(void* ptr, void* handle64);
int hcg_xchg_ipc_open(const void* handle64, void** ptr);
int hcg_xchg_ipc_close(void* ptr);

/* ---- on-device collation (f1): gather the graphs of an HBM-resident dataset into PyG-style batches, ONE launch for one
 * batch or for every batch of an epoch.
 * Dataset side: x_all [N_all, F], local edge lists src_all / dst_all (int32 ids inside their graph),
 * node_ptr_all / edge_ptr_all [G+1] (int64), y_all [G], idx_all [G].
 * A slot = one batch: `ids` [B] selects its graphs (device); graph_ptr / edge_ptr [B+1] (int32) are the batch's prefix sums
 * (the host knows the sizes; they double as the fused path's plan).  Writes x_out [N_out, F], edge_index_out [2, E_out]
 * (batch-local ids), batch_out [N_out], y_out / idx_out [B] (nullable).
 * nslots == 1: the batch is `slot` (host values).  nslots > 1: `slots_dev` is a DEVICE array of nslots descriptors (every
 * batch of a shuffled epoch: train.EpochWindow collates an epoch in one launch in front of its steps), max_B the largest
 * slot.B; the caller has validated them.  Replaces reference data/datasets.py:74-78 + PyG collate (call_methods.py:41-46). */
typedef struct hcg_collate_slot {
  const int64_t* ids;
  const int32_t* graph_ptr;
  const int32_t* edge_ptr;
  float* x_out;
  int64_t* edge_index_out;
  int64_t* batch_out;
  float* y_out;               /* nullable */
  int64_t* idx_out;           /* nullable */
  int64_t B, N_out, E_out;
} hcg_collate_slot;
typedef struct hcg_collate_args {
  const float* x_all;
  const int32_t* src_all;
  const int32_t* dst_all;
  const int64_t* node_ptr_all;
  const int64_t* edge_ptr_all;
  const float* y_all;         /* nullable when no slot has y_out */
  const int64_t* idx_all;     /* nullable when no slot has idx_out */
  int64_t F;
  int32_t nslots;
  int32_t reserved;
  hcg_collate_slot slot;                 /* nslots == 1 */
  const hcg_collate_slot* slots_dev;     /* nslots > 1 */
  int64_t max_B;                         /* nslots > 1 */
} hcg_collate_args;
int hcg_collate(const hcg_collate_args* args_host, hcg_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* HCATGNET_HIP_H */
