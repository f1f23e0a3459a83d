/* gnnrag.h - C ABI of libgnnrag_hip.so: the MI355X (gfx950) implementation of the
 * GNN-RAG / ReaRev reasoning hot path.
 *
 * The reference (cmavro/GNN-RAG @ v2) is pure Python: it has NO plugin / operator / FFI
 * interface for this path.  The seam is the Python class
 * `ReasonGNNLayer` (gnn/modules/kg_reasoning/reasongnn.py:10-174) plus `TypeLayer`
 * (gnn/modules/layer_init.py:9-65); the entry points below are what a ctypes binding inside
 * those classes calls (the binding itself is shown in INTEGRATION.md and shipped as
 * gnn-rag_amd/modules/).  Each entry point cites the reference code it replaces.
 *
 * Conventions
 *  - plain C: pointers + sizes only, no torch types.  All data pointers are DEVICE pointers
 *    owned by the caller; the library never allocates, frees or retains them.
 *  - all work is enqueued on the given hipStream_t (passed as void*); no internal
 *    synchronisation (one exception: gnnrag_csr_build waits for the stream once, to hand the
 *    relation counts back to the host), re-entrant per stream.  No state that affects results:
 *    the only process-wide data are idempotent per-device caches of launch attributes (raised
 *    dynamic-LDS caps, CU counts), keyed by the device current at the call, so one process may
 *    drive several GPUs.
 *  - fp32 values, int32 indices, row-major contiguous.
 *  - return value: 0 = success; > 0 = hipError_t of a failed runtime call / launch;
 *    < 0 = GNNRAG_E_* argument error.  gnnrag_error_string() renders either.
 *  - node index = question * N + slot (the reference pre-offsets heads/tails the same
 *    way, gnn/dataset_load.py:483,492-493); direction 0 = forward (head -> tail),
 *    direction 1 = inverse (tail -> head).
 */
#ifndef GNNRAG_H_
#define GNNRAG_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GNNRAG_ABI_VERSION 16

#define GNNRAG_E_BADARG      (-1)  /* null pointer / negative size / inconsistent sizes   */
#define GNNRAG_E_UNSUPPORTED (-2)  /* shape outside the compiled kernel set (see DESIGN)  */
#define GNNRAG_E_WORKSPACE   (-3)  /* caller-provided buffer too small                    */
#define GNNRAG_E_TUPLE       (-4)  /* edge tuple invalid: node / relation id out of range,
                                      or a fact whose head and tail lie in two questions  */

typedef void* gnnrag_stream_t; /* hipStream_t */

/* Destination-sorted structure of one batch of question subgraphs, both directions.
 * Replaces the 7 COO tensors of BaseGNNLayer.build_matrix (base_gnn.py:19-51), of which
 * ReaRev uses fact2tail/head2fact (forward) and fact2head/tail2fact (inverse).
 * Facts of one destination are stored in ascending fact id - except hub rows (more than heavy_deg
 * facts), which are stored in ascending (relation id, fact id) so that the walk can add the priors of
 * a run of equal relations and fetch the run's table row once - so every sum has one fixed order
 * (bit-reproducible, independent of how a batch is sharded over GPUs). */
typedef struct gnnrag_csr {
  int32_t B;            /* questions in the batch                                          */
  int32_t N;            /* max_local_entity: node slots per question                       */
  int32_t R1;           /* rows of the relation tables (num_kb_relation + 1)               */
  int32_t heavy_deg;    /* rows with more facts than this are walked by a whole workgroup  */
  int64_t F;            /* facts (typed edges + self loops) in the batch                   */
  int32_t* row_ptr[2];  /* [B*N+1]  first fact of each destination node                    */
  int32_t* edge[2];     /* [F][2]   (source node, relation id) per fact, sorted by dst     */
  int32_t* perm[2];     /* [F]      sorted position -> fact id of the caller's tuple       */
  float*   w_gnn[2];    /* [F] v_f^2 (normalized_gnn: weight enters both sparse products,
                                      base_gnn.py:38-47) or NULL                           */
  float*   w_rel[2];    /* [F] weight_rel_list (TypeLayer norm_rel, layer_init.py:39-40)
                                      or NULL                                              */
  int32_t* heavy[2];    /* [heavy_cap] list of heavy destination nodes, ascending          */
  int32_t* chunk_off[2];/* [heavy_cap+1] first 256-fact chunk of each heavy node (prefix)  */
  int32_t* n_heavy;     /* [2] device counters: heavy nodes per direction                  */
  int32_t* n_chunks;    /* [2] device counters: heavy chunks per direction                 */
  int32_t  heavy_cap;
  int32_t  max_chunks;  /* upper bound of n_chunks[d] (sizes the partial-sum workspace)    */
  int32_t* big_cnt;     /* [B]    nodes of each question with > big_deg facts in a direction */
  int32_t* big_nodes;   /* [B][N] their node ids (order irrelevant)                         */
  int32_t  big_deg;
  int32_t  hub_sorted;  /* 1: hub rows are in (relation, fact id) order (built with R1 > 1024), 0: fact order   */
  /* Per-question relation compaction (fused path).  A question's subgraph touches a small part of
   * the KB's relation vocabulary (hundreds of Freebase's ~6k relations), so its relation tables
   * are built over the relations it USES: compact row rel_off[b] + j  <->  (b, j-th smallest
   * relation id used by question b). */
  int32_t* edge_l[2];   /* [F][2]   (source node, compact relation index within its question)  */
  int32_t* rel_off;     /* [B+1]    first compact row of each question (prefix sum)            */
  int32_t* rel_rows;    /* [rel_total][2]  (question, relation id) of every compact row        */
  int32_t  rel_total;   /* compact rows in the batch (host copy, filled by gnnrag_csr_build)   */
  int32_t  rel_max;     /* largest number of relations used by one question                    */
  /* Merged rows (fused LDS walk): the facts arriving at node n in direction 0 followed by those of direction 1 form
   * ONE run [row_ptr[0][n] + row_ptr[1][n], row_ptr[0][n+1] + row_ptr[1][n+1]) of a 2F-long stream.  edge_m holds the
   * (source node, compact relation) records in that order - direction 1's relation index offset by the question's
   * relation count + 1, so that it addresses the second table slice behind the first one's zero row - and m_from
   * where each record came from (d * F + sorted position in direction d: the per-fact weights are read through it).
   * The walk then steps through a node's facts of both directions in one loop (same summation order: direction 0's
   * facts in ascending fact id, then direction 1's). */
  int32_t* edge_m;      /* [2F][2]                                                              */
  int32_t* m_from;      /* [2F]                                                                 */
  int32_t* m_dst;       /* [2F]  destination node of every merged record: for callers that cut the merged stream into
                                 equal fact ranges (no kernel of this library reads it any more)         */
  /* Dense hub form of the gather walk (tables larger than LDS): the hubs of question b are the list entries
   * hub_q_off[d][b] .. hub_q_off[d][b+1]; hub_wbase[d][b] = sum over earlier questions of hubs x relations in use
   * (rounded up to 4) = offset of the question's hub-by-relation weight block (saturates at INT32_MAX). */
  int32_t* hub_q_off[2];/* [B+1]                                                                */
  int32_t* hub_wbase[2];/* [B+1]                                                                */
} gnnrag_csr;

/* Bytes of caller-owned device memory a gnnrag_csr needs (persistent part / build scratch). */
size_t gnnrag_csr_bytes(int64_t F, int32_t B, int32_t N, int32_t R1, int has_w_gnn, int has_w_rel);
size_t gnnrag_csr_scratch_bytes(int64_t F, int32_t B, int32_t N, int32_t R1);

/* Builds the structure on the device.  heads/rels/tails are the batch tuple's first three
 * arrays (dataset_load.py:527) narrowed to int32; w_gnn = weight_list (used when
 * args['normalized_gnn']), w_rel = weight_rel_list (used when norm_rel), either may be NULL.
 * Replaces BaseGNNLayer.build_matrix (base_gnn.py:19-51) and the two COO builds inside
 * TypeLayer.forward (layer_init.py:35-36,53-54).  The tuple is validated on the device (node ids in
 * [0, B*N), relation ids in [0, R1), no fact across two questions): GNNRAG_E_TUPLE otherwise.
 * Synchronises `stream` once at the end (rel_total / rel_max are returned in *out). */
int gnnrag_csr_build(const int32_t* heads, const int32_t* rels, const int32_t* tails,
                     const float* w_gnn, const float* w_rel,
                     int64_t F, int32_t B, int32_t N, int32_t R1,
                     void* csr_mem, size_t csr_bytes, void* scratch, size_t scratch_bytes,
                     gnnrag_csr* out, gnnrag_stream_t stream);

/* gnnrag_csr_build WITHOUT the wait for the stream: the caller passes what the build would otherwise read back -
 * rel_total = sum over the questions of the distinct relation ids among a question's facts, rel_max = the largest such
 * count (a fact cache knows both per question when it caches it).  Everything is enqueued and
 * the call returns; the tuple's validation bits and the device-side counts stay behind the structure and
 * gnnrag_csr_status reads them back (one stream wait) whenever the caller wants the check.  Wrong counts are the
 * caller's error: a too small rel_total makes the fused path's tables too short.  rel_total < 0 or rel_max < 0: the
 * waiting form (= gnnrag_csr_build).  Replaces the same reference code (base_gnn.py:19-51). */
int gnnrag_csr_build_counts(const int32_t* heads, const int32_t* rels, const int32_t* tails,
                            const float* w_gnn, const float* w_rel,
                            int64_t F, int32_t B, int32_t N, int32_t R1, int32_t rel_total, int32_t rel_max,
                            void* csr_mem, size_t csr_bytes, void* scratch, size_t scratch_bytes,
                            gnnrag_csr* out, gnnrag_stream_t stream);
/* 0 when the structure's tuple passed the device-side validation and its relation counts equal the device's;
 * GNNRAG_E_TUPLE / GNNRAG_E_BADARG otherwise.  Waits for `stream`. */
int gnnrag_csr_status(const gnnrag_csr* csr, gnnrag_stream_t stream);

/* The structure of a batch as the CONCATENATION of per-question structures that are already on the device (SURVEY.md
 * section 8 f-1: "cached per-question int32 CSR built once at load time, batch = concatenation with offsets").
 * parts: HOST array of B structures, each built by gnnrag_csr_build with B = 1 for ONE question (node ids 0 .. N-1,
 * the same N and R1); question b of the batch gets node ids b * N .., facts in the order of `parts` (the order of the
 * reference's batch tuple, dataset_load.py:481-506).  Questions are disjoint node ranges and the structure is sorted
 * by destination node, so no sort is needed: every array is a copy with offsets added.  The result is bit-identical
 * to gnnrag_csr_build on the concatenated tuple.  No scratch, no synchronisation (rel_total / rel_max are sums / the
 * maximum of the parts' host fields).  Per-fact weights are attached afterwards (gnnrag_csr_permute_weight).
 * csr_mem: gnnrag_csr_bytes(sum of the parts' F, B, N, R1, 0, 0) bytes. */
int gnnrag_csr_concat(const gnnrag_csr* const* parts, int32_t B, int32_t N, int32_t R1, void* csr_mem, size_t csr_bytes,
                      gnnrag_csr* out, gnnrag_stream_t stream);

/* HOST helper (no device work): narrows the batch tuple's int64 id arrays (dataset_load.py:527) into one [3, F]
 * int32 block (heads, rels, tails) with up to `nthreads` threads and checks that every id lies in [0, 2^31):
 * GNNRAG_E_TUPLE otherwise.  `out` is host memory (pinned memory makes the following upload faster). */
int gnnrag_narrow_tuple(const int64_t* heads, const int64_t* rels, const int64_t* tails, int64_t F,
                        int32_t* out, int32_t nthreads);

/* Permutes a per-fact weight array of the caller's tuple into the structure's sorted order for
 * both directions: out_d[i] = w[perm_d[i]] (squared if square != 0).  Lets a binding attach
 * weight_list / weight_rel_list lazily (only when normalized_gnn / norm_rel ask for them). */
int gnnrag_csr_permute_weight(const gnnrag_csr* csr, const float* w_per_fact, int square,
                              float* out_fwd, float* out_inv, gnnrag_stream_t stream);

/* Math mode of the dense projections, an argument of every entry point that multiplies matrices
 * (gnnrag_linear*, gnnrag_update_score*, gnnrag_relation_tables, gnnrag_reason_layer):
 *   GNNRAG_MATH_FP32   v_mfma_f32_16x16x4_f32: bit-exact fp32 fmaf chains;
 *   GNNRAG_MATH_BF16X3 each fp32 operand split EXACTLY into three bf16 planes, six plane products
 *                      on v_mfma_f32_16x16x32_bf16 with fp32 accumulation: per-product relative
 *                      error <= 3*2^-24 (fp32 class), about 2.5x the fp32 MFMA rate. */
#define GNNRAG_MATH_FP32   0
#define GNNRAG_MATH_BF16X3 1
/* GNNRAG_MATH_MIXED: per kernel the faster of the two fp32-class forms - bf16x3 where a W-resident bf16x3 kernel
 * exists (relation tables, self-block update at hidden size 193..208) and on the k-tiled kernel, exact fp32 on the
 * W-resident fp32 update kernel (other hidden sizes) and on the skinny kernel (small M: the relation transforms). */
#define GNNRAG_MATH_MIXED  2

/* C[M,Nout] = act( A[M,K] . W[Nout,K]^T + bias[Nout] + add[row < add_rows, :] ), fp32 MFMA.
 * Used for  T_d = rel_linear_step(rel_features_d) (+ pos_emb_d(rel)), computed once per
 * relation row instead of once per fact (reasongnn.py:71,75-79 / :98,102-105), and for
 * TypeLayer's kb_self_linear(rel_features) (layer_init.py:47-49).
 * bias/add may be NULL.  relu != 0 applies max(.,0). */
int gnnrag_linear(const float* A, int64_t M, int32_t K, const float* W, const float* bias,
                  const float* add, int64_t add_rows, int relu,
                  float* C, int32_t Nout, int32_t math, gnnrag_stream_t stream);

/* Two gnnrag_linear problems that share W, bias and shapes (the forward and inverse relation
 * transforms of one layer call) in ONE launch. */
int gnnrag_linear_pair(const float* A0, const float* A1, int64_t M, int32_t K, const float* W,
                       const float* bias, const float* add0, const float* add1, int64_t add_rows,
                       float* C0, float* C1, int32_t Nout, int32_t math, gnnrag_stream_t stream);

/* agg[n, 2i+d, :] = sum_{f: dst_d(f)=n} w_f * dist[src_d(f)] * relu(T_d[rel_f,:] * ins[n/N, i, :])
 * = reason_layer (reasongnn.py:61-89, d=0) and reason_layer_inv (reasongnn.py:91-116, d=1)
 * for every instruction i, in the concat order of reasongnn.py:150-158.
 * dist [B*N], ins [B,I,D], T_fwd/T_inv [R1,D], agg [B*N, 2*I*D]. */
size_t gnnrag_aggregate_workspace_bytes(const gnnrag_csr* csr, int32_t D, int32_t I);
int gnnrag_aggregate(const gnnrag_csr* csr, const float* dist, const float* ins,
                     const float* T_fwd, const float* T_inv, float* agg,
                     int32_t D, int32_t I, void* workspace, size_t workspace_bytes,
                     gnnrag_stream_t stream);

/* Fused form of the same aggregation: e2e_linear is linear over the concatenated blocks, so it is
 * applied to the per-question relation tables first (gnnrag_relation_tables) and the walk emits
 *   out[n,:] = sum_d sum_{f: dst_d(f)=n} w_f * dist[src_d(f)] * P[d, row(n/N, rel_f), :]    [B*N, D]
 * = sum_k W_e2e[:, block k] . agg[n,k,:], i.e. the neighbour part of reasongnn.py:161-163 without
 * ever writing agg.  P [2,rel_total,D] over the compact rows.
 * workspace: gnnrag_aggregate_workspace_bytes(csr, D, 1). */
int gnnrag_aggregate_fused(const gnnrag_csr* csr, const float* dist, const float* P, float* out,
                           int32_t D, void* workspace, size_t workspace_bytes,
                           gnnrag_stream_t stream);

/* Which kernel gnnrag_aggregate_fused dispatches for this structure and hidden size (decided on the
 * host from rel_max and D, nothing is launched): lets tests and bench.py name the kernel they ran. */
#define GNNRAG_WALK_L2_GATHER 0  /* k_walk_light<FUSED>: table rows gathered from L2 (tables exceed a CU's LDS) */
#define GNNRAG_WALK_LDS_16    1  /* k_walk_slice<FUSED,1>: 16-column table slices in LDS                      */
#define GNNRAG_WALK_LDS_32    2  /* k_walk_slice<FUSED,2>: 32-column slices (small per-question tables)       */
int gnnrag_aggregate_fused_variant(const gnnrag_csr* csr, int32_t D);

/* Which form the HUB ROWS of the gather walk take in a gnnrag_aggregate_fused call with this structure, hidden size and
 * workspace (the sizes of the hub-by-relation weight blocks live on the device, so the kernels decide; this entry runs
 * the same predicate on the same kernel arguments in a one-thread launch).  form_dev: 4 device int32 -
 * [0] GNNRAG_HUB_FORM_*, [1] / [2] hub rows of direction 0 / 1, [3] relation ranges per question.  Diagnostics for
 * tests and bench.py ("which kernel ran"); nothing on the product path calls it. */
#define GNNRAG_HUB_FORM_NONE    0  /* the call has no dense hub kernels (LDS walk, or the form is switched off)   */
#define GNNRAG_HUB_FORM_DENSE   1  /* k_hub_weights / k_hub_dense / k_hub_finish                                   */
#define GNNRAG_HUB_FORM_CHUNKED 2  /* weight blocks do not fit the workspace: k_heavy_partial / k_heavy_reduce     */
int gnnrag_aggregate_fused_hub_form(const gnnrag_csr* csr, int32_t D, void* workspace, size_t workspace_bytes,
                                    int32_t* form_dev, gnnrag_stream_t stream);

/* h_out = relu(e2e_linear(cat(h, agg)))            (reasongnn.py:161-163)
 * score = score_func(h_out) + (1 - mask) * -1e11   (reasongnn.py:165-168), mask add in fp32.
 * h [BN,D], agg [BN,2I*D], W [D,(2I+1)D], b [D], w_s [D], b_s [1] (device), mask [BN]. */
int gnnrag_update_score(const float* h, const float* agg, const float* W, const float* b,
                        const float* w_s, const float* b_s, const float* mask,
                        float* h_out, float* score, int64_t BN, int32_t D, int32_t I, int32_t math,
                        gnnrag_stream_t stream);

/* dist[g,:] = softmax(score[g,:]) over the N slots of each question (reasongnn.py:169). */
int gnnrag_masked_softmax(const float* score, float* dist, int32_t B, int32_t N,
                          gnnrag_stream_t stream);

/* h0[n,:] = relu( sum_{tail_f=n} v_f T[rel_f,:] + sum_{head_f=n} v_f T[rel_f,:] ),
 * v_f = w_rel if use_w_rel else 1   (TypeLayer.forward, layer_init.py:53-57);
 * T [R1,D] = kb_self_linear(rel_features) from gnnrag_linear.
 * workspace: gnnrag_aggregate_workspace_bytes(csr, D, 1). */
int gnnrag_typelayer(const gnnrag_csr* csr, const float* T, int use_w_rel, float* h0,
                     int32_t D, void* workspace, size_t workspace_bytes, gnnrag_stream_t stream);

/* Facts of a batch ordered by (question, relation), for the backward's table gradients: row = compact
 * relation row of gnnrag_csr (rel_rows), facts of a row in ascending fact id, rows cut into chunks of
 * at most 256 facts.  Built on first use (training only) from the same int32 tuple as the structure. */
typedef struct gnnrag_relorder {
  int64_t  F;
  int32_t  rel_total;
  int32_t  n_chunks;    /* host copy, filled by gnnrag_relorder_build                                */
  int32_t* ht;          /* [F][2]  (head node, tail node) per fact in (question, relation) order     */
  int32_t* perm;        /* [F]     position in that order -> fact id of the caller's tuple           */
  float*   w;           /* [F]     v_f^2 in that order (normalized_gnn) or NULL                      */
  int32_t* row_ptr;     /* [rel_total+1] first position of each compact relation row                 */
  int32_t* chunk_ptr;   /* [rel_total+1] first chunk of each row (prefix of ceil(len/256))           */
} gnnrag_relorder;
size_t gnnrag_relorder_bytes(const gnnrag_csr* csr, int has_w);
size_t gnnrag_relorder_scratch_bytes(const gnnrag_csr* csr);
/* heads/rels/tails: the tuple the structure was built from; w_per_fact: weight_list (squared inside) or
 * NULL.  Synchronises `stream` once (n_chunks is returned in *out). */
int gnnrag_relorder_build(const gnnrag_csr* csr, const int32_t* heads, const int32_t* rels,
                          const int32_t* tails, const float* w_per_fact,
                          void* mem, size_t mem_bytes, void* scratch, size_t scratch_bytes,
                          gnnrag_relorder* out, gnnrag_stream_t stream);

/* Backward of gnnrag_aggregate (what autograd derives for reasongnn.py:61-116), so that training
 * (train_model.py:209-233) runs on the HIP operator.  g_agg [BN,2I*D] is the gradient of agg;
 *   g_dist [BN], g_ins [B,I,D], g_T_fwd / g_T_inv [R1,D]  are fully written (not accumulated into).
 * relorder != NULL (and D % 4 == 0, I <= 4): the table / instruction gradients are gathered over the
 * facts of each (question, relation) row - no atomics, one fixed summation order, any number of
 * relations per question.  relorder == NULL: relation-bucketed sums in LDS (ds_add_f32; reproducible
 * to rounding only; GNNRAG_E_UNSUPPORTED when one question uses more relations than one CU's LDS
 * holds, ~1200).
 * workspace: gnnrag_backward_workspace_bytes(csr, relorder, D, I) bytes of device scratch.  D <= 1024. */
size_t gnnrag_backward_workspace_bytes(const gnnrag_csr* csr, const gnnrag_relorder* relorder, int32_t D,
                                       int32_t I);
int gnnrag_aggregate_backward(const gnnrag_csr* csr, const gnnrag_relorder* relorder,
                              const float* dist, const float* ins,
                              const float* T_fwd, const float* T_inv, const float* g_agg,
                              float* g_dist, float* g_ins, float* g_T_fwd, float* g_T_inv,
                              int32_t D, int32_t I, void* workspace, size_t workspace_bytes,
                              gnnrag_stream_t stream);

/* Backward of gnnrag_aggregate_fused (training on the fused form, linear_dropout = 0):
 *   nbr[n, :] = sum_d sum_{f: dst_d(f)=n} w_f * dist[src_d(f)] * P[d, row(b, rel_f), :]
 *   g_dist[s]        = sum_d sum_{f: src_d(f)=s} w_f * < g_nbr[dst_d(f), :], P[d, row(b, rel_f), :] >
 *   g_P[d, row, :]   = sum_{f in row} w_f * dist[src_d(f)] * g_nbr[dst_d(f), :]
 * both fully written; the relation tables P themselves are a differentiable dense expression of a few ten thousand
 * rows on the caller's side (gnn-rag_amd/autograd.py: relation_tables_dense).  Needs the (question, relation) ordering
 * (gnnrag_relorder) and D % 4 == 0; gather kernels, chunk partials summed in a fixed order, no atomics.
 * workspace: gnnrag_backward_workspace_bytes(csr, relorder, D, 1). */
int gnnrag_aggregate_fused_backward(const gnnrag_csr* csr, const gnnrag_relorder* relorder, const float* dist,
                                    const float* P, const float* g_nbr, float* g_dist, float* g_P, int32_t D,
                                    void* workspace, size_t workspace_bytes, gnnrag_stream_t stream);

/* Backward of gnnrag_typelayer with respect to T (layer_init.py:47-57):
 *   g_T[r,:] = sum_{f: rel_f=r} v_f (g_pre[tail_f,:] + g_pre[head_f,:]),
 * g_pre [BN,D] = gradient of the pre-activation (= g_h0 where h0 > 0, else 0).  g_T [R1,D] is fully
 * written.  relorder != NULL and D % 4 == 0: atomic-free gather over the (question, relation) rows, v_f read
 * from w_rel_per_fact (weight_rel_list in the caller's fact order) when use_w_rel; else the LDS form with the
 * structure's own w_rel (limits as for gnnrag_aggregate_backward).
 * workspace: gnnrag_backward_workspace_bytes(csr, relorder, D, 1). */
int gnnrag_typelayer_backward(const gnnrag_csr* csr, const gnnrag_relorder* relorder, const float* g_pre,
                              const float* w_rel_per_fact, int use_w_rel, float* g_T,
                              int32_t D, void* workspace, size_t workspace_bytes, gnnrag_stream_t stream);

/* Weight gradient of a dense projection (what autograd derives for nn.Linear.weight):
 *   C[N1, N2] = A[M, N1]^T . B[M, N2]      e.g. dW = dY^T . X with A = dY [M, Nout], B = X [M, K].
 * Exact fp32 on the matrix cores; the row range is cut into chunks whose partial blocks are added in chunk order
 * (no atomics).  N1 % 4 == 0, N2 % 4 == 0, 16-byte aligned operands, else GNNRAG_E_UNSUPPORTED.
 * workspace: gnnrag_gemm_tn_workspace_bytes(M, N1, N2) bytes of device scratch. */
size_t gnnrag_gemm_tn_workspace_bytes(int64_t M, int32_t N1, int32_t N2);
int gnnrag_gemm_tn(const float* A, const float* B, int64_t M, int32_t N1, int32_t N2, float* C,
                   void* workspace, size_t workspace_bytes, gnnrag_stream_t stream);

/* Per-question relation tables of the fused path, one row per (question b, relation r used by b):
 *   P[d,row(b,r),:] = sum_i W_e2e[:, (1+2i+d)D : (2+2i+d)D] . relu(T_d[r,:] * ins[b,i,:])   [2,rel_total,D]
 * (the e2e_linear column blocks in the concat order of reasongnn.py:150-161).  The operand
 * relu(T_d * ins) is generated inside the GEMM's tile loader and never stored. */
int gnnrag_relation_tables(const gnnrag_csr* csr, const float* T_fwd, const float* T_inv, const float* ins,
                           const float* W_e2e, float* P, int32_t D, int32_t I, int32_t math,
                           gnnrag_stream_t stream);

/* h_out = relu(h . W_e2e[:, 0:D]^T + b + nbr), nbr [BN,D] from gnnrag_aggregate_fused; score as in
 * gnnrag_update_score.  Together: reasongnn.py:161-168. */
int gnnrag_update_score_fused(const float* h, const float* nbr, const float* W_e2e, const float* b,
                              const float* w_s, const float* b_s, const float* mask,
                              float* h_out, float* score, int64_t BN, int32_t D, int32_t I, int32_t math,
                              gnnrag_stream_t stream);

/* One whole ReasonGNNLayer.forward (reasongnn.py:134-174) enqueued with a single call:
 * rel transform (both directions) -> aggregation -> update+score -> softmax.
 * path: GNNRAG_PATH_UNFUSED = aggregate [BN,2I*D] then one [(2I+1)D -> D] GEMM (the reference's
 * operator boundaries); GNNRAG_PATH_FUSED = relation tables -> fused aggregation -> self-block
 * GEMM (2.3x fewer flops and 4x less aggregation traffic when 2*rel_total*I < ~0.8*B*N*2I);
 * GNNRAG_PATH_AUTO picks by that flop model.  Results agree to fp32 rounding (re-association).
 * pos_fwd/pos_inv: pos_emb{step}.weight / pos_emb_inv{step}.weight [pos_rows,D] or NULL.
 * workspace: gnnrag_layer_workspace_bytes() bytes of device scratch. */
#define GNNRAG_PATH_AUTO    0
#define GNNRAG_PATH_UNFUSED 1
#define GNNRAG_PATH_FUSED   2
/* OR-ed into `path`: a layer that aggregates along ONE direction (NSMLayer: head -> tail, NSMLayer_back: tail ->
 * head; the caller's e2e weight block of the other direction must be zero).  Where the kernels at hand can leave a
 * direction out (V-form tables + LDS walk) only that direction's tables are built and walked; elsewhere both run and
 * the zero block makes the other contribute exactly 0 - results are identical either way. */
#define GNNRAG_PATH_ONLY_FWD 0x10
#define GNNRAG_PATH_ONLY_INV 0x20
/* OR-ed into `path`: the caller states that `dist` (gnnrag_reason_stack: dist0, i.e. layer 0 only) is a SEED
 * distribution - the first layer of every ReaRev iteration (rearev.py:208 resets curr_dist to seed_dist).  fact_prior
 * (reasongnn.py:80,106) is then zero for every fact that does not start at a seed, so the fused path computes only the
 * relation-table rows the seeds' facts use and the neighbour sums of the nodes they reach (the frontier, derived from
 * `dist` on the device: gnnrag_frontier_build).  A hint, not a promise: any prior gives the same results as without
 * the flag (to rounding of the relation-table products), a dense one slowly.  Needs D % 4 == 0, D <= 256, both
 * directions, the fused path; ignored otherwise. */
#define GNNRAG_PATH_SEED_PRIOR 0x40
/* OR-ed into `path` of gnnrag_reason_stack (with a gnnrag_stack_workspace_bytes workspace): the workspace still holds
 * the relation projections (and their bf16 planes) that an EARLIER gnnrag_reason_stack call on the same workspace
 * computed for the same layers' parameters and the same relation features - they depend on nothing else
 * (reasongnn.py:75-79: rel_linear{j}(rel_features)), so the iterations 2..T of one ReaRev forward skip that launch.
 * The caller vouches for "same parameters, same relation features, workspace untouched in between". */
#define GNNRAG_PATH_REUSE_PROJ 0x80
size_t gnnrag_layer_workspace_bytes(const gnnrag_csr* csr, int32_t D, int32_t I);
int gnnrag_reason_layer(const gnnrag_csr* csr,
                        const float* h, const float* dist, const float* ins,
                        const float* relfeat_fwd, const float* relfeat_inv,
                        const float* W_rel, const float* b_rel,
                        const float* pos_fwd, const float* pos_inv, int32_t pos_rows,
                        const float* W_e2e, const float* b_e2e,
                        const float* w_score, const float* b_score, const float* mask,
                        float* h_out, float* score_out, float* dist_out,
                        void* workspace, size_t workspace_bytes,
                        int32_t D, int32_t I, int32_t path, int32_t math, gnnrag_stream_t stream);

/* L consecutive ReasonGNNLayer.forward calls on one batch - what one iteration of ReaRev.forward does
 * (rearev.py:208-210: `for j in range(num_gnn): curr_dist, global_rep = reasoning(curr_dist, relation_ins, step=j)`)
 * enqueued with ONE call: layer j reads h[j-1] / dist[j-1] (h0 / dist0 for j = 0) and writes h_out[j], score_out[j],
 * dist_out[j] ([L, B*N, D] / [L, B*N] / [L, B*N], every layer's outputs are kept: the reference returns each of them
 * to its caller).  `layers` is a HOST array of L parameter sets.  Same arithmetic, kernels and workspace as L calls
 * of gnnrag_reason_layer (bit-identical results); what it removes is the per-call host work in front of ~8 short
 * kernels per layer, which bounds small batches (one question per batch: SURVEY.md section 8 f-3). */
typedef struct gnnrag_layer_params {
  const float* W_rel;    /* rel_linear{j}.weight [D, D]                      */
  const float* b_rel;    /* rel_linear{j}.bias   [D]                         */
  const float* pos_fwd;  /* pos_emb{j}.weight [pos_rows, D] or NULL          */
  const float* pos_inv;  /* pos_emb_inv{j}.weight [pos_rows, D] or NULL      */
  const float* W_e2e;    /* e2e_linear{j}.weight [D, (2I+1) D]               */
  const float* b_e2e;    /* e2e_linear{j}.bias [D]                           */
} gnnrag_layer_params;
/* Workspace of gnnrag_reason_stack that lets it compute the relation projections of all L layers up front in ONE
 * launch (gnnrag_layer_workspace_bytes + an [L][2][R1][D] block).  With only gnnrag_layer_workspace_bytes the stack
 * call still works and projects per layer (same results bit for bit). */
size_t gnnrag_stack_workspace_bytes(const gnnrag_csr* csr, int32_t L, int32_t D, int32_t I);

/* T_out[j][d][r, :] = rel_linear{j}(rel_features_d[r, :]) (+ pos_emb{j}_d[r, :] for r < pos_rows), j < L, d = 0
 * forward / 1 inverse (reasongnn.py:75-79, :102-105): the relation projections of all layers in one launch, exact
 * fp32 on the matrix cores.  `layers` is a HOST array (only W_rel, b_rel, pos_fwd, pos_inv are read; pos_* are
 * ignored when pos_rows == 0).  D % 4 == 0, else GNNRAG_E_UNSUPPORTED.  T_out: [L, 2, R1, D] floats.
 * planes_out (may be NULL; D <= 224): gnnrag_rel_planes_bytes(R1, D, L) bytes, [L][2][3][R1][448] bf16 - per layer
 * and direction the three planes of the exact 3-way bf16 split of [relu(T[r, :]) | relu(-T[r, :])], each half padded
 * with zeros to 224 columns: the left operand of gnnrag_relation_tables_planes. */
size_t gnnrag_rel_planes_bytes(int64_t R1, int32_t D, int32_t L);
int gnnrag_rel_transform(const float* relfeat_fwd, const float* relfeat_inv, int64_t R1, int32_t D, int32_t L,
                         const gnnrag_layer_params* layers, int32_t pos_rows, float* T_out, void* planes_out,
                         gnnrag_stream_t stream);

/* gnnrag_relation_tables in the bf16x3 math mode from ONE layer's pre-split relation planes (gnnrag_rel_transform's
 * planes_out for that layer: [2][3][R1][448] bf16): since relu(t q) = max(q,0) relu(t) + max(-q,0) relu(-t),
 *   P[d,row(b,r),:] = [relu(T_d[r,:]), relu(-T_d[r,:])] . V_{b,d},
 *   V_{b,d} = [sum_i W_{i,d}^T diag(max(ins[b,i,:],0)) ; sum_i W_{i,d}^T diag(max(-ins[b,i,:],0))]   (W_{i,d} as above)
 * - the left operand is question independent and arrives as ready matrix-core fragments, the question sits in the
 * per-question right operand built in LDS.  Shapes: 193 <= D <= 208, D % 8 == 0, rel_total >= 1024; otherwise
 * GNNRAG_E_UNSUPPORTED (nothing launched; use gnnrag_relation_tables). */
int gnnrag_relation_tables_planes(const gnnrag_csr* csr, const void* planes, const float* ins, const float* W,
                                  float* P, int32_t D, int32_t I, gnnrag_stream_t stream);

int gnnrag_reason_stack(const gnnrag_csr* csr, int32_t L, const gnnrag_layer_params* layers,
                        const float* h0, const float* dist0, const float* ins,
                        const float* relfeat_fwd, const float* relfeat_inv, int32_t pos_rows,
                        const float* w_score, const float* b_score, const float* mask,
                        float* h_out, float* score_out, float* dist_out,
                        void* workspace, size_t workspace_bytes,
                        int32_t D, int32_t I, int32_t path, int32_t math, gnnrag_stream_t stream);

/* The same L-layer sequence captured as a hipGraph (stream capture on `stream`, which must not be capturing): every
 * pointer and size is baked in, so a replay repeats the sequence on whatever the buffers hold then.  One batch's
 * T iterations replay one graph when the caller keeps the buffers in place: h0 = h_out + (L-1)*B*N*D (the previous
 * iteration's last layer; rearev.py:208-211 carries the node state over), dist0 = the seed distribution, `ins`
 * rewritten in place between replays (rearev.py:217-221).  Results are bit-identical to the eager sequence.
 * Run the eager call once before capturing (launch attributes are raised on first use). */
typedef struct gnnrag_graph gnnrag_graph;
int gnnrag_reason_stack_capture(const gnnrag_csr* csr, int32_t L, const gnnrag_layer_params* layers,
                                const float* h0, const float* dist0, const float* ins,
                                const float* relfeat_fwd, const float* relfeat_inv, int32_t pos_rows,
                                const float* w_score, const float* b_score, const float* mask,
                                float* h_out, float* score_out, float* dist_out,
                                void* workspace, size_t workspace_bytes,
                                int32_t D, int32_t I, int32_t path, int32_t math, gnnrag_stream_t stream,
                                gnnrag_graph** out);
int gnnrag_graph_launch(gnnrag_graph* graph, gnnrag_stream_t stream);
int gnnrag_graph_destroy(gnnrag_graph* graph);

/* The frontier form of the fused layer for a sparse prior (see GNNRAG_PATH_SEED_PRIOR), exposed piecewise for tests:
 *   gnnrag_frontier_build: nodes with dist != 0 are the sources; marks the nodes their facts reach (row gates: one
 *     byte per node at workspace offset ...) and the compact relation rows those facts use, and lists both;
 *   gnnrag_relation_tables_frontier: the listed rows of P [2, rel_total, D], written in place (exact fp32 MFMA);
 *   gnnrag_aggregate_fused_frontier: out[n, :] = sum_d sum_f p_f P[d, row(b, rel_f), :] for the listed nodes n ONLY
 *     (facts with p_f = 0 are skipped: they add exact zeros); every other row of `out` is left untouched.
 * fws: gnnrag_frontier_workspace_bytes(csr) bytes of device scratch shared by the three calls.
 * gnnrag_frontier_read copies (rows listed, relation rows listed) to the host (synchronises the stream; tests). */
size_t gnnrag_frontier_workspace_bytes(const gnnrag_csr* csr);
int gnnrag_frontier_supported(const gnnrag_csr* csr, int32_t D);
int gnnrag_frontier_build(const gnnrag_csr* csr, const float* dist, void* fws, size_t fws_bytes, gnnrag_stream_t stream);
int gnnrag_relation_tables_frontier(const gnnrag_csr* csr, const void* fws, const float* T_fwd, const float* T_inv,
                                    const float* ins, const float* W_e2e, float* P, int32_t D, int32_t I,
                                    gnnrag_stream_t stream);
int gnnrag_aggregate_fused_frontier(const gnnrag_csr* csr, const void* fws, const float* dist, const float* P,
                                    float* out, int32_t D, gnnrag_stream_t stream);
int gnnrag_frontier_read(const gnnrag_csr* csr, const void* fws, int32_t* counts2, uint8_t* row_flag_host,
                         gnnrag_stream_t stream);

/* Candidate selection of Evaluator.evaluate (evaluate.py:188-207) + the sort and top-p cut of
 * f1_and_hits (evaluate.py:34-51), one workgroup per question:
 *   keep slot j iff eligible[b,j] (host: query_entities != 1 and local_entity != pad id) and
 *   (double)pred_dist[b,j] >= ignore_prob;  order: probability descending, ties by ascending slot;
 *   out_slot [B,N]: the kept slots in that order, then -1;  out_cnt [B,2]: (kept, retrieved) where
 *   retrieved = shortest prefix whose running fp64 sum exceeds eps (or all kept).  N <= 16384. */
int gnnrag_topp_candidates(const float* pred_dist, const uint8_t* eligible, int32_t B, int32_t N,
                           double ignore_prob, double eps, int32_t* out_slot, int32_t* out_cnt,
                           gnnrag_stream_t stream);
/* The same selection for ANY N: questions with more than 16384 node slots (BASELINE config 5: 20 000) filter first,
 * compact the survivors into `workspace` (gnnrag_topp_workspace_bytes(B, N) bytes of device scratch; 0 for N <= 16384)
 * and sort them there - in LDS when at most 16384 slots survive the filter, else in the workspace. */
size_t gnnrag_topp_workspace_bytes(int32_t B, int32_t N);
int gnnrag_topp_candidates_ws(const float* pred_dist, const uint8_t* eligible, int32_t B, int32_t N,
                              double ignore_prob, double eps, int32_t* out_slot, int32_t* out_cnt,
                              void* workspace, size_t workspace_bytes, gnnrag_stream_t stream);

/* out[b,:] = sum_n seed_info[b,n] * ent_emb[b,n,:]  - the seed retrieval of QueryReform.forward
 * (gnn/modules/query_update.py:40, torch.bmm over all N rows); only rows with a non-zero flag are
 * read, in ascending n.  seed_info [B,N], ent_emb [B,N,D], out [B,D]. */
int gnnrag_seed_retrieve(const float* seed_info, const float* ent_emb, float* out, int32_t B, int32_t N,
                         int32_t D, gnnrag_stream_t stream);

/* QueryReform.forward as ONE launch (gnn/modules/query_update.py:26-44; Fusion :6-16; called once per instruction between
 * two ReaRev iterations, gnn/models/ReaRev/rearev.py:217-221):
 *   y = seed retrieval as above;  feats = [x, y, x - y];  out = sigmoid(W_g feats) * (W_r feats) + (1 - sigmoid(..)) * x
 * with x = q_node [B, D], W_r / W_g = fusion.r.weight / fusion.g.weight [D, 3 D] (no bias), ent_emb [B, N, ld_ent]
 * (row stride ld_ent >= D: a zero-padded node state is read in place), out [B, D].  The reference's attention over all
 * N node states (:36-38) does not enter its return value and is not computed.  D <= 4096 (GNNRAG_E_UNSUPPORTED beyond).
 * (ABI 16) */
int gnnrag_query_reform(const float* q_node, const float* seed_info, const float* ent_emb, int64_t ld_ent,
                        const float* W_r, const float* W_g, float* out, int32_t B, int32_t N, int32_t D,
                        gnnrag_stream_t stream);

/* The question encoder's LSTM (the instruction path): one layer, one direction, batch_first,
 * torch.nn.LSTM semantics and parameter layout (gate order i, f, g, o) - what
 * gnn/modules/question_encoding/lstm_encoder.py:27-36 builds and calls as
 *   query_hidden_emb, (h_n, c_n) = self.node_encoder(x, (h0, c0))          x [B, T, E], h0 = c0 = zeros [1, B, H]
 * x [B,T,E], w_ih [4H,E], w_hh [4H,H], b_ih / b_hh [4H] or NULL, h0 / c0 [B,H] or NULL (zeros); out [B,T,H], h_n and
 * c_n [B,H].  4 H <= 1024.  workspace: gnnrag_lstm_workspace_bytes(E, H) bytes (transposed weights). */
size_t gnnrag_lstm_workspace_bytes(int32_t E, int32_t H);
int gnnrag_lstm_forward(const float* x, const float* w_ih, const float* w_hh, const float* b_ih, const float* b_hh,
                        const float* h0, const float* c0, float* out, float* h_n, float* c_n, int32_t B, int32_t T,
                        int32_t E, int32_t H, void* workspace, size_t workspace_bytes, gnnrag_stream_t stream);

/* Plain HBM copy kernel (float4 per lane) used by bench.py to measure the achievable
 * streaming ceiling next to the 8 TB/s spec.  n = number of floats (multiple of 4). */
int gnnrag_stream_copy(const float* src, float* dst, int64_t n, gnnrag_stream_t stream);

int gnnrag_abi_version(void);
const char* gnnrag_error_string(int code);

#ifdef __cplusplus
}
#endif
#endif /* GNNRAG_H_ */
