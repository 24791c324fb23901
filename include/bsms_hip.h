/* bsms_hip.h -- C ABI of libbsms_hip.so: the MI355X (gfx950) engine for the BSMS-GNN hot path.
 *
 * The reference (Eydcao/BSMS-GNN @ 2024_10_08) has no FFI layer: its hot path sits behind the
 * Python nn.Module API of src/ops + src/models.  This header is the boundary a maintainer would
 * bind instead (ctypes stub in INTEGRATION.md); each entry names the reference code it replaces
 * (paths relative to /root/reference/src).
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no torch types.  Every call returns 0 (BSMS_OK) or a
 *     negative bsms_status; bsms_last_error() gives a thread-local message.  Never throws/exits.
 *   - All tensor arguments are CALLER-OWNED DEVICE pointers: fp32, row-major, contiguous,
 *     16-byte aligned.  The library never allocates, frees or synchronises on the data path; the
 *     one exception is bsms_plan_create/destroy, which own the small integer index buffers of a
 *     mesh level (lifetime = plan).  Scratch ("work") and saved-for-backward ("saved") buffers are
 *     caller-provided; their sizes come from the *_bytes() queries.
 *   - Every launch goes to the hipStream_t passed as `stream` (void* here so C callers need no HIP
 *     headers).  Re-entrant and thread-safe per stream.  Global state: the thread-local error string and,
 *     per device, two lazily created internal side streams (+ two events each) that bsms_gmp_bwd uses to
 *     overlap its weight-gradient kernels with its gradient scatters (forked from and joined back into
 *     `stream` inside the call, also legal under HIP-graph capture).
 *     The FIRST call that needs a side stream creates it and checks once, with a few device fills on a temporary
 *     64 MB allocation (~1 ms), that it does not share its hardware queue with the default stream or the other side
 *     stream -- HIP places streams on four hardware queues and two streams on one queue run in order.
 *     If `stream` is being CAPTURED at that first call the check is skipped (it allocates and synchronises, which would
 *     invalidate the capture): the side stream is created plainly and may share a queue -- less overlap, same results.
 *     Run one eager call before capturing to get the checked streams.
 *   - Arithmetic: fp32 in, fp32 out, fp32 accumulation.  Matrix products run on the f16 matrix cores as three
 *     partial products of two-way fp16 splits (11 + 11 significand bits) of power-of-two-scaled fp32 operands.
 *     Forward and input-gradient products scale per activation ROW and per weight MATRIX: the result is at least as
 *     accurate as an fp32 fused-multiply-add chain and as v_mfma_f32_16x16x4_f32 over the whole fp32 range
 *     (chain.h; profiles/census/f16split.hip; tests/test_hip_parity.py).  The WEIGHT gradients reduce over rows, so a
 *     scale cannot vary by row.  NODE-level weight gradients (node MLP, the two projections of the first edge Linear,
 *     bsms_mlp_bwd -- every job whose operand may be a caller's tensor) use the RANGE-FREE arithmetic: the exact
 *     three-way bf16 split (8 + 8 + 8 bits, fp32's exponent range, no scale), six partial products: a feature column
 *     1e-7 of the tensor's maximum is as accurate as in fp32 arithmetic (test_weight_gradient_precision_per_column).
 *     EDGE-level weight gradients of the fp32 GMP (85 % of the weight-gradient work; all six range-free would cost 3.8 % of
 *     the step, profiles/r05_wgrad_bf3_ab.txt) keep fp16 x 2 pieces with one power-of-two scale per operand TENSOR from
 *     the magnitude bound the chain kernels record: an element within 2^-18 of its tensor's largest magnitude keeps all
 *     22 bits; below that the low piece is an fp16 subnormal and one bit is lost per octave.  Their operands are the edge
 *     MLP's own post-ReLU activations and layer gradients -- never a caller's tensor -- which span far less than 2^18 per
 *     tensor (measured at full size, tests/test_hip_fullsize.py: gradients as close to fp64 as the fp32 oracle's);
 *     test_edge_weight_gradient_envelope pins that envelope.
 *   - Edge lists follow the reference: g = int64 [2,E], g[0] = source i, g[1] = target j
 *     (ops/basic.py:66); aggregation target is j.  "Edge order" below = the caller's order of g.
 *   - `D` (latent width) must be a multiple of 32 for the MLP/GMP entries (MFMA tile width).
 *   - An MLP is `hidden` x (Linear,ReLU) + Linear (+LayerNorm, no affine, eps 1e-5)
 *     (ops/basic.py:6-23).  `params` is a HOST array of 2*(hidden+1) device pointers in state_dict
 *     order: seq.0.weight, seq.0.bias, seq.2.weight, seq.2.bias, ...  Weights are [out,in] row-major
 *     exactly as torch.nn.Linear stores them.  `grads` arrays have the same order and shapes and
 *     are OVERWRITTEN (not accumulated).
 */
#ifndef BSMS_HIP_H
#define BSMS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
  BSMS_OK = 0,
  BSMS_E_INVALID_ARG = -1,
  BSMS_E_SHAPE = -2,
  BSMS_E_UNSUPPORTED = -3,
  BSMS_E_HIP = -4
} bsms_status;

/* Precision of the tensors inside a GMP block (the *_p entries of the U-Net).  BSMS_F32 is the reference's arithmetic.
 * The other two are build extensions without a reference parity target (the reference has no mixed precision):
 * BSMS_BF16: the saved EDGE activations, the messages and the edge layer gradients are stored in HBM as bf16 (round to
 *   nearest even) and the D x D Linears of the edge MLP multiply bf16 operands (weights rounded once per call) with fp32
 *   accumulation; bias, ReLU, LayerNorm, the aggregation sums, every node-level tensor and kernel (projections, node
 *   MLP, transitions, skip connections), the encoder / decoder and all weight-gradient accumulators stay fp32.
 * BSMS_BF16_NODES (round 4): BSMS_BF16 plus the NODE MLP: its four Linears multiply bf16 operands (the rows [x, aggr] and
 *   the hidden activations rounded to bf16 as they enter a Linear, weights rounded once per call; fp32 accumulation,
 *   bias, ReLU, LayerNorm), its hidden activations are saved as bf16 and its layer gradients gN[1..H] are stored as bf16
 *   (one-product weight-gradient jobs); the block's input / output rows [B,N,D], the residual and skip additions, the
 *   projections, gN[0] and the weight gradient of the first node Linear (operands x, aggr in fp32) stay fp32. */
typedef enum { BSMS_F32 = 0, BSMS_BF16 = 1, BSMS_BF16_NODES = 2 } bsms_precision;

typedef struct bsms_plan bsms_plan_t; /* one mesh level: dst-sorted CSR + src-sorted transpose */
typedef void* bsms_stream_t;          /* hipStream_t */

int bsms_abi_version(void);           /* bumps when a signature changes */
const char* bsms_last_error(void);

/* ---------------------------------------------------------------- graph plan (host side) ----
 * Replaces the per-call index handling of utils/basic.py:312-343 (broadcast + scatter_add_) and
 * ops/basic.py:66-72,127-138 (x[:, i], x[:, j] gathers): the COO list is turned ONCE per mesh
 * into a destination-sorted CSR (stable, so each target sums its edges in the caller's edge
 * order, like a sequential scatter_add_) plus the source-sorted transpose used by the backward
 * gathers and the up-pass.  `coo_host` is a HOST pointer to int64 [2,E].  Indices must be
 * < 2^31.  bsms_plan_set_pool attaches the kept-node ids of the level (m_ids[l],
 * graph_wrappers/bsms_graph_wrapper.py:97-98; HOST int64 [Nk], ascending) for the fused
 * restrict / prolong kernels.
 * The index buffers of a plan are ONE device block, uploaded on a private stream (creating a plan does not wait for work
 * queued on the caller's streams).  bsms_plan_destroy / a second bsms_plan_set_pool hand the block to an internal pool
 * instead of hipFree (which would wait for the whole device): the caller guarantees that nothing using the plan is
 * still in flight, as for any buffer it owns.  bsms_plan_pool_trim releases the pooled blocks (waits for the device). */
int bsms_plan_create(const int64_t* coo_host, int64_t E, int64_t N, bsms_plan_t** out);
int bsms_plan_set_pool(bsms_plan_t* plan, const int64_t* ids_host, int64_t Nk);
/* Binds the DEVICE edge weights `ew` [E] (edge order; WeightedEdgeConv.cal_ew, ops/basic.py:142-167 -- mesh-static, the
 * reference recomputes them every forward, BSMS.py:73) to a plan with a pool: they are gathered once into the slot orders
 * of the two pooled transitions, and every later bsms_edge_conv(..., ew, ..., pooled = 1) / U-Net call that passes the
 * SAME pointer takes compact index + weight streams instead of four levels of dependent index loads.  The caller
 * guarantees that the content of `ew` is unchanged for as long as it passes that pointer; ew = NULL unbinds.  Results
 * are bit-identical to the unbound path.  A new bsms_plan_set_pool unbinds.  A plan already bound to ANOTHER non-null
 * pointer keeps that binding (captured HIP graphs and kernels still queued have the gathered copies baked in); the new
 * tensor then simply takes the unbound path.  To move a binding: bind NULL first, once nothing uses the old one.
 * Binding the pointer a plan is ALREADY bound to is a no-op (the copies are not gathered again): after changing the
 * content of `ew` in place, bind NULL and then `ew` again to refresh them.  The binding is by ADDRESS: the caller keeps
 * the bound tensor alive (and its address unrecycled) until it unbinds or destroys the plan;
 * bsms_plan_bound_edge_weights returns the pointer a plan is bound to (NULL: unbound) so that a host wrapper can tell
 * whether its bind call took effect and which tensor it has to keep alive. */
int bsms_plan_bind_edge_weights(bsms_plan_t* plan, const float* ew, bsms_stream_t stream);
const float* bsms_plan_bound_edge_weights(const bsms_plan_t* plan);
int bsms_plan_destroy(bsms_plan_t* plan);
int bsms_plan_pool_trim(void);
int64_t bsms_plan_num_nodes(const bsms_plan_t* plan);
int64_t bsms_plan_num_edges(const bsms_plan_t* plan);
int64_t bsms_plan_num_pooled(const bsms_plan_t* plan);  /* Nk, 0 if no pool attached */
int64_t bsms_plan_min_out_degree(const bsms_plan_t* plan);
int64_t bsms_plan_max_source(const bsms_plan_t* plan);  /* max(g[0]); degree() length-1, utils/basic.py:305 */
/* Block-diagonal union of `nparts` plans, built ON THE DEVICE from the parts' index blocks (one kernel per 16 parts on `stream`;
 * no host CSR build, no upload): the plan of a batch of DIFFERENT meshes -- the reference's variable-mesh path, PyG `Batch`
 * collation of datasets/base.py:325-349 consumed at models/model.py:194-200 -- from per-mesh plans that stay resident in HBM.
 * Equal, array for array, to bsms_plan_create on the offset-concatenated edge list + bsms_plan_set_pool on the offset kept ids
 * (all parts have pools, or none).  `ew_cat` (nullable): the DEVICE concatenation of the parts' bound edge-weight tensors, in part
 * order; every part must then be bound (bsms_plan_bind_edge_weights), the union is bound to `ew_cat` and its gathered weight
 * copies are the parts'.  `coo_out` (nullable): DEVICE int64 [2, E] that receives the union's edge list in the caller's edge
 * order; `ids_out` (nullable): DEVICE int64 [Nk], its kept ids (what PyG's Batch would have produced).  The arrays of the new
 * plan are complete in STREAM ORDER: use it on `stream`, or synchronise first. */
int bsms_plan_concat(const bsms_plan_t* const* parts, int nparts, const float* ew_cat, int64_t* coo_out, int64_t* ids_out,
                     bsms_stream_t stream, bsms_plan_t** out);
/* debug/test accessors: copy index arrays to HOST int32 buffers (sizes N+1, E, E, E). */
int bsms_plan_export(const bsms_plan_t* plan, int32_t* rowptr, int32_t* src_sorted,
                     int32_t* perm, int32_t* t_rowptr);
/* array `which` of a plan as int32 words on the HOST (host == NULL: only its length is returned; < 0: error):
 * 0 rowptr 1 src 2 dst 3 perm 4 t_rowptr 5 t_dst 6 t_eid 7 t_pos 8 ids 9 inv 10 k_rowptr 11 k_src 12 k_eid 13 p_rowptr 14 p_src
 * 15 p_eid 16 k_w 17 p_w (bit patterns of the gathered edge weights; length 0 while unbound).  Synchronises the device. */
int64_t bsms_plan_export_ex(const bsms_plan_t* plan, int which, int32_t* host);

/* ---------------------------------------------------------------- A1: edge aggregation ------
 * scatter_sum(src, index=g[1], dim=-2, dim_size=N)  (utils/basic.py:324-343, call site
 * ops/basic.py:94): out[b,n,:] = sum over edges e with g[1][e]==n of src[b,e,:], summed in edge
 * order.  plan_order=0: `src` rows are in the caller's edge order; 1: already dst-sorted
 * (plan order, what the fused GMP path produces).  The backward is the gather grad[b, g[1][e], :]
 * (autograd of scatter_add_), written in edge order. */
int bsms_segment_sum_fwd(const bsms_plan_t* plan, const float* src, int64_t B, int64_t D,
                         int plan_order, float* out, bsms_stream_t stream);
int bsms_segment_sum_bwd(const bsms_plan_t* plan, const float* grad_out, int64_t B, int64_t D,
                         float* grad_src, bsms_stream_t stream);
/* The aggregation of the BSMS_BF16 precision: `src_bf16` [B,E,D] bf16 in PLAN order (what the edge MLP of that precision
 * writes), fp32 sums in edge order, fp32 out [B,N,D].  D = 128 or 256. */
int bsms_segment_sum_bf16(const bsms_plan_t* plan, const void* src_bf16, int64_t B, int64_t D, float* out,
                          bsms_stream_t stream);

/* ---------------------------------------------------------------- A2+A6: cal_ew -------------
 * WeightedEdgeConv.cal_ew (ops/basic.py:142-167) incl. degree() (utils/basic.py:287-309):
 * ec[e] = (w[i]/deg[i]) / (sum_{e'->j} w[i']/deg[i'] + 1e-12), aggr_w[n] = that sum + 1e-12.
 * `w` [N], `ec` [E] in edge order, `aggr_w` [N]. */
int bsms_cal_ew(const bsms_plan_t* plan, const float* w, float* ec, float* aggr_w,
                bsms_stream_t stream);

/* ---------------------------------------------------------------- A5/A7/A8: transitions -----
 * WeightedEdgeConv.forward (ops/basic.py:107-140), optionally fused with the pooling gather
 * h[:, m_ids] (ops/BSMS.py:79-89) or with Unpool (ops/basic.py:176-201):
 *   aggregating=1, pooled=0: out[b,j,:] = sum_{e->j} ew[e]*x[b,i_e,:]          x,out [B,N,D]
 *   aggregating=1, pooled=1: only kept rows j = ids[k] ("restrict")       x [B,N,D], out [B,Nk,D]
 *   aggregating=0, pooled=0: out[b,i,:] = sum_{e: i_e=i} ew[e]*x[b,j_e,:]      x,out [B,N,D]
 *   aggregating=0, pooled=1: x is the coarse tensor [B,Nk,D], zero-filled unpooling is implied
 *                            ("prolong"): out[b,i,:] = sum_{e: i_e=i, j_e kept} ew[e]*x[b,inv[j_e],:]
 * Any D >= 1 (also used for positions, D = pos_dim).  `ew` [E] in edge order.  The backward of a
 * call w.r.t. x is the same entry with `aggregating` flipped (exact adjoint), same `pooled`. */
int bsms_edge_conv(const bsms_plan_t* plan, const float* x, int64_t B, int64_t D,
                   const float* ew, int aggregating, int pooled, float* out, bsms_stream_t stream);
/* standalone Unpool / pooling gather with a DEVICE int64 index (ops/basic.py:194-199, BSMS.py:79-83) */
int bsms_scatter_rows(const float* h, int64_t B, int64_t Nk, int64_t D, const int64_t* idx_dev,
                      int64_t N, float* out /* [B,N,D], zero-filled here */, bsms_stream_t stream);
int bsms_gather_rows(const float* x, int64_t B, int64_t N, int64_t D, const int64_t* idx_dev,
                     int64_t Nk, float* out /* [B,Nk,D] */, bsms_stream_t stream);

/* ---------------------------------------------------------------- A3: MLP -------------------
 * MLP.forward (ops/basic.py:6-23) over R rows: x [R,in_dim] -> y [R,out_dim].  Used for the
 * encoder (in_dim = out_dim_model+1, LN) and decoder (out_dim = C, no LN) of
 * models/model.py:20-22.  `saved` = NULL selects INFERENCE (nothing is kept for a backward; `work`
 * must then be non-NULL).  Supported shapes: (in_dim <= 8 or in_dim == D) and
 * (out_dim == D with layer_norm, or out_dim <= 8 without).  `saved` keeps the activations the backward
 * needs.  need_dx=0 skips the input gradient (encoder input is data). */
size_t bsms_mlp_saved_bytes(int64_t R, int64_t in_dim, int64_t D, int64_t out_dim, int hidden);
size_t bsms_mlp_work_bytes(int64_t R, int64_t in_dim, int64_t D, int64_t out_dim, int hidden);
int bsms_mlp_fwd(const float* x, int64_t R, int64_t in_dim, int64_t D, int64_t out_dim, int hidden,
                 int layer_norm, const float* const* params, float* y, void* saved, void* work,
                 bsms_stream_t stream);
int bsms_mlp_bwd(const float* x, const float* grad_y, int64_t R, int64_t in_dim, int64_t D,
                 int64_t out_dim, int hidden, int layer_norm, const float* const* params,
                 const void* saved, void* work, float* grad_x /* nullable */,
                 float* const* grads, bsms_stream_t stream);
/* bsms_mlp_fwd with `flags`.  BSMS_MLP_REUSE_PACKS (inference only, saved = NULL): the weight packs written into `work` by
 * the previous bsms_mlp_fwd / _ex call with the same shape and the same parameter VALUES are still there (a private
 * `work` buffer of an autoregressive caller: utils/rollout_utils.py:49-62 applies the same encoder / decoder every
 * step) -- the prepack launches are skipped. */
enum { BSMS_MLP_REUSE_PACKS = 1 };
int bsms_mlp_fwd_ex(const float* x, int64_t R, int64_t in_dim, int64_t D, int64_t out_dim, int hidden,
                    int layer_norm, const float* const* params, float* y, void* saved, void* work,
                    int flags, bsms_stream_t stream);

/* ---------------------------------------------------------------- A4: GMP block -------------
 * GMP.forward (ops/basic.py:48-98) incl. both MLPs, the gathers, the fiber [pos_i-pos_j, |.|]
 * and the aggregation: out = mlp_node([x, scatter_sum(mlp_edge([fiber, x_i, x_j]), j)]) + x.
 * x,out [B,N,D]; pos [B,N,p] (pos_batch_stride = N*p) or [N,p] (pos_batch_stride = 0, the
 * `repeat` branch ops/basic.py:87-88); 1 <= p <= 7.  `params`: 2*(hidden+1) pointers of mlp_node
 * followed by 2*(hidden+1) of mlp_edge (state_dict order of a GMP module).  pos gets no gradient
 * (SURVEY.md quirk 5).  `saved` = NULL in bsms_gmp_fwd selects INFERENCE (rollout, utils/rollout_utils.py:14-64):
 * no activation is written for a backward, the messages live in `work`. */
size_t bsms_gmp_saved_bytes(int64_t B, int64_t N, int64_t E, int64_t D, int hidden);
size_t bsms_gmp_work_bytes(int64_t B, int64_t N, int64_t E, int64_t D, int hidden);
int bsms_gmp_fwd(const bsms_plan_t* plan, const float* x, const float* pos, int64_t B, int64_t D,
                 int64_t p, int64_t pos_batch_stride, int hidden, const float* const* params,
                 float* out, void* saved, void* work, bsms_stream_t stream);
int bsms_gmp_bwd(const bsms_plan_t* plan, const float* x, const float* pos, const float* grad_out,
                 int64_t B, int64_t D, int64_t p, int64_t pos_batch_stride, int hidden,
                 const float* const* params, const void* saved, void* work, float* grad_x,
                 float* const* grads, bsms_stream_t stream);

/* ---------------------------------------------------------------- A9: BSGMP (whole U-Net) ---
 * BSGMP.forward (ops/BSMS.py:39-104) in one call: down blocks + restrict, bottom block, prolong + up blocks + skip
 * connections.  `plans`: L+1 HOST pointers, levels 0..L; levels < L have their pool attached (bsms_plan_set_pool with
 * m_ids[l]) and plans[l]->Nk == nodes of level l+1.  `ew`: L HOST pointers to the DEVICE edge weights of levels 0..L-1
 * (cal_ew chain, BSMS.py:64,73,89 -- mesh-static, the caller caches them).  h,out [B,N_0,D]; pos [B,N_0,p]
 * (pos_batch_stride = N_0*p) or [N_0,p] (0).  `params`/`grads`: HOST arrays of (2L+1) x 4 (hidden+1) device
 * pointers, blocks in the order down_gmps[0..L-1], bottom_gmp, up_gmps[0..L-1] (up_gmps[i] acts on level L-1-i,
 * BSMS.py:96-101), each block laid out as for bsms_gmp_fwd.  `saved` = NULL selects inference. */
size_t bsms_bsgmp_saved_bytes(const bsms_plan_t* const* plans, int L, int64_t B, int64_t D, int64_t p, int hidden);
size_t bsms_bsgmp_work_bytes(const bsms_plan_t* const* plans, int L, int64_t B, int64_t D, int64_t p, int hidden);
/* `work` size for callers that only ever run INFERENCE forwards (saved == NULL; rollout_utils.py:14-64) with this buffer:
 * the forward's part of the layout without the backward's per-block scratch sets (airfoil batch 8: 1.4 GB against 4.7 GB).
 * A buffer of bsms_bsgmp_work_bytes serves inference calls too. */
size_t bsms_bsgmp_infer_work_bytes(const bsms_plan_t* const* plans, int L, int64_t B, int64_t D, int64_t p, int hidden);
int bsms_bsgmp_fwd(const bsms_plan_t* const* plans, const float* const* ew, int L, const float* h, const float* pos,
                   int64_t B, int64_t D, int64_t p, int64_t pos_batch_stride, int hidden,
                   const float* const* params, float* out, void* saved, void* work, bsms_stream_t stream);
/* As bsms_bsgmp_fwd, with `reuse` for INFERENCE calls (saved == NULL) that pass the same `work` buffer as their previous
 * call and let nothing else write to it: bit 0 = the weights are unchanged (skip the weight prepacks), bit 1 = pos and
 * the mesh are unchanged (skip the coarse positions).  The autoregressive rollout (utils/rollout_utils.py:49-62: fixed
 * weights, fixed mesh_pos) sets both from its second step on. */
int bsms_bsgmp_fwd_ex(const bsms_plan_t* const* plans, const float* const* ew, int L, const float* h, const float* pos,
                      int64_t B, int64_t D, int64_t p, int64_t pos_batch_stride, int hidden,
                      const float* const* params, float* out, void* saved, void* work, int reuse, bsms_stream_t stream);
/* The U-Net with a choice of precision (see bsms_precision): sizes, forward (+ reuse flags) and backward. */
size_t bsms_bsgmp_saved_bytes_p(const bsms_plan_t* const* plans, int L, int64_t B, int64_t D, int64_t p, int hidden, int precision);
int bsms_bsgmp_fwd_p(const bsms_plan_t* const* plans, const float* const* ew, int L, const float* h, const float* pos,
                     int64_t B, int64_t D, int64_t p, int64_t pos_batch_stride, int hidden, const float* const* params,
                     float* out, void* saved, void* work, int reuse, int precision, bsms_stream_t stream);
int bsms_bsgmp_bwd_p(const bsms_plan_t* const* plans, const float* const* ew, int L, const float* h, const float* pos,
                     const float* grad_out, int64_t B, int64_t D, int64_t p, int64_t pos_batch_stride, int hidden,
                     const float* const* params, const void* saved, void* work, float* grad_h, float* const* grads,
                     int precision, bsms_stream_t stream);
int bsms_bsgmp_bwd(const bsms_plan_t* const* plans, const float* const* ew, int L, const float* h, const float* pos,
                   const float* grad_out, int64_t B, int64_t D, int64_t p, int64_t pos_batch_stride, int hidden,
                   const float* const* params, const void* saved, void* work, float* grad_h,
                   float* const* grads, bsms_stream_t stream);
/* bsms_bsgmp_bwd_p with `flags`.  BSMS_BWD_DEFER_JOIN: the weight gradients of the last blocks may still be running on
 * the engine's internal side streams when the call returns; `grad_h` is complete in stream order.  The caller may enqueue
 * work that touches neither `work` nor `grads` (the fused training step runs the encoder's backward there, with its own
 * scratch) and MUST call bsms_side_lanes_join(stream) before anything reads `grads`, reuses `work`, or the step ends. */
enum { BSMS_BWD_DEFER_JOIN = 1 };
int bsms_bsgmp_bwd_ex(const bsms_plan_t* const* plans, const float* const* ew, int L, const float* h, const float* pos,
                      const float* grad_out, int64_t B, int64_t D, int64_t p, int64_t pos_batch_stride, int hidden,
                      const float* const* params, const void* saved, void* work, float* grad_h, float* const* grads,
                      int precision, int flags, bsms_stream_t stream);
/* bsms_bsgmp_bwd_ex with a hand-off for data-parallel callers that all-reduce their gradients in buckets WHILE the backward is
 * still running (the reference never got there: trainer/trainer.py:15-18 wraps nn.DataParallel and
 * train.py:16 disables it).  `block_done_events`: nullable HOST array of 2L+1 hipEvent_t (entries may be NULL), indexed by the
 * position of a block in the backward's EXECUTION order -- up_gmps[L-1] .. up_gmps[0] (levels 0 .. L-1), bottom_gmp,
 * down_gmps[L-1] .. down_gmps[0].  Event e is recorded on an internal side stream at the point where every weight gradient
 * of blocks 0..e -- and of a bsms_mlp_bwd_ex(BSMS_BWD_DEFER_JOIN) issued before this call -- has been written to `grads`:
 * a communication stream that waits for it may read those slots.  The events are the caller's (created without timing). */
int bsms_bsgmp_bwd_ev(const bsms_plan_t* const* plans, const float* const* ew, int L, const float* h, const float* pos,
                      const float* grad_out, int64_t B, int64_t D, int64_t p, int64_t pos_batch_stride, int hidden,
                      const float* const* params, const void* saved, void* work, float* grad_h, float* const* grads,
                      int precision, int flags, void* const* block_done_events, bsms_stream_t stream);
int bsms_side_lanes_join(bsms_stream_t stream);
/* Do two streams overlap?  HIP places its streams on a few hardware queues (four by default, by reference counts at creation time) and
 * two streams on one queue run in order, whatever their flags.  Returns 1 if work queued on `b` can overtake work queued on `a`, 0 if not
 * (or a == b), < 0 on error.  Probes with ~250 us of device fills on `a` and a small one on `b` over a temporary 64 MB allocation:
 * synchronises, not for the data path or graph capture.  The engine uses it for its own side streams; the host loader uses it to pick a
 * copy stream that really runs beside the compute stream (graph.py: _to_device_async; no counterpart in the reference, whose loader
 * copies on the compute stream, src/datasets/base.py via trainer/trainer.py:52-56). */
int bsms_streams_overlap(bsms_stream_t a, bsms_stream_t b);
/* bsms_mlp_bwd with `flags`.  BSMS_BWD_DEFER_JOIN: `grad_x` is complete in stream order when the call returns, the weight
 * gradients run on an internal side stream; same contract as above (`work`, `grads`, bsms_side_lanes_join). */
int bsms_mlp_bwd_ex(const float* x, const float* grad_y, int64_t R, int64_t in_dim, int64_t D, int64_t out_dim, int H,
                    int layer_norm, const float* const* params, const void* saved, void* work, float* grad_x,
                    float* const* grads, int flags, bsms_stream_t stream);


/* ---------------------------------------------------------------- A10-A12: model glue + loss -
 * BSMS_Simulator._forward (models/model.py:127-164) around encode / process / decode, the Normalizer arithmetic
 * (utils/normalizer.py:40-52,80-90: fp64, cast to fp32) and the masked RMSE (trainer/trainer.py:96-97), fused:
 *   bsms_sim_prologue  node_in [R, C+p+1] = [state(C) | mesh_pos(p) | node_type]  ->  norm_in [R, C+1] = normalised
 *                      [state | node_type] (model.py:43-46,153), pos [R,p] contiguous (model.py:62)
 *   bsms_sim_epilogue  norm_pred [R,C] (decoder output) -> pred = state + float(double(norm_pred) * std + mean) * mask
 *                      (model.py:160-163).  Optional: `sums` (device float[2]) <- (sum se*mask, sum mask) of the loss
 *                      against `target`; `next_in` [R, C+p+1] <- the next autoregressive input
 *                      where(mask == 0, ic, cat[pred, mesh_pos | type]) (utils/rollout_utils.py:57-62; may alias node_in).
 *   bsms_sim_loss_bwd  loss = sqrt(S / M / C) from `sums` (device; under data parallelism the caller all-reduces them
 *                      first, so the loss is the exact global one) and d loss / d norm_pred.
 * `mean`, `meansq`, `std_eps` are the DEVICE fp64 fields _E_data, _E_data_squared, std_eps of the reference's
 * Normalizer (state_dict layout); std = max(nan_to_num(sqrt(meansq - mean^2)), std_eps).  mask is [R] (the [B,N,1]
 * tensor of the reference, flat).  C <= 8.  Nothing here synchronises or reads device memory on the host. */
size_t bsms_sim_work_bytes(int64_t R);
int bsms_sim_prologue(const float* node_in, int64_t R, int64_t C, int64_t p, const double* mean, const double* meansq,
                      const double* std_eps, float* norm_in, float* pos, bsms_stream_t stream);
int bsms_sim_epilogue(const float* norm_pred, const float* node_in, const float* mask, const float* target /* nullable */,
                      int64_t R, int64_t C, int64_t p, const double* mean, const double* meansq, const double* std_eps,
                      float* pred, float* next_in /* nullable */, const float* ic /* nullable */, float* sums /* nullable */,
                      void* work, bsms_stream_t stream);
int bsms_sim_loss_bwd(const float* pred, const float* target, const float* mask, int64_t R, int64_t C, const double* mean,
                      const double* meansq, const double* std_eps, const float* sums, float* loss_out /* nullable */,
                      float* grad_norm_pred, bsms_stream_t stream);

/* ---------------------------------------------------------------- hierarchy builder (host) ---
 * BistrideMultiLayerGraph (graph_wrappers/bsms_graph_wrapper.py:8-154 + graph_wrapper.py:67-134): the
 * bi-stride multi-level hierarchy of a mesh, built natively on the HOST (no GPU needed, no SciPy/MKL).
 * Inputs are HOST pointers: coo int64 [2,E] (level-0 flat edges), pos [N,pos_dim] in fp64 (bsms_hierarchy_create) or
 * fp32 (bsms_hierarchy_create_f32).  The seed of a cluster is the node nearest its centroid
 * (bsms_graph_wrapper.py:118-124) and the reference evaluates that in the dtype of `pos_mesh` (datasets/base.py hands
 * it float32 mesh_pos): the two entries do the arithmetic in fp64 / fp32 respectively, in NumPy's evaluation order, so
 * the argmin -- hence m_ids -- is bit-exact for either dtype.  Level l has
 * level_nodes(l) nodes and level_edges(l) directed edges; copy_edges writes int64 [2,E_l] (level 0: the
 * caller's edges unchanged; coarser levels row-major with sorted columns), copy_ids writes the kept node
 * ids of level l (ascending, relative to level l; `m_ids[l]`), bit-exact w.r.t. the reference. */
typedef struct bsms_hierarchy bsms_hierarchy_t;
int bsms_hierarchy_create(const int64_t* coo_host, int64_t E, int64_t N, const double* pos_host,
                          int64_t pos_dim, int num_layers, bsms_hierarchy_t** out);
int bsms_hierarchy_create_f32(const int64_t* coo_host, int64_t E, int64_t N, const float* pos_host,
                              int64_t pos_dim, int num_layers, bsms_hierarchy_t** out);
int bsms_hierarchy_destroy(bsms_hierarchy_t* h);
int64_t bsms_hierarchy_level_nodes(const bsms_hierarchy_t* h, int level);
int64_t bsms_hierarchy_level_edges(const bsms_hierarchy_t* h, int level);
int bsms_hierarchy_copy_edges(const bsms_hierarchy_t* h, int level, int64_t* out);
int bsms_hierarchy_copy_ids(const bsms_hierarchy_t* h, int level, int64_t* out);

/* ---------------------------------------------------------------- optimizer step ------------
 * torch.nn.utils.clip_grad_norm_(params, max_grad_norm) + torch.optim.AdamW.step()
 * (trainer/trainer.py:150-152) fused over ONE flat fp32 array of all trainable parameters (the
 * data-parallel gradient buffer has the same layout).  `step` counts from 1 (bias correction);
 * max_grad_norm <= 0 disables clipping; grad_norm_out (device scalar, nullable) receives the
 * pre-clip global norm.  Decoupled weight decay and bias-corrected moments exactly as torch.optim.AdamW. */
size_t bsms_adamw_work_bytes(void);
int bsms_adamw_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n,
                    float lr, float beta1, float beta2, float eps, float weight_decay, int64_t step,
                    float max_grad_norm, float* grad_norm_out, void* work, bsms_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* BSMS_HIP_H */
