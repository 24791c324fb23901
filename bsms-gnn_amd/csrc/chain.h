// MLP "chain" kernels for gfx950: the dense per-edge / per-node MLPs of the BSMS path on the matrix cores.
//
// Orientation.  Every Linear y = x W^T is computed TRANSPOSED, Y^T = W X^T: the A operand of an MFMA is a weight
// fragment (rows = output features), the B operand an activation fragment (columns = rows of x, i.e. edges /
// nodes).  Each wave owns 16 rows of x: lane l <-> row (l & 15), lane group g = l >> 4.  The 16x16 result block
// leaves a lane holding, for ITS row, output features  f = 16 t + 4 g + r  (r = accumulator register 0..3,
// t = 16-feature block).  An MFMA sums over K slots (g, i) supplied by the four lane groups; we are free to decide
// WHICH input feature each slot stands for, as long as A and B agree.  Choosing, for the 32-feature K block kb2,
//          feature(g, i) = 16 (2 kb2 + (i >> 2)) + 4 g + (i & 3),      i = 0..7
// makes the accumulator layout of one layer exactly the B-operand layout of the next: activations never leave
// registers between layers -- no LDS round trip, no HBM traffic.  LayerNorm over a row is 32 in-lane values + two
// cross-group exchanges.
//
// Arithmetic: fp32 results from v_mfma_f32_16x16x32_f16 by TWO-WAY fp16 splitting with power-of-two scaling.
// A value is written as  x * 2^k = h + l,  h = fp16(x 2^k), l = fp16(x 2^k - h)  (round to nearest: 11 + 11 significand
// bits, sign included in each piece); k is chosen per ROW of the activation tile (row maximum -> [2^12, 2^13), so h never
// overflows and l stays a normal fp16 for every element within 2^15 of the row maximum) and per weight MATRIX.  A product
// is accumulated in fp32 as the three partial products
//          h_w * l_x  +  h_w * h_x  +  l_w * h_x;
// the dropped l_w * l_x is <= 2^-22 |x w|, and the result is un-scaled by the exact power of two 2^-(k_x + k_w).
// Measured against an fp64 product (profiles/census/f16split.hip, profiles/r03_f16split.md) this is MORE accurate than
// the exact three-way bf16 split of rounds 1-2 (six products, truncated pieces), than v_mfma_f32_16x16x4_f32 and than a
// sequential fp32 dot product, for activations, gradients of magnitude 1e-7, rows mixing 1e4 / 1 / 1e-6 and rows of 1e-30
// or 1e30 alike -- at half the matrix instructions (3 instead of 6 per fragment pair), 2/3 of the weight bytes and 4
// instead of 9-11 VALU operations per pair of elements (v_fma_mix{lo,hi}_f16 folds scaling, subtraction and rounding).
// Inputs, outputs, accumulation, bias, LayerNorm and everything stored to HBM stay fp32.
//
// Weights are re-laid out once per call ("prepack") into chunks, one per 32-feature K block:
//          [1 KB header]  [t][plane h|l][lane] 16 bytes = the 8 fp16 of A-fragment (t, kb2) for that lane
// (17 KB per chunk at D = 128).  Header of chunk 0: float 255 = 2^-k_w (the un-scaling factor of the matrix); header of
// the LAST chunk: the Linear's bias (needed when the stage finishes).  A LOADER wave streams chunks L2 -> LDS with
// LDS-DMA (global_load_lds_dwordx4, no registers, two chunks in flight) through a 3-deep ring, one workgroup barrier per
// chunk; compute waves read their A fragments with conflict-free lane-linear ds_read_b128.
// (The bf16 PRECISION of the edge MLP -- bsms_precision -- keeps one bf16 plane per weight and its bias in chunk 0.)
#pragma once
#include "common.h"

namespace bsms {

constexpr int kMaxStages = 8;   // max Linear layers per MLP handled by one chain launch
constexpr int kComputeWaves = 4;                       // compute waves per workgroup (each owns 16 rows)
constexpr int kTileRows = kComputeWaves * 16;          // rows of x per workgroup
constexpr int kChainThreads = (kComputeWaves + 1) * 64;  // + 1 loader wave
constexpr int kChainMaxThreads = 8 * 64;               // generic chain kernels: up to 7 compute waves (launcher's choice)
constexpr int kChunkHdrFloats = 256;                     // 1 KB chunk header (bias)
// Precision of the EDGE-level tensors of a GMP block (include/bsms_hip.h: bsms_precision).  BF16: the saved edge
// activations, the messages y and the edge layer gradients are stored as bf16 and the edge MLP's products take bf16
// operands (weights rounded once per call) with fp32 accumulation; everything at node level stays fp32.
// A saved activation [rows, D] is followed by its ReLU sign bits, one bit per element packed per (row, lane group):
// the backward chain masks gradients from 16 bytes per row instead of re-reading 4 D bytes.
constexpr size_t mask_words_per_row(int64_t D) { return size_t(4) * (D <= 128 ? 1 : D / 128); }
// Tensors the chain kernels write with their paired streaming stores are allocated for whole tiles (the largest tile:
// kPadRows rows, the edge kernels with two row blocks per wave): the stores then need no per-lane bounds test (rows
// past R hold garbage nobody reads; the sign bits start after the padding and are padded the same way).
constexpr int kPadRows = 2 * kTileRows;
constexpr size_t pad_rows(size_t rows) { return (rows + kPadRows - 1) / kPadRows * kPadRows + 2 * kPadRows; }   // + 256 rows: covers the last tile of any tile size up to 7 waves x 32 rows
constexpr size_t act_floats(size_t rows, int64_t D) { return pad_rows(rows) * (size_t(D) + mask_words_per_row(D)); }
// row pitch (floats) of the saved fiber tensor: p + 1 values padded to one or two 16-byte pieces
constexpr int fiber_ld(int64_t p) { return p + 1 <= 4 ? 4 : 8; }
constexpr int kScaleSlot = 255;                          // header float of chunk 0 that holds 2^-k_w
// floats in one weight pack of a D x D Linear (fp16 x 2 planes + headers)
constexpr size_t pack_floats(int64_t D) { return size_t(D / 32) * (kChunkHdrFloats + size_t(D / 16) * 512); }

enum ChainIn { IN_ROWS = 0, IN_ROWS2 = 1, IN_SMALL = 2, IN_EDGE = 3 };
enum ChainOut { OUT_LN = 0, OUT_PLAIN = 1, OUT_SMALL = 2, OUT_PLAIN2 = 3 };  // PLAIN2: stage 0 -> y, stage 1 -> y2, same input
enum GradIn { G_ROWS_LN = 0, G_EDGE_LN = 1, G_SMALL = 2 };
enum GradFirst { F_NONE = 0, F_HEADS1 = 1, F_HEADS2 = 2 };

struct ChainFwdArgs {
  int64_t R;  // rows
  int ntiles; // filled by the launcher: ceil(R / kTileRows); workgroups stride over tiles (persistent)
  // ---- input stage
  const float* x;    // IN_ROWS/IN_ROWS2: [R,D]; IN_SMALL: [R,K0]
  const float* x2;   // IN_ROWS2: second source [R,D]
  int K0;            // IN_SMALL: input width; IN_EDGE: p+1
  const float* w0t;  // IN_SMALL: W0^T [K0][D]; IN_EDGE: fiber weights^T [p+1][D]
  const float* bias_in;  // IN_SMALL: b0 [D]
  float* store_in;   // IN_SMALL/IN_EDGE: activation after the input stage, act_floats(R, D) floats (nullable)
  const int32_t *src, *dst;  // IN_EDGE: plan-order endpoints
  int32_t E, N;
  const float *Ps, *Pd;      // IN_EDGE: per-node pre-projections [B*N, D]
  const float* pos;          // IN_EDGE
  int64_t pos_bstride;
  int p;
  float* fiber_out;          // IN_EDGE: [R, fiber_ld(p)] the fiber [pos_i - pos_j, |pos_i - pos_j|, 0...] of every edge row, kept
                             // for the backward's narrow weight gradient (nullable)
  // ---- MFMA stages
  int nstage;
  const float4* wp[kMaxStages];  // packs (the Linear's bias travels in the pack header, see PackDesc)
  const float4* wp0b;  // IN_ROWS2: pack for x2 in stage 0 (same scale as wp[0]: PackDesc::mate; the stage's bias rides in THIS pack)
  float* store[kMaxStages];  // post-ReLU activation of stage l, act_floats(R, D) floats: values + sign bits (nullable)
  int pieces;                // IN_EDGE, D = 128 (round 6): store_in / store[] receive the fp16 x 2 PIECES of the rows instead of fp32 values
                             // (row r: K block c at byte 512 r + 128 c = 64 bytes of h pieces, 64 of l pieces, in the lane order of the B
                             // operand; same size, sign bits in the same place) and store_exp[] the rows' scale exponents (RowScale::E)
  int* store_exp[kMaxStages];  // pieces: [0] for store_in, [l + 1] for store[l]; pad_rows(R) ints each
  // ---- output
  float* y;           // OUT_LN / OUT_PLAIN: [R,D]; OUT_SMALL: [R,C]
  float* y2;          // OUT_PLAIN2: second head [R,D]
  float* yln;         // OUT_LN: normalised output before the residual (nullable)
  float* rstd;        // OUT_LN: [R] (nullable)
  const float* resid; // OUT_LN: residual rows added to y (nullable)
  const float* resid2; // OUT_LN: second residual (the U-Net skip connection, ops/BSMS.py:102), added after `resid` (nullable)
  int accumulate;     // OUT_PLAIN: y += result
  const float* wout;  // OUT_SMALL: [C][D]
  const float* bout;  // OUT_SMALL: [C]
  int C;
  unsigned long long* timing;  // experiments only: per-workgroup s_memtime stamps (16 slots), null in production
  int bf16;           // bf16 precision (IN_EDGE / OUT_LN only): bf16 MFMA operands, saved activations and y stored as bf16
  int store_mode;     // saved-activation stores: 0 plain, 1 non-temporal (keeps L2 for weights / gathered rows)
  int out_mode;       // same for the final output y
  // ---- filled by launch_chain_fwd: weight packs in execution order for the loader wave(s)
  int nseq;
  const float4* wseq[kMaxStages + 2];
  int nload;          // loader waves per workgroup (the last `nload` waves of the block)
  int nring;          // depth of the LDS weight ring of this launch (3..6)
  // ---- magnitude bounds for the weight-gradient kernel (wgrad.hip): bound slots (kBoundWidth floats each, see above)
  // for the tensor ENTERING stage l -- amax[0]: x / [x, x2] / the activation of the narrow or edge input stage;
  // amax[l + 1]: the post-ReLU activation of stage l.  Nullable.
  float* amax[kMaxStages + 1];
};

struct ChainBwdArgs {
  int64_t R;
  int ntiles;         // filled by the launcher (persistent workgroups stride over tiles)
  const float* dy;    // G_ROWS_LN: [R,D]; G_EDGE_LN: [B*N,D] node gradient gathered by dst; G_SMALL: [R,C]
  const float* yln;   // LN cases: normalised forward output [R,D]
  const float* rstd;  // LN cases: [R]
  const int32_t* dst; // G_EDGE_LN
  int32_t E, N;
  const float* wout;  // G_SMALL: [C][D]
  int C;
  const float* mask_in;  // G_SMALL: activation masking the VALU-produced gradient
  int nstage;
  const float4* wpt[kMaxStages];  // transposed packs, execution order (top layer first)
  const float* mask[kMaxStages];  // saved activation (act_floats layout) whose sign BITS mask the output of stage k
  float* gstore[kMaxStages + 1];  // [0]: gradient entering stage 0; [k+1]: output of stage k (nullable)
  // ---- first-layer input gradient
  const float4 *wh0, *wh1;
  float *dx, *dx2;
  const float* dres;  // added to dx (residual branch), nullable
  int store_mode;     // layer-gradient stores: 0 plain, 1 non-temporal
  int bf16;           // bf16 precision (G_EDGE_LN / F_NONE only): yln is bf16, layer gradients are stored as bf16
  // ---- filled by launch_chain_bwd: weight packs in execution order for the loader wave(s)
  int nseq;
  const float4* wseq[kMaxStages + 2];
  int nload;          // loader waves per workgroup (the last `nload` waves of the block)
  int nring;          // depth of the LDS weight ring of this launch (3..6)
  float* gmax[kMaxStages + 1];   // like ChainFwdArgs::amax for gstore[k] (the gradient entering stage k; [nstage]: the last one). Nullable.
  int ablate;         // experiment builds only (k_edge_bwd): every tile stores its layer gradients gstore[0..nstage-1] to tile 0 (no HBM stream)
};

// prepack table ---------------------------------------------------------------------------------
enum PackKind { PACK_FRAG = 0, PACK_FRAG_T = 1, PACK_TRANSPOSE = 2, PACK_ROWS_BF16 = 3 };
struct PackDesc {
  const float* W;  // row-major, leading dimension ld
  const float* bias;  // FRAG kinds: [N] copied into the header of the last chunk (bf16: of chunk 0); nullable -> zeros
  float* dst;      // FRAG kinds: pack_floats(N) floats
  int ld, row0, col0;
  int N, K;  // logical matrix M[n][k], n < N (outputs), k < K (reduction)
  int kind;  // FRAG: M[n][k] = W[row0+n][col0+k]; FRAG_T: M[n][k] = W[row0+k][col0+n];   (N == K, multiple of 32)
             // TRANSPOSE: dst[k*N+n] = W[row0+n][col0+k] (plain, for the small VALU layers)
             // ROWS_BF16 (N = K = 128): the LDS image of efuse.hip -- row n at byte 288 n, its 8-byte piece `q` (bf16 of
             //   M[n][4 q .. 4 q + 3], M as FRAG) at piece slot q ^ ((n >> 2) & 3); 36 KB, copied into LDS by LDS-DMA
  int bf16;  // FRAG kinds: ONE plane = the weight rounded to bf16 (bf16 precision)
  int mate;  // FRAG kinds, fp32 path: 1 + index of a pack of the same table that must share this pack's scale 2^k_w (the two
             // halves of a Linear over concatenated inputs accumulate into one set of registers); 0 = none
};
constexpr int kMaxPack = 40;
// Magnitude bounds of the tensors the weight-gradient kernel multiplies (ChainFwdArgs::amax, ChainBwdArgs::gmax,
// WgradJob::g_bound).  A bound "slot" is an array of kBoundWidth floats: every compute wave of a chain launch writes the
// largest |value| it saw into its own entry (blockIdx * 8 + wave; plain stores, no atomics: same-address atomics from
// ~10^5 waves cost milliseconds), the consumer takes the maximum over the whole array (unused entries are zero: the
// prepack of the forward call clears all slots of a block).
constexpr int kBoundWidth = 8192;  // 8 x the largest persistent chain grid (4 workgroups per CU at D <= 64, 256 CUs); beyond that: atomic max on the last entry
constexpr int kBoundSlots = 32;    // slots of one GMP block / MLP
struct PackTable {
  int n;
  PackDesc d[kMaxPack];
  float* zero;   // nullable: kBoundSlots x kBoundWidth floats the prepack sets to zero
};
int launch_prepack(const PackTable& t, hipStream_t s);

int launch_chain_fwd(int D, int in_mode, int out_mode, const ChainFwdArgs& a, hipStream_t s);
int launch_chain_bwd(int D, int gin_mode, int first_mode, const ChainBwdArgs& a, hipStream_t s);

// weight gradients ------------------------------------------------------------------------------
struct WgradJob {
  const float* G;  // [R, ldg] gradient w.r.t. the Linear's output
  const float* A;  // [R, lda] the Linear's input
  float* dW;       // dW[n*ldw + col0 + k] = sum_r G[r][n] * A[r][k]   (n,k < D)
  float* db;       // db[n] = sum_r G[r][n]   (nullable)
  int64_t R;
  int ldg, lda, ldw, col0;
  int bf16;        // G and A are bf16 tensors (ld in elements): one bf16 product instead of the split products
  // Magnitude bounds of the two operands (bound slots written by the chain kernels, ChainFwdArgs::amax / ChainBwdArgs::gmax;
  // bound = max(slot[0 .. kBoundWidth)) * mul >= max |value|).  Both given: the products run as fp16 x 2 pieces with per-TENSOR power-of-two
  // scales (chain.h; the reduction index is the row, so a scale cannot vary by row) -- three MFMAs per fragment pair.
  // Either missing: the exact three-way bf16 split of rounds 1-2 (six MFMAs), which needs no range information.
  const float *g_bound, *a_bound;
  float g_mul, a_mul;
};
constexpr int kMaxWgradJobs = 20;
size_t wgrad_work_bytes(int D, int njobs);
int launch_wgrad(int D, const WgradJob* jobs, int njobs, void* work, hipStream_t s);

// small-side weight gradients: out[s*os + f*of] = sum_r G[r][f] * S[r][s],  s < S (<= 8)
struct SmallWgradArgs {
  const float* G;  // [R,D]
  const float* S;  // [R,S_ld] narrow matrix, S_cols <= S_ld columns used (null when fiber mode)
  int S_cols;
  int S_ld;        // row pitch of S (0 = S_cols)
  // fiber mode: S[r] = [pos_i - pos_j, |pos_i - pos_j|] recomputed from the plan (edge layer 0)
  const int32_t *src, *dst;
  int32_t E, N;
  const float* pos;
  int64_t pos_bstride;
  int p;
  float* out;  // element (s,f) at out[s*os + f*of]
  int64_t os, of;
  float* colsum;  // optional: colsum[f] = sum_r G[r][f]
  float* colsum_S;  // optional: colsum_S[s] = sum_r S[r][s]  (decoder output bias)
  int64_t R;
  int D;
};
size_t small_wgrad_work_bytes(int D);
size_t small_wgrad_work_bytes_rows(int D, int64_t workers);   // sized for the fused pair kernel: one block per 256 / (D/4) row workers
int launch_small_wgrad(const SmallWgradArgs& a, void* work, hipStream_t s);

// efuse.hip: the fused backward of the edge MLP in the bf16 precisions (D = 128, hidden = 3): forward recompute + LayerNorm
// backward + dgrad chain + the weight / bias gradients of Linears 1..3 accumulated on chip; g_0 leaves as bf16
struct EdgeFusedBwdArgs {
  int64_t R;                  // B * E edge rows (plan order)
  int32_t E, N;
  const int32_t *src, *dst;   // plan-order endpoints
  const float *Ps, *Pd;       // the forward's node projections [B*N, D] (kept in the saved blob in this mode)
  const float* fiber;         // [R, 4]: the fiber rows the forward kept
  const float* wft;           // fiber weights^T [p+1][D]
  int p;
  const float4* wr[3];        // PACK_ROWS_BF16 images of edge Linears 1..3 (the rounding of the forward's packs)
  const float* b[3];          // their biases (fp32 parameters)
  const float* dy;            // [B*N, D] gradient of the aggregate (gathered by target)
  const void* y;              // [R, D] bf16 messages
  const float* rstd;          // [R]
  void* g0;                   // [pad_rows(R), D] bf16: gradient w.r.t. the first edge Linear's output
  float* gmax;                // bound slot of g0 (kBoundWidth floats), nullable
  float* part;                // per-workgroup partials, edge_fused_part_floats() floats
  int ntiles;                 // filled by the launcher
  unsigned long long* timing; // experiments only: phase stamps (null in production)
};
// csrc/experiments/efuse32.hip (EXPERIMENT builds only): fp32 (BSMS_F32), round 6: LayerNorm backward + dgrad chain + dW / db of Linears 1..3 on chip WITHOUT forward recompute --
// the forward saved a_0..a_2 as fp16 x 2 pieces (ChainFwdArgs::pieces), the backward brings their tiles HBM -> LDS by LDS-DMA; the
// transposed packs stream through the LDS ring; partials in the layout of efuse.hip (launch_edge_fused_reduce sums them)
struct EdgeFused32Args {
  int64_t R;                  // B * E edge rows (plan order)
  int32_t E, N;
  const int32_t* dst;         // plan-order targets
  const float* act[3];        // a_2, a_1, a_0 (execution order) as saved by k_edge_fwd SAVE == 2: pieces + sign bits (act_floats layout)
  const int* aexp[3];         // their rows' scale exponents
  const float4* wseq[3];      // FRAG_T packs of Linears 3, 2, 1
  const float* dy;            // [B*N, D] gradient of the aggregate (gathered by target)
  const float* y;             // [R, D] messages
  const float* rstd;          // [R]
  float* g0;                  // [pad_rows(R), D]: gradient w.r.t. the first edge Linear's output
  float* part;                // per-workgroup partials, edge_fused_part_floats() floats
  int ntiles;                 // filled by the launcher
  unsigned long long* timing; // experiments only: phase stamps (null in production)
};
bool edge_fused32_supported(int64_t D, int H, int64_t p, int precision);
int launch_edge_fused32_bwd(EdgeFused32Args a, int* nwg_out, hipStream_t s);
constexpr int kEdgeFusedMaxWg = 256;
bool edge_fused_supported(int64_t D, int H, int64_t p, int precision);
size_t edge_fused_part_floats(int64_t rows);   // for a launch over `rows` edge rows: one partial per workgroup, at most kEdgeFusedMaxWg
int launch_edge_fused_bwd(EdgeFusedBwdArgs a, int* nwg_out, hipStream_t s);
int launch_edge_fused_reduce(const float* part, int nwg, float* const dW[3], float* const db[3], hipStream_t s);

// efwd.hip: the edge MLP forward of the bf16 precisions with the three D x D weights resident in LDS (D = 128, hidden = 3):
// 16 independent waves per workgroup, no weight ring, no barriers; bit-identical to k_chain_fwd<8, IN_EDGE, OUT_LN, BF>
struct EdgeFwdResArgs {
  int64_t R;                  // B * E edge rows (plan order)
  int32_t E, N;
  const int32_t *src, *dst;   // plan-order endpoints
  const float *Ps, *Pd;       // node projections [B*N, D]
  const float* pos;           // [B, N, p] (batch stride pos_bstride; 0 = one point set for every sample)
  int64_t pos_bstride;
  int p;
  const float* wft;           // fiber weights^T [p+1][D]
  const float4* wp[3];        // the one-plane bf16 packs of edge Linears 1..3 (PackDesc::bf16; bias in the header of chunk 0)
  void* y;                    // [R, D] bf16 messages
  float* rstd;                // [R] (nullable: inference)
  float* fiber_out;           // [R, 4] (nullable: inference)
  int ntiles;                 // filled by the launcher
};
bool edge_fwd_res_supported(int64_t D, int H, int64_t p, int precision);
int launch_edge_fwd_res(EdgeFwdResArgs a, hipStream_t s);

// from rowsum.hip
int rowsum_plan_order(const bsms_plan* p, const float* x, int64_t B, int64_t D, float* out, hipStream_t s);
int rowsum_by_source(const bsms_plan* p, const float* x, int64_t B, int64_t D, float* out, hipStream_t s);
int rowsum_source_and_target(const bsms_plan* p, const float* x, int64_t B, int64_t D, float* outS, float* outD, hipStream_t s);
int rowsum_source_target_fiber(const bsms_plan* p, const float* x, int64_t B, int64_t D, float* outS, float* outD,
                               const float* fiber, int ld, int ncols, float* part, int64_t part_blocks, int* nwg, hipStream_t s,
                               bool x_bf16 = false);
int rowsum_plan_order_bf16(const bsms_plan* p, const float* x_bf16, int64_t B, int64_t D, float* out, hipStream_t s);
// wgrad.hip: partial blocks of the narrow weight gradients ([blocks][10][D] floats) and their fixed-order reduction
int64_t small_wgrad_part_blocks(size_t work_bytes, int D);
int launch_small_reduce(const SmallWgradArgs& a, const void* work, int nwg, hipStream_t s);

}  // namespace bsms
