/* grl_hip.h -- C ABI of libgrl_hip.so: hand-written gfx950 (MI355X) kernels for the policy-update hot path of
 * thobotics/geometry_rl (HEPi / EMPN actor, DeepSets critic, TRPL objective, GAE, Adam).
 *
 * Conventions (all entry points):
 *   - every pointer is a DEVICE pointer borrowed for the duration of the call unless marked HOST; the caller allocates
 *     every output and workspace; fp32 row-major tensors; index arrays are int32;
 *   - launches are asynchronous on `stream`; return value 0 = enqueued, -2 = unsupported shape, <= -1000 = -(1000+hipError_t);
 *   - thread safety: launches on different streams may be issued from different host threads; the library keeps three pieces of
 *     process-wide state, all initialised once under std::call_once or guarded explicitly: (1) the per-kernel
 *     "max dynamic LDS" function attributes (set once per process), (2) the optional event-timing records of grl_prof_*
 *     (measurement aid: NOT thread-safe, keep it off in multi-threaded hosts), (3) nothing else -- there is no cached device
 *     memory (but the 4 KB stamp buffer of grl_prof_enable(2)), no allocator and no stream owned by the library;
 *   - "partial" buffers are per-workgroup weight-gradient rows [n_rows][partial_size]; sum them with grl_reduce_partials.
 * The reference has no native ABI for this path (it is Python on PyG / torch_scatter / ITPAL); each group below cites the
 * reference code it replaces (paths relative to the reference checkout).
 */
#pragma once
#include <hip/hip_runtime_api.h>
#ifdef __cplusplus
extern "C" {
#endif

/* ABI version (major*10000 + minor*100 + patch); grl_version() returns the value the library was built with. */
#define GRL_HIP_VERSION 205   /* 205 (round 6): grl_source_hash added; 204 (round 5): grl_head_fused / grl_head_fused_rows removed, grl_calib_mfma / grl_calib_copy added; the exports are exactly this header */
int grl_version(void);
/* The hash of the sources this binary was built from (16 hex digits + NUL into buf; returns the length).  geometry_rl_amd/hip.py
   source_hash() recomputes it from csrc/, this header and the build's flag tables and refuses a library that disagrees. */
int grl_source_hash(char* buf, int cap);

/* ---- lift + node encoder: geometry_rl/modules/pyg_models/hepi.py:136-143, ponita/utils/to_from_sphere.py:4-9 ------------
 * x[n,o,:] = [scal[n,:] | vec[n,v,:] . grid[o,:]] W_enc^T ;  scal [N,S], vec [N,V,3], grid [16,3] (z = 0 for S1), W_enc [64,S+V] */
int grl_lift_encode_fwd(const float* scal, const float* vec, const float* grid, const float* Wenc, float* x, int n_nodes,
                        int n_scal, int n_vec, hipStream_t stream);
int grl_lift_bwd_blocks(int n_nodes);
int grl_lift_encode_bwd(const float* scal, const float* vec, const float* grid, const float* dx, float* partial, int n_nodes,
                        int n_scal, int n_vec, hipStream_t stream);
/* (ABI 203) every node type of a graph in ONE launch (the types share the encoder and the feature widths: hepi.py:136-143):
 * scal / vec / x (dx): HOST arrays of n_types <= 4 device pointers, n_nodes: HOST int array (<= 0: the type is skipped);
 * backward: partial [grl_lift_bwd_blocks_multi(n_types, n_nodes)][64 * (n_scal + n_vec)], the types' rows stacked -- sum all rows */
int grl_lift_encode_fwd_multi(int n_types, const float* const* scal, const float* const* vec, const float* grid, const float* Wenc,
                              float* const* x, const int* n_nodes, int n_scal, int n_vec, hipStream_t stream);
int grl_lift_bwd_blocks_multi(int n_types, const int* n_nodes);
int grl_lift_encode_bwd_multi(int n_types, const float* const* scal, const float* const* vec, const float* grid, const float* const* dx,
                              float* partial, const int* n_nodes, int n_scal, int n_vec, hipStream_t stream);

/* ---- fused edge pipeline: hepi.py:76-82,109-123,145-157 (invariants, PolynomialFeatures, basis MLP),
 *      ponita/conv.py:79-86,115-149 (kernel Linear, message, torch_scatter sum)  ==  ponita/ponita.py:153,161,327-345 ------
 * rowptr/e_src/e_dst = destination-sorted CSR (both directions); x1 [n_dst,16,64] (every row written);
 * backward: rowptr_s [n_src+1] / src_s [E] / dst_s [E] = the same edges in SOURCE-sorted CSR order (d x_src rows are summed
 *           in registers per source node, like the forward sums per destination node: no scratch, no atomics);
 *           dx_src [n_src,16,64] fully overwritten;
 *           partial [grl_edge_bwd_blocks(n_edges)][grl_edge_partial_size()] = [dW1 64x14 | db1 64 | dW2 64x64 | db2 64 | dWk 64x64]
 *           (one row per workgroup: its four waves are folded through LDS at the end of the launch).
 *           Since round 2 the backward is ONE launch (d x_src and the five weight gradients share one recompute of the basis MLP).
 * (libgrl_hip.so exports EXACTLY the functions this header declares: the link uses a version script generated from it; cross-file
 * helpers of the 16-row kernels -- grl_edge16_launch, grl_edge_bwd16_launch, grl_node_mlp_bwd16_launch -- are internal symbols.) */
int grl_edge_conv_fwd(const float* x_src, const float* pos_src, const float* pos_dst, const int* rowptr, const int* e_src,
                      const int* e_dst, int n_dst, const float* grid, int dim, const float* W1, const float* b1,
                      const float* W2, const float* b2, const float* Wk, float* x1, hipStream_t stream);
/* (ABI 202) the same with split_d [n_slots + 1], n_slots a multiple of 4 (or NULL / 0): node boundaries of an in-edge-balanced partition of
 * the destination-sorted CSR over n_slots wave slots (n_slots / 4 workgroups are launched); see grl_edge_conv_bwd_balanced.
 * (ABI 203) wimg16 / wimg32: optional pre-split weight images of this forward pass (grl_weight_images kinds 0 / 1, below); the launch
 * copies the one its kernel needs (grl_edge_fwd_image_kind(n_dst)) and stages W1 / W2 / Wk itself when that one is NULL.
 * grl_edge_fwd_slots(n_dst): the wave slots split_d has to cover for this n_dst (0: the launch ignores partitions);
 * grl_edge_fwd_chunk_nodes / grl_edge_bwd_chunk_nodes: destination / source nodes per round-robin chunk (what a partition replaces). */
int grl_edge_conv_fwd_balanced(const float* x_src, const float* pos_src, const float* pos_dst, const int* rowptr, const int* e_src,
                               const int* e_dst, int n_dst, const float* grid, int dim, const float* W1, const float* b1,
                               const float* W2, const float* b2, const float* Wk, float* x1, const int* split_d, int n_slots,
                               const void* wimg16, const void* wimg32, hipStream_t stream);
int grl_edge_fwd_image_kind(int n_dst);
int grl_edge_fwd_slots(int n_dst);
int grl_edge_fwd_chunk_nodes(int n_dst);
int grl_edge_bwd_chunk_nodes(int n_src);
int grl_edge_partial_size(void);
int grl_edge_bwd_blocks(int n_edges);
int grl_edge_conv_bwd(const float* x_src, const float* pos_src, const float* pos_dst, const int* rowptr, const int* e_src,
                      const int* e_dst, int n_dst, int n_edges, const int* rowptr_s, const int* src_s, const int* dst_s,
                      int n_src, const float* grid, int dim, const float* W1, const float* b1, const float* W2,
                      const float* b2, const float* Wk, const float* dx1, const float* dres /* optional [n_src,16,64] added to dx_src */,
                      float* dx_src, float* partial, hipStream_t stream);
/* (ABI 202) the same with split_s [4 * grl_edge_bwd_blocks(n_edges) + 1], or NULL: node boundaries of an edge-balanced partition of the
 * source-sorted CSR over the launch's wave slots (split_s[0] = 0, split_s[last] = n_src, non-decreasing; wave slot s walks the source
 * nodes split_s[s] .. split_s[s+1]).  The reference has no counterpart (PyG scatter kernels balance per element); results do not depend
 * on the partition beyond the summation order of the weight-gradient partial rows, and are reproducible for a given partition.
 * (ABI 203) n_slots_s = the length of split_s minus one: a partition built for another slot count is ignored (never read past its end);
 * wimg16: optional Edge16Image of this step's weights (grl_weight_images kind 0), NULL = the kernel stages its five images itself. */
int grl_edge_conv_bwd_balanced(const float* x_src, const float* pos_src, const float* pos_dst, const int* rowptr, const int* e_src,
                               const int* e_dst, int n_dst, int n_edges, const int* rowptr_s, const int* src_s, const int* dst_s,
                               int n_src, const float* grid, int dim, const float* W1, const float* b1, const float* W2,
                               const float* b2, const float* Wk, const float* dx1, const float* dres, float* dx_src, float* partial,
                               const int* split_s, int n_slots_s, const void* wimg16, hipStream_t stream);

/* ---- pre-split weight images (ABI 203; no reference counterpart: the reference's GEMMs read fp32 weights) -------------------------------
 * The MFMA kernels run their dense products as split-bf16 (three bf16 MFMAs per fp32 product) from LDS images of the weights.  Until
 * round 3 every launch rebuilt those images in each workgroup's prologue (15-40 us per launch at any batch size); grl_weight_images
 * builds them ONCE per forward pass, for all convolutions of the pass, in one launch -- the image bytes are the kernels' LDS structs, a
 * kernel's prologue is a linear copy.  kinds: 0 = edge chain, 16-row kernels, forward prefix + the backward's transposes (W1 [64,14], b1,
 * W2 [64,64], b2, Wk [64,64], grid [16,3]); 1 = edge chain of the few-tile 32-row forward (same six sources); 2 = ConvNeXt block forward
 * (W3 [256,64], b3, W4 [64,256], b4, gamma, beta); 3 = ConvNeXt block backward, per-lane operand fragments (W3, W4).
 * kinds [n] (HOST), srcs [n][6] (HOST array of device pointers, unused slots NULL), outs [n] (HOST array of device buffers of
 * grl_wimg_bytes(kind) bytes, 16-byte aligned).  The images are valid until the weights change (the optimizer step). */
int grl_wimg_bytes(int kind);
int grl_wimg_max_jobs(void);
int grl_weight_images(int n, const int* kinds, const float* const* srcs, void* const* outs, hipStream_t stream);
int grl_weight_images_bf16(int n, const int* kinds, const float* const* srcs, void* const* outs, hipStream_t stream);

/* ---- attention aggregation: FiberBundleConv(aggr="AttentionalAggregation"), ponita/conv.py:21-26,58-61,138-139;
 *      configs/algorithm/pyg_agent/model/hepi_attention.yaml; PyG 2.5.2 AttentionalAggregation / utils.softmax [upstream] ---------------
 * The messages m_e = K_e * x_src[src(e)] are materialised per edge (msg [E,16,64], rows in DESTINATION-sorted edge order), gated by
 * gate = ReLU(Linear(msg)) (a plain library GEMM on the caller's side), and summed per destination with softmax weights:
 * x1[d,o,c] = sum_e softmax_e(gate[e,o,c]) msg[e,o,c].  Backward: grl_softmax_aggregate_bwd hands back d gate and the direct part of
 * d msg; grl_edge_messages_bwd takes the complete per-edge d msg (s2d [E]: row of the i-th SOURCE-sorted edge in the destination-sorted
 * order) and produces d x_src and the five weight-gradient partial rows exactly like grl_edge_conv_bwd. */
int grl_edge_messages_fwd(const float* x_src, const float* pos_src, const float* pos_dst, const int* rowptr, const int* e_src,
                          const int* e_dst, int n_dst, int n_edges, const float* grid, int dim, const float* W1, const float* b1,
                          const float* W2, const float* b2, const float* Wk, float* msg, hipStream_t stream);
int grl_edge_messages_bwd(const float* x_src, const float* pos_src, const float* pos_dst, const int* rowptr, const int* e_src,
                          const int* e_dst, int n_dst, int n_edges, const int* rowptr_s, const int* src_s, const int* dst_s,
                          const int* s2d, int n_src, const float* grid, int dim, const float* W1, const float* b1, const float* W2,
                          const float* b2, const float* Wk, const float* dmsg, const float* dres, float* dx_src, float* partial,
                          hipStream_t stream);
int grl_softmax_aggregate_fwd(const float* gate, const float* msg, const int* rowptr, int n_dst, float* x1, hipStream_t stream);
int grl_softmax_aggregate_bwd(const float* gate, const float* msg, const float* x1, const float* dx1, const int* rowptr, int n_dst,
                              float* dgate, float* dmsg, hipStream_t stream);

/* ---- reduced-precision twins (BASELINE config 5: rope_shaping_hepi_trpl, "bf16 storage / MFMA, fp32 accumulate") --------------------
 * Same argument lists as the fp32 entry points above / below, with two differences: (1) every dense product is ONE bf16 MFMA (operands
 * rounded to nearest bf16, fp32 accumulation) instead of the split-bf16 triple, (2) the node latents -- x, x1, x2, out and their
 * gradients, i.e. every [N,16,64] tensor -- are stored as bf16 (`grl_bf16` = raw bits) and widened to fp32 in registers.  Weights,
 * positions, partial slabs and everything at the loss stay fp32.  Tolerance: tests/test_gpu_bf16_rope.py (<= 2e-2 relative on the
 * loss entries). */
typedef unsigned short grl_bf16;
int grl_lift_encode_fwd_bf16(const float* scal, const float* vec, const float* grid, const float* Wenc, grl_bf16* x, int n_nodes,
                             int n_scal, int n_vec, hipStream_t stream);
int grl_lift_encode_bwd_bf16(const float* scal, const float* vec, const float* grid, const grl_bf16* dx, float* partial, int n_nodes,
                             int n_scal, int n_vec, hipStream_t stream);
int grl_lift_encode_fwd_multi_bf16(int n_types, const float* const* scal, const float* const* vec, const float* grid, const float* Wenc,
                                   grl_bf16* const* x, const int* n_nodes, int n_scal, int n_vec, hipStream_t stream);
int grl_lift_encode_bwd_multi_bf16(int n_types, const float* const* scal, const float* const* vec, const float* grid,
                                   const grl_bf16* const* dx, float* partial, const int* n_nodes, int n_scal, int n_vec, hipStream_t stream);
int grl_edge_conv_fwd_bf16(const grl_bf16* x_src, const float* pos_src, const float* pos_dst, const int* rowptr, const int* e_src,
                           const int* e_dst, int n_dst, const float* grid, int dim, const float* W1, const float* b1,
                           const float* W2, const float* b2, const float* Wk, grl_bf16* x1, hipStream_t stream);
int grl_edge_conv_fwd_balanced_bf16(const grl_bf16* x_src, const float* pos_src, const float* pos_dst, const int* rowptr, const int* e_src,
                                    const int* e_dst, int n_dst, const float* grid, int dim, const float* W1, const float* b1,
                                    const float* W2, const float* b2, const float* Wk, grl_bf16* x1, const int* split_d, int n_slots,
                                    const void* wimg16, const void* wimg32, hipStream_t stream);
int grl_edge_conv_bwd_bf16(const grl_bf16* x_src, const float* pos_src, const float* pos_dst, const int* rowptr, const int* e_src,
                           const int* e_dst, int n_dst, int n_edges, const int* rowptr_s, const int* src_s, const int* dst_s,
                           int n_src, const float* grid, int dim, const float* W1, const float* b1, const float* W2,
                           const float* b2, const float* Wk, const grl_bf16* dx1, const grl_bf16* dres, grl_bf16* dx_src,
                           float* partial, hipStream_t stream);
int grl_edge_conv_bwd_balanced_bf16(const grl_bf16* x_src, const float* pos_src, const float* pos_dst, const int* rowptr, const int* e_src,
                                    const int* e_dst, int n_dst, int n_edges, const int* rowptr_s, const int* src_s, const int* dst_s,
                                    int n_src, const float* grid, int dim, const float* W1, const float* b1, const float* W2,
                                    const float* b2, const float* Wk, const grl_bf16* dx1, const grl_bf16* dres, grl_bf16* dx_src,
                                    float* partial, const int* split_s, int n_slots_s, const void* wimg16, hipStream_t stream);
int grl_edge_messages_fwd_bf16(const grl_bf16* x_src, const float* pos_src, const float* pos_dst, const int* rowptr, const int* e_src,
                               const int* e_dst, int n_dst, int n_edges, const float* grid, int dim, const float* W1,
                               const float* b1, const float* W2, const float* b2, const float* Wk, grl_bf16* msg, hipStream_t stream);
int grl_edge_messages_bwd_bf16(const grl_bf16* x_src, const float* pos_src, const float* pos_dst, const int* rowptr, const int* e_src,
                               const int* e_dst, int n_dst, int n_edges, const int* rowptr_s, const int* src_s, const int* dst_s,
                               const int* s2d, int n_src, const float* grid, int dim, const float* W1, const float* b1,
                               const float* W2, const float* b2, const float* Wk, const grl_bf16* dmsg, const grl_bf16* dres,
                               grl_bf16* dx_src, float* partial, hipStream_t stream);
int grl_softmax_aggregate_fwd_bf16(const float* gate, const grl_bf16* msg, const int* rowptr, int n_dst, grl_bf16* x1,
                                   hipStream_t stream);
int grl_softmax_aggregate_bwd_bf16(const float* gate, const grl_bf16* msg, const grl_bf16* x1, const grl_bf16* dx1, const int* rowptr,
                                   int n_dst, float* dgate, grl_bf16* dmsg, hipStream_t stream);
int grl_fiber_conv_fwd_bf16(const grl_bf16* x1, const float* fk, const float* bias, grl_bf16* x2, int n_nodes, hipStream_t stream);
int grl_fiber_conv_bwd_bf16(const grl_bf16* x1, const float* fk, const grl_bf16* dx2, grl_bf16* dx1, float* partial, int n_nodes,
                            hipStream_t stream);
int grl_node_mlp_fwd_bf16(const grl_bf16* x2, const grl_bf16* x_dst, const float* W3, const float* b3, const float* W4, const float* b4,
                          const float* gamma, const float* beta, grl_bf16* out, int n_rows, int accumulate, hipStream_t stream);
int grl_node_mlp_bwd_bf16(const grl_bf16* x2, const grl_bf16* dout, const float* W3, const float* b3, const float* W4, const float* b4,
                          const float* gamma, const float* beta, grl_bf16* dx2, float* partial, int n_rows, hipStream_t stream);
int grl_node_mlp_fwd_img_bf16(const grl_bf16* x2, const grl_bf16* x_dst, const float* W3, const float* b3, const float* W4, const float* b4,
                              const float* gamma, const float* beta, grl_bf16* out, int n_rows, int accumulate, const void* wimg,
                              hipStream_t stream);
int grl_node_mlp_bwd_img_bf16(const grl_bf16* x2, const grl_bf16* dout, const float* W3, const float* b3, const float* W4, const float* b4,
                              const float* gamma, const float* beta, grl_bf16* dx2, float* partial, int n_rows, const void* wimg,
                              hipStream_t stream);

/* ---- fiber kernel basis (parameter-only, 256 rows): hepi.py:109-123,157 / ponita.py:246-268 + conv.py:62 ------------------------
 * Phi = GELU(W2 GELU(W1 poly + b1) + b2), fk_i = Phi Wf_i^T for n_conv <= 4 convolutions, one launch each way.
 * wf / fk / dfk: HOST arrays of device pointers; saved: scratch [4,256,64] kept for the backward;
 * partial [grl_fiber_basis_blocks()][grl_fiber_basis_partial_size(n_conv)] = [dWf_0..dWf_{n-1} (4096 each) | dW2 4096 | db2 64 | dW1 192 | db1 64] */
int grl_fiber_basis_fwd(const float* poly, const float* W1, const float* b1, const float* W2, const float* b2, const float* const* wf,
                        int n_conv, float* saved, float* const* fk, hipStream_t stream);
int grl_fiber_basis_partial_size(int n_conv);
int grl_fiber_basis_blocks(void);
int grl_fiber_basis_bwd(const float* poly, const float* W2, const float* const* wf, int n_conv, const float* saved,
                        const float* const* dfk, float* partial, hipStream_t stream);

/* ---- depthwise fiber convolution + bias: ponita/conv.py:88-90,108-109 (ponita.py:164-166,183) ----------------------------
 * x2[n,p,c] = 1/16 sum_o x1[n,o,c] fk[o,p,c] + bias[c];  partial [grl_fiber_bwd_blocks][grl_fiber_partial_size] = [dfk | dbias] */
int grl_fiber_conv_fwd(const float* x1, const float* fk, const float* bias, float* x2, int n_nodes, hipStream_t stream);
int grl_fiber_partial_size(void);
int grl_fiber_bwd_blocks(int n_nodes);
int grl_fiber_conv_bwd(const float* x1, const float* fk, const float* dx2, float* dx1, float* partial, int n_nodes,
                       hipStream_t stream);

/* ---- ConvNeXt node block: ponita/conv.py:64-69,112 (ponita.py:219-230); hetero sum hetero_fiber_conv.py:63-64 -------------
 * out = (accumulate ? out : 0) + x_dst + W4 GELU(W3 LN(x2) + b3) + b4 ; n_rows = n_nodes*16
 * bwd: one fused launch (dx2 + all six parameter gradients) behind a one-workgroup launch that images W3 as MFMA fragments;
 * partial [grl_node_mlp_bwd_blocks(n_rows) + 1][grl_node_mlp_partial_size()]: rows 0 .. blocks-1 = [dW3 | db3 | dW4 | db4 | dgamma | dbeta]
 * (sum THOSE), the last row is scratch for that fragment image (since ABI 201; 200 wanted `blocks` rows) */
int grl_node_mlp_fwd(const float* x2, const float* x_dst, const float* W3, const float* b3, const float* W4, const float* b4,
                     const float* gamma, const float* beta, float* out, int n_rows, int accumulate, hipStream_t stream);
int grl_node_mlp_partial_size(void);
int grl_node_mlp_bwd_blocks(int n_rows);
int grl_node_mlp_bwd(const float* x2, const float* dout, const float* W3, const float* b3, const float* W4, const float* b4,
                     const float* gamma, const float* beta, float* dx2, float* partial, int n_rows, hipStream_t stream);
/* (ABI 203) the same with an optional pre-split weight image of this step (grl_weight_images kind 2 for the forward, kind 3 for the
 * backward; NULL = the kernel stages / fragments the weights itself) */
int grl_node_mlp_fwd_img(const float* x2, const float* x_dst, const float* W3, const float* b3, const float* W4, const float* b4,
                         const float* gamma, const float* beta, float* out, int n_rows, int accumulate, const void* wimg, hipStream_t stream);
int grl_node_mlp_bwd_img(const float* x2, const float* dout, const float* W3, const float* b3, const float* W4, const float* b4,
                         const float* gamma, const float* beta, float* dx2, float* partial, int n_rows, const void* wimg,
                         hipStream_t stream);

/* ---- read-out + contextual std head: hepi.py:173-190 (ponita_gcn.py:129-146),
 *      algorithms/trust_region_projections/models/policy/gnn_gaussian_policy_diag.py:65-87 ------------------------------------
 * mean [n,ov,3], sigma [n,3 ov], hidden [n,64]; partial [grl_readout_blocks][grl_readout_partial_size] */
int grl_readout_fwd(const float* lat, const float* grid, const float* Wd, const float* bd, const float* Ws, const float* bs,
                    float shift, float min_std, float* mean, float* sigma, float* hidden, int n_nodes, int output_dim,
                    int output_dim_vec, hipStream_t stream);
int grl_readout_partial_size(void);
int grl_readout_blocks(int n_nodes);
int grl_readout_bwd(const float* lat, const float* grid, const float* Wd, const float* bd, const float* Ws, const float* bs,
                    float shift, const float* dmean, const float* dsigma, const float* dhidden_ext, float* dlat, float* partial,
                    int n_nodes, int output_dim, int output_dim_vec, hipStream_t stream);

/* ---- TRPL objective: objectives/trpl.py:231-321, projections/base_projection_layer.py:71-100,292-384,
 *      projections/kl_projection_layer.py:15-111 (+ ITPAL BatchedDiagCovOnlyProjection), utils/projection_utils.py:34-67,
 *      objectives/utils.py:5-28 ---------------------------------------------------------------------------------------------
 *      projections/frob_projection_layer.py:9-88, projections/w2_projection_layer.py:14-76 (diagonal policy, closed forms)
 * cfg9 (HOST, TEN doubles since ABI 203): {mean_bound, cov_bound, trust_region_coeff, entropy_coef, critic_coef, clip_value, 1/B_global, B_global,
 *               projection type: 0 KL | 1 Frobenius | 2 Wasserstein (commutative, precision-scaled),
 *               adv_local: 1 = the advantage statistics are summed inside the kernel from this launch's batch (one rank), 0 = adv_stats}
 * sums fp64[12]: loss_objective, loss_trust_region, entropy(dist), loss_critic, sum w, sum w^2, mean_constraint,
 *               cov_constraint, entropy(p), entropy_diff, count, kl  (per-frame sums; divide by count);  maxes u32[2] (float bits) */
/* (ABI 203) the critic's share of the loss on its own: clipped l2 value loss (trpl.py:213-228, objectives/utils.py:5-28) and d loss / d V per
 * frame (already scaled by critic_coef / B_global) -- elementwise, so the critic's lane needs nothing from the fused actor kernel (which is
 * then called with value = NULL).  out2 fp64[2] = {sum over the frames of critic_coef * loss, that sum * inv_batch}; one workgroup */
int grl_value_loss(const float* value, const float* old_value, const float* value_target, double clip_value, double critic_coef,
                   double inv_batch, float* dvalue, double* out2, float* mean_out /* float[1] or NULL */, int batch, hipStream_t stream);
/* (ABI 203) data parallel: a rank's slots -> ONE record of 14 doubles (12 sums, 2 maxes); all-gather the records (one collective instead of
 * a SUM and a MAX all-reduce), then grl_trpl_report_records(records [n][14]) sums / maximises over the ranks and evaluates the values */
int grl_trpl_fold_record(const double* slots, int batch, double* rec14, hipStream_t stream);
int grl_trpl_report_records(const double* records, int n_records, double* sums, unsigned int* maxes, float entropy_coef, float* out14,
                            hipStream_t stream);
/* (ABI 203) the records as (hi, lo) float pairs inside a float buffer that is SUM-all-reduced -- the flat gradient's collective carries them:
 * region float[world][14][2]; a rank writes its own row and zeroes the others (x + 0 is exact in any summation order), so the reduced
 * region holds every rank's record; hi + lo restores the double to ~2^-48.  8-byte aligned. */
int grl_trpl_fold_record_pairs(const double* slots, int batch, int rank, int world, float* region, hipStream_t stream);
int grl_trpl_report_record_pairs(const float* region, int n_records, double* sums, unsigned int* maxes, float entropy_coef, float* out14,
                                 hipStream_t stream);
/* (ABI 203) grl_trpl_fold + grl_trpl_loss_values in one launch (one rank: nothing is all-reduced in between) */
int grl_trpl_report(const double* slots, int batch, double* sums, unsigned int* maxes, float entropy_coef, float* out14, hipStream_t stream);
/* stats fp64[2] = (sum, sum of squares) of the advantages, WRITTEN (ABI 203; <= 202 added to a zeroed slot): one workgroup, fixed order */
int grl_adv_stats(const float* advantage, double* stats, int batch, hipStream_t stream);
/* (ABI 202) sums == NULL in grl_trpl_fwd_bwd: the per-workgroup slots are left unfolded and the caller runs grl_trpl_fold later, e.g. on a
 * side stream -- the sums / maxes are reported values (trpl.py:280-321), nothing on the gradient path reads them. */
int grl_trpl_fold(const double* slots, int batch, double* sums /* [12] */, unsigned int* maxes /* [2] */, hipStream_t stream);
int grl_trpl_slot_doubles(int batch);   /* size (in doubles) of the per-workgroup slot workspace `slots` below */
int grl_trpl_fwd_bwd(const double* cfg9, int action_dim, const float* mean, const float* sigma, const float* action,
                     const float* old_mean, const float* old_var, const float* old_logp, const float* advantage,
                     const float* value, const float* old_value, const float* value_target, float* dmean, float* dsigma,
                     float* dvalue, float* proj_mean, float* proj_var, const double* adv_stats, double* sums,
                     unsigned int* maxes, double* slots /* scratch, grl_trpl_slot_doubles(batch) doubles: per-workgroup sums, folded
                     in a fixed order into sums / maxes (written, not accumulated; bitwise reproducible) */, int batch,
                     hipStream_t stream);
/* projection-layer boundary methods for an arbitrary DETACHED target (base_projection_layer.py:292-327 get_trust_region_loss,
 * :332-384 compute_metrics): the same kernel with its projection step skipped.  tgt_S = the target's "std" diagonal as the layer
 * sees it (= covariance diagonal of the policy).  sums[1] = trust_region_coeff * sum measure(p, target), sums[6..9,11] / maxes = the
 * metrics; dmean / dsigma = gradient of sums[1] / B_global.  zeros_b: device float[batch] of zeros. */
int grl_trpl_target_terms(const double* cfg9, int action_dim, const float* mean, const float* sigma, const float* tgt_mean,
                          const float* tgt_S, float* dmean, float* dsigma, double* sums, unsigned int* maxes, double* slots,
                          const float* zeros_b, int batch, hipStream_t stream);
/* reported loss-dict values (trpl.py:280-321) from the globally reduced sums / maxes:
 * out14 = [actor loss, critic loss, loss_trust_region, loss_entropy, ESS, kl, mean_constraint, mean_constraint_max,
 *          cov_constraint, cov_constraint_max, entropy, entropy_diff, loss_objective, constraint] */
int grl_trpl_loss_values(const double* sums, const unsigned int* maxes, float entropy_coef, float* out14, hipStream_t stream);

/* ---- collector-side sampling: ProbabilisticActor(..., torch.distributions.MultivariateNormal, return_log_prob=True)
 * (examples/torchrl/builders/utils_algo_graph.py:146-158; configs/algorithm/policy/default.yaml:6) */
int grl_gaussian_sample(const float* loc, const float* sigma, const float* eps, float* action, float* logp, float* var, int batch,
                        int action_dim, hipStream_t stream);

/* ---- DeepSets critic: geometry_rl/modules/pyg_models/deepsets.py:34-53, models/value/gnn_vf_net.py:50-86 ---------------------
 * three forward and three backward stages around the whole-tensor LayerNorm statistics (PyG LayerNorm mode="graph").
 * stats1/stats2/bstats1/bstats2 are slot arrays fp64[grl_deepsets_stat_slots()][2] (per-workgroup sums, fully written by the
 * producing stage, added up in a fixed order by the consuming one; a data-parallel caller all-reduces the whole array) */
int grl_deepsets_blocks(int batch);
int grl_deepsets_stat_slots(void);
int grl_deepsets_partial3(void);
int grl_deepsets_partial2(void);
int grl_deepsets_fwd1(const float* x, const float* W1, const float* b1, float* h1, double* stats1, int batch, int n_nodes, int d,
                      hipStream_t stream);
int grl_deepsets_fwd2(const float* h1, const double* stats1, double count1, const float* g1, const float* be1, const float* W2,
                      const float* b2, const float* W3, const float* b3, float* z, float* u1, double* stats2, int batch,
                      int n_nodes, hipStream_t stream);
int grl_deepsets_fwd3(const float* u1, const double* stats2, double count2, const float* g2, const float* be2, const float* W4,
                      const float* b4, const float* wv, const float* bv, float* value, int batch, hipStream_t stream);
/* grouped forward (ABI 202): `groups` independent batches per launch, x [groups][batch][n_nodes][d], slot arrays [groups][slots][2],
 * value [groups][batch] -- the critic pass over the T+1 frames of a rollout (gnn_vf_net.py:72-80 loops over T: statistics per time step;
 * examples/torchrl/train.py:249-251).  Each group's result is bitwise what the ungrouped call gives. */
int grl_deepsets_fwd1_groups(const float* x, const float* W1, const float* b1, float* h1, double* stats1, int batch, int n_nodes, int d,
                             int groups, hipStream_t stream);
int grl_deepsets_fwd2_groups(const float* h1, const double* stats1, double count1, const float* g1, const float* be1, const float* W2,
                             const float* b2, const float* W3, const float* b3, float* z, float* u1, double* stats2, int batch,
                             int n_nodes, int groups, hipStream_t stream);
int grl_deepsets_fwd3_groups(const float* u1, const double* stats2, double count2, const float* g2, const float* be2, const float* W4,
                             const float* b4, const float* wv, const float* bv, float* value, int batch, int groups, hipStream_t stream);
int grl_deepsets_bwd3(const float* u1, const double* stats2, double count2, const float* g2, const float* be2, const float* W4,
                      const float* b4, const float* wv, const float* dvalue, float* q2, double* bstats2, float* partial,
                      int batch, hipStream_t stream);
int grl_deepsets_bwd2(const float* h1, const double* stats1, double count1, const float* g1, const float* be1, const float* W2,
                      const float* W3, const float* z, const float* u1, const double* stats2, double count2, const float* q2,
                      const double* bstats2, float* q1, double* bstats1, float* partial, int batch, int n_nodes,
                      hipStream_t stream);
int grl_deepsets_bwd1(const float* x, const float* h1, const double* stats1, double count1, const float* q1,
                      const double* bstats1, float* partial, int batch, int n_nodes, int d, hipStream_t stream);

/* ---- training loop: examples/torchrl/train.py:134-146,249-251,308-316; pyg_data/rigid_tasks_data.py:285-287 ------------- */
int grl_reduce_partials(const float* partial, float* out, int n_rows, int n, hipStream_t stream);
/* dst[i][0..len[i]) += sum_rows partial[row*ld + start[i] + j], i < n_seg <= 8; dst/start/len are HOST arrays */
int grl_reduce_partials_seg(const float* partial, int n_rows, int ld, int n_seg, float* const* dst, const int* start, const int* len,
                            int overwrite_mask /* bit i: dst[i] = sum instead of += */, hipStream_t stream);
/* n_seg <= 64 independent folds in ONE launch (all arrays HOST arrays of length n_seg): every leaf gradient of a backward pass */
int grl_reduce_partials_multi(int n_seg, const float* const* partial, const int* n_rows, const int* ld, const int* start,
                              const int* len, float* const* dst, hipStream_t stream);
/* (ABI 203) overwrite != 0: every destination is WRITTEN with the sum of its slabs instead of accumulated into (no zeroed gradient buffer
 * needed), provided all slabs of a destination are in this call; the launch is a flat grid of one workgroup per 64 columns */
int grl_reduce_partials_multi_ow(int n_seg, const float* const* partial, const int* n_rows, const int* ld, const int* start,
                                 const int* len, float* const* dst, int overwrite, hipStream_t stream);
/* (ABI 203) the step's tail in ONE launch (one rank, no gradient clipping; train.py:308-316): the fold above, plus -- adam != 0 -- the Adam
 * update (grl_adam_step_dev's arithmetic, bit for bit) of every parameter entry whose gradient this launch produces (grads / params /
 * exp_avg / exp_avg_sq: parallel flat buffers, every dst inside grads; the gradient is stored too), plus -- slots != NULL -- one extra
 * workgroup that does grl_trpl_report's work.  step_dev: the count of THIS step (advanced earlier, e.g. by grl_build_features_bump). */
int grl_fold_adam_report(int n_seg, const float* const* partial, const int* n_rows, const int* ld, const int* start, const int* len,
                         float* const* dst, int overwrite, int adam, const float* grads, float* params, float* exp_avg, float* exp_avg_sq,
                         const float* lr_dev, float beta1, float beta2, float eps, const int* step_dev, const double* slots, int batch,
                         double* sums, unsigned int* maxes, float entropy_coef, float* out14, hipStream_t stream);
int grl_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int n, float lr, float beta1, float beta2,
                  float eps, int step, const float* scale_dev, float scale_host, hipStream_t stream);
/* the same update with the step count (int[1]) AND the learning rate (float[1]) in device memory: recordable into a hipGraph, and
 * a learning-rate schedule (anneal_lr, train.py:264-271; configs/algorithm/optim/default.yaml:5) reaches the replayed launch */
int grl_adam_step_dev(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int n, const float* lr_dev, float beta1,
                      float beta2, float eps, const int* step_dev, const float* scale_dev, float scale_host, hipStream_t stream);
/* torch.nn.utils.clip_grad_norm_ (train.py:308-310): sqnorm fp64[1] = |g|^2 (WRITTEN since ABI 203), coef float[1] = min(1, max_norm / (|g| + 1e-6)); one launch */
int grl_clip_coef(const float* grads, int n, float max_norm, double* sqnorm, float* coef, hipStream_t stream);
int grl_gae_scan(const float* reward, const unsigned char* done, const unsigned char* terminated, const float* values,
                 float* advantage, float* value_target, int n_env, int n_steps, float gamma, float lmbda, hipStream_t stream);
/* node features of the batched graph in ONE launch (pyg_data/rigid_tasks_data.py:150-250, cloth_tasks_data.py, rope_tasks_data.py):
 * every 3-vector feature = slice A of an observation group [- slice B], gathered per node.  descs: HOST array of n_desc <= 24
 * records of 18 8-byte words: [out, a, b, gather (device pointers, 0 = absent), out_row_stride, out_col, rows_per_sample,
 * row_off, n_nodes, n_per, a_stride, a_off, a_bcast, b_stride, b_off, b_bcast, onehot_col, n_types] */
int grl_build_features(const long long* descs, int n_desc, hipStream_t stream);
/* (ABI 203) the same; bump (device int[1] or NULL) is advanced by one by the launch: the optimizer's step count of a recorded step */
int grl_build_features_bump(const long long* descs, int n_desc, int* bump, hipStream_t stream);

/* ---- (ABI 205, round 6) merged launches of the recorded policy-update step and lane signals -----------------------------------------
 * Replaces nothing of the reference (it has no launches to merge: examples/torchrl/train.py:279-316 runs eager PyTorch); these shorten the
 * serial chain of a shard-sized step, where ~35 dependent launches of 5-15 us each are the cost.
 *
 * grl_step_head: the three mutually independent launches at the head of the actor's lane in ONE -- grl_build_features_bump (descs, n_desc,
 * bump), grl_fiber_basis_fwd (poly .. fk; n_conv = 0: role absent) and grl_weight_images (n_img, kinds, srcs, outs; n_img = 0: absent).
 * Arguments exactly as in those entry points; the results are bitwise theirs (the same device functions). */
int grl_step_head(const long long* descs, int n_desc, int* bump, const float* poly, const float* W1, const float* b1, const float* W2,
                  const float* b2, const float* const* wf, int n_conv, float* saved, float* const* fk, int n_img, const int* kinds,
                  const float* const* srcs, void* const* outs, hipStream_t stream);
int grl_step_head_bf16(const long long* descs, int n_desc, int* bump, const float* poly, const float* W1, const float* b1, const float* W2,
                       const float* b2, const float* const* wf, int n_conv, float* saved, float* const* fk, int n_img, const int* kinds,
                       const float* const* srcs, void* const* outs, hipStream_t stream);
/* grl_lift_fiber_basis_bwd: grl_lift_encode_bwd_multi (n_types .. n_vec; n_types = 0: absent) + grl_fiber_basis_bwd (poly .. fb_partial;
 * n_conv = 0: absent) in ONE launch at the end of the actor's backward; arguments and results as in those two entry points. */
int grl_lift_fiber_basis_bwd(int n_types, const float* const* scal, const float* const* vec, const float* grid, const float* const* dx,
                             float* lift_partial, const int* n_nodes, int n_scal, int n_vec, const float* poly, const float* W2,
                             const float* const* wf, int n_conv, const float* saved, const float* const* dfk, float* fb_partial,
                             hipStream_t stream);
int grl_lift_fiber_basis_bwd_bf16(int n_types, const float* const* scal, const float* const* vec, const float* grid,
                                  const grl_bf16* const* dx, float* lift_partial, const int* n_nodes, int n_scal, int n_vec,
                                  const float* poly, const float* W2, const float* const* wf, int n_conv, const float* saved,
                                  const float* const* dfk, float* fb_partial, hipStream_t stream);
/* Lane signals riding on a launch: flag_dst[0] := flag_src[0] (device int32, system-scope store) by one thread when the launch STARTS,
 * i.e. when everything in front of it on the stream has finished -- what another stream's hipStreamWaitValue32 waits for, without a copy
 * launch of its own on the chain.  grl_fiber_conv_fwd_sig = grl_fiber_conv_fwd (+ signal: "the edge convolution in front is done");
 * grl_fold_adam_report_sig = grl_fold_adam_report (+ signal at the lane's end). */
int grl_fiber_conv_fwd_sig(const float* x1, const float* fk, const float* bias, float* x2, int n_nodes, int* flag_dst, const int* flag_src,
                           hipStream_t stream);
int grl_fiber_conv_fwd_sig_bf16(const grl_bf16* x1, const float* fk, const float* bias, grl_bf16* x2, int n_nodes, int* flag_dst,
                                const int* flag_src, hipStream_t stream);
int grl_fold_adam_report_sig(int n_seg, const float* const* partial, const int* n_rows, const int* ld, const int* start, const int* len,
                             float* const* dst, int overwrite, int adam, const float* grads, float* params, float* exp_avg,
                             float* exp_avg_sq, const float* lr_dev, float beta1, float beta2, float eps, const int* step_dev,
                             const double* slots, int batch, double* sums, unsigned int* maxes, float entropy_coef, float* out14,
                             int* flag_dst, const int* flag_src, hipStream_t stream);
/* ---- (ABI 205, round 6) one-shot all-reduce over directly mapped peer buffers (csrc/oneshot.hip) -----------------------------------
 * For the data-parallel step's one bandwidth-relevant collective, the actor's gradient slice + loss records (the
 * reference has no distributed code -- the semantics kept are examples/torchrl/train.py:304-316: every replica applies Adam to the same
 * summed gradient).  Each rank writes its contribution to chunk p into rank p's staging row, rank p sums its chunk in RANK ORDER and writes
 * the result into every rank's buffer: bitwise identical on all ranks.  bufs / stages / flags: HOST arrays of `world` (<= 8) device
 * pointers, entry p = rank p's areas as mapped into this process (hipIpcOpenMemHandle for peers; the same pointers in a single-process
 * test) -- bufs[p]: the n payload floats (n a multiple of 4), stages[p]: grl_oneshot_stage_floats floats, flags[p]:
 * grl_oneshot_flag_words 32-bit words zeroed ONCE.  seq: the call's number, the same on all ranks, strictly increasing from 1.  status:
 * device int32 written only on a timeout (1 / 2).  Every rank must enqueue the call; the kernels wait for each other on the device, each
 * wait bounded by timeout_ms. */
int grl_oneshot_chunk_floats(int n, int world);
int grl_oneshot_stage_floats(int n, int world);
int grl_oneshot_flag_words(int world);
int grl_oneshot_blocks(void);
int grl_oneshot_allreduce(float* const* bufs, float* const* stages, unsigned* const* flags, int rank, int world, int n, unsigned seq,
                          int timeout_ms, int* status, hipStream_t stream);
/* the same protocol with all `world` stand-in ranks in ONE launch (rank = blockIdx.y; status: world ints): the single-process test */
int grl_oneshot_allreduce_local(float* const* bufs, float* const* stages, unsigned* const* flags, int world, int n, unsigned seq,
                                int timeout_ms, int* status, hipStream_t stream);
/* A lane gate as a launch: the stream waits (one-thread kernel, capturable into a hipGraph) until flag[0] >= count[0] + add -- both read from
 * DEVICE memory when the kernel runs -- or timeout_us microseconds have passed (a gate is a scheduling hint: the lane then simply goes on). */
int grl_wait_flag_ge(const int* flag, const int* count, int add, int timeout_us, hipStream_t stream);
/* Data parallel, two more merged launches (round 6): grl_fold_record_pairs = grl_reduce_partials_multi_ow (n_seg .. overwrite) + grl_trpl_fold_record_pairs
 * (slots .. region) -- both only feed the lane's all-reduce; grl_adam_report_record_pairs = grl_adam_step_dev (params .. step_dev; scale 1) +
 * grl_trpl_report_record_pairs (region .. out14) -- both only read what the all-reduce delivered.  Arguments and results as in those entry points. */
int grl_fold_record_pairs(int n_seg, const float* const* partial, const int* n_rows, const int* ld, const int* start, const int* len,
                          float* const* dst, int overwrite, const double* slots, int batch, int rank, int world, float* region,
                          hipStream_t stream);
int grl_adam_report_record_pairs(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int n, const float* lr_dev, float beta1,
                                 float beta2, float eps, const int* step_dev, const float* region, int n_records, double* sums,
                                 unsigned int* maxes, float entropy_coef, float* out14, hipStream_t stream);
/* 1 if the current device supports hipStreamWaitValue32 (hipDeviceAttributeCanUseStreamWaitValue), else 0 (host query, no stream). */
int grl_can_stream_wait_value(void);
/* n <= 24 small device-to-device copies in one launch (host arrays of device pointers / byte counts) */
int grl_copy_many(void* const* dst, const void* const* src, const long long* bytes, int n, hipStream_t stream);
/* minibatch assembly from a device-resident rollout (train.py:120,128,258-261): dst[k][i,:] = src[k][idx[i],:] for k < n <= 24
 * tensors in one launch; dst / src / row_bytes: HOST arrays; idx: DEVICE int64[n_rows]; row_bytes multiples of 4 */
int grl_gather_rows_many(void* const* dst, const void* const* src, const long long* row_bytes, int n, const long long* idx, int n_rows,
                         hipStream_t stream);
/* (ABI 205) the same with the index row chosen ON THE DEVICE: idx = a matrix [n_idx_rows, n_rows] (device int64), the launch gathers line
 * (count[0] - base[0]) mod n_idx_rows (device int32[1] each) -- a recorded step takes "the next minibatch of the epoch"
 * (examples/torchrl/train.py:258-261) without the host touching its arguments.  count = NULL: idx is the row itself. */
int grl_gather_rows_many_cur(void* const* dst, const void* const* src, const long long* row_bytes, int n, const long long* idx, int n_rows,
                             const int* count, const int* base, int n_idx_rows, hipStream_t stream);
/* ---- collector-side observation transform: NDVecNorm / VecNorm running normalisation + ClipTransform,
 * geometry_rl/torchrl/envs/transforms.py:141-163 (on torchrl's VecNorm), configs/rigid_insertion_multi_hepi_trpl_cfg.yaml:47-72.
 * x [rows, K<=64]; state: device float[2K+1] = [sum | ssq | count], updated in place when update != 0;
 * y_norm = clip((x - mean)/max(std, eps)), y_clip = clip(x) (either may be NULL); scratch: grl_vecnorm_scratch_bytes(K) device bytes */
int grl_vecnorm(const float* x, long long rows, int K, float decay, float eps, int update, float lo, float hi, float* state,
                void* scratch, float* y_norm, float* y_clip, hipStream_t stream);
int grl_vecnorm_scratch_bytes(int K);
int grl_knn_topology(const float* pos, const int* n_valid, int* out_nbr, int batch, int n_points, int k, hipStream_t stream);


/* ---- measurement aid (no reference counterpart): timing of the individual kernels inside the entry points, recorded on the launch
 * stream.  Off by default.  grl_prof_enable(1): HIP events around the kernels of the multi-kernel entry points (grl_edge_conv_bwd,
 * grl_node_mlp_bwd).  grl_prof_enable(2): one-thread kernels that store the device wall clock in front of and behind the MFMA launches
 * (edge forward / backward, node-MLP forward / backward) -- ordinary kernel nodes when the caller records the launches into a hipGraph, so
 * grl_prof_get returns the duration of the LATEST execution, replays included (the only state the library allocates: one 4 KB device
 * buffer, kept for the life of the process).  grl_prof_enable(0) clears the records. */
int grl_prof_enable(int on);
int grl_prof_count(void);
int grl_prof_get(int i, char* name, int cap, float* ms);

/* ---- measurement support (bench.py `box_calibration`; not part of the policy-update path, nothing in the package calls them) -------------
 * Two FIXED kernels that identify the speed of the box a bench line was taken on (the pool's boxes hold different clocks under load):
 * grl_calib_mfma: 1024 waves (one per SIMD) issue v_mfma_f32_32x32x16_bf16 back to back on random register operands; FLOPs per call =
 *                1024 * iters * 16 * 32768; out: 65536 floats (written).
 * grl_calib_copy: float4 grid-stride copy, bytes a multiple of 16; bytes moved per call = 2 * bytes. */
int grl_calib_mfma(int iters, float* out, hipStream_t stream);
int grl_calib_copy(const void* src, void* dst, long long bytes, hipStream_t stream);
/* one wave that idles for `us` microseconds (s_memrealtime): a delay node for lane-placement experiments (tools/critic_delay_ab.sh) */
int grl_calib_spin(int us, hipStream_t stream);

#ifdef __cplusplus
}
#endif
