/*
 * mpnhip.h -- C ABI of the MI355X-native MPNTrackSeg message-passing hot path.
 *
 * The reference (ocetintas/MPNTrackSeg) has no FFI: the path sits behind a Python nn.Module API.
 * Each entry point below names the reference interface it replaces (paths relative to
 * /root/reference/src/mot_neural_solver/); INTEGRATION.md shows the ctypes binding a maintainer
 * would add.  All pointers are DEVICE pointers (HIP) unless stated otherwise; buffers are caller
 * owned; `stream` is a hipStream_t passed as void*; nothing here allocates, frees or synchronises,
 * so every call can be captured in a HIP graph.  Return value: 0 on success, negative MPNHIP_ERR_*
 * otherwise (mpnhip_last_error() gives a host string).  Feature matrices are dense row-major fp32.
 */
#ifndef MPNHIP_H
#define MPNHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MPNHIP_OK 0
#define MPNHIP_ERR_ARG (-1)         /* bad argument (null pointer, inconsistent dims, misalignment) */
#define MPNHIP_ERR_HIP (-2)         /* a HIP runtime call / kernel launch failed */
#define MPNHIP_ERR_WORKSPACE (-3)   /* caller buffer too small */
#define MPNHIP_ERR_UNSUPPORTED (-4) /* valid reference configuration this build does not cover */

#define MPNHIP_MAX_LAYERS 8

/* Operand precision of every Linear layer's product.  FP32 (default): fp32 operands and accumulation, the reference's
 * arithmetic.  BF16: activations and weights are rounded to bf16 (round to nearest even) as they enter the product,
 * accumulation, biases, gather-adds and aggregation stay fp32 -- BASELINE.json's "bf16 MLP GEMMs on MFMA"
 * configuration (SURVEY.md section 8d cfg-E).  mpnhip_backward rounds the operands of its products the same way (round 3).
 * Round 4: at the fused chain's widths (edge dim 16 ... 128) mpnhip_forward_saved keeps the hidden activations of the per-edge
 * modules as bf16 rows -- the values the backward's products would round them to anyway -- and the ReLU decisions as bits, and
 * mpnhip_backward runs one fused chain kernel per step over them; the workspaces of the two calls belong together as before. */
#define MPNHIP_PREC_FP32 0
#define MPNHIP_PREC_BF16 1
/* FP32_SPLIT: fp32 results from bf16 matrix instructions.  In the fused per-edge chain kernels (forward and backward) every
 * fp32 operand is taken as the exact sum of three bfloat16 pieces (x = h + m + l, each the RNE rounding of the remainder)
 * and a product is accumulated in fp32 from the six piece products whose weight is at least 2^-16 of the leading one; the
 * three dropped products are below 2^-26 |a b|.  Nothing is rounded to bf16: measured against float64 the logits' error is the
 * FP32 mode's (3 ... 7e-7 relative at cfg-B, 12 steps) and every gradient's is too (decision-pinned comparison, tests/
 * test_gpu_pinned.py: <= 5e-6, the FP32 mode's bound), at 3/8 of the fp32 MFMA's cycles.  One hardware property is handled in the
 * backward kernel: v_mfma_f32_32x32x16_bf16 adds its products to the accumulator with a small bias toward -infinity (mean error
 * -0.06 ... -0.11 of the rms error of a six-product result; tools/micro/mfma_bias.hip), which the backward's sums over edges
 * and steps would add up coherently -- the kernel keeps the gradients of every other edge negated in its registers, so the
 * bias cancels in every sum.  (The forward keeps it: a common offset of ~4e-7 of the logits' scale.)  The larger K-contiguous
 * GEMMs (node projections, encoders) use the same six products; the remaining products (small GEMMs, weight gradients) stay
 * fp32 MFMAs.  Forward and backward.  Operands must be finite and below 3.3e38 in magnitude: an infinite operand gives NaN
 * (inf - inf in the split) where the FP32 mode gives +-inf; pieces below the bf16 normal range (|x| < 1e-33) may be flushed. */
#define MPNHIP_PREC_FP32_SPLIT 2
/* FP32_WGSPLIT: MPNHIP_PREC_FP32 in every product of the forward and of the backward's activation-gradient chain (fp32 MFMAs), and
 * the FP32_SPLIT form for the WEIGHT-GRADIENT products only: there it is the batched row-panel kernel (csrc/wgrad_panel.hip: all
 * products of a group of steps in one launch) that pays -- at the reference's graph sizes (configs[2] / configs[3]: a few hundred
 * nodes) a training step is bound by its ~50 small weight-gradient launches, while the chain kernels, one wave per SIMD, are
 * faster on fp32 MFMAs.  Same accuracy class as the other two fp32 modes; what 'auto' selects for small graphs. */
#define MPNHIP_PREC_FP32_WGSPLIT 3

#define MPNHIP_AGG_SUM 0  /* torch_scatter.scatter_add  (models/mpn.py:273) */
#define MPNHIP_AGG_MEAN 1 /* torch_scatter.scatter_mean (models/mpn.py:267) */
#define MPNHIP_AGG_MAX 2  /* torch_scatter.scatter_max  (models/mpn.py:270), empty segment -> 0 */

/* One reference `MLP` (models/mlp.py:4-28) with dropout_p = 0 and use_batchnorm = False:
 * n_layers Linear layers, weight[i] is nn.Linear layout [out_dims[i], in] row-major, ReLU after every
 * layer whose out dim != 1 (mlp.py:17).  grad pointers (same shapes) are used by mpnhip_backward only
 * and may be NULL otherwise; gradients are ACCUMULATED into them (+=), like autograd's .grad. */
typedef struct {
    int n_layers;
    int in_dim;
    int out_dims[MPNHIP_MAX_LAYERS];
    const float* weight[MPNHIP_MAX_LAYERS];
    const float* bias[MPNHIP_MAX_LAYERS];
    float* grad_weight[MPNHIP_MAX_LAYERS];
    float* grad_bias[MPNHIP_MAX_LAYERS];
} mpnhip_mlp;

/* The hot-path sub-modules of MOTMPNet (models/mpn.py:220-317) -- what
 * `MOTMPNet.__init__(model_params)` builds from `graph_model_params` (configs/tracking_cfg.yaml:134-168). */
typedef struct {
    int dn;               /* encoder_feats_dict.node_out_dim */
    int de;               /* encoder_feats_dict.edge_out_dim */
    int reattach_nodes;   /* reattach_initial_nodes (mpn.py:276) */
    int reattach_edges;   /* reattach_initial_edges (mpn.py:277) */
    int agg;              /* node_agg_fn: MPNHIP_AGG_*  (mpn.py:263-273) */
    int num_enc_steps;    /* mpn.py:249 */
    mpnhip_mlp enc_node;  /* encoder.node_model       (mpn.py:153,237) */
    mpnhip_mlp enc_edge;  /* encoder.edge_model       (mpn.py:159,237) */
    mpnhip_mlp edge;      /* MPNet.edge_model.edge_model, in = nf*2*dn + ef*de (mpn.py:282-283,294) */
    mpnhip_mlp flow_in;   /* MPNet.node_model.flow_in_model,  in = nf*dn + de (mpn.py:285,299) */
    mpnhip_mlp flow_out;  /* MPNet.node_model.flow_out_model (mpn.py:304) */
    mpnhip_mlp node;      /* MPNet.node_model.node_model: ONE Linear(2dn -> dn) + ReLU (mpn.py:309-310) */
    mpnhip_mlp classifier;/* classifier.edge_model    (mpn.py:238) */
    int precision;        /* MPNHIP_PREC_*: operand precision of the Linear layers' products (forward and backward) */
    int weights_prepacked;/* != 0: the head of `workspace` still holds the weight images a previous mpnhip_forward
                           * (save_for_backward = 0) of THIS model wrote there and no weight has changed since: skip
                           * re-packing them (about 20 small launches).  The caller vouches for it; 0 is always safe. */
} mpnhip_model;

const char* mpnhip_version(void);
/* Host string describing the last error raised on the calling thread ("" if none). */
const char* mpnhip_last_error(void);

/* Test instrumentation: host-side counters of the kernel variants launched by this process (which code path a call took:
 * fused chain or GEMMs, block-per-segment or short-segment reductions, ...).  Copies min(capacity, count) values into
 * counts (may be NULL), zeroes them when reset != 0, returns the number of counters; mpnhip_debug_counter_name(i) names
 * counter i ("" past the end).  No reference counterpart: the parity tests use it to prove that the kernel they mean to
 * check is the one that ran. */
int mpnhip_debug_counters(int64_t* counts, int capacity, int reset);
const char* mpnhip_debug_counter_name(int index);

/* Test instrumentation: one activation that mpnhip_forward(save_for_backward = 1) left in its workspace, copied to `out`
 * [rows, width] fp32 in ORIGINAL node / edge order (the workspace keeps per-edge tensors in sorted order).  The parity tests
 * read the forward's ReLU / arg-max DECISIONS from these (value > 0) and differentiate the oracle on the same branch of the
 * piecewise-linear function, so that a gradient comparison is not at the mercy of a pre-activation that sits within fp32
 * noise of zero (tests/test_gpu_pinned.py).  out == NULL: only rows / width are reported.
 *   what                         step        layer   contents (reference line)
 *   MPNHIP_SAVED_ENC_NODE / EDGE -           i       encoder hidden layer i, post-ReLU (mpn.py:355)
 *   MPNHIP_SAVED_X / _E          0..L        -       latent node / edge features after `step` steps (0 = encoder output)
 *   MPNHIP_SAVED_EDGE_HIDDEN     1..L        i       EdgeModel MLP hidden layer i (mpn.py:69)
 *   MPNHIP_SAVED_CLS_HIDDEN      1..L        i       classifier hidden layer i (mpn.py:114)
 *   MPNHIP_SAVED_FLOW_HIDDEN     1..L        i       flow_out / flow_in MLP hidden layer i of each edge's direction (mpn.py:88,95)
 *   MPNHIP_SAVED_MSG             1..L        -       the messages the aggregation reads (mpn.py:89,96)
 *   MPNHIP_SAVED_AGG             1..L        -       [flow_in | flow_out] (mpn.py:97)
 *   MPNHIP_SAVED_ARGMAX          1..L        -       max aggregation: original edge id chosen per (node, column), -1 = none */
#define MPNHIP_SAVED_ENC_NODE 0
#define MPNHIP_SAVED_ENC_EDGE 1
#define MPNHIP_SAVED_X 2
#define MPNHIP_SAVED_E 3
#define MPNHIP_SAVED_EDGE_HIDDEN 4
#define MPNHIP_SAVED_CLS_HIDDEN 5
#define MPNHIP_SAVED_FLOW_HIDDEN 6
#define MPNHIP_SAVED_MSG 7
#define MPNHIP_SAVED_AGG 8
#define MPNHIP_SAVED_ARGMAX 9
int mpnhip_debug_saved(const mpnhip_model* model, const void* graph_buf, int n_nodes, int64_t n_edges, const void* fwd_workspace,
                       size_t fwd_workspace_bytes, int what, int step, int layer, float* out, int64_t* rows, int* width,
                       void* stream);
/* The same for mpnhip_backward: one [rows, width] block of the pre-activation gradients it keeps per step in its workspace
 * (for the batched weight-gradient products), in the order it is stored (edges: sorted order).  step = 1 .. num_enc_steps. */
#define MPNHIP_BWD_SAVED_DZ_NODE 0 /* [N, dn]   gradient of the node update's pre-activation */
#define MPNHIP_BWD_SAVED_DP 1      /* [N, pw]   gradient of the per-node projections */
#define MPNHIP_BWD_SAVED_DZ_FLOW 2 /* layer i of the flow MLPs (last layer: the masked message gradient) */
#define MPNHIP_BWD_SAVED_DZ_EDGE 3 /* layer i of the edge MLP */
#define MPNHIP_BWD_SAVED_DZ_CLS 4  /* hidden layer i of the classifier */
int mpnhip_debug_backward_saved(const mpnhip_model* model, int n_nodes, int64_t n_edges, const void* bwd_workspace,
                                size_t bwd_workspace_bytes, int what, int step, int layer, float* out, int64_t* rows,
                                int* width, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Graph preparation -- replaces the six boolean-mask indexings per step of
 * TimeAwareNodeModel.forward (models/mpn.py:85-87,91-93) and the implicit index structures behind
 * torch_scatter / index_put_.  Done once per edge_index, reused by all steps, forward and backward.
 *
 * edge_index: int64 [2,E] row-major (row = edge_index[0], col = edge_index[1]), as the reference's
 * Graph.edge_index (data/mot_graph.py:312).  Edges are stably sorted by (direction, row) with
 * direction 0 = row<col (flow_out), 1 = row>col (flow_in), 2 = row==col (in neither aggregate,
 * mpn.py:85,91).  The prepared graph lives in `graph_buf` (mpnhip_graph_bytes) and is opaque.
 * Out-of-range indices set an error flag readable with mpnhip_graph_status (which synchronises).
 * ------------------------------------------------------------------------------------------- */
size_t mpnhip_graph_bytes(int n_nodes, int64_t n_edges);
size_t mpnhip_graph_prep_workspace_bytes(int n_nodes, int64_t n_edges);
int mpnhip_graph_prep(const int64_t* edge_index, int n_nodes, int64_t n_edges, void* graph_buf, size_t graph_bytes,
                      void* workspace, size_t workspace_bytes, void* stream);
/* The primary (direction, row) order only -- all that mpnhip_forward / mpnhip_meta_layer_forward / the forward of
 * mpnhip_attention_aggregate read (a quarter of the sorting work; sliding-window inference prepares a graph per window).
 * mpnhip_backward, mpnhip_attention_aggregate_backward and mpnhip_step_metrics need mpnhip_graph_prep. */
int mpnhip_graph_prep_forward(const int64_t* edge_index, int n_nodes, int64_t n_edges, void* graph_buf, size_t graph_bytes,
                              void* workspace, size_t workspace_bytes, void* stream);
/* Synchronising debug helper: host copy of {error_flag, E_flow_out, E_flow_in, E_self}. */
int mpnhip_graph_status(const void* graph_buf, int n_nodes, int64_t n_edges, int32_t status[4], void* stream);

/* ---------------------------------------------------------------------------------------------
 * MOTMPNet.forward hot path (models/mpn.py:349-392 minus the x_ext / mask lines): encoder, then
 * num_enc_steps x { reattach (:369-373), MetaLayer (:376 -> :33-54), classifier (:377 -> :114) }.
 *
 * x         [N, enc_node.in_dim]   node inputs after the avg-pool of mpn.py:351-352
 * edge_attr [E, enc_edge.in_dim]   in edge_index order
 * logits    [max(L,1), E]          OUT: classifier output of EVERY step, edge_index order (the
 *                                  reference's classified_edges are the last num_class_steps rows)
 * x_out [N,dn], e_out [E,de]       OUT, optional (NULL): final latent node / edge features
 * save_for_backward != 0 keeps every step's activations in `workspace` for mpnhip_backward.
 * ------------------------------------------------------------------------------------------- */
size_t mpnhip_forward_workspace_bytes(const mpnhip_model* model, int n_nodes, int64_t n_edges, int save_for_backward);
int mpnhip_forward(const mpnhip_model* model, const void* graph_buf, int n_nodes, int64_t n_edges, const float* x,
                   const float* edge_attr, float* logits, float* x_out, float* e_out, void* workspace,
                   size_t workspace_bytes, int save_for_backward, void* stream);

/* Autograd of the above (what torch.autograd derives for mpn.py:349-392).
 * grad_logits [max(L,1), E] incoming gradient for every step's logits (zeros where unused);
 * grad_x_out / grad_e_out optional incoming gradients for the final latents (NULL = 0);
 * grad_x [N, enc_node.in_dim], grad_edge_attr [E, enc_edge.in_dim]: OUT, optional (overwritten).
 * Parameter gradients are accumulated into model->*.grad_weight / grad_bias.
 * `workspace` must be the buffer the matching forward ran with save_for_backward = 1. */
size_t mpnhip_backward_workspace_bytes(const mpnhip_model* model, int n_nodes, int64_t n_edges);
int mpnhip_backward(const mpnhip_model* model, const void* graph_buf, int n_nodes, int64_t n_edges, const float* x,
                    const float* edge_attr, const float* grad_logits, const float* grad_x_out,
                    const float* grad_e_out, float* grad_x, float* grad_edge_attr, void* fwd_workspace,
                    size_t fwd_workspace_bytes, void* bwd_workspace, size_t bwd_workspace_bytes, void* stream);

/* Data-parallel training: the same backward with flags.  MPNHIP_BWD_DEFER_SIDE_JOIN: when the
 * backward runs its weight-gradient groups on the library's side stream (mpnhip_backward_uses_side_stream(model) == 1: four
 * or more steps), do NOT make `stream` wait for that stream before returning.  On return the gradients of the message-passing
 * modules and of the classifier are complete in SIDE-stream order, the encoder's gradients and grad_x / grad_edge_attr in
 * `stream` order: the trainer enqueues the all-reduce of the first bucket on mpnhip_side_stream() -- it then overlaps the
 * encoder's backward -- and joins with mpnhip_side_stream_join(stream) (or its collective's own wait) before it reads them. */
#define MPNHIP_BWD_DEFER_SIDE_JOIN 1
int mpnhip_backward_flags(const mpnhip_model* model, const void* graph_buf, int n_nodes, int64_t n_edges, const float* x,
                          const float* edge_attr, const float* grad_logits, const float* grad_x_out, const float* grad_e_out,
                          float* grad_x, float* grad_edge_attr, void* fwd_workspace, size_t fwd_workspace_bytes,
                          void* bwd_workspace, size_t bwd_workspace_bytes, int flags, void* stream);
int mpnhip_backward_uses_side_stream(const mpnhip_model* model);
void* mpnhip_side_stream(void);              /* hipStream_t of the current device's side stream (created on first use) */
int mpnhip_side_stream_join(void* stream);   /* `stream` waits for everything enqueued on the side stream so far */

/* ---------------------------------------------------------------------------------------------
 * Operator level.
 * ------------------------------------------------------------------------------------------- */

/* MetaLayer.forward(x, edge_index, edge_attr) -> (x', e')  (models/mpn.py:33-54) with
 * x [N, nf*dn], e [E, ef*de] (already re-attached, edge_index order); x_new [N,dn], e_new [E,de].
 * Only model->edge / flow_in / flow_out / node and dn, de, agg, reattach_* are read. */
size_t mpnhip_meta_layer_workspace_bytes(const mpnhip_model* model, int n_nodes, int64_t n_edges);
int mpnhip_meta_layer_forward(const mpnhip_model* model, const void* graph_buf, int n_nodes, int64_t n_edges,
                              const float* x, const float* e, float* x_new, float* e_new, void* workspace,
                              size_t workspace_bytes, void* stream);

/* node_agg_fn(out, row, x_size) (models/mpn.py:266-273): out[i] = AGG over {j : row[j] == i} src[j];
 * src [M, dim], row int64 [M] (any order), out [x_size, dim]; empty -> 0; max returns values only.
 * argmax (optional, int32 [x_size, dim]): for MAX the source row chosen (first maximum in index
 * order, -1 for empty segments), as torch_scatter's CPU kernel picks it. */
size_t mpnhip_segment_reduce_workspace_bytes(int64_t m, int x_size);
int mpnhip_segment_reduce(const float* src, const int64_t* row, int64_t m, int dim, int x_size, int agg, float* out,
                          int32_t* argmax, void* workspace, size_t workspace_bytes, void* stream);

/* y = act(x W^T + b): one layer of models/mlp.py (nn.Linear + optional ReLU).
 * x [M,K] (row stride ldx), w [N,K], b [N] or NULL, y [M,N] (row stride ldy). */
int mpnhip_linear(const float* x, int64_t ldx, const float* w, const float* b, float* y, int64_t ldy, int64_t m, int n,
                  int k, int relu, void* stream);

/* One nn.Linear (+ ReLU) of models/mlp.py:27 in BASELINE.json's configs[4] arithmetic -- what mpnhip_forward / mpnhip_backward run
 * for the node-side products of models/mpn.py:69,87,93,97-99 when mpnhip_model.precision == MPNHIP_PREC_BF16 (round 5: the 128 x 128
 * tiled kernel, csrc/gemm_bf16.hip):
 *     y[m][n] = mask( act( sum_k bf16(xcat[m][k]) * bf16(w[n][k]) + b[n] + c_in[m][n] ) (+ y[m][n]) ),  fp32 accumulation,
 * xcat = [x | x2] (torch.cat of the re-attached initial and the current features, mpn.py:369-373, never materialised).
 * Operands are fp32 rows rounded to bf16 (RNE) as they are staged, or ALREADY bf16 rows in memory (x_bf16 / w_bf16 != 0: the
 * pointers are bfloat16 bit patterns, leading dims count elements; K, ksplit and the leading dims multiples of 8, 16-byte bases).
 * b, c_in, mask (value kept where mask > 0), x2 and y16 (a bf16 mirror of the result) may be NULL.  n % 4 == 0, k % 4 == 0. */
typedef struct mpnhip_linear_bf16_args {
    const void* x;  int64_t ldx;        /* [m, ksplit] */
    const void* x2; int64_t ldx2;       /* [m, k - ksplit] or NULL (then ksplit == k) */
    const void* w;  int64_t ldw;        /* [n, k] */
    const float* b;                     /* [n] */
    const float* c_in; int64_t ldc_in;  /* [m, n] added before the activation */
    const float* mask; int64_t ldmask;  /* [m, n] */
    float* y; int64_t ldy;              /* [m, n] */
    uint16_t* y16; int64_t ldy16;       /* [m, n] bf16 mirror of y */
    int64_t m;
    int n, k, ksplit;
    int x_bf16, w_bf16, relu, accumulate;
} mpnhip_linear_bf16_args;
int mpnhip_linear_bf16(const mpnhip_linear_bf16_args* args, void* stream);
/* dst[i] = bf16(src[i]) (round to nearest even; NaN stays NaN): the packed weight images / feature mirrors of that mode. n % 4 == 0. */
int mpnhip_to_bf16(const float* src, uint16_t* dst, int64_t n, void* stream);

/* The weight / bias gradient autograd derives for one nn.Linear of models/mlp.py, batched over the
 * message-passing steps that share the weight:  grad_w[o][c] += sum_b sum_m dZ[b][m][o] * H[b][m][c],  grad_b[o] += sum dZ[b][m][o].
 * dZ [nbatch][rows][n_out] (gradient at the layer's pre-activation), H [nbatch][rows][k_in] (the layer's input), both dense
 * row-major; grad_w [n_out][k_in], grad_b [n_out] or NULL.  Fixed summation order (no float atomics): bitwise reproducible. */
size_t mpnhip_weight_grad_workspace_bytes(int n_out, int k_in, int64_t rows, int nbatch);
int mpnhip_weight_grad(const float* dZ, const float* H, int64_t rows, int n_out, int k_in, int nbatch, float* grad_w, float* grad_b,
                       void* workspace, size_t workspace_bytes, void* stream);
/* The same with the operand form chosen: MPNHIP_PREC_FP32 (fp32 MFMAs; what mpnhip_weight_grad does) or MPNHIP_PREC_FP32_SPLIT
 * (every fp32 operand as the exact sum of three bf16 pieces, six piece products with fp32 accumulate: the same accuracy class;
 * what mpnhip_backward uses for a model in that precision). */
int mpnhip_weight_grad_prec(const float* dZ, const float* H, int64_t rows, int n_out, int k_in, int nbatch, int precision, float* grad_w,
                            float* grad_b, void* workspace, size_t workspace_bytes, void* stream);
/* The same product over operands that are ALREADY bf16 rows in memory (round 4: what the bf16-operand training path keeps -- the
 * backward chain kernel's dZ blocks, the forward chain kernel's saved activations; mlp.py:27-28 under autograd in BASELINE.json's
 * configs[4] arithmetic): dZ [nbatch][rows][n_out], H [nbatch][rows][k_in] as bfloat16 bit patterns, n_out and k_in multiples of 4
 * (or a narrow shape, k_in <= 32 and n_out <= 32), fp32 accumulation; grad_w / grad_b fp32, "+=". */
size_t mpnhip_weight_grad_bf16_rows_workspace_bytes(int n_out, int k_in, int64_t rows, int nbatch);
int mpnhip_weight_grad_bf16_rows(const uint16_t* dZ, const uint16_t* H, int64_t rows, int n_out, int k_in, int nbatch, float* grad_w,
                                 float* grad_b, void* workspace, size_t workspace_bytes, void* stream);

/* The gradient of node_agg_fn (models/mpn.py:266-273; torch_scatter's scatter_add / scatter_mean / scatter_max backward), gather
 * form: grad_src[j] = grad_out[row[j]] (sum), / count[row[j]] (mean; count int32 [x_size]), or only where argmax[row[j]][d] == j
 * (max; `argmax` as mpnhip_segment_reduce returned it).  grad_out [x_size, dim], grad_src [M, dim]. */
int mpnhip_segment_reduce_backward(const float* grad_out, const int64_t* row, const int32_t* argmax, const int32_t* count, int64_t m,
                                   int dim, int x_size, int agg, float* grad_src, void* stream);

/* nn.BatchNorm1d (TRAINING mode: batch statistics) -> nn.ReLU -> nn.Dropout as models/mlp.py:12-23 stacks them behind each
 * nn.Linear, and their gradients -- the layer-by-layer training path of a model built with use_batchnorm / dropout_p
 * (mpntrackseg_amd/modular.py; no shipped configuration enables them, configs/tracking_cfg.yaml:150-167; in eval mode BatchNorm
 * folds into the Linear layers and the fused path runs).  z [M, N] dense = the Linear's output.
 *   forward : use_bn: mean / biased variance over the M rows (two passes), y = relu(gamma (z - mean) invstd + beta) keep / (1 - p);
 *             running_mean / running_var (may be NULL) updated like nn.BatchNorm1d (momentum, unbiased variance); save_mean /
 *             save_invstd [N] receive the batch statistics for the backward.  M == 1 with use_bn is refused like torch does.
 *             keep(r, c) = hash(seed, r N + c) >= p: the backward regenerates it from the same seed, nothing is stored.
 *   backward: dz [M, N], dgamma / dbeta [N] (may be NULL; written, not accumulated).
 * Column sums in a fixed order (no float atomics): bitwise reproducible.  workspace: mpnhip_bn_dropout_workspace_bytes. */
size_t mpnhip_bn_dropout_workspace_bytes(int64_t m, int n);
int mpnhip_bn_relu_dropout_forward(const float* z, int64_t m, int n, int use_bn, const float* gamma, const float* beta,
                                   float* running_mean, float* running_var, float momentum, float eps, int relu, float dropout_p,
                                   uint64_t seed, float* y, float* save_mean, float* save_invstd, void* workspace,
                                   size_t workspace_bytes, void* stream);
int mpnhip_bn_relu_dropout_backward(const float* dy, const float* z, int64_t m, int n, int use_bn, const float* gamma,
                                    const float* beta, const float* save_mean, const float* save_invstd, int relu, float dropout_p,
                                    uint64_t seed, float* dz, float* dgamma, float* dbeta, void* workspace, size_t workspace_bytes,
                                    void* stream);

/* MLP.forward (models/mlp.py:27-28): all layers; scratch [2, M, max(out_dims)] floats. */
size_t mpnhip_mlp_workspace_bytes(const mpnhip_mlp* mlp, int64_t m);
int mpnhip_mlp_forward(const mpnhip_mlp* mlp, const float* x, float* y, int64_t m, void* workspace,
                       size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * The callers' next step, device side, no host sync.
 * ------------------------------------------------------------------------------------------- */

/* Tracking term of MOTNeuralSolver._compute_loss (pl_module/pl_module.py:88-107):
 *   loss = weight * sum_{s >= first_step} BCEWithLogits(logits[s], labels, pos_weight = #neg / #pos), mean over edges.
 * logits [n_steps, E] (every step, as mpnhip_forward returns them), labels [E] in {0,1} (edge_index order),
 * first_step = num_enc_steps - num_class_steps (steps before it carry no loss).
 * loss_out [1 + n_steps] (device): total, then per step.  grad_logits [n_steps, E]: d loss / d logits (zeros for
 * the unclassified steps) -- exactly the grad_logits argument of mpnhip_backward. */
size_t mpnhip_tracking_loss_workspace_bytes(int n_steps, int64_t n_edges);
int mpnhip_tracking_loss(const float* logits, const float* labels, int n_steps, int64_t n_edges, int first_step,
                         float weight, float* loss_out, float* grad_logits, void* workspace, size_t workspace_bytes,
                         void* stream);

/* The same loss over the n_graphs graphs of ONE block-diagonal batch (torch_geometric's Batch; edge_graph[e] = graph of edge e, int32,
 * edge_index order): every graph its own pos_weight and its own mean, as the reference computes them graph by graph with
 * batch_size 1, and the n_graphs losses AVERAGED -- what accumulate_grad_batches = n_graphs backward passes add up to
 * (configs/tracking_cfg.yaml:3-4, pl_module.py:88-107).  loss_out [1 + n_steps]; grad_logits [n_steps, E]. */
size_t mpnhip_tracking_loss_graphs_workspace_bytes(int n_steps, int64_t n_edges, int n_graphs);
int mpnhip_tracking_loss_graphs(const float* logits, const float* labels, const int32_t* edge_graph, int n_graphs, int n_steps,
                                int64_t n_edges, int first_step, float weight, float* loss_out, float* grad_logits, void* workspace,
                                size_t workspace_bytes, void* stream);

/* compute_perform_metrics (utils/evaluation.py:416-437) on the last step's logits [E] (edge_index order):
 * counts[0..3] = TP, FP, TN, FN of (logit > 0) vs labels (fast_compute_class_metric, :340-366);
 * counts[4..5] = nodes whose outgoing / incoming flow exceeds 1, counts[6..7] = nodes that have an outgoing /
 * incoming constraint (compute_constr_satisfaction_rate, :370-414, undirected_edges = True).
 * counts: int32[8] on the device; the caller reads them back when it wants the ratios. */
int mpnhip_step_metrics(const void* graph_buf, int n_nodes, int64_t n_edges, const float* logits, const float* labels,
                        int32_t* counts, void* stream);

/* Mask branch: the neighbour aggregation of TimeAwareAttentionModel.forward
 * (models/mpn.py:117-134): per (node, direction) segment w = scatter_softmax(logits) and
 * out_dir[n] = sum_j w_j x[col_j], x [N, feat] with feat = C*H*W (64*14*14).  logits [E] in edge_index order (the
 * classifier output of this step); out_in / out_out [N, feat] (flow_in: row > col, flow_out: row < col);
 * weights [E] (sorted edge order, optional) keeps w for the backward. */
int mpnhip_attention_aggregate(const void* graph_buf, int n_nodes, int64_t n_edges, const float* x, int64_t feat,
                               const float* logits, float* out_in, float* out_out, float* weights, void* stream);
/* Its autograd: grad_x [N, feat] (overwritten, or += when accumulate_grad_x) and grad_logits [E] (+=, edge_index
 * order) from grad_in / grad_out [N, feat]; workspace_dw: E floats. */
int mpnhip_attention_aggregate_backward(const void* graph_buf, int n_nodes, int64_t n_edges, const float* x, int64_t feat,
                                        const float* weights, const float* grad_in, const float* grad_out, float* grad_x,
                                        int accumulate_grad_x, float* grad_logits, float* workspace_dw, void* stream);

/* nn.AdaptiveAvgPool2d((1,1)) + view (models/mpn.py:252,351-352): x [rows, hw] -> y [rows] = mean over
 * the hw contiguous spatial positions (rows = N * C). */
int mpnhip_avgpool(const float* x, int64_t rows, int hw, float* y, void* stream);

/* torch.optim.Adam(lr, betas, eps, weight_decay).step() (the optimizer of pl_module.py:76-77, configs/tracking_cfg.yaml:6-10)
 * over FLAT fp32 buffers of n elements: parameters, gradients (as the backward / all-reduce left them) and the two
 * moment buffers (zero before step 1); step = 1, 2, ... counts the calls. */
int mpnhip_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1,
                     float beta2, float eps, float weight_decay, int step, void* stream);
/* The same, skipped as a whole when *skip_flag (device float, may be NULL) is non-zero.  Data-parallel training (one graph per
 * rank, scripts/train.py:65-77 executed in space): every rank adds "my graph's edge_index left [0, N)" (the reference's
 * IndexError, mpn.py:69) to one spare element of the gradient all-reduce; with the reduced element as skip_flag no rank steps
 * on gradients of a clamped graph, without a host read on the ranks whose own graph is fine. */
int mpnhip_adam_step_guarded(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1,
                             float beta2, float eps, float weight_decay, int step, const float* skip_flag, void* stream);
/* The same with the step count kept where the skip decision is made: `calls` = 1, 2, ... counts the CALLS, *skipped_calls (device
 * int, zero before the first call, may be NULL) the calls that were skipped; a skipped call adds one to it and an applied update uses
 * the bias corrections of step calls - *skipped_calls -- torch.optim.Adam's count of applied steps -- so a skipped step followed by a
 * good one equals ONE reference step (pl_module.py:76-77).  skipped_calls == NULL: mpnhip_adam_step_guarded. */
int mpnhip_adam_step_counted(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1,
                             float beta2, float eps, float weight_decay, int calls, const float* skip_flag, int* skipped_calls,
                             void* stream);

/* ---------------------------------------------------------------------------------------------
 * Graph construction on the device: what MOTGraph.construct_graph_object
 * (data/mot_graph.py:283-317) computes before the model runs.  All indices int64 like the reference's tensors.
 * ------------------------------------------------------------------------------------------- */
/* get_time_valid_conn_ixs(frame_num, max_frame_dist, return_undirected=True) (utils/graph.py:6-37): the pairs
 * (i, j), i < j, whose frames differ and are at most max_frame_dist apart (max_frame_dist < 0 = the reference's
 * 'max'), in the reference's order (ascending i, then j).  Two calls, because the caller allocates the result:
 *   count: offsets [N + 1] (device) <- exclusive scan of the per-node pair counts; offsets[N] = number of pairs
 *   fill : edge_ixs [2, n_pairs] row-major, n_pairs = offsets[N] read back by the caller. */
size_t mpnhip_time_valid_conn_workspace_bytes(int n_nodes);
int mpnhip_time_valid_conn_count(const int64_t* frame_num, int n_nodes, int64_t max_frame_dist, int64_t* offsets,
                                 void* workspace, size_t workspace_bytes, void* stream);
int mpnhip_time_valid_conn_fill(const int64_t* frame_num, int n_nodes, int64_t max_frame_dist, const int64_t* offsets,
                                int64_t n_pairs, int64_t* edge_ixs, void* stream);
/* compute_edge_feats_dict (utils/graph.py:90-124): edge_feats [E, 5] = secs_time_dists, norm_feet_x_dists,
 * norm_feet_y_dists, bb_height_dists, bb_width_dists (the dict's order) for edge_ixs [2, E]; per-node columns of
 * the detection frame: frame_num int64 [N], bb_height / bb_width / feet_x / feet_y float32 [N]. */
int mpnhip_edge_features(const int64_t* edge_ixs, int64_t n_edges, int n_nodes, const int64_t* frame_num, float fps,
                         const float* bb_height, const float* bb_width, const float* feet_x, const float* feet_y,
                         float* edge_feats, void* stream);
/* F.pairwise_distance(emb[edge_ixs[0]], emb[edge_ixs[1]]) (data/mot_graph.py:298-301; p = 2, eps as given, the
 * reference uses torch's default 1e-6): dist [E]; emb [N, dim] with row stride ld. */
int mpnhip_pairwise_distance(const float* emb, int64_t ld, int dim, const int64_t* edge_ixs, int64_t n_edges, float eps,
                             float* dist, void* stream);

/* load_precomputed_embeddings (utils/rgb.py:150-188) once the per-frame files are in device memory: stored [n_stored, ld]
 * fp32 rows whose element 0 carries the detection id (1D files: column 0; 3D files [n, 1 + C, H, W]: element [0, 0, 0], i.e.
 * ld = (1 + C) H W).  keep [n_stored] = 1 iff that id occurs in det_ids_sorted [n_det] (int32 ascending): the np.isin
 * filter of rgb.py:179 / :185.  The kept rows without their id column / channel are then one mpnhip_gather_rows. */
int mpnhip_embedding_keep(const float* stored, int64_t ld, int64_t n_stored, const int32_t* det_ids_sorted, int64_t n_det,
                          unsigned char* keep, void* stream);
/* the order assertion of rgb.py:180 / :186: mismatch[0] (device int32) = number of j < n with id(stored[rows[j]]) != det_ids[j] */
int mpnhip_embedding_check(const float* stored, int64_t ld, const int32_t* rows, int64_t n, const int32_t* det_ids,
                           int32_t* mismatch, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Sliding-window inference: MPNTracker._predict_edges_and_masks /
 * _evaluate_graph_in_batches (tracker/mpn_tracker.py:96-210) around mpnhip_forward.
 * ------------------------------------------------------------------------------------------- */
/* get_knn_mask (utils/graph.py:40-87): pruned_mask [E] (1 = keep) for edge_ixs [2, E] with distances pwise_dist [E].
 * symmetric_edges != 0: edge_ixs lists both directions of every pair (inference); 0: one direction (training).
 * Ties in distance rank by column index (stable argsort); the reference leaves them to torch.argsort. */
size_t mpnhip_knn_mask_workspace_bytes(int64_t n_edges, int symmetric_edges);
int mpnhip_knn_mask(const float* pwise_dist, const int64_t* edge_ixs, int n_nodes, int64_t n_edges, int top_k_nns,
                    int reciprocal_k_nns, int symmetric_edges, unsigned char* pruned_mask, void* workspace,
                    size_t workspace_bytes, void* stream);
/* edges_mask of a window (mpn_tracker.py:171-173): flags [E] = both end points in [node_begin, node_end). */
int mpnhip_window_flags(const int64_t* edge_index, int64_t n_edges, int64_t node_begin, int64_t node_end,
                        unsigned char* flags, void* stream);
/* torch.where(flags)[0] as int32 ids (ascending) + their number (device int32), e.g. tensor[mask] selections. */
size_t mpnhip_compact_workspace_bytes(int64_t n);
int mpnhip_compact(const unsigned char* flags, int64_t n, int32_t* ids, int32_t* count, void* workspace,
                   size_t workspace_bytes, void* stream);
/* out [n, dim] = src[ids] (rows of stride ld);  out [2, n] = edge_index[:, ids] - node_begin. */
int mpnhip_gather_rows(const float* src, int64_t ld, const int32_t* ids, int64_t n, int dim, float* out, void* stream);
int mpnhip_gather_edges(const int64_t* edge_index, int64_t n_edges, const int32_t* ids, int64_t n, int64_t node_begin,
                        int64_t* out, void* stream);
/* mpn_tracker.py:126-141,188-190: overall_edge_preds[full id] += sigmoid(logit) for the window's kept edges and
 * overall_num_preds += 1 for the kept edges (or, with set_pruned_edges_to_inactive, for every edge of the window).
 * logits [n_kept]; kept_ids [n_kept] index the window's edge list (NULL = identity); window_ids [n_window] index
 * the full sequence graph's edges. */
int mpnhip_window_accumulate(const float* logits, const int32_t* kept_ids, int64_t n_kept, const int32_t* window_ids,
                             int64_t n_window, int set_pruned_edges_to_inactive, float* overall_edge_preds,
                             float* overall_num_preds, void* stream);
/* final_edge_preds = overall_edge_preds / overall_num_preds, NaN -> 0 (mpn_tracker.py:195-197). */
int mpnhip_average_preds(const float* overall_preds, const float* overall_num, int64_t n, float* final_preds, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Measurement helpers used by bench.py (HIP events on the launch stream; these synchronise).
 * ------------------------------------------------------------------------------------------- */
/* In-stream kernel timing of the real hot path: while enabled, mpnhip_forward brackets (a) the first-layer
 * edge-MLP GEMM (the dominant MFMA kernel) and (b) the aggregation kernel (the HBM-bound one) of every
 * message-passing step with HIP events on the launch stream.  mpnhip_profile_read synchronises, returns
 * the average duration (us) and launch count of each since the last read, and resets the counters.
 * The empty-pair cost applies to the default (NULL) stream the calibration pairs are recorded on. */
int mpnhip_profile_enable(int on);   /* 0 = off, 1 = time every launch of the two kernels, n > 1 = every n-th launch */
/* 1 when mpnhip_forward evaluates the per-edge chain (edge MLP + classifier + flow MLPs) of this model with the
 * fused edge_chain kernel (then THAT kernel is the one bracketed as "gemm" by the profile hooks), 2 when it does so with
 * the bf16-operand chain kernel (MPNHIP_PREC_BF16, edge_chain_bf16.hip), else 0. */
int mpnhip_edge_chain_active(const mpnhip_model* model);
int mpnhip_profile_read(float* gemm_avg_us, int* gemm_launches, float* agg_avg_us, int* agg_launches,
                        float* empty_pair_us /* cost of an event pair with nothing between, for calibration */);

/* The same hooks, one kernel kind at a time: MPNHIP_PROF_CHAIN = the kernel mpnhip_profile_read reports as "gemm" (forward),
 * _AGG = the aggregation kernel, _CHAIN_BWD = the fused backward chain of mpnhip_backward, _WEIGHT_GRAD = the MFMA
 * weight-gradient product kernel (gemm_tn_kernel; launched on the library's side stream as well as on the caller's -- the events
 * are attached to each dispatch on whatever stream it goes to).  avg_work: average algorithmic flops per timed launch (only
 * _WEIGHT_GRAD reports it: its launches differ in shape).  Synchronises and resets the kind's counters. */
#define MPNHIP_PROF_CHAIN 0
#define MPNHIP_PROF_AGG 1
#define MPNHIP_PROF_CHAIN_BWD 2
#define MPNHIP_PROF_WEIGHT_GRAD 3
int mpnhip_profile_read_kind(int kind, float* avg_us, int* launches, double* avg_work);

/* Average duration in microseconds of `iters` back-to-back launches of the aggregation kernel on
 * a prepared graph: src [E, dim] in SORTED edge order, out [N, 2*dim]. */
int mpnhip_time_aggregate(const void* graph_buf, int n_nodes, int64_t n_edges, const float* src, int dim, int agg,
                          float* out, int iters, float* avg_us, void* stream);
/* Average duration (us) of `iters` back-to-back mpnhip_weight_grad calls (product + slab sum). */
int mpnhip_time_weight_grad(const float* dZ, const float* H, int64_t rows, int n_out, int k_in, int nbatch, float* grad_w, float* grad_b,
                            void* workspace, size_t workspace_bytes, int iters, float* avg_us, void* stream);
int mpnhip_time_weight_grad_prec(const float* dZ, const float* H, int64_t rows, int n_out, int k_in, int nbatch, int precision, float* grad_w,
                                 float* grad_b, void* workspace, size_t workspace_bytes, int iters, float* avg_us, void* stream);
/* Average duration (us) of `iters` launches of y = relu(x W^T + b). */
int mpnhip_time_linear(const float* x, const float* w, const float* b, float* y, int64_t m, int n, int k, int iters,
                       float* avg_us, void* stream);

/* Average duration (us) of `iters` launches of mpnhip_linear_bf16. */
int mpnhip_time_linear_bf16(const mpnhip_linear_bf16_args* args, int iters, float* avg_us, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MPNHIP_H */
