// Model checks and workspace planning shared by the forward (mpn.hip) and backward (backward.hip)
// orchestration.  Everything here is host code; offsets are a pure function of (model dims, N, E, save).
#pragma once
#include <cstdlib>
#include <cstring>

#include "common.h"
#include "edge_chain.h"

namespace mpnhip {

// ------------------------------------------------------------------------------------ model checks
struct Dims {
    int dn, de, nf, ef;
    int he;   // first hidden width of the edge MLP
    int hn;   // first hidden width of the flow MLPs
    int pw;   // width of the per-node projection block: 2 he + 2 hn
    int kx;   // nf * dn
    int ke;   // ef * de
    int L;
};

static inline int mlp_ok(const mpnhip_mlp& m, const char* name, bool need_ptrs) {
    MPN_CHECK_ARG(m.n_layers >= 1 && m.n_layers <= MPNHIP_MAX_LAYERS, "%s: n_layers %d", name, m.n_layers);
    MPN_CHECK_ARG(m.in_dim >= 1, "%s: in_dim %d", name, m.in_dim);
    for (int i = 0; i < m.n_layers; ++i) {
        MPN_CHECK_ARG(m.out_dims[i] >= 1, "%s: out_dims[%d] = %d", name, i, m.out_dims[i]);
        if (need_ptrs) MPN_CHECK_ARG(m.weight[i] && m.bias[i], "%s: null weight/bias in layer %d", name, i);
    }
    return MPNHIP_OK;
}

static inline int check_core(const mpnhip_model& m, Dims* d, bool ptrs = true) {
    MPN_CHECK_ARG(m.dn >= 1 && m.de >= 1, "model: dn/de");
    MPN_CHECK_ARG(m.agg >= 0 && m.agg <= 2, "model: node_agg_fn code %d (reference asserts 'mean'|'max'|'sum', mpn.py:264)", m.agg);
    d->dn = m.dn;
    d->de = m.de;
    d->nf = m.reattach_nodes ? 2 : 1;
    d->ef = m.reattach_edges ? 2 : 1;
    d->kx = d->nf * d->dn;
    d->ke = d->ef * d->de;
    d->L = m.num_enc_steps;
    MPN_TRY(mlp_ok(m.edge, "edge_model", ptrs));
    MPN_TRY(mlp_ok(m.flow_in, "flow_in_model", ptrs));
    MPN_TRY(mlp_ok(m.flow_out, "flow_out_model", ptrs));
    MPN_TRY(mlp_ok(m.node, "node_model", ptrs));
    MPN_CHECK_ARG(m.edge.in_dim == 2 * d->kx + d->ke, "edge_model in_dim %d != %d (mpn.py:282-283)", m.edge.in_dim, 2 * d->kx + d->ke);
    MPN_CHECK_ARG(m.edge.out_dims[m.edge.n_layers - 1] == d->de, "edge_model must end at edge_out_dim");
    MPN_CHECK_ARG(m.flow_in.in_dim == d->kx + d->de && m.flow_out.in_dim == d->kx + d->de, "flow model in_dim != %d (mpn.py:285)", d->kx + d->de);
    MPN_CHECK_ARG(m.flow_in.n_layers == m.flow_out.n_layers, "flow_in / flow_out depth differ");
    for (int i = 0; i < m.flow_in.n_layers; ++i)
        MPN_CHECK_ARG(m.flow_in.out_dims[i] == m.flow_out.out_dims[i], "flow_in / flow_out dims differ");
    MPN_CHECK_ARG(m.flow_in.out_dims[m.flow_in.n_layers - 1] == d->dn, "flow models must end at node_out_dim");
    MPN_CHECK_ARG(m.node.n_layers == 1 && m.node.in_dim == 2 * d->dn && m.node.out_dims[0] == d->dn, "node_model must be Linear(2dn, dn) (mpn.py:309)");
    d->he = m.edge.out_dims[0];
    d->hn = m.flow_in.out_dims[0];
    d->pw = 2 * d->he + 2 * d->hn;
    return MPNHIP_OK;
}

static inline int check_full(const mpnhip_model& m, Dims* d, bool ptrs = true) {
    MPN_TRY(check_core(m, d, ptrs));
    MPN_CHECK_ARG(m.num_enc_steps >= 0, "model: num_enc_steps");
    MPN_TRY(mlp_ok(m.enc_node, "encoder.node_model", ptrs));
    MPN_TRY(mlp_ok(m.enc_edge, "encoder.edge_model", ptrs));
    MPN_TRY(mlp_ok(m.classifier, "classifier.edge_model", ptrs));
    MPN_CHECK_ARG(m.enc_node.out_dims[m.enc_node.n_layers - 1] == d->dn, "encoder node_out_dim mismatch");
    MPN_CHECK_ARG(m.enc_edge.out_dims[m.enc_edge.n_layers - 1] == d->de, "encoder edge_out_dim mismatch");
    MPN_CHECK_ARG(m.classifier.in_dim == d->de, "classifier edge_in_dim != edge_out_dim");
    MPN_CHECK_ARG(m.classifier.out_dims[m.classifier.n_layers - 1] == 1, "classifier must end with out dim 1");
    return MPNHIP_OK;
}

static inline int max_hidden(const mpnhip_mlp& m) {
    int mx = 0;
    for (int i = 0; i + 1 < m.n_layers; ++i) mx = m.out_dims[i] > mx ? m.out_dims[i] : mx;
    return mx;
}
static inline int64_t sum_hidden(const mpnhip_mlp& m) {
    int64_t s = 0;
    for (int i = 0; i + 1 < m.n_layers; ++i) s += m.out_dims[i];
    return s;
}

// the bf16-operand chain kernels (edge_chain_bf16.hip, edge_chain_bf16_bwd.hip) cover this model's per-edge modules
static inline bool chain_bf16_ok(const mpnhip_model& m, const Dims& d) {
    if (getenv("MPNHIP_NO_CHAIN") || getenv("MPNHIP_NO_CHAIN_BF16")) return false;
    return m.precision == MPNHIP_PREC_BF16 && m.edge.n_layers == 2 && m.flow_in.n_layers == 2 && m.classifier.n_layers == 2 &&
           m.classifier.out_dims[1] == 1 && edge_chain_bf16_supported(d.he, d.de, d.hn, d.dn, m.classifier.out_dims[0], d.ef);
}
// TRAINING in that mode on the fused kernels (round 4): the forward chain kernel saves the hidden activations as bf16 rows and
// the ReLU decisions as bits, the backward is one fused chain kernel per step writing bf16 dZ blocks, the weight-gradient and
// scatter-add kernels read bf16 sources.  One predicate for the forward's and the backward's plans (MPNHIP_NO_CHAIN_BF16_TRAIN=1:
// round 3's unfused training path, A-B switch).
static inline bool chain_bf16_train_ok(const mpnhip_model& m, const Dims& d) {
    return chain_bf16_ok(m, d) && d.L >= 1 && d.he % 8 == 0 && d.de % 8 == 0 && d.hn % 8 == 0 && d.dn % 8 == 0 &&
           m.classifier.out_dims[0] % 8 == 0 && edge_chain_bf16_bwd_supported(d.he, d.de, d.hn, d.dn, m.classifier.out_dims[0]) &&
           !getenv("MPNHIP_NO_CHAIN_BF16_TRAIN");
}

// ------------------------------------------------------------------------------------ workspace
struct Arena {
    char* base;
    size_t off;
    float* f(size_t n) {
        size_t o = off;
        off = align_up(off + n * sizeof(float), 256);
        return base ? reinterpret_cast<float*>(base + o) : nullptr;
    }
    int* i(size_t n) { return reinterpret_cast<int*>(f(n)); }
};

// per-step activation buffers of one MetaLayer + classifier evaluation
struct StepBufs {
    float* P;                          // [N, pw]   per-node projections (+ folded biases)
    float* HE[MPNHIP_MAX_LAYERS];      // hidden activations of the edge MLP   [E, edge.out_dims[i]]
    float* HC[MPNHIP_MAX_LAYERS];      // hidden activations of the classifier [E, cls.out_dims[i]]
    float* HF[MPNHIP_MAX_LAYERS];      // hidden activations of the flow MLPs  [E, flow.out_dims[i]]
    float* M;                          // [E, dn]   messages (post-ReLU), sorted order
    float* AGG;                        // [N, 2dn]  [flow_in | flow_out]
    int* ARG;                          // [N, 2dn]  argmax (max aggregation, training only)
    int* MK;                           // ReLU masks of the fused chain kernels as bits (edge_chain.h: chain_mask_ints / chain_bf16_mask_ints)
    // (FwdPlan::b16 -- bf16-operand training on the fused kernels: HE[0] / HC[0] / HF[0] hold bf16 [E, width] rows, not floats)
};

static inline int pad32(int v) { return (v + 31) / 32 * 32; }

// pre-transposed, zero-padded weight images of the fused edge-chain kernel (edge_chain.hip); P(.) = pad32
struct ChainWeights {
    float* w1T;       // [ke][P(he)]
    float* w2T;       // [P(he)][P(de)]
    float* wc1T;      // [P(de)][32]
    float* wf1T[2];   // [P(de)][P(hn)]  (0: flow_out, 1: flow_in)
    float* wf2T[2];   // [P(hn)][P(dn)]
    bool ok;          // the model's shapes are covered by the fused kernel
    bool split;       // the images are three-piece bf16 split images (edge_chain.hip), 3/2 the size, same logical layout
};

// pair images of the bf16-operand chain kernel (edge_chain_bf16.hip)
struct ChainBf16 {
    char* img;        // [edge | classifier | flow_out | flow_in]
    size_t off_cls, off_flow[2];
    bool ok;          // the model's shapes are covered and mpnhip_model.precision == MPNHIP_PREC_BF16
    float* piece;     // scratch of the kernel's fused aggregation (depends on E: carved at the END of the plan)
    int* start_row;
};

struct FwdPlan {
    float* Wnode;  // [pw, kx]
    float* bnode;  // [pw]
    ChainWeights cw;
    ChainBf16 cb;
    unsigned short* nc_img;   // unit images of the fused node-side kernel (node_chain.hip) or nullptr
    // MPNHIP_PREC_BF16, round 5 (gemm_bf16.hip): bf16 images of the packed projection weights [pw][kx] and of the node update's
    // weight [dn][2 dn] (weight images: at the head of the plan), and bf16 mirrors of the node features, one per history slot
    // [hist_slots][N, dn] -- the A operand of the projections is bf16 rows in memory, [x0 | x] as two K segments
    unsigned short* Wnode16;
    unsigned short* Wu16;
    unsigned short* xb_hist;
    float* P_alt;  // [N, pw] (inference, dn = 32) or nullptr
    int* barrier;  // 4 words: grid barrier of the one-launch step loop
    float* P0;     // [N, pw] step-invariant half of the per-node projections: x0 Wnode[:, :dn]^T + bnode
    float* Q0;     // [E, he] step-invariant share of the edge MLP's first layer: e0 W1[:, e0 columns]^T (fused chain only)
    float* enc_n[2];
    float* enc_e[2];
    float* x_hist;  // [(L+1) or 3][N, dn]   x_hist[0] = encoder output
    float* e_hist;  // [(L+1) or 3][E, de]   sorted order
    int hist_slots;
    StepBufs step0;
    size_t step_stride_bytes;  // 0 when the step buffers are reused (inference)
    float* splitk;             // scratch of the split-K form of the node encoder's layers (few rows, long K), or nullptr
    size_t splitk_floats;
    bool b16;                  // training in the bf16-operand mode on the fused kernels (chain_bf16_train_ok): bf16 saves
    unsigned short* eb_hist;   // [hist_slots][E, de] bf16 copies of e_hist: the chain kernel's first-layer input (b16 training and
                               // bf16 inference) and, in training, operands of the weight-gradient products; or nullptr
    size_t total;
};

static inline void carve_step(Arena& a, const mpnhip_model& m, const Dims& d, int64_t N, int64_t E, bool with_cls, bool with_arg,
                       StepBufs* sb, bool b16 = false) {
    StepBufs s = {};
    s.P = a.f((size_t)N * d.pw);
    // (b16: the hidden activations are kept as bf16 rows -- half the floats)
    auto hid = [&](int64_t rows, int width) { return a.f(b16 ? ((size_t)rows * width + 1) / 2 : (size_t)rows * width); };
    for (int i = 0; i + 1 < m.edge.n_layers; ++i) s.HE[i] = hid(E, m.edge.out_dims[i]);
    if (with_cls)
        for (int i = 0; i + 1 < m.classifier.n_layers; ++i) s.HC[i] = hid(E, m.classifier.out_dims[i]);
    for (int i = 0; i + 1 < m.flow_in.n_layers; ++i) s.HF[i] = hid(E, m.flow_in.out_dims[i]);
    s.M = a.f((size_t)E * d.dn);
    s.AGG = a.f((size_t)N * 2 * d.dn);
    s.ARG = with_arg ? a.i((size_t)N * 2 * d.dn) : nullptr;
    s.MK = a.i(b16 ? chain_bf16_mask_ints(E, d.he, d.de, d.hn, d.dn, m.classifier.n_layers >= 1 ? m.classifier.out_dims[0] : 1)
                   : chain_mask_ints(E, d.he, d.de, d.hn, d.dn));
    if (sb) *sb = s;
}

static inline StepBufs step_at(const FwdPlan& p, int s) {
    StepBufs b = p.step0;
    size_t sh = p.step_stride_bytes * (size_t)s;
    auto mv = [&](float*& q) { if (q) q = reinterpret_cast<float*>(reinterpret_cast<char*>(q) + sh); };
    mv(b.P);
    for (int i = 0; i < MPNHIP_MAX_LAYERS; ++i) { mv(b.HE[i]); mv(b.HC[i]); mv(b.HF[i]); }
    mv(b.M);
    mv(b.AGG);
    if (b.ARG) b.ARG = reinterpret_cast<int*>(reinterpret_cast<char*>(b.ARG) + sh);
    if (b.MK) b.MK = reinterpret_cast<int*>(reinterpret_cast<char*>(b.MK) + sh);
    return b;
}

// ONE definition, for the forward (mpn.hip) and the backward (backward.hip), of "the node side of this call runs over bf16 rows": the
// plan carved the images (dims + environment: `rows16` in plan_forward) AND the run-time alignments launch_gemm's bf16-row path insists
// on hold (16-byte weight / bias bases, whole 16-byte result vectors).  The forward fills xb_hist only under this test; the backward
// reads it only under the same one.
static inline bool node_rows16_runtime(const FwdPlan& p, const mpnhip_model& m, const Dims& d) {
    auto al = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    return p.Wnode16 && p.xb_hist && ((size_t)d.pw * d.kx) % 4 == 0 && d.pw % 4 == 0 && d.dn % 4 == 0 &&
           al(m.node.weight[0]) && (!m.node.bias[0] || al(m.node.bias[0]));
}

static inline size_t plan_forward(const mpnhip_model& m, const Dims& d, int64_t N, int64_t E, int save, void* base, FwdPlan* out) {
    Arena a = {static_cast<char*>(base), 0};
    FwdPlan p = {};
    // weight images first: their offsets depend on the model's dims only, never on N / E, so a caller that keeps the
    // workspace can keep them across calls (mpnhip_model.weights_prepacked)
    p.Wnode = a.f((size_t)d.pw * d.kx);
    p.bnode = a.f((size_t)d.pw);
    {
        const size_t HE = pad32(d.he), DE = pad32(d.de), HN = pad32(d.hn), DN = pad32(d.dn);
        // (sized for the split images, 3/2 of the fp32 ones: ChainWeights.split)
        const size_t KE = (size_t)(d.ke + 15) / 16 * 16;
        p.cw.w1T = a.f(KE * HE * 3 / 2);
        p.cw.w2T = a.f(HE * DE * 3 / 2);
        p.cw.wc1T = a.f(DE * 32 * 3 / 2);
        for (int q = 0; q < 2; ++q) {
            p.cw.wf1T[q] = a.f(DE * HN * 3 / 2);
            p.cw.wf2T[q] = a.f(HN * DN * 3 / 2);
        }
        p.cw.ok = false;
    }
    bool cb_shapes = false;   // (not p.cb.img: the sizing pass runs without a base pointer)
    {
        const int hc = m.classifier.n_layers >= 1 ? m.classifier.out_dims[0] : 0;
        p.cb = {};
        cb_shapes = edge_chain_bf16_supported(d.he, d.de, d.hn, d.dn, hc, d.ef);
        if (cb_shapes) {
            const size_t bytes = chain_bf16_image_bytes(d.he, d.de, d.hn, d.dn, hc, d.ef, &p.cb.off_cls, &p.cb.off_flow[0], &p.cb.off_flow[1]);
            p.cb.img = reinterpret_cast<char*>(a.f(bytes / 4));
        }
    }
    p.nc_img = nullptr;
    if (node_chain_supported(d.dn, d.pw, d.kx) && m.node.n_layers == 1)
        p.nc_img = reinterpret_cast<unsigned short*>(a.f((node_chain_image_shorts(d.dn, d.pw, nullptr) + 1) / 2));
    // (bf16 rows for the tiled / ring GEMM kernels: 16-byte pieces of 8 elements)
    // (d.pw % 4: launch_gemm's bf16-row path stores 16-byte result vectors -- N = pw of the projections, N = dn of the node update)
    const bool rows16 = m.precision == MPNHIP_PREC_BF16 && d.kx % 8 == 0 && d.dn % 8 == 0 && d.pw % 4 == 0 && m.node.n_layers == 1 && !getenv("MPNHIP_NO_GEMM_BF16_ROWS");
    p.Wnode16 = rows16 ? reinterpret_cast<unsigned short*>(a.f(((size_t)d.pw * d.kx + 1) / 2)) : nullptr;
    p.Wu16 = rows16 ? reinterpret_cast<unsigned short*>(a.f(((size_t)d.dn * 2 * d.dn + 1) / 2)) : nullptr;
    p.P0 = a.f((size_t)N * d.pw);
    p.P_alt = (!save && d.dn == 32) ? a.f((size_t)N * d.pw) : nullptr;   // second projection buffer of the one-launch step loop (persist32.hip)
    p.barrier = a.i(4);
    p.Q0 = a.f((size_t)E * d.he);
    int hn_ = max_hidden(m.enc_node), he_ = max_hidden(m.enc_edge);
    if (save) {
        // keep every encoder activation for the backward pass: one buffer per hidden layer
        p.enc_n[0] = a.f((size_t)N * (sum_hidden(m.enc_node) > 0 ? sum_hidden(m.enc_node) : 1));
        p.enc_e[0] = a.f((size_t)E * (sum_hidden(m.enc_edge) > 0 ? sum_hidden(m.enc_edge) : 1));
        p.enc_n[1] = p.enc_e[1] = nullptr;
    } else {
        for (int i = 0; i < 2; ++i) {
            p.enc_n[i] = a.f((size_t)N * (hn_ > 0 ? hn_ : 1));
            p.enc_e[i] = a.f((size_t)E * (he_ > 0 ? he_ : 1));
        }
    }
    p.hist_slots = save ? d.L + 1 : 3;
    p.x_hist = a.f((size_t)p.hist_slots * N * d.dn);
    p.e_hist = a.f((size_t)p.hist_slots * E * d.de);
    p.xb_hist = rows16 ? reinterpret_cast<unsigned short*>(a.f(((size_t)p.hist_slots * N * d.dn + 1) / 2)) : nullptr;
    p.b16 = save && cb_shapes && chain_bf16_train_ok(m, d);
    // (inference in the bf16-operand mode keeps the same mirror over its three history slots: the chain kernel reads its first-layer
    // input as bf16 rows -- the values it rounds to anyway -- and writes the new features as bf16 for the next step)
    const bool eb_inf = !save && cb_shapes && chain_bf16_ok(m, d) && d.L >= 1 && d.de % 8 == 0 && !getenv("MPNHIP_NO_CHAIN_BF16_E16");
    p.eb_hist = (p.b16 || eb_inf) ? reinterpret_cast<unsigned short*>(a.f(((size_t)p.hist_slots * E * d.de + 1) / 2)) : nullptr;
    size_t before = a.off;
    carve_step(a, m, d, N, E, true, save && m.agg == MPNHIP_AGG_MAX, &p.step0, p.b16);
    p.step_stride_bytes = 0;
    if (save && d.L > 1) {
        p.step_stride_bytes = a.off - before;
        a.off = before + p.step_stride_bytes * (size_t)d.L;
    }
    {
        size_t sk = 0;
        for (int i = 0; i < m.enc_node.n_layers; ++i) {
            const size_t f = linear_splitk_scratch_floats(N, m.enc_node.out_dims[i], i == 0 ? m.enc_node.in_dim : m.enc_node.out_dims[i - 1]);
            sk = f > sk ? f : sk;
        }
        p.splitk_floats = sk;
        p.splitk = sk ? a.f(sk) : nullptr;
    }
    if (cb_shapes && (!save || p.b16)) {
        size_t off = 0;
        const size_t fl = chain_bf16_agg_scratch_floats(E, d.dn, &off);
        p.cb.piece = a.f(fl);
        p.cb.start_row = p.cb.piece ? reinterpret_cast<int*>(p.cb.piece + off) : nullptr;
    }
    p.total = a.off;
    if (out) *out = p;
    return p.total;
}

static inline void hidden_ptrs(const mpnhip_mlp& m, float* const two[2], int64_t rows, bool per_layer, float** out) {
    size_t off = 0;
    for (int i = 0; i + 1 < m.n_layers; ++i) {
        if (per_layer) {
            out[i] = two[0] + off;
            off += (size_t)rows * m.out_dims[i];
        } else {
            out[i] = two[i & 1];
        }
    }
}

// the fused edge-chain kernels (edge_chain.hip) cover this model's per-edge modules
static inline bool chain_shapes_ok(const mpnhip_model& m, const Dims& d) {
    if (getenv("MPNHIP_NO_CHAIN")) return false;  // tuning / A-B switch
    if (m.precision == MPNHIP_PREC_BF16) return false;  // the fused chain kernels compute fp32 results (FP32 / FP32_SPLIT)
    return m.edge.n_layers == 2 && m.flow_in.n_layers == 2 && m.classifier.n_layers == 2 && m.classifier.out_dims[1] == 1 &&
           edge_chain_supported(d.he, d.de, d.hn, d.dn, m.classifier.out_dims[0], d.de, d.ef == 2 ? d.de : 0);
}

// split (three-piece bf16) weight images and six-product MFMAs in the fused chain kernels: mpnhip_model.precision ==
// MPNHIP_PREC_FP32_SPLIT (MPNHIP_CHAIN_SPLIT=0/1 in the environment overrides it: A-B switch for measurements)
static inline bool chain_split(const mpnhip_model& m) {
    if (const char* e = getenv("MPNHIP_CHAIN_SPLIT")) return e[0] == '1';
    return m.precision == MPNHIP_PREC_FP32_SPLIT;
}

static inline void init_group(GemmGroup& g) { memset(&g, 0, sizeof(g)); }

}  // namespace mpnhip
