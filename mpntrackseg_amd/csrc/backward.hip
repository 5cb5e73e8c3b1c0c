// Hand-written backward pass of the MPN hot path (what torch.autograd derives for the reference's
// models/mpn.py:349-392), mirroring the forward's project-then-gather structure:
//
//   activation gradients   dH_{i-1} = (dZ_i W_i) (.) [H_{i-1} > 0]       -> gemm_kernel (B N-contiguous,
//                                                                          ReLU mask fused in the epilogue)
//   weight / bias gradients dW_i += dZ_i^T H_{i-1}, db_i += colsum(dZ_i)   -> gemm_tn_kernel (split over edges)
//   index_put_(accumulate) of the reference's x[row] / x[col] gathers      -> segmented sums over the
//       row- and col-sorted CSR lists built by graph prep (no atomics, fixed order)
//   scatter_{add,mean,max} backward                                        -> one gather kernel (k_agg_bwd)
//
// Every per-step activation was saved by the forward (288 GB of HBM: nothing is recomputed).
#include "common.h"
#include "plan.h"

namespace mpnhip {

// ------------------------------------------------------------------------------------ small kernels
// out = g (.) [act > 0]   (ReLU backward), float4
__global__ void k_relu_mask(const float* __restrict__ g, const float* __restrict__ act, float* __restrict__ out, int64_t n) {
    int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i + 3 < n) {
        float4 a = *reinterpret_cast<const float4*>(act + i), v = *reinterpret_cast<const float4*>(g + i);
        v.x = a.x > 0.f ? v.x : 0.f; v.y = a.y > 0.f ? v.y : 0.f; v.z = a.z > 0.f ? v.z : 0.f; v.w = a.w > 0.f ? v.w : 0.f;
        *reinterpret_cast<float4*>(out + i) = v;
    } else {
        for (; i < n; ++i) out[i] = act[i] > 0.f ? g[i] : 0.f;
    }
}

// Backward of node_agg_fn followed by the ReLU of the last flow layer:
//   dZM[j][c] = [M[j][c] > 0] * dAGG[srow[j]][half(j) + c] (* 1/cnt for mean) (* [ARG == j] for max)
// j in sorted order; half = dn for flow_out rows (j < E_out), 0 for flow_in rows; self-loop rows -> 0.
__global__ void k_agg_bwd(const float* __restrict__ dagg, const float* __restrict__ msg, const int* __restrict__ arg,
                          const int* __restrict__ srow, const int* __restrict__ seg_ptr, const int* __restrict__ header,
                          int N, int64_t E, int dn, int agg, int has_relu, float* __restrict__ out) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t j = t / dn;
    int c = (int)(t % dn);
    if (j >= E) return;
    const int e_out = header[1], e_in = header[2];
    float v = 0.f;
    if (j < e_out + e_in) {
        const int dir = j < e_out ? 0 : 1;
        const int row = srow[j];
        const int64_t o = (int64_t)row * 2 * dn + (dir == 0 ? dn : 0) + c;
        v = dagg[o];
        if (agg == MPNHIP_AGG_MEAN) {
            const int key = dir * N + row;
            const int cnt = seg_ptr[key + 1] - seg_ptr[key];
            v = v / (float)(cnt > 0 ? cnt : 1);
        } else if (agg == MPNHIP_AGG_MAX) {
            v = arg[o] == (int)j ? v : 0.f;
        }
        if (has_relu) v = msg[j * dn + c] > 0.f ? v : 0.f;
    }
    out[j * dn + c] = v;
}

// dst[r][c0 + c] += src[r][c]
__global__ void k_add_block(const float* __restrict__ src, int64_t lds, float* __restrict__ dst, int64_t ldd, int c0,
                            int rows, int cols) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)rows * cols) return;
    int r = (int)(i / cols), c = (int)(i % cols);
    dst[(int64_t)r * ldd + c0 + c] += src[(int64_t)r * lds + c];
}

// src [rows, 2d] (or [rows, d] when !two): acc[r][c] += src[r][c]; other[r][c] (=|+=) src[r][d + c]
__global__ void k_split_cat(const float* __restrict__ src, int64_t rows, int d, int two, float* __restrict__ acc,
                            float* __restrict__ other, int other_accumulate) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * d) return;
    int64_t r = i / d;
    int c = (int)(i % d);
    if (two) {
        acc[i] += src[r * 2 * d + c];
        float v = src[r * 2 * d + d + c];
        other[i] = other_accumulate ? other[i] + v : v;
    } else {
        float v = src[i];
        other[i] = other_accumulate ? other[i] + v : v;
    }
}

// dst[i][:] = src[idx[i]][:]
__global__ void k_gather_rows(const float* __restrict__ src, const int* __restrict__ idx, float* __restrict__ dst,
                              int64_t rows, int cols) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * cols) return;
    int64_t r = i / cols;
    int c = (int)(i % cols);
    dst[i] = src[(int64_t)idx[r] * cols + c];
}

static int relu_mask(const float* g, const float* act, float* out, int64_t n, hipStream_t s) {
    if (n <= 0) return MPNHIP_OK;
    hipLaunchKernelGGL(k_relu_mask, dim3((unsigned)((n / 4 + 256) / 256)), dim3(256), 0, s, g, act, out, n);
    MPN_LAUNCH_CHECK();
    return MPNHIP_OK;
}

// ------------------------------------------------------------------------------------ plan
struct BwdPlan {
    float* dX[2];
    float* dX0;
    float* dE[2];
    float* dE0;
    float* dP;
    float* dAGG;
    float* dZn;
    float* dCat;     // [max(E ke, N kx)] gradient w.r.t. the concatenated [initial | current] features
    float* T[2];     // [max(E,N), maxw] scratch for the dZ chain of the MLPs
    float* gWnode;   // [pw, kx] gradient of the packed node-projection weights, accumulated over steps
    float* slab;     // split partials of the weight-gradient products (2 groups)
    size_t slab_floats_per_group;
    size_t total;
};

static int maxw_of(const mpnhip_model& m, const Dims& d) {
    int w = d.dn > d.de ? d.dn : d.de;
    const mpnhip_mlp* all[] = {&m.enc_node, &m.enc_edge, &m.edge, &m.flow_in, &m.classifier};
    for (const mpnhip_mlp* p : all)
        for (int i = 0; i < p->n_layers; ++i) w = p->out_dims[i] > w ? p->out_dims[i] : w;
    return w;
}

static size_t mlp_slab(const mpnhip_mlp& m, int64_t rows) {
    size_t mx = 0;
    for (int i = 0; i < m.n_layers; ++i) {
        size_t f = tn_slab_floats(m.out_dims[i], i == 0 ? m.in_dim : m.out_dims[i - 1], rows);
        mx = f > mx ? f : mx;
    }
    return mx;
}

static size_t plan_backward(const mpnhip_model& m, const Dims& d, int64_t N, int64_t E, void* base, BwdPlan* out) {
    Arena a = {static_cast<char*>(base), 0};
    BwdPlan p = {};
    for (int i = 0; i < 2; ++i) p.dX[i] = a.f((size_t)N * d.dn);
    p.dX0 = a.f((size_t)N * d.dn);
    for (int i = 0; i < 2; ++i) p.dE[i] = a.f((size_t)E * d.de);
    p.dE0 = a.f((size_t)E * d.de);
    p.dP = a.f((size_t)N * d.pw);
    p.dAGG = a.f((size_t)N * 2 * d.dn);
    p.dZn = a.f((size_t)N * d.dn);
    {
        size_t a1 = (size_t)E * d.ke, a2 = (size_t)N * d.kx;
        p.dCat = a.f(a1 > a2 ? a1 : a2);
    }
    int64_t rows = E > N ? E : N;
    int mw = maxw_of(m, d);
    for (int i = 0; i < 2; ++i) p.T[i] = a.f((size_t)rows * mw);
    p.gWnode = a.f((size_t)d.pw * d.kx);
    size_t sl = 0;
    auto upd = [&](size_t f) { sl = f > sl ? f : sl; };
    upd(mlp_slab(m.enc_node, N));
    upd(mlp_slab(m.enc_edge, E));
    upd(mlp_slab(m.edge, E));
    upd(mlp_slab(m.flow_in, E));
    upd(mlp_slab(m.classifier, E));
    upd(tn_slab_floats(d.dn, 2 * d.dn, N));
    upd(tn_slab_floats(d.pw, d.kx, N));
    p.slab_floats_per_group = sl;
    p.slab = a.f(2 * sl);
    p.total = a.off;
    if (out) *out = p;
    return p.total;
}

// ------------------------------------------------------------------------------------ helpers
struct RowRange {
    const int* begin;
    const int* end;
};

// dW += dZ^T H (+ bias) for one or two groups
static int weight_grad(const BwdPlan& p, int ngroups, const float* const dZ[2], int64_t ldz, const int* dz_idx,
                       const float* H, int64_t ldh, const float* H2, int64_t ldh2, int csplit, const int* h_idx,
                       int n_out, int k_in, float* const gw[2], int64_t ldw, float* const gb[2], const RowRange rr[2],
                       int64_t rows, hipStream_t s) {
    TnArgs a = {};
    a.ngroups = ngroups;
    a.n_out = n_out;
    a.k_in = k_in;
    a.csplit = H2 ? csplit : k_in;
    a.m_upper = rows;
    for (int q = 0; q < ngroups; ++q) {
        TnGroup& g = a.g[q];
        g.dZ = dZ[q];
        g.ldz = ldz;
        g.dz_idx = dz_idx;
        g.H = H;
        g.ldh = ldh;
        g.H2 = H2;
        g.ldh2 = ldh2;
        g.h_idx = h_idx;
        g.row_begin = rr ? rr[q].begin : nullptr;
        g.row_end = rr ? rr[q].end : nullptr;
        g.m_static = rows;
        g.slab = p.slab + q * p.slab_floats_per_group;
        g.grad_w = gw[q];
        g.ldw = ldw;
        g.grad_b = gb ? gb[q] : nullptr;
    }
    return launch_gemm_tn(a, s);
}

// C = mask( A B (+ C) ) with B given as weight rows: B[k][n] = W[k * ldw + n]  (dH = dZ W)
static int act_grad(int ngroups, const float* const A[2], int64_t lda, const int* a_idx, const float* const W[2],
                    int64_t ldw, int K, int N, float* C, int64_t ldc, const int* c_idx, const float* mask, int64_t ldmask,
                    int accumulate, const RowRange rr[2], int64_t rows, hipStream_t s) {
    GemmArgs a = {};
    a.ngroups = ngroups;
    a.N = N;
    a.K = K;
    a.ksplit = K;
    a.relu = 0;
    a.accumulate = accumulate;
    a.m_upper = rows;
    for (int q = 0; q < ngroups; ++q) {
        GemmGroup& g = a.g[q];
        init_group(g);
        g.A = A[q];
        g.lda = lda;
        g.a_idx = a_idx;
        g.B = W[q];
        g.ldb = ldw;
        g.C = C;
        g.ldc = ldc;
        g.c_idx = c_idx;
        g.mask = mask;
        g.ldmask = ldmask;
        g.m_static = rows;
        g.row_begin = rr ? rr[q].begin : nullptr;
        g.row_end = rr ? rr[q].end : nullptr;
    }
    return launch_gemm(a, A_KCONTIG, B_NCONTIG, s);
}

// Backward through layers n-1 .. 1 of an MLP (optionally the two direction-specific flow MLPs).
// On entry *dz holds dZ of the LAST layer (pre-activation gradient) in buffer `cur_buf`; on exit it
// holds dZ of layer 0 [rows, out_dims[0]].  hidden[i] = saved post-activation output of layer i.
static int mlp_tail_backward(const BwdPlan& p, const mpnhip_mlp& m0, const mpnhip_mlp* m1, float* const* hidden,
                             const float** dz, int* cur_buf, const RowRange* rr, int64_t rows, hipStream_t s) {
    const int ng = m1 ? 2 : 1;
    for (int i = m0.n_layers - 1; i >= 1; --i) {
        const int n_out = m0.out_dims[i], k_in = m0.out_dims[i - 1];
        const float* dzq[2] = {*dz, *dz};
        float* gw[2] = {m0.grad_weight[i], m1 ? m1->grad_weight[i] : nullptr};
        float* gb[2] = {m0.grad_bias[i], m1 ? m1->grad_bias[i] : nullptr};
        MPN_TRY(weight_grad(p, ng, dzq, n_out, nullptr, hidden[i - 1], k_in, nullptr, 0, k_in, nullptr, n_out, k_in, gw,
                            k_in, gb, rr, rows, s));
        const float* Wq[2] = {m0.weight[i], m1 ? m1->weight[i] : nullptr};
        float* dst = p.T[*cur_buf ^ 1];
        const bool relu_prev = k_in != 1;
        MPN_TRY(act_grad(ng, dzq, n_out, nullptr, Wq, k_in, n_out, k_in, dst, k_in, nullptr,
                         relu_prev ? hidden[i - 1] : nullptr, k_in, 0, rr, rows, s));
        *cur_buf ^= 1;
        *dz = dst;
    }
    return MPNHIP_OK;
}

}  // namespace mpnhip

using namespace mpnhip;

extern "C" size_t mpnhip_backward_workspace_bytes(const mpnhip_model* model, int n_nodes, int64_t n_edges) {
    Dims d;
    if (!model) return 256;  // non-zero: "the backward pass is built into this library"
    if (check_full(*model, &d, false) != MPNHIP_OK) return 0;
    return plan_backward(*model, d, n_nodes, n_edges, nullptr, nullptr);
}

extern "C" int mpnhip_backward(const mpnhip_model* model, const void* graph_buf, int n_nodes, int64_t n_edges,
                               const float* x, const float* edge_attr, const float* grad_logits, const float* grad_x_out,
                               const float* grad_e_out, float* grad_x, float* grad_edge_attr, void* fwd_workspace,
                               size_t fwd_workspace_bytes, void* bwd_workspace, size_t bwd_workspace_bytes, void* stream_) {
    hipStream_t s = static_cast<hipStream_t>(stream_);
    MPN_CHECK_ARG(model && graph_buf, "backward: null model / graph");
    const mpnhip_model& m = *model;
    Dims d;
    MPN_TRY(check_full(m, &d));
    const int64_t N = n_nodes, E = n_edges;
    MPN_CHECK_ARG(N >= 0 && E >= 0, "backward: negative sizes");
    MPN_CHECK_ARG((x || N == 0) && (edge_attr || E == 0) && (grad_logits || E == 0), "backward: null tensor");
    {
        const mpnhip_mlp* all[] = {&m.enc_node, &m.enc_edge, &m.edge, &m.flow_in, &m.flow_out, &m.node, &m.classifier};
        for (const mpnhip_mlp* q : all)
            for (int i = 0; i < q->n_layers; ++i)
                MPN_CHECK_ARG(q->grad_weight[i] && q->grad_bias[i], "backward: null gradient buffer");
    }
    FwdPlan f;
    size_t fneed = plan_forward(m, d, N, E, 1, fwd_workspace, &f);
    if (!fwd_workspace || fwd_workspace_bytes < fneed) {
        set_error("backward: forward workspace %zu < %zu (must be the save_for_backward buffer)", fwd_workspace_bytes, fneed);
        return MPNHIP_ERR_WORKSPACE;
    }
    BwdPlan p;
    size_t need = plan_backward(m, d, N, E, bwd_workspace, &p);
    if (!bwd_workspace || bwd_workspace_bytes < need) {
        set_error("backward: workspace %zu < %zu", bwd_workspace_bytes, need);
        return MPNHIP_ERR_WORKSPACE;
    }
    GraphView g;
    graph_layout(n_nodes, n_edges, &g, const_cast<void*>(graph_buf));
    const int he = d.he, hn = d.hn, dn = d.dn, de = d.de, kx = d.kx, ke = d.ke, pw = d.pw;
    const size_t xs = (size_t)N * dn, es = (size_t)E * de;
    const RowRange dir_rr[2] = {{nullptr, g.header + 4}, {g.header + 4, g.header + 5}};

    // ---- seeds ----------------------------------------------------------------------------------
    int cx = 0, ce = 0;  // which of dX[2] / dE[2] holds the gradient w.r.t. (x_s, e_s)
    if (grad_x_out) MPN_HIP(hipMemcpyAsync(p.dX[0], grad_x_out, xs * 4, hipMemcpyDeviceToDevice, s));
    else if (xs) MPN_HIP(hipMemsetAsync(p.dX[0], 0, xs * 4, s));
    if (grad_e_out && es) {
        hipLaunchKernelGGL(k_gather_rows, dim3((unsigned)((es + 255) / 256)), dim3(256), 0, s, grad_e_out, g.perm, p.dE[0], E, de);
        MPN_LAUNCH_CHECK();
    } else if (es) {
        MPN_HIP(hipMemsetAsync(p.dE[0], 0, es * 4, s));
    }
    if (xs) MPN_HIP(hipMemsetAsync(p.dX0, 0, xs * 4, s));
    if (es) MPN_HIP(hipMemsetAsync(p.dE0, 0, es * 4, s));
    MPN_HIP(hipMemsetAsync(p.gWnode, 0, (size_t)pw * kx * 4, s));

    const float* x0 = f.x_hist;
    const float* e0 = f.e_hist;
    const mpnhip_mlp& cls = m.classifier;

    // classifier backward on edge features `ef` (sorted order) for step-row `lrow` of grad_logits;
    // accumulates into dEdst and finally applies `mask` (ReLU of the producing edge layer)
    auto classifier_backward = [&](const float* ef, float* const* HC, const float* dlog, float* dEdst, const float* mask) -> int {
        if (E == 0) return MPNHIP_OK;
        const float* dz = dlog;  // [E, 1] in ORIGINAL order -> gathered through perm
        int64_t ldz = 1;
        const int* zidx = g.perm;
        int cur = 0;
        for (int i = cls.n_layers - 1; i >= 0; --i) {
            const int n_out = cls.out_dims[i], k_in = i == 0 ? de : cls.out_dims[i - 1];
            const float* Hin = i == 0 ? ef : HC[i - 1];
            const float* dzq[2] = {dz, dz};
            float* gw[2] = {cls.grad_weight[i], nullptr};
            float* gb[2] = {cls.grad_bias[i], nullptr};
            MPN_TRY(weight_grad(p, 1, dzq, ldz, zidx, Hin, k_in, nullptr, 0, k_in, nullptr, n_out, k_in, gw, k_in, gb,
                                nullptr, E, s));
            const float* Wq[2] = {cls.weight[i], nullptr};
            if (i == 0) {
                MPN_TRY(act_grad(1, dzq, ldz, zidx, Wq, k_in, n_out, k_in, dEdst, de, nullptr, mask, de, 1, nullptr, E, s));
            } else {
                float* dst = p.T[cur];
                const bool relu_prev = k_in != 1;
                MPN_TRY(act_grad(1, dzq, ldz, zidx, Wq, k_in, n_out, k_in, dst, k_in, nullptr,
                                 relu_prev ? HC[i - 1] : nullptr, k_in, 0, nullptr, E, s));
                dz = dst;
                ldz = k_in;
                zidx = nullptr;
                cur ^= 1;
            }
        }
        return MPNHIP_OK;
    };

    for (int step = d.L; step >= 1; --step) {
        const StepBufs b = step_at(f, step - 1);
        const float* x_s = f.x_hist + xs * step;
        const float* e_s = f.e_hist + es * step;
        const float* x_p = f.x_hist + xs * (step - 1);
        const float* e_p = f.e_hist + es * (step - 1);
        float* dXc = p.dX[cx];
        float* dEc = p.dE[ce];

        // ---- A. node update  x_s = relu(AGG W^T + b)  (mpn.py:97-99) ---------------------------
        MPN_TRY(relu_mask(dXc, x_s, p.dZn, (int64_t)xs, s));
        {
            const float* dzq[2] = {p.dZn, nullptr};
            float* gw[2] = {m.node.grad_weight[0], nullptr};
            float* gb[2] = {m.node.grad_bias[0], nullptr};
            MPN_TRY(weight_grad(p, 1, dzq, dn, nullptr, b.AGG, 2 * dn, nullptr, 0, 2 * dn, nullptr, dn, 2 * dn, gw, 2 * dn,
                                gb, nullptr, N, s));
            const float* Wq[2] = {m.node.weight[0], nullptr};
            MPN_TRY(act_grad(1, dzq, dn, nullptr, Wq, 2 * dn, dn, 2 * dn, p.dAGG, 2 * dn, nullptr, nullptr, 0, 0, nullptr, N, s));
        }
        if (E > 0) {
            // ---- B. aggregation backward + ReLU of the last flow layer ---------------------------
            float* dZM = p.T[0];
            {
                int64_t tot = E * dn;
                hipLaunchKernelGGL(k_agg_bwd, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, p.dAGG, b.M, b.ARG,
                                   g.srow, g.seg_ptr, g.header, (int)N, E, dn, m.agg, dn != 1 ? 1 : 0, dZM);
                MPN_LAUNCH_CHECK();
            }
            // ---- C. flow MLPs (both directions grouped) -------------------------------------------
            const float* dz = dZM;
            int cur = 0;
            MPN_TRY(mlp_tail_backward(p, m.flow_out, &m.flow_in, b.HF, &dz, &cur, dir_rr, E, s));
            {
                // layer 0:  Z = e_s Wfe^T + Pf[col]
                const float* dzq[2] = {dz, dz};
                float* gw[2] = {m.flow_out.grad_weight[0] + kx, m.flow_in.grad_weight[0] + kx};
                float* gb[2] = {m.flow_out.grad_bias[0], m.flow_in.grad_bias[0]};
                MPN_TRY(weight_grad(p, 2, dzq, hn, nullptr, e_s, de, nullptr, 0, de, nullptr, hn, de, gw, m.flow_out.in_dim,
                                    gb, dir_rr, E, s));
                // dPf[n] = sum over the direction's edges with col == n   (index_put_ of x[flow_col], mpn.py:87,93)
                MPN_TRY(segment_reduce_csr2(dz, hn, g.cperm, g.cseg_ptr, 2 * (int)N, hn, p.dP, pw, (int)N, 2 * he, 2 * he + hn, s));
                const float* Wq[2] = {m.flow_out.weight[0] + kx, m.flow_in.weight[0] + kx};
                MPN_TRY(act_grad(2, dzq, hn, nullptr, Wq, m.flow_out.in_dim, hn, de, dEc, de, nullptr, nullptr, 0, 1, dir_rr, E, s));
            }
            // ---- D. classifier (mpn.py:377 -> :114); also applies the ReLU mask of e_s -------------
            MPN_TRY(classifier_backward(e_s, b.HC, grad_logits + (size_t)(step - 1) * E, dEc, de != 1 ? e_s : nullptr));
            // ---- E. edge MLP (EdgeModel, mpn.py:67-69) ----------------------------------------------
            dz = dEc;
            cur = 0;  // T[0] (dZM) is dead by now; the chain ping-pongs T[1], T[0], ...
            MPN_TRY(mlp_tail_backward(p, m.edge, nullptr, b.HE, &dz, &cur, nullptr, E, s));
            {
                const float* dzq[2] = {dz, nullptr};
                float* gw[2] = {m.edge.grad_weight[0] + 2 * kx, nullptr};
                float* gb[2] = {m.edge.grad_bias[0], nullptr};
                const bool two = d.ef == 2;
                MPN_TRY(weight_grad(p, 1, dzq, he, nullptr, two ? e0 : e_p, de, two ? e_p : nullptr, de, de, nullptr, he, ke,
                                    gw, m.edge.in_dim, gb, nullptr, E, s));
                // dPr / dPc: index_put_(accumulate) of x[row], x[col] (mpn.py:69)
                MPN_TRY(segment_reduce_csr2(dz, he, g.rperm, g.rseg_ptr, (int)N, he, p.dP, pw, (int)N, 0, 0, s));
                MPN_TRY(segment_reduce_csr2(dz, he, g.cperm_all, g.cseg_all, (int)N, he, p.dP, pw, (int)N, he, he, s));
                // gradient w.r.t. [e0 | e_{s-1}]: one product, then split (e_0 IS e0 at step 1)
                float* dEp = p.dE[ce ^ 1];
                const float* Wa[2] = {m.edge.weight[0] + 2 * kx, nullptr};
                MPN_TRY(act_grad(1, dzq, he, nullptr, Wa, m.edge.in_dim, he, ke, p.dCat, ke, nullptr, nullptr, 0, 0, nullptr, E, s));
                hipLaunchKernelGGL(k_split_cat, dim3((unsigned)((es + 255) / 256)), dim3(256), 0, s, p.dCat, E, de, two ? 1 : 0,
                                   p.dE0, step == 1 ? p.dE0 : dEp, step == 1 ? 1 : 0);
                MPN_LAUNCH_CHECK();
            }
        } else {
            MPN_HIP(hipMemsetAsync(p.dP, 0, (size_t)N * pw * 4, s));
        }
        // ---- F. per-node projections  P = [x0 | x_{s-1}] Wnode^T -----------------------------------
        {
            const float* dzq[2] = {p.dP, nullptr};
            float* gw[2] = {p.gWnode, nullptr};
            const bool two = d.nf == 2;
            MPN_TRY(weight_grad(p, 1, dzq, pw, nullptr, two ? x0 : x_p, dn, two ? x_p : nullptr, dn, dn, nullptr, pw, kx, gw,
                                kx, nullptr, nullptr, N, s));
            float* dXp = p.dX[cx ^ 1];
            const float* Wa[2] = {f.Wnode, nullptr};
            MPN_TRY(act_grad(1, dzq, pw, nullptr, Wa, kx, pw, kx, p.dCat, kx, nullptr, nullptr, 0, 0, nullptr, N, s));
            if (xs) {
                hipLaunchKernelGGL(k_split_cat, dim3((unsigned)((xs + 255) / 256)), dim3(256), 0, s, p.dCat, N, dn, two ? 1 : 0,
                                   p.dX0, step == 1 ? p.dX0 : dXp, step == 1 ? 1 : 0);
                MPN_LAUNCH_CHECK();
            }
        }
        cx ^= 1;
        ce ^= 1;
    }

    if (d.L == 0) {
        // mpn.py:387-389: only the classifier sits between the encoder output and the logits
        StepBufs b = f.step0;
        MPN_TRY(classifier_backward(e0, b.HC, grad_logits, p.dE0, nullptr));
        // incoming gradients of the final latents ARE gradients of the encoder outputs
        if (xs) {
            hipLaunchKernelGGL(k_add_block, dim3((unsigned)((xs + 255) / 256)), dim3(256), 0, s, p.dX[0], dn, p.dX0, dn, 0, (int)N, dn);
            MPN_LAUNCH_CHECK();
        }
        if (es) {
            hipLaunchKernelGGL(k_add_block, dim3((unsigned)((es + 255) / 256)), dim3(256), 0, s, p.dE[0], de, p.dE0, de, 0, (int)E, de);
            MPN_LAUNCH_CHECK();
        }
    }

    // ---- unpack the packed node-projection weight gradient into the four first-layer blocks --------
    if (d.L > 0) {
        struct { float* dst; int64_t ld; int c0; int r0; int rows; } parts[4] = {
            {m.edge.grad_weight[0], m.edge.in_dim, 0, 0, he},
            {m.edge.grad_weight[0], m.edge.in_dim, kx, he, he},
            {m.flow_out.grad_weight[0], m.flow_out.in_dim, 0, 2 * he, hn},
            {m.flow_in.grad_weight[0], m.flow_in.in_dim, 0, 2 * he + hn, hn}};
        for (auto& q : parts) {
            int64_t tot = (int64_t)q.rows * kx;
            hipLaunchKernelGGL(k_add_block, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, p.gWnode + (size_t)q.r0 * kx,
                               kx, q.dst, q.ld, q.c0, q.rows, kx);
            MPN_LAUNCH_CHECK();
        }
    }

    // ---- encoder (MLPGraphIndependent, mpn.py:355) -------------------------------------------------
    {
        float* two[2] = {f.enc_n[0], nullptr};
        float* hid[MPNHIP_MAX_LAYERS];
        hidden_ptrs(m.enc_node, two, N, true, hid);
        const mpnhip_mlp& en = m.enc_node;
        const float* dz = p.dX0;
        int cur = 0;
        if (N > 0) {
            if (en.out_dims[en.n_layers - 1] != 1) {
                MPN_TRY(relu_mask(p.dX0, x0, p.T[0], (int64_t)xs, s));
                dz = p.T[0];
            } else {
                cur = 1;  // keep T[0] free: the chain below writes T[cur ^ 1] first
            }
            MPN_TRY(mlp_tail_backward(p, en, nullptr, hid, &dz, &cur, nullptr, N, s));
            const float* dzq[2] = {dz, nullptr};
            float* gw[2] = {en.grad_weight[0], nullptr};
            float* gb[2] = {en.grad_bias[0], nullptr};
            MPN_TRY(weight_grad(p, 1, dzq, en.out_dims[0], nullptr, x, en.in_dim, nullptr, 0, en.in_dim, nullptr, en.out_dims[0],
                                en.in_dim, gw, en.in_dim, gb, nullptr, N, s));
            if (grad_x) {
                const float* Wq[2] = {en.weight[0], nullptr};
                MPN_TRY(act_grad(1, dzq, en.out_dims[0], nullptr, Wq, en.in_dim, en.out_dims[0], en.in_dim, grad_x, en.in_dim,
                                 nullptr, nullptr, 0, 0, nullptr, N, s));
            }
        }
    }
    {
        float* two[2] = {f.enc_e[0], nullptr};
        float* hid[MPNHIP_MAX_LAYERS];
        hidden_ptrs(m.enc_edge, two, E, true, hid);
        const mpnhip_mlp& ee = m.enc_edge;
        const float* dz = p.dE0;
        int cur = 0;
        if (E > 0) {
            if (ee.out_dims[ee.n_layers - 1] != 1) {
                MPN_TRY(relu_mask(p.dE0, e0, p.T[0], (int64_t)es, s));
                dz = p.T[0];
            } else {
                cur = 1;
            }
            MPN_TRY(mlp_tail_backward(p, ee, nullptr, hid, &dz, &cur, nullptr, E, s));
            const float* dzq[2] = {dz, nullptr};
            float* gw[2] = {ee.grad_weight[0], nullptr};
            float* gb[2] = {ee.grad_bias[0], nullptr};
            // layer 0 read edge_attr through the sort permutation
            MPN_TRY(weight_grad(p, 1, dzq, ee.out_dims[0], nullptr, edge_attr, ee.in_dim, nullptr, 0, ee.in_dim, g.perm,
                                ee.out_dims[0], ee.in_dim, gw, ee.in_dim, gb, nullptr, E, s));
            if (grad_edge_attr) {
                const float* Wq[2] = {ee.weight[0], nullptr};
                MPN_TRY(act_grad(1, dzq, ee.out_dims[0], nullptr, Wq, ee.in_dim, ee.out_dims[0], ee.in_dim, grad_edge_attr,
                                 ee.in_dim, g.perm, nullptr, 0, 0, nullptr, E, s));
            }
        } else if (grad_edge_attr) {
            // nothing to write
        }
    }
    return MPNHIP_OK;
}
