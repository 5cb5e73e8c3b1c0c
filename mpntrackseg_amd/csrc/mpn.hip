// Host-side orchestration of the MPN hot path on one MI355X: MOTMPNet.forward restricted to the
// encoder -> L x {reattach, MetaLayer, classifier} loop (reference models/mpn.py:349-392), expressed
// as fused-epilogue MFMA GEMMs + one segmented aggregation per step.  No allocation, no host sync.
//
// Re-association used (fp32 throughout, only the summation order differs from the reference):
//   EdgeModel  (mpn.py:67-69):  W1 [x[row] | x[col] | e] = (W1r x)[row] + (W1c x)[col] + W1e e
//   flow MLPs  (mpn.py:87-88):  Wf [x[col] | e']         = (Wfx x)[col] + Wfe e'
// The node-side products are computed once per node per step by ONE GEMM against the row-stacked
// weight block [W1r; W1c; Wfo_x; Wfi_x] (biases folded in) and added per edge in the epilogue of the
// per-edge GEMMs; torch.cat([x0, x]) / torch.cat([e0, e]) (mpn.py:369-373) are never materialised --
// the GEMM's A operand is read from two K segments.
#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "common.h"
#include "edge_chain.h"
#include "plan.h"

namespace mpnhip {

static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// ------------------------------------------------------------------------------------ path counters (common.h)
static std::atomic<long long> g_path_counts[PC_COUNT];
static const char* const g_path_names[PC_COUNT] = {
    "edge_chain_fwd", "edge_chain_fwd_split", "edge_chain_bwd", "edge_chain_bwd_split", "aggregate", "aggregate_block",
    "node_step32", "node_step32_bwd", "segment_reduce", "segment_reduce_block", "segment_reduce_block3", "edge_encoder",
    "edge_encoder_bwd", "gemm_tn_mfma", "gemm_tn_small", "gemm_tn_generic", "gemm_fp32", "gemm_split", "gemm_bf16", "weight_pack", "segment_reduce3", "gemm_splitk", "edge_chain_fwd_bf16", "gemm_tn_panel", "wgrad_panel_launches", "node_chain", "persist32", "wgrad_panel_fallback", "edge_chain_bwd_bf16", "node_chain_bwd", "gemm_bf16_tiled", "gemm_bf16_ring", "wgrad_rows16", "wgrad_rows16_launches", "wgrad_panel_narrow_launches"};
void count_path(int id) {
    if (id >= 0 && id < PC_COUNT) g_path_counts[id].fetch_add(1, std::memory_order_relaxed);
}

// ------------------------------------------------------------------------------------ event profiling
namespace {
constexpr int PROF_MAX = 2048;
struct ProfState {
    bool on = false;
    hipEvent_t ev[PROF_KINDS][PROF_MAX][2];
    double work[PROF_KINDS][PROF_MAX];
    bool made[PROF_KINDS] = {};
    int n[PROF_KINDS] = {};
    int open_kind = -1;
    bool taken = false;
    hipStream_t stream = nullptr;
    int stride = 1;        // time every stride-th launch of each kind
    int seen[PROF_KINDS] = {};
    double open_work = 0.0;
} g_prof;
}  // namespace

void prof_begin(int kind, hipStream_t s, double work) {
    if (!g_prof.on || g_prof.n[kind] >= PROF_MAX - 64) return;
    if (g_prof.seen[kind]++ % g_prof.stride != 0) return;  // sampled: the attached events are not free (~0.5 us each)
    if (!g_prof.made[kind]) {
        for (int i = 0; i < PROF_MAX; ++i) {
            (void)hipEventCreate(&g_prof.ev[kind][i][0]);
            (void)hipEventCreate(&g_prof.ev[kind][i][1]);
        }
        g_prof.made[kind] = true;
    }
    // The designated kernel's launch site picks the pair up with prof_launch_events() and hands it to
    // hipExtLaunchKernelGGL, which stamps the events with the dispatch's own begin / end times -- no record packets
    // before and after the kernel inflate a 7 us measurement.  A site that does not pick them up falls back to records.
    g_prof.open_kind = kind;
    g_prof.taken = false;
    g_prof.stream = s;
    g_prof.open_work = work;
}

bool prof_launch_events(hipEvent_t* start, hipEvent_t* stop) {
    if (!g_prof.on || g_prof.open_kind < 0 || g_prof.taken) return false;
    const int kind = g_prof.open_kind;
    *start = g_prof.ev[kind][g_prof.n[kind]][0];
    *stop = g_prof.ev[kind][g_prof.n[kind]][1];
    g_prof.taken = true;
    return true;
}

void prof_end(int kind, hipStream_t s) {
    if (!g_prof.on || g_prof.open_kind != kind || g_prof.n[kind] >= PROF_MAX - 64) return;
    if (g_prof.taken) {  // (a bracket nobody launched into is dropped)
        g_prof.work[kind][g_prof.n[kind]] = g_prof.open_work;
        g_prof.n[kind]++;
    }
    g_prof.open_kind = -1;
}

// ------------------------------------------------------------------------------------ small kernels
// dst[r][c] = src[r][c0 + c] for r < rows, c < cols  (weight sub-block copy)
__global__ void k_copy_block(const float* __restrict__ src, int64_t lds, int c0, float* __restrict__ dst, int64_t ldd,
                             int rows, int cols) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)rows * cols) return;
    int r = (int)(i / cols), c = (int)(i % cols);
    dst[(int64_t)r * ldd + c] = src[(int64_t)r * lds + c0 + c];
}

__global__ void k_copy_vec(const float* __restrict__ src, float* __restrict__ dst, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src ? src[i] : 0.f;
}

// dst[idx ? idx[r] : r][:] = src[r][:]
__global__ void k_scatter_rows(const float* __restrict__ src, const int* __restrict__ idx, float* __restrict__ dst,
                               int64_t rows, int cols) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * cols) return;
    int64_t r = i / cols;
    int c = (int)(i % cols);
    dst[(int64_t)(idx ? idx[r] : r) * cols + c] = src[i];
}

// y[r] = mean(x[r][0:hw]); `sub` lanes (power of two) share one row and reduce with shuffles
__global__ void k_avgpool(const float* __restrict__ x, int64_t rows, int hw, float* __restrict__ y, int sub) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t r = t / sub;
    int l = (int)(t % sub);
    float acc = 0.f;
    if (r < rows) {
        const float* p = x + r * hw;
        for (int i = l; i < hw; i += sub) acc += p[i];
    }
    for (int o = sub >> 1; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
    if (r < rows && l == 0) y[r] = acc / (float)hw;
}

// ------------------------------------------------------------------------------------ deferred packing (common.h)
static thread_local PackBatch* g_pack_batch = nullptr;

__device__ __forceinline__ void pack_one(const PackOp& o, int64_t i) {
    const int64_t n = (int64_t)o.rows_pad * o.cols_pad;
    if (i >= n) return;
    const int r = (int)(i / o.cols_pad), c = (int)(i % o.cols_pad);
    const bool in = o.src && r < o.rows && c < o.cols;
    if (o.transposed) o.dst[i] = in ? o.src[(int64_t)c * o.lds + o.c0 + r] : 0.f;
    else o.dst[(int64_t)r * o.ldd + o.dst_c0 + c] = in ? o.src[(int64_t)r * o.lds + o.c0 + c] : 0.f;
}

// blockIdx.y = op (static indices into the by-value argument: a dynamic one would move the whole table to scratch memory)
__global__ __launch_bounds__(256) void k_pack_multi(PackBatch b) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    switch (blockIdx.y) {
#define MPN_PACK_CASE(k) case k: pack_one(b.op[k], i); break;
        MPN_PACK_CASE(0) MPN_PACK_CASE(1) MPN_PACK_CASE(2) MPN_PACK_CASE(3) MPN_PACK_CASE(4) MPN_PACK_CASE(5) MPN_PACK_CASE(6) MPN_PACK_CASE(7)
        MPN_PACK_CASE(8) MPN_PACK_CASE(9) MPN_PACK_CASE(10) MPN_PACK_CASE(11) MPN_PACK_CASE(12) MPN_PACK_CASE(13) MPN_PACK_CASE(14) MPN_PACK_CASE(15)
        MPN_PACK_CASE(16) MPN_PACK_CASE(17) MPN_PACK_CASE(18) MPN_PACK_CASE(19) MPN_PACK_CASE(20) MPN_PACK_CASE(21) MPN_PACK_CASE(22) MPN_PACK_CASE(23)
        MPN_PACK_CASE(24) MPN_PACK_CASE(25) MPN_PACK_CASE(26) MPN_PACK_CASE(27) MPN_PACK_CASE(28) MPN_PACK_CASE(29) MPN_PACK_CASE(30) MPN_PACK_CASE(31)
#undef MPN_PACK_CASE
        default: break;
    }
}

void pack_batch_begin(PackBatch* b) {
    b->n = 0;
    g_pack_batch = getenv("MPNHIP_NO_PACK_BATCH") ? nullptr : b;
}

bool pack_batch_add(const PackOp& op) {
    PackBatch* b = g_pack_batch;
    if (!b || b->n >= PackBatch::MAX) return false;
    if ((int64_t)op.rows_pad * op.cols_pad > 0) b->op[b->n++] = op;
    return true;
}

void pack_batch_abort() { g_pack_batch = nullptr; }
int pack_batch_flush(hipStream_t s) {
    PackBatch* b = g_pack_batch;
    g_pack_batch = nullptr;
    if (!b || b->n == 0) return MPNHIP_OK;
    int64_t mx = 0;
    for (int i = 0; i < b->n; ++i) {
        const int64_t n = (int64_t)b->op[i].rows_pad * b->op[i].cols_pad;
        mx = n > mx ? n : mx;
    }
    hipLaunchKernelGGL(k_pack_multi, dim3((unsigned)((mx + 255) / 256), b->n), dim3(256), 0, s, *b);
    MPN_LAUNCH_CHECK();
    return MPNHIP_OK;
}

static int copy_block(const float* src, int64_t lds, int c0, float* dst, int64_t ldd, int rows, int cols, hipStream_t s) {
    int64_t n = (int64_t)rows * cols;
    if (n <= 0) return MPNHIP_OK;
    if (pack_batch_add({src, dst, lds, c0, rows, cols, rows, cols, (int)ldd, 0, 0})) return MPNHIP_OK;
    hipLaunchKernelGGL(k_copy_block, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, src, lds, c0, dst, ldd, rows, cols);
    MPN_LAUNCH_CHECK();
    return MPNHIP_OK;
}
static int copy_vec(const float* src, float* dst, int n, hipStream_t s) {
    if (n <= 0) return MPNHIP_OK;
    if (pack_batch_add({src, dst, n, 0, 1, n, 1, n, n, 0, 0})) return MPNHIP_OK;
    hipLaunchKernelGGL(k_copy_vec, dim3((n + 255) / 256), dim3(256), 0, s, src, dst, n);
    MPN_LAUNCH_CHECK();
    return MPNHIP_OK;
}
static int scatter_rows(const float* src, const int* idx, float* dst, int64_t rows, int cols, hipStream_t s) {
    int64_t n = rows * cols;
    if (n <= 0) return MPNHIP_OK;
    hipLaunchKernelGGL(k_scatter_rows, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, src, idx, dst, rows, cols);
    MPN_LAUNCH_CHECK();
    return MPNHIP_OK;
}

// ------------------------------------------------------------------------------------ building blocks

// hidden / output layers i >= 1 of an MLP on [rows, :] activations; `two` = direction-grouped run
// of the flow MLPs (group 0: flow_out weights on rows [0, E_out), group 1: flow_in on [E_out, E_out+E_in)).
static int mlp_tail(const mpnhip_mlp& m0, const mpnhip_mlp* m1, const GraphView* g, float* const* hidden,
                    float* out_last, int64_t ld_last, const int* c_idx_last, int64_t rows, hipStream_t s) {
    for (int i = 1; i < m0.n_layers; ++i) {
        GemmArgs a = {};
        a.ngroups = m1 ? 2 : 1;
        a.N = m0.out_dims[i];
        a.K = m0.out_dims[i - 1];
        a.ksplit = a.K;
        a.relu = m0.out_dims[i] != 1;  // mlp.py:17
        a.m_upper = rows;
        bool last = i == m0.n_layers - 1;
        for (int q = 0; q < a.ngroups; ++q) {
            const mpnhip_mlp& m = q == 0 ? m0 : *m1;
            GemmGroup& G = a.g[q];
            init_group(G);
            G.A = hidden[i - 1];
            G.lda = a.K;
            G.B = m.weight[i];
            G.ldb = a.K;
            G.bias = m.bias[i];
            G.C = last ? out_last : hidden[i];
            G.ldc = last ? ld_last : a.N;
            G.c_idx = last ? c_idx_last : nullptr;
            G.m_static = rows;
            if (m1) {
                G.row_begin = q == 0 ? nullptr : g->header + 4;
                G.row_end = q == 0 ? g->header + 4 : g->header + 5;
            }
        }
        MPN_TRY(launch_gemm(a, A_KCONTIG, B_KCONTIG, s));
    }
    return MPNHIP_OK;
}

static int pack_node_weights(const mpnhip_model& m, const Dims& d, float* Wnode, float* bnode, hipStream_t s) {
    const int he = d.he, hn = d.hn, kx = d.kx;
    MPN_TRY(copy_block(m.edge.weight[0], m.edge.in_dim, 0, Wnode, kx, he, kx, s));                         // W1 row part
    MPN_TRY(copy_block(m.edge.weight[0], m.edge.in_dim, kx, Wnode + (size_t)he * kx, kx, he, kx, s));      // W1 col part
    MPN_TRY(copy_block(m.flow_out.weight[0], m.flow_out.in_dim, 0, Wnode + (size_t)2 * he * kx, kx, hn, kx, s));
    MPN_TRY(copy_block(m.flow_in.weight[0], m.flow_in.in_dim, 0, Wnode + (size_t)(2 * he + hn) * kx, kx, hn, kx, s));
    MPN_TRY(copy_vec(m.edge.bias[0], bnode, he, s));
    MPN_TRY(copy_vec(nullptr, bnode + he, he, s));
    MPN_TRY(copy_vec(m.flow_out.bias[0], bnode + 2 * he, hn, s));
    MPN_TRY(copy_vec(m.flow_in.bias[0], bnode + 2 * he + hn, hn, s));
    return MPNHIP_OK;
}

static int pack_chain_weights(const mpnhip_model& m, const Dims& d, ChainWeights& cw, hipStream_t s) {
    cw.ok = chain_shapes_ok(m, d);
    if (!cw.ok) return MPNHIP_OK;
    const int HE = pad32(d.he), DE = pad32(d.de), HN = pad32(d.hn), DN = pad32(d.dn);
    const int hc = m.classifier.out_dims[0];
    const mpnhip_mlp* fl[2] = {&m.flow_out, &m.flow_in};
    cw.split = chain_split(m);
    if (cw.split) {
        // the same logical images WT[k][n] = W[n][k0 + k] as below, as split images (3/2 the size, offsets scale alike)
        MPN_TRY(pack_split(m.edge.weight[0] + 2 * d.kx, 1, m.edge.in_dim, d.ke, d.he, d.ke, HE, cw.w1T, s));
        MPN_TRY(pack_split(m.edge.weight[1], 1, d.he, d.he, d.de, HE, DE, cw.w2T, s));
        MPN_TRY(pack_split(m.classifier.weight[0], 1, d.de, d.de, hc, DE, 32, cw.wc1T, s));
        for (int q = 0; q < 2; ++q) {
            for (int n0 = 0; n0 < HN; n0 += 64) {
                const int ncw = HN - n0 < 64 ? HN - n0 : 64;
                const int rows = d.hn - n0 < 0 ? 0 : (d.hn - n0 < ncw ? d.hn - n0 : ncw);
                MPN_TRY(pack_split(fl[q]->weight[0] + (int64_t)n0 * fl[q]->in_dim + d.kx, 1, fl[q]->in_dim, d.de, rows, DE, ncw,
                                   cw.wf1T[q] + (int64_t)DE * n0 * 3 / 2, s));
            }
            MPN_TRY(pack_split(fl[q]->weight[1], 1, d.hn, d.hn, d.dn, HN, DN, cw.wf2T[q], s));
        }
        return MPNHIP_OK;
    }
    MPN_TRY(transpose_padded(m.edge.weight[0], m.edge.in_dim, 2 * d.kx, d.he, d.ke, cw.w1T, HE, d.ke, s));
    MPN_TRY(transpose_padded(m.edge.weight[1], d.he, 0, d.de, d.he, cw.w2T, DE, HE, s));
    MPN_TRY(transpose_padded(m.classifier.weight[0], d.de, 0, hc, d.de, cw.wc1T, 32, DE, s));
    for (int q = 0; q < 2; ++q) {
        // flow layer 0, e'-part: one image [DE][<= 64] per block of 64 output features (the kernel streams whole blocks)
        for (int n0 = 0; n0 < HN; n0 += 64) {
            const int ncw = HN - n0 < 64 ? HN - n0 : 64;
            const int rows = d.hn - n0 < 0 ? 0 : (d.hn - n0 < ncw ? d.hn - n0 : ncw);
            MPN_TRY(transpose_padded(fl[q]->weight[0] + (int64_t)n0 * fl[q]->in_dim, fl[q]->in_dim, d.kx, rows, d.de,
                                     cw.wf1T[q] + (int64_t)DE * n0, ncw, DE, s));
        }
        MPN_TRY(transpose_padded(fl[q]->weight[1], d.hn, 0, d.dn, d.hn, cw.wf2T[q], DN, HN, s));
    }
    return MPNHIP_OK;
}

static int pack_chain_bf16_weights(const mpnhip_model& m, const Dims& d, ChainBf16& cb, hipStream_t s) {
    cb.ok = cb.img && chain_bf16_ok(m, d);
    if (!cb.ok) return MPNHIP_OK;
    const float* f0[2] = {m.flow_out.weight[0], m.flow_in.weight[0]};
    const float* f1[2] = {m.flow_out.weight[1], m.flow_in.weight[1]};
    return pack_chain_bf16(m.edge.weight[0], m.edge.in_dim, 2 * d.kx, d.ef, m.edge.weight[1], m.classifier.weight[0], f0, m.flow_out.in_dim,
                           d.kx, f1, d.he, d.de, d.hn, d.dn, m.classifier.out_dims[0], cb.img, s);
}

// Reference widths, few nodes: the hoisted projections P0 = x0 Wnode[:, :dn]^T + bnode AND the first step's P = P0 + x0 Wnode[:, dn:]^T
// (x_0 IS the re-attached x0 there) in one launch, one thread per output -- two 5 us GEMM launches for 140 x 272 outputs otherwise.
__global__ __launch_bounds__(256) void k_proj_hoist32(const float* __restrict__ x0, const float* __restrict__ Wnode,
                                                     const float* __restrict__ bnode, float* __restrict__ P0, float* __restrict__ P,
                                                     int64_t total, int pw, int dn) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int64_t n = i / pw;
    const int p = (int)(i - n * pw);
    const float* w = Wnode + (int64_t)p * 2 * dn;
    const float* x = x0 + n * dn;
    float s0 = bnode[p], s1 = 0.f;
    for (int k = 0; k < dn; k += 4) {
        const float4 xv = *reinterpret_cast<const float4*>(x + k);
        const float4 wa = *reinterpret_cast<const float4*>(w + k);
        const float4 wb = *reinterpret_cast<const float4*>(w + dn + k);
        s0 = fmaf(xv.x, wa.x, s0); s0 = fmaf(xv.y, wa.y, s0); s0 = fmaf(xv.z, wa.z, s0); s0 = fmaf(xv.w, wa.w, s0);
        s1 = fmaf(xv.x, wb.x, s1); s1 = fmaf(xv.y, wb.y, s1); s1 = fmaf(xv.z, wb.z, s1); s1 = fmaf(xv.w, wb.w, s1);
    }
    P0[i] = s0;
    P[i] = s0 + s1;
}

struct StepIO {
    // node features as one or two K segments (x0 | x) -- together kx columns
    const float* xa; int64_t ldxa; const float* xb; int64_t ldxb; int kxa;
    // edge features (e0 | e) -- together ke columns; optional row indirection (original-order input)
    const float* ea; int64_t ldea; const float* eb; int64_t ldeb; int kea; const int* e_idx;
    float* e_new; const int* e_new_idx;   // [E, de]; row scatter when the caller wants original order
    const int* e_new_read_idx;            // how the flow MLPs must index e_new (nullptr = sorted order)
    float* x_new;                         // [N, dn]
    float* logits;                        // [E] original order, or nullptr (operator-level call)
    const float* P0;                      // optional: xa's share of the projections (+ biases), precomputed [N, pw];
                                          // then only xb is multiplied here (the weights are shared by all steps)
    const float* Q0;                      // optional (fused chain): ea's share of the edge MLP's first layer, [E, he]
    int p_ready;                          // the previous step's fused node kernel already wrote this step's projections into b.P
    int fuse_node;                        // 1: aggregate + node update (+ the NEXT step's projections unless `last`) in one launch
    int last;                             //    (node_step32: the reference's node width)
    float* P_next;                        //    where the next step's projections go (its step buffers; inference: the shared ones)
    const unsigned short* nc_img;         //    non-null: node_chain.hip's kernel (dn = 64 / 128, split precision) instead of node_step32
    // MPNHIP_PREC_BF16 with bf16 rows (gemm_bf16.hip): images of the packed projection / node-update weights, mirrors of xa / xb,
    // and where the mirror of x_new goes
    const unsigned short* Wnode16; const unsigned short* Wu16; const unsigned short* xa16; const unsigned short* xb16;
    unsigned short* x_new16;
};

// One MetaLayer.forward (mpn.py:33-54) (+ classifier, mpn.py:114) on prepared weights.
// e16: bf16 mirror of the edge features of this step (FwdPlan::eb_hist): [0] the re-attached initial ones, [1] the current ones,
// [2] where the new ones go; write_e32: the fp32 features of this step are needed (the last step's are returned / kept)
struct StepE16 { const unsigned short* e0; const unsigned short* cur; unsigned short* out; bool write_e32; };

static int run_step(const mpnhip_model& m, const Dims& d, const GraphView& g, const float* Wnode, const float* bnode,
                    const StepIO& io, const StepBufs& b, bool save_arg, hipStream_t s, const ChainWeights* cw = nullptr,
                    bool save_acts = false, const ChainBf16* cb = nullptr, const StepE16* e16 = nullptr) {
    unsigned short* const save_eb = e16 ? e16->out : nullptr;
    const int64_t N = g.N, E = g.E;
    const int he = d.he, hn = d.hn;
    // (1) per-node projections P = [xa | xb] Wnode^T + bnode
    if (!io.p_ready) {
        GemmArgs a = {};
        a.ngroups = 1; a.N = d.pw; a.relu = 0; a.m_upper = N;
        GemmGroup& G = a.g[0];
        init_group(G);
        if (io.Wnode16 && io.xa16) {
            // bf16 rows in memory, the whole K (both segments), biases in the epilogue: the LDS-DMA ring kernel
            a.K = d.kx; a.ksplit = io.xb16 ? io.kxa : d.kx;
            G.A = reinterpret_cast<const float*>(io.xa16); G.lda = io.ldxa; G.a16 = 1;
            G.A2 = reinterpret_cast<const float*>(io.xb16); G.lda2 = io.ldxb;
            G.B = reinterpret_cast<const float*>(io.Wnode16); G.ldb = d.kx; G.b16 = 1; G.bias = bnode;
        } else if (io.P0 && io.xb) {
            // xa (the re-attached initial features) does not change from step to step: its product is P0
            a.K = d.kx - io.kxa; a.ksplit = a.K;
            G.A = io.xb; G.lda = io.ldxb;
            G.B = Wnode + io.kxa; G.ldb = d.kx;
            G.G1 = io.P0; G.ldg1 = d.pw;
        } else {
            a.K = d.kx; a.ksplit = io.xb ? io.kxa : d.kx;
            G.A = io.xa; G.lda = io.ldxa; G.A2 = io.xb; G.lda2 = io.ldxb;
            G.B = Wnode; G.ldb = d.kx; G.bias = bnode;
        }
        G.C = b.P; G.ldc = d.pw; G.m_static = N;
        MPN_TRY(launch_gemm(a, A_KCONTIG, B_KCONTIG, s));
    }
    const bool chain = cw && cw->ok && E > 0 && io.logits && !io.e_idx && !io.e_new_idx && !io.e_new_read_idx;
    bool agg_done = false;   // the chain kernel aggregated the messages itself
    // (training: only with the bf16 save buffers of FwdPlan::b16 -- save_eb is given exactly then)
    const bool chain_bf16 = !chain && cb && cb->ok && (!save_acts || save_eb) && E > 0 && io.logits && io.eb && !io.e_idx && !io.e_new_idx &&
                            !io.e_new_read_idx;
    if (e16 && !chain_bf16) { set_error("forward: the bf16 mirror of the edge features needs the bf16 chain kernel"); return MPNHIP_ERR_ARG; }
    if (chain_bf16) {
        // (2)-(4) fused, bf16 operands / fp32 accumulation (edge_chain_bf16.hip)
        EdgeChainBf16Args a = {};
        a.E = (int)E; a.N = (int)N; a.header = g.header; a.srow = g.srow; a.scol = g.scol; a.perm = g.perm;
        a.he = d.he; a.de = d.de; a.hn = d.hn; a.dn = d.dn; a.hc = m.classifier.out_dims[0];
        a.xa = io.ea; a.ldxa = io.ldea; a.xb = io.eb; a.ldxb = io.ldeb;
        a.P = b.P; a.pw = d.pw;
        a.img_edge = cb->img; a.img_cls = cb->img + cb->off_cls; a.img_flow[0] = cb->img + cb->off_flow[0]; a.img_flow[1] = cb->img + cb->off_flow[1];
        a.b2 = m.edge.bias[1]; a.bc1 = m.classifier.bias[0]; a.wc2 = m.classifier.weight[1]; a.bc2 = m.classifier.bias[1];
        a.bf2_out = m.flow_out.bias[1]; a.bf2_in = m.flow_in.bias[1];
        a.e_new = io.e_new; a.msg = b.M; a.logits = io.logits;
        // the aggregation of the messages inside the kernel (they never reach HBM); MPNHIP_NO_AGG_FUSION=1: k_aggregate as before
        // (training with max keeps the separate kernel: it records the arg max the backward needs)
        agg_done = cb->piece && !io.fuse_node && !save_arg && !getenv("MPNHIP_NO_AGG_FUSION");
        if (e16) {
            a.xa16 = e16->e0; a.xb16 = e16->cur; a.e16_out = e16->out;
            if (!e16->write_e32) a.e_new = nullptr;
        }
        if (save_acts) {
            a.save_h1 = reinterpret_cast<unsigned short*>(b.HE[0]);
            a.save_hc = reinterpret_cast<unsigned short*>(b.HC[0]);
            a.save_hf = reinterpret_cast<unsigned short*>(b.HF[0]);
            a.save_eb = save_eb;
            a.save_mask = reinterpret_cast<unsigned*>(b.MK);
        }
        if (agg_done) {
            a.seg_ptr = g.seg_ptr; a.agg_out = b.AGG; a.piece = cb->piece; a.start_row = cb->start_row; a.agg = m.agg;
            MPN_HIP(hipMemsetAsync(b.AGG, 0, (size_t)N * 2 * d.dn * sizeof(float), s));   // (empty segments)
        }
        prof_begin(PROF_GEMM, s);
        MPN_TRY(launch_edge_chain_bf16(a, s));
        prof_end(PROF_GEMM, s);
    } else if (chain) {
        // (2)-(4) fused: edge MLP, classifier and both flow MLPs in one kernel (edge_chain.hip)
        EdgeChainArgs a = {};
        a.E = (int)E; a.N = (int)N; a.header = g.header; a.srow = g.srow; a.scol = g.scol; a.perm = g.perm; a.split = cw->split ? 1 : 0;
        a.he = d.he; a.de = d.de; a.hn = d.hn; a.dn = d.dn; a.hc = m.classifier.out_dims[0];
        a.xa = io.ea; a.ldxa = io.ldea; a.k1a = io.eb ? io.kea : d.ke;
        a.xb = io.eb; a.ldxb = io.ldeb; a.k1b = io.eb ? d.ke - io.kea : 0;
        a.P = b.P; a.pw = d.pw;
        a.w1T = cw->w1T; a.w2T = cw->w2T; a.b2 = m.edge.bias[1];
        if (io.Q0 && io.eb) {
            // ea = the re-attached initial edge features: their product with W1's e0 columns is Q0 (once per forward);
            // the kernel multiplies the current features only (rows kea.. of the [k][n] weight image)
            a.Q0 = io.Q0;
            a.xa = io.eb; a.ldxa = io.ldeb; a.k1a = d.ke - io.kea;
            a.xb = nullptr; a.ldxb = 0; a.k1b = 0;
            a.w1T = cw->w1T + (size_t)io.kea * pad32(d.he) * (cw->split ? 3 : 2) / 2;
        }
        a.wc1T = cw->wc1T; a.bc1 = m.classifier.bias[0]; a.wc2 = m.classifier.weight[1]; a.bc2 = m.classifier.bias[1];
        a.wf1T_out = cw->wf1T[0]; a.wf1T_in = cw->wf1T[1]; a.wf2T_out = cw->wf2T[0]; a.wf2T_in = cw->wf2T[1];
        a.bf2_out = m.flow_out.bias[1]; a.bf2_in = m.flow_in.bias[1];
        a.e_new = io.e_new; a.msg = b.M; a.logits = io.logits;
        a.save_h1 = save_acts ? b.HE[0] : nullptr;
        a.save_hc = save_acts ? b.HC[0] : nullptr;
        a.save_hf = save_acts ? b.HF[0] : nullptr;
        a.save_mask = save_acts ? reinterpret_cast<unsigned*>(b.MK) : nullptr;
        prof_begin(PROF_GEMM, s);
        MPN_TRY(launch_edge_chain(a, s));
        prof_end(PROF_GEMM, s);
    } else if (E > 0) {
        // (2) edge MLP layer 0: relu([ea | eb] W1e^T + P_r[row] + P_c[col])      (EdgeModel, mpn.py:67-69)
        {
            GemmArgs a = {};
            a.ngroups = 1; a.N = he; a.K = d.ke; a.ksplit = io.eb ? io.kea : d.ke; a.m_upper = E;
            a.relu = he != 1;
            GemmGroup& G = a.g[0];
            init_group(G);
            G.A = io.ea; G.lda = io.ldea; G.A2 = io.eb; G.lda2 = io.ldeb; G.a_idx = io.e_idx;
            G.B = m.edge.weight[0] + 2 * d.kx; G.ldb = m.edge.in_dim;
            G.G1 = b.P; G.g1_idx = g.srow; G.ldg1 = d.pw;
            G.G2 = b.P + he; G.g2_idx = g.scol; G.ldg2 = d.pw;
            bool last = m.edge.n_layers == 1;
            G.C = last ? io.e_new : b.HE[0]; G.ldc = last ? d.de : he; G.c_idx = last ? io.e_new_idx : nullptr;
            G.m_static = E;
            prof_begin(PROF_GEMM, s);
            MPN_TRY(launch_gemm(a, A_KCONTIG, B_KCONTIG, s));
            prof_end(PROF_GEMM, s);
        }
        MPN_TRY(mlp_tail(m.edge, nullptr, nullptr, b.HE, io.e_new, d.de, io.e_new_idx, E, s));
        // (3) classifier on the NEW edge features (mpn.py:377 -> :114)
        if (io.logits) {
            const mpnhip_mlp& c = m.classifier;
            GemmArgs a = {};
            a.ngroups = 1; a.N = c.out_dims[0]; a.K = d.de; a.ksplit = d.de; a.relu = c.out_dims[0] != 1; a.m_upper = E;
            GemmGroup& G = a.g[0];
            init_group(G);
            G.A = io.e_new; G.lda = d.de; G.a_idx = io.e_new_read_idx; G.B = c.weight[0]; G.ldb = d.de; G.bias = c.bias[0];
            bool last = c.n_layers == 1;
            G.C = last ? io.logits : b.HC[0]; G.ldc = last ? 1 : a.N; G.c_idx = last ? g.perm : nullptr; G.m_static = E;
            MPN_TRY(launch_gemm(a, A_KCONTIG, B_KCONTIG, s));
            MPN_TRY(mlp_tail(c, nullptr, nullptr, b.HC, io.logits, 1, g.perm, E, s));
        }
        // (4) flow MLP layer 0, both directions in one grouped launch (TimeAwareNodeModel, mpn.py:85-94)
        {
            GemmArgs a = {};
            a.ngroups = 2; a.N = hn; a.K = d.de; a.ksplit = d.de; a.relu = hn != 1; a.m_upper = E;
            bool last = m.flow_in.n_layers == 1;
            for (int q = 0; q < 2; ++q) {
                const mpnhip_mlp& f = q == 0 ? m.flow_out : m.flow_in;
                GemmGroup& G = a.g[q];
                init_group(G);
                G.A = io.e_new; G.lda = d.de; G.a_idx = io.e_new_read_idx;
                G.B = f.weight[0] + d.kx; G.ldb = f.in_dim;
                G.G1 = b.P + 2 * he + q * hn; G.g1_idx = g.scol; G.ldg1 = d.pw;
                G.C = last ? b.M : b.HF[0]; G.ldc = last ? d.dn : hn;
                G.m_static = E;
                G.row_begin = q == 0 ? nullptr : g.header + 4;
                G.row_end = q == 0 ? g.header + 4 : g.header + 5;
            }
            MPN_TRY(launch_gemm(a, A_KCONTIG, B_KCONTIG, s));
        }
        MPN_TRY(mlp_tail(m.flow_out, &m.flow_in, &g, b.HF, b.M, d.dn, nullptr, E, s));
    }
    // (5) aggregation (node_agg_fn, mpn.py:89,96) and node update (mpn.py:97-99)
    if (io.fuse_node && io.nc_img) {
        size_t off_wx = 0;
        node_chain_image_shorts(d.dn, d.pw, &off_wx);
        NodeChainArgs a = {(int)N, d.dn, d.pw, m.agg, g.seg_ptr, b.M, io.nc_img, m.node.bias[0], io.nc_img + off_wx, io.P0,
                           io.last ? nullptr : io.P_next, io.x_new, save_acts ? b.AGG : nullptr};
        // (profiled under the aggregation's kind: node_agg_fn runs inside this launch)
        prof_begin(PROF_AGG, s);
        const int st_nc = launch_node_chain(a, s);
        prof_end(PROF_AGG, s);
        return st_nc;
    }
    if (io.fuse_node) {
        MPN_TRY(node_step32(g, b.M, m.agg, m.node.weight[0], m.node.bias[0], io.x_new, save_acts ? b.AGG : nullptr, Wnode + io.kxa, d.kx,
                            io.P0, io.last ? nullptr : io.P_next, d.pw, s));
        return MPNHIP_OK;
    }
    if (!agg_done) {
        prof_begin(PROF_AGG, s);
        MPN_TRY(aggregate(g, b.M, d.dn, m.agg, b.AGG, save_arg ? b.ARG : nullptr, s));
        prof_end(PROF_AGG, s);
    }
    if (io.Wu16 && N > 0) {
        // bf16 weight image; the bf16 mirror of the new node features (the next step's projections read it) leaves with the result
        GemmArgs a = {};
        a.ngroups = 1; a.N = d.dn; a.K = 2 * d.dn; a.ksplit = a.K; a.relu = 1; a.m_upper = N;
        GemmGroup& G = a.g[0];
        init_group(G);
        G.A = b.AGG; G.lda = 2 * d.dn; G.B = reinterpret_cast<const float*>(io.Wu16); G.ldb = 2 * d.dn; G.b16 = 1;
        G.bias = m.node.bias[0]; G.C = io.x_new; G.ldc = d.dn; G.C16 = io.x_new16; G.ldc16 = d.dn; G.m_static = N;
        return launch_gemm(a, A_KCONTIG, B_KCONTIG, s);
    }
    MPN_TRY(linear(b.AGG, 2 * d.dn, m.node.weight[0], m.node.bias[0], io.x_new, d.dn, N, d.dn, 2 * d.dn, 1, s));
    return MPNHIP_OK;
}

// ---- the reference's edge encoder (6 -> 18 -> 18 -> 16, ReLU after every layer; tracking_cfg.yaml:136-139) in one launch ----
// Three Linear layers this narrow are three launches of the any-shape GEMM kernel (one thread per OUTPUT element, ~25 us each at
// 78k edges): here one thread carries one edge through all three layers (720 FMAs), the weights sit in LDS and are read as
// broadcasts, the hidden activations stay in registers (written out only when the backward pass needs them).
template <int IN, int H1, int H2, int OUT>
__global__ __launch_bounds__(256) void k_edge_encoder(const float* __restrict__ x, const int* __restrict__ idx, int64_t rows,
                                                      const float* __restrict__ w0, const float* __restrict__ b0,
                                                      const float* __restrict__ w1, const float* __restrict__ b1,
                                                      const float* __restrict__ w2, const float* __restrict__ b2,
                                                      float* __restrict__ h1_out, float* __restrict__ h2_out, float* __restrict__ y) {
    __shared__ float sw0[H1 * IN], sb0[H1], sw1[H2 * H1], sb1[H2], sw2[OUT * H2], sb2[OUT];
    for (int i = threadIdx.x; i < H1 * IN; i += 256) sw0[i] = w0[i];
    for (int i = threadIdx.x; i < H2 * H1; i += 256) sw1[i] = w1[i];
    for (int i = threadIdx.x; i < OUT * H2; i += 256) sw2[i] = w2[i];
    if (threadIdx.x < H1) sb0[threadIdx.x] = b0[threadIdx.x];
    if (threadIdx.x < H2) sb1[threadIdx.x] = b1[threadIdx.x];
    if (threadIdx.x < OUT) sb2[threadIdx.x] = b2[threadIdx.x];
    __syncthreads();
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= rows) return;
    const float* xr = x + (int64_t)(idx ? idx[r] : r) * IN;
    float a[IN], h1[H1], h2[H2];
#pragma unroll
    for (int k = 0; k < IN; ++k) a[k] = xr[k];
#pragma unroll
    for (int o = 0; o < H1; ++o) {
        float s = sb0[o];
#pragma unroll
        for (int k = 0; k < IN; ++k) s = fmaf(a[k], sw0[o * IN + k], s);
        h1[o] = fmaxf(s, 0.f);
    }
#pragma unroll
    for (int o = 0; o < H2; ++o) {
        float s = sb1[o];
#pragma unroll
        for (int k = 0; k < H1; ++k) s = fmaf(h1[k], sw1[o * H1 + k], s);
        h2[o] = fmaxf(s, 0.f);
    }
    if (h1_out) {
#pragma unroll
        for (int o = 0; o < H1; ++o) h1_out[r * H1 + o] = h1[o];
#pragma unroll
        for (int o = 0; o < H2; ++o) h2_out[r * H2 + o] = h2[o];
    }
#pragma unroll
    for (int o = 0; o < OUT; ++o) {
        float s = sb2[o];
#pragma unroll
        for (int k = 0; k < H2; ++k) s = fmaf(h2[k], sw2[o * H2 + k], s);
        y[r * OUT + o] = fmaxf(s, 0.f);
    }
}

// true (and launched) when `m` is exactly that encoder; hidden[0..1] = where the backward wants the activations, or nullptr
static bool edge_encoder_fused(const mpnhip_mlp& m, const float* x, const int* idx, float* const* hidden, bool keep_hidden, float* y,
                               int64_t rows, hipStream_t s, int* status) {
    *status = MPNHIP_OK;
    if (getenv("MPNHIP_NO_ENCODER_FUSION")) return false;
    if (m.n_layers == 3 && rows > 0 && !(m.in_dim == 6 && m.out_dims[0] == 18 && m.out_dims[1] == 18 && m.out_dims[2] == 16)) {
        // the wider models' encoder (e.g. 6 -> 72 -> 72 -> 64 at d = 128): the MFMA form (edge_chain.hip, k_edge_encoder_mfma)
        const float* w[3] = {m.weight[0], m.weight[1], m.weight[2]};
        const float* b[3] = {m.bias[0], m.bias[1], m.bias[2]};
        const int dims[3] = {m.out_dims[0], m.out_dims[1], m.out_dims[2]};
        const int r = launch_edge_encoder_mfma(x, idx, rows, m.in_dim, w, b, dims, keep_hidden ? hidden[0] : nullptr, keep_hidden ? hidden[1] : nullptr, y, s);
        if (r < 0) *status = r;
        if (r != 0) count_path(PC_EDGE_ENCODER);
        return r != 0;
    }
    if (m.n_layers != 3 || m.in_dim != 6 || m.out_dims[0] != 18 || m.out_dims[1] != 18 || m.out_dims[2] != 16 || rows <= 0) return false;
    count_path(PC_EDGE_ENCODER);
    hipLaunchKernelGGL((k_edge_encoder<6, 18, 18, 16>), dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, s, x, idx, rows, m.weight[0],
                       m.bias[0], m.weight[1], m.bias[1], m.weight[2], m.bias[2], keep_hidden ? hidden[0] : nullptr,
                       keep_hidden ? hidden[1] : nullptr, y);
    if (hipGetLastError() != hipSuccess) { set_error("edge encoder: launch failed"); *status = MPNHIP_ERR_HIP; }
    return true;
}

// full MLP (all layers) with ping-pong or per-layer hidden buffers; a_idx permutes the input rows
static int mlp_forward(const mpnhip_mlp& m, const float* x, int64_t ldx, const int* a_idx, float* const* hidden, float* y,
                       int64_t rows, hipStream_t s, float* splitk = nullptr, size_t splitk_floats = 0) {
    for (int i = 0; i < m.n_layers; ++i) {
        if (splitk && !a_idx) {
            int st = MPNHIP_OK;
            const int k_in = i == 0 ? m.in_dim : m.out_dims[i - 1];
            // (the layer after it rides along when it is the MLP's last and narrow: the reference's 2048 -> 128 -> 32 node encoder)
            SplitkNext nx = {};
            const bool has_next = i + 2 == m.n_layers;
            if (has_next) nx = {m.weight[i + 1], m.bias[i + 1], m.out_dims[i + 1], m.out_dims[i + 1] != 1, y, m.out_dims[i + 1], false};
            if (linear_splitk(i == 0 ? x : hidden[i - 1], i == 0 ? ldx : k_in, m.weight[i], m.bias[i], i == m.n_layers - 1 ? y : hidden[i],
                              m.out_dims[i], rows, m.out_dims[i], k_in, m.out_dims[i] != 1, splitk, splitk_floats, s, &st,
                              has_next ? &nx : nullptr)) {
                if (st != MPNHIP_OK) return st;
                if (nx.done) ++i;   // both layers are evaluated
                continue;
            }
        }
        GemmArgs a = {};
        a.ngroups = 1;
        a.N = m.out_dims[i];
        a.K = i == 0 ? m.in_dim : m.out_dims[i - 1];
        a.ksplit = a.K;
        a.relu = m.out_dims[i] != 1;
        a.m_upper = rows;
        GemmGroup& G = a.g[0];
        init_group(G);
        G.A = i == 0 ? x : hidden[i - 1];
        G.lda = i == 0 ? ldx : a.K;
        G.a_idx = i == 0 ? a_idx : nullptr;
        G.B = m.weight[i];
        G.ldb = a.K;
        G.bias = m.bias[i];
        G.C = i == m.n_layers - 1 ? y : hidden[i];
        G.ldc = a.N;
        G.m_static = rows;
        MPN_TRY(launch_gemm(a, A_KCONTIG, B_KCONTIG, s));
    }
    return MPNHIP_OK;
}


}  // namespace mpnhip

using namespace mpnhip;

extern "C" const char* mpnhip_version(void) { return "mpnhip 0.1 (gfx950)"; }
extern "C" const char* mpnhip_last_error(void) { return g_err; }

extern "C" int mpnhip_debug_counters(int64_t* counts, int capacity, int reset) {
    for (int i = 0; i < PC_COUNT; ++i) {
        if (counts && i < capacity) counts[i] = (int64_t)g_path_counts[i].load(std::memory_order_relaxed);
        if (reset) g_path_counts[i].store(0, std::memory_order_relaxed);
    }
    return PC_COUNT;
}
extern "C" const char* mpnhip_debug_counter_name(int index) { return index >= 0 && index < PC_COUNT ? g_path_names[index] : ""; }

extern "C" size_t mpnhip_forward_workspace_bytes(const mpnhip_model* model, int n_nodes, int64_t n_edges, int save) {
    Dims d;
    if (!model || check_full(*model, &d, false) != MPNHIP_OK) return 0;
    return plan_forward(*model, d, n_nodes, n_edges, save, nullptr, nullptr);
}

extern "C" int mpnhip_forward(const mpnhip_model* model, const void* graph_buf, int n_nodes, int64_t n_edges,
                              const float* x, const float* edge_attr, float* logits, float* x_out, float* e_out,
                              void* workspace, size_t workspace_bytes, int save, void* stream_) {
    hipStream_t s = static_cast<hipStream_t>(stream_);
    MPN_CHECK_ARG(model && graph_buf, "forward: null model / graph");
    const mpnhip_model& m = *model;
    Dims d;
    MPN_TRY(check_full(m, &d));
    const int64_t N = n_nodes, E = n_edges;
    MPN_CHECK_ARG(N >= 0 && E >= 0, "forward: negative sizes");
    MPN_CHECK_ARG((x || N == 0) && (edge_attr || E == 0) && (logits || E == 0), "forward: null tensor");
    FwdPlan p;
    size_t need = plan_forward(m, d, N, E, save, workspace, &p);
    if (!workspace || workspace_bytes < need) {
        set_error("forward: workspace %zu < %zu", workspace_bytes, need);
        return MPNHIP_ERR_WORKSPACE;
    }
    GraphView g;
    graph_layout(n_nodes, n_edges, &g, const_cast<void*>(graph_buf));
    MPN_CHECK_ARG(m.precision == MPNHIP_PREC_FP32 || m.precision == MPNHIP_PREC_BF16 || m.precision == MPNHIP_PREC_FP32_SPLIT ||
                  m.precision == MPNHIP_PREC_FP32_WGSPLIT, "forward: unknown precision %d", m.precision);
    // every product of this call rounds its operands as the model asks (restored on every exit path)
    struct PrecisionScope {
        int old;
        explicit PrecisionScope(int p) : old(gemm_precision()) { set_gemm_precision(p); }
        ~PrecisionScope() { set_gemm_precision(old); }
    } precision_scope(m.precision == MPNHIP_PREC_FP32_WGSPLIT ? MPNHIP_PREC_FP32 : m.precision);  // (FP32_SPLIT: the fused chain
    // kernels, and the larger K-contiguous GEMMs -- gemm.hip; FP32_WGSPLIT differs from FP32 in the backward's weight gradients only)

    if (m.weights_prepacked && !save) {
        p.cw.ok = chain_shapes_ok(m, d);  // the images are already at the head of the workspace
        p.cw.split = chain_split(m);
        p.cb.ok = p.cb.img && chain_bf16_ok(m, d);
    } else {
        count_path(PC_WEIGHT_PACK);
        PackBatch pb;
        SplitBatch sb;
        PackBatchGuard pbg;
        SplitBatchGuard sbg;
        pack_batch_begin(&pb);   // (the fp32 images' copies / transposes are recorded and run as one launch,
        split_batch_begin(&sb);  //  the split images as another)
        int rc = pack_node_weights(m, d, p.Wnode, p.bnode, s);
        if (rc == MPNHIP_OK) rc = pack_chain_weights(m, d, p.cw, s);
        if (rc == MPNHIP_OK) rc = pack_chain_bf16_weights(m, d, p.cb, s);
        const int rf = pack_batch_flush(s);
        const int rs = split_batch_flush(s);
        MPN_TRY(rc);
        MPN_TRY(rf);
        MPN_TRY(rs);
        // (after the flush: the unit images of the fused node-side kernel read the packed projection weights)
        if (p.nc_img && m.precision == MPNHIP_PREC_FP32_SPLIT)
            MPN_TRY(pack_node_chain(m.node.weight[0], p.Wnode, d.dn, d.pw, d.kx, p.nc_img, s));
        // ... and so do the bf16 images of the node-side weights (gemm_bf16.hip reads bf16 rows)
        if (p.Wnode16 && ((size_t)d.pw * d.kx) % 4 == 0 && ((size_t)d.dn * 2 * d.dn) % 4 == 0 && (((uintptr_t)m.node.weight[0]) & 15) == 0) {
            MPN_TRY(to_bf16_rows(p.Wnode, p.Wnode16, (int64_t)d.pw * d.kx, s));
            MPN_TRY(to_bf16_rows(m.node.weight[0], p.Wu16, (int64_t)d.dn * 2 * d.dn, s));
        }
    }
    // bf16 rows for the node-side products of this call (weights at the head of the workspace: kept with the other images)
    const bool rows16 = node_rows16_runtime(p, m, d);   // (plan.h: the backward applies the same test)
    // encoder (MLPGraphIndependent, mpn.py:355 -> :164-178); the edge encoder reads edge_attr through
    // the sort permutation so that every per-edge tensor downstream lives in sorted order
    float* hid[MPNHIP_MAX_LAYERS];
    float* x0 = p.x_hist;
    float* e0 = p.e_hist;
    hidden_ptrs(m.enc_node, p.enc_n, N, save != 0, hid);
    MPN_TRY(mlp_forward(m.enc_node, x, m.enc_node.in_dim, nullptr, hid, x0, N, s, p.splitk, p.splitk_floats));
    hidden_ptrs(m.enc_edge, p.enc_e, E, save != 0, hid);
    {
        int st = MPNHIP_OK;
        // (bf16-operand mode: every Linear product rounds its operands -- the GEMM path does that, the fused kernel is fp32)
        if (m.precision == MPNHIP_PREC_BF16 || !edge_encoder_fused(m.enc_edge, edge_attr, g.perm, hid, save != 0, e0, E, s, &st))
            MPN_TRY(mlp_forward(m.enc_edge, edge_attr, m.enc_edge.in_dim, g.perm, hid, e0, E, s));
        else if (st != MPNHIP_OK) return st;
    }

    const size_t xs = (size_t)N * d.dn, es = (size_t)E * d.de;
    if (p.eb_hist && p.cb.ok && (!save || p.b16) && es && d.L > 0) MPN_TRY(to_bf16_rows(e0, p.eb_hist, (int64_t)es, s));   // the encoder output as slot 0 of the bf16 mirror
    if (rows16 && xs && d.L > 0) MPN_TRY(to_bf16_rows(x0, p.xb_hist, (int64_t)xs, s));   // the encoder output as slot 0 of the node mirror
    // (bf16 rows: the projections multiply [x0 | x] whole -- twice the MFMAs, which this product has to spare, instead of streaming
    // the [N, pw] fp32 table P0 back in every step)
    const bool hoist = d.nf == 2 && d.L > 1 && !rows16;
    // few nodes at the reference's width: P0 and the first step's projections by one small kernel (decided with fuse_node below)
    const bool proj_small = hoist && d.dn == 32 && N > 0 && N <= 4096 && d.kx == 2 * d.dn && m.precision != MPNHIP_PREC_BF16 &&
                            ((((uintptr_t)p.Wnode) | ((uintptr_t)p.P0) | ((uintptr_t)x0)) & 15) == 0 && !getenv("MPNHIP_NO_NODE_FUSION");
    if (hoist && !proj_small) {  // P0 = x0 Wnode[:, :dn]^T + bnode, once per forward
        GemmArgs a = {};
        a.ngroups = 1; a.N = d.pw; a.K = d.dn; a.ksplit = d.dn; a.m_upper = N;
        GemmGroup& G = a.g[0];
        init_group(G);
        G.A = x0; G.lda = d.dn; G.B = p.Wnode; G.ldb = d.kx; G.bias = p.bnode; G.C = p.P0; G.ldc = d.pw; G.m_static = N;
        MPN_TRY(launch_gemm(a, A_KCONTIG, B_KCONTIG, s));
    }
    // the same for the edge side of the fused chain: Q0 = e0 W1[:, e0 columns]^T, [E, he] (64 MB at cfg-B), saves a
    // quarter of the chain's first-layer MFMAs in every step
    // (measured: pays from de = 32 up; at the reference's de = 16 the extra C-in loads cost more than the one chunk saved)
    // (FP32_SPLIT: the chain kernel is bound by its row traffic, not by MFMA cycles -- re-reading 4 he bytes per edge and step
    // costs more than the k blocks it saves: measured 1.59 -> 1.55 ms per cfg-B forward without the hoist)
    const bool hoist_e = p.cw.ok && !p.cw.split && d.ef == 2 && d.L > 1 && E > 0 && d.de >= 32 && (d.ke - d.de) % 16 == 0 &&
                         d.de % 16 == 0 && !getenv("MPNHIP_NO_Q0");
    if (hoist_e) {
        GemmArgs a = {};
        a.ngroups = 1; a.N = d.he; a.K = d.de; a.ksplit = d.de; a.m_upper = E;
        GemmGroup& G = a.g[0];
        init_group(G);
        G.A = e0; G.lda = d.de; G.B = m.edge.weight[0] + 2 * d.kx; G.ldb = m.edge.in_dim; G.C = p.Q0; G.ldc = d.he; G.m_static = E;
        MPN_TRY(launch_gemm(a, A_KCONTIG, B_KCONTIG, s));
    }
    // the reference's node width: the three node-side kernels of a step in one launch (segment.hip, node_step32); needs the
    // hoisted P0 form of the projections and 16-byte aligned weights; training with max aggregation keeps the separate
    // kernels (the fused one does not record the arg max)
    // (every 2-node block re-reads the 43 KB of node-side weights from L2: beyond a few thousand nodes the GEMMs win again)
    const bool fuse_node32 = (!save || m.agg != MPNHIP_AGG_MAX) && hoist && d.dn == 32 && d.pw % 4 == 0 && E > 0 && N > 0 && N <= 4096 && m.precision != MPNHIP_PREC_BF16 &&
                             ((((uintptr_t)m.node.weight[0]) | ((uintptr_t)p.P0) | ((uintptr_t)p.Wnode)) & 15) == 0 &&
                             !getenv("MPNHIP_NO_NODE_FUSION");
    // wider models in the split precision: the same three launches as one (node_chain.hip)
    const bool fuse_node_chain = (!save || m.agg != MPNHIP_AGG_MAX) && hoist && p.nc_img && m.precision == MPNHIP_PREC_FP32_SPLIT && E > 0 && N > 0 &&
                                 m.node.n_layers == 1 && (((uintptr_t)m.node.bias[0]) & 15) == 0;
    const bool fuse_node = fuse_node32 || fuse_node_chain;
    if (proj_small) {
        const int64_t total = N * d.pw;
        hipLaunchKernelGGL(k_proj_hoist32, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, x0, p.Wnode, p.bnode, p.P0, step_at(p, 0).P,
                           total, d.pw, d.dn);
        MPN_LAUNCH_CHECK();
    }
    int prev = 0;
    // inference at the reference's widths: the whole step loop in one launch (persist32.hip); every fp32-class precision (the
    // kernel computes on fp32 MFMAs)
    const bool persist = !save && proj_small && d.ef == 2 && d.L >= 1 && m.precision != MPNHIP_PREC_BF16 && m.edge.n_layers == 2 &&
                         m.flow_in.n_layers == 2 && m.flow_out.n_layers == 2 && m.classifier.n_layers == 2 && m.classifier.out_dims[1] == 1 &&
                         m.node.n_layers == 1 && p.P_alt && (((uintptr_t)e0) & 15) == 0 &&
                         persist32_supported(d.dn, d.de, d.he, d.hn, m.classifier.out_dims[0], d.pw, d.kx, N, E);
    if (persist) {
        Persist32Args a = {};
        a.N = (int)N; a.L = d.L; a.agg = m.agg; a.pw = d.pw; a.he = d.he; a.hn = d.hn; a.hc = m.classifier.out_dims[0]; a.E = E;
        a.seg_ptr = g.seg_ptr; a.srow = g.srow; a.scol = g.scol; a.perm = g.perm;
        a.e0 = e0; a.e = p.e_hist + es; a.P0 = p.P0; a.P[0] = step_at(p, 0).P; a.P[1] = p.P_alt;
        a.W1 = m.edge.weight[0]; a.ld_w1 = m.edge.in_dim; a.col_w1 = 2 * d.kx;
        a.W2 = m.edge.weight[1]; a.Wc1 = m.classifier.weight[0];
        a.Wf1[0] = m.flow_out.weight[0]; a.Wf1[1] = m.flow_in.weight[0]; a.ld_wf1 = m.flow_out.in_dim; a.col_wf1 = d.kx;
        a.Wf2[0] = m.flow_out.weight[1]; a.Wf2[1] = m.flow_in.weight[1];
        a.Wu = m.node.weight[0]; a.Wnode = p.Wnode;
        a.b2 = m.edge.bias[1]; a.bc1 = m.classifier.bias[0]; a.wc2 = m.classifier.weight[1]; a.bc2 = m.classifier.bias[1];
        a.bf2[0] = m.flow_out.bias[1]; a.bf2[1] = m.flow_in.bias[1]; a.bu = m.node.bias[0];
        a.logits = logits; a.x_out = p.x_hist + xs; a.barrier = reinterpret_cast<unsigned*>(p.barrier);
        MPN_TRY(launch_persist32(a, s));
        prev = 1;
    }
    for (int step = 0; step < d.L && !persist; ++step) {
        int cur = save ? step + 1 : 1 + (step & 1);
        StepBufs b = step_at(p, step);
        StepIO io = {};
        const float* xp = p.x_hist + xs * prev;
        const float* ep = p.e_hist + es * prev;
        if (d.nf == 2) { io.xa = x0; io.ldxa = d.dn; io.xb = xp; io.ldxb = d.dn; io.kxa = d.dn; }
        else { io.xa = xp; io.ldxa = d.dn; }
        if (d.ef == 2) { io.ea = e0; io.ldea = d.de; io.eb = ep; io.ldeb = d.de; io.kea = d.de; }
        else { io.ea = ep; io.ldea = d.de; }
        io.e_new = p.e_hist + es * cur;
        io.x_new = p.x_hist + xs * cur;
        io.logits = logits + (size_t)step * E;
        io.P0 = hoist ? p.P0 : nullptr;
        if (rows16) {
            io.Wnode16 = p.Wnode16; io.Wu16 = p.Wu16;
            io.xa16 = d.nf == 2 ? p.xb_hist : p.xb_hist + xs * prev;
            io.xb16 = d.nf == 2 ? p.xb_hist + xs * prev : nullptr;
            io.x_new16 = p.xb_hist + xs * cur;
        }
        io.Q0 = hoist_e ? p.Q0 : nullptr;
        io.fuse_node = fuse_node ? 1 : 0;
        io.nc_img = fuse_node_chain ? p.nc_img : nullptr;
        io.p_ready = (fuse_node && step > 0) || (proj_small && step == 0) ? 1 : 0;
        io.last = step + 1 == d.L ? 1 : 0;
        io.P_next = io.last ? nullptr : step_at(p, step + 1).P;
        // (bf16-operand chain: the edge features travel between the steps as bf16 rows; the fp32 ones only where they are returned)
        // (a graph with nodes and no edges launches no chain kernel: nothing mirrors, run_step takes its E == 0 path)
        const bool e16_on = p.eb_hist && p.cb.ok && (!save || p.b16) && E > 0;
        const StepE16 e16 = {p.eb_hist, p.eb_hist ? p.eb_hist + es * prev : nullptr, p.eb_hist ? p.eb_hist + es * cur : nullptr,
                             step + 1 == d.L || getenv("MPNHIP_CHAIN_BF16_KEEP_E32") != nullptr};
        MPN_TRY(run_step(m, d, g, p.Wnode, p.bnode, io, b, save && m.agg == MPNHIP_AGG_MAX, s, &p.cw, save != 0, &p.cb, e16_on ? &e16 : nullptr));
        prev = cur;
    }
    if (d.L == 0 && E > 0) {
        // mpn.py:387-389: classify the encoder output once
        StepBufs b = p.step0;
        const mpnhip_mlp& c = m.classifier;
        float* hidc[MPNHIP_MAX_LAYERS];
        for (int i = 0; i < MPNHIP_MAX_LAYERS; ++i) hidc[i] = b.HC[i];
        if (c.n_layers == 1) {
            GemmArgs a = {};
            a.ngroups = 1; a.N = 1; a.K = d.de; a.ksplit = d.de; a.relu = 0; a.m_upper = E;
            GemmGroup& G = a.g[0];
            init_group(G);
            G.A = e0; G.lda = d.de; G.B = c.weight[0]; G.ldb = d.de; G.bias = c.bias[0];
            G.C = logits; G.ldc = 1; G.c_idx = g.perm; G.m_static = E;
            MPN_TRY(launch_gemm(a, A_KCONTIG, B_KCONTIG, s));
        } else {
            MPN_TRY(linear(e0, d.de, c.weight[0], c.bias[0], hidc[0], c.out_dims[0], E, c.out_dims[0], d.de,
                           c.out_dims[0] != 1, s));
            MPN_TRY(mlp_tail(c, nullptr, nullptr, hidc, logits, 1, g.perm, E, s));
        }
    }
    if (x_out && N > 0) MPN_HIP(hipMemcpyAsync(x_out, p.x_hist + xs * prev, xs * sizeof(float), hipMemcpyDeviceToDevice, s));
    if (e_out) MPN_TRY(scatter_rows(p.e_hist + es * prev, g.perm, e_out, E, d.de, s));
    return MPNHIP_OK;
}

// ------------------------------------------------------------------------------------ MetaLayer op
static size_t plan_meta(const mpnhip_model& m, const Dims& d, int64_t N, int64_t E, void* base, float** Wnode,
                        float** bnode, StepBufs* sb) {
    Arena a = {static_cast<char*>(base), 0};
    float* w = a.f((size_t)d.pw * d.kx);
    float* b = a.f((size_t)d.pw);
    carve_step(a, m, d, N, E, false, false, sb);
    if (Wnode) *Wnode = w;
    if (bnode) *bnode = b;
    return a.off;
}

extern "C" size_t mpnhip_meta_layer_workspace_bytes(const mpnhip_model* model, int n_nodes, int64_t n_edges) {
    Dims d;
    if (!model || check_core(*model, &d, false) != MPNHIP_OK) return 0;
    return plan_meta(*model, d, n_nodes, n_edges, nullptr, nullptr, nullptr, nullptr);
}

extern "C" int mpnhip_meta_layer_forward(const mpnhip_model* model, const void* graph_buf, int n_nodes, int64_t n_edges,
                                         const float* x, const float* e, float* x_new, float* e_new, void* workspace,
                                         size_t workspace_bytes, void* stream_) {
    hipStream_t s = static_cast<hipStream_t>(stream_);
    MPN_CHECK_ARG(model && graph_buf, "meta_layer: null model / graph");
    Dims d;
    MPN_TRY(check_core(*model, &d));
    const int64_t N = n_nodes, E = n_edges;
    MPN_CHECK_ARG((x && x_new) || N == 0, "meta_layer: null node tensors");
    MPN_CHECK_ARG((e && e_new) || E == 0, "meta_layer: null edge tensors");
    float *Wnode, *bnode;
    StepBufs b;
    size_t need = plan_meta(*model, d, N, E, workspace, &Wnode, &bnode, &b);
    if (!workspace || workspace_bytes < need) {
        set_error("meta_layer: workspace %zu < %zu", workspace_bytes, need);
        return MPNHIP_ERR_WORKSPACE;
    }
    GraphView g;
    graph_layout(n_nodes, n_edges, &g, const_cast<void*>(graph_buf));
    MPN_TRY(pack_node_weights(*model, d, Wnode, bnode, s));
    StepIO io = {};
    io.xa = x; io.ldxa = d.kx;
    io.ea = e; io.ldea = d.ke; io.e_idx = g.perm;
    io.e_new = e_new; io.e_new_idx = g.perm; io.e_new_read_idx = g.perm;
    io.x_new = x_new;
    io.logits = nullptr;
    return run_step(*model, d, g, Wnode, bnode, io, b, false, s);
}

// ------------------------------------------------------------------------------------ Linear / MLP ops
extern "C" int mpnhip_linear(const float* x, int64_t ldx, const float* w, const float* b, float* y, int64_t ldy,
                             int64_t m, int n, int k, int relu, void* stream_) {
    MPN_CHECK_ARG(m >= 0 && n >= 1 && k >= 1, "linear: bad sizes");
    if (m == 0) return MPNHIP_OK;
    MPN_CHECK_ARG(x && w && y, "linear: null pointer");
    return linear(x, ldx, w, b, y, ldy, m, n, k, relu, static_cast<hipStream_t>(stream_));
}

extern "C" size_t mpnhip_mlp_workspace_bytes(const mpnhip_mlp* mlp, int64_t m) {
    if (!mlp) return 0;
    int h = max_hidden(*mlp);
    return 2 * align_up((size_t)(m > 0 ? m : 1) * (h > 0 ? h : 1) * sizeof(float), 256);
}

extern "C" int mpnhip_mlp_forward(const mpnhip_mlp* mlp, const float* x, float* y, int64_t m, void* workspace,
                                  size_t workspace_bytes, void* stream_) {
    MPN_CHECK_ARG(mlp, "mlp_forward: null mlp");
    MPN_TRY(mlp_ok(*mlp, "mlp", true));
    if (m == 0) return MPNHIP_OK;
    MPN_CHECK_ARG(x && y && m > 0, "mlp_forward: null pointer");
    size_t need = mpnhip_mlp_workspace_bytes(mlp, m);
    if (mlp->n_layers > 1 && (!workspace || workspace_bytes < need)) {
        set_error("mlp_forward: workspace %zu < %zu", workspace_bytes, need);
        return MPNHIP_ERR_WORKSPACE;
    }
    float* two[2] = {static_cast<float*>(workspace), reinterpret_cast<float*>(static_cast<char*>(workspace) + need / 2)};
    float* hid[MPNHIP_MAX_LAYERS];
    hidden_ptrs(*mlp, two, m, false, hid);
    return mlp_forward(*mlp, x, mlp->in_dim, nullptr, hid, y, m, static_cast<hipStream_t>(stream_));
}

extern "C" int mpnhip_avgpool(const float* x, int64_t rows, int hw, float* y, void* stream_) {
    MPN_CHECK_ARG(rows >= 0 && hw >= 1, "avgpool: bad sizes");
    if (rows == 0) return MPNHIP_OK;
    MPN_CHECK_ARG(x && y, "avgpool: null pointer");
    int sub = 1;
    while (sub < hw && sub < 64) sub <<= 1;
    int64_t threads = rows * sub;
    hipLaunchKernelGGL(k_avgpool, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream_),
                       x, rows, hw, y, sub);
    MPN_LAUNCH_CHECK();
    return MPNHIP_OK;
}

// ------------------------------------------------------------------------------------ test instrumentation
// argmax entries are positions in SORTED edge order (or -1): as floats holding the ORIGINAL edge id
__global__ void k_arg_to_original(const int* __restrict__ arg, const int* __restrict__ perm, float* __restrict__ out, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int j = arg[i];
    out[i] = j >= 0 ? (float)perm[j] : -1.f;
}

extern "C" int mpnhip_debug_saved(const mpnhip_model* model, const void* graph_buf, int n_nodes, int64_t n_edges,
                                  const void* fwd_workspace, size_t fwd_workspace_bytes, int what, int step, int layer, float* out,
                                  int64_t* rows_out, int* width_out, void* stream_) {
    hipStream_t s = static_cast<hipStream_t>(stream_);
    MPN_CHECK_ARG(model && graph_buf && fwd_workspace, "debug_saved: null argument");
    const mpnhip_model& m = *model;
    Dims d;
    MPN_TRY(check_full(m, &d));
    const int64_t N = n_nodes, E = n_edges;
    FwdPlan p;
    const size_t need = plan_forward(m, d, N, E, 1, const_cast<void*>(fwd_workspace), &p);
    if (fwd_workspace_bytes < need) {
        set_error("debug_saved: workspace %zu < %zu (must be a save_for_backward buffer)", fwd_workspace_bytes, need);
        return MPNHIP_ERR_WORKSPACE;
    }
    GraphView g;
    graph_layout(n_nodes, n_edges, &g, const_cast<void*>(graph_buf));
    const size_t xs = (size_t)N * d.dn, es = (size_t)E * d.de;
    const float* src = nullptr;
    int64_t rows = 0;
    int width = 0;
    bool per_edge = false;
    auto enc_layer = [&](const mpnhip_mlp& e, float* base, int64_t r) -> const float* {
        if (layer < 0 || layer + 1 >= e.n_layers) return nullptr;
        size_t off = 0;
        for (int i = 0; i < layer; ++i) off += (size_t)r * e.out_dims[i];
        width = e.out_dims[layer];
        return base + off;
    };
    const bool step_ok = step >= 1 && step <= d.L;
    const StepBufs b = step_ok ? step_at(p, step - 1) : p.step0;
    switch (what) {
        case MPNHIP_SAVED_ENC_NODE: src = enc_layer(m.enc_node, p.enc_n[0], N); rows = N; break;
        case MPNHIP_SAVED_ENC_EDGE: src = enc_layer(m.enc_edge, p.enc_e[0], E); rows = E; per_edge = true; break;
        case MPNHIP_SAVED_X: if (step >= 0 && step <= d.L) { src = p.x_hist + xs * step; rows = N; width = d.dn; } break;
        case MPNHIP_SAVED_E: if (step >= 0 && step <= d.L) { src = p.e_hist + es * step; rows = E; width = d.de; per_edge = true; } break;
        case MPNHIP_SAVED_EDGE_HIDDEN:
            if (step_ok && layer >= 0 && layer + 1 < m.edge.n_layers) { src = b.HE[layer]; rows = E; width = m.edge.out_dims[layer]; per_edge = true; }
            break;
        case MPNHIP_SAVED_CLS_HIDDEN:
            if (step_ok && layer >= 0 && layer + 1 < m.classifier.n_layers) { src = b.HC[layer]; rows = E; width = m.classifier.out_dims[layer]; per_edge = true; }
            break;
        case MPNHIP_SAVED_FLOW_HIDDEN:
            if (step_ok && layer >= 0 && layer + 1 < m.flow_in.n_layers) { src = b.HF[layer]; rows = E; width = m.flow_in.out_dims[layer]; per_edge = true; }
            break;
        case MPNHIP_SAVED_MSG: if (step_ok) { src = b.M; rows = E; width = d.dn; per_edge = true; } break;
        case MPNHIP_SAVED_AGG: if (step_ok) { src = b.AGG; rows = N; width = 2 * d.dn; } break;
        case MPNHIP_SAVED_ARGMAX: if (step_ok && b.ARG) { src = reinterpret_cast<const float*>(b.ARG); rows = N; width = 2 * d.dn; } break;
        default: break;
    }
    if (!src) {
        set_error("debug_saved: nothing saved for what = %d, step = %d, layer = %d", what, step, layer);
        return MPNHIP_ERR_ARG;
    }
    if (rows_out) *rows_out = rows;
    if (width_out) *width_out = width;
    if (!out || rows * width == 0) return MPNHIP_OK;
    if (p.b16 && step_ok && E > 0) {
        // bf16-operand training on the fused kernels: the hidden activations are bf16 rows (returned as floats), and for sum / mean
        // the messages themselves are never stored -- their ReLU decisions are (returned as 1.0 / 0.0: what a test reads them for)
        if (what == MPNHIP_SAVED_EDGE_HIDDEN || what == MPNHIP_SAVED_CLS_HIDDEN || what == MPNHIP_SAVED_FLOW_HIDDEN)
            return chain_bf16_debug_rows(reinterpret_cast<const unsigned short*>(src), g.perm, E, width, out, s);
        if (what == MPNHIP_SAVED_E && step >= 1 && step < d.L)   // (only the last step's fp32 features are written)
            return chain_bf16_debug_rows(p.eb_hist + es * step, g.perm, E, width, out, s);
        if (what == MPNHIP_SAVED_MSG && m.agg != MPNHIP_AGG_MAX)
            return chain_bf16_debug_mask(reinterpret_cast<const unsigned*>(b.MK), 4, g.header, g.perm, E, d.he, d.de, d.hn, d.dn,
                                         m.classifier.out_dims[0], out, s);
    }
    if (what == MPNHIP_SAVED_ARGMAX) {
        const int64_t n = rows * width;
        hipLaunchKernelGGL(k_arg_to_original, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, reinterpret_cast<const int*>(src), g.perm, out, n);
        MPN_LAUNCH_CHECK();
        return MPNHIP_OK;
    }
    if (per_edge) return scatter_rows(src, g.perm, out, rows, width, s);   // sorted -> original edge order
    MPN_HIP(hipMemcpyAsync(out, src, (size_t)rows * width * sizeof(float), hipMemcpyDeviceToDevice, s));
    return MPNHIP_OK;
}

extern "C" int mpnhip_edge_chain_active(const mpnhip_model* model) {
    Dims d;
    if (!model || check_full(*model, &d, false) != MPNHIP_OK) return 0;
    return chain_shapes_ok(*model, d) ? 1 : (chain_bf16_ok(*model, d) ? 2 : 0);
}

extern "C" int mpnhip_profile_enable(int on) {
    g_prof.on = on != 0;
    g_prof.stride = on > 1 ? on : 1;
    for (int k = 0; k < PROF_KINDS; ++k) g_prof.seen[k] = g_prof.n[k] = 0;
    g_prof.open_kind = -1;
    return MPNHIP_OK;
}

extern "C" int mpnhip_profile_read_kind(int kind, float* avg_us, int* launches, double* avg_work) {
    MPN_CHECK_ARG(kind >= 0 && kind < PROF_KINDS, "profile_read_kind: kind %d", kind);
    MPN_HIP(hipDeviceSynchronize());
    double tot = 0.0, work = 0.0;
    for (int i = 0; i < g_prof.n[kind]; ++i) {
        float ms = 0.f;
        MPN_HIP(hipEventElapsedTime(&ms, g_prof.ev[kind][i][0], g_prof.ev[kind][i][1]));
        tot += ms;
        work += g_prof.work[kind][i];
    }
    const int n = g_prof.n[kind];
    if (avg_us) *avg_us = n ? (float)(tot * 1000.0 / n) : 0.f;
    if (launches) *launches = n;
    if (avg_work) *avg_work = n ? work / n : 0.0;
    g_prof.n[kind] = 0;
    return MPNHIP_OK;
}

extern "C" int mpnhip_profile_read(float* gemm_avg_us, int* gemm_launches, float* agg_avg_us, int* agg_launches,
                                   float* empty_pair_us) {
    MPN_HIP(hipDeviceSynchronize());
    if (empty_pair_us) {
        // what a begin/end event pair costs with NO kernel between them (same stream, same API calls)
        *empty_pair_us = 0.f;
        if (g_prof.made[PROF_AGG] || g_prof.made[PROF_GEMM]) {
            const int k = g_prof.made[PROF_AGG] ? PROF_AGG : PROF_GEMM;
            const int reps = 64;
            for (int i = 0; i < reps; ++i) {
                MPN_HIP(hipEventRecord(g_prof.ev[k][PROF_MAX - 1 - i][0], 0));
                MPN_HIP(hipEventRecord(g_prof.ev[k][PROF_MAX - 1 - i][1], 0));
            }
            MPN_HIP(hipDeviceSynchronize());
            double tot = 0.0;
            for (int i = 0; i < reps; ++i) {
                float ms = 0.f;
                MPN_HIP(hipEventElapsedTime(&ms, g_prof.ev[k][PROF_MAX - 1 - i][0], g_prof.ev[k][PROF_MAX - 1 - i][1]));
                tot += ms;
            }
            *empty_pair_us = (float)(tot * 1000.0 / reps);
        }
    }
    float* outs[2] = {gemm_avg_us, agg_avg_us};
    int* cnts[2] = {gemm_launches, agg_launches};
    for (int k = 0; k < 2; ++k) {
        double tot = 0.0;
        for (int i = 0; i < g_prof.n[k]; ++i) {
            float ms = 0.f;
            MPN_HIP(hipEventElapsedTime(&ms, g_prof.ev[k][i][0], g_prof.ev[k][i][1]));
            tot += ms;
        }
        if (outs[k]) *outs[k] = g_prof.n[k] ? (float)(tot * 1000.0 / g_prof.n[k]) : 0.f;
        if (cnts[k]) *cnts[k] = g_prof.n[k];
        g_prof.n[k] = 0;
    }
    return MPNHIP_OK;
}

extern "C" int mpnhip_time_linear(const float* x, const float* w, const float* b, float* y, int64_t m, int n, int k,
                                  int iters, float* avg_us, void* stream_) {
    hipStream_t s = static_cast<hipStream_t>(stream_);
    MPN_CHECK_ARG(x && w && y && avg_us && iters > 0, "time_linear: bad argument");
    hipEvent_t t0, t1;
    MPN_HIP(hipEventCreate(&t0));
    MPN_HIP(hipEventCreate(&t1));
    MPN_TRY(linear(x, k, w, b, y, n, m, n, k, 1, s));
    MPN_HIP(hipEventRecord(t0, s));
    for (int i = 0; i < iters; ++i) MPN_TRY(linear(x, k, w, b, y, n, m, n, k, 1, s));
    MPN_HIP(hipEventRecord(t1, s));
    MPN_HIP(hipEventSynchronize(t1));
    float ms = 0.f;
    MPN_HIP(hipEventElapsedTime(&ms, t0, t1));
    *avg_us = ms * 1000.f / iters;
    (void)hipEventDestroy(t0);
    (void)hipEventDestroy(t1);
    return MPNHIP_OK;
}
