// Attention-weighted neighbour aggregation of the mask branch (reference models/mpn.py:117-134,
// TimeAwareAttentionModel.forward):
//     w      = scatter_softmax(logit[dir mask], row[dir mask])        per (node, direction) segment
//     out[n] = sum_{j in segment(n, dir)} w_j * x[col_j]              x: [N, F], F = C*H*W = 64*14*14 = 12,544
// The genuinely HBM-bound kernel of the model: every edge gathers a 50 KB feature row.  One block per
// (node, direction) segment of the sorted edge list: softmax of the segment's logits in LDS, then the block
// streams the neighbour rows (each a contiguous 50 KB read, 16 B per lane) and accumulates in registers in
// ascending edge order (the order of the reference's sequential CPU scatter_add).
// Backward: dx[c] += sum over edges with col == c of w_j * dout[row_j] (col-sorted CSR, no atomics) and
// dlogit via the softmax Jacobian with dw_j = <dout[row_j], x[col_j]>.
#include "common.h"

namespace mpnhip {

constexpr int AT = 256;

__device__ inline float block_reduce(float v, float* red, bool is_max) {
    red[threadIdx.x] = v;
    __syncthreads();
    for (int w = AT / 2; w > 0; w >>= 1) {
        if (threadIdx.x < w) red[threadIdx.x] = is_max ? fmaxf(red[threadIdx.x], red[threadIdx.x + w]) : red[threadIdx.x] + red[threadIdx.x + w];
        __syncthreads();
    }
    float r = red[0];
    __syncthreads();
    return r;
}

// grid: 2N blocks; key k = blockIdx.x: direction k / N (0 flow_out, 1 flow_in), node k % N
__global__ __launch_bounds__(AT) void k_attention_fwd(GraphView g, const float* __restrict__ x, int64_t F,
                                                      const float* __restrict__ logits, float* __restrict__ out_in,
                                                      float* __restrict__ out_out, float* __restrict__ wts, int cap) {
    __shared__ float red[AT];
    extern __shared__ float wseg[];  // softmax weights of this segment (dynamic: segment length floats)
    const int key = blockIdx.x, N = g.N;
    const int dir = key / N, n = key % N;
    const int beg = g.seg_ptr[key], end = g.seg_ptr[key + 1], len = end - beg;
    float* out = (dir == 0 ? out_out : out_in) + (int64_t)n * F;
    // ---- scatter_softmax (torch_scatter 2.0.4 composite): exp(l - max) / (sum + 1e-12)
    float mx = -INFINITY;
    for (int j = threadIdx.x; j < len; j += AT) mx = fmaxf(mx, logits[g.perm[beg + j]]);
    mx = block_reduce(mx, red, true);
    const bool fits = len <= cap;  // hub segments longer than the LDS buffer recompute their weights on the fly
    float sm = 0.f;
    for (int j = threadIdx.x; j < len; j += AT) sm += expf(logits[g.perm[beg + j]] - mx);
    sm = block_reduce(sm, red, false);
    const float inv_den = sm + 1e-12f;
    for (int j = threadIdx.x; j < len; j += AT) {
        const float w = expf(logits[g.perm[beg + j]] - mx) / inv_den;
        if (fits) wseg[j] = w;
        if (wts) wts[beg + j] = w;
    }
    __syncthreads();
    // ---- weighted sum of the neighbour rows, 16 bytes per lane, ascending edge order
    const int64_t F4 = F >> 2;
    for (int64_t c = threadIdx.x; c < F4; c += AT) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int j = 0; j < len; j += 4) {
            float4 v[4];
            float w[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int jj = j + u < len ? j + u : len - 1;
                v[u] = *reinterpret_cast<const float4*>(x + (int64_t)g.scol[beg + jj] * F + 4 * c);
                w[u] = j + u < len ? (fits ? wseg[jj] : expf(logits[g.perm[beg + jj]] - mx) / inv_den) : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                acc.x = fmaf(w[u], v[u].x, acc.x); acc.y = fmaf(w[u], v[u].y, acc.y);
                acc.z = fmaf(w[u], v[u].z, acc.z); acc.w = fmaf(w[u], v[u].w, acc.w);
            }
        }
        *reinterpret_cast<float4*>(out + 4 * c) = acc;
    }
}

// dw[j] = <dout[row_j], x[col_j]> for every sorted edge j of the two directions: one block per edge
__global__ __launch_bounds__(AT) void k_attention_dw(GraphView g, const float* __restrict__ x, int64_t F,
                                                     const float* __restrict__ d_in, const float* __restrict__ d_out,
                                                     float* __restrict__ dw) {
    __shared__ float red[AT];
    const int j = blockIdx.x;
    const int e_out = g.header[1], e_in = g.header[2];
    if (j >= e_out + e_in) return;
    const float* d = (j < e_out ? d_out : d_in) + (int64_t)g.srow[j] * F;
    const float* xr = x + (int64_t)g.scol[j] * F;
    float s = 0.f;
    for (int64_t c = threadIdx.x; c < (F >> 2); c += AT) {
        const float4 a = *reinterpret_cast<const float4*>(d + 4 * c), b = *reinterpret_cast<const float4*>(xr + 4 * c);
        s += a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w;
    }
    s = block_reduce(s, red, false);
    if (threadIdx.x == 0) dw[j] = s;
}

// softmax backward per segment: dlogit_j = w_j (dw_j - sum_k w_k dw_k), scattered to ORIGINAL edge order (+=)
__global__ __launch_bounds__(AT) void k_attention_dlogit(GraphView g, const float* __restrict__ wts, const float* __restrict__ dw,
                                                         float* __restrict__ dlogits) {
    __shared__ float red[AT];
    const int key = blockIdx.x;
    const int beg = g.seg_ptr[key], end = g.seg_ptr[key + 1];
    float s = 0.f;
    for (int j = beg + threadIdx.x; j < end; j += AT) s += wts[j] * dw[j];
    s = block_reduce(s, red, false);
    for (int j = beg + threadIdx.x; j < end; j += AT) dlogits[g.perm[j]] += wts[j] * (dw[j] - s);
}

// dx[c] (+)= sum over sorted edges q with col == c (both directions) of w_q * dout_dir(q)[row_q]; one block per node c
__global__ __launch_bounds__(AT) void k_attention_dx(GraphView g, int64_t F, const float* __restrict__ wts,
                                                     const float* __restrict__ d_in, const float* __restrict__ d_out,
                                                     float* __restrict__ dx, int accumulate) {
    const int c_node = blockIdx.x, N = g.N;
    const int e_out = g.header[1];
    float* o = dx + (int64_t)c_node * F;
    for (int64_t c = threadIdx.x; c < (F >> 2); c += AT) {
        float4 acc = accumulate ? *reinterpret_cast<const float4*>(o + 4 * c) : make_float4(0.f, 0.f, 0.f, 0.f);
        for (int dir = 0; dir < 2; ++dir) {
            const int beg = g.cseg_ptr[dir * N + c_node], end = g.cseg_ptr[dir * N + c_node + 1];
            for (int q = beg; q < end; ++q) {
                const int j = g.cperm[q];
                const float w = wts[j];
                const float* d = (j < e_out ? d_out : d_in) + (int64_t)g.srow[j] * F;
                const float4 v = *reinterpret_cast<const float4*>(d + 4 * c);
                acc.x = fmaf(w, v.x, acc.x); acc.y = fmaf(w, v.y, acc.y); acc.z = fmaf(w, v.z, acc.z); acc.w = fmaf(w, v.w, acc.w);
            }
        }
        *reinterpret_cast<float4*>(o + 4 * c) = acc;
    }
}

__global__ void k_max_seg_len(const int* __restrict__ seg_ptr, int nseg, int* __restrict__ out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nseg) atomicMax(out, seg_ptr[i + 1] - seg_ptr[i]);
}

}  // namespace mpnhip

using namespace mpnhip;

// The dynamic LDS holds one segment's weights; sized for the worst case the graph can have (<= E), capped at
// 60 KB (15,360 neighbours in one direction) -- larger segments are rejected.
static int attention_lds_bytes(int64_t n_edges) {
    int64_t b = n_edges * 4;
    return (int)(b < 60 * 1024 ? (b < 256 ? 256 : b) : 60 * 1024);
}

extern "C" int mpnhip_attention_aggregate(const void* graph_buf, int n_nodes, int64_t n_edges, const float* x, int64_t feat,
                                          const float* logits, float* out_in, float* out_out, float* weights, void* stream_) {
    hipStream_t s = static_cast<hipStream_t>(stream_);
    MPN_CHECK_ARG(graph_buf && (n_nodes == 0 || (x && out_in && out_out)), "attention_aggregate: null pointer");
    MPN_CHECK_ARG(feat > 0 && feat % 4 == 0, "attention_aggregate: feature size must be a multiple of 4");
    MPN_CHECK_ARG(n_edges == 0 || logits, "attention_aggregate: null logits");
    if (n_nodes == 0) return MPNHIP_OK;
    GraphView g;
    graph_layout(n_nodes, n_edges, &g, const_cast<void*>(graph_buf));
    const int lds = attention_lds_bytes(n_edges);
    hipLaunchKernelGGL(k_attention_fwd, dim3(2 * n_nodes), dim3(AT), lds, s, g, x, feat, logits, out_in, out_out, weights, lds / 4);
    MPN_LAUNCH_CHECK();
    return MPNHIP_OK;
}

extern "C" int mpnhip_attention_aggregate_backward(const void* graph_buf, int n_nodes, int64_t n_edges, const float* x,
                                                   int64_t feat, const float* weights, const float* grad_in,
                                                   const float* grad_out, float* grad_x, int accumulate_grad_x,
                                                   float* grad_logits, float* workspace_dw, void* stream_) {
    hipStream_t s = static_cast<hipStream_t>(stream_);
    MPN_CHECK_ARG(graph_buf, "attention_backward: null graph");
    MPN_CHECK_ARG(feat > 0 && feat % 4 == 0, "attention_backward: feature size must be a multiple of 4");
    if (n_nodes == 0) return MPNHIP_OK;
    MPN_CHECK_ARG(x && grad_in && grad_out, "attention_backward: null tensor");
    GraphView g;
    graph_layout(n_nodes, n_edges, &g, const_cast<void*>(graph_buf));
    if (grad_x) {
        hipLaunchKernelGGL(k_attention_dx, dim3(n_nodes), dim3(AT), 0, s, g, feat, weights, grad_in, grad_out, grad_x, accumulate_grad_x);
        MPN_LAUNCH_CHECK();
    }
    if (grad_logits && n_edges > 0) {
        MPN_CHECK_ARG(weights && workspace_dw, "attention_backward: null weights / workspace");
        hipLaunchKernelGGL(k_attention_dw, dim3((unsigned)n_edges), dim3(AT), 0, s, g, x, feat, grad_in, grad_out, workspace_dw);
        MPN_LAUNCH_CHECK();
        hipLaunchKernelGGL(k_attention_dlogit, dim3(2 * n_nodes), dim3(AT), 0, s, g, weights, workspace_dw, grad_logits);
        MPN_LAUNCH_CHECK();
    }
    return MPNHIP_OK;
}
