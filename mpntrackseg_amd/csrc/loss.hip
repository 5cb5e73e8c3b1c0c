// The two steps right after the hot path in the reference's training loop, on device
// and without host synchronisation:
//   * tracking loss of pl_module.py:88-107: sum over the classified steps of
//     F.binary_cross_entropy_with_logits(logits_s, edge_labels, pos_weight = #neg / #pos) * loss_weight,
//     together with d loss / d logits -- the seed of mpnhip_backward;
//   * compute_perform_metrics (utils/evaluation.py:416-437): confusion counts of (logit > 0) vs labels
//     (fast_compute_class_metric, :340-366) and the flow-conservation violation counts
//     (compute_constr_satisfaction_rate, :370-414) as segmented sums over the prepared graph's CSR lists
//     (the reference sorts every edge's endpoints and scatter-adds).
#include "common.h"

namespace mpnhip {

__global__ __launch_bounds__(1024) void k_count_pos(const float* __restrict__ labels, int64_t E, float* __restrict__ out) {
    __shared__ float red[1024];
    // (labels are 0 / 1: the sum is an exact integer in fp32 whatever the order; four independent 16-byte loads in flight per
    // lane instead of a chain of 4-byte ones: 23 -> 6 us at 50k edges)
    float s = 0.f;
    if ((reinterpret_cast<uintptr_t>(labels) & 15) == 0) {
        const int64_t n4 = E >> 2;
        const float4* l4 = reinterpret_cast<const float4*>(labels);
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        int64_t i = threadIdx.x;
        for (; i + 3 * 1024 < n4; i += 4 * 1024) {
            const float4 v0 = l4[i], v1 = l4[i + 1024], v2 = l4[i + 2048], v3 = l4[i + 3072];
            a0 += (v0.x + v0.y) + (v0.z + v0.w); a1 += (v1.x + v1.y) + (v1.z + v1.w);
            a2 += (v2.x + v2.y) + (v2.z + v2.w); a3 += (v3.x + v3.y) + (v3.z + v3.w);
        }
        for (; i < n4; i += 1024) { const float4 v = l4[i]; a0 += (v.x + v.y) + (v.z + v.w); }
        s = (a0 + a1) + (a2 + a3);
        for (int64_t j = (n4 << 2) + threadIdx.x; j < E; j += 1024) s += labels[j];
    } else {
        for (int64_t i = threadIdx.x; i < E; i += 1024) s += labels[i];
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int w = 512; w > 0; w >>= 1) {
        if (threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = red[0];
}

// grid (ceil(E / 256), L)
__global__ __launch_bounds__(256) void k_bce(const float* __restrict__ logits, const float* __restrict__ labels, int64_t E,
                                             int first_step, float weight, const float* __restrict__ pos_count,
                                             float* __restrict__ dlogits, float* __restrict__ partial) {
    const int step = blockIdx.y;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const float P = pos_count[0];
    const float pw = P > 0.f ? ((float)E - P) / P : 0.f;  // pl_module.py:92-96
    float term = 0.f;
    if (i < E) {
        float g = 0.f;
        if (step >= first_step) {
            const float z = logits[(int64_t)step * E + i], y = labels[i];
            const float lw = 1.f + (pw - 1.f) * y;
            // aten's stable form: (1 - y) z + lw (log1p(exp(-|z|)) + max(-z, 0))
            term = (1.f - y) * z + lw * (log1pf(expf(-fabsf(z))) + fmaxf(-z, 0.f));
            const float sg = z >= 0.f ? 1.f / (1.f + expf(-z)) : expf(z) / (1.f + expf(z));
            g = (sg * lw - pw * y) * (weight / (float)E);
        }
        dlogits[(int64_t)step * E + i] = g;
    }
    __shared__ float red[256];
    red[threadIdx.x] = term;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if (threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[(int64_t)step * gridDim.x + blockIdx.x] = red[0];
}

// one block; loss_out[0] = total, loss_out[1 + s] = weighted mean BCE of step s
__global__ __launch_bounds__(256) void k_loss_reduce(const float* __restrict__ partial, int nblk, int L, int64_t E, float weight,
                                                     float* __restrict__ loss_out) {
    __shared__ double red[256];
    double total = 0.0;
    for (int s = 0; s < L; ++s) {
        double acc = 0.0;
        for (int b = threadIdx.x; b < nblk; b += 256) acc += (double)partial[(int64_t)s * nblk + b];
        red[threadIdx.x] = acc;
        __syncthreads();
        for (int w = 128; w > 0; w >>= 1) {
            if (threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
            __syncthreads();
        }
        const double ls = red[0] * (double)weight / (double)(E > 0 ? E : 1);
        if (threadIdx.x == 0) loss_out[1 + s] = (float)ls;
        total += ls;
        __syncthreads();
    }
    if (threadIdx.x == 0) loss_out[0] = (float)total;
}

// ---- the same loss over K graphs of one block-diagonal batch: each graph its own pos_weight and its own mean, the K losses averaged
// (accumulate_grad_batches executed in space: configs/tracking_cfg.yaml:3-4, pl_module.py:88-107 per graph) ------------------------
// counts[g] = {edges, positive labels} of graph g (exact integers); K <= 1024
__global__ __launch_bounds__(256) void k_graph_counts(const float* __restrict__ labels, const int* __restrict__ edge_graph, int64_t E, int K,
                                                      int* __restrict__ counts) {
    extern __shared__ int sc[];   // [2 K]
    for (int i = threadIdx.x; i < 2 * K; i += 256) sc[i] = 0;
    __syncthreads();
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < E) {
        const int g = edge_graph[i];
        if (g >= 0 && g < K) {
            atomicAdd(&sc[2 * g], 1);
            if (labels[i] == 1.f) atomicAdd(&sc[2 * g + 1], 1);
        }
    }
    __syncthreads();
    for (int j = threadIdx.x; j < 2 * K; j += 256)
        if (sc[j]) atomicAdd(&counts[j], sc[j]);
}

// grid (ceil(E / 256), L): the per-edge term divided by its graph's edge count; partial[step][block] = sum of those
__global__ __launch_bounds__(256) void k_bce_graphs(const float* __restrict__ logits, const float* __restrict__ labels,
                                                    const int* __restrict__ edge_graph, int64_t E, int K, int first_step, float weight,
                                                    const int* __restrict__ counts, float* __restrict__ dlogits, float* __restrict__ partial) {
    const int step = blockIdx.y;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    float term = 0.f;
    if (i < E) {
        float gr = 0.f;
        const int g = edge_graph[i];
        if (step >= first_step && g >= 0 && g < K) {
            const float Eg = (float)counts[2 * g], P = (float)counts[2 * g + 1];
            const float pw = P > 0.f ? (Eg - P) / P : 0.f;  // pl_module.py:92-96, per graph
            const float z = logits[(int64_t)step * E + i], y = labels[i];
            const float lw = 1.f + (pw - 1.f) * y;
            term = ((1.f - y) * z + lw * (log1pf(expf(-fabsf(z))) + fmaxf(-z, 0.f))) / Eg;
            const float sg = z >= 0.f ? 1.f / (1.f + expf(-z)) : expf(z) / (1.f + expf(z));
            gr = (sg * lw - pw * y) * (weight / (Eg * (float)K));
        }
        dlogits[(int64_t)step * E + i] = gr;
    }
    __shared__ float red[256];
    red[threadIdx.x] = term;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if (threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[(int64_t)step * gridDim.x + blockIdx.x] = red[0];
}

__global__ void k_confusion(const float* __restrict__ logits, const float* __restrict__ labels, int64_t E, int* __restrict__ out) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= E) return;
    const int pred = logits[i] > 0.f, y = labels[i] == 1.f, y0 = labels[i] == 0.f;
    if (y && pred) atomicAdd(out + 0, 1);        // TP
    else if (y0 && pred) atomicAdd(out + 1, 1);  // FP
    else if (y0 && !pred) atomicAdd(out + 2, 1); // TN
    else if (y && !pred) atomicAdd(out + 3, 1);  // FN
}

// per node: flow_out = sum of predictions over edges whose SMALLER endpoint is n, flow_in: LARGER endpoint
__global__ void k_flow_constraints(GraphView g, const float* __restrict__ logits, int* __restrict__ out) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= g.N) return;
    const int N = g.N;
    auto seg_sum = [&](const int* ptr, const int* list, int key, int& cnt) {
        float s = 0.f;
        const int b = ptr[key], e = ptr[key + 1];
        for (int j = b; j < e; ++j) {
            const int pos = list ? list[j] : j;
            s += logits[g.perm[pos]] > 0.f ? 1.f : 0.f;
        }
        cnt += e - b;
        return s;
    };
    int c_out = 0, c_in = 0;
    float self_cnt;
    int c_self = 0;
    self_cnt = seg_sum(g.seg_ptr, nullptr, 2 * N + n, c_self);
    const float f_out = (seg_sum(g.seg_ptr, nullptr, n, c_out) + seg_sum(g.cseg_ptr, g.cperm, N + n, c_out) + self_cnt) * 0.5f;
    const float f_in = (seg_sum(g.seg_ptr, nullptr, N + n, c_in) + seg_sum(g.cseg_ptr, g.cperm, n, c_in) + self_cnt) * 0.5f;
    c_out += c_self;
    c_in += c_self;
    if (f_out > 1.f) atomicAdd(out + 4, 1);
    if (f_in > 1.f) atomicAdd(out + 5, 1);
    if (c_out > 0) atomicAdd(out + 6, 1);
    if (c_in > 0) atomicAdd(out + 7, 1);
}

}  // namespace mpnhip

using namespace mpnhip;

extern "C" size_t mpnhip_tracking_loss_workspace_bytes(int n_steps, int64_t n_edges) {
    const size_t nblk = (size_t)((n_edges + 255) / 256);
    return 256 + align_up((size_t)(n_steps > 0 ? n_steps : 1) * (nblk > 0 ? nblk : 1) * sizeof(float), 256);
}

extern "C" int mpnhip_tracking_loss(const float* logits, const float* labels, int n_steps, int64_t n_edges, int first_step,
                                    float weight, float* loss_out, float* grad_logits, void* workspace, size_t workspace_bytes,
                                    void* stream_) {
    hipStream_t s = static_cast<hipStream_t>(stream_);
    MPN_CHECK_ARG(n_steps >= 1 && n_edges >= 0 && first_step >= 0, "tracking_loss: bad sizes");
    MPN_CHECK_ARG(loss_out, "tracking_loss: null loss_out");
    if (n_edges == 0) {
        MPN_HIP(hipMemsetAsync(loss_out, 0, (size_t)(1 + n_steps) * sizeof(float), s));
        return MPNHIP_OK;
    }
    MPN_CHECK_ARG(logits && labels && grad_logits, "tracking_loss: null tensor");
    if (!workspace || workspace_bytes < mpnhip_tracking_loss_workspace_bytes(n_steps, n_edges)) {
        set_error("tracking_loss: workspace %zu < %zu", workspace_bytes, mpnhip_tracking_loss_workspace_bytes(n_steps, n_edges));
        return MPNHIP_ERR_WORKSPACE;
    }
    float* pos = static_cast<float*>(workspace);
    float* partial = reinterpret_cast<float*>(static_cast<char*>(workspace) + 256);
    const int nblk = (int)((n_edges + 255) / 256);
    hipLaunchKernelGGL(k_count_pos, dim3(1), dim3(1024), 0, s, labels, n_edges, pos);
    MPN_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_bce, dim3(nblk, n_steps), dim3(256), 0, s, logits, labels, n_edges, first_step, weight, pos, grad_logits, partial);
    MPN_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_loss_reduce, dim3(1), dim3(256), 0, s, partial, nblk, n_steps, n_edges, weight, loss_out);
    MPN_LAUNCH_CHECK();
    return MPNHIP_OK;
}

extern "C" size_t mpnhip_tracking_loss_graphs_workspace_bytes(int n_steps, int64_t n_edges, int n_graphs) {
    return mpnhip_tracking_loss_workspace_bytes(n_steps, n_edges) + align_up((size_t)2 * (n_graphs > 0 ? n_graphs : 1) * sizeof(int), 256);
}

extern "C" int mpnhip_tracking_loss_graphs(const float* logits, const float* labels, const int32_t* edge_graph, int n_graphs, int n_steps,
                                           int64_t n_edges, int first_step, float weight, float* loss_out, float* grad_logits,
                                           void* workspace, size_t workspace_bytes, void* stream_) {
    hipStream_t s = static_cast<hipStream_t>(stream_);
    MPN_CHECK_ARG(n_steps >= 1 && n_edges >= 0 && first_step >= 0 && n_graphs >= 1 && n_graphs <= 1024, "tracking_loss_graphs: bad sizes");
    MPN_CHECK_ARG(loss_out, "tracking_loss_graphs: null loss_out");
    if (n_edges == 0) {
        MPN_HIP(hipMemsetAsync(loss_out, 0, (size_t)(1 + n_steps) * sizeof(float), s));
        return MPNHIP_OK;
    }
    MPN_CHECK_ARG(logits && labels && grad_logits && edge_graph, "tracking_loss_graphs: null tensor");
    const size_t need = mpnhip_tracking_loss_graphs_workspace_bytes(n_steps, n_edges, n_graphs);
    if (!workspace || workspace_bytes < need) {
        set_error("tracking_loss_graphs: workspace %zu < %zu", workspace_bytes, need);
        return MPNHIP_ERR_WORKSPACE;
    }
    const size_t base = mpnhip_tracking_loss_workspace_bytes(n_steps, n_edges);
    float* partial = reinterpret_cast<float*>(static_cast<char*>(workspace) + 256);
    int* counts = reinterpret_cast<int*>(static_cast<char*>(workspace) + base);
    const int nblk = (int)((n_edges + 255) / 256);
    MPN_HIP(hipMemsetAsync(counts, 0, (size_t)2 * n_graphs * sizeof(int), s));
    hipLaunchKernelGGL(k_graph_counts, dim3(nblk), dim3(256), (size_t)2 * n_graphs * sizeof(int), s, labels, edge_graph, n_edges, n_graphs, counts);
    MPN_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_bce_graphs, dim3(nblk, n_steps), dim3(256), 0, s, logits, labels, edge_graph, n_edges, n_graphs, first_step, weight, counts,
                       grad_logits, partial);
    MPN_LAUNCH_CHECK();
    // (the per-edge terms were divided by their graph's edge count: E = 1 here; the mean over the graphs goes into the weight)
    hipLaunchKernelGGL(k_loss_reduce, dim3(1), dim3(256), 0, s, partial, nblk, n_steps, (int64_t)1, weight / (float)n_graphs, loss_out);
    MPN_LAUNCH_CHECK();
    return MPNHIP_OK;
}

extern "C" int mpnhip_step_metrics(const void* graph_buf, int n_nodes, int64_t n_edges, const float* logits, const float* labels,
                                   int32_t* counts, void* stream_) {
    hipStream_t s = static_cast<hipStream_t>(stream_);
    MPN_CHECK_ARG(graph_buf && counts, "step_metrics: null pointer");
    MPN_HIP(hipMemsetAsync(counts, 0, 8 * sizeof(int32_t), s));
    if (n_edges == 0) return MPNHIP_OK;
    MPN_CHECK_ARG(logits && labels, "step_metrics: null tensor");
    GraphView g;
    graph_layout(n_nodes, n_edges, &g, const_cast<void*>(graph_buf));
    hipLaunchKernelGGL(k_confusion, dim3((unsigned)((n_edges + 255) / 256)), dim3(256), 0, s, logits, labels, n_edges, counts);
    MPN_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_flow_constraints, dim3((n_nodes + 255) / 256), dim3(256), 0, s, g, logits, counts);
    MPN_LAUNCH_CHECK();
    return MPNHIP_OK;
}

// ---- optimizer step of the training loop (pl_module.py:76-77: torch.optim.Adam(lr, weight_decay)) over flat buffers ----
namespace mpnhip {
namespace {
__global__ void k_adam(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, int64_t n,
                       float lr, float b1, float b2, float eps, float wd, float bc1, float bc2_sqrt, const float* __restrict__ skip,
                       int calls, int* __restrict__ skipped) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (skip && *skip != 0.f) {   // (a rank of the job reported an invalid graph: nobody steps -- mpnhip_adam_step_guarded)
        // the skipped call does not count as an optimizer step: remembered on the device (no other thread of this launch reads
        // the counter on this branch, and no thread writes it on the other one)
        if (skipped && i == 0) skipped[0] += 1;
        return;
    }
    if (i >= n) return;
    if (skipped) {
        // bias corrections of the t-th APPLIED update, t = calls so far - skipped ones (torch.optim.Adam counts applied steps)
        const float t = (float)(calls - skipped[0]);
        bc1 = 1.f - powf(b1, t);
        bc2_sqrt = sqrtf(1.f - powf(b2, t));
    }
    // torch.optim.Adam (not AdamW): L2 decay joins the gradient; m, v exponential averages; bias corrections
    const float pi = p[i];
    const float gi = g[i] + wd * pi;
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[i] = pi - (lr / bc1) * (mi / denom);
}
}  // namespace
}  // namespace mpnhip

extern "C" int mpnhip_adam_step_counted(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, float lr,
                                        float beta1, float beta2, float eps, float weight_decay, int calls, const float* skip_flag,
                                        int* skipped_calls, void* stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    MPN_CHECK_ARG(n >= 0 && calls >= 1, "adam_step: bad size / step");
    if (n == 0) return MPNHIP_OK;
    MPN_CHECK_ARG(params && grads && exp_avg && exp_avg_sq, "adam_step: null pointer");
    const float bc1 = 1.f - powf(beta1, (float)calls);
    const float bc2_sqrt = sqrtf(1.f - powf(beta2, (float)calls));
    hipLaunchKernelGGL(mpnhip::k_adam, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, params, grads, exp_avg, exp_avg_sq, n,
                       lr, beta1, beta2, eps, weight_decay, bc1, bc2_sqrt, skip_flag, calls, skipped_calls);
    MPN_LAUNCH_CHECK();
    return MPNHIP_OK;
}

extern "C" int mpnhip_adam_step_guarded(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, float lr,
                                        float beta1, float beta2, float eps, float weight_decay, int step, const float* skip_flag,
                                        void* stream_) {
    return mpnhip_adam_step_counted(params, grads, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay, step, skip_flag, nullptr,
                                    stream_);
}

extern "C" int mpnhip_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, float lr,
                                float beta1, float beta2, float eps, float weight_decay, int step, void* stream_) {
    return mpnhip_adam_step_counted(params, grads, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay, step, nullptr, nullptr,
                                    stream_);
}
