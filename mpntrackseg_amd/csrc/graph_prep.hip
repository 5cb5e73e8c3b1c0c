// Graph preparation: one stable sort of the edges by (direction, row) replaces the per-step boolean
// masks / nonzero() syncs of TimeAwareNodeModel.forward (reference models/mpn.py:85-87,91-93) and
// gives every scatter of the path (torch_scatter calls at mpn.py:266-273, index_put_ in autograd) a
// CSR form that needs no atomics and fixes the summation order to the reference CPU order
// (ascending edge id inside a segment).
#include <cstring>

#include "common.h"

#include <rocprim/device/device_radix_sort.hpp>

namespace mpnhip {

size_t graph_layout(int N, int64_t E, GraphView* v, void* base) {
    size_t off = 0;
    auto take = [&](size_t n_ints) {
        size_t o = off;
        off = align_up(off + n_ints * sizeof(int), 256);
        return o;
    };
    size_t o_header = take(8), o_perm = take(E), o_srow = take(E), o_scol = take(E), o_seg = take(3 * (size_t)N + 1);
    size_t o_cperm = take(E), o_cseg = take(3 * (size_t)N + 1), o_rperm = take(E), o_rseg = take((size_t)N + 1);
    size_t o_call = take(E), o_csall = take((size_t)N + 1);
    if (v) {
        char* b = static_cast<char*>(base);
        v->N = N;
        v->E = E;
        v->header = reinterpret_cast<int*>(b + o_header);
        v->perm = reinterpret_cast<int*>(b + o_perm);
        v->srow = reinterpret_cast<int*>(b + o_srow);
        v->scol = reinterpret_cast<int*>(b + o_scol);
        v->seg_ptr = reinterpret_cast<int*>(b + o_seg);
        v->cperm = reinterpret_cast<int*>(b + o_cperm);
        v->cseg_ptr = reinterpret_cast<int*>(b + o_cseg);
        v->rperm = reinterpret_cast<int*>(b + o_rperm);
        v->rseg_ptr = reinterpret_cast<int*>(b + o_rseg);
        v->cperm_all = reinterpret_cast<int*>(b + o_call);
        v->cseg_all = reinterpret_cast<int*>(b + o_csall);
    }
    return off;
}

__global__ void k_make_keys(const int64_t* __restrict__ ei, int64_t E, int N, unsigned* __restrict__ keys,
                            int* __restrict__ vals, int* __restrict__ header) {
    int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    int64_t r = ei[e], c = ei[E + e];
    if (r < 0 || r >= N || c < 0 || c >= N) {
        header[0] = 1;  // benign race: every writer stores the same value
        r = r < 0 ? 0 : (r >= N ? N - 1 : r);
        c = c < 0 ? 0 : (c >= N ? N - 1 : c);
    }
    unsigned dir = r < c ? 0u : (r > c ? 1u : 2u);
    keys[e] = dir * (unsigned)N + (unsigned)r;
    vals[e] = (int)e;
}

__global__ void k_gather_rc(const int64_t* __restrict__ ei, int64_t E, int N, const int* __restrict__ perm,
                            int* __restrict__ srow, int* __restrict__ scol) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= E) return;
    int64_t e = perm[i];
    int64_t r = ei[e], c = ei[E + e];
    r = r < 0 ? 0 : (r >= N ? N - 1 : r);
    c = c < 0 ? 0 : (c >= N ? N - 1 : c);
    srow[i] = (int)r;
    scol[i] = (int)c;
}

// ptr[k] = first position whose sorted key is >= k, k = 0..nkeys (ptr[nkeys] = E)
__global__ void k_lower_bound(const unsigned* __restrict__ skeys, int64_t E, int nkeys, int* __restrict__ ptr) {
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k > nkeys) return;
    int64_t lo = 0, hi = E;
    while (lo < hi) {
        int64_t mid = (lo + hi) >> 1;
        if (skeys[mid] < (unsigned)k) lo = mid + 1; else hi = mid;
    }
    ptr[k] = (int)lo;
}

__global__ void k_header(const int* __restrict__ seg_ptr, int N, int64_t E, int* __restrict__ header) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        int e_out = seg_ptr[N], e_in = seg_ptr[2 * N] - seg_ptr[N];
        header[1] = e_out;
        header[2] = e_in;
        header[3] = (int)E - e_out - e_in;
        header[4] = e_out;         // row_end of the flow_out group / row_begin of the flow_in group
        header[5] = e_out + e_in;  // row_end of the flow_in group
        header[6] = 0;
        header[7] = (int)E;
    }
}

// keys for the secondary (backward) orders, over SORTED positions
__global__ void k_keys2(const int* __restrict__ srow, const int* __restrict__ scol, int64_t E, int N, int mode,
                        unsigned* __restrict__ keys, int* __restrict__ vals) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= E) return;
    int r = srow[i], c = scol[i];
    unsigned dir = r < c ? 0u : (r > c ? 1u : 2u);
    unsigned k = mode == 0 ? dir * (unsigned)N + (unsigned)c : (mode == 1 ? (unsigned)r : (unsigned)c);
    keys[i] = k;
    vals[i] = (int)i;
}

static int bits_for(unsigned maxkey) {
    int b = 1;
    while (b < 32 && (maxkey >> b)) ++b;
    return b;
}

static size_t sort_temp_bytes(int64_t E) {
    size_t bytes = 0;
    unsigned* k = nullptr;
    int* v = nullptr;
    (void)rocprim::radix_sort_pairs(nullptr, bytes, k, k, v, v, (size_t)(E > 0 ? E : 1), 0, 32, (hipStream_t)0);
    return bytes;
}

}  // namespace mpnhip

using namespace mpnhip;

extern "C" size_t mpnhip_graph_bytes(int n_nodes, int64_t n_edges) {
    return graph_layout(n_nodes, n_edges, nullptr, nullptr);
}

extern "C" size_t mpnhip_graph_prep_workspace_bytes(int n_nodes, int64_t n_edges) {
    size_t e = (size_t)(n_edges > 0 ? n_edges : 1);
    return 3 * align_up(e * 4, 256) + align_up(sort_temp_bytes(n_edges), 256) + 256;
}

static int graph_prep_impl(const int64_t* edge_index, int n_nodes, int64_t n_edges, void* graph_buf, size_t graph_bytes,
                           void* workspace, size_t workspace_bytes, bool secondary, void* stream_);

extern "C" int mpnhip_graph_prep(const int64_t* edge_index, int n_nodes, int64_t n_edges, void* graph_buf,
                                 size_t graph_bytes, void* workspace, size_t workspace_bytes, void* stream_) {
    return graph_prep_impl(edge_index, n_nodes, n_edges, graph_buf, graph_bytes, workspace, workspace_bytes, true, stream_);
}

extern "C" int mpnhip_graph_prep_forward(const int64_t* edge_index, int n_nodes, int64_t n_edges, void* graph_buf,
                                         size_t graph_bytes, void* workspace, size_t workspace_bytes, void* stream_) {
    return graph_prep_impl(edge_index, n_nodes, n_edges, graph_buf, graph_bytes, workspace, workspace_bytes, false, stream_);
}

static int graph_prep_impl(const int64_t* edge_index, int n_nodes, int64_t n_edges, void* graph_buf, size_t graph_bytes,
                           void* workspace, size_t workspace_bytes, bool secondary, void* stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const int N = n_nodes;
    const int64_t E = n_edges;
    MPN_CHECK_ARG(N >= 0 && E >= 0, "graph_prep: negative sizes");
    MPN_CHECK_ARG((int64_t)3 * N + 1 < 2147483647LL && E < 2147483647LL, "graph_prep: graph too large for int32 indices");
    MPN_CHECK_ARG(graph_buf && (E == 0 || edge_index), "graph_prep: null pointer");
    if (graph_bytes < mpnhip_graph_bytes(N, E)) {
        set_error("graph_prep: graph buffer %zu < %zu", graph_bytes, mpnhip_graph_bytes(N, E));
        return MPNHIP_ERR_WORKSPACE;
    }
    if (workspace_bytes < mpnhip_graph_prep_workspace_bytes(N, E) || (!workspace && E > 0)) {
        set_error("graph_prep: workspace %zu < %zu", workspace_bytes, mpnhip_graph_prep_workspace_bytes(N, E));
        return MPNHIP_ERR_WORKSPACE;
    }
    GraphView g;
    graph_layout(N, E, &g, graph_buf);
    MPN_HIP(hipMemsetAsync(g.header, 0, 8 * sizeof(int), stream));
    const int T = 256;
    const unsigned nbE = (unsigned)((E + T - 1) / T);
    if (E == 0) {
        MPN_HIP(hipMemsetAsync(g.seg_ptr, 0, (3 * (size_t)N + 1) * sizeof(int), stream));
        MPN_HIP(hipMemsetAsync(g.cseg_ptr, 0, (3 * (size_t)N + 1) * sizeof(int), stream));
        MPN_HIP(hipMemsetAsync(g.rseg_ptr, 0, ((size_t)N + 1) * sizeof(int), stream));
        MPN_HIP(hipMemsetAsync(g.cseg_all, 0, ((size_t)N + 1) * sizeof(int), stream));
        return MPNHIP_OK;
    }
    char* ws = static_cast<char*>(workspace);
    size_t esz = align_up((size_t)E * 4, 256);
    unsigned* keys_in = reinterpret_cast<unsigned*>(ws);
    unsigned* keys_out = reinterpret_cast<unsigned*>(ws + esz);
    int* vals_in = reinterpret_cast<int*>(ws + 2 * esz);
    void* tmp = ws + 3 * esz;
    size_t tmp_bytes = workspace_bytes - 3 * esz;

    // primary order: (direction, row), stable
    hipLaunchKernelGGL(k_make_keys, dim3(nbE), dim3(T), 0, stream, edge_index, E, N, keys_in, vals_in, g.header);
    MPN_LAUNCH_CHECK();
    MPN_HIP(rocprim::radix_sort_pairs(tmp, tmp_bytes, keys_in, keys_out, vals_in, g.perm, (size_t)E, 0,
                                      bits_for(3u * (unsigned)N), stream));
    hipLaunchKernelGGL(k_gather_rc, dim3(nbE), dim3(T), 0, stream, edge_index, E, N, g.perm, g.srow, g.scol);
    MPN_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_lower_bound, dim3((3 * N + 1 + T) / T), dim3(T), 0, stream, keys_out, E, 3 * N, g.seg_ptr);
    MPN_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_header, dim3(1), dim3(64), 0, stream, g.seg_ptr, N, E, g.header);
    MPN_LAUNCH_CHECK();

    if (!secondary) return MPNHIP_OK;  // inference: mpnhip_forward reads the primary order only
    // secondary orders for the backward scatter-adds (index_put_ of x[row], x[col])
    struct { int mode; int* perm; int* ptr; int nkeys; } sec[3] = {
        {0, g.cperm, g.cseg_ptr, 3 * N}, {1, g.rperm, g.rseg_ptr, N}, {2, g.cperm_all, g.cseg_all, N}};
    for (auto& s : sec) {
        hipLaunchKernelGGL(k_keys2, dim3(nbE), dim3(T), 0, stream, g.srow, g.scol, E, N, s.mode, keys_in, vals_in);
        MPN_LAUNCH_CHECK();
        MPN_HIP(rocprim::radix_sort_pairs(tmp, tmp_bytes, keys_in, keys_out, vals_in, s.perm, (size_t)E, 0,
                                          bits_for((unsigned)s.nkeys), stream));
        hipLaunchKernelGGL(k_lower_bound, dim3((s.nkeys + 1 + T) / T), dim3(T), 0, stream, keys_out, E, s.nkeys, s.ptr);
        MPN_LAUNCH_CHECK();
    }
    return MPNHIP_OK;
}

extern "C" int mpnhip_graph_status(const void* graph_buf, int n_nodes, int64_t n_edges, int32_t status[4], void* stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    MPN_CHECK_ARG(graph_buf && status, "graph_status: null pointer");
    GraphView g;
    graph_layout(n_nodes, n_edges, &g, const_cast<void*>(graph_buf));
    MPN_HIP(hipMemcpyAsync(status, g.header, 4 * sizeof(int), hipMemcpyDeviceToHost, stream));
    MPN_HIP(hipStreamSynchronize(stream));
    return MPNHIP_OK;
}
