// Sliding-window inference support:
//   get_knn_mask                  reference utils/graph.py:40-87      (reciprocal) top-k pruning of a window's edges
//   _evaluate_graph_in_batches    reference tracker/mpn_tracker.py:143-210   window selection, accumulation, averaging
// The reference fills a dense N x N distance matrix with inf, argsorts every row and reads ranks back through a second
// N x N matrix.  Here the (directed) entries are radix-sorted by (row, distance, col) -- the rank of an entry is its
// position inside its row's run -- and the transpose needed by the reciprocal test is a binary search in the entries
// sorted by (row, col).  Ties in distance resolve by column index (what a stable argsort gives); the reference's
// torch.argsort leaves them unspecified.
#include "common.h"

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_select.hpp>
#include <rocprim/iterator/counting_iterator.hpp>

namespace mpnhip {
namespace {

__device__ __forceinline__ unsigned ordered_bits(float d) {
    const unsigned u = __float_as_uint(d);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);  // ascending unsigned order == ascending float order
}

// entry m < E: (row[m], col[m]); entry E + m (only when the list holds one direction per pair): (col[m], row[m])
__device__ __forceinline__ void entry_rc(const int64_t* ei, int64_t E, int64_t m, unsigned& r, unsigned& c) {
    if (m < E) { r = (unsigned)ei[m]; c = (unsigned)ei[E + m]; }
    else { r = (unsigned)ei[E + (m - E)]; c = (unsigned)ei[m - E]; }
}

__global__ void k_knn_keys_rc(const int64_t* __restrict__ ei, int64_t E, int64_t M, unsigned long long* __restrict__ keys,
                              int* __restrict__ vals) {
    const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= M) return;
    unsigned r, c;
    entry_rc(ei, E, m, r, c);
    keys[m] = ((unsigned long long)r << 32) | c;
    vals[m] = (int)m;
}

__global__ void k_knn_keys_rd(const int64_t* __restrict__ ei, const float* __restrict__ dist, int64_t E, int64_t M,
                              const int* __restrict__ order, unsigned long long* __restrict__ keys) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M) return;
    const int64_t m = order[i];
    unsigned r, c;
    entry_rc(ei, E, m, r, c);
    keys[i] = ((unsigned long long)r << 32) | ordered_bits(dist[m < E ? m : m - E]);
}

// in_k[entry] = rank of the entry inside its row's run < top_k  (ranking_mat < top_k_nns, graph.py:72)
__global__ void k_knn_rank(const unsigned long long* __restrict__ skeys, const int* __restrict__ svals, int64_t M, int top_k,
                           unsigned char* __restrict__ in_k) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M) return;
    const unsigned long long row_key = skeys[i] & 0xFFFFFFFF00000000ull;
    int64_t lo = 0, hi = i;  // first position of this row's run
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (skeys[mid] < row_key) lo = mid + 1; else hi = mid;
    }
    in_k[svals[i]] = (i - lo) < top_k ? 1 : 0;
}

// pruned_mask = (in_k op in_k^T)[row, col]  (graph.py:73-85)
__global__ void k_knn_mask(const int64_t* __restrict__ ei, int64_t E, int64_t M, int symmetric, int reciprocal,
                           const unsigned long long* __restrict__ rc_keys, const int* __restrict__ rc_vals,
                           const unsigned char* __restrict__ in_k, unsigned char* __restrict__ mask) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    const bool fwd = in_k[e] != 0;
    bool rev = false;
    if (!symmetric) {
        rev = in_k[E + e] != 0;
    } else {
        const unsigned long long key = ((unsigned long long)(unsigned)ei[E + e] << 32) | (unsigned)ei[e];  // (col, row)
        int64_t lo = 0, hi = M;
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if (rc_keys[mid] < key) lo = mid + 1; else hi = mid;
        }
        // (a pair listed in one direction only has an inf entry on the other side; its rank among the infs is
        // unspecified in the reference -- treated as not a neighbour)
        rev = lo < M && rc_keys[lo] == key && in_k[rc_vals[lo]] != 0;
    }
    mask[e] = (reciprocal ? (fwd && rev) : (fwd || rev)) ? 1 : 0;
}

// edges of the full sequence graph whose two detections lie inside the window's node range (mpn_tracker.py:171-173;
// detections are ordered by frame, so a window of frames is a node range [n0, n1))
__global__ void k_window_flags(const int64_t* __restrict__ ei, int64_t E, int64_t n0, int64_t n1, unsigned char* __restrict__ flags) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    const int64_t r = ei[e], c = ei[E + e];
    flags[e] = (r >= n0 && r < n1 && c >= n0 && c < n1) ? 1 : 0;
}

struct FlagSet {
    const unsigned char* flags;
    __device__ bool operator()(const int& i) const { return flags[i] != 0; }
};

__global__ void k_gather_rows(const float* __restrict__ src, int64_t ld, const int* __restrict__ ids, int64_t n, int dim,
                              float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * dim) return;
    const int64_t r = i / dim;
    const int c = (int)(i - r * dim);
    out[i] = src[(int64_t)ids[r] * ld + c];
}

__global__ void k_gather_edges(const int64_t* __restrict__ ei, int64_t E, const int* __restrict__ ids, int64_t n, int64_t node0,
                               int64_t* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int64_t e = ids[i];
    out[i] = ei[e] - node0;
    out[n + i] = ei[E + e] - node0;
}

// overall_edge_preds[edges_mask][knn_mask] += sigmoid(logit);  overall_num_preds[...] += 1  (mpn_tracker.py:126-141,188-190)
__global__ void k_accumulate(const float* __restrict__ logits, const int* __restrict__ kept_ids, int64_t n_kept,
                             const int* __restrict__ win_ids, int count_kept, float* __restrict__ preds, float* __restrict__ num) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_kept) return;
    const int e = win_ids[kept_ids ? kept_ids[i] : (int)i];
    preds[e] += 1.f / (1.f + expf(-logits[i]));
    if (count_kept) num[e] += 1.f;
}
__global__ void k_count_window(const int* __restrict__ win_ids, int64_t n_win, float* __restrict__ num) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_win) return;
    num[win_ids[i]] += 1.f;
}
__global__ void k_average(const float* __restrict__ preds, const float* __restrict__ num, int64_t n, float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float v = preds[i] / num[i];
    out[i] = v != v ? 0.f : v;  // final_edge_preds[isnan] = 0  (mpn_tracker.py:197)
}

static size_t sort64_temp(int64_t M) {
    size_t bytes = 0;
    unsigned long long* k = nullptr;
    int* v = nullptr;
    (void)rocprim::radix_sort_pairs(nullptr, bytes, k, k, v, v, (size_t)(M > 0 ? M : 1), 0, 64, (hipStream_t)0);
    return bytes;
}
static size_t select_temp(int64_t n) {
    size_t bytes = 0;
    int* out = nullptr;
    FlagSet pred{nullptr};
    (void)rocprim::select(nullptr, bytes, rocprim::counting_iterator<int>(0), out, out, (size_t)(n > 0 ? n : 1), pred, (hipStream_t)0);
    return bytes;
}

}  // namespace
}  // namespace mpnhip

using namespace mpnhip;

extern "C" size_t mpnhip_knn_mask_workspace_bytes(int64_t n_edges, int symmetric_edges) {
    const size_t M = (size_t)(symmetric_edges ? n_edges : 2 * n_edges) + 1;
    return 4 * align_up(M * 8, 256) + 3 * align_up(M * 4, 256) + align_up(M, 256) + align_up(sort64_temp((int64_t)M), 256) + 256;
}

extern "C" int mpnhip_knn_mask(const float* pwise_dist, const int64_t* edge_ixs, int n_nodes, int64_t n_edges, int top_k_nns,
                               int reciprocal_k_nns, int symmetric_edges, unsigned char* pruned_mask, void* workspace,
                               size_t workspace_bytes, void* stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    MPN_CHECK_ARG(n_nodes >= 0 && n_edges >= 0 && n_edges < (1LL << 30), "knn_mask: bad sizes");
    MPN_CHECK_ARG(top_k_nns >= 0, "knn_mask: negative top_k_nns");
    if (n_edges == 0) return MPNHIP_OK;
    MPN_CHECK_ARG(pwise_dist && edge_ixs && pruned_mask, "knn_mask: null pointer");
    if (!workspace || workspace_bytes < mpnhip_knn_mask_workspace_bytes(n_edges, symmetric_edges)) {
        set_error("knn_mask: workspace %zu < %zu", workspace_bytes, mpnhip_knn_mask_workspace_bytes(n_edges, symmetric_edges));
        return MPNHIP_ERR_WORKSPACE;
    }
    const int64_t E = n_edges, M = symmetric_edges ? E : 2 * E;
    char* w = static_cast<char*>(workspace);
    auto take = [&](size_t bytes) { char* p = w; w += align_up(bytes, 256); return p; };
    auto* k_a = reinterpret_cast<unsigned long long*>(take((size_t)(M + 1) * 8));
    auto* rc_keys = reinterpret_cast<unsigned long long*>(take((size_t)(M + 1) * 8));
    auto* k_b = reinterpret_cast<unsigned long long*>(take((size_t)(M + 1) * 8));
    auto* rd_keys = reinterpret_cast<unsigned long long*>(take((size_t)(M + 1) * 8));
    int* v_a = reinterpret_cast<int*>(take((size_t)(M + 1) * 4));
    int* rc_vals = reinterpret_cast<int*>(take((size_t)(M + 1) * 4));
    int* rd_vals = reinterpret_cast<int*>(take((size_t)(M + 1) * 4));
    auto* in_k = reinterpret_cast<unsigned char*>(take((size_t)(M + 1)));
    void* tmp = w;
    size_t tmp_bytes = sort64_temp(M + 1);
    const unsigned blocks = (unsigned)((M + 255) / 256);
    // (1) entries sorted by (row, col): the transpose lookup table, and the tie order of (2)
    hipLaunchKernelGGL(k_knn_keys_rc, dim3(blocks), dim3(256), 0, stream, edge_ixs, E, M, k_a, v_a);
    MPN_LAUNCH_CHECK();
    MPN_HIP(rocprim::radix_sort_pairs(tmp, tmp_bytes, k_a, rc_keys, v_a, rc_vals, (size_t)M, 0, 64, stream));
    // (2) stable sort of that order by (row, distance): position inside the row's run = the entry's rank
    hipLaunchKernelGGL(k_knn_keys_rd, dim3(blocks), dim3(256), 0, stream, edge_ixs, pwise_dist, E, M, rc_vals, k_b);
    MPN_LAUNCH_CHECK();
    MPN_HIP(rocprim::radix_sort_pairs(tmp, tmp_bytes, k_b, rd_keys, rc_vals, rd_vals, (size_t)M, 0, 64, stream));
    hipLaunchKernelGGL(k_knn_rank, dim3(blocks), dim3(256), 0, stream, rd_keys, rd_vals, M, top_k_nns, in_k);
    MPN_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_knn_mask, dim3((unsigned)((E + 255) / 256)), dim3(256), 0, stream, edge_ixs, E, M, symmetric_edges,
                       reciprocal_k_nns, rc_keys, rc_vals, in_k, pruned_mask);
    MPN_LAUNCH_CHECK();
    return MPNHIP_OK;
}

extern "C" int mpnhip_window_flags(const int64_t* edge_index, int64_t n_edges, int64_t node_begin, int64_t node_end,
                                   unsigned char* flags, void* stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    MPN_CHECK_ARG(n_edges >= 0, "window_flags: bad sizes");
    if (n_edges == 0) return MPNHIP_OK;
    MPN_CHECK_ARG(edge_index && flags, "window_flags: null pointer");
    hipLaunchKernelGGL(k_window_flags, dim3((unsigned)((n_edges + 255) / 256)), dim3(256), 0, stream, edge_index, n_edges,
                       node_begin, node_end, flags);
    MPN_LAUNCH_CHECK();
    return MPNHIP_OK;
}

extern "C" size_t mpnhip_compact_workspace_bytes(int64_t n) { return align_up(select_temp(n), 256) + 256; }

extern "C" int mpnhip_compact(const unsigned char* flags, int64_t n, int32_t* ids, int32_t* count, void* workspace,
                              size_t workspace_bytes, void* stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    MPN_CHECK_ARG(n >= 0 && n < 2147483647LL, "compact: bad size");
    MPN_CHECK_ARG(count, "compact: null count");
    if (n == 0) {
        MPN_HIP(hipMemsetAsync(count, 0, 4, stream));
        return MPNHIP_OK;
    }
    MPN_CHECK_ARG(flags && ids, "compact: null pointer");
    if (!workspace || workspace_bytes < mpnhip_compact_workspace_bytes(n)) {
        set_error("compact: workspace %zu < %zu", workspace_bytes, mpnhip_compact_workspace_bytes(n));
        return MPNHIP_ERR_WORKSPACE;
    }
    size_t tmp_bytes = select_temp(n);
    MPN_HIP(rocprim::select(workspace, tmp_bytes, rocprim::counting_iterator<int>(0), ids, count, (size_t)n, FlagSet{flags}, stream));
    return MPNHIP_OK;
}

extern "C" int mpnhip_gather_rows(const float* src, int64_t ld, const int32_t* ids, int64_t n, int dim, float* out, void* stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    MPN_CHECK_ARG(n >= 0 && dim >= 0 && ld >= dim, "gather_rows: bad sizes");
    if (n == 0 || dim == 0) return MPNHIP_OK;
    MPN_CHECK_ARG(src && ids && out, "gather_rows: null pointer");
    hipLaunchKernelGGL(k_gather_rows, dim3((unsigned)((n * dim + 255) / 256)), dim3(256), 0, stream, src, ld, ids, n, dim, out);
    MPN_LAUNCH_CHECK();
    return MPNHIP_OK;
}

extern "C" int mpnhip_gather_edges(const int64_t* edge_index, int64_t n_edges, const int32_t* ids, int64_t n, int64_t node_begin,
                                   int64_t* out, void* stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    MPN_CHECK_ARG(n >= 0 && n_edges >= 0, "gather_edges: bad sizes");
    if (n == 0) return MPNHIP_OK;
    MPN_CHECK_ARG(edge_index && ids && out, "gather_edges: null pointer");
    hipLaunchKernelGGL(k_gather_edges, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, edge_index, n_edges, ids, n, node_begin,
                       out);
    MPN_LAUNCH_CHECK();
    return MPNHIP_OK;
}

extern "C" int mpnhip_window_accumulate(const float* logits, const int32_t* kept_ids, int64_t n_kept, const int32_t* window_ids,
                                        int64_t n_window, int set_pruned_edges_to_inactive, float* overall_edge_preds,
                                        float* overall_num_preds, void* stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    MPN_CHECK_ARG(n_kept >= 0 && n_window >= 0 && n_kept <= n_window, "window_accumulate: bad sizes");
    if (n_window == 0) return MPNHIP_OK;
    MPN_CHECK_ARG(window_ids && overall_edge_preds && overall_num_preds, "window_accumulate: null pointer");
    MPN_CHECK_ARG(n_kept == 0 || logits, "window_accumulate: null logits");
    if (n_kept > 0) {
        hipLaunchKernelGGL(k_accumulate, dim3((unsigned)((n_kept + 255) / 256)), dim3(256), 0, stream, logits, kept_ids, n_kept,
                           window_ids, set_pruned_edges_to_inactive ? 0 : 1, overall_edge_preds, overall_num_preds);
        MPN_LAUNCH_CHECK();
    }
    if (set_pruned_edges_to_inactive) {  // every edge of the window counts as predicted (pruned ones as 0)
        hipLaunchKernelGGL(k_count_window, dim3((unsigned)((n_window + 255) / 256)), dim3(256), 0, stream, window_ids, n_window,
                           overall_num_preds);
        MPN_LAUNCH_CHECK();
    }
    return MPNHIP_OK;
}

extern "C" int mpnhip_average_preds(const float* overall_preds, const float* overall_num, int64_t n, float* final_preds, void* stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    MPN_CHECK_ARG(n >= 0, "average_preds: bad size");
    if (n == 0) return MPNHIP_OK;
    MPN_CHECK_ARG(overall_preds && overall_num && final_preds, "average_preds: null pointer");
    hipLaunchKernelGGL(k_average, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, overall_preds, overall_num, n, final_preds);
    MPN_LAUNCH_CHECK();
    return MPNHIP_OK;
}
