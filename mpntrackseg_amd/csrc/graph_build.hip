// Graph construction on the device:
//   get_time_valid_conn_ixs   reference utils/graph.py:6-37    which detection pairs may be connected
//   compute_edge_feats_dict   reference utils/graph.py:90-124   the five geometric edge features
//   F.pairwise_distance       reference data/mot_graph.py:298-301  ReID embedding distance per edge
// The reference builds a dense N x N boolean matrix and calls torch.where (row-major order); here every node counts
// its partners, an exclusive scan turns the counts into offsets and a second pass writes the pairs -- same pairs,
// same (row, col) order, no N x N array.  All integer results are bit-exact; the features follow the reference's
// operation order in fp32 (log via logf, the embedding norm re-associated across a wavefront).
#include "common.h"

#include <rocprim/device/device_scan.hpp>

namespace mpnhip {
namespace {

// pairs (i, j), i < j, with 0 < |frame[i] - frame[j]| <= max_dist (max_dist < 0: no upper limit).  One wavefront
// per row i; lanes stride over j.  pass 0 counts, pass 1 writes at offs[i] in ascending j.
template <bool FILL>
__global__ __launch_bounds__(256) void k_time_pairs(const int64_t* __restrict__ frame, int n, int64_t max_dist,
                                                    int64_t* __restrict__ counts, const int64_t* __restrict__ offs,
                                                    int64_t* __restrict__ out_row, int64_t* __restrict__ out_col) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n) return;
    const int64_t fi = frame[i];
    int64_t base = FILL ? offs[i] : 0;
    int64_t total = 0;
    for (int j0 = i + 1; j0 < n; j0 += 64) {
        const int j = j0 + lane;
        bool ok = false;
        if (j < n) {
            int64_t d = frame[j] - fi;
            d = d < 0 ? -d : d;
            ok = d > 0 && (max_dist < 0 || d <= max_dist);
        }
        const unsigned long long m = __ballot(ok);
        if (FILL && ok) {
            const int64_t pos = base + __popcll(m & ((1ull << lane) - 1ull));
            out_row[pos] = i;
            out_col[pos] = j;
        }
        const int c = __popcll(m);
        base += c;
        total += c;
    }
    if (!FILL && lane == 0) counts[i] = total;
}

// utils/graph.py:104-122, output columns in the order of the reference's dict:
//   secs_time_dists, norm_feet_x_dists, norm_feet_y_dists, bb_height_dists, bb_width_dists
__global__ void k_edge_feats(const int64_t* __restrict__ ei, int64_t E, const int64_t* __restrict__ frame, float fps,
                             const float* __restrict__ bb_h, const float* __restrict__ bb_w, const float* __restrict__ feet_x,
                             const float* __restrict__ feet_y, float* __restrict__ out) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    const int64_t r = ei[e], c = ei[E + e];
    const float tr = (float)frame[r] / fps, tc = (float)frame[c] / fps;   // .float() / fps  (graph.py:107)
    const float hr = bb_h[r], hc = bb_h[c];
    const float mean_h = (hr + hc) / 2.f;                                 // graph.py:115
    float* o = out + e * 5;
    o[0] = tc - tr;
    o[1] = (feet_x[c] - feet_x[r]) / mean_h;
    o[2] = (feet_y[c] - feet_y[r]) / mean_h;
    o[3] = logf(hc / hr);
    o[4] = logf(bb_w[c] / bb_w[r]);
}

// || a - b + eps ||_2 per edge (torch.nn.functional.pairwise_distance, p = 2, eps = 1e-6): one wavefront per edge,
// 16-byte pieces of both rows
__global__ __launch_bounds__(256) void k_pairwise_dist(const float* __restrict__ emb, int64_t ld, int dim,
                                                       const int64_t* __restrict__ ei, int64_t E, float eps,
                                                       float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t e = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (e >= E) return;
    const float* a = emb + ei[e] * ld;
    const float* b = emb + ei[E + e] * ld;
    float acc = 0.f;
    if (dim % 4 == 0 && ld % 4 == 0 && (((uintptr_t)emb) & 15) == 0) {
        for (int k = lane * 4; k < dim; k += 256) {
            const float4 x = *reinterpret_cast<const float4*>(a + k);
            const float4 y = *reinterpret_cast<const float4*>(b + k);
            float d;
            d = x.x - y.x + eps; acc = fmaf(d, d, acc);
            d = x.y - y.y + eps; acc = fmaf(d, d, acc);
            d = x.z - y.z + eps; acc = fmaf(d, d, acc);
            d = x.w - y.w + eps; acc = fmaf(d, d, acc);
        }
    } else {
        for (int k = lane; k < dim; k += 64) {
            const float d = a[k] - b[k] + eps;
            acc = fmaf(d, d, acc);
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if (lane == 0) out[e] = sqrtf(acc);
}

static size_t scan_temp(int n) {
    size_t bytes = 0;
    int64_t* p = nullptr;
    (void)rocprim::exclusive_scan(nullptr, bytes, p, p, (int64_t)0, (size_t)(n > 0 ? n : 1), rocprim::plus<int64_t>(), (hipStream_t)0);
    return bytes;
}

}  // namespace
// ---- load_precomputed_embeddings (utils/rgb.py:150-188): id membership and the order check ---------------------------
// keep[i] = the detection id stored in element 0 of row i occurs in det_ids_sorted (binary search; np.isin of rgb.py:179,185)
__global__ void k_embedding_keep(const float* __restrict__ stored, int64_t ld, int64_t n_stored, const int* __restrict__ ids_sorted,
                                 int64_t n_det, unsigned char* __restrict__ keep) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_stored) return;
    const int id = (int)stored[i * ld];   // (.int() of the reference: truncation)
    int64_t lo = 0, hi = n_det;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (ids_sorted[mid] < id) lo = mid + 1; else hi = mid;
    }
    keep[i] = (lo < n_det && ids_sorted[lo] == id) ? 1 : 0;
}
// mismatch[0] += number of j whose kept row's id differs from det_ids[j] (the assertion of rgb.py:180,186)
__global__ void k_embedding_check(const float* __restrict__ stored, int64_t ld, const int* __restrict__ rows, int64_t n,
                                  const int* __restrict__ det_ids, int* __restrict__ mismatch) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    if ((int)stored[(int64_t)rows[j] * ld] != det_ids[j]) atomicAdd(mismatch, 1);
}

}  // namespace mpnhip

using namespace mpnhip;

extern "C" size_t mpnhip_time_valid_conn_workspace_bytes(int n_nodes) {
    return align_up(((size_t)n_nodes + 1) * 8, 256) + align_up(scan_temp(n_nodes + 1), 256) + 256;
}

extern "C" int mpnhip_time_valid_conn_count(const int64_t* frame_num, int n_nodes, int64_t max_frame_dist, int64_t* offsets,
                                            void* workspace, size_t workspace_bytes, void* stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    MPN_CHECK_ARG(n_nodes >= 0, "time_valid_conn: bad node count");
    MPN_CHECK_ARG(offsets, "time_valid_conn: null offsets");
    if (n_nodes == 0) {
        MPN_HIP(hipMemsetAsync(offsets, 0, 8, stream));
        return MPNHIP_OK;
    }
    MPN_CHECK_ARG(frame_num, "time_valid_conn: null frames");
    if (!workspace || workspace_bytes < mpnhip_time_valid_conn_workspace_bytes(n_nodes)) {
        set_error("time_valid_conn: workspace %zu < %zu", workspace_bytes, mpnhip_time_valid_conn_workspace_bytes(n_nodes));
        return MPNHIP_ERR_WORKSPACE;
    }
    char* w = static_cast<char*>(workspace);
    int64_t* counts = reinterpret_cast<int64_t*>(w);
    void* tmp = w + align_up(((size_t)n_nodes + 1) * 8, 256);
    size_t tmp_bytes = scan_temp(n_nodes + 1);
    MPN_HIP(hipMemsetAsync(counts, 0, ((size_t)n_nodes + 1) * 8, stream));
    hipLaunchKernelGGL(k_time_pairs<false>, dim3((unsigned)((n_nodes + 3) / 4)), dim3(256), 0, stream, frame_num, n_nodes,
                       max_frame_dist, counts, nullptr, nullptr, nullptr);
    MPN_LAUNCH_CHECK();
    // offsets[i] = pairs of rows < i; offsets[N] = number of pairs
    MPN_HIP(rocprim::exclusive_scan(tmp, tmp_bytes, counts, offsets, (int64_t)0, (size_t)n_nodes + 1, rocprim::plus<int64_t>(), stream));
    return MPNHIP_OK;
}

extern "C" int mpnhip_time_valid_conn_fill(const int64_t* frame_num, int n_nodes, int64_t max_frame_dist,
                                           const int64_t* offsets, int64_t n_pairs, int64_t* edge_ixs, void* stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    MPN_CHECK_ARG(n_nodes >= 0 && n_pairs >= 0, "time_valid_conn: bad sizes");
    if (n_nodes == 0 || n_pairs == 0) return MPNHIP_OK;
    MPN_CHECK_ARG(frame_num && offsets && edge_ixs, "time_valid_conn: null pointer");
    hipLaunchKernelGGL(k_time_pairs<true>, dim3((unsigned)((n_nodes + 3) / 4)), dim3(256), 0, stream, frame_num, n_nodes,
                       max_frame_dist, nullptr, offsets, edge_ixs, edge_ixs + n_pairs);
    MPN_LAUNCH_CHECK();
    return MPNHIP_OK;
}

extern "C" int mpnhip_edge_features(const int64_t* edge_ixs, int64_t n_edges, int n_nodes, const int64_t* frame_num, float fps,
                                    const float* bb_height, const float* bb_width, const float* feet_x, const float* feet_y,
                                    float* edge_feats, void* stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    MPN_CHECK_ARG(n_edges >= 0 && n_nodes >= 0, "edge_features: bad sizes");
    if (n_edges == 0) return MPNHIP_OK;
    MPN_CHECK_ARG(edge_ixs && frame_num && bb_height && bb_width && feet_x && feet_y && edge_feats, "edge_features: null pointer");
    MPN_CHECK_ARG(fps > 0.f, "edge_features: fps must be positive");
    hipLaunchKernelGGL(k_edge_feats, dim3((unsigned)((n_edges + 255) / 256)), dim3(256), 0, stream, edge_ixs, n_edges, frame_num,
                       fps, bb_height, bb_width, feet_x, feet_y, edge_feats);
    MPN_LAUNCH_CHECK();
    return MPNHIP_OK;
}

extern "C" int mpnhip_pairwise_distance(const float* emb, int64_t ld, int dim, const int64_t* edge_ixs, int64_t n_edges, float eps,
                                        float* dist, void* stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    MPN_CHECK_ARG(n_edges >= 0 && dim >= 0 && ld >= dim, "pairwise_distance: bad sizes");
    if (n_edges == 0) return MPNHIP_OK;
    MPN_CHECK_ARG(emb && edge_ixs && dist, "pairwise_distance: null pointer");
    hipLaunchKernelGGL(k_pairwise_dist, dim3((unsigned)((n_edges + 3) / 4)), dim3(256), 0, stream, emb, ld, dim, edge_ixs, n_edges,
                       eps, dist);
    MPN_LAUNCH_CHECK();
    return MPNHIP_OK;
}

extern "C" int mpnhip_embedding_keep(const float* stored, int64_t ld, int64_t n_stored, const int32_t* det_ids_sorted, int64_t n_det,
                                     unsigned char* keep, void* stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    MPN_CHECK_ARG(n_stored >= 0 && n_det >= 0 && ld >= 1, "embedding_keep: bad sizes");
    if (n_stored == 0) return MPNHIP_OK;
    MPN_CHECK_ARG(stored && keep && (det_ids_sorted || n_det == 0), "embedding_keep: null pointer");
    hipLaunchKernelGGL(k_embedding_keep, dim3((unsigned)((n_stored + 255) / 256)), dim3(256), 0, stream, stored, ld, n_stored,
                       det_ids_sorted, n_det, keep);
    MPN_LAUNCH_CHECK();
    return MPNHIP_OK;
}

extern "C" int mpnhip_embedding_check(const float* stored, int64_t ld, const int32_t* rows, int64_t n, const int32_t* det_ids,
                                      int32_t* mismatch, void* stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    MPN_CHECK_ARG(n >= 0 && ld >= 1, "embedding_check: bad sizes");
    MPN_CHECK_ARG(mismatch, "embedding_check: null pointer");
    MPN_HIP(hipMemsetAsync(mismatch, 0, sizeof(int32_t), stream));
    if (n == 0) return MPNHIP_OK;
    MPN_CHECK_ARG(stored && rows && det_ids, "embedding_check: null pointer");
    hipLaunchKernelGGL(k_embedding_check, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, stored, ld, rows, n, det_ids, mismatch);
    MPN_LAUNCH_CHECK();
    return MPNHIP_OK;
}
