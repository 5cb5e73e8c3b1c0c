// Training-mode BatchNorm1d / ReLU / Dropout of the reference's MLP builder (models/mlp.py:12-23: Linear -> [BatchNorm1d] -> ReLU ->
// [Dropout]) and their gradients, for the layer-by-layer training path (mpntrackseg_amd/modular.py).
//
// Batch statistics need every row of the layer's output before the first activation can be formed, so this configuration cannot
// run inside the fused chain kernels (one launch per message-passing step): the host runs it layer by layer -- mpnhip_linear,
// this file, mpnhip_weight_grad -- as the reference does.  No shipped configuration enables either module
// (configs/tracking_cfg.yaml:150-167); the path exists so that a model built with use_batchnorm / dropout_p trains at all.
//
//   forward   mu = mean_r z, var = mean_r (z - mu)^2 (two passes: no E[z^2] - mu^2 cancellation), xh = (z - mu) / sqrt(var + eps),
//             u = gamma xh + beta, y = relu(u) keep / (1 - p);   running_mean / running_var as nn.BatchNorm1d updates them
//             (momentum, UNBIASED variance, torch/nn/modules/batchnorm.py semantics)
//   backward  du = dy keep / (1 - p) [u > 0];  dgamma = sum du xh, dbeta = sum du;
//             dz = gamma / sqrt(var + eps) (du - mean_r du - xh mean_r(du xh))
// Column sums: row chunks -> partials -> one fixed-order sum per column (no float atomics: bitwise reproducible).
// keep(r, c) is a counter-based hash of (seed, r n + c): the backward regenerates it, nothing is stored.
#include "common.h"

namespace mpnhip {
namespace {

constexpr int BN_ROWS = 512;   // rows per chunk of the column-sum kernels

__device__ __forceinline__ bool drop_keep(uint64_t seed, uint64_t idx, float p) {
    // splitmix64 of (seed + idx * golden): 24 uniform bits against p
    uint64_t z = seed + (idx + 1) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    const float u = (float)(z >> 40) * (1.0f / 16777216.0f);
    return u >= p;
}

struct BnArgs {
    const float* z;        // [m, n] pre-activation (the Linear's output)
    const float* dy;       // backward: gradient at y
    const float* gamma;    // [n] or null (1)
    const float* beta;     // [n] or null (0)
    const float* mean;     // [n] batch mean (null: no BatchNorm)
    const float* invstd;   // [n]
    float* out;            // forward: y; backward: dz
    float* part;           // [2][nchunks][n] partial column sums
    const float* colsum;   // backward: [2][n] = sum du, sum du xh
    int64_t m;
    int n, relu, use_bn;
    float p, scale;        // dropout probability, 1 / (1 - p)
    uint64_t seed;
};

__device__ __forceinline__ float bn_u(const BnArgs& a, float z, int c, float& xh) {
    if (!a.use_bn) { xh = 0.f; return z; }
    xh = (z - a.mean[c]) * a.invstd[c];
    return (a.gamma ? a.gamma[c] : 1.f) * xh + (a.beta ? a.beta[c] : 0.f);
}
__device__ __forceinline__ float bn_du(const BnArgs& a, int64_t r, int c, float z, float& xh) {
    const float u = bn_u(a, z, c, xh);
    float d = a.dy[r * a.n + c];
    if (a.p > 0.f) d = drop_keep(a.seed, (uint64_t)r * a.n + c, a.p) ? d * a.scale : 0.f;
    if (a.relu && !(u > 0.f)) d = 0.f;
    return d;
}

// MODE 0: sum z; 1: sum (z - mean)^2; 2: sum du and sum du xh.  Block = 4 row lanes x 64 columns; grid (chunks, column groups).
template <int MODE>
__global__ __launch_bounds__(256) void k_col_partial(BnArgs a) {
    __shared__ float red[2][4][64];
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int c = blockIdx.y * 64 + cl;
    const int64_t r0 = (int64_t)blockIdx.x * BN_ROWS;
    const int64_t r1 = r0 + BN_ROWS < a.m ? r0 + BN_ROWS : a.m;
    float s0 = 0.f, s1 = 0.f;
    if (c < a.n) {
        const float mu = MODE == 1 ? a.mean[c] : 0.f;
        for (int64_t r = r0 + rl; r < r1; r += 4) {
            const float z = a.z[r * a.n + c];
            if (MODE == 0) s0 += z;
            else if (MODE == 1) s0 += (z - mu) * (z - mu);
            else {
                float xh;
                const float d = bn_du(a, r, c, z, xh);
                s0 += d;
                s1 += d * xh;
            }
        }
    }
    red[0][rl][cl] = s0;
    red[1][rl][cl] = s1;
    __syncthreads();
    if (rl == 0 && c < a.n) {
        const size_t nch = gridDim.x;
        a.part[(size_t)blockIdx.x * a.n + c] = (red[0][0][cl] + red[0][1][cl]) + (red[0][2][cl] + red[0][3][cl]);
        if (MODE == 2) a.part[(nch + blockIdx.x) * a.n + c] = (red[1][0][cl] + red[1][1][cl]) + (red[1][2][cl] + red[1][3][cl]);
    }
}

// One thread per column: partials in chunk order.  STAGE 0: mean; 1: invstd (+ running statistics); 2: the two backward sums
// (+ dgamma, dbeta).
template <int STAGE>
__global__ __launch_bounds__(64) void k_col_finish(const float* __restrict__ part, int nch, int n, int64_t m, float eps, float momentum,
                                                   float* __restrict__ o0, float* __restrict__ o1, float* __restrict__ running_mean,
                                                   float* __restrict__ running_var, const float* __restrict__ mean,
                                                   float* __restrict__ dgamma, float* __restrict__ dbeta) {
    const int c = blockIdx.x * 64 + threadIdx.x;
    if (c >= n) return;
    float s0 = 0.f, s1 = 0.f;
    for (int k = 0; k < nch; ++k) {
        s0 += part[(size_t)k * n + c];
        if (STAGE == 2) s1 += part[((size_t)nch + k) * n + c];
    }
    if (STAGE == 0) {
        o0[c] = s0 / (float)m;
    } else if (STAGE == 1) {
        const float var = s0 / (float)m;
        o0[c] = 1.f / sqrtf(var + eps);
        if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean[c];
        if (running_var) running_var[c] = (1.f - momentum) * running_var[c] + momentum * (s0 / (float)(m - 1));
    } else {
        o0[c] = s0;
        o1[c] = s1;
        if (dgamma) dgamma[c] = s1;
        if (dbeta) dbeta[c] = s0;
    }
}

__global__ __launch_bounds__(256) void k_bn_apply(BnArgs a) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= a.m * a.n) return;
    const int c = (int)(i % a.n);
    float xh;
    float u = bn_u(a, a.z[i], c, xh);
    if (a.relu) u = fmaxf(u, 0.f);
    if (a.p > 0.f) u = drop_keep(a.seed, (uint64_t)i, a.p) ? u * a.scale : 0.f;
    a.out[i] = u;
}

__global__ __launch_bounds__(256) void k_bn_bwd_apply(BnArgs a) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= a.m * a.n) return;
    const int c = (int)(i % a.n);
    float xh;
    const float d = bn_du(a, i / a.n, c, a.z[i], xh);
    if (!a.use_bn) { a.out[i] = d; return; }
    const float inv_m = 1.f / (float)a.m;
    const float g = (a.gamma ? a.gamma[c] : 1.f) * a.invstd[c];
    a.out[i] = g * (d - a.colsum[c] * inv_m - xh * (a.colsum[a.n + c] * inv_m));
}

int nchunks(int64_t m) { return (int)((m + BN_ROWS - 1) / BN_ROWS); }

}  // namespace
}  // namespace mpnhip

using namespace mpnhip;

extern "C" size_t mpnhip_bn_dropout_workspace_bytes(int64_t m, int n) {
    if (m <= 0 || n <= 0) return 0;
    return ((size_t)2 * nchunks(m) * n + 2 * (size_t)n) * sizeof(float);
}

extern "C" int mpnhip_bn_relu_dropout_forward(const float* z, int64_t m, int n, int use_bn, const float* gamma, const float* beta,
                                              float* running_mean, float* running_var, float momentum, float eps, int relu,
                                              float dropout_p, uint64_t seed, float* y, float* save_mean, float* save_invstd,
                                              void* workspace, size_t workspace_bytes, void* stream_) {
    hipStream_t s = static_cast<hipStream_t>(stream_);
    MPN_CHECK_ARG(m >= 0 && n >= 1, "bn_relu_dropout_forward: bad shape");
    MPN_CHECK_ARG(dropout_p >= 0.f && dropout_p < 1.f, "bn_relu_dropout_forward: dropout probability must lie in [0, 1)");
    if (m == 0) return MPNHIP_OK;
    MPN_CHECK_ARG(z && y, "bn_relu_dropout_forward: null pointer");
    BnArgs a = {};
    a.z = z; a.out = y; a.m = m; a.n = n; a.relu = relu; a.use_bn = use_bn ? 1 : 0;
    a.p = dropout_p; a.scale = 1.f / (1.f - dropout_p); a.seed = seed;
    if (use_bn) {
        // nn.BatchNorm1d in training mode refuses a single row ("Expected more than 1 value per channel when training")
        MPN_CHECK_ARG(m > 1, "bn_relu_dropout_forward: BatchNorm1d in training mode needs more than one row");
        MPN_CHECK_ARG(save_mean && save_invstd, "bn_relu_dropout_forward: save_mean / save_invstd required with BatchNorm");
        MPN_CHECK_ARG(workspace && workspace_bytes >= mpnhip_bn_dropout_workspace_bytes(m, n), "bn_relu_dropout_forward: workspace too small");
        a.gamma = gamma; a.beta = beta; a.part = static_cast<float*>(workspace);
        const int nch = nchunks(m);
        const dim3 grid((unsigned)nch, (unsigned)((n + 63) / 64));
        hipLaunchKernelGGL(k_col_partial<0>, grid, dim3(256), 0, s, a);
        hipLaunchKernelGGL(k_col_finish<0>, dim3((n + 63) / 64), dim3(64), 0, s, a.part, nch, n, m, eps, momentum, save_mean, nullptr,
                           nullptr, nullptr, nullptr, nullptr, nullptr);
        a.mean = save_mean;
        hipLaunchKernelGGL(k_col_partial<1>, grid, dim3(256), 0, s, a);
        hipLaunchKernelGGL(k_col_finish<1>, dim3((n + 63) / 64), dim3(64), 0, s, a.part, nch, n, m, eps, momentum, save_invstd, nullptr,
                           running_mean, running_var, save_mean, nullptr, nullptr);
        a.invstd = save_invstd;
    }
    hipLaunchKernelGGL(k_bn_apply, dim3((unsigned)((m * n + 255) / 256)), dim3(256), 0, s, a);
    MPN_LAUNCH_CHECK();
    return MPNHIP_OK;
}

extern "C" int mpnhip_bn_relu_dropout_backward(const float* dy, const float* z, int64_t m, int n, int use_bn, const float* gamma,
                                               const float* beta, const float* save_mean, const float* save_invstd, int relu,
                                               float dropout_p, uint64_t seed, float* dz, float* dgamma, float* dbeta,
                                               void* workspace, size_t workspace_bytes, void* stream_) {
    hipStream_t s = static_cast<hipStream_t>(stream_);
    MPN_CHECK_ARG(m >= 0 && n >= 1, "bn_relu_dropout_backward: bad shape");
    MPN_CHECK_ARG(dropout_p >= 0.f && dropout_p < 1.f, "bn_relu_dropout_backward: dropout probability must lie in [0, 1)");
    if (m == 0) return MPNHIP_OK;
    MPN_CHECK_ARG(dy && z && dz, "bn_relu_dropout_backward: null pointer");
    BnArgs a = {};
    a.z = z; a.dy = dy; a.out = dz; a.m = m; a.n = n; a.relu = relu; a.use_bn = use_bn ? 1 : 0;
    a.p = dropout_p; a.scale = 1.f / (1.f - dropout_p); a.seed = seed;
    if (use_bn) {
        MPN_CHECK_ARG(save_mean && save_invstd, "bn_relu_dropout_backward: save_mean / save_invstd required with BatchNorm");
        MPN_CHECK_ARG(workspace && workspace_bytes >= mpnhip_bn_dropout_workspace_bytes(m, n), "bn_relu_dropout_backward: workspace too small");
        a.gamma = gamma; a.beta = beta; a.mean = save_mean; a.invstd = save_invstd; a.part = static_cast<float*>(workspace);
        const int nch = nchunks(m);
        float* colsum = a.part + (size_t)2 * nch * n;
        hipLaunchKernelGGL(k_col_partial<2>, dim3((unsigned)nch, (unsigned)((n + 63) / 64)), dim3(256), 0, s, a);
        hipLaunchKernelGGL(k_col_finish<2>, dim3((n + 63) / 64), dim3(64), 0, s, a.part, nch, n, m, 0.f, 0.f, colsum, colsum + n, nullptr,
                           nullptr, nullptr, dgamma, dbeta);
        a.colsum = colsum;
    }
    hipLaunchKernelGGL(k_bn_bwd_apply, dim3((unsigned)((m * n + 255) / 256)), dim3(256), 0, s, a);
    MPN_LAUNCH_CHECK();
    return MPNHIP_OK;
}

// Gradient of node_agg_fn (mpn.py:266-273; torch_scatter's scatter_add / scatter_mean / scatter_max backward): gather form, one
// thread per source element.  sum: d src[j] = d out[row[j]]; mean: / count[row[j]]; max: only the element argmax names
// (mpnhip_segment_reduce's `argmax`: the first maximum in index order) receives the gradient.
namespace mpnhip {
namespace {
__global__ __launch_bounds__(256) void k_segment_reduce_bwd(const float* __restrict__ dout, const int64_t* __restrict__ row,
                                                            const int32_t* __restrict__ argmax, const int32_t* __restrict__ count,
                                                            int64_t m, int dim, int agg, float* __restrict__ dsrc) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= m * dim) return;
    const int64_t j = i / dim;
    const int d = (int)(i - j * dim);
    const int64_t r = row[j];
    float g = dout[r * dim + d];
    if (agg == MPNHIP_AGG_MEAN) { const int c = count[r]; g /= (float)(c > 0 ? c : 1); }
    else if (agg == MPNHIP_AGG_MAX) g = argmax[r * dim + d] == (int32_t)j ? g : 0.f;
    dsrc[i] = g;
}
}  // namespace
}  // namespace mpnhip

extern "C" int mpnhip_segment_reduce_backward(const float* grad_out, const int64_t* row, const int32_t* argmax, const int32_t* count,
                                              int64_t m, int dim, int x_size, int agg, float* grad_src, void* stream_) {
    hipStream_t s = static_cast<hipStream_t>(stream_);
    MPN_CHECK_ARG(m >= 0 && dim >= 1 && x_size >= 0, "segment_reduce_backward: bad shape");
    MPN_CHECK_ARG(agg == MPNHIP_AGG_SUM || agg == MPNHIP_AGG_MEAN || agg == MPNHIP_AGG_MAX, "segment_reduce_backward: unknown aggregation");
    if (m == 0) return MPNHIP_OK;
    MPN_CHECK_ARG(grad_out && row && grad_src, "segment_reduce_backward: null pointer");
    MPN_CHECK_ARG(agg != MPNHIP_AGG_MAX || argmax, "segment_reduce_backward: max needs the forward's argmax");
    MPN_CHECK_ARG(agg != MPNHIP_AGG_MEAN || count, "segment_reduce_backward: mean needs the segment counts");
    hipLaunchKernelGGL(k_segment_reduce_bwd, dim3((unsigned)((m * dim + 255) / 256)), dim3(256), 0, s, grad_out, row, argmax, count, m,
                       dim, agg, grad_src);
    MPN_LAUNCH_CHECK();
    return MPNHIP_OK;
}
