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
#include <cstdlib>
#include <mutex>

#include "common.h"
#include "edge_chain.h"
#include "plan.h"

namespace mpnhip {

// ------------------------------------------------------------------------------------ small kernels
// out = (g (+ g2)) (.) [act > 0]   (ReLU backward), float4; g2: optional second addend (the other K half of a split product)
__global__ void k_relu_mask(const float* __restrict__ g, const float* __restrict__ g2, const float* __restrict__ act,
                            float* __restrict__ out, int64_t n) {
    int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i + 3 < n) {
        float4 a = *reinterpret_cast<const float4*>(act + i), v = *reinterpret_cast<const float4*>(g + i);
        if (g2) {
            const float4 w = *reinterpret_cast<const float4*>(g2 + i);
            v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
        }
        v.x = a.x > 0.f ? v.x : 0.f; v.y = a.y > 0.f ? v.y : 0.f; v.z = a.z > 0.f ? v.z : 0.f; v.w = a.w > 0.f ? v.w : 0.f;
        *reinterpret_cast<float4*>(out + i) = v;
    } else {
        for (; i < n; ++i) out[i] = act[i] > 0.f ? g[i] + (g2 ? g2[i] : 0.f) : 0.f;
    }
}

// Backward of node_agg_fn followed by the ReLU of the last flow layer:
//   dZM[j][c] = [M[j][c] > 0] * dAGG[srow[j]][half(j) + c] (* 1/cnt for mean) (* [ARG == j] for max)
// j in sorted order; half = dn for flow_out rows (j < E_out), 0 for flow_in rows; self-loop rows -> 0.
// One thread per 4 columns (dn % 4 == 0) or per column.
__global__ void k_agg_bwd(const float* __restrict__ dagg, const float* __restrict__ msg, const int* __restrict__ arg,
                          const int* __restrict__ srow, const int* __restrict__ seg_ptr, const int* __restrict__ header,
                          int N, int64_t E, int dn, int agg, int has_relu, float* __restrict__ out) {
    const bool vec = (dn & 3) == 0;
    const int per = vec ? dn >> 2 : dn;
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t j = t / per;
    int c = (int)(t % per) * (vec ? 4 : 1);
    if (j >= E) return;
    const int e_out = header[1], e_in = header[2];
    const int w = vec ? 4 : 1;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (j < e_out + e_in) {
        const int dir = j < e_out ? 0 : 1;
        const int row = srow[j];
        const int64_t o = (int64_t)row * 2 * dn + (dir == 0 ? dn : 0) + c;
        float scale = 1.f;
        if (agg == MPNHIP_AGG_MEAN) {
            const int key = dir * N + row;
            const int cnt = seg_ptr[key + 1] - seg_ptr[key];
            scale = (float)(cnt > 0 ? cnt : 1);
        }
        for (int q = 0; q < w; ++q) {
            float x = dagg[o + q];
            if (agg == MPNHIP_AGG_MEAN) x = x / scale;
            else if (agg == MPNHIP_AGG_MAX) x = arg[o + q] == (int)j ? x : 0.f;
            if (has_relu) x = msg[j * dn + c + q] > 0.f ? x : 0.f;
            v[q] = x;
        }
    }
    if (vec) *reinterpret_cast<float4*>(out + j * dn + c) = make_float4(v[0], v[1], v[2], v[3]);
    else out[j * dn + c] = v[0];
}

// dst[r][c0 + c] += src[r][c]
__global__ void k_add_block(const float* __restrict__ src, int64_t lds, float* __restrict__ dst, int64_t ldd, int c0,
                            int rows, int cols) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)rows * cols) return;
    int r = (int)(i / cols), c = (int)(i % cols);
    dst[(int64_t)r * ldd + c0 + c] += src[(int64_t)r * lds + c];
}

// the same for four row blocks of one source in one launch (the packed node-projection gradient back into its layers' gradients)
struct AddParts { float* dst[4]; int64_t ldd[4]; int c0[4]; int r0[4]; int rows[4]; };
__global__ void k_add_block4(const float* __restrict__ src, int64_t lds, int cols, AddParts P) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int q = 0;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int64_t n = (int64_t)P.rows[k] * cols;
        if (q == k && i >= n) { i -= n; q = k + 1; }
    }
    if (i >= (int64_t)P.rows[q] * cols) return;
    const int r = (int)(i / cols), c = (int)(i % cols);
    P.dst[q][(int64_t)r * P.ldd[q] + P.c0[q] + c] += src[(int64_t)(P.r0[q] + r) * lds + c];
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

// out[i] = sum_b src[b * stride + i]   (n4 float4 elements, ascending b: deterministic)
__global__ void k_sum_blocks(const float* __restrict__ src, int64_t stride, int nb, int64_t n4, float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    float4 acc = *reinterpret_cast<const float4*>(src + 4 * i);
    for (int b = 1; b < nb; ++b) {
        const float4 v = *reinterpret_cast<const float4*>(src + (int64_t)b * stride + 4 * i);
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    *reinterpret_cast<float4*>(out + 4 * i) = acc;
}

// out[i] = sum_b src[b * stride + i] over bf16 blocks (n8 groups of eight elements, ascending b), fp32 result; up to 12 blocks'
// 16-byte loads are in flight per lane at once (a serial chain of 8-byte loads ran at 1.9 TB/s at cfg-E)
// OUT16: the sum leaves as bf16 (RNE) -- where every consumer rounds it to bf16 as an operand anyway (the hoisted e0 share: a bf16
// GEMM's A operand and a one-piece weight-gradient product), the same values at half the bytes written and read twice
template <bool OUT16>
__global__ __launch_bounds__(256) void k_sum_blocks_bf16(const unsigned short* __restrict__ src, int64_t stride, int nb, int64_t n8,
                                                         float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n8) return;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int b0 = 0; b0 < nb; b0 += 12) {
        uint4 v[12];
#pragma unroll
        for (int u = 0; u < 12; ++u) {
            const int b = b0 + u < nb ? b0 + u : nb - 1;   // clamped, unconditional loads (summed in ascending b below)
            v[u] = *reinterpret_cast<const uint4*>(src + (int64_t)b * stride + 8 * i);
        }
#pragma unroll
        for (int u = 0; u < 12; ++u) {
            if (b0 + u < nb) {
                const unsigned w[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    acc[2 * q] += __uint_as_float(w[q] << 16);
                    acc[2 * q + 1] += __uint_as_float(w[q] & 0xffff0000u);
                }
            }
        }
    }
    if (OUT16) {
        typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
        const bf16x8 o = {(__bf16)acc[0], (__bf16)acc[1], (__bf16)acc[2], (__bf16)acc[3], (__bf16)acc[4], (__bf16)acc[5], (__bf16)acc[6], (__bf16)acc[7]};
        *reinterpret_cast<uint4*>(reinterpret_cast<unsigned short*>(out) + 8 * i) = __builtin_bit_cast(uint4, o);
    } else {
        *reinterpret_cast<float4*>(out + 8 * i) = make_float4(acc[0], acc[1], acc[2], acc[3]);
        *reinterpret_cast<float4*>(out + 8 * i + 4) = make_float4(acc[4], acc[5], acc[6], acc[7]);
    }
}

__global__ void k_bf16_to_f32(const unsigned short* __restrict__ src, float* __restrict__ dst, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = __uint_as_float((unsigned)src[i] << 16);
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

// 16-byte zeros over two regions (grid-stride)
__global__ __launch_bounds__(256) void k_zero_pair(uint4* __restrict__ a, int64_t na, uint4* __restrict__ b, int64_t nb) {
    const uint4 z = make_uint4(0u, 0u, 0u, 0u);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < na + nb; i += (int64_t)gridDim.x * 256) {
        if (i < na) a[i] = z;
        else b[i - na] = z;
    }
}

static int relu_mask(const float* g, const float* act, float* out, int64_t n, hipStream_t s, const float* g2 = nullptr) {
    if (n <= 0) return MPNHIP_OK;
    hipLaunchKernelGGL(k_relu_mask, dim3((unsigned)((n / 4 + 256) / 256)), dim3(256), 0, s, g, g2, act, out, n);
    MPN_LAUNCH_CHECK();
    return MPNHIP_OK;
}

// ------------------------------------------------------------------------------------ side stream
// One internal stream + two events per process (one process per GPU), created on the first backward call:
// the weight-gradient products of the later steps run there, concurrently with the earlier steps' chains.
struct SideStream {
    hipStream_t stream = nullptr;
    hipStream_t stream2 = nullptr;   // the tail batch (hoisted shares, encoder layers): beside the last group of steps, not behind it
    hipEvent_t ready = nullptr, done = nullptr, done2 = nullptr;
};
// one per device, created on first use; g_side_mu serialises the ENQUEUE phase of concurrent mpnhip_backward calls (threads /
// caller streams of one process): each call's event record -> wait pairs are then issued as a unit, and a wait refers to the
// record made just before it.  (Device work of different caller streams still overlaps; their slab buffers are per call.)
constexpr int MAX_DEVICES = 64;
static SideStream g_side_dev[MAX_DEVICES];
static std::mutex g_side_mu;

static int side_stream_ready(SideStream** out) {
    int dev = -1;
    MPN_HIP(hipGetDevice(&dev));
    MPN_CHECK_ARG(dev >= 0 && dev < MAX_DEVICES, "backward: device ordinal %d", dev);
    SideStream& ss = g_side_dev[dev];
    if (!ss.stream) {
        // lowest priority: the caller's stream carries the critical path (the step loop); when both have workgroups pending, the
        // dispatcher should place the loop's first (MPNHIP_SIDE_PRIORITY=0: default priority, for measurements)
        int least = 0, greatest = 0;
        (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
        static const bool low = [] { const char* e = getenv("MPNHIP_SIDE_PRIORITY"); return !(e && e[0] == '0'); }();
        MPN_HIP(hipStreamCreateWithPriority(&ss.stream, hipStreamNonBlocking, low ? least : 0));
        MPN_HIP(hipStreamCreateWithPriority(&ss.stream2, hipStreamNonBlocking, low ? least : 0));
        MPN_HIP(hipEventCreateWithFlags(&ss.ready, hipEventDisableTiming));
        MPN_HIP(hipEventCreateWithFlags(&ss.done, hipEventDisableTiming));
        MPN_HIP(hipEventCreateWithFlags(&ss.done2, hipEventDisableTiming));
    }
    *out = &ss;
    return MPNHIP_OK;
}

// Once work has been forked to the side stream the caller's stream must wait for it on EVERY exit path -- also when a later
// launch fails and the function returns early: the side stream still reads the forward / backward workspaces and adds into
// the gradient buffers, all of which the caller is free to release as soon as this call returns.
struct SideJoin {
    SideStream* ss = nullptr;
    hipStream_t caller = nullptr;
    bool forked = false, joined = false;
    ~SideJoin() {
        if (forked && !joined && ss) {
            if (hipEventRecord(ss->done2, ss->stream2) != hipSuccess || hipStreamWaitEvent(caller, ss->done2, 0) != hipSuccess)
                (void)hipStreamSynchronize(ss->stream2);
            if (hipEventRecord(ss->done, ss->stream) != hipSuccess || hipStreamWaitEvent(caller, ss->done, 0) != hipSuccess)
                (void)hipStreamSynchronize(ss->stream);   // last resort: block the host rather than leave the work unordered
        }
    }
};

// ------------------------------------------------------------------------------------ plan
// Pre-activation gradients (dZ) of EVERY step are kept ([L][rows][width] blocks): the activation-gradient
// chain of a step reads its predecessor's block, and after the step loop each weight's gradient is ONE
// batched split-row product over all L steps (12x fewer launches and slab reductions than per step).
struct BwdPlan {
    float* dX[2];                       // [N, dn] ping-pong: gradient w.r.t. x_s
    float* dXh;                         // [N, dn] second K half of the per-node projection's activation gradient (split product)
    float* dX0;                         // [N, dn] gradient w.r.t. the encoder's node output
    float* dE0;                         // [E, de]
    float* dAGG;                        // [N, 2dn]
    float* dCat;                        // [max(E ke, N kx)] gradient w.r.t. the concatenated [initial | current] features
    float* dPsum;                       // [N, pw] sum over the steps of dP (the re-attached x0's share is one product)
    float* dZ1sum;                      // [E, he] sum over the steps of the edge MLP's first-layer dZ (the re-attached e0's share)
    float* dZn;                         // [L][N, dn]
    float* dP;                          // [L][N, pw]
    unsigned short* dP16;               // bf16-operand training: the same blocks rounded to bf16 rows (the node-level weight-gradient
    unsigned short* dPsum16;            // products over bf16 rows, wgrad_rows16.hip), and the rounded sum over the steps [N, pw]
    unsigned short* dZn16;              // ... and the node update's operands: [L][N, dn] pre-activation gradients, [L][N, 2 dn] aggregated
    unsigned short* AGG16;              // messages, rounded on the side stream just before the group's products
    unsigned short* enc16;              // ... and the node encoder's dZ / input blocks [N, sum of (out_i + in_i)] (rounded on the tail's stream)
    size_t enc16_elems;
    float* dZfl[MPNHIP_MAX_LAYERS];     // flow MLP layer i:   [L][E, out_i]
    float* dZed[MPNHIP_MAX_LAYERS];     // edge MLP layer i:   [L][E, out_i]  (last layer: the masked dE_s)
    float* dZcl[MPNHIP_MAX_LAYERS];     // classifier layer i: [L][E, out_i]  (i < n-1; the last one is grad_logits)
    float* T[3];                        // edge-encoder chain scratch [E, max encoder width], rotating (layer i -> buffer (n - 1 - i) % 3)
    float* Tn[3];                       // node-encoder chain scratch [N, max encoder width]: chains of <= 3 layers never overwrite a
                                        // dZ block, so their weight-gradient products may run later, on another stream
    int t_width;                        // that width (floats per row)
    // zero-padded copies of the per-edge weights for the fused backward chain (native [n][k] orientation)
    float* wf2p[2]; float* wfep[2]; float* wc1p; float* w2p; float* w1ep;
    float* gWnode;                      // [pw, kx] gradient of the packed node-projection weights
    float* zero_end;                    // end of the run dX[0] | dX0 | dE0 | gWnode that a backward zero-fills at once
    unsigned short* ncb_img;            // unit images of the fused node-side backward kernel (node_chain.hip) or nullptr
    float* wt_scratch;                  // MPNHIP_PREC_BF16: transposed weight blocks of the activation-gradient products
    size_t wt_scratch_floats;
    float* wt_keep[2];                  // ... the node update's (0) and the projections' (1) blocks, transposed once per backward
    size_t wt_keep_floats[2];
    unsigned short* wt_keep16[2];       // ... and their bf16 images: the B operand of the tiled bf16 GEMM as bf16 rows (gemm_bf16.hip)
    // bf16-operand training on the fused kernels (FwdPlan::b16): the dZ blocks above are bf16 rows (half the floats), the gradient
    // w.r.t. e_s travels between the steps in two fp32 buffers, and the backward chain kernel has its own pair images
    bool b16;
    float* dEpp[2];                     // [E, de] fp32 ping-pong
    char* cb16_img;                     // backward pair images (edge | classifier | flow_out | flow_in)
    size_t cb16_off_cls, cb16_off_flow[2];
    float* slab;                        // split partials of the weight-gradient products (2 groups)
    float* slab_side;                   // the same for the products issued on the side stream
    size_t slab_floats_per_group;
    float* slab_wp;                     // slabs of a batch of row-panel products (all jobs of one group of steps)
    size_t slab_wp_floats;
    float* slab_tail;                   // slabs of the tail batch (hoisted shares + encoder layers)
    size_t slab_tail_floats;
    size_t total;
};

static int enc_maxw(const mpnhip_model& m, const Dims& d) {
    int w = d.dn > d.de ? d.dn : d.de;
    const mpnhip_mlp* all[] = {&m.enc_node, &m.enc_edge};
    for (const mpnhip_mlp* p : all)
        for (int i = 0; i < p->n_layers; ++i) w = p->out_dims[i] > w ? p->out_dims[i] : w;
    return w;
}

// slab floats of the weight-gradient products of layers first..n-1 of an MLP (layer 0's k_in given)
static size_t mlp_slab(const mpnhip_mlp& m, int k_in0, int64_t rows, int nbatch) {
    size_t mx = 0;
    for (int i = 0; i < m.n_layers; ++i) {
        size_t f = tn_slab_floats(m.out_dims[i], i == 0 ? k_in0 : m.out_dims[i - 1], rows, nbatch);
        mx = f > mx ? f : mx;
    }
    return mx;
}

static size_t plan_backward(const mpnhip_model& m, const Dims& d, int64_t N, int64_t E, void* base, BwdPlan* out) {
    Arena a = {static_cast<char*>(base), 0};
    BwdPlan p = {};
    const size_t L = d.L > 0 ? d.L : 1;
    // (what every backward starts from as zeros lies side by side -- dX[0] | dX0 | dE0 | gWnode: ONE fill instead of four, ~12 us of
    // the 0.5 ms step of a KITTIMOTS-size graph)
    p.dX[0] = a.f((size_t)N * d.dn);
    p.dX0 = a.f((size_t)N * d.dn);
    p.dE0 = a.f((size_t)E * d.de);
    p.gWnode = a.f((size_t)d.pw * d.kx);
    p.zero_end = a.f(0);
    p.dX[1] = a.f((size_t)N * d.dn);
    p.dXh = a.f((size_t)N * d.dn);
    p.dPsum = a.f((size_t)N * d.pw);
    p.dZ1sum = a.f((size_t)E * d.he);
    p.dAGG = a.f((size_t)N * 2 * d.dn);
    {
        size_t a1 = (size_t)E * d.ke, a2 = (size_t)N * d.kx;
        p.dCat = a.f(a1 > a2 ? a1 : a2);
    }
    p.dZn = a.f(L * N * d.dn);
    p.dP = a.f(L * N * d.pw);
    p.b16 = chain_bf16_train_ok(m, d);
    p.dP16 = p.dPsum16 = p.dZn16 = p.AGG16 = p.enc16 = nullptr;
    p.enc16_elems = 0;
    const bool node16 = p.b16 && d.pw % 8 == 0 && d.dn % 8 == 0;   // (not `p.dP16 != nullptr`: a size query plans without a base)
    if (node16) {
        p.dP16 = reinterpret_cast<unsigned short*>(a.f((L * N * d.pw + 1) / 2));
        p.dPsum16 = reinterpret_cast<unsigned short*>(a.f(((size_t)N * d.pw + 1) / 2));
        p.dZn16 = reinterpret_cast<unsigned short*>(a.f((L * N * d.dn + 1) / 2));
        p.AGG16 = reinterpret_cast<unsigned short*>(a.f((L * N * 2 * d.dn + 1) / 2));
        size_t w = 0;
        for (int i = 0; i < m.enc_node.n_layers; ++i) w += (size_t)m.enc_node.out_dims[i] + (i == 0 ? m.enc_node.in_dim : m.enc_node.out_dims[i - 1]);
        p.enc16_elems = (size_t)N * w;
        p.enc16 = reinterpret_cast<unsigned short*>(a.f((p.enc16_elems + 1) / 2));
    }
    auto dzb = [&](int width) { return a.f(p.b16 ? (L * E * width + 1) / 2 : L * E * width); };
    for (int i = 0; i < m.flow_in.n_layers; ++i) p.dZfl[i] = dzb(m.flow_in.out_dims[i]);
    for (int i = 0; i < m.edge.n_layers; ++i) p.dZed[i] = dzb(m.edge.out_dims[i]);
    for (int i = 0; i + 1 < m.classifier.n_layers; ++i) p.dZcl[i] = dzb(m.classifier.out_dims[i]);
    p.dEpp[0] = p.dEpp[1] = nullptr;
    p.cb16_img = nullptr;
    if (p.b16) {
        for (int i = 0; i < 2; ++i) p.dEpp[i] = a.f((size_t)E * d.de);
        const size_t bytes = chain_bf16_bwd_image_bytes(d.he, d.de, d.hn, d.dn, m.classifier.out_dims[0], &p.cb16_off_cls, &p.cb16_off_flow[0],
                                                        &p.cb16_off_flow[1]);
        p.cb16_img = reinterpret_cast<char*>(a.f(bytes / 4));
    }
    int mw = enc_maxw(m, d);
    if (mw < 52) mw = 52;   // (the fused reference edge encoder keeps dz2 | dz1 | dz0 side by side in T[0])
    for (int i = 0; i < 3; ++i) p.T[i] = a.f((size_t)E * mw);
    for (int i = 0; i < 3; ++i) p.Tn[i] = a.f((size_t)N * mw);
    p.t_width = (int)mw;
    {
        const size_t HE = pad32(d.he), DE = pad32(d.de), HN = pad32(d.hn), DN = pad32(d.dn);
        // (sized for the split images, 3/2 of the fp32 ones)
        for (int q = 0; q < 2; ++q) { p.wf2p[q] = a.f(DN * HN * 3 / 2); p.wfep[q] = a.f(HN * DE * 3 / 2); }
        p.wc1p = a.f(32 * DE * 3 / 2);
        p.w2p = a.f(DE * HE * 3 / 2);
        p.w1ep = a.f(HE * 2 * DE * 3 / 2);
    }
    p.ncb_img = nullptr;
    if (node_chain_bwd_supported(d.dn, d.pw, d.kx) && m.node.n_layers == 1 && m.precision == MPNHIP_PREC_FP32_SPLIT)
        p.ncb_img = reinterpret_cast<unsigned short*>(a.f((node_chain_bwd_image_shorts(d.dn, d.pw, nullptr) + 1) / 2));
    {   // MPNHIP_PREC_BF16: transposition scratch of the activation-gradient products (two direction groups of the largest weight)
        size_t mx = (size_t)d.pw * d.kx;
        const mpnhip_mlp* all[] = {&m.enc_node, &m.enc_edge, &m.edge, &m.flow_in, &m.flow_out, &m.node, &m.classifier};
        for (const mpnhip_mlp* q : all)
            for (int i = 0; i < q->n_layers; ++i) {
                const size_t w = (size_t)q->out_dims[i] * (i == 0 ? q->in_dim : q->out_dims[i - 1]);
                mx = w > mx ? w : mx;
            }
        p.wt_scratch_floats = m.precision == MPNHIP_PREC_BF16 ? 2 * mx : 0;
        p.wt_scratch = a.f(p.wt_scratch_floats);
        p.wt_keep_floats[0] = m.precision == MPNHIP_PREC_BF16 && m.node.n_layers >= 1 ? (size_t)d.dn * m.node.in_dim : 0;
        p.wt_keep_floats[1] = m.precision == MPNHIP_PREC_BF16 ? (size_t)d.pw * d.dn : 0;
        for (int i = 0; i < 2; ++i) p.wt_keep[i] = a.f(p.wt_keep_floats[i]);
        for (int i = 0; i < 2; ++i) p.wt_keep16[i] = reinterpret_cast<unsigned short*>(a.f((p.wt_keep_floats[i] + 1) / 2));
    }
    size_t sl = 0;
    auto upd = [&](size_t f) { sl = f > sl ? f : sl; };
    // exactly the products mpnhip_backward launches (the slab size depends on the shape through tn_plan)
    upd(mlp_slab(m.enc_node, m.enc_node.in_dim, N, 1));
    upd(mlp_slab(m.enc_edge, m.enc_edge.in_dim, E, 1));
    upd(mlp_slab(m.edge, d.ke, E, (int)L));
    for (int nb = 1; nb <= (int)L; ++nb) upd(tn_slab_floats(d.he, d.de, E, nb));   // (the e0-hoisted forms of the edge layer-0 product)
    upd(mlp_slab(m.flow_in, d.de, E, (int)L));
    upd(mlp_slab(m.classifier, d.de, E, (int)L));
    upd(tn_slab_floats(d.dn, 2 * d.dn, N, (int)L));
    upd(tn_slab_floats(d.pw, d.kx, N, (int)L));
    // the same products in the row-panel form (MPNHIP_PREC_FP32_SPLIT): alone on the caller's stream ...
    auto updw = [&](int n_out, int k_in, int64_t rows, int nb) { if (rows > 0) upd((wp_slab_floats(n_out, k_in, rows, nb, false, false) + 1) / 2); };
    for (int i = 0; i < m.enc_node.n_layers; ++i) updw(m.enc_node.out_dims[i], i == 0 ? m.enc_node.in_dim : m.enc_node.out_dims[i - 1], N, 1);
    for (int i = 0; i < m.enc_edge.n_layers; ++i) updw(m.enc_edge.out_dims[i], i == 0 ? m.enc_edge.in_dim : m.enc_edge.out_dims[i - 1], E, 1);
    updw(d.he, d.de, E, 1);
    if (p.b16 && E > 0) upd((wp_slab_floats(d.he, d.de, E, 1, false, false, true) + 1) / 2);   // (the hoisted e0 share over bf16 rows)
    updw(d.pw, d.dn, N, 1);
    if (node16 && N > 0) upd((wp_slab_floats(d.pw, d.dn, N, 1, false, false, true) + 1) / 2);   // (the hoisted x0 share over bf16 rows)
    for (int i = 0; i < m.classifier.n_layers; ++i) updw(m.classifier.out_dims[i], i == 0 ? d.de : m.classifier.out_dims[i - 1], E, 1);
    p.slab_floats_per_group = sl;
    p.slab = a.f(2 * sl);
    p.slab_side = a.f(2 * sl);
    // ... and all products of a group of nb steps in one batch (mp_weight_grads): every job has its own slabs
    size_t wpmax = 0;
    for (int nb = 1; nb <= (int)L; ++nb) {
        size_t t = 0;
        auto addw = [&](int n_out, int k_in, int64_t rows, bool ranged) {
            if (rows <= 0) return;
            const size_t f32 = wp_slab_floats(n_out, k_in, rows, nb, ranged, true), f16 = p.b16 ? wp_slab_floats(n_out, k_in, rows, nb, ranged, true, true) : 0;
            t += f32 > f16 ? f32 : f16;
        };
        addw(d.dn, 2 * d.dn, N, false);
        for (int i = 1; i < m.flow_in.n_layers; ++i) { addw(m.flow_in.out_dims[i], m.flow_in.out_dims[i - 1], E, true); addw(m.flow_in.out_dims[i], m.flow_in.out_dims[i - 1], E, true); }
        addw(d.hn, d.de, E, true); addw(d.hn, d.de, E, true);
        for (int i = 0; i < m.classifier.n_layers; ++i) addw(m.classifier.out_dims[i], i == 0 ? d.de : m.classifier.out_dims[i - 1], E, false);
        for (int i = 1; i < m.edge.n_layers; ++i) addw(m.edge.out_dims[i], m.edge.out_dims[i - 1], E, false);
        // (a product that runs in one of two forms -- the first-layer inputs whole or with the re-attached share hoisted -- reserves
        // the larger of the two: a narrower k_in can mean MORE row chunks, i.e. more slabs)
        auto addw2 = [&](int n_out, int k_a, int k_b, int64_t rows) {
            const size_t before = t;
            addw(n_out, k_a, rows, false);
            const size_t fa = t - before;
            t = before;
            addw(n_out, k_b, rows, false);
            if (t - before < fa) t = before + fa;
        };
        addw2(d.he, d.ke, d.de, E);
        addw2(d.pw, d.kx, d.dn, N);
        wpmax = t > wpmax ? t : wpmax;
    }
    p.slab_wp_floats = wpmax;
    p.slab_wp = a.f(wpmax);
    {   // the tail batch: hoisted shares + encoder layers
        size_t t = 0;
        auto addt = [&](int n_out, int k_in, int64_t rows) { if (rows > 0) t += wp_slab_floats(n_out, k_in, rows, 1, false, true); };
        if (node16 && N > 0) {
            const size_t f32 = wp_slab_floats(d.pw, d.dn, N, 1, false, true), f16 = wp_slab_floats(d.pw, d.dn, N, 1, false, true, true);
            t += f32 > f16 ? f32 : f16;
        } else {
            addt(d.pw, d.dn, N);
        }
        addt(d.he, d.de, E);
        if (p.b16 && E > 0) {   // (bf16-operand training keeps S = sum_s dZ1_s as bf16 rows: the bf16-row kernels' chunking)
            const size_t f32 = wp_slab_floats(d.he, d.de, E, 1, false, true), f16 = wp_slab_floats(d.he, d.de, E, 1, false, true, true);
            if (f16 > f32) t += f16 - f32;
        }
        for (int i = 0; i < m.enc_node.n_layers; ++i) {
            const int n_out = m.enc_node.out_dims[i], k_in = i == 0 ? m.enc_node.in_dim : m.enc_node.out_dims[i - 1];
            if (node16 && N > 0) {
                const size_t f32 = wp_slab_floats(n_out, k_in, N, 1, false, true), f16 = wp_slab_floats(n_out, k_in, N, 1, false, true, true);
                t += f32 > f16 ? f32 : f16;
            } else {
                addt(n_out, k_in, N);
            }
        }
        for (int i = 0; i < m.enc_edge.n_layers; ++i) addt(m.enc_edge.out_dims[i], i == 0 ? m.enc_edge.in_dim : m.enc_edge.out_dims[i - 1], E);
        p.slab_tail_floats = t;
        p.slab_tail = a.f(t);
    }
    p.total = a.off;
    if (out) *out = p;
    return p.total;
}

// ------------------------------------------------------------------------------------ helpers
struct RowRange {
    const int* begin;
    const int* end;
};

struct Operand {        // one side of a (batched) weight-gradient product
    const float* p;
    int64_t ld;
    int64_t bstride;    // floats between consecutive batches (0: same block every batch)
};

// set by mpnhip_backward for a model in MPNHIP_PREC_FP32_SPLIT: weight gradients in the three-piece operand form (wgrad_panel.hip)
static thread_local bool g_wgrad_split = false;
// MPNHIP_PREC_BF16 (training): every product of the backward rounds its operands to bf16 like the forward's -- the activation
// gradients dH = dZ W through the K-contiguous bf16 GEMM (the weight block transposed into g_wt_scratch first: the kernel takes
// nn.Linear-style [out][in] operands), the weight gradients through the row-panel kernel with ONE bf16 piece per operand
static thread_local bool g_bwd_bf16 = false;
// set around the products whose operands are bf16 rows (WpProduct::src16): the dZ blocks / saved activations of the fused bf16 chain
static thread_local bool g_wg_src16 = false;
struct Src16Scope { bool old; explicit Src16Scope(bool v) : old(g_wg_src16) { g_wg_src16 = v; } ~Src16Scope() { g_wg_src16 = old; } };
static thread_local float* g_wt_scratch = nullptr;
static thread_local size_t g_wt_scratch_floats = 0;
// ... and two blocks that are KEPT for the whole backward: the node update's and the per-node projections' weights are the operands of
// an activation-gradient product in every step (22 transpositions of the same two blocks per cfg-E step before round 4, 25 - 30 us each)
struct WtKeep { float* wt; size_t floats; bool valid; unsigned short* wt16; };
static thread_local WtKeep g_wt_keep[2] = {{nullptr, 0, false, nullptr}, {nullptr, 0, false, nullptr}};

// dW += dZ^T [H | H2] (+ bias) for one or two groups, over nbatch row blocks
static int weight_grad(const BwdPlan& p, float* slab_base, int ngroups, Operand dZ, const int* dz_idx, Operand H, Operand H2, int csplit,
                       const int* h_idx, int n_out, int k_in, float* const gw[2], int64_t ldw, float* const gb[2],
                       const RowRange rr[2], int64_t rows, int nbatch, hipStream_t s) {
    if (g_wgrad_split && rows > 0) {
        // MPNHIP_PREC_FP32_SPLIT: the row-panel kernel (wgrad_panel.hip) -- recorded into the open batch (all products of a group of
        // steps: one product launch + one slab-sum launch), or run as a batch of its own
        WpProduct wp[2];
        for (int q = 0; q < ngroups; ++q)
            wp[q] = {dZ.p, dZ.ld, dZ.bstride, H.p, H.ld, H.bstride, rr ? rr[q].begin : nullptr, rr ? rr[q].end : nullptr, rows, nbatch,
                     n_out, k_in, gw[q], ldw, gb ? gb[q] : nullptr, dz_idx, h_idx, H2.p, H2.ld, H2.bstride, csplit, g_bwd_bf16 ? 1 : 3,
                     g_wg_src16 ? 1 : 0};
        if (wp_batch_open()) {
            if (wp_batch_add(wp, ngroups)) return MPNHIP_OK;
            bool eligible = true;
            for (int q = 0; q < ngroups; ++q) eligible = eligible && wp_eligible(wp[q]);
            if (eligible) {
                // the open batch is full (more than WP_MAX_JOBS jobs in a group of steps: deeper MLPs; or its slab region): run it
                // and go on in a fresh one rather than switch kernels in the middle of a group
                int st = MPNHIP_OK;
                if (wp_batch_roll(&st) && wp_batch_add(wp, ngroups)) return MPNHIP_OK;
                MPN_TRY(st);
            }
            // not a shape / alignment of the row-panel kernel (or the batch cannot be rolled): the fp32 kernel below, counted
            count_path(PC_TN_PANEL_FALLBACK);
            if (g_wg_src16) { set_error("backward (bf16): a product over bf16 rows [%d x %d] is not covered by the row-panel kernel", n_out, k_in); return MPNHIP_ERR_UNSUPPORTED; }
        } else {
            WpBatch own;
            WpBatchGuard guard;
            wp_batch_begin(&own, slab_base, 2 * p.slab_floats_per_group, false);
            if (wp_batch_add(wp, ngroups)) return wp_batch_flush(s);
            wp_batch_abort();
            count_path(PC_TN_PANEL_FALLBACK);
            if (g_wg_src16) { set_error("backward (bf16): a product over bf16 rows [%d x %d] is not covered by the row-panel kernel", n_out, k_in); return MPNHIP_ERR_UNSUPPORTED; }
        }
    }
    TnArgs a = {};
    a.ngroups = ngroups;
    a.n_out = n_out;
    a.k_in = k_in;
    a.csplit = H2.p ? csplit : k_in;
    a.m_upper = rows;
    a.nbatch = nbatch;
    // (with device-side row ranges the groups partition the `rows` rows between them; without, each group covers all of them)
    a.flops = 2.0 * (double)rows * (rr ? 1 : ngroups) * nbatch * n_out * k_in;
    for (int q = 0; q < ngroups; ++q) {
        TnGroup& g = a.g[q];
        g.dZ = dZ.p;
        g.ldz = dZ.ld;
        g.z_bstride = dZ.bstride;
        g.dz_idx = dz_idx;
        g.H = H.p;
        g.ldh = H.ld;
        g.h_bstride = H.bstride;
        g.H2 = H2.p;
        g.ldh2 = H2.ld;
        g.h2_bstride = H2.bstride;
        g.h_idx = h_idx;
        g.row_begin = rr ? rr[q].begin : nullptr;
        g.row_end = rr ? rr[q].end : nullptr;
        g.m_static = rows;
        g.slab = slab_base + q * p.slab_floats_per_group;
        g.grad_w = gw[q];
        g.ldw = ldw;
        g.grad_b = gb ? gb[q] : nullptr;
    }
    return launch_gemm_tn(a, s);
}

// C = mask( A B (+ C) ) with B given as weight rows: B[k][n] = W[k * ldw + n]  (dH = dZ W)
// a16: A is bf16 rows in memory (unsigned shorts behind the pointer, lda counts them; MPNHIP_PREC_BF16 only: gemm_bf16.hip reads them)
static int act_grad(int ngroups, const float* A, int64_t lda, const int* a_idx, const float* const W[2], int64_t ldw, int K,
                    int N, float* C, int64_t ldc, const int* c_idx, const float* mask, int64_t ldmask, int accumulate,
                    const RowRange rr[2], int64_t rows, hipStream_t s, int keep = -1, bool a16 = false) {
    GemmArgs a = {};
    a.ngroups = ngroups;
    a.N = N;
    a.K = K;
    a.ksplit = K;
    a.relu = 0;
    a.accumulate = accumulate;
    a.m_upper = rows;
    const bool bf16 = g_bwd_bf16 && rows > 0 && (size_t)ngroups * K * N <= g_wt_scratch_floats;
    if (g_bwd_bf16 && !bf16 && rows > 0) { set_error("backward (bf16): weight block %d x %d exceeds the transposition scratch", K, N); return MPNHIP_ERR_WORKSPACE; }
    for (int q = 0; q < ngroups; ++q) {
        GemmGroup& g = a.g[q];
        init_group(g);
        g.A = A;
        g.lda = lda;
        g.a16 = a16 ? 1 : 0;
        g.a_idx = a_idx;
        g.B = W[q];
        g.ldb = ldw;
        if (bf16) {
            // WT[n][k] = W[k][n]: the block as an nn.Linear weight of the product C = A WT^T (bf16 operands, K-contiguous)
            float* wt = g_wt_scratch + (size_t)q * K * N;
            WtKeep* kp = (keep >= 0 && ngroups == 1 && g_wt_keep[keep].wt && (size_t)K * N <= g_wt_keep[keep].floats) ? &g_wt_keep[keep] : nullptr;
            if (kp) wt = kp->wt;
            if (!kp || !kp->valid) MPN_TRY(transpose_padded(W[q], ldw, 0, K, N, wt, K, N, s));
            // the kept blocks also as bf16 rows (rounded once per backward instead of in every block of every step's product)
            // (N % 4: launch_gemm's bf16-row path needs whole 16-byte result vectors; otherwise the fp32 image on the register-staged path)
            const bool img16 = kp && kp->wt16 && K % 8 == 0 && N % 4 == 0 && ((size_t)K * N) % 4 == 0 && !getenv("MPNHIP_NO_GEMM_BF16_ROWS");
            if (img16 && !kp->valid) MPN_TRY(to_bf16_rows(wt, kp->wt16, (int64_t)K * N, s));
            if (kp) kp->valid = true;   // (one stream: the later steps' products are ordered behind this transposition)
            g.B = img16 ? reinterpret_cast<const float*>(kp->wt16) : wt;
            g.b16 = img16 ? 1 : 0;
            g.ldb = K;
        }
        g.C = C;
        g.ldc = ldc;
        g.c_idx = c_idx;
        g.mask = mask;
        g.ldmask = ldmask;
        g.m_static = rows;
        g.row_begin = rr ? rr[q].begin : nullptr;
        g.row_end = rr ? rr[q].end : nullptr;
    }
    if (bf16) {
        struct Scope { int old; Scope() : old(gemm_precision()) { set_gemm_precision(MPNHIP_PREC_BF16); } ~Scope() { set_gemm_precision(old); } } scope;
        return launch_gemm(a, A_KCONTIG, B_KCONTIG, s);
    }
    return launch_gemm(a, A_KCONTIG, B_NCONTIG, s);
}

// dZ_{i-1} = (dZ_i W_i) (.) [H_{i-1} > 0] for i = n-1 .. 1.  dz[i] / hidden[i]: this step's blocks.
static int mlp_chain_backward(const mpnhip_mlp& m0, const mpnhip_mlp* m1, float* const* dz, float* const* hidden,
                              const RowRange* rr, int64_t rows, hipStream_t s) {
    const int ng = m1 ? 2 : 1;
    for (int i = m0.n_layers - 1; i >= 1; --i) {
        const int n_out = m0.out_dims[i], k_in = m0.out_dims[i - 1];
        const float* Wq[2] = {m0.weight[i], m1 ? m1->weight[i] : nullptr};
        MPN_TRY(act_grad(ng, dz[i], n_out, nullptr, Wq, k_in, n_out, k_in, dz[i - 1], k_in, nullptr,
                         k_in != 1 ? hidden[i - 1] : nullptr, k_in, 0, rr, rows, s));
    }
    return MPNHIP_OK;
}

// batched weight gradients of layers >= 1 of an MLP whose dZ / hidden blocks repeat every step
static int mlp_weight_grads(const BwdPlan& p, float* slab_base, const mpnhip_mlp& m0, const mpnhip_mlp* m1, float* const* dz_all,
                            float* const* hidden0, int64_t hidden_bstride, const RowRange* rr, int64_t rows, int nbatch,
                            hipStream_t s) {
    for (int i = m0.n_layers - 1; i >= 1; --i) {
        const int n_out = m0.out_dims[i], k_in = m0.out_dims[i - 1];
        float* gw[2] = {m0.grad_weight[i], m1 ? m1->grad_weight[i] : nullptr};
        float* gb[2] = {m0.grad_bias[i], m1 ? m1->grad_bias[i] : nullptr};
        MPN_TRY(weight_grad(p, slab_base, m1 ? 2 : 1, {dz_all[i], n_out, rows * n_out}, nullptr, {hidden0[i - 1], k_in, hidden_bstride},
                            {nullptr, 0, 0}, k_in, nullptr, n_out, k_in, gw, k_in, gb, rr, rows, nbatch, s));
    }
    return MPNHIP_OK;
}

// encoder-style chain with ping-pong scratch (one batch): returns dZ of layer 0 in *dz
// ---- the reference's edge encoder (6 -> 18 -> 18 -> 16) backwards: the three activation gradients of an edge in one thread ------
//   dz2 = dE0 (.) [e0 > 0];   dz1 = (dz2 W2) (.) [H2 > 0];   dz0 = (dz1 W1) (.) [H1 > 0]
// (instead of a ReLU-mask kernel and two launches of the any-shape GEMM kernel; the weight-gradient products read dz2 / dz1 / dz0)
template <int H1W, int H2W, int OUTW>
__global__ __launch_bounds__(256) void k_edge_encoder_bwd(const float* __restrict__ dE0, const float* __restrict__ e0,
                                                          const float* __restrict__ h2, const float* __restrict__ h1,
                                                          const float* __restrict__ w2, const float* __restrict__ w1, int64_t rows,
                                                          float* __restrict__ dz2, float* __restrict__ dz1, float* __restrict__ dz0) {
    __shared__ float sw2[OUTW * H2W], sw1[H2W * H1W];
    for (int i = threadIdx.x; i < OUTW * H2W; i += 256) sw2[i] = w2[i];
    for (int i = threadIdx.x; i < H2W * H1W; i += 256) sw1[i] = w1[i];
    __syncthreads();
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= rows) return;
    float g2[OUTW], g1[H2W];
#pragma unroll
    for (int o = 0; o < OUTW; ++o) {
        g2[o] = e0[r * OUTW + o] > 0.f ? dE0[r * OUTW + o] : 0.f;
        dz2[r * OUTW + o] = g2[o];
    }
#pragma unroll
    for (int k = 0; k < H2W; ++k) {
        float s = 0.f;
#pragma unroll
        for (int o = 0; o < OUTW; ++o) s = fmaf(g2[o], sw2[o * H2W + k], s);
        g1[k] = h2[r * H2W + k] > 0.f ? s : 0.f;
        dz1[r * H2W + k] = g1[k];
    }
#pragma unroll
    for (int k = 0; k < H1W; ++k) {
        float s = 0.f;
#pragma unroll
        for (int o = 0; o < H2W; ++o) s = fmaf(g1[o], sw1[o * H1W + k], s);
        dz0[r * H1W + k] = h1[r * H1W + k] > 0.f ? s : 0.f;
    }
}

// bf16-operand training, deferred tail batch: the GEMM-shaped weight-gradient products of the node encoder ([1024 x 2048] over 20,000 rows
// at cfg-E: 128 output tiles of the row-panel kernel, every block walking all rows) read bf16 ROWS on the LDS-DMA kernel's 256 x 256
// tiles instead.  Their operands are fp32 blocks that stay put until the batch runs (BwdPlan::Tn, the saved hidden activations, the
// input features): the roundings are RECORDED here and launched on the tail's own stream just before its products.
struct Rows16Later {
    struct Item { const float* src; unsigned short* dst; int64_t n; };
    Item item[2 * MPNHIP_MAX_LAYERS];
    int n = 0;
    unsigned short* pool = nullptr;
    size_t pool_elems = 0, used = 0;
    unsigned short* take(const float* src, int64_t elems) {
        if (n >= 2 * MPNHIP_MAX_LAYERS || used + (size_t)elems > pool_elems || elems % 8 != 0 || (((uintptr_t)src) & 15) != 0) return nullptr;
        unsigned short* d = pool + used;
        item[n++] = {src, d, elems};
        used += (size_t)elems;
        return d;
    }
};
static thread_local Rows16Later* g_rows16_later = nullptr;

// one weight-gradient product of an encoder layer: over bf16 rows when a Rows16Later is open and the shape is one of the tiled ones
static int encoder_weight_grad(const BwdPlan& p, const float* dz, const float* h, int n_out, int k_in, float* gw0, float* gb0, int64_t rows,
                               hipStream_t s) {
    float* gw[2] = {gw0, nullptr};
    float* gb[2] = {gb0, nullptr};
    Rows16Later* L16 = g_rows16_later;
    int to = 1, tc = 1;
    if (L16 && wp_batch_open() && (int64_t)n_out * k_in >= 65536 && r16_variant(n_out, k_in, &to, &tc) >= 16 && L16->n + 2 <= 2 * MPNHIP_MAX_LAYERS &&
        L16->used + (size_t)rows * (n_out + k_in) <= L16->pool_elems) {
        const unsigned short* z16 = L16->take(dz, rows * n_out);
        const unsigned short* h16 = z16 ? L16->take(h, rows * k_in) : nullptr;
        if (z16 && h16) {
            Src16Scope rows16(true);
            return weight_grad(p, p.slab, 1, {reinterpret_cast<const float*>(z16), n_out, 0}, nullptr, {reinterpret_cast<const float*>(h16), k_in, 0},
                               {nullptr, 0, 0}, k_in, nullptr, n_out, k_in, gw, k_in, gb, nullptr, rows, 1, s);
        }
        if (z16) { --L16->n; L16->used -= (size_t)rows * n_out; }   // (the partner did not qualify: fp32 rows for both)
    }
    return weight_grad(p, p.slab, 1, {dz, n_out, 0}, nullptr, {h, k_in, 0}, {nullptr, 0, 0}, k_in, nullptr, n_out, k_in, gw, k_in, gb, nullptr, rows, 1, s);
}

static int mlp_tail_backward(const BwdPlan& p, float* const* T, const mpnhip_mlp& m0, float* const* hidden, const float** dz, int* cur_buf,
                             int64_t rows, hipStream_t s) {
    for (int i = m0.n_layers - 1; i >= 1; --i) {
        const int n_out = m0.out_dims[i], k_in = m0.out_dims[i - 1];
        MPN_TRY(encoder_weight_grad(p, *dz, hidden[i - 1], n_out, k_in, m0.grad_weight[i], m0.grad_bias[i], rows, s));
        const float* Wq[2] = {m0.weight[i], nullptr};
        float* dst = T[(*cur_buf + 1) % 3];
        MPN_TRY(act_grad(1, *dz, n_out, nullptr, Wq, k_in, n_out, k_in, dst, k_in, nullptr,
                         k_in != 1 ? hidden[i - 1] : nullptr, k_in, 0, nullptr, rows, s));
        *cur_buf = (*cur_buf + 1) % 3;
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

static bool backward_forks(const mpnhip_model& m) { return m.num_enc_steps >= 4 && !getenv("MPNHIP_NO_SIDE_STREAM"); }

extern "C" int mpnhip_backward_uses_side_stream(const mpnhip_model* model) { return model && backward_forks(*model) ? 1 : 0; }

extern "C" void* mpnhip_side_stream(void) {
    SideStream* side = nullptr;
    std::lock_guard<std::mutex> lock(g_side_mu);
    return side_stream_ready(&side) == MPNHIP_OK ? static_cast<void*>(side->stream) : nullptr;
}

extern "C" int mpnhip_side_stream_join(void* stream_) {
    SideStream* side = nullptr;
    std::lock_guard<std::mutex> lock(g_side_mu);
    MPN_TRY(side_stream_ready(&side));
    MPN_HIP(hipEventRecord(side->done, side->stream));
    MPN_HIP(hipStreamWaitEvent(static_cast<hipStream_t>(stream_), side->done, 0));
    return MPNHIP_OK;
}

extern "C" int mpnhip_backward(const mpnhip_model* model, const void* graph_buf, int n_nodes, int64_t n_edges,
                               const float* x, const float* edge_attr, const float* grad_logits, const float* grad_x_out,
                               const float* grad_e_out, float* grad_x, float* grad_edge_attr, void* fwd_workspace,
                               size_t fwd_workspace_bytes, void* bwd_workspace, size_t bwd_workspace_bytes, void* stream_) {
    return mpnhip_backward_flags(model, graph_buf, n_nodes, n_edges, x, edge_attr, grad_logits, grad_x_out, grad_e_out, grad_x,
                                 grad_edge_attr, fwd_workspace, fwd_workspace_bytes, bwd_workspace, bwd_workspace_bytes, 0, stream_);
}

extern "C" int mpnhip_backward_flags(const mpnhip_model* model, const void* graph_buf, int n_nodes, int64_t n_edges,
                                     const float* x, const float* edge_attr, const float* grad_logits, const float* grad_x_out,
                                     const float* grad_e_out, float* grad_x, float* grad_edge_attr, void* fwd_workspace,
                                     size_t fwd_workspace_bytes, void* bwd_workspace, size_t bwd_workspace_bytes, int flags,
                                     void* stream_) {
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
    struct WgradScope {
        bool old;
        explicit WgradScope(bool v) : old(g_wgrad_split) { g_wgrad_split = v; }
        ~WgradScope() { g_wgrad_split = old; }
    } wgrad_scope((m.precision == MPNHIP_PREC_FP32_SPLIT || m.precision == MPNHIP_PREC_FP32_WGSPLIT || m.precision == MPNHIP_PREC_BF16) &&
                  !getenv("MPNHIP_NO_WGRAD_PANEL"));
    struct Bf16Scope {
        bool old; float* olds; size_t oldn;
        Bf16Scope(bool v) : old(g_bwd_bf16), olds(g_wt_scratch), oldn(g_wt_scratch_floats) { g_bwd_bf16 = v; }
        ~Bf16Scope() { g_bwd_bf16 = old; g_wt_scratch = olds; g_wt_scratch_floats = oldn; }
    } bf16_scope(m.precision == MPNHIP_PREC_BF16);
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
    g_wt_scratch = p.wt_scratch;
    g_wt_scratch_floats = p.wt_scratch_floats;
    struct KeepScope {   // (valid only inside this call: the blocks live in this call's workspace)
        KeepScope(const BwdPlan& q) { for (int i = 0; i < 2; ++i) g_wt_keep[i] = {q.wt_keep_floats[i] ? q.wt_keep[i] : nullptr, q.wt_keep_floats[i], false, q.wt_keep_floats[i] ? q.wt_keep16[i] : nullptr}; }
        ~KeepScope() { for (int i = 0; i < 2; ++i) g_wt_keep[i] = {nullptr, 0, false, nullptr}; }
    } keep_scope(p);
    if (getenv("MPNHIP_NO_WT_KEEP")) for (int i = 0; i < 2; ++i) g_wt_keep[i].wt = nullptr;
    GraphView g;
    graph_layout(n_nodes, n_edges, &g, const_cast<void*>(graph_buf));
    const int he = d.he, hn = d.hn, dn = d.dn, de = d.de, kx = d.kx, ke = d.ke, pw = d.pw, L = d.L;
    const size_t xs = (size_t)N * dn, es = (size_t)E * de;
    const RowRange dir_rr[2] = {{nullptr, g.header + 4}, {g.header + 4, g.header + 5}};
    const mpnhip_mlp& cls = m.classifier;
    const int ne = m.edge.n_layers, nfl = m.flow_in.n_layers, nc = cls.n_layers;
    const int64_t sstride = (int64_t)(f.step_stride_bytes / sizeof(float));  // floats between two steps' saved activations

    // ---- seeds ----------------------------------------------------------------------------------
    int cx = 0;
    // zeros: dX[0] (unless the caller seeds it) | dX0 | dE0 | gWnode, contiguous in the plan -- and, below, the seed of the gradient
    // w.r.t. e_L: ONE launch of a plain kernel for both (hipMemsetAsync's fill was preceded by ~18 us of idle queue in every step
    // of the cfg-D timeline; two of them opened every backward)
    char* const z0 = reinterpret_cast<char*>(grad_x_out ? p.dX0 : p.dX[0]);
    const size_t z0_bytes = (size_t)(reinterpret_cast<char*>(p.zero_end) - z0);
    if (grad_x_out) MPN_HIP(hipMemcpyAsync(p.dX[0], grad_x_out, xs * 4, hipMemcpyDeviceToDevice, s));
    // the gradient w.r.t. e_s lives in the LAST edge-layer dZ block of step s (it becomes that dZ once masked)
    if (f.b16 != p.b16) { set_error("backward: the forward workspace was saved in another mode (bf16 fused training %d vs %d)", (int)f.b16, (int)p.b16); return MPNHIP_ERR_ARG; }
    const bool use_b16 = p.b16 && E > 0 && L > 0;   // bf16-operand training on the fused kernels (plan.h: chain_bf16_train_ok)
    int ce = 0;                                      // ... the gradient w.r.t. e_s travels in p.dEpp[ce]
    float* dE_last = use_b16 ? p.dEpp[0] : (L > 0 ? p.dZed[ne - 1] + (size_t)(L - 1) * es : p.dE0);
    {
        bool zero_e = !(grad_e_out && es) && es && L > 0;
        if (zero_e && ((((uintptr_t)dE_last) & 15) != 0 || (es * 4) % 16 != 0)) {   // (odd E de: a block inside the dZ array is not 16-byte aligned)
            MPN_HIP(hipMemsetAsync(dE_last, 0, es * 4, s));
            zero_e = false;
        }
        const int64_t n0 = (int64_t)(z0_bytes / 16), n1 = zero_e ? (int64_t)(es * 4 / 16) : 0;   // (arena blocks: 256-byte multiples)
        if (n0 + n1 > 0) {
            int64_t blocks = (n0 + n1 + 255) / 256;
            blocks = blocks > 4096 ? 4096 : blocks;
            hipLaunchKernelGGL(k_zero_pair, dim3((unsigned)blocks), dim3(256), 0, s, reinterpret_cast<uint4*>(z0), n0, reinterpret_cast<uint4*>(dE_last), n1);
            MPN_LAUNCH_CHECK();
        }
    }
    if (grad_e_out && es) {
        hipLaunchKernelGGL(k_gather_rows, dim3((unsigned)((es + 255) / 256)), dim3(256), 0, s, grad_e_out, g.perm, dE_last, E, de);
        MPN_LAUNCH_CHECK();
    }
    const float* x0 = f.x_hist;
    const float* e0 = f.e_hist;
    if (use_b16) {
        const float* f0[2] = {m.flow_out.weight[0], m.flow_in.weight[0]};
        const float* f1[2] = {m.flow_out.weight[1], m.flow_in.weight[1]};
        MPN_TRY(pack_chain_bf16_bwd(m.edge.weight[0], m.edge.in_dim, 2 * kx + de, m.edge.weight[1], m.classifier.weight[0], f0, m.flow_out.in_dim, kx, f1,
                                    he, de, hn, dn, m.classifier.out_dims[0], p.cb16_img, s));
    }
    const bool use_chain = chain_shapes_ok(m, d) && E > 0 && L > 0 && !getenv("MPNHIP_NO_CHAIN_BWD");
    const bool bwd_split = use_chain && chain_split(m);
    if (use_chain) {
        const int HE = pad32(he), DE = pad32(de), HN = pad32(hn), DN = pad32(dn);
        const int KEp = d.ef * DE;
        const mpnhip_mlp* fl[2] = {&m.flow_out, &m.flow_in};
        const int ncol6 = KEp < 64 ? KEp : 64;
        if (bwd_split) {
            // the same logical images (rows = contraction index n, columns = k) as split images (edge_chain.hip, split8);
            // recorded and packed by one launch (pack_split cannot fail once its sizes are valid: nothing returns before the flush)
            SplitBatch sb;
            SplitBatchGuard sbg;
            split_batch_begin(&sb);
            for (int q = 0; q < 2; ++q) {
                MPN_TRY(pack_split(fl[q]->weight[1], hn, 1, dn, hn, DN, HN, p.wf2p[q], s));
                MPN_TRY(pack_split(fl[q]->weight[0] + kx, fl[q]->in_dim, 1, hn, de, HN, DE, p.wfep[q], s));
            }
            MPN_TRY(pack_split(m.classifier.weight[0], de, 1, m.classifier.out_dims[0], de, 32, DE, p.wc1p, s));
            MPN_TRY(pack_split(m.edge.weight[1], he, 1, de, he, DE, HE, p.w2p, s));
            for (int hlf = 0; hlf < d.ef; ++hlf) {
                const int col0 = hlf * DE;
                MPN_TRY(pack_split(m.edge.weight[0] + 2 * kx + hlf * de, m.edge.in_dim, 1, he, de, HE, DE,
                                   p.w1ep + (int64_t)(col0 / 64) * HE * ncol6 * 3 / 2, s, ncol6 / 32, (col0 % 64) / 32));
            }
            MPN_TRY(split_batch_flush(s));
        } else {
            PackBatch pb;
            PackBatchGuard pbg;
            pack_batch_begin(&pb);   // (the eight images as one launch)
            for (int q = 0; q < 2; ++q) {
                MPN_TRY(pack_padded(fl[q]->weight[1], hn, 0, dn, hn, p.wf2p[q], DN, HN, HN, 0, s));
                MPN_TRY(pack_padded(fl[q]->weight[0], fl[q]->in_dim, kx, hn, de, p.wfep[q], HN, DE, DE, 0, s));
            }
            MPN_TRY(pack_padded(m.classifier.weight[0], de, 0, m.classifier.out_dims[0], de, p.wc1p, 32, DE, DE, 0, s));
            MPN_TRY(pack_padded(m.edge.weight[1], he, 0, de, he, p.w2p, DE, HE, HE, 0, s));
            // e-part columns of edge layer 0 as one image [HE][ncol6] per pass of <= 64 (padded) columns of [e0 | e_{s-1}]
            // (the kernel streams whole rows of one pass image)
            for (int hlf = 0; hlf < d.ef; ++hlf) {
                const int col0 = hlf * DE;
                MPN_TRY(pack_padded(m.edge.weight[0], m.edge.in_dim, 2 * kx + hlf * de, he, de,
                                    p.w1ep + (int64_t)(col0 / 64) * HE * ncol6, HE, DE, ncol6, col0 % 64, s));
            }
            MPN_TRY(pack_batch_flush(s));
        }
    }

    // activation-gradient chain of the classifier for one step: dz chain blocks in dzc[], final product
    // accumulated into dEdst and masked by `mask` (ReLU of the edge layer that produced ef)
    auto classifier_chain = [&](float* const* HC, float* const* dzc, const float* dlog, float* dEdst, const float* mask) -> int {
        if (E == 0) return MPNHIP_OK;
        for (int i = nc - 1; i >= 0; --i) {
            const int n_out = cls.out_dims[i], k_in = i == 0 ? de : cls.out_dims[i - 1];
            const bool top = i == nc - 1;  // dZ of the last layer is grad_logits, [E, 1] in ORIGINAL order
            const float* A = top ? dlog : dzc[i];
            const float* Wq[2] = {cls.weight[i], nullptr};
            if (i == 0)
                MPN_TRY(act_grad(1, A, top ? 1 : n_out, top ? g.perm : nullptr, Wq, k_in, n_out, k_in, dEdst, de, nullptr, mask, de,
                                 1, nullptr, E, s));
            else
                MPN_TRY(act_grad(1, A, top ? 1 : n_out, top ? g.perm : nullptr, Wq, k_in, n_out, k_in, dzc[i - 1], k_in, nullptr,
                                 k_in != 1 ? HC[i - 1] : nullptr, k_in, 0, nullptr, E, s));
        }
        return MPNHIP_OK;
    };

    // The re-attached x0 does not change from step to step: its share of dX (and of the packed projection weight's gradient)
    // comes from the SUM of the steps' dP, once, after the loop
    const bool hoist_x = d.nf == 2 && L > 1 && N > 0 && pw % 4 == 0 && dn % 4 == 0 && !getenv("MPNHIP_NO_DX0_HOIST");
    // The same for the re-attached e0 on the edge side (fused chain, whole 64-column passes: 32 < de <= 64): the gradient w.r.t. e0
    // through the edge MLP's first layer and the e0 columns of that layer's weight gradient are ONE product each with
    // S = sum_s dZ1_s after the loop -- 2 E he de MACs less in every step's backward chain and in every step's weight gradient
    // (2 x 2.05 of ~31 GFLOP per step at cfg-B), for one pass over the kept dZ1 blocks.
    // (the bf16 backward chain kernel never contracts the e0 columns: always hoisted there, also at L = 1)
    const bool hoist_e0 = use_b16 || (use_chain && d.ef == 2 && L > 1 && pad32(de) == 64 && he % 4 == 0 && de % 4 == 0 && !getenv("MPNHIP_NO_DE0_HOIST"));
    // bf16-operand training: the per-node projections' weight gradient over bf16 ROWS -- dP_s rounded once by the scatter-add kernel that
    // produces it, x_{s-1} from the forward's bf16 mirror -- on the LDS-DMA kernel in 256 x 256 output tiles (wgrad_rows16.hip)
    // (f.xb_hist is filled under the forward's own run-time test: plan.h node_rows16_runtime, the one definition both sides use)
    const bool node16 = use_b16 && hoist_x && p.dP16 && node_rows16_runtime(f, m, d) && !getenv("MPNHIP_NO_NODE_ROWS16");
    // bf16 rows: pointer `elems` unsigned shorts into a buffer the plans type as float*
    auto u16 = [](const float* p0, int64_t elems) { return reinterpret_cast<const float*>(reinterpret_cast<const unsigned short*>(p0) + elems); };
    // Weight gradients of the message-passing modules for steps b0+1 .. b0+nb: ONE batched split-row product per
    // weight (batch index = step - 1).  Issued on `st` with the slab buffer `slab`.
    auto mp_weight_grads = [&](int b0, int nb, hipStream_t st, float* slab) -> int {
        if (nb <= 0) return MPNHIP_OK;
        const int64_t zb = b0;  // first batch
        // MPNHIP_PREC_FP32_SPLIT: the products below are recorded and run as ONE product launch + ONE slab-sum launch at the end
        // (calls are serialised on `st`, so successive groups may share the slab region)
        WpBatch wpb;
        WpBatchGuard wpg;
        if (g_wgrad_split) { wp_batch_begin(&wpb, p.slab_wp, p.slab_wp_floats, true); wp_batch_set_stream(st); }
        auto finish = [&]() -> int { return wp_batch_open() ? wp_batch_flush(st) : MPNHIP_OK; };
        {   // node update Linear
            float* gw[2] = {m.node.grad_weight[0], nullptr};
            float* gb[2] = {m.node.grad_bias[0], nullptr};
            if (node16 && p.dZn16 && sstride % 4 == 0) {
                // bf16 rows (rounded here, on the products' own stream, ahead of the recorded batch): [dn x 2 dn] = two column tiles of
                // the LDS-DMA kernel instead of 2 x 4 tiles of the row-panel kernel over fp32 rows
                MPN_TRY(to_bf16_rows(p.dZn + zb * xs, p.dZn16 + zb * xs, (int64_t)nb * xs, st));
                for (int q = 0; q < nb; ++q)
                    MPN_TRY(to_bf16_rows(f.step0.AGG + (zb + q) * sstride, p.AGG16 + (zb + q) * N * 2 * dn, N * 2 * dn, st));
                Src16Scope rows16(true);
                MPN_TRY(weight_grad(p, slab, 1, {reinterpret_cast<const float*>(p.dZn16 + zb * xs), dn, (int64_t)xs}, nullptr,
                                    {reinterpret_cast<const float*>(p.AGG16 + zb * N * 2 * dn), 2 * dn, N * 2 * dn}, {nullptr, 0, 0}, 2 * dn, nullptr, dn,
                                    2 * dn, gw, 2 * dn, gb, nullptr, N, nb, st));
            } else {
                MPN_TRY(weight_grad(p, slab, 1, {p.dZn + zb * xs, dn, (int64_t)xs}, nullptr, {f.step0.AGG + zb * sstride, 2 * dn, sstride},
                                    {nullptr, 0, 0}, 2 * dn, nullptr, dn, 2 * dn, gw, 2 * dn, gb, nullptr, N, nb, st));
            }
        }
        if (use_b16) {
            // every operand of these products is a bf16 row block: the backward chain kernel's dZ outputs, the forward chain kernel's
            // saved activations and the bf16 mirror of e_hist (WpProduct::src16: loaded as they are, one product per k block)
            Src16Scope src16(true);
            const int hcw = cls.out_dims[0];
            const int64_t ss16 = 2 * sstride;   // unsigned shorts between two steps' saved activations
            const float* dZF = u16(p.dZfl[0], zb * E * hn); const float* dZM = u16(p.dZfl[1], zb * E * dn);
            const float* dZ1 = u16(p.dZed[0], zb * E * he); const float* dZ2 = u16(p.dZed[1], zb * E * de);
            const float* dZc = u16(p.dZcl[0], zb * E * hcw);
            const float* H1 = u16(f.step0.HE[0], zb * ss16); const float* HFb = u16(f.step0.HF[0], zb * ss16);
            const float* HCb = u16(f.step0.HC[0], zb * ss16);
            const float* eb_new = reinterpret_cast<const float*>(f.eb_hist + es * (zb + 1));
            const float* eb_prev = reinterpret_cast<const float*>(f.eb_hist + es * zb);
            const Operand none = {nullptr, 0, 0};
            {   // flow layer 1, both directions
                float* gw[2] = {m.flow_out.grad_weight[1], m.flow_in.grad_weight[1]};
                float* gb[2] = {m.flow_out.grad_bias[1], m.flow_in.grad_bias[1]};
                MPN_TRY(weight_grad(p, slab, 2, {dZM, dn, (int64_t)E * dn}, nullptr, {HFb, hn, ss16}, none, hn, nullptr, dn, hn, gw, hn, gb, dir_rr, E, nb, st));
            }
            {   // flow layer 0: e' columns [kx, kx + de) and the bias (folded into P in the forward)
                float* gw[2] = {m.flow_out.grad_weight[0] + kx, m.flow_in.grad_weight[0] + kx};
                float* gb[2] = {m.flow_out.grad_bias[0], m.flow_in.grad_bias[0]};
                MPN_TRY(weight_grad(p, slab, 2, {dZF, hn, (int64_t)E * hn}, nullptr, {eb_new, de, (int64_t)es}, none, de, nullptr, hn, de, gw, m.flow_out.in_dim, gb,
                                    dir_rr, E, nb, st));
            }
            {   // classifier output layer [1 x hc]: dZ = grad_logits (fp32, original order -> perm), H = HC rows (bf16)
                float* gw[2] = {cls.grad_weight[1], nullptr};
                float* gb[2] = {cls.grad_bias[1], nullptr};
                MPN_TRY(weight_grad(p, slab, 1, {grad_logits + zb * E, 1, (int64_t)E}, g.perm, {HCb, hcw, ss16}, none, hcw, nullptr, 1, hcw, gw, hcw, gb, nullptr, E,
                                    nb, st));
            }
            {   // classifier layer 0
                float* gw[2] = {cls.grad_weight[0], nullptr};
                float* gb[2] = {cls.grad_bias[0], nullptr};
                MPN_TRY(weight_grad(p, slab, 1, {dZc, hcw, (int64_t)E * hcw}, nullptr, {eb_new, de, (int64_t)es}, none, de, nullptr, hcw, de, gw, de, gb, nullptr, E, nb,
                                    st));
            }
            {   // edge layer 1
                float* gw[2] = {m.edge.grad_weight[1], nullptr};
                float* gb[2] = {m.edge.grad_bias[1], nullptr};
                MPN_TRY(weight_grad(p, slab, 1, {dZ2, de, (int64_t)E * de}, nullptr, {H1, he, ss16}, none, he, nullptr, de, he, gw, he, gb, nullptr, E, nb, st));
            }
            {   // edge layer 0: the e_{s-1} columns and the bias (the e0 columns: one product with the summed dZ1 after the loop)
                float* gw[2] = {m.edge.grad_weight[0] + 2 * kx + de, nullptr};
                float* gb[2] = {m.edge.grad_bias[0], nullptr};
                MPN_TRY(weight_grad(p, slab, 1, {dZ1, he, (int64_t)E * he}, nullptr, {eb_prev, de, (int64_t)es}, none, de, nullptr, he, de, gw, m.edge.in_dim, gb,
                                    nullptr, E, nb, st));
            }
        } else if (E > 0) {
            float* dzfl_b[MPNHIP_MAX_LAYERS];
            float* dzed_b[MPNHIP_MAX_LAYERS];
            float* hf_b[MPNHIP_MAX_LAYERS];
            float* he_b[MPNHIP_MAX_LAYERS];
            for (int i = 0; i < nfl; ++i) { dzfl_b[i] = p.dZfl[i] + zb * E * m.flow_in.out_dims[i]; hf_b[i] = f.step0.HF[i] ? f.step0.HF[i] + zb * sstride : nullptr; }
            for (int i = 0; i < ne; ++i) { dzed_b[i] = p.dZed[i] + zb * E * m.edge.out_dims[i]; he_b[i] = f.step0.HE[i] ? f.step0.HE[i] + zb * sstride : nullptr; }
            MPN_TRY(mlp_weight_grads(p, slab, m.flow_out, &m.flow_in, dzfl_b, hf_b, sstride, dir_rr, E, nb, st));
            {   // flow layer 0: e-part columns [kx, kx + de) and the bias (folded into P in the forward)
                float* gw[2] = {m.flow_out.grad_weight[0] + kx, m.flow_in.grad_weight[0] + kx};
                float* gb[2] = {m.flow_out.grad_bias[0], m.flow_in.grad_bias[0]};
                MPN_TRY(weight_grad(p, slab, 2, {dzfl_b[0], hn, (int64_t)E * hn}, nullptr, {f.e_hist + es * (zb + 1), de, (int64_t)es},
                                    {nullptr, 0, 0}, de, nullptr, hn, de, gw, m.flow_out.in_dim, gb, dir_rr, E, nb, st));
            }
            // classifier: dZ of the last layer is grad_logits ([L, E], original order -> perm)
            for (int i = nc - 1; i >= 0; --i) {
                const int n_out = cls.out_dims[i], k_in = i == 0 ? de : cls.out_dims[i - 1];
                const bool top = i == nc - 1;
                float* gw[2] = {cls.grad_weight[i], nullptr};
                float* gb[2] = {cls.grad_bias[i], nullptr};
                Operand dz = top ? Operand{grad_logits + zb * E, 1, (int64_t)E} : Operand{p.dZcl[i] + zb * E * n_out, n_out, (int64_t)E * n_out};
                Operand h = i == 0 ? Operand{f.e_hist + es * (zb + 1), de, (int64_t)es} : Operand{f.step0.HC[i - 1] + zb * sstride, k_in, sstride};
                MPN_TRY(weight_grad(p, slab, 1, dz, top ? g.perm : nullptr, h, {nullptr, 0, 0}, k_in, nullptr, n_out, k_in, gw, k_in, gb,
                                    nullptr, E, nb, st));
            }
            MPN_TRY(mlp_weight_grads(p, slab, m.edge, nullptr, dzed_b, he_b, sstride, nullptr, E, nb, st));
            if (hoist_e0) {
                // edge layer 0: the e_{s-1} columns [2kx + de, 2kx + 2 de) and the bias; the e0 columns follow after the loop
                float* gw[2] = {m.edge.grad_weight[0] + 2 * kx + de, nullptr};
                float* gb[2] = {m.edge.grad_bias[0], nullptr};
                MPN_TRY(weight_grad(p, slab, 1, {dzed_b[0], he, (int64_t)E * he}, nullptr, {f.e_hist + es * zb, de, (int64_t)es},
                                    {nullptr, 0, 0}, de, nullptr, he, de, gw, m.edge.in_dim, gb, nullptr, E, nb, st));
            } else {   // edge layer 0: e-part columns [2kx, 2kx + ke) = [e0 | e_{s-1}], and the bias
                float* gw[2] = {m.edge.grad_weight[0] + 2 * kx, nullptr};
                float* gb[2] = {m.edge.grad_bias[0], nullptr};
                const bool two = d.ef == 2;
                Operand h1 = two ? Operand{e0, de, 0} : Operand{f.e_hist + es * zb, de, (int64_t)es};
                Operand h2 = two ? Operand{f.e_hist + es * zb, de, (int64_t)es} : Operand{nullptr, 0, 0};
                MPN_TRY(weight_grad(p, slab, 1, {dzed_b[0], he, (int64_t)E * he}, nullptr, h1, h2, de, nullptr, he, ke, gw, m.edge.in_dim,
                                    gb, nullptr, E, nb, st));
            }
        }
        {   // per-node projections (packed), accumulated into gWnode
            if (hoist_x) {
                // columns [dn, 2 dn) (the current features) here; the x0 columns are one product with the summed dP after the loop
                float* gw[2] = {p.gWnode + dn, nullptr};
                if (node16) {
                    Src16Scope rows16(true);
                    MPN_TRY(weight_grad(p, slab, 1, {reinterpret_cast<const float*>(p.dP16 + zb * N * pw), pw, (int64_t)N * pw}, nullptr,
                                        {reinterpret_cast<const float*>(f.xb_hist + xs * zb), dn, (int64_t)xs}, {nullptr, 0, 0}, dn, nullptr, pw, dn, gw, kx,
                                        nullptr, nullptr, N, nb, st));
                    return finish();
                }
                MPN_TRY(weight_grad(p, slab, 1, {p.dP + zb * N * pw, pw, (int64_t)N * pw}, nullptr, {f.x_hist + xs * zb, dn, (int64_t)xs},
                                    {nullptr, 0, 0}, dn, nullptr, pw, dn, gw, kx, nullptr, nullptr, N, nb, st));
                return finish();
            }
            float* gw[2] = {p.gWnode, nullptr};
            const bool two = d.nf == 2;
            Operand h1 = two ? Operand{x0, dn, 0} : Operand{f.x_hist + xs * zb, dn, (int64_t)xs};
            Operand h2 = two ? Operand{f.x_hist + xs * zb, dn, (int64_t)xs} : Operand{nullptr, 0, 0};
            MPN_TRY(weight_grad(p, slab, 1, {p.dP + zb * N * pw, pw, (int64_t)N * pw}, nullptr, h1, h2, dn, nullptr, pw, kx, gw, kx, nullptr,
                                nullptr, N, nb, st));
        }
        return finish();
    };
    // The steps are finished from L down to 1.  Their weight gradients go in groups: as soon as a group's steps are done
    // its batched products run on a side stream UNDER the remaining steps' chain kernels (which leave 120 of the 256
    // CUs idle for the last third of their run at cfg-B); only the last group runs after the loop.  Groups are issued
    // to ONE side stream in order, so they can share its slab buffer and their "+=" into the gradients stay ordered.
    SideStream* side = nullptr;
    const bool want_fork = backward_forks(m) && side_stream_ready(&side) == MPNHIP_OK;
    std::unique_lock<std::mutex> side_lock(g_side_mu, std::defer_lock);
    if (want_fork) side_lock.lock();
    SideJoin join;
    join.ss = side;
    join.caller = s;
    // Group sizes (in steps, first-finished group first).  The side stream is serial, so the last group should be the
    // smallest: it starts only when the loop is over and what it has not finished when the encoder's backward ends is
    // exposed.  Default for L = 12: 5 + 4 + 3; in general three groups with sizes ~ (5 : 4 : 3), two below six steps or when
    // the products are small.
    int gsize[8] = {0};
    int ngroups = 1;
    if (want_fork) {
        static const char* spec = getenv("MPNHIP_WGRAD_SPLIT");  // tuning override, e.g. "4,4,4"
        int parsed = 0, total = 0;
        if (spec) {
            const char* q = spec;
            while (*q && parsed < 8) {
                gsize[parsed] = atoi(q);
                total += gsize[parsed] > 0 ? gsize[parsed] : 0;
                ++parsed;
                while (*q && *q != ',') ++q;
                if (*q == ',') ++q;
            }
        }
        bool ok = parsed >= 2 && total == (int)L;
        for (int i = 0; i < parsed; ++i) ok = ok && gsize[i] > 0;
        if (ok) {
            ngroups = parsed;
        } else if (L >= 6 && (double)E * dn * dn >= 3e8 && !g_wgrad_split) {
            // (enough work per group to pay for a third round of ~30 launches: cfg-B 8e8; the reference's 32-d widths, 8e7 at
            // cfg-C, do better with two groups -- measured 2.59 -> 2.48 ms.  The row-panel products of MPNHIP_PREC_FP32_SPLIT are
            // two launches per group whatever its size, and a larger group needs fewer slabs per row: two groups there,
            // cfg-B 5.27 -> 5.23 ms)
            ngroups = 3;
            gsize[0] = (int)((5 * L + 6) / 12);
            gsize[2] = (int)(L / 4);
            gsize[1] = (int)L - gsize[0] - gsize[2];
        } else {
            ngroups = 2;
            gsize[0] = (int)(L - L / 2);
            gsize[1] = (int)(L / 2);
        }
    } else {
        gsize[0] = (int)L;
    }
    // group g (g = 0 is finished first) covers batches [glo(g), glo(g - 1))
    auto glo = [&](int g) {
        int lo = (int)L;
        for (int i = 0; i <= g && i < ngroups; ++i) lo -= gsize[i];
        return g < 0 ? (int)L : lo;
    };
    int next_group = 0;
    bool& forked = join.forked;
    bool dx_split = false;   // the gradient w.r.t. x_s arrived as two K halves (p.dX[cx] + p.dXh)
    bool node_a_done = false;   // dZn / dAGG of the coming step were already produced by node_step32_bwd
    // (k_node_step32_bwd stages the weights in LDS: 128 pw + 8 KB + ... <= 64 KB, 16-byte aligned rows)
    // wider models in the split precision: the same three launches as one MFMA kernel (node_chain.hip, node_chain_bwd_kernel)
    const bool fuse_node_chain_bwd = p.ncb_img && !g_bwd_bf16 && N > 0 && L > 1 && d.nf == 2;
    size_t ncb_off_wu = 0;
    if (fuse_node_chain_bwd) {
        node_chain_bwd_image_shorts(dn, pw, &ncb_off_wu);
        MPN_TRY(pack_node_chain_bwd(m.node.weight[0], f.Wnode, dn, pw, kx, p.ncb_img, s));
    }
    const bool fuse_node_bwd = !g_bwd_bf16 && dn == 32 && N <= 4096 && pw <= 384 && kx % 4 == 0 &&
                               ((((uintptr_t)f.Wnode) | ((uintptr_t)m.node.weight[0])) & 15) == 0 && !getenv("MPNHIP_NO_NODE_FUSION");

    for (int step = L; step >= 1; --step) {
        const int b_ = step - 1;  // batch (block) index of this step
        const StepBufs b = step_at(f, b_);
        const float* x_s = f.x_hist + xs * step;
        const float* e_s = f.e_hist + es * step;
        float* dXc = p.dX[cx];
        float* dZn = p.dZn + (size_t)b_ * xs;
        float* dP = p.dP + (size_t)b_ * N * pw;
        float* dzfl[MPNHIP_MAX_LAYERS];
        float* dzed[MPNHIP_MAX_LAYERS];
        float* dzcl[MPNHIP_MAX_LAYERS];
        for (int i = 0; i < nfl; ++i) dzfl[i] = p.dZfl[i] + (size_t)b_ * E * m.flow_in.out_dims[i];
        for (int i = 0; i < ne; ++i) dzed[i] = p.dZed[i] + (size_t)b_ * E * m.edge.out_dims[i];
        for (int i = 0; i + 1 < nc; ++i) dzcl[i] = p.dZcl[i] + (size_t)b_ * E * cls.out_dims[i];
        float* dEc = dzed[ne - 1];  // gradient w.r.t. e_s (seeded by the later step / the caller)

        // ---- A. node update  x_s = relu(AGG W^T + b)  (mpn.py:97-99) ---------------------------
        if (!node_a_done) {
            MPN_TRY(relu_mask(dXc, x_s, dZn, (int64_t)xs, s, dx_split ? p.dXh : nullptr));
            const float* Wq[2] = {m.node.weight[0], nullptr};
            MPN_TRY(act_grad(1, dZn, dn, nullptr, Wq, 2 * dn, dn, 2 * dn, p.dAGG, 2 * dn, nullptr, nullptr, 0, 0, nullptr, N, s, 0));
        }
        dx_split = false;
        node_a_done = false;
        if (use_b16) {
            // ---- B-E fused, bf16 operands: the mirror of the forward chain kernel (edge_chain_bf16_bwd.hip) ------------------
            const int hcw = cls.out_dims[0];
            auto w16 = [&](float* p0, int64_t elems) { return reinterpret_cast<unsigned short*>(p0) + elems; };
            EdgeChainBf16BwdArgs a = {};
            a.E = (int)E; a.N = (int)N; a.agg = m.agg; a.first_step = step == 1 ? 1 : 0;
            a.he = he; a.de = de; a.hn = hn; a.dn = dn; a.hc = hcw;
            a.header = g.header; a.srow = g.srow; a.perm = g.perm; a.seg_ptr = g.seg_ptr;
            a.dAGG = p.dAGG; a.mask = reinterpret_cast<const unsigned*>(b.MK); a.ARG = b.ARG;
            a.dlog = grad_logits + (size_t)b_ * E;
            a.dE_in = p.dEpp[ce];
            a.dZM = w16(p.dZfl[1], (int64_t)b_ * E * dn); a.dZF = w16(p.dZfl[0], (int64_t)b_ * E * hn);
            a.dZc = w16(p.dZcl[0], (int64_t)b_ * E * hcw); a.dZ2 = w16(p.dZed[1], (int64_t)b_ * E * de);
            a.dZ1 = w16(p.dZed[0], (int64_t)b_ * E * he);
            a.dE0 = p.dE0; a.dEprev = p.dEpp[ce ^ 1];
            a.img_edge = p.cb16_img; a.img_cls = p.cb16_img + p.cb16_off_cls;
            a.img_flow[0] = p.cb16_img + p.cb16_off_flow[0]; a.img_flow[1] = p.cb16_img + p.cb16_off_flow[1];
            a.wc2 = cls.weight[1];
            prof_begin(PROF_CHAIN_BWD, s);
            MPN_TRY(launch_edge_chain_bf16_bwd(a, s));
            prof_end(PROF_CHAIN_BWD, s);
            ce ^= 1;
            // index_put_(accumulate) of the gathers x[flow_col] (mpn.py:87,93) and x[row], x[col] (mpn.py:69) over the bf16 dZ rows
            const float* zf = reinterpret_cast<const float*>(a.dZF);
            const float* z1 = reinterpret_cast<const float*>(a.dZ1);
            unsigned short* const dP16 = node16 ? p.dP16 + (size_t)b_ * N * pw : nullptr;
            const SegReduce2 c3[3] = {{zf, hn, g.cperm, g.cseg_ptr, 2 * (int)N, hn, dP, pw, (int)N, 2 * he, 2 * he + hn, 0, 0, dP16, pw},
                                      {z1, he, nullptr, g.seg_ptr, (int)N, he, dP, pw, (int)N, 0, 0, 3, (int)N, dP16, pw},
                                      {z1, he, g.cperm_all, g.cseg_all, (int)N, he, dP, pw, (int)N, he, he, 0, 0, dP16, pw}};
            MPN_TRY(segment_reduce_csr2_x3_bf16(c3, s));
        } else if (use_chain) {
            // ---- B-E fused: every activation-gradient product of the per-edge modules in one kernel --------
            EdgeChainBwdArgs a = {};
            a.E = (int)E; a.N = (int)N; a.agg = m.agg; a.first_step = step == 1 ? 1 : 0; a.cat_two = d.ef == 2 ? 1 : 0; a.split = bwd_split ? 1 : 0;
            a.skip_e0 = hoist_e0 ? 1 : 0;
            a.he = he; a.de = de; a.hn = hn; a.dn = dn; a.hc = cls.out_dims[0];
            a.header = g.header; a.srow = g.srow; a.perm = g.perm; a.seg_ptr = g.seg_ptr;
            a.dAGG = p.dAGG; a.mask = reinterpret_cast<const unsigned*>(b.MK); a.ARG = b.ARG;
            a.dlog = grad_logits + (size_t)b_ * E;
            a.dE_io = dzed[1]; a.dZM = dzfl[1]; a.dZF = dzfl[0]; a.dZc = dzcl[0]; a.dZ1 = dzed[0];
            a.dE0 = p.dE0; a.dEprev = step == 1 ? p.dE0 : p.dZed[ne - 1] + (size_t)(b_ - 1) * es;
            a.wf2_out = p.wf2p[0]; a.wf2_in = p.wf2p[1]; a.wfe_out = p.wfep[0]; a.wfe_in = p.wfep[1];
            a.wc1 = p.wc1p; a.wc2 = cls.weight[1]; a.w2 = p.w2p; a.w1e = p.w1ep;
            prof_begin(PROF_CHAIN_BWD, s);
            MPN_TRY(launch_edge_chain_bwd(a, s));
            prof_end(PROF_CHAIN_BWD, s);
            // index_put_(accumulate) of the gathers x[flow_col] (mpn.py:87,93) and x[row], x[col] (mpn.py:69)
            if (E >= 48 * N) {
                // dense graphs (long segments): the three reductions as ONE launch of the block-per-segment kernel
                const SegReduce2 c3[3] = {{dzfl[0], hn, g.cperm, g.cseg_ptr, 2 * (int)N, hn, dP, pw, (int)N, 2 * he, 2 * he + hn, 0, 0},
                                          {dzed[0], he, g.rperm, g.rseg_ptr, (int)N, he, dP, pw, (int)N, 0, 0, 0, 0},
                                          {dzed[0], he, g.cperm_all, g.cseg_all, (int)N, he, dP, pw, (int)N, he, he, 0, 0}};
                MPN_TRY(segment_reduce_csr2_x3(c3, E, s));
            } else {
                // sparse graphs: ONE launch of the short-segment kernel for the three (segment.hip, k_segment_reduce3).
                // (by row: the sorted order IS (direction, row), so a node's rows are three contiguous runs of the CSR -- no list)
                const SegReduce2 c3[3] = {{dzfl[0], hn, g.cperm, g.cseg_ptr, 2 * (int)N, hn, dP, pw, (int)N, 2 * he, 2 * he + hn, 0, 0},
                                          {dzed[0], he, nullptr, g.seg_ptr, (int)N, he, dP, pw, (int)N, 0, 0, 3, (int)N},
                                          {dzed[0], he, g.cperm_all, g.cseg_all, (int)N, he, dP, pw, (int)N, he, he, 0, 0}};
                MPN_TRY(segment_reduce_csr2_x3(c3, E, s));
            }
        } else if (E > 0) {
            // ---- B. aggregation backward + ReLU of the last flow layer ---------------------------
            {
                int64_t tot = E * (dn / 4 > 0 && dn % 4 == 0 ? dn / 4 : dn);
                hipLaunchKernelGGL(k_agg_bwd, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, p.dAGG, b.M, b.ARG,
                                   g.srow, g.seg_ptr, g.header, (int)N, E, dn, m.agg, dn != 1 ? 1 : 0, dzfl[nfl - 1]);
                MPN_LAUNCH_CHECK();
            }
            // ---- C. flow MLPs (both directions grouped) -------------------------------------------
            MPN_TRY(mlp_chain_backward(m.flow_out, &m.flow_in, dzfl, b.HF, dir_rr, E, s));
            {
                // layer 0:  Z = e_s Wfe^T + Pf[col];  dPf[n] = sum over the direction's edges with col == n
                // (index_put_ of x[flow_col], mpn.py:87,93)
                MPN_TRY(segment_reduce_csr2(dzfl[0], hn, g.cperm, g.cseg_ptr, 2 * (int)N, hn, dP, pw, (int)N, 2 * he, 2 * he + hn, s, E));
                const float* Wq[2] = {m.flow_out.weight[0] + kx, m.flow_in.weight[0] + kx};
                MPN_TRY(act_grad(2, dzfl[0], hn, nullptr, Wq, m.flow_out.in_dim, hn, de, dEc, de, nullptr, nullptr, 0, 1, dir_rr, E, s));
            }
            // ---- D. classifier (mpn.py:377 -> :114); also applies the ReLU mask of e_s -------------
            MPN_TRY(classifier_chain(b.HC, dzcl, grad_logits + (size_t)b_ * E, dEc, de != 1 ? e_s : nullptr));
            // ---- E. edge MLP (EdgeModel, mpn.py:67-69) ----------------------------------------------
            MPN_TRY(mlp_chain_backward(m.edge, nullptr, dzed, b.HE, nullptr, E, s));
            {
                // dPr / dPc: index_put_(accumulate) of x[row], x[col] (mpn.py:69)
                MPN_TRY(segment_reduce_csr2(dzed[0], he, g.rperm, g.rseg_ptr, (int)N, he, dP, pw, (int)N, 0, 0, s, E));
                MPN_TRY(segment_reduce_csr2(dzed[0], he, g.cperm_all, g.cseg_all, (int)N, he, dP, pw, (int)N, he, he, s, E));
                // gradient w.r.t. [e0 | e_{s-1}]: one product, then split (e_0 IS e0 at step 1)
                const float* Wa[2] = {m.edge.weight[0] + 2 * kx, nullptr};
                MPN_TRY(act_grad(1, dzed[0], he, nullptr, Wa, m.edge.in_dim, he, ke, p.dCat, ke, nullptr, nullptr, 0, 0, nullptr, E, s));
                float* dEp = step == 1 ? p.dE0 : p.dZed[ne - 1] + (size_t)(b_ - 1) * es;
                hipLaunchKernelGGL(k_split_cat, dim3((unsigned)((es + 255) / 256)), dim3(256), 0, s, p.dCat, E, de, d.ef == 2 ? 1 : 0,
                                   p.dE0, dEp, step == 1 ? 1 : 0);
                MPN_LAUNCH_CHECK();
            }
        } else {
            MPN_HIP(hipMemsetAsync(dP, 0, (size_t)N * pw * 4, s));
        }
        // ---- F. per-node projections  P = [x0 | x_{s-1}] Wnode^T -----------------------------------
        {
            float* dXp = p.dX[cx ^ 1];
            if (hoist_x && step > 1 && fuse_node_bwd) {
                // the reference's node width: this product, the previous step's ReLU mask and its node-update activation gradient
                // in one launch (segment.hip, node_step32_bwd); the next iteration starts at the chain kernel
                MPN_TRY(node_step32_bwd(dP, (int)N, pw, f.Wnode + dn, kx, f.x_hist + xs * (step - 1), m.node.weight[0],
                                        p.dZn + (size_t)(b_ - 1) * xs, p.dAGG, s));
                node_a_done = true;
            } else if (hoist_x && step > 1 && fuse_node_chain_bwd) {
                NodeChainBwdArgs a = {(int)N, dn, pw, dP, p.ncb_img, f.x_hist + xs * (step - 1), p.ncb_img + ncb_off_wu,
                                      p.dZn + (size_t)(b_ - 1) * xs, p.dAGG};
                MPN_TRY(launch_node_chain_bwd(a, s));
                node_a_done = true;
            } else if (hoist_x && step > 1 && pw % 8 == 0 && N * 2 < 2000000000 && !g_bwd_bf16) {
                // [N, pw] x [pw, dn] is 157 tiles of 64 x 64 at cfg-B -- not enough blocks for 256 CUs and 34 K steps each: the two K
                // halves run as the two groups of ONE grouped launch into dXp / dXh; the next step's ReLU-mask kernel adds them
                GemmArgs a = {};
                a.ngroups = 2; a.N = dn; a.K = pw / 2; a.ksplit = pw / 2; a.m_upper = 2 * N; a.small_tiles = 1;
                for (int q = 0; q < 2; ++q) {
                    GemmGroup& G = a.g[q];
                    init_group(G);
                    G.A = dP + q * (pw / 2); G.lda = pw;
                    G.B = f.Wnode + dn + (size_t)q * (pw / 2) * kx; G.ldb = kx;
                    G.C = q == 0 ? dXp : p.dXh; G.ldc = dn;
                    G.m_static = N;
                }
                MPN_TRY(launch_gemm(a, A_KCONTIG, B_NCONTIG, s));
                dx_split = true;
            } else if (hoist_x) {
                // only the x_{s-1} columns here (at step 1 x_0 IS x0); the re-attached x0's columns take the SUM of the steps'
                // dP after the loop: one product instead of L
                const float* Wa[2] = {f.Wnode + dn, nullptr};
                MPN_TRY(act_grad(1, dP, pw, nullptr, Wa, kx, pw, dn, step == 1 ? p.dX0 : dXp, dn, nullptr, nullptr, 0, step == 1 ? 1 : 0,
                                 nullptr, N, s, 1));
            } else {
                const float* Wa[2] = {f.Wnode, nullptr};
                MPN_TRY(act_grad(1, dP, pw, nullptr, Wa, kx, pw, kx, p.dCat, kx, nullptr, nullptr, 0, 0, nullptr, N, s));
                if (xs) {
                    hipLaunchKernelGGL(k_split_cat, dim3((unsigned)((xs + 255) / 256)), dim3(256), 0, s, p.dCat, N, dn, d.nf == 2 ? 1 : 0,
                                       p.dX0, step == 1 ? p.dX0 : dXp, step == 1 ? 1 : 0);
                    MPN_LAUNCH_CHECK();
                }
            }
        }
        cx ^= 1;
        if (next_group < ngroups - 1 && b_ == glo(next_group)) {
            MPN_HIP(hipEventRecord(side->ready, s));
            MPN_HIP(hipStreamWaitEvent(side->stream, side->ready, 0));
            forked = true;
            MPN_TRY(mp_weight_grads(glo(next_group), glo(next_group - 1) - glo(next_group), side->stream, p.slab_side));
            ++next_group;
        }
    }

    // The last group of steps goes to the side stream as soon as the loop is over (in order behind the earlier groups), under the
    // sums / products of the hoisted shares and the encoder's backward below.
    const int last_n = glo(ngroups - 2 < 0 ? -1 : ngroups - 2);
    if (L > 0 && forked) {
        MPN_HIP(hipEventRecord(side->ready, s));
        MPN_HIP(hipStreamWaitEvent(side->stream, side->ready, 0));
        MPN_TRY(mp_weight_grads(0, last_n, side->stream, p.slab_side));
    }
    // MPNHIP_PREC_FP32_SPLIT with a side stream: the weight-gradient products of this tail (hoisted shares, encoder layers) are
    // leaves -- nothing on the caller's stream reads their results -- so they are RECORDED here and run as ONE batch at the end
    // (flush_tail below), while the caller's stream goes on with the activation-gradient chain (the encoder chains keep every dZ
    // block of <= 3 layers: BwdPlan::T / Tn).  With MPNHIP_BWD_DEFER_SIDE_JOIN the encoder's products stay on the caller's stream
    // as separate launches and only the hoisted shares are batched: there the side stream's order must end with the
    // message-passing modules' gradients (a trainer puts their all-reduce behind it while the encoder's backward still runs).
    WpBatch tailb;
    WpBatchGuard tail_guard;
    const bool defer_tail = g_wgrad_split && L > 0 && forked && !getenv("MPNHIP_NO_TAIL_DEFER");
    const bool defer_encoder = defer_tail && !(flags & MPNHIP_BWD_DEFER_SIDE_JOIN) && m.enc_node.n_layers <= 3 && m.enc_edge.n_layers <= 3;
    if (defer_tail) wp_batch_begin(&tailb, p.slab_tail, p.slab_tail_floats, true);
    // flush_tail runs what was recorded and closes the batch (its "+=" go to gradient columns disjoint from the groups': it may run
    // beside the last group of steps).  Where it runs.  With the encoder's products deferred to the end anyway (no MPNHIP_BWD_DEFER_SIDE_JOIN) it runs on the
    // CALLER's stream itself, then the join, the unpacking and whatever the caller enqueues next: no hop ahead of the batch (event
    // record -> wait: 10 - 18 us each) and one instead of two behind it.  Same-box A-B, three runs each, second side stream / caller's
    // stream / first side stream: cfg-B 4.94 - 5.00 / 4.90 - 4.93 / 4.92 - 4.97 ms, cfg-C 1.76 - 1.78 / 1.72 - 1.74 / 1.80 - 1.82, cfg-E 33.5 - 33.6 /
    // 33.6 / 34.0, cfg-D 0.494 / 0.444 / 0.470 -- the second stream was round 4's answer to a tail of ~0.3 ms behind the last group; with
    // the launches' blocks dispatched longest first it no longer pays.  A trainer's path (MPNHIP_BWD_DEFER_SIDE_JOIN: the hoisted
    // shares' products run early, the message-passing gradients' collective follows on the first side stream) keeps the second stream
    // on large graphs -- the first side stream is ordered behind it, so the unpacking and the collective that follow there see both.
    // MPNHIP_TAIL_STREAM2=1 / MPNHIP_NO_TAIL_STREAM2=1 force either, MPNHIP_NO_TAIL_INLINE=1 keeps the batch off the caller's stream.
    const bool tail_beside = !getenv("MPNHIP_NO_TAIL_STREAM2") &&
                             (getenv("MPNHIP_TAIL_STREAM2") || ((flags & MPNHIP_BWD_DEFER_SIDE_JOIN) && (double)E * dn >= 1e6));
    const bool tail_inline = defer_encoder && !tail_beside && !getenv("MPNHIP_NO_TAIL_INLINE");
    Rows16Later later16;
    later16.pool = p.enc16;
    later16.pool_elems = p.enc16_elems;
    auto flush_tail = [&]() -> int {
        if (!wp_batch_open()) return MPNHIP_OK;
        hipStream_t st = tail_inline ? s : tail_beside ? side->stream2 : side->stream;
        if (!tail_inline) {
            MPN_HIP(hipEventRecord(side->ready, s));
            MPN_HIP(hipStreamWaitEvent(st, side->ready, 0));
        }
        for (int i = 0; i < later16.n; ++i) MPN_TRY(to_bf16_rows(later16.item[i].src, later16.item[i].dst, later16.item[i].n, st));
        later16.n = 0;
        MPN_TRY(wp_batch_flush(st));
        if (tail_beside) {
            MPN_HIP(hipEventRecord(side->done2, side->stream2));
            MPN_HIP(hipStreamWaitEvent(side->stream, side->done2, 0));
        }
        return MPNHIP_OK;
    };
    if (hoist_x) {
        const int64_t n4 = N * pw / 4;
        hipLaunchKernelGGL(k_sum_blocks, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, p.dP, N * pw, (int)L, n4, p.dPsum);
        MPN_LAUNCH_CHECK();
        const float* Wa[2] = {f.Wnode, nullptr};
        MPN_TRY(act_grad(1, p.dPsum, pw, nullptr, Wa, kx, pw, dn, p.dX0, dn, nullptr, nullptr, 0, 1, nullptr, N, s));
        // ... and the x0 columns [0, dn) of the packed projection weight's gradient: (sum of dP)^T x0 (disjoint from the columns the
        // side stream accumulates; main-stream slab buffer)
        float* gw[2] = {p.gWnode, nullptr};
        if (node16) {
            MPN_TRY(to_bf16_rows(p.dPsum, p.dPsum16, N * pw, s));
            Src16Scope rows16(true);
            MPN_TRY(weight_grad(p, p.slab, 1, {reinterpret_cast<const float*>(p.dPsum16), pw, 0}, nullptr, {reinterpret_cast<const float*>(f.xb_hist), dn, 0},
                                {nullptr, 0, 0}, dn, nullptr, pw, dn, gw, kx, nullptr, nullptr, N, 1, s));
        } else {
            MPN_TRY(weight_grad(p, p.slab, 1, {p.dPsum, pw, 0}, nullptr, {x0, dn, 0}, {nullptr, 0, 0}, dn, nullptr, pw, dn, gw, kx, nullptr, nullptr,
                                N, 1, s));
        }
    }
    if (hoist_e0) {
        // S = sum_s dZ1_s (the blocks are all kept for the weight gradients);  dE0 += S W1[:, e0 columns];  dW1[:, e0 columns] += S^T e0
        const int64_t n4 = E * he / 4;
        // bf16-operand training: both consumers of S round it to bf16 as they stage it, so S itself is kept as bf16 rows (rounded once,
        // the same values) and its partner e0 is the forward's bf16 mirror: half the bytes written and read twice, the GEMM reads
        // bf16 rows, the weight-gradient product runs on the bf16-row kernels
        const bool s16 = use_b16 && f.eb_hist && he % 8 == 0 && de % 8 == 0 && !getenv("MPNHIP_NO_S16");
        if (s16) hipLaunchKernelGGL(k_sum_blocks_bf16<true>, dim3((unsigned)((n4 / 2 + 255) / 256)), dim3(256), 0, s, reinterpret_cast<const unsigned short*>(p.dZed[0]),
                                    E * he, (int)L, n4 / 2, p.dZ1sum);
        else if (use_b16) hipLaunchKernelGGL(k_sum_blocks_bf16<false>, dim3((unsigned)((n4 / 2 + 255) / 256)), dim3(256), 0, s, reinterpret_cast<const unsigned short*>(p.dZed[0]),
                                             E * he, (int)L, n4 / 2, p.dZ1sum);   // (he % 8 == 0: chain_bf16_train_ok)
        else hipLaunchKernelGGL(k_sum_blocks, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, p.dZed[0], E * he, (int)L, n4, p.dZ1sum);
        MPN_LAUNCH_CHECK();
        const float* Wa[2] = {m.edge.weight[0] + 2 * kx, nullptr};
        MPN_TRY(act_grad(1, p.dZ1sum, he, nullptr, Wa, m.edge.in_dim, he, de, p.dE0, de, nullptr, nullptr, 0, 1, nullptr, E, s, -1, s16));
        float* gw[2] = {m.edge.grad_weight[0] + 2 * kx, nullptr};
        if (s16) {
            Src16Scope rows16(true);
            MPN_TRY(weight_grad(p, p.slab, 1, {p.dZ1sum, he, 0}, nullptr, {reinterpret_cast<const float*>(f.eb_hist), de, 0}, {nullptr, 0, 0}, de, nullptr,
                                he, de, gw, m.edge.in_dim, nullptr, nullptr, E, 1, s));
        } else {
            MPN_TRY(weight_grad(p, p.slab, 1, {p.dZ1sum, he, 0}, nullptr, {e0, de, 0}, {nullptr, 0, 0}, de, nullptr, he, de, gw, m.edge.in_dim,
                                nullptr, nullptr, E, 1, s));
        }
    }
    auto unpack_node_grads = [&](hipStream_t us) -> int {
        // the packed node-projection gradient [W1r; W1c; Wfo_x; Wfi_x] back into the layers' grads (their biases were handled above)
        struct { float* dst; int64_t ld; int c0; int r0; int rows; } parts[4] = {
            {m.edge.grad_weight[0], m.edge.in_dim, 0, 0, he},
            {m.edge.grad_weight[0], m.edge.in_dim, kx, he, he},
            {m.flow_out.grad_weight[0], m.flow_out.in_dim, 0, 2 * he, hn},
            {m.flow_in.grad_weight[0], m.flow_in.in_dim, 0, 2 * he + hn, hn}};
        AddParts P;
        int64_t tot = 0;
        for (int i = 0; i < 4; ++i) {
            P.dst[i] = parts[i].dst; P.ldd[i] = parts[i].ld; P.c0[i] = parts[i].c0; P.r0[i] = parts[i].r0; P.rows[i] = parts[i].rows;
            tot += (int64_t)parts[i].rows * kx;
        }
        if (tot > 0) {   // (one launch for the four blocks: they are disjoint destinations)
            hipLaunchKernelGGL(k_add_block4, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, us, p.gWnode, (int64_t)kx, kx, P);
            MPN_LAUNCH_CHECK();
        }
        return MPNHIP_OK;
    };
    bool unpack_pending = false;
    if (L > 0) {
        if (forked) {
            // (the unpacking follows on the side stream as well: with it every gradient of the message-passing modules and of
            // the classifier is final IN SIDE-STREAM ORDER, while the caller's stream still runs the encoder's backward)
            if (defer_encoder) {
                unpack_pending = true;   // after the tail batch (the hoisted x0 product adds into gWnode), at the end
            } else {
                // the hoisted products: recorded -> run them now on the side stream; run on the caller's stream -> order the
                // unpacking behind them
                if (defer_tail) MPN_TRY(flush_tail());
                else {
                    MPN_HIP(hipEventRecord(side->ready, s));
                    MPN_HIP(hipStreamWaitEvent(side->stream, side->ready, 0));
                }
                MPN_TRY(unpack_node_grads(side->stream));
            }
        } else {
            MPN_TRY(mp_weight_grads(0, last_n, s, p.slab));
            MPN_TRY(unpack_node_grads(s));
        }
    } else {
        // mpn.py:387-389: only the classifier sits between the encoder output and the logits
        StepBufs b = f.step0;
        float* dzc[MPNHIP_MAX_LAYERS];
        for (int i = 0; i + 1 < nc; ++i) dzc[i] = p.dZcl[i];
        MPN_TRY(classifier_chain(b.HC, dzc, grad_logits, p.dE0, nullptr));
        for (int i = nc - 1; i >= 0 && E > 0; --i) {
            const int n_out = cls.out_dims[i], k_in = i == 0 ? de : cls.out_dims[i - 1];
            const bool top = i == nc - 1;
            float* gw[2] = {cls.grad_weight[i], nullptr};
            float* gb[2] = {cls.grad_bias[i], nullptr};
            Operand dz = top ? Operand{grad_logits, 1, 0} : Operand{p.dZcl[i], n_out, 0};
            Operand h = i == 0 ? Operand{e0, de, 0} : Operand{b.HC[i - 1], k_in, 0};
            MPN_TRY(weight_grad(p, p.slab, 1, dz, top ? g.perm : nullptr, h, {nullptr, 0, 0}, k_in, nullptr, n_out, k_in, gw, k_in, gb, nullptr,
                                E, 1, s));
        }
        // incoming gradients of the final latents ARE gradients of the encoder outputs
        if (xs) {
            hipLaunchKernelGGL(k_add_block, dim3((unsigned)((xs + 255) / 256)), dim3(256), 0, s, p.dX[0], dn, p.dX0, dn, 0, (int)N, dn);
            MPN_LAUNCH_CHECK();
        }
        if (grad_e_out && es) {
            // dE0 was seeded with the gathered grad_e_out (dE_last aliases dE0 when L == 0): nothing to add
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
                MPN_TRY(relu_mask(p.dX0, x0, p.Tn[0], (int64_t)xs, s));
                dz = p.Tn[0];
            }
            // (bf16-operand training with the deferred tail batch: the wide layers' products over bf16 rows, rounded on the tail's stream)
            struct LaterScope { explicit LaterScope(Rows16Later* l) { g_rows16_later = l; } ~LaterScope() { g_rows16_later = nullptr; } };
            LaterScope later_scope(defer_encoder && node16 && p.enc16 ? &later16 : nullptr);
            MPN_TRY(mlp_tail_backward(p, p.Tn, en, hid, &dz, &cur, N, s));
            MPN_TRY(encoder_weight_grad(p, dz, x, en.out_dims[0], en.in_dim, en.grad_weight[0], en.grad_bias[0], N, s));
            if (grad_x) {
                const float* Wq[2] = {en.weight[0], nullptr};
                MPN_TRY(act_grad(1, dz, en.out_dims[0], nullptr, Wq, en.in_dim, en.out_dims[0], en.in_dim, grad_x, en.in_dim,
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
        const bool ref_encoder = !g_bwd_bf16 && ee.n_layers == 3 && ee.in_dim == 6 && ee.out_dims[0] == 18 && ee.out_dims[1] == 18 && ee.out_dims[2] == 16 &&
                                 p.t_width >= 52 && !getenv("MPNHIP_NO_ENCODER_FUSION");
        if (E > 0 && ref_encoder) {
            // the reference's edge encoder: all three activation gradients in one launch (dz2 | dz1 | dz0 side by side in T[0]),
            // then the weight-gradient products as before
            float* dz2 = p.T[0];
            float* dz1 = dz2 + (size_t)E * 16;
            float* dz0 = dz1 + (size_t)E * 18;
            count_path(PC_EDGE_ENCODER_BWD);
            hipLaunchKernelGGL((k_edge_encoder_bwd<18, 18, 16>), dim3((unsigned)((E + 255) / 256)), dim3(256), 0, s, p.dE0, e0, hid[1], hid[0],
                               ee.weight[2], ee.weight[1], E, dz2, dz1, dz0);
            MPN_LAUNCH_CHECK();
            {
                float* gw[2] = {ee.grad_weight[2], nullptr};
                float* gb[2] = {ee.grad_bias[2], nullptr};
                MPN_TRY(weight_grad(p, p.slab, 1, {dz2, 16, 0}, nullptr, {hid[1], 18, 0}, {nullptr, 0, 0}, 18, nullptr, 16, 18, gw, 18, gb, nullptr, E, 1, s));
            }
            {
                float* gw[2] = {ee.grad_weight[1], nullptr};
                float* gb[2] = {ee.grad_bias[1], nullptr};
                MPN_TRY(weight_grad(p, p.slab, 1, {dz1, 18, 0}, nullptr, {hid[0], 18, 0}, {nullptr, 0, 0}, 18, nullptr, 18, 18, gw, 18, gb, nullptr, E, 1, s));
            }
            dz = dz0;
        } else if (E > 0) {
            if (ee.out_dims[ee.n_layers - 1] != 1) {
                MPN_TRY(relu_mask(p.dE0, e0, p.T[0], (int64_t)es, s));
                dz = p.T[0];
            }
            MPN_TRY(mlp_tail_backward(p, p.T, ee, hid, &dz, &cur, E, s));
        }
        if (E > 0) {
            float* gw[2] = {ee.grad_weight[0], nullptr};
            float* gb[2] = {ee.grad_bias[0], nullptr};
            // layer 0 read edge_attr through the sort permutation
            MPN_TRY(weight_grad(p, p.slab, 1, {dz, ee.out_dims[0], 0}, nullptr, {edge_attr, ee.in_dim, 0}, {nullptr, 0, 0}, ee.in_dim,
                                g.perm, ee.out_dims[0], ee.in_dim, gw, ee.in_dim, gb, nullptr, E, 1, s));
            if (grad_edge_attr) {
                const float* Wq[2] = {ee.weight[0], nullptr};
                MPN_TRY(act_grad(1, dz, ee.out_dims[0], nullptr, Wq, ee.in_dim, ee.out_dims[0], ee.in_dim, grad_edge_attr,
                                 ee.in_dim, g.perm, nullptr, 0, 0, nullptr, E, s));
            }
        }
    }
    if (unpack_pending) {
        MPN_TRY(flush_tail());
        if (tail_inline) {
            // the groups' "+=" first (side stream), then the unpacking on the caller's stream
            MPN_HIP(hipEventRecord(side->done, side->stream));
            MPN_HIP(hipStreamWaitEvent(s, side->done, 0));
            join.joined = true;
            MPN_TRY(unpack_node_grads(s));
        } else {
            MPN_TRY(unpack_node_grads(side->stream));
        }
    }
    if ((flags & MPNHIP_BWD_DEFER_SIDE_JOIN) && !forked && side) {
        // nothing was forked (no side stream work: few steps, or the fork was not possible), but the caller was promised that the
        // side stream's order ends with the message-passing modules' gradients: order it behind the caller's stream
        MPN_HIP(hipEventRecord(side->ready, s));
        MPN_HIP(hipStreamWaitEvent(side->stream, side->ready, 0));
    }
    if (L > 0 && forked) {
        // join: every "+=" of the side stream's groups (and the unpacking) is in.  MPNHIP_BWD_DEFER_SIDE_JOIN leaves it to the
        // caller (mpnhip_side_stream_join): a data-parallel trainer first puts the all-reduce of the message-passing modules'
        // gradients on the side stream, where it overlaps the encoder's backward still running on the caller's stream.
        if (flags & MPNHIP_BWD_DEFER_SIDE_JOIN) {
            join.joined = true;
        } else if (!join.joined) {   // (tail_inline: joined ahead of the unpacking)
            MPN_HIP(hipEventRecord(side->done, side->stream));
            MPN_HIP(hipStreamWaitEvent(s, side->done, 0));
            join.joined = true;
        }
    }
    return MPNHIP_OK;
}

// Test / diagnosis instrumentation: one block of pre-activation gradients mpnhip_backward left in its workspace (sorted edge
// order / node order as stored; include/mpnhip.h).
extern "C" int mpnhip_debug_backward_saved(const mpnhip_model* model, int n_nodes, int64_t n_edges, const void* bwd_workspace,
                                           size_t bwd_workspace_bytes, int what, int step, int layer, float* out, int64_t* rows_out,
                                           int* width_out, void* stream_) {
    hipStream_t s = static_cast<hipStream_t>(stream_);
    MPN_CHECK_ARG(model && bwd_workspace, "debug_backward_saved: null argument");
    const mpnhip_model& m = *model;
    Dims d;
    MPN_TRY(check_full(m, &d));
    const int64_t N = n_nodes, E = n_edges;
    BwdPlan p;
    const size_t need = plan_backward(m, d, N, E, const_cast<void*>(bwd_workspace), &p);
    if (bwd_workspace_bytes < need) {
        set_error("debug_backward_saved: workspace %zu < %zu", bwd_workspace_bytes, need);
        return MPNHIP_ERR_WORKSPACE;
    }
    MPN_CHECK_ARG(step >= 1 && step <= (d.L > 0 ? d.L : 1), "debug_backward_saved: step %d", step);
    const size_t b = (size_t)(step - 1);
    const float* src = nullptr;
    int64_t rows = 0;
    int width = 0;
    switch (what) {
        case MPNHIP_BWD_SAVED_DZ_NODE: src = p.dZn + b * N * d.dn; rows = N; width = d.dn; break;
        case MPNHIP_BWD_SAVED_DP: src = p.dP + b * N * d.pw; rows = N; width = d.pw; break;
        case MPNHIP_BWD_SAVED_DZ_FLOW:
            if (layer >= 0 && layer < m.flow_in.n_layers) { width = m.flow_in.out_dims[layer]; src = p.dZfl[layer] + b * E * width; rows = E; }
            break;
        case MPNHIP_BWD_SAVED_DZ_EDGE:
            if (layer >= 0 && layer < m.edge.n_layers) { width = m.edge.out_dims[layer]; src = p.dZed[layer] + b * E * width; rows = E; }
            break;
        case MPNHIP_BWD_SAVED_DZ_CLS:
            if (layer >= 0 && layer + 1 < m.classifier.n_layers) { width = m.classifier.out_dims[layer]; src = p.dZcl[layer] + b * E * width; rows = E; }
            break;
        default: break;
    }
    if (!src) {
        set_error("debug_backward_saved: nothing kept for what = %d, step = %d, layer = %d", what, step, layer);
        return MPNHIP_ERR_ARG;
    }
    if (rows_out) *rows_out = rows;
    if (width_out) *width_out = width;
    if (out && rows > 0 && p.b16 && (what == MPNHIP_BWD_SAVED_DZ_FLOW || what == MPNHIP_BWD_SAVED_DZ_EDGE || what == MPNHIP_BWD_SAVED_DZ_CLS)) {
        // bf16-operand training on the fused kernels: the per-edge dZ blocks are bf16 rows (block b at b * E * width shorts)
        const float* base = what == MPNHIP_BWD_SAVED_DZ_FLOW ? p.dZfl[layer] : (what == MPNHIP_BWD_SAVED_DZ_EDGE ? p.dZed[layer] : p.dZcl[layer]);
        const unsigned short* s16 = reinterpret_cast<const unsigned short*>(base) + b * E * width;
        const int64_t n = rows * width;
        hipLaunchKernelGGL(k_bf16_to_f32, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, s16, out, n);
        MPN_LAUNCH_CHECK();
        return MPNHIP_OK;
    }
    if (out && rows > 0) MPN_HIP(hipMemcpyAsync(out, src, (size_t)rows * width * sizeof(float), hipMemcpyDeviceToDevice, s));
    return MPNHIP_OK;
}
