// Weight-gradient products of the backward pass:
//   dW[o, c] += sum_m dZ[m, o] * H[m, c]        db[o] += sum_m dZ[m, o]
// i.e. a GEMM whose reduction runs over the EDGES (or nodes): tens of thousands of rows against a
// small [n_out, k_in] output.  Parallelism comes from splitting the row range: grid.y enumerates row
// chunks, every block reduces its chunk with fp32 MFMAs (v_mfma_f32_32x32x2_f32) into a private slab
// [n_out][k_in + 4] (column k_in carries the bias partial), and a second kernel sums the slabs in a
// fixed order into the caller's gradient buffers -- deterministic, no float atomics.
//
// Both operands are read exactly as stored (row-major, rows = reduction index), 16 bytes per lane,
// straight into k-major LDS images (the row index IS the MFMA k index, so nothing is transposed).
// H may come as two column segments (the reference's torch.cat([initial, current]) input), rows of
// either operand may be gathered through an index (edge_attr / dlogits live in original edge order).
#include "common.h"

namespace mpnhip {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int TBK = 32;
constexpr int TBM = 64;    // n_out tile
constexpr int TBN = 128;   // k_in tile
constexpr int TNT = 256;

__device__ __forceinline__ float4 ld4t(const float* p) { return *reinterpret_cast<const float4*>(p); }

__global__ __launch_bounds__(TNT) void gemm_tn_kernel(TnArgs args) {
    constexpr int PA = TBM + 4, PB = TBN + 4;
    __shared__ __attribute__((aligned(16))) float As[TBK * PA];
    __shared__ __attribute__((aligned(16))) float Bs[TBK * PB];
    const TnGroup& G = args.g[blockIdx.z];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 31, lh = lane >> 5;
    const int n_out = args.n_out, k_in = args.k_in, csplit = args.csplit;
    const int ntile_c = (k_in + TBN - 1) / TBN;
    const int o0 = (blockIdx.x / ntile_c) * TBM;
    const int c0 = (blockIdx.x % ntile_c) * TBN;

    const int rb = G.row_begin ? *G.row_begin : 0;
    const int re = G.row_end ? *G.row_end : (int)G.m_static;
    int r0 = rb + blockIdx.y * args.chunk;
    int r1 = r0 + args.chunk;
    r1 = r1 < re ? r1 : re;

    // loader geometry: A' tile [32 rows][64 cols] = 512 float4 (2 / thread); B' tile [32][128] = 1024 (4 / thread)
    const int ar = tid >> 4, ac = (tid & 15) * 4;        // + 16 rows for the second
    const int br = tid >> 5, bc = (tid & 31) * 4;        // + 8 rows per j
    int oc = o0 + ac;
    oc = oc + 3 < n_out ? oc : n_out - 4;                // clamped columns are never stored
    int cc = c0 + bc;
    cc = cc + 3 < k_in ? cc : k_in - 4;
    const bool bseg2 = cc >= csplit;                     // csplit % 4 == 0: a float4 lies in one segment
    const float* hbase = bseg2 ? G.H2 : G.H;
    const int64_t ldh = bseg2 ? G.ldh2 : G.ldh;
    const int hcol = bseg2 ? cc - csplit : cc;

    f32x16 acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    float bsum = 0.f;
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 a_reg[2], b_reg[4];

    auto load = [&](int m0) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            int m = m0 + ar + 16 * j;
            bool ok = m < r1;
            int mc = ok ? m : r1 - 1;
            int64_t ri = G.dz_idx ? G.dz_idx[mc] : mc;
            float4 v = ld4t(G.dZ + ri * G.ldz + oc);
            a_reg[j] = ok ? v : z4;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int m = m0 + br + 8 * j;
            bool ok = m < r1;
            int mc = ok ? m : r1 - 1;
            int64_t ri = G.h_idx ? G.h_idx[mc] : mc;
            float4 v = ld4t(hbase + ri * ldh + hcol);
            b_reg[j] = ok ? v : z4;
        }
    };
    auto store = [&]() {
#pragma unroll
        for (int j = 0; j < 2; ++j) *reinterpret_cast<float4*>(&As[(ar + 16 * j) * PA + ac]) = a_reg[j];
#pragma unroll
        for (int j = 0; j < 4; ++j) *reinterpret_cast<float4*>(&Bs[(br + 8 * j) * PB + bc]) = b_reg[j];
    };

    if (r0 < r1) {
        load(r0);
        for (int m0 = r0; m0 < r1; m0 += TBK) {
            store();
            __syncthreads();
            if (m0 + TBK < r1) load(m0 + TBK);
#pragma unroll
            for (int kk = 0; kk < TBK; kk += 2) {
                const float a = As[(kk + lh) * PA + wm * 32 + li];
                const float b0 = Bs[(kk + lh) * PB + wn * 64 + li];
                const float b1 = Bs[(kk + lh) * PB + wn * 64 + 32 + li];
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b0, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b1, acc[1], 0, 0, 0);
            }
            if (c0 == 0 && tid < TBM) {
#pragma unroll
                for (int kk = 0; kk < TBK; ++kk) bsum += As[kk * PA + tid];
            }
            __syncthreads();
        }
    }
    // ---- write the partial tile into this chunk's slab (zeros when the chunk is empty) ------------
    const int kpad = k_in + 4;
    float* slab = G.slab + (size_t)blockIdx.y * n_out * kpad;
#pragma unroll
    for (int tj = 0; tj < 2; ++tj) {
        const int c = c0 + wn * 64 + tj * 32 + li;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int o = o0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (o < n_out && c < k_in) slab[(size_t)o * kpad + c] = acc[tj][r];
        }
    }
    if (c0 == 0 && tid < TBM && o0 + tid < n_out) slab[(size_t)(o0 + tid) * kpad + k_in] = bsum;
}

// grad_w[o * ldw + c] += sum_s slab[s][o][c];  grad_b[o] += sum_s slab[s][o][k_in]
__global__ void slab_reduce_kernel(TnArgs args) {
    const TnGroup& G = args.g[blockIdx.z];
    const int kpad = args.k_in + 4;
    const int64_t total = (int64_t)args.n_out * (args.k_in + 1);
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const int o = (int)(t / (args.k_in + 1)), c = (int)(t % (args.k_in + 1));
    const float* p = G.slab + (size_t)o * kpad + (c < args.k_in ? c : args.k_in);
    const size_t stride = (size_t)args.n_out * kpad;
    float s = 0.f;
    for (int i = 0; i < args.nsplit; ++i) s += p[i * stride];
    if (c < args.k_in) {
        if (G.grad_w) G.grad_w[(int64_t)o * G.ldw + c] += s;
    } else if (G.grad_b) {
        G.grad_b[o] += s;
    }
}

// Any shape / alignment: one block per output element (o, c), c == k_in is the bias column.
__global__ __launch_bounds__(256) void gemm_tn_generic_kernel(TnArgs args) {
    const TnGroup& G = args.g[blockIdx.z];
    const int o = blockIdx.x / (args.k_in + 1), c = blockIdx.x % (args.k_in + 1);
    const int rb = G.row_begin ? *G.row_begin : 0;
    const int re = G.row_end ? *G.row_end : (int)G.m_static;
    float s = 0.f;
    for (int m = rb + threadIdx.x; m < re; m += blockDim.x) {
        int64_t rz = G.dz_idx ? G.dz_idx[m] : m;
        float z = G.dZ[rz * G.ldz + o];
        float h = 1.f;
        if (c < args.k_in) {
            int64_t rh = G.h_idx ? G.h_idx[m] : m;
            h = c >= args.csplit ? G.H2[rh * G.ldh2 + c - args.csplit] : G.H[rh * G.ldh + c];
        }
        s = fmaf(z, h, s);
    }
    __shared__ float red[256];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if (threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        if (c < args.k_in) {
            if (G.grad_w) G.grad_w[(int64_t)o * G.ldw + c] += red[0];
        } else if (G.grad_b) {
            G.grad_b[o] += red[0];
        }
    }
}

static bool al16t(const void* p) { return (((uintptr_t)p) & 15) == 0; }

size_t tn_slab_floats(int n_out, int k_in, int64_t m_upper) {
    TnArgs a = {};
    a.n_out = n_out;
    a.k_in = k_in;
    a.m_upper = m_upper;
    tn_plan(a);
    return (size_t)a.nsplit * n_out * (k_in + 4);
}

void tn_plan(TnArgs& a) {
    int tiles = ((a.n_out + TBM - 1) / TBM) * ((a.k_in + TBN - 1) / TBN);
    int target = 1024 / (tiles > 0 ? tiles : 1);
    if (target < 1) target = 1;
    if (target > 256) target = 256;
    int64_t chunk = (a.m_upper + target - 1) / target;
    chunk = (chunk + TBK - 1) / TBK * TBK;
    if (chunk < 256) chunk = 256;
    a.chunk = (int)chunk;
    a.nsplit = (int)((a.m_upper + chunk - 1) / chunk);
    if (a.nsplit < 1) a.nsplit = 1;
}

int launch_gemm_tn(const TnArgs& a_in, hipStream_t s) {
    TnArgs a = a_in;
    MPN_CHECK_ARG(a.ngroups == 1 || a.ngroups == 2, "gemm_tn: ngroups");
    MPN_CHECK_ARG(a.n_out >= 1 && a.k_in >= 1 && a.csplit >= 0 && a.csplit <= a.k_in, "gemm_tn: dims");
    if (a.m_upper <= 0) return MPNHIP_OK;
    bool fast = (a.n_out % 4 == 0) && (a.k_in % 4 == 0) && (a.csplit % 4 == 0);
    for (int i = 0; i < a.ngroups; ++i) {
        const TnGroup& g = a.g[i];
        MPN_CHECK_ARG(g.dZ && g.H && (a.csplit == a.k_in || g.H2), "gemm_tn: null operand");
        fast = fast && al16t(g.dZ) && g.ldz % 4 == 0 && al16t(g.H) && g.ldh % 4 == 0 &&
               (!g.H2 || (al16t(g.H2) && g.ldh2 % 4 == 0)) && g.slab;
    }
    if (!fast) {
        dim3 grid((unsigned)(a.n_out * (a.k_in + 1)), 1, a.ngroups);
        hipLaunchKernelGGL(gemm_tn_generic_kernel, grid, dim3(256), 0, s, a);
        MPN_LAUNCH_CHECK();
        return MPNHIP_OK;
    }
    tn_plan(a);
    int tiles = ((a.n_out + TBM - 1) / TBM) * ((a.k_in + TBN - 1) / TBN);
    hipLaunchKernelGGL(gemm_tn_kernel, dim3(tiles, a.nsplit, a.ngroups), dim3(TNT), 0, s, a);
    MPN_LAUNCH_CHECK();
    int64_t total = (int64_t)a.n_out * (a.k_in + 1);
    hipLaunchKernelGGL(slab_reduce_kernel, dim3((unsigned)((total + 255) / 256), 1, a.ngroups), dim3(256), 0, s, a);
    MPN_LAUNCH_CHECK();
    return MPNHIP_OK;
}

}  // namespace mpnhip
