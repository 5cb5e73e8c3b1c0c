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
__device__ __forceinline__ float4 keep4t(bool ok, float4 v) {
    v.x = ok ? v.x : 0.f; v.y = ok ? v.y : 0.f; v.z = ok ? v.z : 0.f; v.w = ok ? v.w : 0.f;
    return v;
}
// field-wise select of the group (indexing the by-value kernel argument with blockIdx.z would force
// the whole struct into scratch memory)
#define TN_G(field) (blockIdx.z == 0 ? args.g[0].field : args.g[1].field)

__global__ __launch_bounds__(TNT) void gemm_tn_kernel(TnArgs args) {
    constexpr int PA = TBM + 4, PB = TBN + 4;
    __shared__ __attribute__((aligned(16))) float As[TBK * PA];
    __shared__ __attribute__((aligned(16))) float Bs[TBK * PB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 31, lh = lane >> 5;
    const int n_out = args.n_out, k_in = args.k_in, csplit = args.csplit;
    const int ntile_c = (k_in + TBN - 1) / TBN;
    const int o0 = (blockIdx.x / ntile_c) * TBM;
    const int c0 = (blockIdx.x % ntile_c) * TBN;

    const int* rbp = TN_G(row_begin);
    const int* rep = TN_G(row_end);
    const int rb = rbp ? *rbp : 0;
    const int re = rep ? *rep : (int)TN_G(m_static);
    const int batch = blockIdx.y / args.nsplit;
    const int r0 = rb + (blockIdx.y % args.nsplit) * args.chunk;
    int r1 = r0 + args.chunk;
    r1 = r1 < re ? r1 : re;
    if (r0 >= r1) return;  // empty chunk: the reduce kernel skips it too

    // loader geometry: A' tile [32 rows][64 cols] = 512 float4 (2 / thread); B' tile [32][128] = 1024 (4 / thread)
    const int ar = tid >> 4, ac = (tid & 15) * 4;        // + 16 rows for the second
    const int br = tid >> 5, bc = (tid & 31) * 4;        // + 8 rows per j
    int oc = o0 + ac;
    oc = oc + 3 < n_out ? oc : n_out - 4;                // clamped columns are never stored
    int cc = c0 + bc;
    cc = cc + 3 < k_in ? cc : k_in - 4;
    const bool bseg2 = cc >= csplit;                     // csplit % 4 == 0: a float4 lies in one segment
    const float* hbase = (bseg2 ? TN_G(H2) : TN_G(H)) + (bseg2 ? cc - csplit : cc) +
                         (int64_t)batch * (bseg2 ? TN_G(h2_bstride) : TN_G(h_bstride));
    const int64_t ldh = bseg2 ? TN_G(ldh2) : TN_G(ldh);
    const float* zbase = TN_G(dZ) + oc + (int64_t)batch * TN_G(z_bstride);
    const int64_t ldz = TN_G(ldz);

    f32x16 acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    float bsum = 0.f;
    float4 a_reg[2], b_reg[4];

    auto load = [&](int m0) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            int m = m0 + ar + 16 * j;
            bool ok = m < r1;
            a_reg[j] = keep4t(ok, ld4t(zbase + (int64_t)(ok ? m : r1 - 1) * ldz));
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int m = m0 + br + 8 * j;
            bool ok = m < r1;
            b_reg[j] = keep4t(ok, ld4t(hbase + (int64_t)(ok ? m : r1 - 1) * ldh));
        }
    };
    auto store = [&]() {
#pragma unroll
        for (int j = 0; j < 2; ++j) *reinterpret_cast<float4*>(&As[(ar + 16 * j) * PA + ac]) = a_reg[j];
#pragma unroll
        for (int j = 0; j < 4; ++j) *reinterpret_cast<float4*>(&Bs[(br + 8 * j) * PB + bc]) = b_reg[j];
    };

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
    // ---- write the partial tile into this chunk's slab --------------------------------------------
    const int kpad = k_in + 4;
    float* slab = TN_G(slab) + (size_t)blockIdx.y * n_out * kpad;
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

// Any shape / alignment / row gathers: grid (output element (o, c), chunk); c == k_in is the bias column.
__global__ __launch_bounds__(256) void gemm_tn_generic_kernel(TnArgs args) {
    const int o = blockIdx.x / (args.k_in + 1), c = blockIdx.x % (args.k_in + 1);
    const int* rbp = TN_G(row_begin);
    const int* rep = TN_G(row_end);
    const int rb = rbp ? *rbp : 0;
    const int re = rep ? *rep : (int)TN_G(m_static);
    const int batch = blockIdx.y / args.nsplit;
    const int r0 = rb + (blockIdx.y % args.nsplit) * args.chunk;
    int r1 = r0 + args.chunk;
    r1 = r1 < re ? r1 : re;
    if (r0 >= r1) return;
    const float* dZ = TN_G(dZ) + (int64_t)batch * TN_G(z_bstride);
    const float* H = TN_G(H) + (int64_t)batch * TN_G(h_bstride);
    const float* H2 = TN_G(H2) ? TN_G(H2) + (int64_t)batch * TN_G(h2_bstride) : nullptr;
    const int* zi = TN_G(dz_idx);
    const int* hi = TN_G(h_idx);
    const int64_t ldz = TN_G(ldz), ldh = TN_G(ldh), ldh2 = TN_G(ldh2);
    float s = 0.f;
    for (int m = r0 + threadIdx.x; m < r1; m += blockDim.x) {
        int64_t rz = zi ? zi[m] : m;
        float z = dZ[rz * ldz + o];
        float h = 1.f;
        if (c < args.k_in) {
            int64_t rh = hi ? hi[m] : m;
            h = c >= args.csplit ? H2[rh * ldh2 + c - args.csplit] : H[rh * ldh + c];
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
    if (threadIdx.x == 0) TN_G(slab)[((size_t)blockIdx.y * args.n_out + o) * (args.k_in + 4) + c] = red[0];
}

// grad_w[o * ldw + c] += sum_s slab[s][o][c];  grad_b[o] += sum_s slab[s][o][k_in]
// over the non-empty chunks, in chunk order; 8 lanes share one output element.
__global__ __launch_bounds__(256) void slab_reduce_kernel(TnArgs args) {
    const int kpad = args.k_in + 4;
    const int64_t total = (int64_t)args.n_out * (args.k_in + 1);
    const int64_t t = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 3;
    const int l = threadIdx.x & 7;
    const int* rbp = TN_G(row_begin);
    const int* rep = TN_G(row_end);
    const int rb = rbp ? *rbp : 0;
    const int re = rep ? *rep : (int)TN_G(m_static);
    int nvalid = (re - rb + args.chunk - 1) / args.chunk;
    nvalid = nvalid < 0 ? 0 : (nvalid > args.nsplit ? args.nsplit : nvalid);
    float s = 0.f;
    int o = 0, c = 0;
    if (t < total) {
        o = (int)(t / (args.k_in + 1));
        c = (int)(t % (args.k_in + 1));
        const float* p = TN_G(slab) + (size_t)o * kpad + c;
        const size_t stride = (size_t)args.n_out * kpad;
        for (int b = 0; b < args.nbatch; ++b)
            for (int i = l; i < nvalid; i += 8) s += p[((size_t)b * args.nsplit + i) * stride];
    }
    s += __shfl_xor(s, 1, 64);
    s += __shfl_xor(s, 2, 64);
    s += __shfl_xor(s, 4, 64);
    if (t < total && l == 0) {
        if (c < args.k_in) {
            float* gw = TN_G(grad_w);
            if (gw) gw[(int64_t)o * TN_G(ldw) + c] += s;
        } else {
            float* gb = TN_G(grad_b);
            if (gb) gb[o] += s;
        }
    }
}

static bool al16t(const void* p) { return (((uintptr_t)p) & 15) == 0; }

size_t tn_slab_floats(int n_out, int k_in, int64_t m_upper, int nbatch) {
    TnArgs a = {};
    a.n_out = n_out;
    a.k_in = k_in;
    a.m_upper = m_upper;
    a.nbatch = nbatch;
    tn_plan(a);
    return (size_t)a.nbatch * a.nsplit * n_out * (k_in + 4);
}

void tn_plan(TnArgs& a) {
    // enough row chunks that tiles x batches x chunks fills the chip a few times over, but never chunks so
    // short that the slab traffic (chunks n_out k_in) rivals the operand traffic (rows (n_out + k_in))
    if (a.nbatch < 1) a.nbatch = 1;
    int tiles = ((a.n_out + TBM - 1) / TBM) * ((a.k_in + TBN - 1) / TBN) * a.nbatch;
    int target = 1536 / (tiles > 0 ? tiles : 1);
    if (target < 1) target = 1;
    if (target > 128) target = 128;
    int64_t chunk = (a.m_upper + target - 1) / target;
    chunk = (chunk + TBK - 1) / TBK * TBK;
    if (chunk < 128) chunk = 128;
    a.chunk = (int)chunk;
    a.nsplit = (int)((a.m_upper + chunk - 1) / chunk);
    if (a.nsplit < 1) a.nsplit = 1;
}

int launch_gemm_tn(const TnArgs& a_in, hipStream_t s) {
    TnArgs a = a_in;
    MPN_CHECK_ARG(a.ngroups == 1 || a.ngroups == 2, "gemm_tn: ngroups");
    MPN_CHECK_ARG(a.n_out >= 1 && a.k_in >= 1 && a.csplit >= 0 && a.csplit <= a.k_in, "gemm_tn: dims");
    if (a.m_upper <= 0) return MPNHIP_OK;
    if (a.nbatch < 1) a.nbatch = 1;
    bool fast = (a.n_out % 4 == 0) && (a.k_in % 4 == 0) && (a.csplit % 4 == 0);
    for (int i = 0; i < a.ngroups; ++i) {
        const TnGroup& g = a.g[i];
        MPN_CHECK_ARG(g.dZ && g.H && (a.csplit == a.k_in || g.H2) && g.slab, "gemm_tn: null operand");
        fast = fast && !g.dz_idx && !g.h_idx && al16t(g.dZ) && g.ldz % 4 == 0 && al16t(g.H) && g.ldh % 4 == 0 &&
               (!g.H2 || (al16t(g.H2) && g.ldh2 % 4 == 0)) && g.z_bstride % 4 == 0 && g.h_bstride % 4 == 0 &&
               g.h2_bstride % 4 == 0;
    }
    tn_plan(a);
    if (fast) {
        int tiles = ((a.n_out + TBM - 1) / TBM) * ((a.k_in + TBN - 1) / TBN);
        hipLaunchKernelGGL(gemm_tn_kernel, dim3(tiles, a.nsplit * a.nbatch, a.ngroups), dim3(TNT), 0, s, a);
    } else {
        hipLaunchKernelGGL(gemm_tn_generic_kernel, dim3((unsigned)(a.n_out * (a.k_in + 1)), a.nsplit * a.nbatch, a.ngroups),
                           dim3(256), 0, s, a);
    }
    MPN_LAUNCH_CHECK();
    int64_t total = (int64_t)a.n_out * (a.k_in + 1) * 8;
    hipLaunchKernelGGL(slab_reduce_kernel, dim3((unsigned)((total + 255) / 256), 1, a.ngroups), dim3(256), 0, s, a);
    MPN_LAUNCH_CHECK();
    return MPNHIP_OK;
}

}  // namespace mpnhip
