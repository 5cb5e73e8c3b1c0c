// fp32 MFMA GEMM with fused prologue / epilogue for the MPN hot path (gfx950).
//
// One kernel template covers every dense product of the path:
//   C[m, n] = act( sum_k A[m, k] * B[k, n] + bias[n] + G1[i1(m)][n] + G2[i2(m)][n] ) (* mask) (+ C)
// Arithmetic: v_mfma_f32_32x32x2_f32 (exact fp32 FMA chain, 64 FLOP/clk/SIMD).
// Tiling: 256 threads = 4 waves; block tile BM x BN, K step 32; operands are staged
// global -> registers -> LDS in k-major images As[k][m], Bs[k][n] (pitch == 1 mod 8 dwords), so
//   * the transposing ds_write_b32 of a K-contiguous operand hits 32 distinct banks per half wave,
//   * every MFMA operand fetch is one conflict-free ds_read_b32 (lanes 0-31 consecutive m, lanes
//     32-63 the next k),
// double buffered with one barrier per K step; the next tile's global loads are in flight while the
// current tile's MFMAs issue.
//
// The gather-add epilogue is what makes "project-then-gather" possible: the node
// halves of the edge / flow MLP's first layer are computed once per NODE and added per edge here,
// instead of gathering [E, 4dn] rows and multiplying them per edge as the reference does
// (mpn.py:69,87,93).
#include "common.h"

namespace mpnhip {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BK = 32;
constexpr int NTHREADS = 256;

template <int BM, int BN, int WM, int WN, int ALAY, int BLAY>
__global__ __launch_bounds__(NTHREADS) void gemm_kernel(GemmArgs args) {
    constexpr int PA = BM + 1;  // LDS pitches (dwords); BM, BN are multiples of 8 -> pitch % 8 == 1
    constexpr int PB = BN + 1;
    constexpr int TM = BM / WM / 32;  // 32x32 MFMA tiles per wave
    constexpr int TN = BN / WN / 32;
    static_assert(WM * WN == 4, "4 waves");
    static_assert(TM >= 1 && TN >= 1, "tile too small");
    constexpr int A_F4 = BM * BK / 4 / NTHREADS;  // float4 loads per thread (K-contiguous A)
    constexpr int B_F4 = BN * BK / 4 / NTHREADS;
    constexpr int A_DW = BM * BK / NTHREADS;  // dword loads per thread (M-contiguous A)
    constexpr int B_DW = BN * BK / NTHREADS;
    static_assert(A_F4 >= 1 && B_F4 >= 1, "tile too small for the loader");

    __shared__ float smem[2 * BK * (PA + PB)];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 31, lh = lane >> 5;

    // ---- which group / which rows -------------------------------------------------------------
    int grp = 0;
    int row0, row_end;
    {
        const GemmGroup& g0 = args.g[0];
        int b0 = g0.row_begin ? *g0.row_begin : 0;
        int e0 = g0.row_end ? *g0.row_end : (int)g0.m_static;
        int nb0 = (e0 - b0 + BM - 1) / BM;
        if (nb0 < 0) nb0 = 0;
        int by = blockIdx.y;
        if (by < nb0) {
            row0 = b0 + by * BM;
            row_end = e0;
        } else {
            if (args.ngroups < 2) return;
            grp = 1;
            const GemmGroup& g1 = args.g[1];
            int b1 = g1.row_begin ? *g1.row_begin : 0;
            int e1 = g1.row_end ? *g1.row_end : (int)g1.m_static;
            row0 = b1 + (by - nb0) * BM;
            row_end = e1;
            if (row0 >= row_end) return;
        }
    }
    const GemmGroup& G = args.g[grp];
    const int col0 = blockIdx.x * BN;
    const int N = args.N, K = args.K, ksplit = args.ksplit;

    // ---- loader set-up --------------------------------------------------------------------------
    // K-contiguous operand: thread covers rows (tid/8 + 32 j), k offset (tid%8)*4 of each K step.
    const int ld_r = tid >> 3, ld_k4 = (tid & 7) * 4;
    int64_t a_off[ALAY == A_KCONTIG ? A_F4 : 1], a_off2[ALAY == A_KCONTIG ? A_F4 : 1];
    bool a_ok[ALAY == A_KCONTIG ? A_F4 : 1];
    if (ALAY == A_KCONTIG) {
#pragma unroll
        for (int j = 0; j < A_F4; ++j) {
            int r = row0 + ld_r + 32 * j;
            a_ok[j] = r < row_end;
            int64_t ri = a_ok[j] ? (G.a_idx ? (int64_t)G.a_idx[r] : (int64_t)r) : 0;
            a_off[j] = ri * G.lda;
            a_off2[j] = ri * G.lda2;
        }
    }
    int64_t b_off[BLAY == B_KCONTIG ? B_F4 : 1];
    bool b_ok[BLAY == B_KCONTIG ? B_F4 : 1];
    if (BLAY == B_KCONTIG) {
#pragma unroll
        for (int j = 0; j < B_F4; ++j) {
            int n = col0 + ld_r + 32 * j;
            b_ok[j] = n < N;
            b_off[j] = (int64_t)(b_ok[j] ? n : 0) * G.ldb;
        }
    }
    const bool a_vec = ((G.lda & 3) == 0) && ((G.lda2 & 3) == 0) && ((ksplit & 3) == 0) &&
                       ((((uintptr_t)G.A) & 15) == 0) && ((((uintptr_t)G.A2) & 15) == 0);
    const bool b_vec = ((G.ldb & 3) == 0) && ((((uintptr_t)G.B) & 15) == 0);

    float4 a_reg[ALAY == A_KCONTIG ? A_F4 : (A_DW + 3) / 4];
    float4 b_reg[BLAY == B_KCONTIG ? B_F4 : (B_DW + 3) / 4];

    auto load_tile = [&](int kt) {
        const int kbase = kt * BK;
        if (ALAY == A_KCONTIG) {
            const int k = kbase + ld_k4;
            const bool seg2 = k >= ksplit;
            const float* base = seg2 ? G.A2 : G.A;
            const int kk = seg2 ? k - ksplit : k;
            const int klim = seg2 ? K - ksplit : ksplit;  // elements available in this segment
#pragma unroll
            for (int j = 0; j < A_F4; ++j) {
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (a_ok[j] && kk < klim) {
                    const float* p = base + (seg2 ? a_off2[j] : a_off[j]) + kk;
                    if (a_vec && kk + 3 < klim) {
                        v = *reinterpret_cast<const float4*>(p);
                    } else {
                        // ragged tail (or unaligned operand): element-wise, may cross into segment 2
                        float t[4];
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            int kq = k + q;
                            float val = 0.f;
                            if (kq < K) {
                                bool s2 = kq >= ksplit;
                                const float* bq = s2 ? G.A2 : G.A;
                                val = bq[(s2 ? a_off2[j] : a_off[j]) + (s2 ? kq - ksplit : kq)];
                            }
                            t[q] = val;
                        }
                        v = make_float4(t[0], t[1], t[2], t[3]);
                    }
                }
                a_reg[j] = v;
            }
        } else {
            // A stored [k][m]: thread covers m = tid % BM, k = tid / BM + (256/BM) * j
            constexpr int KSTEP = NTHREADS / BM;
            const int m = row0 + (tid % BM);
            float* ar = reinterpret_cast<float*>(a_reg);
#pragma unroll
            for (int j = 0; j < A_DW; ++j) {
                int k = kbase + tid / BM + KSTEP * j;
                ar[j] = (m < row_end && k < K) ? G.A[(int64_t)k * G.lda + m] : 0.f;
            }
        }
        if (BLAY == B_KCONTIG) {
            const int k = kbase + ld_k4;
#pragma unroll
            for (int j = 0; j < B_F4; ++j) {
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (b_ok[j] && k < K) {
                    const float* p = G.B + b_off[j] + k;
                    if (b_vec && k + 3 < K) {
                        v = *reinterpret_cast<const float4*>(p);
                    } else {
                        v.x = p[0];
                        v.y = (k + 1 < K) ? p[1] : 0.f;
                        v.z = (k + 2 < K) ? p[2] : 0.f;
                        v.w = (k + 3 < K) ? p[3] : 0.f;
                    }
                }
                b_reg[j] = v;
            }
        } else {
            constexpr int KSTEP = NTHREADS / BN;
            const int n = col0 + (tid % BN);
            float* br = reinterpret_cast<float*>(b_reg);
#pragma unroll
            for (int j = 0; j < B_DW; ++j) {
                int k = kbase + tid / BN + KSTEP * j;
                br[j] = (n < N && k < K) ? G.B[(int64_t)k * G.ldb + n] : 0.f;
            }
        }
    };

    auto store_tile = [&](int buf) {
        float* As = smem + buf * BK * (PA + PB);
        float* Bs = As + BK * PA;
        if (ALAY == A_KCONTIG) {
#pragma unroll
            for (int j = 0; j < A_F4; ++j) {
                int r = ld_r + 32 * j;
                As[(ld_k4 + 0) * PA + r] = a_reg[j].x;
                As[(ld_k4 + 1) * PA + r] = a_reg[j].y;
                As[(ld_k4 + 2) * PA + r] = a_reg[j].z;
                As[(ld_k4 + 3) * PA + r] = a_reg[j].w;
            }
        } else {
            constexpr int KSTEP = NTHREADS / BM;
            const float* ar = reinterpret_cast<const float*>(a_reg);
#pragma unroll
            for (int j = 0; j < A_DW; ++j) As[(tid / BM + KSTEP * j) * PA + (tid % BM)] = ar[j];
        }
        if (BLAY == B_KCONTIG) {
#pragma unroll
            for (int j = 0; j < B_F4; ++j) {
                int r = ld_r + 32 * j;
                Bs[(ld_k4 + 0) * PB + r] = b_reg[j].x;
                Bs[(ld_k4 + 1) * PB + r] = b_reg[j].y;
                Bs[(ld_k4 + 2) * PB + r] = b_reg[j].z;
                Bs[(ld_k4 + 3) * PB + r] = b_reg[j].w;
            }
        } else {
            constexpr int KSTEP = NTHREADS / BN;
            const float* br = reinterpret_cast<const float*>(b_reg);
#pragma unroll
            for (int j = 0; j < B_DW; ++j) Bs[(tid / BN + KSTEP * j) * PB + (tid % BN)] = br[j];
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = (K + BK - 1) / BK;
    if (nk > 0) {
        load_tile(0);
        store_tile(0);
    }
    __syncthreads();

    const int a_base = wm * (BM / WM) + li;
    const int b_base = wn * (BN / WN) + li;
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) load_tile(kt + 1);
        const float* As = smem + buf * BK * (PA + PB);
        const float* Bs = As + BK * PA;
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            float a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = As[(kk + lh) * PA + a_base + 32 * i];
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = Bs[(kk + lh) * PB + b_base + 32 * j];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < nk) store_tile(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue: D[i][j] of a 32x32 tile: j = lane & 31, i = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
#pragma unroll
    for (int tj = 0; tj < TN; ++tj) {
        const int n = col0 + wn * (BN / WN) + tj * 32 + li;
        const bool n_ok = n < N;
        const float bias = (n_ok && G.bias) ? G.bias[n] : 0.f;
#pragma unroll
        for (int ti = 0; ti < TM; ++ti) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = row0 + wm * (BM / WM) + ti * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (m < row_end && n_ok) {
                    float v = acc[ti][tj][r] + bias;
                    if (G.G1) v += G.G1[(int64_t)(G.g1_idx ? G.g1_idx[m] : m) * G.ldg1 + n];
                    if (G.G2) v += G.G2[(int64_t)(G.g2_idx ? G.g2_idx[m] : m) * G.ldg2 + n];
                    if (args.relu) v = fmaxf(v, 0.f);
                    if (G.mask) v = (G.mask[(int64_t)m * G.ldmask + n] > 0.f) ? v : 0.f;
                    float* cp = G.C + (int64_t)(G.c_idx ? G.c_idx[m] : m) * G.ldc + n;
                    if (args.accumulate) v += *cp;
                    *cp = v;
                }
            }
        }
    }
}

template <int BM, int BN, int WM, int WN>
static int launch_cfg(const GemmArgs& a, int al, int bl, hipStream_t s) {
    int64_t nby = (a.m_upper + BM - 1) / BM + (a.ngroups > 1 ? 1 : 0);
    if (nby <= 0 || a.N <= 0) return MPNHIP_OK;
    dim3 grid((a.N + BN - 1) / BN, (unsigned)nby, 1);
    if (al == A_KCONTIG && bl == B_KCONTIG)
        hipLaunchKernelGGL((gemm_kernel<BM, BN, WM, WN, A_KCONTIG, B_KCONTIG>), grid, dim3(NTHREADS), 0, s, a);
    else if (al == A_KCONTIG && bl == B_NCONTIG)
        hipLaunchKernelGGL((gemm_kernel<BM, BN, WM, WN, A_KCONTIG, B_NCONTIG>), grid, dim3(NTHREADS), 0, s, a);
    else if (al == A_MCONTIG && bl == B_NCONTIG)
        hipLaunchKernelGGL((gemm_kernel<BM, BN, WM, WN, A_MCONTIG, B_NCONTIG>), grid, dim3(NTHREADS), 0, s, a);
    else {
        set_error("gemm: unsupported operand layout %d/%d", al, bl);
        return MPNHIP_ERR_UNSUPPORTED;
    }
    MPN_LAUNCH_CHECK();
    return MPNHIP_OK;
}

int launch_gemm(const GemmArgs& a, int al, int bl, hipStream_t s) {
    MPN_CHECK_ARG(a.ngroups == 1 || a.ngroups == 2, "gemm: ngroups %d", a.ngroups);
    MPN_CHECK_ARG(a.K >= 0 && a.N >= 0 && a.ksplit >= 0 && a.ksplit <= a.K, "gemm: bad N/K/ksplit");
    MPN_CHECK_ARG(a.m_upper < (int64_t)2147483647 - 256, "gemm: too many rows for int32 indexing");
    if (a.m_upper <= 0 || a.N == 0) return MPNHIP_OK;  // nothing to compute (empty graph)
    for (int i = 0; i < a.ngroups; ++i) {
        MPN_CHECK_ARG(a.g[i].A && a.g[i].B && a.g[i].C, "gemm: null operand");
        MPN_CHECK_ARG(a.ksplit == a.K || a.g[i].A2, "gemm: ksplit without a second A segment");
        MPN_CHECK_ARG(al == A_KCONTIG || (!a.g[i].a_idx && a.ksplit == a.K), "gemm: M-contiguous A is plain");
    }
    // tile choice: widest N tile that is not mostly padding; shrink M tile when the grid would not
    // cover the 256 CUs.
    const int N = a.N;
    const int64_t M = a.m_upper;
    if (N > 64) {
        int64_t blocks = ((M + 127) / 128) * ((N + 127) / 128);
        if (blocks >= 192 || M > 4096) return launch_cfg<128, 128, 2, 2>(a, al, bl, s);
        return launch_cfg<64, 64, 2, 2>(a, al, bl, s);
    }
    if (N > 32) {
        int64_t blocks = (M + 127) / 128;
        if (blocks >= 192) return launch_cfg<128, 64, 2, 2>(a, al, bl, s);
        return launch_cfg<64, 64, 2, 2>(a, al, bl, s);
    }
    return launch_cfg<128, 32, 4, 1>(a, al, bl, s);
}

int linear(const float* x, int64_t ldx, const float* w, const float* b, float* y, int64_t ldy, int64_t m, int n, int k,
           int relu, hipStream_t stream) {
    GemmArgs a = {};
    a.ngroups = 1;
    a.N = n;
    a.K = k;
    a.ksplit = k;
    a.relu = relu;
    a.m_upper = m;
    GemmGroup& g = a.g[0];
    g.A = x;
    g.lda = ldx;
    g.B = w;
    g.ldb = k;
    g.bias = b;
    g.C = y;
    g.ldc = ldy;
    g.m_static = m;
    return launch_gemm(a, A_KCONTIG, B_KCONTIG, stream);
}

}  // namespace mpnhip
