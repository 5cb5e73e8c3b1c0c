// fp32 MFMA GEMM with fused prologue / epilogue for the MPN hot path (gfx950).
//
// One kernel template covers every dense product of the forward path (SURVEY.md section 2.4,
// K2/K4/K6/K8/K9) and the activation-gradient products of the backward path:
//   C[m, n] = mask( act( sum_k A[m, k] * B[k, n] + bias[n] + G1[i1(m)][n] + G2[i2(m)][n] ) (+ C) )
// Arithmetic: v_mfma_f32_32x32x2_f32 (exact fp32 FMA chain, 64 FLOP/clk/SIMD = the fp32 peak).
//
// Geometry: 256 threads = 4 waves arranged WM x WN; every wave owns a 32 x (32 TN) strip of C
// (TN accumulator tiles of 16 VGPRs), so the block tile is (32 WM) x (32 TN WN).  TN is chosen per
// launch so that the MLP widths of the model (320, 224, 128, 64, 1088 ...) are covered without padding.
// K step 32: both operands are staged global -> registers -> LDS as k-major images As[k][m], Bs[k][n]
//   * K-contiguous operands are written transposed with ds_write_b32 at pitch == 1 (mod 8) dwords:
//     32 distinct banks per half wave;
//   * every MFMA operand fetch is one conflict-free ds_read_b32 (lanes 0-31 consecutive m / n,
//     lanes 32-63 the next k).
// The next K step's global loads are issued before the current step's MFMAs and written to LDS after
// them (register prefetch); fp32 MFMAs are slow enough (64 clk each) that one LDS stage suffices and
// the smaller footprint buys 3-4 resident blocks per CU.  All global loads are unconditional (clamped
// addresses, zero-selected values): no branch, hence no vmcnt(0) serialisation between them.
//
// Epilogue: each wave passes its 32x32 accumulator tiles through a private LDS patch and leaves it
// as whole 128-byte row segments: bias, the two row gathers, ReLU(-mask) and the store are all 16-byte
// vector accesses of contiguous row pieces -- the access shape gathers of whole rows want.
//
// The gather-add epilogue is what makes "project-then-gather" possible: the node
// halves of the edge / flow MLP's first layer are computed once per NODE and added per edge here,
// instead of gathering [E, 4dn] rows and multiplying them per edge as the reference does
// (mpn.py:69,87,93).
#include <cstdlib>

#include "common.h"

namespace mpnhip {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

// Operand precision of the MFMA products of the calling thread's current forward (set by mpnhip_forward from
// mpnhip_model.precision): 0 = fp32 operands (v_mfma_f32_32x32x2_f32), 1 = operands rounded to bf16 (RNE) when they
// are staged into LDS, fp32 accumulation (v_mfma_f32_32x32x16_bf16) -- BASELINE.json's "bf16 MLP GEMMs on MFMA" mode;
// 2 = MPNHIP_PREC_FP32_SPLIT: fp32 operands split into three bf16 pieces as they are staged, six products per multiply.
static thread_local int g_precision = 0;
void set_gemm_precision(int p) { g_precision = p; }
int gemm_precision() { return g_precision; }

constexpr int BK = 32;
constexpr int PKB = 40;  // bf16 images: row pitch in elements (32 k + 8 pad = 80 bytes: 16-byte aligned rows)
constexpr int NTHREADS = 256;

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
// component-wise select (a float4 ?: makes hipcc spill both operands to scratch and select the address)
__device__ __forceinline__ float4 keep4(bool ok, float4 v) {
    v.x = ok ? v.x : 0.f; v.y = ok ? v.y : 0.f; v.z = ok ? v.z : 0.f; v.w = ok ? v.w : 0.f;
    return v;
}

// x = h + m + l with every piece a bfloat16 (RNE): h = bf16(x), m = bf16(x - h), l = bf16(x - h - m).  The residuals are
// exact in fp32 and the pieces carry 8 + 8 + 8 significant bits, so the three-term sum reproduces a finite fp32 x exactly
// (up to bf16 underflow of l).  5.5 VALU operations per element (v_cvt_pk_bf16_f32 converts two at a time).
__device__ __forceinline__ void split3(float4 v, bf16x4& h, bf16x4& m, bf16x4& l) {
    h = bf16x4{(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};
    const float r0 = v.x - (float)h[0], r1 = v.y - (float)h[1], r2 = v.z - (float)h[2], r3 = v.w - (float)h[3];
    m = bf16x4{(__bf16)r0, (__bf16)r1, (__bf16)r2, (__bf16)r3};
    l = bf16x4{(__bf16)(r0 - (float)m[0]), (__bf16)(r1 - (float)m[1]), (__bf16)(r2 - (float)m[2]), (__bf16)(r3 - (float)m[3])};
}

// waves per SIMD the register allocation must leave room for (accumulators: 16 TN registers)
constexpr int min_waves(int tn) { return tn <= 2 ? 4 : (tn <= 4 ? 3 : 2); }

// BF16: same loader / epilogue; the LDS images are row-major bf16 [m][k], [n][k] (k contiguous: the fp32 rows are
// rounded and written without a transpose) and every MFMA operand is one ds_read_b128 of 8 consecutive k.
// ... and what the LDS images allow: the split variant's three-piece images are 6 (BM + BN) PKB bytes per block
constexpr int min_waves_lds(int wm, int wn, int tn, int prec) {
    const int by_regs = min_waves(tn);
    if (prec != 2) return by_regs;
    const int lds = 3 * (32 * wm + 32 * tn * wn) * PKB * 2;
    const int by_lds = 163840 / lds < 1 ? 1 : 163840 / lds;
    return by_lds < by_regs ? by_lds : by_regs;
}

template <int WM, int WN, int TN, int BLAY, int PREC = 0>
__global__ __launch_bounds__(NTHREADS, min_waves_lds(WM, WN, TN, PREC)) void gemm_kernel(GemmArgs args) {
    static_assert(WM * WN == 4, "4 waves");
    constexpr bool BF16 = PREC == 1, SPLIT = PREC == 2, LOWP = PREC != 0;
    constexpr int NIMG = SPLIT ? 3 : 1;                        // bf16 images per operand
    static_assert(!LOWP || BLAY == B_KCONTIG, "bf16 operands: nn.Linear weights only");
    constexpr int BM = 32 * WM;
    constexpr int BN = 32 * TN * WN;
    constexpr int PA = BM + 1;                                 // pitch % 8 == 1 (transposing stores)
    constexpr int PB = BLAY == B_KCONTIG ? BN + 1 : BN + 4;    // N-contiguous B keeps 16-byte rows
    constexpr int A_F4 = BM / 32;                              // float4 loads per thread and K step
    constexpr int B_F4 = BN / 32;
    constexpr int PATCH = 32 * 36;                             // per-wave epilogue patch [32][36]
    constexpr int TILE_FLOATS = LOWP ? NIMG * (BM + BN) * PKB / 2 : BK * (PA + PB);
    constexpr int SMEM_FLOATS = TILE_FLOATS > 4 * PATCH ? TILE_FLOATS : 4 * PATCH;

    __shared__ __attribute__((aligned(16))) float smem[SMEM_FLOATS];
    float* As = smem;
    float* Bs = smem + BK * PA;
    __bf16* const Ab = reinterpret_cast<__bf16*>(smem);   // BF16 / SPLIT: NIMG x [BM][PKB], then NIMG x [BN][PKB]
    __bf16* const Bb = Ab + NIMG * BM * PKB;

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

    // ---- loader set-up: thread covers tile rows (tid/8 + 32 j) at k offset (tid%8)*4 --------------
    // 32-bit element offsets from the (scalar) operand bases keep the address state small
    const int ld_r = tid >> 3, ld_k4 = (tid & 7) * 4;
    const float* const Abase = G.A;
    const float* const A2base = G.A2 ? G.A2 - ksplit : G.A;
    const float* const Bbase = G.B;
    int a_o1[A_F4], a_o2[A_F4];
#pragma unroll
    for (int j = 0; j < A_F4; ++j) {
        int r = row0 + ld_r + 32 * j;
        r = r < row_end ? r : row_end - 1;  // rows past the end are computed but never stored
        int ri = G.a_idx ? G.a_idx[r] : r;
        a_o1[j] = ri * (int)G.lda;
        a_o2[j] = G.A2 ? ri * (int)G.lda2 : a_o1[j];
    }
    // B_KCONTIG: rows are output columns n (weight rows); B_NCONTIG: thread covers k = tid/(BN/4) + .., n4
    int b_o[B_F4];
    int b_k[B_F4];  // B_NCONTIG: k row inside the tile
#pragma unroll
    for (int j = 0; j < B_F4; ++j) {
        if (BLAY == B_KCONTIG) {
            int n = col0 + ld_r + 32 * j;
            n = n < N ? n : N - 1;
            b_o[j] = n * (int)G.ldb;
            b_k[j] = 0;
        } else {
            int f = tid + NTHREADS * j;          // float4 index inside the [BK][BN/4] tile
            int kr = f / (BN / 4), n4 = f % (BN / 4);
            int n = col0 + n4 * 4;
            n = n + 3 < N ? n : (N - 4);          // N % 4 == 0 on this path; clamped columns are never stored
            b_o[j] = n;
            b_k[j] = kr;
        }
    }

    float4 a_reg[A_F4], b_reg[B_F4];
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);

    // Issue the global loads of K step kt; nothing here consumes the results (the zero-select for the K tail
    // happens in store_tile), so the loads stay in flight across the MFMA phase of the previous step.
    auto load_tile = [&](int kt) {
        const int k = kt * BK + ld_k4;
        const int kc = k < K ? k : 0;
        const bool seg2 = kc >= ksplit;
#pragma unroll
        for (int j = 0; j < A_F4; ++j) a_reg[j] = ld4((seg2 ? A2base : Abase) + (seg2 ? a_o2[j] : a_o1[j]) + kc);
#pragma unroll
        for (int j = 0; j < B_F4; ++j) {
            if (BLAY == B_KCONTIG) {
                b_reg[j] = ld4(Bbase + b_o[j] + kc);
            } else {
                const int kk = kt * BK + b_k[j];
                b_reg[j] = ld4(Bbase + b_o[j] + (kk < K ? kk : 0) * (int)G.ldb);
            }
        }
    };

    auto store_tile = [&](int kt) {
        const bool k_ok = kt * BK + ld_k4 < K;   // K % 4 == 0: a float4 is entirely in or out
        if (SPLIT) {
#pragma unroll
            for (int j = 0; j < A_F4; ++j) {
                bf16x4 h, m, l;
                split3(keep4(k_ok, a_reg[j]), h, m, l);
                __bf16* d = Ab + (ld_r + 32 * j) * PKB + ld_k4;
                *reinterpret_cast<bf16x4*>(d) = h;
                *reinterpret_cast<bf16x4*>(d + BM * PKB) = m;
                *reinterpret_cast<bf16x4*>(d + 2 * BM * PKB) = l;
            }
#pragma unroll
            for (int j = 0; j < B_F4; ++j) {
                bf16x4 h, m, l;
                split3(keep4(k_ok, b_reg[j]), h, m, l);
                __bf16* d = Bb + (ld_r + 32 * j) * PKB + ld_k4;
                *reinterpret_cast<bf16x4*>(d) = h;
                *reinterpret_cast<bf16x4*>(d + BN * PKB) = m;
                *reinterpret_cast<bf16x4*>(d + 2 * BN * PKB) = l;
            }
            return;
        }
        if (BF16) {
#pragma unroll
            for (int j = 0; j < A_F4; ++j) {
                const float4 v = keep4(k_ok, a_reg[j]);
                bf16x4 o = {(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};
                *reinterpret_cast<bf16x4*>(Ab + (ld_r + 32 * j) * PKB + ld_k4) = o;
            }
#pragma unroll
            for (int j = 0; j < B_F4; ++j) {
                const float4 v = keep4(k_ok, b_reg[j]);
                bf16x4 o = {(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};
                *reinterpret_cast<bf16x4*>(Bb + (ld_r + 32 * j) * PKB + ld_k4) = o;
            }
            return;
        }
#pragma unroll
        for (int j = 0; j < A_F4; ++j) {
            const int r = ld_r + 32 * j;
            const float4 v = keep4(k_ok, a_reg[j]);
            As[(ld_k4 + 0) * PA + r] = v.x;
            As[(ld_k4 + 1) * PA + r] = v.y;
            As[(ld_k4 + 2) * PA + r] = v.z;
            As[(ld_k4 + 3) * PA + r] = v.w;
        }
#pragma unroll
        for (int j = 0; j < B_F4; ++j) {
            if (BLAY == B_KCONTIG) {
                const int r = ld_r + 32 * j;
                const float4 v = keep4(k_ok, b_reg[j]);
                Bs[(ld_k4 + 0) * PB + r] = v.x;
                Bs[(ld_k4 + 1) * PB + r] = v.y;
                Bs[(ld_k4 + 2) * PB + r] = v.z;
                Bs[(ld_k4 + 3) * PB + r] = v.w;
            } else {
                const int f = tid + NTHREADS * j;
                const int kr = f / (BN / 4), n4 = f % (BN / 4);
                *reinterpret_cast<float4*>(&Bs[kr * PB + n4 * 4]) = keep4(kt * BK + b_k[j] < K, b_reg[j]);
            }
        }
    };

    f32x16 acc[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

    const int nk = (K + BK - 1) / BK;
    load_tile(0);
    const float* a_ptr = As + lh * PA + wm * 32 + li;
    const float* b_ptr = Bs + lh * PB + wn * 32 * TN + li;
    for (int kt = 0; kt < nk; ++kt) {
        store_tile(kt);
        __syncthreads();
        if (kt + 1 < nk) load_tile(kt + 1);
        if (SPLIT) {
            // x = h + m + l exactly representable pieces (split3); the six products of weight >= 2^-16 relative to
            // h_a h_b, smallest first, accumulate in fp32: the dropped ones (m l, l m, l l) are below 2^-24 |a b|
            const __bf16* ap = Ab + (wm * 32 + li) * PKB + lh * 8;
            const __bf16* bp = Bb + (wn * 32 * TN + li) * PKB + lh * 8;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                bf16x8 av[3], bv[3][TN];
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    av[q] = *reinterpret_cast<const bf16x8*>(ap + q * BM * PKB + kb * 16);
#pragma unroll
                    for (int j = 0; j < TN; ++j) bv[q][j] = *reinterpret_cast<const bf16x8*>(bp + q * BN * PKB + 32 * j * PKB + kb * 16);
                }
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[2], bv[0][j], acc[j], 0, 0, 0);
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[0], bv[2][j], acc[j], 0, 0, 0);
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[1], bv[1][j], acc[j], 0, 0, 0);
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[1], bv[0][j], acc[j], 0, 0, 0);
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[0], bv[1][j], acc[j], 0, 0, 0);
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[0], bv[0][j], acc[j], 0, 0, 0);
                }
            }
            __syncthreads();
            continue;
        }
        if (BF16) {
            // two 16-deep k blocks per K step; lane (i, h) supplies k = 8h .. 8h+7 of its row
            const __bf16* ap = Ab + (wm * 32 + li) * PKB + lh * 8;
            const __bf16* bp = Bb + (wn * 32 * TN + li) * PKB + lh * 8;
            bf16x8 av[2], bv[2][TN];
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                av[kb] = *reinterpret_cast<const bf16x8*>(ap + kb * 16);
#pragma unroll
                for (int j = 0; j < TN; ++j) bv[kb][j] = *reinterpret_cast<const bf16x8*>(bp + 32 * j * PKB + kb * 16);
            }
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[kb], bv[kb][j], acc[j], 0, 0, 0);
            __syncthreads();
            continue;
        }
        // operand fetch one k pair ahead of the MFMAs that use it (explicit register double buffer)
        float a_cur = a_ptr[0], b_cur[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) b_cur[j] = b_ptr[32 * j];
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            float a_nxt = 0.f, b_nxt[TN];
            if (kk + 2 < BK) {
                a_nxt = a_ptr[(kk + 2) * PA];
#pragma unroll
                for (int j = 0; j < TN; ++j) b_nxt[j] = b_ptr[(kk + 2) * PB + 32 * j];
            }
            // keep the fetch of the NEXT pair above this pair's MFMAs (hipcc otherwise sinks every ds_read to
            // just before its use and waits lgkmcnt(0) in front of each MFMA)
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur, b_cur[j], acc[j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (kk + 2 < BK) {
                a_cur = a_nxt;
#pragma unroll
                for (int j = 0; j < TN; ++j) b_cur[j] = b_nxt[j];
            }
        }
        __syncthreads();
    }

    // ---- epilogue ------------------------------------------------------------------------------
    // D[i][j] of a 32x32 tile: j = lane & 31, i = (r & 3) + 8 (r >> 2) + 4 (lane >> 5).
    // The wave's patch is [32 rows][36]: written column-wise (conflict free), read back as float4 rows:
    // lane -> row (lane / 8 + 8 p), columns 4 (lane % 8) .. +3.
    float* patch = smem + wave * PATCH;
    const int er = lane >> 3, ec = (lane & 7) * 4;
    const bool vec_ok = args.epi_vec != 0;
    int m_row[4];
    int g1_off[4], g2_off[4], c_off[4];
    bool m_ok[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        int m = row0 + wm * 32 + er + 8 * p;
        m_ok[p] = m < row_end;
        int mc = m_ok[p] ? m : row_end - 1;
        m_row[p] = mc;
        g1_off[p] = (G.g1_idx ? G.g1_idx[mc] : mc) * (int)G.ldg1;
        g2_off[p] = (G.g2_idx ? G.g2_idx[mc] : mc) * (int)G.ldg2;
        c_off[p] = (G.c_idx ? G.c_idx[mc] : mc) * (int)G.ldc;
    }
#pragma unroll
    for (int tj = 0; tj < TN; ++tj) {
        const int ncol0 = col0 + wn * 32 * TN + tj * 32;
        if (ncol0 >= N) break;  // wave-uniform
#pragma unroll
        for (int r = 0; r < 16; ++r) patch[((r & 3) + 8 * (r >> 2) + 4 * lh) * 36 + li] = acc[tj][r];
        // same wave wrote and reads: LDS operations of one wave complete in order
        const int n = ncol0 + ec;
        if (vec_ok) {
            const bool n_ok = n < N;          // N % 4 == 0 here
            const int nc = n_ok ? n : N - 4;
            const float4 bias = G.bias ? ld4(G.bias + nc) : zero4;
#pragma unroll
            for (int ph = 0; ph < 4; ph += 2) {
                float4 v[2], g1[2], g2[2], mk[2], old[2];
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int p = ph + q;
                    v[q] = *reinterpret_cast<const float4*>(&patch[(er + 8 * p) * 36 + ec]);
                    g1[q] = G.G1 ? ld4(G.G1 + g1_off[p] + nc) : zero4;
                    g2[q] = G.G2 ? ld4(G.G2 + g2_off[p] + nc) : zero4;
                    mk[q] = G.mask ? ld4(G.mask + m_row[p] * (int)G.ldmask + nc) : make_float4(1.f, 1.f, 1.f, 1.f);
                    old[q] = args.accumulate ? ld4(G.C + c_off[p] + nc) : zero4;
                }
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int p = ph + q;
                    float4 o;
                    o.x = v[q].x + bias.x + g1[q].x + g2[q].x;
                    o.y = v[q].y + bias.y + g1[q].y + g2[q].y;
                    o.z = v[q].z + bias.z + g1[q].z + g2[q].z;
                    o.w = v[q].w + bias.w + g1[q].w + g2[q].w;
                    if (args.relu) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
                    o.x += old[q].x; o.y += old[q].y; o.z += old[q].z; o.w += old[q].w;
                    o.x = mk[q].x > 0.f ? o.x : 0.f; o.y = mk[q].y > 0.f ? o.y : 0.f;
                    o.z = mk[q].z > 0.f ? o.z : 0.f; o.w = mk[q].w > 0.f ? o.w : 0.f;
                    if (m_ok[p] && n_ok) *reinterpret_cast<float4*>(G.C + c_off[p] + nc) = o;
                }
            }
        } else {
#pragma unroll
            for (int p = 0; p < 4; ++p) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int nq = n + q;
                    const bool ok = m_ok[p] && nq < N;
                    const int nc = nq < N ? nq : N - 1;
                    float o = patch[(er + 8 * p) * 36 + ec + q];
                    if (G.bias) o += G.bias[nc];
                    if (G.G1) o += G.G1[g1_off[p] + nc];
                    if (G.G2) o += G.G2[g2_off[p] + nc];
                    if (args.relu) o = fmaxf(o, 0.f);
                    if (args.accumulate) o += G.C[c_off[p] + nc];
                    if (G.mask) o = G.mask[m_row[p] * (int)G.ldmask + nc] > 0.f ? o : 0.f;
                    if (ok) G.C[c_off[p] + nc] = o;
                }
            }
        }
    }
}

// Any shape / alignment, one thread per output element: the K = 6 / 18-wide encoder layers of the
// d = 32 configuration and other operands that are not 16-byte aligned.  Not a hot kernel.
__global__ __launch_bounds__(NTHREADS) void gemm_generic_kernel(GemmArgs args, int b_layout, int bf16) {
    const int N = args.N, K = args.K, ksplit = args.ksplit;
    for (int grp = 0; grp < args.ngroups; ++grp) {
        const GemmGroup& G = args.g[grp];
        const int b = G.row_begin ? *G.row_begin : 0;
        const int e = G.row_end ? *G.row_end : (int)G.m_static;
        const int64_t total = (int64_t)(e - b) * N;
        for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
            const int m = b + (int)(t / N), n = (int)(t % N);
            const int64_t ri = G.a_idx ? G.a_idx[m] : m;
            const float* a1 = G.A + ri * G.lda;
            const float* a2 = G.A2 ? G.A2 + ri * G.lda2 - ksplit : a1;
            float acc = 0.f;
            for (int k = 0; k < K; ++k) {
                float av = (k >= ksplit ? a2 : a1)[k];
                float bv = b_layout == B_KCONTIG ? G.B[(int64_t)n * G.ldb + k] : G.B[(int64_t)k * G.ldb + n];
                if (bf16) { av = (float)(__bf16)av; bv = (float)(__bf16)bv; }  // same operand rounding as the MFMA path
                acc = fmaf(av, bv, acc);
            }
            if (G.bias) acc += G.bias[n];
            if (G.G1) acc += G.G1[(int64_t)(G.g1_idx ? G.g1_idx[m] : m) * G.ldg1 + n];
            if (G.G2) acc += G.G2[(int64_t)(G.g2_idx ? G.g2_idx[m] : m) * G.ldg2 + n];
            if (args.relu) acc = fmaxf(acc, 0.f);
            float* cp = G.C + (int64_t)(G.c_idx ? G.c_idx[m] : m) * G.ldc + n;
            if (args.accumulate) acc += *cp;
            if (G.mask) acc = G.mask[(int64_t)m * G.ldmask + n] > 0.f ? acc : 0.f;
            *cp = acc;
        }
    }
}

// K <= 8 (the edge encoder's first layer: edge_attr [E, 6] read through the sort permutation -> 18 d / 32 features): a thread
// owns four neighbouring outputs of a row -- its 4 x K weights and 4 biases stay in registers while it walks rows -- so a row costs
// K broadcast loads and one 16-byte store.  The one-thread-per-output kernel above took 414 us at cfg-E (57.6 M outputs, 12 loads
// and a 64-bit division each) for a layer that writes 230 MB; bf16 mode rounds the operands like the MFMA path (products of two
// bf16 values are exact in fp32, the sum runs in fp32).
__global__ __launch_bounds__(NTHREADS) void gemm_smallk_kernel(GemmArgs args, int bf16) {
    const GemmGroup& G = args.g[0];
    const int N = args.N, K = args.K;
    const int tpr = N >> 2, rpp = NTHREADS / tpr;
    const int n4 = threadIdx.x % tpr, r0 = threadIdx.x / tpr;
    if (r0 >= rpp) return;
    const int n = 4 * n4;
    float w[4][8], bias[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        bias[j] = G.bias ? G.bias[n + j] : 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            float v = k < K ? G.B[(int64_t)(n + j) * G.ldb + k] : 0.f;
            w[j][k] = bf16 ? (float)(__bf16)v : v;
        }
    }
    const int b = G.row_begin ? *G.row_begin : 0;
    const int e = G.row_end ? *G.row_end : (int)G.m_static;
    for (int64_t m = b + (int64_t)blockIdx.x * rpp + r0; m < e; m += (int64_t)gridDim.x * rpp) {
        const int64_t ri = G.a_idx ? G.a_idx[m] : m;
        const float* ar = G.A + ri * G.lda;
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (k < K) {
                float av = ar[k];
                if (bf16) av = (float)(__bf16)av;
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] = fmaf(av, w[j][k], acc[j]);
            }
        }
        float4 o = make_float4(acc[0] + bias[0], acc[1] + bias[1], acc[2] + bias[2], acc[3] + bias[3]);
        if (args.relu) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
        *reinterpret_cast<float4*>(G.C + m * G.ldc + n) = o;
    }
}

template <int WM, int WN, int TN>
static int launch_cfg(const GemmArgs& a, int bl, hipStream_t s) {
    constexpr int BM = 32 * WM, BN = 32 * TN * WN;
    int64_t nby = (a.m_upper + BM - 1) / BM + (a.ngroups > 1 ? 1 : 0);
    dim3 grid((a.N + BN - 1) / BN, (unsigned)nby, 1);
    int prec = g_precision;
    // FP32_SPLIT (three-piece bf16 operands, six products: split3): pays where the product is large enough to be MFMA-bound --
    // measured on MI355X (tools/gemm_bench.py --check): 5000 x 1088 x 256 38.9 -> 31.6 us, 5000 x 512 x 2048 141 -> 122 us,
    // 50000 x 320 x 128 65.6 -> 48.5 us; narrow outputs (N = 128) and short K lose to the fp32 MFMA kernel's smaller LDS image
    if (prec == 2 && !(a.K >= 128 && a.N >= 256)) prec = 0;
    if (const char* e = getenv("MPNHIP_GEMM_PREC")) prec = atoi(e);  // tuning override (tools/gemm_bench.py)
    count_path(bl == B_KCONTIG && prec == 2 ? PC_GEMM_SPLIT : (bl == B_KCONTIG && prec == 1 ? PC_GEMM_BF16 : PC_GEMM_FP32));
    if (bl == B_KCONTIG && prec == 2)
        MPN_LAUNCH_PROFILED((gemm_kernel<WM, WN, TN, B_KCONTIG, 2>), grid, dim3(NTHREADS), s, a);
    else if (bl == B_KCONTIG && prec == 1)
        MPN_LAUNCH_PROFILED((gemm_kernel<WM, WN, TN, B_KCONTIG, 1>), grid, dim3(NTHREADS), s, a);
    else if (bl == B_KCONTIG)
        MPN_LAUNCH_PROFILED((gemm_kernel<WM, WN, TN, B_KCONTIG>), grid, dim3(NTHREADS), s, a);
    else
        hipLaunchKernelGGL((gemm_kernel<WM, WN, TN, B_NCONTIG>), grid, dim3(NTHREADS), 0, s, a);
    MPN_LAUNCH_CHECK();
    return MPNHIP_OK;
}

static bool al16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

int launch_gemm(const GemmArgs& a_in, int al, int bl, hipStream_t s) {
    GemmArgs a = a_in;
    MPN_CHECK_ARG(a.ngroups == 1 || a.ngroups == 2, "gemm: ngroups %d", a.ngroups);
    MPN_CHECK_ARG(a.K >= 0 && a.N >= 0 && a.ksplit >= 0 && a.ksplit <= a.K, "gemm: bad N/K/ksplit");
    MPN_CHECK_ARG(a.m_upper < (int64_t)2147483647 - 256, "gemm: too many rows for int32 indexing");
    MPN_CHECK_ARG(al == A_KCONTIG, "gemm: A must be K-contiguous (weight-gradient products use gemm_tn)");
    if (a.m_upper <= 0 || a.N == 0) return MPNHIP_OK;  // nothing to compute (empty graph)
    bool fast = (a.K % 4 == 0) && (a.ksplit % 4 == 0) && a.K > 0;
    // the MFMA kernel addresses operands with 32-bit element offsets
    const int64_t lim = (int64_t)1 << 31;
    bool epi_vec = (a.N % 4 == 0);
    for (int i = 0; i < a.ngroups; ++i) {
        const GemmGroup& g = a.g[i];
        MPN_CHECK_ARG(g.A && g.B && g.C, "gemm: null operand");
        MPN_CHECK_ARG(a.ksplit == a.K || g.A2, "gemm: ksplit without a second A segment");
        fast = fast && al16(g.A) && (g.lda % 4 == 0) && (!g.A2 || (al16(g.A2) && g.lda2 % 4 == 0)) && al16(g.B) && (g.ldb % 4 == 0);
        if (bl == B_NCONTIG) fast = fast && (a.N % 4 == 0);
        const int64_t rows = a.m_upper + 1;
        MPN_CHECK_ARG(rows * g.lda < lim && rows * g.lda2 < lim && rows * g.ldc < lim && rows * g.ldg1 < lim &&
                          rows * g.ldg2 < lim && rows * g.ldmask < lim && (int64_t)(a.N + a.K) * g.ldb < lim,
                      "gemm: operand too large for 32-bit element offsets");
        epi_vec = epi_vec && al16(g.C) && (g.ldc % 4 == 0) && (!g.bias || al16(g.bias)) &&
                  (!g.G1 || (al16(g.G1) && g.ldg1 % 4 == 0)) && (!g.G2 || (al16(g.G2) && g.ldg2 % 4 == 0)) &&
                  (!g.mask || (al16(g.mask) && g.ldmask % 4 == 0));
    }
    a.epi_vec = epi_vec ? 1 : 0;
    for (int i = 0; i < a.ngroups; ++i)
        if ((a.g[i].a16 || a.g[i].b16 || a.g[i].C16) && (!fast || !epi_vec || bl != B_KCONTIG)) {
            set_error("gemm: bf16 operand rows need the tiled bf16 kernel's alignment (K, leading dims, N multiples of 4 / 8; 16-byte bases)");
            return MPNHIP_ERR_UNSUPPORTED;
        }
    if (!fast) {
        const GemmGroup& G0 = a.g[0];
        if (a.ngroups == 1 && a.K >= 1 && a.K <= 8 && a.ksplit == a.K && bl == B_KCONTIG && !G0.A2 && !G0.G1 && !G0.G2 && !G0.mask &&
            !G0.c_idx && !a.accumulate && a.N % 4 == 0 && a.N >= 4 && a.N <= 4 * NTHREADS && G0.ldc % 4 == 0 &&
            (((uintptr_t)G0.C) & 15) == 0 && a.m_upper >= 4096 && !getenv("MPNHIP_NO_SMALLK")) {
            const int rpp = NTHREADS / (a.N / 4);
            // (a thread's 4 x K weights are 28 scalar loads: at least eight row passes per block to pay for them)
            int64_t nb = ((a.m_upper + rpp - 1) / rpp + 7) / 8;
            nb = nb > 256 * 16 ? 256 * 16 : (nb < 1 ? 1 : nb);
            hipLaunchKernelGGL(gemm_smallk_kernel, dim3((unsigned)nb), dim3(NTHREADS), 0, s, a, g_precision == 1 ? 1 : 0);
            MPN_LAUNCH_CHECK();
            return MPNHIP_OK;
        }
        int64_t total = a.m_upper * a.N;
        unsigned blocks = (unsigned)((total + NTHREADS - 1) / NTHREADS);
        if (blocks > 65535u * 16) blocks = 65535u * 16;
        hipLaunchKernelGGL(gemm_generic_kernel, dim3(blocks), dim3(NTHREADS), 0, s, a, bl, g_precision == 1 ? 1 : 0);
        MPN_LAUNCH_CHECK();
        return MPNHIP_OK;
    }
    const int N = a.N;
    const int64_t M = a.m_upper;
    const int nt = (N + 31) / 32;  // 32-wide column tiles needed
    {
        // bf16-operand mode: the 128 x 128 tiled kernel (gemm_bf16.hip) from a few thousand rows, and whenever an operand is bf16
        // rows in memory (only that kernel reads them)
        bool rows16 = false;
        for (int i = 0; i < a.ngroups; ++i) rows16 = rows16 || a.g[i].a16 || a.g[i].b16 || a.g[i].C16;
        int prec = g_precision;
        if (const char* e = getenv("MPNHIP_GEMM_PREC")) prec = atoi(e);
        // (short K with a ragged column count -- the edge encoder's 144 / 160-wide layers over 400,000 rows -- stays on the strip
        // kernel: 122 against 170 us at 400,000 x 160 x 144, tools/gemm_bf16_bench.py)
        const bool shape_ok = a.K >= 192 && (N % 128 == 0 || N >= 384);
        if (bl == B_KCONTIG && (rows16 || (prec == 1 && M >= 4096 && !a.small_tiles && shape_ok))) {
            int st = MPNHIP_OK;
            if (launch_gemm_bf16_tiled(a, s, &st)) {
                if (st == MPNHIP_OK) count_path(PC_GEMM_BF16);
                return st;
            }
        }
        if (rows16) { set_error("gemm: bf16 operand rows [%lld x %d x %d] are not a shape of the tiled bf16 kernel", (long long)M, N, a.K); return MPNHIP_ERR_UNSUPPORTED; }
    }
    if (M >= 8192 && !a.small_tiles) {
        // waves stacked along M (block 128 x 32 TN): pick the strip width that wastes the fewest tiles,
        // widest first (A is then re-read from L2 the fewest times)
        int best = 1, best_cost = 1 << 30;
        for (int tn = 8; tn >= 1; --tn) {
            int cost = ((nt + tn - 1) / tn) * tn;
            if (cost < best_cost) { best_cost = cost; best = tn; }
        }
        // ... but not so wide that the grid leaves CUs idle: with fewer than 256 blocks (20,000 rows x 256 columns at cfg-E: 157
        // blocks of 128 x 256) halve the strip until every CU has one (cfg-E forward 9.84 -> 9.65 ms)
        {
            const int64_t rb = (M + 127) / 128;
            while (best > 1 && rb * ((nt + best - 1) / best) < 256) best = (best + 1) / 2;
        }
        if (const char* e = getenv("MPNHIP_TN")) {  // tuning override
            int v = atoi(e);
            if (v >= 1 && v <= 8) best = v;
        }
        switch (best) {
            case 8: return launch_cfg<4, 1, 8>(a, bl, s);
            case 7: return launch_cfg<4, 1, 7>(a, bl, s);
            case 6: return launch_cfg<4, 1, 6>(a, bl, s);
            case 5: return launch_cfg<4, 1, 5>(a, bl, s);
            case 4: return launch_cfg<4, 1, 4>(a, bl, s);
            case 3: return launch_cfg<4, 1, 3>(a, bl, s);
            case 2: return launch_cfg<4, 1, 2>(a, bl, s);
            default: return launch_cfg<4, 1, 1>(a, bl, s);
        }
    }
    // few rows (node-level products): spread the columns over the waves so that the grid fills the chip
    if (const char* e = getenv("MPNHIP_SMALL_CFG")) {  // tuning override: 412 = <4,1,2>, 141 = <1,4,1>, ...
        switch (atoi(e)) {
            case 411: return launch_cfg<4, 1, 1>(a, bl, s);
            case 412: return launch_cfg<4, 1, 2>(a, bl, s);
            case 414: return launch_cfg<4, 1, 4>(a, bl, s);
            case 141: return launch_cfg<1, 4, 1>(a, bl, s);
            case 142: return launch_cfg<1, 4, 2>(a, bl, s);
            case 221: return launch_cfg<2, 2, 1>(a, bl, s);
            default: break;
        }
    }
    // measured on MI355X at M = 5,000 (tools/gemm_bench.py): 64 x 64 tiles beat every wider strip for
    // N = 128 ... 1088 and K = 128 ... 2048 (more, shorter blocks: the chip is latency- not MFMA-bound here)
    if (nt >= 2) return launch_cfg<2, 2, 1>(a, bl, s);
    return launch_cfg<4, 1, 1>(a, bl, s);
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

// ---- few rows, long K: split-K ------------------------------------------------------------------------------------------
// y = act(x W^T + b) with M of a few hundred rows and K in the thousands (the node encoder's first layer at the reference's graph
// sizes: [140 .. 500, 2048] x [2048, 128]): the tiled kernel above has ceil(M / 64) x ceil(N / 64) = 6 .. 16 blocks walking the
// whole K -- 54 us on an otherwise idle chip, 40 % of a KITTI-sized forward.  Here the K range is cut into `S` slices as well
// (grid.z): every block reduces one slice of one 64 x 64 tile with fp32 MFMAs into a partial tile [S][M][N], and a second
// small kernel sums the S partials in a fixed order and applies bias / ReLU.  Deterministic; needs M N S floats of scratch.
constexpr int SK_BK = 32, SK_P = 65;

__global__ __launch_bounds__(256) void k_gemm_splitk(const float* __restrict__ x, int64_t ldx, const float* __restrict__ w, int M, int N,
                                                     int K, int kc, float* __restrict__ part) {
    __shared__ float As[SK_BK * SK_P], Bs[SK_BK * SK_P];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, li = lane & 31, lh = lane >> 5;
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
    const int k0 = blockIdx.z * kc;
    const int k1 = k0 + kc < K ? k0 + kc : K;
    // loader: thread covers rows (tid / 8 + 32 j) of the tile at k offset (tid % 8) * 4
    const int lr = tid >> 3, lk = (tid & 7) * 4;
    const float* ap[2];
    const float* bp[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        int m = m0 + lr + 32 * j, n = n0 + lr + 32 * j;
        m = m < M ? m : M - 1;
        n = n < N ? n : N - 1;
        ap[j] = x + (int64_t)m * ldx;
        bp[j] = w + (int64_t)n * K;
    }
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int kt = k0; kt < k1; kt += SK_BK) {
        const int k = kt + lk;
        const bool ok = k < k1;                 // K % 4 == 0 and kc % 32 == 0: a float4 is entirely in or out of the slice
        float4 a[2], b[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            a[j] = ok ? *reinterpret_cast<const float4*>(ap[j] + k) : make_float4(0.f, 0.f, 0.f, 0.f);
            b[j] = ok ? *reinterpret_cast<const float4*>(bp[j] + k) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int c = lr + 32 * j;
            As[(lk + 0) * SK_P + c] = a[j].x; As[(lk + 1) * SK_P + c] = a[j].y; As[(lk + 2) * SK_P + c] = a[j].z; As[(lk + 3) * SK_P + c] = a[j].w;
            Bs[(lk + 0) * SK_P + c] = b[j].x; Bs[(lk + 1) * SK_P + c] = b[j].y; Bs[(lk + 2) * SK_P + c] = b[j].z; Bs[(lk + 3) * SK_P + c] = b[j].w;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < SK_BK; kk += 2)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(As[(kk + lh) * SK_P + wm * 32 + li], Bs[(kk + lh) * SK_P + wn * 32 + li], acc, 0, 0, 0);
    }
    float* out = part + (size_t)blockIdx.z * M * N;
    const int n = n0 + wn * 32 + li;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (m < M && n < N) out[(size_t)m * N + n] = acc[r];
    }
}

// (four partial sums over the chunks z = 0, 1, 2, 3 (mod 4), combined as (s0 + s1) + (s2 + s3): a fixed order, and four loads in
// flight per thread instead of a serial chain -- at 140 x 128 outputs the kernel is pure latency)
__device__ __forceinline__ float splitk_total(const float* __restrict__ part, int64_t mn, int S, int64_t i) {
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int z = 0;
    for (; z + 3 < S; z += 4) {
        s0 += part[(int64_t)z * mn + i];
        s1 += part[(int64_t)(z + 1) * mn + i];
        s2 += part[(int64_t)(z + 2) * mn + i];
        s3 += part[(int64_t)(z + 3) * mn + i];
    }
    if (z < S) s0 += part[(int64_t)z * mn + i];
    if (z + 1 < S) s1 += part[(int64_t)(z + 1) * mn + i];
    if (z + 2 < S) s2 += part[(int64_t)(z + 2) * mn + i];
    return (s0 + s1) + (s2 + s3);
}

__global__ void k_splitk_sum(const float* __restrict__ part, int64_t mn, int S, const float* __restrict__ bias, int N, int relu,
                             float* __restrict__ y, int64_t ldy) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= mn) return;
    float s = splitk_total(part, mn, S, i);
    const int n = (int)(i % N);
    if (bias) s += bias[n];
    if (relu) s = fmaxf(s, 0.f);
    y[(i / N) * ldy + n] = s;
}

// The same followed by the NEXT (last, narrow) Linear layer of the MLP: block = one row; its n1 <= 256 summed activations go to
// y1 (the backward's saved activation) and, through LDS, into y2[row] = act(W2 h + b2) -- the reference's node encoder
// (2048 -> 128 -> 32) on a few hundred nodes: two launches instead of three, and no 8 us GEMM launch for a 140 x 32 output.
__global__ __launch_bounds__(256) void k_splitk_sum_l2(const float* __restrict__ part, int64_t mn, int S, const float* __restrict__ b1,
                                                      int n1, int relu1, float* __restrict__ y1, int64_t ldy1,
                                                      const float* __restrict__ W2, const float* __restrict__ b2, int n2, int relu2,
                                                      float* __restrict__ y2, int64_t ldy2) {
    __shared__ __attribute__((aligned(16))) float h[256];
    const int row = blockIdx.x, t = threadIdx.x;
    if (t < n1) {
        float s = splitk_total(part, mn, S, (int64_t)row * n1 + t);
        if (b1) s += b1[t];
        if (relu1) s = fmaxf(s, 0.f);
        h[t] = s;
        y1[(int64_t)row * ldy1 + t] = s;
    }
    __syncthreads();
    // four lanes per output (k = q, q + 4, ... in float4 steps of 16), reduced by two shuffles: fixed order
    const int o = t >> 2, q = t & 3;
    float acc = 0.f;
    if (o < n2) {
        const float* w = W2 + (int64_t)o * n1;
        for (int k = 4 * q; k + 3 < n1; k += 16) {
            const float4 wv = *reinterpret_cast<const float4*>(w + k);
            const float4 hv = *reinterpret_cast<const float4*>(h + k);
            acc = fmaf(wv.x, hv.x, acc); acc = fmaf(wv.y, hv.y, acc); acc = fmaf(wv.z, hv.z, acc); acc = fmaf(wv.w, hv.w, acc);
        }
    }
    acc += __shfl_xor(acc, 1, 64);
    acc += __shfl_xor(acc, 2, 64);
    if (o < n2 && q == 0) {
        if (b2) acc += b2[o];
        if (relu2) acc = fmaxf(acc, 0.f);
        y2[(int64_t)row * ldy2 + o] = acc;
    }
}

// true (and launched) when the shape calls for it and the scratch suffices; false: the caller takes the tiled kernel
// few output tiles, long K: up to a few hundred 64 x 64 tiles (the node encoder's first layer: 2048 -> 128 over the reference's few
// hundred nodes, and over cfg-B's 5,000 -- 158 tiles, each walking the whole K: 104 us; four K slices: ~35 us)
constexpr int64_t SPLITK_MAX_ROWS = 8192, SPLITK_MAX_TILES = 320;
static inline int64_t splitk_blocks(int64_t tiles) { return tiles < 96 ? 384 : 768; }

bool linear_splitk(const float* x, int64_t ldx, const float* w, const float* b, float* y, int64_t ldy, int64_t m, int n, int k, int relu,
                   float* scratch, size_t scratch_floats, hipStream_t stream, int* status, SplitkNext* next) {
    *status = MPNHIP_OK;
    // (fp32 MFMAs: exact fp32 products -- also what the split precision may use; the bf16-operand mode must round its operands)
    if (g_precision == 1 || getenv("MPNHIP_NO_SPLITK")) return false;
    const int64_t tiles = ((m + 63) / 64) * ((n + 63) / 64);
    if (!scratch || m <= 0 || m > SPLITK_MAX_ROWS || k < 512 || k % 4 != 0 || ldx % 4 != 0 || tiles >= SPLITK_MAX_TILES || (((uintptr_t)x | (uintptr_t)w) & 15)) return false;
    int S = (int)(splitk_blocks(tiles) / tiles);
    if (S > k / 64) S = k / 64;
    if (S < 2) return false;
    int kc = ((k + S - 1) / S + 31) / 32 * 32;
    S = (k + kc - 1) / kc;
    if ((size_t)S * m * n > scratch_floats) return false;
    count_path(PC_GEMM_SPLITK);
    hipLaunchKernelGGL(k_gemm_splitk, dim3((n + 63) / 64, (unsigned)((m + 63) / 64), S), dim3(256), 0, stream, x, ldx, w, (int)m, n, k, kc, scratch);
    const int64_t mn = m * n;
    if (next && next->w && n % 4 == 0 && n <= 256 && ldy == n && next->n >= 1 && next->n <= 64 && ((uintptr_t)next->w & 15) == 0) {
        // ... and the following (last, narrow) layer in the same launch
        hipLaunchKernelGGL(k_splitk_sum_l2, dim3((unsigned)m), dim3(256), 0, stream, scratch, mn, S, b, n, relu, y, ldy, next->w, next->b,
                           next->n, next->relu, next->y, next->ldy);
        next->done = true;
    } else {
        hipLaunchKernelGGL(k_splitk_sum, dim3((unsigned)((mn + 255) / 256)), dim3(256), 0, stream, scratch, mn, S, b, n, relu, y, ldy);
    }
    if (hipGetLastError() != hipSuccess) { set_error("split-K linear: launch failed"); *status = MPNHIP_ERR_HIP; }
    return true;
}

size_t linear_splitk_scratch_floats(int64_t m, int n, int k) {
    const int64_t tiles = ((m + 63) / 64) * ((n + 63) / 64);
    if (m <= 0 || m > SPLITK_MAX_ROWS || k < 512 || tiles >= SPLITK_MAX_TILES) return 0;
    int S = (int)(splitk_blocks(tiles) / tiles);
    if (S > k / 64) S = k / 64;
    return S >= 2 ? (size_t)(S + 1) * m * n : 0;
}

}  // namespace mpnhip
