// bf16-operand GEMM for the node-side / encoder / tail products of MPNHIP_PREC_BF16 (BASELINE.json configs[4]: "bf16 MLP GEMMs on
// MFMA") at large row counts -- the reference's nn.Linear products of models/mpn.py:69,87,93,97-99 (per-node projections, node
// update) and, under autograd, dX = dP Wx, dAGG = dZn Wu and the hoisted e0 / x0 shares (mlp.py:27-28).
//
//   C[m, n] = mask( act( sum_k bf16(A[m, k]) bf16(B[n, k]) + bias[n] + G1[i1(m)][n] + G2[i2(m)][n] ) (+ C) ),  fp32 accumulate
//
// Same interface as gemm_kernel (GemmArgs, common.h), same arithmetic as its bf16 instantiation (operands rounded to bf16 RNE,
// v_mfma_f32_32x32x16_bf16, fp32 accumulation, fp32 epilogue) -- another summation order only.  What is different is everything
// around the MFMAs, because at these shapes (20,000 x 2,176 x 512: 174 MB of fp32 output for 44 GFLOP) the product is bound by
// its operand and result STREAMS, not by the matrix pipe:
//   * block tile 128 x 128, 8 waves (2 x 4, wave tile 64 x 32), K step 64: a block stages 2 x 16 KB of bf16 per K step instead of
//     the old kernel's 128 x (32 TN) strip at K step 32 with two barriers per 8 MFMAs;
//   * operands may be bf16 ROWS in memory (GemmGroup::a16 / b16: the packed weight images, the bf16 mirrors of x0 / x kept by the
//     forward) -- 16-byte loads of 8 elements straight into the LDS image, no conversion, half the L2 -> LDS bytes -- or fp32 rows
//     converted while they are staged (v_cvt_pk_bf16_f32);
//   * two LDS images, ONE barrier per K step: the global loads of step k+1 are issued before the MFMAs of step k and written to
//     the other image after them (register staging, loads in flight across the whole MFMA phase);
//   * LDS rows are 64 elements + 16 bytes of padding (pitch 144 B): every ds_read_b128 operand fetch is conflict-free;
//   * 1-D grid with an XCD-aware block -> tile map: the 8 XCDs each walk a contiguous run of (row panel, column block) tiles,
//     column block fastest, so a row panel of A is pulled into ONE L2 and the weight image stays resident in all of them;
//   * epilogue through a per-wave LDS patch as whole 128-byte row pieces (8 lines per store instruction); the plain form (bias /
//     ReLU / store, optional bf16 mirror of the result) is its own instantiation without the gather / mask / accumulate operands;
//     two blocks per CU, so one block's stores overlap the other's K loop.
#include <cstdlib>

#include "common.h"
#include "edge_chain.h"

namespace mpnhip {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int GB_THREADS = 512;
constexpr int GB_BK = 64;
constexpr int GB_PITCH = 72;            // bf16 elements per LDS row: 64 + 8 (16 bytes of padding)
constexpr int GB_PATCH = 32 * 36;       // floats of a wave's epilogue patch [32][36]

__device__ __forceinline__ float4 gb_ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 gb_keep4(bool ok, float4 v) {
    v.x = ok ? v.x : 0.f; v.y = ok ? v.y : 0.f; v.z = ok ? v.z : 0.f; v.w = ok ? v.w : 0.f;
    return v;
}
__device__ __forceinline__ uint4 gb_keep16(bool ok, uint4 v) {
    v.x = ok ? v.x : 0u; v.y = ok ? v.y : 0u; v.z = ok ? v.z : 0u; v.w = ok ? v.w : 0u;
    return v;
}

// One operand tile [ROWS][64] of a K step: per thread NLD loads (fp32: float4 = 4 elements, rows (t / 16) + 32 j; bf16: uint4 = 8
// elements, rows (t / 8) + 64 j), kept in registers across the MFMA phase, then written to the LDS image.
template <int ROWS, bool SRC16>
struct OperandStage {
    static constexpr int NLD = SRC16 ? ROWS / 64 : ROWS / 32;
    uint4 r[NLD];
    int off1[NLD], off2[NLD];   // element offsets of the thread's rows in the two K segments
    int kq;                     // first k of the thread's piece inside a K step
    int row0;                   // first tile row of the thread

    __device__ __forceinline__ void init(int tid) {
        if (SRC16) { kq = (tid & 7) * 8; row0 = tid >> 3; }
        else { kq = (tid & 15) * 4; row0 = tid >> 4; }
    }
    __device__ __forceinline__ int tile_row(int j) const { return row0 + (SRC16 ? 64 : 32) * j; }

    // base / base2: the operand's two K segments (base2 already shifted by -ksplit elements); K tail -> zeros (selected in write)
    __device__ __forceinline__ void load(const void* base, const void* base2, int kt, int K, int ksplit) {
        const int k = kt * GB_BK + kq;
        const int kc = k < K ? k : 0;
        const bool seg2 = kc >= ksplit;
#pragma unroll
        for (int j = 0; j < NLD; ++j) {
            const int o = (seg2 ? off2[j] : off1[j]) + kc;
            if (SRC16) r[j] = *reinterpret_cast<const uint4*>(static_cast<const unsigned short*>(seg2 ? base2 : base) + o);
            else {
                const float4 v = gb_ld4(static_cast<const float*>(seg2 ? base2 : base) + o);
                r[j] = make_uint4(__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w));
            }
        }
    }
    __device__ __forceinline__ void write(__bf16* img, int kt, int K) const {
        const bool ok = kt * GB_BK + kq < K;
#pragma unroll
        for (int j = 0; j < NLD; ++j) {
            __bf16* d = img + tile_row(j) * GB_PITCH + kq;
            if (SRC16) *reinterpret_cast<uint4*>(d) = gb_keep16(ok, r[j]);
            else {
                const float4 v = gb_keep4(ok, make_float4(__uint_as_float(r[j].x), __uint_as_float(r[j].y), __uint_as_float(r[j].z),
                                                          __uint_as_float(r[j].w)));
                *reinterpret_cast<bf16x4*>(d) = bf16x4{(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};
            }
        }
    }
};

// XCD-aware linear block id -> tile index: the blocks that share an XCD (equal id % 8 under round-robin placement: speed only)
// walk a contiguous run of tiles (bijective for any tile count; cdna_hip_programming.md 5.5 T1)
__device__ __forceinline__ int gb_tile_of_block(int id, int total) {
    const int q = total >> 3, r = total & 7, x = id & 7, s = id >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + s;
}

template <int WM, int WN, int TM, int TN, bool A16, bool B16, bool FULL>
__global__ __launch_bounds__(GB_THREADS, 4) void gemm_bf16_kernel(GemmArgs args, int nbx, int nby) {
    static_assert(WM * WN == 8, "8 waves");
    constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN;
    static_assert(BM % 64 == 0 && BN % 64 == 0, "tile rows per loader pass");
    constexpr int IMG = (BM + BN) * GB_PITCH;                 // bf16 elements of one stage image (A rows, then B rows)
    constexpr int SMEM_BYTES = 2 * IMG * 2 > 8 * GB_PATCH * 4 ? 2 * IMG * 2 : 8 * GB_PATCH * 4;
    __shared__ __attribute__((aligned(16))) unsigned char smem_raw[SMEM_BYTES];
    __bf16* const img0 = reinterpret_cast<__bf16*>(smem_raw);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 31, lh = lane >> 5;

    const int t = gb_tile_of_block(blockIdx.x, nbx * nby);
    const int by = t / nbx, bx = t - by * nbx;

    // ---- which group / which rows (device-side row ranges: gemm.hip) -----------------------------------
    int grp = 0;
    int row0, row_end;
    {
        const GemmGroup& g0 = args.g[0];
        const int b0 = g0.row_begin ? *g0.row_begin : 0;
        const int e0 = g0.row_end ? *g0.row_end : (int)g0.m_static;
        int nb0 = (e0 - b0 + BM - 1) / BM;
        if (nb0 < 0) nb0 = 0;
        if (by < nb0) {
            row0 = b0 + by * BM;
            row_end = e0;
        } else {
            if (args.ngroups < 2) return;
            grp = 1;
            const GemmGroup& g1 = args.g[1];
            const int b1 = g1.row_begin ? *g1.row_begin : 0;
            const int e1 = g1.row_end ? *g1.row_end : (int)g1.m_static;
            row0 = b1 + (by - nb0) * BM;
            row_end = e1;
            if (row0 >= row_end) return;
        }
    }
    const GemmGroup& G = args.g[grp];
    const int col0 = bx * BN;
    const int N = args.N, K = args.K, ksplit = args.ksplit;

    OperandStage<BM, A16> sa;
    OperandStage<BN, B16> sb;
    sa.init(tid);
    sb.init(tid);
    const void* const Abase = G.A;
    const void* const A2base = G.A2 ? (A16 ? static_cast<const void*>(reinterpret_cast<const unsigned short*>(G.A2) - ksplit)
                                           : static_cast<const void*>(G.A2 - ksplit))
                                    : G.A;
    const void* const Bbase = G.B;
#pragma unroll
    for (int j = 0; j < sa.NLD; ++j) {
        int r = row0 + sa.tile_row(j);
        r = r < row_end ? r : row_end - 1;    // rows past the end are computed and never stored
        const int ri = G.a_idx ? G.a_idx[r] : r;
        sa.off1[j] = ri * (int)G.lda;
        sa.off2[j] = G.A2 ? ri * (int)G.lda2 : sa.off1[j];
    }
#pragma unroll
    for (int j = 0; j < sb.NLD; ++j) {
        int n = col0 + sb.tile_row(j);
        n = n < N ? n : N - 1;
        sb.off1[j] = n * (int)G.ldb;
        sb.off2[j] = sb.off1[j];
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = (K + GB_BK - 1) / GB_BK;
    sa.load(Abase, A2base, 0, K, ksplit);
    sb.load(Bbase, Bbase, 0, K, K);
    sa.write(img0, 0, K);
    sb.write(img0 + BM * GB_PITCH, 0, K);
    for (int kt = 0; kt < nk; ++kt) {
        __bf16* const cur = img0 + (kt & 1) * IMG;
        __bf16* const nxt = img0 + ((kt + 1) & 1) * IMG;
        // image kt complete; every wave has finished its reads of the other image (iteration kt - 1)
        __syncthreads();
        const bool more = kt + 1 < nk;
        if (more) {
            sa.load(Abase, A2base, kt + 1, K, ksplit);
            sb.load(Bbase, Bbase, kt + 1, K, K);
        }
        const __bf16* ap = cur + (wm * 32 * TM + li) * GB_PITCH + lh * 8;
        const __bf16* bp = cur + (BM + wn * 32 * TN + li) * GB_PITCH + lh * 8;
#pragma unroll
        for (int kb = 0; kb < GB_BK / 16; ++kb) {
            bf16x8 av[TM], bv[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) av[i] = *reinterpret_cast<const bf16x8*>(ap + 32 * i * GB_PITCH + kb * 16);
#pragma unroll
            for (int j = 0; j < TN; ++j) bv[j] = *reinterpret_cast<const bf16x8*>(bp + 32 * j * GB_PITCH + kb * 16);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[i], bv[j], acc[i][j], 0, 0, 0);
        }
        if (more) {
            sa.write(nxt, kt + 1, K);
            sb.write(nxt + BM * GB_PITCH, kt + 1, K);
        }
    }
    __syncthreads();   // the patches below overlay the stage images

    // ---- epilogue ------------------------------------------------------------------------------------
    // D[i][j] of a 32 x 32 tile: j = lane & 31, i = (r & 3) + 8 (r >> 2) + 4 (lane >> 5).  Patch [32 rows][36]: written column-wise
    // (conflict-free), read back as float4 row pieces: lane -> row (lane / 8 + 8 p), columns 4 (lane % 8) .. +3.
    float* const patch = reinterpret_cast<float*>(smem_raw) + wave * GB_PATCH;
    const int er = lane >> 3, ec = (lane & 7) * 4;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int ti = 0; ti < TM; ++ti) {
        int m_row[4], c_off[4], g1_off[4], g2_off[4];
        bool m_ok[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int m = row0 + (wm * TM + ti) * 32 + er + 8 * p;
            m_ok[p] = m < row_end;
            const int mc = m_ok[p] ? m : row_end - 1;
            m_row[p] = mc;
            c_off[p] = (G.c_idx ? G.c_idx[mc] : mc) * (int)G.ldc;
            if (FULL) {
                g1_off[p] = (G.g1_idx ? G.g1_idx[mc] : mc) * (int)G.ldg1;
                g2_off[p] = (G.g2_idx ? G.g2_idx[mc] : mc) * (int)G.ldg2;
            }
        }
#pragma unroll
        for (int tj = 0; tj < TN; ++tj) {
            const int ncol0 = col0 + (wn * TN + tj) * 32;
            if (ncol0 >= N) break;   // wave-uniform
#pragma unroll
            for (int r = 0; r < 16; ++r) patch[((r & 3) + 8 * (r >> 2) + 4 * lh) * 36 + li] = acc[ti][tj][r];
            // (one wave wrote and reads: a wave's LDS operations complete in order)
            const int n = ncol0 + ec;
            const bool n_ok = n < N;           // N % 4 == 0 on this path
            const int nc = n_ok ? n : N - 4;
            const float4 bias = G.bias ? gb_ld4(G.bias + nc) : zero4;
            if (!FULL) {
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    float4 o = *reinterpret_cast<const float4*>(&patch[(er + 8 * p) * 36 + ec]);
                    o.x += bias.x; o.y += bias.y; o.z += bias.z; o.w += bias.w;
                    if (args.relu) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
                    if (m_ok[p] && n_ok) {
                        *reinterpret_cast<float4*>(G.C + c_off[p] + nc) = o;
                        if (G.C16) {
                            const bf16x4 ob = {(__bf16)o.x, (__bf16)o.y, (__bf16)o.z, (__bf16)o.w};
                            *reinterpret_cast<bf16x4*>(reinterpret_cast<__bf16*>(G.C16) + (int64_t)m_row[p] * G.ldc16 + nc) = ob;
                        }
                    }
                }
            } else {
#pragma unroll
                for (int ph = 0; ph < 4; ph += 2) {
                    float4 v[2], g1[2], g2[2], mk[2], old[2];
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const int p = ph + q;
                        v[q] = *reinterpret_cast<const float4*>(&patch[(er + 8 * p) * 36 + ec]);
                        g1[q] = G.G1 ? gb_ld4(G.G1 + g1_off[p] + nc) : zero4;
                        g2[q] = G.G2 ? gb_ld4(G.G2 + g2_off[p] + nc) : zero4;
                        mk[q] = G.mask ? gb_ld4(G.mask + m_row[p] * (int)G.ldmask + nc) : make_float4(1.f, 1.f, 1.f, 1.f);
                        old[q] = args.accumulate ? gb_ld4(G.C + c_off[p] + nc) : zero4;
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
                        if (m_ok[p] && n_ok) {
                            *reinterpret_cast<float4*>(G.C + c_off[p] + nc) = o;
                            if (G.C16) {
                                const bf16x4 ob = {(__bf16)o.x, (__bf16)o.y, (__bf16)o.z, (__bf16)o.w};
                                *reinterpret_cast<bf16x4*>(reinterpret_cast<__bf16*>(G.C16) + (int64_t)m_row[p] * G.ldc16 + nc) = ob;
                            }
                        }
                    }
                }
            }
        }
    }
}

// ---- bf16 x bf16 rows, plain epilogue: persistent blocks over an LDS-DMA ring -------------------------------------------------
// The register-staged kernel above keeps one K step in flight per block and needs registers for it.  When BOTH operands are bf16
// rows in memory nothing has to pass through registers:
//   * one 768-thread block per CU, persistent (8 multiplying waves + 4 loader waves): the XCD's 32 blocks walk the XCD's contiguous
//     run of tiles;
//   * block tile 128 x 128 (waves 2 x 4, wave tile 64 x 32); a 256 x 128 form (waves 4 x 2, wave tile 64 x 64: 16 MFMAs per wave between
//     two barriers) is instantiated behind MPNHIP_GEMM_RING_TILE=256 -- measured no faster;
//   * a ring of NST stages (A rows [BM][64] | B rows [BN][64], 128-byte rows, 16-byte chunks XOR-swizzled by (row >> 1) & 7 through
//     the per-lane SOURCE address: the LDS-DMA destination is lane-linear; conflict-free for the ds_read_b128 lane groups), filled
//     by global_load_lds_dwordx4 in 1 KiB pieces, NST - 1 stages ahead -- across tile boundaries: the next tile's first stages
//     are in flight while this tile's epilogue runs;
//   * one counted s_waitcnt vmcnt + s_barrier per K step (the pieces of the following stages stay in flight; a tile's first steps
//     over-wait for the previous epilogue's stores, which is safe); operand fetches by inline-assembly ds_read_b128 (a
//     compiler-visible read of a DMA target is preceded by vmcnt(0)), consumed behind counted lgkmcnt waits;
//   * biases sit in LDS for the whole launch (a global load's first use would drain the ring); results leave straight from the
//     accumulator layout (two whole 128-byte lines per store instruction), no LDS patch.
constexpr int RG_BIAS_MAX = 3072;

struct RingArgs {
    const unsigned short* A; const unsigned short* A2; const unsigned short* B;   // A2 already shifted by -ksplit elements
    const float* bias; float* C;
    int lda, lda2, ldb, ldc;
    int M, N, K, ksplit, relu, nbx, nby;
    int debug;   // timing ablations (MPNHIP_GEMM_RING_DEBUG; results wrong): 1 no operand fetch / MFMA, 2 no DMA, 4 no stores, 8 linear chunks
};

template <int OFF>
__device__ __forceinline__ void rg_read(unsigned addr, bf16x8& d) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF) : "memory");
}
template <int N>
__device__ __forceinline__ void rg_wait(bf16x8& a, bf16x8& b, bf16x8& c) {
    asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(a), "+v"(b), "+v"(c) : "n"(N));
}
template <int N>
__device__ __forceinline__ void rg_wait(bf16x8& a, bf16x8& b, bf16x8& c, bf16x8& d) {
    asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "n"(N));
}

// waves 0-7 multiply and store, waves 8 .. 8 + NLW - 1 load; BPC blocks per CU (launch bounds: 64 (8 + NLW) BPC / 256 waves per SIMD)
template <int WM, int WN, int TM, int TN, int NST, int NLW, int BPC>
__global__ __launch_bounds__(64 * (8 + NLW), (8 + NLW) * BPC / 4) void gemm_bf16_ring_kernel(RingArgs a) {
    static_assert(WM * WN == 8 && TM == 2 && (TN == 1 || TN == 2), "wave arrangement");
    constexpr int RG_THREADS = 64 * (8 + NLW);
    constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN;
    constexpr int PA = BM / 8 / NLW, PB = BN / 8 / NLW;       // 1 KiB pieces (8 rows) per LOADER wave and stage
    constexpr int NP = PA + PB;
    constexpr int STAGE = (BM + BN) * 128;
    constexpr int BIAS_OFF = NST * STAGE;
    constexpr int LDS = BIAS_OFF + RG_BIAS_MAX * 4;
    static_assert(LDS * BPC <= 163840, "LDS");
    __shared__ __attribute__((aligned(16))) char smem[LDS];
    float* const sbias = reinterpret_cast<float*>(smem + BIAS_OFF);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int N = a.N, M = a.M, nbx = a.nbx;

    // this block's tiles: the blocks with equal id % 8 (one XCD under round-robin placement: speed only) share a contiguous run
    const int total = nbx * a.nby, g8 = (int)gridDim.x >> 3;
    const int xl = blockIdx.x & 7, jb = blockIdx.x >> 3;
    const int tq = total >> 3, tr = total & 7;
    const int lo = xl < tr ? xl * (tq + 1) : tr * (tq + 1) + (xl - tr) * tq;
    const int cnt = tq + (xl < tr ? 1 : 0);
    const int ntl = jb < cnt ? (cnt - jb + g8 - 1) / g8 : 0;
    if (ntl == 0) return;
    const int nk = a.K >> 6;
    const int nstages = ntl * nk;

    for (int i = tid; i < N; i += RG_THREADS) sbias[i] = a.bias ? a.bias[i] : 0.f;
    __syncthreads();           // biases in place (no DMA in flight yet)

    if (wave >= 8) {
        // ---- loader waves: stage (is_i, is_kt) of this block's sequence -> ring slot is_slot ----------------
        // Roles are split because vmcnt counts loads and stores together, in order: a wave that both stores a tile's results and
        // waits for the next stage's pieces would wait for its stores to COMPLETE at every tile boundary (measured on the
        // unsplit form: store time and multiply time added up instead of overlapping).  These waves only load and wait for
        // their own pieces; the multiplying waves never execute a vmcnt wait.
        const int lw = wave - 8;
        int is_i = 0, is_kt = 0, is_slot = 0;
        int oA1[PA], oA2[PA], oB[PB];       // element offsets of this lane's rows
        auto issue = [&]() {
            if (is_kt == 0) {
                const int t = lo + jb + is_i * g8;
                const int by = t / nbx, bx = t - by * nbx;
#pragma unroll
                for (int j = 0; j < PA; ++j) {
                    const int rt = 8 * (PA * lw + j) + (lane >> 3);           // tile row of this lane's chunk
                    const int c = ((lane & 7) ^ ((rt >> 1) & 7)) * 8;         // the k chunk that belongs at this LDS position
                    int m = by * BM + rt;  m = m < M ? m : M - 1;
                    oA1[j] = m * a.lda + c;
                    oA2[j] = m * a.lda2 + c;
                }
#pragma unroll
                for (int j = 0; j < PB; ++j) {
                    const int rt = 8 * (PB * lw + j) + (lane >> 3);
                    const int c = ((lane & 7) ^ ((rt >> 1) & 7)) * 8;
                    int n = bx * BN + rt;  n = n < N ? n : N - 1;
                    oB[j] = n * a.ldb + c;
                }
            }
            const int k0 = is_kt * 64;
            const bool seg2 = k0 >= a.ksplit;
            const unsigned short* ab = seg2 ? a.A2 : a.A;
            char* const dst = smem + is_slot * STAGE;
            if (!(a.debug & 2)) {
#pragma unroll
                for (int j = 0; j < PA; ++j)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ab + (seg2 ? oA2[j] : oA1[j]) + k0),
                                                     (__attribute__((address_space(3))) void*)(dst + (PA * lw + j) * 1024), 16, 0, 0);
#pragma unroll
                for (int j = 0; j < PB; ++j)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a.B + oB[j] + k0),
                                                     (__attribute__((address_space(3))) void*)(dst + BM * 128 + (PB * lw + j) * 1024), 16, 0, 0);
            }
            is_slot = is_slot == NST - 1 ? 0 : is_slot + 1;
            if (++is_kt == nk) { is_kt = 0; ++is_i; }
        };
#pragma unroll
        for (int q = 0; q < NST - 1; ++q)
            if (q < nstages) issue();
        for (int g = 0; g < nstages; ++g) {
            // stage g has landed (the NST - 2 following stages' pieces may stay in flight) -> barrier: the multiplying waves may
            // read it, and they are done with stage g - 1, whose slot stage g + NST - 1 refills
            if (g + NST - 2 < nstages) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NP * (NST - 2)) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (g + NST - 1 < nstages) issue();
        }
        return;
    }

    // ---- multiplying waves ---------------------------------------------------------------------------
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 31, lh = lane >> 5;
    // operand fetch addresses: row (wm 32 TM + 32 i + li) of A / (wn 32 TN + 32 j + li) of B, logical chunk 2 kb + lh at position
    // ^ ((li >> 1) & 7)
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    unsigned fa[4], fb[4];
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
        const unsigned pos = (unsigned)((2 * kb + lh) ^ ((li >> 1) & 7)) * 16u;
        fa[kb] = lds0 + (unsigned)(wm * 32 * TM + li) * 128u + pos;
        fb[kb] = lds0 + (unsigned)(BM + wn * 32 * TN + li) * 128u + pos;
    }
    unsigned slot_off = 0;
    for (int i = 0; i < ntl; ++i) {
        const int t = lo + jb + i * g8;
        const int by = t / nbx, bx = t - by * nbx;
        f32x16 acc[TM][TN];
#pragma unroll
        for (int q = 0; q < TM; ++q)
#pragma unroll
            for (int u = 0; u < TN; ++u)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[q][u][r] = 0.f;
        for (int kt = 0; kt < nk; ++kt) {
            __builtin_amdgcn_s_barrier();
            if (a.debug & 1) { slot_off = slot_off == (NST - 1) * STAGE ? 0u : slot_off + STAGE; continue; }
            bf16x8 av[4][TM], bv[4][TN];
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) {
                rg_read<0>(fa[kb] + slot_off, av[kb][0]);
                rg_read<4096>(fa[kb] + slot_off, av[kb][1]);
                rg_read<0>(fb[kb] + slot_off, bv[kb][0]);
                if constexpr (TN == 2) rg_read<4096>(fb[kb] + slot_off, bv[kb][1]);
            }
#define RG_STEP(kb, left)                                                                                           \
            if constexpr (TN == 2) rg_wait<(left)>(av[kb][0], av[kb][1], bv[kb][0], bv[kb][1]);                        \
            else rg_wait<(left)>(av[kb][0], av[kb][1], bv[kb][0]);                                                     \
            _Pragma("unroll") for (int q = 0; q < TM; ++q)                                                             \
                _Pragma("unroll") for (int u = 0; u < TN; ++u)                                                         \
                    acc[q][u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[kb][q], bv[kb][u], acc[q][u], 0, 0, 0);
            RG_STEP(0, 3 * (TM + TN))
            RG_STEP(1, 2 * (TM + TN))
            RG_STEP(2, 1 * (TM + TN))
            RG_STEP(3, 0)
#undef RG_STEP
            slot_off = slot_off == (NST - 1) * STAGE ? 0u : slot_off + STAGE;
        }
        // ---- epilogue of tile (by, bx): the loader waves have the next tile's first stages in flight meanwhile ----
        // Straight from the accumulator layout: register r of a 32 x 32 tile holds rows (r & 3) + 8 (r >> 2) + 4 lh at column li, so
        // one store instruction writes two whole 128-byte lines.  No LDS patch here (a compiler-visible LDS access next to an
        // LDS-DMA target draws s_waitcnt vmcnt(0)); the biases come out of LDS by inline assembly for the same reason.
        if (a.debug & 4) { asm volatile("" :: "v"(acc[0][0][0]), "v"(acc[1][0][5])); continue; }
        const bool interior = by * BM + BM <= M && bx * BN + BN <= N;     // block-uniform: no per-store predicates
#pragma unroll
        for (int u = 0; u < TN; ++u) {
            const int n = bx * BN + (wn * TN + u) * 32 + li;
            const bool n_ok = n < N;
            float bias;
            asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(bias) : "v"(lds0 + (unsigned)BIAS_OFF + 4u * (unsigned)(n_ok ? n : 0)) : "memory");
#pragma unroll
            for (int q = 0; q < TM; ++q) {
                const int m0 = by * BM + (wm * TM + q) * 32 + 4 * lh;
                float* const cp = a.C + (int64_t)m0 * a.ldc + (n_ok ? n : 0);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int dr = (r & 3) + 8 * (r >> 2);
                    float o = acc[q][u][r] + bias;
                    if (a.relu) o = fmaxf(o, 0.f);
                    if (interior) cp[(int64_t)dr * a.ldc] = o;
                    else if (m0 + dr < M && n_ok) cp[(int64_t)dr * a.ldc] = o;
                }
            }
        }
    }
}

int launch_ring(const GemmArgs& a, hipStream_t s) {
    const GemmGroup& g = a.g[0];
    RingArgs r = {};
    r.A = reinterpret_cast<const unsigned short*>(g.A);
    r.A2 = g.A2 ? reinterpret_cast<const unsigned short*>(g.A2) - a.ksplit : r.A;
    r.B = reinterpret_cast<const unsigned short*>(g.B);
    r.bias = g.bias; r.C = g.C;
    r.lda = (int)g.lda; r.lda2 = g.A2 ? (int)g.lda2 : (int)g.lda; r.ldb = (int)g.ldb; r.ldc = (int)g.ldc;
    r.M = (int)a.m_upper; r.N = a.N; r.K = a.K; r.ksplit = g.A2 ? a.ksplit : a.K; r.relu = a.relu;
    // 128 x 128 tiles; MPNHIP_GEMM_RING_TILE=256: the 256 x 128 form (measured over four boxes at the projections' shape,
    // 20,000 x 2,176 x 512: 76-80 us against 72-85 -- no faster on average and less even; 40 against 46 us at 20,000 x 256 x 2,176)
    bool big = false;
    if (const char* e = getenv("MPNHIP_GEMM_RING_TILE")) big = atoi(e) == 256;
    const int bm = big ? 256 : 128;
    r.nbx = (a.N + 127) / 128; r.nby = (int)((a.m_upper + bm - 1) / bm);
    const int64_t total = (int64_t)r.nbx * r.nby;
    int grid = total >= 256 ? 256 : (int)((total + 7) / 8 * 8);
    if (const char* e = getenv("MPNHIP_GEMM_RING_DEBUG")) r.debug = atoi(e);
    if (const char* e = getenv("MPNHIP_GEMM_RING_BLOCKS")) { const int v = atoi(e); if (v >= 8 && v <= 1024) grid = v / 8 * 8; }
    if (big) MPN_LAUNCH_PROFILED((gemm_bf16_ring_kernel<4, 2, 2, 2, 3, 4, 1>), dim3((unsigned)grid), dim3(768), s, r);
    else MPN_LAUNCH_PROFILED((gemm_bf16_ring_kernel<2, 4, 2, 1, 4, 4, 1>), dim3((unsigned)grid), dim3(768), s, r);
    MPN_LAUNCH_CHECK();
    return MPNHIP_OK;
}

// the ring kernel's shapes: both operands bf16 rows, one group over static rows, plain epilogue, whole 64-deep K steps
bool ring_eligible(const GemmArgs& a, bool a16, bool b16, bool full) {
    if (!a16 || !b16 || full || a.ngroups != 1 || getenv("MPNHIP_NO_GEMM_BF16_RING")) return false;
    const GemmGroup& g = a.g[0];
    if (g.row_begin || g.row_end || g.a_idx || g.c_idx || g.C16 || g.m_static != a.m_upper) return false;
    if (a.K % 64 != 0 || a.ksplit % 64 != 0 || a.N > RG_BIAS_MAX || a.N % 4 != 0) return false;
    if (a.m_upper * (g.ldc > g.lda ? g.ldc : g.lda) >= ((int64_t)1 << 31)) return false;
    return true;
}

template <bool A16, bool B16, bool FULL>
int launch_variant(const GemmArgs& a, hipStream_t s) {
    constexpr int BM = 128, BN = 128;
    const int nbx = (a.N + BN - 1) / BN;
    const int64_t nby = (a.m_upper + BM - 1) / BM + (a.ngroups > 1 ? 1 : 0);
    if (nbx * nby >= ((int64_t)1 << 31)) { set_error("gemm (bf16): too many tiles"); return MPNHIP_ERR_ARG; }
    MPN_LAUNCH_PROFILED((gemm_bf16_kernel<2, 4, 2, 1, A16, B16, FULL>), dim3((unsigned)(nbx * nby)), dim3(GB_THREADS), s, a, nbx, (int)nby);
    MPN_LAUNCH_CHECK();
    return MPNHIP_OK;
}

bool al16p(const void* p) { return (((uintptr_t)p) & 15) == 0; }

}  // namespace

// true: taken (status = the launch's result); false: not a shape / alignment of this kernel -- the caller's older kernel runs
bool launch_gemm_bf16_tiled(const GemmArgs& a, hipStream_t s, int* status) {
    *status = MPNHIP_OK;
    if (getenv("MPNHIP_NO_GEMM_BF16_TILED")) return false;
    if (!a.epi_vec || a.K <= 0) return false;
    {   // (narrow outputs stay with the older kernel's 32-column strips -- unless an operand is bf16 rows, which only this kernel reads)
        bool rows16 = false;
        for (int i = 0; i < a.ngroups; ++i) rows16 = rows16 || a.g[i].a16 || a.g[i].b16 || a.g[i].C16;
        if (a.N < 32 && !rows16) return false;
    }
    bool a16 = false, b16 = false, full = a.accumulate != 0;
    for (int i = 0; i < a.ngroups; ++i) {
        const GemmGroup& g = a.g[i];
        if (i == 0) { a16 = g.a16 != 0; b16 = g.b16 != 0; }
        else if ((g.a16 != 0) != a16 || (g.b16 != 0) != b16) return false;
        full = full || g.G1 || g.G2 || g.mask;
        if (g.C16 && (g.c_idx || (g.ldc16 % 4) != 0 || (((uintptr_t)g.C16) & 7) != 0)) return false;
    }
    // bf16 rows: 16-byte pieces of 8 elements
    if (a16 && (a.K % 8 != 0 || a.ksplit % 8 != 0)) return false;
    if (b16 && a.K % 8 != 0) return false;
    for (int i = 0; i < a.ngroups; ++i) {
        const GemmGroup& g = a.g[i];
        if (a16 && (g.lda % 8 != 0 || !al16p(g.A) || (g.A2 && (g.lda2 % 8 != 0 || !al16p(g.A2))))) return false;
        if (b16 && (g.ldb % 8 != 0 || !al16p(g.B))) return false;
    }
    count_path(PC_GEMM_BF16_TILED);
    if (ring_eligible(a, a16, b16, full)) { count_path(PC_GEMM_BF16_RING); *status = launch_ring(a, s); return true; }
    if (a16 && b16) *status = full ? launch_variant<true, true, true>(a, s) : launch_variant<true, true, false>(a, s);
    else if (a16) *status = full ? launch_variant<true, false, true>(a, s) : launch_variant<true, false, false>(a, s);
    else if (b16) *status = full ? launch_variant<false, true, true>(a, s) : launch_variant<false, true, false>(a, s);
    else *status = full ? launch_variant<false, false, true>(a, s) : launch_variant<false, false, false>(a, s);
    return true;
}

}  // namespace mpnhip

// ---- C ABI (include/mpnhip.h) ---------------------------------------------------------------------------------------------
namespace {
int linear_bf16_call(const mpnhip_linear_bf16_args* p, hipStream_t s) {
    using namespace mpnhip;
    MPN_CHECK_ARG(p && p->x && p->w && p->y && p->m >= 0 && p->n > 0 && p->k > 0, "linear_bf16: bad argument");
    MPN_CHECK_ARG(p->ksplit >= 0 && p->ksplit <= p->k && (p->ksplit == p->k || p->x2), "linear_bf16: ksplit without a second segment");
    if (p->m == 0) return MPNHIP_OK;
    GemmArgs a = {};
    a.ngroups = 1; a.N = p->n; a.K = p->k; a.ksplit = p->x2 ? p->ksplit : p->k; a.relu = p->relu; a.accumulate = p->accumulate; a.m_upper = p->m;
    GemmGroup& g = a.g[0];
    g.A = static_cast<const float*>(p->x); g.lda = p->ldx;
    g.A2 = static_cast<const float*>(p->x2); g.lda2 = p->ldx2;
    g.B = static_cast<const float*>(p->w); g.ldb = p->ldw;
    g.a16 = p->x_bf16 ? 1 : 0; g.b16 = p->w_bf16 ? 1 : 0;
    g.bias = p->b;
    g.G1 = p->c_in; g.ldg1 = p->ldc_in;
    g.mask = p->mask; g.ldmask = p->ldmask;
    g.C = p->y; g.ldc = p->ldy;
    g.C16 = p->y16; g.ldc16 = p->ldy16;
    g.m_static = p->m;
    struct Scope { int old; Scope() : old(gemm_precision()) { set_gemm_precision(MPNHIP_PREC_BF16); } ~Scope() { set_gemm_precision(old); } } scope;
    return launch_gemm(a, A_KCONTIG, B_KCONTIG, s);
}
}  // namespace

extern "C" int mpnhip_linear_bf16(const mpnhip_linear_bf16_args* args, void* stream) {
    return linear_bf16_call(args, static_cast<hipStream_t>(stream));
}

extern "C" int mpnhip_to_bf16(const float* src, uint16_t* dst, int64_t n, void* stream) {
    using namespace mpnhip;
    MPN_CHECK_ARG(n >= 0 && (n == 0 || (src && dst)), "to_bf16: bad argument");
    if (n == 0) return MPNHIP_OK;
    return to_bf16_rows(src, dst, n, static_cast<hipStream_t>(stream));
}

extern "C" int mpnhip_time_linear_bf16(const mpnhip_linear_bf16_args* args, int iters, float* avg_us, void* stream) {
    using namespace mpnhip;
    hipStream_t s = static_cast<hipStream_t>(stream);
    MPN_CHECK_ARG(args && avg_us && iters > 0, "time_linear_bf16: bad argument");
    // (the first call validates the arguments and the shape BEFORE any event exists; the guard destroys both on every later exit)
    MPN_TRY(linear_bf16_call(args, s));
    struct Events {
        hipEvent_t t0 = nullptr, t1 = nullptr;
        ~Events() { if (t0) (void)hipEventDestroy(t0); if (t1) (void)hipEventDestroy(t1); }
    } ev;
    MPN_HIP(hipEventCreate(&ev.t0));
    MPN_HIP(hipEventCreate(&ev.t1));
    const hipEvent_t t0 = ev.t0, t1 = ev.t1;
    MPN_HIP(hipEventRecord(t0, s));
    for (int i = 0; i < iters; ++i) MPN_TRY(linear_bf16_call(args, s));
    MPN_HIP(hipEventRecord(t1, s));
    MPN_HIP(hipEventSynchronize(t1));
    float ms = 0.f;
    MPN_HIP(hipEventElapsedTime(&ms, t0, t1));
    *avg_us = ms * 1000.f / iters;
    return MPNHIP_OK;
}
