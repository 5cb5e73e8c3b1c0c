// Weight-gradient products in the three-piece operand form (MPNHIP_PREC_FP32_SPLIT), SURVEY.md section 3.4 / mlp.py:27-28 under autograd:
//   dW[o, c] += sum_m dZ[m, o] * H[m, c]        db[o] += sum_m dZ[m, o]
// The reduction runs over the rows (edges or nodes) and both operands are stored with the reduction index as the ROW: an HBM
// stream of (n_out + k_in) floats per row against a small [n_out, k_in] output -- 27 flop per byte for the chain's skinny
// layers.  What this kernel is built around:
//   * every operand row is fetched ONCE: a block owns a ROW PANEL (a chunk of rows) and the whole output [n_out, k_in] (or one of
//     a few large output tiles when n_out x k_in exceeds a block's accumulators), not a 64 x 128 tile that re-reads rows 2-5 times;
//   * fp32 operands are split into three bf16 pieces ONCE, as a 16-row stage passes through the loader's registers
//     (x = h + m + l exactly, edge_chain.hip), and written row-major into LDS (ds_write_b64 per piece: 4 columns of one row);
//   * the MFMA operands need the reduction index along the lane's 8 elements, i.e. the transpose of that image:
//     ds_read_b64_tr_b16 delivers it (4 rows x 16 columns per 16-lane group, column-major) -- no transposing store, no shuffles;
//     the row pitch is 64 (mod 256) bytes, so a half-wave's 4 rows x 64 bytes cover the 64 banks once;
//   * six piece products per k block on v_mfma_f32_32x32x16_bf16 (fp32 accumulate) = 3/8 of the fp32 MFMA's cycles, which puts
//     the product under the operand stream: the kernel is HBM-bound by design;
//   * all products of one group of steps go in ONE launch (job table: block -> (product, output tile, row chunk)) and their slabs
//     are summed by ONE launch (fixed order: deterministic, no float atomics).  Chunks with an odd index carry NEGATED dZ and
//     are subtracted: the bf16 MFMA's accumulate is biased toward -inf, and the alternation cancels the
//     bias across neighbouring chunks instead of adding it up over all rows.
#include <cstdlib>
#include <type_traits>
#include <utility>

#include "common.h"

namespace mpnhip {

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
#define WP_LDS __attribute__((address_space(3)))

constexpr int WP_NT = 256;   // threads per block: 4 waves as 2 (output rows) x 2 (output columns)
// wgrad_panel_narrow_kernel: <1, 1> (three 5 KB piece images), wp_block_vec, wp_block_small (up to 33.5 KB).  34 KB also keeps the
// CU at FOUR blocks (the kernel needs 100 registers: five would fit, and measured slower -- cfg-C 1.86 - 1.89 against 1.84 ms)
constexpr int WP_NARROW_LDS = 34 * 1024;
constexpr int WP_KB = 16;    // operand rows per stage = one k block of the bf16 MFMA

// loader passes for an operand of at most B columns: B / 4 threads cover a row, 256 / (B / 4) rows per pass (at most 16)
constexpr int wp_passes(int B) {
    int rp = WP_NT / (B / 4);
    rp = rp > WP_KB ? WP_KB : rp;
    return (WP_KB + rp - 1) / rp;
}
constexpr int wp_pitch(int BO, int BC) { return 2 * (BO + BC) + 64; }   // bytes; BO + BC is a multiple of 64 -> pitch = 64 or 192 (mod 256)

__device__ __forceinline__ float fneg_if(float x, unsigned sx) { return __uint_as_float(__float_as_uint(x) ^ sx); }

struct Pk3 { uint2 p[3]; };
// x = h + m + l, each piece a bf16 (round to nearest even; the residuals are exact in fp32).  Written pair-wise: one
// v_cvt_pk_bf16_f32 per piece and pair, the piece back as two floats by a shift and a mask, one packed subtraction.
__device__ __forceinline__ void split2(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    const bf16x2 hv = {(__bf16)x0, (__bf16)x1};
    h = __builtin_bit_cast(unsigned, hv);
    const float a0 = x0 - __uint_as_float(h << 16), a1 = x1 - __uint_as_float(h & 0xffff0000u);
    const bf16x2 mv = {(__bf16)a0, (__bf16)a1};
    m = __builtin_bit_cast(unsigned, mv);
    const float c0 = a0 - __uint_as_float(m << 16), c1 = a1 - __uint_as_float(m & 0xffff0000u);
    const bf16x2 lv = {(__bf16)c0, (__bf16)c1};
    l = __builtin_bit_cast(unsigned, lv);
}
__device__ __forceinline__ Pk3 split4(float4 v) {
    Pk3 o;
    split2(v.x, v.y, o.p[0].x, o.p[1].x, o.p[2].x);
    split2(v.z, v.w, o.p[0].y, o.p[1].y, o.p[2].y);
    return o;
}

// one MFMA operand piece: rows 8h .. 8h+7 of the stage, column (col0 + lane % 32), as 8 bf16 along k (two transposing reads)
template <int P>
__device__ __forceinline__ bf16x8 tr_read(const char* base) {
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((WP_LDS s16x4*)(base));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((WP_LDS s16x4*)(base + 4 * P));
    const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
}

template <int I, int N, class F>
__device__ __forceinline__ void wp_static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        wp_static_for<I + 1, N>(f);
    }
}

__device__ __forceinline__ void mfma6(f32x16& acc, const bf16x8 (&a)[3], const bf16x8 (&b)[3]) {
    // smallest products first (edge_chain.hip mfma6)
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], acc, 0, 0, 0);
}

// One block: output tile (tile_o, tile_c) of BO = 64 TM x BC = 64 TN, row chunk `by` of job J.
// B16: both operands are bf16 rows in memory (the dZ blocks and saved activations of the bf16-operand chain kernels,
// edge_chain_bf16*.hip): a stage is loaded as 8-byte pieces (4 columns) and stored to LDS as it is -- one piece, no split.
template <int TM, int TN, bool B16 = false>
__device__ __forceinline__ void wp_block(const WpJob& J, const int tile, const int by, char* lds, const int dbg = 0) {
    constexpr int BO = 64 * TM, BC = 64 * TN;
    constexpr int ES = B16 ? 2 : 4;   // bytes per source element
    using StageT = std::conditional_t<B16, uint2, f32x4>;
    constexpr int P = wp_pitch(BO, BC), PIECE = WP_KB * P;
    constexpr int PZ = wp_passes(BO), PH = wp_passes(BC);
    constexpr bool NEG_Z = BO <= BC;   // the sign of odd chunks goes onto the narrower operand (fewer XORs)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    const int rb = J.row_begin ? *J.row_begin : 0;
    const int re = J.row_end ? *J.row_end : (int)J.m_static;
    const int batch = by / J.nsplit, ci = by - batch * J.nsplit;
    const int r0 = rb + ci * J.chunk;
    int r1 = r0 + J.chunk;
    r1 = r1 < re ? r1 : re;
    if (r0 >= r1) return;   // empty chunk: the slab sum skips it too
    const unsigned sx = (ci & 1) ? 0x80000000u : 0u;   // odd chunks: one operand negated (see the header)

    const int tile_o = tile / J.tiles_c, tile_c = tile - tile_o * J.tiles_c;
    const int o0 = tile_o * BO, c0 = tile_c * BC;
    const int wo = J.n_out - o0 < BO ? J.n_out - o0 : BO;   // live columns of the dZ / H panels (multiples of 4)
    const int wc = J.k_in - c0 < BC ? J.k_in - c0 : BC;

    // loader geometry: T threads per row, RP rows per pass; thread (row, col) of pass j reads row + RP j
    const int tz = wo >> 2, th = wc >> 2;
    int rpz = WP_NT / tz; rpz = rpz > WP_KB ? WP_KB : rpz;
    int rph = WP_NT / th; rph = rph > WP_KB ? WP_KB : rph;
    const int zrow = tid / tz, zcol = (tid - zrow * tz) * 4;
    const int hrow = tid / th, hcol = (tid - hrow * th) * 4;
    const bool zact = zrow < rpz, hact = hrow < rph;
    // H may come as two column segments (the reference's torch.cat([initial, current]) input, mpn.py:369-373): columns
    // [0, csplit) from H, [csplit, k_in) from H2; csplit % 4 == 0, so a thread's four columns lie in one segment.  The H loads
    // therefore take per-thread 64-bit addresses (the dZ loads: one uniform base + 32-bit offsets)
    const bool hseg2 = J.H2 && c0 + hcol >= J.csplit;
    const int64_t ldz = J.ldz, ldh = hseg2 ? J.ldh2 : J.ldh;
    // uniform bases (SGPRs) + 32-bit per-thread byte offsets: the row of pass j clamped into the stage (clamped lanes do not store)
    const char* zbase = reinterpret_cast<const char*>(J.dZ) + ((int64_t)batch * J.z_bstride + o0) * ES;
    const char* hbase = hseg2 ? reinterpret_cast<const char*>(J.H2) + ((int64_t)batch * J.h2_bstride + (c0 - J.csplit)) * ES
                              : reinterpret_cast<const char*>(J.H) + ((int64_t)batch * J.h_bstride + c0) * ES;
    unsigned zoff[PZ], hoff[PH];
#pragma unroll
    for (int j = 0; j < PZ; ++j) {
        int r = zrow + rpz * j;
        r = (zact && r < WP_KB) ? r : WP_KB - 1;
        zoff[j] = (unsigned)(((int64_t)r * ldz + (zact ? zcol : 0)) * ES);
    }
#pragma unroll
    for (int j = 0; j < PH; ++j) {
        int r = hrow + rph * j;
        r = (hact && r < WP_KB) ? r : WP_KB - 1;
        hoff[j] = (unsigned)(((int64_t)r * ldh + hcol) * ES);   // (hcol < wc for every thread: in bounds in either segment)
    }
    char* zdst = lds + zrow * P + zcol * 2;
    char* hdst = lds + hrow * P + (BO + hcol) * 2;
    // (the pipelined split loop below keeps TWO stage images: stores go to image wimg, products read image rimg -- byte offsets)
    int wimg = 0, rimg = 0;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    float4 bsum = make_float4(0.f, 0.f, 0.f, 0.f);
    // stages in flight under the current one's products: ONE for the wide variants (a stage is 20 - 24 registers there; two fit
    // without spills since the loops exist once per operand form, and measure the same: 118.7 / 113.2 us against 114.2 / 110.3 on
    // the edge-level products, the step unchanged -- the kernel is not bound by its bytes in flight),
    // two or four for the narrow ones (a <1, 1> stage is 16 rows x 128 columns = 8 KB per block: with one in flight the kernel waits out
    // the HBM latency every 16 rows -- 1.4 TB/s on the 32-d products of the reference's configuration)
    // (bf16 source rows: a stage is half the registers -- two in flight up to eight accumulator tiles; with ten they spill)
    constexpr int D = B16 ? (TM + TN <= 3 ? 4 : (TM * TN <= 8 ? 2 : 1)) : (TM + TN <= 2 ? 4 : (TM + TN <= 5 && TM * TN <= 4 ? 2 : 1));
    StageT zreg[D][PZ], hreg[D][PH];

    // full stages: rows m0 .. m0 + 15 all inside the chunk.  (Compiler-visible loads on purpose.  Inline-assembly loads into TWO
    // stage buffers with hand-counted s_waitcnt kept two whole stages in flight -- hipcc's own placement waits for every
    // outstanding load before the first use of a stage -- and were 3 % faster on the long products (110.5 vs 113.9 us), but the
    // registers of an asm load are, to the compiler, ready when the statement ends: wherever its allocation put a copy or reused
    // one of them before the hand-placed wait, the late-landing load overwrote live values -- memory faults on short chunks of
    // a since-removed <2, 4> variant.  Not worth 3 %.)
#define WP_LOAD(ZR, HR, m0)                                                                                              \
    do {                                                                                                                 \
        const char* zb_ = zbase + (int64_t)(m0) * ldz * ES;                                                              \
        const char* hb_ = hbase + (int64_t)(m0) * ldh * ES;                                                              \
        _Pragma("unroll") for (int j = 0; j < PZ; ++j) ZR[j] = *reinterpret_cast<const StageT*>(zb_ + zoff[j]);          \
        _Pragma("unroll") for (int j = 0; j < PH; ++j) HR[j] = *reinterpret_cast<const StageT*>(hb_ + hoff[j]);          \
    } while (0)
    const bool one = B16 || J.pieces == 1;   // MPNHIP_PREC_BF16: operands rounded to bf16 (the first piece alone), one product per k block
    const unsigned sx16 = sx ? 0x80008000u : 0u;   // (the sign of two packed bf16)
    auto put = [&](char* d, float4 v, bool negate) {
        if (negate) { v.x = fneg_if(v.x, sx); v.y = fneg_if(v.y, sx); v.z = fneg_if(v.z, sx); v.w = fneg_if(v.w, sx); }
        const Pk3 s = split4(v);
        *reinterpret_cast<uint2*>(d) = s.p[0];
        if (!one) {
#pragma unroll
            for (int q = 1; q < 3; ++q) *reinterpret_cast<uint2*>(d + q * PIECE) = s.p[q];
        }
    };
    // nrows: live rows of the stage (16 for a full one); rows past it are stored as zeros
    auto store = [&](auto bsel, int nrows) {
        constexpr int B = decltype(bsel)::value;
#pragma unroll
        for (int j = 0; j < PZ; ++j) {
            const int row = zrow + rpz * j;
            if (zact && row < WP_KB) {
                if constexpr (B16) {
                    uint2 v = zreg[B][j];
                    if (row >= nrows) v = make_uint2(0u, 0u);
                    bsum.x += __uint_as_float(v.x << 16); bsum.y += __uint_as_float(v.x & 0xffff0000u);
                    bsum.z += __uint_as_float(v.y << 16); bsum.w += __uint_as_float(v.y & 0xffff0000u);
                    if (NEG_Z) { v.x ^= sx16; v.y ^= sx16; }
                    *reinterpret_cast<uint2*>(zdst + wimg + rpz * j * P) = v;
                } else {
                    float4 v = make_float4(zreg[B][j][0], zreg[B][j][1], zreg[B][j][2], zreg[B][j][3]);
                    if (row >= nrows) v = make_float4(0.f, 0.f, 0.f, 0.f);
                    bsum.x += v.x; bsum.y += v.y; bsum.z += v.z; bsum.w += v.w;
                    put(zdst + wimg + rpz * j * P, v, NEG_Z);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < PH; ++j) {
            const int row = hrow + rph * j;
            if (hact && row < WP_KB) {
                if constexpr (B16) {
                    uint2 v = hreg[B][j];
                    if (row >= nrows) v = make_uint2(0u, 0u);
                    if (!NEG_Z) { v.x ^= sx16; v.y ^= sx16; }
                    *reinterpret_cast<uint2*>(hdst + wimg + rph * j * P) = v;
                } else {
                    float4 v = make_float4(hreg[B][j][0], hreg[B][j][1], hreg[B][j][2], hreg[B][j][3]);
                    if (row >= nrows) v = make_float4(0.f, 0.f, 0.f, 0.f);
                    put(hdst + wimg + rph * j * P, v, !NEG_Z);
                }
            }
        }
    };

    // transposing-read lane geometry: lane 4q + p of a 16-lane group supplies row q, columns 4p .. 4p+3 of the group's block
    const int lh = lane >> 5, gi = (lane >> 4) & 1, lq = (lane & 15) >> 2, lp = lane & 3;
    const char* const rda0 = lds + (8 * lh + lq) * P + (16 * gi + 4 * lp) * 2 + 32 * (wm * TM) * 2;
    const char* const rdb0 = lds + (8 * lh + lq) * P + (16 * gi + 4 * lp) * 2 + (BO + 32 * (wn * TN)) * 2;
    const char* const rda = rda0;
    const char* const rdb = rdb0;
    // every tile of the wave is computed (columns past n_out / k_in hold whatever the LDS held: those outputs are never stored);
    // the operand pieces of the next tile are fetched before the six products of the current one
    auto products1 = [&]() {   // one bf16 piece per operand
        bf16x8 b[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) b[j] = tr_read<P>(rdb + 64 * j);
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const bf16x8 a = tr_read<P>(rda + 64 * i);
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b[j], acc[i][j], 0, 0, 0);
        }
    };
    auto products6 = [&]() {
        const char* const rda = rda0 + rimg;
        const char* const rdb = rdb0 + rimg;
        if constexpr (TM >= TN) {
            bf16x8 b[TN][3];
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int q = 0; q < 3; ++q) b[j][q] = tr_read<P>(rdb + q * PIECE + 64 * j);
            bf16x8 a[2][3];
#pragma unroll
            for (int q = 0; q < 3; ++q) a[0][q] = tr_read<P>(rda + q * PIECE);
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                if (i + 1 < TM)
#pragma unroll
                    for (int q = 0; q < 3; ++q) a[(i + 1) & 1][q] = tr_read<P>(rda + q * PIECE + 64 * (i + 1));
#pragma unroll
                for (int j = 0; j < TN; ++j) mfma6(acc[i][j], a[i & 1], b[j]);
            }
        } else {
            bf16x8 a[TM][3];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int q = 0; q < 3; ++q) a[i][q] = tr_read<P>(rda + q * PIECE + 64 * i);
            bf16x8 b[2][3];
#pragma unroll
            for (int q = 0; q < 3; ++q) b[0][q] = tr_read<P>(rdb + q * PIECE);
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                if (j + 1 < TN)
#pragma unroll
                    for (int q = 0; q < 3; ++q) b[(j + 1) & 1][q] = tr_read<P>(rdb + q * PIECE + 64 * (j + 1));
#pragma unroll
                for (int i = 0; i < TM; ++i) mfma6(acc[i][j], a[i], b[(j & 1)]);
            }
        }
    };
    using B0 = std::integral_constant<int, 0>;
    const int nfull = (r1 - r0) / WP_KB;          // full stages
    const int tail = (r1 - r0) - nfull * WP_KB;   // rows of the partial last stage (0: none)
    // (one copy of the loops per operand form: a branch on `one` inside them makes every accumulator a phi of two versions --
    // register copies and spills)
    auto run = [&](auto&& products) {
        if (nfull > 0) {
            wp_static_for<0, D>([&](auto d) {
                if (d.value < nfull) WP_LOAD(zreg[d.value], hreg[d.value], r0 + d.value * WP_KB);
            });
            int st = 0;
            // steady state (every stage of the round has a successor D stages on: straight-line code, the loads' waits are counted)
            for (; st + 2 * D <= nfull; st += D) {
                wp_static_for<0, D>([&](auto d) {
#ifndef MPNHIP_WP_ABLATE
                    store(d, WP_KB);
                    __syncthreads();
                    WP_LOAD(zreg[d.value], hreg[d.value], r0 + (st + d.value + D) * WP_KB);
                    products();
                    __syncthreads();
#else
                    {   // ablation build (make EXTRA=-DMPNHIP_WP_ABLATE, MPNHIP_WP_DEBUG=bits): 1 no products, 2 no split / LDS stores, 4 no loads, 8 no barriers
                        if (!(dbg & 2)) store(d, WP_KB);
                        else {
                            _Pragma("unroll") for (int j = 0; j < PZ; ++j) asm volatile("" :: "v"(zreg[d.value][j]));
                            _Pragma("unroll") for (int j = 0; j < PH; ++j) asm volatile("" :: "v"(hreg[d.value][j]));
                        }
                        if (!(dbg & 8)) __syncthreads();
                        if (!(dbg & 4)) WP_LOAD(zreg[d.value], hreg[d.value], r0 + (st + d.value + D) * WP_KB);
                        if (!(dbg & 1)) products();
                        if (!(dbg & 8)) __syncthreads();
                    }
#endif
                });
            }
            for (; st < nfull; st += D) {
                wp_static_for<0, D>([&](auto d) {
                    if (st + d.value < nfull) {   // (block-uniform)
                        store(d, WP_KB);
                        __syncthreads();
                        if (st + d.value + D < nfull) WP_LOAD(zreg[d.value], hreg[d.value], r0 + (st + d.value + D) * WP_KB);
                        products();
                        __syncthreads();
                    }
                });
            }
        }
        if (tail > 0) {
            // the partial stage: rows clamped to the chunk's last row, zeros stored past it
            const int m0 = r0 + nfull * WP_KB;
    #pragma unroll
            for (int j = 0; j < PZ; ++j) {
                int r = zrow + rpz * j;
                r = (zact && r < tail) ? r : tail - 1;
                zreg[0][j] = *reinterpret_cast<const StageT*>(zbase + ((int64_t)(m0 + r) * ldz + (zact ? zcol : 0)) * ES);
            }
    #pragma unroll
            for (int j = 0; j < PH; ++j) {
                int r = hrow + rph * j;
                r = (hact && r < tail) ? r : tail - 1;
                hreg[0][j] = *reinterpret_cast<const StageT*>(hbase + ((int64_t)(m0 + r) * ldh + hcol) * ES);
            }
            store(B0{}, tail);
            __syncthreads();
            products();
            __syncthreads();
        }
    };
    // MPNHIP_PREC_FP32_SPLIT, wide variants (one register stage): the split of stage st + 1 (VALU + ds_write into the OTHER image)
    // and the products of stage st (transposing reads + MFMAs) sit in one scheduling region between two barriers -- ONE barrier
    // per stage -- so that the operand split runs in the MFMAs' issue gaps instead of before them (ablation build, cfg-B's
    // 50,000 x 320 x 64 product alone: loads only 60 us, split + stores only 28, products only 34, the old loop's split +
    // products 65 = their SUM: the two blocks of a CU did not overlap one's split with the other's MFMAs).
    auto run_pipelined = [&]() {
        constexpr int IMG = 3 * PIECE;
        if (nfull > 0) {
            WP_LOAD(zreg[0], hreg[0], r0);
            wimg = 0;
            store(B0{}, WP_KB);
            if (nfull > 1) WP_LOAD(zreg[0], hreg[0], r0 + WP_KB);
            __syncthreads();
            int st = 0;
            for (; st + 2 < nfull; st += 2) {   // straight-line pairs: image offsets are compile-time constants
                wimg = IMG; rimg = 0;
                store(B0{}, WP_KB);                                   // stage st + 1 -> image 1
                WP_LOAD(zreg[0], hreg[0], r0 + (st + 2) * WP_KB);
                products6();                                          // stage st <- image 0
                __syncthreads();
                wimg = 0; rimg = IMG;
                store(B0{}, WP_KB);                                   // stage st + 2 -> image 0
                if (st + 3 < nfull) WP_LOAD(zreg[0], hreg[0], r0 + (st + 3) * WP_KB);
                products6();                                          // stage st + 1 <- image 1
                __syncthreads();
            }
            // one or two stages left: stage st is in image 0, stage st + 1 (if any) in the registers
            if (st + 1 < nfull) {
                wimg = IMG; rimg = 0;
                store(B0{}, WP_KB);
                products6();
                __syncthreads();
                rimg = IMG;
                products6();
                __syncthreads();
            } else {
                rimg = 0;
                products6();
                __syncthreads();
            }
            wimg = 0; rimg = 0;
        }
        if (tail > 0) {
            const int m0 = r0 + nfull * WP_KB;
#pragma unroll
            for (int j = 0; j < PZ; ++j) {
                int r = zrow + rpz * j;
                r = (zact && r < tail) ? r : tail - 1;
                zreg[0][j] = *reinterpret_cast<const StageT*>(zbase + ((int64_t)(m0 + r) * ldz + (zact ? zcol : 0)) * ES);
            }
#pragma unroll
            for (int j = 0; j < PH; ++j) {
                int r = hrow + rph * j;
                r = (hact && r < tail) ? r : tail - 1;
                hreg[0][j] = *reinterpret_cast<const StageT*>(hbase + ((int64_t)(m0 + r) * ldh + hcol) * ES);
            }
            store(B0{}, tail);
            __syncthreads();
            products6();
            __syncthreads();
        }
    };
    if constexpr (B16) {
        run(products1);
    } else {
        if (one) run(products1);
        else if (D == 1 && J.lds2) run_pipelined();
        else run(products6);
    }

    // ---- the partial output tile into this chunk's slab (odd chunks negated as a whole: the slab sum subtracts them) ----
    const int kpad = tn_kpad(J.k_in);
    float* slab = J.slab + (size_t)by * J.n_out * kpad;
    const int li = lane & 31;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int c = c0 + 32 * (wn * TN + j) + li;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int o = o0 + 32 * (wm * TM + i) + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (o < J.n_out && c < J.k_in) slab[(size_t)o * kpad + c] = acc[i][j][r];
            }
        }
    if (tile_c == 0) {
        // bias partials: the rpz row groups' column sums meet in LDS (free after the loop's last barrier), fixed order; summed
        // from the dZ values as loaded, so they take the chunk's sign here
        float* bs = reinterpret_cast<float*>(lds);
        if (zact) *reinterpret_cast<float4*>(bs + zrow * BO + zcol) = bsum;
        __syncthreads();
        for (int t = tid; t < wo; t += WP_NT) {
            float s = 0.f;
            for (int r = 0; r < rpz; ++r) s += bs[r * BO + t];
            slab[(size_t)(o0 + t) * kpad + J.k_in] = sx ? -s : s;
        }
    }
}

#undef WP_LOAD

// n_out == 1, dZ rows optionally gathered (the classifier's output layer: dZ = grad_logits, one float per edge in ORIGINAL edge
// order, read through the sort permutation): out[c] = sum_r z[idx[r]] H[r][c] with plain fp32 FMAs -- k_in / 4 threads per row,
// 256 / (k_in / 4) row lanes whose partial sums meet in LDS in a fixed order.  Rides in the products' launch instead of three
// small launches of its own per group of steps.
__device__ __forceinline__ void wp_block_vec(const WpJob& J, const int by, char* lds) {
    const int tid = threadIdx.x;
    const int rb = J.row_begin ? *J.row_begin : 0;
    const int re = J.row_end ? *J.row_end : (int)J.m_static;
    const int batch = by / J.nsplit, ci = by - batch * J.nsplit;
    const int r0 = rb + ci * J.chunk;
    int r1 = r0 + J.chunk;
    r1 = r1 < re ? r1 : re;
    if (r0 >= r1) return;
    const int kc = J.k_in, ng = kc >> 2, rl = WP_NT / ng;
    const int rlane = tid / ng, cg = tid - rlane * ng;
    const float* z = J.dZ + (int64_t)batch * J.z_bstride;
    const int* zi = J.dz_idx;
    const float* h = J.H + (int64_t)batch * J.h_bstride + 4 * cg;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    float bs = 0.f;
    if (rlane < rl) {
        // eight rows in flight per thread: the index, the gathered dZ element and the H piece of a row are a chain of dependent loads
        // (one row at a time this loop ran at one memory round trip per 16 rows: 1.57 ms alone for the 19,056-row chunks of cfg-E)
        constexpr int U = 8;
        for (int r = r0 + rlane; r < r1; r += rl * U) {
            int64_t zo[U];
            int rc[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                rc[u] = r + u * rl < r1 ? r + u * rl : r;   // clamped: unconditional loads, predicated accumulation
                zo[u] = (int64_t)(zi ? zi[rc[u]] : rc[u]) * J.ldz;
            }
            float zr[U];
            float4 hq[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                zr[u] = z[zo[u]];
                if (J.src16) {   // H as bf16 rows (J.H points at unsigned shorts; ldh / h_bstride count them); dZ stays fp32 here
                    const uint2 q = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(J.H) + (int64_t)batch * J.h_bstride +
                                                                    4 * cg + (int64_t)rc[u] * J.ldh);
                    hq[u] = make_float4(__uint_as_float(q.x << 16), __uint_as_float(q.x & 0xffff0000u), __uint_as_float(q.y << 16),
                                        __uint_as_float(q.y & 0xffff0000u));
                } else {
                    hq[u] = *reinterpret_cast<const float4*>(h + (int64_t)rc[u] * J.ldh);
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (r + u * rl < r1) {
                    float zz = zr[u];
                    float4 hv = hq[u];
                    if (J.pieces == 1) {   // (MPNHIP_PREC_BF16: the same operand rounding as the MFMA jobs)
                        zz = (float)(__bf16)zz;
                        hv = make_float4((float)(__bf16)hv.x, (float)(__bf16)hv.y, (float)(__bf16)hv.z, (float)(__bf16)hv.w);
                    }
                    acc.x = fmaf(zz, hv.x, acc.x); acc.y = fmaf(zz, hv.y, acc.y); acc.z = fmaf(zz, hv.z, acc.z); acc.w = fmaf(zz, hv.w, acc.w);
                    bs += zr[u];   // (the bias gradient is a plain sum: not a product, nothing rounded)
                }
            }
        }
    }
    float* part = reinterpret_cast<float*>(lds);   // [rl][kc + 4]
    const int pitch = kc + 4;
    if (rlane < rl) {
        *reinterpret_cast<float4*>(part + rlane * pitch + 4 * cg) = acc;
        if (cg == 0) part[rlane * pitch + kc] = bs;
    }
    __syncthreads();
    if (tid <= kc) {
        float sum = 0.f;
        for (int r = 0; r < rl; ++r) sum += part[r * pitch + tid];
        J.slab[(size_t)by * tn_kpad(kc) + tid] = (ci & 1) ? -sum : sum;   // (odd chunks are subtracted by the slab sum)
    }
}

// Narrow products (k_in <= 32 and n_out <= 32 -- or n_out <= 96 while n_out (k_in + 1) <= 1280: the edge encoder's first layer
// at 128-d, [72 x 6] --, any alignment; rows of either operand optionally gathered: the reference's 18-wide
// edge encoder -- its first layer reads edge_attr through the sort permutation --, classifier layers): a block stages 64 rows of
// dZ and H in LDS and every thread owns up to five output elements (o, c) -- c == k_in is the bias column, fed by a column of
// ones -- reading dZ as a broadcast and H conflict-free; plain fp32 FMAs (gemm_tn_small_kernel's scheme, as a job of this launch).
__device__ __forceinline__ void wp_block_small(const WpJob& J, const int by, char* lds) {
    constexpr int ROWS = 64;
    const int n_out = J.n_out, k_in = J.k_in, kc = k_in + 1;
    const int zp = n_out <= 32 ? 33 : 97;   // (odd row pitch)
    float* const zs = reinterpret_cast<float*>(lds);
    float (*hs)[34] = reinterpret_cast<float (*)[34]>(lds + ROWS * zp * sizeof(float));
    const int nout_total = n_out * kc;
    const int rb = J.row_begin ? *J.row_begin : 0;
    const int re = J.row_end ? *J.row_end : (int)J.m_static;
    const int batch = by / J.nsplit, ci = by - batch * J.nsplit;
    const int r0 = rb + ci * J.chunk;
    int r1 = r0 + J.chunk;
    r1 = r1 < re ? r1 : re;
    if (r0 >= r1) return;
    const float* dZ = J.dZ + (int64_t)batch * J.z_bstride;
    const float* H = J.H + (int64_t)batch * J.h_bstride;
    const int* zi = J.dz_idx;
    const int* hi = J.h_idx;
    float acc[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    int oo[5], cc[5];
#pragma unroll
    for (int q = 0; q < 5; ++q) {
        const int t = threadIdx.x + WP_NT * q;
        oo[q] = t < nout_total ? t / kc : 0;
        cc[q] = t < nout_total ? t % kc : 0;
    }
    for (int m0 = r0; m0 < r1; m0 += ROWS) {
        const int nr = r1 - m0 < ROWS ? r1 - m0 : ROWS;
        for (int i = threadIdx.x; i < ROWS * n_out; i += WP_NT) {
            const int r = i / n_out, o = i - r * n_out;
            const int64_t row = r < nr ? (zi ? zi[m0 + r] : m0 + r) : 0;
            float v = 0.f;
            if (r < nr) v = J.src16 ? __uint_as_float((unsigned)reinterpret_cast<const unsigned short*>(J.dZ)[(int64_t)batch * J.z_bstride + row * J.ldz + o] << 16)
                                    : dZ[row * J.ldz + o];
            zs[r * zp + o] = v;
        }
        for (int i = threadIdx.x; i < ROWS * kc; i += WP_NT) {
            const int r = i / kc, c = i - r * kc;
            const int64_t row = r < nr ? (hi ? hi[m0 + r] : m0 + r) : 0;
            float v = c == k_in ? (r < nr ? 1.f : 0.f) : 0.f;
            if (c != k_in && r < nr) v = J.src16 ? __uint_as_float((unsigned)reinterpret_cast<const unsigned short*>(J.H)[(int64_t)batch * J.h_bstride + row * J.ldh + c] << 16)
                                                 : H[row * J.ldh + c];
            hs[r][c] = J.pieces == 1 ? (float)(__bf16)v : v;
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 5; ++q) {
            if (threadIdx.x + WP_NT * q < nout_total) {
                float sacc = acc[q];
                if (J.pieces == 1 && cc[q] != k_in) {   // (bias column: a plain sum of dZ, nothing rounded)
#pragma unroll 8
                    for (int r = 0; r < ROWS; ++r) sacc = fmaf((float)(__bf16)zs[r * zp + oo[q]], hs[r][cc[q]], sacc);
                } else {
#pragma unroll 8
                    for (int r = 0; r < ROWS; ++r) sacc = fmaf(zs[r * zp + oo[q]], hs[r][cc[q]], sacc);
                }
                acc[q] = sacc;
            }
        }
        __syncthreads();
    }
    float* slab = J.slab + (size_t)by * n_out * tn_kpad(k_in);
    const float sg = (ci & 1) ? -1.f : 1.f;   // (odd chunks are subtracted by the slab sum)
#pragma unroll
    for (int q = 0; q < 5; ++q)
        if (threadIdx.x + WP_NT * q < nout_total) slab[(size_t)oo[q] * tn_kpad(k_in) + cc[q]] = sg * acc[q];
}

}  // namespace

// NARROW: the launch holds <1, 1> tiles, [1 x k] and small-shape jobs only (the 32-d models: every job re-tiled to 64 x 64 at flush time,
// wp_batch_flush) -- 128 registers and 34 KB of LDS per block, FOUR blocks per CU instead of two: such a block spends its time in the
// two barriers and the LDS round trip of 16-row stages that hold six MFMAs per wave, and more resident blocks are what hides them
template <bool NARROW>
__device__ __forceinline__ void wp_kernel_body(const WpTable& tab, char* wp_lds) {
    const int b = blockIdx.x;
    int j = -1;
    for (int i = 0; i < tab.njobs; ++i)
        if (tab.job[i].variant < 16 && b >= tab.job[i].block0 &&
            b < tab.job[i].block0 + tab.job[i].tiles_o * tab.job[i].tiles_c * tab.job[i].nsplit * tab.job[i].nbatch) j = i;   // (variants >= 16: wgrad_rows16.hip's launch)
    if (j < 0) return;
    const WpJob& J = tab.job[j];
    const int local = b - J.block0;
    const int ntiles = J.tiles_o * J.tiles_c;
    const int by = local / ntiles, tile = local - by * ntiles;
    if constexpr (NARROW) {
        switch (J.variant) {
            case 6: wp_block_vec(J, by, wp_lds); break;
            case 7: wp_block_small(J, by, wp_lds); break;
            default: wp_block<1, 1>(J, tile, by, wp_lds, tab.debug); break;
        }
    } else {
        switch (J.variant) {
            case 0: wp_block<5, 1>(J, tile, by, wp_lds, tab.debug); break;
            case 1: wp_block<1, 5>(J, tile, by, wp_lds, tab.debug); break;
            case 2: wp_block<4, 1>(J, tile, by, wp_lds, tab.debug); break;
            case 3: wp_block<1, 1>(J, tile, by, wp_lds, tab.debug); break;
            case 6: wp_block_vec(J, by, wp_lds); break;
            case 7: wp_block_small(J, by, wp_lds); break;
            // bf16 source rows (one piece: ten accumulator tiles per wave fit): the shapes of the 256-d / 128-d / narrower models
            case 8: wp_block<5, 2, true>(J, tile, by, wp_lds, tab.debug); break;
            case 9: wp_block<2, 5, true>(J, tile, by, wp_lds, tab.debug); break;
            case 10: wp_block<4, 2, true>(J, tile, by, wp_lds, tab.debug); break;
            case 11: wp_block<2, 4, true>(J, tile, by, wp_lds, tab.debug); break;
            case 12: wp_block<1, 2, true>(J, tile, by, wp_lds, tab.debug); break;
            case 13: wp_block<2, 1, true>(J, tile, by, wp_lds, tab.debug); break;
            case 14: wp_block<1, 1, true>(J, tile, by, wp_lds, tab.debug); break;
            case 15: wp_block<2, 2, true>(J, tile, by, wp_lds, tab.debug); break;
            default: wp_block<2, 2>(J, tile, by, wp_lds, tab.debug); break;
        }
    }
}

__global__ __launch_bounds__(WP_NT, 2) void wgrad_panel_kernel(WpTable tab) {
    extern __shared__ __attribute__((aligned(16))) char wp_lds[];
    wp_kernel_body<false>(tab, wp_lds);
}
__global__ __launch_bounds__(WP_NT, 4) void wgrad_panel_narrow_kernel(WpTable tab) {
    extern __shared__ __attribute__((aligned(16))) char wp_lds[];
    wp_kernel_body<true>(tab, wp_lds);
}

// grad_w[o * ldw + c] += sum_j (-1)^chunk(j) slab_j[o][c];  grad_b[o] += ... slab_j[o][k_in]  over the non-empty chunks of every
// job of the table.  Block = 32 x 16-byte columns of one job's padded slab image x 8 slab groups (a wave reads 512 contiguous
// bytes of one slab, four slabs in flight per lane); the groups' partial sums meet in LDS in a fixed order.
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(WpTable tab) {
    __shared__ float4 part[8][32];
    const int b = blockIdx.x;
    int j = 0;
    for (int i = 1; i < tab.njobs; ++i) j = b >= tab.job[i].red_block0 ? i : j;
    const WpJob& J = tab.job[j];
    const int kpad = tn_kpad(J.k_in);
    const int64_t total4 = (int64_t)J.n_out * kpad / 4;
    const int64_t q = (int64_t)(b - J.red_block0) * 32 + (threadIdx.x & 31);
    const int grp = threadIdx.x >> 5;
    const int rb = J.row_begin ? *J.row_begin : 0;
    const int re = J.row_end ? *J.row_end : (int)J.m_static;
    int nvalid = (re - rb + J.chunk - 1) / J.chunk;
    nvalid = nvalid < 0 ? 0 : (nvalid > J.nsplit ? J.nsplit : nvalid);
    const int nslab = nvalid * J.nbatch;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const size_t stride = (size_t)J.n_out * kpad;
    if (q < total4) {
        const float* p = J.slab + q * 4;
        for (int j0 = grp; j0 < nslab; j0 += 32) {
            float4 v[4];
            float sg[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                int jj = j0 + 8 * u;
                const bool ok = jj < nslab;
                jj = ok ? jj : j0;   // clamped, unconditional loads
                const int bb = jj / nvalid, ci = jj - bb * nvalid;
                sg[u] = ok ? ((ci & 1) ? -1.f : 1.f) : 0.f;
                v[u] = *reinterpret_cast<const float4*>(p + ((size_t)bb * J.nsplit + ci) * stride);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                acc.x = fmaf(sg[u], v[u].x, acc.x); acc.y = fmaf(sg[u], v[u].y, acc.y);
                acc.z = fmaf(sg[u], v[u].z, acc.z); acc.w = fmaf(sg[u], v[u].w, acc.w);
            }
        }
    }
    part[grp][threadIdx.x & 31] = acc;
    __syncthreads();
    if (grp == 0 && q < total4) {
        float4 s = part[0][threadIdx.x];
#pragma unroll
        for (int g = 1; g < 8; ++g) {
            const float4 o = part[g][threadIdx.x];
            s.x += o.x; s.y += o.y; s.z += o.z; s.w += o.w;
        }
        const float sv[4] = {s.x, s.y, s.z, s.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int64_t t = q * 4 + e;
            const int o = (int)(t / kpad), c = (int)(t - (int64_t)o * kpad);
            if (c < J.k_in) { if (J.grad_w) J.grad_w[(int64_t)o * J.ldw + c] += sv[e]; }
            else if (c == J.k_in) { if (J.grad_b) J.grad_b[o] += sv[e]; }
        }
    }
}

// ------------------------------------------------------------------------------------ host side
namespace {

struct WpVariant { int tm, tn; };
const WpVariant kVariants[6] = {{5, 1}, {1, 5}, {4, 1}, {1, 1}, {2, 4}, {2, 2}};

// the variant whose tiles cover [n_out, k_in] with the fewest staged columns per operand row (ties: fewer tiles)
bool wp_is_small(int n_out, int k_in) {   // wp_block_small's shapes
    return k_in <= 32 && (n_out <= 32 || (n_out <= 96 && k_in % 4 != 0));
}
const WpVariant kVariants16[8] = {{5, 2}, {2, 5}, {4, 2}, {2, 4}, {1, 2}, {2, 1}, {1, 1}, {2, 2}};   // kernel cases 8 .. 15

// rows16: the product's operands meet the LDS-DMA kernel's alignment (16-byte rows): its variants 16 .. 19 where the shape has one
void wp_choose(int n_out, int k_in, int* variant, int* tiles_o, int* tiles_c, bool src16 = false, bool rows16 = false) {
    if (n_out == 1 && k_in % 4 == 0 && k_in <= 64) { *variant = 6; *tiles_o = 1; *tiles_c = 1; return; }   // wp_block_vec
    if (wp_is_small(n_out, k_in)) { *variant = 7; *tiles_o = 1; *tiles_c = 1; return; }                   // wp_block_small
    if (src16 && rows16) {
        int to = 1, tc = 1;
        const int v = r16_variant(n_out, k_in, &to, &tc);
        if (v >= 0) { *variant = v; *tiles_o = to; *tiles_c = tc; return; }
    }
    if (src16) {
        // cost = the columns every operand row is staged with, summed over the output tiles (a partial last tile stages only its
        // live columns), + a term for the accumulator tiles that stay empty (MFMAs on zeros)
        long best = -1;
        for (int v = 0; v < 8; ++v) {
            const int bo = 64 * kVariants16[v].tm, bc = 64 * kVariants16[v].tn;
            const int to = (n_out + bo - 1) / bo, tc = (k_in + bc - 1) / bc;
            long staged = 0;
            for (int i = 0; i < to; ++i)
                for (int j = 0; j < tc; ++j) staged += (n_out - i * bo < bo ? n_out - i * bo : bo) + (k_in - j * bc < bc ? k_in - j * bc : bc);
            const long waste = (long)to * tc * bo * bc - (long)n_out * k_in;
            const long cost = staged * 4096 + waste / 16 + to * tc;
            if (best < 0 || cost < best) { best = cost; *variant = 8 + v; *tiles_o = to; *tiles_c = tc; }
        }
        return;
    }
    long best = -1;
    for (int v = 0; v < 6; ++v) {
        if (v == 4) continue;   // (<2, 4>: 128 accumulator registers + a stage in flight spill; 128 x 224 takes two <2, 2> tiles)
        const int bo = 64 * kVariants[v].tm, bc = 64 * kVariants[v].tn;
        const int to = (n_out + bo - 1) / bo, tc = (k_in + bc - 1) / bc;
        const long cost = (long)to * tc * (bo + bc) * 64 + to * tc;
        if (best < 0 || cost < best) { best = cost; *variant = v; *tiles_o = to; *tiles_c = tc; }
    }
}

// blocks wanted per job: a job alone needs enough row chunks to fill the chip (2 blocks per CU); the jobs of a batch run side by
// side in one launch, and every extra chunk costs a slab (written, then read by the slab sum: at 512 chunks per job the slabs of a
// cfg-B group were 30 % of the launch's traffic).  192 per batched job -- same-box sweeps, three runs each, with the launches' blocks
// dispatched longest first: cfg-B step 5.27 / 4.93 - 4.97 / 4.91 - 4.93 / 4.94 - 4.96 / 4.98 - 4.99 / 5.02 - 5.08 ms at 96 / 128 / 192 / 224 /
// 256 / 320; cfg-C 1.82 - 1.83 / 1.75 / 1.74 - 1.76 at 128 / 192 / 256; cfg-E (every job) 33.74 - 33.85 / 33.48 - 33.57 / 33.02 - 33.24 /
// 33.47 - 33.58 / 33.40 - 33.57 at 128 / 160 / 192 / 224 / 256 (round 3, before the dispatch order: 5.43 / 5.37 / 5.29 at 320 / 192 / 128)
int wp_target_blocks(bool batched) {
    static const int alone = [] { const char* e = getenv("MPNHIP_WP_BLOCKS"); const int x = e ? atoi(e) : 0; return x >= 16 ? x : 512; }();
    static const int shared = [] { const char* e = getenv("MPNHIP_WP_BLOCKS_BATCH"); const int x = e ? atoi(e) : 0; return x >= 16 ? x : 192; }();
    return batched ? shared : alone;
}

// rows per chunk / chunks per batch of one job: ~wp_target_blocks() blocks per job, chunks of at least 256 rows
// light (a slab of at most 16K floats per chunk: [1 x k], the small shapes, skinny products like the classifier's [64 x 128]): chunks of
// at most 4096 rows -- a block of this kernel walks its rows at ~0.7 - 1.5 us per 16-row stage whatever the width (one stage in
// flight), and extra chunks of these jobs cost next to nothing in slabs (the three skinny jobs of a six-step group at cfg-E,
// 19,056-row chunks: 1.76 ms alone at the end of the step for 1.2 GB)
void wp_plan(int64_t rows_expected, int64_t rows_upper, int nbatch, int tiles, bool batched, int* chunk, int* nsplit, int min_chunk = 256,
             bool light = false) {
    if (rows_expected < 1) rows_expected = 1;
    int want = wp_target_blocks(batched) / (nbatch * tiles);
    if (want < 1) want = 1;
    int64_t c = (rows_expected + want - 1) / want;
    static const bool cap = !getenv("MPNHIP_WP_NO_LIGHT_CAP");
    if (light && cap && c > 4096) c = 4096;
    if (c < min_chunk) c = min_chunk;
    c = (c + WP_KB - 1) / WP_KB * WP_KB;
    *chunk = (int)c;
    *nsplit = (int)((rows_upper + c - 1) / c);
    if (*nsplit < 1) *nsplit = 1;
}

bool wp_light(int variant, int n_out, int k_in) { return variant < 16 && (size_t)n_out * tn_kpad(k_in) <= 16384; }

thread_local WpBatch* g_wp = nullptr;

size_t wp_lds_bytes() {
    // the largest variant image (BO + BC = 384): 39 KB.  MPNHIP_WP_LDS=<bytes> asks for more than the kernel uses: above 80 KB
    // only ONE block of this kernel fits a CU, which leaves the other wave slot of every SIMD to the caller's stream (A-B switch)
    // (two stage images: the pipelined split loop; MPNHIP_WP_ONE_IMAGE=1: the older two-barrier loop on one image)
    static const size_t need = (getenv("MPNHIP_WP_ONE_IMAGE") ? 1 : 2) * 3 * WP_KB * wp_pitch(320, 64);
    static const size_t ask = [] { const char* e = getenv("MPNHIP_WP_LDS"); const long x = e ? atol(e) : 0; return (size_t)(x > 0 ? x : 0); }();
    return ask > need ? ask : need;
}

}  // namespace

bool wp_eligible(const WpProduct& p) {
    auto al16 = [](const void* q) { return (((uintptr_t)q) & 15) == 0; };
    if (p.rows <= 0 || p.rows >= (int64_t)1 << 31 || p.nbatch < 1 || !p.dZ || !p.H || p.n_out < 1 || p.k_in < 1) return false;
    if (p.src16) {
        // bf16 source rows: H (and, except for the [1 x k] form whose dZ is the fp32 logit gradient, dZ) are unsigned shorts
        auto al8 = [](const void* q) { return (((uintptr_t)q) & 7) == 0; };
        if (p.pieces != 1 || p.H2 || p.h_idx) return false;
        if (p.n_out == 1 && p.k_in % 4 == 0 && p.k_in <= 64) return al8(p.H) && p.ldh % 4 == 0 && p.h_bstride % 4 == 0;
        if (p.dz_idx) return false;
        if (wp_is_small(p.n_out, p.k_in)) return p.n_out * (p.k_in + 1) <= 5 * WP_NT;
        return p.n_out % 4 == 0 && p.k_in % 4 == 0 && al8(p.dZ) && al8(p.H) && p.ldz % 4 == 0 && p.ldh % 4 == 0 && p.z_bstride % 4 == 0 &&
               p.h_bstride % 4 == 0;
    }
    if (p.n_out == 1 && p.k_in % 4 == 0 && p.k_in <= 64 && !p.h_idx && !p.H2 && al16(p.H) && p.ldh % 4 == 0 && p.h_bstride % 4 == 0)
        return true;   // wp_block_vec: gathered dZ allowed
    if (wp_is_small(p.n_out, p.k_in))   // wp_block_small: any alignment, gathers allowed
        return !p.H2 && p.n_out * (p.k_in + 1) <= 5 * WP_NT;
    return !p.dz_idx && !p.h_idx && p.n_out % 4 == 0 && p.k_in % 4 == 0 &&
           al16(p.dZ) && al16(p.H) && p.ldz % 4 == 0 && p.ldh % 4 == 0 && p.z_bstride % 4 == 0 && p.h_bstride % 4 == 0 &&
           (!p.H2 || (al16(p.H2) && p.ldh2 % 4 == 0 && p.h2_bstride % 4 == 0 && p.csplit % 4 == 0 && p.csplit > 0 && p.csplit < p.k_in));
}

size_t wp_slab_floats(int n_out, int k_in, int64_t rows, int nbatch, bool ranged, bool batched, bool src16) {
    int v, to, tc, chunk, nsplit;
    size_t need = 0;
    // (bf16 rows: which of the two kernels takes the product depends on the operands' alignment, known only when it is added --
    // reserve for the one that cuts more row chunks)
    for (int pass = 0; pass < (src16 ? 2 : 1); ++pass) {
        wp_choose(n_out, k_in, &v, &to, &tc, src16, pass == 1);
        wp_plan(ranged ? (rows + 1) / 2 : rows, rows, nbatch, to * tc, batched, &chunk, &nsplit, 256, wp_light(v, n_out, k_in));
        const size_t f = ((size_t)nsplit * nbatch * n_out * tn_kpad(k_in) + 63) / 64 * 64;
        need = f > need ? f : need;
    }
    return need;
}

void wp_batch_begin(WpBatch* b, float* slab, size_t slab_floats, bool batched) {
    b->batched = batched;
    b->tab.njobs = 0;
    b->slab = slab;
    b->slab_floats = slab_floats;
    b->used = 0;
    b->flops = 0.0;
    b->bytes = 0.0;
    b->nblocks = 0;
    b->nblocks2 = 0;
    b->bytes2 = 0.0;
    b->nred = 0;
    b->stream = nullptr;
    b->has_stream = false;
    g_wp = b;
}
void wp_batch_set_stream(hipStream_t s) {
    if (g_wp) { g_wp->stream = s; g_wp->has_stream = true; }
}
bool wp_batch_roll(int* status) {
    WpBatch* b = g_wp;
    *status = MPNHIP_OK;
    if (!b || !b->has_stream || b->tab.njobs == 0) return false;
    float* slab = b->slab;
    const size_t fl = b->slab_floats;
    const bool batched = b->batched;
    const hipStream_t s = b->stream;
    *status = wp_batch_flush(s);
    wp_batch_begin(b, slab, fl, batched);
    wp_batch_set_stream(s);
    return *status == MPNHIP_OK;
}
bool wp_batch_open() { return g_wp != nullptr; }
void wp_batch_abort() { g_wp = nullptr; }

static bool ranged_gather(const WpProduct& p) { return p.dz_idx || p.h_idx || p.H2; }

// records the n (1 or 2: the direction groups of one product) jobs, or none of them
bool wp_batch_add(const WpProduct* ps, int n) {
    WpBatch* b = g_wp;
    if (!b || b->tab.njobs + n > WP_MAX_JOBS) return false;
    size_t need = 0;
    for (int i = 0; i < n; ++i) {
        if (!wp_eligible(ps[i])) return false;
        need += wp_slab_floats(ps[i].n_out, ps[i].k_in, ps[i].rows, ps[i].nbatch, ps[i].row_begin || ps[i].row_end, b->batched, ps[i].src16 != 0);
    }
    if (b->used + need > b->slab_floats) {
        if (getenv("MPNHIP_WP_TRACE"))
            fprintf(stderr, "wp batch: job %d [%d x %d] rows %lld x%d needs %zu slab floats, %zu of %zu used\n", b->tab.njobs, ps[0].n_out, ps[0].k_in,
                    (long long)ps[0].rows, ps[0].nbatch, need, b->used, b->slab_floats);
        return false;
    }
    for (int i = 0; i < n; ++i) {
        const WpProduct& p = ps[i];
        const bool ranged = p.row_begin || p.row_end;
        WpJob& J = b->tab.job[b->tab.njobs];
        J = WpJob{};
        auto al16 = [](const void* q) { return (((uintptr_t)q) & 15) == 0; };
        const bool rows16 = p.src16 && !ranged_gather(p) && al16(p.dZ) && al16(p.H) && p.ldz % 8 == 0 && p.ldh % 8 == 0 && p.z_bstride % 8 == 0 &&
                            p.h_bstride % 8 == 0 && p.ldz < ((int64_t)1 << 28) && p.ldh < ((int64_t)1 << 28);
        wp_choose(p.n_out, p.k_in, &J.variant, &J.tiles_o, &J.tiles_c, p.src16 != 0, rows16);
        J.src16 = p.src16 ? 1 : 0;
        wp_plan(ranged ? (p.rows + 1) / 2 : p.rows, p.rows, p.nbatch, J.tiles_o * J.tiles_c, b->batched, &J.chunk, &J.nsplit, 256,
                wp_light(J.variant, p.n_out, p.k_in));
        J.dZ = p.dZ; J.H = p.H; J.ldz = p.ldz; J.ldh = p.ldh; J.z_bstride = p.z_bstride; J.h_bstride = p.h_bstride;
        J.H2 = p.H2; J.ldh2 = p.ldh2; J.h2_bstride = p.h2_bstride; J.csplit = p.H2 ? p.csplit : p.k_in;
        J.pieces = p.pieces == 1 ? 1 : 3;
        J.lds2 = getenv("MPNHIP_WP_ONE_IMAGE") ? 0 : 1;
        J.row_begin = p.row_begin; J.row_end = p.row_end; J.m_static = p.rows; J.dz_idx = p.dz_idx; J.h_idx = p.h_idx;
        J.slab = b->slab + b->used;
        J.grad_w = p.grad_w; J.ldw = p.ldw; J.grad_b = p.grad_b;
        J.n_out = p.n_out; J.k_in = p.k_in; J.nbatch = p.nbatch;
        if (J.variant >= 16) {   // wgrad_rows16.hip's launch
            J.block0 = b->nblocks2;
            b->nblocks2 += J.tiles_o * J.tiles_c * J.nsplit * J.nbatch;
            b->bytes2 += 2.0 * (ranged ? p.rows / 2.0 : (double)p.rows) * p.nbatch * (p.n_out + p.k_in);
            count_path(PC_TN_ROWS16);
        } else {
            J.block0 = b->nblocks;
            b->nblocks += J.tiles_o * J.tiles_c * J.nsplit * J.nbatch;
        }
        J.red_block0 = b->nred;
        b->nred += (int)(((int64_t)p.n_out * tn_kpad(p.k_in) / 4 + 31) / 32);
        b->used += wp_slab_floats(p.n_out, p.k_in, p.rows, p.nbatch, ranged, b->batched, p.src16 != 0);
        b->flops += 2.0 * (ranged ? p.rows / 2.0 : (double)p.rows) * p.nbatch * p.n_out * p.k_in;
        b->bytes += (p.src16 ? 2.0 : 4.0) * (ranged ? p.rows / 2.0 : (double)p.rows) * p.nbatch * (p.n_out + p.k_in);
        ++b->tab.njobs;
        count_path(PC_TN_PANEL);
    }
    return true;
}
bool wp_batch_add(const WpProduct& p) { return wp_batch_add(&p, 1); }

int wp_batch_flush(hipStream_t s) {
    WpBatch* b = g_wp;
    g_wp = nullptr;
    if (!b || b->tab.njobs == 0) return MPNHIP_OK;
    static const bool attr_set = [] {
        return hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_panel_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)wp_lds_bytes()) == hipSuccess;
    }();
    (void)attr_set;
    count_path(PC_TN_PANEL_LAUNCH);
    if (const char* e = getenv("MPNHIP_WP_DEBUG")) b->tab.debug = atoi(e); else b->tab.debug = 0;
    static const bool trace = getenv("MPNHIP_WP_TRACE") != nullptr;   // diagnosis: the jobs of every flush, one line each
    if (trace) {
        for (int i = 0; i < b->tab.njobs; ++i) {
            const WpJob& J = b->tab.job[i];
            fprintf(stderr, "wp job %2d/%d: rows %lld x%d  [%d x %d]  variant %d tiles %dx%d chunk %d nsplit %d  %s pieces %d%s%s\n", i, b->tab.njobs,
                    (long long)J.m_static, J.nbatch, J.n_out, J.k_in, J.variant, J.tiles_o, J.tiles_c, J.chunk, J.nsplit, J.src16 ? "bf16 rows" : "fp32 rows",
                    J.pieces, (J.row_begin || J.row_end) ? " ranged" : "", (J.dz_idx || J.h_idx) ? " gathered" : "");
        }
    }
    // Blocks are dispatched in index order: the jobs with the longest blocks get the lowest indices, so that a launch ends on its short
    // blocks (longest-processing-time-first; the order the products were recorded in put the widest -- edge layer 0 / 1, the per-node
    // projections -- last).  A block's length ~ its chunk's rows x the columns it stages per row.  MPNHIP_WP_NO_LPT=1: recording order.
    // A launch of narrow jobs only (the reference's own widths, d = 32): every MFMA job re-tiled to <1, 1> (64 x 64) tiles -- slabs and
    // chunks are untouched by the tiling -- and run by the four-blocks-per-CU instantiation.  Taken when the re-tiling re-reads
    // at most 15 % more operand columns over the whole launch (a [80 x 32] product as 2 x 1 tiles reads its 32 H columns twice).
    static const bool narrow_on = !getenv("MPNHIP_WP_NO_NARROW");
    bool narrow = narrow_on && b->nblocks > 0;
    const int narrow_lds = WP_NARROW_LDS;
    {
        double cols = 0.0, extra = 0.0;
        for (int i = 0; i < b->tab.njobs && narrow; ++i) {
            const WpJob& J = b->tab.job[i];
            if (J.variant >= 16) continue;
            if (J.src16 || J.variant > 7 || J.variant == 4) { narrow = false; break; }
            const double rows = (double)J.m_static * J.nbatch * ((J.row_begin || J.row_end) ? 0.5 : 1.0);
            if (J.variant == 6 || J.variant == 7) { cols += rows * (J.n_out + J.k_in); continue; }
            const int to = (J.n_out + 63) / 64, tc = (J.k_in + 63) / 64;
            cols += rows * ((double)J.n_out * J.tiles_c + (double)J.k_in * J.tiles_o);
            extra += rows * ((double)J.n_out * tc + (double)J.k_in * to) - rows * ((double)J.n_out * J.tiles_c + (double)J.k_in * J.tiles_o);
        }
        if (narrow && extra > 0.15 * cols) narrow = false;
        if (narrow) {
            for (int i = 0; i < b->tab.njobs; ++i) {
                WpJob& J = b->tab.job[i];
                if (J.variant >= 16 || J.variant == 6 || J.variant == 7) continue;
                J.variant = 3;
                J.tiles_o = (J.n_out + 63) / 64;
                J.tiles_c = (J.k_in + 63) / 64;
            }
            count_path(PC_TN_PANEL_NARROW);
        }
    }
    static const bool lpt = !getenv("MPNHIP_WP_NO_LPT");
    {   // (block counts follow the tiling: recomputed here, in recording order unless re-indexed below)
        int n1 = 0, n2 = 0;
        for (int i = 0; i < b->tab.njobs; ++i) {
            WpJob& J = b->tab.job[i];
            const int nb = J.tiles_o * J.tiles_c * J.nsplit * J.nbatch;
            if (J.variant >= 16) { J.block0 = n2; n2 += nb; } else { J.block0 = n1; n1 += nb; }
        }
        b->nblocks = n1;
        b->nblocks2 = n2;
    }
    if (lpt) {
        for (int pass = 0; pass < 2; ++pass) {   // the row-panel kernel's jobs, then wgrad_rows16.hip's: each launch has its own indices
            int idx[WP_MAX_JOBS], n = 0;
            double w[WP_MAX_JOBS];
            for (int i = 0; i < b->tab.njobs; ++i) {
                const WpJob& J = b->tab.job[i];
                if ((J.variant >= 16) != (pass == 1)) continue;
                const double rows = (J.row_begin || J.row_end) ? 0.5 * J.chunk : (double)J.chunk;
                w[n] = rows * ((double)J.n_out / J.tiles_o + (double)J.k_in / J.tiles_c);
                idx[n++] = i;
            }
            for (int a = 1; a < n; ++a)   // (insertion sort, stable: equal weights keep the recording order)
                for (int c = a; c > 0 && w[c] > w[c - 1]; --c) { std::swap(w[c], w[c - 1]); std::swap(idx[c], idx[c - 1]); }
            int next = 0;
            for (int a = 0; a < n; ++a) {
                WpJob& J = b->tab.job[idx[a]];
                J.block0 = next;
                next += J.tiles_o * J.tiles_c * J.nsplit * J.nbatch;
            }
        }
    }
    if (b->nblocks2 > 0) {
        // the bf16-row jobs of the LDS-DMA kernel (wgrad_rows16.hip): the profiled launch of a batch that has them (their bytes)
        count_path(PC_TN_ROWS16_LAUNCH);
        prof_begin(PROF_TN, s, b->bytes2);
        const int r2 = launch_wgrad_rows16(b->tab, b->nblocks2, s);
        prof_end(PROF_TN, s);
        MPN_TRY(r2);
        if (b->nblocks > 0) {
            if (narrow) hipLaunchKernelGGL(wgrad_panel_narrow_kernel, dim3((unsigned)b->nblocks), dim3(WP_NT), narrow_lds, s, b->tab);
            else hipLaunchKernelGGL(wgrad_panel_kernel, dim3((unsigned)b->nblocks), dim3(WP_NT), wp_lds_bytes(), s, b->tab);
        }
        MPN_LAUNCH_CHECK();
    } else if (b->nblocks > 0) {
        prof_begin(PROF_TN, s, b->bytes);   // (HBM-bound by design: the hook's work figure is the launch's operand bytes)
        {
            hipEvent_t e0, e1;
            const bool ev = prof_launch_events(&e0, &e1);
            if (narrow && ev) hipExtLaunchKernelGGL(wgrad_panel_narrow_kernel, dim3((unsigned)b->nblocks), dim3(WP_NT), narrow_lds, s, e0, e1, 0, b->tab);
            else if (narrow) hipLaunchKernelGGL(wgrad_panel_narrow_kernel, dim3((unsigned)b->nblocks), dim3(WP_NT), narrow_lds, s, b->tab);
            else if (ev) hipExtLaunchKernelGGL(wgrad_panel_kernel, dim3((unsigned)b->nblocks), dim3(WP_NT), wp_lds_bytes(), s, e0, e1, 0, b->tab);
            else hipLaunchKernelGGL(wgrad_panel_kernel, dim3((unsigned)b->nblocks), dim3(WP_NT), wp_lds_bytes(), s, b->tab);
        }
        prof_end(PROF_TN, s);
        MPN_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)b->nred), dim3(256), 0, s, b->tab);
    MPN_LAUNCH_CHECK();
    return MPNHIP_OK;
}

}  // namespace mpnhip
