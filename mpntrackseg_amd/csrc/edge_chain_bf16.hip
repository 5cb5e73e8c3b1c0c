// Fused per-edge chain of one message-passing step with bf16 OPERANDS (mpnhip_model.precision == MPNHIP_PREC_BF16,
// BASELINE.json's "bf16 MLP GEMMs on MFMA" configuration, inference): the same modules as edge_chain.hip --
//   EdgeModel   (reference models/mpn.py:67-69)   H1 = relu(W1e [e0|e] + Pr[row] + Pc[col]);  e' = relu(W2 H1 + b2)
//   classifier  (mpn.py:377 -> :114)              logit = wc2 . relu(Wc1 e' + bc1) + bc2
//   flow MLPs   (mpn.py:85-94, per direction)     M = relu(Wf2 relu(Wfe e' + Pf[col]) + bf2)
// -- with every Linear product's operands (activations AND weights) rounded to bfloat16 (RNE) as they enter the product,
// fp32 accumulation, fp32 biases and fp32 gather-adds: the arithmetic of the unfused bf16 GEMM path (gemm.hip) and of
// oracle/mpn_oracle.py's precision("bf16").
//
// Why a second kernel and not a third template flavour of edge_chain.hip: that kernel keeps the whole hidden layer of a
// 32-edge wave tile in accumulators (T1 x 16 registers), which ends at 128-d (he 320).  At BASELINE.json's configs[4]
// widths (256-d: he 640, de 128, hn 448, dn 256, hc 64) the hidden layers are N-TILED here instead: a 32-feature tile of
// H1 is accumulated (v_mfma_f32_32x32x16_bf16, transposed formulation: features in the accumulator registers, the wave's 32
// edges on the lanes), finished (+ gathered projections, ReLU), rounded to bf16 -- its 16 registers ARE the B operand of two
// k blocks of the next layer (the weight images are packed in that contraction order) -- and multiplied straight into the
// e' accumulators; the same for HF -> M.  No hidden activation ever exists outside registers, and the register need is
// independent of the hidden widths: X (first-layer input, bf16) + output accumulators + one tile.
//
// A block is 8 waves = 256 edges of one direction group (2 waves per SIMD): twice the edges per weight byte of the fp32
// kernel, because at these widths the weight stream (0.7 MB per block, L2 -> LDS by LDS-DMA) is what a block moves most of
// -- or, at 256-d since round 3, 4 waves = 128 edges with ONE hidden tile per chunk (51 KB of LDS: two independent blocks per CU,
// whose serial phases -- first-layer input rows at the start, aggregation at the end -- run under each other's MFMA phases).
// Weight images ("pair" images, pack_pair_bf16): one SECTION per hidden tile t = [first-layer units of tile t (all k blocks)
// | second-layer units of the two k blocks the tile feeds (x all output tiles)], units of 1 KiB = one MFMA A operand
// (64 lanes x 8 bf16; element i of lane (m, g) = W[n0 + m][k0 + (i & 3) + 8 (i >> 2) + 4 g], the accumulator layout's
// order).  A chunk = the sections of two hidden tiles, contiguous in the image, double buffered, one barrier per chunk.
#include <cstdio>
#include <cstdlib>

#include "common.h"
#include "edge_chain.h"
#include "edge_chain_bf16_common.h"

#ifndef MPNHIP_ROWSTORE_AB
#define MPNHIP_ROWSTORE_AB 1   // 0: compile the A-B switches of the row stores out
#endif

namespace mpnhip {

// Debug build (make EXTRA=-DMPNHIP_CHAIN_TS): lane 0 of every wave stamps s_memtime at the phase boundaries and after every hidden
// tile; with MPNHIP_CHAIN_TS=<file prefix> in the environment the 20th launch dumps its stamps (tools/chain_stamps.py bf16)
#ifdef MPNHIP_CHAIN_TS
#define TS16_INIT() long long* tsp = A.ts ? A.ts + ((int64_t)blockIdx.x * NW + wave) * 48 : nullptr
#define TS16(i) do { if (tsp && lane == 0) tsp[i] = clock64(); } while (0)
#else
#define TS16_INIT() do {} while (0)
#define TS16(i) do {} while (0)
#endif

// T1 = ceil(he/32), T2 = ceil(de/32), TF = ceil(hn/32), TD = ceil(dn/32), TC = ceil(hc/32); EF = 1: first-layer input e,
// 2: [e0 | e] (each half padded to 32 T2 columns in the image).
// NW waves per block (8: 256 edges, one block per CU; 4: 128 edges, two independent blocks per CU whose serial phases -- the
// first-layer input rows at the start, the aggregation at the end -- run under each other's MFMA phases, for twice the weight
// stream), CTI hidden tiles per weight chunk.
// SAVE (training): the hidden activations H1 / HC / HF and a bf16 copy of e' are written row-major as bf16 [E, width] (every consumer
// -- the weight-gradient products -- rounds them to bf16 anyway: exact for this mode), and every ReLU decision (H1, e', HC, HF, M)
// as one bit in a lane-private layout the backward chain kernel (edge_chain_bf16_bwd.hip) reads back with the same (block, wave,
// lane) -> edge mapping: word w of wave tile wt at save_mask[(wt * NWORDS + w) * 64 + lane] (chain_bf16_mask_words()).
// GD (round 5; the 256-d four-wave form): the COL-side gathered C-in -- Pc[col] of an H1 tile, Pf[col] of an HF tile: 128 bytes of a random
// table row per edge -- lands by LDS-DMA in a per-wave 4 KB patch, 8 lanes x 16 bytes per row = eight whole 128-byte lines per
// instruction, instead of 32-byte pieces of 32 different rows per instruction into registers (the line touches of those four
// loads per tile were 16 % of the launch: with the col-side gathers pointed at the sorted ROW's table row, 513 -> 430 us).
template <int T1, int T2, int TF, int TD, int TC, int EF, bool EXACT, int NW = 8, int CTI = 2, bool SAVE = false, int DEP = DEPTH, bool GD = false>
__global__ __launch_bounds__(64 * NW, NW == 8 ? 1 : 2) void edge_chain_bf16_kernel(EdgeChainBf16Args A) {
    constexpr int EPB = 32 * NW;           // edges per block
    constexpr int DE = 32 * T2, DN = 32 * TD, HC = 32 * TC;
    constexpr int KBE = 2 * T2;            // k blocks of e'
    constexpr int KB1 = KBE * EF;          // k blocks of the first-layer input
    constexpr int SEC1 = KB1 + 2 * T2;     // units (KiB) per H1 tile section
    constexpr int SECF = KBE + 2 * TD;     // per HF tile section
    constexpr int SECC = KBE;              // per HC tile section (classifier layer 0 only; its layer 1 is a dot product)
    constexpr int CT = CTI;                // hidden tiles per chunk
    constexpr int NCH1 = (T1 + CT - 1) / CT, NCHF = (TF + CT - 1) / CT;
    constexpr int CHU = bmax(bmax(CT * SEC1, CT * SECF), TC * SECC);   // KiB per chunk buffer
    // one LDS object: two chunk buffers, then the biases [b2 (DE) | bc1 (HC) | wc2 (HC) | bf2 (DN)], zero-padded
    // (the fused aggregation's per-wave slabs reuse the chunk buffers after the last chunk: 8 x [32 edges][32 RT + 1] floats)
    constexpr int AGG_RT = TD >= 2 ? 2 : 1, AGG_BYTES = NW * (32 * AGG_RT) * 36 * 4;
    constexpr int WB_BYTES = bmax(2 * CHU * 1024, (AGG_BYTES + 15) / 16 * 16);
    static_assert(!GD || (EXACT && !SAVE && ROW_SLAB_BYTES >= 4096), "the DMA gathers fetch whole 128-byte pieces into the row slabs");
    __shared__ __attribute__((aligned(16))) char smem[WB_BYTES + (DE + 2 * HC + DN) * 4];
    // per-wave slabs of the full-line row stores (RowStage): a separate object, never the target of an LDS-DMA
    __shared__ __attribute__((aligned(16))) char rowslab[NW * ROW_SLAB_BYTES];
    float* const sbias = reinterpret_cast<float*>(smem + WB_BYTES);
#define WBUF(i) (smem + ((i) & 1) * (CHU * 1024))

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lj = lane & 31, lh = lane >> 5;
    const int he = A.he, de = A.de, hn = A.hn, dn = A.dn, hc = A.hc;

    // ---- which direction group / which 256 edges ------------------------------------------------------
    const int e_out = A.header[1], e_in = A.header[2];
    int grp, beg, end, blk = blockIdx.x;
    {
        const int nb0 = (e_out + EPB - 1) / EPB, nb1 = (e_in + EPB - 1) / EPB;
        if (blk < nb0) { grp = 0; beg = 0; end = e_out; }
        else if (blk < nb0 + nb1) { grp = 1; blk -= nb0; beg = e_out; end = e_out + e_in; }
        else { grp = 2; blk -= nb0 + nb1; beg = e_out + e_in; end = A.E; }
    }
    const int tile0 = beg + blk * EPB;
    if (tile0 >= end) return;
    const int edge_raw = tile0 + wave * 32 + lj;
    const bool edge_ok = edge_raw < end;
    const int edge = edge_ok ? edge_raw : end - 1;
    const bool flow = grp < 2;
    const char* const img1 = static_cast<const char*>(A.img_edge);
    const char* const imgf = static_cast<const char*>(grp == 1 ? A.img_flow[1] : A.img_flow[0]);

    TS16_INIT();
    TS16(0);
    chunk_fetch<bmin(CT, T1) * SEC1, NW>(img1, WBUF(0), wave, lane);
    {
        const float* bf2 = grp == 1 ? A.bf2_in : A.bf2_out;
        for (int i = tid; i < DE + 2 * HC + DN; i += 64 * NW) {
            float v = 0.f;
            if (i < DE) v = i < de ? A.b2[i] : 0.f;
            else if (i < DE + HC) v = i - DE < hc ? A.bc1[i - DE] : 0.f;
            else if (i < DE + 2 * HC) v = i - DE - HC < hc ? A.wc2[i - DE - HC] : 0.f;
            else v = i - DE - 2 * HC < dn ? bf2[i - DE - 2 * HC] : 0.f;
            sbias[i] = v;
        }
    }
    // (the edge's end points first: the C-in gathers that depend on them then go out under the input rows' loads)
    const int row = A.srow[edge], col = A.scol[edge];
    const unsigned pro = (unsigned)row * (unsigned)A.pw;
    // (timing ablation, MPNHIP_CHAIN_BF16_DEBUG_SKIP bit 16: the col-side gathers take the ROW's table row -- sorted edges, a few
    // distinct lines per wave instruction instead of 32; results wrong)
    const unsigned pco = (unsigned)((A.debug_skip & 16) ? row : col) * (unsigned)A.pw + (unsigned)he;
    const unsigned pfo = pco + (unsigned)(he + (grp == 1 ? hn : 0));
    // GD: lane l of DMA instruction q fetches chunk (l & 7) ^ swizzle of edge 8 q + (l >> 3)'s row; the patch image is [edge][8 chunks],
    // chunk positions XOR-ed with (edge >> 1) & 7 (through the SOURCE address: the DMA destination is lane-linear), which makes the
    // read-back -- lane (lj, lh) takes chunks 2 g + lh of edge lj -- conflict-free for the ds_read_b128 lane groups
    unsigned gco[4];
    unsigned gp_wr = 0, gp_rd[4] = {0, 0, 0, 0};
    if constexpr (GD) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int eq_raw = tile0 + wave * 32 + 8 * q + (lane >> 3);
            const int eq = eq_raw < end ? eq_raw : end - 1;
            const int cq = (A.debug_skip & 16) ? A.srow[eq] : A.scol[eq];
            const int rq = 8 * q + (lane >> 3);
            gco[q] = (unsigned)cq * (unsigned)A.pw + (unsigned)he + (unsigned)(((lane & 7) ^ ((rq >> 1) & 7)) * 4);
        }
        gp_wr = lds_addr(rowslab + wave * ROW_SLAB_BYTES);
#pragma unroll
        for (int g = 0; g < 4; ++g) gp_rd[g] = gp_wr + (unsigned)(lj * 128 + (((2 * g + lh) ^ ((lj >> 1) & 7)) * 16));
    }
    // (the per-wave gather patch [32 edges][128 B] IS the wave's row slab: inference uses the slab only for the e' rows between the
    // H1 and the HF tiles, when no gather is in flight -- the first Pf gather goes out after those stores; the SAVE variant stores
    // rows through its slab in every tile, and a separate 16 KB of patches would cost the second block per CU: it keeps the register form)
    char* const gp_ptr = rowslab + wave * ROW_SLAB_BYTES;
    // the four row pieces of the patch as this lane's C-in (waits for the wave's own DMA first: nothing else orders them)
    auto gd_take = [&](float4& v0, float4& v1, float4& v2, float4& v3) {
        typedef float gdx4 __attribute__((ext_vector_type(4)));   // (a native vector: HIP's float4 struct is no "+v" operand)
        gdx4 a, b, c4, d;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("ds_read_b128 %0, %1" : "=v"(a) : "v"(gp_rd[0]) : "memory");
        asm volatile("ds_read_b128 %0, %1" : "=v"(b) : "v"(gp_rd[1]) : "memory");
        asm volatile("ds_read_b128 %0, %1" : "=v"(c4) : "v"(gp_rd[2]) : "memory");
        asm volatile("ds_read_b128 %0, %1" : "=v"(d) : "v"(gp_rd[3]) : "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c4), "+v"(d)::"memory");
        v0 = make_float4(a[0], a[1], a[2], a[3]); v1 = make_float4(b[0], b[1], b[2], b[3]);
        v2 = make_float4(c4[0], c4[1], c4[2], c4[3]); v3 = make_float4(d[0], d[1], d[2], d[3]);
    };
    auto gd_issue = [&](unsigned extra) {   // extra: float offset past the Pc block (0: Pc tile t -> 32 t; the Pf blocks: + he + ...)
#pragma unroll
        for (int q = 0; q < 4; ++q)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(A.P + gco[q] + extra),
                                             (__attribute__((address_space(3))) void*)(gp_ptr + q * 1024), 16, 0, 0);
    };
    // training saves: mask words of this wave tile (sections H1 | e' | HC | HF | M, two tiles per word)
    constexpr int WB_E = (T1 + 1) / 2, WB_C = WB_E + (T2 + 1) / 2, WB_F = WB_C + (TC + 1) / 2, WB_M = WB_F + (TF + 1) / 2;
    constexpr int NWORDS = WB_M + (TD + 1) / 2;
    unsigned mw = 0;
    const unsigned ones = 0x00010001u;
    unsigned* const mask_wt = SAVE ? A.save_mask + ((size_t)(blockIdx.x * NW + wave) * NWORDS) * 64 + lane : nullptr;
    auto mask_put = [&](int wbase, int t, int T, unsigned bits) {
        mw = (t & 1) ? (mw | (bits << 8)) : bits;
        if (((t & 1) || t + 1 == T) && !(A.debug_skip & 2)) mask_wt[(size_t)(wbase + (t >> 1)) * 64] = mw;
    };
    RowStage rs;
    rs.init(rowslab + wave * ROW_SLAB_BYTES, lane, tile0 + wave * 32, end);
    // a finished bf16 tile t of T to its rows [E, width]: tiles go out in pairs, 128 bytes per row = whole lines (RowStage); a lone
    // last tile straight from the registers (32 contiguous bytes per lane: tile_rows16).  NT: non-temporal -- rows that are read
    // again only in the backward pass, a whole forward later (984 -> 935 us per launch at cfg-E; MPNHIP_CHAIN_BF16_DEBUG_SKIP=4: plain)
    auto save_tile = [&](unsigned short* base, int width, int t, int T, const bf16x8& h0, const bf16x8& h1, bool nt_ok) {
        if (A.debug_skip & 1) return;
        uint4 lo, hi;
        tile_rows16(h0, h1, lo, hi);
        const bool nt = nt_ok && !(A.debug_skip & 4);
        if ((!(t & 1) && t + 1 == T) || (MPNHIP_ROWSTORE_AB && (A.debug_skip & 8))) {   // (8: A-B, every tile straight from the registers)
            const int f = 32 * t + 16 * lh;
            unsigned short* q = base + (size_t)edge * width + f;
            if (edge_ok && (EXACT || f < width)) *reinterpret_cast<uint4*>(q) = lo;
            if (edge_ok && (EXACT || f + 8 < width)) *reinterpret_cast<uint4*>(q + 8) = hi;
            return;
        }
        rs.put16(t & 1, lo, hi);
        if (t & 1) {
            if (nt) rs.flush<true>(reinterpret_cast<char*>(base), (size_t)width * 2, 64 * (t - 1), (width - 32 * (t - 1)) * 2);
            else rs.flush<false>(reinterpret_cast<char*>(base), (size_t)width * 2, 64 * (t - 1), (width - 32 * (t - 1)) * 2);
        }
    };
    // ---- first-layer input: this lane's edge row(s), k = 16 kb + 8 h + (0..7) per k block (natural order: pack_chain_bf16) ------
    bf16x8 X[KB1];
#pragma unroll
    for (int sg = 0; sg < EF; ++sg) {
        const unsigned short* x16 = sg == 0 ? A.xa16 : A.xb16;
        if (x16) {
            // bf16 rows: the same values the fp32 path rounds to, as two 8-byte pieces per k block
            const unsigned short* xr = x16 + (size_t)edge * de;
#pragma unroll
            for (int kb = 0; kb < KBE; ++kb) {
                // k = 16 kb + 8 lh + (0..7): the first-layer units are packed in this (natural) order -- one 16-byte load
                const int n0 = 16 * kb + 8 * lh;
                uint4 a = *reinterpret_cast<const uint4*>(xr + (EXACT || n0 < de ? n0 : 0));
                if (!EXACT && n0 >= de) a = make_uint4(0u, 0u, 0u, 0u);
                const u32x4 v = {a.x, a.y, a.z, a.w};
                X[sg * KBE + kb] = __builtin_bit_cast(bf16x8, v);
            }
        } else {
            const float* xr = sg == 0 ? A.xa + (int64_t)edge * A.ldxa : A.xb + (int64_t)edge * A.ldxb;
#pragma unroll
            for (int kb = 0; kb < KBE; ++kb)
                X[sg * KBE + kb] = pack8(ldrow<EXACT>(xr, 0u, 16 * kb + 8 * lh, de), ldrow<EXACT>(xr, 0u, 16 * kb + 8 * lh + 4, de));
        }
    }
    // gathered C-in of one H1 tile: Pr[row] and Pc[col], 4 row pieces each; fetched one tile ahead
    // (CIN_LOADS / PF_LOADS: the vector-memory operations cin_issue / pf_issue put behind a chunk's LDS-DMA -- what the counted
    // chunk barriers below leave in flight; change the loops and these together)
    constexpr int CIN_LOADS = 8, PF_LOADS = 4;
    float4 cin[CIN_LOADS];
    static_assert(sizeof(cin) / sizeof(cin[0]) == 2 * 4 && CIN_LOADS == 2 * 4, "cin_issue issues 2 x 4 row loads");
    auto cin_issue = [&](int t) {
        if constexpr (GD) {
#pragma unroll
            for (int g = 0; g < 4; ++g) cin[g] = ldrow<EXACT>(A.P, pro, 32 * t + 8 * g + 4 * lh, he);
            gd_issue((unsigned)(32 * t));          // (4 register loads + 4 DMA pieces = CIN_LOADS vector-memory operations)
        } else {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                cin[g] = ldrow<EXACT>(A.P, pro, 32 * t + 8 * g + 4 * lh, he);
                cin[4 + g] = ldrow<EXACT>(A.P, pco, 32 * t + 8 * g + 4 * lh, he);
            }
        }
    };
    float4 pf[PF_LOADS];
    static_assert(sizeof(pf) / sizeof(pf[0]) == 4 && PF_LOADS == 4, "pf_issue issues 4 row loads");
    auto pf_issue = [&](int t) {
        if constexpr (GD) {
            gd_issue((unsigned)(he + (grp == 1 ? hn : 0) + 32 * t));   // (PF_LOADS DMA pieces)
        } else {
#pragma unroll
            for (int g = 0; g < 4; ++g) pf[g] = ldrow<EXACT>(A.P, pfo, 32 * t + 8 * g + 4 * lh, hn);
        }
    };
    cin_issue(0);
    f32x16 en[T2];
#pragma unroll
    for (int o = 0; o < T2; ++o)
#pragma unroll
        for (int r = 0; r < 16; ++r) en[o][r] = 0.f;
    __syncthreads();   // chunk 0 and the biases are in LDS
    TS16(1);

    // ---- phases 1 + 2: per H1 tile  H1_t = relu(W1e_t X + Pr + Pc)  ->  e' += W2[:, tile t] H1_t -----------------
    int c = 0;
#pragma unroll
    for (int ch = 0; ch < NCH1; ++ch) {
        const int nt = bmin(CT, T1 - ch * CT);
        // the next chunk's DMA goes out AFTER the first tile has taken its gathered C-in (the compiler's wait for those loads is
        // vmcnt(0): issued earlier, the DMA would be waited for right there, its whole L2 latency exposed once per chunk)
        auto fetch_next = [&]() {
            if (ch + 1 < NCH1) {
                if (T1 - (ch + 1) * CT >= CT) chunk_fetch<CT * SEC1, NW>(img1 + (size_t)(ch + 1) * CT * SEC1 * 1024, WBUF(c + 1), wave, lane);
                else chunk_fetch<(T1 % CT ? T1 % CT : CT) * SEC1, NW>(img1 + (size_t)(ch + 1) * CT * SEC1 * 1024, WBUF(c + 1), wave, lane);
            } else {
                chunk_fetch<TC * SECC, NW>(static_cast<const char*>(A.img_cls), WBUF(c + 1), wave, lane);
            }
            pin_order();
        };
#pragma unroll
        for (int tt = 0; tt < CT; ++tt) {
            if (tt < nt) {
                const int t = ch * CT + tt;
                f32x16 acc;
                if constexpr (GD) gd_take(cin[4], cin[5], cin[6], cin[7]);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    acc[4 * g + 0] = cin[g].x + cin[4 + g].x; acc[4 * g + 1] = cin[g].y + cin[4 + g].y;
                    acc[4 * g + 2] = cin[g].z + cin[4 + g].z; acc[4 * g + 3] = cin[g].w + cin[4 + g].w;
                }
                if (tt == 0) { asm volatile("" : "+v"(acc)::"memory"); fetch_next(); }   // (acc is formed before the DMA goes out)
                if (t + 1 < T1) cin_issue(t + 1);
                else if (flow && !GD) pf_issue(0);
                auto fin = [&](const bf16x8& h0, const bf16x8& h1) {
                    if constexpr (SAVE) {
                        save_tile(A.save_h1, he, t, T1, h0, h1, true);
                        mask_put(0, t, T1, tile_mask_bits(h0, h1, ones));
                    }
                };
                auto act = [](f32x16& v) { relu16(v); };
                if (tt == 0) hidden_tile<0, KB1, T2, DEP>(lds_addr(WBUF(c)) + lane * 16, X, acc, en, act, fin);
                else hidden_tile<SEC1 * 1024, KB1, T2, DEP>(lds_addr(WBUF(c)) + lane * 16, X, acc, en, act, fin);
            }
        }
        // (the chunk's last tile issued the 8 gathers of the tile after it -- except the very last one: 4 or none, drain)
        if (ch * CT + nt < T1) chunk_barrier<CIN_LOADS>(A.plain_barriers != 0);
        else __syncthreads();
        TS16(2 + (ch < 20 ? ch : 19));
        ++c;
    }
    // ---- e' = relu(. + b2): out, and as the B operand of the classifier and the flow MLPs -------------------------------
    bf16x8 eb[KBE];
    {
#pragma unroll
        for (int o = 0; o < T2; ++o) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 b = *reinterpret_cast<const float4*>(sbias + 32 * o + 8 * g + 4 * lh);
                en[o][4 * g + 0] += b.x; en[o][4 * g + 1] += b.y; en[o][4 * g + 2] += b.z; en[o][4 * g + 3] += b.w;
            }
            relu16(en[o]);
            if (A.e_new) {   // (an fp32 tile is 128 bytes per row: whole lines through the slab)
                rs.put32(en[o]);
                rs.flush<false>(reinterpret_cast<char*>(A.e_new), (size_t)de * 4, 128 * o, (de - 32 * o) * 4);
            }
            eb[2 * o] = pack_regs(en[o], 0);
            eb[2 * o + 1] = pack_regs(en[o], 1);
        }
        // (a second loop: a bf16 tile waits in the slab for its partner, and an fp32 tile in between would overwrite it)
#pragma unroll
        for (int o = 0; o < T2; ++o) {
            if (A.e16_out) save_tile(A.e16_out, de, o, T2, eb[2 * o], eb[2 * o + 1], false);   // (plain stores: the next step reads these rows)
            if constexpr (SAVE) mask_put(WB_E, o, T2, tile_mask_bits(eb[2 * o], eb[2 * o + 1], ones));
        }
    }
    if constexpr (GD) { if (flow) pf_issue(0); }   // (after the e' rows left through the slab the patch shares)
    TS16(22);
    // ---- phase 3: classifier (its image is in the current buffer) -------------------------------------------------------
    if (flow) chunk_fetch<bmin(CT, TF) * SECF, NW>(imgf, WBUF(c + 1), wave, lane);
    {
        float part = 0.f;
#pragma unroll
        for (int q = 0; q < TC; ++q) {
            f32x16 acc;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 b = *reinterpret_cast<const float4*>(sbias + DE + 32 * q + 8 * g + 4 * lh);
                acc[4 * g + 0] = b.x; acc[4 * g + 1] = b.y; acc[4 * g + 2] = b.z; acc[4 * g + 3] = b.w;
            }
            const unsigned wa = lds_addr(WBUF(c)) + lane * 16;
            auto cls_tile = [&](auto Q) {
                stream_units<Q.value * SECC * 1024, KBE, DEP>(wa, [&](auto U, const bf16x8& a) {
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, eb[U.value], acc, 0, 0, 0);
                });
            };
            if (q == 0) cls_tile(std::integral_constant<int, 0>{});
            else cls_tile(std::integral_constant<int, (TC > 1 ? 1 : 0)>{});
            static_assert(TC <= 2, "classifier hidden width up to 64");
            if constexpr (SAVE) {
                relu16(acc);
                const bf16x8 h0 = pack_regs(acc, 0), h1 = pack_regs(acc, 1);
                save_tile(A.save_hc, hc, q, TC, h0, h1, true);
                mask_put(WB_C, q, TC, tile_mask_bits(h0, h1, ones));
            }
            // layer 1 (out dim 1): the same operand rounding, fp32 accumulation over this lane's 16 features
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 w = *reinterpret_cast<const float4*>(sbias + DE + HC + 32 * q + 8 * g + 4 * lh);
                part = fmaf((float)(__bf16)w.x, (float)(__bf16)fmaxf(acc[4 * g + 0], 0.f), part);
                part = fmaf((float)(__bf16)w.y, (float)(__bf16)fmaxf(acc[4 * g + 1], 0.f), part);
                part = fmaf((float)(__bf16)w.z, (float)(__bf16)fmaxf(acc[4 * g + 2], 0.f), part);
                part = fmaf((float)(__bf16)w.w, (float)(__bf16)fmaxf(acc[4 * g + 3], 0.f), part);
            }
        }
        const float other = __shfl_xor(part, 32, 64);
        if (A.logits && edge_ok && lh == 0) A.logits[A.perm[edge]] = part + other + A.bc2[0];
    }
    if (!flow) return;  // self loops take part in the edge update only (mpn.py:85,91)
    __syncthreads();
    ++c;
    TS16(23);

    // ---- phases 4 + 5: per HF tile  HF_t = relu(Wfe_t e' + Pf[col])  ->  M += Wf2[:, tile t] HF_t ----------------------
    f32x16 mm[TD];
#pragma unroll
    for (int o = 0; o < TD; ++o)
#pragma unroll
        for (int r = 0; r < 16; ++r) mm[o][r] = 0.f;
#pragma unroll
    for (int ch = 0; ch < NCHF; ++ch) {
        const int nt = bmin(CT, TF - ch * CT);
        auto fetch_next = [&]() {
            if (ch + 1 < NCHF) {
                if (TF - (ch + 1) * CT >= CT) chunk_fetch<CT * SECF, NW>(imgf + (size_t)(ch + 1) * CT * SECF * 1024, WBUF(c + 1), wave, lane);
                else chunk_fetch<(TF % CT ? TF % CT : CT) * SECF, NW>(imgf + (size_t)(ch + 1) * CT * SECF * 1024, WBUF(c + 1), wave, lane);
            }
            pin_order();
        };
#pragma unroll
        for (int tt = 0; tt < CT; ++tt) {
            if (tt < nt) {
                const int t = ch * CT + tt;
                f32x16 acc;
                if constexpr (GD) gd_take(pf[0], pf[1], pf[2], pf[3]);
#pragma unroll
                for (int g = 0; g < 4; ++g) { acc[4 * g + 0] = pf[g].x; acc[4 * g + 1] = pf[g].y; acc[4 * g + 2] = pf[g].z; acc[4 * g + 3] = pf[g].w; }
                if (tt == 0) { asm volatile("" : "+v"(acc)::"memory"); fetch_next(); }
                if (t + 1 < TF) pf_issue(t + 1);
                auto fin = [&](const bf16x8& h0, const bf16x8& h1) {
                    if constexpr (SAVE) {
                        save_tile(A.save_hf, hn, t, TF, h0, h1, true);
                        mask_put(WB_F, t, TF, tile_mask_bits(h0, h1, ones));
                    }
                };
                auto act = [](f32x16& v) { relu16(v); };
                if (tt == 0) hidden_tile<0, KBE, TD, DEP>(lds_addr(WBUF(c)) + lane * 16, eb, acc, mm, act, fin);
                else hidden_tile<SECF * 1024, KBE, TD, DEP>(lds_addr(WBUF(c)) + lane * 16, eb, acc, mm, act, fin);
            }
        }
        if (ch + 1 < NCHF) {
            chunk_barrier<PF_LOADS>(A.plain_barriers != 0);   // (the Pf gathers of the next tile stay in flight)
            ++c;
        }
        TS16(24 + (ch < 14 ? ch : 13));
    }
    if constexpr (SAVE) {
        // ReLU decisions of the messages (the backward of node_agg_fn needs nothing else of them for sum / mean)
#pragma unroll
        for (int o = 0; o < TD; ++o) {
            unsigned bits = 0;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 b = *reinterpret_cast<const float4*>(sbias + DE + 2 * HC + 32 * o + 8 * g + 4 * lh);
                const float bv[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int r = 4 * g + i;
                    bits |= (mm[o][r] + bv[i] > 0.f) ? (1u << ((r >> 1) + 16 * (r & 1))) : 0u;
                }
            }
            mask_put(WB_M, o, TD, bits);
        }
    }
    if (!A.agg_out) {
        const unsigned mo = (unsigned)edge * (unsigned)dn;
#pragma unroll
        for (int o = 0; o < TD; ++o) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 b = *reinterpret_cast<const float4*>(sbias + DE + 2 * HC + 32 * o + 8 * g + 4 * lh);
                float4 v = get4(mm[o], g);
                v.x = fmaxf(v.x + b.x, 0.f); v.y = fmaxf(v.y + b.y, 0.f); v.z = fmaxf(v.z + b.z, 0.f); v.w = fmaxf(v.w + b.w, 0.f);
                strow<EXACT>(A.msg, mo, 32 * o + 8 * g + 4 * lh, dn, v, edge_ok);
            }
        }
    } else {
        // ---- aggregation in the kernel (node_agg_fn, mpn.py:89,96): the messages never reach HBM --------------------------------
        // The edges of a (direction, row) segment are consecutive, so a wave's 32 edges are a few whole segments plus at most
        // one that began in an earlier wave tile and one that goes on into the next.  Per round of RT M tiles the wave
        // transposes them through its own LDS slab ([edge][feature], the chunk buffers are free now), then lane = feature walks
        // the 32 edges IN ORDER (the walk's control flow is wave-uniform: tail bits from a ballot) and emits at every segment
        // end: whole segments straight into agg_out, the partial pieces into piece[tile][0 | 1] (0: the segment began earlier,
        // 1: it begins here and goes on), which k_agg_fixup adds up in tile order -- deterministic, no float atomics.
        // slab layout [feature][edge], 36 floats per feature row: the writes of one accumulator register are two runs of 32
        // consecutive floats, the walk reads a lane's row with eight conflict-free ds_read_b128
        constexpr int RT = AGG_RT, FR = 32 * RT, LP = 36, ROUNDS = TD / RT;
        static_assert(TD % RT == 0 && NW * FR * LP * 4 <= WB_BYTES, "aggregation slabs do not fit the chunk buffers");
        __syncthreads();   // every wave has consumed the last weight chunk
        TS16(38);
        float* const wl = reinterpret_cast<float*>(smem) + wave * (FR * LP);
        const int wt = blockIdx.x * NW + wave;
        const int tile_first = tile0 + wave * 32;
        const int key = grp * A.N + row;
        const int s0 = A.seg_ptr[key], s1 = A.seg_ptr[key + 1];
        const bool tail = edge_ok && (lj == 31 || edge_raw + 1 >= s1);
        const bool starts_here = s0 >= tile_first, ends_here = s1 <= tile_first + 32;
        const unsigned tailm = (unsigned)__ballot(tail);                       // (lanes 32..63 repeat lanes 0..31)
        const unsigned directm = (unsigned)__ballot(starts_here && ends_here);
        const unsigned slot1m = (unsigned)__ballot(starts_here && !ends_here);
        const int cnt = s1 - s0;
        const int doff = grp == 0 ? dn : 0;       // torch.cat((flow_in, flow_out)) (mpn.py:97)
#pragma unroll
        for (int q = 0; q < ROUNDS; ++q) {
#pragma unroll
            for (int tt = 0; tt < RT; ++tt) {
                const int t = RT * q + tt;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int fo = (r & 3) + 8 * (r >> 2) + 4 * lh;
                    const float v = fmaxf(mm[t][r] + sbias[DE + 2 * HC + 32 * t + fo], 0.f);
                    wl[(32 * tt + fo) * LP + lj] = edge_ok ? v : 0.f;
                }
            }
            const int f = FR * q + lane;
            const bool fok = lane < FR && f < dn;
            float acc = 0.f;
            float4 rowv[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) rowv[k] = *reinterpret_cast<const float4*>(wl + (lane < FR ? lane : 0) * LP + 4 * k);
#pragma unroll
            for (int j = 0; j < 32; ++j) {
                const float4 q4 = rowv[j >> 2];
                const float v = (j & 3) == 0 ? q4.x : ((j & 3) == 1 ? q4.y : ((j & 3) == 2 ? q4.z : q4.w));
                acc = A.agg == MPNHIP_AGG_MAX ? fmaxf(acc, v) : acc + v;   // (messages are post-ReLU: 0 is the identity of max too)
                if ((tailm >> j) & 1) {
                    const int rj = __builtin_amdgcn_readlane(row, j);
                    if ((directm >> j) & 1) {
                        float o = acc;
                        if (A.agg == MPNHIP_AGG_MEAN) { const int cj = __builtin_amdgcn_readlane(cnt, j); o /= (float)(cj > 0 ? cj : 1); }
                        if (fok) A.agg_out[(int64_t)rj * 2 * dn + doff + f] = o;
                    } else if (fok) {
                        A.piece[((int64_t)wt * 2 + ((slot1m >> j) & 1)) * DN + f] = acc;
                    }
                    acc = 0.f;
                }
            }
        }
        TS16(39);
        // the segment that begins in this tile and goes on (necessarily the tile's last): where k_agg_fixup starts
        if (lane == 0) {
            int r1 = -1;
            if (tailm) {
                const int jl = 31 - __builtin_clz(tailm);
                if ((slot1m >> jl) & 1) r1 = __builtin_amdgcn_readlane(row, jl);
            }
            A.start_row[wt] = r1;
        }
    }
    TS16(40);
#undef WBUF
}

// Adds up the pieces of the segments that cross wave tiles (edge_chain_bf16_kernel's fused aggregation): block = the tile in
// which such a segment BEGINS; it walks the following tiles' continuation pieces in order until the segment ends.
__global__ __launch_bounds__(64) void k_agg_fixup(const int* __restrict__ header, const int* __restrict__ seg_ptr,
                                                  const int* __restrict__ start_row, const float* __restrict__ piece,
                                                  float* __restrict__ agg_out, int N, int dn, int DN, int agg, int nw) {
    const int wt = blockIdx.x, blk = wt / nw, wave = wt - blk * nw;
    const int e_out = header[1], e_in = header[2];
    const int epb = 32 * nw;
    const int nb0 = (e_out + epb - 1) / epb, nb1 = (e_in + epb - 1) / epb;
    int grp, beg, end, bl;
    if (blk < nb0) { grp = 0; beg = 0; end = e_out; bl = blk; }
    else if (blk < nb0 + nb1) { grp = 1; beg = e_out; end = e_out + e_in; bl = blk - nb0; }
    else return;
    const int first = beg + bl * epb + wave * 32;
    if (first >= end) return;
    const int r = start_row[wt];
    if (r < 0) return;
    const int key = grp * N + r;
    const int s0 = seg_ptr[key], s1 = seg_ptr[key + 1];
    const int doff = grp == 0 ? dn : 0;
    for (int f = threadIdx.x; f < dn; f += 64) {
        float acc = piece[((int64_t)wt * 2 + 1) * DN + f];
        int u = wt + 1, fe = first + 32;
        while (true) {
            const float v = piece[((int64_t)u * 2) * DN + f];
            acc = agg == MPNHIP_AGG_MAX ? fmaxf(acc, v) : acc + v;
            if (s1 <= fe + 32) break;
            fe += 32;
            ++u;
        }
        if (agg == MPNHIP_AGG_MEAN) acc /= (float)(s1 - s0);
        agg_out[(int64_t)r * 2 * dn + doff + f] = acc;
    }
}

// ---- pair images -------------------------------------------------------------------------------------------------------
// One block of 64 threads per unit.  Section t of the image: KA units of the first layer's output tile t (k blocks over the
// padded input: nseg segments of seg_real columns, each padded to seg_pad), then 2 x TO units of the second layer: k block
// (2 t + c) of its input, output tile o, at index c * TO + o.
struct PairPack {
    const float* Wa; int64_t sa_n, sa_k; int a_col0, seg_real, seg_pad, nseg, H;   // first layer: Wa[n sa_n + (a_col0 + seg seg_real + k) sa_k], n < H
    const float* Wb; int64_t sb_o, sb_k; int O;                                    // second layer: Wb[o sb_o + k sb_k], o < O, k < H
    int KA, TO;
    int a_natural;   // first-layer units: element i of lane half g = k 16 kb + 8 g + i (the B operand comes from MEMORY: one 16-byte
                     // load per k block) instead of the accumulator layout's order (i & 3) + 8 (i >> 2) + 4 g
    __bf16* dst;
};
__global__ __launch_bounds__(64) void k_pack_pair_bf16(PairPack p) {
    const int sec = p.KA + 2 * p.TO;
    const int t = blockIdx.x / sec, r = blockIdx.x % sec;
    const int lane = threadIdx.x, m = lane & 31, g = lane >> 5;
    bf16x8 v;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int kk = (i & 3) + 8 * (i >> 2) + 4 * g;
        float x = 0.f;
        if (r < p.KA) {
            const int kp = 16 * r + (p.a_natural ? 8 * g + i : kk), sg = kp / p.seg_pad, k = kp % p.seg_pad, n = 32 * t + m;
            if (n < p.H && sg < p.nseg && k < p.seg_real) x = p.Wa[(int64_t)n * p.sa_n + (int64_t)(p.a_col0 + sg * p.seg_real + k) * p.sa_k];
        } else {
            const int q = r - p.KA, cblk = q / p.TO, o = q % p.TO;
            const int k = 32 * t + 16 * cblk + kk, n = 32 * o + m;
            if (n < p.O && k < p.H) x = p.Wb[(int64_t)n * p.sb_o + (int64_t)k * p.sb_k];
        }
        v[i] = (__bf16)x;
    }
    *reinterpret_cast<bf16x8*>(p.dst + ((int64_t)blockIdx.x * 64 + lane) * 8) = v;
}

int pack_pair_bf16_general(const float* Wa, int64_t sa_n, int64_t sa_k, int a_col0, int seg_real, int seg_pad, int nseg, int H,
                           const float* Wb, int64_t sb_o, int64_t sb_k, int O, int KA, int TO, int nsec, void* dst, hipStream_t s,
                           int a_natural) {
    PairPack p = {};
    p.a_natural = a_natural;
    p.Wa = Wa; p.sa_n = sa_n; p.sa_k = sa_k; p.a_col0 = a_col0; p.seg_real = seg_real; p.seg_pad = seg_pad > 0 ? seg_pad : 32; p.nseg = nseg; p.H = H;
    p.Wb = Wb; p.sb_o = sb_o; p.sb_k = sb_k; p.O = O; p.KA = KA; p.TO = TO; p.dst = static_cast<__bf16*>(dst);
    if (nsec * (KA + 2 * TO) <= 0) return MPNHIP_OK;
    hipLaunchKernelGGL(k_pack_pair_bf16, dim3(nsec * (KA + 2 * TO)), dim3(64), 0, s, p);
    MPN_LAUNCH_CHECK();
    return MPNHIP_OK;
}

static int chain_bf16_variant(int he, int de, int hn, int dn, int hc) {
    const int t1 = (he + 31) / 32, t2 = (de + 31) / 32, tf = (hn + 31) / 32, td = (dn + 31) / 32, tc = (hc + 31) / 32;
    if (t1 == 20 && t2 == 4 && tf == 14 && td == 8 && tc == 2) return 256;
    if (t1 == 10 && t2 == 2 && tf == 7 && td == 4 && tc == 1) return 128;
    if (t1 == 5 && t2 == 1 && tf == 4 && td == 2 && tc == 1) return 64;
    if (t1 == 3 && t2 == 1 && tf == 2 && td == 1 && tc == 1) return 32;
    return 0;
}

bool edge_chain_bf16_supported(int he, int de, int hn, int dn, int hc, int ef) {
    const int v = chain_bf16_variant(he, de, hn, dn, hc);
    if (v == 256 && !(he % 32 == 0 && de % 32 == 0 && hn % 32 == 0 && dn % 32 == 0)) return false;
    return v != 0 && ef == 2 && he % 4 == 0 && de % 4 == 0 && hn % 4 == 0 && dn % 4 == 0 && hc >= 1;
}

size_t chain_bf16_image_bytes(int he, int de, int hn, int dn, int hc, int ef, size_t* off_cls, size_t* off_flow0, size_t* off_flow1) {
    const size_t T1 = (he + 31) / 32, T2 = (de + 31) / 32, TF = (hn + 31) / 32, TD = (dn + 31) / 32, TC = (hc + 31) / 32;
    const size_t edge = T1 * (2 * T2 * ef + 2 * T2) * 1024, cls = TC * 2 * T2 * 1024, fl = TF * (2 * T2 + 2 * TD) * 1024;
    if (off_cls) *off_cls = edge;
    if (off_flow0) *off_flow0 = edge + cls;
    if (off_flow1) *off_flow1 = edge + cls + fl;
    return edge + cls + 2 * fl;
}

int pack_chain_bf16(const float* w_edge0, int ld_edge0, int col0_edge, int ef, const float* w_edge1, const float* w_cls0,
                    const float* const w_flow0[2], int ld_flow0, int col0_flow, const float* const w_flow1[2],
                    int he, int de, int hn, int dn, int hc, void* image, hipStream_t s) {
    const int T1 = (he + 31) / 32, T2 = (de + 31) / 32, TF = (hn + 31) / 32, TD = (dn + 31) / 32, TC = (hc + 31) / 32;
    size_t oc, of0, of1;
    chain_bf16_image_bytes(he, de, hn, dn, hc, ef, &oc, &of0, &of1);
    char* base = static_cast<char*>(image);
    // (the edge MLP's first layer takes its B operand from memory: natural k order, one 16-byte load per k block and lane)
    MPN_TRY(pack_pair_bf16_general(w_edge0, ld_edge0, 1, col0_edge, de, 32 * T2, ef, he, w_edge1, he, 1, de, 2 * T2 * ef, T2, T1, base, s, 1));
    MPN_TRY(pack_pair_bf16_general(w_cls0, de, 1, 0, de, 32 * T2, 1, hc, nullptr, 0, 0, 0, 2 * T2, 0, TC, base + oc, s));
    for (int q = 0; q < 2; ++q)
        MPN_TRY(pack_pair_bf16_general(w_flow0[q], ld_flow0, 1, col0_flow, de, 32 * T2, 1, hn, w_flow1[q], hn, 1, dn, 2 * T2, TD, TF,
                                       base + (q == 0 ? of0 : of1), s));
    return MPNHIP_OK;
}

int launch_edge_chain_bf16(const EdgeChainBf16Args& a_in, hipStream_t s) {
    if (a_in.E <= 0) return MPNHIP_OK;
    EdgeChainBf16Args a = a_in;
    if (const char* e = getenv("MPNHIP_CHAIN_BF16_PLAIN_BARRIERS")) a.plain_barriers = e[0] == '1' ? 1 : 0;
    if (const char* e = getenv("MPNHIP_CHAIN_BF16_DEBUG_SKIP")) a.debug_skip = atoi(e);
    if ((int64_t)a.E * bmax(bmax(a.he, a.dn), a.de) >= ((int64_t)1 << 32) || (int64_t)a.N * a.pw >= ((int64_t)1 << 32)) {
        set_error("edge_chain_bf16: graph too large for 32-bit row offsets");
        return MPNHIP_ERR_UNSUPPORTED;
    }
    // 256-d: 4-wave blocks, two per CU (cfg-E: 643 -> 581 us per launch; A-B switch MPNHIP_CHAIN_BF16_NW=8 for one 8-wave block)
    const int variant = chain_bf16_variant(a.he, a.de, a.hn, a.dn, a.hc);
    int epb, nwv;
    chain_bf16_geometry(a.he, a.de, a.hn, a.dn, a.hc, &epb, &nwv);
    const bool four = nwv == 4;
    const unsigned blocks = (unsigned)((a.E + epb - 1) / epb + 3);
    count_path(PC_CHAIN_FWD_BF16);
    const bool exact = a.he % 32 == 0 && a.de % 32 == 0 && a.hn % 32 == 0 && a.dn % 32 == 0;
#ifdef MPNHIP_CHAIN_TS
    static long long* ts_buf = nullptr;
    static size_t ts_cap = 0;
    static int ts_launches = 0;
    a.ts = nullptr;
    if (getenv("MPNHIP_CHAIN_TS")) {
        const size_t need = (size_t)blocks * 8 * 48 * sizeof(long long);
        if (need > ts_cap) { if (ts_buf) (void)hipFree(ts_buf); ts_cap = hipMalloc(&ts_buf, need) == hipSuccess ? need : 0; }
        if (ts_cap) { (void)hipMemsetAsync(ts_buf, 0, need, s); a.ts = ts_buf; }
    }
#endif
    static const int dep_env = [] { const char* e = getenv("MPNHIP_CHAIN16_DEPTH"); return e ? atoi(e) : 0; }();   // A-B: LDS operand reads in flight
    const bool save = a.save_mask != nullptr;
    if (save && !(a.save_h1 && a.save_hc && a.save_hf && a.save_eb)) { set_error("edge_chain_bf16: incomplete save buffers"); return MPNHIP_ERR_ARG; }
    if (save) a.e16_out = a.save_eb;
    if (!a.e_new && !a.e16_out) { set_error("edge_chain_bf16: no output for the edge features"); return MPNHIP_ERR_ARG; }
#define MPN_CB16(SV, ...) do { if (SV) MPN_LAUNCH_PROFILED((edge_chain_bf16_kernel<__VA_ARGS__, true>), dim3(blocks), dim3(four ? 256 : 512), s, a); \
                              else MPN_LAUNCH_PROFILED((edge_chain_bf16_kernel<__VA_ARGS__, false>), dim3(blocks), dim3(four ? 256 : 512), s, a); } while (0)
    switch (variant) {
        case 256:
            // (widths that are multiples of 32 only: the masked form of this variant does not fit the register budget)
            if (four && !save && dep_env == 8) MPN_LAUNCH_PROFILED((edge_chain_bf16_kernel<20, 4, 14, 8, 2, 2, true, 4, 1, false, 8>), dim3(blocks), dim3(256), s, a);
            else if (four && !save && dep_env == 6) MPN_LAUNCH_PROFILED((edge_chain_bf16_kernel<20, 4, 14, 8, 2, 2, true, 4, 1, false, 6>), dim3(blocks), dim3(256), s, a);
            else if (four && !save && !getenv("MPNHIP_CHAIN_BF16_NO_GDMA"))
                MPN_LAUNCH_PROFILED((edge_chain_bf16_kernel<20, 4, 14, 8, 2, 2, true, 4, 1, false, DEPTH, true>), dim3(blocks), dim3(256), s, a);
            else if (four) MPN_CB16(save, 20, 4, 14, 8, 2, 2, true, 4, 1);
            else MPN_CB16(save, 20, 4, 14, 8, 2, 2, true, 8, 2);
            break;
        case 128:
            if (exact) MPN_CB16(save, 10, 2, 7, 4, 1, 2, true, 8, 2);
            else MPN_CB16(save, 10, 2, 7, 4, 1, 2, false, 8, 2);
            break;
        case 64: MPN_CB16(save, 5, 1, 4, 2, 1, 2, false, 8, 2); break;
        case 32: MPN_CB16(save, 3, 1, 2, 1, 1, 2, false, 8, 2); break;
        default: set_error("edge_chain_bf16: unsupported widths"); return MPNHIP_ERR_UNSUPPORTED;
    }
#undef MPN_CB16
    MPN_LAUNCH_CHECK();
#ifdef MPNHIP_CHAIN_TS
    if (a.ts && ++ts_launches == 20) {
        (void)hipStreamSynchronize(s);
        const size_t n = (size_t)blocks * (four ? 4 : 8) * 48;
        long long* h = (long long*)malloc(n * sizeof(long long));
        (void)hipMemcpy(h, a.ts, n * sizeof(long long), hipMemcpyDeviceToHost);
        char path[512];
        snprintf(path, sizeof(path), "%s_bf16fwd.txt", getenv("MPNHIP_CHAIN_TS"));
        if (FILE* f = fopen(path, "w")) {
            for (size_t w = 0; w < n / 48; ++w) { for (int i = 0; i < 48; ++i) fprintf(f, "%lld ", h[w * 48 + i]); fprintf(f, "\n"); }
            fclose(f);
        }
        free(h);
    }
#endif
    if (a.agg_out) {
        hipLaunchKernelGGL(k_agg_fixup, dim3(blocks * (four ? 4 : 8)), dim3(64), 0, s, a.header, a.seg_ptr, a.start_row, a.piece, a.agg_out,
                           a.N, a.dn, (a.dn + 31) / 32 * 32, a.agg, four ? 4 : 8);
        MPN_LAUNCH_CHECK();
    }
    return MPNHIP_OK;
}

void chain_bf16_geometry(int he, int de, int hn, int dn, int hc, int* epb, int* nw) {
    static const int nw_env = [] { const char* e = getenv("MPNHIP_CHAIN_BF16_NW"); return e ? atoi(e) : 0; }();
    const bool four = chain_bf16_variant(he, de, hn, dn, hc) == 256 && nw_env != 8;
    *epb = four ? 128 : 256;
    *nw = four ? 4 : 8;
}

__global__ __launch_bounds__(256) void k_to_bf16(const float* __restrict__ src, unsigned short* __restrict__ dst, int64_t n4) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const float4 v = *reinterpret_cast<const float4*>(src + 4 * i);
    typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
    const bf16x4 o = {(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};
    *reinterpret_cast<uint2*>(dst + 4 * i) = __builtin_bit_cast(uint2, o);
}
int to_bf16_rows(const float* src, unsigned short* dst, int64_t n, hipStream_t s) {
    if (n <= 0) return MPNHIP_OK;
    MPN_CHECK_ARG(n % 4 == 0 && (((uintptr_t)src) & 15) == 0 && (((uintptr_t)dst) & 7) == 0, "to_bf16_rows: alignment");
    hipLaunchKernelGGL(k_to_bf16, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, s, src, dst, n / 4);
    MPN_LAUNCH_CHECK();
    return MPNHIP_OK;
}

__global__ __launch_bounds__(256) void k_bf16_debug_rows(const unsigned short* __restrict__ src, const int* __restrict__ perm, int64_t E, int width,
                                                       float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= E * width) return;
    const int64_t r = i / width;
    const int c = (int)(i - r * width);
    out[(int64_t)perm[r] * width + c] = __uint_as_float((unsigned)src[i] << 16);
}
int chain_bf16_debug_rows(const unsigned short* src, const int* perm, int64_t E, int width, float* out, hipStream_t s) {
    if (E * width <= 0) return MPNHIP_OK;
    hipLaunchKernelGGL(k_bf16_debug_rows, dim3((unsigned)((E * width + 255) / 256)), dim3(256), 0, s, src, perm, E, width, out);
    MPN_LAUNCH_CHECK();
    return MPNHIP_OK;
}

// one thread per (wave tile, lane): the lane's decisions of one section, 16 per 32-feature tile, to out[perm[edge]][feature]
__global__ __launch_bounds__(64) void k_bf16_debug_mask(const unsigned* __restrict__ mask, int wbase, int nwords, int ntiles, int width, int flow_section,
                                                      const int* __restrict__ header, const int* __restrict__ perm, int E, int epb, int nw,
                                                      float* __restrict__ out) {
    const int wt = blockIdx.x, blk = wt / nw, wave = wt - blk * nw, lane = threadIdx.x;
    const int lj = lane & 31, lh = lane >> 5;
    const int e_out = header[1], e_in = header[2];
    const int nb0 = (e_out + epb - 1) / epb, nb1 = (e_in + epb - 1) / epb;
    int beg, end, bl = blk;
    if (bl < nb0) { beg = 0; end = e_out; }
    else if (bl < nb0 + nb1) { bl -= nb0; beg = e_out; end = e_out + e_in; }
    else { bl -= nb0 + nb1; beg = e_out + e_in; end = E; if (flow_section) return; }   // (self loops: no flow MLPs, words never written)
    const int edge = beg + bl * epb + wave * 32 + lj;
    if (edge >= end) return;
    float* o = out + (int64_t)perm[edge] * width;
    for (int t = 0; t < ntiles; ++t) {
        const unsigned w = mask[((size_t)wt * nwords + wbase + (t >> 1)) * 64 + lane];
        for (int r = 0; r < 16; ++r) {
            const int f = 32 * t + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (f < width) o[f] = (w >> chain_bf16_mask_bit(t & 1, r)) & 1u ? 1.f : 0.f;
        }
    }
}
int chain_bf16_debug_mask(const unsigned* mask, int section, const int* header, const int* perm, int64_t E, int he, int de, int hn, int dn,
                          int hc, float* out, hipStream_t s) {
    if (E <= 0) return MPNHIP_OK;
    const int widths[5] = {he, de, hc, hn, dn};
    MPN_CHECK_ARG(section >= 0 && section < 5, "chain_bf16_debug_mask: section %d", section);
    int wbase = 0;
    for (int i = 0; i < section; ++i) wbase += ((widths[i] + 31) / 32 + 1) / 2;
    int epb, nw;
    chain_bf16_geometry(he, de, hn, dn, hc, &epb, &nw);
    const unsigned blocks = (unsigned)((E + epb - 1) / epb + 3);
    // (self-loop edges take no part in the flow MLPs: their HF / M words are never written -- zero the output first)
    MPN_HIP(hipMemsetAsync(out, 0, (size_t)E * widths[section] * sizeof(float), s));
    hipLaunchKernelGGL(k_bf16_debug_mask, dim3(blocks * nw), dim3(64), 0, s, mask, wbase, chain_bf16_mask_words(he, de, hn, dn, hc),
                       (widths[section] + 31) / 32, widths[section], section >= 3 ? 1 : 0, header, perm, (int)E, epb, nw, out);
    MPN_LAUNCH_CHECK();
    return MPNHIP_OK;
}

size_t chain_bf16_agg_scratch_floats(int64_t E, int dn, size_t* off_start_row) {
    const size_t tiles = (size_t)((E + 255) / 256 + 3) * 8, DN = (size_t)(dn + 31) / 32 * 32;
    if (off_start_row) *off_start_row = tiles * 2 * DN;
    return tiles * 2 * DN + tiles;
}

}  // namespace mpnhip
