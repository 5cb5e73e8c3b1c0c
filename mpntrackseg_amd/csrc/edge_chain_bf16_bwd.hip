// Backward of the bf16-operand per-edge chain of one message-passing step (mpnhip_model.precision == MPNHIP_PREC_BF16, training;
// BASELINE.json configs[4] "bf16 MLP GEMMs on MFMA" under autograd): every activation-gradient product of
//   EdgeModel   (reference models/mpn.py:67-69)   H1 = relu(W1e [e0|e] + Pr[row] + Pc[col]);  e' = relu(W2 H1 + b2)
//   classifier  (mpn.py:377 -> :114)              logit = wc2 . relu(Wc1 e' + bc1) + bc2
//   flow MLPs   (mpn.py:85-94, per direction)     M = relu(Wf2 relu(Wfe e' + Pf[col]) + bf2);  node_agg_fn (mpn.py:89,96)
// in ONE kernel, the mirror image of edge_chain_bf16.hip (same 32-edge wave tiles, same block -> edge mapping, same N-tiling):
//   B1  dZM = gather(dAGG)[row] (.) [M > 0]                       (sum / mean / max through the saved arg max)
//   B2  per HF tile t:  dZF_t = (Wf2[:, t]^T dZM) (.) [HF_t > 0]   -> B3  dE' += Wfe[t, :]^T dZF_t
//   B4  dZc = (dlog wc2) (.) [HC > 0];  dE' += Wc1^T dZc;  dZ2 = (dE' + dE_in) (.) [e' > 0]
//   B5  per H1 tile t:  dZ1_t = (W2[:, t]^T dZ2) (.) [H1_t > 0]    -> B6  dE_prev += W1e[t, e columns]^T dZ1_t
// Operands of every product (the incoming gradient and the weights) are rounded to bf16 as they enter it, fp32 accumulation --
// the arithmetic of round 3's unfused bf16 backward (gemm.hip in MPNHIP_PREC_BF16).  The ReLU decisions come as BITS written by
// the forward kernel's SAVE variant in a lane-private layout (no activation is re-read); a finished dZ tile (a lane's 16
// values) is rounded to bf16 once: those registers ARE the B operand of the next product and are what is stored -- every dZ
// block leaves as bf16 rows [E, width], which is what its consumers (the weight-gradient products, rounding their operands to
// bf16 anyway, and the scatter-adds of the node projections' gradient) read.
// The re-attached e0's share of B6 (and of the first layer's weight gradient) is hoisted: one product with the sum of the steps'
// dZ1 after the step loop (backward.hip), so this kernel contracts the e_{s-1} columns only.
#include "common.h"
#include "edge_chain.h"
#include "edge_chain_bf16_common.h"

#ifndef MPNHIP_ROWSTORE_AB
#define MPNHIP_ROWSTORE_AB 1   // 0: compile the A-B switches of the row stores out
#endif

namespace mpnhip {

// T1 = he / 32 ... as in edge_chain_bf16_kernel; NW waves per block, CT hidden tiles per weight chunk -- the SAME NW as the
// forward launch of these widths (chain_bf16_geometry): the mask words are addressed by (block, wave, lane).
template <int T1, int T2, int TF, int TD, int TC, bool EXACT, int NW, int CT>
__global__ __launch_bounds__(64 * NW, NW == 8 ? 1 : 2) void edge_chain_bf16_bwd_kernel(EdgeChainBf16BwdArgs A) {
    constexpr int EPB = 32 * NW;
    constexpr int HC = 32 * TC;
    constexpr int KBE = 2 * T2, KBM = 2 * TD;
    constexpr int SECF = KBM + 2 * T2;      // units per HF tile section: Wf2^T (k blocks of dn) | Wfe^T (2 k blocks x T2 output tiles)
    constexpr int SECC = 2 * T2;            // per HC tile: Wc1^T (2 k blocks x T2 output tiles)
    constexpr int SEC1 = KBE + 2 * T2;      // per H1 tile: W2^T (k blocks of de) | W1e^T (2 k blocks x T2 output tiles)
    constexpr int NCHF = (TF + CT - 1) / CT, NCH1 = (T1 + CT - 1) / CT;
    constexpr int CHU = bmax(bmax(CT * SECF, CT * SEC1), TC * SECC);
    constexpr int WB_E = (T1 + 1) / 2, WB_C = WB_E + (T2 + 1) / 2, WB_F = WB_C + (TC + 1) / 2, WB_M = WB_F + (TF + 1) / 2;
    constexpr int NWORDS = WB_M + (TD + 1) / 2;
    __shared__ __attribute__((aligned(16))) char smem[2 * CHU * 1024 + HC * 4];
    __shared__ __attribute__((aligned(16))) char rowslab[NW * ROW_SLAB_BYTES];   // full-line row stores (RowStage)
    float* const swc2 = reinterpret_cast<float*>(smem + 2 * CHU * 1024);
#define WBUF(i) (smem + ((i) & 1) * (CHU * 1024))

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lj = lane & 31, lh = lane >> 5;
    const int he = A.he, de = A.de, hn = A.hn, dn = A.dn, hc = A.hc;

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
    const char* const imgf = static_cast<const char*>(grp == 1 ? A.img_flow[1] : A.img_flow[0]);
    const char* const img1 = static_cast<const char*>(A.img_edge);

    if (flow) chunk_fetch<bmin(CT, TF) * SECF, NW>(imgf, WBUF(0), wave, lane);
    else chunk_fetch<TC * SECC, NW>(static_cast<const char*>(A.img_cls), WBUF(0), wave, lane);
    for (int i = tid; i < HC; i += 64 * NW) swc2[i] = i < hc ? A.wc2[i] : 0.f;

    const unsigned* const mask_wt = A.mask + ((size_t)(blockIdx.x * NW + wave) * NWORDS) * 64 + lane;
    RowStage rs;
    rs.init(rowslab + wave * ROW_SLAB_BYTES, lane, tile0 + wave * 32, end);
    // a finished dZ tile t of T to its bf16 rows [E, width]: in pairs, 128 bytes per row = whole lines (RowStage); a lone last tile
    // straight from the registers.  Plain stores: the scatter-adds and the weight-gradient products read these rows right away
    // (non-temporal: 611 -> 756 us per launch at cfg-E; MPNHIP_CHAIN_BF16_DEBUG_SKIP=4 selects them, A-B)
    auto save_tile = [&](unsigned short* base, int width, int t, int T, const bf16x8& h0, const bf16x8& h1) {
        if (A.debug_skip & 1) return;
        uint4 lo, hi;
        tile_rows16(h0, h1, lo, hi);
        if ((!(t & 1) && t + 1 == T) || (MPNHIP_ROWSTORE_AB && (A.debug_skip & 8))) {   // (8: A-B, every tile straight from the registers)
            const int f = 32 * t + 16 * lh;
            unsigned short* q = base + (size_t)edge * width + f;
            if (edge_ok && (EXACT || f < width)) *reinterpret_cast<uint4*>(q) = lo;
            if (edge_ok && (EXACT || f + 8 < width)) *reinterpret_cast<uint4*>(q + 8) = hi;
            return;
        }
        rs.put16(t & 1, lo, hi);
        if (t & 1) {
            if (A.debug_skip & 4) rs.flush<true>(reinterpret_cast<char*>(base), (size_t)width * 2, 64 * (t - 1), (width - 32 * (t - 1)) * 2);
            else rs.flush<false>(reinterpret_cast<char*>(base), (size_t)width * 2, 64 * (t - 1), (width - 32 * (t - 1)) * 2);
        }
    };
    const float dl = A.dlog[A.perm[edge]];

    f32x16 dEa[T2];   // dE': gradient w.r.t. e' (pre-mask), accumulated over B3 / B4
#pragma unroll
    for (int o = 0; o < T2; ++o)
#pragma unroll
        for (int r = 0; r < 16; ++r) dEa[o][r] = 0.f;
    int c = 0;

    if (flow) {
        // ---- B1: dZM = gather(dAGG)[row] (.) [M > 0]  (node_agg_fn backward, mpn.py:89,96) ---------------------------------
        bf16x8 X[KBM];
        unsigned mwf[(TF + 1) / 2];
        {
            const int row = A.srow[edge];
            const unsigned ro = (unsigned)row * (unsigned)(2 * dn) + (unsigned)(grp == 0 ? dn : 0);
            float scale = 1.f;
            if (A.agg == MPNHIP_AGG_MEAN) {
                const int key = grp * A.N + row;
                const int cnt = A.seg_ptr[key + 1] - A.seg_ptr[key];
                scale = (float)(cnt > 0 ? cnt : 1);
            }
#pragma unroll
            for (int w = 0; w < (TF + 1) / 2; ++w) mwf[w] = mask_wt[(size_t)(WB_F + w) * 64];
#pragma unroll
            for (int t = 0; t < TD; ++t) {
                f32x16 v;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int n = 32 * t + 8 * g + 4 * lh;
                    float4 q = ldrow<EXACT>(A.dAGG, ro, n, dn);
                    if (A.agg == MPNHIP_AGG_MEAN) { q.x /= scale; q.y /= scale; q.z /= scale; q.w /= scale; }
                    if (A.agg == MPNHIP_AGG_MAX) {
                        const int4 ar = *reinterpret_cast<const int4*>(A.ARG + (size_t)ro + (EXACT || n < dn ? n : 0));
                        q.x = ar.x == edge_raw ? q.x : 0.f; q.y = ar.y == edge_raw ? q.y : 0.f;
                        q.z = ar.z == edge_raw ? q.z : 0.f; q.w = ar.w == edge_raw ? q.w : 0.f;
                    }
                    v[4 * g + 0] = q.x; v[4 * g + 1] = q.y; v[4 * g + 2] = q.z; v[4 * g + 3] = q.w;
                }
                const unsigned mw = mask_wt[(size_t)(WB_M + (t >> 1)) * 64];
                if (t & 1) apply_mask16<1>(v, mw); else apply_mask16<0>(v, mw);
                X[2 * t] = pack_regs(v, 0);
                X[2 * t + 1] = pack_regs(v, 1);
                save_tile(A.dZM, dn, t, TD, X[2 * t], X[2 * t + 1]);
            }
        }
        __syncthreads();   // chunk 0 and wc2 are in LDS (every load above has landed)

        // ---- B2 + B3: per HF tile  dZF_t = (Wf2[:, t]^T dZM) (.) [HF_t > 0]  ->  dE' += Wfe[t, :]^T dZF_t ---------------------
#pragma unroll
        for (int ch = 0; ch < NCHF; ++ch) {
            const int nt = bmin(CT, TF - ch * CT);
            if (ch + 1 < NCHF) {
                if (TF - (ch + 1) * CT >= CT) chunk_fetch<CT * SECF, NW>(imgf + (size_t)(ch + 1) * CT * SECF * 1024, WBUF(c + 1), wave, lane);
                else chunk_fetch<(TF % CT ? TF % CT : CT) * SECF, NW>(imgf + (size_t)(ch + 1) * CT * SECF * 1024, WBUF(c + 1), wave, lane);
            } else {
                chunk_fetch<TC * SECC, NW>(static_cast<const char*>(A.img_cls), WBUF(c + 1), wave, lane);
            }
            pin_order();
#pragma unroll
            for (int tt = 0; tt < CT; ++tt) {
                if (tt < nt) {
                    const int t = ch * CT + tt;
                    f32x16 acc;
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
                    const unsigned mw = mwf[t >> 1];
                    auto act = [&](f32x16& v) { if (t & 1) apply_mask16<1>(v, mw); else apply_mask16<0>(v, mw); };
                    auto fin = [&](const bf16x8& h0, const bf16x8& h1) { save_tile(A.dZF, hn, t, TF, h0, h1); };
                    if (tt == 0) hidden_tile<0, KBM, T2>(lds_addr(WBUF(c)) + lane * 16, X, acc, dEa, act, fin);
                    else hidden_tile<SECF * 1024, KBM, T2>(lds_addr(WBUF(c)) + lane * 16, X, acc, dEa, act, fin);
                }
            }
            __syncthreads();   // (counted waits that leave the row stores in flight measured the same: 42.0 vs 42.4 ms per cfg-E step)
            ++c;
        }
    } else {
        __syncthreads();
    }

    // ---- B4: classifier (its image is in the current buffer): dZc = (dlog wc2) (.) [HC > 0];  dE' += Wc1^T dZc ----------------
    // the gradient arriving from the later step joins here (fetched under the classifier's products)
    chunk_fetch<bmin(CT, T1) * SEC1, NW>(img1, WBUF(c + 1), wave, lane);
    pin_order();
    float4 din[4 * T2];
    {
        const unsigned eo = (unsigned)edge * (unsigned)de;
#pragma unroll
        for (int o = 0; o < T2; ++o)
#pragma unroll
            for (int g = 0; g < 4; ++g) din[4 * o + g] = ldrow<EXACT>(A.dE_in, eo, 32 * o + 8 * g + 4 * lh, de);
    }
    unsigned mw1[(T1 + 1) / 2];
#pragma unroll
    for (int w = 0; w < (T1 + 1) / 2; ++w) mw1[w] = mask_wt[(size_t)w * 64];
    {
        const unsigned mwc = mask_wt[(size_t)WB_C * 64];
        static_assert(TC <= 2, "classifier hidden width up to 64");
#pragma unroll
        for (int q = 0; q < TC; ++q) {
            f32x16 v;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 w = *reinterpret_cast<const float4*>(swc2 + 32 * q + 8 * g + 4 * lh);
                // (the forward's dot product rounds wc2 to bf16: its gradient w.r.t. the hidden activation is dlog x that value)
                v[4 * g + 0] = dl * (float)(__bf16)w.x; v[4 * g + 1] = dl * (float)(__bf16)w.y;
                v[4 * g + 2] = dl * (float)(__bf16)w.z; v[4 * g + 3] = dl * (float)(__bf16)w.w;
            }
            if (q & 1) apply_mask16<1>(v, mwc); else apply_mask16<0>(v, mwc);
            bf16x8 hb[2] = {pack_regs(v, 0), pack_regs(v, 1)};
            save_tile(A.dZc, hc, q, TC, hb[0], hb[1]);
            const unsigned wa = lds_addr(WBUF(c)) + lane * 16;
            auto cls_tile = [&](auto Q) {
                stream_units<Q.value * SECC * 1024, 2 * T2>(wa, [&](auto U, const bf16x8& a) {
                    constexpr int u = U.value;
                    dEa[u % T2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, hb[u / T2], dEa[u % T2], 0, 0, 0);
                });
            };
            if (q == 0) cls_tile(std::integral_constant<int, 0>{});
            else cls_tile(std::integral_constant<int, (TC > 1 ? 1 : 0)>{});
        }
    }
    // ---- dZ2 = (dE' + dE_in) (.) [e' > 0]: out, and as the B operand of B5 -----------------------------------------------------
    bf16x8 X2[KBE];
    {
#pragma unroll
        for (int o = 0; o < T2; ++o) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                dEa[o][4 * g + 0] += din[4 * o + g].x; dEa[o][4 * g + 1] += din[4 * o + g].y;
                dEa[o][4 * g + 2] += din[4 * o + g].z; dEa[o][4 * g + 3] += din[4 * o + g].w;
            }
            const unsigned mwe = mask_wt[(size_t)(WB_E + (o >> 1)) * 64];
            if (o & 1) apply_mask16<1>(dEa[o], mwe); else apply_mask16<0>(dEa[o], mwe);
            X2[2 * o] = pack_regs(dEa[o], 0);
            X2[2 * o + 1] = pack_regs(dEa[o], 1);
            save_tile(A.dZ2, de, o, T2, X2[2 * o], X2[2 * o + 1]);
        }
    }
    // dE_prev accumulators; at the first step e_{s-1} IS the re-attached e0: add into its running gradient (read here, used as C-in)
    f32x16 dp[T2];
    float* const dst = A.first_step ? A.dE0 : A.dEprev;
    {
        const unsigned eo = (unsigned)edge * (unsigned)de;
#pragma unroll
        for (int o = 0; o < T2; ++o)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
                if (A.first_step) q = ldrow<EXACT>(A.dE0, eo, 32 * o + 8 * g + 4 * lh, de);
                dp[o][4 * g + 0] = q.x; dp[o][4 * g + 1] = q.y; dp[o][4 * g + 2] = q.z; dp[o][4 * g + 3] = q.w;
            }
    }
    __syncthreads();   // every wave is done with the classifier image; the first edge chunk has landed (all loads above too)
    ++c;

    // ---- B5 + B6: per H1 tile  dZ1_t = (W2[:, t]^T dZ2) (.) [H1_t > 0]  ->  dE_prev += W1e[t, e columns]^T dZ1_t ---------------
#pragma unroll
    for (int ch = 0; ch < NCH1; ++ch) {
        const int nt = bmin(CT, T1 - ch * CT);
        if (ch + 1 < NCH1) {
            if (T1 - (ch + 1) * CT >= CT) chunk_fetch<CT * SEC1, NW>(img1 + (size_t)(ch + 1) * CT * SEC1 * 1024, WBUF(c + 1), wave, lane);
            else chunk_fetch<(T1 % CT ? T1 % CT : CT) * SEC1, NW>(img1 + (size_t)(ch + 1) * CT * SEC1 * 1024, WBUF(c + 1), wave, lane);
            pin_order();
        }
#pragma unroll
        for (int tt = 0; tt < CT; ++tt) {
            if (tt < nt) {
                const int t = ch * CT + tt;
                f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 0.f;
                const unsigned mw = mw1[t >> 1];
                auto act = [&](f32x16& v) { if (t & 1) apply_mask16<1>(v, mw); else apply_mask16<0>(v, mw); };
                auto fin = [&](const bf16x8& h0, const bf16x8& h1) { save_tile(A.dZ1, he, t, T1, h0, h1); };
                if (tt == 0) hidden_tile<0, KBE, T2>(lds_addr(WBUF(c)) + lane * 16, X2, acc, dp, act, fin);
                else hidden_tile<SEC1 * 1024, KBE, T2>(lds_addr(WBUF(c)) + lane * 16, X2, acc, dp, act, fin);
            }
        }
        if (ch + 1 < NCH1) {
            __syncthreads();
            ++c;
        }
    }
#pragma unroll
    for (int o = 0; o < T2; ++o) {   // (fp32 tiles: 128 bytes per row, whole lines through the slab)
        if (MPNHIP_ROWSTORE_AB && (A.debug_skip & 16)) {
            const unsigned eo = (unsigned)edge * (unsigned)de;
#pragma unroll
            for (int g = 0; g < 4; ++g) strow<EXACT>(dst, eo, 32 * o + 8 * g + 4 * lh, de, get4(dp[o], g), edge_ok);
            continue;
        }
        rs.put32(dp[o]);
        rs.flush<false>(reinterpret_cast<char*>(dst), (size_t)de * 4, 128 * o, (de - 32 * o) * 4);
    }
#undef WBUF
}

// ---- backward pair images (the forward's unit format, transposed roles) --------------------------------------------------------
size_t chain_bf16_bwd_image_bytes(int he, int de, int hn, int dn, int hc, size_t* off_cls, size_t* off_flow0, size_t* off_flow1) {
    const size_t T1 = (he + 31) / 32, T2 = (de + 31) / 32, TF = (hn + 31) / 32, TD = (dn + 31) / 32, TC = (hc + 31) / 32;
    const size_t edge = T1 * (2 * T2 + 2 * T2) * 1024, cls = TC * 2 * T2 * 1024, fl = TF * (2 * TD + 2 * T2) * 1024;
    if (off_cls) *off_cls = edge;
    if (off_flow0) *off_flow0 = edge + cls;
    if (off_flow1) *off_flow1 = edge + cls + fl;
    return edge + cls + 2 * fl;
}

int pack_chain_bf16_bwd(const float* w_edge0, int ld_edge0, int col_e, const float* w_edge1, const float* w_cls0,
                        const float* const w_flow0[2], int ld_flow0, int col0_flow, const float* const w_flow1[2],
                        int he, int de, int hn, int dn, int hc, void* image, hipStream_t s) {
    const int T1 = (he + 31) / 32, T2 = (de + 31) / 32, TF = (hn + 31) / 32, TD = (dn + 31) / 32, TC = (hc + 31) / 32;
    size_t oc, of0, of1;
    chain_bf16_bwd_image_bytes(he, de, hn, dn, hc, &oc, &of0, &of1);
    char* base = static_cast<char*>(image);
    // edge: first layer of the tile = W2^T (rows n = H1 features, k over de: W2[k][n]); second = W1e^T (rows o = e_{s-1} columns,
    // k over he: W1[k][col_e + o])
    MPN_TRY(pack_pair_bf16_general(w_edge1, 1, he, 0, de, 32 * T2, 1, he, w_edge0 + col_e, 1, ld_edge0, de, 2 * T2, T2, T1, base, s));
    // classifier: no first layer; second = Wc1^T (rows o = e' features, k over hc: Wc1[k][o])
    MPN_TRY(pack_pair_bf16_general(nullptr, 0, 0, 0, 0, 32, 0, hc, w_cls0, 1, de, de, 0, T2, TC, base + oc, s));
    for (int q = 0; q < 2; ++q)
        // flow: first = Wf2^T (rows n = HF features, k over dn: Wf2[k][n]); second = Wfe^T (rows o = e' features, k over hn)
        MPN_TRY(pack_pair_bf16_general(w_flow1[q], 1, hn, 0, dn, 32 * TD, 1, hn, w_flow0[q] + col0_flow, 1, ld_flow0, de, 2 * TD, T2, TF,
                                       base + (q == 0 ? of0 : of1), s));
    return MPNHIP_OK;
}

int launch_edge_chain_bf16_bwd(const EdgeChainBf16BwdArgs& a_in, hipStream_t s) {
    if (a_in.E <= 0) return MPNHIP_OK;
    EdgeChainBf16BwdArgs a = a_in;
    if (const char* e = getenv("MPNHIP_CHAIN_BF16_PLAIN_BARRIERS")) a.plain_barriers = e[0] == '1' ? 1 : 0;
    if (const char* e = getenv("MPNHIP_CHAIN_BF16_DEBUG_SKIP")) a.debug_skip = atoi(e);
    const int wmax = a.he > a.dn ? a.he : a.dn;
    if ((int64_t)a.E * wmax >= ((int64_t)1 << 32) || (int64_t)a.N * 2 * a.dn >= ((int64_t)1 << 32)) {
        set_error("edge_chain_bf16_bwd: graph too large for 32-bit row offsets");
        return MPNHIP_ERR_UNSUPPORTED;
    }
    int epb, nw;
    chain_bf16_geometry(a.he, a.de, a.hn, a.dn, a.hc, &epb, &nw);
    const unsigned blocks = (unsigned)((a.E + epb - 1) / epb + 3);
    const bool exact = a.he % 32 == 0 && a.de % 32 == 0 && a.hn % 32 == 0 && a.dn % 32 == 0 && a.hc % 32 == 0;
    const int t1 = (a.he + 31) / 32, t2 = (a.de + 31) / 32, tf = (a.hn + 31) / 32, td = (a.dn + 31) / 32, tc = (a.hc + 31) / 32;
    if (!edge_chain_bf16_bwd_supported(a.he, a.de, a.hn, a.dn, a.hc)) {
        set_error("edge_chain_bf16_bwd: unsupported widths");
        return MPNHIP_ERR_UNSUPPORTED;
    }
    count_path(PC_CHAIN_BWD_BF16);
#define MPN_CBB(...) MPN_LAUNCH_PROFILED((edge_chain_bf16_bwd_kernel<__VA_ARGS__>), dim3(blocks), dim3(64 * nw), s, a)
    if (t1 == 20 && t2 == 4 && tf == 14 && td == 8 && tc == 2 && exact) {
        if (nw == 4) MPN_CBB(20, 4, 14, 8, 2, true, 4, 1);
        else MPN_CBB(20, 4, 14, 8, 2, true, 8, 2);
    } else if (t1 == 10 && t2 == 2 && tf == 7 && td == 4 && tc == 1) {
        if (exact) MPN_CBB(10, 2, 7, 4, 1, true, 8, 2);
        else MPN_CBB(10, 2, 7, 4, 1, false, 8, 2);
    } else if (t1 == 5 && t2 == 1 && tf == 4 && td == 2 && tc == 1) {
        MPN_CBB(5, 1, 4, 2, 1, false, 8, 2);
    } else if (t1 == 3 && t2 == 1 && tf == 2 && td == 1 && tc == 1) {
        MPN_CBB(3, 1, 2, 1, 1, false, 8, 2);
    } else {
        set_error("edge_chain_bf16_bwd: unsupported widths");
        return MPNHIP_ERR_UNSUPPORTED;
    }
#undef MPN_CBB
    MPN_LAUNCH_CHECK();
    return MPNHIP_OK;
}

}  // namespace mpnhip
