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
#include "common.h"
#include "edge_chain.h"
#include <type_traits>

namespace mpnhip {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

namespace {

constexpr int bmax(int a, int b) { return a > b ? a : b; }
constexpr int bmin(int a, int b) { return a < b ? a : b; }

__device__ __forceinline__ float4 ldg4(const float* p) { return *reinterpret_cast<const float4*>(p); }
template <bool EXACT>
__device__ __forceinline__ float4 ldrow(const float* base, unsigned off, int n, int dim) {
    if (EXACT) return ldg4(base + (size_t)off + n);
    const bool ok = n < dim;
    float4 v = ldg4(base + (size_t)off + (ok ? n : 0));
    v.x = ok ? v.x : 0.f; v.y = ok ? v.y : 0.f; v.z = ok ? v.z : 0.f; v.w = ok ? v.w : 0.f;
    return v;
}
template <bool EXACT>
__device__ __forceinline__ void strow(float* base, unsigned off, int n, int dim, float4 v, bool ok) {
    // (plain stores: a row's 128-byte lines are completed by several wave instructions and L2 merges the pieces; non-temporal
    // stores / loads here took the kernel from 0.63 to 0.92 ms at cfg-E)
    if (ok && (EXACT || n < dim)) *reinterpret_cast<float4*>(base + (size_t)off + n) = v;
}
__device__ __forceinline__ float4 get4(const f32x16& a, int g) {
    return make_float4(a[4 * g + 0], a[4 * g + 1], a[4 * g + 2], a[4 * g + 3]);
}
__device__ __forceinline__ void relu16(f32x16& a) {
#pragma unroll
    for (int r = 0; r < 16; ++r) a[r] = fmaxf(a[r], 0.f);
}
__device__ __forceinline__ bf16x8 pack8(float4 u, float4 v) {
    return bf16x8{(__bf16)u.x, (__bf16)u.y, (__bf16)u.z, (__bf16)u.w, (__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};
}
// registers 8c .. 8c+7 of a finished tile = the B operand of k block c of the next layer
__device__ __forceinline__ bf16x8 pack_regs(const f32x16& s, int c) {
    return c == 0 ? bf16x8{(__bf16)s[0], (__bf16)s[1], (__bf16)s[2], (__bf16)s[3], (__bf16)s[4], (__bf16)s[5], (__bf16)s[6], (__bf16)s[7]}
                  : bf16x8{(__bf16)s[8], (__bf16)s[9], (__bf16)s[10], (__bf16)s[11], (__bf16)s[12], (__bf16)s[13], (__bf16)s[14], (__bf16)s[15]};
}

__device__ __forceinline__ unsigned lds_addr(const void* p) {
    return (unsigned)(size_t)(const __attribute__((address_space(3))) void*)p;
}
// A operands come out of the LDS-DMA target by inline assembly (edge_chain.hip, lds_read3: a compiler-visible read of that
// object is preceded by s_waitcnt vmcnt(0), which would drain the next chunk's DMA and the gathers in flight); the waits are
// placed by hand (a wave's LDS operations complete in order).
template <int OFF>
__device__ __forceinline__ void lds_read(unsigned base, bf16x8& a) {
    // (the unit's offset rides in the instruction: as register values the ~50 distinct addresses of a chunk stay live across
    // the whole kernel and spill)
    static_assert(OFF >= 0 && OFF < 65536, "ds_read offset field");
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(a) : "v"(base), "n"(OFF) : "memory");
}
template <int N>
__device__ __forceinline__ void lds_wait(bf16x8& a) {
    asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(a) : "n"(N));
}
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// NU KiB of an image -> LDS by LDS-DMA: unit 8 q + wave is moved by wave `wave` (one 1 KiB wave instruction each)
template <int NU, int NW = 8>
__device__ __forceinline__ void chunk_fetch(const char* src, char* buf, int wave, int lane) {
#pragma unroll
    for (int q = 0; q < (NU + NW - 1) / NW; ++q) {
        if (NW * q + wave < NU)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (size_t)(NW * q + wave) * 1024 + lane * 16),
                                             (__attribute__((address_space(3))) void*)(buf + (NW * q + wave) * 1024), 16, 0, 0);
    }
}

// Barrier at the end of a weight chunk WITHOUT draining the gathers in flight: __syncthreads() carries a fence, i.e.
// s_waitcnt vmcnt(0), which also waits for the next tile's row gathers issued a moment ago -- once per chunk, the whole gather
// latency exposed (ablation at cfg-E: 151 of the launch's 592 us).  What the barrier needs is the NEXT chunk's LDS-DMA (issued at
// the top of this chunk, pinned there by a compiler barrier) and nothing younger: vmcnt(N) with N = the loads issued after it,
// which the loop knows exactly (N row gathers of the next tile).  Other waves' DMA pieces are covered by their own waits.
// lgkmcnt(0) rides along (free here): a wave must not cross the barrier with ds_reads of the current buffer outstanding while
// another wave's DMA of the chunk after next overwrites it.  `plain` (EdgeChainBf16Args::plain_barriers, MPNHIP_CHAIN_BF16_PLAIN_BARRIERS=1):
// the A-B fallback to __syncthreads() -- a test compares the two bit for bit (tests/test_gpu_parity.py).
template <int N>
__device__ __forceinline__ void chunk_barrier(bool plain) {
    if (plain) __syncthreads();
    else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(N) : "memory");
}
__device__ __forceinline__ void pin_order() { asm volatile("" ::: "memory"); }

constexpr int DEPTH = 4;   // A operands in flight per wave

// NU units from LDS address wa + OFF0 (+ 1 KiB per unit), DEPTH reads in flight; use(u, a) consumes unit u's A operand
template <int OFF0, int NU, class USE>
__device__ __forceinline__ void stream_units(unsigned wa, USE&& use) {
    bf16x8 a[DEPTH];
    static_for<0, (DEPTH - 1 < NU ? DEPTH - 1 : NU)>([&](auto U) { lds_read<OFF0 + U.value * 1024>(wa, a[U.value]); });
    static_for<0, NU>([&](auto U) {
        constexpr int u = U.value;
        if constexpr (u + DEPTH - 1 < NU) {
            lds_read<OFF0 + (u + DEPTH - 1) * 1024>(wa, a[(u + DEPTH - 1) % DEPTH]);
            lds_wait<DEPTH - 1>(a[u % DEPTH]);
        } else {
            lds_wait<0>(a[u % DEPTH]);
        }
        use(U, a[u % DEPTH]);
    });
}

// One hidden tile: KA first-layer units against the k blocks xin[0 .. KA) into `acc`; ReLU; then 2 x TO second-layer units:
// k block c of the finished tile (rounded to bf16) into out[o].  wa = LDS address of the chunk buffer (+ lane * 16), OFF0 = the
// section's offset in it.
template <int OFF0, int KA, int TO>
__device__ __forceinline__ void hidden_tile(unsigned wa, const bf16x8* xin, f32x16& acc, f32x16* out) {
    bf16x8 hb[2];
    stream_units<OFF0, KA + 2 * TO>(wa, [&](auto U, const bf16x8& a) {
        constexpr int u = U.value;
        if constexpr (u < KA) {
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, xin[u], acc, 0, 0, 0);
        } else {
            if constexpr (u == KA) {
                relu16(acc);
                hb[0] = pack_regs(acc, 0);
                hb[1] = pack_regs(acc, 1);
            }
            constexpr int q = u - KA;
            out[q % TO] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, hb[q / TO], out[q % TO], 0, 0, 0);
        }
    });
}

}  // namespace

// T1 = ceil(he/32), T2 = ceil(de/32), TF = ceil(hn/32), TD = ceil(dn/32), TC = ceil(hc/32); EF = 1: first-layer input e,
// 2: [e0 | e] (each half padded to 32 T2 columns in the image).
// NW waves per block (8: 256 edges, one block per CU; 4: 128 edges, two independent blocks per CU whose serial phases -- the
// first-layer input rows at the start, the aggregation at the end -- run under each other's MFMA phases, for twice the weight
// stream), CTI hidden tiles per weight chunk.
template <int T1, int T2, int TF, int TD, int TC, int EF, bool EXACT, int NW = 8, int CTI = 2>
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
    __shared__ __attribute__((aligned(16))) char smem[WB_BYTES + (DE + 2 * HC + DN) * 4];
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
    const unsigned pco = (unsigned)col * (unsigned)A.pw + (unsigned)he;
    const unsigned pfo = pco + (unsigned)(he + (grp == 1 ? hn : 0));
    // ---- first-layer input: this lane's edge row(s), k = 16 kb + 4h + (0..3), 16 kb + 8 + 4h + (0..3) per k block ------
    bf16x8 X[KB1];
#pragma unroll
    for (int sg = 0; sg < EF; ++sg) {
        const float* xr = sg == 0 ? A.xa + (int64_t)edge * A.ldxa : A.xb + (int64_t)edge * A.ldxb;
#pragma unroll
        for (int kb = 0; kb < KBE; ++kb)
            X[sg * KBE + kb] = pack8(ldrow<EXACT>(xr, 0u, 16 * kb + 4 * lh, de), ldrow<EXACT>(xr, 0u, 16 * kb + 8 + 4 * lh, de));
    }
    // gathered C-in of one H1 tile: Pr[row] and Pc[col], 4 row pieces each; fetched one tile ahead
    // (CIN_LOADS / PF_LOADS: the vector-memory operations cin_issue / pf_issue put behind a chunk's LDS-DMA -- what the counted
    // chunk barriers below leave in flight; change the loops and these together)
    constexpr int CIN_LOADS = 8, PF_LOADS = 4;
    float4 cin[CIN_LOADS];
    static_assert(sizeof(cin) / sizeof(cin[0]) == 2 * 4 && CIN_LOADS == 2 * 4, "cin_issue issues 2 x 4 row loads");
    auto cin_issue = [&](int t) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            cin[g] = ldrow<EXACT>(A.P, pro, 32 * t + 8 * g + 4 * lh, he);
            cin[4 + g] = ldrow<EXACT>(A.P, pco, 32 * t + 8 * g + 4 * lh, he);
        }
    };
    float4 pf[PF_LOADS];
    static_assert(sizeof(pf) / sizeof(pf[0]) == 4 && PF_LOADS == 4, "pf_issue issues 4 row loads");
    auto pf_issue = [&](int t) {
#pragma unroll
        for (int g = 0; g < 4; ++g) pf[g] = ldrow<EXACT>(A.P, pfo, 32 * t + 8 * g + 4 * lh, hn);
    };
    cin_issue(0);
    f32x16 en[T2];
#pragma unroll
    for (int o = 0; o < T2; ++o)
#pragma unroll
        for (int r = 0; r < 16; ++r) en[o][r] = 0.f;
    __syncthreads();   // chunk 0 and the biases are in LDS

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
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    acc[4 * g + 0] = cin[g].x + cin[4 + g].x; acc[4 * g + 1] = cin[g].y + cin[4 + g].y;
                    acc[4 * g + 2] = cin[g].z + cin[4 + g].z; acc[4 * g + 3] = cin[g].w + cin[4 + g].w;
                }
                if (tt == 0) { asm volatile("" : "+v"(acc)::"memory"); fetch_next(); }   // (acc is formed before the DMA goes out)
                if (t + 1 < T1) cin_issue(t + 1);
                else if (flow) pf_issue(0);
                if (tt == 0) hidden_tile<0, KB1, T2>(lds_addr(WBUF(c)) + lane * 16, X, acc, en);
                else hidden_tile<SEC1 * 1024, KB1, T2>(lds_addr(WBUF(c)) + lane * 16, X, acc, en);
            }
        }
        // (the chunk's last tile issued the 8 gathers of the tile after it -- except the very last one: 4 or none, drain)
        if (ch * CT + nt < T1) chunk_barrier<CIN_LOADS>(A.plain_barriers != 0);
        else __syncthreads();
        ++c;
    }
    // ---- e' = relu(. + b2): out, and as the B operand of the classifier and the flow MLPs -------------------------------
    bf16x8 eb[KBE];
    {
        const unsigned eo = (unsigned)edge * (unsigned)de;
#pragma unroll
        for (int o = 0; o < T2; ++o) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 b = *reinterpret_cast<const float4*>(sbias + 32 * o + 8 * g + 4 * lh);
                en[o][4 * g + 0] += b.x; en[o][4 * g + 1] += b.y; en[o][4 * g + 2] += b.z; en[o][4 * g + 3] += b.w;
            }
            relu16(en[o]);
#pragma unroll
            for (int g = 0; g < 4; ++g) strow<EXACT>(A.e_new, eo, 32 * o + 8 * g + 4 * lh, de, get4(en[o], g), edge_ok);
            eb[2 * o] = pack_regs(en[o], 0);
            eb[2 * o + 1] = pack_regs(en[o], 1);
        }
    }
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
                stream_units<Q.value * SECC * 1024, KBE>(wa, [&](auto U, const bf16x8& a) {
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, eb[U.value], acc, 0, 0, 0);
                });
            };
            if (q == 0) cls_tile(std::integral_constant<int, 0>{});
            else cls_tile(std::integral_constant<int, (TC > 1 ? 1 : 0)>{});
            static_assert(TC <= 2, "classifier hidden width up to 64");
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
#pragma unroll
                for (int g = 0; g < 4; ++g) { acc[4 * g + 0] = pf[g].x; acc[4 * g + 1] = pf[g].y; acc[4 * g + 2] = pf[g].z; acc[4 * g + 3] = pf[g].w; }
                if (tt == 0) { asm volatile("" : "+v"(acc)::"memory"); fetch_next(); }
                if (t + 1 < TF) pf_issue(t + 1);
                if (tt == 0) hidden_tile<0, KBE, TD>(lds_addr(WBUF(c)) + lane * 16, eb, acc, mm);
                else hidden_tile<SECF * 1024, KBE, TD>(lds_addr(WBUF(c)) + lane * 16, eb, acc, mm);
            }
        }
        if (ch + 1 < NCHF) {
            chunk_barrier<PF_LOADS>(A.plain_barriers != 0);   // (the Pf gathers of the next tile stay in flight)
            ++c;
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
    const float* Wa; int lda, a_col0, seg_real, seg_pad, nseg, H;   // first layer: W[n][a_col0 + seg * seg_real + k], n < H
    const float* Wb; int ldb, O;                                   // second layer: W[o][k], o < O, k < H
    int KA, TO;
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
            const int kp = 16 * r + kk, sg = kp / p.seg_pad, k = kp % p.seg_pad, n = 32 * t + m;
            if (n < p.H && sg < p.nseg && k < p.seg_real) x = p.Wa[(int64_t)n * p.lda + p.a_col0 + sg * p.seg_real + k];
        } else {
            const int q = r - p.KA, cblk = q / p.TO, o = q % p.TO;
            const int k = 32 * t + 16 * cblk + kk, n = 32 * o + m;
            if (n < p.O && k < p.H) x = p.Wb[(int64_t)n * p.ldb + k];
        }
        v[i] = (__bf16)x;
    }
    *reinterpret_cast<bf16x8*>(p.dst + ((int64_t)blockIdx.x * 64 + lane) * 8) = v;
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
    PairPack p = {};
    p.Wa = w_edge0; p.lda = ld_edge0; p.a_col0 = col0_edge; p.seg_real = de; p.seg_pad = 32 * T2; p.nseg = ef; p.H = he;
    p.Wb = w_edge1; p.ldb = he; p.O = de; p.KA = 2 * T2 * ef; p.TO = T2; p.dst = reinterpret_cast<__bf16*>(base);
    hipLaunchKernelGGL(k_pack_pair_bf16, dim3(T1 * (p.KA + 2 * p.TO)), dim3(64), 0, s, p);
    p = {};
    p.Wa = w_cls0; p.lda = de; p.a_col0 = 0; p.seg_real = de; p.seg_pad = 32 * T2; p.nseg = 1; p.H = hc;
    p.Wb = nullptr; p.ldb = 0; p.O = 0; p.KA = 2 * T2; p.TO = 0; p.dst = reinterpret_cast<__bf16*>(base + oc);
    hipLaunchKernelGGL(k_pack_pair_bf16, dim3(TC * p.KA), dim3(64), 0, s, p);
    for (int q = 0; q < 2; ++q) {
        p = {};
        p.Wa = w_flow0[q]; p.lda = ld_flow0; p.a_col0 = col0_flow; p.seg_real = de; p.seg_pad = 32 * T2; p.nseg = 1; p.H = hn;
        p.Wb = w_flow1[q]; p.ldb = hn; p.O = dn; p.KA = 2 * T2; p.TO = TD; p.dst = reinterpret_cast<__bf16*>(base + (q == 0 ? of0 : of1));
        hipLaunchKernelGGL(k_pack_pair_bf16, dim3(TF * (p.KA + 2 * p.TO)), dim3(64), 0, s, p);
    }
    MPN_LAUNCH_CHECK();
    return MPNHIP_OK;
}

int launch_edge_chain_bf16(const EdgeChainBf16Args& a_in, hipStream_t s) {
    if (a_in.E <= 0) return MPNHIP_OK;
    EdgeChainBf16Args a = a_in;
    if (const char* e = getenv("MPNHIP_CHAIN_BF16_PLAIN_BARRIERS")) a.plain_barriers = e[0] == '1' ? 1 : 0;
    if ((int64_t)a.E * bmax(bmax(a.he, a.dn), a.de) >= ((int64_t)1 << 32) || (int64_t)a.N * a.pw >= ((int64_t)1 << 32)) {
        set_error("edge_chain_bf16: graph too large for 32-bit row offsets");
        return MPNHIP_ERR_UNSUPPORTED;
    }
    // 256-d: 4-wave blocks, two per CU (cfg-E: 643 -> 581 us per launch; A-B switch MPNHIP_CHAIN_BF16_NW=8 for one 8-wave block)
    static const int nw_env = [] { const char* e = getenv("MPNHIP_CHAIN_BF16_NW"); return e ? atoi(e) : 0; }();
    const int variant = chain_bf16_variant(a.he, a.de, a.hn, a.dn, a.hc);
    const bool four = variant == 256 && nw_env != 8;
    const int epb = four ? 128 : 256;
    const unsigned blocks = (unsigned)((a.E + epb - 1) / epb + 3);
    count_path(PC_CHAIN_FWD_BF16);
    const bool exact = a.he % 32 == 0 && a.de % 32 == 0 && a.hn % 32 == 0 && a.dn % 32 == 0;
    switch (variant) {
        case 256:
            // (widths that are multiples of 32 only: the masked form of this variant does not fit the register budget)
            if (four) MPN_LAUNCH_PROFILED((edge_chain_bf16_kernel<20, 4, 14, 8, 2, 2, true, 4, 1>), dim3(blocks), dim3(256), s, a);
            else MPN_LAUNCH_PROFILED((edge_chain_bf16_kernel<20, 4, 14, 8, 2, 2, true>), dim3(blocks), dim3(512), s, a);
            break;
        case 128:
            if (exact) MPN_LAUNCH_PROFILED((edge_chain_bf16_kernel<10, 2, 7, 4, 1, 2, true>), dim3(blocks), dim3(512), s, a);
            else MPN_LAUNCH_PROFILED((edge_chain_bf16_kernel<10, 2, 7, 4, 1, 2, false>), dim3(blocks), dim3(512), s, a);
            break;
        case 64: MPN_LAUNCH_PROFILED((edge_chain_bf16_kernel<5, 1, 4, 2, 1, 2, false>), dim3(blocks), dim3(512), s, a); break;
        case 32: MPN_LAUNCH_PROFILED((edge_chain_bf16_kernel<3, 1, 2, 1, 1, 2, false>), dim3(blocks), dim3(512), s, a); break;
        default: set_error("edge_chain_bf16: unsupported widths"); return MPNHIP_ERR_UNSUPPORTED;
    }
    MPN_LAUNCH_CHECK();
    if (a.agg_out) {
        hipLaunchKernelGGL(k_agg_fixup, dim3(blocks * (four ? 4 : 8)), dim3(64), 0, s, a.header, a.seg_ptr, a.start_row, a.piece, a.agg_out,
                           a.N, a.dn, (a.dn + 31) / 32 * 32, a.agg, four ? 4 : 8);
        MPN_LAUNCH_CHECK();
    }
    return MPNHIP_OK;
}

size_t chain_bf16_agg_scratch_floats(int64_t E, int dn, size_t* off_start_row) {
    const size_t tiles = (size_t)((E + 255) / 256 + 3) * 8, DN = (size_t)(dn + 31) / 32 * 32;
    if (off_start_row) *off_start_row = tiles * 2 * DN;
    return tiles * 2 * DN + tiles;
}

}  // namespace mpnhip
