// Fused per-edge chain of one message-passing step (forward):
//   EdgeModel   (reference models/mpn.py:67-69)   H1 = relu(W1e [e0|e] + Pr[row] + Pc[col]);  e' = relu(W2 H1 + b2)
//   classifier  (mpn.py:377 -> :114)              logit = wc2 . relu(Wc1 e' + bc1) + bc2
//   flow MLPs   (mpn.py:85-94, per direction)     M = relu(Wf2 relu(Wfe e' + Pf[col]) + bf2)
// in ONE kernel, so that H1 / HC / HF never round-trip through HBM in inference and the five GEMM
// prologues / epilogues per step collapse into one.
//
// Formulation: every product is computed TRANSPOSED, D^T[n][edge] = W[n][k] X^T[k][edge], with the 32
// edges of a wave on the MFMA's lane (j) dimension and the output features in the accumulator registers:
//   v_mfma_f32_32x32x2_f32:  A = weights (lane (i, h) supplies W[n0 + i][k_h]),
//                            B = activations (lane (j, h) supplies X[k_h][edge j]),
//                            D: lane (j, h), register r holds D[n0 + (r&3) + 8(r>>2) + 4h][edge j].
// The accumulator tile of one layer (after bias / ReLU in place) IS the B operand of the next layer:
// MFMA number r of source tile t contracts k = 32t + (r&3) + 8(r>>2) + 4h -- no data movement, only the
// weight fetch follows that k order.  First-layer inputs come straight from global memory (each lane reads
// its own edge's row, 16 bytes at a time); gathered per-node projections initialise the accumulators (C-in).
// Only the WEIGHTS go through LDS: pre-transposed [k][n] images are streamed in <= 20 KB chunks, double
// buffered, one barrier per chunk, shared by the block's four waves (128 edges of one direction).
//
// Instantiated for the BASELINE.json 128-d configuration (he 320, de 64, hn 224, dn 128, hc 32); other
// widths use the unfused GEMM path.
#include "common.h"
#include "edge_chain.h"

namespace mpnhip {

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int CH_FLOATS = 5120;  // floats per weight chunk buffer (20 KB)

__device__ __forceinline__ float4 ldg4(const float* p) { return *reinterpret_cast<const float4*>(p); }

struct ChunkDesc {
    const float* w;  // pre-transposed weight image WT[k][n], leading dim ldw
    int ldw, k0, kc, n0, nc;
};

// register-staged copy of one chunk: up to 5 float4 per thread
struct ChunkRegs {
    float4 v[5];
};

__device__ __forceinline__ void chunk_load(const ChunkDesc& d, int tid, ChunkRegs& r) {
    const int nc4 = d.nc >> 2;
    const int total = d.kc * nc4;
#pragma unroll
    for (int q = 0; q < 5; ++q) {
        int f = tid + 256 * q;
        f = f < total ? f : total - 1;  // clamped (unconditional loads); surplus copies are not written
        const int kr = f / nc4, c4 = f - kr * nc4;
        r.v[q] = ldg4(d.w + (int64_t)(d.k0 + kr) * d.ldw + d.n0 + 4 * c4);
    }
}

__device__ __forceinline__ void chunk_store(const ChunkDesc& d, int tid, const ChunkRegs& r, float* buf) {
    const int nc4 = d.nc >> 2;
    const int total = d.kc * nc4;
#pragma unroll
    for (int q = 0; q < 5; ++q) {
        const int f = tid + 256 * q;
        if (f < total) *reinterpret_cast<float4*>(buf + 4 * f) = r.v[q];  // image [kc][nc], pitch nc
    }
}

__device__ __forceinline__ void relu16(f32x16& a) {
#pragma unroll
    for (int r = 0; r < 16; ++r) a[r] = fmaxf(a[r], 0.f);
}

// D tile rows of register group g (registers 4g..4g+3): n = 8g + 4h + (0..3)
__device__ __forceinline__ void set4(f32x16& a, int g, float4 v) {
    a[4 * g + 0] = v.x; a[4 * g + 1] = v.y; a[4 * g + 2] = v.z; a[4 * g + 3] = v.w;
}
__device__ __forceinline__ void add4(f32x16& a, int g, float4 v) {
    a[4 * g + 0] += v.x; a[4 * g + 1] += v.y; a[4 * g + 2] += v.z; a[4 * g + 3] += v.w;
}
__device__ __forceinline__ float4 get4(const f32x16& a, int g) {
    return make_float4(a[4 * g + 0], a[4 * g + 1], a[4 * g + 2], a[4 * g + 3]);
}

// One chained product step: `src` (a 32-feature accumulator tile, already activated) is the B operand for
// the TOUT output tiles whose weights sit in the chunk image `ws` ([kc][nc], this source tile at rows
// krow0 .. krow0+31, output tile t at columns ncol0 + 32 t).  lane_off = 4h * nc + i.
template <int TOUT>
__device__ __forceinline__ void chain_tile(const f32x16& src, f32x16* out, const float* ws, int nc, int krow0, int ncol0,
                                           int lane_off) {
    // weight fetch runs two k pairs ahead of the MFMAs (pinned with sched_barrier: hipcc would otherwise sink
    // every ds_read to just before its MFMA and wait lgkmcnt(0) there)
    const float* base = ws + lane_off + ncol0;
    float a[3][TOUT];
    auto fetch = [&](int r, float* dst) {
        const int krow = krow0 + (r & 3) + 8 * (r >> 2);
#pragma unroll
        for (int t = 0; t < TOUT; ++t) dst[t] = base[krow * nc + 32 * t];
    };
    fetch(0, a[0]);
    fetch(1, a[1]);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        if (r + 2 < 16) fetch(r + 2, a[(r + 2) % 3]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < TOUT; ++t) out[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[r % 3][t], src[r], out[t], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// Half a source tile: registers r0 .. r0+7 of `src` (contraction indices 2 r0 .. 2 r0 + 15 of its 32) against a
// chunk that holds exactly those 16 rows.
template <int TOUT>
__device__ __forceinline__ void chain_half(const f32x16& src, int r0, f32x16* out, const float* ws, int nc, int lane_off) {
    // wide outputs: TOUT MFMAs per step already cover the LDS latency -> fetch one step ahead (fewer registers)
    constexpr int DEPTH = TOUT >= 5 ? 1 : 2;
    const float* base = ws + lane_off;
    float a[DEPTH + 1][TOUT];
    auto fetch = [&](int q, float* dst) {
        const int krow = (q & 3) + 8 * (q >> 2);  // q = 0..7 -> chunk rows 0-3, 8-11 (+4h through lane_off)
#pragma unroll
        for (int t = 0; t < TOUT; ++t) dst[t] = base[krow * nc + 32 * t];
    };
#pragma unroll
    for (int q = 0; q < DEPTH; ++q) fetch(q, a[q]);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        if (q + DEPTH < 8) fetch(q + DEPTH, a[(q + DEPTH) % (DEPTH + 1)]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < TOUT; ++t)
            out[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q % (DEPTH + 1)][t], r0 == 0 ? src[q] : src[q + 8], out[t], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
}

}  // namespace

// T1 = he/32, T2 = de/32, TF = hn/32, TD = dn/32 (hc = 32)
template <int T1, int T2, int TF, int TD>
__global__ __launch_bounds__(256, 2) void edge_chain_kernel(EdgeChainArgs A) {
    constexpr int HE = 32 * T1, DE = 32 * T2, HN = 32 * TF, DN = 32 * TD, HC = 32;
    constexpr int KC1 = 16;                       // phase-1 chunk: [16 k][HE]   (HE * 16 <= 5120)
    constexpr int KC2 = 64;                       // phase-2 chunk: [64 k][DE]
    constexpr int NC4 = 64;                       // phase-4 chunk: [DE k][64 n]
    constexpr int KC5 = 32;                       // phase-5 chunk: [32 k][DN]
    static_assert(HE * KC1 <= CH_FLOATS && KC2 * DE <= CH_FLOATS && DE * NC4 <= CH_FLOATS && KC5 * DN <= CH_FLOATS &&
                      DE * HC <= CH_FLOATS, "chunk too large");
    static_assert(HE % KC2 == 0 && DE % 32 == 0 && T1 % 2 == 0, "dims");

    __shared__ __attribute__((aligned(16))) float wbuf[2][CH_FLOATS];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lj = lane & 31, lh = lane >> 5;

    // ---- which direction group / which 128 edges ---------------------------------------------------
    const int e_out = A.header[1], e_in = A.header[2];
    const int E = A.E;
    int grp, beg, end, blk = blockIdx.x;
    {
        const int nb0 = (e_out + 127) >> 7, nb1 = (e_in + 127) >> 7;
        if (blk < nb0) { grp = 0; beg = 0; end = e_out; }
        else if (blk < nb0 + nb1) { grp = 1; blk -= nb0; beg = e_out; end = e_out + e_in; }
        else { grp = 2; blk -= nb0 + nb1; beg = e_out + e_in; end = E; }
    }
    const int tile0 = beg + blk * 128;
    if (tile0 >= end) return;
    const int edge_raw = tile0 + wave * 32 + lj;
    const bool edge_ok = edge_raw < end;
    const int edge = edge_ok ? edge_raw : end - 1;
    const int K1 = A.k1a + A.k1b;                 // columns of [e0 | e]
    const int nch1 = K1 / KC1;
    const bool flow = grp < 2;

    // ---- chunk schedule ----------------------------------------------------------------------------
    constexpr int NCH2 = T2 * 0 + HE / KC2;       // phase 2: HE / 64 chunks, all DE columns each
    constexpr int NCH4 = (HN + NC4 - 1) / NC4;
    constexpr int NCH5 = HN / KC5;
    const int c2 = nch1, c3 = c2 + NCH2, c4 = c3 + 1, c5 = c4 + NCH4, cend_flow = c5 + NCH5;
    const int nchunks = flow ? cend_flow : c4;
    const float* wf1 = grp == 1 ? A.wf1T_in : A.wf1T_out;
    const float* wf2 = grp == 1 ? A.wf2T_in : A.wf2T_out;
    auto desc = [&](int c) {
        ChunkDesc d;
        if (c < c2) { d.w = A.w1T; d.ldw = HE; d.k0 = c * KC1; d.kc = KC1; d.n0 = 0; d.nc = HE; }
        else if (c < c3) { d.w = A.w2T; d.ldw = DE; d.k0 = (c - c2) * KC2; d.kc = KC2; d.n0 = 0; d.nc = DE; }
        else if (c < c4) { d.w = A.wc1T; d.ldw = HC; d.k0 = 0; d.kc = DE; d.n0 = 0; d.nc = HC; }
        else if (c < c5) { d.w = wf1; d.ldw = HN; d.k0 = 0; d.kc = DE; d.n0 = (c - c4) * NC4;
                           d.nc = HN - d.n0 < NC4 ? HN - d.n0 : NC4; }
        else { d.w = wf2; d.ldw = DN; d.k0 = (c - c5) * KC5; d.kc = KC5; d.n0 = 0; d.nc = DN; }
        return d;
    };

    ChunkRegs creg;
    int c = 0;  // chunk being computed
    {
        ChunkDesc d0 = desc(0);
        chunk_load(d0, tid, creg);
        chunk_store(d0, tid, creg, wbuf[0]);
    }
    // prefetch / commit of the NEXT chunk around the compute of chunk c
    auto prefetch = [&]() { if (c + 1 < nchunks) { ChunkDesc d = desc(c + 1); chunk_load(d, tid, creg); } };
    auto commit = [&]() {
        if (c + 1 < nchunks) { ChunkDesc d = desc(c + 1); chunk_store(d, tid, creg, wbuf[(c + 1) & 1]); }
        __syncthreads();
        ++c;
    };

    // ---- phase 1: H1^T = W1e [e0|e]^T, C-in = Pr[row] + Pc[col] --------------------------------------
    const int row = A.srow[edge], col = A.scol[edge];
    f32x16 h1[T1];
    {
        // C-in = Pr[row]: loaded straight into the accumulators (Pc[col] is added after the MFMAs, when the
        // weight / input staging registers are free again)
        const float* pr = A.P + (int64_t)row * A.pw + 4 * lh;
#pragma unroll
        for (int t = 0; t < T1; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) set4(h1[t], g, ldg4(pr + 32 * t + 8 * g));
    }
    {
        // lane (j, h) reads its edge's features 16 bytes at a time: k = 8u + 4h + (0..3)
        const float* xa = A.xa + (int64_t)edge * A.ldxa + 4 * lh;
        const float* xb = A.xb ? A.xb + (int64_t)edge * A.ldxb + 4 * lh - A.k1a : xa;
        auto xload = [&](int k) { return ldg4((k < A.k1a ? xa : xb) + k); };
        float4 xcur[2], xnxt[2];
        xcur[0] = xload(0);
        xcur[1] = xload(8);
        __syncthreads();  // chunk 0 is in wbuf[0]
        for (int i = 0; i < nch1; ++i) {
            prefetch();
            if (i + 1 < nch1) { xnxt[0] = xload((i + 1) * KC1); xnxt[1] = xload((i + 1) * KC1 + 8); }
            const float* ws = wbuf[c & 1] + 4 * lh * HE + lj;
            // 8 steps (u, q) of T1 MFMAs each; the weights of step s+1 are fetched before the MFMAs of step s
            float a[2][T1];
#pragma unroll
            for (int t = 0; t < T1; ++t) a[0][t] = ws[32 * t];
#pragma unroll
            for (int st = 0; st < 8; ++st) {
                const int u = st >> 2, q = st & 3;
                if (st + 1 < 8) {
                    const int u1 = (st + 1) >> 2, q1 = (st + 1) & 3;
#pragma unroll
                    for (int t = 0; t < T1; ++t) a[(st + 1) & 1][t] = ws[(8 * u1 + q1) * HE + 32 * t];
                }
                const float xv = q == 0 ? xcur[u].x : (q == 1 ? xcur[u].y : (q == 2 ? xcur[u].z : xcur[u].w));
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int t = 0; t < T1; ++t) h1[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[st & 1][t], xv, h1[t], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (i + 1 < nch1) { xcur[0] = xnxt[0]; xcur[1] = xnxt[1]; }
            commit();
        }
    }
    {
        const float* pc = A.P + (int64_t)col * A.pw + HE + 4 * lh;
#pragma unroll
        for (int t = 0; t < T1; t += 2) {
            __builtin_amdgcn_sched_barrier(0);  // two tiles (8 row pieces) of gathers in flight at a time
            float4 v[8];
#pragma unroll
            for (int g = 0; g < 8; ++g) v[g] = ldg4(pc + 32 * (t + (g >> 2)) + 8 * (g & 3));
#pragma unroll
            for (int g = 0; g < 8; ++g) add4(h1[t + (g >> 2)], g & 3, v[g]);
            relu16(h1[t]);
            relu16(h1[t + 1]);
        }
    }
    if (A.save_h1 && edge_ok) {
        float* o = A.save_h1 + (int64_t)edge * HE + 4 * lh;
#pragma unroll
        for (int t = 0; t < T1; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) *reinterpret_cast<float4*>(o + 32 * t + 8 * g) = get4(h1[t], g);
    }

    // ---- phase 2: e'^T = relu(W2 H1^T + b2) -----------------------------------------------------------
    f32x16 en[T2];
#pragma unroll
    for (int t = 0; t < T2; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) set4(en[t], g, ldg4(A.b2 + 32 * t + 8 * g + 4 * lh));
#pragma unroll
    for (int i = 0; i < NCH2; ++i) {
        prefetch();
        const float* ws = wbuf[c & 1];
        chain_tile<T2>(h1[2 * i], en, ws, DE, 0, 0, 4 * lh * DE + lj);
        chain_tile<T2>(h1[2 * i + 1], en, ws, DE, 32, 0, 4 * lh * DE + lj);
        commit();
    }
#pragma unroll
    for (int t = 0; t < T2; ++t) relu16(en[t]);
    if (edge_ok) {
        float* o = A.e_new + (int64_t)edge * DE + 4 * lh;
#pragma unroll
        for (int t = 0; t < T2; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) *reinterpret_cast<float4*>(o + 32 * t + 8 * g) = get4(en[t], g);
    }

    // ---- phase 3: classifier ----------------------------------------------------------------------------
    {
        f32x16 hc;
#pragma unroll
        for (int g = 0; g < 4; ++g) set4(hc, g, ldg4(A.bc1 + 8 * g + 4 * lh));
        prefetch();
        {
            const float* ws = wbuf[c & 1];
#pragma unroll
            for (int t = 0; t < T2; ++t) chain_tile<1>(en[t], &hc, ws, HC, 32 * t, 0, 4 * lh * HC + lj);
        }
        commit();
        relu16(hc);
        if (A.save_hc && edge_ok) {
            float* o = A.save_hc + (int64_t)edge * HC + 4 * lh;
#pragma unroll
            for (int g = 0; g < 4; ++g) *reinterpret_cast<float4*>(o + 8 * g) = get4(hc, g);
        }
        float part = 0.f;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 w = ldg4(A.wc2 + 8 * g + 4 * lh);
            part = fmaf(w.x, hc[4 * g + 0], part);
            part = fmaf(w.y, hc[4 * g + 1], part);
            part = fmaf(w.z, hc[4 * g + 2], part);
            part = fmaf(w.w, hc[4 * g + 3], part);
        }
        const float other = __shfl_xor(part, 32, 64);
        if (A.logits && edge_ok && lh == 0) A.logits[A.perm[edge]] = part + other + A.bc2[0];
    }
    if (!flow) return;  // self loops take part in the edge update only (mpn.py:85,91)

    // ---- phase 4: HF^T = relu(Wfe e'^T + Pf[col]) ---------------------------------------------------------
    f32x16 hf[TF];
    {
        const float* pf = A.P + (int64_t)col * A.pw + 2 * HE + grp * HN + 4 * lh;
#pragma unroll
        for (int t = 0; t < TF; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) set4(hf[t], g, ldg4(pf + 32 * t + 8 * g));
    }
#pragma unroll
    for (int i = 0; i < NCH4; ++i) {
        prefetch();
        const float* ws = wbuf[c & 1];
        constexpr int full = NC4 / 32;
        const int ncw = (HN - i * NC4) < NC4 ? (HN - i * NC4) : NC4;  // compile-time per unrolled i
#pragma unroll
        for (int t = 0; t < T2; ++t) {
            if (ncw == NC4) chain_tile<full>(en[t], &hf[i * full], ws, NC4, 32 * t, 0, 4 * lh * NC4 + lj);
            else chain_tile<1>(en[t], &hf[i * full], ws, 32, 32 * t, 0, 4 * lh * 32 + lj);
        }
        commit();
    }
#pragma unroll
    for (int t = 0; t < TF; ++t) relu16(hf[t]);
    if (A.save_hf && edge_ok) {
        float* o = A.save_hf + (int64_t)edge * HN + 4 * lh;
#pragma unroll
        for (int t = 0; t < TF; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) *reinterpret_cast<float4*>(o + 32 * t + 8 * g) = get4(hf[t], g);
    }

    // ---- phase 5: M^T = relu(Wf2 HF^T + bf2) -----------------------------------------------------------------
    f32x16 mm[TD];
    {
        const float* bf2 = (grp == 1 ? A.bf2_in : A.bf2_out) + 4 * lh;
#pragma unroll
        for (int t = 0; t < TD; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) set4(mm[t], g, ldg4(bf2 + 32 * t + 8 * g));
    }
#pragma unroll
    for (int i = 0; i < NCH5; ++i) {
        prefetch();
        const float* ws = wbuf[c & 1];
        chain_tile<TD>(hf[i], mm, ws, DN, 0, 0, 4 * lh * DN + lj);
        commit();
    }
    if (edge_ok) {
        float* o = A.msg + (int64_t)edge * DN + 4 * lh;
#pragma unroll
        for (int t = 0; t < TD; ++t) {
            relu16(mm[t]);
#pragma unroll
            for (int g = 0; g < 4; ++g) *reinterpret_cast<float4*>(o + 32 * t + 8 * g) = get4(mm[t], g);
        }
    }
}


// ------------------------------------------------------------------------------------------------------
// Backward chain of one step.  Same machinery, transposed weights: dH^T[k][edge] = sum_n W[n][k] dZ^T[n][edge],
// so the LDS chunk image is W in its native [n][k] layout (rows = contraction index).
//   B1  dZM = gather(dAGG)[row] (.) [M > 0]                      (node_agg_fn backward, mpn.py:89,96)
//   B2  dZF = (Wf2^T dZM) (.) [HF > 0]
//   B3  dE' = dE_in + Wfe^T dZF
//   B4  dZc = (dlog wc2) (.) [HC > 0];  dE' += Wc1^T dZc;  dZ2 = dE' (.) [e_s > 0]
//   B5  dZ1 = (W2^T dZ2) (.) [H1 > 0]
//   B6  d[e0 | e_{s-1}] = W1e^T dZ1  ->  dE0 += ..., dEprev = ...
template <int T1, int T2, int TF, int TD>
__global__ __launch_bounds__(256, 2) void edge_chain_bwd_kernel(EdgeChainBwdArgs A) {
    constexpr int HE = 32 * T1, DE = 32 * T2, HN = 32 * TF, DN = 32 * TD, HC = 32;
    constexpr int NR2 = 16;   // B2 chunk: [16 n][HN]
    constexpr int NR3 = 64;   // B3 chunk: [64 n][DE]
    constexpr int NR5 = 16;   // B5 chunk: [16 n][HE]
    constexpr int NR6 = 64;   // B6 chunk: [64 n][64 k]
    static_assert(NR2 * HN <= CH_FLOATS && NR3 * DE <= CH_FLOATS && NR5 * HE <= CH_FLOATS && NR6 * 64 <= CH_FLOATS, "chunk");
    static_assert(DN % NR2 == 0 && DE % NR5 == 0 && HE % NR6 == 0, "dims");

    __shared__ __attribute__((aligned(16))) float wbuf[2][CH_FLOATS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lj = lane & 31, lh = lane >> 5;

    const int e_out = A.header[1], e_in = A.header[2];
    int grp, beg, end, blk = blockIdx.x;
    {
        const int nb0 = (e_out + 127) >> 7, nb1 = (e_in + 127) >> 7;
        if (blk < nb0) { grp = 0; beg = 0; end = e_out; }
        else if (blk < nb0 + nb1) { grp = 1; blk -= nb0; beg = e_out; end = e_out + e_in; }
        else { grp = 2; blk -= nb0 + nb1; beg = e_out + e_in; end = A.E; }
    }
    const int tile0 = beg + blk * 128;
    if (tile0 >= end) return;
    const int edge_raw = tile0 + wave * 32 + lj;
    const bool edge_ok = edge_raw < end;
    const int edge = edge_ok ? edge_raw : end - 1;
    const bool flow = grp < 2;
    const int KE = A.cat_two ? 2 * DE : DE;   // columns of [e0 | e_{s-1}]
    const int npass6 = KE / 64;               // B6 passes of 64 output columns

    // ---- chunk schedule: [B2 | B3] (flow groups only) B4 B5 B6 ------------------------------------------
    constexpr int NCH2 = DN / NR2, NCH3 = (HN + NR3 - 1) / NR3, NCH5 = DE / NR5, NCH6 = HE / NR6;
    const int c3 = flow ? NCH2 : 0, c4 = c3 + (flow ? NCH3 : 0), c5 = c4 + 1, c6 = c5 + NCH5;
    const int nchunks = c6 + npass6 * NCH6;
    const float* wf2 = grp == 1 ? A.wf2_in : A.wf2_out;
    const float* wfe = grp == 1 ? A.wfe_in : A.wfe_out;
    auto desc = [&](int c) {
        ChunkDesc d;
        if (c < c3) { d.w = wf2; d.ldw = HN; d.k0 = c * NR2; d.kc = NR2; d.n0 = 0; d.nc = HN; }
        else if (c < c4) { d.w = wfe; d.ldw = A.ldwfe; d.k0 = (c - c3) * NR3; d.kc = HN - d.k0 < NR3 ? HN - d.k0 : NR3; d.n0 = 0; d.nc = DE; }
        else if (c < c5) { d.w = A.wc1; d.ldw = DE; d.k0 = 0; d.kc = HC; d.n0 = 0; d.nc = DE; }
        else if (c < c6) { d.w = A.w2; d.ldw = HE; d.k0 = (c - c5) * NR5; d.kc = NR5; d.n0 = 0; d.nc = HE; }
        else { const int q = c - c6; d.w = A.w1e; d.ldw = A.ldw1e; d.k0 = (q % NCH6) * NR6; d.kc = NR6; d.n0 = (q / NCH6) * 64; d.nc = 64; }
        return d;
    };
    ChunkRegs creg;
    int c = 0;
    {
        ChunkDesc d0 = desc(0);
        chunk_load(d0, tid, creg);
        chunk_store(d0, tid, creg, wbuf[0]);
    }
    auto prefetch = [&]() { if (c + 1 < nchunks) { ChunkDesc d = desc(c + 1); chunk_load(d, tid, creg); } };
    auto commit = [&]() {
        if (c + 1 < nchunks) { ChunkDesc d = desc(c + 1); chunk_store(d, tid, creg, wbuf[(c + 1) & 1]); }
        __syncthreads();
        ++c;
    };

    // gradient w.r.t. e_s arriving from the later step: C-in of the dE' accumulators (loaded after B2, when the
    // dZM tiles are dead -- B2 is the register peak of this kernel)
    f32x16 de[T2];
    auto load_de = [&]() {
        const float* p = A.dE_io + (int64_t)edge * DE + 4 * lh;
#pragma unroll
        for (int t = 0; t < T2; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) set4(de[t], g, ldg4(p + 32 * t + 8 * g));
    };
    __syncthreads();  // chunk 0 is in wbuf[0]

    if (flow) {
        // ---- B1: dZM ------------------------------------------------------------------------------------
        f32x16 dzm[TD];
        {
            const int row = A.srow[edge];
            const int64_t o = (int64_t)row * 2 * DN + (grp == 0 ? DN : 0) + 4 * lh;
            float scale = 1.f;
            if (A.agg == MPNHIP_AGG_MEAN) {
                const int key = grp * A.N + row;
                const int cnt = A.seg_ptr[key + 1] - A.seg_ptr[key];
                scale = 1.f / 1.f;  // (placeholder keeps the division below exact: v / cnt, as the reference divides)
                scale = (float)(cnt > 0 ? cnt : 1);
            }
            const float* mp = A.M + (int64_t)edge * DN + 4 * lh;
#pragma unroll
            for (int t = 0; t < TD; ++t)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float4 v = ldg4(A.dAGG + o + 32 * t + 8 * g);
                    const float4 m = ldg4(mp + 32 * t + 8 * g);
                    if (A.agg == MPNHIP_AGG_MEAN) { v.x /= scale; v.y /= scale; v.z /= scale; v.w /= scale; }
                    if (A.agg == MPNHIP_AGG_MAX) {
                        const int4 a = *reinterpret_cast<const int4*>(A.ARG + o + 32 * t + 8 * g);
                        v.x = a.x == edge_raw ? v.x : 0.f; v.y = a.y == edge_raw ? v.y : 0.f;
                        v.z = a.z == edge_raw ? v.z : 0.f; v.w = a.w == edge_raw ? v.w : 0.f;
                    }
                    v.x = m.x > 0.f ? v.x : 0.f; v.y = m.y > 0.f ? v.y : 0.f;
                    v.z = m.z > 0.f ? v.z : 0.f; v.w = m.w > 0.f ? v.w : 0.f;
                    set4(dzm[t], g, v);
                }
            if (edge_ok) {
                float* o2 = A.dZM + (int64_t)edge * DN + 4 * lh;
#pragma unroll
                for (int t = 0; t < TD; ++t)
#pragma unroll
                    for (int g = 0; g < 4; ++g) *reinterpret_cast<float4*>(o2 + 32 * t + 8 * g) = get4(dzm[t], g);
            }
        }
        // ---- B2: dZF = (Wf2^T dZM) (.) [HF > 0] -------------------------------------------------------------
        f32x16 dzf[TF];
#pragma unroll
        for (int t = 0; t < TF; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) dzf[t][r] = 0.f;
#pragma unroll
        for (int i = 0; i < NCH2; ++i) {
            prefetch();
            // chunk rows = 16 contraction indices n = 16 i .. 16 i + 15 = registers 8 (i & 1) .. + 7 of source tile i / 2
            chain_half<TF>(dzm[i >> 1], (i & 1) * 8, dzf, wbuf[c & 1], HN, 4 * lh * HN + lj);
            commit();
        }
        {
            const float* hp = A.HF + (int64_t)edge * HN + 4 * lh;
            float* o2 = A.dZF + (int64_t)edge * HN + 4 * lh;
#pragma unroll
            for (int t = 0; t < TF; ++t) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 m = ldg4(hp + 32 * t + 8 * g);
                    float4 v = get4(dzf[t], g);
                    v.x = m.x > 0.f ? v.x : 0.f; v.y = m.y > 0.f ? v.y : 0.f;
                    v.z = m.z > 0.f ? v.z : 0.f; v.w = m.w > 0.f ? v.w : 0.f;
                    set4(dzf[t], g, v);
                    if (edge_ok) *reinterpret_cast<float4*>(o2 + 32 * t + 8 * g) = v;
                }
            }
        }
        // ---- B3: dE' += Wfe^T dZF -------------------------------------------------------------------------------
        load_de();
#pragma unroll
        for (int i = 0; i < NCH3; ++i) {
            prefetch();
            const float* ws = wbuf[c & 1];
            chain_tile<T2>(dzf[2 * i], de, ws, DE, 0, 0, 4 * lh * DE + lj);
            if (2 * i + 1 < TF) chain_tile<T2>(dzf[2 * i + 1], de, ws, DE, 32, 0, 4 * lh * DE + lj);
            commit();
        }
    }

    if (!flow) load_de();
    // ---- B4: classifier ---------------------------------------------------------------------------------------
    {
        f32x16 dzc;
        const float dl = A.dlog[A.perm[edge]];
        const float* hp = A.HC + (int64_t)edge * HC + 4 * lh;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 w = ldg4(A.wc2 + 8 * g + 4 * lh);
            const float4 m = ldg4(hp + 8 * g);
            float4 v;
            v.x = m.x > 0.f ? dl * w.x : 0.f; v.y = m.y > 0.f ? dl * w.y : 0.f;
            v.z = m.z > 0.f ? dl * w.z : 0.f; v.w = m.w > 0.f ? dl * w.w : 0.f;
            set4(dzc, g, v);
            if (edge_ok) *reinterpret_cast<float4*>(A.dZc + (int64_t)edge * HC + 4 * lh + 8 * g) = v;
        }
        prefetch();
        chain_tile<T2>(dzc, de, wbuf[c & 1], DE, 0, 0, 4 * lh * DE + lj);
        commit();
    }
    // dZ2 = dE' (.) [e_s > 0]  (written over the incoming gradient)
    {
        const float* ep = A.e_s + (int64_t)edge * DE + 4 * lh;
        float* o2 = A.dE_io + (int64_t)edge * DE + 4 * lh;
#pragma unroll
        for (int t = 0; t < T2; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 m = ldg4(ep + 32 * t + 8 * g);
                float4 v = get4(de[t], g);
                v.x = m.x > 0.f ? v.x : 0.f; v.y = m.y > 0.f ? v.y : 0.f;
                v.z = m.z > 0.f ? v.z : 0.f; v.w = m.w > 0.f ? v.w : 0.f;
                set4(de[t], g, v);
                if (edge_ok) *reinterpret_cast<float4*>(o2 + 32 * t + 8 * g) = v;
            }
    }

    // ---- B5: dZ1 = (W2^T dZ2) (.) [H1 > 0] -----------------------------------------------------------------------
    f32x16 dz1[T1];
#pragma unroll
    for (int t = 0; t < T1; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) dz1[t][r] = 0.f;
#pragma unroll
    for (int i = 0; i < NCH5; ++i) {
        prefetch();
        // two half-width sweeps over the same chunk keep the weight staging registers at T1 / 2 per step
        chain_half<T1 / 2>(de[i >> 1], (i & 1) * 8, dz1, wbuf[c & 1], HE, 4 * lh * HE + lj);
        chain_half<T1 / 2>(de[i >> 1], (i & 1) * 8, dz1 + T1 / 2, wbuf[c & 1], HE, 4 * lh * HE + lj + 16 * T1);
        commit();
    }
    {
        const float* hp = A.H1 + (int64_t)edge * HE + 4 * lh;
        float* o2 = A.dZ1 + (int64_t)edge * HE + 4 * lh;
#pragma unroll
        for (int t = 0; t < T1; t += 2) {
            __builtin_amdgcn_sched_barrier(0);
            float4 m[8];
#pragma unroll
            for (int g = 0; g < 8; ++g) m[g] = ldg4(hp + 32 * (t + (g >> 2)) + 8 * (g & 3));
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                float4 v = get4(dz1[t + (g >> 2)], g & 3);
                v.x = m[g].x > 0.f ? v.x : 0.f; v.y = m[g].y > 0.f ? v.y : 0.f;
                v.z = m[g].z > 0.f ? v.z : 0.f; v.w = m[g].w > 0.f ? v.w : 0.f;
                set4(dz1[t + (g >> 2)], g & 3, v);
                if (edge_ok) *reinterpret_cast<float4*>(o2 + 32 * (t + (g >> 2)) + 8 * (g & 3)) = v;
            }
        }
    }

    // ---- B6: d[e0 | e_{s-1}] = W1e^T dZ1, 64 output columns per pass ---------------------------------------------------
    for (int pass = 0; pass < npass6; ++pass) {
        f32x16 dc[2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) dc[t][r] = 0.f;
#pragma unroll
        for (int i = 0; i < NCH6; ++i) {
            prefetch();
            const float* ws = wbuf[c & 1];
            chain_tile<2>(dz1[2 * i], dc, ws, 64, 0, 0, 4 * lh * 64 + lj);
            chain_tile<2>(dz1[2 * i + 1], dc, ws, 64, 32, 0, 4 * lh * 64 + lj);
            commit();
        }
        // pass 0 of a two-segment input is the re-attached initial features (accumulated over all steps);
        // the last pass is e_{s-1} -- which IS e0 at the first step
        const bool to_e0 = (A.cat_two && pass == 0) || A.first_step;
        float* dst = (to_e0 ? A.dE0 : A.dEprev) + (int64_t)edge * DE + 4 * lh;
        if (edge_ok) {
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float4 v = get4(dc[t], g);
                    if (to_e0) {
                        const float4 o = ldg4(dst + 32 * t + 8 * g);
                        v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
                    }
                    *reinterpret_cast<float4*>(dst + 32 * t + 8 * g) = v;
                }
        }
    }
}

int launch_edge_chain_bwd(const EdgeChainBwdArgs& a, hipStream_t s) {
    if (a.E <= 0) return MPNHIP_OK;
    const unsigned blocks = (unsigned)((a.E + 127) / 128 + 3);
    hipLaunchKernelGGL((edge_chain_bwd_kernel<10, 2, 7, 4>), dim3(blocks), dim3(256), 0, s, a);
    MPN_LAUNCH_CHECK();
    return MPNHIP_OK;
}

// WT[k][n] = W[n][k0 + k]   (n < n_rows, k < k_cols), W leading dim ldw
__global__ void k_transpose_block(const float* __restrict__ W, int64_t ldw, int k0, int n_rows, int k_cols,
                                  float* __restrict__ WT) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)n_rows * k_cols) return;
    const int k = (int)(i / n_rows), n = (int)(i % n_rows);
    WT[i] = W[(int64_t)n * ldw + k0 + k];
}

int transpose_block(const float* W, int64_t ldw, int k0, int n_rows, int k_cols, float* WT, hipStream_t s) {
    const int64_t n = (int64_t)n_rows * k_cols;
    if (n <= 0) return MPNHIP_OK;
    hipLaunchKernelGGL(k_transpose_block, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, W, ldw, k0, n_rows, k_cols, WT);
    MPN_LAUNCH_CHECK();
    return MPNHIP_OK;
}

bool edge_chain_supported(int he, int de, int hn, int dn, int hc, int k1a, int k1b) {
    return he == 320 && de == 64 && hn == 224 && dn == 128 && hc == 32 && (k1a % 16 == 0) && (k1b % 16 == 0) &&
           (k1a + k1b) >= 16;
}

int launch_edge_chain(const EdgeChainArgs& a, hipStream_t s) {
    if (a.E <= 0) return MPNHIP_OK;
    const unsigned blocks = (unsigned)((a.E + 127) / 128 + 3);
    hipLaunchKernelGGL((edge_chain_kernel<10, 2, 7, 4>), dim3(blocks), dim3(256), 0, s, a);
    MPN_LAUNCH_CHECK();
    return MPNHIP_OK;
}

}  // namespace mpnhip
