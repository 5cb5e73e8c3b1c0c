// The whole message-passing loop of an inference forward at the reference's widths (dn = 32, de = 16, he <= 96, hn <= 64) in ONE
// launch: L x { reattach, MetaLayer (EdgeModel, TimeAwareNodeModel with scatter sum / mean / max), classifier }
// (reference models/mpn.py:366-385, :59-99, :114).  At these sizes (a few hundred nodes, 10^4 - 10^5 edges) a step of the launch-per-
// module path is two dependent kernels of 6-30 us whose weights (29k floats) are streamed again every launch.
//
// Work split: a block owns a contiguous range of NODES and all their edges -- the (direction, row)-sorted edge order makes those
// three contiguous runs (flow_out, flow_in, self loops) -- so a step needs no edge -> node exchange between blocks:
//   * per 32-edge tile (4 waves = 4 tiles per round) the fused per-edge chain on fp32 MFMAs (v_mfma_f32_32x32x2_f32, edges on the
//     lane dimension, a layer's accumulator tile is the next layer's B operand -- edge_chain.hip), weights resident in LDS for the
//     whole launch; e' updated in place, logits written through the sort permutation;
//   * the round's message tiles go to an LDS slab and are added (sum / mean / max) into the block's per-(node, direction)
//     accumulators in edge order (the reference's scatter order);
//   * node update x' = relu(Wu [agg_in | agg_out] + bu) and the NEXT step's projection rows P'[n] = P0[n] + Wx x'[n] for the
//     block's nodes; P is double-buffered in global memory;
//   * ONE grid barrier per step (the other blocks' P' rows are what the next step's edges gather): monotonic counter, agent-scope
//     release before the arrival, acquire after the wait (cdna_hip_programming.md Guideline 16).
// Every block must be resident: the grid is at most one block per CU (the launcher sizes it), LDS ~120 KB per block.
#include <cstdlib>
#include <type_traits>

#include "common.h"

namespace mpnhip {

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(1))) unsigned gu32;

constexpr int PS_NB = 16;       // nodes per block at most (N <= 16 x grid)
constexpr int PS_HE = 96, PS_HN = 64;

__device__ __forceinline__ f32x16 zero16() {
    f32x16 a;
#pragma unroll
    for (int r = 0; r < 16; ++r) a[r] = 0.f;
    return a;
}
__device__ __forceinline__ void relu16p(f32x16& a) {
#pragma unroll
    for (int r = 0; r < 16; ++r) a[r] = fmaxf(a[r], 0.f);
}
// feature index of accumulator register r in lane half h
__device__ __forceinline__ constexpr int fidx(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// C-in of one 32-feature tile from a projection row: columns c0 + 8 g + 4 h + (0..3), zero past `width`
__device__ __forceinline__ void add_row_tile(f32x16& acc, const float* row, int c0, int width, int lh) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int n = c0 + 8 * g + 4 * lh;
        if (n < width) {
            const float4 v = *reinterpret_cast<const float4*>(row + n);
            acc[4 * g + 0] += v.x; acc[4 * g + 1] += v.y; acc[4 * g + 2] += v.z; acc[4 * g + 3] += v.w;
        }
    }
}

__global__ __launch_bounds__(256, 1) void mpn_persist32_kernel(Persist32Args A) {
    extern __shared__ __attribute__((aligned(16))) char ps_lds[];
    float* w1 = reinterpret_cast<float*>(ps_lds);          // [32][96]   edge layer 0, e-part: k = [e0 (16) | e (16)]
    float* w2 = w1 + 32 * PS_HE;                          // [96][32]   edge layer 1
    float* wc1 = w2 + PS_HE * 32;                         // [16][32]   classifier layer 0
    float* wf1 = wc1 + 16 * 32;                           // [2][16][64] flow layer 0, e'-part (0: flow_out, 1: flow_in)
    float* wf2 = wf1 + 2 * 16 * PS_HN;                    // [2][64][32] flow layer 1
    float* wu = wf2 + 2 * PS_HN * 32;                     // [64][32]   node update, transposed
    float* wx = wu + 64 * 32;                             // [32][pw]   projections' current-feature columns, transposed
    float* bias = wx + 32 * A.pw;                         // b2[32] | bc1[32] | wc2[32] | bf2_out[32] | bf2_in[32] | bu[32]
    float* slab = bias + 6 * 32;                          // [4 x 32 edges][36]
    float* nacc = slab + 128 * 36;                        // [4 waves][PS_NB][2][32]   per-wave partial (node, direction) aggregates
    float* xs = nacc + 4 * PS_NB * 64;                    // [PS_NB][32]
    int* lptr = reinterpret_cast<int*>(xs + PS_NB * 32);  // [2][PS_NB + 1] segment offsets of the block's nodes

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lj = lane & 31, lh = lane >> 5;
    const int N = A.N, pw = A.pw, he = A.he, hn = A.hn, hc = A.hc;
    const int64_t E = A.E;

    // ---- weights into LDS (once) ---------------------------------------------------------------------------------------------
    // (global reads run along a weight row -- the contraction index k -- so a wave reads contiguous bytes; the LDS images are [k][n])
    for (int i = tid; i < 32 * PS_HE; i += 256) { const int n = i >> 5, k = i & 31; w1[k * PS_HE + n] = n < he ? A.W1[(int64_t)n * A.ld_w1 + A.col_w1 + k] : 0.f; }
    for (int i = tid; i < PS_HE * 32; i += 256) { const int n = i / PS_HE, k = i - n * PS_HE; w2[k * 32 + n] = (k < he && n < 16) ? A.W2[(int64_t)n * he + k] : 0.f; }
    for (int i = tid; i < 16 * 32; i += 256) { const int n = i >> 4, k = i & 15; wc1[k * 32 + n] = n < hc ? A.Wc1[(int64_t)n * 16 + k] : 0.f; }
    for (int i = tid; i < 2 * 16 * PS_HN; i += 256) {
        const int q = i / (16 * PS_HN), r = i - q * 16 * PS_HN, n = r >> 4, k = r & 15;
        wf1[q * 16 * PS_HN + k * PS_HN + n] = n < hn ? A.Wf1[q][(int64_t)n * A.ld_wf1 + A.col_wf1 + k] : 0.f;
    }
    for (int i = tid; i < 2 * PS_HN * 32; i += 256) {
        const int q = i / (PS_HN * 32), r = i - q * PS_HN * 32, n = r / PS_HN, k = r - n * PS_HN;
        wf2[q * PS_HN * 32 + k * 32 + n] = k < hn ? A.Wf2[q][(int64_t)n * hn + k] : 0.f;
    }
    for (int i = tid; i < 64 * 32; i += 256) { const int n = i >> 6, k = i & 63; wu[k * 32 + n] = A.Wu[(int64_t)n * 64 + k]; }
    for (int i = tid; i < 32 * pw; i += 256) { const int o = i >> 5, k = i & 31; wx[k * pw + o] = A.Wnode[(int64_t)o * 64 + 32 + k]; }
    if (tid < 32) {
        bias[tid] = tid < 16 ? A.b2[tid] : 0.f;
        bias[32 + tid] = tid < hc ? A.bc1[tid] : 0.f;
        bias[64 + tid] = tid < hc ? A.wc2[tid] : 0.f;
        bias[96 + tid] = A.bf2[0][tid];
        bias[128 + tid] = A.bf2[1][tid];
        bias[160 + tid] = A.bu[tid];
    }
    // ---- this block's nodes and edge runs ---------------------------------------------------------------------------------------
    const int n_lo = blockIdx.x * A.nodes_per_block;
    int n_hi = n_lo + A.nodes_per_block;
    n_hi = n_hi < N ? n_hi : N;
    const int nb = n_hi > n_lo ? n_hi - n_lo : 0;
    if (tid < 2 * (PS_NB + 1)) {
        const int d = tid / (PS_NB + 1), i = tid - d * (PS_NB + 1);
        lptr[tid] = A.seg_ptr[d * N + (n_lo + (i < nb ? i : nb) < N ? n_lo + (i < nb ? i : nb) : N)];
    }
    int run0[3], runn[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const int a0 = nb > 0 ? A.seg_ptr[d * N + n_lo] : 0, a1 = nb > 0 ? A.seg_ptr[d * N + n_hi] : 0;
        run0[d] = a0;
        runn[d] = a1 - a0;
    }
    const int t_out = (runn[0] + 31) >> 5, t_in = (runn[1] + 31) >> 5, t_self = (runn[2] + 31) >> 5;
    const int T = t_out + t_in + t_self;
    __syncthreads();
    // ---- the chain's MFMA A operands of THIS lane in registers for the whole launch (one wave per SIMD: 512 registers per lane).
    // An A operand is W[n0 + lane % 32][k(step, lane / 32)]: 200 values per lane cover all five layers -- read from LDS per MFMA
    // they cost a dependent ds_read (hipcc re-used one register pair: read, wait, two MFMAs, read, ...)
    float rw1[16][3], rw2[3][16], rwc1[8], rwf1[2][8][2], rwf2[2][2][16];
#pragma unroll
    for (int j = 0; j < 16; ++j)
#pragma unroll
        for (int t = 0; t < 3; ++t) rw1[j][t] = w1[(16 * lh + j) * PS_HE + 32 * t + lj];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) rw2[t][r] = w2[(32 * t + fidx(r, 0) + 4 * lh) * 32 + lj];
#pragma unroll
    for (int r = 0; r < 8; ++r) rwc1[r] = wc1[(fidx(r, 0) + 4 * lh) * 32 + lj];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int t = 0; t < 2; ++t) rwf1[q][r][t] = wf1[q * 16 * PS_HN + (fidx(r, 0) + 4 * lh) * PS_HN + 32 * t + lj];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) rwf2[q][t][r] = wf2[q * PS_HN * 32 + (32 * t + fidx(r, 0) + 4 * lh) * 32 + lj];
    }

    for (int step = 0; step < A.L; ++step) {
        const float* Pc = A.P[step & 1];
        float* Pn = A.P[(step + 1) & 1];
        const float* ein = step == 0 ? A.e0 : A.e;
        for (int i = tid; i < 4 * PS_NB * 64; i += 256) nacc[i] = A.agg == MPNHIP_AGG_MAX ? -INFINITY : 0.f;
        __syncthreads();
        for (int tile = wave; tile < T && !(A.debug & 2); tile += 4) {
            // ---- one 32-edge tile per wave and iteration ---------------------------------------------------------------------------
            int dir = 3, pos0 = 0, cnt = 0;
            {
                int tt = tile;
                if (tt < t_out) dir = 0;
                else if ((tt -= t_out) < t_in) dir = 1;
                else { tt -= t_in; dir = 2; }
                pos0 = run0[dir] + 32 * tt;
                cnt = runn[dir] - 32 * tt;
                cnt = cnt < 32 ? cnt : 32;
            }
            if (cnt > 0) {
                const int ed = pos0 + (lj < cnt ? lj : cnt - 1);
                const int row = A.srow[ed], col = A.scol[ed];
                // first-layer input: lane half 0 carries the re-attached e0 row, half 1 the current e row (k = 16 h + j)
                const float* xr = (lh == 0 ? A.e0 : ein) + (int64_t)ed * 16;
                float xin[16];
#pragma unroll
                for (int j = 0; j < 16; j += 4) {
                    const float4 v = *reinterpret_cast<const float4*>(xr + j);
                    xin[j] = v.x; xin[j + 1] = v.y; xin[j + 2] = v.z; xin[j + 3] = v.w;
                }
                const float* prow = Pc + (int64_t)row * pw;
                const float* pcol = Pc + (int64_t)col * pw;
                // ---- edge MLP layer 0: H1 = relu(W1e [e0 | e] + Pr[row] + Pc[col]) ------------------------------------------------
                f32x16 h1[3];
#pragma unroll
                for (int t = 0; t < 3; ++t) {
                    h1[t] = zero16();
                    add_row_tile(h1[t], prow, 32 * t, he, lh);
                    add_row_tile(h1[t], pcol + he, 32 * t, he, lh);
                }
                // flow layer 0's gathered share, fetched early
                f32x16 hf[2];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    hf[t] = zero16();
                    if (dir < 2) add_row_tile(hf[t], pcol + 2 * he + dir * hn, 32 * t, hn, lh);
                }
#pragma unroll
                for (int j = 0; j < 16; ++j)
#pragma unroll
                    for (int t = 0; t < 3; ++t) h1[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(rw1[j][t], xin[j], h1[t], 0, 0, 0);
#pragma unroll
                for (int t = 0; t < 3; ++t) relu16p(h1[t]);
                // ---- edge MLP layer 1: e' = relu(W2 H1 + b2) -----------------------------------------------------------------------
                f32x16 en;
#pragma unroll
                for (int r = 0; r < 16; ++r) en[r] = bias[fidx(r, 0) + 4 * lh];
#pragma unroll
                for (int t = 0; t < 3; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        en = __builtin_amdgcn_mfma_f32_32x32x2f32(rw2[t][r], h1[t][r], en, 0, 0, 0);
                relu16p(en);
                if (lj < cnt) {   // features 0..15 live in registers 0..7: columns 4 h .. and 8 + 4 h ..
                    float* er = A.e + (int64_t)ed * 16 + 4 * lh;
                    *reinterpret_cast<float4*>(er) = make_float4(en[0], en[1], en[2], en[3]);
                    *reinterpret_cast<float4*>(er + 8) = make_float4(en[4], en[5], en[6], en[7]);
                }
                // ---- classifier: logit = wc2 . relu(Wc1 e' + bc1) + bc2 ----------------------------------------------------------
                {
                    f32x16 c;
#pragma unroll
                    for (int r = 0; r < 16; ++r) c[r] = bias[32 + fidx(r, 0) + 4 * lh];
#pragma unroll
                    for (int r = 0; r < 8; ++r) c = __builtin_amdgcn_mfma_f32_32x32x2f32(rwc1[r], en[r], c, 0, 0, 0);
                    float part = 0.f;
#pragma unroll
                    for (int r = 0; r < 16; ++r) part = fmaf(bias[64 + fidx(r, 0) + 4 * lh], fmaxf(c[r], 0.f), part);
                    part += __shfl_xor(part, 32, 64);
                    if (lh == 0 && lj < cnt) A.logits[(int64_t)step * E + A.perm[ed]] = part + A.bc2[0];
                }
                // ---- flow MLP of the tile's direction: m = relu(Wf2 relu(Wfe e' + Pf[col]) + bf2) ----------------------------------
                if (dir < 2) {
                    f32x16 mm;
#pragma unroll
                    for (int r = 0; r < 16; ++r) mm[r] = bias[96 + 32 * dir + fidx(r, 0) + 4 * lh];
                    auto flow = [&](auto qsel) {
                        constexpr int Q = decltype(qsel)::value;
#pragma unroll
                        for (int r = 0; r < 8; ++r)
#pragma unroll
                            for (int t = 0; t < 2; ++t) hf[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(rwf1[Q][r][t], en[r], hf[t], 0, 0, 0);
#pragma unroll
                        for (int t = 0; t < 2; ++t) relu16p(hf[t]);
#pragma unroll
                        for (int t = 0; t < 2; ++t)
#pragma unroll
                            for (int r = 0; r < 16; ++r) mm = __builtin_amdgcn_mfma_f32_32x32x2f32(rwf2[Q][t][r], hf[t][r], mm, 0, 0, 0);
                    };
                    if (dir == 0) flow(std::integral_constant<int, 0>{});
                    else flow(std::integral_constant<int, 1>{});
                    relu16p(mm);
                    float* sr = slab + (wave * 32 + lj) * 36 + 4 * lh;
#pragma unroll
                    for (int g = 0; g < 4; ++g) *reinterpret_cast<float4*>(sr + 8 * g) = make_float4(mm[4 * g], mm[4 * g + 1], mm[4 * g + 2], mm[4 * g + 3]);
                }
            }
            // ---- this tile's messages into the WAVE's per-(node, direction) partial aggregates, in edge order (the slab tile is the
            // wave's own: LDS operations of a wave complete in order, no block barrier in the tile loop).  Lane = (feature, half): the
            // halves take alternate nodes of the tile.
            if (cnt > 0 && dir < 2) {
                float* wacc = nacc + wave * PS_NB * 64;
                for (int i = lh; i < nb; i += 2) {
                    int a = lptr[dir * (PS_NB + 1) + i], b = lptr[dir * (PS_NB + 1) + i + 1];
                    a = a > pos0 ? a : pos0;
                    b = b < pos0 + cnt ? b : pos0 + cnt;
                    if (a >= b) continue;
                    float v = wacc[(i * 2 + dir) * 32 + lj];
                    const float* sp = slab + (wave * 32 + a - pos0) * 36 + lj;
                    if (A.agg == MPNHIP_AGG_MAX) { for (int p = a; p < b; ++p, sp += 36) v = fmaxf(v, *sp); }
                    else { for (int p = a; p < b; ++p, sp += 36) v += *sp; }
                    wacc[(i * 2 + dir) * 32 + lj] = v;
                }
            }
        }
        __syncthreads();
        // ---- node update for the block's nodes: x' = relu(Wu [agg_in | agg_out] + bu) -------------------------------------------------
        for (int i0 = 0; i0 < nb && !(A.debug & 4); i0 += 8) {
            const int i = i0 + (tid >> 5), f = tid & 31;
            if (i < nb) {
                float s = bias[160 + f];
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    const int d = half == 0 ? 1 : 0;                       // torch.cat((flow_in, flow_out)) (mpn.py:97)
                    const int len = lptr[d * (PS_NB + 1) + i + 1] - lptr[d * (PS_NB + 1) + i];
                    const float sc = A.agg == MPNHIP_AGG_MEAN ? 1.f / (float)(len > 1 ? len : 1) : 1.f;
                    const float* ag = nacc + (i * 2 + d) * 32;
                    if (len > 0) {                                         // (empty segment: zeros, also for max)
#pragma unroll 8
                        for (int k = 0; k < 32; ++k) {
                            // the four waves' partials in wave order (a fixed order: bitwise reproducible)
                            float av;
                            if (A.agg == MPNHIP_AGG_MAX)
                                av = fmaxf(fmaxf(ag[k], ag[PS_NB * 64 + k]), fmaxf(ag[2 * PS_NB * 64 + k], ag[3 * PS_NB * 64 + k]));
                            else
                                av = ((ag[k] + ag[PS_NB * 64 + k]) + ag[2 * PS_NB * 64 + k]) + ag[3 * PS_NB * 64 + k];
                            s = fmaf(wu[(32 * half + k) * 32 + f], av * sc, s);
                        }
                    }
                }
                s = fmaxf(s, 0.f);
                xs[i * 32 + f] = s;
                A.x_out[(int64_t)(n_lo + i) * 32 + f] = s;
            }
        }
        __syncthreads();
        if (step + 1 < A.L) {
            // ---- the next step's projection rows of the block's nodes: P'[n] = P0[n] + Wx x'[n] -----------------------------------------
            for (int i = 0; i < nb && !(A.debug & 4); ++i) {
                const float* xv = xs + i * 32;
                for (int o = tid; o < pw; o += 256) {
                    float s = A.P0[(int64_t)(n_lo + i) * pw + o];
#pragma unroll 8
                    for (int k = 0; k < 32; ++k) s = fmaf(wx[k * pw + o], xv[k], s);
                    Pn[(int64_t)(n_lo + i) * pw + o] = s;
                }
            }
            // ---- grid barrier: every block's P' rows (and e') are visible before any block gathers them ------------------------------
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0 && !(A.debug & 1)) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __hip_atomic_fetch_add((gu32*)A.barrier, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned want = (unsigned)(step + 1) * gridDim.x;
                unsigned spins = 0;
                while (__hip_atomic_load((gu32*)A.barrier, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
                    __builtin_amdgcn_s_sleep(2);
                    if (++spins > (1u << 26)) { A.barrier[1] = 1u; break; }   // (give up instead of hanging: the host reads this word)
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __syncthreads();
        }
    }
}

}  // namespace

size_t persist32_lds_bytes(int pw) {
    return (size_t)(32 * PS_HE + PS_HE * 32 + 16 * 32 + 2 * 16 * PS_HN + 2 * PS_HN * 32 + 64 * 32 + 32 * pw + 6 * 32 + 128 * 36 + 4 * PS_NB * 64 +
                    PS_NB * 32) * 4 + (2 * (PS_NB + 1) + 12) * 4;
}

bool persist32_supported(int dn, int de, int he, int hn, int hc, int pw, int kx, int64_t N, int64_t E) {
    static const int cus = [] { hipDeviceProp_t p; int d = 0; return hipGetDevice(&d) == hipSuccess && hipGetDeviceProperties(&p, d) == hipSuccess ? p.multiProcessorCount : 0; }();
    return dn == 32 && de == 16 && he % 4 == 0 && he <= PS_HE && hn % 4 == 0 && hn <= PS_HN && hc >= 1 && hc <= 32 && pw == 2 * he + 2 * hn && pw % 4 == 0 &&
           kx == 64 && N >= 1 && E >= 1 && E < ((int64_t)1 << 30) && cus >= 32 && N <= (int64_t)PS_NB * (cus - 8) && persist32_lds_bytes(pw) <= 160 * 1024 &&
           getenv("MPNHIP_PERSIST");   // opt-in: measured SLOWER than the launch-per-module path -- kept for that evidence
}

int launch_persist32(Persist32Args a, hipStream_t s) {
    static const int cus = [] { hipDeviceProp_t p; int d = 0; return hipGetDevice(&d) == hipSuccess && hipGetDeviceProperties(&p, d) == hipSuccess ? p.multiProcessorCount : 0; }();
    // one block per CU at most (every block must be resident: the step barrier); a few CUs are left to whatever else runs
    int grid = cus - 8;
    if (grid > a.N) grid = a.N;
    a.nodes_per_block = (a.N + grid - 1) / grid;
    grid = (a.N + a.nodes_per_block - 1) / a.nodes_per_block;
    if (a.nodes_per_block > PS_NB) { set_error("persist32: %d nodes per block", a.nodes_per_block); return MPNHIP_ERR_UNSUPPORTED; }
    const size_t lds = persist32_lds_bytes(a.pw);
    static bool attr_ok = false;
    if (!attr_ok) {
        MPN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(mpn_persist32_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_ok = true;
    }
    MPN_HIP(hipMemsetAsync(a.barrier, 0, 16, s));
    if (const char* e = getenv("MPNHIP_PERSIST_DEBUG")) a.debug = atoi(e);
    count_path(PC_PERSIST32);
    hipLaunchKernelGGL(mpn_persist32_kernel, dim3((unsigned)grid), dim3(256), lds, s, a);
    MPN_LAUNCH_CHECK();
    return MPNHIP_OK;
}

}  // namespace mpnhip
