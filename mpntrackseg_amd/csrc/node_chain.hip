// The node side of one message-passing step in ONE launch (TimeAwareNodeModel.forward, reference models/mpn.py:85-99, and the
// per-node projections the next step's edge kernel gathers):
//   AGG[n]  = [ node_agg_fn over n's flow_in messages | ... flow_out messages ]          (mpn.py:89,96,97)
//   x'[n]   = relu(Wu AGG[n] + bu)                                                        (mpn.py:98-99)
//   P'[n]   = P0[n] + Wx x'[n]          the next step's [Pr | Pc | Pf_out | Pf_in] rows (P0: the re-attached x0's share + biases)
// for widths above the reference's (dn = 64 / 128; dn = 32 has k_node_step32, segment.hip).  Before: k_aggregate + two launches of
// the GEMM kernel (7 + 11 + 22 us at cfg-B: 157 row tiles of a [5000 x 128] x [128 x 1088] product are four K steps each -- all
// prologue and epilogue).  Here a block owns 32 nodes (round 6 form, 8 waves):
//   A. aggregation: NWV lanes per (node, direction) segment read whole message rows, up to 4 rows in flight, in segment order; each lane
//      splits its run of the aggregate row into three bf16 pieces and leaves them in LDS as MFMA B-operand units (the fp32 row goes
//      to HBM from the registers when the backward needs it);
//   B. x' tile (wave % DT), the contraction shared by the NWV / DT waves of a tile: transposed product D^T[feature][node] (nodes on the
//      MFMA's lane dimension), six v_mfma_f32_32x32x16_bf16 per 16 contraction steps (edge_chain.hip section "split operands"); partial
//      tiles meet in LDS in a fixed order; bias, ReLU; x' to LDS as fp32 rows (stored as whole lines) and, split once, as B-operand units;
//   C. every wave takes every NWV-th 32-feature tile of P': C-in = the P0 tile, 6 x dn / 16 products; P0 in and P' out travel as whole
//      128-byte lines through a per-wave [32][36] LDS patch; the order "MFMAs of k block kb, then refill kb's ring slot for the next
//      tile" is pinned with sched_barriers (left alone hipcc drained the whole ring in front of every tile).
// Weights: packed once per forward as 1 KiB MFMA A-operand units [output tile][k block][piece] (k_pack_node_units), read
// straight from L2 by contiguous 16-byte-per-lane loads, a whole output tile ahead -- every block reads the same 1 MB image.
#include "common.h"

namespace mpnhip {

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

struct NSplit8 { bf16x8 p[3]; };
__device__ __forceinline__ NSplit8 nsplit8(const float4 lo, const float4 hi) {
    const float x[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    NSplit8 o;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const __bf16 h = (__bf16)x[i];
        const float a = x[i] - (float)h;
        const __bf16 m = (__bf16)a;
        o.p[0][i] = h; o.p[1][i] = m; o.p[2][i] = (__bf16)(a - (float)m);
    }
    return o;
}
__device__ __forceinline__ void nmfma6(f32x16& acc, const bf16x8 (&a)[3], const NSplit8& b) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b.p[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b.p[2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b.p[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b.p[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b.p[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b.p[0], acc, 0, 0, 0);
}
// the three pieces of one (tile, k block): 3 x 1 KiB, lane * 16 bytes each
__device__ __forceinline__ void ld_units(const unsigned short* img, int unit0, int lane, bf16x8 (&a)[3]) {
    const uint4* p = reinterpret_cast<const uint4*>(img + (size_t)unit0 * 512) + lane;
#pragma unroll
    for (int q = 0; q < 3; ++q) a[q] = __builtin_bit_cast(bf16x8, p[q * 64]);
}

// NWV waves per block (round 6: 8 -- two per SIMD; 4 was round 3's form): phases A and C are latency chains over a block's own rows
// (ablation, profiles/r06/node_chain_ablation_before.txt: of 35 us at cfg-B, aggregation 10, node update 5.4, projections 19.6 of which
// the MFMAs are 1.2), so more waves = more rows / tiles in flight per CU; 157 blocks leave 99 of the 256 CUs idle either way.
template <int DT, int NWV>
__global__ __launch_bounds__(64 * NWV) void node_chain_kernel(NodeChainArgs A) {
    constexpr int DN = 32 * DT, K2 = 2 * DN, KB2 = K2 / 16, KB1 = DN / 16, NTHR = 64 * NWV;
    constexpr int XP = DN + 4;                   // LDS row pitch (floats) of the x' tile
    constexpr int RD = 4;                        // k blocks of weight units in flight (phase B)
    static_assert(KB2 % RD == 0, "ring depth");
    // one LDS block: [asp | part_s | x_s | xsp].  asp / xsp: the aggregate tile / the x' tile as three bf16 pieces in MFMA B-operand
    // order -- 16-byte units [piece][k block][lane (node lj, k half lh)], written once, read conflict free by every wave that multiplies.
    // Phase C's per-wave [32][36] transposition patches re-use the asp | part_s region (dead behind the barrier that ends phase B).
    constexpr int ASPN = 3 * KB2 * 64 * 4;
    constexpr int PARTN = (NWV / DT > 1 ? NWV / DT - 1 : 0) * DT * 4 * 64 * 4;   // [k share - 1][tile][g][lane][4]
    constexpr int PATCH = 32 * 36;
    constexpr int HEADN = (ASPN + PARTN) > NWV * PATCH ? (ASPN + PARTN) : NWV * PATCH;
    constexpr int XSPN = 3 * KB1 * 64 * 4;
    __shared__ __attribute__((aligned(16))) float smem[HEADN + 32 * XP + XSPN];
    uint4* const asp = reinterpret_cast<uint4*>(smem);
    float* const part_s = smem + ASPN;
    float* const x_s = smem + HEADN;
    uint4* const xsp = reinterpret_cast<uint4*>(smem + HEADN + 32 * XP);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lj = lane & 31, lh = lane >> 5;
    const int n0 = blockIdx.x * 32;
    const int N = A.N;

    // ---- A. aggregation (into LDS as split B-operand units) ------------------------------------------------------------------------------------------
    // all 64 (node, direction) segments at once, 4 lanes each (a lane owns DN / 4 consecutive features = DN / 16 16-byte pieces of a
    // row); two rows in flight per lane, added in segment order.  (One round: the offsets -> rows latency chain is paid once,
    // not once per round of a few wide workers.)
    {
        constexpr int LPS = NWV;                 // lanes per (node, direction) segment: 64 segments over 64 NWV threads
        constexpr int PC = DN / LPS / 4;         // 16-byte pieces per lane and row (a segment's lanes read one row's consecutive pieces)
        static_assert(DN % (4 * LPS) == 0 && PC >= 1, "row pieces");
        const int sgm = tid / LPS, c0 = (tid % LPS) * (DN / LPS);
        const int nl = sgm >> 1, q = sgm & 1;    // q = 0: flow_out (keys [0, N)), 1: flow_in ([N, 2N))
        const int node = n0 + nl;
        float4 acc[PC];
#pragma unroll
        for (int u = 0; u < PC; ++u) acc[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (node < N) {
            const int key = q * N + node;
            const int b = A.seg_ptr[key], e = A.seg_ptr[key + 1];
            const float* src = A.M + c0;
#ifdef MPNHIP_NODE_FWD_DEBUG
            if (A.debug & 1) { if (e < b) acc[0].x = 1.f; } else
#endif
            if (A.agg == MPNHIP_AGG_MAX) {
                if (e > b) {
#pragma unroll
                    for (int u = 0; u < PC; ++u) acc[u] = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
                    int j = b;
                    for (; j + 2 <= e; j += 2) {
                        float4 v0[PC], v1[PC];
#pragma unroll
                        for (int u = 0; u < PC; ++u) {
                            v0[u] = *reinterpret_cast<const float4*>(src + (int64_t)j * DN + 4 * u);
                            v1[u] = *reinterpret_cast<const float4*>(src + (int64_t)(j + 1) * DN + 4 * u);
                        }
#pragma unroll
                        for (int u = 0; u < PC; ++u) {
                            acc[u].x = fmaxf(fmaxf(acc[u].x, v0[u].x), v1[u].x); acc[u].y = fmaxf(fmaxf(acc[u].y, v0[u].y), v1[u].y);
                            acc[u].z = fmaxf(fmaxf(acc[u].z, v0[u].z), v1[u].z); acc[u].w = fmaxf(fmaxf(acc[u].w, v0[u].w), v1[u].w);
                        }
                    }
                    if (j < e) {
#pragma unroll
                        for (int u = 0; u < PC; ++u) {
                            const float4 v = *reinterpret_cast<const float4*>(src + (int64_t)j * DN + 4 * u);
                            acc[u].x = fmaxf(acc[u].x, v.x); acc[u].y = fmaxf(acc[u].y, v.y); acc[u].z = fmaxf(acc[u].z, v.z); acc[u].w = fmaxf(acc[u].w, v.w);
                        }
                    }
                }                                 // empty segment -> 0 (torch_scatter fills with 0)
            } else {
                // up to four rows in flight per lane, added in segment order (ascending sorted edge position: the reference's CPU order)
                int j = b;
                for (; j + 4 <= e; j += 4) {
                    float4 v[4][PC];
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int u = 0; u < PC; ++u) v[r][u] = *reinterpret_cast<const float4*>(src + (int64_t)(j + r) * DN + 4 * u);
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int u = 0; u < PC; ++u) { acc[u].x += v[r][u].x; acc[u].y += v[r][u].y; acc[u].z += v[r][u].z; acc[u].w += v[r][u].w; }
                }
                if (j + 2 <= e) {
                    float4 v0[PC], v1[PC];
#pragma unroll
                    for (int u = 0; u < PC; ++u) {
                        v0[u] = *reinterpret_cast<const float4*>(src + (int64_t)j * DN + 4 * u);
                        v1[u] = *reinterpret_cast<const float4*>(src + (int64_t)(j + 1) * DN + 4 * u);
                    }
#pragma unroll
                    for (int u = 0; u < PC; ++u) {
                        acc[u].x += v0[u].x; acc[u].y += v0[u].y; acc[u].z += v0[u].z; acc[u].w += v0[u].w;
                        acc[u].x += v1[u].x; acc[u].y += v1[u].y; acc[u].z += v1[u].z; acc[u].w += v1[u].w;
                    }
                    j += 2;
                }
                if (j < e) {
#pragma unroll
                    for (int u = 0; u < PC; ++u) {
                        const float4 v = *reinterpret_cast<const float4*>(src + (int64_t)j * DN + 4 * u);
                        acc[u].x += v.x; acc[u].y += v.y; acc[u].z += v.z; acc[u].w += v.w;
                    }
                }
                if (A.agg == MPNHIP_AGG_MEAN) {
                    const float inv = 1.f / (float)(e - b > 1 ? e - b : 1);
#pragma unroll
                    for (int u = 0; u < PC; ++u) { acc[u].x *= inv; acc[u].y *= inv; acc[u].z *= inv; acc[u].w *= inv; }
                }
            }
        }
        // torch.cat((flow_in, flow_out)) (mpn.py:97): flow_in on the left.  The lane's DN / LPS consecutive features are whole 8-column
        // halves of k blocks of node nl's row: split here, once, and left as B-operand units; the fp32 row (kept for the backward pass)
        // leaves from the registers -- a segment's lanes cover consecutive pieces of one row
        const int col0 = (q == 0 ? DN : 0) + c0;
        static_assert(PC % 2 == 0, "whole 8-column halves per lane");
#pragma unroll
        for (int h = 0; h < PC / 2; ++h) {
            const int col = col0 + 8 * h;
            const NSplit8 sp = nsplit8(acc[2 * h], acc[2 * h + 1]);
#pragma unroll
            for (int pz = 0; pz < 3; ++pz) asp[(pz * KB2 + (col >> 4)) * 64 + nl + 32 * ((col >> 3) & 1)] = __builtin_bit_cast(uint4, sp.p[pz]);
        }
        if (A.agg_out && node < N) {
#pragma unroll
            for (int u = 0; u < PC; ++u) *reinterpret_cast<float4*>(A.agg_out + (int64_t)node * K2 + col0 + 4 * u) = acc[u];
        }
    }
    __syncthreads();

#ifdef MPNHIP_NODE_FWD_DEBUG
    if (A.debug & 2) return;
#endif
    // ---- C's first loads: issued BEFORE phase B (they depend on nothing computed here) -- the waves beyond the DT that compute the
    // node update would otherwise sit at the barrier with nothing in flight ------------------------------------------------------------
    const int NT = A.pw / 32;
    // this wave's tiles: wave, wave + NWV, ...; every block walks them from a different starting tile (all blocks stream the same
    // 1 MB image: started together on the same units they queue on the same L2 channels)
    const bool has_c = A.P_next != nullptr && wave < NT;
    const int cnt = has_c ? (NT - wave + NWV - 1) / NWV : 1;
    int ii = (int)((blockIdx.x * 5u) % (unsigned)cnt);
    int t = has_c ? wave + NWV * ii : 0;
    constexpr int RC = KB1;                      // a whole output tile of weight units ahead (L2 latency under 157 blocks' streams ~ 1 us)
    bf16x8 ringc[RC][3];
    // P0 rows in and P' rows out travel as WHOLE 128-byte lines: lane -> row (lane / 8 + 8 p), columns 4 (lane % 8) .. +3 of the tile --
    // four instructions of eight full lines each.  In the accumulator layout (lane = node, 32 bytes per row and instruction) the same
    // 4 KB are 128 partial-line accesses, and the launch is bound by exactly that count (profiles/r06/node_chain_ablation.txt: weight
    // units 192 line accesses per tile / 5 us, P0 128 / 5 us, stores 128 / 5.5 us, additive).  The transposition between the two layouts
    // goes through a per-wave LDS patch (LDS operations of one wave complete in order: no barrier).
    const int er = lane >> 3, ec = (lane & 7) * 4;
    // (named scalars, not arrays: hipcc left `float4 pr[4]` in scratch memory -- or, at 8 waves, promoted it to 32 KB of LDS)
#define NC_ROWS4(X) X(0) X(1) X(2) X(3)
#define NC_DECL(P) float4 pr##P = make_float4(0.f, 0.f, 0.f, 0.f); const bool rok##P = n0 + er + 8 * P < N; \
                   const int64_t rowoff##P = (int64_t)(rok##P ? n0 + er + 8 * P : N - 1) * A.pw + ec;
    NC_ROWS4(NC_DECL)
#undef NC_DECL
    if (has_c) {
#pragma unroll
        for (int s = 0; s < RC; ++s) ld_units(A.wx_img, (t * KB1 + s) * 3, lane, ringc[s]);
#define NC_LOAD(P) pr##P = *reinterpret_cast<const float4*>(A.P0 + rowoff##P + 32 * t);
        NC_ROWS4(NC_LOAD)
#undef NC_LOAD
    }

    // ---- B. x' = relu(Wu AGG + bu): tile (wave % DT), the contraction split over the NWV / DT waves that share a tile; the partial
    // tiles of the upper shares meet the first share's in LDS (fixed order: reproducible) -----------------------------------------------
    {
        constexpr int KS = NWV / DT, KPW = KB2 / KS, RB = KPW < RD ? KPW : RD;
        static_assert(NWV % DT == 0 && KB2 % KS == 0 && KPW % RB == 0, "k shares");
        const int tb = wave % DT, ks = wave / DT, kb0 = ks * KPW;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        bf16x8 ring[RB][3];
#pragma unroll
        for (int s = 0; s < RB; ++s) ld_units(A.wu_img, (tb * KB2 + kb0 + s) * 3, lane, ring[s]);
        float4 bias4[4];     // (requested here, not behind the products: an L2 round trip off the block's critical path)
#pragma unroll
        for (int g = 0; g < 4; ++g) bias4[g] = *reinterpret_cast<const float4*>(A.bu + 32 * tb + 8 * g + 4 * lh);
        NSplit8 bq[2];
#pragma unroll
        for (int q = 0; q < 3; ++q) bq[0].p[q] = __builtin_bit_cast(bf16x8, asp[(q * KB2 + kb0) * 64 + lane]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < KPW; ++k) {
            if (k + 1 < KPW) {
#pragma unroll
                for (int q = 0; q < 3; ++q) bq[(k + 1) & 1].p[q] = __builtin_bit_cast(bf16x8, asp[(q * KB2 + kb0 + k + 1) * 64 + lane]);
                __builtin_amdgcn_sched_barrier(0);
            }
            nmfma6(acc, ring[k % RB], bq[k & 1]);
            __builtin_amdgcn_sched_barrier(0);
            if (k + RB < KPW) {
                ld_units(A.wu_img, (tb * KB2 + kb0 + k + RB) * 3, lane, ring[k % RB]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (KS > 1) {
            if (ks > 0) {
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *reinterpret_cast<float4*>(&part_s[((((ks - 1) * DT + tb) * 4 + g) * 64 + lane) * 4]) =
                        make_float4(acc[4 * g + 0], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]);
            }
            __syncthreads();
        }
        if (ks == 0) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float4 a4 = make_float4(acc[4 * g + 0], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]);
#pragma unroll
                for (int q = 1; q < KS; ++q) {
                    const float4 u = *reinterpret_cast<const float4*>(&part_s[((((q - 1) * DT + tb) * 4 + g) * 64 + lane) * 4]);
                    a4.x += u.x; a4.y += u.y; a4.z += u.z; a4.w += u.w;
                }
                const int n = 32 * tb + 8 * g + 4 * lh;
                const float4 bias = bias4[g];
                const float4 v = make_float4(fmaxf(a4.x + bias.x, 0.f), fmaxf(a4.y + bias.y, 0.f), fmaxf(a4.z + bias.z, 0.f), fmaxf(a4.w + bias.w, 0.f));
                *reinterpret_cast<float4*>(&x_s[lj * XP + n]) = v;
            }
        }
    }
    __syncthreads();
    // x' rows leave from the LDS tile as whole lines (all waves; the accumulator layout would store 32 bytes per row and instruction)
    for (int i = tid; i < 32 * (DN / 4); i += NTHR) {
        const int nl = i / (DN / 4), c = (i - nl * (DN / 4)) * 4;
        if (n0 + nl < N) *reinterpret_cast<float4*>(A.x_new + (int64_t)(n0 + nl) * DN + c) = *reinterpret_cast<const float4*>(&x_s[nl * XP + c]);
    }
    if (!A.P_next) return;   // last step: no projections needed (block-uniform)
#ifdef MPNHIP_NODE_FWD_DEBUG
    if (A.debug & 4) return;
#endif
    // x' split ONCE into its three bf16 pieces, k block kb by wave kb (round 3's form: every wave split the whole tile into 96 registers
    // of its own -- with the ring and the row staging that spilled); phase C reads the pieces as 16-byte units, conflict free
    for (int kb = wave; kb < KB1; kb += NWV) {
        const float* xr = &x_s[lj * XP + 16 * kb + 8 * lh];
        const NSplit8 b = nsplit8(*reinterpret_cast<const float4*>(xr), *reinterpret_cast<const float4*>(xr + 4));
#pragma unroll
        for (int q = 0; q < 3; ++q) xsp[(q * KB1 + kb) * 64 + lane] = __builtin_bit_cast(uint4, b.p[q]);
    }
    __syncthreads();
    if (!has_c) return;      // more waves than tiles

    // ---- C. P' = P0 + Wx x': tiles wave, wave + NWV, ... ----------------------------------------------------------------------
    float* const patch = smem + wave * PATCH;
    // hipcc's scheduler, left alone, sinks all 24 weight-unit loads of the next tile below the tile's last MFMA and the wait-count pass
    // then drains them in front of the next tile's first one: load latency and MFMAs in series, ~10k cycles per tile for 1.5k of MFMA
    // (round 3's form of this loop; seen in the ISA in round 6).  The sched_barriers pin the intended order: MFMAs of k block kb, then
    // the loads that refill its ring slot for the next tile.
    const bool full = n0 + 32 <= N;
    {
        for (int it = 0; it < cnt; ++it) {
            f32x16 acc;
            // C-in: the P0 tile, rows -> patch -> accumulator layout
#define NC_STAGE(P) *reinterpret_cast<float4*>(&patch[(er + 8 * P) * 36 + ec]) = pr##P;
            NC_ROWS4(NC_STAGE)
#undef NC_STAGE
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 c = *reinterpret_cast<const float4*>(&patch[lj * 36 + 8 * g + 4 * lh]);
                acc[4 * g + 0] = c.x; acc[4 * g + 1] = c.y; acc[4 * g + 2] = c.z; acc[4 * g + 3] = c.w;
            }
            ii = ii + 1 < cnt ? ii + 1 : 0;
            const int tn = wave + NWV * ii;          // (after the last tile: the first one again -- unconditional loads)
#define NC_LOAD(P) pr##P = *reinterpret_cast<const float4*>(A.P0 + rowoff##P + 32 * tn);
            NC_ROWS4(NC_LOAD)
#undef NC_LOAD
            NSplit8 bq[2];
#pragma unroll
            for (int q = 0; q < 3; ++q) bq[0].p[q] = __builtin_bit_cast(bf16x8, xsp[(q * KB1 + 0) * 64 + lane]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int kb = 0; kb < KB1; ++kb) {
                if (kb + 1 < KB1) {
#pragma unroll
                    for (int q = 0; q < 3; ++q) bq[(kb + 1) & 1].p[q] = __builtin_bit_cast(bf16x8, xsp[(q * KB1 + kb + 1) * 64 + lane]);
                    __builtin_amdgcn_sched_barrier(0);   // (the next block's pieces are requested before this block's six MFMAs, not after the fifth)
                }
#ifdef MPNHIP_NODE_FWD_DEBUG
                if (A.debug & 8) {
#pragma unroll
                    for (int u = 0; u < 3; ++u) asm volatile("" ::"v"(ringc[kb][u]), "v"(bq[kb & 1].p[u]));
                } else
#endif
                nmfma6(acc, ringc[kb], bq[kb & 1]);
                __builtin_amdgcn_sched_barrier(0);
#ifdef MPNHIP_NODE_FWD_DEBUG
                if (!(A.debug & 16))
#endif
                ld_units(A.wx_img, (tn * KB1 + kb) * 3, lane, ringc[kb]);
                __builtin_amdgcn_sched_barrier(0);
            }
#ifdef MPNHIP_NODE_FWD_DEBUG
            if (A.debug & 32) { t = tn; continue; }
#endif
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *reinterpret_cast<float4*>(&patch[lj * 36 + 8 * g + 4 * lh]) = make_float4(acc[4 * g + 0], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]);
#define NC_STORE(P) { const float4 v = *reinterpret_cast<const float4*>(&patch[(er + 8 * P) * 36 + ec]); \
                      if (full || rok##P) *reinterpret_cast<float4*>(A.P_next + rowoff##P + 32 * t) = v; }
            NC_ROWS4(NC_STORE)
#undef NC_STORE
            __builtin_amdgcn_sched_barrier(0);
            t = tn;
        }
    }
#undef NC_ROWS4
}


// The node side of one step of the BACKWARD pass in one launch (the mirror of the kernel above; autograd of mpn.py:97-99 and of the
// per-node projections): for the 32 nodes of a block
//   dX    = dP Wx                       [32 x pw] x [pw x dn]   gradient w.r.t. x_{s-1} through step s's projections
//   dZn   = dX (.) [x_{s-1} > 0]        ReLU of step s-1's node update
//   dAGG  = dZn Wu                      [32 x dn] x [dn x 2 dn]
// Before: a grouped GEMM of 2 x 158 blocks and 34 K steps each (28 us at cfg-B), k_relu_mask (5 us) and a GEMM (10 us) on the
// caller's stream between two chain kernels.  Split operands as everywhere in MPNHIP_PREC_FP32_SPLIT: three bf16 pieces, six MFMAs.
//   1. the pw / 16 contraction blocks of the first product are dealt to the block's NWV waves round-robin; a wave splits ITS blocks of
//      the dP rows once (32 bytes per lane and block, straight from global memory) and multiplies them into all dn / 32 output
//      tiles; weight units from L2 as in the forward kernel, NB blocks ahead;
//   2. the partial tiles meet in LDS; wave t sums tile t, applies the mask, stores dZn rows and leaves the tile in LDS;
//   3. every wave takes every NWV-th 32-column tile of dAGG (dn / 16 blocks each).
// Measured at cfg-B (N = 5000: 157 blocks, K = 1088; ablation build make EXTRA=-DMPNHIP_NODE_BWD_DEBUG, rocprofv3 minimum of 275
// launches): 22.1 us whole, 20.7 without the phase-1 MFMAs, 19.1 without phase 3, 8.4 without phase 1, 4.7 without both -- the 12 us
// of phase 1 are its loads: every block pulls the same 835 KB of weight units (+ 139 KB of dP rows) through its CU's L1 at
// ~37 bytes per clock.  The three launches it replaces took 28 + 5 + 10 us.
template <int DT, int NWV>
__global__ __launch_bounds__(64 * NWV) void node_chain_bwd_kernel(NodeChainBwdArgs A) {
    constexpr int DN = 32 * DT, KB1 = DN / 16, XP = DN + 4;
    constexpr int NB = NWV > 4 ? 2 : 4;          // contraction blocks of weight units / dP pieces in flight per wave (phase 1)
    __shared__ __attribute__((aligned(16))) float part_s[NWV * DT * 4 * 64 * 4];   // [wave][tile][g][lane][4]
    __shared__ __attribute__((aligned(16))) float z_s[32 * XP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lj = lane & 31, lh = lane >> 5;
    const int n0 = blockIdx.x * 32;
    const int N = A.N;
    const int node = n0 + lj < N ? n0 + lj : N - 1;
    const bool ok = n0 + lj < N;
    const int KBP = A.pw / 16;
    // v_mfma_f32_32x32x16_bf16 adds its products with a small bias toward -infinity (tools/micro/mfma_bias.hip); every sum the
    // backward takes over nodes / edges would add it up coherently.  As in the split backward chain kernel (edge_chain.hip) the
    // gradients of every other node are kept NEGATED in the registers and LDS (flipped as they are loaded and as they are stored)
    const unsigned sx = (lj & 1) ? 0x80000000u : 0u;
    auto fx = [&](float v) { return __uint_as_float(__float_as_uint(v) ^ sx); };
    auto fx4 = [&](float4 v) { return make_float4(fx(v.x), fx(v.y), fx(v.z), fx(v.w)); };

    // ---- 1. partial dX^T tiles over this wave's contraction blocks wave, wave + 4, ... -------------------------------------
    {
        f32x16 acc[DT];
#pragma unroll
        for (int t = 0; t < DT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
        const int cnt = (KBP - wave + NWV - 1) / NWV;    // >= NB is not required: out-of-range blocks are clamped to the last one and unused
        const float* dpr = A.dP + (int64_t)node * A.pw + 8 * lh;
        const int start = cnt > 0 ? (int)((blockIdx.x * 5u) % (unsigned)cnt) : 0;
        bf16x8 ring[NB][DT][3];
        float4 xr[NB][2];
        auto fetch = [&](int it, int slot) {
            // (every block walks its blocks from a different start: all N / 32 blocks stream the same image -- started together on
            // the same units they queue on the same L2 channels, as in the forward kernel)
            int ii = (it < cnt ? it : cnt - 1) + start;
            ii = ii >= cnt ? ii - cnt : ii;
            const int kb = wave + NWV * ii;
#pragma unroll
            for (int t = 0; t < DT; ++t) ld_units(A.wxT_img, (t * KBP + kb) * 3, lane, ring[slot][t]);
            xr[slot][0] = *reinterpret_cast<const float4*>(dpr + 16 * kb);
            xr[slot][1] = *reinterpret_cast<const float4*>(dpr + 16 * kb + 4);
        };
#ifdef MPNHIP_NODE_BWD_DEBUG
        if (cnt > 0 && !(A.debug & 4)) {
#else
        if (cnt > 0) {
#endif
#pragma unroll
            for (int q = 0; q < NB; ++q) fetch(q, q);
            // whole rounds of NB blocks WITHOUT a branch inside (a conditional stage makes hipcc's wait-count pass merge two load
            // orders at every join: it then drains vmcnt(0) before each stage -- no loads in flight under the MFMAs, 24 us instead
            // of ~12 at cfg-B), then the remaining < NB stages
            const int full = cnt / NB;
            for (int r = 0; r < full; ++r) {
#pragma unroll
                for (int q = 0; q < NB; ++q) {
                    const NSplit8 b = nsplit8(fx4(xr[q][0]), fx4(xr[q][1]));
#ifdef MPNHIP_NODE_BWD_DEBUG
                    if (A.debug & 1) {
#pragma unroll
                        for (int t = 0; t < DT; ++t)
#pragma unroll
                            for (int u = 0; u < 3; ++u) asm volatile("" ::"v"(ring[q][t][u]), "v"(b.p[u]));
                    } else
#endif
#pragma unroll
                    for (int t = 0; t < DT; ++t) nmfma6(acc[t], ring[q][t], b);
                    fetch((r + 1) * NB + q, q);
                    __builtin_amdgcn_sched_barrier(0);   // (the scheduler would hoist every stage's split to the top of the round: a full drain)
                }
            }
            const int rem = cnt - full * NB;
#pragma unroll
            for (int q = 0; q < NB - 1; ++q) {
                if (q < rem) {               // (wave-uniform)
                    const NSplit8 b = nsplit8(fx4(xr[q][0]), fx4(xr[q][1]));
#pragma unroll
                    for (int t = 0; t < DT; ++t) nmfma6(acc[t], ring[q][t], b);
                }
            }
        }
#pragma unroll
        for (int t = 0; t < DT; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *reinterpret_cast<float4*>(&part_s[(((wave * DT + t) * 4 + g) * 64 + lane) * 4]) =
                    make_float4(acc[t][4 * g + 0], acc[t][4 * g + 1], acc[t][4 * g + 2], acc[t][4 * g + 3]);
    }
    // (the ReLU source rows of phase 2 are requested in front of the barrier, not one by one behind it: written in place the compiler
    // waits each of the four loads out before the next is issued -- four L2 round trips in a row on the block's critical path)
    float4 xp[4];
    if (wave < DT) {
#pragma unroll
        for (int g = 0; g < 4; ++g) xp[g] = *reinterpret_cast<const float4*>(A.x_prev + (int64_t)node * DN + 32 * wave + 8 * g + 4 * lh);
    }
    __syncthreads();

    // ---- 2. dZn tile `wave` = (sum of the partials) (.) [x_prev > 0] -----------------------------------------------------
    if (wave < DT) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float4 v = *reinterpret_cast<const float4*>(&part_s[(((0 * DT + wave) * 4 + g) * 64 + lane) * 4]);
#pragma unroll
            for (int w = 1; w < NWV; ++w) {
                const float4 u = *reinterpret_cast<const float4*>(&part_s[(((w * DT + wave) * 4 + g) * 64 + lane) * 4]);
                v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
            }
            const int n = 32 * wave + 8 * g + 4 * lh;
            const float4 x = xp[g];
            v.x = x.x > 0.f ? v.x : 0.f; v.y = x.y > 0.f ? v.y : 0.f; v.z = x.z > 0.f ? v.z : 0.f; v.w = x.w > 0.f ? v.w : 0.f;
            *reinterpret_cast<float4*>(&z_s[lj * XP + n]) = v;
            if (ok) *reinterpret_cast<float4*>(A.dZn + (int64_t)(n0 + lj) * DN + n) = fx4(v);
        }
    }
    __syncthreads();

    // ---- 3. dAGG = dZn Wu: tiles wave, wave + 4, ... of the 2 dn columns ---------------------------------------------------------
    {
        NSplit8 bz[KB1];
#pragma unroll
        for (int kb = 0; kb < KB1; ++kb) {
            const float* zr = &z_s[lj * XP + 16 * kb + 8 * lh];
            bz[kb] = nsplit8(*reinterpret_cast<const float4*>(zr), *reinterpret_cast<const float4*>(zr + 4));
        }
        constexpr int NT = 2 * DT;
        float* out = A.dAGG + (int64_t)node * (2 * DN) + 4 * lh;
        bf16x8 ring[KB1][3];
        if (wave < NT) {
#pragma unroll
            for (int kb = 0; kb < KB1; ++kb) ld_units(A.wuT_img, (wave * KB1 + kb) * 3, lane, ring[kb]);
        }
#ifdef MPNHIP_NODE_BWD_DEBUG
        if (A.debug & 2) return;
#endif
        for (int t = wave; t < NT; t += NWV) {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            const int tn = t + NWV < NT ? t + NWV : t;
#pragma unroll
            for (int kb = 0; kb < KB1; ++kb) {
                nmfma6(acc, ring[kb], bz[kb]);
                ld_units(A.wuT_img, (tn * KB1 + kb) * 3, lane, ring[kb]);
            }
            if (ok) {
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *reinterpret_cast<float4*>(out + 32 * t + 8 * g) = fx4(make_float4(acc[4 * g + 0], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]));
            }
        }
    }
}

// units [tile t][k block kb][piece]: lane (m = lane & 31, g = lane >> 5), element i = piece(A[32 t + m][16 kb + 8 g + i]) of the
// logical [n_out x K] operand A[n][k] = W[n * ldw + (col0 + k) * sk]  (sk = 1: rows of W; the backward's transposed operands
// read W column-wise: ldw = 1, sk = W's row pitch)
__global__ __launch_bounds__(64) void k_pack_node_units(const float* __restrict__ W, int64_t ldw, int col0, int n_out, int K,
                                                         unsigned short* __restrict__ dst, int64_t sk = 1) {
    const int lane = threadIdx.x, m = lane & 31, g = lane >> 5;
    const int kbs = (K + 15) / 16;
    const int t = blockIdx.x / kbs, kb = blockIdx.x - t * kbs;
    float x[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int n = 32 * t + m, k = 16 * kb + 8 * g + i;
        x[i] = (n < n_out && k < K) ? W[(int64_t)n * ldw + (int64_t)(col0 + k) * sk] : 0.f;
    }
    const NSplit8 s = nsplit8(make_float4(x[0], x[1], x[2], x[3]), make_float4(x[4], x[5], x[6], x[7]));
    uint4* out = reinterpret_cast<uint4*>(dst + (size_t)blockIdx.x * 3 * 512) + lane;
#pragma unroll
    for (int q = 0; q < 3; ++q) out[q * 64] = __builtin_bit_cast(uint4, s.p[q]);
}

}  // namespace

bool node_chain_supported(int dn, int pw, int kx) {
    return (dn == 64 || dn == 128) && pw % 32 == 0 && pw >= 32 && kx == 2 * dn && !getenv("MPNHIP_NO_NODE_CHAIN");
}
size_t node_chain_image_shorts(int dn, int pw, size_t* off_wx) {
    const size_t wu = (size_t)(dn / 32) * (2 * dn / 16) * 3 * 512;
    if (off_wx) *off_wx = wu;
    return wu + (size_t)(pw / 32) * (dn / 16) * 3 * 512;
}
// Wu: node update weight [dn, 2 dn]; Wnode: packed projection weights [pw, kx], their current-feature columns [dn, 2 dn)
int pack_node_chain(const float* Wu, const float* Wnode, int dn, int pw, int kx, unsigned short* img, hipStream_t s) {
    size_t off = 0;
    node_chain_image_shorts(dn, pw, &off);
    hipLaunchKernelGGL(k_pack_node_units, dim3((dn / 32) * (2 * dn / 16)), dim3(64), 0, s, Wu, (int64_t)2 * dn, 0, dn, 2 * dn, img);
    hipLaunchKernelGGL(k_pack_node_units, dim3((pw / 32) * (dn / 16)), dim3(64), 0, s, Wnode, (int64_t)kx, dn, pw, dn, img + off);
    MPN_LAUNCH_CHECK();
    return MPNHIP_OK;
}

int launch_node_chain(const NodeChainArgs& a_in, hipStream_t s) {
    if (a_in.N <= 0) return MPNHIP_OK;
    count_path(PC_NODE_CHAIN);
#ifdef MPNHIP_NODE_FWD_DEBUG
    NodeChainArgs a = a_in;
    a.debug = getenv("MPNHIP_NODE_FWD_DEBUG") ? atoi(getenv("MPNHIP_NODE_FWD_DEBUG")) : 0;
#else
    const NodeChainArgs& a = a_in;
#endif
    const unsigned blocks = (unsigned)((a.N + 31) / 32);
    // (MPNHIP_NODE_CHAIN_WAVES=4: round 3's four-wave blocks, A-B)
    static const int nwv = getenv("MPNHIP_NODE_CHAIN_WAVES") ? atoi(getenv("MPNHIP_NODE_CHAIN_WAVES")) : 8;
    if (a.dn == 128 && nwv == 4) MPN_LAUNCH_PROFILED((node_chain_kernel<4, 4>), dim3(blocks), dim3(256), s, a);
    else if (a.dn == 128) MPN_LAUNCH_PROFILED((node_chain_kernel<4, 8>), dim3(blocks), dim3(512), s, a);
    else if (a.dn == 64 && nwv == 4) MPN_LAUNCH_PROFILED((node_chain_kernel<2, 4>), dim3(blocks), dim3(256), s, a);
    else if (a.dn == 64) MPN_LAUNCH_PROFILED((node_chain_kernel<2, 8>), dim3(blocks), dim3(512), s, a);
    else { set_error("node_chain: unsupported width %d", a.dn); return MPNHIP_ERR_UNSUPPORTED; }
    MPN_LAUNCH_CHECK();
    return MPNHIP_OK;
}

bool node_chain_bwd_supported(int dn, int pw, int kx) {
    return (dn == 64 || dn == 128) && pw % 16 == 0 && pw >= 64 && kx == 2 * dn && !getenv("MPNHIP_NO_NODE_CHAIN_BWD");
}
size_t node_chain_bwd_image_shorts(int dn, int pw, size_t* off_wu) {
    const size_t wx = (size_t)(dn / 32) * (pw / 16) * 3 * 512;
    if (off_wu) *off_wu = wx;
    return wx + (size_t)(2 * dn / 32) * (dn / 16) * 3 * 512;
}
// the transposed operands: A1[k][p] = Wnode[p][dn + k] (dn x pw), A2[c][k] = Wu[k][c] (2 dn x dn)
int pack_node_chain_bwd(const float* Wu, const float* Wnode, int dn, int pw, int kx, unsigned short* img, hipStream_t s) {
    size_t off = 0;
    node_chain_bwd_image_shorts(dn, pw, &off);
    hipLaunchKernelGGL(k_pack_node_units, dim3((dn / 32) * (pw / 16)), dim3(64), 0, s, Wnode + dn, (int64_t)1, 0, dn, pw, img, (int64_t)kx);
    hipLaunchKernelGGL(k_pack_node_units, dim3((2 * dn / 32) * (dn / 16)), dim3(64), 0, s, Wu, (int64_t)1, 0, 2 * dn, dn, img + off, (int64_t)2 * dn);
    MPN_LAUNCH_CHECK();
    return MPNHIP_OK;
}

int launch_node_chain_bwd(const NodeChainBwdArgs& a_in, hipStream_t s) {
    if (a_in.N <= 0) return MPNHIP_OK;
    count_path(PC_NODE_CHAIN_BWD);
#ifdef MPNHIP_NODE_BWD_DEBUG
    NodeChainBwdArgs a = a_in;
    a.debug = getenv("MPNHIP_NODE_BWD_DEBUG") ? atoi(getenv("MPNHIP_NODE_BWD_DEBUG")) : 0;
#else
    const NodeChainBwdArgs& a = a_in;
#endif
    const unsigned blocks = (unsigned)((a.N + 31) / 32);
    // (4 or 8 waves per block -- one or two per SIMD -- measure the same, 21.3 / 21.0 us at cfg-B: the launch is bound by the weight
    // units every block streams from L2, see below; 4 is the default for its smaller LDS footprint beside the side stream's blocks)
    static const int nwv = getenv("MPNHIP_NODE_BWD_WAVES") ? atoi(getenv("MPNHIP_NODE_BWD_WAVES")) : 4;
    if (a.dn == 128 && nwv == 4) hipLaunchKernelGGL((node_chain_bwd_kernel<4, 4>), dim3(blocks), dim3(256), 0, s, a);
    else if (a.dn == 128) hipLaunchKernelGGL((node_chain_bwd_kernel<4, 8>), dim3(blocks), dim3(512), 0, s, a);
    else if (a.dn == 64 && nwv == 4) hipLaunchKernelGGL((node_chain_bwd_kernel<2, 4>), dim3(blocks), dim3(256), 0, s, a);
    else if (a.dn == 64) hipLaunchKernelGGL((node_chain_bwd_kernel<2, 8>), dim3(blocks), dim3(512), 0, s, a);
    else { set_error("node_chain_bwd: unsupported width %d", a.dn); return MPNHIP_ERR_UNSUPPORTED; }
    MPN_LAUNCH_CHECK();
    return MPNHIP_OK;
}

}  // namespace mpnhip
