// Device-side building blocks shared by the bf16-operand chain kernels (edge_chain_bf16.hip: forward; edge_chain_bf16_bwd.hip:
// backward): 1 KiB weight units streamed L2 -> LDS by LDS-DMA, A operands read back by hand-placed ds_read_b128, the N-tiled
// "hidden tile" step (first-layer units -> activation -> the finished tile IS the next layer's B operand), training saves.
#pragma once
#include "common.h"
#include "edge_chain.h"
#include "row_stage.h"
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

constexpr int DEPTH = 4;   // A operands in flight per wave (default; the kernels' DEP template parameter)

// NU units from LDS address wa + OFF0 (+ 1 KiB per unit), DEP reads in flight; use(u, a) consumes unit u's A operand
template <int OFF0, int NU, int DEP = DEPTH, class USE>
__device__ __forceinline__ void stream_units(unsigned wa, USE&& use) {
    bf16x8 a[DEP];
    static_for<0, (DEP - 1 < NU ? DEP - 1 : NU)>([&](auto U) { lds_read<OFF0 + U.value * 1024>(wa, a[U.value]); });
    static_for<0, NU>([&](auto U) {
        constexpr int u = U.value;
        if constexpr (u + DEP - 1 < NU) {
            lds_read<OFF0 + (u + DEP - 1) * 1024>(wa, a[(u + DEP - 1) % DEP]);
            lds_wait<DEP - 1>(a[u % DEP]);
        } else {
            lds_wait<0>(a[u % DEP]);
        }
        use(U, a[u % DEP]);
    });
}

// One hidden tile: KA first-layer units against the k blocks xin[0 .. KA) into `acc`; act(acc): the activation (forward: ReLU;
// backward: the saved ReLU decisions as a mask); then 2 x TO second-layer units: k block c of the finished tile (rounded to
// bf16) into out[o].  wa = LDS address of the chunk buffer (+ lane * 16), OFF0 = the section's offset in it.
// fin(hb0, hb1): called once with the finished tile as bf16 (registers 0-7 | 8-15), before the second-layer products (the
// tile is saved from here).
template <int OFF0, int KA, int TO, int DEP = DEPTH, class ACT, class FIN>
__device__ __forceinline__ void hidden_tile(unsigned wa, const bf16x8* xin, f32x16& acc, f32x16* out, ACT&& act, FIN&& fin) {
    bf16x8 hb[2];
    stream_units<OFF0, KA + 2 * TO, DEP>(wa, [&](auto U, const bf16x8& a) {
        constexpr int u = U.value;
        if constexpr (u < KA) {
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, xin[u], acc, 0, 0, 0);
        } else {
            if constexpr (u == KA) {
                act(acc);
                hb[0] = pack_regs(acc, 0);
                hb[1] = pack_regs(acc, 1);
                fin(hb[0], hb[1]);
            }
            constexpr int q = u - KA;
            out[q % TO] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, hb[q / TO], out[q % TO], 0, 0, 0);
        }
    });
}

// ---- training saves (SAVE variant) ----------------------------------------------------------------------------------------
// A finished tile lives in a lane as 16 values: features 32 t + 8 g + 4 h + (0..3), g = 0..3 (h = lane / 32), i.e. four runs
// of four consecutive features; as bf16 that is four 8-byte pieces of the edge's row.  Row-major bf16 [E, width] is what the
// consumers want (the weight-gradient kernel streams whole rows, the scatter-adds gather rows).
// A finished tile as two 16-byte pieces per lane: lanes (edge, 0) and (edge, 1) each hold four runs of 4 features (8 g + 4 h ..);
// four v_permlane32_swap exchange runs between the two halves so that lane (edge, h) ends up with the 16 CONSECUTIVE features
// 32 t + 16 h .. + 15 of the edge's row -- two dwordx4 stores to 32 contiguous bytes instead of four dwordx2 stores to 8-byte
// pieces (store issue, not bandwidth, is what a row-per-lane epilogue pays: cdna_hip_programming.md T21)
__device__ __forceinline__ void tile_rows16(const bf16x8& h0, const bf16x8& h1, uint4& lo, uint4& hi) {
    const u32x4 a = __builtin_bit_cast(u32x4, h0), b = __builtin_bit_cast(u32x4, h1);
    // swap(x, y): x keeps its lanes 0..31 and takes y's lanes 0..31 into its lanes 32..63; y takes x's lanes 32..63 into its lanes 0..31
    const auto r0 = __builtin_amdgcn_permlane32_swap(a[0], b[0], false, false);
    const auto r1 = __builtin_amdgcn_permlane32_swap(a[1], b[1], false, false);
    const auto r2 = __builtin_amdgcn_permlane32_swap(a[2], b[2], false, false);
    const auto r3 = __builtin_amdgcn_permlane32_swap(a[3], b[3], false, false);
    lo = make_uint4(r0[0], r1[0], r0[1], r1[1]);   // h = 0: features 0..7   (run g = 0 of both halves); h = 1: 16..23 (g = 2)
    hi = make_uint4(r2[0], r3[0], r2[1], r3[1]);   // h = 0: features 8..15  (g = 1);                     h = 1: 24..31 (g = 3)
}

// ReLU decisions of a finished tile as 16 bits: register pair i (elements 2i, 2i+1 of the tile = bf16 halves lo, hi of packed
// register i) -> bits i and 16 + i, shifted by 8 for the odd tile of a pair: one 32-bit word per two tiles
// (bit of element r of tile parity p: 8 p + (r >> 1) + 16 (r & 1); chain_bf16_mask_bit).  min(x, 1) on the bf16 bit patterns
// (non-negative after the ReLU) is the "non-zero" flag of both halves in one packed instruction.
__device__ __forceinline__ unsigned tile_mask_bits(const bf16x8& h0, const bf16x8& h1, unsigned ones) {
    const u32x4 a = __builtin_bit_cast(u32x4, h0), b = __builtin_bit_cast(u32x4, h1);
    unsigned w = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        unsigned m0, m1;
        asm("v_pk_min_u16 %0, %1, %2" : "=v"(m0) : "v"(a[i]), "v"(ones));
        asm("v_pk_min_u16 %0, %1, %2" : "=v"(m1) : "v"(b[i]), "v"(ones));
        w |= m0 << i;
        w |= m1 << (4 + i);
    }
    return w;
}

// backward: acc[r] *= decision bit of element r (tile parity p in mask word w; chain_bf16_mask_bit): one v_bfe_i32 (0 / -1) and
// one v_and per element
template <int PAR>
__device__ __forceinline__ void apply_mask16(f32x16& a, unsigned w) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int keep = __builtin_amdgcn_sbfe((int)w, 8 * PAR + (r >> 1) + 16 * (r & 1), 1);
        a[r] = __uint_as_float(__float_as_uint(a[r]) & (unsigned)keep);
    }
}

}  // namespace

}  // namespace mpnhip
