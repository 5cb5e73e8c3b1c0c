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
// Widths are padded to multiples of 32 inside the kernel (weight images are zero-padded when they are packed, row
// pieces beyond the real width are masked on load / store), so one template serves the BASELINE.json 128-d
// configuration (he 320, de 64, hn 224, dn 128, hc 32 -> tiles 10/2/7/4), the reference's shipped 32-d dims
// (80/16/56/32/8 -> 3/1/2/1) and 64-d (5/1/4/2); anything else uses the unfused GEMM path.
#include <type_traits>

#include "common.h"
#include "edge_chain.h"
#include "row_stage.h"
#ifdef MPNHIP_CHAIN_TS
#include <cstdio>
#include <string>
#include <vector>
#endif

namespace mpnhip {

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

// Debug build (make EXTRA=-DMPNHIP_CHAIN_TS): lane 0 of every wave stamps s_memtime at the phase boundaries; with
// MPNHIP_CHAIN_TS=<file prefix> in the environment the 40th launch of each kernel dumps its stamps as text.
#ifdef MPNHIP_CHAIN_TS
#define TS_INIT() long long* tsp = A.ts ? A.ts + ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 16 : nullptr
#define TS(i) do { if (tsp && (threadIdx.x & 63) == 0) tsp[i] = clock64(); } while (0)
// slot 15: where the wave ran -- HW_ID (wave / simd / cu / sh / se fields) | XCC_ID << 32
#define TS_WHERE() do { if (tsp && (threadIdx.x & 63) == 0) { unsigned hw, xcc; \
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw)); asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc)); \
    tsp[15] = (long long)hw | ((long long)(xcc & 0xf) << 32); } } while (0)
#else
#define TS_INIT() do {} while (0)
#define TS(i) do {} while (0)
#define TS_WHERE() do {} while (0)
#endif

constexpr int CH_FLOATS = 7680;  // floats per weight chunk buffer (20 KB fp32 images, 30 KB split images)

__device__ __forceinline__ float4 ldg4(const float* p) { return *reinterpret_cast<const float4*>(p); }

// Weight chunks are CONTIGUOUS runs of floats in the packed images (the images are laid out so that every chunk is:
// whole rows of a [k][n] image, or one column block stored as its own image).  They are copied with the gfx950
// LDS-DMA load (global_load_lds_dwordx4: 1 KiB per wave instruction, lane-linear destination, no staging registers,
// no ds_write pass), issued at the start of the chunk BEFORE the one that uses them and drained by the vmcnt(0) that
// __syncthreads() carries.  Every load is unconditional (source index clamped into the chunk; surplus lanes land in
// the unused tail of the 20 KB buffer) and sizes are compile-time constants wherever the schedule is static: a load
// behind a branch makes the compiler's s_waitcnt bookkeeping conservative.
template <int Q, bool EXACT_UNITS = false>
__device__ __forceinline__ void chunk_fetch(const float* src, int n4, int tid, float* buf) {
    static_assert(Q >= 1 && Q <= 8, "chunk larger than the largest buffer");
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        if (EXACT_UNITS) {
            // split images: chunks are whole 1 KiB units = whole wave instructions; a wave skips the pieces past the end
            // (wave-uniform), so the buffer needs no tail
            if (256 * q + 64 * wave < n4)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + 4 * (tid + 256 * q)),
                                                 (__attribute__((address_space(3))) void*)(buf + 4 * (256 * q + 64 * wave)), 16, 0, 0);
            continue;
        }
        int f = tid + 256 * q;
        f = f < n4 ? f : n4 - 1;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + 4 * f),
                                         (__attribute__((address_space(3))) void*)(buf + 4 * (256 * q + 64 * wave)), 16, 0, 0);
    }
}

constexpr int cmax(int a, int b) { return a > b ? a : b; }
constexpr int cmin(int a, int b) { return a < b ? a : b; }
constexpr int chunk_q(int n4) { return (n4 + 255) / 256; }

// 16-byte piece of a feature row at column n (n % 4 == 0, dim % 4 == 0): zero beyond the real width
template <bool EXACT>
__device__ __forceinline__ float4 ldrow(const float* row, int n, int dim) {
    if (EXACT) return ldg4(row + n);  // widths are multiples of 32: nothing to mask
    const bool ok = n < dim;
    float4 v = ldg4(row + (ok ? n : 0));
    v.x = ok ? v.x : 0.f; v.y = ok ? v.y : 0.f; v.z = ok ? v.z : 0.f; v.w = ok ? v.w : 0.f;
    return v;
}
template <bool EXACT>
__device__ __forceinline__ void strow(float* row, int n, int dim, float4 v, bool ok) {
    if (ok && (EXACT || n < dim)) *reinterpret_cast<float4*>(row + n) = v;
}

// the same through a (scalar) base pointer and a 32-bit element offset: one address register per row instead of two
template <bool EXACT>
__device__ __forceinline__ float4 ldrow(const float* base, unsigned off, int n, int dim) {
    // (base + zext(off)) + n: scalar base, 32-bit register offset, n in the instruction's immediate field
    if (EXACT) return ldg4(base + (size_t)off + n);
    const bool ok = n < dim;
    float4 v = ldg4(base + (size_t)off + (ok ? n : 0));
    v.x = ok ? v.x : 0.f; v.y = ok ? v.y : 0.f; v.z = ok ? v.z : 0.f; v.w = ok ? v.w : 0.f;
    return v;
}
template <bool EXACT>
__device__ __forceinline__ void strow(float* base, unsigned off, int n, int dim, float4 v, bool ok) {
    if (ok && (EXACT || n < dim)) *reinterpret_cast<float4*>(base + (size_t)off + n) = v;
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
// sign flips by XOR (sx = 0x80000000 or 0): the split backward kernel keeps the gradients of every other edge NEGATED in its
// registers (see edge_chain_bwd_kernel)
__device__ __forceinline__ float fxor(float x, unsigned sx) { return __uint_as_float(__float_as_uint(x) ^ sx); }
__device__ __forceinline__ float4 flip4(float4 v, unsigned sx) { return make_float4(fxor(v.x, sx), fxor(v.y, sx), fxor(v.z, sx), fxor(v.w, sx)); }
__device__ __forceinline__ void flip16(f32x16& a, unsigned sx) {
#pragma unroll
    for (int r = 0; r < 16; ++r) { const float x = a[r]; a[r] = fxor(x, sx); }
}

// ReLU masks travel from the forward to the backward chain kernel as bits: bit r of a tile's 16-bit mask = accumulator
// register r of that lane is > 0.  Both kernels map (block, wave, lane) to the same edge and feature rows, so the
// words are private to a lane: word w of wave-tile q sits at mask[(q * NW + w) * 64 + lane] (256 contiguous bytes per
// wave store / load).  Word ranges per section (two tiles per word): H1 | e' | HC | HF | M.
__device__ __forceinline__ unsigned mask16(const f32x16& a) {
    unsigned m = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) m |= (a[r] > 0.f ? 1u : 0u) << r;
    return m;
}
__device__ __forceinline__ void apply_mask(f32x16& a, unsigned word, int shift) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        // bit -> all-ones / zero, AND with the float's bits
        const int keep = __builtin_amdgcn_sbfe(word, shift + r, 1);
        const float x = a[r];  // (bit_cast straight on the vector element reads element 0: clang 19 / ROCm 7.2)
        a[r] = __int_as_float(__float_as_int(x) & keep);
    }
}

// ---- split operands (EdgeChainArgs.split): every fp32 operand x is the exact sum of three bfloat16 pieces
//   h = bf16(x),  m = bf16(x - h),  l = bf16(x - h - m)        (round to nearest even; residuals are exact in fp32)
// and a product a b is accumulated (fp32) from the six piece products of relative weight >= 2^-16:
//   a_h b_h + a_h b_m + a_m b_h + a_m b_m + a_h b_l + a_l b_h ;   the dropped a_m b_l + a_l b_m + a_l b_l < 2^-24 |a b|,
// i.e. below fp32 rounding: the same accuracy class as the fp32 MFMA (measured against float64: tools/gemm_bench.py
// --check, tests/test_gpu_parity.py), at 6 x 32 instead of 8 x 64 MFMA cycles per 16 contraction steps
// (v_mfma_f32_32x32x16_bf16 against v_mfma_f32_32x32x2_f32).
// Weight images: units of 1 KiB = one MFMA A operand (64 lanes x 8 bf16), [k block of 16][n tile of 32][piece]; element i
// of lane (m, g) holds W[n0 + m][k0 + (i & 3) + 8 (i >> 2) + 4 g] -- the contraction order of the accumulator layout, so
// that registers 8c .. 8c+7 of an activation tile ARE the B operand of k block c.  A [K][N] image takes K N 6 bytes,
// 3/2 of the fp32 image: the chunk schedule is the fp32 one with every offset and size scaled by 3/2.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
struct Split8 { bf16x8 p[3]; };

__device__ __forceinline__ Split8 split8(float x0, float x1, float x2, float x3, float x4, float x5, float x6, float x7) {
    const float x[8] = {x0, x1, x2, x3, x4, x5, x6, x7};
    Split8 o;
#pragma unroll
    for (int i = 0; i < 8; i += 2) {
        const __bf16 h0 = (__bf16)x[i], h1 = (__bf16)x[i + 1];
        const float a0 = x[i] - (float)h0, a1 = x[i + 1] - (float)h1;
        const __bf16 m0 = (__bf16)a0, m1 = (__bf16)a1;
        const __bf16 l0 = (__bf16)(a0 - (float)m0), l1 = (__bf16)(a1 - (float)m1);
        o.p[0][i] = h0; o.p[0][i + 1] = h1;
        o.p[1][i] = m0; o.p[1][i + 1] = m1;
        o.p[2][i] = l0; o.p[2][i + 1] = l1;
    }
    return o;
}
__device__ __forceinline__ Split8 split_regs(const f32x16& s, int r0) {
    return r0 == 0 ? split8(s[0], s[1], s[2], s[3], s[4], s[5], s[6], s[7]) : split8(s[8], s[9], s[10], s[11], s[12], s[13], s[14], s[15]);
}

// (lds_addr: row_stage.h)
// The three pieces of one unit.  Inline assembly on purpose: a compiler-visible ds_read_b128 of the object the LDS-DMA
// writes into is preceded by s_waitcnt vmcnt(0) (the next chunk's DMA would be drained in front of every operand
// fetch); the waits for these reads are placed by hand (lds_wait: LDS operations of a wave complete in order).
__device__ __forceinline__ void lds_read3(unsigned addr, bf16x8 (&a)[3]) {
    asm volatile("ds_read_b128 %0, %3\n\tds_read_b128 %1, %3 offset:1024\n\tds_read_b128 %2, %3 offset:2048"
                 : "=&v"(a[0]), "=&v"(a[1]), "=&v"(a[2]) : "v"(addr) : "memory");
}
template <int N>
__device__ __forceinline__ void lds_wait(bf16x8 (&a)[3]) {
    asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]) : "n"(N));
}
__device__ __forceinline__ void mfma6(f32x16& acc, const bf16x8 (&a)[3], const Split8& b) {
#ifdef MPNHIP_SPLIT9   // diagnostic build: all nine piece products
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b.p[2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b.p[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b.p[2], acc, 0, 0, 0);
#endif
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b.p[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b.p[2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b.p[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b.p[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b.p[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b.p[0], acc, 0, 0, 0);
}

// Split counterpart of chain_tile / chain_half: NKB k blocks (registers 8 c0 .. of `src`) against the units
// [kb0 + c][t0 + t] of the chunk image at LDS address wsaddr (+ lane * 16 already added), NTR tiles per k block.
// One step = one unit: the next unit's three reads go out before this unit's six MFMAs.
template <int TOUT, int NKB, bool PIPE = true>
__device__ __forceinline__ void chain_units(const f32x16& src, int c0, f32x16* out, unsigned wsaddr, int ntr, int kb0, int t0) {
    auto unit = [&](int st) { return wsaddr + (unsigned)((((kb0 + st / TOUT) * ntr + t0 + st % TOUT) * 3) << 10); };
    if (!PIPE) {
        // register-starved phases: one operand set, fetched just before its MFMAs (the SIMD's other wave covers the LDS latency)
        Split8 b = split_regs(src, 8 * c0);
#pragma unroll
        for (int st = 0; st < NKB * TOUT; ++st) {
            bf16x8 a[3];
            lds_read3(unit(st), a);
            if (NKB == 2 && st == TOUT) b = split_regs(src, 8);
            lds_wait<0>(a);
            mfma6(out[st % TOUT], a, b);
            __builtin_amdgcn_sched_barrier(0);  // (keeps the next unit's fetch behind these MFMAs: one operand set live)
        }
        return;
    }
    bf16x8 a[2][3];
    lds_read3(unit(0), a[0]);
    Split8 b = split_regs(src, 8 * c0);
#pragma unroll
    for (int st = 0; st < NKB * TOUT; ++st) {
        if (st + 1 < NKB * TOUT) {
            lds_read3(unit(st + 1), a[(st + 1) & 1]);
            lds_wait<3>(a[st & 1]);
        } else {
            lds_wait<0>(a[st & 1]);
        }
        if (NKB == 2 && st == TOUT) b = split_regs(src, 8);
        mfma6(out[st % TOUT], a[st & 1], b);
    }
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

// T1 = ceil(he/32), T2 = ceil(de/32), TF = ceil(hn/32), TD = ceil(dn/32); hc <= 32.  Capital names = padded widths
// (weight-image pitches, loop bounds); A.he / A.de / ... = real widths (global row strides, load / store masks).
constexpr int chain_waves(int t1) { return t1 <= 3 ? 3 : 2; }  // waves per SIMD the register budget allows

template <int T1, int T2, int TF, int TD, bool EXACT, bool SP>
__global__ __launch_bounds__(256, chain_waves(T1)) void edge_chain_kernel(EdgeChainArgs A) {
    constexpr int HE = 32 * T1, DE = 32 * T2, HN = 32 * TF, DN = 32 * TD, HC = 32;
    constexpr int KC1 = 16;                       // phase-1 chunk: [16 k][HE]
    constexpr int KC2 = 64;                       // phase-2 chunk: [<=64 k][DE]
    constexpr int NC4 = 64;                       // phase-4 chunk: [DE k][<=64 n]
    constexpr int KC5 = 32;                       // phase-5 chunk: [32 k][DN]

    // ---- chunk schedule (float4 counts; every chunk is a contiguous run of its image) -----------------------
    constexpr int NCH2 = (HE + KC2 - 1) / KC2;
    constexpr int NCH4 = (HN + NC4 - 1) / NC4;
    constexpr int NCH5 = HN / KC5;
    constexpr bool P2PIPE = T1 < 10;                         // phase 2 of the widest variant has no registers for a second operand set
    constexpr int SCN = SP ? 3 : 2;                          // image size in halves of the fp32 image's (split images: 3/2)
    constexpr int N4_1 = KC1 * HE / 4 * SCN / 2;             // phase 1: 16 rows of W1T
    constexpr int N4_2_0 = cmin(KC2, HE) * DE / 4 * SCN / 2; // phase 2, first chunk
    constexpr int N4_3 = DE * HC / 4 * SCN / 2;              // classifier layer 0, whole
    constexpr int N4_4_0 = DE * cmin(NC4, HN) / 4 * SCN / 2; // phase 4, first column block
    constexpr int N4_5 = KC5 * DN / 4 * SCN / 2;             // phase 5: 32 rows of Wf2T
    constexpr int N4_MAX = cmax(cmax(N4_1, N4_2_0), cmax(cmax(N4_3, N4_4_0), N4_5));
    // chunk buffer: whole 1 KiB DMA pieces of the largest chunk (split images: exactly the largest chunk)
    constexpr int CHF = SP ? 4 * N4_MAX : 1024 * chunk_q(N4_MAX);
    static_assert(CHF <= CH_FLOATS, "chunk too large");
    // ReLU-mask words per lane for the backward kernel (edge_chain.h: chain_mask_words)
    constexpr int W_H1 = 0, W_E = W_H1 + (T1 + 1) / 2, W_HC = W_E + (T2 + 1) / 2, W_HF = W_HC + 1, W_M = W_HF + (TF + 1) / 2,
                  NW = W_M + (TD + 1) / 2;
    // ONE LDS object (a second one beside the LDS-DMA target makes hipcc drain vmcnt before unrelated ds_reads):
    // two weight-chunk buffers, two phase-1 input staging buffers, then the biases [b2 (DE) | bc1 (32) | wc2 (32) | bf2 (DN)], zero-padded
    constexpr int XS_FLOATS = 4 * 2 * 64 * 4;  // phase-1 input staging per buffer: 4 waves x 2 pieces x 64 lanes x 16 B
    __shared__ __attribute__((aligned(16))) float smem[2 * CHF + 2 * XS_FLOATS + DE + 64 + DN];
    float* const wbuf0 = smem;
    float* const xs0 = smem + 2 * CHF;
    float* const sbias = xs0 + 2 * XS_FLOATS;
#define wbuf_at(i) (wbuf0 + ((i) & 1) * CHF)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lj = lane & 31, lh = lane >> 5;
    const int he = A.he, de = A.de, hn = A.hn, dn = A.dn, hc = A.hc;

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
    const int K1 = A.k1a + A.k1b;                 // columns of [e0 | e] (multiples of 16)
    const int nch1 = K1 / KC1;
    const bool flow = grp < 2;
    // (Row stores of the FORWARD kernel stay as they are -- 16-byte pieces straight from the accumulator layout.  Measured with the
    // backward kernel's slab scheme (row_stage.h; slabs in place of the phase-1 staging buffers + 2 KB): all saves through slabs 83.3 ->
    // 88.7 us per training launch (30 registers spilled at the phase-2 peak), HC / HF / M only 83.0 us: the forward is not bound
    // by its stores.)
    // (wave-uniform base + lane: no address registers; rows of P / Q0 / save_h1 are addressed as base + 32-bit offset)
    unsigned* const mkb = A.save_mask ? A.save_mask + (int64_t)__builtin_amdgcn_readfirstlane(blockIdx.x * 4 + wave) * NW * 64 : nullptr;
#define MKP(w) mkb[(w) * 64 + lane]

    const float* wf1 = grp == 1 ? A.wf1T_in : A.wf1T_out;   // NCH4 column-block images [DE][<=64], block i at DE * 64 * i
    const float* wf2 = grp == 1 ? A.wf2T_in : A.wf2T_out;

    int c = 0;  // chunk being computed (buffer parity)
    TS_INIT();
    TS(0);
    TS_WHERE();
    chunk_fetch<chunk_q(N4_1), SP>(A.w1T, N4_1, tid, wbuf_at(0));
    {
        // biases -> LDS (ordinary loads; drained with chunk 0 by the first barrier)
        const float* bf2 = grp == 1 ? A.bf2_in : A.bf2_out;
        float v = 0.f;
        if (tid < DE) v = tid < de ? A.b2[tid] : 0.f;
        else if (tid < DE + 32) v = tid - DE < hc ? A.bc1[tid - DE] : 0.f;
        else if (tid < DE + 64) v = tid - DE - 32 < hc ? A.wc2[tid - DE - 32] : 0.f;
        else if (tid < DE + 64 + DN) v = tid - DE - 64 < dn ? bf2[tid - DE - 64] : 0.f;
        if (tid < DE + 64 + DN) sbias[tid] = v;
    }

    // ---- phase 1: H1^T = W1e [e0|e]^T, C-in = Pr[row] (+ Pc[col] after the MFMAs) ---------------------------
    const int row = A.srow[edge], col = A.scol[edge];
    const unsigned eh = (unsigned)edge * (unsigned)he;   // this edge's row of Q0 / save_h1
    f32x16 h1[T1];
    {
        const float* pr = A.P + (int64_t)row * A.pw;
#pragma unroll
        for (int t = 0; t < T1; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) set4(h1[t], g, ldrow<EXACT>(pr, 32 * t + 8 * g + 4 * lh, he));
        if (A.Q0) {  // + the step-invariant share of the layer (hoisted out of the step loop like P0)
            // four tiles' pieces in flight at a time (written tile by tile the compiler waits for every tile's four loads before it
            // issues the next: ten load latencies in a row at the head of every wave)
#pragma unroll
            for (int t0 = 0; t0 < T1; t0 += 4) {
                float4 q[4][4];
#pragma unroll
                for (int tt = 0; tt < 4; ++tt)
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        if (t0 + tt < T1) q[tt][g] = ldrow<EXACT>(A.Q0, eh, 32 * (t0 + tt) + 8 * g + 4 * lh, he);
                __builtin_amdgcn_sched_barrier(0);   // (the scheduler would sink every load to its add again)
#pragma unroll
                for (int tt = 0; tt < 4; ++tt)
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        if (t0 + tt < T1) add4(h1[t0 + tt], g, q[tt][g]);
            }
        }
    }
    // Pc[col] joins H1 after the MFMAs, two tiles (8 row pieces) per gather round; round r is issued one chunk before
    // the phase-2 chunk that consumes tiles 2r, 2r+1 (round 0: in the last phase-1 chunk)
    float4 pcv[8];
    const unsigned pco = (unsigned)col * (unsigned)A.pw + (unsigned)he;
    float* const sv = A.save_h1;
    auto pc_issue = [&](int r) {
#pragma unroll
        for (int g = 0; g < 8; ++g)
            if (2 * r + (g >> 2) < T1) pcv[g] = ldrow<EXACT>(A.P, pco, 32 * (2 * r + (g >> 2)) + 8 * (g & 3) + 4 * lh, he);
    };
    auto pc_finish = [&](int r) {
#pragma unroll
        for (int g = 0; g < 8; ++g)
            if (2 * r + (g >> 2) < T1) add4(h1[2 * r + (g >> 2)], g & 3, pcv[g]);
        relu16(h1[2 * r]);
        if (2 * r + 1 < T1) relu16(h1[2 * r + 1]);
        if (sv) {
#pragma unroll
            for (int g = 0; g < 8; ++g)
                if (2 * r + (g >> 2) < T1)
                    strow<EXACT>(sv, eh, 32 * (2 * r + (g >> 2)) + 8 * (g & 3) + 4 * lh, he, get4(h1[2 * r + (g >> 2)], g & 3), edge_ok);
        }
        if (mkb) MKP(W_H1 + r) = mask16(h1[2 * r]) | (2 * r + 1 < T1 ? mask16(h1[2 * r + 1]) << 16 : 0u);
    };
    {
        // lane (j, h) reads its edge's features 16 bytes at a time: k = 8u + 4h + (0..3)
        const float* xa = A.xa + (int64_t)edge * A.ldxa + 4 * lh;
        const float* xb = A.xb ? A.xb + (int64_t)edge * A.ldxb + 4 * lh - A.k1a : xa;
        // ... through LDS by LDS-DMA as well (each lane's 16 bytes land at its own lane-linear slot and are read back by
        // the same lane): with only DMA loads in the loop the compiler's vmcnt waits stay where the barriers are
        const int wave_u = __builtin_amdgcn_readfirstlane(wave);
        auto xfetch = [&](int k, int buf) {
#pragma unroll
            for (int u = 0; u < 2; ++u)
                __builtin_amdgcn_global_load_lds(
                    (const __attribute__((address_space(1))) void*)((k + 8 * u < A.k1a ? xa : xb) + k + 8 * u),
                    (__attribute__((address_space(3))) void*)(xs0 + (buf & 1) * XS_FLOATS + (wave_u * 2 + u) * 256), 16, 0, 0);
        };
        xfetch(0, 0);
        __syncthreads();  // chunk 0 is in wbuf[0]
        TS(1);
        // One phase-1 chunk (16 contraction rows).  `last`: the chunk also carries the first Pc gather round.
        auto p1_chunk = [&](int i, bool last1) {
            // next chunk: the following 16 rows of W1T, or the first chunk of phase 2
            const float* nsrc = last1 ? A.w2T : A.w1T + (int64_t)(i + 1) * (KC1 * HE * SCN / 2);
            const int nn4 = last1 ? N4_2_0 : N4_1;
            // (read this chunk's inputs BEFORE the DMAs go out: hipcc drains vmcnt in front of a plain ds_read_b128 that
            // follows an LDS-DMA into the same object)
            float4 xcur[2];
#pragma unroll
            for (int u = 0; u < 2; ++u)
                xcur[u] = *reinterpret_cast<const float4*>(xs0 + (c & 1) * XS_FLOATS + (wave * 2 + u) * 256 + lane * 4);
            __builtin_amdgcn_sched_barrier(0);
            chunk_fetch<chunk_q(cmax(N4_1, N4_2_0)), SP>(nsrc, nn4, tid, wbuf_at(c + 1));
            if (last1) pc_issue(0);
            else xfetch((i + 1) * KC1, c + 1);
            if constexpr (SP) {
                // one k block: the lane's 8 inputs (k = 4h + 0..3, 8 + 4h + 0..3) are the B operand as they are
                const Split8 xb8 = split8(xcur[0].x, xcur[0].y, xcur[0].z, xcur[0].w, xcur[1].x, xcur[1].y, xcur[1].z, xcur[1].w);
                const unsigned wa = lds_addr(wbuf_at(c)) + lane * 16;
                bf16x8 wv[2][3];
                lds_read3(wa, wv[0]);
#pragma unroll
                for (int t = 0; t < T1; ++t) {
                    if (t + 1 < T1) {
                        lds_read3(wa + (unsigned)(((t + 1) * 3) << 10), wv[(t + 1) & 1]);
                        lds_wait<3>(wv[t & 1]);
                    } else {
                        lds_wait<0>(wv[t & 1]);
                    }
                    mfma6(h1[t], wv[t & 1], xb8);
                }
            } else {
                const float* ws = wbuf_at(c) + 4 * lh * HE + lj;
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
            }
            __syncthreads();
            ++c;
        };
        for (int i = 0; i + 1 < nch1; ++i) p1_chunk(i, false);
        p1_chunk(nch1 - 1, true);
    }

    TS(3);
    // ---- phase 2: e'^T = relu(W2 H1^T + b2) -----------------------------------------------------------
    // Chunk i contracts H1 tiles 2i and 2i+1.  Their finishing touches -- + Pc[col] (gather round i, issued one chunk
    // earlier), ReLU, the training-mode save -- come first; then the prefetches for the NEXT chunk go out (weights by
    // LDS-DMA, gather round i+1, and two tiles of phase 4's C-in into the registers the consumed H1 tiles free), so that
    // every gather has a whole chunk of MFMAs to land and is drained by the barrier that ends the chunk.
    f32x16 en[T2];
    f32x16 hf[TF];
    const unsigned pfo = pco + (unsigned)(he + (grp == 1 ? hn : 0));  // (self-loop blocks gather flow_out's and drop it)
    auto pf_issue = [&](int t) {
#pragma unroll
        for (int g = 0; g < 4; ++g) set4(hf[t], g, ldrow<EXACT>(A.P, pfo, 32 * t + 8 * g + 4 * lh, hn));
    };
#pragma unroll
    for (int t = 0; t < T2; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) set4(en[t], g, *reinterpret_cast<const float4*>(sbias + 32 * t + 8 * g + 4 * lh));
    static_assert(NCH2 == (T1 + 1) / 2, "one Pc gather round per phase-2 chunk");
#pragma unroll
    for (int i = 0; i < NCH2; ++i) {
        pc_finish(i);
        // next: rows 64 (i + 1) .. of W2T, or the classifier image
        const bool more = i + 1 < NCH2;
        const int rows_n = more ? (HE - (i + 1) * KC2 < KC2 ? HE - (i + 1) * KC2 : KC2) : 0;  // folds: i is unrolled
        const float* nsrc = more ? A.w2T + (i + 1) * (KC2 * DE * SCN / 2) : A.wc1T;
        const int nn4 = more ? rows_n * DE / 4 * SCN / 2 : N4_3;
        __builtin_amdgcn_sched_barrier(0);
        chunk_fetch<chunk_q(cmax(N4_2_0, N4_3)), SP>(nsrc, nn4, tid, wbuf_at(c + 1));
        if (more) pc_issue(i + 1);
        if (i >= 1) {
            if (2 * (i - 1) < TF) pf_issue(2 * (i - 1));
            if (2 * (i - 1) + 1 < TF) pf_issue(2 * (i - 1) + 1);
        }
        const float* ws = wbuf_at(c);
        if constexpr (SP) {
            const unsigned wa = lds_addr(ws) + lane * 16;
            chain_units<T2, 2, P2PIPE>(h1[2 * i], 0, en, wa, T2, 0, 0);
            if (2 * i + 1 < T1) chain_units<T2, 2, P2PIPE>(h1[2 * i + 1], 0, en, wa, T2, 2, 0);
        } else {
            chain_tile<T2>(h1[2 * i], en, ws, DE, 0, 0, 4 * lh * DE + lj);
            if (2 * i + 1 < T1) chain_tile<T2>(h1[2 * i + 1], en, ws, DE, 32, 0, 4 * lh * DE + lj);
        }
        __syncthreads();
        ++c;
    }
    {
        float* o = A.e_new + (int64_t)edge * de;
#pragma unroll
        for (int t = 0; t < T2; ++t) {
            relu16(en[t]);
#pragma unroll
            for (int g = 0; g < 4; ++g) strow<EXACT>(o, 32 * t + 8 * g + 4 * lh, de, get4(en[t], g), edge_ok);
        }
        if (mkb) {
#pragma unroll
            for (int t = 0; t < T2; t += 2) MKP(W_E + (t >> 1)) = mask16(en[t]) | (t + 1 < T2 ? mask16(en[t + 1]) << 16 : 0u);
        }
    }

    TS(4);
    // phase-4 C-in tiles the phase-2 chunks did not cover (none for the shipped width sets)
#pragma unroll
    for (int t = 2 * (NCH2 - 1); t < TF; ++t) pf_issue(t);
    // ---- phase 3: classifier ----------------------------------------------------------------------------
    {
        f32x16 hcv;
#pragma unroll
        for (int g = 0; g < 4; ++g) set4(hcv, g, *reinterpret_cast<const float4*>(sbias + DE + 8 * g + 4 * lh));
        chunk_fetch<chunk_q(N4_4_0), SP>(wf1, N4_4_0, tid, wbuf_at(c + 1));  // (self-loop blocks fetch it too and never use it)
        {
            const float* ws = wbuf_at(c);
#pragma unroll
            for (int t = 0; t < T2; ++t) {
                if constexpr (SP) chain_units<1, 2>(en[t], 0, &hcv, lds_addr(ws) + lane * 16, 1, 2 * t, 0);
                else chain_tile<1>(en[t], &hcv, ws, HC, 32 * t, 0, 4 * lh * HC + lj);
            }
        }
        __syncthreads();
        ++c;
        relu16(hcv);
        if (mkb) MKP(W_HC) = mask16(hcv);
        if (A.save_hc) {
            float* o = A.save_hc + (int64_t)edge * hc;
#pragma unroll
            for (int g = 0; g < 4; ++g) strow<EXACT>(o, 8 * g + 4 * lh, hc, get4(hcv, g), edge_ok);
        }
        float part = 0.f;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 w = *reinterpret_cast<const float4*>(sbias + DE + 32 + 8 * g + 4 * lh);
            part = fmaf(w.x, hcv[4 * g + 0], part);
            part = fmaf(w.y, hcv[4 * g + 1], part);
            part = fmaf(w.z, hcv[4 * g + 2], part);
            part = fmaf(w.w, hcv[4 * g + 3], part);
        }
        const float other = __shfl_xor(part, 32, 64);
        if (A.logits && edge_ok && lh == 0) A.logits[A.perm[edge]] = part + other + A.bc2[0];
    }
#undef wbuf_at
#define wbuf_at(i) (wbuf0 + ((i) & 1) * CHF)
    TS(5);
    if (!flow) return;  // self loops take part in the edge update only (mpn.py:85,91)

    // ---- phase 4: HF^T = relu(Wfe e'^T + Pf[col]) ---------------------------------------------------------
#pragma unroll
    for (int i = 0; i < NCH4; ++i) {
        const bool more = i + 1 < NCH4;
        const int ncw_n = more ? ((HN - (i + 1) * NC4) < NC4 ? (HN - (i + 1) * NC4) : NC4) : 0;
        const float* nsrc = more ? wf1 + (DE * NC4 * SCN / 2) * (i + 1) : wf2;
        const int nn4 = more ? DE * ncw_n / 4 * SCN / 2 : N4_5;
        chunk_fetch<chunk_q(cmax(N4_4_0, N4_5)), SP>(nsrc, nn4, tid, wbuf_at(c + 1));
        const float* ws = wbuf_at(c);
        constexpr int full = NC4 / 32;
        const int ncw = (HN - i * NC4) < NC4 ? (HN - i * NC4) : NC4;  // compile-time per unrolled i
#pragma unroll
        for (int t = 0; t < T2; ++t) {
            if constexpr (SP) {
                if (ncw == NC4) chain_units<full, 2>(en[t], 0, &hf[i * full], lds_addr(ws) + lane * 16, full, 2 * t, 0);
                else chain_units<1, 2>(en[t], 0, &hf[i * full], lds_addr(ws) + lane * 16, 1, 2 * t, 0);
            } else {
                if (ncw == NC4) chain_tile<full>(en[t], &hf[i * full], ws, NC4, 32 * t, 0, 4 * lh * NC4 + lj);
                else chain_tile<1>(en[t], &hf[i * full], ws, 32, 32 * t, 0, 4 * lh * 32 + lj);
            }
        }
        __syncthreads();
        ++c;
    }
    {
        float* o = A.save_hf ? A.save_hf + (int64_t)edge * hn : nullptr;
#pragma unroll
        for (int t = 0; t < TF; ++t) {
            relu16(hf[t]);
            if (o) {
#pragma unroll
                for (int g = 0; g < 4; ++g) strow<EXACT>(o, 32 * t + 8 * g + 4 * lh, hn, get4(hf[t], g), edge_ok);
            }
        }
        if (mkb) {
#pragma unroll
            for (int t = 0; t < TF; t += 2) MKP(W_HF + (t >> 1)) = mask16(hf[t]) | (t + 1 < TF ? mask16(hf[t + 1]) << 16 : 0u);
        }
    }

    TS(6);
    // ---- phase 5: M^T = relu(Wf2 HF^T + bf2) -----------------------------------------------------------------
    f32x16 mm[TD];
#pragma unroll
    for (int t = 0; t < TD; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) set4(mm[t], g, *reinterpret_cast<const float4*>(sbias + DE + 64 + 32 * t + 8 * g + 4 * lh));
#pragma unroll
    for (int i = 0; i < NCH5; ++i) {
        if (i + 1 < NCH5) chunk_fetch<chunk_q(N4_5), SP>(wf2 + (i + 1) * (KC5 * DN * SCN / 2), N4_5, tid, wbuf_at(c + 1));  // static: i is unrolled
        const float* ws = wbuf_at(c);
        if constexpr (SP) chain_units<TD, 2>(hf[i], 0, mm, lds_addr(ws) + lane * 16, TD, 0, 0);
        else chain_tile<TD>(hf[i], mm, ws, DN, 0, 0, 4 * lh * DN + lj);
        if (i + 1 < NCH5) {
            __syncthreads();
            ++c;
        }
    }
    TS(7);
    {
        float* o = A.msg + (int64_t)edge * dn;
#pragma unroll
        for (int t = 0; t < TD; ++t) {
            relu16(mm[t]);
#pragma unroll
            for (int g = 0; g < 4; ++g) strow<EXACT>(o, 32 * t + 8 * g + 4 * lh, dn, get4(mm[t], g), edge_ok);
        }
        if (mkb) {
#pragma unroll
            for (int t = 0; t < TD; t += 2) MKP(W_M + (t >> 1)) = mask16(mm[t]) | (t + 1 < TD ? mask16(mm[t + 1]) << 16 : 0u);
        }
    }
    TS(8);
#undef MKP
}
#undef wbuf_at

// ------------------------------------------------------------------------------------------------------
// Backward chain of one step.  Same machinery, transposed weights: dH^T[k][edge] = sum_n W[n][k] dZ^T[n][edge],
// so the LDS chunk image is W in its native [n][k] layout (rows = contraction index), zero-padded to multiples of 32.
//   B1  dZM = gather(dAGG)[row] (.) [M > 0]                      (node_agg_fn backward, mpn.py:89,96)
//   B2  dZF = (Wf2^T dZM) (.) [HF > 0]
//   B3  dE' = dE_in + Wfe^T dZF
//   B4  dZc = (dlog wc2) (.) [HC > 0];  dE' += Wc1^T dZc;  dZ2 = dE' (.) [e_s > 0]
//   B5  dZ1 = (W2^T dZ2) (.) [H1 > 0]
//   B6  d[e0 | e_{s-1}] = W1e^T dZ1  ->  dE0 += ..., dEprev = ...
template <int T1, int T2, int TF, int TD, bool EXACT, bool SP>
__global__ __launch_bounds__(256, chain_waves(T1)) void edge_chain_bwd_kernel(EdgeChainBwdArgs A) {
    constexpr int HE = 32 * T1, DE = 32 * T2, HN = 32 * TF, DN = 32 * TD, HC = 32;
    constexpr int NR2 = 16;   // B2 chunk: [16 n][HN]
    constexpr int NR3 = 64;   // B3 chunk: [<=64 n][DE]
    constexpr int NR5 = 16;   // B5 chunk: [16 n][HE]
    constexpr int NR6 = 64;   // B6 chunk: [<=64 n][64 k]
    // ---- chunk schedule: [B2 | B3] (flow groups only) B4 B5 B6; float4 counts of the contiguous chunks ------------
    constexpr int NCH2 = DN / NR2, NCH3 = (HN + NR3 - 1) / NR3, NCH5 = DE / NR5, NCH6 = (HE + NR6 - 1) / NR6;
    constexpr int SCN = SP ? 3 : 2;                                // split images: 3/2 the size (edge_chain.hip, split8)
    constexpr int N4_2 = NR2 * HN / 4 * SCN / 2;                   // 16 rows of Wf2
    constexpr int N4_3_0 = cmin(NR3, HN) * DE / 4 * SCN / 2;       // first 64 rows of Wfe
    constexpr int N4_4 = HC * DE / 4 * SCN / 2;                    // Wc1, whole
    constexpr int N4_5 = NR5 * HE / 4 * SCN / 2;                   // 16 rows of W2
    constexpr int N4_6MAX = cmin(NR6, HE) * 64 / 4 * SCN / 2;      // <= 64 rows of one W1e column-pass image [HE][ncol6]
    constexpr int N4_MAX = cmax(cmax(N4_2, N4_3_0), cmax(cmax(N4_4, N4_5), N4_6MAX));
    constexpr int CHF = SP ? 4 * N4_MAX : 1024 * chunk_q(N4_MAX);
    static_assert(CHF <= CH_FLOATS, "chunk too large");
    // mask words per lane (edge_chain.h: chain_mask_words)
    constexpr int W_H1 = 0, W_E = W_H1 + (T1 + 1) / 2, W_HC = W_E + (T2 + 1) / 2, W_HF = W_HC + 1, W_M = W_HF + (TF + 1) / 2,
                  NW = W_M + (TD + 1) / 2;

    // one LDS object: two weight-chunk buffers, then wc2 (zero-padded to 32)
    __shared__ __attribute__((aligned(16))) float smem[2 * CHF + 32];
    float* const wbuf0 = smem;
    float* const swc2 = smem + 2 * CHF;
#define wbuf_at(i) (wbuf0 + ((i) & 1) * CHF)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lj = lane & 31, lh = lane >> 5;
    const int he = A.he, de = A.de, hn = A.hn, dn = A.dn, hc = A.hc;

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
    // Split operands: v_mfma_f32_32x32x16_bf16 adds its products to the accumulator with a small bias toward -infinity
    // (tools/micro/mfma_bias.hip: mean error -0.06 ... -0.11 of the rms error of a six-product fp32 result, the fp32 MFMA
    // +-0.002; negating an operand and the result flips it).  Unbiased rounding noise averages out in the sums the backward
    // takes over edges and steps (bias and weight gradients, segment sums); a bias adds up coherently and is amplified by the
    // step recursion: measured 5e-5 on the cfg-B parameter gradients after 12 steps against 1e-6 in the fp32 mode.  The chain is
    // LINEAR in the gradients (the ReLU masks are bits), so the kernel keeps the gradients of every other edge negated in its
    // registers -- inputs are negated as they are loaded, outputs as they are stored -- and the bias enters neighbouring edges
    // with opposite signs: zero mean over any sum.
    const unsigned sx = SP && (lj & 1) ? 0x80000000u : 0u;
    // dZ rows leave through a per-wave LDS slab as whole 128-byte lines (row_stage.h; tools/micro/store_pattern.hip: the 16-byte
    // pieces of 32 different rows a wave instruction writes straight from the accumulator layout run at 1.65 TB/s, 8 whole lines
    // per instruction at 5.4).  The 128-d template only: 18 KB more LDS per block would cost the narrower ones their third wave.
#ifdef MPNHIP_CHAIN_DIRECT_ROWS
    constexpr bool SLAB = false;     // (A-B build: make EXTRA=-DMPNHIP_CHAIN_DIRECT_ROWS)
#else
    constexpr bool SLAB = T1 >= 10;
#endif
    __shared__ __attribute__((aligned(16))) char rowslab[SLAB ? 4 * ROW_SLAB_BYTES : 16];   // (touched by inline assembly only)
    RowStage rs;
    rs.init(rowslab + (SLAB ? wave * ROW_SLAB_BYTES : 0), lane, tile0 + wave * 32, end);
    // tile t (32 columns) of this wave's rows of a row-major [*, width] matrix; v in the sign the memory image has
    auto store_tile = [&](float* mat, int width, int t, const f32x16& v) {
        if constexpr (SLAB) {
            rs.put32v(v);   // (every tile stored here went through a ReLU / mask / sign flip: VALU results)
            rs.template flush<false>(reinterpret_cast<char*>(mat), (size_t)width * 4, 128 * t, (width - 32 * t) * 4);
        } else {
            float* o2 = mat + (int64_t)edge * width;
#pragma unroll
            for (int g = 0; g < 4; ++g) strow<EXACT>(o2, 32 * t + 8 * g + 4 * lh, width, get4(v, g), edge_ok);
        }
    };
    auto flipped = [&](const f32x16& a) { f32x16 v = a; flip16(v, sx); return v; };
    const int KEp = A.cat_two ? 2 * DE : DE;  // padded columns of [e0 | e_{s-1}] (each half padded to DE)
    const int npass6 = KEp / 64 > 0 ? KEp / 64 : 1;
    const int ncol6 = KEp < 64 ? KEp : 64;    // columns per B6 pass

    const float* wf2 = grp == 1 ? A.wf2_in : A.wf2_out;
    const float* wfe = grp == 1 ? A.wfe_in : A.wfe_out;
    int c = 0;
    TS_INIT();
    TS(0);
    if (flow) {
        chunk_fetch<chunk_q(N4_2), SP>(wf2, N4_2, tid, wbuf_at(0));
    } else {
        chunk_fetch<chunk_q(N4_4), SP>(A.wc1, N4_4, tid, wbuf_at(0));
    }
    if (tid < 32) swc2[tid] = tid < hc ? A.wc2[tid] : 0.f;

    // ---- everything this wave reads from global memory besides the weights goes out NOW, before the first barrier:
    // the ReLU masks (a few words), the gathered aggregate gradient, the incoming edge gradient, the logit gradient.
    unsigned mk[NW];
    {
        const unsigned* mp = A.mask + ((int64_t)(blockIdx.x * 4 + wave) * NW) * 64 + lane;
#pragma unroll
        for (int w = 0; w < NW; ++w) mk[w] = mp[w * 64];
    }
    f32x16 dE[T2];   // gradient w.r.t. e_s arriving from the later step: C-in of the dE' accumulators (B3 / B4)
    auto load_dE = [&]() {
        const float* p = A.dE_io + (int64_t)edge * de;
#pragma unroll
        for (int t = 0; t < T2; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) set4(dE[t], g, ldrow<EXACT>(p, 32 * t + 8 * g + 4 * lh, de));
    };
    // flow blocks fetch it in B2's last chunk (into the registers the consumed dZM tiles free: B2 is the register peak)
    if (!flow) load_dE();
    const float dl = A.dlog[A.perm[edge]];
    f32x16 dzm[TD];
    if (flow) {
        // ---- B1 loads: dZM = gather(dAGG)[row] (node_agg_fn backward, mpn.py:89,96) ----------------------------
        const int row = A.srow[edge];
        const float* da = A.dAGG + (int64_t)row * 2 * dn + (grp == 0 ? dn : 0);
#pragma unroll
        for (int t = 0; t < TD; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) set4(dzm[t], g, ldrow<EXACT>(da, 32 * t + 8 * g + 4 * lh, dn));
        if (A.agg == MPNHIP_AGG_MEAN) {
            const int key = grp * A.N + row;
            const int cnt = A.seg_ptr[key + 1] - A.seg_ptr[key];
            const float scale = (float)(cnt > 0 ? cnt : 1);
#pragma unroll
            for (int t = 0; t < TD; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) dzm[t][r] /= scale;
        }
        if (A.agg == MPNHIP_AGG_MAX) {
            const int* ar = A.ARG + (int64_t)row * 2 * dn + (grp == 0 ? dn : 0);
#pragma unroll
            for (int t = 0; t < TD; ++t)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int n = 32 * t + 8 * g + 4 * lh;
                    const int4 q = *reinterpret_cast<const int4*>(ar + (EXACT || n < dn ? n : 0));
                    float4 v = get4(dzm[t], g);
                    v.x = q.x == edge_raw ? v.x : 0.f; v.y = q.y == edge_raw ? v.y : 0.f;
                    v.z = q.z == edge_raw ? v.z : 0.f; v.w = q.w == edge_raw ? v.w : 0.f;
                    set4(dzm[t], g, v);
                }
        }
    }
    __syncthreads();  // chunk 0 is in wbuf[0]; every load above has landed
    TS(1);

    if (flow) {
        // ---- B1: dZM = gathered gradient (.) [M > 0] -----------------------------------------------------------
        {
#pragma unroll
            for (int t = 0; t < TD; ++t) {
                apply_mask(dzm[t], mk[W_M + (t >> 1)], 16 * (t & 1));
                store_tile(A.dZM, dn, t, dzm[t]);
                if (SP) flip16(dzm[t], sx);
            }
        }
        TS(2);
        // ---- B2: dZF = (Wf2^T dZM) (.) [HF > 0] -------------------------------------------------------------
        f32x16 dzf[TF];
#pragma unroll
        for (int t = 0; t < TF; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) dzf[t][r] = 0.f;
#pragma unroll
        for (int i = 0; i < NCH2; ++i) {
            const bool more = i + 1 < NCH2;  // folds: i is unrolled
            const float* nsrc = more ? wf2 + (i + 1) * (NR2 * HN * SCN / 2) : wfe;
            const int nn4 = more ? N4_2 : N4_3_0;
            chunk_fetch<chunk_q(cmax(N4_2, N4_3_0)), SP>(nsrc, nn4, tid, wbuf_at(c + 1));
            if (i == NCH2 - 1) load_dE();
            // chunk rows = 16 contraction indices n = 16 i .. 16 i + 15 = registers 8 (i & 1) .. + 7 of source tile i / 2
            if constexpr (SP) chain_units<TF, 1>(dzm[i >> 1], i & 1, dzf, lds_addr(wbuf_at(c)) + lane * 16, TF, 0, 0);
            else chain_half<TF>(dzm[i >> 1], (i & 1) * 8, dzf, wbuf_at(c), HN, 4 * lh * HN + lj);
            __syncthreads();
            ++c;
        }
        TS(3);
        {
#pragma unroll
            for (int t = 0; t < TF; ++t) {
                apply_mask(dzf[t], mk[W_HF + (t >> 1)], 16 * (t & 1));
                store_tile(A.dZF, hn, t, flipped(dzf[t]));
            }
        }
        TS(4);
        // ---- B3: dE' += Wfe^T dZF -------------------------------------------------------------------------------
        if (SP) {
#pragma unroll
            for (int t = 0; t < T2; ++t) flip16(dE[t], sx);   // (the incoming gradient joins the registers' sign convention)
        }
#pragma unroll
        for (int i = 0; i < NCH3; ++i) {
            const bool more = i + 1 < NCH3;
            const int rows_n = more ? (HN - (i + 1) * NR3 < NR3 ? HN - (i + 1) * NR3 : NR3) : 0;
            const float* nsrc = more ? wfe + (i + 1) * (NR3 * DE * SCN / 2) : A.wc1;
            const int nn4 = more ? rows_n * DE / 4 * SCN / 2 : N4_4;
            chunk_fetch<chunk_q(cmax(N4_3_0, N4_4)), SP>(nsrc, nn4, tid, wbuf_at(c + 1));
            const float* ws = wbuf_at(c);
            if constexpr (SP) {
                chain_units<T2, 2>(dzf[2 * i], 0, dE, lds_addr(ws) + lane * 16, T2, 0, 0);
                if (2 * i + 1 < TF) chain_units<T2, 2>(dzf[2 * i + 1], 0, dE, lds_addr(ws) + lane * 16, T2, 2, 0);
            } else {
                chain_tile<T2>(dzf[2 * i], dE, ws, DE, 0, 0, 4 * lh * DE + lj);
                if (2 * i + 1 < TF) chain_tile<T2>(dzf[2 * i + 1], dE, ws, DE, 32, 0, 4 * lh * DE + lj);
            }
            __syncthreads();
            ++c;
        }
    }

    TS(5);
    // ---- B4: classifier: dZc = (dlog wc2) (.) [HC > 0];  dE' += Wc1^T dZc ----------------------------------------
    {
        if (SP && !flow) {
#pragma unroll
            for (int t = 0; t < T2; ++t) flip16(dE[t], sx);
        }
        f32x16 dzc;
        const float dls = fxor(dl, sx);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 w = *reinterpret_cast<const float4*>(swc2 + 8 * g + 4 * lh);
            set4(dzc, g, make_float4(dls * w.x, dls * w.y, dls * w.z, dls * w.w));
        }
        apply_mask(dzc, mk[W_HC], 0);
        store_tile(A.dZc, hc, 0, flipped(dzc));
        chunk_fetch<chunk_q(N4_5), SP>(A.w2, N4_5, tid, wbuf_at(c + 1));
        if constexpr (SP) chain_units<T2, 2>(dzc, 0, dE, lds_addr(wbuf_at(c)) + lane * 16, T2, 0, 0);
        else chain_tile<T2>(dzc, dE, wbuf_at(c), DE, 0, 0, 4 * lh * DE + lj);
        __syncthreads();
        ++c;
    }
    // dZ2 = dE' (.) [e_s > 0]  (written over the incoming gradient)
    {
#pragma unroll
        for (int t = 0; t < T2; ++t) {
            apply_mask(dE[t], mk[W_E + (t >> 1)], 16 * (t & 1));
            store_tile(A.dE_io, de, t, flipped(dE[t]));
        }
    }

    TS(6);
    // ---- B5: dZ1 = (W2^T dZ2) (.) [H1 > 0] -----------------------------------------------------------------------
    f32x16 dz1[T1];
#pragma unroll
    for (int t = 0; t < T1; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) dz1[t][r] = 0.f;
    const int rows6_0 = HE < NR6 ? HE : NR6;
    // (skip_e0: the passes of the e0 half -- the first KEp / 128 of them -- are not computed here)
    const int pass6_0 = A.skip_e0 ? npass6 / 2 : 0;
#pragma unroll
    for (int i = 0; i < NCH5; ++i) {
        const bool more = i + 1 < NCH5;
        const float* nsrc = more ? A.w2 + (i + 1) * (NR5 * HE * SCN / 2) : A.w1e + (int64_t)pass6_0 * HE * ncol6 * SCN / 2;
        const int nn4 = more ? N4_5 : rows6_0 * ncol6 / 4 * SCN / 2;
        chunk_fetch<chunk_q(cmax(N4_5, N4_6MAX)), SP>(nsrc, nn4, tid, wbuf_at(c + 1));
        // two sweeps over the same chunk keep the weight staging registers at about T1 / 2 per step
        constexpr int TA = (T1 + 1) / 2, TB = T1 - TA;
        if constexpr (SP) {
            chain_units<T1, 1>(dE[i >> 1], i & 1, dz1, lds_addr(wbuf_at(c)) + lane * 16, T1, 0, 0);
        } else {
            chain_half<TA>(dE[i >> 1], (i & 1) * 8, dz1, wbuf_at(c), HE, 4 * lh * HE + lj);
            if (TB > 0) chain_half<(TB > 0 ? TB : 1)>(dE[i >> 1], (i & 1) * 8, dz1 + TA, wbuf_at(c), HE, 4 * lh * HE + lj + 32 * TA);
        }
        __syncthreads();
        ++c;
    }
    TS(7);
    {
#pragma unroll
        for (int t = 0; t < T1; ++t) {
            apply_mask(dz1[t], mk[W_H1 + (t >> 1)], 16 * (t & 1));
            store_tile(A.dZ1, he, t, flipped(dz1[t]));
        }
    }

    TS(8);
    // ---- B6: d[e0 | e_{s-1}] = W1e^T dZ1, up to 64 (padded) output columns per pass ---------------------------------
    for (int pass = pass6_0; pass < npass6; ++pass) {
        // output tile tt (32 padded columns) of the pass belongs to half tt / T2 of [e0 | e_{s-1}], tile tt % T2 in it;
        // the first half is the re-attached initial features (accumulated over all steps: C-in = the running sum, read
        // here so that the MFMAs hide the load), the second e_{s-1} -- which IS e0 at the first step
        f32x16 dc[2];
        float* dst[2];
        bool live[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int tt = pass * 2 + t;
            live[t] = tt * 32 < KEp;
            const int half = tt / T2, tin = tt % T2;
            const bool to_e0 = (A.cat_two && half == 0) || A.first_step;
            dst[t] = (to_e0 ? A.dE0 : A.dEprev) + (int64_t)edge * de + 32 * tin;
            // (T2 == 1, first step: both tiles of the pass add into the same dE0 columns -- the second starts from zero and
            // is folded into the first after the MFMAs)
            const bool second_of_same = t == 1 && T2 == 1 && A.cat_two && A.first_step;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = 8 * g + 4 * lh;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (live[t] && to_e0 && !second_of_same) v = flip4(ldrow<EXACT>(dst[t], n, de - 32 * tin), sx);
                set4(dc[t], g, v);
            }
        }
#pragma unroll
        for (int i = 0; i < NCH6; ++i) {
            // next: the following <= 64 rows of this pass's image [HE][ncol6], or the first rows of the next pass's
            // image; after the very last chunk the same rows are fetched again and dropped (keeps the load unconditional)
            const bool more = i + 1 < NCH6;
            const bool more_pass = pass + 1 < npass6;
            const int in = more ? i + 1 : (more_pass ? 0 : i);
            const int pn = more ? pass : (more_pass ? pass + 1 : pass);
            const int rows_n = HE - in * NR6 < NR6 ? HE - in * NR6 : NR6;
            const float* nsrc = A.w1e + ((int64_t)pn * HE + in * NR6) * ncol6 * SCN / 2;
            const int nn4 = rows_n * ncol6 / 4 * SCN / 2;
            chunk_fetch<chunk_q(N4_6MAX), SP>(nsrc, nn4, tid, wbuf_at(c + 1));
            const float* ws = wbuf_at(c);
            if constexpr (SP) {
                const unsigned wa = lds_addr(ws) + lane * 16;
                // (the split of a dZ1 tile does not depend on the pass: without this the compiler computes all of them
                // ahead of the pass loop and spills them)
                asm volatile("" : "+v"(dz1[2 * i]));
                if (2 * i + 1 < T1) asm volatile("" : "+v"(dz1[2 * i + 1 < T1 ? 2 * i + 1 : 0]));
                if (ncol6 == 64) {
                    chain_units<2, 2>(dz1[2 * i], 0, dc, wa, 2, 0, 0);
                    if (2 * i + 1 < T1) chain_units<2, 2>(dz1[2 * i + 1], 0, dc, wa, 2, 2, 0);
                } else {
                    chain_units<1, 2>(dz1[2 * i], 0, dc, wa, 1, 0, 0);
                    if (2 * i + 1 < T1) chain_units<1, 2>(dz1[2 * i + 1], 0, dc, wa, 1, 2, 0);
                }
            } else if (ncol6 == 64) {
                chain_tile<2>(dz1[2 * i], dc, ws, 64, 0, 0, 4 * lh * 64 + lj);
                if (2 * i + 1 < T1) chain_tile<2>(dz1[2 * i + 1], dc, ws, 64, 32, 0, 4 * lh * 64 + lj);
            } else {
                chain_tile<1>(dz1[2 * i], dc, ws, 32, 0, 0, 4 * lh * 32 + lj);
                if (2 * i + 1 < T1) chain_tile<1>(dz1[2 * i + 1], dc, ws, 32, 32, 0, 4 * lh * 32 + lj);
            }
            __syncthreads();
            ++c;
        }
        if (T2 == 1 && A.cat_two && A.first_step && live[1]) {
#pragma unroll
            for (int r = 0; r < 16; ++r) dc[0][r] += dc[1][r];
            live[1] = false;
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            if (!live[t]) break;
            const int tin = (pass * 2 + t) % T2;
#pragma unroll
            for (int g = 0; g < 4; ++g) strow<EXACT>(dst[t], 8 * g + 4 * lh, de - 32 * tin, flip4(get4(dc[t], g), sx), edge_ok);
        }
    }
    TS(9);
#undef wbuf_at
}

// dst[r][c] = (r < rows && c < cols) ? src[r * lds + c0 + c] : 0   for r < rows_pad, c < cols_pad (ld = cols_pad)
__global__ void k_pack_padded(const float* __restrict__ src, int64_t lds, int c0, int rows, int cols, float* __restrict__ dst,
                              int rows_pad, int cols_pad, int ldd, int dst_c0) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)rows_pad * cols_pad) return;
    const int r = (int)(i / cols_pad), c = (int)(i % cols_pad);
    dst[(int64_t)r * ldd + dst_c0 + c] = (r < rows && c < cols) ? src[(int64_t)r * lds + c0 + c] : 0.f;
}

// dst[k][n] = (k < k_cols && n < n_rows) ? W[n * ldw + k0 + k] : 0   for k < k_pad, n < n_pad  (transposed, padded)
__global__ void k_transpose_padded(const float* __restrict__ W, int64_t ldw, int k0, int n_rows, int k_cols,
                                   float* __restrict__ WT, int n_pad, int k_pad) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)n_pad * k_pad) return;
    const int k = (int)(i / n_pad), n = (int)(i % n_pad);
    WT[i] = (k < k_cols && n < n_rows) ? W[(int64_t)n * ldw + k0 + k] : 0.f;
}

// Split image (see the comment at split8): thread = one element of one unit; writes its three pieces.
__device__ __forceinline__ void pack_split_one(const SplitOp& o, int64_t idx) {
    const int ntr = o.Np / 32;
    if (idx >= (int64_t)(o.Kp / 16) * ntr * 512) return;
    const int i = (int)(idx & 7), lane = (int)((idx >> 3) & 63);
    const int64_t unit = idx >> 9;
    const int t = (int)(unit % ntr), kb = (int)(unit / ntr);
    const int k = 16 * kb + (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5), n = 32 * t + (lane & 31);
    const float x = (k < o.K && n < o.N) ? o.src[k * o.sk + n * o.sn] : 0.f;
    const __bf16 h = (__bf16)x;
    const float r1 = x - (float)h;
    const __bf16 m = (__bf16)r1;
    const __bf16 l = (__bf16)(r1 - (float)m);
    unsigned short* q = o.dst + ((int64_t)kb * o.ntr_image + o.t0 + t) * 3 * 512 + lane * 8 + i;
    q[0] = __builtin_bit_cast(unsigned short, h);
    q[512] = __builtin_bit_cast(unsigned short, m);
    q[1024] = __builtin_bit_cast(unsigned short, l);
}
__global__ void k_pack_split(SplitOp o) { pack_split_one(o, (int64_t)blockIdx.x * blockDim.x + threadIdx.x); }
// blockIdx.y = op (static indices into the by-value table: a dynamic one would move it to scratch memory)
__global__ __launch_bounds__(256) void k_pack_split_multi(SplitBatch b) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    switch (blockIdx.y) {
#define MPN_SPLIT_CASE(k) case k: pack_split_one(b.op[k], i); break;
        MPN_SPLIT_CASE(0) MPN_SPLIT_CASE(1) MPN_SPLIT_CASE(2) MPN_SPLIT_CASE(3) MPN_SPLIT_CASE(4) MPN_SPLIT_CASE(5) MPN_SPLIT_CASE(6) MPN_SPLIT_CASE(7)
        MPN_SPLIT_CASE(8) MPN_SPLIT_CASE(9) MPN_SPLIT_CASE(10) MPN_SPLIT_CASE(11) MPN_SPLIT_CASE(12) MPN_SPLIT_CASE(13) MPN_SPLIT_CASE(14) MPN_SPLIT_CASE(15)
#undef MPN_SPLIT_CASE
        default: break;
    }
}

static thread_local SplitBatch* g_split_batch = nullptr;
void split_batch_begin(SplitBatch* b) {
    b->n = 0;
    g_split_batch = getenv("MPNHIP_NO_PACK_BATCH") ? nullptr : b;
}
void split_batch_abort() { g_split_batch = nullptr; }
int split_batch_flush(hipStream_t s) {
    SplitBatch* b = g_split_batch;
    g_split_batch = nullptr;
    if (!b || b->n == 0) return MPNHIP_OK;
    int64_t mx = 0;
    for (int i = 0; i < b->n; ++i) {
        const int64_t n = (int64_t)(b->op[i].Kp / 16) * (b->op[i].Np / 32) * 512;
        mx = n > mx ? n : mx;
    }
    hipLaunchKernelGGL(k_pack_split_multi, dim3((unsigned)((mx + 255) / 256), b->n), dim3(256), 0, s, *b);
    MPN_LAUNCH_CHECK();
    return MPNHIP_OK;
}

int pack_split(const float* src, int64_t sk, int64_t sn, int K, int N, int Kp, int Np, float* dst, hipStream_t s, int ntr_image, int t0) {
    if (ntr_image <= 0) ntr_image = Np / 32;
    const int64_t n = (int64_t)(Kp / 16) * (Np / 32) * 512;
    if (n <= 0) return MPNHIP_OK;
    if (Kp % 16 != 0 || Np % 32 != 0) { set_error("pack_split: padded sizes must be multiples of 16 x 32"); return MPNHIP_ERR_ARG; }
    const SplitOp o = {src, sk, sn, K, N, Kp, Np, reinterpret_cast<unsigned short*>(dst), ntr_image, t0};
    if (g_split_batch && g_split_batch->n < SplitBatch::MAX) {
        g_split_batch->op[g_split_batch->n++] = o;
        return MPNHIP_OK;
    }
    hipLaunchKernelGGL(k_pack_split, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, o);
    MPN_LAUNCH_CHECK();
    return MPNHIP_OK;
}

int transpose_padded(const float* W, int64_t ldw, int k0, int n_rows, int k_cols, float* WT, int n_pad, int k_pad, hipStream_t s) {
    const int64_t n = (int64_t)n_pad * k_pad;
    if (n <= 0) return MPNHIP_OK;
    if (pack_batch_add({W, WT, ldw, k0, k_cols, n_rows, k_pad, n_pad, n_pad, 0, 1})) return MPNHIP_OK;
    hipLaunchKernelGGL(k_transpose_padded, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, W, ldw, k0, n_rows, k_cols, WT, n_pad, k_pad);
    MPN_LAUNCH_CHECK();
    return MPNHIP_OK;
}

int pack_padded(const float* src, int64_t lds, int c0, int rows, int cols, float* dst, int rows_pad, int cols_pad, int ldd,
                int dst_c0, hipStream_t s) {
    const int64_t n = (int64_t)rows_pad * cols_pad;
    if (n <= 0) return MPNHIP_OK;
    if (pack_batch_add({src, dst, lds, c0, rows, cols, rows_pad, cols_pad, ldd, dst_c0, 0})) return MPNHIP_OK;
    hipLaunchKernelGGL(k_pack_padded, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, src, lds, c0, rows, cols, dst, rows_pad,
                       cols_pad, ldd, dst_c0);
    MPN_LAUNCH_CHECK();
    return MPNHIP_OK;
}

#ifdef MPNHIP_CHAIN_TS
// stamp buffer of one kernel kind; the 40th launch is synchronised and dumped to "<MPNHIP_CHAIN_TS>_<name>.txt"
struct StampDump {
    long long* buf = nullptr;
    size_t cap = 0;
    int launches = 0;
    long long* prepare(unsigned blocks, hipStream_t s) {
        if (!getenv("MPNHIP_CHAIN_TS")) return nullptr;
        const size_t need = (size_t)blocks * 4 * 16 * sizeof(long long);
        if (need > cap) { if (buf) (void)hipFree(buf); if (hipMalloc(&buf, need) != hipSuccess) return nullptr; cap = need; }
        (void)hipMemsetAsync(buf, 0, need, s);
        return buf;
    }
    void finish(const char* name, unsigned blocks, hipStream_t s) {
        if (!buf || !getenv("MPNHIP_CHAIN_TS") || ++launches != 40) return;
        (void)hipStreamSynchronize(s);
        std::vector<long long> h((size_t)blocks * 64);
        (void)hipMemcpy(h.data(), buf, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
        std::string path = std::string(getenv("MPNHIP_CHAIN_TS")) + "_" + name + ".txt";
        if (FILE* f = fopen(path.c_str(), "w")) {
            for (unsigned w = 0; w < blocks * 4; ++w) {
                for (int i = 0; i < 16; ++i) fprintf(f, "%lld ", h[(size_t)w * 16 + i]);
                fprintf(f, "\n");
            }
            fclose(f);
        }
    }
};
static StampDump g_stamp_fwd, g_stamp_bwd;
#endif

// ---- the edge encoder at the wider models' dims (MLPGraphIndependent, mpn.py:355 / :164-178; mlp.py:27: three Linear + ReLU layers,
// e.g. 6 -> 72 -> 72 -> 64 at d = 128) in ONE launch -- round 6.  As three launches of the GEMM kernels (K = 6 FMA kernel 20 us, two
// [50000 x 72] products 20 + 16 us at cfg-B) the layers were all launch and latency: 14 MB of activations each.  Here a wave carries 32
// edges through the three layers with the chain kernels' transposed products (edges on the MFMA lane dimension, a layer's accumulator tile
// is the next layer's B operand; fp32 MFMAs -- exact fp32 in every operand form), the three weight matrices sit in LDS as zero-padded
// [k][n] images built by the block itself from the nn.Linear [out][in] rows (68 KB at T = 3 / 3 / 2: two blocks per CU), hidden
// activations leave only when the backward needs them.  T1 / T2 / TO = 32-wide tiles of the two hidden widths and the output.
template <int T1, int T2, int TO>
__global__ __launch_bounds__(256, 2) void k_edge_encoder_mfma(const float* __restrict__ x, const int* __restrict__ idx, int64_t rows, int in_dim,
                                                              const float* __restrict__ w0, const float* __restrict__ b0, int h1,
                                                              const float* __restrict__ w1, const float* __restrict__ b1, int h2,
                                                              const float* __restrict__ w2, const float* __restrict__ b2, int od,
                                                              float* __restrict__ h1_out, float* __restrict__ h2_out, float* __restrict__ y) {
    constexpr int N1 = 32 * T1, N2 = 32 * T2, NO = 32 * TO;
    __shared__ __attribute__((aligned(16))) float smem[16 * N1 + N1 * N2 + N2 * NO + N1 + N2 + NO];
    float* const w0t = smem;                      // [16 k][N1]
    float* const w1t = w0t + 16 * N1;             // [N1 k][N2]
    float* const w2t = w1t + N1 * N2;             // [N2 k][NO]
    float* const sb = w2t + N2 * NO;              // biases b0 | b1 | b2, zero-padded
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lj = lane & 31, lh = lane >> 5;
    // images: zero, then the live [k < K][n < N] part transposed out of the row-major [N][K] weights (reads along k are contiguous)
    for (int i = tid; i < 16 * N1 + N1 * N2 + N2 * NO + N1 + N2 + NO; i += 256) smem[i] = 0.f;
    __syncthreads();
    for (int i = tid; i < h1 * in_dim; i += 256) { const int n = i / in_dim, k = i - n * in_dim; w0t[k * N1 + n] = w0[i]; }
    for (int i = tid; i < h2 * h1; i += 256) { const int n = i / h1, k = i - n * h1; w1t[k * N2 + n] = w1[i]; }
    for (int i = tid; i < od * h2; i += 256) { const int n = i / h2, k = i - n * h2; w2t[k * NO + n] = w2[i]; }
    for (int i = tid; i < h1; i += 256) sb[i] = b0[i];
    for (int i = tid; i < h2; i += 256) sb[N1 + i] = b1[i];
    for (int i = tid; i < od; i += 256) sb[N1 + N2 + i] = b2[i];
    __syncthreads();
    const int64_t e_raw = (int64_t)blockIdx.x * 128 + wave * 32 + lj;
    const bool ok = e_raw < rows;
    const int64_t e = ok ? e_raw : rows - 1;
    // layer-0 input in the accumulator layout: register r of lane (j, h) = feature (r & 3) + 8 (r >> 2) + 4 h of edge j (in_dim <= 16)
    f32x16 src;
    {
        const float* xr = x + (int64_t)(idx ? idx[e] : e) * in_dim;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int k = (r & 3) + 8 * (r >> 2) + 4 * lh;
            src[r] = (r < 8 && k < in_dim) ? xr[k < in_dim ? k : 0] : 0.f;
        }
    }
    // (tile counts as template arguments: a run-time bound would index the register arrays dynamically, i.e. put them in scratch memory)
    auto bias_tiles = [&](auto nt_tag, f32x16* a, const float* b) {
        constexpr int NT = decltype(nt_tag)::value;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) set4(a[t], g, *reinterpret_cast<const float4*>(b + 32 * t + 8 * g + 4 * lh));
    };
    auto store_rows = [&](auto nt_tag, float* out, int width, const f32x16* a) {
        constexpr int NT = decltype(nt_tag)::value;
        float* o = out + e * width;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) strow<false>(o, 32 * t + 8 * g + 4 * lh, width, get4(a[t], g), ok);
    };
    using I1 = std::integral_constant<int, T1>;
    using I2 = std::integral_constant<int, T2>;
    using IO = std::integral_constant<int, TO>;
    f32x16 a1[T1], a2[T2], a3[TO];
    bias_tiles(I1{}, a1, sb);
    chain_half<T1>(src, 0, a1, w0t, N1, 4 * lh * N1 + lj);
#pragma unroll
    for (int t = 0; t < T1; ++t) relu16(a1[t]);
    if (h1_out) store_rows(I1{}, h1_out, h1, a1);
    bias_tiles(I2{}, a2, sb + N1);
#pragma unroll
    for (int t = 0; t < T1; ++t) chain_tile<T2>(a1[t], a2, w1t, N2, 32 * t, 0, 4 * lh * N2 + lj);
#pragma unroll
    for (int t = 0; t < T2; ++t) relu16(a2[t]);
    if (h2_out) store_rows(I2{}, h2_out, h2, a2);
    bias_tiles(IO{}, a3, sb + N1 + N2);
#pragma unroll
    for (int t = 0; t < T2; ++t) chain_tile<TO>(a2[t], a3, w2t, NO, 32 * t, 0, 4 * lh * NO + lj);
#pragma unroll
    for (int t = 0; t < TO; ++t) relu16(a3[t]);
    store_rows(IO{}, y, od, a3);
}

// 1: launched; 0: not this kernel's shape (three layers, in_dim <= 16, hidden widths <= 96, output <= 64, every width a multiple of 4,
// 16-byte aligned outputs); < 0: error
int launch_edge_encoder_mfma(const float* x, const int* idx, int64_t rows, int in_dim, const float* const w[3], const float* const b[3],
                             const int dims[3], float* h1_out, float* h2_out, float* y, hipStream_t s) {
    if (rows <= 0 || in_dim < 1 || in_dim > 16 || dims[0] > 96 || dims[1] > 96 || dims[2] > 64 || dims[0] % 4 || dims[1] % 4 || dims[2] % 4 ||
        dims[0] < 4 || dims[1] < 4 || dims[2] < 4 || getenv("MPNHIP_NO_ENCODER_MFMA"))
        return 0;
    auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    if (!al16(y) || (h1_out && !al16(h1_out)) || (h2_out && !al16(h2_out))) return 0;
    const int t1 = (dims[0] + 31) / 32, t2 = (dims[1] + 31) / 32, to = (dims[2] + 31) / 32;
    const dim3 grid((unsigned)((rows + 127) / 128)), block(256);
#define MPN_EE(A, B, C) hipLaunchKernelGGL((k_edge_encoder_mfma<A, B, C>), grid, block, 0, s, x, idx, rows, in_dim, w[0], b[0], dims[0], w[1], b[1], \
                                          dims[1], w[2], b[2], dims[2], h1_out, h2_out, y)
    if (t1 == 3 && t2 == 3 && to == 2) MPN_EE(3, 3, 2);
    else if (t1 == 2 && t2 == 2 && to == 1) MPN_EE(2, 2, 1);
    else if (t1 <= 3 && t2 <= 3 && to <= 2) MPN_EE(3, 3, 2);   // (narrower widths ride on the widest instantiation: zero-padded tiles)
    else return 0;
#undef MPN_EE
    if (hipGetLastError() != hipSuccess) { set_error("edge encoder (mfma): launch failed"); return MPNHIP_ERR_HIP; }
    return 1;
}

static int chain_variant(int he, int de, int hn, int dn) {
    const int t1 = (he + 31) / 32, t2 = (de + 31) / 32, tf = (hn + 31) / 32, td = (dn + 31) / 32;
    if (t1 == 10 && t2 == 2 && tf == 7 && td == 4) return 128;
    if (t1 == 5 && t2 == 1 && tf == 4 && td == 2) return 64;
    if (t1 == 3 && t2 == 1 && tf == 2 && td == 1) return 32;
    return 0;
}

bool edge_chain_supported(int he, int de, int hn, int dn, int hc, int k1a, int k1b) {
    return chain_variant(he, de, hn, dn) != 0 && hc >= 4 && hc <= 32 && he % 4 == 0 && de % 4 == 0 && hn % 4 == 0 && dn % 4 == 0 &&
           hc % 4 == 0 && (k1a % 16 == 0) && (k1b % 16 == 0) && (k1a + k1b) >= 16;
}

int launch_edge_chain(const EdgeChainArgs& a_in, hipStream_t s) {
    if (a_in.E <= 0) return MPNHIP_OK;
    EdgeChainArgs a = a_in;
    // rows of P / Q0 / save_h1 are addressed as base + unsigned 32-bit element offset
    if ((int64_t)a.E * a.he >= ((int64_t)1 << 32) || (int64_t)a.N * a.pw >= ((int64_t)1 << 32)) {
        set_error("edge_chain: graph too large for 32-bit row offsets (E * he or N * pw >= 2^32)");
        return MPNHIP_ERR_UNSUPPORTED;
    }
    const unsigned blocks = (unsigned)((a.E + 127) / 128 + 3);
    if (getenv("MPNHIP_CHAIN_ABLATE_COL")) a.scol = a.srow;   // timing ablation: the col-side gathers read the (sorted) row's table row; results wrong
    count_path(a.split ? PC_CHAIN_FWD_SPLIT : PC_CHAIN_FWD);
#ifdef MPNHIP_CHAIN_TS
    a.ts = g_stamp_fwd.prepare(blocks, s);
#endif
    const bool exact = a.he % 32 == 0 && a.de % 32 == 0 && a.hn % 32 == 0 && a.dn % 32 == 0 && a.hc == 32;
    switch (chain_variant(a.he, a.de, a.hn, a.dn)) {
        case 128:
            if (a.split && exact)
                MPN_LAUNCH_PROFILED((edge_chain_kernel<10, 2, 7, 4, true, true>), dim3(blocks), dim3(256), s, a);
            else if (a.split)
                MPN_LAUNCH_PROFILED((edge_chain_kernel<10, 2, 7, 4, false, true>), dim3(blocks), dim3(256), s, a);
            else if (exact)
                MPN_LAUNCH_PROFILED((edge_chain_kernel<10, 2, 7, 4, true, false>), dim3(blocks), dim3(256), s, a);
            else
                MPN_LAUNCH_PROFILED((edge_chain_kernel<10, 2, 7, 4, false, false>), dim3(blocks), dim3(256), s, a);
            break;
        case 64:
            if (a.split) MPN_LAUNCH_PROFILED((edge_chain_kernel<5, 1, 4, 2, false, true>), dim3(blocks), dim3(256), s, a);
            else MPN_LAUNCH_PROFILED((edge_chain_kernel<5, 1, 4, 2, false, false>), dim3(blocks), dim3(256), s, a);
            break;
        case 32:
            if (a.split) MPN_LAUNCH_PROFILED((edge_chain_kernel<3, 1, 2, 1, false, true>), dim3(blocks), dim3(256), s, a);
            else MPN_LAUNCH_PROFILED((edge_chain_kernel<3, 1, 2, 1, false, false>), dim3(blocks), dim3(256), s, a);
            break;
        default: set_error("edge_chain: unsupported widths"); return MPNHIP_ERR_UNSUPPORTED;
    }
#ifdef MPNHIP_CHAIN_TS
    g_stamp_fwd.finish("fwd", blocks, s);
#endif
    MPN_LAUNCH_CHECK();
    return MPNHIP_OK;
}

int launch_edge_chain_bwd(const EdgeChainBwdArgs& a_in, hipStream_t s) {
    if (a_in.E <= 0) return MPNHIP_OK;
    EdgeChainBwdArgs a = a_in;
    const unsigned blocks = (unsigned)((a.E + 127) / 128 + 3);
    count_path(a.split ? PC_CHAIN_BWD_SPLIT : PC_CHAIN_BWD);
#ifdef MPNHIP_CHAIN_TS
    a.ts = g_stamp_bwd.prepare(blocks, s);
#endif
    const bool exact = a.he % 32 == 0 && a.de % 32 == 0 && a.hn % 32 == 0 && a.dn % 32 == 0 && a.hc == 32;
    switch (chain_variant(a.he, a.de, a.hn, a.dn)) {
        case 128:
            if (a.split && exact)
                MPN_LAUNCH_PROFILED((edge_chain_bwd_kernel<10, 2, 7, 4, true, true>), dim3(blocks), dim3(256), s, a);
            else if (a.split)
                MPN_LAUNCH_PROFILED((edge_chain_bwd_kernel<10, 2, 7, 4, false, true>), dim3(blocks), dim3(256), s, a);
            else if (exact)
                MPN_LAUNCH_PROFILED((edge_chain_bwd_kernel<10, 2, 7, 4, true, false>), dim3(blocks), dim3(256), s, a);
            else
                MPN_LAUNCH_PROFILED((edge_chain_bwd_kernel<10, 2, 7, 4, false, false>), dim3(blocks), dim3(256), s, a);
            break;
        case 64:
            if (a.split) MPN_LAUNCH_PROFILED((edge_chain_bwd_kernel<5, 1, 4, 2, false, true>), dim3(blocks), dim3(256), s, a);
            else MPN_LAUNCH_PROFILED((edge_chain_bwd_kernel<5, 1, 4, 2, false, false>), dim3(blocks), dim3(256), s, a);
            break;
        case 32:
            if (a.split) MPN_LAUNCH_PROFILED((edge_chain_bwd_kernel<3, 1, 2, 1, false, true>), dim3(blocks), dim3(256), s, a);
            else MPN_LAUNCH_PROFILED((edge_chain_bwd_kernel<3, 1, 2, 1, false, false>), dim3(blocks), dim3(256), s, a);
            break;
        default: set_error("edge_chain_bwd: unsupported widths"); return MPNHIP_ERR_UNSUPPORTED;
    }
#ifdef MPNHIP_CHAIN_TS
    g_stamp_bwd.finish("bwd", blocks, s);
#endif
    MPN_LAUNCH_CHECK();
    return MPNHIP_OK;
}

}  // namespace mpnhip
