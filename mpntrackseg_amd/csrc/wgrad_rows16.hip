// Weight-gradient products over bf16 ROWS at the widths of BASELINE.json configs[4] (256-d: 640 / 448 / 256 / 128 columns) --
// mlp.py:27-28 under autograd in the bf16-operand mode:  dW[o, c] += sum_m dZ[m, o] H[m, c],  db[o] += sum_m dZ[m, o].
//
// The row-panel kernel (wgrad_panel.hip) reads these operands through registers in 8-byte pieces, one 16-row stage ahead, and
// needs 2 x 2 output tiles for 640 x 128 / 256 x 448 (every operand row fetched 1.6 times: 0.30 of HBM at cfg-E).
// bf16 rows need no arithmetic on their way in, so here NOTHING passes through registers:
//   * one 512-thread block per CU (8 waves as WO x WC) owns a row chunk and the WHOLE output where it fits the accumulators
//     (640 x 128, 128 x 640, 448 x 128: ten / seven 32 x 32 tiles per wave) -- every operand row is fetched once; 256 x 448 takes
//     two column tiles (256 x 256 + 256 x 192: the dZ rows twice, 1.36 x instead of 2.1 x);
//   * a stage = 16 rows of dZ and of H, each operand its own LDS image with a row pitch == 64 (mod 256) bytes (the transposing
//     reads of a half-wave -- 4 rows x 64 bytes -- cover the 64 banks once); rows land by LDS-DMA (global_load_lds_dwordx4,
//     16 bytes per lane, one row per wave instruction, lanes past the row's end masked), a ring of NST stages, NST - 1 ahead;
//   * one counted s_waitcnt vmcnt + s_barrier per stage; MFMA operands by ds_read_b64_tr_b16 (inline assembly: a
//     compiler-visible read of a DMA target draws vmcnt(0)); chunks with an odd index multiply NEGATED operands (the sign is
//     applied to the narrower operand's registers after the transposing read) and are subtracted by the slab sum: the bf16
//     MFMA's accumulate bias cancels across neighbouring chunks;
//   * the bias gradient: n_out / 4 threads sum their four columns of the dZ image, 16 rows per stage, in fp32;
//   * slabs and the fixed-order slab sum are the row-panel kernel's (wgrad_reduce_kernel): bitwise reproducible.
#include <cstdlib>

#include "common.h"

namespace mpnhip {

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

constexpr int R16_NT = 512;
constexpr int R16_KB = 16;      // rows per stage = one k block of the bf16 MFMA
constexpr int R16_NST = 4;      // stages in the ring

// row pitch (bytes) of an image of W bf16 columns: >= 2 W and == 64 (mod 256)
constexpr int r16_pitch(int W) { return ((2 * W - 64 + 255) / 256) * 256 + 64; }

// One block: job J, output tile (`tile_o`: BO dZ columns = output rows, `tile_c`: BC H columns), row chunk `by`.
template <int WO, int WC, int TM, int TN>
__device__ __forceinline__ void r16_block(const WpJob& J, const int tile_o, const int tile_c, const int by, char* lds, const int dbg) {
    static_assert(WO * WC == 8, "8 waves");
    constexpr int BO = 32 * TM * WO, BC = 32 * TN * WC;
    constexpr int PZ = r16_pitch(BO), PH = r16_pitch(BC);
    constexpr int IMGH = R16_KB * PZ;                    // offset of the H image inside a stage
    constexpr int STAGE = R16_KB * (PZ + PH);
    constexpr int NPZ = (BO / 8 + 63) / 64, NPH = (BC / 8 + 63) / 64;   // 1 KiB pieces per row
    constexpr int NPW = 2 * (NPZ + NPH);                 // DMA instructions per wave and stage (two rows per wave)
    constexpr bool NEG_Z = BO <= BC;                     // the sign of odd chunks goes onto the narrower operand
    static_assert(R16_NST * STAGE <= 160 * 1024, "LDS");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wo = wave / WC, wc = wave % WC;

    const int rb = J.row_begin ? *J.row_begin : 0;
    const int re = J.row_end ? *J.row_end : (int)J.m_static;
    const int batch = by / J.nsplit, ci = by - batch * J.nsplit;
    const int r0 = rb + ci * J.chunk;
    int r1 = r0 + J.chunk;
    r1 = r1 < re ? r1 : re;
    if (r0 >= r1) return;   // empty chunk: the slab sum skips it too
    const bool odd = ci & 1;

    const int c0 = tile_c * BC, o0 = tile_o * BO;
    const int wz = J.n_out - o0 < BO ? J.n_out - o0 : BO;    // live dZ columns of this tile, a multiple of 8
    const int wh = J.k_in - c0 < BC ? J.k_in - c0 : BC;      // live H columns of this tile, a multiple of 8
    const unsigned short* const zsrc = reinterpret_cast<const unsigned short*>(J.dZ) + (int64_t)batch * J.z_bstride + o0;
    const unsigned short* const hsrc = reinterpret_cast<const unsigned short*>(J.H) + (int64_t)batch * J.h_bstride + c0;
    const int64_t ldz = J.ldz, ldh = J.ldh;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)lds;

    // rows (wave) and (wave + 8) of stage `st` (first row m0) into ring slot `slot`; rows >= nrows are not loaded
    auto issue = [&](int m0, int slot, int nrows) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int row = wave + 8 * h;
            if (row < nrows) {   // (wave-uniform)
                const unsigned short* zr = zsrc + (int64_t)(m0 + row) * ldz;
                const unsigned short* hr = hsrc + (int64_t)(m0 + row) * ldh;
                char* const dz = lds + slot * STAGE + row * PZ;
                char* const dh = lds + slot * STAGE + IMGH + row * PH;
#pragma unroll
                for (int p = 0; p < NPZ; ++p) {
                    const int ch = 64 * p + lane;
                    if (ch * 8 < wz)
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(zr + ch * 8),
                                                         (__attribute__((address_space(3))) void*)(dz + p * 1024), 16, 0, 0);
                }
#pragma unroll
                for (int p = 0; p < NPH; ++p) {
                    const int ch = 64 * p + lane;
                    if (ch * 8 < wh)
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(hr + ch * 8),
                                                         (__attribute__((address_space(3))) void*)(dh + p * 1024), 16, 0, 0);
                }
            }
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    float bsum[4] = {0.f, 0.f, 0.f, 0.f};
    const bool bias_thread = tile_c == 0 && tid * 4 < wz;

    // transposing-read lane geometry: lane 4q + p of a 16-lane group supplies row q, columns 4p .. 4p+3 of the group's block
    const int lh = lane >> 5, gi = (lane >> 4) & 1, lq = (lane & 15) >> 2, lp = lane & 3;
    const unsigned rda = lds0 + (unsigned)((8 * lh + lq) * PZ + (16 * gi + 4 * lp) * 2 + 32 * (wo * TM) * 2);
    const unsigned rdb = lds0 + (unsigned)(IMGH + (8 * lh + lq) * PH + (16 * gi + 4 * lp) * 2 + 32 * (wc * TN) * 2);
    const unsigned rbias = lds0 + (unsigned)(tid * 8);
    const unsigned sgn = odd ? 0x80008000u : 0u;

    auto frag = [&](s16x4 lo, s16x4 hi, bool negate) {
        if (negate) {
            typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
            u32x2 a = __builtin_bit_cast(u32x2, lo), b = __builtin_bit_cast(u32x2, hi);
            a[0] ^= sgn; a[1] ^= sgn; b[0] ^= sgn; b[1] ^= sgn;
            lo = __builtin_bit_cast(s16x4, a); hi = __builtin_bit_cast(s16x4, b);
        }
        const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        return __builtin_bit_cast(bf16x8, v);
    };
    // the products of one stage in ring slot `slot` (its rows all landed and every wave past the barrier).  Operand fragments are
    // fetched in groups of at most five tiles (20 registers): ten at once, next to 160 accumulator registers, spill -- and a scratch
    // access counts in vmcnt like the DMA pieces do
    auto products = [&](unsigned so) {
        if (bias_thread) {
            // four columns x 16 rows of the dZ image (un-negated: the chunk's sign is applied once, at the end)
#pragma unroll
            for (int r4 = 0; r4 < R16_KB; r4 += 4) {
                uint2 v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    asm volatile("ds_read_b64 %0, %1" : "=v"(v[r]) : "v"(rbias + so + (unsigned)((r4 + r) * PZ)) : "memory");
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3])::"memory");
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    bsum[0] += __uint_as_float(v[r].x << 16); bsum[1] += __uint_as_float(v[r].x & 0xffff0000u);
                    bsum[2] += __uint_as_float(v[r].y << 16); bsum[3] += __uint_as_float(v[r].y & 0xffff0000u);
                }
            }
        }
        constexpr bool A_OUTER = TM >= TN;            // the longer tile dimension is walked in groups
        constexpr int NI = A_OUTER ? TN : TM;         // tiles of the operand held whole (1 or 2)
        constexpr int NO = A_OUTER ? TM : TN;
        constexpr int PI = A_OUTER ? PH : PZ, PO = A_OUTER ? PZ : PH;
        const unsigned ri = (A_OUTER ? rdb : rda) + so, ro = (A_OUTER ? rda : rdb) + so;
        s16x4 ilo[NI], ihi[NI];
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(ilo[j]) : "v"(ri + (unsigned)(64 * j)) : "memory");
            asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(ihi[j]) : "v"(ri + (unsigned)(64 * j + 4 * PI)) : "memory");
        }
        bf16x8 fi[NI];
        bool first = true;
#pragma unroll
        for (int g0 = 0; g0 < NO; g0 += 5) {
            constexpr int GMAX = 5;
            s16x4 olo[GMAX], ohi[GMAX];
#pragma unroll
            for (int u = 0; u < GMAX; ++u)
                if (g0 + u < NO) {
                    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(olo[u]) : "v"(ro + (unsigned)(64 * (g0 + u))) : "memory");
                    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(ohi[u]) : "v"(ro + (unsigned)(64 * (g0 + u) + 4 * PO)) : "memory");
                }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);   // (nothing register-only is hoisted above the wait)
            if (first) {
#pragma unroll
                for (int j = 0; j < NI; ++j) {
                    asm volatile("" : "+v"(ilo[j]), "+v"(ihi[j]));
                    fi[j] = frag(ilo[j], ihi[j], A_OUTER ? !NEG_Z : NEG_Z);
                }
                first = false;
            }
#pragma unroll
            for (int u = 0; u < GMAX; ++u)
                if (g0 + u < NO) {
                    asm volatile("" : "+v"(olo[u]), "+v"(ohi[u]));
                    const bf16x8 fo = frag(olo[u], ohi[u], A_OUTER ? NEG_Z : !NEG_Z);
#pragma unroll
                    for (int j = 0; j < NI; ++j) {
                        if constexpr (A_OUTER) acc[g0 + u][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fo, fi[j], acc[g0 + u][j], 0, 0, 0);
                        else acc[j][g0 + u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fi[j], fo, acc[j][g0 + u], 0, 0, 0);
                    }
                }
        }
    };

    const int nfull = (r1 - r0) / R16_KB;
    const int tail = (r1 - r0) - nfull * R16_KB;
#pragma unroll
    for (int q = 0; q < R16_NST - 1; ++q)
        if (q < nfull) issue(r0 + q * R16_KB, q, R16_KB);
    int slot = 0;
    for (int st = 0; st < nfull; ++st) {
        // stage st has landed: every wave waits for its own rows of it (those of the NST - 2 following stages may stay in
        // flight), then the barrier -- which also says every wave is done with stage st - 1, whose slot stage st + NST - 1 refills
        if (st + R16_NST - 2 < nfull) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPW * (R16_NST - 2)) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (st + R16_NST - 1 < nfull) {
            const int s2 = slot == 0 ? R16_NST - 1 : slot - 1;
            issue(r0 + (st + R16_NST - 1) * R16_KB, s2, R16_KB);
        }
        if (!(dbg & 1)) products((unsigned)(slot * STAGE));
        slot = slot == R16_NST - 1 ? 0 : slot + 1;
    }
    if (tail > 0) {
        // the partial last stage: rows past the chunk's end are zeros in LDS (plain stores: nothing is in flight here)
        __syncthreads();
        for (int i = tid; i < (R16_KB - tail) * (PZ + PH) / 16; i += R16_NT) {
            const int per_z = (R16_KB - tail) * PZ / 16;
            char* d = i < per_z ? lds + tail * PZ + i * 16 : lds + IMGH + tail * PH + (i - per_z) * 16;
            *reinterpret_cast<uint4*>(d) = make_uint4(0u, 0u, 0u, 0u);
        }
        issue(r0 + nfull * R16_KB, 0, tail);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        products(0u);
    }
    // ---- the partial output into this chunk's slab (odd chunks negated as a whole: the slab sum subtracts them) ----
    const int kpad = tn_kpad(J.k_in);
    float* slab = J.slab + (size_t)by * J.n_out * kpad;
    const int li = lane & 31;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int c = c0 + 32 * (wc * TN + j) + li;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int o = o0 + 32 * (wo * TM + i) + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (o < J.n_out && c < J.k_in && c < c0 + BC) slab[(size_t)o * kpad + c] = acc[i][j][r];
            }
        }
    if (bias_thread) {
#pragma unroll
        for (int e = 0; e < 4; ++e) slab[(size_t)(o0 + tid * 4 + e) * kpad + J.k_in] = odd ? -bsum[e] : bsum[e];
    }
}

}  // namespace

__global__ __launch_bounds__(R16_NT, 2) void wgrad_rows16_kernel(WpTable tab) {
    extern __shared__ __attribute__((aligned(16))) char r16_lds[];
    const int b = blockIdx.x;
    int j = -1;
    for (int i = 0; i < tab.njobs; ++i)
        if (tab.job[i].variant >= 16 && b >= tab.job[i].block0 &&
            b < tab.job[i].block0 + tab.job[i].tiles_o * tab.job[i].tiles_c * tab.job[i].nsplit * tab.job[i].nbatch) j = i;
    if (j < 0) return;
    const WpJob& J = tab.job[j];
    const int local = b - J.block0;
    const int ntiles = J.tiles_o * J.tiles_c;
    const int by = local / ntiles, tile = local - by * ntiles;
    const int tile_o = tile / J.tiles_c, tile_c = tile - tile_o * J.tiles_c;
    switch (J.variant) {
        case 16: r16_block<2, 4, 10, 1>(J, tile_o, tile_c, by, r16_lds, tab.debug); break;   // 640 x 128
        case 17: r16_block<4, 2, 1, 10>(J, tile_o, tile_c, by, r16_lds, tab.debug); break;   // 128 x 640
        case 18: r16_block<2, 4, 7, 1>(J, tile_o, tile_c, by, r16_lds, tab.debug); break;    // 448 x 128
        default: r16_block<2, 4, 4, 2>(J, tile_o, tile_c, by, r16_lds, tab.debug); break;    // tiles of 256 x 256
    }
}

// the variant (16 .. 19) of this kernel for an [n_out x k_in] product over bf16 rows, or -1: the one-pass shapes of the 256-d model,
// and -- variant 19 in tiles of 256 x 256 -- what needs several output tiles anyway: the node-level products ([2176 x 256] per-node
// projections, [1024 x 2048] encoder layers: GEMM-shaped, few rows and a wide output; the row-panel kernel's 128 x 128 tiles re-read
// the fp32 operand rows 17 / 2 times there).  Every 1 KiB piece of a row must have a live lane (the counted vmcnt waits assume every DMA
// instruction was issued): 256-column tiles are one piece per row, live for any non-empty tile.
// the invariant behind the kernel's counted `s_waitcnt vmcnt(NPW * (NST - 2))`: every wave issues EVERY DMA instruction of a stage, i.e.
// every 512-column (1 KiB) piece of the variant's [bo | bc] row image has at least one live 8-column chunk in EVERY tile of the job -- a
// dead piece is skipped by the compiler's execz branch and the count under-waits.  Checked here, where the variant is
// chosen: a shape that would break it goes to the row-panel kernel instead.
static bool r16_pieces_live(int bo, int bc, int n_out, int k_in, int to, int tc) {
    auto live = [](int width, int tiles, int b) {
        const int last = width - (tiles - 1) * b;                 // live columns of the last (narrowest) tile
        const int pieces = (b + 511) / 512;
        return last >= 8 && last > 512 * (pieces - 1);            // ... reach into the tile's last piece
    };
    return live(n_out, to, bo) && live(k_in, tc, bc);
}

int r16_variant(int n_out, int k_in, int* tiles_o, int* tiles_c) {
    if (getenv("MPNHIP_NO_WGRAD_ROWS16") || n_out % 8 != 0 || k_in % 8 != 0) return -1;
    const int bo[4] = {640, 128, 448, 256}, bc[4] = {128, 640, 128, 256};
    int v = -1, to = 1, tc = 1;
    static const bool tiled = !getenv("MPNHIP_NO_WGRAD_ROWS16_TILED");
    if (n_out > 512 && n_out <= 640 && k_in <= 128) v = 16;
    else if (n_out <= 128 && k_in > 512 && k_in <= 640) v = 17;
    else if (n_out > 256 && n_out <= 448 && k_in <= 128) v = 18;
    else if (n_out > 128 && n_out <= 256 && k_in > 256 && k_in <= 512) { tc = 2; v = 19; }
    else if (tiled && ((n_out > 640 && k_in > 128) || (n_out > 128 && k_in > 640))) {
        to = (n_out + 255) / 256;
        tc = (k_in + 255) / 256;
        v = 19;
    }
    if (v < 0 || !r16_pieces_live(bo[v - 16], bc[v - 16], n_out, k_in, to, tc)) return -1;
    *tiles_o = to;
    *tiles_c = tc;
    return v;
}

size_t r16_lds_bytes() {
    size_t m = 0;
    const int bo[4] = {640, 128, 448, 256}, bc[4] = {128, 640, 128, 256};
    for (int i = 0; i < 4; ++i) {
        const size_t s = (size_t)R16_NST * R16_KB * (r16_pitch(bo[i]) + r16_pitch(bc[i]));
        m = s > m ? s : m;
    }
    return m > 84 * 1024 ? m : 84 * 1024;   // (above 80 KB: one block per CU -- the accumulators need 2 waves per SIMD at most)
}

int launch_wgrad_rows16(const WpTable& tab, int nblocks, hipStream_t s) {
    static const bool attr_set = [] {
        return hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_rows16_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)r16_lds_bytes()) == hipSuccess;
    }();
    (void)attr_set;
    hipEvent_t e0, e1;
    if (prof_launch_events(&e0, &e1))
        hipExtLaunchKernelGGL(wgrad_rows16_kernel, dim3((unsigned)nblocks), dim3(R16_NT), r16_lds_bytes(), s, e0, e1, 0, tab);
    else
        hipLaunchKernelGGL(wgrad_rows16_kernel, dim3((unsigned)nblocks), dim3(R16_NT), r16_lds_bytes(), s, tab);
    MPN_LAUNCH_CHECK();
    return MPNHIP_OK;
}

}  // namespace mpnhip
