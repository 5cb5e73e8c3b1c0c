// Weight-gradient products of the backward pass:
//   dW[o, c] += sum_m dZ[m, o] * H[m, c]        db[o] += sum_m dZ[m, o]
// i.e. a GEMM whose reduction runs over the EDGES (or nodes): tens of thousands of rows against a
// small [n_out, k_in] output.  Parallelism comes from splitting the row range: grid.y enumerates row
// chunks, every block reduces its chunk with fp32 MFMAs (v_mfma_f32_32x32x2_f32) into a private slab
// [n_out][tn_kpad(k_in)] (column k_in carries the bias partial), and a second kernel sums the slabs in a
// fixed order into the caller's gradient buffers -- deterministic, no float atomics.
//
// Both operands are read exactly as stored (row-major, rows = reduction index), 16 bytes per lane,
// straight into k-major LDS images (the row index IS the MFMA k index, so nothing is transposed).
// H may come as two column segments (the reference's torch.cat([initial, current]) input), rows of
// either operand may be gathered through an index (edge_attr / dlogits live in original edge order).
#include <cstdlib>

#include "common.h"

namespace mpnhip {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int TBK = 32;
constexpr int TBM = 64;    // n_out tile
constexpr int TNT = 256;   // k_in tile: template parameter TBN (128, or 64 for k_in <= 64)

__device__ __forceinline__ float4 ld4t(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 keep4t(bool ok, float4 v) {
    v.x = ok ? v.x : 0.f; v.y = ok ? v.y : 0.f; v.z = ok ? v.z : 0.f; v.w = ok ? v.w : 0.f;
    return v;
}
// field-wise select of the group (indexing the by-value kernel argument with blockIdx.z would force
// the whole struct into scratch memory)
#define TN_G(field) (blockIdx.z == 0 ? args.g[0].field : args.g[1].field)
#define TN_GG(grp, field) ((grp) == 0 ? args.g[0].field : args.g[1].field)

template <int TBN>
__global__ __launch_bounds__(TNT) void gemm_tn_kernel(TnArgs args) {
    constexpr int PA = TBM + 4, PB = TBN + 4;
    constexpr int NB = TBN / 32;          // float4 loads of the B' tile per thread; columns per row = TBN / 4
    constexpr int TW = TBN / 64;          // 32-wide accumulator tiles per wave (waves 2 x 2)
    __shared__ __attribute__((aligned(16))) float As[TBK * PA];
    __shared__ __attribute__((aligned(16))) float Bs[TBK * PB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 31, lh = lane >> 5;
    const int n_out = args.n_out, k_in = args.k_in, csplit = args.csplit;
    const int ntile_c = (k_in + TBN - 1) / TBN;
    // XCD-aware block -> (tile, row chunk) mapping.  Workgroups are dealt round-robin over the 8 XCDs (block b runs on XCD
    // b % 8, each with its own 4 MB L2): the output tiles of ONE row chunk all read the same dZ / H rows, so they are given
    // to consecutive slots of the SAME XCD -- they start together, and the rows one of them pulls into that L2 serve the
    // others (before: the tiles of a chunk were consecutive block ids = 5 different XCDs, every re-read went to the
    // Infinity Cache / HBM and the kernel ran at that bandwidth, not at the MFMA rate).  Placement is for speed only.
    int bid = blockIdx.x;
    const int per_group = args.tiles * args.ny8;
    const int grp = bid / per_group;
    bid -= grp * per_group;
    const int slot = args.xcd_map ? bid >> 3 : bid;
    const int tile = slot % args.tiles;
    const int by = args.xcd_map ? (slot / args.tiles) * 8 + (bid & 7) : slot / args.tiles;
    if (by >= args.nsplit * args.nbatch) return;
    const int o0 = (tile / ntile_c) * TBM;
    const int c0 = (tile % ntile_c) * TBN;

    const int* rbp = TN_GG(grp, row_begin);
    const int* rep = TN_GG(grp, row_end);
    const int rb = rbp ? *rbp : 0;
    const int re = rep ? *rep : (int)TN_GG(grp, m_static);
    const int batch = by / args.nsplit;
    const int r0 = rb + (by % args.nsplit) * args.chunk;
    int r1 = r0 + args.chunk;
    r1 = r1 < re ? r1 : re;
    if (r0 >= r1) return;  // empty chunk: the reduce kernel skips it too

    // loader geometry: A' tile [32 rows][64 cols] = 512 float4 (2 / thread); B' tile [32][TBN] (NB / thread)
    const int ar = tid >> 4, ac = (tid & 15) * 4;        // + 16 rows for the second
    const int br = tid / (TBN / 4), bc = (tid % (TBN / 4)) * 4;   // + 1024 / TBN rows per j
    int oc = o0 + ac;
    oc = oc + 3 < n_out ? oc : n_out - 4;                // clamped columns are never stored
    int cc = c0 + bc;
    cc = cc + 3 < k_in ? cc : k_in - 4;
    const bool bseg2 = cc >= csplit;                     // csplit % 4 == 0: a float4 lies in one segment
    const float* hbase = (bseg2 ? TN_GG(grp, H2) : TN_GG(grp, H)) + (bseg2 ? cc - csplit : cc) +
                         (int64_t)batch * (bseg2 ? TN_GG(grp, h2_bstride) : TN_GG(grp, h_bstride));
    const int64_t ldh = bseg2 ? TN_GG(grp, ldh2) : TN_GG(grp, ldh);
    const float* zbase = TN_GG(grp, dZ) + oc + (int64_t)batch * TN_GG(grp, z_bstride);
    const int64_t ldz = TN_GG(grp, ldz);

    f32x16 acc[TW];
#pragma unroll
    for (int j = 0; j < TW; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    // bias gradient = column sums of dZ: accumulated from the loader's own registers (every thread owns four columns of two rows
    // of each stage: 8 adds), combined across the 16 row groups once at the end -- no LDS column walk in the stage loop, and
    // no wave that is slower than the others at the barrier
    float4 bacc = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 a_reg[2], b_reg[NB];

    // raw loads only (rows clamped); the zero-select for rows past the chunk end happens at LDS-store time so
    // that nothing consumes the loads before the MFMA phase of the previous step has been issued
    auto load = [&](int m0) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int m = m0 + ar + 16 * j;
            a_reg[j] = ld4t(zbase + (int64_t)(m < r1 ? m : r1 - 1) * ldz);
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int m = m0 + br + (1024 / TBN) * j;
            b_reg[j] = ld4t(hbase + (int64_t)(m < r1 ? m : r1 - 1) * ldh);
        }
    };
    auto store = [&](int m0) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const float4 v = keep4t(m0 + ar + 16 * j < r1, a_reg[j]);
            *reinterpret_cast<float4*>(&As[(ar + 16 * j) * PA + ac]) = v;
            bacc.x += v.x; bacc.y += v.y; bacc.z += v.z; bacc.w += v.w;
        }
#pragma unroll
        for (int j = 0; j < NB; ++j)
            *reinterpret_cast<float4*>(&Bs[(br + (1024 / TBN) * j) * PB + bc]) = keep4t(m0 + br + (1024 / TBN) * j < r1, b_reg[j]);
    };

    load(r0);
    const float* a_ptr = As + lh * PA + wm * 32 + li;
    const float* b_ptr = Bs + lh * PB + wn * 32 * TW + li;
    for (int m0 = r0; m0 < r1; m0 += TBK) {
        store(m0);
        __syncthreads();
        if (m0 + TBK < r1) load(m0 + TBK);
        // operand fetch one k pair ahead of its MFMAs, pinned (see gemm.hip)
        float a_cur = a_ptr[0], b_cur[TW];
#pragma unroll
        for (int j = 0; j < TW; ++j) b_cur[j] = b_ptr[32 * j];
#pragma unroll
        for (int kk = 0; kk < TBK; kk += 2) {
            float a_nxt = 0.f, b_nxt[TW];
            if (kk + 2 < TBK) {
                a_nxt = a_ptr[(kk + 2) * PA];
#pragma unroll
                for (int j = 0; j < TW; ++j) b_nxt[j] = b_ptr[(kk + 2) * PB + 32 * j];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < TW; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur, b_cur[j], acc[j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (kk + 2 < TBK) {
                a_cur = a_nxt;
#pragma unroll
                for (int j = 0; j < TW; ++j) b_cur[j] = b_nxt[j];
            }
        }
        __syncthreads();
    }
    // ---- write the partial tile into this chunk's slab --------------------------------------------
    const int kpad = tn_kpad(k_in);
    float* slab = TN_GG(grp, slab) + (size_t)by * n_out * kpad;
#pragma unroll
    for (int tj = 0; tj < TW; ++tj) {
        const int c = c0 + wn * 32 * TW + tj * 32 + li;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int o = o0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (o < n_out && c < k_in) slab[(size_t)o * kpad + c] = acc[tj][r];
        }
    }
    if (c0 == 0) {
        // the 16 row groups' partial column sums meet in LDS (As is free after the loop's last barrier), fixed order
        *reinterpret_cast<float4*>(&As[ar * PA + ac]) = bacc;
        __syncthreads();
        if (tid < TBM && o0 + tid < n_out) {
            float bsum = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) bsum += As[r * PA + tid];
            slab[(size_t)(o0 + tid) * kpad + k_in] = bsum;
        }
    }
}

// Any shape / alignment / row gathers (the K = 6 edge-encoder input read through the sort permutation, the
// [1 x hc] classifier output layer whose dZ is grad_logits in original edge order): one thread per output
// element (o, c) -- c == k_in is the bias column -- looping over the rows of its chunk; a block covers 256
// outputs of one chunk, so the chunk's dZ / H rows are shared through L1.
__global__ __launch_bounds__(256) void gemm_tn_generic_kernel(TnArgs args) {
    // 8 lanes share one output element (rows strided by 8, four independent partial sums each so that
    // several gathers are in flight), 32 output elements per block
    const int64_t nout_total = (int64_t)args.n_out * (args.k_in + 1);
    const int l = threadIdx.x & 7;
    const int64_t t = (int64_t)blockIdx.x * 32 + (threadIdx.x >> 3);
    const int* rbp = TN_G(row_begin);
    const int* rep = TN_G(row_end);
    const int rb = rbp ? *rbp : 0;
    const int re = rep ? *rep : (int)TN_G(m_static);
    const int batch = blockIdx.y / args.nsplit;
    const int r0 = rb + (blockIdx.y % args.nsplit) * args.chunk;
    int r1 = r0 + args.chunk;
    r1 = r1 < re ? r1 : re;
    if (r0 >= r1) return;
    const bool live = t < nout_total;
    const int64_t tt = live ? t : 0;
    const int o = (int)(tt / (args.k_in + 1)), c = (int)(tt % (args.k_in + 1));
    const float* dZ = TN_G(dZ) + (int64_t)batch * TN_G(z_bstride) + o;
    const bool seg2 = c >= args.csplit && c < args.k_in;
    const float* H = seg2 ? TN_G(H2) + (int64_t)batch * TN_G(h2_bstride) + (c - args.csplit)
                          : TN_G(H) + (int64_t)batch * TN_G(h_bstride) + (c < args.k_in ? c : 0);
    const int64_t ldh = seg2 ? TN_G(ldh2) : TN_G(ldh);
    const int* zi = TN_G(dz_idx);
    const int* hi = TN_G(h_idx);
    const int64_t ldz = TN_G(ldz);
    const bool bias_col = c >= args.k_in;
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    for (int m0 = r0 + l; m0 < r1; m0 += 32) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int m = m0 + 8 * u;
            const int mc = m < r1 ? m : r1 - 1;
            const int64_t rz = zi ? zi[mc] : mc;
            const int64_t rh = hi ? hi[mc] : mc;
            const float z = dZ[rz * ldz];
            const float h = bias_col ? 1.f : H[rh * ldh];
            s[u] = fmaf(m < r1 ? z : 0.f, h, s[u]);
        }
    }
    float tot = (s[0] + s[1]) + (s[2] + s[3]);
    tot += __shfl_xor(tot, 1, 64);
    tot += __shfl_xor(tot, 2, 64);
    tot += __shfl_xor(tot, 4, 64);
    if (live && l == 0) TN_G(slab)[((size_t)blockIdx.y * args.n_out + o) * tn_kpad(args.k_in) + c] = tot;
}

// Few input columns, many outputs (k_in <= 8, 32 < n_out <= 256: the edge encoder's first layer at the wider models' widths -- [144 x 6]
// at 256-d -- whose H rows are edge_attr read through the sort permutation): the any-shape kernel above gives every output element
// its own 8 lanes, so 32 blocks per row chunk each walk all of the chunk's dZ rows with 4-byte strided loads (0.83 ms of a cfg-E
// training step).  Here a block owns a row chunk and ALL outputs: thread o keeps the k_in + 1 sums of output row o, reads dZ[row][o]
// (consecutive threads: whole lines) and takes the row's k_in H values as LDS broadcasts from a 64-row stage.  Same chunking and
// slab format; plain fp32 FMAs in a fixed order.
constexpr int NK_ROWS = 64;
__global__ __launch_bounds__(256) void gemm_tn_narrowk_kernel(TnArgs args) {
    __shared__ __attribute__((aligned(16))) float hs[NK_ROWS][8];
    __shared__ int zr[NK_ROWS];
    const int n_out = args.n_out, k_in = args.k_in;
    const int* rbp = TN_G(row_begin);
    const int* rep = TN_G(row_end);
    const int rb = rbp ? *rbp : 0;
    const int re = rep ? *rep : (int)TN_G(m_static);
    const int batch = blockIdx.y / args.nsplit;
    const int r0 = rb + (blockIdx.y % args.nsplit) * args.chunk;
    int r1 = r0 + args.chunk;
    r1 = r1 < re ? r1 : re;
    if (r0 >= r1) return;
    const int tid = threadIdx.x;
    const bool live = tid < n_out;
    const float* dZ = TN_G(dZ) + (int64_t)batch * TN_G(z_bstride) + (live ? tid : 0);
    const float* H = TN_G(H) + (int64_t)batch * TN_G(h_bstride);
    const int* zi = TN_G(dz_idx);
    const int* hi = TN_G(h_idx);
    const int64_t ldz = TN_G(ldz), ldh = TN_G(ldh);
    float acc[9];
#pragma unroll
    for (int c = 0; c < 9; ++c) acc[c] = 0.f;
    for (int m0 = r0; m0 < r1; m0 += NK_ROWS) {
        const int nr = r1 - m0 < NK_ROWS ? r1 - m0 : NK_ROWS;
        for (int i = tid; i < NK_ROWS * 8; i += 256) {
            const int r = i >> 3, c = i & 7;
            float v = 0.f;
            if (r < nr && c < k_in) v = H[(int64_t)(hi ? hi[m0 + r] : m0 + r) * ldh + c];
            hs[r][c] = v;
        }
        if (tid < NK_ROWS) zr[tid] = tid < nr ? (zi ? zi[m0 + tid] : m0 + tid) : (zi ? zi[m0] : m0);
        __syncthreads();
        if (live) {
#pragma unroll 8
            for (int r = 0; r < NK_ROWS; ++r) {
                const float z = r < nr ? dZ[(int64_t)zr[r] * ldz] : 0.f;
                const float4 h0 = *reinterpret_cast<const float4*>(&hs[r][0]);
                const float4 h1 = *reinterpret_cast<const float4*>(&hs[r][4]);
                acc[0] = fmaf(z, h0.x, acc[0]); acc[1] = fmaf(z, h0.y, acc[1]); acc[2] = fmaf(z, h0.z, acc[2]); acc[3] = fmaf(z, h0.w, acc[3]);
                acc[4] = fmaf(z, h1.x, acc[4]); acc[5] = fmaf(z, h1.y, acc[5]); acc[6] = fmaf(z, h1.z, acc[6]); acc[7] = fmaf(z, h1.w, acc[7]);
                acc[8] += z;
            }
        }
        __syncthreads();
    }
    if (live) {
        float* sl = TN_G(slab) + ((size_t)blockIdx.y * n_out + tid) * tn_kpad(k_in);
#pragma unroll
        for (int c = 0; c < 8; ++c)
            if (c < k_in) sl[c] = acc[c];
        sl[k_in] = acc[8];
    }
}
static inline bool tn_narrowk(int n_out, int k_in) { return k_in <= 8 && n_out > 32 && n_out <= 256 && !getenv("MPNHIP_TN_NO_NARROWK"); }

// Narrow outputs (n_out <= 32, k_in <= 32: the reference's 18-wide edge encoder, the [1 x hc] classifier output layer): the
// any-shape kernel above is latency-bound there (one dependent gather chain per row and lane).  Here a block stages 64 rows of
// dZ and H (through the row indices, if any) in LDS and every thread owns up to five output elements (o, c) -- c == k_in is
// the bias column, fed by a column of ones -- reading dZ as a broadcast and H conflict-free.  Same chunking and slab format.
constexpr int TS_ROWS = 64;
__global__ __launch_bounds__(256) void gemm_tn_small_kernel(TnArgs args) {
    __shared__ float zs[TS_ROWS][33];
    __shared__ float hs[TS_ROWS][34];
    const int n_out = args.n_out, k_in = args.k_in, kc = k_in + 1;
    const int nout_total = n_out * kc;
    const int* rbp = TN_G(row_begin);
    const int* rep = TN_G(row_end);
    const int rb = rbp ? *rbp : 0;
    const int re = rep ? *rep : (int)TN_G(m_static);
    const int batch = blockIdx.y / args.nsplit;
    const int r0 = rb + (blockIdx.y % args.nsplit) * args.chunk;
    int r1 = r0 + args.chunk;
    r1 = r1 < re ? r1 : re;
    if (r0 >= r1) return;
    const float* dZ = TN_G(dZ) + (int64_t)batch * TN_G(z_bstride);
    const float* H = TN_G(H) + (int64_t)batch * TN_G(h_bstride);
    const int* zi = TN_G(dz_idx);
    const int* hi = TN_G(h_idx);
    const int64_t ldz = TN_G(ldz), ldh = TN_G(ldh);
    float acc[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    int oo[5], cc[5];
#pragma unroll
    for (int q = 0; q < 5; ++q) {
        const int t = threadIdx.x + 256 * q;
        oo[q] = t < nout_total ? t / kc : 0;
        cc[q] = t < nout_total ? t % kc : 0;
    }
    for (int m0 = r0; m0 < r1; m0 += TS_ROWS) {
        const int nr = r1 - m0 < TS_ROWS ? r1 - m0 : TS_ROWS;
        for (int i = threadIdx.x; i < TS_ROWS * n_out; i += 256) {
            const int r = i / n_out, o = i - r * n_out;
            const int64_t row = r < nr ? (zi ? zi[m0 + r] : m0 + r) : 0;
            zs[r][o] = r < nr ? dZ[row * ldz + o] : 0.f;
        }
        for (int i = threadIdx.x; i < TS_ROWS * kc; i += 256) {
            const int r = i / kc, c = i - r * kc;
            const int64_t row = r < nr ? (hi ? hi[m0 + r] : m0 + r) : 0;
            hs[r][c] = c == k_in ? 1.f : (r < nr ? H[row * ldh + c] : 0.f);
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 5; ++q) {
            if (threadIdx.x + 256 * q < nout_total) {
                float s = acc[q];
#pragma unroll 8
                for (int r = 0; r < TS_ROWS; ++r) s = fmaf(zs[r][oo[q]], hs[r][cc[q]], s);
                acc[q] = s;
            }
        }
        __syncthreads();
    }
    float* slab = TN_G(slab) + (size_t)blockIdx.y * n_out * tn_kpad(k_in);
#pragma unroll
    for (int q = 0; q < 5; ++q)
        if (threadIdx.x + 256 * q < nout_total) slab[(size_t)oo[q] * tn_kpad(k_in) + cc[q]] = acc[q];
}

// grad_w[o * ldw + c] += sum_s slab[s][o][c];  grad_b[o] += sum_s slab[s][o][k_in]  over the non-empty chunks.
// Block = 32 x 16-byte columns of the padded slab image x 8 slab groups: a wave reads 512 contiguous bytes of one
// slab, four slabs in flight per lane; the eight groups' partial sums meet in LDS in a fixed order (deterministic).
// The slab pitch (tn_kpad: k_in + 1 columns padded to whole 16-byte columns) is a multiple of 4 floats for every k_in; pad
// columns hold whatever the buffer held and are summed but never stored.  The scalar kernel below serves unaligned slab bases.
__global__ __launch_bounds__(256) void slab_reduce_kernel(TnArgs args) {
    __shared__ float4 part[8][32];
    const int kpad = tn_kpad(args.k_in);
    const int64_t total4 = (int64_t)args.n_out * kpad / 4;
    const int64_t q = (int64_t)blockIdx.x * 32 + (threadIdx.x & 31);
    const int grp = threadIdx.x >> 5;
    const int* rbp = TN_G(row_begin);
    const int* rep = TN_G(row_end);
    const int rb = rbp ? *rbp : 0;
    const int re = rep ? *rep : (int)TN_G(m_static);
    int nvalid = (re - rb + args.chunk - 1) / args.chunk;
    nvalid = nvalid < 0 ? 0 : (nvalid > args.nsplit ? args.nsplit : nvalid);
    const int nslab_all = nvalid * args.nbatch;
    // two-stage form for outputs too small to fill the chip with one block per 32 columns: stage 1 (grid.y = ny) sums
    // the slabs j = y (mod ny) into slab y IN PLACE (the thread that writes an element is the one that read it), stage
    // 2 sums those ny partial slabs into the gradient.  Single stage: ny = 1, stage = 2.
    const int ny = args.red_ny > 1 ? args.red_ny : 1;
    const int first = args.red_stage == 1 ? (int)blockIdx.y : 0;
    const int step = args.red_stage == 1 ? ny : 1;
    const int nslab = args.red_stage == 1 ? nslab_all : (nslab_all < ny ? nslab_all : (ny > 1 ? ny : nslab_all));
    auto slab_at = [&](int j) { const int b = j / nvalid, i = j - b * nvalid; return (size_t)b * args.nsplit + i; };
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const size_t stride = (size_t)args.n_out * kpad;
    if (q < total4) {
        const float* p = TN_G(slab) + q * 4;
        for (int j0 = first + step * grp; j0 < nslab; j0 += 32 * step) {
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                int j = j0 + 8 * step * u;
                j = j < nslab ? j : j0;  // clamped, unconditional loads
                v[u] = *reinterpret_cast<const float4*>(p + slab_at(j) * stride);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (j0 + 8 * step * u < nslab) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
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
        if (args.red_stage == 1) {
            if (first < nslab) *reinterpret_cast<float4*>(TN_G(slab) + q * 4 + slab_at(first) * stride) = s;
            return;
        }
        const float sv[4] = {s.x, s.y, s.z, s.w};
        float* gw = TN_G(grad_w);
        float* gb = TN_G(grad_b);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int64_t t = q * 4 + e;
            const int o = (int)(t / kpad), c = (int)(t - (int64_t)o * kpad);
            if (c < args.k_in) { if (gw) gw[(int64_t)o * TN_G(ldw) + c] += sv[e]; }
            else if (c == args.k_in) { if (gb) gb[o] += sv[e]; }
        }
    }
}

// scalar variant (k_in % 4 != 0): 8 lanes share one output element
__global__ __launch_bounds__(256) void slab_reduce_scalar_kernel(TnArgs args) {
    const int kpad = tn_kpad(args.k_in);
    const int64_t total = (int64_t)args.n_out * (args.k_in + 1);
    const int64_t t = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 3;
    const int l = threadIdx.x & 7;
    const int* rbp = TN_G(row_begin);
    const int* rep = TN_G(row_end);
    const int rb = rbp ? *rbp : 0;
    const int re = rep ? *rep : (int)TN_G(m_static);
    int nvalid = (re - rb + args.chunk - 1) / args.chunk;
    nvalid = nvalid < 0 ? 0 : (nvalid > args.nsplit ? args.nsplit : nvalid);
    float s = 0.f;
    int o = 0, c = 0;
    if (t < total) {
        o = (int)(t / (args.k_in + 1));
        c = (int)(t % (args.k_in + 1));
        const float* p = TN_G(slab) + (size_t)o * kpad + c;
        const size_t stride = (size_t)args.n_out * kpad;
        for (int b = 0; b < args.nbatch; ++b)
            for (int i = l; i < nvalid; i += 8) s += p[((size_t)b * args.nsplit + i) * stride];
    }
    s += __shfl_xor(s, 1, 64);
    s += __shfl_xor(s, 2, 64);
    s += __shfl_xor(s, 4, 64);
    if (t < total && l == 0) {
        if (c < args.k_in) {
            float* gw = TN_G(grad_w);
            if (gw) gw[(int64_t)o * TN_G(ldw) + c] += s;
        } else {
            float* gb = TN_G(grad_b);
            if (gb) gb[o] += s;
        }
    }
}

static bool al16t(const void* p) { return (((uintptr_t)p) & 15) == 0; }

size_t tn_slab_floats(int n_out, int k_in, int64_t m_upper, int nbatch) {
    TnArgs a = {};
    a.n_out = n_out;
    a.k_in = k_in;
    a.m_upper = m_upper;
    a.nbatch = nbatch;
    tn_plan(a);
    return (size_t)a.nbatch * a.nsplit * n_out * tn_kpad(k_in);
}

void tn_plan(TnArgs& a) {
    // enough row chunks that tiles x batches x chunks fills the chip a few times over, but never chunks so
    // short that the slab traffic (chunks n_out k_in) rivals the operand traffic (rows (n_out + k_in))
    if (a.nbatch < 1) a.nbatch = 1;
    const int tbn = a.k_in <= 64 ? 64 : 128;
    int tiles = ((a.n_out + TBM - 1) / TBM) * ((a.k_in + tbn - 1) / tbn) * a.nbatch;
    // measured on MI355X (cfg-B / C / D training steps): 1024 blocks for the long products (>= 100k rows over the batch: the
    // edge-level ones), 1536 for the short ones (node level, small graphs), which need the extra row chunks to fill the chip
    static const int blocks_env = [] { const char* e = getenv("MPNHIP_TN_BLOCKS"); const int v = e ? atoi(e) : 0; return v >= 64 ? v : 0; }();  // tuning override
    const int blocks_wanted = blocks_env ? blocks_env : (a.m_upper * a.nbatch >= 100000 ? 1024 : 1536);
    int target = blocks_wanted / (tiles > 0 ? tiles : 1);
    if (target < 1) target = 1;
    // narrow outputs (the reference's 18-wide edge encoder, the classifier's output layer): one block per 64-row staging round.
    // A block's rounds are serial and each is two dependent global-load latencies long (row index -> row), so 128 blocks x 10
    // rounds took 45-52 us per product at cfg-C's 77.8k edges; the slabs stay small (n_out (k_in + 4) floats each)
    const bool narrow = a.n_out <= 32 && a.k_in <= 32;
    // (few input columns, one block per chunk for all outputs -- gemm_tn_narrowk_kernel: two blocks per CU, slabs of n_out x 8 floats)
    const int cap = narrow ? 2048 : (tn_narrowk(a.n_out, a.k_in) ? 512 : 128);
    if (target > cap) target = cap;
    int64_t chunk = (a.m_upper + target - 1) / target;
    chunk = (chunk + TBK - 1) / TBK * TBK;
    const int min_chunk = narrow ? 64 : 128;
    if (chunk < min_chunk) chunk = min_chunk;
    a.chunk = (int)chunk;
    a.nsplit = (int)((a.m_upper + chunk - 1) / chunk);
    if (a.nsplit < 1) a.nsplit = 1;
}

int launch_gemm_tn(const TnArgs& a_in, hipStream_t s) {
    TnArgs a = a_in;
    MPN_CHECK_ARG(a.ngroups == 1 || a.ngroups == 2, "gemm_tn: ngroups");
    MPN_CHECK_ARG(a.n_out >= 1 && a.k_in >= 1 && a.csplit >= 0 && a.csplit <= a.k_in, "gemm_tn: dims");
    if (a.m_upper <= 0) return MPNHIP_OK;
    if (a.nbatch < 1) a.nbatch = 1;
    bool fast = (a.n_out % 4 == 0) && (a.k_in % 4 == 0) && (a.csplit % 4 == 0);
    for (int i = 0; i < a.ngroups; ++i) {
        const TnGroup& g = a.g[i];
        MPN_CHECK_ARG(g.dZ && g.H && (a.csplit == a.k_in || g.H2) && g.slab, "gemm_tn: null operand");
        fast = fast && !g.dz_idx && !g.h_idx && al16t(g.dZ) && g.ldz % 4 == 0 && al16t(g.H) && g.ldh % 4 == 0 &&
               (!g.H2 || (al16t(g.H2) && g.ldh2 % 4 == 0)) && g.z_bstride % 4 == 0 && g.h_bstride % 4 == 0 &&
               g.h2_bstride % 4 == 0;
    }
    tn_plan(a);
    count_path(fast ? PC_TN_MFMA : (a.n_out <= 32 && a.k_in <= 32 && a.csplit == a.k_in && !getenv("MPNHIP_TN_NO_SMALL") ? PC_TN_SMALL : PC_TN_GENERIC));
    if (fast) {
        const int tbn = a.k_in <= 64 ? 64 : 128;
        a.tiles = ((a.n_out + TBM - 1) / TBM) * ((a.k_in + tbn - 1) / tbn);
        a.ny8 = (a.nsplit * a.nbatch + 7) / 8 * 8;   // (row chunks x batches, padded to the 8 XCDs: see the kernel's block mapping)
        // measured on MI355X (tools/wgrad_bench.py, cfg-B shapes, 5 steps per product): the XCD-aware mapping pays where a handful
        // of output tiles share every H row (edge layer 1, 5 tiles: 282 -> 259 us) and costs a little elsewhere (3 tiles: 156 ->
        // 176 us; 128 tiles: 140 -> 173 us -- there the chunks' tiles no longer start together)
        a.xcd_map = a.tiles >= 4 && a.tiles <= 8 ? 1 : 0;
        if (const char* e = getenv("MPNHIP_TN_XCD")) a.xcd_map = e[0] == '1';   // A-B switch for measurements
        const unsigned nblocks = (unsigned)(a.tiles * a.ny8 * a.ngroups);
        prof_begin(PROF_TN, s, a.flops);
        if (tbn == 64)
            MPN_LAUNCH_PROFILED(gemm_tn_kernel<64>, dim3(nblocks), dim3(TNT), s, a);
        else
            MPN_LAUNCH_PROFILED(gemm_tn_kernel<128>, dim3(nblocks), dim3(TNT), s, a);
        prof_end(PROF_TN, s);
    } else if (tn_narrowk(a.n_out, a.k_in) && a.csplit == a.k_in) {
        hipLaunchKernelGGL(gemm_tn_narrowk_kernel, dim3(1, a.nsplit * a.nbatch, a.ngroups), dim3(256), 0, s, a);
    } else if (a.n_out <= 32 && a.k_in <= 32 && a.csplit == a.k_in && !getenv("MPNHIP_TN_NO_SMALL")) {
        hipLaunchKernelGGL(gemm_tn_small_kernel, dim3(1, a.nsplit * a.nbatch, a.ngroups), dim3(256), 0, s, a);
    } else {
        const int64_t nout_total = (int64_t)a.n_out * (a.k_in + 1);
        hipLaunchKernelGGL(gemm_tn_generic_kernel, dim3((unsigned)((nout_total + 31) / 32), a.nsplit * a.nbatch, a.ngroups),
                           dim3(256), 0, s, a);
    }
    MPN_LAUNCH_CHECK();
    bool slab16 = true;   // (the slab pitch tn_kpad is a multiple of 4 floats for every k_in)
    for (int i = 0; i < a.ngroups; ++i) slab16 = slab16 && al16t(a.g[i].slab);
    if (slab16) {
        const int64_t total4 = (int64_t)a.n_out * tn_kpad(a.k_in) / 4;
        const unsigned bx = (unsigned)((total4 + 31) / 32);
        const int nslab_upper = a.nsplit * a.nbatch;
        a.red_ny = 1;
        a.red_stage = 2;
        if (bx * a.ngroups < 512 && nslab_upper >= 64) {
            int ny = (int)(1024 / (bx * a.ngroups));
            ny = ny > 32 ? 32 : ny;
            ny = ny > nslab_upper / 8 ? nslab_upper / 8 : ny;
            if (ny >= 2) {
                a.red_ny = ny;
                a.red_stage = 1;
                hipLaunchKernelGGL(slab_reduce_kernel, dim3(bx, ny, a.ngroups), dim3(256), 0, s, a);
                MPN_LAUNCH_CHECK();
                a.red_stage = 2;
            }
        }
        hipLaunchKernelGGL(slab_reduce_kernel, dim3(bx, 1, a.ngroups), dim3(256), 0, s, a);
    } else {
        const int64_t total = (int64_t)a.n_out * (a.k_in + 1) * 8;
        hipLaunchKernelGGL(slab_reduce_scalar_kernel, dim3((unsigned)((total + 255) / 256), 1, a.ngroups), dim3(256), 0, s, a);
    }
    MPN_LAUNCH_CHECK();
    return MPNHIP_OK;
}

}  // namespace mpnhip

using namespace mpnhip;

extern "C" size_t mpnhip_weight_grad_workspace_bytes(int n_out, int k_in, int64_t rows, int nbatch) {
    if (n_out < 1 || k_in < 1 || rows < 0) return 0;
    const size_t a = tn_slab_floats(n_out, k_in, rows, nbatch < 1 ? 1 : nbatch);
    const size_t b = rows > 0 ? wp_slab_floats(n_out, k_in, rows, nbatch < 1 ? 1 : nbatch, false, false) : 0;   // (MPNHIP_PREC_FP32_SPLIT form)
    return align_up((a > b ? a : b) * sizeof(float), 256) + 256;
}

static int weight_grad_args(const float* dZ, const float* H, int64_t rows, int n_out, int k_in, int nbatch, float* grad_w, float* grad_b,
                            void* workspace, size_t workspace_bytes, TnArgs* out) {
    MPN_CHECK_ARG(n_out >= 1 && k_in >= 1 && rows >= 0 && nbatch >= 1, "weight_grad: bad sizes");
    MPN_CHECK_ARG((dZ && H && grad_w) || rows == 0, "weight_grad: null pointer");
    const size_t need = mpnhip_weight_grad_workspace_bytes(n_out, k_in, rows, nbatch);
    if (rows > 0 && (!workspace || workspace_bytes < need)) {
        set_error("weight_grad: workspace %zu < %zu", workspace_bytes, need);
        return MPNHIP_ERR_WORKSPACE;
    }
    TnArgs a = {};
    a.ngroups = 1; a.n_out = n_out; a.k_in = k_in; a.csplit = k_in; a.m_upper = rows; a.nbatch = nbatch;
    TnGroup& g = a.g[0];
    g.dZ = dZ; g.ldz = n_out; g.z_bstride = rows * n_out;
    g.H = H; g.ldh = k_in; g.h_bstride = rows * k_in;
    g.m_static = rows;
    g.slab = reinterpret_cast<float*>((reinterpret_cast<uintptr_t>(workspace) + 255) / 256 * 256);
    g.grad_w = grad_w; g.ldw = k_in; g.grad_b = grad_b;
    a.flops = 2.0 * (double)rows * nbatch * n_out * k_in;
    *out = a;
    return MPNHIP_OK;
}

// precision MPNHIP_PREC_FP32_SPLIT: the row-panel kernel with three-piece bf16 operands (wgrad_panel.hip) where the shape allows it
static int weight_grad_run(const TnArgs& a, int precision, void* workspace, size_t workspace_bytes, hipStream_t s) {
    if ((precision == MPNHIP_PREC_FP32_SPLIT || precision == MPNHIP_PREC_BF16) && !getenv("MPNHIP_NO_WGRAD_PANEL")) {
        const TnGroup& g = a.g[0];
        WpProduct p = {g.dZ, g.ldz, g.z_bstride, g.H, g.ldh, g.h_bstride, nullptr, nullptr, a.m_upper, a.nbatch, a.n_out, a.k_in,
                       g.grad_w, g.ldw, g.grad_b, nullptr, nullptr, nullptr, 0, 0, 0, precision == MPNHIP_PREC_BF16 ? 1 : 3};
        WpBatch b;
        WpBatchGuard guard;
        wp_batch_begin(&b, g.slab, (workspace_bytes - 256) / sizeof(float), false);
        if (wp_batch_add(p)) return wp_batch_flush(s);
        wp_batch_abort();
    }
    return launch_gemm_tn(a, s);
}

extern "C" int mpnhip_weight_grad_prec(const float* dZ, const float* H, int64_t rows, int n_out, int k_in, int nbatch, int precision,
                                       float* grad_w, float* grad_b, void* workspace, size_t workspace_bytes, void* stream) {
    MPN_CHECK_ARG(precision == MPNHIP_PREC_FP32 || precision == MPNHIP_PREC_FP32_SPLIT || precision == MPNHIP_PREC_BF16, "weight_grad: precision %d", precision);
    TnArgs a;
    MPN_TRY(weight_grad_args(dZ, H, rows, n_out, k_in, nbatch, grad_w, grad_b, workspace, workspace_bytes, &a));
    if (rows == 0) return MPNHIP_OK;
    return weight_grad_run(a, precision, workspace, workspace_bytes, static_cast<hipStream_t>(stream));
}

extern "C" int mpnhip_weight_grad_bf16_rows(const uint16_t* dZ, const uint16_t* H, int64_t rows, int n_out, int k_in, int nbatch,
                                            float* grad_w, float* grad_b, void* workspace, size_t workspace_bytes, void* stream) {
    MPN_CHECK_ARG(n_out >= 1 && k_in >= 1 && rows >= 0 && nbatch >= 1, "weight_grad_bf16_rows: bad sizes");
    if (rows == 0) return MPNHIP_OK;
    MPN_CHECK_ARG(dZ && H && grad_w, "weight_grad_bf16_rows: null pointer");
    const size_t need = align_up(wp_slab_floats(n_out, k_in, rows, nbatch, false, false, true) * sizeof(float), 256) + 256;
    if (!workspace || workspace_bytes < need) {
        set_error("weight_grad_bf16_rows: workspace %zu < %zu", workspace_bytes, need);
        return MPNHIP_ERR_WORKSPACE;
    }
    float* slab = reinterpret_cast<float*>((reinterpret_cast<uintptr_t>(workspace) + 255) / 256 * 256);
    WpProduct p = {reinterpret_cast<const float*>(dZ), n_out, rows * n_out, reinterpret_cast<const float*>(H), k_in, rows * k_in, nullptr, nullptr,
                   rows, nbatch, n_out, k_in, grad_w, k_in, grad_b, nullptr, nullptr, nullptr, 0, 0, 0, 1, 1};
    WpBatch b;
    WpBatchGuard guard;
    wp_batch_begin(&b, slab, (workspace_bytes - 256) / sizeof(float), false);
    if (wp_batch_add(p)) return wp_batch_flush(static_cast<hipStream_t>(stream));
    wp_batch_abort();
    set_error("weight_grad_bf16_rows: [%d x %d] over bf16 rows is not a shape of the row-panel kernel (multiples of 4, 8-byte aligned rows)", n_out, k_in);
    return MPNHIP_ERR_UNSUPPORTED;
}

extern "C" size_t mpnhip_weight_grad_bf16_rows_workspace_bytes(int n_out, int k_in, int64_t rows, int nbatch) {
    if (n_out < 1 || k_in < 1 || rows < 0) return 0;
    return align_up((rows > 0 ? wp_slab_floats(n_out, k_in, rows, nbatch < 1 ? 1 : nbatch, false, false, true) : 0) * sizeof(float), 256) + 256;
}

extern "C" int mpnhip_weight_grad(const float* dZ, const float* H, int64_t rows, int n_out, int k_in, int nbatch, float* grad_w,
                                  float* grad_b, void* workspace, size_t workspace_bytes, void* stream) {
    return mpnhip_weight_grad_prec(dZ, H, rows, n_out, k_in, nbatch, MPNHIP_PREC_FP32, grad_w, grad_b, workspace, workspace_bytes, stream);
}

extern "C" int mpnhip_time_weight_grad_prec(const float* dZ, const float* H, int64_t rows, int n_out, int k_in, int nbatch, int precision,
                                            float* grad_w, float* grad_b, void* workspace, size_t workspace_bytes, int iters, float* avg_us,
                                            void* stream_) {
    hipStream_t s = static_cast<hipStream_t>(stream_);
    MPN_CHECK_ARG(avg_us && iters > 0 && rows > 0, "time_weight_grad: bad argument");
    MPN_CHECK_ARG(precision == MPNHIP_PREC_FP32 || precision == MPNHIP_PREC_FP32_SPLIT || precision == MPNHIP_PREC_BF16, "time_weight_grad: precision %d", precision);
    TnArgs a;
    MPN_TRY(weight_grad_args(dZ, H, rows, n_out, k_in, nbatch, grad_w, grad_b, workspace, workspace_bytes, &a));
    hipEvent_t t0, t1;
    MPN_HIP(hipEventCreate(&t0));
    MPN_HIP(hipEventCreate(&t1));
    MPN_TRY(weight_grad_run(a, precision, workspace, workspace_bytes, s));
    MPN_HIP(hipEventRecord(t0, s));
    for (int i = 0; i < iters; ++i) MPN_TRY(weight_grad_run(a, precision, workspace, workspace_bytes, s));
    MPN_HIP(hipEventRecord(t1, s));
    MPN_HIP(hipEventSynchronize(t1));
    float ms = 0.f;
    MPN_HIP(hipEventElapsedTime(&ms, t0, t1));
    *avg_us = ms * 1000.f / iters;
    (void)hipEventDestroy(t0);
    (void)hipEventDestroy(t1);
    return MPNHIP_OK;
}

extern "C" int mpnhip_time_weight_grad(const float* dZ, const float* H, int64_t rows, int n_out, int k_in, int nbatch, float* grad_w,
                                       float* grad_b, void* workspace, size_t workspace_bytes, int iters, float* avg_us, void* stream_) {
    return mpnhip_time_weight_grad_prec(dZ, H, rows, n_out, k_in, nbatch, MPNHIP_PREC_FP32, grad_w, grad_b, workspace, workspace_bytes, iters,
                                        avg_us, stream_);
}
