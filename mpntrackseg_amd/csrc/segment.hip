// Segmented neighbour aggregation (the reference's node_agg_fn, models/mpn.py:266-273 ->
// torch_scatter.scatter_{add,mean,max}) over CSR segments, without atomics.
//
// HBM-bound kernel: every message row is read exactly once with 16-byte loads; the dim/4 lanes that
// share a segment read one whole contiguous row per instruction (a 128-d row is one 512-B
// transaction of a half wave), and because the edges are sorted by (direction, row) a segment is a
// contiguous run of rows.  Each segment is summed strictly in ascending edge order -- the order of
// the reference's sequential CPU scatter -- with four independent row loads in flight, so results are
// bit-reproducible and independent of the launch geometry.
#include <cstring>

#include "common.h"

#include <rocprim/device/device_radix_sort.hpp>

namespace mpnhip {

struct SegArgs {
    const float* src;
    int64_t lds;
    const int* list;  // optional row indirection
    const int* ptr;   // [nseg + 1]
    int nseg;
    int dim;
    int agg;
    float* out;
    int64_t ldo;
    int* argmax;      // same geometry as out (ld = ldo), or nullptr
    int accumulate;
    int nmod;         // segment s -> out row s % nmod, column offset (s / nmod == 0 ? off0 : off1)
    int off0, off1;
    int sub;          // lanes per work item (power of two <= 64)
    int runs, run_stride;  // runs > 1 (sum only, no list): segment s is the union of the runs [ptr[s + r * run_stride],
                      // ptr[s + r * run_stride + 1]), r < runs -- e.g. a node's rows of all three directions of the (dir,row) CSR
    int nblk;         // column blocks per segment: work item = (segment, block of sub * VEC columns); 0 / 1 = one item per segment
    unsigned short* out16;  // (bf16-row sources, float4 path) the sums also as bf16 rows (RNE), same geometry with ld = ldo16; or nullptr
    int64_t ldo16;
};

template <int VEC>
struct Vec;
template <>
struct Vec<4> {
    typedef float4 T;
    static __device__ T load(const float* p) { return *reinterpret_cast<const float4*>(p); }
    static __device__ void store(float* p, T v) { *reinterpret_cast<float4*>(p) = v; }
};
template <>
struct Vec<1> {
    typedef float T;
    static __device__ T load(const float* p) { return *p; }
    static __device__ void store(float* p, T v) { *p = v; }
};

__device__ inline void vadd(float4& a, const float4& b) { a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; }
__device__ inline void vadd(float& a, const float& b) { a += b; }
__device__ inline void vdiv(float4& a, float c) { a.x /= c; a.y /= c; a.z /= c; a.w /= c; }
__device__ inline void vdiv(float& a, float c) { a /= c; }
__device__ inline void vset(float4& a, float v) { a = make_float4(v, v, v, v); }
__device__ inline void vset(float& a, float v) { a = v; }
// first maximum in index order wins (strict >), like torch_scatter's sequential CPU kernel
__device__ inline void vmax(float4& a, int* ia, const float4& b, int id) {
    if (b.x > a.x) { a.x = b.x; ia[0] = id; }
    if (b.y > a.y) { a.y = b.y; ia[1] = id; }
    if (b.z > a.z) { a.z = b.z; ia[2] = id; }
    if (b.w > a.w) { a.w = b.w; ia[3] = id; }
}
__device__ inline void vmax(float& a, int* ia, const float& b, int id) {
    if (b > a) { a = b; ia[0] = id; }
}

// B16: the source rows are bf16 (a.src points at unsigned shorts, a.lds counts them); sums stay fp32
template <int VEC, bool B16 = false>
__device__ __forceinline__ void seg_short_body(const SegArgs& a, const int gtid);
__device__ __forceinline__ float4 load_bf16x4(const float* base, int64_t elem) {
    const uint2 r = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(base) + elem);
    return make_float4(__uint_as_float(r.x << 16), __uint_as_float(r.x & 0xffff0000u), __uint_as_float(r.y << 16), __uint_as_float(r.y & 0xffff0000u));
}

template <int VEC>
__global__ __launch_bounds__(256) void k_segment_reduce(SegArgs a) {
    seg_short_body<VEC>(a, blockIdx.x * blockDim.x + threadIdx.x);
}

// three independent short-segment reductions in one launch (the backward's three scatter-adds of a step on sparse graphs: each
// alone is ~6,000 waves that live for three dependent memory round trips -- offsets, indices, rows -- i.e. latency-bound at
// ~25 us; together they overlap one another's round trips and pay one launch).  b0 / b1: blocks of the first / first two.
__global__ __launch_bounds__(256) void k_segment_reduce3(SegArgs a0, SegArgs a1, SegArgs a2, int b0, int b1) {
    const int b = blockIdx.x;
    if (b < b0) seg_short_body<4>(a0, b * 256 + threadIdx.x);
    else if (b < b1) seg_short_body<4>(a1, (b - b0) * 256 + threadIdx.x);
    else seg_short_body<4>(a2, (b - b1) * 256 + threadIdx.x);
}

// the same three reductions over bf16 source rows (the dZ blocks of the bf16-operand backward chain, edge_chain_bf16_bwd.hip)
__global__ __launch_bounds__(256) void k_segment_reduce3_b16(SegArgs a0, SegArgs a1, SegArgs a2, int b0, int b1) {
    const int b = blockIdx.x;
    if (b < b0) seg_short_body<4, true>(a0, b * 256 + threadIdx.x);
    else if (b < b1) seg_short_body<4, true>(a1, (b - b0) * 256 + threadIdx.x);
    else seg_short_body<4, true>(a2, (b - b1) * 256 + threadIdx.x);
}

template <int VEC, bool B16>
__device__ __forceinline__ void seg_short_body(const SegArgs& a, const int gtid) {
    typedef typename Vec<VEC>::T V;
    const int nblk = a.nblk > 1 ? a.nblk : 1;
    const int item = gtid / a.sub;
    const int s = item / nblk;
    const int l = gtid % a.sub + (item % nblk) * a.sub;
    if (s >= a.nseg) return;
    const int beg = a.ptr[s], end = a.ptr[s + 1];
    const int64_t orow = (int64_t)(s % a.nmod) * a.ldo + ((s / a.nmod) == 0 ? a.off0 : a.off1);
    const bool is_max = a.agg == MPNHIP_AGG_MAX;
    for (int c = l * VEC; c < a.dim; c += a.sub * nblk * VEC) {
        V acc;
        vset(acc, is_max ? -INFINITY : 0.f);
        int arg_s[4] = {-1, -1, -1, -1};
        const int nruns = a.runs > 1 ? a.runs : 1;
        for (int r = 0; r < nruns; ++r) {
            const int rbeg = r == 0 ? beg : a.ptr[s + r * a.run_stride], rend = r == 0 ? end : a.ptr[s + r * a.run_stride + 1];
            for (int j = rbeg; j < rend; j += 8) {
                // eight rows in flight: clamped (unconditional) index and row loads, predicated accumulation
                V v[8];
                int id[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int jj = j + u < rend ? j + u : rend - 1;
                    id[u] = a.list ? a.list[jj] : jj;
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    if constexpr (B16 && VEC == 4) v[u] = load_bf16x4(a.src, (int64_t)id[u] * a.lds + c);
                    else v[u] = Vec<VEC>::load(a.src + (int64_t)id[u] * a.lds + c);
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    if (j + u < rend) {
                        if (is_max) vmax(acc, arg_s, v[u], id[u]);
                        else vadd(acc, v[u]);
                    }
                }
            }
        }
        if (a.agg == MPNHIP_AGG_MEAN) {
            int cnt = end - beg;
            vdiv(acc, (float)(cnt > 0 ? cnt : 1));
        } else if (is_max && end == beg) {
            vset(acc, 0.f);
        }
        float* op = a.out + orow + c;
        if (a.accumulate) {
            V old = Vec<VEC>::load(op);
            vadd(acc, old);
        }
        Vec<VEC>::store(op, acc);
        if constexpr (B16 && VEC == 4) {
            if (a.out16) {   // (the consumers of these sums in the bf16-operand backward round them to bf16 anyway: once, here)
                typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
                const bf16x2 lo = {(__bf16)acc.x, (__bf16)acc.y}, hi = {(__bf16)acc.z, (__bf16)acc.w};
                const int64_t o16 = (int64_t)(s % a.nmod) * a.ldo16 + ((s / a.nmod) == 0 ? a.off0 : a.off1) + c;
                *reinterpret_cast<uint2*>(a.out16 + o16) = make_uint2(__builtin_bit_cast(unsigned, lo), __builtin_bit_cast(unsigned, hi));
            }
        }
        if (a.argmax) {
            int* ap = a.argmax + orow + c;
#pragma unroll
            for (int q = 0; q < VEC; ++q) ap[q] = arg_s[q];
        }
    }
}

// Forward neighbour aggregation of one message-passing step (the kernel the HBM-roofline target of
// BASELINE.json is quoted on).  Same arithmetic as k_segment_reduce<4> on the (direction, row) CSR of
// the sorted edges -- rows of a segment are contiguous, no index list -- with eight independent 16-byte
// row loads in flight per lane, so the typical segment (E/N = 10: five rows per direction) costs ONE
// memory round trip after the CSR offsets.  dim % 4 == 0.
__global__ __launch_bounds__(256) void k_aggregate(SegArgs a) {
    const int gtid = blockIdx.x * blockDim.x + threadIdx.x;
    const int s = gtid / a.sub;
    const int l = gtid % a.sub;
    if (s >= a.nseg) return;
    const int beg = a.ptr[s], end = a.ptr[s + 1];
    const int64_t orow = (int64_t)(s % a.nmod) * a.ldo + ((s / a.nmod) == 0 ? a.off0 : a.off1);
    const bool is_max = a.agg == MPNHIP_AGG_MAX;
    for (int c = l * 4; c < a.dim; c += a.sub * 4) {
        float4 acc;
        vset(acc, is_max ? -INFINITY : 0.f);
        int arg_s[4] = {-1, -1, -1, -1};
        const float* base = a.src + c;
        for (int j = beg; j < end; j += 8) {
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int jj = j + u < end ? j + u : end - 1;  // clamped: unconditional loads
                v[u] = *reinterpret_cast<const float4*>(base + (int64_t)jj * a.lds);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (j + u < end) {
                    if (is_max) vmax(acc, arg_s, v[u], j + u);
                    else vadd(acc, v[u]);
                }
            }
        }
        if (a.agg == MPNHIP_AGG_MEAN) {
            const int cnt = end - beg;
            vdiv(acc, (float)(cnt > 0 ? cnt : 1));
        } else if (is_max && end == beg) {
            vset(acc, 0.f);
        }
        *reinterpret_cast<float4*>(a.out + orow + c) = acc;
        if (a.argmax) {
            int* ap = a.argmax + orow + c;
#pragma unroll
            for (int q = 0; q < 4; ++q) ap[q] = arg_s[q];
        }
    }
}

// Long segments (dense kNN graphs: E / N ~ 150, a few hundred nodes): one 256-thread block per segment.  `sub`
// lanes cover the columns (16 bytes each), 256 / sub row lanes stride over the segment's rows, and the row lanes'
// partial results are combined in a fixed tree through LDS -- deterministic, but not the sequential order of the
// short-segment kernels.  dim % 4 == 0, dim <= 4 * 64.
__device__ __forceinline__ void seg_block_body(const SegArgs& a, const int s, float4* redv, int4* redi) {
    const int sub = a.sub, rl_n = 256 / sub;
    const int cl = threadIdx.x % sub, rl = threadIdx.x / sub;
    const int beg = a.ptr[s], end = a.ptr[s + 1];
    const int64_t orow = (int64_t)(s % a.nmod) * a.ldo + ((s / a.nmod) == 0 ? a.off0 : a.off1);
    const bool is_max = a.agg == MPNHIP_AGG_MAX;
    const int c = cl * 4;
    const bool col_ok = c < a.dim;
    float4 acc;
    vset(acc, is_max ? -INFINITY : 0.f);
    int arg_s[4] = {-1, -1, -1, -1};
    if (col_ok) {
        for (int j = beg + rl; j < end; j += 4 * rl_n) {
            float4 v[4];
            int id[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int jj = j + u * rl_n < end ? j + u * rl_n : end - 1;
                id[u] = a.list ? a.list[jj] : jj;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const float4*>(a.src + (int64_t)id[u] * a.lds + c);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (j + u * rl_n < end) {
                    if (is_max) vmax(acc, arg_s, v[u], id[u]);
                    else vadd(acc, v[u]);
                }
            }
        }
    }
    redv[threadIdx.x] = acc;
    redi[threadIdx.x] = make_int4(arg_s[0], arg_s[1], arg_s[2], arg_s[3]);
    __syncthreads();
    for (int w = rl_n >> 1; w > 0; w >>= 1) {
        if (rl < w) {
            float4 o = redv[threadIdx.x + w * sub];
            float4 m = redv[threadIdx.x];
            if (is_max) {
                int4 oi = redi[threadIdx.x + w * sub], mi = redi[threadIdx.x];
                // larger value wins; equal values: the smaller (earlier) edge index, as the sequential scan would pick
                auto pick = [](float& mv, int& mx, float ov, int ox) {
                    if (ov > mv || (ov == mv && ox >= 0 && (mx < 0 || ox < mx))) { mv = ov; mx = ox; }
                };
                pick(m.x, mi.x, o.x, oi.x); pick(m.y, mi.y, o.y, oi.y); pick(m.z, mi.z, o.z, oi.z); pick(m.w, mi.w, o.w, oi.w);
                redi[threadIdx.x] = mi;
            } else {
                m.x += o.x; m.y += o.y; m.z += o.z; m.w += o.w;
            }
            redv[threadIdx.x] = m;
        }
        __syncthreads();
    }
    if (rl == 0 && col_ok) {
        float4 r = redv[threadIdx.x];
        if (a.agg == MPNHIP_AGG_MEAN) {
            const int cnt = end - beg;
            vdiv(r, (float)(cnt > 0 ? cnt : 1));
        } else if (is_max && end == beg) {
            vset(r, 0.f);
        }
        float* op = a.out + orow + c;
        if (a.accumulate) vadd(r, *reinterpret_cast<const float4*>(op));
        *reinterpret_cast<float4*>(op) = r;
        if (a.argmax) *reinterpret_cast<int4*>(a.argmax + orow + c) = redi[threadIdx.x];
    }
}

__global__ __launch_bounds__(256) void k_segment_reduce_block(SegArgs a) {
    __shared__ float4 redv[256];
    __shared__ int4 redi[256];
    seg_block_body(a, blockIdx.x, redv, redi);
}

// three independent reductions in one launch (the backward's scatter-adds of a step on dense graphs: each alone is a few
// hundred blocks and ~10 us, mostly launch floor and tail); the branch is block-uniform and every call site reads its own
// kernel argument (selecting a struct by index would copy it to scratch)
__global__ __launch_bounds__(256) void k_segment_reduce_block3(SegArgs a0, SegArgs a1, SegArgs a2) {
    __shared__ float4 redv[256];
    __shared__ int4 redi[256];
    const int b = blockIdx.x;
    if (b < a0.nseg) seg_block_body(a0, b, redv, redi);
    else if (b < a0.nseg + a1.nseg) seg_block_body(a1, b - a0.nseg, redv, redi);
    else seg_block_body(a2, b - a0.nseg - a1.nseg, redv, redi);
}

// average segment length above which the block-per-segment kernel is used
constexpr int64_t LONG_SEGMENT = 48;

static bool block_eligible(SegArgs& a, int64_t total_rows) {
    const bool vec = (a.dim % 4 == 0) && (a.lds % 4 == 0) && (a.ldo % 4 == 0) && (a.off0 % 4 == 0) && (a.off1 % 4 == 0) &&
                     (((uintptr_t)a.src & 15) == 0) && (((uintptr_t)a.out & 15) == 0) && (!a.argmax || ((uintptr_t)a.argmax & 15) == 0);
    if (!vec || a.dim > 256 || a.nseg <= 0 || total_rows < LONG_SEGMENT * a.nseg) return false;
    int sub = 1;
    while (sub < a.dim / 4) sub <<= 1;
    a.sub = sub;
    return true;
}

static bool try_launch_block(SegArgs& a, int64_t total_rows, hipStream_t stream) {
    const bool vec = (a.dim % 4 == 0) && (a.lds % 4 == 0) && (a.ldo % 4 == 0) && (a.off0 % 4 == 0) && (a.off1 % 4 == 0) &&
                     (((uintptr_t)a.src & 15) == 0) && (((uintptr_t)a.out & 15) == 0) && (!a.argmax || ((uintptr_t)a.argmax & 15) == 0);
    if (!vec || a.dim > 256 || a.nseg <= 0 || total_rows < LONG_SEGMENT * a.nseg) return false;
    int sub = 1;
    while (sub < a.dim / 4) sub <<= 1;
    a.sub = sub;
    count_path(PC_SEG_BLOCK);
    MPN_LAUNCH_PROFILED(k_segment_reduce_block, dim3(a.nseg), dim3(256), stream, a);
    return true;
}

// lanes per work item / column blocks of the short-segment kernel; returns the number of 256-thread blocks, *vec_out = float4 path
static unsigned seg_short_geometry(SegArgs& a, bool* vec_out) {
    bool vec = (a.dim % 4 == 0) && (a.lds % 4 == 0) && (a.ldo % 4 == 0) && (a.off0 % 4 == 0) && (a.off1 % 4 == 0) &&
               (((uintptr_t)a.src & 15) == 0) && (((uintptr_t)a.out & 15) == 0);
    int per = vec ? a.dim / 4 : a.dim;
    int sub = 1;
    while (sub < per && sub < 64) sub <<= 1;
    a.nblk = 1;
    if (per > sub || (per & (per - 1))) {
        int p2 = per & -per;
        if (p2 > 64) p2 = 64;
        if (p2 * (vec ? 16 : 4) >= 256) { sub = p2; a.nblk = per / p2; }
    }
    a.sub = sub;
    *vec_out = vec;
    return (unsigned)(((int64_t)a.nseg * sub * a.nblk + 255) / 256);
}

static int launch_seg(SegArgs a, hipStream_t stream) {
    if (a.nseg <= 0 || a.dim <= 0) return MPNHIP_OK;
    // (rows that are not one power-of-two group of lanes, e.g. 80 or 56 x 16 bytes: column blocks of the largest power of two
    // that divides them, one work item per (segment, block), so that no lane idles and no lane walks the segment twice; rows
    // are still read in pieces of >= 256 contiguous bytes -- 128-byte pieces measured slower than idle lanes)
    bool vec = false;
    const unsigned blocks = seg_short_geometry(a, &vec);
    count_path(PC_SEG_SHORT);
    if (vec) hipLaunchKernelGGL(k_segment_reduce<4>, dim3(blocks), dim3(256), 0, stream, a);
    else hipLaunchKernelGGL(k_segment_reduce<1>, dim3(blocks), dim3(256), 0, stream, a);
    MPN_LAUNCH_CHECK();
    return MPNHIP_OK;
}

// ---- fused node side of one inference step at the reference's width (dn = 32) ------------------------------------------
// aggregate (node_agg_fn, mpn.py:89,96) -> node update x' = relu(W [agg_in | agg_out] + b) (mpn.py:97-99) -> the NEXT step's
// per-node projections P = P0 + x' Wx^T, in ONE launch.  At these sizes (a few hundred nodes, 32-d features) the three
// separate kernels are each at the ~4.5 us floor of a dependent launch; the work itself is a few microseconds.
// Block = 2 nodes: one wave per (node, direction) segment (8 column lanes x 8 row lanes, four 16-byte row loads in flight per
// lane, fixed shuffle tree across the row lanes), then the two small products with plain fp32 FMAs (W rows from L2).
struct NodeStepArgs {
    const float* msg;        // [E, 32] messages in sorted edge order
    const int* seg_ptr;      // CSR over keys dir * N + row
    int N, agg;
    const float* Wu;         // node update Linear [32, 64] (nn.Linear layout), bias bu [32]
    const float* bu;
    float* x_new;            // [N, 32]
    float* agg_out;          // [N, 64] aggregated messages kept for the backward pass (training), or nullptr
    const float* Wx;         // projection weights of the CURRENT features: row p at Wx + p * ldwx, 32 columns
    int64_t ldwx;
    const float* P0;         // [N, pw] step-invariant share (+ biases)
    float* P;                // [N, pw] out, or nullptr (last step)
    int pw;
};

__global__ __launch_bounds__(256) void k_node_step32(NodeStepArgs a) {
    __shared__ __attribute__((aligned(16))) float agg_s[2][64];
    __shared__ __attribute__((aligned(16))) float x_s[2][32];
    const int tid = threadIdx.x, lane = tid & 63, seg = tid >> 6;
    const int node_l = seg >> 1, dir = seg & 1;
    const int node = blockIdx.x * 2 + node_l;
    const bool node_ok = node < a.N;
    // ---- aggregation: this wave's segment -----------------------------------------------------------------------
    {
        const int c4 = lane & 7, rl = lane >> 3;
        const int key = dir * a.N + (node_ok ? node : 0);
        const int beg = node_ok ? a.seg_ptr[key] : 0, end = node_ok ? a.seg_ptr[key + 1] : 0;
        const bool is_max = a.agg == MPNHIP_AGG_MAX;
        float4 acc = is_max ? make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY) : make_float4(0.f, 0.f, 0.f, 0.f);
        const float* base = a.msg + c4 * 4;
        for (int j = beg + rl; j < end; j += 32) {
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int jj = j + 8 * u < end ? j + 8 * u : end - 1;   // clamped: unconditional loads
                v[u] = *reinterpret_cast<const float4*>(base + (int64_t)jj * 32);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (j + 8 * u < end) {
                    if (is_max) { acc.x = fmaxf(acc.x, v[u].x); acc.y = fmaxf(acc.y, v[u].y); acc.z = fmaxf(acc.z, v[u].z); acc.w = fmaxf(acc.w, v[u].w); }
                    else { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
                }
        }
#pragma unroll
        for (int m = 8; m < 64; m <<= 1) {   // the 8 row lanes of a column: fixed tree
            const float ox = __shfl_xor(acc.x, m, 64), oy = __shfl_xor(acc.y, m, 64), oz = __shfl_xor(acc.z, m, 64), ow = __shfl_xor(acc.w, m, 64);
            if (is_max) { acc.x = fmaxf(acc.x, ox); acc.y = fmaxf(acc.y, oy); acc.z = fmaxf(acc.z, oz); acc.w = fmaxf(acc.w, ow); }
            else { acc.x += ox; acc.y += oy; acc.z += oz; acc.w += ow; }
        }
        const int cnt = end - beg;
        if (a.agg == MPNHIP_AGG_MEAN) { const float d = (float)(cnt > 0 ? cnt : 1); acc.x /= d; acc.y /= d; acc.z /= d; acc.w /= d; }
        else if (is_max && cnt == 0) acc = make_float4(0.f, 0.f, 0.f, 0.f);
        // torch.cat((flow_in, flow_out)) (mpn.py:97): direction 0 (row < col, flow_out) is the right half
        if (rl == 0) *reinterpret_cast<float4*>(&agg_s[node_l][(dir == 0 ? 32 : 0) + c4 * 4]) = acc;
    }
    __syncthreads();
    if (a.agg_out && tid < 32) {   // (training) [flow_in | flow_out] of the two nodes, 16 bytes per thread
        const int nl = tid >> 4, q = tid & 15;
        if (blockIdx.x * 2 + nl < a.N)
            *reinterpret_cast<float4*>(a.agg_out + (int64_t)(blockIdx.x * 2 + nl) * 64 + q * 4) = *reinterpret_cast<const float4*>(&agg_s[nl][q * 4]);
    }
    // ---- node update: 64 outputs (2 nodes x 32), four threads per output over K = 64 ---------------------------------
    {
        const int o = tid >> 2, kq = tid & 3;
        const int nl = o >> 5, c = o & 31;
        const float* w = a.Wu + c * 64 + kq * 16;
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 16; i += 4) {
            const float4 wv = *reinterpret_cast<const float4*>(w + i);
            const float4 av = *reinterpret_cast<const float4*>(&agg_s[nl][kq * 16 + i]);
            s = fmaf(av.x, wv.x, s); s = fmaf(av.y, wv.y, s); s = fmaf(av.z, wv.z, s); s = fmaf(av.w, wv.w, s);
        }
        s += __shfl_xor(s, 1, 64);
        s += __shfl_xor(s, 2, 64);
        if (kq == 0) {
            const float xv = fmaxf(s + a.bu[c], 0.f);
            x_s[nl][c] = xv;
            if (blockIdx.x * 2 + nl < a.N) a.x_new[(int64_t)(blockIdx.x * 2 + nl) * 32 + c] = xv;
        }
    }
    if (!a.P) return;   // (uniform)
    __syncthreads();
    // ---- the next step's projections of these two nodes: P = P0 + x' Wx^T -------------------------------------------------
    for (int p = tid; p < a.pw; p += 256) {
        const float* w = a.Wx + (int64_t)p * a.ldwx;
        float s0 = 0.f, s1 = 0.f;
#pragma unroll
        for (int i = 0; i < 32; i += 4) {
            const float4 wv = *reinterpret_cast<const float4*>(w + i);
            const float4 x0 = *reinterpret_cast<const float4*>(&x_s[0][i]);
            const float4 x1 = *reinterpret_cast<const float4*>(&x_s[1][i]);
            s0 = fmaf(x0.x, wv.x, s0); s0 = fmaf(x0.y, wv.y, s0); s0 = fmaf(x0.z, wv.z, s0); s0 = fmaf(x0.w, wv.w, s0);
            s1 = fmaf(x1.x, wv.x, s1); s1 = fmaf(x1.y, wv.y, s1); s1 = fmaf(x1.z, wv.z, s1); s1 = fmaf(x1.w, wv.w, s1);
        }
        const int n0 = blockIdx.x * 2;
        if (n0 < a.N) a.P[(int64_t)n0 * a.pw + p] = a.P0[(int64_t)n0 * a.pw + p] + s0;
        if (n0 + 1 < a.N) a.P[(int64_t)(n0 + 1) * a.pw + p] = a.P0[(int64_t)(n0 + 1) * a.pw + p] + s1;
    }
}

int node_step32(const GraphView& g, const float* msg, int agg, const float* Wu, const float* bu, float* x_new, float* agg_out,
                const float* Wx, int64_t ldwx, const float* P0, float* P, int pw, hipStream_t stream) {
    if (g.N <= 0) return MPNHIP_OK;
    NodeStepArgs a = {msg, g.seg_ptr, g.N, agg, Wu, bu, x_new, agg_out, Wx, ldwx, P0, P, pw};
    count_path(PC_NODE_STEP32);
    hipLaunchKernelGGL(k_node_step32, dim3((unsigned)((g.N + 1) / 2)), dim3(256), 0, stream, a);
    MPN_LAUNCH_CHECK();
    return MPNHIP_OK;
}

// Backward counterpart at the same width: the activation gradient of step s's projections, the ReLU mask of the node update of
// step s - 1 and that update's activation gradient, for two nodes per block (fp32 FMAs, weight rows from L2):
//   dX = dP Wx  ([pw] x [pw, 32]);   dZn = dX (.) [x_{s-1} > 0];   dAGG = dZn Wu  ([32] x [32, 64])
// -- three dependent launches (a grouped GEMM, k_relu_mask, a GEMM) in one.
struct NodeStepBwdArgs {
    const float* dP;         // [N, pw] gradient of this step's projections
    int N, pw;
    const float* Wx;         // rows p of the packed projection weights, current-feature columns: Wx + p * ldwx, 32 columns
    int64_t ldwx;
    const float* x_prev;     // [N, 32] output of the previous step's node update (its ReLU mask)
    const float* Wu;         // node update Linear [32, 64]
    float* dZn;              // [N, 32] out: gradient at the previous step's node-update pre-activation
    float* dAGG;             // [N, 64] out
};

// Block = 4 nodes.  The projection weights' current-feature columns (pw x 32 floats, 35 KB at the reference dims) and the node
// update weights (8 KB) are staged in LDS with coalesced 16-byte loads -- the first version read them element by element from L2
// inside the contraction loops (68 dependent loads per thread: 20-28 us per launch, latency); now ~6 us.
constexpr int NSB_NODES = 4;
__global__ __launch_bounds__(256) void k_node_step32_bwd(NodeStepBwdArgs a) {
    extern __shared__ float nsb_smem[];
    float* wx_s = nsb_smem;                          // [pw][32]
    float* wu_s = wx_s + (size_t)a.pw * 32;          // [32][64]
    float* dp_s = wu_s + 32 * 64;                    // [4][pw]
    float* part = dp_s + NSB_NODES * (size_t)a.pw;   // [2][128]
    float* dz_s = part + 2 * 128;                    // [4][32]
    const int tid = threadIdx.x;
    const int n0 = blockIdx.x * NSB_NODES;
    for (int i = tid; i < a.pw * 8; i += 256) {      // rows of 32 floats = 8 float4 (ldwx % 4 == 0, 16-byte aligned: checked by the launcher)
        const int pp = i >> 3, q = i & 7;
        *reinterpret_cast<float4*>(wx_s + pp * 32 + 4 * q) = *reinterpret_cast<const float4*>(a.Wx + (int64_t)pp * a.ldwx + 4 * q);
    }
    for (int i = tid; i < 32 * 16; i += 256) *reinterpret_cast<float4*>(wu_s + 4 * i) = *reinterpret_cast<const float4*>(a.Wu + 4 * i);
    for (int i = tid; i < NSB_NODES * a.pw; i += 256) {
        const int nl = i / a.pw, pidx = i - nl * a.pw;
        dp_s[i] = n0 + nl < a.N ? a.dP[(int64_t)(n0 + nl) * a.pw + pidx] : 0.f;
    }
    __syncthreads();
    {   // dX: 4 nodes x 32 outputs, the contraction over pw split in two halves (fixed order when they meet)
        const int o = tid & 127, half = tid >> 7;
        const int nl = o >> 5, c = o & 31;
        const int per = (a.pw + 1) / 2, p0 = half * per, p1 = p0 + per < a.pw ? p0 + per : a.pw;
        const float* dp = dp_s + nl * a.pw;
        float s0 = 0.f, s1 = 0.f;
        int pp = p0;
        for (; pp + 1 < p1; pp += 2) {
            s0 = fmaf(dp[pp], wx_s[pp * 32 + c], s0);
            s1 = fmaf(dp[pp + 1], wx_s[(pp + 1) * 32 + c], s1);
        }
        if (pp < p1) s0 = fmaf(dp[pp], wx_s[pp * 32 + c], s0);
        part[half * 128 + o] = s0 + s1;
    }
    __syncthreads();
    if (tid < 128) {
        const int nl = tid >> 5, c = tid & 31;
        const float dx = part[tid] + part[128 + tid];
        const bool ok = n0 + nl < a.N;
        const float xv = ok ? a.x_prev[(int64_t)(n0 + nl) * 32 + c] : 0.f;
        const float dz = xv > 0.f ? dx : 0.f;
        dz_s[tid] = dz;
        if (ok) a.dZn[(int64_t)(n0 + nl) * 32 + c] = dz;
    }
    __syncthreads();
    {   // dAGG: 4 nodes x 64 outputs, 32 FMAs each
        const int nl = tid >> 6, j = tid & 63;
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < 32; ++c) s = fmaf(dz_s[nl * 32 + c], wu_s[c * 64 + j], s);
        if (n0 + nl < a.N) a.dAGG[(int64_t)(n0 + nl) * 64 + j] = s;
    }
}

int node_step32_bwd(const float* dP, int N, int pw, const float* Wx, int64_t ldwx, const float* x_prev, const float* Wu, float* dZn,
                    float* dAGG, hipStream_t stream) {
    if (N <= 0) return MPNHIP_OK;
    if ((ldwx & 3) || (((uintptr_t)Wx | (uintptr_t)Wu) & 15)) { set_error("node_step32_bwd: weights must be 16-byte aligned"); return MPNHIP_ERR_ARG; }
    // projection columns + node-update weights + the block's dP rows: 63.5 KB at the reference's pw = 384; the launch is made
    // without raising the kernel's dynamic-LDS limit, so anything above 64 KB is refused here (the caller's gate is pw <= 384)
    const size_t smem = ((size_t)pw * 32 + 32 * 64 + NSB_NODES * (size_t)pw + 2 * 128 + NSB_NODES * 32) * sizeof(float);
    if (pw < 1 || smem > 65536) { set_error("node_step32_bwd: projection width %d needs %zu bytes of LDS (> 64 KB)", pw, smem); return MPNHIP_ERR_UNSUPPORTED; }
    NodeStepBwdArgs a = {dP, N, pw, Wx, ldwx, x_prev, Wu, dZn, dAGG};
    count_path(PC_NODE_STEP32_BWD);
    hipLaunchKernelGGL(k_node_step32_bwd, dim3((unsigned)((N + NSB_NODES - 1) / NSB_NODES)), dim3(256), smem, stream, a);
    MPN_LAUNCH_CHECK();
    return MPNHIP_OK;
}

int aggregate(const GraphView& g, const float* src, int dim, int agg, float* out, int* argmax, hipStream_t stream) {
    SegArgs a = {};
    a.src = src;
    a.lds = dim;
    a.list = nullptr;
    a.ptr = g.seg_ptr;
    a.nseg = 2 * g.N;  // keys [0,N): flow_out segments, [N,2N): flow_in segments
    a.dim = dim;
    a.agg = agg;
    a.out = out;
    a.ldo = 2 * (int64_t)dim;
    a.argmax = argmax;
    a.nmod = g.N > 0 ? g.N : 1;
    a.off0 = dim;  // flow_out goes to the right half: torch.cat((flow_in, flow_out)) (mpn.py:97)
    a.off1 = 0;
    if (try_launch_block(a, g.E, stream)) {
        count_path(PC_AGGREGATE_BLOCK);
        MPN_LAUNCH_CHECK();
        return MPNHIP_OK;
    }
    if (dim % 4 == 0 && (((uintptr_t)src | (uintptr_t)out) & 15) == 0 && a.nseg > 0) {
        int sub = 1;
        while (sub < dim / 4 && sub < 64) sub <<= 1;
        a.sub = sub;
        const int64_t threads = (int64_t)a.nseg * sub;
        count_path(PC_AGGREGATE);
        MPN_LAUNCH_PROFILED(k_aggregate, dim3((unsigned)((threads + 255) / 256)), dim3(256), stream, a);
        MPN_LAUNCH_CHECK();
        return MPNHIP_OK;
    }
    return launch_seg(a, stream);
}

int segment_reduce_csr(const float* src, int64_t lds, const int* list, const int* ptr, int nseg, int dim, int agg,
                       float* out, int64_t ldo, int* argmax, int accumulate, hipStream_t stream) {
    SegArgs a = {};
    a.src = src;
    a.lds = lds;
    a.list = list;
    a.ptr = ptr;
    a.nseg = nseg;
    a.dim = dim;
    a.agg = agg;
    a.out = out;
    a.ldo = ldo;
    a.argmax = argmax;
    a.accumulate = accumulate;
    a.nmod = nseg > 0 ? nseg : 1;
    return launch_seg(a, stream);
}

// segments [0, nmod) go to column offset off0, segments [nmod, 2 nmod) to off1 of out row (s % nmod)
int segment_reduce_csr2(const float* src, int64_t lds, const int* list, const int* ptr, int nseg, int dim, float* out,
                        int64_t ldo, int nmod, int off0, int off1, hipStream_t stream, int64_t total_rows, int runs, int run_stride) {
    SegArgs a = {};
    a.src = src;
    a.lds = lds;
    a.list = list;
    a.ptr = ptr;
    a.nseg = nseg;
    a.dim = dim;
    a.agg = MPNHIP_AGG_SUM;
    a.out = out;
    a.ldo = ldo;
    a.nmod = nmod > 0 ? nmod : 1;
    a.off0 = off0;
    a.off1 = off1;
    a.runs = runs;
    a.run_stride = run_stride;
    if (runs <= 1 && try_launch_block(a, total_rows, stream)) {
        MPN_LAUNCH_CHECK();
        return MPNHIP_OK;
    }
    return launch_seg(a, stream);
}

static SegArgs seg_args2(const SegReduce2& c) {
    SegArgs a = {};
    a.src = c.src; a.lds = c.lds; a.list = c.list; a.ptr = c.ptr; a.nseg = c.nseg; a.dim = c.dim; a.agg = MPNHIP_AGG_SUM;
    a.out = c.out; a.ldo = c.ldo; a.nmod = c.nmod > 0 ? c.nmod : 1; a.off0 = c.off0; a.off1 = c.off1;
    a.runs = c.runs; a.run_stride = c.run_stride;
    a.out16 = c.out16; a.ldo16 = c.ldo16;
    return a;
}

// three segment_reduce_csr2 calls; ONE launch when all three take the block-per-segment kernel (dense graphs)
int segment_reduce_csr2_x3(const SegReduce2 c[3], int64_t total_rows, hipStream_t stream) {
    SegArgs a[3] = {seg_args2(c[0]), seg_args2(c[1]), seg_args2(c[2])};
    const bool no_runs = c[0].runs <= 1 && c[1].runs <= 1 && c[2].runs <= 1;
    if (no_runs && block_eligible(a[0], total_rows) && block_eligible(a[1], total_rows) && block_eligible(a[2], total_rows)) {
        count_path(PC_SEG_BLOCK3);
        hipLaunchKernelGGL(k_segment_reduce_block3, dim3(a[0].nseg + a[1].nseg + a[2].nseg), dim3(256), 0, stream, a[0], a[1], a[2]);
        MPN_LAUNCH_CHECK();
        return MPNHIP_OK;
    }
    // sparse graphs (short segments): the three in one launch of the short-segment kernel when all take its float4 path and none
    // would take the block-per-segment kernel on its own
    {
        bool v[3], blk = false;
        unsigned nb[3];
        for (int i = 0; i < 3; ++i) {
            SegArgs t = a[i];
            blk = blk || (c[i].runs <= 1 && block_eligible(t, total_rows));
            nb[i] = a[i].nseg > 0 && a[i].dim > 0 ? seg_short_geometry(a[i], &v[i]) : 0;
            if (nb[i] == 0) v[i] = true;
        }
        if (!blk && v[0] && v[1] && v[2] && nb[0] + nb[1] + nb[2] > 0 && !getenv("MPNHIP_NO_SEG3")) {
            count_path(PC_SEG_SHORT3);
            hipLaunchKernelGGL(k_segment_reduce3, dim3(nb[0] + nb[1] + nb[2]), dim3(256), 0, stream, a[0], a[1], a[2], (int)nb[0], (int)(nb[0] + nb[1]));
            MPN_LAUNCH_CHECK();
            return MPNHIP_OK;
        }
    }
    for (int i = 0; i < 3; ++i)
        MPN_TRY(segment_reduce_csr2(c[i].src, c[i].lds, c[i].list, c[i].ptr, c[i].nseg, c[i].dim, c[i].out, c[i].ldo, c[i].nmod, c[i].off0,
                                    c[i].off1, stream, total_rows, c[i].runs > 1 ? c[i].runs : 1, c[i].run_stride));
    return MPNHIP_OK;
}

// the three scatter-adds of a backward step over bf16 source rows: always the short-segment kernel, one launch
int segment_reduce_csr2_x3_bf16(const SegReduce2 c[3], hipStream_t stream) {
    SegArgs a[3] = {seg_args2(c[0]), seg_args2(c[1]), seg_args2(c[2])};
    unsigned nb[3];
    for (int i = 0; i < 3; ++i) {
        bool v = false;
        // (the geometry's 16-byte source alignment test is the fp32 kernels'; bf16 rows need 8 bytes per 4 columns)
        MPN_CHECK_ARG(a[i].dim % 4 == 0 && a[i].lds % 4 == 0 && a[i].ldo % 4 == 0 && a[i].off0 % 4 == 0 && a[i].off1 % 4 == 0 &&
                      (((uintptr_t)a[i].src) & 7) == 0 && (((uintptr_t)a[i].out) & 15) == 0 && a[i].ldo16 % 4 == 0 &&
                      (((uintptr_t)a[i].out16) & 7) == 0, "segment_reduce (bf16 rows): alignment");
        const float* keep = a[i].src;
        a[i].src = reinterpret_cast<const float*>(((uintptr_t)keep) & ~(uintptr_t)15);   // (only for the geometry's alignment test)
        nb[i] = a[i].nseg > 0 && a[i].dim > 0 ? seg_short_geometry(a[i], &v) : 0;
        a[i].src = keep;
    }
    if (nb[0] + nb[1] + nb[2] == 0) return MPNHIP_OK;
    count_path(PC_SEG_SHORT3);
    hipLaunchKernelGGL(k_segment_reduce3_b16, dim3(nb[0] + nb[1] + nb[2]), dim3(256), 0, stream, a[0], a[1], a[2], (int)nb[0], (int)(nb[0] + nb[1]));
    MPN_LAUNCH_CHECK();
    return MPNHIP_OK;
}

// ---- stand-alone node_agg_fn with an arbitrary (unsorted) int64 index ---------------------------
__global__ void k_row_keys(const int64_t* __restrict__ row, int64_t M, int x_size, unsigned* __restrict__ keys,
                           int* __restrict__ vals) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M) return;
    int64_t r = row[i];
    // out-of-range rows are parked behind the last segment and never read
    keys[i] = (r < 0 || r >= x_size) ? (unsigned)x_size : (unsigned)r;
    vals[i] = (int)i;
}

__global__ void k_lower_bound_u32(const unsigned* __restrict__ skeys, int64_t M, int nkeys, int* __restrict__ ptr) {
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k > nkeys) return;
    int64_t lo = 0, hi = M;
    while (lo < hi) {
        int64_t mid = (lo + hi) >> 1;
        if (skeys[mid] < (unsigned)k) lo = mid + 1; else hi = mid;
    }
    ptr[k] = (int)lo;
}

static size_t seg_sort_temp(int64_t M) {
    size_t bytes = 0;
    unsigned* k = nullptr;
    int* v = nullptr;
    (void)rocprim::radix_sort_pairs(nullptr, bytes, k, k, v, v, (size_t)(M > 0 ? M : 1), 0, 32, (hipStream_t)0);
    return bytes;
}

}  // namespace mpnhip

using namespace mpnhip;

extern "C" size_t mpnhip_segment_reduce_workspace_bytes(int64_t m, int x_size) {
    size_t e = align_up((size_t)(m > 0 ? m : 1) * 4, 256);
    return 4 * e + align_up(((size_t)x_size + 2) * 4, 256) + align_up(seg_sort_temp(m), 256) + 256;
}

extern "C" int mpnhip_segment_reduce(const float* src, const int64_t* row, int64_t m, int dim, int x_size, int agg,
                                     float* out, int32_t* argmax, void* workspace, size_t workspace_bytes, void* stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    MPN_CHECK_ARG(m >= 0 && dim >= 0 && x_size >= 0 && m < 2147483647LL, "segment_reduce: bad sizes");
    MPN_CHECK_ARG(agg >= 0 && agg <= 2, "segment_reduce: unknown aggregation %d", agg);
    MPN_CHECK_ARG(out || x_size == 0 || dim == 0, "segment_reduce: null output");
    if (x_size == 0 || dim == 0) return MPNHIP_OK;
    if (m == 0) {
        MPN_HIP(hipMemsetAsync(out, 0, (size_t)x_size * dim * sizeof(float), stream));
        if (argmax) MPN_HIP(hipMemsetAsync(argmax, 0xFF, (size_t)x_size * dim * sizeof(int), stream));
        return MPNHIP_OK;
    }
    MPN_CHECK_ARG(src && row, "segment_reduce: null input");
    if (!workspace || workspace_bytes < mpnhip_segment_reduce_workspace_bytes(m, x_size)) {
        set_error("segment_reduce: workspace %zu < %zu", workspace_bytes, mpnhip_segment_reduce_workspace_bytes(m, x_size));
        return MPNHIP_ERR_WORKSPACE;
    }
    char* ws = static_cast<char*>(workspace);
    size_t e = align_up((size_t)m * 4, 256);
    unsigned* keys_in = reinterpret_cast<unsigned*>(ws);
    unsigned* keys_out = reinterpret_cast<unsigned*>(ws + e);
    int* vals_in = reinterpret_cast<int*>(ws + 2 * e);
    int* list = reinterpret_cast<int*>(ws + 3 * e);
    int* ptr = reinterpret_cast<int*>(ws + 4 * e);
    size_t poff = 4 * e + align_up(((size_t)x_size + 2) * 4, 256);
    void* tmp = ws + poff;
    size_t tmp_bytes = workspace_bytes - poff;
    const int T = 256;
    hipLaunchKernelGGL(k_row_keys, dim3((unsigned)((m + T - 1) / T)), dim3(T), 0, stream, row, m, x_size, keys_in, vals_in);
    MPN_LAUNCH_CHECK();
    int bits = 1;
    while (bits < 32 && ((unsigned)x_size >> bits)) ++bits;
    MPN_HIP(rocprim::radix_sort_pairs(tmp, tmp_bytes, keys_in, keys_out, vals_in, list, (size_t)m, 0, bits, stream));
    hipLaunchKernelGGL(k_lower_bound_u32, dim3((x_size + 1 + T) / T), dim3(T), 0, stream, keys_out, m, x_size, ptr);
    MPN_LAUNCH_CHECK();
    return segment_reduce_csr(src, dim, list, ptr, x_size, dim, agg, out, dim, argmax, 0, stream);
}

extern "C" int mpnhip_time_aggregate(const void* graph_buf, int n_nodes, int64_t n_edges, const float* src, int dim,
                                     int agg, float* out, int iters, float* avg_us, void* stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    MPN_CHECK_ARG(graph_buf && src && out && avg_us && iters > 0, "time_aggregate: bad argument");
    GraphView g;
    graph_layout(n_nodes, n_edges, &g, const_cast<void*>(graph_buf));
    hipEvent_t t0, t1;
    MPN_HIP(hipEventCreate(&t0));
    MPN_HIP(hipEventCreate(&t1));
    MPN_TRY(aggregate(g, src, dim, agg, out, nullptr, stream));  // warm-up
    MPN_HIP(hipEventRecord(t0, stream));
    for (int i = 0; i < iters; ++i) MPN_TRY(aggregate(g, src, dim, agg, out, nullptr, stream));
    MPN_HIP(hipEventRecord(t1, stream));
    MPN_HIP(hipEventSynchronize(t1));
    float ms = 0.f;
    MPN_HIP(hipEventElapsedTime(&ms, t0, t1));
    *avg_us = ms * 1000.f / iters;
    (void)hipEventDestroy(t0);
    (void)hipEventDestroy(t1);
    return MPNHIP_OK;
}
