// Full-line row stores for the chain kernels (edge_chain.hip, edge_chain_bf16.hip, edge_chain_bf16_bwd.hip): see RowStage.
#pragma once
#include <hip/hip_runtime.h>

namespace mpnhip {

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned lds_addr(const void* p) {
    return (unsigned)(size_t)(const __attribute__((address_space(3))) void*)p;
}

// ---- full-line row stores through a per-wave LDS slab ------------------------------------------------------------------------
// A wave holds a tile as (edge on the lane, features in registers): stored straight from there, one instruction writes 16- or
// 32-byte pieces of 32 DIFFERENT rows -- and the store path is bound by the lines an instruction touches, not by its bytes
// (tools/micro/store_pattern.hip: the same 512 MB as 32-byte pieces of 32 rows per instruction 1.63 TB/s, as 8 complete 128-byte
// lines per instruction 5.40 TB/s).  So a wave passes 128 bytes per row (two bf16 tiles or one fp32 tile) through its own slab of
// 32 rows x 144 bytes (pitch = 36 dwords: conflict-free both ways) and stores lane l = 16 bytes of row 8 i + l / 8, i = 0..3.
// A wave's LDS operations complete in order: no wait between the writes and the reads; inline assembly, because a compiler-visible
// LDS access next to an LDS-DMA in flight is preceded by s_waitcnt vmcnt(0).
constexpr int ROW_PITCH = 144;
constexpr int ROW_SLAB_BYTES = 32 * ROW_PITCH;
__device__ __forceinline__ void slab_write16(unsigned addr, const uint4& v) {
    const u32x4 q = {v.x, v.y, v.z, v.w};
    // (s_nop 1: hipcc pads no hazard in front of an asm statement, and the data registers may have been written by the instruction
    // just before it, e.g. a v_permlane32_swap of tile_rows16)
    asm volatile("s_nop 1\n\tds_write_b128 %0, %1" ::"v"(addr), "v"(q) : "memory");
}
__device__ __forceinline__ void slab_read4(unsigned addr, u32x4 (&v)[4]) {
    // rows 8 i + lane / 8: 8 * ROW_PITCH = 1152 bytes apart
    asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:1152\n\tds_read_b128 %2, %4 offset:2304\n\tds_read_b128 %3, %4 offset:3456\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]) : "v"(addr) : "memory");
}
struct RowStage {
    unsigned wr;      // LDS address this lane writes its pieces to: slab + lj * ROW_PITCH + 32 * lh
    unsigned rd;      // ... and reads its 16-byte pieces from: slab + (lane / 8) * ROW_PITCH + 16 * (lane % 8)
    int edge0, end;   // first edge of the wave tile, end of its direction group
    int lane;
    __device__ __forceinline__ void init(char* slab, int lane_, int edge0_, int end_) {
        lane = lane_; edge0 = edge0_; end = end_;
        wr = lds_addr(slab) + (lane_ & 31) * ROW_PITCH + 32 * (lane_ >> 5);
        rd = lds_addr(slab) + (lane_ >> 3) * ROW_PITCH + 16 * (lane_ & 7);
    }
    // one bf16 tile of a pair (parity 0 / 1): this lane's 32 contiguous bytes (tile_rows16's lo | hi) of its row
    __device__ __forceinline__ void put16(int parity, const uint4& lo, const uint4& hi) const {
        slab_write16(wr + 64 * parity, lo);
        slab_write16(wr + 64 * parity + 16, hi);
    }
    // one fp32 tile: the lane's four runs of four features (8 g + 4 lh ..) = bytes 32 g + 16 lh of the row's 128.
    // put32v: the tile's registers were last written by VALU instructions (ReLU, mask, sign flip) -- no hazard to pad
    __device__ __forceinline__ void put32v(const f32x16& a) const {
        const unsigned w = wr - 16 * (lane >> 5);   // (+ 16 lh instead of + 32 lh)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const uint4 v = make_uint4(__float_as_uint(a[4 * g + 0]), __float_as_uint(a[4 * g + 1]), __float_as_uint(a[4 * g + 2]), __float_as_uint(a[4 * g + 3]));
            slab_write16(w + 32 * g, v);
        }
    }
    __device__ __forceinline__ void put32(const f32x16& a_in) const {
        // The tile may be the raw result of an MFMA (the backward chain's dE_prev accumulators): an MFMA's D needs 12 wait states
        // (8-pass XDL) before anything but the next accumulating MFMA reads it, and hipcc pads no hazard whose reader sits inside an
        // asm statement (cdna_hip_programming.md section 5.7 item 2) -- a build in which the ds_write followed the last MFMA closely
        // stored stale accumulator values on some waves (flaky gradients); the wait states are tied to the tile here.
        f32x16 a = a_in;
        asm volatile("s_nop 15" : "+v"(a));
        put32v(a);
    }
    // the slab's 32 rows x 128 bytes to rows edge0 .. of a row-major matrix: `base` + row * row_bytes + col_bytes (+ 16 (lane % 8));
    // live_bytes: the row's bytes from col_bytes on that exist (partial last tiles); NT: non-temporal
    template <bool NT>
    __device__ __forceinline__ void flush(char* base, size_t row_bytes, int col_bytes, int live_bytes) const {
        u32x4 v[4];
        slab_read4(rd, v);
        const int pb = 16 * (lane & 7);
        if (pb >= live_bytes) return;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = edge0 + 8 * i + (lane >> 3);
            if (e < end) {
                u32x4* q = reinterpret_cast<u32x4*>(base + (size_t)e * row_bytes + col_bytes + pb);
                if (NT) __builtin_nontemporal_store(v[i], q);
                else *q = v[i];
            }
        }
    }
};

}  // namespace

}  // namespace mpnhip
