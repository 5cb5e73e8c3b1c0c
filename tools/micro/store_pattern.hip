// Store-pattern micro-benchmark (round 4): what the row saves of the bf16 chain kernels cost by themselves.
// A wave owns 32 consecutive rows of a [E, W] bf16 matrix (W = 640: 1280-byte rows) and writes them tile by tile (32 features =
// 64 bytes per row and tile), 16 bytes per lane and instruction:
//   pattern 0 (the kernels' layout after v_permlane32_swap): lane (row r = l % 32, half h = l / 32) writes bytes [64 t + 32 h, + 32)
//             of row r -- an instruction touches 32 rows, 2 x 16 bytes each;
//   pattern 2 (the fp32 chain kernels' layout, fp32 rows of 4 W bytes): lane (r, h) writes 16 bytes at byte 128 t + 32 g + 16 h of
//             row r, g = 0..3 -- an instruction touches 32 rows, 32 contiguous bytes each;
//   pattern 1 (full lines): two tiles at a time, lane l writes 16 bytes at row 8 i + l / 8, bytes [128 (t / 2) + 16 (l % 8), + 16),
//             i = 0..3 -- an instruction writes 8 complete 128-byte lines.
// Same bytes, same instruction count.  build/micro/store_pattern [rows] [reps]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int PATTERN>
__global__ __launch_bounds__(256, 2) void k_store(unsigned short* base, int rows, int W, unsigned seed) {
    const int lane = threadIdx.x & 63, wave = (blockIdx.x * 4) + (threadIdx.x >> 6);
    const int r0 = wave * 32;
    if (r0 >= rows) return;
    const int T = W / 32;
    uint4 v = make_uint4(seed + lane, seed * 3 + wave, lane, wave);
    if (PATTERN == 0) {
        const int r = r0 + (lane & 31), h = lane >> 5;
        unsigned short* row = base + (size_t)r * W;
        for (int t = 0; t < T; ++t) {
            *reinterpret_cast<uint4*>(row + 32 * t + 16 * h) = v;
            *reinterpret_cast<uint4*>(row + 32 * t + 16 * h + 8) = v;
            v.x += 1;
        }
    } else if (PATTERN == 2) {
        const int r = r0 + (lane & 31), h = lane >> 5;
        unsigned short* row = base + (size_t)r * W;
        for (int t = 0; t < T; t += 2) {   // (one fp32 tile = 128 bytes = two bf16 tiles' worth of the row)
#pragma unroll
            for (int g = 0; g < 4; ++g) *reinterpret_cast<uint4*>(row + 32 * t + 16 * g + 8 * h) = v;
            v.x += 1;
        }
    } else {
        for (int t = 0; t < T; t += 2) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                unsigned short* row = base + (size_t)(r0 + 8 * i + (lane >> 3)) * W;
                *reinterpret_cast<uint4*>(row + 32 * t + 8 * (lane & 7)) = v;
            }
            v.x += 1;
        }
    }
}

int main(int argc, char** argv) {
    const int rows = argc > 1 ? atoi(argv[1]) : 400000, reps = argc > 2 ? atoi(argv[2]) : 20, W = 640;
    unsigned short* buf;
    hipMalloc(&buf, (size_t)rows * W * 2);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = (rows + 127) / 128;
    for (int p = 0; p < 3; ++p) {
        for (int pass = 0; pass < 2; ++pass) {
            hipEventRecord(e0);
            for (int i = 0; i < reps; ++i) {
                if (p == 0) hipLaunchKernelGGL(k_store<0>, dim3(blocks), dim3(256), 0, 0, buf, rows, W, (unsigned)i);
                else if (p == 1) hipLaunchKernelGGL(k_store<1>, dim3(blocks), dim3(256), 0, 0, buf, rows, W, (unsigned)i);
                else hipLaunchKernelGGL(k_store<2>, dim3(blocks), dim3(256), 0, 0, buf, rows, W, (unsigned)i);
            }
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            if (pass) printf("pattern %d: %.1f us per launch, %.2f TB/s (%.0f MB)\n", p, ms * 1e3 / reps, (double)rows * W * 2 / (ms * 1e-3 / reps) / 1e12, (double)rows * W * 2 / 1e6);
        }
    }
    return 0;
}
