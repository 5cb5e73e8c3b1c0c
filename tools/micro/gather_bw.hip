// Micro-benchmark: what the memory system gives the bf16 chain kernel's gather pattern (edge_chain_bf16.hip: per wave tile, 32
// edges x one 128-byte piece of a random row of a [N, 2176] fp32 table, 34 pieces per edge), as a function of the pieces a wave
// keeps in flight.  Build: make micro; run: build/micro/gather_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int DEPTH>
__global__ __launch_bounds__(512, 1) void k_gather(const float* __restrict__ P, const int* __restrict__ col, int E, int pw, int tiles,
                                                   float* __restrict__ out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lj = lane & 31, lh = lane >> 5;
    int e = blockIdx.x * 256 + wave * 32 + lj;
    e = e < E ? e : E - 1;
    const float* row = P + (size_t)col[e] * pw + 4 * lh;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int t = 0; t < tiles; t += DEPTH) {
        float4 v[DEPTH][4];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d)
#pragma unroll
            for (int g = 0; g < 4; ++g) v[d][g] = *reinterpret_cast<const float4*>(row + 32 * (t + d) + 8 * g);
#pragma unroll
        for (int d = 0; d < DEPTH; ++d)
#pragma unroll
            for (int g = 0; g < 4; ++g) { acc.x += v[d][g].x; acc.y += v[d][g].y; acc.z += v[d][g].z; acc.w += v[d][g].w; }
    }
    if (acc.x == 12345.678f) out[threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
}

template <int DEPTH>
static float run(const float* P, const int* col, int E, int pw, int tiles, float* out, int iters) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    const int blocks = (E + 255) / 256;
    hipLaunchKernelGGL(k_gather<DEPTH>, dim3(blocks), dim3(512), 0, 0, P, col, E, pw, tiles, out);
    hipEventRecord(a, 0);
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(k_gather<DEPTH>, dim3(blocks), dim3(512), 0, 0, P, col, E, pw, tiles, out);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms = 0.f;
    hipEventElapsedTime(&ms, a, b);
    return ms * 1e3f / iters;
}

int main(int argc, char** argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 20000, E = argc > 2 ? atoi(argv[2]) : 400000, pw = 2176, tiles = 32;
    float* P; int* col; float* out;
    hipMalloc(&P, (size_t)N * pw * 4); hipMalloc(&col, (size_t)E * 4); hipMalloc(&out, 4096);
    hipMemset(P, 0, (size_t)N * pw * 4);
    std::vector<int> h(E);
    unsigned long long s = 88172645463325252ull;
    for (int i = 0; i < E; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[i] = (int)(s % (unsigned long long)N); }
    hipMemcpy(col, h.data(), (size_t)E * 4, hipMemcpyHostToDevice);
    const double bytes = (double)E * tiles * 128.0;
    printf("table %.0f MB, %d edges x %d pieces of 128 B = %.2f GB per launch (8 waves per CU)\n", (double)N * pw * 4 / 1e6, E, tiles, bytes / 1e9);
    float us;
    us = run<1>(P, col, E, pw, tiles, out, 10); printf("pieces in flight per wave 1: %8.1f us  %.2f TB/s\n", us, bytes / us / 1e6);
    us = run<2>(P, col, E, pw, tiles, out, 10); printf("pieces in flight per wave 2: %8.1f us  %.2f TB/s\n", us, bytes / us / 1e6);
    us = run<4>(P, col, E, pw, tiles, out, 10); printf("pieces in flight per wave 4: %8.1f us  %.2f TB/s\n", us, bytes / us / 1e6);
    us = run<8>(P, col, E, pw, tiles, out, 10); printf("pieces in flight per wave 8: %8.1f us  %.2f TB/s\n", us, bytes / us / 1e6);
    return 0;
}
