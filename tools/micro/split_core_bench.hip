// Micro-benchmark: the chained-product core of the fused edge kernel (weights in LDS, activations in accumulator
// registers) with native fp32 MFMAs vs the 3-way bf16 split (6 products, fp32 accumulate).  Standalone:
//   make micro && build/micro/split_core_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cmath>
#include <cstring>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

constexpr int TIN = 4, TOUT = 2;           // 128 -> 64 features per layer
constexpr int K = 32 * TIN, N = 32 * TOUT;

// ---- fp32 core: W image [k][n] fp32 in LDS ---------------------------------------------------------------------------
__device__ __forceinline__ void core_f32(const f32x16& src, f32x16* out, const float* ws, int krow0, int lane_off) {
    const float* base = ws + lane_off;
    float a[3][TOUT];
    auto fetch = [&](int r, float* dst) {
        const int krow = krow0 + (r & 3) + 8 * (r >> 2);
#pragma unroll
        for (int t = 0; t < TOUT; ++t) dst[t] = base[krow * N + 32 * t];
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

__global__ __launch_bounds__(256, 2) void k_f32(const float* w, float* y, int reps) {
    __shared__ __attribute__((aligned(16))) float ws[K * N];
    for (int i = threadIdx.x; i < K * N; i += 256) ws[i] = w[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, li = lane & 31, lh = lane >> 5;
    f32x16 h[TIN];
    for (int t = 0; t < TIN; ++t)
        for (int r = 0; r < 16; ++r) h[t][r] = 0.01f * ((lane * 7 + r * 3 + t) % 13) - 0.05f;
    for (int it = 0; it < reps; ++it) {
        f32x16 o[TOUT];
        for (int t = 0; t < TOUT; ++t)
            for (int r = 0; r < 16; ++r) o[t][r] = 0.f;
#pragma unroll
        for (int t = 0; t < TIN; ++t) core_f32(h[t], o, ws, 32 * t, 4 * lh * N + li);
        // feed back (keeps a dependence chain like the layers of the real kernel)
#pragma unroll
        for (int t = 0; t < TIN; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) h[t][r] = fminf(fmaxf(o[t % TOUT][r] * 0.25f + h[t][r] * 0.5f + 0.01f * ((r + t) % 5), -1.f), 1.f);
    }
    float s = 0.f;
    for (int t = 0; t < TIN; ++t)
        for (int r = 0; r < 16; ++r) s += h[t][r];
    y[blockIdx.x * 256 + threadIdx.x] = s;
}

// ---- split core: three bf16 images, lane-linear 16-byte operands ------------------------------------------------------
// image index ((kb * TOUT + t) * 3 + piece) * 64 + lane, 8 bf16 each; kb = 16-deep k block (2 per source tile)
struct Split8 { bf16x8 p[3]; };
__device__ __forceinline__ Split8 split8(const f32x16& s, int r0) {
    Split8 o;
#pragma unroll
    for (int i = 0; i < 8; i += 2) {
        const float x0 = s[r0 + i], x1 = s[r0 + i + 1];
        const __bf16 h0 = (__bf16)x0, h1 = (__bf16)x1;
        const float a0 = x0 - (float)h0, a1 = x1 - (float)h1;
        const __bf16 m0 = (__bf16)a0, m1 = (__bf16)a1;
        const __bf16 l0 = (__bf16)(a0 - (float)m0), l1 = (__bf16)(a1 - (float)m1);
        o.p[0][i] = h0; o.p[0][i + 1] = h1;
        o.p[1][i] = m0; o.p[1][i + 1] = m1;
        o.p[2][i] = l0; o.p[2][i + 1] = l1;
    }
    return o;
}

template <int NPROD>
__device__ __forceinline__ void core_split(const f32x16& src, f32x16* out, const bf16x8* ws, int kb0, int lane) {
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        __builtin_amdgcn_sched_barrier(0);
        const Split8 b = split8(src, 8 * c);
        bf16x8 a[TOUT][3];
#pragma unroll
        for (int t = 0; t < TOUT; ++t)
#pragma unroll
            for (int q = 0; q < 3; ++q) a[t][q] = ws[(((kb0 + c) * TOUT + t) * 3 + q) * 64 + lane];
#pragma unroll
        for (int t = 0; t < TOUT; ++t) {
            if (NPROD >= 6) {
                out[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[t][2], b.p[0], out[t], 0, 0, 0);
                out[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[t][0], b.p[2], out[t], 0, 0, 0);
                out[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[t][1], b.p[1], out[t], 0, 0, 0);
            }
            out[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[t][1], b.p[0], out[t], 0, 0, 0);
            out[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[t][0], b.p[1], out[t], 0, 0, 0);
            out[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[t][0], b.p[0], out[t], 0, 0, 0);
        }
    }
}

template <int NPROD>
__global__ __launch_bounds__(256, 2) void k_split(const bf16x8* w, float* y, int reps) {
    constexpr int NV = 2 * TIN * TOUT * 3 * 64;
    __shared__ __attribute__((aligned(16))) bf16x8 ws[NV];
    for (int i = threadIdx.x; i < NV; i += 256) ws[i] = w[i];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    f32x16 h[TIN];
    for (int t = 0; t < TIN; ++t)
        for (int r = 0; r < 16; ++r) h[t][r] = 0.01f * ((lane * 7 + r * 3 + t) % 13) - 0.05f;
    for (int it = 0; it < reps; ++it) {
        f32x16 o[TOUT];
        for (int t = 0; t < TOUT; ++t)
            for (int r = 0; r < 16; ++r) o[t][r] = 0.f;
#pragma unroll
        for (int t = 0; t < TIN; ++t) core_split<NPROD>(h[t], o, ws, 2 * t, lane);
#pragma unroll
        for (int t = 0; t < TIN; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) h[t][r] = fminf(fmaxf(o[t % TOUT][r] * 0.25f + h[t][r] * 0.5f + 0.01f * ((r + t) % 5), -1.f), 1.f);
    }
    float s = 0.f;
    for (int t = 0; t < TIN; ++t)
        for (int r = 0; r < 16; ++r) s += h[t][r];
    y[blockIdx.x * 256 + threadIdx.x] = s;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

static unsigned short bf16_rne(float f) {
    unsigned u; memcpy(&u, &f, 4);
    u += 0x7fff + ((u >> 16) & 1);
    return (unsigned short)(u >> 16);
}
static float bf16_f(unsigned short b) { unsigned u = (unsigned)b << 16; float f; memcpy(&f, &u, 4); return f; }

int main() {
    const int blocks = 2048, reps = 200;
    std::vector<float> w(K * N);
    for (int k = 0; k < K; ++k)
        for (int n = 0; n < N; ++n) w[k * N + n] = 0.05f * sinf(0.37f * k + 1.3f * n);
    // split images with the k order of the accumulator layout: element i of lane (m, g) in block kb of source tile t_in:
    // k = 32 t_in + 16 c + (i & 3) + 8 (i >> 2) + 4 g
    std::vector<unsigned short> ws((size_t)2 * TIN * TOUT * 3 * 64 * 8);
    for (int kb = 0; kb < 2 * TIN; ++kb)
        for (int t = 0; t < TOUT; ++t)
            for (int lane = 0; lane < 64; ++lane)
                for (int i = 0; i < 8; ++i) {
                    const int m = lane & 31, g = lane >> 5;
                    const int k = 16 * kb + (i & 3) + 8 * (i >> 2) + 4 * g;
                    const float x = w[k * N + 32 * t + m];
                    const unsigned short h = bf16_rne(x);
                    const float r1 = x - bf16_f(h);
                    const unsigned short mm = bf16_rne(r1);
                    const unsigned short l = bf16_rne(r1 - bf16_f(mm));
                    const size_t base = ((size_t)(kb * TOUT + t) * 3) * 64 * 8 + (size_t)lane * 8 + i;
                    ws[base] = h; ws[base + 64 * 8] = mm; ws[base + 2 * 64 * 8] = l;
                }
    float *dw, *dy0, *dy1, *dy2; void* dws;
    CK(hipMalloc(&dw, w.size() * 4)); CK(hipMalloc(&dws, ws.size() * 2));
    CK(hipMalloc(&dy0, blocks * 256 * 4)); CK(hipMalloc(&dy1, blocks * 256 * 4)); CK(hipMalloc(&dy2, blocks * 256 * 4));
    CK(hipMemcpy(dw, w.data(), w.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dws, ws.data(), ws.size() * 2, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms[3];
    for (int v = 0; v < 3; ++v) {
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(e0));
            if (v == 0) hipLaunchKernelGGL(k_f32, dim3(blocks), dim3(256), 0, 0, dw, dy0, reps);
            if (v == 1) hipLaunchKernelGGL(k_split<6>, dim3(blocks), dim3(256), 0, 0, (const bf16x8*)dws, dy1, reps);
            if (v == 2) hipLaunchKernelGGL(k_split<3>, dim3(blocks), dim3(256), 0, 0, (const bf16x8*)dws, dy2, reps);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms[v], e0, e1));
        }
    }
    std::vector<float> y0(blocks * 256), y1(blocks * 256), y2(blocks * 256);
    CK(hipMemcpy(y0.data(), dy0, y0.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(y1.data(), dy1, y1.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(y2.data(), dy2, y2.size() * 4, hipMemcpyDeviceToHost));
    double d1 = 0, d2 = 0, mx = 0;
    for (int i = 0; i < 256; ++i) { d1 = fmax(d1, fabs(y0[i] - y1[i])); d2 = fmax(d2, fabs(y0[i] - y2[i])); mx = fmax(mx, fabs(y0[i])); }
    const double flop = 2.0 * K * N * 32 * 4 * (double)blocks * reps;   // logical fp32 flops
    printf("fp32 MFMA core : %.3f ms  %.1f TFLOP/s\n", ms[0], flop / ms[0] / 1e9);
    printf("bf16x6 split   : %.3f ms  %.1f TFLOP/s (fp32-equivalent)  max|diff| %.3e of %.3e\n", ms[1], flop / ms[1] / 1e9, d1, mx);
    printf("bf16x3 split   : %.3f ms  %.1f TFLOP/s (fp32-equivalent)  max|diff| %.3e\n", ms[2], flop / ms[2] / 1e9, d2);
    return 0;
}
