// Is the fp32-from-three-bf16-pieces product (six v_mfma_f32_32x32x16_bf16, edge_chain.hip mfma6) BIASED?  D = A B for random
// A [32 x K], B [K x 32] with the six-product scheme and with the fp32 MFMA, against float64 on the host: rms and MEAN of the
// error in units of the result's rms (a mean far from 0 +- 1/sqrt(n) is a bias).  Also with B negated and the result negated back.
// build: make micro   (-> build/micro/mfma_bias)
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
struct P3 { bf16x8 p[3]; };
__device__ P3 split8(const float* x) {
    P3 o;
    for (int i = 0; i < 8; ++i) {
        const __bf16 h = (__bf16)x[i];
        const float r = x[i] - (float)h;
        const __bf16 m = (__bf16)r;
        o.p[0][i] = h; o.p[1][i] = m; o.p[2][i] = (__bf16)(r - (float)m);
    }
    return o;
}
// tiles: blockIdx.x = tile; A [T][32][K] row-major, B [T][K][32]; D [T][32][32].  mode 0: six bf16 products, 1: fp32 MFMA,
// 2: six products on -B, result negated, 3: nine products
__global__ void k(const float* A, const float* B, float* D, int K, int mode) {
    const int lane = threadIdx.x, i = lane & 31, g = lane >> 5;
    const float* a = A + (size_t)blockIdx.x * 32 * K;
    const float* b = B + (size_t)blockIdx.x * K * 32;
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    if (mode == 1) {
        for (int k2 = 0; k2 < K; k2 += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i * K + k2 + g], b[(k2 + g) * 32 + i], acc, 0, 0, 0);
    } else {
        const float sg = mode == 2 ? -1.f : 1.f;
        for (int k0 = 0; k0 < K; k0 += 16) {
            float av[8], bv[8];
            for (int e = 0; e < 8; ++e) { av[e] = a[i * K + k0 + 8 * g + e]; bv[e] = sg * b[(k0 + 8 * g + e) * 32 + i]; }
            const P3 x = split8(av), y = split8(bv);
            if (mode == 3) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x.p[2], y.p[2], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x.p[2], y.p[1], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x.p[1], y.p[2], acc, 0, 0, 0);
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x.p[2], y.p[0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x.p[0], y.p[2], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x.p[1], y.p[1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x.p[1], y.p[0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x.p[0], y.p[1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x.p[0], y.p[0], acc, 0, 0, 0);
        }
        if (mode == 2) for (int r = 0; r < 16; ++r) acc[r] = -acc[r];
    }
    float* d = D + (size_t)blockIdx.x * 1024;
    for (int r = 0; r < 16; ++r) d[((r & 3) + 8 * (r >> 2) + 4 * g) * 32 + i] = acc[r];
}
int main() {
    const int T = 256;
    for (int K : {32, 64, 128, 320}) {
        std::vector<float> A((size_t)T * 32 * K), B((size_t)T * K * 32), D((size_t)T * 1024);
        srand(1234 + K);
        auto rnd = [] { float s = 0; for (int i = 0; i < 6; ++i) s += (float)rand() / RAND_MAX - 0.5f; return s; };
        for (auto& v : A) v = rnd();
        for (auto& v : B) v = rnd() * ((rand() & 3) ? 1.f : 0.f);   // a quarter of the entries zero, as behind a ReLU mask
        std::vector<double> R((size_t)T * 1024);
        for (int t = 0; t < T; ++t)
            for (int i = 0; i < 32; ++i)
                for (int j = 0; j < 32; ++j) {
                    double s = 0;
                    for (int k = 0; k < K; ++k) s += (double)A[((size_t)t * 32 + i) * K + k] * B[((size_t)t * K + k) * 32 + j];
                    R[(size_t)t * 1024 + i * 32 + j] = s;
                }
        float *dA, *dB, *dD;
        hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dD, D.size() * 4);
        hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
        double rr = 0; for (double v : R) rr += v * v; rr = sqrt(rr / R.size());
        const char* names[4] = {"six bf16 products", "fp32 MFMA", "six products, -B, negated", "nine bf16 products"};
        for (int mode = 0; mode < 4; ++mode) {
            hipLaunchKernelGGL(k, dim3(T), dim3(64), 0, 0, dA, dB, dD, K, mode);
            hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
            double se = 0, me = 0;
            for (size_t q = 0; q < R.size(); ++q) { const double e = D[q] - R[q]; se += e * e; me += e; }
            const double rms = sqrt(se / R.size()), mean = me / R.size();
            printf("K %3d  %-28s rms err %.3e  mean err %+.3e  (of rms result)   mean/rms %+.4f   (noise level +-%.4f)\n", K, names[mode],
                   rms / rr, mean / rr, mean / rms, 1.0 / sqrt((double)R.size()));
        }
        hipFree(dA); hipFree(dB); hipFree(dD);
    }
    return 0;
}
