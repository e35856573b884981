// Signed error of the six-product bf16 split (x3) accumulation on random fp32 data vs exact fp64.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ void split(float x, __bf16 (&p)[3]) {
    for (int i = 0; i < 3; ++i) { p[i] = (__bf16)x; x -= (float)p[i]; }
}
// MODE 0: one accumulator, small terms first; 1: one accumulator, big term first; 2: big term in its own accumulator
template <int MODE>
__global__ void chain(const float* A, const float* B, float* out, int n) {
    const int l = threadIdx.x, i = l & 31, h = l >> 5, K = 16 * n;
    const float* a_ = A + ((size_t)blockIdx.x * 32 + i) * K;
    const float* b_ = B + (size_t)blockIdx.x * K;
    f32x16 acc, acc2; for (int r = 0; r < 16; ++r) { acc[r] = 0.f; acc2[r] = 0.f; }
    for (int s = 0; s < n; ++s) {
        bf16x8 a[3], b[3];
        for (int e = 0; e < 8; ++e) {
            __bf16 pa[3], pb[3];
            split(a_[s * 16 + 8 * h + e], pa); split(b_[s * 16 + 8 * h + e], pb);
            for (int q = 0; q < 3; ++q) { a[q][e] = pa[q]; b[q][e] = pb[q]; }
        }
        auto mm = [&](int x, int y, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[x], b[y], c, 0, 0, 0); };
        if (MODE == 0) { acc = mm(2, 0, acc); acc = mm(0, 2, acc); acc = mm(1, 1, acc); acc = mm(1, 0, acc); acc = mm(0, 1, acc); acc = mm(0, 0, acc); }
        if (MODE == 1) { acc = mm(0, 0, acc); acc = mm(0, 1, acc); acc = mm(1, 0, acc); acc = mm(1, 1, acc); acc = mm(0, 2, acc); acc = mm(2, 0, acc); }
        if (MODE == 2) { acc2 = mm(2, 0, acc2); acc2 = mm(0, 2, acc2); acc2 = mm(1, 1, acc2); acc2 = mm(1, 0, acc2); acc2 = mm(0, 1, acc2); acc = mm(0, 0, acc); }
    }
    if ((l & 31) == 0) for (int r = 0; r < 16; ++r) out[blockIdx.x * 32 + (r & 3) + 8 * (r >> 2) + 4 * h] = acc[r] + acc2[r];
}
int main() {
    const int NB = 8192;
    for (int n : {64, 256}) {
        const int K = 16 * n;
        std::vector<float> A((size_t)NB * 32 * K), B((size_t)NB * K);
        srand(7 + n);
        auto rnd = [&]() { float u = 0; for (int q = 0; q < 6; ++q) u += rand() / (float)RAND_MAX - 0.5f; return u; };
        for (auto& v : A) v = rnd();
        for (auto& v : B) v = rnd();
        float *dA, *dB, *dO; (void)hipMalloc(&dA, A.size() * 4); (void)hipMalloc(&dB, B.size() * 4); (void)hipMalloc(&dO, NB * 32 * 4);
        (void)hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); (void)hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
        std::vector<float> o(NB * 32);
        std::vector<double> ex(NB * 32);
        double sa = 0;
        for (int blk = 0; blk < NB; ++blk) for (int i = 0; i < 32; ++i) {
            double e = 0; for (int k = 0; k < K; ++k) e += (double)A[((size_t)blk * 32 + i) * K + k] * B[(size_t)blk * K + k];
            ex[blk * 32 + i] = e; sa += fabs(e);
        }
        for (int mode = 0; mode < 3; ++mode) {
            if (mode == 0) chain<0><<<NB, 64>>>(dA, dB, dO, n); else if (mode == 1) chain<1><<<NB, 64>>>(dA, dB, dO, n); else chain<2><<<NB, 64>>>(dA, dB, dO, n);
            (void)hipMemcpy(o.data(), dO, o.size() * 4, hipMemcpyDeviceToHost);
            double se = 0, s2 = 0;
            for (int q = 0; q < NB * 32; ++q) { double d = o[q] - ex[q]; se += d; s2 += d * d; }
            printf("K=%5d mode %d: mean signed err / mean|ref| %+.2e  (noise %.1e)  rms err / mean|ref| %.2e\n", K, mode, se / sa,
                   sqrt(s2) / sa, sqrt(s2 / (NB * 32)) / (sa / (NB * 32)));
        }
        (void)hipFree(dA); (void)hipFree(dB); (void)hipFree(dO);
    }
    return 0;
}
