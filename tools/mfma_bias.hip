// Signed error of long bf16 / fp32 MFMA accumulation chains on random data, against exact fp64 sums.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <cstring>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
// A[blk][i][k], B[blk][k]; k < 16*n.  out[blk][i] = D[i][0]
__global__ void chain_bf16(const float* A, const float* B, float* out, int n) {
    const int l = threadIdx.x, i = l & 31, h = l >> 5, K = 16 * n;
    const float* a_ = A + ((size_t)blockIdx.x * 32 + i) * K;
    const float* b_ = B + (size_t)blockIdx.x * K;
    f32x16 acc; for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int s = 0; s < n; ++s) {
        bf16x8 a, b;
        for (int e = 0; e < 8; ++e) { a[e] = (__bf16)a_[s * 16 + 8 * h + e]; b[e] = (__bf16)b_[s * 16 + 8 * h + e]; }
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
    }
    // D[row][col=l&31]: rows (r&3)+8(r>>2)+4h
    if ((l & 31) == 0) for (int r = 0; r < 16; ++r) out[blockIdx.x * 32 + (r & 3) + 8 * (r >> 2) + 4 * h] = acc[r];
}
__global__ void chain_f32(const float* A, const float* B, float* out, int n) {
    const int l = threadIdx.x, i = l & 31, h = l >> 5, K = 16 * n;
    const float* a_ = A + ((size_t)blockIdx.x * 32 + i) * K;
    const float* b_ = B + (size_t)blockIdx.x * K;
    f32x16 acc; for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int s = 0; s < K / 2; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a_[2 * s + h], b_[2 * s + h], acc, 0, 0, 0);
    if ((l & 31) == 0) for (int r = 0; r < 16; ++r) out[blockIdx.x * 32 + (r & 3) + 8 * (r >> 2) + 4 * h] = acc[r];
}
static float tobf(float x) { unsigned u; memcpy(&u, &x, 4); u = (u + 0x7fff + ((u >> 16) & 1)) & 0xffff0000u; memcpy(&x, &u, 4); return x; }
int main() {
    const int NB = 512;
    for (int mode = 0; mode < 3; ++mode)
    for (int n : {16, 64, 256}) {
        const int K = 16 * n;
        std::vector<float> A((size_t)NB * 32 * K), B((size_t)NB * K);
        srand(7 + n);
        auto rnd = [&]() { float u = 0; for (int q = 0; q < 6; ++q) u += rand() / (float)RAND_MAX - 0.5f; return u; };   // ~normal
        for (auto& v : A) { v = tobf(rnd()); if (mode == 1) v = fabsf(v); if (mode == 2) v = tobf(v * expf(3 * rnd())); }
        for (auto& v : B) { v = tobf(rnd()); if (mode == 1) v = fabsf(v); if (mode == 2) v = tobf(v * expf(3 * rnd())); }
        float *dA, *dB, *dO; (void)hipMalloc(&dA, A.size() * 4); (void)hipMalloc(&dB, B.size() * 4); (void)hipMalloc(&dO, NB * 32 * 4);
        (void)hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); (void)hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
        std::vector<float> o(NB * 32);
        for (int which = 0; which < 2; ++which) {
            if (which == 0) chain_bf16<<<NB, 64>>>(dA, dB, dO, n); else chain_f32<<<NB, 64>>>(dA, dB, dO, n);
            (void)hipMemcpy(o.data(), dO, o.size() * 4, hipMemcpyDeviceToHost);
            double se = 0, sa = 0, s2 = 0;
            for (int blk = 0; blk < NB; ++blk) for (int i = 0; i < 32; ++i) {
                double ex = 0; for (int k = 0; k < K; ++k) ex += (double)A[((size_t)blk * 32 + i) * K + k] * B[(size_t)blk * K + k];
                double d = o[blk * 32 + i] - ex; se += d; sa += fabs(ex); s2 += d * d;
            }
            printf("%s K=%5d %-4s: mean signed err / mean|ref| %+.2e   rms err / mean|ref| %.2e\n",
                   mode == 0 ? "randn   " : mode == 1 ? "positive" : "wide    ", K, which ? "fp32" : "bf16", se / sa, sqrt(s2 / (NB * 32)) / (sa / (NB * 32)));
        }
        (void)hipFree(dA); (void)hipFree(dB); (void)hipFree(dO);
    }
    return 0;
}
