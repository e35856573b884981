// bf16 32x32x16 MFMA issue-rate probe with constant / random operands.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <int CH, bool RND>
__global__ __launch_bounds__(256) void probe(float* out, const float* in, int iters) {
    f32x16 acc[CH];
    for (int c = 0; c < CH; ++c) for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    bf16x8 a[4], b[4];
    for (int k = 0; k < 4; ++k)
        for (int e = 0; e < 8; ++e) {
            a[k][e] = (__bf16)(RND ? in[(threadIdx.x * 32 + k * 8 + e) & 4095] : 1e-3f);
            b[k][e] = (__bf16)(RND ? in[(threadIdx.x * 32 + k * 8 + e + 2048) & 4095] : 1e-3f);
        }
    for (int i = 0; i < iters; i += 4) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int c = 0; c < CH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[k], b[(k + c) & 3], acc[c], 0, 0, 0);
    }
    float s = 0.f;
    for (int c = 0; c < CH; ++c) for (int r = 0; r < 16; ++r) s += acc[c][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int CH, bool RND>
void run(int blocks, int iters, const float* in) {
    float* out; (void)hipMalloc(&out, sizeof(float) * blocks * 256);
    hipEvent_t s, e; (void)hipEventCreate(&s); (void)hipEventCreate(&e);
    probe<CH, RND><<<blocks, 256>>>(out, in, iters);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(s);
    for (int r = 0; r < 20; ++r) probe<CH, RND><<<blocks, 256>>>(out, in, iters);
    (void)hipEventRecord(e); (void)hipEventSynchronize(e);
    float ms; (void)hipEventElapsedTime(&ms, s, e); ms /= 20;
    double fl = 32768.0 * CH * iters * 4.0 * blocks;
    printf("bf16 %s chains %d blocks %5d iters %6d: %8.1f us  %7.1f TF\n", RND ? "random  " : "constant", CH, blocks, iters, ms * 1e3, fl / ms / 1e9);
    (void)hipFree(out);
}
int main() {
    float h[4096]; srand(1);
    for (int i = 0; i < 4096; ++i) h[i] = (rand() / (float)RAND_MAX - 0.5f) * 2.f;
    float* in; (void)hipMalloc(&in, sizeof(h)); (void)hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
    for (int rep = 0; rep < 2; ++rep) {
        run<2, false>(768, 8192, in); run<2, true>(768, 8192, in);
        run<4, false>(512, 8192, in); run<4, true>(512, 8192, in);
        run<1, true>(1024, 8192, in);
    }
    return 0;
}
