// MFMA issue-rate probe: what fp32 32x32x2 rate does the chip sustain with no memory traffic,
// (a) with constant operands, (b) with random operands (data-dependent power)?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int CH, bool RND>
__global__ __launch_bounds__(256) void probe(float* out, const float* in, int iters) {
    f32x16 acc[CH];
    for (int c = 0; c < CH; ++c) for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    float a[8], b[8];
    for (int k = 0; k < 8; ++k) {
        a[k] = RND ? in[(threadIdx.x * 8 + k) & 4095] : 1e-3f;
        b[k] = RND ? in[(threadIdx.x * 8 + k + 2048) & 4095] : 1e-3f;
    }
    for (int i = 0; i < iters; i += 8) {
#pragma unroll
        for (int k = 0; k < 8; ++k)
#pragma unroll
            for (int c = 0; c < CH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[k], b[(k + c) & 7], acc[c], 0, 0, 0);
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
    double fl = 4096.0 * CH * iters * 4.0 * blocks;
    printf("%s chains %d blocks %5d iters %6d: %8.1f us  %7.1f TF\n", RND ? "random  " : "constant", CH, blocks, iters, ms * 1e3, fl / ms / 1e9);
    (void)hipFree(out);
}
int main() {
    float h[4096]; srand(1);
    for (int i = 0; i < 4096; ++i) h[i] = (rand() / (float)RAND_MAX - 0.5f) * 2.f;
    float* in; (void)hipMalloc(&in, sizeof(h)); (void)hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
    for (int rep = 0; rep < 2; ++rep) {
        run<2, false>(768, 4096, in); run<2, true>(768, 4096, in);
        run<1, false>(512, 8192, in); run<1, true>(512, 8192, in);
        run<4, true>(768, 512, in); run<4, true>(768, 16384, in);
    }
    return 0;
}
