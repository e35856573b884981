// How does v_mfma_f32_32x32x16_bf16 round?  c = 1 (or -1) plus 16 equal products of 2^-(23+s).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__global__ void probe(float* out, float cval, float aval, float bval, int nz) {
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = cval;
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)aval; b[e] = (__bf16)bval; }
    // only the first nz k-values of each half carry a product
    for (int e = nz; e < 8; ++e) a[e] = (__bf16)0.f;
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
    if (threadIdx.x == 0) out[0] = acc[0];
}
__global__ void probe32(float* out, float cval, float aval, float bval) {
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = cval;
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(aval, bval, acc, 0, 0, 0);
    if (threadIdx.x == 0) out[0] = acc[0];
}
int main() {
    float* d; (void)hipMalloc(&d, 4); float h;
    for (float c : {1.0f, -1.0f}) {
        for (float sgn : {1.0f, -1.0f}) {
            printf("c=%+.0f products %s: ", c, sgn > 0 ? "+" : "-");
            for (int s = 1; s <= 12; ++s) {
                // 16 products of sgn * 2^-(23+s); exact sum = sgn * 2^-(19+s)
                probe<<<1, 64>>>(d, c, sgn * ldexpf(1.f, -12), ldexpf(1.f, -(11 + s)), 8);
                (void)hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
                double exact = (double)c + sgn * ldexp(1.0, -(19 + s));
                printf("s=%d:%+.2f ", s, ((double)h - exact) / ldexp(1.0, -23));
            }
            printf(" (error in ulp(1) = 2^-23)\n");
        }
    }
    // one product only, of 0.75 ulp / 0.5 / 0.25 / ... to see guard bits per product
    for (float c : {1.0f, -1.0f}) for (float sgn : {1.0f, -1.0f}) {
        printf("c=%+.0f single product %s 1.5*2^-(23+s): ", c, sgn > 0 ? "+" : "-");
        for (int s = 0; s <= 10; ++s) {
            probe<<<1, 64>>>(d, c, sgn * 1.5f * ldexpf(1.f, -12), ldexpf(1.f, -(11 + s)), 1);
            (void)hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
            double exact = (double)c + sgn * 2 * 1.5 * ldexp(1.0, -(23 + s));   // two halves (k=0 and k=8) carry it
            printf("s=%d:%+.3f ", s, ((double)h - exact) / ldexp(1.0, -23));
        }
        printf("\n");
    }
    for (float c : {1.0f, -1.0f}) for (float sgn : {1.0f, -1.0f}) {
        printf("fp32 mfma c=%+.0f 2 products %s 1.5*2^-(23+s): ", c, sgn > 0 ? "+" : "-");
        for (int s = 0; s <= 6; ++s) {
            probe32<<<1, 64>>>(d, c, sgn * 1.5f * ldexpf(1.f, -12), ldexpf(1.f, -(11 + s)));
            (void)hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
            double exact = (double)c + sgn * 2 * 1.5 * ldexp(1.0, -(23 + s));
            printf("s=%d:%+.3f ", s, ((double)h - exact) / ldexp(1.0, -23));
        }
        printf("\n");
    }
    return 0;
}
