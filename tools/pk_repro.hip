// Standalone reproduction of the packed-fp32 fault on gfx950: a kernel whose v_pk_fma_f32 instructions take the HIGH word
// of a source pair for the LOW result (op_sel:[0,1,0]) returns wrong low results in lanes 48-63 while short-lived
// matrix-pipe blocks with barriers run beside it on another stream; the same arithmetic with v_fma_f32 never does.
//
// Victims (all inline assembly, so the compiler has no choice):
//   pk-cross  acc = x * w.hi + acc  as v_pk_fma_f32 ... op_sel:[0,1,0]      (both results use the high word of w)
//   pk-plain  acc = x * w + acc      as v_pk_fma_f32 without op_sel           (low with low, high with high)
//   pk-bcast  acc = x * w.lo + acc   as v_pk_fma_f32 ... op_sel_hi:[1,0,1]   (both results use the low word of w)
//   scalar    the pk-cross arithmetic as two v_fma_f32
//   fma_mix   v_fma_mix_f32 with an fp16 source taken from the high half-word (op_sel:[0,0,1] op_sel_hi:[0,0,1])
//   pk-mul/add  v_pk_mul_f32 / v_pk_add_f32 with op_sel:[0,1];  pk-f16  v_pk_fma_f16 op_sel:[0,1,0];  pk-mov  v_pk_mov_b32 op_sel:[1,0]
// Neighbours: see the table in main().  Every victim launch is compared bit for bit with its own standalone result.
//   hipcc --offload-arch=gfx950 -O2 -o tools/pk_repro tools/pk_repro.hip && tools/pk_repro [launches per cell]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int KIND>
__global__ __launch_bounds__(256) void victim(float* __restrict__ out, int iters) {
    const int gid = blockIdx.x * 256 + threadIdx.x;
    f32x2 acc[4], x[4], w[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        acc[i] = f32x2{0.f, 0.f};
        x[i] = f32x2{1.0f + 1e-3f * (float)((gid + 7 * i) % 101), 0.5f - 1e-3f * (float)((gid + 3 * i) % 67)};
        w[i] = f32x2{0.25f + 1e-3f * (float)((gid + i) % 31), -0.125f + 1e-3f * (float)((gid + 5 * i) % 13)};
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if constexpr (KIND == 0) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0]" : "+v"(acc[i]) : "v"(x[i]), "v"(w[i]));
            if constexpr (KIND == 1) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(x[i]), "v"(w[i]));
            if constexpr (KIND == 2) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc[i]) : "v"(x[i]), "v"(w[i]));
            if constexpr (KIND == 3) {
                float lo = acc[i].x, hi = acc[i].y;
                asm volatile("v_fma_f32 %0, %2, %4, %0\n\tv_fma_f32 %1, %3, %4, %1" : "+v"(lo), "+v"(hi) : "v"(x[i].x), "v"(x[i].y), "v"(w[i].y));
                acc[i] = f32x2{lo, hi};
            }
            // other instructions that select a half with op_sel
            if constexpr (KIND == 4) {        // fp16 source from the high half-word of a register (as the fp16 split's residuals)
                const uint32_t hh = __builtin_bit_cast(uint32_t, w[i].y);      // high half-word: sign, exponent, top mantissa of w.y
                asm volatile("v_fma_mix_f32 %0, %2, %3, %4 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
                             "v_fma_mix_f32 %1, %2, %3, %4 op_sel_hi:[0,0,1]"
                             : "+v"(acc[i].x), "+v"(acc[i].y) : "v"(x[i].x), "v"(w[i].x), "v"(hh));
            }
            if constexpr (KIND == 5) {
                asm volatile("v_pk_mul_f32 %0, %0, %1 op_sel:[0,1]" : "+v"(acc[i]) : "v"(w[i]));
                asm volatile("v_pk_add_f32 %0, %0, %1 op_sel:[0,1]" : "+v"(acc[i]) : "v"(x[i]));
            }
            if constexpr (KIND == 6) {        // packed fp16: 32-bit registers of two halves
                uint32_t a = __builtin_bit_cast(uint32_t, acc[i].x);
                const uint32_t xx = 0x38003c00u + (uint32_t)(gid & 15), ww = 0x34003800u + (uint32_t)((gid + it) & 15);
                asm volatile("v_pk_fma_f16 %0, %1, %2, %0 op_sel:[0,1,0]" : "+v"(a) : "v"(xx), "v"(ww));
                asm volatile("v_pk_mul_f16 %0, %0, %1 op_sel_hi:[1,0]" : "+v"(a) : "v"(0x38003800u));
                acc[i].x = __builtin_bit_cast(float, a);
            }
            if constexpr (KIND == 7) {
                f32x2 t;
                asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[1,0]" : "=v"(t) : "v"(x[i]), "v"(w[i]));
                acc[i] += t;
            }
            // keep the sums bounded and the operands changing
            acc[i] *= 0.5f;
            w[i] = f32x2{w[i].y, w[i].x};
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        out[(int64_t)gid * 8 + 2 * i] = acc[i].x;
        out[(int64_t)gid * 8 + 2 * i + 1] = acc[i].y;
    }
}

// Neighbour: per step TILES x (two dependent MFMAs on one accumulator, one on another) on random finite fp16 patterns;
// BARRIER: a block barrier every four steps; EPI: 0 nothing is stored unless a sum hits a magic value, 1 accumulators
// combined by fp32 FMAs and stored (16 floats per lane)
template <int TILES, bool MFMA, bool BARRIER, int EPI>
__global__ __launch_bounds__(256) void neighbour(float* __restrict__ out, int iters) {
    u32x4 ra[2], rb[2];
    uint32_t h = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + 12345u;
    for (int k = 0; k < 2; ++k)
        for (int i = 0; i < 4; ++i) {
            h = h * 1664525u + 1013904223u;
            ra[k][i] = (h & 0xbbffbbffu) | 0x20002000u;
            h = h * 1664525u + 1013904223u;
            rb[k][i] = (h & 0xbbffbbffu) | 0x20002000u;
        }
    f32x16 hi[TILES], lo[TILES];
    for (int k = 0; k < TILES; ++k) { hi[k] = f32x16{}; lo[k] = f32x16{}; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < TILES; ++k) {
            if constexpr (MFMA) {
                lo[k] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ra[1]), __builtin_bit_cast(f16x8, rb[0]), lo[k], 0, 0, 0);
                lo[k] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ra[0]), __builtin_bit_cast(f16x8, rb[1]), lo[k], 0, 0, 0);
                hi[k] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ra[0]), __builtin_bit_cast(f16x8, rb[0]), hi[k], 0, 0, 0);
            } else {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    lo[k][i] = __builtin_fmaf(lo[k][i], 0.999f, __builtin_bit_cast(float, ra[1][i & 3] & 0x3fffffffu));
                    hi[k][i] = __builtin_fmaf(hi[k][i], 0.998f, __builtin_bit_cast(float, rb[0][i & 3] & 0x3fffffffu));
                }
            }
        }
        if (BARRIER && (it & 3) == 3) __syncthreads();
    }
    if constexpr (EPI == 1) {
        float* dst = out + ((size_t)(blockIdx.x & 1023) * 256 + threadIdx.x) * 16;
        for (int i = 0; i < 16; ++i) {
            float s = 0.f;
            for (int k = 0; k < TILES; ++k) s += __builtin_fmaf(lo[k][i], 1.f / 2048.f, hi[k][i]);
            dst[i] = s;
        }
    } else {
        float s = 0.f;
        for (int k = 0; k < TILES; ++k)
            for (int i = 0; i < 16; ++i) s += hi[k][i] + lo[k][i];
        if (s == 12345.678f) out[0] = s;
    }
}

struct Nb { const char* name; void (*launch)(float*, hipStream_t); };
#define NB(T, M, B, E, blocks, iters) [](float* o, hipStream_t s) { neighbour<T, M, B, E><<<blocks, 256, 0, s>>>(o, iters); }

int main(int argc, char** argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 100;
    const int blocks = 2048, lanes = blocks * 256, iters = 400;
    float *d_out, *d_nb;
    CK(hipMalloc(&d_out, (size_t)lanes * 32));
    CK(hipMalloc(&d_nb, (size_t)1024 * 256 * 16 * 4));
    hipStream_t s1, s2;
    CK(hipStreamCreate(&s1));
    CK(hipStreamCreate(&s2));
    const Nb nbs[] = {
        {"nothing", nullptr},
        {"MFMA triples x2, barriers, stores; 60000 blocks of 24 steps", NB(2, true, true, 1, 60000, 24)},
        {"the same without barriers", NB(2, true, false, 1, 60000, 24)},
        {"the same without stores (barriers kept)", NB(2, true, true, 0, 60000, 24)},
        {"the same with fp32 FMAs in place of the MFMAs", NB(2, false, true, 1, 60000, 24)},
        {"MFMA triples x2, barriers, stores; 1024 blocks of 1500 steps", NB(2, true, true, 1, 1024, 1500)},
        {"MFMA triples x2, no barriers, no stores; 1024 blocks of 1500 steps", NB(2, true, false, 0, 1024, 1500)},
    };
    const char* vnames[] = {"pk-cross", "pk-plain", "pk-bcast", "scalar", "fma_mix", "pk-mul/add", "pk-f16", "pk-mov"};
    void (*victims[])(float*, int) = {victim<0>, victim<1>, victim<2>, victim<3>, victim<4>, victim<5>, victim<6>, victim<7>};
    std::vector<float> ref((size_t)lanes * 8), got((size_t)lanes * 8);
    printf("%d launches per cell; a cell: launches with wrong values / wrong values [lanes mod 64 hit; low / high results]\n", reps);
    for (const Nb& nb : nbs) {
        printf("beside %s\n", nb.name);
        for (int v = 0; v < 8; ++v) {
            hipLaunchKernelGGL(victims[v], dim3(blocks), dim3(256), 0, s1, d_out, iters);
            CK(hipStreamSynchronize(s1));
            CK(hipMemcpy(ref.data(), d_out, ref.size() * 4, hipMemcpyDeviceToHost));
            long long bad_runs = 0, bad_vals = 0, by_lane[64] = {0}, by_half[2] = {0};
            for (int r = 0; r < reps; ++r) {
                if (nb.launch) nb.launch(d_nb, s2);
                hipLaunchKernelGGL(victims[v], dim3(blocks), dim3(256), 0, s1, d_out, iters);
                CK(hipStreamSynchronize(s1));
                CK(hipMemcpy(got.data(), d_out, got.size() * 4, hipMemcpyDeviceToHost));
                CK(hipStreamSynchronize(s2));
                long long n = 0;
                for (size_t i = 0; i < got.size(); ++i)
                    if (memcmp(&got[i], &ref[i], 4)) { ++n; ++by_lane[(i / 8) % 64]; ++by_half[i & 1]; }
                if (n) { ++bad_runs; bad_vals += n; }
            }
            printf("    %-9s %4lld / %-9lld", vnames[v], bad_runs, bad_vals);
            if (bad_vals) {
                int lo = 64, hi = -1;
                for (int l = 0; l < 64; ++l) if (by_lane[l]) { lo = l < lo ? l : lo; hi = l; }
                printf(" [lanes %d-%d; low %lld / high %lld]", lo, hi, by_half[0], by_half[1]);
            }
            printf("\n");
        }
    }
    return 0;
}
