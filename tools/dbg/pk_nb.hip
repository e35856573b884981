// Synthetic neighbour kernels for tools/dbg/pk_neighbours2.py: each keeps one kind of unit busy on every CU.
//   hipcc --offload-arch=gfx950 -O2 -shared -fPIC -o tools/_pk/libpk_nb.so tools/dbg/pk_nb.hip
#include <hip/hip_runtime.h>
#include <cstdint>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// 0: matrix pipe only, operands in registers
__global__ __launch_bounds__(256) void nb_mfma(float* sink, int iters) {
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * (threadIdx.x + i)); b[i] = (_Float16)(0.002f * (threadIdx.x - i)); }
    f32x16 c0 = {}, c1 = {};
    for (int it = 0; it < iters; ++it) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, c1, 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += c0[i] + c1[i];
    if (s == 12345.678f) sink[0] = s;
}
// 1: matrix pipe fed from LDS by 16-byte reads (LDS bytes: template), one barrier per step
template <int LDS_FLOATS>
__global__ __launch_bounds__(256) void nb_mfma_lds(float* sink, int iters) {
    __shared__ __attribute__((aligned(16))) float buf[LDS_FLOATS];
    for (int i = threadIdx.x; i < LDS_FLOATS; i += 256) buf[i] = 1e-3f * (float)(i & 1023);
    __syncthreads();
    f32x16 c0 = {}, c1 = {};
    int p = threadIdx.x * 4;
    for (int it = 0; it < iters; ++it) {
        const f32x4 u = *reinterpret_cast<const f32x4*>(&buf[p]);
        const f32x4 v = *reinterpret_cast<const f32x4*>(&buf[(p + 2048) % LDS_FLOATS]);
        f16x8 a, b;
        for (int i = 0; i < 4; ++i) { a[i] = (_Float16)u[i]; a[4 + i] = (_Float16)v[i]; b[i] = (_Float16)v[i]; b[4 + i] = (_Float16)u[i]; }
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, c1, 0, 0, 0);
        p = (p + 1028) % (LDS_FLOATS - 4);
        p &= ~3;
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += c0[i] + c1[i];
    if (s == 12345.678f) sink[0] = s;
}
// 2: LDS traffic only (16-byte reads and writes)
__global__ __launch_bounds__(256) void nb_lds(float* sink, int iters) {
    __shared__ __attribute__((aligned(16))) float buf[8192];
    for (int i = threadIdx.x; i < 8192; i += 256) buf[i] = (float)i;
    __syncthreads();
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    int p = threadIdx.x * 4;
    for (int it = 0; it < iters; ++it) {
        s += *reinterpret_cast<f32x4*>(&buf[p]);
        p = (p * 5 + 12) & 8188;
        *reinterpret_cast<f32x4*>(&buf[(p + 4096) & 8188]) = s * 1e-6f;
    }
    if (s[0] == 12345.678f) sink[0] = s[0];
}
// 3: matrix pipe fed by 16-byte global loads from a cache-resident buffer
__global__ __launch_bounds__(256) void nb_mfma_vmem(const float* __restrict__ src, float* sink, int iters) {
    f32x16 c0 = {}, c1 = {};
    int64_t p = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4 % (1 << 20);
    for (int it = 0; it < iters; ++it) {
        const f32x4 u = *reinterpret_cast<const f32x4*>(src + p);
        const f32x4 v = *reinterpret_cast<const f32x4*>(src + (p + 65536) % (1 << 20));
        f16x8 a, b;
        for (int i = 0; i < 4; ++i) { a[i] = (_Float16)u[i]; a[4 + i] = (_Float16)v[i]; b[i] = (_Float16)v[i]; b[4 + i] = (_Float16)u[i]; }
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, c1, 0, 0, 0);
        p = (p + 4100 * 4) % (1 << 20);
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += c0[i] + c1[i];
    if (s == 12345.678f) sink[0] = s;
}
// 4: many accumulators: the wave's registers spill over into the accumulation half of the register file
__global__ __launch_bounds__(256) void nb_mfma_wide(float* sink, int iters) {
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * (threadIdx.x + i)); b[i] = (_Float16)(0.002f * (threadIdx.x - i)); }
    f32x16 c[12];
    for (int k = 0; k < 12; ++k) c[k] = f32x16{};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 12; ++k) c[k] = __builtin_amdgcn_mfma_f32_32x32x16_f16(k & 1 ? a : b, k & 2 ? a : b, c[k], 0, 0, 0);
    }
    float s = 0.f;
    for (int k = 0; k < 12; ++k)
        for (int i = 0; i < 16; ++i) s += c[k][i];
    if (s == 12345.678f) sink[0] = s;
}
// 5: vector ALU only (fp32 fused multiply-adds, four chains)
__global__ __launch_bounds__(256) void nb_valu(float* sink, int iters) {
    float x0 = threadIdx.x * 1e-3f, x1 = x0 + 1.f, x2 = x0 - 1.f, x3 = x0 * 0.5f;
    for (int it = 0; it < iters; ++it) {
        x0 = __builtin_fmaf(x0, 0.999f, 0.01f); x1 = __builtin_fmaf(x1, 0.998f, 0.02f);
        x2 = __builtin_fmaf(x2, 0.997f, 0.03f); x3 = __builtin_fmaf(x3, 0.996f, 0.04f);
    }
    if (x0 + x1 + x2 + x3 == 12345.678f) sink[0] = x0;
}

// 7: sub-dword-addressed conversions (SDWA: the high half-word of a register as the fp16 source), as the fp16 split's
// unpacking of 16-bit pairs compiles to
__global__ __launch_bounds__(256) void nb_sdwa(float* sink, int iters) {
    uint32_t h = 0x3c003800u + threadIdx.x;       // two fp16 values
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
        float a, b;
        asm volatile("v_cvt_f32_f16_sdwa %0, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n\t"
                     "v_cvt_f32_f16_e32 %1, %2\n\t" : "=v"(a), "=v"(b) : "v"(h));
        acc += a * 1e-6f + b * 1e-7f;
        h += 1;
    }
    if (acc == 12345.678f) sink[0] = acc;
}
// 8: packed conversions fp32 pair -> fp16 pair / bf16 pair
__global__ __launch_bounds__(256) void nb_cvtpk(float* sink, int iters) {
    float x = 1e-3f * threadIdx.x, y = 2.f - x;
    uint32_t acc = 0;
    for (int it = 0; it < iters; ++it) {
        uint32_t p, q;
        asm volatile("v_cvt_pk_f16_f32 %0, %2, %3\n\t"
                     "v_cvt_pk_bf16_f32 %1, %2, %3\n\t" : "=v"(p), "=v"(q) : "v"(x), "v"(y));
        acc ^= p + q;
        x += 1e-4f;
        y -= 1e-4f;
    }
    if (acc == 0x12345678u) sink[0] = 1.f;
}
// 9: private (scratch) memory traffic: a per-lane array indexed at run time
__global__ __launch_bounds__(256) void nb_scratch(float* sink, int iters, int stride) {
    float arr[64];
    for (int i = 0; i < 64; ++i) arr[i] = (float)(i + threadIdx.x);
    float acc = 0.f;
    int p = threadIdx.x & 63;
    for (int it = 0; it < iters; ++it) {
        acc += arr[p];
        arr[(p + 7) & 63] = acc * 1e-6f;
        p = (p + stride) & 63;
    }
    if (acc == 12345.678f) sink[0] = acc;
}
// 10: SDWA compares (as the split kernels' edge tests compile to)
__global__ __launch_bounds__(256) void nb_sdwa_cmp(float* sink, int iters) {
    uint32_t a = threadIdx.x * 0x10001u, n = 0;
    for (int it = 0; it < iters; ++it) {
        uint64_t m;
        asm volatile("v_cmp_gt_u32_sdwa %0, %1, %2 src0_sel:WORD_1 src1_sel:DWORD" : "=s"(m) : "v"(a), "v"(n));
        n += (uint32_t)__builtin_popcountll(m);
        a += 0x10000u;
    }
    if (n == 0x12345678u) sink[0] = 1.f;
}

// 11: the matrix pipe at full rate on operands whose bits change with every instruction (random finite fp16 patterns,
// eight independent accumulators): as much switching activity as a register-only loop can make
__global__ __launch_bounds__(256) void nb_mfma_random(float* sink, int iters) {
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    u32x4 ra[4], rb[4];
    uint32_t h = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + 12345u;
    for (int k = 0; k < 4; ++k)
        for (int i = 0; i < 4; ++i) {
            h = h * 1664525u + 1013904223u;
            ra[k][i] = (h & 0xbbffbbffu) | 0x20002000u;       // fp16 pairs, exponents 8 .. 14: finite, around 1e-2 .. 1
            h = h * 1664525u + 1013904223u;
            rb[k][i] = (h & 0xbbffbbffu) | 0x20002000u;
        }
    f32x16 c[8];
    for (int k = 0; k < 8; ++k) c[k] = f32x16{};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 8; ++k)
            c[k] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ra[(k + it) & 3]), __builtin_bit_cast(f16x8, rb[(k * 3 + it) & 3]),
                                                         c[k], 0, 0, 0);
    }
    float s = 0.f;
    for (int k = 0; k < 8; ++k)
        for (int i = 0; i < 16; ++i) s += c[k][i];
    if (s == 12345.678f) sink[0] = s;
}

// 12: the split kernels' MFMA pattern: per tile two dependent MFMAs back to back on the correction accumulator, then one
// on the leading accumulator (operands: random finite fp16 patterns in registers); TILES tiles per step
template <int TILES>
__global__ __launch_bounds__(256) void nb_mfma_x3(float* sink, int iters) {
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
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
            lo[k] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ra[1]), __builtin_bit_cast(f16x8, rb[0]), lo[k], 0, 0, 0);
            lo[k] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ra[0]), __builtin_bit_cast(f16x8, rb[1]), lo[k], 0, 0, 0);
            hi[k] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ra[0]), __builtin_bit_cast(f16x8, rb[0]), hi[k], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int k = 0; k < TILES; ++k)
        for (int i = 0; i < 16; ++i) s += hi[k][i] + lo[k][i];
    if (s == 12345.678f) sink[0] = s;
}

// 15: as 12 with a block barrier every four steps (the split kernels have one per K chunk) and a prologue / epilogue
// like theirs: accumulators combined by fp32 FMAs and stored
template <int TILES>
__global__ __launch_bounds__(256) void nb_mfma_x3_barrier(float* __restrict__ out, int iters) {
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
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
            lo[k] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ra[1]), __builtin_bit_cast(f16x8, rb[0]), lo[k], 0, 0, 0);
            lo[k] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ra[0]), __builtin_bit_cast(f16x8, rb[1]), lo[k], 0, 0, 0);
            hi[k] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ra[0]), __builtin_bit_cast(f16x8, rb[0]), hi[k], 0, 0, 0);
        }
        if ((it & 3) == 3) __syncthreads();
    }
    float* dst = out + ((size_t)(blockIdx.x & 1023) * 256 + threadIdx.x) * 16;
    for (int i = 0; i < 16; ++i) {
        float s = 0.f;
        for (int k = 0; k < TILES; ++k) s += __builtin_fmaf(lo[k][i], 1.f / 2048.f, hi[k][i]);
        dst[i] = s;
    }
}

extern "C" int pk_nb_launch(int kind, const float* src, float* sink, int blocks, int iters, hipStream_t st) {
    switch (kind) {
    case 0: nb_mfma<<<blocks, 256, 0, st>>>(sink, iters); break;
    case 1: nb_mfma_lds<8192><<<blocks, 256, 0, st>>>(sink, iters); break;
    case 2: nb_lds<<<blocks, 256, 0, st>>>(sink, iters); break;
    case 3: nb_mfma_vmem<<<blocks, 256, 0, st>>>(src, sink, iters); break;
    case 4: nb_mfma_wide<<<blocks, 256, 0, st>>>(sink, iters); break;
    case 5: nb_valu<<<blocks, 256, 0, st>>>(sink, iters); break;
    case 6: nb_mfma_lds<16000><<<blocks, 256, 0, st>>>(sink, iters); break;     // 64 KB of LDS per block: two blocks per CU
    case 7: nb_sdwa<<<blocks, 256, 0, st>>>(sink, iters); break;
    case 8: nb_cvtpk<<<blocks, 256, 0, st>>>(sink, iters); break;
    case 9: nb_scratch<<<blocks, 256, 0, st>>>(sink, iters, 5); break;
    case 10: nb_sdwa_cmp<<<blocks, 256, 0, st>>>(sink, iters); break;
    case 11: nb_mfma_random<<<blocks, 256, 0, st>>>(sink, iters); break;
    case 12: nb_mfma_x3<1><<<blocks, 256, 0, st>>>(sink, iters); break;
    case 13: nb_mfma_x3<2><<<blocks, 256, 0, st>>>(sink, iters); break;
    case 14: nb_mfma_x3<4><<<blocks, 256, 0, st>>>(sink, iters); break;
    case 15: nb_mfma_x3_barrier<2><<<blocks, 256, 0, st>>>(const_cast<float*>(src), iters); break;   // src: >= 4 M floats
    case 16: nb_mfma_x3_barrier<4><<<blocks, 256, 0, st>>>(const_cast<float*>(src), iters); break;
    default: return 1;
    }
    return (int)hipGetLastError();
}
