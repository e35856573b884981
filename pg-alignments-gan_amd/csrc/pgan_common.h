// Shared device helpers for the gfx950 kernels of the DCWGAN-GP step.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define PGAN_OK 0
#define PGAN_EINVAL (-1)   // unsupported shape / bad argument
#define PGAN_ELAUNCH (-2)  // HIP reported a launch error
#define PGAN_ELIMIT (-3)   // size beyond the fixed maximum of an entry point that allocates nothing

#define PGAN_SLOPE 0.1f
#define PGAN_EPS 1e-5f

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

static inline int pgan_launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? PGAN_OK : PGAN_ELAUNCH;
}

__device__ __forceinline__ float lrelu(float y) { return y > 0.f ? y : PGAN_SLOPE * y; }
// derivative of LeakyReLU expressed through its OUTPUT (sign is preserved; 0 takes the slope,
// as torch's leaky_relu backward does)
__device__ __forceinline__ float lrelu_mask(float h) { return h > 0.f ? 1.f : PGAN_SLOPE; }

// Lane exchange inside a row of 16 lanes by data-parallel-primitive moves (vector-ALU rate, no LDS traffic):
// 0xB1 / 0x4E swap neighbours / pairs inside a quad, 0x141 / 0x140 mirror 8 / 16 lanes.
template <int CTRL>
__device__ __forceinline__ float dpp_move(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
template <int CTRL>
__device__ __forceinline__ double dpp_move(double v) {
    const uint64_t u = __builtin_bit_cast(uint64_t, v);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)u, CTRL, 0xf, 0xf, true);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(u >> 32), CTRL, 0xf, 0xf, true);
    return __builtin_bit_cast(double, ((uint64_t)hi << 32) | lo);
}
// Sum over the 64 lanes of a wave, in every lane: a butterfly over lane distances 1, 2, 4, 8 (after the quad steps all
// lanes of a quad hold the same value, so the mirrors pair each lane with its partner's sum), then 16 and 32 through
// the LDS crossbar.  The order is fixed: the result does not depend on timing.
template <class T>
__device__ __forceinline__ T wave_sum_t(T v) {
    v += dpp_move<0xB1>(v);
    v += dpp_move<0x4E>(v);
    v += dpp_move<0x141>(v);
    v += dpp_move<0x140>(v);
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}
__device__ __forceinline__ double wave_sum(double v) { return wave_sum_t(v); }
__device__ __forceinline__ float wave_sum(float v) { return wave_sum_t(v); }

// Block-wide sums of NV doubles; every thread gets the totals.  `sh` needs NV * (blockDim/64) doubles.
template <int NV>
__device__ __forceinline__ void block_sum(double (&v)[NV], double* sh) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = wave_sum(v[i]);
    __syncthreads();  // protect sh from a previous use
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < NV; ++i) sh[i * nw + wave] = v[i];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        double s = 0.0;
        for (int w = 0; w < nw; ++w) s += sh[i * nw + w];
        v[i] = s;
    }
}
