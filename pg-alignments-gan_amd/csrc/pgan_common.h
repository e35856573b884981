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

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

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
