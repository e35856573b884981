// Small HBM-bound kernels of the DCWGAN-GP step: linear head, penalty terms, output-activation
// derivative, input normalisation, loss scalars, fused Adam, Philox random numbers.
// Reference lines: head src/layers.py:482,491-493; gp penalty src/layers.py:37-54; div penalty
// src/layers.py:61-77; losses src/train_wgan_v2.py:584-597,629; Adam src/train_wgan_v2.py:489-490.
#include "pgan_common.h"

namespace {

// score[n] = <w[0:K2], h4[n]> + <w[K2:K2+K1], f1[n]> + b ; one block per sample
__global__ __launch_bounds__(256) void head_fwd_kernel(const float* __restrict__ h4, const float* __restrict__ f1,
                                                       const float* __restrict__ w, const float* __restrict__ b,
                                                       float* __restrict__ score, int K2, int K1) {
    __shared__ double sh[4];
    const int n = blockIdx.x, tid = threadIdx.x;
    float s = 0.f;
    const f32x4* x4 = reinterpret_cast<const f32x4*>(h4 + (int64_t)n * K2);
    const f32x4* w4 = reinterpret_cast<const f32x4*>(w);
    for (int i = tid; i < K2 / 4; i += 256) {
        const f32x4 x = x4[i], ww = w4[i];
        s += (x[0] * ww[0] + x[1] * ww[1]) + (x[2] * ww[2] + x[3] * ww[3]);
    }
    const float* f = f1 + (int64_t)n * K1;
    for (int i = tid; i < K1; i += 256) s += f[i] * w[K2 + i];
    double red[1] = {s};
    block_sum<1>(red, sh);
    if (tid == 0) score[n] = (float)red[0] + b[0];
}

__global__ void head_seed_kernel(const float* __restrict__ w, const float* __restrict__ coef, float* __restrict__ u4,
                                 int N, int K) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)N * K / 4) return;
    const int n = (int)(i / (K / 4)), k4 = (int)(i % (K / 4));
    const float c = coef[n];
    const f32x4 ww = reinterpret_cast<const f32x4*>(w)[k4];
    reinterpret_cast<f32x4*>(u4)[i] = ww * c;
}

__global__ void interpolate_kernel(const float* __restrict__ t, const float* __restrict__ real,
                                   const float* __restrict__ fake, float* __restrict__ out, int B, int M) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)B * M / 4) return;
    const float tt = t[(int)(i / (M / 4))];
    const f32x4 r = reinterpret_cast<const f32x4*>(real)[i], f = reinterpret_cast<const f32x4*>(fake)[i];
    reinterpret_cast<f32x4*>(out)[i] = r * tt + f * (1.f - tt);
}

// gp: v = (2 lam / n) (|g| - 1) sign(g); partial[block] = sum (|g|-1)^2
__global__ __launch_bounds__(256) void gp_penalty_kernel(const float* __restrict__ g, float* __restrict__ v,
                                                         float* __restrict__ partial, int64_t n, float scale) {
    __shared__ double sh[4];
    float s = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n / 4; i += (int64_t)gridDim.x * 256) {
        const f32x4 x = reinterpret_cast<const f32x4*>(g)[i];
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float a = fabsf(x[e]) - 1.f;
            s += a * a;
            const float sg = (x[e] > 0.f) ? 1.f : (x[e] < 0.f ? -1.f : 0.f);
            o[e] = scale * a * sg;
        }
        reinterpret_cast<f32x4*>(v)[i] = o;
    }
    double red[1] = {s};
    block_sum<1>(red, sh);
    if (threadIdx.x == 0) partial[blockIdx.x] = (float)red[0];
}

// div: one block per sample; s = sum g^2; v = (6/nb) s^2 g; partial[n] = s^3
__global__ __launch_bounds__(256) void div_penalty_kernel(const float* __restrict__ g, float* __restrict__ v,
                                                          float* __restrict__ partial, int M, float inv_nb) {
    __shared__ double sh[4];
    const int n = blockIdx.x, tid = threadIdx.x;
    const f32x4* g4 = reinterpret_cast<const f32x4*>(g + (int64_t)n * M);
    f32x4* v4 = reinterpret_cast<f32x4*>(v + (int64_t)n * M);
    float s = 0.f;
    for (int i = tid; i < M / 4; i += 256) {
        const f32x4 x = g4[i];
        s += (x[0] * x[0] + x[1] * x[1]) + (x[2] * x[2] + x[3] * x[3]);
    }
    double red[1] = {s};
    block_sum<1>(red, sh);
    const float ss = (float)red[0];
    const float k = 6.f * inv_nb * ss * ss;
    for (int i = tid; i < M / 4; i += 256) v4[i] = g4[i] * k;
    if (tid == 0) partial[n] = ss * ss * ss;
}

// out[0] = scale * sum(partial[0:n])
__global__ __launch_bounds__(256) void sum_scale_kernel(const float* __restrict__ partial, float* __restrict__ out, int n,
                                                        float scale) {
    __shared__ double sh[4];
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) s += partial[i];
    double red[1] = {s};
    block_sum<1>(red, sh);
    if (threadIdx.x == 0) out[0] = (float)(red[0] * scale);
}

// out = [W, mean real, mean fake, pen], W = mean(score[nb:2nb]) - mean(score[0:nb]) + pen
__global__ __launch_bounds__(256) void critic_loss_kernel(const float* __restrict__ score, const float* __restrict__ pen,
                                                          float* __restrict__ out, int nb) {
    __shared__ double sh[8];
    double sr = 0.0, sf = 0.0;
    for (int i = threadIdx.x; i < nb; i += 256) {
        sr += score[i];
        sf += score[nb + i];
    }
    double red[2] = {sr, sf};
    block_sum<2>(red, sh);
    if (threadIdx.x == 0) {
        const float mr = (float)(red[0] / nb), mf = (float)(red[1] / nb);
        out[0] = mf - mr + pen[0];
        out[1] = mr;
        out[2] = mf;
        out[3] = pen[0];
    }
}

__global__ void act_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y, float* __restrict__ dx,
                               int64_t n, int act) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float yy = y[i];
    dx[i] = dy[i] * (act == 1 ? yy * (1.f - yy) : (1.f - yy * yy));
}

__global__ void affine_kernel(const float* __restrict__ x, float* __restrict__ out, int64_t n, float scale, float shift) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = x[i] * scale + shift;
}

// Adam (torch.optim.Adam semantics, no weight decay / amsgrad).  The step count lives in device
// memory when step_dev != nullptr so a captured graph advances it on every replay.
__global__ void adam_tick_kernel(int* step_dev) { step_dev[0] += 1; }

__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                            float* __restrict__ v, int64_t n, float lr, float b1, float b2, float eps,
                            float grad_scale, const int* __restrict__ step_dev, int step_host) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n / 4) return;
    const int step = step_dev ? step_dev[0] : step_host;
    const double bc1 = 1.0 - pow((double)b1, (double)step);
    const double bc2 = 1.0 - pow((double)b2, (double)step);
    const float step_size = (float)((double)lr / bc1);
    const float inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
    // grad_scale: 1 / ranks of a data-parallel run whose gradient buffers hold sums over the ranks (times 1 is exact)
    const f32x4 gg = reinterpret_cast<const f32x4*>(g)[i] * grad_scale;
    f32x4 mm = reinterpret_cast<f32x4*>(m)[i], vv = reinterpret_cast<f32x4*>(v)[i], pp = reinterpret_cast<f32x4*>(p)[i];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        mm[e] = b1 * mm[e] + (1.f - b1) * gg[e];
        vv[e] = b2 * vv[e] + (1.f - b2) * gg[e] * gg[e];
        const float denom = sqrtf(vv[e]) * inv_sqrt_bc2 + eps;
        pp[e] -= step_size * (mm[e] / denom);
    }
    reinterpret_cast<f32x4*>(m)[i] = mm;
    reinterpret_cast<f32x4*>(v)[i] = vv;
    reinterpret_cast<f32x4*>(p)[i] = pp;
}

// Philox4x32-10 counter-based generator; state[0..1] = seed (lo, hi), state[2..3] = 64-bit offset
// counter in units of 4 outputs.  The draw kernel reads the offset; the advance kernel bumps it, so
// graph replays produce fresh numbers without any host involvement.
__device__ __forceinline__ void philox_round(uint32_t (&c)[4], uint32_t (&k)[2]) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
    const uint32_t hi0 = __umulhi(M0, c[0]), lo0 = M0 * c[0];
    const uint32_t hi1 = __umulhi(M1, c[2]), lo1 = M1 * c[2];
    const uint32_t n0 = hi1 ^ c[1] ^ k[0], n1 = lo1, n2 = hi0 ^ c[3] ^ k[1], n3 = lo0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
    k[0] += 0x9E3779B9u;
    k[1] += 0xBB67AE85u;
}

// Values [skip, skip + n) of the draw that starts at the state's offset (value v comes from Philox call v / 4).
__global__ void philox_kernel(float* __restrict__ out, int64_t n, int64_t skip, const uint32_t* __restrict__ state,
                              int normal) {
    const int64_t i = (skip >> 2) + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // group of 4 outputs
    if (i * 4 >= skip + n) return;
    const uint64_t off = ((uint64_t)state[3] << 32 | state[2]) + (uint64_t)i;
    uint32_t c[4] = {(uint32_t)off, (uint32_t)(off >> 32), 0u, 0u};
    uint32_t k[2] = {state[0], state[1]};
#pragma unroll
    for (int r = 0; r < 10; ++r) philox_round(c, k);
    float o[4];
    if (normal) {   // Box-Muller on two pairs
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const float u1 = ((c[2 * e] >> 8) + 1u) * (1.0f / 16777216.0f);       // (0, 1]
            const float u2 = (c[2 * e + 1] >> 8) * (1.0f / 16777216.0f);          // [0, 1)
            const float rad = sqrtf(-2.f * logf(u1));
            float sn, cs;
            sincosf(6.28318530717958647692f * u2, &sn, &cs);
            o[2 * e] = rad * cs;
            o[2 * e + 1] = rad * sn;
        }
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (c[e] >> 8) * (1.0f / 16777216.0f);   // [0, 1)
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int64_t v = i * 4 + e - skip;
        if (v >= 0 && v < n) out[v] = o[e];
    }
}

__global__ void philox_advance_kernel(uint32_t* state, uint64_t groups) {
    const uint64_t off = ((uint64_t)state[3] << 32 | state[2]) + groups;
    state[2] = (uint32_t)off;
    state[3] = (uint32_t)(off >> 32);
}

// out_s[b] = pool_s[(cursor + b) % npool], same for positions; cursor lives in device memory.
__global__ void take_batch_kernel(const float* __restrict__ pool_s, const float* __restrict__ pool_p,
                                  float* __restrict__ out_s, float* __restrict__ out_p, const int* __restrict__ cursor,
                                  int npool, int B, int Ms, int Mp, int offset) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int c0 = (cursor[0] + offset) % npool;
    if (i < (int64_t)B * Ms / 4) {
        const int b = (int)(i / (Ms / 4)), e = (int)(i % (Ms / 4));
        const int src = (c0 + b) % npool;
        reinterpret_cast<f32x4*>(out_s)[i] = reinterpret_cast<const f32x4*>(pool_s + (int64_t)src * Ms)[e];
    }
    if (i < (int64_t)B * Mp / 4) {
        const int b = (int)(i / (Mp / 4)), e = (int)(i % (Mp / 4));
        const int src = (c0 + b) % npool;
        reinterpret_cast<f32x4*>(out_p)[i] = reinterpret_cast<const f32x4*>(pool_p + (int64_t)src * Mp)[e];
    }
}
__global__ void cursor_advance_kernel(int* cursor, int npool, int B) { cursor[0] = (cursor[0] + B) % npool; }

inline int grid_for(int64_t n, int bs) { return (int)((n + bs - 1) / bs); }

// up to ZERO_MAX buffers cleared in one launch (blockIdx.y selects the buffer); sizes in 16-byte units
constexpr int ZERO_MAX = 16;
struct ZeroSet {
    f32x4* p[ZERO_MAX];
    long long n4[ZERO_MAX];
};
__global__ __launch_bounds__(256) void zero_multi_kernel(ZeroSet zs) {
    f32x4* p = zs.p[blockIdx.y];
    const long long n4 = zs.n4[blockIdx.y];
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) p[i] = z;
}

}  // namespace

extern "C" {

// Clears n <= 16 device buffers (16-byte aligned, sizes multiples of 16 bytes) in one launch: the outputs of the
// convolution launches of a step that add into zeroed memory (pgan_conv_down_*_zeroed), cleared together at the step's
// start instead of by one launch each in the middle of the chains.
int pgan_zero_multi(void* const* bufs, const long long* bytes, int n, hipStream_t st) {
    if (n <= 0) return PGAN_OK;
    if (n > ZERO_MAX || !bufs || !bytes) return PGAN_EINVAL;
    ZeroSet zs;
    long long most = 0;
    for (int i = 0; i < n; ++i) {
        if (!bufs[i] || bytes[i] <= 0 || bytes[i] % 16 || reinterpret_cast<uintptr_t>(bufs[i]) % 16) return PGAN_EINVAL;
        zs.p[i] = static_cast<f32x4*>(bufs[i]);
        zs.n4[i] = bytes[i] / 16;
        most = zs.n4[i] > most ? zs.n4[i] : most;
    }
    for (int i = n; i < ZERO_MAX; ++i) { zs.p[i] = nullptr; zs.n4[i] = 0; }
    long long bx = (most + 255) / 256;
    if (bx > 1024) bx = 1024;
    zero_multi_kernel<<<dim3((unsigned)bx, (unsigned)n), 256, 0, st>>>(zs);
    return pgan_launch_status();
}

int pgan_head_fwd(const float* h4, const float* f1, const float* w, const float* b, float* score, int N, int K2, int K1,
                  hipStream_t st) {
    if (N <= 0 || K2 % 4) return PGAN_EINVAL;
    head_fwd_kernel<<<N, 256, 0, st>>>(h4, f1, w, b, score, K2, K1);
    return pgan_launch_status();
}

int pgan_head_seed(const float* w, const float* coef, float* u4, int N, int K, hipStream_t st) {
    if (N <= 0 || K % 4) return PGAN_EINVAL;
    head_seed_kernel<<<grid_for((int64_t)N * K / 4, 256), 256, 0, st>>>(w, coef, u4, N, K);
    return pgan_launch_status();
}

int pgan_interpolate(const float* t, const float* real, const float* fake, float* out, int B, int M, hipStream_t st) {
    if (B <= 0 || M % 4) return PGAN_EINVAL;
    interpolate_kernel<<<grid_for((int64_t)B * M / 4, 256), 256, 0, st>>>(t, real, fake, out, B, M);
    return pgan_launch_status();
}

// pen[0] = lam * mean((|g|-1)^2), v = d pen / d g.  ws: >= 256 floats of scratch.
int pgan_gp_penalty(const float* g, float* v, float* pen, float* ws, long long n, float lam, hipStream_t st) {
    if (n <= 0 || n % 4) return PGAN_EINVAL;
    int blocks = grid_for(n / 4, 256);
    if (blocks > 256) blocks = 256;
    gp_penalty_kernel<<<blocks, 256, 0, st>>>(g, v, ws, n, 2.f * lam / (float)n);
    sum_scale_kernel<<<1, 256, 0, st>>>(ws, pen, blocks, lam / (float)n);
    return pgan_launch_status();
}

// g: [2*nb, M]; pen[0] = (1/nb) sum_n |g_n|^6, v = d pen / d g.  ws: >= 2*nb floats of scratch.
int pgan_div_penalty(const float* g, float* v, float* pen, float* ws, int nb, int M, hipStream_t st) {
    if (nb <= 0 || M % 4) return PGAN_EINVAL;
    div_penalty_kernel<<<2 * nb, 256, 0, st>>>(g, v, ws, M, 1.f / (float)nb);
    sum_scale_kernel<<<1, 256, 0, st>>>(ws, pen, 2 * nb, 1.f / (float)nb);
    return pgan_launch_status();
}

int pgan_critic_loss(const float* score, const float* pen, float* out, int nb, hipStream_t st) {
    if (nb <= 0) return PGAN_EINVAL;
    critic_loss_kernel<<<1, 256, 0, st>>>(score, pen, out, nb);
    return pgan_launch_status();
}

// out[0] = scale * sum(x[0:n])
int pgan_sum_scale(const float* x, float* out, int n, float scale, hipStream_t st) {
    if (n <= 0) return PGAN_EINVAL;
    sum_scale_kernel<<<1, 256, 0, st>>>(x, out, n, scale);
    return pgan_launch_status();
}

int pgan_act_bwd(const float* dy, const float* y, float* dx, long long n, int act, hipStream_t st) {
    if (n <= 0 || (act != 1 && act != 2)) return PGAN_EINVAL;
    act_bwd_kernel<<<grid_for(n, 256), 256, 0, st>>>(dy, y, dx, n, act);
    return pgan_launch_status();
}

int pgan_affine(const float* x, float* out, long long n, float scale, float shift, hipStream_t st) {
    if (n <= 0) return PGAN_EINVAL;
    affine_kernel<<<grid_for(n, 256), 256, 0, st>>>(x, out, n, scale, shift);
    return pgan_launch_status();
}

// n must be a multiple of 4 (flat parameter stores are padded).  step_dev (device int, may be null):
// incremented, then used as the step count; otherwise `step` is used.
int pgan_adam(float* p, const float* g, float* m, float* v, long long n, float lr, float b1, float b2, float eps,
              float grad_scale, int* step_dev, int step, hipStream_t st) {
    if (n <= 0 || n % 4) return PGAN_EINVAL;
    if (step_dev) adam_tick_kernel<<<1, 1, 0, st>>>(step_dev);
    adam_kernel<<<grid_for(n / 4, 256), 256, 0, st>>>(p, g, m, v, n, lr, b1, b2, eps, grad_scale, step_dev, step);
    return pgan_launch_status();
}

// Next B alignments of a device-resident pool (wrapping), then cursor += B.
int pgan_take_batch_slice(const float* pool_s, const float* pool_p, float* out_s, float* out_p, int* cursor, int npool,
                          int B, int Ms, int Mp, int offset, int advance, hipStream_t st) {
    if (B <= 0 || npool <= 0 || Ms % 4 || Mp % 4 || Mp > Ms || offset < 0 || advance < 0 || !cursor) return PGAN_EINVAL;
    take_batch_kernel<<<grid_for((int64_t)B * Ms / 4, 256), 256, 0, st>>>(pool_s, pool_p, out_s, out_p, cursor, npool, B, Ms,
                                                                          Mp, offset % npool);
    cursor_advance_kernel<<<1, 1, 0, st>>>(cursor, npool, advance % npool);
    return pgan_launch_status();
}

int pgan_take_batch(const float* pool_s, const float* pool_p, float* out_s, float* out_p, int* cursor, int npool, int B,
                    int Ms, int Mp, hipStream_t st) {
    return pgan_take_batch_slice(pool_s, pool_p, out_s, out_p, cursor, npool, B, Ms, Mp, 0, B, st);
}

// state: 4 x uint32 in device memory (seed lo, seed hi, offset lo, offset hi).  normal: 1 -> N(0,1), 0 -> U[0,1).
// Values [skip, skip + n) of a draw of `total` values; the state advances by the whole draw.  Ranks of a data-parallel run
// that share one generator state each take their rows of the global draw this way.
int pgan_philox_slice(float* out, long long n, long long skip, long long total, unsigned int* state, int normal,
                      hipStream_t st) {
    if (n <= 0 || skip < 0 || skip + n > total || !out || !state) return PGAN_EINVAL;
    const long long groups = (skip + n + 3) / 4 - skip / 4;
    philox_kernel<<<grid_for(groups, 256), 256, 0, st>>>(out, n, skip, state, normal);
    philox_advance_kernel<<<1, 1, 0, st>>>(state, (uint64_t)((total + 3) / 4));
    return pgan_launch_status();
}

int pgan_philox(float* out, long long n, unsigned int* state, int normal, hipStream_t st) {
    return pgan_philox_slice(out, n, 0, n, state, normal, st);
}

// One training step (any sequence of the entry points above, on `st` and on streams forked from it with events) as
// a hipGraph: begin, enqueue the step once, end -> executable graph; launch replays it.  Relaxed capture mode:
// other host threads may keep calling HIP.  Nothing is launched between begin and end.
int pgan_step_graph_begin(hipStream_t st) {
    return hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed) == hipSuccess ? PGAN_OK : PGAN_ELAUNCH;
}

int pgan_step_graph_end(hipStream_t st, void** graph_exec) {
    if (!graph_exec) return PGAN_EINVAL;
    hipGraph_t g = nullptr;
    if (hipStreamEndCapture(st, &g) != hipSuccess || !g) return PGAN_ELAUNCH;
    hipGraphExec_t ex = nullptr;
    const hipError_t e = hipGraphInstantiate(&ex, g, nullptr, nullptr, 0);
    (void)hipGraphDestroy(g);
    if (e != hipSuccess) return PGAN_ELAUNCH;
    *graph_exec = ex;
    return PGAN_OK;
}

int pgan_step_graph_launch(void* graph_exec, hipStream_t st) {
    if (!graph_exec) return PGAN_EINVAL;
    return hipGraphLaunch(static_cast<hipGraphExec_t>(graph_exec), st) == hipSuccess ? PGAN_OK : PGAN_ELAUNCH;
}

int pgan_step_graph_destroy(void* graph_exec) {
    if (!graph_exec) return PGAN_EINVAL;
    return hipGraphExecDestroy(static_cast<hipGraphExec_t>(graph_exec)) == hipSuccess ? PGAN_OK : PGAN_ELAUNCH;
}

}  // extern "C"
