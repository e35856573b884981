// GroupNorm(1,C)+LeakyReLU (discriminator, per-sample statistics) and training-mode
// BatchNorm+LeakyReLU (generator, per-channel batch statistics): forward, backward, and for
// GroupNorm the tangent (JVP) and second-order backward the analytic penalty gradients need.
// HBM/L2-bound streaming kernels: float4 accesses, fp32 per-thread partial sums, fp64
// cross-thread reductions.  Reference ops: src/layers.py:442-465 (GroupNorm), :93-108,:119-134
// (BatchNorm), LeakyReLU(0.1) throughout.
//
// Notation per sample (M = C*HW elements): xh = (a - mu) * r, r = rsqrt(var + eps),
// y = gamma_c xh + beta_c, h = lrelu(y), m = lrelu'(y).  J = r (I - 11^T/M - xh xh^T/M) is both
// the Jacobian d xh / d a and (being symmetric) its transpose.
#include "pgan_common.h"

#include <atomic>
#include <type_traits>

namespace {


// Adds v into this wave's row of sh[i] (CW rows of 64 channels, one row per wave of the block) at channel c after
// reducing it over the GL consecutive lanes that share the channel.  Rows are private to a wave and the lanes that
// write hold different channels, so no atomics are needed and the sums do not depend on the waves' timing; the caller
// adds the rows in a fixed order (channel_total) after a barrier.
constexpr int CW = 4;
// Sum of v over the GL consecutive lanes (4 ... 64) that share a channel, in every one of them: lane distances 1-8
// are data-parallel-primitive adds at vector-ALU rate (the quad steps leave all lanes of a quad equal, so the 8- and
// 16-lane mirrors pair every lane with its partner's sum), only 16 and 32 go through the LDS crossbar.  These per-channel
// reductions run once per loop iteration of the backward kernels' sums passes; done as six dependent crossbar shuffles
// each, they -- not the bytes streamed -- set those kernels' time.
template <int GL>
__device__ __forceinline__ float group_sum(float x) {
    static_assert(GL == 4 || GL == 8 || GL == 16 || GL == 32 || GL == 64, "lanes per channel");
    x += dpp_move<0xB1>(x);
    x += dpp_move<0x4E>(x);
    if constexpr (GL >= 8) x += dpp_move<0x141>(x);
    if constexpr (GL >= 16) x += dpp_move<0x140>(x);
    if constexpr (GL >= 32) x += __shfl_xor(x, 16, 64);
    if constexpr (GL >= 64) x += __shfl_xor(x, 32, 64);
    return x;
}
template <int GL, int NV>
__device__ __forceinline__ void channel_accumulate(float (&v)[NV], float* const (&sh)[NV], int c) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float x[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) x[i] = group_sum<GL>(v[i]);        // independent chains: their latencies overlap
    if ((lane & (GL - 1)) == 0) {
#pragma unroll
        for (int i = 0; i < NV; ++i) sh[i][wave * 64 + c] += x[i];
    }
}
__device__ __forceinline__ float channel_total(const float* sh, int c) {
    return (sh[c] + sh[64 + c]) + (sh[128 + c] + sh[192 + c]);
}

// Every GroupNorm op is two launches over (sample, slice) blocks, SL slices per sample so that small batches
// still fill the chip: a sums pass that leaves each slice's partial sums (fp64) in `ws`, and an apply pass
// in which every block adds the SL partials of its sample in a fixed order (deterministic) and streams its
// slice.  A slice is a whole number of channels, so per-channel gamma / beta sums belong to one block.
constexpr int SL = 8;     // slices per sample
constexpr int ST = 256;   // threads per block

struct Slice {
    int n, s, tid;
    int64_t base;     // first element of the slice
    int q;            // float4 units in the slice
    int c0;           // first channel of the slice
    __device__ Slice(int M, int C) {
        n = blockIdx.x / SL; s = blockIdx.x % SL; tid = threadIdx.x;
        base = (int64_t)n * M + (int64_t)s * (M / SL);
        q = M / SL / 4;
        c0 = s * (C / SL);
    }
};

template <int NV>
__device__ __forceinline__ void put_partials(double (&red)[NV], double* sh, double* ws, const Slice& sl) {
    block_sum<NV>(red, sh);
    if (sl.tid == 0) {
#pragma unroll
        for (int i = 0; i < NV; ++i) ws[((int64_t)sl.n * SL + sl.s) * NV + i] = red[i];
    }
}
template <int NV>
__device__ __forceinline__ void get_totals(double (&tot)[NV], const double* ws, int n) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        double t = 0.0;
        for (int s = 0; s < SL; ++s) t += ws[((int64_t)n * SL + s) * NV + i];
        tot[i] = t;
    }
}

// Largest |value| a block wrote, for the consumers that scale their operands (the fp16-split convolutions): every
// thread passes the largest magnitude among its outputs; `dst` (may be null) receives the block's.  256-thread blocks.
__device__ __forceinline__ float block_amax256(float m) {
    __shared__ float shm[4];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0) shm[threadIdx.x >> 6] = m;
    __syncthreads();
    return fmaxf(fmaxf(shm[0], shm[1]), fmaxf(shm[2], shm[3]));
}
__device__ __forceinline__ float amax4(float m, const f32x4& o) {
    return fmaxf(fmaxf(m, fmaxf(fabsf(o[0]), fabsf(o[1]))), fmaxf(fabsf(o[2]), fabsf(o[3])));
}

__global__ __launch_bounds__(ST) void gn_fwd_sums(const float* __restrict__ a, double* __restrict__ ws, int M, int C) {
    __shared__ double sh[2 * ST / 64];
    const Slice sl(M, C);
    const f32x4* a4 = reinterpret_cast<const f32x4*>(a + sl.base);
    float s = 0.f, ss = 0.f;
    for (int i = sl.tid; i < sl.q; i += ST) {
        const f32x4 v = a4[i];
        s += (v[0] + v[1]) + (v[2] + v[3]);
        ss += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
    }
    double red[2] = {s, ss};
    put_partials<2>(red, sh, ws, sl);
}
__global__ __launch_bounds__(ST) void gn_fwd_apply(const float* __restrict__ a, const float* __restrict__ gamma,
                                                   const float* __restrict__ beta, float* __restrict__ h,
                                                   float* __restrict__ stats, const double* __restrict__ ws, int M,
                                                   int HW, int C, float* __restrict__ amax) {
    const Slice sl(M, C);
    double tot[2];
    get_totals<2>(tot, ws, sl.n);
    const double mu_d = tot[0] / M;
    const double var = fmax(tot[1] / M - mu_d * mu_d, 0.0);
    const float mu = (float)mu_d, r = (float)(1.0 / sqrt(var + (double)PGAN_EPS));
    if (sl.s == 0 && sl.tid == 0) {
        stats[2 * sl.n] = mu;
        stats[2 * sl.n + 1] = r;
    }
    const f32x4* a4 = reinterpret_cast<const f32x4*>(a + sl.base);
    f32x4* h4 = reinterpret_cast<f32x4*>(h + sl.base);
    float am = 0.f;
    for (int i = sl.tid; i < sl.q; i += ST) {
        const int c = sl.c0 + (i * 4) / HW;
        const float g = gamma[c], b = beta[c];
        const f32x4 v = a4[i];
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = lrelu(g * ((v[e] - mu) * r) + b);
        am = amax4(am, o);
        h4[i] = o;
    }
    if (amax) {      // uniform over the block
        am = block_amax256(am);
        if (sl.tid == 0) amax[sl.n * SL + sl.s] = am;
    }
}

// The backward kernels need the LeakyReLU mask m of the forward output.  MA = false: taken from h.  MA = true: from the
// sign of gamma xh + beta, the forward pass's own expression on the same a, mean and rstd -- h is then not read at all
// (these kernels are bound by the bytes they stream: one tensor of three to five fewer per pass).
template <bool MA>
__device__ __forceinline__ float mask_of(float hv, float g, float xh, float b) {
    return MA ? lrelu_mask(g * xh + b) : lrelu_mask(hv);
}

// first-order backward: d = J (gamma m u); per-sample stats (mean w, mean w xh); per-sample
// per-channel partial sums of the gamma / beta gradients.
template <bool MA, int GL>
__global__ __launch_bounds__(ST) void gn_bwd_sums(const float* __restrict__ u, const float* __restrict__ h,
                                                  const float* __restrict__ a, const float* __restrict__ stats,
                                                  const float* __restrict__ gamma, const float* __restrict__ beta,
                                                  float* __restrict__ pg,
                                                  float* __restrict__ pb, double* __restrict__ ws, int M, int HW, int C,
                                                  int pld) {
    __shared__ double sh[2 * ST / 64];
    __shared__ float cg[CW * 64], cb[CW * 64];
    const Slice sl(M, C);
    const int cps = C / SL;
    const f32x4* u4 = reinterpret_cast<const f32x4*>(u + sl.base);
    const f32x4* h4 = MA ? nullptr : reinterpret_cast<const f32x4*>(h + sl.base);
    const f32x4* a4 = reinterpret_cast<const f32x4*>(a + sl.base);
    const float mu = stats[2 * sl.n], r = stats[2 * sl.n + 1];
    for (int c = sl.tid; c < CW * 64; c += ST) { cg[c] = 0.f; cb[c] = 0.f; }
    __syncthreads();
    float sw = 0.f, swx = 0.f;
    for (int i = sl.tid; i < sl.q; i += ST) {
        const int cl = (i * 4) / HW;
        const float g = gamma[sl.c0 + cl], b = MA ? beta[sl.c0 + cl] : 0.f;
        const f32x4 uv = u4[i], av = a4[i];
        f32x4 hv = {0.f, 0.f, 0.f, 0.f};
        if constexpr (!MA) hv = h4[i];
        float pe[2] = {0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float xh = (av[e] - mu) * r, ee = mask_of<MA>(hv[e], g, xh, b) * uv[e], w = g * ee;
            sw += w;
            swx += w * xh;
            pe[0] += ee * xh;
            pe[1] += ee;
        }
        float* const dst[2] = {cg, cb};
        channel_accumulate<GL, 2>(pe, dst, cl);
    }
    double red[2] = {sw, swx};
    put_partials<2>(red, sh, ws, sl);   // its barriers also order the per-wave channel rows before the reads below
    for (int c = sl.tid; c < cps; c += ST) {
        pg[(int64_t)sl.n * pld + sl.c0 + c] = channel_total(cg, c);
        pb[(int64_t)sl.n * pld + sl.c0 + c] = channel_total(cb, c);
    }
}
template <bool MA>
__global__ __launch_bounds__(ST) void gn_bwd_apply(const float* __restrict__ u, const float* __restrict__ h,
                                                   const float* __restrict__ a, const float* __restrict__ stats,
                                                   const float* __restrict__ gamma, const float* __restrict__ beta,
                                                   float* __restrict__ d,
                                                   float* __restrict__ stats_b, const double* __restrict__ ws, int M,
                                                   int HW, int C, float* __restrict__ amax) {
    const Slice sl(M, C);
    double tot[2];
    get_totals<2>(tot, ws, sl.n);
    const float mw = (float)(tot[0] / M), mwx = (float)(tot[1] / M);
    if (sl.s == 0 && sl.tid == 0) {
        stats_b[2 * sl.n] = mw;
        stats_b[2 * sl.n + 1] = mwx;
    }
    const f32x4* u4 = reinterpret_cast<const f32x4*>(u + sl.base);
    const f32x4* h4 = MA ? nullptr : reinterpret_cast<const f32x4*>(h + sl.base);
    const f32x4* a4 = reinterpret_cast<const f32x4*>(a + sl.base);
    f32x4* d4 = reinterpret_cast<f32x4*>(d + sl.base);
    const float mu = stats[2 * sl.n], r = stats[2 * sl.n + 1];
    float am = 0.f;
    for (int i = sl.tid; i < sl.q; i += ST) {
        const int c = sl.c0 + (i * 4) / HW;
        const float g = gamma[c], b = MA ? beta[c] : 0.f;
        const f32x4 uv = u4[i], av = a4[i];
        f32x4 hv = {0.f, 0.f, 0.f, 0.f};
        if constexpr (!MA) hv = h4[i];
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float xh = (av[e] - mu) * r, w = g * mask_of<MA>(hv[e], g, xh, b) * uv[e];
            o[e] = r * (w - mw - xh * mwx);
        }
        am = amax4(am, o);
        d4[i] = o;
    }
    if (amax) {      // uniform over the block
        am = block_amax256(am);
        if (sl.tid == 0) amax[sl.n * SL + sl.s] = am;
    }
}

// ------------------------------------------------------------------------------------------------
// One-launch forms.  A slice (an eighth of a sample) is at most 32 elements per thread and tensor, so a block can keep
// what it read for the sums in registers and apply from there: every input is read once instead of twice.  What the
// eight blocks of a sample must exchange in between are their partial sums (2-3 doubles each).  They do that through
// 64-bit slots in device memory that hold a sentinel (all ones: a NaN pattern no sum produces, see publish) between
// launches: a block stores its partials into its slots with device-scope atomic stores and polls the sample's slots
// with device-scope atomic loads until none holds the sentinel -- every slot is its own flag, so no ordering between
// different addresses is needed.  The last block of a sample to finish reading (a per-sample counter) puts the
// sentinels back, so every launch leaves the workspace as it found it.
//   Blocks of one sample are adjacent in launch order (and land on one XCD when N is a multiple of 8, see
// onepass_block), blocks are dispatched in order, and a block that waits only waits for blocks launched at most seven
// positions after it: the earliest sample with a block still waiting always has its missing blocks at the head of the
// dispatch queue, so the waits end.  They are bounded all the same: after SPIN_LIMIT polls a block stops waiting, sets
// the workspace's error word and finishes with whatever it read, so that a broken workspace shows as wrong numbers and
// a flag rather than as a hung device.
//   The arithmetic, its order and therefore the bits of every result equal those of the sums / apply pair.
// ------------------------------------------------------------------------------------------------
constexpr unsigned long long SENTINEL = ~0ull;
constexpr int SPIN_LIMIT = 1 << 22;
constexpr int NVMAX = 3;
// workspace: a 64-byte header (word 0: error flag) and one 200-byte record per sample -- SL * NVMAX slots and the
// sample's counter -- so that the layout does not depend on how many samples a call has
constexpr int REC = SL * NVMAX + 1;      // 64-bit words per sample
struct OnePassWs {
    unsigned long long* rec;     // [capacity][REC]
    unsigned int* error;
    __device__ unsigned long long* slots(int n) const { return rec + (size_t)n * REC; }
    __device__ unsigned int* done(int n) const { return reinterpret_cast<unsigned int*>(rec + (size_t)n * REC + SL * NVMAX); }
};
__host__ __device__ inline long long onepass_bytes(int N) { return 64 + (long long)N * REC * 8; }
inline OnePassWs onepass_ws(void* ws) {
    OnePassWs w;
    w.error = static_cast<unsigned int*>(ws);
    w.rec = reinterpret_cast<unsigned long long*>(static_cast<char*>(ws) + 64);
    return w;
}
// (sample, slice) of this block: with N a multiple of 8 the eight slices of a sample are eight blocks whose indices are
// equal mod 8 -- blocks are dealt to the 8 XCDs round-robin, so a sample's exchange stays inside one XCD's L2 and its
// blocks are neighbours in that XCD's dispatch order; otherwise consecutive blocks.
__device__ __forceinline__ void onepass_block(int N, int& n, int& s) {
    const int b = blockIdx.x;
    if ((N & 7) == 0) {
        const int x = b & 7, j = b >> 3;
        n = (j >> 3) * 8 + x;
        s = j & 7;
    } else {
        n = b / SL;
        s = b % SL;
    }
}
struct SliceAt : Slice {
    __device__ SliceAt(int M, int C, int n_, int s_) : Slice(M, C) {
        n = n_; s = s_;
        base = (int64_t)n * M + (int64_t)s * (M / SL);
        c0 = s * (C / SL);
    }
};
// Block sums of red[] -> the sample's totals in tot[], through the slots (see above).  shv: SL * NV doubles of LDS.
template <int NV>
__device__ __forceinline__ void exchange_partials(double (&red)[NV], double (&tot)[NV], double* sh, double* shv,
                                                  const OnePassWs& w, const Slice& sl) {
    block_sum<NV>(red, sh);
    unsigned long long* mine = w.slots(sl.n);
    if (sl.tid == 0) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            unsigned long long bits = __builtin_bit_cast(unsigned long long, red[i]);
            if (bits == SENTINEL) bits ^= 1ull;            // still a NaN, no longer the sentinel
            __hip_atomic_store(mine + sl.s * NVMAX + i, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (sl.tid < SL * NV) {
        const int s = sl.tid / NV, i = sl.tid % NV;
        unsigned long long bits = SENTINEL;
        int polls = 0;
        while (true) {
            bits = __hip_atomic_load(mine + s * NVMAX + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (bits != SENTINEL || ++polls >= SPIN_LIMIT) break;
            __builtin_amdgcn_s_sleep(1);
        }
        if (bits == SENTINEL) __hip_atomic_store(w.error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        shv[sl.tid] = __builtin_bit_cast(double, bits);
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        double t = 0.0;
        for (int s = 0; s < SL; ++s) t += shv[s * NV + i];
        tot[i] = t;
    }
    // every poll of this block has returned (the barrier above): count it; the sample's last block restores the slots
    if (sl.tid == 0) {
        const unsigned int before = __hip_atomic_fetch_add(w.done(sl.n), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (before == SL - 1) {
            for (int k = 0; k < SL * NVMAX; ++k)
                __hip_atomic_store(mine + k, SENTINEL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(w.done(sl.n), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}
__global__ void onepass_init_kernel(unsigned int* header, unsigned long long* rec, long long nwords) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 16) header[i] = 0u;
    if (i < nwords) rec[i] = (i % REC == SL * NVMAX) ? 0ull : SENTINEL;
}

// first-order backward in one launch: gn_bwd_sums and gn_bwd_apply on the same registers.  Q = float4 per thread and tensor.
template <bool MA, int GL, int Q>
__global__ __launch_bounds__(ST) void gn_bwd_onepass(const float* __restrict__ u, const float* __restrict__ h,
                                                     const float* __restrict__ a, const float* __restrict__ stats,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     float* __restrict__ d, float* __restrict__ stats_b,
                                                     float* __restrict__ pg, float* __restrict__ pb, OnePassWs w, int N,
                                                     int M, int HW, int C, int pld, float* __restrict__ amax) {
    __shared__ double sh[2 * ST / 64], shv[SL * 2];
    __shared__ float cg[CW * 64], cb[CW * 64];
    int n_, s_;
    onepass_block(N, n_, s_);
    const SliceAt sl(M, C, n_, s_);
    const int cps = C / SL;
    const f32x4* u4 = reinterpret_cast<const f32x4*>(u + sl.base);
    const f32x4* h4 = MA ? nullptr : reinterpret_cast<const f32x4*>(h + sl.base);
    const f32x4* a4 = reinterpret_cast<const f32x4*>(a + sl.base);
    f32x4 uv[Q], av[Q], mk[Q];
#pragma unroll
    for (int k = 0; k < Q; ++k) {
        uv[k] = u4[sl.tid + k * ST];
        av[k] = a4[sl.tid + k * ST];
    }
    const float mu = stats[2 * sl.n], r = stats[2 * sl.n + 1];
    for (int c = sl.tid; c < CW * 64; c += ST) { cg[c] = 0.f; cb[c] = 0.f; }
    __syncthreads();
    float sw = 0.f, swx = 0.f;
#pragma unroll
    for (int k = 0; k < Q; ++k) {
        const int i = sl.tid + k * ST, cl = (i * 4) / HW;
        const float g = gamma[sl.c0 + cl], b = MA ? beta[sl.c0 + cl] : 0.f;
        f32x4 hv = {0.f, 0.f, 0.f, 0.f};
        if constexpr (!MA) hv = h4[i];
        float pe[2] = {0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float xh = (av[k][e] - mu) * r, m = mask_of<MA>(hv[e], g, xh, b), ee = m * uv[k][e], wv = g * ee;
            mk[k][e] = m;
            sw += wv;
            swx += wv * xh;
            pe[0] += ee * xh;
            pe[1] += ee;
        }
        float* const dst[2] = {cg, cb};
        channel_accumulate<GL, 2>(pe, dst, cl);
    }
    double red[2] = {sw, swx}, tot[2];
    exchange_partials<2>(red, tot, sh, shv, w, sl);      // its barriers also order the per-wave channel rows
    for (int c = sl.tid; c < cps; c += ST) {
        pg[(int64_t)sl.n * pld + sl.c0 + c] = channel_total(cg, c);
        pb[(int64_t)sl.n * pld + sl.c0 + c] = channel_total(cb, c);
    }
    const float mw = (float)(tot[0] / M), mwx = (float)(tot[1] / M);
    if (sl.s == 0 && sl.tid == 0) {
        stats_b[2 * sl.n] = mw;
        stats_b[2 * sl.n + 1] = mwx;
    }
    f32x4* d4 = reinterpret_cast<f32x4*>(d + sl.base);
    float am = 0.f;
#pragma unroll
    for (int k = 0; k < Q; ++k) {
        const int i = sl.tid + k * ST, c = sl.c0 + (i * 4) / HW;
        const float g = gamma[c];
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float xh = (av[k][e] - mu) * r, wv = g * mk[k][e] * uv[k][e];
            o[e] = r * (wv - mw - xh * mwx);
        }
        am = amax4(am, o);
        d4[i] = o;
    }
    if (amax) {      // uniform over the block
        am = block_amax256(am);
        if (sl.tid == 0) amax[sl.n * SL + sl.s] = am;
    }
}

// tangent forward: hd = m gamma J t.  Primal state (h, a, stats) of sample n; stats_t = (mean t, mean xh t).
__global__ __launch_bounds__(ST) void gn_jvp_sums(const float* __restrict__ t, const float* __restrict__ a,
                                                  const float* __restrict__ stats, double* __restrict__ ws, int M, int C) {
    __shared__ double sh[2 * ST / 64];
    const Slice sl(M, C);
    const f32x4* t4 = reinterpret_cast<const f32x4*>(t + sl.base);
    const f32x4* a4 = reinterpret_cast<const f32x4*>(a + sl.base);
    const float mu = stats[2 * sl.n], r = stats[2 * sl.n + 1];
    float st = 0.f, sxt = 0.f;
    for (int i = sl.tid; i < sl.q; i += ST) {
        const f32x4 tv = t4[i], av = a4[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            st += tv[e];
            sxt += (av[e] - mu) * r * tv[e];
        }
    }
    double red[2] = {st, sxt};
    put_partials<2>(red, sh, ws, sl);
}
template <bool MA>
__global__ __launch_bounds__(ST) void gn_jvp_apply(const float* __restrict__ t, const float* __restrict__ h,
                                                   const float* __restrict__ a, const float* __restrict__ stats,
                                                   const float* __restrict__ gamma, const float* __restrict__ beta,
                                                   float* __restrict__ hd,
                                                   float* __restrict__ stats_t, const double* __restrict__ ws, int M,
                                                   int HW, int C, float* __restrict__ amax) {
    const Slice sl(M, C);
    double tot[2];
    get_totals<2>(tot, ws, sl.n);
    const float tb = (float)(tot[0] / M), c_ = (float)(tot[1] / M);
    if (sl.s == 0 && sl.tid == 0) {
        stats_t[2 * sl.n] = tb;
        stats_t[2 * sl.n + 1] = c_;
    }
    const f32x4* t4 = reinterpret_cast<const f32x4*>(t + sl.base);
    const f32x4* h4 = MA ? nullptr : reinterpret_cast<const f32x4*>(h + sl.base);
    const f32x4* a4 = reinterpret_cast<const f32x4*>(a + sl.base);
    f32x4* o4 = reinterpret_cast<f32x4*>(hd + sl.base);
    const float mu = stats[2 * sl.n], r = stats[2 * sl.n + 1];
    float am = 0.f;
    for (int i = sl.tid; i < sl.q; i += ST) {
        const int c = sl.c0 + (i * 4) / HW;
        const float g = gamma[c], b = MA ? beta[c] : 0.f;
        const f32x4 tv = t4[i], av = a4[i];
        f32x4 hv = {0.f, 0.f, 0.f, 0.f};
        if constexpr (!MA) hv = h4[i];
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float xh = (av[e] - mu) * r;
            o[e] = mask_of<MA>(hv[e], g, xh, b) * g * (r * (tv[e] - tb - xh * c_));
        }
        am = amax4(am, o);
        o4[i] = o;
    }
    if (amax) {      // uniform over the block
        am = block_amax256(am);
        if (sl.tid == 0) amax[sl.n * SL + sl.s] = am;
    }
}

// second-order backward of S = <hbar-path> + sum tangent score, for the primal input a:
//   w = gamma m u, q = gamma m hbar, Q = q - r (c w + mwx t)
//   p = r [ Q - mean Q - xh (mean(Q xh) + mean(w xd)) ],  xd = r (t - tb - xh c)
// plus per-channel partials  pg = sum m u xd + m hbar xh,  pb = sum m hbar.   hbar may be null (= 0).
template <bool MA, int GL>
__global__ __launch_bounds__(ST) void gn_bwd2_sums(const float* __restrict__ hbar, const float* __restrict__ u,
                                                   const float* __restrict__ t, const float* __restrict__ h,
                                                   const float* __restrict__ a, const float* __restrict__ stats,
                                                   const float* __restrict__ stats_t, const float* __restrict__ gamma,
                                                   const float* __restrict__ beta,
                                                   float* __restrict__ pg, float* __restrict__ pb,
                                                   double* __restrict__ ws, int M, int HW, int C, int pld) {
    __shared__ double sh[3 * ST / 64];
    __shared__ float cg[CW * 64], cb[CW * 64];
    const Slice sl(M, C);
    const int cps = C / SL;
    const f32x4* hb4 = hbar ? reinterpret_cast<const f32x4*>(hbar + sl.base) : nullptr;
    const f32x4* u4 = reinterpret_cast<const f32x4*>(u + sl.base);
    const f32x4* t4 = reinterpret_cast<const f32x4*>(t + sl.base);
    const f32x4* h4 = MA ? nullptr : reinterpret_cast<const f32x4*>(h + sl.base);
    const f32x4* a4 = reinterpret_cast<const f32x4*>(a + sl.base);
    const float mu = stats[2 * sl.n], r = stats[2 * sl.n + 1];
    const float tb = stats_t[2 * sl.n], c_ = stats_t[2 * sl.n + 1];
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    for (int c = sl.tid; c < CW * 64; c += ST) { cg[c] = 0.f; cb[c] = 0.f; }
    __syncthreads();
    float sq = 0.f, sqx = 0.f, swt = 0.f;
    for (int i = sl.tid; i < sl.q; i += ST) {
        const int cl = (i * 4) / HW;
        const float g = gamma[sl.c0 + cl], b = MA ? beta[sl.c0 + cl] : 0.f;
        const f32x4 uv = u4[i], av = a4[i], tv = t4[i], bv = hb4 ? hb4[i] : zero;
        f32x4 hv = zero;
        if constexpr (!MA) hv = h4[i];
        float pe[2] = {0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float xh = (av[e] - mu) * r, m = mask_of<MA>(hv[e], g, xh, b);
            const float q = g * m * bv[e], w = g * m * uv[e];
            const float xd = r * (tv[e] - tb - xh * c_);
            sq += q;
            sqx += q * xh;
            swt += w * tv[e];
            pe[0] += m * uv[e] * xd + m * bv[e] * xh;
            pe[1] += m * bv[e];
        }
        float* const dst[2] = {cg, cb};
        channel_accumulate<GL, 2>(pe, dst, cl);
    }
    double red[3] = {sq, sqx, swt};
    put_partials<3>(red, sh, ws, sl);
    for (int c = sl.tid; c < cps; c += ST) {
        pg[(int64_t)sl.n * pld + sl.c0 + c] = channel_total(cg, c);
        pb[(int64_t)sl.n * pld + sl.c0 + c] = channel_total(cb, c);
    }
}
template <bool MA>
__global__ __launch_bounds__(ST) void gn_bwd2_apply(const float* __restrict__ hbar, const float* __restrict__ u,
                                                    const float* __restrict__ t, const float* __restrict__ h,
                                                    const float* __restrict__ a, const float* __restrict__ stats,
                                                    const float* __restrict__ stats_b, const float* __restrict__ stats_t,
                                                    const float* __restrict__ gamma, const float* __restrict__ beta,
                                                    float* __restrict__ p,
                                                    const double* __restrict__ ws, int M, int HW, int C, float* __restrict__ amax) {
    const Slice sl(M, C);
    double tot[3];
    get_totals<3>(tot, ws, sl.n);
    const float mu = stats[2 * sl.n], r = stats[2 * sl.n + 1];
    const float mw = stats_b[2 * sl.n], mwx = stats_b[2 * sl.n + 1];
    const float tb = stats_t[2 * sl.n], c_ = stats_t[2 * sl.n + 1];
    const float mq = (float)(tot[0] / M), mqx = (float)(tot[1] / M), mwt = (float)(tot[2] / M);
    const float mQ = mq - r * (c_ * mw + mwx * tb);
    const float mQx = mqx - 2.f * r * c_ * mwx;
    const float mwxd = r * (mwt - mw * tb - mwx * c_);
    const float kx = mQx + mwxd;
    const f32x4* hb4 = hbar ? reinterpret_cast<const f32x4*>(hbar + sl.base) : nullptr;
    const f32x4* u4 = reinterpret_cast<const f32x4*>(u + sl.base);
    const f32x4* t4 = reinterpret_cast<const f32x4*>(t + sl.base);
    const f32x4* h4 = MA ? nullptr : reinterpret_cast<const f32x4*>(h + sl.base);
    const f32x4* a4 = reinterpret_cast<const f32x4*>(a + sl.base);
    f32x4* p4 = reinterpret_cast<f32x4*>(p + sl.base);
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    float am = 0.f;
    for (int i = sl.tid; i < sl.q; i += ST) {
        const int c = sl.c0 + (i * 4) / HW;
        const float g = gamma[c], b = MA ? beta[c] : 0.f;
        const f32x4 uv = u4[i], av = a4[i], tv = t4[i], bv = hb4 ? hb4[i] : zero;
        f32x4 hv = zero;
        if constexpr (!MA) hv = h4[i];
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float xh = (av[e] - mu) * r, m = mask_of<MA>(hv[e], g, xh, b);
            const float q = g * m * bv[e], w = g * m * uv[e];
            const float Q = q - r * (c_ * w + mwx * tv[e]);
            o[e] = r * (Q - mQ - xh * kx);
        }
        am = amax4(am, o);
        p4[i] = o;
    }
    if (amax) {      // uniform over the block
        am = block_amax256(am);
        if (sl.tid == 0) amax[sl.n * SL + sl.s] = am;
    }
}

// ---------------- BatchNorm (training mode) + LeakyReLU: one block per channel ----------------
constexpr int BT = 256;

__global__ __launch_bounds__(BT) void bn_fwd_kernel(const float* __restrict__ a, const float* __restrict__ gamma,
                                                    const float* __restrict__ beta, float* __restrict__ rmean,
                                                    float* __restrict__ rvar, float* __restrict__ h,
                                                    float* __restrict__ stats, int B, int C, int HW, float momentum,
                                                    int groups, float* __restrict__ amax) {
    __shared__ double sh[2 * BT / 64];
    const int c = blockIdx.x, tid = threadIdx.x;
    const int q = HW / 4, total = B * q;   // float4 units of one group
    float am = 0.f;
    const float g = gamma[c], bt = beta[c];
    for (int grp = 0; grp < groups; ++grp) {     // groups are consecutive forward passes: running stats in that order
        const float* ag = a + (int64_t)grp * B * C * HW;
        float* hg = h + (int64_t)grp * B * C * HW;
        float s = 0.f, ss = 0.f;
        for (int i = tid; i < total; i += BT) {
            const int b = i / q, p = i % q;
            const f32x4 v = *reinterpret_cast<const f32x4*>(ag + ((int64_t)b * C + c) * HW + p * 4);
            s += (v[0] + v[1]) + (v[2] + v[3]);
            ss += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
        }
        double red[2] = {s, ss};
        block_sum<2>(red, sh);
        const double cnt = (double)B * HW;
        const double mu_d = red[0] / cnt;
        const double var = fmax(red[1] / cnt - mu_d * mu_d, 0.0);
        const float mu = (float)mu_d, r = (float)(1.0 / sqrt(var + (double)PGAN_EPS));
        if (tid == 0) {
            stats[(grp * 2) * C + c] = mu;
            stats[(grp * 2 + 1) * C + c] = r;
            const double unbiased = cnt > 1.0 ? var * cnt / (cnt - 1.0) : var;
            rmean[c] = (1.f - momentum) * rmean[c] + momentum * mu;
            rvar[c] = (1.f - momentum) * rvar[c] + momentum * (float)unbiased;
        }
        for (int i = tid; i < total; i += BT) {
            const int b = i / q, p = i % q;
            const int64_t off = ((int64_t)b * C + c) * HW + p * 4;
            const f32x4 v = *reinterpret_cast<const f32x4*>(ag + off);
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = lrelu(g * ((v[e] - mu) * r) + bt);
            am = amax4(am, o);
            *reinterpret_cast<f32x4*>(hg + off) = o;
        }
        __syncthreads();      // block_sum's scratch is reused by the next group
    }
    if (amax) {
        am = block_amax256(am);
        if (tid == 0) amax[c] = am;
    }
}

__global__ __launch_bounds__(BT) void bn_bwd_kernel(const float* __restrict__ dh, const float* __restrict__ h,
                                                    const float* __restrict__ a, const float* __restrict__ stats,
                                                    const float* __restrict__ gamma, float* __restrict__ da,
                                                    float* __restrict__ dgamma, float* __restrict__ dbeta, int B, int C,
                                                    int HW, float* __restrict__ amax) {
    __shared__ double sh[2 * BT / 64];
    const int c = blockIdx.x, tid = threadIdx.x;
    const int q = HW / 4, total = B * q;
    const float mu = stats[c], r = stats[C + c], g = gamma[c];
    float sb = 0.f, sg = 0.f;
    for (int i = tid; i < total; i += BT) {
        const int b = i / q, p = i % q;
        const int64_t off = ((int64_t)b * C + c) * HW + p * 4;
        const f32x4 dv = *reinterpret_cast<const f32x4*>(dh + off);
        const f32x4 hv = *reinterpret_cast<const f32x4*>(h + off);
        const f32x4 av = *reinterpret_cast<const f32x4*>(a + off);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float dy = lrelu_mask(hv[e]) * dv[e];
            sb += dy;
            sg += dy * ((av[e] - mu) * r);
        }
    }
    double red[2] = {sb, sg};
    block_sum<2>(red, sh);
    const double cnt = (double)B * HW;
    const float mb = (float)(red[0] / cnt), mg = (float)(red[1] / cnt);
    if (tid == 0) {
        dbeta[c] = (float)red[0];
        dgamma[c] = (float)red[1];
    }
    float am = 0.f;
    for (int i = tid; i < total; i += BT) {
        const int b = i / q, p = i % q;
        const int64_t off = ((int64_t)b * C + c) * HW + p * 4;
        const f32x4 dv = *reinterpret_cast<const f32x4*>(dh + off);
        const f32x4 hv = *reinterpret_cast<const f32x4*>(h + off);
        const f32x4 av = *reinterpret_cast<const f32x4*>(a + off);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float dy = lrelu_mask(hv[e]) * dv[e], xh = (av[e] - mu) * r;
            o[e] = g * r * (dy - mb - xh * mg);
        }
        am = amax4(am, o);
        *reinterpret_cast<f32x4*>(da + off) = o;
    }
    if (amax) {
        am = block_amax256(am);
        if (tid == 0) amax[c] = am;
    }
}

// Sliced BatchNorm: SB blocks per channel, each over B / SB samples; same two-launch scheme as GroupNorm.
// Blocks are (channel, group, slice); a group is one forward pass of B samples, SB slices of B / SB samples each.
__global__ __launch_bounds__(BT) void bn_fwd_sums(const float* __restrict__ a, double* __restrict__ ws, int B, int C,
                                                  int HW, int SB, int G, float* __restrict__ amax) {
    __shared__ double sh[2 * BT / 64];
    const int c = blockIdx.x / (SB * G), gs = blockIdx.x % (SB * G), grp = gs / SB, sl = gs % SB, tid = threadIdx.x;
    const int q = HW / 4, bs = B / SB, total = bs * q;
    float s = 0.f, ss = 0.f;
    for (int i = tid; i < total; i += BT) {
        const int b = grp * B + sl * bs + i / q, p = i % q;
        const f32x4 v = *reinterpret_cast<const f32x4*>(a + ((int64_t)b * C + c) * HW + p * 4);
        s += (v[0] + v[1]) + (v[2] + v[3]);
        ss += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
    }
    double red[2] = {s, ss};
    block_sum<2>(red, sh);
    if (amax && gs == 0 && tid == 0) amax[c] = 0.f;      // the apply launch raises it (atomic max: order-free)
    if (tid == 0) { ws[(c * SB * G + gs) * 2] = red[0]; ws[(c * SB * G + gs) * 2 + 1] = red[1]; }
}
__global__ __launch_bounds__(BT) void bn_fwd_apply(const float* __restrict__ a, const float* __restrict__ gamma,
                                                   const float* __restrict__ beta, float* __restrict__ rmean,
                                                   float* __restrict__ rvar, float* __restrict__ h,
                                                   float* __restrict__ stats, const double* __restrict__ ws, int B, int C,
                                                   int HW, float momentum, int SB, int world, int G, float* __restrict__ amax) {
    const int c = blockIdx.x / (SB * G), gs = blockIdx.x % (SB * G), grp = gs / SB, sl = gs % SB, tid = threadIdx.x;
    const double cnt = (double)B * HW * world;   // world > 1: ws holds the partial sums added over all ranks
    auto group_stats = [&](int gi, double& mu_d, double& var) {
        double t0 = 0.0, t1 = 0.0;
        for (int k = 0; k < SB; ++k) { t0 += ws[((c * G + gi) * SB + k) * 2]; t1 += ws[((c * G + gi) * SB + k) * 2 + 1]; }
        mu_d = t0 / cnt;
        var = fmax(t1 / cnt - mu_d * mu_d, 0.0);
    };
    double mu_d, var;
    group_stats(grp, mu_d, var);
    const float mu = (float)mu_d, r = (float)(1.0 / sqrt(var + (double)PGAN_EPS));
    if (sl == 0 && tid == 0) {
        stats[(grp * 2) * C + c] = mu;
        stats[(grp * 2 + 1) * C + c] = r;
    }
    if (gs == 0 && tid == 0) {      // one thread per channel steps the running statistics through the groups in order
        float rm = rmean[c], rv = rvar[c];
        for (int gi = 0; gi < G; ++gi) {
            double m, v;
            group_stats(gi, m, v);
            const double unbiased = cnt > 1.0 ? v * cnt / (cnt - 1.0) : v;
            rm = (1.f - momentum) * rm + momentum * (float)m;
            rv = (1.f - momentum) * rv + momentum * (float)unbiased;
        }
        rmean[c] = rm;
        rvar[c] = rv;
    }
    const float g = gamma[c], bt = beta[c];
    const int q = HW / 4, bs = B / SB, total = bs * q;
    float am = 0.f;
    for (int i = tid; i < total; i += BT) {
        const int b = grp * B + sl * bs + i / q, p = i % q;
        const int64_t off = ((int64_t)b * C + c) * HW + p * 4;
        const f32x4 v = *reinterpret_cast<const f32x4*>(a + off);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = lrelu(g * ((v[e] - mu) * r) + bt);
        am = amax4(am, o);
        *reinterpret_cast<f32x4*>(h + off) = o;
    }
    if (amax) {
        am = block_amax256(am);       // non-negative floats order like their bit patterns
        if (tid == 0) atomicMax(reinterpret_cast<unsigned int*>(amax) + c, __builtin_bit_cast(unsigned int, am));
    }
}
__global__ __launch_bounds__(BT) void bn_bwd_sums(const float* __restrict__ dh, const float* __restrict__ h,
                                                  const float* __restrict__ a, const float* __restrict__ stats,
                                                  double* __restrict__ ws, int B, int C, int HW, int SB, float* __restrict__ amax) {
    __shared__ double sh[2 * BT / 64];
    const int c = blockIdx.x / SB, sl = blockIdx.x % SB, tid = threadIdx.x;
    const int q = HW / 4, bs = B / SB, total = bs * q;
    const float mu = stats[c], r = stats[C + c];
    float sb = 0.f, sg = 0.f;
    for (int i = tid; i < total; i += BT) {
        const int b = sl * bs + i / q, p = i % q;
        const int64_t off = ((int64_t)b * C + c) * HW + p * 4;
        const f32x4 dv = *reinterpret_cast<const f32x4*>(dh + off);
        const f32x4 hv = *reinterpret_cast<const f32x4*>(h + off);
        const f32x4 av = *reinterpret_cast<const f32x4*>(a + off);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float dy = lrelu_mask(hv[e]) * dv[e];
            sb += dy;
            sg += dy * ((av[e] - mu) * r);
        }
    }
    double red[2] = {sb, sg};
    block_sum<2>(red, sh);
    if (amax && sl == 0 && tid == 0) amax[c] = 0.f;
    if (tid == 0) { ws[(c * SB + sl) * 2] = red[0]; ws[(c * SB + sl) * 2 + 1] = red[1]; }
}
__global__ __launch_bounds__(BT) void bn_bwd_apply(const float* __restrict__ dh, const float* __restrict__ h,
                                                   const float* __restrict__ a, const float* __restrict__ stats,
                                                   const float* __restrict__ gamma, float* __restrict__ da,
                                                   float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                   const double* __restrict__ ws, int B, int C, int HW, int SB,
                                                   int world, float* __restrict__ amax) {
    const int c = blockIdx.x / SB, sl = blockIdx.x % SB, tid = threadIdx.x;
    double t0 = 0.0, t1 = 0.0;
    for (int k = 0; k < SB; ++k) { t0 += ws[(c * SB + k) * 2]; t1 += ws[(c * SB + k) * 2 + 1]; }
    const double cnt = (double)B * HW * world;
    const float mb = (float)(t0 / cnt), mg = (float)(t1 / cnt);
    if (sl == 0 && tid == 0) {   // every rank holds the all-rank totals: 1/world of them each, so that the ranks' sum is the total
        dbeta[c] = (float)(t0 / world);
        dgamma[c] = (float)(t1 / world);
    }
    const float mu = stats[c], r = stats[C + c], g = gamma[c];
    const int q = HW / 4, bs = B / SB, total = bs * q;
    float am = 0.f;
    for (int i = tid; i < total; i += BT) {
        const int b = sl * bs + i / q, p = i % q;
        const int64_t off = ((int64_t)b * C + c) * HW + p * 4;
        const f32x4 dv = *reinterpret_cast<const f32x4*>(dh + off);
        const f32x4 hv = *reinterpret_cast<const f32x4*>(h + off);
        const f32x4 av = *reinterpret_cast<const f32x4*>(a + off);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float dy = lrelu_mask(hv[e]) * dv[e], xh = (av[e] - mu) * r;
            o[e] = g * r * (dy - mb - xh * mg);
        }
        am = amax4(am, o);
        *reinterpret_cast<f32x4*>(da + off) = o;
    }
    if (amax) {
        am = block_amax256(am);
        if (tid == 0) atomicMax(reinterpret_cast<unsigned int*>(amax) + c, __builtin_bit_cast(unsigned int, am));
    }
}

// slices per channel: enough blocks to fill the chip, a whole number of samples each, only for large layers
int bn_slices(int B, int C, int HW) {
    if ((long long)B * HW < 8192) return 1;
    int sb = 1;
    while (C * sb * 2 <= 1024 && B % (sb * 2) == 0 && sb < 16) sb *= 2;
    return sb;
}

bool gn_shape_ok(int N, int C, int HW, const void* ws, long long ws_bytes) {
    return N > 0 && C > 0 && C <= 512 && C % SL == 0 && HW >= 16 && (HW & (HW - 1)) == 0 && (C * HW) % (4 * SL) == 0 &&
           ws != nullptr && ws_bytes >= (long long)N * SL * 3 * 8;
}

// pgan_gn_fwd_static: slice partial sums of up to GN_STATIC_MAX_N samples (384 KiB per GPU, held until the process
// ends).  In host code the array's name only stands for the symbol; g_scratch_ptr[device] is its device address.
constexpr int GN_STATIC_MAX_N = 2048;
constexpr int MAX_DEVICES = 64;
static __device__ __align__(256) double scratch[GN_STATIC_MAX_N * SL * 3];
std::atomic<double*> g_scratch_ptr[MAX_DEVICES];

bool onepass_shape_ok(int N, int C, int HW, const void* ws1, long long ws1_bytes, int& q) {
    if (!(N > 0 && C > 0 && C <= 512 && C % SL == 0 && HW >= 16 && (HW & (HW - 1)) == 0 && ws1 && ws1_bytes >= onepass_bytes(N)))
        return false;
    const int M = C * HW;
    if (M % (SL * 4 * ST)) return false;
    q = M / (SL * 4 * ST);
    return q == 1 || q == 2 || q == 4 || q == 8;
}
// f(ma, gl, q) with the three as compile-time constants: mask from a or from h; lanes per channel; float4 per thread
template <typename F>
void onepass_dispatch(bool ma, int HW, int q, F&& f) {
    auto with_q = [&](auto ma_, auto gl_) {
        switch (q) {
            case 1: f(ma_, gl_, std::integral_constant<int, 1>{}); break;
            case 2: f(ma_, gl_, std::integral_constant<int, 2>{}); break;
            case 4: f(ma_, gl_, std::integral_constant<int, 4>{}); break;
            default: f(ma_, gl_, std::integral_constant<int, 8>{});
        }
    };
    auto with_gl = [&](auto ma_) {
        switch (HW) {
            case 16: with_q(ma_, std::integral_constant<int, 4>{}); break;
            case 32: with_q(ma_, std::integral_constant<int, 8>{}); break;
            case 64: with_q(ma_, std::integral_constant<int, 16>{}); break;
            case 128: with_q(ma_, std::integral_constant<int, 32>{}); break;
            default: with_q(ma_, std::integral_constant<int, 64>{});
        }
    };
    if (ma) with_gl(std::true_type{});
    else with_gl(std::false_type{});
}
}  // namespace

extern "C" {

// Scratch bytes the GroupNorm entry points need for N samples (slice partial sums).
long long pgan_gn_workspace_bytes(int N) { return (long long)N * SL * 3 * 8; }

// h = lrelu(GroupNorm(1,C)(a)); stats[n] = (mean, rstd).  a, h: [N, C, HW].
int pgan_gn_fwd(const float* a, const float* gamma, const float* beta, float* h, float* stats, void* ws,
                long long ws_bytes, int N, int C, int HW, float* amax, hipStream_t st) {
    if (!gn_shape_ok(N, C, HW, ws, ws_bytes)) return PGAN_EINVAL;
    double* w = static_cast<double*>(ws);
    gn_fwd_sums<<<N * SL, ST, 0, st>>>(a, w, C * HW, C);
    gn_fwd_apply<<<N * SL, ST, 0, st>>>(a, gamma, beta, h, stats, w, C * HW, HW, C, amax);
    return pgan_launch_status();
}

// pgan_gn_fwd without its sums launch: ws already holds the slice partial sums ([N][8][2] doubles: sum and sum of
// squares of each eighth of every sample), left there by the producer of a on the same stream
// (pgan_conv_down_gn_sums).
int pgan_gn_fwd_presummed(const float* a, const float* gamma, const float* beta, float* h, float* stats, const void* ws,
                          long long ws_bytes, int N, int C, int HW, float* amax, hipStream_t st) {
    if (!gn_shape_ok(N, C, HW, ws, ws_bytes)) return PGAN_EINVAL;
    gn_fwd_apply<<<N * SL, ST, 0, st>>>(a, gamma, beta, h, stats, static_cast<const double*>(ws), C * HW, HW, C, amax);
    return pgan_launch_status();
}

// The same operation for callers that cannot allocate (code under graph capture, serving loops, harnesses that forbid
// hipMalloc): the slice partial sums live in `scratch`, a fixed array inside the library's code object, instead of a
// caller workspace.  Each GPU holds its own copy of the array at its own address: the address is looked up once per
// device, with that device current, and kept.  Every partial sum that the apply launch reads is written by the sums
// launch of the same call, so nothing needs clearing between calls; two calls that overlap on one GPU (different
// streams) share the array and race.
// Address of `scratch` on the device that owns stream st (the current device for the null stream), looked up once per
// device and kept.  The lookup is a runtime call that may load the code object: it must not happen while st is being
// captured, so a first use under capture is refused (PGAN_EINVAL) -- call pgan_gn_fwd_static_init beforehand.
static int static_scratch_of(hipStream_t st, double** out) {
    int dev = -1;
    hipDevice_t sdev;
    if (st && hipStreamGetDevice(st, &sdev) == hipSuccess) dev = (int)sdev;
    else if (hipGetDevice(&dev) != hipSuccess) return PGAN_ELAUNCH;
    if (dev < 0 || dev >= MAX_DEVICES) return PGAN_ELAUNCH;
    double* w = g_scratch_ptr[dev].load(std::memory_order_acquire);
    if (!w) {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (st && hipStreamIsCapturing(st, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone) return PGAN_EINVAL;
        int cur = -1;
        if (hipGetDevice(&cur) != hipSuccess) return PGAN_ELAUNCH;
        if (cur != dev && hipSetDevice(dev) != hipSuccess) return PGAN_ELAUNCH;       // each GPU has its own copy
        void* p = nullptr;
        const hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (cur != dev) (void)hipSetDevice(cur);
        if (e != hipSuccess || !p) return PGAN_ELAUNCH;
        w = static_cast<double*>(p);
        g_scratch_ptr[dev].store(w, std::memory_order_release);
    }
    *out = w;
    return PGAN_OK;
}

// Looks the array's address up for the current device now, outside any capture (optional: the first
// pgan_gn_fwd_static call on a device does it too, unless its stream is being captured).
int pgan_gn_fwd_static_init(void) {
    double* w = nullptr;
    return static_scratch_of(nullptr, &w);
}

int pgan_gn_fwd_static(const float* a, const float* gamma, const float* beta, float* h, float* stats, int N, int C,
                       int HW, hipStream_t st) {
    if (N > GN_STATIC_MAX_N) return PGAN_ELIMIT;
    double* w = nullptr;
    const int rc = static_scratch_of(st, &w);
    if (rc != PGAN_OK) return rc;
    if (!gn_shape_ok(N, C, HW, w, (long long)sizeof(double) * GN_STATIC_MAX_N * SL * 3)) return PGAN_EINVAL;
    gn_fwd_sums<<<N * SL, ST, 0, st>>>(a, w, C * HW, C);
    gn_fwd_apply<<<N * SL, ST, 0, st>>>(a, gamma, beta, h, stats, w, C * HW, HW, C, nullptr);
    return pgan_launch_status();
}
int pgan_gn_fwd_static_max_samples(void) { return GN_STATIC_MAX_N; }

// beta (here and in pgan_gn_jvp / pgan_gn_bwd2) may be null: the LeakyReLU mask is then read from h.  With beta given it
// is recomputed from a (h may be null and is not read).
int pgan_gn_bwd(const float* u, const float* h, const float* a, const float* stats, const float* gamma, const float* beta,
                float* d, float* stats_b, float* pg, float* pb, int pld, void* ws, long long ws_bytes, int N, int C, int HW,
                float* amax, hipStream_t st) {
    if (!gn_shape_ok(N, C, HW, ws, ws_bytes) || pld < C || (!beta && !h)) return PGAN_EINVAL;
    double* w = static_cast<double*>(ws);
    // lanes per channel in the sums pass: a thread takes four consecutive elements, a channel is HW of them
    auto sums = [&](auto ma, auto gl) {
        gn_bwd_sums<decltype(ma)::value, decltype(gl)::value><<<N * SL, ST, 0, st>>>(u, h, a, stats, gamma, beta, pg, pb, w,
                                                                                     C * HW, HW, C, pld);
    };
    auto by_hw = [&](auto ma) {
        switch (HW) {
            case 16: sums(ma, std::integral_constant<int, 4>{}); break;
            case 32: sums(ma, std::integral_constant<int, 8>{}); break;
            case 64: sums(ma, std::integral_constant<int, 16>{}); break;
            case 128: sums(ma, std::integral_constant<int, 32>{}); break;
            default: sums(ma, std::integral_constant<int, 64>{});
        }
    };
    if (beta) {
        by_hw(std::true_type{});
        gn_bwd_apply<true><<<N * SL, ST, 0, st>>>(u, h, a, stats, gamma, beta, d, stats_b, w, C * HW, HW, C, amax);
    } else {
        by_hw(std::false_type{});
        gn_bwd_apply<false><<<N * SL, ST, 0, st>>>(u, h, a, stats, gamma, beta, d, stats_b, w, C * HW, HW, C, amax);
    }
    return pgan_launch_status();
}

// ---- one-launch forms (see exchange_partials).  ws1: pgan_gn_onepass_workspace_bytes(N) bytes, passed once through
// pgan_gn_onepass_workspace_init after allocation (every call leaves it in that state); one workspace per stream, as for
// the two-launch calls.  Shapes: C * HW / 8192 in {1, 2, 4, 8} (the four stages of the 64 x 64 networks); others return -1
// and launch nothing -- call the two-launch form.  Results are bit-identical to the two-launch form's.
long long pgan_gn_onepass_workspace_bytes(int N) { return N > 0 ? onepass_bytes(N) : -1; }
int pgan_gn_onepass_workspace_init(void* ws1, long long ws1_bytes, int N, hipStream_t st) {
    if (!ws1 || N <= 0 || ws1_bytes < onepass_bytes(N)) return PGAN_EINVAL;
    const OnePassWs w = onepass_ws(ws1);
    const long long nwords = (long long)N * REC;
    onepass_init_kernel<<<(unsigned)((nwords + 255) / 256), 256, 0, st>>>(w.error, w.rec, nwords);
    return pgan_launch_status();
}
// 1 if a one-launch call on this workspace ever gave up waiting (the results of that call are wrong), else 0; -2 on error.
// Synchronises the stream.
int pgan_gn_onepass_workspace_error(const void* ws1, int N, hipStream_t st) {
    if (!ws1 || N <= 0) return PGAN_EINVAL;
    const OnePassWs w = onepass_ws(const_cast<void*>(ws1));
    unsigned int e = 0;
    if (hipMemcpyAsync(&e, w.error, 4, hipMemcpyDeviceToHost, st) != hipSuccess) return PGAN_ELAUNCH;
    if (hipStreamSynchronize(st) != hipSuccess) return PGAN_ELAUNCH;
    return e ? 1 : 0;
}
int pgan_gn_bwd_onepass(const float* u, const float* h, const float* a, const float* stats, const float* gamma,
                        const float* beta, float* d, float* stats_b, float* pg, float* pb, int pld, void* ws1,
                        long long ws1_bytes, int N, int C, int HW, float* amax, hipStream_t st) {
    int q = 0;
    if (!onepass_shape_ok(N, C, HW, ws1, ws1_bytes, q) || pld < C || (!beta && !h)) return PGAN_EINVAL;
    const OnePassWs w = onepass_ws(ws1);
    onepass_dispatch(beta != nullptr, HW, q, [&](auto ma, auto gl, auto qq) {
        gn_bwd_onepass<decltype(ma)::value, decltype(gl)::value, decltype(qq)::value><<<N * SL, ST, 0, st>>>(
            u, h, a, stats, gamma, beta, d, stats_b, pg, pb, w, N, C * HW, HW, C, pld, amax);
    });
    return pgan_launch_status();
}

int pgan_gn_jvp(const float* t, const float* h, const float* a, const float* stats, const float* gamma, const float* beta,
                float* hd, float* stats_t, void* ws, long long ws_bytes, int N, int C, int HW, float* amax, hipStream_t st) {
    if (!gn_shape_ok(N, C, HW, ws, ws_bytes) || (!beta && !h)) return PGAN_EINVAL;
    double* w = static_cast<double*>(ws);
    gn_jvp_sums<<<N * SL, ST, 0, st>>>(t, a, stats, w, C * HW, C);
    if (beta) gn_jvp_apply<true><<<N * SL, ST, 0, st>>>(t, h, a, stats, gamma, beta, hd, stats_t, w, C * HW, HW, C, amax);
    else gn_jvp_apply<false><<<N * SL, ST, 0, st>>>(t, h, a, stats, gamma, beta, hd, stats_t, w, C * HW, HW, C, amax);
    return pgan_launch_status();
}

int pgan_gn_bwd2(const float* hbar, const float* u, const float* t, const float* h, const float* a, const float* stats,
                 const float* stats_b, const float* stats_t, const float* gamma, const float* beta, float* p, float* pg,
                 float* pb, int pld, void* ws, long long ws_bytes, int N, int C, int HW, float* amax, hipStream_t st) {
    if (!gn_shape_ok(N, C, HW, ws, ws_bytes) || pld < C || (!beta && !h)) return PGAN_EINVAL;
    double* w = static_cast<double*>(ws);
    auto sums = [&](auto ma, auto gl) {
        gn_bwd2_sums<decltype(ma)::value, decltype(gl)::value><<<N * SL, ST, 0, st>>>(hbar, u, t, h, a, stats, stats_t, gamma,
                                                                                      beta, pg, pb, w, C * HW, HW, C, pld);
    };
    auto by_hw = [&](auto ma) {
        switch (HW) {
            case 16: sums(ma, std::integral_constant<int, 4>{}); break;
            case 32: sums(ma, std::integral_constant<int, 8>{}); break;
            case 64: sums(ma, std::integral_constant<int, 16>{}); break;
            case 128: sums(ma, std::integral_constant<int, 32>{}); break;
            default: sums(ma, std::integral_constant<int, 64>{});
        }
    };
    if (beta) {
        by_hw(std::true_type{});
        gn_bwd2_apply<true><<<N * SL, ST, 0, st>>>(hbar, u, t, h, a, stats, stats_b, stats_t, gamma, beta, p, w, C * HW, HW, C, amax);
    } else {
        by_hw(std::false_type{});
        gn_bwd2_apply<false><<<N * SL, ST, 0, st>>>(hbar, u, t, h, a, stats, stats_b, stats_t, gamma, beta, p, w, C * HW, HW, C, amax);
    }
    return pgan_launch_status();
}

// a, h: [B, C, HW] (HW multiple of 4); stats: [2, C] = (batch mean, rstd); running stats updated in place.
// ws: scratch of pgan_bn_workspace_bytes(C) bytes.
long long pgan_bn_workspace_bytes(int C) { return (long long)C * 16 * 2 * 8; }

// groups > 1: a holds `groups` consecutive forward passes of B samples each ([groups * B, C, HW]); every pass is
// normalised with its own batch statistics (stats: [groups][2][C]) and the running statistics are stepped through
// the passes in order, exactly as `groups` separate calls would.
int pgan_bn_fwd(const float* a, const float* gamma, const float* beta, float* rmean, float* rvar, float* h, float* stats,
                void* ws, long long ws_bytes, int B, int C, int HW, float momentum, int groups, float* amax,
                hipStream_t st) {
    if (B <= 0 || C <= 0 || HW % 4 || groups < 1) return PGAN_EINVAL;
    const int sb = bn_slices(B, C * groups, HW);
    if (sb == 1 && groups == 1) {
        bn_fwd_kernel<<<C, BT, 0, st>>>(a, gamma, beta, rmean, rvar, h, stats, B, C, HW, momentum, 1, amax);
    } else if (sb == 1 && (long long)B * HW * groups < 8192) {
        bn_fwd_kernel<<<C, BT, 0, st>>>(a, gamma, beta, rmean, rvar, h, stats, B, C, HW, momentum, groups, amax);
    } else {
        if (!ws || ws_bytes < (long long)C * sb * groups * 16) return PGAN_EINVAL;
        double* w = static_cast<double*>(ws);
        bn_fwd_sums<<<C * sb * groups, BT, 0, st>>>(a, w, B, C, HW, sb, groups, amax);
        bn_fwd_apply<<<C * sb * groups, BT, 0, st>>>(a, gamma, beta, rmean, rvar, h, stats, w, B, C, HW, momentum, sb, 1,
                                                     groups, amax);
    }
    return pgan_launch_status();
}

int pgan_bn_bwd(const float* dh, const float* h, const float* a, const float* stats, const float* gamma, float* da,
                float* dgamma, float* dbeta, void* ws, long long ws_bytes, int B, int C, int HW, float* amax,
                hipStream_t st) {
    if (B <= 0 || C <= 0 || HW % 4) return PGAN_EINVAL;
    const int sb = bn_slices(B, C, HW);
    if (sb == 1) {
        bn_bwd_kernel<<<C, BT, 0, st>>>(dh, h, a, stats, gamma, da, dgamma, dbeta, B, C, HW, amax);
    } else {
        if (!ws || ws_bytes < pgan_bn_workspace_bytes(C)) return PGAN_EINVAL;
        double* w = static_cast<double*>(ws);
        bn_bwd_sums<<<C * sb, BT, 0, st>>>(dh, h, a, stats, w, B, C, HW, sb, amax);
        bn_bwd_apply<<<C * sb, BT, 0, st>>>(dh, h, a, stats, gamma, da, dgamma, dbeta, w, B, C, HW, sb, 1, amax);
    }
    return pgan_launch_status();
}

// The same two passes as separate calls, for batch statistics over several GPUs: *_sums writes
// pgan_bn_partials(B, C, HW) doubles of partial sums, the caller adds them over the ranks, *_apply (world = number of
// ranks, each with B samples) finishes from the totals.  dgamma / dbeta come out as 1/world of the all-rank totals.
long long pgan_bn_partials(int B, int C, int HW) { return (long long)C * bn_slices(B, C, HW) * 2; }
// forward with `groups` passes in one call (see pgan_bn_fwd)
long long pgan_bn_fwd_partials(int B, int C, int HW, int groups) {
    return groups < 1 ? -1 : (long long)C * bn_slices(B, C * groups, HW) * groups * 2;
}

int pgan_bn_fwd_sums(const float* a, double* part, int B, int C, int HW, int groups, float* amax, hipStream_t st) {
    if (B <= 0 || C <= 0 || HW % 4 || !part || groups < 1) return PGAN_EINVAL;
    const int sb = bn_slices(B, C * groups, HW);
    bn_fwd_sums<<<C * sb * groups, BT, 0, st>>>(a, part, B, C, HW, sb, groups, amax);
    return pgan_launch_status();
}

int pgan_bn_fwd_apply(const float* a, const float* gamma, const float* beta, float* rmean, float* rvar, float* h,
                      float* stats, const double* part, int B, int C, int HW, int world, float momentum, int groups,
                      float* amax, hipStream_t st) {
    if (B <= 0 || C <= 0 || HW % 4 || !part || world < 1 || groups < 1) return PGAN_EINVAL;
    const int sb = bn_slices(B, C * groups, HW);
    bn_fwd_apply<<<C * sb * groups, BT, 0, st>>>(a, gamma, beta, rmean, rvar, h, stats, part, B, C, HW, momentum, sb, world,
                                                 groups, amax);
    return pgan_launch_status();
}

int pgan_bn_bwd_sums(const float* dh, const float* h, const float* a, const float* stats, double* part, int B, int C,
                     int HW, float* amax, hipStream_t st) {
    if (B <= 0 || C <= 0 || HW % 4 || !part) return PGAN_EINVAL;
    const int sb = bn_slices(B, C, HW);
    bn_bwd_sums<<<C * sb, BT, 0, st>>>(dh, h, a, stats, part, B, C, HW, sb, amax);
    return pgan_launch_status();
}

int pgan_bn_bwd_apply(const float* dh, const float* h, const float* a, const float* stats, const float* gamma, float* da,
                      float* dgamma, float* dbeta, const double* part, int B, int C, int HW, int world, float* amax,
                      hipStream_t st) {
    if (B <= 0 || C <= 0 || HW % 4 || !part || world < 1) return PGAN_EINVAL;
    const int sb = bn_slices(B, C, HW);
    bn_bwd_apply<<<C * sb, BT, 0, st>>>(dh, h, a, stats, gamma, da, dgamma, dbeta, part, B, C, HW, sb, world, amax);
    return pgan_launch_status();
}

}  // extern "C"
