// The position-vector ("1-D") towers and the two plain GEMMs of the generator's first layers.
// Together they are < 0.5 % of the step's FLOPs, so these kernels aim at few launches and exact
// arithmetic rather than peak rate.
//   discriminator 1-D tower  src/layers.py:470-480  Conv1d(k4,bias) -> GroupNorm(1,C) -> LeakyReLU, x3
//   generator 1-D tower      src/layers.py:117-140  ConvTranspose1d(k4,s2,p1) stages
//   generator first layers   src/layers.py:92,117   ConvTranspose on a 1x1 / length-1 input = GEMM
#include "pgan_common.h"

namespace {

constexpr int D1_C0 = 1, D1_C1 = 16, D1_C2 = 32, D1_C3 = 64;
// position vectors of length L0 = 64 (the reference's alignments) or 128 (the scaled network); every k=4 valid
// convolution shortens them by 3

struct D1Params {
    const float* cw[3];
    const float* cb[3];
    const float* gw[3];
    const float* gb[3];
};
struct D1Offsets { int o[12]; };   // cw0 cb0 gw0 gb0 cw1 ... gb2, offsets into a partial-gradient row

// out[co][l] = b[co] + sum_{ci,k} in[ci][l+k] w[co][ci][k]
template <int CI, int CO, int LO>
__device__ __forceinline__ void conv1d_fwd(const float* in, const float* __restrict__ w, const float* __restrict__ b,
                                           float* out) {
    constexpr int LI = LO + 3;
    for (int idx = threadIdx.x; idx < CO * LO; idx += blockDim.x) {
        const int co = idx / LO, l = idx % LO;
        float s = b[co];
        for (int ci = 0; ci < CI; ++ci) {
            const float* wp = w + (co * CI + ci) * 4;
            const float* ip = in + ci * LI + l;
            s = fmaf(ip[0], wp[0], s);
            s = fmaf(ip[1], wp[1], s);
            s = fmaf(ip[2], wp[2], s);
            s = fmaf(ip[3], wp[3], s);
        }
        out[idx] = s;
    }
}

// mean / rstd of a[0:M] (block-wide); every thread returns them
__device__ __forceinline__ void gn_stats(const float* a, int M, double* sh, float& mu, float& r) {
    float s = 0.f, ss = 0.f;
    for (int i = threadIdx.x; i < M; i += blockDim.x) { s += a[i]; ss += a[i] * a[i]; }
    double red[2] = {s, ss};
    block_sum<2>(red, sh);
    const double m = red[0] / M, var = fmax(red[1] / M - m * m, 0.0);
    mu = (float)m;
    r = (float)(1.0 / sqrt(var + (double)PGAN_EPS));
}

template <int D1_L0>
__global__ __launch_bounds__(1024) void d1d_fwd_kernel(const float* __restrict__ xp, D1Params P, float* __restrict__ a1s,
                                                      float* __restrict__ a2s, float* __restrict__ a3s,
                                                      float* __restrict__ feat) {
    constexpr int D1_L1 = D1_L0 - 3, D1_L2 = D1_L0 - 6, D1_L3 = D1_L0 - 9;
    __shared__ double sh[32];
    __shared__ float x0[D1_L0], a1[D1_C1 * D1_L1], a2[D1_C2 * D1_L2], a3[D1_C3 * D1_L3];
    const int n = blockIdx.x, tid = threadIdx.x;
    for (int i = tid; i < D1_L0; i += blockDim.x) x0[i] = xp[(int64_t)n * D1_L0 + i];
    __syncthreads();
    float mu, r;
    conv1d_fwd<D1_C0, D1_C1, D1_L1>(x0, P.cw[0], P.cb[0], a1);
    __syncthreads();
    gn_stats(a1, D1_C1 * D1_L1, sh, mu, r);
    for (int i = tid; i < D1_C1 * D1_L1; i += blockDim.x) {
        const int c = i / D1_L1;
        const float v = a1[i];
        a1s[(int64_t)n * D1_C1 * D1_L1 + i] = v;
        a1[i] = lrelu(P.gw[0][c] * ((v - mu) * r) + P.gb[0][c]);
    }
    __syncthreads();
    conv1d_fwd<D1_C1, D1_C2, D1_L2>(a1, P.cw[1], P.cb[1], a2);
    __syncthreads();
    gn_stats(a2, D1_C2 * D1_L2, sh, mu, r);
    for (int i = tid; i < D1_C2 * D1_L2; i += blockDim.x) {
        const int c = i / D1_L2;
        const float v = a2[i];
        a2s[(int64_t)n * D1_C2 * D1_L2 + i] = v;
        a2[i] = lrelu(P.gw[1][c] * ((v - mu) * r) + P.gb[1][c]);
    }
    __syncthreads();
    conv1d_fwd<D1_C2, D1_C3, D1_L3>(a2, P.cw[2], P.cb[2], a3);
    __syncthreads();
    gn_stats(a3, D1_C3 * D1_L3, sh, mu, r);
    for (int i = tid; i < D1_C3 * D1_L3; i += blockDim.x) {
        const int c = i / D1_L3;
        const float v = a3[i];
        a3s[(int64_t)n * D1_C3 * D1_L3 + i] = v;
        feat[(int64_t)n * D1_C3 * D1_L3 + i] = lrelu(P.gw[2][c] * ((v - mu) * r) + P.gb[2][c]);
    }
}

// GroupNorm+LeakyReLU backward in place: on entry g[i] = d loss / d h, a = conv output; on exit
// g[i] = d loss / d a.  Writes the per-sample gamma/beta gradients when pg != nullptr.
template <int C, int L>
__device__ __forceinline__ void gn_lrelu_bwd(float* g, const float* a, const float* __restrict__ gamma,
                                             const float* __restrict__ beta, float mu, float r, double* sh, float* pg,
                                             float* pb) {
    constexpr int M = C * L;
    float sw = 0.f, swx = 0.f;
    for (int i = threadIdx.x; i < M; i += blockDim.x) {
        const int c = i / L;
        const float xh = (a[i] - mu) * r;
        const float y = gamma[c] * xh + beta[c];
        const float e = (y > 0.f ? 1.f : PGAN_SLOPE) * g[i];
        g[i] = e;
        const float w = gamma[c] * e;
        sw += w;
        swx += w * xh;
    }
    double red[2] = {sw, swx};
    block_sum<2>(red, sh);
    const float mw = (float)(red[0] / M), mwx = (float)(red[1] / M);
    if (pg) {
        for (int c = threadIdx.x; c < C; c += blockDim.x) {
            float sg = 0.f, sb = 0.f;
            for (int l = 0; l < L; ++l) {
                const float xh = (a[c * L + l] - mu) * r;
                sg += g[c * L + l] * xh;
                sb += g[c * L + l];
            }
            pg[c] = sg;
            pb[c] = sb;
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < M; i += blockDim.x) {
        const int c = i / L;
        const float xh = (a[i] - mu) * r;
        g[i] = r * (gamma[c] * g[i] - mw - xh * mwx);
    }
    __syncthreads();
}

// conv1d backward: d = d loss / d out [CO][LO]; in = conv input [CI][LO+3]; w = the layer's weights
// [CO][CI][4] (in LDS).  Weight gradient: one thread per (co, ci) pair keeps the four taps' sums and slides a
// window over `in`, two LDS reads per four multiply-adds.
template <int CI, int CO, int LO>
__device__ __forceinline__ void conv1d_bwd(const float* d, const float* in, const float* w, float* pw,
                                           float* pbias, float* din) {
    constexpr int LI = LO + 3;
    if (pw) {
        for (int pr = threadIdx.x; pr < CO * CI; pr += blockDim.x) {
            const int co = pr / CI, ci = pr % CI;
            const float* dp = d + co * LO;
            const float* ip = in + ci * LI;
            float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
            float x0 = ip[0], x1 = ip[1], x2 = ip[2];
#pragma unroll 5
            for (int l = 0; l < LO; ++l) {
                const float x3 = ip[l + 3], dv = dp[l];
                s0 = fmaf(dv, x0, s0);
                s1 = fmaf(dv, x1, s1);
                s2 = fmaf(dv, x2, s2);
                s3 = fmaf(dv, x3, s3);
                x0 = x1; x1 = x2; x2 = x3;
            }
            *reinterpret_cast<f32x4*>(pw + pr * 4) = f32x4{s0, s1, s2, s3};
        }
        for (int co = threadIdx.x; co < CO; co += blockDim.x) {
            float s = 0.f;
            for (int l = 0; l < LO; ++l) s += d[co * LO + l];
            pbias[co] = s;
        }
    }
    if (din) {
        for (int idx = threadIdx.x; idx < CI * LI; idx += blockDim.x) {
            const int ci = idx / LI, li = idx % LI;
            const bool c0 = li < LO, c1 = li >= 1 && li - 1 < LO, c2 = li >= 2 && li - 2 < LO, c3 = li >= 3;
            float s = 0.f;
#pragma unroll 4
            for (int co = 0; co < CO; ++co) {
                const f32x4 wv = *reinterpret_cast<const f32x4*>(w + (co * CI + ci) * 4);
                // all four reads are made and the ones outside the row dropped by a select: with a branch around each
                // read the loop is a chain of LDS round trips (a read up to three floats outside the array stays inside
                // the block's LDS or returns zero)
                const float* dp = d + co * LO + li;
                const float d0 = dp[0], d1 = dp[-1], d2 = dp[-2], d3 = dp[-3];
                s = fmaf(c0 ? d0 : 0.f, wv[0], s);
                s = fmaf(c1 ? d1 : 0.f, wv[1], s);
                s = fmaf(c2 ? d2 : 0.f, wv[2], s);
                s = fmaf(c3 ? d3 : 0.f, wv[3], s);
            }
            din[idx] = s;
        }
    }
}

// One block per sample, a serial chain of small phases: the kernel is latency-bound and runs beside the 2-D towers'
// convolutions on another stream, so what it holds of a CU matters more than its own speed.  Fully unrolled, its inner
// loops took all 256 registers per lane: with its 104 KB of LDS one block then owned a whole CU and every convolution
// block waited for it.  The loops are unrolled by 4-5 (161 registers, no spills): a third of the register file and
// 56 KB of LDS stay free for the other streams' blocks.
// L0 = 128: the last stage's weights (32 KB) stay in global memory -- with them the kernel's LDS would exceed a CU's.
template <int D1_L0>
__global__ __launch_bounds__(512, 4) void d1d_bwd_kernel(const float* __restrict__ xp, D1Params P,
                                                      const float* __restrict__ a1s, const float* __restrict__ a2s,
                                                      const float* __restrict__ a3s, const float* __restrict__ coef,
                                                      const float* __restrict__ w_tail, float* __restrict__ part,
                                                      int part_ld, D1Offsets off, float* __restrict__ dxp) {
    constexpr int D1_L1 = D1_L0 - 3, D1_L2 = D1_L0 - 6, D1_L3 = D1_L0 - 9;
    constexpr bool W3_LDS = D1_L0 <= 64;
    __shared__ double sh[32];
    __shared__ float x0[D1_L0], a1[D1_C1 * D1_L1], h1[D1_C1 * D1_L1], a2[D1_C2 * D1_L2], h2[D1_C2 * D1_L2],
        a3[D1_C3 * D1_L3], g3[D1_C3 * D1_L3], g2[D1_C2 * D1_L2], g1[D1_C1 * D1_L1], g0[D1_L0];
    __shared__ __attribute__((aligned(16))) float w1s[D1_C1 * D1_C0 * 4], w2s[D1_C2 * D1_C1 * 4],
        w3buf[W3_LDS ? D1_C3 * D1_C2 * 4 : 4];
    const int n = blockIdx.x, tid = threadIdx.x;
    float* prow = part ? part + (int64_t)n * part_ld : nullptr;
    for (int i = tid; i < D1_C1 * D1_C0 * 4; i += blockDim.x) w1s[i] = P.cw[0][i];
    for (int i = tid; i < D1_C2 * D1_C1 * 4; i += blockDim.x) w2s[i] = P.cw[1][i];
    if constexpr (W3_LDS)
        for (int i = tid; i < D1_C3 * D1_C2 * 4; i += blockDim.x) w3buf[i] = P.cw[2][i];
    const float* w3s = W3_LDS ? w3buf : P.cw[2];
    for (int i = tid; i < D1_L0; i += blockDim.x) x0[i] = xp[(int64_t)n * D1_L0 + i];
    for (int i = tid; i < D1_C1 * D1_L1; i += blockDim.x) a1[i] = a1s[(int64_t)n * D1_C1 * D1_L1 + i];
    for (int i = tid; i < D1_C2 * D1_L2; i += blockDim.x) a2[i] = a2s[(int64_t)n * D1_C2 * D1_L2 + i];
    const float cf = coef[n];
    for (int i = tid; i < D1_C3 * D1_L3; i += blockDim.x) {
        a3[i] = a3s[(int64_t)n * D1_C3 * D1_L3 + i];
        g3[i] = cf * w_tail[i];
    }
    __syncthreads();
    float mu1, r1, mu2, r2, mu3, r3;
    gn_stats(a1, D1_C1 * D1_L1, sh, mu1, r1);
    gn_stats(a2, D1_C2 * D1_L2, sh, mu2, r2);
    gn_stats(a3, D1_C3 * D1_L3, sh, mu3, r3);
    for (int i = tid; i < D1_C1 * D1_L1; i += blockDim.x) {
        const int c = i / D1_L1;
        h1[i] = lrelu(P.gw[0][c] * ((a1[i] - mu1) * r1) + P.gb[0][c]);
    }
    for (int i = tid; i < D1_C2 * D1_L2; i += blockDim.x) {
        const int c = i / D1_L2;
        h2[i] = lrelu(P.gw[1][c] * ((a2[i] - mu2) * r2) + P.gb[1][c]);
    }
    __syncthreads();
    gn_lrelu_bwd<D1_C3, D1_L3>(g3, a3, P.gw[2], P.gb[2], mu3, r3, sh, prow ? prow + off.o[10] : nullptr,
                               prow ? prow + off.o[11] : nullptr);
    conv1d_bwd<D1_C2, D1_C3, D1_L3>(g3, h2, w3s, prow ? prow + off.o[8] : nullptr, prow ? prow + off.o[9] : nullptr, g2);
    __syncthreads();
    gn_lrelu_bwd<D1_C2, D1_L2>(g2, a2, P.gw[1], P.gb[1], mu2, r2, sh, prow ? prow + off.o[6] : nullptr,
                               prow ? prow + off.o[7] : nullptr);
    conv1d_bwd<D1_C1, D1_C2, D1_L2>(g2, h1, w2s, prow ? prow + off.o[4] : nullptr, prow ? prow + off.o[5] : nullptr, g1);
    __syncthreads();
    gn_lrelu_bwd<D1_C1, D1_L1>(g1, a1, P.gw[0], P.gb[0], mu1, r1, sh, prow ? prow + off.o[2] : nullptr,
                               prow ? prow + off.o[3] : nullptr);
    conv1d_bwd<D1_C0, D1_C1, D1_L1>(g1, x0, w1s, prow ? prow + off.o[0] : nullptr, prow ? prow + off.o[1] : nullptr,
                                    dxp ? g0 : nullptr);
    __syncthreads();
    if (dxp)
        for (int i = tid; i < D1_L0; i += blockDim.x) dxp[(int64_t)n * D1_L0 + i] = g0[i];
}

// ---------------- generator 1-D tower: ConvTranspose1d(k4, s2, p1) pair, weight w[Cs][Cl][4] ----------------
// up:    x[n][cl][i] = sum_{cs,k : i = 2o-1+k} y[n][cs][o] w[cs][cl][k]   (+ optional sigmoid/tanh)
__global__ void ct1d_up_kernel(const float* __restrict__ y, const float* __restrict__ w, float* __restrict__ x, int N,
                               int Cs, int Cl, int Ls, int act) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int Ll = 2 * Ls;
    if (idx >= (int64_t)N * Cl * Ll) return;
    const int i = (int)(idx % Ll), cl = (int)((idx / Ll) % Cl), n = (int)(idx / ((int64_t)Ll * Cl));
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int o2 = i + 1 - k;
        if (o2 < 0 || (o2 & 1) || o2 / 2 >= Ls) continue;
        const int o = o2 / 2;
#pragma unroll 8
        for (int cs = 0; cs < Cs; ++cs) s = fmaf(y[((int64_t)n * Cs + cs) * Ls + o], w[(cs * Cl + cl) * 4 + k], s);
    }
    if (act == 1) s = 1.f / (1.f + expf(-s));
    else if (act == 2) s = tanhf(s);
    x[idx] = s;
}
// down:  y[n][cs][o] = sum_{cl,k} x[n][cl][2o-1+k] w[cs][cl][k]
__global__ void ct1d_down_kernel(const float* __restrict__ x, const float* __restrict__ w, float* __restrict__ y, int N,
                                 int Cs, int Cl, int Ls) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int Ll = 2 * Ls;
    if (idx >= (int64_t)N * Cs * Ls) return;
    const int o = (int)(idx % Ls), cs = (int)((idx / Ls) % Cs), n = (int)(idx / ((int64_t)Ls * Cs));
    float s = 0.f;
#pragma unroll 4
    for (int cl = 0; cl < Cl; ++cl) {
        const float* xp = x + ((int64_t)n * Cl + cl) * Ll;
        const float* wp = w + (cs * Cl + cl) * 4;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = 2 * o - 1 + k;
            if (i >= 0 && i < Ll) s = fmaf(xp[i], wp[k], s);
        }
    }
    y[idx] = s;
}
// wgrad: dw[cs][cl][k] = sum_{n,o} ya[n][cs][o] x[n][cl][2o-1+k]; one wave per output element
__global__ __launch_bounds__(256) void ct1d_wgrad_kernel(const float* __restrict__ ya, const float* __restrict__ x,
                                                         float* __restrict__ dw, int N, int Cs, int Cl, int Ls) {
    const int out = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int Ll = 2 * Ls;
    if (out >= Cs * Cl * 4) return;
    const int k = out % 4, cl = (out / 4) % Cl, cs = out / (4 * Cl);
    float s = 0.f;
    for (int e = lane; e < N * Ls; e += 64) {
        const int n = e / Ls, o = e % Ls, i = 2 * o - 1 + k;
        if (i >= 0 && i < Ll) s = fmaf(ya[((int64_t)n * Cs + cs) * Ls + o], x[((int64_t)n * Cl + cl) * Ll + i], s);
    }
    s = wave_sum(s);
    if (lane == 0) dw[out] = s;
}

// ---------------- plain fp32 GEMM, 64x64 block tile, 4x4 per thread ----------------
// C[M,N] = op(A) B with B[K,N] row-major; TRANSA = false: A[M,K]; true: A[K,M].
template <bool TRANSA>
__global__ __launch_bounds__(256) void gemm_kernel(const float* __restrict__ A, const float* __restrict__ Bm,
                                                   float* __restrict__ C, int M, int N, int K) {
    __shared__ float As[16][64 + 1], Bs[16][64 + 4];
    const int tid = threadIdx.x, tx = tid % 16, ty = tid / 16;
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
    float acc[4][4] = {};
    for (int k0 = 0; k0 < K; k0 += 16) {
        for (int e = tid; e < 16 * 64; e += 256) {
            int kk, mm;
            if (TRANSA) { kk = e / 64; mm = e % 64; } else { mm = e / 16; kk = e % 16; }
            const int gm = m0 + mm, gk = k0 + kk;
            float v = 0.f;
            if (gm < M && gk < K) v = TRANSA ? A[(int64_t)gk * M + gm] : A[(int64_t)gm * K + gk];
            As[kk][mm] = v;
        }
        for (int e = tid; e < 16 * 64; e += 256) {
            const int kk = e / 64, nn = e % 64, gk = k0 + kk, gn = n0 + nn;
            Bs[kk][nn] = (gk < K && gn < N) ? Bm[(int64_t)gk * N + gn] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            float a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = As[kk][ty * 4 + i];
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = Bs[kk][tx * 4 + j];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(a[i], b[j], acc[i][j]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int gm = m0 + ty * 4 + i, gn = n0 + tx * 4 + j;
            if (gm < M && gn < N) C[(int64_t)gm * N + gn] = acc[i][j];
        }
}

// C[M,N] = A[M,K] B[K,N] for short, wide products (the generators' first layers: M = batch, K = latent):
// one wave per 32x32 tile of C on the fp32 matrix pipe, operands straight from global memory (A is tiny
// and stays in cache, B rows are read coalesced).  M % 32 == N % 32 == K % 8 == 0.
__global__ __launch_bounds__(256) void gemm_nn_mfma_kernel(const float* __restrict__ A, const float* __restrict__ Bm,
                                                           float* __restrict__ C, int M, int N, int K) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 31, h = lane >> 5;
    const int tile = blockIdx.x * 4 + wave, ntn = N / 32;
    if (tile >= (M / 32) * ntn) return;
    const int m0 = (tile / ntn) * 32, n0 = (tile % ntn) * 32;
    const float* a = A + (int64_t)(m0 + j) * K + h;
    const float* b = Bm + (int64_t)h * N + n0 + j;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int k = 0; k < K; k += 8) {
        float av[4], bv[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) { av[q] = a[k + 2 * q]; bv[q] = b[(int64_t)(k + 2 * q) * N]; }
#pragma unroll
        for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q], bv[q], acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r)
        C[(int64_t)(m0 + (r & 3) + 8 * (r >> 2) + 4 * h) * N + n0 + j] = acc[r];
}

D1Params make_params(const float* const* p) {
    D1Params P;
    for (int i = 0; i < 3; ++i) {
        P.cw[i] = p[4 * i + 0];
        P.cb[i] = p[4 * i + 1];
        P.gw[i] = p[4 * i + 2];
        P.gb[i] = p[4 * i + 3];
    }
    return P;
}

}  // namespace

extern "C" {

// params: 12 device pointers in the order cw0 cb0 gw0 gb0 cw1 cb1 gw1 gb1 cw2 cb2 gw2 gb2.
// xp [N,1,L] -> conv outputs a1 [N,16,L-3], a2 [N,32,L-6], a3 [N,64,L-9] (saved for backward) and
// feat [N, 64*(L-9)]; L = 64 or 128.
int pgan_d1d_fwd_len(const float* xp, const float* const* params, float* a1, float* a2, float* a3, float* feat, int N,
                     int L, hipStream_t st) {
    if (N <= 0 || (L != 64 && L != 128)) return PGAN_EINVAL;
    if (L == 64) d1d_fwd_kernel<64><<<N, 1024, 0, st>>>(xp, make_params(params), a1, a2, a3, feat);
    else d1d_fwd_kernel<128><<<N, 1024, 0, st>>>(xp, make_params(params), a1, a2, a3, feat);
    return pgan_launch_status();
}
int pgan_d1d_fwd(const float* xp, const float* const* params, float* a1, float* a2, float* a3, float* feat, int N,
                 hipStream_t st) {
    return pgan_d1d_fwd_len(xp, params, a1, a2, a3, feat, N, 64, st);
}

// Backward of sum_n coef[n] <w_tail, feat[n]>.  part (may be null): [N, part_ld] per-sample parameter
// gradient rows, the 12 tensors placed at offs[0..11] (same order as params); dxp (may be null): [N,1,L].
int pgan_d1d_bwd_len(const float* xp, const float* const* params, const float* a1, const float* a2, const float* a3,
                     const float* coef, const float* w_tail, float* part, int part_ld, const int* offs, float* dxp,
                     int N, int L, hipStream_t st) {
    if (N <= 0 || (part && !offs) || (L != 64 && L != 128)) return PGAN_EINVAL;
    D1Offsets o;
    for (int i = 0; i < 12; ++i) o.o[i] = offs ? offs[i] : 0;
    if (L == 64)
        d1d_bwd_kernel<64><<<N, 512, 0, st>>>(xp, make_params(params), a1, a2, a3, coef, w_tail, part, part_ld, o, dxp);
    else
        d1d_bwd_kernel<128><<<N, 512, 0, st>>>(xp, make_params(params), a1, a2, a3, coef, w_tail, part, part_ld, o, dxp);
    return pgan_launch_status();
}
int pgan_d1d_bwd(const float* xp, const float* const* params, const float* a1, const float* a2, const float* a3,
                 const float* coef, const float* w_tail, float* part, int part_ld, const int* offs, float* dxp, int N,
                 hipStream_t st) {
    return pgan_d1d_bwd_len(xp, params, a1, a2, a3, coef, w_tail, part, part_ld, offs, dxp, N, 64, st);
}

int pgan_ct1d_up(const float* y, const float* w, float* x, int N, int Cs, int Cl, int Ls, int act, hipStream_t st) {
    if (N <= 0) return PGAN_EINVAL;
    const long long n = (long long)N * Cl * 2 * Ls;
    ct1d_up_kernel<<<(int)((n + 255) / 256), 256, 0, st>>>(y, w, x, N, Cs, Cl, Ls, act);
    return pgan_launch_status();
}

int pgan_ct1d_down(const float* x, const float* w, float* y, int N, int Cs, int Cl, int Ls, hipStream_t st) {
    if (N <= 0) return PGAN_EINVAL;
    const long long n = (long long)N * Cs * Ls;
    ct1d_down_kernel<<<(int)((n + 255) / 256), 256, 0, st>>>(x, w, y, N, Cs, Cl, Ls);
    return pgan_launch_status();
}

int pgan_ct1d_wgrad(const float* ya, const float* x, float* dw, int N, int Cs, int Cl, int Ls, hipStream_t st) {
    if (N <= 0) return PGAN_EINVAL;
    ct1d_wgrad_kernel<<<(Cs * Cl * 4 + 3) / 4, 256, 0, st>>>(ya, x, dw, N, Cs, Cl, Ls);
    return pgan_launch_status();
}

// C[M,N] = A[M,K] B[K,N]  (transa = 0)  or  A[K,M]^T B[K,N]  (transa = 1); all row-major fp32.
int pgan_gemm(const float* A, const float* B, float* C, int M, int N, int K, int transa, hipStream_t st) {
    if (M <= 0 || N <= 0 || K <= 0) return PGAN_EINVAL;
    if (!transa && M % 32 == 0 && N % 32 == 0 && K % 8 == 0 && M <= 256) {
        const int tiles = (M / 32) * (N / 32);
        gemm_nn_mfma_kernel<<<(tiles + 3) / 4, 256, 0, st>>>(A, B, C, M, N, K);
        return pgan_launch_status();
    }
    dim3 grid((N + 63) / 64, (M + 63) / 64);
    if (transa) gemm_kernel<true><<<grid, 256, 0, st>>>(A, B, C, M, N, K);
    else gemm_kernel<false><<<grid, 256, 0, st>>>(A, B, C, M, N, K);
    return pgan_launch_status();
}

}  // extern "C"
