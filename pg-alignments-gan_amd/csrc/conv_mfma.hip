// fp32 MFMA implicit-GEMM kernels for the k=4, stride=2, pad=1 convolution pair that carries
// 98 % of the DCWGAN-GP step's FLOPs (gfx950, v_mfma_f32_32x32x2_f32).
//
// One weight tensor W[Cs][Cl][4][4] links a SMALL side S[n][Cs][H][H] and a LARGE side
// L[n][Cl][2H][2H]:
//   DOWN  (L -> S)  S[cs][oh][ow] = sum_{cl,kh,kw} L[cl][2oh-1+kh][2ow-1+kw] W[cs][cl][kh][kw]
//                   = Conv2d fprop of the discriminator  (reference src/layers.py:441-463)
//                   = ConvTranspose2d input-gradient of the generator (src/layers.py:97-112)
//   UP    (S -> L)  L[cl][ih][iw] = sum_{cs,kh,kw : ih = 2oh-1+kh} S[cs][oh][ow] W[cs][cl][kh][kw]
//                   = Conv2d input-gradient of the discriminator = ConvTranspose2d fprop of the generator
//   WGRAD           dW[cs][cl][kh][kw] = sum_{n,oh,ow} S[n][cs][oh][ow] L[r(n)][cl][2oh-1+kh][2ow-1+kw]
// All tensors fp32, NCHW contiguous.  MFMA operand maps (32x32x2): lane l supplies
// A[i = l&31][k = l>>5] and B[k = l>>5][j = l&31]; result register r of lane l is
// D[i = (r&3) + 8(r>>2) + 4(l>>5)][j = l&31].
#include "conv_tiles.h"

namespace {
using namespace pgan;

// ------------------------------------------------------------------------------------------------
// DOWN: GEMM M = Cs (BM per block), N = pixels (BN per block), K = Cl*16 (CLC channels per step)
// ------------------------------------------------------------------------------------------------
template <int HS, int BM, int BN, int CLC, int KG>
__global__ __launch_bounds__(256 * KG) void conv_down_kernel(const float* __restrict__ X, const float* __restrict__ W,
                                                             float* __restrict__ Y, int N, int Cs, int Cl) {
    using T = PixTile<HS, BN>;
    constexpr int WS = HS, HL = 2 * HS, WL = 2 * HS;
    constexpr int SEG = T::SEG, R = T::R;
    constexpr int PR = 2 * R + 2, PC = 2 * WS + 2, PCP = 3 * WS;
    constexpr int CHS = SEG * PR * PCP;
    constexpr int ALD = CLC * 16 + 4;
    constexpr int TM = BM / 64, TN = BN / 64;
    constexpr int PE = SEG * PR * PC, NQ = (PE + 255) / 256;
    constexpr int AV = BM * CLC * 4 / 256;
    constexpr int GRP = BM * ALD + CLC * CHS;            // LDS floats of one K group
    static_assert(BM % 64 == 0 && BN % 64 == 0 && AV >= 1, "tile shape");
    static_assert(KG == 1 || GRP * KG >= (KG - 1) * 4 * TM * TN * 16 * 64, "reduction scratch must fit");
    __shared__ __attribute__((aligned(16))) float lds[GRP * KG];
    // K is split over KG groups of 4 waves; each group stages and multiplies its own channels and
    // the partial tiles are summed through LDS at the end.
    const int grp = threadIdx.x >> 8;
    float* As = lds + grp * GRP;
    float* Bs = As + BM * ALD;

    const int tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, j = lane & 31, h = lane >> 5;
    const int nmt = Cs / BM, npt = T::ntiles(N);
    const int b = xcd_remap(blockIdx.x, nmt * npt);
    const int ptile = b / nmt, cs0 = (b % nmt) * BM;
    int n0, oh0;
    T::origin(ptile, n0, oh0);

    // per-thread patch elements (same geometry for every channel)
    int64_t poff[NQ];
    int loff[NQ];
    bool pval[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int e = tid + 256 * q;
        const int seg = e / (PR * PC), rem = e % (PR * PC), pr = rem / PC, pc = rem % PC;
        const int n = n0 + seg, ih = 2 * oh0 - 1 + pr, iw = pc - 1;
        pval[q] = (e < PE) && n < N && ih >= 0 && ih < HL && iw >= 0 && iw < WL;
        poff[q] = ((int64_t)n * Cl * HL + ih) * WL + iw;
        loff[q] = (e < PE) ? (seg * PR + pr) * PCP + pc : -1;
    }
    f32x4 areg[AV];
    float breg[CLC][NQ];
    auto gload = [&](int cl0) {
#pragma unroll
        for (int q = 0; q < AV; ++q) {
            const int idx = tid + 256 * q, row = idx / (CLC * 4), c4 = idx % (CLC * 4);
            areg[q] = *reinterpret_cast<const f32x4*>(W + ((int64_t)(cs0 + row) * Cl + cl0) * 16 + c4 * 4);
        }
#pragma unroll
        for (int c = 0; c < CLC; ++c)
#pragma unroll
            for (int q = 0; q < NQ; ++q)
                breg[c][q] = pval[q] ? X[poff[q] + (int64_t)(cl0 + c) * HL * WL] : 0.f;
    };
    auto lstore = [&]() {
#pragma unroll
        for (int q = 0; q < AV; ++q) {
            const int idx = tid + 256 * q, row = idx / (CLC * 4), c4 = idx % (CLC * 4);
            *reinterpret_cast<f32x4*>(As + row * ALD + c4 * 4) = areg[q];
        }
#pragma unroll
        for (int c = 0; c < CLC; ++c)
#pragma unroll
            for (int q = 0; q < NQ; ++q)
                if (loff[q] >= 0) Bs[c * CHS + loff[q]] = breg[c][q];
    };

    int abase[TM], bbase[TN];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) abase[tm] = (wm * (BM / 2) + tm * 32 + j) * ALD + h * 4;
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
        const int pt = wn * (BN / 2) + tn * 32 + j;
        const int seg = pt / (R * WS), rem = pt % (R * WS), ohl = rem / WS, ow = rem % WS;
        bbase[tn] = (seg * PR + 2 * ohl + h) * PCP + 2 * ow;
    }
    f32x16 acc[TM][TN];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[tm][tn][r] = 0.f;

    // gridDim.y > 1: the channel range is also split across blocks, which then add into a zeroed Y
    const int nk = Cl / (CLC * KG * (int)gridDim.y);
    const int kc0 = blockIdx.y * nk;
    gload((kc0 * KG + grp) * CLC);
    for (int kc = 0; kc < nk; ++kc) {
        lstore();
        __syncthreads();
        if (kc + 1 < nk) gload(((kc0 + kc + 1) * KG + grp) * CLC);
#pragma unroll
        for (int c = 0; c < CLC; ++c)
#pragma unroll
            for (int khp = 0; khp < 2; ++khp) {
                f32x4 a[TM];
                float2 b0[TN], b1[TN];
#pragma unroll
                for (int tm = 0; tm < TM; ++tm)
                    a[tm] = *reinterpret_cast<const f32x4*>(As + abase[tm] + c * 16 + khp * 8);
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) {
                    const float* p = Bs + c * CHS + bbase[tn] + khp * 2 * PCP;
                    b0[tn] = *reinterpret_cast<const float2*>(p);
                    b1[tn] = *reinterpret_cast<const float2*>(p + 2);
                }
#pragma unroll
                for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn) {
                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tm][0], b0[tn].x, acc[tm][tn], 0, 0, 0);
                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tm][1], b0[tn].y, acc[tm][tn], 0, 0, 0);
                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tm][2], b1[tn].x, acc[tm][tn], 0, 0, 0);
                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tm][3], b1[tn].y, acc[tm][tn], 0, 0, 0);
                    }
            }
        __syncthreads();
    }

    if (KG > 1) {   // sum the K groups' partial tiles: groups 1.. park theirs in LDS, group 0 adds
        float* red = lds + ((grp - 1) * 4 + wave) * (TM * TN * 16 * 64) + lane;
        if (grp > 0) {
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                    for (int r = 0; r < 16; ++r) red[((tm * TN + tn) * 16 + r) * 64] = acc[tm][tn][r];
        }
        __syncthreads();
        if (grp > 0) return;
#pragma unroll
        for (int g = 1; g < KG; ++g) {
            const float* src = lds + ((g - 1) * 4 + wave) * (TM * TN * 16 * 64) + lane;
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[tm][tn][r] += src[((tm * TN + tn) * 16 + r) * 64];
        }
    }

#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
        const int pt = wn * (BN / 2) + tn * 32 + j;
        const int seg = pt / (R * WS), rem = pt % (R * WS), ohl = rem / WS, ow = rem % WS;
        const int n = n0 + seg;
        if (n >= N) continue;
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int cs = cs0 + wm * (BM / 2) + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                float* dst = Y + (((int64_t)n * Cs + cs) * HS + oh0 + ohl) * WS + ow;
                if (gridDim.y > 1) atomicAdd(dst, acc[tm][tn][r]);
                else *dst = acc[tm][tn][r];
            }
    }
}

// ------------------------------------------------------------------------------------------------
// DOWN, software-pipelined form.  A 32x32x2 fp32 MFMA occupies the matrix pipe for 64 cycles, so
// everything else a wave has to do fits in its shadow if it is issued between the MFMAs instead of
// in a phase of its own: LDS is double-buffered (one barrier per K chunk), the operands of step
// s+1 are read before the MFMAs of step s issue, chunk k+1 goes from registers to the other LDS
// buffer a few stores per step, and chunk k+2 is in flight from global memory into a second
// register set for a whole chunk.  Halo / out-of-range elements load from a valid address and are
// zeroed by a select, so the staging code has no branches.
// ------------------------------------------------------------------------------------------------
template <int HS, int BM, int BN, int CLC>
__global__ __launch_bounds__(256) void conv_down_pipe_kernel(const float* __restrict__ X, const float* __restrict__ W,
                                                             float* __restrict__ Y, int N, int Cs, int Cl) {
    using T = PixTile<HS, BN>;
    constexpr int WS = HS, HL = 2 * HS, WL = 2 * HS;
    constexpr int SEG = T::SEG, R = T::R;
    constexpr int PR = 2 * R + 2, PC = 2 * WS + 2, PCP = 3 * WS;    // columns PC.. of a patch row are padding
    constexpr int CHS = SEG * PR * PCP;
    constexpr int ALD = CLC * 16 + 4;
    constexpr int TM = BM / 64, TN = BN / 64;
    constexpr int PE = SEG * PR * PC, NQ = (PE + 255) / 256;
    constexpr int AV = BM * CLC * 4 / 256;
    constexpr int BUF = BM * ALD + CLC * CHS;
    constexpr int S = CLC * 2;                 // MFMA steps per chunk: (channel, tap-row pair)
    constexpr int NST = AV + CLC * NQ;         // staging stores per thread per chunk
    static_assert(BM % 64 == 0 && BN % 64 == 0 && AV >= 1 && PCP > PC, "tile shape");
    __shared__ __attribute__((aligned(16))) float lds[2 * BUF];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, j = lane & 31, h = lane >> 5;
    const int nmt = Cs / BM, npt = T::ntiles(N);
    const int b = xcd_remap(blockIdx.x, nmt * npt);
    const int ptile = b / nmt, cs0 = (b % nmt) * BM;
    int n0, oh0;
    T::origin(ptile, n0, oh0);

    uint32_t poff[NQ];
    int loff[NQ];
    bool pval[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int e = tid + 256 * q;
        const int seg = e / (PR * PC), rem = e % (PR * PC), pr = rem / PC, pc = rem % PC;
        const int n = n0 + seg, ih = 2 * oh0 - 1 + pr, iw = pc - 1;
        pval[q] = (e < PE) && n < N && ih >= 0 && ih < HL && iw >= 0 && iw < WL;
        poff[q] = pval[q] ? (uint32_t)(((n * Cl * HL + ih) * WL) + iw) : 0u;
        loff[q] = (e < PE) ? (seg * PR + pr) * PCP + pc : PC;     // threads past the patch write a padding slot
    }
    uint32_t aoff[AV];
    int alds[AV];
#pragma unroll
    for (int q = 0; q < AV; ++q) {
        const int idx = tid + 256 * q, row = idx / (CLC * 4), c4 = idx % (CLC * 4);
        aoff[q] = (uint32_t)((cs0 + row) * Cl * 16 + c4 * 4);
        alds[q] = row * ALD + c4 * 4;
    }
    f32x4 areg[2][AV];
    float breg[2][CLC][NQ];
    auto gload = [&](auto setc, int cl0) {
        constexpr int ST = decltype(setc)::value;
        const float* Wc = W + (int64_t)cl0 * 16;
#pragma unroll
        for (int q = 0; q < AV; ++q) areg[ST][q] = *reinterpret_cast<const f32x4*>(Wc + aoff[q]);
#pragma unroll
        for (int c = 0; c < CLC; ++c) {
            const float* Xc = X + (int64_t)(cl0 + c) * HL * WL;
#pragma unroll
            for (int q = 0; q < NQ; ++q) breg[ST][c][q] = Xc[poff[q]];
        }
    };
    // staging store number it (0 .. NST-1) of register set ST into the buffer at An / Bn
    auto stage = [&](auto setc, auto itc, float* An, float* Bn) {
        constexpr int ST = decltype(setc)::value, it = decltype(itc)::value;
        if constexpr (it < AV) {
            *reinterpret_cast<f32x4*>(An + alds[it]) = areg[ST][it];
        } else {
            constexpr int c = (it - AV) / NQ, q = (it - AV) % NQ;
            Bn[c * CHS + loff[q]] = pval[q] ? breg[ST][c][q] : 0.f;
        }
    };

    int abase[TM], bbase[TN];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) abase[tm] = (wm * (BM / 2) + tm * 32 + j) * ALD + h * 4;
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
        const int pt = wn * (BN / 2) + tn * 32 + j;
        const int seg = pt / (R * WS), rem = pt % (R * WS), ohl = rem / WS, ow = rem % WS;
        bbase[tn] = (seg * PR + 2 * ohl + h) * PCP + 2 * ow;
    }
    f32x16 acc[TM][TN];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[tm][tn][r] = 0.f;

    f32x4 a[2][TM];
    float2 b0[2][TN], b1[2][TN];
    auto ldops = [&](const float* Ac, const float* Bc, auto sc) {
        constexpr int s = decltype(sc)::value, c = s >> 1, khp = s & 1, set = s & 1;
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) a[set][tm] = *reinterpret_cast<const f32x4*>(Ac + abase[tm] + c * 16 + khp * 8);
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
            const float* p = Bc + c * CHS + bbase[tn] + khp * 2 * PCP;
            b0[set][tn] = *reinterpret_cast<const float2*>(p);
            b1[set][tn] = *reinterpret_cast<const float2*>(p + 2);
        }
    };

    // gridDim.y > 1: the channel range is also split across blocks, which then add into a zeroed Y
    const int nk = Cl / (CLC * (int)gridDim.y);       // even (checked by the launcher)
    const int kc0 = blockIdx.y * nk;
    auto chunk_cl = [&](int kc) { return (kc0 + (kc < nk ? kc : nk - 1)) * CLC; };

    auto step_loop = [&](auto setc, const float* Ac, const float* Bc, float* An, float* Bn) {
        ldops(Ac, Bc, IC<0>{});
        [&]<int... Ss>(std::integer_sequence<int, Ss...>) {
            ([&] {
                constexpr int s = Ss, set = s & 1;
                if constexpr (s + 1 < S) ldops(Ac, Bc, IC<s + 1>{});
#pragma unroll
                for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn) {
                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[set][tm][0], b0[set][tn].x, acc[tm][tn], 0, 0, 0);
                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[set][tm][1], b0[set][tn].y, acc[tm][tn], 0, 0, 0);
                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[set][tm][2], b1[set][tn].x, acc[tm][tn], 0, 0, 0);
                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[set][tm][3], b1[set][tn].y, acc[tm][tn], 0, 0, 0);
                    }
                // this step's share of the staging stores
                constexpr int lo = s * NST / S, hi = (s + 1) * NST / S;
                [&]<int... Is>(std::integer_sequence<int, Is...>) {
                    (stage(setc, IC<lo + Is>{}, An, Bn), ...);
                }(std::make_integer_sequence<int, hi - lo>{});
            }(), ...);
        }(std::make_integer_sequence<int, S>{});
    };

    // prologue: chunk 0 into buffer 0, chunk 1 into register set 0
    gload(IC<0>{}, chunk_cl(0));
    [&]<int... Is>(std::integer_sequence<int, Is...>) {
        (stage(IC<0>{}, IC<Is>{}, lds, lds + BM * ALD), ...);
    }(std::make_integer_sequence<int, NST>{});
    gload(IC<0>{}, chunk_cl(1));
    __syncthreads();
    for (int kc = 0; kc < nk; kc += 2) {
        // even chunk: multiply buffer 0; set 0 (chunk kc+1) -> buffer 1; chunk kc+2 -> set 1
        gload(IC<1>{}, chunk_cl(kc + 2));
        __builtin_amdgcn_sched_barrier(0);      // keep the loads ahead of the chunk's MFMAs
        step_loop(IC<0>{}, lds, lds + BM * ALD, lds + BUF, lds + BUF + BM * ALD);
        __syncthreads();
        gload(IC<0>{}, chunk_cl(kc + 3));
        __builtin_amdgcn_sched_barrier(0);
        step_loop(IC<1>{}, lds + BUF, lds + BUF + BM * ALD, lds, lds + BM * ALD);
        __syncthreads();
    }

#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
        const int pt = wn * (BN / 2) + tn * 32 + j;
        const int seg = pt / (R * WS), rem = pt % (R * WS), ohl = rem / WS, ow = rem % WS;
        const int n = n0 + seg;
        if (n >= N) continue;
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int cs = cs0 + wm * (BM / 2) + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                float* dst = Y + (((int64_t)n * Cs + cs) * HS + oh0 + ohl) * WS + ow;
                if (gridDim.y > 1) atomicAdd(dst, acc[tm][tn][r]);
                else *dst = acc[tm][tn][r];
            }
    }
}

// ------------------------------------------------------------------------------------------------
// UP: four sub-pixel phases, each a GEMM M = Cl (BM), N = S-grid pixels (BN), K = Cs*4.
// Tap kh feeds output-row phase PH[kh] from S row a + DH[kh]:  kh 0:(1,+1) 1:(0,0) 2:(1,0) 3:(0,-1).
// ------------------------------------------------------------------------------------------------
template <int HS, int BM, int BN, int CSC>
__global__ __launch_bounds__(256) void conv_up_kernel(const float* __restrict__ Y, const float* __restrict__ W,
                                                      float* __restrict__ X, int N, int Cs, int Cl) {
    using T = PixTile<HS, BN>;
    constexpr int WS = HS, HL = 2 * HS, WL = 2 * HS;
    constexpr int SEG = T::SEG, R = T::R;
    constexpr int PR = R + 2, PC = WS + 2, PCP = PC | 1;
    constexpr int CHS = SEG * PR * PCP;
    constexpr int ALD = BM + 1;
    constexpr int TM = BM / 64, TN = BN / 64;
    constexpr int PE = SEG * PR * PC, NQ = (PE + 255) / 256;
    constexpr int AV = CSC * BM * 4 / 256;
    static_assert(BM % 64 == 0 && BN % 64 == 0 && CSC % 2 == 0 && AV >= 1, "tile shape");
    __shared__ __attribute__((aligned(16))) float lds[CSC * 16 * ALD + CSC * CHS];
    float* As = lds;
    float* Bs = lds + CSC * 16 * ALD;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, j = lane & 31, h = lane >> 5;
    const int nmt = Cl / BM, npt = T::ntiles(N);
    const int b = xcd_remap(blockIdx.x, nmt * npt);
    const int ptile = b / nmt, cl0 = (b % nmt) * BM;
    int n0, oh0;
    T::origin(ptile, n0, oh0);

    int64_t poff[NQ];
    int loff[NQ];
    bool pval[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int e = tid + 256 * q;
        const int seg = e / (PR * PC), rem = e % (PR * PC), pr = rem / PC, pc = rem % PC;
        const int n = n0 + seg, a = oh0 - 1 + pr, bb = pc - 1;
        pval[q] = (e < PE) && n < N && a >= 0 && a < HS && bb >= 0 && bb < WS;
        poff[q] = ((int64_t)n * Cs * HS + a) * WS + bb;
        loff[q] = (e < PE) ? (seg * PR + pr) * PCP + pc : -1;
    }
    f32x4 areg[AV];
    float breg[CSC][NQ];
    auto gload = [&](int cs0) {
#pragma unroll
        for (int q = 0; q < AV; ++q) {
            const int idx = tid + 256 * q, csl = idx / (BM * 4), rem = idx % (BM * 4);
            areg[q] = *reinterpret_cast<const f32x4*>(W + ((int64_t)(cs0 + csl) * Cl + cl0) * 16 + rem * 4);
        }
#pragma unroll
        for (int c = 0; c < CSC; ++c)
#pragma unroll
            for (int q = 0; q < NQ; ++q)
                breg[c][q] = pval[q] ? Y[poff[q] + (int64_t)(cs0 + c) * HS * WS] : 0.f;
    };
    auto lstore = [&]() {
#pragma unroll
        for (int q = 0; q < AV; ++q) {
            const int idx = tid + 256 * q, csl = idx / (BM * 4), rem = idx % (BM * 4);
            const int cll = rem / 4, t4 = rem % 4;
#pragma unroll
            for (int e = 0; e < 4; ++e) As[(csl * 16 + t4 * 4 + e) * ALD + cll] = areg[q][e];
        }
#pragma unroll
        for (int c = 0; c < CSC; ++c)
#pragma unroll
            for (int q = 0; q < NQ; ++q)
                if (loff[q] >= 0) Bs[c * CHS + loff[q]] = breg[c][q];
    };

    int abase[TM], bbase[TN];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) abase[tm] = wm * (BM / 2) + tm * 32 + j;
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
        const int pt = wn * (BN / 2) + tn * 32 + j;
        const int seg = pt / (R * WS), rem = pt % (R * WS), al = rem / WS, bl = rem % WS;
        bbase[tn] = (seg * PR + al + 1) * PCP + bl + 1;
    }
    f32x16 acc[TM][TN][4];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
            for (int p = 0; p < 4; ++p)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[tm][tn][p][r] = 0.f;

    const int nk = Cs / CSC;
    gload(0);
    for (int kc = 0; kc < nk; ++kc) {
        lstore();
        __syncthreads();
        if (kc + 1 < nk) gload((kc + 1) * CSC);
#pragma unroll
        for (int g = 0; g < CSC / 2; ++g) {
            const int csl = 2 * g + h;
            float a[TM][16], bv[TN][3][3];
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int t = 0; t < 16; ++t) a[tm][t] = As[(csl * 16 + t) * ALD + abase[tm]];
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                for (int dh = 0; dh < 3; ++dh)
#pragma unroll
                    for (int dw = 0; dw < 3; ++dw)
                        bv[tn][dh][dw] = Bs[csl * CHS + bbase[tn] + (dh - 1) * PCP + (dw - 1)];
#pragma unroll
            for (int kh = 0; kh < 4; ++kh)
#pragma unroll
                for (int kw = 0; kw < 4; ++kw) {
                    const int ph = (kh == 0 || kh == 2) ? 1 : 0, dh = (kh == 0) ? 2 : (kh == 3 ? 0 : 1);
                    const int pw = (kw == 0 || kw == 2) ? 1 : 0, dw = (kw == 0) ? 2 : (kw == 3 ? 0 : 1);
#pragma unroll
                    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                        for (int tn = 0; tn < TN; ++tn)
                            acc[tm][tn][ph * 2 + pw] = __builtin_amdgcn_mfma_f32_32x32x2f32(
                                a[tm][kh * 4 + kw], bv[tn][dh][dw], acc[tm][tn][ph * 2 + pw], 0, 0, 0);
                }
        }
        __syncthreads();
    }

#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
        const int pt = wn * (BN / 2) + tn * 32 + j;
        const int seg = pt / (R * WS), rem = pt % (R * WS), al = rem / WS, bl = rem % WS;
        const int n = n0 + seg;
        if (n >= N) continue;
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int cl = cl0 + wm * (BM / 2) + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                float* o = X + (((int64_t)n * Cl + cl) * HL + 2 * (oh0 + al)) * WL + 2 * bl;
                *reinterpret_cast<float2*>(o) = make_float2(acc[tm][tn][0][r], acc[tm][tn][1][r]);
                *reinterpret_cast<float2*>(o + WL) = make_float2(acc[tm][tn][2][r], acc[tm][tn][3][r]);
            }
    }
}

// ------------------------------------------------------------------------------------------------
// UP, phase-split form for launches that would otherwise leave SIMDs idle: block tile = 32 cl x 32
// S-grid pixels, wave p of a group computes sub-pixel phase p only (one accumulator tile, K = Cs*4),
// and K (cs) may additionally be split over KG groups of 4 waves that are summed through LDS.
// ------------------------------------------------------------------------------------------------
template <int HS, int CSC, int KG>
__global__ __launch_bounds__(256 * KG) void conv_up_ps_kernel(const float* __restrict__ Y, const float* __restrict__ W,
                                                              float* __restrict__ X, int N, int Cs, int Cl) {
    constexpr int BM = 32, BN = 32;
    using T = PixTile<HS, BN>;
    constexpr int WS = HS, HL = 2 * HS, WL = 2 * HS;
    constexpr int SEG = T::SEG, R = T::R;
    constexpr int PR = R + 2, PC = WS + 2, PCP = PC | 1;
    constexpr int CHS = SEG * PR * PCP;
    constexpr int ALD = BM + 1;
    constexpr int PE = SEG * PR * PC, NQ = (PE + 255) / 256;
    constexpr int AV = CSC * BM * 4 / 256;
    constexpr int GRP = CSC * 16 * ALD + CSC * CHS;
    static_assert(CSC % 2 == 0 && AV >= 1, "tile shape");
    static_assert(KG == 1 || GRP * KG >= (KG - 1) * 4 * 16 * 64, "reduction scratch must fit");
    __shared__ __attribute__((aligned(16))) float lds[GRP * KG];
    const int grp = threadIdx.x >> 8;
    float* As = lds + grp * GRP;
    float* Bs = As + CSC * 16 * ALD;

    const int tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6;
    const int ph = wave >> 1, pw = wave & 1, j = lane & 31, h = lane >> 5;
    const int nmt = Cl / BM, npt = T::ntiles(N);
    const int b = xcd_remap(blockIdx.x, nmt * npt);
    const int ptile = b / nmt, cl0 = (b % nmt) * BM;
    int n0, oh0;
    T::origin(ptile, n0, oh0);

    int64_t poff[NQ];
    int loff[NQ];
    bool pval[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int e = tid + 256 * q;
        const int seg = e / (PR * PC), rem = e % (PR * PC), pr = rem / PC, pc = rem % PC;
        const int n = n0 + seg, a = oh0 - 1 + pr, bb = pc - 1;
        pval[q] = (e < PE) && n < N && a >= 0 && a < HS && bb >= 0 && bb < WS;
        poff[q] = ((int64_t)n * Cs * HS + a) * WS + bb;
        loff[q] = (e < PE) ? (seg * PR + pr) * PCP + pc : -1;
    }
    f32x4 areg[AV];
    float breg[CSC][NQ];
    auto gload = [&](int cs0) {
#pragma unroll
        for (int q = 0; q < AV; ++q) {
            const int idx = tid + 256 * q, csl = idx / (BM * 4), rem = idx % (BM * 4);
            areg[q] = *reinterpret_cast<const f32x4*>(W + ((int64_t)(cs0 + csl) * Cl + cl0) * 16 + rem * 4);
        }
#pragma unroll
        for (int c = 0; c < CSC; ++c)
#pragma unroll
            for (int q = 0; q < NQ; ++q)
                breg[c][q] = pval[q] ? Y[poff[q] + (int64_t)(cs0 + c) * HS * WS] : 0.f;
    };
    auto lstore = [&]() {
#pragma unroll
        for (int q = 0; q < AV; ++q) {
            const int idx = tid + 256 * q, csl = idx / (BM * 4), rem = idx % (BM * 4);
            const int cll = rem / 4, t4 = rem % 4;
#pragma unroll
            for (int e = 0; e < 4; ++e) As[(csl * 16 + t4 * 4 + e) * ALD + cll] = areg[q][e];
        }
#pragma unroll
        for (int c = 0; c < CSC; ++c)
#pragma unroll
            for (int q = 0; q < NQ; ++q)
                if (loff[q] >= 0) Bs[c * CHS + loff[q]] = breg[c][q];
    };

    // this wave's 4 taps: rows kh in {1,3} (ph 0) or {2,0} (ph 1) reading S rows a+{0,-1} / a+{0,+1}
    int aoff[4], boff[4];
    const int seg = j / (R * WS), al = (j % (R * WS)) / WS, bl = j % WS;
#pragma unroll
    for (int th = 0; th < 2; ++th)
#pragma unroll
        for (int tw = 0; tw < 2; ++tw) {
            const int kh = ph ? (th ? 0 : 2) : (th ? 3 : 1), dh = ph ? th : -th;
            const int kw = pw ? (tw ? 0 : 2) : (tw ? 3 : 1), dw = pw ? tw : -tw;
            aoff[th * 2 + tw] = (kh * 4 + kw) * ALD + j;
            boff[th * 2 + tw] = (seg * PR + al + 1 + dh) * PCP + bl + 1 + dw;
        }
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;

    const int nk = Cs / (CSC * KG);
    gload(grp * CSC);
    for (int kc = 0; kc < nk; ++kc) {
        lstore();
        __syncthreads();
        if (kc + 1 < nk) gload(((kc + 1) * KG + grp) * CSC);
#pragma unroll
        for (int g = 0; g < CSC / 2; ++g) {
            const int csl = 2 * g + h;
            float a[4], bv[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                a[t] = As[csl * 16 * ALD + aoff[t]];
                bv[t] = Bs[csl * CHS + boff[t]];
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], bv[t], acc, 0, 0, 0);
        }
        __syncthreads();
    }

    if (KG > 1) {
        float* red = lds + ((grp - 1) * 4 + wave) * (16 * 64) + lane;
        if (grp > 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) red[r * 64] = acc[r];
        }
        __syncthreads();
        if (grp > 0) return;
#pragma unroll
        for (int g = 1; g < KG; ++g) {
            const float* src = lds + ((g - 1) * 4 + wave) * (16 * 64) + lane;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] += src[r * 64];
        }
    }

    const int n = n0 + seg;
    if (n < N) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int cl = cl0 + (r & 3) + 8 * (r >> 2) + 4 * h;
            X[(((int64_t)n * Cl + cl) * HL + 2 * (oh0 + al) + ph) * WL + 2 * bl + pw] = acc[r];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// WGRAD: GEMM M = Cs (BM), N = (cl, tap) (16*CLB), K = pixels (PIXC per step), split over blockIdx.y.
// Each block writes its partial into slab[blockIdx.y]; wgrad_reduce sums the slabs.
// ------------------------------------------------------------------------------------------------
template <int HS, int BM, int CLB, int PIXC>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const float* __restrict__ YA, const float* __restrict__ X,
                                                         float* __restrict__ slab, int N, int Cs, int Cl, int n_keep,
                                                         int n_swap, int chunks_per_block) {
    using T = PixTile<HS, PIXC>;
    constexpr int WS = HS, HL = 2 * HS, WL = 2 * HS;
    constexpr int SEG = T::SEG, R = T::R;
    constexpr int PR = 2 * R + 2, PC = 2 * WS + 2, PCP = PC + 2;               // PCP = 4 (mod 8)
    constexpr int CHS = ((SEG * PR * PCP - 16 + 31) / 32) * 32 + 16;           // = 16 (mod 32)
    constexpr int ALD = PIXC + 2;
    constexpr int BN = 16 * CLB, TM = BM / 64, TN = BN / 64;
    constexpr int PE = SEG * PR * PC, NQ = (PE + 255) / 256;
    constexpr int AV = BM * PIXC / 4 / 256;
    static_assert(BM % 64 == 0 && BN % 64 == 0 && PIXC % 4 == 0 && AV >= 1, "tile shape");
    __shared__ __attribute__((aligned(16))) float lds[BM * ALD + CLB * CHS];
    float* As = lds;
    float* Bs = lds + BM * ALD;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, j = lane & 31, h = lane >> 5;
    const int nct = Cl / CLB;
    const int cs0 = (blockIdx.x / nct) * BM, cl0 = (blockIdx.x % nct) * CLB;
    const int total_chunks = T::ntiles(N);
    const int c_begin = blockIdx.y * chunks_per_block;
    const int c_end = min(total_chunks, c_begin + chunks_per_block);

    // chunk-invariant parts of the staging indices
    int a_row[AV], a_seg[AV], a_in[AV];
#pragma unroll
    for (int q = 0; q < AV; ++q) {
        const int idx = tid + 256 * q, row = idx / (PIXC / 4), p = (idx % (PIXC / 4)) * 4;
        a_row[q] = row;
        a_seg[q] = p / (R * WS);
        a_in[q] = p % (R * WS);   // = ohl * WS + ow, 4 pixels in one row
    }
    int b_seg[NQ], b_pr[NQ], b_pc[NQ], loff[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int e = tid + 256 * q;
        b_seg[q] = e / (PR * PC);
        const int rem = e % (PR * PC);
        b_pr[q] = rem / PC;
        b_pc[q] = rem % PC;
        loff[q] = (e < PE) ? (b_seg[q] * PR + b_pr[q]) * PCP + b_pc[q] : -1;
    }
    f32x4 areg[AV];
    float breg[CLB][NQ];
    auto gload = [&](int chunk) {
        int n0, oh0;
        T::origin(chunk, n0, oh0);
#pragma unroll
        for (int q = 0; q < AV; ++q) {
            const int n = n0 + a_seg[q];
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (n < N)
                v = *reinterpret_cast<const f32x4*>(YA + (((int64_t)n * Cs + cs0 + a_row[q]) * HS + oh0) * WS + a_in[q]);
            areg[q] = v;
        }
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int n = n0 + b_seg[q];
            const int rn = (n < n_keep) ? n : (n < n_keep + n_swap ? n + n_swap : n - n_swap);
            const int ih = 2 * oh0 - 1 + b_pr[q], iw = b_pc[q] - 1;
            const bool ok = loff[q] >= 0 && n < N && ih >= 0 && ih < HL && iw >= 0 && iw < WL;
            const float* src = X + (((int64_t)rn * Cl + cl0) * HL + ih) * WL + iw;
#pragma unroll
            for (int c = 0; c < CLB; ++c) breg[c][q] = ok ? src[(int64_t)c * HL * WL] : 0.f;
        }
    };
    auto lstore = [&]() {
#pragma unroll
        for (int q = 0; q < AV; ++q) {
            const int idx = tid + 256 * q, p = (idx % (PIXC / 4)) * 4;
            float* d = As + a_row[q] * ALD + p;
            *reinterpret_cast<float2*>(d) = make_float2(areg[q][0], areg[q][1]);
            *reinterpret_cast<float2*>(d + 2) = make_float2(areg[q][2], areg[q][3]);
        }
#pragma unroll
        for (int c = 0; c < CLB; ++c)
#pragma unroll
            for (int q = 0; q < NQ; ++q)
                if (loff[q] >= 0) Bs[c * CHS + loff[q]] = breg[c][q];
    };

    int abase[TM], bbase[TN];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) abase[tm] = (wm * (BM / 2) + tm * 32 + j) * ALD + 2 * h;
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
        bbase[tn] = (wn * (CLB / 2) + tn * 2 + (j >> 4)) * CHS + ((j >> 2) & 3) * PCP + (j & 3) + 4 * h;
    f32x16 acc[TM][TN];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[tm][tn][r] = 0.f;

    if (c_begin < c_end) gload(c_begin);
    for (int ch = c_begin; ch < c_end; ++ch) {
        lstore();
        __syncthreads();
        if (ch + 1 < c_end) gload(ch + 1);
#pragma unroll
        for (int q = 0; q < PIXC / 4; ++q) {
            const int p4 = 4 * q;  // compile-time after unrolling
            const int pofs = ((p4 / (R * WS)) * PR + 2 * ((p4 % (R * WS)) / WS)) * PCP + 2 * (p4 % WS);
            float2 a[TM];
            float b0[TN], b1[TN];
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) a[tm] = *reinterpret_cast<const float2*>(As + abase[tm] + p4);
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) {
                b0[tn] = Bs[bbase[tn] + pofs];
                b1[tn] = Bs[bbase[tn] + pofs + 2];
            }
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) {
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tm].x, b0[tn], acc[tm][tn], 0, 0, 0);
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tm].y, b1[tn], acc[tm][tn], 0, 0, 0);
                }
        }
        __syncthreads();
    }

    float* out = slab + (int64_t)blockIdx.y * Cs * Cl * 16;
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int cs = cs0 + wm * (BM / 2) + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                const int cl = cl0 + wn * (CLB / 2) + tn * 2 + (j >> 4);
                out[((int64_t)cs * Cl + cl) * 16 + (j & 15)] = acc[tm][tn][r];
            }
}

__global__ void slab_reduce_kernel(const float* __restrict__ slab, float* __restrict__ out, int nslab, int64_t len,
                                   int accumulate) {
    const int64_t i4 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i4 * 4 >= len) return;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    for (int z = 0; z < nslab; ++z) s += *reinterpret_cast<const f32x4*>(slab + z * len + i4 * 4);
    if (accumulate) s += *reinterpret_cast<const f32x4*>(out + i4 * 4);
    *reinterpret_cast<f32x4*>(out + i4 * 4) = s;
}

// ------------------------------------------------------------------------------------------------
// Cl == 1 stages (first discriminator conv / last generator conv-transpose): bandwidth-bound.
// ------------------------------------------------------------------------------------------------
// DOWN, Cl = 1: block = one sample x 1024 output pixels (RT rows) x 1 / gridDim.y of the cs (small batches are split
// so that every CU has several blocks' stores in flight); a thread takes four consecutive pixels of a row: their 4 x 10
// inputs stay in registers, every cs costs four broadcast weight reads, 64 multiply-adds and one 16-byte store.
// STATS: the block also leaves the sum and the sum of squares of its outputs in gn_ws[(n * gridDim.y + blockIdx.y) * 2 ..]
// -- with one block per sample and eight channel groups (hs = 32, gridDim.y = 8) these are the eight slice partial sums
// per sample that the GroupNorm forward's apply launch reads (norm.hip), so the layer's separate sums launch and its
// pass over the activation are not needed.
template <int HS, bool STATS>
__global__ __launch_bounds__(256) void conv_down_c1_kernel(const float* __restrict__ X, const float* __restrict__ W,
                                                           float* __restrict__ Y, int N, int Cs, double* __restrict__ gn_ws) {
    constexpr int WS = HS, HL = 2 * HS, WL = 2 * HS, RT = 1024 / WS;   // rows per block
    constexpr int PR = 2 * RT + 2, PC = 2 * WS + 4;                    // even row stride: 8-byte aligned pairs
    __shared__ __attribute__((aligned(16))) float ws[64 * 16];
    __shared__ __attribute__((aligned(16))) float xs[PR * PC];
    const int tid = threadIdx.x;
    const int n = blockIdx.x / (HS / RT), oh0 = (blockIdx.x % (HS / RT)) * RT;
    const int csn = Cs / (int)gridDim.y, cs_lo = (int)blockIdx.y * csn;
    for (int i = tid; i < csn * 16; i += 256) ws[i] = W[cs_lo * 16 + i];
    const float* xn = X + (int64_t)n * HL * WL;
    for (int i = tid; i < PR * (WL / 4); i += 256) {       // whole image rows by 16-byte loads, then the two halo columns
        const int pr = i / (WL / 4), q = i % (WL / 4), ih = 2 * oh0 - 1 + pr;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (ih >= 0 && ih < HL) v = *reinterpret_cast<const f32x4*>(xn + ih * WL + 4 * q);
        float* dst = xs + pr * PC + 1 + 4 * q;
        dst[0] = v[0]; dst[1] = v[1]; dst[2] = v[2]; dst[3] = v[3];
    }
    for (int i = tid; i < PR * 2; i += 256) xs[(i >> 1) * PC + (i & 1) * (WL + 1)] = 0.f;
    __syncthreads();
    const int ohl = tid / (WS / 4), ow0 = (tid % (WS / 4)) * 4;
    float x[4][10];
#pragma unroll
    for (int kh = 0; kh < 4; ++kh)
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const float2 v = *reinterpret_cast<const float2*>(xs + (2 * ohl + kh) * PC + 2 * ow0 + 2 * i);
            x[kh][2 * i] = v.x;
            x[kh][2 * i + 1] = v.y;
        }
    float* yn = Y + (((int64_t)n * Cs + cs_lo) * HS + oh0 + ohl) * WS + ow0;
    float s1 = 0.f, s2 = 0.f;
    for (int cs = 0; cs < csn; ++cs) {
        const f32x4* w4 = reinterpret_cast<const f32x4*>(ws + cs * 16);
        f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kh = 0; kh < 4; ++kh) {
            const f32x4 wv = w4[kh];
#pragma unroll
            for (int px = 0; px < 4; ++px)
#pragma unroll
                for (int kw = 0; kw < 4; ++kw) o[px] = fmaf(x[kh][2 * px + kw], wv[kw], o[px]);
        }
        *reinterpret_cast<f32x4*>(yn + (int64_t)cs * HS * WS) = o;
        if constexpr (STATS) {
            s1 += (o[0] + o[1]) + (o[2] + o[3]);
            s2 += (o[0] * o[0] + o[1] * o[1]) + (o[2] * o[2] + o[3] * o[3]);
        }
    }
    if constexpr (STATS) {       // fixed order: lanes by xor distance, then the four waves in turn
        __shared__ double red[4][2];
        double d1 = s1, d2 = s2;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            d1 += __shfl_xor(d1, o, 64);
            d2 += __shfl_xor(d2, o, 64);
        }
        if ((tid & 63) == 0) { red[tid >> 6][0] = d1; red[tid >> 6][1] = d2; }
        __syncthreads();
        if (tid == 0) {
            double* dst = gn_ws + ((int64_t)blockIdx.x * gridDim.y + blockIdx.y) * 2;
            dst[0] = (red[0][0] + red[1][0]) + (red[2][0] + red[3][0]);
            dst[1] = (red[0][1] + red[1][1]) + (red[2][1] + red[3][1]);
        }
    }
}

// UP, Cl = 1 (+ optional sigmoid / tanh).  Block = one sample x 4 S rows, 1024 threads = 8 channel groups x
// 128 S pixels: the whole Cs-channel patch is staged once, every group sums its channels into the 2x2
// outputs of its pixel, and the groups are combined through LDS.
template <int HS>
__global__ __launch_bounds__(1024) void conv_up_c1_kernel(const float* __restrict__ Y, const float* __restrict__ W,
                                                          float* __restrict__ X, int N, int Cs, int act) {
    constexpr int WS = HS, HL = 2 * HS, WL = 2 * HS, RT = 4, PX = RT * WS, NG = 1024 / PX;
    constexpr int PR = RT + 2, PC = WS + 2, PCP = PC | 1;
    static_assert(PX * NG == 1024 && NG * 4 * PX <= 64 * PR * PCP, "block shape");
    __shared__ __attribute__((aligned(16))) float ws[64 * 16];
    __shared__ float ys[64 * PR * PCP];          // reused for the group partial sums
    const int tid = threadIdx.x;
    const int n = blockIdx.x / (HS / RT), a0 = (blockIdx.x % (HS / RT)) * RT;
    for (int i = tid; i < Cs * 16; i += 1024) ws[i] = W[i];
    // the patch rows are whole image rows: 16-byte loads (WS / 4 per row, a few per thread, all in flight together);
    // the two halo columns are always outside the image
    constexpr int QR = WS / 4;
    const float* yn = Y + (int64_t)n * Cs * HS * WS;
    for (int i = tid; i < Cs * PR * QR; i += 1024) {
        const int c = i / (PR * QR), pr = (i / QR) % PR, q = i % QR, a = a0 - 1 + pr;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (a >= 0 && a < HS) v = *reinterpret_cast<const f32x4*>(yn + ((int64_t)c * HS + a) * WS + 4 * q);
        float* dst = ys + (c * PR + pr) * PCP + 1 + 4 * q;
        dst[0] = v[0]; dst[1] = v[1]; dst[2] = v[2]; dst[3] = v[3];
    }
    for (int i = tid; i < Cs * PR * 2; i += 1024) ys[(i >> 1) * PCP + (i & 1) * (PC - 1)] = 0.f;
    __syncthreads();
    const int g = tid / PX, p = tid % PX, al = p / WS, bl = p % WS, cpg = Cs / NG;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int c = g * cpg; c < (g + 1) * cpg; ++c) {
        float v[3][3];
#pragma unroll
        for (int dh = 0; dh < 3; ++dh)
#pragma unroll
            for (int dw = 0; dw < 3; ++dw) v[dh][dw] = ys[(c * PR + al + dh) * PCP + bl + dw];
        const float* w = ws + c * 16;
#pragma unroll
        for (int kh = 0; kh < 4; ++kh)
#pragma unroll
            for (int kw = 0; kw < 4; ++kw) {
                const int ph = (kh == 0 || kh == 2) ? 1 : 0, dh = (kh == 0) ? 2 : (kh == 3 ? 0 : 1);
                const int pw = (kw == 0 || kw == 2) ? 1 : 0, dw = (kw == 0) ? 2 : (kw == 3 ? 0 : 1);
                acc[ph * 2 + pw] = fmaf(v[dh][dw], w[kh * 4 + kw], acc[ph * 2 + pw]);
            }
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; ++q) ys[(g * 4 + q) * PX + p] = acc[q];
    __syncthreads();
    if (g < 2) {     // group g finishes output-row phase g of its pixel: both column phases, one float2
        float s0 = 0.f, s1 = 0.f;
        for (int k = 0; k < NG; ++k) {     // fixed order: the sum does not depend on timing
            s0 += ys[(k * 4 + g * 2) * PX + p];
            s1 += ys[(k * 4 + g * 2 + 1) * PX + p];
        }
        if (act == 1) { s0 = 1.f / (1.f + expf(-s0)); s1 = 1.f / (1.f + expf(-s1)); }
        else if (act == 2) { s0 = tanhf(s0); s1 = tanhf(s1); }
        float* o = X + ((int64_t)n * HL + 2 * (a0 + al) + g) * WL + 2 * bl;
        *reinterpret_cast<float2*>(o) = make_float2(s0, s1);
    }
}

// WGRAD, Cl = 1: block = (sample, group of 16 cs); wave = 4 cs.  A lane takes four consecutive pixels of a row per step
// (one 16-byte load per channel, the next step's loads in flight during the products) and reads the 4 x 10 input values
// they touch from the staged sample as 8-byte LDS reads.  Writes per-sample partials part[n][cs*16 + tap]; a row
// reduction finishes the sum.
template <int HS>
__global__ __launch_bounds__(256, HS == 32 ? 4 : 2) void conv_wgrad_c1_kernel(const float* __restrict__ YA, const float* __restrict__ X,
                                                            float* __restrict__ part, int N, int Cs, int n_keep, int n_swap) {
    constexpr int WS = HS, HL = 2 * HS, WL = 2 * HS, PC = WL + 2, PCP = WL + 4;     // even row stride: 8-byte aligned pairs
    constexpr int GPR = WS / 4, STEPS = HS * WS / 256;                              // pixel groups per row; steps per wave
    __shared__ __attribute__((aligned(16))) float xs[(HL + 2) * PCP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ngrp = Cs / 16;
    const int n = blockIdx.x / ngrp, cs0 = (blockIdx.x % ngrp) * 16 + wave * 4;
    const int rn = (n < n_keep) ? n : (n < n_keep + n_swap ? n + n_swap : n - n_swap);
    const float* xn = X + (int64_t)rn * HL * WL;
    // the sample's rows by 16-byte loads (all of a thread's in flight together), then the zero frame around them
    for (int i = tid; i < HL * (WL / 4); i += 256) {
        const int ih = i / (WL / 4), q = i % (WL / 4);
        const f32x4 v = *reinterpret_cast<const f32x4*>(xn + ih * WL + 4 * q);
        float* dst = xs + (ih + 1) * PCP + 1 + 4 * q;
        dst[0] = v[0]; dst[1] = v[1]; dst[2] = v[2]; dst[3] = v[3];
    }
    for (int i = tid; i < PC; i += 256) { xs[i] = 0.f; xs[(HL + 1) * PCP + i] = 0.f; }
    for (int i = tid; i < HL; i += 256) { xs[(i + 1) * PCP] = 0.f; xs[(i + 1) * PCP + PC - 1] = 0.f; }
    float acc[4][16];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int t = 0; t < 16; ++t) acc[c][t] = 0.f;
    const float* ya = YA + ((int64_t)n * Cs + cs0) * HS * WS + 4 * lane;
    f32x4 y[4], yn[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) y[c] = *reinterpret_cast<const f32x4*>(ya + c * HS * WS);
    __syncthreads();
#pragma unroll 1
    for (int st = 0; st < STEPS; ++st) {
        if (st + 1 < STEPS)
#pragma unroll
            for (int c = 0; c < 4; ++c) yn[c] = *reinterpret_cast<const f32x4*>(ya + c * HS * WS + (st + 1) * 256);
        const int g = st * 64 + lane, oh = g / GPR, ow0 = (g % GPR) * 4;
        const float* xr = xs + (2 * oh) * PCP + 2 * ow0;          // input row 2 oh - 1, column 2 ow0 - 1
#pragma unroll
        for (int kh = 0; kh < 4; ++kh) {
            float xv[10];
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                const float2 v = *reinterpret_cast<const float2*>(xr + kh * PCP + 2 * i);
                xv[2 * i] = v.x;
                xv[2 * i + 1] = v.y;
            }
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int px = 0; px < 4; ++px)
#pragma unroll
                    for (int kw = 0; kw < 4; ++kw) acc[c][kh * 4 + kw] = fmaf(y[c][px], xv[2 * px + kw], acc[c][kh * 4 + kw]);
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) y[c] = yn[c];
    }
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const float s = wave_sum(acc[c][t]);
            if (lane == 0) part[(int64_t)n * Cs * 16 + (cs0 + c) * 16 + t] = s;
        }
}

// out[c] (+)= sum_r coef[r] * part[r * ld + c] over rows [r0, r1) and [r2, r3) (coef == nullptr: 1).
// Block = 32 columns x RL row lanes (1024 threads): the rows are strided over the lanes, four loads in flight per lane,
// and combined through LDS in a fixed order.  These launches have few columns (a few thousand) and up to a few hundred
// rows: they are bound by how many loads are in flight, not by bytes.
constexpr int RL = 32;
__device__ __forceinline__ float rows_partial(const float* __restrict__ part, int ld, const float* __restrict__ coef, int c,
                                              int lo, int hi, int ry) {
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int r = lo + ry;
    for (; r + 3 * RL < hi; r += 4 * RL) {
        const float v0 = part[(int64_t)r * ld + c], v1 = part[(int64_t)(r + RL) * ld + c];
        const float v2 = part[(int64_t)(r + 2 * RL) * ld + c], v3 = part[(int64_t)(r + 3 * RL) * ld + c];
        s0 = fmaf(coef ? coef[r] : 1.f, v0, s0);
        s1 = fmaf(coef ? coef[r + RL] : 1.f, v1, s1);
        s2 = fmaf(coef ? coef[r + 2 * RL] : 1.f, v2, s2);
        s3 = fmaf(coef ? coef[r + 3 * RL] : 1.f, v3, s3);
    }
    for (; r < hi; r += RL) s0 = fmaf(coef ? coef[r] : 1.f, part[(int64_t)r * ld + c], s0);
    return (s0 + s1) + (s2 + s3);
}
__global__ __launch_bounds__(32 * RL) void rows_sum_kernel(const float* __restrict__ part, int ld,
                                                           const float* __restrict__ coef, float* __restrict__ out, int C,
                                                           int r0, int r1, int r2, int r3, int accumulate) {
    __shared__ float sh[RL][33];
    const int cx = threadIdx.x & 31, ry = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cx;
    float s = 0.f;
    if (c < C) s = rows_partial(part, ld, coef, c, r0, r1, ry) + rows_partial(part, ld, coef, c, r2, r3, ry);
    sh[ry][cx] = s;
    __syncthreads();
    if (ry == 0 && c < C) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < RL; ++k) t += sh[k][cx];
        out[c] = accumulate ? out[c] + t : t;
    }
}

// wave tiles (32 x 32 outputs each) a launch should reach before it stops splitting K: two per SIMD
constexpr int FILL_WAVES = 2048;
// tuning overrides (0 = choose from the launch size): see pgan_set_option
int g_c1_gy = 0;       // tuning override "c1_gy": cs split of the single-channel DOWN kernel
int g_down_kg = 0;     // 1, 2, 4: K groups of the DOWN kernel
int g_down_ks = 0;     // 1, 2: K split across blocks of the DOWN kernel (2 partial sums added atomically)
int g_down_variant = 0;  // experiment: 1 = 128x64 tile, 2 = 64x128, 3 = 128x128 (2 / 2 / 4 accumulators per wave)
int g_up_mode = 0;     // 1: all-phase 64x64 tiles; 2: phase-split, no K split; 3: phase-split, 2 K groups

// Up to 4 independent row reductions in one launch (blockIdx.y selects the segment).
struct RowSegs {
    const float* part[4];
    float* out[4];
    int ld[4];
    int C[4];
};
__global__ __launch_bounds__(32 * RL) void rows_sum_multi_kernel(RowSegs sg, int r0, int r1, int r2, int r3) {
    __shared__ float sh[RL][33];
    const int seg = blockIdx.y;
    const int C = sg.C[seg], ld = sg.ld[seg];
    const float* part = sg.part[seg];
    const int cx = threadIdx.x & 31, ry = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cx;
    if (blockIdx.x * 32 >= C) return;
    float s = 0.f;
    if (c < C) s = rows_partial(part, ld, nullptr, c, r0, r1, ry) + rows_partial(part, ld, nullptr, c, r2, r3, ry);
    sh[ry][cx] = s;
    __syncthreads();
    if (ry == 0 && c < C) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < RL; ++k) t += sh[k][cx];
        sg.out[seg][c] = t;
    }
}

template <int HS>
int launch_down(const float* X, const float* W, float* Y, int N, int Cs, int Cl, hipStream_t st) {
    constexpr int BM = 64, BN = 64;
    if (Cs % BM || Cl % 16) return PGAN_EINVAL;
    const int grid = (Cs / BM) * PixTile<HS, BN>::ntiles(N);
    // Fewer than ~2 blocks per CU: split K over two blocks as well (two partial sums added onto a
    // zeroed output: the sum of two terms does not depend on their order).
    if (g_down_variant == 1 && Cs % 128 == 0) {
        conv_down_kernel<HS, 128, 64, 4, 1><<<(Cs / 128) * PixTile<HS, 64>::ntiles(N), 256, 0, st>>>(X, W, Y, N, Cs, Cl);
        return pgan_launch_status();
    }
    if (g_down_variant == 2) {
        conv_down_kernel<HS, 64, 128, 8, 1><<<(Cs / 64) * PixTile<HS, 128>::ntiles(N), 256, 0, st>>>(X, W, Y, N, Cs, Cl);
        return pgan_launch_status();
    }
    if (g_down_variant == 3 && Cs % 128 == 0) {
        conv_down_kernel<HS, 128, 128, 4, 1><<<(Cs / 128) * PixTile<HS, 128>::ntiles(N), 256, 0, st>>>(X, W, Y, N, Cs, Cl);
        return pgan_launch_status();
    }
    if (g_down_variant >= 4) {
        const int ksp = g_down_ks ? g_down_ks : 1;
        if ((int64_t)N * Cl * HS * HS * 4 >= (1ll << 31) || (int64_t)Cs * Cl * 16 >= (1ll << 31)) return PGAN_EINVAL;
        if (ksp > 1 && hipMemsetAsync(Y, 0, sizeof(float) * (size_t)N * Cs * HS * HS, st) != hipSuccess) return PGAN_ELAUNCH;
        if (g_down_variant == 4 && Cl % (8 * ksp) == 0)
            conv_down_pipe_kernel<HS, 64, 64, 4><<<dim3((Cs / 64) * PixTile<HS, 64>::ntiles(N), ksp), 256, 0, st>>>(X, W, Y, N, Cs, Cl);
        else if (g_down_variant == 5 && Cs % 128 == 0 && Cl % (4 * ksp) == 0)
            conv_down_pipe_kernel<HS, 128, 64, 2><<<dim3((Cs / 128) * PixTile<HS, 64>::ntiles(N), ksp), 256, 0, st>>>(X, W, Y, N, Cs, Cl);
        else if (g_down_variant == 6 && Cl % (8 * ksp) == 0)
            conv_down_pipe_kernel<HS, 64, 128, 4><<<dim3((Cs / 64) * PixTile<HS, 128>::ntiles(N), ksp), 256, 0, st>>>(X, W, Y, N, Cs, Cl);
        else if (g_down_variant == 7 && Cs % 128 == 0 && Cl % (4 * ksp) == 0)
            conv_down_pipe_kernel<HS, 128, 128, 2><<<dim3((Cs / 128) * PixTile<HS, 128>::ntiles(N), ksp), 256, 0, st>>>(X, W, Y, N, Cs, Cl);
        else return PGAN_EINVAL;
        return pgan_launch_status();
    }
    const int ks = g_down_ks ? g_down_ks : ((grid < 512 && Cl >= 128) ? 2 : 1);
    const int waves = grid * 4 * ks;
    const int kg = g_down_kg ? g_down_kg : (waves >= FILL_WAVES ? 1 : (waves * 2 >= FILL_WAVES ? 2 : 4));
    if (Cl % (8 * ks) || Cl % (4 * kg * ks)) return PGAN_EINVAL;
    if (ks > 1 && hipMemsetAsync(Y, 0, sizeof(float) * (size_t)N * Cs * HS * HS, st) != hipSuccess) return PGAN_ELAUNCH;
    const dim3 g(grid, ks);
    if (kg == 1) conv_down_kernel<HS, BM, BN, 8, 1><<<g, 256, 0, st>>>(X, W, Y, N, Cs, Cl);
    else if (kg == 2) conv_down_kernel<HS, BM, BN, 4, 2><<<g, 512, 0, st>>>(X, W, Y, N, Cs, Cl);
    else conv_down_kernel<HS, BM, BN, 4, 4><<<g, 1024, 0, st>>>(X, W, Y, N, Cs, Cl);
    return pgan_launch_status();
}
template <int HS>
int launch_up(const float* Y, const float* W, float* X, int N, int Cs, int Cl, hipStream_t st) {
    constexpr int BM = 64, BN = 64, CSC = 8;
    if (Cl % BM || Cs % 32) return PGAN_EINVAL;
    const int grid = (Cl / BM) * PixTile<HS, BN>::ntiles(N);
    const int grid_ps = (Cl / 32) * PixTile<HS, 32>::ntiles(N);
    // all-phase 64x64 tiles run 2 blocks per CU: use them when they fill whole rounds of 512 blocks
    // (always at hs = 16, where the phase-split form's stride-2 stores cost more than the idle CUs)
    int mode = g_up_mode;
    if (!mode) mode = (HS == 16 || grid * 10 >= ((grid + 511) / 512) * 512 * 9) ? 1 : 3;
    (void)FILL_WAVES;
    if (mode == 1) conv_up_kernel<HS, BM, BN, CSC><<<grid, 256, 0, st>>>(Y, W, X, N, Cs, Cl);
    else if (mode == 2) conv_up_ps_kernel<HS, 16, 1><<<grid_ps, 256, 0, st>>>(Y, W, X, N, Cs, Cl);
    else conv_up_ps_kernel<HS, 8, 2><<<grid_ps, 512, 0, st>>>(Y, W, X, N, Cs, Cl);
    return pgan_launch_status();
}
constexpr int WG_BM = 64, WG_CLB = 8, WG_PIXC = 64;
template <int HS>
int wgrad_splits(int N, int Cs, int Cl) {
    const int tiles = (Cs / WG_BM) * (Cl / WG_CLB);
    const int chunks = PixTile<HS, WG_PIXC>::ntiles(N);
    int want = (1024 + tiles - 1) / tiles;       // aim at ~4 blocks per CU
    if (want > chunks) want = chunks;
    if (want < 1) want = 1;
    const int cpb = (chunks + want - 1) / want;
    return (chunks + cpb - 1) / cpb;
}
template <int HS>
int launch_wgrad(const float* YA, const float* X, float* dW, float* ws, int64_t ws_floats, int N, int Cs, int Cl,
                 int n_keep, int n_swap, int accumulate, hipStream_t st) {
    if (Cs % WG_BM || Cl % WG_CLB) return PGAN_EINVAL;
    const int chunks = PixTile<HS, WG_PIXC>::ntiles(N);
    const int ns = wgrad_splits<HS>(N, Cs, Cl);
    const int cpb = (chunks + ns - 1) / ns;
    const int64_t len = (int64_t)Cs * Cl * 16;
    if (ns * len > ws_floats) return PGAN_EINVAL;
    dim3 grid((Cs / WG_BM) * (Cl / WG_CLB), ns);
    conv_wgrad_kernel<HS, WG_BM, WG_CLB, WG_PIXC><<<grid, 256, 0, st>>>(YA, X, ws, N, Cs, Cl, n_keep, n_swap, cpb);
    slab_reduce_kernel<<<(int)((len / 4 + 255) / 256), 256, 0, st>>>(ws, dW, ns, len, accumulate);
    return pgan_launch_status();
}

}  // namespace

extern "C" {

// Tuning overrides for experiments; 0 restores the size-based choice.  "down_kg": 1|2|4, "up_mode": 1|2|3.
int pgan_set_option(const char* name, int value) {
    if (!name) return PGAN_EINVAL;
    const char* a = "down_kg";
    const char* b = "up_mode";
    auto eq = [](const char* x, const char* y) { while (*x && *x == *y) { ++x; ++y; } return *x == *y; };
    if (eq(name, a) && (value == 0 || value == 1 || value == 2 || value == 4)) { g_down_kg = value; return PGAN_OK; }
    if (eq(name, b) && value >= 0 && value <= 3) { g_up_mode = value; return PGAN_OK; }
    if (eq(name, "down_ks") && value >= 0 && value <= 8) { g_down_ks = value; return PGAN_OK; }
    if (eq(name, "down_variant") && value >= 0 && value <= 7) { g_down_variant = value; return PGAN_OK; }
    if (eq(name, "c1_gy") && (value == 0 || value == 1 || value == 2 || value == 4 || value == 8)) { g_c1_gy = value; return PGAN_OK; }
    return PGAN_EINVAL;
}

// pgan_conv_down for the discriminator's first stage (Cl = 1, hs = 32, Cs = 64) that also leaves the GroupNorm slice
// partial sums of its output in gn_ws (the layout pgan_gn_fwd's apply launch reads: [N][8][2] doubles): follow it with
// pgan_gn_fwd_presummed on the same stream and workspace.  Other shapes: PGAN_EINVAL, nothing launched.
int pgan_conv_down_gn_sums(const float* L, const float* W, float* S, int N, int Cs, int Cl, int hs, void* gn_ws,
                           long long gn_ws_bytes, hipStream_t st) {
    if (N <= 0) return PGAN_OK;
    if (Cl != 1 || hs != 32 || Cs != 64 || !gn_ws || gn_ws_bytes < (long long)N * 8 * 2 * 8) return PGAN_EINVAL;
    conv_down_c1_kernel<32, true><<<dim3(N, 8), 256, 0, st>>>(L, W, S, N, Cs, static_cast<double*>(gn_ws));
    return pgan_launch_status();
}

// S[N,Cs,H,H] = conv2d(L[N,Cl,2H,2H], W[Cs,Cl,4,4], stride 2, pad 1).  H = hs.
int pgan_conv_down(const float* L, const float* W, float* S, int N, int Cs, int Cl, int hs, hipStream_t st) {
    if (N <= 0) return PGAN_OK;
    if (Cl == 1) {
        if ((hs != 32 && hs != 64) || Cs > 64) return PGAN_EINVAL;
        // blocks of 1024 pixels: one (hs = 32) or four (hs = 64) per sample; the cs are split until about two blocks per CU
        const int per = hs == 32 ? 1 : 4;
        int gy = 1;
        while (gy < 8 && Cs % (2 * gy) == 0 && N * per * gy < 512) gy *= 2;
        if (g_c1_gy) gy = g_c1_gy;
        if (Cs % gy) return PGAN_EINVAL;
        if (hs == 32) conv_down_c1_kernel<32, false><<<dim3(N * per, gy), 256, 0, st>>>(L, W, S, N, Cs, nullptr);
        else conv_down_c1_kernel<64, false><<<dim3(N * per, gy), 256, 0, st>>>(L, W, S, N, Cs, nullptr);
        return pgan_launch_status();
    }
    switch (hs) {
        case 4: return launch_down<4>(L, W, S, N, Cs, Cl, st);
        case 8: return launch_down<8>(L, W, S, N, Cs, Cl, st);
        case 16: return launch_down<16>(L, W, S, N, Cs, Cl, st);
        default: return PGAN_EINVAL;
    }
}

// L[N,Cl,2H,2H] = conv_transpose2d(S[N,Cs,H,H], W[Cs,Cl,4,4], stride 2, pad 1); act: 0 none, 1 sigmoid,
// 2 tanh (only for Cl == 1).
int pgan_conv_up(const float* S, const float* W, float* L, int N, int Cs, int Cl, int hs, int act, hipStream_t st) {
    if (N <= 0) return PGAN_OK;
    if (Cl == 1) {
        if ((hs != 32 && hs != 64) || Cs > 64 || Cs % 8) return PGAN_EINVAL;
        if (hs == 32) conv_up_c1_kernel<32><<<N * (32 / 4), 1024, 0, st>>>(S, W, L, N, Cs, act);
        else conv_up_c1_kernel<64><<<N * (64 / 4), 1024, 0, st>>>(S, W, L, N, Cs, act);
        return pgan_launch_status();
    }
    if (act != 0) return PGAN_EINVAL;
    switch (hs) {
        case 4: return launch_up<4>(S, W, L, N, Cs, Cl, st);
        case 8: return launch_up<8>(S, W, L, N, Cs, Cl, st);
        case 16: return launch_up<16>(S, W, L, N, Cs, Cl, st);
        default: return PGAN_EINVAL;
    }
}

// Scratch floats pgan_conv_wgrad needs for these dimensions.
long long pgan_conv_wgrad_workspace_floats(int N, int Cs, int Cl, int hs) {
    if (Cl == 1) return (long long)N * Cs * 16;
    int ns = 1;
    switch (hs) {
        case 4: ns = wgrad_splits<4>(N, Cs, Cl); break;
        case 8: ns = wgrad_splits<8>(N, Cs, Cl); break;
        case 16: ns = wgrad_splits<16>(N, Cs, Cl); break;
        default: return -1;
    }
    return (long long)ns * Cs * Cl * 16;
}

// dW[Cs,Cl,4,4] (+)= sum_n S[n] (x) L[r(n)],  r(n) = n (n < n_keep), n + n_swap (n < n_keep + n_swap), else n - n_swap.
int pgan_conv_wgrad(const float* S, const float* L, float* dW, float* ws, long long ws_floats, int N, int Cs, int Cl,
                    int hs, int n_keep, int n_swap, int accumulate, hipStream_t st) {
    if (N <= 0) return PGAN_EINVAL;
    if (Cl == 1) {
        if ((hs != 32 && hs != 64) || Cs % 16 || (long long)N * Cs * 16 > ws_floats) return PGAN_EINVAL;
        if (hs == 32) conv_wgrad_c1_kernel<32><<<N * (Cs / 16), 256, 0, st>>>(S, L, ws, N, Cs, n_keep, n_swap);
        else conv_wgrad_c1_kernel<64><<<N * (Cs / 16), 256, 0, st>>>(S, L, ws, N, Cs, n_keep, n_swap);
        rows_sum_kernel<<<(Cs * 16 + 31) / 32, 32 * RL, 0, st>>>(ws, Cs * 16, nullptr, dW, Cs * 16, 0, N, 0, 0, accumulate);
        return pgan_launch_status();
    }
    switch (hs) {
        case 4: return launch_wgrad<4>(S, L, dW, ws, ws_floats, N, Cs, Cl, n_keep, n_swap, accumulate, st);
        case 8: return launch_wgrad<8>(S, L, dW, ws, ws_floats, N, Cs, Cl, n_keep, n_swap, accumulate, st);
        case 16: return launch_wgrad<16>(S, L, dW, ws, ws_floats, N, Cs, Cl, n_keep, n_swap, accumulate, st);
        default: return PGAN_EINVAL;
    }
}

// out[c] (+)= sum of part[r * ld + c] over rows [r0, r1) and [r2, r3), c < C.
int pgan_rows_reduce(const float* part, int ld, float* out, int C, int r0, int r1, int r2, int r3, int accumulate,
                     hipStream_t st) {
    if (C <= 0 || ld < C) return PGAN_EINVAL;
    rows_sum_kernel<<<(C + 31) / 32, 32 * RL, 0, st>>>(part, ld, nullptr, out, C, r0, r1, r2, r3, accumulate);
    return pgan_launch_status();
}

// nseg (<= 4) reductions out[i][c] = sum of part[i][r * ld[i] + c] over rows [r0,r1) and [r2,r3), one launch.
int pgan_rows_reduce_multi(const float* const* part, const int* ld, float* const* out, const int* C, int nseg, int r0,
                           int r1, int r2, int r3, hipStream_t st) {
    if (nseg < 1 || nseg > 4) return PGAN_EINVAL;
    RowSegs sg;
    int cmax = 0;
    for (int i = 0; i < 4; ++i) {
        const int k = i < nseg ? i : 0;
        if (C[k] <= 0 || ld[k] < C[k]) return PGAN_EINVAL;
        sg.part[i] = part[k]; sg.out[i] = out[k]; sg.ld[i] = ld[k]; sg.C[i] = C[k];
        if (C[k] > cmax) cmax = C[k];
    }
    rows_sum_multi_kernel<<<dim3((cmax + 31) / 32, nseg), 32 * RL, 0, st>>>(sg, r0, r1, r2, r3);
    return pgan_launch_status();
}

// dw[j] = sum_n coef[n] * x[n * J + j]: weight gradient of the linear head and its bias.
int pgan_head_wgrad(const float* x, const float* coef, float* dw, int N, int J, hipStream_t st) {
    if (N <= 0 || J <= 0) return PGAN_EINVAL;
    rows_sum_kernel<<<(J + 31) / 32, 32 * RL, 0, st>>>(x, J, coef, dw, J, 0, N, 0, 0, 0);
    return pgan_launch_status();
}

}  // extern "C"
