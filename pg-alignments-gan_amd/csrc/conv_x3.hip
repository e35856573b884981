// fp32-accurate k=4 / stride=2 / pad=1 convolutions on the bf16 matrix pipe (gfx950,
// v_mfma_f32_32x32x16_bf16).
//
// Each fp32 operand x is split into three bf16 numbers x = x1 + x2 + x3 (x1 = bf16(x),
// x2 = bf16(x - x1), x3 = bf16(x - x1 - x2): 3 x 8 significant bits cover the 24 of fp32) and a
// product x*y is evaluated as the six terms x1y1 + x1y2 + x2y1 + x2y2 + x1y3 + x3y1, every one an
// exact bf16 product accumulated in fp32 by the MFMA.  The dropped terms are below 2^-26 |xy|, i.e.
// under the rounding error of an fp32 multiply, so results carry fp32 accuracy (measured: closer to
// the fp64 result than the fp32 MFMA kernels).  Six bf16 MFMAs (K = 16, 8 passes) replace eight fp32
// MFMAs (K = 2, 16 passes), and the fp32 pipe is power-limited to about 92 TFLOP/s on random data where
// the bf16 pipe sustains about 1770.
//
// Weights enter as prepared bf16 planes (pgan_conv_x3_prep, once per weight update) laid out so that one
// wave's MFMA fragment is one contiguous 1 KB read: the waves load them straight from L2/L1 into registers,
// one step ahead of their use, and only the activations (split while they are staged) go through LDS --
// with both operands in LDS these kernels were bound by LDS bandwidth, not by the matrix pipe.
//
// Same operator definitions and tensor layouts as conv_mfma.hip (reference src/layers.py:441-463
// Conv2d, :97-112 ConvTranspose2d).  Operand maps of the 32x32x16 MFMA: lane l supplies
// A[i = l&31][k = 8(l>>5) .. +7] and B[k = 8(l>>5) .. +7][j = l&31]; result register r of lane l is
// D[i = (r&3) + 8(r>>2) + 4(l>>5)][j = l&31].
#include "conv_tiles.h"

namespace {
using namespace pgan;

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

// (x, y) -> three packed bf16 pairs; low half belongs to x.
__device__ __forceinline__ void split3(float x, float y, uint32_t (&p)[3]) {
    f32x2 v = {x, y};
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const uint32_t u = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
        p[i] = u;
        if (i < 2) {
            v.x -= __builtin_bit_cast(float, u << 16);
            v.y -= __builtin_bit_cast(float, u & 0xffff0000u);
        }
    }
}

__device__ __forceinline__ f32x16 mfma_bf16(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// The six significant partial products of (a1+a2+a3)(b1+b2+b3).  The leading term a1*b1 has its own
// accumulator: the matrix pipe truncates (toward -inf) the bits of a product that fall below the
// accumulator's last place, so correction terms added straight into the large sum would each lose up to
// 2^-9 of their value in one direction -- a bias that grows with K (measured -2.7e-7 relative at
// K = 4096 against -1e-9 with the split accumulators, and a 2.5x larger rms error).
__device__ __forceinline__ void mfma_x3(const u32x4 (&a)[3], const u32x4 (&b)[3], f32x16& hi, f32x16& lo) {
    lo = mfma_bf16(a[2], b[0], lo);
    lo = mfma_bf16(a[0], b[2], lo);
    lo = mfma_bf16(a[1], b[1], lo);
    lo = mfma_bf16(a[1], b[0], lo);
    lo = mfma_bf16(a[0], b[1], lo);
    hi = mfma_bf16(a[0], b[0], hi);
}

// ------------------------------------------------------------------------------------------------
// Weight preparation: W[Cs][Cl][16] fp32 -> bf16 planes in the two fragment orders the kernels read:
//   dn[p][cl][cs][tap']           DOWN: rows cs, one K step = the 16 taps of channel cl, stored in the order
//                                 the patch is read: tap' = 8(kh>>1) + 4(kw>>1) + 2(kh&1) + (kw&1)
//   up[p][tap][cs/16][cl][cs%16]  UP:   rows cl, one K step = 16 cs at a fixed tap
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void prep_kernel(const float* __restrict__ W, uint16_t* __restrict__ dn,
                                                   uint16_t* __restrict__ up, int Cs, int Cl) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= Cs * Cl) return;
    const int cs = idx % Cs, cl = idx / Cs;
    const size_t plane = (size_t)Cs * Cl * 16;
    const float* w = W + ((size_t)cs * Cl + cl) * 16;
    uint32_t pk[3][8];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(w + 4 * q);
        uint32_t a[3], b[3];
        split3(v[0], v[1], a);
        split3(v[2], v[3], b);
#pragma unroll
        for (int i = 0; i < 3; ++i) { pk[i][2 * q] = a[i]; pk[i][2 * q + 1] = b[i]; }
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        // pk[i][2kh + (kw>>1)] holds taps (kh, kw even | odd): dword order (kh0,kw01) (kh1,kw01) (kh0,kw23) (kh1,kw23)
        u32x4* d = reinterpret_cast<u32x4*>(dn + i * plane + ((size_t)cl * Cs + cs) * 16);
        u32x4 lo = {pk[i][0], pk[i][2], pk[i][1], pk[i][3]}, hi = {pk[i][4], pk[i][6], pk[i][5], pk[i][7]};
        d[0] = lo;
        d[1] = hi;
#pragma unroll
        for (int t = 0; t < 16; ++t)
            up[i * plane + (((size_t)t * (Cs / 16) + cs / 16) * Cl + cl) * 16 + (cs & 15)] =
                (uint16_t)(pk[i][t >> 1] >> (16 * (t & 1)));
    }
}

// ------------------------------------------------------------------------------------------------
// DOWN: GEMM M = Cs (BM per block), N = pixels (BN per block), K = Cl*16; one MFMA K step is one
// input channel's 16 taps.  Weight fragments come straight from global memory one channel ahead.
// LDS holds, per bf16 plane, the input patches [CLC][row pair][column pair][2 rows] (two bf16 per dword, so
// a lane's eight K values are 16 contiguous bytes), double-buffered
// with one barrier per chunk of CLC channels: chunk k+1 goes from registers to LDS (split on the way)
// between the MFMAs of chunk k while chunk k+2 is in flight from global memory into a second register set.
// ------------------------------------------------------------------------------------------------
template <int HS, int BM, int BN, int CLC>
__global__ __launch_bounds__(256) void conv_down_x3_kernel(const float* __restrict__ X, const uint16_t* __restrict__ WP,
                                                           float* __restrict__ Y, int N, int Cs, int Cl) {
    using T = PixTile<HS, BN>;
    constexpr int WS = HS, HL = 2 * HS, WL = 2 * HS;
    constexpr int SEG = T::SEG, R = T::R;
    constexpr int PR = 2 * R + 2, PCW = WS + 1, RSW = PCW + 1;    // patch rows; column pairs per row; their padded count
    constexpr int CHW = SEG * PR * RSW;                           // dwords per channel per plane
    constexpr int BPL = CLC * CHW;
    constexpr int BUF = 3 * BPL;
    constexpr int TM = BM / 64, TN = BN / 64;
    constexpr int PE = SEG * PR * PCW, NQ = (PE + 255) / 256;
    constexpr int S = CLC * TM * TN;            // micro steps (six MFMAs each) per chunk
    constexpr int NST = CLC * NQ;               // staging items per thread per chunk
    static_assert(BM % 64 == 0 && BN % 64 == 0 && CLC % 2 == 0, "tile shape");
    __shared__ __attribute__((aligned(16))) uint32_t lds[2 * BUF];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, j = lane & 31, h = lane >> 5;
    const int nmt = Cs / BM, npt = T::ntiles(N);
    const int b = xcd_remap(blockIdx.x, nmt * npt);
    const int ptile = b / nmt, cs0 = (b % nmt) * BM;
    int n0, oh0;
    T::origin(ptile, n0, oh0);

    // patch column pair pw holds input columns 2pw-1 and 2pw
    uint32_t off0[NQ], off1[NQ];
    int loff[NQ];
    bool val0[NQ], val1[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int e = tid + 256 * q;
        const int seg = e / (PR * PCW), rem = e % (PR * PCW), pr = rem / PCW, pw = rem % PCW;
        const int n = n0 + seg, ih = 2 * oh0 - 1 + pr;
        const bool row_ok = (e < PE) && n < N && ih >= 0 && ih < HL;
        val0[q] = row_ok && pw >= 1;
        val1[q] = row_ok && pw < WS;
        const uint32_t base = (uint32_t)((n * Cl * HL + ih) * WL + 2 * pw);
        off0[q] = val0[q] ? base - 1u : 0u;
        off1[q] = val1[q] ? base : 0u;
        // threads past the patch write a padding slot
        loff[q] = (e < PE) ? ((seg * PR + pr) >> 1) * (2 * RSW) + 2 * pw + (pr & 1) : 2 * PCW;
    }
    float breg[2][CLC][NQ][2];
    auto gload = [&](auto setc, int cl0) {
        constexpr int ST = decltype(setc)::value;
#pragma unroll
        for (int c = 0; c < CLC; ++c) {
            const float* Xc = X + (int64_t)(cl0 + c) * HL * WL;
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                breg[ST][c][q][0] = Xc[off0[q]];
                breg[ST][c][q][1] = Xc[off1[q]];
            }
        }
    };
    // staging item `it` of register set ST into buffer `buf`
    auto stage = [&](auto setc, auto itc, uint32_t* buf) {
        constexpr int ST = decltype(setc)::value, it = decltype(itc)::value;
        constexpr int c = it / NQ, q = it % NQ;
        uint32_t p[3];
        split3(val0[q] ? breg[ST][c][q][0] : 0.f, val1[q] ? breg[ST][c][q][1] : 0.f, p);
#pragma unroll
        for (int i = 0; i < 3; ++i) buf[i * BPL + c * CHW + loff[q]] = p[i];
    };

    uint32_t abase[TM];
    int bbase[TN];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) abase[tm] = (uint32_t)((cs0 + wm * (BM / 2) + tm * 32 + j) * 16 + 8 * h);
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
        const int pt = wn * (BN / 2) + tn * 32 + j;
        const int seg = pt / (R * WS), rem = pt % (R * WS), ohl = rem / WS, ow = rem % WS;
        bbase[tn] = (seg * (PR / 2) + ohl + h) * (2 * RSW) + 2 * ow;
    }
    f32x16 acc[TM][TN], acl[TM][TN];      // leading-term sums, correction-term sums
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[tm][tn][r] = 0.f; acl[tm][tn][r] = 0.f; }

    u32x4 a[2][TM][3], bo[2][TN][3];
    const size_t wplane = (size_t)Cl * Cs * 16;
    auto aload = [&](auto setc, int cl) {          // weight fragments of channel cl
        constexpr int set = decltype(setc)::value;
        const uint16_t* Wc = WP + (size_t)cl * Cs * 16;
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int i = 0; i < 3; ++i) a[set][tm][i] = *reinterpret_cast<const u32x4*>(Wc + i * wplane + abase[tm]);
    };
    auto ldops = [&](const uint32_t* buf, auto cc) {
        constexpr int c = decltype(cc)::value, set = c & 1;
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const u32x2* p = reinterpret_cast<const u32x2*>(buf + i * BPL + c * CHW + bbase[tn]);
                const u32x2 lo = p[0], hi = p[1];
                u32x4 v = {lo[0], lo[1], hi[0], hi[1]};
                bo[set][tn][i] = v;
            }
    };

    // gridDim.y > 1: the channel range is also split across blocks, which then add into a zeroed Y
    const int nk = Cl / (CLC * (int)gridDim.y);       // even (checked by the launcher)
    const int kc0 = blockIdx.y * nk;
    const int cl_last = (kc0 + nk) * CLC - 1;
    auto chunk_cl = [&](int kc) { return (kc0 + (kc < nk ? kc : nk - 1)) * CLC; };

    auto chunk = [&](auto setc, int clbase, const uint32_t* cur, uint32_t* nxt) {
        ldops(cur, IC<0>{});
        static_for<S>([&](auto sc) {
            constexpr int s = decltype(sc)::value, c = s / (TM * TN), t = s % (TM * TN), tm = t / TN, tn = t % TN;
            if constexpr (t == 0) {
                if constexpr (c + 1 < CLC) ldops(cur, IC<c + 1>{});
                aload(IC<(c + 1) & 1>{}, min(clbase + c + 1, cl_last));
            }
            mfma_x3(a[c & 1][tm], bo[c & 1][tn], acc[tm][tn], acl[tm][tn]);
            constexpr int lo = s * NST / S, hi = (s + 1) * NST / S;
            static_for<hi - lo>([&](auto ic) { stage(setc, IC<lo + decltype(ic)::value>{}, nxt); });
        });
    };

    // prologue: chunk 0 into buffer 0, chunk 1 into register set 0, first weight fragments
    gload(IC<0>{}, chunk_cl(0));
    aload(IC<0>{}, kc0 * CLC);
    static_for<NST>([&](auto ic) { stage(IC<0>{}, ic, lds); });
    gload(IC<0>{}, chunk_cl(1));
    __syncthreads();
    for (int kc = 0; kc < nk; kc += 2) {
        gload(IC<1>{}, chunk_cl(kc + 2));
        __builtin_amdgcn_sched_barrier(0);      // keep the loads ahead of the chunk's MFMAs
        chunk(IC<0>{}, (kc0 + kc) * CLC, lds, lds + BUF);
        __syncthreads();
        gload(IC<0>{}, chunk_cl(kc + 3));
        __builtin_amdgcn_sched_barrier(0);
        chunk(IC<1>{}, (kc0 + kc + 1) * CLC, lds + BUF, lds);
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
                const float v = acc[tm][tn][r] + acl[tm][tn][r];
                if (gridDim.y > 1) atomicAdd(dst, v);
                else *dst = v;
            }
    }
}

// ------------------------------------------------------------------------------------------------
// UP: the four sub-pixel phases of a tile go to four waves; NC such groups (different 32-cl row tiles)
// share the block's S patch.  A phase is a 2x2 stride-1 convolution, evaluated tap by tap with
// K = 16 cs per MFMA; a wave keeps the weight fragments of its four taps in registers and refills each
// for the next K step right after using it.  LDS holds the S patch channel-innermost
// [plane][patch pixel][16 cs], double-buffered, one barrier per K step; the two 16-byte halves of a
// 32-byte pixel row are swapped on every other group of four pixels so that the eight lanes of a b128
// access phase hit distinct banks.
// Tap kh feeds output-row phase PH[kh] from S row a + DH[kh]:  kh 0:(1,+1) 1:(0,0) 2:(1,0) 3:(0,-1).
// ------------------------------------------------------------------------------------------------
template <int HS, int BN, int NC>
__global__ __launch_bounds__(256 * NC) void conv_up_x3_kernel(const float* __restrict__ Y, const uint16_t* __restrict__ WP,
                                                              float* __restrict__ X, int N, int Cs, int Cl) {
    using T = PixTile<HS, BN>;
    constexpr int WS = HS, HL = 2 * HS, WL = 2 * HS;
    constexpr int SEG = T::SEG, R = T::R;
    constexpr int PR = R + 2, PC = WS + 2, NP = SEG * PR * PC;      // patch pixels
    constexpr int BM = 32 * NC, TN = BN / 32, NT = 256 * NC;
    constexpr int BPL = NP * 8;                                     // dwords per plane
    constexpr int NI = NP * 2, NQ = (NI + NT - 1) / NT;             // (pixel, channel octet) items per thread
    __shared__ __attribute__((aligned(16))) uint32_t lds[2 * 3 * BPL];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ph = (wave >> 1) & 1, pw = wave & 1, wc = wave >> 2, j = lane & 31, h = lane >> 5;
    const int nmt = Cl / BM, npt = T::ntiles(N);
    const int b = xcd_remap(blockIdx.x, nmt * npt);
    const int ptile = b / nmt, cl0 = (b % nmt) * BM + wc * 32;
    int n0, oh0;
    T::origin(ptile, n0, oh0);

    // patch item e: (pixel, octet)
    uint32_t boff[NQ];
    int blds[NQ];
    bool bval[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int e = tid + NT * q, p = e >> 1, oct = e & 1;
        const int seg = p / (PR * PC), rem = p % (PR * PC), pr = rem / PC, pc = rem % PC;
        const int n = n0 + seg, a = oh0 - 1 + pr, bb = pc - 1;
        bval[q] = (e < NI) && n < N && a >= 0 && a < HS && bb >= 0 && bb < WS;
        boff[q] = bval[q] ? (uint32_t)(((n * Cs + oct * 8) * HS + a) * WS + bb) : 0u;
        blds[q] = (e < NI) ? p * 8 + ((oct ^ (p >> 2)) & 1) * 4 : -1;
    }
    float breg[NQ][8];
    auto gload = [&](int cs0) {
        const float* Yc = Y + (int64_t)cs0 * HS * WS;
#pragma unroll
        for (int q = 0; q < NQ; ++q)
#pragma unroll
            for (int i = 0; i < 8; ++i) breg[q][i] = Yc[boff[q] + (bval[q] ? i * HS * WS : 0)];
    };
    auto lstore = [&](uint32_t* Bs) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            uint32_t pk[3][4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                uint32_t p3[3];
                split3(bval[q] ? breg[q][2 * i] : 0.f, bval[q] ? breg[q][2 * i + 1] : 0.f, p3);
                pk[0][i] = p3[0]; pk[1][i] = p3[1]; pk[2][i] = p3[2];
            }
            if (blds[q] >= 0) {
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    u32x4 v = {pk[i][0], pk[i][1], pk[i][2], pk[i][3]};
                    *reinterpret_cast<u32x4*>(Bs + i * BPL + blds[q]) = v;
                }
            }
        }
    };

    // this wave's four taps: global offsets of the weight fragment, LDS offsets of each pixel tile's patch granule
    uint32_t aofs[4];
    int bofs[4][TN];
    const size_t wplane = (size_t)Cs * Cl * 16;
#pragma unroll
    for (int th = 0; th < 2; ++th)
#pragma unroll
        for (int tw = 0; tw < 2; ++tw) {
            const int kh = ph ? (th ? 0 : 2) : (th ? 3 : 1), dh = ph ? th : -th;
            const int kw = pw ? (tw ? 0 : 2) : (tw ? 3 : 1), dw = pw ? tw : -tw;
            aofs[th * 2 + tw] = (uint32_t)(((kh * 4 + kw) * (Cs / 16) * Cl + cl0 + j) * 16 + 8 * h);
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) {
                const int pt = tn * 32 + j;
                const int seg = pt / (R * WS), rem = pt % (R * WS), al = rem / WS, bl = rem % WS;
                const int p = (seg * PR + al + 1 + dh) * PC + bl + 1 + dw;
                bofs[th * 2 + tw][tn] = p * 8 + ((h ^ (p >> 2)) & 1) * 4;
            }
        }
    f32x16 acc[TN], acl[TN];
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[tn][r] = 0.f; acl[tn][r] = 0.f; }

    // gridDim.y > 1: the cs range is split across blocks, which then add into a zeroed X
    const int nk = Cs / (16 * (int)gridDim.y), k0 = blockIdx.y * nk;
    u32x4 a[4][3];
    auto aload = [&](int t, int ks) {                 // weight fragments of tap t, K step ks
        const uint16_t* Wk = WP + (size_t)ks * Cl * 16;
#pragma unroll
        for (int i = 0; i < 3; ++i) a[t][i] = *reinterpret_cast<const u32x4*>(Wk + i * wplane + aofs[t]);
    };
    gload(k0 * 16);
#pragma unroll
    for (int t = 0; t < 4; ++t) aload(t, k0);
    lstore(lds);
    if (nk > 1) gload((k0 + 1) * 16);
    __syncthreads();
    for (int kc = 0; kc < nk; ++kc) {
        const uint32_t* Bs = lds + (kc & 1) * 3 * BPL;
        uint32_t* Bn = lds + ((kc + 1) & 1) * 3 * BPL;
        const int knext = k0 + min(kc + 1, nk - 1);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            u32x4 bo[TN][3];
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                for (int i = 0; i < 3; ++i) bo[tn][i] = *reinterpret_cast<const u32x4*>(Bs + i * BPL + bofs[t][tn]);
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) mfma_x3(a[t], bo[tn], acc[tn], acl[tn]);
            aload(t, knext);
            if (t == 0 && kc + 1 < nk) lstore(Bn);                       // K step kc+1: registers -> other buffer
            if (t == 1 && kc + 2 < nk) gload((k0 + kc + 2) * 16);        // K step kc+2: global -> registers
        }
        __syncthreads();
    }

#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
        const int pt = tn * 32 + j;
        const int seg = pt / (R * WS), rem = pt % (R * WS), al = rem / WS, bl = rem % WS;
        const int n = n0 + seg;
        if (n >= N) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int cl = cl0 + (r & 3) + 8 * (r >> 2) + 4 * h;
            float* dst = X + (((int64_t)n * Cl + cl) * HL + 2 * (oh0 + al) + ph) * WL + 2 * bl + pw;
            if (gridDim.y > 1) atomicAdd(dst, acc[tn][r] + acl[tn][r]);
            else *dst = acc[tn][r] + acl[tn][r];
        }
    }
}

template <int HS, int BM, int BN, int CLC>
int launch_down_tile(const float* X, const uint16_t* WP, float* Y, int N, int Cs, int Cl, int ks, hipStream_t st) {
    if (Cs % BM || Cl % (2 * CLC * ks)) return PGAN_EINVAL;
    const dim3 g((Cs / BM) * PixTile<HS, BN>::ntiles(N), ks);
    conv_down_x3_kernel<HS, BM, BN, CLC><<<g, 256, 0, st>>>(X, WP, Y, N, Cs, Cl);
    return pgan_launch_status();
}

int g_dn_variant = 0, g_dn_ks = 0, g_up_variant = 0, g_up_ks = 0;   // tuning overrides (0 = by size)

template <int HS>
int launch_down(const float* X, const uint16_t* WP, float* Y, int N, int Cs, int Cl, hipStream_t st) {
    if ((int64_t)N * Cl * HS * HS * 4 >= (1ll << 31) || (int64_t)Cs * Cl * 48 >= (1ll << 31)) return PGAN_EINVAL;
    int variant = g_dn_variant, ks = g_dn_ks;
    if (variant == 0 || ks < 1) {
        // 128x64 tiles when they still give two blocks per CU, else 64x64 tiles (three blocks per CU);
        // a grid that leaves CUs idle also splits K over 2 or 4 blocks
        const int b4 = (Cs / 64) * PixTile<HS, 64>::ntiles(N);
        int v = 4, k = b4 >= 512 ? 1 : (b4 * 2 >= 512 ? 2 : 4);
        if (Cs % 128 == 0 && b4 >= 1024) { v = 2; k = 1; }
        while (k > 1 && Cl % (8 * k)) k >>= 1;
        if (variant == 0) variant = v;
        if (ks < 1) ks = k;
    }
    if (ks > 1 && hipMemsetAsync(Y, 0, sizeof(float) * (size_t)N * Cs * HS * HS, st) != hipSuccess) return PGAN_ELAUNCH;
    switch (variant) {
        case 1: return launch_down_tile<HS, 128, 128, 4>(X, WP, Y, N, Cs, Cl, ks, st);
        case 2: return launch_down_tile<HS, 128, 64, 4>(X, WP, Y, N, Cs, Cl, ks, st);
        case 3: return launch_down_tile<HS, 64, 128, 4>(X, WP, Y, N, Cs, Cl, ks, st);
        case 4: return launch_down_tile<HS, 64, 64, 4>(X, WP, Y, N, Cs, Cl, ks, st);
        default: return PGAN_EINVAL;
    }
}

template <int HS>
int launch_up(const float* Y, const uint16_t* WP, float* X, int N, int Cs, int Cl, hipStream_t st) {
    if (Cl % 32 || Cs % 16) return PGAN_EINVAL;
    if ((int64_t)N * Cs * HS * HS >= (1ll << 31) || (int64_t)Cs * Cl * 48 >= (1ll << 31)) return PGAN_EINVAL;
    const uint16_t* up = WP + (size_t)3 * Cs * Cl * 16;
    // 64-pixel tiles; 32-pixel tiles when those would give fewer than two blocks per CU
    int variant = g_up_variant ? g_up_variant : ((Cl / 32) * PixTile<HS, 64>::ntiles(N) < 512 ? 5 : 1);
    if ((variant == 3 || variant == 4) && Cl % 64) variant -= 2;
    const int nc = (variant == 3 || variant == 4) ? 2 : 1, bn = variant == 5 ? 32 : ((variant & 1) ? 64 : 128);
    const int blocks = (Cl / (32 * nc)) * (bn == 128 ? PixTile<HS, 128>::ntiles(N)
                                                     : (bn == 64 ? PixTile<HS, 64>::ntiles(N) : PixTile<HS, 32>::ntiles(N)));
    int ks = g_up_ks ? g_up_ks : 1;
    while (ks > 1 && Cs % (32 * ks)) ks >>= 1;
    if (ks > 1 && hipMemsetAsync(X, 0, sizeof(float) * (size_t)N * Cl * HS * HS * 4, st) != hipSuccess) return PGAN_ELAUNCH;
    const dim3 g(blocks, ks);
    switch (variant) {
        case 1: conv_up_x3_kernel<HS, 64, 1><<<g, 256, 0, st>>>(Y, up, X, N, Cs, Cl); break;
        case 2: conv_up_x3_kernel<HS, 128, 1><<<g, 256, 0, st>>>(Y, up, X, N, Cs, Cl); break;
        case 3: conv_up_x3_kernel<HS, 64, 2><<<g, 512, 0, st>>>(Y, up, X, N, Cs, Cl); break;
        case 4: conv_up_x3_kernel<HS, 128, 2><<<g, 512, 0, st>>>(Y, up, X, N, Cs, Cl); break;
        case 5: conv_up_x3_kernel<HS, 32, 1><<<g, 256, 0, st>>>(Y, up, X, N, Cs, Cl); break;
        default: return PGAN_EINVAL;
    }
    return pgan_launch_status();
}

}  // namespace

extern "C" {

int pgan_x3_set_option(const char* name, int value) {
    if (!name) return PGAN_EINVAL;
    auto eq = [](const char* x, const char* y) { while (*x && *x == *y) { ++x; ++y; } return *x == *y; };
    if (eq(name, "down_variant") && value >= 0 && value <= 4) { g_dn_variant = value; return PGAN_OK; }
    if (eq(name, "down_ks") && value >= 0 && value <= 8) { g_dn_ks = value; return PGAN_OK; }
    if (eq(name, "up_variant") && value >= 0 && value <= 5) { g_up_variant = value; return PGAN_OK; }
    if (eq(name, "up_ks") && value >= 0 && value <= 8) { g_up_ks = value; return PGAN_OK; }
    return PGAN_EINVAL;
}

// Bytes of the prepared-weight buffer for W[Cs][Cl][4][4].
long long pgan_conv_x3_prep_bytes(int Cs, int Cl) { return (long long)Cs * Cl * 16 * 2 * 6; }

// Split W[Cs][Cl][4][4] (fp32) into the bf16 planes the x3 kernels read.  Run again whenever W changes.
int pgan_conv_x3_prep(const float* W, void* planes, int Cs, int Cl, hipStream_t st) {
    if (Cs <= 0 || Cl <= 0) return PGAN_EINVAL;
    uint16_t* dn = static_cast<uint16_t*>(planes);
    prep_kernel<<<(Cs * Cl + 255) / 256, 256, 0, st>>>(W, dn, dn + (size_t)3 * Cs * Cl * 16, Cs, Cl);
    return pgan_launch_status();
}

// S[N,Cs,H,H] = conv2d(L[N,Cl,2H,2H], W, stride 2, pad 1) with W given as prepared planes.  H = hs.
int pgan_conv_down_x3(const float* L, const void* planes, float* S, int N, int Cs, int Cl, int hs, hipStream_t st) {
    if (N <= 0) return PGAN_OK;
    const uint16_t* wp = static_cast<const uint16_t*>(planes);
    switch (hs) {
        case 4: return launch_down<4>(L, wp, S, N, Cs, Cl, st);
        case 8: return launch_down<8>(L, wp, S, N, Cs, Cl, st);
        case 16: return launch_down<16>(L, wp, S, N, Cs, Cl, st);
        default: return PGAN_EINVAL;
    }
}

// L[N,Cl,2H,2H] = conv_transpose2d(S[N,Cs,H,H], W, stride 2, pad 1) with W given as prepared planes.
int pgan_conv_up_x3(const float* S, const void* planes, float* L, int N, int Cs, int Cl, int hs, hipStream_t st) {
    if (N <= 0) return PGAN_OK;
    const uint16_t* wp = static_cast<const uint16_t*>(planes);
    switch (hs) {
        case 4: return launch_up<4>(S, wp, L, N, Cs, Cl, st);
        case 8: return launch_up<8>(S, wp, L, N, Cs, Cl, st);
        case 16: return launch_up<16>(S, wp, L, N, Cs, Cl, st);
        default: return PGAN_EINVAL;
    }
}

}  // extern "C"
