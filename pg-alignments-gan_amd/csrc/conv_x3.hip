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

// ------------------------------------------------------------------------------------------------
// Two ways of splitting an fp32 operand, chosen at compile time (this source is compiled twice):
//   default          three bf16 planes, six products            -> entry points pgan_conv_*_x3
//   PGAN_SPLIT_F16   two fp16 planes, three products            -> entry points pgan_conv_*_h2
// fp16 carries 11 significant bits: x = h1 + 2^-11 h2 with h1 = fp16(x), h2 = fp16(2^11 (x - h1)) leaves a
// residual below 2^-22 |x|, and x y = h1 g1 + 2^-11 (h1 g2 + h2 g1) drops only h2 g2 <= 2^-22 |x y|: fp32-class
// accuracy (about the error of an fp32 MFMA dot product) from half the matrix work and two thirds of the operand
// traffic.  Storing the second plane scaled by 2^11 keeps it in the range of the first, so only the magnitude of x
// itself has to fit fp16's exponent range: operands are scaled by a power of two chosen from their largest
// magnitude (exact, undone on the accumulators).
// ------------------------------------------------------------------------------------------------
#ifdef PGAN_SPLIT_F16
constexpr int NPL = 2;
constexpr bool F16 = true;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
#define FN_PREP_BYTES pgan_conv_h2_prep_bytes
#define FN_PREP pgan_conv_h2_prep
#define FN_PREP_MULTI pgan_conv_h2_prep_multi
#define FN_SET_OPTION pgan_h2_set_option
#define FN_DOWN pgan_conv_down_h2
#define FN_DOWN_ZEROED pgan_conv_down_h2_zeroed
#define FN_DOWN_ADDS pgan_conv_down_h2_adds
#define FN_UP pgan_conv_up_h2
#define FN_WGRAD_WS pgan_conv_wgrad_h2_workspace_floats
#define FN_WGRAD pgan_conv_wgrad_h2
#else
constexpr int NPL = 3;
constexpr bool F16 = false;
#define FN_PREP_BYTES pgan_conv_x3_prep_bytes
#define FN_PREP pgan_conv_x3_prep
#define FN_PREP_MULTI pgan_conv_x3_prep_multi
#define FN_SET_OPTION pgan_x3_set_option
#define FN_DOWN pgan_conv_down_x3
#define FN_DOWN_ZEROED pgan_conv_down_x3_zeroed
#define FN_DOWN_ADDS pgan_conv_down_x3_adds
#define FN_UP pgan_conv_up_x3
#define FN_WGRAD_WS pgan_conv_wgrad_x3_workspace_floats
#define FN_WGRAD pgan_conv_wgrad_x3
#endif
constexpr float LO_SCALE = 2048.f;      // 2^11: the second fp16 plane is stored times this

// (x, y) -> NPL packed 16-bit pairs; low half belongs to x.
__device__ __forceinline__ void split3(float x, float y, uint32_t (&p)[NPL]) {
    f32x2 v = {x, y};
#if defined(PGAN_SPLIT_F16) && defined(PGAN_ABLATE_SPLIT)
    // timing experiment only (results are meaningless): no split arithmetic, as if the operand arrived as prepared planes
    p[0] = __builtin_bit_cast(uint32_t, x) & 0x3BFF3BFFu;
    p[1] = __builtin_bit_cast(uint32_t, y) & 0x3BFF3BFFu;
#elif defined(PGAN_SPLIT_F16)
    const f16x2 h1 = __builtin_convertvector(v, f16x2);
    p[0] = __builtin_bit_cast(uint32_t, h1);
    const f32x2 r = (v - __builtin_convertvector(h1, f32x2)) * LO_SCALE;
    p[1] = __builtin_bit_cast(uint32_t, __builtin_convertvector(r, f16x2));
#else
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const uint32_t u = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
        p[i] = u;
        if (i < 2) {
            v.x -= __builtin_bit_cast(float, u << 16);
            v.y -= __builtin_bit_cast(float, u & 0xffff0000u);
        }
    }
#endif
}

__device__ __forceinline__ f32x16 mfma_16(u32x4 a, u32x4 b, f32x16 c) {
#ifdef PGAN_SPLIT_F16
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
#else
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
#endif
}

// The significant partial products of the split operands.  The leading term a1*b1 has its own
// accumulator: the matrix pipe truncates (toward -inf) the bits of a product that fall below the
// accumulator's last place, so correction terms added straight into the large sum would each lose up to
// 2^-9 of their value in one direction -- a bias that grows with K (bf16: measured -2.7e-7 relative at
// K = 4096 against -1e-9 with the split accumulators, and a 2.5x larger rms error).  In the fp16 split the
// correction accumulator also carries the 2^11 of the second planes.
__device__ __forceinline__ void mfma_x3(const u32x4 (&a)[NPL], const u32x4 (&b)[NPL], f32x16& hi, f32x16& lo) {
#ifdef PGAN_SPLIT_F16
    lo = mfma_16(a[1], b[0], lo);
    lo = mfma_16(a[0], b[1], lo);
    hi = mfma_16(a[0], b[0], hi);
#else
    lo = mfma_16(a[2], b[0], lo);
    lo = mfma_16(a[0], b[2], lo);
    lo = mfma_16(a[1], b[1], lo);
    lo = mfma_16(a[1], b[0], lo);
    lo = mfma_16(a[0], b[1], lo);
    hi = mfma_16(a[0], b[0], hi);
#endif
}
// leading-term sum + correction-term sum
__device__ __forceinline__ float combine(float hi, float lo) { return F16 ? fmaf(lo, 1.f / LO_SCALE, hi) : hi + lo; }

// ---- range of the fp16 split.  An operand set (the activations one block stages over its whole K loop; a weight
// tensor) is multiplied by 2^-e with e chosen from its largest magnitude so that the scaled values peak in
// [2^13, 2^14): no overflow (fp16 reaches 2^16), and every element keeps an absolute precision of 2^-36 of that
// peak through the two planes.  Powers of two: exact; undone on the fp32 results.
constexpr int WHDR_BYTES = 256;          // header of a prepared-weight buffer: word 0 = bits of the largest |w| seen
__device__ __forceinline__ int scale_exp(float amax) {
    const int eb = (int)((__builtin_bit_cast(uint32_t, amax) >> 23) & 0xffu);      // biased exponent, 0 for 0 / subnormal
    return max(eb - 127 - 13, -126);
}
__device__ __forceinline__ float pow2f(int e) {       // 2^e for e in [-126, 127]
    return __builtin_bit_cast(float, (uint32_t)(min(max(e, -126), 127) + 127) << 23);
}
// 2^-e, also for the few e beyond float's exponent range of a single factor
__device__ __forceinline__ float scaled(float v, int e_neg) { return v * pow2f(-e_neg); }
// largest of v (>= 0) over the block; sh: LDS scratch of >= blockDim/64 floats, free to reuse after the call
__device__ __forceinline__ float block_amax(float v, float* sh) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    const int nw = (int)(blockDim.x >> 6);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    float m = 0.f;
    for (int w = 0; w < nw; ++w) m = fmaxf(m, sh[w]);
    __syncthreads();
    return m;
}
// Upper bounds of an activation operand's magnitude supplied by whoever produced it (the norm kernels write them), so
// that a block need not scan its operands before staging them.  per_sample > 0: p[n * per_sample + i], i < per_sample,
// bound sample n of the launch; per_sample == 0: every one of p[0 .. count) bounds the whole operand.  p == nullptr:
// no bounds given, the kernel scans.
struct Bound {
    const float* p;
    int per_sample, count;
};
// largest bound over samples [n_lo, n_hi) (mapped through `remap`) as seen by this thread; reduce with block_amax
template <class F>
__device__ __forceinline__ float bound_max(const Bound& b, int n_lo, int n_hi, F remap) {
    float m = 0.f;
    if (b.per_sample > 0) {
        const int cnt = (n_hi - n_lo) * b.per_sample;
        for (int i = threadIdx.x; i < cnt; i += blockDim.x)
            m = fmaxf(m, b.p[(int64_t)remap(n_lo + i / b.per_sample) * b.per_sample + i % b.per_sample]);
    } else {
        for (int i = threadIdx.x; i < b.count; i += blockDim.x) m = fmaxf(m, b.p[i]);
    }
    return m;
}
// blk of nblk blocks over the n4 float4 of W
__device__ __forceinline__ void weight_amax_body(const float* __restrict__ W, uint32_t* __restrict__ hdr, int64_t n4, int blk,
                                                 int nblk) {
    __shared__ float sh[4];
    float m = 0.f;
    for (int64_t i = (int64_t)blk * 256 + threadIdx.x; i < n4; i += (int64_t)nblk * 256) {
        const f32x4 v = reinterpret_cast<const f32x4*>(W)[i];
        m = fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(m, fmaxf(fabsf(v[2]), fabsf(v[3]))));
    }
    m = block_amax(m, sh);
    if (threadIdx.x == 0) atomicMax(hdr, __builtin_bit_cast(uint32_t, m));       // non-negative floats order like their bits
}
__global__ __launch_bounds__(256) void weight_amax_kernel(const float* __restrict__ W, uint32_t* __restrict__ hdr, int64_t n4) {
    weight_amax_body(W, hdr, n4, blockIdx.x, gridDim.x);
}
// Several weight tensors per launch (the stages of one network are prepared together): block ranges [first[t], first[t+1]).
constexpr int PREP_MAX = 6;
struct PrepSet {
    const float* W[PREP_MAX];
    unsigned char* planes[PREP_MAX];       // start of each prepared-weight buffer (header first in the fp16 split)
    int Cs[PREP_MAX], Cl[PREP_MAX];
    int first_amax[PREP_MAX + 1], first_prep[PREP_MAX + 1];
    int n;
};
__global__ __launch_bounds__(256) void weight_amax_multi_kernel(PrepSet ps) {
    int t = 0;
    while (t + 1 < ps.n && (int)blockIdx.x >= ps.first_amax[t + 1]) ++t;
    weight_amax_body(ps.W[t], reinterpret_cast<uint32_t*>(ps.planes[t]), (int64_t)ps.Cs[t] * ps.Cl[t] * 4,
                     blockIdx.x - ps.first_amax[t], ps.first_amax[t + 1] - ps.first_amax[t]);
}

// ------------------------------------------------------------------------------------------------
// Weight preparation: W[Cs][Cl][16] fp32 -> bf16 planes in the two fragment orders the kernels read:
//   dn[p][cl][cs][tap']           DOWN: rows cs, one K step = the 16 taps of channel cl, stored in the order
//                                 the patch is read: tap' = 8(kh>>1) + 4(kw>>1) + 2(kh&1) + (kw&1)
//   up[p][tap][cs/16][cl][cs%16]  UP:   rows cl, one K step = 16 cs at a fixed tap
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void prep_body(const float* __restrict__ W, uint16_t* __restrict__ dn, uint16_t* __restrict__ up,
                                          int Cs, int Cl, const uint32_t* __restrict__ whdr, int blk) {
    const int idx = blk * 256 + threadIdx.x;
    if (idx >= Cs * Cl) return;
    const float sw = F16 ? pow2f(-scale_exp(__builtin_bit_cast(float, whdr[0]))) : 1.f;
    const int cs = idx % Cs, cl = idx / Cs;
    const size_t plane = (size_t)Cs * Cl * 16;
    const float* w = W + ((size_t)cs * Cl + cl) * 16;
    uint32_t pk[NPL][8];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(w + 4 * q) * sw;
        uint32_t a[NPL], b[NPL];
        split3(v[0], v[1], a);
        split3(v[2], v[3], b);
#pragma unroll
        for (int i = 0; i < NPL; ++i) { pk[i][2 * q] = a[i]; pk[i][2 * q + 1] = b[i]; }
    }
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
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

__global__ __launch_bounds__(256) void prep_kernel(const float* __restrict__ W, uint16_t* __restrict__ dn,
                                                   uint16_t* __restrict__ up, int Cs, int Cl,
                                                   const uint32_t* __restrict__ whdr) {
    prep_body(W, dn, up, Cs, Cl, whdr, blockIdx.x);
}
__global__ __launch_bounds__(256) void prep_multi_kernel(PrepSet ps) {
    int t = 0;
    while (t + 1 < ps.n && (int)blockIdx.x >= ps.first_prep[t + 1]) ++t;
    unsigned char* base = ps.planes[t];
    uint16_t* dn = reinterpret_cast<uint16_t*>(base + (F16 ? WHDR_BYTES : 0));
    prep_body(ps.W[t], dn, dn + (size_t)NPL * ps.Cs[t] * ps.Cl[t] * 16, ps.Cs[t], ps.Cl[t],
              F16 ? reinterpret_cast<const uint32_t*>(base) : nullptr, blockIdx.x - ps.first_prep[t]);
}

// ------------------------------------------------------------------------------------------------
// DOWN: GEMM M = Cs (BM per block), N = pixels (BN per block), K = Cl*16; one MFMA K step is one
// input channel's 16 taps.  Weight fragments come straight from global memory one channel ahead.
// LDS holds, per bf16 plane, the input patches [CLC][row pair][column pair][2 rows] (two bf16 per dword, so
// a lane's eight K values are 16 contiguous bytes), double-buffered
// with one barrier per chunk of CLC channels: chunk k+1 goes from registers to LDS (split on the way)
// between the MFMAs of chunk k while chunk k+2 is in flight from global memory into a second register set.
// ------------------------------------------------------------------------------------------------
// KG = 2: two groups of four waves share the block's output tile and take alternate halves of the block's channel
// range, each with its own LDS buffers; group 1 hands its sums to group 0 through LDS at the end (a fixed order:
// deterministic).  gridDim.y = 2 splits the range once more over two blocks that add into a zeroed Y -- two
// addends, so the result does not depend on which block comes first.
// PGAN_ABL_DOWN (debugging builds only, results are wrong): leave out parts of the DOWN kernel's main loop -- 1 the MFMAs
// (operands are folded into the sums with one integer add so that their loads stay), 2 the weight-fragment loads, 4 the
// staging stores into LDS, 16 the LDS reads.  Used to find which part of this kernel disturbs a neighbour.
#ifndef PGAN_ABL_DOWN
#define PGAN_ABL_DOWN 0
#endif
template <int HS, int BM, int BN, int CLC, int KG>
__global__ __launch_bounds__(256 * KG, KG == 1 ? ((BN == 64) ? 3 : 2) : 1) void conv_down_x3_kernel(const float* __restrict__ X, const uint16_t* __restrict__ WP,
                                                           float* __restrict__ Y, int N, int Cs, int Cl,
                                                           const uint32_t* __restrict__ whdr, Bound bx, int cs_major) {
    using T = PixTile<HS, BN>;
    constexpr int WS = HS, HL = 2 * HS, WL = 2 * HS;
    constexpr int SEG = T::SEG, R = T::R;
    constexpr int PR = 2 * R + 2, PCW = WS + 1, RSW = PCW + 1;    // patch rows; column pairs per row; their padded count
    constexpr int CHW = SEG * PR * RSW;                           // dwords per channel per plane
    constexpr int BPL = CLC * CHW;
    constexpr int BUF = NPL * BPL;
    // Waves as WR row groups x WC pixel groups.  A weight fragment goes from L2 into the registers of every wave that uses it,
    // and that path (64 B/clk per CU) is what the kernel saturates first: with 128 rows every wave takes its own 32
    // rows over all the block's pixels, so no fragment is loaded twice.
    constexpr int WR = (BM >= 128) ? 4 : 2, WC = 4 / WR;
    constexpr int TM = BM / (32 * WR), TN = BN / (32 * WC);
    constexpr int PE = SEG * PR * PCW, NQ = (PE + 255) / 256;
    constexpr int S = CLC * TM * TN;            // micro steps (six MFMAs each) per chunk
    constexpr int NST = CLC * NQ;               // staging items per thread per chunk
    static_assert(BM % 64 == 0 && BN % 64 == 0 && CLC % 2 == 0, "tile shape");
    constexpr int LDS_WORDS = (KG > 1 && BM * BN > 2 * BUF * KG) ? BM * BN : 2 * BUF * KG;
    __shared__ __attribute__((aligned(16))) uint32_t lds_all[LDS_WORDS];

    const int grp = (KG > 1) ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 8)) : 0;      // the same in all of a wave: scalar address arithmetic
    uint32_t* const lds = lds_all + grp * 2 * BUF;
    const int tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WC, wn = wave % WC, j = lane & 31, h = lane >> 5;
    const int nmt = Cs / BM, npt = T::ntiles(N);
    const int b = xcd_remap(blockIdx.x, nmt * npt);
    // Each XCD gets a contiguous run of b.  cs_major: a run is one row tile (its weight planes stay in that XCD's L2)
    // over all pixel tiles -- for layers whose weight planes outweigh the activations; otherwise a run is a range of
    // pixel tiles over all row tiles.
    const int ptile = cs_major ? b % npt : b / nmt, cs0 = (cs_major ? b / npt : b % nmt) * BM;
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
        const uint32_t base = (uint32_t)((seg * Cl * HL + ih) * WL + 2 * pw);       // from the block's first sample
        off0[q] = val0[q] ? base - 1u : 0u;
        off1[q] = val1[q] ? base : 0u;
        // threads past the patch write a padding slot
        loff[q] = (e < PE) ? ((seg * PR + pr) >> 1) * (2 * RSW) + 2 * pw + (pr & 1) : 2 * PCW;
    }
    float breg[2][CLC][NQ][2];
    float sx = 1.f;        // fp16 split: the block's activations are staged times this power of two (2^-ex)
    int ex = 0;
    // Loads through buffer descriptors: a lane's byte offset is a 32-bit register fixed before the K loop, the chunk
    // moves a scalar offset -- no vector address arithmetic inside the loop.  X's descriptor starts at the block's first
    // sample and ends with the tensor (reads past it return zero).
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(X) + (int64_t)n0 * Cl * HL * WL, 0, (int)min((int64_t)(N - n0) * Cl * HL * WL * 4, (int64_t)0x7fffffff), 0x00020000);
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(WP), 0, (int)((int64_t)NPL * Cs * Cl * 32), 0x00020000);
    auto gload = [&](auto setc, int cl0) {
        constexpr int ST = decltype(setc)::value;
#pragma unroll
        for (int c = 0; c < CLC; ++c) {
            const int so = (cl0 + c) * (HL * WL * 4);
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                breg[ST][c][q][0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrs, (int)(off0[q] * 4u), so, 0));
                breg[ST][c][q][1] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrs, (int)(off1[q] * 4u), so, 0));
            }
        }
    };
    // staging item `it` of register set ST into buffer `buf`
    auto stage = [&](auto setc, auto itc, uint32_t* buf) {
        constexpr int ST = decltype(setc)::value, it = decltype(itc)::value;
        constexpr int c = it / NQ, q = it % NQ;
        uint32_t p[NPL];
        // (items outside the image hold finite values of other pixels, or the zero of a read past the tensor: times zero)
        split3(breg[ST][c][q][0] * (val0[q] ? sx : 0.f), breg[ST][c][q][1] * (val1[q] ? sx : 0.f), p);
#pragma unroll
        for (int i = 0; i < NPL; ++i) buf[i * BPL + c * CHW + loff[q]] = p[i];
    };

    uint32_t abase[TM];
    int bbase[TN];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) abase[tm] = (uint32_t)((cs0 + wm * (BM / WR) + tm * 32 + j) * 16 + 8 * h);
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
        const int pt = wn * (BN / WC) + tn * 32 + j;
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

    u32x4 a[2][TM][NPL], bo[2][TN][NPL];
    const size_t wplane = (size_t)Cl * Cs * 16;
    auto aload = [&](auto setc, int cl) {          // weight fragments of channel cl
        constexpr int set = decltype(setc)::value;
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int i = 0; i < NPL; ++i)
                a[set][tm][i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(wrs, (int)(abase[tm] * 2u), (cl * Cs * 16 + i * (int)wplane) * 2, 0));
    };
    auto ldops = [&](const uint32_t* buf, auto cc) {
        constexpr int c = decltype(cc)::value, set = c & 1;
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
            for (int i = 0; i < NPL; ++i) {
                const u32x2* p = reinterpret_cast<const u32x2*>(buf + i * BPL + c * CHW + bbase[tn]);
                const u32x2 lo = p[0], hi = p[1];
                u32x4 v = {lo[0], lo[1], hi[0], hi[1]};
                bo[set][tn][i] = v;
            }
    };

    // gridDim.y > 1: the channel range is also split across blocks, which then add into a zeroed Y
    const int nk = Cl / (CLC * (int)gridDim.y * KG);       // even (checked by the launcher)
    const int kc0 = ((int)blockIdx.y * KG + grp) * nk;
    const int cl_last = (kc0 + nk) * CLC - 1;
    auto chunk_cl = [&](int kc) { return (kc0 + (kc < nk ? kc : nk - 1)) * CLC; };

    auto chunk = [&](auto setc, int clbase, const uint32_t* cur, uint32_t* nxt) {
        if constexpr (!(PGAN_ABL_DOWN & 16)) ldops(cur, IC<0>{});
        static_for<S>([&](auto sc) {
            constexpr int s = decltype(sc)::value, c = s / (TM * TN), t = s % (TM * TN), tm = t / TN, tn = t % TN;
            if constexpr (t == 0) {
                if constexpr (c + 1 < CLC && !(PGAN_ABL_DOWN & 16)) ldops(cur, IC<c + 1>{});
                if constexpr (!(PGAN_ABL_DOWN & 2)) aload(IC<(c + 1) & 1>{}, min(clbase + c + 1, cl_last));
            }
            if constexpr (PGAN_ABL_DOWN & 1) {
#pragma unroll
                for (int i = 0; i < NPL; ++i)
                    acc[tm][tn][i] += __builtin_bit_cast(float, (a[c & 1][tm][i][0] + bo[c & 1][tn][i][0]) & 0x3fffffffu);
            } else {
                mfma_x3(a[c & 1][tm], bo[c & 1][tn], acc[tm][tn], acl[tm][tn]);
            }
            constexpr int lo = s * NST / S, hi = (s + 1) * NST / S;
            if constexpr (!(PGAN_ABL_DOWN & 4)) static_for<hi - lo>([&](auto ic) { stage(setc, IC<lo + decltype(ic)::value>{}, nxt); });
        });
    };

    // fp16 split: one pass over this block's patches for their largest magnitude -> the block's scale
    if constexpr (F16) {
        float m = 0.f;
        if (bx.p) m = bound_max(bx, n0, min(n0 + SEG, N), [](int n) { return n; });
        else for (int kc = 0; kc < nk; ++kc) {
            gload(IC<0>{}, chunk_cl(kc));
#pragma unroll
            for (int c = 0; c < CLC; ++c)
#pragma unroll
                for (int q = 0; q < NQ; ++q)
                    m = fmaxf(m, fmaxf(val0[q] ? fabsf(breg[0][c][q][0]) : 0.f, val1[q] ? fabsf(breg[0][c][q][1]) : 0.f));
        }
        ex = scale_exp(block_amax(m, reinterpret_cast<float*>(lds_all)));      // one scale for the whole block
        sx = pow2f(-ex);
    }
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

    // fp16 split: undo the two scales (a product of two in-range powers of two)
    const float unscale = F16 ? pow2f(ex) * pow2f(scale_exp(__builtin_bit_cast(float, whdr[0]))) : 1.f;
    if constexpr (KG > 1) {       // group 1 -> LDS -> group 0 (the K loop ended on a barrier: the buffers are free)
        float* xch = reinterpret_cast<float*>(lds_all);
        if (grp == 1) {
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        xch[((tm * TN + tn) * 16 + r) * 256 + tid] = combine(acc[tm][tn][r], acl[tm][tn][r]);
        }
        __syncthreads();
        if (grp == 1) return;
    }
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
        const int pt = wn * (BN / WC) + tn * 32 + j;
        const int seg = pt / (R * WS), rem = pt % (R * WS), ohl = rem / WS, ow = rem % WS;
        const int n = n0 + seg;
        if (n >= N) continue;
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int cs = cs0 + wm * (BM / WR) + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                float* dst = Y + (((int64_t)n * Cs + cs) * HS + oh0 + ohl) * WS + ow;
                float v = combine(acc[tm][tn][r], acl[tm][tn][r]);
                if constexpr (KG > 1) v += reinterpret_cast<const float*>(lds_all)[((tm * TN + tn) * 16 + r) * 256 + tid];
                v *= unscale;
                if (gridDim.y > 1) atomicAdd(dst, v);
                else *dst = v;
            }
    }
}

// ds_read_b128 serves a wave in four groups of 16 lanes -- {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the same + 32 --
// over 64 banks: a group is conflict-free when its sixteen 16-byte reads fall into sixteen different 16-byte slots of a
// 256-byte LDS row.  The 32 columns of an MFMA B tile are therefore handed to the tile's pixels group by group: the
// first group's lanes take pixels 0-15 of the tile, the second group's pixels 16-31.
__device__ __forceinline__ int lds_group_pixel(int j) {
    return j < 4 ? j : (j < 12 ? j + 12 : (j < 16 ? j - 8 : (j < 20 ? j + 8 : (j < 28 ? j - 12 : j))));
}

// ------------------------------------------------------------------------------------------------
// UP: the four sub-pixel phases of a tile go to four waves; NC such groups (different 32-cl row tiles)
// share the block's S patch.  A phase is a 2x2 stride-1 convolution, evaluated tap by tap with
// K = 16 cs per MFMA; a wave keeps the weight fragments of its four taps in registers and refills each
// for the next K step right after using it.  LDS holds the S patch channel-innermost
// [plane][patch pixel][16 cs], double-buffered, one barrier per K step.  A pixel's 32 bytes are two 16-byte slots.  The
// sixteen pixels one lane group reads cover every slot pair of a 256-byte LDS row exactly twice: at hs = 16 with columns
// in pixel order (a group is 8 + 8 pixels of two rows), at hs = 8 / 4 once the columns are handed out group by group
// (lds_group_pixel: a group is two rows / one whole sample) -- both times the two pixels of a pair lie in rows of
// different parity; at hs = 32 (a group is 16 pixels of one row) they lie an odd multiple of eight columns apart.  The
// two halves are swapped by that bit (up_swz), so a group's reads hit sixteen different slots.  Counters before, with a
// swap every four pixels (which suits 8-lane groups on 32 banks): 43 % of the LDS cycles at hs = 16 and 68 % at hs = 4
// were bank-conflict cycles; now none.  (Pixel order is kept at hs >= 16, where the output stores dominate: with the
// group-wise order every 16-lane store chunk touched two rows and the kernel lost 6 %.)
// Tap kh feeds output-row phase PH[kh] from S row a + DH[kh]:  kh 0:(1,+1) 1:(0,0) 2:(1,0) 3:(0,-1).
// ------------------------------------------------------------------------------------------------
// PGAN_ABL_UP (debugging builds only, results are wrong): leave out parts of the UP kernel's main loop -- 1 the MFMAs, 2 the
// weight-fragment loads, 4 the staging of the next K steps (global loads, split, LDS stores), 8 the barrier per K step
// (only together with 4).
#ifndef PGAN_ABL_UP
#define PGAN_ABL_UP 0
#endif
// PGAN_UP_TIMING (debugging builds only): thread 0 of the first 64 blocks of an UP launch notes the 100 MHz wall clock at
// six points of the kernel; the debug entry point at the end of this file copies the notes out.
#ifdef PGAN_UP_TIMING
__device__ unsigned long long g_up_stamps[64][8];
#define UP_STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x < 64 && blockIdx.y == 0) g_up_stamps[blockIdx.x][i] = wall_clock64(); } while (0)
#else
#define UP_STAMP(i) do { } while (0)
#endif
// KG = 2 (NC = 1 only): two groups of four waves share the block's output tile and take the two halves of the block's cs
// range, each with its own LDS buffers; group 1 hands its sums to group 0 through LDS at the end (a fixed order:
// deterministic).  For launches whose grid gives a CU only one block: a second wave per SIMD to cover the other's waits.
template <int HS, int BN, int NC, int KG = 1>
__global__ __launch_bounds__(256 * NC * KG, KG > 1 ? 1 : ((BN <= 64 && NC == 1 && HS >= 8) ? 3 : ((BN >= 128 && NC == 1) ? 1 : 2))) void conv_up_x3_kernel(const float* __restrict__ Y, const uint16_t* __restrict__ WP,
                                                              float* __restrict__ X, int N, int Cs, int Cl,
                                                              int cl_major, const uint32_t* __restrict__ whdr, Bound by) {
    UP_STAMP(0);
    using T = PixTile<HS, BN>;
    constexpr int WS = HS, HL = 2 * HS, WL = 2 * HS;
    constexpr int SEG = T::SEG, R = T::R;
    constexpr int PR = R + 2, PC = WS + 2, NP = SEG * PR * PC;      // patch pixels
    constexpr int BM = 32 * NC, TN = BN / 32, NT = 256 * NC;
    constexpr int BPL = NP * 8;                                     // dwords per plane
    constexpr int NI = NP * 2, NQ = (NI + NT - 1) / NT;             // (pixel, channel octet) items per thread
    static_assert(KG == 1 || NC == 1, "wave groups along K only with one row tile per block");
    constexpr int BUF2 = 2 * NPL * BPL, XCH = KG > 1 ? TN * 16 * 256 : 0;
    __shared__ __attribute__((aligned(16))) uint32_t lds_all[KG * BUF2 > XCH ? KG * BUF2 : XCH];

    const int grp = KG > 1 ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 8)) : 0;      // the same in all of a wave: scalar address arithmetic
    uint32_t* const lds = lds_all + grp * BUF2;
    const int tid = KG > 1 ? (int)(threadIdx.x & 255) : (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ph = (wave >> 1) & 1, pw = wave & 1, wc = wave >> 2, j = lane & 31, h = lane >> 5;
    // the pixel of the tile that column j of the MFMA tile stands for, and the bit that swaps a pixel's two LDS slots
    const int jp = WS >= 16 ? j : lds_group_pixel(j);
    auto up_swz = [](int prow, int pcol) { return (WS >= 32 ? (pcol >> 3) : prow) & 1; };
    const int nmt = Cl / BM, npt = T::ntiles(N);
    const int b = xcd_remap(blockIdx.x, nmt * npt);
    // Each XCD gets a contiguous run of b.  cl_major: a run is one row tile (its weights stay in that XCD's L2)
    // over all pixel tiles -- for layers whose weight planes outweigh the activations; otherwise a run is a
    // range of pixel tiles over all row tiles.
    const int ptile = cl_major ? b % npt : b / nmt;
    const int cl0 = (cl_major ? b / npt : b % nmt) * BM + wc * 32;
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
        boff[q] = bval[q] ? (uint32_t)(((seg * Cs + oct * 8) * HS + a) * WS + bb) : 0u;       // from the block's first sample
        blds[q] = (e < NI) ? p * 8 + ((oct ^ up_swz(seg * PR + pr, pc)) & 1) * 4 : -1;
    }
    float breg[NQ][8];
    float sx = 1.f;        // fp16 split: the block's activations are staged times this power of two (2^-ex)
    int ex = 0;
    // Loads through buffer descriptors: a lane's byte offset is a 32-bit register fixed before the K loop, the K step
    // moves a scalar offset, the channel / plane is an immediate -- no vector address arithmetic inside the loop (with
    // 64-bit pointers it was 16 two-word adds per wave and K step beside 12 MFMAs).
    // (Y's descriptor starts at the block's first sample and ends with the tensor: reads past it return zero)
    const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(Y) + (int64_t)n0 * Cs * HS * WS, 0, (int)min((int64_t)(N - n0) * Cs * HS * WS * 4, (int64_t)0x7fffffff), 0x00020000);
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(WP), 0, (int)((int64_t)NPL * Cs * Cl * 32), 0x00020000);
    auto gload = [&](int cs0) {
        const int so = cs0 * (HS * WS * 4);
#pragma unroll
        for (int q = 0; q < NQ; ++q)
#pragma unroll
            for (int i = 0; i < 8; ++i)       // (items outside the image: channel i of pixel 0, not used)
                breg[q][i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(yrs, (int)(boff[q] * 4u) + i * HS * WS * 4, so, 0));
    };
    auto lstore = [&](uint32_t* Bs) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            uint32_t pk[NPL][4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                uint32_t p3[NPL];
                const float sq = bval[q] ? sx : 0.f;      // items outside the image hold finite values of other pixels: times zero
                split3(breg[q][2 * i] * sq, breg[q][2 * i + 1] * sq, p3);
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl) pk[pl][i] = p3[pl];
            }
            if (blds[q] >= 0) {
#pragma unroll
                for (int i = 0; i < NPL; ++i) {
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
                const int pt = tn * 32 + jp;
                const int seg = pt / (R * WS), rem = pt % (R * WS), al = rem / WS, bl = rem % WS;
                const int prow = seg * PR + al + 1 + dh, pcol = bl + 1 + dw;
                bofs[th * 2 + tw][tn] = (prow * PC + pcol) * 8 + ((h ^ up_swz(prow, pcol)) & 1) * 4;
            }
        }
    f32x16 acc[TN], acl[TN];
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[tn][r] = 0.f; acl[tn][r] = 0.f; }

    // gridDim.y > 1: the cs range is split across blocks, which then add into a zeroed X
    const int nk = Cs / (16 * (int)gridDim.y * KG), k0 = ((int)blockIdx.y * KG + grp) * nk;
    // Weight fragments are loaded PD K steps ahead of their use.  With 32-pixel tiles a K step is 12 MFMAs per wave
    // (about 400 cycles) at one or two waves per SIMD -- less than an L2 round trip under load, and the waves sat parked
    // on s_waitcnt vmcnt for half of their cycles (hs = 4: 49-59 % by the SQ counters); two register sets, used in turn,
    // double the cover for 32 more registers.  The wider tiles run 24-48 MFMAs per K step and have no registers to spare.
#ifndef PGAN_UP_PD32
#define PGAN_UP_PD32 2
#endif
#ifndef PGAN_UP_PD64
#define PGAN_UP_PD64 1
#endif
    constexpr int PD = (BN == 32) ? PGAN_UP_PD32 : (BN == 64 ? PGAN_UP_PD64 : 1);
    u32x4 a[PD][4][NPL];
    auto aload = [&](auto d, int t, int ks) {         // weight fragments of tap t, K step ks, into register set d
        constexpr int D = decltype(d)::value;
#pragma unroll
        for (int i = 0; i < NPL; ++i)
            a[D][t][i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(wrs, (int)(aofs[t] * 2u), (ks * Cl * 16 + i * (int)wplane) * 2, 0));
    };
    if constexpr (F16) {       // one pass over this block's patches for their largest magnitude -> the block's scale
        float m = 0.f;
        if (by.p) m = bound_max(by, n0, min(n0 + SEG, N), [](int n) { return n; });
        else for (int kc = 0; kc < nk; ++kc) {
            gload((k0 + kc) * 16);
#pragma unroll
            for (int q = 0; q < NQ; ++q)
#pragma unroll
                for (int i = 0; i < 8; ++i) m = fmaxf(m, bval[q] ? fabsf(breg[q][i]) : 0.f);
        }
        ex = scale_exp(block_amax(m, reinterpret_cast<float*>(lds_all)));     // one scale for the whole block
        sx = pow2f(-ex);
    }
    UP_STAMP(1);
    gload(k0 * 16);
    static_for<PD>([&](auto d) {
#pragma unroll
        for (int t = 0; t < 4; ++t) aload(d, t, k0 + min((int)decltype(d)::value, nk - 1));
    });
    lstore(lds);
    if (nk > 1) gload((k0 + 1) * 16);
    __syncthreads();
    UP_STAMP(2);
    for (int kb = 0; kb < nk; kb += PD) {
        static_for<PD>([&](auto d) {
            constexpr int D = decltype(d)::value;
            const int kc = kb + D;
            if (kc < nk) {                                    // uniform over the block
                const uint32_t* Bs = lds + (kc & 1) * NPL * BPL;
                uint32_t* Bn = lds + ((kc + 1) & 1) * NPL * BPL;
                const int knext = k0 + min(kc + PD, nk - 1);
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    u32x4 bo[TN][NPL];
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                        for (int i = 0; i < NPL; ++i) bo[tn][i] = *reinterpret_cast<const u32x4*>(Bs + i * BPL + bofs[t][tn]);
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn) {
                        if constexpr (PGAN_ABL_UP & 1) {      // (debugging builds) operands folded into the sums, no MFMA
#pragma unroll
                            for (int i = 0; i < NPL; ++i)
                                acc[tn][i] += __builtin_bit_cast(float, (a[D][t][i][0] + bo[tn][i][0]) & 0x3fffffffu);
                        } else {
                            mfma_x3(a[D][t], bo[tn], acc[tn], acl[tn]);
                        }
                    }
                    if constexpr (!(PGAN_ABL_UP & 2)) aload(d, t, knext);
                    if constexpr (!(PGAN_ABL_UP & 4)) {
                        if (t == 0 && kc + 1 < nk) lstore(Bn);                       // K step kc+1: registers -> other buffer
                        if (t == 1 && kc + 2 < nk) gload((k0 + kc + 2) * 16);        // K step kc+2: global -> registers
                    }
                }
                if constexpr (!(PGAN_ABL_UP & 8)) __syncthreads();
            }
        });
    }

    UP_STAMP(3);
    const float unscale = F16 ? pow2f(ex) * pow2f(scale_exp(__builtin_bit_cast(float, whdr[0]))) : 1.f;
    if constexpr (KG > 1) {       // group 1 -> LDS -> group 0 (the K loop ended on a barrier: the buffers are free)
        float* xch = reinterpret_cast<float*>(lds_all);
        if (grp == 1) {
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                for (int r = 0; r < 16; ++r) xch[(tn * 16 + r) * 256 + tid] = combine(acc[tn][r], acl[tn][r]);
        }
        __syncthreads();
        if (grp == 1) return;
    }
    UP_STAMP(4);
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
        const int pt = tn * 32 + jp;
        const int seg = pt / (R * WS), rem = pt % (R * WS), al = rem / WS, bl = rem % WS;
        const int n = n0 + seg;
        if (n >= N) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int cl = cl0 + (r & 3) + 8 * (r >> 2) + 4 * h;
            float* dst = X + (((int64_t)n * Cl + cl) * HL + 2 * (oh0 + al) + ph) * WL + 2 * bl + pw;
            float v = combine(acc[tn][r], acl[tn][r]);
            if constexpr (KG > 1) v += reinterpret_cast<const float*>(lds_all)[(tn * 16 + r) * 256 + tid];
            if (gridDim.y > 1) atomicAdd(dst, v * unscale);
            else *dst = v * unscale;
        }
    }
    UP_STAMP(5);
}

// ------------------------------------------------------------------------------------------------
// WGRAD: dW[cs][cl][kh][kw] = sum over (n, oh, ow) of S[n][cs][oh][ow] * L[r(n)][cl][2oh-1+kh][2ow-1+kw].
// GEMM M = cs (64 or 128 per block), N = (cl, tap) (8 cl x 16 taps per block), K = pixels, 64 per chunk, split
// over blocks into partial slabs.  Both operands are activations: both are split while they are
// staged into LDS.  Two LDS images: the MFMAs of a chunk run on one while the next chunk is split into the other
// between them, and the chunk after that is in flight from global memory (one barrier per chunk).
// A MFMA K step is 16 consecutive pixels (8 per lane half), so the L side is stored as an im2col along the
// K direction only: for HS >= 8, per channel four kw-planes P[kw][row][ow] = L[row][2ow-1+kw], whose eight
// consecutive ow are 16 contiguous bytes; for HS = 4 (eight pixels = two rows) a full im2col [tap][pixel].
// Plane strides (HS >= 8) and a swap of the two 16-byte pixel halves for kh >= 2 (HS = 4) spread the 16
// (kh, kw) lanes of a channel evenly over the banks.
// ------------------------------------------------------------------------------------------------
template <int HS, int BM_>
struct WgGeom {
    static constexpr int PIXC = 64, BM = BM_, CLB = 8;
    using T = PixTile<HS, PIXC>;
    static constexpr int SEG = T::SEG, R = T::R, WS = HS, HL = 2 * HS, WL = 2 * HS;
    static constexpr int ARS = PIXC * 2 + 16;                        // bytes per A row
    static constexpr int APL = BM * ARS;                             // bytes per A plane
    // HS >= 8: kw planes
    static constexpr int PRR = 2 * R + 2;                            // raw rows per chunk
    static constexpr int SR = WS * 2;                                // bytes per plane row (WS bf16)
    // kw-plane and channel strides chosen (by enumeration over the lane groups of a 16-byte LDS read) so that the 16
    // (kh, kw) lanes x 2 channels of a fragment read fall on distinct banks
    static constexpr int KWB = PRR * SR + (HS == 8 ? 32 : (HS == 32 ? 16 : 0));
    static constexpr int CHB = (HS == 4) ? 16 * 32 : 4 * KWB + (HS == 16 ? 16 : (HS == 32 ? 192 : 0));        // bytes per (segment,) channel
    static constexpr int BPL = (HS == 4) ? SEG * CLB * CHB : CLB * CHB;
    static constexpr int LDS_BYTES = NPL * APL + NPL * BPL;
    static constexpr bool PIPE = F16;           // two LDS images and a software-pipelined K loop (two planes only)
    static constexpr int IMAGES = PIPE ? 2 : 1;
    static_assert(IMAGES * LDS_BYTES <= 160 * 1024, "LDS of one CU");
    // staging items of the L side
    static constexpr int NIB = (HS == 4) ? SEG * CLB * 4 * 4 : CLB * PRR * (WS / 2);
    static constexpr int NQB = (NIB + 255) / 256;
};

// Everything a thread needs per chunk beyond two block-uniform base pointers is fixed before the K loop: byte offsets of
// its global loads (32-bit, added to the uniform base by the load itself), LDS addresses of its stores, and which of its
// patch rows fall outside the image at the top / bottom chunk of a sample.
template <int HS, int BM>
__global__ __launch_bounds__(256) void conv_wgrad_x3_kernel(const float* __restrict__ YA, const float* __restrict__ X,
                                                            float* __restrict__ slab, int N, int Cs, int Cl, int n_keep,
                                                            int n_swap, int chunks_per_block, Bound bs, Bound bl) {
    using G = WgGeom<HS, BM>;
    using T = typename G::T;
    constexpr int WS = HS, HL = 2 * HS, WL = 2 * HS, SEG = G::SEG, R = G::R, PIXC = G::PIXC, CLB = G::CLB;
    constexpr int AV = BM * PIXC / 4 / 256;          // f32x4 of the S tile per thread
    constexpr int TMW = BM / 64;                     // 32-row tiles per wave
    constexpr int NQB = G::NQB;
    __shared__ __attribute__((aligned(16))) unsigned char lds[G::IMAGES * G::LDS_BYTES];      // image(s) of a chunk

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, j = lane & 31, h = lane >> 5;
    // Blocks that work on the same pixels (one K range, all the output tiles) read the same S rows and L patches: an
    // XCD gets a contiguous run of the (K range, cl tile, cs tile) list, K range slowest, so those re-reads stay in its
    // L2.  Where one K range's tiles span several runs, each run takes all the cs tiles of some cl tiles: it re-reads
    // S, which is half the size of L.
    const int nmt = Cs / BM, ntile = nmt * (Cl / CLB);
    const int b = xcd_remap(blockIdx.x, gridDim.x), ky = b / ntile, tile = b % ntile;
    const int cs0 = (tile % nmt) * BM, cl0 = (tile / nmt) * CLB;
    const int total_chunks = T::ntiles(N);
    const int c_begin = ky * chunks_per_block;
    const int c_end = min(total_chunks, c_begin + chunks_per_block);
    auto remap = [&](int n) { return n + (n < n_keep ? 0 : (n < n_keep + n_swap ? n_swap : -n_swap)); };

    // S tile: item q = (row, 4 pixels)
    uint32_t a_off[AV];
    int a_seg[AV], a_lds[AV];
#pragma unroll
    for (int q = 0; q < AV; ++q) {
        const int idx = tid + 256 * q, row = idx / (PIXC / 4), p = (idx % (PIXC / 4)) * 4;
        a_seg[q] = p / (R * WS);
        a_off[q] = 4u * (uint32_t)((a_seg[q] * Cs + cs0 + row) * (HS * HS) + p % (R * WS));
        a_lds[q] = row * G::ARS + p * 2;
    }
    f32x4 areg[AV];
    float sa = 1.f, sb = 1.f;      // fp16 split: scales of the block's S rows and L patches (2^-ea, 2^-eb)
    int ea = 0, eb = 0;
    // L side items.  HS = 4: (segment, channel, kh, oh) -> one input row of 8; else (channel, patch row, 4 columns) -> 6 values
    constexpr int NBR = (HS == 4) ? 8 : 6;            // raw floats per item
    uint32_t b_off[NQB];
    int b_lds[NQB], b_flag[NQB];      // flags: 1 item exists, 2 first / 4 last patch row, 8 has a left / 16 a right neighbour, 32 (HS = 4) row inside the image
    const uint32_t xsample = 4u * (uint32_t)(Cl * HL * WL);       // bytes per sample of L
#pragma unroll
    for (int q = 0; q < NQB; ++q) {
        const int e = tid + 256 * q;
        if constexpr (HS == 4) {
            const int seg = e / (CLB * 16), ch = (e / 16) % CLB, kh = (e / 4) % 4, oh = e % 4;
            const int ih = 2 * oh - 1 + kh;
            b_flag[q] = (e < G::NIB ? 1 : 0) | ((ih >= 0 && ih < HL) ? 32 : 0);
            b_off[q] = 4u * (uint32_t)(((cl0 + ch) * HL + ((b_flag[q] & 32) ? ih : 0)) * WL);
            b_lds[q] = ((seg * CLB + ch) * 16 + kh * 4) * 32 + (((oh >> 1) ^ (kh >> 1)) * 2 + (oh & 1)) * 8;
        } else {
            constexpr int ND = WS / 2;
            const int ch = e / (G::PRR * ND), r = (e / ND) % G::PRR, d = e % ND;
            b_flag[q] = (e < G::NIB ? 1 : 0) | (r == 0 ? 2 : 0) | (r == G::PRR - 1 ? 4 : 0) | (d > 0 ? 8 : 0) |
                        (4 * d + 4 < WL ? 16 : 0);
            b_off[q] = 4u * (uint32_t)(((cl0 + ch) * HL + r) * WL + 4 * d);       // from row 2 oh0 - 1 of the sample
            b_lds[q] = ch * G::CHB + r * G::SR + d * 4;
        }
    }
    float breg[NQB][NBR];
    // Block-uniform part of a chunk's addresses: a buffer descriptor per operand that starts at the chunk's first S row /
    // L patch row and ends with the tensor.  A lane's byte offset into it is fixed before the K loop; a lane whose item
    // does not exist in this chunk (sample past the batch, patch row above or below the image, no left / right neighbour)
    // reads at an offset past every tensor, which returns zero without a branch -- the chunk body is one basic block.
    struct Src {
        __amdgpu_buffer_rsrc_t ya, xb;
        int n0, out_mask;
    };
    // (2^31 as an unsigned byte offset: past every descriptor here, whose sizes are capped at 2^31 - 1 -- also when more
    // than that remains of a tensor behind the chunk, where a smaller offset would read real values)
    constexpr int NOWHERE = (int)0x80000000u;
    const int64_t ya_bytes = (int64_t)N * Cs * (HS * HS) * 4, xb_bytes = (int64_t)N * xsample;
    auto src_of = [&](int chunk) {
        int n0, oh0;
        T::origin(chunk, n0, oh0);
        Src r;
        const int64_t ao = 4 * ((int64_t)n0 * Cs * (HS * HS) + oh0 * WS);
        // HS >= 8: rows 2 oh0 - 1 .. of sample r(n0); the base may point one row before the sample, which is never read
        const int64_t bo = HS == 4 ? (int64_t)0 : (int64_t)remap(n0) * xsample + 4 * (2 * oh0 - 1) * WL;
        r.ya = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(YA)) + ao, 0,
                                                 (int)min(ya_bytes - ao, (int64_t)0x7fffffff), 0x00020000);
        r.xb = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(X)) + bo, 0,
                                                 (int)min(xb_bytes - bo, (int64_t)0x7fffffff), 0x00020000);
        r.n0 = n0;
        r.out_mask = (oh0 == 0 ? 2 : 0) | (oh0 + R == HS ? 4 : 0);      // patch rows outside the image
        return r;
    };
    auto ld4 = [](const __amdgpu_buffer_rsrc_t& rs, int off) {
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0));
    };
    auto ld1 = [](const __amdgpu_buffer_rsrc_t& rs, int off) {
        return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, off, 0, 0));
    };
    constexpr int NST = AV + NQB;         // staging items per thread and chunk: S first, then L
    auto load_item = [&](auto itc, const Src& sc) {
        constexpr int it = decltype(itc)::value;
        if constexpr (it < AV) {
            constexpr int q = it;
            areg[q] = ld4(sc.ya, (int)a_off[q]);        // a sample past the batch lies past the end of S
        } else if constexpr (HS == 4) {
            constexpr int q = it - AV;
            const int n = sc.n0 + (tid + 256 * q) / (CLB * 16);
            const int off = ((b_flag[q] & 33) == 33 && n < N) ? (int)((uint32_t)remap(n) * xsample + b_off[q]) : NOWHERE;
            const f32x4 v0 = ld4(sc.xb, off), v1 = ld4(sc.xb, off + 16);
#pragma unroll
            for (int i = 0; i < 4; ++i) { breg[q][i] = v0[i]; breg[q][4 + i] = v1[i]; }
        } else {
            constexpr int q = it - AV;
            const bool row = (b_flag[q] & 1) && !(b_flag[q] & sc.out_mask);
            const f32x4 v = ld4(sc.xb, row ? (int)b_off[q] : NOWHERE);
            breg[q][0] = ld1(sc.xb, (row && (b_flag[q] & 8)) ? (int)b_off[q] - 4 : NOWHERE);
#pragma unroll
            for (int i = 0; i < 4; ++i) breg[q][1 + i] = v[i];
            breg[q][5] = ld1(sc.xb, (row && (b_flag[q] & 16)) ? (int)b_off[q] + 16 : NOWHERE);
        }
    };
    // split item `it` of the registers into the LDS image at `img` (A planes, then B planes)
    auto stage_item = [&](auto itc, unsigned char* img) {
        constexpr int it = decltype(itc)::value;
        if constexpr (it < AV) {
            constexpr int q = it;
            uint32_t pa[NPL], pb[NPL];
            split3(areg[q][0] * sa, areg[q][1] * sa, pa);
            split3(areg[q][2] * sa, areg[q][3] * sa, pb);
#pragma unroll
            for (int i = 0; i < NPL; ++i) {
                u32x2 v = {pa[i], pb[i]};
                *reinterpret_cast<u32x2*>(img + i * G::APL + a_lds[q]) = v;
            }
        } else {
            constexpr int q = it - AV;
            if (!(b_flag[q] & 1)) return;
            unsigned char* base = img + NPL * G::APL + b_lds[q];
            if constexpr (HS == 4) {
                uint32_t s01[NPL], s23[NPL], s45[NPL], s67[NPL];
                split3(breg[q][0] * sb, breg[q][1] * sb, s01);
                split3(breg[q][2] * sb, breg[q][3] * sb, s23);
                split3(breg[q][4] * sb, breg[q][5] * sb, s45);
                split3(breg[q][6] * sb, breg[q][7] * sb, s67);
#pragma unroll
                for (int i = 0; i < NPL; ++i) {
                    // values at input columns 2ow-1+kw for ow = 0..3: kw 0: (-,1,3,5)  1: (0,2,4,6)  2: (1,3,5,7)  3: (2,4,6,-)
                    const u32x2 k0 = {s01[i] & 0xffff0000u, (s23[i] >> 16) | (s45[i] & 0xffff0000u)};
                    const u32x2 k1 = {(s01[i] & 0xffffu) | (s23[i] << 16), (s45[i] & 0xffffu) | (s67[i] << 16)};
                    const u32x2 k2 = {(s01[i] >> 16) | (s23[i] & 0xffff0000u), (s45[i] >> 16) | (s67[i] & 0xffff0000u)};
                    const u32x2 k3 = {(s23[i] & 0xffffu) | (s45[i] << 16), s67[i] & 0xffffu};
                    unsigned char* b = base + i * G::BPL;
                    *reinterpret_cast<u32x2*>(b) = k0;
                    *reinterpret_cast<u32x2*>(b + 32) = k1;
                    *reinterpret_cast<u32x2*>(b + 64) = k2;
                    *reinterpret_cast<u32x2*>(b + 96) = k3;
                }
            } else {
                uint32_t s01[NPL], s23[NPL], s45[NPL];
                split3(breg[q][0] * sb, breg[q][1] * sb, s01);
                split3(breg[q][2] * sb, breg[q][3] * sb, s23);
                split3(breg[q][4] * sb, breg[q][5] * sb, s45);
#pragma unroll
                for (int i = 0; i < NPL; ++i) {
                    // plane kw, positions (2d, 2d+1) = input columns (4d-1+kw, 4d+1+kw)
                    unsigned char* b = base + i * G::BPL;
                    *reinterpret_cast<uint32_t*>(b) = (s01[i] & 0xffffu) | (s23[i] << 16);
                    *reinterpret_cast<uint32_t*>(b + G::KWB) = (s01[i] >> 16) | (s23[i] & 0xffff0000u);
                    *reinterpret_cast<uint32_t*>(b + 2 * G::KWB) = (s23[i] & 0xffffu) | (s45[i] << 16);
                    *reinterpret_cast<uint32_t*>(b + 3 * G::KWB) = (s23[i] >> 16) | (s45[i] & 0xffff0000u);
                }
            }
        }
    };
    auto gload = [&](int chunk) {
        const Src sc = src_of(chunk);
        static_for<NST>([&](auto ic) { load_item(ic, sc); });
    };

    // fragment addresses (bytes): A rows of this lane; B column (cl, kh, kw) of this lane per pixel tile
    int abase[TMW];
#pragma unroll
    for (int tm = 0; tm < TMW; ++tm) abase[tm] = (wm * (BM / 2) + tm * 32 + j) * G::ARS + h * 16;
    int bbase[2];
#pragma unroll
    for (int tn = 0; tn < 2; ++tn) {
        const int cl_l = wn * 4 + tn * 2 + (j >> 4), kh = (j >> 2) & 3, kw = j & 3;
        if constexpr (HS >= 16) bbase[tn] = cl_l * G::CHB + kw * G::KWB + kh * G::SR + h * 16;
        else if constexpr (HS == 8) bbase[tn] = cl_l * G::CHB + kw * G::KWB + (2 * h + kh) * G::SR;
        else bbase[tn] = (cl_l * 16 + (j & 15)) * 32 + (h ^ (kh >> 1)) * 16;
    }
    // byte offset of the B fragment of K step s (16 consecutive pixels: half a row at HS = 32, one row at 16, two at 8,
    // a whole sample at 4)
    auto bstep = [](int s) constexpr {
        return (HS == 32) ? (s >> 1) * 2 * G::SR + (s & 1) * 32
                          : s * ((HS == 16) ? 2 * G::SR : (HS == 8 ? 4 * G::SR : CLB * G::CHB));
    };
    f32x16 acc[TMW][2], acl[TMW][2];
#pragma unroll
    for (int tm = 0; tm < TMW; ++tm)
#pragma unroll
        for (int tn = 0; tn < 2; ++tn)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[tm][tn][r] = 0.f; acl[tm][tn][r] = 0.f; }

    if constexpr (F16) {       // one pass over this block's chunks for the largest magnitude of each operand
        float ma = 0.f, mb = 0.f;
        int n_lo = 0, n_hi = 0, oh_;
        if (c_begin < c_end) {      // samples of this block's chunks
            T::origin(c_begin, n_lo, oh_);
            T::origin(c_end - 1, n_hi, oh_);
            n_hi = min(n_hi + SEG, N);
        }
        if (bs.p) ma = bound_max(bs, n_lo, n_hi, [](int n) { return n; });
        if (bl.p) mb = bound_max(bl, n_lo, n_hi, remap);
        if (!bs.p || !bl.p) for (int ch = c_begin; ch < c_end; ++ch) {
            gload(ch);
#pragma unroll
            for (int q = 0; q < AV; ++q)
#pragma unroll
                for (int i = 0; i < 4; ++i) ma = fmaxf(ma, fabsf(areg[q][i]));
#pragma unroll
            for (int q = 0; q < NQB; ++q)
#pragma unroll
                for (int i = 0; i < NBR; ++i) mb = fmaxf(mb, fabsf(breg[q][i]));
        }
        ea = scale_exp(block_amax(ma, reinterpret_cast<float*>(lds)));
        eb = scale_exp(block_amax(mb, reinterpret_cast<float*>(lds)));
        sa = pow2f(-ea);
        sb = pow2f(-eb);
    }
    // K loop, one barrier per chunk: the MFMAs of chunk c read LDS image c & 1 while the registers (chunk c + 1) are split
    // into the other image item by item between them, and every item's registers are refilled from chunk c + 2 as soon as
    // they are free.  Chunks past the block's range repeat its last one; what they stage is never read.
    constexpr int KS = PIXC / 16;
    auto chunk_of = [&](int c) { return min(c, c_end - 1); };
    auto frags = [&](auto setc, auto sc, const unsigned char* img, u32x4 (&a)[2][TMW][NPL], u32x4 (&bo)[2][2][NPL]) {
        constexpr int set = decltype(setc)::value, s = decltype(sc)::value;
#pragma unroll
        for (int tm = 0; tm < TMW; ++tm)
#pragma unroll
            for (int i = 0; i < NPL; ++i) a[set][tm][i] = *reinterpret_cast<const u32x4*>(img + i * G::APL + abase[tm] + s * 32);
#pragma unroll
        for (int tn = 0; tn < 2; ++tn)
#pragma unroll
            for (int i = 0; i < NPL; ++i)
                bo[set][tn][i] = *reinterpret_cast<const u32x4*>(img + NPL * G::APL + i * G::BPL + bbase[tn] + bstep(s));
    };
    auto chunk = [&](const unsigned char* cur, unsigned char* nxt, const Src& refill) {
        u32x4 a[2][TMW][NPL], bo[2][2][NPL];
        frags(IC<0>{}, IC<0>{}, cur, a, bo);
        static_for<KS>([&](auto sc) {
            constexpr int s = decltype(sc)::value;
            if constexpr (s + 1 < KS) frags(IC<(s + 1) & 1>{}, IC<s + 1>{}, cur, a, bo);
#pragma unroll
            for (int tm = 0; tm < TMW; ++tm)
#pragma unroll
                for (int tn = 0; tn < 2; ++tn) mfma_x3(a[s & 1][tm], bo[s & 1][tn], acc[tm][tn], acl[tm][tn]);
            constexpr int lo = s * NST / KS, hi = (s + 1) * NST / KS;
            static_for<hi - lo>([&](auto ic) {
                stage_item(IC<lo + decltype(ic)::value>{}, nxt);
                load_item(IC<lo + decltype(ic)::value>{}, refill);
            });
        });
    };
    if constexpr (G::PIPE) {
        if (c_begin < c_end) {
            gload(c_begin);
            static_for<NST>([&](auto ic) { stage_item(ic, lds); });
            gload(chunk_of(c_begin + 1));
            __syncthreads();
            for (int ch = c_begin; ch < c_end; ch += 2) {
                chunk(lds, lds + G::LDS_BYTES, src_of(chunk_of(ch + 2)));
                __syncthreads();
                if (ch + 1 < c_end) {
                    chunk(lds + G::LDS_BYTES, lds, src_of(chunk_of(ch + 3)));
                    __syncthreads();
                }
            }
        }
    } else {
        // three planes: two images would leave one block per CU; one image, staged and consumed in turn, and the two or three
        // co-resident blocks overlap each other's phases instead
        if (c_begin < c_end) gload(c_begin);
        for (int ch = c_begin; ch < c_end; ++ch) {
            static_for<NST>([&](auto ic) { stage_item(ic, lds); });
            __syncthreads();
            if (ch + 1 < c_end) gload(ch + 1);
            u32x4 a[2][TMW][NPL], bo[2][2][NPL];
            static_for<KS>([&](auto sc) {
                constexpr int s = decltype(sc)::value;
                frags(IC<0>{}, sc, lds, a, bo);
#pragma unroll
                for (int tm = 0; tm < TMW; ++tm)
#pragma unroll
                    for (int tn = 0; tn < 2; ++tn) mfma_x3(a[0][tm], bo[0][tn], acc[tm][tn], acl[tm][tn]);
            });
            __syncthreads();
        }
    }

    const float unscale = F16 ? pow2f(ea) * pow2f(eb) : 1.f;
    float* out = slab + (int64_t)ky * Cs * Cl * 16;
#pragma unroll
    for (int tm = 0; tm < TMW; ++tm)
#pragma unroll
        for (int tn = 0; tn < 2; ++tn)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int cs = cs0 + wm * (BM / 2) + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                const int cl = cl0 + wn * 4 + tn * 2 + (j >> 4);
                out[((int64_t)cs * Cl + cl) * 16 + (j & 15)] = combine(acc[tm][tn][r], acl[tm][tn][r]) * unscale;
            }
}

__global__ void slab_sum_kernel(const float* __restrict__ slab, float* __restrict__ out, int nslab, int64_t len,
                                int accumulate) {
    const int64_t i4 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i4 * 4 >= len) return;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    for (int z = 0; z < nslab; ++z) s += *reinterpret_cast<const f32x4*>(slab + z * len + i4 * 4);
    if (accumulate) s += *reinterpret_cast<const f32x4*>(out + i4 * 4);
    *reinterpret_cast<f32x4*>(out + i4 * 4) = s;
}
// The same for many slabs of a small gradient (hs = 16: 32 slabs of 512 KB): 64 columns x 4 slab lanes per block, two
// loads in flight per lane, lanes combined through LDS in a fixed order -- four times the blocks, an eighth of the
// dependent additions per thread.
__global__ __launch_bounds__(256) void slab_sum4_kernel(const float* __restrict__ slab, float* __restrict__ out, int nslab,
                                                        int64_t len, int accumulate) {
    __shared__ f32x4 sh[4][64];
    const int cx = threadIdx.x & 63, zy = threadIdx.x >> 6;
    const int64_t i4 = (int64_t)blockIdx.x * 64 + cx;
    const bool in = i4 * 4 < len;
    f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0;
    if (in) {
        int z = zy;
        for (; z + 4 < nslab; z += 8) {
            s0 += *reinterpret_cast<const f32x4*>(slab + z * len + i4 * 4);
            s1 += *reinterpret_cast<const f32x4*>(slab + (z + 4) * len + i4 * 4);
        }
        if (z < nslab) s0 += *reinterpret_cast<const f32x4*>(slab + z * len + i4 * 4);
    }
    sh[zy][cx] = s0 + s1;
    __syncthreads();
    if (zy == 0 && in) {
        f32x4 s = (sh[0][cx] + sh[1][cx]) + (sh[2][cx] + sh[3][cx]);
        if (accumulate) s += *reinterpret_cast<const f32x4*>(out + i4 * 4);
        *reinterpret_cast<f32x4*>(out + i4 * 4) = s;
    }
}
void launch_slab_sum(const float* ws, float* dW, int ns, int64_t len, int accumulate, hipStream_t st) {
    if (ns >= 8) slab_sum4_kernel<<<(int)((len / 4 + 63) / 64), 256, 0, st>>>(ws, dW, ns, len, accumulate);
    else slab_sum_kernel<<<(int)((len / 4 + 255) / 256), 256, 0, st>>>(ws, dW, ns, len, accumulate);
}

int g_wg_target = 0;     // tuning override "wg_target": blocks aimed at (0 = by size)
int g_wg_bm = 0;         // tuning override "wg_bm": rows of S per block, 64 or 128 (0 = by size)

int wgrad_bm(int hs, int N, int Cs, int Cl) {
    if (!F16) return 64;        // three planes of a 128-row tile do not fit the LDS twice
    if (g_wg_bm) return (g_wg_bm == 128 && Cs % 128 == 0) ? 128 : 64;
    // hs = 4: the two LDS images of a 64-row tile leave room for one block per CU only; a 128-row block does twice the
    // work with the same L staging (measured 75 against 101 us at N = 256, 31.8 against 34.9 at N = 64)
    return (hs == 4 && Cs % 128 == 0) ? 128 : 64;
}

template <int HS>
int wgrad_x3_splits(int N, int Cs, int Cl) {
    const int tiles = (Cs / wgrad_bm(HS, N, Cs, Cl)) * (Cl / 8);
    const int chunks = PixTile<HS, 64>::ntiles(N);
    // blocks aimed at: as many as are resident at once -- two per CU, one where the two LDS images take over half a CU's LDS
    const int target = g_wg_target ? g_wg_target : ((F16 && (HS == 4 || HS == 32)) ? 256 : 512);
    int want = (target + tiles - 1) / tiles;
    if (want > chunks) want = chunks;
    if (want < 1) want = 1;
    const int cpb = (chunks + want - 1) / want;
    return (chunks + cpb - 1) / cpb;
}

template <int HS>
int launch_wgrad(const float* YA, const float* X, float* dW, float* ws, int64_t ws_floats, int N, int Cs, int Cl,
                 int n_keep, int n_swap, int accumulate, hipStream_t st, Bound bs, Bound bl) {
    if (Cs % 64 || Cl % 8) return PGAN_EINVAL;
    // the kernels address one sample's rows with 32-bit byte offsets (at hs = 4: the whole of L)
    if ((int64_t)Cs * HS * HS * 64 >= (1ll << 31) || (int64_t)(HS == 4 ? N : 1) * Cl * HS * HS * 16 >= (1ll << 31)) return PGAN_EINVAL;
    const int chunks = PixTile<HS, 64>::ntiles(N);
    const int ns = wgrad_x3_splits<HS>(N, Cs, Cl);
    const int cpb = (chunks + ns - 1) / ns;
    const int64_t len = (int64_t)Cs * Cl * 16;
    if (ns * len > ws_floats) return PGAN_EINVAL;
    if constexpr (F16) if (wgrad_bm(HS, N, Cs, Cl) == 128) {
        conv_wgrad_x3_kernel<HS, 128><<<(Cs / 128) * (Cl / 8) * ns, 256, 0, st>>>(YA, X, ws, N, Cs, Cl, n_keep, n_swap, cpb, bs, bl);
        launch_slab_sum(ws, dW, ns, len, accumulate, st);
        return pgan_launch_status();
    }
    conv_wgrad_x3_kernel<HS, 64><<<(Cs / 64) * (Cl / 8) * ns, 256, 0, st>>>(YA, X, ws, N, Cs, Cl, n_keep, n_swap, cpb, bs, bl);
    launch_slab_sum(ws, dW, ns, len, accumulate, st);
    return pgan_launch_status();
}

int g_dn_order = 0;      // tuning override "down_order": 1 pixel-tile-major, 2 row-tile-major block order (0 = by size)
// ks: 1 | 2 (blocks along K, two-addend sums) | 3 (one block of two wave groups: no sums through memory) | 4 (two blocks of
// two wave groups each)
// zeroed: the caller states that Y holds zeros already (K-split launches add into it); -1 only asks whether this launch
// would need them (returns 1 / 0, launches nothing)
template <int HS, int BM, int BN, int CLC>
int launch_down_tile(const float* X, const uint16_t* WP, float* Y, int N, int Cs, int Cl, int ks, hipStream_t st,
                     const uint32_t* whdr, Bound bx, int zeroed) {
    if (ks < 1 || ks > 4 || Cs % BM || Cl % (2 * CLC * (ks == 3 ? 2 : ks))) return PGAN_EINVAL;
    const int gy = (ks == 2 || ks == 4) ? 2 : 1;
    if (zeroed < 0) return gy > 1;
    if (gy > 1 && !zeroed && hipMemsetAsync(Y, 0, sizeof(float) * (size_t)N * Cs * HS * HS, st) != hipSuccess) return PGAN_ELAUNCH;
    const dim3 g((Cs / BM) * PixTile<HS, BN>::ntiles(N), gy);
    const int csm = g_dn_order ? g_dn_order - 1 : ((int64_t)Cs * Cl * 16 * NPL * 2 > (int64_t)N * Cl * HS * HS * 16);
    if (ks >= 3) conv_down_x3_kernel<HS, BM, BN, CLC, 2><<<g, 512, 0, st>>>(X, WP, Y, N, Cs, Cl, whdr, bx, csm);
    else conv_down_x3_kernel<HS, BM, BN, CLC, 1><<<g, 256, 0, st>>>(X, WP, Y, N, Cs, Cl, whdr, bx, csm);
    return pgan_launch_status();
}

int g_dn_variant = 0, g_dn_ks = 0, g_up_variant = 0, g_up_ks = 0;   // tuning overrides (0 = by size)
int g_up_order = 0;                                                 // 1: pixel-major, 2: row-tile-major block order
int g_up_kg = 0;                                                    // 1: never two wave groups along K in UP

template <int HS>
int launch_down(const float* X, const uint16_t* WP, float* Y, int N, int Cs, int Cl, hipStream_t st, const uint32_t* whdr,
                Bound bx, int zeroed = 0) {
    // the kernel addresses a block's samples of X and the weight planes with 32-bit byte offsets
    if ((int64_t)N * Cl * HS * HS * 4 >= (1ll << 31) || (int64_t)Cs * Cl * 32 * NPL * 2 >= (1ll << 31) || (int64_t)Cl * HS * HS * 16 * 64 >= (1ll << 31)) return PGAN_EINVAL;
    int variant = g_dn_variant, ks = g_dn_ks;
    if (variant == 0 || ks < 1) {
        // 128x64 tiles when they still give two blocks per CU, else 64x64 tiles (three blocks per CU);
        // a grid that leaves CUs idle also splits K over 2 or 4 blocks
        const int b4 = (Cs / 64) * PixTile<HS, 64>::ntiles(N);
        int v = 4, k = b4 >= 512 ? 1 : (b4 * 2 >= 512 ? 2 : 4);
        if (Cs % 128 == 0 && b4 >= 1024) { v = 2; k = 1; }
        // (ks = 3 -- one block of two wave groups per tile, no sums through memory -- is faster in isolation where a grid has
        // at most one block per CU and K is short (N = 128, hs = 8: 30.6 against 35.4 us; N = 64, hs = 8: 21.5 against 22.9),
        // and costs the iteration 5 % when chosen there: eight waves with the registers of a 128x64 tile leave the other
        // streams' blocks no room on the CU.  Not chosen by size; kept as a tuning option.)
        else if (Cs % 128 == 0 && b4 >= 512 && Cl <= 128) { v = 2; k = 1; }      // short K: one 128x64 block per CU (Cl = 64: 18.5 against 21.2 us; Cl = 128: 35.0 against 38.7)
        else if (Cs % 128 == 0 && b4 >= 512 && Cl >= 256) { v = 2; k = 2; }      // long K: 128x64 tiles, K over two blocks
        else if (Cs % 128 == 0 && b4 >= 256 && b4 < 512 && Cl >= 128) { v = 2; k = 4; }   // ... over two blocks x two wave groups
        while (k > 1 && k != 3 && Cl % (8 * k)) k >>= 1;
        if (variant == 0) variant = v;
        if (ks < 1) ks = k;
    }
    if (ks > 4) return PGAN_EINVAL;
    switch (variant) {
        case 1: return launch_down_tile<HS, 128, 128, 4>(X, WP, Y, N, Cs, Cl, ks, st, whdr, bx, zeroed);
        case 2: return launch_down_tile<HS, 128, 64, 4>(X, WP, Y, N, Cs, Cl, ks, st, whdr, bx, zeroed);
        case 3: return launch_down_tile<HS, 64, 128, 4>(X, WP, Y, N, Cs, Cl, ks, st, whdr, bx, zeroed);
        case 4: return launch_down_tile<HS, 64, 64, 4>(X, WP, Y, N, Cs, Cl, ks, st, whdr, bx, zeroed);
        default: return PGAN_EINVAL;
    }
}

template <int HS>
int launch_up(const float* Y, const uint16_t* WP, float* X, int N, int Cs, int Cl, hipStream_t st, const uint32_t* whdr,
              Bound by) {
    if (Cl % 32 || Cs % 16) return PGAN_EINVAL;
    // the kernel addresses a block's samples of Y and the weight planes with 32-bit byte offsets
    if ((int64_t)N * Cs * HS * HS >= (1ll << 31) || (int64_t)Cs * Cl * 32 * NPL * 2 >= (1ll << 31) || (int64_t)Cs * HS * HS * 64 >= (1ll << 31)) return PGAN_EINVAL;
    const uint16_t* up = WP + (size_t)NPL * Cs * Cl * 16;
    // 64-pixel tiles; 32-pixel tiles when those would give fewer than two blocks per CU, and always at hs = 4 (measured
    // 135 against 149 us at N = 320; the 64-pixel tile spills there unless its occupancy drops to 2 waves per SIMD)
    const int b64 = (Cl / 32) * PixTile<HS, 64>::ntiles(N), b32 = (Cl / 32) * PixTile<HS, 32>::ntiles(N);
    // (fp16 split: 64-pixel tiles from 1.5 blocks per CU on; at hs = 4 from two on -- 82.5 against 89.6 us at N = 320)
    int variant = g_up_variant ? g_up_variant : (F16 ? ((HS == 4 && b64 < 512) ? 5 : 1) : ((HS == 4 || b64 < 512) ? 5 : 1));
    // grids of at most one block per CU: two wave groups along K per block, on the widest tile that still gives such a
    // grid (fp16 split, isolated: hs = 4, 64 samples 26.3 -> 23.0 us, 128 samples 40.1 -> 37.0; hs = 8, 64 samples 22.2 -> 19.3)
    if (F16 && !g_up_variant && !g_up_kg && Cs % 32 == 0) variant = b32 <= 256 ? 6 : (b64 <= 256 ? 7 : variant);
    if ((variant == 3 || variant == 4) && Cl % 64) variant -= 2;
    if (variant == 6 && Cs % 32) variant = 5;
    if (variant == 7 && Cs % 32) variant = 1;
    const int nc = (variant == 3 || variant == 4) ? 2 : 1;
    const int bn = (variant == 5 || variant == 6) ? 32 : ((variant & 1) ? 64 : 128);
    const int blocks = (Cl / (32 * nc)) * (bn == 128 ? PixTile<HS, 128>::ntiles(N)
                                                     : (bn == 64 ? PixTile<HS, 64>::ntiles(N) : PixTile<HS, 32>::ntiles(N)));
    int ks = g_up_ks ? g_up_ks : 1;
    while (ks > 1 && Cs % (32 * ks * (variant >= 6 ? 2 : 1))) ks >>= 1;
    if (ks > 1 && hipMemsetAsync(X, 0, sizeof(float) * (size_t)N * Cl * HS * HS * 4, st) != hipSuccess) return PGAN_ELAUNCH;
    const dim3 g(blocks, ks);
    const int clm = g_up_order ? g_up_order - 1 : ((int64_t)Cs * Cl * 96 > (int64_t)N * Cs * HS * HS * 4);
    switch (variant) {
        case 1: conv_up_x3_kernel<HS, 64, 1><<<g, 256, 0, st>>>(Y, up, X, N, Cs, Cl, clm, whdr, by); break;
        case 2: conv_up_x3_kernel<HS, 128, 1><<<g, 256, 0, st>>>(Y, up, X, N, Cs, Cl, clm, whdr, by); break;
        case 3: conv_up_x3_kernel<HS, 64, 2><<<g, 512, 0, st>>>(Y, up, X, N, Cs, Cl, clm, whdr, by); break;
        case 4: conv_up_x3_kernel<HS, 128, 2><<<g, 512, 0, st>>>(Y, up, X, N, Cs, Cl, clm, whdr, by); break;
        case 5: conv_up_x3_kernel<HS, 32, 1><<<g, 256, 0, st>>>(Y, up, X, N, Cs, Cl, clm, whdr, by); break;
        case 6: conv_up_x3_kernel<HS, 32, 1, 2><<<g, 512, 0, st>>>(Y, up, X, N, Cs, Cl, clm, whdr, by); break;
        case 7: conv_up_x3_kernel<HS, 64, 1, 2><<<g, 512, 0, st>>>(Y, up, X, N, Cs, Cl, clm, whdr, by); break;
        default: return PGAN_EINVAL;
    }
    return pgan_launch_status();
}

// a bound is absent, per-sample (rows of 1..64 entries) or an array of at least one entry
bool bound_ok(const Bound& b) {
    return !b.p || (b.per_sample > 0 && b.per_sample <= 64) || (b.per_sample == 0 && b.count > 0);
}

}  // namespace

extern "C" {

int FN_SET_OPTION(const char* name, int value) {
    if (!name) return PGAN_EINVAL;
    auto eq = [](const char* x, const char* y) { while (*x && *x == *y) { ++x; ++y; } return *x == *y; };
    if (eq(name, "down_variant") && value >= 0 && value <= 4) { g_dn_variant = value; return PGAN_OK; }
    if (eq(name, "down_ks") && value >= 0 && value <= 8) { g_dn_ks = value; return PGAN_OK; }
    if (eq(name, "up_variant") && value >= 0 && value <= 7) { g_up_variant = value; return PGAN_OK; }
    if (eq(name, "up_order") && value >= 0 && value <= 2) { g_up_order = value; return PGAN_OK; }
    if (eq(name, "down_order") && value >= 0 && value <= 2) { g_dn_order = value; return PGAN_OK; }
    if (eq(name, "wg_bm") && (value == 0 || value == 64 || value == 128)) { g_wg_bm = value; return PGAN_OK; }
    if (eq(name, "wg_target") && (value == 0 || (value >= 64 && value <= 4096))) { g_wg_target = value; return PGAN_OK; }
    if (eq(name, "up_ks") && value >= 0 && value <= 8) { g_up_ks = value; return PGAN_OK; }
    if (eq(name, "up_kg") && value >= 0 && value <= 1) { g_up_kg = value; return PGAN_OK; }
    return PGAN_EINVAL;
}

// Bytes of the prepared-weight buffer for W[Cs][Cl][4][4].
long long FN_PREP_BYTES(int Cs, int Cl) { return (long long)Cs * Cl * 16 * 2 * 2 * NPL + (F16 ? WHDR_BYTES : 0); }

// Split W[Cs][Cl][4][4] (fp32) into the bf16 planes the x3 kernels read.  Run again whenever W changes.
int FN_PREP(const float* W, void* planes, int Cs, int Cl, hipStream_t st) {
    if (Cs <= 0 || Cl <= 0) return PGAN_EINVAL;
    // fp16 split: the buffer starts with a header whose first word holds (the bits of) the largest |w| ever prepared
    // into it -- zero it once when the buffer is created; it only grows, which keeps the scale safe and stable
    uint32_t* hdr = F16 ? static_cast<uint32_t*>(planes) : nullptr;
    uint16_t* dn = reinterpret_cast<uint16_t*>(static_cast<unsigned char*>(planes) + (F16 ? WHDR_BYTES : 0));
    if (F16) {
        const int64_t n4 = (int64_t)Cs * Cl * 4;
        weight_amax_kernel<<<(int)((n4 + 1023) / 1024 < 256 ? (n4 + 1023) / 1024 : 256), 256, 0, st>>>(W, hdr, n4);
    }
    prep_kernel<<<(Cs * Cl + 255) / 256, 256, 0, st>>>(W, dn, dn + (size_t)NPL * Cs * Cl * 16, Cs, Cl, hdr);
    return pgan_launch_status();
}

#ifdef PGAN_SPLIT_F16
// Forget the largest |w| a prepared-weight buffer has seen (its header word), stream-ordered: the next preparation
// starts the record again.  For weights that are unrelated to the ones prepared before (a load, another tensor).
int pgan_conv_h2_forget(void* planes, hipStream_t st) {
    if (!planes) return PGAN_EINVAL;
    return hipMemsetAsync(planes, 0, WHDR_BYTES, st) == hipSuccess ? PGAN_OK : PGAN_ELAUNCH;
}
#endif

// The same for n <= 6 weight tensors in two launches (one in the bf16 split) instead of two per tensor.
int FN_PREP_MULTI(const float* const* W, void* const* planes, const int* Cs, const int* Cl, int n, hipStream_t st) {
    if (n <= 0 || n > PREP_MAX || !W || !planes || !Cs || !Cl) return PGAN_EINVAL;
    PrepSet ps;
    ps.n = n;
    ps.first_amax[0] = ps.first_prep[0] = 0;
    for (int t = 0; t < n; ++t) {
        if (Cs[t] <= 0 || Cl[t] <= 0 || !W[t] || !planes[t]) return PGAN_EINVAL;
        ps.W[t] = W[t];
        ps.planes[t] = static_cast<unsigned char*>(planes[t]);
        ps.Cs[t] = Cs[t];
        ps.Cl[t] = Cl[t];
        const int64_t n4 = (int64_t)Cs[t] * Cl[t] * 4;
        ps.first_amax[t + 1] = ps.first_amax[t] + (int)((n4 + 1023) / 1024 < 256 ? (n4 + 1023) / 1024 : 256);
        ps.first_prep[t + 1] = ps.first_prep[t] + (Cs[t] * Cl[t] + 255) / 256;
    }
    if (F16) weight_amax_multi_kernel<<<ps.first_amax[n], 256, 0, st>>>(ps);
    prep_multi_kernel<<<ps.first_prep[n], 256, 0, st>>>(ps);
    return pgan_launch_status();
}

// S[N,Cs,H,H] = conv2d(L[N,Cl,2H,2H], W, stride 2, pad 1) with W given as prepared planes.  H = hs.
static int down_entry(const float* L, const void* planes, float* S, int N, int Cs, int Cl, int hs, const float* amax,
                      int amax_per_sample, int amax_count, hipStream_t st, int zeroed) {
    if (N <= 0) return PGAN_OK;
    const Bound bx{amax, amax_per_sample, amax_count};
    if (zeroed >= 0 && !bound_ok(bx)) return PGAN_EINVAL;
    const uint32_t* hdr = F16 ? static_cast<const uint32_t*>(planes) : nullptr;
    const uint16_t* wp = planes ? reinterpret_cast<const uint16_t*>(static_cast<const unsigned char*>(planes) + (F16 ? WHDR_BYTES : 0))
                                : nullptr;
    switch (hs) {
        case 4: return launch_down<4>(L, wp, S, N, Cs, Cl, st, hdr, bx, zeroed);
        case 8: return launch_down<8>(L, wp, S, N, Cs, Cl, st, hdr, bx, zeroed);
        case 16: return launch_down<16>(L, wp, S, N, Cs, Cl, st, hdr, bx, zeroed);
        case 32: return launch_down<32>(L, wp, S, N, Cs, Cl, st, hdr, bx, zeroed);
        default: return PGAN_EINVAL;
    }
}
int FN_DOWN(const float* L, const void* planes, float* S, int N, int Cs, int Cl, int hs, const float* amax,
            int amax_per_sample, int amax_count, hipStream_t st) {
    return down_entry(L, planes, S, N, Cs, Cl, hs, amax, amax_per_sample, amax_count, st, 0);
}
// The same for an output that already holds zeros where the launch needs them: launches that split K over two blocks add
// into S and otherwise clear it first with a launch of their own; FN_DOWN_ADDS tells whether a shape is such a launch
// (1 / 0; -1: invalid shape).  For other shapes this is FN_DOWN.
int FN_DOWN_ZEROED(const float* L, const void* planes, float* S, int N, int Cs, int Cl, int hs, const float* amax,
                   int amax_per_sample, int amax_count, hipStream_t st) {
    return down_entry(L, planes, S, N, Cs, Cl, hs, amax, amax_per_sample, amax_count, st, 1);
}
int FN_DOWN_ADDS(int N, int Cs, int Cl, int hs) {
    if (N <= 0) return 0;
    const int r = down_entry(nullptr, nullptr, nullptr, N, Cs, Cl, hs, nullptr, 0, 0, nullptr, -1);
    return r < 0 ? -1 : r;
}

// L[N,Cl,2H,2H] = conv_transpose2d(S[N,Cs,H,H], W, stride 2, pad 1) with W given as prepared planes.
int FN_UP(const float* S, const void* planes, float* L, int N, int Cs, int Cl, int hs, const float* amax,
          int amax_per_sample, int amax_count, hipStream_t st) {
    if (N <= 0) return PGAN_OK;
    const Bound by{amax, amax_per_sample, amax_count};
    if (!bound_ok(by)) return PGAN_EINVAL;
    const uint32_t* hdr = F16 ? static_cast<const uint32_t*>(planes) : nullptr;
    const uint16_t* wp = reinterpret_cast<const uint16_t*>(static_cast<const unsigned char*>(planes) + (F16 ? WHDR_BYTES : 0));
    switch (hs) {
        case 4: return launch_up<4>(S, wp, L, N, Cs, Cl, st, hdr, by);
        case 8: return launch_up<8>(S, wp, L, N, Cs, Cl, st, hdr, by);
        case 16: return launch_up<16>(S, wp, L, N, Cs, Cl, st, hdr, by);
        case 32: return launch_up<32>(S, wp, L, N, Cs, Cl, st, hdr, by);
        default: return PGAN_EINVAL;
    }
}

// Scratch floats pgan_conv_wgrad_x3 needs for these dimensions.
long long FN_WGRAD_WS(int N, int Cs, int Cl, int hs) {
    int ns = 1;
    switch (hs) {
        case 4: ns = wgrad_x3_splits<4>(N, Cs, Cl); break;
        case 8: ns = wgrad_x3_splits<8>(N, Cs, Cl); break;
        case 16: ns = wgrad_x3_splits<16>(N, Cs, Cl); break;
        case 32: ns = wgrad_x3_splits<32>(N, Cs, Cl); break;
        default: return -1;
    }
    return (long long)ns * Cs * Cl * 16;
}

// dW[Cs,Cl,4,4] = sum_n S[n] (x) L[r(n)] as pgan_conv_wgrad, with the six-product bf16 split of both operands.
int FN_WGRAD(const float* S, const float* L, float* dW, float* ws, long long ws_floats, int N, int Cs, int Cl, int hs,
             int n_keep, int n_swap, int accumulate, const float* amax_s, int amax_s_per_sample, int amax_s_count,
             const float* amax_l, int amax_l_per_sample, int amax_l_count, hipStream_t st) {
    if (N <= 0) return PGAN_EINVAL;
    const Bound bs{amax_s, amax_s_per_sample, amax_s_count}, bl{amax_l, amax_l_per_sample, amax_l_count};
    if (!bound_ok(bs) || !bound_ok(bl)) return PGAN_EINVAL;
    switch (hs) {
        case 4: return launch_wgrad<4>(S, L, dW, ws, ws_floats, N, Cs, Cl, n_keep, n_swap, accumulate, st, bs, bl);
        case 8: return launch_wgrad<8>(S, L, dW, ws, ws_floats, N, Cs, Cl, n_keep, n_swap, accumulate, st, bs, bl);
        case 16: return launch_wgrad<16>(S, L, dW, ws, ws_floats, N, Cs, Cl, n_keep, n_swap, accumulate, st, bs, bl);
        case 32: return launch_wgrad<32>(S, L, dW, ws, ws_floats, N, Cs, Cl, n_keep, n_swap, accumulate, st, bs, bl);
        default: return PGAN_EINVAL;
    }
}

#if defined(PGAN_UP_TIMING) && defined(PGAN_SPLIT_F16)
// (debugging builds) the notes of the last UP launch: 64 blocks x 8 clock values, to host memory
int pgan_h2_debug_up_stamps(unsigned long long* host_out) {
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_up_stamps), sizeof(unsigned long long) * 64 * 8) == hipSuccess ? PGAN_OK : PGAN_ELAUNCH;
}
#endif

}  // extern "C"
