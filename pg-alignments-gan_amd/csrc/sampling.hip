// Device-side random sampling steps around the training step, all driven by the same Philox4x32-10 state words as
// pgan_philox (state[0..1] seed, state[2..3] 64-bit offset in units of one Philox call = 4 outputs):
//   * per-batch augmentations of the real alignments: label smoothing (src/train_wgan_v2.py:265-268) and monomorphic
//     site switching (:129-160,523-524), drawn afresh for every batch inside the recorded step;
//   * replay buffer of generated alignments (FakeBuffer, src/train_wgan_v2.py:98-117): bounded FIFO + a batch drawn
//     from it without replacement, as three tiny launches so that the critic step that uses it stays inside a
//     recorded launch tape / hipGraph (no host random numbers, no host integers);
//   * genotype sampling of generated alignments (sample_genotypes_bernoulli, src/train_wgan_v2.py:162-192).
#include "pgan_common.h"

namespace {

__device__ __forceinline__ void philox4(uint64_t ctr, uint32_t k0, uint32_t k1, uint32_t (&c)[4]) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
    c[0] = (uint32_t)ctr; c[1] = (uint32_t)(ctr >> 32); c[2] = 0u; c[3] = 0u;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(M0, c[0]), lo0 = M0 * c[0];
        const uint32_t hi1 = __umulhi(M1, c[2]), lo1 = M1 * c[2];
        const uint32_t n0 = hi1 ^ c[1] ^ k0, n2 = hi0 ^ c[3] ^ k1;
        c[0] = n0; c[1] = lo1; c[2] = n2; c[3] = lo0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
}
__device__ __forceinline__ uint64_t philox_offset(const uint32_t* state) { return ((uint64_t)state[3] << 32) | state[2]; }
__device__ __forceinline__ void philox_bump(uint32_t* state, uint64_t calls) {
    const uint64_t off = philox_offset(state) + calls;
    state[2] = (uint32_t)off;
    state[3] = (uint32_t)(off >> 32);
}

// ---------------- replay buffer ----------------
// ring state: st[0] = head (next slot to overwrite), st[1] = count (valid alignments, <= size).
// Logical index u (0 = oldest, count - 1 = newest: the order of the reference's deque) lives in physical slot
// (head - count + u) mod size.

// append: new alignment b goes to slot (head + b) mod size.  Reads the state only.
__global__ void fakebuf_append_kernel(float* __restrict__ buf_s, float* __restrict__ buf_p,
                                      const float* __restrict__ new_s, const float* __restrict__ new_p,
                                      const int* __restrict__ st, int size, int B, int Ms, int Mp) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int head = st[0];
    if (i < (int64_t)B * Ms / 4) {
        const int b = (int)(i / (Ms / 4)), e = (int)(i % (Ms / 4));
        reinterpret_cast<f32x4*>(buf_s + (int64_t)((head + b) % size) * Ms)[e] = reinterpret_cast<const f32x4*>(new_s)[i];
    }
    if (i < (int64_t)B * Mp / 4) {
        const int b = (int)(i / (Mp / 4)), e = (int)(i % (Mp / 4));
        reinterpret_cast<f32x4*>(buf_p + (int64_t)((head + b) % size) * Mp)[e] = reinterpret_cast<const f32x4*>(new_p)[i];
    }
}

// draw: one block.  Advances the ring state by the B alignments just appended, then picks B distinct logical indices
// in [0, count) -- a partial Fisher-Yates shuffle of 0..count-1 held in LDS, one 32-bit Philox word per pick (or the
// caller's indices, for tests) -- and writes their physical slots.  picks_in: B logical indices or null.
__global__ __launch_bounds__(256) void fakebuf_draw_kernel(int* __restrict__ st, int* __restrict__ slots,
                                                          const int* __restrict__ picks_in, uint32_t* __restrict__ rng,
                                                          int size, int B) {
    extern __shared__ int perm[];
    __shared__ int bad;
    const int head = (st[0] + B) % size;
    const int count = min(size, st[1] + B);
    if (threadIdx.x == 0) bad = 0;
    for (int i = threadIdx.x; i < count; i += 256) perm[i] = i;
    __syncthreads();
    if (picks_in) {
        for (int k = threadIdx.x; k < B; k += 256) {
            const int u = picks_in[k];
            if (u < 0 || u >= count) bad = 1;
            slots[k] = ((head - count + (u < 0 ? 0 : u % count)) % size + size) % size;
        }
    } else if (threadIdx.x == 0) {
        const uint64_t off = philox_offset(rng);
        uint32_t c[4];
        for (int k = 0; k < B; ++k) {
            if ((k & 3) == 0) philox4(off + (uint64_t)(k >> 2), rng[0], rng[1], c);
            const int j = k + (int)__umulhi(c[k & 3], (uint32_t)(count - k));
            const int v = perm[j];
            perm[j] = perm[k];
            perm[k] = v;
            slots[k] = ((head - count + v) % size + size) % size;
        }
        philox_bump(rng, (uint64_t)((B + 3) / 4));
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        st[0] = head;
        st[1] = count;
        st[2] |= bad;          // sticky error flag: an injected index outside the buffer
    }
}

__global__ void fakebuf_gather_kernel(const float* __restrict__ buf_s, const float* __restrict__ buf_p,
                                      const int* __restrict__ slots, float* __restrict__ out_s,
                                      float* __restrict__ out_p, int B, int Ms, int Mp) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < (int64_t)B * Ms / 4) {
        const int b = (int)(i / (Ms / 4)), e = (int)(i % (Ms / 4));
        reinterpret_cast<f32x4*>(out_s)[i] = reinterpret_cast<const f32x4*>(buf_s + (int64_t)slots[b] * Ms)[e];
    }
    if (i < (int64_t)B * Mp / 4) {
        const int b = (int)(i / (Mp / 4)), e = (int)(i % (Mp / 4));
        reinterpret_cast<f32x4*>(out_p)[i] = reinterpret_cast<const f32x4*>(buf_p + (int64_t)slots[b] * Mp)[e];
    }
}

// ---------------- genotype sampling ----------------
// One thread per (alignment, column): ROWS Bernoulli draws down the column (u < p, as torch.bernoulli); a column that
// came out monomorphic is drawn again and one individual, picked in proportion to its probability of carrying the
// missing allele (+1e-7), is set to it.  GT_CALLS Philox calls are reserved per column (2 ROWS + 1 uniforms at most).
constexpr int GT_CALLS = 68;

template <int ROWS>
__global__ __launch_bounds__(256) void sample_genotypes_kernel(const float* __restrict__ p, float* __restrict__ out,
                                                               int64_t ncol, int S, const uint32_t* __restrict__ rng,
                                                               int require_variation) {
    constexpr int W = ROWS / 64;
    static_assert(ROWS % 64 == 0 && 2 * (ROWS / 4) + 1 <= GT_CALLS, "column length");
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= ncol) return;
    const int64_t n = t / S;
    const int j = (int)(t % S);
    const float* col = p + n * ROWS * S + j;
    float* ocol = out + n * ROWS * S + j;
    const uint64_t base = philox_offset(rng) + (uint64_t)t * GT_CALLS;
    const uint32_t k0 = rng[0], k1 = rng[1];
    uint32_t c[4];
    uint64_t m[W];
    auto draw_column = [&](int call0) {
#pragma unroll
        for (int w = 0; w < W; ++w) m[w] = 0;
        for (int i = 0; i < ROWS; ++i) {
            if ((i & 3) == 0) philox4(base + call0 + (i >> 2), k0, k1, c);
            const float u = (c[i & 3] >> 8) * (1.0f / 16777216.0f);
            if (u < col[(int64_t)i * S]) m[i >> 6] |= 1ull << (i & 63);
        }
    };
    auto uniform = [&](uint64_t v) {
        bool all = true;
#pragma unroll
        for (int w = 0; w < W; ++w) all = all && m[w] == v;
        return all;
    };
    draw_column(0);
    if (require_variation && (uniform(0ull) || uniform(~0ull))) {
        const bool have = m[0] != 0ull;                    // the allele every individual carries
        double total = 0.0;
        for (int i = 0; i < ROWS; ++i) {
            const float q = col[(int64_t)i * S];
            total += (double)(have ? 1.f - q : q) + 1e-7;
        }
        draw_column(ROWS / 4);
        philox4(base + 2 * (ROWS / 4), k0, k1, c);
        const double target = (double)((c[0] >> 8) * (1.0f / 16777216.0f)) * total;
        double acc = 0.0;
        int pick = ROWS - 1;
        for (int i = 0; i < ROWS; ++i) {
            const float q = col[(int64_t)i * S];
            acc += (double)(have ? 1.f - q : q) + 1e-7;
            if (target < acc) { pick = i; break; }
        }
        if (have) m[pick >> 6] &= ~(1ull << (pick & 63));
        else m[pick >> 6] |= 1ull << (pick & 63);
    }
    for (int i = 0; i < ROWS; ++i) ocol[(int64_t)i * S] = (float)((m[i >> 6] >> (i & 63)) & 1ull);
}
// ---------------- per-batch augmentations of the real alignments ----------------
// Label smoothing: x' = x (1 - ey) + centre ey with ey ~ U(0, noise), one draw per element; value v of the batch's draw
// comes from Philox call (skip + v) / 4.  normalized: x holds (raw - 0.5) / 0.5 (the pool is normalised once when it is
// loaded; the reference smooths the raw values inside get_batch and normalises afterwards, :265-268 then :519-521).
// ey_given: the draws themselves (tests).
__global__ void label_smooth_kernel(float* __restrict__ x, int64_t n4, int64_t skip4, float noise, float centre,
                                    int normalized, const uint32_t* __restrict__ rng, const float* __restrict__ ey_given) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    f32x4 ey;
    if (ey_given) {
        ey = reinterpret_cast<const f32x4*>(ey_given)[i];
    } else {
        uint32_t c[4];
        philox4(philox_offset(rng) + (uint64_t)(skip4 + i), rng[0], rng[1], c);
#pragma unroll
        for (int e = 0; e < 4; ++e) ey[e] = noise * ((c[e] >> 8) * (1.0f / 16777216.0f));
    }
    f32x4 v = reinterpret_cast<f32x4*>(x)[i];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float raw = normalized ? fmaf(v[e], 0.5f, 0.5f) : v[e];
        const float sm = fmaf(raw, 1.f - ey[e], centre * ey[e]);
        v[e] = normalized ? fmaf(sm, 2.f, -1.f) : sm;
    }
    reinterpret_cast<f32x4*>(x)[i] = v;
}

// switch_monomorphic (src/train_wgan_v2.py:129-160), one block per alignment.  Thread 0 makes the alignment's draws --
// n ~ U{lower..upper} columns (random.randint), an ordered sample of n distinct columns (np.random.choice without
// replacement: partial Fisher-Yates), one uniform per chosen column -- and redraws the chosen columns' positions one
// after the other between the CURRENT neighbours (0 below the first column, 1 above the last), in double precision as
// numpy does, so a neighbour redrawn earlier in the loop already counts; then all threads write the columns: the first
// n / 2 all 0.0, the rest all 1.0 (these literals also after --normalize, as in the reference, where the switch runs
// on the normalised batch).  MONO_CALLS Philox calls are reserved per alignment, keyed by its index in the global batch.
constexpr int MONO_CALLS = 8;       // 1 + 2 * MONO_MAX words
constexpr int MONO_MAX = 15;        // most columns switched per alignment
__global__ __launch_bounds__(64) void mono_switch_kernel(float* __restrict__ sites, float* __restrict__ pos, int rows, int S,
                                                         int lower, int upper, const uint32_t* __restrict__ rng,
                                                         const int* __restrict__ n_given, const int* __restrict__ cols_given,
                                                         const float* __restrict__ u_given, int first) {
    __shared__ int perm[128];
    __shared__ int cols[MONO_MAX];
    __shared__ int count;
    const int a = blockIdx.x;
    float* P = pos + (int64_t)a * S;
    for (int i = threadIdx.x; i < S; i += blockDim.x) perm[i] = i;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t w[4 * MONO_CALLS];
        int n;
        if (n_given) {
            n = min(max(n_given[a], 0), min(upper, S));
        } else {
            const uint64_t base = philox_offset(rng) + (uint64_t)(first + a) * MONO_CALLS;
#pragma unroll
            for (int k = 0; k < MONO_CALLS; ++k) {
                uint32_t c[4];
                philox4(base + k, rng[0], rng[1], c);
#pragma unroll
                for (int e = 0; e < 4; ++e) w[4 * k + e] = c[e];
            }
            n = lower + (int)__umulhi(w[0], (uint32_t)(upper - lower + 1));
        }
        for (int k = 0; k < n; ++k) {
            int col;
            if (cols_given) {
                col = min(max(cols_given[a * upper + k], 0), S - 1);
            } else {
                const int j = k + (int)__umulhi(w[1 + k], (uint32_t)(S - k));
                col = perm[j];
                perm[j] = perm[k];
                perm[k] = col;
            }
            cols[k] = col;
        }
        for (int k = 0; k < n; ++k) {
            const int i = cols[k];
            const float u = u_given ? u_given[a * upper + k] : (w[1 + MONO_MAX + k] >> 8) * (1.0f / 16777216.0f);
            const double lo = i == 0 ? 0.0 : (double)P[i - 1], hi = i == S - 1 ? 1.0 : (double)P[i + 1];
            P[i] = (float)(lo + (hi - lo) * (double)u);
        }
        count = n;
    }
    __syncthreads();
    const int n = count;
    float* X = sites + (int64_t)a * rows * S;
    for (int k = 0; k < n; ++k) {
        const float v = k < n / 2 ? 0.f : 1.f;
        for (int r = threadIdx.x; r < rows; r += blockDim.x) X[(int64_t)r * S + cols[k]] = v;
    }
}
__global__ void rng_advance_kernel(uint32_t* rng, uint64_t calls) { philox_bump(rng, calls); }

}  // namespace

extern "C" {

// sites[n] <- label-smoothed sites (see label_smooth_kernel): n values of a batch whose first value is number `skip` of
// a draw of `total` values (the global batch of a data-parallel run; one GPU: skip 0, total n).  n, skip % 4 == 0.
int pgan_label_smooth(float* sites, long long n, float noise, float centre, int normalized, unsigned int* rng,
                      const float* ey_given, long long skip, long long total, hipStream_t st) {
    if (n <= 0 || n % 4 || skip < 0 || skip % 4 || skip + n > total || !sites || (!rng && !ey_given)) return PGAN_EINVAL;
    label_smooth_kernel<<<(unsigned)((n / 4 + 255) / 256), 256, 0, st>>>(sites, n / 4, skip / 4, noise, centre, normalized,
                                                                       rng, ey_given);
    if (!ey_given) rng_advance_kernel<<<1, 1, 0, st>>>(rng, (uint64_t)((total + 3) / 4));
    return pgan_launch_status();
}

// switch_monomorphic on a batch of B alignments (sites [B][1][rows][S], pos [B][1][S]) that are alignments
// first .. first + B - 1 of a global batch of `total`; n_given / cols_given / u_given ([B], [B][upper], [B][upper]) replace
// the draws (tests); upper <= 15, S <= 128.
int pgan_mono_switch(float* sites, float* pos, int B, int rows, int S, int lower, int upper, unsigned int* rng,
                     const int* n_given, const int* cols_given, const float* u_given, int first, int total,
                     hipStream_t st) {
    if (B <= 0 || rows <= 0 || S <= 1 || S > 128 || lower < 0 || upper < lower || upper > MONO_MAX || upper > S || !sites ||
        !pos || first < 0 || first + B > total)
        return PGAN_EINVAL;
    const bool given = n_given && cols_given && u_given;
    if (!given && (!rng || n_given || cols_given || u_given)) return PGAN_EINVAL;      // all three or none
    mono_switch_kernel<<<B, 64, 0, st>>>(sites, pos, rows, S, lower, upper, rng, n_given, cols_given, u_given, first);
    if (!given) rng_advance_kernel<<<1, 1, 0, st>>>(rng, (uint64_t)total * MONO_CALLS);
    return pgan_launch_status();
}

// One replay-buffer step: append the B alignments new_s / new_p (Ms / Mp floats each) to the ring buf_s / buf_p
// (`size` <= 16320 slots), then write B alignments drawn from the buffer without replacement to out_s / out_p.
// state: 4 device ints (head, count, error flag, unused), zero at the start; slots: B device ints of scratch;
// rng: Philox state as pgan_philox; picks: null, or B device ints = logical indices (0 oldest) used instead of
// random draws (state[2] becomes 1 if one lies outside the buffer).
int pgan_fakebuf_step(float* buf_s, float* buf_p, int size, int* state, int* slots, const float* new_s,
                      const float* new_p, float* out_s, float* out_p, int B, int Ms, int Mp, unsigned int* rng,
                      const int* picks, hipStream_t st) {
    if (B <= 0 || size < B || size > 16320 || Ms % 4 || Mp % 4 || Mp > Ms || !state || !slots || (!rng && !picks))
        return PGAN_EINVAL;
    const int grid = (int)(((int64_t)B * Ms / 4 + 255) / 256);
    fakebuf_append_kernel<<<grid, 256, 0, st>>>(buf_s, buf_p, new_s, new_p, state, size, B, Ms, Mp);
    fakebuf_draw_kernel<<<1, 256, (size_t)size * sizeof(int), st>>>(state, slots, picks, rng, size, B);
    fakebuf_gather_kernel<<<grid, 256, 0, st>>>(buf_s, buf_p, slots, out_s, out_p, B, Ms, Mp);
    return pgan_launch_status();
}

// out[n][1][rows][S] ~ Bernoulli(p) with every column polymorphic when require_variation != 0 (see the kernel);
// rows = 64 or 128 individuals.
int pgan_sample_genotypes(const float* p, float* out, long long n, int rows, int S, unsigned int* rng,
                          int require_variation, hipStream_t st) {
    if (n <= 0 || S <= 0 || !p || !out || !rng || (rows != 64 && rows != 128)) return PGAN_EINVAL;
    const int64_t ncol = (int64_t)n * S;
    const unsigned grid = (unsigned)((ncol + 255) / 256);
    if (rows == 64) sample_genotypes_kernel<64><<<grid, 256, 0, st>>>(p, out, ncol, S, rng, require_variation);
    else sample_genotypes_kernel<128><<<grid, 256, 0, st>>>(p, out, ncol, S, rng, require_variation);
    rng_advance_kernel<<<1, 1, 0, st>>>(rng, (uint64_t)ncol * GT_CALLS);
    return pgan_launch_status();
}

}  // extern "C"
