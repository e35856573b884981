/* C-ABI of libpgan_hip.so: hand-written gfx950 (MI355X) kernels for the DCWGAN-GP training step of
 * PG-Alignments-GAN.
 *
 * Conventions for every entry point: plain device pointers and sizes, fp32 tensors contiguous in
 * NCHW / NCL order, work is only ENQUEUED on the caller's hipStream_t (no synchronisation, no
 * allocation, no host reads), scratch comes from the caller, return 0 on success, -1 for an
 * unsupported shape or argument (nothing was launched), -2 if HIP reported a launch error, -3 for a
 * size beyond the fixed maximum of pgan_gn_fwd_static.
 *
 * The reference has no native boundary (it is pure PyTorch); each entry point replaces the ATen
 * op(s) that the cited reference line dispatches to.  Paths are relative to /root/reference.
 */
#ifndef PGAN_HIP_H
#define PGAN_HIP_H

#include <hip/hip_runtime_api.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- k=4, stride=2, pad=1 convolution pair, weight W[Cs][Cl][4][4] between a small side
 * S[N][Cs][hs][hs] and a large side L[N][Cl][2hs][2hs].  Cl > 1: fp32 MFMA implicit GEMM
 * (Cs % 64 == 0, Cl % 64 == 0 for up / % 8 for down and wgrad, hs in {4, 8, 16}); Cl == 1: hs == 32 or 64, Cs <= 64.
 * The bf16-pipe entry points below also take hs == 32 (Cl % 32 == 0): the extra stage of the 128 x 128 network. */

/* Kernel-variant overrides for tuning runs (no counterpart in the reference): "down_kg" = 1|2|4 K groups
 * of the DOWN kernel, "up_mode" = 1 all-phase tiles | 2 phase-split | 3 phase-split + 2 K groups; 0 = auto. */
int pgan_set_option(const char* name, int value);

/* S = conv2d(L, W, stride 2, pad 1): nn.Conv2d forward, src/layers.py:441-463 (discriminator);
 * input-gradient of nn.ConvTranspose2d, src/layers.py:97-112 (generator backward). */
int pgan_conv_down(const float* L, const float* W, float* S, int N, int Cs, int Cl, int hs, hipStream_t st);
/* The same for the discriminator's first stage (Cl == 1, hs == 32, Cs == 64; src/layers.py:441) with the statistics pass
 * of the GroupNorm that follows it (:442) folded in: the kernel also leaves the [N][8][2] slice partial sums (sum, sum of
 * squares; doubles) of S in gn_ws, the workspace of the pgan_gn_fwd_presummed call that must follow on the same stream.
 * Any other shape returns -1 and launches nothing. */
int pgan_conv_down_gn_sums(const float* L, const float* W, float* S, int N, int Cs, int Cl, int hs, void* gn_ws,
                           long long gn_ws_bytes, hipStream_t st);

/* L = conv_transpose2d(S, W, stride 2, pad 1): nn.ConvTranspose2d forward, src/layers.py:97-112, with
 * the trailing nn.Sigmoid (:114) / nn.Tanh (:255) fused when act = 1 / 2 (Cl == 1 only);
 * input-gradient of nn.Conv2d (autograd through src/layers.py:441-463, src/train_wgan_v2.py:58-60,609). */
int pgan_conv_up(const float* S, const float* W, float* L, int N, int Cs, int Cl, int hs, int act, hipStream_t st);

/* dW = sum_n S[n] (x) L[r(n)] (weight gradient of both of the above; aten::convolution_backward
 * reached from src/train_wgan_v2.py:609,633).  r(n) = n for n < n_keep, n + n_swap for
 * n < n_keep + n_swap, n - n_swap otherwise (pairs the engine's adjoint slots with its activation
 * slots).  accumulate != 0 adds to dW instead of overwriting it (a gradient assembled from several passes over
 * different sample ranges).  ws: scratch of pgan_conv_wgrad_workspace_floats(...) floats. */
long long pgan_conv_wgrad_workspace_floats(int N, int Cs, int Cl, int hs);
int pgan_conv_wgrad(const float* S, const float* L, float* dW, float* ws, long long ws_floats, int N, int Cs, int Cl,
                    int hs, int n_keep, int n_swap, int accumulate, hipStream_t st);

/* out[c] (+)= sum of part[r * ld + c] over rows [r0,r1) and [r2,r3), c < C <= ld: finishes the
 * per-sample partial sums of the norm-parameter and 1-D-tower gradients. */
int pgan_rows_reduce(const float* part, int ld, float* out, int C, int r0, int r1, int r2, int r3, int accumulate,
                     hipStream_t st);

/* ---- The same convolutions evaluated on the bf16 matrix pipe with fp32 accuracy: every fp32 operand is split
 * into three bf16 terms and a product is the sum of its six significant partial products, accumulated in
 * fp32 (error below that of an fp32 multiply).  Weights are passed as prepared planes:
 *   planes = device buffer of pgan_conv_x3_prep_bytes(Cs, Cl) bytes filled by pgan_conv_x3_prep(W, ...);
 *   prepare again whenever W changes.  Same reference ops as pgan_conv_down / pgan_conv_up
 *   (src/layers.py:441-463 Conv2d, :97-112 ConvTranspose2d; Cl >= 32, no fused activation). */
long long pgan_conv_x3_prep_bytes(int Cs, int Cl);
int pgan_conv_x3_prep(const float* W, void* planes, int Cs, int Cl, hipStream_t st);
/* the same for n <= 6 weight tensors (host arrays of n pointers / sizes) in one launch: a network's stages together */
int pgan_conv_x3_prep_multi(const float* const* W, void* const* planes, const int* Cs, const int* Cl, int n,
                            hipStream_t st);
/* amax, amax_per_sample, amax_count: optional magnitude bounds of the activation operand, used only by the fp16 split
 * below (this split ignores them; pass NULL, 0, 0). */
int pgan_conv_down_x3(const float* L, const void* planes, float* S, int N, int Cs, int Cl, int hs, const float* amax,
                      int amax_per_sample, int amax_count, hipStream_t st);
int pgan_conv_down_x3_zeroed(const float* L, const void* planes, float* S, int N, int Cs, int Cl, int hs, const float* amax,
                             int amax_per_sample, int amax_count, hipStream_t st);      /* see pgan_conv_down_h2_zeroed */
int pgan_conv_down_x3_adds(int N, int Cs, int Cl, int hs);
int pgan_conv_up_x3(const float* S, const void* planes, float* L, int N, int Cs, int Cl, int hs, const float* amax,
                    int amax_per_sample, int amax_count, hipStream_t st);
/* weight gradient with both operands split in the kernel (same arguments and scratch contract as pgan_conv_wgrad) */
long long pgan_conv_wgrad_x3_workspace_floats(int N, int Cs, int Cl, int hs);
int pgan_conv_wgrad_x3(const float* S, const float* L, float* dW, float* ws, long long ws_floats, int N, int Cs, int Cl,
                       int hs, int n_keep, int n_swap, int accumulate, const float* amax_s, int amax_s_per_sample,
                       int amax_s_count, const float* amax_l, int amax_l_per_sample, int amax_l_count, hipStream_t st);
/* tile overrides of those kernels for tuning runs ("down_variant", "down_ks", "up_variant" (6, 7: two wave groups
 * along K per block), "up_ks", "up_kg" = 1: never two wave groups in UP, "up_order", "down_order", "wg_bm",
 * "wg_target" = blocks the weight-gradient split aims at; 0 = by size) */
int pgan_x3_set_option(const char* name, int value);

/* ---- The same convolutions with two fp16 planes per operand and three partial products per fp32 product
 * (x = h1 + 2^-11 h2; the dropped h2*g2 term is below 2^-22 |x y|): half the matrix work of the bf16 split at the
 * accuracy of an fp32 dot product.  fp16's narrow exponent range is handled inside the calls by exact power-of-two
 * scaling: every block scales the activations it stages by their own largest magnitude, a weight tensor by the
 * largest |w| its planes buffer has seen.  That running maximum lives in the first 256 bytes of the planes buffer:
 *   planes = device buffer of pgan_conv_h2_prep_bytes(Cs, Cl) bytes, ZEROED once by the caller before its first
 *   pgan_conv_h2_prep (and zeroed again to forget the history, e.g. after loading other weights); every
 *   pgan_conv_h2_prep(W, ...) raises the maximum to max|W| and rewrites the planes; prepare again whenever W changes.
 * Argument lists, layouts, return codes and the reference ops are those of the _x3 calls above
 * (src/layers.py:441-463 Conv2d, :97-112 ConvTranspose2d; Cl >= 32).
 * Scanning the activations costs a block a pass over its operands before it can stage them (15-30 % of these
 * kernels' time).  A caller that knows upper bounds of the operand's magnitude -- the norm entry points below
 * report them for their outputs -- passes them instead:
 *   amax != NULL, amax_per_sample = k in 1..64: amax[n * k + i], i < k, each bound sample n of the call
 *     (pgan_gn_* write k = 8 per sample);
 *   amax != NULL, amax_per_sample = 0: every one of amax[0 .. amax_count) bounds the whole operand
 *     (pgan_bn_* write one entry per channel);
 *   amax == NULL: the kernel scans.
 * Bounds may be loose (a factor 2^k costs k of the 2^28 of dynamic range below the bound); a bound below the true
 * magnitude can overflow fp16.  The weight-gradient call takes one bound per operand; rows of a per-sample bound of
 * L follow L's own sample order (the call applies r(n) to them). */
long long pgan_conv_h2_prep_bytes(int Cs, int Cl);
int pgan_conv_h2_prep(const float* W, void* planes, int Cs, int Cl, hipStream_t st);
/* zeroes the header of a planes buffer on st (the running maximum starts again at the next preparation) */
int pgan_conv_h2_forget(void* planes, hipStream_t st);
int pgan_conv_h2_prep_multi(const float* const* W, void* const* planes, const int* Cs, const int* Cl, int n,
                            hipStream_t st);      /* n <= 6 tensors in two launches */
int pgan_conv_down_h2(const float* L, const void* planes, float* S, int N, int Cs, int Cl, int hs, const float* amax,
                      int amax_per_sample, int amax_count, hipStream_t st);
/* pgan_conv_down_h2 for an output that already holds zeros where the launch needs them.  A launch that splits its
 * reduction over two blocks per output tile (small grids: pgan_conv_down_h2_adds(N, Cs, Cl, hs) == 1) adds into S and
 * otherwise clears S first with a launch of its own in the middle of the caller's chain; a caller that clears the
 * outputs of a whole step together beforehand (pgan_zero_multi) saves those launches.  For other shapes the call is
 * pgan_conv_down_h2.  (_x3_zeroed / _x3_adds: the same for the bf16 split.)  Reference op as pgan_conv_down. */
int pgan_conv_down_h2_zeroed(const float* L, const void* planes, float* S, int N, int Cs, int Cl, int hs, const float* amax,
                             int amax_per_sample, int amax_count, hipStream_t st);
int pgan_conv_down_h2_adds(int N, int Cs, int Cl, int hs);
int pgan_conv_up_h2(const float* S, const void* planes, float* L, int N, int Cs, int Cl, int hs, const float* amax,
                    int amax_per_sample, int amax_count, hipStream_t st);
long long pgan_conv_wgrad_h2_workspace_floats(int N, int Cs, int Cl, int hs);
int pgan_conv_wgrad_h2(const float* S, const float* L, float* dW, float* ws, long long ws_floats, int N, int Cs, int Cl,
                       int hs, int n_keep, int n_swap, int accumulate, const float* amax_s, int amax_s_per_sample,
                       int amax_s_count, const float* amax_l, int amax_l_per_sample, int amax_l_count, hipStream_t st);
int pgan_h2_set_option(const char* name, int value);

/* up to 4 such reductions in one launch (the norm-parameter gradients of the four stages) */
int pgan_rows_reduce_multi(const float* const* part, const int* ld, float* const* out, const int* C, int nseg, int r0,
                           int r1, int r2, int r3, hipStream_t st);

/* ---- GroupNorm(1,C) + LeakyReLU(0.1), per-sample statistics (src/layers.py:442-443 and the three
 * repeats below it).  a, h, u, d, t, hd, hbar, p: [N][C][HW]; stats*: [N][2]; pg, pb: rows of per-channel sums, row stride pld.
 * C <= 512, C % 8 == 0, HW a power of two >= 16.  ws: scratch of pgan_gn_workspace_bytes(N) bytes (per-slice partial
 * sums; every op is a sums launch plus an apply launch over 8 slices per sample).
 * amax (every op; may be NULL): [N][8] output, the largest magnitude of each of the 8 slices of every sample of the
 * op's main output (h, d, hd, p) -- the per-sample bounds the fp16-split convolutions take for that operand. */
long long pgan_gn_workspace_bytes(int N);
int pgan_gn_fwd(const float* a, const float* gamma, const float* beta, float* h, float* stats, void* ws,
                long long ws_bytes, int N, int C, int HW, float* amax, hipStream_t st);
/* pgan_gn_fwd whose slice partial sums are already in ws (written by pgan_conv_down_gn_sums): the apply launch only. */
int pgan_gn_fwd_presummed(const float* a, const float* gamma, const float* beta, float* h, float* stats, const void* ws,
                          long long ws_bytes, int N, int C, int HW, float* amax, hipStream_t st);
/* pgan_gn_fwd for callers that cannot allocate device memory or bring a workspace (code under hipGraph capture, tight
 * serving loops, harnesses that forbid hipMalloc).  No counterpart in the reference.  Same result as pgan_gn_fwd; the
 * slice partial sums live in a fixed array inside the library (384 KiB per GPU, for up to
 * pgan_gn_fwd_static_max_samples() = 2048 samples per call; more returns -3 and launches nothing).  Calls no hipMalloc /
 * hipFree.  All calls on one GPU share that array: calls that overlap there (different streams) race.  The array's
 * address on the stream's device is looked up at the first call there; a first call whose stream is being captured is
 * refused (-1) instead -- pgan_gn_fwd_static_init() does the lookup for the current device ahead of time. */
int pgan_gn_fwd_static(const float* a, const float* gamma, const float* beta, float* h, float* stats, int N, int C,
                       int HW, hipStream_t st);
int pgan_gn_fwd_static_init(void);
int pgan_gn_fwd_static_max_samples(void);
/* first-order backward (autograd through GroupNorm/LeakyReLU): d = dL/da from u = dL/dh;
 * stats_b = (mean w, mean w*xhat), w = gamma*m*u; pg/pb = per-sample sums for dL/dgamma, dL/dbeta.
 * The LeakyReLU mask m (this op, pgan_gn_jvp, pgan_gn_bwd2): with beta == NULL it is read from h, the forward output;
 * with beta given it is recomputed from a as the forward pass computed it (sign of gamma*xhat + beta) and h is not read
 * (it may be NULL) -- one tensor fewer to stream in kernels that are bound by the bytes they move. */
int pgan_gn_bwd(const float* u, const float* h, const float* a, const float* stats, const float* gamma, const float* beta,
                float* d, float* stats_b, float* pg, float* pb, int pld /* row stride of pg, pb */, void* ws,
                long long ws_bytes, int N, int C, int HW, float* amax, hipStream_t st);
/* One-launch forms of the GroupNorm ops (same reference ops, src/layers.py:442-466 and autograd's backward / double
 * backward through them): a block keeps the eighth of a sample it read for the sums in registers and applies from there,
 * so every input is read once; the eight blocks of a sample exchange their partial sums through ws1 inside the launch.
 * ws1: pgan_gn_onepass_workspace_bytes(N) bytes of device memory, one per stream that issues these calls, passed ONCE
 * through pgan_gn_onepass_workspace_init after allocation (every call leaves it in that state again; do not share it
 * between streams).  Shapes: C * HW / 8192 in {1, 2, 4, 8}; others return -1 and launch nothing (use the two-launch
 * call).  Results equal the two-launch call's bit for bit.  A block never waits unboundedly for its sample's other
 * blocks: if it gives up (a workspace that was not initialised, or shared), the call's results are wrong and
 * pgan_gn_onepass_workspace_error returns 1 from then on. */
long long pgan_gn_onepass_workspace_bytes(int N);
int pgan_gn_onepass_workspace_init(void* ws1, long long ws1_bytes, int N, hipStream_t st);
int pgan_gn_onepass_workspace_error(const void* ws1, int N, hipStream_t st);      /* synchronises st */
int pgan_gn_bwd_onepass(const float* u, const float* h, const float* a, const float* stats, const float* gamma,
                        const float* beta, float* d, float* stats_b, float* pg, float* pb, int pld, void* ws1,
                        long long ws1_bytes, int N, int C, int HW, float* amax, hipStream_t st);
/* tangent (JVP) of the same block: hd from the tangent t of a; stats_t = (mean t, mean xhat*t). */
int pgan_gn_jvp(const float* t, const float* h, const float* a, const float* stats, const float* gamma, const float* beta,
                float* hd, float* stats_t, void* ws, long long ws_bytes, int N, int C, int HW, float* amax, hipStream_t st);
/* second-order backward: the piece of autograd's double backward (create_graph=True at
 * src/layers.py:51-53,66-74) that belongs to this block.  hbar may be NULL (= 0). */
int pgan_gn_bwd2(const float* hbar, const float* u, const float* t, const float* h, const float* a, const float* stats,
                 const float* stats_b, const float* stats_t, const float* gamma, const float* beta, float* p, float* pg,
                 float* pb, int pld, void* ws, long long ws_bytes, int N, int C, int HW, float* amax, hipStream_t st);

/* ---- BatchNorm (training mode) + LeakyReLU(0.1), src/layers.py:93-110,119-135.  a, h: [B][C][HW],
 * HW % 4 == 0; stats: [2][C] = batch mean, rstd; running stats updated in place (unbiased variance). */
/* ws: scratch of pgan_bn_workspace_bytes(C) bytes (large layers run as a sums launch plus an apply launch over
 * several blocks per channel).
 * amax (every op; may be NULL): [C] output, the largest magnitude per channel of the op's main output (h, da) over all
 * samples and groups of the call -- whole-operand bounds for the fp16-split convolutions (amax_per_sample = 0). */
long long pgan_bn_workspace_bytes(int C);
/* groups > 1: a / h hold `groups` consecutive forward passes of B samples each; every pass gets its own batch statistics
 * (stats: [groups][2][C]) and the running statistics step through the passes in order, as `groups` calls would (the
 * generator forwards of one iteration, src/train_wgan_v2.py:535,622, share their weights). */
int pgan_bn_fwd(const float* a, const float* gamma, const float* beta, float* rmean, float* rvar, float* h,
                float* stats, void* ws, long long ws_bytes, int B, int C, int HW, float momentum, int groups,
                float* amax, hipStream_t st);
int pgan_bn_bwd(const float* dh, const float* h, const float* a, const float* stats, const float* gamma, float* da,
                float* dgamma, float* dbeta, void* ws, long long ws_bytes, int B, int C, int HW, float* amax,
                hipStream_t st);
/* The same layers (src/layers.py:93-110,119-135) with batch statistics over the global batch of a data-parallel run
 * (one process per GPU, B samples each): *_sums writes pgan_bn_partials(B, C, HW) doubles of per-channel partial sums
 * (forward: sum a, sum a^2; backward: sum dy, sum dy xhat), the caller adds them over the ranks (all-reduce), *_apply
 * finishes from the totals with world = number of ranks.  dgamma / dbeta are 1/world of the all-rank totals, so that
 * the later sum of the parameter gradients over the ranks gives the totals.  amax: *_sums clears the array that the
 * following *_apply fills (pass the same pointer to both, or NULL to both). */
long long pgan_bn_partials(int B, int C, int HW);                     /* backward */
long long pgan_bn_fwd_partials(int B, int C, int HW, int groups);     /* forward */
int pgan_bn_fwd_sums(const float* a, double* part, int B, int C, int HW, int groups, float* amax, hipStream_t st);
int pgan_bn_fwd_apply(const float* a, const float* gamma, const float* beta, float* rmean, float* rvar, float* h,
                      float* stats, const double* part, int B, int C, int HW, int world, float momentum, int groups,
                      float* amax, hipStream_t st);
int pgan_bn_bwd_sums(const float* dh, const float* h, const float* a, const float* stats, double* part, int B, int C,
                     int HW, float* amax, hipStream_t st);
int pgan_bn_bwd_apply(const float* dh, const float* h, const float* a, const float* stats, const float* gamma, float* da,
                      float* dgamma, float* dbeta, const double* part, int B, int C, int HW, int world, float* amax,
                      hipStream_t st);

/* ---- linear head nn.Linear(11712, 1) over cat(2-D features, 1-D features), src/layers.py:482,491-493 */
int pgan_head_fwd(const float* h4, const float* f1, const float* w, const float* b, float* score, int N, int K2, int K1,
                  hipStream_t st);
int pgan_head_seed(const float* w, const float* coef, float* u4, int N, int K, hipStream_t st); /* u4[n] = coef[n] w */
int pgan_head_wgrad(const float* x, const float* coef, float* dw, int N, int J, hipStream_t st); /* dw = sum coef[n] x[n] */

/* ---- penalties.  pgan_interpolate: src/layers.py:38-41; pgan_gp_penalty: value of src/layers.py:54
 * (elementwise |g|, mean over all elements, times lambda) and v = d penalty / d g; pgan_div_penalty:
 * value of src/layers.py:69-77 (per-sample |g|^6 of real then fake, g: [2 nb][M]) and its v. */
int pgan_interpolate(const float* t, const float* real, const float* fake, float* out, int B, int M, hipStream_t st);
int pgan_gp_penalty(const float* g, float* v, float* pen, float* ws /* >= 256 floats */, long long n, float lam,
                    hipStream_t st);
int pgan_div_penalty(const float* g, float* v, float* pen, float* ws /* >= 2 nb floats */, int nb, int M,
                     hipStream_t st);
/* out = [W, mean real, mean fake, pen], W = mean(score[nb:2nb]) - mean(score[:nb]) + pen
 * (src/train_wgan_v2.py:584-587,594-597). */
int pgan_critic_loss(const float* score, const float* pen, float* out, int nb, hipStream_t st);
int pgan_sum_scale(const float* x, float* out, int n, float scale, hipStream_t st); /* out[0] = scale * sum x */

/* derivative of the generator's output activation through its output y (1 sigmoid, 2 tanh) */
int pgan_act_bwd(const float* dy, const float* y, float* dx, long long n, int act, hipStream_t st);
/* out = x * scale + shift: torchvision Normalize(0.5, 0.5) of src/train_wgan_v2.py:453,519-521 */
int pgan_affine(const float* x, float* out, long long n, float scale, float shift, hipStream_t st);
/* clears n <= 16 device buffers (16-byte aligned, byte counts multiples of 16) in one launch; no reference counterpart
 * (the outputs that pgan_conv_down_*_zeroed adds into, cleared together at the start of a step) */
int pgan_zero_multi(void* const* bufs, const long long* bytes, int n, hipStream_t st);

/* torch.optim.Adam step (src/train_wgan_v2.py:489-490,612,634) over one flat, 4-float-padded buffer.
 * step_dev != NULL: device-resident step counter, incremented then used (graph replay advances it).
 * grad_scale multiplies g as it is read: 1 / ranks when g holds the sum of a data-parallel run's gradients (the mean
 * of src/train_wgan_v2.py:584-585 over the global batch), 1 otherwise. */
int pgan_adam(float* p, const float* g, float* m, float* v, long long n, float lr, float b1, float b2, float eps,
              float grad_scale, int* step_dev, int step, hipStream_t st);

/* Philox4x32-10 draws replacing torch.randn (src/train_wgan_v2.py:534,621) and np.random.random
 * (src/layers.py:37).  state: 4 uint32 on the device (seed lo, hi, offset lo, hi), advanced by the call. */
int pgan_philox(float* out, long long n, unsigned int* state, int normal, hipStream_t st);
/* values [skip, skip + n) of a draw of `total` values; the state advances by the whole draw.  The ranks of a
 * data-parallel run that keep one generator state in step each take their rows of the global batch's draw this way
 * (rows [r B/P, (r+1) B/P) of z and t), so that P ranks draw what one device at the global batch would. */
int pgan_philox_slice(float* out, long long n, long long skip, long long total, unsigned int* state, int normal,
                      hipStream_t st);

/* Next B alignments (sites [.,1,64,64] = Ms floats, positions [.,1,64] = Mp floats) of a pool that is
 * resident in HBM, wrapping at npool; the cursor is a device int advanced by the call.  Replaces the
 * host loader's get_batch + .to(device) (src/train_wgan_v2.py:517,526-527) inside a captured step. */
int pgan_take_batch(const float* pool_s, const float* pool_p, float* out_s, float* out_p, int* cursor, int npool, int B,
                    int Ms, int Mp, hipStream_t st);
/* the B alignments from cursor + offset on, cursor advanced by `advance`: a rank's rows of the global batch */
int pgan_take_batch_slice(const float* pool_s, const float* pool_p, float* out_s, float* out_p, int* cursor, int npool,
                          int B, int Ms, int Mp, int offset, int advance, hipStream_t st);

/* ---- per-batch augmentations of the real alignments, drawn on the device inside the step (Philox state as
 * pgan_philox).  pgan_label_smooth: sites <- sites (1 - ey) + centre ey, ey ~ U(0, noise) per element
 * (DataGeneratorDisk.get_batch(label_smooth=True), src/train_wgan_v2.py:265-268: noise = centre = 0.05); normalized != 0:
 * sites hold (raw - 0.5) / 0.5 and the raw values are smoothed (the reference smooths before Normalize, :519-521).
 * ey_given: NULL, or the n draws to use (tests).  The n values are values [skip, skip + n) of a draw of `total` (a rank's
 * rows of the global batch); n and skip multiples of 4.
 * pgan_mono_switch: switch_monomorphic (src/train_wgan_v2.py:129-160, called per critic batch at :523-524 with bounds
 * 0, 10): per alignment n ~ U{lower..upper} distinct random columns, the first n / 2 set to 0.0 and the rest to 1.0 over
 * all rows, each one's position redrawn uniformly between its current neighbours (0 / 1 at the ends), in the order
 * drawn.  n_given [B], cols_given [B][upper], u_given [B][upper]: all NULL, or the draws to use (tests).  The B
 * alignments are numbers first .. first + B - 1 of a global batch of `total`.  upper <= 15, S <= 128. */
int pgan_label_smooth(float* sites, long long n, float noise, float centre, int normalized, unsigned int* rng,
                      const float* ey_given, long long skip, long long total, hipStream_t st);
int pgan_mono_switch(float* sites, float* pos, int B, int rows, int S, int lower, int upper, unsigned int* rng,
                     const int* n_given, const int* cols_given, const float* u_given, int first, int total,
                     hipStream_t st);

/* ---- discriminator 1-D tower, src/layers.py:470-480.  params: 12 device pointers
 * (conv w, conv b, gn w, gn b) x 3.  a1 [N][16][61], a2 [N][32][58], a3 [N][64][55] keep the conv
 * outputs for the backward; feat [N][3520]. */
int pgan_d1d_fwd(const float* xp, const float* const* params, float* a1, float* a2, float* a3, float* feat, int N,
                 hipStream_t st);
/* backward of sum_n coef[n] <w_tail, feat[n]>: per-sample parameter-gradient rows part[N][part_ld]
 * (tensor i at offs[i], may be NULL) and dxp [N][1][64] (may be NULL). */
int pgan_d1d_bwd(const float* xp, const float* const* params, const float* a1, const float* a2, const float* a3,
                 const float* coef, const float* w_tail, float* part, int part_ld, const int* offs, float* dxp, int N,
                 hipStream_t st);

/* The same tower on position vectors of length L = 64 or 128 (128: the scaled network for 128 x 128 alignments, which the
 * reference does not have -- its head is fixed to 64 x 64 inputs, src/layers.py:482): a1 [N][16][L-3], a2 [N][32][L-6],
 * a3 [N][64][L-9], feat [N][64 (L-9)], dxp [N][1][L]. */
int pgan_d1d_fwd_len(const float* xp, const float* const* params, float* a1, float* a2, float* a3, float* feat, int N,
                     int L, hipStream_t st);
int pgan_d1d_bwd_len(const float* xp, const float* const* params, const float* a1, const float* a2, const float* a3,
                     const float* coef, const float* w_tail, float* part, int part_ld, const int* offs, float* dxp,
                     int N, int L, hipStream_t st);

/* ---- generator 1-D tower stages nn.ConvTranspose1d(k4, s2, p1), src/layers.py:122-137: forward
 * (+ fused sigmoid/tanh), input gradient, weight gradient.  y [N][Cs][Ls], x [N][Cl][2 Ls], w [Cs][Cl][4]. */
int pgan_ct1d_up(const float* y, const float* w, float* x, int N, int Cs, int Cl, int Ls, int act, hipStream_t st);
int pgan_ct1d_down(const float* x, const float* w, float* y, int N, int Cs, int Cl, int Ls, hipStream_t st);
int pgan_ct1d_wgrad(const float* ya, const float* x, float* dw, int N, int Cs, int Cl, int Ls, hipStream_t st);

/* ---- C[M][N] = A[M][K] B[K][N] (transa 0) or A[K][M]^T B[K][N] (transa 1): the generator's first
 * ConvTranspose layers on a 1x1 / length-1 input (src/layers.py:92,117) and their weight gradients. */
int pgan_gemm(const float* A, const float* B, float* C, int M, int N, int K, int transa, hipStream_t st);

/* ---- input pipeline: the reference's alignment files (`<n>_sites.csv`, 64 lines of 64 comma-separated 0/1,
 * src/sim_conversion/convert_ms.py:39; `<n>_pos.csv`, 64 lines of one float, convert_ms.py:28-29) parsed natively into
 * bit-packed form (replaces the two np.loadtxt calls per alignment of DataGeneratorDisk.get_batch,
 * src/train_wgan_v2.py:254-255), and the device kernels between packed bits and the fp32 pool.
 * bits[n][512]: site j of individual i is bit j % 8 of byte i * 8 + j / 8.  pgan_parse_alignments runs on the host
 * (threads worker threads), fills status[k] (0 ok, 1 unreadable, 2 wrong number of values, 3 a site value other than
 * 0 / 1) and returns the number of failed pairs.  pgan_unpack_sites: sites[n][1][64][64] = scale * bit + shift (scale 2,
 * shift -1 is --normalize, src/train_wgan_v2.py:453,519-521).  pgan_pack_sites: bit = (x >= 0.5). */
int pgan_parse_alignments(const char* const* sites_paths, const char* const* pos_paths, int n, unsigned char* bits,
                          float* pos, int* status, int threads);
int pgan_unpack_sites(const unsigned char* bits, float* sites, long long n, float scale, float shift, hipStream_t st);
int pgan_pack_sites(const float* sites, unsigned char* bits, long long n, hipStream_t st);

/* Text output of Hudson's ms, as SimulatorGenerator.get_batch reads it from the subprocess (src/train_wgan_v2.py:342-344),
 * parsed on the host the way read_ms_data does (src/train_wgan_v2.py:311-330; the last replicate of the text is
 * dropped): sites[r][rows][cols], pos[r][cols].  Returns the replicates written (<= max_reps), -1 bad arguments,
 * -3 a replicate that is not rows x cols, -4 a genotype character that is not a digit. */
int pgan_parse_ms(const char* text, long long len, int rows, int cols, int max_reps, float* sites, float* pos);

/* ---- device-side sampling around the step (Philox state as pgan_philox).
 * pgan_fakebuf_step: FakeBuffer.add then FakeBuffer.sample (src/train_wgan_v2.py:98-117) on a ring of `size` (<= 16320)
 * alignments resident in HBM: appends the B new alignments (oldest overwritten), then writes B alignments drawn without
 * replacement to out_s / out_p.  state: 4 device ints (head, count, error flag, spare), zeroed by the caller once;
 * slots: B device ints of scratch; picks: NULL, or B device ints with the logical indices to take (0 = oldest) in
 * place of random draws (state[2] is set if one lies outside the buffer).
 * pgan_sample_genotypes: sample_genotypes_bernoulli (src/train_wgan_v2.py:162-192) on p[n][1][rows][S], rows = 64 | 128. */
int pgan_fakebuf_step(float* buf_s, float* buf_p, int size, int* state, int* slots, const float* new_s,
                      const float* new_p, float* out_s, float* out_p, int B, int Ms, int Mp, unsigned int* rng,
                      const int* picks, hipStream_t st);
int pgan_sample_genotypes(const float* p, float* out, long long n, int rows, int S, unsigned int* rng,
                          int require_variation, hipStream_t st);

/* ---- one step as a hipGraph (the reference launches its step op by op from Python, src/train_wgan_v2.py:534-634):
 * begin capture on st, enqueue the step once through the entry points above (side streams join st through events),
 * end -> executable graph; launch replays it on st; destroy frees it.  No kernel runs between begin and end. */
int pgan_step_graph_begin(hipStream_t st);
int pgan_step_graph_end(hipStream_t st, void** graph_exec);
int pgan_step_graph_launch(void* graph_exec, hipStream_t st);
int pgan_step_graph_destroy(void* graph_exec);

/* "gfx950 packed-fp32-ops=off" for a library whose device code was built without packed-fp32 vector instructions (the
 * only supported build: see csrc/runtime.hip), "...=on" otherwise.  No counterpart in the reference. */
const char* pgan_build_info(void);

/* ---- cross-stream ordering of a step's side streams (the reference runs on one stream; no counterpart).  Events
 * are created once, before anything is recorded, and reused: pgan_stream_edge(ev, from, to) makes work enqueued on `to`
 * afterwards wait for what `from` holds now (record + wait; under an open capture it becomes a graph edge);
 * pgan_event_record / pgan_stream_wait_event are its two halves, for a mark that is waited for later.  No call here
 * creates or destroys anything except _create / _destroy; pgan_events_created() counts the former, so a caller can
 * assert that none happened while a capture was open. */
int pgan_event_create(void** ev);
int pgan_event_destroy(void* ev);
long long pgan_events_created(void);
int pgan_event_record(void* ev, hipStream_t st);
int pgan_stream_wait_event(hipStream_t st, void* ev);
int pgan_stream_edge(void* ev, hipStream_t from, hipStream_t to);

/* ---- launch tape: one iteration's enqueue calls (the loop body of src/train_wgan_v2.py:516-634, which the reference
 * drives op by op from Python) noted once and replayed by one call.  pgan_tape_append notes a call of the named entry
 * point of this header (enqueue entry points only) with one 64-bit word per argument in declaration order: integers
 * sign-extended, pointers as addresses, floats as their 32 bits in the low half; host arrays an argument points to
 * must stay alive as long as the tape.  pgan_tape_replay makes the calls in order; it stops at the first one that does
 * not return 0, stores its index in *failed_at (may be NULL) and returns that call's code.  pgan_tape_arg_count: words
 * the named entry point takes, -1 if it cannot be noted. */
void* pgan_tape_create(void);
int pgan_tape_destroy(void* tape);
int pgan_tape_arg_count(const char* entry_point);
int pgan_tape_append(void* tape, const char* entry_point, const unsigned long long* words, int nwords);
int pgan_tape_length(void* tape);
int pgan_tape_replay(void* tape, int* failed_at);
const char* pgan_tape_entry_name(void* tape, int index);

#ifdef __cplusplus
}
#endif
#endif /* PGAN_HIP_H */
