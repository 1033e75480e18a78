/* libfavae_hip.so -- C ABI of the MI355X (gfx950) implementation of the FA-VAE training hot path.
 *
 * The reference (aguerrerolopez/FA-VAE) is pure Python; every entry point below replaces a NumPy expression or a
 * torch/ATen call on the path  SSHIBA.update() -> {update_w, update_Z, ImgVAE.trainloop, update_x, update_tau,
 * update_b, update_alpha}.  The citation after each declaration is the reference file:line it stands in for
 * (cA = condition_a_VAE, DA = domain_adaptation).  A reference-side binding is a ctypes stub.
 *
 * Conventions
 *   - all pointers are DEVICE pointers unless the name ends in _host; sizes are element counts;
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); calls are asynchronous w.r.t. the device
 *     unless stated otherwise;
 *   - return value: 0 ok; >0 numerical flag (FAVAE_NOT_SPD: the reference's "print and skip the update" case,
 *     cA/favae.py:917-922); <0 argument / HIP error, text available from favae_last_error();
 *   - activations are NHWC, conv weights are [Cout][kh*kw][Cin]; dtype selects the storage/MFMA type of GEMM
 *     operands (FAVAE_F32: v_mfma_f32_32x32x2_f32, FAVAE_BF16: v_mfma_f32_32x32x16_bf16); accumulation is fp32;
 *   - the factor-analysis half is fp64 throughout (the reference does it in NumPy float64).
 */
#ifndef FAVAE_HIP_H
#define FAVAE_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FAVAE_F32 0
#define FAVAE_BF16 1
/* Storage flags, OR-ed into a `dtype` argument together with FAVAE_BF16 (the fp32 parity path never uses them): the
 * conv output / BN input tensor `raw`, or the activation gradient `dact`, is stored as bf16 instead of fp32.  Used for
 * the 64-channel layers of 4-17 M elements per batch, where every fp32 pass over the tensor is HBM time. */
#define FAVAE_RAW_BF16 0x10
#define FAVAE_DACT_BF16 0x20
#define FAVAE_OUT_BF16 0x40
/* favae_bn_lrelu_bwd_bins: dbias is not written; `scratch` keeps the column sums of draw per slab of rows,
 * [favae_bn_lrelu_bwd_slabs(R, C)][C] doubles, and favae_bn_dbias_finish adds them up later, on any stream that is ordered
 * behind the launch.  The launch then ends with its last row: no wait for its stores, no ticket, no block that reads every
 * partial.  (The bias of a convolution in front of a BatchNorm has a gradient of rounding noise, and nothing on the main
 * chain of a training step waits for it.) */
#define FAVAE_DBIAS_DEFER 0x80

const char* favae_last_error(void);
int favae_version(void);

/* ---- conv VAE: GEMM-shaped kernels ------------------------------------------------------------------------- */

/* Implicit-GEMM convolution, gather form (scatter=0) or transposed/scatter form (scatter=1).
 *   scatter=0: nn.Conv2d forward (cA/celeba_deepvanilla_vae.py:31-33, 74-76), the input gradient of
 *              nn.ConvTranspose2d and nn.Linear forward / input gradient (:41-45) with ks=1 on a 1x1 grid.
 *   scatter=1: nn.ConvTranspose2d forward (:50-57, 64-71) and the input gradient of nn.Conv2d (loss.backward() :249).
 * in [Nb,Hi,Wi,Cin] dtype, w [Cout][ks*ks][Cin] dtype, bias [Cout] fp32 or NULL, out [Nb,Ho,Wo,Cout] fp32.
 * splitk: 0 = choose; >1 splits the reduction and accumulates with fp32 atomics (out is zeroed first); <0 = choose, and
 * the caller guarantees that out is already all zero (lets one memset serve many launches). */
int favae_conv_igemm(const void* in, const void* w, const float* bias, float* out, int Nb, int Hi, int Wi, int Cin,
                     int Ho, int Wo, int Cout, int ks, int stride, int pad, int scatter, int splitk, int dtype,
                     void* stream);

/* favae_conv_igemm with per-channel sums of the result taken in the GEMM epilogue, so that the BatchNorm layer that
 * follows (forward, :34,58,72) or precedes (backward, autograd of :34-35 under :249) needs no reduction pass of its own:
 *   mode 1: bins[b][0][c] += sum y, bins[b][1][c] += sum y^2 over the pixels of a tile (nn.BatchNorm2d batch statistics);
 *   mode 2: y is the gradient w.r.t. the output `act` of a BN+LeakyReLU whose input was `raw` (same [pixels][Cout]
 *           shape as out): bins[b][0][c] += sum dd, bins[b][1][c] += sum dd * (raw - mean) * invstd, dd = y * lrelu'(act).
 * bins: [nbins][2][Cout] doubles, zeroed by the caller; tiles spread over the bins (fp64 atomics), the consumer adds
 * them up (favae_bn_train_fwd_bins / favae_bn_lrelu_bwd_bins).  *fused = 1 if the launch took the sums, 0 if this shape
 * runs with a split reduction or a tile without that epilogue (then bins is untouched and the caller reduces itself).
 * dtype may carry FAVAE_OUT_BF16 (out stored as bf16; the sums are taken from the fp32 accumulators) and, for mode 2,
 * FAVAE_RAW_BF16 (raw is stored as bf16). */
int favae_conv_igemm_red(const void* in, const void* w, const float* bias, float* out, int Nb, int Hi, int Wi, int Cin,
                         int Ho, int Wo, int Cout, int ks, int stride, int pad, int scatter, int splitk, int dtype,
                         double* bins, int nbins, int mode, const void* act, const void* raw, const float* mean,
                         const float* invstd, int* fused, void* stream);

/* Conv -> BatchNorm(train) -> LeakyReLU -> Conv chains of encode() / decode() (:80-87, 94-99; blocks :25-42, 45-78) without a
 * launch between the convolutions.  favae_conv_igemm with two optional fusions:
 *   output side (ab != NULL; needs bins): as favae_conv_igemm_red mode 1 the epilogue adds sum / sum of squares per channel
 *     into bins [nbins][2][Cout] (zeroed by the caller); the block that finishes last turns them into the BatchNorm(train)
 *     quantities of the layer that follows this convolution: mean, invstd [Cout], ab [2*Cout] = (a | b) with
 *     a = invstd * gamma, b = beta - mean * a, and the running statistics (run_mean / run_var, may be NULL; unbiased variance,
 *     `momentum`).  *fused = 1 if the launch did so; 0 if this shape runs with a split reduction (nothing was written: the
 *     caller applies favae_bn_train_fwd to out instead).
 *   input side (in_ab != NULL): `in` is the fp32 output [Nb,Hi,Wi,Cin] of the preceding convolution BEFORE its BatchNorm +
 *     LeakyReLU and in_ab the (a | b) its launch published; the activation lrelu(in * a[c] + b[c]), rounded to dtype, is formed
 *     when a chunk of the A operand goes to LDS.  Cin in {64, 128, 192, 256}, Cout > 32.  The materialised activation, where the
 *     backward pass wants it, is favae_bn_lrelu_fwd_ab(in, in_ab, ...): same bytes. */
/* favae_conv_wgrad whose epilogue takes the Adam step of the layer's weights (torch.optim.Adam, cA/celeba_deepvanilla_vae.py:131,
 * applied by optimizer.step() :250; the arithmetic of favae_adam_step) and writes their dtype copy wc [Rc][ks*ks][Cc] and transpose
 * wt [Cc][ks*ks][Rc] (either may be NULL) where the launch runs unsplit in 64x64 tiles (Rc, Cc multiples of 64): *applied = 1, dw is
 * not touched, p / m / v [Rc][ks*ks][Cc] are updated in place.  Otherwise *applied = 0 and the call is favae_conv_wgrad into dw. */
int favae_conv_wgrad_adam(const void* rows, int Rc, int rows_large, const void* cols, int Cc, int cols_large, float* dw, float* p,
                          float* m, float* v, void* wc, void* wt, int Nb, int Hs, int Ws, int Hl, int Wl, int ks, int stride, int pad,
                          int splitk, float lr, float beta1, float beta2, float eps, long long step, int dtype, int* applied,
                          void* stream);

/* Grouped-batch mode (no counterpart launch in the reference; SURVEY.md 7 "Hard parts": a larger batch per step with the BatchNorm
 * group size kept at the reference's 64, celeba_deepvanilla_vae.py:218 with nn.BatchNorm2d :34,58,72).  A batch of G x 64 images
 * whose BatchNorm statistics are taken per group of 64: with the mean loss this is the G-rank data-parallel step on
 * one GPU.  favae_conv_igemm_red_groups = favae_conv_igemm_red with the sums of group g at bins + g * nbins * 2 * Cout (mode 2 reads
 * mean / invstd [G][Cout]); favae_gbn_sums fills such bins for a tensor whose producer could not; favae_gbn_fwd / favae_gbn_bwd are
 * BatchNorm(train) + LeakyReLU forward / backward of the G groups (running statistics: average over the groups of the batch
 * statistics; dgamma / dbeta / dbias are ADDED to).  Tensors [G * Rg][C] NHWC, C <= 1024 dividing 1024; raw / dact fp32, or bf16
 * with FAVAE_RAW_BF16 / FAVAE_DACT_BF16 in dtype (the conv launch: FAVAE_OUT_BF16 for its output, FAVAE_RAW_BF16 for the raw it reads).
 * ab (may be NULL) [G][2][C]: the affine form act = lrelu(raw * a + b) of the forward pass, written by favae_gbn_fwd; the backward
 * launches that are given it take the LeakyReLU slope from the sign of raw * a + b and do not read act (2 bytes per element less). */
int favae_conv_igemm_red_groups(const void* in, const void* w, const float* bias, float* out, int Nb, int Hi, int Wi, int Cin, int Ho,
                                int Wo, int Cout, int ks, int stride, int pad, int scatter, int splitk, int dtype, double* bins,
                                int nbins, int mode, const void* act, const void* raw, const float* mean, const float* invstd,
                                const float* ab, int group_images, int* fused, void* stream);
int favae_gbn_sums(const void* y, const void* act, const void* raw, const float* mean, const float* invstd, const float* ab, double* bins,
                   int nbins, int G, long long Rg, int C, int mode, int dtype, void* stream);
int favae_gbn_fwd(const void* raw, const double* bins, int nbins, const float* gamma, const float* beta, void* act, float* mean,
                  float* invstd, float* ab, float* run_mean, float* run_var, int G, long long Rg, int C, float eps, float momentum,
                  int dtype, void* stream);
/* The running statistics of favae_gbn_fwd alone (favae_gbn_fwd with run_mean == NULL leaves them out): running_mean / running_var of
 * nn.BatchNorm2d(train) (celeba_deepvanilla_vae.py:34,58,72) from mean / invstd [G][C], averaged over the groups.  No launch of a training
 * step reads them: the executor queues this one beside the main chain. */
int favae_gbn_running(const float* mean, const float* invstd, int G, long long Rg, int C, float eps, float momentum, float* run_mean,
                      float* run_var, void* stream);
long long favae_vae_epoch_steps(long long n, int batch);
int favae_gbn_bwd(const void* dact, const void* act, const void* raw, const float* mean, const float* invstd, const float* gamma,
                  const float* ab, const double* bins, int nbins, void* draw, float* dgamma, float* dbeta, float* dbias, int G, long long Rg, int C,
                  int dtype, void* stream);

int favae_conv_igemm_bn(const void* in, const float* in_ab, const void* w, const float* bias, float* out, int Nb, int Hi, int Wi,
                        int Cin, int Ho, int Wo, int Cout, int ks, int stride, int pad, int scatter, int splitk, int dtype,
                        double* bins, int nbins, const float* gamma, const float* beta, float eps, float momentum, float* mean,
                        float* invstd, float* ab, float* run_mean, float* run_var, int* fused, void* stream);

/* favae_conv_igemm followed by eval-mode BatchNorm + LeakyReLU in the epilogue (self.eval() passes: update_x :343-355,
 * reconstruction :368-381): out (dtype with FAVAE_OUT_BF16, else fp32) = lrelu((conv + bias) * a[c] + b[c]),
 * ab [2*Cout] = (a | b) from favae_bn_eval_ab.  Same values as favae_conv_igemm + favae_bn_lrelu_fwd.  Unsplit reduction,
 * Cout > 32 and a multiple of 4. */
int favae_conv_igemm_affine(const void* in, const void* w, const float* bias, void* out, int Nb, int Hi, int Wi, int Cin, int Ho,
                            int Wo, int Cout, int ks, int stride, int pad, int scatter, int dtype, const float* ab, void* stream);
/* a = gamma / sqrt(running_var + eps), b = beta - running_mean * a -> ab [2*C]. */
int favae_bn_eval_ab(const float* run_mean, const float* run_var, const float* gamma, const float* beta, int C, float eps, float* ab,
                     void* stream);

/* Weight gradient  dw[r][tap][c] += sum_pixels rows[pix][r] * cols[pix'][c]  (fp32 atomics; caller zeroes dw).
 * splitk: 0 = choose, > 0 = that many slices of the pixel range, < 0 = choose and dw is known to be all zero (an unsplit
 * reduction then stores its result instead of reading the zeros back).
 * Exactly one of the two operands may be the large-spatial tensor (Hl x Wl), read at (s*h-p+kh, s*w-p+kw);
 * the other lives on the small grid (Hs x Ws).  Conv2d: rows = dout (small), cols = input (large).
 * ConvTranspose2d: rows = dout (large), cols = input (small).  Linear: 1x1 grids, ks=1.
 * Replaces the weight half of loss.backward(), cA/celeba_deepvanilla_vae.py:249. */
int favae_conv_wgrad(const void* rows, int Rc, int rows_large, const void* cols, int Cc, int cols_large, float* dw,
                     int Nb, int Hs, int Ws, int Hl, int Wl, int ks, int stride, int pad, int splitk, int dtype,
                     void* stream);

/* The same weight gradient (ks 3, pad 1) for the two layers with an image-sized channel count on one side, as ONE GEMM
 * over the pixels: the 9 taps x <= 3 channels of the narrow operand are gathered into a [pixels][32] matrix in `scratch`
 * first, so the wide operand is read once instead of under 9 shifted windows.
 *   Cc <= 3 (encoder.0, :31-33): rows = dout [Nb,Hs,Ws,Rc], cols = the layer input [Nb,Hl,Wl,Cc], stride 1 or 2;
 *   Rc <= 3, stride 1 (final_layer.3, :74-76): rows = dout [Nb,Hs,Ws,Rc], cols = the layer input [Nb,Hl,Wl,Cc], Hs = Hl.
 * dw [Rc][9][Cc] += ... (fp32 atomics, caller zeroes dw); scratch >= favae_conv_wgrad_folded_scratch(Nb, Hl, Wl, dtype). */
size_t favae_conv_wgrad_folded_scratch(int Nb, int Hl, int Wl, int dtype);
int favae_conv_wgrad_folded(const void* rows, int Rc, const void* cols, int Cc, float* dw, int Nb, int Hs, int Ws, int Hl,
                            int Wl, int stride, int splitk, int dtype, void* scratch, size_t scratch_bytes, void* stream);

/* The two halves of favae_conv_wgrad_folded on their own, so that the gather of an operand that does not depend on the
 * backward pass (the input batch of encoder.0) can run long before its gradient GEMM:
 *   fold_taps: dst [Nb*Hs*Ws][32] <- the 9 taps x C (<= 3) channels of src; shifted = 0: src on the large grid
 *     [Nb,Hl,Wl,C] read at (s*h-1+kh, s*w-1+kw), column tap*C+c; shifted = 1 (stride 1, Hs = Hl): src read at
 *     (h+1-kh, w+1-kw), column c*9+tap.  Columns past 9*C are zero.
 *   wgrad_prefolded: dw [Rst][Cst] += sum_k rows[k][r] * cols[k][c] over K pixel rows of two plain matrices
 *     ([K][Rc], [K][Cc]; Rst <= Rc, Cst <= Cc: the padding columns are not stored).  splitk: 0 auto, > 0 that many
 *     slices of the reduction, -1 auto with dw known to be zero, < -1 that many slices with dw known to be zero. */
/* Forward convolution over the tap-folded input (no counterpart launch in the reference: it is Conv2d(3 -> 64, k3 s2 p1) of
 * celeba_deepvanilla_vae.py:31-33 computed as one K <= 32 GEMM): out [P][Cout] fp32 = fold [P][32] x w [Cout][K]^T + bias, with
 * fold from favae_conv_fold_taps(shifted = 0) and w the layer's [Cout][9][C] weights (K = 9 * C).  bf16 operands, Cout = 64.
 * bins != NULL: per-channel sum / sum of squares of the output (fp64) are added to bins [nbins][2][Cout], the input of
 * favae_bn_train_fwd_bins (the caller zeroes them). */
int favae_conv_folded_fwd(const void* fold, const void* w, const float* bias, float* out, long long P, int K, int Cout, double* bins,
                          int nbins, int dtype, void* stream);

int favae_conv_fold_taps(const void* src, void* dst, size_t dst_bytes, int Nb, int Hs, int Ws, int Hl, int Wl, int C,
                         int stride, int shifted, int dtype, void* stream);
int favae_conv_wgrad_prefolded(const void* rows, int Rc, int Rst, const void* cols, int Cc, int Cst, float* dw, long long K,
                               int splitk, int dtype, void* stream);
/* favae_conv_wgrad_prefolded whose column operand is the conv output cols_raw [K][Cc] (bf16) in front of a BatchNorm2d(train) + LeakyReLU
 * (final_layer.1/.2 in front of final_layer.3, celeba_deepvanilla_vae.py:72-76; the gradient loss.backward() :249 takes): the activation
 * lrelu(x * a + b), rounded to bf16, is formed at the operand load from ab [groups][2][Cc] (a | b per BatchNorm group of group_rows rows of
 * cols_raw; 0: one group).  bf16, Rc <= 32, K and group_rows multiples of 64. */
int favae_conv_wgrad_prefolded_bn(const void* rows, int Rc, int Rst, const void* cols_raw, int Cc, int Cst, const float* ab,
                                  long long group_rows, float* dw, long long K, int splitk, int dtype, void* stream);

/* The same pair with (almost) fp32 operands on bf16 matrix cores, for encoder.0 of the bf16 training step (the image is the one
 * operand whose bf16 rounding passes through every BatchNorm of the encoder; celeba_deepvanilla_vae.py:31-33, 80-87):
 *   fold_taps_split: fp32 src [Nb,Hl,Wl,C] -> two bf16 planes of [Nb*Hs*Ws][32]: dst = bf16(x) (exactly what
 *     favae_conv_fold_taps(shifted = 0, bf16) writes: the weight gradient keeps reading it), dst + Nb*Hs*Ws*32 = bf16(x - bf16(x));
 *     dst_bytes >= Nb*Hs*Ws*128.
 *   folded_fwd_split: out = x_hi w_hi + x_lo w_hi + x_hi w_lo + bias with w the fp32 master weights [Cout][K], split in the
 *     kernel; fold_hl the two planes above; bins as in favae_conv_folded_fwd. */
int favae_conv_fold_taps_split(const float* src, void* dst, size_t dst_bytes, int Nb, int Hs, int Ws, int Hl, int Wl, int C,
                               int stride, void* stream);
int favae_conv_folded_fwd_split(const void* fold_hl, const float* w, const float* bias, float* out, long long P, int K, int Cout,
                                double* bins, int nbins, void* stream);

/* Input gradient of final_layer.3 = Conv2d(64 -> C <= 3, k3 s1 p1) (loss.backward(), cA/celeba_deepvanilla_vae.py:249 through :74-76)
 * over the tap-folded output gradient that the layer's weight gradient uses as well (favae_conv_fold_taps with shifted = 1:
 * fold [P][32], column c * 9 + tap): out [P][64] = fold x w, w = the layer's weight [C][9][64] in dtype bf16, K = 9 * C.  out is fp32,
 * or bf16 with FAVAE_OUT_BF16.  bins != NULL: the BatchNorm-backward sums of the layer in front (favae_conv_igemm_red mode 2) for
 * `groups` BatchNorm groups of P / groups pixels: bins [groups][nbins][2][64], mean / invstd [groups][64], act bf16, raw fp32 or
 * bf16 (FAVAE_RAW_BF16); ab (may be NULL) [groups][2][64] as in favae_gbn_fwd: the slope from raw * a + b, act is not read. */
int favae_conv_folded_dgrad(const void* fold, const void* w, void* out, long long P, int K, int Cin, double* bins, int nbins,
                            const void* act, const void* raw, const float* mean, const float* invstd, const float* ab, int groups,
                            int dtype, void* stream);
/* The input gradient of favae_conv_folded_dgrad formed again and the backward of the BatchNorm2d(train) + LeakyReLU in front of the layer
 * applied to it at once (final_layer.1/.2, celeba_deepvanilla_vae.py:72-73, under loss.backward() :249), so that the gradient tensor is neither
 * written nor read back: first favae_conv_folded_dgrad with out == NULL (the two sums into bins only), then this call: draw [P][64] (bf16) =
 * gamma invstd (dd - s1 / R - xhat s2 / R) per BatchNorm group (R = P / groups), s1 / s2 the totals of bins [groups][nbins][2][64];
 * dgamma / dbeta are ADDED to; dbias_partial (may be NULL) [*slabs][64] doubles takes the column sums of draw per block, at most slabs_cap
 * of them, for favae_bn_dbias_finish / _ride.  FAVAE_OUT_BF16 in dtype: the gradient is rounded to bf16 before use, as a stored bf16
 * gradient would be (draw then equals that of favae_conv_folded_dgrad + favae_bn_lrelu_bwd_bins_ab / favae_gbn_bwd up to the odd last bf16 bit). */
int favae_conv_folded_dgrad_bn(const void* fold, const void* w, void* draw, long long P, int K, int Cin, const double* bins, int nbins,
                               const void* act, const void* raw, const float* mean, const float* invstd, const float* gamma, const float* ab,
                               int groups, float* dgamma, float* dbeta, double* dbias_partial, int slabs_cap, int* slabs, int dtype, void* stream);

/* Live HIP-event timing of selected launches (measurement only; no reference counterpart). Every stride-th launch of each
 * kernel class is timed on its launch stream; collect() synchronises and returns [16][3] = launches, seconds, algorithmic
 * work per class: flops for the GEMM classes, bytes for the HBM-bound ones (7 BatchNorm-backward apply, 13 Adam); classes
 * are listed in csrc/api.hip.  Choose a stride coprime to the launches per step of the classes of interest. */
int favae_prof_enable(int enabled, int stride);
int favae_prof_collect(double* out48);
/* Developer hook for tile tuning (tools/tune_gemm.py; no reference counterpart): force the tile of the gather-form
 * kernels (0 = 64x64, 1 = 128x32) and of the weight-gradient kernels (0..2 = 32x128, 128x32, 64x64; 9 = the all-taps kernel
 * wherever its shape conditions hold, whatever the size of the reduction); -1 restores the built-in choice.  Process-wide, not
 * thread-safe. */
int favae_debug_force_tile(int kc_tile, int pm_tile);
/* Test hook (no counterpart in the reference): n floats through the library's fp32 -> bf16 conversion (hw: gfx950's conversion
 * instruction, what every kernel uses) and through the integer round-to-nearest-even form (sw). */
int favae_debug_bf16_convert(const float* x, void* hw, void* sw, long long n, void* stream);

/* ---- conv VAE: HBM-bound kernels ----------------------------------------------------------------------------- */

/* nn.BatchNorm2d training statistics over the rows of raw [R][C] (cA/celeba_deepvanilla_vae.py:34,58,72): batch mean
 * and 1/sqrt(biased var + eps) -> mean, invstd; running stats updated with `momentum` (unbiased var) when non-NULL.
 * scratch: fp64 partial sums, at least 2*1024*64 doubles. */
int favae_bn_stats(const float* raw, long long R, int C, float eps, float momentum, float* mean, float* invstd,
                   float* run_mean, float* run_var, double* scratch, size_t scratch_doubles, void* stream);
/* eval mode (self.eval(), :343,368): mean/invstd from the running statistics. */
int favae_bn_eval_prep(const float* run_mean, const float* run_var, int C, float eps, float* mean, float* invstd,
                       void* stream);
/* act = LeakyReLU_0.01((raw-mean)*invstd*gamma+beta) stored as dtype (:34-35); apply_act=0 skips the LeakyReLU. */
int favae_bn_lrelu_fwd(const void* raw, const float* mean, const float* invstd, const float* gamma, const float* beta,
                       void* act, long long R, int C, int apply_act, int dtype, void* stream);
/* BatchNorm2d (training) + LeakyReLU of a small feature map (R <= 1024 rows; encoder.4, :34-35) in one channel-owner launch that
 * writes the activation twice: act (bf16, what the backward pass and the weight gradient read) and act_f32 (unrounded), which
 * fc_mu | fc_var of the bf16 step read (:82-85), so the last rounding in front of the latent means is fp32's.  _ok: shape test. */
int favae_bn_train_fwd_keep32(const float* raw, const float* gamma, const float* beta, void* act, float* act_f32, float* mean,
                              float* invstd, float* run_mean, float* run_var, long long R, int C, float eps, float momentum,
                              int dtype, void* stream);
int favae_bn_train_fwd_keep32_ok(long long R, int C);
/* act = LeakyReLU_0.01(raw * a[c] + b[c]) with ab [2*C] = (a | b) as published by favae_conv_igemm_bn: the activation that a consuming
 * favae_conv_igemm_bn forms on the fly, materialised (bit-identical) for the backward pass (:34-35 under loss.backward(), :249). */
int favae_bn_lrelu_fwd_ab(const float* raw, const float* ab, void* act, long long R, int C, int dtype, void* stream);
/* Training-mode BN + LeakyReLU in one call (statistics, running-stat update, apply; :34-35 under self.train()): same
 * results as favae_bn_stats followed by favae_bn_lrelu_fwd; small layers (R <= 1024) run as one launch. */
int favae_bn_train_fwd(const void* raw, const float* gamma, const float* beta, void* act, float* mean, float* invstd,
                       float* run_mean, float* run_var, long long R, int C, float eps, float momentum, double* scratch,
                       size_t scratch_doubles, int dtype, void* stream);
/* favae_bn_train_fwd with the statistics already summed by favae_conv_igemm_red (mode 1) into bins [nbins][2][C]; raw fp32,
 * or bf16 with FAVAE_RAW_BF16 in dtype. */
int favae_bn_train_fwd_bins(const void* raw, const double* bins, int nbins, const float* gamma, const float* beta,
                            void* act, float* mean, float* invstd, float* run_mean, float* run_var, long long R, int C,
                            float eps, float momentum, int dtype, void* stream);
/* backward of BN(train)+LeakyReLU (autograd of :34-35 under loss.backward() :249): draw (dtype) = d loss / d raw,
 * dgamma, dbeta, and dbias = column sums of draw (gradient of the preceding conv bias; NULL to skip).
 * s1s2: 2*C floats of scratch. */
int favae_bn_lrelu_bwd(const void* dact, const void* act, const void* raw, const float* mean, const float* invstd,
                       const float* gamma, void* draw, float* dgamma, float* dbeta, float* dbias, long long R, int C,
                       double* scratch, size_t scratch_doubles, float* s1s2, int dtype, void* stream);
/* favae_bn_lrelu_bwd with its two reductions already summed by favae_conv_igemm_red (mode 2) into bins [nbins][2][C]:
 * one pass (draw, dbias) instead of two; dact / raw fp32, or bf16 with FAVAE_DACT_BF16 / FAVAE_RAW_BF16 in dtype. */
int favae_bn_lrelu_bwd_bins(const void* dact, const void* act, const void* raw, const float* mean, const float* invstd,
                            const float* gamma, const double* bins, int nbins, void* draw, float* dgamma, float* dbeta,
                            float* dbias, long long R, int C, double* scratch, size_t scratch_doubles, int dtype,
                            void* stream);
/* The pair that does without the stored activation in the backward pass: favae_bn_train_fwd_bins_ab also publishes ab [2 C] = (a | b),
 * a = invstd * gamma, b = beta - mean * a (act = lrelu(raw * a + b)); favae_bn_lrelu_bwd_bins_ab takes the LeakyReLU slope from the sign
 * of raw * a + b, the forward pass's own expression, instead of reading act (nn.LeakyReLU backward, cA/celeba_deepvanilla_vae.py:30,
 * under loss.backward() :249).  Otherwise as favae_bn_train_fwd_bins / favae_bn_lrelu_bwd_bins. */
int favae_bn_train_fwd_bins_ab(const void* raw, const double* bins, int nbins, const float* gamma, const float* beta, void* act, float* mean,
                               float* invstd, float* ab, float* run_mean, float* run_var, long long R, int C, float eps, float momentum,
                               int dtype, void* stream);
int favae_bn_lrelu_bwd_bins_ab(const void* dact, const void* raw, const float* mean, const float* invstd, const float* gamma, const float* ab,
                               const double* bins, int nbins, void* draw, float* dgamma, float* dbeta, float* dbias, long long R, int C,
                               double* scratch, size_t scratch_doubles, int dtype, void* stream);
/* the two halves of FAVAE_DBIAS_DEFER: the slab count of favae_bn_lrelu_bwd_bins for (R, C) (host arithmetic only), and
 * dbias[c] = sum over the slabs, in slab order, of partial[slab][c]. */
int favae_bn_lrelu_bwd_slabs(long long R, int C);
int favae_bn_dbias_finish(const double* partial, int nslab, int C, float* dbias, void* stream);
/* favae_bn_dbias_finish without a launch of its own: the calling thread's NEXT favae_conv_wgrad launch does the sums in its
 * first blocks, ahead of their own work (nslab <= 256).  What the training step uses: the weight gradient of the same layer
 * is queued next, on the stream beside the main chain, and a separate launch in front of it costs as much as the deferral
 * saves. */
int favae_bn_dbias_ride(const double* partial, int nslab, int C, float* dbias);
/* out[c] = sum_r x[r][c]  (bias gradients of layers without BN). */
int favae_colsum(const void* x, long long R, int C, float* out, double* scratch, size_t scratch_doubles, int dtype,
                 void* stream);
/* favae_colsum of an fp32 matrix x that is stored as dtype into y on the way (y = x when dtype is fp32). */
int favae_colsum_cast(const float* x, void* y, long long R, int C, float* out, double* scratch, size_t scratch_doubles,
                      int dtype, void* stream);
/* Fused reparameterisation + KL against the FA prior (:86-92 encoder head, :89-92 reparameterize, :137-150 kl_div,
 * :231-232 prior std).  fc [B][2z] = (mu | log_var_raw); eps [B][z]; prior [N][z] fp64 rows selected by idx[b]
 * (NULL prior = N(0,I), :213-214; NULL idx = identity); tau as the fp32 the reference casts it to (:210).
 * Outputs zlat [B][z] (dtype), stdq [B][z], *kl_out = mean_b sum_z KL (fp64). */
int favae_reparam_kl_fwd(const float* fc, const float* eps, const double* prior, const int* idx, float tau, int B, int z,
                         void* zlat, float* stdq, double* kl_out, int dtype, void* stream);
/* favae_reparam_kl_fwd followed by decoder_input (nn.Linear z -> N, :41-45, the first layer of decode(), :100-104) in one
 * launch: out [B][N] (dtype) = zlat (as stored, i.e. rounded to dtype) @ w^T + bias, w [N][z] dtype, bias [N] fp32.
 * z <= 128 (two instances: z <= 32 and z <= 128), B*z <= 12288.  Same results as the two calls up to the summation order of
 * the z products. */
int favae_reparam_kl_decoder_input(const float* fc, const float* eps, const double* prior, const int* idx, float tau, int B,
                                   int z, void* zlat, float* stdq, double* kl_out, const void* w, const float* bias, void* out,
                                   int N, int dtype, void* stream);
/* Backward of decoder_input and of the reparameterisation in one launch (autograd of :41-45 and :89-92/:137-150 under
 * loss.backward(), :249): given dact [B][N] fp32 (gradient w.r.t. the decoder_input output) and w [N][z] (dtype):
 * draw [B][N] = dact as dtype, dbias_di [N] = column sums of dact, dz [B][z] += draw @ w (fp32 atomics, dz ZERO on entry),
 * then, as favae_reparam_kl_bwd_bias with that dz: dfc [B][2z] (dtype) and dbias_fc [2z].  z <= 128, N % 64 == 0,
 * B*2z <= 8192 (z <= 32) / 16384 (z <= 128). */
int favae_decoder_input_bwd_reparam(const float* dact, const void* w, void* draw, float* dbias_di, float* dz, const float* fc,
                                    const float* eps, const double* prior, const int* idx, const float* stdq, float tau,
                                    float beta, int B, int z, int N, void* dfc, float* dbias_fc, int dtype, void* stream);
/* Gradient of (rec + beta*KL) w.r.t. fc given dz = d rec / d zlat; dfc [B][2z] (dtype). */
int favae_reparam_kl_bwd(const float* fc, const float* eps, const double* prior, const int* idx, const float* stdq,
                         const float* dz, float tau, float beta, int B, int z, void* dfc, int dtype, void* stream);
/* favae_reparam_kl_bwd that also sums the columns of dfc: dbias [2z] = bias gradient of fc_mu | fc_var (z <= 128, B * 2z <= 8192). */
int favae_reparam_kl_bwd_bias(const float* fc, const float* eps, const double* prior, const int* idx, const float* stdq,
                              const float* dz, float tau, float beta, int B, int z, void* dfc, float* dbias, int dtype,
                              void* stream);
/* update_x head (:348-354): X.mean = mu, X.cov = std^2 as float64 rows. */
int favae_encode_head(const float* fc, int B, int z, double* xmean, double* xcov, void* stream);
/* Sigmoid (:77) + -mean_b gaussian_LL (:152-171) + gradient w.r.t. the pre-sigmoid output (dtype).
 * raw, x: [B][D] fp32; *rec_err (fp64) = reconstruction_error of :238. scratch >= 1024 doubles. */
int favae_sigmoid_ll(const float* raw, const float* x, int B, long long D, float var_x, void* draw, double* rec_err,
                     double* scratch, size_t scratch_doubles, int dtype, void* stream);
/* favae_sigmoid_ll in one launch that also returns the bias gradient of the output conv: dbias[c] = sum over pixels of
 * the gradient written to draw, for C <= 4 interleaved channels (D = pixels * C per sample); dbias may be NULL.
 * scratch >= 5 * 1024 doubles. */
int favae_sigmoid_ll_bias(const float* raw, const float* x, int B, long long D, int C, float var_x, void* draw,
                          double* rec_err, float* dbias, double* scratch, size_t scratch_doubles, int dtype, void* stream);
/* final_layer.3-4 + the loss in one launch (:74-77 Conv2d(64 -> Cout <= 4, k3 s1 p1) + Sigmoid, :152-171 gaussian_LL,
 * :238): raw [Nb,H,W,Cout] = conv output (fp32), draw (dtype) = gradient w.r.t. raw, *rec_err, dbias [Cout] = bias
 * gradient.  act [Nb,H,W,64] dtype; wk = favae_conv_out_wprep of the [Cout][9][64] dtype weights (2304 dtype values in
 * the operand order of the 4x4x4 MFMA: tap, k/4, cout padded to 4, k%4);
 * x [Nb,H,W,Cout] fp32 targets.  scratch >= 5 * ceil(Nb*H*W / 256) doubles.  Same results as favae_conv_igemm +
 * favae_sigmoid_ll_bias up to fp32 summation order. */
int favae_conv_out_wprep(const void* w, void* wk, int Cout, int dtype, void* stream);
/* favae_conv_out_ll with FAVAE_DBIAS_DEFER in dtype writes neither rec_err nor dbias: scratch keeps five sums per block of 256
 * pixels ([ceil(Nb*H*W / 256)][5] doubles: squared error, then the bias-gradient sums), and the launch ends without the
 * hand-over between its blocks.  _finish adds them up with a launch of its own; _ride lets the calling thread's next
 * favae_conv_wgrad / favae_conv_wgrad_folded launch do it in its first block (the training step: final_layer.3's weight
 * gradient is queued next, beside the main chain).  No reference counterpart (a split of :368-381 + :249). */
int favae_conv_out_ll_finish(const double* partial, int Nb, int H, int W, int Cout, float var_x, double* rec_err, float* dbias,
                             void* stream);
int favae_conv_out_ll_ride(const double* partial, int Nb, int H, int W, int Cout, float var_x, double* rec_err, float* dbias);
/* favae_conv_out_ll whose input is the conv output raw_in [Nb][H][W][64] (bf16) in front of final_layer.1/.2 (BatchNorm2d(train) + LeakyReLU,
 * condition_a_VAE/celeba_deepvanilla_vae.py:72-73, applied by decode() :86-92): the activation lrelu(raw_in * a + b), rounded to bf16 as
 * favae_bn_train_fwd_bins / favae_gbn_fwd store it, is formed at the operand load from ab [groups][2][64] (a | b per BatchNorm group of
 * group_images images; 0: one group), so the activation tensor need not exist.  bf16 only (dtype FAVAE_BF16, optionally | FAVAE_DBIAS_DEFER),
 * W a multiple of 64. */
int favae_conv_out_ll_bn(const void* raw_in, const float* ab, int group_images, const void* wk, const float* bias, const float* x, int Nb,
                         int H, int W, int Cout, float var_x, float* raw, void* draw, double* rec_err, float* dbias, double* scratch,
                         size_t scratch_doubles, int dtype, void* stream);
/* The convolution of favae_conv_out_ll alone (final_layer.3 of the eval decoder, cA/celeba_deepvanilla_vae.py:71-77 under
 * decoder() :94-99): raw [Nb][H][W][Cout] fp32 = conv3x3(act) + bias.  bf16 activations, W a multiple of 64, wk from favae_conv_out_wprep. */
int favae_conv_out_raw(const void* act, const void* wk, const float* bias, int Nb, int H, int W, int Cout, float* raw, int dtype,
                       void* stream);
int favae_conv_out_ll(const void* act, const void* wk, const float* bias, const float* x, int Nb, int H, int W, int Cout,
                      float var_x, float* raw, void* draw, double* rec_err, float* dbias, double* scratch,
                      size_t scratch_doubles, int dtype, void* stream);
int favae_sigmoid(const float* raw, float* out, long long n, void* stream);
int favae_sigmoid_sse_u8(const float* raw, const unsigned char* y, long long n, double* acc3, double* scratch,
                         size_t scratch_doubles, void* stream);
/* acc3 += (sum (y - sigmoid(raw))^2, sum y, sum y^2). scratch >= 3*1024 doubles. */
int favae_sigmoid_sse(const float* raw, const float* y, long long n, double* acc3, double* scratch, size_t scratch_doubles,
                      void* stream);
/* torch.optim.Adam step (defaults, no weight decay; :131,250) over a flat fp32 arena of n (multiple of 4) floats. */
int favae_adam_step(float* p, const float* g, float* m, float* v, long long n, float lr, float beta1, float beta2,
                    float eps, long long step, void* stream);
/* The same step, and the gradient arena is cleared behind it: optim.step() (:250) fused with the optim.zero_grad() (:246)
 * of the iteration that follows, so that an epoch of steps needs no memset of the gradients. */
int favae_adam_step_clear(float* p, float* g, float* m, float* v, long long n, float lr, float beta1, float beta2,
                          float eps, long long step, void* stream);

/* favae_adam_step_clear on gscale * g (optim.step(), :250, on the averaged gradient): the data-parallel epoch all-reduces the SUM
 * of the ranks' gradients and passes 1 / ranks.  No counterpart of its own in the reference (which trains on one device). */
int favae_adam_step_clear_scaled(float* p, float* g, float* m, float* v, long long n, float lr, float beta1, float beta2, float eps,
                                 long long step, float gscale, void* stream);
/* favae_adam_step_clear_scaled that leaves g[0 .. keep0_hi) and g[keep1_lo .. keep1_hi) as they are (a gradient bucket of the data-parallel
 * epoch: the weight slices that the next step's unsplit weight-gradient GEMMs overwrite).  No counterpart in the reference. */
int favae_adam_step_clear_scaled_keep(float* p, float* g, float* m, float* v, long long n, float lr, float beta1, float beta2, float eps,
                                      long long step, float gscale, long long keep0_hi, long long keep1_lo, long long keep1_hi, void* stream);
/* favae_adam_step_clear that leaves g[0 .. keep) as it is (the training epoch: a weight gradient that the next step's GEMM overwrites
 * whole needs no zeroing; the bias / BatchNorm gradients behind it are accumulated into and do).  No counterpart in the reference. */
int favae_adam_step_clear_from(float* p, float* g, float* m, float* v, long long n, long long keep, float lr, float beta1, float beta2,
                               float eps, long long step, void* stream);
/* The number of reduction slices favae_conv_igemm picks for splitk <= 0 (1: every output element is stored once, the output need not be
 * zero; > 1: fp32 atomics into a zeroed output).  The executor keeps only the outputs of launches that split in its per-step memset. */
int favae_conv_igemm_auto_splitk(int Nb, int Ho, int Wo, int Cin, int Cout, int ks, int stride, int scatter);
/* The number of slices favae_conv_wgrad picks for splitk <= 0 (1: the launch stores its result, no accumulation into dw). */
int favae_conv_wgrad_auto_splitk(int Rc, int Cc, long long K, int ks);
/* favae_adam_step (clear_g: + the zeroing of favae_adam_step_clear) that also writes the bf16 copy of the first `ncopy`
 * updated parameters (a layer's weights, same [Cout][tap][Cin] layout) in the same pass: the optimiser step and the
 * dtype shadow of encoder.0, whose update is the last link of a training step (optim.step(), :250). */
int favae_adam_step_copy(float* p, float* g, float* m, float* v, long long n, float lr, float beta1, float beta2, float eps,
                         long long step, int clear_g, void* copy_bf16, long long ncopy, void* stream);
/* Adam step over the arena slice of ONE layer (n floats starting with its weight tensor [Co][taps][Ci], then bias / BN
 * affine) fused with the re-layout of the fresh weights (favae_weight_prep): the weights are read and written once.
 * clear_grad != 0 also zeroes the consumed gradients (see favae_adam_step_clear).  Ci, Co, n multiples of 4. */
int favae_adam_wprep(float* p, float* g, float* m, float* v, long long n, int Co, int taps, int Ci, void* w_copy,
                     void* w_transposed, int clear_grad, float lr, float beta1, float beta2, float eps, long long step,
                     int dtype, void* stream);
/* master fp32 [Co][taps][Ci] -> dtype copy (same layout, may be NULL) and dtype transpose [Ci][taps][Co] (may be NULL). */
/* The same for n weight tensors in one launch (host arrays of device pointers / shapes; n <= 16). */
int favae_weight_prep_batch(int n, const float* const* w, void* const* w_copy, void* const* w_transposed, const int* Co,
                            const int* taps, const int* Ci, int dtype, void* stream);
int favae_weight_prep(const float* w, void* w_copy, void* w_transposed, int Co, int taps, int Ci, int dtype, void* stream);
/* rows idx[b] of the resident fp32 NHWC dataset -> fp32 batch and/or dtype batch (DataLoader batch, :227-229). */
int favae_gather_batch(const float* data, const int* idx, int B, long long row_floats, float* x_f32, void* x_t, int dtype,
                       void* stream);
/* The same from a uint8 resident data set [N][row_elems] (NHWC): value k stands for k / 255 as a correctly rounded fp32
 * division, i.e. what torchvision's ToTensor makes of an 8-bit image (the trainers' loader, cA/favae_trainer.py:21-28). */
int favae_gather_batch_u8(const unsigned char* data, const int* idx, int B, long long row_elems, float* x_f32, void* x_t, int dtype,
                          void* stream);
/* fp32 NCHW -> uint8 NHWC; *mismatches += the number of values that are not exactly k / 255 (then the data set stays fp32). */
int favae_nchw_to_nhwc_u8(const float* in, unsigned char* out, long long N, int C, int HW, unsigned long long* mismatches,
                          void* stream);
int favae_cast(const float* in, void* out, long long n, int dtype, void* stream);
int favae_nchw_to_nhwc(const float* in, float* out, long long N, int C, int HW, int Cpad, void* stream);
int favae_nhwc_to_nchw_f64(const float* in, double* out, long long N, int C, int HW, int Cpad, void* stream);

/* ---- conv VAE executor (the training step of ImgVAE.trainloop, cA/celeba_deepvanilla_vae.py:227-250) ------------
 * The network is the reference's fixed architecture (:25-78: 5 x [Conv3x3 s2 + BN + LeakyReLU], two Linear heads,
 * Linear + 4 x [ConvT3x3 s2 + BN + LeakyReLU], ConvT + BN + LeakyReLU + Conv3x3 + Sigmoid) on 64x64 images.
 * The caller owns every buffer (params/grads/Adam moments: favae_vae_param_count floats each; BN running stats:
 * favae_vae_buffer_count floats; shadow + workspace: sizes from the *_bytes queries); all 256-byte aligned.
 * Layer order: 0-4 enc.i, 5 (fc_mu|fc_var), 6 decoder_input, 7-10 dec.i, 11 final_layer.0/1, 12 final_layer.3. */
typedef struct favae_vae favae_vae;
long long favae_vae_param_count(int z, int channels);
long long favae_vae_buffer_count(int z, int channels);
long long favae_vae_shadow_bytes(int z, int channels, int dtype);
long long favae_vae_workspace_bytes(int z, int channels, int batch_max, int dtype);
int favae_vae_num_layers(void);
/* info[11] = kind(0 conv s2,1 linear,2 convT s2,3 conv s1), Cin, Cout, taps, has_bn, w_off, b_off, gamma_off, beta_off,
 * running_mean_off, running_var_off (offsets in floats; weights are [Cout][taps][Cin]). */
int favae_vae_layer_info(int z, int channels, int layer, long long* info);
int favae_vae_create(favae_vae** out, int z, int channels, int batch_max, int dtype, float* params, float* grads,
                     float* adam_m, float* adam_v, float* bn_buffers, void* shadow, size_t shadow_bytes, void* workspace,
                     size_t workspace_bytes);
int favae_vae_destroy(favae_vae* h);
/* on != 0: the data_nhwc arguments of the calls below are uint8 NHWC arrays (value k = k / 255, see favae_gather_batch_u8)
 * instead of fp32: a quarter of the HBM residency (2.5 instead of 10 GB at N = 202 599), identical arithmetic for 8-bit images. */
int favae_vae_set_data_u8(favae_vae* h, int on);
int favae_vae_refresh_weights(favae_vae* h, void* stream);
/* forward + loss + backward of one mini-batch (:229-249); metrics2_out (device) = (reconstruction_error, kl_div). */
int favae_vae_train_fwdbwd(favae_vae* h, const float* data_nhwc, const int* idx, int B, const float* eps,
                           const double* prior, float tau, float beta, float var_x, double* metrics2_out, void* stream);
/* The same step in pieces, for data-parallel training that all-reduces gradient buckets (RCCL over xGMI) while later
 * layers are still in their backward pass: stage_lo == 0 runs the forward pass first; backward stages
 * [stage_lo, stage_hi) follow (0 final conv, 1-5 final_layer.0 / dec.3..dec.0, 6 decoder_input, 7 fc_mu|fc_var,
 * 8-12 enc.4..enc.0); stage_hi == 13 ends the step.  Same arguments on every call of a step, ascending gap-free ranges.
 * No reference counterpart (the reference is single-device; loss.backward() :249 is one call). */
int favae_vae_train_stages(favae_vae* h, const float* data_nhwc, const int* idx, int B, const float* eps,
                           const double* prior, float tau, float beta, float var_x, double* metrics2_out, int stage_lo,
                           int stage_hi, void* stream);
/* Orders stream `waiting` behind everything queued so far on `stream` and on the handle's weight-gradient stream (the
 * gradients of the stages issued so far are complete for work enqueued on `waiting` afterwards).  Blocks nothing. */
int favae_vae_fence(favae_vae* h, void* stream, void* waiting);
/* No reference counterpart (error plumbing of the threaded epochs): a stream-word wait that never got its write gives up after
 * ~4 s instead of hanging the process and leaves a mark; this returns FAVAE_ERR_HIP once if there is one (the results of that
 * epoch are invalid), 0 otherwise.  Call it after the epoch's outputs have been synchronised. */
int favae_vae_poll_error(favae_vae* h);

/* ---- data-parallel epoch over RCCL (xGMI) ---------------------------------------------------------------------- */
/* No reference counterpart (the reference is single-device).  One process per GPU; rank 0 calls favae_comm_unique_id, the
 * 128 bytes travel to the other ranks (the caller's bootstrap, e.g. a torch.distributed broadcast), every rank calls
 * favae_comm_init with its device current (collective: ncclCommInitRank).  RCCL is resolved at run time from the copy the
 * process already holds (PyTorch's); rccl_path (may be NULL) names a librccl.so to load if there is none. */
typedef struct favae_comm favae_comm;
int favae_comm_unique_id(const char* rccl_path, void* id128);
int favae_comm_init(favae_comm** out, const char* rccl_path, const void* id128, int rank, int nranks);
int favae_comm_destroy(favae_comm* c);
/* In-place all-reduce (sum, or mean with average != 0) of n floats on `stream`. */
int favae_allreduce_f32(favae_comm* c, float* buf, long long n, int average, void* stream);
/* The same for n doubles: the sufficient statistics of the FA updates (X^T Z, Z^T Z, the sums of the multilabel update and of the
 * reconstruction diagnostic) travel over the same communicator as the gradient buckets, on the caller's stream. */
int favae_allreduce_f64(favae_comm* c, double* buf, long long n, int average, void* stream);
/* favae_vae_train_epoch for data-parallel training (each rank passes its own shard / permutation / eps): per step the
 * gradient arena is averaged over the ranks in four buckets (decoder | enc.4 + heads | enc.3 | enc.0-2); a bucket's all-reduce
 * and the Adam step + weight re-layout of its layers run on the communicator's stream as soon as its last weight
 * gradient is complete, under the backward pass of the layers behind it; the next forward pass waits for that stream.
 * Same arithmetic as fwd/bwd -> all-reduce(mean) -> Adam (replicated parameters stay bit-identical across ranks). */
int favae_vae_train_epoch_dp(favae_vae* h, favae_comm* c, const float* data_nhwc, const int* perm, long long n, int batch,
                             const float* eps, const double* prior, float tau, float beta, float var_x, float lr,
                             long long step0, double* metrics, void* stream);
/* optim.step() (:250) + weight re-layout. */
int favae_vae_adam(favae_vae* h, float lr, float beta1, float beta2, float eps, long long step, void* stream);
/* all mini-batches of one epoch (:227-255) on one device. perm [n], eps [n][z] in visiting order, metrics [steps][2]. */
int favae_vae_train_epoch(favae_vae* h, const float* data_nhwc, const int* perm, long long n, int batch, const float* eps,
                          const double* prior, float tau, float beta, float var_x, float lr, long long step0,
                          double* metrics, void* stream);
/* eval-mode encoder of update_x (:331-358) and eval-mode decoder of reconstruction/decoder (:94-99, 360-385). */
int favae_vae_encode(favae_vae* h, const float* data_nhwc, const int* idx, int B, double* xmean, double* xcov, int train,
                     void* stream);
int favae_vae_encode_all(favae_vae* h, const float* data_nhwc, long long n, int batch, double* xmean, double* xcov,
                         void* stream);
int favae_vae_decode(favae_vae* h, const float* zin, int B, float* out_nhwc, int train, void* stream);
/* reconstruction + r2_score sums of SSHIBA.update (cA/favae.py:839-840) without leaving the device:
 * acc3 = (sum (y-yhat)^2, sum y, sum y^2) over the n decoded rows against the resident data. */
int favae_vae_reconstruct_stats(favae_vae* h, const float* zin, const float* data_nhwc, long long n, int batch,
                                double* acc3, void* stream);
int favae_vae_debug_ptr(favae_vae* h, int which, int layer, void** out);

/* ---- factor-analysis half (SSHIBA mean-field updates, float64) -------------------------------------------------- */

/* C[da][db+ones] = A^T [B | 1] over N rows: the sufficient statistics X^T Z, Z^T Z, X^T X and the column sums
 * (cA/favae.py:989 Z^T Z, :1070 (X-b)^T Z, :1091 column sums, :1135-1139 traces, :1213 X^T X).  fp64 MFMA, slabs
 * summed in a fixed order.  scratch >= favae_fa_atb_scratch_doubles(...). db may be 0 with ones=1 (column sums).
 * B == A with db == da and ldb == lda (Z^T [Z | 1]): the rows are read once and C[:, :da] is exactly symmetric (the tiles below
 * the diagonal are the transposes of those above; bitwise the values of the full product). */
long long favae_fa_atb_scratch_doubles(long long N, int da, int db, int ones);
int favae_fa_atb(const double* A, int lda, int da, const double* B, int ldb, int db, int ones, long long N, double* C,
                 double* scratch, size_t scratch_doubles, void* stream);
/* favae_fa_atb for callers that cannot allocate device memory (code under HIP graph capture, tight serving loops,
 * harnesses that forbid hipMalloc).  No counterpart in the reference.  The slab partials go to a file-scope device array of
 * the library (32 MB = 4 194 304 doubles per GPU, held until the process ends) instead of a caller-supplied scratch; the
 * call allocates nothing and returns an error if da * (db + ones) > 4 194 304 (any N fits below that).  Same kernels and
 * geometry as favae_fa_atb: bit-identical results.  Every call on a GPU shares that GPU's array: calls that overlap on
 * one GPU (two streams) race; calls on one stream, or on different GPUs, do not. */
int favae_fa_atb_static(const double* A, int lda, int da, const double* B, int ldb, int db, int ones, long long N, double* C,
                        void* stream);
/* out[N][K] = alpha (X[N][d] P[d][K] + c[K]) + beta base[N][K]; accumulate=1: out += alpha X P (any K: column panels of 208).
 * q.Z['mean'] (cA/favae.py:986-987), the predictive means (:996-997), the FA prior Z W^T + b
 * (cA/celeba_deepvanilla_vae.py:209) and Z W^T + b of the multilabel update (cA/favae.py:1212). */
int favae_fa_rows_gemm(const double* X, int ldx, int d, const double* P, int ldp, int K, const double* c, double alpha,
                       const double* base, int ldbase, double beta, double* out, int ldo, long long N, int accumulate,
                       void* stream);
/* Multilabel view, update_x + update_xi fused (cA/favae.py:1197-1229) with the entropy / bound sums (:1649, :1678):
 * out2[0] = sum log cov, out2[1] = sum log sig(xi) + mean*t - (mean+xi)/2.  n = N*d elements. scratch >= 4096 doubles.
 * The reference's nested forms (sigmoid = exp(-log(1 + exp(-x))), lambda = exp(log(sig - 1/2) - log(2 xi)), :924-949) are evaluated
 * as 1 / (1 + exp(-x)) and (sig - 1/2) / (2 xi): equal to a few roundings (1e-13 against the forms as written is tested). */
int favae_fa_mult_update(const double* t, const double* zwb, double* xi, double* mean, double* cov, double tau, long long n,
                         double* out2, double* scratch, size_t scratch_doubles, void* stream);
/* Semi-supervised multilabel view, update_t (cA/favae.py:1176-1195) + t.mean[mask] = sigmoid(X.mean)[mask] (:858-861):
 * mask [n] bytes (1 = label unobserved), t updated in place; out2 = (sum over all entries of log tS.cov,
 * sum_mask X.mean * (t_new - t_old) = the change of the bound term sum X.mean * t.mean of :1678).  scratch as above. */
int favae_fa_mult_ss_update(const double* mean, const unsigned char* mask, double* t, long long n, double* out2,
                            double* scratch, size_t scratch_doubles, void* stream);
int favae_fa_xi_init(const double* mean, const double* cov, double* xi, long long n, void* stream);
/* Multilabel output of SSHIBA.predict (cA/favae.py:1469-1481): for rows of m_x [N][d] and scale[d] = (1 + pi/8 var_x[d,d])^-1/2,
 * mean_t = sigmoid(m_x scale), var_t = exp(s) / (1 + exp(s)^2) (as the reference writes it). */
int favae_fa_mult_predict(const double* m_x, const double* scale, double* mean_t, double* var_t, long long N, int d,
                          void* stream);
/* Batched SPD inverse through the Cholesky factor (myInverse cA/favae.py:904-922; the W covariance :1062-1065 with
 * s = alpha.mean^-1/2, diag_add = out_scale = 1/tau; the M + 1 -- variant A: 2M + 1 -- inverses of update_Z :970-980): nb <= 12
 * matrices of size K in ONE launch, one 16-wave workgroup per matrix, the matrix resident in LDS as a packed lower triangle
 * (K <= 201), Cholesky / triangular inverse / X^T X blocked in 16 x 16 tiles on the fp64 matrix cores.  Matrix b:
 *   A'_b = (s_b s_b^T) o (sum_{t < nterm[b]} coef[8 b + t] * A[8 b + t]) + diag_add[b] I     (the sums of :971, :979 are formed on load)
 *   out[b] = out_scale[b] (s_b s_b^T) o A'_b^-1,   logdet2[b] = (log det A'_b, log det out[b]),   info[b] = 0, or 1: not SPD, NaN
 * s may be NULL, s[b] may be NULL (ones); s_sqrt[b] = 1 takes s_b[i] = sqrt(s[b][i] * s_mul[b]) (alpha.b / alpha.a on the device).
 * ASYNCHRONOUS: nothing is read back; look at info when the results are used.  K > 201 (<= 1024): matrix by matrix through
 * `work` (>= 2 K^2 + 2 K doubles) in global memory; work may be NULL otherwise. */
int favae_chol_inv_batched(int nb, const double* const* A, const double* coef, const int* nterm, int lda, int K,
                           const double* const* s, const double* s_mul, const int* s_sqrt, const double* diag_add,
                           const double* out_scale, double* const* out, int ldo, double* logdet2, int* info_dev, double* work,
                           size_t work_doubles, void* stream);
/* The same for one matrix with one term; s may be NULL; work: 2*K*K + 2*K doubles.  SYNCHRONOUS (reads info back); returns
 * FAVAE_NOT_SPD and fills out with NaN when a pivot is not positive (the reference's NaN path). */
/* out2 = (sign, log|det|) of a general K x K matrix (np.linalg.slogdet in HGauss, cA/favae.py:1603). work: K*K doubles. */
int favae_slogdet(const double* A, int lda, int K, double* work, double* out2, void* stream);
/* C[M][N] = alpha op(A) op(B) + beta C for the small (d x K, K x K) products of update_w / update_Z / predict. */
int favae_small_gemm(int transA, int transB, int M, int N, int Kd, double alpha, const double* A, int lda, const double* B,
                     int ldb, double beta, double* C, int ldc, void* stream);
/* W.mean = tau (X^T Z - b^T 1^T Z) Sigma_W, W.prodT = W^T W + d Sigma_W (cA/favae.py:1070-1072). xtz1 = X^T [Z|1].
 * info (may be NULL): device flag of the inverse that produced wcov (favae_chol_inv_batched); when it is set nothing is written
 * -- the reference's "not invertible, not updated" (:1073-1075) without a read-back.  wcov_out (may be NULL): receives a copy of
 * wcov, so that Sigma_W too is only replaced when the inverse exists. */
int favae_fa_w_update(const double* xtz1, const double* zsum, const double* b, const double* wcov, double tau, int d, int K,
                      double* wmean, double* wprodT, double* tmp, const int* info, double* wcov_out, void* stream);
/* b update (cA/favae.py:1090-1092): bmean[d]; scal[0] = Sigma_b[0,0], scal[1] = <b b^T>. */
int favae_fa_b_update(const double* xsum, const double* zsum, const double* wmean, double tau, double N, int d, int K,
                      double* bmean, double* scal, void* stream);
/* tau update (cA/favae.py:1134-1139) from reduced statistics: scal[2] = a, scal[3] = b. trxx = tr<X^T X> incl. sum cov.
 * Device-resident forms of two scalars (each may be NULL): bprodT_dev overrides bprodT with the value favae_fa_b_update left
 * at scal[1]; xtx1_dev (X^T [X | 1], d x (d + 1)) with covsum_dev (column sums of X.cov, or NULL) overrides trxx. */
int favae_fa_tau_update(const double* xsum, const double* zsum, const double* xtz1, double trxx, const double* zprodT,
                        const double* wprodT, const double* wmean, const double* bmean, double bprodT, double a0, double b0,
                        double N, int d, int K, double* scal, const double* bprodT_dev, const double* xtx1_dev,
                        const double* covsum_dev, void* stream);
/* alpha.b = (b0 + diag(W.prodT)/2)/d (cA/favae.py:1107-1112). */
int favae_fa_alpha_update(const double* wprodT, double b0, int d, int K, double* alpha_b, void* stream);
/* aux = I + sum_m coef[m] WprodT_m (cA/favae.py:970-971, 976-980). wprodT: HOST array of m device pointers, m <= 8. */
int favae_fa_z_aux(const double* const* wprodT, const double* coef, int m, int K, double* aux, void* stream);
/* out = a x + b y (y may be NULL): Z.prodT = Z^T Z + N Sigma_Z (cA/favae.py:989) and similar small combinations. */
int favae_axpby(long long n, double a, const double* x, double b, const double* y, double* out, void* stream);
/* out[N][Kn] = in[N][K][:, sel] (pruning of q.Z['mean'], cA/favae.py:507). */
int favae_gather_cols(const double* in, int K, const int* sel, int Kn, double* out, long long N, void* stream);

#ifdef __cplusplus
}
#endif
#endif
