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

const char* favae_last_error(void);
int favae_version(void);

/* ---- conv VAE: GEMM-shaped kernels ------------------------------------------------------------------------- */

/* Implicit-GEMM convolution, gather form (scatter=0) or transposed/scatter form (scatter=1).
 *   scatter=0: nn.Conv2d forward (cA/celeba_deepvanilla_vae.py:31-33, 74-76), the input gradient of
 *              nn.ConvTranspose2d and nn.Linear forward / input gradient (:41-45) with ks=1 on a 1x1 grid.
 *   scatter=1: nn.ConvTranspose2d forward (:50-57, 64-71) and the input gradient of nn.Conv2d (loss.backward() :249).
 * in [Nb,Hi,Wi,Cin] dtype, w [Cout][ks*ks][Cin] dtype, bias [Cout] fp32 or NULL, out [Nb,Ho,Wo,Cout] fp32.
 * splitk: 0 = choose; >1 splits the reduction and accumulates with fp32 atomics (out is zeroed first). */
int favae_conv_igemm(const void* in, const void* w, const float* bias, float* out, int Nb, int Hi, int Wi, int Cin,
                     int Ho, int Wo, int Cout, int ks, int stride, int pad, int scatter, int splitk, int dtype,
                     void* stream);

/* Weight gradient  dw[r][tap][c] += sum_pixels rows[pix][r] * cols[pix'][c]  (fp32 atomics; caller zeroes dw).
 * Exactly one of the two operands may be the large-spatial tensor (Hl x Wl), read at (s*h-p+kh, s*w-p+kw);
 * the other lives on the small grid (Hs x Ws).  Conv2d: rows = dout (small), cols = input (large).
 * ConvTranspose2d: rows = dout (large), cols = input (small).  Linear: 1x1 grids, ks=1.
 * Replaces the weight half of loss.backward(), cA/celeba_deepvanilla_vae.py:249. */
int favae_conv_wgrad(const void* rows, int Rc, int rows_large, const void* cols, int Cc, int cols_large, float* dw,
                     int Nb, int Hs, int Ws, int Hl, int Wl, int ks, int stride, int pad, int splitk, int dtype,
                     void* stream);

#ifdef __cplusplus
}
#endif
#endif
