// Internal interface of smallconv.hip (weight gradient with an image-sized channel count on one side).
#pragma once
#include <hip/hip_runtime.h>
#include "favae_hip.h"

bool wgrad_nw_applicable(int Cw, int Cs, int ks, int stride, int narrow_is_large);
int wgrad_nw_launch(const void* wide, const void* narrow, float* dw, int Nb, int Hq, int Wq, int Cw, int Hn, int Wn, int Cs,
                    int stride, int narrow_is_large, int s_ch, int s_tap, int s_c, int dtype, hipStream_t st);
bool thin_in_applicable(int Cs, int Co, int ks, int stride, int scatter);
// optional BatchNorm-backward sums taken while the result is stored (bins == nullptr: none); see favae_conv_igemm_red mode 2
// group_pixels > 0 (grouped-batch mode): output pixels per BatchNorm group (a multiple of 256); group g = pixel / group_pixels takes
// bins + g * nbin * 2 * Co and mean / invstd + g * Co
struct ThinRed { double* bins; int nbin; const void* act; const void* raw; const float *mean, *invstd; int group_pixels; int raw_bf16; };
int thin_in_launch(const void* in, const void* w, const float* bias, float* out, int Nb, int Hi, int Wi, int Cs, int Ho, int Wo,
                   int Co, int stride, int scatter, int dtype, int out_bf16, const ThinRed& red, hipStream_t st);

void thin_in_release_stream(hipStream_t st);   // st is going away: free its slot of re-laid-out weights

// Gather of the 3x3 neighbourhood of a <= 3-channel NHWC tensor into rows of 32 values (zero padded), one row per
// pixel of the (Hq x Wq) grid the caller's reduction runs over.
//   shifted = 0: dst[pix][tap*Cs + c] = src[n][s*y - 1 + kh][s*x - 1 + kw][c]      (src on the Hn x Wn grid, stride s)
//   shifted = 1: dst[pix][c*9 + tap] = src[n][y + 1 - kh][x + 1 - kw][c]          (stride 1, same grid)
int fold_taps_launch(const void* src, void* dst, int Nb, int Hq, int Wq, int Hn, int Wn, int Cs, int stride, int shifted,
                     int dtype, hipStream_t st);

// out [P][64] fp32 = fold [P][32] (bf16, columns >= K zero) x w [64][K]^T (bf16) + bias; optional BatchNorm sums (sum, sum of
// squares per channel, fp64) added to bins [nbin][2][64].  See folded_fwd_kernel.
// split != 0: fold holds the hi and the lo plane of fold_taps_hl_launch ([2][P][32]) and w is the fp32 weight [64][K].
// groups > 1 (grouped-batch mode): P / groups pixels per BatchNorm group (a multiple of 32), bins [groups][nbin][2][64]
int folded_fwd_launch(const void* fold, const void* w, int K, const float* bias, float* out, long long P, double* bins, int nbin,
                      int split, hipStream_t st, int groups = 1);
// fold_taps (shifted = 0) of an fp32 source into two bf16 planes: dst [P][32] = bf16(x), dst + P * 32 = bf16(x - bf16(x))
// Input gradient of a 3x3 stride-1 convolution with <= 3 output channels over its tap-folded output gradient (fold_taps_launch,
// shifted = 1): out [P][64] (fp32, or bf16 with out_bf16) = fold [P][32] x w [K][64], w = the layer's weight [Cout][9][64] as it lies.
// bins != NULL: the BatchNorm-backward sums of the layer in front per group of P / groups pixels (bins [groups][nbin][2][64], act bf16,
// raw fp32 or bf16, mean / invstd [groups][64]).
int folded_dgrad_launch(const void* fold, const void* w, int K, void* out, long long P, int groups, int out_bf16, double* bins, int nbin,
                        const void* act, const void* raw, int raw_bf16, const float* mean, const float* invstd, const float* ab, hipStream_t st);
int folded_dgrad_bn_launch(const void* fold, const void* w, int K, void* draw, long long P, int groups, int grad_bf16, const double* bins, int nbin,
                           const void* act, const void* raw, int raw_bf16, const float* mean, const float* invstd, const float* gamma, const float* ab,
                           float* dgamma, float* dbeta, double* dbias_part, int slabs_cap, int* slabs_out, hipStream_t st);
int fold_taps_hl_launch(const float* src, void* dst, int Nb, int Hq, int Wq, int Hn, int Wn, int Cs, int stride, hipStream_t st);
