// Implicit-GEMM convolution kernels for the FA-VAE conv VAE on gfx950 (MI355X).
//
// Replaces the ATen/cuDNN calls behind reference condition_a_VAE/celeba_deepvanilla_vae.py:25-99
// (nn.Conv2d k3 s2 p1, nn.ConvTranspose2d k3 s2 p1 op1, nn.Conv2d k3 s1 p1, nn.Linear) and their autograd
// backward (loss.backward(), :249).  Activations are NHWC, weights are [Cout][tap][Cin] ("OHWI"), so both
// GEMM operands are contiguous along the reduction index and one 32x32 MFMA tile loop serves all layers:
//
//   igemm_kc  gather  form : out[n,oy,ox,co] = sum_{tap,ci} in[n, oy*s-p+kh, ox*s-p+kw, ci] * w[co][tap][ci]
//                            (Conv2d forward, ConvTranspose2d input gradient, Linear forward / input gradient)
//             scatter form : out[n,y,x,co]   = sum_{tap,ci : (y+p-kh)%s==0} in[n,(y+p-kh)/s,(x+p-kw)/s,ci] * w[co][tap][ci]
//                            (ConvTranspose2d forward, Conv2d input gradient), one launch z-slice per output
//                            parity class so that no MFMA is spent on structurally-zero taps
//   igemm_pm               : dw[r][tap][c] += sum_{pixels} R[pix][r] * C[pix'][c]   (all weight gradients)
//
// T = float  -> v_mfma_f32_32x32x2_f32 (exact fp32, parity path);  T = bf16 -> v_mfma_f32_32x32x16_bf16.
// Accumulation is always fp32; outputs are fp32.
#include <algorithm>
#include <cstdlib>
#include "common.h"
#include "favae_hip.h"
#include "smallconv.h"

#include <atomic>
#include <mutex>
#include <vector>

namespace {

constexpr int BK = 64;  // reduction elements per LDS stage

int g_force_kc_tile = -1, g_force_pm_tile = -1;  // developer override (favae_debug_force_tile), -1 = heuristic

#ifdef FAVAE_KC_STAMPS   // developer build: cycle stamps of one wave of the middle block (tools/kc_stamps.py)
__device__ unsigned long long g_kc_stamps[64];
#define KC_STAMP(i) do { if (blockIdx.x == gridDim.x / 2 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0) g_kc_stamps[i] = __builtin_readcyclecounter(); } while (0)
// start / end of every block on the constant 100 MHz clock (comparable across CUs), up to 8192 blocks
__device__ unsigned long long g_kc_blocks[8192 * 2];
#define KC_BLOCK(e) do { const unsigned b_ = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z); \
                         if (threadIdx.x == 0 && b_ < 8192) g_kc_blocks[2 * b_ + (e)] = wall_clock64(); } while (0)
#else
#define KC_STAMP(i)
#define KC_BLOCK(e)
#endif

template <class T> struct Frag;
template <> struct Frag<float> { float v[8]; };
template <> struct Frag<bf16_t> { bf16x8 v; };

__device__ __forceinline__ void mma32(f32x16& acc, const Frag<float>& a, const Frag<float>& b) {
    // lane (r, h) holds k = 8h..8h+7 of its row/column; the k order inside the 16-deep step is a
    // permutation applied to both operands, so 8 x (32x32x2) sums exactly the 16 products.
#pragma unroll
    for (int j = 0; j < 8; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.v[j], b.v[j], acc, 0, 0, 0);
}
__device__ __forceinline__ void mma32(f32x16& acc, const Frag<bf16_t>& a, const Frag<bf16_t>& b) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v, b.v, acc, 0, 0, 0);
}

// fragment of 8 consecutive reduction elements from a [row][k] LDS image
__device__ __forceinline__ void frag_kc(Frag<float>& f, const char* p) {
    const float4 lo = *(const float4*)p, hi = *(const float4*)(p + 16);
    f.v[0] = lo.x; f.v[1] = lo.y; f.v[2] = lo.z; f.v[3] = lo.w;
    f.v[4] = hi.x; f.v[5] = hi.y; f.v[6] = hi.z; f.v[7] = hi.w;
}
__device__ __forceinline__ void frag_kc(Frag<bf16_t>& f, const char* p) {
    union { uint4 u; bf16x8 b; } c;
    c.u = *(const uint4*)p;
    f.v = c.b;
}

// fragment from a [k][channel] LDS image (reduction index strided): lane (r, h) needs k = k0+8h+j, channel cb+r
__device__ __forceinline__ void frag_pm(Frag<float>& f, const char* lds, int RS, int k0, int cb, int lane) {
    const char* p = lds + (k0 + 8 * (lane >> 5)) * RS + (cb + (lane & 31)) * 4;
#pragma unroll
    for (int j = 0; j < 8; ++j) f.v[j] = *(const float*)(p + j * RS);
}
// bf16 image: rows of RS = 64, 128 or 256 bytes (32 / 64 / 128 channels), NO padding, the 64-byte pieces of a row permuted by
// pm_swz(row).  ds_read_b64_tr_b16 banks are (a / 4) % 64 and conflicts count per 32-lane half; a half reads, for each of 4
// consecutive k-rows, one 64-byte piece (two 16-lane groups side by side).  Conflict-free means the four pieces lie in the four
// 64-byte quarters of the 256-byte bank row: 64-byte rows do by themselves; 128-byte rows put rows r and r + 2 on one quarter, so
// the piece index is XORed with bit 1 of the row; 256-byte rows put all four on one quarter, so it is XORed with row & 3.
// (Round 2 padded the rows by 16 bytes instead: 144-byte rows leave rows 0 / 2 and 1 / 3 of a half overlapping in 8 banks -- the
// 33 % LDS bank-conflict cycles of profiles/r2_sq_counters_gemm_lean.txt.)
__device__ __forceinline__ int pm_swz(int row, int RS) { return RS == 128 ? ((row >> 1) & 1) : RS == 256 ? (row & 3) : 0; }
__device__ __forceinline__ void frag_pm(Frag<bf16_t>& f, const char* lds, int RS, int k0, int cb, int lane) {
    // ds_read_b64_tr_b16: each 16-lane group reads a 4(k) x 16(channel) block and receives it column-major
    const int i = lane & 15, q = i >> 2;
    const int row = k0 + 8 * (lane >> 5) + q;                        // (k0 + 8 h is a multiple of 8: the swizzle of row + 4 is the same)
    const int piece = (cb >> 5) ^ pm_swz(q, RS);
    const char* p = lds + row * RS + piece * 64 + 32 * ((lane >> 4) & 1) + 8 * (i & 3);
    union { s16x4 s[2]; bf16x8 b; } c;
    c.s[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p));
    c.s[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p + 4 * RS));
    f.v = c.b;
}
// byte offset of 16-byte chunk `chunk` of k-row `row` in that image (fp32 images: plain padded rows)
template <class T> __device__ __forceinline__ int pm_off(int row, int chunk, int RS) {
    if constexpr (sizeof(T) == 2) return row * RS + ((((chunk >> 2) ^ pm_swz(row, RS)) << 2) | (chunk & 3)) * 16;
    else return row * RS + chunk * 16;
}

// value select (a ?: on two uint4 lvalues would be an lvalue select and force the register arrays into memory)
__device__ __forceinline__ uint4 keep_if(bool ok, uint4 v) {
    v.x = ok ? v.x : 0u; v.y = ok ? v.y : 0u; v.z = ok ? v.z : 0u; v.w = ok ? v.w : 0u;
    return v;
}

template <class T> __device__ __forceinline__ void put_elem(uint4& v, int e, T x);
template <> __device__ __forceinline__ void put_elem<float>(uint4& v, int e, float x) {
    unsigned u = __float_as_uint(x);
    if (e == 0) v.x = u; else if (e == 1) v.y = u; else if (e == 2) v.z = u; else v.w = u;
}
template <> __device__ __forceinline__ void put_elem<bf16_t>(uint4& v, int e, bf16_t x) {
    unsigned u = ((unsigned)x) << (16 * (e & 1));
    const int w = e >> 1;
    if (w == 0) v.x |= u; else if (w == 1) v.y |= u; else if (w == 2) v.z |= u; else v.w |= u;
}

struct KcParams {
    const void* in;
    const void* w;
    const float* bias;
    float* out;
    int Nb, Hi, Wi, Cin, Ho, Wo, Cout;
    int ks, stride, pad, ldw, splitk, M, ntn, HoC, WoC;
    // log2 of HoC * WoC and of WoC where both are powers of two (every layer of this network), else -1: a row index splits into
    // (image, y, x) with shifts; the integer divisions were ~35 instructions each, 8-16 per thread of a block whose whole
    // reduction is a few hundred
    int hw_shift, woc_shift;
    int out_bf16;   // store the result as bf16 (unsplit reductions only)
    // > 0: blocks are dealt to the 8 XCDs round-robin in launch order, so block b of the launch computes tile
    // (b % 8) * xcd_chunk + b / 8: every XCD works on one contiguous run of tiles, and the tiles that share activation rows
    // (the n-tiles of one m-tile, the halo rows of neighbouring m-tiles) meet in one L2 instead of eight
    int xcd_chunk;
    // optional per-channel sums over the output tile, added to red_bins[bin][2][Cout] (fp64 atomics; staged 64x64
    // epilogue of unsplit reductions only).  mode 1: sum(y), sum(y^2) of the stored output (BatchNorm statistics);
    // mode 2: y is the gradient w.r.t. a BN+LeakyReLU output: sum(dd), sum(dd * xhat) with dd = y * lrelu'(act),
    // xhat = (raw - mean) * invstd (the two reductions of the BatchNorm backward).
    double* red_bins;
    int red_nbin, red_mode;
    // grouped-batch mode (BatchNorm groups of red_group_rows consecutive rows of this launch's row space; 0: one group): the tile's
    // group g = m0 / red_group_rows takes bins + g * nbin * 2 * Cout and, in mode 2, mean / invstd + g * Cout.  Tiles never
    // straddle groups: a group is 64 images, i.e. a multiple of 64 rows.
    int red_group_rows;
    const void* red_act;
    const void* red_raw;   // fp32, or bf16 with red_raw_bf16
    const float *red_mean, *red_invstd;
    const float* red_ab;   // grouped-batch mode, mode 2: [groups][2][Cout] affine form of the forward pass; the slope from raw * a + b, act is not read
    int red_raw_bf16;
    // FI kernels: `in` is the fp32 output `raw` [pixels][Cin] of a convolution whose BatchNorm(train) + LeakyReLU has not been
    // applied; the A operand is lrelu(raw * a[c] + b[c]) rounded to T, formed when a chunk goes to LDS.  in_ab = a[Cin] | b[Cin]
    // as published by the producing GEMM (fin_* below).
    const float* in_ab;
    // red_mode 1 with fin_ab != NULL: the block that takes the last ticket of the launch adds the bins and publishes the
    // BatchNorm(train) quantities of THIS layer: mean, invstd, a = invstd * gamma, b = beta - mean * a, running statistics.
    const float *fin_gamma, *fin_beta;
    float *fin_mean, *fin_invstd, *fin_ab, *fin_run_mean, *fin_run_var;
    double fin_R; float fin_eps, fin_mom;
    unsigned* fin_ticket;
    // eval-mode BatchNorm + LeakyReLU in the epilogue (staged 64x64 epilogue of unsplit reductions): the stored value is
    // lrelu(y * a[c] + b[c]) with ep_ab = a[Cout] | b[Cout] from the running statistics
    const float* ep_ab;
    FlagCarry carry;   // dependency words of the launch stream, stored by the first block (common.h)
};

// 8 (bf16) or 4 (fp32) channels c0.. of the fused-input operand: BN affine + LeakyReLU(0.01), same expressions and rounding as
// bn_lrelu_fwd_ab (vae_kernels.hip), so the bytes equal those of the materialised activation
template <class T> __device__ __forceinline__ uint4 bn_act_vec(const uint4* raw, const float* ab, int c0, int Cin);
template <> __device__ __forceinline__ uint4 bn_act_vec<float>(const uint4* raw, const float* ab, int c0, int Cin) {
    const float4 a = *(const float4*)(ab + c0), b = *(const float4*)(ab + Cin + c0);
    float y0 = fmaf(__uint_as_float(raw[0].x), a.x, b.x), y1 = fmaf(__uint_as_float(raw[0].y), a.y, b.y);
    float y2 = fmaf(__uint_as_float(raw[0].z), a.z, b.z), y3 = fmaf(__uint_as_float(raw[0].w), a.w, b.w);
    y0 = y0 > 0.f ? y0 : 0.01f * y0; y1 = y1 > 0.f ? y1 : 0.01f * y1; y2 = y2 > 0.f ? y2 : 0.01f * y2; y3 = y3 > 0.f ? y3 : 0.01f * y3;
    return make_uint4(__float_as_uint(y0), __float_as_uint(y1), __float_as_uint(y2), __float_as_uint(y3));
}
template <> __device__ __forceinline__ uint4 bn_act_vec<bf16_t>(const uint4* raw, const float* ab, int c0, int Cin) {
    unsigned o[4];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const float4 a = *(const float4*)(ab + c0 + 4 * h), b = *(const float4*)(ab + Cin + c0 + 4 * h);
        float y0 = fmaf(__uint_as_float(raw[h].x), a.x, b.x), y1 = fmaf(__uint_as_float(raw[h].y), a.y, b.y);
        float y2 = fmaf(__uint_as_float(raw[h].z), a.z, b.z), y3 = fmaf(__uint_as_float(raw[h].w), a.w, b.w);
        y0 = y0 > 0.f ? y0 : 0.01f * y0; y1 = y1 > 0.f ? y1 : 0.01f * y1; y2 = y2 > 0.f ? y2 : 0.01f * y2; y3 = y3 > 0.f ? y3 : 0.01f * y3;
        o[2 * h] = pack_bf16x2(y0, y1);
        o[2 * h + 1] = pack_bf16x2(y2, y3);
    }
    return make_uint4(o[0], o[1], o[2], o[3]);
}

// 16 bytes through a buffer descriptor: a 32-bit per-lane byte offset + a wave-uniform offset, and the range check of the
// descriptor returns zeros for an offset at or past its size -- which is how halo taps, tail rows and tail output channels
// read as zero without a select per element (their offsets get bit 31 set; descriptors are at most 0x7FFFFFF0 bytes long)
typedef unsigned v4u_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 buf_ld16(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    const v4u_t v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0);
    return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t buf_rsrc(const void* base, size_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)(bytes < (size_t)0x7FFFFFF0 ? bytes : (size_t)0x7FFFFFF0), 0x00020000);
}
constexpr unsigned BUF_OOB = 0x80000000u;
// sum of two byte offsets either of which may carry the out-of-range bit
__device__ __forceinline__ unsigned buf_off_add(unsigned a, unsigned b) { return ((a & ~BUF_OOB) + (b & ~BUF_OOB)) | ((a | b) & BUF_OOB); }
__device__ __forceinline__ unsigned outside_bit(int v, int n) { return (unsigned)v >= (unsigned)n ? 1u : 0u; }
// kernel row / column of the i-th valid tap of a launch slice: stride-2 scatter form: parity class pp uses taps {1} or {0, 2}
template <bool TR> __device__ __forceinline__ int tap_k(bool s2, int pp, int i) { return (TR && s2) ? (pp ? 2 * i : 1) : i; }

// ---------------------------------------------------------------------------------------------------------
// K-contiguous implicit GEMM (gather / scatter forms)
// ---------------------------------------------------------------------------------------------------------
// KG = 2: the block has two groups of 4 waves; the groups take alternate reduction chunks of the same output tile through LDS
// stages of their own and are added through LDS at the end (split-K inside the block).  The batch-64 layers are 256-512 tiles
// of 9-36 chunks on 256 CUs, i.e. one or two 4-wave blocks per CU, and with that few waves nothing of a chunk overlaps
// (LDS write -> barrier -> LDS read -> MFMA runs serially, ~0.5 us per chunk measured); a second group of waves per tile
// doubles the waves per SIMD without atomics or more tiles.
// FI: fused input (see KcParams::in_bins): BatchNorm + LeakyReLU of the producing layer applied to the A operand on the fly.
// row index m of a launch -> image nb, output position (yy, xx) of its (parity-class) grid
#define KC_SPLIT_M(m_, nb_, yy_, xx_)                                                                              \
    do {                                                                                                           \
        if (p.hw_shift >= 0) {                                                                                     \
            nb_ = (m_) >> p.hw_shift;                                                                              \
            const int rem_ = (m_) & ((1 << p.hw_shift) - 1);                                                       \
            yy_ = rem_ >> p.woc_shift;                                                                             \
            xx_ = rem_ & ((1 << p.woc_shift) - 1);                                                                 \
        } else {                                                                                                   \
            const int hw_ = p.HoC * p.WoC;                                                                         \
            nb_ = (m_) / hw_;                                                                                      \
            const int rem_ = (m_) - nb_ * hw_;                                                                     \
            yy_ = rem_ / p.WoC;                                                                                    \
            xx_ = rem_ - yy_ * p.WoC;                                                                              \
        }                                                                                                          \
    } while (0)

template <class T, int BM, int BN, int WM, int WN, bool TR, bool SLOW, int KG = 1, bool FI = false>
__global__ __launch_bounds__(256 * KG) void igemm_kc(const KcParams p) {
    constexpr int VE = 16 / sizeof(T);
    constexpr int VPR = BK / VE;
    constexpr int RS = BK * sizeof(T) + 16;
    constexpr int NVA = BM * VPR, NVB = BN * VPR;             // 16-byte vectors per stage
    constexpr int VA = (NVA + 255) / 256, VB = (NVB + 255) / 256;
    constexpr int WTM = BM / WM, WTN = BN / WN, TM = WTM / 32, TN = WTN / 32;
    static_assert(WM * WN == 4 && TM >= 1 && TN >= 1, "wave layout");
    // Pipeline: D chunks of both operands are in flight in registers (the layers are short reductions on small grids,
    // so the exposed global-load latency per chunk, not MFMA issue, sets the kernel time), LDS is double-buffered
    // where two stages fit in 64 KB so that one barrier per chunk is enough.
    constexpr int D = SLOW ? 1 : ((VA + VB) * 4 <= 16 ? 4 : (VA + VB) * 4 <= 32 ? 3 : 2);
    constexpr int STAGE = (BM + BN) * RS;
    constexpr int NBUF = 2 * STAGE <= 65536 ? 2 : 1;
    static_assert(KG == 1 || (BM == 64 && BN == 64 && !SLOW), "two reduction groups: 64x64 tiles only");
    static_assert(!FI || (BM == 64 && BN == 64 && !SLOW), "fused input: 64x64 tiles, whole-chunk channel counts");
    constexpr int VAL = (FI && sizeof(T) == 2) ? 2 : 1;   // 16-byte loads per A vector (fp32 source of a bf16 operand: two)
    __shared__ __attribute__((aligned(16))) char lds_all[KG * NBUF * STAGE];
    __shared__ __attribute__((aligned(16))) float ab_s[FI ? 512 : 4];   // a[Cin] | b[Cin], Cin <= 256

    KC_STAMP(0);
    KC_BLOCK(0);
    carry_store(p.carry);
    const int grp = KG == 1 ? 0 : (int)(threadIdx.x >> 8);          // reduction group of this wave
    const int tid = KG == 1 ? (int)threadIdx.x : (int)(threadIdx.x & 255);   // thread index inside the group
    const int lane = tid & 63, wave = tid >> 6;
    char* const lds = lds_all + grp * (NBUF * STAGE);
    const int wm = wave / WN, wn = wave % WN;
    const unsigned bx = p.xcd_chunk ? (blockIdx.x & 7u) * (unsigned)p.xcd_chunk + (blockIdx.x >> 3) : blockIdx.x;
    const int m0 = (bx / p.ntn) * BM, n0 = (bx % p.ntn) * BN;
    // parity classes in the order 4, 2, 2, 1 taps: the blocks with the longest reduction are dispatched first
    // (unsplit reductions only: with split-K, measured on MI355X, the light classes first is the faster order)
    const int zc = TR ? (p.splitk == 1 ? (int)(gridDim.z - 1 - blockIdx.z) : (int)blockIdx.z) : 0;
    const int py = zc >> 1, px = zc & 1;
    // valid taps of this launch slice
    int nkh, nkw;
    if (TR && p.stride == 2) { nkh = 1 + py; nkw = 1 + px; } else { nkh = p.ks; nkw = p.ks; }
    const int ntap = nkh * nkw;
    const bool s2 = p.stride == 2;
    // (free functions of values, not lambdas holding references: a closure that ends up in memory drags the whole parameter
    // struct into scratch with it)
#define tap_kh(i) tap_k<TR>(s2, py, (i))
#define tap_kw(i) tap_k<TR>(s2, px, (i))

    const int cpt = SLOW ? 1 : p.Cin / BK;
    const int nchunk = SLOW ? (ntap * p.Cin + BK - 1) / BK : ntap * cpt;
    const int per = (nchunk + p.splitk - 1) / p.splitk;
    const int c_begin = blockIdx.y * per, c_end = min(nchunk, c_begin + per);

    // the rows this thread stages
    int a_n[VA], a_y[VA], a_x[VA];
    bool a_ok[VA];
#pragma unroll
    for (int i = 0; i < VA; ++i) {
        const int r = (tid + i * 256) / VPR, m = m0 + r;
        a_ok[i] = m < p.M && (tid + i * 256) < NVA;
        const int mm = a_ok[i] ? m : 0;
        KC_SPLIT_M(mm, a_n[i], a_y[i], a_x[i]);
    }
    KC_STAMP(30);
    const int vecA = tid % VPR;  // same for every i because 256 % VPR == 0
    int b_row[VB];
    bool b_ok[VB];
#pragma unroll
    for (int j = 0; j < VB; ++j) {
        const int n = n0 + (tid + j * 256) / VPR;
        b_ok[j] = n < p.Cout && (tid + j * 256) < NVB;
        b_row[j] = b_ok[j] ? n : 0;
    }

    auto in_coord = [&](int yy, int xx, int kh, int kw, int& iy, int& ix) -> bool {
        if (TR) {
            iy = (yy * p.stride + py + p.pad - kh) / p.stride;   // exact for the valid taps of this slice
            ix = (xx * p.stride + px + p.pad - kw) / p.stride;
        } else {
            iy = yy * p.stride - p.pad + kh;
            ix = xx * p.stride - p.pad + kw;
        }
        return iy >= 0 && iy < p.Hi && ix >= 0 && ix < p.Wi;
    };

    // ---- whole-chunk channel counts: lean addressing.  Measured on MI355X (SQ counters, tools/pmc_gemm.sh) the batch-64 layers
    // were bound by instruction issue, not by MFMA or memory: ~140 instructions of index arithmetic per chunk (two scalar
    // divisions, 64-bit address products, clamps and selects) against 4 MFMAs.  Everything that does not change from chunk
    // to chunk is hoisted: per staged row one 32-bit byte offset relative to the tile's first image and a 9-bit mask of the
    // taps that fall outside the image; per chunk a wave-uniform tap offset and the channel offset in the scalar offset of
    // the buffer load.  Invalid taps / rows / output channels get bit 31 of the offset, past the descriptor: they read zeros.
    KC_STAMP(31);
    constexpr int ESZ_IN = FI ? 4 : (int)sizeof(T);
    [[maybe_unused]] unsigned a_off[VA], a_inv[VA], b_off[VB];
    // (plain scalars, not an array: a dynamically indexed private array is promoted to LDS, and the promoted form reads the
    // workgroup size from the dispatch packet in host memory -- measured 28 000 cycles at the top of every block)
    [[maybe_unused]] int dy0 = 0, dy1 = 0, dy2 = 0, dx0 = 0, dx1 = 0, dx2 = 0;
    [[maybe_unused]] __amdgpu_buffer_rsrc_t rs_in, rs_w;
    [[maybe_unused]] int ld_t = 0, ld_cc = 0;   // tap / channel chunk of the next load (advanced by KG per load)
    if constexpr (!SLOW) {
        const int img0 = p.hw_shift >= 0 ? (m0 >> p.hw_shift) : m0 / (p.HoC * p.WoC);   // first image of this tile (block-uniform)
        const size_t img_bytes = (size_t)p.Hi * p.Wi * p.Cin * ESZ_IN;
        rs_in = buf_rsrc((const char*)p.in + (size_t)img0 * img_bytes, (size_t)(p.Nb - img0) * img_bytes);
        rs_w = buf_rsrc(p.w, (size_t)p.Cout * p.ldw * sizeof(T));
        KC_STAMP(32);
        const int st_ = p.stride, pd_ = p.pad;
        if (TR) {
            dy0 = (py + pd_ - tap_kh(0)) / st_; dy1 = (py + pd_ - tap_kh(1)) / st_; dy2 = (py + pd_ - tap_kh(2)) / st_;
            dx0 = (px + pd_ - tap_kw(0)) / st_; dx1 = (px + pd_ - tap_kw(1)) / st_; dx2 = (px + pd_ - tap_kw(2)) / st_;
        } else {
            dy0 = tap_kh(0) - pd_; dy1 = tap_kh(1) - pd_; dy2 = tap_kh(2) - pd_;
            dx0 = tap_kw(0) - pd_; dx1 = tap_kw(1) - pd_; dx2 = tap_kw(2) - pd_;
        }
        KC_STAMP(33);
#pragma unroll
        for (int i = 0; i < VA; ++i) {
            const int by = TR ? a_y[i] : a_y[i] * p.stride, bx = TR ? a_x[i] : a_x[i] * p.stride;
            a_off[i] = (unsigned)((((a_n[i] - img0) * p.Hi + by) * p.Wi + bx) * p.Cin * ESZ_IN + vecA * VE * ESZ_IN);
            unsigned invh = 0, invw = 0;
            invh = outside_bit(by + dy0, p.Hi) | (outside_bit(by + dy1, p.Hi) << 1) | (outside_bit(by + dy2, p.Hi) << 2);
            invw = outside_bit(bx + dx0, p.Wi) | (outside_bit(bx + dx1, p.Wi) << 1) | (outside_bit(bx + dx2, p.Wi) << 2);
            // bit (kh * nkw + kw): a tap row outside the image invalidates its nkw taps, otherwise the columns decide (bits of tap
            // rows >= nkh land at or above ntap and are never tested)
            const unsigned roww = (1u << nkw) - 1u, colw = invw & roww;
            const unsigned inv = ((invh & 1u) ? roww : colw) | (((invh & 2u) ? roww : colw) << nkw) | (((invh & 4u) ? roww : colw) << (2 * nkw));
            a_inv[i] = a_ok[i] ? inv : 0x1FFu;
        }
#pragma unroll
        for (int j = 0; j < VB; ++j) b_off[j] = b_ok[j] ? (unsigned)((b_row[j] * p.ldw + vecA * VE) * (int)sizeof(T)) : BUF_OOB;
        KC_STAMP(34);
        const int c0 = c_begin + grp;
        ld_t = c0 / cpt;
        ld_cc = c0 - ld_t * cpt;
    }
    // the next chunk of this group (chunks are loaded in ascending order, KG apart); returns the validity bits of the A
    // vectors (fused input only: there a zero input is not a zero operand)
    auto load_next = [&](uint4 (&va)[VA * VAL], uint4 (&vb)[VB]) __attribute__((always_inline)) -> unsigned {
        unsigned mask = 0;
        const int ti = ld_t >= 2 * nkw ? 2 : ld_t >= nkw ? 1 : 0, tj = ld_t - ti * nkw;
        // (the tap offsets are linear in the tap index in every form; a select among the variables would become a load through a
        // selected address and pin them, the closure and the parameter struct to scratch memory)
        const int dy = dy0 + ti * (dy1 - dy0), dx = dx0 + tj * (dx1 - dx0);
        const unsigned tap_off = (unsigned)((dy * p.Wi + dx) * p.Cin * ESZ_IN);
        const unsigned so_a = (unsigned)(ld_cc * BK * ESZ_IN);
        const unsigned so_w = (unsigned)(((tap_kh(ti) * p.ks + tap_kw(tj)) * p.Cin + ld_cc * BK) * (int)sizeof(T));
        const unsigned sh = 31u - (unsigned)ld_t;
#pragma unroll
        for (int i = 0; i < VA; ++i) {
            const unsigned off = (a_off[i] + tap_off) | ((a_inv[i] << sh) & BUF_OOB);
            if constexpr (FI) mask |= ((a_inv[i] >> ld_t) & 1u) ? 0u : (1u << i);
            va[i * VAL] = buf_ld16(rs_in, off, so_a);
            if constexpr (VAL == 2) va[i * VAL + 1] = buf_ld16(rs_in, off + 16u, so_a);
        }
#pragma unroll
        for (int j = 0; j < VB; ++j) vb[j] = buf_ld16(rs_w, b_off[j], so_w);
        ld_cc += KG;
        while (ld_cc >= cpt) { ld_cc -= cpt; ++ld_t; }
        return mask;
    };

    // elementwise-load form (channel counts that do not fill chunks): returns the validity bits of the A vectors (bit i)
    auto load_chunk = [&](int c, uint4 (&va)[VA * VAL], uint4 (&vb)[VB]) __attribute__((always_inline)) -> unsigned {
        unsigned mask = 0;
        if constexpr (!SLOW) {
            return load_next(va, vb);
        } else {
            mask = ~0u;
            const int ktot = ntap * p.Cin;
#pragma unroll
            for (int i = 0; i < VA; ++i) {
                va[i] = make_uint4(0, 0, 0, 0);
#pragma unroll
                for (int e = 0; e < VE; ++e) {
                    const int k = c * BK + vecA * VE + e;
                    T x = (T)0;
                    if (k < ktot && a_ok[i]) {
                        const int t = k / p.Cin, ci = k - t * p.Cin;
                        const int kh = tap_kh(t / nkw), kw = tap_kw(t % nkw);
                        int iy, ix;
                        if (in_coord(a_y[i], a_x[i], kh, kw, iy, ix))
                            x = ((const T*)p.in)[((size_t)(a_n[i] * p.Hi + iy) * p.Wi + ix) * p.Cin + ci];
                    }
                    put_elem<T>(va[i], e, x);
                }
            }
#pragma unroll
            for (int j = 0; j < VB; ++j) {
                vb[j] = make_uint4(0, 0, 0, 0);
#pragma unroll
                for (int e = 0; e < VE; ++e) {
                    const int k = c * BK + vecA * VE + e;
                    T x = (T)0;
                    if (k < ktot && b_ok[j]) {
                        const int t = k / p.Cin, ci = k - t * p.Cin;
                        const int kh = tap_kh(t / nkw), kw = tap_kw(t % nkw);
                        x = ((const T*)p.w)[(size_t)b_row[j] * p.ldw + (kh * p.ks + kw) * p.Cin + ci];
                    }
                    put_elem<T>(vb[j], e, x);
                }
            }
        }
        return mask;
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    // group g reduces chunks c_begin + g, c_begin + g + KG, ...; every group runs the same number of iterations (barriers)
    uint4 va[D][VA * VAL], vb[D][VB];
    unsigned am[D];
    const int n_it = (max(c_end - c_begin, 0) + KG - 1) / KG;
    auto chunk_of = [&](int j) { return c_begin + grp + KG * j; };
    KC_STAMP(1);
#pragma unroll
    for (int d = 0; d < D; ++d)
        if (chunk_of(d) < c_end) am[d] = load_chunk(chunk_of(d), va[d], vb[d]);
    KC_STAMP(2);
    if constexpr (FI) {
        for (int i = (int)threadIdx.x; i < 2 * p.Cin; i += 256 * KG) ab_s[i] = p.in_ab[i];
        __syncthreads();
    }
    int buf = 0;
    for (int j0 = 0; j0 < n_it; j0 += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const int j = j0 + d;
            if (j < n_it) {   // block-uniform
                const bool mine = KG == 1 || chunk_of(j) < c_end;   // group-uniform
                char* ldsA = lds + buf * STAGE;
                char* ldsB = ldsA + BM * RS;
                if (mine) {
                    int cfi = 0;   // first channel of this thread's A vectors in the chunk (fused input only)
                    if constexpr (FI) { const int cc = chunk_of(j); cfi = (cc - (cc / cpt) * cpt) * BK + vecA * VE; }
#pragma unroll
                    for (int i = 0; i < VA; ++i)
                        if (NVA % 256 == 0 || (tid + i * 256) < NVA) {
                            uint4 v;
                            if constexpr (FI) v = keep_if((am[d] >> i) & 1u, bn_act_vec<T>(&va[d][i * VAL], ab_s, cfi, p.Cin));
                            else if constexpr (SLOW) v = keep_if((am[d] >> i) & 1u, va[d][i]);
                            else v = va[d][i];   // (out-of-range loads came back as zeros)
                            *(uint4*)(ldsA + ((tid + i * 256) / VPR) * RS + vecA * 16) = v;
                        }
#pragma unroll
                    for (int jj = 0; jj < VB; ++jj)
                        if (NVB % 256 == 0 || (tid + jj * 256) < NVB) *(uint4*)(ldsB + ((tid + jj * 256) / VPR) * RS + vecA * 16) =
                            SLOW ? keep_if(b_ok[jj], vb[d][jj]) : vb[d][jj];
                }
                __syncthreads();
                KC_STAMP(3 + (j < 16 ? j : 16));
                if (chunk_of(j + D) < c_end) am[d] = load_chunk(chunk_of(j + D), va[d], vb[d]);  // D chunks ahead of the MFMAs below
                if (mine) {
#pragma unroll
                    for (int kk = 0; kk < BK / 16; ++kk) {
                        Frag<T> fa[TM], fb[TN];
                        const int koff = (kk * 16 + 8 * (lane >> 5)) * (int)sizeof(T);
#pragma unroll
                        for (int a = 0; a < TM; ++a) frag_kc(fa[a], ldsA + (wm * WTM + a * 32 + (lane & 31)) * RS + koff);
#pragma unroll
                        for (int b = 0; b < TN; ++b) frag_kc(fb[b], ldsB + (wn * WTN + b * 32 + (lane & 31)) * RS + koff);
#pragma unroll
                        for (int a = 0; a < TM; ++a)
#pragma unroll
                            for (int b = 0; b < TN; ++b) mma32(acc[a][b], fa[a], fb[b]);
                    }
                }
                if (NBUF == 1) __syncthreads(); else buf ^= 1;
            }
        }
    }
    KC_STAMP(20);
    if (c_begin >= c_end && !(blockIdx.y == 0)) return;

    // epilogue: C/D map of the 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
    const bool add_bias = p.bias != nullptr && blockIdx.y == 0;
    if constexpr (BM == 64 && BN == 64) {
        // unsplit reduction: the tile goes through LDS and leaves as 16 B per lane, 256 contiguous bytes per output
        // row (4 B per lane straight from the accumulators writes the same bytes in 4x the store instructions)
        if (p.splitk == 1 && (p.Cout & 3) == 0) {
            constexpr int TS = BN + 4;   // floats per LDS row
            static_assert(BM * TS * 4 <= STAGE, "output tile must fit the staging buffer");
            float* tile = (float*)lds;   // (each reduction group stages its own partial tile)
            __syncthreads();             // every wave is done with the operand stages
#pragma unroll
            for (int r = 0; r < 16; ++r)
                tile[(wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * TS + wn * 32 + (lane & 31)] = acc[0][0][r];
            __syncthreads();
            const float* tile0 = (const float*)lds_all;
            const float* tile1 = (const float*)(lds_all + NBUF * STAGE);   // KG == 2 only
            const int ftid = (int)threadIdx.x;                             // all 256 * KG threads store the tile
            constexpr int RPP = 16 * KG, NPASS = BM / RPP;                 // rows per pass, passes
            const int c4 = (tid & 15) * 4, n = n0 + c4;
            float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
            if (add_bias && n < p.Cout) bv = *(const float4*)(p.bias + n);
            float4 ea = make_float4(1.f, 1.f, 1.f, 1.f), eb = make_float4(0.f, 0.f, 0.f, 0.f);
            if (p.ep_ab && n < p.Cout) { ea = *(const float4*)(p.ep_ab + n); eb = *(const float4*)(p.ep_ab + p.Cout + n); }
            const int red = p.red_bins ? p.red_mode : 0;   // block-uniform
            double rs[4] = {0.0, 0.0, 0.0, 0.0}, rq[4] = {0.0, 0.0, 0.0, 0.0};
            float4 mu = make_float4(0.f, 0.f, 0.f, 0.f), is = mu;
            const int rgrp = p.red_group_rows ? m0 / p.red_group_rows : 0;   // block-uniform
            if (red == 2 && n < p.Cout) { mu = *(const float4*)(p.red_mean + (size_t)rgrp * p.Cout + n); is = *(const float4*)(p.red_invstd + (size_t)rgrp * p.Cout + n); }
            float4 fa = make_float4(0.f, 0.f, 0.f, 0.f), fb = fa;
            if (red == 2 && p.red_ab && n < p.Cout) {
                fa = *(const float4*)(p.red_ab + ((size_t)rgrp * 2) * p.Cout + n);
                fb = *(const float4*)(p.red_ab + ((size_t)rgrp * 2 + 1) * p.Cout + n);
            }
#pragma unroll
            for (int i = 0; i < NPASS; ++i) {
                const int row = (ftid >> 4) + RPP * i, m = m0 + row;
                if (m >= p.M || n >= p.Cout) continue;
                size_t opix = (size_t)m;
                if (TR) {
                    int nb, yy, xx;
                    KC_SPLIT_M(m, nb, yy, xx);
                    opix = ((size_t)nb * p.Ho + (yy * p.stride + py)) * p.Wo + (xx * p.stride + px);
                }
                float4 v = *(const float4*)(tile0 + row * TS + c4);
                if constexpr (KG == 2) {
                    const float4 u = *(const float4*)(tile1 + row * TS + c4);
                    v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
                }
                v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
                if (p.ep_ab) {   // block-uniform
                    v.x = fmaf(v.x, ea.x, eb.x); v.y = fmaf(v.y, ea.y, eb.y); v.z = fmaf(v.z, ea.z, eb.z); v.w = fmaf(v.w, ea.w, eb.w);
                    v.x = v.x > 0.f ? v.x : 0.01f * v.x; v.y = v.y > 0.f ? v.y : 0.01f * v.y;
                    v.z = v.z > 0.f ? v.z : 0.01f * v.z; v.w = v.w > 0.f ? v.w : 0.01f * v.w;
                }
                if (p.out_bf16) {
                    uint2 u;
                    u.x = pack_bf16x2(v.x, v.y);
                    u.y = pack_bf16x2(v.z, v.w);
                    *(uint2*)((bf16_t*)p.out + opix * p.Cout + n) = u;
                } else {
                    *(float4*)(p.out + opix * p.Cout + n) = v;
                }
                if (red == 1) {
                    rs[0] += v.x; rs[1] += v.y; rs[2] += v.z; rs[3] += v.w;
                    rq[0] += (double)v.x * v.x; rq[1] += (double)v.y * v.y; rq[2] += (double)v.z * v.z; rq[3] += (double)v.w * v.w;
                } else if (red == 2) {
                    const size_t e = opix * p.Cout + n;
                    float4 x;
                    if (p.red_raw_bf16) {
                        const uint2 u = *(const uint2*)((const bf16_t*)p.red_raw + e);
                        x = make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16),
                                        __uint_as_float(u.y & 0xffff0000u));
                    } else {
                        x = *(const float4*)((const float*)p.red_raw + e);
                    }
                    float a[4];
                    if (p.red_ab) {   // block-uniform: the forward pass's expression, its sign is the sign of the stored activation
                        a[0] = x.x * fa.x + fb.x; a[1] = x.y * fa.y + fb.y; a[2] = x.z * fa.z + fb.z; a[3] = x.w * fa.w + fb.w;
                    } else if constexpr (sizeof(T) == 2) {
                        const uint2 u = *(const uint2*)((const bf16_t*)p.red_act + e);
                        a[0] = __uint_as_float(u.x << 16); a[1] = __uint_as_float(u.x & 0xffff0000u);
                        a[2] = __uint_as_float(u.y << 16); a[3] = __uint_as_float(u.y & 0xffff0000u);
                    } else {
                        const float4 t = *(const float4*)((const float*)p.red_act + e);
                        a[0] = t.x; a[1] = t.y; a[2] = t.z; a[3] = t.w;
                    }
                    const float d0 = v.x * (a[0] > 0.f ? 1.f : 0.01f), d1 = v.y * (a[1] > 0.f ? 1.f : 0.01f);
                    const float d2 = v.z * (a[2] > 0.f ? 1.f : 0.01f), d3 = v.w * (a[3] > 0.f ? 1.f : 0.01f);
                    rs[0] += d0; rs[1] += d1; rs[2] += d2; rs[3] += d3;
                    rq[0] += (double)d0 * ((x.x - mu.x) * is.x); rq[1] += (double)d1 * ((x.y - mu.y) * is.y);
                    rq[2] += (double)d2 * ((x.z - mu.z) * is.z); rq[3] += (double)d3 * ((x.w - mu.w) * is.w);
                }
            }
            if (red) {
                // lanes tid, tid^16, tid^32 hold the same 4 columns; then the 4 waves through LDS; 128 atomics per block
                // into one of red_nbin bins, so that a few dozen blocks at most meet on an address
                static_assert(BM * TS * 4 + 4 * KG * 2 * 64 * 8 <= NBUF * STAGE, "the sums are staged behind group 0's output tile");
                double* redsm = (double*)(lds_all + BM * TS * 4);   // [4 * KG waves][2][64]
                const int fwave = ftid >> 6;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    rs[j] += __shfl_xor(rs[j], 16, 64); rs[j] += __shfl_xor(rs[j], 32, 64);
                    rq[j] += __shfl_xor(rq[j], 16, 64); rq[j] += __shfl_xor(rq[j], 32, 64);
                }
                if (lane < 16) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) { redsm[(fwave * 2 + 0) * 64 + c4 + j] = rs[j]; redsm[(fwave * 2 + 1) * 64 + c4 + j] = rq[j]; }
                }
                __syncthreads();
                if (ftid < 128) {
                    const int qq = ftid >> 6, col = ftid & 63;
                    if (n0 + col < p.Cout) {
                        double t = (redsm[(0 * 2 + qq) * 64 + col] + redsm[(1 * 2 + qq) * 64 + col]) +
                                   (redsm[(2 * 2 + qq) * 64 + col] + redsm[(3 * 2 + qq) * 64 + col]);
                        if constexpr (KG == 2)
                            t += (redsm[(4 * 2 + qq) * 64 + col] + redsm[(5 * 2 + qq) * 64 + col]) +
                                 (redsm[(6 * 2 + qq) * 64 + col] + redsm[(7 * 2 + qq) * 64 + col]);
                        const int bin = (int)((bx / p.ntn + blockIdx.z * 5u) % (unsigned)p.red_nbin);
                        unsafeAtomicAdd(p.red_bins + (((size_t)rgrp * p.red_nbin + bin) * 2 + qq) * p.Cout + n0 + col, t);
                    }
                }
                if (red == 1 && p.fin_ab) {   // block-uniform
                    // the last block of the launch turns the sums into the layer's BatchNorm quantities, so that the consumer
                    // (favae_conv_igemm_bn_in) needs 2 * Cout floats instead of the bins
                    __shared__ int is_last;
                    wait_vm_writes();          // this thread's atomics have been performed
                    __syncthreads();
                    if (ftid == 0)
                        is_last = __hip_atomic_fetch_add(p.fin_ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x * gridDim.z - 1;
                    __syncthreads();
                    if (is_last) {
                        // 2 * Cout sums x nbin bins, read with agent-scope loads (other blocks' atomics), 8 in flight per thread:
                        // thread = (sum index qc, part of the bins); the parts meet in LDS (the staging area of the sums above)
                        constexpr int NT = 256 * KG;
                        const int C2 = 2 * p.Cout, parts = C2 >= NT ? 1 : NT / C2;
                        double* part_sm = redsm;   // [parts][C2] <= NT doubles (C2 < NT) -- all reads of redsm are behind the barrier above
                        double* tot = part_sm + NT;   // [C2] for C2 <= 512: behind the parts (LDS: BM*TS*4 + (NT + 512) * 8 <= NBUF * STAGE)
                        static_assert(BM * TS * 4 + (256 * KG + 512) * 8 <= NBUF * STAGE, "finalisation scratch behind the output tile");
                        for (int q0 = 0; q0 < C2; q0 += NT) {
                            const int qc = C2 >= NT ? q0 + ftid : ftid % C2, part = C2 >= NT ? 0 : ftid / C2;
                            double t = 0.0;
                            if (qc < C2 && part < parts) {
                                for (int b0 = part; b0 < p.red_nbin; b0 += 8 * parts) {
                                    double x[8];
#pragma unroll
                                    for (int u = 0; u < 8; ++u) {
                                        const int b = b0 + u * parts;
                                        x[u] = b < p.red_nbin ? __hip_atomic_load(p.red_bins + (size_t)b * C2 + qc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
                                    }
#pragma unroll
                                    for (int u = 0; u < 8; ++u) t += x[u];
                                }
                            }
                            if (C2 >= NT) { if (qc < C2) tot[qc] = t; }
                            else {
                                part_sm[ftid] = t;
                                __syncthreads();
                                if (ftid < C2) {
                                    double u = 0.0;
                                    for (int k = 0; k < parts; ++k) u += part_sm[k * C2 + ftid];
                                    tot[ftid] = u;
                                }
                            }
                        }
                        __syncthreads();
                        for (int c = ftid; c < p.Cout; c += NT) {
                            const double s0 = tot[c], s1 = tot[p.Cout + c];
                            const double m = s0 / p.fin_R;
                            double var = s1 / p.fin_R - m * m;
                            if (var < 0) var = 0;
                            const float is = (float)(1.0 / sqrt(var + (double)p.fin_eps));
                            const float a = is * p.fin_gamma[c];
                            p.fin_mean[c] = (float)m;
                            p.fin_invstd[c] = is;
                            p.fin_ab[c] = a;
                            p.fin_ab[p.Cout + c] = p.fin_beta[c] - (float)m * a;
                            if (p.fin_run_mean) {
                                const double unb = p.fin_R > 1 ? var * p.fin_R / (p.fin_R - 1) : var;
                                p.fin_run_mean[c] = (1.f - p.fin_mom) * p.fin_run_mean[c] + p.fin_mom * (float)m;
                                p.fin_run_var[c] = (1.f - p.fin_mom) * p.fin_run_var[c] + p.fin_mom * (float)unb;
                            }
                        }
                        if (ftid == 0) __hip_atomic_store(p.fin_ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
            }
            KC_STAMP(21);
            KC_BLOCK(1);
            return;
        }
    }
    if constexpr (KG == 2) {
        float* hand = (float*)(lds_all + NBUF * STAGE);   // [16][256] floats in group 1's stages (both groups are done with them)
        __syncthreads();
        if (grp == 1) {
#pragma unroll
            for (int r = 0; r < 16; ++r) hand[r * 256 + tid] = acc[0][0][r];
        }
        __syncthreads();
        if (grp == 1) return;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][0][r] += hand[r * 256 + tid];
    }
#pragma unroll
    for (int a = 0; a < TM; ++a) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + wm * WTM + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (m >= p.M) continue;
            size_t opix = (size_t)m;
            if (TR) {
                int nb, yy, xx;
                KC_SPLIT_M(m, nb, yy, xx);
                opix = ((size_t)nb * p.Ho + (yy * p.stride + py)) * p.Wo + (xx * p.stride + px);
            }
#pragma unroll
            for (int b = 0; b < TN; ++b) {
                const int n = n0 + wn * WTN + b * 32 + (lane & 31);
                if (n >= p.Cout) continue;
                float v = acc[a][b][r];
                if (add_bias) v += p.bias[n];
                float* dst = p.out + opix * p.Cout + n;
                if (p.out_bf16) ((bf16_t*)p.out)[opix * p.Cout + n] = f32_to_bf16(v);
                else if (p.splitk > 1) atomicAdd(dst, v);
                else *dst = v;
            }
        }
    }
    KC_BLOCK(1);
}

#undef tap_kh
#undef tap_kw
#undef KC_SPLIT_M

// ---------------------------------------------------------------------------------------------------------
// final_layer.0 forward: ConvTranspose2d(64 -> 64, 3x3, stride 2, pad 1, output_padding 1) over 32-pixel-wide rows, bf16
// (cA/celeba_deepvanilla_vae.py:63-70).  In the scatter form of igemm_kc this layer is 4 parity classes of 1 / 2 / 2 / 4 chunks per
// 64x64 tile: a block's whole reduction is one to four chunks, and the launch is all prologue and epilogue (230 us at batch 512
// for 268 MB of output).  Here a block keeps all nine 64x64 tap matrices in LDS (81 KB) for its
// lifetime and walks over units of two input rows: three rows (two + the halo row) go to LDS once, every wave takes one input row x
// 32 output channels and accumulates the four parity classes (9 tap products of 32 x 64 x 32, 36 MFMAs) from the four shifted
// views of the rows, and the four output rows of the unit leave through an LDS tile as whole 8 / 16 KB rows.  The BatchNorm sums
// stay in registers (one channel per lane) until the block changes its BatchNorm group or ends.  Same products in the same order as
// the scatter form: bit-identical output.
// ---------------------------------------------------------------------------------------------------------
struct Ct64Params {
    const char* in;        // [Nb][Hi][32][64] bf16
    const char* w;         // [64 co][9 taps][64 ci] bf16
    const float* bias;     // [64] or null
    char* out;             // [Nb][2 Hi][64][64] fp32 or bf16
    const float* ep_ab;    // a[64] | b[64]: out = lrelu(a v + b), or null
    double* red_bins;      // [groups][nbin][2][64] or null: sum v, sum v^2
    int Nb, Hi, out_bf16, red_nbin, group_images, units, upb;
    FlagCarry carry;
};
constexpr int CT_RS = 144;                          // bytes of a 64-channel bf16 row in LDS (128 + 16)
constexpr int CT_W_BYTES = 9 * 64 * CT_RS;          // 82 944
constexpr int CT_A_ROW = 33 * CT_RS;                // 32 pixels + a zero pixel
constexpr int CT_A_BYTES = 3 * CT_A_ROW;            // one unit: two input rows + the halo row
constexpr int CT_O_PX = 64 * 4 + 16;                // bytes per output pixel in the staging tile
constexpr int CT_O_BYTES = 2 * 64 * CT_O_PX;        // two output rows
constexpr int CT_LDS = CT_W_BYTES + 2 * CT_A_BYTES + CT_O_BYTES;   // 146 272

__global__ __launch_bounds__(256) void convt64_fwd_kernel(const Ct64Params p) {
    extern __shared__ __attribute__((aligned(16))) char ct_lds[];
    char* const Ws = ct_lds;
    char* const As = ct_lds + CT_W_BYTES;
    char* const Os = As + 2 * CT_A_BYTES;
    carry_store(p.carry);
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int rsel = wave >> 1, nt = wave & 1;          // input row of the unit, half of the output channels
    const int u0 = (int)blockIdx.x * p.upb, u1 = min(p.units, u0 + p.upb);
    if (u0 >= u1) return;
    const int half = p.Hi >> 1, Ho = 2 * p.Hi;
    for (int q = tid; q < 64 * 72; q += 256) {          // global [co][tap][8 pieces] -> LDS [tap][co][row of 144 B]
        const int co = q / 72, rem = q - co * 72, tap = rem >> 3, pc = rem & 7;
        *(uint4*)(Ws + (tap * 64 + co) * CT_RS + pc * 16) = *(const uint4*)(p.w + (size_t)q * 16);
    }
    if (tid < 48) {                                     // the pixel right of the row (x = 32) reads as zero
        const int b = tid / 24, r = (tid % 24) >> 3, pc = tid & 7;
        *(uint4*)(As + b * CT_A_BYTES + r * CT_A_ROW + 32 * CT_RS + pc * 16) = make_uint4(0, 0, 0, 0);
    }
    const int lpx = tid >> 3, lpc = tid & 7;            // pixel / 16-byte piece this thread stages
    auto load_unit = [&](int u, uint4 (&v)[3]) __attribute__((always_inline)) {
        const int n = u / half, j = u - n * half;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int y = 2 * j + i;
            v[i] = y < p.Hi ? *(const uint4*)(p.in + ((((size_t)n * p.Hi + y) * 32 + lpx) * 64) * 2 + lpc * 16) : make_uint4(0, 0, 0, 0);
        }
    };
    auto store_unit = [&](int buf, const uint4 (&v)[3]) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 3; ++i) *(uint4*)(As + buf * CT_A_BYTES + i * CT_A_ROW + lpx * CT_RS + lpc * 16) = v[i];
    };
    uint4 nx[3];
    load_unit(u0, nx);
    store_unit(0, nx);
    if (u0 + 1 < u1) load_unit(u0 + 1, nx);
    const int co = nt * 32 + (lane & 31);
    // epilogue roles: a thread stores the 16-byte pieces (4 channels c4 .. c4+3, fixed for its lifetime) of pixels (tid >> 4) + 16 i
    const int c4 = (tid & 15) * 4;
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f), ea = make_float4(1.f, 1.f, 1.f, 1.f), eb = bv;
    if (p.bias) bv = *(const float4*)(p.bias + c4);
    if (p.ep_ab) { ea = *(const float4*)(p.ep_ab + c4); eb = *(const float4*)(p.ep_ab + 64 + c4); }
    const int osz = p.out_bf16 ? 2 : 4;
    // staging tile: fp32, [2 output rows][64 pixels][64 channels + pad]; the accumulator of pixel x, register r goes to pixel 2 x + px
    char* const st_w = Os + (rsel * 64 + 8 * (lane >> 5)) * CT_O_PX + co * 4;          // + (2 (r & 3) + 16 (r >> 2) + px) * CT_O_PX
    const char* const st_r = Os + (tid >> 4) * CT_O_PX + c4 * 4;                         // + 16 i * CT_O_PX
    const size_t g_thr = ((size_t)(tid >> 4) * 64 + c4) * (size_t)osz;                   // this thread's offset inside an output row
    double rs[4] = {0.0, 0.0, 0.0, 0.0}, rq[4] = {0.0, 0.0, 0.0, 0.0};
    int cur_grp = -1;
    auto flush = [&]() __attribute__((always_inline)) {
        if (!p.red_bins || cur_grp < 0) return;
        double* dst = p.red_bins + (((size_t)cur_grp * p.red_nbin + (int)(blockIdx.x % (unsigned)p.red_nbin)) * 2) * 64 + c4;
#pragma unroll
        for (int k = 0; k < 4; ++k) {     // lanes l, l ^ 16, l ^ 32 hold the same four channels
            double a = rs[k], b = rq[k];
            a += __shfl_xor(a, 16, 64); a += __shfl_xor(a, 32, 64);
            b += __shfl_xor(b, 16, 64); b += __shfl_xor(b, 32, 64);
            if (lane < 16) { unsafeAtomicAdd(dst + k, a); unsafeAtomicAdd(dst + 64 + k, b); }
            rs[k] = rq[k] = 0.0;
        }
    };
    __syncthreads();
    for (int u = u0; u < u1; ++u) {
        const int buf = (u - u0) & 1;
        const char* A = As + buf * CT_A_BYTES;
        f32x16 acc[4];
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
        // the four shifted views (dy, dx) of the rows in the order (1,1), (1,0), (0,1), (0,0): the tap order of the scatter form
#pragma unroll
        for (int cb = 3; cb >= 0; --cb) {
            const int dy = cb >> 1, dx = cb & 1;
            const char* arow = A + (rsel + dy) * CT_A_ROW + ((lane & 31) + dx) * CT_RS + 16 * (lane >> 5);
            Frag<bf16_t> fa[4];
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) frag_kc(fa[kk], arow + kk * 32);
#pragma unroll
            for (int py = dy; py <= 1; ++py)
#pragma unroll
                for (int px = dx; px <= 1; ++px) {
                    const int kh = py == 0 ? 1 : (dy ? 0 : 2), kw = px == 0 ? 1 : (dx ? 0 : 2);
                    const char* brow = Ws + ((kh * 3 + kw) * 64 + co) * CT_RS + 16 * (lane >> 5);
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) {
                        Frag<bf16_t> fb;
                        frag_kc(fb, brow + kk * 32);
                        mma32(acc[py * 2 + px], fa[kk], fb);
                    }
                }
        }
        const int n = u / half, j = u - n * half;
        if (p.red_bins) {   // block-uniform
            const int grp = p.group_images > 0 ? n / p.group_images : 0;
            if (grp != cur_grp) { flush(); cur_grp = grp; }
        }
#pragma unroll
        for (int ph = 0; ph < 2; ++ph) {                // output rows of parity ph of both input rows
            if (ph) __syncthreads();                    // the readers of the first half are done with the tile
#pragma unroll
            for (int px = 0; px < 2; ++px)
#pragma unroll
                for (int r = 0; r < 16; ++r) *(float*)(st_w + (2 * (r & 3) + 16 * (r >> 2) + px) * CT_O_PX) = acc[ph * 2 + px][r];
            __syncthreads();
            // output rows 2 (2 j + rr) + ph, rr = 0, 1: 64 pixels x 64 channels contiguous
            char* const g_row = p.out + ((((size_t)n * Ho + 4 * j + ph) * 64) * 64) * (size_t)osz + g_thr;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int rr = i >> 2;
                float4 v = *(const float4*)(st_r + 16 * i * CT_O_PX);
                v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
                if (p.red_bins) {
                    rs[0] += v.x; rs[1] += v.y; rs[2] += v.z; rs[3] += v.w;
                    rq[0] += (double)v.x * v.x; rq[1] += (double)v.y * v.y; rq[2] += (double)v.z * v.z; rq[3] += (double)v.w * v.w;
                }
                if (p.ep_ab) {
                    v.x = fmaf(v.x, ea.x, eb.x); v.y = fmaf(v.y, ea.y, eb.y); v.z = fmaf(v.z, ea.z, eb.z); v.w = fmaf(v.w, ea.w, eb.w);
                    v.x = v.x > 0.f ? v.x : 0.01f * v.x; v.y = v.y > 0.f ? v.y : 0.01f * v.y;
                    v.z = v.z > 0.f ? v.z : 0.01f * v.z; v.w = v.w > 0.f ? v.w : 0.01f * v.w;
                }
                // pixel (tid >> 4) + 16 (i & 3) of output row rr: rows are 2 output rows apart
                char* const g = g_row + ((size_t)rr * 2 * 64 * 64 + (size_t)(16 * (i & 3)) * 64) * (size_t)osz;
                if (p.out_bf16) {
                    uint2 o;
                    o.x = pack_bf16x2(v.x, v.y);
                    o.y = pack_bf16x2(v.z, v.w);
                    *(uint2*)g = o;
                } else {
                    *(float4*)g = v;
                }
            }
        }
        if (u + 1 < u1) {
            store_unit(buf ^ 1, nx);
            if (u + 2 < u1) load_unit(u + 2, nx);
        }
        __syncthreads();
    }
    flush();
}

int convt64_fwd_launch(const void* in, const void* w, const float* bias, void* out, int Nb, int Hi, int out_bf16, const float* ep_ab,
                       double* bins, int nbin, int group_images, hipStream_t st) {
    static bool attr_set[64] = {false};
    int dev = 0;
    FAVAE_HIP(hipGetDevice(&dev));
    FAVAE_CHECK_ARG(dev >= 0 && dev < 64, "favae_conv_igemm: device index %d", dev);
    if (!attr_set[dev]) {
        FAVAE_HIP(hipFuncSetAttribute((const void*)convt64_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, CT_LDS));
        attr_set[dev] = true;
    }
    Ct64Params p;
    p.in = (const char*)in; p.w = (const char*)w; p.bias = bias; p.out = (char*)out; p.ep_ab = ep_ab;
    p.red_bins = bins; p.red_nbin = bins ? nbin : 1; p.group_images = group_images;
    p.Nb = Nb; p.Hi = Hi; p.out_bf16 = out_bf16;
    p.units = Nb * (Hi / 2);
    static const int blocks = getenv("FAVAE_CT64_BLOCKS") ? atoi(getenv("FAVAE_CT64_BLOCKS")) : 256;
    p.upb = cdiv(p.units, std::max(1, std::min(blocks, p.units)));
    p.carry = favae_take_flag_carry();
    ProfScope prof(6, 2.0 * Nb * (2.0 * Hi) * 64 * 64.0 * 2.25 * 64, st);
    hipLaunchKernelGGL(convt64_fwd_kernel, dim3(cdiv(p.units, p.upb)), dim3(256), CT_LDS, st, p);
    FAVAE_LAUNCH_CHECK();
    return FAVAE_OK;
}

// ---------------------------------------------------------------------------------------------------------
// pixel-major implicit GEMM (weight gradients)
// ---------------------------------------------------------------------------------------------------------
struct PmParams {
    const void* rp;
    const void* cp;
    float* dw;
    int Rc, Cc, r_large, c_large;
    int Nb, Hs, Ws, Hl, Wl, ks, stride, pad, splitk, ntn, K;
    int overwrite;  // unsplit reductions store instead of adding (the caller guarantees dw == 0: splitk < 0)
    int Rst, Cst;   // rows / columns of dw that exist (<= Rc, Cc: operands padded to whole vectors, see favae_conv_wgrad_folded)
    FlagCarry carry;   // dependency words of the launch stream, stored by the first block (common.h)
    SlabRider rider;   // a small reduction for the first blocks (common.h); partial == NULL: none
    int xcd_units;     // > 0: XCD-contiguous work units (see the kernel), = blocks / 8
    int hw_shift, ws_shift;   // log2 of Hs*Ws and Ws when both are powers of two (every layer of the model), else -1
    // ADAM instantiation (unsplit 64x64 launches whose Rst, Cst are multiples of 64): the tile is not stored as a gradient; the
    // optimiser step of its weights is taken in the epilogue, with the arithmetic of adam_kernel (vae_kernels.hip), and the dtype copy
    // ad_wc [Rst][taps][Cst] and the transpose ad_wt [Cst][taps][Rst] (either may be NULL) are written from the fresh weights.
    float *ad_p, *ad_m, *ad_v;
    void *ad_wc, *ad_wt;
    float ad_b1, ad_b2, ad_eps, ad_step, ad_bc2s;
    // CAB instantiation (bf16, lean, 32x128 tiles, neither operand gathered, K a multiple of 64): the column operand is the conv output in
    // front of a BatchNorm(train) + LeakyReLU and the activation lrelu(x * a + b), rounded to bf16 as the BatchNorm kernels store it, is
    // formed when a chunk goes to LDS.  c_ab [groups][2][Cc] = (a | b) per BatchNorm group of c_ab_group_pix reduction elements (0: one group).
    const float* c_ab = nullptr;
    int c_ab_group_pix = 0;
};

// The small reduction a weight-gradient launch carries for the step (SlabRider, common.h): 8 channels x 32 slab parts per block, the
// parts added in LDS in a fixed order.  `lds`: 2 KB of the block's LDS, free again on return; `lin`: the block's linear index.
__device__ __forceinline__ void pm_rider(const SlabRider& rider, char* lds, int tid, unsigned lin) {
    if (!rider.partial) return;                  // uniform
    if (lin * 8 >= (unsigned)rider.C) return;    // block-uniform
    double* rs = (double*)lds;                   // [32][8]
    const int ch = tid & 7, part = tid >> 3, c = (int)lin * 8 + ch;
    double t = 0.0;
    if (c < rider.C) {
        for (int k0 = part; k0 < rider.nslab; k0 += 32 * 8) {   // 8 independent loads in flight
            double x[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) { const int k = k0 + 32 * q; x[q] = k < rider.nslab ? rider.partial[(size_t)k * rider.C + c] : 0.0; }
#pragma unroll
            for (int q = 0; q < 8; ++q) t += x[q];
        }
    }
    rs[part * 8 + ch] = t;
    __syncthreads();
    if (tid < 8 && c < rider.C) {
        double u = 0.0;
#pragma unroll 8
        for (int k = 0; k < 32; ++k) u += rs[k * 8 + tid];
        if (!rider.out0) rider.out[c] = (float)u;
        else if (c == 0) *rider.out0 = rider.off0 + rider.scale0 * u;
        else if (c - 1 < rider.nout) rider.out[c - 1] = (float)u;
    }
    __syncthreads();   // (the caller reuses the LDS)
}

// LEAN: both channel counts fill 16-byte vectors and both operands are below 2 GB.  The operands come through buffer
// descriptors: the small-grid operand's descriptor is moved to the chunk (scalar arithmetic), so its per-lane offset is a
// constant and the tail of the reduction falls off the end of the descriptor; the gathered operand's offset is rebuilt per chunk
// from shifts (power-of-two feature maps) and halo pixels get an out-of-range offset.  Out-of-range loads return zeros: no
// select per vector.  (The general form below spent ~100 instructions per vector on two integer divisions, clamps and 64-bit
// products, against two MFMAs per chunk and wave: issue-bound, like igemm_kc before its addressing was hoisted.)
template <class T, int BM, int BN, int WM, int WN, bool LEAN, bool ADAM = false, bool CAB = false>
__global__ __launch_bounds__(256) void igemm_pm(const PmParams p) {
    static_assert(!CAB || (LEAN && sizeof(T) == 2 && 256 % (BN / 8) == 0), "operand-load BatchNorm: bf16, lean loads, a fixed channel piece per thread");
    constexpr int VE = 16 / sizeof(T);
    constexpr int PADB = sizeof(T) == 2 ? 0 : 16;   // bf16: swizzled rows without padding (frag_pm); fp32: 16 bytes of padding
    constexpr int RSR = BM * sizeof(T) + PADB, RSC = BN * sizeof(T) + PADB;
    constexpr int VPR_R = BM / VE, VPR_C = BN / VE;
    constexpr int NVR = BK * VPR_R, NVC = BK * VPR_C;
    constexpr int VR = (NVR + 255) / 256, VC = (NVC + 255) / 256;
    constexpr int WTM = BM / WM, WTN = BN / WN, TM = WTM / 32, TN = WTN / 32;
    static_assert(WM * WN == 4 && TM >= 1 && TN >= 1, "wave layout");
    constexpr int D = (VR + VC) * 4 <= 16 ? 4 : (VR + VC) * 4 <= 32 ? 3 : 2;   // chunks in flight (see igemm_kc)
    constexpr int STAGE = BK * (RSR + RSC);
    constexpr int NBUF = 2 * STAGE <= 65536 ? 2 : 1;
    __shared__ __attribute__((aligned(16))) char lds[NBUF * STAGE];

    carry_store(p.carry);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    pm_rider(p.rider, lds, tid, blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z));
    const int wm = wave / WN, wn = wave % WN;
    // Blocks are dealt to the 8 XCDs round-robin in launch order.  With xcd_units > 0 block b takes work unit
    // (b % 8) * xcd_units + b / 8, and the units are ordered tap-fastest: the nine taps of one (tile, slice) read the same pixels of
    // both operands, shifted, and now meet in one L2 instead of up to eight.
    unsigned bxx = blockIdx.x, byy = blockIdx.y, bzz = blockIdx.z;
    if (p.xcd_units) {
        const unsigned lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
        const unsigned w = (lin & 7u) * (unsigned)p.xcd_units + (lin >> 3);
        bzz = w % gridDim.z;
        const unsigned t = w / gridDim.z;
        bxx = t % gridDim.x;
        byy = t / gridDim.x;
    }
    const int r0 = (bxx / p.ntn) * BM, c0 = (bxx % p.ntn) * BN;
    const int tap = bzz, kh = tap / p.ks, kw = tap % p.ks;
    const int nchunk = (p.K + BK - 1) / BK;
    const int per = (nchunk + p.splitk - 1) / p.splitk;
    const int c_begin = byy * per, c_end = min(nchunk, c_begin + per);
    if (c_begin >= c_end) return;

    // branch-free for whole-vector channel counts: clamped address, validity returned separately and applied when the
    // registers go to LDS (tail pixels, halo, tail channels)
    auto load_vec = [&](const void* base, int C, bool large, int ch0, int v, int vpr, int chunk, bool& valid) -> uint4 {
        const int prow = v / vpr, cv = v - prow * vpr;
        const int pix = chunk * BK + prow;
        bool ok = pix < p.K && prow < BK;
        const int pc = min(pix, p.K - 1);
        size_t spix = (size_t)pc;
        if (large) {
            const int hw = p.Hs * p.Ws;
            const int n = pc / hw, rem = pc - n * hw, h = rem / p.Ws, w = rem - h * p.Ws;
            int y = h * p.stride - p.pad + kh, x = w * p.stride - p.pad + kw;
            ok = ok && y >= 0 && y < p.Hl && x >= 0 && x < p.Wl;
            y = min(max(y, 0), p.Hl - 1);
            x = min(max(x, 0), p.Wl - 1);
            spix = ((size_t)n * p.Hl + y) * p.Wl + x;
        }
        const int ch = ch0 + cv * VE;
        uint4 out = make_uint4(0, 0, 0, 0);
        if (C % VE == 0) {
            const int chc = min(ch, C - VE);
            out = *(const uint4*)((const T*)base + spix * C + chc);
            valid = ok && ch < C;
        } else {
            valid = true;
            if (ok) {
#pragma unroll
                for (int e = 0; e < VE; ++e)
                    if (ch + e < C) put_elem<T>(out, e, ((const T*)base)[spix * C + ch + e]);
            }
        }
        return out;
    };

    auto load_chunk = [&](int chunk, uint4 (&vr)[VR], uint4 (&vc)[VC]) -> unsigned {
        unsigned mask = 0;
#pragma unroll
        for (int i = 0; i < VR; ++i) {
            bool ok;
            vr[i] = load_vec(p.rp, p.Rc, p.r_large, r0, tid + i * 256, VPR_R, chunk, ok);
            mask |= ok ? (1u << i) : 0u;
        }
#pragma unroll
        for (int i = 0; i < VC; ++i) {
            bool ok;
            vc[i] = load_vec(p.cp, p.Cc, p.c_large, c0, tid + i * 256, VPR_C, chunk, ok);
            mask |= ok ? (1u << (16 + i)) : 0u;
        }
        return mask;
    };

    // ---- lean form ----
    [[maybe_unused]] unsigned r_off[VR], c_off[VC];   // small-grid operand: (row in chunk * C + channel) bytes; gathered: channel bytes
    [[maybe_unused]] int r_prow[VR], c_prow[VC];
    if constexpr (LEAN) {
#pragma unroll
        for (int i = 0; i < VR; ++i) {
            const int v = tid + i * 256, prow = v / VPR_R, ch = r0 + (v - prow * VPR_R) * VE;
            r_prow[i] = prow;
            r_off[i] = (v < NVR && ch < p.Rc) ? (unsigned)(((p.r_large ? 0 : prow * p.Rc) + ch) * (int)sizeof(T)) : BUF_OOB;
        }
#pragma unroll
        for (int i = 0; i < VC; ++i) {
            const int v = tid + i * 256, prow = v / VPR_C, ch = c0 + (v - prow * VPR_C) * VE;
            c_prow[i] = prow;
            c_off[i] = (v < NVC && ch < p.Cc) ? (unsigned)(((p.c_large ? 0 : prow * p.Cc) + ch) * (int)sizeof(T)) : BUF_OOB;
        }
    }
    // byte offset of the gathered pixel that small-grid pixel `pix` meets at this launch's tap (bit 31: outside the image / past the end)
    auto gathered = [&](int pix, int C) __attribute__((always_inline)) -> unsigned {
        int n, h, w;
        if (p.hw_shift >= 0) {   // uniform
            n = pix >> p.hw_shift;
            const int rem = pix & ((1 << p.hw_shift) - 1);
            h = rem >> p.ws_shift;
            w = rem & ((1 << p.ws_shift) - 1);
        } else {
            const int hw = p.Hs * p.Ws;
            n = pix / hw;
            const int rem = pix - n * hw;
            h = rem / p.Ws;
            w = rem - h * p.Ws;
        }
        const int y = h * p.stride - p.pad + kh, x = w * p.stride - p.pad + kw;
        const bool ok = pix < p.K && (unsigned)y < (unsigned)p.Hl && (unsigned)x < (unsigned)p.Wl;
        return ok ? (unsigned)((((n * p.Hl + y) * p.Wl + x) * C) * (int)sizeof(T)) : BUF_OOB;
    };
    auto load_lean = [&](int chunk, uint4 (&vr)[VR], uint4 (&vc)[VC]) __attribute__((always_inline)) {
        const int pix0 = chunk * BK;
        if (p.r_large) {   // uniform
            const __amdgpu_buffer_rsrc_t rs = buf_rsrc(p.rp, (size_t)p.Nb * p.Hl * p.Wl * p.Rc * sizeof(T));
#pragma unroll
            for (int i = 0; i < VR; ++i) vr[i] = buf_ld16(rs, buf_off_add(gathered(pix0 + r_prow[i], p.Rc), r_off[i]), 0u);
        } else {
            const __amdgpu_buffer_rsrc_t rs = buf_rsrc((const char*)p.rp + (size_t)pix0 * p.Rc * sizeof(T), (size_t)(p.K - pix0) * p.Rc * sizeof(T));
#pragma unroll
            for (int i = 0; i < VR; ++i) vr[i] = buf_ld16(rs, r_off[i], 0u);
        }
        if (p.c_large) {
            const __amdgpu_buffer_rsrc_t rs = buf_rsrc(p.cp, (size_t)p.Nb * p.Hl * p.Wl * p.Cc * sizeof(T));
#pragma unroll
            for (int i = 0; i < VC; ++i) vc[i] = buf_ld16(rs, buf_off_add(gathered(pix0 + c_prow[i], p.Cc), c_off[i]), 0u);
        } else {
            const __amdgpu_buffer_rsrc_t rs = buf_rsrc((const char*)p.cp + (size_t)pix0 * p.Cc * sizeof(T), (size_t)(p.K - pix0) * p.Cc * sizeof(T));
#pragma unroll
            for (int i = 0; i < VC; ++i) vc[i] = buf_ld16(rs, c_off[i], 0u);
        }
    };
    auto load_any = [&](int chunk, uint4 (&vr)[VR], uint4 (&vc)[VC]) __attribute__((always_inline)) -> unsigned {
        if constexpr (LEAN) { load_lean(chunk, vr, vc); return ~0u; }
        else return load_chunk(chunk, vr, vc);
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    [[maybe_unused]] float cab_a[8], cab_b[8];
    [[maybe_unused]] int cab_g = -1;
    [[maybe_unused]] const int cab_ch = c0 + (tid % VPR_C) * VE;          // (256 % VPR_C == 0: the same piece for every vector of the thread)
    [[maybe_unused]] const bool cab_live = cab_ch < p.Cc;
    if constexpr (CAB) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { cab_a[j] = 0.f; cab_b[j] = 0.f; }
    }

    uint4 vr[D][VR], vc[D][VC];
    unsigned vm[D];
#pragma unroll
    for (int d = 0; d < D; ++d)
        if (c_begin + d < c_end) vm[d] = load_any(c_begin + d, vr[d], vc[d]);
    int buf = 0;
    for (int c = c_begin; c < c_end; c += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const int cc = c + d;
            if (cc < c_end) {   // block-uniform
                char* ldsR = lds + buf * STAGE;
                char* ldsC = ldsR + BK * RSR;
                if constexpr (CAB) {   // (a, b) of this thread's 8 channels for the BatchNorm group of chunk cc (chunks never straddle groups)
                    const int g = p.c_ab_group_pix > 0 ? (cc * BK) / p.c_ab_group_pix : 0;
                    if (g != cab_g) {   // block-uniform
                        cab_g = g;
                        if (cab_live) {
                            const float* ab = p.c_ab + (size_t)g * 2 * p.Cc + cab_ch;
#pragma unroll
                            for (int j = 0; j < 8; ++j) { cab_a[j] = ab[j]; cab_b[j] = ab[p.Cc + j]; }
                        }
                    }
                }
#pragma unroll
                for (int i = 0; i < VR; ++i) {
                    const int v = tid + i * 256;
                    if (NVR % 256 == 0 || v < NVR) *(uint4*)(ldsR + pm_off<T>(v / VPR_R, v % VPR_R, RSR)) = LEAN ? vr[d][i] : keep_if((vm[d] >> i) & 1u, vr[d][i]);
                }
#pragma unroll
                for (int i = 0; i < VC; ++i) {
                    const int v = tid + i * 256;
                    if constexpr (CAB) {
                        uint4 t = vc[d][i];
                        if (cab_live) {   // (padding channels beyond Cc stay zero)
                            unsigned w4[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
                            for (int k = 0; k < 4; ++k) {
                                float y0 = __uint_as_float(w4[k] << 16) * cab_a[2 * k] + cab_b[2 * k];
                                float y1 = __uint_as_float(w4[k] & 0xffff0000u) * cab_a[2 * k + 1] + cab_b[2 * k + 1];
                                y0 = y0 > 0.f ? y0 : 0.01f * y0;
                                y1 = y1 > 0.f ? y1 : 0.01f * y1;
                                w4[k] = pack_bf16x2(y0, y1);
                            }
                            t = make_uint4(w4[0], w4[1], w4[2], w4[3]);
                        }
                        if (NVC % 256 == 0 || v < NVC) *(uint4*)(ldsC + pm_off<T>(v / VPR_C, v % VPR_C, RSC)) = t;
                    } else {
                        if (NVC % 256 == 0 || v < NVC) *(uint4*)(ldsC + pm_off<T>(v / VPR_C, v % VPR_C, RSC)) = LEAN ? vc[d][i] : keep_if((vm[d] >> (16 + i)) & 1u, vc[d][i]);
                    }
                }
                __syncthreads();
                if (cc + D < c_end) vm[d] = load_any(cc + D, vr[d], vc[d]);
#pragma unroll
                for (int kk = 0; kk < BK / 16; ++kk) {
                    Frag<T> fa[TM], fb[TN];
#pragma unroll
                    for (int a = 0; a < TM; ++a) frag_pm(fa[a], ldsR, RSR, kk * 16, wm * WTM + a * 32, lane);
#pragma unroll
                    for (int b = 0; b < TN; ++b) frag_pm(fb[b], ldsC, RSC, kk * 16, wn * WTN + b * 32, lane);
#pragma unroll
                    for (int a = 0; a < TM; ++a)
#pragma unroll
                        for (int b = 0; b < TN; ++b) mma32(acc[a][b], fa[a], fb[b]);
                }
                if (NBUF == 1) __syncthreads(); else buf ^= 1;
            }
        }
    }

    const int ntaps = p.ks * p.ks;
    if constexpr (ADAM) {
        static_assert(BM == 64 && BN == 64 && TM == 1 && TN == 1, "Adam in the epilogue: 64x64 tiles");
        // Adam on the tile (torch.optim.Adam, cA/celeba_deepvanilla_vae.py:131,250; same expressions as adam_kernel), then the two
        // dtype layouts of the fresh weights: wc in place, wt through an LDS transpose so that its rows leave as 128 contiguous bytes
        constexpr int TRS = 64 + 16 / (int)sizeof(T);   // elements per LDS row (+16 B against bank conflicts)
        static_assert(64 * TRS * (int)sizeof(T) <= STAGE, "the transposed tile must fit a stage");
        __syncthreads();                                // every wave is done with the operand stages
        T* tr = (T*)lds;
        const int col_l = wn * 32 + (lane & 31), col = c0 + col_l;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row_l = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), row = r0 + row_l;
            const size_t idx = ((size_t)row * ntaps + tap) * p.Cst + col;
            const float g = acc[0][0][r];
            float P = p.ad_p[idx], M = p.ad_m[idx], V = p.ad_v[idx];
            adam_update(P, g, M, V, p.ad_b1, p.ad_b2, p.ad_eps, p.ad_step, p.ad_bc2s);
            p.ad_p[idx] = P; p.ad_m[idx] = M; p.ad_v[idx] = V;
            if constexpr (sizeof(T) == 2) {
                const bf16_t q = f32_to_bf16(P);
                if (p.ad_wc) ((bf16_t*)p.ad_wc)[idx] = q;
                tr[col_l * TRS + row_l] = q;
            } else {
                if (p.ad_wc) ((float*)p.ad_wc)[idx] = P;
                tr[col_l * TRS + row_l] = P;
            }
        }
        __syncthreads();
        if (p.ad_wt) {   // block-uniform: 64 columns x 4 pieces of 16 rows per thread pass
            const int cl = tid >> 2, piece = tid & 3;
            T* dst = (T*)p.ad_wt + ((size_t)(c0 + cl) * ntaps + tap) * p.Rst + r0 + piece * 16;
            const uint4* src = (const uint4*)(tr + cl * TRS + piece * 16);
            constexpr int NV = 16 * (int)sizeof(T) / 16;   // 16-byte vectors per piece
#pragma unroll
            for (int i = 0; i < NV; ++i) ((uint4*)dst)[i] = src[i];
        }
        return;
    }
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = r0 + wm * WTM + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (row >= p.Rst) continue;
#pragma unroll
            for (int b = 0; b < TN; ++b) {
                const int col = c0 + wn * WTN + b * 32 + (lane & 31);
                if (col < p.Cst) {
                    // one block per output element when the reduction is not split: add without an atomic (the
                    // gradient arena is zeroed per step; chip-wide float atomics run at ~1.3 TB/s, plain stores at 6)
                    float* dst = p.dw + ((size_t)row * ntaps + tap) * p.Cst + col;
                    if (p.splitk > 1) atomicAdd(dst, acc[a][b][r]); else if (p.overwrite) *dst = acc[a][b][r]; else *dst += acc[a][b][r];
                }
            }
        }
}

// ---------------------------------------------------------------------------------------------------------
// weight gradient of a stride-2 3x3 layer, all nine taps from ONE pass over the operands (bf16, channel tiles of 64 x 64)
// ---------------------------------------------------------------------------------------------------------
// igemm_pm gives every tap a block of its own: nine blocks read the same 64 small-grid pixels and, shifted, the same large-grid
// pixels: 16 KB from L2 per 64-pixel chunk and tap for 4 MFMAs per wave (32 FLOP per byte).  Beyond batch 64 the operands no longer
// stay in the caches between the taps and the launch runs at the rate of those re-reads (final_layer.0 at 512 images: 1.2 GB
// fetched for 335 MB of operands, 116 TFLOP/s).  Here a block keeps nine accumulator tiles (144 registers per lane) and reads a
// chunk once: the 64 small-grid pixels (rows iy0 .. iy0 + NR - 1 of one image, NR = 64 / Ws) and the patch of the large grid they
// meet, rows 2 iy0 - 1 .. 2 (iy0 + NR - 1) + 1, columns -1 .. 2 Ws - 1: 36 MFMAs per wave and chunk on 48 KB.
// Patch image in LDS: per large-grid row two planes (even / odd columns: tap kx walks ONE plane at unit stride), a plane =
// Ws + 2 slots of 128 bytes (64 channels), slot s = (x >> 1) + 1; odd-plane slot 0 is column -1 (zeros, written once).  The two
// 64-byte halves of a slot are swapped where bit 1 of s is set, which puts the four consecutive slots one ds_read_b64_tr_b16
// half-wave touches on the four quarters of the 256-byte bank row (same reasoning as pm_swz).
template <int LW>
struct W9 {
    static constexpr int WS = 1 << LW, NR = 64 >> LW, NPR = 2 * NR + 1, WL = 2 * WS, PS = WS + 2;
    static constexpr int PLANE = PS * 128, PATCH = NPR * 2 * PLANE, SBYTES = 64 * 128, STAGE = SBYTES + PATCH;
    static constexpr int NVL = NPR * WL * 8, VL = (NVL + 255) / 256;
};

template <int LW, bool R_LARGE>
__global__ __launch_bounds__(256) void wgrad9_kernel(const PmParams p) {
    using G = W9<LW>;
    using T = bf16_t;
    extern __shared__ __attribute__((aligned(16))) char lds9[];
    carry_store(p.carry);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    const unsigned lin = blockIdx.x + gridDim.x * blockIdx.y;
    pm_rider(p.rider, lds9, tid, lin);
    // column -1 of every patch row, both stages
    for (int i = tid; i < 2 * G::NPR * 8; i += 256) {
        const int st = i / (G::NPR * 8), pr = (i >> 3) % G::NPR, v = i & 7;
        *(uint4*)(lds9 + st * G::STAGE + G::SBYTES + (pr * 2 + 1) * G::PLANE + v * 16) = make_uint4(0, 0, 0, 0);
    }
    unsigned bx = blockIdx.x, by = blockIdx.y;
    if (p.xcd_units) {   // the tiles of one slice of the reduction read the same pixels: one XCD, one L2 (see igemm_pm)
        const unsigned w = (lin & 7u) * (unsigned)p.xcd_units + (lin >> 3);
        bx = w % gridDim.x;
        by = w / gridDim.x;
    }
    const int r0 = (int)(bx / p.ntn) * 64, c0 = (int)(bx % p.ntn) * 64;
    const int nchunk = p.K >> 6;
    const int per = (nchunk + p.splitk - 1) / p.splitk;
    const int c_begin = (int)by * per, c_end = min(nchunk, c_begin + per);
    if (c_begin >= c_end) return;

    const void* sp = R_LARGE ? p.cp : p.rp;
    const void* lp = R_LARGE ? p.rp : p.cp;
    const int Cs = R_LARGE ? p.Cc : p.Rc, Cl = R_LARGE ? p.Rc : p.Cc;
    const int s0 = R_LARGE ? c0 : r0, l0 = R_LARGE ? r0 : c0;
    const unsigned row_bytes = (unsigned)(p.Wl * Cl) * 2u;
    const __amdgpu_buffer_rsrc_t rl = buf_rsrc(lp, (size_t)p.Nb * p.Hl * p.Wl * Cl * sizeof(T));

    // per-thread constants of the loads and of their places in LDS
    unsigned s_off[2], l_off[G::VL];
    int s_lds[2], l_lds[G::VL];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int v = tid + i * 256, prow = v >> 3, cv = v & 7;
        s_off[i] = (unsigned)((prow * Cs + s0 + cv * 8) * 2);
        s_lds[i] = pm_off<T>(prow, cv, 128);
    }
    unsigned pr0_mask = 0;   // bit j: vector j lies in patch row 0 (large-grid row 2 iy0 - 1)
#pragma unroll
    for (int j = 0; j < G::VL; ++j) {
        const int v = tid + j * 256;
        const int pr = v >> (LW + 4), x = (v >> 3) & (G::WL - 1), cv = v & 7, s = (x >> 1) + 1;
        const bool in = G::NVL % 256 == 0 || v < G::NVL;
        if (pr == 0) pr0_mask |= 1u << j;
        // offset from the start of large-grid row 2 iy0 (patch row 1); patch row 0 is addressed from one row earlier
        l_off[j] = in ? (unsigned)(((max(pr, 1) - 1) * p.Wl + x) * Cl + l0 + cv * 8) * 2u : BUF_OOB;
        l_lds[j] = G::SBYTES + (pr * 2 + (x & 1)) * G::PLANE + s * 128 + ((((cv >> 2) ^ ((s >> 1) & 1)) << 2) | (cv & 3)) * 16;
    }
    // two chunks in flight in registers (a chunk is 48 KB per block: with one, the loads of chunk c + 1 had the ~1 us of chunk c's
    // MFMAs to arrive and the loop waited on them: 3.3 us per chunk at 512 images)
    uint4 vs[2][2], vl[2][G::VL];
    auto load = [&](int chunk, uint4 (&ws)[2], uint4 (&wl)[G::VL]) __attribute__((always_inline)) {
        const int pix0 = chunk << 6;
        const __amdgpu_buffer_rsrc_t rs = buf_rsrc((const char*)sp + (size_t)pix0 * Cs * sizeof(T), (size_t)(p.K - pix0) * Cs * sizeof(T));
#pragma unroll
        for (int i = 0; i < 2; ++i) ws[i] = buf_ld16(rs, s_off[i], 0u);
        const int n = pix0 >> p.hw_shift, iy0 = (pix0 & ((1 << p.hw_shift) - 1)) >> LW;
        const unsigned base1 = (unsigned)((n * p.Hl + 2 * iy0) * p.Wl * Cl) * 2u;       // row 2 iy0
        const unsigned base0 = iy0 ? base1 - row_bytes : BUF_OOB;                       // row 2 iy0 - 1: above the image for iy0 = 0
#pragma unroll
        for (int j = 0; j < G::VL; ++j) wl[j] = buf_ld16(rl, buf_off_add(l_off[j], ((pr0_mask >> j) & 1u) ? base0 : base1), 0u);
    };

    // fragment addresses (the same for every chunk): lane (r, h) of k-step kk needs small-grid pixels k = 16 kk + 8 h + q + {0, 4},
    // q = (lane & 15) >> 2; pixel k = (ry, ix) meets patch row 2 ry + ky, plane (kx != 1), slot ix + (kx != 0)
    const int q = (lane & 15) >> 2;
    const int lane_part = 32 * ((lane >> 4) & 1) + 8 * (lane & 3);
    const int lt = R_LARGE ? wm : wn;       // 32-channel half of the large-grid operand's tile this wave works on
    int fbase[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        const int k = kk * 16 + 8 * (lane >> 5) + q, ry = k >> LW, ix = k & (G::WS - 1);
        fbase[kk] = G::SBYTES + (4 * ry) * G::PLANE + ix * 128 + lane_part;
    }
    const int swz0 = ((lt ^ ((q >> 1) & 1)) * 64), swz1 = ((lt ^ (((q + 1) >> 1) & 1)) * 64);   // slot ix (ix = q mod 4) / slot ix + 1

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    load(c_begin, vs[0], vl[0]);
    if (c_begin + 1 < c_end) load(c_begin + 1, vs[1], vl[1]);
    for (int c = c_begin; c < c_end; c += 2) {
#pragma unroll
        for (int d = 0; d < 2; ++d) {
            const int cc = c + d;
            if (cc < c_end) {   // block-uniform
                // (two stages: a wave writes stage d again only after the barrier of the iteration in between)
                char* stage = lds9 + d * G::STAGE;
#pragma unroll
                for (int i = 0; i < 2; ++i) *(uint4*)(stage + s_lds[i]) = vs[d][i];
#pragma unroll
                for (int j = 0; j < G::VL; ++j)
                    if (G::NVL % 256 == 0 || tid + j * 256 < G::NVL) *(uint4*)(stage + l_lds[j]) = vl[d][j];
                __syncthreads();
                if (cc + 2 < c_end) load(cc + 2, vs[d], vl[d]);
                // (requesting the ten fragments of k-step kk + 1 before the nine MFMAs of k-step kk, behind scheduling barriers, changed
                // nothing: the loop runs at the rate the operands arrive; what is left of a launch is the atomic epilogue)
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    Frag<T> fs;
                    frag_pm(fs, stage, 128, kk * 16, (R_LARGE ? wn : wm) * 32, lane);
#pragma unroll
                    for (int t = 0; t < 9; ++t) {
                        const int ky = t / 3, kx = t % 3;
                        const char* a = stage + fbase[kk] + (2 * ky + (kx != 1 ? 1 : 0)) * G::PLANE + (kx != 0 ? 128 : 0) + (kx != 0 ? swz1 : swz0);
                        union { s16x4 s[2]; bf16x8 b; } u;
                        u.s[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a));
                        u.s[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a + 512));
                        Frag<T> fl;
                        fl.v = u.b;
                        if (R_LARGE) mma32(acc[t], fl, fs); else mma32(acc[t], fs, fl);
                    }
                }
            }
        }
    }

#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = r0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), col = c0 + wn * 32 + (lane & 31);
            float* dst = p.dw + ((size_t)row * 9 + t) * p.Cst + col;
            if (p.splitk > 1) atomicAdd(dst, acc[t][r]); else if (p.overwrite) *dst = acc[t][r]; else *dst += acc[t][r];
        }
}

// blocks of a wgrad9 launch: ~target over the tiles, every block at least `minch` chunks
inline int w9_splitk(int tiles, int nchunk, int target, int minch) {
    return std::max(1, std::min(std::max(1, (target + tiles / 2) / tiles), std::max(1, nchunk / minch)));
}

thread_local int g_pm_adam_applied = 0;   // did the last weight-gradient launch of this thread take the optimiser step in its epilogue?

// wgrad9_kernel where it applies and pays, else *taken = 0 and nothing is launched.  `old_sk`: igemm_pm's split of this reduction;
// `caller_splitk`: the splitk argument of the call (0 / -1: the library chooses).
// FAVAE_WG9: 0 never, 1 (default) launches over at least FAVAE_WG9_MINNB (128) images -- at batch 64 the operands stay cached between
// the taps and nine times the atomic volume costs more than the re-reads (measured per layer, tools/bench_gemm.py) --, 2 wherever the
// shape allows.
int launch_w9(PmParams p, int old_sk, int caller_splitk, int* taken, hipStream_t st) {
    *taken = 0;
    static const int mode = getenv("FAVAE_WG9") ? atoi(getenv("FAVAE_WG9")) : 1;
    // ~96 blocks: a block holds 101 KB of LDS and 4 x 340 registers per lane, so a CU that runs one has little room for the blocks of
    // the main chain's kernels; with one on every CU the BatchNorm passes of the grouped step ran 4-7x slower than alone.  Same-box sweep
    // at 8 groups: 256 blocks 3.045 ms per step, 192 2.978, 128 2.896, 96 2.870, 64 3.012, 32 3.445 (the weight gradients become the chain).
    static const int target = getenv("FAVAE_WG9_TARGET") ? atoi(getenv("FAVAE_WG9_TARGET")) : 96;
    static const int minch = getenv("FAVAE_WG9_MINCH") ? atoi(getenv("FAVAE_WG9_MINCH")) : 8;
    static const int minnb = getenv("FAVAE_WG9_MINNB") ? atoi(getenv("FAVAE_WG9_MINNB")) : 128;
    const bool shape_ok = p.ks == 3 && p.stride == 2 && p.pad == 1 && (p.r_large != 0) != (p.c_large != 0) && p.Rc % 64 == 0 && p.Cc % 64 == 0 &&
                          p.Rst == p.Rc && p.Cst == p.Cc && p.Hs == p.Ws && (p.Ws == 8 || p.Ws == 16 || p.Ws == 32) && p.Hl == 2 * p.Hs &&
                          p.Wl == 2 * p.Ws && (size_t)p.Nb * p.Hl * p.Wl * (size_t)std::max(p.Rc, p.Cc) * 2 < (size_t)0x7FFFFFF0;
    const bool forced = g_force_pm_tile == 9;   // favae_debug_force_tile: wherever the shape allows
    if (!((mode || forced) && (g_force_pm_tile < 0 || forced) && shape_ok && old_sk > 1 && (caller_splitk == 0 || caller_splitk == -1) &&
          (mode == 2 || forced || p.Nb >= minnb)))
        return FAVAE_OK;
    const int tiles = (p.Rc / 64) * (p.Cc / 64), nchunk = p.K / 64;
    p.ntn = p.Cc / 64;
    p.splitk = w9_splitk(tiles, nchunk, target, minch);
    const unsigned total = (unsigned)tiles * (unsigned)p.splitk;
    if (p.rider.partial && !p.rider.out0 && (unsigned)cdiv(p.rider.C, 8) > total) {   // too few blocks to carry it: a launch of its own
        const int rc = favae_bn_dbias_finish(p.rider.partial, p.rider.nslab, p.rider.C, p.rider.out, st);
        if (rc != FAVAE_OK) return rc;
        p.rider.partial = nullptr;
    }
    p.xcd_units = (tiles > 1 && total % 8 == 0 && total >= 64) ? (int)(total / 8) : 0;
    p.hw_shift = __builtin_ctz((unsigned)(p.Hs * p.Ws));
    p.ws_shift = __builtin_ctz((unsigned)p.Ws);
    p.ad_p = nullptr;
    g_pm_adam_applied = 0;
    static bool attr_set[64] = {false};
    int dev = 0;
    FAVAE_HIP(hipGetDevice(&dev));
    FAVAE_CHECK_ARG(dev >= 0 && dev < 64, "favae_conv_wgrad: device index %d", dev);
    if (!attr_set[dev]) {
        FAVAE_HIP(hipFuncSetAttribute((const void*)wgrad9_kernel<5, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * W9<5>::STAGE));
        FAVAE_HIP(hipFuncSetAttribute((const void*)wgrad9_kernel<5, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * W9<5>::STAGE));
        FAVAE_HIP(hipFuncSetAttribute((const void*)wgrad9_kernel<4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * W9<4>::STAGE));
        FAVAE_HIP(hipFuncSetAttribute((const void*)wgrad9_kernel<4, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * W9<4>::STAGE));
        FAVAE_HIP(hipFuncSetAttribute((const void*)wgrad9_kernel<3, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * W9<3>::STAGE));
        FAVAE_HIP(hipFuncSetAttribute((const void*)wgrad9_kernel<3, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * W9<3>::STAGE));
        attr_set[dev] = true;
    }
    ProfScope prof(5, 2.0 * p.Rc * (double)p.Cc * p.K * 9, st);
    const dim3 grid(tiles, p.splitk);
#define W9_LAUNCH(LW_)                                                                                                 \
    do {                                                                                                               \
        if (p.r_large) hipLaunchKernelGGL((wgrad9_kernel<LW_, true>), grid, dim3(256), 2 * W9<LW_>::STAGE, st, p);     \
        else hipLaunchKernelGGL((wgrad9_kernel<LW_, false>), grid, dim3(256), 2 * W9<LW_>::STAGE, st, p);              \
    } while (0)
    if (p.Ws == 32) W9_LAUNCH(5); else if (p.Ws == 16) W9_LAUNCH(4); else W9_LAUNCH(3);
#undef W9_LAUNCH
    FAVAE_LAUNCH_CHECK();
    *taken = 1;
    return FAVAE_OK;
}

template <class T, bool TR, bool SLOW, int BM, int BN, int WM, int WN, int KG = 1, bool FI = false>
int launch_kc(KcParams p, int classes, hipStream_t st) {
    dim3 grid(cdiv(p.M, BM) * p.ntn, p.splitk, classes);
    // FAVAE_KC_XCD: 0 launch order = tile order, 1 XCD-contiguous tiles wherever the grid allows, 2 (default) only where
    // activation rows are what the tiles share (few n-tiles); with many n-tiles launch order already gives every XCD its own
    // slice of the weights (1 is slower there: enc.4 forward 15.1 -> 16.3 us).  Measured alone (tools/bench_gemm.py): dec.3 input
    // gradient 11.6 -> 9.7 us, the other layers within noise; no measurable change of the step (these launches are not bound by L2 misses).
    static const int xcd_mode = getenv("FAVAE_KC_XCD") ? atoi(getenv("FAVAE_KC_XCD")) : 2;
    p.xcd_chunk = (xcd_mode && grid.x % 8 == 0 && grid.x >= 16 && (xcd_mode == 1 || p.ntn <= 4)) ? (int)(grid.x / 8) : 0;
    const int cls = SLOW ? (TR ? 9 : 8) : (TR ? 4 : 0) + (BM == 64 ? 2 : 3);
    // algorithmic work, dense-tap convention: gather form ks*ks taps per output pixel, stride-2 scatter form 9/4
    const double taps = (TR && p.stride == 2) ? 2.25 : (double)(p.ks * p.ks);
    ProfScope prof(cls, 2.0 * p.Nb * p.Ho * p.Wo * (double)p.Cout * taps * p.Cin, st);
    hipLaunchKernelGGL((igemm_kc<T, BM, BN, WM, WN, TR, SLOW, KG, FI>), grid, dim3(256 * KG), 0, st, p);
    FAVAE_LAUNCH_CHECK();
    return FAVAE_OK;
}

// split of the reduction where the tile grid alone cannot fill the chip (dispatch_kc and favae_conv_igemm_auto_splitk)
inline int kc_auto_splitk(int blocks, int nchunk, int target_blocks) {
    int sk = 1;
    if (2 * blocks < target_blocks) sk = std::min(std::max(1, nchunk / 4), (target_blocks + blocks / 2) / blocks);
    return std::max(1, std::min(sk, nchunk));
}

template <class T, bool TR>
int dispatch_kc(KcParams p, int classes, int target_blocks, int* fused, hipStream_t st) {
    const bool slow = (p.Cin % BK) != 0;
    p.carry = favae_take_flag_carry();
    // Batch-64 layers are 2-5 GFLOP each: a 64x64 tile grid of 256-1024 blocks beat 128-wide tiles with deeper split-K
    // on every layer of the step (sweep on MI355X), because the fp32 atomic traffic of split-K grows with the tile
    // area while these grids already cover the 256 CUs.  128x32 serves the few-output-channel gather layers.
    bool narrow_n = !slow && p.Cout <= 32 && !TR;
    if (!slow && !TR && g_force_kc_tile >= 0 && !p.in_ab && !p.ep_ab) narrow_n = g_force_kc_tile == 1;
    const int BM = narrow_n ? 128 : 64, BN = narrow_n ? 32 : 64;
    p.ntn = cdiv(p.Cout, BN);
    // split the reduction when the tile grid alone cannot fill 256 CUs
    const int ntap_max = p.ks * p.ks;
    const int nchunk = slow ? cdiv(ntap_max * p.Cin, BK) : ntap_max * (p.Cin / BK);
    const int blocks = cdiv(p.M, BM) * p.ntn * classes;
    const bool prezeroed = p.splitk < 0;   // caller guarantees out == 0 (one pooled memset per step, vae_model.hip)
    int sk = 1;
    if (p.splitk > 0) sk = p.splitk;
    else sk = kc_auto_splitk(blocks, nchunk, target_blocks);
    p.splitk = std::max(1, std::min(sk, nchunk));
    if (p.out_bf16) p.splitk = 1;   // a bf16 result cannot be accumulated across blocks
    // the per-channel epilogue sums exist in the staged 64x64 epilogue of unsplit reductions only
    if (!(p.splitk == 1 && !narrow_n && (p.Cout & 3) == 0)) p.red_bins = nullptr;
    if (fused) *fused = p.red_bins != nullptr;
    if (p.splitk > 1 && !prezeroed)
        FAVAE_HIP(hipMemsetAsync(p.out, 0, (size_t)p.Nb * p.Ho * p.Wo * p.Cout * sizeof(float), st));
    if (slow) return launch_kc<T, TR, true, 64, 64, 2, 2>(p, classes, st);
    if constexpr (!TR) {
        if (narrow_n) return launch_kc<T, TR, false, 128, 32, 4, 1>(p, classes, st);
    }
    // two reduction groups per tile where the grid leaves the CUs with one or two 4-wave blocks and a block's share of
    // the reduction is long enough to be worth halving (FAVAE_KG: 1 never, 2 always)
    static const int kg_env = getenv("FAVAE_KG") ? atoi(getenv("FAVAE_KG")) : 0;
    const int per_block = cdiv(nchunk, p.splitk);
    // Measured on MI355X (tools/bench_gemm.py): it pays only where the grid is one block per CU and the reduction long
    // (enc.2 forward, dec.2 input gradient: 256 tiles x 18 chunks, 14.1 -> 11.6 us); with two blocks per CU already
    // resident the second group only adds its hand-over (12.1 -> 14.1 us).  With every piece of the chunk loop removed
    // these launches still take ~9 us (launch, first loads, epilogue, drain): the loop itself is 3-5 us of a 12-17 us launch.
    const bool two = kg_env == 2 || (kg_env == 0 && blocks * p.splitk <= (TR ? 512 : 256) && per_block >= 12);
    if (p.in_ab) {   // fused input (checked by the caller: whole-chunk channel count <= 256, 64x64 tile)
        if (two) return launch_kc<T, TR, false, 64, 64, 2, 2, 2, true>(p, classes, st);
        return launch_kc<T, TR, false, 64, 64, 2, 2, 1, true>(p, classes, st);
    }
    if (two) return launch_kc<T, TR, false, 64, 64, 2, 2, 2>(p, classes, st);
    return launch_kc<T, TR, false, 64, 64, 2, 2>(p, classes, st);
}

// tile (0: 32x128, 1: 128x32, 2: 64x64) and automatic split of the weight-gradient reduction
inline int pm_tile_of(int Rc, int Cc) { return Rc <= 32 ? 0 : Cc <= 32 ? 1 : 2; }
int pm_auto_splitk(int Rc, int Cc, long long K, int ks, int tile) {
    const int BM = tile == 0 ? 32 : tile == 1 ? 128 : 64, BN = tile == 0 ? 128 : tile == 1 ? 32 : 64;
    const int nchunk = (int)((K + BK - 1) / BK);
    const int blocks = cdiv(Rc, BM) * cdiv(Cc, BN) * ks * ks;
    // ~256 blocks: the weight gradients run on the side stream beside the main chain; a grid that fills the chip several
    // times over delays the main-stream kernels more than it shortens itself (tools/prof_step.py sweep)
    static const int pm_target = getenv("FAVAE_PM_TARGET") ? atoi(getenv("FAVAE_PM_TARGET")) : 256;
    return std::max(1, std::min(std::min(std::max(1, nchunk / 8), std::max(1, (pm_target + blocks / 2) / blocks)), nchunk));
}

template <class T>
int dispatch_pm(PmParams p, hipStream_t st) {
    p.carry = favae_take_flag_carry();
    p.rider = favae_take_slab_rider();
    FAVAE_CHECK_ARG(!(p.rider.partial && (p.rider.nslab < 1 || p.rider.nslab > 65536 || p.rider.C < 1 || p.rider.C > 2048)),
                    "weight gradient: slab rider of %d slabs x %d channels", p.rider.nslab, p.rider.C);
    int tile = pm_tile_of(p.Rc, p.Cc);   // 32x128, 128x32, 64x64
    if (g_force_pm_tile >= 0 && g_force_pm_tile <= 2) tile = g_force_pm_tile;
    const int BM = tile == 0 ? 32 : tile == 1 ? 128 : 64, BN = tile == 0 ? 128 : tile == 1 ? 32 : 64;
    p.ntn = cdiv(p.Cc, BN);
    const int nchunk = cdiv(p.K, BK);
    const int blocks = cdiv(p.Rc, BM) * p.ntn * p.ks * p.ks;
    const int orig_splitk = p.splitk;
    p.overwrite = p.splitk < 0;
    int sk = p.splitk > 0 ? p.splitk : p.splitk < -1 ? -p.splitk   // < -1: that many slices of a gradient the caller zeroed
                                     : pm_auto_splitk(p.Rc, p.Cc, p.K, p.ks, tile);
    if constexpr (sizeof(T) == 2) {   // all nine taps from one pass over the operands, where that pays
        int taken = 0;
        const int rc = launch_w9(p, std::max(1, std::min(sk, nchunk)), orig_splitk, &taken, st);
        if (rc != FAVAE_OK || taken) return rc;
    }
    p.splitk = std::max(1, std::min(sk, nchunk));
    dim3 grid(cdiv(p.Rc, BM) * p.ntn, p.splitk, p.ks * p.ks);
    if (p.rider.partial && !p.rider.out0 && (long long)cdiv(p.rider.C, 8) > (long long)grid.x * grid.y * grid.z) {   // too few blocks to carry it: a launch of its own
        const int rc = favae_bn_dbias_finish(p.rider.partial, p.rider.nslab, p.rider.C, p.rider.out, st);
        if (rc != FAVAE_OK) return rc;
        p.rider.partial = nullptr;
    }
    {
        static const bool xcd_off = getenv("FAVAE_PM_XCD") && getenv("FAVAE_PM_XCD")[0] == '0';
        const unsigned total = grid.x * grid.y * grid.z;
        p.xcd_units = (!xcd_off && grid.z > 1 && total % 8 == 0 && total >= 64) ? (int)(total / 8) : 0;
    }
    ProfScope prof(10 + tile, 2.0 * p.Rc * (double)p.Cc * p.K * p.ks * p.ks, st);
    constexpr int VE = 16 / (int)sizeof(T);
    const size_t big = std::max((size_t)p.Nb * p.Hl * p.Wl, (size_t)p.K) * (size_t)std::max(p.Rc, p.Cc) * sizeof(T);
    static const bool lean_off = getenv("FAVAE_PM_LEAN") && getenv("FAVAE_PM_LEAN")[0] == '0';
    const bool lean = !lean_off && p.Rc % VE == 0 && p.Cc % VE == 0 && big < (size_t)0x7FFFFFF0;
    const int hw = p.Hs * p.Ws;
    p.hw_shift = p.ws_shift = -1;
    if ((hw & (hw - 1)) == 0 && (p.Ws & (p.Ws - 1)) == 0) { p.hw_shift = __builtin_ctz((unsigned)hw); p.ws_shift = __builtin_ctz((unsigned)p.Ws); }
#define PM_LAUNCH(BM_, BN_, WM_, WN_)                                                                        \
    do {                                                                                                     \
        if (lean) hipLaunchKernelGGL((igemm_pm<T, BM_, BN_, WM_, WN_, true>), grid, dim3(256), 0, st, p);    \
        else hipLaunchKernelGGL((igemm_pm<T, BM_, BN_, WM_, WN_, false>), grid, dim3(256), 0, st, p);        \
    } while (0)
    // the optimiser step in the epilogue: unsplit 64x64 launches over whole tiles, 16-byte aligned transposed rows
    g_pm_adam_applied = 0;
    if (p.ad_p && !(tile == 2 && p.splitk == 1 && lean && p.Rst % 64 == 0 && p.Cst % 64 == 0 && p.Rst == p.Rc && p.Cst == p.Cc)) p.ad_p = nullptr;
    if (p.ad_p) {
        g_pm_adam_applied = 1;
        hipLaunchKernelGGL((igemm_pm<T, 64, 64, 2, 2, true, true>), grid, dim3(256), 0, st, p);
    }
    else if (p.c_ab) {
        if constexpr (sizeof(T) == 2) {
            FAVAE_CHECK_ARG(tile == 0 && lean && !p.r_large && !p.c_large && p.K % BK == 0 && (p.c_ab_group_pix == 0 || p.c_ab_group_pix % BK == 0),
                            "weight gradient with the BatchNorm of its column operand at the load: 32-row tile, lean loads, K and the group size multiples of 64");
            hipLaunchKernelGGL((igemm_pm<T, 32, 128, 1, 4, true, false, true>), grid, dim3(256), 0, st, p);
        } else {
            FAVAE_CHECK_ARG(false, "weight gradient with the BatchNorm of its column operand at the load: bf16 only");
        }
    }
    else if (tile == 0) PM_LAUNCH(32, 128, 1, 4);
    else if (tile == 1) PM_LAUNCH(128, 32, 4, 1);
    else PM_LAUNCH(64, 64, 2, 2);
#undef PM_LAUNCH
    FAVAE_LAUNCH_CHECK();
    return FAVAE_OK;
}

}  // namespace

#define RC_(x) do { int rc_ = (x); if (rc_ != FAVAE_OK) return rc_; } while (0)
namespace {
struct RedArgs { double* bins; int nbin, mode; const void* act; const void* raw; const float *mean, *invstd; int* fused; int raw_bf16; int group_images = 0; const float* ab = nullptr; };
struct Fin { const float *gamma, *beta; float *mean, *invstd, *ab, *run_mean, *run_var; float eps, mom; };

// a ticket per launch that finalises BatchNorm statistics: slots of a zero-initialised device array handed out in rotation
// (a launch leaves its slot at zero); launches that overlap in time must not share a slot, 256 in flight is far beyond a step
__device__ unsigned g_fin_tickets[256];
int fin_ticket(unsigned** out) {
    static unsigned* per_device[64] = {nullptr};
    static std::atomic<unsigned> next{0};
    int dev = 0;
    FAVAE_HIP(hipGetDevice(&dev));
    FAVAE_CHECK_ARG(dev >= 0 && dev < 64, "favae_conv_igemm: device index %d", dev);
    if (!per_device[dev]) FAVAE_HIP(hipGetSymbolAddress((void**)&per_device[dev], HIP_SYMBOL(g_fin_tickets)));
    *out = per_device[dev] + (next.fetch_add(1u) & 255u);
    return FAVAE_OK;
}

int conv_igemm_impl(const void* in, const void* w, const float* bias, float* out, int Nb, int Hi, int Wi, int Cin, int Ho,
                    int Wo, int Cout, int ks, int stride, int pad, int scatter, int splitk, int dtype, const RedArgs& red,
                    void* stream, const float* in_ab = nullptr, const Fin* fin = nullptr, const float* ep_ab = nullptr) {
    FAVAE_CHECK_ARG(in && w && out, "favae_conv_igemm: null pointer");
    FAVAE_CHECK_ARG(Nb > 0 && Hi > 0 && Wi > 0 && Cin > 0 && Ho > 0 && Wo > 0 && Cout > 0, "favae_conv_igemm: bad shape");
    FAVAE_CHECK_ARG((ks == 1 && pad == 0) || (ks == 3 && pad == 1), "favae_conv_igemm: ks/pad must be 1/0 or 3/1");
    FAVAE_CHECK_ARG(stride == 1 || stride == 2, "favae_conv_igemm: stride must be 1 or 2");
    const int out_bf16 = (dtype & FAVAE_OUT_BF16) != 0;
    dtype &= 0xF;
    FAVAE_CHECK_ARG(dtype == FAVAE_F32 || dtype == FAVAE_BF16, "favae_conv_igemm: dtype");
    FAVAE_CHECK_ARG(!(out_bf16 && dtype != FAVAE_BF16), "favae_conv_igemm: FAVAE_OUT_BF16 needs FAVAE_BF16");
    FAVAE_CHECK_ARG(!(stride == 2 && ks != 3), "favae_conv_igemm: stride 2 needs ks 3");
    if (scatter) FAVAE_CHECK_ARG(Ho == Hi * stride && Wo == Wi * stride, "favae_conv_igemm: scatter form needs Ho = s*Hi");
    else FAVAE_CHECK_ARG(Ho == (Hi + 2 * pad - ks) / stride + 1 && Wo == (Wi + 2 * pad - ks) / stride + 1,
                         "favae_conv_igemm: gather form output size mismatch");
    FAVAE_CHECK_ARG((size_t)Nb * Ho * Wo < (1u << 31) && (size_t)Nb * Hi * Wi < (1u << 31), "favae_conv_igemm: too many pixels");
    const int esz = dtype == FAVAE_F32 ? 4 : 2;
    if (Cin % BK == 0) FAVAE_CHECK_ARG(((uintptr_t)in % 16) == 0 && ((uintptr_t)w % 16) == 0, "favae_conv_igemm: 16-byte alignment");
    (void)esz;
    hipStream_t st = (hipStream_t)stream;
    if (red.fused) *red.fused = 0;
    // one thread per output pixel pays off once the pixel count fills the chip several times over (the decoder side)
    if (!in_ab && !ep_ab && g_force_kc_tile < 0 && (long long)Nb * Ho * Wo >= 131072 && thin_in_applicable(Cin, Cout, ks, stride, scatter)) {
        ProfScope prof(14, 2.0 * Nb * Ho * Wo * (double)Cout * 9 * Cin, (hipStream_t)stream);
        ThinRed tr{nullptr, 0, nullptr, nullptr, nullptr, nullptr, 0, 0};
        // (bf16 output / bf16 raw with the sums: grouped-batch mode only, where that form is tested)
        if (red.bins && red.mode == 2 && ((!out_bf16 && !red.raw_bf16) || red.group_images > 0)) {   // the statistics form (mode 1) is not needed on this path
            tr = ThinRed{red.bins, red.nbin, red.act, red.raw, red.mean, red.invstd, red.group_images * Ho * Wo, red.raw_bf16};
            if (red.fused) *red.fused = 1;
        }
        return thin_in_launch(in, w, bias, out, Nb, Hi, Wi, Cin, Ho, Wo, Cout, stride, scatter, dtype, out_bf16, tr, (hipStream_t)stream);
    }
    // final_layer.0's shape (transposed 64 -> 64 convolution over 32-pixel rows, bf16) at 128 images or more (grouped-batch steps, eval
    // passes): the kernel with resident tap matrices.  Measured on MI355X alone: 123 us against 183 us at 512 images (fp32 out; with
    // the BatchNorm sums 141 against 230), 23 against 29 us at 64 images, where most of it is the start-up (81 KB of weights per
    // block) and the training step does not change (0.852 ms either way), so the batch-64 step keeps the scatter form of igemm_kc.
    // In the step at 8 groups: 3.19 -> 3.16 ms; reconstruction pass 42.5 -> 41.4 ms per 32 768 images.
    // (FAVAE_CT64=0 or a forced tile: the scatter form everywhere; FAVAE_CT64=1: this kernel at any batch)
    {
        static const int ct64_mode = getenv("FAVAE_CT64") ? atoi(getenv("FAVAE_CT64")) : -1;
        if (ct64_mode != 0 && (Nb >= 128 || ct64_mode == 1) && g_force_kc_tile < 0 && dtype == FAVAE_BF16 && scatter && stride == 2 && ks == 3 && Cin == 64 && Cout == 64 && Wi == 32 &&
            Hi % 2 == 0 && !in_ab && !fin && splitk <= 1 && !(red.bins && red.mode != 1) && ((uintptr_t)out % 16) == 0 &&
            (!ep_ab || ((uintptr_t)ep_ab % 4) == 0)) {
            if (red.bins && red.fused) *red.fused = 1;
            return convt64_fwd_launch(in, w, bias, out, Nb, Hi, out_bf16, ep_ab, red.bins, red.nbin, red.group_images, st);
        }
    }
    KcParams p;
    p.in = in; p.w = w; p.bias = bias; p.out = out;
    p.Nb = Nb; p.Hi = Hi; p.Wi = Wi; p.Cin = Cin; p.Ho = Ho; p.Wo = Wo; p.Cout = Cout;
    p.ks = ks; p.stride = stride; p.pad = pad; p.ldw = ks * ks * Cin; p.splitk = splitk; p.out_bf16 = out_bf16;
    p.red_bins = red.bins; p.red_nbin = red.nbin; p.red_mode = red.mode; p.red_act = red.act; p.red_raw = red.raw;
    p.red_mean = red.mean; p.red_invstd = red.invstd; p.red_raw_bf16 = red.raw_bf16; p.red_ab = red.ab;
    p.ep_ab = ep_ab;
    if (ep_ab) FAVAE_CHECK_ARG(splitk == 1 && Cout % 4 == 0 && Cout > 32 && ((uintptr_t)ep_ab % 16) == 0,
                               "favae_conv_igemm_affine: needs splitk = 1 and Cout > 32, a multiple of 4");
    p.in_ab = in_ab;
    if (in_ab) {
        FAVAE_CHECK_ARG(Cin % BK == 0 && Cin <= 256 && Cout > 32, "favae_conv_igemm_bn_in: Cin must be 64, 128, 192 or 256 and Cout > 32");
        FAVAE_CHECK_ARG(((uintptr_t)in % 16) == 0 && ((uintptr_t)in_ab % 16) == 0, "favae_conv_igemm_bn_in: 16-byte alignment");
    }
    p.fin_gamma = p.fin_beta = nullptr; p.fin_mean = p.fin_invstd = p.fin_ab = p.fin_run_mean = p.fin_run_var = nullptr;
    p.fin_R = 1.0; p.fin_eps = p.fin_mom = 0.f; p.fin_ticket = nullptr;
    if (fin && red.bins && red.mode == 1) {
        p.fin_gamma = fin->gamma; p.fin_beta = fin->beta; p.fin_mean = fin->mean; p.fin_invstd = fin->invstd; p.fin_ab = fin->ab;
        p.fin_run_mean = fin->run_mean; p.fin_run_var = fin->run_var; p.fin_R = (double)Nb * Ho * Wo; p.fin_eps = fin->eps; p.fin_mom = fin->mom;
        RC_(fin_ticket(&p.fin_ticket));
    }
    int classes = 1;
    if (scatter && stride == 2) { classes = 4; p.HoC = Ho / 2; p.WoC = Wo / 2; }
    else { p.HoC = Ho; p.WoC = Wo; }
    p.M = Nb * p.HoC * p.WoC;
    p.hw_shift = p.woc_shift = -1;
    {
        const int hw = p.HoC * p.WoC;
        if ((hw & (hw - 1)) == 0 && (p.WoC & (p.WoC - 1)) == 0) { p.hw_shift = __builtin_ctz((unsigned)hw); p.woc_shift = __builtin_ctz((unsigned)p.WoC); }
    }
    p.ntn = 1;
    p.red_group_rows = red.group_images > 0 ? red.group_images * p.HoC * p.WoC : 0;
    if (p.red_group_rows) FAVAE_CHECK_ARG(p.red_group_rows % 64 == 0 && !fin, "favae_conv_igemm_red_groups: a group must be a multiple of 64 rows");
    // scatter form: four parity classes of 1 / 2 / 2 / 4 taps.  Up to 512 tiles in all the reduction stays whole (fp32 atomics of a
    // split reduction cost more than they bring: decoder.1 forward 23.5 -> 16.7 us unsplit with two reduction groups per tile,
    // and the BatchNorm sums ride in the epilogue again); below that it is split 4-way
    const int target = 512;
    if (dtype == FAVAE_F32)
        return scatter ? dispatch_kc<float, true>(p, classes, target, red.fused, st) : dispatch_kc<float, false>(p, classes, target, red.fused, st);
    return scatter ? dispatch_kc<bf16_t, true>(p, classes, target, red.fused, st) : dispatch_kc<bf16_t, false>(p, classes, target, red.fused, st);
}
}  // namespace

extern "C" int favae_conv_igemm(const void* in, const void* w, const float* bias, float* out, int Nb, int Hi, int Wi,
                                int Cin, int Ho, int Wo, int Cout, int ks, int stride, int pad, int scatter, int splitk,
                                int dtype, void* stream) {
    const RedArgs none{nullptr, 0, 0, nullptr, nullptr, nullptr, nullptr, nullptr, 0};
    return conv_igemm_impl(in, w, bias, out, Nb, Hi, Wi, Cin, Ho, Wo, Cout, ks, stride, pad, scatter, splitk, dtype, none, stream);
}

/* The number of reduction slices favae_conv_igemm picks for splitk <= 0 (1: every output element is stored once; > 1: accumulated with
 * atomics into an output that is zero).  0 while a developer override of the tile is active. */
extern "C" int favae_conv_igemm_auto_splitk(int Nb, int Ho, int Wo, int Cin, int Cout, int ks, int stride, int scatter) {
    if (Nb <= 0 || Ho <= 0 || Wo <= 0 || Cin <= 0 || Cout <= 0 || ks <= 0 || g_force_kc_tile >= 0) return 0;
    const bool slow = (Cin % BK) != 0, tr = scatter != 0;
    int classes = 1, HoC = Ho, WoC = Wo;
    if (scatter && stride == 2) { classes = 4; HoC = Ho / 2; WoC = Wo / 2; }
    const bool narrow_n = !slow && Cout <= 32 && !tr;
    const int BM = narrow_n ? 128 : 64, BN = narrow_n ? 32 : 64;
    const int nchunk = slow ? cdiv(ks * ks * Cin, BK) : ks * ks * (Cin / BK);
    const int blocks = cdiv(Nb * HoC * WoC, BM) * cdiv(Cout, BN) * classes;
    return kc_auto_splitk(blocks, nchunk, 512);
}

extern "C" int favae_conv_igemm_red(const void* in, const void* w, const float* bias, float* out, int Nb, int Hi, int Wi,
                                    int Cin, int Ho, int Wo, int Cout, int ks, int stride, int pad, int scatter, int splitk,
                                    int dtype, double* bins, int nbins, int mode, const void* act, const void* raw,
                                    const float* mean, const float* invstd, int* fused, void* stream) {
    FAVAE_CHECK_ARG(bins && fused && nbins >= 1 && nbins <= 256, "favae_conv_igemm_red: bins / fused / nbins");
    FAVAE_CHECK_ARG(mode == 1 || (mode == 2 && act && raw && mean && invstd), "favae_conv_igemm_red: mode 1, or mode 2 with act/raw/mean/invstd");
    const int raw_bf16 = (dtype & FAVAE_RAW_BF16) != 0;
    FAVAE_CHECK_ARG(!(raw_bf16 && (dtype & 0xF) != FAVAE_BF16), "favae_conv_igemm_red: FAVAE_RAW_BF16 needs FAVAE_BF16");
    const RedArgs red{bins, nbins, mode, act, raw, mean, invstd, fused, raw_bf16};
    return conv_igemm_impl(in, w, bias, out, Nb, Hi, Wi, Cin, Ho, Wo, Cout, ks, stride, pad, scatter, splitk, dtype & ~FAVAE_RAW_BF16, red, stream);
}

/* favae_conv_igemm_red for a batch made of BatchNorm groups of `group_images` images (grouped-batch mode): the sums of group g go
 * to bins + g * nbins * 2 * Cout, and mode 2 reads mean / invstd of group g at + g * Cout.  group_images * Ho * Wo (of a parity
 * class in the stride-2 scatter form) must be a multiple of 64. */
extern "C" int favae_conv_igemm_red_groups(const void* in, const void* w, const float* bias, float* out, int Nb, int Hi, int Wi,
                                           int Cin, int Ho, int Wo, int Cout, int ks, int stride, int pad, int scatter, int splitk,
                                           int dtype, double* bins, int nbins, int mode, const void* act, const void* raw,
                                           const float* mean, const float* invstd, const float* ab, int group_images, int* fused,
                                           void* stream) {
    FAVAE_CHECK_ARG(bins && fused && nbins >= 1 && nbins <= 256 && group_images >= 1, "favae_conv_igemm_red_groups: bins / fused / nbins / group_images");
    FAVAE_CHECK_ARG(mode == 1 || (mode == 2 && act && raw && mean && invstd), "favae_conv_igemm_red_groups: mode 1, or mode 2 with act/raw/mean/invstd");
    FAVAE_CHECK_ARG(!ab || ((uintptr_t)ab % 16 == 0 && Cout % 4 == 0), "favae_conv_igemm_red_groups: ab must be 16-byte aligned, Cout a multiple of 4");
    const int raw_bf16 = (dtype & FAVAE_RAW_BF16) != 0;
    FAVAE_CHECK_ARG(!(raw_bf16 && (dtype & 0xF) != FAVAE_BF16), "favae_conv_igemm_red_groups: FAVAE_RAW_BF16 needs FAVAE_BF16");
    RedArgs red{bins, nbins, mode, act, raw, mean, invstd, fused, raw_bf16};
    red.group_images = group_images;
    red.ab = mode == 2 ? ab : nullptr;
    return conv_igemm_impl(in, w, bias, out, Nb, Hi, Wi, Cin, Ho, Wo, Cout, ks, stride, pad, scatter, splitk, dtype & ~FAVAE_RAW_BF16, red, stream);
}

extern "C" int favae_conv_igemm_bn(const void* in, const float* in_ab, const void* w, const float* bias, float* out, int Nb, int Hi,
                                   int Wi, int Cin, int Ho, int Wo, int Cout, int ks, int stride, int pad, int scatter, int splitk,
                                   int dtype, double* bins, int nbins, const float* gamma, const float* beta, float eps,
                                   float momentum, float* mean, float* invstd, float* ab, float* run_mean, float* run_var, int* fused,
                                   void* stream) {
    FAVAE_CHECK_ARG(!(bins && !(fused && nbins >= 1 && nbins <= 256)), "favae_conv_igemm_bn: bins / fused / nbins");
    FAVAE_CHECK_ARG(!(ab && !(bins && gamma && beta && mean && invstd)), "favae_conv_igemm_bn: ab needs bins, gamma, beta, mean, invstd");
    const Fin fin{gamma, beta, mean, invstd, ab, run_mean, run_var, eps, momentum};
    const RedArgs red{bins, nbins, 1, nullptr, nullptr, nullptr, nullptr, bins ? fused : nullptr, 0};
    if (fused) *fused = 0;
    return conv_igemm_impl(in, w, bias, out, Nb, Hi, Wi, Cin, Ho, Wo, Cout, ks, stride, pad, scatter, splitk, dtype, red, stream, in_ab,
                           ab ? &fin : nullptr);
}

extern "C" int favae_conv_igemm_affine(const void* in, const void* w, const float* bias, void* out, int Nb, int Hi, int Wi, int Cin,
                                       int Ho, int Wo, int Cout, int ks, int stride, int pad, int scatter, int dtype,
                                       const float* ab, void* stream) {
    FAVAE_CHECK_ARG(ab, "favae_conv_igemm_affine: null ab");
    const RedArgs none{nullptr, 0, 0, nullptr, nullptr, nullptr, nullptr, nullptr, 0};
    return conv_igemm_impl(in, w, bias, (float*)out, Nb, Hi, Wi, Cin, Ho, Wo, Cout, ks, stride, pad, scatter, 1, dtype, none, stream, nullptr,
                           nullptr, ab);
}

extern "C" int favae_conv_wgrad(const void* rows, int Rc, int rows_large, const void* cols, int Cc, int cols_large,
                                float* dw, int Nb, int Hs, int Ws, int Hl, int Wl, int ks, int stride, int pad,
                                int splitk, int dtype, void* stream) {
    FAVAE_CHECK_ARG(rows && cols && dw, "favae_conv_wgrad: null pointer");
    FAVAE_CHECK_ARG(Nb > 0 && Hs > 0 && Ws > 0 && Rc > 0 && Cc > 0, "favae_conv_wgrad: bad shape");
    FAVAE_CHECK_ARG((ks == 1 && pad == 0) || (ks == 3 && pad == 1), "favae_conv_wgrad: ks/pad must be 1/0 or 3/1");
    FAVAE_CHECK_ARG(stride == 1 || stride == 2, "favae_conv_wgrad: stride");
    FAVAE_CHECK_ARG(dtype == FAVAE_F32 || dtype == FAVAE_BF16, "favae_conv_wgrad: dtype");
    FAVAE_CHECK_ARG(!(rows_large && cols_large), "favae_conv_wgrad: at most one gathered operand");
    FAVAE_CHECK_ARG(Hs == (Hl + 2 * pad - ks) / stride + 1 && Ws == (Wl + 2 * pad - ks) / stride + 1,
                    "favae_conv_wgrad: small/large size mismatch");
    FAVAE_CHECK_ARG((size_t)Nb * Hl * Wl < (1u << 31), "favae_conv_wgrad: too many pixels");
    FAVAE_CHECK_ARG(((uintptr_t)rows % 16) == 0 && ((uintptr_t)cols % 16) == 0, "favae_conv_wgrad: 16-byte alignment");
    hipStream_t st = (hipStream_t)stream;
    if (g_force_pm_tile < 0 && cols_large && Cc <= 4 && wgrad_nw_applicable(Rc, Cc, ks, stride, 1)) {
        ProfScope prof(15, 2.0 * Rc * (double)Cc * Nb * Hs * Ws * 9, st);   // wide = rows (small grid), narrow = cols
        return wgrad_nw_launch(rows, cols, dw, Nb, Hs, Ws, Rc, Hl, Wl, Cc, stride, 1, 9 * Cc, Cc, 1, dtype, st);
    }
    if (g_force_pm_tile < 0 && cols_large && Rc <= 4 && stride == 1 && wgrad_nw_applicable(Cc, Rc, ks, stride, 0)) {
        ProfScope prof(15, 2.0 * Rc * (double)Cc * Nb * Hs * Ws * 9, st);   // wide = cols (large grid), narrow = rows
        return wgrad_nw_launch(cols, rows, dw, Nb, Hl, Wl, Cc, Hs, Ws, Rc, stride, 0, 1, Cc, 9 * Cc, dtype, st);
    }
    PmParams p;
    p.rp = rows; p.cp = cols; p.dw = dw; p.Rc = Rc; p.Cc = Cc; p.r_large = rows_large; p.c_large = cols_large;
    p.Nb = Nb; p.Hs = Hs; p.Ws = Ws; p.Hl = Hl; p.Wl = Wl; p.ks = ks; p.stride = stride; p.pad = pad;
    p.splitk = splitk; p.ntn = 1; p.K = Nb * Hs * Ws; p.Rst = Rc; p.Cst = Cc;
    p.ad_p = nullptr;
    return dtype == FAVAE_F32 ? dispatch_pm<float>(p, st) : dispatch_pm<bf16_t>(p, st);
}

/* favae_conv_wgrad whose epilogue takes the Adam step of the layer's weights (torch.optim.Adam of cA/celeba_deepvanilla_vae.py:131
 * applied by optimizer.step() :250; same arithmetic as favae_adam_step) and writes their dtype copy wc [Rc][ks*ks][Cc] and transpose
 * wt [Cc][ks*ks][Rc] (either may be NULL), where the launch runs unsplit in 64x64 tiles (Rc, Cc multiples of 64): *applied = 1, dw is
 * not touched, p / m / v [Rc][ks*ks][Cc] are updated in place.  Otherwise *applied = 0 and the call is favae_conv_wgrad into dw. */
extern "C" int favae_conv_wgrad_adam(const void* rows, int Rc, int rows_large, const void* cols, int Cc, int cols_large, float* dw,
                                     float* p_, float* m_, float* v_, void* wc, void* wt, int Nb, int Hs, int Ws, int Hl, int Wl, int ks,
                                     int stride, int pad, int splitk, float lr, float beta1, float beta2, float eps, long long step,
                                     int dtype, int* applied, void* stream) {
    FAVAE_CHECK_ARG(rows && cols && dw && p_ && m_ && v_ && applied && step >= 1, "favae_conv_wgrad_adam: null pointer / step");
    FAVAE_CHECK_ARG(Nb > 0 && Hs > 0 && Ws > 0 && Rc > 0 && Cc > 0, "favae_conv_wgrad_adam: bad shape");
    FAVAE_CHECK_ARG((ks == 1 && pad == 0) || (ks == 3 && pad == 1), "favae_conv_wgrad_adam: ks/pad must be 1/0 or 3/1");
    FAVAE_CHECK_ARG(stride == 1 || stride == 2, "favae_conv_wgrad_adam: stride");
    FAVAE_CHECK_ARG(dtype == FAVAE_F32 || dtype == FAVAE_BF16, "favae_conv_wgrad_adam: dtype");
    FAVAE_CHECK_ARG(!(rows_large && cols_large), "favae_conv_wgrad_adam: at most one gathered operand");
    FAVAE_CHECK_ARG(Hs == (Hl + 2 * pad - ks) / stride + 1 && Ws == (Wl + 2 * pad - ks) / stride + 1, "favae_conv_wgrad_adam: small/large size mismatch");
    FAVAE_CHECK_ARG((size_t)Nb * Hl * Wl < (1u << 31), "favae_conv_wgrad_adam: too many pixels");
    FAVAE_CHECK_ARG(((uintptr_t)rows % 16) == 0 && ((uintptr_t)cols % 16) == 0 && (!wt || (uintptr_t)wt % 16 == 0), "favae_conv_wgrad_adam: 16-byte alignment");
    *applied = 0;
    if (Rc <= 4 || Cc <= 4) return favae_conv_wgrad(rows, Rc, rows_large, cols, Cc, cols_large, dw, Nb, Hs, Ws, Hl, Wl, ks, stride, pad, splitk, dtype, stream);
    PmParams p;
    p.rp = rows; p.cp = cols; p.dw = dw; p.Rc = Rc; p.Cc = Cc; p.r_large = rows_large; p.c_large = cols_large;
    p.Nb = Nb; p.Hs = Hs; p.Ws = Ws; p.Hl = Hl; p.Wl = Wl; p.ks = ks; p.stride = stride; p.pad = pad;
    p.splitk = splitk; p.ntn = 1; p.K = Nb * Hs * Ws; p.Rst = Rc; p.Cst = Cc;
    const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    p.ad_p = p_; p.ad_m = m_; p.ad_v = v_; p.ad_wc = wc; p.ad_wt = wt;
    p.ad_b1 = beta1; p.ad_b2 = beta2; p.ad_eps = eps; p.ad_step = (float)((double)lr / bc1); p.ad_bc2s = (float)sqrt(bc2);
    const int rc = dtype == FAVAE_F32 ? dispatch_pm<float>(p, (hipStream_t)stream) : dispatch_pm<bf16_t>(p, (hipStream_t)stream);
    if (rc == FAVAE_OK) *applied = g_pm_adam_applied;
    return rc;
}

extern "C" int favae_conv_wgrad_auto_splitk(int Rc, int Cc, long long K, int ks) {
    if (Rc <= 0 || Cc <= 0 || K <= 0 || ks <= 0) return 0;
    if (g_force_pm_tile >= 0) return 0;   // (developer override active: no promise)
    return pm_auto_splitk(Rc, Cc, K, ks, pm_tile_of(Rc, Cc));
}

extern "C" size_t favae_conv_wgrad_folded_scratch(int Nb, int Hl, int Wl, int dtype) {
    return (size_t)Nb * Hl * Wl * 32 * (dtype == FAVAE_F32 ? 4 : 2) + 256;
}

extern "C" int favae_conv_fold_taps(const void* src, void* dst, size_t dst_bytes, int Nb, int Hs, int Ws, int Hl, int Wl, int C,
                                    int stride, int shifted, int dtype, void* stream) {
    FAVAE_CHECK_ARG(src && dst, "favae_conv_fold_taps: null pointer");
    FAVAE_CHECK_ARG(Nb > 0 && Hs > 0 && Ws > 0 && C >= 1 && C <= 3, "favae_conv_fold_taps: bad shape (1..3 channels)");
    FAVAE_CHECK_ARG(stride == 1 || (stride == 2 && !shifted), "favae_conv_fold_taps: stride");
    FAVAE_CHECK_ARG(dtype == FAVAE_F32 || dtype == FAVAE_BF16, "favae_conv_fold_taps: dtype");
    FAVAE_CHECK_ARG(shifted ? (Hs == Hl && Ws == Wl) : (Hs == (Hl + 2 - 3) / stride + 1 && Ws == (Wl + 2 - 3) / stride + 1),
                    "favae_conv_fold_taps: small/large size mismatch");
    FAVAE_CHECK_ARG((size_t)Nb * Hl * Wl < (1u << 31), "favae_conv_fold_taps: too many pixels");
    const size_t need = (size_t)Nb * Hs * Ws * 32 * (dtype == FAVAE_F32 ? 4 : 2);
    FAVAE_CHECK_ARG(need <= dst_bytes && ((uintptr_t)dst % 16) == 0, "favae_conv_fold_taps: dst holds %zu bytes, %zu needed (16-byte aligned)", dst_bytes, need);
    return fold_taps_launch(src, dst, Nb, Hs, Ws, Hl, Wl, C, stride, shifted, dtype, (hipStream_t)stream);
}

extern "C" int favae_conv_folded_fwd(const void* fold, const void* w, const float* bias, float* out, long long P, int K, int Cout,
                                     double* bins, int nbins, int dtype, void* stream) {
    FAVAE_CHECK_ARG(fold && w && out && P > 0 && P < (1ll << 31), "favae_conv_folded_fwd: bad argument");
    FAVAE_CHECK_ARG(dtype == FAVAE_BF16 && Cout == 64 && K >= 1 && K <= 32, "favae_conv_folded_fwd: bf16, 64 outputs, K <= 32");
    FAVAE_CHECK_ARG(!bins || nbins >= 1, "favae_conv_folded_fwd: nbins");
    FAVAE_CHECK_ARG(((uintptr_t)fold % 16) == 0 && ((uintptr_t)out % 16) == 0, "favae_conv_folded_fwd: 16-byte alignment");
    return folded_fwd_launch(fold, w, K, bias, out, P, bins, nbins, 0, (hipStream_t)stream);
}

extern "C" int favae_conv_fold_taps_split(const float* src, void* dst, size_t dst_bytes, int Nb, int Hs, int Ws, int Hl, int Wl, int C,
                                          int stride, void* stream) {
    FAVAE_CHECK_ARG(src && dst, "favae_conv_fold_taps_split: null pointer");
    FAVAE_CHECK_ARG(Nb > 0 && Hs > 0 && Ws > 0 && C >= 1 && C <= 3, "favae_conv_fold_taps_split: bad shape (1..3 channels)");
    FAVAE_CHECK_ARG(stride == 1 || stride == 2, "favae_conv_fold_taps_split: stride");
    FAVAE_CHECK_ARG(Hs == (Hl + 2 - 3) / stride + 1 && Ws == (Wl + 2 - 3) / stride + 1, "favae_conv_fold_taps_split: small/large size mismatch");
    FAVAE_CHECK_ARG((size_t)Nb * Hl * Wl < (1u << 31), "favae_conv_fold_taps_split: too many pixels");
    const size_t need = (size_t)Nb * Hs * Ws * 32 * 2 * 2;
    FAVAE_CHECK_ARG(need <= dst_bytes && ((uintptr_t)dst % 16) == 0, "favae_conv_fold_taps_split: dst holds %zu bytes, %zu needed (16-byte aligned)", dst_bytes, need);
    return fold_taps_hl_launch(src, dst, Nb, Hs, Ws, Hl, Wl, C, stride, (hipStream_t)stream);
}

extern "C" int favae_conv_folded_fwd_split(const void* fold_hl, const float* w, const float* bias, float* out, long long P, int K, int Cout,
                                           double* bins, int nbins, void* stream) {
    FAVAE_CHECK_ARG(fold_hl && w && out && P > 0 && P < (1ll << 30), "favae_conv_folded_fwd_split: bad argument");
    FAVAE_CHECK_ARG(Cout == 64 && K >= 1 && K <= 32, "favae_conv_folded_fwd_split: 64 outputs, K <= 32");
    FAVAE_CHECK_ARG(!bins || nbins >= 1, "favae_conv_folded_fwd_split: nbins");
    FAVAE_CHECK_ARG(((uintptr_t)fold_hl % 16) == 0 && ((uintptr_t)out % 16) == 0, "favae_conv_folded_fwd_split: 16-byte alignment");
    return folded_fwd_launch(fold_hl, w, K, bias, out, P, bins, nbins, 1, (hipStream_t)stream);
}

extern "C" int favae_conv_folded_dgrad(const void* fold, const void* w, void* out, long long P, int K, int Cin, double* bins, int nbins,
                                       const void* act, const void* raw, const float* mean, const float* invstd, const float* ab, int groups,
                                       int dtype, void* stream) {
    FAVAE_CHECK_ARG(fold && w && (out || bins) && P > 0 && P < (1ll << 31) && groups >= 1, "favae_conv_folded_dgrad: bad argument");
    FAVAE_CHECK_ARG((dtype & 0xF) == FAVAE_BF16 && Cin == 64 && K >= 1 && K <= 32, "favae_conv_folded_dgrad: bf16, 64 input channels, K <= 32");
    FAVAE_CHECK_ARG((dtype & ~(0xF | FAVAE_OUT_BF16 | FAVAE_RAW_BF16)) == 0, "favae_conv_folded_dgrad: unknown storage flag in dtype 0x%x", dtype);
    FAVAE_CHECK_ARG(!bins || (nbins >= 1 && (act || ab) && raw && mean && invstd), "favae_conv_folded_dgrad: the sums need nbins, act (or ab), raw, mean, invstd");
    FAVAE_CHECK_ARG(((uintptr_t)fold % 16) == 0 && ((uintptr_t)out % 16) == 0, "favae_conv_folded_dgrad: 16-byte alignment");
    return folded_dgrad_launch(fold, w, K, out, P, groups, (dtype & FAVAE_OUT_BF16) != 0, bins, nbins, act, raw, (dtype & FAVAE_RAW_BF16) != 0, mean,
                               invstd, ab, (hipStream_t)stream);
}

/* The input gradient of favae_conv_folded_dgrad formed again and the backward of the BatchNorm2d(train) + LeakyReLU in front applied to it at
 * once (final_layer.1/.2, celeba_deepvanilla_vae.py:72-73, under loss.backward() :249): draw [P][64] (bf16) = gamma invstd (dd - s1 / R - xhat s2 / R)
 * per BatchNorm group, s1 / s2 the totals of bins [groups][nbins][2][64] that a favae_conv_folded_dgrad call with out == NULL left; dgamma / dbeta
 * are ADDED to (first block of a group); dbias_partial (may be NULL) [*slabs][64] doubles takes the column sums of draw per block, at most
 * slabs_cap of them, for favae_bn_dbias_finish / _ride.  FAVAE_OUT_BF16 in dtype: the gradient is rounded to bf16 before use, as a stored
 * bf16 gradient would be (draw then equals that of favae_conv_folded_dgrad + favae_bn_lrelu_bwd_bins_ab / favae_gbn_bwd up to the odd last bf16 bit). */
extern "C" int favae_conv_folded_dgrad_bn(const void* fold, const void* w, void* draw, long long P, int K, int Cin, const double* bins, int nbins,
                                          const void* act, const void* raw, const float* mean, const float* invstd, const float* gamma, const float* ab,
                                          int groups, float* dgamma, float* dbeta, double* dbias_partial, int slabs_cap, int* slabs, int dtype,
                                          void* stream) {
    FAVAE_CHECK_ARG(fold && w && draw && bins && nbins >= 1 && (act || ab) && raw && mean && invstd && gamma && dgamma && dbeta && P > 0 && P < (1ll << 31) &&
                        groups >= 1, "favae_conv_folded_dgrad_bn: bad argument");
    FAVAE_CHECK_ARG((dtype & 0xF) == FAVAE_BF16 && Cin == 64 && K >= 1 && K <= 32, "favae_conv_folded_dgrad_bn: bf16, 64 input channels, K <= 32");
    FAVAE_CHECK_ARG((dtype & ~(0xF | FAVAE_OUT_BF16 | FAVAE_RAW_BF16)) == 0, "favae_conv_folded_dgrad_bn: unknown storage flag in dtype 0x%x", dtype);
    FAVAE_CHECK_ARG(!dbias_partial || (slabs && slabs_cap >= groups), "favae_conv_folded_dgrad_bn: dbias_partial needs slabs and slabs_cap >= groups");
    FAVAE_CHECK_ARG(((uintptr_t)fold % 16) == 0 && ((uintptr_t)draw % 16) == 0, "favae_conv_folded_dgrad_bn: 16-byte alignment");
    return folded_dgrad_bn_launch(fold, w, K, draw, P, groups, (dtype & FAVAE_OUT_BF16) != 0, bins, nbins, act, raw, (dtype & FAVAE_RAW_BF16) != 0, mean,
                                  invstd, gamma, ab, dgamma, dbeta, dbias_partial, slabs_cap, slabs, (hipStream_t)stream);
}

extern "C" int favae_conv_wgrad_prefolded(const void* rows, int Rc, int Rst, const void* cols, int Cc, int Cst, float* dw,
                                          long long K, int splitk, int dtype, void* stream) {
    FAVAE_CHECK_ARG(rows && cols && dw, "favae_conv_wgrad_prefolded: null pointer");
    FAVAE_CHECK_ARG(K > 0 && K < (1ll << 31) && Rc > 0 && Cc > 0 && Rst >= 1 && Rst <= Rc && Cst >= 1 && Cst <= Cc, "favae_conv_wgrad_prefolded: bad shape");
    FAVAE_CHECK_ARG(dtype == FAVAE_F32 || dtype == FAVAE_BF16, "favae_conv_wgrad_prefolded: dtype");
    const int ve = dtype == FAVAE_F32 ? 4 : 8;
    FAVAE_CHECK_ARG(Rc % ve == 0 && Cc % ve == 0, "favae_conv_wgrad_prefolded: channel counts must fill 16-byte vectors");
    FAVAE_CHECK_ARG(((uintptr_t)rows % 16) == 0 && ((uintptr_t)cols % 16) == 0, "favae_conv_wgrad_prefolded: 16-byte alignment");
    PmParams p;
    p.rp = rows; p.cp = cols; p.dw = dw; p.Rc = Rc; p.Cc = Cc; p.Rst = Rst; p.Cst = Cst; p.r_large = 0; p.c_large = 0;
    p.Nb = 1; p.Hs = 1; p.Ws = (int)K; p.Hl = 1; p.Wl = (int)K; p.ks = 1; p.stride = 1; p.pad = 0; p.splitk = splitk; p.ntn = 1; p.K = (int)K;
    p.ad_p = nullptr;
    return dtype == FAVAE_F32 ? dispatch_pm<float>(p, (hipStream_t)stream) : dispatch_pm<bf16_t>(p, (hipStream_t)stream);
}

/* favae_conv_wgrad_prefolded whose column operand is the conv output cols_raw [K][Cc] (bf16) in front of a BatchNorm2d(train) + LeakyReLU
 * (final_layer.1/.2 in front of final_layer.3, celeba_deepvanilla_vae.py:72-76): the activation lrelu(x * a + b), rounded to bf16, is formed at
 * the operand load from ab [groups][2][Cc] (a | b per BatchNorm group of group_rows rows of cols_raw; 0: one group).  bf16, Rc <= 32, K and
 * group_rows multiples of 64. */
extern "C" int favae_conv_wgrad_prefolded_bn(const void* rows, int Rc, int Rst, const void* cols_raw, int Cc, int Cst, const float* ab,
                                             long long group_rows, float* dw, long long K, int splitk, int dtype, void* stream) {
    FAVAE_CHECK_ARG(rows && cols_raw && dw && ab, "favae_conv_wgrad_prefolded_bn: null pointer");
    FAVAE_CHECK_ARG(K > 0 && K < (1ll << 31) && Rc > 0 && Rc <= 32 && Cc > 0 && Rst >= 1 && Rst <= Rc && Cst >= 1 && Cst <= Cc, "favae_conv_wgrad_prefolded_bn: bad shape");
    FAVAE_CHECK_ARG(dtype == FAVAE_BF16 && Rc % 8 == 0 && Cc % 8 == 0, "favae_conv_wgrad_prefolded_bn: bf16, channel counts that fill 16-byte vectors");
    FAVAE_CHECK_ARG(K % 64 == 0 && group_rows >= 0 && group_rows % 64 == 0 && group_rows < (1ll << 31), "favae_conv_wgrad_prefolded_bn: K and group_rows multiples of 64");
    FAVAE_CHECK_ARG(((uintptr_t)rows % 16) == 0 && ((uintptr_t)cols_raw % 16) == 0, "favae_conv_wgrad_prefolded_bn: 16-byte alignment");
    FAVAE_CHECK_ARG(g_force_pm_tile < 0 || g_force_pm_tile == 0, "favae_conv_wgrad_prefolded_bn: the developer tile override excludes this form");
    PmParams p;
    p.rp = rows; p.cp = cols_raw; p.dw = dw; p.Rc = Rc; p.Cc = Cc; p.Rst = Rst; p.Cst = Cst; p.r_large = 0; p.c_large = 0;
    p.Nb = 1; p.Hs = 1; p.Ws = (int)K; p.Hl = 1; p.Wl = (int)K; p.ks = 1; p.stride = 1; p.pad = 0; p.splitk = splitk; p.ntn = 1; p.K = (int)K;
    p.ad_p = nullptr;
    p.c_ab = ab; p.c_ab_group_pix = (int)group_rows;
    return dispatch_pm<bf16_t>(p, (hipStream_t)stream);
}

extern "C" int favae_conv_wgrad_folded(const void* rows, int Rc, const void* cols, int Cc, float* dw, int Nb, int Hs, int Ws,
                                       int Hl, int Wl, int stride, int splitk, int dtype, void* scratch, size_t scratch_bytes,
                                       void* stream) {
    FAVAE_CHECK_ARG(rows && cols && dw && scratch, "favae_conv_wgrad_folded: null pointer");
    FAVAE_CHECK_ARG(Nb > 0 && Hs > 0 && Ws > 0 && Rc > 0 && Cc > 0, "favae_conv_wgrad_folded: bad shape");
    FAVAE_CHECK_ARG(stride == 1 || stride == 2, "favae_conv_wgrad_folded: stride");
    FAVAE_CHECK_ARG(dtype == FAVAE_F32 || dtype == FAVAE_BF16, "favae_conv_wgrad_folded: dtype");
    FAVAE_CHECK_ARG(Hs == (Hl + 2 - 3) / stride + 1 && Ws == (Wl + 2 - 3) / stride + 1, "favae_conv_wgrad_folded: small/large size mismatch");
    const bool fold_cols = Cc <= 3, fold_rows = !fold_cols && Rc <= 3 && stride == 1;
    FAVAE_CHECK_ARG(fold_cols || fold_rows, "favae_conv_wgrad_folded: needs <= 3 channels on the image side (cols) or, at stride 1, on rows");
    if (fold_cols) {
        // dw[r][tap][c] = sum_pix rows[pix][r] * cols[nb(pix, tap)][c]: columns (tap, c) gathered once into [pix][32]
        int rc = favae_conv_fold_taps(cols, scratch, scratch_bytes, Nb, Hs, Ws, Hl, Wl, Cc, stride, 0, dtype, stream);
        if (rc) return rc;
        return favae_conv_wgrad_prefolded(rows, Rc, Rc, scratch, 32, 9 * Cc, dw, (long long)Nb * Hs * Ws, splitk, dtype, stream);
    }
    // stride 1: the same sum over the pixels of the wide operand, the narrow one shifted: rows (r, tap) into [pix'][32]
    int rc = favae_conv_fold_taps(rows, scratch, scratch_bytes, Nb, Hl, Wl, Hs, Ws, Rc, 1, 1, dtype, stream);
    if (rc) return rc;
    return favae_conv_wgrad_prefolded(scratch, 32, 9 * Rc, cols, Cc, Cc, dw, (long long)Nb * Hl * Wl, splitk, dtype, stream);
}

#ifdef FAVAE_KC_STAMPS
extern "C" int favae_debug_kc_stamps(unsigned long long* out64) {
    FAVAE_HIP(hipDeviceSynchronize());
    FAVAE_HIP(hipMemcpyFromSymbol(out64, HIP_SYMBOL(g_kc_stamps), 64 * sizeof(unsigned long long)));
    return FAVAE_OK;
}
extern "C" int favae_debug_kc_blocks(unsigned long long* out16384) {
    FAVAE_HIP(hipDeviceSynchronize());
    FAVAE_HIP(hipMemcpyFromSymbol(out16384, HIP_SYMBOL(g_kc_blocks), 8192 * 2 * sizeof(unsigned long long)));
    return FAVAE_OK;
}
#endif

extern "C" int favae_bn_dbias_ride(const double* partial, int nslab, int C, float* dbias) {
    FAVAE_CHECK_ARG(partial && dbias && nslab >= 1 && nslab <= 256 && C >= 1 && C <= 2048, "favae_bn_dbias_ride: bad argument");
    favae_slab_rider() = SlabRider{partial, dbias, nslab, C, nullptr, 0.0, 0.0, C};
    return FAVAE_OK;
}

extern "C" int favae_conv_out_ll_ride(const double* partial, int Nb, int H, int W, int Cout, float var_x, double* rec_err, float* dbias) {
    FAVAE_CHECK_ARG(partial && rec_err && dbias && Nb > 0 && H > 0 && W > 0 && Cout >= 1 && Cout <= 4 && var_x > 0, "favae_conv_out_ll_ride: bad argument");
    const long long blocks = ((long long)Nb * H * W + 255) / 256;
    FAVAE_CHECK_ARG(blocks <= 65536, "favae_conv_out_ll_ride: %lld blocks", blocks);
    const double D = (double)H * W * Cout;
    favae_slab_rider() = SlabRider{partial, dbias, (int)blocks, 5, rec_err, 0.5 * (D * log(2.0 * M_PI) + D * log((double)var_x)),
                                   0.5 / ((double)var_x * Nb), Cout};
    return FAVAE_OK;
}

extern "C" int favae_debug_force_tile(int kc_tile, int pm_tile) {
    g_force_kc_tile = kc_tile;
    g_force_pm_tile = pm_tile;
    return FAVAE_OK;
}

