// Weight gradient of a 3x3 convolution whose output (or input) has an image-sized channel count (<= 4), e.g.
// final_layer.3 = Conv2d(64 -> 3, k3 s1 p1) of the FA-VAE decoder (celeba_deepvanilla_vae.py:74-76).  With 3 channels
// on one side the GEMM is 27 x 64 with a reduction over every pixel of the batch: far below an MFMA tile and bound by
// streaming the wide operand once, so it runs on the vector ALUs:
//   dw[..] += sum_q wide[q][ch] * narrow[coord(q, tap)][c]      c < Cs <= 4, 4 wide channels per lane
// coord(q, tap) = q*stride - 1 + k (narrow on the large grid) or q + 1 - k (narrow on the small grid, stride 1).
// Border taps use a clamped coordinate and a 0/1 mask, so all loads of a pixel are issued together.
#include <algorithm>
#include <atomic>
#include <mutex>
#include <cstdlib>
#include "common.h"
#include <type_traits>
#include "smallconv.h"

namespace {

template <class T> __device__ __forceinline__ void load4(const T* p, float* o);
template <> __device__ __forceinline__ void load4<float>(const float* p, float* o) {
    const float4 v = *(const float4*)p; o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
}
template <> __device__ __forceinline__ void load4<bf16_t>(const bf16_t* p, float* o) {
    const uint2 u = *(const uint2*)p;
    o[0] = __uint_as_float(u.x << 16); o[1] = __uint_as_float(u.x & 0xffff0000u);
    o[2] = __uint_as_float(u.y << 16); o[3] = __uint_as_float(u.y & 0xffff0000u);
}

__device__ __forceinline__ int coord(int q, int k, int stride, int flip) { return flip ? q + 1 - k : q * stride - 1 + k; }

// Border taps are handled without control flow: the coordinate is clamped into the image and the value multiplied by
// a 0/1 mask, so that all loads of one pixel are issued back to back and waited for once.
__device__ __forceinline__ int clampi(int v, int hi) { return min(max(v, 0), hi - 1); }

// ---------------------------------------------------------------------------------------------------------
template <class T, int CS>
__global__ __launch_bounds__(256) void wgrad_nw(const T* __restrict__ wide, const T* __restrict__ narrow, float* __restrict__ dw,
                                                int Hq, int Wq, int Cw, int Hn, int Wn, int stride, int flip, int total, int ppb,
                                                int s_ch, int s_tap, int s_c) {
    __shared__ float red[256 * 4];
    const int tid = threadIdx.x;
    const int L = Cw >> 2, PP = 256 / L, l = tid % L, pl = tid / L;
    float acc[9 * CS][4];
#pragma unroll
    for (int i = 0; i < 9 * CS; ++i) acc[i][0] = acc[i][1] = acc[i][2] = acc[i][3] = 0.f;
    const int q0 = blockIdx.x * ppb, q1 = min(total, q0 + ppb);
    for (int q = q0 + pl; q < q1; q += PP) {
        const int n = q / (Hq * Wq), rem = q - n * Hq * Wq, y = rem / Wq, x = rem - y * Wq;
        float wv[4], v[9 * CS];
        load4<T>(wide + (size_t)q * Cw + l * 4, wv);
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const int iy = coord(y, kh, stride, flip);
            const bool oky = iy >= 0 && iy < Hn;
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int ix = coord(x, kw, stride, flip);
                const float m = (oky && ix >= 0 && ix < Wn) ? 1.f : 0.f;
                const T* src = narrow + ((size_t)(n * Hn + clampi(iy, Hn)) * Wn + clampi(ix, Wn)) * CS;
#pragma unroll
                for (int c = 0; c < CS; ++c) v[(kh * 3 + kw) * CS + c] = to_f32<T>(src[c]) * m;
            }
        }
#pragma unroll
        for (int k = 0; k < 9 * CS; ++k) {
            acc[k][0] += v[k] * wv[0]; acc[k][1] += v[k] * wv[1]; acc[k][2] += v[k] * wv[2]; acc[k][3] += v[k] * wv[3];
        }
    }
    // block reduction over the PP pixel lanes, one (tap, c) at a time, into an LDS image of the whole output; then one
    // pass of atomics in output order (256 contiguous bytes per wave instruction, the full-rate shape)
    __shared__ float S[36 * 256];
#pragma unroll
    for (int i = 0; i < 9 * CS; ++i) {
        *(float4*)(red + tid * 4) = make_float4(acc[i][0], acc[i][1], acc[i][2], acc[i][3]);
        __syncthreads();
        if (tid < Cw) {
            float s = 0.f;
            for (int p = 0; p < PP; ++p) s += red[(p * L + (tid >> 2)) * 4 + (tid & 3)];
            S[i * Cw + tid] = s;
        }
        __syncthreads();
    }
    const int nout = 9 * CS * Cw;
    for (int o = tid; o < nout; o += 256) {
        int ch, tap, c;
        if (s_ch == 1) { ch = o % Cw; const int r = o / Cw; c = r / 9; tap = r - c * 9; }       // dw[c][tap][ch]
        else { ch = o / (9 * CS); const int r = o - ch * 9 * CS; tap = r / CS; c = r - tap * CS; }   // dw[ch][tap][c]
        atomicAdd(dw + o, S[(tap * CS + c) * Cw + ch]);
    }
}

inline bool pow2(int x) { return x > 0 && (x & (x - 1)) == 0; }

}  // namespace

bool wgrad_nw_applicable(int Cw, int Cs, int ks, int stride, int narrow_is_large) {
    return ks == 3 && Cs >= 1 && Cs <= 4 && Cw % 4 == 0 && pow2(Cw / 4) && Cw >= 16 && Cw <= 256 && (narrow_is_large || stride == 1);
}

template <class T>
static int wgrad_nw_dispatch(const void* wide, const void* narrow, float* dw, int Hq, int Wq, int Cw, int Hn, int Wn, int Cs,
                             int stride, int flip, int total, int ppb, int grid, int s_ch, int s_tap, int s_c, hipStream_t st) {
#define NW_CASE(CS_)                                                                                                        \
    case CS_:                                                                                                               \
        hipLaunchKernelGGL((wgrad_nw<T, CS_>), dim3(grid), dim3(256), 0, st, (const T*)wide, (const T*)narrow, dw, Hq, Wq, Cw, \
                           Hn, Wn, stride, flip, total, ppb, s_ch, s_tap, s_c);                                             \
        break;
    switch (Cs) { NW_CASE(1) NW_CASE(2) NW_CASE(3) NW_CASE(4) }
#undef NW_CASE
    FAVAE_LAUNCH_CHECK();
    return FAVAE_OK;
}

int wgrad_nw_launch(const void* wide, const void* narrow, float* dw, int Nb, int Hq, int Wq, int Cw, int Hn, int Wn, int Cs,
                    int stride, int narrow_is_large, int s_ch, int s_tap, int s_c, int dtype, hipStream_t st) {
    const int total = Nb * Hq * Wq;
    const int PP = 256 / (Cw / 4);
    // >= 256 pixels per block: the block epilogue (27 LDS reductions + one atomic pass) is worth ~16 pixel iterations
    static const int nw_blocks = getenv("FAVAE_NW_BLOCKS") ? atoi(getenv("FAVAE_NW_BLOCKS")) : 256;
    int ppb = std::max(256, cdiv(cdiv(total, nw_blocks), PP) * PP);
    const int grid = cdiv(total, ppb);
    const int flip = narrow_is_large ? 0 : 1;
    return dtype == FAVAE_F32
               ? wgrad_nw_dispatch<float>(wide, narrow, dw, Hq, Wq, Cw, Hn, Wn, Cs, stride, flip, total, ppb, grid, s_ch, s_tap, s_c, st)
               : wgrad_nw_dispatch<bf16_t>(wide, narrow, dw, Hq, Wq, Cw, Hn, Wn, Cs, stride, flip, total, ppb, grid, s_ch, s_tap, s_c, st);
}

// ---------------------------------------------------------------------------------------------------------
// Convolution with an image-sized INPUT channel count (Cs <= 4) and a wide output (Co = 64): encoder.0 forward
// (k3 s2 p1, celeba_deepvanilla_vae.py:31-33) and the input gradient of final_layer.3 (k3 s1 p1 with the transposed
// weights, loss.backward() :249).  The reduction is only 9*Cs <= 36 long, so one thread owns one output pixel and all
// Co channels: its 9*Cs inputs sit in registers, the weights are wave-uniform (scalar loads, SGPR operands of the
// FMAs), and nothing is shared between threads.  VALU-bound at 9*Cs*Co FMAs per pixel.
//   out[pix][co] = bias[co] + sum_{tap, c} in[coord(pix, tap)][c] * w[co][tap][c]
namespace {

template <class T, int CS, int CO, bool RED>
__global__ __launch_bounds__(256) void conv_thin_in(const T* __restrict__ in, const float* __restrict__ wk /* [9*CS][CO] fp32 */,
                                                    const float* __restrict__ bias, float* __restrict__ out, int Hi, int Wi,
                                                    int Ho, int Wo, int stride, int flip, int total, int out_bf16, const ThinRed red,
                                                    const FlagCarry carry) {
    carry_store(carry);
    const int pix = min(blockIdx.x * 256 + (int)threadIdx.x, total - 1);   // tail threads redo the last pixel, store nothing
    const int n = pix / (Ho * Wo), rem = pix - n * Ho * Wo, y = rem / Wo, x = rem - y * Wo;
    float v[9 * CS];
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
        const int iy = coord(y, kh, stride, flip);
        const bool oky = iy >= 0 && iy < Hi;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
            const int ix = coord(x, kw, stride, flip);
            const float m = (oky && ix >= 0 && ix < Wi) ? 1.f : 0.f;
            const T* src = in + ((size_t)(n * Hi + clampi(iy, Hi)) * Wi + clampi(ix, Wi)) * CS;
#pragma unroll
            for (int c = 0; c < CS; ++c) v[(kh * 3 + kw) * CS + c] = to_f32<T>(src[c]) * m;
        }
    }
    // results go through a per-wave LDS tile [64 pixels][32 channels] so that every global store instruction writes
    // whole 128-B segments of the pixel rows (a thread storing its own row 16 B at a time costs 3x the HBM write
    // traffic: partial lines)
    extern __shared__ __attribute__((aligned(16))) float tile[];
    constexpr int HC = 32, RSF = HC + 4;   // channels per pass, floats per LDS row (+16 B against bank conflicts)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* wt = tile + wave * 64 * RSF;
    const int pix_w0 = blockIdx.x * 256 + wave * 64;   // first pixel of this wave
    constexpr int LPR = HC / 4;                        // lanes per pixel row (float4 each)
    constexpr int PPI = 64 / LPR;                      // pixel rows per store instruction
    __shared__ double redsm[4 * 2 * HC];
    using ActVec = typename std::conditional<sizeof(T) == 2, uint2, float4>::type;
    for (int hb = 0; hb < CO; hb += HC) {
        // optional BatchNorm-backward sums of the layer this gradient belongs to (see KcParams::red_mode 2 in igemm.hip):
        // a lane keeps its 4 channels over the 8 pixel rows it stores per pass; their act / raw values are fetched here,
        // ahead of the FMAs, so that the store loop does not wait on memory
        ActVec pa[64 / PPI];
        float4 px[64 / PPI];
        float4 mu = make_float4(0.f, 0.f, 0.f, 0.f), is = mu;
        const int rgrp = (RED && red.group_pixels) ? (int)(blockIdx.x * 256u) / red.group_pixels : 0;   // block-uniform
        if (RED) {
            mu = *(const float4*)(red.mean + rgrp * CO + hb + (lane % LPR) * 4);
            is = *(const float4*)(red.invstd + rgrp * CO + hb + (lane % LPR) * 4);
#pragma unroll
            for (int i = 0; i < 64 / PPI; ++i) {
                const int prow = min(pix_w0 + i * PPI + lane / LPR, total - 1);
                const size_t e = (size_t)prow * CO + hb + (lane % LPR) * 4;
                pa[i] = *(const ActVec*)((const T*)red.act + e);
                if (red.raw_bf16) {   // block-uniform
                    const uint2 u = *(const uint2*)((const bf16_t*)red.raw + e);
                    px[i] = make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u));
                } else {
                    px[i] = *(const float4*)((const float*)red.raw + e);
                }
            }
        }
#pragma unroll
        for (int cb = 0; cb < HC; cb += 16) {   // 16 output channels at a time keeps the accumulators at 16 registers
            float acc[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[j] = bias ? bias[hb + cb + j] : 0.f;
#pragma unroll
            for (int k = 0; k < 9 * CS; ++k)
#pragma unroll
                for (int j = 0; j < 16; ++j) acc[j] += v[k] * wk[k * CO + hb + cb + j];
#pragma unroll
            for (int j = 0; j < 16; j += 4) *(float4*)(wt + lane * RSF + cb + j) = make_float4(acc[j], acc[j + 1], acc[j + 2], acc[j + 3]);
        }
        // wave-local hand-over: the tile is written and read by the same wave
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        double rs[4] = {0.0, 0.0, 0.0, 0.0}, rq[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int i = 0; i < 64 / PPI; ++i) {
            const int pr = i * PPI + lane / LPR, c4 = (lane % LPR) * 4;
            if (pix_w0 + pr < total) {
                const float4 r4 = *(const float4*)(wt + pr * RSF + c4);
                if (RED) {
                    float a[4];
                    if constexpr (sizeof(T) == 2) {
                        a[0] = __uint_as_float(pa[i].x << 16); a[1] = __uint_as_float(pa[i].x & 0xffff0000u);
                        a[2] = __uint_as_float(pa[i].y << 16); a[3] = __uint_as_float(pa[i].y & 0xffff0000u);
                    } else {
                        a[0] = pa[i].x; a[1] = pa[i].y; a[2] = pa[i].z; a[3] = pa[i].w;
                    }
                    const float4 xr = px[i];
                    const float d0 = r4.x * (a[0] > 0.f ? 1.f : 0.01f), d1 = r4.y * (a[1] > 0.f ? 1.f : 0.01f);
                    const float d2 = r4.z * (a[2] > 0.f ? 1.f : 0.01f), d3 = r4.w * (a[3] > 0.f ? 1.f : 0.01f);
                    rs[0] += d0; rs[1] += d1; rs[2] += d2; rs[3] += d3;
                    rq[0] += (double)d0 * ((xr.x - mu.x) * is.x); rq[1] += (double)d1 * ((xr.y - mu.y) * is.y);
                    rq[2] += (double)d2 * ((xr.z - mu.z) * is.z); rq[3] += (double)d3 * ((xr.w - mu.w) * is.w);
                }
                if (out_bf16) {
                    uint2 u;
                    u.x = pack_bf16x2(r4.x, r4.y);
                    u.y = pack_bf16x2(r4.z, r4.w);
                    *(uint2*)((bf16_t*)out + (size_t)(pix_w0 + pr) * CO + hb + c4) = u;
                } else {
                    *(float4*)(out + (size_t)(pix_w0 + pr) * CO + hb + c4) = r4;
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (RED) {   // block-uniform
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                for (int o = LPR; o < 64; o <<= 1) { rs[j] += __shfl_xor(rs[j], o, 64); rq[j] += __shfl_xor(rq[j], o, 64); }
            }
            if (lane < LPR) {
#pragma unroll
                for (int j = 0; j < 4; ++j) { redsm[(wave * 2 + 0) * HC + lane * 4 + j] = rs[j]; redsm[(wave * 2 + 1) * HC + lane * 4 + j] = rq[j]; }
            }
            __syncthreads();
            if (threadIdx.x < 2 * HC) {
                const int qq = threadIdx.x / HC, col = threadIdx.x % HC;
                const double t = (redsm[(0 * 2 + qq) * HC + col] + redsm[(1 * 2 + qq) * HC + col]) +
                                 (redsm[(2 * 2 + qq) * HC + col] + redsm[(3 * 2 + qq) * HC + col]);
                unsafeAtomicAdd(red.bins + (((size_t)rgrp * red.nbin + blockIdx.x % (unsigned)red.nbin) * 2 + qq) * CO + hb + col, t);
            }
            __syncthreads();
        }
    }
}

// w [CO][9][CS] (dtype) -> wk [9*CS][CO] fp32
template <class T>
__global__ void thin_in_wprep(const T* __restrict__ w, float* __restrict__ wk, int K, int CO) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= K * CO) return;
    const int k = i / CO, co = i - k * CO;
    wk[i] = to_f32<T>(w[(size_t)co * K + k]);
}

}  // namespace

bool thin_in_applicable(int Cs, int Co, int ks, int stride, int scatter) {
    return ks == 3 && Cs >= 1 && Cs <= 4 && Co == 64 && (scatter ? stride == 1 : true);
}

namespace {
// Re-laid-out weights of the launches in flight: one slot per (device, stream).  Launches of one stream run in order, so the
// re-layout of a launch cannot overtake the convolution of the previous one that reads the same slot; launches of different
// streams never share a slot.  A process that uses more than WK_SLOTS - 1 streams for this kernel shares the last slot among
// the rest, and those launches wait for the device first (correct, slow, and not a case the executors produce).
constexpr int WK_SLOTS = 16, WK_FLOATS = 36 * 64;
__device__ float g_thin_wk[WK_SLOTS * WK_FLOATS];
struct WkOwner { int dev; hipStream_t st; };
std::mutex g_wk_mu;
WkOwner g_wk_owner[WK_SLOTS];
int g_wk_owners = 0;

int wk_slot(int dev, hipStream_t st, bool* shared) {
    std::lock_guard<std::mutex> lk(g_wk_mu);
    *shared = false;
    int free_slot = -1;
    for (int i = 0; i < g_wk_owners; ++i) {
        if (g_wk_owner[i].dev == dev && g_wk_owner[i].st == st) return i;
        if (g_wk_owner[i].dev < 0 && free_slot < 0) free_slot = i;
    }
    if (free_slot >= 0) { g_wk_owner[free_slot] = WkOwner{dev, st}; return free_slot; }
    if (g_wk_owners < WK_SLOTS - 1) {
        g_wk_owner[g_wk_owners] = WkOwner{dev, st};
        return g_wk_owners++;
    }
    *shared = true;
    return WK_SLOTS - 1;
}
}  // namespace

// The stream is about to be destroyed (its work has completed): its slot may serve another stream.  Without this a process that
// creates executors over and over ran out of slots, and every further launch of the kernel synchronised the device.
void thin_in_release_stream(hipStream_t st) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return;
    std::lock_guard<std::mutex> lk(g_wk_mu);
    for (int i = 0; i < g_wk_owners; ++i)
        if (g_wk_owner[i].dev == dev && g_wk_owner[i].st == st) g_wk_owner[i] = WkOwner{-1, nullptr};
}

int thin_in_launch(const void* in, const void* w, const float* bias, float* out, int Nb, int Hi, int Wi, int Cs, int Ho, int Wo,
                   int Co, int stride, int scatter, int dtype, int out_bf16, const ThinRed& red, hipStream_t st) {
    static float* per_device[64] = {nullptr};
    int dev = 0;
    FAVAE_HIP(hipGetDevice(&dev));
    FAVAE_CHECK_ARG(dev >= 0 && dev < 64, "thin_in_launch: device index %d", dev);
    if (!per_device[dev]) FAVAE_HIP(hipGetSymbolAddress((void**)&per_device[dev], HIP_SYMBOL(g_thin_wk)));
    bool shared = false;
    float* wk_scratch = per_device[dev] + (size_t)wk_slot(dev, st, &shared) * WK_FLOATS;
    if (shared) FAVAE_HIP(hipDeviceSynchronize());
    const int total = Nb * Ho * Wo, K = 9 * Cs;
    const int grid = cdiv(total, 256);
    const FlagCarry carry = favae_take_flag_carry();
    if (dtype == FAVAE_F32)
        hipLaunchKernelGGL(thin_in_wprep<float>, dim3(cdiv(K * Co, 256)), dim3(256), 0, st, (const float*)w, wk_scratch, K, Co);
    else
        hipLaunchKernelGGL(thin_in_wprep<bf16_t>, dim3(cdiv(K * Co, 256)), dim3(256), 0, st, (const bf16_t*)w, wk_scratch, K, Co);
#define TI_CASE(T_, CS_)                                                                                                   \
    if (red.bins)                                                                                                          \
        hipLaunchKernelGGL((conv_thin_in<T_, CS_, 64, true>), dim3(grid), dim3(256), 4 * 64 * (32 + 4) * sizeof(float), st, (const T_*)in, wk_scratch, bias, out, Hi, Wi, \
                           Ho, Wo, stride, scatter, total, out_bf16, red, carry);                                          \
    else                                                                                                                   \
        hipLaunchKernelGGL((conv_thin_in<T_, CS_, 64, false>), dim3(grid), dim3(256), 4 * 64 * (32 + 4) * sizeof(float), st, (const T_*)in, wk_scratch, bias, out, Hi, Wi, \
                           Ho, Wo, stride, scatter, total, out_bf16, red, carry)
    if (dtype == FAVAE_F32) {
        switch (Cs) { case 1: TI_CASE(float, 1); break; case 2: TI_CASE(float, 2); break; case 3: TI_CASE(float, 3); break; default: TI_CASE(float, 4); }
    } else {
        switch (Cs) { case 1: TI_CASE(bf16_t, 1); break; case 2: TI_CASE(bf16_t, 2); break; case 3: TI_CASE(bf16_t, 3); break; default: TI_CASE(bf16_t, 4); }
    }
#undef TI_CASE
    FAVAE_LAUNCH_CHECK();
    return FAVAE_OK;
}

// ---------------------------------------------------------------------------------------------------------
// Forward convolution with an image-sized input channel count over the tap-FOLDED input (bf16): out [P][64] =
// fold [P][32] x w [64][K]^T + bias, K = 9 * Cs <= 27 (columns K..31 of fold are zero).  encoder.0 of the training step: the
// folded batch exists anyway (its weight gradient reads it) and is built before the step begins, so the layer becomes a
// plain K = 32 GEMM on the matrix cores -- 4 MFMAs (32x32x16) per 32 pixels instead of 27 * 64 FMAs per pixel -- bound by
// its 16.8 MB fp32 output.  A wave owns 32 pixels x 64 outputs per turn: the A fragments come straight from global memory
// (a pixel row is 64 contiguous bytes), the weights sit in LDS (zero padded to [64][32]), the result leaves from the
// accumulators (a store instruction writes two whole 128-byte pixel-row halves), and the per-channel sums of the BatchNorm
// statistics (sum, sum of squares; fp64) go to bins[bin][2][64] like the epilogue sums of igemm_kc (red_mode 1).
namespace {
__device__ __forceinline__ f32x16 mfma_bf16_32x32x16(bf16x8 a, bf16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// SPLIT: the operands carry (almost) fp32 precision as two bf16 terms each.  `fold` holds two planes of P rows -- the bf16
// rounding of the fp32 pixels and the bf16 rounding of what that rounding dropped (fold_taps_hl) -- and `w` is the fp32 master
// weight, split the same way into LDS here: out = x_hi w_hi + x_lo w_hi + x_hi w_lo (the dropped x_lo w_lo term is 2^-18 relative).
// 12 MFMAs per 32 pixels instead of 4; the launch stays bound by its fp32 output.  The rounding of the IMAGE to bf16 was the largest
// single contribution to the error of the bf16 path's latent means (it passes through all five BatchNorm layers): tools/dbg_parity_r3.py.
template <bool SPLIT>
__global__ __launch_bounds__(256) void folded_fwd_kernel(const bf16_t* __restrict__ fold, const void* __restrict__ w_, int K,
                                                         const float* __restrict__ bias, float* __restrict__ out, int P,
                                                         double* __restrict__ bins, int nbin, const FlagCarry carry, int plane) {
    carry_store(carry);
    // blockIdx.y = BatchNorm group (grouped-batch mode): P pixels of the `plane` pixels of the launch, bins of its own
    fold += (size_t)blockIdx.y * P * 32;
    out += (size_t)blockIdx.y * P * 64;
    if (bins) bins += (size_t)blockIdx.y * nbin * 2 * 64;
    __shared__ __attribute__((aligned(16))) bf16_t wl[(SPLIT ? 2 : 1) * 64 * 40];   // [cout][32] + 8 columns of padding (80-byte rows: conflict-free 16-B reads)
    __shared__ double redsm[4 * 2 * 64];
    for (int i = threadIdx.x; i < 64 * 32; i += 256) {
        const int co = i >> 5, k = i & 31;
        if constexpr (SPLIT) {
            const float wf = k < K ? ((const float*)w_)[co * K + k] : 0.f;
            const bf16_t hi = f32_to_bf16(wf);
            wl[co * 40 + k] = hi;
            wl[64 * 40 + co * 40 + k] = f32_to_bf16(wf - bf16_to_f32(hi));
        } else {
            wl[co * 40 + k] = k < K ? ((const bf16_t*)w_)[co * K + k] : (bf16_t)0;
        }
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, h = lane >> 5;
    union U { uint4 u; bf16x8 b; };
    U b00, b01, b10, b11;   // [n tile][k step]
    b00.u = *(const uint4*)(wl + r * 40 + h * 8);
    b01.u = *(const uint4*)(wl + r * 40 + 16 + h * 8);
    b10.u = *(const uint4*)(wl + (32 + r) * 40 + h * 8);
    b11.u = *(const uint4*)(wl + (32 + r) * 40 + 16 + h * 8);
    [[maybe_unused]] U l00, l01, l10, l11;   // the low halves of the weights
    if constexpr (SPLIT) {
        const bf16_t* wl2 = wl + 64 * 40;
        l00.u = *(const uint4*)(wl2 + r * 40 + h * 8);
        l01.u = *(const uint4*)(wl2 + r * 40 + 16 + h * 8);
        l10.u = *(const uint4*)(wl2 + (32 + r) * 40 + h * 8);
        l11.u = *(const uint4*)(wl2 + (32 + r) * 40 + 16 + h * 8);
    }
    const float bias0 = bias ? bias[r] : 0.f, bias1 = bias ? bias[32 + r] : 0.f;
    double s0 = 0.0, q0 = 0.0, s1 = 0.0, q1 = 0.0;   // sums of this lane's two output channels (r, 32 + r) over its pixel rows
    const int ngroup = (P + 31) / 32, nw = gridDim.x * 4;
    for (int g = blockIdx.x * 4 + wave; g < ngroup; g += nw) {
        const int p0 = g * 32, pa = min(p0 + r, P - 1);
        U a0, a1;
        a0.u = *(const uint4*)(fold + (size_t)pa * 32 + h * 8);
        a1.u = *(const uint4*)(fold + (size_t)pa * 32 + 16 + h * 8);
        f32x16 c0, c1;
#pragma unroll
        for (int i = 0; i < 16; ++i) { c0[i] = bias0; c1[i] = bias1; }
        if constexpr (SPLIT) {   // smallest terms first
            U x0, x1;
            x0.u = *(const uint4*)(fold + ((size_t)plane + pa) * 32 + h * 8);
            x1.u = *(const uint4*)(fold + ((size_t)plane + pa) * 32 + 16 + h * 8);
            c0 = mfma_bf16_32x32x16(x0.b, b00.b, c0);
            c0 = mfma_bf16_32x32x16(x1.b, b01.b, c0);
            c1 = mfma_bf16_32x32x16(x0.b, b10.b, c1);
            c1 = mfma_bf16_32x32x16(x1.b, b11.b, c1);
            c0 = mfma_bf16_32x32x16(a0.b, l00.b, c0);
            c0 = mfma_bf16_32x32x16(a1.b, l01.b, c0);
            c1 = mfma_bf16_32x32x16(a0.b, l10.b, c1);
            c1 = mfma_bf16_32x32x16(a1.b, l11.b, c1);
        }
        c0 = mfma_bf16_32x32x16(a0.b, b00.b, c0);
        c0 = mfma_bf16_32x32x16(a1.b, b01.b, c0);
        c1 = mfma_bf16_32x32x16(a0.b, b10.b, c1);
        c1 = mfma_bf16_32x32x16(a1.b, b11.b, c1);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int p = p0 + (i & 3) + 8 * (i >> 2) + 4 * h;
            if (p < P) {
                out[(size_t)p * 64 + r] = c0[i];
                out[(size_t)p * 64 + 32 + r] = c1[i];
                if (bins) {
                    s0 += c0[i]; q0 += (double)c0[i] * c0[i];
                    s1 += c1[i]; q1 += (double)c1[i] * c1[i];
                }
            }
        }
    }
    if (!bins) return;   // uniform
    // lanes r and r + 32 hold the same channels; then the four waves through LDS; 128 fp64 atomics per block
    s0 += __shfl_xor(s0, 32, 64); q0 += __shfl_xor(q0, 32, 64);
    s1 += __shfl_xor(s1, 32, 64); q1 += __shfl_xor(q1, 32, 64);
    if (h == 0) {
        redsm[(wave * 2 + 0) * 64 + r] = s0; redsm[(wave * 2 + 0) * 64 + 32 + r] = s1;
        redsm[(wave * 2 + 1) * 64 + r] = q0; redsm[(wave * 2 + 1) * 64 + 32 + r] = q1;
    }
    __syncthreads();
    if (threadIdx.x < 128) {
        const int qq = threadIdx.x >> 6, col = threadIdx.x & 63;
        const double t = (redsm[(0 * 2 + qq) * 64 + col] + redsm[(1 * 2 + qq) * 64 + col]) + (redsm[(2 * 2 + qq) * 64 + col] + redsm[(3 * 2 + qq) * 64 + col]);
        unsafeAtomicAdd(bins + ((size_t)(blockIdx.x % (unsigned)nbin) * 2 + qq) * 64 + col, t);
    }
}

// ---------------------------------------------------------------------------------------------------------
// Input gradient of final_layer.3 = Conv2d(64 -> 3, k3 s1 p1) (loss.backward(), celeba_deepvanilla_vae.py:249) on the matrix cores,
// over the tap-folded output gradient that the layer's weight gradient needs anyway (fold_taps, shifted form: fold[p][c * 9 + tap] =
// draw[p + 1 - tap][c]):   dact[p][ci] = sum_k fold[p][k] * w[k][ci],  w = the layer's weight [3][9][64] read as [27][64].
// Same wave-level shape as folded_fwd_kernel (32 pixels x 64 channels x K = 32: four 32x32x16 MFMAs), but MFMA column r of the two
// n-tiles stands for the channels 2r and 2r + 1, so that a lane holds two neighbouring channels of a pixel and a store instruction
// writes whole pixel rows (256 B fp32 / 128 B bf16 per 32 lanes).  RED: the two BatchNorm-backward sums of the layer in front
// (final_layer.0) per channel, sum dd and sum dd * xhat with dd = dact * lrelu'(act), from the fp32 accumulators (igemm_kc red_mode 2),
// fp64 atomics into bins[nbin][2][64].  blockIdx.y = BatchNorm group (P pixels each, own bins / mean / invstd).
// The VALU form (conv_thin_in: 27 * 64 FMAs per pixel in one thread, 203 registers, two waves per SIMD) took 71 us per 64 images.
// MODE 0: dact only; 1: dact and the two sums (RED); 2: the two sums only, dact is not stored; 3: dact is formed again and the BatchNorm
// backward applied to it at once: out = draw (bf16) = gamma invstd (dd - s1 / R - xhat s2 / R) with s1, s2 the totals of `bins`, dgamma / dbeta
// added by the first block of a group, the column sums of draw (the bias gradient in front of the BatchNorm: rounding noise) per block into
// dbias_part [gridDim.x][64] (may be NULL).  MODE 2 + MODE 3 replace MODE 1 + the BatchNorm-backward apply pass: the gradient tensor
// (16.8 M elements per 64 images) is neither written nor read back, the folded output gradient (half its size) is read twice.  In MODE 3
// OUT_BF says that the gradient would have been stored as bf16: it is rounded so before use, as the two-pass form reads it.
template <bool OUT_BF, bool RAW_BF, int MODE>
__global__ __launch_bounds__(256) void folded_dgrad_kernel(const bf16_t* __restrict__ fold, const bf16_t* __restrict__ w, int K,
                                                           void* __restrict__ out_, int P, double* __restrict__ bins, int nbin,
                                                           const bf16_t* __restrict__ act, const void* __restrict__ raw_,
                                                           const float* __restrict__ mean, const float* __restrict__ invstd,
                                                           const float* __restrict__ ab, const FlagCarry carry,
                                                           const float* __restrict__ gamma = nullptr, float* __restrict__ dgamma = nullptr,
                                                           float* __restrict__ dbeta = nullptr, double* __restrict__ dbias_part = nullptr,
                                                           float invR = 0.f) {
    constexpr bool RED = MODE == 1 || MODE == 2, APPLY = MODE == 3, STORE = MODE <= 1, NEED = MODE != 0;
    carry_store(carry);
    const size_t goff = (size_t)blockIdx.y * P;
    fold += goff * 32;
    __shared__ __attribute__((aligned(16))) bf16_t wl[64 * 40];   // [channel][32] + 8 columns of padding
    __shared__ double redsm[4 * 2 * 64];
    for (int i = threadIdx.x; i < 64 * 32; i += 256) {
        const int k = i >> 6, o = i & 63;                          // (consecutive threads read consecutive channels of row k)
        wl[o * 40 + k] = k < K ? w[k * 64 + o] : (bf16_t)0;
    }
    if (APPLY) {   // totals of the group's bins, in bin order (as gbn_bwd_kernel adds them)
        if (threadIdx.x < 128) {
            const int qq = threadIdx.x >> 6, col = threadIdx.x & 63;
            double t = 0.0;
            for (int b = 0; b < nbin; ++b) t += bins[(((size_t)blockIdx.y * nbin + b) * 2 + qq) * 64 + col];
            redsm[qq * 64 + col] = t;
            if (blockIdx.x == 0) atomicAdd((qq ? dgamma : dbeta) + col, (float)t);
        }
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, h = lane >> 5;
    union U { uint4 u; bf16x8 b; };
    U b00, b01, b10, b11;   // [n tile][k step]; n tile 0 = even channels, 1 = odd channels
    b00.u = *(const uint4*)(wl + (2 * r) * 40 + h * 8);
    b01.u = *(const uint4*)(wl + (2 * r) * 40 + 16 + h * 8);
    b10.u = *(const uint4*)(wl + (2 * r + 1) * 40 + h * 8);
    b11.u = *(const uint4*)(wl + (2 * r + 1) * 40 + 16 + h * 8);
    float mu0 = 0.f, mu1 = 0.f, is0 = 0.f, is1 = 0.f;
    if (NEED) {
        mu0 = mean[blockIdx.y * 64 + 2 * r]; mu1 = mean[blockIdx.y * 64 + 2 * r + 1];
        is0 = invstd[blockIdx.y * 64 + 2 * r]; is1 = invstd[blockIdx.y * 64 + 2 * r + 1];
    }
    // ab != NULL: the slope of the LeakyReLU from the forward pass's affine form raw * a + b instead of the stored activation
    float fa0 = 0.f, fa1 = 0.f, fb0 = 0.f, fb1 = 0.f;
    if (NEED && ab) {
        fa0 = ab[(blockIdx.y * 2) * 64 + 2 * r]; fa1 = ab[(blockIdx.y * 2) * 64 + 2 * r + 1];
        fb0 = ab[(blockIdx.y * 2 + 1) * 64 + 2 * r]; fb1 = ab[(blockIdx.y * 2 + 1) * 64 + 2 * r + 1];
    }
    [[maybe_unused]] float gi0 = 0.f, gi1 = 0.f, k10 = 0.f, k11 = 0.f, k20 = 0.f, k21 = 0.f;
    if (APPLY) {
        gi0 = gamma[2 * r] * is0; gi1 = gamma[2 * r + 1] * is1;
        k10 = (float)redsm[2 * r] * invR; k11 = (float)redsm[2 * r + 1] * invR;
        k20 = (float)redsm[64 + 2 * r] * invR; k21 = (float)redsm[64 + 2 * r + 1] * invR;
        __syncthreads();   // (redsm is used again below)
    }
    double s0 = 0.0, q0 = 0.0, s1 = 0.0, q1 = 0.0;
    const int ngroup = (P + 31) / 32, nw = gridDim.x * 4;
    for (int g = blockIdx.x * 4 + wave; g < ngroup; g += nw) {
        const int p0 = g * 32, pa = min(p0 + r, P - 1);
        U a0, a1;
        a0.u = *(const uint4*)(fold + (size_t)pa * 32 + h * 8);
        a1.u = *(const uint4*)(fold + (size_t)pa * 32 + 16 + h * 8);
        // act / raw of the 16 pixel rows this lane finishes, fetched ahead of the MFMAs
        [[maybe_unused]] unsigned pa2[16];
        [[maybe_unused]] float2 px[16];
        if (NEED) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int p = min(p0 + (i & 3) + 8 * (i >> 2) + 4 * h, P - 1);
                const size_t e = (goff + p) * 64 + 2 * r;
                pa2[i] = ab ? 0u : *(const unsigned*)(act + e);
                if (RAW_BF) {
                    const unsigned u = *(const unsigned*)((const bf16_t*)raw_ + e);
                    px[i] = make_float2(__uint_as_float(u << 16), __uint_as_float(u & 0xffff0000u));
                } else {
                    px[i] = *(const float2*)((const float*)raw_ + e);
                }
            }
        }
        f32x16 c0, c1;
#pragma unroll
        for (int i = 0; i < 16; ++i) { c0[i] = 0.f; c1[i] = 0.f; }
        c0 = mfma_bf16_32x32x16(a0.b, b00.b, c0);
        c0 = mfma_bf16_32x32x16(a1.b, b01.b, c0);
        c1 = mfma_bf16_32x32x16(a0.b, b10.b, c1);
        c1 = mfma_bf16_32x32x16(a1.b, b11.b, c1);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int p = p0 + (i & 3) + 8 * (i >> 2) + 4 * h;
            if (p < P) {
                const size_t e = (goff + p) * 64 + 2 * r;
                if (STORE) {
                    if (OUT_BF) *(unsigned*)((bf16_t*)out_ + e) = pack_bf16x2(c0[i], c1[i]);
                    else *(float2*)((float*)out_ + e) = make_float2(c0[i], c1[i]);
                }
                if (NEED) {
                    float a_e = __uint_as_float(pa2[i] << 16), a_o = __uint_as_float(pa2[i] & 0xffff0000u);
                    if (ab) { a_e = px[i].x * fa0 + fb0; a_o = px[i].y * fa1 + fb1; }   // uniform
                    if (RED) {
                        const float d0 = c0[i] * (a_e > 0.f ? 1.f : 0.01f), d1 = c1[i] * (a_o > 0.f ? 1.f : 0.01f);
                        s0 += d0; s1 += d1;
                        q0 += (double)d0 * ((px[i].x - mu0) * is0); q1 += (double)d1 * ((px[i].y - mu1) * is1);
                    } else {   // APPLY
                        float g0 = c0[i], g1 = c1[i];
                        if (OUT_BF) { const unsigned u = pack_bf16x2(g0, g1); g0 = __uint_as_float(u << 16); g1 = __uint_as_float(u & 0xffff0000u); }
                        const float d0 = g0 * (a_e > 0.f ? 1.f : 0.01f), d1 = g1 * (a_o > 0.f ? 1.f : 0.01f);
                        const float xh0 = (px[i].x - mu0) * is0, xh1 = (px[i].y - mu1) * is1;
                        const float o0 = gi0 * (d0 - k10 - xh0 * k20), o1 = gi1 * (d1 - k11 - xh1 * k21);
                        *(unsigned*)((bf16_t*)out_ + e) = pack_bf16x2(o0, o1);
                        s0 += o0; s1 += o1;
                    }
                }
            }
        }
    }
    if (!NEED) return;
    if (APPLY && !dbias_part) return;   // uniform
    s0 += __shfl_xor(s0, 32, 64); q0 += __shfl_xor(q0, 32, 64);
    s1 += __shfl_xor(s1, 32, 64); q1 += __shfl_xor(q1, 32, 64);
    if (h == 0) {
        redsm[(wave * 2 + 0) * 64 + 2 * r] = s0; redsm[(wave * 2 + 0) * 64 + 2 * r + 1] = s1;
        redsm[(wave * 2 + 1) * 64 + 2 * r] = q0; redsm[(wave * 2 + 1) * 64 + 2 * r + 1] = q1;
    }
    __syncthreads();
    if (threadIdx.x < 128) {
        const int qq = threadIdx.x >> 6, col = threadIdx.x & 63;
        const double t = (redsm[(0 * 2 + qq) * 64 + col] + redsm[(1 * 2 + qq) * 64 + col]) + (redsm[(2 * 2 + qq) * 64 + col] + redsm[(3 * 2 + qq) * 64 + col]);
        if (APPLY) { if (qq == 0) dbias_part[(size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 64 + col] = t; }
        else unsafeAtomicAdd(bins + (((size_t)blockIdx.y * nbin + blockIdx.x % (unsigned)nbin) * 2 + qq) * 64 + col, t);
    }
}
}  // namespace

int folded_dgrad_launch(const void* fold, const void* w, int K, void* out, long long P, int groups, int out_bf16, double* bins, int nbin,
                        const void* act, const void* raw, int raw_bf16, const float* mean, const float* invstd, const float* ab, hipStream_t st) {
    const FlagCarry carry = favae_take_flag_carry();
    FAVAE_CHECK_ARG(groups >= 1 && P % groups == 0 && (groups == 1 || (P / groups) % 32 == 0), "folded input gradient: %lld pixels in %d groups", P, groups);
    const int Pg = (int)(P / groups);
    const int ngroup = (Pg + 31) / 32;
    const int grid = std::max(1, std::min((ngroup + 3) / 4, std::max(128, 1024 / groups)));
#define FD_LAUNCH(OB, RB, MD) hipLaunchKernelGGL((folded_dgrad_kernel<OB, RB, MD>), dim3(grid, groups), dim3(256), 0, st, (const bf16_t*)fold, (const bf16_t*)w, K, \
                                                 out, Pg, bins, std::max(1, nbin), (const bf16_t*)act, raw, mean, invstd, ab, carry)
    if (!bins) { if (out_bf16) FD_LAUNCH(true, false, 0); else FD_LAUNCH(false, false, 0); }
    else if (!out) { if (raw_bf16) FD_LAUNCH(false, true, 2); else FD_LAUNCH(false, false, 2); }   // the sums only
    else if (out_bf16) { if (raw_bf16) FD_LAUNCH(true, true, 1); else FD_LAUNCH(true, false, 1); }
    else { if (raw_bf16) FD_LAUNCH(false, true, 1); else FD_LAUNCH(false, false, 1); }
#undef FD_LAUNCH
    FAVAE_LAUNCH_CHECK();
    return FAVAE_OK;
}

// MODE 3 of the kernel (see there).  grad_bf16: the gradient is rounded to bf16 before use, as its stored form would be.  slabs_cap: blocks per
// group at most when dbias_part is given ([groups * blocks][64] doubles); *slabs_out = groups * blocks.
int folded_dgrad_bn_launch(const void* fold, const void* w, int K, void* draw, long long P, int groups, int grad_bf16, const double* bins, int nbin,
                           const void* act, const void* raw, int raw_bf16, const float* mean, const float* invstd, const float* gamma, const float* ab,
                           float* dgamma, float* dbeta, double* dbias_part, int slabs_cap, int* slabs_out, hipStream_t st) {
    const FlagCarry carry = favae_take_flag_carry();
    FAVAE_CHECK_ARG(groups >= 1 && P % groups == 0 && (groups == 1 || (P / groups) % 32 == 0), "folded input gradient: %lld pixels in %d groups", P, groups);
    const int Pg = (int)(P / groups);
    const int ngroup = (Pg + 31) / 32;
    int grid = std::max(1, std::min((ngroup + 3) / 4, std::max(128, 1024 / groups)));
    if (dbias_part) {
        FAVAE_CHECK_ARG(slabs_cap >= groups, "folded input gradient: room for %d slabs, %d groups", slabs_cap, groups);
        grid = std::max(1, std::min(grid, slabs_cap / groups));
    }
    if (slabs_out) *slabs_out = grid * groups;
    const float invR = (float)(1.0 / (double)Pg);
#define FD_LAUNCH(OB, RB) hipLaunchKernelGGL((folded_dgrad_kernel<OB, RB, 3>), dim3(grid, groups), dim3(256), 0, st, (const bf16_t*)fold, (const bf16_t*)w, K, \
                                             draw, Pg, (double*)bins, std::max(1, nbin), (const bf16_t*)act, raw, mean, invstd, ab, carry, gamma, dgamma, dbeta, \
                                             dbias_part, invR)
    if (grad_bf16) { if (raw_bf16) FD_LAUNCH(true, true); else FD_LAUNCH(true, false); }
    else { if (raw_bf16) FD_LAUNCH(false, true); else FD_LAUNCH(false, false); }
#undef FD_LAUNCH
    FAVAE_LAUNCH_CHECK();
    return FAVAE_OK;
}

namespace {
}  // namespace

int folded_fwd_launch(const void* fold, const void* w, int K, const float* bias, float* out, long long P, double* bins, int nbin,
                      int split, hipStream_t st, int groups) {
    const FlagCarry carry = favae_take_flag_carry();
    FAVAE_CHECK_ARG(groups >= 1 && P % groups == 0 && (groups == 1 || (P / groups) % 32 == 0), "folded forward: %lld pixels in %d groups", P, groups);
    const int Pg = (int)(P / groups);
    const int ngroup = (Pg + 31) / 32;
    const int grid = std::max(1, std::min(std::max(64, 512 / groups), (ngroup + 3) / 4));
    if (split)
        hipLaunchKernelGGL(folded_fwd_kernel<true>, dim3(grid, groups), dim3(256), 0, st, (const bf16_t*)fold, w, K, bias, out, Pg, bins,
                           std::max(1, nbin), carry, (int)P);
    else
        hipLaunchKernelGGL(folded_fwd_kernel<false>, dim3(grid, groups), dim3(256), 0, st, (const bf16_t*)fold, w, K, bias, out, Pg, bins,
                           std::max(1, nbin), carry, (int)P);
    FAVAE_LAUNCH_CHECK();
    return FAVAE_OK;
}

// ---------------------------------------------------------------------------------------------------------
// Tap folding for the weight gradients with an image-sized channel count on one side (encoder.0, final_layer.3):
// the 9 taps x <= 3 channels of the narrow operand become the 27 (padded to 32) columns of a plain [pixels][32] matrix,
// after which the weight gradient is ONE pixel-major GEMM over the pixels (igemm.hip) that reads the wide operand
// exactly once, instead of 9 shifted passes.  One thread per pixel, 64 or 128 B written per thread.
namespace {
template <class T, int CS>
__global__ __launch_bounds__(256) void fold_taps(const T* __restrict__ src, T* __restrict__ dst, int Hq, int Wq, int Hn, int Wn,
                                                 int stride, int shifted, int total) {
    const int pix = blockIdx.x * 256 + threadIdx.x;
    T v[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) v[i] = (T)0;
    if (pix < total) {
        const int n = pix / (Hq * Wq), rem = pix - n * Hq * Wq, y = rem / Wq, x = rem - y * Wq;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int sy = shifted ? y + 1 - kh : y * stride - 1 + kh, sx = shifted ? x + 1 - kw : x * stride - 1 + kw;
                if (sy < 0 || sy >= Hn || sx < 0 || sx >= Wn) continue;
                const T* s = src + ((size_t)(n * Hn + sy) * Wn + sx) * CS;
#pragma unroll
                for (int c = 0; c < CS; ++c) {
                    if (shifted) v[c * 9 + kh * 3 + kw] = s[c];
                    else v[(kh * 3 + kw) * CS + c] = s[c];
                }
            }
    }
    const uint4* vv = (const uint4*)v;
    if constexpr (sizeof(T) == 2) {
        // the 64-byte rows leave through LDS so that every store instruction of a wave writes 1 KB of consecutive bytes (a thread
        // storing its own row in 16-byte pieces writes every 128-byte line several times; see fold_taps_hl)
        __shared__ __attribute__((aligned(16))) T sh[256 * 32];
        const int t = threadIdx.x;
#pragma unroll
        for (int i = 0; i < 4; ++i) *(uint4*)(sh + t * 32 + ((i + (t >> 2)) & 3) * 8) = vv[i];
        __syncthreads();
        const int p0 = blockIdx.x * 256;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int q = i * 256 + t, row = q >> 2, piece = q & 3;
            if (p0 + row < total) *(uint4*)(dst + (size_t)(p0 + row) * 32 + piece * 8) = *(const uint4*)(sh + row * 32 + ((piece + (row >> 2)) & 3) * 8);
        }
    } else {
        if (pix >= total) return;
        uint4* out = (uint4*)(dst + (size_t)pix * 32);
#pragma unroll
        for (int i = 0; i < (int)(32 * sizeof(T) / 16); ++i) out[i] = vv[i];
    }
}
}  // namespace

// The same gather (shifted = 0) from an fp32 source into TWO bf16 planes of [pixels][32]: hi = bf16(x) at dst, lo = bf16(x - hi) at
// dst + pixels * 32.  The weight gradient reads the hi plane (what fold_taps<bf16_t> writes); the SPLIT forward reads both.
namespace {
template <int CS>
__global__ __launch_bounds__(256) void fold_taps_hl(const float* __restrict__ src, bf16_t* __restrict__ dst, int Hq, int Wq, int Hn, int Wn,
                                                    int stride, int total) {
    // a thread gathers the 27 values of its pixel; the rows leave through LDS so that every store instruction of a wave writes 1 KB of
    // consecutive bytes (a thread storing its own 64-byte row in 16-byte pieces writes every line four times: 10.1 -> 4 us)
    __shared__ __attribute__((aligned(16))) bf16_t sh[2][256 * 32];
    const int pix = blockIdx.x * 256 + threadIdx.x;
    bf16_t hi[32], lo[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) hi[i] = lo[i] = (bf16_t)0;
    if (pix < total) {
        const int n = pix / (Hq * Wq), rem = pix - n * Hq * Wq, y = rem / Wq, x = rem - y * Wq;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int sy = y * stride - 1 + kh, sx = x * stride - 1 + kw;
                if (sy < 0 || sy >= Hn || sx < 0 || sx >= Wn) continue;
                const float* s = src + ((size_t)(n * Hn + sy) * Wn + sx) * CS;
#pragma unroll
                for (int c = 0; c < CS; ++c) {
                    const float f = s[c];
                    const bf16_t h = f32_to_bf16(f);
                    hi[(kh * 3 + kw) * CS + c] = h;
                    lo[(kh * 3 + kw) * CS + c] = f32_to_bf16(f - bf16_to_f32(h));
                }
            }
    }
    // the 16-byte pieces of an LDS row are rotated by (row / 4) so that the 64 lanes of a wave, each writing piece i of its own
    // 64-byte row, spread evenly over the 16 bank groups (4 lanes each: the minimum)
    const int t = threadIdx.x;
    const uint4 *vh = (const uint4*)hi, *vl = (const uint4*)lo;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        *(uint4*)(sh[0] + t * 32 + ((i + (t >> 2)) & 3) * 8) = vh[i];
        *(uint4*)(sh[1] + t * 32 + ((i + (t >> 2)) & 3) * 8) = vl[i];
    }
    __syncthreads();
    const int p0 = blockIdx.x * 256;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int v = i * 256 + t, row = v >> 2, piece = v & 3;   // 16-byte piece v of the block's 16 KB plane slice
        if (p0 + row < total) {
            *(uint4*)(dst + (size_t)(p0 + row) * 32 + piece * 8) = *(const uint4*)(sh[0] + row * 32 + ((piece + (row >> 2)) & 3) * 8);
            *(uint4*)(dst + ((size_t)total + p0 + row) * 32 + piece * 8) = *(const uint4*)(sh[1] + row * 32 + ((piece + (row >> 2)) & 3) * 8);
        }
    }
}
}  // namespace

int fold_taps_hl_launch(const float* src, void* dst, int Nb, int Hq, int Wq, int Hn, int Wn, int Cs, int stride, hipStream_t st) {
    FAVAE_CHECK_ARG(Cs >= 1 && Cs <= 3, "fold_taps_hl: 1..3 channels");
    const int total = Nb * Hq * Wq;
#define FH_CASE(CS_) hipLaunchKernelGGL((fold_taps_hl<CS_>), dim3(cdiv(total, 256)), dim3(256), 0, st, src, (bf16_t*)dst, Hq, Wq, Hn, Wn, stride, total)
    switch (Cs) { case 1: FH_CASE(1); break; case 2: FH_CASE(2); break; default: FH_CASE(3); }
#undef FH_CASE
    FAVAE_LAUNCH_CHECK();
    return FAVAE_OK;
}

int fold_taps_launch(const void* src, void* dst, int Nb, int Hq, int Wq, int Hn, int Wn, int Cs, int stride, int shifted,
                     int dtype, hipStream_t st) {
    FAVAE_CHECK_ARG(Cs >= 1 && Cs <= 3, "fold_taps: 1..3 channels");
    const int total = Nb * Hq * Wq;
#define FT_CASE(T_, CS_) hipLaunchKernelGGL((fold_taps<T_, CS_>), dim3(cdiv(total, 256)), dim3(256), 0, st, (const T_*)src, (T_*)dst, Hq, Wq, Hn, Wn, stride, shifted, total)
    if (dtype == FAVAE_F32) {
        switch (Cs) { case 1: FT_CASE(float, 1); break; case 2: FT_CASE(float, 2); break; default: FT_CASE(float, 3); }
    } else {
        switch (Cs) { case 1: FT_CASE(bf16_t, 1); break; case 2: FT_CASE(bf16_t, 2); break; default: FT_CASE(bf16_t, 3); }
    }
#undef FT_CASE
    FAVAE_LAUNCH_CHECK();
    return FAVAE_OK;
}
