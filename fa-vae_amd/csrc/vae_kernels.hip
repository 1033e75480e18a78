// Non-GEMM kernels of the conv-VAE training step on gfx950: BatchNorm(train/eval)+LeakyReLU forward and
// backward, reparameterisation + KL against the FA prior, sigmoid + Gaussian log-likelihood, Adam, weight
// re-layout, batch gather.  All of them are HBM-bound column reductions or element-wise passes over NHWC
// [rows][C] matrices.  Reference: condition_a_VAE/celeba_deepvanilla_vae.py (line numbers cited per kernel).
#include <algorithm>
#include <atomic>
#include <type_traits>
#include <cstdlib>
#include "common.h"
#include "favae_hip.h"

namespace {

// ---------------------------------------------------------------------------------------------------------
// deterministic column reduction over a row-major [R][C] matrix, one launch:
//   every block (slab of rows, tile of channels) writes fp64 partial sums partial[slab][v][c]; the block that takes
//   the last ticket of its channel tile adds the slabs in a fixed order and runs the per-channel finish functor.
//   The result does not depend on which block happens to be last, so it is bit-reproducible.
// Threads hold VW consecutive channels (16-byte loads for VW = 4).
// ---------------------------------------------------------------------------------------------------------
constexpr int TICKET_SLOTS = 4096;
__device__ unsigned g_tickets[TICKET_SLOTS];   // zero at load; every launch leaves the slots it used at zero
// host: rotating slot base, so launches on different streams do not share.  Slots are touched only while a launch EXECUTES (a queued
// launch holds nothing) and a launch takes at most 256 of the 3 840 positions, so two launches could only meet on a slot if 15 or
// more ticketed launches were executing at the same time; a process has 4 hardware queues.
std::atomic<unsigned> g_ticket_next{0};

int ticket_base(int nslots, unsigned** out) {
    static unsigned* per_device[64] = {nullptr};
    int dev = 0;
    FAVAE_HIP(hipGetDevice(&dev));
    FAVAE_CHECK_ARG(dev >= 0 && dev < 64, "column reduction: device index %d", dev);
    if (!per_device[dev]) FAVAE_HIP(hipGetSymbolAddress((void**)&per_device[dev], HIP_SYMBOL(g_tickets)));
    const unsigned base = g_ticket_next.fetch_add((unsigned)nslots) % (unsigned)(TICKET_SLOTS - 256);
    *out = per_device[dev] + base;
    return FAVAE_OK;
}

struct RedGeom { int VW, TPR, RG, ctiles, nslab, rps; };

constexpr int RED_THREADS = 512;    // 8 waves: few, fat blocks keep the slab count (and the last block's tail) small
constexpr int RED_WAVES = RED_THREADS / 64;

RedGeom red_geom(long long R, int C) {
    RedGeom g;
    g.VW = (C % 4 == 0) ? 4 : 1;
    const int cv = C / g.VW, cap = g.VW == 4 ? 16 : 64;
    g.TPR = 1;
    while (g.TPR < cv && g.TPR < cap) g.TPR *= 2;
    g.RG = RED_THREADS / g.TPR;
    g.ctiles = cdiv(cv, g.TPR);
    static const int red_cap = getenv("FAVAE_RED_CAP") ? atoi(getenv("FAVAE_RED_CAP")) : 256;   // (128 until the BN-backward apply lost its cross-block tail: more slabs then meant a longer tail; now 192 - 256 is 0.5 % faster)
    static const int red_cap_big = getenv("FAVAE_RED_CAP_BIG") ? atoi(getenv("FAVAE_RED_CAP_BIG")) : 256;
    static const int red_max = getenv("FAVAE_RED_MAX") ? atoi(getenv("FAVAE_RED_MAX")) : 512;
    const int target = std::max(1, std::min(R >= 200000 ? red_cap_big : red_cap, red_max / g.ctiles));
    const long long min_rows = (long long)g.RG * 4;
    long long ns = std::max(1LL, std::min((long long)target, cdivll(R, min_rows)));
    if (R <= (long long)g.RG * 4) ns = 1;    // a single slab needs no cross-block hand-over at all
    g.rps = (int)cdivll(R, ns);
    g.nslab = (int)cdivll(R, g.rps);
    return g;
}

template <int NV, int VW, class F>
__global__ __launch_bounds__(RED_THREADS) void colreduce(int R, int C, int TPR, int rps, int nslab, double* partial,
                                                         unsigned* tickets, F f) {
    __shared__ double sm[NV * RED_THREADS];
    __shared__ double btot[F::BLOCK_INIT ? 128 : 1];   // per-tile totals a functor may prepare for its init()
    __shared__ int is_last;
    const int tid = threadIdx.x, tl = tid % TPR, rg = tid / TPR, RG = RED_THREADS / TPR;
    if constexpr (F::BLOCK_INIT) f.block_init(sm, btot, TPR * VW);
    const int lane = tid & 63, wave = tid >> 6;
    const int c0 = (blockIdx.y * TPR + tl) * VW;
    const bool live = c0 < C;
    double acc[NV][VW];
#pragma unroll
    for (int v = 0; v < NV; ++v)
#pragma unroll
        for (int j = 0; j < VW; ++j) acc[v][j] = 0.0;
    if (live) {
        typename F::Ctx ctx;
        f.init(ctx, c0, btot + (F::BLOCK_INIT ? tl * VW * 2 : 0));
        const int r0 = blockIdx.x * rps, r1 = min(R, r0 + rps);
#pragma unroll 4
        for (int r = r0 + rg; r < r1; r += RG) f.row(ctx, r, c0, acc);
    }
    // row groups of one wave (lanes tl, tl+TPR, ...) by a fixed xor tree, then the waves in order
#pragma unroll
    for (int v = 0; v < NV; ++v)
#pragma unroll
        for (int j = 0; j < VW; ++j) {
            double t = acc[v][j];
            for (int o = TPR; o < 64; o <<= 1) t += __shfl_xor(t, o, 64);
            if (lane < TPR) sm[((v * VW + j) * RED_WAVES + wave) * TPR + lane] = t;
        }
    __syncthreads();
    if constexpr (F::DEFER) {   // the slab's sums, plain stores; whoever needs the totals adds the slabs (slab_sums_finish)
        if (tid < TPR && live) {
#pragma unroll
            for (int v = 0; v < NV; ++v)
#pragma unroll
                for (int j = 0; j < VW; ++j) {
                    double t = 0.0;
#pragma unroll
                    for (int w = 0; w < RED_WAVES; ++w) t += sm[((v * VW + j) * RED_WAVES + w) * TPR + tid];
                    if (c0 + j < C) partial[((size_t)blockIdx.x * NV + v) * C + c0 + j] = t;
                }
        }
        return;
    }
    if (nslab == 1) {
        if (tid < TPR && live) {
#pragma unroll
            for (int j = 0; j < VW; ++j) {
                double tot[NV];
#pragma unroll
                for (int v = 0; v < NV; ++v) {
                    double t = 0.0;
#pragma unroll
                    for (int w = 0; w < RED_WAVES; ++w) t += sm[((v * VW + j) * RED_WAVES + w) * TPR + tid];
                    tot[v] = 0.0 + t;   // same association as the multi-slab tail with one slab
                }
                f.finish(c0 + j, tot);
            }
        }
        return;
    }
    if (tid < TPR && live) {
#pragma unroll
        for (int v = 0; v < NV; ++v)
#pragma unroll
            for (int j = 0; j < VW; ++j) {
                double t = 0.0;
#pragma unroll
                for (int w = 0; w < RED_WAVES; ++w) t += sm[((v * VW + j) * RED_WAVES + w) * TPR + tid];
                __hip_atomic_store(&partial[((size_t)blockIdx.x * NV + v) * C + c0 + j], t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
    }
    // The partial sums travel between blocks as agent-scope (write-through / cache-bypassing) atomic stores and loads, so
    // only their completion has to be awaited before the ticket is taken: no L2 write-back fence per block.  A
    // workgroup-scope release fence emits no wait on gfx9, and s_barrier does not wait for memory either: every thread
    // waits for its own stores explicitly, then the barrier, then the ticket.
    wait_vm_writes();
    __syncthreads();
    if (tid == 0) is_last = __hip_atomic_fetch_add(&tickets[blockIdx.y], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)(nslab - 1);
    __syncthreads();
    if (!is_last) return;
    // last block of this channel tile: CT channels x (RED_THREADS / CT) slab parts
    const int CT = TPR * VW, ch = tid % CT, part = tid / CT, parts = RED_THREADS / CT;
    const int c = blockIdx.y * CT + ch;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        double t = 0.0;
        for (int base = 0; base < nslab; base += 8 * parts) {   // batches of 8 independent loads, summed in slab order
            double x[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int i = base + part + q * parts;
                x[q] = (c < C && i < nslab) ? __hip_atomic_load(&partial[((size_t)i * NV + v) * C + c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) t += x[q];
        }
        sm[v * RED_THREADS + tid] = t;
    }
    __syncthreads();
    if (tid < CT && c < C) {
        double tot[NV];
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            double t = 0.0;
            for (int q = 0; q < parts; ++q) t += sm[v * RED_THREADS + q * CT + tid];
            tot[v] = t;
        }
        f.finish(c, tot);
    }
    if (tid == 0) __hip_atomic_store(&tickets[blockIdx.y], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// out[c] = sum over the slabs, in slab order (fp64, reproducible), of partial[slab][c]: 64 channels x 4 slab parts per block
__global__ __launch_bounds__(256) void slab_sums_finish(const double* __restrict__ partial, int nslab, int C, float* __restrict__ out) {
    __shared__ double sm[4][64];
    const int cl = threadIdx.x & 63, part = threadIdx.x >> 6, c = blockIdx.x * 64 + cl;
    // contiguous quarter of the slabs per part, 8 independent loads in flight
    const int per = (nslab + 3) / 4, k0 = part * per, k1 = min(nslab, k0 + per);
    double t = 0.0;
    if (c < C) {
        for (int k = k0; k < k1; k += 8) {
            double x[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) x[q] = k + q < k1 ? partial[(size_t)(k + q) * C + c] : 0.0;
#pragma unroll
            for (int q = 0; q < 8; ++q) t += x[q];
        }
    }
    sm[part][cl] = t;
    __syncthreads();
    if (part == 0 && c < C) out[c] = (float)(((sm[0][cl] + sm[1][cl]) + sm[2][cl]) + sm[3][cl]);
}

template <int NV, class F>
int run_colreduce(long long R, int C, double* partial, size_t partial_cap, F f, hipStream_t st) {
    const RedGeom g = red_geom(R, C);
    FAVAE_CHECK_ARG((size_t)g.nslab * NV * C <= partial_cap, "column reduction: scratch too small (%zu < %zu doubles)",
                    partial_cap, (size_t)g.nslab * NV * C);
    FAVAE_CHECK_ARG(g.ctiles <= 256, "column reduction: too many channels (%d)", C);
    unsigned* tickets = nullptr;
    int trc = ticket_base(g.ctiles, &tickets);
    if (trc) return trc;
    if (g.VW == 4)
        hipLaunchKernelGGL((colreduce<NV, 4, F>), dim3(g.nslab, g.ctiles), dim3(RED_THREADS), 0, st, (int)R, C, g.TPR, g.rps, g.nslab, partial, tickets, f);
    else
        hipLaunchKernelGGL((colreduce<NV, 1, F>), dim3(g.nslab, g.ctiles), dim3(RED_THREADS), 0, st, (int)R, C, g.TPR, g.rps, g.nslab, partial, tickets, f);
    FAVAE_LAUNCH_CHECK();
    return FAVAE_OK;
}

template <class T, int VW> struct VecLoad;
template <> struct VecLoad<float, 4> {
    static __device__ __forceinline__ void ld(const float* p, float* o) { const float4 v = *(const float4*)p; o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w; }
    static __device__ __forceinline__ void st(float* p, const float* o) { *(float4*)p = make_float4(o[0], o[1], o[2], o[3]); }
};
template <> struct VecLoad<bf16_t, 4> {
    static __device__ __forceinline__ void ld(const bf16_t* p, float* o) {
        const uint2 u = *(const uint2*)p;
        o[0] = __uint_as_float(u.x << 16); o[1] = __uint_as_float(u.x & 0xffff0000u);
        o[2] = __uint_as_float(u.y << 16); o[3] = __uint_as_float(u.y & 0xffff0000u);
    }
    static __device__ __forceinline__ void st(bf16_t* p, const float* o) {
        uint2 u;
        u.x = pack_bf16x2(o[0], o[1]);
        u.y = pack_bf16x2(o[2], o[3]);
        *(uint2*)p = u;
    }
};
template <class T> struct VecLoad<T, 1> {
    static __device__ __forceinline__ void ld(const T* p, float* o) { o[0] = to_f32<T>(*p); }
    static __device__ __forceinline__ void st(T* p, const float* o) { *p = from_f32<T>(o[0]); }
};

// ---- BatchNorm2d (train) statistics: nn.BatchNorm2d(h_dim), celeba_deepvanilla_vae.py:34,58,72 -------------
template <class RT>
struct StatsF {
    const RT* x; int C; double R; float eps, mom; float *mean, *invstd, *run_mean, *run_var;
    static constexpr bool BLOCK_INIT = false;
    static constexpr bool DEFER = false;
    struct Ctx {};
    __device__ void init(Ctx&, int, const double*) const {}
    template <int VW> __device__ void row(const Ctx&, int r, int c0, double (*acc)[VW]) const {
        float v[VW];
        VecLoad<RT, VW>::ld(x + (size_t)r * C + c0, v);
#pragma unroll
        for (int j = 0; j < VW; ++j) { acc[0][j] += v[j]; acc[1][j] += (double)v[j] * v[j]; }
    }
    __device__ void finish(int c, const double* tot) const {
        const double m = tot[0] / R;
        double var = tot[1] / R - m * m;
        if (var < 0) var = 0;
        mean[c] = (float)m;
        invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
        if (run_mean) {
            const double unb = R > 1 ? var * R / (R - 1) : var;
            run_mean[c] = (1.f - mom) * run_mean[c] + mom * (float)m;
            run_var[c] = (1.f - mom) * run_var[c] + mom * (float)unb;
        }
    }
};

__global__ void bn_eval_prep(const float* run_mean, const float* run_var, int C, float eps, float* mean, float* invstd) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    mean[c] = run_mean[c];
    invstd[c] = 1.f / sqrtf(run_var[c] + eps);
}

// the affine form of eval-mode BN with the same operations as bn_eval_prep + bn_lrelu_fwd (invstd = 1 / sqrt(var + eps), a = invstd * gamma)
__global__ void bn_eval_ab(const float* run_mean, const float* run_var, const float* gamma, const float* beta, int C, float eps, float* ab) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float is = 1.f / sqrtf(run_var[c] + eps);
    const float a = is * gamma[c];
    ab[c] = a;
    ab[C + c] = beta[c] - run_mean[c] * a;
}

// ---- BN apply + LeakyReLU(0.01) forward, 4 channels per thread -----------------------------------------------
template <class T> __device__ __forceinline__ void store4(T* p, float a, float b, float c, float d);
template <> __device__ __forceinline__ void store4<float>(float* p, float a, float b, float c, float d) {
    *(float4*)p = make_float4(a, b, c, d);
}
template <> __device__ __forceinline__ void store4<bf16_t>(bf16_t* p, float a, float b, float c, float d) {
    uint2 u;
    u.x = pack_bf16x2(a, b);
    u.y = pack_bf16x2(c, d);
    *(uint2*)p = u;
}
template <class T> __device__ __forceinline__ float4 load4(const T* p);
template <> __device__ __forceinline__ float4 load4<float>(const float* p) { return *(const float4*)p; }
template <> __device__ __forceinline__ float4 load4<bf16_t>(const bf16_t* p) {
    const uint2 u = *(const uint2*)p;
    return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16),
                       __uint_as_float(u.y & 0xffff0000u));
}

__device__ __forceinline__ float lrelu(float y) { return y > 0.f ? y : 0.01f * y; }

// The grid stride is a multiple of C (host picks it), so a thread keeps its 4 channels for the whole loop and the
// affine form y = x * a + b is set up once.
template <class T, class RT>
__global__ __launch_bounds__(256) void bn_lrelu_fwd(const RT* __restrict__ raw, const float* __restrict__ mean,
                                                    const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                    const float* __restrict__ beta, T* __restrict__ act, size_t nvec, int C,
                                                    int apply_act) {
    const size_t i0 = (size_t)blockIdx.x * 256 + threadIdx.x, stride = (size_t)gridDim.x * 256;
    const int c = (int)((i0 * 4) % C);
    float a[4], b[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        a[j] = invstd[c + j] * gamma[c + j];
        b[j] = beta[c + j] - mean[c + j] * a[j];
    }
#pragma unroll 4
    for (size_t i = i0; i < nvec; i += stride) {
        float x[4];
        VecLoad<RT, 4>::ld(raw + i * 4, x);
        float y[4] = {x[0] * a[0] + b[0], x[1] * a[1] + b[1], x[2] * a[2] + b[2], x[3] * a[3] + b[3]};
        if (apply_act) {
#pragma unroll
            for (int j = 0; j < 4; ++j) y[j] = lrelu(y[j]);
        }
        store4<T>(act + i * 4, y[0], y[1], y[2], y[3]);
    }
}

// The same with the affine form given (ab = a[C] | b[C], published by the fused-input GEMM, igemm.hip): explicit fmaf so that
// the stored activation equals what that GEMM formed on the fly, bit for bit.
template <class T>
__global__ __launch_bounds__(256) void bn_lrelu_fwd_ab(const float* __restrict__ raw, const float* __restrict__ ab, T* __restrict__ act,
                                                       size_t nvec, int C) {
    const size_t i0 = (size_t)blockIdx.x * 256 + threadIdx.x, stride = (size_t)gridDim.x * 256;
    const int c = (int)((i0 * 4) % C);
    const float4 a = *(const float4*)(ab + c), b = *(const float4*)(ab + C + c);
#pragma unroll 4
    for (size_t i = i0; i < nvec; i += stride) {
        const float4 x = *(const float4*)(raw + i * 4);
        const float y0 = fmaf(x.x, a.x, b.x), y1 = fmaf(x.y, a.y, b.y), y2 = fmaf(x.z, a.z, b.z), y3 = fmaf(x.w, a.w, b.w);
        store4<T>(act + i * 4, lrelu(y0), lrelu(y1), lrelu(y2), lrelu(y3));
    }
}

// Per-block prologue of the kernels that consume epilogue sums (igemm.hip, KcParams::red_bins): add the nbin bins of the
// 2*C per-channel sums; tot[q*C + c] is valid after the call.  Needs 2*C <= 512 and 2*C | 256 or 256 | 2*C.
__device__ __forceinline__ void bins_total(const double* __restrict__ bins, int nbin, int C, double* sm /*[256]*/, double* tot /*[512]*/) {
    const int tid = threadIdx.x, C2 = 2 * C;
    if (C2 <= 256) {
        const int qc = tid % C2, part = tid / C2, parts = 256 / C2;
        double t = 0.0;
#pragma unroll 8
        for (int b = part; b < nbin; b += parts) t += bins[(size_t)b * C2 + qc];
        sm[tid] = t;
        __syncthreads();
        if (tid < C2) {
            double u = 0.0;
            for (int k = 0; k < parts; ++k) u += sm[k * C2 + tid];
            tot[tid] = u;
        }
    } else {
        for (int qc = tid; qc < C2; qc += 256) {
            double t = 0.0;
#pragma unroll 8
            for (int b = 0; b < nbin; ++b) t += bins[(size_t)b * C2 + qc];
            tot[qc] = t;
        }
    }
    __syncthreads();
}
inline bool bins_applicable(int C) { return C % 4 == 0 && C <= 256 && (2 * C <= 256 ? 256 % (2 * C) == 0 : (2 * C) % 256 == 0); }

// BN(train) + LeakyReLU forward whose statistics were accumulated by the producing GEMM's epilogue (sum, sum of squares
// per channel in `bins`): every block derives mean / invstd itself, block 0 also publishes them and the running stats.
template <class T, class RT>
__global__ __launch_bounds__(256) void bn_lrelu_fwd_bins(const RT* __restrict__ raw, const double* __restrict__ bins, int nbin,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         T* __restrict__ act, float* __restrict__ mean, float* __restrict__ invstd,
                                                         float* __restrict__ run_mean, float* __restrict__ run_var, size_t nvec,
                                                         int C, double R, float eps, float mom, float* __restrict__ ab_out = nullptr) {
    __shared__ double sm[256], tot[512];
    __shared__ float ab[512];
    bins_total(bins, nbin, C, sm, tot);
    const int tid = threadIdx.x;
    if (tid < C) {
        const double m = tot[tid] / R;
        double var = tot[C + tid] / R - m * m;
        if (var < 0) var = 0;
        const float is = (float)(1.0 / sqrt(var + (double)eps));
        const float a = is * gamma[tid];
        ab[tid] = a;
        ab[C + tid] = beta[tid] - (float)m * a;
        if (blockIdx.x == 0) {
            mean[tid] = (float)m;
            invstd[tid] = is;
            if (ab_out) { ab_out[tid] = a; ab_out[C + tid] = ab[C + tid]; }   // the (a | b) the backward pass takes the LeakyReLU slope from
            if (run_mean) {
                const double unb = R > 1 ? var * R / (R - 1) : var;
                run_mean[tid] = (1.f - mom) * run_mean[tid] + mom * (float)m;
                run_var[tid] = (1.f - mom) * run_var[tid] + mom * (float)unb;
            }
        }
    }
    __syncthreads();
    const size_t i0 = (size_t)blockIdx.x * 256 + tid, stride = (size_t)gridDim.x * 256;
    const int c = (int)((i0 * 4) % C);
    float a[4], b[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { a[j] = ab[c + j]; b[j] = ab[C + c + j]; }
#pragma unroll 4
    for (size_t i = i0; i < nvec; i += stride) {
        float x[4];
        VecLoad<RT, 4>::ld(raw + i * 4, x);
        store4<T>(act + i * 4, lrelu(x[0] * a[0] + b[0]), lrelu(x[1] * a[1] + b[1]), lrelu(x[2] * a[2] + b[2]), lrelu(x[3] * a[3] + b[3]));
    }
}

// ---- BN + LeakyReLU backward ---------------------------------------------------------------------------------
template <class T, class RT, class DT>
struct BnBwdRedF {
    const DT* dact; const T* act; const RT* raw; const float* mean; const float* invstd; int C;
    float *dgamma, *dbeta, *s1s2;
    static constexpr bool BLOCK_INIT = false;
    static constexpr bool DEFER = false;
    struct Ctx { float mu[4], is[4]; };
    __device__ void init(Ctx& k, int c0, const double*) const {
        for (int j = 0; j < 4 && c0 + j < C; ++j) { k.mu[j] = mean[c0 + j]; k.is[j] = invstd[c0 + j]; }
    }
    template <int VW> __device__ void row(const Ctx& k, int r, int c0, double (*acc)[VW]) const {
        const size_t i = (size_t)r * C + c0;
        float a[VW], d[VW], x[VW];
        VecLoad<T, VW>::ld(act + i, a);
        VecLoad<DT, VW>::ld(dact + i, d);
        VecLoad<RT, VW>::ld(raw + i, x);
#pragma unroll
        for (int j = 0; j < VW; ++j) {
            const float dd = d[j] * (a[j] > 0.f ? 1.f : 0.01f);
            const float xh = (x[j] - k.mu[j]) * k.is[j];
            acc[0][j] += dd; acc[1][j] += (double)dd * xh;
        }
    }
    __device__ void finish(int c, const double* tot) const {
        dbeta[c] = (float)tot[0];
        dgamma[c] = (float)tot[1];
        s1s2[c] = (float)tot[0];
        s1s2[C + c] = (float)tot[1];
    }
};

// DEFERRED (FAVAE_DBIAS_DEFER): the per-slab column sums stay in the caller's buffer and favae_bn_dbias_finish adds them up
// later, on any stream: the launch ends with its last row instead of waiting for its stores, taking a ticket and leaving
// one block to read every partial (2 % of the training step over its seven launches)
template <class T, class RT, class DT, bool DEFERRED = false>
struct BnBwdApplyF {
    static constexpr bool DEFER = DEFERRED;
    const DT* dact; const T* act; const RT* raw; const float* mean; const float* invstd; const float* gamma;
    const float* s1s2; T* draw; int C; float invR; float* dbias;
    // bins != NULL: the two sums come from the epilogue of the GEMM that produced dact (igemm.hip, red_mode 2) instead of
    // BnBwdRedF; every block adds the bins of its channel tile, the first slab also writes dgamma / dbeta
    const double* bins; int nbin; float *dgamma, *dbeta;
    // ab != NULL: the (a | b) of the forward pass (favae_bn_train_fwd_bins_ab): the LeakyReLU slope is the sign of raw * a + b, the
    // forward pass's own expression, and the stored activation is not read (2 of the 12 bytes per element)
    const float* ab = nullptr;
    static constexpr bool BLOCK_INIT = true;
    struct Ctx { float mu[4], is[4], gi[4], k1[4], k2[4], fa[4], fb[4]; };
    __device__ void block_init(double* sm, double* btot, int CT) const {
        if (!bins) return;   // uniform
        const int tid = threadIdx.x, n2 = 2 * CT, v = tid % n2, part = tid / n2, parts = RED_THREADS / n2;
        const int q = v / CT, c = blockIdx.y * CT + v % CT;
        double t = 0.0;
        if (c < C) {
#pragma unroll 8
            for (int b = part; b < nbin; b += parts) t += bins[((size_t)b * 2 + q) * C + c];
        }
        sm[tid] = t;
        __syncthreads();
        if (tid < n2) {
            double u = 0.0;
            for (int k = 0; k < parts; ++k) u += sm[k * n2 + tid];
            btot[(tid % CT) * 2 + q] = u;   // q = tid / CT for tid < n2
            if (blockIdx.x == 0 && c < C) (q ? dgamma : dbeta)[c] = (float)u;
        }
        __syncthreads();
    }
    __device__ void init(Ctx& k, int c0, const double* bt) const {
        for (int j = 0; j < 4 && c0 + j < C; ++j) {
            k.mu[j] = mean[c0 + j]; k.is[j] = invstd[c0 + j]; k.gi[j] = gamma[c0 + j] * invstd[c0 + j];
            if (ab) { k.fa[j] = ab[c0 + j]; k.fb[j] = ab[C + c0 + j]; }
            if (bins) { k.k1[j] = (float)bt[2 * j] * invR; k.k2[j] = (float)bt[2 * j + 1] * invR; }
            else { k.k1[j] = s1s2[c0 + j] * invR; k.k2[j] = s1s2[C + c0 + j] * invR; }
        }
    }
    template <int VW> __device__ void row(const Ctx& k, int r, int c0, double (*acc)[VW]) const {
        const size_t i = (size_t)r * C + c0;
        float a[VW], d[VW], x[VW], g[VW];
        VecLoad<DT, VW>::ld(dact + i, d);
        VecLoad<RT, VW>::ld(raw + i, x);
        if (ab) {   // uniform
#pragma unroll
            for (int j = 0; j < VW; ++j) a[j] = x[j] * k.fa[j] + k.fb[j];
        } else {
            VecLoad<T, VW>::ld(act + i, a);
        }
#pragma unroll
        for (int j = 0; j < VW; ++j) {
            const float dd = d[j] * (a[j] > 0.f ? 1.f : 0.01f);
            const float xh = (x[j] - k.mu[j]) * k.is[j];
            g[j] = k.gi[j] * (dd - k.k1[j] - xh * k.k2[j]);
            acc[0][j] += g[j];
        }
        VecLoad<T, VW>::st(draw + i, g);
    }
    __device__ void finish(int c, const double* tot) const { if (dbias) dbias[c] = (float)tot[0]; }
};

// ---------------------------------------------------------------------------------------------------------
// "Channel-owner" BatchNorm kernels for the small layers (R <= 1024 rows: the 4x4 and 2x2 feature maps at batch 64).
// One 512-thread block owns OWN_CT (8) channels and ALL rows, keeps its elements in registers between the statistics and the
// apply phase, and needs no hand-over between blocks: statistics + apply (forward) or both reduction passes + apply
// (backward) are one launch instead of two, which is what counts for tensors of 0.25-0.5 M elements.
// Sums are fp64 in a fixed order (bit-reproducible), as in colreduce.
// ---------------------------------------------------------------------------------------------------------
// channels per block: 8 measured best on MI355X (64 blocks for the 512-channel layers; 16 -> 8: -1.0 % step time, 4: -0.5 %)
#ifndef FAVAE_OWN_CT
#define FAVAE_OWN_CT 8
#endif
constexpr int OWN_THREADS = 512, OWN_CT = FAVAE_OWN_CT, OWN_TPR = OWN_CT / 4 /* threads per row */, OWN_RG = OWN_THREADS / OWN_TPR,
              OWN_MAXP = 1024 / OWN_RG, OWN_WAVES = OWN_THREADS / 64;

// sum the per-thread fp64 partials val[NQ][4] over the row groups of the block; result for channel (tl*4+j) of
// quantity q lands in tot[q][OWN_CT] (shared), valid after the trailing barrier
template <int NQ>
__device__ __forceinline__ void own_reduce(double (*val)[4], double* sm /* [NQ][OWN_WAVES][OWN_CT] */, double* tot /* [NQ][OWN_CT] */) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, tl = tid % OWN_TPR;
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            double t = val[q][j];
            for (int o = OWN_TPR; o < 64; o <<= 1) t += __shfl_xor(t, o, 64);
            if (lane < OWN_TPR) sm[(q * OWN_WAVES + wave) * OWN_CT + tl * 4 + j] = t;
        }
    __syncthreads();
    if (tid < NQ * OWN_CT) {
        const int q = tid / OWN_CT, c = tid % OWN_CT;
        double t = 0.0;
#pragma unroll
        for (int w = 0; w < OWN_WAVES; ++w) t += sm[(q * OWN_WAVES + w) * OWN_CT + c];
        tot[q * OWN_CT + c] = t;
    }
    __syncthreads();
}

template <class T, int MAXP = OWN_MAXP>
__global__ __launch_bounds__(OWN_THREADS) void bn_fwd_owner(const float* __restrict__ raw, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, T* __restrict__ act,
                                                            float* __restrict__ mean, float* __restrict__ invstd,
                                                            float* __restrict__ run_mean, float* __restrict__ run_var, int R,
                                                            int C, float eps, float mom, float* __restrict__ act32 = nullptr) {
    __shared__ double sm[2 * OWN_WAVES * OWN_CT], tot[2 * OWN_CT];
    __shared__ float ab[2 * OWN_CT];
    const int tid = threadIdx.x, tl = tid % OWN_TPR, rg = tid / OWN_TPR;
    const int c0 = blockIdx.x * OWN_CT + tl * 4;
    float x[MAXP][4];
    double val[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
#pragma unroll
    for (int p = 0; p < MAXP; ++p) {
        const int r = rg + p * OWN_RG;
        if (r < R) {
            VecLoad<float, 4>::ld(raw + (size_t)r * C + c0, x[p]);
#pragma unroll
            for (int j = 0; j < 4; ++j) { val[0][j] += x[p][j]; val[1][j] += (double)x[p][j] * x[p][j]; }
        }
    }
    own_reduce<2>(val, sm, tot);
    if (tid < OWN_CT) {
        const int c = blockIdx.x * OWN_CT + tid;
        const double Rd = (double)R, m = tot[tid] / Rd;
        double var = tot[OWN_CT + tid] / Rd - m * m;
        if (var < 0) var = 0;
        const float is = (float)(1.0 / sqrt(var + (double)eps));
        mean[c] = (float)m;
        invstd[c] = is;
        if (run_mean) {
            const double unb = R > 1 ? var * Rd / (Rd - 1) : var;
            run_mean[c] = (1.f - mom) * run_mean[c] + mom * (float)m;
            run_var[c] = (1.f - mom) * run_var[c] + mom * (float)unb;
        }
        const float a = is * gamma[c];
        ab[tid] = a;
        ab[OWN_CT + tid] = beta[c] - (float)m * a;
    }
    __syncthreads();
    float a[4], b[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { a[j] = ab[tl * 4 + j]; b[j] = ab[OWN_CT + tl * 4 + j]; }
#pragma unroll
    for (int p = 0; p < MAXP; ++p) {
        const int r = rg + p * OWN_RG;
        if (r < R) {
            float y[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) y[j] = lrelu(x[p][j] * a[j] + b[j]);
            VecLoad<T, 4>::st(act + (size_t)r * C + c0, y);
            if (act32) VecLoad<float, 4>::st(act32 + (size_t)r * C + c0, y);   // uniform
        }
    }
}

template <class T>
__global__ __launch_bounds__(OWN_THREADS) void bn_bwd_owner(const float* __restrict__ dact, const T* __restrict__ act,
                                                            const float* __restrict__ raw, const float* __restrict__ mean,
                                                            const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                            T* __restrict__ draw, float* __restrict__ dgamma,
                                                            float* __restrict__ dbeta, float* __restrict__ dbias, int R, int C) {
    __shared__ double sm[2 * OWN_WAVES * OWN_CT], tot[2 * OWN_CT];
    const int tid = threadIdx.x, tl = tid % OWN_TPR, rg = tid / OWN_TPR;
    const int c0 = blockIdx.x * OWN_CT + tl * 4;
    float mu[4], is[4], gi[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { mu[j] = mean[c0 + j]; is[j] = invstd[c0 + j]; gi[j] = gamma[c0 + j] * is[j]; }
    float d[OWN_MAXP][4], xh[OWN_MAXP][4];
    double val[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
#pragma unroll
    for (int p = 0; p < OWN_MAXP; ++p) {
        const int r = rg + p * OWN_RG;
        if (r < R) {
            const size_t i = (size_t)r * C + c0;
            float a[4], x[4];
            VecLoad<T, 4>::ld(act + i, a);
            VecLoad<float, 4>::ld(dact + i, d[p]);
            VecLoad<float, 4>::ld(raw + i, x);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                d[p][j] *= (a[j] > 0.f ? 1.f : 0.01f);
                xh[p][j] = (x[j] - mu[j]) * is[j];
                val[0][j] += d[p][j]; val[1][j] += (double)d[p][j] * xh[p][j];
            }
        }
    }
    own_reduce<2>(val, sm, tot);
    if (tid < OWN_CT) {
        const int c = blockIdx.x * OWN_CT + tid;
        dbeta[c] = (float)tot[tid];
        dgamma[c] = (float)tot[OWN_CT + tid];
    }
    const float invR = (float)(1.0 / (double)R);
    float k1[4], k2[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { k1[j] = (float)tot[tl * 4 + j] * invR; k2[j] = (float)tot[OWN_CT + tl * 4 + j] * invR; }
    double gs[1][4] = {{0, 0, 0, 0}};
#pragma unroll
    for (int p = 0; p < OWN_MAXP; ++p) {
        const int r = rg + p * OWN_RG;
        if (r < R) {
            float g[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) { g[j] = gi[j] * (d[p][j] - k1[j] - xh[p][j] * k2[j]); gs[0][j] += g[j]; }
            VecLoad<T, 4>::st(draw + (size_t)r * C + c0, g);
        }
    }
    if (dbias) {   // uniform
        __syncthreads();   // tot is re-used
        own_reduce<1>(gs, sm, tot);
        if (tid < OWN_CT) dbias[blockIdx.x * OWN_CT + tid] = (float)tot[tid];
    }
}

inline bool owner_applicable(long long R, int C) { return R <= (long long)OWN_RG * OWN_MAXP && C % OWN_CT == 0; }
// forward only: up to 4096 rows (dec.1 at batch 64) with 16 row groups per thread: one launch instead of statistics + apply
constexpr int OWN_MAXP_BIG = 4 * OWN_MAXP;
inline bool owner_fwd_big_applicable(long long R, int C) { return R <= (long long)OWN_RG * OWN_MAXP_BIG && C % OWN_CT == 0; }

template <class T>
struct ColsumF {
    const T* x; int C; float* out;
    static constexpr bool BLOCK_INIT = false;
    static constexpr bool DEFER = false;
    struct Ctx {};
    __device__ void init(Ctx&, int, const double*) const {}
    template <int VW> __device__ void row(const Ctx&, int r, int c0, double (*acc)[VW]) const {
        float v[VW];
        VecLoad<T, VW>::ld(x + (size_t)r * C + c0, v);
#pragma unroll
        for (int j = 0; j < VW; ++j) acc[0][j] += v[j];
    }
    __device__ void finish(int c, const double* tot) const { out[c] = (float)tot[0]; }
};

// column sums of an fp32 matrix that is also stored as T on the way (gradient hand-over of a layer without BN:
// out[c] = bias gradient, y = the dtype copy the weight / input gradient GEMMs read)
template <class T>
struct ColsumCastF {
    const float* x; T* y; int C; float* out;
    static constexpr bool BLOCK_INIT = false;
    static constexpr bool DEFER = false;
    struct Ctx {};
    __device__ void init(Ctx&, int, const double*) const {}
    template <int VW> __device__ void row(const Ctx&, int r, int c0, double (*acc)[VW]) const {
        float v[VW];
        VecLoad<float, VW>::ld(x + (size_t)r * C + c0, v);
        VecLoad<T, VW>::st(y + (size_t)r * C + c0, v);
#pragma unroll
        for (int j = 0; j < VW; ++j) acc[0][j] += v[j];
    }
    __device__ void finish(int c, const double* tot) const { out[c] = (float)tot[0]; }
};

// ---- reparameterise + KL(q || N(mu_P, 1/tau)) : celeba_deepvanilla_vae.py:86-92,137-150,231-232 --------------
// one block of 1024 threads (B*z is 1280 elements at batch 64: the launch is a chain of dependent loads, idx -> prior row,
// and a double-precision log; more lanes shorten that chain); B*z is at most a few tens of thousands of elements
constexpr int RKF_THREADS = 1024;
template <class T>
__global__ __launch_bounds__(RKF_THREADS) void reparam_kl_fwd(const float* __restrict__ fc, const float* __restrict__ eps,
                                                      const double* __restrict__ prior, const int* __restrict__ idx,
                                                      float tau, int B, int z, T* __restrict__ zlat,
                                                      float* __restrict__ stdq, double* __restrict__ kl_out) {
    __shared__ double sm[RKF_THREADS / 64];
    const double std_p = sqrt((double)(1.0f / tau));
    const double sp = fmin(fmax(std_p, 1e-6), 1e20);
    double acc = 0.0;
    for (int i = threadIdx.x; i < B * z; i += RKF_THREADS) {
        const int b = i / z, k = i - b * z;
        const float mu = fc[b * 2 * z + k];
        const float lv = fc[b * 2 * z + z + k] + 1e-6f;
        const float sd = expf(0.5f * lv);
        stdq[i] = sd;
        zlat[i] = from_f32<T>(eps[i] * sd + mu);
        const double mp = prior ? prior[(size_t)(idx ? idx[b] : b) * z + k] : 0.0;
        const double sq = fmin(fmax((double)sd, 1e-6), 1e20);
        const double vr = (sq / sp) * (sq / sp);
        const double t1 = ((double)mu - mp) / sp;
        acc += 0.5 * (vr + t1 * t1 - 1.0 - log(vr));
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
#pragma unroll
        for (int w = 0; w < RKF_THREADS / 64; ++w) t += sm[w];
        *kl_out = t / B;
    }
}

// The same + decoder_input (:41-45, nn.Linear z -> N): every block recomputes z = eps * std + mu (B*z elements, rounded to
// the GEMM dtype as the stand-alone path stores it) into LDS and produces 64 of the N output columns for all B rows with
// K = z FMAs per output; block 0 also writes zlat / stdq and the KL term.  One launch instead of two on the main chain of
// the training step, both of which are chains of dependent loads rather than work (K = 20).
// Two instances: z <= 32 with RKF_THREADS threads (the headline latent sizes) and z <= 128 with 256 threads and the weight row
// of the thread's column still in registers (the 100-dimensional latent of the DA / TL trainers).
constexpr int RKD_ZMAX = 32, RKD_ZMAX_BIG = 128;
template <class T, int ZMAX, int THREADS>
__global__ __launch_bounds__(THREADS) void reparam_kl_decin(const float* __restrict__ fc, const float* __restrict__ eps,
                                                                 const double* __restrict__ prior, const int* __restrict__ idx,
                                                                 float tau, int B, int z, T* __restrict__ zlat,
                                                                 float* __restrict__ stdq, double* __restrict__ kl_out,
                                                                 const T* __restrict__ w, const float* __restrict__ bias,
                                                                 T* __restrict__ out, int N) {
    extern __shared__ float zs[];   // [B][z]
    __shared__ double sm[THREADS / 64];
    const bool first = blockIdx.x == 0;
    const int n = blockIdx.x * 64 + (threadIdx.x & 63), bg = threadIdx.x >> 6;
    // the weight row and the bias of this thread's column travel while z is computed
    float wr[ZMAX];
#pragma unroll
    for (int k = 0; k < ZMAX; ++k) wr[k] = (k < z && n < N) ? to_f32<T>(w[(size_t)n * z + k]) : 0.f;
    const float bn = n < N ? bias[n] : 0.f;
    const double std_p = sqrt((double)(1.0f / tau));
    const double sp = fmin(fmax(std_p, 1e-6), 1e20);
    double acc = 0.0;
    for (int i = threadIdx.x; i < B * z; i += THREADS) {
        const int b = i / z, k = i - b * z;
        const float mu = fc[b * 2 * z + k];
        const float lv = fc[b * 2 * z + z + k] + 1e-6f;
        const float sd = expf(0.5f * lv);
        const T zt = from_f32<T>(eps[i] * sd + mu);
        zs[i] = to_f32<T>(zt);
        if (first) {
            stdq[i] = sd;
            zlat[i] = zt;
            const double mp = prior ? prior[(size_t)(idx ? idx[b] : b) * z + k] : 0.0;
            const double sq = fmin(fmax((double)sd, 1e-6), 1e20);
            const double vr = (sq / sp) * (sq / sp);
            const double t1 = ((double)mu - mp) / sp;
            acc += 0.5 * (vr + t1 * t1 - 1.0 - log(vr));
        }
    }
    if (first) {
        acc = wave_sum(acc);
        if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = acc;
    }
    __syncthreads();
    if (first && threadIdx.x == 0) {
        double t = 0.0;
#pragma unroll
        for (int q = 0; q < THREADS / 64; ++q) t += sm[q];
        *kl_out = t / B;
    }
    if (n >= N) return;
    for (int b = bg; b < B; b += THREADS / 64) {
        float a = 0.f;
#pragma unroll
        for (int k = 0; k < ZMAX; ++k)
            if (k < z) a = fmaf(zs[b * z + k], wr[k], a);
        out[(size_t)b * N + n] = from_f32<T>(a + bn);
    }
}

template <class T>
__global__ __launch_bounds__(256) void reparam_kl_bwd(const float* __restrict__ fc, const float* __restrict__ eps,
                                                      const double* __restrict__ prior, const int* __restrict__ idx,
                                                      const float* __restrict__ stdq, const float* __restrict__ dz, float tau,
                                                      float beta, int B, int z, T* __restrict__ dfc) {
    const double std_p = sqrt((double)(1.0f / tau));
    const double sp = fmin(fmax(std_p, 1e-6), 1e20);
    const double w = (double)beta / B;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < B * z; i += gridDim.x * 256) {
        const int b = i / z, k = i - b * z;
        const float mu = fc[b * 2 * z + k];
        const float sd = stdq[i];
        const double mp = prior ? prior[(size_t)(idx ? idx[b] : b) * z + k] : 0.0;
        const bool in_rng = sd >= 1e-6f && sd <= 1e20f;
        const double sq = fmin(fmax((double)sd, 1e-6), 1e20);
        const float dkl_mu = (float)(w * ((double)mu - mp) / (sp * sp));
        const float dkl_sd = in_rng ? (float)(w * (sq / (sp * sp) - 1.0 / sq)) : 0.f;
        const float g = dz[i];
        const float dmu = g + dkl_mu;
        const float dsd = g * eps[i] + dkl_sd;
        dfc[b * 2 * z + k] = from_f32<T>(dmu);
        dfc[b * 2 * z + z + k] = from_f32<T>(dsd * 0.5f * sd);
    }
}

// The same gradient in one block that also sums its columns: dbias [2z] = bias gradient of the encoder head (fc_mu | fc_var
// rows of dfc).  Element-parallel first (values parked in LDS), then one thread per column adds its rows in order.
constexpr int RKB_THREADS = 1024, RKB_MAX = 8192;   // B * 2z <= RKB_MAX
template <class T, bool ATOMIC_DZ>
__device__ __forceinline__ void reparam_bwd_bias_body(const float* __restrict__ fc, const float* __restrict__ eps,
                                                      const double* __restrict__ prior, const int* __restrict__ idx,
                                                      const float* __restrict__ stdq, const float* dz, float tau, float beta, int B,
                                                      int z, T* __restrict__ dfc, float* __restrict__ dbias, float* vals /* [B*2z] */,
                                                      double* part /* [4*256] */) {
    const double std_p = sqrt((double)(1.0f / tau));
    const double sp = fmin(fmax(std_p, 1e-6), 1e20);
    const double w = (double)beta / B;
    const int z2 = 2 * z;
    for (int i = threadIdx.x; i < B * z; i += RKB_THREADS) {
        const int b = i / z, k = i - b * z;
        const float mu = fc[b * z2 + k];
        const float sd = stdq[i];
        const double mp = prior ? prior[(size_t)(idx ? idx[b] : b) * z + k] : 0.0;
        const bool in_rng = sd >= 1e-6f && sd <= 1e20f;
        const double sq = fmin(fmax((double)sd, 1e-6), 1e20);
        const float dkl_mu = (float)(w * ((double)mu - mp) / (sp * sp));
        const float dkl_sd = in_rng ? (float)(w * (sq / (sp * sp) - 1.0 / sq)) : 0.f;
        const float g = ATOMIC_DZ ? __hip_atomic_load(dz + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : dz[i];
        const float dmu = g + dkl_mu;
        const float dls = (g * eps[i] + dkl_sd) * 0.5f * sd;
        dfc[b * z2 + k] = from_f32<T>(dmu);
        dfc[b * z2 + z + k] = from_f32<T>(dls);
        vals[b * z2 + k] = dmu;
        vals[b * z2 + z + k] = dls;
    }
    __syncthreads();
    // 4 partial sums per column (rows b % 4 == part), combined in a fixed order
    const int c = threadIdx.x % 256, p4 = threadIdx.x / 256;
    if (c < z2) {
        double t = 0.0;
        for (int b = p4; b < B; b += 4) t += vals[b * z2 + c];
        part[p4 * 256 + c] = t;
    }
    __syncthreads();
    if ((int)threadIdx.x < z2) dbias[threadIdx.x] = (float)((part[threadIdx.x] + part[256 + threadIdx.x]) + (part[512 + threadIdx.x] + part[768 + threadIdx.x]));
}

template <class T>
__global__ __launch_bounds__(RKB_THREADS) void reparam_kl_bwd_bias(const float* __restrict__ fc, const float* __restrict__ eps,
                                                                   const double* __restrict__ prior, const int* __restrict__ idx,
                                                                   const float* __restrict__ stdq, const float* __restrict__ dz,
                                                                   float tau, float beta, int B, int z, T* __restrict__ dfc,
                                                                   float* __restrict__ dbias) {
    __shared__ float vals[RKB_MAX];
    __shared__ double part[4 * 256];
    reparam_bwd_bias_body<T, false>(fc, eps, prior, idx, stdq, dz, tau, beta, B, z, dfc, dbias, vals, part);
}

// LDS plan of decin_bwd_reparam<T, ZMAX>: phase 1 = ds [64][65] + ws [64][ZMAX+1] floats + colpart [16][64] doubles;
// phase 2 (last block) = vals [B*2z <= VALS] floats + part [4][256] doubles
template <int ZMAX> struct DecinBwdSmem {
    static constexpr int VALS = ZMAX <= 32 ? RKB_MAX : 2 * RKB_MAX;
    static constexpr int COLPART = ((64 * 65 + 64 * (ZMAX + 1)) * 4 + 15) / 16 * 16;
    static constexpr int P1 = COLPART + 16 * 64 * 8, P2 = VALS * 4 + 4 * 256 * 8;
    static constexpr int BYTES = P1 > P2 ? P1 : P2;
};

// Backward of decoder_input (:41-45) and of the reparameterisation in one launch (the middle of the backward pass is a chain
// of latency-bound launches).  Block j owns 64 of the N columns for all B rows: it stores the dtype copy of the incoming
// gradient (operand of the weight gradient), sums its columns (bias gradient, fixed order), and adds its share of
// dz = d @ W (fp32 atomics into the zeroed dz, as the split-K GEMM did); the block that takes the last ticket then has the
// complete dz and runs the reparameterisation / KL backward + the bias gradient of the encoder head.
template <class T, int ZMAX>
__global__ __launch_bounds__(RKB_THREADS) void decin_bwd_reparam(const float* __restrict__ dact, const T* __restrict__ w,
                                                                 T* __restrict__ draw, float* __restrict__ dbias_di, float* dz,
                                                                 unsigned* ticket, const float* __restrict__ fc,
                                                                 const float* __restrict__ eps, const double* __restrict__ prior,
                                                                 const int* __restrict__ idx, const float* __restrict__ stdq,
                                                                 float tau, float beta, int B, int z, int N, T* __restrict__ dfc,
                                                                 float* __restrict__ dbias_fc) {
    constexpr int ZP = ZMAX + 1;
    extern __shared__ __attribute__((aligned(16))) char smem[];   // DecinBwdSmem<ZMAX>::BYTES (40 KB / 72 KB): two phases share it
    __shared__ int is_last;
    float* ds = (float*)smem;                              // [64][65]
    float* ws = ds + 64 * 65;                              // [64][ZP]
    double* colpart = (double*)(smem + DecinBwdSmem<ZMAX>::COLPART);   // [16][64]
    const int tid = threadIdx.x, c = tid & 63, r = tid >> 6, n0 = blockIdx.x * 64;
    for (int i = tid; i < 64 * z; i += RKB_THREADS) {
        const int cc = i / z, k = i - cc * z;
        ws[cc * ZP + k] = to_f32<T>(w[(size_t)(n0 + cc) * z + k]);
    }
    double colacc = 0.0;
    for (int b0 = 0; b0 < B; b0 += 64) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int bl = r + 16 * j, b = b0 + bl;
            float dr = 0.f;
            if (b < B) {
                const float d = dact[(size_t)b * N + n0 + c];
                const T dt = from_f32<T>(d);
                draw[(size_t)b * N + n0 + c] = dt;
                dr = to_f32<T>(dt);
                colacc += (double)d;
            }
            ds[bl * 65 + c] = dr;
        }
        __syncthreads();
        for (int o = tid; o < 64 * z; o += RKB_THREADS) {
            const int bl = o / z, k = o - bl * z;
            if (b0 + bl < B) {
                float a = 0.f;
#pragma unroll 16
                for (int cc = 0; cc < 64; ++cc) a = fmaf(ds[bl * 65 + cc], ws[cc * ZP + k], a);
                atomicAdd(dz + (size_t)(b0 + bl) * z + k, a);
            }
        }
        __syncthreads();
    }
    colpart[r * 64 + c] = colacc;
    wait_vm_writes();   // this thread's dz atomics have been performed before the barrier that precedes the ticket
    __syncthreads();
    if (tid < 64) {
        double t = 0.0;
#pragma unroll
        for (int q = 0; q < 16; ++q) t += colpart[q * 64 + tid];
        dbias_di[n0 + tid] = (float)t;
    }
    if (tid == 0) is_last = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1;
    __syncthreads();
    if (!is_last) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // dz is read with plain loads below: drop any cached copy
    reparam_bwd_bias_body<T, true>(fc, eps, prior, idx, stdq, dz, tau, beta, B, z, dfc, dbias_fc, (float*)smem, (double*)(smem + DecinBwdSmem<ZMAX>::VALS * 4));
    if (tid == 0) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// eval-mode head of update_x: mu and var = std^2 as float64 rows of X.mean / X.cov (:331-358)
__global__ void encode_head(const float* __restrict__ fc, int B, int z, double* __restrict__ xmean, double* __restrict__ xcov) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B * z) return;
    const int b = i / z, k = i - b * z;
    const float sd = expf(0.5f * (fc[b * 2 * z + z + k] + 1e-6f));
    xmean[i] = (double)fc[b * 2 * z + k];
    xcov[i] = (double)(sd * sd);
}

// ---- sigmoid + Gaussian log-likelihood and its gradient (:77, 152-171) ---------------------------------------
template <class T>
__global__ __launch_bounds__(256) void sigmoid_ll(const float* __restrict__ raw, const float* __restrict__ x, size_t n,
                                                  float inv_var_B, T* __restrict__ draw, double* __restrict__ partial) {
    __shared__ double sm[4];
    double acc = 0.0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float s = 1.f / (1.f + expf(-raw[i]));
        const float d = s - x[i];
        acc += (double)d * d;
        draw[i] = from_f32<T>(d * inv_var_B * s * (1.f - s));
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = sm[0] + sm[1] + sm[2] + sm[3];
}

__global__ void ll_finish(const double* partial, int n, double D, double var_x, double B, double* rec_err) {
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += 64) s += partial[i];
    s = wave_sum(s);
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        // -mean_b LL_b ,  LL_b = -0.5 (D ln 2pi + D ln var + SSE_b / var)
        *rec_err = 0.5 * (D * log(2.0 * M_PI) + D * log(var_x)) + 0.5 * s / (var_x * B);
    }
}

// Same, one launch: the block that takes the last ticket adds the per-block partials in a fixed order and writes the
// reconstruction error; with dbias != NULL the column sums of the gradient (bias gradient of the output conv, C <= 4
// interleaved channels) come out of the same pass.  Replaces sigmoid_ll + ll_finish + a column reduction.
template <class T>
__global__ __launch_bounds__(256) void sigmoid_ll_bias(const float* __restrict__ raw, const float* __restrict__ x, size_t n, int C,
                                                       float inv_var_B, T* __restrict__ draw, double* __restrict__ partial,
                                                       unsigned* __restrict__ ticket, double D, double var_x, double B,
                                                       double* __restrict__ rec_err, float* __restrict__ dbias) {
    __shared__ double sm[5 * 4];
    __shared__ int is_last;
    double a[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
    const size_t stride = (size_t)gridDim.x * 256;
    const int cstep = (int)(stride % (size_t)C);
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    int c = (int)(i % (size_t)C);
    for (; i < n; i += stride) {
        const float s = 1.f / (1.f + expf(-raw[i]));
        const float d = s - x[i];
        a[0] += (double)d * d;
        const float g = d * inv_var_B * s * (1.f - s);
        draw[i] = from_f32<T>(g);
#pragma unroll
        for (int j = 0; j < 4; ++j) a[1 + j] += (c == j) ? g : 0.f;
        c += cstep;
        if (c >= C) c -= C;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        a[k] = wave_sum(a[k]);
        if (lane == 0) sm[k * 4 + wave] = a[k];
    }
    __syncthreads();
    if (threadIdx.x < 5)
        __hip_atomic_store(&partial[(size_t)blockIdx.x * 5 + threadIdx.x],
                           (sm[threadIdx.x * 4] + sm[threadIdx.x * 4 + 1]) + (sm[threadIdx.x * 4 + 2] + sm[threadIdx.x * 4 + 3]),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    wait_vm_writes();   // partial stores performed before the barrier that precedes the ticket
    __syncthreads();
    if (threadIdx.x == 0) is_last = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1;
    __syncthreads();
    if (!is_last) return;
    double t[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
    for (unsigned b = threadIdx.x; b < gridDim.x; b += 256) {
#pragma unroll
        for (int k = 0; k < 5; ++k) t[k] += __hip_atomic_load(&partial[(size_t)b * 5 + k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        t[k] = wave_sum(t[k]);
        if (lane == 0) sm[k * 4 + wave] = t[k];
    }
    __syncthreads();
    if (threadIdx.x < 5) {
        const double tot = (sm[threadIdx.x * 4] + sm[threadIdx.x * 4 + 1]) + (sm[threadIdx.x * 4 + 2] + sm[threadIdx.x * 4 + 3]);
        if (threadIdx.x == 0) *rec_err = 0.5 * (D * log(2.0 * M_PI) + D * log(var_x)) + 0.5 * tot / (var_x * B);
        else if (dbias && (int)threadIdx.x - 1 < C) dbias[threadIdx.x - 1] = (float)tot;
    }
    if (threadIdx.x == 0) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---- final_layer.3-4 + loss in one launch (:74-77, 152-171, 238) ------------------------------------------------
// Conv2d(64 -> CO <= 4, k3 s1 p1) + Sigmoid + Gaussian log-likelihood + its gradient + the conv's bias gradient.  The
// output is CO values per pixel, i.e. a 576-long dot product per pixel and channel: a lane loads the 64 input channels
// of its own pixel per tap (8 x 16 B, shared with the neighbouring pixels through L1/L2) and feeds them, as they are, to
// the 4x4x4 MFMA (16 independent blocks of 4 pixels x 4 output channels per wave; 4x4x1 for fp32), whose B operand (the
// weights of the tap, 4 KB in LDS) is the same for every block.  Nothing of the CO-channel result travels through memory
// before the loss.  wk: [9][16][4][4] dtype = (tap, k/4, cout, k%4), zero for cout >= CO (conv_out_wprep).
// Reduction of (SSE, gradient sums) as in sigmoid_ll_bias.
typedef short short4v __attribute__((ext_vector_type(4)));
typedef float float4v __attribute__((ext_vector_type(4)));

// Shared tail of the two conv_out_ll kernels: acc = D[i][j] of the lane's MFMA block (j = lane & 3 output channel, i = pixel
// 4 * (lane / 4) + i of the wave); sigmoid, log-likelihood, gradient, bias gradient and the ticketed reduction.
template <class T, int CO>
__device__ __forceinline__ void conv_out_ll_tail(const float4v acc, const float* __restrict__ bias, const float* __restrict__ x, int total,
                                                 float inv_var_B, float* __restrict__ raw, T* __restrict__ draw,
                                                 double* __restrict__ partial, unsigned* __restrict__ ticket, double D, double var_x,
                                                 double B, double* __restrict__ rec_err, float* __restrict__ dbias, double* sm,
                                                 int* is_last_p) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // this lane: output channel j of the 4 pixels of its block
    const int j = lane & 3, p0 = blockIdx.x * 256 + wave * 64 + (lane >> 2) * 4;
    if (!x) {   // uniform (favae_conv_out_raw, the eval decoder): the convolution output alone
        if (j < CO) {
            const float bj = bias ? bias[j] : 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (p0 + i < total) raw[(size_t)(p0 + i) * CO + j] = acc[i] + bj;
        }
        return;
    }
    double a[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
    if (j < CO) {
        const float bj = bias ? bias[j] : 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int p = p0 + i;
            if (p < total) {
                const size_t o = (size_t)p * CO + j;
                const float r = acc[i] + bj;
                const float s = 1.f / (1.f + expf(-r));
                const float d = s - x[o];
                const float g = d * inv_var_B * s * (1.f - s);
                raw[o] = r;
                draw[o] = from_f32<T>(g);
                a[0] += (double)d * d;
#pragma unroll
                for (int c = 0; c < CO; ++c) a[1 + c] += (j == c) ? g : 0.f;
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        a[k] = wave_sum(a[k]);
        if (lane == 0) sm[k * 4 + wave] = a[k];
    }
    __syncthreads();
    if (!ticket) {   // uniform, FAVAE_DBIAS_DEFER: the block's five sums stay in the caller's buffer, someone else adds the blocks
        if (threadIdx.x < 5)
            partial[(size_t)blockIdx.x * 5 + threadIdx.x] =
                (sm[threadIdx.x * 4] + sm[threadIdx.x * 4 + 1]) + (sm[threadIdx.x * 4 + 2] + sm[threadIdx.x * 4 + 3]);
        return;
    }
    if (threadIdx.x < 5)
        __hip_atomic_store(&partial[(size_t)blockIdx.x * 5 + threadIdx.x],
                           (sm[threadIdx.x * 4] + sm[threadIdx.x * 4 + 1]) + (sm[threadIdx.x * 4 + 2] + sm[threadIdx.x * 4 + 3]),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    wait_vm_writes();   // partial stores performed before the barrier that precedes the ticket
    __syncthreads();
    if (threadIdx.x == 0) *is_last_p = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1;
    __syncthreads();
    if (!*is_last_p) return;
    double t[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
    for (unsigned b0 = 0; b0 < gridDim.x; b0 += 1024) {   // 4 blocks' partials in flight per thread
        double q[4][5];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const unsigned b = b0 + threadIdx.x + 256 * u;
#pragma unroll
            for (int k = 0; k < 5; ++k)
                q[u][k] = b < gridDim.x ? __hip_atomic_load(&partial[(size_t)b * 5 + k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int k = 0; k < 5; ++k) t[k] += q[u][k];
    }
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        t[k] = wave_sum(t[k]);
        if (lane == 0) sm[k * 4 + wave] = t[k];
    }
    __syncthreads();
    if (threadIdx.x < 5) {
        const double tot = (sm[threadIdx.x * 4] + sm[threadIdx.x * 4 + 1]) + (sm[threadIdx.x * 4 + 2] + sm[threadIdx.x * 4 + 3]);
        if (threadIdx.x == 0) *rec_err = 0.5 * (D * log(2.0 * M_PI) + D * log(var_x)) + 0.5 * tot / (var_x * B);
        else if (dbias && (int)threadIdx.x - 1 < CO) dbias[threadIdx.x - 1] = (float)tot;
    }
    if (threadIdx.x == 0) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <class T, int CO>
__global__ __launch_bounds__(256) void conv_out_ll(const T* __restrict__ act, const T* __restrict__ wk, const float* __restrict__ bias,
                                                   const float* __restrict__ x, int H, int W, int total, float inv_var_B,
                                                   float* __restrict__ raw, T* __restrict__ draw, double* __restrict__ partial,
                                                   unsigned* __restrict__ ticket, double D, double var_x, double B,
                                                   double* __restrict__ rec_err, float* __restrict__ dbias) {
    constexpr int CI = 64, VE = 16 / sizeof(T), NV = CI / VE;
    __shared__ __attribute__((aligned(16))) T wl[9 * 16 * 4 * 4];
    __shared__ double sm[5 * 4];
    __shared__ int is_last;
    for (int i = threadIdx.x; i < (int)(9 * 16 * 16 * sizeof(T) / 16); i += 256) ((uint4*)wl)[i] = ((const uint4*)wk)[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int gid = blockIdx.x * 256 + threadIdx.x;
    const int pix = min(gid, total - 1);
    const int n = pix / (H * W), rem = pix - n * H * W, y = rem / W, xx = rem - y * W;
    // taps two at a time from two register buffers: the 8 loads of the next tap are in flight under the MFMAs of the
    // current one (a fully unrolled tap loop lets the compiler hoist all 72 loads and spill)
    auto load_tap = [&](int t, uint4 (&v)[NV]) -> bool {
        const int kh = t / 3, kw = t - kh * 3;
        const int iy = y - 1 + kh, ix = xx - 1 + kw;
        const int cy = min(max(iy, 0), H - 1), cx = min(max(ix, 0), W - 1);
        const uint4* src = (const uint4*)(act + ((size_t)(n * H + cy) * W + cx) * CI);
#pragma unroll
        for (int i = 0; i < NV; ++i) v[i] = src[i];
        return iy >= 0 && iy < H && ix >= 0 && ix < W;
    };
    float4v acc = {0.f, 0.f, 0.f, 0.f};   // D[i][j]: j = lane & 3 (output channel), i = register (pixel 4 * (lane / 4) + i of the wave)
    auto mma_tap = [&](int t, const uint4 (&v)[NV], bool ok) {
        const T* wt = wl + (size_t)t * 16 * 16 + (lane & 3) * 4;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const uint4 u = make_uint4(ok ? v[i].x : 0u, ok ? v[i].y : 0u, ok ? v[i].z : 0u, ok ? v[i].w : 0u);
            if constexpr (sizeof(T) == 2) {
                union { uint2 q; short4v s; } a0, a1, b0, b1;
                a0.q = make_uint2(u.x, u.y); a1.q = make_uint2(u.z, u.w);
                b0.q = *(const uint2*)(wt + (i * 2 + 0) * 16); b1.q = *(const uint2*)(wt + (i * 2 + 1) * 16);
                acc = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(a0.s, b0.s, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(a1.s, b1.s, acc, 0, 0, 0);
            } else {
                const float4 bw = *(const float4*)(wt + i * 16);
                acc = __builtin_amdgcn_mfma_f32_4x4x1f32(__uint_as_float(u.x), bw.x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_4x4x1f32(__uint_as_float(u.y), bw.y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_4x4x1f32(__uint_as_float(u.z), bw.z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_4x4x1f32(__uint_as_float(u.w), bw.w, acc, 0, 0, 0);
            }
        }
    };
    uint4 va[NV], vb[NV];
    bool ma = load_tap(0, va), mb = false;
#pragma unroll 1
    for (int t = 0; t < 8; t += 2) {
        mb = load_tap(t + 1, vb);
        mma_tap(t, va, ma);
        ma = load_tap(t + 2, va);
        mma_tap(t + 1, vb, mb);
    }
    mma_tap(8, va, ma);
    conv_out_ll_tail<T, CO>(acc, bias, x, total, inv_var_B, raw, draw, partial, ticket, D, var_x, B, rec_err, dbias, sm, &is_last);
}

// The same layer for bf16 images whose width is a multiple of 64: a wave owns 64 consecutive pixels of ONE image row, so
// the 66-pixel input segment of a kernel row is loaded once, coalesced (whole 128-B lines, 1 KB per instruction instead of
// 64 separate lines), parked in LDS, and read back per lane for the three horizontal taps.  Rows outside the image are
// skipped as a whole, columns outside are zero in LDS.
// AB: `act` holds the conv output in front of the layer's BatchNorm (bf16) and the activation lrelu(raw * a + b), rounded to bf16 as the
// BatchNorm kernels store it, is formed when a segment is loaded; in_ab [groups][2][64] = (a | b) per BatchNorm group of group_images
// images (0: one group).  The activation tensor is then never written: at 512 images that pass was 106 us and 536 MB per step.
template <int CO, bool AB = false>
__global__ __launch_bounds__(256) void conv_out_ll_rows(const bf16_t* __restrict__ act, const bf16_t* __restrict__ wk,
                                                        const float* __restrict__ bias, const float* __restrict__ x, int H, int W,
                                                        int total, float inv_var_B, float* __restrict__ raw, bf16_t* __restrict__ draw,
                                                        double* __restrict__ partial, unsigned* __restrict__ ticket, double D,
                                                        double var_x, double B, double* __restrict__ rec_err, float* __restrict__ dbias,
                                                        const float* __restrict__ in_ab = nullptr, int group_images = 0) {
    constexpr int SEG = 66, RV = 9;                    // pixels per segment, uint4 per LDS pixel row (8 data + 1 pad)
    constexpr int NQ = (SEG * 8 + 63) / 64;            // coalesced 16-B loads per lane and segment
    __shared__ __attribute__((aligned(16))) bf16_t wl[9 * 16 * 4 * 4];
    __shared__ uint4 rowbuf[4][SEG * RV];
    __shared__ double sm[5 * 4];
    __shared__ int is_last;
    for (int i = threadIdx.x; i < 9 * 16 * 16 * 2 / 16; i += 256) ((uint4*)wl)[i] = ((const uint4*)wk)[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int pw0 = min(blockIdx.x * 256 + wave * 64, total - 64);   // total % 64 == 0 (host); a surplus wave redoes the last segment
    const bool live = blockIdx.x * 256 + wave * 64 < total;
    const int n = pw0 / (H * W), rem = pw0 - n * H * W, y = rem / W, x0 = rem - y * W;
    uint4* rb = rowbuf[wave];
    [[maybe_unused]] float fa[8], fb[8];   // AB: (a, b) of this lane's 8 channels (part = lane & 7 for every load of the lane)
    if constexpr (AB) {
        const float* g_ab = in_ab + (size_t)(group_images > 0 ? n / group_images : 0) * 128 + (lane & 7) * 8;
#pragma unroll
        for (int j = 0; j < 8; ++j) { fa[j] = g_ab[j]; fb[j] = g_ab[64 + j]; }
    }
    auto load_seg = [&](int kh, uint4 (&v)[NQ]) -> bool {
        const int iy = y - 1 + kh;
        if (iy < 0 || iy >= H) return false;           // wave-uniform
        const uint4* src = (const uint4*)(act + ((size_t)(n * H + iy) * W) * 64);
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int e = lane + 64 * q, px = e >> 3, part = e & 7, gx = x0 - 1 + px;
            const bool ok = e < SEG * 8 && gx >= 0 && gx < W;
            uint4 t = src[(size_t)min(max(gx, 0), W - 1) * 8 + part];
            if constexpr (AB) {
                unsigned w4[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float y0 = lrelu(__uint_as_float(w4[k] << 16) * fa[2 * k] + fb[2 * k]);
                    const float y1 = lrelu(__uint_as_float(w4[k] & 0xffff0000u) * fa[2 * k + 1] + fb[2 * k + 1]);
                    w4[k] = pack_bf16x2(y0, y1);
                }
                t = make_uint4(w4[0], w4[1], w4[2], w4[3]);
            }
            v[q] = make_uint4(ok ? t.x : 0u, ok ? t.y : 0u, ok ? t.z : 0u, ok ? t.w : 0u);
        }
        return true;
    };
    float4v acc = {0.f, 0.f, 0.f, 0.f};
    uint4 va[NQ], vb[NQ];
    bool oka = load_seg(0, va), okb = false;
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
        uint4 (&cur)[NQ] = (kh & 1) ? vb : va;
        uint4 (&nxt)[NQ] = (kh & 1) ? va : vb;
        const bool okc = (kh & 1) ? okb : oka;
        bool okn = false;
        if (kh < 2) okn = load_seg(kh + 1, nxt);       // in flight under this row's LDS round trip and MFMAs
        if (kh & 1) oka = okn; else okb = okn;
        if (!okc) continue;                            // wave-uniform
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int e = lane + 64 * q, px = e >> 3, part = e & 7;
            if (e < SEG * 8) rb[px * RV + part] = cur[q];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
            const bf16_t* wt = wl + (size_t)(kh * 3 + kw) * 16 * 16 + (lane & 3) * 4;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const uint4 u = rb[(lane + kw) * RV + i];
                union { uint2 q; short4v s; } a0, a1, b0, b1;
                a0.q = make_uint2(u.x, u.y); a1.q = make_uint2(u.z, u.w);
                b0.q = *(const uint2*)(wt + (i * 2 + 0) * 16); b1.q = *(const uint2*)(wt + (i * 2 + 1) * 16);
                acc = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(a0.s, b0.s, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(a1.s, b1.s, acc, 0, 0, 0);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    if (!live) acc = float4v{0.f, 0.f, 0.f, 0.f};     // its pixels are >= total in the tail, nothing is stored or summed
    conv_out_ll_tail<bf16_t, CO>(acc, bias, x, total, inv_var_B, raw, draw, partial, ticket, D, var_x, B, rec_err, dbias, sm, &is_last);
}

// w [CO][9][64] (dtype) -> wk [9][16][4][4] (dtype): (tap, k / 4, cout (zero for cout >= CO), k % 4)
template <class T>
__global__ void conv_out_wprep(const T* __restrict__ w, T* __restrict__ wk, int CO) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= 9 * 16 * 16) return;
    const int kk = i & 3, co = (i >> 2) & 3, kg = (i >> 4) & 15, tap = i >> 8;
    wk[i] = co < CO ? w[((size_t)co * 9 + tap) * 64 + kg * 4 + kk] : (T)0;
}

// sigmoid(raw) vs y: per-block partial (sum (y-s)^2, sum y, sum y^2), then one block adds them onto acc3
template <class YT>
__global__ __launch_bounds__(256) void sigmoid_sse(const float* __restrict__ raw, const YT* __restrict__ y, size_t n,
                                                   double* __restrict__ partial) {
    __shared__ double sm[12];
    double a0 = 0.0, a1 = 0.0, a2 = 0.0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float s = 1.f / (1.f + expf(-raw[i]));
        double t;
        if constexpr (sizeof(YT) == 1) t = (double)((float)y[i] / 255.f); else t = (double)y[i];
        const double d = t - (double)s;
        a0 += d * d; a1 += t; a2 += t * t;
    }
    a0 = wave_sum(a0); a1 = wave_sum(a1); a2 = wave_sum(a2);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { sm[w] = a0; sm[4 + w] = a1; sm[8 + w] = a2; }
    __syncthreads();
    if (threadIdx.x < 3) partial[3 * blockIdx.x + threadIdx.x] = sm[4 * threadIdx.x] + sm[4 * threadIdx.x + 1] + sm[4 * threadIdx.x + 2] + sm[4 * threadIdx.x + 3];
}
__global__ void sse_finish(const double* partial, int n, double* acc3) {
    for (int v = 0; v < 3; ++v) {
        double s = 0.0;
        for (int i = threadIdx.x; i < n; i += 64) s += partial[3 * i + v];
        s = wave_sum(s);
        if (threadIdx.x == 0) acc3[v] += s;
    }
}

__global__ void sigmoid_only(const float* __restrict__ raw, float* __restrict__ out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        out[i] = 1.f / (1.f + expf(-raw[i]));
}

// ---- Adam (torch.optim.Adam defaults, :131,250) over the flat parameter arena ---------------------------------
template <bool CLEAR_G, bool COPY = false>
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, size_t n4, float b1, float b2, float eps,
                                                   float step_size, float bc2_sqrt, bf16_t* __restrict__ copy = nullptr,
                                                   size_t ncopy4 = 0, float gscale = 1.f, size_t keep4 = 0, size_t keepb_lo4 = 0,
                                                   size_t keepb_hi4 = 0) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        float4 P = ((float4*)p)[i], G = ((const float4*)g)[i], M = ((float4*)m)[i], V = ((float4*)v)[i];
        float* pp = (float*)&P; float* gg = (float*)&G; float* mm = (float*)&M; float* vv = (float*)&V;
#pragma unroll
        for (int j = 0; j < 4; ++j) adam_update(pp[j], mul_rounded(gg[j], gscale), mm[j], vv[j], b1, b2, eps, step_size, bc2_sqrt);
        ((float4*)p)[i] = P; ((float4*)m)[i] = M; ((float4*)v)[i] = V;
        // (keep4: a leading part the next step overwrites; [keepb_lo4, keepb_hi4): a second such range)
        if (CLEAR_G && i >= keep4 && !(i >= keepb_lo4 && i < keepb_hi4)) ((float4*)g)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (COPY && i < ncopy4) {   // bf16 shadow of the leading (weight) part of the slice, same layout
            ushort4 c;
            c.x = f32_to_bf16(pp[0]); c.y = f32_to_bf16(pp[1]); c.z = f32_to_bf16(pp[2]); c.w = f32_to_bf16(pp[3]);
            ((ushort4*)copy)[i] = c;
        }
    }
}

// ---- Adam + weight re-layout of one layer in one pass ---------------------------------------------------------
// Blocks [0, ntiles): a 64 (co) x 64 (ci) tile of the weight tensor at one tap: update p/m/v (16-B accesses along ci),
// write the dtype copy wc and, through LDS, the transpose wt [Ci][taps][Co] in whole 128-B rows.  Blocks >= ntiles: plain
// Adam over the rest of the layer's arena slice (bias, BN affine).  Saves the re-read of the fresh weights and a launch
// per layer compared with adam_kernel + wprep_kernel; needs Ci % 4 == 0 and Co % 4 == 0.
template <class T, bool CLEAR_G>
__global__ __launch_bounds__(256) void adam_wprep_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                         float* __restrict__ v, int Co, int taps, int Ci, T* __restrict__ wc,
                                                         T* __restrict__ wt, int ntiles, size_t tail_lo4, size_t tail_n4, float b1,
                                                         float b2, float eps, float step_size, float bc2_sqrt) {
    auto upd = [&](float4& P, const float4& G, float4& M, float4& V) {
        float* pp = (float*)&P; const float* gg = (const float*)&G; float* mm = (float*)&M; float* vv = (float*)&V;
#pragma unroll
        for (int j = 0; j < 4; ++j) adam_update(pp[j], gg[j], mm[j], vv[j], b1, b2, eps, step_size, bc2_sqrt);
    };
    if ((int)blockIdx.x >= ntiles) {
        const size_t nb = gridDim.x - ntiles;
        for (size_t i = (size_t)(blockIdx.x - ntiles) * 256 + threadIdx.x; i < tail_n4; i += nb * 256) {
            const size_t k = tail_lo4 + i;
            float4 P = ((float4*)p)[k], G = ((const float4*)g)[k], M = ((float4*)m)[k], V = ((float4*)v)[k];
            upd(P, G, M, V);
            ((float4*)p)[k] = P; ((float4*)m)[k] = M; ((float4*)v)[k] = V;
            if (CLEAR_G) ((float4*)g)[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        return;
    }
    __shared__ float tile[64][65];
    int b = blockIdx.x;
    const int nci = (Ci + 63) / 64, nco = (Co + 63) / 64;
    const int ci0 = (b % nci) * 64; b /= nci;
    const int co0 = (b % nco) * 64;
    const int tap = b / nco;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int ci = ci0 + tx * 4;
    float4 P[4], G[4], M[4], V[4];
    bool ok[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int co = co0 + ty + 16 * i;
        ok[i] = co < Co && ci < Ci;
        if (ok[i]) {
            const size_t o = (((size_t)co * taps + tap) * Ci + ci) / 4;
            P[i] = ((float4*)p)[o]; G[i] = ((const float4*)g)[o]; M[i] = ((float4*)m)[o]; V[i] = ((float4*)v)[o];
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = ty + 16 * i;
        if (ok[i]) {
            const size_t o = ((size_t)(co0 + row) * taps + tap) * Ci + ci;
            upd(P[i], G[i], M[i], V[i]);
            ((float4*)p)[o / 4] = P[i]; ((float4*)m)[o / 4] = M[i]; ((float4*)v)[o / 4] = V[i];
            if (CLEAR_G) ((float4*)g)[o / 4] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (wc) store4<T>(wc + o, P[i].x, P[i].y, P[i].z, P[i].w);
            tile[row][tx * 4 + 0] = P[i].x; tile[row][tx * 4 + 1] = P[i].y; tile[row][tx * 4 + 2] = P[i].z; tile[row][tx * 4 + 3] = P[i].w;
        }
    }
    if (!wt) return;   // uniform
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int cil = ty + 16 * i, co = co0 + tx * 4;
        if (ci0 + cil < Ci && co < Co)
            store4<T>(wt + ((size_t)(ci0 + cil) * taps + tap) * Co + co, tile[tx * 4 + 0][cil], tile[tx * 4 + 1][cil], tile[tx * 4 + 2][cil],
                      tile[tx * 4 + 3][cil]);
    }
}

// ---- weight re-layout: master fp32 [Co][taps][Ci] -> T copy (same layout) and T transpose [Ci][taps][Co] -----
template <class T>
__global__ __launch_bounds__(256) void wprep_kernel(const float* __restrict__ w, T* __restrict__ wc, T* __restrict__ wt, int Co,
                                                    int taps, int Ci) {
    __shared__ float tile[32][33];
    const int tap = blockIdx.z, ci0 = blockIdx.x * 32, co0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    for (int i = ty; i < 32; i += 8) {
        const int co = co0 + i, ci = ci0 + tx;
        float v = 0.f;
        if (co < Co && ci < Ci) {
            const size_t o = ((size_t)co * taps + tap) * Ci + ci;
            v = w[o];
            if (wc) wc[o] = from_f32<T>(v);
        }
        tile[i][tx] = v;
    }
    __syncthreads();
    if (wt) {
        for (int i = ty; i < 32; i += 8) {
            const int ci = ci0 + i, co = co0 + tx;
            if (co < Co && ci < Ci) wt[((size_t)ci * taps + tap) * Co + co] = from_f32<T>(tile[tx][i]);
        }
    }
}

// The same for channel counts that are multiples of 4 (every layer but the image-side ends): a 64 (co) x 64 (ci) tile per block,
// 16-byte loads along ci, the copy in 8-byte stores, the transpose through LDS in whole 128-byte rows along co -- a quarter of
// the blocks of the 32x32 form and whole cache lines on both sides (the re-layout launches are 7 % of the step's kernel time,
// beside the main chain).
template <class T>
__device__ __forceinline__ void wprep64_tile(const float* __restrict__ w, T* __restrict__ wc, T* __restrict__ wt, int Co, int taps, int Ci,
                                             int b, float (*tile)[65]) {
    const int nci = (Ci + 63) / 64, nco = (Co + 63) / 64;
    const int ci0 = (b % nci) * 64; b /= nci;
    const int co0 = (b % nco) * 64;
    const int tap = b / nco;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int ci = ci0 + tx * 4;
    float4 P[4];
    bool ok[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int co = co0 + ty + 16 * i;
        ok[i] = co < Co && ci < Ci;
        P[i] = ok[i] ? *(const float4*)(w + ((size_t)co * taps + tap) * Ci + ci) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = ty + 16 * i;
        if (ok[i] && wc) store4<T>(wc + ((size_t)(co0 + row) * taps + tap) * Ci + ci, P[i].x, P[i].y, P[i].z, P[i].w);
        tile[row][tx * 4 + 0] = P[i].x; tile[row][tx * 4 + 1] = P[i].y; tile[row][tx * 4 + 2] = P[i].z; tile[row][tx * 4 + 3] = P[i].w;
    }
    if (!wt) return;   // uniform
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int cil = ty + 16 * i, co = co0 + tx * 4;
        if (ci0 + cil < Ci && co < Co)
            store4<T>(wt + ((size_t)(ci0 + cil) * taps + tap) * Co + co, tile[tx * 4 + 0][cil], tile[tx * 4 + 1][cil], tile[tx * 4 + 2][cil],
                      tile[tx * 4 + 3][cil]);
    }
}
template <class T>
__global__ __launch_bounds__(256) void wprep64_kernel(const float* __restrict__ w, T* __restrict__ wc, T* __restrict__ wt, int Co,
                                                      int taps, int Ci) {
    __shared__ float tile[64][65];
    wprep64_tile<T>(w, wc, wt, Co, taps, Ci, (int)blockIdx.x, tile);
}

struct WprepTable {
    const float* w[16]; void* wc[16]; void* wt[16];
    int Co[16], taps[16], Ci[16], start[17];
    int t64[16];   // this tensor goes in 64x64 tiles (wprep64_tile: channel counts that are multiples of 4, aligned pointers)
};
template <class T>
__global__ __launch_bounds__(256) void wprep_batch_kernel(const WprepTable tb, int n) {
    __shared__ float tile64[64][65];
    float (*tile)[33] = (float (*)[33])tile64;
    int l = 0;
    while (l + 1 < n && (int)blockIdx.x >= tb.start[l + 1]) ++l;
    const int Co = tb.Co[l], taps = tb.taps[l], Ci = tb.Ci[l];
    const float* __restrict__ w = tb.w[l];
    T* __restrict__ wc = (T*)tb.wc[l];
    T* __restrict__ wt = (T*)tb.wt[l];
    int b = blockIdx.x - tb.start[l];
    if (tb.t64[l]) {   // block-uniform
        wprep64_tile<T>(w, wc, wt, Co, taps, Ci, b, tile64);
        return;
    }
    const int nci = (Ci + 31) / 32, nco = (Co + 31) / 32;
    const int ci0 = (b % nci) * 32; b /= nci;
    const int co0 = (b % nco) * 32;
    const int tap = b / nco;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int i = ty; i < 32; i += 8) {
        const int co = co0 + i, ci = ci0 + tx;
        float v = 0.f;
        if (co < Co && ci < Ci) {
            const size_t o = ((size_t)co * taps + tap) * Ci + ci;
            v = w[o];
            if (wc) wc[o] = from_f32<T>(v);
        }
        tile[i][tx] = v;
    }
    __syncthreads();
    if (wt) {
        for (int i = ty; i < 32; i += 8) {
            const int ci = ci0 + i, co = co0 + tx;
            if (co < Co && ci < Ci) wt[((size_t)ci * taps + tap) * Co + co] = from_f32<T>(tile[tx][i]);
        }
    }
}

// ---- batch gather: rows of the resident fp32 NHWC dataset -> fp32 batch (LL target) + T batch (conv input) ----
template <class T>
__global__ __launch_bounds__(256) void gather_batch(const float* __restrict__ data, const int* __restrict__ idx, int B,
                                                    size_t row4, float* __restrict__ xf, T* __restrict__ xt) {
    const size_t total = (size_t)B * row4;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t b = i / row4, j = i - b * row4;
        const size_t src = (size_t)(idx ? idx[b] : (int)b) * row4 + j;
        const float4 v = ((const float4*)data)[src];
        if (xf) ((float4*)xf)[i] = v;
        if (xt) store4<T>(xt + i * 4, v.x, v.y, v.z, v.w);
    }
}

// the same from a uint8 resident data set (pixel value k stands for k / 255, exactly what torchvision's ToTensor makes of an
// 8-bit image: a correctly rounded fp32 division)
template <class T>
__global__ __launch_bounds__(256) void gather_batch_u8(const unsigned char* __restrict__ data, const int* __restrict__ idx, int B,
                                                       size_t row4, float* __restrict__ xf, T* __restrict__ xt) {
    const size_t total = (size_t)B * row4;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t b = i / row4, j = i - b * row4;
        const size_t src = (size_t)(idx ? idx[b] : (int)b) * row4 + j;
        const uchar4 u = ((const uchar4*)data)[src];
        const float4 v = make_float4((float)u.x / 255.f, (float)u.y / 255.f, (float)u.z / 255.f, (float)u.w / 255.f);
        if (xf) ((float4*)xf)[i] = v;
        if (xt) store4<T>(xt + i * 4, v.x, v.y, v.z, v.w);
    }
}

// fp32 NCHW -> uint8 NHWC, counting the values that are not exactly k / 255 (a data set with any such value stays fp32)
__global__ void nchw_to_nhwc_u8(const float* __restrict__ in, unsigned char* __restrict__ out, size_t N, int C, int HW,
                                unsigned long long* __restrict__ mismatches) {
    const size_t total = N * HW * C;
    unsigned bad = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % C);
        const size_t p = i / C, n = p / HW, hw = p - n * HW;
        const float x = in[(n * C + c) * HW + hw];
        const float k = rintf(x * 255.f);
        const bool ok = k >= 0.f && k <= 255.f && k / 255.f == x;
        out[i] = ok ? (unsigned char)k : 0;
        bad += ok ? 0u : 1u;
    }
    if (bad) atomicAdd(mismatches, (unsigned long long)bad);
}

template <class T>
__global__ void cast_kernel(const float* __restrict__ in, T* __restrict__ out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) out[i] = from_f32<T>(in[i]);
}

// NCHW <-> NHWC with optional channel padding (Cp >= C, pad = 0)
__global__ void nchw_to_nhwc(const float* __restrict__ in, float* __restrict__ out, size_t N, int C, int HW, int Cp) {
    const size_t total = N * HW * Cp;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % Cp);
        const size_t p = i / Cp, n = p / HW, hw = p - n * HW;
        out[i] = c < C ? in[(n * C + c) * HW + hw] : 0.f;
    }
}
__global__ void nhwc_to_nchw_f64(const float* __restrict__ in, double* __restrict__ out, size_t N, int C, int HW, int Cp) {
    const size_t total = N * C * HW;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t hw = i % HW, q = i / HW, c = q % C, n = q / C;
        out[i] = (double)in[(n * HW + hw) * Cp + c];
    }
}

inline int ew_grid(size_t n) { return (int)std::min<size_t>(4096, std::max<size_t>(1, (n + 255) / 256)); }
// optimiser launches run beside the main chain of the step: a grid that floods every CU slows the chain's GEMMs more than it
// shortens the update, which has slack until the next forward pass (FAVAE_ADAM_BLOCKS)
// (256 blocks measured best for the layer-wise updates of an epoch: step 0.904 -> 0.895 ms; 128: 0.912, 64: 0.978)
inline int adam_grid(size_t n) {
    static const int cap = getenv("FAVAE_ADAM_BLOCKS") ? std::max(1, atoi(getenv("FAVAE_ADAM_BLOCKS"))) : 256;
    return std::min(ew_grid(n), cap);
}

}  // namespace

#define ST ((hipStream_t)stream)

#define DT_BASE(d) ((d) & 0xF)
#define RAW_IS_BF16(d) (((d) & FAVAE_RAW_BF16) != 0)
#define DACT_IS_BF16(d) (((d) & FAVAE_DACT_BF16) != 0)

extern "C" int favae_bn_stats(const float* raw, long long R, int C, float eps, float momentum, float* mean, float* invstd,
                              float* run_mean, float* run_var, double* scratch, size_t scratch_doubles, void* stream) {
    FAVAE_CHECK_ARG(raw && mean && invstd && scratch && R > 0 && C > 0 && R < (1LL << 31), "favae_bn_stats: bad argument");
    return run_colreduce<2>(R, C, scratch, scratch_doubles, StatsF<float>{raw, C, (double)R, eps, momentum, mean, invstd, run_mean, run_var}, ST);
}

extern "C" int favae_bn_eval_prep(const float* run_mean, const float* run_var, int C, float eps, float* mean, float* invstd,
                                  void* stream) {
    FAVAE_CHECK_ARG(run_mean && run_var && mean && invstd && C > 0, "favae_bn_eval_prep: bad argument");
    hipLaunchKernelGGL(bn_eval_prep, dim3(cdiv(C, 64)), dim3(64), 0, ST, run_mean, run_var, C, eps, mean, invstd);
    FAVAE_LAUNCH_CHECK();
    return FAVAE_OK;
}

extern "C" int favae_bn_lrelu_fwd(const void* raw, const float* mean, const float* invstd, const float* gamma,
                                  const float* beta, void* act, long long R, int C, int apply_act, int dtype, void* stream) {
    FAVAE_CHECK_ARG(raw && mean && invstd && gamma && beta && act && R > 0 && C > 0 && C % 4 == 0, "favae_bn_lrelu_fwd: bad argument");
    FAVAE_CHECK_ARG(!(RAW_IS_BF16(dtype) && DT_BASE(dtype) != FAVAE_BF16), "favae_bn_lrelu_fwd: FAVAE_RAW_BF16 needs FAVAE_BF16");
    const size_t nvec = (size_t)R * C / 4;
    // grid stride (vectors) must be a multiple of C/4: blocks of 256 vectors, grid a multiple of C/4 / gcd(C/4, 256)
    int c4 = C / 4, gg = c4, t = 256;
    while (t) { const int r = gg % t; gg = t; t = r; }
    const int mult = c4 / gg;
    long long grid = (long long)std::min<size_t>(2048, (nvec + 255) / 256);
    grid = std::max<long long>(mult, grid / mult * mult);
    if (DT_BASE(dtype) == FAVAE_F32)
        hipLaunchKernelGGL((bn_lrelu_fwd<float, float>), dim3((unsigned)grid), dim3(256), 0, ST, (const float*)raw, mean, invstd, gamma, beta, (float*)act, nvec, C, apply_act);
    else if (RAW_IS_BF16(dtype))
        hipLaunchKernelGGL((bn_lrelu_fwd<bf16_t, bf16_t>), dim3((unsigned)grid), dim3(256), 0, ST, (const bf16_t*)raw, mean, invstd, gamma, beta, (bf16_t*)act, nvec, C, apply_act);
    else
        hipLaunchKernelGGL((bn_lrelu_fwd<bf16_t, float>), dim3((unsigned)grid), dim3(256), 0, ST, (const float*)raw, mean, invstd, gamma, beta, (bf16_t*)act, nvec, C, apply_act);
    FAVAE_LAUNCH_CHECK();
    return FAVAE_OK;
}

extern "C" int favae_bn_eval_ab(const float* run_mean, const float* run_var, const float* gamma, const float* beta, int C, float eps,
                                float* ab, void* stream) {
    FAVAE_CHECK_ARG(run_mean && run_var && gamma && beta && ab && C > 0, "favae_bn_eval_ab: bad argument");
    hipLaunchKernelGGL(bn_eval_ab, dim3(cdiv(C, 64)), dim3(64), 0, ST, run_mean, run_var, gamma, beta, C, eps, ab);
    FAVAE_LAUNCH_CHECK();
    return FAVAE_OK;
}

extern "C" int favae_bn_lrelu_fwd_ab(const float* raw, const float* ab, void* act, long long R, int C, int dtype, void* stream) {
    FAVAE_CHECK_ARG(raw && ab && act && R > 0 && C > 0 && C % 4 == 0, "favae_bn_lrelu_fwd_ab: bad argument");
    FAVAE_CHECK_ARG(dtype == FAVAE_F32 || dtype == FAVAE_BF16, "favae_bn_lrelu_fwd_ab: dtype");
    const size_t nvec = (size_t)R * C / 4;
    int c4 = C / 4, gg = c4, t = 256;
    while (t) { const int r = gg % t; gg = t; t = r; }
    const int mult = c4 / gg;   // the grid stride must be a multiple of C: a thread keeps its 4 channels
    long long grid = (long long)std::min<size_t>(1024, (nvec + 255) / 256);
    grid = std::max<long long>(mult, grid / mult * mult);
    if (dtype == FAVAE_F32) hipLaunchKernelGGL(bn_lrelu_fwd_ab<float>, dim3((unsigned)grid), dim3(256), 0, ST, raw, ab, (float*)act, nvec, C);
    else hipLaunchKernelGGL(bn_lrelu_fwd_ab<bf16_t>, dim3((unsigned)grid), dim3(256), 0, ST, raw, ab, (bf16_t*)act, nvec, C);
    FAVAE_LAUNCH_CHECK();
    return FAVAE_OK;
}

extern "C" int favae_bn_train_fwd(const void* raw, const float* gamma, const float* beta, void* act, float* mean,
                                  float* invstd, float* run_mean, float* run_var, long long R, int C, float eps, float momentum,
                                  double* scratch, size_t scratch_doubles, int dtype, void* stream) {
    FAVAE_CHECK_ARG(raw && gamma && beta && act && mean && invstd && scratch && R > 0 && R < (1LL << 31) && C > 0 && C % 4 == 0,
                    "favae_bn_train_fwd: bad argument");
    FAVAE_CHECK_ARG(!(RAW_IS_BF16(dtype) && DT_BASE(dtype) != FAVAE_BF16), "favae_bn_train_fwd: FAVAE_RAW_BF16 needs FAVAE_BF16");
    if (RAW_IS_BF16(dtype)) {
        int rc = run_colreduce<2>(R, C, scratch, scratch_doubles,
                                  StatsF<bf16_t>{(const bf16_t*)raw, C, (double)R, eps, momentum, mean, invstd, run_mean, run_var}, ST);
        if (rc) return rc;
        return favae_bn_lrelu_fwd(raw, mean, invstd, gamma, beta, act, R, C, 1, dtype, stream);
    }
    const float* rawf = (const float*)raw;
    if (owner_applicable(R, C)) {
        if (DT_BASE(dtype) == FAVAE_F32)
            hipLaunchKernelGGL(bn_fwd_owner<float>, dim3(C / OWN_CT), dim3(OWN_THREADS), 0, ST, rawf, gamma, beta, (float*)act, mean,
                               invstd, run_mean, run_var, (int)R, C, eps, momentum);
        else
            hipLaunchKernelGGL(bn_fwd_owner<bf16_t>, dim3(C / OWN_CT), dim3(OWN_THREADS), 0, ST, rawf, gamma, beta, (bf16_t*)act, mean,
                               invstd, run_mean, run_var, (int)R, C, eps, momentum);
        FAVAE_LAUNCH_CHECK();
        return FAVAE_OK;
    }
    if (owner_fwd_big_applicable(R, C)) {
        if (DT_BASE(dtype) == FAVAE_F32)
            hipLaunchKernelGGL((bn_fwd_owner<float, OWN_MAXP_BIG>), dim3(C / OWN_CT), dim3(OWN_THREADS), 0, ST, rawf, gamma, beta, (float*)act,
                               mean, invstd, run_mean, run_var, (int)R, C, eps, momentum);
        else
            hipLaunchKernelGGL((bn_fwd_owner<bf16_t, OWN_MAXP_BIG>), dim3(C / OWN_CT), dim3(OWN_THREADS), 0, ST, rawf, gamma, beta, (bf16_t*)act,
                               mean, invstd, run_mean, run_var, (int)R, C, eps, momentum);
        FAVAE_LAUNCH_CHECK();
        return FAVAE_OK;
    }
    int rc = favae_bn_stats(rawf, R, C, eps, momentum, mean, invstd, run_mean, run_var, scratch, scratch_doubles, stream);
    if (rc) return rc;
    return favae_bn_lrelu_fwd(raw, mean, invstd, gamma, beta, act, R, C, 1, dtype, stream);
}

/* favae_bn_train_fwd for the small feature maps (one channel-owner launch) that also keeps the activation unrounded: act (dtype)
 * for the backward pass and act_f32 for a consumer that wants fp32 (fc_mu | fc_var of the bf16 step). */
extern "C" int favae_bn_train_fwd_keep32(const float* raw, const float* gamma, const float* beta, void* act, float* act_f32, float* mean,
                                         float* invstd, float* run_mean, float* run_var, long long R, int C, float eps, float momentum,
                                         int dtype, void* stream) {
    FAVAE_CHECK_ARG(raw && gamma && beta && act && act_f32 && mean && invstd && R > 0 && C > 0, "favae_bn_train_fwd_keep32: bad argument");
    FAVAE_CHECK_ARG(dtype == FAVAE_BF16 && owner_applicable(R, C), "favae_bn_train_fwd_keep32: bf16, at most %d rows, C a multiple of %d",
                    OWN_RG * OWN_MAXP, OWN_CT);
    hipLaunchKernelGGL(bn_fwd_owner<bf16_t>, dim3(C / OWN_CT), dim3(OWN_THREADS), 0, ST, raw, gamma, beta, (bf16_t*)act, mean, invstd,
                       run_mean, run_var, (int)R, C, eps, momentum, act_f32);
    FAVAE_LAUNCH_CHECK();
    return FAVAE_OK;
}
extern "C" int favae_bn_train_fwd_keep32_ok(long long R, int C) { return owner_applicable(R, C) ? 1 : 0; }

static int bn_train_fwd_bins_impl(const void* raw, const double* bins, int nbins, const float* gamma, const float* beta,
                                       void* act, float* mean, float* invstd, float* run_mean, float* run_var, long long R,
                                       int C, float eps, float momentum, int dtype, float* ab_out, void* stream) {
    FAVAE_CHECK_ARG(raw && bins && gamma && beta && (act || ab_out) && mean && invstd && R > 0 && R < (1LL << 31) && nbins >= 1,
                    "favae_bn_train_fwd_bins: bad argument");
    FAVAE_CHECK_ARG(bins_applicable(C), "favae_bn_train_fwd_bins: C = %d not supported (64, 128, 256 ...)", C);
    FAVAE_CHECK_ARG(DT_BASE(dtype) == FAVAE_F32 || DT_BASE(dtype) == FAVAE_BF16, "favae_bn_train_fwd_bins: dtype");
    FAVAE_CHECK_ARG(!(RAW_IS_BF16(dtype) && DT_BASE(dtype) != FAVAE_BF16), "favae_bn_train_fwd_bins: FAVAE_RAW_BF16 needs FAVAE_BF16");
    const size_t nvec = act ? (size_t)R * C / 4 : 0;   // act == NULL: the statistics only (one block)
    int c4 = C / 4, gg = c4, t = 256;
    while (t) { const int r = gg % t; gg = t; t = r; }
    const int mult = act ? c4 / gg : 1;
    static const int grid_cap = getenv("FAVAE_BNF_GRID") ? atoi(getenv("FAVAE_BNF_GRID")) : 1024;   // every block sums the bins itself: fewer, longer blocks (2048 -> 1024: -0.7 % step time)
    long long grid = (long long)std::min<size_t>(grid_cap, (nvec + 255) / 256);
    grid = std::max<long long>(mult, grid / mult * mult);
#define FB_ARGS bins, nbins, gamma, beta
    if (DT_BASE(dtype) == FAVAE_F32)
        hipLaunchKernelGGL((bn_lrelu_fwd_bins<float, float>), dim3((unsigned)grid), dim3(256), 0, ST, (const float*)raw, FB_ARGS, (float*)act,
                           mean, invstd, run_mean, run_var, nvec, C, (double)R, eps, momentum, ab_out);
    else if (RAW_IS_BF16(dtype))
        hipLaunchKernelGGL((bn_lrelu_fwd_bins<bf16_t, bf16_t>), dim3((unsigned)grid), dim3(256), 0, ST, (const bf16_t*)raw, FB_ARGS, (bf16_t*)act,
                           mean, invstd, run_mean, run_var, nvec, C, (double)R, eps, momentum, ab_out);
    else
        hipLaunchKernelGGL((bn_lrelu_fwd_bins<bf16_t, float>), dim3((unsigned)grid), dim3(256), 0, ST, (const float*)raw, FB_ARGS, (bf16_t*)act,
                           mean, invstd, run_mean, run_var, nvec, C, (double)R, eps, momentum, ab_out);
#undef FB_ARGS
    FAVAE_LAUNCH_CHECK();
    return FAVAE_OK;
}
extern "C" int favae_bn_train_fwd_bins(const void* raw, const double* bins, int nbins, const float* gamma, const float* beta,
                                       void* act, float* mean, float* invstd, float* run_mean, float* run_var, long long R,
                                       int C, float eps, float momentum, int dtype, void* stream) {
    return bn_train_fwd_bins_impl(raw, bins, nbins, gamma, beta, act, mean, invstd, run_mean, run_var, R, C, eps, momentum, dtype, nullptr, stream);
}
extern "C" int favae_bn_train_fwd_bins_ab(const void* raw, const double* bins, int nbins, const float* gamma, const float* beta,
                                          void* act, float* mean, float* invstd, float* ab, float* run_mean, float* run_var, long long R,
                                          int C, float eps, float momentum, int dtype, void* stream) {
    FAVAE_CHECK_ARG(ab, "favae_bn_train_fwd_bins_ab: null ab");
    return bn_train_fwd_bins_impl(raw, bins, nbins, gamma, beta, act, mean, invstd, run_mean, run_var, R, C, eps, momentum, dtype, ab, stream);
}

template <class T, class RT, class DT>
static int bn_bwd_impl(const void* dact_, const void* act_, const void* raw_, const float* mean, const float* invstd,
                       const float* gamma, void* draw_, float* dgamma, float* dbeta, float* dbias, long long R, int C,
                       double* scratch, size_t cap, float* s1s2, hipStream_t st) {
    const DT* dact = (const DT*)dact_; const T* act = (const T*)act_; const RT* raw = (const RT*)raw_; T* draw = (T*)draw_;
    if constexpr (std::is_same<RT, float>::value && std::is_same<DT, float>::value) {
        if (owner_applicable(R, C)) {
            hipLaunchKernelGGL(bn_bwd_owner<T>, dim3(C / OWN_CT), dim3(OWN_THREADS), 0, st, dact, act, raw, mean, invstd, gamma, draw,
                               dgamma, dbeta, dbias, (int)R, C);
            FAVAE_LAUNCH_CHECK();
            return FAVAE_OK;
        }
    }
    int rc = run_colreduce<2>(R, C, scratch, cap, BnBwdRedF<T, RT, DT>{dact, act, raw, mean, invstd, C, dgamma, dbeta, s1s2}, st);
    if (rc) return rc;
    return run_colreduce<1>(R, C, scratch, cap,
                            BnBwdApplyF<T, RT, DT>{dact, act, raw, mean, invstd, gamma, s1s2, draw, C, (float)(1.0 / (double)R), dbias,
                                                   nullptr, 0, nullptr, nullptr}, st);
}

template <class T, class RT, class DT>
static int bn_bwd_bins_impl(const void* dact, const void* act, const void* raw, const float* mean, const float* invstd,
                            const float* gamma, const double* bins, int nbins, void* draw, float* dgamma, float* dbeta,
                            float* dbias, long long R, int C, double* scratch, size_t cap, hipStream_t st, bool defer, const float* ab) {
    // algorithmic bytes: dact + raw (+ act) read, draw written
    ProfScope prof(7, (double)R * C * (sizeof(DT) + sizeof(RT) + (ab ? 1 : 2) * sizeof(T)), st);
    if (defer)
        return run_colreduce<1>(R, C, scratch, cap,
                                BnBwdApplyF<T, RT, DT, true>{(const DT*)dact, (const T*)act, (const RT*)raw, mean, invstd, gamma, nullptr, (T*)draw,
                                                             C, (float)(1.0 / (double)R), nullptr, bins, nbins, dgamma, dbeta, ab}, st);
    return run_colreduce<1>(R, C, scratch, cap,
                            BnBwdApplyF<T, RT, DT>{(const DT*)dact, (const T*)act, (const RT*)raw, mean, invstd, gamma, nullptr, (T*)draw, C,
                                                   (float)(1.0 / (double)R), dbias, bins, nbins, dgamma, dbeta, ab}, st);
}

static int bn_lrelu_bwd_bins_any(const void* dact, const void* act, const void* raw, const float* mean,
                                       const float* invstd, const float* gamma, const double* bins, int nbins, void* draw,
                                       float* dgamma, float* dbeta, float* dbias, long long R, int C, double* scratch,
                                       size_t scratch_doubles, int dtype, const float* ab, void* stream) {
    FAVAE_CHECK_ARG(dact && (act || ab) && raw && mean && invstd && gamma && bins && draw && dgamma && dbeta && scratch && nbins >= 1,
                    "favae_bn_lrelu_bwd_bins: null pointer");
    FAVAE_CHECK_ARG(R > 0 && R < (1LL << 31) && C > 0, "favae_bn_lrelu_bwd_bins: bad shape");
    FAVAE_CHECK_ARG(DT_BASE(dtype) == FAVAE_F32 || DT_BASE(dtype) == FAVAE_BF16, "favae_bn_lrelu_bwd_bins: dtype");
    FAVAE_CHECK_ARG(!((RAW_IS_BF16(dtype) || DACT_IS_BF16(dtype)) && (DT_BASE(dtype) != FAVAE_BF16 || C % 4 != 0)),
                    "favae_bn_lrelu_bwd_bins: FAVAE_RAW_BF16 / FAVAE_DACT_BF16 need FAVAE_BF16 and C %% 4 == 0");
    const bool defer = (dtype & FAVAE_DBIAS_DEFER) != 0;   // scratch keeps the per-slab column sums, dbias is not written
#define BB_ARGS dact, act, raw, mean, invstd, gamma, bins, nbins, draw, dgamma, dbeta, dbias, R, C, scratch, scratch_doubles, ST, defer, ab
    if (DT_BASE(dtype) == FAVAE_F32) return bn_bwd_bins_impl<float, float, float>(BB_ARGS);
    if (RAW_IS_BF16(dtype) && DACT_IS_BF16(dtype)) return bn_bwd_bins_impl<bf16_t, bf16_t, bf16_t>(BB_ARGS);
    if (RAW_IS_BF16(dtype)) return bn_bwd_bins_impl<bf16_t, bf16_t, float>(BB_ARGS);
    if (DACT_IS_BF16(dtype)) return bn_bwd_bins_impl<bf16_t, float, bf16_t>(BB_ARGS);
    return bn_bwd_bins_impl<bf16_t, float, float>(BB_ARGS);
#undef BB_ARGS
}
extern "C" int favae_bn_lrelu_bwd_bins(const void* dact, const void* act, const void* raw, const float* mean,
                                       const float* invstd, const float* gamma, const double* bins, int nbins, void* draw,
                                       float* dgamma, float* dbeta, float* dbias, long long R, int C, double* scratch,
                                       size_t scratch_doubles, int dtype, void* stream) {
    FAVAE_CHECK_ARG(act, "favae_bn_lrelu_bwd_bins: null pointer");
    return bn_lrelu_bwd_bins_any(dact, act, raw, mean, invstd, gamma, bins, nbins, draw, dgamma, dbeta, dbias, R, C, scratch, scratch_doubles, dtype,
                                 nullptr, stream);
}
extern "C" int favae_bn_lrelu_bwd_bins_ab(const void* dact, const void* raw, const float* mean, const float* invstd, const float* gamma,
                                          const float* ab, const double* bins, int nbins, void* draw, float* dgamma, float* dbeta, float* dbias,
                                          long long R, int C, double* scratch, size_t scratch_doubles, int dtype, void* stream) {
    FAVAE_CHECK_ARG(ab, "favae_bn_lrelu_bwd_bins_ab: null ab");
    return bn_lrelu_bwd_bins_any(dact, nullptr, raw, mean, invstd, gamma, bins, nbins, draw, dgamma, dbeta, dbias, R, C, scratch, scratch_doubles, dtype,
                                 ab, stream);
}

extern "C" int favae_bn_lrelu_bwd(const void* dact, const void* act, const void* raw, const float* mean,
                                  const float* invstd, const float* gamma, void* draw, float* dgamma, float* dbeta,
                                  float* dbias, long long R, int C, double* scratch, size_t scratch_doubles, float* s1s2,
                                  int dtype, void* stream) {
    FAVAE_CHECK_ARG(dact && act && raw && mean && invstd && gamma && draw && dgamma && dbeta && scratch && s1s2,
                    "favae_bn_lrelu_bwd: null pointer");
    FAVAE_CHECK_ARG(R > 0 && R < (1LL << 31) && C > 0, "favae_bn_lrelu_bwd: bad shape");
    FAVAE_CHECK_ARG(!((RAW_IS_BF16(dtype) || DACT_IS_BF16(dtype)) && (DT_BASE(dtype) != FAVAE_BF16 || C % 4 != 0)),
                    "favae_bn_lrelu_bwd: FAVAE_RAW_BF16 / FAVAE_DACT_BF16 need FAVAE_BF16 and C %% 4 == 0");
#define BWD_ARGS dact, act, raw, mean, invstd, gamma, draw, dgamma, dbeta, dbias, R, C, scratch, scratch_doubles, s1s2, ST
    if (DT_BASE(dtype) == FAVAE_F32) return bn_bwd_impl<float, float, float>(BWD_ARGS);
    if (RAW_IS_BF16(dtype) && DACT_IS_BF16(dtype)) return bn_bwd_impl<bf16_t, bf16_t, bf16_t>(BWD_ARGS);
    if (RAW_IS_BF16(dtype)) return bn_bwd_impl<bf16_t, bf16_t, float>(BWD_ARGS);
    if (DACT_IS_BF16(dtype)) return bn_bwd_impl<bf16_t, float, bf16_t>(BWD_ARGS);
    return bn_bwd_impl<bf16_t, float, float>(BWD_ARGS);
#undef BWD_ARGS
}

// the second half of favae_conv_out_ll with FAVAE_DBIAS_DEFER as a launch of its own (one block; the training step lets the
// sums ride with the layer's weight gradient instead: favae_conv_out_ll_ride)
__global__ __launch_bounds__(256) void conv_out_ll_finish_kernel(const double* __restrict__ partial, int blocks, double off0, double scale0,
                                                                 int CO, double* __restrict__ rec_err, float* __restrict__ dbias) {
    __shared__ double sm[32][8];
    const int ch = threadIdx.x & 7, part = threadIdx.x >> 3;
    double t = 0.0;
    if (ch < 5)
        for (int k = part; k < blocks; k += 32) t += partial[(size_t)k * 5 + ch];
    sm[part][ch] = t;
    __syncthreads();
    if (threadIdx.x < 5) {
        double u = 0.0;
        for (int k = 0; k < 32; ++k) u += sm[k][threadIdx.x];
        if (threadIdx.x == 0) *rec_err = off0 + scale0 * u;
        else if ((int)threadIdx.x - 1 < CO) dbias[threadIdx.x - 1] = (float)u;
    }
}

extern "C" int favae_conv_out_ll_finish(const double* partial, int Nb, int H, int W, int Cout, float var_x, double* rec_err, float* dbias,
                                        void* stream) {
    FAVAE_CHECK_ARG(partial && rec_err && dbias && Nb > 0 && H > 0 && W > 0 && Cout >= 1 && Cout <= 4 && var_x > 0, "favae_conv_out_ll_finish: bad argument");
    const int blocks = cdiv(Nb * H * W, 256);
    const double D = (double)H * W * Cout;
    hipLaunchKernelGGL(conv_out_ll_finish_kernel, dim3(1), dim3(256), 0, ST, partial, blocks, 0.5 * (D * log(2.0 * M_PI) + D * log((double)var_x)),
                       0.5 / ((double)var_x * Nb), Cout, rec_err, dbias);
    FAVAE_LAUNCH_CHECK();
    return FAVAE_OK;
}

extern "C" int favae_bn_lrelu_bwd_slabs(long long R, int C) {
    return (R > 0 && C > 0) ? red_geom(R, C).nslab : 0;
}

extern "C" int favae_bn_dbias_finish(const double* partial, int nslab, int C, float* dbias, void* stream) {
    FAVAE_CHECK_ARG(partial && dbias && nslab >= 1 && C >= 1, "favae_bn_dbias_finish: bad argument");
    hipLaunchKernelGGL(slab_sums_finish, dim3(cdiv(C, 64)), dim3(256), 0, ST, partial, nslab, C, dbias);
    FAVAE_LAUNCH_CHECK();
    return FAVAE_OK;
}

extern "C" int favae_colsum(const void* x, long long R, int C, float* out, double* scratch, size_t scratch_doubles, int dtype,
                            void* stream) {
    FAVAE_CHECK_ARG(x && out && scratch && R > 0 && R < (1LL << 31) && C > 0, "favae_colsum: bad argument");
    return dtype == FAVAE_F32 ? run_colreduce<1>(R, C, scratch, scratch_doubles, ColsumF<float>{(const float*)x, C, out}, ST)
                              : run_colreduce<1>(R, C, scratch, scratch_doubles, ColsumF<bf16_t>{(const bf16_t*)x, C, out}, ST);
}

extern "C" int favae_colsum_cast(const float* x, void* y, long long R, int C, float* out, double* scratch, size_t scratch_doubles,
                                 int dtype, void* stream) {
    FAVAE_CHECK_ARG(x && y && out && scratch && R > 0 && R < (1LL << 31) && C > 0, "favae_colsum_cast: bad argument");
    return dtype == FAVAE_F32 ? run_colreduce<1>(R, C, scratch, scratch_doubles, ColsumCastF<float>{x, (float*)y, C, out}, ST)
                              : run_colreduce<1>(R, C, scratch, scratch_doubles, ColsumCastF<bf16_t>{x, (bf16_t*)y, C, out}, ST);
}

extern "C" int favae_reparam_kl_fwd(const float* fc, const float* eps, const double* prior, const int* idx, float tau, int B,
                                    int z, void* zlat, float* stdq, double* kl_out, int dtype, void* stream) {
    FAVAE_CHECK_ARG(fc && eps && zlat && stdq && kl_out && B > 0 && z > 0 && tau > 0, "favae_reparam_kl_fwd: bad argument");
    if (dtype == FAVAE_F32)
        hipLaunchKernelGGL(reparam_kl_fwd<float>, dim3(1), dim3(RKF_THREADS), 0, ST, fc, eps, prior, idx, tau, B, z, (float*)zlat, stdq, kl_out);
    else
        hipLaunchKernelGGL(reparam_kl_fwd<bf16_t>, dim3(1), dim3(RKF_THREADS), 0, ST, fc, eps, prior, idx, tau, B, z, (bf16_t*)zlat, stdq, kl_out);
    FAVAE_LAUNCH_CHECK();
    return FAVAE_OK;
}

extern "C" int favae_reparam_kl_decoder_input(const float* fc, const float* eps, const double* prior, const int* idx, float tau,
                                              int B, int z, void* zlat, float* stdq, double* kl_out, const void* w,
                                              const float* bias, void* out, int N, int dtype, void* stream) {
    FAVAE_CHECK_ARG(fc && eps && zlat && stdq && kl_out && w && bias && out && B > 0 && z > 0 && N > 0 && tau > 0,
                    "favae_reparam_kl_decoder_input: bad argument");
    FAVAE_CHECK_ARG(z <= RKD_ZMAX_BIG && (size_t)B * z * sizeof(float) <= 48 * 1024, "favae_reparam_kl_decoder_input: z <= %d, B*z <= 12288", RKD_ZMAX_BIG);
    FAVAE_CHECK_ARG(dtype == FAVAE_F32 || dtype == FAVAE_BF16, "favae_reparam_kl_decoder_input: dtype");
    const size_t lds = (size_t)B * z * sizeof(float);
    const dim3 grid((N + 63) / 64);
#define RKD_LAUNCH(TT, ZM, TH) hipLaunchKernelGGL((reparam_kl_decin<TT, ZM, TH>), grid, dim3(TH), lds, ST, fc, eps, prior, idx, tau, B, z, (TT*)zlat, \
                                                  stdq, kl_out, (const TT*)w, bias, (TT*)out, N)
    if (z <= RKD_ZMAX) { if (dtype == FAVAE_F32) RKD_LAUNCH(float, RKD_ZMAX, RKF_THREADS); else RKD_LAUNCH(bf16_t, RKD_ZMAX, RKF_THREADS); }
    else { if (dtype == FAVAE_F32) RKD_LAUNCH(float, RKD_ZMAX_BIG, 256); else RKD_LAUNCH(bf16_t, RKD_ZMAX_BIG, 256); }
#undef RKD_LAUNCH
    FAVAE_LAUNCH_CHECK();
    return FAVAE_OK;
}

extern "C" int favae_reparam_kl_bwd(const float* fc, const float* eps, const double* prior, const int* idx, const float* stdq,
                                    const float* dz, float tau, float beta, int B, int z, void* dfc, int dtype, void* stream) {
    FAVAE_CHECK_ARG(fc && eps && stdq && dz && dfc && B > 0 && z > 0 && tau > 0, "favae_reparam_kl_bwd: bad argument");
    const int grid = cdiv(B * z, 256);
    if (dtype == FAVAE_F32)
        hipLaunchKernelGGL(reparam_kl_bwd<float>, dim3(grid), dim3(256), 0, ST, fc, eps, prior, idx, stdq, dz, tau, beta, B, z, (float*)dfc);
    else
        hipLaunchKernelGGL(reparam_kl_bwd<bf16_t>, dim3(grid), dim3(256), 0, ST, fc, eps, prior, idx, stdq, dz, tau, beta, B, z, (bf16_t*)dfc);
    FAVAE_LAUNCH_CHECK();
    return FAVAE_OK;
}

extern "C" int favae_reparam_kl_bwd_bias(const float* fc, const float* eps, const double* prior, const int* idx,
                                         const float* stdq, const float* dz, float tau, float beta, int B, int z, void* dfc,
                                         float* dbias, int dtype, void* stream) {
    FAVAE_CHECK_ARG(fc && eps && stdq && dz && dfc && dbias && B > 0 && z > 0 && 2 * z <= 256 && (long long)B * 2 * z <= RKB_MAX,
                    "favae_reparam_kl_bwd_bias: bad argument (z <= 128, B * 2z <= 8192)");
    if (dtype == FAVAE_F32)
        hipLaunchKernelGGL(reparam_kl_bwd_bias<float>, dim3(1), dim3(RKB_THREADS), 0, ST, fc, eps, prior, idx, stdq, dz, tau, beta, B, z, (float*)dfc, dbias);
    else
        hipLaunchKernelGGL(reparam_kl_bwd_bias<bf16_t>, dim3(1), dim3(RKB_THREADS), 0, ST, fc, eps, prior, idx, stdq, dz, tau, beta, B, z, (bf16_t*)dfc, dbias);
    FAVAE_LAUNCH_CHECK();
    return FAVAE_OK;
}

extern "C" int favae_decoder_input_bwd_reparam(const float* dact, const void* w, void* draw, float* dbias_di, float* dz,
                                               const float* fc, const float* eps, const double* prior, const int* idx,
                                               const float* stdq, float tau, float beta, int B, int z, int N, void* dfc,
                                               float* dbias_fc, int dtype, void* stream) {
    FAVAE_CHECK_ARG(dact && w && draw && dbias_di && dz && fc && eps && stdq && dfc && dbias_fc && B > 0 && z > 0 && N > 0 && tau > 0,
                    "favae_decoder_input_bwd_reparam: bad argument");
    const bool big = z > RKD_ZMAX;
    const int vmax = big ? DecinBwdSmem<RKD_ZMAX_BIG>::VALS : DecinBwdSmem<RKD_ZMAX>::VALS;
    FAVAE_CHECK_ARG(z <= RKD_ZMAX_BIG && N % 64 == 0 && (long long)B * 2 * z <= vmax,
                    "favae_decoder_input_bwd_reparam: z <= %d, N a multiple of 64, B * 2z <= %d", RKD_ZMAX_BIG, vmax);
    FAVAE_CHECK_ARG(dtype == FAVAE_F32 || dtype == FAVAE_BF16, "favae_decoder_input_bwd_reparam: dtype");
    unsigned* ticket = nullptr;
    int trc = ticket_base(1, &ticket);
    if (trc) return trc;
#define DBR_LAUNCH(TT, ZM) hipLaunchKernelGGL((decin_bwd_reparam<TT, ZM>), dim3(N / 64), dim3(RKB_THREADS), DecinBwdSmem<ZM>::BYTES, ST, dact, (const TT*)w, \
                                              (TT*)draw, dbias_di, dz, ticket, fc, eps, prior, idx, stdq, tau, beta, B, z, N, (TT*)dfc, dbias_fc)
    if (!big) {
        if (dtype == FAVAE_F32) DBR_LAUNCH(float, RKD_ZMAX); else DBR_LAUNCH(bf16_t, RKD_ZMAX);
    } else {
        // 72 KB of dynamic LDS: above the 64 KB a kernel gets without asking
        static bool raised = false;
        if (!raised) {
            FAVAE_HIP(hipFuncSetAttribute((const void*)decin_bwd_reparam<float, RKD_ZMAX_BIG>, hipFuncAttributeMaxDynamicSharedMemorySize, DecinBwdSmem<RKD_ZMAX_BIG>::BYTES));
            FAVAE_HIP(hipFuncSetAttribute((const void*)decin_bwd_reparam<bf16_t, RKD_ZMAX_BIG>, hipFuncAttributeMaxDynamicSharedMemorySize, DecinBwdSmem<RKD_ZMAX_BIG>::BYTES));
            raised = true;
        }
        if (dtype == FAVAE_F32) DBR_LAUNCH(float, RKD_ZMAX_BIG); else DBR_LAUNCH(bf16_t, RKD_ZMAX_BIG);
    }
#undef DBR_LAUNCH
    FAVAE_LAUNCH_CHECK();
    return FAVAE_OK;
}

extern "C" int favae_encode_head(const float* fc, int B, int z, double* xmean, double* xcov, void* stream) {
    FAVAE_CHECK_ARG(fc && xmean && xcov && B > 0 && z > 0, "favae_encode_head: bad argument");
    hipLaunchKernelGGL(encode_head, dim3(cdiv(B * z, 256)), dim3(256), 0, ST, fc, B, z, xmean, xcov);
    FAVAE_LAUNCH_CHECK();
    return FAVAE_OK;
}

extern "C" int favae_sigmoid_ll(const float* raw, const float* x, int B, long long D, float var_x, void* draw, double* rec_err,
                                double* scratch, size_t scratch_doubles, int dtype, void* stream) {
    FAVAE_CHECK_ARG(raw && x && draw && rec_err && scratch && B > 0 && D > 0 && var_x > 0, "favae_sigmoid_ll: bad argument");
    const size_t n = (size_t)B * D;
    const int grid = (int)std::min<size_t>(1024, (n + 255) / 256);
    FAVAE_CHECK_ARG((size_t)grid <= scratch_doubles, "favae_sigmoid_ll: scratch too small");
    const float ivb = 1.f / (var_x * (float)B);
    if (dtype == FAVAE_F32)
        hipLaunchKernelGGL(sigmoid_ll<float>, dim3(grid), dim3(256), 0, ST, raw, x, n, ivb, (float*)draw, scratch);
    else
        hipLaunchKernelGGL(sigmoid_ll<bf16_t>, dim3(grid), dim3(256), 0, ST, raw, x, n, ivb, (bf16_t*)draw, scratch);
    FAVAE_LAUNCH_CHECK();
    hipLaunchKernelGGL(ll_finish, dim3(1), dim3(64), 0, ST, scratch, grid, (double)D, (double)var_x, (double)B, rec_err);
    FAVAE_LAUNCH_CHECK();
    return FAVAE_OK;
}

extern "C" int favae_sigmoid_ll_bias(const float* raw, const float* x, int B, long long D, int C, float var_x, void* draw,
                                     double* rec_err, float* dbias, double* scratch, size_t scratch_doubles, int dtype,
                                     void* stream) {
    FAVAE_CHECK_ARG(raw && x && draw && rec_err && scratch && B > 0 && D > 0 && var_x > 0, "favae_sigmoid_ll_bias: bad argument");
    FAVAE_CHECK_ARG(C >= 1 && C <= 4 && D % C == 0, "favae_sigmoid_ll_bias: 1..4 interleaved channels");
    const size_t n = (size_t)B * D;
    const int grid = (int)std::min<size_t>(256, (n + 255) / 256);   // one batch of partial loads per thread of the last block
    FAVAE_CHECK_ARG((size_t)grid * 5 <= scratch_doubles, "favae_sigmoid_ll_bias: scratch too small");
    unsigned* ticket = nullptr;
    int trc = ticket_base(1, &ticket);
    if (trc) return trc;
    const float ivb = 1.f / (var_x * (float)B);
    if (dtype == FAVAE_F32)
        hipLaunchKernelGGL(sigmoid_ll_bias<float>, dim3(grid), dim3(256), 0, ST, raw, x, n, C, ivb, (float*)draw, scratch, ticket,
                           (double)D, (double)var_x, (double)B, rec_err, dbias);
    else
        hipLaunchKernelGGL(sigmoid_ll_bias<bf16_t>, dim3(grid), dim3(256), 0, ST, raw, x, n, C, ivb, (bf16_t*)draw, scratch, ticket,
                           (double)D, (double)var_x, (double)B, rec_err, dbias);
    FAVAE_LAUNCH_CHECK();
    return FAVAE_OK;
}

extern "C" int favae_conv_out_wprep(const void* w, void* wk, int Cout, int dtype, void* stream) {
    FAVAE_CHECK_ARG(w && wk && Cout >= 1 && Cout <= 4 && (dtype == FAVAE_F32 || dtype == FAVAE_BF16), "favae_conv_out_wprep: bad argument");
    FAVAE_CHECK_ARG(((uintptr_t)wk % 16) == 0, "favae_conv_out_wprep: 16-byte alignment");
    const int n = 9 * 16 * 16;
    if (dtype == FAVAE_F32) hipLaunchKernelGGL(conv_out_wprep<float>, dim3(cdiv(n, 256)), dim3(256), 0, ST, (const float*)w, (float*)wk, Cout);
    else hipLaunchKernelGGL(conv_out_wprep<bf16_t>, dim3(cdiv(n, 256)), dim3(256), 0, ST, (const bf16_t*)w, (bf16_t*)wk, Cout);
    FAVAE_LAUNCH_CHECK();
    return FAVAE_OK;
}

extern "C" int favae_conv_out_ll(const void* act, const void* wk, const float* bias, const float* x, int Nb, int H, int W,
                                 int Cout, float var_x, float* raw, void* draw, double* rec_err, float* dbias, double* scratch,
                                 size_t scratch_doubles, int dtype, void* stream) {
    FAVAE_CHECK_ARG(act && wk && x && raw && draw && rec_err && scratch && Nb > 0 && H > 0 && W > 0 && var_x > 0, "favae_conv_out_ll: bad argument");
    FAVAE_CHECK_ARG(Cout >= 1 && Cout <= 4, "favae_conv_out_ll: 1..4 output channels");
    FAVAE_CHECK_ARG((dtype & ~FAVAE_DBIAS_DEFER) == FAVAE_F32 || (dtype & ~FAVAE_DBIAS_DEFER) == FAVAE_BF16, "favae_conv_out_ll: dtype");
    FAVAE_CHECK_ARG(((uintptr_t)act % 16) == 0 && ((uintptr_t)wk % 16) == 0 && (size_t)Nb * H * W < (1u << 31), "favae_conv_out_ll: alignment / size");
    const int total = Nb * H * W, grid = cdiv(total, 256);
    FAVAE_CHECK_ARG((size_t)grid * 5 <= scratch_doubles, "favae_conv_out_ll: scratch too small (%zu < %zu doubles)", scratch_doubles, (size_t)grid * 5);
    unsigned* ticket = nullptr;
    const bool defer = (dtype & FAVAE_DBIAS_DEFER) != 0;   // rec_err / dbias are left to favae_conv_out_ll_finish / _ride (ticket == NULL in the kernel)
    dtype &= ~FAVAE_DBIAS_DEFER;
    if (!defer) {
        int trc = ticket_base(1, &ticket);
        if (trc) return trc;
    }
    const float ivb = 1.f / (var_x * (float)Nb);
    const double D = (double)H * W * Cout;
#define CO_LAUNCH(T_, CO_)                                                                                                   \
    hipLaunchKernelGGL((conv_out_ll<T_, CO_>), dim3(grid), dim3(256), 0, ST, (const T_*)act, (const T_*)wk, bias, x, H, W, total, ivb, raw, \
                       (T_*)draw, scratch, ticket, D, (double)var_x, (double)Nb, rec_err, dbias)
#define CO_ROWS(CO_)                                                                                                         \
    hipLaunchKernelGGL((conv_out_ll_rows<CO_>), dim3(grid), dim3(256), 0, ST, (const bf16_t*)act, (const bf16_t*)wk, bias, x, H, W, total, \
                       ivb, raw, (bf16_t*)draw, scratch, ticket, D, (double)var_x, (double)Nb, rec_err, dbias)
    if (dtype == FAVAE_F32) {
        switch (Cout) { case 1: CO_LAUNCH(float, 1); break; case 2: CO_LAUNCH(float, 2); break; case 3: CO_LAUNCH(float, 3); break; default: CO_LAUNCH(float, 4); }
    } else if (W % 64 == 0) {   // a wave = 64 consecutive pixels of one image row
        switch (Cout) { case 1: CO_ROWS(1); break; case 2: CO_ROWS(2); break; case 3: CO_ROWS(3); break; default: CO_ROWS(4); }
    } else {
        switch (Cout) { case 1: CO_LAUNCH(bf16_t, 1); break; case 2: CO_LAUNCH(bf16_t, 2); break; case 3: CO_LAUNCH(bf16_t, 3); break; default: CO_LAUNCH(bf16_t, 4); }
    }
#undef CO_ROWS
#undef CO_LAUNCH
    FAVAE_LAUNCH_CHECK();
    return FAVAE_OK;
}

/* favae_conv_out_ll whose input is the conv output `raw_in` [Nb][H][W][64] (bf16) in front of final_layer.1/.2 (BatchNorm2d(train) + LeakyReLU,
 * celeba_deepvanilla_vae.py:72-73): the activation lrelu(raw_in * a + b), rounded to bf16, is formed at the operand load from ab
 * [groups][2][64] (a | b per BatchNorm group of group_images images; 0: one group), so the activation tensor need not exist.  bf16, W a multiple of 64. */
extern "C" int favae_conv_out_ll_bn(const void* raw_in, const float* ab, int group_images, const void* wk, const float* bias, const float* x, int Nb,
                                    int H, int W, int Cout, float var_x, float* raw, void* draw, double* rec_err, float* dbias, double* scratch,
                                    size_t scratch_doubles, int dtype, void* stream) {
    FAVAE_CHECK_ARG(raw_in && ab && wk && x && raw && draw && rec_err && scratch && Nb > 0 && H > 0 && W > 0 && var_x > 0 && group_images >= 0,
                    "favae_conv_out_ll_bn: bad argument");
    FAVAE_CHECK_ARG(Cout >= 1 && Cout <= 4, "favae_conv_out_ll_bn: 1..4 output channels");
    FAVAE_CHECK_ARG((dtype & ~FAVAE_DBIAS_DEFER) == FAVAE_BF16 && W % 64 == 0, "favae_conv_out_ll_bn: bf16, image width a multiple of 64");
    FAVAE_CHECK_ARG(((uintptr_t)raw_in % 16) == 0 && ((uintptr_t)wk % 16) == 0 && ((uintptr_t)ab % 16) == 0 && (size_t)Nb * H * W < (1u << 31),
                    "favae_conv_out_ll_bn: alignment / size");
    const int total = Nb * H * W, grid = cdiv(total, 256);
    FAVAE_CHECK_ARG((size_t)grid * 5 <= scratch_doubles, "favae_conv_out_ll_bn: scratch too small (%zu < %zu doubles)", scratch_doubles, (size_t)grid * 5);
    unsigned* ticket = nullptr;
    const bool defer = (dtype & FAVAE_DBIAS_DEFER) != 0;
    if (!defer) {
        int trc = ticket_base(1, &ticket);
        if (trc) return trc;
    }
    const float ivb = 1.f / (var_x * (float)Nb);
    const double D = (double)H * W * Cout;
#define CO_ROWS(CO_)                                                                                                                        \
    hipLaunchKernelGGL((conv_out_ll_rows<CO_, true>), dim3(grid), dim3(256), 0, ST, (const bf16_t*)raw_in, (const bf16_t*)wk, bias, x, H, W, total, \
                       ivb, raw, (bf16_t*)draw, scratch, ticket, D, (double)var_x, (double)Nb, rec_err, dbias, ab, group_images)
    switch (Cout) { case 1: CO_ROWS(1); break; case 2: CO_ROWS(2); break; case 3: CO_ROWS(3); break; default: CO_ROWS(4); }
#undef CO_ROWS
    FAVAE_LAUNCH_CHECK();
    return FAVAE_OK;
}

/* final_layer.3 (Conv2d 64 -> Cout <= 4, 3x3, stride 1) alone, for the eval decoder: the convolution of favae_conv_out_ll (a wave per
 * 64 pixels of an image row, 4x4x4 MFMAs) without the likelihood: raw [Nb][H][W][Cout] fp32.  bf16, W a multiple of 64. */
extern "C" int favae_conv_out_raw(const void* act, const void* wk, const float* bias, int Nb, int H, int W, int Cout, float* raw, int dtype,
                                  void* stream) {
    FAVAE_CHECK_ARG(act && wk && raw && Nb > 0 && H > 0 && W > 0, "favae_conv_out_raw: bad argument");
    FAVAE_CHECK_ARG(Cout >= 1 && Cout <= 4, "favae_conv_out_raw: 1..4 output channels");
    FAVAE_CHECK_ARG(dtype == FAVAE_BF16 && W % 64 == 0, "favae_conv_out_raw: bf16 activations, image width a multiple of 64");
    FAVAE_CHECK_ARG(((uintptr_t)act % 16) == 0 && ((uintptr_t)wk % 16) == 0 && (size_t)Nb * H * W < (1u << 31), "favae_conv_out_raw: alignment / size");
    const int total = Nb * H * W, grid = cdiv(total, 256);
#define CO_ROWS(CO_)                                                                                                                      \
    hipLaunchKernelGGL((conv_out_ll_rows<CO_>), dim3(grid), dim3(256), 0, ST, (const bf16_t*)act, (const bf16_t*)wk, bias, (const float*)nullptr, H, W, \
                       total, 0.f, raw, (bf16_t*)nullptr, (double*)nullptr, (unsigned*)nullptr, 0.0, 1.0, 1.0, (double*)nullptr, (float*)nullptr)
    switch (Cout) { case 1: CO_ROWS(1); break; case 2: CO_ROWS(2); break; case 3: CO_ROWS(3); break; default: CO_ROWS(4); }
#undef CO_ROWS
    FAVAE_LAUNCH_CHECK();
    return FAVAE_OK;
}

static int sigmoid_sse_impl(const float* raw, const void* y, int y_u8, long long n, double* acc3, double* scratch, size_t scratch_doubles,
                            void* stream) {
    FAVAE_CHECK_ARG(raw && y && acc3 && scratch && n > 0, "favae_sigmoid_sse: bad argument");
    const int grid = (int)std::min<size_t>(1024, ((size_t)n + 255) / 256);
    FAVAE_CHECK_ARG((size_t)3 * grid <= scratch_doubles, "favae_sigmoid_sse: scratch too small");
    if (y_u8) hipLaunchKernelGGL(sigmoid_sse<unsigned char>, dim3(grid), dim3(256), 0, ST, raw, (const unsigned char*)y, (size_t)n, scratch);
    else hipLaunchKernelGGL(sigmoid_sse<float>, dim3(grid), dim3(256), 0, ST, raw, (const float*)y, (size_t)n, scratch);
    FAVAE_LAUNCH_CHECK();
    hipLaunchKernelGGL(sse_finish, dim3(1), dim3(64), 0, ST, scratch, grid, acc3);
    FAVAE_LAUNCH_CHECK();
    return FAVAE_OK;
}
extern "C" int favae_sigmoid_sse(const float* raw, const float* y, long long n, double* acc3, double* scratch,
                                 size_t scratch_doubles, void* stream) {
    return sigmoid_sse_impl(raw, y, 0, n, acc3, scratch, scratch_doubles, stream);
}
extern "C" int favae_sigmoid_sse_u8(const float* raw, const unsigned char* y, long long n, double* acc3, double* scratch,
                                    size_t scratch_doubles, void* stream) {
    return sigmoid_sse_impl(raw, y, 1, n, acc3, scratch, scratch_doubles, stream);
}

__global__ void bf16_convert_kernel(const float* __restrict__ x, bf16_t* __restrict__ hw, bf16_t* __restrict__ sw, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        hw[i] = f32_to_bf16(x[i]);
        sw[i] = f32_to_bf16_sw(x[i]);
    }
}
/* Test hook: the fp32 -> bf16 conversion every kernel of the library uses (gfx950's conversion instruction) next to the integer
 * round-to-nearest-even form it replaced. */
extern "C" int favae_debug_bf16_convert(const float* x, void* hw, void* sw, long long n, void* stream) {
    FAVAE_CHECK_ARG(x && hw && sw && n > 0, "favae_debug_bf16_convert: bad argument");
    hipLaunchKernelGGL(bf16_convert_kernel, dim3(ew_grid((size_t)n)), dim3(256), 0, ST, x, (bf16_t*)hw, (bf16_t*)sw, (size_t)n);
    FAVAE_LAUNCH_CHECK();
    return FAVAE_OK;
}

extern "C" int favae_sigmoid(const float* raw, float* out, long long n, void* stream) {
    FAVAE_CHECK_ARG(raw && out && n > 0, "favae_sigmoid: bad argument");
    hipLaunchKernelGGL(sigmoid_only, dim3(ew_grid((size_t)n)), dim3(256), 0, ST, raw, out, (size_t)n);
    FAVAE_LAUNCH_CHECK();
    return FAVAE_OK;
}

extern "C" int favae_adam_step(float* p, const float* g, float* m, float* v, long long n, float lr, float beta1, float beta2,
                               float eps, long long step, void* stream) {
    FAVAE_CHECK_ARG(p && g && m && v && n > 0 && n % 4 == 0 && step >= 1, "favae_adam_step: bad argument (n must be a multiple of 4)");
    const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    const float step_size = (float)((double)lr / bc1), bc2_sqrt = (float)sqrt(bc2);
    hipLaunchKernelGGL(adam_kernel<false>, dim3(ew_grid((size_t)n / 4)), dim3(256), 0, ST, p, (float*)g, m, v, (size_t)n / 4, beta1, beta2, eps, step_size, bc2_sqrt);
    FAVAE_LAUNCH_CHECK();
    return FAVAE_OK;
}

extern "C" int favae_adam_step_clear(float* p, float* g, float* m, float* v, long long n, float lr, float beta1, float beta2,
                                     float eps, long long step, void* stream) {
    FAVAE_CHECK_ARG(p && g && m && v && n > 0 && n % 4 == 0 && step >= 1, "favae_adam_step_clear: bad argument (n must be a multiple of 4)");
    const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    const float step_size = (float)((double)lr / bc1), bc2_sqrt = (float)sqrt(bc2);
    ProfScope prof(13, 32.0 * (double)n, ST);   // p, g, m, v read; p, m, v written; g cleared
    hipLaunchKernelGGL(adam_kernel<true>, dim3(adam_grid((size_t)n / 4)), dim3(256), 0, ST, p, g, m, v, (size_t)n / 4, beta1, beta2, eps, step_size, bc2_sqrt);
    FAVAE_LAUNCH_CHECK();
    return FAVAE_OK;
}

extern "C" int favae_adam_step_clear_scaled(float* p, float* g, float* m, float* v, long long n, float lr, float beta1, float beta2,
                                            float eps, long long step, float gscale, void* stream) {
    FAVAE_CHECK_ARG(p && g && m && v && n > 0 && n % 4 == 0 && step >= 1, "favae_adam_step_clear_scaled: bad argument (n must be a multiple of 4)");
    const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    const float step_size = (float)((double)lr / bc1), bc2_sqrt = (float)sqrt(bc2);
    ProfScope prof(13, 32.0 * (double)n, ST);
    hipLaunchKernelGGL(adam_kernel<true>, dim3(adam_grid((size_t)n / 4)), dim3(256), 0, ST, p, g, m, v, (size_t)n / 4, beta1, beta2, eps, step_size, bc2_sqrt,
                       (bf16_t*)nullptr, (size_t)0, gscale);
    FAVAE_LAUNCH_CHECK();
    return FAVAE_OK;
}

extern "C" int favae_adam_step_clear_from(float* p, float* g, float* m, float* v, long long n, long long keep, float lr, float beta1,
                                          float beta2, float eps, long long step, void* stream) {
    FAVAE_CHECK_ARG(p && g && m && v && n > 0 && n % 4 == 0 && step >= 1, "favae_adam_step_clear_from: bad argument (n must be a multiple of 4)");
    FAVAE_CHECK_ARG(keep >= 0 && keep <= n && keep % 4 == 0, "favae_adam_step_clear_from: keep (a multiple of 4, <= n)");
    const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    const float step_size = (float)((double)lr / bc1), bc2_sqrt = (float)sqrt(bc2);
    ProfScope prof(13, 32.0 * (double)n - 4.0 * (double)keep, ST);
    hipLaunchKernelGGL(adam_kernel<true>, dim3(adam_grid((size_t)n / 4)), dim3(256), 0, ST, p, g, m, v, (size_t)n / 4, beta1, beta2, eps, step_size, bc2_sqrt,
                       (bf16_t*)nullptr, (size_t)0, 1.f, (size_t)keep / 4);
    FAVAE_LAUNCH_CHECK();
    return FAVAE_OK;
}

/* favae_adam_step_clear_scaled that leaves g[keep_lo[k] .. keep_hi[k]) as they are, k = 0, 1 (a bucket of the data-parallel epoch:
 * the weight-gradient slices that the next step's unsplit GEMMs overwrite; the first range must start at 0, either may be empty). */
extern "C" int favae_adam_step_clear_scaled_keep(float* p, float* g, float* m, float* v, long long n, float lr, float beta1, float beta2,
                                                 float eps, long long step, float gscale, long long keep0_hi, long long keep1_lo,
                                                 long long keep1_hi, void* stream) {
    FAVAE_CHECK_ARG(p && g && m && v && n > 0 && n % 4 == 0 && step >= 1, "favae_adam_step_clear_scaled_keep: bad argument (n must be a multiple of 4)");
    FAVAE_CHECK_ARG(keep0_hi >= 0 && keep0_hi <= n && keep0_hi % 4 == 0 && keep1_lo >= 0 && keep1_lo <= keep1_hi && keep1_hi <= n && keep1_lo % 4 == 0 &&
                        keep1_hi % 4 == 0, "favae_adam_step_clear_scaled_keep: ranges (multiples of 4 inside [0, n])");
    const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    const float step_size = (float)((double)lr / bc1), bc2_sqrt = (float)sqrt(bc2);
    ProfScope prof(13, 32.0 * (double)n - 4.0 * (double)(keep0_hi + keep1_hi - keep1_lo), ST);
    hipLaunchKernelGGL(adam_kernel<true>, dim3(adam_grid((size_t)n / 4)), dim3(256), 0, ST, p, g, m, v, (size_t)n / 4, beta1, beta2, eps, step_size, bc2_sqrt,
                       (bf16_t*)nullptr, (size_t)0, gscale, (size_t)keep0_hi / 4, (size_t)keep1_lo / 4, (size_t)keep1_hi / 4);
    FAVAE_LAUNCH_CHECK();
    return FAVAE_OK;
}

extern "C" int favae_adam_step_copy(float* p, float* g, float* m, float* v, long long n, float lr, float beta1, float beta2,
                                    float eps, long long step, int clear_g, void* copy_bf16, long long ncopy, void* stream) {
    FAVAE_CHECK_ARG(p && g && m && v && copy_bf16 && n > 0 && n % 4 == 0 && step >= 1, "favae_adam_step_copy: bad argument (n must be a multiple of 4)");
    FAVAE_CHECK_ARG(ncopy > 0 && ncopy <= n && ncopy % 4 == 0 && ((uintptr_t)copy_bf16 % 8) == 0, "favae_adam_step_copy: ncopy (multiple of 4, <= n), copy 8-byte aligned");
    const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    const float step_size = (float)((double)lr / bc1), bc2_sqrt = (float)sqrt(bc2);
    if (clear_g)
        hipLaunchKernelGGL((adam_kernel<true, true>), dim3(adam_grid((size_t)n / 4)), dim3(256), 0, ST, p, g, m, v, (size_t)n / 4, beta1, beta2, eps,
                           step_size, bc2_sqrt, (bf16_t*)copy_bf16, (size_t)ncopy / 4);
    else
        hipLaunchKernelGGL((adam_kernel<false, true>), dim3(adam_grid((size_t)n / 4)), dim3(256), 0, ST, p, g, m, v, (size_t)n / 4, beta1, beta2, eps,
                           step_size, bc2_sqrt, (bf16_t*)copy_bf16, (size_t)ncopy / 4);
    FAVAE_LAUNCH_CHECK();
    return FAVAE_OK;
}

extern "C" int favae_adam_wprep(float* p, float* g, float* m, float* v, long long n, int Co, int taps, int Ci, void* w_copy,
                                void* w_transposed, int clear_grad, float lr, float beta1, float beta2, float eps, long long step,
                                int dtype, void* stream) {
    FAVAE_CHECK_ARG(p && g && m && v && step >= 1 && Co > 0 && taps > 0 && Ci > 0, "favae_adam_wprep: bad argument");
    FAVAE_CHECK_ARG(Ci % 4 == 0 && Co % 4 == 0 && n % 4 == 0, "favae_adam_wprep: Ci, Co and n must be multiples of 4");
    FAVAE_CHECK_ARG(dtype == FAVAE_F32 || dtype == FAVAE_BF16, "favae_adam_wprep: dtype");
    const long long nw = (long long)Co * taps * Ci;
    FAVAE_CHECK_ARG(n >= nw, "favae_adam_wprep: slice shorter than the weight tensor");
    FAVAE_CHECK_ARG(((uintptr_t)p % 16) == 0 && ((uintptr_t)g % 16) == 0 && ((uintptr_t)m % 16) == 0 && ((uintptr_t)v % 16) == 0 &&
                    ((uintptr_t)w_copy % 16) == 0 && ((uintptr_t)w_transposed % 16) == 0, "favae_adam_wprep: 16-byte alignment");
    const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    const float step_size = (float)((double)lr / bc1), bc2_sqrt = (float)sqrt(bc2);
    const int ntiles = cdiv(Ci, 64) * cdiv(Co, 64) * taps;
    const size_t tail_lo4 = (size_t)nw / 4, tail_n4 = (size_t)(n - nw) / 4;
    const int tail_blocks = tail_n4 ? (int)std::min<size_t>(64, (tail_n4 + 255) / 256) : 0;
#define AW_LAUNCH(T_, CG_)                                                                                                   \
    hipLaunchKernelGGL((adam_wprep_kernel<T_, CG_>), dim3(ntiles + tail_blocks), dim3(256), 0, ST, p, g, m, v, Co, taps, Ci,  \
                       (T_*)w_copy, (T_*)w_transposed, ntiles, tail_lo4, tail_n4, beta1, beta2, eps, step_size, bc2_sqrt)
    if (dtype == FAVAE_F32) { if (clear_grad) AW_LAUNCH(float, true); else AW_LAUNCH(float, false); }
    else { if (clear_grad) AW_LAUNCH(bf16_t, true); else AW_LAUNCH(bf16_t, false); }
#undef AW_LAUNCH
    FAVAE_LAUNCH_CHECK();
    return FAVAE_OK;
}

extern "C" int favae_weight_prep(const float* w, void* w_copy, void* w_transposed, int Co, int taps, int Ci, int dtype,
                                 void* stream) {
    FAVAE_CHECK_ARG(w && Co > 0 && taps > 0 && Ci > 0 && (w_copy || w_transposed), "favae_weight_prep: bad argument");
    static const bool tile32 = getenv("FAVAE_WPREP64") && getenv("FAVAE_WPREP64")[0] == '0';   // (A/B: the 32x32 form everywhere)
    if (Ci % 4 == 0 && Co % 4 == 0 && !tile32 && ((uintptr_t)w % 16) == 0 && ((uintptr_t)w_copy % 8) == 0 && ((uintptr_t)w_transposed % 8) == 0) {
        const unsigned nb = (unsigned)(cdiv(Ci, 64) * cdiv(Co, 64) * taps);
        if (dtype == FAVAE_F32)
            hipLaunchKernelGGL(wprep64_kernel<float>, dim3(nb), dim3(256), 0, ST, w, (float*)w_copy, (float*)w_transposed, Co, taps, Ci);
        else
            hipLaunchKernelGGL(wprep64_kernel<bf16_t>, dim3(nb), dim3(256), 0, ST, w, (bf16_t*)w_copy, (bf16_t*)w_transposed, Co, taps, Ci);
        FAVAE_LAUNCH_CHECK();
        return FAVAE_OK;
    }
    dim3 grid(cdiv(Ci, 32), cdiv(Co, 32), taps);
    if (dtype == FAVAE_F32)
        hipLaunchKernelGGL(wprep_kernel<float>, grid, dim3(256), 0, ST, w, (float*)w_copy, (float*)w_transposed, Co, taps, Ci);
    else
        hipLaunchKernelGGL(wprep_kernel<bf16_t>, grid, dim3(256), 0, ST, w, (bf16_t*)w_copy, (bf16_t*)w_transposed, Co, taps, Ci);
    FAVAE_LAUNCH_CHECK();
    return FAVAE_OK;
}

extern "C" int favae_weight_prep_batch(int n, const float* const* w, void* const* w_copy, void* const* w_transposed,
                                       const int* Co, const int* taps, const int* Ci, int dtype, void* stream) {
    FAVAE_CHECK_ARG(n >= 1 && n <= 16 && w && w_copy && w_transposed && Co && taps && Ci, "favae_weight_prep_batch: bad argument");
    WprepTable tb;
    int total = 0;
    for (int i = 0; i < n; ++i) {
        FAVAE_CHECK_ARG(w[i] && Co[i] > 0 && taps[i] > 0 && Ci[i] > 0 && (w_copy[i] || w_transposed[i]), "favae_weight_prep_batch: tensor %d", i);
        tb.w[i] = w[i]; tb.wc[i] = w_copy[i]; tb.wt[i] = w_transposed[i];
        tb.Co[i] = Co[i]; tb.taps[i] = taps[i]; tb.Ci[i] = Ci[i];
        tb.start[i] = total;
        static const bool tile32 = getenv("FAVAE_WPREP64") && getenv("FAVAE_WPREP64")[0] == '0';   // (A/B: the 32x32 form everywhere)
        tb.t64[i] = !tile32 && Ci[i] % 4 == 0 && Co[i] % 4 == 0 && ((uintptr_t)w[i] % 16) == 0 && ((uintptr_t)w_copy[i] % 8) == 0 &&
                    ((uintptr_t)w_transposed[i] % 8) == 0;
        total += tb.t64[i] ? cdiv(Ci[i], 64) * cdiv(Co[i], 64) * taps[i] : cdiv(Ci[i], 32) * cdiv(Co[i], 32) * taps[i];
    }
    tb.start[n] = total;
    if (dtype == FAVAE_F32)
        hipLaunchKernelGGL(wprep_batch_kernel<float>, dim3(total), dim3(256), 0, ST, tb, n);
    else
        hipLaunchKernelGGL(wprep_batch_kernel<bf16_t>, dim3(total), dim3(256), 0, ST, tb, n);
    FAVAE_LAUNCH_CHECK();
    return FAVAE_OK;
}

extern "C" int favae_gather_batch(const float* data, const int* idx, int B, long long row_floats, float* x_f32, void* x_t,
                                  int dtype, void* stream) {
    FAVAE_CHECK_ARG(data && B > 0 && row_floats > 0 && row_floats % 4 == 0 && (x_f32 || x_t), "favae_gather_batch: bad argument");
    const size_t row4 = (size_t)row_floats / 4;
    const int grid = ew_grid((size_t)B * row4);
    if (dtype == FAVAE_F32)
        hipLaunchKernelGGL(gather_batch<float>, dim3(grid), dim3(256), 0, ST, data, idx, B, row4, x_f32, (float*)x_t);
    else
        hipLaunchKernelGGL(gather_batch<bf16_t>, dim3(grid), dim3(256), 0, ST, data, idx, B, row4, x_f32, (bf16_t*)x_t);
    FAVAE_LAUNCH_CHECK();
    return FAVAE_OK;
}

extern "C" int favae_gather_batch_u8(const unsigned char* data, const int* idx, int B, long long row_elems, float* x_f32, void* x_t,
                                     int dtype, void* stream) {
    FAVAE_CHECK_ARG(data && B > 0 && row_elems > 0 && row_elems % 4 == 0 && (x_f32 || x_t), "favae_gather_batch_u8: bad argument");
    const size_t row4 = (size_t)row_elems / 4;
    const int grid = ew_grid((size_t)B * row4);
    if (dtype == FAVAE_F32)
        hipLaunchKernelGGL(gather_batch_u8<float>, dim3(grid), dim3(256), 0, ST, data, idx, B, row4, x_f32, (float*)x_t);
    else
        hipLaunchKernelGGL(gather_batch_u8<bf16_t>, dim3(grid), dim3(256), 0, ST, data, idx, B, row4, x_f32, (bf16_t*)x_t);
    FAVAE_LAUNCH_CHECK();
    return FAVAE_OK;
}

extern "C" int favae_nchw_to_nhwc_u8(const float* in, unsigned char* out, long long N, int C, int HW, unsigned long long* mismatches,
                                     void* stream) {
    FAVAE_CHECK_ARG(in && out && mismatches && N > 0 && C > 0 && HW > 0, "favae_nchw_to_nhwc_u8: bad argument");
    hipLaunchKernelGGL(nchw_to_nhwc_u8, dim3(ew_grid((size_t)N * HW * C)), dim3(256), 0, ST, in, out, (size_t)N, C, HW, mismatches);
    FAVAE_LAUNCH_CHECK();
    return FAVAE_OK;
}

extern "C" int favae_cast(const float* in, void* out, long long n, int dtype, void* stream) {
    FAVAE_CHECK_ARG(in && out && n > 0, "favae_cast: bad argument");
    if (dtype == FAVAE_F32)
        hipLaunchKernelGGL(cast_kernel<float>, dim3(ew_grid((size_t)n)), dim3(256), 0, ST, in, (float*)out, (size_t)n);
    else
        hipLaunchKernelGGL(cast_kernel<bf16_t>, dim3(ew_grid((size_t)n)), dim3(256), 0, ST, in, (bf16_t*)out, (size_t)n);
    FAVAE_LAUNCH_CHECK();
    return FAVAE_OK;
}

extern "C" int favae_nchw_to_nhwc(const float* in, float* out, long long N, int C, int HW, int Cpad, void* stream) {
    FAVAE_CHECK_ARG(in && out && N > 0 && C > 0 && HW > 0 && Cpad >= C, "favae_nchw_to_nhwc: bad argument");
    hipLaunchKernelGGL(nchw_to_nhwc, dim3(ew_grid((size_t)N * HW * Cpad)), dim3(256), 0, ST, in, out, (size_t)N, C, HW, Cpad);
    FAVAE_LAUNCH_CHECK();
    return FAVAE_OK;
}

extern "C" int favae_nhwc_to_nchw_f64(const float* in, double* out, long long N, int C, int HW, int Cpad, void* stream) {
    FAVAE_CHECK_ARG(in && out && N > 0 && C > 0 && HW > 0 && Cpad >= C, "favae_nhwc_to_nchw_f64: bad argument");
    hipLaunchKernelGGL(nhwc_to_nchw_f64, dim3(ew_grid((size_t)N * C * HW)), dim3(256), 0, ST, in, out, (size_t)N, C, HW, Cpad);
    FAVAE_LAUNCH_CHECK();
    return FAVAE_OK;
}
