// BatchNorm2d (train) + LeakyReLU over a batch made of G groups of 64 images, each group with batch statistics of its own:
// the "grouped-batch" throughput mode of the conv VAE ("Hard parts": larger batches change the trajectory, so the
// BatchNorm group size stays at the reference's 64, celeba_deepvanilla_vae.py:218 with nn.BatchNorm2d :34,58,72).  One step over
// G x 64 images with the mean loss is, gradient for gradient, the G-rank data-parallel step of DESIGN.md 7 (every rank a batch of
// 64, gradients averaged) on one GPU; the GEMMs see G times the rows, the launch count per sample falls G-fold.
//
// All tensors are NHWC [G * Rg][C] (Rg rows per group), the per-channel sums live in bins [G][nbin][2][C] (fp64, filled by the
// epilogue of the producing GEMM, igemm.hip red_group_rows, or by gbn_sums_kernel), mean / invstd are [G][C].
// These kernels are used with G > 1 only; the batch-64 step keeps its own tuned BatchNorm kernels (vae_kernels.hip).
#include <algorithm>
#include "common.h"
#include "favae_hip.h"

namespace {

__device__ __forceinline__ float lrelu_g(float y) { return y > 0.f ? y : 0.01f * y; }

template <class T> __device__ __forceinline__ float4 ld4(const T* p);
template <> __device__ __forceinline__ float4 ld4<float>(const float* p) { return *(const float4*)p; }
template <> __device__ __forceinline__ float4 ld4<bf16_t>(const bf16_t* p) {
    const uint2 u = *(const uint2*)p;
    return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u));
}
template <class T> __device__ __forceinline__ void st4(T* p, float a, float b, float c, float d);
template <> __device__ __forceinline__ void st4<float>(float* p, float a, float b, float c, float d) { *(float4*)p = make_float4(a, b, c, d); }
template <> __device__ __forceinline__ void st4<bf16_t>(bf16_t* p, float a, float b, float c, float d) {
    uint2 u;
    u.x = pack_bf16x2(a, b);
    u.y = pack_bf16x2(c, d);
    *(uint2*)p = u;
}

// ---- sums of a tensor whose producer could not take them in its epilogue ------------------------------------------------
// MODE 1: (sum y, sum y^2) of y = raw;  MODE 2: (sum dd, sum dd * xhat) with dd = y * lrelu'(act), xhat = (raw - mean) * invstd, y = dact.
// grid (C / 64, slabs, G), 256 threads = 16 channel quads x 16 row lanes; 128 fp64 atomics per block into bins[g][slab % nbin].
// (T: activation storage; YT: storage of y; RT: storage of raw)
template <class T, class YT, class RT, int MODE>
__global__ __launch_bounds__(256) void gbn_sums_kernel(const YT* __restrict__ y, const T* __restrict__ act, const RT* __restrict__ raw,
                                                       const float* __restrict__ mean, const float* __restrict__ invstd,
                                                       const float* __restrict__ ab, double* __restrict__ bins, int nbin, int Rg, int C, int rps) {
    __shared__ double sm[2][16][64];
    const int tid = threadIdx.x, cq = tid & 15, rl = tid >> 4, g = blockIdx.z;
    const int c0 = blockIdx.x * 64 + cq * 4;
    const int r0 = blockIdx.y * rps, r1 = min(Rg, r0 + rps);
    double s[4] = {0, 0, 0, 0}, q[4] = {0, 0, 0, 0};
    float4 mu = make_float4(0.f, 0.f, 0.f, 0.f), is = mu;
    float4 fa = mu, fb = mu;   // ab != NULL: the forward pass's affine form, the LeakyReLU slope follows the sign of raw * a + b (act is not read)
    if (MODE == 2) { mu = *(const float4*)(mean + (size_t)g * C + c0); is = *(const float4*)(invstd + (size_t)g * C + c0); }
    if (MODE == 2 && ab) { fa = *(const float4*)(ab + ((size_t)g * 2) * C + c0); fb = *(const float4*)(ab + ((size_t)g * 2 + 1) * C + c0); }
    for (int r = r0 + rl; r < r1; r += 16) {
        const size_t e = ((size_t)g * Rg + r) * C + c0;
        const float4 v = ld4<YT>(y + e);
        if (MODE == 1) {
            s[0] += v.x; s[1] += v.y; s[2] += v.z; s[3] += v.w;
            q[0] += (double)v.x * v.x; q[1] += (double)v.y * v.y; q[2] += (double)v.z * v.z; q[3] += (double)v.w * v.w;
        } else {
            const float4 x = ld4<RT>(raw + e);
            const float4 a = ab ? make_float4(x.x * fa.x + fb.x, x.y * fa.y + fb.y, x.z * fa.z + fb.z, x.w * fa.w + fb.w) : ld4<T>(act + e);
            const float d0 = v.x * (a.x > 0.f ? 1.f : 0.01f), d1 = v.y * (a.y > 0.f ? 1.f : 0.01f);
            const float d2 = v.z * (a.z > 0.f ? 1.f : 0.01f), d3 = v.w * (a.w > 0.f ? 1.f : 0.01f);
            s[0] += d0; s[1] += d1; s[2] += d2; s[3] += d3;
            q[0] += (double)d0 * ((x.x - mu.x) * is.x); q[1] += (double)d1 * ((x.y - mu.y) * is.y);
            q[2] += (double)d2 * ((x.z - mu.z) * is.z); q[3] += (double)d3 * ((x.w - mu.w) * is.w);
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) { sm[0][rl][cq * 4 + j] = s[j]; sm[1][rl][cq * 4 + j] = q[j]; }
    __syncthreads();
    if (tid < 128) {
        const int qq = tid >> 6, col = tid & 63;
        double t = 0.0;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += sm[qq][k][col];
        unsafeAtomicAdd(bins + (((size_t)g * nbin + blockIdx.y % (unsigned)nbin) * 2 + qq) * C + blockIdx.x * 64 + col, t);
    }
}

// totals of the group's bins for channel c (the threads of a block take consecutive channels: coalesced over c)
__device__ __forceinline__ void group_totals(const double* __restrict__ bins, int nbin, int C, int g, int c, double& t0, double& t1) {
    const double* b = bins + (size_t)g * nbin * 2 * C;
    double a = 0.0, q = 0.0;
    for (int k = 0; k < nbin; ++k) { a += b[((size_t)k * 2) * C + c]; q += b[((size_t)k * 2 + 1) * C + c]; }
    t0 = a; t1 = q;
}

// ---- forward: act = lrelu((raw - mean_g) * invstd_g * gamma + beta); block 0 of a group publishes mean / invstd of the group -----
// grid (blocks per group, G); the element stride of a block round is 1024, a multiple of C, so a thread keeps its 4 channels.
template <class T, class RT>
__global__ __launch_bounds__(256) void gbn_fwd_kernel(const RT* __restrict__ raw, const double* __restrict__ bins, int nbin,
                                                      const float* __restrict__ gamma, const float* __restrict__ beta, T* __restrict__ act,
                                                      float* __restrict__ mean, float* __restrict__ invstd, float* __restrict__ ab_out, int Rg, int C,
                                                      float eps) {
    // (8 KB of LDS: the 2048 blocks of a G = 8 launch are resident at once; with the fp64 totals staged in LDS too, 24 KB, only
    // 1536 were and the launch ran a second, third-full round)
    __shared__ float ab[2048];
    const int g = blockIdx.y, tid = threadIdx.x;
    for (int c = tid; c < C; c += 256) {
        double t0, t1;
        group_totals(bins, nbin, C, g, c, t0, t1);
        const double m = t0 / (double)Rg;
        double var = t1 / (double)Rg - m * m;
        if (var < 0) var = 0;
        const float is = (float)(1.0 / sqrt(var + (double)eps));
        const float a = is * gamma[c];
        ab[c] = a;
        ab[C + c] = beta[c] - (float)m * a;
        if (blockIdx.x == 0) {
            mean[(size_t)g * C + c] = (float)m; invstd[(size_t)g * C + c] = is;
            if (ab_out) { ab_out[((size_t)g * 2) * C + c] = a; ab_out[((size_t)g * 2 + 1) * C + c] = beta[c] - (float)m * a; }
        }
    }
    __syncthreads();
    if (!act) return;   // uniform: the statistics only (a consumer applies a, b itself)
    const size_t nvec = (size_t)Rg * C / 4, base = (size_t)g * Rg * C;
    const int c = (tid * 4) % C;
    const float a0 = ab[c], a1 = ab[c + 1], a2 = ab[c + 2], a3 = ab[c + 3];
    const float b0 = ab[C + c], b1 = ab[C + c + 1], b2 = ab[C + c + 2], b3 = ab[C + c + 3];
#pragma unroll 4
    for (size_t i = (size_t)blockIdx.x * 256 + tid; i < nvec; i += (size_t)gridDim.x * 256) {
        const float4 x = ld4<RT>(raw + base + i * 4);
        st4<T>(act + base + i * 4, lrelu_g(x.x * a0 + b0), lrelu_g(x.y * a1 + b1), lrelu_g(x.z * a2 + b2), lrelu_g(x.w * a3 + b3));
    }
}

// running statistics from the G groups: what averaging the ranks' running statistics gives in the data-parallel run
// (run = (1 - mom) run + mom * mean_g(stat_g): the update is linear in the batch statistic)
__global__ void gbn_running_kernel(const float* __restrict__ mean, const float* __restrict__ invstd, int G, int C, int Rg, float eps, float mom,
                                   float* __restrict__ run_mean, float* __restrict__ run_var) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    double m = 0.0, v = 0.0;
    for (int g = 0; g < G; ++g) {
        m += mean[(size_t)g * C + c];
        const double is = invstd[(size_t)g * C + c];
        double var = 1.0 / (is * is) - (double)eps;
        if (var < 0) var = 0;
        v += Rg > 1 ? var * Rg / (Rg - 1.0) : var;
    }
    run_mean[c] = (1.f - mom) * run_mean[c] + mom * (float)(m / G);
    run_var[c] = (1.f - mom) * run_var[c] + mom * (float)(v / G);
}

// ---- backward: draw = gamma invstd_g (dd - s1_g / Rg - xhat s2_g / Rg); dgamma += s2_g, dbeta += s1_g (block 0 of a group),
// dbias += column sums of draw (the bias of the convolution in front of the BatchNorm: analytically zero, rounding noise) ----------
template <class T, class DT, class RT>
__global__ __launch_bounds__(256) void gbn_bwd_kernel(const DT* __restrict__ dact, const T* __restrict__ act, const RT* __restrict__ raw,
                                                      const float* __restrict__ mean, const float* __restrict__ invstd,
                                                      const float* __restrict__ gamma, const float* __restrict__ ab, const double* __restrict__ bins,
                                                      int nbin, T* __restrict__ draw, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                      float* __restrict__ dbias, int Rg, int C) {
    __shared__ float kk[2048];
    __shared__ float cs[1024];
    const int g = blockIdx.y, tid = threadIdx.x;
    const float invR = (float)(1.0 / (double)Rg);
    for (int c = tid; c < C; c += 256) {
        double t0, t1;
        group_totals(bins, nbin, C, g, c, t0, t1);
        if (blockIdx.x == 0) { atomicAdd(dbeta + c, (float)t0); atomicAdd(dgamma + c, (float)t1); }
        kk[c] = (float)t0 * invR; kk[C + c] = (float)t1 * invR;
        cs[c] = 0.f;
    }
    __syncthreads();
    const int c = (tid * 4) % C;
    float mu[4], is[4], gi[4], k1[4], k2[4], fa[4], fb[4], acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        mu[j] = mean[(size_t)g * C + c + j]; is[j] = invstd[(size_t)g * C + c + j]; gi[j] = gamma[c + j] * is[j];
        fa[j] = ab ? ab[((size_t)g * 2) * C + c + j] : 0.f; fb[j] = ab ? ab[((size_t)g * 2 + 1) * C + c + j] : 0.f;
        k1[j] = kk[c + j]; k2[j] = kk[C + c + j];
    }
    const size_t nvec = (size_t)Rg * C / 4, base = (size_t)g * Rg * C;
#pragma unroll 2
    for (size_t i = (size_t)blockIdx.x * 256 + tid; i < nvec; i += (size_t)gridDim.x * 256) {
        const size_t e = base + i * 4;
        const float4 d = ld4<DT>(dact + e), x = ld4<RT>(raw + e);
        float4 a = make_float4(x.x * fa[0] + fb[0], x.y * fa[1] + fb[1], x.z * fa[2] + fb[2], x.w * fa[3] + fb[3]);   // (the forward pass's expression)
        if (!ab) a = ld4<T>(act + e);   // uniform
        const float dv[4] = {d.x, d.y, d.z, d.w}, av[4] = {a.x, a.y, a.z, a.w}, xv[4] = {x.x, x.y, x.z, x.w};
        float o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float dd = dv[j] * (av[j] > 0.f ? 1.f : 0.01f);
            const float xh = (xv[j] - mu[j]) * is[j];
            o[j] = gi[j] * (dd - k1[j] - xh * k2[j]);
            acc[j] += o[j];
        }
        st4<T>(draw + e, o[0], o[1], o[2], o[3]);
    }
    if (dbias) {   // uniform
#pragma unroll
        for (int j = 0; j < 4; ++j) atomicAdd(&cs[c + j], acc[j]);
        __syncthreads();
        for (int k = tid; k < C; k += 256) atomicAdd(dbias + k, cs[k]);
    }
}

inline int gbn_grid(long long Rg, int C) {
    const long long nvec = Rg * C / 4;
    return (int)std::max<long long>(1, std::min<long long>(256, (nvec + 1023) / 1024));
}

}  // namespace

#define ST ((hipStream_t)stream)
#define GB_BASE(d) ((d) & 0xF)
#define GB_RAWBF(d) (((d) & FAVAE_RAW_BF16) != 0)
#define GB_DACTBF(d) (((d) & FAVAE_DACT_BF16) != 0)

/* Sums of a [G * Rg][C] tensor into bins [G][nbins][2][C] (the caller zeroes them): mode 1 = (sum, sum of squares) of y = raw (BatchNorm
 * statistics), mode 2 = the two BatchNorm-backward sums of y = dact given act, raw, mean [G][C], invstd [G][C].  For tensors whose
 * producing GEMM split its reduction (no epilogue sums).  C a multiple of 64.  raw / dact are fp32, or bf16 with FAVAE_RAW_BF16 /
 * FAVAE_DACT_BF16 in dtype (FAVAE_BF16 only). */
extern "C" int favae_gbn_sums(const void* y, const void* act, const void* raw, const float* mean, const float* invstd, const float* ab,
                              double* bins, int nbins, int G, long long Rg, int C, int mode, int dtype, void* stream) {
    FAVAE_CHECK_ARG(y && bins && nbins >= 1 && G >= 1 && Rg >= 1 && Rg < (1LL << 30) && C >= 64 && C % 64 == 0, "favae_gbn_sums: bad argument");
    FAVAE_CHECK_ARG(mode == 1 || (mode == 2 && (act || ab) && raw && mean && invstd), "favae_gbn_sums: mode 1, or mode 2 with act (or ab) / raw / mean / invstd");
    FAVAE_CHECK_ARG(GB_BASE(dtype) == FAVAE_F32 || GB_BASE(dtype) == FAVAE_BF16, "favae_gbn_sums: dtype");
    FAVAE_CHECK_ARG(!((GB_RAWBF(dtype) || GB_DACTBF(dtype)) && GB_BASE(dtype) != FAVAE_BF16), "favae_gbn_sums: bf16 raw / dact need FAVAE_BF16");
    const int slabs = (int)std::max<long long>(1, std::min<long long>(std::max(1, 512 / (G * (C / 64))), (Rg + 63) / 64));
    const int rps = (int)((Rg + slabs - 1) / slabs);
    dim3 grid(C / 64, slabs, G);
    const bool rb = GB_RAWBF(dtype), db = GB_DACTBF(dtype);
#define GS_LAUNCH(T, YT, RT, MODE) hipLaunchKernelGGL((gbn_sums_kernel<T, YT, RT, MODE>), grid, dim3(256), 0, ST, (const YT*)y, (const T*)act, (const RT*)raw, \
                                                      mean, invstd, ab, bins, nbins, (int)Rg, C, rps)
    if (mode == 1) {
        if (rb) GS_LAUNCH(bf16_t, bf16_t, bf16_t, 1); else GS_LAUNCH(float, float, float, 1);
    } else if (GB_BASE(dtype) == FAVAE_F32) {
        GS_LAUNCH(float, float, float, 2);
    } else if (db) {
        if (rb) GS_LAUNCH(bf16_t, bf16_t, bf16_t, 2); else GS_LAUNCH(bf16_t, bf16_t, float, 2);
    } else {
        if (rb) GS_LAUNCH(bf16_t, float, bf16_t, 2); else GS_LAUNCH(bf16_t, float, float, 2);
    }
#undef GS_LAUNCH
    FAVAE_LAUNCH_CHECK();
    return FAVAE_OK;
}

/* BatchNorm2d(train) + LeakyReLU forward of G groups from bins [G][nbins][2][C]: act (dtype) and mean / invstd [G][C]; the running
 * statistics take the average over the groups of the batch statistics (what averaging the ranks' running statistics gives in
 * the data-parallel run).  C <= 1024, a multiple of 4 that divides 1024.  raw fp32, or bf16 with FAVAE_RAW_BF16.  act == NULL (with ab): the
 * statistics only, for a consumer that applies (a | b) at its operand load (favae_conv_out_ll_bn). */
extern "C" int favae_gbn_fwd(const void* raw, const double* bins, int nbins, const float* gamma, const float* beta, void* act, float* mean,
                             float* invstd, float* ab, float* run_mean, float* run_var, int G, long long Rg, int C, float eps, float momentum,
                             int dtype, void* stream) {
    FAVAE_CHECK_ARG(raw && bins && gamma && beta && (act || ab) && mean && invstd && nbins >= 1 && G >= 1 && Rg >= 1 && Rg < (1LL << 30), "favae_gbn_fwd: bad argument");
    FAVAE_CHECK_ARG(C >= 4 && C <= 1024 && 1024 % C == 0, "favae_gbn_fwd: C = %d must divide 1024", C);
    FAVAE_CHECK_ARG(GB_BASE(dtype) == FAVAE_F32 || GB_BASE(dtype) == FAVAE_BF16, "favae_gbn_fwd: dtype");
    FAVAE_CHECK_ARG(!(GB_RAWBF(dtype) && GB_BASE(dtype) != FAVAE_BF16), "favae_gbn_fwd: FAVAE_RAW_BF16 needs FAVAE_BF16");
    dim3 grid(act ? gbn_grid(Rg, C) : 1, G);   // act == NULL: mean / invstd / ab / running statistics only
#define GF_LAUNCH(T, RT) hipLaunchKernelGGL((gbn_fwd_kernel<T, RT>), grid, dim3(256), 0, ST, (const RT*)raw, bins, nbins, gamma, beta, (T*)act, mean, invstd, ab, (int)Rg, C, eps)
    if (GB_BASE(dtype) == FAVAE_F32) GF_LAUNCH(float, float);
    else if (GB_RAWBF(dtype)) GF_LAUNCH(bf16_t, bf16_t);
    else GF_LAUNCH(bf16_t, float);
#undef GF_LAUNCH
    FAVAE_LAUNCH_CHECK();
    if (run_mean && run_var) {
        hipLaunchKernelGGL(gbn_running_kernel, dim3((C + 255) / 256), dim3(256), 0, ST, mean, invstd, G, C, (int)Rg, eps, momentum, run_mean, run_var);
        FAVAE_LAUNCH_CHECK();
    }
    return FAVAE_OK;
}

/* The running statistics alone, from mean / invstd [G][C] as favae_gbn_fwd left them (favae_gbn_fwd with run_mean == NULL skips them): what
 * nn.BatchNorm2d(train) does to running_mean / running_var (celeba_deepvanilla_vae.py:34,58,72), averaged over the groups.  Nothing in a
 * training step reads them, so the executor takes this launch off the main chain (the stream of the layer's weight gradient). */
extern "C" int favae_gbn_running(const float* mean, const float* invstd, int G, long long Rg, int C, float eps, float momentum, float* run_mean,
                                 float* run_var, void* stream) {
    FAVAE_CHECK_ARG(mean && invstd && run_mean && run_var && G >= 1 && Rg >= 1 && C >= 1, "favae_gbn_running: bad argument");
    hipLaunchKernelGGL(gbn_running_kernel, dim3((C + 255) / 256), dim3(256), 0, ST, mean, invstd, G, C, (int)Rg, eps, momentum, run_mean, run_var);
    FAVAE_LAUNCH_CHECK();
    return FAVAE_OK;
}

/* BatchNorm2d(train) + LeakyReLU backward of G groups: bins [G][nbins][2][C] hold (sum dd, sum dd xhat) per group; draw (dtype) =
 * gradient w.r.t. the conv output; dgamma / dbeta / dbias (may be NULL) are ADDED to (fp32 atomics: the caller's gradient arena is
 * zero at the start of a step).  dact / raw fp32, or bf16 with FAVAE_DACT_BF16 / FAVAE_RAW_BF16. */
extern "C" int favae_gbn_bwd(const void* dact, const void* act, const void* raw, const float* mean, const float* invstd, const float* gamma,
                             const float* ab, const double* bins, int nbins, void* draw, float* dgamma, float* dbeta, float* dbias, int G, long long Rg,
                             int C, int dtype, void* stream) {
    FAVAE_CHECK_ARG(dact && (act || ab) && raw && mean && invstd && gamma && bins && draw && dgamma && dbeta && nbins >= 1 && G >= 1 && Rg >= 1 && Rg < (1LL << 30),
                    "favae_gbn_bwd: bad argument");
    FAVAE_CHECK_ARG(C >= 4 && C <= 1024 && 1024 % C == 0, "favae_gbn_bwd: C = %d must divide 1024", C);
    FAVAE_CHECK_ARG(GB_BASE(dtype) == FAVAE_F32 || GB_BASE(dtype) == FAVAE_BF16, "favae_gbn_bwd: dtype");
    FAVAE_CHECK_ARG(!((GB_RAWBF(dtype) || GB_DACTBF(dtype)) && GB_BASE(dtype) != FAVAE_BF16), "favae_gbn_bwd: bf16 raw / dact need FAVAE_BF16");
    dim3 grid(gbn_grid(Rg, C), G);
    const bool rb = GB_RAWBF(dtype), db = GB_DACTBF(dtype);
#define GB_LAUNCH(T, DT, RT) hipLaunchKernelGGL((gbn_bwd_kernel<T, DT, RT>), grid, dim3(256), 0, ST, (const DT*)dact, (const T*)act, (const RT*)raw, mean, invstd, \
                                                gamma, ab, bins, nbins, (T*)draw, dgamma, dbeta, dbias, (int)Rg, C)
    if (GB_BASE(dtype) == FAVAE_F32) GB_LAUNCH(float, float, float);
    else if (db) { if (rb) GB_LAUNCH(bf16_t, bf16_t, bf16_t); else GB_LAUNCH(bf16_t, bf16_t, float); }
    else { if (rb) GB_LAUNCH(bf16_t, float, bf16_t); else GB_LAUNCH(bf16_t, float, float); }
#undef GB_LAUNCH
    FAVAE_LAUNCH_CHECK();
    return FAVAE_OK;
}
