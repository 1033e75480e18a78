// Factor-analysis (SSHIBA mean-field) kernels of FA-VAE on gfx950, all float64 like the reference's NumPy code.
//
//   fa_atb        C = A^T [B | 1]      tall-skinny sufficient statistics X^T Z, Z^T Z, X^T X, column sums
//                                      (reference condition_a_VAE/favae.py:989,1070,1091,1135-1139,1213) on
//                                      v_mfma_f64_16x16x4_f64, deterministic two-pass slab reduction
//   fa_rows_gemm  out = alpha (X P + c) + beta base   row-parallel N x d x K products: the Z mean (:986-987, 996-997),
//                                      the FA prior mean Z W^T + b (celeba_deepvanilla_vae.py:209) and Z W^T + b of
//                                      the multilabel update (:1212)
//   fa_mult       fused multilabel X / xi update with the bound terms (:1197-1229, 1649, 1678)
//   chol_inv      K x K SPD inverse + log-determinant in LDS (myInverse :904-922, update_w :1062-1065)
//   slogdet       K x K LU log-determinant (HGauss :1586-1604 on the de-pruned covariances)
//   small algebra single-block GEMM and the tau / b updates from the reduced statistics (:1077-1139)
#include <algorithm>
#include "common.h"
#include "favae_hip.h"

namespace {

__device__ __forceinline__ f64x4 mfma_f64(double a, double b, f64x4 c) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}

// ---------------------------------------------------------------------------------------------------------
// C[da][dbp] = A^T [B | 1]   (dbp = db + ones)
// One block = 8 waves per (row slab, tile group).  Every wave accumulates TI x TJ tiles of 16x16 over its share of the
// slab's rows (groups of 4 rows, interleaved between the waves), 4 rows per MFMA: lane l supplies row l>>4, column l&15
// of both operands straight from global memory (128-B row segments; the operands are read exactly once, so there is
// nothing to stage).  The eight waves are added through LDS in wave order, the slabs in slab order by atb_finish =>
// bitwise reproducible.  HBM-bound: 8 B * N * (da + db) read, 8 B * nslab * da * dbp of partials written and re-read.
// ---------------------------------------------------------------------------------------------------------
// SYM (A and B are the same matrix, Z^T [Z | 1] and X^T [X | 1]): the tile groups on the diagonal load the rows once for both
// operands and run the 10 tiles on or above the diagonal of their 4 x 4 instead of 16; the tiles below leave as the transposes
// of those above (bitwise the value the full product gives: the same products in the same order).
constexpr int TI = 4, TJ = 4;
constexpr int ATB_WAVES = 8;   // 8: two waves per SIMD at one block per CU (see atb_geom)

template <bool SYM>
__global__ __launch_bounds__(64 * ATB_WAVES) void atb_kernel(const double* __restrict__ A, int lda, int da,
                                                             const double* __restrict__ B, int ldb, int db, int ones, long long N,
                                                             int rps, int gj, double* __restrict__ partial) {
    __shared__ double red[ATB_WAVES][TJ * 4][64];   // 64 KB: one row of tiles (TJ x 4 registers) of every wave
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane >> 4, lc = lane & 15;
    const int dbp = db + ones;
    const int i0 = (blockIdx.y / gj) * (TI * 16), j0 = (blockIdx.y % gj) * (TJ * 16);
    const long long n0 = (long long)blockIdx.x * rps, n1 = min(N, n0 + (long long)rps);
    f64x4 acc[TI][TJ];
#pragma unroll
    for (int a = 0; a < TI; ++a)
#pragma unroll
        for (int b = 0; b < TJ; ++b) acc[a][b] = f64x4{0.0, 0.0, 0.0, 0.0};
    // tiles of this group that exist (block-uniform): a 20 x 21 statistic is 2 x 2 tiles, not the 4 x 4 of the group
    const int ti = min(TI, (da - i0 + 15) / 16), tj = min(TJ, (dbp - j0 + 15) / 16);
    const bool diag = SYM && i0 == j0;   // block-uniform
    // Two 4-row steps per iteration: up to 16 independent 8-byte loads per lane in flight.  The loads of the NEXT iteration are
    // issued before the MFMAs of this one (two register sets, the loop unrolled by two): the kernel is a stream over A and B, and
    // with ~3 waves per SIMD a wave that waits for its loads before every batch of MFMAs left the matrix pipe idle two thirds of the time.
    auto load = [&](long long n, double (&av)[2][TI], double (&bv)[2][TJ]) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const long long r = n + 4 * u + lr;
            const bool ok = r < n1;
#pragma unroll
            for (int a = 0; a < TI; ++a) {
                const int i = i0 + a * 16 + lc;
                av[u][a] = (a < ti && ok && i < da) ? A[r * lda + i] : 0.0;
            }
            if (diag) {   // (TI == TJ: column j of tile b is row i of tile b)
#pragma unroll
                for (int b = 0; b < TJ; ++b) {
                    const int j = j0 + b * 16 + lc;
                    bv[u][b] = (b >= tj || !ok) ? 0.0 : (j < db ? av[u][b] : (j == db && ones ? 1.0 : 0.0));
                }
            } else {
#pragma unroll
                for (int b = 0; b < TJ; ++b) {
                    const int j = j0 + b * 16 + lc;
                    bv[u][b] = (b >= tj || !ok) ? 0.0 : (j < db ? B[r * ldb + j] : (j == db && ones ? 1.0 : 0.0));
                }
            }
        }
    };
    auto mma = [&](const double (&av)[2][TI], const double (&bv)[2][TJ]) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int a = 0; a < TI; ++a)
#pragma unroll
                for (int b = 0; b < TJ; ++b)
                    if (a < ti && b < tj && !(diag && b < a)) acc[a][b] = mfma_f64(av[u][a], bv[u][b], acc[a][b]);   // uniform
    };
    constexpr long long STEP = 8 * ATB_WAVES;
    double av0[2][TI], bv0[2][TJ], av1[2][TI], bv1[2][TJ];
    long long n = n0 + 8 * wave;
    if (n < n1) load(n, av0, bv0);
    while (n < n1) {
        if (n + STEP < n1) load(n + STEP, av1, bv1);
        mma(av0, bv0);
        n += STEP;
        if (n >= n1) break;
        if (n + STEP < n1) load(n + STEP, av0, bv0);
        mma(av1, bv1);
        n += STEP;
    }
    // C/D map of v_mfma_f64_16x16x4_f64: col = lane&15, row = (lane>>4) + 4*reg
    double* out = partial + (size_t)blockIdx.x * da * dbp;
#pragma unroll
    for (int a = 0; a < TI; ++a) {
        if (a >= ti) break;   // block-uniform
#pragma unroll
        for (int b = 0; b < TJ; ++b)
#pragma unroll
            for (int g = 0; g < 4; ++g) red[wave][b * 4 + g][lane] = acc[a][b][g];
        __syncthreads();
#pragma unroll
        for (int k = 0; k < TJ * 4 / ATB_WAVES; ++k) {
            const int bg = wave + ATB_WAVES * k, b = bg >> 2, g = bg & 3;
            double t = (red[0][bg][lane] + red[1][bg][lane]) + (red[2][bg][lane] + red[3][bg][lane]);
            if constexpr (ATB_WAVES == 8) t += (red[4][bg][lane] + red[5][bg][lane]) + (red[6][bg][lane] + red[7][bg][lane]);
            const int i = i0 + a * 16 + lr + 4 * g, j = j0 + b * 16 + lc;
            if (diag && b < a) continue;   // (written as the transpose of tile (b, a))
            if (i < da && j < dbp) {
                out[(size_t)i * dbp + j] = t;
                if (diag && b > a && j < da) out[(size_t)j * dbp + i] = t;
            }
        }
        __syncthreads();
    }
}

__global__ void atb_finish(const double* __restrict__ partial, int nslab, int n, double* __restrict__ C) {
    const int i = blockIdx.x * 8 + (threadIdx.x >> 5), part = threadIdx.x & 31;   // 8 outputs x 32 slab-parts
    double s = 0.0;
    if (i < n)
        for (int k = part; k < nslab; k += 32) s += partial[(size_t)k * n + i];
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (i < n && part == 0) C[i] = s;
}

// ---------------------------------------------------------------------------------------------------------
// out[N][K] = alpha * (X[N][d] @ P[d][K] + c[K]) + beta * base[N][K]      (accumulate: out += alpha * X @ P)
// one wave = 16 rows x all K columns; A operand lane (row l&15, k l>>4), B operand lane (k l>>4, col l&15).
// The kernel is a stream over X and out with ~50 waves of 16 rows per CU; with one 8-byte load per lane in flight a wave holds
// 512 B and the chip ~4 MB, a third of what 8 TB/s needs at the loaded latency (measured 2.8 TB/s).  So a wave issues the loads of
// 64 columns of its rows at once (16 per lane) and, PL, the block first stages P, zero-padded to whole tiles, in LDS (d * K <= 7168
// doubles, the FA shapes; larger P is read through the caches as before), so that the MFMA loop waits for LDS, not for L2.
// ---------------------------------------------------------------------------------------------------------
constexpr int MAXTJ = 13;  // K <= 208
constexpr int RG_LDS_DOUBLES = 7168;   // 56 KB

__host__ __device__ constexpr int rg_lds_ld(int nt) { return nt * 16 + ((nt & 1) ? 0 : 16); }   // row stride = 16 mod 32 doubles: the 4 k-rows of a B fragment fall into different banks

template <int NT, bool PL>
__global__ __launch_bounds__(256) void rows_gemm_kernel(const double* __restrict__ X, int ldx, int d, const double* __restrict__ P,
                                                        int ldp, int K, const double* __restrict__ c, double alpha,
                                                        const double* __restrict__ base, int ldbase, double beta,
                                                        double* __restrict__ out, int ldo, long long N, int accumulate) {
    extern __shared__ __attribute__((aligned(16))) double Ps[];   // PL: [(d + 3) & ~3][LDS_LD]
    constexpr int LDS_LD = rg_lds_ld(NT);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lr = lane >> 4, lc = lane & 15;
    const long long n0 = ((long long)blockIdx.x * 4 + wave) * 16;
    const long long ra = n0 + lc;
    const bool oka = ra < N;
    constexpr int KU = 4;   // 4-column steps per panel (more in flight costs the registers that keep 8 waves per SIMD resident)
    double a[KU];
    // the first panel of X is on its way while P is staged
#pragma unroll
    for (int u = 0; u < KU; ++u) {
        const int k = 4 * u + lr;
        a[u] = (oka && k < d) ? X[ra * ldx + k] : 0.0;
    }
    if constexpr (PL) {
        const int d4 = (d + 3) & ~3;
        for (int e = threadIdx.x; e < d4 * NT * 16; e += 256) {
            const int k = e / (NT * 16), j = e - k * (NT * 16);
            Ps[k * LDS_LD + j] = (k < d && j < K) ? P[(size_t)k * ldp + j] : 0.0;
        }
        __syncthreads();
    }
    if (n0 >= N) return;
    f64x4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = f64x4{0.0, 0.0, 0.0, 0.0};
    for (int kb = 0; kb < d; kb += 4 * KU) {
        if (kb > 0) {
#pragma unroll
            for (int u = 0; u < KU; ++u) {
                const int k = kb + 4 * u + lr;
                a[u] = (oka && k < d) ? X[ra * ldx + k] : 0.0;
            }
        }
#pragma unroll
        for (int u = 0; u < KU; ++u) {
            const int k = kb + 4 * u + lr;
            if (kb + 4 * u < d) {   // wave-uniform
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const int j = t * 16 + lc;
                    double b;
                    if constexpr (PL) b = Ps[k * LDS_LD + j];
                    else b = (k < d && j < K) ? P[(size_t)k * ldp + j] : 0.0;
                    acc[t] = mfma_f64(a[u], b, acc[t]);
                }
            }
        }
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int j = t * 16 + lc;
        if (j >= K) continue;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const long long r = n0 + lr + 4 * g;
            if (r >= N) continue;
            double v = acc[t][g];
            if (accumulate) {
                out[r * ldo + j] += alpha * v;
            } else {
                if (c) v += c[j];
                v *= alpha;
                if (base) v += beta * base[r * ldbase + j];
                out[r * ldo + j] = v;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// multilabel view: X.cov, X.mean, xi and the bound pieces in one pass (cA/favae.py:1197-1229, 1649, 1678)
//   lam(xi)  = exp(log(sig(xi) - 1/2) - log(2 xi)),  sig(x) = exp(-log(1 + exp(-x)))      (:924-949, as written)
//   cov = 1/(tau + 2 lam(xi)); mean = (t - 1/2 + tau*zwb) * cov; xi' = sqrt(cov + mean^2)
//   partial[0] += log cov ; partial[1] += log sig(xi') + mean*t - (mean + xi')/2
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ double ref_sigmoid(double x) { return exp(-log(1.0 + exp(-x))); }
// The reference's expressions nest exp(log(.)) pairs: 12 fp64 transcendentals per element as written, which made mult_kernel bound
// by the fp64 ALUs (1.65 TB/s).  exp(-log(u)) = 1 / u, exp(log(p) - log(q)) = p / q and log(exp(-y)) = -y to within two or three
// roundings of the nested form (1e-16 relative; x = xi >= sqrt(cov) ~ 0.9 here, sig(x) - 1/2 does not cancel): 4 transcendentals,
// 3 divisions, 1 square root.  Held to 1e-13 against the oracle's expressions as written (tests/test_gpu_fa_kernels.py).
__device__ __forceinline__ double sigmoid_div(double x) { return 1.0 / (1.0 + exp(-x)); }

__global__ __launch_bounds__(256) void mult_kernel(const double* __restrict__ t, const double* __restrict__ zwb, double* __restrict__ xi,
                                                   double* __restrict__ mean, double* __restrict__ cov, double tau, size_t n,
                                                   double* __restrict__ partial) {
    __shared__ double sm[8];
    double a0 = 0.0, a1 = 0.0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const double x = xi[i];
        const double ti = t[i];
        const double lam = (sigmoid_div(x) - 0.5) / (2.0 * x);
        const double cv = 1.0 / (tau + 2.0 * lam);
        const double mn = (ti - 0.5 + tau * zwb[i]) * cv;
        const double xn = sqrt(cv + mn * mn);
        cov[i] = cv; mean[i] = mn; xi[i] = xn;
        a0 += log(cv);
        a1 += -log(1.0 + exp(-xn)) + mn * ti - 0.5 * (mn + xn);
    }
    a0 = wave_sum(a0); a1 = wave_sum(a1);
    if ((threadIdx.x & 63) == 0) { sm[threadIdx.x >> 6] = a0; sm[4 + (threadIdx.x >> 6)] = a1; }
    __syncthreads();
    if (threadIdx.x == 0) {
        partial[2 * blockIdx.x] = sm[0] + sm[1] + sm[2] + sm[3];
        partial[2 * blockIdx.x + 1] = sm[4] + sm[5] + sm[6] + sm[7];
    }
}

__global__ void pair_finish(const double* partial, int n, double* out2) {
    double s0 = 0.0, s1 = 0.0;
    for (int i = threadIdx.x; i < n; i += 64) { s0 += partial[2 * i]; s1 += partial[2 * i + 1]; }
    s0 = wave_sum(s0); s1 = wave_sum(s1);
    if (threadIdx.x == 0 && blockIdx.x == 0) { out2[0] = s0; out2[1] = s1; }
}

// xi = sqrt(cov + mean^2)  (Qdistribution init, :1912-1914)
__global__ void xi_init_kernel(const double* mean, const double* cov, double* xi, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) xi[i] = sqrt(cov[i] + mean[i] * mean[i]);
}

// ---------------------------------------------------------------------------------------------------------
// SPD inverse through the Cholesky factor, one workgroup, matrix resident in LDS (K <= 128) or in a global workspace.
//   A' = (s s^T) o A + diag_add I ;  out = out_scale (s s^T) o A'^-1 ;  info[0] = 0 ok / 1 not SPD ;
//   logdet[0] = log det A' ; logdet[1] = log det out
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void chol_inv_kernel(const double* __restrict__ A, int lda, int K, const double* __restrict__ s,
                                                       double diag_add, double out_scale, double* __restrict__ out, int ldo,
                                                       double* __restrict__ gwork, int use_lds, double* __restrict__ logdet,
                                                       int* __restrict__ info) {
    extern __shared__ __attribute__((aligned(16))) double smem[];   // [2 flag words][K*K][K] (no static LDS in front)
    int& bad = *(int*)smem;
    double* W = use_lds ? smem + 2 : gwork;        // K x K, row-major, lower triangle used
    double* col = use_lds ? smem + 2 + (size_t)K * K : gwork + (size_t)K * K;   // K
    const int tid = threadIdx.x;
    if (tid == 0) bad = 0;
    for (int e = tid; e < K * K; e += 256) {
        const int i = e / K, j = e - i * K;
        const double si = s ? s[i] : 1.0, sj = s ? s[j] : 1.0;
        W[e] = si * sj * A[(size_t)i * lda + j] + (i == j ? diag_add : 0.0);
    }
    __syncthreads();
    // right-looking Cholesky, lower
    for (int j = 0; j < K; ++j) {
        const double piv = W[j * K + j];
        if (!(piv > 0.0)) { if (tid == 0) bad = 1; }
        __syncthreads();
        if (bad) break;
        const double dj = sqrt(piv);
        for (int i = j + tid; i < K; i += 256) W[i * K + j] = (i == j) ? dj : W[i * K + j] / dj;
        __syncthreads();
        const int m = K - j - 1;
        for (int e = tid; e < m * m; e += 256) {
            const int i = j + 1 + e / m, k = j + 1 + e % m;
            if (k <= i) W[i * K + k] -= W[i * K + j] * W[k * K + j];
        }
        __syncthreads();
    }
    if (bad) {
        const double nan = __longlong_as_double(0x7ff8000000000000LL);
        for (int e = tid; e < K * K; e += 256) out[(size_t)(e / K) * ldo + (e % K)] = nan;
        if (tid == 0) { info[0] = 1; logdet[0] = nan; logdet[1] = nan; }
        return;
    }
    if (tid == 0) {
        double ld = 0.0, ls = 0.0;
        for (int j = 0; j < K; ++j) { ld += log(W[j * K + j]); if (s) ls += log(s[j]); }
        logdet[0] = 2.0 * ld;
        logdet[1] = K * log(out_scale) + 2.0 * ls - 2.0 * ld;
        info[0] = 0;
    }
    // in-place inverse of the lower-triangular factor, last column first
    for (int j = K - 1; j >= 0; --j) {
        const double ljj = W[j * K + j];
        for (int i = j + 1 + tid; i < K; i += 256) col[i] = W[i * K + j];
        __syncthreads();
        for (int i = j + 1 + tid; i < K; i += 256) {
            double acc = 0.0;
            for (int k = j + 1; k <= i; ++k) acc += W[i * K + k] * col[k];   // rows of the already inverted trailing block
            W[i * K + j] = -acc / ljj;
        }
        if (tid == 0) W[j * K + j] = 1.0 / ljj;
        __syncthreads();
    }
    // out = out_scale * (s s^T) o (Linv^T Linv)
    for (int e = tid; e < K * K; e += 256) {
        const int i = e / K, j = e - i * K;
        if (j > i) continue;
        double acc = 0.0;
        for (int k = i; k < K; ++k) acc += W[k * K + i] * W[k * K + j];
        const double si = s ? s[i] : 1.0, sj = s ? s[j] : 1.0;
        const double v = out_scale * si * sj * acc;
        out[(size_t)i * ldo + j] = v;
        out[(size_t)j * ldo + i] = v;
    }
}

// ---------------------------------------------------------------------------------------------------------
// Batched SPD inverse: ONE launch, one 16-wave workgroup per matrix, the matrix resident in LDS as a packed lower triangle
// (K (K + 1) / 2 doubles: K <= 201 fits the 160 KB of a CU), every O(K^3) part on v_mfma_f64_16x16x4_f64 over 16 x 16 tiles:
//   A'  = (s s^T) o (sum_t coef_t A_t) + diag_add I            the combination of update_Z (:970-980) is formed while loading
//   L   = chol(A')        right-looking, panels of 16 columns: diagonal block by one wave (wave-synchronous, no block barrier),
//                         panel solve one row per lane, trailing update as MFMA tiles          3 barriers per panel
//   X   = L^-1            block columns right to left: diagonal block by one wave, T = L21 X_pp one row per lane,
//                         X21 = -X22 T as MFMA tile rows held in registers across the barrier  (in place, no second buffer)
//   out = out_scale (s s^T) o (X^T X), log det A', log det out; a pivot <= 0 gives NaN and info = 1 (myInverse :917-922)
// Nothing is read back: the caller looks at info[] when it needs to (once per iteration).  myInverse (:904-922), the W
// covariance (:1062-1065) and the M + 1 (variant A: 2M + 1) inverses of update_Z (:970-980) of one iteration are two launches.
// ---------------------------------------------------------------------------------------------------------
constexpr int CB_MAXB = 12, CB_MAXT = 8, CB_THREADS = 1024, CB_KMAX_LDS = 201;
struct CholOne {
    const double* a[CB_MAXT]; double coef[CB_MAXT]; int nterm;
    const double* s; double s_mul; int s_sqrt;   // s_i = s ? (s_sqrt ? sqrt(s[i] * s_mul) : s[i]) : 1
    double diag_add, out_scale;
    double* out;
};
struct CholBatchArgs { CholOne m[CB_MAXB]; int nb, K, lda, ldo; double* logdet; int* info; };

__device__ __forceinline__ int pk(int i, int j) { return ((i * (i + 1)) >> 1) + j; }   // j <= i
__device__ __forceinline__ void wave_sync_lds() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

__global__ __launch_bounds__(CB_THREADS) void chol_inv_batched_kernel(const CholBatchArgs args_by_value) {
    // (the per-matrix record is picked by blockIdx: read it through the kernel-argument segment, a by-value struct indexed at run
    // time would be copied to scratch memory first)
    typedef const __attribute__((address_space(4))) CholBatchArgs* KArgs;
    KArgs ap = (KArgs)__builtin_amdgcn_kernarg_segment_ptr();
    (void)args_by_value;
    const int b = blockIdx.x;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    int& bad = *(int*)smem;
    double* W = smem + 2;
    const int K = ap->K, lda = ap->lda, ldo = ap->ldo;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = CB_THREADS / 64, lr = lane >> 4, lc = lane & 15;
    const int nt = (K + 15) >> 4;
    const double* sp = ap->m[b].s;
    const double s_mul = ap->m[b].s_mul, diag_add = ap->m[b].diag_add, out_scale = ap->m[b].out_scale;
    const int s_sqrt = ap->m[b].s_sqrt, nterm = ap->m[b].nterm;
    double* const outp = ap->m[b].out;
    double* const logdet = ap->logdet;
    int* const info = ap->info;
    auto sval = [&](int i) -> double { return sp ? (s_sqrt ? sqrt(sp[i] * s_mul) : sp[i]) : 1.0; };
    if (tid == 0) bad = 0;
    for (int e = tid; e < K * K; e += CB_THREADS) {
        const int i = e / K, j = e - i * K;
        if (j > i) continue;
        double v = 0.0;
        for (int t = 0; t < nterm; ++t) v += ap->m[b].coef[t] * ap->m[b].a[t][(size_t)i * lda + j];
        W[pk(i, j)] = sval(i) * sval(j) * v + (i == j ? diag_add : 0.0);
    }
    __syncthreads();
    // ---- Cholesky -------------------------------------------------------------------------------------------
    for (int p = 0; p < nt; ++p) {
        const int j0 = p << 4, w = min(16, K - j0);
        if (wave == 0) {
            for (int jj = 0; jj < w; ++jj) {
                const int j = j0 + jj;
                const double piv = W[pk(j, j)];
                if (!(piv > 0.0)) { if (lane == 0) bad = 1; break; }   // wave-uniform
                const double d = sqrt(piv);
                if (lane < w - jj) { const int r = j + lane; W[pk(r, j)] = lane == 0 ? d : W[pk(r, j)] / d; }
                wave_sync_lds();
#pragma unroll
                for (int e0 = 0; e0 < 256; e0 += 64) {
                    const int e = e0 + lane, r = j0 + (e >> 4), c = j0 + (e & 15);
                    if (c > j && c <= r && r < j0 + w) W[pk(r, c)] -= W[pk(r, j)] * W[pk(c, j)];
                }
                wave_sync_lds();
            }
        }
        __syncthreads();
        if (bad) break;
        {   // panel solve: row i of the panel times the inverse transpose of the diagonal block, one row per lane
            const int i = j0 + w + tid;
            if (i < K) {
                double a[16];
#pragma unroll
                for (int c = 0; c < 16; ++c) a[c] = c < w ? W[pk(i, j0 + c)] : 0.0;
#pragma unroll
                for (int c = 0; c < 16; ++c) {
                    if (c < w) {
                        a[c] /= W[pk(j0 + c, j0 + c)];
#pragma unroll
                        for (int q = c + 1; q < 16; ++q)
                            if (q < w) a[q] -= a[c] * W[pk(j0 + q, j0 + c)];
                    }
                }
#pragma unroll
                for (int c = 0; c < 16; ++c)
                    if (c < w) W[pk(i, j0 + c)] = a[c];
            }
        }
        __syncthreads();
        {   // trailing update: tile (I, J) -= L[I, panel] L[J, panel]^T, p < J <= I
            int t = 0;
            for (int I = p + 1; I < nt; ++I)
                for (int J = p + 1; J <= I; ++J, ++t) {
                    if (t % nw != wave) continue;
                    f64x4 acc = f64x4{0.0, 0.0, 0.0, 0.0};
                    const int ra = (I << 4) + lc, rb = (J << 4) + lc;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int col = j0 + 4 * q + lr;
                        const bool ck = col < j0 + w;
                        const double av = (ck && ra < K) ? W[pk(ra, col)] : 0.0;
                        const double bv = (ck && rb < K) ? W[pk(rb, col)] : 0.0;
                        acc = mfma_f64(av, bv, acc);
                    }
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int r = (I << 4) + lr + 4 * g, c = (J << 4) + lc;
                        if (r < K && c <= r) W[pk(r, c)] -= acc[g];
                    }
                }
        }
        __syncthreads();
    }
    if (bad) {
        const double nan = __longlong_as_double(0x7ff8000000000000LL);
        for (int e = tid; e < K * K; e += CB_THREADS) outp[(size_t)(e / K) * ldo + (e % K)] = nan;
        if (tid == 0) { info[b] = 1; logdet[2 * b] = nan; logdet[2 * b + 1] = nan; }
        return;
    }
    if (wave == nw - 1) {   // log-determinants from the factor's diagonal (one wave, fixed order), beside the first inverse panel
        double ld = 0.0, ls = 0.0;
        for (int j = lane; j < K; j += 64) { ld += log(W[pk(j, j)]); if (sp) ls += log(sval(j)); }
        ld = wave_sum(ld); ls = wave_sum(ls);
        if (lane == 0) {
            logdet[2 * b] = 2.0 * ld;
            logdet[2 * b + 1] = K * log(out_scale) + 2.0 * ls - 2.0 * ld;
            info[b] = 0;
        }
    }
    __syncthreads();   // (the diagonal is overwritten below)
    // ---- X = L^-1, block columns right to left ------------------------------------------------------------------
    for (int p = nt - 1; p >= 0; --p) {
        const int j0 = p << 4, w = min(16, K - j0);
        if (wave == 0) {   // diagonal block: columns right to left; a column is read whole before its entries are replaced
            for (int jj = w - 1; jj >= 0; --jj) {
                const int j = j0 + jj, i = j + 1 + lane;
                const double ljj = W[pk(j, j)];
                double acc = 0.0;
                if (lane < w - 1 - jj)
                    for (int k = j + 1; k <= i; ++k) acc += W[pk(i, k)] * W[pk(k, j)];
                wave_sync_lds();
                if (lane < w - 1 - jj) W[pk(i, j)] = -acc / ljj;
                if (lane == 0) W[pk(j, j)] = 1.0 / ljj;
                wave_sync_lds();
            }
        }
        __syncthreads();
        {   // T = L21 X_pp, one row per lane, in place
            const int i = j0 + w + tid;
            if (i < K) {
                double a[16], tt[16];
#pragma unroll
                for (int c = 0; c < 16; ++c) a[c] = c < w ? W[pk(i, j0 + c)] : 0.0;
#pragma unroll
                for (int c = 0; c < 16; ++c) {
                    double acc = 0.0;
#pragma unroll
                    for (int q = c; q < 16; ++q)
                        if (q < w && c < w) acc += a[q] * W[pk(j0 + q, j0 + c)];
                    tt[c] = acc;
                }
#pragma unroll
                for (int c = 0; c < 16; ++c)
                    if (c < w) W[pk(i, j0 + c)] = tt[c];
            }
        }
        __syncthreads();
        // X[I, p] = -sum_{p < J <= I} X[I, J] T[J]: one tile row per wave, kept in registers until every wave has read T
        f64x4 acc = f64x4{0.0, 0.0, 0.0, 0.0};
        const int I = p + 1 + wave;
        if (I < nt) {
            for (int J = p + 1; J <= I; ++J) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int ra = (I << 4) + lc, ca = (J << 4) + 4 * q + lr;     // X22[I][J]: row lc, k = lr
                    const double av = (ra < K && ca <= ra) ? W[pk(ra, ca)] : 0.0;
                    const int rb = (J << 4) + 4 * q + lr;                           // T[J]: k = lr, column lc of the panel
                    const double bv = (rb < K && lc < w) ? W[pk(rb, j0 + lc)] : 0.0;
                    acc = mfma_f64(av, bv, acc);
                }
            }
        }
        __syncthreads();
        if (I < nt) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int r = (I << 4) + lr + 4 * g;
                if (r < K && lc < w) W[pk(r, j0 + lc)] = -acc[g];
            }
        }
        __syncthreads();
    }
    // ---- out = out_scale (s s^T) o (X^T X) ----------------------------------------------------------------------
    {
        int t = 0;
        for (int I = 0; I < nt; ++I)
            for (int J = 0; J <= I; ++J, ++t) {
                if (t % nw != wave) continue;
                f64x4 acc = f64x4{0.0, 0.0, 0.0, 0.0};
                for (int Kt = I; Kt < nt; ++Kt) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int r = (Kt << 4) + 4 * q + lr, ca = (I << 4) + lc, cb = (J << 4) + lc;
                        const double av = (r < K && ca <= r) ? W[pk(r, ca)] : 0.0;   // A[m = lc][k = lr] = X[r][I*16 + m]
                        const double bv = (r < K && cb <= r) ? W[pk(r, cb)] : 0.0;   // B[k = lr][n = lc] = X[r][J*16 + n]
                        acc = mfma_f64(av, bv, acc);
                    }
                }
                const int j = (J << 4) + lc;
                const double sj = j < K ? sval(j) : 0.0;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int i = (I << 4) + lr + 4 * g;
                    if (i < K && j < K && j <= i) {
                        const double v = out_scale * sval(i) * sj * acc[g];
                        outp[(size_t)i * ldo + j] = v;
                        outp[(size_t)j * ldo + i] = v;
                    }
                }
            }
    }
}

// LU with partial pivoting: sign and log|det| of a general K x K matrix (np.linalg.slogdet)
__global__ __launch_bounds__(256) void slogdet_kernel(const double* __restrict__ A, int lda, int K, double* __restrict__ gwork,
                                                      int use_lds, double* __restrict__ out2) {
    extern __shared__ __attribute__((aligned(16))) double smem[];   // [piv_row, sgn, lad, pad][K*K]
    int& piv_row = *(int*)smem;
    double& sgn = smem[1];
    double& lad = smem[2];
    double* W = use_lds ? smem + 4 : gwork;
    const int tid = threadIdx.x;
    for (int e = tid; e < K * K; e += 256) W[e] = A[(size_t)(e / K) * lda + (e % K)];
    if (tid == 0) { sgn = 1.0; lad = 0.0; }
    __syncthreads();
    for (int j = 0; j < K; ++j) {
        if (tid == 0) {
            int p = j; double best = fabs(W[j * K + j]);
            for (int i = j + 1; i < K; ++i) { const double v = fabs(W[i * K + j]); if (v > best) { best = v; p = i; } }
            piv_row = p;
        }
        __syncthreads();
        const int p = piv_row;
        if (p != j) {
            for (int k = tid; k < K; k += 256) { const double tmp = W[j * K + k]; W[j * K + k] = W[p * K + k]; W[p * K + k] = tmp; }
        }
        __syncthreads();
        const double pv = W[j * K + j];
        if (tid == 0) {
            if (p != j) sgn = -sgn;
            if (pv < 0) sgn = -sgn;
            if (pv == 0.0) sgn = 0.0;
            lad += log(fabs(pv));
        }
        if (pv == 0.0) break;
        const int m = K - j - 1;
        for (int i = j + 1 + tid; i < K; i += 256) W[i * K + j] /= pv;
        __syncthreads();
        for (int e = tid; e < m * m; e += 256) {
            const int i = j + 1 + e / m, k = j + 1 + e % m;
            W[i * K + k] -= W[i * K + j] * W[j * K + k];
        }
        __syncthreads();
    }
    __syncthreads();
    if (tid == 0) { out2[0] = sgn; out2[1] = sgn == 0.0 ? -INFINITY : lad; }
}

// ---------------------------------------------------------------------------------------------------------
// small dense fp64 GEMM (operands up to a few hundred on a side): C = alpha op(A) op(B) + beta C
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void small_gemm_kernel(int ta, int tb, int M, int N, int Kd, double alpha, const double* __restrict__ A,
                                                         int lda, const double* __restrict__ B, int ldb, double beta,
                                                         double* __restrict__ C, int ldc) {
    for (int e = blockIdx.x * 256 + threadIdx.x; e < M * N; e += gridDim.x * 256) {
        const int i = e / N, j = e - i * N;
        double acc = 0.0;
        for (int k = 0; k < Kd; ++k) {
            const double a = ta ? A[(size_t)k * lda + i] : A[(size_t)i * lda + k];
            const double b = tb ? B[(size_t)j * ldb + k] : B[(size_t)k * ldb + j];
            acc += a * b;
        }
        C[(size_t)i * ldc + j] = alpha * acc + (beta != 0.0 ? beta * C[(size_t)i * ldc + j] : 0.0);
    }
}

// W.mean = tau * (XtZ - b^T colsumZ) Sigma_W   and   W.prodT = W^T W + d Sigma_W         (cA/favae.py:1070-1072)
// xtz1: [d][K+1] = X^T [Z | 1]; ztz1 row K.. provides colsumZ via zsum[K]
__global__ __launch_bounds__(256) void w_update_kernel(const double* __restrict__ xtz1, const double* __restrict__ zsum,
                                                       const double* __restrict__ b, const double* __restrict__ wcov, double tau, int d,
                                                       int K, double* __restrict__ wmean, double* __restrict__ wprodT,
                                                       double* __restrict__ tmp, const int* __restrict__ info, double* __restrict__ wcov_out) {
    // info != NULL: the flag of the inverse that produced wcov (favae_chol_inv_batched); not SPD -> "not updated" (:1073-1075)
    if (info && *info != 0) return;   // uniform
    // tmp[d][K] = XtZ - b^T colsumZ
    for (int e = threadIdx.x; e < d * K; e += 256) {
        const int i = e / K, k = e - i * K;
        tmp[e] = xtz1[(size_t)i * (K + 1) + k] - b[i] * zsum[k];
    }
    __syncthreads();
    for (int e = threadIdx.x; e < d * K; e += 256) {
        const int i = e / K, k = e - i * K;
        double acc = 0.0;
        for (int l = 0; l < K; ++l) acc += tmp[(size_t)i * K + l] * wcov[(size_t)l * K + k];
        wmean[e] = tau * acc;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < K * K; e += 256) {
        const int k = e / K, l = e - k * K;
        double acc = 0.0;
        for (int i = 0; i < d; ++i) acc += wmean[(size_t)i * K + k] * wmean[(size_t)i * K + l];
        wprodT[e] = acc + d * wcov[e];
        if (wcov_out) wcov_out[e] = wcov[e];
    }
}

// b update (cA/favae.py:1077-1092) and tau update (:1121-1139) from reduced statistics.
//   stats: xsum[d] (1^T X), zsum[K] (1^T Z), xtz1 [d][K+1], trxx = tr<X^T X> (incl. sum cov), zprodT, wprodT, wmean
// out: bmean[d], scal[0]=b_cov00, scal[1]=b_prodT ; tau: scal[2]=a, scal[3]=b
__global__ __launch_bounds__(256) void b_update_kernel(const double* __restrict__ xsum, const double* __restrict__ zsum,
                                                       const double* __restrict__ wmean, double tau, double N, int d, int K,
                                                       double* __restrict__ bmean, double* __restrict__ scal) {
    __shared__ double sm[4];
    const double bc = 1.0 / (1.0 + N * tau);
    double acc = 0.0;
    for (int i = threadIdx.x; i < d; i += 256) {
        double zw = 0.0;
        for (int k = 0; k < K; ++k) zw += zsum[k] * wmean[(size_t)i * K + k];
        const double v = tau * (xsum[i] - zw) * bc;
        bmean[i] = v;
        acc += v * v;
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) { scal[0] = bc; scal[1] = sm[0] + sm[1] + sm[2] + sm[3] + d * bc; }
}

__global__ __launch_bounds__(256) void tau_update_kernel(const double* __restrict__ xsum, const double* __restrict__ zsum,
                                                         const double* __restrict__ xtz1, double trxx, const double* __restrict__ zprodT,
                                                         const double* __restrict__ wprodT, const double* __restrict__ wmean,
                                                         const double* __restrict__ bmean, double bprodT, double a0, double b0, double N,
                                                         int d, int K, double* __restrict__ scal, const double* __restrict__ bprodT_dev,
                                                         const double* __restrict__ xtx1_dev, const double* __restrict__ covsum_dev) {
    __shared__ double sm[4];
    if (bprodT_dev) bprodT = *bprodT_dev;   // <b b^T> as favae_fa_b_update left it on the device
    if (xtx1_dev) {                         // tr<X^T X> from X^T [X | 1] (d x (d + 1)) and the column sums of X.cov, in a fixed order
        trxx = 0.0;
        for (int i = 0; i < d; ++i) trxx += xtx1_dev[(size_t)i * (d + 1) + i];
        if (covsum_dev) for (int i = 0; i < d; ++i) trxx += covsum_dev[i];
    }
    double acc = 0.0;
    for (int e = threadIdx.x; e < K * K; e += 256) acc += wprodT[e] * zprodT[(size_t)(e % K) * K + e / K];       // tr(WW ZZ)
    for (int e = threadIdx.x; e < d * K; e += 256) {
        const int i = e / K, k = e - i * K;
        acc += -2.0 * wmean[e] * xtz1[(size_t)i * (K + 1) + k]          // -2 tr(W Z^T X)
               + 2.0 * zsum[k] * wmean[e] * bmean[i];                  // +2 sum(Z W^T b^T)
    }
    for (int i = threadIdx.x; i < d; i += 256) acc += -2.0 * xsum[i] * bmean[i];                                 // -2 sum(X b^T)
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        const double s = trxx + sm[0] + sm[1] + sm[2] + sm[3] + N * bprodT;
        scal[2] = (a0 + 0.5 * d * N) / (d * N);
        scal[3] = (b0 + 0.5 * s) / (d * N);
    }
}

// alpha.b[k] = (b0 + diag(W.prodT)[k] / 2) / d   (:1107-1112)
__global__ void alpha_update_kernel(const double* __restrict__ wprodT, double b0, int d, int K, double* __restrict__ alpha_b) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < K) alpha_b[k] = (b0 + 0.5 * wprodT[(size_t)k * K + k]) / d;
}

// aux = I + sum_m coef[m] * WprodT_m   (update_Z :970-971, 976-980) -- up to 8 views
struct ZAuxArgs { const double* wp[8]; double coef[8]; int m; };
__global__ void z_aux_kernel(ZAuxArgs a, int K, double* __restrict__ aux) {
    for (int e = blockIdx.x * 256 + threadIdx.x; e < K * K; e += gridDim.x * 256) {
        double v = (e / K == e % K) ? 1.0 : 0.0;
        for (int m = 0; m < a.m; ++m) v += a.coef[m] * a.wp[m][e];
        aux[e] = v;
    }
}

// keep the columns sel[0..Kn) of a row-major [N][K] matrix (pruning, :507)
__global__ void gather_cols_kernel(const double* __restrict__ in, int K, const int* __restrict__ sel, int Kn, double* __restrict__ out,
                                   size_t N) {
    const size_t total = N * Kn;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t r = i / Kn;
        out[i] = in[r * K + sel[i - r * Kn]];
    }
}

// out = a x + b y  (small K x K / d x K combinations such as Z.prodT = Z^T Z + N Sigma_Z, :989)
__global__ void axpby_kernel(size_t n, double a, const double* __restrict__ x, double b, const double* __restrict__ y, double* __restrict__ out) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) out[i] = a * x[i] + (y ? b * y[i] : 0.0);
}

inline int ew_grid(size_t n) { return (int)std::min<size_t>(2048, std::max<size_t>(1, (n + 255) / 256)); }

}  // namespace

#define ST ((hipStream_t)stream)

// slabs of rows: ONE block of 8 waves per CU over the (tile group, slab) grid.  The kernel needs ~235 registers, i.e. two waves
// per SIMD: with 792 four-wave blocks a CU held two and the grid ran in 1.5 rounds of short blocks (Z^T [Z | 1] at N = 202 599:
// 88 us); 256 eight-wave blocks run in one round, with 3x fewer partials to write and add (47 us; 128: 75 us, 512: 55 us).
// At least 256 rows per slab.
static void atb_geom(long long N, int da, int dbp, int* nslab, int* rps, int* gi, int* gj) {
    *gi = cdiv(cdiv(da, 16), TI);
    *gj = cdiv(cdiv(dbp, 16), TJ);
    const int groups = (*gi) * (*gj);
    static const int blocks = getenv("FAVAE_ATB_BLOCKS") ? atoi(getenv("FAVAE_ATB_BLOCKS")) : 256;
    long long ns = std::max(1LL, std::min<long long>(cdivll(N, 256), std::max(1, blocks / groups)));
    long long r = cdivll(N, ns);
    r = (r + 15) & ~15LL;
    *rps = (int)r;
    *nslab = (int)cdivll(N, r);
}

extern "C" long long favae_fa_atb_scratch_doubles(long long N, int da, int db, int ones) {
    int nslab, rps, gi, gj;
    atb_geom(N, da, db + (ones ? 1 : 0), &nslab, &rps, &gi, &gj);
    return (long long)nslab * da * (db + (ones ? 1 : 0));
}

static int atb_run(const double* A, int lda, int da, const double* B, int ldb, int db, int ones, long long N, double* C,
                   double* scratch, size_t scratch_doubles, const char* who, void* stream) {
    const int dbp = db + (ones ? 1 : 0);
    int nslab, rps, gi, gj;
    atb_geom(N, da, dbp, &nslab, &rps, &gi, &gj);
    FAVAE_CHECK_ARG((size_t)nslab * da * dbp <= scratch_doubles, "%s: scratch too small (%zu < %zu doubles)", who, scratch_doubles,
                    (size_t)nslab * da * dbp);
    FAVAE_CHECK_ARG(lda >= da && (db == 0 || ldb >= db), "%s: leading dimension smaller than the row", who);
    static const bool no_sym = getenv("FAVAE_ATB_NO_SYM") != nullptr;   // (developer switch: the full product)
    if ((!B || B == A) && db == da && ldb == lda && !no_sym)
        hipLaunchKernelGGL(atb_kernel<true>, dim3(nslab, gi * gj), dim3(64 * ATB_WAVES), 0, ST, A, lda, da, A, lda, db, ones ? 1 : 0, N, rps, gj, scratch);
    else
        hipLaunchKernelGGL(atb_kernel<false>, dim3(nslab, gi * gj), dim3(64 * ATB_WAVES), 0, ST, A, lda, da, B ? B : A, ldb, db, ones ? 1 : 0, N, rps, gj, scratch);
    FAVAE_LAUNCH_CHECK();
    const int n = da * dbp;
    hipLaunchKernelGGL(atb_finish, dim3(cdiv(n, 8)), dim3(256), 0, ST, scratch, nslab, n, C);
    FAVAE_LAUNCH_CHECK();
    return FAVAE_OK;
}

extern "C" int favae_fa_atb(const double* A, int lda, int da, const double* B, int ldb, int db, int ones, long long N, double* C,
                            double* scratch, size_t scratch_doubles, void* stream) {
    FAVAE_CHECK_ARG(A && C && scratch && N > 0 && da > 0 && db >= 0 && (B || db == 0) && (db > 0 || ones), "favae_fa_atb: bad argument");
    return atb_run(A, lda, da, B, ldb, db, ones, N, C, scratch, scratch_doubles, "favae_fa_atb", stream);
}

// ---- favae_fa_atb for callers that cannot allocate (HIP graph capture, serving loops) ---------------------------------
// The slab partials live in a file-scope device array instead of a caller-supplied scratch buffer.  One array per GPU
// (each device holds its own copy at its own address), shared by every call on that GPU: calls that overlap on one GPU
// (two streams) race on it.  Nothing in it has to start at zero: atb_kernel overwrites every partial that atb_finish reads.
constexpr size_t ATB_STATIC_DOUBLES = (size_t)1 << 22;   // 32 MB; atb_geom never needs more while da * (db + ones) <= 4194304
static __device__ __align__(256) double g_atb_static[ATB_STATIC_DOUBLES];
static double* g_atb_static_ptr[64] = {nullptr};          // host: device address of the array on device i (not the symbol)

extern "C" int favae_fa_atb_static(const double* A, int lda, int da, const double* B, int ldb, int db, int ones, long long N,
                                   double* C, void* stream) {
    FAVAE_CHECK_ARG(A && C && N > 0 && da > 0 && db >= 0 && (B || db == 0) && (db > 0 || ones), "favae_fa_atb_static: bad argument");
    const size_t dbp = (size_t)db + (ones ? 1 : 0);
    FAVAE_CHECK_ARG((size_t)da * dbp <= ATB_STATIC_DOUBLES,
                    "favae_fa_atb_static: da * (db + ones) = %zu exceeds the static scratch of %zu doubles; use favae_fa_atb with a "
                    "caller-owned scratch", (size_t)da * dbp, ATB_STATIC_DOUBLES);
    int dev = 0;
    FAVAE_HIP(hipGetDevice(&dev));
    FAVAE_CHECK_ARG(dev >= 0 && dev < 64, "favae_fa_atb_static: device index %d", dev);
    if (!g_atb_static_ptr[dev]) {   // once per device, with that device current (no allocation: the array came with the code object)
        double* p = nullptr;
        FAVAE_HIP(hipGetSymbolAddress((void**)&p, HIP_SYMBOL(g_atb_static)));
        FAVAE_CHECK_ARG(p != nullptr, "favae_fa_atb_static: no device address for the static scratch");
        g_atb_static_ptr[dev] = p;
    }
    return atb_run(A, lda, da, B, ldb, db, ones, N, C, g_atb_static_ptr[dev], ATB_STATIC_DOUBLES, "favae_fa_atb_static", stream);
}

extern "C" int favae_fa_rows_gemm(const double* X, int ldx, int d, const double* P, int ldp, int K, const double* c, double alpha,
                                  const double* base, int ldbase, double beta, double* out, int ldo, long long N, int accumulate,
                                  void* stream) {
    FAVAE_CHECK_ARG(X && P && out && N > 0 && d > 0 && K > 0, "favae_fa_rows_gemm: bad argument");
    dim3 grid((unsigned)cdivll(N, 64)), block(256);
    // a wave holds 16 rows x up to 13 column tiles in registers; wider outputs (the FA prior of a 512-dimensional latent)
    // run as column panels of 208
    for (int k0 = 0; k0 < K; k0 += 16 * MAXTJ) {
        const int Kp = std::min(K - k0, 16 * MAXTJ), nt = cdiv(Kp, 16);
        const double* Pp = P + k0;
        const double* cp = c ? c + k0 : nullptr;
        const double* bp = base ? base + k0 : nullptr;
        double* op = out + k0;
        static const bool no_pl = getenv("FAVAE_RG_NO_LDS") != nullptr;   // (developer switch)
        const size_t pdoubles = (size_t)((d + 3) & ~3) * rg_lds_ld(nt);
        const bool pl = pdoubles <= (size_t)RG_LDS_DOUBLES && !no_pl;
#define RG_CASE(n) case n: if (pl) hipLaunchKernelGGL((rows_gemm_kernel<n, true>), grid, block, pdoubles * 8, ST, X, ldx, d, Pp, ldp, Kp, cp, alpha, bp, ldbase, beta, op, ldo, N, accumulate); \
                           else hipLaunchKernelGGL((rows_gemm_kernel<n, false>), grid, block, 0, ST, X, ldx, d, Pp, ldp, Kp, cp, alpha, bp, ldbase, beta, op, ldo, N, accumulate); break;
        switch (nt) {
            RG_CASE(1) RG_CASE(2) RG_CASE(3) RG_CASE(4) RG_CASE(5) RG_CASE(6) RG_CASE(7) RG_CASE(8) RG_CASE(9) RG_CASE(10) RG_CASE(11) RG_CASE(12) RG_CASE(13)
        }
#undef RG_CASE
        FAVAE_LAUNCH_CHECK();
    }
    return FAVAE_OK;
}

extern "C" int favae_fa_mult_update(const double* t, const double* zwb, double* xi, double* mean, double* cov, double tau, long long n,
                                    double* out2, double* scratch, size_t scratch_doubles, void* stream) {
    FAVAE_CHECK_ARG(t && zwb && xi && mean && cov && out2 && scratch && n > 0, "favae_fa_mult_update: bad argument");
    const int grid = ew_grid((size_t)n);
    FAVAE_CHECK_ARG((size_t)2 * grid <= scratch_doubles, "favae_fa_mult_update: scratch too small");
    hipLaunchKernelGGL(mult_kernel, dim3(grid), dim3(256), 0, ST, t, zwb, xi, mean, cov, tau, (size_t)n, scratch);
    FAVAE_LAUNCH_CHECK();
    hipLaunchKernelGGL(pair_finish, dim3(1), dim3(64), 0, ST, scratch, grid, out2);
    FAVAE_LAUNCH_CHECK();
    return FAVAE_OK;
}

// Semi-supervised multilabel view: update_t (cA/favae.py:1176-1195) and its write-back t.mean[mask] = tS.mean[mask]
// (:858-861) in one pass.  tS.mean = sigmoid(X.mean), tS.cov = exp(log tS.mean - log(1 + exp(X.mean))) over ALL entries;
// out2 = (sum log tS.cov, sum_mask X.mean * (t_new - t_old)): the second is what the bound term sum X.mean * t.mean
// (:1678), taken by mult_kernel with the labels of before, gains from the new labels.
__global__ __launch_bounds__(256) void mult_ss_kernel(const double* __restrict__ mean, const unsigned char* __restrict__ mask,
                                                      double* __restrict__ t, size_t n, double* __restrict__ partial) {
    __shared__ double sm[8];
    double a0 = 0.0, a1 = 0.0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const double x = mean[i];
        const double sg = ref_sigmoid(x);
        a0 += log(sg) - log(1.0 + exp(x));
        if (mask[i]) {
            a1 += x * (sg - t[i]);
            t[i] = sg;
        }
    }
    a0 = wave_sum(a0); a1 = wave_sum(a1);
    if ((threadIdx.x & 63) == 0) { sm[threadIdx.x >> 6] = a0; sm[4 + (threadIdx.x >> 6)] = a1; }
    __syncthreads();
    if (threadIdx.x == 0) {
        partial[2 * blockIdx.x] = sm[0] + sm[1] + sm[2] + sm[3];
        partial[2 * blockIdx.x + 1] = sm[4] + sm[5] + sm[6] + sm[7];
    }
}

extern "C" int favae_fa_mult_ss_update(const double* mean, const unsigned char* mask, double* t, long long n, double* out2,
                                       double* scratch, size_t scratch_doubles, void* stream) {
    FAVAE_CHECK_ARG(mean && mask && t && out2 && scratch && n > 0, "favae_fa_mult_ss_update: bad argument");
    const int grid = ew_grid((size_t)n);
    FAVAE_CHECK_ARG((size_t)2 * grid <= scratch_doubles, "favae_fa_mult_ss_update: scratch too small");
    hipLaunchKernelGGL(mult_ss_kernel, dim3(grid), dim3(256), 0, ST, mean, mask, t, (size_t)n, scratch);
    FAVAE_LAUNCH_CHECK();
    hipLaunchKernelGGL(pair_finish, dim3(1), dim3(64), 0, ST, scratch, grid, out2);
    FAVAE_LAUNCH_CHECK();
    return FAVAE_OK;
}

// SSHIBA.predict, multilabel output (cA/favae.py:1476-1479): s = m_x * scale[d]; mean_t = sigmoid(s) written the
// reference's way exp(-log(1+exp(-s))) (:924-935), var_t = exp(s) / (1 + exp(s)^2) as written at :1479.
__global__ __launch_bounds__(256) void mult_predict_kernel(const double* __restrict__ mx, const double* __restrict__ scale,
                                                           double* __restrict__ mean_t, double* __restrict__ var_t, size_t n, int d) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const double s = mx[i] * scale[i % d];
        mean_t[i] = exp(-log(1.0 + exp(-s)));
        const double e = exp(s);
        var_t[i] = e / (1.0 + e * e);
    }
}

extern "C" int favae_fa_mult_predict(const double* m_x, const double* scale, double* mean_t, double* var_t, long long N, int d,
                                     void* stream) {
    FAVAE_CHECK_ARG(m_x && scale && mean_t && var_t && N > 0 && d > 0, "favae_fa_mult_predict: bad argument");
    const size_t n = (size_t)N * d;
    const int grid = (int)std::min<size_t>(4096, (n + 255) / 256);
    hipLaunchKernelGGL(mult_predict_kernel, dim3(grid), dim3(256), 0, ST, m_x, scale, mean_t, var_t, n, d);
    FAVAE_LAUNCH_CHECK();
    return FAVAE_OK;
}

extern "C" int favae_fa_xi_init(const double* mean, const double* cov, double* xi, long long n, void* stream) {
    FAVAE_CHECK_ARG(mean && cov && xi && n > 0, "favae_fa_xi_init: bad argument");
    hipLaunchKernelGGL(xi_init_kernel, dim3(ew_grid((size_t)n)), dim3(256), 0, ST, mean, cov, xi, (size_t)n);
    FAVAE_LAUNCH_CHECK();
    return FAVAE_OK;
}

// combination kernel of the global-memory fallback (K > 201): aux = sum_t coef_t A_t
__global__ void chol_comb_kernel(CholOne m, int K, int lda, double* __restrict__ aux) {
    for (int e = blockIdx.x * 256 + threadIdx.x; e < K * K; e += gridDim.x * 256) {
        const int i = e / K, j = e - i * K;
        double v = 0.0;
#pragma unroll
        for (int t = 0; t < CB_MAXT; ++t)
            if (t < m.nterm) v += m.coef[t] * m.a[t][(size_t)i * lda + j];
        aux[e] = v;
    }
}
__global__ void chol_svec_kernel(const double* __restrict__ s, double s_mul, int K, double* __restrict__ out) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i < K) out[i] = sqrt(s[i] * s_mul);
}

/* nb <= 12 SPD inverses of size K in one launch (see chol_inv_batched_kernel).  Matrix b:
 *   A'_b = (s_b s_b^T) o (sum_{t < nterm[b]} coef[b*8 + t] * A[b*8 + t]) + diag_add[b] I,   out[b] = out_scale[b] (s_b s_b^T) o A'_b^-1
 * with s_b = NULL (ones), s[b] as given (s_sqrt[b] = 0) or sqrt(s[b][i] * s_mul[b]) (s_sqrt[b] = 1: 1 / sqrt(alpha.mean) from alpha.b).
 * logdet2 [nb][2] = (log det A'_b, log det out[b]); info [nb] = 0 / 1 (not SPD: out[b] and its log-determinants are NaN).
 * Asynchronous: nothing is read back.  K <= 201: LDS-resident; larger K (up to 1024) runs matrix by matrix through `work`
 * (>= 2 K^2 + 2 K doubles) in global memory. */
extern "C" int favae_chol_inv_batched(int nb, const double* const* A, const double* coef, const int* nterm, int lda, int K,
                                      const double* const* s, const double* s_mul, const int* s_sqrt, const double* diag_add,
                                      const double* out_scale, double* const* out, int ldo, double* logdet2, int* info_dev,
                                      double* work, size_t work_doubles, void* stream) {
    FAVAE_CHECK_ARG(nb >= 1 && nb <= CB_MAXB && A && coef && nterm && diag_add && out_scale && out && logdet2 && info_dev,
                    "favae_chol_inv_batched: bad argument (1..%d matrices)", CB_MAXB);
    FAVAE_CHECK_ARG(K > 0 && K <= 1024 && lda >= K && ldo >= K, "favae_chol_inv_batched: K = %d, lda = %d, ldo = %d", K, lda, ldo);
    CholBatchArgs a;
    a.nb = nb; a.K = K; a.lda = lda; a.ldo = ldo; a.logdet = logdet2; a.info = info_dev;
    for (int b = 0; b < nb; ++b) {
        CholOne& m = a.m[b];
        FAVAE_CHECK_ARG(nterm[b] >= 1 && nterm[b] <= CB_MAXT && out[b], "favae_chol_inv_batched: matrix %d: 1..%d terms, non-null out", b, CB_MAXT);
        m.nterm = nterm[b];
        for (int t = 0; t < CB_MAXT; ++t) {
            m.a[t] = t < nterm[b] ? A[b * CB_MAXT + t] : nullptr;
            m.coef[t] = t < nterm[b] ? coef[b * CB_MAXT + t] : 0.0;
            FAVAE_CHECK_ARG(t >= nterm[b] || m.a[t], "favae_chol_inv_batched: matrix %d term %d is null", b, t);
        }
        m.s = s ? s[b] : nullptr;
        m.s_mul = s_mul ? s_mul[b] : 1.0;
        m.s_sqrt = s_sqrt ? s_sqrt[b] : 0;
        m.diag_add = diag_add[b];
        m.out_scale = out_scale[b];
        m.out = out[b];
    }
    for (int b = nb; b < CB_MAXB; ++b) a.m[b] = a.m[0];
    if (K <= CB_KMAX_LDS) {
        const size_t lds_bytes = (2 + (size_t)K * (K + 1) / 2) * sizeof(double);
        if (lds_bytes > 64 * 1024)
            FAVAE_HIP(hipFuncSetAttribute((const void*)chol_inv_batched_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
        hipLaunchKernelGGL(chol_inv_batched_kernel, dim3(nb), dim3(CB_THREADS), lds_bytes, ST, a);
        FAVAE_LAUNCH_CHECK();
        return FAVAE_OK;
    }
    // beyond the LDS: the single-workgroup kernel on a global workspace, one matrix after the other (same stream)
    const size_t KK = (size_t)K * K;
    FAVAE_CHECK_ARG(work && work_doubles >= 2 * KK + 2 * (size_t)K, "favae_chol_inv_batched: K = %d needs a workspace of %zu doubles", K, 2 * KK + 2 * (size_t)K);
    for (int b = 0; b < nb; ++b) {
        const CholOne& m = a.m[b];
        const double* src = m.a[0];
        int ld = lda;
        if (m.nterm > 1 || m.coef[0] != 1.0) {
            hipLaunchKernelGGL(chol_comb_kernel, dim3(std::min(64, cdiv(K * K, 256))), dim3(256), 0, ST, m, K, lda, work);
            src = work; ld = K;
        }
        const double* sv = m.s;
        if (m.s && m.s_sqrt) {
            hipLaunchKernelGGL(chol_svec_kernel, dim3(cdiv(K, 64)), dim3(64), 0, ST, m.s, m.s_mul, K, work + 2 * KK + K);
            sv = work + 2 * KK + K;
        }
        hipLaunchKernelGGL(chol_inv_kernel, dim3(1), dim3(256), 16, ST, src, ld, K, sv, m.diag_add, m.out_scale, m.out, ldo, work + KK, 0,
                           logdet2 + 2 * b, info_dev + b);
        FAVAE_LAUNCH_CHECK();
    }
    return FAVAE_OK;
}

/* One matrix, synchronous: returns FAVAE_NOT_SPD (and fills out with NaN) when a pivot is not positive. */
extern "C" int favae_chol_inv(const double* A, int lda, int K, const double* s, double diag_add, double out_scale, double* out,
                              int ldo, double* work, double* logdet2, int* info_dev, void* stream) {
    FAVAE_CHECK_ARG(A && out && work && logdet2 && info_dev && K > 0 && K <= 1024, "favae_chol_inv: bad argument");
    const double* Ap[CB_MAXT] = {A};
    const double coef[CB_MAXT] = {1.0};
    const int nterm = 1, s_sqrt = 0;
    const double s_mul = 1.0;
    int rc = favae_chol_inv_batched(1, Ap, coef, &nterm, lda, K, s ? &s : nullptr, &s_mul, &s_sqrt, &diag_add, &out_scale, &out, ldo, logdet2,
                                    info_dev, work, (size_t)2 * K * K + 2 * K, stream);
    if (rc) return rc;
    int info = 0;
    FAVAE_HIP(hipMemcpyAsync(&info, info_dev, sizeof(int), hipMemcpyDeviceToHost, ST));
    FAVAE_HIP(hipStreamSynchronize(ST));
    return info ? FAVAE_NOT_SPD : FAVAE_OK;
}

extern "C" int favae_slogdet(const double* A, int lda, int K, double* work, double* out2, void* stream) {
    FAVAE_CHECK_ARG(A && work && out2 && K > 0 && K <= 1024, "favae_slogdet: bad argument");
    const size_t full = ((size_t)K * K + 4) * sizeof(double);
    const int use_lds = full <= 150 * 1024;
    const size_t lds_bytes = use_lds ? full : 32;
    if (lds_bytes > 64 * 1024)
        FAVAE_HIP(hipFuncSetAttribute((const void*)slogdet_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    hipLaunchKernelGGL(slogdet_kernel, dim3(1), dim3(256), lds_bytes, ST, A, lda, K, work, use_lds, out2);
    FAVAE_LAUNCH_CHECK();
    return FAVAE_OK;
}

extern "C" int favae_small_gemm(int transA, int transB, int M, int N, int Kd, double alpha, const double* A, int lda, const double* B,
                                int ldb, double beta, double* C, int ldc, void* stream) {
    FAVAE_CHECK_ARG(A && B && C && M > 0 && N > 0 && Kd > 0, "favae_small_gemm: bad argument");
    hipLaunchKernelGGL(small_gemm_kernel, dim3(std::min(256, cdiv(M * N, 256))), dim3(256), 0, ST, transA, transB, M, N, Kd, alpha, A, lda, B,
                       ldb, beta, C, ldc);
    FAVAE_LAUNCH_CHECK();
    return FAVAE_OK;
}

extern "C" int favae_fa_w_update(const double* xtz1, const double* zsum, const double* b, const double* wcov, double tau, int d, int K,
                                 double* wmean, double* wprodT, double* tmp, const int* info, double* wcov_out, void* stream) {
    FAVAE_CHECK_ARG(xtz1 && zsum && b && wcov && wmean && wprodT && tmp && d > 0 && K > 0, "favae_fa_w_update: bad argument");
    hipLaunchKernelGGL(w_update_kernel, dim3(1), dim3(256), 0, ST, xtz1, zsum, b, wcov, tau, d, K, wmean, wprodT, tmp, info, wcov_out);
    FAVAE_LAUNCH_CHECK();
    return FAVAE_OK;
}

extern "C" int favae_fa_b_update(const double* xsum, const double* zsum, const double* wmean, double tau, double N, int d, int K,
                                 double* bmean, double* scal, void* stream) {
    FAVAE_CHECK_ARG(xsum && zsum && wmean && bmean && scal && d > 0 && K > 0, "favae_fa_b_update: bad argument");
    hipLaunchKernelGGL(b_update_kernel, dim3(1), dim3(256), 0, ST, xsum, zsum, wmean, tau, N, d, K, bmean, scal);
    FAVAE_LAUNCH_CHECK();
    return FAVAE_OK;
}

extern "C" int favae_fa_tau_update(const double* xsum, const double* zsum, const double* xtz1, double trxx, const double* zprodT,
                                   const double* wprodT, const double* wmean, const double* bmean, double bprodT, double a0, double b0,
                                   double N, int d, int K, double* scal, const double* bprodT_dev, const double* xtx1_dev,
                                   const double* covsum_dev, void* stream) {
    FAVAE_CHECK_ARG(xsum && zsum && xtz1 && zprodT && wprodT && wmean && bmean && scal && d > 0 && K > 0, "favae_fa_tau_update: bad argument");
    hipLaunchKernelGGL(tau_update_kernel, dim3(1), dim3(256), 0, ST, xsum, zsum, xtz1, trxx, zprodT, wprodT, wmean, bmean, bprodT, a0, b0, N,
                       d, K, scal, bprodT_dev, xtx1_dev, covsum_dev);
    FAVAE_LAUNCH_CHECK();
    return FAVAE_OK;
}

extern "C" int favae_fa_alpha_update(const double* wprodT, double b0, int d, int K, double* alpha_b, void* stream) {
    FAVAE_CHECK_ARG(wprodT && alpha_b && d > 0 && K > 0, "favae_fa_alpha_update: bad argument");
    hipLaunchKernelGGL(alpha_update_kernel, dim3(cdiv(K, 64)), dim3(64), 0, ST, wprodT, b0, d, K, alpha_b);
    FAVAE_LAUNCH_CHECK();
    return FAVAE_OK;
}

extern "C" int favae_fa_z_aux(const double* const* wprodT, const double* coef, int m, int K, double* aux, void* stream) {
    FAVAE_CHECK_ARG(wprodT && coef && aux && m >= 0 && m <= 8 && K > 0, "favae_fa_z_aux: bad argument (at most 8 views)");
    ZAuxArgs a;
    a.m = m;
    for (int i = 0; i < m; ++i) { a.wp[i] = wprodT[i]; a.coef[i] = coef[i]; }
    hipLaunchKernelGGL(z_aux_kernel, dim3(std::min(64, cdiv(K * K, 256))), dim3(256), 0, ST, a, K, aux);
    FAVAE_LAUNCH_CHECK();
    return FAVAE_OK;
}

extern "C" int favae_gather_cols(const double* in, int K, const int* sel, int Kn, double* out, long long N, void* stream) {
    FAVAE_CHECK_ARG(in && sel && out && K > 0 && Kn > 0 && Kn <= K && N > 0, "favae_gather_cols: bad argument");
    hipLaunchKernelGGL(gather_cols_kernel, dim3(ew_grid((size_t)N * Kn)), dim3(256), 0, ST, in, K, sel, Kn, out, (size_t)N);
    FAVAE_LAUNCH_CHECK();
    return FAVAE_OK;
}

extern "C" int favae_axpby(long long n, double a, const double* x, double b, const double* y, double* out, void* stream) {
    FAVAE_CHECK_ARG(n > 0 && x && out, "favae_axpby: bad argument");
    hipLaunchKernelGGL(axpby_kernel, dim3(ew_grid((size_t)n)), dim3(256), 0, ST, (size_t)n, a, x, b, y, out);
    FAVAE_LAUNCH_CHECK();
    return FAVAE_OK;
}
