// Shared device/host helpers for libfavae_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define FAVAE_OK 0
#define FAVAE_NOT_SPD 1          // numerical flag: the matrix handed to the Cholesky inverse is not SPD
#define FAVAE_ERR_ARG (-1)
#define FAVAE_ERR_HIP (-2)

extern "C" void favae_set_error(const char* fmt, ...);

#define FAVAE_CHECK_ARG(cond, ...)                 \
    do {                                           \
        if (!(cond)) {                             \
            favae_set_error(__VA_ARGS__);          \
            return FAVAE_ERR_ARG;                  \
        }                                          \
    } while (0)

#define FAVAE_HIP(call)                                                                         \
    do {                                                                                        \
        hipError_t e_ = (call);                                                                 \
        if (e_ != hipSuccess) {                                                                 \
            favae_set_error("%s:%d %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(e_)); \
            return FAVAE_ERR_HIP;                                                               \
        }                                                                                       \
    } while (0)

extern "C" int favae_debug_sync_enabled(void);
#define FAVAE_LAUNCH_CHECK()                                              \
    do {                                                                  \
        FAVAE_HIP(hipGetLastError());                                     \
        if (favae_debug_sync_enabled()) FAVAE_HIP(hipDeviceSynchronize()); \
    } while (0)

typedef unsigned short bf16_t;  // storage type of a bfloat16 element

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) double f64x4;

__device__ __forceinline__ float bf16_to_f32(bf16_t h) { return __uint_as_float(((unsigned)h) << 16); }

// round-to-nearest-even; NaN stays NaN (quiet).  The integer form (7 VALU instructions), kept as the definition the hardware
// conversion is tested against (favae_debug_bf16_convert)
__device__ __forceinline__ bf16_t f32_to_bf16_sw(float f) {
    unsigned u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);
    return (bf16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}
#ifdef FAVAE_SW_BF16
__device__ __forceinline__ bf16_t f32_to_bf16(float f) { return f32_to_bf16_sw(f); }
#else
// gfx950's v_cvt_pk_bf16_f32 (one instruction; the compiler emits it for the conversion to __bf16)
__device__ __forceinline__ bf16_t f32_to_bf16(float f) { return __builtin_bit_cast(unsigned short, (__bf16)f); }
#endif
// two values in one 32-bit word (a in the low half): one v_cvt_pk_bf16_f32
__device__ __forceinline__ unsigned pack_bf16x2(float a, float b) {
#ifdef FAVAE_SW_BF16
    return (unsigned)f32_to_bf16(a) | ((unsigned)f32_to_bf16(b) << 16);
#else
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    const f32x2_t v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
#endif
}

template <class T> __device__ __forceinline__ float to_f32(T v);
template <> __device__ __forceinline__ float to_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ float to_f32<bf16_t>(bf16_t v) { return bf16_to_f32(v); }
template <class T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float v) { return f32_to_bf16(v); }

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

// Every global store / atomic this thread has issued has been performed (gfx9 counts stores and atomics in vmcnt).
// Used before the barrier that precedes a cross-block ticket: neither a workgroup-scope release fence nor s_barrier
// waits for outstanding vector memory operations on gfx950.
__device__ __forceinline__ void wait_vm_writes() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// Words to be stored by the first block of the next GEMM launch of this host thread (vae_model.hip, sp_record): dependency
// points of the stream the launch goes to, passed when everything queued before that launch is complete.
// A small reduction that rides with this thread's next weight-gradient launch (igemm_pm): out[c] = sum over the slabs of
// partial[slab][c], the second half of favae_bn_lrelu_bwd_bins with FAVAE_DBIAS_DEFER.  The first cdiv(C, 8) blocks of the launch
// take 8 channels each before their own work: no launch of its own on the stream beside the main chain.
// out0 != NULL (the sums of favae_conv_out_ll with FAVAE_DBIAS_DEFER): channel 0 goes to *out0 = off0 + scale0 * sum as a double
// and channel c >= 1 to out[c - 1].
// nout: entries of `out` that exist (out0 form: Cout of the convolution, 1..4; the launch holds five sums per block whatever Cout is).
struct SlabRider { const double* partial; float* out; int nslab, C; double* out0; double off0, scale0; int nout; };
SlabRider& favae_slab_rider();                       // this thread's pending rider (api.hip)
inline SlabRider favae_take_slab_rider() { SlabRider& r = favae_slab_rider(); const SlabRider k = r; r.partial = nullptr; return k; }
struct FlagCarry { unsigned* p[4]; unsigned val; int n; };
FlagCarry& favae_flag_carry();                       // this thread's pending words (api.hip)
inline FlagCarry favae_take_flag_carry() { FlagCarry& c = favae_flag_carry(); const FlagCarry k = c; c.n = 0; return k; }
#ifdef __HIPCC__
__device__ __forceinline__ void carry_store(const FlagCarry& c) {
    if (c.n > 0 && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0) {
        // (explicit members: a dynamically indexed kernel argument would be copied to scratch memory)
        __hip_atomic_store(c.p[0], c.val, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        if (c.n > 1) __hip_atomic_store(c.p[1], c.val, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        if (c.n > 2) __hip_atomic_store(c.p[2], c.val, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        if (c.n > 3) __hip_atomic_store(c.p[3], c.val, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
#endif

#ifdef __HIPCC__
// One Adam update (torch.optim.Adam defaults: no weight decay, no amsgrad; cA/celeba_deepvanilla_vae.py:131,250) with every rounding
// spelled out, so that the three kernels that take the step (adam_kernel, adam_wprep_kernel, the epilogue of igemm_pm<..., ADAM>)
// give the same bits whatever the compiler would contract in each of them (the __f*_rn forms of this toolchain are plain operators,
// so contraction is switched off as well):
//   m += (1 - b1) (g - m);  v = v b2 + (1 - b2) g g;  p -= step_size * m / (sqrt(v) / sqrt(1 - b2^t) + eps),  step_size = lr / (1 - b1^t)
// a product that must be rounded before its consumer sees it (g * gscale in front of g - m would otherwise become one fma)
__device__ __forceinline__ float mul_rounded(float a, float b) {
#pragma clang fp contract(off)
    return a * b;
}
__device__ __forceinline__ void adam_update(float& p, float g, float& m, float& v, float b1, float b2, float eps, float step_size, float bc2_sqrt) {
#pragma clang fp contract(off)
    m = __fmaf_rn(__fsub_rn(1.f, b1), __fsub_rn(g, m), m);
    v = __fmaf_rn(g, __fmul_rn(__fsub_rn(1.f, b2), g), __fmul_rn(v, b2));
    const float denom = __fadd_rn(__fdiv_rn(__fsqrt_rn(v), bc2_sqrt), eps);
    p = __fmaf_rn(-step_size, __fdiv_rn(m, denom), p);
}
#endif

// Optional live timing of selected launches (bench.py roofline): every `stride`-th launch of a kernel class is bracketed by
// HIP events on its launch stream (api.hip); `work` is the algorithmic flops (GEMM classes) or bytes (HBM-bound classes).
extern "C" int favae_prof_begin(int cls, double work, void* stream);
extern "C" void favae_prof_end(int idx, void* stream);
struct ProfScope {
    void* st; int idx;
    ProfScope(int cls, double work, hipStream_t s) : st((void*)s), idx(favae_prof_begin(cls, work, (void*)s)) {}
    ~ProfScope() { if (idx >= 0) favae_prof_end(idx, st); }
};

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline long long cdivll(long long a, long long b) { return (a + b - 1) / b; }
