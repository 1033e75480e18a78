// Floor of a dependent kernel chain on one stream (developer tool): empty grid, grid that reads, grid that writes 16 KB per block.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ __launch_bounds__(256) void k_none(float4* p) { if (p == nullptr && threadIdx.x == 9999) printf("x"); }
__global__ __launch_bounds__(256) void k_read(const float4* __restrict__ in, float4* out) {
    float4 a = make_float4(0, 0, 0, 0);
    for (int i = 0; i < 4; ++i) { const float4 v = in[((size_t)blockIdx.x * 4 + i) * 256 + threadIdx.x]; a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w; }
    if (a.x == 12345.f) out[0] = a;
}
__global__ __launch_bounds__(256) void k_write(float4* out, float s) {
    for (int i = 0; i < 4; ++i) out[((size_t)blockIdx.x * 4 + i) * 256 + threadIdx.x] = make_float4(s, s, s, s);
}
__global__ __launch_bounds__(256) void k_rw(const float4* __restrict__ in, float4* out) {
    for (int i = 0; i < 4; ++i) { float4 v = in[((size_t)blockIdx.x * 4 + i) * 256 + threadIdx.x]; v.x += 1.f; out[((size_t)blockIdx.x * 4 + i) * 256 + threadIdx.x] = v; }
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    const int blocks[3] = {256, 512, 4096};
    float4 *a, *b; hipMalloc(&a, (size_t)4096 * 16384); hipMalloc(&b, (size_t)4096 * 16384);
    hipMemset(a, 0, (size_t)4096 * 16384);
    const int N = 400;
    for (int nb : blocks) {
        for (int which = 0; which < 4; ++which) {
            auto launch = [&](int i) {
                float4* x = (i & 1) ? a : b; float4* y = (i & 1) ? b : a;
                if (which == 0) hipLaunchKernelGGL(k_none, dim3(nb), dim3(256), 0, s, x);
                else if (which == 1) hipLaunchKernelGGL(k_read, dim3(nb), dim3(256), 0, s, x, y);
                else if (which == 2) hipLaunchKernelGGL(k_write, dim3(nb), dim3(256), 0, s, y, (float)i);
                else hipLaunchKernelGGL(k_rw, dim3(nb), dim3(256), 0, s, x, y);
            };
            for (int i = 0; i < 20; ++i) launch(i);
            hipStreamSynchronize(s);
            const double t = now();
            for (int i = 0; i < N; ++i) launch(i);
            hipStreamSynchronize(s);
            const char* nm[4] = {"empty", "read 16 KB/block", "write 16 KB/block", "read+write 16 KB/block"};
            printf("%5d blocks  %-24s %6.2f us per launch\n", nb, nm[which], 1e6 * (now() - t) / N);
        }
    }
    return 0;
}
