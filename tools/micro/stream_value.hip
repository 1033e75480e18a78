// hipStreamWriteValue32 / hipStreamWaitValue32 as a cross-stream dependency whose two halves may be issued by different host
// threads in any order (developer tool): host cost per call and the GPU-side hand-over latency, against event record + wait.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void k_empty(int* p) { if (p && threadIdx.x == 9999) *p = 1; }
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main() {
    hipStream_t a, b;
    CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
    unsigned* flag = nullptr;
    hipError_t e = hipExtMallocWithFlags((void**)&flag, 256, hipMallocSignalMemory);  // (not available here: plain device memory below)
    printf("signal memory: %s\n", hipGetErrorString(e));
    if (e != hipSuccess) CK(hipMalloc((void**)&flag, 256));
    CK(hipMemset(flag, 0, 256));
    const int N = 1000;
    hipEvent_t ev[64];
    for (auto& x : ev) hipEventCreateWithFlags(&x, hipEventDisableTiming);
    for (int i = 0; i < 100; ++i) hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, a, nullptr);
    CK(hipDeviceSynchronize());
    // reference: kernel a, record, wait b, kernel b
    double t = now();
    for (int i = 0; i < N; ++i) {
        hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, a, nullptr);
        hipEventRecord(ev[i & 63], a);
        hipStreamWaitEvent(b, ev[i & 63], 0);
        hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, b, nullptr);
        hipEventRecord(ev[(i + 32) & 63], b);
        hipStreamWaitEvent(a, ev[(i + 32) & 63], 0);
    }
    double t1 = now(); CK(hipDeviceSynchronize()); double t2 = now();
    printf("events: ping-pong a->b->a: host %.2f us/iter, total %.2f us/iter\n", 1e6 * (t1 - t) / N, 1e6 * (t2 - t) / N);
    // the same through memory values; the wait is issued BEFORE the write (as a second host thread might)
    t = now();
    double tw = 0, tv = 0;
    for (int i = 0; i < N; ++i) {
        const unsigned s1 = 2 * i + 1, s2 = 2 * i + 2;
        double x0 = now();
        CK(hipStreamWaitValue32(b, flag, s1, hipStreamWaitValueGte, 0xFFFFFFFFu));
        double x1 = now();
        hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, b, nullptr);
        double x2 = now();
        CK(hipStreamWriteValue32(b, flag + 8, s2, 0));
        double x3 = now();
        hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, a, nullptr);
        CK(hipStreamWriteValue32(a, flag, s1, 0));
        CK(hipStreamWaitValue32(a, flag + 8, s2, hipStreamWaitValueGte, 0xFFFFFFFFu));
        tw += x1 - x0; tv += x3 - x2;
    }
    t1 = now(); CK(hipDeviceSynchronize()); t2 = now();
    printf("values: ping-pong a->b->a: host %.2f us/iter (wait %.2f us, write %.2f us per call), total %.2f us/iter\n", 1e6 * (t1 - t) / N,
           1e6 * tw / N, 1e6 * tv / N, 1e6 * (t2 - t) / N);
    // GPU-side cost of a record between two kernels of one chain: nothing / event record / value write
    for (int mode = 0; mode < 3; ++mode) {
        CK(hipDeviceSynchronize());
        t = now();
        for (int i = 0; i < N; ++i) {
            hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, a, nullptr);
            if (mode == 1) hipEventRecord(ev[i & 63], a);
            if (mode == 2) CK(hipStreamWriteValue32(a, flag + 16, (unsigned)i, 0));
        }
        CK(hipDeviceSynchronize()); t2 = now();
        printf("chain of kernels with %s after each: %.2f us/kernel\n", mode == 0 ? "nothing" : mode == 1 ? "an event record" : "a value write", 1e6 * (t2 - t) / N);
    }
    // a satisfied wait in front of each kernel: event / value
    for (int mode = 1; mode < 3; ++mode) {
        hipEventRecord(ev[0], b);
        CK(hipStreamWriteValue32(b, flag + 24, 5u, 0));
        CK(hipDeviceSynchronize());
        t = now();
        for (int i = 0; i < N; ++i) {
            if (mode == 1) hipStreamWaitEvent(a, ev[0], 0);
            if (mode == 2) CK(hipStreamWaitValue32(a, flag + 24, 5u, hipStreamWaitValueGte, 0xFFFFFFFFu));
            hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, a, nullptr);
        }
        CK(hipDeviceSynchronize()); t2 = now();
        printf("chain of kernels with a satisfied %s wait before each: %.2f us/kernel\n", mode == 1 ? "event" : "value", 1e6 * (t2 - t) / N);
    }
    return 0;
}
