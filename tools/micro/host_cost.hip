// Host-side cost of the HIP calls a training step is made of (developer tool): launches, event records, stream waits.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void k_empty(int* p) { if (p && threadIdx.x == 9999) *p = 1; }
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    hipStream_t a, b;
    hipStreamCreateWithFlags(&a, hipStreamNonBlocking);
    hipStreamCreateWithFlags(&b, hipStreamNonBlocking);
    hipEvent_t ev[64];
    for (auto& e : ev) hipEventCreateWithFlags(&e, hipEventDisableTiming);
    const int N = 2000;
    for (int i = 0; i < 100; ++i) hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, a, nullptr);
    hipDeviceSynchronize();
    double t = now();
    for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, a, nullptr);
    double t1 = now(); hipDeviceSynchronize(); double t2 = now();
    printf("launch (one stream): host %.2f us/call, total incl. drain %.2f us/call\n", 1e6 * (t1 - t) / N, 1e6 * (t2 - t) / N);
    t = now();
    for (int i = 0; i < N; ++i) hipEventRecord(ev[i & 63], a);
    t1 = now(); hipDeviceSynchronize();
    printf("hipEventRecord: host %.2f us/call\n", 1e6 * (t1 - t) / N);
    t = now();
    for (int i = 0; i < N; ++i) { hipEventRecord(ev[i & 63], a); hipStreamWaitEvent(b, ev[i & 63], 0); }
    t1 = now(); hipDeviceSynchronize();
    printf("record + wait on another stream: host %.2f us/pair\n", 1e6 * (t1 - t) / N);
    t = now();
    for (int i = 0; i < N; ++i) { hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, a, nullptr); hipEventRecord(ev[i & 63], a); hipStreamWaitEvent(b, ev[i & 63], 0); hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, b, nullptr); }
    t1 = now(); hipDeviceSynchronize(); t2 = now();
    printf("launch a, record, wait b, launch b: host %.2f us/iter, total %.2f us/iter\n", 1e6 * (t1 - t) / N, 1e6 * (t2 - t) / N);
    // dependent chain on one stream: GPU-side per-launch latency of an empty kernel
    t = now();
    for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, a, nullptr);
    hipDeviceSynchronize(); t2 = now();
    printf("empty kernel chain: %.2f us/kernel\n", 1e6 * (t2 - t) / N);
    hipMemsetAsync(nullptr, 0, 0, a);
    return 0;
}
