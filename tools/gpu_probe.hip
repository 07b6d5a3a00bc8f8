// gpu_probe.hip — launch-floor probe: how long do back-to-back trivial kernels,
// small memsets and a captured hipGraph of them take on this box?
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void k_empty(int* p) { if (p && threadIdx.x == 9999) *p = 1; }
__global__ void k_touch(int* p, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] += 1; }
int main()
{
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    int* d; CK(hipMalloc(&d, 64 << 20));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const int N = 2000;
    auto run = [&](const char* name, auto fn) {
        for (int i = 0; i < 50; ++i) fn();
        hipStreamSynchronize(s);
        auto t0 = std::chrono::steady_clock::now();
        hipEventRecord(a, s);
        for (int i = 0; i < N; ++i) fn();
        hipEventRecord(b, s);
        auto t1 = std::chrono::steady_clock::now();
        hipStreamSynchronize(s);
        auto t2 = std::chrono::steady_clock::now();
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("%-34s gpu %.2f us/op   host-enqueue %.2f us/op   wall %.2f us/op\n", name, ms * 1e3 / N,
               std::chrono::duration<double, std::micro>(t1 - t0).count() / N,
               std::chrono::duration<double, std::micro>(t2 - t0).count() / N);
    };
    run("empty kernel 1x64", [&] { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s, d); });
    run("empty kernel 1024x256", [&] { hipLaunchKernelGGL(k_empty, dim3(1024), dim3(256), 0, s, d); });
    run("touch 1M ints (4MB rw)", [&] { hipLaunchKernelGGL(k_touch, dim3(4096), dim3(256), 0, s, d, 1 << 20); });
    run("memsetAsync 16B", [&] { hipMemsetAsync(d, 0, 16, s); });
    // graph of 8 kernels
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    for (int i = 0; i < 8; ++i) hipLaunchKernelGGL(k_empty, dim3(1024), dim3(256), 0, s, d);
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    run("graph of 8 empty kernels (per graph)", [&] { hipGraphLaunch(ge, s); });
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    for (int i = 0; i < 8; ++i) hipLaunchKernelGGL(k_touch, dim3(4096), dim3(256), 0, s, d, 1 << 20);
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    run("graph of 8 touch kernels (per graph)", [&] { hipGraphLaunch(ge, s); });
    return 0;
}
