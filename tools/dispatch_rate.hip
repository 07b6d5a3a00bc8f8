// dispatch_rate.hip — how many kernel dispatches per second does the GPU accept from 1, 2, 4, 8 streams?
// (empty kernels and small 256-block kernels; each stream's kernels are ordered, streams are independent)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void k_empty(int* p) { if (p && threadIdx.x == 9999) *p = 1; }
__global__ void k_spin(int* p, int cycles) { long long t0 = clock64(); while (clock64() - t0 < cycles) {} if (p && threadIdx.x == 9999) *p = 1; }
int main()
{
    int* d; CK(hipMalloc(&d, 1 << 20));
    for (int ns : {1, 2, 4, 8}) {
        std::vector<hipStream_t> st(ns);
        for (auto& s : st) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        for (int variant = 0; variant < 3; ++variant) {
            const int N = 2000;
            auto body = [&](int i) {
                hipStream_t s = st[i % ns];
                if (variant == 0) hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s, d);
                else if (variant == 1) hipLaunchKernelGGL(k_empty, dim3(1024), dim3(1024), 0, s, d);
                else hipLaunchKernelGGL(k_spin, dim3(256), dim3(256), 0, s, d, 10000); // ~4 us of work on 256 blocks
            };
            for (int i = 0; i < 200; ++i) body(i);
            CK(hipDeviceSynchronize());
            auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < N; ++i) body(i);
            auto t1 = std::chrono::steady_clock::now();
            CK(hipDeviceSynchronize());
            auto t2 = std::chrono::steady_clock::now();
            printf("%d stream(s), %-22s: enqueue %.2f us/launch, total %.2f us/launch\n", ns,
                   variant == 0 ? "empty 1x64" : variant == 1 ? "empty 1024x1024" : "spin 4us 256x256",
                   std::chrono::duration<double, std::micro>(t1 - t0).count() / N, std::chrono::duration<double, std::micro>(t2 - t0).count() / N);
        }
        for (auto& s : st) hipStreamDestroy(s);
    }
    return 0;
}
