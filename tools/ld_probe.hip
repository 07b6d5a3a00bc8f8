// ld_probe.hip — cost of (un)aligned 4/16-byte loads streaming an L2-resident byte array,
// in the access shape of the coarse similarity kernel: every wave reads 64 consecutive
// 4- or 16-byte words starting at an arbitrary byte offset.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t __attribute__((aligned(1))) u32u;
struct __attribute__((aligned(1))) u128u { uint32_t a, b, c, d; };
template <int MODE>
__global__ __launch_bounds__(256) void k(const uint8_t* __restrict__ buf, int span, int nfeat, int mis, uint32_t* out)
{
    // block handles 1024 (MODE<2) or 4096 (MODE>=2) consecutive bytes for nfeat pseudo-random offsets
    uint32_t acc = 0;
    const int lane_off = (MODE < 2 ? 4 : 16) * threadIdx.x + blockIdx.x * (MODE < 2 ? 1024 : 4096);
    uint32_t seed = blockIdx.y * 2654435761u;
    for (int f = 0; f < nfeat; f += 8) {
        uint32_t v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            seed = seed * 1664525u + 1013904223u;
            int off = (int)((seed >> 8) % (uint32_t)span);
            off = (off & ~15) + mis;
            const uint8_t* p = buf + off + lane_off;
            if (MODE < 2) v[u] = *(const u32u*)p;
            else { u128u w = *(const u128u*)p; v[u] = w.a + w.b + w.c + w.d; }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += v[u];
    }
    if (acc == 0x12345678u) out[0] = acc;
}
int main()
{
    const int N = 2 << 20;
    uint8_t* d; hipMalloc(&d, N + (1 << 16)); hipMemset(d, 1, N + (1 << 16));
    uint32_t* o; hipMalloc(&o, 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    auto run = [&](const char* name, int mode, int mis) {
        const int nfeat = 72, templates = 360;
        const int span = N - 8192;
        dim3 grid(mode < 2 ? 3 : 1, templates);  // ~3000 positions per template
        for (int it = 0; it < 3; ++it) {
            if (mode == 0) hipLaunchKernelGGL(k<0>, grid, dim3(256), 0, 0, d, span, nfeat, mis, o);
            else hipLaunchKernelGGL(k<2>, dim3(1, templates), dim3(192), 0, 0, d, span, nfeat, mis, o);
        }
        hipEventRecord(a, 0);
        const int reps = 20;
        for (int it = 0; it < reps; ++it) {
            if (mode == 0) hipLaunchKernelGGL(k<0>, grid, dim3(256), 0, 0, d, span, nfeat, mis, o);
            else hipLaunchKernelGGL(k<2>, dim3(1, templates), dim3(192), 0, 0, d, span, nfeat, mis, o);
        }
        hipEventRecord(b, 0); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        double bytes = (double)templates * nfeat * 3072;
        printf("%-28s mis=%2d  %.2f us/launch  %.1f GB/s\n", name, mis, ms * 1e3 / reps, bytes / (ms * 1e-3 / reps) / 1e9);
    };
    for (int mis : {0, 1, 2, 3, 4, 8}) run("dword  (4 pos/lane)", 0, mis);
    for (int mis : {0, 1, 4, 8, 13}) run("dwordx4 (16 pos/lane)", 2, mis);
    return 0;
}
