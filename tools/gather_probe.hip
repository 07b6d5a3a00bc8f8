// gather_probe.hip — rate of the refinement pass's patch reads (sbm_local_bits.h): a wave instruction fetches, for G features at
// unrelated addresses, 64 contiguous bytes each (16 rows x 4 bytes of a bit strip).  MODE 0: G = 4, one dword per lane (lane =
// slot, row); MODE 1: G = 8, 8 bytes per lane (lane = slot, row pair); MODE 2: G = 16, 16 bytes per lane (lane = slot, row quad).
// Same bytes per feature in all modes; what differs is the number of load instructions per feature.
// build: hipcc --offload-arch=gfx950 -O3 -o gather_probe tools/gather_probe.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
typedef uint32_t u2 __attribute__((ext_vector_type(2)));
typedef uint32_t u4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(64) void k(const uint32_t* __restrict__ buf, uint32_t span_dw, int features, uint32_t* out)
{
    constexpr int G = MODE == 0 ? 4 : MODE == 1 ? 8 : 16, LPG = 64 / G; // lanes per feature
    const int lane = threadIdx.x, g = lane / LPG, r = lane % LPG;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)buf, 0, 0x7fffffff, 0x00020000);
    uint32_t acc = 0, seed = (blockIdx.x * 64u + g) * 2654435761u + 12345u;
    for (int f = 0; f < features; f += 8 * G) { // 8 load instructions in flight
        uint32_t v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            seed = seed * 1664525u + 1013904223u;
            const uint32_t off = ((seed >> 7) % span_dw) * 4u + (uint32_t)r * (64u / LPG);
            if (MODE == 0) v[u] = __builtin_amdgcn_raw_buffer_load_b32(rs, (int)off, 0, 0);
            else if (MODE == 1) { const u2 w = __builtin_amdgcn_raw_buffer_load_b64(rs, (int)off, 0, 0); v[u] = w.x + w.y; }
            else { const u4 w = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, 0); v[u] = w.x + w.y + w.z + w.w; }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += v[u];
    }
    if (acc == 0x12345678u) out[0] = acc;
}
int main(int argc, char** argv)
{
    const size_t N = (argc > 1 ? (size_t)atoi(argv[1]) : 32u) << 20; // MiB; 32 = the bit strips of 16 frames of 1024 x 1024, 2 = one frame
    uint32_t *d, *o;
    hipMalloc(&d, N + 4096); hipMemset(d, 1, N + 4096); hipMalloc(&o, 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int waves = 6800, features = 128 * 16; // candidates x (features per candidate, 16 passes to get past the launch overhead)
    for (int rep = 0; rep < 2; ++rep)
        for (int mode = 0; mode < 3; ++mode) {
            auto launch = [&] {
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(waves), dim3(64), 0, 0, d, (uint32_t)(N / 4 - 64), features, o);
                else if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(waves), dim3(64), 0, 0, d, (uint32_t)(N / 4 - 64), features, o);
                else hipLaunchKernelGGL(k<2>, dim3(waves), dim3(64), 0, 0, d, (uint32_t)(N / 4 - 64), features, o);
            };
            launch(); hipDeviceSynchronize();
            hipEventRecord(a, 0);
            for (int i = 0; i < 10; ++i) launch();
            hipEventRecord(b, 0); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            const double us = ms * 1e3 / 10, feats = (double)waves * features;
            printf("mode %d (%2d bytes per lane): %7.1f us per launch, %5.2f ns per 1000 features, %6.1f GB/s, %5.1f G load instr/s\n", mode,
                   mode == 0 ? 4 : mode == 1 ? 8 : 16, us, us * 1e3 / feats * 1e3, feats * 64 / us / 1e3, feats / (mode == 0 ? 4 : mode == 1 ? 8 : 16) / us / 1e3);
        }
    return 0;
}
