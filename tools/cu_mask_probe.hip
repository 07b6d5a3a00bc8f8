// cu_mask_probe.hip — production k_quantize (no stamps) pinned to one CU at a time through a CU-masked
// stream: is a tile slower on some shader engines?  Prints the per-launch time of a 4-tile image by (XCC, SE).
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>
#include "../shape_based_matching_amd/csrc/sbm_kernels.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
using namespace sbm;
#ifndef PROBE_QN
#define PROBE_QN QN_LATENCY
#endif
__global__ void k_whoami(unsigned* out) { out[0] = __builtin_amdgcn_s_getreg(4 | (31 << 11)); out[1] = __builtin_amdgcn_s_getreg(20 | (31 << 11)); }
int main()
{
    const int rows = 64, cols = 64 * 3; // 4 x 3 = 12 tiles, all on the one CU
    std::vector<uint8_t> img((size_t)rows * cols * 3);
    srand(7);
    for (auto& b : img) b = (uint8_t)(rand() >> 7);
    uint8_t *d_img, *d_out, *d_pyr; unsigned* d_id;
    CK(hipMalloc(&d_img, img.size())); CK(hipMalloc(&d_out, (size_t)rows * cols)); CK(hipMalloc(&d_pyr, img.size())); CK(hipMalloc(&d_id, 8));
    CK(hipMemcpy(d_img, img.data(), img.size(), hipMemcpyHostToDevice));
    const dim3 grid((cols + QT_C - 1) / QT_C, (rows + QT_R - 1) / QT_R);
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    std::map<int, std::vector<double>> se;
    for (int bit = 0; bit < 256; ++bit) {
        uint32_t mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        mask[bit >> 5] = 1u << (bit & 31);
        hipStream_t s;
        if (hipExtStreamCreateWithCUMask(&s, 8, mask) != hipSuccess) { printf("mask bit %d refused\n", bit); continue; }
        auto launch = [&] { hipLaunchKernelGGL((k_quantize<3, false, PROBE_QN>), grid, dim3(PROBE_QN), 0, s, d_img, rows, cols, cols * 3, (const uint8_t*)nullptr, 900.f, d_out, (float*)nullptr, (float*)nullptr, d_pyr, (int64_t)0, (int64_t)0, (int64_t)0); };
        hipLaunchKernelGGL(k_whoami, dim3(1), dim3(64), 0, s, d_id);
        for (int i = 0; i < 5; ++i) launch();
        CK(hipEventRecord(a, s));
        for (int i = 0; i < 20; ++i) launch();
        CK(hipEventRecord(b, s)); CK(hipStreamSynchronize(s));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        unsigned id[2]; CK(hipMemcpy(id, d_id, 8, hipMemcpyDeviceToHost));
        se[(int)((id[1] & 0xf) << 3 | ((id[0] >> 13) & 7))].push_back(ms * 1e3 / 20);
        CK(hipStreamDestroy(s));
    }
    printf("12 tiles on one CU, us per launch, median by (xcc.se) [n CUs]:\n");
    for (auto& kv : se) { auto& v = kv.second; std::sort(v.begin(), v.end()); printf(" %d.%d:%.1f[%zu]", kv.first >> 3, kv.first & 7, v[v.size() / 2], v.size()); }
    printf("\n");
    return 0;
}
