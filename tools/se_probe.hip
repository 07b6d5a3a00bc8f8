// se_probe.hip — do all shader engines run the same instruction mix at the same speed?
// One 256-thread block per CU-slot runs small loops (VALU add, v_dot4, v_perm, LDS read b32, LDS write b16,
// s_barrier) and reports cycles (s_memtime) per loop; medians are grouped by (XCC, SE).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <map>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int NT = 10;
__global__ __launch_bounds__(256) void k_probe(unsigned long long* out, int iters)
{
    __shared__ uint32_t lds[4096];
    const int tid = threadIdx.x;
    for (int i = tid; i < 4096; i += 256) lds[i] = i * 2654435761u;
    __syncthreads();
    unsigned long long t[NT + 1];
    uint32_t a = tid, b = tid * 3 + 1, c = 7;
    t[0] = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) { a = a + b; b = b + c; c = c + a; a ^= b; }
    t[1] = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) { a = __builtin_amdgcn_udot4(a, b, c, false); b = __builtin_amdgcn_udot4(b, c, a, false); c = __builtin_amdgcn_udot4(c, a, b, false); a = __builtin_amdgcn_udot4(a, 0x01020304u, b, false); }
    t[2] = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) { a = __builtin_amdgcn_perm(a, b, 0x05040100u + (c & 1)); b = __builtin_amdgcn_alignbyte(b, c, 1); c = __builtin_amdgcn_perm(c, a, 0x07060302u); a += c; }
    t[3] = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) { a += lds[(tid + 36 * (i & 7) + (b & 3)) & 4095]; b += lds[(tid + 36 * ((i + 1) & 7)) & 4095]; c += lds[(tid + 72 + (a & 1)) & 4095]; a += lds[(tid + 108) & 4095]; }
    t[4] = __builtin_amdgcn_s_memtime();
    __syncthreads();
    for (int i = 0; i < iters; ++i) { ((uint16_t*)lds)[(tid + 72 * (i & 15)) & 8191] = (uint16_t)a; ((uint8_t*)lds)[(tid * 3 + 96 * (i & 15)) & 16383] = (uint8_t)b; a += i; b ^= a; }
    t[5] = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) { __syncthreads(); a += i; }
    t[6] = __builtin_amdgcn_s_memtime();
    // the vertical-pass mix: LDS reads feeding perms + dot2
    typedef unsigned short us2 __attribute__((ext_vector_type(2)));
    for (int i = 0; i < iters; ++i) {
        const uint32_t x = lds[(tid + 36 * (i & 7)) & 4095], y = lds[(tid + 36 * ((i & 7) + 1)) & 4095];
        a = __builtin_amdgcn_udot2(__builtin_bit_cast(us2, __builtin_amdgcn_perm(y, x, 0x05040100u)), __builtin_bit_cast(us2, 0x001c0008u), a, false);
        b = __builtin_amdgcn_udot2(__builtin_bit_cast(us2, __builtin_amdgcn_perm(y, x, 0x07060302u)), __builtin_bit_cast(us2, 0x001c0008u), b, false);
    }
    t[7] = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) { a = a * 6 + b; b = b * 4 + c; c = (c * a) + 1; }
    t[8] = __builtin_amdgcn_s_memtime();
    { // dependent s_memtime reads: latency of the SMEM time path
        unsigned long long x = t[8];
        for (int i = 0; i < 32; ++i) { unsigned long long y = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); x += y & 1; }
        a += (uint32_t)x;
        t[9] = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < 32; ++i) { unsigned long long y = __builtin_amdgcn_s_memrealtime(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); x += y & 1; }
        a += (uint32_t)x;
        t[10] = __builtin_amdgcn_s_memtime();
        t[9] = t[8] + (t[9] - t[8]) * (unsigned long long)iters / 32;   // scaled so that the /iters below gives per-read cycles
        t[10] = t[9] + (t[10] - (t[8] + (t[9] - t[8]) * 32 / (unsigned long long)iters)) * (unsigned long long)iters / 32;
    }
    if (a + b + c == 0x12345678u) lds[0] = a;
    if (tid == 0) {
        unsigned long long* o = out + (size_t)blockIdx.x * (NT + 2);
        for (int i = 0; i < NT; ++i) o[i] = t[i + 1] - t[i];
        o[NT] = __builtin_amdgcn_s_getreg(4 | (31 << 11));
        o[NT + 1] = __builtin_amdgcn_s_getreg(20 | (31 << 11));
    }
}
int main()
{
    const int nb = 2048, iters = 256;
    unsigned long long* d; CK(hipMalloc(&d, (size_t)nb * (NT + 2) * 8));
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k_probe, dim3(nb), dim3(256), 0, 0, d, iters);
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> h((size_t)nb * (NT + 2));
    CK(hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost));
    const char* names[NT] = {"valu", "dot4", "perm", "ldsrd", "ldswr16/8", "barrier", "lds+perm+dot2", "mul", "s_memtime", "s_memrealtime"};
    std::map<int, std::vector<std::vector<double>>> se;
    for (int b = 0; b < nb; ++b) {
        const unsigned long long* o = &h[(size_t)b * (NT + 2)];
        const int key = (int)((o[NT + 1] & 0xf) << 3 | ((o[NT] >> 13) & 7));
        auto& v = se[key]; v.resize(NT);
        for (int i = 0; i < NT; ++i) v[i].push_back((double)o[i] / iters);
    }
    printf("cycles per loop iteration (median over blocks), 8 blocks of 256 threads per CU\n%-8s", "xcc.se");
    for (int i = 0; i < NT; ++i) printf(" %14s", names[i]);
    printf("\n");
    for (auto& kv : se) {
        printf("%d.%d     ", kv.first >> 3, kv.first & 7);
        for (int i = 0; i < NT; ++i) { auto& v = kv.second[i]; std::sort(v.begin(), v.end()); printf(" %14.1f", v[v.size() / 2]); }
        printf("  (%zu blocks)\n", kv.second[0].size());
    }
    return 0;
}
