// tools/mfma_gauss_probe.hip — keep-or-kill measurement for VERDICT round 2, item 2c: the horizontal pass of the 7x7
// Gaussian of the row-streaming gradient kernel (sbm_quantize_stream.h: 18 v_dot2_u32_u16 + 4 DPP moves per channel and
// row) on the matrix pipe instead of the vector ALU.
//
// Variant V (what the kernel does today): per channel g = (E, O) u16 pairs of 4 pixels per lane; neighbours by DPP; 18
// dot2; smoothed bytes = bits 16..23 of the four sums.
// Variant M: the 256-pixel strip row goes through LDS into the A operand of v_mfma_i32_16x16x64_i8 -- M index = one of 16
// segments of 16 pixels, K = the 64 bytes (32 u16 values, low and high bytes interleaved as they lie in memory) of the
// segment's 32-pixel window, N = the 16 output pixels of the segment; two MFMAs (the weights against the high bytes,
// shifted left 8 by one v_lshlrev per accumulator register, then the weights against the low bytes on top), bias for the
// signed-byte arithmetic and the rounding constant folded into the first MFMA's addend; the smoothed byte (bits 16..23 of
// the accumulator) leaves through ds_write_b8_d16_hi and comes back as one dword per lane in the streaming layout.
// Both variants end with the same two v_perm per channel (bytes -> u16 pairs) and run beside FILL other vector
// instructions per row, standing in for the rest of the row's work (the kernel is bound by vector issue).
//
// build: hipcc --offload-arch=gfx950 -O3 -w -mllvm -amdgpu-mfma-vgpr-form=1 -o tools/bin/mfma_gauss_probe tools/mfma_gauss_probe.hip
//        (the flag keeps the MFMA accumulators in VGPRs: without it the compiler parks them in AGPRs and pays four
//        v_accvgpr_read + four v_accvgpr_write per channel and row for the shift between the two MFMAs)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef int v4i __attribute__((ext_vector_type(4)));
typedef unsigned short us2 __attribute__((ext_vector_type(2)));
#define K2(lo, hi) ((uint32_t)(lo) | ((uint32_t)(hi) << 16))

__device__ __forceinline__ uint32_t udot2(uint32_t a, uint32_t b, uint32_t c)
{
    return __builtin_amdgcn_udot2(__builtin_bit_cast(us2, a), __builtin_bit_cast(us2, b), c, false);
}
__device__ __forceinline__ uint32_t from_left(uint32_t x) { return (uint32_t)__builtin_amdgcn_mov_dpp((int)x, 0x138, 0xf, 0xf, true); }
__device__ __forceinline__ uint32_t from_right(uint32_t x) { return (uint32_t)__builtin_amdgcn_mov_dpp((int)x, 0x130, 0xf, 0xf, true); }
__device__ __forceinline__ uint32_t perm(uint32_t hi, uint32_t lo, uint32_t sel) { return __builtin_amdgcn_perm(hi, lo, sel); }

// the kernel's horizontal pass, one channel: returns the four smoothed bytes of the lane as one dword (p0 p1 p2 p3)
__device__ __forceinline__ uint32_t hpass_valu(uint32_t E, uint32_t O)
{
    const uint32_t pE = from_left(E), pO = from_left(O), nE = from_right(E), nO = from_right(O);
    uint32_t a0 = udot2(pO, K2(8, 56), 32768u);
    a0 = udot2(O, K2(56, 8), a0);
    a0 = udot2(pE, K2(0, 28), a0);
    a0 = udot2(E, K2(72, 28), a0);
    uint32_t a1 = udot2(pE, K2(0, 8), 32768u);
    a1 = udot2(E, K2(56, 56), a1);
    a1 = udot2(nE, K2(8, 0), a1);
    a1 = udot2(pO, K2(0, 28), a1);
    a1 = udot2(O, K2(72, 28), a1);
    uint32_t a2 = udot2(pO, K2(0, 8), 32768u);
    a2 = udot2(O, K2(56, 56), a2);
    a2 = udot2(nO, K2(8, 0), a2);
    a2 = udot2(E, K2(28, 72), a2);
    a2 = udot2(nE, K2(28, 0), a2);
    uint32_t a3 = udot2(E, K2(8, 56), 32768u);
    a3 = udot2(nE, K2(56, 8), a3);
    a3 = udot2(O, K2(28, 72), a3);
    a3 = udot2(nO, K2(28, 0), a3);
    // bytes 2 of a0, a1, a2, a3 -> one dword (two perms, as the kernel's SE / SO extraction)
    return perm(a1, a0, 0x0c0c0602u) | (perm(a3, a2, 0x0c0c0602u) << 16);
}

constexpr int ROW_G = 16 + 512 + 16; // bytes of one channel's g row in LDS: 8 halo pixels of zeros on either side
constexpr int ROW_S = 256;           // smoothed bytes
constexpr int WAVE_LDS = 3 * ROW_G + 3 * ROW_S;

// B operands (weights) of the two MFMAs for lane l: column n = l & 15, k = 16 (l >> 4) + j, j = 0 .. 15.
// Byte k of a segment's window: pixel w(k) = 8 (k / 16) + 4 ((k % 16) / 8) + {0, 2, 1, 3}[(k % 8) / 2], high byte iff k odd.
__device__ __forceinline__ void make_weights(int lane, v4i& b_lo, v4i& b_hi)
{
    const int W7[7] = {8, 28, 56, 72, 56, 28, 8};
    const int n = lane & 15, q = lane >> 4;
    uint32_t lo[4] = {0, 0, 0, 0}, hi[4] = {0, 0, 0, 0};
    for (int j = 0; j < 16; ++j) {
        const int k = 16 * q + j;
        const int off[4] = {0, 2, 1, 3};
        const int w = 8 * (k / 16) + 4 * ((k % 16) / 8) + off[(k % 8) / 2];
        const int t = w - (n + 5); // window pixel w is pixel 16 m - 8 + w; output pixel 16 m + n uses w in [n + 5, n + 11]
        const uint32_t wt = (t >= 0 && t < 7) ? (uint32_t)W7[t] : 0u;
        if (k & 1) hi[j >> 2] |= wt << (8 * (j & 3));
        else lo[j >> 2] |= wt << (8 * (j & 3));
    }
    b_lo = v4i{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3]};
    b_hi = v4i{(int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
}

// one channel through the matrix pipe.  lds_g / lds_s: this wave's row buffers for the channel.
__device__ __forceinline__ uint32_t hpass_mfma(uint32_t E, uint32_t O, uint8_t* lds_g, uint8_t* lds_s, int lane, const v4i& b_lo, const v4i& b_hi,
                                               const v4i& c0)
{
    // bytes as signed: x ^ 0x80 = x - 128 (the constant part is in c0)
    *(uint2*)(lds_g + 16 + 8 * lane) = uint2{E ^ 0x80808080u, O ^ 0x80808080u};
    const int m = lane & 15, q = lane >> 4;
    const v4i a = *(const v4i*)(lds_g + 32 * m + 16 * q); // window bytes 16 q .. 16 q + 15 of segment m (16-byte aligned)
    v4i acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b_hi, c0, 0, 0, 0);
    acc = acc << 8;
    acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b_lo, acc, 0, 0, 0);
    // D layout: lane (n = lane & 15, q), register j: segment 4 q + j, pixel 16 (4 q + j) + n; smoothed byte = bits 16..23
    uint8_t* d = lds_s + 64 * q + (lane & 15);
    d[0] = (uint8_t)((uint32_t)acc.x >> 16);
    d[16] = (uint8_t)((uint32_t)acc.y >> 16);
    d[32] = (uint8_t)((uint32_t)acc.z >> 16);
    d[48] = (uint8_t)((uint32_t)acc.w >> 16);
    return *(const uint32_t*)(lds_s + 4 * lane);
}

template <int MODE, int FILL>
__global__ __launch_bounds__(256) void k_probe(const uint32_t* __restrict__ src, uint32_t* __restrict__ out, int rows, int check)
{
    extern __shared__ uint8_t lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint8_t* my = lds + wave * WAVE_LDS;
    for (int i = lane; i < WAVE_LDS / 4; i += 64) ((uint32_t*)my)[i] = 0x80808080u; // halo = "0" in the signed encoding
    v4i b_lo, b_hi;
    make_weights(lane, b_lo, b_hi);
    // c0: rounding 2^15 and the -128 bias of both byte planes (sum of weights 256): ((x + 32768 + 256) << 8) = 256 hi + 65536
    const v4i c0 = v4i{33024, 33024, 33024, 33024};
    const size_t gw = (size_t)blockIdx.x * 4 + wave;
    uint32_t sum = 0, f[8];
    for (int i = 0; i < 8; ++i) f[i] = lane * 0x9e3779b9u + i;
    for (int r = 0; r < rows; ++r) {
        uint32_t res[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            // vertical-pass results stand-in: two u16 pairs <= 65280 from memory
            const uint32_t v = src[(gw * 64 + lane + (size_t)(r * 3 + c) * 977) & 0xfffff];
            const uint32_t E = (v & 0xfeff) | ((v >> 3) & 0xfeff0000u), O = ((v >> 5) & 0xfeff) | ((v << 7) & 0xfeff0000u);
            if (MODE == 0) res[c] = hpass_valu(E, O);
            else if (MODE == 1) res[c] = hpass_mfma(E, O, my + c * ROW_G, my + 3 * ROW_G + c * ROW_S, lane, b_lo, b_hi, c0);
            else res[c] = E ^ O; // MODE 2: neither (the filler alone)
            if (check) out[((gw * rows + r) * 3 + c) * 64 + lane] = res[c];
        }
        // the rest of the row: FILL vector instructions of the kernel's half-rate kind, 8 independent chains
#pragma unroll
        for (int i = 0; i < FILL; ++i) f[i & 7] = perm(f[i & 7], res[i % 3], 0x06010704u + (uint32_t)i);
        sum += perm(res[1], res[0], 0x0c020c00u) + perm(res[2], res[0], 0x0c030c01u);
    }
    for (int i = 0; i < 8; ++i) sum += f[i];
    if (!check) out[gw * 64 + lane] = sum;
}

template <int MODE, int FILL>
static float run(const uint32_t* d_src, uint32_t* d_out, int blocks, int rows, int lds_bytes, int reps)
{
    hipFuncSetAttribute((const void*)k_probe<MODE, FILL>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    hipLaunchKernelGGL((k_probe<MODE, FILL>), dim3(blocks), dim3(256), lds_bytes, 0, d_src, d_out, rows, 0);
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k_probe<MODE, FILL>), dim3(blocks), dim3(256), lds_bytes, 0, d_src, d_out, rows, 0);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    return ms / reps * 1e3f; // us per launch
}

int main()
{
    const int n_src = 1 << 20;
    std::vector<uint32_t> h(n_src);
    uint32_t x = 12345;
    for (auto& v : h) v = x = x * 1664525u + 1013904223u;
    uint32_t *d_src, *d_out;
    hipMalloc(&d_src, n_src * 4);
    hipMemcpy(d_src, h.data(), n_src * 4, hipMemcpyHostToDevice);
    // correctness: both variants on the same data, every useful pixel (lanes 1 .. 62: pixels 4 .. 251) identical
    const int cb = 8, crows = 16;
    const size_t n_chk = (size_t)cb * 4 * crows * 3 * 64;
    hipMalloc(&d_out, std::max<size_t>(n_chk, (size_t)4096 * 256) * 4);
    std::vector<uint32_t> o0(n_chk), o1(n_chk);
    const int lds_small = 4 * WAVE_LDS;
    hipFuncSetAttribute((const void*)k_probe<1, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_small);
    hipLaunchKernelGGL((k_probe<0, 0>), dim3(cb), dim3(256), lds_small, 0, d_src, d_out, crows, 1);
    hipMemcpy(o0.data(), d_out, n_chk * 4, hipMemcpyDeviceToHost);
    hipLaunchKernelGGL((k_probe<1, 0>), dim3(cb), dim3(256), lds_small, 0, d_src, d_out, crows, 1);
    hipMemcpy(o1.data(), d_out, n_chk * 4, hipMemcpyDeviceToHost);
    size_t bad = 0, first = 0;
    for (size_t i = 0; i < n_chk; ++i) {
        const int lane = (int)(i & 63);
        if (lane < 1 || lane > 62) continue; // the strip's first and last lane see the strip edge differently (halo lanes in the kernel)
        if (o0[i] != o1[i] && !bad++) first = i;
    }
    printf("check: %zu of %zu useful dwords differ between the VALU and the MFMA horizontal pass%s\n", bad, n_chk, bad ? " (FAIL)" : " (identical)");
    if (bad) printf("  first at %zu: valu %08x mfma %08x\n", first, o0[first], o1[first]);
    // timing: 768 workgroups x 4 waves = 3 waves per SIMD (the gradient kernel's residency; LDS padding caps it), 512 rows per wave
    const int blocks = 768, rows = 512, lds_cap = 52 * 1024;
#define TRIPLE(FILL_)                                                                                                               \
    {                                                                                                                               \
        const float tv = run<0, FILL_>(d_src, d_out, blocks, rows, lds_cap, 5), tm = run<1, FILL_>(d_src, d_out, blocks, rows, lds_cap, 5), \
                    tf = run<2, FILL_>(d_src, d_out, blocks, rows, lds_cap, 5);                                                       \
        printf("FILL %3d other vector instructions per row: VALU pass %7.1f us  MFMA pass %7.1f us  neither %7.1f us  -> per wave-row: "  \
               "%.0f / %.0f / %.0f ns; MFMA / VALU = %.3f\n",                                                                       \
               FILL_, tv, tm, tf, tv * 1e3 / rows, tm * 1e3 / rows, tf * 1e3 / rows, tm / tv);                                      \
    }
    TRIPLE(0)
    TRIPLE(120)
    TRIPLE(240)
    TRIPLE(340)
    return bad ? 1 : 0;
}
