// sbm_quantize_tile.h — gfx950 kernels of the gradient stage in 16 x 64 LDS tiles (single frames, float outputs, any width) and the
// stage kernels around it.  Reference functions replaced (file:line in ddcr/shape_based_matching):
//   k_quantize            quantizedOrientations + hysteresisGradient   line2Dup.cpp:313-404, 218-311
//   k_orientation_bins    the 16-bin rule of hysteresisGradient        line2Dup.cpp:225
//   k_pyrdown             cv::pyrDown in ColorGradientPyramid::pyrDown line2Dup.cpp:431-433
//   k_resize_linear_u8    cv::resize of shapeInfo_producer::transform  line2Dup.h:379-405
//   k_resize_mask         resize(mask, INTER_NEAREST)                  line2Dup.cpp:439
// (the batch path's gradient kernel is sbm_quantize_stream.h)
#pragma once
#include "sbm_common.h"

namespace sbm {

// ---------------------------------------------------------------------------
// Gradient stage
// ---------------------------------------------------------------------------
constexpr int QT_R = 16;          // output tile rows
constexpr int QT_C = 64;          // output tile cols
constexpr int QS_R = QT_R + 10;   // source tile (halo 5: 3 gauss + 1 sobel + 1 vote)
constexpr int QS_C = QT_C + 10;
constexpr int QM_R = QT_R + 4;    // smoothed tile (halo 2)
constexpr int QM_C = QT_C + 4;
constexpr int QQ_R = QT_R + 2;    // quantised tile (halo 1)
constexpr int QQ_C = QT_C + 2;

// fastAtan2 polynomial in degrees (OpenCV atan_f32); every operation rounded.
__device__ __forceinline__ float fast_atan2_deg(float y, float x)
{
    const float p1 = 0.9997878412794807f * (float)(180 / 3.14159265358979323846);
    const float p3 = -0.3258083974640975f * (float)(180 / 3.14159265358979323846);
    const float p5 = 0.1555786518463281f * (float)(180 / 3.14159265358979323846);
    const float p7 = -0.04432655554792128f * (float)(180 / 3.14159265358979323846);
    const float eps = 2.2204460492503131e-16f; // (float)DBL_EPSILON
    float ax = fabsf(x), ay = fabsf(y);
    float a, c, c2;
    if (ax >= ay) {
        c = __fdiv_rn(ay, __fadd_rn(ax, eps));
        c2 = __fmul_rn(c, c);
        a = __fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(p7, c2), p5), c2), p3), c2), p1), c);
    } else {
        c = __fdiv_rn(ax, __fadd_rn(ay, eps));
        c2 = __fmul_rn(c, c);
        a = __fsub_rn(90.f, __fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(p7, c2), p5), c2), p3), c2), p1), c));
    }
    if (x < 0) a = __fsub_rn(180.f, a);
    if (y < 0) a = __fsub_rn(360.f, a);
    return a;
}

// 16-bin orientation index of an integer gradient, equal for every |gx|, |gy| <= 1020 (the range of a 3x3
// Sobel on 8-bit data) to saturate_cast<uchar>(cvRound(fastAtan2(gy, gx) * 16/360)) of line2Dup.cpp:225:
// per octant the float pipeline is a step function of min/max with the same two thresholds, which lie in
// (182/915, 73/367] and (661/989, 264/395] (tools/derive_orientation_thresholds.py scans all pairs; the
// parity tests check all 2041^2 of them on the GPU).  ~15 integer ops instead of a division + polynomial.
__device__ __forceinline__ int orientation_bin16(int gx, int gy)
{
    const int ax = gx < 0 ? -gx : gx, ay = gy < 0 ? -gy : gy;
    const int mx = ax > ay ? ax : ay, mn = ax > ay ? ay : ax;
    int k = 0;
    if (mx > 0) k = (__mul24(mn, 367) >= __mul24(73, mx) ? 1 : 0) + (__mul24(mn, 395) >= __mul24(264, mx) ? 1 : 0);
    if (ay > ax) k = 4 - k;
    if (gx < 0) k = 8 - k;
    if (gy < 0) k = 16 - k;
    return k;
}

__global__ __launch_bounds__(256) void k_orientation_bins(const int16_t* __restrict__ gx, const int16_t* __restrict__ gy,
                                                          int64_t n, uint8_t* __restrict__ q16)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) q16[i] = (uint8_t)orientation_bin16(gx[i], gy[i]);
}

// ---------------------------------------------------------------------------
// k_quantize: one block = one 16 x 64 output tile, 4 pixels per lane in every
// phase, planar (per-channel) LDS images read and written as dwords.
//   A  source tile (halo 5, origin C0-8 so that columns are dword aligned) ->
//      LDS, channels de-interleaved with v_perm; border tiles take the clamped
//      (BORDER_REPLICATE) byte path.
//   P  optional: cv::pyrDown of the tile (the next pyramid level's source
//      image) straight from the LDS tile.
//   B  horizontal 7-tap {8,28,56,72,56,28,8}: two v_dot4_u32_u8 per output,
//      windows assembled with v_alignbyte; exact 8.8 fixed point (u16).
//   C  vertical 7-tap on the u16 rows with v_dot2_u32_u16, one rounding
//      (sum + 2^15) >> 16 -> u8 smoothed tile; border tiles then replicate the
//      smoothed edge outwards (that is Sobel's BORDER_REPLICATE).
//   D  Sobel 3x3 per channel, channel of maximum magnitude, fastAtan2, 16 -> 8
//      orientation bins, 'strong' flag (mag > weak^2).
//   E  3x3 majority vote (>= 5 of 9) around strong pixels -> one-hot byte.
// ---------------------------------------------------------------------------
// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains vmcnt, i.e. every wave would
// sit at the barrier until its outstanding GLOBAL stores are acknowledged; the tiles below exchange data
// through LDS alone, so lgkmcnt(0) is the whole requirement.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Diagnostic build only (tools/quantize_probe.hip defines SBM_QSTAMP): thread 0 of every tile stores
// s_memtime (and s_memrealtime at entry / exit) at the phase boundaries into a buffer nothing else reads.
#ifdef SBM_QSTAMP
__device__ unsigned long long* g_qstamp = nullptr;
#ifdef SBM_QSTAMP_LIGHT /* entry and exit only */
#define QSTAMP_ON(i) ((i) == 0 || (i) == 6)
#else
#define QSTAMP_ON(i) true
#endif
#define QSTAMP(i)                                                                               \
    if (QSTAMP_ON(i) && threadIdx.x == 0 && g_qstamp) {                                         \
        unsigned long long* q_ = g_qstamp + (size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 64; \
        if (i == 0 || i == 6) q_[i] = __builtin_amdgcn_s_memrealtime();                         \
        q_[8 + i] = __builtin_amdgcn_s_memtime();                                               \
        if (i == 0) {                                                                           \
            q_[16] = __builtin_amdgcn_s_getreg(4 | (31 << 11));                                 \
            q_[17] = __builtin_amdgcn_s_getreg(20 | (31 << 11));                                \
        }                                                                                       \
    }
#define QSTAMP_WAVES(base)                                        \
    if (QSTAMP_ON(1) && (threadIdx.x & 63) == 0 && g_qstamp)      \
        g_qstamp[(size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 64 + (base) + (threadIdx.x >> 6)] = __builtin_amdgcn_s_memtime();
#else
#define QSTAMP(i)
#define QSTAMP_WAVES(base)
#endif
// Threads per tile.  1024 (16 waves: short per-wave instruction streams, the waves hide each other's latency)
// when a launch gives a CU one or two tiles -- tile latency is then the launch's duration; 512 when it gives
// every CU many (a batch of frames): four tiles per CU instead of two overlap each other's barrier and LDS
// stalls (measured: 4096 textured tiles 52.8 -> 46.0 us, 256 tiles 6.0 -> 7.0 us).
constexpr int QN_LATENCY = 1024, QN_THROUGHPUT = 512;
constexpr int QS_W = 80; // source tile width  (cols C0-8 .. C0+71)
constexpr int QH_W = 72; // h / smoothed / q tile width (cols C0-4 .. C0+67)

typedef unsigned short us2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t udot2(uint32_t a, uint32_t b, uint32_t c)
{
    return __builtin_amdgcn_udot2(__builtin_bit_cast(us2_t, a), __builtin_bit_cast(us2_t, b), c, false);
}

typedef short ss2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ int sdot2(uint32_t a, uint32_t b, int c)
{
    return __builtin_amdgcn_sdot2(__builtin_bit_cast(ss2_t, a), __builtin_bit_cast(ss2_t, b), c, false);
}
// packed 2 x 16-bit lanes in a dword (wrap-around; the same bits for signed and unsigned operands)
__device__ __forceinline__ uint32_t pk_add(uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, (us2_t)(__builtin_bit_cast(us2_t, a) + __builtin_bit_cast(us2_t, b))); }
__device__ __forceinline__ uint32_t pk_sub(uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, (us2_t)(__builtin_bit_cast(us2_t, a) - __builtin_bit_cast(us2_t, b))); }
__device__ __forceinline__ uint32_t pk_2a_plus_b(uint32_t a, uint32_t b)
{
    const us2_t two = {2, 2};
    return __builtin_bit_cast(uint32_t, (us2_t)(__builtin_bit_cast(us2_t, a) * two + __builtin_bit_cast(us2_t, b)));
}

// Vote word 1 << (4 * label) of a packed integer gradient v = (gx & 0xffff) | gy << 16, label = the
// 8-bin orientation (orientation_bin16(gx, gy) & 7).  In the first quadrant the 16-bin index is the number
// of bin boundaries below the direction, four sign tests of linear forms in (|gx|, |gy|) (same two
// thresholds as orientation_bin16, mirrored about the diagonal); the other quadrants map k -> (8-k)&7 when
// exactly one of gx, gy is negative.  A zero gradient is bin 0.
__device__ __forceinline__ uint32_t vote_word(uint32_t v)
{
    const ss2_t sv = __builtin_bit_cast(ss2_t, v);
    const uint32_t av = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(sv, (ss2_t)(-sv))); // (|gx|, |gy|)
    const uint32_t u1 = (uint32_t)sdot2(av, (uint32_t)(uint16_t)(-73) | (367u << 16), 0);   // 367|gy| -  73|gx| >= 0
    const uint32_t u2 = (uint32_t)sdot2(av, (uint32_t)(uint16_t)(-264) | (395u << 16), 0);  // 395|gy| - 264|gx| >= 0
    const uint32_t u3 = (uint32_t)sdot2(av, (uint32_t)(uint16_t)(-395) | (264u << 16), -1); // 264|gy| - 395|gx| >  0
    const uint32_t u4 = (uint32_t)sdot2(av, (uint32_t)(uint16_t)(-367) | (73u << 16), -1);  //  73|gy| - 367|gx| >  0
    uint32_t neg = u1 >> 31; // collect the four sign bits
    neg = __builtin_amdgcn_alignbit(neg, u2, 31);
    neg = __builtin_amdgcn_alignbit(neg, u3, 31);
    neg = __builtin_amdgcn_alignbit(neg, u4, 31);
    const int sh = 16 - 4 * __builtin_popcount(neg);          // 4 * k, k = 0..4
    const int m = (int)((v << 16) ^ v) >> 31;                   // -1 iff exactly one component is negative
    const uint32_t w = 1u << (((sh ^ m) - m) & 31);             // k -> (8 - k) & 7
    return v ? w : 1u;
}

template <int CH, bool WITH_FLOAT, int QN>
__global__ __launch_bounds__(QN) void k_quantize(const uint8_t* __restrict__ img, int rows, int cols,
                                                  int stride, const uint8_t* __restrict__ mask,
                                                  float thr_sq, uint8_t* __restrict__ out,
                                                  float* __restrict__ mag_out, float* __restrict__ ori_out,
                                                  uint8_t* __restrict__ pyr_out, int64_t img_fs, int64_t out_fs,
                                                  int64_t pyr_fs, int tile_row0)
{
    // a batch of frames of one geometry: frame = blockIdx.z, *_fs = bytes from one frame to the next
    img += (size_t)blockIdx.z * img_fs;
    out += (size_t)blockIdx.z * out_fs;
    if (pyr_out) pyr_out += (size_t)blockIdx.z * pyr_fs;
    __shared__ uint32_t s_src[CH][QS_R][QS_W / 4]; // u8 x4
    __shared__ __attribute__((aligned(16))) uint32_t s_h[CH][QS_R][QH_W / 2]; // u16 x2
    __shared__ uint32_t s_sm[CH][QM_R][QH_W / 4];  // u8 x4
    __shared__ __attribute__((aligned(16))) uint32_t s_w[QQ_R][QH_W]; // vote word 1 << 4*label per pixel
    __shared__ uint32_t s_st[QQ_R][QH_W / 4];      // u8 x4: non-zero = magnitude above the weak threshold
    __shared__ uint32_t s_pyr[QT_R / 2][QT_C / 2 * CH / 4]; // pyrDown of the tile, interleaved channels, as dwords
    __shared__ __attribute__((aligned(16))) uint32_t s_nonflat[16]; // per wave: it loaded a word that differs from the tile's first pixel
    const int tid = threadIdx.x;
    // tile_row0: first tile row of this launch (a launch may cover a band of tile rows: the host entry point starts the
    // gradient stage on the rows that have arrived while the rest of the frame is still crossing PCIe)
    const int R0 = ((int)blockIdx.y + tile_row0) * QT_R, C0 = blockIdx.x * QT_C;
    const bool interior = R0 >= 5 && R0 + QT_R + 5 <= rows && C0 >= 8 && C0 + QT_C + 8 <= cols;
    QSTAMP(0)
    // every kernel argument is fetched here, with the first ones: a scalar load issued in the middle of the
    // tile costs a scalar-cache round trip on the critical path of its phase
    asm volatile("" ::"s"(thr_sq), "s"(out), "s"(mask), "s"(pyr_out), "s"(mag_out), "s"(ori_out), "s"(stride));

    // ---- A: source tile -> planar LDS (rows clamped; a 4-pixel group that sticks out of the image
    //         left or right takes the per-byte clamped path = BORDER_REPLICATE) ----
    const bool whole_groups = cols >= 4 && (cols & 3) == 0; // 4-pixel groups lie entirely inside or outside the image
    // flat-tile test, folded into the load: every word of the tile against the tile's first pixel
    const bool flat_check = !WITH_FLOAT && thr_sq >= 0.f;
    uint32_t ref[CH];
    bool nonflat = false;
    if (flat_check) {
        const uint8_t* p0 = img + (size_t)clampi(R0 - 5, 0, rows - 1) * stride + (size_t)clampi(C0 - 8, 0, cols - 1) * CH;
#pragma unroll
        for (int k = 0; k < CH; ++k) ref[k] = (uint32_t)p0[k] * 0x01010101u;
    }
    constexpr int A_ITEMS = QS_R * (QS_W / 4), A_ROUNDS = (A_ITEMS + QN - 1) / QN;
    if (whole_groups) {
        // A group outside the image replicates the first / last pixel of the row: load the nearest inside group and
        // broadcast its edge byte (one memory round trip for every lane, no divergent byte path).  The loads of every
        // round are issued before any of them is consumed: 520 items on 512 threads must not cost two round trips.
        uint32_t d[A_ROUNDS][CH == 1 ? 1 : 3];
#pragma unroll
        for (int rd = 0; rd < A_ROUNDS; ++rd) {
            const int it = tid + rd * QN;
            if (it < A_ITEMS) {
                const int r = it / (QS_W / 4), g = it - r * (QS_W / 4);
                const uint8_t* p = img + (size_t)clampi(R0 - 5 + r, 0, rows - 1) * stride + (size_t)clampi(C0 - 8 + 4 * g, 0, cols - 4) * CH;
                d[rd][0] = ld_u32_any(p);
                if (CH > 1) {
                    d[rd][1 % (CH == 1 ? 1 : 3)] = ld_u32_any(p + 4);
                    d[rd][2 % (CH == 1 ? 1 : 3)] = ld_u32_any(p + 8);
                }
            }
        }
#pragma unroll
        for (int rd = 0; rd < A_ROUNDS; ++rd) {
            const int it = tid + rd * QN;
            if (it < A_ITEMS) {
                const int r = it / (QS_W / 4), g = it - r * (QS_W / 4);
                const int c = C0 - 8 + 4 * g;
                uint32_t w[CH];
                if (CH == 1) {
                    w[0] = d[rd][0];
                } else {
                    const uint32_t d0 = d[rd][0], d1 = d[rd][1 % (CH == 1 ? 1 : 3)], d2 = d[rd][2 % (CH == 1 ? 1 : 3)];
                    // 12 interleaved bytes b0..b11 -> channel k = {b[k], b[k+3], b[k+6], b[k+9]}
                    w[0] = __builtin_amdgcn_perm(d2, __builtin_amdgcn_perm(d1, d0, 0x0c060300u), 0x05020100u);
                    w[1 % CH] = __builtin_amdgcn_perm(d2, __builtin_amdgcn_perm(d1, d0, 0x0c070401u), 0x06020100u);
                    w[2 % CH] = __builtin_amdgcn_perm(d2, __builtin_amdgcn_perm(d1, d0, 0x0c0c0502u), 0x07040100u);
                }
#pragma unroll
                for (int k = 0; k < CH; ++k) {
                    if (c < 0) w[k] = (w[k] & 0xffu) * 0x01010101u;
                    if (c >= cols) w[k] = (w[k] >> 24) * 0x01010101u;
                    s_src[k][r][g] = w[k];
                    if (flat_check) nonflat = nonflat || w[k] != ref[k];
                }
            }
        }
    } else {
        for (int it = tid; it < A_ITEMS; it += QN) {
            const int r = it / (QS_W / 4), g = it - r * (QS_W / 4);
            const int c = C0 - 8 + 4 * g;
            const uint8_t* rowp = img + (size_t)clampi(R0 - 5 + r, 0, rows - 1) * stride;
#pragma unroll
            for (int k = 0; k < CH; ++k) {
                uint32_t w = 0;
#pragma unroll
                for (int m = 0; m < 4; ++m) w |= (uint32_t)rowp[clampi(c + m, 0, cols - 1) * CH + k] << (8 * m);
                s_src[k][r][g] = w;
                if (flat_check) nonflat = nonflat || w != ref[k];
            }
        }
    }
    if (flat_check) {
        const bool wave_nonflat = __builtin_amdgcn_ballot_w64(nonflat) != 0;
        if ((tid & 63) == 0) s_nonflat[tid >> 6] = wave_nonflat ? 1u : 0u;
    }
    lds_barrier();
    QSTAMP(1)

    // ---- flat-tile shortcut: if every channel of the source tile is constant, every gradient is 0,
    //      no pixel passes `mag > weak^2` and the whole output tile is 0 (padding / saturated background:
    //      the reference's own demo pads the frame with 250 black pixels, test.cpp:344-347); the pyrDown of
    //      a constant tile is the same constant.  WITH_FLOAT keeps the full path so that magnitude / angle
    //      are written everywhere. ----
    if (flat_check) {
        uint32_t any = 0;
#pragma unroll
        for (int w4 = 0; w4 < QN / 64 / 4; ++w4) {
            const uint4 f = *(const uint4*)&s_nonflat[4 * w4];
            any |= f.x | f.y | f.z | f.w;
        }
        if (any == 0) {
            const int er = tid >> 4, g = tid & 15;
            const int r = R0 + er;
            if (tid < 256 && r < rows) {
                uint8_t* o = out + (size_t)r * cols + C0 + 4 * g;
                if (C0 + 4 * g + 3 < cols && ((((size_t)r * cols + C0 + 4 * g) & 3) == 0)) {
                    *(uint32_t*)o = 0;
                } else {
                    for (int m = 0; m < 4; ++m)
                        if (C0 + 4 * g + m < cols) o[m] = 0;
                }
            }
            const int ps = tid - 256;
            if (pyr_out && ps >= 0 && ps < (QT_R / 2) * (QT_C / 2 * CH / 4)) {
                const int yy = ps / (QT_C / 2 * CH / 4), w = ps - yy * (QT_C / 2 * CH / 4);
                const int oy = (R0 >> 1) + yy, drows = rows >> 1, dcols = cols >> 1;
                const int b0 = (C0 >> 1) * CH + 4 * w; // byte column in the destination row (a multiple of CH at w = 0)
                if (oy < drows && b0 < dcols * CH) {
                    uint32_t v = 0;
#pragma unroll
                    for (int m = 0; m < 4; ++m) v |= (s_src[(4 * w + m) % CH][0][0] & 0xffu) << (8 * m);
                    uint8_t* o = pyr_out + (size_t)oy * dcols * CH + b0;
                    if (b0 + 4 <= dcols * CH && (((uintptr_t)o) & 3) == 0) {
                        *(uint32_t*)o = v;
                    } else {
                        for (int m = 0; m < 4; ++m)
                            if (b0 + m < dcols * CH) o[m] = (uint8_t)(v >> (8 * m));
                    }
                }
            }
            return;
        }
    }
    QSTAMP(2)

    // The last 256 threads compute cv::pyrDown of the tile (8 x 32 outputs, REFLECT_101 at the image border:
    // the next pyramid level's source image) while the others run the two Gaussian passes: channel 0 beside
    // the horizontal pass, the remaining channels beside the vertical pass.
    const int nwork = pyr_out ? QN - 256 : QN; // threads of the Gaussian passes
    auto pyr_channels = [&](int k0, int k1) {
        const int pw = tid - (QN - 256);
        const int yy = pw >> 5, xx = pw & 31;
        const int oy = (R0 >> 1) + yy, ox = (C0 >> 1) + xx;
        const int drows = rows >> 1, dcols = cols >> 1;
        if (oy >= drows || ox >= dcols) return;
        if (rows >= 4 && cols >= 4) {
            // taps are 5 consecutive bytes of 5 consecutive tile rows: v_dot4 + 1.  REFLECT_101 touches only the
            // taps -2, -1 of the first output row / column (they fold onto +2, +1) and the tap +2 of the last one
            // when the extent is even (it folds onto the centre): per-lane weights, no divergent border path.
            const int r0t = 2 * yy + 3; // tile row of image row 2*oy - 2
            const int b0 = 2 * xx + 6;  // tile byte column of image col 2*ox - 2 (even: shift 0 or 2)
            const int w = b0 >> 2, sh = b0 & 3;
            uint32_t wlo = 0x04060401u, w4 = 1u; // weights of window bytes 0..3 and of byte 4
            if (ox == 0) wlo = 0x08060000u, w4 = 2u;
            else if (2 * ox + 2 >= cols) wlo = 0x04070401u, w4 = 0u;
            uint32_t wr[5] = {1u, 4u, 6u, 4u, 1u};
            if (oy == 0) wr[0] = 0u, wr[1] = 0u, wr[3] = 8u, wr[4] = 2u;
            else if (2 * oy + 2 >= rows) wr[2] = 7u, wr[4] = 0u;
#pragma unroll
            for (int k = 0; k < CH; ++k) {
                if (k < k0 || k >= k1) continue;
                uint32_t acc = 128u;
#pragma unroll
                for (int j = 0; j < 5; ++j) {
                    const uint32_t d0 = s_src[k][r0t + j][w], d1 = s_src[k][r0t + j][w + 1];
                    const uint32_t lo4 = __builtin_amdgcn_alignbyte(d1, d0, sh);
                    const uint32_t b4 = (d1 >> (8 * sh)) & 0xffu;
                    acc += wr[j] * __builtin_amdgcn_udot4(lo4, wlo, b4 * w4, false);
                }
                ((uint8_t*)&s_pyr[yy][0])[xx * CH + k] = (uint8_t)(acc >> 8);
            }
        } else { // degenerate extents: the literal reflect loop
            const int K5[5] = {1, 4, 6, 4, 1};
            int ri[5], ci[5];
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                ri[j] = reflect101(2 * oy + j - 2, rows) - (R0 - 5);
                ci[j] = reflect101(2 * ox + j - 2, cols) - (C0 - 8);
            }
            for (int k = k0; k < k1; ++k) {
                const uint8_t* sb = (const uint8_t*)&s_src[k][0][0];
                int acc = 0;
                for (int j = 0; j < 5; ++j) {
                    int h = 0;
                    for (int i = 0; i < 5; ++i) h += K5[i] * sb[ri[j] * QS_W + ci[i]];
                    acc += K5[j] * h;
                }
                ((uint8_t*)&s_pyr[yy][0])[xx * CH + k] = (uint8_t)((acc + 128) >> 8);
            }
        }
    };

    // ---- B: horizontal 7-tap: output x (image col C0-4+x) reads source bytes x+1 .. x+7 ----
    if (tid >= nwork) {
        pyr_channels(0, 1);
    } else {
        // an item = 8 outputs (two dwords of s_src columns): 4 source dwords, 16 v_dot4
        static_assert((QH_W / 4) % 2 == 0, "pairs of 4-pixel groups");
        for (int it = tid; it < CH * QS_R * (QH_W / 8); it += nwork) {
            const int k = it / (QS_R * (QH_W / 8)), rem = it - k * (QS_R * (QH_W / 8));
            const int r = rem / (QH_W / 8), g = 2 * (rem - r * (QH_W / 8));
            const uint32_t d0 = s_src[k][r][g], d1 = s_src[k][r][g + 1], d2 = s_src[k][r][g + 2], d3 = s_src[k][r][g + 3];
            const uint32_t KLO = 0x48381C08u; // 8, 28, 56, 72
            const uint32_t KHI = 0x00081C38u; // 56, 28, 8, 0
            const uint32_t a1 = __builtin_amdgcn_alignbyte(d1, d0, 1), a2 = __builtin_amdgcn_alignbyte(d1, d0, 2), a3 = __builtin_amdgcn_alignbyte(d1, d0, 3);
            const uint32_t b1 = __builtin_amdgcn_alignbyte(d2, d1, 1), b2 = __builtin_amdgcn_alignbyte(d2, d1, 2), b3 = __builtin_amdgcn_alignbyte(d2, d1, 3);
            const uint32_t c1 = __builtin_amdgcn_alignbyte(d3, d2, 1), c2 = __builtin_amdgcn_alignbyte(d3, d2, 2), c3 = __builtin_amdgcn_alignbyte(d3, d2, 3);
            const uint32_t h0 = __builtin_amdgcn_udot4(b1, KHI, __builtin_amdgcn_udot4(a1, KLO, 0u, false), false);
            const uint32_t h1 = __builtin_amdgcn_udot4(b2, KHI, __builtin_amdgcn_udot4(a2, KLO, 0u, false), false);
            const uint32_t h2 = __builtin_amdgcn_udot4(b3, KHI, __builtin_amdgcn_udot4(a3, KLO, 0u, false), false);
            const uint32_t h3 = __builtin_amdgcn_udot4(d2, KHI, __builtin_amdgcn_udot4(d1, KLO, 0u, false), false);
            const uint32_t h4 = __builtin_amdgcn_udot4(c1, KHI, __builtin_amdgcn_udot4(b1, KLO, 0u, false), false);
            const uint32_t h5 = __builtin_amdgcn_udot4(c2, KHI, __builtin_amdgcn_udot4(b2, KLO, 0u, false), false);
            const uint32_t h6 = __builtin_amdgcn_udot4(c3, KHI, __builtin_amdgcn_udot4(b3, KLO, 0u, false), false);
            const uint32_t h7 = __builtin_amdgcn_udot4(d3, KHI, __builtin_amdgcn_udot4(d2, KLO, 0u, false), false);
            *(uint4*)&s_h[k][r][2 * g] = make_uint4(h0 | (h1 << 16), h2 | (h3 << 16), h4 | (h5 << 16), h6 | (h7 << 16));
        }
    }
#ifdef SBM_QSTAMP_B
    QSTAMP_WAVES(24)
#endif
    lds_barrier();
    QSTAMP(3)

    // ---- C: vertical 7-tap over s_h rows jr .. jr+6; an item = one dword column (two pixels) x CG consecutive
    //         output rows, so the CG+6 source rows and the (row j, row j+1) pairings are shared ----
    constexpr int CG = 5;
    static_assert(QM_R % CG == 0, "row groups tile the smoothed rows");
    if (tid >= nwork) pyr_channels(1, CH);
    for (int it = tid; it < CH * (QM_R / CG) * (QH_W / 2) && tid < nwork; it += nwork) {
        const int kg = it / (QH_W / 2), d = it - kg * (QH_W / 2);
        const int k = kg / (QM_R / CG), jr0 = (kg - k * (QM_R / CG)) * CG;
        uint32_t a[CG + 6];
#pragma unroll
        for (int j = 0; j < CG + 6; ++j) a[j] = s_h[k][jr0 + j][d];
        uint32_t plo[CG + 5], phi[CG + 5]; // (row j, row j+1) of the left / right pixel
#pragma unroll
        for (int j = 0; j < CG + 5; ++j) {
            plo[j] = __builtin_amdgcn_perm(a[j + 1], a[j], 0x05040100u);
            phi[j] = __builtin_amdgcn_perm(a[j + 1], a[j], 0x07060302u);
        }
        const uint32_t K01 = 8u | (28u << 16), K23 = 56u | (72u << 16), K45 = 56u | (28u << 16), K6 = 8u;
#pragma unroll
        for (int o = 0; o < CG; ++o) {
            uint32_t lo = udot2(plo[o], K01, 32768u), hi = udot2(phi[o], K01, 32768u);
            lo = udot2(plo[o + 2], K23, lo);
            hi = udot2(phi[o + 2], K23, hi);
            lo = udot2(plo[o + 4], K45, lo);
            hi = udot2(phi[o + 4], K45, hi);
            lo = udot2(a[o + 6], K6, lo);            // low half = left pixel of row o+6
            hi = udot2(a[o + 6], K6 << 16, hi);      // high half = right pixel
            ((uint16_t*)&s_sm[k][jr0 + o][0])[d] = (uint16_t)((lo >> 16) | ((hi >> 16) << 8));
        }
    }
    QSTAMP(7)
#ifndef SBM_QSTAMP_B
    QSTAMP_WAVES(24)
#endif
#if defined(SBM_QSTAMP) && !defined(SBM_QSTAMP_LIGHT) && !defined(SBM_QSTAMP_B)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // split the barrier: LDS drain, then arrival
    QSTAMP_WAVES(40)
#endif
    lds_barrier();
    if (!interior) { // replicate the smoothed image outwards: Sobel runs with BORDER_REPLICATE
        // one dword (4 pixels) per item; sources are in-image positions, which no item writes
        for (int it = tid; it < CH * QM_R * (QH_W / 4); it += QN) {
            const int k = it / (QM_R * (QH_W / 4)), rem = it - k * (QM_R * (QH_W / 4));
            const int jr = rem / (QH_W / 4), g = rem - jr * (QH_W / 4);
            const int r = R0 - 2 + jr, c0 = C0 - 4 + 4 * g;
            const bool col_in = c0 >= 0 && c0 + 4 <= cols;
            if (r >= 0 && r < rows && col_in) continue;
            const int jrc = clampi(clampi(r, 0, rows - 1) - (R0 - 2), 0, QM_R - 1);
            uint32_t v;
            if (col_in) {
                v = s_sm[k][jrc][g];
            } else {
                const uint8_t* srow = (const uint8_t*)&s_sm[k][jrc][0];
                v = 0;
#pragma unroll
                for (int m = 0; m < 4; ++m) v |= (uint32_t)srow[clampi(clampi(c0 + m, 0, cols - 1) - (C0 - 4), 0, QH_W - 1)] << (8 * m);
            }
            s_sm[k][jr][g] = v;
        }
        lds_barrier();
    }

    QSTAMP(4)
    // the tile of the next pyramid level leaves as dwords (row segments of 32 * CH bytes), beside phase D
    if (pyr_out && tid >= QN - (QT_R / 2) * (QT_C / 2 * CH / 4)) {
        const int ps = tid - (QN - (QT_R / 2) * (QT_C / 2 * CH / 4));
        const int yy = ps / (QT_C / 2 * CH / 4), w = ps - yy * (QT_C / 2 * CH / 4);
        const int oy = (R0 >> 1) + yy, drows = rows >> 1, dcols = cols >> 1;
        const int b0 = (C0 >> 1) * CH + 4 * w; // byte column in the destination row
        if (oy < drows && b0 < dcols * CH) {
            uint8_t* o = pyr_out + (size_t)oy * dcols * CH + b0;
            const uint32_t v = s_pyr[yy][w];
            if (b0 + 4 <= dcols * CH && (((uintptr_t)o) & 3) == 0) {
                *(uint32_t*)o = v;
            } else {
                for (int m = 0; m < 4; ++m)
                    if (b0 + m < dcols * CH) o[m] = (uint8_t)(v >> (8 * m));
            }
        }
    }
    // ---- D: Sobel + magnitude + orientation; q row qr <-> image row R0-1+qr <-> s_sm row qr+1.
    //         An item = two adjacent pixels (x, x+1), held as the two 16-bit lanes of a dword: column sums /
    //         differences of the 3x3 window with packed adds, then (gx, gy) of one pixel in one dword so that
    //         gx^2 + gy^2 is a single v_dot2_i32_i16. ----
    // mag is an integer < 2^24, so `mag > thr` (float, line2Dup.cpp:241) is `mag > floor(thr)` in integers
    const int thr_i = (thr_sq < 2147483000.f) ? (int)floorf(thr_sq) : 0x7fffffff;
    for (int it = tid; it < QQ_R * (QH_W / 2); it += QN) {
        const int qr = it / (QH_W / 2), j = it - qr * (QH_W / 2);
        // the four columns x-1 .. x+2 sit in two consecutive dwords (lo, hi) of the smoothed row, at byte 3
        // (x = 4g, 4g+1) or at byte 1 (x = 4g+2, 4g+3) of lo
        const int g = j >> 1, odd = j & 1;
        const int glo = odd ? g : (g > 0 ? g - 1 : 0), ghi = odd ? (g < QH_W / 4 - 1 ? g + 1 : g) : g;
        const uint32_t selX = odd ? 0x0c020c01u : 0x0c040c03u; // (x-1, x)
        const uint32_t selY = odd ? 0x0c040c03u : 0x0c060c05u; // (x+1, x+2)
        uint32_t bv[2] = {0, 0}; // (gx, gy) of the strongest channel so far
        int bm[2] = {-1, -1};
#pragma unroll
        for (int k = 0; k < CH; ++k) {
            uint32_t X[3], Y[3];
#pragma unroll
            for (int rw = 0; rw < 3; ++rw) {
                const uint32_t lo = s_sm[k][qr + rw][glo], hi = s_sm[k][qr + rw][ghi];
                X[rw] = __builtin_amdgcn_perm(hi, lo, selX);
                Y[rw] = __builtin_amdgcn_perm(hi, lo, selY);
            }
            const uint32_t cwX = pk_add(pk_2a_plus_b(X[1], X[0]), X[2]), ddX = pk_sub(X[2], X[0]);
            const uint32_t cwY = pk_add(pk_2a_plus_b(Y[1], Y[0]), Y[2]), ddY = pk_sub(Y[2], Y[0]);
            const uint32_t gx = pk_sub(cwY, cwX);
            const uint32_t gy = pk_add(pk_2a_plus_b(__builtin_amdgcn_alignbit(ddY, ddX, 16), ddX), ddY);
            uint32_t v[2];
            v[0] = __builtin_amdgcn_perm(gy, gx, 0x05040100u);
            v[1] = __builtin_amdgcn_perm(gy, gx, 0x07060302u);
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const int mg = sdot2(v[m], v[m], 0);
                if (mg > bm[m]) { // maximum magnitude, ties to the lower channel (:370-387)
                    bm[m] = mg;
                    bv[m] = v[m];
                }
            }
        }
        uint32_t words[2], strong = 0;
        const int r = R0 - 1 + qr;
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const int c = C0 - 4 + 2 * j + m;
            uint32_t w;
            if (WITH_FLOAT) { // the unquantised angle is an output here: the float pipeline itself
                const int gx = (int)(short)(bv[m] & 0xffffu), gy = (int)bv[m] >> 16;
                const float ang = fast_atan2_deg((float)gy, (float)gx);
                int q16 = __float2int_rn(__fmul_rn(ang, (float)(16.0 / 360.0)));
                q16 = q16 < 0 ? 0 : (q16 > 255 ? 255 : q16);
                w = 1u << (4 * (q16 & 7));
                if (r >= 0 && r < rows && c >= 0 && c < cols && qr >= 1 && qr <= QT_R && c >= C0 && c < C0 + QT_C) {
                    if (mag_out) mag_out[(size_t)r * cols + c] = (float)bm[m];
                    if (ori_out) ori_out[(size_t)r * cols + c] = ang;
                }
            } else {
                w = vote_word(bv[m]);
            }
            uint32_t st = bm[m] > thr_i ? 0x80u : 0u;
            if (!interior) { // pixels outside the image and the 1-pixel ring (:232-238) vote for bin 0
                const bool inside = r >= 0 && r < rows && c >= 0 && c < cols;
                const bool ring = (r == 0) || (r == rows - 1) || (c == 0) || (c == cols - 1);
                if (!inside || ring) w = 1u;
                if (!inside) st = 0;
            }
            words[m] = w;
            strong |= st << (8 * m);
        }
        *(uint2*)&s_w[qr][2 * j] = make_uint2(words[0], words[1]);
        ((uint16_t*)&s_st[qr][0])[j] = (uint16_t)strong;
    }
    lds_barrier();
    QSTAMP(5)

    // ---- E: 3x3 majority vote (>= 5 of 9) around strong pixels: the nine vote words add up to eight 4-bit
    //         counters; an item = two output pixels, cols C0+2j, +1 <-> q columns 4+2j, 5+2j ----
    if (tid < QT_R * (QT_C / 2)) {
        const int er = tid >> 5, j = tid & 31;
        const int r = R0 + er;
        if (r < rows) {
            const uint32_t centre = ((const uint16_t*)&s_st[er + 1][0])[2 + j];
            uint32_t packed = 0;
            if (centre) {
                uint32_t acc0 = 0, acc1 = 0;
#pragma unroll
                for (int rw = 0; rw < 3; ++rw) {
                    const uint32_t w0 = s_w[er + rw][2 * j + 3];
                    const uint2 wc = *(const uint2*)&s_w[er + rw][2 * j + 4];
                    const uint32_t w3 = s_w[er + rw][2 * j + 6];
                    const uint32_t mid = wc.x + wc.y;
                    acc0 += w0 + mid;
                    acc1 += mid + w3;
                }
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    const int c = C0 + 2 * j + m;
                    const uint32_t hit = ((m ? acc1 : acc0) + 0x33333333u) & 0x88888888u; // counters >= 5 (at most one)
                    bool on = hit != 0 && ((centre >> (8 * m)) & 0xffu) != 0;
                    if (!interior) on = on && c < cols && r >= 1 && r < rows - 1 && c >= 1 && c < cols - 1;
                    uint32_t v = on ? 1u << (__builtin_ctz(hit) >> 2) : 0u;
                    if (mask && v && !mask[(size_t)r * cols + c]) v = 0;
                    packed |= v << (8 * m);
                }
            }
            uint8_t* o = out + (size_t)r * cols + C0 + 2 * j;
            if (C0 + 2 * j + 1 < cols && (((uintptr_t)o) & 1) == 0) {
                *(uint16_t*)o = (uint16_t)packed;
            } else {
                for (int m = 0; m < 2; ++m)
                    if (C0 + 2 * j + m < cols) o[m] = (uint8_t)(packed >> (8 * m));
            }
        }
    }
    QSTAMP(6)
}

// cv::pyrDown: [1 4 6 4 1]^2, (sum + 128) >> 8, REFLECT_101; dst = (rows/2, cols/2).
// One lane per output pixel (all channels).  Interior pixels of 1- and 3-channel
// images read each of the 5 source rows as unaligned dwords; the border ring
// takes the reflect path.
__global__ __launch_bounds__(256) void k_pyrdown(const uint8_t* __restrict__ src, int rows, int cols, int ch,
                                                 int stride, uint8_t* __restrict__ dst)
{
    const int dr = rows / 2, dc = cols / 2;
    const int n = dr * dc;
    const int K[5] = {1, 4, 6, 4, 1};
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < n; idx += gridDim.x * 256) {
        const int x = idx % dc, y = idx / dc;
        const bool interior = x >= 1 && y >= 1 && 2 * x + 2 < cols && 2 * y + 2 < rows &&
                              (2 * x - 2) * ch + 16 <= stride; // the 16-byte row read stays inside the row
        if (interior && ch == 3) {
            int acc[3] = {0, 0, 0};
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                const uint8_t* s = src + (size_t)(2 * y + j - 2) * stride + (2 * x - 2) * 3;
                uint32_t w[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) w[q] = ld_u32_any(s + 4 * q);
                int h[3] = {0, 0, 0};
#pragma unroll
                for (int i = 0; i < 5; ++i)
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        const int b = i * 3 + k;
                        h[k] += K[i] * (int)((w[b >> 2] >> (8 * (b & 3))) & 0xff);
                    }
#pragma unroll
                for (int k = 0; k < 3; ++k) acc[k] += K[j] * h[k];
            }
#pragma unroll
            for (int k = 0; k < 3; ++k) dst[(size_t)idx * 3 + k] = (uint8_t)((acc[k] + 128) >> 8);
        } else if (interior && ch == 1) {
            int acc = 0;
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                const uint8_t* s = src + (size_t)(2 * y + j - 2) * stride + (2 * x - 2);
                uint32_t w0 = ld_u32_any(s);
                int h = (int)(w0 & 0xff) + 4 * (int)((w0 >> 8) & 0xff) + 6 * (int)((w0 >> 16) & 0xff) + 4 * (int)(w0 >> 24) + (int)s[4];
                acc += K[j] * h;
            }
            dst[idx] = (uint8_t)((acc + 128) >> 8);
        } else {
            for (int k = 0; k < ch; ++k) {
                int acc = 0;
                for (int j = 0; j < 5; ++j) {
                    const uint8_t* s = src + (size_t)reflect101(2 * y + j - 2, rows) * stride;
                    int h = 0;
                    for (int i = 0; i < 5; ++i) h += K[i] * s[reflect101(2 * x + i - 2, cols) * ch + k];
                    acc += K[j] * h;
                }
                dst[(size_t)idx * ch + k] = (uint8_t)((acc + 128) >> 8);
            }
        }
    }
}

// cv::resize(INTER_LINEAR) of an 8-bit image with the host-computed coefficient tables (sbm_resize_table.h):
// one thread per destination sample; training-side helper (shapeInfo_producer::transform, line2Dup.h:379-405)
__global__ __launch_bounds__(256) void k_resize_linear_u8(const uint8_t* __restrict__ src, int rows, int cols, int ch, int stride,
                                                          const int32_t* __restrict__ xi, const int16_t* __restrict__ xa,
                                                          const int32_t* __restrict__ yi, const int16_t* __restrict__ ya,
                                                          uint8_t* __restrict__ dst, int drows, int dcols)
{
    const int64_t n = (int64_t)drows * dcols * ch;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int k = (int)(i % ch);
        const int64_t px = i / ch;
        const int x = (int)(px % dcols), y = (int)(px / dcols);
        const int x0 = xi[x], x1 = x0 + 1 < cols ? x0 + 1 : cols - 1;
        const int y0 = yi[y], y1 = y0 + 1 < rows ? y0 + 1 : rows - 1;
        const uint8_t* r0 = src + (size_t)y0 * stride;
        const uint8_t* r1 = src + (size_t)y1 * stride;
        dst[i] = resize_linear_sample(r0[x0 * ch + k], r0[x1 * ch + k], r1[x0 * ch + k], r1[x1 * ch + k], xa[2 * x], xa[2 * x + 1],
                                      ya[2 * y], ya[2 * y + 1]);
    }
}

__global__ __launch_bounds__(256) void k_resize_mask(const uint8_t* __restrict__ src, int rows, int cols,
                                                     uint8_t* __restrict__ dst, int drows, int dcols)
{
    top_wave_priority(); // a small kernel between two gradient launches of its batch
    const double fx = (double)cols / dcols, fy = (double)rows / drows;
    const int n = drows * dcols;
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < n; idx += gridDim.x * 256) {
        int x = idx % dcols, y = idx / dcols;
        int sx = (int)floor(x * fx), sy = (int)floor(y * fy);
        sx = sx > cols - 1 ? cols - 1 : sx;
        sy = sy > rows - 1 ? rows - 1 : sy;
        dst[idx] = src[(size_t)sy * cols + sx];
    }
}


} // namespace sbm
