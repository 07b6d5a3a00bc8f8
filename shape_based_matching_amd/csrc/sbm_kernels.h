// sbm_kernels.h — hand-written HIP kernels for gfx950 (MI355X, wave64) of the
// LINE-2D match() hot path.  Integer/byte work bound by HBM/L2 bandwidth: no
// MFMA anywhere.  Compiled with -ffp-contract=off: the gradient stage's float
// expressions must round after every operation (bit-exact with the oracle).
//
// Reference functions replaced (file:line in ddcr/shape_based_matching):
//   k_quantize            quantizedOrientations + hysteresisGradient   line2Dup.cpp:313-404, 218-311
//   k_pyrdown             cv::pyrDown in ColorGradientPyramid::pyrDown line2Dup.cpp:431-433
//   k_resize_mask         resize(mask, INTER_NEAREST)                  line2Dup.cpp:439
//   k_build_lm            spread + computeResponseMaps + linearize     line2Dup.cpp:616-630, 637-747, 749-777
//   k_spread/k_response/k_linearize   the same three, unfused (stage entry points)
//   k_prep_features       accessLinearMemory address arithmetic        line2Dup.cpp:782-805
//   k_similarity_coarse   similarity / similarity_64 + candidate scan  line2Dup.cpp:807-858, 924-984, 1199-1216
//   k_similarity_map      similarity / similarity_64 (score map out)   same
//   k_similarity_local    similarityLocal(_64) + best-of-16x16 + filter line2Dup.cpp:860-922, 986-1048, 1221-1293
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include "../../include/sbm_types.h"
#include "sbm_resize_table.h"

namespace sbm {

// per (template, level) record on the device
struct DevTL {
    int32_t width, height, nf, feat_off;
};
// coarse candidate / refinement state (Match under construction)
// Carries everything the NEXT refinement pass needs (the template record of that level, its keep
// threshold, the ids of the final Match), so that pass starts its feature loads after one
// dependent read instead of three.
struct Cand {
    int32_t t;   // template index (into the uploaded list)
    int32_t x;   // Match::x at the level just processed
    int32_t y;
    int32_t raw; // integer similarity sum; < 0 = dropped by the per-level filter
    int32_t next_width, next_height, next_nf, next_feat_off; // DevTL of the level refined next
    int32_t next_keep;                                       // raw_keep of that level
    int32_t class_idx, template_id;
    int32_t pad;
};
static_assert(sizeof(Cand) == 48, "Cand layout");

__device__ __forceinline__ void cand_fill_next(Cand& c, const DevTL* __restrict__ tls, const int32_t* __restrict__ raw_keep,
                                               const int32_t* __restrict__ class_idx, const int32_t* __restrict__ template_id,
                                               int L, int next_level)
{
    if (next_level >= 0) {
        const DevTL tn = tls[(size_t)c.t * L + next_level];
        c.next_width = tn.width;
        c.next_height = tn.height;
        c.next_nf = tn.nf;
        c.next_feat_off = tn.feat_off;
        c.next_keep = raw_keep[(size_t)c.t * L + next_level];
    }
    c.class_idx = class_idx[c.t];
    c.template_id = template_id[c.t];
    c.pad = 0;
}

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
__device__ __forceinline__ int reflect101(int p, int len)
{
    if (len == 1) return 0;
    while (p < 0 || p >= len) p = p < 0 ? -p : 2 * len - 2 - p;
    return p;
}

typedef uint32_t __attribute__((aligned(1))) u32_unaligned;
__device__ __forceinline__ uint32_t ld_u32_any(const uint8_t* p) { return *(const u32_unaligned*)p; }
struct __attribute__((aligned(4))) u128_a4 { uint32_t x, y, z, w; }; // 16-byte load at 4-byte alignment

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
                                                  int64_t pyr_fs)
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
    const int R0 = blockIdx.y * QT_R, C0 = blockIdx.x * QT_C;
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

// ---------------------------------------------------------------------------
// Linear memories: spread + response + linearize fused
// ---------------------------------------------------------------------------
constexpr int LM_GX = 64; // grid cells per block along x

// response of 4 packed spread bytes for orientation o: 4 if bit o set, else 3
// if a circular neighbour bit is set, else 0 (== SIMILARITY_LUT, line2Dup.cpp:635)
__device__ __forceinline__ uint32_t response4(uint32_t v, int o)
{
    uint32_t self = (v >> o) & 0x01010101u;
    uint32_t nb = ((v >> ((o + 1) & 7)) | (v >> ((o + 7) & 7))) & 0x01010101u & ~self;
    return (self << 2) | (nb * 3u);
}

// One block = one row of grid cells (T pixel rows) x 64 cells.  The one-hot
// tile (+T-1 halo, zero outside the image = the clipped window of spread())
// is OR-reduced separably in LDS, stored there already in linear-memory order
// [ty][tx][gx], then each lane turns 4 consecutive cells into the 8
// orientation responses and writes 8 dwords, coalesced along gx.
__global__ __launch_bounds__(256) void k_build_lm(const uint8_t* __restrict__ q, int rows, int cols, int T,
                                                  int W, int H, uint8_t* __restrict__ lm, int64_t lm_stride)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int tid = threadIdx.x;
    const int gx0 = blockIdx.x * LM_GX, gy = blockIdx.y;
    const int tw = LM_GX * T;      // tile width in pixels
    const int lw = tw + T - 1;     // + halo
    const int lwp = (lw + 3) & ~3; // padded row pitch
    const int lh = 2 * T - 1;
    uint8_t* s_q = smem;                  // [lh][lwp]
    uint8_t* s_h = s_q + lh * lwp;        // [lh][tw]   horizontal OR
    uint8_t* s_sp = s_h + lh * tw;        // [T*T][LM_GX] spread, linearized order
    const int pr0 = gy * T, pc0 = gx0 * T;
    for (int idx = tid; idx < lh * lwp; idx += 256) {
        int r = idx / lwp, c = idx - r * lwp;
        int gr = pr0 + r, gc = pc0 + c;
        s_q[idx] = (c < lw && gr < rows && gc < cols) ? q[(size_t)gr * cols + gc] : (uint8_t)0;
    }
    __syncthreads();
    for (int idx = tid; idx < lh * tw; idx += 256) {
        int r = idx / tw, c = idx - r * tw;
        uint8_t v = 0;
        for (int d = 0; d < T; ++d) v |= s_q[r * lwp + c + d];
        s_h[idx] = v;
    }
    __syncthreads();
    for (int idx = tid; idx < T * tw; idx += 256) {
        int ty = idx / tw, c = idx - ty * tw;
        uint8_t v = 0;
        for (int d = 0; d < T; ++d) v |= s_h[(ty + d) * tw + c];
        int g = c / T, tx = c - g * T;
        s_sp[(ty * T + tx) * LM_GX + g] = v;
    }
    __syncthreads();
    const int64_t WH = (int64_t)W * H;
    const int items = T * T * (LM_GX / 4);
    const bool fast = (W & 3) == 0;
    for (int it = tid; it < items; it += 256) {
        int sub = it / (LM_GX / 4), g4 = it - sub * (LM_GX / 4);
        int gx = gx0 + g4 * 4;
        if (gx >= W) continue;
        uint32_t v = ((const uint32_t*)s_sp)[sub * (LM_GX / 4) + g4];
        int64_t dst = (int64_t)sub * WH + (int64_t)gy * W + gx;
        if (fast) {
#pragma unroll
            for (int o = 0; o < 8; ++o) *(uint32_t*)(lm + o * lm_stride + dst) = response4(v, o);
        } else {
            for (int o = 0; o < 8; ++o) {
                uint32_t rsp = response4(v, o);
                for (int k = 0; k < 4 && gx + k < W; ++k) lm[o * lm_stride + dst + k] = (uint8_t)(rsp >> (8 * k));
            }
        }
    }
}

// Strip-interleaved compact plane of a refinement-only level (needs W % 16 == 0): inside sub-plane
// sub = (y%T)*T + x%T the W x H grid is cut into W/16 column strips of 16 cells, and a strip is stored row after
// row, 16 bytes per row:  offset = sub*W*H + ((gx / 16) * H + gy) * 16 + gx % 16.
// similarityLocal (line2Dup.cpp:860-922) reads 16 x 16 cells per feature; row-major that is 16 pieces of 16 bytes in
// 16 different 128-byte lines, here it is two runs of 256 contiguous bytes at most (one when gx % 16 == 0).
__host__ __device__ __forceinline__ int64_t lm_strip_offset(int sub, int gy, int gx, int W, int H)
{
    return (int64_t)sub * W * H + ((int64_t)(gx >> 4) * H + gy) * 16 + (gx & 15);
}

// ---- register-only variant for T = 4 and T = 8 (the reference's strides) ----
// One lane owns 4 consecutive grid cells (4*T pixels) of one (ty, gy) pixel row
// r0 = gy*T + ty: it ORs the T source rows r0..r0+T-1 (16-byte loads), ORs T
// pixels forward with funnel shifts, transposes the 4 cells x T sub-columns in
// registers (v_perm), and for every tx and orientation stores one dword =
// responses of cells gx..gx+3.  A wave's store is one contiguous run of
// LM[o][ty*T+tx][gy*W + ...]: no LDS, no barrier, fully coalesced.
// Requires W % 4 == 0 and cols % 16 == 0 (host checks; else k_build_lm).
__device__ __forceinline__ uint32_t perm_b32(uint32_t hi, uint32_t lo, uint32_t sel)
{
    return __builtin_amdgcn_perm(hi, lo, sel);
}

constexpr int LM_FULL_SPLIT = 4; // work items per (pixel row, 4 cells) of a level stored as 8 response planes

template <int T>
__device__ __forceinline__ void build_lm_rows_item(const uint8_t* __restrict__ q, int rows, int cols, int W, int H,
                                                   uint8_t* __restrict__ lm, int64_t lm_stride, int64_t item,
                                                   int compact, int split, uint32_t* s_tile)
{
    constexpr int NQ = T / 4 * 4; // dwords of own pixels per lane (4 cells * T px / 4)
    int r0, k, part = 0;
    bool active = true; // strip mode: every thread of the workgroup reaches the barrier below
    int tile_cb = 0, tile_gyb = 0, tile_ty = 0;
    if (compact == 2) {
        // strip-interleaved plane: a WORKGROUP = one ty x 16 grid rows x 64 cells (4 strips).  Thread (row = t / 16,
        // kk = t % 16) reads 4 cells of its row -- a wave's loads are 4 runs of 256 bytes -- and the spread dwords go
        // through LDS so that wave w then stores strip w of the tile: 16 strip rows of 16 bytes = one run of 256 bytes
        // per store.  (Round 2's first form let a wave own one strip end to end: contiguous stores, but each load
        // gathered 64-byte pieces of 16 different rows; 14.9 us of the 16-frame launch were level 0.)
        const int n_cb = (W + 63) >> 6, gyb_n = (H + 15) >> 4;
        const int t = (int)(item & 255);
        const int64_t blk = item >> 8;
        tile_cb = (int)(blk % n_cb);
        const int64_t rest = blk / n_cb;
        tile_gyb = (int)(rest % gyb_n);
        tile_ty = (int)(rest / gyb_n);
        const int gy_ = tile_gyb * 16 + (t >> 4);
        k = tile_cb * 16 + (t & 15);
        active = tile_ty < T && gy_ < H && k * 4 < W;
        r0 = active ? gy_ * T + tile_ty : 0;
        if (!active) k = 0;
    } else {
        const int lanes_per_row = W >> 2;
        int64_t it = item;
        if (!compact && split > 1) { // LM_FULL_SPLIT items per (pixel row, 4 cells): part = (tx half, orientation half)
            const int64_t base_items = (int64_t)rows * lanes_per_row;
            part = (int)(item / base_items);
            if (part >= LM_FULL_SPLIT) return;
            it = item - part * base_items;
        }
        const int64_t row_id = it / lanes_per_row; // = gy * T + ty  (a pixel row index)
        k = (int)(it - row_id * lanes_per_row);
        if (row_id >= rows) return;
        r0 = (int)row_id;
    }
    const int gy = r0 / T, ty = r0 - gy * T;
    const int c0 = k * 4 * T; // first pixel column of this lane
    // vertical OR of rows r0 .. r0+T-1 (clipped at the bottom, :626-627), own pixels + T px of right halo
    uint32_t v[NQ + T / 4];
#pragma unroll
    for (int i = 0; i < NQ + T / 4; ++i) v[i] = 0;
#pragma unroll
    for (int d = 0; d < T; ++d) {
        const int r = r0 + d;
        if (r < rows && active) {
            const uint8_t* src = q + (size_t)r * cols + c0;
#pragma unroll
            for (int i = 0; i < NQ; i += 4) {
                const uint4 w = *(const uint4*)(src + 4 * i);
                v[i] |= w.x;
                v[i + 1] |= w.y;
                v[i + 2] |= w.z;
                v[i + 3] |= w.w;
            }
            if (c0 + 4 * NQ < cols) { // right halo (zero past the last column)
#pragma unroll
                for (int i = 0; i < T / 4; ++i) v[NQ + i] |= *(const uint32_t*)(src + 4 * (NQ + i));
            }
        }
    }
    // horizontal OR over T pixels forward: byte c |= bytes c+1 .. c+T-1
    uint32_t s[NQ];
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
        uint32_t a = v[i], b = v[i + 1];
        uint32_t o = a | __builtin_amdgcn_alignbyte(b, a, 1) | __builtin_amdgcn_alignbyte(b, a, 2) |
                     __builtin_amdgcn_alignbyte(b, a, 3);
        if (T == 8) {
            uint32_t c2 = v[i + 2];
            o |= b | __builtin_amdgcn_alignbyte(c2, b, 1) | __builtin_amdgcn_alignbyte(c2, b, 2) |
                 __builtin_amdgcn_alignbyte(c2, b, 3);
        }
        s[i] = o;
    }
    // s[] holds 4 cells x T sub-columns; gather, per tx, the 4 cells' bytes into one dword
    const int64_t WH = (int64_t)W * H;
    const int64_t cell = (int64_t)gy * W + k * 4;
    auto spread_dword = [&](int tx) {
        // cell j lives in dwords s[j*T/4 .. ], sub-column tx is byte (tx & 3) of dword j*(T/4) + (tx >> 2)
        const int dsel = tx >> 2, b = tx & 3;
        const uint32_t d0 = s[0 * (T / 4) + dsel], d1 = s[1 * (T / 4) + dsel], d2 = s[2 * (T / 4) + dsel],
                       d3 = s[3 * (T / 4) + dsel];
        // v_perm selector bytes: 0-3 pick from 'lo', 4-7 from 'hi'
        const uint32_t p01 = perm_b32(d1, d0, 0x0c0c0000u | ((4 + b) << 8) | b);        // {d0.b, d1.b, 0, 0}
        const uint32_t p23 = perm_b32(d3, d2, 0x00000c0cu | ((4 + b) << 24) | (b << 16)); // {0, 0, d2.b, d3.b}
        return p01 | p23;
    };
    if (compact == 2) { // strip-interleaved spread plane (lm_strip_offset): a 16 x 16 patch is 2 - 4 cache lines
        const int t = (int)(item & 255);
        const int kk = t & 15, row = t >> 4;
#pragma unroll
        for (int tx = 0; tx < T; ++tx) s_tile[tx * 256 + (kk >> 2) * 64 + row * 4 + (kk & 3)] = spread_dword(tx);
        __syncthreads();
        const int S = tile_cb * 4 + (t >> 6), gy2 = tile_gyb * 16 + ((t & 63) >> 2);
        if (tile_ty < T && S < (W >> 4) && gy2 < H) {
#pragma unroll
            for (int tx = 0; tx < T; ++tx)
                *(uint32_t*)(lm + lm_strip_offset(tile_ty * T + tx, gy2, S * 16 + (t & 3) * 4, W, H)) = s_tile[tx * 256 + t];
        }
    } else if (compact) { // one plane of spread bytes: the reader applies the response LUT for its own orientation
#pragma unroll
        for (int tx = 0; tx < T; ++tx) *(uint32_t*)(lm + (int64_t)(ty * T + tx) * WH + cell) = spread_dword(tx);
    } else {
        // 8 response planes: T * 8 dword stores per (row, 4 cells) -- 64 at T = 8, on a level with few rows.  For a
        // single frame (split > 1) the item is cut into LM_FULL_SPLIT parts (tx half = part >> 1, orientation half =
        // part & 1; the loads and ORs are repeated) so that the level is four times as many, four times shorter waves:
        // 7.7 -> 6.9 us.  A batch of frames has enough waves and only pays the repeats (23 -> 25 us): split = 1.
        if (split > 1) {
            const int txh = part >> 1, o0 = (part & 1) * 4;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                if (h != txh) continue;
#pragma unroll
                for (int t = 0; t < T / 2; ++t) {
                    const int tx = h * (T / 2) + t;
                    const uint32_t sp = spread_dword(tx);
                    const int64_t dst = (int64_t)(ty * T + tx) * WH + cell;
#pragma unroll
                    for (int o = 0; o < 4; ++o) *(uint32_t*)(lm + (o0 + o) * lm_stride + dst) = response4(sp, o0 + o);
                }
            }
        } else {
#pragma unroll
            for (int tx = 0; tx < T; ++tx) {
                const uint32_t sp = spread_dword(tx);
                const int64_t dst = (int64_t)(ty * T + tx) * WH + cell;
#pragma unroll
                for (int o = 0; o < 8; ++o) *(uint32_t*)(lm + o * lm_stride + dst) = response4(sp, o);
            }
        }
    }
}

// All pyramid levels in one launch: block ranges per level (levels whose T is 4 or 8), plus the
// reset of the per-frame counters, so the whole linear-memory stage costs one kernel boundary.
struct LmLevelArgs {
    const uint8_t* q;
    uint8_t* lm;
    int64_t lm_stride;
    int32_t rows, cols, W, H, T;
    int32_t block_begin; // first block of this level
    int64_t q_fs, lm_fs; // bytes from one frame of a batch to the next
    int32_t compact;     // 1: lm is ONE plane [T*T][W*H] of spread bytes (a level that only the refinement pass
                         // reads): 1/8 of the stores and of the HBM write-back; 2: the same plane strip-interleaved
    int32_t split;       // 8-plane levels: LM_FULL_SPLIT work items per (pixel row, 4 cells) instead of 1
};
struct LmArgs {
    LmLevelArgs lv[SBM_MAX_LEVELS];
    int32_t n_levels;
    int32_t* counters;  // may be null
    int32_t* out_count; // may be null
};

__global__ __launch_bounds__(256) void k_build_lm_rows(const LmArgs a)
{
    __shared__ uint32_t s_tile[8 * 256]; // strip levels: the workgroup's spread dwords, [tx][strip][row][4 cells]
    const size_t frame = blockIdx.y; // batch of frames: one grid row each
    if (blockIdx.x == 0 && a.counters) {
        if (threadIdx.x < 40) a.counters[frame * 40 + threadIdx.x] = 0;
        if (threadIdx.x < 2 && a.out_count) a.out_count[frame * 2 + threadIdx.x] = 0;
    }
    // the level whose block range holds this block (the host orders the ranges heaviest blocks first)
    int l = 0, lb = -1;
#pragma unroll
    for (int i = 0; i < SBM_MAX_LEVELS; ++i)
        if (i < a.n_levels && (int)blockIdx.x >= a.lv[i].block_begin && a.lv[i].block_begin > lb) l = i, lb = a.lv[i].block_begin;
    const LmLevelArgs& p = a.lv[l];
    const int64_t item = (int64_t)((int)blockIdx.x - p.block_begin) * 256 + threadIdx.x;
    const uint8_t* q = p.q + frame * p.q_fs;
    uint8_t* lm = p.lm + frame * p.lm_fs;
    if (p.T == 4) build_lm_rows_item<4>(q, p.rows, p.cols, p.W, p.H, lm, p.lm_stride, item, p.compact, p.split, s_tile);
    else build_lm_rows_item<8>(q, p.rows, p.cols, p.W, p.H, lm, p.lm_stride, item, p.compact, p.split, s_tile);
}

// compact plane (spread bytes) -> the 8 response planes, for the stage entry points that hand out or read a
// full linear memory of a refinement-only level
__global__ __launch_bounds__(256) void k_expand_lm(const uint8_t* __restrict__ lmc, int64_t n_bytes,
                                                   uint8_t* __restrict__ lm, int64_t lm_stride, int strip, int W, int H)
{
    const int64_t n4 = n_bytes >> 2; // T*T*W*H is a multiple of 16
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        int64_t src = 4 * i;
        if (strip) { // i enumerates the row-major output: [sub][gy][gx .. gx+3]
            const int64_t wh = (int64_t)W * H, sub = (4 * i) / wh, rem = 4 * i - sub * wh;
            src = lm_strip_offset((int)sub, (int)(rem / W), (int)(rem % W), W, H);
        }
        const uint32_t v = *(const uint32_t*)(lmc + src);
#pragma unroll
        for (int o = 0; o < 8; ++o) *(uint32_t*)(lm + o * lm_stride + 4 * i) = response4(v, o);
    }
}

// unfused single-function kernels (stage entry points / parity tests)
__global__ __launch_bounds__(256) void k_spread(const uint8_t* __restrict__ src, int rows, int cols, int T,
                                                uint8_t* __restrict__ dst)
{
    const int64_t n = (int64_t)rows * cols;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < n; idx += (int64_t)gridDim.x * 256) {
        int r = (int)(idx / cols), c = (int)(idx - (int64_t)r * cols);
        uint8_t v = 0;
        for (int dr = 0; dr < T && r + dr < rows; ++dr)
            for (int dc = 0; dc < T && c + dc < cols; ++dc) v |= src[(size_t)(r + dr) * cols + c + dc];
        dst[idx] = v;
    }
}

__global__ __launch_bounds__(256) void k_response(const uint8_t* __restrict__ spread, int64_t n,
                                                  uint8_t* __restrict__ maps)
{
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < n; idx += (int64_t)gridDim.x * 256) {
        uint32_t v = spread[idx];
#pragma unroll
        for (int o = 0; o < 8; ++o) maps[(int64_t)o * n + idx] = (uint8_t)response4(v, o);
    }
}

__global__ __launch_bounds__(256) void k_linearize(const uint8_t* __restrict__ map, int rows, int cols, int T,
                                                   uint8_t* __restrict__ lm)
{
    const int W = cols / T, H = rows / T;
    const int64_t n = (int64_t)rows * cols;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < n; idx += (int64_t)gridDim.x * 256) {
        // idx enumerates the OUTPUT: [ty][tx][gy][gx]
        int gx = (int)(idx % W);
        int64_t t1 = idx / W;
        int gyy = (int)(t1 % H);
        int sub = (int)(t1 / H);
        int ty = sub / T, tx = sub - ty * T;
        lm[idx] = map[(size_t)(gyy * T + ty) * cols + gx * T + tx];
    }
}

// ---------------------------------------------------------------------------
// Similarity
// ---------------------------------------------------------------------------

// per-feature linear-memory offset for the current level geometry
__global__ __launch_bounds__(256) void k_prep_features(const uint32_t* __restrict__ fxy,
                                                       const uint8_t* __restrict__ flabel,
                                                       const uint8_t* __restrict__ flevel, int64_t n,
                                                       const int* __restrict__ Tl, const int* __restrict__ Wl,
                                                       const int* __restrict__ Hl,
                                                       const int64_t* __restrict__ stride_l,
                                                       int32_t* __restrict__ foff, int lc)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int l = flevel[i];
        const int T = Tl[l], W = Wl[l], H = Hl[l];
        const int x = fxy[i] & 0xffff, y = fxy[i] >> 16;
        int64_t off = (int64_t)flabel[i] * stride_l[l] + (int64_t)((y % T) * T + (x % T)) * W * H + (int64_t)(y / T) * W + x / T;
        // Coarsest level: a feature outside the image is skipped by the reference (:836-837) -- it adds nothing.  Point it
        // at the zero tail of the last orientation's plane (every position of the span reads 0 there), so that the coarse
        // pass needs neither the feature's coordinates nor a bounds test.  The refinement levels keep the plain offset:
        // there the test is on the feature moved by the candidate's patch origin.
        if (l == lc && (x >= W * T || y >= H * T)) off = 7 * stride_l[l] + (int64_t)T * T * W * H;
        foff[i] = (int32_t)off;
    }
}

// What one work item of the coarse pass needs about its template, in one 32-byte record (one scalar load instead of
// the chain active[] -> tls[] -> raw_min[]), and the template's coarsest-level feature offsets SORTED BY THEIR BYTE
// MISALIGNMENT (offset & 3): features of one class are summed as raw dwords and re-aligned once per batch instead of
// once per feature (a sum does not care about the order of its terms).  e01 / e2: where classes 0, 1, 2 end in the
// sorted list (class 3 ends at nf).  cfoff: the first 64 sorted offsets, indexed by the item's slot (a vector load that
// does not wait for the record).
struct CoarseItem {
    int32_t t, soff_base, nf, npos, rmin, e01, e2, pad;
};
__global__ __launch_bounds__(64) void k_prep_coarse_items(const int32_t* __restrict__ active, int n_active, const DevTL* __restrict__ tls,
                                                          int L, int lc, const int32_t* __restrict__ raw_min,
                                                          const int32_t* __restrict__ foff, const int32_t* __restrict__ soff_base,
                                                          int T, int W, int H, int zero_off, CoarseItem* __restrict__ items,
                                                          int32_t* __restrict__ soff, int32_t* __restrict__ cfoff)
{
    const int slot = blockIdx.x * 64 + threadIdx.x;
    if (slot >= n_active) return;
    const int t = active[slot];
    const DevTL tl = tls[(size_t)t * L + lc];
    const int32_t* f = foff + tl.feat_off;
    int32_t* out = soff + soff_base[slot];
    int cnt[4] = {0, 0, 0, 0};
    for (int i = 0; i < tl.nf; ++i) ++cnt[f[i] & 3];
    int pos[4] = {0, cnt[0], cnt[0] + cnt[1], cnt[0] + cnt[1] + cnt[2]};
    const int e0 = pos[1], e1 = pos[2], e2 = pos[3];
    for (int i = 0; i < tl.nf; ++i) { // stable counting sort over the four classes
        const int o = f[i];
        out[pos[o & 3]++] = o;
    }
    for (int i = 0; i < 64; ++i) cfoff[(size_t)slot * 64 + i] = i < tl.nf ? out[i] : zero_off;
    const int wf = (tl.width - 1) / T + 1, hf = (tl.height - 1) / T + 1;
    CoarseItem it;
    it.t = t;
    it.soff_base = soff_base[slot];
    it.nf = tl.nf;
    it.npos = (H - hf) * W + (W - wf) + 1;
    it.rmin = raw_min[(size_t)t * L + lc];
    it.e01 = e0 | (e1 << 16);
    it.e2 = e2;
    it.pad = 0;
    items[slot] = it;
}

__device__ __forceinline__ int template_positions(const DevTL& tl, int W, int H, int T)
{
    int wf = (tl.width - 1) / T + 1, hf = (tl.height - 1) / T + 1;
    return (H - hf) * W + (W - wf) + 1;
}

constexpr int SIM_POS_PER_BLOCK = 1024; // 256 lanes x 4 positions
constexpr int FB = 8;                   // features per batch: FB independent loads in flight per lane
constexpr int FBL = 16;                 // refinement pass: few features per wave, all in flight at once
constexpr int FB16 = 8;                 // same for the 16-byte variant (coarse pass)

// Sum `count` features starting at `f0` into packed 16-bit lanes.  Each lane
// owns 4 consecutive bytes at lm + foff[f] + lane_off.  The wave first loads 64
// features' (position, offset) records with one vector load per array, then
// broadcasts them one at a time with v_readlane: the inner loop is branch-free
// and keeps FB independent 4-byte loads in flight per lane.  MUST be called with
// all 64 lanes of the wave active (v_readlane reads lanes regardless of EXEC, and
// an inactive lane's record would be stale).  A feature that is
// out of the image (skipped by the reference, :836-837 / :878-879) or past the
// end of the range reads zero bytes from the linear memories' zero tail.
// Packed-u8 partial sums are carry-free for <= 63 addends (63 * 4 = 252) and
// are flushed into two packed-u16 words before that.
// COMPACT: lm is the single spread plane of a refinement-only level; the feature's orientation plane offset
// (label * lm_stride, part of foff) is taken out again and the response LUT is applied to the loaded bytes.
template <bool COMPACT = false>
__device__ __forceinline__ void accumulate_features(const uint8_t* __restrict__ lm,
                                                    const uint32_t* __restrict__ fxy,
                                                    const int32_t* __restrict__ foff, int f0, int count,
                                                    int rows, int cols, int ox, int oy, int lane_off,
                                                    const uint8_t* __restrict__ zero_addr, int zero_lane_off,
                                                    uint32_t& lo, uint32_t& hi,
                                                    const uint8_t* __restrict__ flabel = nullptr, int lm_stride = 0)
{
    const int lane = threadIdx.x & 63;
    const uint8_t* p_ok = lm + lane_off;
    const uint8_t* p_zero = zero_addr + zero_lane_off;
    uint32_t acc = 0;
    lo = hi = 0;
    int pending = 0;
    count = __builtin_amdgcn_readfirstlane(count); // wave-uniform by contract: keep the loop control scalar
    f0 = __builtin_amdgcn_readfirstlane(f0);
    for (int b = 0; b < count; b += 64) {
        int sel = -1, lab = 0;
        if (b + lane < count) {
            const uint32_t xy = fxy[f0 + b + lane];
            const int x = (int)(xy & 0xffff) + ox, y = (int)(xy >> 16) + oy;
            if (x >= 0 && y >= 0 && x < cols && y < rows) sel = foff[f0 + b + lane];
            if (COMPACT) {
                lab = flabel[f0 + b + lane];
                if (sel >= 0) sel -= lab * lm_stride;
            }
        }
        const int nb = count - b < 64 ? count - b : 64;
        // straight-line batches (no branches between the loads of a batch, so all of a batch's
        // loads are in flight together); the tail runs as batches of 4, 2, 1: no padding loads
        auto batch = [&](auto N, int u) {
            constexpr int n = decltype(N)::value;
            uint32_t v[n];
#pragma unroll
            for (int k = 0; k < n; ++k) {
                const int o = __builtin_amdgcn_readlane(sel, (u + k) & 63);
                v[k] = ld_u32_any(o >= 0 ? p_ok + o : p_zero);
            }
            if (COMPACT) {
#pragma unroll
                for (int k = 0; k < n; ++k) v[k] = response4(v[k], __builtin_amdgcn_readlane(lab, (u + k) & 63));
            }
#pragma unroll
            for (int k = 0; k < n; ++k) acc += v[k];
            pending += n;
            if (pending + FBL > 63) {
                lo += acc & 0x00ff00ffu;
                hi += (acc >> 8) & 0x00ff00ffu;
                acc = 0;
                pending = 0;
            }
        };
        int u = 0;
        for (; u + FBL <= nb; u += FBL) batch(std::integral_constant<int, FBL>{}, u);
        if (FBL > 8 && nb - u >= 8) { batch(std::integral_constant<int, 8>{}, u); u += 8; }
        if (nb - u >= 4) { batch(std::integral_constant<int, 4>{}, u); u += 4; }
        if (nb - u >= 2) { batch(std::integral_constant<int, 2>{}, u); u += 2; }
        if (nb - u >= 1) batch(std::integral_constant<int, 1>{}, u);
    }
    lo += acc & 0x00ff00ffu;
    hi += (acc >> 8) & 0x00ff00ffu;
}


__device__ __forceinline__ int unpack4(uint32_t lo, uint32_t hi, int k)
{
    return (k & 1) ? (int)((hi >> (8 * (k - 1))) & 0xffff) : (int)((lo >> (8 * k)) & 0xffff);
}

// 16 consecutive positions per lane: one 4-byte-aligned 16-byte load plus one
// 4-byte load per feature, re-aligned with v_alignbyte by the (wave-uniform)
// byte misalignment of the feature's linear-memory offset.  Measured on
// MI355X (tools/ld_probe.hip): 4-byte-aligned dwordx4 streams from L2 at
// ~12 TB/s, twice the rate of dword loads and of byte-misaligned dwordx4.
// j0 must be a multiple of 4.  Same calling convention as accumulate_features
// (all 64 lanes active).  lo[i]/hi[i]: packed u16 sums of bytes 4i..4i+3.
template <bool RESET = true>
__device__ __forceinline__ void accumulate_features16(const uint8_t* __restrict__ lm,
                                                      const uint32_t* __restrict__ fxy,
                                                      const int32_t* __restrict__ foff, int count,
                                                      int rows, int cols, int j0, bool lane_on, int zero_off,
                                                      uint32_t (&lo)[4], uint32_t (&hi)[4])
{
    const int lane = threadIdx.x & 63;
    const uint8_t* p = lm + j0;
    uint32_t acc[4] = {0, 0, 0, 0};
    if (RESET) { // RESET == false: add to the sums the caller already holds
#pragma unroll
        for (int i = 0; i < 4; ++i) lo[i] = hi[i] = 0;
    }
    int pending = 0;
    count = __builtin_amdgcn_readfirstlane(count); // wave-uniform by contract: keep the loop control scalar
    for (int b = 0; b < count; b += 64) {
        int sel = zero_off; // computed with every lane active: v_readlane below reads any lane's copy
        if (b + lane < count) {
            // both loads issued together (a load of foff under the bounds test would wait for fxy first)
            const uint32_t xy = fxy[b + lane];
            const int off = foff[b + lane];
            const int x = (int)(xy & 0xffff), y = (int)(xy >> 16);
            if (x < cols && y < rows) sel = off;
        }
        const int nb = count - b < 64 ? count - b : 64;
        // lanes past the template's span (lane_on == false) issue no loads at all; lane 63 fetches the
        // 4 bytes that follow its 16, every other lane takes them from its right neighbour's load (DPP)
        auto batch = [&](auto N, int u) {
            constexpr int n = decltype(N)::value;
            u128_a4 q[n];
            uint32_t e[n];
            int sh[n];
            if (lane_on) {
#pragma unroll
                for (int k = 0; k < n; ++k) {
                    const int o = __builtin_amdgcn_readlane(sel, (u + k) & 63);
                    sh[k] = o & 3;
                    const uint8_t* a = p + (o & ~3);
                    q[k] = *(const u128_a4*)a;
                    e[k] = 0;
                    if (lane == 63) e[k] = *(const uint32_t*)(a + 16);
                }
            } else {
#pragma unroll
                for (int k = 0; k < n; ++k) {
                    q[k] = u128_a4{0, 0, 0, 0};
                    e[k] = 0;
                    sh[k] = 0;
                }
            }
#pragma unroll
            for (int k = 0; k < n; ++k) {
                // wave_shl:1 -> lane i receives lane i+1's first dword; lane 63 keeps its own loaded value
                const uint32_t nx = (uint32_t)__builtin_amdgcn_update_dpp((int)e[k], (int)q[k].x, 0x130, 0xf, 0xf, false);
                acc[0] += __builtin_amdgcn_alignbyte(q[k].y, q[k].x, sh[k]);
                acc[1] += __builtin_amdgcn_alignbyte(q[k].z, q[k].y, sh[k]);
                acc[2] += __builtin_amdgcn_alignbyte(q[k].w, q[k].z, sh[k]);
                acc[3] += __builtin_amdgcn_alignbyte(nx, q[k].w, sh[k]);
            }
            pending += n;
            if (pending + FB16 > 63) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    lo[i] += acc[i] & 0x00ff00ffu;
                    hi[i] += (acc[i] >> 8) & 0x00ff00ffu;
                    acc[i] = 0;
                }
                pending = 0;
            }
        };
        int u = 0;
        for (; u + FB16 <= nb; u += FB16) batch(std::integral_constant<int, FB16>{}, u);
        if (nb - u >= 4) { batch(std::integral_constant<int, 4>{}, u); u += 4; }
        if (nb - u >= 2) { batch(std::integral_constant<int, 2>{}, u); u += 2; }
        if (nb - u >= 1) batch(std::integral_constant<int, 1>{}, u);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        lo[i] += acc[i] & 0x00ff00ffu;
        hi[i] += (acc[i] >> 8) & 0x00ff00ffu;
    }
}

typedef unsigned short coarse_us2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t coarse_pk_max_u16(uint32_t a, uint32_t b) // v_pk_max_u16
{
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(coarse_us2, a), __builtin_bit_cast(coarse_us2, b)));
}

// The coarse pass of a wave that owns ALL nf features of its item, with exact pruning.
//
// Pruning: a feature adds at most 4, so once `done` features are in, a position whose partial sum is below
// rmin - 4*(nf - done) cannot reach rmin any more; after every batch the wave looks whether any of its 1024 positions is
// still alive and returns false if none is.  Positions past the template's span may hold anything: they can only keep
// the wave going, the caller's scan ignores them.  Returns true with the complete sums.
//
// Round 3: the feature offsets arrive sorted by byte misalignment (k_prep_coarse_items), a batch (up to 8 features)
// never crosses a class boundary, and its features are summed as RAW dwords -- the 16 bytes at the 4-byte-aligned
// address, the dword after them (lane 63 loads it, every other lane takes its right neighbour's first dword: a lane
// permutation commutes with the sum, so ONE DPP move per batch) -- then re-aligned once (4 v_alignbyte per batch instead
// of per feature).  While no position can exceed 255 (<= 60 features) the totals stay packed bytes and the aliveness test
// is a carry-free byte trick: with K = 128 - bound, byte + K has its top bit set iff byte >= bound, and byte + K never
// exceeds 128 + (4 nf - rmin) (needs 4 nf - rmin <= 127: thresholds of ~70 % and up at 128 features; below that, and
// once a bound passes 128, the sums are widened to packed u16 as before).
// soff: the template's sorted offsets (out-of-image features point at the zero tail); sel0: the first 64 of them.
__device__ __forceinline__ bool accumulate_features16_pruned(const uint8_t* __restrict__ lm, const int32_t* __restrict__ soff, int sel0,
                                                             int nf, int e0, int e1, int e2, int j0, bool lane_on, int zero_off,
                                                             int rmin, uint32_t (&lo)[4], uint32_t (&hi)[4])
{
    const int lane = threadIdx.x & 63;
    const uint8_t* p = lm + j0;
#pragma unroll
    for (int i = 0; i < 4; ++i) lo[i] = hi[i] = 0;
    nf = __builtin_amdgcn_readfirstlane(nf);
    rmin = __builtin_amdgcn_readfirstlane(rmin);
    e0 = __builtin_amdgcn_readfirstlane(e0);
    e1 = __builtin_amdgcn_readfirstlane(e1);
    e2 = __builtin_amdgcn_readfirstlane(e2);
    const int slack = 4 * nf - rmin;
    uint32_t acc8[4] = {0, 0, 0, 0}; // packed-byte totals of the features since the last widening
    int in8 = 0;                     // features in acc8
    int sel = sel0;
    int g = 0;
    while (g < nf) {
        if (g && (g & 63) == 0) sel = g + lane < nf ? soff[g + lane] : zero_off; // next 64 offsets
        const int cls = g < e0 ? 0 : (g < e1 ? 1 : (g < e2 ? 2 : 3));
        const int cend = cls == 0 ? e0 : (cls == 1 ? e1 : (cls == 2 ? e2 : nf));
        int lim = cend - g;                 // features left in this class ...
        const int chunk_left = 64 - (g & 63); // ... and in this 64-offset chunk
        lim = lim < chunk_left ? lim : chunk_left;
        uint32_t raw[5] = {0, 0, 0, 0, 0};
        int n = 0;
        auto batch = [&](auto N) {
            constexpr int nn = decltype(N)::value;
            if (lane_on) {
                u128_a4 q[nn];
                uint32_t e[nn];
#pragma unroll
                for (int k = 0; k < nn; ++k) {
                    const int o = __builtin_amdgcn_readlane(sel, (g + k) & 63);
                    const uint8_t* a = p + (o & ~3);
                    q[k] = *(const u128_a4*)a;
                    e[k] = 0;
                    if (lane == 63) e[k] = *(const uint32_t*)(a + 16);
                }
#pragma unroll
                for (int k = 0; k < nn; ++k) { // <= 8 addends of <= 4: no carry between the bytes
                    raw[0] += q[k].x;
                    raw[1] += q[k].y;
                    raw[2] += q[k].z;
                    raw[3] += q[k].w;
                    raw[4] += e[k];
                }
            }
            n = nn;
        };
        if (lim >= FB16) batch(std::integral_constant<int, FB16>{});
        else if (lim >= 4) batch(std::integral_constant<int, 4>{});
        else if (lim >= 2) batch(std::integral_constant<int, 2>{});
        else batch(std::integral_constant<int, 1>{});
        // the dword after the lane's 16 bytes: the right neighbour's first dword (lane 63: what it loaded itself)
        const uint32_t nx = (uint32_t)__builtin_amdgcn_update_dpp((int)raw[4], (int)raw[0], 0x130, 0xf, 0xf, false);
        acc8[0] += __builtin_amdgcn_alignbyte(raw[1], raw[0], cls);
        acc8[1] += __builtin_amdgcn_alignbyte(raw[2], raw[1], cls);
        acc8[2] += __builtin_amdgcn_alignbyte(raw[3], raw[2], cls);
        acc8[3] += __builtin_amdgcn_alignbyte(nx, raw[3], cls);
        g += n;
        in8 += n;
        const int bound = rmin - 4 * (nf - g);
        const bool test = bound > 0 && g < nf;
        const bool in_bytes = test && slack <= 127 && bound <= 128 && in8 == g; // nothing widened yet: the totals are the packed bytes
        if (in_bytes) {
            const uint32_t K = (uint32_t)(128 - bound) * 0x01010101u; // byte + K <= 128 + slack: no carry between the bytes
            const uint32_t hit = ((acc8[0] + K) | (acc8[1] + K) | (acc8[2] + K) | (acc8[3] + K)) & 0x80808080u;
            if (__builtin_amdgcn_ballot_w64(hit != 0u) == 0ull) return false;
        }
        if ((test && !in_bytes) || in8 > 52) {
            // widen: packed bytes into the packed-u16 totals (a byte may hold up to 4 * 60)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                lo[i] += acc8[i] & 0x00ff00ffu;
                hi[i] += (acc8[i] >> 8) & 0x00ff00ffu;
                acc8[i] = 0;
            }
            in8 = 0;
            if (test && !in_bytes) {
                const uint32_t m = coarse_pk_max_u16(coarse_pk_max_u16(coarse_pk_max_u16(lo[0], hi[0]), coarse_pk_max_u16(lo[1], hi[1])),
                                                     coarse_pk_max_u16(coarse_pk_max_u16(lo[2], hi[2]), coarse_pk_max_u16(lo[3], hi[3])));
                const int best = (int)((m & 0xffff) > (m >> 16) ? (m & 0xffff) : (m >> 16));
                if (__builtin_amdgcn_ballot_w64(best >= bound) == 0ull) return false;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        lo[i] += acc8[i] & 0x00ff00ffu;
        hi[i] += (acc8[i] >> 8) & 0x00ff00ffu;
    }
    return true;
}

// plain byte copy (gathered match lists -> pinned host mirror); n multiple of 8 by construction
__global__ __launch_bounds__(256) void k_copy_bytes(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, size_t n)
{
    const size_t n8 = n / 8;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (size_t)gridDim.x * 256)
        ((uint64_t*)dst)[i] = ((const uint64_t*)src)[i];
    if (blockIdx.x == 0 && threadIdx.x < (n & 7)) dst[n8 * 8 + threadIdx.x] = src[n8 * 8 + threadIdx.x];
}

// zero the per-call counters (one launch instead of two memsets)
__global__ void k_reset(int32_t* __restrict__ counters, int32_t* __restrict__ out_count)
{
    if (threadIdx.x < 40) counters[threadIdx.x] = 0; // [0] candidates [2,3] refine bytes [4] arrivals [8..39] sub-arrivals
    if (threadIdx.x < 2) out_count[threadIdx.x] = 0;
}

// Which (position chunk, template slot, frame) a workgroup of the coarse pass works on.  Pure speed: any bijection
// of the grid is correct.
__device__ __forceinline__ void coarse_block_item(int& chunk_id, int& templ_slot, int& frame)
{
    chunk_id = blockIdx.x, templ_slot = blockIdx.y, frame = blockIdx.z;
    const bool frame_affinity = (gridDim.z & 7) == 0;
    if (frame_affinity) {
        // Workgroups are dealt to the 8 XCDs round-robin by linear id and each XCD has its own 4 MiB L2.  With a batch
        // of 8k frames give XCD x the frames x, x+8, ... one after the other: its L2 then holds ONE frame's linear
        // memories at a time (2 MiB on the bench configuration) instead of every frame's.
        const uint32_t lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
        const uint32_t xcd = lin & 7, j = lin >> 3, per = gridDim.x * gridDim.y;
        const uint32_t fl = j / per, rem = j - fl * per;
        frame = (int)(xcd + 8 * fl);
        templ_slot = (int)(rem / gridDim.x);
        chunk_id = (int)(rem - (uint32_t)templ_slot * gridDim.x);
    }
    // XCD-aware (chunk, template) assignment: give XCD x the position chunks [x*cpx, (x+1)*cpx) of EVERY
    // template, so that one L2 only ever sees the slice of the linear memories those chunks read
    // (chunk span + template extent) instead of all of them.
    if (!frame_affinity && (gridDim.x & 7) == 0) {
        const int lin = (int)(blockIdx.y * gridDim.x + blockIdx.x);
        const int cpx = (int)gridDim.x >> 3;
        const int xcd = lin & 7, slot = lin >> 3;
        templ_slot = slot / cpx;
        // the chunk ranges rotate over the XCDs every 64 templates: ranges past the templates' span are
        // empty, and this spreads them evenly while an XCD still works on one slice for 64 templates
        chunk_id = ((xcd + (templ_slot >> 6)) & 7) * cpx + slot % cpx;
    }
}

// The prefix length k1 and the partial-sum bound thr1 of the coarse pass's exact pruning.  A feature adds at most 4,
// so after the first k1 of nf features a position whose partial sum is below rmin - 4*(nf - k1) cannot reach rmin any
// more.  k1 is the shortest prefix (multiple of 4, at least 8) for which that bound is at least 62.5 % of the
// prefix's own maximum 4*k1 (a weaker bound prunes little on busy images); k1 == nf means no pruning.
__device__ __forceinline__ void coarse_prune_split(int nf, int rmin, int& k1, int& thr1)
{
    k1 = nf, thr1 = 0;
    if (rmin > 0) {
        int k = ((4 * nf - rmin) * 5 + 7) >> 3;
        k = ((k < 8 ? 8 : k) + 3) & ~3;
        if (4 * k <= 3 * nf) {
            k1 = k;
            thr1 = rmin - 4 * (nf - k1);
        }
    }
}

// grid = (position chunks of 1024, active templates); block = 4 waves.  Every
// wave covers the same 1024 positions (16 per lane) for a contiguous quarter
// of the template's features, so four times as many loads are in flight per
// position chunk; the partial sums meet in LDS and each wave then scans a
// quarter of the positions.  Fused threshold scan: no score map is written;
// candidates are appended with an atomic counter.
constexpr int COARSE_POS_PER_BLOCK = 1024;
__global__ __launch_bounds__(256) void k_similarity_coarse(
    const uint8_t* __restrict__ lm, int64_t lm_stride, int rows, int cols, int T, int W, int H, int L, int lc,
    const DevTL* __restrict__ tls, const uint32_t* __restrict__ fxy, const int32_t* __restrict__ foff,
    const int32_t* __restrict__ active, const int32_t* __restrict__ raw_min, const int32_t* __restrict__ raw_keep,
    const int32_t* __restrict__ class_idx, const int32_t* __restrict__ template_id, Cand* __restrict__ cands,
    int32_t* __restrict__ counters, int cap, int64_t lm_fs)
{
    // batch of frames: per-frame linear memories, candidate list and counters
    __shared__ uint32_t s_red[4][8][64];
    __shared__ int s_alive[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int chunk_id, templ_slot, frame;
    coarse_block_item(chunk_id, templ_slot, frame);
    lm += (size_t)frame * lm_fs;
    cands += (size_t)frame * cap;
    counters += (size_t)frame * 40;
    const int t = active[templ_slot];
    const DevTL tl = tls[(size_t)t * L + lc];
    const int npos = template_positions(tl, W, H, T);
    const int rmin = raw_min[(size_t)t * L + lc];
    const int base = chunk_id * COARSE_POS_PER_BLOCK;
    const int total = W * H;
    if (base >= total) return;
    if (base >= npos && rmin > 0) return; // beyond the span every score is 0
    const int j0 = base + lane * 16;
    const int zero_off = (int)(7 * lm_stride + (int64_t)T * T * W * H); // zero tail of the last orientation
    // lanes past the span take no part in the loads (one extra lane does: its first dword is its
    // left neighbour's bytes 16..19)
    const bool lane_on = j0 < npos + 16;
    // features [fb, fe) over the block's 1024 positions: a quarter of them per wave, partial sums meet in LDS;
    // returns the packed sums of positions j0 + 4*wave .. +3 of every lane
    auto partial = [&](int fb, int fe, uint32_t& slo, uint32_t& shi) {
        uint32_t lo[4] = {0, 0, 0, 0}, hi[4] = {0, 0, 0, 0};
        if (base < npos) { // block-uniform: every lane takes part (features are spread over the wave's lanes)
            const int chunk = (fe - fb + 3) >> 2;
            const int f0 = fb + wave * chunk;
            int cnt = fe - f0;
            cnt = cnt < 0 ? 0 : (cnt > chunk ? chunk : cnt);
            accumulate_features16(lm, fxy + tl.feat_off + f0, foff + tl.feat_off + f0, cnt, rows, cols, lane_on ? j0 : 0,
                                  lane_on, zero_off, lo, hi);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            s_red[wave][i][lane] = lo[i];
            s_red[wave][4 + i][lane] = hi[i];
        }
        __syncthreads();
        slo = s_red[0][wave][lane] + s_red[1][wave][lane] + s_red[2][wave][lane] + s_red[3][wave][lane];
        shi = s_red[0][4 + wave][lane] + s_red[1][4 + wave][lane] + s_red[2][4 + wave][lane] + s_red[3][4 + wave][lane];
    };
    // Exact pruning in two phases (coarse_prune_split): the block stops after the prefix unless one of its 1024
    // positions is still alive.  The candidate set is the same as without pruning: nothing is dropped that could
    // have reached rmin.
    int k1, thr1;
    coarse_prune_split(tl.nf, rmin, k1, thr1);
    uint32_t slo, shi;
    partial(0, k1, slo, shi);
    if (k1 < tl.nf) {
        bool alive = false;
#pragma unroll
        for (int k = 0; k < 4; ++k) alive |= (j0 + 4 * wave + k < npos) && unpack4(slo, shi, k) >= thr1;
        if (lane == 0) s_alive[wave] = 0;
        if (__builtin_amdgcn_ballot_w64(alive) != 0ull && lane == 0) s_alive[wave] = 1;
        __syncthreads(); // also: every wave has read s_red before the second phase rewrites it
        if ((s_alive[0] | s_alive[1] | s_alive[2] | s_alive[3]) == 0) return;
        uint32_t lo2, hi2;
        partial(k1, tl.nf, lo2, hi2);
        slo += lo2;
        shi += hi2;
    }
    const int offset = T / 2 + (T % 2 - 1);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int j = j0 + 4 * wave + k;
        int raw = unpack4(slo, shi, k);
        if (j >= npos) raw = 0;
        if (j < total && raw >= rmin) {
            int idx = atomicAdd(&counters[0], 1);
            if (idx < cap) {
                Cand c;
                c.t = t;
                c.x = (j % W) * T + offset;
                c.y = (j / W) * T + offset;
                c.raw = raw;
                cand_fill_next(c, tls, raw_keep, class_idx, template_id, L, lc - 1);
                cands[idx] = c;
            }
        }
    }
}

// The coarse pass for large launches (batches of frames, thousands of templates): one WAVE per (position chunk,
// template, frame), four such items per workgroup (template slots 4*blockIdx.y .. +3), no LDS and no barrier.
// With the exact pruning (accumulate_features16_pruned: re-checked after every 8 features) most waves stop after
// two or three batches, so splitting a template's features over four waves (the kernel above, which keeps
// single-frame latency low) only multiplies the per-item overhead.  Same arguments, same candidates.
__global__ __launch_bounds__(256) void k_similarity_coarse_wave(
    const uint8_t* __restrict__ lm, int64_t lm_stride, int rows, int cols, int T, int W, int H, int L, int lc,
    const DevTL* __restrict__ tls, const int32_t* __restrict__ soff, const CoarseItem* __restrict__ items,
    const int32_t* __restrict__ cfoff, int n_active, const int32_t* __restrict__ raw_keep,
    const int32_t* __restrict__ class_idx, const int32_t* __restrict__ template_id, Cand* __restrict__ cands,
    int32_t* __restrict__ counters, int cap, int64_t lm_fs)
{
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    int chunk_id, slot4, frame;
    coarse_block_item(chunk_id, slot4, frame);
    const int templ_slot = __builtin_amdgcn_readfirstlane(slot4 * 4 + wave);
    if (templ_slot >= n_active) return;
    // The item's dependent chain is two loads deep: {template record (scalar cache), its first 64 feature offsets
    // (vector load)} -- both indexed by the slot -- then the linear memories.  (Round 2: active[] -> tls[] -> raw_min[]
    // and fxy[] / foff[] -> linear memories: a third of the launch's time at threshold 100.)
    const int sel0 = cfoff[(size_t)templ_slot * 64 + lane];
    const CoarseItem it = items[templ_slot];
    lm += (size_t)frame * lm_fs;
    cands += (size_t)frame * cap;
    counters += (size_t)frame * 40;
    const int t = it.t;
    const int npos = it.npos;
    const int rmin = it.rmin;
    const int base = chunk_id * COARSE_POS_PER_BLOCK;
    const int total = W * H;
    if (base >= total) return;
    if (base >= npos && rmin > 0) return; // beyond the span every score is 0
    const int j0 = base + lane * 16;
    uint32_t lo[4] = {0, 0, 0, 0}, hi[4] = {0, 0, 0, 0};
    if (base < npos) {
        const int zero_off = (int)(7 * lm_stride + (int64_t)T * T * W * H);
        const bool lane_on = j0 < npos + 16;
        if (!accumulate_features16_pruned(lm, soff + it.soff_base, sel0, it.nf, it.e01 & 0xffff, (int)((uint32_t)it.e01 >> 16), it.e2,
                                          lane_on ? j0 : 0, lane_on, zero_off, rmin, lo, hi))
            return; // no position of this item can reach rmin
    }
    const int offset = T / 2 + (T % 2 - 1);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int j = j0 + 4 * i + k;
            int raw = unpack4(lo[i], hi[i], k);
            if (j >= npos) raw = 0;
            if (j < total && raw >= rmin) {
                int idx = atomicAdd(&counters[0], 1);
                if (idx < cap) {
                    Cand c;
                    c.t = t;
                    c.x = (j % W) * T + offset;
                    c.y = (j / W) * T + offset;
                    c.raw = raw;
                    cand_fill_next(c, tls, raw_keep, class_idx, template_id, L, lc - 1);
                    cands[idx] = c;
                }
            }
        }
    }
}

// same accumulation, score map out (stage entry point sbm_similarity)
__global__ __launch_bounds__(256) void k_similarity_map(const uint8_t* __restrict__ lm, int64_t lm_stride,
                                                        int rows, int cols, int T, int W, int H, DevTL tl,
                                                        const uint32_t* __restrict__ fxy,
                                                        const int32_t* __restrict__ foff,
                                                        uint16_t* __restrict__ dst)
{
    const int npos = template_positions(tl, W, H, T);
    const int total = W * H;
    const int base = blockIdx.x * SIM_POS_PER_BLOCK;
    const int j0 = base + threadIdx.x * 4;
    uint32_t lo = 0, hi = 0;
    if (base < npos) { // block-uniform, all lanes take part
        const uint8_t* zero_addr = lm + 7 * lm_stride + (int64_t)T * T * W * H;
        const int jl = j0 < npos ? j0 : 0;
        accumulate_features(lm, fxy + tl.feat_off, foff + tl.feat_off, 0, tl.nf, rows, cols, 0, 0, jl, zero_addr, 0, lo, hi);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int j = j0 + k;
        if (j < total) dst[j] = (uint16_t)(j < npos ? unpack4(lo, hi, k) : 0);
    }
}

// End of the emitting kernel: the last WORKING block to finish (arrival counter counters[4]) writes the
// overflow status and mirrors the final {count, overflow} pair to the caller's mirror (plain stores:
// it may be pinned host memory).  Only blocks that had a candidate arrive — atomics on one address
// serialise at ~25 ns each, so idle blocks must not queue up behind them; with no candidate at all
// block 0 publishes.  out_count itself is only ever updated with device-scope atomics, so the atomic
// read below sees every block's increments once all of them have arrived.
__device__ __forceinline__ void publish_counts(int32_t* __restrict__ counters, int n_all, int n_work_blocks, bool worked,
                                               int cand_cap, int32_t* __restrict__ out_count,
                                               int32_t* __restrict__ mirror_count, int slot)
{
    if (!worked && !(n_work_blocks == 0 && slot == 0)) return; // block-uniform
    __syncthreads();
    if (threadIdx.x == 0) {
        bool last = n_work_blocks == 0;
        if (!last) {
            // two-level arrival (32 sub-counters): arrivals on one address serialise at ~25 ns each
            const int sub = slot & 31;
            const int expect = (n_work_blocks - sub + 31) / 32; // working blocks with this residue
            const int groups = n_work_blocks < 32 ? n_work_blocks : 32;
            // no fence: the last block only reads out_count (device-scope atomics); the records themselves
            // become visible to the host and to later kernels at the end of the kernel
            if (atomicAdd(&counters[8 + sub], 1) == expect - 1) last = atomicAdd(&counters[4], 1) == groups - 1;
        }
        if (last) {
            const int n_out = atomicAdd(out_count, 0);
            const int flag = n_all > cand_cap ? 1 : 0;
            out_count[1] = flag;
            if (mirror_count) {
                mirror_count[0] = n_out;
                mirror_count[1] = flag;
            }
        }
    }
}

// ---- refinement pass on the strip-interleaved plane: four features per wave instruction ----------------------
// response4 with a per-lane orientation (the four 16-lane groups of a wave work on four different features)
__device__ __forceinline__ uint32_t response4v(uint32_t v, uint32_t o, uint32_t o1, uint32_t o7)
{
    const uint32_t self = (v >> o) & 0x01010101u;
    const uint32_t nb = ((v >> o1) | (v >> o7)) & (0x01010101u ^ self);
    return (self << 2) | (nb * 3u);
}

struct __attribute__((aligned(16))) u128_a16 { uint32_t x, y, z, w; };

// Lane = (feature slot g = lane >> 4, patch row r = lane & 15): a step of the wave adds FOUR features to the 16 x 16
// patch; a lane loads the two aligned 16-byte strip rows that hold its row of the patch (strip s and s + 1: 32 bytes
// of which the 16 starting at column c0 = gx0 & 15 are the patch row) and keeps 16 positions (4 packed dwords).
// The dword misalignment Q = c0 >> 2 picks WHICH of the eight loaded dwords are used and must not cost a per-lane
// select: the host sorts every template's features by (x / T) & 15, the class that fixes c0 for a given candidate
// (c0 = (class + ox / T) & 15), and wave w of the workgroup takes the four (cyclically consecutive) classes whose Q
// is w & 3 -- Q is a template parameter of the body.  Per feature that is 2 loads per FOUR features instead of 2 per
// feature, no per-feature scalar address arithmetic, and ~15 vector instructions instead of ~25.
// cls[0..16]: class start offsets inside the (sorted) feature range of this template level.
// Result: lo[i] / hi[i] = packed u16 sums of columns 4i .. 4i+3 of row r, valid in lanes 0..15 of every wave
// (the wave's share of the features), to be added over the workgroup's waves.
template <int LOG2T, int Q, int NPARTS>
__device__ __forceinline__ void accumulate_rows16_q(const uint8_t* __restrict__ lmc, const uint32_t* __restrict__ fxy,
                                                    const uint8_t* __restrict__ flabel, const uint16_t* __restrict__ cls,
                                                    int part, int rows, int cols, int ox, int oy, int W, int H,
                                                    uint32_t (&lo)[4], uint32_t (&hi)[4])
{
    constexpr int T = 1 << LOG2T;
    const int lane = threadIdx.x & 63;
    const int g = lane >> 4, r = lane & 15;
    const int X15 = (ox >> LOG2T) & 15;
    const int k0 = (4 * Q - X15) & 15; // classes k0 .. k0+3 (mod 16) end up with dword misalignment Q
    int a0 = cls[k0], a1, b1 = 0;      // up to two runs of the sorted list: [a0, a1) and [0, b1)
    if (k0 <= 12) a1 = cls[k0 + 4];
    else a1 = cls[16], b1 = cls[k0 - 12];
    a0 = __builtin_amdgcn_readfirstlane(a0);
    a1 = __builtin_amdgcn_readfirstlane(a1);
    b1 = __builtin_amdgcn_readfirstlane(b1);
    const int len1 = a1 - a0, n = len1 + b1;
    const int per = ((((n + 3) >> 2) + NPARTS - 1) / NPARTS) << 2; // features per part (whole steps)
    int j0 = per * part, j1 = j0 + per;
    j0 = j0 < n ? j0 : n;
    j1 = j1 < n ? j1 : n;
    const uint32_t zero16 = (uint32_t)(((int64_t)T * T * W * H) >> 4); // the plane's zero tail (>= H*16 + 272 bytes)
    const uint32_t pitch = (uint32_t)H * 16u;                         // strip s -> strip s + 1, same row
    const uint32_t row_off = (uint32_t)r * 16u;
    uint32_t acc[4] = {0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 4; ++i) lo[i] = hi[i] = 0;
    int pending = 0;
    for (int jb = j0; jb < j1; jb += 64) {
        // lane = feature: (strip row address / 16, column misalignment | label)
        uint32_t recA = zero16, recB = 0;
        const int jj = jb + lane;
        if (jj < j1) {
            const int idx = jj < len1 ? a0 + jj : jj - len1;
            const uint32_t xy = fxy[idx];
            const uint32_t lab = flabel[idx];
            const int x = (int)(xy & 0xffff) + ox, y = (int)(xy >> 16) + oy;
            const int gx0 = x >> LOG2T, gy0 = y >> LOG2T;
            if (x >= 0 && y >= 0 && x < cols && y < rows)
                recA = (uint32_t)(lm_strip_offset(((y & (T - 1)) << LOG2T) | (x & (T - 1)), gy0, gx0 & ~15, W, H) >> 4);
            recB = (uint32_t)(gx0 & 3) | (lab << 2);
        }
        const int nbf = j1 - jb < 64 ? j1 - jb : 64;
        auto batch = [&](auto N, int u) { // N steps = 4 N features, u = first feature of the batch (multiple of 4)
            constexpr int nst = decltype(N)::value;
            u128_a16 A[nst], B[nst];
            uint32_t rb[nst];
#pragma unroll
            for (int k = 0; k < nst; ++k) {
                const int f = (u + 4 * k + g) << 2; // lanes past the last feature hold the zero record
                const uint32_t a = (uint32_t)__builtin_amdgcn_ds_bpermute(f, (int)recA);
                rb[k] = (uint32_t)__builtin_amdgcn_ds_bpermute(f, (int)recB);
                const uint8_t* p = lmc + ((size_t)a << 4) + row_off;
                A[k] = *(const u128_a16*)p;
                B[k] = *(const u128_a16*)(p + pitch);
            }
#pragma unroll
            for (int k = 0; k < nst; ++k) {
                const uint32_t sh = rb[k] & 3u, o = rb[k] >> 2, o1 = (o + 1u) & 7u, o7 = (o + 7u) & 7u;
                const uint32_t d0 = Q == 0 ? A[k].x : Q == 1 ? A[k].y : Q == 2 ? A[k].z : A[k].w;
                const uint32_t d1 = Q == 0 ? A[k].y : Q == 1 ? A[k].z : Q == 2 ? A[k].w : B[k].x;
                const uint32_t d2 = Q == 0 ? A[k].z : Q == 1 ? A[k].w : Q == 2 ? B[k].x : B[k].y;
                const uint32_t d3 = Q == 0 ? A[k].w : Q == 1 ? B[k].x : Q == 2 ? B[k].y : B[k].z;
                const uint32_t d4 = Q == 0 ? B[k].x : Q == 1 ? B[k].y : Q == 2 ? B[k].z : B[k].w;
                acc[0] += response4v(__builtin_amdgcn_alignbyte(d1, d0, sh), o, o1, o7);
                acc[1] += response4v(__builtin_amdgcn_alignbyte(d2, d1, sh), o, o1, o7);
                acc[2] += response4v(__builtin_amdgcn_alignbyte(d3, d2, sh), o, o1, o7);
                acc[3] += response4v(__builtin_amdgcn_alignbyte(d4, d3, sh), o, o1, o7);
            }
            pending += nst;
            if (pending + 4 > 63) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    lo[i] += acc[i] & 0x00ff00ffu;
                    hi[i] += (acc[i] >> 8) & 0x00ff00ffu;
                    acc[i] = 0;
                }
                pending = 0;
            }
        };
        int u = 0;
        for (; u + 16 <= nbf; u += 16) batch(std::integral_constant<int, 4>{}, u);
        if (nbf - u > 8) { batch(std::integral_constant<int, 2>{}, u); u += 8; }
        if (nbf - u > 4) { batch(std::integral_constant<int, 2>{}, u); u += 8; }
        else if (nbf - u > 0) batch(std::integral_constant<int, 1>{}, u);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        lo[i] += acc[i] & 0x00ff00ffu;
        hi[i] += (acc[i] >> 8) & 0x00ff00ffu;
    }
}

// 16x16 patch of one candidate: lane = (row = lane>>2, 4 columns); the block's
// LOCAL_WAVES waves each take a contiguous slice of the features and the partial
// sums meet in LDS.  Result (packed u16) valid in wave 0.
constexpr int LOCAL_WAVES = 16;
template <int COMPACT = 0, int LW = LOCAL_WAVES>
__device__ __forceinline__ void local_patch(const uint8_t* __restrict__ lm, int64_t lm_stride, const DevTL tl,
                                            const uint32_t* __restrict__ fxy, const int32_t* __restrict__ foff,
                                            int rows, int cols, int W, int H, int T, int ox, int oy,
                                            uint32_t (*s_part)[2][64], uint32_t& lo, uint32_t& hi,
                                            const uint8_t* __restrict__ flabel = nullptr,
                                            const uint16_t* __restrict__ cls = nullptr)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (COMPACT == 2) {
        // strip plane: fxy / flabel are the class-sorted copies, cls the 17 class offsets of this template level;
        // wave w takes the features whose dword misalignment is w & 3 (and, with 16 waves, a quarter of those)
        static_assert(COMPACT != 2 || (LW & 3) == 0, "one wave (or LW / 4) per dword misalignment");
        const int q = __builtin_amdgcn_readfirstlane(wave) & 3, part = __builtin_amdgcn_readfirstlane(wave) >> 2;
        uint32_t l4[4], h4[4];
        const uint32_t* fx = fxy + tl.feat_off;
        const uint8_t* fl = flabel + tl.feat_off;
#define SBM_ROWS16(LOG2T_)                                                                                              \
        do {                                                                                                            \
            if (q == 0) accumulate_rows16_q<LOG2T_, 0, LW / 4>(lm, fx, fl, cls, part, rows, cols, ox, oy, W, H, l4, h4);      \
            else if (q == 1) accumulate_rows16_q<LOG2T_, 1, LW / 4>(lm, fx, fl, cls, part, rows, cols, ox, oy, W, H, l4, h4); \
            else if (q == 2) accumulate_rows16_q<LOG2T_, 2, LW / 4>(lm, fx, fl, cls, part, rows, cols, ox, oy, W, H, l4, h4); \
            else accumulate_rows16_q<LOG2T_, 3, LW / 4>(lm, fx, fl, cls, part, rows, cols, ox, oy, W, H, l4, h4);             \
        } while (0)
        if (T == 4) SBM_ROWS16(2);
        else SBM_ROWS16(3);
#undef SBM_ROWS16
        // the four feature slots of the wave (lanes r, r+16, r+32, r+48), then the waves of the workgroup; the last
        // read also turns (row, 16 columns) per lane into the (row, 4 columns) per lane the caller scans
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            l4[i] += (uint32_t)__shfl_xor((int)l4[i], 16, 64);
            h4[i] += (uint32_t)__shfl_xor((int)h4[i], 16, 64);
            l4[i] += (uint32_t)__shfl_xor((int)l4[i], 32, 64);
            h4[i] += (uint32_t)__shfl_xor((int)h4[i], 32, 64);
        }
        uint32_t(*s8)[8][16] = (uint32_t(*)[8][16])s_part; // same bytes: [LW][2][64] == [LW][8][16]
        if (lane < 16) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                s8[wave][i][lane] = l4[i];
                s8[wave][4 + i][lane] = h4[i];
            }
        }
        __syncthreads();
        if (wave == 0) {
            lo = hi = 0;
#pragma unroll
            for (int w = 0; w < LW; ++w) {
                lo += s8[w][lane & 3][lane >> 2];
                hi += s8[w][4 + (lane & 3)][lane >> 2];
            }
        }
        __syncthreads();
        return;
    }
    const int r = lane >> 2, c4 = (lane & 3) * 4;
    const int delta = (oy / T) * W + ox / T + r * W + c4;
    const uint8_t* zero_addr = lm + (COMPACT ? 0 : 7) * lm_stride + (int64_t)T * T * W * H;
    const int chunk = (tl.nf + LW - 1) / LW;
    const int f0 = wave * chunk;
    int cnt = tl.nf - f0;
    cnt = cnt < 0 ? 0 : (cnt > chunk ? chunk : cnt);
    accumulate_features<COMPACT != 0>(lm, fxy + tl.feat_off, foff + tl.feat_off, f0, cnt, rows, cols, ox, oy, delta, zero_addr, 0, lo, hi,
                                      COMPACT ? flabel + tl.feat_off : nullptr, (int)lm_stride);
    s_part[wave][0][lane] = lo;
    s_part[wave][1][lane] = hi;
    __syncthreads();
    if (wave == 0) {
        lo = hi = 0;
#pragma unroll
        for (int w = 0; w < LW; ++w) {
            lo += s_part[w][0][lane];
            hi += s_part[w][1][lane];
        }
    }
    __syncthreads();
}

template <int COMPACT, int LW>
__global__ __launch_bounds__(64 * LW) void k_similarity_local(
    const uint8_t* __restrict__ lm, int64_t lm_stride, int rows, int cols, int T, int W, int H, int L, int l,
    const DevTL* __restrict__ tls, const uint32_t* __restrict__ fxy, const int32_t* __restrict__ foff,
    const int32_t* __restrict__ raw_keep, const int32_t* __restrict__ class_idx,
    const int32_t* __restrict__ template_id, Cand* __restrict__ cands, int32_t* __restrict__ counters,
    int cand_cap, int is_last, sbm_match_rec* __restrict__ out, int32_t* __restrict__ out_count,
    int out_cap, sbm_match_rec* __restrict__ mirror_out, int32_t* __restrict__ mirror_count, int collect_stats,
    int64_t lm_fs, const uint8_t* __restrict__ flabel, const uint16_t* __restrict__ fcls)
{
    // grid = (frames, candidate slots): the frame is the FAST grid dimension, so the blocks dispatched first are
    // the low slots of every frame -- the ones that have a candidate -- and the idle slots come last
    const int slot = blockIdx.y, n_slots = gridDim.y;
    {
        const size_t frame = blockIdx.x;
        lm += frame * lm_fs;
        cands += frame * cand_cap;
        counters += frame * 40;
        out += frame * out_cap;
        out_count += frame * 2;
        if (mirror_out) mirror_out += frame * out_cap;
        if (mirror_count) mirror_count += frame * 2;
    }
    __shared__ uint32_t s_part[LW][2][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // the candidate count and this block's first candidate are fetched together (the record is read
    // speculatively: the buffer always exists, the count decides whether it is used)
    Cand c = cands[slot < cand_cap ? slot : 0];
    const int n_all = counters[0];
    const int n = n_all < cand_cap ? n_all : cand_cap;
    const int border = 8 * T, offset = T / 2 + (T % 2 - 1);
    for (int ci = slot; ci < n; ci += n_slots) {
        if (ci != slot) c = cands[ci];
        if (c.raw < 0) continue; // dropped at a coarser level (uniform per block)
        DevTL tl;
        tl.width = c.next_width;
        tl.height = c.next_height;
        tl.nf = c.next_nf;
        tl.feat_off = c.next_feat_off;
        int x = c.x * 2 + 1, y = c.y * 2 + 1;
        const int max_x = cols - tl.width - border, max_y = rows - tl.height - border;
        x = x < border ? border : x;
        y = y < border ? border : y;
        x = x > max_x ? max_x : x;
        y = y > max_y ? max_y : y;
        const int ox = (x / T - 8) * T, oy = (y / T - 8) * T;
        uint32_t lo, hi;
        local_patch<COMPACT, LW>(lm, lm_stride, tl, fxy, foff, rows, cols, W, H, T, ox, oy, s_part, lo, hi, flabel,
                                 COMPACT == 2 ? fcls + ((size_t)c.t * L + l) * 17 : nullptr);
        if (wave != 0) continue;
        // first maximum in row-major order, strict '>' from 0 (:1265-1282): maximise (raw, -position)
        uint32_t best = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            uint32_t raw = (uint32_t)unpack4(lo, hi, k);
            uint32_t p = (uint32_t)(lane * 4 + k);
            uint32_t key = (raw << 8) | (255u - p);
            best = key > best ? key : best;
        }
#pragma unroll
        for (int s = 32; s >= 1; s >>= 1) {
            uint32_t o = __shfl_xor(best, s, 64);
            best = o > best ? o : best;
        }
        if (lane == 0) {
            if (collect_stats) atomicAdd((unsigned long long*)(counters + 2), (unsigned long long)tl.nf * 256ull); // refinement bytes
            const int raw = (int)(best >> 8);
            int br = -1, bc = -1;
            if (raw > 0) {
                int p = 255 - (int)(best & 255u);
                br = p >> 4;
                bc = p & 15;
            }
            const int nx = (x / T - 8 + bc) * T + offset;
            const int ny = (y / T - 8 + br) * T + offset;
            const bool keep = raw >= c.next_keep;
            if (is_last) {
                if (keep) {
                    int idx = atomicAdd(out_count, 1);
                    if (idx < out_cap) {
                        sbm_match_rec m;
                        m.x = nx;
                        m.y = ny;
                        m.similarity = __fdiv_rn(__fmul_rn((float)raw, 100.f), (float)(4 * tl.nf));
                        m.raw = raw;
                        m.class_idx = c.class_idx;
                        m.template_id = c.template_id;
                        out[idx] = m;
                        if (mirror_out) mirror_out[idx] = m; // e.g. pinned host memory: no copy engine involved
                    }
                }
            } else {
                c.x = nx;
                c.y = ny;
                c.raw = keep ? raw : -1;
                if (keep) cand_fill_next(c, tls, raw_keep, class_idx, template_id, L, l - 1);
                cands[ci] = c;
            }
        }
    }
    if (is_last) {
        const int n_work = n < n_slots ? n : n_slots;
        publish_counts(counters, n_all, n_work, slot < n, cand_cap, out_count, mirror_count, slot);
    }
}

// stage entry point: one 16x16 patch
__global__ __launch_bounds__(64 * LOCAL_WAVES) void k_similarity_local_patch(const uint8_t* __restrict__ lm, int64_t lm_stride,
                                                                int rows, int cols, int T, int W, int H, DevTL tl,
                                                                const uint32_t* __restrict__ fxy,
                                                                const int32_t* __restrict__ foff, int cx, int cy,
                                                                uint16_t* __restrict__ dst)
{
    __shared__ uint32_t s_part[LOCAL_WAVES][2][64];
    const int ox = (cx / T - 8) * T, oy = (cy / T - 8) * T;
    uint32_t lo, hi;
    local_patch<0, LOCAL_WAVES>(lm, lm_stride, tl, fxy, foff, rows, cols, W, H, T, ox, oy, s_part, lo, hi);
    if (threadIdx.x < 64) {
#pragma unroll
        for (int k = 0; k < 4; ++k) dst[threadIdx.x * 4 + k] = (uint16_t)unpack4(lo, hi, k);
    }
}

// single-level pyramids: coarse candidates are the final matches
__global__ __launch_bounds__(256) void k_emit_coarse(const Cand* __restrict__ cands,
                                                     int32_t* __restrict__ counters, int cand_cap,
                                                     const DevTL* __restrict__ tls, int L, int lc,
                                                     const int32_t* __restrict__ class_idx,
                                                     const int32_t* __restrict__ template_id,
                                                     sbm_match_rec* __restrict__ out,
                                                     int32_t* __restrict__ out_count, int out_cap,
                                                     sbm_match_rec* __restrict__ mirror_out,
                                                     int32_t* __restrict__ mirror_count)
{
    { // batch of frames: frame = blockIdx.y
        const size_t frame = blockIdx.y;
        cands += frame * cand_cap;
        counters += frame * 40;
        out += frame * out_cap;
        out_count += frame * 2;
        if (mirror_out) mirror_out += frame * out_cap;
        if (mirror_count) mirror_count += frame * 2;
    }
    const int n_all = counters[0];
    const int n = n_all < cand_cap ? n_all : cand_cap;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const Cand c = cands[i];
        const DevTL tl = tls[(size_t)c.t * L + lc];
        int idx = atomicAdd(out_count, 1);
        if (idx < out_cap) {
            sbm_match_rec m;
            m.x = c.x;
            m.y = c.y;
            m.similarity = __fdiv_rn(__fmul_rn((float)c.raw, 100.f), (float)(4 * tl.nf));
            m.raw = c.raw;
            m.class_idx = class_idx[c.t];
            m.template_id = template_id[c.t];
            out[idx] = m;
            if (mirror_out) mirror_out[idx] = m;
        }
    }
    {
        const int per = (int)gridDim.x * 256;
        const int n_work = (n + 255) / 256 < (int)gridDim.x ? (n + 255) / 256 : (int)gridDim.x;
        (void)per;
        publish_counts(counters, n_all, n_work, (int)blockIdx.x < n_work, cand_cap, out_count, mirror_count, (int)blockIdx.x);
    }
}

} // namespace sbm
