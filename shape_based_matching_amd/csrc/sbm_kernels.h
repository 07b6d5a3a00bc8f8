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
#include "../../include/sbm_types.h"

namespace sbm {

// per (template, level) record on the device
struct DevTL {
    int32_t width, height, nf, feat_off;
};
// coarse candidate / refinement state (Match under construction)
struct Cand {
    int32_t t;   // template index (into the uploaded list)
    int32_t x;   // Match::x at the level just processed
    int32_t y;
    int32_t raw; // integer similarity sum; < 0 = dropped by the per-level filter
};

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
__device__ __forceinline__ int reflect101(int p, int len)
{
    if (len == 1) return 0;
    while (p < 0 || p >= len) p = p < 0 ? -p : 2 * len - 2 - p;
    return p;
}

typedef uint32_t __attribute__((aligned(1))) u32_unaligned;
__device__ __forceinline__ uint32_t ld_u32_any(const uint8_t* p) { return *(const u32_unaligned*)p; }

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

// One block = one 16x64 output tile.  Source tile -> LDS, separable 7-tap
// fixed-point Gaussian through LDS, Sobel + magnitude + orientation bin into
// LDS, 3x3 majority vote, one-hot bytes out (4 per lane).
template <int CH>
__global__ __launch_bounds__(256) void k_quantize(const uint8_t* __restrict__ img, int rows, int cols,
                                                  int stride, const uint8_t* __restrict__ mask,
                                                  float thr_sq, uint8_t* __restrict__ out,
                                                  float* __restrict__ mag_out, float* __restrict__ ori_out)
{
    __shared__ uint8_t s_src[QS_R][QS_C * CH];
    __shared__ uint16_t s_h[QS_R][QM_C * CH];
    __shared__ uint8_t s_sm[QM_R][QM_C * CH];
    __shared__ uint8_t s_q[QQ_R][QQ_C];
    const int tid = threadIdx.x;
    const int R0 = blockIdx.y * QT_R, C0 = blockIdx.x * QT_C;
    const int K[7] = {8, 28, 56, 72, 56, 28, 8};

    // A: source tile, BORDER_REPLICATE by clamping the coordinates
    for (int idx = tid; idx < QS_R * QS_C * CH; idx += 256) {
        int r = idx / (QS_C * CH), rem = idx - r * (QS_C * CH);
        int c = rem / CH, k = rem - c * CH;
        int gr = clampi(R0 - 5 + r, 0, rows - 1), gc = clampi(C0 - 5 + c, 0, cols - 1);
        s_src[r][rem] = img[(size_t)gr * stride + gc * CH + k];
    }
    __syncthreads();
    // B: horizontal 7-tap, exact 8.8 fixed point, evaluated AT the clamped column
    for (int idx = tid; idx < QS_R * QM_C * CH; idx += 256) {
        int r = idx / (QM_C * CH), rem = idx - r * (QM_C * CH);
        int j = rem / CH, k = rem - j * CH;
        int cc = clampi(C0 - 2 + j, 0, cols - 1);
        int base = cc - C0 + 2;
        int acc = 0;
#pragma unroll
        for (int i = 0; i < 7; ++i) acc += K[i] * s_src[r][(base + i) * CH + k];
        s_h[r][rem] = (uint16_t)acc;
    }
    __syncthreads();
    // C: vertical 7-tap, one rounding: (sum + 2^15) >> 16
    for (int idx = tid; idx < QM_R * QM_C * CH; idx += 256) {
        int jr = idx / (QM_C * CH), x = idx - jr * (QM_C * CH);
        int rr = clampi(R0 - 2 + jr, 0, rows - 1);
        int rb = rr - R0 + 2;
        uint32_t acc = 0;
#pragma unroll
        for (int j = 0; j < 7; ++j) acc += (uint32_t)K[j] * s_h[rb + j][x];
        s_sm[jr][x] = (uint8_t)((acc + 32768u) >> 16);
    }
    __syncthreads();
    // D: Sobel 3x3 on the (replicate-clamped) smoothed tile, magnitude, phase, 16->8 bins
    for (int idx = tid; idx < QQ_R * QQ_C; idx += 256) {
        int qr = idx / QQ_C, qc = idx - qr * QQ_C;
        int r = R0 - 1 + qr, c = C0 - 1 + qc;
        uint8_t code = 0;
        if (r >= 0 && r < rows && c >= 0 && c < cols) {
            const int jr = qr + 1, jc = qc + 1;
            int bx = 0, by = 0, bm = -1;
#pragma unroll
            for (int k = 0; k < CH; ++k) {
                int a00 = s_sm[jr - 1][(jc - 1) * CH + k], a01 = s_sm[jr - 1][jc * CH + k], a02 = s_sm[jr - 1][(jc + 1) * CH + k];
                int a10 = s_sm[jr][(jc - 1) * CH + k], a12 = s_sm[jr][(jc + 1) * CH + k];
                int a20 = s_sm[jr + 1][(jc - 1) * CH + k], a21 = s_sm[jr + 1][jc * CH + k], a22 = s_sm[jr + 1][(jc + 1) * CH + k];
                int gx = (a02 - a00) + 2 * (a12 - a10) + (a22 - a20);
                int gy = (a20 - a00) + 2 * (a21 - a01) + (a22 - a02);
                int m = gx * gx + gy * gy;
                // channel of maximum magnitude, ties to the lower channel index (:370-387)
                if (m > bm) {
                    bm = m;
                    bx = gx;
                    by = gy;
                }
            }
            float fx = (float)bx, fy = (float)by;
            float mag = (float)bm;
            float ang = fast_atan2_deg(fy, fx);
            int q16 = __float2int_rn(__fmul_rn(ang, (float)(16.0 / 360.0)));
            q16 = q16 < 0 ? 0 : (q16 > 255 ? 255 : q16);
            bool ring = (r == 0) || (r == rows - 1) || (c == 0) || (c == cols - 1);
            int q = ring ? 0 : (q16 & 7);
            code = (uint8_t)(q | ((mag > thr_sq) ? 8 : 0));
            if (qr >= 1 && qr <= QT_R && qc >= 1 && qc <= QT_C) {
                if (mag_out) mag_out[(size_t)r * cols + c] = mag;
                if (ori_out) ori_out[(size_t)r * cols + c] = ang;
            }
        }
        s_q[qr][qc] = code;
    }
    __syncthreads();
    // E: 3x3 majority vote (>= 5 of 9), one-hot byte; 4 pixels per lane
    {
        const int er = tid >> 4, c4 = (tid & 15) * 4;
        const int r = R0 + er;
        if (r < rows) {
            uint32_t packed = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int c = C0 + c4 + k;
                uint32_t v = 0;
                if (c < cols && r >= 1 && r < rows - 1 && c >= 1 && c < cols - 1) {
                    const int qr = er + 1, qc = c4 + k + 1;
                    if (s_q[qr][qc] & 8) {
                        uint32_t h = 0; // eight 4-bit vote counters
#pragma unroll
                        for (int dr = -1; dr <= 1; ++dr)
#pragma unroll
                            for (int dc = -1; dc <= 1; ++dc) h += 1u << (4 * (s_q[qr + dr][qc + dc] & 7));
#pragma unroll
                        for (int i = 0; i < 8; ++i)
                            if (((h >> (4 * i)) & 15u) >= 5u) v = 1u << i;
                    }
                    if (mask && !mask[(size_t)r * cols + c]) v = 0;
                }
                packed |= v << (8 * k);
            }
            uint8_t* o = out + (size_t)r * cols + C0 + c4;
            if (C0 + c4 + 3 < cols && ((((size_t)r * cols + C0 + c4) & 3) == 0)) {
                *(uint32_t*)o = packed;
            } else {
                for (int k = 0; k < 4; ++k)
                    if (C0 + c4 + k < cols) o[k] = (uint8_t)(packed >> (8 * k));
            }
        }
    }
}

// cv::pyrDown: [1 4 6 4 1]^2, (sum + 128) >> 8, REFLECT_101; dst = (rows/2, cols/2)
__global__ __launch_bounds__(256) void k_pyrdown(const uint8_t* __restrict__ src, int rows, int cols, int ch,
                                                 int stride, uint8_t* __restrict__ dst)
{
    const int dr = rows / 2, dc = cols / 2;
    const int n = dr * dc * ch;
    const int K[5] = {1, 4, 6, 4, 1};
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < n; idx += gridDim.x * 256) {
        int k = idx % ch, x = (idx / ch) % dc, y = idx / (ch * dc);
        int acc = 0;
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const uint8_t* s = src + (size_t)reflect101(2 * y + j - 2, rows) * stride;
            int h = 0;
#pragma unroll
            for (int i = 0; i < 5; ++i) h += K[i] * s[reflect101(2 * x + i - 2, cols) * ch + k];
            acc += K[j] * h;
        }
        dst[idx] = (uint8_t)((acc + 128) >> 8);
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
                                                       int32_t* __restrict__ foff)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int l = flevel[i];
        const int T = Tl[l], W = Wl[l], H = Hl[l];
        const int x = fxy[i] & 0xffff, y = fxy[i] >> 16;
        int64_t off = (int64_t)flabel[i] * stride_l[l] + (int64_t)((y % T) * T + (x % T)) * W * H + (int64_t)(y / T) * W + x / T;
        foff[i] = (int32_t)off;
    }
}

__device__ __forceinline__ int template_positions(const DevTL& tl, int W, int H, int T)
{
    int wf = (tl.width - 1) / T + 1, hf = (tl.height - 1) / T + 1;
    return (H - hf) * W + (W - wf) + 1;
}

constexpr int SIM_POS_PER_BLOCK = 1024; // 256 lanes x 4 positions

// Accumulate 4 consecutive positions per lane.  Packed-u8 adds are carry-free
// for up to 63 features (63 * 4 = 252); flushed into two packed-u16 words.
__device__ __forceinline__ void coarse_accumulate(const uint8_t* __restrict__ lm, const DevTL tl,
                                                  const uint32_t* __restrict__ fxy,
                                                  const int32_t* __restrict__ foff, int rows, int cols,
                                                  int j0, uint32_t& lo, uint32_t& hi)
{
    uint32_t acc = 0;
    lo = hi = 0;
    int pending = 0;
    for (int f = 0; f < tl.nf; ++f) {
        const uint32_t xy = fxy[tl.feat_off + f];
        const int x = xy & 0xffff, y = xy >> 16;
        if (x >= cols || y >= rows) continue; // out-of-image features are skipped (:836-837)
        acc += ld_u32_any(lm + foff[tl.feat_off + f] + j0);
        if (++pending == 63) {
            lo += acc & 0x00ff00ffu;
            hi += (acc >> 8) & 0x00ff00ffu;
            acc = 0;
            pending = 0;
        }
    }
    lo += acc & 0x00ff00ffu;
    hi += (acc >> 8) & 0x00ff00ffu;
}

// grid = (position chunks, active templates).  Fused threshold scan: no score
// map is written; candidates are appended with one atomic per wave.
__global__ __launch_bounds__(256) void k_similarity_coarse(
    const uint8_t* __restrict__ lm, int rows, int cols, int T, int W, int H, int L, int lc,
    const DevTL* __restrict__ tls, const uint32_t* __restrict__ fxy, const int32_t* __restrict__ foff,
    const int32_t* __restrict__ active, const int32_t* __restrict__ raw_min, Cand* __restrict__ cands,
    int32_t* __restrict__ counters, int cap)
{
    const int t = active[blockIdx.y];
    const DevTL tl = tls[(size_t)t * L + lc];
    const int npos = template_positions(tl, W, H, T);
    const int rmin = raw_min[(size_t)t * L + lc];
    const int base = blockIdx.x * SIM_POS_PER_BLOCK;
    const int total = W * H;
    if (base >= total) return;
    if (base >= npos && rmin > 0) return; // beyond the span every score is 0
    const int j0 = base + threadIdx.x * 4;
    uint32_t lo = 0, hi = 0;
    if (j0 < npos) coarse_accumulate(lm, tl, fxy, foff, rows, cols, j0, lo, hi);
    const int offset = T / 2 + (T % 2 - 1);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int j = j0 + k;
        int raw = (k & 1) ? ((hi >> (8 * (k - 1))) & 0xffff) : ((lo >> (8 * k)) & 0xffff);
        if (j >= npos) raw = 0;
        if (j < total && raw >= rmin) {
            int idx = atomicAdd(&counters[0], 1);
            if (idx < cap) {
                Cand c;
                c.t = t;
                c.x = (j % W) * T + offset;
                c.y = (j / W) * T + offset;
                c.raw = raw;
                cands[idx] = c;
            }
        }
    }
}

// same accumulation, score map out (stage entry point sbm_similarity)
__global__ __launch_bounds__(256) void k_similarity_map(const uint8_t* __restrict__ lm, int rows, int cols,
                                                        int T, int W, int H, DevTL tl,
                                                        const uint32_t* __restrict__ fxy,
                                                        const int32_t* __restrict__ foff,
                                                        uint16_t* __restrict__ dst)
{
    const int npos = template_positions(tl, W, H, T);
    const int total = W * H;
    const int j0 = blockIdx.x * SIM_POS_PER_BLOCK + threadIdx.x * 4;
    if (j0 >= total) return;
    uint32_t lo = 0, hi = 0;
    if (j0 < npos) coarse_accumulate(lm, tl, fxy, foff, rows, cols, j0, lo, hi);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int j = j0 + k;
        int raw = (k & 1) ? ((hi >> (8 * (k - 1))) & 0xffff) : ((lo >> (8 * k)) & 0xffff);
        if (j < total) dst[j] = (uint16_t)(j < npos ? raw : 0);
    }
}

// 16x16 patch of one candidate held by one wave: lane = (row = lane>>2,
// 4 columns).  Returns packed u16 sums in lo (cols 0,2) / hi (cols 1,3).
__device__ __forceinline__ void local_accumulate(const uint8_t* __restrict__ lm, const DevTL tl,
                                                 const uint32_t* __restrict__ fxy,
                                                 const int32_t* __restrict__ foff, int rows, int cols,
                                                 int W, int T, int ox, int oy, int lane, uint32_t& lo,
                                                 uint32_t& hi)
{
    const int r = lane >> 2, c4 = (lane & 3) * 4;
    const int delta = (oy / T) * W + ox / T + r * W + c4;
    uint32_t acc = 0;
    lo = hi = 0;
    int pending = 0;
    for (int f = 0; f < tl.nf; ++f) {
        const uint32_t xy = fxy[tl.feat_off + f];
        const int x = (int)(xy & 0xffff) + ox, y = (int)(xy >> 16) + oy;
        if (x < 0 || y < 0 || x >= cols || y >= rows) continue; // :878-879
        acc += ld_u32_any(lm + foff[tl.feat_off + f] + delta);
        if (++pending == 63) {
            lo += acc & 0x00ff00ffu;
            hi += (acc >> 8) & 0x00ff00ffu;
            acc = 0;
            pending = 0;
        }
    }
    lo += acc & 0x00ff00ffu;
    hi += (acc >> 8) & 0x00ff00ffu;
}

// One wave per candidate: refine at level l (line2Dup.cpp:1233-1287), apply the
// per-level filter (:1290-1292); at level 0 emit the final Match record.
__global__ __launch_bounds__(256) void k_similarity_local(
    const uint8_t* __restrict__ lm, int rows, int cols, int T, int W, int L, int l,
    const DevTL* __restrict__ tls, const uint32_t* __restrict__ fxy, const int32_t* __restrict__ foff,
    const int32_t* __restrict__ raw_keep, const int32_t* __restrict__ class_idx,
    const int32_t* __restrict__ template_id, Cand* __restrict__ cands, int32_t* __restrict__ counters,
    int cand_cap, int is_last, sbm_match_rec* __restrict__ out, int32_t* __restrict__ out_count,
    int out_cap)
{
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int nwaves = gridDim.x * 4;
    int n = counters[0];
    n = n < cand_cap ? n : cand_cap;
    const int border = 8 * T, offset = T / 2 + (T % 2 - 1);
    for (int ci = wave; ci < n; ci += nwaves) {
        Cand c = cands[ci];
        if (c.raw < 0) continue; // dropped at a coarser level
        const DevTL tl = tls[(size_t)c.t * L + l];
        int x = c.x * 2 + 1, y = c.y * 2 + 1;
        const int max_x = cols - tl.width - border, max_y = rows - tl.height - border;
        x = x < border ? border : x;
        y = y < border ? border : y;
        x = x > max_x ? max_x : x;
        y = y > max_y ? max_y : y;
        const int ox = (x / T - 8) * T, oy = (y / T - 8) * T;
        uint32_t lo, hi;
        local_accumulate(lm, tl, fxy, foff, rows, cols, W, T, ox, oy, lane, lo, hi);
        // first maximum in row-major order, strict '>' from 0 (:1265-1282):
        // maximise (raw, -position)
        uint32_t best = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            uint32_t raw = (k & 1) ? ((hi >> (8 * (k - 1))) & 0xffff) : ((lo >> (8 * k)) & 0xffff);
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
            atomicAdd((unsigned long long*)(counters + 2), (unsigned long long)tl.nf * 256ull); // refinement bytes (stats)
            const int raw = (int)(best >> 8);
            int br = -1, bc = -1;
            if (raw > 0) {
                int p = 255 - (int)(best & 255u);
                br = p >> 4;
                bc = p & 15;
            }
            const int nx = (x / T - 8 + bc) * T + offset;
            const int ny = (y / T - 8 + br) * T + offset;
            const bool keep = raw >= raw_keep[(size_t)c.t * L + l];
            if (is_last) {
                if (keep) {
                    int idx = atomicAdd(out_count, 1);
                    if (idx < out_cap) {
                        sbm_match_rec m;
                        m.x = nx;
                        m.y = ny;
                        m.similarity = __fdiv_rn(__fmul_rn((float)raw, 100.f), (float)(4 * tl.nf));
                        m.raw = raw;
                        m.class_idx = class_idx[c.t];
                        m.template_id = template_id[c.t];
                        out[idx] = m;
                    }
                }
            } else {
                c.x = nx;
                c.y = ny;
                c.raw = keep ? raw : -1;
                cands[ci] = c;
            }
        }
    }
}

// stage entry point: one 16x16 patch, one wave
__global__ __launch_bounds__(64) void k_similarity_local_patch(const uint8_t* __restrict__ lm, int rows,
                                                               int cols, int T, int W, DevTL tl,
                                                               const uint32_t* __restrict__ fxy,
                                                               const int32_t* __restrict__ foff, int cx,
                                                               int cy, uint16_t* __restrict__ dst)
{
    const int lane = threadIdx.x;
    const int ox = (cx / T - 8) * T, oy = (cy / T - 8) * T;
    uint32_t lo, hi;
    local_accumulate(lm, tl, fxy, foff, rows, cols, W, T, ox, oy, lane, lo, hi);
#pragma unroll
    for (int k = 0; k < 4; ++k)
        dst[lane * 4 + k] = (uint16_t)((k & 1) ? ((hi >> (8 * (k - 1))) & 0xffff) : ((lo >> (8 * k)) & 0xffff));
}

// single-level pyramids: coarse candidates are the final matches
__global__ __launch_bounds__(256) void k_emit_coarse(const Cand* __restrict__ cands,
                                                     const int32_t* __restrict__ counters, int cand_cap,
                                                     const DevTL* __restrict__ tls, int L, int lc,
                                                     const int32_t* __restrict__ class_idx,
                                                     const int32_t* __restrict__ template_id,
                                                     sbm_match_rec* __restrict__ out,
                                                     int32_t* __restrict__ out_count, int out_cap)
{
    int n = counters[0];
    n = n < cand_cap ? n : cand_cap;
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
        }
    }
}

// status word: counters[1] = 1 when the candidate list overflowed
__global__ void k_finish(int32_t* __restrict__ counters, int cand_cap, int32_t* __restrict__ out_count)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) out_count[1] = counters[0] > cand_cap ? 1 : 0;
}

} // namespace sbm
