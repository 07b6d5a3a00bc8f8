// sbm_train_kernels.h — training side of the engine (gfx950): the per-pixel scan of
// ColorGradientPyramid::extractTemplate (line2Dup.cpp:452-539) as a data-parallel kernel.
//
// The reference walks the image in row-major order with a `magnitude_valid` map: a pixel that is still valid and has no
// 5x5 neighbour of strictly larger squared magnitude (:485) is a local maximum, invalidates its 24 neighbours (:494-500)
// and becomes a feature candidate if its score exceeds strong_threshold^2 and it has a quantised orientation (:504).
// That scan is order-dependent only through ties: a maximum q can invalidate a pixel p that would itself have been a
// maximum only if score(p) == score(q) (each is >= the other).  So:
//   1. (this kernel) every pixel of the scanned region [2, rows-2) x [2, cols-2) that passes the eroded mask, scores
//      above strong_threshold^2 and has no strictly larger 5x5 neighbour is emitted -- no order involved;
//   2. (host, sbm_extract_local_maxima) the emitted pixels are sorted row-major and a pixel is dropped when an earlier
//      KEPT pixel lies within its 5x5 window -- the reference's invalidation among equal-score neighbours, exactly.
// Maxima at or below the threshold never influence the result: to invalidate a candidate they would need its score.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sbm {

// out_xy[k] = x | y << 16; *count may exceed cap (the caller retries with a larger buffer)
__global__ __launch_bounds__(256) void k_local_maxima5(const float* __restrict__ mag, const uint8_t* __restrict__ mask, int rows, int cols,
                                                       float thr_sq, int32_t* __restrict__ out_xy, int32_t* __restrict__ count, int cap)
{
    const int iw = cols - 4, ih = rows - 4;
    const int n = iw * ih;
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < n; idx += gridDim.x * 256) {
        const int r = 2 + idx / iw, c = 2 + idx % iw;
        const float s = mag[(size_t)r * cols + c];
        if (!(s > thr_sq)) continue;
        if (mask) { // cv::erode(mask, 3x3, BORDER_REPLICATE): every pixel of the 3x3 window (clamped) must be set (:459-463)
            bool keep = true;
            for (int dr = -1; dr <= 1 && keep; ++dr) {
                const int rr = min(max(r + dr, 0), rows - 1);
                for (int dc = -1; dc <= 1; ++dc) keep = keep && mask[(size_t)rr * cols + min(max(c + dc, 0), cols - 1)] != 0;
            }
            if (!keep) continue;
        }
        bool is_max = true;
        for (int dr = -2; dr <= 2 && is_max; ++dr) {
            const float* row = mag + (size_t)(r + dr) * cols + c;
            for (int dc = -2; dc <= 2; ++dc) is_max = is_max && !(s < row[dc]);
        }
        if (!is_max) continue;
        const int k = atomicAdd(count, 1);
        if (k < cap) out_xy[k] = c | (r << 16);
    }
}

} // namespace sbm
