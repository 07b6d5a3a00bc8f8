// sbm_resize_table.h — coefficient tables of cv::resize(..., INTER_LINEAR) for 8-bit images, shared by the HIP
// stage entry point (sbm_resize_linear) and the host-side cv::resize of the bundled cv:: subset.
// shapeInfo_producer::transform (line2Dup.h:379-405) calls cv::resize(src, dst, cv::Size(), scale, scale); the
// arithmetic is OpenCV's (imgproc/resize.cpp: resizeGeneric_, HResizeLinear / VResizeLinear for uchar): per
// destination coordinate d:  f = (float)((d + 0.5) / scale - 0.5), s = floor(f), f -= s, clamped at the borders;
// 11-bit coefficients cvRound((1 - f) * 2048), cvRound(f * 2048).
#pragma once
#include <cmath>
#include <cstdint>
#include <vector>

namespace sbm {

inline int resize_round(double v) { return (int)std::lrint(v); } // cvRound: half to even

inline void resize_linear_dims(int rows, int cols, double fx, double fy, int* drows, int* dcols)
{
    *dcols = resize_round(cols * fx);
    *drows = resize_round(rows * fy);
}

// idx[d]: first source sample; coef[2d], coef[2d+1]: its weight and the next sample's (x 2048)
inline void resize_linear_table(int dn, int sn, double inv_scale, std::vector<int32_t>& idx, std::vector<int16_t>& coef)
{
    idx.resize(dn);
    coef.resize(2 * (size_t)dn);
    for (int d = 0; d < dn; ++d) {
        float f = (float)((d + 0.5) * inv_scale - 0.5);
        int s = (int)std::floor(f);
        f -= (float)s;
        if (s < 0) { f = 0.f; s = 0; }
        if (s >= sn - 1) { f = 0.f; s = sn - 1; }
        idx[d] = s;
        coef[2 * d] = (int16_t)std::lrint((1.f - f) * 2048.f);
        coef[2 * d + 1] = (int16_t)std::lrint(f * 2048.f);
    }
}

// one output sample from the four neighbours (OpenCV's two fixed-point passes)
#if defined(__HIPCC__)
__host__ __device__
#endif
inline uint8_t resize_linear_sample(int p00, int p01, int p10, int p11, int ax0, int ax1, int ay0, int ay1)
{
    const int h0 = p00 * ax0 + p01 * ax1, h1 = p10 * ax0 + p11 * ax1;
    const int v = (((ay0 * (h0 >> 4)) >> 16) + ((ay1 * (h1 >> 4)) >> 16) + 2) >> 2;
    return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

} // namespace sbm
