/*
 * sbm_oracle.c — CPU ORACLE.  TEST INFRASTRUCTURE ONLY (see sbm_oracle.h).
 *
 * Plain-C restatement of ddcr/shape_based_matching's LINE-2D match() path.
 * Written from the behaviour of line2Dup.cpp (cited per function) and from the
 * documented semantics of the OpenCV-4 primitives it calls; it shares no code
 * with either.  Build: oracle/Makefile (-O3 -mavx2 -fopenmp -ffp-contract=off).
 */
#include "sbm_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
static inline int reflect101(int p, int len)
{
    if (len == 1) return 0;
    while (p < 0 || p >= len) {
        if (p < 0) p = -p;
        else p = 2 * len - 2 - p;
    }
    return p;
}

/* ------------------------------------------------------------------------- */
/* GaussianBlur(src, 7x7, sigma=0, BORDER_REPLICATE) on 8-bit data.            */
/* line2Dup.cpp:320.  OpenCV 4 8-bit path: fixed kernel {8,28,56,72,56,28,8}   */
/* /256 per axis, exact 8.8 horizontal pass, vertical pass rounded once:      */
/* (sum + 2^15) >> 16.                                                        */
/* ------------------------------------------------------------------------- */
static const int GK7[7] = {8, 28, 56, 72, 56, 28, 8};

/* Threads of the pyramid build (sbo_set_build_threads; 1 = the plain serial restatement the parity tests use).  The
 * reference's build is a chain of OpenCV calls (line2Dup.cpp:1084-1120) that OpenCV itself runs over row bands with
 * parallel_for_; the bench's cpu_baseline leg sets this to the host's physical cores so that the CPU figure has the
 * reference's shape (row-band parallel build + the OpenMP template loop of :1166-1170).  Every loop below that carries
 * the pragma writes disjoint rows: results do not depend on the thread count. */
static int g_build_threads = 1;
void sbo_set_build_threads(int n) { g_build_threads = n < 1 ? 1 : n; }
int sbo_get_build_threads(void) { return g_build_threads; }
#define SBO_ROWS_PARALLEL _Pragma("omp parallel for schedule(static) num_threads(g_build_threads) if (g_build_threads > 1)")

void sbo_gaussian7(const uint8_t* src, int rows, int cols, int ch, int stride, uint8_t* dst)
{
    const int n = cols * ch;
    uint16_t* tmp = (uint16_t*)malloc((size_t)rows * n * sizeof(uint16_t));
    SBO_ROWS_PARALLEL
    for (int r = 0; r < rows; ++r) {
        const uint8_t* s = src + (size_t)r * stride;
        uint16_t* t = tmp + (size_t)r * n;
        for (int c = 0; c < cols; ++c)
            for (int k = 0; k < ch; ++k) {
                int acc = 0;
                for (int i = 0; i < 7; ++i) acc += GK7[i] * s[clampi(c + i - 3, 0, cols - 1) * ch + k];
                t[c * ch + k] = (uint16_t)acc;
            }
    }
    SBO_ROWS_PARALLEL
    for (int r = 0; r < rows; ++r) {
        uint8_t* d = dst + (size_t)r * n;
        for (int x = 0; x < n; ++x) {
            uint32_t acc = 0;
            for (int j = 0; j < 7; ++j)
                acc += (uint32_t)GK7[j] * tmp[(size_t)clampi(r + j - 3, 0, rows - 1) * n + x];
            d[x] = (uint8_t)((acc + 32768u) >> 16);
        }
    }
    free(tmp);
}

/* Sobel(ksize 3, BORDER_REPLICATE), dx = d/dx, dy = d/dy.  line2Dup.cpp:324-325, 343-344. */
void sbo_sobel3(const uint8_t* sm, int rows, int cols, int ch, int16_t* dx, int16_t* dy)
{
    const int n = cols * ch;
    SBO_ROWS_PARALLEL
    for (int r = 0; r < rows; ++r) {
        const uint8_t* r0 = sm + (size_t)clampi(r - 1, 0, rows - 1) * n;
        const uint8_t* r1 = sm + (size_t)r * n;
        const uint8_t* r2 = sm + (size_t)clampi(r + 1, 0, rows - 1) * n;
        for (int c = 0; c < cols; ++c) {
            const int cl = clampi(c - 1, 0, cols - 1) * ch, cc = c * ch, cr = clampi(c + 1, 0, cols - 1) * ch;
            for (int k = 0; k < ch; ++k) {
                int gx = (r0[cr + k] - r0[cl + k]) + 2 * (r1[cr + k] - r1[cl + k]) + (r2[cr + k] - r2[cl + k]);
                int gy = (r2[cl + k] - r0[cl + k]) + 2 * (r2[cc + k] - r0[cc + k]) + (r2[cr + k] - r0[cr + k]);
                dx[(size_t)r * n + cc + k] = (int16_t)gx;
                dy[(size_t)r * n + cc + k] = (int16_t)gy;
            }
        }
    }
}

/* cv::phase(x, y, angle, angleInDegrees=true) -> fastAtan2 polynomial
 * (OpenCV core mathfuncs_core: atan_f32).  line2Dup.cpp:327, :398. */
float sbo_fast_atan2_deg(float y, float x)
{
    const float p1 = 0.9997878412794807f * (float)(180 / 3.14159265358979323846);
    const float p3 = -0.3258083974640975f * (float)(180 / 3.14159265358979323846);
    const float p5 = 0.1555786518463281f * (float)(180 / 3.14159265358979323846);
    const float p7 = -0.04432655554792128f * (float)(180 / 3.14159265358979323846);
    float ax = fabsf(x), ay = fabsf(y);
    float a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + (float)DBL_EPSILON);
        c2 = c * c;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
        c = ax / (ay + (float)DBL_EPSILON);
        c2 = c * c;
        a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

/* The 16-bin orientation index hysteresisGradient derives from the phase image (:225):
 * saturate_cast<uchar>(cvRound(fastAtan2(dy, dx) * (16/360))), vectorised over integer gradients. */
void sbo_orientation_bins(const int16_t* dx, const int16_t* dy, int64_t n, uint8_t* q16)
{
    const float scale = (float)(16.0 / 360.0);
    for (int64_t i = 0; i < n; ++i) {
        long v = lrintf(sbo_fast_atan2_deg((float)dy[i], (float)dx[i]) * scale);
        q16[i] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
    }
}

/* cv::pyrDown(src, dst, Size(cols/2, rows/2)), BORDER_DEFAULT (REFLECT_101):
 * [1 4 6 4 1]^2 / 256, (sum + 128) >> 8.  line2Dup.cpp:431-433. */
void sbo_pyrdown(const uint8_t* src, int rows, int cols, int ch, int stride, uint8_t* dst)
{
    static const int K[5] = {1, 4, 6, 4, 1};
    const int dr = rows / 2, dc = cols / 2;
    SBO_ROWS_PARALLEL
    for (int y = 0; y < dr; ++y)
        for (int x = 0; x < dc; ++x)
            for (int k = 0; k < ch; ++k) {
                int acc = 0;
                for (int j = 0; j < 5; ++j) {
                    const uint8_t* s = src + (size_t)reflect101(2 * y + j - 2, rows) * stride;
                    int h = 0;
                    for (int i = 0; i < 5; ++i) h += K[i] * s[reflect101(2 * x + i - 2, cols) * ch + k];
                    acc += K[j] * h;
                }
                dst[((size_t)y * dc + x) * ch + k] = (uint8_t)((acc + 128) >> 8);
            }
}

/* resize(mask, next, size, 0, 0, INTER_NEAREST).  line2Dup.cpp:439. */
/* cv::resize(src, dst, Size(), fx, fy, INTER_LINEAR) on 8-bit images, as shapeInfo_producer::transform calls it
 * (line2Dup.h:383-397).  The arithmetic lives in OpenCV (pinned only as "4", CMakeLists.txt:36); restated from its
 * published generic path (imgproc/resize.cpp, resizeGeneric_ with HResizeLinear<uchar,int,short> and
 * VResizeLinear<uchar,int,short>): dsize = (cvRound(cols*fx), cvRound(rows*fy)); per destination coordinate
 * f = (float)((d + 0.5) / fx - 0.5), s = floor(f), f -= s, clamped to the image; 11-bit coefficients
 * cvRound((1-f)*2048), cvRound(f*2048); horizontal pass in int, vertical pass
 * (((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2.  Unverifiable offline (no OpenCV in the image). */
static int sbo_cv_round(double v) { return (int)lrint(v); }
void sbo_resize_linear_dims(int rows, int cols, double fx, double fy, int* drows, int* dcols)
{
    *dcols = sbo_cv_round(cols * fx);
    *drows = sbo_cv_round(rows * fy);
}
static void resize_linear_table(int dn, int sn, double scale, int* idx, short* coef)
{
    for (int d = 0; d < dn; ++d) {
        float f = (float)((d + 0.5) * scale - 0.5);
        int s = (int)floorf(f);
        f -= (float)s;
        if (s < 0) { f = 0.f; s = 0; }
        if (s >= sn - 1) { f = 0.f; s = sn - 1; }
        idx[d] = s;
        coef[2 * d] = (short)lrintf((1.f - f) * 2048.f);
        coef[2 * d + 1] = (short)lrintf(f * 2048.f);
    }
}
void sbo_resize_linear_u8(const uint8_t* src, int rows, int cols, int ch, int stride, double fx, double fy, uint8_t* dst)
{
    int drows, dcols;
    sbo_resize_linear_dims(rows, cols, fx, fy, &drows, &dcols);
    if (drows <= 0 || dcols <= 0) return;
    int* xi = (int*)malloc(sizeof(int) * (size_t)dcols);
    int* yi = (int*)malloc(sizeof(int) * (size_t)drows);
    short* xa = (short*)malloc(sizeof(short) * 2 * (size_t)dcols);
    short* ya = (short*)malloc(sizeof(short) * 2 * (size_t)drows);
    resize_linear_table(dcols, cols, 1.0 / fx, xi, xa);
    resize_linear_table(drows, rows, 1.0 / fy, yi, ya);
    for (int y = 0; y < drows; ++y) {
        const uint8_t* r0 = src + (size_t)yi[y] * stride;
        const uint8_t* r1 = src + (size_t)(yi[y] + 1 < rows ? yi[y] + 1 : rows - 1) * stride;
        for (int x = 0; x < dcols; ++x) {
            const int x0 = xi[x], x1 = x0 + 1 < cols ? x0 + 1 : cols - 1;
            for (int k = 0; k < ch; ++k) {
                const int h0 = r0[x0 * ch + k] * xa[2 * x] + r0[x1 * ch + k] * xa[2 * x + 1];
                const int h1 = r1[x0 * ch + k] * xa[2 * x] + r1[x1 * ch + k] * xa[2 * x + 1];
                const int v = (((ya[2 * y] * (h0 >> 4)) >> 16) + ((ya[2 * y + 1] * (h1 >> 4)) >> 16) + 2) >> 2;
                dst[((size_t)y * dcols + x) * ch + k] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
            }
        }
    }
    free(xi);
    free(yi);
    free(xa);
    free(ya);
}

void sbo_resize_nearest_u8(const uint8_t* src, int rows, int cols, uint8_t* dst, int drows, int dcols)
{
    const double fx = (double)cols / dcols, fy = (double)rows / drows;
    for (int y = 0; y < drows; ++y) {
        int sy = (int)floor(y * fy);
        if (sy > rows - 1) sy = rows - 1;
        for (int x = 0; x < dcols; ++x) {
            int sx = (int)floor(x * fx);
            if (sx > cols - 1) sx = cols - 1;
            dst[(size_t)y * dcols + x] = src[(size_t)sy * cols + sx];
        }
    }
}

/* hysteresisGradient.  line2Dup.cpp:218-311 (PATCH_2843 == 0). */
static void hysteresis(const float* magnitude, const float* angle_deg, int rows, int cols,
                       float threshold, uint8_t* out)
{
    uint8_t* q = (uint8_t*)malloc((size_t)rows * cols);
    const float scale = (float)(16.0 / 360.0);
    SBO_ROWS_PARALLEL
    for (size_t i = 0; i < (size_t)rows * cols; ++i) {
        /* convertTo(CV_8U, 16/360): saturate_cast<uchar>(cvRound(v * alpha)), round-half-even */
        long v = lrintf(angle_deg[i] * scale);
        q[i] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
    }
    memset(q, 0, cols);
    memset(q + (size_t)(rows - 1) * cols, 0, cols);
    for (int r = 0; r < rows; ++r) {
        q[(size_t)r * cols] = 0;
        q[(size_t)r * cols + cols - 1] = 0;
    }
    for (int r = 1; r < rows - 1; ++r)
        for (int c = 1; c < cols - 1; ++c) q[(size_t)r * cols + c] &= 7;

    memset(out, 0, (size_t)rows * cols);
    SBO_ROWS_PARALLEL
    for (int r = 1; r < rows - 1; ++r)
        for (int c = 1; c < cols - 1; ++c) {
            if (!(magnitude[(size_t)r * cols + c] > threshold)) continue;
            int hist[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (int dr = -1; dr <= 1; ++dr)
                for (int dc = -1; dc <= 1; ++dc) hist[q[(size_t)(r + dr) * cols + c + dc]]++;
            int max_votes = 0, index = -1;
            for (int i = 0; i < 8; ++i)
                if (max_votes < hist[i]) {
                    index = i;
                    max_votes = hist[i];
                }
            if (max_votes >= 5) out[(size_t)r * cols + c] = (uint8_t)(1 << index);
        }
    free(q);
}

/* quantizedOrientations.  line2Dup.cpp:313-404. */
void sbo_quantized_orientations(const uint8_t* src, int rows, int cols, int ch, int stride,
                                float weak_threshold, float* magnitude, uint8_t* angle,
                                float* angle_ori)
{
    const size_t npx = (size_t)rows * cols;
    uint8_t* sm = (uint8_t*)malloc(npx * ch);
    int16_t* dx = (int16_t*)malloc(npx * ch * sizeof(int16_t));
    int16_t* dy = (int16_t*)malloc(npx * ch * sizeof(int16_t));
    float* mag = magnitude ? magnitude : (float*)malloc(npx * sizeof(float));
    float* ang = angle_ori ? angle_ori : (float*)malloc(npx * sizeof(float));
    sbo_gaussian7(src, rows, cols, ch, stride, sm);
    sbo_sobel3(sm, rows, cols, ch, dx, dy);
    SBO_ROWS_PARALLEL
    for (size_t i = 0; i < npx; ++i) {
        float fx, fy, m;
        if (ch == 1) {
            fx = (float)dx[i];
            fy = (float)dy[i];
            m = fx * fx + fy * fy;
        } else {
            /* channel of maximum magnitude, ties to the lower index (:370-387) */
            int best = 0, bm = -1;
            int m3[3];
            for (int k = 0; k < 3; ++k) m3[k] = dx[i * ch + k] * dx[i * ch + k] + dy[i * ch + k] * dy[i * ch + k];
            if (m3[0] >= m3[1] && m3[0] >= m3[2]) best = 0;
            else if (m3[1] >= m3[0] && m3[1] >= m3[2]) best = 1;
            else best = 2;
            bm = m3[best];
            fx = (float)dx[i * ch + best];
            fy = (float)dy[i * ch + best];
            m = (float)bm;
        }
        mag[i] = m;
        ang[i] = sbo_fast_atan2_deg(fy, fx);
    }
    hysteresis(mag, ang, rows, cols, weak_threshold * weak_threshold, angle);
    free(sm);
    free(dx);
    free(dy);
    if (!magnitude) free(mag);
    if (!angle_ori) free(ang);
}

/* ------------------------------------------------------------------------- */
/* spread / computeResponseMaps / linearize                                   */
/* ------------------------------------------------------------------------- */

/* spread.  line2Dup.cpp:616-630 (+ orUnaligned8u :583-614). */
void sbo_spread(const uint8_t* src, int rows, int cols, int T, uint8_t* dst)
{
    memset(dst, 0, (size_t)rows * cols);
    SBO_ROWS_PARALLEL
    for (int r = 0; r < rows; ++r) { /* a destination row is the OR of T source rows: rows are independent */
        uint8_t* d = dst + (size_t)r * cols;
        for (int dr = 0; dr < T && r + dr < rows; ++dr)
            for (int dc = 0; dc < T; ++dc) {
                const uint8_t* s = src + (size_t)(r + dr) * cols + dc;
                for (int c = 0; c < cols - dc; ++c) d[c] |= s[c];
            }
    }
}

/* computeResponseMaps.  line2Dup.cpp:637-747.  SIMILARITY_LUT (:632-635) in
 * closed form: 4 if bit o is set, else 3 if a circular neighbour bit is set,
 * else 0 (checked entry by entry against the table in tests/test_oracle_pins.py). */
void sbo_response_maps(const uint8_t* spread, int64_t n, uint8_t* maps)
{
    SBO_ROWS_PARALLEL
    for (int o = 0; o < 8; ++o) {
        const unsigned self = 1u << o;
        const unsigned nb = (1u << ((o + 1) & 7)) | (1u << ((o + 7) & 7));
        uint8_t* m = maps + (size_t)o * n;
        for (int64_t i = 0; i < n; ++i) {
            unsigned v = spread[i];
            m[i] = (uint8_t)((v & self) ? 4 : ((v & nb) ? 3 : 0));
        }
    }
}

/* linearize.  line2Dup.cpp:749-777. */
void sbo_linearize(const uint8_t* map, int rows, int cols, int T, uint8_t* lm)
{
    const int W = cols / T, H = rows / T;
    size_t k = 0;
    for (int rs = 0; rs < T; ++rs)
        for (int cs = 0; cs < T; ++cs)
            for (int r = rs; r < rows; r += T)
                for (int c = cs; c < cols; c += T) lm[k++] = map[(size_t)r * cols + c];
    (void)W;
    (void)H;
}

/* ------------------------------------------------------------------------- */
/* pyramid of flat linear memories                                            */
/* ------------------------------------------------------------------------- */
struct sbo_pyramid {
    int n_levels;
    int rows[SBM_MAX_LEVELS], cols[SBM_MAX_LEVELS], T[SBM_MAX_LEVELS];
    int64_t lm_stride[SBM_MAX_LEVELS];
    uint8_t* lm[SBM_MAX_LEVELS]; /* [8][lm_stride], zero tail */
    uint8_t* quant[SBM_MAX_LEVELS];
};

static int64_t lm_stride_for(int rows, int cols, int T)
{
    int64_t W = cols / T, H = rows / T;
    int64_t s = (int64_t)T * T * W * H + W * H + 16 * W + 80;
    return (s + 63) / 64 * 64;
}

static int build_level(sbo_pyramid* p, int l, const uint8_t* q)
{
    const int rows = p->rows[l], cols = p->cols[l], T = p->T[l];
    if (rows % T || cols % T || ((int64_t)rows * cols) % 16) return -1; /* :639, :751-752 */
    const size_t n = (size_t)rows * cols;
    p->lm_stride[l] = lm_stride_for(rows, cols, T);
    p->lm[l] = (uint8_t*)calloc((size_t)8 * p->lm_stride[l], 1);
    p->quant[l] = (uint8_t*)malloc(n);
    memcpy(p->quant[l], q, n);
    uint8_t* sp = (uint8_t*)malloc(n);
    uint8_t* maps = (uint8_t*)malloc(8 * n);
    sbo_spread(q, rows, cols, T, sp);
    sbo_response_maps(sp, (int64_t)n, maps);
    SBO_ROWS_PARALLEL
    for (int o = 0; o < 8; ++o) sbo_linearize(maps + o * n, rows, cols, T, p->lm[l] + (size_t)o * p->lm_stride[l]);
    free(sp);
    free(maps);
    return 0;
}

sbo_pyramid* sbo_pyramid_from_quantized(const uint8_t* const* q, const int* rows, const int* cols,
                                        int n_levels, const int* T)
{
    sbo_pyramid* p = (sbo_pyramid*)calloc(1, sizeof(*p));
    p->n_levels = n_levels;
    for (int l = 0; l < n_levels; ++l) {
        p->rows[l] = rows[l];
        p->cols[l] = cols[l];
        p->T[l] = T[l];
        if (build_level(p, l, q[l])) {
            sbo_pyramid_free(p);
            return NULL;
        }
    }
    return p;
}

/* Detector::match pyramid construction.  line2Dup.cpp:1084-1120. */
sbo_pyramid* sbo_pyramid_build(const uint8_t* img, int rows, int cols, int stride, int ch,
                               const uint8_t* mask, int n_levels, const int* T, float weak)
{
    sbo_pyramid* p = (sbo_pyramid*)calloc(1, sizeof(*p));
    p->n_levels = n_levels;
    uint8_t* cur = (uint8_t*)malloc((size_t)rows * cols * ch);
    for (int r = 0; r < rows; ++r) memcpy(cur + (size_t)r * cols * ch, img + (size_t)r * stride, (size_t)cols * ch);
    uint8_t* cmask = NULL;
    if (mask) {
        cmask = (uint8_t*)malloc((size_t)rows * cols);
        memcpy(cmask, mask, (size_t)rows * cols);
    }
    int cr = rows, cc = cols, ok = 1;
    for (int l = 0; l < n_levels && ok; ++l) {
        if (l > 0) { /* ColorGradientPyramid::pyrDown :424-444 */
            int nr = cr / 2, nc = cc / 2;
            uint8_t* nxt = (uint8_t*)malloc((size_t)nr * nc * ch);
            sbo_pyrdown(cur, cr, cc, ch, cc * ch, nxt);
            free(cur);
            cur = nxt;
            if (cmask) {
                uint8_t* nm = (uint8_t*)malloc((size_t)nr * nc);
                sbo_resize_nearest_u8(cmask, cr, cc, nm, nr, nc);
                free(cmask);
                cmask = nm;
            }
            cr = nr;
            cc = nc;
        }
        uint8_t* q = (uint8_t*)malloc((size_t)cr * cc);
        sbo_quantized_orientations(cur, cr, cc, ch, cc * ch, weak, NULL, q, NULL);
        if (cmask) /* quantize(): angle.copyTo(dst, mask) :446-450 */
            for (size_t i = 0; i < (size_t)cr * cc; ++i)
                if (!cmask[i]) q[i] = 0;
        p->rows[l] = cr;
        p->cols[l] = cc;
        p->T[l] = T[l];
        if (build_level(p, l, q)) ok = 0;
        free(q);
    }
    free(cur);
    free(cmask);
    if (!ok) {
        sbo_pyramid_free(p);
        return NULL;
    }
    return p;
}

void sbo_pyramid_free(sbo_pyramid* p)
{
    if (!p) return;
    for (int l = 0; l < SBM_MAX_LEVELS; ++l) {
        free(p->lm[l]);
        free(p->quant[l]);
    }
    free(p);
}
int sbo_pyramid_rows(const sbo_pyramid* p, int l) { return p->rows[l]; }
int sbo_pyramid_cols(const sbo_pyramid* p, int l) { return p->cols[l]; }
int64_t sbo_pyramid_lm_stride(const sbo_pyramid* p, int l) { return p->lm_stride[l]; }
const uint8_t* sbo_pyramid_lm(const sbo_pyramid* p, int l) { return p->lm[l]; }
const uint8_t* sbo_pyramid_quantized(const sbo_pyramid* p, int l) { return p->quant[l]; }

/* ------------------------------------------------------------------------- */
/* similarity / similarityLocal                                               */
/* ------------------------------------------------------------------------- */

/* accessLinearMemory as an offset into the flat [8][lm_stride] block.  :782-805 */
static inline int64_t lm_offset(int64_t lm_stride, int label, int x, int y, int T, int W, int H)
{
    return (int64_t)label * lm_stride + (int64_t)((y % T) * T + (x % T)) * W * H + (int64_t)(y / T) * W + x / T;
}

static inline int template_positions(const sbm_template_level* tl, int W, int H, int T)
{
    int wf = (tl->width - 1) / T + 1, hf = (tl->height - 1) / T + 1; /* :818-819 */
    return (H - hf) * W + (W - wf) + 1;                              /* :822-825 */
}

/* similarity / similarity_64.  line2Dup.cpp:807-858, 924-984. */
void sbo_similarity(const uint8_t* lm, int64_t lm_stride, int rows, int cols, int T,
                    const sbm_template_level* tl, const sbm_feature* feats, uint16_t* dst)
{
    const int W = cols / T, H = rows / T;
    const int npos = template_positions(tl, W, H, T);
    memset(dst, 0, (size_t)W * H * sizeof(uint16_t));
    for (int i = 0; i < tl->n_features; ++i) {
        const sbm_feature f = feats[tl->feature_offset + i];
        if (f.x < 0 || f.x >= cols || f.y < 0 || f.y >= rows) continue;
        const uint8_t* s = lm + lm_offset(lm_stride, f.label, f.x, f.y, T, W, H);
        for (int j = 0; j < npos; ++j) dst[j] += s[j];
    }
}

/* similarityLocal / similarityLocal_64.  line2Dup.cpp:860-922, 986-1048. */
void sbo_similarity_local(const uint8_t* lm, int64_t lm_stride, int rows, int cols, int T,
                          const sbm_template_level* tl, const sbm_feature* feats, int cx, int cy,
                          uint16_t* dst)
{
    const int W = cols / T, H = rows / T;
    const int ox = (cx / T - 8) * T, oy = (cy / T - 8) * T; /* :868-869 */
    memset(dst, 0, 256 * sizeof(uint16_t));
    for (int i = 0; i < tl->n_features; ++i) {
        sbm_feature f = feats[tl->feature_offset + i];
        f.x += ox;
        f.y += oy;
        if (f.x < 0 || f.y < 0 || f.x >= cols || f.y >= rows) continue;
        const uint8_t* s = lm + lm_offset(lm_stride, f.label, f.x, f.y, T, W, H);
        for (int r = 0; r < 16; ++r)
            for (int c = 0; c < 16; ++c) dst[r * 16 + c] += s[(size_t)r * W + c];
    }
}

/* ------------------------------------------------------------------------- */
/* matchClass for one template.  line2Dup.cpp:1170-1296.                      */
/* ------------------------------------------------------------------------- */
typedef struct {
    sbm_match_rec* v;
    int64_t n, cap;
} recvec;
static void rv_push(recvec* rv, sbm_match_rec m)
{
    if (rv->n == rv->cap) {
        rv->cap = rv->cap ? rv->cap * 2 : 64;
        rv->v = (sbm_match_rec*)realloc(rv->v, (size_t)rv->cap * sizeof(sbm_match_rec));
    }
    rv->v[rv->n++] = m;
}

static void match_one(const sbo_pyramid* p, const sbm_template_level* tp, const sbm_feature* feats,
                      int class_idx, int template_id, float threshold, recvec* out, uint16_t* sim)
{
    const int L = p->n_levels;
    recvec cand = {0, 0, 0};
    {
        const int l = L - 1, T = p->T[l], W = p->cols[l] / T, H = p->rows[l] / T;
        const sbm_template_level* tl = &tp[l];
        const int nf = tl->n_features;
        if (nf >= 8192) return; /* CV_Error :1195 — rejected earlier by callers */
        sbo_similarity(p->lm[l], p->lm_stride[l], p->rows[l], p->cols[l], T, tl, feats, sim);
        const int offset = T / 2 + (T % 2 - 1);
        for (int r = 0; r < H; ++r)
            for (int c = 0; c < W; ++c) {
                int raw = sim[(size_t)r * W + c];
                float score = (raw * 100.f) / (4 * nf); /* :1206 */
                if (score > threshold) {
                    sbm_match_rec m = {c * T + offset, r * T + offset, score, raw, class_idx, template_id};
                    rv_push(&cand, m);
                }
            }
    }
    uint16_t patch[256];
    for (int l = L - 2; l >= 0; --l) {
        const int T = p->T[l], rows = p->rows[l], cols = p->cols[l];
        const sbm_template_level* tl = &tp[l];
        const int nf = tl->n_features;
        const int border = 8 * T, offset = T / 2 + (T % 2 - 1);
        const int max_x = cols - tl->width - border, max_y = rows - tl->height - border;
        for (int64_t m = 0; m < cand.n; ++m) {
            sbm_match_rec* mt = &cand.v[m];
            int x = mt->x * 2 + 1, y = mt->y * 2 + 1;
            if (x < border) x = border;
            if (y < border) y = border;
            if (x > max_x) x = max_x;
            if (y > max_y) y = max_y;
            sbo_similarity_local(p->lm[l], p->lm_stride[l], rows, cols, T, tl, feats, x, y, patch);
            float best = 0;
            int br = -1, bc = -1, braw = 0;
            for (int r = 0; r < 16; ++r)
                for (int c = 0; c < 16; ++c) {
                    int raw = patch[r * 16 + c];
                    float score = (raw * 100.f) / (4 * nf);
                    if (score > best) {
                        best = score;
                        br = r;
                        bc = c;
                        braw = raw;
                    }
                }
            mt->similarity = best;
            mt->raw = braw;
            mt->x = (x / T - 8 + bc) * T + offset;
            mt->y = (y / T - 8 + br) * T + offset;
        }
        int64_t k = 0;
        for (int64_t m = 0; m < cand.n; ++m)
            if (!(cand.v[m].similarity < threshold)) cand.v[k++] = cand.v[m]; /* :1290-1292 */
        cand.n = k;
    }
    for (int64_t m = 0; m < cand.n; ++m) rv_push(out, cand.v[m]);
    free(cand.v);
}

int sbo_match_templates(const sbo_pyramid* p, const sbm_template_level* levels,
                        const sbm_feature* feats, int n_templates, const int32_t* class_idx,
                        const int32_t* template_id, float threshold, int n_threads,
                        sbm_match_rec* out, int64_t cap, int64_t* n_out)
{
    const int L = p->n_levels;
    const int lc = L - 1;
    const size_t simn = (size_t)(p->cols[lc] / p->T[lc]) * (p->rows[lc] / p->T[lc]);
    recvec* per = (recvec*)calloc((size_t)n_templates, sizeof(recvec));
    if (n_threads < 1) n_threads = 1;
    /* one parallel-for over templates, as line2Dup.cpp:1166-1170 */
#pragma omp parallel num_threads(n_threads)
    {
        uint16_t* sim = (uint16_t*)malloc(simn * sizeof(uint16_t));
#pragma omp for schedule(static)
        for (int t = 0; t < n_templates; ++t)
            match_one(p, levels + (size_t)t * L, feats, class_idx ? class_idx[t] : 0,
                      template_id ? template_id[t] : t, threshold, &per[t], sim);
        free(sim);
    }
    int64_t total = 0;
    for (int t = 0; t < n_templates; ++t) total += per[t].n;
    *n_out = total;
    int rc = 0;
    if (total > cap) rc = -1;
    else {
        int64_t k = 0;
        for (int t = 0; t < n_templates; ++t) {
            if (per[t].n) memcpy(out + k, per[t].v, (size_t)per[t].n * sizeof(sbm_match_rec));
            k += per[t].n;
        }
    }
    for (int t = 0; t < n_templates; ++t) free(per[t].v);
    free(per);
    return rc;
}

static int rec_cmp(const void* a, const void* b)
{
    const sbm_match_rec* x = (const sbm_match_rec*)a;
    const sbm_match_rec* y = (const sbm_match_rec*)b;
    if (x->similarity != y->similarity) return x->similarity > y->similarity ? -1 : 1;
    if (x->template_id != y->template_id) return x->template_id < y->template_id ? -1 : 1;
    if (x->class_idx != y->class_idx) return x->class_idx < y->class_idx ? -1 : 1;
    if (x->y != y->y) return x->y < y->y ? -1 : 1;
    if (x->x != y->x) return x->x < y->x ? -1 : 1;
    return 0;
}

int64_t sbo_canonicalize(sbm_match_rec* recs, int64_t n)
{
    if (n <= 0) return 0;
    qsort(recs, (size_t)n, sizeof(sbm_match_rec), rec_cmp);
    int64_t k = 1;
    for (int64_t i = 1; i < n; ++i)
        if (rec_cmp(&recs[i], &recs[k - 1]) != 0) recs[k++] = recs[i];
    return k;
}

int64_t sbo_coarse_bytes(const sbo_pyramid* p, const sbm_template_level* levels,
                         const sbm_feature* feats, int n_templates)
{
    const int L = p->n_levels, l = L - 1, T = p->T[l];
    const int W = p->cols[l] / T, H = p->rows[l] / T;
    int64_t total = 0;
    for (int t = 0; t < n_templates; ++t) {
        const sbm_template_level* tl = &levels[(size_t)t * L + l];
        int npos = template_positions(tl, W, H, T);
        if (npos <= 0) continue;
        for (int i = 0; i < tl->n_features; ++i) {
            const sbm_feature f = feats[tl->feature_offset + i];
            if (f.x < 0 || f.x >= p->cols[l] || f.y < 0 || f.y >= p->rows[l]) continue;
            total += npos;
        }
    }
    return total;
}

/* ------------------------------------------------------------------------- */
/* training path (pins the gradient stage through the reference's fixtures)   */
/* ------------------------------------------------------------------------- */
typedef struct {
    int x, y, label;
    float theta, score;
    int order;
} cand_t;

static int cand_cmp(const void* a, const void* b)
{ /* stable_sort by score desc (:176-179, :522): ties keep scan order */
    const cand_t* x = (const cand_t*)a;
    const cand_t* y = (const cand_t*)b;
    if (x->score != y->score) return x->score > y->score ? -1 : 1;
    return x->order < y->order ? -1 : (x->order > y->order ? 1 : 0);
}

/* selectScatteredFeatures.  line2Dup.cpp:163-212. */
static int select_scattered(const cand_t* cands, int n_cands, size_t num_features, float distance,
                            sbo_train_feature* out, int64_t max_out)
{
    int n = 0, i = 0, first_select = 1;
    float distance_sq = distance * distance;
    for (;;) {
        const cand_t c = cands[i];
        int keep = 1;
        for (int j = 0; j < n && keep; ++j) {
            int ddx = c.x - out[j].x, ddy = c.y - out[j].y;
            keep = (float)(ddx * ddx + ddy * ddy) >= distance_sq;
        }
        if (keep) {
            if (n >= max_out) return -1;
            out[n].x = c.x;
            out[n].y = c.y;
            out[n].label = c.label;
            out[n].theta = c.theta;
            ++n;
        }
        if (++i == n_cands) {
            int num_ok = (size_t)n >= num_features;
            if (first_select) {
                if (num_ok) {
                    n = 0;
                    i = 0;
                    distance += 1.0f;
                    distance_sq = distance * distance;
                    continue;
                } else
                    first_select = 0;
            }
            i = 0;
            distance -= 1.0f;
            distance_sq = distance * distance;
            if (num_ok || distance < 3) break;
        }
    }
    return n;
}

/* extractTemplate.  line2Dup.cpp:452-539.  Returns feature count or -1. */
static int extract_template(const float* magnitude, const uint8_t* angle, const float* angle_ori,
                            const uint8_t* mask, int rows, int cols, size_t num_features,
                            float strong_threshold, sbo_train_feature* out, int64_t max_out)
{
    const size_t npx = (size_t)rows * cols;
    uint8_t* local_mask = NULL;
    if (mask) { /* erode 3x3, BORDER_REPLICATE :458 */
        local_mask = (uint8_t*)malloc(npx);
        for (int r = 0; r < rows; ++r)
            for (int c = 0; c < cols; ++c) {
                uint8_t m = 255;
                for (int dr = -1; dr <= 1; ++dr)
                    for (int dc = -1; dc <= 1; ++dc) {
                        uint8_t v = mask[(size_t)clampi(r + dr, 0, rows - 1) * cols + clampi(c + dc, 0, cols - 1)];
                        if (v < m) m = v;
                    }
                local_mask[(size_t)r * cols + c] = m;
            }
    }
    uint8_t* valid = (uint8_t*)malloc(npx);
    memset(valid, 255, npx);
    cand_t* cands = NULL;
    int n_cands = 0, cap = 0;
    const float thr_sq = strong_threshold * strong_threshold;
    for (int r = 2; r < rows - 2; ++r)
        for (int c = 2; c < cols - 2; ++c) {
            if (local_mask && !local_mask[(size_t)r * cols + c]) continue;
            float score = 0;
            if (valid[(size_t)r * cols + c] > 0) {
                score = magnitude[(size_t)r * cols + c];
                int is_max = 1;
                for (int dr = -2; dr <= 2 && is_max; ++dr)
                    for (int dc = -2; dc <= 2; ++dc) {
                        if (dr == 0 && dc == 0) continue;
                        if (score < magnitude[(size_t)(r + dr) * cols + c + dc]) {
                            score = 0;
                            is_max = 0;
                            break;
                        }
                    }
                if (is_max)
                    for (int dr = -2; dr <= 2; ++dr)
                        for (int dc = -2; dc <= 2; ++dc) {
                            if (dr == 0 && dc == 0) continue;
                            valid[(size_t)(r + dr) * cols + c + dc] = 0;
                        }
            }
            uint8_t a = angle[(size_t)r * cols + c];
            if (score > thr_sq && a > 0) {
                if (n_cands == cap) {
                    cap = cap ? cap * 2 : 256;
                    cands = (cand_t*)realloc(cands, (size_t)cap * sizeof(cand_t));
                }
                int label = 0;
                while (!((a >> label) & 1)) ++label;
                cand_t cd = {c, r, label, angle_ori[(size_t)r * cols + c], score, n_cands};
                cands[n_cands++] = cd;
            }
        }
    free(valid);
    free(local_mask);
    int n = -1;
    if (!((size_t)n_cands < num_features && n_cands <= 4)) { /* :513-519 */
        qsort(cands, (size_t)n_cands, sizeof(cand_t), cand_cmp);
        float distance = (float)((size_t)n_cands / num_features + 1); /* :525 */
        n = select_scattered(cands, n_cands, num_features, distance, out, max_out);
    }
    free(cands);
    return n;
}

/* cropTemplates.  line2Dup.cpp:115-161. */
static void crop_templates(sbm_template_level* lv, sbo_train_feature* feats, int n_levels)
{
    int min_x = INT32_MAX, min_y = INT32_MAX, max_x = INT32_MIN, max_y = INT32_MIN;
    for (int l = 0; l < n_levels; ++l)
        for (int j = 0; j < lv[l].n_features; ++j) {
            const sbo_train_feature* f = &feats[lv[l].feature_offset + j];
            int x = f->x << lv[l].pyramid_level, y = f->y << lv[l].pyramid_level;
            if (x < min_x) min_x = x;
            if (y < min_y) min_y = y;
            if (x > max_x) max_x = x;
            if (y > max_y) max_y = y;
        }
    if (min_x % 2 == 1) --min_x;
    if (min_y % 2 == 1) --min_y;
    for (int l = 0; l < n_levels; ++l) {
        const int s = lv[l].pyramid_level;
        lv[l].width = (max_x - min_x) >> s;
        lv[l].height = (max_y - min_y) >> s;
        lv[l].tl_x = min_x >> s;
        lv[l].tl_y = min_y >> s;
        for (int j = 0; j < lv[l].n_features; ++j) {
            feats[lv[l].feature_offset + j].x -= lv[l].tl_x;
            feats[lv[l].feature_offset + j].y -= lv[l].tl_y;
        }
    }
}

/* Detector::addTemplate.  line2Dup.cpp:1299-1353. */
int sbo_add_template(const uint8_t* img, int rows, int cols, int stride, int ch,
                     const uint8_t* mask, int n_levels, float weak, float strong,
                     int num_features, sbm_template_level* out_levels,
                     sbo_train_feature* out_feats, int64_t max_feats)
{
    uint8_t* cur = (uint8_t*)malloc((size_t)rows * cols * ch);
    for (int r = 0; r < rows; ++r) memcpy(cur + (size_t)r * cols * ch, img + (size_t)r * stride, (size_t)cols * ch);
    uint8_t* cmask = NULL;
    if (mask) {
        cmask = (uint8_t*)malloc((size_t)rows * cols);
        memcpy(cmask, mask, (size_t)rows * cols);
    }
    int cr = rows, cc = cols, ok = 1;
    size_t nfeat = (size_t)num_features;
    int64_t used = 0;
    for (int l = 0; l < n_levels && ok; ++l) {
        if (l > 0) {
            nfeat /= 2; /* :427 */
            int nr = cr / 2, nc = cc / 2;
            uint8_t* nxt = (uint8_t*)malloc((size_t)nr * nc * ch);
            sbo_pyrdown(cur, cr, cc, ch, cc * ch, nxt);
            free(cur);
            cur = nxt;
            if (cmask) {
                uint8_t* nm = (uint8_t*)malloc((size_t)nr * nc);
                sbo_resize_nearest_u8(cmask, cr, cc, nm, nr, nc);
                free(cmask);
                cmask = nm;
            }
            cr = nr;
            cc = nc;
        }
        const size_t npx = (size_t)cr * cc;
        float* mag = (float*)malloc(npx * sizeof(float));
        float* ori = (float*)malloc(npx * sizeof(float));
        uint8_t* ang = (uint8_t*)malloc(npx);
        sbo_quantized_orientations(cur, cr, cc, ch, cc * ch, weak, mag, ang, ori);
        int n = extract_template(mag, ang, ori, cmask, cr, cc, nfeat, strong, out_feats + used, max_feats - used);
        free(mag);
        free(ori);
        free(ang);
        if (n < 0) {
            ok = 0;
            break;
        }
        out_levels[l].width = -1;
        out_levels[l].height = -1;
        out_levels[l].tl_x = out_levels[l].tl_y = 0;
        out_levels[l].pyramid_level = l;
        out_levels[l].n_features = n;
        out_levels[l].feature_offset = used;
        used += n;
    }
    free(cur);
    free(cmask);
    if (!ok) return -1;
    crop_templates(out_levels, out_feats, n_levels);
    return n_levels;
}

/* Detector::addTemplate_rotate.  line2Dup.cpp:1395-1451. */
int sbo_add_template_rotate(const sbm_template_level* in_levels, const sbo_train_feature* in_feats,
                            int n_levels, float theta, float center_x, float center_y,
                            sbm_template_level* out_levels, sbo_train_feature* out_feats)
{
    float cx = center_x, cy = center_y;
    const double ang = -theta / 180 * 3.1415926535897932384626433832795; /* CV_PI */
    for (int l = 0; l < n_levels; ++l) {
        if (l > 0) {
            cx /= 2;
            cy /= 2;
        }
        out_levels[l] = in_levels[l];
        out_levels[l].pyramid_level = l;
        for (int j = 0; j < in_levels[l].n_features; ++j) {
            const sbo_train_feature* f = &in_feats[in_levels[l].feature_offset + j];
            float px = (float)(f->x + in_levels[l].tl_x), py = (float)(f->y + in_levels[l].tl_y);
            float qx = px - cx, qy = py - cy;
            float rx = (float)(cos(ang) * qx - sin(ang) * qy);
            float ry = (float)(sin(ang) * qx + cos(ang) * qy);
            rx = rx + cx;
            ry = ry + cy;
            sbo_train_feature g;
            g.x = (int)(rx + 0.5f);
            g.y = (int)(ry + 0.5f);
            g.theta = f->theta - theta;
            while (g.theta > 360) g.theta -= 360;
            while (g.theta < 0) g.theta += 360;
            g.label = (int)(g.theta * 16 / 360 + 0.5f);
            g.label &= 7;
            out_feats[in_levels[l].feature_offset + j] = g;
        }
    }
    crop_templates(out_levels, out_feats, n_levels);
    return n_levels;
}
