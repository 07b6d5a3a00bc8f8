/*
 * sbm_oracle.h — CPU ORACLE.  TEST INFRASTRUCTURE ONLY.
 *
 * A plain-C restatement of the reference's match() hot path (and of the
 * training path that produced the reference's template fixtures, which is what
 * pins the gradient stage).  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this library; the product
 * (shape_based_matching_amd/) never links, imports or calls it.
 *
 * Pinning status (details in DESIGN.md "Oracle"):
 *   - The reference cannot be built in this image (it needs OpenCV 4, absent),
 *     so this restatement is not checked against a compiled reference.
 *   - computeResponseMaps: pinned against the literal SIMILARITY_LUT table.
 *   - quantizedOrientations / hysteresisGradient / pyrDown / extractTemplate /
 *     selectScatteredFeatures / cropTemplates / addTemplate_rotate: pinned by
 *     re-training the reference's committed template fixtures (test/case1,
 *     test/case2 template YAMLs) from the reference's training images.
 *   - spread / linearize / similarity* / matchClass: restated line by line from
 *     the source; the reference holds no golden vector for them:
 *     PARITY UNPINNED for these stages beyond the end-to-end case runs.
 *
 * All citations are file:line in ddcr/shape_based_matching.
 */
#ifndef SBM_ORACLE_H
#define SBM_ORACLE_H

#include <stdint.h>
#include "../include/sbm_types.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- gradient stage (line2Dup.cpp:218-450; OpenCV-4 primitive semantics) ---- */
void sbo_gaussian7(const uint8_t* src, int rows, int cols, int ch, int stride, uint8_t* dst);
void sbo_sobel3(const uint8_t* sm, int rows, int cols, int ch, int16_t* dx, int16_t* dy);
float sbo_fast_atan2_deg(float y, float x);
void sbo_orientation_bins(const int16_t* dx, const int16_t* dy, int64_t n, uint8_t* q16);
void sbo_pyrdown(const uint8_t* src, int rows, int cols, int ch, int stride, uint8_t* dst);
void sbo_resize_nearest_u8(const uint8_t* src, int rows, int cols, uint8_t* dst, int drows, int dcols);
void sbo_resize_linear_dims(int rows, int cols, double fx, double fy, int* drows, int* dcols);
void sbo_resize_linear_u8(const uint8_t* src, int rows, int cols, int ch, int stride, double fx, double fy, uint8_t* dst);
/* quantizedOrientations + hysteresisGradient.  magnitude/angle_ori may be NULL. */
void sbo_quantized_orientations(const uint8_t* src, int rows, int cols, int ch, int stride,
                                float weak_threshold, float* magnitude, uint8_t* angle,
                                float* angle_ori);

/* ---- response maps (line2Dup.cpp:583-777) ---- */
void sbo_spread(const uint8_t* src, int rows, int cols, int T, uint8_t* dst);
void sbo_response_maps(const uint8_t* spread, int64_t n, uint8_t* maps /* [8][n] */);
void sbo_linearize(const uint8_t* map, int rows, int cols, int T, uint8_t* lm /* [T*T][W*H] */);

/* ---- pyramid of flat linear memories ---- */
typedef struct sbo_pyramid sbo_pyramid;
/* threads of the pyramid build's row loops (default 1); results do not depend on it */
void sbo_set_build_threads(int n);
int sbo_get_build_threads(void);
sbo_pyramid* sbo_pyramid_build(const uint8_t* img, int rows, int cols, int stride, int ch,
                               const uint8_t* mask, int n_levels, const int* T, float weak);
sbo_pyramid* sbo_pyramid_from_quantized(const uint8_t* const* q, const int* rows, const int* cols,
                                        int n_levels, const int* T);
void sbo_pyramid_free(sbo_pyramid* p);
int sbo_pyramid_rows(const sbo_pyramid* p, int level);
int sbo_pyramid_cols(const sbo_pyramid* p, int level);
int64_t sbo_pyramid_lm_stride(const sbo_pyramid* p, int level); /* bytes per orientation incl. zero tail */
const uint8_t* sbo_pyramid_lm(const sbo_pyramid* p, int level); /* [8][lm_stride] */
const uint8_t* sbo_pyramid_quantized(const sbo_pyramid* p, int level);

/* ---- similarity (line2Dup.cpp:807-1048) ---- */
/* dst: H*W uint16 (the u8 path of similarity_64 holds the same sums). */
void sbo_similarity(const uint8_t* lm, int64_t lm_stride, int rows, int cols, int T,
                    const sbm_template_level* tl, const sbm_feature* feats, uint16_t* dst);
/* dst: 256 uint16, centre (cx, cy) as passed to similarityLocal. */
void sbo_similarity_local(const uint8_t* lm, int64_t lm_stride, int rows, int cols, int T,
                          const sbm_template_level* tl, const sbm_feature* feats, int cx, int cy,
                          uint16_t* dst);

/* ---- matchClass over a flat template list (line2Dup.cpp:1160-1297) ----
 * levels: [n_templates][n_levels].  Emits the multiset BEFORE sort/unique, in
 * template order then coarse row-major order.  Returns 0, or -1 if cap is too
 * small (n_out then holds the required count). */
int sbo_match_templates(const sbo_pyramid* p, const sbm_template_level* levels,
                        const sbm_feature* feats, int n_templates, const int32_t* class_idx,
                        const int32_t* template_id, float threshold, int n_threads,
                        sbm_match_rec* out, int64_t cap, int64_t* n_out);

/* Detector::match epilogue in canonical form: sort by (similarity desc,
 * template_id asc, class_idx, y, x) and drop exact duplicates; returns new n. */
int64_t sbo_canonicalize(sbm_match_rec* recs, int64_t n);

/* algorithmic byte count of the coarse pass: sum over templates and
 * in-bounds coarsest-level features of max(template_positions, 0). */
int64_t sbo_coarse_bytes(const sbo_pyramid* p, const sbm_template_level* levels,
                         const sbm_feature* feats, int n_templates);

/* ---- training path (line2Dup.cpp:115-212, 452-539, 1299-1353, 1395-1451) ---- */
typedef struct sbo_train_feature {
    int32_t x, y, label;
    float theta;
} sbo_train_feature;
/* addTemplate: returns the number of levels written (== n_levels) or -1.
 * out_levels[n_levels]; out_feats has room for max_feats entries; features of
 * level l start at out_levels[l].feature_offset. */
int sbo_add_template(const uint8_t* img, int rows, int cols, int stride, int ch,
                     const uint8_t* mask, int n_levels, float weak, float strong,
                     int num_features, sbm_template_level* out_levels,
                     sbo_train_feature* out_feats, int64_t max_feats);
/* addTemplate_rotate on a pyramid produced by sbo_add_template. */
int sbo_add_template_rotate(const sbm_template_level* in_levels, const sbo_train_feature* in_feats,
                            int n_levels, float theta, float center_x, float center_y,
                            sbm_template_level* out_levels, sbo_train_feature* out_feats);

#ifdef __cplusplus
}
#endif
#endif
