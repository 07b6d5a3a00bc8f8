/*
 * sbm_types.h — plain-old-data types shared by the C ABI (include/sbm.h), the
 * C++ Detector facade (include/line2Dup.h) and the CPU oracle (oracle/).
 *
 * Every struct here is the flat, pointer-free counterpart of a type on the
 * reference's match() hot path (citations are file:line in the reference
 * repository ddcr/shape_based_matching):
 *
 *   sbm_feature         <- line2Dup::Feature           line2Dup.h:116-129
 *   sbm_template_level  <- line2Dup::Template          line2Dup.h:131-153
 *   sbm_match_rec       <- line2Dup::Match             line2Dup.h:222-255
 *
 * A "template pyramid" (reference typedef TemplatePyramid, line2Dup.h:319) is
 * n_levels consecutive sbm_template_level entries, level 0 first.
 */
#ifndef SBM_TYPES_H
#define SBM_TYPES_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SBM_MAX_LEVELS 8
#define SBM_MAX_FEATURES 8191 /* line2Dup.cpp:811, :863 (features.size() < 8192) */

/* One template feature: position relative to the template's top-left corner
 * and quantised orientation label 0..7.  (Feature::theta is training-only.) */
typedef struct sbm_feature {
    int32_t x;
    int32_t y;
    int32_t label;
} sbm_feature;

/* One Template = one pyramid level of one template pyramid. */
typedef struct sbm_template_level {
    int32_t width;
    int32_t height;
    int32_t tl_x;
    int32_t tl_y;
    int32_t pyramid_level;
    int32_t n_features;     /* counts ALL features (denominator 4*nf, line2Dup.cpp:1187) */
    int64_t feature_offset; /* first feature in the flat sbm_feature array */
} sbm_template_level;

/* One Match.  `raw` is the integer similarity sum the float was derived from:
 * similarity == (raw * 100.f) / (4 * nf)   (line2Dup.cpp:1206, :1273). */
typedef struct sbm_match_rec {
    int32_t x;
    int32_t y;
    float similarity;
    int32_t raw;
    int32_t class_idx;   /* index into the caller's class-id list */
    int32_t template_id; /* index of the pyramid inside its class (line2Dup.cpp:1312) */
} sbm_match_rec;

#ifdef __cplusplus
}
#endif
#endif /* SBM_TYPES_H */
