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
//
// Round 3: one header per stage (this file only includes them).
#pragma once
#include "sbm_common.h"
#include "sbm_quantize_tile.h"
#include "sbm_lm_kernels.h"
#include "sbm_similarity_kernels.h"
#include "sbm_coarse_bits.h"
