/*
 * nms.hpp — drop-in for the reference's nms.hpp (cv_dnn::NMSBoxes, nms.hpp:91-96; also
 * duplicated in test.cpp:38-128): greedy IoU non-maximum suppression over the match boxes,
 * the step every reference caller runs right after Detector::match (test.cpp:491,
 * test_jabil.cpp:148).  Host-side post-processing; same signature and semantics:
 *   keep score > score_threshold; stable sort by score, descending; optional top_k;
 *   walk the list keeping a box iff its overlap (1 - Jaccard distance on integer Rect
 *   areas) with every kept box is <= the threshold; the threshold is multiplied by eta
 *   after each kept box while eta < 1 and the threshold is > 0.5.
 */
#ifndef SBM_NMS_HPP
#define SBM_NMS_HPP

#include <algorithm>
#include <utility>
#include <vector>

#include "line2Dup.h"

namespace cv_dnn {

inline float rectOverlap(const cv::Rect& a, const cv::Rect& b)
{
    const int area_a = a.area(), area_b = b.area();
    if (area_a + area_b <= 0) return 1.f; // two empty boxes: Jaccard distance defined as 0
    const double inter = (double)(a & b).area();
    const double distance = 1.0 - inter / (double)(area_a + area_b - inter);
    return 1.f - static_cast<float>(distance);
}

inline void NMSBoxes(const std::vector<cv::Rect>& bboxes, const std::vector<float>& scores, const float score_threshold,
                     const float nms_threshold, std::vector<int>& indices, const float eta = 1, const int top_k = 0)
{
    CV_Assert(bboxes.size() == scores.size());
    std::vector<std::pair<float, int>> order;
    for (size_t i = 0; i < scores.size(); ++i)
        if (scores[i] > score_threshold) order.emplace_back(scores[i], (int)i);
    std::stable_sort(order.begin(), order.end(), [](const std::pair<float, int>& x, const std::pair<float, int>& y) { return x.first > y.first; });
    if (top_k > 0 && top_k < (int)order.size()) order.resize(top_k);

    float threshold = nms_threshold;
    indices.clear();
    for (const auto& cand : order) {
        bool keep = true;
        for (size_t k = 0; k < indices.size() && keep; ++k) keep = rectOverlap(bboxes[cand.second], bboxes[indices[k]]) <= threshold;
        if (!keep) continue;
        indices.push_back(cand.second);
        if (eta < 1 && threshold > 0.5) threshold *= eta;
    }
}

} // namespace cv_dnn

#endif
