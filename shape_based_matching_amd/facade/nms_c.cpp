// C entry point over include/nms.hpp so that non-C++ hosts (and the tests) can call it.
#include "../../include/nms.hpp"

extern "C" int sbm_nms_boxes(const int* boxes_xywh, const float* scores, int n, float score_threshold, float nms_threshold,
                             float eta, int top_k, int* out_indices, int* n_out)
{
    try {
        std::vector<cv::Rect> boxes((size_t)n);
        std::vector<float> sc(scores, scores + n);
        for (int i = 0; i < n; ++i) boxes[i] = cv::Rect(boxes_xywh[4 * i], boxes_xywh[4 * i + 1], boxes_xywh[4 * i + 2], boxes_xywh[4 * i + 3]);
        std::vector<int> idx;
        cv_dnn::NMSBoxes(boxes, sc, score_threshold, nms_threshold, idx, eta, top_k);
        for (size_t i = 0; i < idx.size(); ++i) out_indices[i] = idx[i];
        *n_out = (int)idx.size();
        return 0;
    } catch (...) {
        return -1;
    }
}
