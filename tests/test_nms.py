"""NMS: include/nms.hpp's cv_dnn::NMSBoxes against a direct Python restatement of the
reference semantics (nms.hpp:21-96): score filter, stable sort, greedy IoU with adaptive threshold."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import ROOT

LIB = os.path.join(ROOT, "shape_based_matching_amd", "libsbm_facade.so")


def py_nms(boxes, scores, score_thr, nms_thr, eta=1.0, top_k=0):
    order = [i for i in range(len(scores)) if scores[i] > score_thr]
    order.sort(key=lambda i: -scores[i])  # Python's sort is stable
    if 0 < top_k < len(order):
        order = order[:top_k]

    def overlap(a, b):
        aa, ab = a[2] * a[3], b[2] * b[3]
        if aa + ab <= 0:
            return np.float32(1.0)
        x1, y1 = max(a[0], b[0]), max(a[1], b[1])
        x2, y2 = min(a[0] + a[2], b[0] + b[2]), min(a[1] + a[3], b[1] + b[3])
        inter = 0.0 if (x2 <= x1 or y2 <= y1) else float((x2 - x1) * (y2 - y1))
        return np.float32(1.0) - np.float32(1.0 - inter / (aa + ab - inter))

    thr = np.float32(nms_thr)
    keep = []
    for i in order:
        if all(overlap(boxes[i], boxes[k]) <= thr for k in keep):
            keep.append(i)
            if eta < 1 and thr > 0.5:
                thr = np.float32(thr * np.float32(eta))
    return keep


@pytest.mark.skipif(not os.path.exists(LIB), reason="facade not built")
@pytest.mark.parametrize("seed,eta,top_k", [(1, 1.0, 0), (2, 0.9, 0), (3, 1.0, 20), (4, 0.8, 50)])
def test_nms_boxes(seed, eta, top_k):
    L = C.CDLL(LIB)
    rs = np.random.RandomState(seed)
    n = 300
    boxes = np.stack([rs.randint(0, 400, n), rs.randint(0, 400, n), rs.randint(1, 120, n), rs.randint(1, 120, n)], axis=1).astype(np.int32)
    boxes[::17, 2] = 0  # some empty boxes
    scores = (rs.randint(0, 40, n) * 2.5).astype(np.float32)  # many ties: stable sort matters
    out = np.zeros(n, np.int32)
    n_out = C.c_int(0)
    for thr in (0.5, 0.3, 0.7):
        rc = L.sbm_nms_boxes(boxes.ctypes.data_as(C.c_void_p), scores.ctypes.data_as(C.c_void_p), n, C.c_float(10.0), C.c_float(thr),
                             C.c_float(eta), top_k, out.ctypes.data_as(C.c_void_p), C.byref(n_out))
        assert rc == 0
        assert out[: n_out.value].tolist() == py_nms(boxes.tolist(), scores.tolist(), 10.0, thr, eta, top_k)
