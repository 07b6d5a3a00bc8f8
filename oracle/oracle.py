"""ctypes binding of the CPU oracle (oracle/libsbm_oracle.so).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  The product package never imports this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import List, Optional, Sequence, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libsbm_oracle.so")

FEATURE_DTYPE = np.dtype([("x", "<i4"), ("y", "<i4"), ("label", "<i4")])
LEVEL_DTYPE = np.dtype(
    [("width", "<i4"), ("height", "<i4"), ("tl_x", "<i4"), ("tl_y", "<i4"), ("pyramid_level", "<i4"),
     ("n_features", "<i4"), ("feature_offset", "<i8")]
)
MATCH_DTYPE = np.dtype(
    [("x", "<i4"), ("y", "<i4"), ("similarity", "<f4"), ("raw", "<i4"), ("class_idx", "<i4"), ("template_id", "<i4")]
)
TRAIN_FEATURE_DTYPE = np.dtype([("x", "<i4"), ("y", "<i4"), ("label", "<i4"), ("theta", "<f4")])


def build(force: bool = False) -> str:
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(
        os.path.join(_HERE, "sbm_oracle.c")
    ):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB_PATH


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        vp, i32, i64, f32 = C.c_void_p, C.c_int, C.c_int64, C.c_float
        L.sbo_gaussian7.argtypes = [vp, i32, i32, i32, i32, vp]
        L.sbo_sobel3.argtypes = [vp, i32, i32, i32, vp, vp]
        L.sbo_fast_atan2_deg.argtypes = [f32, f32]
        L.sbo_fast_atan2_deg.restype = f32
        L.sbo_orientation_bins.argtypes = [vp, vp, i64, vp]
        L.sbo_pyrdown.argtypes = [vp, i32, i32, i32, i32, vp]
        L.sbo_resize_linear_dims.argtypes = [i32, i32, C.c_double, C.c_double, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.sbo_resize_linear_dims.restype = None
        L.sbo_resize_linear_u8.argtypes = [vp, i32, i32, i32, i32, C.c_double, C.c_double, vp]
        L.sbo_resize_linear_u8.restype = None
        L.sbo_quantized_orientations.argtypes = [vp, i32, i32, i32, i32, f32, vp, vp, vp]
        L.sbo_spread.argtypes = [vp, i32, i32, i32, vp]
        L.sbo_response_maps.argtypes = [vp, i64, vp]
        L.sbo_linearize.argtypes = [vp, i32, i32, i32, vp]
        L.sbo_set_build_threads.argtypes = [i32]
        L.sbo_set_build_threads.restype = None
        L.sbo_pyramid_build.argtypes = [vp, i32, i32, i32, i32, vp, i32, vp, f32]
        L.sbo_pyramid_build.restype = vp
        L.sbo_pyramid_from_quantized.argtypes = [vp, vp, vp, i32, vp]
        L.sbo_pyramid_from_quantized.restype = vp
        L.sbo_pyramid_free.argtypes = [vp]
        for n in ("sbo_pyramid_rows", "sbo_pyramid_cols"):
            getattr(L, n).argtypes = [vp, i32]
            getattr(L, n).restype = i32
        L.sbo_pyramid_lm_stride.argtypes = [vp, i32]
        L.sbo_pyramid_lm_stride.restype = i64
        L.sbo_pyramid_lm.argtypes = [vp, i32]
        L.sbo_pyramid_lm.restype = vp
        L.sbo_pyramid_quantized.argtypes = [vp, i32]
        L.sbo_pyramid_quantized.restype = vp
        L.sbo_similarity.argtypes = [vp, i64, i32, i32, i32, vp, vp, vp]
        L.sbo_similarity_local.argtypes = [vp, i64, i32, i32, i32, vp, vp, i32, i32, vp]
        L.sbo_match_templates.argtypes = [vp, vp, vp, i32, vp, vp, f32, i32, vp, i64, vp]
        L.sbo_match_templates.restype = i32
        L.sbo_canonicalize.argtypes = [vp, i64]
        L.sbo_canonicalize.restype = i64
        L.sbo_coarse_bytes.argtypes = [vp, vp, vp, i32]
        L.sbo_coarse_bytes.restype = i64
        L.sbo_add_template.argtypes = [vp, i32, i32, i32, i32, vp, i32, f32, f32, i32, vp, vp, i64]
        L.sbo_add_template.restype = i32
        L.sbo_add_template_rotate.argtypes = [vp, vp, i32, f32, f32, f32, vp, vp]
        L.sbo_add_template_rotate.restype = i32
        _lib = L
    return _lib


def _p(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _img(img: np.ndarray) -> Tuple[np.ndarray, int, int, int]:
    img = np.ascontiguousarray(img, dtype=np.uint8)
    if img.ndim == 2:
        return img, img.shape[0], img.shape[1], 1
    return img, img.shape[0], img.shape[1], img.shape[2]


# ---- gradient stage -------------------------------------------------------
def gaussian7(img: np.ndarray) -> np.ndarray:
    img, r, c, ch = _img(img)
    out = np.empty_like(img)
    lib().sbo_gaussian7(_p(img), r, c, ch, c * ch, _p(out))
    return out


def sobel3(sm: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    sm, r, c, ch = _img(sm)
    dx = np.empty(sm.shape, np.int16)
    dy = np.empty(sm.shape, np.int16)
    lib().sbo_sobel3(_p(sm), r, c, ch, _p(dx), _p(dy))
    return dx, dy


def fast_atan2_deg(y: float, x: float) -> float:
    return float(lib().sbo_fast_atan2_deg(C.c_float(y), C.c_float(x)))


def orientation_bins(dx: np.ndarray, dy: np.ndarray) -> np.ndarray:
    """16-bin index of cvRound(fastAtan2(dy, dx) * 16/360) for int16 gradient arrays."""
    dx = np.ascontiguousarray(dx, np.int16)
    dy = np.ascontiguousarray(dy, np.int16)
    out = np.empty(dx.shape, np.uint8)
    lib().sbo_orientation_bins(_p(dx), _p(dy), dx.size, _p(out))
    return out


def pyrdown(img: np.ndarray) -> np.ndarray:
    img, r, c, ch = _img(img)
    shape = (r // 2, c // 2) if img.ndim == 2 else (r // 2, c // 2, ch)
    out = np.empty(shape, np.uint8)
    lib().sbo_pyrdown(_p(img), r, c, ch, c * ch, _p(out))
    return out


def resize_linear(img: np.ndarray, fx: float, fy: float) -> np.ndarray:
    """cv::resize(img, Size(), fx, fy, INTER_LINEAR) for 8-bit images (shapeInfo_producer::transform)."""
    img, r, c, ch = _img(img)
    dr, dc = C.c_int(0), C.c_int(0)
    lib().sbo_resize_linear_dims(r, c, C.c_double(fx), C.c_double(fy), C.byref(dr), C.byref(dc))
    shape = (dr.value, dc.value) if img.ndim == 2 else (dr.value, dc.value, ch)
    out = np.empty(shape, np.uint8)
    lib().sbo_resize_linear_u8(_p(img), r, c, ch, c * ch, C.c_double(fx), C.c_double(fy), _p(out))
    return out


def quantized_orientations(img: np.ndarray, weak: float):
    """Returns (magnitude f32, one-hot angle u8, angle_ori f32)."""
    img, r, c, ch = _img(img)
    mag = np.empty((r, c), np.float32)
    ang = np.empty((r, c), np.uint8)
    ori = np.empty((r, c), np.float32)
    lib().sbo_quantized_orientations(_p(img), r, c, ch, c * ch, C.c_float(weak), _p(mag), _p(ang), _p(ori))
    return mag, ang, ori


# ---- response maps --------------------------------------------------------
def spread(q: np.ndarray, T: int) -> np.ndarray:
    q = np.ascontiguousarray(q, np.uint8)
    out = np.empty_like(q)
    lib().sbo_spread(_p(q), q.shape[0], q.shape[1], T, _p(out))
    return out


def response_maps(sp: np.ndarray) -> np.ndarray:
    sp = np.ascontiguousarray(sp, np.uint8)
    out = np.empty((8,) + sp.shape, np.uint8)
    lib().sbo_response_maps(_p(sp), sp.size, _p(out))
    return out


def linearize(m: np.ndarray, T: int) -> np.ndarray:
    m = np.ascontiguousarray(m, np.uint8)
    r, c = m.shape
    out = np.empty((T * T, (r // T) * (c // T)), np.uint8)
    lib().sbo_linearize(_p(m), r, c, T, _p(out))
    return out


# ---- pyramid ---------------------------------------------------------------
class Pyramid:
    def __init__(self, handle, n_levels: int, T: Sequence[int]):
        if not handle:
            raise ValueError("oracle: image size not compatible with T (rows%T, cols%T, (rows*cols)%16)")
        self.h = C.c_void_p(handle)
        self.n_levels = n_levels
        self.T = list(T)

    @staticmethod
    def build(img: np.ndarray, T: Sequence[int], weak: float = 30.0, mask: Optional[np.ndarray] = None) -> "Pyramid":
        img, r, c, ch = _img(img)
        Ta = np.asarray(T, np.int32)
        m = None if mask is None else np.ascontiguousarray(mask, np.uint8)
        h = lib().sbo_pyramid_build(_p(img), r, c, c * ch, ch, _p(m), len(T), _p(Ta), C.c_float(weak))
        return Pyramid(h, len(T), T)

    @staticmethod
    def from_quantized(qs: Sequence[np.ndarray], T: Sequence[int]) -> "Pyramid":
        qs = [np.ascontiguousarray(q, np.uint8) for q in qs]
        ptrs = (C.c_void_p * len(qs))(*[q.ctypes.data for q in qs])
        rows = np.asarray([q.shape[0] for q in qs], np.int32)
        cols = np.asarray([q.shape[1] for q in qs], np.int32)
        Ta = np.asarray(T, np.int32)
        h = lib().sbo_pyramid_from_quantized(ptrs, _p(rows), _p(cols), len(qs), _p(Ta))
        return Pyramid(h, len(qs), T)

    def rows(self, l: int) -> int:
        return lib().sbo_pyramid_rows(self.h, l)

    def cols(self, l: int) -> int:
        return lib().sbo_pyramid_cols(self.h, l)

    def lm_stride(self, l: int) -> int:
        return lib().sbo_pyramid_lm_stride(self.h, l)

    def lm(self, l: int) -> np.ndarray:
        """[8, lm_stride] view (copy) of the flat linear memories of level l."""
        s = self.lm_stride(l)
        buf = (C.c_uint8 * (8 * s)).from_address(lib().sbo_pyramid_lm(self.h, l))
        return np.frombuffer(buf, np.uint8).reshape(8, s).copy()

    def quantized(self, l: int) -> np.ndarray:
        r, c = self.rows(l), self.cols(l)
        buf = (C.c_uint8 * (r * c)).from_address(lib().sbo_pyramid_quantized(self.h, l))
        return np.frombuffer(buf, np.uint8).reshape(r, c).copy()

    def similarity(self, level_rec: np.ndarray, feats: np.ndarray, l: int) -> np.ndarray:
        T = self.T[l]
        r, c = self.rows(l), self.cols(l)
        out = np.empty((r // T, c // T), np.uint16)
        lv = np.ascontiguousarray(level_rec.reshape(1), LEVEL_DTYPE)
        lib().sbo_similarity(lib().sbo_pyramid_lm(self.h, l), self.lm_stride(l), r, c, T, _p(lv), _p(feats), _p(out))
        return out

    def similarity_local(self, level_rec: np.ndarray, feats: np.ndarray, l: int, cx: int, cy: int) -> np.ndarray:
        T = self.T[l]
        out = np.empty((16, 16), np.uint16)
        lv = np.ascontiguousarray(level_rec.reshape(1), LEVEL_DTYPE)
        lib().sbo_similarity_local(
            lib().sbo_pyramid_lm(self.h, l), self.lm_stride(l), self.rows(l), self.cols(l), T, _p(lv), _p(feats), cx, cy, _p(out)
        )
        return out

    def match(self, levels: np.ndarray, feats: np.ndarray, class_idx, template_id, threshold: float,
              n_threads: int = 1, cap: int = 1 << 20) -> np.ndarray:
        """Pre-dedup match multiset (template order, then coarse row-major order)."""
        levels = np.ascontiguousarray(levels, LEVEL_DTYPE)
        feats = np.ascontiguousarray(feats, FEATURE_DTYPE)
        n = levels.shape[0]
        ci = None if class_idx is None else np.ascontiguousarray(class_idx, np.int32)
        ti = None if template_id is None else np.ascontiguousarray(template_id, np.int32)
        while True:
            out = np.empty(cap, MATCH_DTYPE)
            n_out = C.c_int64(0)
            rc = lib().sbo_match_templates(self.h, _p(levels), _p(feats), n, _p(ci), _p(ti), C.c_float(threshold),
                                           n_threads, _p(out), cap, C.byref(n_out))
            if rc == 0:
                return out[: n_out.value].copy()
            cap = int(n_out.value)

    def coarse_bytes(self, levels: np.ndarray, feats: np.ndarray) -> int:
        levels = np.ascontiguousarray(levels, LEVEL_DTYPE)
        feats = np.ascontiguousarray(feats, FEATURE_DTYPE)
        return int(lib().sbo_coarse_bytes(self.h, _p(levels), _p(feats), levels.shape[0]))

    def free(self):
        if self.h:
            lib().sbo_pyramid_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def canonicalize(recs: np.ndarray) -> np.ndarray:
    recs = np.ascontiguousarray(recs, MATCH_DTYPE).copy()
    n = lib().sbo_canonicalize(_p(recs), len(recs))
    return recs[:n]


def set_build_threads(n: int) -> None:
    """threads of the pyramid build's row loops (1 = serial, the default); results do not depend on it"""
    lib().sbo_set_build_threads(int(n))


def match_set(recs: np.ndarray) -> set:
    """Distinct (x, y, similarity bits, class_idx, template_id) tuples — the parity contract."""
    r = np.ascontiguousarray(recs, MATCH_DTYPE)
    bits = r["similarity"].view(np.uint32)
    return set(zip(r["x"].tolist(), r["y"].tolist(), bits.tolist(), r["class_idx"].tolist(), r["template_id"].tolist()))


# ---- training path ---------------------------------------------------------
def add_template(img: np.ndarray, mask: Optional[np.ndarray], n_levels: int, num_features: int,
                 weak: float = 30.0, strong: float = 60.0):
    """Detector::addTemplate.  Returns (levels[n_levels], train_feats) or None on failure."""
    img, r, c, ch = _img(img)
    m = None if mask is None else np.ascontiguousarray(mask, np.uint8)
    levels = np.zeros(n_levels, LEVEL_DTYPE)
    cap = r * c
    feats = np.zeros(cap, TRAIN_FEATURE_DTYPE)
    rc = lib().sbo_add_template(_p(img), r, c, c * ch, ch, _p(m), n_levels, C.c_float(weak), C.c_float(strong),
                                num_features, _p(levels), _p(feats), cap)
    if rc < 0:
        return None
    n = int(levels["n_features"].sum())
    return levels, feats[:n].copy()


def add_template_rotate(levels: np.ndarray, feats: np.ndarray, theta: float, center: Tuple[float, float]):
    out_l = np.zeros_like(levels)
    out_f = np.zeros_like(feats)
    lib().sbo_add_template_rotate(_p(levels), _p(feats), len(levels), C.c_float(theta), C.c_float(center[0]),
                                  C.c_float(center[1]), _p(out_l), _p(out_f))
    return out_l, out_f
