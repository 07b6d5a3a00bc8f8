"""ctypes binding of libsbm_hip.so (the C ABI declared in include/sbm.h).

There is no CPU fallback: if the shared library is missing, or no MI355X is
visible, every call raises.  The library is built in-tree by
``__graft_entry__.build()`` / ``make -C shape_based_matching_amd/csrc``.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional, Sequence, Tuple

import numpy as np

from .templates import FEATURE_DTYPE, LEVEL_DTYPE, MATCH_DTYPE, TemplateSet

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libsbm_hip.so")
SBM_MAX_LEVELS = 8

# every symbol include/sbm.h declares (checked by tests/test_abi.py)
ABI_SYMBOLS = [
    "sbm_last_error", "sbm_abi_version", "sbm_create", "sbm_destroy", "sbm_upload_templates",
    "sbm_select_classes", "sbm_select_range", "sbm_match", "sbm_match_device", "sbm_match_batch_device", "sbm_canonicalize",
    "sbm_build_pyramid", "sbm_set_quantized", "sbm_get_quantized", "sbm_get_linear_memories",
    "sbm_level_dims", "sbm_match_templates", "sbm_quantized_orientations", "sbm_pyrdown", "sbm_spread",
    "sbm_compute_response_maps", "sbm_linearize", "sbm_similarity", "sbm_similarity_local",
    "sbm_set_profiling", "sbm_get_timings", "sbm_coarse_bytes", "sbm_get_stats",
    "sbm_set_result_mirror", "sbm_set_graph_mode", "sbm_set_quantize_mode", "sbm_resize_linear",
    "sbm_match_templates_device", "sbm_orientation_bins",
    "sbm_comm_unique_id", "sbm_comm_init", "sbm_comm_destroy", "sbm_match_device_sharded", "sbm_match_batch_device_sharded",
    "sbm_match_batch_device_banded", "sbm_pin_host_buffer", "sbm_unpin_host_buffer",
    "sbm_select_templates", "sbm_partition_templates", "sbm_match_sharded",
    "sbm_match_batch_host", "sbm_match_batch_host_begin", "sbm_match_batch_host_end", "sbm_extract_local_maxima",
    "sbm_set_pipeline_depth", "sbm_set_coarse_mode", "sbm_set_refine_order", "sbm_set_refine_bits", "sbm_get_coarse_bitplanes",
    "sbm_comm_count", "sbm_match_templates_device_sharded", "sbm_graph_count",
]


class SbmConfig(C.Structure):
    _fields_ = [
        ("n_levels", C.c_int32),
        ("T", C.c_int32 * SBM_MAX_LEVELS),
        ("weak_threshold", C.c_float),
        ("device_id", C.c_int32),
        ("max_candidates", C.c_int64),
    ]


class SbmError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"libsbm_hip error {code}: {msg}")
        self.code = code


_lib = None


def _one_hip_runtime_per_process() -> str:
    """PyTorch-ROCm wheels bundle their own HIP runtime (torch/lib/libamdhip64.so, SONAME libamdhip64.so.7) and link it
    by the unversioned file name, so it is never matched to a system runtime that is already loaded: a process that loads
    libsbm_hip.so (NEEDED libamdhip64.so.7 -> /opt/rocm) FIRST and imports torch LATER ends up with two HIP runtimes, and
    the second one to initialise finds no device ("No HIP GPUs are available" from torch.cuda, round 2).  The other order
    is fine: with torch's copy loaded first the loader resolves libsbm_hip's NEEDED entry to it by SONAME.  So, when a
    torch with a bundled runtime is installed, make that order the only one: map torch's copy before libsbm_hip.so --
    whether or not torch has been imported yet (it need not be, and is not imported here).  SBM_HIP_RUNTIME=system
    keeps the system runtime (processes that never import torch)."""
    if os.environ.get("SBM_HIP_RUNTIME", "") == "system":
        return "system"
    import importlib.util
    import sys

    t = sys.modules.get("torch")
    if t is not None:
        base = os.path.dirname(t.__file__)
    else:
        try:
            spec = importlib.util.find_spec("torch")
        except (ImportError, ValueError):
            spec = None
        if spec is None or not spec.submodule_search_locations:
            return "system"
        base = list(spec.submodule_search_locations)[0]
    rt = os.path.join(base, "lib", "libamdhip64.so")
    if not os.path.exists(rt):
        return "system"
    # Only a runtime of the SONAME libsbm_hip.so was linked against may stand in for it: another SONAME would still give
    # two runtimes, and mapping it would be pointless.  Both names are read from the ELF dynamic sections.
    want = [n for n in _elf_dynamic_strings(LIB_PATH, 1) if n.startswith("libamdhip64")]   # DT_NEEDED
    have = _elf_dynamic_strings(rt, 14)                                                      # DT_SONAME
    if want and have and have[0] not in want:
        import warnings

        warnings.warn(f"torch bundles {have[0]} but libsbm_hip.so needs {want[0]}: keeping the system HIP runtime "
                      "(import torch before creating a context, or expect two runtimes in this process)")
        return "system"
    C.CDLL(rt, mode=C.RTLD_GLOBAL)
    return rt


def _elf_dynamic_strings(path: str, tag: int):
    """the strings of the ELF64 little-endian dynamic entries with this tag (1 = DT_NEEDED, 14 = DT_SONAME); [] on any surprise"""
    import struct

    try:
        with open(path, "rb") as f:
            data = f.read()
        if data[:6] != b"\x7fELF\x02\x01":
            return []
        shoff, = struct.unpack_from("<Q", data, 0x28)
        shentsize, shnum = struct.unpack_from("<HH", data, 0x3A)
        secs = [struct.unpack_from("<IIQQQQIIQQ", data, shoff + i * shentsize) for i in range(shnum)]
        out = []
        for sec in secs:
            if sec[1] != 6:  # SHT_DYNAMIC
                continue
            strtab = secs[sec[6]]  # sh_link
            for off in range(sec[4], sec[4] + sec[5], 16):
                t, v = struct.unpack_from("<qQ", data, off)
                if t == 0:
                    break
                if t == tag:
                    s0 = strtab[4] + v
                    out.append(data[s0:data.index(b"\0", s0)].decode())
        return out
    except Exception:
        return []


HIP_RUNTIME = None  # which HIP runtime lib() made the process's one: "system" or the path of torch's copy


def lib() -> C.CDLL:
    """Load libsbm_hip.so; raises if the HIP extension has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: the HIP extension is not built (run __graft_entry__.build()); "
            "shape_based_matching_amd has no CPU fallback"
        )
    global HIP_RUNTIME
    HIP_RUNTIME = _one_hip_runtime_per_process()
    L = C.CDLL(LIB_PATH)
    vp, i32, i64, f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float
    L.sbm_last_error.restype = C.c_char_p
    L.sbm_abi_version.restype = i32
    L.sbm_create.argtypes = [C.POINTER(SbmConfig), C.POINTER(vp)]
    L.sbm_destroy.argtypes = [vp]
    L.sbm_destroy.restype = None
    L.sbm_upload_templates.argtypes = [vp, i32, vp, vp, i64, vp, vp]
    L.sbm_select_classes.argtypes = [vp, vp, i32]
    L.sbm_select_range.argtypes = [vp, i32, i32]
    L.sbm_match.argtypes = [vp, vp, i32, i32, i32, i32, vp, f32, vp, i64, C.POINTER(i64)]
    L.sbm_match_device.argtypes = [vp, vp, i32, i32, i32, i32, vp, f32, vp, i64, vp, vp]
    L.sbm_match_batch_device.argtypes = [vp, vp, i64, i32, i32, i32, i32, i32, vp, f32, vp, i64, vp, vp]
    L.sbm_canonicalize.argtypes = [vp, i64]
    L.sbm_canonicalize.restype = i64
    L.sbm_build_pyramid.argtypes = [vp, vp, i32, i32, i32, i32, vp]
    L.sbm_set_quantized.argtypes = [vp, i32, vp, i32, i32]
    L.sbm_get_quantized.argtypes = [vp, i32, vp]
    L.sbm_get_linear_memories.argtypes = [vp, i32, vp, i64, C.POINTER(i64)]
    L.sbm_get_coarse_bitplanes.argtypes = [vp, i32, vp, i64]
    L.sbm_level_dims.argtypes = [vp, i32, C.POINTER(i32), C.POINTER(i32)]
    L.sbm_match_templates.argtypes = [vp, f32, vp, i64, C.POINTER(i64)]
    L.sbm_quantized_orientations.argtypes = [vp, vp, i32, i32, i32, i32, f32, vp, vp, vp]
    L.sbm_pyrdown.argtypes = [vp, vp, i32, i32, i32, i32, vp]
    L.sbm_orientation_bins.argtypes = [vp, vp, vp, i64, vp]
    L.sbm_spread.argtypes = [vp, vp, i32, i32, i32, vp]
    L.sbm_compute_response_maps.argtypes = [vp, vp, i32, i32, vp]
    L.sbm_linearize.argtypes = [vp, vp, i32, i32, i32, vp]
    L.sbm_similarity.argtypes = [vp, i32, vp]
    L.sbm_similarity_local.argtypes = [vp, i32, i32, i32, i32, vp]
    L.sbm_set_profiling.argtypes = [vp, i32]
    L.sbm_get_timings.argtypes = [vp, vp, vp, i32, C.POINTER(i32)]
    L.sbm_coarse_bytes.argtypes = [vp, C.POINTER(i64)]
    L.sbm_get_stats.argtypes = [vp, C.POINTER(i64), C.POINTER(i64)]
    L.sbm_set_result_mirror.argtypes = [vp, vp, vp]
    L.sbm_set_graph_mode.argtypes = [vp, i32]
    L.sbm_graph_count.argtypes = [vp, C.POINTER(i32)]
    L.sbm_set_quantize_mode.argtypes = [vp, i32, i32]
    L.sbm_resize_linear.argtypes = [vp, vp, i32, i32, i32, i32, C.c_double, C.c_double, vp, i64, C.POINTER(i32), C.POINTER(i32)]
    L.sbm_match_templates_device.argtypes = [vp, f32, vp, i64, vp, vp]
    L.sbm_comm_unique_id.argtypes = [vp]
    L.sbm_comm_init.argtypes = [vp, i32, i32, vp]
    L.sbm_comm_count.argtypes = [vp, C.POINTER(i32)]
    L.sbm_match_templates_device_sharded.argtypes = [vp, f32, vp, i64, vp, vp, vp]
    L.sbm_comm_destroy.argtypes = [vp]
    L.sbm_match_device_sharded.argtypes = [vp, vp, i32, i32, i32, i32, vp, f32, vp, i64, vp, vp, vp]
    L.sbm_match_batch_device_sharded.argtypes = [vp, vp, i64, i32, i32, i32, i32, i32, vp, f32, vp, i64, vp, vp, vp]
    L.sbm_extract_local_maxima.argtypes = [vp, vp, vp, i32, i32, f32, vp, i64, C.POINTER(i64)]
    L.sbm_set_pipeline_depth.argtypes = [vp, i32]
    L.sbm_set_coarse_mode.argtypes = [vp, i32]
    L.sbm_set_refine_order.argtypes = [vp, i32]
    L.sbm_set_refine_bits.argtypes = [vp, i32]
    L.sbm_select_templates.argtypes = [vp, vp, i32]
    L.sbm_partition_templates.argtypes = [vp, i32, i32, vp, i32, i32, vp, vp]
    L.sbm_match_sharded.argtypes = [vp, i32, vp, i32, i32, i32, i32, vp, f32, vp, i64, C.POINTER(i64)]
    L.sbm_match_batch_host_begin.argtypes = [vp, vp, i32, i32, i32, i32, i32, vp, f32, i64, i32]
    L.sbm_match_batch_host_end.argtypes = [vp, vp, vp]
    L.sbm_match_batch_host.argtypes = [vp, vp, i32, i32, i32, i32, i32, vp, f32, vp, i64, vp, i32]
    L.sbm_pin_host_buffer.argtypes = [vp, vp, i64]
    L.sbm_unpin_host_buffer.argtypes = [vp, vp]
    L.sbm_match_batch_device_banded.argtypes = [vp, vp, i64, i32, i32, i32, i32, i32, vp, f32, vp, i64, vp, vp, i32, vp]
    for name in ABI_SYMBOLS:
        f = getattr(L, name)
        if name not in ("sbm_last_error", "sbm_destroy", "sbm_canonicalize"):
            f.restype = i32
    _lib = L
    return L


def _check(rc: int):
    if rc != 0:
        raise SbmError(rc, lib().sbm_last_error().decode("utf-8", "replace"))


def _p(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _img(img: np.ndarray) -> Tuple[np.ndarray, int, int, int]:
    img = np.ascontiguousarray(img, dtype=np.uint8)
    if img.ndim == 2:
        return img, img.shape[0], img.shape[1], 1
    if img.ndim == 3 and img.shape[2] in (1, 3):
        return img, img.shape[0], img.shape[1], img.shape[2]
    raise ValueError("image must be HxW or HxWx{1,3} uint8")


def canonicalize(recs: np.ndarray) -> np.ndarray:
    recs = np.ascontiguousarray(recs, MATCH_DTYPE).copy()
    n = lib().sbm_canonicalize(_p(recs), len(recs))
    return recs[:n]


class Context:
    """One GPU context (``sbm_ctx``)."""

    def __init__(self, T: Sequence[int] = (4, 8), weak_threshold: float = 30.0, device_id: int = 0,
                 max_candidates: int = 0):
        cfg = SbmConfig()
        cfg.n_levels = len(T)
        for i, t in enumerate(T):
            cfg.T[i] = int(t)
        cfg.weak_threshold = float(weak_threshold)
        cfg.device_id = int(device_id)
        cfg.max_candidates = int(max_candidates)
        self.T = [int(t) for t in T]
        self.n_levels = len(T)
        self._h = C.c_void_p()
        _check(lib().sbm_create(C.byref(cfg), C.byref(self._h)))
        self._cap = max_candidates if max_candidates > 0 else 1 << 20
        self.n_templates = 0

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            lib().sbm_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- templates ------------------------------------------------------------
    def upload_templates(self, ts: TemplateSet):
        if ts.n_levels != self.n_levels:
            raise ValueError(f"template pyramids have {ts.n_levels} levels, detector has {self.n_levels}")
        levels = np.ascontiguousarray(ts.levels, LEVEL_DTYPE)
        feats = np.ascontiguousarray(ts.features, FEATURE_DTYPE)
        ci = np.ascontiguousarray(ts.class_idx, np.int32)
        ti = np.ascontiguousarray(ts.template_id, np.int32)
        _check(lib().sbm_upload_templates(self._h, ts.n_templates, _p(levels), _p(feats), len(feats), _p(ci), _p(ti)))
        self.n_templates = ts.n_templates

    def select_classes(self, class_idx: Sequence[int]):
        a = np.asarray(list(class_idx), np.int32)
        _check(lib().sbm_select_classes(self._h, _p(a) if len(a) else None, len(a)))

    def select_range(self, first: int, count: int):
        _check(lib().sbm_select_range(self._h, first, count))

    # -- whole path -------------------------------------------------------------
    def match(self, img: np.ndarray, threshold: float, mask: Optional[np.ndarray] = None) -> np.ndarray:
        img, r, c, ch = _img(img)
        m = None if mask is None else np.ascontiguousarray(mask, np.uint8)
        out = np.empty(self._cap, MATCH_DTYPE)
        n = C.c_int64(0)
        _check(lib().sbm_match(self._h, _p(img), r, c, c * ch, ch, _p(m), C.c_float(threshold), _p(out), self._cap, C.byref(n)))
        return out[: n.value].copy()

    def match_device(self, d_img: int, rows: int, cols: int, stride: int, channels: int, threshold: float,
                     d_out: int, cap: int, d_count: int, stream: int = 0, d_mask: int = 0):
        _check(lib().sbm_match_device(self._h, C.c_void_p(d_img), rows, cols, stride, channels,
                                      C.c_void_p(d_mask) if d_mask else None, C.c_float(threshold),
                                      C.c_void_p(d_out), cap, C.c_void_p(d_count),
                                      C.c_void_p(stream) if stream else None))

    def match_batch_device(self, d_imgs: int, frame_stride: int, n_frames: int, rows: int, cols: int, stride: int,
                           channels: int, threshold: float, d_out: int, cap: int, d_counts: int, stream: int = 0,
                           d_mask: int = 0):
        """n_frames frames (frame f at d_imgs + f*frame_stride) in one launch of every kernel; frame f's records
        at d_out + f*cap records, its {n, overflow} at d_counts + 2*f int32."""
        _check(lib().sbm_match_batch_device(self._h, C.c_void_p(d_imgs), frame_stride, n_frames, rows, cols, stride,
                                            channels, C.c_void_p(d_mask) if d_mask else None, C.c_float(threshold),
                                            C.c_void_p(d_out), cap, C.c_void_p(d_counts),
                                            C.c_void_p(stream) if stream else None))

    def set_result_mirror(self, mirror_out: int, mirror_count: int):
        """Device-visible (e.g. pinned host) addresses that receive a copy of every result."""
        _check(lib().sbm_set_result_mirror(self._h, C.c_void_p(mirror_out) if mirror_out else None,
                                           C.c_void_p(mirror_count) if mirror_count else None))

    # -- multi-GPU exchange (RCCL inside the library) -------------------------------------
    @staticmethod
    def comm_unique_id() -> bytes:
        buf = C.create_string_buffer(128)
        _check(lib().sbm_comm_unique_id(buf))
        return buf.raw

    def comm_init(self, world: int, rank: int, uid: bytes):
        assert len(uid) == 128
        _check(lib().sbm_comm_init(self._h, world, rank, C.c_char_p(uid)))

    def comm_count(self) -> int:
        """ranks of this context's communicator as RCCL reports them (ncclCommCount)"""
        n = C.c_int32(0)
        _check(lib().sbm_comm_count(self._h, C.byref(n)))
        return int(n.value)

    def match_templates_device_sharded(self, threshold: float, d_local: int, cap: int, d_gathered: int, gathered_mirror: int = 0,
                                       stream: int = 0):
        _check(lib().sbm_match_templates_device_sharded(self._h, C.c_float(threshold), C.c_void_p(d_local), cap, C.c_void_p(d_gathered),
                                                        C.c_void_p(gathered_mirror) if gathered_mirror else None,
                                                        C.c_void_p(stream) if stream else None))

    def match_device_sharded(self, d_img: int, rows: int, cols: int, stride: int, channels: int, threshold: float,
                             d_local: int, cap: int, d_gathered: int, gathered_mirror: int = 0, stream: int = 0,
                             d_mask: int = 0):
        _check(lib().sbm_match_device_sharded(self._h, C.c_void_p(d_img), rows, cols, stride, channels,
                                              C.c_void_p(d_mask) if d_mask else None, C.c_float(threshold),
                                              C.c_void_p(d_local), cap, C.c_void_p(d_gathered),
                                              C.c_void_p(gathered_mirror) if gathered_mirror else None,
                                              C.c_void_p(stream) if stream else None))

    def match_batch_device_sharded(self, d_imgs: int, frame_stride: int, n_frames: int, rows: int, cols: int, stride: int,
                                   channels: int, threshold: float, d_local: int, cap: int, d_gathered: int,
                                   gathered_mirror: int = 0, stream: int = 0, d_mask: int = 0):
        _check(lib().sbm_match_batch_device_sharded(self._h, C.c_void_p(d_imgs), frame_stride, n_frames, rows, cols, stride,
                                                    channels, C.c_void_p(d_mask) if d_mask else None, C.c_float(threshold),
                                                    C.c_void_p(d_local), cap, C.c_void_p(d_gathered),
                                                    C.c_void_p(gathered_mirror) if gathered_mirror else None,
                                                    C.c_void_p(stream) if stream else None))

    def match_batch_device_banded(self, d_imgs: int, frame_stride: int, n_frames: int, rows: int, cols: int, stride: int,
                                  channels: int, threshold: float, d_local: int, cap: int, d_gathered: int = 0,
                                  gathered_mirror: int = 0, n_bands: int = 0, stream: int = 0, d_mask: int = 0):
        """build-sharded step: this rank's row band of every level's orientation map, grouped in-place all-gather of the
        bands, linear memories, template loop, gather of the match lists (one GPU: all n_bands bands, one launch each)"""
        _check(lib().sbm_match_batch_device_banded(self._h, C.c_void_p(d_imgs), frame_stride, n_frames, rows, cols, stride,
                                                   channels, C.c_void_p(d_mask) if d_mask else None, C.c_float(threshold),
                                                   C.c_void_p(d_local), cap, C.c_void_p(d_gathered) if d_gathered else None,
                                                   C.c_void_p(gathered_mirror) if gathered_mirror else None, n_bands,
                                                   C.c_void_p(stream) if stream else None))

    def extract_local_maxima(self, magnitude: np.ndarray, strong_threshold: float, mask: Optional[np.ndarray] = None) -> np.ndarray:
        """[(x, y)] of the 5x5 local maxima extractTemplate accepts (row-major order)"""
        mag = np.ascontiguousarray(magnitude, np.float32)
        m = None if mask is None else np.ascontiguousarray(mask, np.uint8)
        cap = max(16, mag.size // 4)
        xy = np.zeros(cap, np.int32)
        n = C.c_int64(0)
        _check(lib().sbm_extract_local_maxima(self._h, _p(mag), _p(m), mag.shape[0], mag.shape[1], C.c_float(strong_threshold), _p(xy), cap,
                                              C.byref(n)))
        xy = xy[: n.value]
        return np.stack([xy & 0xFFFF, xy >> 16], axis=1)

    def select_templates(self, idx: Sequence[int]):
        a = np.ascontiguousarray(idx, np.int32)
        _check(lib().sbm_select_templates(self._h, _p(a), len(a)))

    def partition_templates(self, rows: int, cols: int, n_shards: int, idx: Optional[Sequence[int]] = None):
        """work-balanced contiguous shards [(first, count)] of the template list ``idx`` (None: all, upload order)"""
        a = None if idx is None else np.ascontiguousarray(idx, np.int32)
        first = np.zeros(n_shards, np.int32)
        count = np.zeros(n_shards, np.int32)
        _check(lib().sbm_partition_templates(self._h, rows, cols, _p(a), 0 if a is None else len(a), n_shards, _p(first), _p(count)))
        return list(zip(first.tolist(), count.tolist()))

    @staticmethod
    def match_sharded(ctxs: Sequence["Context"], img: np.ndarray, threshold: float, mask: Optional[np.ndarray] = None,
                      cap: int = 1 << 16) -> np.ndarray:
        """single-process multi-GPU match: one context (and host thread) per GPU, lists concatenated on the host"""
        img, r, c, ch = _img(img)
        m = None if mask is None else np.ascontiguousarray(mask, np.uint8)
        hs = (C.c_void_p * len(ctxs))(*[x._h for x in ctxs])
        out = np.zeros(cap, MATCH_DTYPE)
        n = C.c_int64(0)
        _check(lib().sbm_match_sharded(hs, len(ctxs), _p(img), r, c, c * ch, ch, _p(m), C.c_float(threshold), _p(out), cap, C.byref(n)))
        return out[: n.value].copy()

    def match_batch_host(self, frames: Sequence[np.ndarray], threshold: float, cap: int = 1024, sub_batch: int = 0,
                         mask: Optional[np.ndarray] = None, split: bool = False):
        """frames in host memory, pipelined uploads; returns a list of record arrays (one per frame).
        split=True: sbm_match_batch_host_begin, then _end (the asynchronous form)"""
        arrs = [_img(f) for f in frames]
        r, c, ch = arrs[0][1:]
        assert all(a[1:] == (r, c, ch) for a in arrs)
        ptrs = (C.c_void_p * len(arrs))(*[a[0].ctypes.data for a in arrs])
        m = None if mask is None else np.ascontiguousarray(mask, np.uint8)
        out = np.zeros((len(arrs), cap), MATCH_DTYPE)
        counts = np.zeros((len(arrs), 2), np.int32)
        if split:
            _check(lib().sbm_match_batch_host_begin(self._h, ptrs, len(arrs), r, c, c * ch, ch, _p(m), C.c_float(threshold), cap, sub_batch))
            _check(lib().sbm_match_batch_host_end(self._h, _p(out), _p(counts)))
        else:
            _check(lib().sbm_match_batch_host(self._h, ptrs, len(arrs), r, c, c * ch, ch, _p(m), C.c_float(threshold), _p(out), cap,
                                              _p(counts), sub_batch))
        return [out[f, : counts[f, 0]].copy() for f in range(len(arrs))]

    def pin_host_buffer(self, a: np.ndarray):
        """explicit opt-in: frames inside ``a`` are uploaded by direct DMA until unpin_host_buffer(a) / close()"""
        _check(lib().sbm_pin_host_buffer(self._h, C.c_void_p(a.ctypes.data), a.nbytes))

    def unpin_host_buffer(self, a: np.ndarray):
        _check(lib().sbm_unpin_host_buffer(self._h, C.c_void_p(a.ctypes.data)))

    def set_quantize_mode(self, mode: str = "auto", rows_per_wave: int = 0):
        """gradient kernel choice: "auto" (by launch size), "tile" or "stream"; bit-identical results"""
        _check(lib().sbm_set_quantize_mode(self._h, {"auto": 0, "tile": 1, "stream": 2}[mode], rows_per_wave))

    def set_coarse_mode(self, mode: str = "auto"):
        """coarse-pass kernel: "auto" (bit planes whenever the threshold is >= 0), "block" (bytes, four waves per item), "wave"
        (bytes, one wave per item), "bits", "bytes" (block or wave by launch size); identical candidates"""
        _check(lib().sbm_set_coarse_mode(self._h, {"auto": 0, "": 0, "block": 1, "wave": 2, "bits": 3, "bytes": 4}[mode]))

    def set_refine_bits(self, mode=None):
        """None: the process default; True: T = 4 refinement levels as bit strips; False: as spread bytes (sbm.h)."""
        _check(lib().sbm_set_refine_bits(self._h, -1 if mode is None else (1 if mode else 0)))

    def set_refine_order(self, order: str = "auto"):
        """refinement pass: "auto" (by the batch's plane size), "slots" (per-frame slots) or "list" (one frame-major list)"""
        _check(lib().sbm_set_refine_order(self._h, {"auto": -1, "slots": 0, "list": 2}[order]))

    def set_pipeline_depth(self, batches_in_flight: int):
        """hint: the caller keeps this many batches in flight on the GPU -> launches are sized for throughput (>= 2)"""
        _check(lib().sbm_set_pipeline_depth(self._h, batches_in_flight))

    def graph_count(self) -> int:
        n = C.c_int32(0)
        _check(lib().sbm_graph_count(self._h, C.byref(n)))
        return int(n.value)

    def set_graph_mode(self, on):
        """True: always replay captured hipGraphs; False: never; None / "auto" (the library's default): batch and template-loop
        entry points replay once several calls are in flight (set_pipeline_depth >= 2) and an argument tuple repeats"""
        _check(lib().sbm_set_graph_mode(self._h, -1 if on is None or on == "auto" else (1 if on else 0)))

    # -- pyramid state ------------------------------------------------------------
    def build_pyramid(self, img: np.ndarray, mask: Optional[np.ndarray] = None):
        img, r, c, ch = _img(img)
        m = None if mask is None else np.ascontiguousarray(mask, np.uint8)
        _check(lib().sbm_build_pyramid(self._h, _p(img), r, c, c * ch, ch, _p(m)))

    def set_quantized(self, level: int, q: np.ndarray):
        q = np.ascontiguousarray(q, np.uint8)
        _check(lib().sbm_set_quantized(self._h, level, _p(q), q.shape[0], q.shape[1]))

    def level_dims(self, level: int) -> Tuple[int, int]:
        r, c = C.c_int32(), C.c_int32()
        _check(lib().sbm_level_dims(self._h, level, C.byref(r), C.byref(c)))
        return r.value, c.value

    def get_quantized(self, level: int) -> np.ndarray:
        r, c = self.level_dims(level)
        out = np.empty((r, c), np.uint8)
        _check(lib().sbm_get_quantized(self._h, level, _p(out)))
        return out

    def get_linear_memories(self, level: int) -> np.ndarray:
        s = C.c_int64()
        _check(lib().sbm_get_linear_memories(self._h, level, None, 0, C.byref(s)))
        out = np.empty((8, s.value), np.uint8)
        _check(lib().sbm_get_linear_memories(self._h, level, _p(out), out.nbytes, C.byref(s)))
        return out

    def get_coarse_bitplanes(self, frame: int = 0) -> np.ndarray:
        """[16][lm_stride / 8] bytes: the coarsest level's "response > 0" (0..7) and "response == 4" (8..15) bit planes"""
        s = C.c_int64()
        _check(lib().sbm_get_linear_memories(self._h, self.n_levels - 1, None, 0, C.byref(s)))
        out = np.empty((16, s.value // 8), np.uint8)
        _check(lib().sbm_get_coarse_bitplanes(self._h, frame, _p(out), out.nbytes))
        return out

    def match_templates(self, threshold: float) -> np.ndarray:
        out = np.empty(self._cap, MATCH_DTYPE)
        n = C.c_int64(0)
        _check(lib().sbm_match_templates(self._h, C.c_float(threshold), _p(out), self._cap, C.byref(n)))
        return out[: n.value].copy()

    def match_templates_device(self, threshold: float, d_out: int, cap: int, d_count: int, stream: int = 0):
        _check(lib().sbm_match_templates_device(self._h, C.c_float(threshold), C.c_void_p(d_out), cap, C.c_void_p(d_count),
                                                C.c_void_p(stream) if stream else None))

    # -- single reference functions -------------------------------------------
    def quantized_orientations(self, img: np.ndarray, weak: float, want_float: bool = True):
        img, r, c, ch = _img(img)
        ang = np.empty((r, c), np.uint8)
        mag = np.empty((r, c), np.float32) if want_float else None
        ori = np.empty((r, c), np.float32) if want_float else None
        _check(lib().sbm_quantized_orientations(self._h, _p(img), r, c, c * ch, ch, C.c_float(weak), _p(mag), _p(ang), _p(ori)))
        return mag, ang, ori

    def orientation_bins(self, gx: np.ndarray, gy: np.ndarray) -> np.ndarray:
        gx = np.ascontiguousarray(gx, np.int16)
        gy = np.ascontiguousarray(gy, np.int16)
        out = np.empty(gx.shape, np.uint8)
        _check(lib().sbm_orientation_bins(self._h, _p(gx), _p(gy), gx.size, _p(out)))
        return out

    def pyrdown(self, img: np.ndarray) -> np.ndarray:
        img, r, c, ch = _img(img)
        shape = (r // 2, c // 2) if img.ndim == 2 else (r // 2, c // 2, ch)
        out = np.empty(shape, np.uint8)
        _check(lib().sbm_pyrdown(self._h, _p(img), r, c, c * ch, ch, _p(out)))
        return out

    def resize_linear(self, img: np.ndarray, fx: float, fy: float) -> np.ndarray:
        """cv::resize(img, Size(), fx, fy, INTER_LINEAR) of an 8-bit image (shapeInfo_producer::transform)"""
        a, r, c, ch = _img(img)
        dr, dc = C.c_int32(0), C.c_int32(0)
        _check(lib().sbm_resize_linear(self._h, _p(a), r, c, c * ch, ch, fx, fy, None, 0, C.byref(dr), C.byref(dc)))
        out = np.empty((dr.value, dc.value) if a.ndim == 2 else (dr.value, dc.value, ch), np.uint8)
        _check(lib().sbm_resize_linear(self._h, _p(a), r, c, c * ch, ch, fx, fy, _p(out), out.size, C.byref(dr), C.byref(dc)))
        return out

    def spread(self, q: np.ndarray, T: int) -> np.ndarray:
        q = np.ascontiguousarray(q, np.uint8)
        out = np.empty_like(q)
        _check(lib().sbm_spread(self._h, _p(q), q.shape[0], q.shape[1], T, _p(out)))
        return out

    def compute_response_maps(self, sp: np.ndarray) -> np.ndarray:
        sp = np.ascontiguousarray(sp, np.uint8)
        out = np.empty((8,) + sp.shape, np.uint8)
        _check(lib().sbm_compute_response_maps(self._h, _p(sp), sp.shape[0], sp.shape[1], _p(out)))
        return out

    def linearize(self, m: np.ndarray, T: int) -> np.ndarray:
        m = np.ascontiguousarray(m, np.uint8)
        r, c = m.shape
        out = np.empty((T * T, (r // T) * (c // T)), np.uint8)
        _check(lib().sbm_linearize(self._h, _p(m), r, c, T, _p(out)))
        return out

    def similarity(self, template_index: int) -> np.ndarray:
        lc = self.n_levels - 1
        r, c = self.level_dims(lc)
        T = self.T[lc]
        out = np.empty((r // T, c // T), np.uint16)
        _check(lib().sbm_similarity(self._h, template_index, _p(out)))
        return out

    def similarity_local(self, level: int, template_index: int, cx: int, cy: int) -> np.ndarray:
        out = np.empty((16, 16), np.uint16)
        _check(lib().sbm_similarity_local(self._h, level, template_index, cx, cy, _p(out)))
        return out

    # -- measurement -----------------------------------------------------------
    def set_profiling(self, on: bool, accumulate: bool = False):
        """accumulate: timings of successive asynchronous calls pile up until timings() reads them"""
        _check(lib().sbm_set_profiling(self._h, (2 if accumulate else 1) if on else 0))

    def timings(self) -> List[Tuple[str, float]]:
        n = C.c_int32(0)
        _check(lib().sbm_get_timings(self._h, None, None, 0, C.byref(n)))
        names = (C.c_char_p * max(n.value, 1))()
        ms = (C.c_float * max(n.value, 1))()
        _check(lib().sbm_get_timings(self._h, names, ms, n.value, C.byref(n)))
        return [(names[i].decode(), float(ms[i])) for i in range(n.value)]

    def stats(self) -> Tuple[int, int]:
        """(coarse candidates, refinement bytes) of the last template-matching call."""
        a, b = C.c_int64(0), C.c_int64(0)
        _check(lib().sbm_get_stats(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def coarse_bytes(self) -> int:
        b = C.c_int64(0)
        _check(lib().sbm_coarse_bytes(self._h, C.byref(b)))
        return b.value
