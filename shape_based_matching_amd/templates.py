"""Template containers and (de)serialisation for the LINE-2D matcher.

Mirrors the reference's data model (``line2Dup.h:116-153, 319-320``):
``Feature`` / ``Template`` / ``TemplatePyramid`` / ``TemplatesMap`` flattened into
numpy structured arrays whose layouts are exactly ``sbm_feature`` and
``sbm_template_level`` of ``include/sbm_types.h``.

The YAML reader/writer follows the OpenCV ``FileStorage`` layout the reference
produces in ``Detector::writeClass`` / ``Template::write`` / ``Feature::write``
(``line2Dup.cpp:42-113, 1507-1575``), including the fork-only optional keys
(``scale``, ``orientation``, ``tagFieldID``, ``fiducial_src``) which default to
0 / "" when absent, as ``cv::FileNode`` does.
"""
from __future__ import annotations

import gzip
import re
from dataclasses import dataclass, field
from typing import Dict, List, Sequence

import numpy as np

FEATURE_DTYPE = np.dtype([("x", "<i4"), ("y", "<i4"), ("label", "<i4")])
LEVEL_DTYPE = np.dtype(
    [
        ("width", "<i4"),
        ("height", "<i4"),
        ("tl_x", "<i4"),
        ("tl_y", "<i4"),
        ("pyramid_level", "<i4"),
        ("n_features", "<i4"),
        ("feature_offset", "<i8"),
    ]
)
MATCH_DTYPE = np.dtype(
    [
        ("x", "<i4"),
        ("y", "<i4"),
        ("similarity", "<f4"),
        ("raw", "<i4"),
        ("class_idx", "<i4"),
        ("template_id", "<i4"),
    ]
)
assert FEATURE_DTYPE.itemsize == 12 and LEVEL_DTYPE.itemsize == 32 and MATCH_DTYPE.itemsize == 24


@dataclass
class TemplateSet:
    """A flat list of template pyramids, possibly from several classes.

    ``levels`` has shape ``[n_templates, n_levels]``; ``class_idx[t]`` indexes
    ``class_ids``; ``template_id[t]`` is the index of pyramid ``t`` inside its
    class (the reference's template_id, ``line2Dup.cpp:1312``).
    """

    n_levels: int
    levels: np.ndarray
    features: np.ndarray
    class_idx: np.ndarray
    template_id: np.ndarray
    class_ids: List[str] = field(default_factory=list)
    # fork-only per-level metadata (line2Dup.h:140-148); kept for round-tripping
    meta: Dict[str, np.ndarray] = field(default_factory=dict)

    @property
    def n_templates(self) -> int:
        return int(self.levels.shape[0])

    def feats_of(self, t: int, level: int) -> np.ndarray:
        lv = self.levels[t, level]
        o = int(lv["feature_offset"])
        return self.features[o : o + int(lv["n_features"])]

    def subset(self, idx: Sequence[int]) -> "TemplateSet":
        """Re-pack the pyramids listed in ``idx`` (keeps ids and classes)."""
        idx = np.asarray(idx, dtype=np.int64)
        levels = self.levels[idx].copy()
        chunks = []
        off = 0
        for i in range(levels.shape[0]):
            for l in range(self.n_levels):
                o = int(levels[i, l]["feature_offset"])
                n = int(levels[i, l]["n_features"])
                chunks.append(self.features[o : o + n])
                levels[i, l]["feature_offset"] = off
                off += n
        feats = np.concatenate(chunks) if chunks else np.zeros(0, FEATURE_DTYPE)
        return TemplateSet(
            self.n_levels, levels, feats, self.class_idx[idx].copy(), self.template_id[idx].copy(), list(self.class_ids)
        )

    @staticmethod
    def concat(sets: Sequence["TemplateSet"]) -> "TemplateSet":
        assert sets and all(s.n_levels == sets[0].n_levels for s in sets)
        class_ids: List[str] = []
        levels, feats, cidx, tid = [], [], [], []
        off = 0
        for s in sets:
            remap = []
            for c in s.class_ids:
                if c not in class_ids:
                    class_ids.append(c)
                remap.append(class_ids.index(c))
            lv = s.levels.copy()
            lv["feature_offset"] += off
            off += len(s.features)
            levels.append(lv)
            feats.append(s.features)
            cidx.append(np.asarray(remap, np.int32)[s.class_idx] if len(s.class_idx) else s.class_idx)
            tid.append(s.template_id)
        return TemplateSet(
            sets[0].n_levels,
            np.concatenate(levels),
            np.concatenate(feats),
            np.concatenate(cidx).astype(np.int32),
            np.concatenate(tid).astype(np.int32),
            class_ids,
        )

    # -- compact binary form (used for committed fixtures) ------------------
    def save_npz(self, path: str) -> None:
        np.savez_compressed(
            path,
            n_levels=np.int32(self.n_levels),
            levels=self.levels,
            fx=self.features["x"].astype(np.int16),
            fy=self.features["y"].astype(np.int16),
            fl=self.features["label"].astype(np.uint8),
            class_idx=self.class_idx,
            template_id=self.template_id,
            class_ids=np.array(self.class_ids),
        )

    @staticmethod
    def load_npz(path: str) -> "TemplateSet":
        z = np.load(path, allow_pickle=False)
        feats = np.zeros(len(z["fx"]), FEATURE_DTYPE)
        feats["x"], feats["y"], feats["label"] = z["fx"], z["fy"], z["fl"]
        return TemplateSet(
            int(z["n_levels"]),
            z["levels"].astype(LEVEL_DTYPE),
            feats,
            z["class_idx"].astype(np.int32),
            z["template_id"].astype(np.int32),
            [str(s) for s in z["class_ids"]],
        )


def from_pyramids(pyramids: Sequence[Sequence[dict]], class_id: str = "test") -> TemplateSet:
    """Build a TemplateSet from ``[[{width,height,tl_x,tl_y,features:[(x,y,l)..]}, ..levels], ..]``."""
    n_levels = len(pyramids[0]) if pyramids else 0
    levels = np.zeros((len(pyramids), n_levels), LEVEL_DTYPE)
    feats = []
    off = 0
    for t, tp in enumerate(pyramids):
        assert len(tp) == n_levels
        for l, tm in enumerate(tp):
            f = np.asarray(tm["features"], dtype=np.int32).reshape(-1, 3)
            lv = levels[t, l]
            lv["width"], lv["height"] = tm["width"], tm["height"]
            lv["tl_x"], lv["tl_y"] = tm.get("tl_x", 0), tm.get("tl_y", 0)
            lv["pyramid_level"] = tm.get("pyramid_level", l)
            lv["n_features"] = len(f)
            lv["feature_offset"] = off
            off += len(f)
            feats.append(f)
    flat = np.zeros(off, FEATURE_DTYPE)
    if off:
        allf = np.concatenate(feats)
        flat["x"], flat["y"], flat["label"] = allf[:, 0], allf[:, 1], allf[:, 2]
    n = len(pyramids)
    return TemplateSet(n_levels, levels, flat, np.zeros(n, np.int32), np.arange(n, dtype=np.int32), [class_id])


# ---------------------------------------------------------------------------
# OpenCV FileStorage YAML subset
# ---------------------------------------------------------------------------
_FEAT_RE = re.compile(r"^\s*-\s*\[\s*(-?\d+)\s*,\s*(-?\d+)\s*,\s*(-?\d+)\s*\]\s*$")
_KV_RE = re.compile(r"^\s*([A-Za-z_][A-Za-z0-9_]*)\s*:\s*(.*?)\s*$")


def _open_text(path: str):
    if path.endswith(".gz"):
        return gzip.open(path, "rt")
    return open(path, "rt")


def read_class_yaml(path: str, class_id_override: str = "") -> TemplateSet:
    """``Detector::readClass`` (``line2Dup.cpp:1507-1546``) for one class file."""
    class_id = ""
    n_levels = 0
    pyramids: List[List[dict]] = []
    metas: List[List[dict]] = []
    cur_tm = None
    with _open_text(path) as fh:
        for line in fh:
            m = _FEAT_RE.match(line)
            if m:
                cur_tm["features"].append((int(m.group(1)), int(m.group(2)), int(m.group(3))))
                continue
            m = _KV_RE.match(line)
            if not m:
                continue
            k, v = m.group(1), m.group(2)
            if k == "class_id":
                class_id = v.strip('"')
            elif k == "pyramid_levels":
                n_levels = int(v)
            elif k == "template_id":
                if int(v) != len(pyramids):  # CV_Assert(template_id == expected_id) :1532
                    raise ValueError(f"{path}: template_id {v} != expected {len(pyramids)}")
                pyramids.append([])
                metas.append([])
            elif k == "width":
                cur_tm = {"width": int(v), "features": []}
                cur_meta = {"scale": 0.0, "orientation": 0.0, "tagFieldID": 0, "fiducial_src": ""}
                pyramids[-1].append(cur_tm)
                metas[-1].append(cur_meta)
            elif k in ("height", "tl_x", "tl_y", "pyramid_level"):
                cur_tm[k] = int(v)
            elif k in ("scale", "orientation"):
                cur_meta[k] = float(v)
            elif k == "tagFieldID":
                cur_meta[k] = int(v)
            elif k == "fiducial_src":
                cur_meta[k] = v.strip('"')
    if not pyramids:
        raise ValueError(f"{path}: no template pyramids")
    ts = from_pyramids(pyramids, class_id_override or class_id)
    if n_levels and ts.n_levels != n_levels:
        raise ValueError(f"{path}: pyramid_levels {n_levels} != {ts.n_levels} templates per pyramid")
    ts.meta = {
        "scale": np.array([[m["scale"] for m in mm] for mm in metas], np.float32),
        "orientation": np.array([[m["orientation"] for m in mm] for mm in metas], np.float32),
        "tagFieldID": np.array([[m["tagFieldID"] for m in mm] for mm in metas], np.int32),
    }
    return ts


def read_classes(class_ids: Sequence[str], fmt: str) -> TemplateSet:
    """``Detector::readClasses`` (``line2Dup.cpp:1577-1587``): ``fmt % class_id`` per id."""
    return TemplateSet.concat([read_class_yaml(fmt % cid) for cid in class_ids])


def write_class_yaml(ts: TemplateSet, path: str, class_idx: int = 0) -> None:
    """``Detector::writeClass`` (``line2Dup.cpp:1548-1575``) layout."""
    sel = np.nonzero(ts.class_idx == class_idx)[0]
    out = ["%YAML:1.0", "---", f"class_id: {ts.class_ids[class_idx]}", f"pyramid_levels: {ts.n_levels}", "template_pyramids:"]
    for t in sel:
        out += ["   -", f"      template_id: {int(ts.template_id[t])}", "      templates:"]
        for l in range(ts.n_levels):
            lv = ts.levels[t, l]
            out += [
                "         -",
                f"            width: {int(lv['width'])}",
                f"            height: {int(lv['height'])}",
                f"            tl_x: {int(lv['tl_x'])}",
                f"            tl_y: {int(lv['tl_y'])}",
                f"            pyramid_level: {int(lv['pyramid_level'])}",
                "            features:",
            ]
            for f in ts.feats_of(int(t), l):
                out.append(f"               - [ {int(f['x'])}, {int(f['y'])}, {int(f['label'])} ]")
    opener = gzip.open if path.endswith(".gz") else open
    with opener(path, "wt") as fh:
        fh.write("\n".join(out) + "\n")
