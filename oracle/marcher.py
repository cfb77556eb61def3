"""ORACLE (test infrastructure only): ctypes front-end of oracle/nerfacc_grid.c.

Mirrors the call surface of the reference's nerfacc Python layer so the parity tests
read like the reference's own tests:

  ray_aabb_intersect  <- perception/nerfacc/nerfacc/grid.py:13-51
  traverse_grids      <- perception/nerfacc/nerfacc/grid.py:93-192 (+ host logic of
                         cuda/csrc/grid.cu:320-474 and include/data_spec.hpp:86-106:
                         two-pass count/fill, or one over-allocated pass)
  exclusive_sum       <- perception/nerfacc/nerfacc/scan.py:57-97 (packed branch)
  pack_info           <- perception/nerfacc/nerfacc/pack.py:10-49

All arrays are numpy, CPU.
"""
import ctypes
import os
import subprocess
from dataclasses import dataclass
from typing import Optional, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "liboracle.so")
_LIB_FMAD_PATH = os.path.join(_HERE, "_build", "liboracle_fmad.so")
_lib = None
_lib_fmad = None


class _Segments(ctypes.Structure):
    _fields_ = [
        ("vals", ctypes.c_void_p),
        ("ray_indices", ctypes.c_void_p),
        ("is_left", ctypes.c_void_p),
        ("is_right", ctypes.c_void_p),
        ("is_valid", ctypes.c_void_p),
        ("chunk_starts", ctypes.c_void_p),
        ("chunk_cnts", ctypes.c_void_p),
    ]


def build(force: bool = False) -> str:
    """Compile oracle/nerfacc_grid.c -> oracle/_build/liboracle.so (gcc, seconds)."""
    src = os.path.join(_HERE, "nerfacc_grid.c")
    if force or any(not os.path.exists(p) or os.path.getmtime(p) < os.path.getmtime(src) for p in (_LIB_PATH, _LIB_FMAD_PATH)):
        subprocess.check_call(["make", "-s", "-C", _HERE, "all"])
    return _LIB_PATH


def _load(path):
    build()
    l = ctypes.CDLL(path)
    l.orc_ray_aabb_intersect.restype = None
    l.orc_traverse_grids.restype = None
    l.orc_exclusive_sum.restype = None
    return l


def lib(fmad: bool = False):
    """The restatement built with FMA contraction off (the parity reference), or — `fmad=True` — the variant in which every
    product feeding an addition is fused as nvcc's default -fmad=true may do (an exposure estimate, never the reference)."""
    global _lib, _lib_fmad
    if fmad:
        if _lib_fmad is None:
            _lib_fmad = _load(_LIB_FMAD_PATH)
        return _lib_fmad
    if _lib is None:
        _lib = _load(_LIB_PATH)
    return _lib


def _p(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def ray_aabb_intersect(rays_o, rays_d, aabbs, near_plane=-np.inf, far_plane=np.inf, miss_value=np.inf):
    rays_o, rays_d, aabbs = _f32(rays_o), _f32(rays_d), _f32(aabbs)
    n, m = rays_o.shape[0], aabbs.shape[0]
    t_mins = np.empty((n, m), np.float32)
    t_maxs = np.empty((n, m), np.float32)
    hits = np.empty((n, m), np.uint8)
    lib().orc_ray_aabb_intersect(
        ctypes.c_int32(n), _p(rays_o), _p(rays_d), ctypes.c_float(near_plane), ctypes.c_float(far_plane),
        ctypes.c_int32(m), _p(aabbs), ctypes.c_float(miss_value), _p(t_mins), _p(t_maxs), _p(hits))
    return t_mins, t_maxs, hits.astype(bool)


@dataclass
class RayIntervals:
    """data_specs.py:87-180"""
    vals: np.ndarray
    packed_info: np.ndarray
    ray_indices: np.ndarray
    is_left: np.ndarray
    is_right: np.ndarray


@dataclass
class RaySamples:
    """data_specs.py:12-84"""
    vals: np.ndarray
    packed_info: np.ndarray
    ray_indices: np.ndarray
    is_valid: Optional[np.ndarray]


def _alloc(cnts: np.ndarray, masks: bool, valid: bool):
    cumsum = np.cumsum(cnts, dtype=np.int64)
    n_edges = int(cumsum[-1]) if len(cumsum) else 0
    starts = cumsum - cnts
    vals = np.zeros(n_edges, np.float32)
    ridx = np.zeros(n_edges, np.int64)
    left = np.zeros(n_edges, np.uint8) if masks else None
    right = np.zeros(n_edges, np.uint8) if masks else None
    isvalid = np.zeros(n_edges, np.uint8) if valid else None
    return starts, vals, ridx, left, right, isvalid


def traverse_grids(rays_o, rays_d, binaries, aabbs, near_planes=None, far_planes=None,
                   step_size=1e-3, cone_angle=0.0, traverse_steps_limit=None, over_allocate=False,
                   rays_mask=None, t_sorted=None, t_indices=None, hits=None, fmad=False
                   ) -> Tuple[RayIntervals, RaySamples, np.ndarray]:
    rays_o, rays_d, aabbs = _f32(rays_o), _f32(rays_d), _f32(aabbs)
    binaries = np.ascontiguousarray(binaries).astype(np.uint8)
    n = rays_o.shape[0]
    n_grids = binaries.shape[0]
    res = np.asarray(binaries.shape[1:], np.int32)
    near_planes = np.zeros(n, np.float32) if near_planes is None else _f32(near_planes)
    far_planes = np.full(n, np.inf, np.float32) if far_planes is None else _f32(far_planes)
    mask = np.ones(n, np.uint8) if rays_mask is None else np.ascontiguousarray(rays_mask).astype(np.uint8)
    limit = -1 if traverse_steps_limit is None else int(traverse_steps_limit)
    if over_allocate:
        assert limit > 0, "traverse_steps_limit must be set if over_allocate is True."
    if t_sorted is None or t_indices is None or hits is None:
        t_mins, t_maxs, hits = ray_aabb_intersect(rays_o, rays_d, aabbs)
        cat = np.concatenate([t_mins, t_maxs], -1)
        t_indices = np.argsort(cat, axis=-1, kind="stable").astype(np.int64)
        t_sorted = np.take_along_axis(cat, t_indices, -1)
    t_sorted = _f32(t_sorted)
    t_indices = np.ascontiguousarray(t_indices, dtype=np.int64)
    hits_u8 = np.ascontiguousarray(hits).astype(np.uint8)
    term = np.empty(n, np.float32)

    def call(first_pass, iv, sm, use_mask, term_arr):
        lib(fmad).orc_traverse_grids(
            ctypes.c_int32(n), _p(rays_o), _p(rays_d), _p(mask) if use_mask else None,
            ctypes.c_int32(n_grids), _p(res), _p(binaries), _p(aabbs),
            _p(hits_u8), _p(t_sorted), _p(t_indices), _p(near_planes), _p(far_planes),
            ctypes.c_float(step_size), ctypes.c_float(cone_angle), ctypes.c_int32(limit),
            ctypes.c_int32(1 if first_pass else 0), ctypes.byref(iv), ctypes.byref(sm), _p(term_arr))

    if over_allocate:
        # grid.cu:364-404: chunk_cnts = limit * mask, one pass, then starts recomputed from actual cnts
        iv_cnts = (np.full(n, limit * 2, np.int64) * mask).astype(np.int64)
        sm_cnts = (np.full(n, limit, np.int64) * mask).astype(np.int64)
        iv_st, iv_vals, iv_r, iv_l, iv_rt, _ = _alloc(iv_cnts, True, False)
        sm_st, sm_vals, sm_r, _, _, sm_valid = _alloc(sm_cnts, False, True)
        iv = _Segments(_p(iv_vals), _p(iv_r), _p(iv_l), _p(iv_rt), None, _p(iv_st), _p(iv_cnts))
        sm = _Segments(_p(sm_vals), _p(sm_r), None, None, _p(sm_valid), _p(sm_st), _p(sm_cnts))
        # NB: the reference leaves terminate_planes uninitialised for masked-out rays; the oracle
        # pre-fills with the incoming near plane so the value is defined.
        term[:] = near_planes
        call(False, iv, sm, True, term)
        iv_st = np.cumsum(iv_cnts) - iv_cnts
        sm_st = np.cumsum(sm_cnts) - sm_cnts
    else:
        iv_cnts = np.empty(n, np.int64)
        sm_cnts = np.empty(n, np.int64)
        iv = _Segments(None, None, None, None, None, None, _p(iv_cnts))
        sm = _Segments(None, None, None, None, None, None, _p(sm_cnts))
        call(True, iv, sm, False, None)
        iv_st, iv_vals, iv_r, iv_l, iv_rt, _ = _alloc(iv_cnts, True, False)
        sm_st, sm_vals, sm_r, _, _, sm_valid = _alloc(sm_cnts, False, False)
        iv = _Segments(_p(iv_vals), _p(iv_r), _p(iv_l), _p(iv_rt), None, _p(iv_st), _p(iv_cnts))
        sm = _Segments(_p(sm_vals), _p(sm_r), None, None, None, _p(sm_st), _p(sm_cnts))
        call(False, iv, sm, False, term)

    intervals = RayIntervals(iv_vals, np.stack([iv_st, iv_cnts], -1), iv_r, iv_l.astype(bool), iv_rt.astype(bool))
    samples = RaySamples(sm_vals, np.stack([sm_st, sm_cnts], -1), sm_r,
                         None if sm_valid is None else sm_valid.astype(bool))
    return intervals, samples, term


def exclusive_sum(inputs, packed_info, backward=False):
    inputs = _f32(inputs)
    starts = np.ascontiguousarray(packed_info[:, 0], dtype=np.int64)
    cnts = np.ascontiguousarray(packed_info[:, 1], dtype=np.int64)
    out = np.zeros_like(inputs)
    lib().orc_exclusive_sum(ctypes.c_int32(len(starts)), _p(starts), _p(cnts), _p(inputs), _p(out),
                            ctypes.c_int32(1 if backward else 0))
    return out


def pack_info(ray_indices, n_rays=None):
    ray_indices = np.asarray(ray_indices, np.int64)
    if n_rays is None:
        n_rays = int(ray_indices.max()) + 1
    cnts = np.bincount(ray_indices, minlength=n_rays).astype(np.int64)
    starts = np.cumsum(cnts) - cnts
    return np.stack([starts, cnts], -1)
