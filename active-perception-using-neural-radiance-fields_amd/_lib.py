"""ctypes binding of libmi355nerf.so (include/mi355nerf.h).  PyTorch is used only to own device
memory and streams; tensors cross the boundary as raw device pointers."""
import ctypes
import os
from ctypes import POINTER, c_char_p, c_double, c_float, c_int32, c_int64, c_uint32, c_uint64, c_void_p

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_FILE = os.environ.get("MNF_LIB_PATH") or os.path.join(_HERE, "libmi355nerf.so")     # MNF_LIB_PATH: experiment builds (tools/)
_lib = None


class MnfError(RuntimeError):
    """Raised for any non-zero return code of the C ABI (mirrors TORCH_CHECK -> RuntimeError)."""


class _Sized(ctypes.Structure):
    """include/mi355nerf.h MNF_INIT: zero-filled (ctypes does that) with `struct_size` = sizeof(the struct)."""

    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        self.struct_size = ctypes.sizeof(self)


class FieldConfig(_Sized):
    _fields_ = [("struct_size", c_uint32), ("aabb", c_float * 6), ("neurons", c_int32), ("layers", c_int32),
                ("num_semantic_classes", c_int32), ("n_levels", c_int32), ("n_features", c_int32),
                ("log2_hashmap_size", c_int32), ("base_resolution", c_int32), ("max_resolution", c_int32),
                ("output_fp16", c_int32), ("mfma_bf16", c_int32), ("blend_fp16", c_int32)]


class TrainOpts(_Sized):
    _fields_ = [("struct_size", c_uint32), ("near_plane", c_float), ("far_plane", c_float), ("render_step_size", c_float), ("cone_angle", c_float),
                ("alpha_thre", c_float), ("early_stop_eps", c_float), ("render_bkgd", c_float * 3), ("loss_scale", c_float),
                ("stratified", c_int32), ("seed", c_uint64), ("render_bkgd_dev", c_void_p), ("deterministic", c_int32), ("n_levels", c_int32),
                ("presampled", c_void_p)]


class VanillaConfig(_Sized):
    _fields_ = [("struct_size", c_uint32), ("net_depth", c_int32), ("net_width", c_int32), ("skip_layer", c_int32), ("net_depth_condition", c_int32),
                ("net_width_condition", c_int32)]


class RenderOpts(_Sized):
    _fields_ = [("struct_size", c_uint32), ("near_plane", c_float), ("far_plane", c_float), ("render_step_size", c_float),
                ("cone_angle", c_float), ("alpha_thre", c_float), ("early_stop_eps", c_float),
                ("render_bkgd", c_float * 3), ("max_samples", c_int32), ("probabilistic", c_int32),
                ("rays_per_view", c_int32), ("sync_every", c_int32), ("view_order", c_void_p), ("bitgrid", c_void_p), ("n_levels", c_int32)]


class RenderJob(_Sized):
    _fields_ = [("struct_size", c_uint32), ("field", c_void_p), ("binaries", c_void_p), ("bitgrid", c_void_p), ("rays_o", c_void_p), ("rays_d", c_void_p),
                ("n_rays", c_int64), ("rgb", c_void_p), ("acc", c_void_p), ("depth", c_void_p), ("sem", c_void_p), ("rgb_var", c_void_p),
                ("depth_var", c_void_p), ("total_samples", c_void_p), ("workspace", c_void_p), ("workspace_bytes", c_int64)]


# name -> (restype, argtypes); every symbol declared in include/mi355nerf.h
SIGNATURES = {
    "mnf_last_error": (c_char_p, []),
    "mnf_version": (c_int32, []),
    "mnf_device_count": (c_int32, []),
    "mnf_ray_aabb_intersect": (c_int32, [c_void_p, c_void_p, c_int32, c_void_p, c_int32, c_float, c_float, c_float,
                                         c_void_p, c_void_p, c_void_p, c_void_p]),
    "mnf_traverse_grids": (c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_int32,
                                     c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                     c_float, c_float, c_int32, c_int32,
                                     c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                     c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "mnf_sample_rays": (c_int32, [c_void_p, c_void_p, c_int32, c_void_p, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p,
                                  c_float, c_float, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "mnf_sample_rays_levels": (c_int32, [c_void_p, c_void_p, c_int32, c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p,
                                         c_float, c_float, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "mnf_compact_samples": (c_int32, [c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_void_p]),
    "mnf_exclusive_sum": (c_int32, [c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_int64, c_int32, c_void_p]),
    "mnf_render_weight_from_density": (c_int32, [c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_void_p,
                                                 c_int64, c_void_p, c_void_p, c_void_p, c_void_p]),
    "mnf_run_bounds": (c_int32, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p]),
    "mnf_visible_samples": (c_int32, [c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_void_p,
                                      c_void_p, c_void_p, c_void_p, c_void_p]),
    "mnf_composite_train_forward": (c_int32, [c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                              c_int32, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                              c_void_p, c_void_p, c_void_p]),
    "mnf_composite_train_backward": (c_int32, [c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                               c_int32, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                               c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "mnf_adam_step": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_float, c_float, c_float, c_float, c_int32, c_void_p]),
    "mnf_adam_step_guarded": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_float, c_float, c_float, c_float, c_void_p, c_void_p,
                                        c_void_p, c_void_p, c_int64, c_void_p]),
    "mnf_field_optimizer_step": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_float, c_float, c_float, c_void_p,
                                           c_int32, c_void_p, c_void_p]),
    "mnf_field_optimizer_step_report": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_float, c_float, c_float, c_void_p,
                                           c_int32, c_void_p, c_void_p, c_void_p, c_void_p]),
    "mnf_count_nan": (c_int32, [c_void_p, c_int64, c_void_p, c_void_p]),
    "mnf_scan_workspace_bytes": (c_int64, [c_int64]),
    "mnf_pack_info": (c_int32, [c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_int64, c_void_p]),
    "mnf_exclusive_scan_i64": (c_int32, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "mnf_accumulate_along_rays": (c_int32, [c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_void_p, c_void_p]),
    "mnf_accumulate_along_rays_backward": (c_int32, [c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_void_p, c_void_p, c_void_p, c_void_p]),
    "mnf_occ_workspace_bytes": (c_int64, [c_int64, c_int32]),
    "mnf_occ_list_capacity": (c_int64, [c_int64, c_int32, c_int32]),
    "mnf_pack_bitgrid": (c_int32, [c_void_p, c_int64, c_int32, c_void_p, c_void_p]),
    "mnf_occ_sample_cells": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_void_p, c_int32, c_int32, c_uint64, c_void_p, c_void_p,
                                       c_int64, c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_void_p]),
    "mnf_occ_apply": (c_int32, [c_void_p, c_void_p, c_void_p, c_float, c_int64, c_int64, c_float, c_void_p, c_int64, c_void_p]),
    "mnf_occ_binarize": (c_int32, [c_void_p, c_int64, c_int32, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "mnf_update_occupancy": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_void_p, c_int32, c_int32, c_float,
                                       c_float, c_float, c_uint64, c_void_p, c_int64, c_void_p]),
    "mnf_generate_rays": (c_int32, [c_void_p, c_int32, c_int32, c_int32, c_float, c_void_p, c_int64, c_void_p, c_void_p, c_void_p]),
    "mnf_gather_pixels": (c_int32, [c_void_p, c_void_p, c_int32, c_void_p, c_int32, c_int64, c_void_p, c_void_p, c_int64, c_void_p, c_void_p,
                                    c_void_p, c_void_p]),
    "mnf_field_create": (c_int32, [POINTER(FieldConfig), POINTER(c_void_p)]),
    "mnf_field_destroy": (c_int32, [c_void_p]),
    "mnf_field_param_count": (c_int64, [c_void_p, c_int32]),
    "mnf_field_grid_meta_host": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "mnf_field_set_params": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "mnf_field_refresh_weights": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "mnf_field_table_mirror": (c_void_p, [c_void_p, POINTER(c_int64)]),
    "mnf_field_forward": (c_int32, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p]),
    "mnf_field_density": (c_int32, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p]),
    "mnf_field_forward_samples": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64,
                                            c_void_p, c_void_p, c_void_p, c_void_p]),
    "mnf_field_density_rays": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int64,
                                         c_float, c_void_p, c_void_p]),
    "mnf_vanilla_create": (c_int32, [POINTER(VanillaConfig), POINTER(c_void_p)]),
    "mnf_vanilla_destroy": (c_int32, [c_void_p]),
    "mnf_vanilla_param_count": (c_int64, [c_void_p]),
    "mnf_vanilla_param_layout_host": (c_int32, [c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_void_p]),
    "mnf_vanilla_set_params": (c_int32, [c_void_p, c_void_p, c_void_p]),
    "mnf_vanilla_train_workspace_bytes": (c_int64, [c_void_p, c_int64]),
    "mnf_vanilla_forward": (c_int32, [c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "mnf_vanilla_density": (c_int32, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p]),
    "mnf_vanilla_backward": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p]),
    "mnf_field_train_workspace_bytes": (c_int64, [c_void_p, c_int64]),
    "mnf_field_forward_train": (c_int32, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "mnf_field_backward": (c_int32, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64,
                                     c_float, c_void_p, c_void_p, c_void_p, c_void_p]),
    "mnf_field_forward_train_samples": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p,
                                                  c_void_p, c_void_p, c_int64, c_void_p]),
    "mnf_presample_create": (c_int32, [POINTER(c_void_p)]),
    "mnf_presample_destroy": (None, [c_void_p]),
    "mnf_train_presample_workspace_bytes": (c_int64, [c_int32, c_int64]),
    "mnf_train_presample": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_int32,
                                      POINTER(TrainOpts), c_int64, c_void_p, c_int64, c_void_p]),
    "mnf_presample_wait": (c_int32, [c_void_p, c_void_p]),
    "mnf_train_step_workspace_bytes": (c_int64, [c_void_p, c_int32, c_int64, c_int64]),
    "mnf_train_step": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_int32,
                                 c_void_p, c_void_p, c_void_p, POINTER(TrainOpts), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64,
                                 c_int64, c_void_p, c_int64, c_void_p]),
    "mnf_score_poses_workspace_bytes": (c_int64, [c_int32, c_int32, c_int32, c_int32]),
    "mnf_score_poses": (c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_int32, c_int32, c_int32,
                                  c_float, c_void_p, c_int64, POINTER(RenderOpts), c_void_p, c_void_p, c_int64, c_void_p]),
    "mnf_render_workspace_bytes": (c_int64, [c_int64, c_int32]),
    "mnf_render_test": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_int32, POINTER(c_float), c_void_p, c_void_p,
                                  c_int64, POINTER(RenderOpts), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                  c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "mnf_render_jobs": (c_int32, [POINTER(RenderJob), c_int32, c_int32, c_int32, c_int32, POINTER(c_float), POINTER(RenderOpts), c_void_p]),
    "mnf_planner_map": (c_int32, [c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p]),
    "mnf_profile_begin": (c_int32, []),
    "mnf_profile_end": (c_int32, [POINTER(c_double), POINTER(c_int64)]),
    "mnf_profile_query": (c_int32, [c_char_p, POINTER(c_double), POINTER(c_int64)]),
    "mnf_score_views": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p]),
}


def lib_path() -> str:
    return _LIB_FILE


def load_library():
    """dlopen the in-tree library and bind every symbol; raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_FILE):
            raise MnfError(f"{_LIB_FILE} is missing: build it with `python __graft_entry__.py build` "
                           "(hipcc, gfx950). There is no CPU fallback.")
        lib = ctypes.CDLL(_LIB_FILE)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        _lib = lib
    return _lib


def check(rc: int) -> None:
    if rc != 0:
        raise MnfError(load_library().mnf_last_error().decode() or f"libmi355nerf error {rc}")


def require_gpu(*tensors) -> None:
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise MnfError("libmi355nerf operates on GPU tensors only (got a CPU tensor); there is no CPU fallback")


class DevPtr(c_void_p):
    """A device pointer that remembers which GPU owns it (`launch` derives the device guard and the stream from it)."""
    device = None


def ptr(t):
    if t is None:
        return None
    p = DevPtr(t.data_ptr())
    p.device = t.device if t.is_cuda else None
    return p


def stream(device=None):
    """hipStream_t of torch's current stream on `device` (default: the current device)."""
    return c_void_p(torch.cuda.current_stream(device).cuda_stream)


def launch(fn, *args):
    """Call a stream-taking entry point of the C ABI: the GPU that owns the tensors (first device pointer among
    `args`) is made current for the duration of the call and the work goes onto torch's current stream OF THAT GPU —
    a model on cuda:1 runs on GPU 1 whatever the caller's current device is (the reference's `DEVICE_GUARD`,
    utils_cuda.cuh:23-24).  Appends the stream argument and raises MnfError on a non-zero return code."""
    dev = next((a.device for a in args if isinstance(a, DevPtr) and a.device is not None), None)
    if dev is None:      # an anchor object (`.p` = DevPtr) names the GPU of a call whose device pointers sit inside a host array
        anchor = next((a for a in args if hasattr(a, "p") and isinstance(getattr(a, "p"), DevPtr)), None)
        if anchor is not None:
            dev = anchor.p.device
            args = tuple(a for a in args if a is not anchor)
    if dev is None:
        check(fn(*args, stream()))
        return
    with torch.cuda.device(dev):
        check(fn(*args, stream(dev)))


def contig(t, dtype=None):
    if t is None:
        return None
    if dtype is not None and t.dtype != dtype:
        t = t.to(dtype)
    return t.contiguous()
