"""Generator-side only (imported by make_golden.py in the build container; never by a test, the package or bench.py).

Lets the reference's OWN Python glue run on a CPU: perception/models/utils.py, perception/nerfacc/nerfacc/{grid,volrend,scan,
pack,data_specs}.py, estimators/occ_grid.py, perception/data_proc/habitat_to_data.py and scripts/pipeline.py are imported
from /root/reference unmodified; only what sits BELOW them — the three entry points of the CUDA extension `nerfacc_cuda`
that the path reaches, and the `is_cuda` gate of pack_info — is substituted:

  _C.ray_aabb_intersect   -> the reference's own torch twin nerfacc.grid._ray_aabb_intersect (grid.py:54-90)
  _C.exclusive_sum        -> the reference's own batched branch (scan.py:85-88: cat + cumsum) applied chunk by chunk; the
                             backward flag runs the same branch on the flipped chunk (scan.cu:68-125 semantics)
  _C.traverse_grids       -> oracle/nerfacc_grid.c (the restatement of grid.cu:68-282 the parity tests already use), its
                             outputs wrapped as RaySegmentsSpec-like objects of torch tensors (data_spec.hpp:6-106)
  nerfacc.pack.pack_info  -> the reference's own function body, re-compiled from ITS source with the expression
                             `ray_indices.is_cuda` replaced by True (pack.py:37-48; nothing is re-typed)

and the modules that are absent from this container and unused by the path (imageio, cv2, skimage, lpips, habitat_sim,
tinycudann, the planner / simulator imports of pipeline.py) are registered as empty placeholders so the imports succeed.
"""
import ast
import inspect
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"


class _Spec:
    """data_spec.hpp:6-27 RaySegmentsSpec as the Python layer reads it (data_specs.py:64-83, :159-176)."""
    vals = ray_indices = is_left = is_right = is_valid = chunk_starts = chunk_cnts = None


def _b(a):
    return None if a is None else torch.from_numpy(np.ascontiguousarray(a).astype(bool))


class ShimC:
    """Stands where `nerfacc.cuda._backend._C` (the pybind module of nerfacc.cpp:100-128) would."""

    def __init__(self):
        self.traverse_log = []       # (n_masked_in, traverse_steps_limit) per call: the round schedule of the test renderers

    def RaySegmentsSpec(self):
        return _Spec()

    @staticmethod
    def ray_aabb_intersect(rays_o, rays_d, aabbs, near_plane, far_plane, miss_value):
        from nerfacc.grid import _ray_aabb_intersect
        return _ray_aabb_intersect(rays_o, rays_d, aabbs, near_plane, far_plane, miss_value)

    @staticmethod
    def exclusive_sum(chunk_starts, chunk_cnts, inputs, normalize, backward):
        from nerfacc.scan import exclusive_sum as ref_exclusive_sum
        assert not normalize
        out = torch.zeros_like(inputs)
        for s, c in zip(chunk_starts.tolist(), chunk_cnts.tolist()):
            if c == 0:
                continue
            seg = inputs[s:s + c]
            if backward:
                out[s:s + c] = ref_exclusive_sum(seg.flip(0)[None])[0].flip(0)
            else:
                out[s:s + c] = ref_exclusive_sum(seg[None])[0]
        return out

    def traverse_grids(self, rays_o, rays_d, rays_mask, binaries, aabbs, t_sorted, t_indices, hits, near_planes, far_planes,
                       step_size, cone_angle, compute_intervals, compute_samples, compute_terminate_planes,
                       traverse_steps_limit, over_allocate):
        from oracle import marcher as M
        assert compute_intervals and compute_samples and compute_terminate_planes
        self.traverse_log.append((int(rays_mask.sum().item()), int(traverse_steps_limit)))
        iv, sm, term = M.traverse_grids(
            rays_o.numpy(), rays_d.numpy(), binaries.numpy(), aabbs.numpy(), near_planes.numpy(), far_planes.numpy(),
            float(step_size), float(cone_angle), None if traverse_steps_limit < 0 else int(traverse_steps_limit),
            bool(over_allocate), rays_mask.numpy(), t_sorted.numpy(), t_indices.numpy(), hits.numpy())
        a, b = _Spec(), _Spec()
        a.vals = torch.from_numpy(iv.vals)
        a.ray_indices = torch.from_numpy(iv.ray_indices)
        a.is_left, a.is_right = _b(iv.is_left), _b(iv.is_right)
        a.chunk_starts = torch.from_numpy(np.ascontiguousarray(iv.packed_info[:, 0]))
        a.chunk_cnts = torch.from_numpy(np.ascontiguousarray(iv.packed_info[:, 1]))
        b.vals = torch.from_numpy(sm.vals)
        b.ray_indices = torch.from_numpy(sm.ray_indices)
        b.is_valid = _b(sm.is_valid)
        b.chunk_starts = torch.from_numpy(np.ascontiguousarray(sm.packed_info[:, 0]))
        b.chunk_cnts = torch.from_numpy(np.ascontiguousarray(sm.packed_info[:, 1]))
        return a, b, torch.from_numpy(term)


def _placeholder(name, **attrs):
    if name in sys.modules:
        return sys.modules[name]
    try:
        __import__(name)
        return sys.modules[name]
    except Exception:
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        parent, _, child = name.rpartition(".")
        if parent:
            setattr(_placeholder(parent), child, m)
        return m


def _recompile_pack_info():
    """pack.py:10-49 with `ray_indices.is_cuda` -> True: the reference's own statements, run on CPU tensors."""
    import nerfacc.pack as P
    tree = ast.parse(inspect.getsource(P))

    class Gate(ast.NodeTransformer):
        def visit_Attribute(self, node):
            if node.attr == "is_cuda":
                return ast.copy_location(ast.Constant(True), node)
            return self.generic_visit(node)

    ns = dict(P.__dict__)
    exec(compile(ast.fix_missing_locations(Gate().visit(tree)), P.__file__, "exec"), ns)
    return ns["pack_info"]


def enter_reference(with_pipeline=False):
    """chdir into the reference, extend sys.path as its scripts do, install the substitutions.  Returns the ShimC instance."""
    os.chdir(REF)
    for p in ("perception/nerfacc", "perception/models", "perception/data_proc", "scripts"):
        q = os.path.join(REF, p)
        if q not in sys.path:
            sys.path.insert(0, q)
    for name in ("imageio", "cv2", "skimage", "skimage.io", "skimage.color"):
        _placeholder(name)
    import nerfacc  # noqa: F401
    import nerfacc.cuda._backend  # noqa: F401  (prints "No CUDA toolkit found", leaves _C = None)
    import utils  # noqa: F401  perception/models/utils.py: pulls in the fork's top-level twins `cuda`, `scan`, `pack`, `volrend`
    import cuda._backend  # noqa: F401  the fork's scan.py:12 does `import cuda as _C`: a second module object of the same file
    shim = ShimC()
    pk = _recompile_pack_info()
    for m in list(sys.modules.values()):
        f = getattr(m, "__file__", None) or ""
        if not f.startswith(REF):
            continue
        if f.endswith("cuda/_backend.py"):
            m._C = shim
        if getattr(getattr(m, "pack_info", None), "__module__", "") in ("pack", "nerfacc.pack"):
            m.pack_info = pk
    if with_pipeline:
        _placeholder("lpips", LPIPS=object)
        _placeholder("habitat_sim")
        _placeholder("habitat_sim.utils")
        _placeholder("habitat_sim.utils.common", d3_40_colors_rgb=np.zeros((40, 3), np.uint8))
        _placeholder("tinycudann")
        _placeholder("planning_funcs")
        _placeholder("sim", HabitatSim=object)
    return shim
