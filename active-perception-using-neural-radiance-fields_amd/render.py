"""Render orchestration with the reference's names and signatures
(perception/models/utils.py, perception/data_proc/habitat_to_data.py, scripts/pipeline.py:666-798),
running on libmi355nerf.so.

  Rays                                            perception/models/datasets/utils.py:7
  render_image_with_occgrid_test                  utils.py:555-779
  render_probablistic_image_with_occgrid_test     utils.py:782-1032
  render_image_with_occgrid_with_depth_guide      utils.py:63-219   (forward; see DESIGN.md §Scope for backward)
  sem_rendering                                   utils.py:362-461
  generate_image_rays / render_*_from_pose        habitat_to_data.py:274-549
  probablistic_uncertainty -> score_views         pipeline.py:666-798
"""
import collections
import ctypes
import weakref
from typing import Optional

import numpy as np
import torch

from . import _lib as L
from . import nerfacc as NA

Rays = collections.namedtuple("Rays", ("origins", "viewdirs"))

_WORKSPACES = collections.OrderedDict()
MAX_CACHED_WORKSPACES = 12      # (device, slot, stream) entries kept; the least recently used one goes first.  Grows to the largest job count of a render call + 4


def _workspace(key, nbytes: int) -> torch.Tensor:
    """Cached scratch buffer; `key` is a device or (device, slot) — concurrent render jobs of one call use one slot each.  The buffer belongs to the
    (device, slot) AND the stream the caller enqueues on: two members of an ensemble trained or rendered side by side on two streams must not share
    scratch (round 4: they did, silently, through the per-device cache).
    Cost: one buffer per (device, slot, stream) in use, sized for that key's largest call so far — hundreds of MB for a train step (feature rows, bin lists).
    At most MAX_CACHED_WORKSPACES entries are kept (least recently used evicted: a caller that makes streams per phase or per epoch no longer accumulates
    buffers); `release_workspaces(stream)` drops a stream's buffers at once.  Eviction is safe while kernels still use a buffer: it was allocated
    under the stream it is keyed by, so torch's caching allocator hands its memory only to later work of that same stream."""
    device = key[0] if isinstance(key, tuple) else key
    key = (key, torch.cuda.current_stream(device).cuda_stream) if torch.device(device).type == "cuda" else key
    ws = _WORKSPACES.get(key)
    if ws is None or ws.numel() < nbytes:
        _WORKSPACES[key] = ws = torch.empty(nbytes, dtype=torch.uint8, device=device)
    _WORKSPACES.move_to_end(key)
    while len(_WORKSPACES) > MAX_CACHED_WORKSPACES:
        _WORKSPACES.popitem(last=False)
    return ws


def release_workspaces(stream=None):
    """Drop the cached workspaces: all of them, or those keyed by `stream` (a torch.cuda.Stream or a raw handle)."""
    if stream is None:
        _WORKSPACES.clear()
        return
    h = getattr(stream, "cuda_stream", stream)
    for k in [k for k in _WORKSPACES if isinstance(k, tuple) and len(k) == 2 and k[1] == h]:
        del _WORKSPACES[k]


def _grid_levels(estimator, max_levels=4):
    """-> (binaries u8 [L,X,Y,Z] on device, the L aabbs as one flat list of 6 L host floats).  The host copy of the aabbs lives on the
    estimator and is refreshed when the tensor changes (`OccGridEstimator.aabb_host`), so a render call does not synchronise."""
    L_ = int(estimator.binaries.shape[0])
    if L_ > max_levels:
        raise NotImplementedError(f"the fused renderer supports up to {max_levels} occupancy levels (got {L_}); use nerfacc.traverse_grids")
    b = estimator.binaries
    b = b.contiguous().view(torch.uint8) if b.dtype == torch.bool else b.to(torch.uint8).contiguous()
    aabb = []
    for lvl in range(L_):
        aabb += list(estimator.aabb_host(lvl))
    return b, aabb




_VIEW_ORDERS = {}


def _view_order(h: int, w: int, device, block: int = 8) -> torch.Tensor:
    """Device int32 permutation of the h*w row-major pixels of a view into block x block tiles (tiles row-major, pixels
    row-major inside a tile): the order in which the renderer marches a view's rays (`mnf_render_opts.view_order`).
    A 64-sample tile of the field kernel then covers a compact pixel patch instead of a one-pixel-high strip and touches
    fewer hash-table lines (-5 % render time at 800x800); results are per ray and do not depend on it."""
    key = (h, w, block, str(device))
    t = _VIEW_ORDERS.get(key)
    if t is None:
        ys, xs = np.meshgrid(np.arange(h), np.arange(w), indexing="ij")
        k = ((ys // block) * ((w + block - 1) // block) + xs // block) * (block * block) + (ys % block) * block + xs % block
        t = torch.from_numpy(np.argsort(k.reshape(-1), kind="stable").astype(np.int32)).to(device)
        _VIEW_ORDERS[key] = t
    return t


def _render_jobs(specs, rpv, max_samples, near_plane, far_plane, render_step_size, render_bkgd, cone_angle, alpha_thre, early_stop_eps,
                 probabilistic, sync_every=8, image_hw=None, n_split=2):
    """`specs`: list of (radiance_field, estimator, origins [n,3], viewdirs [n,3]) sharing rays_per_view `rpv` and every option (the
    members of an ensemble looking at the same poses, or one model).  Every spec is cut into up to `n_split` contiguous groups of
    whole views, and all (spec, group) pairs advance side by side as render jobs of ONE C call (`mnf_render_jobs`): while one job's
    marcher or the tail of its field launch leaves compute units idle, the other jobs' kernels use them.  A view's result does not
    depend on the grouping (its round schedule and its sample tiles are its own).  Returns one dict of device tensors per spec."""
    lib = L.load_library()
    opts = L.RenderOpts()
    opts.near_plane, opts.far_plane, opts.render_step_size = near_plane, far_plane, render_step_size
    opts.cone_angle, opts.alpha_thre, opts.early_stop_eps = cone_angle, alpha_thre, early_stop_eps
    bk = [0.0, 0.0, 0.0] if render_bkgd is None else [float(x) for x in render_bkgd.detach().cpu().reshape(-1)[:3]]
    for i in range(3):
        opts.render_bkgd[i] = bk[i]
    opts.max_samples, opts.probabilistic, opts.rays_per_view, opts.sync_every = int(max_samples), int(probabilistic), int(rpv), sync_every
    dev = specs[0][2].device
    order = None
    if image_hw is not None and image_hw[0] * image_hw[1] == rpv and min(image_hw) >= 16:
        order = _view_order(image_hw[0], image_hw[1], dev)
    opts.view_order = None if order is None else order.data_ptr()
    opts.bitgrid = None
    jobs, keep, outs = [], [], []
    grid0 = None
    for radiance_field, estimator, o, d in specs:
        L.require_gpu(o, d)
        n = o.shape[0]
        if n % rpv:
            raise L.MnfError(f"n_rays ({n}) must be a positive multiple of rays_per_view ({rpv})")
        C = radiance_field.num_semantic_classes
        handle = radiance_field._ensure_handle()
        binaries, aabb = _grid_levels(estimator)
        L.require_gpu(binaries)
        if grid0 is None:
            grid0 = (tuple(binaries.shape[1:]), tuple(aabb))
        elif grid0 != (tuple(binaries.shape[1:]), tuple(aabb)):
            raise L.MnfError("render jobs of one call must share the occupancy grid resolution and aabb")
        bits = estimator.bitgrid() if hasattr(estimator, "bitgrid") else None     # the estimator's packed grid (no per-call packing)
        rgb = torch.empty(n, 3, device=dev); acc = torch.empty(n, 1, device=dev); depth = torch.empty(n, 1, device=dev)
        sem = torch.empty(n, C, device=dev)
        rgb_var = torch.empty(n, 3, device=dev) if probabilistic else None
        depth_var = torch.empty(n, 1, device=dev) if probabilistic else None
        V = n // rpv
        # n_split=None: the measured optimum for the shape — small views (the scorer's 64 x 64 sub-samples: a round is latency-bound) advance as four jobs, full
        # images as two (profiles/r03_hw_queues.txt: one member, 256 scoring views 54.3 -> 50.4 ms with four; 800 x 800 views: two and four tie)
        want = (4 if rpv <= 16384 else 2) if n_split is None else int(n_split)
        G = max(1, min(want, V))
        totals = torch.zeros(G, 2, dtype=torch.int64, device=dev)   # per job: [kept (reference total_samples), evaluated]
        keep += [binaries, bits, o, d, totals]
        for g in range(G):
            r0, r1 = (V * g // G) * rpv, (V * (g + 1) // G) * rpv
            if r1 == r0:
                continue
            nbytes = lib.mnf_render_workspace_bytes(r1 - r0, rpv)
            global MAX_CACHED_WORKSPACES
            MAX_CACHED_WORKSPACES = max(MAX_CACHED_WORKSPACES, len(jobs) + 5)      # a call never evicts the slots it is still filling: the cap follows the largest job count seen
            ws = _workspace((dev, len(jobs)), nbytes)
            j = L.RenderJob()
            j.field, j.binaries, j.bitgrid = handle, binaries.data_ptr(), (None if bits is None else bits.data_ptr())
            j.rays_o, j.rays_d, j.n_rays = o[r0:].data_ptr(), d[r0:].data_ptr(), r1 - r0
            j.rgb, j.acc, j.depth, j.sem = rgb[r0:].data_ptr(), acc[r0:].data_ptr(), depth[r0:].data_ptr(), sem[r0:].data_ptr()
            j.rgb_var = rgb_var[r0:].data_ptr() if probabilistic else None
            j.depth_var = depth_var[r0:].data_ptr() if probabilistic else None
            j.total_samples = totals[g].data_ptr()
            j.workspace, j.workspace_bytes = ws.data_ptr(), nbytes
            jobs.append(j)
            keep.append(ws)
        out = dict(rgb=rgb, acc=acc, depth=depth, sem=sem, _totals=totals)
        if probabilistic:
            out.update(rgb_var=rgb_var, depth_var=depth_var)
        outs.append(out)
    if jobs:
        arr = (L.RenderJob * len(jobs))(*jobs)
        res, aabb = grid0
        opts.n_levels = len(aabb) // 6
        anchor = L.ptr(specs[0][2])                  # device guard + stream of the GPU that owns the rays
        L.launch(lib.mnf_render_jobs, arr, len(jobs), res[0], res[1], res[2], (ctypes.c_float * len(aabb))(*aabb), ctypes.byref(opts), _Anchor(anchor))
    for out in outs:
        out["total"] = out.pop("_totals").sum(0)
    return outs


class _Anchor:
    """Carries a DevPtr through `L.launch` without adding an argument to the C call: launch() derives the device guard and the
    stream from the first DevPtr among its arguments; mnf_render_jobs takes its device pointers inside the job array."""

    def __init__(self, p):
        self.p = p


def _render_test(max_samples, radiance_field, estimator, rays, near_plane, far_plane, render_step_size, render_bkgd,
                 cone_angle, alpha_thre, early_stop_eps, probabilistic, rays_per_view=None, sync_every=8, image_hw=None, n_split=2):
    rays_shape = rays.origins.shape
    if image_hw is None and rays_per_view is None and len(rays_shape) == 3:
        image_hw = (int(rays_shape[0]), int(rays_shape[1]))      # [H,W,3] rays (utils.py:574-580): one image
    o = L.contig(rays.origins.reshape(-1, 3), torch.float32)
    d = L.contig(rays.viewdirs.reshape(-1, 3), torch.float32)
    L.require_gpu(o, d)
    n = o.shape[0]
    C = radiance_field.num_semantic_classes
    dev = o.device
    shp = tuple(rays_shape[:-1])
    if n == 0:
        z = lambda *s_: torch.empty(*s_, device=dev)
        out = dict(rgb=z(n, 3), acc=z(n, 1), depth=z(n, 1), sem=z(n, C), total=torch.zeros(2, dtype=torch.int64, device=dev))
        if probabilistic:
            out.update(rgb_var=z(n, 3), depth_var=z(n, 1))
    else:
        rpv = n if rays_per_view is None else int(rays_per_view)
        out = _render_jobs([(radiance_field, estimator, o, d)], rpv, max_samples, near_plane, far_plane, render_step_size, render_bkgd,
                           cone_angle, alpha_thre, early_stop_eps, probabilistic, sync_every, image_hw, n_split)[0]
    res = dict(rgb=out["rgb"].view(*shp, -1), acc=out["acc"].view(*shp, -1), depth=out["depth"].view(*shp, -1), sem=out["sem"].view(*shp, -1),
               total=out["total"])
    if probabilistic:
        res.update(rgb_var=out["rgb_var"].view(*shp, -1), depth_var=out["depth_var"].view(*shp, -1))
    return res


@torch.no_grad()
def render_image_with_occgrid_test(max_samples: int, radiance_field, estimator, rays: Rays, near_plane: float = 0.0,
                                   far_plane: float = 1e10, render_step_size: float = 1e-3,
                                   render_bkgd: Optional[torch.Tensor] = None, cone_angle: float = 0.0,
                                   alpha_thre: float = 0.0, early_stop_eps: float = 1e-4, timestamps=None):
    """utils.py:555-779 -> (rgb, acc, depth, sem, total_samples)."""
    if timestamps is not None:
        raise NotImplementedError("timestamps (D-NeRF) are not part of the hot path")
    r = _render_test(max_samples, radiance_field, estimator, rays, near_plane, far_plane, render_step_size, render_bkgd,
                     cone_angle, alpha_thre, early_stop_eps, False)
    return r["rgb"], r["acc"], r["depth"], r["sem"], int(r["total"][0].item())


@torch.no_grad()
def render_probablistic_image_with_occgrid_test(max_samples: int, radiance_field, estimator, rays: Rays,
                                                near_plane: float = 0.0, far_plane: float = 1e10,
                                                render_step_size: float = 1e-3, render_bkgd: Optional[torch.Tensor] = None,
                                                cone_angle: float = 0.0, alpha_thre: float = 0.0,
                                                early_stop_eps: float = 1e-4, timestamps=None):
    """utils.py:782-1032 -> (rgb, rgb_var, acc, depth, depth_var, sem, total_samples)."""
    if timestamps is not None:
        raise NotImplementedError("timestamps (D-NeRF) are not part of the hot path")
    r = _render_test(max_samples, radiance_field, estimator, rays, near_plane, far_plane, render_step_size, render_bkgd,
                     cone_angle, alpha_thre, early_stop_eps, True)
    return r["rgb"], r["rgb_var"], r["acc"], r["depth"], r["depth_var"], r["sem"], int(r["total"][0].item())


@torch.no_grad()
def render_views(radiance_field, estimator, rays_o, rays_d, rays_per_view, max_samples=1024, near_plane=0.0,
                 far_plane=1e10, render_step_size=1e-3, render_bkgd=None, cone_angle=0.0, alpha_thre=0.0,
                 early_stop_eps=1e-4, probabilistic=False, sync_every=8, image_hw=None, n_split=2):
    """Batched form: rays_o/rays_d [V*rays_per_view, 3]; every group of rays_per_view rays is rendered exactly as one
    call of the reference function (own round schedule).  `image_hw=(H, W)` says that the rays of a view are the row-major
    pixels of an H x W image (a speed hint only: see `_view_order`).  `n_split`: the views are rendered as up to this many
    groups advancing side by side (`_render_jobs`; a speed choice only: results do not depend on it).
    Returns a dict of device tensors."""
    return _render_test(max_samples, radiance_field, estimator, Rays(rays_o, rays_d), near_plane, far_plane, render_step_size,
                        render_bkgd, cone_angle, alpha_thre, early_stop_eps, probabilistic, rays_per_view, sync_every, image_hw, n_split)


# ------------------------------------------------------------------ train-mode forward (utils.py:63-219, :362-461)
_CompositeTrain = NA._CompositeTrain     # csrc/composite_train.hip behind autograd (defined next to `rendering`)


def sem_rendering(radiance_field, rays: Rays, t_starts, t_ends, ray_indices, n_rays, render_bkgd=None):
    """utils.py:362-461 on packed samples.  With autograd enabled the field is evaluated through its differentiable forward — the closure of
    utils.py:122-137 inside the kernel (`forward_samples_grad`), or in torch for a field without it or for rays / distances that themselves want
    gradients (none come back through the field, as in the reference) — and composited by `_CompositeTrain`; otherwise the fused no-grad kernel."""
    C = radiance_field.num_semantic_classes
    dev = t_starts.device
    differentiable = torch.is_grad_enabled() and any(p.requires_grad for p in radiance_field.parameters())
    if t_starts.shape[0] != 0 and differentiable and hasattr(radiance_field, "forward_samples_grad") and not (
            rays.origins.requires_grad or rays.viewdirs.requires_grad or t_starts.requires_grad or t_ends.requires_grad):
        rgbs, sigmas, sems = radiance_field.forward_samples_grad(rays.origins, rays.viewdirs, ray_indices, t_starts, t_ends)
        sigmas = sigmas.squeeze(-1)
    elif t_starts.shape[0] != 0 and differentiable:
        t_dirs = rays.viewdirs[ray_indices]
        positions = rays.origins[ray_indices] + t_dirs * (t_starts + t_ends)[:, None] / 2.0
        rgbs, sigmas, sems = radiance_field(positions, t_dirs)
        sigmas = sigmas.squeeze(-1)
    elif t_starts.shape[0] != 0:
        rgbs, sigmas, sems = radiance_field.forward_samples(rays.origins, rays.viewdirs, ray_indices, t_starts, t_ends)
    else:
        rgbs, sigmas, sems = torch.empty((0, 3), device=dev), torch.empty((0,), device=dev), torch.empty((0, C), device=dev)
    packed = NA.pack_info_grouped(ray_indices, n_rays)
    bk = None if render_bkgd is None else render_bkgd.to(device=dev, dtype=torch.float32).reshape(3).contiguous()
    colors, opacities, depths, semantics, weights, trans, alphas = _CompositeTrain.apply(
        packed[:, 0].contiguous(), packed[:, 1].contiguous(), t_starts, t_ends, sigmas, rgbs, sems, bk)
    return colors, opacities, depths, semantics, dict(weights=weights, alphas=alphas, trans=trans, sigmas=sigmas, rgbs=rgbs)


def render_image_with_occgrid_with_depth_guide(radiance_field, estimator, rays: Rays, near_plane: float = 0.0,
                                               far_plane: float = 1e10, render_step_size: float = 1e-3,
                                               render_bkgd: Optional[torch.Tensor] = None, cone_angle: float = 0.0,
                                               alpha_thre: float = 0.0, test_chunk_size: int = 8192, timestamps=None,
                                               depth: Optional[torch.Tensor] = None):
    """utils.py:63-219: occupancy sampling with the density pre-pass (no grad), then semantic volume rendering.
    Differentiable w.r.t. the field parameters when autograd is enabled (the reference's training forward);
    `depth` is accepted and ignored as in the reference."""
    if timestamps is not None:
        raise NotImplementedError("timestamps (D-NeRF) are not part of the hot path")
    rays_shape = rays.origins.shape
    o, d = rays.origins.reshape(-1, 3), rays.viewdirs.reshape(-1, 3)
    num_rays = o.shape[0]
    chunk = torch.iinfo(torch.int32).max if radiance_field.training else test_chunk_size
    results = []
    for i in range(0, num_rays, chunk):
        co, cd = o[i:i + chunk], d[i:i + chunk]

        from .ngp import RaySigmaFn
        sigma_fn = RaySigmaFn(radiance_field, co, cd)      # utils.py:89-101; lets `sampling` skip what lies behind opaque surfaces

        ray_indices, t_starts, t_ends = estimator.sampling(co, cd, sigma_fn=sigma_fn, near_plane=near_plane, far_plane=far_plane,
                                                           render_step_size=render_step_size, stratified=radiance_field.training,
                                                           cone_angle=cone_angle, alpha_thre=alpha_thre, depth=depth)
        rgb, opacity, dep, semantics, _ = sem_rendering(radiance_field, Rays(co, cd), t_starts, t_ends, ray_indices, co.shape[0],
                                                        render_bkgd)
        results.append((rgb, opacity, dep, semantics, len(t_starts)))
    colors, opacities, depths, semantics = (torch.cat([r[k] for r in results], 0) for k in range(4))
    shp = tuple(rays_shape[:-1])
    return colors.view(*shp, -1), opacities.view(*shp, -1), depths.view(*shp, -1), semantics.view(*shp, -1), sum(r[4] for r in results)


def allreduce_gradients(parameters, group=None, skip=None):
    """Ray-data-parallel training: every rank renders its own slice of the ray batch, then the gradients of
    the three flat parameter vectors are averaged with one all-reduce each (RCCL over xGMI on GPUs; the hash-table vector
    is 100 MB, the two heads a few KB).  `skip` (device int32 scalar): the per-rank "do not apply this step" flag is summed
    over the ranks as well, so every rank takes the same decision.  No-op without an initialised process group of more
    than one rank.  EVERY rank must call it every step (a rank whose rays produced no sample contributes zero gradients)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return
    world = dist.get_world_size(group)
    if world < 2:
        return
    for p_ in parameters:
        if p_.numel() == 0:
            continue                                   # (the reference's `direction_encoding` has no parameters)
        if p_.grad is None:                            # every rank issues the same collectives, whatever its own backward produced
            p_.grad = torch.zeros_like(p_)
        dist.all_reduce(p_.grad, op=dist.ReduceOp.SUM, group=group)
        p_.grad.div_(world)
    if skip is not None:
        dist.all_reduce(skip, op=dist.ReduceOp.SUM, group=group)


_TRAIN_STATE = {}

# status bits of mnf_train_step's counts_dev[3] (include/mi355nerf.h)
_ST_MARCHED, _ST_ROW, _ST_KEPT, _ST_LABEL, _ST_EMPTY = 1, 2, 4, 8, 16


def _grow_caps(st, marched, kept, R, carry):
    """Raise the sample bounds of a field's train state after a step that overflowed them.  The state is per FIELD, not per ray count (the
    reference changes `num_rays` after every iteration, scripts/pipeline.py:494-504, to hold the sample count near `target_sample_batch_size`).
      carry=False (synchronous steps: the overflowing step is repeated at once): only the bounds of this ray count grow — exactly the bounds
                  rounds 1-3 used, which the deterministic mode's partial-sum grouping (and with it the bitwise identity of the stand-in
                  scenes across rounds) depends on;
      carry=True  (asynchronous steps, counts read late): an absolute bound (what the schedule holds constant) and a per-ray bound (what a
                  fixed-size batch holds constant) grow, and every later call — whatever its ray count — uses at least the larger of the two."""
    cm, ck = int(marched * 1.3) + 1024, int(max(kept, marched // 2) * 1.3) + 1024
    if carry:
        st["abs_m"], st["abs_k"] = max(st["abs_m"], cm), max(st["abs_k"], ck)
        st["per_m"], st["per_k"] = max(st["per_m"], cm / max(R, 1)), max(st["per_k"], ck / max(R, 1))
    else:
        bm, bk = st["by_R"].get(R, (max(R * 384, 1 << 18), R * 192))
        st["by_R"][R] = (max(bm, cm), max(bk, ck))


def _caps_for(st, R):
    """(max_marched, max_kept) of a call with R rays: generous per-ray defaults, raised by what earlier steps of this field overflowed."""
    # (the default marched bound has an absolute floor: small batches of the dynamic schedule march ~1000 samples per ray through empty-looking early
    #  grids, and an asynchronous step cannot be repeated; the surviving bound keeps its per-ray form — the deterministic mode's grouping depends on it)
    bm, bk = st["by_R"].get(R, (max(R * 384, 1 << 18), R * 192))
    return max(bm, st["abs_m"], int(st["per_m"] * R)), max(bk, st["abs_k"], int(st["per_k"] * R))


def _train_state(radiance_field):
    """The field's train state (sample bounds, counts in flight, pinned buffers): one per field whatever the ray count of a call (VERDICT r03
    weak 6: under the reference's schedule the ray count changes every step)."""
    key = id(radiance_field)
    st = _TRAIN_STATE.get(key)
    if st is None:
        st = _TRAIN_STATE[key] = dict(by_R={}, abs_m=0, abs_k=0, per_m=0.0, per_k=0.0, pending=[])
        weakref.finalize(radiance_field, _TRAIN_STATE.pop, key, None)      # (the key holds id(field): drop the entry with the field, or a later field could inherit it)
    return st


def reserve_sample_bounds(radiance_field, max_marched: int, max_kept: int):
    """Raise the field's sample bounds up front: every later train step of this field sizes its launches and workspace for at least `max_marched`
    samples in front of the visibility filter and `max_kept` behind it, whatever its ray count.  For asynchronous training (sync=False), where a step
    beyond its bounds cannot be repeated but is skipped on the device: a caller that knows its budget (`target_sample_batch_size` and the sampler's
    marched / surviving ratio) loses no step to the default per-ray bounds."""
    st = _train_state(radiance_field)
    st["abs_m"], st["abs_k"] = max(st["abs_m"], int(max_marched)), max(st["abs_k"], int(max_kept))


def latest_step_counts(radiance_field):
    """(n_rays, marched samples, surviving samples) of the most recent asynchronous train step of this field whose counts have ARRIVED on the
    host (they travel to pinned memory behind every `sync=False` step and are read at the start of the next call), or None.  What the
    reference's dynamic batch size (scripts/pipeline.py:494-504: num_rays *= target_sample_batch_size / n_rendering_samples) can use
    without waiting for the GPU: the schedule then runs one or two steps behind the counts, the sample bounds absorb the lag."""
    st = _TRAIN_STATE.get(id(radiance_field))
    return None if st is None else st.get("last_counts")


def _check_status(status):
    if status & _ST_LABEL:
        raise L.MnfError("train_step: a semantic class id lies outside [0, num_semantic_classes) (F.cross_entropy's device assert)")


class Presample:
    """A batch's march, done ahead of its train step (`presample`): the C handle, the tensors the march reads and wrote, and what it was made for."""

    def __init__(self):
        h = ctypes.c_void_p()
        L.check(L.load_library().mnf_presample_create(ctypes.byref(h)))
        self.handle = h
        self._box = [None]           # what the march reads and wrote; shared with the finalizer (which must not reference `self`)
        # ONE finalizer does both things in order (a separate __del__ could run after the handle was destroyed at interpreter exit):
        # a token dropped without a step first makes the stream wait for its march — its buffers go back to the allocator only behind it — then the handle goes.
        self._fin = weakref.finalize(self, Presample._drop, L.load_library(), h, self._box)

    @staticmethod
    def _drop(lib, h, box):
        try:
            if box[0] is not None:
                lib.mnf_presample_wait(h, L.stream(box[0][0].device))
        except Exception:
            pass
        box[0] = None
        lib.mnf_presample_destroy(h)

    @property
    def keep(self):
        return self._box[0]

    @keep.setter
    def keep(self, v):
        self._box[0] = v

    def wait(self, device=None):
        """torch's current stream waits for the march (no-op if none was launched or the handle is gone)."""
        if self._fin.alive:
            L.check(L.load_library().mnf_presample_wait(self.handle, L.stream(device)))


def _grid_version(estimator):
    b, o = estimator.binaries, estimator.occs
    return (b.data_ptr(), b._version, o.data_ptr(), o._version)


@torch.no_grad()
def presample(radiance_field, estimator, rays: Rays, near_plane=0.1, far_plane=1e10, render_step_size=1e-3, cone_angle=0.004, alpha_thre=0.01,
              stratified=None, handle: "Presample | None" = None, seed: "int | None" = None):
    """March the NEXT batch now, beside the train step that is enqueued after this call (`mnf_train_presample`, include/mi355nerf.h): the sampler reads
    rays and occupancy grid, not the model, and with one lane per ray it leaves most of the chip idle for ~0.2 ms of every step — the data loader's
    "prefetch the next batch" applied to occ_grid.py:181-208.  Usage, the reference loop of scripts/pipeline.py:447-532 with the batch fetched one
    iteration early:

        tok = None
        for step in range(n):
            # (made in front of step `step`, used by step + 1: worth making only if neither of the two refreshes the occupancy grid)
            nxt = presample(field, est, batches[step + 1].rays, ...) if step % 16 and (step + 1) % 16 else None
            train_step(field, est, opt, *batches[step], step=step, presampled=tok)
            tok = nxt

    Same options as the step's (they are checked).  Call it on the stream the adopting `train_step` will run on (the token's buffers belong to torch's caching
    allocator of that stream; `train_step_ensemble` takes care of its members' streams).  Draws the step's jitter seed from torch's CPU generator, as the step itself would have.  Pass
    `handle` to recycle a consumed `Presample`.  A token that does not fit its step any more (grid refreshed in between, other options) is ignored by
    `train_step`, which then marches itself with the token's seed: results never depend on whether the token was used."""
    if estimator.levels > 4:
        return None
    lib = L.load_library()
    o, d = L.contig(rays.origins.reshape(-1, 3), torch.float32), L.contig(rays.viewdirs.reshape(-1, 3), torch.float32)
    L.require_gpu(o, d)
    R = o.shape[0]
    binaries, aabb = _grid_levels(estimator)
    bits = estimator.bitgrid()
    res = binaries.shape[1:]
    opts = L.TrainOpts()
    opts.n_levels = len(aabb) // 6
    opts.near_plane, opts.far_plane, opts.render_step_size, opts.cone_angle, opts.alpha_thre = near_plane, far_plane, render_step_size, cone_angle, alpha_thre
    opts.stratified = int(radiance_field.training if stratified is None else stratified)
    opts.seed = int(torch.randint(0, 2 ** 62, (1,)).item()) if seed is None else int(seed)
    tok = handle if handle is not None else Presample()
    cap_m = _caps_for(_train_state(radiance_field), R)[0]      # the adopting step's bound on marched samples (the march's guard and compaction run here too)
    nbytes = int(lib.mnf_train_presample_workspace_bytes(R, cap_m))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=o.device)
    L.launch(lib.mnf_train_presample, tok.handle, L.ptr(binaries), L.ptr(bits[0]), L.ptr(estimator.occs), res[0], res[1], res[2],
             (ctypes.c_float * len(aabb))(*aabb), L.ptr(o), L.ptr(d), R, ctypes.byref(opts), cap_m, L.ptr(ws), nbytes)
    tok.keep = (o, d, ws, binaries, bits)
    tok.seed, tok.R, tok.version, tok.cap_m = int(opts.seed), R, _grid_version(estimator), cap_m
    tok.key = (float(near_plane), float(far_plane), float(render_step_size), float(cone_angle), float(alpha_thre), int(opts.stratified))
    tok.rays = (rays.origins, rays.viewdirs)
    return tok


REFRESH_EVERY = 16      # the occupancy refresh cadence of `train_step` (pipeline.py:447-470 calls update_every_n_steps with its default n = 16): ONE constant for the refresh
                        # itself, the presample wait in front of a refreshing step, and the batches `presampled_batches` leaves un-presampled


def presampled_batches(batches, radiance_field, estimator, first_step: int = 0, refresh_every: int = REFRESH_EVERY, **render_kw):
    """The reference loop's batches with the march of each one done an iteration early: wraps any iterable of batches (each a tuple or list whose first element is
    the batch's `Rays`) and yields `(step, batch, token)`; pass `presampled=token` to `train_step`.  The next batch is drawn from `batches` and its march enqueued
    (`presample`) BEFORE the current one is yielded, i.e. before the caller enqueues the current step, beside which it then runs.  Batches on either side of an
    occupancy refresh (`step % refresh_every == 0`, as `update_every_n_steps`) are not presampled — their steps march themselves.  `render_kw`: the render
    options the steps will use (near_plane, render_step_size, cone_angle, alpha_thre, stratified).

        for step, (rays, pixels, dep, sem), tok in presampled_batches(loader, field, est, **RENDER_KW):
            train_step(field, est, opt, rays, pixels, dep, sem, bkgd, step=step, sync=False, presampled=tok, **RENDER_KW)"""
    it = iter(batches)
    try:
        cur = next(it)
    except StopIteration:
        return
    if refresh_every != REFRESH_EVERY:
        raise ValueError(f"presampled_batches: train_step refreshes the occupancy grid every {REFRESH_EVERY} steps; a different cadence here would let a refresh "
                         "rewrite the grid under a march in flight")
    step, tok = first_step, None
    while True:
        try:
            nxt = next(it)
        except StopIteration:
            nxt = None
        nxt_tok = None
        if nxt is not None and step % refresh_every and (step + 1) % refresh_every:
            nxt_tok = presample(radiance_field, estimator, nxt[0], **render_kw)
        yield step, cur, tok
        if nxt is None:
            return
        cur, tok, step = nxt, nxt_tok, step + 1


@torch.no_grad()
def fused_forward_backward(radiance_field, estimator, rays: Rays, pixels, dep, sem, render_bkgd=None, near_plane=0.1, far_plane=1e10,
                           render_step_size=1e-3, cone_angle=0.004, alpha_thre=0.01, early_stop_eps=1e-4, stratified=None, sync=True,
                           deterministic=False, presampled: "Presample | None" = None, seed: "int | None" = None, _defer_report=False):
    """scripts/pipeline.py:472-518 for one model as ONE C call (`mnf_train_step`, csrc/trainstep.hip): train render (occupancy
    sampling + density pre-pass + visibility filter + sem_rendering), the three-term loss and its backward.  Fills `.grad` of
    the three flat parameter vectors.  The call itself never waits for the GPU: the sample counts stay on the device.

    Returns dict(loss, loss_rgb, loss_dep, loss_sem [device scalars], counts [device int64: marched, kept, longest ray, status],
    skip [device int32 scalar: non-zero = the gradients must not be applied], n_rendering_samples, n_marched).
      sync=True   (default) the counts are read back once, AFTER everything is enqueued: n_rendering_samples / n_marched are
                  ints as in the reference; a step whose sample bounds were too small is repeated with larger ones.
      sync=False  nothing is read back: n_rendering_samples / n_marched are 0-d device tensors; the bounds are adapted from the
                  previous call's counts (copied to pinned memory in the background); a step beyond its bounds is skipped on
                  the device (skip != 0) and the bounds grow for the next one.
    Returns None when this estimator / batch cannot take the fused path (more than four occupancy levels, a ray longer than the
    single-pass scratch row): the caller then uses the autograd path, which is the same arithmetic in separate calls."""
    if estimator.levels > 4:
        return None
    lib = L.load_library()
    o, d = L.contig(rays.origins.reshape(-1, 3), torch.float32), L.contig(rays.viewdirs.reshape(-1, 3), torch.float32)
    if presampled is not None and presampled.keep is not None and presampled.rays[0] is rays.origins and presampled.rays[1] is rays.viewdirs:
        o, d = presampled.keep[0], presampled.keep[1]        # (the very tensors the march read: the C side compares pointers)
    L.require_gpu(o, d, pixels, dep, sem)
    R, dev = o.shape[0], o.device
    handle = radiance_field._ensure_handle()
    binaries, aabb = _grid_levels(estimator)
    bits = estimator.bitgrid()
    res = binaries.shape[1:]
    opts = L.TrainOpts()
    opts.n_levels = len(aabb) // 6
    opts.near_plane, opts.far_plane, opts.render_step_size, opts.cone_angle = near_plane, far_plane, render_step_size, cone_angle
    opts.alpha_thre, opts.early_stop_eps, opts.loss_scale = alpha_thre, early_stop_eps, float(radiance_field.loss_scale)
    bk_dev = None
    if render_bkgd is not None and render_bkgd.is_cuda:          # pipeline.py:437 draws the colour on the GPU: hand over the pointer
        bk_dev = L.contig(render_bkgd.detach().reshape(-1)[:3].to(device=dev, dtype=torch.float32))
        opts.render_bkgd_dev = bk_dev.data_ptr()
    else:
        bk = [0.0, 0.0, 0.0] if render_bkgd is None else [float(x) for x in render_bkgd.detach().reshape(-1)[:3]]
        for i in range(3):
            opts.render_bkgd[i] = bk[i]
        opts.render_bkgd_dev = None
    opts.stratified = int(radiance_field.training if stratified is None else stratified)
    use_pre = False
    if presampled is not None and presampled.keep is not None:
        # the token's seed in any case (the batch's jitter was drawn when it was presampled); its march only if it still fits this step
        opts.seed = presampled.seed
        use_pre = (presampled.R == R and presampled.keep[0] is o and presampled.version == _grid_version(estimator)
                   and presampled.key == (float(near_plane), float(far_plane), float(render_step_size), float(cone_angle), float(alpha_thre), int(opts.stratified)))
        if not use_pre:
            presampled.wait(dev)                                               # (its buffers are released below: not before the march has run)
    else:
        opts.seed = int(torch.randint(0, 2 ** 62, (1,)).item()) if seed is None else int(seed)      # (`seed`: the jitter's Philox key, for reproducible runs)
    opts.deterministic = int(bool(deterministic))      # bitwise reproducible gradient accumulation (include/mi355nerf.h: mnf_train_opts)
    params = [radiance_field.mlp_base.params, radiance_field.mlp_head.params, radiance_field.mlp_sem.params]
    for p_ in params:
        if p_.grad is None or p_.grad.shape != p_.shape or not p_.grad.is_contiguous():
            p_.grad = torch.empty_like(p_)
    tp, td, tl = L.contig(pixels, torch.float32), L.contig(dep, torch.float32), L.contig(sem, torch.int64)
    st = _train_state(radiance_field)
    # counts of earlier lazy steps (copied to pinned memory behind each step): adapt the bounds, surface their errors.  Everything that
    # has arrived is read; the step enqueued two calls ago is waited for (the host is at least a step ahead of the GPU: no stall), so a
    # bound that is too small is corrected at most two steps late.
    pending = st["pending"]
    while pending and (sync or len(pending) > 1 or pending[0][1].query()):
        host, ev, r_then = pending.pop(0)                       # (whatever ray count that step had: the bounds are the field's)
        ev.synchronize()
        c = host.tolist()
        if len(c) > 4 and c[4] > 0:                             # that step was skipped on the device (no samples, a bound, non-finite gradients): train_step(sync=False)
            st["skipped_steps"] = st.get("skipped_steps", 0) + 1      # had already advanced the LR scheduler for it — it owes one scheduler step
            st["sched_debt"] = st.get("sched_debt", 0) + 1
        c = c[:4]
        st["last_counts"] = (r_then, int(c[0]), int(c[1]))       # latest_step_counts(): the dynamic ray-count schedule without a host round trip
        st.setdefault("pinned", []).append((host, ev))
        _check_status(c[3])
        if c[3] & _ST_ROW:
            raise L.MnfError("train_step: a ray has more samples than a scratch row holds (use the autograd path: fused=False)")
        if c[3] & (_ST_MARCHED | _ST_KEPT):
            st["overflowed_steps"] = st.get("overflowed_steps", 0) + 1
            _grow_caps(st, c[0], c[1], r_then, carry=True)
    for _attempt in range(4):
        losses = torch.empty(4, device=dev)
        counts = torch.empty(4, dtype=torch.int64, device=dev)
        skip = torch.empty((), dtype=torch.int32, device=dev)
        cap_m, cap_k = _caps_for(st, R)
        nbytes = int(lib.mnf_train_step_workspace_bytes(handle, R, cap_m, cap_k))
        ws = _workspace(dev, nbytes)
        # (a repeated step marches itself: the guards cleared the token's counts; so does a step whose bound has grown since its token was made)
        adopt = use_pre and _attempt == 0 and presampled.cap_m == cap_m
        if use_pre and _attempt == 0 and not adopt:
            use_pre = False
            presampled.wait(dev)
        opts.presampled = presampled.handle if adopt else None
        L.launch(lib.mnf_train_step, handle, L.ptr(binaries), L.ptr(bits[0]), L.ptr(estimator.occs), res[0], res[1], res[2],
                 (ctypes.c_float * len(aabb))(*aabb), L.ptr(o), L.ptr(d), R, L.ptr(tp), L.ptr(td), L.ptr(tl), ctypes.byref(opts),
                 L.ptr(params[0].grad), L.ptr(params[1].grad), L.ptr(params[2].grad), L.ptr(losses), L.ptr(counts), L.ptr(skip),
                 cap_m, cap_k, L.ptr(ws), nbytes)
        if not sync:
            break
        c = counts.tolist()                                         # the step's one host round trip, after everything is enqueued
        _check_status(c[3])
        if c[3] & _ST_ROW:
            return None
        if c[3] & (_ST_MARCHED | _ST_KEPT):                         # a sample bound was too small: grow and redo
            _grow_caps(st, c[0], c[1], R, carry=False)
            continue
        break
    else:
        raise L.MnfError("train_step: sample bounds kept growing")
    if presampled is not None:
        presampled.adopted = use_pre
        presampled.keep = None                                      # consumed (or discarded): the step is enqueued, stream order protects the buffers
    for p_ in params:
        torch.autograd.graph.increment_version(p_.grad)
    out = dict(loss=losses[0], loss_rgb=losses[1], loss_dep=losses[2], loss_sem=losses[3], counts=counts, skip=skip, _keep=bk_dev)
    if sync:
        estimator.last_sampling = {"n_marched": int(c[0])}
        out.update(n_rendering_samples=int(c[1]), n_marched=int(c[0]))
    else:
        pool = st.setdefault("pinned", [])                         # (pinned host buffer, event) pairs are recycled: at most 3 are in flight
        host, ev = pool.pop() if pool else (torch.zeros(5, dtype=torch.int64).pin_memory(), torch.cuda.Event())
        host[4] = 0                                                 # (slot 4: the step's final skip flag, filled by train_step behind the optimizer's guard)
        if _defer_report:
            out["_report"] = (host, ev, R)                          # train_step: the optimizer call writes counters + final skip flag into `host`, no copy on the stream
        else:
            host[:4].copy_(counts, non_blocking=True)
            ev.record(torch.cuda.current_stream(dev))
            st["pending"].append((host, ev, R))
        estimator.last_sampling = {"n_marched": counts[0]}
        out.update(n_rendering_samples=counts[1], n_marched=counts[0])
    return out


def train_step(radiance_field, estimator, optimizer, rays: Rays, pixels, dep, sem, render_bkgd, step: int,
               near_plane=0.1, render_step_size=1e-3, cone_angle=0.004, alpha_thre=0.01, occ_thre=1e-3, scheduler=None,
               data_parallel_group=None, data_parallel=False, fused=True, sync=True, stratified=None, deterministic=False, presampled=None, seed=None):
    """One model's training iteration exactly as scripts/pipeline.py:447-532 sequences it: occupancy refresh every 16th
    step (:447-470), train render (:472-489), loss 10*smoothL1(rgb) + smoothL1(depth)/5 + CE(sem)/2 (:506-511),
    backward (:518), NaN-gradient guard (:520-529), optimizer and scheduler step (:531-532).  `data_parallel=True`
    averages the gradients (and sums the skip flags) over the ranks of `data_parallel_group` before the guard (each rank holds
    a slice of the rays; every rank reaches the collective every step, whatever its own batch produced).
    `fused=True` (default) runs render + loss + backward as one C call (`fused_forward_backward`); `fused=False` goes through
    the differentiable Python surface (`render_image_with_occgrid_with_depth_guide` + torch losses + autograd), which is the
    same kernels call by call.

    With `optim.FusedAdam` the decision "skip this iteration" (no surviving sample, non-finite gradient) is taken on the
    device and the iteration is enqueued without waiting for the GPU:
      sync=True   (default) ONE host round trip, behind the enqueued optimizer call (the reference has one per parameter plus two inside the sampler):
                  returns n_rendering_samples as an int and skipped as a bool, steps the scheduler only if the optimizer stepped.
      sync=False  no round trip at all: n_rendering_samples and skipped are device tensors; the scheduler advances every call, and a step whose skip
                  flag arrives raised (one or two calls later, with its counts) gives its scheduler step back: the schedule counts optimizer updates.
    Any other optimizer is stepped from the host after reading the flag (sync=True only).
    `stratified` (fused path): None = jitter the near planes as the reference does in training mode (occ_grid.py:187-189);
    False = no jitter (reproducible sample sets: parity tests).  `deterministic=True` (fused path): gradients are accumulated in an
    order-independent way (64-bit fixed-point integer atomics for the hash table, ordered partial sums for the weights): with a
    seeded torch generator the whole training run is bitwise reproducible.
    `seed` (fused path): the Philox key of the near-plane jitter (None: torch's CPU generator's next draw).
    `presampled` (fused path): the `presample()` token of THIS batch — its march ran beside the previous iteration; ignored (the step marches itself,
    with the token's jitter seed) if the occupancy refresh below or other options made it stale.
    Returns dict(loss, loss_rgb, loss_dep, loss_sem as device tensors, n_rendering_samples, skipped)."""
    import torch.nn.functional as F
    from .optim import FusedAdam, count_nan_gradients
    radiance_field.train()
    estimator.train()
    if presampled is not None and presampled.keep is not None and step % REFRESH_EVERY == 0:
        # only a step that refreshes the grid (update_every_n_steps: step % 16 == 0) needs the march finished before it starts: the refresh must not rewrite the
        # grid under it.  Every other step leaves the wait to mnf_train_step, which skips it when hipEventQuery says the march is done (a cross-queue wait costs
        # the stream ~18 us even then; ADVICE r04: waiting here unconditionally made that shortcut dead).
        presampled.wait(rays.origins.device)
    device_guard = isinstance(optimizer, FusedAdam)
    if not sync and not (fused and device_guard):
        raise ValueError("train_step(sync=False) needs fused=True and optim.FusedAdam (the skip decision is taken on the device)")

    occ_eval_fn = NA.FieldDensityOcc(radiance_field, render_step_size)     # pipeline.py:376-378; one fused C call per refresh

    estimator.update_every_n_steps(step=step, occ_eval_fn=occ_eval_fn, occ_thre=occ_thre, n=REFRESH_EVERY)
    out = None
    if fused and sync and device_guard and not data_parallel:
        # The reference's loop form (n_rendering_samples and the skip decision on the host after every iteration) with ONE host round trip and no bubble on the GPU: render +
        # loss + backward AND the guarded optimizer call are enqueued first, then the host waits once for the five words the optimizer's step-count kernel wrote into pinned
        # memory.  (Rounds 1-3 read the counts between backward and optimizer — the GPU idled through a round trip and the optimizer's launch — and the flag after it.)
        # A step beyond its sample bounds has raised the flag on the device: its optimizer call changed nothing, the bounds grow and the step is repeated.
        st = _train_state(radiance_field)
        dev = rays.origins.device
        if seed is None and (presampled is None or presampled.keep is None):
            seed = int(torch.randint(0, 2 ** 62, (1,)).item())          # one draw per iteration, whatever the number of attempts
        c = None
        for attempt in range(4):
            out = fused_forward_backward(radiance_field, estimator, rays, pixels, dep, sem, render_bkgd, near_plane=near_plane,
                                         render_step_size=render_step_size, cone_angle=cone_angle, alpha_thre=alpha_thre, sync=False,
                                         stratified=stratified, deterministic=deterministic, presampled=presampled if attempt == 0 else None, seed=seed,
                                         _defer_report=True)
            if out is None:
                break
            if presampled is not None and attempt == 0:
                seed = presampled.seed
            host, ev, _ = out["_report"]
            optimizer.step(skip=out["skip"], count_nonfinite=True, report=(out["counts"], host))
            if not optimizer.reported:
                host[:4].copy_(out["counts"], non_blocking=True)
                host[4:5].copy_(out["skip"].reshape(1).to(torch.int64), non_blocking=True)
            ev.record(torch.cuda.current_stream(dev))
            ev.synchronize()                                            # the iteration's one host round trip
            c = host.tolist()
            st.setdefault("pinned", []).append((host, ev))
            _check_status(c[3])
            if c[3] & _ST_ROW:
                out = None                                              # a ray longer than a scratch row: the autograd path below (the flagged optimizer call changed nothing)
                break
            if c[3] & (_ST_MARCHED | _ST_KEPT):
                _grow_caps(st, c[0], c[1], rays.origins.reshape(-1, 3).shape[0], carry=False)
                continue
            break
        else:
            raise L.MnfError("train_step: sample bounds kept growing")
        if out is not None:
            skipped = c[4] > 0
            estimator.last_sampling = {"n_marched": int(c[0])}
            if not skipped and scheduler is not None:
                scheduler.step()
            if int(c[1]) == 0:
                return dict(loss=None, n_rendering_samples=0, skipped=True)
            return dict(loss=out["loss"].detach(), loss_rgb=out["loss_rgb"].detach(), loss_dep=out["loss_dep"].detach(), loss_sem=out["loss_sem"].detach(),
                        n_rendering_samples=int(c[1]), skipped=skipped)
        fused = False
    if fused:
        out = fused_forward_backward(radiance_field, estimator, rays, pixels, dep, sem, render_bkgd, near_plane=near_plane,
                                     render_step_size=render_step_size, cone_angle=cone_angle, alpha_thre=alpha_thre, sync=sync,
                                     stratified=stratified, deterministic=deterministic, presampled=presampled, seed=seed,
                                     _defer_report=(not sync and device_guard))
    if out is not None:
        n_rendering_samples = out["n_rendering_samples"]
        loss, loss_rgb, loss_dep, loss_sem = out["loss"], out["loss_rgb"], out["loss_dep"], out["loss_sem"]
        skip = out["skip"]                         # device flag: raised by the C call for a step without samples
    else:
        # the autograd route (fused=False, or the fused call handed the batch over: more than four occupancy levels / a ray past the sampler's scratch row).  The
        # reference's signature has no jitter switch (it follows radiance_field.training, occ_grid.py:187-189): `stratified=False` is honoured by rendering in eval mode
        no_jitter = stratified is False and radiance_field.training
        if no_jitter:
            radiance_field.eval()
        try:
            rgb, acc, depth, semantic, n_rendering_samples = render_image_with_occgrid_with_depth_guide(
                radiance_field, estimator, rays, near_plane=near_plane, render_step_size=render_step_size, render_bkgd=render_bkgd,
                cone_angle=cone_angle, alpha_thre=alpha_thre, depth=dep)
        finally:
            if no_jitter:
                radiance_field.train()
        dev = rays.origins.device
        skip = torch.zeros((), dtype=torch.int32, device=dev)
        optimizer.zero_grad()
        if n_rendering_samples == 0:
            # pipeline.py:491: `continue`.  With data_parallel the collective below still has to be reached by this rank:
            # it contributes zero gradients and a raised skip flag.
            skip += 1
            for p_ in radiance_field.parameters():
                if p_.numel():
                    p_.grad = torch.zeros_like(p_)
            loss = loss_rgb = loss_dep = loss_sem = torch.zeros((), device=dev)
        else:
            loss_rgb = F.smooth_l1_loss(rgb, pixels)
            loss_dep = F.smooth_l1_loss(depth, dep.unsqueeze(1))
            loss_sem = F.cross_entropy(semantic, sem)
            loss = loss_rgb * 10 + loss_dep / 5 + loss_sem / 2
            loss.backward()
    if data_parallel:
        allreduce_gradients(radiance_field.parameters(), data_parallel_group, skip)
    if device_guard:
        # pipeline.py:520-532: the non-finite count is added to the same flag and the update leaves everything untouched when it is
        # raised — one C call for guard + three updates + handle refresh when the optimizer is bound to the field (`bind_field`)
        rep_ = out.get("_report") if out is not None else None
        optimizer.step(skip=skip, count_nonfinite=True, report=(out["counts"], rep_[0]) if rep_ is not None else None)
        if not sync:
            st = _train_state(radiance_field)
            # the step's counters and its FINAL skip flag (the optimizer's non-finite guard included) travel to pinned host memory; when they arrive — one or two
            # calls later — a skipped step takes back the scheduler step it was given here: the schedule counts optimizer updates, as pipeline.py:491 / :520-532 do
            if rep_ is not None:
                host, ev, r_ = rep_
                if not optimizer.reported:                         # (an optimizer not bound to the field: the copies the bound one's step-count kernel replaces)
                    host[:4].copy_(out["counts"], non_blocking=True)
                    host[4:5].copy_(skip.reshape(1).to(torch.int64), non_blocking=True)
                ev.record(torch.cuda.current_stream(rays.origins.device))
                st["pending"].append((host, ev, r_))
            if scheduler is not None:
                if st.get("sched_debt", 0) > 0:
                    st["sched_debt"] -= 1
                else:
                    scheduler.step()
            return dict(loss=loss.detach(), loss_rgb=loss_rgb.detach(), loss_dep=loss_dep.detach(), loss_sem=loss_sem.detach(),
                        n_rendering_samples=n_rendering_samples, skipped=skip)
        skipped = bool(skip.item() > 0)                                        # one host round trip per iteration
    else:
        count_nan_gradients(radiance_field.parameters(), out=skip)             # pipeline.py:520-529
        skipped = bool(skip.item() > 0)
        if skipped:
            optimizer.zero_grad()
        else:
            optimizer.step()
    if not skipped and scheduler is not None:
        scheduler.step()
    if isinstance(n_rendering_samples, int) and n_rendering_samples == 0:
        return dict(loss=None, n_rendering_samples=0, skipped=True)
    return dict(loss=loss.detach(), loss_rgb=loss_rgb.detach(), loss_dep=loss_dep.detach(), loss_sem=loss_sem.detach(),
                n_rendering_samples=n_rendering_samples, skipped=skipped)


_ENSEMBLE_STREAMS = {}


def ensemble_streams(device, n: int):
    """The streams an ensemble's members train on: the caller's current stream for member 0, one cached extra stream per further member (created
    once per device and kept: the platform dislikes processes that keep creating streams)."""
    device = torch.device(device)
    extra = _ENSEMBLE_STREAMS.setdefault(device, [])
    while len(extra) < n - 1:
        extra.append(torch.cuda.Stream(device))
    return [torch.cuda.current_stream(device)] + extra[:max(n - 1, 0)]


def train_step_ensemble(members, batches, step: int, **kw):
    """One iteration of the reference's ensemble (scripts/pipeline.py:398-412 trains the members ONE AFTER THE OTHER inside every iteration; they are
    independent models): every member's `train_step(sync=False)` is enqueued on a stream of its own, so that the latency-bound phases of one member's
    step (the ray marcher, the small launches, the tail of each kernel) run beside the other member's kernels — at the reference yaml's 2000 rays a
    single step cannot fill 256 compute units (measured: 1.55 -> 1.15 ms per member step with two members, profiles/r04_exp_ensemble.txt).
    `members`: list of (radiance_field, estimator, optimizer[, scheduler]) with `optim.FusedAdam` optimizers (the skip decision is taken on the device);
    `batches`: one (rays, pixels, dep, sem, render_bkgd) per member.  Inputs may have been produced on the caller's stream; on return the caller's stream
    waits for every member.  Results are those of the members stepped one after the other (they share nothing).  Returns the members' result dicts."""
    assert len(members) == len(batches) and len(members) >= 1
    tokens = kw.pop("presampled", None) or [None] * len(members)      # one `presample()` token per member (made on the caller's stream), or None
    dev = batches[0][0].origins.device
    streams = ensemble_streams(dev, len(members))
    cur = streams[0]
    ready = torch.cuda.Event()
    ready.record(cur)
    outs = []
    for m, (member, (rays, pixels, dep, sem, bkgd)) in enumerate(zip(members, batches)):
        field, est, opt = member[:3]
        sched = member[3] if len(member) > 3 else None
        s_ = streams[m]
        if m:
            s_.wait_event(ready)                       # whatever produced this member's inputs on the caller's stream comes first
            for t in (rays.origins, rays.viewdirs, pixels, dep, sem, bkgd) + tuple((tokens[m].keep or ())[:3] if tokens[m] is not None else ()):
                if isinstance(t, torch.Tensor) and t.is_cuda:
                    t.record_stream(s_)                 # the caching allocator must not hand the inputs' memory out again before this stream has read it
        with torch.cuda.stream(s_):
            outs.append(train_step(field, est, opt, rays, pixels, dep, sem, bkgd, step=step, scheduler=sched, sync=False, presampled=tokens[m], **kw))
    for s_ in streams[1:]:
        cur.wait_stream(s_)
    return outs


# ------------------------------------------------------------------ habitat_to_data.py
def pose_to_c2w(p: np.ndarray) -> np.ndarray:
    """habitat_to_data.py:444-451: xyz + quaternion (x, y, z, w) -> 4x4 camera-to-world (float64 on the host)."""
    from scipy.spatial.transform import Rotation as R
    pose = np.eye(4)
    pose[:3, :3] = R.from_quat(p[3:]).as_matrix()
    pose[:3, 3] = p[:3]
    return pose


def subsample_indices(n_total: int, n_keep: int) -> np.ndarray:
    """habitat_to_data.py:462-467 (np.round = half-to-even, evaluated in float64 on the host)."""
    return np.round(np.linspace(0, n_total - 1, n_keep)).astype(np.int64)


@torch.no_grad()
def generate_image_rays(pose, width, height, K, device, pix_idx=None):
    """Dataset.generate_image_rays (habitat_to_data.py:274-301) for `pose` [V,4,4] / [V,3,4]; K is the 3x3
    intrinsics the reference builds (only K[0,0] is used, fx == fy, principal point = image centre).
    With `pix_idx` only those flat pixel indices are generated (the reference builds all W*H rays and then indexes)."""
    pose = torch.as_tensor(np.asarray(pose.detach().cpu()) if isinstance(pose, torch.Tensor) else np.asarray(pose), dtype=torch.float32)
    c2w = pose[:, :3, :4].contiguous().to(device)
    L.require_gpu(c2w)
    K = np.asarray(K.detach().cpu()) if isinstance(K, torch.Tensor) else np.asarray(K)
    if abs(K[0, 2] - width / 2) > 1e-9 or abs(K[1, 2] - height / 2) > 1e-9 or abs(K[0, 0] - K[1, 1]) > 1e-9:
        raise NotImplementedError("generate_image_rays supports the reference intrinsics (fx == fy, centred principal point)")
    V = c2w.shape[0]
    idx_t = None
    n_pix = width * height
    if pix_idx is not None:
        idx_t = torch.as_tensor(np.asarray(pix_idx), dtype=torch.int64).to(device)
        n_pix = idx_t.shape[0]
    origins = torch.empty(V, n_pix, 3, device=device)
    viewdirs = torch.empty(V, n_pix, 3, device=device)
    L.launch(L.load_library().mnf_generate_rays, L.ptr(c2w), V, width, height, float(np.float32(K[0, 0])), L.ptr(idx_t), n_pix,
                                               L.ptr(origins), L.ptr(viewdirs))
    if V == 1:
        return Rays(origins=origins[0], viewdirs=viewdirs[0])
    return Rays(origins=origins, viewdirs=viewdirs)


def _pose_rays(poses, width, height, focal, scale, device):
    c2w = np.stack([pose_to_c2w(np.asarray(p, np.float64)) for p in poses]).astype(np.float32)
    h, w = int(height * scale), int(width * scale)
    idx = subsample_indices(width * height, h * w)
    K = np.array([[focal, 0.0, width / 2], [0.0, focal, height / 2], [0.0, 0.0, 1.0]])
    rays = generate_image_rays(torch.from_numpy(c2w), width, height, K, device, idx)
    V = c2w.shape[0]
    return rays.origins.reshape(V * h * w, 3), rays.viewdirs.reshape(V * h * w, 3), h, w


def _host_stacks_f64(outputs):
    """The float64 host arrays the reference's drivers return (habitat_to_data.py:376-409, :497-544: `np.zeros(...)` stacks filled from `.cpu().numpy()`), made
    with ONE device-to-host transfer: every output is widened to float64 on the device and lands in one pinned block (torch's caching host allocator), one stream
    synchronisation, and the returned arrays are views of that ONE block, which they keep alive (a caller that retains one of them for long retains all 111 MB of a
    640 x 640 pose: copy it out).  Six pageable `.double().cpu()` copies cost a 640 x 640 view more than its render (27.5 against 18.4 ms per pose, round 5).
    Round 6 tried the other order — transfer the float32 outputs (55 MB) and widen on the host with torch's CPU threads: 1.39 against 2.12 ms for the hand-over alone
    (profiles/r06_host_stacks.txt; numpy's single-threaded astype 11.6 ms), but inside the drivers the difference drowns in the thread pool's jitter (15.2-17.0 against
    15.6-16.4 ms per 640 x 640 pose, the 40-pose probabilistic call 19.8-23.2 against 19.7-20.8: profiles/r06_pose_drivers.txt) — not kept."""
    total = sum(int(np.prod(shape)) for _, shape in outputs)
    host = torch.empty(total, dtype=torch.float64, pin_memory=True)
    off = 0
    for t, shape in outputs:
        n = int(np.prod(shape))
        host[off:off + n].copy_(t.reshape(-1), non_blocking=True)
        off += n
    torch.cuda.current_stream(outputs[0][0].device).synchronize()
    arrays, off = [], 0
    for _, shape in outputs:
        n = int(np.prod(shape))
        arrays.append(host[off:off + n].numpy().reshape(shape))
        off += n
    return tuple(arrays)


@torch.no_grad()
def render_image_from_pose(radiance_field, estimator, poses, width, height, focal, near_plane, render_step_size, scale,
                           cone_angle, alpha_thre, downsample, device="cuda:0"):
    """Dataset.render_image_from_pose (habitat_to_data.py:304-411): all poses rendered in one batched call;
    returns host float64 arrays (images [P,h,w,3], depths [P,h,w], accs [P,h,w], sems [P,h,w,C])."""
    poses = np.asarray(poses)
    o, d, h, w = _pose_rays(poses, width, height, focal, scale, device)
    r = render_views(radiance_field, estimator, o, d, h * w, 1024, near_plane=near_plane, render_step_size=render_step_size,
                     render_bkgd=torch.zeros(3), cone_angle=cone_angle, alpha_thre=alpha_thre, image_hw=(h, w), n_split=None)
    P, C = poses.shape[0], radiance_field.num_semantic_classes
    return _host_stacks_f64([(r["rgb"], (P, h, w, 3)), (r["depth"], (P, h, w)), (r["acc"], (P, h, w)), (r["sem"], (P, h, w, C))])


@torch.no_grad()
def render_probablistic_image_from_pose(radiance_field, estimator, poses, width, height, focal, near_plane,
                                        render_step_size, scale, cone_angle, alpha_thre, downsample, device="cuda:0"):
    """Dataset.render_probablistic_image_from_pose (habitat_to_data.py:413-549) ->
    (images, images_var, depths, depths_var, accs, sems) host float64 arrays."""
    poses = np.asarray(poses)
    o, d, h, w = _pose_rays(poses, width, height, focal, scale, device)
    r = render_views(radiance_field, estimator, o, d, h * w, 1024, near_plane=near_plane, render_step_size=render_step_size,
                     render_bkgd=torch.zeros(3), cone_angle=cone_angle, alpha_thre=alpha_thre, probabilistic=True, image_hw=(h, w), n_split=None)
    P, C = poses.shape[0], radiance_field.num_semantic_classes
    return _host_stacks_f64([(r["rgb"], (P, h, w, 3)), (r["rgb_var"], (P, h, w, 3)), (r["depth"], (P, h, w)), (r["depth_var"], (P, h, w)), (r["acc"], (P, h, w)),
                             (r["sem"], (P, h, w, C))])


# ------------------------------------------------------------------ scorer (pipeline.py:666-798)
@torch.no_grad()
def score_view_terms(rgb_var, depth_var, acc, sem):
    """Per-view predictive-information terms on the device.  Inputs are member-major stacks of the probabilistic
    renders of M ensemble members for the same V views of P pixels: rgb_var [M,V,P,3], depth_var [M,V,P],
    acc [M,V,P], sem [M,V,P,C] (fp32, GPU).  Returns [V,4] float64: rgb, depth, semantic, occupancy (un-weighted)."""
    L.require_gpu(rgb_var, depth_var, acc, sem)
    M, V, P, C = sem.shape
    rv, dv, ac, sm = (L.contig(t, torch.float32) for t in (rgb_var, depth_var, acc, sem))
    terms = torch.empty(V, 4, dtype=torch.float64, device=sem.device)
    L.launch(L.load_library().mnf_score_views, L.ptr(rv), L.ptr(dv), L.ptr(ac), L.ptr(sm), M, V, P, C, L.ptr(terms))
    return terms


def shard_views(n_views: int, world: int, rank: int):
    """Contiguous view slice of `rank`: (lo, hi, per) with per = ceil(V / world) rows reserved per rank."""
    per = (n_views + world - 1) // world
    return min(rank * per, n_views), min((rank + 1) * per, n_views), per


def gather_view_terms(terms_local: torch.Tensor, n_views: int, group=None) -> torch.Tensor:
    """One all-gather of the per-rank [per,4] float64 term blocks (RCCL over xGMI on GPUs, gloo on CPU tensors);
    returns the [V,4] terms on every rank.  With no process group it is the identity."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return terms_local[:n_views]
    world = dist.get_world_size(group)
    gathered = torch.empty(world * terms_local.shape[0], 4, dtype=terms_local.dtype, device=terms_local.device)
    all_gather_blocks(gathered, terms_local.contiguous(), group)
    return gathered[:n_views]


def all_gather_blocks(out: torch.Tensor, block: torch.Tensor, group=None) -> None:
    """`dist.all_gather_into_tensor(out, block)`: RCCL for device tensors; under the `gloo` backend (CPU test rigs, two processes sharing one GPU) device tensors
    are staged through the host, which is the only form gloo offers for them."""
    import torch.distributed as dist
    if block.is_cuda and dist.get_backend(group) == "gloo":
        host = torch.empty(out.shape, dtype=out.dtype)
        dist.all_gather_into_tensor(host, block.cpu(), group=group)
        out.copy_(host)
    else:
        dist.all_gather_into_tensor(out, block, group=group)


def trajectory_score(terms: torch.Tensor) -> torch.Tensor:
    """pipeline.py:775-781: rgb + depth + 3*sem + 2*occ, averaged over the views."""
    return (terms[:, 0] + terms[:, 1] + 3 * terms[:, 2] + 2 * terms[:, 3]).mean()


LAST_SCORE_TOTALS = []


@torch.no_grad()
def score_views(radiance_fields, estimators, poses, width, height, focal, near_plane, render_step_size, scale, cone_angle,
                alpha_thre, device="cuda:0", group=None):
    """Predictive information of candidate views, sharded over the ranks of `group` (pipeline.py:666-798 +
    SURVEY.md §8e): every rank renders its contiguous slice of `poses` with every ensemble member, reduces to
    per-view terms on the device, and one all-gather of [V,4] float64 gives every rank all terms.
    Returns (terms [V,4] float64 on device, score = mean_v(t0 + t1 + 3 t2 + 2 t3))."""
    import torch.distributed as dist
    poses = np.asarray(poses)
    V = poses.shape[0]
    world, rank = 1, 0
    if group is not False and dist.is_available() and dist.is_initialized():       # group=False: all views on this rank, no exchange
        world, rank = dist.get_world_size(group), dist.get_rank(group)
    lo, hi, per = shard_views(V, world, rank)
    terms_local = torch.zeros(per, 4, dtype=torch.float64, device=device)
    if hi > lo:
        o, d, h, w = _pose_rays(poses[lo:hi], width, height, focal, scale, device)
        n = hi - lo
        # every ensemble member renders the same rays: the members advance side by side as render jobs of one call (four jobs in
        # flight — the caller's stream + the library's three shared side streams — is the measured optimum for these small views,
        # profiles/r03_hw_queues.txt: two members are cut into two groups of views each, one member into four; rays stay in row-major
        # march order — the 8x8-block order of full images buys nothing on 64x64 sub-sampled views: profiles/r03_split_experiment.txt)
        M = len(radiance_fields)
        outs = _render_jobs([(rf, est, o, d) for rf, est in zip(radiance_fields, estimators)], h * w, 1024, near_plane, 1e10, render_step_size,
                            torch.zeros(3), cone_angle, alpha_thre, 1e-4, True, 8, None, max(1, 4 // M))
        totals = [r["total"] for r in outs]
        rv = [r["rgb_var"].reshape(n, h * w, 3) for r in outs]; dv = [r["depth_var"].reshape(n, h * w) for r in outs]
        ac = [r["acc"].reshape(n, h * w) for r in outs]; sm = [r["sem"].reshape(n, h * w, -1) for r in outs]
        terms_local[:n] = score_view_terms(torch.stack(rv), torch.stack(dv), torch.stack(ac), torch.stack(sm))
        LAST_SCORE_TOTALS[:] = totals           # per member: device int64 [kept, evaluated] samples of this rank's views (measurement aid)
    terms = terms_local[:V] if group is False else gather_view_terms(terms_local, V, group)
    return terms, trajectory_score(terms)


@torch.no_grad()
def score_poses(radiance_fields, estimators, poses, width, height, focal, near_plane, render_step_size, scale, cone_angle,
                alpha_thre, device="cuda:0"):
    """`score_views` on one rank as ONE C call (`mnf_score_poses`, csrc/trainstep.hip): poses -> the sub-sampled rays of every
    view -> probabilistic renders by every ensemble member -> per-view terms.  Returns (terms [V,4] float64, score)."""
    import ctypes
    lib = L.load_library()
    poses = np.asarray(poses)
    V, M = poses.shape[0], len(radiance_fields)
    c2w = torch.from_numpy(np.stack([pose_to_c2w(np.asarray(p, np.float64)) for p in poses]).astype(np.float32)[:, :3, :4].copy()).to(device)
    h, w = int(height * scale), int(width * scale)
    idx = torch.from_numpy(subsample_indices(width * height, h * w)).to(device)
    handles = (ctypes.c_void_p * M)(*[f._ensure_handle() for f in radiance_fields])
    grids = [_grid_levels(e) for e in estimators]
    bins = (ctypes.c_void_p * M)(*[g[0].data_ptr() for g in grids])
    bits = (ctypes.c_void_p * M)(*[e.bitgrid()[0].data_ptr() for e in estimators])
    opts = L.RenderOpts()
    opts.near_plane, opts.far_plane, opts.render_step_size = near_plane, 1e10, render_step_size
    opts.cone_angle, opts.alpha_thre, opts.early_stop_eps = cone_angle, alpha_thre, 1e-4
    opts.max_samples, opts.probabilistic, opts.rays_per_view, opts.sync_every = 1024, 1, h * w, 8
    opts.view_order = None                                # row-major march order, as `score_views`
    opts.n_levels = len(grids[0][1]) // 6
    C = radiance_fields[0].num_semantic_classes
    nbytes = int(lib.mnf_score_poses_workspace_bytes(M, V, h * w, C))
    ws = _workspace(torch.device(device), nbytes)
    terms = torch.empty(V, 4, dtype=torch.float64, device=device)
    res = grids[0][0].shape[1:]
    L.launch(lib.mnf_score_poses, handles, bins, bits, M, res[0], res[1], res[2], (ctypes.c_float * len(grids[0][1]))(*grids[0][1]), L.ptr(c2w), V, width,
             height, float(np.float32(focal)), L.ptr(idx), h * w, ctypes.byref(opts), L.ptr(terms), L.ptr(ws), nbytes)
    return terms, trajectory_score(terms)
