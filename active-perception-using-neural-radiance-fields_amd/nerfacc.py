"""Drop-in for the subset of the reference's nerfacc fork that scripts/pipeline.py reaches
(SURVEY.md §2.2), backed by libmi355nerf.so.  Names, argument order and error behaviour follow
the reference; file:line citations are relative to /root/reference.

  ray_aabb_intersect              perception/nerfacc/nerfacc/grid.py:13-51
  traverse_grids                  grid.py:93-192 (+ host protocol cuda/csrc/grid.cu:320-474)
  RayIntervals / RaySamples       data_specs.py:12-180
  pack_info                       pack.py:10-49
  exclusive_sum                   scan.py:57-97
  render_*_from_density           volrend.py:212-267, :315-365, :424-483
  accumulate_along_rays[_]        volrend.py:486-576
  OccGridEstimator                estimators/occ_grid.py:13-437
"""
from dataclasses import dataclass
from typing import Callable, List, Optional, Tuple, Union

import torch
from torch import Tensor

from . import _lib as L


# ------------------------------------------------------------------ data specs
@dataclass
class RaySamples:
    vals: Tensor
    packed_info: Optional[Tensor] = None
    ray_indices: Optional[Tensor] = None
    is_valid: Optional[Tensor] = None

    @property
    def device(self):
        return self.vals.device


@dataclass
class RayIntervals:
    vals: Tensor
    packed_info: Optional[Tensor] = None
    ray_indices: Optional[Tensor] = None
    is_left: Optional[Tensor] = None
    is_right: Optional[Tensor] = None

    @property
    def device(self):
        return self.vals.device


# ------------------------------------------------------------------ grid.py
@torch.no_grad()
def ray_aabb_intersect(rays_o: Tensor, rays_d: Tensor, aabbs: Tensor, near_plane: float = -float("inf"),
                       far_plane: float = float("inf"), miss_value: float = float("inf")):
    assert rays_o.ndim == 2 and rays_o.shape[-1] == 3
    assert rays_d.ndim == 2 and rays_d.shape[-1] == 3
    assert aabbs.ndim == 2 and aabbs.shape[-1] == 6
    L.require_gpu(rays_o, rays_d, aabbs)
    rays_o, rays_d, aabbs = L.contig(rays_o, torch.float32), L.contig(rays_d, torch.float32), L.contig(aabbs, torch.float32)
    n, m = rays_o.shape[0], aabbs.shape[0]
    t_mins = torch.empty((n, m), device=rays_o.device, dtype=torch.float32)
    t_maxs = torch.empty_like(t_mins)
    hits = torch.empty((n, m), device=rays_o.device, dtype=torch.bool)
    L.launch(L.load_library().mnf_ray_aabb_intersect, L.ptr(rays_o), L.ptr(rays_d), n, L.ptr(aabbs), m, near_plane, far_plane,
                                                    miss_value, L.ptr(t_mins), L.ptr(t_maxs), L.ptr(hits))
    return t_mins, t_maxs, hits


def _alloc_segments(cnts: Tensor, masks: bool, valid: bool):
    """RaySegmentsSpec::memalloc_data_from_chunk (include/data_spec.hpp:86-96): cumsum + one host sync."""
    cumsum = torch.cumsum(cnts, 0, dtype=torch.int64)
    n_edges = int(cumsum[-1].item()) if cnts.numel() else 0
    starts = cumsum - cnts
    dev = cnts.device
    vals = torch.zeros(n_edges, device=dev, dtype=torch.float32)
    ridx = torch.zeros(n_edges, device=dev, dtype=torch.int64)
    left = torch.zeros(n_edges, device=dev, dtype=torch.bool) if masks else None
    right = torch.zeros(n_edges, device=dev, dtype=torch.bool) if masks else None
    isvalid = torch.zeros(n_edges, device=dev, dtype=torch.bool) if valid else None
    return starts, vals, ridx, left, right, isvalid


@torch.no_grad()
def traverse_grids(rays_o: Tensor, rays_d: Tensor, binaries: Tensor, aabbs: Tensor,
                   near_planes: Optional[Tensor] = None, far_planes: Optional[Tensor] = None,
                   step_size: Optional[float] = 1e-3, cone_angle: Optional[float] = 0.0,
                   traverse_steps_limit: Optional[int] = None, over_allocate: Optional[bool] = False,
                   rays_mask: Optional[Tensor] = None, t_sorted: Optional[Tensor] = None,
                   t_indices: Optional[Tensor] = None, hits: Optional[Tensor] = None
                   ) -> Tuple[RayIntervals, RaySamples, Tensor]:
    L.require_gpu(rays_o, rays_d, binaries, aabbs)
    dev = rays_o.device
    if near_planes is None:
        near_planes = torch.zeros_like(rays_o[:, 0])
    if far_planes is None:
        far_planes = torch.full_like(rays_o[:, 0], float("inf"))
    if rays_mask is None:
        rays_mask = torch.ones_like(rays_o[:, 0], dtype=torch.bool)
    if traverse_steps_limit is None:
        traverse_steps_limit = -1
    if over_allocate:
        assert traverse_steps_limit > 0, "traverse_steps_limit must be set if over_allocate is True."
    if t_sorted is None or t_indices is None or hits is None:
        t_mins, t_maxs, hits = ray_aabb_intersect(rays_o, rays_d, aabbs)
        t_sorted, t_indices = torch.sort(torch.cat([t_mins, t_maxs], dim=-1), dim=-1)
    rays_o, rays_d = L.contig(rays_o, torch.float32), L.contig(rays_d, torch.float32)
    binaries_u8 = L.contig(binaries).view(torch.uint8) if binaries.dtype == torch.bool else L.contig(binaries, torch.uint8)
    aabbs = L.contig(aabbs, torch.float32)
    t_sorted, t_indices = L.contig(t_sorted, torch.float32), L.contig(t_indices, torch.int64)
    hits_c, mask_c = L.contig(hits), L.contig(rays_mask)
    near_planes, far_planes = L.contig(near_planes, torch.float32), L.contig(far_planes, torch.float32)
    n = rays_o.shape[0]
    n_grids, rx, ry, rz = binaries.shape
    term = torch.empty(n, device=dev, dtype=torch.float32)
    lib = L.load_library()

    def call(first_pass, iv, sm, mask, term_t):
        iv_vals, iv_r, iv_l, iv_rt, iv_st, iv_cnt = iv
        sm_vals, sm_r, sm_valid, sm_st, sm_cnt = sm
        L.launch(lib.mnf_traverse_grids, 
            L.ptr(rays_o), L.ptr(rays_d), L.ptr(mask), n, L.ptr(binaries_u8), L.ptr(aabbs), n_grids, rx, ry, rz,
            L.ptr(hits_c), L.ptr(t_sorted), L.ptr(t_indices), L.ptr(near_planes), L.ptr(far_planes),
            float(step_size), float(cone_angle), int(traverse_steps_limit), 1 if first_pass else 0,
            L.ptr(iv_vals), L.ptr(iv_r), L.ptr(iv_l), L.ptr(iv_rt), L.ptr(iv_st), L.ptr(iv_cnt),
            L.ptr(sm_vals), L.ptr(sm_r), L.ptr(sm_valid), L.ptr(sm_st), L.ptr(sm_cnt), L.ptr(term_t))

    if over_allocate:
        # grid.cu:364-404
        iv_cnt = torch.full((n,), traverse_steps_limit * 2, device=dev, dtype=torch.int64) * rays_mask
        sm_cnt = torch.full((n,), traverse_steps_limit, device=dev, dtype=torch.int64) * rays_mask
        iv_st, iv_vals, iv_r, iv_l, iv_rt, _ = _alloc_segments(iv_cnt, True, False)
        sm_st, sm_vals, sm_r, _, _, sm_valid = _alloc_segments(sm_cnt, False, True)
        call(False, (iv_vals, iv_r, iv_l, iv_rt, iv_st, iv_cnt), (sm_vals, sm_r, sm_valid, sm_st, sm_cnt), mask_c, term)
        iv_st = torch.cumsum(iv_cnt, 0) - iv_cnt
        sm_st = torch.cumsum(sm_cnt, 0) - sm_cnt
    else:
        # grid.cu:405-470: count pass, allocate, fill pass
        iv_cnt = torch.empty(n, device=dev, dtype=torch.int64)
        sm_cnt = torch.empty(n, device=dev, dtype=torch.int64)
        call(True, (None, None, None, None, None, iv_cnt), (None, None, None, None, sm_cnt), None, None)
        iv_st, iv_vals, iv_r, iv_l, iv_rt, _ = _alloc_segments(iv_cnt, True, False)
        sm_st, sm_vals, sm_r, _, _, sm_valid = _alloc_segments(sm_cnt, False, False)
        call(False, (iv_vals, iv_r, iv_l, iv_rt, iv_st, iv_cnt), (sm_vals, sm_r, None, sm_st, sm_cnt), None, term)
    intervals = RayIntervals(iv_vals, torch.stack([iv_st, iv_cnt], -1), iv_r, iv_l, iv_rt)
    samples = RaySamples(sm_vals, torch.stack([sm_st, sm_cnt], -1), sm_r, sm_valid)
    return intervals, samples, term


def _enlarge_aabb(aabb, factor: float) -> Tensor:
    center = (aabb[:3] + aabb[3:]) / 2
    extent = (aabb[3:] - aabb[:3]) / 2
    return torch.cat([center - extent * factor, center + extent * factor])


# ------------------------------------------------------------------ pack.py / scan.py / volrend.py
def _scratch(device, nbytes: int) -> Tensor:
    return torch.empty(max(int(nbytes), 8), dtype=torch.uint8, device=device)


@torch.no_grad()
def pack_info(ray_indices: Tensor, n_rays: Optional[int] = None) -> Tensor:
    """pack.py:10-38: [n_rays, 2] = (first sample, sample count) per ray, for ray indices in any order
    (`mnf_pack_info`: atomic counts, then a device prefix sum)."""
    assert ray_indices.dim() == 1, "ray_indices must be a 1D tensor with shape (n_samples)."
    if not ray_indices.is_cuda:
        raise NotImplementedError("Only support cuda inputs.")          # pack.py:48, same message
    if n_rays is None:
        n_rays = int(ray_indices.max().item()) + 1
    ri = L.contig(ray_indices, torch.int64)
    out = torch.empty((n_rays, 2), dtype=torch.int64, device=ri.device)
    lib = L.load_library()
    nbytes = 16 * n_rays + lib.mnf_scan_workspace_bytes(n_rays)
    ws = _scratch(ri.device, nbytes)
    L.launch(lib.mnf_pack_info, L.ptr(ri), ri.shape[0], n_rays, L.ptr(out), L.ptr(ws), nbytes)
    return out.to(ray_indices.dtype)


def exclusive_scan_counts(counts: Tensor, want_total: bool = False):
    """Chunk starts from chunk counts on the device (`RaySegmentsSpec::memalloc_data_from_chunk`, data_spec.hpp:86-96);
    optionally also the grand total as a device scalar."""
    counts = L.contig(counts, torch.int64)
    n = counts.shape[0]
    starts = torch.empty_like(counts)
    total = torch.zeros((), dtype=torch.int64, device=counts.device) if want_total else None
    lib = L.load_library()
    nbytes = lib.mnf_scan_workspace_bytes(n)
    ws = _scratch(counts.device, nbytes)
    L.launch(lib.mnf_exclusive_scan_i64, L.ptr(counts), n, L.ptr(starts), L.ptr(total), L.ptr(ws), nbytes)
    return (starts, total) if want_total else starts


@torch.no_grad()
def _select_visible(t_starts: Tensor, t_ends: Tensor, sigmas: Tensor, packed_info: Tensor, early_stop_eps: float, alpha_thre_dev: Tensor):
    """occ_grid.py:209-236 after the density pass: (ray_indices, t_starts, t_ends) of the samples with T >= early_stop_eps and alpha >= alpha_thre
    (`mnf_visible_samples`: count pass, prefix sum, write pass)."""
    lib, dev = L.load_library(), t_starts.device
    starts, cnts = (L.contig(x, torch.int64) for x in packed_info.to(torch.int64).unbind(-1))
    ts, te = L.contig(t_starts, torch.float32), L.contig(t_ends, torch.float32)
    n_rays = starts.shape[0]
    kept = torch.empty((n_rays,), device=dev, dtype=torch.int64)
    L.launch(lib.mnf_visible_samples, L.ptr(starts), L.ptr(cnts), n_rays, L.ptr(ts), L.ptr(te), L.ptr(sigmas), early_stop_eps, L.ptr(alpha_thre_dev),
             L.ptr(kept), None, None, None, None)
    kept_starts, total_t = exclusive_scan_counts(kept, want_total=True)
    total = int(total_t.item())
    o_ts, o_te = torch.empty((total,), device=dev), torch.empty((total,), device=dev)
    o_ray = torch.empty((total,), device=dev, dtype=torch.int64)
    if total:
        L.launch(lib.mnf_visible_samples, L.ptr(starts), L.ptr(cnts), n_rays, L.ptr(ts), L.ptr(te), L.ptr(sigmas), early_stop_eps, L.ptr(alpha_thre_dev),
                 None, L.ptr(kept_starts), L.ptr(o_ts), L.ptr(o_te), L.ptr(o_ray))
    return o_ray, o_ts, o_te


def pack_info_grouped(ray_indices: Tensor, n_rays: int) -> Tensor:
    """`pack_info` for ray indices grouped by ray (every output of `traverse_grids` / `sampling`, masked or not): same
    [n_rays, 2] result, from run boundaries instead of one atomic per sample."""
    ray_indices = ray_indices.contiguous()
    bounds = torch.zeros((2, n_rays), device=ray_indices.device, dtype=torch.int64)
    L.launch(L.load_library().mnf_run_bounds, L.ptr(ray_indices), ray_indices.shape[0], L.ptr(bounds[0]), L.ptr(bounds[1]))
    cnts = bounds[1] - bounds[0]
    return torch.stack([exclusive_scan_counts(cnts), cnts], dim=-1)


class _ExclusiveSum(torch.autograd.Function):
    """scan.py:206-229: forward packed exclusive sum; backward = reverse-direction scan of the grad."""

    @staticmethod
    def forward(ctx, chunk_starts, chunk_cnts, inputs):
        chunk_starts, chunk_cnts = chunk_starts.contiguous(), chunk_cnts.contiguous()
        inputs = inputs.contiguous()
        out = torch.empty_like(inputs)
        L.launch(L.load_library().mnf_exclusive_sum, L.ptr(chunk_starts), L.ptr(chunk_cnts), chunk_cnts.shape[0], L.ptr(inputs),
                                                   L.ptr(out), inputs.shape[0], 0)
        ctx.save_for_backward(chunk_starts, chunk_cnts)
        return out

    @staticmethod
    def backward(ctx, g):
        chunk_starts, chunk_cnts = ctx.saved_tensors
        g = g.contiguous()
        out = torch.empty_like(g)
        L.launch(L.load_library().mnf_exclusive_sum, L.ptr(chunk_starts), L.ptr(chunk_cnts), chunk_cnts.shape[0], L.ptr(g),
                                                   L.ptr(out), g.shape[0], 1)
        return None, None, out


def exclusive_sum(inputs: Tensor, packed_info: Optional[Tensor] = None) -> Tensor:
    if packed_info is None:
        return torch.cumsum(torch.cat([torch.zeros_like(inputs[..., :1]), inputs[..., :-1]], dim=-1), dim=-1)
    assert inputs.dim() == 1, "inputs must be flattened."
    assert packed_info.dim() == 2 and packed_info.shape[-1] == 2, "packed_info must be 2-D with shape (B, 2)."
    L.require_gpu(inputs, packed_info)
    chunk_starts, chunk_cnts = packed_info.to(torch.int64).unbind(dim=-1)
    return _ExclusiveSum.apply(chunk_starts, chunk_cnts, inputs)


def render_transmittance_from_density(t_starts, t_ends, sigmas, packed_info=None, ray_indices=None, n_rays=None,
                                      prefix_trans=None):
    if ray_indices is not None and packed_info is None:
        packed_info = pack_info(ray_indices, n_rays)
    sigmas_dt = sigmas * (t_ends - t_starts)
    alphas = 1.0 - torch.exp(-sigmas_dt)
    trans = torch.exp(-exclusive_sum(sigmas_dt, packed_info))
    if prefix_trans is not None:
        trans = trans * prefix_trans
    return trans, alphas


def render_weight_from_density(t_starts, t_ends, sigmas, packed_info=None, ray_indices=None, n_rays=None,
                               prefix_trans=None):
    trans, alphas = render_transmittance_from_density(t_starts, t_ends, sigmas, packed_info, ray_indices, n_rays, prefix_trans)
    return trans * alphas, trans, alphas


@torch.no_grad()
def render_visibility_from_density(t_starts, t_ends, sigmas, packed_info=None, ray_indices=None, n_rays=None,
                                   early_stop_eps: float = 1e-4, alpha_thre: float = 0.0, prefix_trans=None):
    """volrend.py:424-483; the packed path is one fused kernel (sigma*dt -> alpha, T, threshold)."""
    if ray_indices is not None and packed_info is None:
        packed_info = pack_info(ray_indices, n_rays)
    if packed_info is None:
        trans, alphas = render_transmittance_from_density(t_starts, t_ends, sigmas, prefix_trans=prefix_trans)
    else:
        L.require_gpu(t_starts, t_ends, sigmas, packed_info)
        starts, cnts = (x.contiguous() for x in packed_info.to(torch.int64).unbind(-1))
        ts, te, sg = L.contig(t_starts, torch.float32), L.contig(t_ends, torch.float32), L.contig(sigmas, torch.float32)
        pf = L.contig(prefix_trans, torch.float32)
        trans, alphas = torch.empty_like(sg), torch.empty_like(sg)
        L.launch(L.load_library().mnf_render_weight_from_density, L.ptr(starts), L.ptr(cnts), cnts.shape[0], L.ptr(ts), L.ptr(te),
                                                                L.ptr(sg), L.ptr(pf), sg.shape[0], None, L.ptr(trans),
                                                                L.ptr(alphas))
    vis = trans >= early_stop_eps
    if alpha_thre > 0:
        vis = vis & (alphas >= alpha_thre)
    return vis


class _Accumulate(torch.autograd.Function):
    """Packed branch of volrend.py:486-576 on the HIP kernels: forward scatter-add, backward = the two products autograd
    derives for `index_add_(0, ray_indices, weights[:, None] * values)`."""

    @staticmethod
    def forward(ctx, weights, values, ray_indices, outputs):
        w = L.contig(weights, torch.float32)
        v = None if values is None else L.contig(values, torch.float32)
        ri = L.contig(ray_indices, torch.int64)
        D = 1 if v is None else v.shape[-1]
        L.launch(L.load_library().mnf_accumulate_along_rays, L.ptr(w), L.ptr(v), L.ptr(ri), w.shape[0], D, L.ptr(outputs))
        ctx.save_for_backward(w, v if v is not None else w.new_empty(0), ri)
        ctx.has_values, ctx.D = v is not None, D
        ctx.mark_dirty(outputs)
        return outputs

    @staticmethod
    def backward(ctx, g_out):
        w, v, ri = ctx.saved_tensors
        v = v if ctx.has_values else None
        g_out = L.contig(g_out, torch.float32)
        g_w = torch.empty_like(w) if ctx.needs_input_grad[0] else None
        g_v = torch.empty_like(v) if (v is not None and ctx.needs_input_grad[1]) else None
        if g_w is not None or g_v is not None:
            L.launch(L.load_library().mnf_accumulate_along_rays_backward, L.ptr(w), L.ptr(v), L.ptr(ri), w.shape[0], ctx.D,
                     L.ptr(g_out), L.ptr(g_w), L.ptr(g_v))
        return g_w, g_v, None, g_out


def _check_accumulate_args(weights, values):
    if values is not None:
        assert values.dim() == weights.dim() + 1
        assert weights.shape == values.shape[:-1]


def accumulate_along_rays(weights: Tensor, values: Optional[Tensor] = None, ray_indices: Optional[Tensor] = None,
                          n_rays: Optional[int] = None) -> Tensor:
    """volrend.py:486-535 -> [n_rays, D] (packed samples) or [..., D] (batched samples)."""
    _check_accumulate_args(weights, values)
    if ray_indices is None:                                   # batched [..., S] samples: a plain sum over the sample axis
        return weights.sum(dim=-1, keepdim=True) if values is None else (weights.unsqueeze(-1) * values).sum(dim=-2)
    assert n_rays is not None, "n_rays must be provided"
    assert weights.dim() == 1, "weights must be flattened"
    L.require_gpu(weights, ray_indices)
    D = 1 if values is None else values.shape[-1]
    return _Accumulate.apply(weights, values, ray_indices, torch.zeros((n_rays, D), device=weights.device, dtype=torch.float32))


def accumulate_along_rays_(weights: Tensor, values: Optional[Tensor] = None, ray_indices: Optional[Tensor] = None,
                           outputs: Optional[Tensor] = None) -> None:
    """volrend.py:538-576: the in-place form (the test-time renderers' running accumulators)."""
    _check_accumulate_args(weights, values)
    D = 1 if values is None else values.shape[-1]
    if ray_indices is None:
        outputs.add_(weights.sum(dim=-1, keepdim=True) if values is None else (weights.unsqueeze(-1) * values).sum(dim=-2))
        return
    assert weights.dim() == 1, "weights must be flattened"
    assert outputs.dim() == 2 and outputs.shape[-1] == D, "outputs must be of shape (n_rays, D)"
    L.require_gpu(weights, ray_indices, outputs)
    if outputs.dtype != torch.float32 or not outputs.is_contiguous():
        raise L.MnfError("accumulate_along_rays_: outputs must be a contiguous float32 tensor")
    _Accumulate.apply(weights, values, ray_indices, outputs)


# ------------------------------------------------------------------ fused compositing (utils.py:362-461, volrend.py:20-161)
class _CompositeTrain(torch.autograd.Function):
    """Semantic volume rendering of packed samples after the field query (perception/models/utils.py:362-461; also
    `rendering`, volrend.py:20-161, with zero classes): weights (volrend.py:213-267), the accumulate_along_rays sums,
    background blend and depth normalisation as one HIP launch, with a hand-written adjoint in place of the autograd
    graph the reference records (csrc/composite_train.hip)."""

    @staticmethod
    def forward(ctx, chunk_starts, chunk_cnts, t_starts, t_ends, sigmas, rgbs, sems, bkgd):
        dev, R, N, C = sigmas.device, chunk_cnts.shape[0], sigmas.shape[0], sems.shape[-1]
        t_starts, t_ends = t_starts.contiguous().float(), t_ends.contiguous().float()
        sigmas, rgbs, sems = sigmas.contiguous().float(), rgbs.contiguous().float(), sems.contiguous().float()
        colors, semantics = torch.empty((R, 3), device=dev), torch.empty((R, C), device=dev)
        opacities, depths = torch.empty((R, 1), device=dev), torch.empty((R, 1), device=dev)
        weights, trans, alphas = torch.empty((N,), device=dev), torch.empty((N,), device=dev), torch.empty((N,), device=dev)
        L.launch(L.load_library().mnf_composite_train_forward, 
            L.ptr(chunk_starts), L.ptr(chunk_cnts), R, L.ptr(t_starts), L.ptr(t_ends), L.ptr(sigmas), L.ptr(rgbs), L.ptr(sems), C, N,
            L.ptr(bkgd), L.ptr(colors), L.ptr(opacities), L.ptr(depths), L.ptr(semantics), L.ptr(weights), L.ptr(trans),
            L.ptr(alphas))
        ctx.save_for_backward(chunk_starts, chunk_cnts, t_starts, t_ends, sigmas, rgbs, sems, weights, trans, opacities, depths)
        ctx.bkgd = bkgd
        ctx.mark_non_differentiable(weights, trans, alphas)
        return colors, opacities, depths, semantics, weights, trans, alphas

    @staticmethod
    def backward(ctx, g_rgb, g_acc, g_dep, g_sem, _gw, _gt, _ga):
        chunk_starts, chunk_cnts, t_starts, t_ends, sigmas, rgbs, sems, weights, trans, opacities, depths = ctx.saved_tensors
        R, N, C = chunk_cnts.shape[0], sigmas.shape[0], sems.shape[-1]
        d_sig, d_rgb, d_sem = torch.empty_like(sigmas), torch.empty_like(rgbs), torch.empty_like(sems)
        g = [None if t is None else t.contiguous().float() for t in (g_rgb, g_acc, g_dep, g_sem)]
        L.launch(L.load_library().mnf_composite_train_backward, 
            L.ptr(chunk_starts), L.ptr(chunk_cnts), R, L.ptr(t_starts), L.ptr(t_ends), L.ptr(sigmas), L.ptr(rgbs), L.ptr(sems), C, N,
            L.ptr(ctx.bkgd), L.ptr(weights), L.ptr(trans), L.ptr(opacities), L.ptr(depths), L.ptr(g[0]), L.ptr(g[1]), L.ptr(g[2]),
            L.ptr(g[3]), L.ptr(d_sig), L.ptr(d_rgb), L.ptr(d_sem))
        return None, None, None, None, d_sig, d_rgb, d_sem, None


def rendering(t_starts: Tensor, t_ends: Tensor, ray_indices: Optional[Tensor] = None, n_rays: Optional[int] = None,
              rgb_sigma_fn: Optional[Callable] = None, rgb_alpha_fn: Optional[Callable] = None,
              render_bkgd: Optional[Tensor] = None):
    """volrend.py:20-161 -> (colors [R,3], opacities [R,1], depths [R,1], extras), differentiable with respect to what
    `rgb_sigma_fn` returns.  Batched [R,S] samples (the BASELINE config-1 path) and packed samples grouped by ray both go
    through the fused compositing kernel; `rgb_alpha_fn` is not reachable from scripts/pipeline.py."""
    if ray_indices is not None:
        assert t_starts.shape == t_ends.shape == ray_indices.shape, \
            "Since nerfacc 0.5.0, t_starts, t_ends and ray_indices must have the same shape (N,). "
    if rgb_sigma_fn is None and rgb_alpha_fn is None:
        raise ValueError("At least one of `rgb_sigma_fn` and `rgb_alpha_fn` should be specified.")
    if rgb_sigma_fn is None:
        raise NotImplementedError("rendering(rgb_alpha_fn=...) is outside the hot path (pipeline.py only renders from densities)")
    L.require_gpu(t_starts, t_ends)
    dev = t_starts.device
    if t_starts.shape[0] != 0:
        rgbs, sigmas = rgb_sigma_fn(t_starts, t_ends, ray_indices)
    else:
        rgbs, sigmas = torch.empty((0, 3), device=dev), torch.empty((0,), device=dev)
    assert rgbs.shape[-1] == 3, "rgbs must have 3 channels, got {}".format(rgbs.shape)
    assert sigmas.shape == t_starts.shape, "sigmas must have shape of (N,)! Got {}".format(sigmas.shape)
    if ray_indices is None:                      # [R,S]: ray r owns the S consecutive samples r*S ..
        R, S = t_starts.shape
        starts = torch.arange(R, device=dev, dtype=torch.int64) * S
        cnts = torch.full((R,), S, device=dev, dtype=torch.int64)
    else:
        assert n_rays is not None, "n_rays must be provided"
        packed = pack_info_grouped(ray_indices, n_rays)
        starts, cnts = packed[:, 0].contiguous(), packed[:, 1].contiguous()
    bk = None if render_bkgd is None else render_bkgd.to(device=dev, dtype=torch.float32).reshape(3).contiguous()
    flat = sigmas.reshape(-1)
    colors, opacities, depths, _, weights, trans, alphas = _CompositeTrain.apply(
        starts, cnts, t_starts.reshape(-1), t_ends.reshape(-1), flat, rgbs.reshape(-1, 3), flat.new_empty((flat.shape[0], 0)), bk)
    shape = t_starts.shape
    extras = {"weights": weights.view(shape), "alphas": alphas.view(shape), "trans": trans.view(shape), "sigmas": sigmas, "rgbs": rgbs}
    return colors, opacities, depths, extras


# ------------------------------------------------------------------ estimators/occ_grid.py
class FieldDensityOcc:
    """The `occ_eval_fn` that scripts/pipeline.py:376-378 builds: `radiance_field.query_density(x) * render_step_size`.
    Passing this object (instead of an opaque closure) lets `OccGridEstimator._update` run the whole refresh as ONE C call
    (`mnf_update_occupancy`: cell pick, density query, EMA update and re-binarisation chained on the device)."""

    def __init__(self, radiance_field, render_step_size: float):
        self.field, self.scale = radiance_field, float(render_step_size)

    def __call__(self, x):
        return self.field.query_density(x) * self.scale


class OccGridEstimator(torch.nn.Module):
    """Occupancy-grid estimator with the buffers and methods of estimators/occ_grid.py:13-437 (`resolution`, `aabbs`,
    `occs`, `binaries`; `sampling`, `update_every_n_steps`, `mark_invisible_cells`, `_update`).  The grid refresh and the
    marching run on the device kernels of csrc/occupancy.hip and csrc/march.hip; besides the reference's byte grid the
    estimator keeps the bit-packed copy the marchers stage in LDS (`bitgrid()`)."""

    DIM: int = 3

    def __init__(self, roi_aabb: Union[List[int], Tensor], resolution: Union[int, List[int], Tensor] = 128,
                 levels: int = 1, **kwargs) -> None:
        super().__init__()
        if "contraction_type" in kwargs:
            raise ValueError("`contraction_type` is not supported anymore for nerfacc >= 0.4.0.")   # occ_grid.py:34-37
        res = resolution if isinstance(resolution, Tensor) else torch.tensor(
            [resolution] * self.DIM if isinstance(resolution, int) else list(resolution), dtype=torch.int32)
        box = roi_aabb if isinstance(roi_aabb, Tensor) else torch.tensor(list(roi_aabb), dtype=torch.float32)
        assert res.shape[0] == self.DIM, f"Invalid shape: {res}!"
        assert box.shape[0] == 2 * self.DIM, f"Invalid shape: {box}!"
        box = box.detach().cpu().float()
        self.levels = levels
        self.cells_per_lvl = int(res.prod().item())
        # level i covers the region of interest enlarged 2^i times about its centre (occ_grid.py:58-62)
        self.register_buffer("resolution", res)
        self.register_buffer("aabbs", torch.stack([_enlarge_aabb(box, 2 ** lvl) for lvl in range(levels)]))
        self.register_buffer("occs", torch.zeros(levels * self.cells_per_lvl))
        self.register_buffer("binaries", torch.zeros([levels] + [int(r) for r in res], dtype=torch.bool))
        self._bits, self._bits_key, self._occ_ws = None, None, None

    @property
    def grid_coords(self) -> Tensor:
        """[cells, 3] integer coordinates of the flat cell ids ('ij' order, z fastest: occ_grid.py:440-455); computed on
        demand — the kernels derive them from the flat id."""
        return self._cell_coords(torch.arange(self.cells_per_lvl, device=self.occs.device))

    @property
    def grid_indices(self) -> Tensor:
        return torch.arange(self.cells_per_lvl, device=self.occs.device)

    def _cell_coords(self, ids: Tensor) -> Tensor:
        ry, rz = int(self.resolution[1]), int(self.resolution[2])
        return torch.stack([ids // (ry * rz), (ids // rz) % ry, ids % rz], dim=-1)

    def bitgrid(self) -> Tensor:
        """int32 [levels, ceil(cells/32)]: bit (c & 31) of word (c >> 5) = binaries.flatten()[c] per level.  Maintained by
        `_update`; rebuilt here whenever `binaries` was assigned or edited by someone else (viewer, checkpoint load)."""
        b = self.binaries
        L.require_gpu(b)
        key = (b.data_ptr(), b._version, tuple(b.shape), str(b.device))
        if self._bits_key != key:
            bu = b.contiguous().view(torch.uint8) if b.dtype == torch.bool else b.to(torch.uint8).contiguous()
            words = (self.cells_per_lvl + 31) // 32
            if self._bits is None or self._bits.device != b.device or self._bits.shape != (self.levels, words):
                self._bits = torch.empty((self.levels, words), dtype=torch.int32, device=b.device)
            L.launch(L.load_library().mnf_pack_bitgrid, L.ptr(bu), self.cells_per_lvl, self.levels, L.ptr(self._bits))
            self._bits_key = key
        return self._bits

    @property
    def device(self):
        return self.occs.device

    def aabb_host(self, lvl: int = 0):
        """Six host floats of `aabbs[lvl]`, cached on this module and refreshed whenever the buffer is replaced, moved or
        written in place (keyed on the tensor itself, not on the module's identity)."""
        a = self.aabbs
        key = (a.data_ptr(), a._version, str(a.device), tuple(a.shape))
        if getattr(self, "_aabb_host_key", None) != key:
            self._aabb_host_vals = [[float(x) for x in row] for row in a.detach().cpu().tolist()]
            self._aabb_host_key = key
        return self._aabb_host_vals[lvl]

    @torch.no_grad()
    def sampling(self, rays_o: Tensor, rays_d: Tensor, sigma_fn: Optional[Callable] = None,
                 alpha_fn: Optional[Callable] = None, near_plane: float = 0.0, far_plane: float = 1e10,
                 t_min: Optional[Tensor] = None, t_max: Optional[Tensor] = None, render_step_size: float = 1e-3,
                 early_stop_eps: float = 1e-4, alpha_thre: float = 0.0, stratified: bool = False,
                 cone_angle: float = 0.0, depth: Optional[Tensor] = None) -> Tuple[Tensor, Tensor, Tensor]:
        """occ_grid.py:80-238 (`depth` is accepted and ignored, as in the reference fork)."""
        near_planes = torch.full_like(rays_o[..., 0], fill_value=near_plane)
        far_planes = torch.full_like(rays_o[..., 0], fill_value=far_plane)
        if t_min is not None:
            near_planes = torch.clamp(near_planes, min=t_min)
        if t_max is not None:
            far_planes = torch.clamp(far_planes, max=t_max)
        if stratified:
            near_planes += torch.rand_like(near_planes) * render_step_size
        packed = self._sample_single_pass(rays_o, rays_d, near_planes, far_planes, render_step_size, cone_angle)
        if packed is not None:
            ray_indices, t_starts, t_ends, packed_info = packed
        else:
            intervals, samples, _ = traverse_grids(rays_o, rays_d, self.binaries, self.aabbs, near_planes=near_planes,
                                                   far_planes=far_planes, step_size=render_step_size, cone_angle=cone_angle)
            t_starts = intervals.vals[intervals.is_left]
            t_ends = intervals.vals[intervals.is_right]
            ray_indices = samples.ray_indices
            packed_info = samples.packed_info
        self.last_sampling = {"n_marched": int(t_starts.shape[0])}     # samples the density pre-pass sees (measurement only)
        if (alpha_thre > 0.0 or early_stop_eps > 0.0) and sigma_fn is not None and t_starts.is_cuda and t_starts.shape[0] > 0:
            # visibility test and the three mask selections on the device: two passes around one prefix sum, ONE host round trip (the survivor count)
            # instead of occs.mean().item() plus one per boolean index.  Same T / alpha arithmetic as render_visibility_from_density.
            thre = torch.clamp(self.occs.mean(), max=float(alpha_thre)).to(torch.float32).reshape(1)
            if hasattr(sigma_fn, "ray_major") and early_stop_eps > 0.0:
                sigmas = sigma_fn.ray_major(t_starts, t_ends, ray_indices, packed_info, early_stop_eps)
            else:
                sigmas = sigma_fn(t_starts, t_ends, ray_indices)
            assert sigmas.shape == t_starts.shape, "sigmas must have shape of (N,)! Got {}".format(sigmas.shape)
            return _select_visible(t_starts, t_ends, L.contig(sigmas, torch.float32), packed_info, float(early_stop_eps), thre)
        if (alpha_thre > 0.0 or early_stop_eps > 0.0) and (sigma_fn is not None or alpha_fn is not None):
            alpha_thre = min(alpha_thre, self.occs.mean().item())
            if sigma_fn is not None:
                if t_starts.shape[0] == 0:
                    sigmas = torch.empty((0,), device=t_starts.device)
                elif hasattr(sigma_fn, "ray_major") and early_stop_eps > 0.0:
                    # the field's own density pass, ray by ray, without the samples behind T < early_stop_eps / 2 (same mask)
                    sigmas = sigma_fn.ray_major(t_starts, t_ends, ray_indices, packed_info, early_stop_eps)
                else:
                    sigmas = sigma_fn(t_starts, t_ends, ray_indices)
                assert sigmas.shape == t_starts.shape, "sigmas must have shape of (N,)! Got {}".format(sigmas.shape)
                masks = render_visibility_from_density(t_starts=t_starts, t_ends=t_ends, sigmas=sigmas, packed_info=packed_info,
                                                       early_stop_eps=early_stop_eps, alpha_thre=alpha_thre)
            else:
                alphas = alpha_fn(t_starts, t_ends, ray_indices) if t_starts.shape[0] != 0 else torch.empty((0,), device=t_starts.device)
                assert alphas.shape == t_starts.shape, "alphas must have shape of (N,)! Got {}".format(alphas.shape)
                trans = torch.exp(exclusive_sum(torch.log1p(-alphas), packed_info))
                masks = trans >= early_stop_eps
                if alpha_thre > 0:
                    masks = masks & (alphas >= alpha_thre)
            ray_indices, t_starts, t_ends = ray_indices[masks], t_starts[masks], t_ends[masks]
        return ray_indices, t_starts, t_ends

    @torch.no_grad()
    def _sample_single_pass(self, rays_o, rays_d, near_planes, far_planes, step_size, cone_angle, cap=None):
        """One traversal instead of the count + fill pair of `traverse_grids` (csrc/march.hip, sample_rays_kernel): rays
        are marched once into a bounded scratch and the rows are packed afterwards; the result is the one `traverse_grids`
        gives, bit for bit.  Returns None (caller uses the two-pass path) for more than four levels, step_size <= 0 or when a ray
        overflows its scratch row."""
        if self.levels > 4 or step_size <= 0.0 or rays_o.dim() != 2 or not rays_o.is_cuda:
            return None
        n = rays_o.shape[0]
        dev = rays_o.device
        if n == 0:
            return None
        if cap is None:
            cap = int(max(64, min(2048, (1 << 27) // n)))   # scratch <= 1 GiB; longest reference-config ray: ~1400 samples
        b = self.binaries
        b = b.contiguous().view(torch.uint8) if b.dtype == torch.bool else b.to(torch.uint8).contiguous()
        import ctypes
        aabb_host = (ctypes.c_float * (6 * self.levels))(*[x for lvl in range(self.levels) for x in self.aabb_host(lvl)])
        rays_o, rays_d = L.contig(rays_o, torch.float32), L.contig(rays_d, torch.float32)
        near_planes, far_planes = L.contig(near_planes, torch.float32), L.contig(far_planes, torch.float32)
        scratch = torch.empty((2, n, cap), device=dev, dtype=torch.float32)
        counts = torch.empty((n,), device=dev, dtype=torch.int64)
        res = [int(x) for x in b.shape[1:]]
        lib = L.load_library()
        L.launch(lib.mnf_sample_rays_levels, L.ptr(rays_o), L.ptr(rays_d), n, L.ptr(b), self.levels, res[0], res[1], res[2], aabb_host, L.ptr(near_planes),
                                    L.ptr(far_planes), float(step_size), float(cone_angle), cap, L.ptr(scratch[0]), L.ptr(scratch[1]),
                                    L.ptr(counts), L.ptr(self.bitgrid()[0]))
        starts, total_t = exclusive_scan_counts(counts, want_total=True)
        total, longest = (int(x) for x in torch.stack([total_t, counts.max()]).tolist())   # the one sync (data_spec.hpp:91 has it too)
        if longest > cap:
            return None
        t_starts, t_ends = torch.empty((total,), device=dev), torch.empty((total,), device=dev)
        ray_indices = torch.empty((total,), device=dev, dtype=torch.int64)
        if total:
            L.launch(lib.mnf_compact_samples, L.ptr(scratch[0]), L.ptr(scratch[1]), cap, L.ptr(starts), L.ptr(counts), n, L.ptr(t_starts),
                                            L.ptr(t_ends), L.ptr(ray_indices))
        return ray_indices, t_starts, t_ends, torch.stack([starts, counts], dim=-1)

    @torch.no_grad()
    def update_every_n_steps(self, step: int, occ_eval_fn: Callable, occ_thre: float = 1e-2, ema_decay: float = 0.95,
                             warmup_steps: int = 256, n: int = 16) -> None:
        if not self.training:
            raise RuntimeError("You should only call this function only during training. "
                               "Please call _update() directly if you want to update the field during inference.")
        if step % n == 0 and self.training:
            self._update(step=step, occ_eval_fn=occ_eval_fn, occ_thre=occ_thre, ema_decay=ema_decay, warmup_steps=warmup_steps)

    @torch.no_grad()
    def mark_invisible_cells(self, K: Tensor, c2w: Tensor, width: int, height: int, near_plane: float = 0.0,
                             chunk: int = 32 ** 3) -> None:
        """occ_grid.py:279-342: cells no camera sees (or that sit closer than near_plane to one) get occupancy -1 and are
        never sampled again.  Known answer: tests/test_grid.py:207-233 (77 660 / 53 412 cells)."""
        assert K.dim() == 3 and K.shape[1:] == (3, 3)
        assert c2w.dim() == 3 and (c2w.shape[1:] == (3, 4) or c2w.shape[1:] == (4, 4))
        assert K.shape[0] == c2w.shape[0] or K.shape[0] == 1
        rot = c2w[:, :3, :3].transpose(2, 1)                      # world -> camera rotation, one per camera
        shift = -rot @ c2w[:, :3, 3:]
        for lvl in range(self.levels):
            lvl_occs = self.occs[lvl * self.cells_per_lvl:(lvl + 1) * self.cells_per_lvl]
            cells = torch.nonzero(lvl_occs >= 0.0)[:, 0]                       # cells still eligible (occ_grid.py:328-343)
            lo, hi = self.aabbs[lvl, :3], self.aabbs[lvl, 3:]
            for begin in range(0, len(cells), chunk):
                ids = cells[begin:begin + chunk]
                unit = self._cell_coords(ids) / (self.resolution - 1)         # cell corner in [0, 1]^3
                world = (lo + unit * (hi - lo)).T                             # [3, n]
                pix = K @ (rot @ world + shift)                                # [cams, 3, n]: (u*d, v*d, d)
                depth = pix[:, 2]
                u, v = pix[:, 0] / depth, pix[:, 1] / depth
                inside = (depth >= 0) & (u >= 0) & (u < width) & (v >= 0) & (v < height)
                seen = ((depth >= near_plane) & inside).any(0)
                too_close = ((depth < near_plane) & inside).any(0)
                self.occs[lvl * self.cells_per_lvl + ids] = torch.where(seen & ~too_close, 0.0, -1.0)

    @torch.no_grad()
    def _update(self, step: int, occ_eval_fn: Callable, occ_thre: float = 0.01, ema_decay: float = 0.95,
                warmup_steps: int = 256, _draws=None) -> None:
        """occ_grid.py:377-437 on the device (csrc/occupancy.hip): cells and in-cell points are drawn by
        `mnf_occ_sample_cells`, `occ_eval_fn` gives their occupancy, `mnf_occ_apply` does the EMA-max with the NaN
        roll-back and `mnf_occ_binarize` re-thresholds into `binaries` and the bit grid — no boolean-mask compaction and
        no host sync.  With `occ_eval_fn` a `FieldDensityOcc` the chain is one C call (`mnf_update_occupancy`).
        `_draws` (tests): per level (cell ids int64 [n], offsets f32 [n,3]) in place of the device RNG, e.g. the draws
        recorded from the reference in tests/golden/occgrid.npz.  The seed of the device RNG comes from torch's CPU
        generator, so `torch.manual_seed` makes a run repeatable."""
        import ctypes
        L.require_gpu(self.occs, self.binaries)
        lib, dev, cells = L.load_library(), self.occs.device, self.cells_per_lvl
        if self.occs.dtype != torch.float32 or not self.occs.is_contiguous() or not self.binaries.is_contiguous():
            raise L.MnfError("OccGridEstimator._update: occs must be contiguous float32 and binaries contiguous")
        seed = int(torch.randint(0, 2 ** 62, (1,)).item())
        fused = isinstance(occ_eval_fn, FieldDensityOcc) and self.levels == 1 and _draws is None
        nbytes = int(lib.mnf_occ_workspace_bytes(cells, 1 if fused else 0))
        if self._occ_ws is None or self._occ_ws.device != dev or self._occ_ws.numel() < nbytes:
            self._occ_ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        ws = self._occ_ws
        res = [int(r) for r in self.resolution]
        bin_u8 = self.binaries.view(torch.uint8) if self.binaries.dtype == torch.bool else self.binaries
        bits = self.bitgrid()                                       # current before the refresh (the occupied half reads it)
        if fused:
            handle = occ_eval_fn.field._ensure_handle()
            L.launch(lib.mnf_update_occupancy, handle, L.ptr(self.occs), L.ptr(bin_u8), L.ptr(bits), res[0], res[1], res[2],
                     (ctypes.c_float * 6)(*self.aabb_host(0)), int(step), int(warmup_steps), float(occ_thre), float(ema_decay),
                     float(occ_eval_fn.scale), seed, L.ptr(ws), nbytes)
        else:
            for lvl in range(self.levels):
                lvl_occs = self.occs[lvl * cells:(lvl + 1) * cells]
                aabb = (ctypes.c_float * 6)(*self.aabb_host(lvl))
                if _draws is not None:
                    ids_in = L.contig(_draws[lvl][0].to(dev), torch.int64)
                    jit_in = L.contig(_draws[lvl][1].to(dev), torch.float32)
                    cap = n_in = ids_in.shape[0]
                else:
                    ids_in = jit_in = None
                    n_in, cap = 0, int(lib.mnf_occ_list_capacity(cells, int(step), int(warmup_steps)))
                ids = torch.empty((cap,), dtype=torch.int64, device=dev)
                pts = torch.empty((cap, 3), dtype=torch.float32, device=dev)
                L.launch(lib.mnf_occ_sample_cells, L.ptr(lvl_occs), L.ptr(bits[lvl]), res[0], res[1], res[2], aabb, int(step),
                         int(warmup_steps), seed + lvl, L.ptr(ids_in), L.ptr(jit_in), n_in, L.ptr(ids), L.ptr(pts), cap, L.ptr(ws), nbytes)
                occ = L.contig(occ_eval_fn(pts).reshape(-1), torch.float32) if cap else pts.new_empty(0)
                assert occ.shape[0] == cap, f"occ_eval_fn must return one value per point, got {tuple(occ.shape)} for {cap} points"
                L.launch(lib.mnf_occ_apply, L.ptr(lvl_occs), L.ptr(ids), L.ptr(occ), 1.0, cap, cells, float(ema_decay), L.ptr(ws), nbytes)
            L.launch(lib.mnf_occ_binarize, L.ptr(self.occs), cells, self.levels, float(occ_thre), L.ptr(bin_u8), L.ptr(bits), None,
                     L.ptr(ws), nbytes)
        # the kernels wrote occs / binaries / bits in place, outside autograd's view
        torch.autograd.graph.increment_version(self.occs)
        torch.autograd.graph.increment_version(self.binaries)
        b = self.binaries
        self._bits_key = (b.data_ptr(), b._version, tuple(b.shape), str(b.device))
