"""ORACLE (test infrastructure only): CPU restatement of the reference's volume rendering
and render orchestration.  torch CPU, fp32, packed samples.

  render_weight_from_density     <- perception/nerfacc/nerfacc/volrend.py:212-267, :315-365
  render_visibility_from_density <- volrend.py:424-483
  accumulate_along_rays          <- volrend.py:486-576
  sampling                       <- perception/nerfacc/nerfacc/estimators/occ_grid.py:80-238
  sem_rendering                  <- perception/models/utils.py:362-461
  render_train                   <- utils.py:63-219 (render_image_with_occgrid_with_depth_guide)
  render_test / render_prob_test <- utils.py:555-779 / :782-1032
  generate_image_rays            <- perception/data_proc/habitat_to_data.py:274-301
  subsample_indices              <- habitat_to_data.py:462-467
  pose_to_c2w                    <- habitat_to_data.py:444-451 (scipy Rotation.from_quat, xyzw)
"""
from typing import Callable, Optional

import numpy as np
import torch

from . import marcher as M


def _t(a, dtype=torch.float32):
    return torch.as_tensor(np.asarray(a), dtype=dtype)


# ---------------------------------------------------------------- volrend
class _ExclusiveSum(torch.autograd.Function):
    """scan.py:206-229: the gradient of a packed exclusive sum is the reverse-direction exclusive sum of the gradient."""

    @staticmethod
    def forward(ctx, x, packed_info):
        ctx.packed_info = packed_info
        return torch.from_numpy(M.exclusive_sum(x.detach().numpy(), packed_info))

    @staticmethod
    def backward(ctx, g):
        return torch.from_numpy(M.exclusive_sum(g.contiguous().numpy(), ctx.packed_info, backward=True)), None


def exclusive_sum(x: torch.Tensor, packed_info: np.ndarray) -> torch.Tensor:
    return _ExclusiveSum.apply(x, packed_info)


def render_transmittance_from_density(t_starts, t_ends, sigmas, packed_info, prefix_trans=None):
    sigmas_dt = sigmas * (t_ends - t_starts)
    alphas = 1.0 - torch.exp(-sigmas_dt)
    trans = torch.exp(-exclusive_sum(sigmas_dt, packed_info))
    if prefix_trans is not None:
        trans = trans * prefix_trans
    return trans, alphas


def render_weight_from_density(t_starts, t_ends, sigmas, packed_info, prefix_trans=None):
    trans, alphas = render_transmittance_from_density(t_starts, t_ends, sigmas, packed_info, prefix_trans)
    return trans * alphas, trans, alphas


def render_visibility_from_density(t_starts, t_ends, sigmas, packed_info, early_stop_eps=1e-4, alpha_thre=0.0):
    trans, alphas = render_transmittance_from_density(t_starts, t_ends, sigmas, packed_info)
    vis = trans >= early_stop_eps
    if alpha_thre > 0:
        vis = vis & (alphas >= alpha_thre)
    return vis


def accumulate_along_rays(weights, values, ray_indices, n_rays, out=None):
    src = weights[:, None] if values is None else weights[:, None] * values
    if out is None:
        out = torch.zeros(n_rays, src.shape[-1], dtype=src.dtype)
    out.index_add_(0, ray_indices, src)
    return out


# ---------------------------------------------------------------- ray generation
def _fma32(a, b, c):
    """fmaf emulated through float64 (the product of two fp32 values is exact in fp64)."""
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(np.float32)


def generate_image_rays(c2w, width: int, height: int, focal: float, idx: Optional[np.ndarray] = None):
    """habitat_to_data.py:274-301, explicit fp32 arithmetic (verified bit-exact against the
    reference function through tests/golden/raygen.npz):
      cam = [(x - W/2 + 0.5) / f, -((y - H/2 + 0.5) / f), -1]        x fastest ('xy' meshgrid)
      d_i = (cam0*R[i,0] + cam1*R[i,1]) + cam2*R[i,2]
      viewdir = d / sqrt(fma(dz,dz, fma(dy,dy, dx*dx)))              (torch.linalg.norm's fp32 kernel)
    `idx` optionally selects flat pixel indices (the linspace sub-sampler, habitat_to_data.py:462-467)
    so only those rays are generated.  Returns torch tensors (origins, viewdirs)."""
    f32 = np.float32
    c2w = np.asarray(c2w, f32).reshape(-1, 4)[:3]
    if idx is None:
        idx = np.arange(width * height)
    xs = (idx % width).astype(f32)
    ys = (idx // width).astype(f32)
    fx = f32(focal)
    cam0 = (xs - f32(width / 2) + f32(0.5)) / fx
    cam1 = (ys - f32(height / 2) + f32(0.5)) / fx * f32(-1.0)
    cam2 = np.full_like(cam0, -1.0)
    d = np.stack([(cam0 * c2w[i, 0] + cam1 * c2w[i, 1]) + cam2 * c2w[i, 2] for i in range(3)], -1)
    n = np.sqrt(_fma32(d[:, 2], d[:, 2], _fma32(d[:, 1], d[:, 1], d[:, 0] * d[:, 0])))
    viewdirs = d / n[:, None]
    origins = np.broadcast_to(c2w[:, 3], viewdirs.shape).copy()
    return torch.from_numpy(origins), torch.from_numpy(viewdirs)


def subsample_indices(n_total: int, n_keep: int) -> np.ndarray:
    return np.round(np.linspace(0, n_total - 1, n_keep)).astype(int)


def pose_to_c2w(p: np.ndarray) -> torch.Tensor:
    from scipy.spatial.transform import Rotation as R
    pose = np.eye(4)
    pose[:3, :3] = R.from_quat(p[3:]).as_matrix()
    pose[:3, 3] = p[:3]
    return torch.from_numpy(pose).unsqueeze(0).float()


# ---------------------------------------------------------------- sampling (train mode)
def sampling(binaries, aabbs, occs_mean: float, rays_o, rays_d, sigma_fn: Optional[Callable],
             near_planes: torch.Tensor, far_plane=1e10, render_step_size=1e-3, early_stop_eps=1e-4,
             alpha_thre=0.0, cone_angle=0.0):
    """occ_grid.py:80-238.  `near_planes` already carries the stratified jitter
    (occ_grid.py:158-159 draws it from the CUDA RNG; the caller supplies it here)."""
    n = rays_o.shape[0]
    far_planes = np.full(n, far_plane, np.float32)
    iv, sm, _ = M.traverse_grids(rays_o.numpy(), rays_d.numpy(), binaries, aabbs,
                                 near_planes=near_planes.numpy(), far_planes=far_planes,
                                 step_size=render_step_size, cone_angle=cone_angle)
    t_starts = torch.from_numpy(iv.vals[iv.is_left])
    t_ends = torch.from_numpy(iv.vals[iv.is_right])
    ray_indices = torch.from_numpy(sm.ray_indices)
    packed_info = sm.packed_info
    n_all = t_starts.shape[0]
    if (alpha_thre > 0.0 or early_stop_eps > 0.0) and sigma_fn is not None:
        alpha_thre = min(alpha_thre, occs_mean)                      # occ_grid.py:199
        with torch.no_grad():                                        # occ_grid.py:80 (@torch.no_grad)
            sigmas = sigma_fn(t_starts, t_ends, ray_indices) if n_all else torch.empty(0)
            masks = render_visibility_from_density(t_starts, t_ends, sigmas, packed_info, early_stop_eps, alpha_thre)
        ray_indices, t_starts, t_ends = ray_indices[masks], t_starts[masks], t_ends[masks]
    return ray_indices, t_starts, t_ends, n_all


def _positions(rays_o, rays_d, t_starts, t_ends, ray_indices):
    o, d = rays_o[ray_indices], rays_d[ray_indices]
    return o + d * (t_starts + t_ends)[:, None] / 2.0, d           # utils.py:92, :614


def sem_rendering(field, rays_o, rays_d, t_starts, t_ends, ray_indices, n_rays, render_bkgd=None):
    """utils.py:362-461 (+ the rgb_sigma_sem_fn closure utils.py:122-137)."""
    C = field.num_semantic_classes
    if t_starts.shape[0]:
        pos, d = _positions(rays_o, rays_d, t_starts, t_ends, ray_indices)
        rgbs, sigmas, sems = field(pos, d)
        sigmas = sigmas.squeeze(-1)
    else:
        rgbs, sigmas, sems = torch.empty(0, 3), torch.empty(0), torch.empty(0, C)
    packed_info = M.pack_info(ray_indices.numpy(), n_rays)
    weights, trans, alphas = render_weight_from_density(t_starts, t_ends, sigmas, packed_info)
    colors = accumulate_along_rays(weights, rgbs, ray_indices, n_rays)
    opac = accumulate_along_rays(weights, None, ray_indices, n_rays)
    depths = accumulate_along_rays(weights, (t_starts + t_ends)[:, None] / 2.0, ray_indices, n_rays)
    depths = depths / opac.clamp_min(torch.finfo(torch.float32).eps)
    sem = accumulate_along_rays(weights, sems, ray_indices, n_rays)
    if render_bkgd is not None:
        colors = colors + render_bkgd * (1.0 - opac)
    extras = dict(weights=weights, trans=trans, alphas=alphas, sigmas=sigmas, rgbs=rgbs, sems=sems)
    return colors, opac, depths, sem, extras


def render_train(field, binaries, aabbs, occs_mean, rays_o, rays_d, near_planes, render_step_size=1e-3,
                 render_bkgd=None, cone_angle=0.0, alpha_thre=0.0):
    """utils.py:63-219, training branch (single chunk)."""
    def sigma_fn(ts, te, ri):
        pos, _ = _positions(rays_o, rays_d, ts, te, ri)
        return field.query_density(pos).squeeze(-1)

    ri, ts, te, n_all = sampling(binaries, aabbs, occs_mean, rays_o, rays_d, sigma_fn, near_planes,
                                 render_step_size=render_step_size, alpha_thre=alpha_thre, cone_angle=cone_angle)
    rgb, acc, depth, sem, extras = sem_rendering(field, rays_o, rays_d, ts, te, ri, rays_o.shape[0], render_bkgd)
    extras.update(ray_indices=ri, t_starts=ts, t_ends=te, n_all=n_all)
    return rgb, acc, depth, sem, ts.shape[0], extras


# ---------------------------------------------------------------- test-mode renderers
@torch.no_grad()                      # utils.py:554, :781
def _render_test_impl(max_samples, field, binaries, aabbs, rays_o, rays_d, near_plane, far_plane,
                      render_step_size, render_bkgd, cone_angle, alpha_thre, early_stop_eps, probabilistic):
    """utils.py:555-779 (probabilistic=False) and :782-1032 (True)."""
    n_rays = rays_o.shape[0]
    C = field.num_semantic_classes
    opacity = torch.zeros(n_rays, 1)
    depth = torch.zeros(n_rays, 1)
    rgb = torch.zeros(n_rays, 3)
    sem = torch.zeros(n_rays, C)
    depth_var = torch.zeros(n_rays, 1)
    rgb_var = torch.zeros(n_rays, 3)
    ray_mask = np.ones(n_rays, bool)
    min_samples = 1 if cone_angle == 0 else 4
    iter_samples = total_samples = 0
    near_planes = np.full(n_rays, near_plane, np.float32)
    far_planes = np.full(n_rays, far_plane, np.float32)
    o_np, d_np = rays_o.numpy(), rays_d.numpy()
    t_mins, t_maxs, hits = M.ray_aabb_intersect(o_np, d_np, aabbs)
    n_grids = binaries.shape[0]
    cat = np.concatenate([t_mins, t_maxs], -1)
    if n_grids > 1:
        t_indices = np.argsort(cat, -1, kind="stable").astype(np.int64)
        t_sorted = np.take_along_axis(cat, t_indices, -1)
    else:
        t_sorted = cat
        t_indices = np.broadcast_to(np.arange(2 * n_grids, dtype=np.int64), (n_rays, 2 * n_grids)).copy()
    opc_thre = 1 - early_stop_eps
    rounds = []
    while iter_samples < max_samples:
        n_alive = int(ray_mask.sum())
        if n_alive == 0:
            break
        n_samples = max(min(n_rays // n_alive, 64), min_samples)
        iter_samples += n_samples
        iv, sm, term = M.traverse_grids(o_np, d_np, binaries, aabbs, near_planes, far_planes, render_step_size,
                                        cone_angle, n_samples, True, ray_mask, t_sorted, t_indices, hits)
        t_starts = torch.from_numpy(iv.vals[iv.is_left])
        t_ends = torch.from_numpy(iv.vals[iv.is_right])
        ray_indices = torch.from_numpy(sm.ray_indices[sm.is_valid])
        packed_info = sm.packed_info
        if t_starts.shape[0]:
            pos = rays_o[ray_indices] + rays_d[ray_indices] * (t_starts[:, None] + t_ends[:, None]) / 2.0
            rgbs, sigmas, sems = field(pos, rays_d[ray_indices])
            sigmas = sigmas.squeeze(-1)
        else:
            rgbs, sigmas, sems = torch.zeros(0, 3), torch.zeros(0), torch.zeros(0, C)
        weights, _, alphas = render_weight_from_density(
            t_starts, t_ends, sigmas, M.pack_info(ray_indices.numpy(), n_rays),
            prefix_trans=1 - opacity[ray_indices].squeeze(-1))
        if alpha_thre > 0:
            vis = alphas >= alpha_thre
            ray_indices, rgbs, weights, t_starts, t_ends, sems = (
                ray_indices[vis], rgbs[vis], weights[vis], t_starts[vis], t_ends[vis], sems[vis])
        accumulate_along_rays(weights, rgbs, ray_indices, n_rays, out=rgb)
        accumulate_along_rays(weights, None, ray_indices, n_rays, out=opacity)
        accumulate_along_rays(weights, (t_starts + t_ends)[:, None] / 2.0, ray_indices, n_rays, out=depth)
        accumulate_along_rays(weights, sems, ray_indices, n_rays, out=sem)
        if probabilistic:
            # running, un-normalised means AFTER this round's accumulation (utils.py:984-999)
            accumulate_along_rays(weights, torch.pow(rgbs - rgb[ray_indices], 2), ray_indices, n_rays, out=rgb_var)
            accumulate_along_rays(weights, torch.pow((t_starts + t_ends)[:, None] / 2.0 - depth[ray_indices], 2),
                                  ray_indices, n_rays, out=depth_var)
        near_planes = term
        ray_mask = np.logical_and(opacity.view(-1).numpy() <= opc_thre, packed_info[:, 1] == n_samples)
        total_samples += int(ray_indices.shape[0])
        rounds.append((n_alive, n_samples))
    rgb = rgb + render_bkgd * (1.0 - opacity)
    depth = depth / opacity.clamp_min(torch.finfo(torch.float32).eps)
    out = dict(rgb=rgb, acc=opacity, depth=depth, sem=sem, total_samples=total_samples, rounds=rounds)
    if probabilistic:
        out.update(rgb_var=rgb_var, depth_var=depth_var)
    return out


def render_test(max_samples, field, binaries, aabbs, rays_o, rays_d, near_plane=0.0, far_plane=1e10,
                render_step_size=1e-3, render_bkgd=None, cone_angle=0.0, alpha_thre=0.0, early_stop_eps=1e-4):
    return _render_test_impl(max_samples, field, binaries, aabbs, rays_o, rays_d, near_plane, far_plane,
                             render_step_size, render_bkgd, cone_angle, alpha_thre, early_stop_eps, False)


def render_prob_test(max_samples, field, binaries, aabbs, rays_o, rays_d, near_plane=0.0, far_plane=1e10,
                     render_step_size=1e-3, render_bkgd=None, cone_angle=0.0, alpha_thre=0.0, early_stop_eps=1e-4):
    return _render_test_impl(max_samples, field, binaries, aabbs, rays_o, rays_d, near_plane, far_plane,
                             render_step_size, render_bkgd, cone_angle, alpha_thre, early_stop_eps, True)
