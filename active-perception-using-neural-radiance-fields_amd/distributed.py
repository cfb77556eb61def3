"""Multi-GPU helpers of the hot path, one process per GPU over `torch.distributed` (RCCL on GPUs, gloo on
CPU tensors in the tests).  Nothing here invents a collective: each function is the ONE exchange its row of §8e names.

  render_views_sharded   full-resolution / pose-list rendering split over ranks + one all-gather of the per-ray outputs
                         (scripts/pipeline.py:960-974 renders a pose list; habitat_to_data.py:304-549)
  broadcast_model        weights and occupancy grid of a freshly trained member to every rank (the only
                         bandwidth-relevant transfer, ~100 MB of fp32 master parameters per member)
  ensemble_placement     which rank trains which ensemble member (the members are independent: pipeline.py:398-412)
  (view-sharded scoring and ray-data-parallel gradient averaging live next to their callers in render.py)
"""
from typing import Dict, List, Optional, Tuple

import torch
import torch.distributed as dist

from . import render as RD


def _world(group) -> Tuple[int, int]:
    if group is False or not (dist.is_available() and dist.is_initialized()):
        return 1, 0
    return dist.get_world_size(group), dist.get_rank(group)


def shard_rays(n_views: int, rays_per_view: int, world: int, rank: int) -> Dict[str, int]:
    """How a batch of `n_views` views of `rays_per_view` rays is split over `world` ranks.
    n_views >= world: contiguous slices of WHOLE views (every view keeps the round schedule of its reference call, so the
    gathered result is bit-identical to the single-GPU render).  Fewer views than ranks: every view is cut into
    `tiles_per_view` contiguous row tiles (rays are row-major pixels) and tiles are dealt like views; each tile is then
    rendered as a reference call of its own — the reference never splits a view, so this is a documented deviation in
    the per-round sample budget (`n_samples = R // n_alive` is taken per tile)."""
    tiles_per_view = 1 if n_views >= world else -(-world // n_views)
    while rays_per_view % tiles_per_view:
        tiles_per_view += 1
    units, unit_rays = n_views * tiles_per_view, rays_per_view // tiles_per_view
    per = -(-units // world)
    lo, hi = min(rank * per, units), min((rank + 1) * per, units)
    return dict(tiles_per_view=tiles_per_view, units=units, unit_rays=unit_rays, per=per, lo=lo, hi=hi)


def gather_rows(local: torch.Tensor, rows_per_rank: int, total_rows: int, group=None) -> torch.Tensor:
    """One `all_gather_into_tensor` of equally sized [rows_per_rank, D] blocks (`local` is padded to that size) ->
    [total_rows, D] on every rank."""
    world, _ = _world(group)
    if world == 1:
        return local[:total_rows]
    D = local.shape[1]
    block = local
    if local.shape[0] != rows_per_rank:
        block = local.new_zeros((rows_per_rank, D))
        block[:local.shape[0]] = local
    out = local.new_empty((world * rows_per_rank, D))
    RD.all_gather_blocks(out, block.contiguous(), group)
    return out[:total_rows]


@torch.no_grad()
def render_views_sharded(radiance_field, estimator, rays_o, rays_d, rays_per_view: int, group=None, probabilistic: bool = False,
                         image_hw=None, **render_kw) -> Dict[str, torch.Tensor]:
    """`render.render_views` over the ranks of `group`: every rank renders its share (`shard_rays`) and ONE all-gather of
    the packed per-ray outputs [rgb 3 | acc 1 | depth 1 | sem C (| rgb_var 3 | depth_var 1)] gives every rank the full
    result.  All ranks pass the same rays; weights and occupancy grid are replicas."""
    world, rank = _world(group)
    n = rays_o.shape[0]
    V = n // rays_per_view
    C = radiance_field.num_semantic_classes
    sh = shard_rays(V, rays_per_view, world, rank)
    r0, r1 = sh["lo"] * sh["unit_rays"], sh["hi"] * sh["unit_rays"]
    D = 5 + C + (4 if probabilistic else 0)
    dev = rays_o.device
    packed = torch.zeros((max(r1 - r0, 0), D), device=dev)
    counts = torch.zeros(2, dtype=torch.int64, device=dev)
    if r1 > r0:
        hw = image_hw if sh["tiles_per_view"] == 1 else None
        r = RD.render_views(radiance_field, estimator, rays_o[r0:r1].contiguous(), rays_d[r0:r1].contiguous(), sh["unit_rays"],
                            probabilistic=probabilistic, image_hw=hw, **render_kw)
        parts = [r["rgb"], r["acc"], r["depth"], r["sem"]] + ([r["rgb_var"], r["depth_var"]] if probabilistic else [])
        packed = torch.cat(parts, dim=1)
        counts = r["total"]
    full = gather_rows(packed, sh["per"] * sh["unit_rays"], n, group)
    if world > 1:
        dist.all_reduce(counts, group=group)
    out = dict(rgb=full[:, 0:3], acc=full[:, 3:4], depth=full[:, 4:5], sem=full[:, 5:5 + C], total=counts)
    if probabilistic:
        out.update(rgb_var=full[:, 5 + C:8 + C], depth_var=full[:, 8 + C:9 + C])
    return out


def broadcast_model(radiance_field, estimator=None, src: int = 0, group=None) -> None:
    """After a training phase on rank `src`: its three flat parameter vectors (and, if given, the estimator's `occs` /
    `binaries`) replace every other rank's copies — one broadcast per tensor, ring/tree over xGMI under RCCL."""
    world, _ = _world(group)
    if world == 1:
        return
    tensors: List[torch.Tensor] = [p for p in radiance_field.parameters() if p.numel()]
    if estimator is not None:
        tensors += [estimator.occs]
    for t in tensors:
        dist.broadcast(t.data, src=src, group=group)
        torch.autograd.graph.increment_version(t)          # (`.data` has its own counter) the field handle reloads its fp16 copies on next use
    if estimator is not None:
        b = estimator.binaries.to(torch.uint8)               # bool tensors are not a collective dtype
        dist.broadcast(b, src=src, group=group)
        estimator.binaries = b.to(torch.bool)


def ensemble_placement(n_members: int, world: int) -> List[int]:
    """rank that trains ensemble member m: m mod world (members are independent models: no exchange while training)."""
    return [m % world for m in range(n_members)]


def my_members(n_members: int, group=None) -> List[int]:
    world, rank = _world(group)
    return [m for m, r in enumerate(ensemble_placement(n_members, world)) if r == rank]
