"""Synthetic stand-ins for the Habitat scenes (no Habitat data or checkpoints exist on the GPU box):
scene boxes and grid resolutions from the reference yaml files, a procedural "rooms" occupancy grid,
random-init field parameters and camera poses.  Pure numpy; used by bench.py, the
tests and smoke() to build INPUTS — never to compute results.
"""
import math
from typing import Dict

import numpy as np

# scripts/config_<scene>.yaml: aabb, global_origin (xyz + rpy)
SCENES = {
    "102344250": dict(aabb=[-19.1, -0.2, -19.1, 0.5, 3.2, 0.5], origin=[-14.79389263, 1.5, -10.6045085]),
    "102344529": dict(aabb=[-12.0, -0.2, -12.0, 12.0, 4.2, 12.0], origin=[0.0, 1.5, 0.0]),
    "102344280": dict(aabb=[-13.0, -0.2, -13.0, 14.0, 4.2, 15.0], origin=[0.5, 1.5, 1.0]),
}
# scripts/config_*.yaml render settings
RENDER = dict(near_plane=0.1, render_step_size=1e-3, cone_angle=0.004, alpha_thre=0.01, max_samples=1024)


def grid_resolution(aabb, cell_size=0.2):
    """scripts/pipeline.py:113-120 (float32 arithmetic, then truncation)."""
    a = np.asarray(aabb, np.float32)
    return ((a[3:] - a[:3]) / cell_size).astype(int).tolist()


def make_occupancy(resolution, seed: int = 9, cell: float = 0.2, clutter: float = 0.05, aabb=None, free_at=None,
                   free_radius: float = 0.9) -> np.ndarray:
    """[1,X,Y,Z] bool: floor and ceiling slabs, wall slabs every 4 m with door gaps, random clutter.
    `free_at` (list of xyz, with `aabb`) clears a column of free space around camera positions."""
    X, Y, Z = resolution
    rng = np.random.default_rng(seed)
    occ = np.zeros((X, Y, Z), bool)
    occ[:, 0, :] = True
    occ[:, Y - 1, :] = True
    pitch = max(int(round(4.0 / cell)), 4)
    door = max(int(round(1.0 / cell)), 2)
    door_top = min(int(round(2.2 / cell)), Y - 1)
    for x in range(0, X, pitch):
        occ[x, :, :] = True
        for z0 in range(pitch // 2, Z, pitch):
            occ[x, 1:door_top, z0:z0 + door] = False
    for z in range(0, Z, pitch):
        occ[:, :, z] = True
        for x0 in range(pitch // 2, X, pitch):
            occ[x0:x0 + door, 1:door_top, z] = False
    occ |= rng.random((X, Y, Z)) < clutter
    if free_at is not None and aabb is not None:
        a = np.asarray(aabb, np.float64)
        r = int(math.ceil(free_radius / cell))
        for p in np.asarray(free_at, np.float64).reshape(-1, 3):
            c = ((p - a[:3]) / cell).astype(int)
            occ[max(c[0] - r, 0):c[0] + r + 1, 1:Y - 1, max(c[2] - r, 0):c[2] + r + 1] = False
    return occ[None]


def table_entries(n_levels=16, log2_hashmap_size=19, base_resolution=16, max_resolution=4096) -> int:
    pls = math.exp((math.log(max_resolution) - math.log(base_resolution)) / (n_levels - 1))
    total = 0
    for l in range(n_levels):
        scale = np.float32(2.0 ** (l * math.log2(pls)) * base_resolution - 1.0)
        res = int(math.ceil(float(scale))) + 1
        total += min(((res ** 3 + 7) // 8) * 8, 1 << log2_hashmap_size)
    return total


def make_field_params(neurons=128, layers=2, num_semantic_classes=29, seed=0, grid_scale=0.5, density_gain=8.0,
                      log2_hashmap_size=19, head_gain=1.0) -> Dict[str, np.ndarray]:
    """Flat fp32 parameter vectors in the reference state_dict layout (`mlp_base`, `mlp_head`, `mlp_sem`).
    tcnn's initialisation is grid U(-1e-4,1e-4) + xavier-uniform MLPs, which gives a near-constant density
    of exp(-1) (rays never saturate).  `grid_scale` widens the grid range and the density-logit row becomes
    |w| * density_gain, which with the defaults gives a median density of ~17 /m (5..95 %: 6..60) so that rays
    terminate after ~50 samples, as in a trained scene; grid_scale=1e-4, density_gain=0 restores tcnn's init.
    `head_gain` scales the output rows of the two heads (larger logits for stress tests)."""
    rng = np.random.default_rng(seed)
    W, Wh = neurons, neurons // 2
    sem_pad = ((num_semantic_classes + 15) // 16) * 16

    def mlp(shapes):
        return [rng.uniform(-math.sqrt(6.0 / (o + i)), math.sqrt(6.0 / (o + i)), size=(o, i)).astype(np.float32) for o, i in shapes]

    base = mlp([(W, 64)] + [(W, W)] * (layers - 1) + [(16, W)])
    if density_gain > 0:
        base[-1][0, :] = np.abs(base[-1][0, :]) * np.float32(density_gain)
    table = rng.uniform(-grid_scale, grid_scale, size=table_entries(log2_hashmap_size=log2_hashmap_size) * 4).astype(np.float32)
    head = mlp([(Wh, 32), (Wh, Wh), (16, Wh)])
    sem = mlp([(Wh, 16), (Wh, Wh), (sem_pad, Wh)])
    head[-1] *= np.float32(head_gain)
    sem[-1] *= np.float32(head_gain)
    cat = lambda ws: np.concatenate([w.reshape(-1) for w in ws])
    return {"mlp_base": np.concatenate([cat(base), table]), "mlp_head": cat(head), "mlp_sem": cat(sem)}


def camera_poses(origin, n: int, seed: int = 9, radius: float = 0.0) -> np.ndarray:
    """[n,7] xyz + quaternion (x,y,z,w): a yaw sweep about the vertical axis at `origin`
    (pipeline.py:252-264 initial sweep), optionally jittered inside `radius` metres."""
    rng = np.random.default_rng(seed)
    out = np.zeros((n, 7))
    for i in range(n):
        yaw = 2 * math.pi * i / n
        out[i, :3] = np.asarray(origin[:3]) + (rng.uniform(-radius, radius, 3) * [1, 0.1, 1] if radius > 0 else 0)
        out[i, 3:] = [0.0, math.sin(yaw / 2), 0.0, math.cos(yaw / 2)]
    return out
