"""Trained stand-in scenes for measurement: there are no Habitat captures or checkpoints on the GPU box,
so bench.py trains the field itself — `render.train_step` (the product path) for a few thousand iterations on an ANALYTIC
target built from the procedural "rooms" occupancy grid of `synthetic.py`:

    density   opaque (sigma ~ 50 / m: termination ~2 cm behind the first occupied cell a ray meets)
    colour    fract(xyz) at the termination point
    class     hash of the 0.2 m cell there, mod 29
    depth     distance to the termination point

The result has what a trained scene has and a random initialisation lacks: empty space is empty, surfaces are opaque,
the occupancy grid comes out of `OccGridEstimator.update_every_n_steps`, and rays terminate after tens of samples.
Weights and grid are cached under `cache_dir` so that repeated bench invocations on one box train once.
"""
import os
import time

import numpy as np
import torch

from . import nerfacc as NA
from . import render as RD
from . import synthetic as S


def _procedural_estimator(scene, device):
    """The target geometry: the procedural grid with its outer faces closed (a Habitat apartment has no open sides: every
    ray ends on a surface, so no target has zero opacity)."""
    occ = scene["occ"].copy()
    occ[:, 0, :, :] = occ[:, -1, :, :] = True
    occ[:, :, :, 0] = occ[:, :, :, -1] = True
    est = NA.OccGridEstimator(torch.from_numpy(scene["aabb"]), resolution=scene["res"], levels=1)
    est.binaries = torch.from_numpy(occ)
    return est.to(device).eval()


@torch.no_grad()
def analytic_targets(proc_est, aabb, rays_o, rays_d, near=0.1, step=1e-3, cone=0.004):
    """First occupied cell along every ray of the PROCEDURAL grid (the library's own marcher) -> (pixels [R,3], depth [R],
    label [R] int64); rays that leave the box without meeting an occupied cell get black / depth 0 / class 0."""
    n = rays_o.shape[0]
    nearp = torch.full((n,), near, device=rays_o.device)
    farp = torch.full((n,), 1e10, device=rays_o.device)
    packed = proc_est._sample_single_pass(rays_o, rays_d, nearp, farp, step, cone)
    if packed is None:
        iv, sm, _ = NA.traverse_grids(rays_o, rays_d, proc_est.binaries, proc_est.aabbs, near_planes=nearp, far_planes=farp,
                                      step_size=step, cone_angle=cone)
        ts, info = iv.vals[iv.is_left], sm.packed_info
    else:
        _, ts, _, info = packed
    hit = info[:, 1] > 0
    first = info[:, 0].clamp(max=max(ts.shape[0] - 1, 0))
    t_hit = torch.where(hit, ts[first] if ts.shape[0] else torch.zeros_like(nearp), torch.zeros_like(nearp)) + 0.02
    p = rays_o + rays_d * t_hit[:, None]
    pixels = torch.where(hit[:, None], p - torch.floor(p), torch.zeros_like(p))
    a = torch.as_tensor(aabb, device=p.device, dtype=torch.float32)
    cell = torch.floor((p - a[:3]) / 0.2).to(torch.int64)
    label = ((cell[:, 0] * 73856093) ^ (cell[:, 1] * 19349663) ^ (cell[:, 2] * 83492791)) % 29
    return pixels, torch.where(hit, t_hit, torch.zeros_like(t_hit)), torch.where(hit, label, torch.zeros_like(label))


def _free_space_poses(scene, n, seed):
    """n camera poses (xyz + quaternion xyzw) at eye height inside free cells of the procedural grid, random yaw."""
    rng = np.random.default_rng(seed)
    occ, a = scene["occ"][0], scene["aabb"]
    X, Y, Z = occ.shape
    y_eye = min(int((1.5 - a[1]) / 0.2), Y - 2)
    free = np.argwhere(~occ[:, y_eye, :])
    # keep cells whose 3x3 neighbourhood is free too (the camera is not inside a wall's first sample)
    ok = [c for c in free if 1 <= c[0] < X - 1 and 1 <= c[1] < Z - 1 and not occ[c[0] - 1:c[0] + 2, y_eye, c[1] - 1:c[1] + 2].any()]
    pick = rng.choice(len(ok), size=n, replace=len(ok) < n)
    out = np.zeros((n, 7))
    for i, k in enumerate(pick):
        cx, cz = ok[k]
        yaw = rng.uniform(0, 2 * np.pi)
        out[i, :3] = [a[0] + (cx + 0.5) * 0.2, 1.5, a[2] + (cz + 0.5) * 0.2]
        out[i, 3:] = [0.0, np.sin(yaw / 2), 0.0, np.cos(yaw / 2)]
    return out


PROTOCOL_VERSION = 5      # bump when the training protocol or anything it runs through changes what a given tag would produce


def default_cache_dir() -> str:
    """Per-user cache directory (MNF_CACHE_DIR, else $XDG_CACHE_HOME/mi355nerf, else ~/.cache/mi355nerf, else a per-uid
    directory under the system temp dir), created with mode 0700."""
    import tempfile
    d = os.environ.get("MNF_CACHE_DIR")
    if not d:
        base = os.environ.get("XDG_CACHE_HOME") or (os.path.join(os.path.expanduser("~"), ".cache") if os.path.expanduser("~") not in ("", "/") else None)
        d = os.path.join(base, "mi355nerf") if base else os.path.join(tempfile.gettempdir(), f"mi355nerf-{os.getuid()}")
    try:
        os.makedirs(d, mode=0o700, exist_ok=True)
    except OSError:
        d = os.path.join(tempfile.gettempdir(), f"mi355nerf-{os.getuid()}")
        os.makedirs(d, mode=0o700, exist_ok=True)
    return d


def train_standin(scene, device, steps=2000, max_rays=8192, target_samples=1 << 21, lr=2e-3, seed=9, cache_dir=None,
                  n_poses=64, verbose=False, lr_final=2e-4, keep_optimizer=False):
    """-> (NGPRadianceField, OccGridEstimator, info dict), trained as described in the module docstring with
    `render.train_step` + `optim.FusedAdam`; the learning rate stays at `lr` for the first half and decays geometrically to
    `lr_final` over the second.  Training is bitwise reproducible: every random draw comes from generators seeded here and the
    train step runs in its deterministic mode (order-independent gradient accumulation, `mnf_train_opts.deterministic`), so the
    stand-in — and with it the samples per ray of the benchmark views — is the same on every box and in every run of one build
    (rounds 1-2 trained with float atomics and unseeded jitter: 91-137 samples per ray across boxes).  `scene` is `tests/helpers.make_scene`-shaped (aabb, res, occ, neurons, layers, C,
    log2_hashmap_size).  `keep_optimizer=True` also returns (and caches) the optimizer's state_dict in info["optimizer_state"], so that a
    caller can continue the training run where it stopped (Adam moments warm, learning rate `lr_final`)."""
    from .ngp import NGPRadianceField
    from .optim import FusedAdam
    from . import _lib as L
    import hashlib
    # everything that decides what training produces is part of the tag, including the library build (kernels change between rounds)
    try:
        with open(L.lib_path(), "rb") as fh:
            lib_id = hashlib.md5(fh.read()).hexdigest()[:12]
    except OSError:
        lib_id = "nolib"
    tag = (("opt_" if keep_optimizer else "") + f"v{PROTOCOL_VERSION}_{scene.get('name')}_{tuple(np.round(scene['aabb'], 3))}_{tuple(scene['res'])}_{scene['neurons']}x{scene['layers']}_C{scene['C']}"
           f"_T{scene['log2_hashmap_size']}_s{steps}_r{max_rays}_t{target_samples}_p{n_poses}_lr{lr}_lrf{lr_final}_seed{seed}_lib{lib_id}")
    cache_dir = default_cache_dir() if cache_dir is None else cache_dir
    path = os.path.join(cache_dir, "mnf_standin_" + hashlib.md5(tag.encode()).hexdigest()[:16] + ".pt")
    field = NGPRadianceField(aabb=torch.from_numpy(scene["aabb"]), neurons=scene["neurons"], layers=scene["layers"],
                             num_semantic_classes=scene["C"], log2_hashmap_size=scene["log2_hashmap_size"], seed=seed).to(device)
    est = NA.OccGridEstimator(torch.from_numpy(scene["aabb"]), resolution=scene["res"], levels=1).to(device)
    if os.path.exists(path):
        try:
            ck = torch.load(path, map_location=device, weights_only=True)     # tensors and plain containers only: no code runs on load
        except Exception:
            ck = {}
        if ck.get("tag") == tag:
            field.load_state_dict(ck["model"])
            est.occs.copy_(ck["occs"]); est.binaries = ck["binaries"].to(device)
            info = dict(ck["info"], cached=True)
            if keep_optimizer:
                info["optimizer_state"] = ck["optimizer_state"]
            return field.eval(), est.eval(), info
    proc = _procedural_estimator(scene, device)
    # the captures of the reference's first phase (a yaw sweep about the start pose, pipeline.py:252-264) plus views from
    # elsewhere in the free space, so that the trained region is the one the benchmark's views look at
    sweep = S.camera_poses(S.SCENES[scene["name"]]["origin"], 40) if scene.get("name") in S.SCENES else np.zeros((0, 7))
    poses = np.concatenate([sweep, _free_space_poses(scene, max(n_poses - len(sweep), 8), seed)])
    n_poses = len(poses)
    c2w = np.stack([RD.pose_to_c2w(p) for p in poses]).astype(np.float32)
    K = np.array([[320.0, 0, 320], [0, 320.0, 320], [0, 0, 1.0]])
    opt = FusedAdam(field.parameters(), lr=lr, eps=1e-15).bind_field(field)
    gen = torch.Generator().manual_seed(seed)
    rng_state = torch.random.get_rng_state()
    torch.manual_seed(1000 + seed)          # the near-plane jitter and the occupancy refresh draw their device seeds from torch's CPU generator
    n_rays, t0, losses, skipped, n_samp = 1024, time.perf_counter(), [], 0, 0
    bk = torch.zeros(3, device=device)
    for step in range(steps):
        if lr_final is not None and step >= steps // 2:           # geometric decay over the second half
            for g in opt.param_groups:
                g["lr"] = lr * (lr_final / lr) ** ((step - steps // 2) / max(1, steps - steps // 2 - 1))
        idx = torch.randint(0, 640 * 640, (n_rays,), generator=gen).numpy()
        ys, xs = idx // 640, idx % 640
        idx = idx[np.argsort((ys // 32) * 20 + xs // 32, kind="stable")]      # grouped by 32x32 image block (dataset.fetch_data)
        k = step % n_poses
        rays = RD.generate_image_rays(torch.from_numpy(c2w[k:k + 1]), 640, 640, K, device, idx)
        pix, dep, lab = analytic_targets(proc, scene["aabb"], rays.origins, rays.viewdirs)
        bk = torch.rand(3, generator=gen).to(device)                # habitat_to_data.py:189-191: a random background colour per training batch
        out = RD.train_step(field, est, opt, rays, pix, dep, lab, bk, step=step, near_plane=0.1, render_step_size=1e-3,
                            cone_angle=0.004, alpha_thre=0.01, occ_thre=1e-2, deterministic=True)
        n_samp = out["n_rendering_samples"]
        skipped += int(out["skipped"])
        if n_samp > 0:                                                         # pipeline.py:494-504: keep the sample batch near the target
            n_rays = int(min(max_rays, max(256, n_rays * target_samples / n_samp)))
        if out.get("loss") is not None and (step % 100 == 0 or step == steps - 1):
            losses.append(float(out["loss"]))
            if verbose:
                print(f"[standin] step {step}: loss {losses[-1]:.4f}, rays {n_rays}, samples {n_samp}, "
                      f"occupied {int(est.binaries.sum())}", flush=True)
    torch.cuda.synchronize(device)
    torch.random.set_rng_state(rng_state)
    info = dict(steps=steps, seconds=time.perf_counter() - t0, loss_first=losses[0] if losses else None,
                loss_last=losses[-1] if losses else None, skipped_steps=skipped, final_rays=n_rays, final_samples=n_samp,
                occupied_cells=int(est.binaries.sum()), cells=int(est.binaries.numel()), cached=False)
    info["saved"] = False
    try:
        os.makedirs(cache_dir, mode=0o700, exist_ok=True)
        tmp = f"{path}.{os.getpid()}.tmp"
        blob = {"tag": tag, "model": field.state_dict(), "occs": est.occs, "binaries": est.binaries, "info": dict(info)}
        if keep_optimizer:                                        # Adam moments and step count: training can CONTINUE from here (bench.py's train leg)
            blob["optimizer_state"] = opt.state_dict()
        torch.save(blob, tmp)
        os.replace(tmp, path)                                     # atomic: a reader never sees a half-written file
        info["saved"] = True
    except (OSError, RuntimeError):
        pass                                                      # callers with several ranks broadcast the model instead (bench.py)
    if keep_optimizer:
        info["optimizer_state"] = opt.state_dict()
    return field.eval(), est.eval(), info


def shared_standin(scene, device, steps=2000, seed=9, keep_optimizer=False, group=None, cache_dir=None, log=None):
    """`train_standin` for every rank of a job with ONE training: rank 0 trains (or loads its cache) and says whether the cache file exists now; the other ranks
    load that file, or — when it could not be written (read-only home, full disk) — build empty modules and receive weights and occupancy grid by
    `distributed.broadcast_model`.  Every rank reaches the same collectives whatever happened to the file.  -> (field, estimator, info); info of ranks that
    received the model by broadcast carries `received_by_broadcast=True` and no optimizer state."""
    import torch.distributed as dist
    from . import distributed as DD
    distributed = group is not False and dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
    rank = dist.get_rank(group) if distributed else 0
    field = est = None
    info = {}
    flag_dev = device if (distributed and dist.get_backend(group) == "nccl") else "cpu"
    saved = torch.zeros(1, device=flag_dev)
    if rank == 0:
        field, est, info = train_standin(scene, device, steps=steps, seed=seed, keep_optimizer=keep_optimizer, cache_dir=cache_dir)
        if log:
            log(f"stand-in {scene.get('name')} seed {seed}: " + str({k: v for k, v in info.items() if k != "optimizer_state"}))
        saved.fill_(1.0 if info.get("saved", info.get("cached")) else 0.0)
    if distributed:
        dist.broadcast(saved, src=0, group=group)
        if rank != 0:
            if saved.item() > 0:
                field, est, info = train_standin(scene, device, steps=steps, seed=seed, keep_optimizer=keep_optimizer, cache_dir=cache_dir)      # cache hit
            else:
                from . import scenes as SC
                field = SC.hip_field(scene, device)
                est = NA.OccGridEstimator(torch.from_numpy(scene["aabb"]), resolution=scene["res"], levels=1).to(device)
                info = {"received_by_broadcast": True}
        # Ranks that loaded "the file rank 0 wrote" may have loaded nothing of the kind (a cache directory that is not shared between nodes: every rank then trained its
        # own stand-in) — nothing above would notice, and view-sharded scoring would disagree across ranks.  One small collective settles it: every rank's parameter /
        # grid checksum against rank 0's; any mismatch and the model travels by broadcast after all.
        need = saved.clone().zero_()
        if saved.item() > 0:
            mine = _model_checksum(field, est).to(flag_dev)
            ref = mine.clone()
            dist.broadcast(ref, src=0, group=group)
            need.fill_(0.0 if torch.equal(mine, ref) else 1.0)
            dist.all_reduce(need, op=dist.ReduceOp.MAX, group=group)
            if need.item() > 0 and rank != 0:
                info = dict(info, cache_mismatch=True, received_by_broadcast=True)
                info.pop("optimizer_state", None)
        if saved.item() == 0 or need.item() > 0:
            DD.broadcast_model(field, est, src=0, group=group)
    return field.eval(), est.eval(), info


def _model_checksum(field, est) -> torch.Tensor:
    """[4] float64: sums of the three parameter vectors (as int32 bit patterns: exact, order-independent) and the number of occupied cells"""
    out = [p.detach().view(torch.int32).to(torch.float64).sum() for p in (field.mlp_base.params, field.mlp_head.params, field.mlp_sem.params)]
    out.append(est.binaries.to(torch.float64).sum())
    return torch.stack(out).cpu()
