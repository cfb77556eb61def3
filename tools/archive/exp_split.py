"""A/B of the number of concurrent render jobs (`n_split`: groups of views advancing side by side) on the 800x800 render (4 views
per call, trained stand-in of scene 102344529) and on one member's 256- / 32-view scoring render.
    python tools/exp_split.py"""
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import apnrf_amd  # noqa: F401
from apnrf_amd import render as RD
from apnrf_amd import scenes as SC
from apnrf_amd import standin as SI

dev = "cuda:0"


def timeit(fn, n=6):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        r = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n, r


scene = SC.make_scene("102344529", n_poses=40)
field, est, _ = SI.train_standin(scene, dev, seed=9)
W = 800
focal = 0.5 * W / np.tan(np.pi / 4)
c2w = np.stack([RD.pose_to_c2w(p) for p in scene["poses"][[0, 5, 10, 15]]]).astype(np.float32)
K = np.array([[focal, 0, W / 2], [0, focal, W / 2], [0, 0, 1.0]])
rays = RD.generate_image_rays(torch.from_numpy(c2w), W, W, K, dev)
o, d = rays.origins.reshape(-1, 3).contiguous(), rays.viewdirs.reshape(-1, 3).contiguous()
ref = None
for ns in (1, 2, 4):
    dt, r = timeit(lambda: RD.render_views(field, est, o, d, W * W, 1024, render_bkgd=torch.zeros(3), image_hw=(W, W), n_split=ns, **SC.RENDER_KW))
    ev = int(r["total"][1])
    same = True if ref is None else all(torch.equal(r[k], ref[k]) for k in ("rgb", "acc", "depth", "sem"))
    ref = r if ref is None else ref
    print(f"[exp_split] render800 x4 n_split={ns}: {1e3 * dt:.2f} ms, {ev / dt / 1e9:.3f} G samples/s, bit-identical to n_split=1: {same}", flush=True)

scene = SC.make_scene("102344250", n_poses=40)
f0, e0, _ = SI.train_standin(scene, dev, seed=9)
poses = SI._free_space_poses(scene, 256, seed=9)
for V in (256, 32):
    o, d, h, w = RD._pose_rays(poses[:V], 640, 640, 320.0, 0.1, dev)
    for hw in ((h, w), None):
        for ns in (1, 2, 4, 8):
            dt, r = timeit(lambda: RD.render_views(f0, e0, o, d, h * w, 1024, near_plane=0.1, render_step_size=1e-3, render_bkgd=torch.zeros(3),
                                                   cone_angle=0.004, alpha_thre=0.01, probabilistic=True, image_hw=hw, n_split=ns))
            ev = int(r["total"][1])
            print(f"[exp_split] score V={V} one member, view_order={'8x8 blocks' if hw else 'row-major'} n_split={ns}: {1e3 * dt:.2f} ms, "
                  f"{ev / dt / 1e9:.3f} G samples/s", flush=True)

# both ensemble members as jobs of one call: groups per member 1 / 2
f1, e1, _ = SI.train_standin(scene, dev, seed=10)
for V in (256, 32):
    o, d, h, w = RD._pose_rays(poses[:V], 640, 640, 320.0, 0.1, dev)
    for ns in (1, 2):
        dt, r = timeit(lambda: RD._render_jobs([(f0, e0, o, d), (f1, e1, o, d)], h * w, 1024, 0.1, 1e10, 1e-3, torch.zeros(3), 0.004, 0.01, 1e-4, True, 8, None, ns))
        ev = sum(int(x["total"][1]) for x in r)
        print(f"[exp_split] score V={V} two members in one call, groups per member {ns}: {1e3 * dt:.2f} ms, {ev / dt / 1e9:.3f} G samples/s", flush=True)
    dt0, _ = timeit(lambda: [RD.render_views(f, e, o, d, h * w, 1024, near_plane=0.1, render_step_size=1e-3, render_bkgd=torch.zeros(3), cone_angle=0.004,
                                             alpha_thre=0.01, probabilistic=True, n_split=1) for f, e in ((f0, e0), (f1, e1))])
    print(f"[exp_split] score V={V} two members back to back (round 2's form): {1e3 * dt0:.2f} ms", flush=True)
