"""Soak (GPU box): the same render call and the same scoring pass many times over; every repetition must reproduce the first one bit for bit (the ticket tile order,
the parity-buffered round words and the job interleaving may not leak into results).    python tools/soak_render.py [repeats]"""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import apnrf_amd  # noqa: F401
from apnrf_amd import render as RD
from apnrf_amd import scenes as SC
from apnrf_amd import standin as SI

n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
dev = "cuda:0"
scene = SC.make_scene("102344529", n_poses=40)
field, est, _ = SI.shared_standin(scene, dev, steps=2000, seed=9, keep_optimizer=False, group=False)
field.eval(); est.eval()
c2w = np.stack([RD.pose_to_c2w(p) for p in scene["poses"][[0, 5, 10, 15]]]).astype(np.float32)
K = np.array([[400.0, 0, 400], [0, 400.0, 400], [0, 0, 1.0]])
rays = RD.generate_image_rays(torch.from_numpy(c2w), 800, 800, K, dev)
o, d = rays.origins.reshape(-1, 3).contiguous(), rays.viewdirs.reshape(-1, 3).contiguous()
first = None
for i in range(n):
    r = RD.render_views(field, est, o, d, 640000, 1024, render_bkgd=torch.zeros(3), image_hw=(800, 800), n_split=1 + i % 4, probabilistic=bool(i & 1), **SC.RENDER_KW)
    keys = ("rgb", "acc", "depth", "sem", "total")
    if first is None:
        first = {k: r[k].clone() for k in keys}
    for k in keys:
        assert torch.equal(r[k], first[k]), (i, k)
print(f"[soak] render 4 x 800x800: {n} calls (1-4 jobs, plain and probabilistic) bit-identical; {float(first['total'][1]) / 2.56e6:.2f} samples per ray", flush=True)
scene2 = SC.make_scene("102344250", n_poses=40)
f0, e0, _ = SI.train_standin(scene2, dev, seed=9)
f1, e1, _ = SI.train_standin(scene2, dev, seed=10)
poses = SI._free_space_poses(scene2, 256, seed=9)
t0 = None
for i in range(max(4, n // 4)):
    V = (256, 32, 40)[i % 3]
    terms, score = RD.score_views([f0, f1], [e0, e1], poses[:V], 640, 640, 320.0, 0.1, 1e-3, 0.1, 0.004, 0.01, dev, group=False)
    if i % 3 == 0:
        if t0 is None:
            t0 = terms.clone()
        assert torch.equal(terms, t0), i
    else:
        assert torch.equal(terms, t0[:V]), (i, V)      # a view's terms do not depend on the batch it is scored in
print(f"[soak] scoring: {max(4, n // 4)} passes of 256 / 32 / 40 views bit-identical rows", flush=True)
