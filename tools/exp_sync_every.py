"""How often the host looks at a render job (mnf_render_opts.sync_every: rounds enqueued per block) against pass time, for a latency-bound batch (32 scoring views x 2 members = a rank's share of an
8-GPU pass) and the full 256-view pass; trained stand-ins (cached after the first run).  python tools/exp_sync_every.py"""
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
scene = SC.make_scene("102344250", n_poses=40)
f0, e0, _ = SI.train_standin(scene, dev, seed=9)
f1, e1, _ = SI.train_standin(scene, dev, seed=10)
poses = SI._free_space_poses(scene, 256, seed=9)


def run(V, sync_every, n_split, reps=5):
    o, d, h, w = RD._pose_rays(poses[:V], 640, 640, 320.0, 0.1, dev)
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        RD._render_jobs([(f0, e0, o, d), (f1, e1, o, d)], h * w, 1024, 0.1, 1e10, 1e-3, torch.zeros(3), 0.004, 0.01, 1e-4, True, sync_every, None, n_split)
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return 1e3 * float(np.median(ts))


for V in (32, 256):
    for n_split in (2,):
        for se in (2, 4, 8, 16, 32, 64, 0):
            print(f"[exp_sync_every] {V} views x 2 members, {2 * n_split} jobs, sync_every {se}: {run(V, se, n_split):.2f} ms", flush=True)
