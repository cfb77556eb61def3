"""View-queue renderer (csrc/viewq.hip) at growing sizes with random-init weights: crash / hang check and a first timing.  python tools/exp_viewq.py [max views]"""
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
sc2 = dict(scene); sc2["params"] = SC.S.make_field_params(seed=1)
f0, f1 = SC.hip_field(scene, dev), SC.hip_field(sc2, dev)
e0, e1 = SC.hip_estimator(scene, dev), SC.hip_estimator(scene, dev)
vmax = int(sys.argv[1]) if len(sys.argv) > 1 else 256
poses = SI._free_space_poses(scene, vmax, seed=9)
for V in (1, 4, 32, 128, vmax):
    if V > vmax:
        continue
    p = poses[:V]
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        terms, score = RD.score_views([f0, f1], [e0, e1], p, 640, 640, 320.0, 0.1, 1e-3, 0.1, 0.004, 0.01, dev, group=False)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    tot = sum(int(t[1]) for t in RD.LAST_SCORE_TOTALS)
    print(f"[exp_viewq] {V} views x 4096 rays x 2 members: {1e3 * dt:.2f} ms, {tot} samples, {tot / dt / 1e9:.2f} G samples/s, score {float(score):.6f}", flush=True)
