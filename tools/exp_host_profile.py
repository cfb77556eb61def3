"""Host-side cost of one train step (GPU box): cProfile of N asynchronous steps at 2000 rays — the time the host needs to enqueue a step is what a host-synchronous
loop (the reference's form) leaves the GPU idle for.
    python tools/exp_host_profile.py [steps]"""
import cProfile
import os
import pstats
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
from apnrf_amd.optim import FusedAdam

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dev = "cuda:0"
scene = SC.make_scene("102344280", n_poses=40)
field, est, info = SI.train_standin(scene, dev, seed=11)
field.train(); est.train()
proc = SI._procedural_estimator(scene, dev)
c2w = np.stack([RD.pose_to_c2w(p) for p in scene["poses"][:8]]).astype(np.float32)
K6 = np.array([[320.0, 0, 320], [0, 320.0, 320], [0, 0, 1.0]])
g = torch.Generator(device="cpu").manual_seed(100)
bs = []
for k in range(8):
    idx = torch.randint(0, 640 * 640, (2000,), generator=g).numpy()
    r = RD.generate_image_rays(torch.from_numpy(c2w[k:k + 1]), 640, 640, K6, dev, idx)
    bs.append((r,) + SI.analytic_targets(proc, scene["aabb"], r.origins, r.viewdirs))
opt = FusedAdam(field.parameters(), lr=0.0, eps=1e-15).bind_field(field)
bk = torch.rand(3, device=dev)


def run(n, sync):
    for i in range(n):
        r, pix, dep_, lab = bs[i % 8]
        RD.train_step(field, est, opt, r, pix, dep_, lab, bk, step=1001 + i, sync=sync, occ_thre=1e-2, **SC.RENDER_KW)


run(10, False)
torch.cuda.synchronize()
t0 = time.perf_counter(); run(steps, False); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"[host] async: enqueue {1e3 * (t1 - t0) / steps:.3f} ms/step on the host, {1e3 * (t2 - t0) / steps:.3f} ms/step wall")
t0 = time.perf_counter(); run(steps, True); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"[host] sync: {1e3 * (t2 - t0) / steps:.3f} ms/step wall")
pr = cProfile.Profile()
pr.enable(); run(steps, False); pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
