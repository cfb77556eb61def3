"""Two ensemble members trained side by side on ONE GPU (the reference trains its members one after the other inside every iteration,
scripts/pipeline.py:398-412; they are independent models): ms per iteration (= one step of EACH member) with both members on one stream and with
one stream per member, at the reference yaml's 2000 rays and at 8192.
    python tools/exp_ensemble.py [rays,...]"""
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
from apnrf_amd.optim import FusedAdam

shapes = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [2000, 8192]
dev = "cuda:0"
scene = SC.make_scene("102344280", n_poses=40)
members = []
for seed in (11, 12):
    f, e, _ = SI.train_standin(scene, dev, seed=seed)
    members.append((f.train(), e.train(), FusedAdam(f.parameters(), lr=0.0, eps=1e-15).bind_field(f)))
proc = SI._procedural_estimator(scene, dev)
c2w = np.stack([RD.pose_to_c2w(p) for p in scene["poses"][:8]]).astype(np.float32)
K6 = np.array([[320.0, 0, 320], [0, 320.0, 320], [0, 0, 1.0]])
bk = torch.rand(3, device=dev)


def batches(R):
    g = torch.Generator(device="cpu").manual_seed(100)
    out = []
    for k in range(8):
        idx = torch.randint(0, 640 * 640, (R,), generator=g).numpy()
        ys, xs = idx // 640, idx % 640
        idx = idx[np.argsort((ys // 32) * 20 + xs // 32, kind="stable")]
        r = RD.generate_image_rays(torch.from_numpy(c2w[k:k + 1]), 640, 640, K6, dev, idx)
        out.append((r,) + SI.analytic_targets(proc, scene["aabb"], r.origins, r.viewdirs))
    return out


for R in shapes:
    bs = batches(R)
    streams = [torch.cuda.current_stream(dev), torch.cuda.Stream(dev)]
    for label, per_member_stream in (("one stream", False), ("one stream per member", True), ("one stream", False), ("one stream per member", True)):
        def iteration(i):
            for m, (f, e, o) in enumerate(members):
                r, pix, dep_, lab = bs[(i + 3 * m) % 8]
                s = streams[m] if per_member_stream else streams[0]
                with torch.cuda.stream(s):
                    RD.train_step(f, e, o, r, pix, dep_, lab, bk, step=1001 + i, sync=False, occ_thre=1e-2, **SC.RENDER_KW)
        for i in range(6):
            iteration(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(40):
            iteration(i)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 40
        print(f"[exp_ensemble] rays {R}, two members, {label}: {1e3 * dt:.3f} ms per iteration = {1e3 * dt / 2:.3f} ms per member step", flush=True)
