"""What a field that dies inside a timed region costs (the sporadic 1.7-2x slow train legs of the bench): a field + FusedAdam pair is
dropped (it sits in a reference cycle: only Python's cycle collector frees it), then 10 asynchronous train steps of ANOTHER field are timed three ways:
nothing collected, gc.collect() forced after step 3 (the dead field's mnf_field_destroy = a dozen hipFree calls, each waiting for the device), collected before.
    python tools/exp_gc_stall.py"""
import gc
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import apnrf_amd  # noqa: F401
from apnrf_amd import render as RD, scenes as SC
from apnrf_amd.optim import FusedAdam

dev = "cuda:0"
scene = SC.make_scene("102344280", n_poses=8)
c2w = np.stack([RD.pose_to_c2w(p) for p in scene["poses"][:1]]).astype(np.float32)
K6 = np.array([[320.0, 0, 320], [0, 320.0, 320], [0, 0, 1.0]])
idx = torch.randint(0, 640 * 640, (4096,), generator=torch.Generator().manual_seed(1)).numpy()
r = RD.generate_image_rays(torch.from_numpy(c2w), 640, 640, K6, dev, idx)
pix, dep, lab = torch.rand(4096, 3, device=dev), torch.rand(4096, device=dev) * 3, torch.randint(0, 29, (4096,), device=dev)
bk = torch.zeros(3, device=dev)


def make():
    f, e = SC.hip_field(scene, dev).train(), SC.hip_estimator(scene, dev)
    o = FusedAdam(f.parameters(), lr=1e-4, eps=1e-15).bind_field(f)
    for i in range(3):
        RD.train_step(f, e, o, r, pix, dep, lab, bk, step=1 + i, sync=False, **SC.RENDER_KW)
    torch.cuda.synchronize()
    return f, e, o


def leg(label, collect_at):
    gc.collect(); torch.cuda.synchronize()
    dead = make()
    alive = make()
    n_before = len(gc.get_objects())
    del dead                                   # a cycle keeps it alive until the collector runs
    if collect_at == "before":
        gc.collect(); torch.cuda.synchronize()
    gc.disable()
    t0 = time.perf_counter()
    for i in range(10):
        RD.train_step(*alive, r, pix, dep, lab, bk, step=10 + i, sync=False, **SC.RENDER_KW)
        if collect_at == "inside" and i == 3:
            t1 = time.perf_counter(); n = gc.collect(); t2 = time.perf_counter()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    gc.enable()
    extra = f" (gc.collect() itself: {1e3 * (t2 - t1):.2f} ms host time, {n} objects)" if collect_at == "inside" else ""
    print(f"[gc_stall] {label}: {1e3 * dt / 10:.3f} ms per step over 10 steps{extra}", flush=True)
    del alive


for _ in range(2):
    leg("dead field never collected during the region", "never")
    leg("dead field collected INSIDE the region (after step 3)", "inside")
    leg("dead field collected before the region", "before")
