"""Where the UNCHANGED caller's step goes (GPU box): scripts/pipeline.py:472-532 typed against the drop-in names (autograd route, torch losses, per-parameter isnan loop,
torch.optim.Adam) at 2000 rays — wall per step, the same loop with the caller's host round trips removed, the render call alone, and a cProfile by cumulative time.
    python tools/exp_dropin_profile.py [steps] [short]"""
import cProfile
import os
import pstats
import sys
import time

import numpy as np
import torch
import torch.nn.functional as F

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import apnrf_amd  # noqa: F401
from apnrf_amd import nerfacc as NA
from apnrf_amd import render as RD
from apnrf_amd import scenes as SC
from apnrf_amd import standin as SI

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
if os.environ.get("DROPIN_CLOSURE_IN_TORCH"):     # A/B: the closure of utils.py:122-137 typed in torch in front of the field's forward (the route before round 5)
    from apnrf_amd.ngp import NGPRadianceField
    del NGPRadianceField.forward_samples_grad
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
optimizer = torch.optim.Adam(field.parameters(), lr=0.0, eps=1e-15, weight_decay=0.0)
occ_eval_fn = NA.FieldDensityOcc(field, 1e-3)
bk = torch.rand(3, device=dev)
KW = dict(near_plane=0.1, render_step_size=1e-3, render_bkgd=bk, cone_angle=0.004, alpha_thre=0.01)


def render(i):
    r, pix, dep_, lab = bs[i % 8]
    est.update_every_n_steps(step=1000 + i, occ_eval_fn=occ_eval_fn, occ_thre=1e-2)
    return RD.render_image_with_occgrid_with_depth_guide(field, est, r, depth=dep_, **KW)


def step(i, caller_syncs=True):
    r, pix, dep_, lab = bs[i % 8]
    rgb, acc, depth, semantic, n = render(i)
    loss_rgb = F.smooth_l1_loss(rgb, pix)
    loss_dep = F.smooth_l1_loss(depth, dep_.unsqueeze(1))
    loss_sem = F.cross_entropy(semantic, lab)
    loss = loss_rgb * 10 + loss_dep / 5 + loss_sem / 2
    if caller_syncs:
        loss_rgb.detach().cpu().item(); loss_dep.detach().cpu().item(); loss_sem.detach().cpu().item()
    optimizer.zero_grad()
    loss.backward()
    if caller_syncs:
        for name, param in field.named_parameters():
            if param.grad is not None and torch.sum(torch.isnan(param.grad)) > 0:
                return
    optimizer.step()


def timed(fn, n):
    for i in range(10):
        fn(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        fn(i)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    return 1e3 * (t1 - t0) / n, 1e3 * (t2 - t0) / n


if len(sys.argv) > 2 and sys.argv[2] == "short":   # for a kernel trace: a dozen caller's steps and out
    for i in range(steps):
        step(i)
    torch.cuda.synchronize()
    sys.exit(0)
h, w = timed(step, steps)
print(f"[dropin] caller's loop: {w:.3f} ms/step wall ({h:.3f} on the host)")
h, w = timed(lambda i: step(i, False), steps)
print(f"[dropin] same without the caller's loss / isnan round trips: {w:.3f} ms/step wall ({h:.3f} on the host)")
h, w = timed(lambda i: render(i), steps)
print(f"[dropin] occupancy refresh + render call alone (graph built, no backward): {w:.3f} ms/step wall ({h:.3f} on the host)")


def fwd_bwd(i):
    rgb, acc, depth, semantic, n = render(i)
    optimizer.zero_grad()
    (rgb.sum() + depth.sum() + semantic.sum()).backward()


h, w = timed(fwd_bwd, steps)
print(f"[dropin] render + backward of a plain sum: {w:.3f} ms/step wall ({h:.3f} on the host)")
h, w = timed(lambda i: optimizer.step(), steps)
print(f"[dropin] torch.optim.Adam.step alone: {w:.3f} ms/step wall ({h:.3f} on the host)")
pr = cProfile.Profile()
pr.enable()
for i in range(steps):
    step(i)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
pstats.Stats(pr).sort_stats("tottime").print_stats(30)
