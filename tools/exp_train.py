"""Train-step timing experiment (GPU box): ms per step for the synchronous (reference semantics: one host round trip at the end)
and the fully asynchronous form, at BASELINE config 5's shape (8192 rays) and at the reference yaml's (2000 rays / ~262 k samples).
    python tools/exp_train.py [f16|bf16] [steps] [lr] [shapes] [sync modes] [backward mode] [presample 0|1] [model: NEURONSxLAYERS@SCENE@IMAGE/SEED, default 128x2@102344280@640/11]
    (64x4@102344250@256/9 = BASELINE config 2's train leg of bench.py)"""
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

dtype = sys.argv[1] if len(sys.argv) > 1 else "f16"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
lr = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0          # 0: the parameters (hence the sample counts) stay put across the timed steps
shapes = [int(x) for x in sys.argv[4].split(",")] if len(sys.argv) > 4 else [8192, 2000]
modes = [bool(int(x)) for x in sys.argv[5].split(",")] if len(sys.argv) > 5 else [True, False]
bwd_mode = 0                                                     # (argv[6] was the backward mode of round 4's fused-backward experiment: tools/experiments/fused_backward.patch)
presample = bool(int(sys.argv[7])) if len(sys.argv) > 7 else False   # march batch k+1 beside step k (render.presample)
model = sys.argv[8] if len(sys.argv) > 8 else "128x2@102344280@640/11"
(_shape, _scene, _rest) = model.split("@")
_im, _seed = (int(x) for x in _rest.split("/"))
_neurons, _layers = (int(x) for x in _shape.split("x"))
dev = "cuda:0"
scene = SC.make_scene(_scene, n_poses=40, neurons=_neurons, layers=_layers)
field, est, info = SI.train_standin(scene, dev, seed=_seed)     # the stand-in of bench.py's train legs
print("[exp_train] stand-in:", {k: v for k, v in info.items() if k != "optimizer_state"}, flush=True)
if dtype == "bf16":
    f2 = SC.hip_field(scene, dev, mfma_bf16=True)
    f2.load_state_dict(field.state_dict())
    field = f2
field.train(); est.train()
proc = SI._procedural_estimator(scene, dev)
c2w = np.stack([RD.pose_to_c2w(p) for p in scene["poses"][:8]]).astype(np.float32)
K6 = np.array([[_im / 2.0, 0, _im / 2.0], [0, _im / 2.0, _im / 2.0], [0, 0, 1.0]])
_blk = max(1, _im // 20)


def batches(R):
    g = torch.Generator(device="cpu").manual_seed(100)
    out = []
    for k in range(8):
        idx = torch.randint(0, _im * _im, (R,), generator=g).numpy()
        ys, xs = idx // _im, idx % _im
        idx = idx[np.argsort((ys // _blk) * 20 + xs // _blk, kind="stable")]
        r = RD.generate_image_rays(torch.from_numpy(c2w[k:k + 1]), _im, _im, K6, dev, idx)
        out.append((r,) + SI.analytic_targets(proc, scene["aabb"], r.origins, r.viewdirs))
    return out


for R in shapes:
    bs = batches(R)
    for sync in modes:
        opt = FusedAdam(field.parameters(), lr=lr, eps=1e-15).bind_field(field)
        bk = torch.rand(3, device=dev)
        outs = []
        for i in range(5):
            r, pix, dep_, lab = bs[i % 8]
            RD.train_step(field, est, opt, r, pix, dep_, lab, bk, step=1001 + i, sync=sync, occ_thre=1e-2, **SC.RENDER_KW)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        tok = RD.presample(field, est, bs[0][0], **SC.RENDER_KW) if presample else None
        for i in range(steps):
            r, pix, dep_, lab = bs[i % 8]
            nxt = RD.presample(field, est, bs[(i + 1) % 8][0], **SC.RENDER_KW) if presample and (1001 + i) % 16 and (1001 + i + 1) % 16 else None      # (not across the occupancy refresh of step 1008 + 16 j)
            outs.append(RD.train_step(field, est, opt, r, pix, dep_, lab, bk, step=1001 + i, sync=sync, occ_thre=1e-2, presampled=tok, **SC.RENDER_KW))
            tok = nxt
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        kept = np.mean([int(o["n_rendering_samples"]) for o in outs])
        skipped = sum(int(o["skipped"]) for o in outs)
        print(f"[exp_train] {dtype} rays {R} sync={sync} bwd_mode={bwd_mode} presample={presample}: {1e3 * dt:.3f} ms/step, kept {kept:.0f}, skipped {skipped}", flush=True)
