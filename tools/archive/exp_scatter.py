"""Scatter timing experiment (GPU box, diag library): hash_scatter / wgrad / dgrad milliseconds of the fused forward+backward at BASELINE config 5's
shape on the trained stand-in, under whatever MNF_* diagnostic knobs the environment sets (MNF_HASH_BWD_LEVELS=lo,hi  MNF_NO_WGRAD=1 ...).
    MNF_LIB_PATH=.../libmi355nerf_diag.so python tools/exp_scatter.py [rays] [reps]"""
import ctypes
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import apnrf_amd  # noqa: F401
from apnrf_amd import _lib as L
from apnrf_amd import render as RD
from apnrf_amd import scenes as SC
from apnrf_amd import standin as SI

R = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = "cuda:0"
scene = SC.make_scene("102344280", n_poses=40)
field, est, info = SI.train_standin(scene, dev, seed=11)
field.train(); est.train()
proc = SI._procedural_estimator(scene, dev)
c2w = np.stack([RD.pose_to_c2w(p) for p in scene["poses"][:8]]).astype(np.float32)
K6 = np.array([[320.0, 0, 320], [0, 320.0, 320], [0, 0, 1.0]])
g = torch.Generator(device="cpu").manual_seed(100)
idx = torch.randint(0, 640 * 640, (R,), generator=g).numpy()
ys, xs = idx // 640, idx % 640
idx = idx[np.argsort((ys // 32) * 20 + xs // 32, kind="stable")]
r = RD.generate_image_rays(torch.from_numpy(c2w[0:1]), 640, 640, K6, dev, idx)
pix, dep_, lab = SI.analytic_targets(proc, scene["aabb"], r.origins, r.viewdirs)
kw = dict(SC.RENDER_KW)
bk = torch.rand(3, device=dev)
lib = L.load_library()
for i in range(3):
    out = RD.fused_forward_backward(field, est, r, pix, dep_, lab, bk, **kw)
torch.cuda.synchronize()
lib.mnf_profile_begin()
for i in range(reps):
    out = RD.fused_forward_backward(field, est, r, pix, dep_, lab, bk, **kw)
torch.cuda.synchronize()
ms, n = ctypes.c_double(0), ctypes.c_int64(0)
lib.mnf_profile_end(ctypes.byref(ms), ctypes.byref(n))
line = [f"kept {int(out['n_rendering_samples'])}"]
for label in ("field_train_forward", "dgrad", "wgrad", "hash_scatter", "hash_scatter_bins"):
    lib.mnf_profile_query(label.encode(), ctypes.byref(ms), ctypes.byref(n))
    line.append(f"{label} {ms.value / max(n.value, 1):.3f} ms")
print(os.environ.get("TAG", ""), " | ".join(line), flush=True)
