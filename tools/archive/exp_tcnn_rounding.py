"""Experiment: how far does the tcnn-faithful output rounding (network outputs rounded to fp16, ngp.py:181-220) move the RENDERED
values from the default fp32-output mode?  BASELINE configs 2 (scene 102344250, 256x256, 4x64, C=29) and 3 (scene 102344529,
800x800 geometry, 128x2, C=29): max / mean |delta| of rgb, depth, acc, semantics, and PSNR between the two modes."""
import os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import __graft_entry__ as G
G.build()
import helpers as H
from apnrf_amd import render as RD
for name, kw, wh in (("config 2", dict(scene="102344250", neurons=64, layers=4, C=29, seed=5), 256), ("config 3", dict(scene="102344529", neurons=128, layers=2, C=29, seed=0), 800)):
    sc = H.make_scene(**kw)
    est = H.hip_estimator(sc)
    c2w = RD.pose_to_c2w(sc["poses"][3]).astype(np.float32)[None]
    K = np.array([[wh / 2, 0, wh / 2], [0, wh / 2, wh / 2], [0, 0, 1.0]])
    rays = RD.generate_image_rays(torch.from_numpy(c2w), wh, wh, K, "cuda:0")
    outs = []
    for mode in (False, True):
        f = H.hip_field(sc, tcnn_output_rounding=mode)
        outs.append(RD.render_views(f, est, rays.origins, rays.viewdirs, wh * wh, 1024, render_bkgd=torch.zeros(3), **H.RENDER_KW))
    a, b = outs
    rep = {k: (float((a[k] - b[k]).abs().max()), float((a[k] - b[k]).abs().mean())) for k in ("rgb", "acc", "depth", "sem")}
    mse = float(((a["rgb"] - b["rgb"]) ** 2).mean())
    print(name, {k: f"max {v[0]:.2e} mean {v[1]:.2e}" for k, v in rep.items()}, f"PSNR between modes {10 * np.log10(1 / mse):.1f} dB",
          f"|sem| max {float(a['sem'].abs().max()):.2f}", f"samples {int(a['total'][1])} vs {int(b['total'][1])}")
