"""Per-round profile of one 800x800 render call on the bench's trained scene (GPU box, diagnostic library):
    MNF_LIB_PATH=$PWD/active-perception-using-neural-radiance-fields_amd/libmi355nerf_diag.so MNF_ROUND_LOG=1 python tools/exp_round_log.py [views] 2> rounds.txt
stderr: one "[mnf round k] cols .. field .. ms  march .. ms  active_views .. alive_after .. budgets .." line per round (MNF_ROUND_LOG synchronises every round)."""
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

V = int(sys.argv[1]) if len(sys.argv) > 1 else 2
dev = "cuda:0"
if len(sys.argv) > 2 and sys.argv[2] == "score":      # one render job of a scoring pass: V views of 64 x 64 sub-sampled rays (BASELINE config 4's scene), probabilistic
    scene = SC.make_scene("102344250", n_poses=40)
    field, est, _ = SI.train_standin(scene, dev, seed=9)
    field.eval(); est.eval()
    poses = SI._free_space_poses(scene, 256, seed=9)[:V]
    o, d, h, w = RD._pose_rays(poses, 640, 640, 320.0, 0.1, dev)
    outs = RD._render_jobs([(field, est, o, d)], h * w, 1024, 0.1, 1e10, 1e-3, torch.zeros(3), 0.004, 0.01, 1e-4, True, 8, None, 1)
    torch.cuda.synchronize()
    print(f"[exp_round_log] score job, {V} views of {h * w} rays: {float(outs[0]['total'][1]) / (V * h * w):.2f} evaluated samples per ray", flush=True)
    sys.exit(0)
scene = SC.make_scene("102344529", n_poses=40)
field, est, info = SI.shared_standin(scene, dev, steps=2000, seed=9, keep_optimizer=False, group=False)
field.eval(); est.eval()
poses = scene["poses"][[5 * k % 40 for k in range(V)]]
c2w = np.stack([RD.pose_to_c2w(p) for p in poses]).astype(np.float32)
K = np.array([[400.0, 0, 400], [0, 400.0, 400], [0, 0, 1.0]])
rays = RD.generate_image_rays(torch.from_numpy(c2w), 800, 800, K, dev)
o, d = rays.origins.reshape(-1, 3).contiguous(), rays.viewdirs.reshape(-1, 3).contiguous()
r = RD.render_views(field, est, o, d, 640000, 1024, render_bkgd=torch.zeros(3), image_hw=(800, 800), n_split=1, **SC.RENDER_KW)
torch.cuda.synchronize()
print(f"[exp_round_log] {V} views: {float(r['total'][1]) / (V * 640000):.2f} evaluated samples per ray", flush=True)
