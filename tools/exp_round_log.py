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
