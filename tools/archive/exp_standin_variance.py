"""How reproducible is the trained stand-in?  Trains the render scene several times per protocol (no cache) and renders one
800x800 view each: samples per ray, loss, occupied cells.  usage: python tools/exp_standin_variance.py [reps]"""
import sys, tempfile, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import helpers as H
from apnrf_amd import render as RD, standin as SI
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
dev = 'cuda:0'
scene = H.make_scene("102344529", n_poses=40)
c2w = np.stack([RD.pose_to_c2w(p) for p in scene["poses"][[0, 5]]]).astype(np.float32)
K = np.array([[400.0, 0, 400], [0, 400.0, 400], [0, 0, 1.0]])
rays = RD.generate_image_rays(torch.from_numpy(c2w), 800, 800, K, dev)
for name, kw in (("lr 2e-3 constant", {"lr_final": None}), ("lr 2e-3 -> 2e-4 over the second half", {"lr_final": 2e-4}), ("lr 2e-3 -> 5e-5", {"lr_final": 5e-5})):
    out = []
    for r in range(reps):
        field, est, info = SI.train_standin(scene, dev, cache_dir=tempfile.mkdtemp(), **kw)
        res = RD.render_views(field, est, rays.origins.reshape(-1, 3), rays.viewdirs.reshape(-1, 3), 640000, 1024, render_bkgd=torch.zeros(3),
                              image_hw=(800, 800), **H.RENDER_KW)
        out.append((float(res["total"][1]) / 1280000, info["loss_last"], info["occupied_cells"]))
    spr = np.array([o[0] for o in out])
    print(f"{name}: samples/ray {np.round(spr, 1).tolist()} (spread {100 * (spr.max() - spr.min()) / spr.mean():.0f} %)  loss {[round(o[1], 3) for o in out]}  occupied {[o[2] for o in out]}", flush=True)
