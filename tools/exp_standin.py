"""Experiment: train the analytic stand-in (standin.train_standin) and look at what the trained scene renders like."""
import os, sys, time
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import __graft_entry__ as G
G.build()
import helpers as H
from apnrf_amd import render as RD, standin as SI
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
lr = float(sys.argv[2]) if len(sys.argv) > 2 else 1e-2
name = sys.argv[3] if len(sys.argv) > 3 else "102344529"
sc = H.make_scene(name, n_poses=40)
t0 = time.time()
field, est, info = SI.train_standin(sc, "cuda:0", steps=steps, lr=lr, verbose=True, cache_dir=os.environ.get("STANDIN_CACHE", "/tmp"))
print("train", time.time() - t0, info)
c2w = np.stack([RD.pose_to_c2w(p) for p in sc["poses"][[3, 11, 22, 35]]]).astype(np.float32)
K = np.array([[400.0, 0, 400], [0, 400.0, 400], [0, 0, 1.0]])
rays = RD.generate_image_rays(torch.from_numpy(c2w), 800, 800, K, "cuda:0")
o, d = rays.origins.reshape(-1, 3), rays.viewdirs.reshape(-1, 3)
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.time()
    r = RD.render_views(field, est, o, d, 640000, 1024, render_bkgd=torch.zeros(3), image_hw=(800, 800), **H.RENDER_KW)
    torch.cuda.synchronize(); dt = time.time() - t0
    print(f"render 4 views: {dt*1e3:.2f} ms, evaluated/ray {int(r['total'][1])/o.shape[0]:.1f}, kept/ray {int(r['total'][0])/o.shape[0]:.1f}, acc mean {float(r['acc'].mean()):.3f}")
# quality vs the analytic target
proc = SI._procedural_estimator(sc, "cuda:0")
pix, dep, lab = SI.analytic_targets(proc, sc["aabb"], o[:640000], d[:640000])
mse = float(((r["rgb"][:640000] - pix) ** 2).mean())
print("PSNR vs analytic target", 10 * np.log10(1 / mse), "depth L1", float((r["depth"][:640000, 0] - dep).abs().mean()),
      "sem acc", float((r["sem"][:640000].argmax(-1) == lab).float().mean()))
