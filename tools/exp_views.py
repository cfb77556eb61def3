"""Experiment: ms per 800x800 view when V views are rendered in one batched call (same per-view round schedule)."""
import sys, time, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import helpers as H
from apnrf_amd import render as RD
dev = 'cuda:0'
scene = H.make_scene("102344529", n_poses=8)
field, est = H.hip_field(scene, dev), H.hip_estimator(scene, dev)
c2w = np.stack([RD.pose_to_c2w(p) for p in scene["poses"]]).astype(np.float32)
K = np.array([[400.0, 0, 400], [0, 400.0, 400], [0, 0, 1.0]])
rays = RD.generate_image_rays(torch.from_numpy(c2w), 800, 800, K, dev)
bk = torch.zeros(3)
for V in ([int(a) for a in sys.argv[1:]] or (1, 2, 4, 8)):
    o = rays.origins[:V].reshape(-1, 3).contiguous(); d = rays.viewdirs[:V].reshape(-1, 3).contiguous()
    for i in range(3):
        RD.render_views(field, est, o, d, 640000, 1024, render_bkgd=bk, image_hw=(800, 800), **H.RENDER_KW)
    torch.cuda.synchronize(); t = time.perf_counter()
    n = max(2, 16 // V)
    for i in range(n):
        out = RD.render_views(field, est, o, d, 640000, 1024, render_bkgd=bk, image_hw=(800, 800), **H.RENDER_KW)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / n
    print(f"V={V}: {dt*1e3:.2f} ms/call  {dt*1e3/V:.2f} ms/view  {V*640000/dt/1e6:.1f} Mrays/s  evaluated/ray={float(out['total'][1])/(V*640000):.2f}")
