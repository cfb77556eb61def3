"""Experiment: does the order of the rays of a dense 800x800 view matter?  Row-major (the reference's order) against
block x block pixel tiles."""
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
V = 4
def morton(ys, xs):
    k = np.zeros_like(ys)
    for b in range(10):
        k |= ((xs >> b) & 1) << (2 * b)
        k |= ((ys >> b) & 1) << (2 * b + 1)
    return k
for block in (0, 8, -1, -2):
    if block == -1 or block == -2:
        ys, xs = np.meshgrid(np.arange(800), np.arange(800), indexing="ij")
        if block == -1: key = morton(ys, xs)
        else: key = ((ys // 8) * 100 + xs // 8) * 64 + morton(ys % 8, xs % 8)     # 8x8 blocks, Z-order inside
        perm = torch.from_numpy(np.argsort(key.reshape(-1), kind="stable")).to(dev)
        o = rays.origins[:V][:, perm].reshape(-1, 3).contiguous(); d = rays.viewdirs[:V][:, perm].reshape(-1, 3).contiguous()
    elif block:
        ys, xs = np.meshgrid(np.arange(800), np.arange(800), indexing="ij")
        key = ((ys // block) * (800 // block) + xs // block) * (block * block) + (ys % block) * block + xs % block
        perm = torch.from_numpy(np.argsort(key.reshape(-1), kind="stable")).to(dev)
        o = rays.origins[:V][:, perm].reshape(-1, 3).contiguous(); d = rays.viewdirs[:V][:, perm].reshape(-1, 3).contiguous()
    else:
        o = rays.origins[:V].reshape(-1, 3).contiguous(); d = rays.viewdirs[:V].reshape(-1, 3).contiguous()
    for i in range(3):
        RD.render_views(field, est, o, d, 640000, 1024, render_bkgd=bk, **H.RENDER_KW)
    torch.cuda.synchronize(); t = time.perf_counter()
    for i in range(4):
        out = RD.render_views(field, est, o, d, 640000, 1024, render_bkgd=bk, **H.RENDER_KW)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 4
    print(f"block={block}: {dt*1e3:.2f} ms/call  {V*640000/dt/1e6:.1f} Mrays/s  evaluated/ray={float(out['total'][1])/(V*640000):.2f}")
