import sys, time, numpy as np, torch
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import helpers as H
from apnrf_amd import render as RD
dev='cuda:0'
scene = H.make_scene("102344529", n_poses=8)
field, est = H.hip_field(scene, dev), H.hip_estimator(scene, dev)
c2w = np.stack([RD.pose_to_c2w(p) for p in scene["poses"]]).astype(np.float32)
K = np.array([[400.0,0,400],[0,400.0,400],[0,0,1.0]])
rays = RD.generate_image_rays(torch.from_numpy(c2w), 800, 800, K, dev)
bk=torch.zeros(3)
def loop(idx, use_ev, tag):
    for i in range(2): RD.render_views(field, est, rays.origins[i], rays.viewdirs[i], 640000, 1024, render_bkgd=bk, **H.RENDER_KW)
    torch.cuda.synchronize(); t=time.perf_counter()
    ev = torch.zeros((), dtype=torch.int64, device=dev)
    for i in idx:
        out=RD.render_views(field, est, rays.origins[i%8], rays.viewdirs[i%8], 640000, 1024, render_bkgd=bk, **H.RENDER_KW)
        if use_ev: ev += out["total"][1]
    torch.cuda.synchronize(); print(tag, (time.perf_counter()-t)/len(idx)*1e3, "ms/step")
loop(range(8), False, "8 steps, no ev")
loop(range(8), True, "8 steps, ev")
loop(range(3,13), False, "10 steps (3..12), no ev")
loop(range(3,13), True, "10 steps (3..12), ev")
loop(range(16), False, "16 steps, no ev")
loop(range(8), False, "8 steps, no ev")
