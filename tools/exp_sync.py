import sys, time, numpy as np, torch
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import helpers as H
from apnrf_amd import render as RD
dev='cuda:0'
scene = H.make_scene("102344529", n_poses=8)
field, est = H.hip_field(scene, dev), H.hip_estimator(scene, dev)
focal = 400.0
c2w = np.stack([RD.pose_to_c2w(p) for p in scene["poses"]]).astype(np.float32)
K = np.array([[focal,0,400],[0,focal,400],[0,0,1.0]])
rays = RD.generate_image_rays(torch.from_numpy(c2w), 800, 800, K, dev)
bk=torch.zeros(3)
def run(se, steps=8):
    for i in range(2): RD.render_views(field, est, rays.origins[i], rays.viewdirs[i], 640000, 1024, render_bkgd=bk, sync_every=se, **H.RENDER_KW)
    torch.cuda.synchronize(); t=time.perf_counter()
    for i in range(steps): out=RD.render_views(field, est, rays.origins[i%8], rays.viewdirs[i%8], 640000, 1024, render_bkgd=bk, sync_every=se, **H.RENDER_KW)
    torch.cuda.synchronize(); dt=(time.perf_counter()-t)/steps
    print(f"sync_every={se}: {dt*1e3:.2f} ms/step  {640000/dt/1e6:.1f} Mrays/s")
for se in (8, 4, 16, 32, 0, 8):
    run(se)
