"""Host-side floor of one training iteration: ms per step of `render.train_step` with a batch so small that the GPU work is negligible
(64 rays) — what Python + the C entry points cost per iteration, asynchronous and host-synchronous."""
import os, sys, time
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import apnrf_amd
from apnrf_amd import render as RD, scenes as SC
from apnrf_amd.optim import FusedAdam
dev = "cuda:0"
sc = SC.make_scene(log2_hashmap_size=12)
f, e = SC.hip_field(sc, dev), SC.hip_estimator(sc, dev)
opt = FusedAdam(f.parameters(), lr=1e-4, eps=1e-15).bind_field(f)
c2w = np.stack([RD.pose_to_c2w(p) for p in sc["poses"][:1]]).astype(np.float32)
K6 = np.array([[320.0, 0, 320], [0, 320.0, 320], [0, 0, 1.0]])
r = RD.generate_image_rays(torch.from_numpy(c2w), 640, 640, K6, dev, np.arange(64) * 997)
pix, dep, lab = torch.rand(64, 3, device=dev), torch.rand(64, device=dev), torch.randint(0, 29, (64,), device=dev)
bk = torch.rand(3, device=dev)
for sync in (False, True):
    for i in range(20):
        RD.train_step(f, e, opt, r, pix, dep, lab, bk, step=1001 + i, sync=sync, **SC.RENDER_KW)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(200):
        RD.train_step(f, e, opt, r, pix, dep, lab, bk, step=1001 + i, sync=sync, **SC.RENDER_KW)
    torch.cuda.synchronize()
    print(f"[hostfloor] sync={sync}: {(time.perf_counter() - t0) / 200 * 1e3:.3f} ms per 64-ray step", flush=True)
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for i in range(200):
    RD.train_step(f, e, opt, r, pix, dep, lab, bk, step=1001 + i, sync=False, **SC.RENDER_KW)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
