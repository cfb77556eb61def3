"""Does the train step slow down under sustained load?  Cached stand-in, then 60 x 40 asynchronous 8192-ray steps back to back (about 10 s), ms/step of every block of 40;
sclk from rocm-smi before and after."""
import os, subprocess, sys, time
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import apnrf_amd  # noqa: F401
from apnrf_amd import render as RD, scenes as SC, standin as SI
from apnrf_amd.optim import FusedAdam
dev = "cuda:0"
scene = SC.make_scene("102344280", n_poses=40)
field, est, info = SI.train_standin(scene, dev, seed=11)
print("stand-in cached:", info["cached"], flush=True)
if not info["cached"]:
    print("(trained in this process: rerun for the cached case)")
proc = SI._procedural_estimator(scene, dev)
c2w = np.stack([RD.pose_to_c2w(p) for p in scene["poses"][:8]]).astype(np.float32)
K6 = np.array([[320.0, 0, 320], [0, 320.0, 320], [0, 0, 1.0]])
g = torch.Generator(device="cpu").manual_seed(100)
bs = []
for k in range(8):
    idx = torch.randint(0, 640 * 640, (8192,), generator=g).numpy()
    ys, xs = idx // 640, idx % 640
    idx = idx[np.argsort((ys // 32) * 20 + xs // 32, kind="stable")]
    r = RD.generate_image_rays(torch.from_numpy(c2w[k:k + 1]), 640, 640, K6, dev, idx)
    bs.append((r,) + SI.analytic_targets(proc, scene["aabb"], r.origins, r.viewdirs))
field.train(); est.train()
opt = FusedAdam(field.parameters(), lr=0.0, eps=1e-15).bind_field(field)
bk = torch.rand(3, device=dev)
smi = lambda: subprocess.run("rocm-smi --showclocks --showpower 2>/dev/null | grep -i 'sclk\\|power' | head -3", shell=True, capture_output=True, text=True).stdout.replace("\n", " | ")
for i in range(8):
    RD.train_step(field, est, opt, *bs[i % 8], bk, step=1001 + i, sync=False, occ_thre=1e-2, **SC.RENDER_KW)
torch.cuda.synchronize()
print(smi(), flush=True)
out = []
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 60):
    t0 = time.perf_counter()
    for i in range(40):
        RD.train_step(field, est, opt, *bs[i % 8], bk, step=1001 + i, sync=False, occ_thre=1e-2, **SC.RENDER_KW)
    torch.cuda.synchronize()
    out.append((time.perf_counter() - t0) / 40 * 1e3)
    if rep % 10 == 9:
        print(f"blocks {rep - 9}-{rep}: " + " ".join(f"{x:.2f}" for x in out[-10:]), "|", smi(), flush=True)
