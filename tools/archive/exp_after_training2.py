"""Which part of an in-process training makes later steps slower?  Cached stand-in; the 8192-ray leg is timed (1) fresh, (2) after allocating and freeing 80 GB
through torch, (3) after 300 asynchronous steps on another field, (4) after 300 deterministic steps on another field, (5) after rendering with a 26 GB workspace."""
import gc, os, sys, time
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import apnrf_amd  # noqa: F401
from apnrf_amd import render as RD, scenes as SC, standin as SI
from apnrf_amd.optim import FusedAdam
from apnrf_amd.nerfacc import OccGridEstimator
dev = "cuda:0"
scene = SC.make_scene("102344280", n_poses=40)
field0, est0, info = SI.train_standin(scene, dev, seed=11)
assert info["cached"], "run once before to fill the cache"
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
state0 = {k: v.clone() for k, v in field0.state_dict().items()}


def fresh():
    tf = SC.hip_field(scene, dev); tf.load_state_dict(state0)
    te = OccGridEstimator(torch.from_numpy(scene["aabb"]), resolution=scene["res"], levels=1).to(dev)
    te.occs.copy_(est0.occs); te.binaries = est0.binaries.clone()
    tf.train(); te.train()
    return tf, te, FusedAdam(tf.parameters(), lr=0.0, eps=1e-15).bind_field(tf)


def steps(tf, te, opt, n, **kw):
    bk = torch.rand(3, device=dev)
    return [RD.train_step(tf, te, opt, *bs[i % 8], bk, step=1001 + i, sync=False, occ_thre=1e-2, **kw, **SC.RENDER_KW) for i in range(n)]


def leg(tag):
    tf, te, opt = fresh()
    steps(tf, te, opt, 8)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    outs = steps(tf, te, opt, 40)
    torch.cuda.synchronize()
    print(f"[{tag}] {(time.perf_counter() - t0) / 40 * 1e3:.3f} ms/step | workspaces MB {[v.numel() >> 20 for v in RD._WORKSPACES.values()]} | torch reserved MB {torch.cuda.memory_reserved() >> 20}", flush=True)


leg("1 fresh")
x = [torch.empty(10 << 30, dtype=torch.uint8, device=dev) for _ in range(8)]
for t in x:
    t.zero_()
torch.cuda.synchronize(); del x, t; gc.collect(); torch.cuda.empty_cache()
leg("2 after 80 GB allocated, written and freed")
tf, te, opt = fresh(); steps(tf, te, opt, 300); torch.cuda.synchronize(); del tf, te, opt; gc.collect()
leg("3 after 300 asynchronous steps on another field")
tf, te, opt = fresh(); steps(tf, te, opt, 300, deterministic=True); torch.cuda.synchronize(); del tf, te, opt; gc.collect()
leg("4 after 300 deterministic steps on another field")
RD._workspace(torch.device(dev), 26 << 30).zero_(); torch.cuda.synchronize()
leg("5 with a 26 GB cached workspace")
RD.release_workspaces(); gc.collect(); torch.cuda.empty_cache()
leg("6 workspace dropped again")
