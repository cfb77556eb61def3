"""Why is the train leg slower in the process that trained the stand-in itself (first bench run on a fresh box: 5.2 ms against 3.9 ms with the cached stand-in)?
Times the same 8192-ray asynchronous steps (a) right after the in-process stand-in training, (b) after dropping the cached workspaces / allocator blocks,
(c) after destroying the stand-in's own field (its handle owns the deterministic-mode buffers)."""
import gc, os, sys, time
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import apnrf_amd  # noqa: F401
from apnrf_amd import render as RD, scenes as SC, standin as SI
from apnrf_amd.optim import FusedAdam
dev = "cuda:0"
scene = SC.make_scene("102344280", n_poses=40)
cached = len(sys.argv) > 1 and sys.argv[1] == "cached"
field0, est0, info = SI.train_standin(scene, dev, seed=11, keep_optimizer=True, cache_dir=None if cached else f"/tmp/force_{os.getpid()}")
print("stand-in cached:", info["cached"], "| workspaces MB:", {str(k): v.numel() >> 20 for k, v in RD._WORKSPACES.items()}, "| torch reserved MB", torch.cuda.memory_reserved() >> 20, flush=True)
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


def leg(tag):
    from apnrf_amd.nerfacc import OccGridEstimator
    tf = SC.hip_field(scene, dev); tf.load_state_dict(field0_state)
    te = OccGridEstimator(torch.from_numpy(scene["aabb"]), resolution=scene["res"], levels=1).to(dev)
    te.occs.copy_(occs0); te.binaries = bin0.clone()
    tf.train(); te.train()
    opt = FusedAdam(tf.parameters(), lr=0.0, eps=1e-15).bind_field(tf)
    bk = torch.rand(3, device=dev)
    for i in range(8):
        RD.train_step(tf, te, opt, *bs[i % 8], bk, step=1001 + i, sync=False, occ_thre=1e-2, **SC.RENDER_KW)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    outs = [RD.train_step(tf, te, opt, *bs[i % 8], bk, step=1001 + i, sync=False, occ_thre=1e-2, **SC.RENDER_KW) for i in range(40)]
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 40
    print(f"[{tag}] {1e3 * dt:.3f} ms/step, kept {np.mean([int(o['n_rendering_samples']) for o in outs]):.0f} | workspaces MB {[v.numel() >> 20 for v in RD._WORKSPACES.values()]} | torch reserved MB {torch.cuda.memory_reserved() >> 20} | free GPU MB {torch.cuda.mem_get_info()[0] >> 20}", flush=True)


field0_state = {k: v.clone() for k, v in field0.state_dict().items()}
occs0, bin0 = est0.occs.clone(), est0.binaries.clone()
leg("a: right after the stand-in")
leg("a2: again")
RD.release_workspaces(); gc.collect(); torch.cuda.empty_cache()
leg("b: workspaces and allocator blocks dropped")
del field0, est0; gc.collect(); torch.cuda.empty_cache()
leg("c: stand-in's field destroyed")
