"""Bisect: what in `standin.train_standin` leaves the process with slower train steps?  usage: exp_after_training3.py <variant>
variants: full300 (train_standin, 300 steps) | full50 | nodet (deterministic off) | fixedrays (8192 rays from the start: no resizing)
          | targets (only the analytic targets + ray generation of 300 steps, no training) | early (300 steps of a RANDOM-INIT field at 8192 rays, asynchronous)"""
import gc, os, sys, time
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import apnrf_amd  # noqa: F401
from apnrf_amd import render as RD, scenes as SC, standin as SI
from apnrf_amd.optim import FusedAdam
from apnrf_amd.nerfacc import OccGridEstimator
dev = "cuda:0"
variant = sys.argv[1]
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


def leg(tag):
    tf = SC.hip_field(scene, dev); tf.load_state_dict(state0)
    te = OccGridEstimator(torch.from_numpy(scene["aabb"]), resolution=scene["res"], levels=1).to(dev)
    te.occs.copy_(est0.occs); te.binaries = est0.binaries.clone()
    tf.train(); te.train()
    opt = FusedAdam(tf.parameters(), lr=0.0, eps=1e-15).bind_field(tf)
    bk = torch.rand(3, device=dev)
    run = lambda n: [RD.train_step(tf, te, opt, *bs[i % 8], bk, step=1001 + i, sync=False, occ_thre=1e-2, **SC.RENDER_KW) for i in range(n)]
    run(8)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    run(40)
    torch.cuda.synchronize()
    import subprocess
    smi = subprocess.run("rocm-smi --showclocks --showpower 2>/dev/null | grep -i 'sclk\\|Power (W)' | head -2", shell=True, capture_output=True, text=True).stdout.replace("\n", " | ")
    x = torch.zeros(1024, device=dev)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(500):
        x.add_(1.0)
    b.record(); b.synchronize()
    print("   tiny torch kernels: %.2f us each |" % (a.elapsed_time(b) / 500 * 1e3), smi, "| caps", {k: (v["abs_m"], v["abs_k"]) for k, v in RD._TRAIN_STATE.items()}, flush=True)
    print(f"[{variant}: {tag}] {(time.perf_counter() - t0) / 40 * 1e3:.3f} ms/step | workspaces MB {[v.numel() >> 20 for v in RD._WORKSPACES.values()]} | torch reserved MB {torch.cuda.memory_reserved() >> 20} | train states {len(RD._TRAIN_STATE)}", flush=True)


leg("before")
force = f"/tmp/force_{os.getpid()}"
if variant == "full300":
    SI.train_standin(scene, dev, seed=11, steps=300, cache_dir=force)
elif variant == "full50":
    SI.train_standin(scene, dev, seed=11, steps=50, cache_dir=force)
elif variant == "nodet":
    orig = RD.train_step
    RD.train_step = lambda *a, **k: orig(*a, **{**k, "deterministic": False})
    SI.train_standin(scene, dev, seed=11, steps=300, cache_dir=force)
    RD.train_step = orig
elif variant == "fixedrays":
    SI.train_standin(scene, dev, seed=11, steps=300, cache_dir=force, target_samples=1 << 40)      # n_rays grows to max_rays at once and stays
elif variant == "targets":
    gen = torch.Generator().manual_seed(3)
    for step in range(300):
        idx = torch.randint(0, 640 * 640, (1024 + 16 * step,), generator=gen).numpy()
        rays = RD.generate_image_rays(torch.from_numpy(c2w[:1]), 640, 640, K6, dev, idx)
        SI.analytic_targets(proc, scene["aabb"], rays.origins, rays.viewdirs)
elif variant.startswith("alive"):       # k extra fields with a train state each (5 steps on the trained weights), kept alive
    keep = []
    for j in range(int(variant.split(",")[1]) if "," in variant else 1):
        f2 = SC.hip_field(scene, dev); f2.load_state_dict(state0); f2.train()
        e2 = OccGridEstimator(torch.from_numpy(scene["aabb"]), resolution=scene["res"], levels=1).to(dev)
        e2.occs.copy_(est0.occs); e2.binaries = est0.binaries.clone(); e2.train()
        o2 = FusedAdam(f2.parameters(), lr=0.0, eps=1e-15).bind_field(f2)
        bk = torch.rand(3, device=dev)
        for i in range(5):
            RD.train_step(f2, e2, o2, *bs[i % 8], bk, step=1001 + i, sync=True, occ_thre=1e-2, **SC.RENDER_KW)
        keep.append((f2, e2, o2))
elif variant.startswith("early"):
    from apnrf_amd.ngp import NGPRadianceField
    f = NGPRadianceField(aabb=torch.from_numpy(scene["aabb"]), neurons=scene["neurons"], layers=scene["layers"], num_semantic_classes=scene["C"],
                         log2_hashmap_size=scene["log2_hashmap_size"], seed=11).to(dev).train()
    e = OccGridEstimator(torch.from_numpy(scene["aabb"]), resolution=scene["res"], levels=1).to(dev).train()
    o = FusedAdam(f.parameters(), lr=2e-3, eps=1e-15).bind_field(f)
    bk = torch.rand(3, device=dev)
    opts = dict(x.split("=") for x in variant.split(",")[1:])          # e.g. early,steps=60,det=1,sync=0,rays=2000,noocc=1
    R_ = int(opts.get("rays", 8192))
    for i in range(int(opts.get("steps", 300))):
        r_, pix_, dep_, lab_ = bs[i % 8]
        if R_ != 8192:
            r_, pix_, dep_, lab_ = RD.Rays(r_.origins[:R_], r_.viewdirs[:R_]), pix_[:R_], dep_[:R_], lab_[:R_]
        RD.train_step(f, e, o, r_, pix_, dep_, lab_, bk, step=(i if not int(opts.get("noocc", 0)) else 16 * i + 1), sync=bool(int(opts.get("sync", 1))), occ_thre=1e-2,
                      deterministic=bool(int(opts.get("det", 0))), fused=bool(int(opts.get("fused", 1))), **SC.RENDER_KW)
torch.cuda.synchronize()
leg("after")
gc.collect(); RD.release_workspaces(); torch.cuda.empty_cache()
leg("after, caches dropped")
