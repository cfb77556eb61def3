"""Small-kernel time before / after 300 training steps of a random-init field at 8192 rays in the same process (the state in which every small launch of a train
step takes ~45 us instead of ~5 us).  Reports the GPU time of 2000 back-to-back tiny torch kernels (hipEvents) on the current stream and on a new stream."""
import gc, os, sys, time
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import apnrf_amd  # noqa: F401
from apnrf_amd import render as RD, scenes as SC, standin as SI
from apnrf_amd.optim import FusedAdam
from apnrf_amd.nerfacc import OccGridEstimator
dev = "cuda:0"
mode = sys.argv[1] if len(sys.argv) > 1 else "train"
x = torch.zeros(1024, device=dev)


def tiny(tag, stream=None):
    s = stream or torch.cuda.current_stream()
    with torch.cuda.stream(s):
        for _ in range(100):
            x.add_(1.0)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(s)
        for _ in range(2000):
            x.add_(1.0)
        b.record(s)
    b.synchronize()
    print(f"[{mode}: {tag}] {a.elapsed_time(b) / 2000 * 1e3:.2f} us per tiny kernel | free GPU MB {torch.cuda.mem_get_info()[0] >> 20} | torch reserved MB {torch.cuda.memory_reserved() >> 20}", flush=True)


tiny("start")
scene = SC.make_scene("102344280", n_poses=40)
if mode == "alloc":            # only the allocation pattern: growing hipMallocs through torch, no kernels of this library
    bufs = []
    for i in range(40):
        bufs.append(torch.empty((i + 1) << 30, dtype=torch.uint8, device=dev)); bufs[-1][:: 1 << 20].zero_()
        if i % 3 == 2:
            del bufs[0]; torch.cuda.empty_cache()
    tiny("after growing allocations (live)")
    del bufs; gc.collect(); torch.cuda.empty_cache()
    tiny("after freeing them")
    sys.exit(0)
proc = SI._procedural_estimator(scene, dev)
c2w = np.stack([RD.pose_to_c2w(p) for p in scene["poses"][:8]]).astype(np.float32)
K6 = np.array([[320.0, 0, 320], [0, 320.0, 320], [0, 0, 1.0]])
g = torch.Generator(device="cpu").manual_seed(100)
bs = []
for k in range(8):
    idx = torch.randint(0, 640 * 640, (8192,), generator=g).numpy()
    r = RD.generate_image_rays(torch.from_numpy(c2w[k:k + 1]), 640, 640, K6, dev, idx)
    bs.append((r,) + SI.analytic_targets(proc, scene["aabb"], r.origins, r.viewdirs))
tiny("scene built")
from apnrf_amd.ngp import NGPRadianceField
f = NGPRadianceField(aabb=torch.from_numpy(scene["aabb"]), neurons=scene["neurons"], layers=scene["layers"], num_semantic_classes=scene["C"],
                     log2_hashmap_size=scene["log2_hashmap_size"], seed=11).to(dev).train()
e = OccGridEstimator(torch.from_numpy(scene["aabb"]), resolution=scene["res"], levels=1).to(dev).train()
o = FusedAdam(f.parameters(), lr=2e-3, eps=1e-15).bind_field(f)
bk = torch.rand(3, device=dev)
n_steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
for i in range(n_steps):
    out = RD.train_step(f, e, o, *bs[i % 8], bk, step=i, sync=True, occ_thre=1e-2, deterministic=(mode == "det"), fused=(mode != "autograd"), **SC.RENDER_KW)
    if i in (0, 1, 2, 5, 10, 20, 50, 100, 200) or i == n_steps - 1:
        torch.cuda.synchronize()
        tiny(f"after step {i} (kept {out['n_rendering_samples']})")
tiny("on a new stream", torch.cuda.Stream())
del f, e, o; gc.collect(); RD.release_workspaces(); torch.cuda.empty_cache()
tiny("field destroyed, caches dropped")
