"""Time of the train step's single-pass sampler (`mnf_sample_rays_levels`, csrc/march.hip sample_rays_kernel) alone on the config-5 / reference-yaml batches
of the trained stand-in: us per call over 30 calls.  Experiment builds (MNF_LIB_PATH): threads per workgroup, stores knocked out.
    python tools/exp_sampler.py"""
import ctypes
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import apnrf_amd  # noqa: F401
from apnrf_amd import _lib as L
from apnrf_amd import render as RD, scenes as SC, standin as SI

dev = "cuda:0"
scene = SC.make_scene("102344280", n_poses=40)
if len(sys.argv) > 1 and sys.argv[1] == "proc":      # procedural occupancy grid, no training (experiment builds whose sampler is not usable for training)
    est = SC.hip_estimator(scene, dev)
else:
    field, est, info = SI.train_standin(scene, dev, seed=11)
lib = L.load_library()
c2w = np.stack([RD.pose_to_c2w(p) for p in scene["poses"][:8]]).astype(np.float32)
K6 = np.array([[320.0, 0, 320], [0, 320.0, 320], [0, 0, 1.0]])
g = torch.Generator(device="cpu").manual_seed(100)
b = est.binaries.contiguous().view(torch.uint8)
res = [int(x) for x in b.shape[1:]]
aabb_host = (ctypes.c_float * 6)(*est.aabb_host(0))
for R in (8192, 2000):
    idx = torch.randint(0, 640 * 640, (R,), generator=g).numpy()
    ys, xs = idx // 640, idx % 640
    idx = idx[np.argsort((ys // 32) * 20 + xs // 32, kind="stable")]
    r = RD.generate_image_rays(torch.from_numpy(c2w[0:1]), 640, 640, K6, dev, idx)
    near = torch.full((R,), 0.1, device=dev); far = torch.full((R,), 1e10, device=dev)
    cap = int(max(64, min(2048, (1 << 27) // R)))
    scratch = torch.empty((2, R, cap), device=dev); counts = torch.empty((R,), device=dev, dtype=torch.int64)
    o, d = r.origins.contiguous(), r.viewdirs.contiguous()
    if "same" in sys.argv:        # every lane marches the ray with the most samples: no divergence, the kernel's time is ONE ray's dependent chain
        call_counts = torch.empty((R,), device=dev, dtype=torch.int64)
        L.launch(lib.mnf_sample_rays_levels, L.ptr(o), L.ptr(d), R, L.ptr(b), 1, res[0], res[1], res[2], aabb_host, L.ptr(near), L.ptr(far), 1e-3, 0.004, cap,
                 L.ptr(scratch[0]), L.ptr(scratch[1]), L.ptr(call_counts), L.ptr(est.bitgrid()[0]))
        k = int(call_counts.argmax())
        o, d = o[k:k + 1].repeat(R, 1).contiguous(), d[k:k + 1].repeat(R, 1).contiguous()

    def call():
        L.launch(lib.mnf_sample_rays_levels, L.ptr(o), L.ptr(d), R, L.ptr(b), 1, res[0], res[1], res[2], aabb_host, L.ptr(near), L.ptr(far), 1e-3, 0.004, cap,
                 L.ptr(scratch[0]), L.ptr(scratch[1]), L.ptr(counts), L.ptr(est.bitgrid()[0]))
    for _ in range(3):
        call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30):
        call()
    e1.record(); torch.cuda.synchronize()
    c = counts.cpu().numpy()
    print(f"[exp_sampler] {os.environ.get('MNF_LIB_PATH', 'product')}: rays {R}: {e0.elapsed_time(e1) / 30 * 1e3:.1f} us per call | marched {c.sum()}, mean {c.mean():.0f}, max {c.max()}, checksum {int(scratch[0, :, 0].double().sum().item() * 1e6)}", flush=True)
