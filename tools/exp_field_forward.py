"""Field-kernel timing on FIXED sample positions (mode 0: explicit positions + directions, no compositing), so that variants whose outputs differ (knock-outs) still do
the same work: python tools/exp_field_forward.py [n_samples]   (run with MNF_LIB_PATH=<variant>)"""
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import apnrf_amd  # noqa: F401
from apnrf_amd import scenes as SC

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 22
dev = "cuda:0"
scene = SC.make_scene("102344529")
f = SC.hip_field(scene, dev)
rng = np.random.default_rng(0)
a = scene["aabb"]
# coherent-ish samples: 64 consecutive samples along short ray segments (what a render tile looks like)
o = rng.uniform(a[:3], a[3:], size=(n // 64, 3)).astype(np.float32)
d = rng.normal(size=(n // 64, 3)).astype(np.float32); d /= np.linalg.norm(d, axis=1, keepdims=True)
t = (np.arange(64, dtype=np.float32) * 0.004)[None, :, None]
pos = torch.from_numpy((o[:, None, :] + d[:, None, :] * t).reshape(-1, 3)).to(dev)
dirs = torch.from_numpy(np.repeat(d, 64, axis=0)).to(dev)
with torch.no_grad():
    for _ in range(3):
        f(pos, dirs)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20):
        f(pos, dirs)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
print(f"[exp_field_forward] {os.environ.get('MNF_LIB_PATH', 'product')}: {n} samples, {1e3 * dt:.4f} ms per launch, {n / dt / 1e9:.3f} G samples/s", flush=True)
