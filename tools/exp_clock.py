"""Effective shader clock while the field kernel runs (GPU box): tools/clock_probe.hip on a side stream beside `field(pos, dirs)` on 4 M fixed positions.
    hipcc --offload-arch=gfx950 -O2 -shared -fPIC -o tools/_build/libclock_probe.so tools/clock_probe.hip;  python tools/exp_clock.py"""
import ctypes
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import apnrf_amd  # noqa: F401
from apnrf_amd import scenes as SC

probe = ctypes.CDLL(os.path.join(REPO, "tools", "_build", "libclock_probe.so"))
probe.clock_probe_launch.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_longlong, ctypes.c_void_p, ctypes.c_void_p]
dev = "cuda:0"
n = 1 << 22
scene = SC.make_scene("102344529")
f = SC.hip_field(scene, dev)
rng = np.random.default_rng(0)
a = scene["aabb"]
o = rng.uniform(a[:3], a[3:], size=(n // 64, 3)).astype(np.float32)
d = rng.normal(size=(n // 64, 3)).astype(np.float32); d /= np.linalg.norm(d, axis=1, keepdims=True)
t = (np.arange(64, dtype=np.float32) * 0.004)[None, :, None]
pos = torch.from_numpy((o[:, None, :] + d[:, None, :] * t).reshape(-1, 3)).to(dev)
dirs = torch.from_numpy(np.repeat(d, 64, axis=0)).to(dev)
side = torch.cuda.Stream()
BLOCKS, N, GAP = 8, 400, 1000      # 400 samples 10 us apart = 4 ms
wall_hz = 1e8


def run(load, label, gap=GAP):
    out = torch.zeros((BLOCKS, N, 2), dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    rc = probe.clock_probe_launch(BLOCKS, N, gap, out.data_ptr(), side.cuda_stream)
    assert rc == 0
    with torch.no_grad():
        for _ in range(load):
            f(pos, dirs)
    torch.cuda.synchronize()
    v = out.cpu().numpy().astype(np.float64)
    ds, dw = np.diff(v[:, :, 0], axis=1), np.diff(v[:, :, 1], axis=1)
    mhz = ds / dw * wall_hz / 1e6
    q = np.percentile(mhz, [5, 50, 95])
    print(f"[exp_clock] {label}: shader ticks per wall tick x 100 MHz: median {q[1]:.0f} MHz (5 % {q[0]:.0f}, 95 % {q[2]:.0f}); by 0.5 ms windows: "
          + " ".join(f"{m:.0f}" for m in mhz.mean(axis=0)[:350].reshape(-1, 50).mean(axis=1)), flush=True)


with torch.no_grad():
    for _ in range(3):
        f(pos, dirs)
run(0, "idle GPU")
run(4, "beside 4 field launches (4.2 M samples each)")
run(4, "again")
run(330, "beside 330 field launches back to back (windows of 50 ms)", gap=100000)
x = torch.randn(1 << 28, device=dev)
torch.cuda.synchronize()
out = torch.zeros((BLOCKS, N, 2), dtype=torch.int64, device=dev)
probe.clock_probe_launch(BLOCKS, N, GAP, out.data_ptr(), side.cuda_stream)
for _ in range(6):
    y = x * 2.0
torch.cuda.synchronize()
v = out.cpu().numpy().astype(np.float64)
mhz = np.diff(v[:, :, 0], axis=1) / np.diff(v[:, :, 1], axis=1) * 100
print(f"[exp_clock] beside a streaming torch multiply: median {np.median(mhz):.0f} MHz", flush=True)
