"""The optimizer call of a train step alone (GPU box): `FusedAdam.step(skip=, count_nonfinite=True)` on a bound field with random gradients, per call by
torch events; also a plain device copy of the same bytes as a yardstick.    MNF_LIB_PATH=<variant> python tools/exp_adam.py"""
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import apnrf_amd  # noqa: F401
from apnrf_amd import scenes as SC
from apnrf_amd.optim import FusedAdam

dev = "cuda:0"
f = SC.hip_field(SC.make_scene("102344280"), dev).train()
opt = FusedAdam(f.parameters(), lr=1e-4, eps=1e-15).bind_field(f)
for p in f.parameters():
    p.grad = torch.randn_like(p) * 1e-3
skip = torch.zeros((), dtype=torch.int32, device=dev)
n = sum(p.numel() for p in f.parameters())


def timed(fn, reps=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


t = timed(lambda: opt.step(skip=skip, count_nonfinite=True))
t2 = timed(lambda: opt.step(skip=skip, count_nonfinite=False))
x = torch.empty(n * 7 // 2, device=dev); y = torch.empty_like(x)      # reads 4 n floats, writes 3.5 n floats -> a copy of 3.5 n floats moves 7 n floats' bytes, nearly the same
tc = timed(lambda: y.copy_(x))
print(f"[exp_adam] {os.environ.get('MNF_LIB_PATH', 'product')}: {n} parameters: guard + update {t:.1f} us, update alone {t2:.1f} us "
      f"({n * 30 / t2 / 1e6:.2f} TB/s of its 30 B per parameter); device copy of {x.numel() * 4 / 1e6:.0f} MB {tc:.1f} us ({x.numel() * 8 / tc / 1e6:.2f} TB/s)", flush=True)
