"""Dataset.render_image_from_pose returns float64 HOST arrays (habitat_to_data.py:376-409).  Which hand-over is fastest for one 640 x 640 pose (35 floats per ray)?
  a  widen to float64 on the device, ONE pinned transfer of 111 MB (round 5, the product)
  b  ONE pinned transfer of the float32 outputs (55 MB), widened on the host by torch (all threads) into a reused pageable buffer
  c  the same transfer, widened by numpy astype (one thread, fresh arrays)
(VERDICT r05 next 6 asked for b/c.)   python tools/exp_host_stacks.py"""
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import apnrf_amd  # noqa: F401
from apnrf_amd import render as RD

dev = "cuda:0"
P, h, w, C = 1, 640, 640, 29
g = torch.Generator().manual_seed(0)
outs = [(torch.rand(P * h * w, 3, generator=g).to(dev), (P, h, w, 3)), (torch.rand(P * h * w, 1, generator=g).to(dev), (P, h, w)),
        (torch.rand(P * h * w, 1, generator=g).to(dev), (P, h, w)), (torch.randn(P * h * w, C, generator=g).to(dev), (P, h, w, C))]
total = sum(int(np.prod(s)) for _, s in outs)
buf64 = {}


def a():
    return RD._host_stacks_f64(outs)


def _to_pinned32():
    host = torch.empty(total, dtype=torch.float32, pin_memory=True)
    off = 0
    for t, shape in outs:
        n = int(np.prod(shape))
        host[off:off + n].copy_(t.reshape(-1), non_blocking=True)
        off += n
    torch.cuda.current_stream(dev).synchronize()
    return host


def b():
    host = _to_pinned32()
    if "b" not in buf64:
        buf64["b"] = torch.empty(total, dtype=torch.float64)
    buf64["b"].copy_(host)
    arrays, off = [], 0
    for _, shape in outs:
        n = int(np.prod(shape))
        arrays.append(buf64["b"][off:off + n].numpy().reshape(shape))
        off += n
    return tuple(arrays)


def c():
    host = _to_pinned32().numpy()
    arrays, off = [], 0
    for _, shape in outs:
        n = int(np.prod(shape))
        arrays.append(host[off:off + n].astype(np.float64).reshape(shape))
        off += n
    return tuple(arrays)


ref = a()
for name, fn in (("a device-widened f64, one pinned transfer", a), ("b f32 pinned transfer + torch host widen (reused buffer)", b), ("c f32 pinned transfer + numpy astype", c)):
    got = fn()
    same = all(np.array_equal(x, y) for x, y in zip(got, ref))
    torch.cuda.synchronize()
    ts = []
    for _ in range(10):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    print(f"[exp_host_stacks] {name:58s}: {1e3 * np.median(ts):6.2f} ms (min {1e3 * min(ts):.2f}), bit-identical to a: {same}; torch threads {torch.get_num_threads()}", flush=True)
