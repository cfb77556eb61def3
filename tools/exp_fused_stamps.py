"""Phase profile of the fused backward (GPU box): where a tile's time goes, wave by wave.  Needs an experiment build with -DMNF_FUSED_STAMPS
(tools/build_variants.sh "stamps -DMNF_FUSED_STAMPS"; run with MNF_LIB_PATH=gpurun_exp/lib_stamps.so): workgroup 0 stamps the shader clock at the start of
every tile and on both sides of each of the nine barriers.
    MNF_LIB_PATH=$PWD/gpurun_exp/lib_stamps.so python tools/exp_fused_stamps.py [n_samples]"""
import ctypes
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import apnrf_amd  # noqa: F401
from apnrf_amd import _lib as L
from apnrf_amd.ngp import NGPRadianceField

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
dev = "cuda:0"
torch.manual_seed(3)
C = 29
f = NGPRadianceField(aabb=[-1, -1, -1, 1, 1, 1], neurons=128, layers=2, num_semantic_classes=C, seed=4).to(dev)
f.train()
lib = L.load_library()
L.check(lib.mnf_field_set_backward_mode(f._ensure_handle(), 2))
pos = (torch.rand(n, 3, device=dev) * 2 - 1) * 0.98
dirs = torch.nn.functional.normalize(torch.randn(n, 3, device=dev), dim=-1)
g_rgb = torch.randn(n, 3, device=dev) * 1e-3
g_sig = torch.randn(n, 1, device=dev) * 1e-4
g_sem = torch.randn(n, C, device=dev) * 1e-3
for _ in range(3):
    for p in f.parameters():
        p.grad = None
    rgb, sigma, sem = f(pos, dirs)
    (rgb * g_rgb).sum().add((sigma * g_sig).sum()).add((sem * g_sem).sum()).backward()
    torch.cuda.synchronize()
T, S = 24, 24
buf = (ctypes.c_ulonglong * (T * 4 * S))()
fn = lib.mnf_exp_fused_stamps
fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert fn(buf, T * 4 * S) == 0
st = np.frombuffer(buf, dtype=np.uint64).reshape(T, 4, S).astype(np.int64)
names = ["L0 fwd", "hidden fwd", "base-out + head L1", "head hid fwd + dY + head out bwd", "dZ1 head / geo", "phase 6", "phase 7", "hidden bwd", "phase 9 (to tile end)"]
# per tile (skip the first two): arrival = stamp[2k-1] - stamp[2k-2], wait at the barrier = stamp[2k] - stamp[2k-1]
tiles = range(2, T - 1)
print("cycles per phase, mean over tiles %d..%d of workgroup 0 (shader clock 100 MHz units if s_memtime is the reference clock)" % (tiles[0], tiles[-1]))
tot = np.zeros(4)
for k in range(1, 10):
    work = np.mean([st[t, :, 2 * k - 1] - st[t, :, 2 * k - 2] for t in tiles], axis=0)
    wait = np.mean([st[t, :, 2 * k] - st[t, :, 2 * k - 1] for t in tiles], axis=0)
    tot += work + wait
    print("phase %d  work per wave %s   wait at barrier %s" % (k, " ".join("%7.0f" % w for w in work), " ".join("%7.0f" % w for w in wait)))
loop = np.mean([st[t + 1, :, 0] - st[t, :, 0] for t in tiles], axis=0)
print("tile to tile %s   sum of phases %s" % (" ".join("%7.0f" % w for w in loop), " ".join("%7.0f" % w for w in tot)))
