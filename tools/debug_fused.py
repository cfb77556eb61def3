"""A/B of the two backward paths of the field (GPU box): the split kernels (dgrad + wgrad over the activation dump, mode 1) against the fused
backward (csrc/fused_bwd.h, mode 2) on the same samples and output gradients — parameter gradients per vector and per matrix, and dX through
the hash-table gradient.
    python tools/debug_fused.py [n_samples] [layers] [bf16]"""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import apnrf_amd  # noqa: F401
from apnrf_amd import _lib as L
from apnrf_amd.ngp import NGPRadianceField

n = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
layers = int(sys.argv[2]) if len(sys.argv) > 2 else 2
bf16 = len(sys.argv) > 3 and sys.argv[3] == "bf16"
dev = "cuda:0"
torch.manual_seed(3)
C = 29
f = NGPRadianceField(aabb=[-1, -1, -1, 1, 1, 1], neurons=128, layers=layers, num_semantic_classes=C, log2_hashmap_size=15, seed=4, mfma_bf16=bf16).to(dev)
f.train()
with torch.no_grad():
    n_mlp0 = 128 * 64 + (layers - 1) * 128 * 128 + 16 * 128
    f.mlp_base.params[n_mlp0:] *= 3000.0          # hash features of order 0.3 instead of 1e-4: every layer gets real signal
lib = L.load_library()
pos = (torch.rand(n, 3, device=dev) * 2 - 1) * 0.98
dirs = torch.nn.functional.normalize(torch.randn(n, 3, device=dev), dim=-1)
g_rgb = torch.randn(n, 3, device=dev) * 1e-3
g_sig = torch.randn(n, 1, device=dev) * 1e-4
g_sem = torch.randn(n, C, device=dev) * 1e-3

W, Wh = 128, 64
sem_pad = 32
mats = {"base": [("in", W * 64), *[(f"hid{l}", W * W) for l in range(layers - 1)], ("out", 16 * W)],
        "head": [("in", Wh * 32), ("hid", Wh * Wh), ("out", 16 * Wh)],
        "sem": [("in", Wh * 16), ("hid", Wh * Wh), ("out", sem_pad * Wh)]}


def run(mode):
    L.check(lib.mnf_field_set_backward_mode(f._ensure_handle(), mode))
    for p in f.parameters():
        p.grad = None
    rgb, sigma, sem = f(pos, dirs)
    (rgb * g_rgb).sum().add((sigma * g_sig).sum()).add((sem * g_sem).sum()).backward()
    torch.cuda.synchronize()
    return rgb.detach(), sigma.detach(), sem.detach(), [f.mlp_base.params.grad.clone(), f.mlp_head.params.grad.clone(), f.mlp_sem.params.grad.clone()]


def rel(a, b):
    return float((a - b).norm() / (b.norm() + 1e-30))


r1 = run(1)
r2 = run(2)
r1b = run(1)
print("forward outputs equal:", [bool(torch.equal(a, b)) for a, b in zip(r1[:3], r2[:3])])
n_mlp = sum(k for _, k in mats["base"])
for name, i in (("base", 0), ("head", 1), ("sem", 2)):
    off = 0
    for mname, cnt in mats[name]:
        a, b, c = r2[3][i][off:off + cnt], r1[3][i][off:off + cnt], r1b[3][i][off:off + cnt]
        print(f"{name}.{mname}: rel L2 fused-vs-split {rel(a, b):.3e}   (split run-to-run {rel(c, b):.3e})   |split| {float(b.norm()):.3e}  max abs diff {float((a - b).abs().max()):.3e}")
        off += cnt
ta, tb, tc = r2[3][0][n_mlp:], r1[3][0][n_mlp:], r1b[3][0][n_mlp:]
print(f"hash table grad: rel L2 fused-vs-split {rel(ta, tb):.3e}   (split run-to-run {rel(tc, tb):.3e})   nonzero {int((tb != 0).sum())} vs {int((ta != 0).sum())}")
ok = all(rel(r2[3][i], r1[3][i]) < 2e-3 for i in range(3))
print("OK" if ok else "MISMATCH")
