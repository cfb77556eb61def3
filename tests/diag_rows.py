"""Child process of test_feature_rows_are_bitwise_a_second_gather: runs under MNF_LIB_PATH=libmi355nerf_diag.so (the only build that reads MNF_NO_ROWS) and
runs the same deterministic train steps twice — with the forward reading the rows the density pre-pass left (the product's path) and with both passes
gathering from the hash table (MNF_NO_ROWS=1) — and requires identical sample counts, parameters and gradients, fp16 and bf16."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
import helpers as H  # noqa: E402
from apnrf_amd import _lib as L  # noqa: E402
from apnrf_amd import render as RD  # noqa: E402
from apnrf_amd import scenes as SC  # noqa: E402
from apnrf_amd.optim import FusedAdam  # noqa: E402

assert L.lib_path().endswith("_diag.so"), L.lib_path()
DEV = "cuda:0"
sc = H.make_scene(log2_hashmap_size=15)
bk = torch.tensor([0.3, 0.6, 0.1], device=DEV)
data = []
for k in range(4):
    o, d = H.view_rays(sc, 1 + k % 3, h=40, w=40)
    rng = np.random.default_rng(90 + k)
    n = o.shape[0]
    data.append((RD.Rays(o.to(DEV), d.to(DEV)), torch.from_numpy(rng.random((n, 3)).astype(np.float32)).to(DEV),
                 torch.from_numpy(rng.uniform(0.5, 4.0, n).astype(np.float32)).to(DEV), torch.from_numpy(rng.integers(0, sc["C"], n)).to(DEV), bk))


def run(bf16):
    f = SC.hip_field(sc, DEV, mfma_bf16=bf16).train()
    e = H.hip_estimator(sc)
    opt = FusedAdam(f.parameters(), lr=1e-3, eps=1e-15).bind_field(f)
    ns = []
    for k, b in enumerate(data):
        out = RD.train_step(f, e, opt, *b, step=1 + k, seed=40 + k, deterministic=True, **H.RENDER_KW)
        ns.append(out["n_rendering_samples"])
    torch.cuda.synchronize()
    return ns, [p.detach().clone() for p in f.parameters() if p.numel()], [p.grad.detach().clone() for p in f.parameters() if p.numel()]


for bf16 in (False, True):
    os.environ.pop("MNF_NO_ROWS", None)
    n_a, p_a, g_a = run(bf16)
    os.environ["MNF_NO_ROWS"] = "1"
    n_b, p_b, g_b = run(bf16)
    assert n_a == n_b and min(n_a) > 3000, (n_a, n_b)
    assert all(torch.equal(x, y) for x, y in zip(p_a, p_b)) and all(torch.equal(x, y) for x, y in zip(g_a, g_b)), bf16
    print("bf16" if bf16 else "f16", n_a, "ok", flush=True)
print("DIAG_ROWS_OK")
