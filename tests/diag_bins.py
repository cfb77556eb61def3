"""Child process of test_binned_scatter_equals_walk: runs under MNF_LIB_PATH=libmi355nerf_diag.so (the only build that reads MNF_BIN_LEVEL0 /
MNF_BIN_CAP) and compares the hash-table gradient of one backward pass through (a) the walk alone (float atomics, every level), (b) the binned
path for the fine levels, (c) the binned path with lists of 64 items, so that almost every item takes the full-list route (float atomics from
pass A)."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
import helpers as H  # noqa: E402
from apnrf_amd import _lib as L  # noqa: E402

assert L.lib_path().endswith("_diag.so"), L.lib_path()
for lh, n in ((15, 20011), (19, 60000)):
    sc = H.make_scene(log2_hashmap_size=lh, head_gain=2.0)
    hip = H.hip_field(sc).train()
    rng = np.random.default_rng(5)
    a = sc["aabb"]
    pos = (rng.random((n, 3)) * (a[3:] - a[:3]) * 0.98 + a[:3] + 0.01 * (a[3:] - a[:3])).astype(np.float32)
    pos[: n // 2] = pos[0] + np.cumsum(np.full((n // 2, 3), 2e-3, np.float32), axis=0) % (0.5 * (a[3:] - a[:3]))   # a ray-like run: neighbours share cells
    d = rng.normal(size=(n, 3)).astype(np.float32); d /= np.linalg.norm(d, axis=-1, keepdims=True)
    cu = lambda x: torch.from_numpy(x).cuda()
    gs = [cu((rng.normal(size=(n, 3)) * 1e-3).astype(np.float32)), cu((rng.normal(size=(n, 1)) * 1e-5).astype(np.float32)),
          cu((rng.normal(size=(n, 29)) * 1e-3).astype(np.float32))]
    grads = {}
    for name, env in (("walk", {"MNF_BIN_LEVEL0": "16"}), ("bins", {"MNF_BIN_LEVEL0": "8"}), ("full", {"MNF_BIN_LEVEL0": "8", "MNF_BIN_CAP": "64"})):
        for k in ("MNF_BIN_LEVEL0", "MNF_BIN_CAP"):
            os.environ.pop(k, None)
        os.environ.update(env)
        hip.zero_grad()
        rgb, sigma, sem = hip(cu(pos), cu(d))
        torch.autograd.backward([rgb, sigma, sem], gs)
        torch.cuda.synchronize()
        grads[name] = hip.mlp_base.params.grad.double().clone()
    ref = grads["walk"]
    assert float(ref.norm()) > 0
    for name in ("bins", "full"):
        err = float((grads[name] - ref).norm() / ref.norm())
        worst = float((grads[name] - ref).abs().max() / ref.abs().max())
        assert err < 2e-6 and worst < 2e-5, (lh, name, err, worst)          # items keep 21 significant bits; the sums are formed in a different order
        assert bool(((ref == 0) == (grads[name] == 0)).all()), (lh, name)      # the same entries touched
    print("case", lh, n, "ok", flush=True)
print("DIAG_BINS_OK")
