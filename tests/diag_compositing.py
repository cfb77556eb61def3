"""Child process of test_slot_compositing_equals_general_path: runs under MNF_LIB_PATH=libmi355nerf_diag.so (the -DMNF_DIAG
build, the only one that reads MNF_MIN_SAMPLES / MNF_COMPOSITE_GENERAL) and compares the slot compositing path of the render
epilogue with the general segmented-scan path on the same rays and schedule."""
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

assert L.lib_path().endswith("_diag.so"), L.lib_path()
DEV = "cuda:0"
scene = H.make_scene()
hip, est = H.hip_field(scene), H.hip_estimator(scene)
o, d = H.view_rays(scene, 2, h=48, w=48)
o, d = torch.cat([o, o[:150]]).to(DEV), torch.cat([d, d[:150]]).to(DEV)      # ragged: the last march workgroup is partly idle
n = o.shape[0]
bk = torch.zeros(3)
for min_samples, prob in [(4, False), (4, True), (8, True), (16, False)]:
    os.environ["MNF_MIN_SAMPLES"] = str(min_samples)
    os.environ.pop("MNF_COMPOSITE_GENERAL", None)
    fast = RD.render_views(hip, est, o, d, n, 1024, render_bkgd=bk, probabilistic=prob, **H.RENDER_KW)
    os.environ["MNF_COMPOSITE_GENERAL"] = "1"
    gen = RD.render_views(hip, est, o, d, n, 1024, render_bkgd=bk, probabilistic=prob, **H.RENDER_KW)
    tf, tg = fast["total"].cpu().numpy(), gen["total"].cpu().numpy()
    assert tf[1] > 20 * n                                                         # the schedule really ran
    assert abs(int(tf[0]) - int(tg[0])) <= 1e-3 * tg[0] and abs(int(tf[1]) - int(tg[1])) <= 1e-3 * tg[1]   # threshold ties only
    keys = ("rgb", "acc", "depth", "sem") + (("rgb_var", "depth_var") if prob else ())
    for k in keys:
        a, b = fast[k].cpu().numpy().reshape(n, -1), gen[k].cpu().numpy().reshape(n, -1)
        close = np.all(np.abs(a - b) <= 2e-5 + 2e-5 * np.abs(b), axis=1)
        assert close.mean() > 0.999, (min_samples, prob, k, float(close.mean()), float(np.abs(a - b).max()))   # a ray retired one round apart moves visibly
    print("case", min_samples, prob, "ok", flush=True)
print("DIAG_COMPOSITING_OK")
