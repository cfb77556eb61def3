"""Child process of test_view_queue_renderer_equals_per_round_renderer: runs under MNF_LIB_PATH=libmi355nerf_diag.so (the only build that reads
MNF_NO_VIEWQ) and renders the same batches of small views twice — through the view-queue renderer (csrc/viewq.hip: one persistent launch, the round
schedule decided on the device) and through the per-round launches of csrc/render.hip — and compares them ray by ray.  Same marcher, same field
arithmetic, same compositing code; tile composition differs (per wave of 64 rays instead of per 1024 rays), which changes nothing per ray."""
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
bk = torch.tensor([0.2, 0.4, 0.1])
for (hh, ww, views, prob) in [(32, 32, 3, False), (25, 40, 5, True), (64, 64, 6, True), (10, 7, 9, False)]:
    os, ds = [], []
    for p in range(views):
        o, d = H.view_rays(scene, p % 8, h=hh, w=ww)
        os.append(o); ds.append(d)
    o, d = torch.cat(os).to(DEV), torch.cat(ds).to(DEV)
    rpv = hh * ww
    import os as _os
    _os.environ.pop("MNF_NO_VIEWQ", None)
    q = RD.render_views(hip, est, o, d, rpv, 1024, render_bkgd=bk, probabilistic=prob, **H.RENDER_KW)
    q2 = RD.render_views(hip, est, o, d, rpv, 1024, render_bkgd=bk, probabilistic=prob, n_split=3, **H.RENDER_KW)
    _os.environ["MNF_NO_VIEWQ"] = "1"
    r = RD.render_views(hip, est, o, d, rpv, 1024, render_bkgd=bk, probabilistic=prob, **H.RENDER_KW)
    _os.environ.pop("MNF_NO_VIEWQ", None)
    tq, tr = q["total"].cpu().numpy(), r["total"].cpu().numpy()
    assert tr[1] > 20 * rpv * views, tr
    keys = ("rgb", "acc", "depth", "sem") + (("rgb_var", "depth_var") if prob else ())
    worst = 0.0
    for k in keys:
        assert torch.equal(q[k], q2[k]), ("jobs", k)                       # a view's result does not depend on how the batch is cut into jobs
        a, b = q[k].cpu().numpy().reshape(rpv * views, -1), r[k].cpu().numpy().reshape(rpv * views, -1)
        worst = max(worst, float(np.abs(a - b).max()))
        np.testing.assert_allclose(a, b, atol=2e-5, rtol=2e-5, err_msg=f"{k} {hh}x{ww}x{views}")
    assert (tq == tr).all() and torch.equal(q["total"], q2["total"]), (tq, tr)
    print(f"case {hh}x{ww} x {views} prob={prob}: samples {tq.tolist()} max |diff| {worst:.2e}", flush=True)
print("DIAG_VIEWQ_OK")
