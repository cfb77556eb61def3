"""Debug: multi-level render after the workspaces were used by other calls (stale contents): which side holds non-finite values?"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import helpers as H
from apnrf_amd import render as RD
from apnrf_amd.nerfacc import OccGridEstimator
from oracle import render as R
DEV = "cuda:0"
sc = H.make_scene(log2_hashmap_size=15)
# dirty the workspaces
f1, e1 = H.hip_field(sc), H.hip_estimator(sc)
o, d = H.view_rays(sc, 2, h=64, w=64)
for prob in (False, True):
    RD.render_views(f1, e1, o.to(DEV), d.to(DEV), o.shape[0], 1024, probabilistic=prob, **H.RENDER_KW)
for levels in (2, 3):
    roi = np.array([-16.0, 0.0, -16.0, -6.0, 2.4, -6.0], np.float32)
    est = OccGridEstimator(torch.from_numpy(roi), resolution=[50, 12, 50], levels=levels)
    rng = np.random.default_rng(7)
    occ = rng.random((levels, 50, 12, 50)) < np.array([0.12, 0.08, 0.05])[:levels, None, None, None]
    est.binaries = torch.from_numpy(occ); est = est.to(DEV).eval()
    aabbs = est.aabbs.cpu().numpy()
    fs = dict(sc); fs["aabb"] = aabbs[-1].astype(np.float32)
    hip, orc = H.hip_field(fs), H.oracle_field(fs)
    o, d = H.view_rays(sc, 3, h=24, w=24); o = o + torch.tensor([3.0, 0.0, 3.0])
    bk = torch.tensor([0.2, 0.1, 0.4])
    ref = R.render_test(1024, orc, occ, aabbs, o, d, render_bkgd=bk, **H.RENDER_KW)
    for rep in range(3):
        out = RD.render_views(hip, est, o.to(DEV), d.to(DEV), o.shape[0], 1024, render_bkgd=bk, **H.RENDER_KW)
        for k in ("rgb", "acc", "depth", "sem"):
            a, b = out[k].cpu().reshape(o.shape[0], -1), ref[k].reshape(o.shape[0], -1)
            bad_a, bad_b = (~torch.isfinite(a)).any(1).nonzero().flatten().tolist(), (~torch.isfinite(b)).any(1).nonzero().flatten().tolist()
            if bad_a or bad_b:
                print(levels, rep, k, "non-finite rays hip", bad_a, "oracle", bad_b)
                for r in (bad_a + bad_b)[:3]:
                    print("   ray", r, "o", o[r].tolist(), "d", d[r].tolist(), "hip", a[r, :3].tolist(), "oracle", b[r, :3].tolist(), "acc hip", float(out["acc"][r]), "oracle", float(ref["acc"][r]))
        print(levels, rep, "totals", out["total"].tolist(), ref["total_samples"])
