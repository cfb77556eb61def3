import os, sys
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import helpers as H
from apnrf_amd import render as RD
DEV = "cuda:0"
scene = H.make_scene()
sc2 = dict(scene); sc2["params"] = H.S.make_field_params(seed=1)
fields = [H.hip_field(scene), H.hip_field(sc2)]
ests = [H.hip_estimator(scene), H.hip_estimator(scene)]
poses = scene["poses"][[1, 4, 6]]
a = [RD.score_views(fields, ests, poses, 640, 640, 320.0, 0.1, 1e-3, 0.025, 0.004, 0.01, DEV)[0] for _ in range(3)]
b = [RD.score_poses(fields, ests, poses, 640, 640, 320.0, 0.1, 1e-3, 0.025, 0.004, 0.01, DEV)[0] for _ in range(3)]
print("py==py", [torch.equal(a[0], x) for x in a], "c==c", [torch.equal(b[0], x) for x in b], "py==c", torch.equal(a[0], b[0]))
print((a[0] - b[0]).abs().max().item(), (a[0]-a[1]).abs().max().item(), (b[0]-b[1]).abs().max().item())
# per-output comparison of the two routes' renders
o, d, h, w = RD._pose_rays(poses, 640, 640, 320.0, 0.025, DEV)
for ns in (1, 2, 3):
    r = RD._render_jobs([(f, e, o, d) for f, e in zip(fields, ests)], h * w, 1024, 0.1, 1e10, 1e-3, torch.zeros(3), 0.004, 0.01, 1e-4, True, 8, (h, w), ns)
    if ns == 1:
        ref = r
    else:
        for m in range(2):
            print("n_split", ns, "member", m, {k: bool(torch.equal(r[m][k], ref[m][k])) for k in ("rgb", "acc", "depth", "sem", "rgb_var", "depth_var", "total")})
