import os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import apnrf_amd
from apnrf_amd import render as RD, scenes as SC, standin as SI
from oracle import render as R
from oracle.field import FieldConfig, OracleField
dev = "cuda:0"
scene = SC.make_scene("102344529", n_poses=40)
field, est, info = SI.train_standin(scene, dev, seed=9)
sc = dict(scene)
sc["params"] = {k: getattr(field, k).params.detach().cpu().numpy() for k in ("mlp_base", "mlp_head", "mlp_sem")}
sc["occ"] = est.binaries.cpu().numpy()
cfg = FieldConfig(aabb=tuple(float(x) for x in sc["aabb"]), neurons=128, layers=2, num_semantic_classes=29, log2_hashmap_size=19)
torch.set_num_threads(8)
W = 800; focal = 0.5 * W / np.tan(np.pi / 4)
idx = R.subsample_indices(W * W, 576)
o, d = R.generate_image_rays(R.pose_to_c2w(scene["poses"][0]), W, W, focal, idx)
bk = torch.zeros(3)
got = RD.render_views(field, est, o.to(dev), d.to(dev), 576, 1024, render_bkgd=bk, **SC.RENDER_KW)
for prec in ("f16", "f32"):
    orc = OracleField(cfg, sc["params"], prec, False)
    r = R.render_test(1024, orc, sc["occ"], sc["aabb"][None], o, d, render_bkgd=bk, **SC.RENDER_KW)
    e = (got["sem"].cpu() - r["sem"]).abs()
    mag = r["sem"].abs().max(dim=1).values
    per_ray = e.max(dim=1).values
    print(prec, "sem |max| over rays: median", float(mag.median()), "max", float(mag.max()), "| abs err: max", float(per_ray.max()), "rays > 1e-3:", int((per_ray > 1e-3).sum()),
          "| err / ray max|sem|: max", float((per_ray / mag.clamp_min(1)).max()), "rays > 1e-3 rel:", int((per_ray / mag.clamp_min(1) > 1e-3).sum()),
          "| rgb", float((got["rgb"].cpu() - r["rgb"]).abs().max()), "depth", float((got["depth"].cpu() - r["depth"]).abs().max()), "total", int(got["total"][0]), r["total_samples"])
# sample-level logits magnitude
pos = (o + d * 1.5)
with torch.no_grad():
    rgb, sig, sem = field(pos.to(dev), d.to(dev))
print("per-sample logits |max|", float(sem.abs().max()), "median of row max", float(sem.abs().max(dim=1).values.median()))
