import os, sys, copy
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import apnrf_amd
from apnrf_amd import render as RD, scenes as SC, standin as SI
from apnrf_amd.optim import FusedAdam
dev = "cuda:0"
scene = SC.make_scene("102344280", n_poses=40)
field, est, info = SI.train_standin(scene, dev, seed=11, keep_optimizer=True)
print({k: v for k, v in info.items() if k != "optimizer_state"})
proc = SI._procedural_estimator(scene, dev)
c2w = np.stack([RD.pose_to_c2w(p) for p in scene["poses"][:8]]).astype(np.float32)
K6 = np.array([[320.0, 0, 320], [0, 320.0, 320], [0, 0, 1.0]])
g = torch.Generator(device="cpu").manual_seed(100)
bs = []
for k in range(8):
    idx = torch.randint(0, 640 * 640, (8192,), generator=g).numpy()
    r = RD.generate_image_rays(torch.from_numpy(c2w[k:k + 1]), 640, 640, K6, dev, idx)
    bs.append((r,) + SI.analytic_targets(proc, scene["aabb"], r.origins, r.viewdirs))
for mode in sys.argv[1:] or ["sync"]:
    f2 = SC.hip_field(scene, dev); f2.load_state_dict(field.state_dict())
    from apnrf_amd.nerfacc import OccGridEstimator
    e2 = OccGridEstimator(torch.from_numpy(scene["aabb"]), resolution=scene["res"], levels=1).to(dev)
    e2.occs.copy_(est.occs); e2.binaries = est.binaries.clone()
    opt = FusedAdam(f2.parameters(), lr=2e-4, eps=1e-15).bind_field(f2)
    opt.load_state_dict(copy.deepcopy(info["optimizer_state"]))
    for g_ in opt.param_groups: g_["lr"] = 2e-4
    gen = torch.Generator().manual_seed(7)
    for i in range(20):
        r, pix, dep_, lab = bs[i % 8]
        out = RD.train_step(f2, e2, opt, r, pix, dep_, lab, torch.rand(3, generator=gen).to(dev), step=1000 + i, sync=(mode == "sync"), occ_thre=1e-2, **SC.RENDER_KW)
        print(mode, i, "kept", int(out["n_rendering_samples"]), "marched", int(e2.last_sampling["n_marched"]), "skipped", int(out["skipped"]), "loss", float(out["loss"]) if out["loss"] is not None else None,
              "occupied", int(e2.binaries.sum()), flush=True)
