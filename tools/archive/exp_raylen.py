"""Per-ray sample counts of the config-5 train batch (trained stand-in): the density pre-pass walks a ray's tiles one after the other, so its longest rays bound it."""
import os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import apnrf_amd  # noqa: F401
from apnrf_amd import render as RD, scenes as SC, standin as SI
dev = "cuda:0"
scene = SC.make_scene("102344280", n_poses=40)
field, est, info = SI.train_standin(scene, dev, seed=11)
c2w = np.stack([RD.pose_to_c2w(p) for p in scene["poses"][:8]]).astype(np.float32)
K6 = np.array([[320.0, 0, 320], [0, 320.0, 320], [0, 0, 1.0]])
g = torch.Generator(device="cpu").manual_seed(100)
for R in (8192, 2000):
    idx = torch.randint(0, 640 * 640, (R,), generator=g).numpy()
    r = RD.generate_image_rays(torch.from_numpy(c2w[0:1]), 640, 640, K6, dev, idx)
    near = torch.full((R,), 0.1, device=dev); far = torch.full((R,), 1e10, device=dev)
    ri, ts, te, packed = est._sample_single_pass(r.origins, r.viewdirs, near, far, 1e-3, 0.004)
    c = packed[:, 1].cpu().numpy()
    tiles = (c + 63) // 64
    print(f"rays {R}: marched {c.sum()}, mean {c.mean():.0f}, median {np.median(c):.0f}, p90 {np.percentile(c, 90):.0f}, p99 {np.percentile(c, 99):.0f}, max {c.max()} | tiles: total {tiles.sum()}, "
          f"longest ray {tiles.max()}, lanes used {c.sum() / (64 * tiles.sum()):.3f} | rays >= 512 samples: {(c >= 512).sum()}, >= 256: {(c >= 256).sum()}")
