"""Per-round log of ONE scoring pass (BASELINE config 4: 256 candidate views x 4096 rays, one ensemble member at a time):
columns, field-kernel and marcher time, active views and surviving rays of every render round (MNF_ROUND_LOG=1, which
synchronises every round: the times are per-launch times, not pass times).  Usage (GPU box):
    MNF_LIB_PATH=<repo>/active-perception-using-neural-radiance-fields_amd/libmi355nerf_diag.so MNF_ROUND_LOG=1 python tools/score_roundlog.py [n_views] 2> gpurun_out/score_roundlog.txt"""
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import apnrf_amd  # noqa: F401
from apnrf_amd import render as RD
from apnrf_amd import standin as SI
from apnrf_amd import scenes as SC

n_views = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = "cuda:0"
scene = SC.make_scene("102344250", n_poses=40)
f0, e0, _ = SI.train_standin(scene, dev, seed=9)
poses = SI._free_space_poses(scene, 256, seed=9)[:n_views]
torch.cuda.synchronize()
t0 = time.perf_counter()
terms, score = RD.score_views([f0], [e0], poses, 640, 640, 320.0, 0.1, 1e-3, 0.1, 0.004, 0.01, dev, group=False)
torch.cuda.synchronize()
print(f"[score_roundlog] one member, {n_views} views: {1e3 * (time.perf_counter() - t0):.1f} ms (with per-round syncs)", file=sys.stderr)
