"""Scoring-pass timing (GPU box): BASELINE config 4 (256 candidate views x 4096 rays x 2 members) and its shard-of-8 share
(32 views), through `render.score_views` and through the single C call `render.score_poses`.
    python tools/exp_score.py [n_views,...] [passes] [groups of views per member (A/B of render._render_jobs' n_split; score_views only)]"""
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import apnrf_amd  # noqa: F401
from apnrf_amd import render as RD
from apnrf_amd import scenes as SC
from apnrf_amd import standin as SI

views = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [256, 32]
passes = int(sys.argv[2]) if len(sys.argv) > 2 else 5
if len(sys.argv) > 3:
    _jobs, _n = RD._render_jobs, int(sys.argv[3])
    RD._render_jobs = lambda *a, **k: _jobs(*a[:13], _n) if len(a) > 13 else _jobs(*a, **dict(k, n_split=_n))
dev = "cuda:0"
scene = SC.make_scene("102344250", n_poses=40)
f0, e0, _ = SI.train_standin(scene, dev, seed=9)
f1, e1, _ = SI.train_standin(scene, dev, seed=10)
poses = SI._free_space_poses(scene, 256, seed=9)
for V in views:
    for name, fn in (("score_views", lambda p: RD.score_views([f0, f1], [e0, e1], p, 640, 640, 320.0, 0.1, 1e-3, 0.1, 0.004, 0.01, dev, group=False)),
                     ("score_poses", lambda p: RD.score_poses([f0, f1], [e0, e1], p, 640, 640, 320.0, 0.1, 1e-3, 0.1, 0.004, 0.01, dev))):
        terms, score = fn(poses[:V])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(passes):
            terms, score = fn(poses[:V])
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / passes
        ev = sum(int(t[1]) for t in RD.LAST_SCORE_TOTALS) if name == "score_views" else 0
        print(f"[exp_score] {name} V={V}: {1e3 * dt:.2f} ms/pass, score {float(score):.9f}, evaluated samples {ev} ({ev / dt / 1e9 if ev else 0:.2f} G/s)", flush=True)
