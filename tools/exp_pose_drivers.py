"""(Round-6 experiment, needs the host-widening variant of render._host_stacks_f64 — see its docstring; kept for the record.)  The per-pose drivers with the float64 widening on the host or on the device: Dataset.render_image_from_pose (one 640 x 640 pose) and
Dataset.render_probablistic_image_from_pose (40 poses x 4096 rays), ms per call.   python tools/exp_pose_drivers.py"""
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import apnrf_amd  # noqa: F401
from apnrf_amd import render as RD
from apnrf_amd import scenes as SC
from apnrf_amd import standin as SI
from apnrf_amd.dataset import Dataset

dev = "cuda:0"
scene = SC.make_scene("102344250", n_poses=40)
f0, e0, _ = SI.train_standin(scene, dev, seed=9)
poses = SI._free_space_poses(scene, 256, seed=9)
a_full = (f0, e0, poses[:1], 640, 640, 320.0, 0.1, 1e-3, 1, 0.004, 0.01, 1, dev)
a_40 = (f0, e0, poses[:40], 640, 640, 320.0, 0.1, 1e-3, 0.1, 0.004, 0.01, 4, dev)
print("torch threads", torch.get_num_threads(), flush=True)
for label, thr in (("host widen", 16), ("device widen", 1 << 30), ("host widen", 16), ("device widen", 1 << 30)):
    RD.HOST_WIDEN_MIN_THREADS = thr
    for name, fn in (("render_image_from_pose 640x640", lambda: Dataset.render_image_from_pose(*a_full)),
                     ("render_probablistic_image_from_pose 40 poses", lambda: Dataset.render_probablistic_image_from_pose(*a_40))):
        fn(); torch.cuda.synchronize()
        ts = []
        for _ in range(8):
            t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
        print(f"[exp_pose_drivers] {label:13s} {name:48s}: median {1e3 * np.median(ts):6.2f} ms  min {1e3 * min(ts):6.2f}", flush=True)
