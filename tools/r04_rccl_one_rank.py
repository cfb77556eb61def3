"""The RCCL path executed on hardware (VERDICT r03 missing 3): one rank under torch.distributed.run on the 1-GPU box — process-group
initialisation with backend "nccl" (= RCCL on ROCm), the collectives the package issues (all_gather_into_tensor of per-view score terms,
broadcast of a model's parameter vectors and occupancy grid, all-reduce of the parameter gradients + the skip flag) called on their real
tensors, and the sharded scoring / data-parallel train-step entry points with the group passed in.  With world_size 1 the package's own
wrappers return early (nothing to exchange), so the collectives are ALSO issued directly here; no scaling number comes out of this
(one GPU): it shows that the backend initialises and that every collective of the multi-GPU paths runs on this software stack.
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 tools/r04_rccl_one_rank.py"""
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
local_rank = int(os.environ.get("LOCAL_RANK", "0"))
dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"))
torch.cuda.set_device(local_rank)
dev = f"cuda:{local_rank}"
print(f"[rccl] backend {dist.get_backend()} world {dist.get_world_size()} rank {dist.get_rank()} torch {torch.__version__} hip {torch.version.hip}", flush=True)

import apnrf_amd  # noqa: F401,E402
from apnrf_amd import distributed as DD  # noqa: E402
from apnrf_amd import render as RD  # noqa: E402
from apnrf_amd import scenes as SC  # noqa: E402
from apnrf_amd.optim import FusedAdam  # noqa: E402


def timed(label, fn, n=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    print(f"[rccl] {label}: {(time.perf_counter() - t0) / n * 1e6:.1f} us per call", flush=True)


scene = SC.make_scene("102344250", n_poses=8)
f0, e0 = SC.hip_field(scene, dev), SC.hip_estimator(scene, dev)
group = dist.group.WORLD
# 1. the scoring exchange: [V/N, 4] float64 per rank
local = torch.rand(32, 4, dtype=torch.float64, device=dev)
out = torch.empty(32 * dist.get_world_size(), 4, dtype=torch.float64, device=dev)
timed("all_gather_into_tensor [32,4] f64 (score terms)", lambda: dist.all_gather_into_tensor(out, local, group=group))
assert torch.equal(out, local)
# 2. weights after a training phase: the three parameter vectors + occupancy grid
timed("broadcast mlp_base.params (%.0f MB)" % (f0.mlp_base.params.numel() * 4 / 1e6), lambda: dist.broadcast(f0.mlp_base.params.data, src=0, group=group))
DD.broadcast_model(f0, e0, src=0)
# 3. ray-data-parallel training: gradient all-reduce + skip flag
g = torch.rand_like(f0.mlp_base.params)
timed("all_reduce grad mlp_base (%.0f MB)" % (g.numel() * 4 / 1e6), lambda: dist.all_reduce(g, op=dist.ReduceOp.SUM, group=group))
skip = torch.zeros((), dtype=torch.int32, device=dev)
dist.all_reduce(skip, op=dist.ReduceOp.SUM, group=group)
# 4. the package's entry points with the group handed in
poses = scene["poses"][:4]
terms, score = RD.score_views([f0], [e0], poses, 640, 640, 320.0, 0.1, 1e-3, 0.1, 0.004, 0.01, dev, group=group)
print(f"[rccl] score_views(group=WORLD): terms {tuple(terms.shape)} score {float(score):.6f}", flush=True)
f0.train(); e0.train()
opt = FusedAdam(f0.parameters(), lr=1e-3, eps=1e-15).bind_field(f0)
o = torch.from_numpy(np.tile(scene["poses"][0][:3].astype(np.float32), (256, 1))).to(dev)
d = torch.nn.functional.normalize(torch.randn(256, 3, device=dev), dim=-1)
r = RD.train_step(f0, e0, opt, RD.Rays(o, d), torch.rand(256, 3, device=dev), torch.rand(256, device=dev) * 3, torch.randint(0, 29, (256,), device=dev),
                  torch.zeros(3, device=dev), step=1, data_parallel=True, data_parallel_group=group, **SC.RENDER_KW)
print(f"[rccl] train_step(data_parallel=True): loss {float(r['loss']):.4f} samples {r['n_rendering_samples']} skipped {r['skipped']}", flush=True)
dist.barrier()
dist.destroy_process_group()
print("[rccl] ok", flush=True)
