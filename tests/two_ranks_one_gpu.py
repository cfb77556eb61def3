"""Child of test_two_processes_on_one_gpu_end_to_end: rank `argv[1]` of `argv[2]`, all on cuda:0, `gloo` backend (RCCL refuses two ranks on one device).
The multi-GPU paths of SURVEY 8e END TO END with two real processes and the real kernels:
  (1) a stand-in trained on rank 0 whose cache file cannot be written reaches rank 1 by broadcast (standin.shared_standin, the path bench.py takes on a read-only home);
  (2) view-sharded scoring (render.score_views): every rank holds all terms, equal to its own single-process recomputation BIT FOR BIT;
  (3) view-sharded rendering (distributed.render_views_sharded) == the single-process render bit for bit;
  (4) ray-data-parallel train steps (train_step(data_parallel=True)): parameters stay identical on both ranks."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
import helpers as H  # noqa: E402
from apnrf_amd import distributed as DD  # noqa: E402
from apnrf_amd import render as RD  # noqa: E402
from apnrf_amd import standin as SI  # noqa: E402
from apnrf_amd.optim import FusedAdam  # noqa: E402

rank, world = int(sys.argv[1]), int(sys.argv[2])
dist.init_process_group("gloo", rank=rank, world_size=world)
DEV = "cuda:0"


def same_on_all_ranks(t, what):
    mine = t.detach().cpu().contiguous()
    both = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(both, mine)
    for b in both[1:]:
        assert torch.equal(both[0], b), what


# (1) one training, no cache file: the model travels by broadcast
sc = H.make_scene(log2_hashmap_size=14)
field, est, info = SI.shared_standin(sc, DEV, steps=60, seed=3, cache_dir="/proc/mnf-cache-that-cannot-exist")
assert (rank == 0 and info.get("saved") is False) or (rank != 0 and info.get("received_by_broadcast")), (rank, {k: v for k, v in info.items() if k != "optimizer_state"})
for p in field.parameters():
    if p.numel():
        same_on_all_ranks(p, "stand-in parameters")
same_on_all_ranks(est.binaries.to(torch.uint8), "stand-in occupancy grid")
assert int(est.binaries.sum()) > 0

# (2) view-sharded scoring with two members (the second: the same weights perturbed identically on every rank)
f2 = H.hip_field(sc)
f2.load_state_dict(field.state_dict())
with torch.no_grad():
    f2.mlp_head.params.mul_(1.01)
poses = sc["poses"][[0, 1, 2, 3, 4, 5, 6]]               # 7 views over 2 ranks: 4 + 3
args = (640, 640, 320.0, 0.1, 1e-3, 0.025, 0.004, 0.01, DEV)
terms, score = RD.score_views([field, f2], [est, est], poses, *args)
alone, score_alone = RD.score_views([field, f2], [est, est], poses, *args, group=False)
assert terms.shape == (7, 4) and torch.equal(terms, alone) and float(score) == float(score_alone)
same_on_all_ranks(terms, "gathered score terms")

# (3) view-sharded rendering
os_, ds_ = zip(*[H.view_rays(sc, p, h=16, w=16) for p in range(5)])
o, d = torch.cat(os_).to(DEV), torch.cat(ds_).to(DEV)
bk = torch.zeros(3)
full = RD.render_views(field, est, o, d, 256, 1024, render_bkgd=bk, probabilistic=True, **H.RENDER_KW)
shard = DD.render_views_sharded(field, est, o, d, 256, probabilistic=True, max_samples=1024, render_bkgd=bk, **H.RENDER_KW)
for k in ("rgb", "acc", "depth", "sem", "rgb_var", "depth_var"):
    assert torch.equal(shard[k], full[k]), k
assert torch.equal(shard["total"], full["total"])

# (4) ray-data-parallel training: each rank renders its own slice of the batch, gradients averaged, same update everywhere
field.train(); est.train()
opt = FusedAdam(field.parameters(), lr=1e-3, eps=1e-15)
rng = np.random.default_rng(10 + rank)
for step in range(1, 4):
    ro, rd = H.view_rays(sc, (2 * step + rank) % 8, h=16, w=16)
    pix = torch.from_numpy(rng.random((256, 3)).astype(np.float32)).to(DEV)
    dep = torch.from_numpy(rng.uniform(0.5, 4.0, 256).astype(np.float32)).to(DEV)
    lab = torch.from_numpy(rng.integers(0, 29, 256)).to(DEV)
    out = RD.train_step(field, est, opt, RD.Rays(ro.to(DEV), rd.to(DEV)), pix, dep, lab, torch.zeros(3, device=DEV), step=step, data_parallel=True,
                        stratified=False, **H.RENDER_KW)
    assert not out["skipped"], (rank, step)
for p in field.parameters():
    if p.numel():
        same_on_all_ranks(p, "parameters after data-parallel steps")
dist.barrier()
print("TWO_RANKS_ONE_GPU_OK", rank, flush=True)
dist.destroy_process_group()
