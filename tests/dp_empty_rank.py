"""Child of test_data_parallel_step_with_an_empty_rank_does_not_hang: rank `argv[1]` of `argv[2]`, all on cuda:0, gloo."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
import helpers as H  # noqa: E402
from apnrf_amd import render as RD  # noqa: E402
from apnrf_amd.optim import FusedAdam  # noqa: E402

rank, world = int(sys.argv[1]), int(sys.argv[2])
dist.init_process_group("gloo", rank=rank, world_size=world)
DEV = "cuda:0"
sc = H.make_scene(log2_hashmap_size=14)
hip, est = H.hip_field(sc), H.hip_estimator(sc)
o, d = H.view_rays(sc, 2, h=16, w=16)
rng = np.random.default_rng(1)
pix = torch.from_numpy(rng.random((256, 3)).astype(np.float32)).to(DEV)
dep = torch.from_numpy(rng.uniform(0.5, 4.0, 256).astype(np.float32)).to(DEV)
lab = torch.from_numpy(rng.integers(0, 29, 256)).to(DEV)
o, d = o.to(DEV), d.to(DEV)
if rank == 1:                                   # this rank's slice of the batch misses the grid entirely
    o = o + torch.tensor([0.0, 100.0, 0.0], device=DEV)
    d = torch.zeros_like(d); d[:, 1] = 1.0
for fused in (True, False):
    opt = FusedAdam(hip.parameters(), lr=1e-3, eps=1e-15)
    before = hip.mlp_head.params.detach().clone()
    out = RD.train_step(hip, est, opt, RD.Rays(o, d), pix, dep, lab, torch.zeros(3, device=DEV), step=1, data_parallel=True, fused=fused,
                        stratified=False, **H.RENDER_KW)
    # one rank had no sample -> the summed skip flag is raised on BOTH ranks: nobody steps (the reference's `continue`, taken together)
    assert out["skipped"] and torch.equal(hip.mlp_head.params.detach(), before), (rank, fused, out)
    dist.barrier()
# both ranks with samples: the averaged gradient is applied on both, identically
if rank == 1:
    o, d = H.view_rays(sc, 3, h=16, w=16)
    o, d = o.to(DEV), d.to(DEV)
opt = FusedAdam(hip.parameters(), lr=1e-3, eps=1e-15)
out = RD.train_step(hip, est, opt, RD.Rays(o, d), pix, dep, lab, torch.zeros(3, device=DEV), step=1, data_parallel=True, stratified=False, **H.RENDER_KW)
assert not out["skipped"]
mine = hip.mlp_head.params.detach().cpu()
both = [torch.empty_like(mine) for _ in range(world)]
dist.all_gather(both, mine)
assert torch.equal(both[0], both[1])
dist.barrier()
print("DP_EMPTY_RANK_OK", rank)
dist.destroy_process_group()
