"""The N>1 path on CPU: two `gloo` processes shard the candidate views exactly as score_views does on GPUs
(contiguous slices, one all-gather of the [V,4] float64 terms) and must reproduce the single-process result.
The per-view terms are produced by the oracle scorer here (no GPU in this test); the exchange code under test is
the product's (`render.shard_views`, `render.gather_view_terms`, `render.trajectory_score`)."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _renders(V=7, P=40, C=29, M=2):
    rng = np.random.default_rng(12)
    return ((rng.random((M, 1, V, P, 1, 3)) ** 3 * 0.1).astype(np.float32), (rng.random((M, 1, V, P, 1)) ** 3).astype(np.float32),
            rng.random((M, 1, V, P, 1)).astype(np.float32), (rng.normal(size=(M, 1, V, P, 1, C)) * 2).astype(np.float32))


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, REPO)
    import apnrf_amd  # noqa: F401
    from apnrf_amd import render as RD
    from oracle import scorer as SC
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    rv, dv, ac, sm = _renders()
    V = ac.shape[2]
    lo, hi, per = RD.shard_views(V, world, rank)
    local = torch.zeros(per, 4, dtype=torch.float64)
    if hi > lo:
        sl = (slice(None), slice(None), slice(lo, hi))
        local[:hi - lo] = torch.from_numpy(SC.per_view_terms(rv[sl], dv[sl], ac[sl], sm[sl]))
    terms = RD.gather_view_terms(local, V)
    score = RD.trajectory_score(terms)
    np.save(os.path.join(out_dir, f"terms_{rank}.npy"), terms.numpy())
    np.save(os.path.join(out_dir, f"score_{rank}.npy"), np.asarray(score.item()))
    dist.barrier()
    dist.destroy_process_group()


def test_view_sharding_two_ranks_gloo(tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    sys.path.insert(0, REPO)
    from oracle import scorer as SC
    rv, dv, ac, sm = _renders()
    ref = SC.per_view_terms(rv, dv, ac, sm)
    t0, t1 = np.load(tmp_path / "terms_0.npy"), np.load(tmp_path / "terms_1.npy")
    np.testing.assert_array_equal(t0, t1)                      # every rank holds identical terms -> identical argmax
    np.testing.assert_allclose(t0, ref, rtol=1e-12, atol=0)
    total = SC.predictive_information(rv, dv, ac, sm)
    np.testing.assert_allclose(np.load(tmp_path / "score_0.npy"), total, rtol=1e-12)
    assert np.load(tmp_path / "score_0.npy") == np.load(tmp_path / "score_1.npy")


def test_shard_views_covers_all_views():
    sys.path.insert(0, REPO)
    import apnrf_amd  # noqa: F401
    from apnrf_amd import render as RD
    for V in (1, 7, 8, 256, 257):
        for world in (1, 2, 4, 8):
            seen = []
            for r in range(world):
                lo, hi, per = RD.shard_views(V, world, r)
                assert hi - lo <= per
                seen += list(range(lo, hi))
            assert seen == list(range(V))


def _grad_worker(rank, world, port, out_dir):
    sys.path.insert(0, REPO)
    import apnrf_amd  # noqa: F401
    from apnrf_amd import render as RD
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    g = torch.Generator().manual_seed(100 + rank)
    params = [torch.nn.Parameter(torch.zeros(n)) for n in (1000, 37, 5)]
    for p in params[:2]:
        p.grad = torch.randn(p.shape, generator=g)          # the third parameter has no gradient on any rank
    RD.allreduce_gradients(params)
    np.save(os.path.join(out_dir, f"grads_{rank}.npy"), torch.cat([p.grad for p in params[:2]]).numpy())
    assert params[2].grad is not None and float(params[2].grad.abs().max()) == 0.0      # a parameter without a gradient takes part as zeros (every rank issues the same collectives)
    dist.barrier()
    dist.destroy_process_group()


def test_ray_data_parallel_gradient_average_two_ranks_gloo(tmp_path):
    """`render.allreduce_gradients` (what `train_step(data_parallel=True)` runs between backward and the NaN guard):
    both ranks end with the mean of the per-rank gradients."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_grad_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    per_rank = []
    for r in range(2):
        g = torch.Generator().manual_seed(100 + r)
        per_rank.append(torch.cat([torch.randn(1000, generator=g), torch.randn(37, generator=g)]))
    mean = ((per_rank[0] + per_rank[1]) / 2).numpy()
    g0, g1 = np.load(tmp_path / "grads_0.npy"), np.load(tmp_path / "grads_1.npy")
    np.testing.assert_array_equal(g0, g1)
    np.testing.assert_allclose(g0, mean, rtol=1e-6, atol=1e-7)


# ------------------------------------------------------------------ §8e rows 2-3: sharded rendering exchange, weight broadcast, placement
def _worker_helpers(rank, world, port, out_dir):
    sys.path.insert(0, REPO)
    import apnrf_amd  # noqa: F401
    from apnrf_amd import distributed as DD
    from apnrf_amd.nerfacc import OccGridEstimator
    from apnrf_amd.ngp import NGPRadianceField
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    # the all-gather of packed per-ray rows: 5 views x 7 rays, D = 9; rank-local rows are a deterministic function of the ray id
    V, R, D = 5, 7, 9
    sh = DD.shard_rays(V, R, world, rank)
    r0, r1 = sh["lo"] * sh["unit_rays"], sh["hi"] * sh["unit_rays"]
    ids = torch.arange(r0, r1, dtype=torch.float32)
    local = ids[:, None] * 10 + torch.arange(D, dtype=torch.float32)[None]
    full = DD.gather_rows(local, sh["per"] * sh["unit_rays"], V * R)
    np.save(os.path.join(out_dir, f"rows_{rank}.npy"), full.numpy())
    # weights + occupancy broadcast from rank 1 (the rank that "trained")
    f = NGPRadianceField([0.0, 0, 0, 1, 1, 1], neurons=64, layers=1, num_semantic_classes=5, log2_hashmap_size=8, seed=rank)
    e = OccGridEstimator([0.0, 0, 0, 1, 1, 1], resolution=[4, 3, 5])
    e.occs.fill_(float(rank)); e.binaries[:] = bool(rank)
    v0 = f.mlp_head.params._version
    DD.broadcast_model(f, e, src=1)
    assert f.mlp_head.params._version > v0            # the field handle will reload its fp16 copies
    np.save(os.path.join(out_dir, f"head_{rank}.npy"), f.mlp_head.params.detach().numpy())
    np.save(os.path.join(out_dir, f"occ_{rank}.npy"), np.concatenate([e.occs.numpy(), e.binaries.numpy().reshape(-1).astype(np.float32)]))
    np.save(os.path.join(out_dir, f"members_{rank}.npy"), np.asarray(DD.my_members(5)))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_render_exchange_broadcast_and_placement_gloo(tmp_path):
    sys.path.insert(0, REPO)
    import apnrf_amd  # noqa: F401
    from apnrf_amd import distributed as DD
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker_helpers, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    want = np.arange(35, dtype=np.float32)[:, None] * 10 + np.arange(9, dtype=np.float32)[None]
    for r in (0, 1):
        np.testing.assert_array_equal(np.load(tmp_path / f"rows_{r}.npy"), want)            # every rank holds every ray's row
    np.testing.assert_array_equal(np.load(tmp_path / "head_0.npy"), np.load(tmp_path / "head_1.npy"))
    occ = np.load(tmp_path / "occ_0.npy")
    assert (occ == 1.0).all() and (np.load(tmp_path / "occ_1.npy") == 1.0).all()            # rank 1's grid everywhere
    assert np.load(tmp_path / "members_0.npy").tolist() == [0, 2, 4] and np.load(tmp_path / "members_1.npy").tolist() == [1, 3]
    # sharding arithmetic (pure): whole views when there are enough of them, row tiles otherwise; every ray exactly once
    for V, R, W in ((8, 640000, 8), (5, 7, 2), (1, 640000, 8), (3, 4096, 8), (2, 150, 4)):
        covered = []
        for rank in range(W):
            sh = DD.shard_rays(V, R, W, rank)
            assert R % sh["tiles_per_view"] == 0 and sh["unit_rays"] * sh["tiles_per_view"] == R
            covered += list(range(sh["lo"] * sh["unit_rays"], sh["hi"] * sh["unit_rays"]))
            assert (sh["tiles_per_view"] == 1) == (V >= W)
        assert covered == list(range(V * R))
    assert DD.ensemble_placement(2, 8) == [0, 1] and DD.ensemble_placement(5, 2) == [0, 1, 0, 1, 0]


# ------------------------------------------------------------------ every rank reaches every collective of a data-parallel step
def _uneven_worker(rank, world, port, out_dir):
    sys.path.insert(0, REPO)
    import apnrf_amd  # noqa: F401
    from apnrf_amd import render as RD
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    # three parameter vectors + one without elements (the reference's direction_encoding); rank 1's backward "produced nothing": its
    # gradients are None (the autograd path when a rank's rays give no sample) — it must still take part in the same collectives
    params = [torch.nn.Parameter(torch.zeros(n)) for n in (0, 1000, 37, 5)]
    skip = torch.zeros((), dtype=torch.int32)
    if rank == 0:
        g = torch.Generator().manual_seed(7)
        for p in params:
            if p.numel():
                p.grad = torch.randn(p.shape, generator=g)
    else:
        skip += 1                                   # "no sample survived": this rank raises the step's skip flag
    RD.allreduce_gradients(params, None, skip)
    np.save(os.path.join(out_dir, f"uneven_{rank}.npy"), np.concatenate([p.grad.numpy() for p in params if p.numel()] + [np.array([float(skip)])]))
    dist.barrier()
    dist.destroy_process_group()


def test_data_parallel_collectives_with_an_empty_rank_two_ranks_gloo(tmp_path):
    """`allreduce_gradients` when one rank has no gradients at all (its batch rendered no sample): both ranks issue the same collectives
    (round 2's train_step returned early on the empty rank and left the other one inside dist.all_reduce), the averaged gradient is
    half of rank 0's, and the summed skip flag is raised on BOTH ranks, so both leave the optimizer step out together."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_uneven_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    g = torch.Generator().manual_seed(7)
    want = np.concatenate([(torch.randn(n, generator=g) / 2).numpy() for n in (1000, 37, 5)] + [np.array([1.0])])
    g0, g1 = np.load(tmp_path / "uneven_0.npy"), np.load(tmp_path / "uneven_1.npy")
    np.testing.assert_array_equal(g0, g1)
    np.testing.assert_allclose(g0, want, rtol=1e-6, atol=1e-7)
