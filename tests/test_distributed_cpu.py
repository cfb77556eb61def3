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
    assert params[2].grad is None
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
