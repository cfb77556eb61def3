"""GPU parity tests: the HIP path (through the C ABI) against the oracle on identical seeded inputs
and against the committed golden vectors.  Bit-exact for integer / index / marcher outputs; stated
tolerances for floating point."""
import numpy as np
import pytest
import torch

import helpers as H

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def scene():
    return H.make_scene()


@pytest.fixture(scope="module")
def fields(scene):
    return H.hip_field(scene), H.oracle_field(scene)


def _rays(n, seed, box=(-1.0, 1.0)):
    rng = np.random.default_rng(seed)
    o = rng.uniform(box[0] * 1.5, box[1] * 1.5, size=(n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    return o, d


def _cu(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    return (t if dtype is None else t.to(dtype)).to(DEV)


# ------------------------------------------------------------------ marcher (bit-exact gate)
def test_ray_aabb_bit_exact(golden):
    from apnrf_amd import nerfacc as NA
    from oracle import marcher as M
    g = golden("aabb")
    for near, far in [(-np.inf, np.inf), (0.1, 1.5)]:
        t0, t1, h = NA.ray_aabb_intersect(_cu(g["rays_o"]), _cu(g["rays_d"]), _cu(g["aabbs"]), near, far)
        r0, r1, rh = M.ray_aabb_intersect(g["rays_o"], g["rays_d"], g["aabbs"], near, far)
        np.testing.assert_array_equal(h.cpu().numpy(), rh)
        np.testing.assert_array_equal(t0.cpu().numpy(), r0)
        np.testing.assert_array_equal(t1.cpu().numpy(), r1)
    # empty input
    t0, t1, h = NA.ray_aabb_intersect(torch.zeros(0, 3, device=DEV), torch.zeros(0, 3, device=DEV), _cu(g["aabbs"]))
    assert t0.shape == (0, g["aabbs"].shape[0])


def _assert_traverse_equal(got, ref):
    (iv, sm, term), (riv, rsm, rterm) = got, ref
    np.testing.assert_array_equal(sm.packed_info.cpu().numpy(), rsm.packed_info)
    np.testing.assert_array_equal(iv.packed_info.cpu().numpy(), riv.packed_info)
    np.testing.assert_array_equal(iv.is_left.cpu().numpy(), riv.is_left)
    np.testing.assert_array_equal(iv.is_right.cpu().numpy(), riv.is_right)
    used = riv.is_left | riv.is_right
    np.testing.assert_array_equal(iv.vals.cpu().numpy()[used], riv.vals[used])          # bit-exact t values
    np.testing.assert_array_equal(iv.ray_indices.cpu().numpy()[used], riv.ray_indices[used])
    if rsm.is_valid is not None:
        np.testing.assert_array_equal(sm.is_valid.cpu().numpy(), rsm.is_valid)
        v = rsm.is_valid
    else:
        v = np.ones(len(rsm.vals), bool)
    np.testing.assert_array_equal(sm.vals.cpu().numpy()[v], rsm.vals[v])
    np.testing.assert_array_equal(sm.ray_indices.cpu().numpy()[v], rsm.ray_indices[v])
    return term, rterm


@pytest.mark.parametrize("n_grids,cone", [(1, 0.0), (1, 0.004), (3, 0.004)])
def test_traverse_grids_two_pass_bit_exact(n_grids, cone):
    from apnrf_amd import nerfacc as NA
    from oracle import marcher as M
    from oracle import occgrid as OG
    rng = np.random.default_rng(5)
    o, d = _rays(777, 11)
    d[0] = [1, 0, 0]; d[1] = [0, 0, -1]          # axis-parallel rays (dir == 0 branches)
    o[2] = [5, 5, 5]; d[2] = [1, 0, 0]           # a ray that misses every box
    aabbs = np.stack([OG.enlarge_aabb([-1, -1, -1, 1, 1, 1], 2 ** i) for i in range(n_grids)])
    binaries = rng.random((n_grids, 24, 9, 17)) > 0.6
    near = rng.uniform(0.0, 0.3, 777).astype(np.float32)
    far = np.full(777, 1e10, np.float32)
    ref = M.traverse_grids(o, d, binaries, aabbs, near, far, 5e-3, cone)
    got = NA.traverse_grids(_cu(o), _cu(d), _cu(binaries), _cu(aabbs), _cu(near), _cu(far), 5e-3, cone)
    term, rterm = _assert_traverse_equal(got, ref)
    hit = ref[1].packed_info[:, 1] > 0      # the fill pass skips empty rays before writing their plane (grid.cu:106-110)
    np.testing.assert_array_equal(term.cpu().numpy()[hit], rterm[hit])
    assert ref[1].packed_info[:, 1].sum() > 10000
    # reference property (tests/test_grid.py:39-68): every sample lies in an occupied cell
    iv, sm, _ = got
    ts, te, ri = (t.cpu().numpy() for t in (iv.vals[iv.is_left], iv.vals[iv.is_right], sm.ray_indices))
    assert ts.shape == te.shape == ri.shape
    occ, sel = OG.query(o[ri] + d[ri] * ((ts + te)[:, None] / np.float32(2.0)), binaries, aabbs[0])
    assert occ.all() and sel.all()


@pytest.mark.parametrize("seed", [0, 1, 2, 3, 4, 5])
def test_traverse_grids_randomized_bit_exact(seed):
    """Random grids, boxes, cone angles, step sizes and near planes (including rays that start inside the box, far
    away from it, or very close to a face): every t value of the HIP marcher equals the oracle's, bit for bit.  Covers
    the fast paths of csrc/march_dev.h (hang guard hoisted out of the loops, incremental cell index, branch-free axis
    step), which must not change a single rounding."""
    from apnrf_amd import nerfacc as NA
    from oracle import marcher as M
    rng = np.random.default_rng(100 + seed)
    res = rng.integers(3, 40, 3)
    lo = rng.uniform(-3, 0, 3).astype(np.float32)
    aabb = np.concatenate([lo, lo + rng.uniform(0.5, 6, 3).astype(np.float32)])[None].astype(np.float32)
    binaries = rng.random((1, *res)) > rng.uniform(0.3, 0.9)
    n = 600
    o = (aabb[0, :3] + rng.uniform(-0.5, 1.5, (n, 3)) * (aabb[0, 3:] - aabb[0, :3])).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32); d /= np.linalg.norm(d, axis=-1, keepdims=True)
    d[:20, 0] = 0.0; d[:20] /= np.linalg.norm(d[:20], axis=-1, keepdims=True)          # rays inside a coordinate plane
    o[20:40] = aabb[0, :3] + np.float32(1e-4)                                           # origins hugging a corner
    cone = float(rng.choice([0.0, 0.001, 0.004, 0.02]))
    step = float(rng.choice([1e-3, 5e-3, 3e-2]))
    near = rng.uniform(0.0, 0.5, n).astype(np.float32)
    far = np.where(rng.random(n) < 0.3, rng.uniform(1.0, 4.0, n), 1e10).astype(np.float32)
    ref = M.traverse_grids(o, d, binaries, aabb, near, far, step, cone)
    got = NA.traverse_grids(_cu(o), _cu(d), _cu(binaries), _cu(aabb), _cu(near), _cu(far), step, cone)
    _assert_traverse_equal(got, ref)
    assert ref[1].packed_info[:, 1].sum() > 500


def test_traverse_grids_no_hits_and_empty():
    from apnrf_amd import nerfacc as NA
    o = np.full((5, 3), 10.0, np.float32)
    d = np.tile(np.array([[1.0, 0, 0]], np.float32), (5, 1))
    aabbs = np.array([[0.0, 0, 0, 1, 1, 1]], np.float32)
    iv, sm, term = NA.traverse_grids(_cu(o), _cu(d), _cu(np.ones((1, 4, 4, 4), bool)), _cu(aabbs))
    assert iv.vals.numel() == 0 and sm.vals.numel() == 0 and (sm.packed_info[:, 1] == 0).all()
    iv, sm, term = NA.traverse_grids(torch.zeros(0, 3, device=DEV), torch.zeros(0, 3, device=DEV),
                                     _cu(np.ones((1, 4, 4, 4), bool)), _cu(aabbs))
    assert sm.packed_info.shape == (0, 2)


def test_traverse_grids_chunked_bit_exact(scene):
    """the over-allocated mode the test-time renderer uses (grid.cu:364-404), three chained rounds"""
    from apnrf_amd import nerfacc as NA
    from oracle import marcher as M
    o, d = (t.numpy() for t in H.view_rays(scene, 2, h=24, w=24))
    aabbs = scene["aabb"][None]
    n = o.shape[0]
    near = np.full(n, 0.1, np.float32); far = np.full(n, 1e10, np.float32)
    mask = np.ones(n, bool)
    t_mins, t_maxs, hits = M.ray_aabb_intersect(o, d, aabbs)
    t_sorted = np.concatenate([t_mins, t_maxs], -1)
    t_idx = np.broadcast_to(np.arange(2, dtype=np.int64), (n, 2)).copy()
    for limit in (4, 7, 64):
        ref = M.traverse_grids(o, d, scene["occ"], aabbs, near, far, 1e-3, 0.004, limit, True, mask, t_sorted, t_idx, hits)
        got = NA.traverse_grids(_cu(o), _cu(d), _cu(scene["occ"]), _cu(aabbs), _cu(near), _cu(far), 1e-3, 0.004, limit, True,
                                _cu(mask), _cu(t_sorted), _cu(t_idx), _cu(hits))
        term, rterm = _assert_traverse_equal(got, ref)
        np.testing.assert_array_equal(term.cpu().numpy()[mask], rterm[mask])
        near = rterm.copy()
        mask = ref[1].packed_info[:, 1] == limit
        assert mask.any()


# ------------------------------------------------------------------ ray generation (bit-exact vs the reference golden)
@pytest.mark.parametrize("cone,stratified", [(0.004, False), (0.0, False), (0.004, True)])
def test_single_pass_sampler_bit_exact(scene, cone, stratified):
    """csrc/march.hip sample_rays_kernel + compact_samples_kernel (what OccGridEstimator.sampling runs) against the
    two-pass traverse_grids of the same library and against the oracle marcher: identical samples, bit for bit."""
    from apnrf_amd import nerfacc as NA
    from oracle import marcher as M
    est = H.hip_estimator(scene)
    o, d = H.view_rays(scene, 2, h=24, w=24)
    o, d = o.to(DEV), d.to(DEV)
    step = 1e-3 if cone > 0 else 2e-2
    near = torch.full((o.shape[0],), 0.1, device=DEV)
    if stratified:
        near = near + torch.rand(o.shape[0], generator=torch.Generator().manual_seed(5)).to(DEV) * step
    far = torch.full_like(near, 1e10)
    got = est._sample_single_pass(o, d, near, far, step, cone)
    assert got is not None
    ri, ts, te, packed = got
    iv, sm, _ = NA.traverse_grids(o, d, est.binaries, est.aabbs, near_planes=near, far_planes=far, step_size=step, cone_angle=cone)
    np.testing.assert_array_equal(ts.cpu().numpy(), iv.vals[iv.is_left].cpu().numpy())
    np.testing.assert_array_equal(te.cpu().numpy(), iv.vals[iv.is_right].cpu().numpy())
    np.testing.assert_array_equal(ri.cpu().numpy(), sm.ray_indices.cpu().numpy())
    np.testing.assert_array_equal(packed.cpu().numpy(), sm.packed_info.cpu().numpy())
    assert ts.shape[0] > 1000
    ref = M.traverse_grids(o.cpu().numpy(), d.cpu().numpy(), scene["occ"], scene["aabb"][None], near_planes=near.cpu().numpy(),
                           far_planes=far.cpu().numpy(), step_size=step, cone_angle=cone)
    np.testing.assert_array_equal(ts.cpu().numpy(), ref[0].vals[ref[0].is_left])
    # a scratch row that is too short is reported, not truncated silently: the mirror falls back to two passes
    assert int(packed[:, 1].max()) > 16 and est._sample_single_pass(o, d, near, far, step, cone, cap=16) is None
    ri2, ts2, te2 = est.sampling(o, d, near_plane=0.1, render_step_size=step, cone_angle=cone)
    if not stratified:
        np.testing.assert_array_equal(ts2.cpu().numpy(), ts.cpu().numpy())


def test_generate_image_rays_bit_exact(golden):
    from apnrf_amd import render as RD
    g = golden("raygen")
    for k in range(3):
        W, Hh, scale = g[f"c{k}_whs"]
        W, Hh = int(W), int(Hh)
        focal = float(g[f"c{k}_focal"])
        K = np.array([[focal, 0, W / 2], [0, focal, Hh / 2], [0, 0, 1.0]])
        np.testing.assert_array_equal(RD.pose_to_c2w(g[f"c{k}_pose"]).astype(np.float32), g[f"c{k}_c2w"][0])
        idx = RD.subsample_indices(W * Hh, int(Hh * scale) * int(W * scale))
        np.testing.assert_array_equal(idx, g[f"c{k}_idx"])
        rays = RD.generate_image_rays(torch.from_numpy(g[f"c{k}_c2w"]), W, Hh, K, DEV, idx)
        np.testing.assert_array_equal(rays.origins.cpu().numpy(), g[f"c{k}_origins"])
        np.testing.assert_array_equal(rays.viewdirs.cpu().numpy(), g[f"c{k}_viewdirs"])
    # full image == indexed full image
    full = RD.generate_image_rays(torch.from_numpy(g["c0_c2w"]), 64, 64, np.array([[32.0, 0, 32], [0, 32.0, 32], [0, 0, 1]]), DEV)
    assert full.origins.shape == (4096, 3)


# ------------------------------------------------------------------ scans / volrend
def test_exclusive_sum_and_weights(golden):
    from apnrf_amd import nerfacc as NA
    from oracle import marcher as M
    from oracle import render as R
    rng = np.random.default_rng(3)
    cnts = rng.integers(0, 70, 500)
    cnts[::7] = 0
    ri = np.repeat(np.arange(500), cnts)
    x = rng.random(len(ri)).astype(np.float32)
    pk = M.pack_info(ri, 500)
    np.testing.assert_array_equal(NA.pack_info(_cu(ri), 500).cpu().numpy(), pk)
    np.testing.assert_array_equal(NA.pack_info_grouped(_cu(ri), 500).cpu().numpy(), pk)
    np.testing.assert_array_equal(NA.pack_info(_cu(np.array([0, 2, 2, 2, 2])), 3).cpu().numpy(), [[0, 1], [1, 0], [1, 4]])
    xs = _cu(x).requires_grad_(True)
    y = NA.exclusive_sum(xs, _cu(pk))
    np.testing.assert_array_equal(y.detach().cpu().numpy(), M.exclusive_sum(x, pk))          # same sequential order
    g = rng.random(len(ri)).astype(np.float32)
    y.backward(_cu(g))
    np.testing.assert_array_equal(xs.grad.cpu().numpy(), M.exclusive_sum(g, pk, backward=True))
    # weights / visibility against the golden captured from the reference's batched branch
    gv = golden("volrend")
    Rn, S = gv["sigmas"].shape
    rid = _cu(np.repeat(np.arange(Rn), S))
    ts, te, sg = (_cu(gv[k].reshape(-1)) for k in ("t_starts", "t_ends", "sigmas"))
    w, tr, al = NA.render_weight_from_density(ts, te, sg, ray_indices=rid, n_rays=Rn)
    np.testing.assert_allclose(w.cpu().numpy().reshape(Rn, S), gv["weights"], rtol=3e-5, atol=1e-6)
    np.testing.assert_allclose(tr.cpu().numpy().reshape(Rn, S), gv["trans"], rtol=3e-5, atol=1e-7)
    vis = NA.render_visibility_from_density(ts, te, sg, ray_indices=rid, n_rays=Rn, early_stop_eps=1e-2, alpha_thre=0.05)
    assert (vis.cpu().numpy().reshape(Rn, S) == gv["vis"]).mean() > 0.9995
    # known answers (tests/test_rendering.py:117-133) including the gradient through the packed scan
    sig = _cu(np.array([0.4, 0.8, 0.1, 0.8, 0.1], np.float32)).requires_grad_(True)
    t0 = torch.rand(5, device=DEV)
    w, _, _ = NA.render_weight_from_density(t0, t0 + 1.0, sig, ray_indices=_cu(np.array([0, 2, 2, 2, 2])), n_rays=3)
    w.sum().backward()
    np.testing.assert_allclose(w.detach().cpu().numpy(), [0.3297, 0.5507, 0.0428, 0.2239, 0.0174], atol=1e-4)
    np.testing.assert_allclose(sig.grad.cpu().numpy(), [0.6703, 0.1653, 0.1653, 0.1653, 0.1653], atol=1e-4)


@pytest.mark.parametrize("C,bkgd", [(29, True), (13, False), (64, True), (0, True)])
def test_composite_train_forward_backward_matches_oracle(C, bkgd):
    """csrc/composite_train.hip against the op chain of utils.py:362-461 under CPU autograd (oracle/render.py):
    ragged rays (empty, 1, 63..65, 200 samples), all four heads contributing to the loss."""
    from apnrf_amd.render import _CompositeTrain
    from oracle import marcher as M
    from oracle import render as R
    rng = np.random.default_rng(C + 1)
    cnts = rng.integers(0, 90, 300)
    cnts[::7] = 0
    cnts[1:6] = [1, 63, 64, 65, 200]
    n_rays, ri = len(cnts), np.repeat(np.arange(len(cnts)), cnts)
    N = len(ri)
    ts = (rng.random(N) * 3).astype(np.float32)
    te = ts + (0.01 + rng.random(N) * 0.05).astype(np.float32)
    sig, rgb, sem = (rng.random(N) * 4).astype(np.float32), rng.random((N, 3)).astype(np.float32), rng.normal(size=(N, C)).astype(np.float32)
    bk = np.array([0.2, 0.5, 0.9], np.float32) if bkgd else None
    g = [rng.normal(size=s).astype(np.float32) for s in ((n_rays, 3), (n_rays, 1), (n_rays, 1), (n_rays, C))]
    pk = M.pack_info(ri, n_rays)

    # oracle: CPU autograd over the reference's op chain
    o_sig, o_rgb, o_sem = (torch.from_numpy(a).requires_grad_(True) for a in (sig, rgb, sem))
    t_ts, t_te, t_ri = torch.from_numpy(ts), torch.from_numpy(te), torch.from_numpy(ri)
    w, tr, al = R.render_weight_from_density(t_ts, t_te, o_sig, pk)
    colors = R.accumulate_along_rays(w, o_rgb, t_ri, n_rays)
    opac = R.accumulate_along_rays(w, None, t_ri, n_rays)
    depth = R.accumulate_along_rays(w, (t_ts + t_te)[:, None] / 2.0, t_ri, n_rays) / opac.clamp_min(torch.finfo(torch.float32).eps)
    semo = R.accumulate_along_rays(w, o_sem, t_ri, n_rays)
    if bkgd:
        colors = colors + torch.from_numpy(bk) * (1.0 - opac)
    sum((o * torch.from_numpy(gi)).sum() for o, gi in zip((colors, opac, depth, semo), g)).backward()

    h_sig, h_rgb, h_sem = (_cu(a).requires_grad_(True) for a in (sig, rgb, sem))
    outs = _CompositeTrain.apply(_cu(pk[:, 0]), _cu(pk[:, 1]), _cu(ts), _cu(te), h_sig, h_rgb, h_sem, _cu(bk) if bkgd else None)
    sum((o * _cu(gi)).sum() for o, gi in zip(outs[:4], g)).backward()
    for got, want, name in zip(outs, (colors, opac, depth, semo, w, tr, al), "rgb acc depth sem weights trans alphas".split()):
        np.testing.assert_allclose(got.detach().cpu().numpy(), want.detach().numpy(), rtol=2e-5, atol=2e-6, err_msg=name)
    for got, want, name in ((h_sig, o_sig, "d_sigma"), (h_rgb, o_rgb, "d_rgb"), (h_sem, o_sem, "d_sem")):
        scale = float(want.grad.abs().max()) if want.grad.numel() else 1.0
        np.testing.assert_allclose(got.grad.cpu().numpy(), want.grad.numpy(), rtol=1e-4, atol=2e-5 * max(scale, 1.0), err_msg=name)


# ------------------------------------------------------------------ field (fp: tolerance)
def test_grid_meta_matches_oracle(fields):
    from oracle.field import grid_levels
    hip, orc = fields
    scale, res, size, off, hashed = hip.grid_meta()
    lv, total = grid_levels(orc.cfg)
    assert [int(x["res"]) for x in lv] == res and [x["n"] for x in lv] == size and [x["offset"] for x in lv] == off
    assert [int(x["hashed"]) for x in lv] == hashed
    np.testing.assert_array_equal(np.asarray(scale, np.float32), np.asarray([x["scale"] for x in lv], np.float32))
    assert res[:6] == [16, 24, 34, 49, 71, 102] and hashed[:6] == [0, 0, 0, 0, 0, 1]
    assert total == 6299960 and sum(size) == total      # 6 299 960 entries x 4 features (16 levels, T = 2^19)


@pytest.mark.parametrize("neurons,layers,C,lh", [(128, 2, 29, 19), (64, 4, 29, 15), (128, 4, 13, 14), (64, 1, 32, 12),
                                                 (128, 1, 5, 12), (128, 3, 29, 12), (64, 2, 17, 12), (64, 3, 1, 12)])
def test_field_forward_matches_oracle(neurons, layers, C, lh):
    """Tolerance: 1e-3 abs on rgb and semantic logits, 2e-3 relative on density (north star: 1e-3 abs on
    rendered outputs; density is an exponential of an fp16-rounded network so it is checked relatively)."""
    sc = H.make_scene(neurons=neurons, layers=layers, C=C, log2_hashmap_size=lh, head_gain=4.0)
    hip, orc = H.hip_field(sc), H.oracle_field(sc)
    rng = np.random.default_rng(1)
    n = 5000 + 37                                                  # ragged tail (not a multiple of 64)
    a = sc["aabb"]
    pos = (rng.random((n, 3)) * (a[3:] - a[:3]) * 1.1 + a[:3] - 0.05 * (a[3:] - a[:3])).astype(np.float32)   # some outside the box
    d = rng.normal(size=(n, 3)).astype(np.float32); d /= np.linalg.norm(d, axis=-1, keepdims=True)
    with torch.no_grad():
        rgb, sigma, sem = hip(_cu(pos), _cu(d))
        dens = hip.query_density(_cu(pos))
    r_rgb, r_sigma, r_sem = orc(torch.from_numpy(pos), torch.from_numpy(d))
    assert sem.shape == (n, C) and rgb.shape == (n, 3) and sigma.shape == (n, 1)
    inside = (r_sigma[:, 0] > 0).numpy()
    assert (sigma.cpu()[:, 0] > 0).numpy().tolist() == inside.tolist()          # selector mask is exact
    np.testing.assert_allclose(sigma.cpu().numpy(), r_sigma.numpy(), rtol=2e-3, atol=1e-6)
    np.testing.assert_array_equal(dens.cpu().numpy(), sigma.cpu().numpy())      # density-only kernel == full kernel
    np.testing.assert_allclose(rgb.cpu().numpy()[inside], r_rgb.numpy()[inside], atol=1e-3, rtol=0)
    # head_gain=4 amplifies the logits (|sem| up to ~2): one fp16 rounding flip of a hidden activation moves a logit
    # by ~1e-3 * |logit|, hence the relative term
    np.testing.assert_allclose(sem.cpu().numpy()[inside], r_sem.numpy()[inside], atol=1e-3, rtol=2e-3)
    assert np.abs(r_sem.numpy()).max() > 0.3                                     # the comparison is not vacuous


def test_field_forward_tcnn_init_scale_table():
    """Hash table at tiny-cuda-nn's initialisation scale U(-1e-4, 1e-4): most entries are fp16 SUBNORMALS (< 6.1e-5).
    The gather blend must widen them exactly (no flush to zero), as the oracle's fp16 -> fp32 conversion does."""
    sc = H.make_scene(neurons=64, layers=1, C=13, seed=4, log2_hashmap_size=14)
    n_mlp = 64 * 64 + 16 * 64
    tab = sc["params"]["mlp_base"][n_mlp:]
    tab[:] = np.random.default_rng(0).uniform(-1e-4, 1e-4, tab.shape).astype(np.float32)
    sc["params"]["mlp_base"][:64 * 64] *= 200.0       # amplify the first layer so the tiny features matter in the outputs
    hip, orc = H.hip_field(sc), H.oracle_field(sc)
    rng = np.random.default_rng(1)
    x = (sc["aabb"][:3] + rng.random((4096, 3)).astype(np.float32) * (sc["aabb"][3:] - sc["aabb"][:3])).astype(np.float32)
    d = rng.normal(size=(4096, 3)).astype(np.float32); d /= np.linalg.norm(d, axis=-1, keepdims=True)
    with torch.no_grad():
        rgb, sig, sem = hip(_cu(x), _cu(d))
        r_rgb, r_sig, r_sem = orc(torch.from_numpy(x), torch.from_numpy(d))
    assert float(r_sem.abs().max()) > 1e-3        # the features do reach the outputs
    np.testing.assert_allclose(sig.cpu().numpy(), r_sig.numpy(), rtol=2e-3, atol=1e-6)
    np.testing.assert_allclose(sem.cpu().numpy(), r_sem.numpy(), atol=1e-3 * float(r_sem.abs().max()), rtol=2e-3)
    np.testing.assert_allclose(rgb.cpu().numpy(), r_rgb.numpy(), atol=1e-3)


def test_field_forward_samples_matches_positions(scene, fields):
    hip, _ = fields
    o, d = H.view_rays(scene, 3, h=16, w=16)
    rng = np.random.default_rng(2)
    ri = np.sort(rng.integers(0, 256, 3000))
    ts = rng.uniform(0.2, 4.0, 3000).astype(np.float32)
    te = ts + 0.003
    rgb, sigma, sem = hip.forward_samples(o.to(DEV), d.to(DEV), _cu(ri), _cu(ts), _cu(te))
    pos = o[ri] + d[ri] * (torch.from_numpy(ts) + torch.from_numpy(te))[:, None] / 2.0
    with torch.no_grad():
        rgb2, sigma2, sem2 = hip(pos.to(DEV), d[ri].to(DEV))
    np.testing.assert_array_equal(sigma.cpu().numpy(), sigma2.cpu().numpy()[:, 0])
    np.testing.assert_array_equal(rgb.cpu().numpy(), rgb2.cpu().numpy())
    np.testing.assert_array_equal(sem.cpu().numpy(), sem2.cpu().numpy())
    (s_only,) = hip.forward_samples(o.to(DEV), d.to(DEV), _cu(ri), _cu(ts), _cu(te), density_only=True)
    np.testing.assert_array_equal(s_only.cpu().numpy(), sigma.cpu().numpy())


# ------------------------------------------------------------------ fused renderers
def _check_render(out, ref, prob, max_tie_rays=0):
    """North star: RGB / depth / semantic within 1e-3 abs.  `max_tie_rays` rays may sit outside it: keeping a sample is
    a threshold decision (alpha >= alpha_thre, opacity <= 1 - 1e-4), and a ray whose alpha lands within fp16 rounding of
    the threshold legitimately flips between two correct implementations; such rays must still be close (5e-2)."""
    keys = [("rgb", 1e-3, 0.0), ("acc", 1e-3, 0.0), ("depth", 1e-3, 1e-3), ("sem", 1e-3, 0.0)]
    if prob:
        keys += [("rgb_var", 1e-3, 0.0), ("depth_var", 2e-3, 2e-3)]
    bad = np.zeros(ref["rgb"].shape[0], bool)
    for k, atol, rtol in keys:
        got, want = out[k].cpu().numpy(), ref[k].numpy()
        viol = np.abs(got - want) > atol + rtol * np.abs(want)
        if max_tie_rays == 0:
            np.testing.assert_allclose(got, want, atol=atol, rtol=rtol, err_msg=k)
        else:
            np.testing.assert_allclose(got, want, atol=5e-2, rtol=5e-2, err_msg=k)
        bad |= viol.reshape(viol.shape[0], -1).any(1)
    assert bad.sum() <= max_tie_rays, f"{bad.sum()} rays outside 1e-3"
    mse = float(((out["rgb"].cpu() - ref["rgb"]) ** 2).mean())
    assert 10.0 * np.log10(1.0 / max(mse, 1e-20)) > 50.0       # north star: PSNR within 0.1 dB of the reference render -> > 50 dB against it
    tot = int(out["total"][0].item())
    assert abs(tot - ref["total_samples"]) <= max(3, 0.002 * ref["total_samples"]), (tot, ref["total_samples"])


@pytest.mark.parametrize("prob", [False, True])
def test_render_test_matches_oracle(scene, fields, prob):
    from apnrf_amd import render as RD
    from oracle import render as R
    hip, orc = fields
    est = H.hip_estimator(scene)
    o, d = H.view_rays(scene, 1, h=32, w=32)
    bk = torch.tensor([0.1, 0.3, 0.6])
    fn = R.render_prob_test if prob else R.render_test
    ref = fn(1024, orc, scene["occ"], scene["aabb"][None], o, d, render_bkgd=bk, **H.RENDER_KW)
    assert len(ref["rounds"]) > 5 and ref["total_samples"] > 20000
    out = RD.render_views(hip, est, o.to(DEV), d.to(DEV), 1024, 1024, render_bkgd=bk, probabilistic=prob, **H.RENDER_KW)
    _check_render(out, ref, prob)
    # the reference-signature wrappers, [H,W,3]-shaped rays (utils.py:574-580)
    rays = RD.Rays(o.view(32, 32, 3).to(DEV), d.view(32, 32, 3).to(DEV))
    if prob:
        rgb, rgb_var, acc, depth, depth_var, sem, tot = RD.render_probablistic_image_with_occgrid_test(
            1024, hip, est, rays, render_bkgd=bk.to(DEV), **H.RENDER_KW)
        assert rgb_var.shape == (32, 32, 3) and depth_var.shape == (32, 32, 1)
    else:
        rgb, acc, depth, sem, tot = RD.render_image_with_occgrid_test(1024, hip, est, rays, render_bkgd=bk.to(DEV), **H.RENDER_KW)
    assert rgb.shape == (32, 32, 3) and sem.shape == (32, 32, 29) and isinstance(tot, int)
    np.testing.assert_array_equal(rgb.reshape(-1, 3).cpu().numpy(), out["rgb"].cpu().numpy())


def test_render_baseline_config2_model_matches_oracle():
    """BASELINE config 2: scene 102344250 with the 4 x 64 base MLP (32-wide heads) and a smaller class count: the
    fused renderer's other kernel instantiation (field_kernel<64, 4, 2, ...>), deterministic and probabilistic."""
    from apnrf_amd import render as RD
    from oracle import render as R
    sc = H.make_scene("102344250", neurons=64, layers=4, C=13, seed=3)
    hip, orc, est = H.hip_field(sc), H.oracle_field(sc), H.hip_estimator(sc)
    o, d = H.view_rays(sc, 5, width=256, height=256, h=24, w=24)
    bk = torch.tensor([0.9, 0.1, 0.4])
    for prob in (False, True):
        fn = R.render_prob_test if prob else R.render_test
        ref = fn(1024, orc, sc["occ"], sc["aabb"][None], o, d, render_bkgd=bk, **H.RENDER_KW)
        assert ref["total_samples"] > 5000
        out = RD.render_views(hip, est, o.to(DEV), d.to(DEV), o.shape[0], 1024, render_bkgd=bk, probabilistic=prob, **H.RENDER_KW)
        _check_render(out, ref, prob, max_tie_rays=2)      # 576 rays; the deeper fp16 chain flips one alpha-threshold tie


def test_render_batched_views_equal_single_calls(scene, fields):
    """Each group of rays_per_view rays must behave as its own reference call (own round schedule); sync_every must
    not change results; batched == single calls bit for bit."""
    from apnrf_amd import render as RD
    hip, _ = fields
    est = H.hip_estimator(scene)
    rays = [H.view_rays(scene, k, h=16, w=16) for k in (0, 2, 5)]
    o = torch.cat([r[0] for r in rays]).to(DEV); d = torch.cat([r[1] for r in rays]).to(DEV)
    bk = torch.zeros(3)
    batched = RD.render_views(hip, est, o, d, 256, 1024, render_bkgd=bk, probabilistic=True, sync_every=0, **H.RENDER_KW)
    for k in range(3):
        single = RD.render_views(hip, est, o[k * 256:(k + 1) * 256], d[k * 256:(k + 1) * 256], 256, 1024, render_bkgd=bk,
                                 probabilistic=True, sync_every=3, **H.RENDER_KW)
        for key in ("rgb", "acc", "depth", "sem", "rgb_var", "depth_var"):
            # a view is marched by its own workgroups and packed into its own tiles, so its rays see the same lane
            # positions (= the same summation trees) whether the view is rendered alone or inside a batch: bit-identical.
            # This is what makes view-sharded scoring on N GPUs reproduce the 1-GPU terms exactly (bench.py score256).
            np.testing.assert_array_equal(batched[key][k * 256:(k + 1) * 256].cpu().numpy(), single[key].cpu().numpy(), err_msg=key)
        assert int(batched["total"][0]) > 0
    # marching a view's rays in 8x8 pixel blocks (mnf_render_opts.view_order) changes tile composition, not results
    ordered = RD.render_views(hip, est, o, d, 256, 1024, render_bkgd=bk, probabilistic=True, image_hw=(16, 16), **H.RENDER_KW)
    for key in ("rgb", "acc", "depth", "sem", "rgb_var", "depth_var"):
        np.testing.assert_allclose(ordered[key].cpu().numpy(), batched[key].cpu().numpy(), rtol=2e-6, atol=1e-7, err_msg=key)
    assert int(ordered["total"][0]) == int(batched["total"][0]) and int(ordered["total"][1]) == int(batched["total"][1])
    # ragged sizes: 3 views of 150 rays (neither a multiple of 64 nor of the 1024-ray march workgroup) == single calls
    o3, d3 = o[:450].contiguous(), d[:450].contiguous()
    ragged = RD.render_views(hip, est, o3, d3, 150, 1024, render_bkgd=bk, **H.RENDER_KW)
    for k in range(3):
        single = RD.render_views(hip, est, o3[k * 150:(k + 1) * 150], d3[k * 150:(k + 1) * 150], 150, 1024, render_bkgd=bk, **H.RENDER_KW)
        for key in ("rgb", "acc", "depth", "sem"):
            np.testing.assert_allclose(ragged[key][k * 150:(k + 1) * 150].cpu().numpy(), single[key].cpu().numpy(),
                                       rtol=2e-6, atol=1e-7, err_msg=key)
    # views with very different per-ray budgets side by side (a nearly dead view runs at 64 samples per ray while its
    # neighbours run at 4): 12 views x 256 rays, every third view keeps only 4 rays that hit the grid.  A march workgroup
    # must not stretch the neighbours' column slots to the dying view's budget (the workspace is sized for per-view budgets).
    ov, dv = [], []
    for k in range(12):
        ok, dk = (t.clone() for t in H.view_rays(scene, k % 8, h=16, w=16))
        if k % 3 == 1:
            ok[4:] = 100.0; dk[4:] = torch.tensor([0.0, 1.0, 0.0])
        ov.append(ok); dv.append(dk)
    ov, dv = torch.cat(ov).to(DEV), torch.cat(dv).to(DEV)
    mixed = RD.render_views(hip, est, ov, dv, 256, 1024, render_bkgd=bk, **H.RENDER_KW)
    for k in (0, 1, 2, 10):
        single = RD.render_views(hip, est, ov[k * 256:(k + 1) * 256], dv[k * 256:(k + 1) * 256], 256, 1024, render_bkgd=bk, **H.RENDER_KW)
        for key in ("rgb", "acc", "depth", "sem"):
            np.testing.assert_allclose(mixed[key][k * 256:(k + 1) * 256].cpu().numpy(), single[key].cpu().numpy(),
                                       rtol=2e-6, atol=1e-7, err_msg=f"mixed budgets, view {k}, {key}")
    # rays that miss the grid entirely: zero opacity, background colour, no samples
    far_o = torch.full((64, 3), 100.0, device=DEV)
    far_d = torch.tensor([[0.0, 1.0, 0.0]], device=DEV).repeat(64, 1)
    miss = RD.render_views(hip, est, far_o, far_d, 64, 1024, render_bkgd=torch.tensor([0.2, 0.4, 0.8]), **H.RENDER_KW)
    assert int(miss["total"][0].item()) == 0 and (miss["acc"] == 0).all()
    np.testing.assert_allclose(miss["rgb"].cpu().numpy(), np.tile([[0.2, 0.4, 0.8]], (64, 1)), atol=1e-7)


def test_train_mode_forward_matches_oracle(scene, fields):
    """utils.py:63-219 forward: two-pass sampling + density pre-pass + visibility filter + sem_rendering."""
    from apnrf_amd import render as RD
    from oracle import render as R
    hip, orc = fields
    est = H.hip_estimator(scene)
    o, d = H.view_rays(scene, 4, h=12, w=12)
    bk = torch.tensor([0.5, 0.2, 0.9])
    near = torch.full((o.shape[0],), 0.1)
    ref = R.render_train(orc, scene["occ"], scene["aabb"][None], float(est.occs.mean().item()), o, d, near,
                         render_bkgd=bk, render_step_size=1e-3, cone_angle=0.004, alpha_thre=0.01)
    hip.eval()   # eval: no stratified jitter (the jitter comes from the device RNG), single 8192-ray chunk
    with torch.no_grad():   # forward values through the fused no-grad kernels (the differentiable path: test_train_step_matches_oracle)
        rgb, acc, depth, sem, n = RD.render_image_with_occgrid_with_depth_guide(
            hip, est, RD.Rays(o.to(DEV), d.to(DEV)), render_bkgd=bk.to(DEV), **H.RENDER_KW)
    assert abs(n - ref[4]) <= max(3, 0.002 * ref[4]) and ref[4] > 1000
    np.testing.assert_allclose(rgb.cpu().numpy(), ref[0].numpy(), atol=1e-3)
    np.testing.assert_allclose(acc.cpu().numpy(), ref[1].numpy(), atol=1e-3)
    np.testing.assert_allclose(depth.cpu().numpy(), ref[2].numpy(), atol=1e-3, rtol=1e-3)
    np.testing.assert_allclose(sem.cpu().numpy(), ref[3].numpy(), atol=1e-3)


# ------------------------------------------------------------------ scorer
def test_score_views_matches_oracle():
    from apnrf_amd import render as RD
    from oracle import scorer as SC
    for M in (2, 3, 5):      # (the kernel has an unrolled form for ensembles of up to four members and a general one: csrc/render.hip score_kernel)
        _check_score_views(RD, SC, M)


def _check_score_views(RD, SC, M):
    rng = np.random.default_rng(4)
    V, P, C = 5, 300, 29
    rgb_var = (rng.random((M, V, P, 3)) ** 4 * 0.1).astype(np.float32)
    depth_var = (rng.random((M, V, P)) ** 4).astype(np.float32)
    depth_var[0, 0, :10] = 0.0
    acc = rng.random((M, V, P)).astype(np.float32); acc[1, 2, :20] = 1.0; acc[0, 3, :20] = 0.0
    sem = (rng.normal(size=(M, V, P, C)) * 3).astype(np.float32)
    terms = RD.score_view_terms(_cu(rgb_var), _cu(depth_var), _cu(acc), _cu(sem)).cpu().numpy()
    # oracle layout [M,1,V,h,w,...] (pipeline.py:720-725)
    ref = SC.per_view_terms(rgb_var[:, None, :, :, None], depth_var[:, None, :, :, None], acc[:, None, :, :, None], sem[:, None, :, :, None])
    np.testing.assert_allclose(terms, ref, rtol=1e-9, atol=1e-12)
    total = SC.predictive_information(rgb_var[:, None, :, :, None], depth_var[:, None, :, :, None], acc[:, None, :, :, None],
                                      sem[:, None, :, :, None])
    np.testing.assert_allclose((terms[:, 0] + terms[:, 1] + 3 * terms[:, 2] + 2 * terms[:, 3]).mean(), total, rtol=1e-9)


def test_score_views_end_to_end(scene):
    """poses -> rays -> probabilistic renders of two ensemble members -> per-view terms, vs the oracle pipeline"""
    from apnrf_amd import render as RD
    from oracle import render as R
    from oracle import scorer as SC
    sc2 = dict(scene); sc2["params"] = H.S.make_field_params(seed=1)
    hips = [H.hip_field(scene), H.hip_field(sc2)]
    orcs = [H.oracle_field(scene), H.oracle_field(sc2)]
    ests = [H.hip_estimator(scene), H.hip_estimator(scene)]
    poses = scene["poses"][[1, 6]]
    terms, score = RD.score_views(hips, ests, poses, 640, 640, 320.0, 0.1, 1e-3, 0.025, 0.004, 0.01, DEV)
    outs = []
    for orc in orcs:
        per = []
        for p in poses:
            idx = R.subsample_indices(640 * 640, 16 * 16)
            o, d = R.generate_image_rays(R.pose_to_c2w(p), 640, 640, 320.0, idx)
            per.append(R.render_prob_test(1024, orc, scene["occ"], scene["aabb"][None], o, d, render_bkgd=torch.zeros(3), **H.RENDER_KW))
        outs.append(per)
    stack = lambda key: np.asarray([[[o[key].numpy().reshape(16, 16, -1) for o in per]] for per in outs])
    ref = SC.per_view_terms(stack("rgb_var"), stack("depth_var")[..., 0], stack("acc")[..., 0], stack("sem"))
    np.testing.assert_allclose(terms.cpu().numpy(), ref, atol=5e-3, rtol=5e-3)


# ------------------------------------------------------------------ training: backward of the field and the train step
def _grad_close(got, want, name, rel=2e-2, cos=0.9995):
    got, want = got.double().cpu().numpy().ravel(), want.double().numpy().ravel()
    denom = np.linalg.norm(want)
    assert denom > 0, name
    err = np.linalg.norm(got - want) / denom
    c = float(got @ want / (np.linalg.norm(got) * denom + 1e-300))
    assert err < rel and c > cos, f"{name}: rel L2 err {err:.3e}, cos {c:.6f}"


@pytest.mark.parametrize("neurons,layers,C,lh", [(128, 2, 29, 14), (64, 4, 13, 12)])
def test_field_backward_matches_oracle(neurons, layers, C, lh):
    """dL/d(params) of the HIP backward (fp16 activation gradients, loss scale 128) against torch autograd through the
    oracle (fp32 gradient of the same fp16-rounded forward).  Tolerance: 2e-2 relative L2, cosine > 0.9995 per parameter group."""
    sc = H.make_scene(neurons=neurons, layers=layers, C=C, log2_hashmap_size=lh, head_gain=2.0)
    hip = H.hip_field(sc).train()
    orc = H.oracle_field(sc, requires_grad=True)
    rng = np.random.default_rng(7)
    n = 3000 + 21
    a = sc["aabb"]
    pos = (rng.random((n, 3)) * (a[3:] - a[:3]) * 0.98 + a[:3] + 0.01 * (a[3:] - a[:3])).astype(np.float32)
    pos[:5] = a[:3] - 1.0                                          # outside the box: density gradient must vanish
    d = rng.normal(size=(n, 3)).astype(np.float32); d /= np.linalg.norm(d, axis=-1, keepdims=True)
    g_rgb = (rng.normal(size=(n, 3)) * 1e-3).astype(np.float32)
    g_sig = (rng.normal(size=(n, 1)) * 1e-5).astype(np.float32)
    g_sem = (rng.normal(size=(n, C)) * 1e-3).astype(np.float32)
    rgb, sigma, sem = hip(_cu(pos), _cu(d))
    assert rgb.requires_grad and sem.requires_grad
    torch.autograd.backward([rgb, sigma, sem], [_cu(g_rgb), _cu(g_sig), _cu(g_sem)])
    r_rgb, r_sigma, r_sem = orc(torch.from_numpy(pos), torch.from_numpy(d))
    np.testing.assert_allclose(rgb.detach().cpu().numpy()[5:], r_rgb.detach().numpy()[5:], atol=1e-3)       # train forward == inference forward
    torch.autograd.backward([r_rgb, r_sigma, r_sem], [torch.from_numpy(g_rgb), torch.from_numpy(g_sig), torch.from_numpy(g_sem)])
    n_mlp = sum(o * i for o, i in orc.shapes["base"])
    _grad_close(hip.mlp_base.params.grad[:n_mlp], orc.p_base.grad[:n_mlp], "base mlp")
    _grad_close(hip.mlp_base.params.grad[n_mlp:], orc.p_base.grad[n_mlp:], "hash table")
    _grad_close(hip.mlp_head.params.grad, orc.p_head.grad, "rgb head")
    _grad_close(hip.mlp_sem.params.grad, orc.p_sem.grad, "sem head")
    # entries never touched by a sample get exactly zero gradient
    untouched = (orc.p_base.grad[n_mlp:] == 0).numpy()
    assert (hip.mlp_base.params.grad[n_mlp:].cpu().numpy()[untouched] == 0).all()


def test_train_step_matches_oracle(scene):
    """utils.py:63-219 forward + the loss of pipeline.py:506-511 + backward, then one Adam step (pipeline.py:173-178)."""
    import torch.nn.functional as F
    from apnrf_amd import render as RD
    from oracle import render as R
    sc = H.make_scene(log2_hashmap_size=15)
    hip, orc = H.hip_field(sc), H.oracle_field(sc, requires_grad=True)
    hip.eval()     # no stratified jitter (device RNG); gradients still flow
    est = H.hip_estimator(sc)
    o, d = H.view_rays(sc, 4, h=12, w=12)
    rng = np.random.default_rng(3)
    pix = torch.from_numpy(rng.random((144, 3)).astype(np.float32))
    dep = torch.from_numpy(rng.uniform(0.5, 4.0, 144).astype(np.float32))
    lab = torch.from_numpy(rng.integers(0, 29, 144))
    bk = torch.tensor([0.5, 0.2, 0.9])

    def loss_fn(rgb, depth, sem, pix, dep, lab):
        return F.smooth_l1_loss(rgb, pix) * 10 + F.smooth_l1_loss(depth, dep.unsqueeze(1)) / 5 + F.cross_entropy(sem, lab) / 2

    opt = torch.optim.Adam(hip.parameters(), lr=1e-3, eps=1e-15)
    rgb, acc, depth, sem, n = RD.render_image_with_occgrid_with_depth_guide(hip, est, RD.Rays(o.to(DEV), d.to(DEV)),
                                                                            render_bkgd=bk.to(DEV), **H.RENDER_KW)
    assert n > 1000 and rgb.requires_grad
    loss = loss_fn(rgb, depth, sem, pix.to(DEV), dep.to(DEV), lab.to(DEV))
    opt.zero_grad()
    loss.backward()
    ref = R.render_train(orc, sc["occ"], sc["aabb"][None], float(est.occs.mean().item()), o, d, torch.full((144,), 0.1),
                         render_bkgd=bk, render_step_size=1e-3, cone_angle=0.004, alpha_thre=0.01)
    r_loss = loss_fn(ref[0], ref[2], ref[3], pix, dep, lab)
    r_loss.backward()
    np.testing.assert_allclose(loss.item(), r_loss.item(), rtol=1e-4)
    n_mlp = sum(o_ * i_ for o_, i_ in orc.shapes["base"])
    _grad_close(hip.mlp_base.params.grad[:n_mlp], orc.p_base.grad[:n_mlp], "base mlp", rel=3e-2, cos=0.999)
    _grad_close(hip.mlp_base.params.grad[n_mlp:], orc.p_base.grad[n_mlp:], "hash table", rel=3e-2, cos=0.999)
    _grad_close(hip.mlp_head.params.grad, orc.p_head.grad, "rgb head", rel=3e-2, cos=0.999)
    _grad_close(hip.mlp_sem.params.grad, orc.p_sem.grad, "sem head", rel=3e-2, cos=0.999)
    before = hip.mlp_head.params.detach().clone()
    assert not any(torch.isnan(p.grad).any() for p in hip.parameters() if p.grad is not None)     # pipeline.py:520-524
    opt.step()
    assert (hip.mlp_head.params.detach() != before).any()
    with torch.no_grad():                                          # the updated parameters are picked up by the kernels
        rgb2, *_ = RD.render_image_with_occgrid_with_depth_guide(hip, est, RD.Rays(o.to(DEV), d.to(DEV)), render_bkgd=bk.to(DEV), **H.RENDER_KW)
    assert (rgb2 - rgb.detach()).abs().max() > 0


def test_train_loop_reduces_loss():
    """End to end through `render.train_step` exactly as pipeline.py:447-532 sequences it: occupancy refresh on step 0
    and 16, stratified sampling, train render, loss, HIP backward, NaN guard, FusedAdam at the reference's learning rate.
    Twenty iterations on one fixed ray batch with consistent targets must bring the loss down substantially (the
    degenerate constant target would drive the density to overflow if trained much longer); the estimator stays a valid grid."""
    from apnrf_amd import render as RD
    from apnrf_amd.optim import FusedAdam
    sc = H.make_scene(log2_hashmap_size=15, seed=7)
    hip, est = H.hip_field(sc), H.hip_estimator(sc)
    opt = FusedAdam(hip.parameters(), lr=1e-3, eps=1e-15)
    o, d = H.view_rays(sc, 3, h=16, w=16)
    rays = RD.Rays(o.to(DEV), d.to(DEV))
    rng = np.random.default_rng(1)
    pix = torch.from_numpy(np.tile(rng.random((1, 3)).astype(np.float32), (256, 1))).to(DEV)     # one colour, one depth, one class
    dep = torch.full((256,), 1.5, device=DEV)
    lab = torch.full((256,), 7, dtype=torch.int64, device=DEV)
    bk = torch.tensor([0.3, 0.3, 0.3], device=DEV)
    losses = []
    for step in range(20):
        out = RD.train_step(hip, est, opt, rays, pix, dep, lab, bk, step=step, occ_thre=1e-3, **H.RENDER_KW)
        assert not out["skipped"] and out["n_rendering_samples"] > 0
        losses.append(float(out["loss"]))
    assert np.isfinite(losses).all()
    assert np.mean(losses[-3:]) < 0.5 * np.mean(losses[:3]), losses
    assert est.binaries.dtype == torch.bool and 0 < int(est.binaries.sum()) <= est.binaries.numel()


# ------------------------------------------------------------------ §8(f) rows: dataset ingest, checkpoints, planner map
def test_fused_adam_matches_torch_adam():
    """optim.FusedAdam (one HIP kernel per parameter) against torch.optim.Adam with the reference's hyper-parameters
    (pipeline.py:173-178: lr 1e-3, eps 1e-15) over several steps, state_dict interchange, and the NaN-gradient counter."""
    from apnrf_amd.optim import FusedAdam, count_nan_gradients
    g = torch.Generator().manual_seed(3)
    shapes = (100003, 4096, 37)
    init = [torch.randn(n, generator=g) * 0.1 for n in shapes]
    ref_p = [torch.nn.Parameter(t.clone().to(DEV)) for t in init]
    hip_p = [torch.nn.Parameter(t.clone().to(DEV)) for t in init]
    ref, hip = torch.optim.Adam(ref_p, lr=1e-3, eps=1e-15), FusedAdam(hip_p, lr=1e-3, eps=1e-15)
    for step in range(6):
        if step == 3:                       # a checkpoint written by torch's optimizer loads into the fused one
            import copy
            hip.load_state_dict(copy.deepcopy(ref.state_dict()))     # (a live state_dict aliases the tensors)
        for a, b in zip(ref_p, hip_p):
            grad = (torch.randn(a.shape, generator=g) * (10.0 ** (step - 3))).to(DEV)
            grad[::7] = 0.0                 # untouched hash-table entries have exactly zero gradient
            a.grad, b.grad = grad.clone(), grad.clone()
        v0 = hip_p[0]._version
        ref.step(); hip.step()
        assert hip_p[0]._version > v0       # the field handle notices the in-place update
        for a, b in zip(ref_p, hip_p):
            np.testing.assert_allclose(b.detach().cpu().numpy(), a.detach().cpu().numpy(), rtol=2e-6, atol=5e-8)
    sd = hip.state_dict()
    assert set(sd["state"][0]) == {"step", "exp_avg", "exp_avg_sq"} and float(sd["state"][0]["step"]) == 6
    assert int(count_nan_gradients(hip_p).item()) == 0
    hip_p[1].grad[5] = float("nan"); hip_p[2].grad[:3] = float("nan")
    assert int(count_nan_gradients(hip_p).item()) == 4


def test_dataset_matches_reference(golden, tmp_path):
    from apnrf_amd.dataset import Dataset
    g = golden("dataset")
    ds = Dataset(training=False, save_fp=str(tmp_path), num_models=2, device=DEV)
    np.random.seed(0)
    ds.update_data(g["images"][:2], g["depths"][:2], g["sems"][:2], g["c2w"][:2])
    ds.update_data(g["images"][2:], g["depths"][2:], g["sems"][2:], g["c2w"][2:])
    assert len(ds) == int(g["size"])
    np.testing.assert_array_equal(ds.K.cpu().numpy(), g["K"])
    np.testing.assert_array_equal(ds.bootstrap(0), g["boot0"])
    np.testing.assert_array_equal(ds.bootstrap(1), g["boot1"])         # same np.random stream as the reference
    d = ds[2]                                                           # evaluation branch: image 2, all pixels
    np.testing.assert_array_equal(d["pixels"].cpu().numpy(), g["pixels"])      # mnf_gather_pixels divides by 255.0f exactly as the CPU golden
    np.testing.assert_array_equal(d["dep"].cpu().numpy(), g["dep"])
    np.testing.assert_array_equal(d["sem"].cpu().numpy(), g["sem"])
    np.testing.assert_array_equal(d["rays"].origins.cpu().numpy(), g["origins"])
    np.testing.assert_array_equal(d["rays"].viewdirs.cpu().numpy(), g["viewdirs"])     # bit-exact rays
    np.testing.assert_array_equal(d["color_bkgd"].cpu().numpy(), g["color_bkgd"])
    # training branch: one random image, num_rays random pixels; values must be the gathered pixels of that image
    ds.training = True
    ds.update_num_rays(500)
    t = ds[0]
    iid, x, y = int(t["image_id"].item()), t["x"].cpu().numpy(), t["y"].cpu().numpy()
    assert t["pixels"].shape == (500, 3) and t["rays"].origins.shape == (500, 3) and t["color_bkgd"].shape == (3,)
    np.testing.assert_array_equal(t["pixels"].cpu().numpy(), g["images"][iid, y, x].astype(np.float32) / np.float32(255.0))
    np.testing.assert_array_equal(t["dep"].cpu().numpy(), g["depths"][iid, y, x])
    np.testing.assert_array_equal(t["sem"].cpu().numpy(), g["sems"][iid, y, x])
    ds.training = False
    full = ds[iid]
    np.testing.assert_array_equal(t["rays"].viewdirs.cpu().numpy(), full["rays"].viewdirs.cpu().numpy()[y, x])
    # packed on-device layout (u8 rgb / f16 depth / u8 class): same batches in the reference dtypes, depths to fp16
    pk = Dataset(training=False, save_fp=str(tmp_path), num_models=2, device=DEV, packed=True)
    pk.update_data(g["images"], g["depths"], g["sems"], g["c2w"])
    assert pk.depths.dtype == torch.float16 and pk.semantics.dtype == torch.uint8 and pk.images.dtype == torch.uint8
    dp = pk[2]
    np.testing.assert_array_equal(dp["pixels"].cpu().numpy(), g["pixels"])
    np.testing.assert_array_equal(dp["sem"].cpu().numpy(), g["sem"]); assert dp["sem"].dtype == torch.int64
    np.testing.assert_array_equal(dp["dep"].cpu().numpy(), g["dep"].astype(np.float16).astype(np.float32)); assert dp["dep"].dtype == torch.float32
    # bootstrap fix-forward flag: a training fetch uses the image it is asked for (default: a random one, as the reference)
    bs = Dataset(training=True, save_fp=str(tmp_path), num_rays=64, num_models=2, device=DEV, use_bootstrap_index=True)
    bs.update_data(g["images"], g["depths"], g["sems"], g["c2w"])
    assert all(int(bs[k]["image_id"].item()) == k for k in (2, 0, 1, 2))


def test_checkpoint_roundtrip_and_planner_map(scene, tmp_path):
    from apnrf_amd import dataset as DS
    from oracle import occgrid as OG
    hip = H.hip_field(scene)
    est = H.hip_estimator(scene)
    opt = torch.optim.Adam(hip.parameters(), lr=1e-3, eps=1e-15)
    path = str(tmp_path / "model.pth")
    DS.save_checkpoint(path, est, hip, opt)
    ck = torch.load(path, map_location="cpu")
    assert set(ck) == {"occ_grid", "model", "optimizer_state_dict"}                    # pipeline.py:630-635
    assert ck["occ_grid"].dtype == torch.bool and tuple(ck["occ_grid"].shape) == (1, *scene["res"])
    assert set(ck["model"]) == {"aabb", "direction_encoding.params", "mlp_base.params", "mlp_head.params", "mlp_sem.params"}
    sc2 = dict(scene); sc2["params"] = H.S.make_field_params(seed=5)
    hip2, est2 = H.hip_field(sc2), H.hip_estimator(scene)
    est2.binaries = torch.zeros_like(est2.binaries)
    DS.load_checkpoint(path, est2, hip2, map_location=DEV)
    assert torch.equal(est2.binaries, est.binaries)
    pos = torch.rand(1000, 3, device=DEV) * 3
    np.testing.assert_array_equal(hip2.query_density(pos).cpu().numpy(), hip.query_density(pos).cpu().numpy())
    # planner map of a two-member ensemble
    rng = np.random.default_rng(2)
    est2.binaries = torch.from_numpy(rng.random(scene["occ"].shape) > 0.9).to(DEV)
    a = scene["aabb"]
    state = np.array([-14.79, -10.60, 1.5]); aabb_xzy = np.array([a[0], a[2], a[1], a[3], a[5], a[4]])
    got = DS.planner_path_finding_map([est, est2], state, aabb_xzy, 0.2)
    ref = OG.planner_path_finding_map([est.binaries.cpu().numpy(), est2.binaries.cpu().numpy()], state, aabb_xzy, 0.2)
    np.testing.assert_array_equal(got, ref)
    assert got.shape == (scene["res"][0], scene["res"][2]) and 0 < got.mean() < 1


# ------------------------------------------------------------------ full-size properties (BASELINE config 3: 800x800)
def test_full_resolution_render_properties():
    """At the bench size the oracle cannot render the whole image, so check size-independent properties:
    determinism (two renders are bit-identical), value ranges, the evaluated/kept sample accounting, and a sparse
    sub-set of rays against the oracle rendered alone (a different round schedule, hence the looser tolerance)."""
    from apnrf_amd import render as RD
    from oracle import render as R
    sc = H.make_scene("102344529", n_poses=8)
    hip, est = H.hip_field(sc), H.hip_estimator(sc)
    c2w = RD.pose_to_c2w(sc["poses"][3]).astype(np.float32)[None]
    K = np.array([[400.0, 0, 400], [0, 400.0, 400], [0, 0, 1.0]])
    rays = RD.generate_image_rays(torch.from_numpy(c2w), 800, 800, K, DEV)
    bk = torch.tensor([0.2, 0.4, 0.6])
    a = RD.render_views(hip, est, rays.origins, rays.viewdirs, 640000, 1024, render_bkgd=bk, **H.RENDER_KW)
    b = RD.render_views(hip, est, rays.origins, rays.viewdirs, 640000, 1024, render_bkgd=bk, sync_every=3, **H.RENDER_KW)
    for key in ("rgb", "acc", "depth", "sem"):
        assert torch.equal(a[key], b[key]), key                     # deterministic, independent of the host sync cadence
    assert torch.equal(a["total"], b["total"])
    acc = a["acc"].cpu().numpy()
    assert np.isfinite(a["sem"].cpu().numpy()).all() and acc.min() >= 0 and acc.max() <= 1 + 1e-5
    rgb = a["rgb"].cpu().numpy()
    assert rgb.min() >= -1e-6 and rgb.max() <= 1 + 1e-5
    hit = acc[:, 0] > 0.5
    assert hit.mean() > 0.9 and (a["depth"].cpu().numpy()[hit, 0] >= 0.1).all()
    kept, evaluated = (int(x) for x in a["total"])
    assert 0 < kept <= evaluated and 20 * 640000 < evaluated < 200 * 640000
    # sparse check against the oracle
    idx = R.subsample_indices(640000, 400)
    o, d = rays.origins[idx].cpu(), rays.viewdirs[idx].cpu()
    ref = R.render_test(1024, H.oracle_field(sc), sc["occ"], sc["aabb"][None], o, d, render_bkgd=bk, **H.RENDER_KW)
    np.testing.assert_allclose(a["acc"][idx].cpu().numpy(), ref["acc"].numpy(), atol=2e-3)
    np.testing.assert_allclose(a["rgb"][idx].cpu().numpy(), ref["rgb"].numpy(), atol=2e-3)
    np.testing.assert_allclose(a["depth"][idx].cpu().numpy(), ref["depth"].numpy(), atol=5e-3, rtol=2e-3)
    np.testing.assert_allclose(a["sem"][idx].cpu().numpy(), ref["sem"].numpy(), atol=2e-3)


# ------------------------------------------------------------------ occupancy refresh on the device (a5)
def _golden_draws(g, k, occs, binaries, cells):
    """cell list + offsets of golden step k, derived from the reference's recorded RNG draws exactly as
    occ_grid.py:328-375 derives them from the state before the step"""
    step = int(g[f"s{k}_step"])
    draws = [g[[n for n in g.files if n.startswith(f"s{k}_draw{j}_")][0]] for j in range(int(g[f"s{k}_ndraws"]))]
    if step < 256:
        return step, np.arange(cells)[occs >= 0], draws[0]
    uni = draws[0]
    uni = uni[occs[uni] >= 0]
    occupied = np.nonzero(binaries)[0]
    if cells // 4 < len(occupied):
        return step, np.concatenate([uni, occupied[draws[1]]]), draws[2]
    return step, np.concatenate([uni, occupied]), draws[1]


def test_occupancy_update_golden_trajectory(golden):
    """`OccGridEstimator._update` (csrc/occupancy.hip) driven with the reference's recorded draws reproduces the
    reference's own trajectory (tests/golden/occgrid.npz: five updates, two of them warm-up): occs to 2e-5 relative
    (occ_eval_fn uses sin / cos, evaluated by torch on the GPU here and on the CPU in the golden), binaries bit-exact for
    every cell whose occupancy is not within that tolerance of the threshold, and the bit-packed grid equal to the bytes."""
    from apnrf_amd.nerfacc import OccGridEstimator
    g = golden("occgrid")
    res = g["resolution"].tolist()
    cells = int(np.prod(res))
    est = OccGridEstimator(torch.from_numpy(g["roi_aabb"]), resolution=res, levels=1).to(DEV).train()

    def occ_eval_fn(x):
        return (torch.sin(x[:, :1] * 1.3) * torch.cos(x[:, 2:3] * 0.7) + 0.2 * x[:, 1:2]).clamp_min(0) * 0.02

    occs, binaries = np.zeros(cells, np.float32), np.zeros(cells, bool)
    for k in range(5):
        step, idx, jit = _golden_draws(g, k, occs, binaries, cells)
        est._update(step=step, occ_eval_fn=occ_eval_fn, occ_thre=1e-2, _draws=[(torch.from_numpy(idx), torch.from_numpy(jit))])
        want_occs, want_bin = g[f"s{k}_occs"], g[f"s{k}_binaries"].reshape(-1)
        got_occs, got_bin = est.occs.cpu().numpy(), est.binaries.cpu().numpy().reshape(-1)
        np.testing.assert_allclose(got_occs, want_occs, rtol=2e-5, atol=1e-9, err_msg=f"step {step}")
        thre = min(float(want_occs[want_occs >= 0].mean()), 1e-2)
        clear = np.abs(want_occs - thre) > 2e-5 * max(thre, 1e-6) + 1e-9
        np.testing.assert_array_equal(got_bin[clear], want_bin[clear], err_msg=f"step {step}")
        assert clear.mean() > 0.99 and 0 < want_bin.sum() < cells
        bits = est.bitgrid().cpu().numpy().view(np.uint32).reshape(-1)
        unpacked = ((bits[:, None] >> np.arange(32, dtype=np.uint32)) & 1).reshape(-1)[:cells].astype(bool)
        np.testing.assert_array_equal(unpacked, got_bin)
        occs, binaries = want_occs, want_bin          # the next step's draws were made from the reference's own state
        est.occs.copy_(torch.from_numpy(want_occs)); est.binaries = torch.from_numpy(g[f"s{k}_binaries"]).to(DEV)


def test_occupancy_update_device_rng_and_fused_path(scene):
    """The production path: cells and offsets from the device Philox stream.  Properties that do not depend on the draws:
    warm-up touches every cell once (occs == max(0 * decay, density * step) of a point inside the cell); afterwards only
    listed cells change, the uniform half lands on ~N distinct cells, every occupied cell is re-evaluated while there are
    fewer than N of them, the same seed reproduces the update bit for bit, and the single-call fused form
    (mnf_update_occupancy) equals the three-call form with the same seed."""
    from apnrf_amd.nerfacc import FieldDensityOcc, OccGridEstimator
    hip = H.hip_field(scene)
    step_size = 1e-3
    res, cells = scene["res"], int(np.prod(scene["res"]))

    def fresh():
        return OccGridEstimator(torch.from_numpy(scene["aabb"]), resolution=res, levels=1).to(DEV).train()

    closure = lambda x: hip.query_density(x) * step_size
    a, b, c = fresh(), fresh(), fresh()
    for est, fn in ((a, closure), (b, closure), (c, FieldDensityOcc(hip, step_size))):
        torch.manual_seed(11)
        est._update(step=0, occ_eval_fn=fn, occ_thre=1e-3)
    assert torch.equal(a.occs, b.occs) and torch.equal(a.binaries, b.binaries)            # repeatable under torch.manual_seed
    assert torch.equal(a.occs, c.occs) and torch.equal(a.binaries, c.binaries)            # fused == three calls
    occ0 = a.occs.cpu().numpy()
    assert (occ0 > 0).all()                                                               # every cell was evaluated (density > 0 inside the box)
    thre = min(float(occ0.mean()), 1e-3)
    np.testing.assert_array_equal(a.binaries.cpu().numpy().reshape(-1), occ0 > np.float32(thre))
    # each warm-up value is the density of SOME point inside its own cell: compare with the cell's density range on a 3x3x3 lattice
    ids = np.random.default_rng(0).integers(0, cells, 200)
    X, Y, Z = res
    coords = np.stack([ids // (Y * Z), (ids // Z) % Y, ids % Z], -1).astype(np.float32)
    lat = np.stack(np.meshgrid(*[np.linspace(0.02, 0.98, 5)] * 3, indexing="ij"), -1).reshape(-1, 3).astype(np.float32)
    pts = scene["aabb"][:3] + (coords[:, None, :] + lat[None]) / np.asarray(res, np.float32) * (scene["aabb"][3:] - scene["aabb"][:3])
    dens = hip.query_density(_cu(pts.reshape(-1, 3))).cpu().numpy().reshape(200, -1) * step_size
    inside_range = (occ0[ids] >= dens.min(1) * 0.5) & (occ0[ids] <= dens.max(1) * 2.0)
    assert inside_range.mean() > 0.9
    # after the warm-up: N uniform + occupied cells
    a.occs.mul_(0.5)                                  # so that a re-evaluated cell visibly changes (max(0.475 old, new) != old)
    before = a.occs.clone()
    n_occ = int(a.binaries.sum())
    torch.manual_seed(12)
    a._update(step=256, occ_eval_fn=closure, occ_thre=1e-3)
    changed = (a.occs != before).cpu().numpy()
    N = cells // 4
    expect_uniform = cells * (1 - np.exp(-N / cells))                                    # distinct cells of N draws with replacement
    occupied_before = b.binaries.cpu().numpy().reshape(-1)
    if n_occ <= N:
        assert changed[occupied_before].all()                                            # every occupied cell re-evaluated
    assert 0.8 * expect_uniform < changed.sum() <= N + min(n_occ, N)
    assert not torch.isnan(a.occs).any()
    # NaN candidates leave the cell unchanged (the fork's roll-back, occ_grid.py:409-434)
    before = a.occs.clone()
    a._update(step=272, occ_eval_fn=lambda x: torch.full((x.shape[0], 1), float("nan"), device=x.device), occ_thre=1e-3)
    assert torch.equal(a.occs, before)


def test_pack_info_and_accumulate_kernels():
    """`pack_info` for ungrouped ray indices (pack.py:10-38) and the packed `accumulate_along_rays[_]` with its
    gradients (volrend.py:486-576) against the reference's definition evaluated by torch on the CPU."""
    from apnrf_amd import nerfacc as NA
    rng = np.random.default_rng(8)
    n_rays, n = 5000, 60000
    ri = rng.integers(0, n_rays, n)
    ri[ri % 11 == 0] = 7                                        # empty rays and one heavy ray
    cnts = np.bincount(ri, minlength=n_rays)
    want = np.stack([np.cumsum(cnts) - cnts, cnts], -1)
    np.testing.assert_array_equal(NA.pack_info(_cu(ri), n_rays).cpu().numpy(), want)
    np.testing.assert_array_equal(NA.pack_info(_cu(np.sort(ri)), n_rays).cpu().numpy(), want)
    np.testing.assert_array_equal(NA.exclusive_scan_counts(_cu(cnts)).cpu().numpy(), want[:, 0])
    big = rng.integers(0, 1000, 300001)                         # several scan tiles and a ragged tail
    st, tot = NA.exclusive_scan_counts(_cu(big), want_total=True)
    np.testing.assert_array_equal(st.cpu().numpy(), np.cumsum(big) - big)
    assert int(tot) == int(big.sum())
    for D in (1, 3, 29):
        w = rng.random(n).astype(np.float32)
        v = rng.normal(size=(n, D)).astype(np.float32)
        g = rng.normal(size=(n_rays, D)).astype(np.float32)
        cw, cv = torch.from_numpy(w).requires_grad_(True), torch.from_numpy(v).requires_grad_(True)
        ref = torch.zeros(n_rays, D).index_add_(0, torch.from_numpy(ri), cw[:, None] * cv)
        ref.backward(torch.from_numpy(g))
        hw, hv = _cu(w).requires_grad_(True), _cu(v).requires_grad_(True)
        out = NA.accumulate_along_rays(hw, hv, _cu(ri), n_rays)
        out.backward(_cu(g))
        np.testing.assert_allclose(out.detach().cpu().numpy(), ref.detach().numpy(), rtol=1e-4, atol=2e-5)
        np.testing.assert_allclose(hw.grad.cpu().numpy(), cw.grad.numpy(), rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(hv.grad.cpu().numpy(), cv.grad.numpy(), rtol=1e-5, atol=1e-6)
    acc = torch.ones(n_rays, 1, device=DEV)
    NA.accumulate_along_rays_(_cu(w), None, _cu(ri), acc)
    np.testing.assert_allclose(acc.cpu().numpy()[:, 0], 1.0 + np.bincount(ri, w, n_rays), rtol=1e-4, atol=1e-4)
    # batched branch (no ray indices): plain sums, as the reference
    wb, vb = torch.rand(4, 7, device=DEV), torch.rand(4, 7, 3, device=DEV)
    np.testing.assert_allclose(NA.accumulate_along_rays(wb, vb).cpu().numpy(), (wb[..., None] * vb).sum(-2).cpu().numpy(), rtol=1e-6)


# ------------------------------------------------------------------ a20: frequency-PE + biased-MLP field, pinned to the reference's own outputs
def _vanilla_from_golden(g, **kw):
    from apnrf_amd.mlp import VanillaNeRFRadianceField
    field = VanillaNeRFRadianceField(**kw)
    sd = {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd.")}
    field.load_state_dict(sd, strict=True)                        # same keys as the reference's state_dict
    return field.to(DEV)


def test_vanilla_field_matches_reference_golden(golden):
    """BASELINE config 1 end to end on the HIP path against tests/golden/vanilla.npz, which was captured from the
    reference's own `VanillaNeRFRadianceField` + `nerfacc.rendering` (CPU, fp32): 64x64 rays x 32 samples, forward values,
    the smooth-L1 loss and the gradient of EVERY parameter.  Tolerance 1e-5 (fp32 on both sides; only the summation order
    and the last ulp of sin / exp differ)."""
    import torch.nn.functional as F
    from apnrf_amd import nerfacc as NA
    g = golden("vanilla")
    field = _vanilla_from_golden(g, net_depth=2, net_width=64, skip_layer=None, net_depth_condition=1, net_width_condition=64)
    # the encoder alone (mlp.py:184-203), through the first layer's saved inputs: enc(x) rows are what the kernel feeds the MLP
    rays_o, rays_d = _cu(g["rays_o"]), _cu(g["rays_d"])
    edges = _cu(g["t_edges"])
    R, S = rays_o.shape[0], edges.shape[0] - 1
    t_starts, t_ends = edges[:-1].expand(R, S).contiguous(), edges[1:].expand(R, S).contiguous()

    def rgb_sigma_fn(ts, te, ri):
        pos = rays_o[:, None, :] + rays_d[:, None, :] * ((ts + te) / 2.0)[..., None]
        rgb, sigma = field(pos, rays_d)                           # [R,S,3] positions, [R,3] condition (mlp.py:154-160 broadcast)
        return rgb, sigma.squeeze(-1)

    colors, opac, depths, extras = NA.rendering(t_starts, t_ends, rgb_sigma_fn=rgb_sigma_fn, render_bkgd=torch.zeros(3, device=DEV))
    np.testing.assert_allclose(colors.detach().cpu().numpy(), g["colors"], atol=1e-5, rtol=1e-5)
    np.testing.assert_allclose(opac.detach().cpu().numpy(), g["opacities"], atol=1e-5, rtol=1e-5)
    np.testing.assert_allclose(depths.detach().cpu().numpy(), g["depths"], atol=1e-5, rtol=1e-5)
    loss = F.smooth_l1_loss(colors, _cu(g["target"]))
    np.testing.assert_allclose(loss.item(), float(g["loss"]), rtol=1e-5)
    loss.backward()
    for name, p in field.named_parameters():
        want = g["grad." + name]
        got = p.grad.cpu().numpy()
        assert got.shape == want.shape, name
        # 1e-5 of the gradient's scale per tensor (the loss is a mean over 12 288 values: the gradients themselves are ~1e-3)
        np.testing.assert_allclose(got, want, atol=1e-5 * max(float(np.abs(want).max()), 1e-30) + 1e-9, rtol=1e-4, err_msg=name)
        assert np.abs(want).max() > 0, name
    # inference path (no autograd) and query_density agree with the training forward
    with torch.no_grad():
        pos = rays_o[:, None, :] + rays_d[:, None, :] * ((t_starts + t_ends) / 2.0)[..., None]
        rgb2, sig2 = field(pos, rays_d)
        rgb3, sig3 = field(pos[:100], rays_d[:100, None, :].expand(100, S, 3))      # per-sample condition form
    np.testing.assert_array_equal(sig2.squeeze(-1).cpu().numpy(), extras["sigmas"].detach().cpu().numpy())
    np.testing.assert_array_equal(rgb3.cpu().numpy(), rgb2[:100].cpu().numpy())
    np.testing.assert_array_equal(field.query_density(pos).cpu().numpy(), sig2.cpu().numpy())


def test_vanilla_field_general_shapes_match_cpu_autograd():
    """The reference's default-like shape family (skip connection, deeper condition MLP, ragged sample count) against the
    same network evaluated by plain torch on the CPU (the definition in mlp.py:86-101, :153-165, :184-203, :238-243)."""
    from apnrf_amd.mlp import VanillaNeRFRadianceField
    torch.manual_seed(3)
    for kw in (dict(net_depth=6, net_width=64, skip_layer=4, net_depth_condition=2, net_width_condition=32),
               dict(net_depth=1, net_width=128, skip_layer=None, net_depth_condition=1, net_width_condition=64)):
        field = VanillaNeRFRadianceField(**kw)
        for p in field.parameters():
            if p.dim() == 1:
                torch.nn.init.uniform_(p, -0.1, 0.1)              # non-zero biases
        n = 1000 + 13
        x, d = torch.randn(n, 3) * 1.5, torch.nn.functional.normalize(torch.randn(n, 3), dim=-1)
        g_rgb, g_sig = torch.randn(n, 3), torch.randn(n, 1)
        cpu = {k: v.detach().clone().requires_grad_(True) for k, v in field.named_parameters()}

        def enc(v, deg):
            xb = (v[:, None, :] * torch.tensor([2.0 ** i for i in range(deg)])[:, None]).reshape(v.shape[0], -1)
            return torch.cat([v, torch.sin(torch.cat([xb, xb + 0.5 * np.pi], -1))], -1)

        def lin(h, name):
            return h @ cpu[name + ".weight"].t() + cpu[name + ".bias"]

        e = enc(x, 10)
        h = e
        for i in range(kw["net_depth"]):
            h = torch.relu(lin(h, f"mlp.base.hidden_layers.{i}"))
            if kw["skip_layer"] is not None and i % kw["skip_layer"] == 0 and i > 0:
                h = torch.cat([h, e], -1)
        raw_sigma = lin(h, "mlp.sigma_layer.output_layer")
        z = torch.cat([lin(h, "mlp.bottleneck_layer.output_layer"), enc(d, 4)], -1)
        for i in range(kw["net_depth_condition"]):
            z = torch.relu(lin(z, f"mlp.rgb_layer.hidden_layers.{i}"))
        r_rgb, r_sig = torch.sigmoid(lin(z, "mlp.rgb_layer.output_layer")), torch.relu(raw_sigma)
        torch.autograd.backward([r_rgb, r_sig], [g_rgb, g_sig])
        field = field.to(DEV)
        rgb, sig = field(x.to(DEV), d.to(DEV))
        torch.autograd.backward([rgb, sig], [g_rgb.to(DEV), g_sig.to(DEV)])
        np.testing.assert_allclose(rgb.detach().cpu().numpy(), r_rgb.detach().numpy(), atol=2e-6, rtol=1e-5)
        np.testing.assert_allclose(sig.detach().cpu().numpy(), r_sig.detach().numpy(), atol=2e-6, rtol=1e-5)
        for name, p in field.named_parameters():
            want = cpu[name].grad.numpy()
            np.testing.assert_allclose(p.grad.cpu().numpy(), want, atol=2e-5 * float(np.abs(want).max()) + 1e-9, rtol=1e-4, err_msg=name)


# ------------------------------------------------------------------ tcnn-faithful output rounding, remaining BASELINE configs, per-pose drivers
def test_field_tcnn_output_rounding_matches_oracle():
    """`tcnn_output_rounding=True` (mnf_field_config.output_fp16): every network output is rounded to fp16 where tiny-cuda-nn
    hands it over (ngp.py:181-200, :210-220) — against the oracle's precision="tcnn".  The two roundings agree except where
    an fp32 pre-image sits within the HIP/CPU summation-order difference of a rounding boundary, which moves the value by one
    fp16 ulp (2^-10 relative): hence the relative term."""
    sc = H.make_scene(neurons=128, layers=2, C=29, log2_hashmap_size=14, head_gain=4.0)
    hip, orc = H.hip_field(sc, tcnn_output_rounding=True), H.oracle_field(sc, precision="tcnn")
    hip32 = H.hip_field(sc)
    rng = np.random.default_rng(1)
    n = 4000 + 11
    a = sc["aabb"]
    pos = (a[:3] + rng.random((n, 3)) * (a[3:] - a[:3])).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32); d /= np.linalg.norm(d, axis=-1, keepdims=True)
    with torch.no_grad():
        rgb, sigma, sem = hip(_cu(pos), _cu(d))
        rgb32, sigma32, sem32 = hip32(_cu(pos), _cu(d))
    r_rgb, r_sigma, r_sem = orc(torch.from_numpy(pos), torch.from_numpy(d))
    np.testing.assert_allclose(sem.cpu().numpy(), r_sem.numpy(), atol=1e-3, rtol=2e-3)
    np.testing.assert_allclose(rgb.cpu().numpy(), r_rgb.numpy(), atol=1e-3)
    # density = exp(fp16 logit - 1): a boundary flip of the logit (3 of 4000 samples) is one fp16 ulp = 2^-8 at |logit| in [4,8),
    # i.e. 0.4 % of the density; everything else agrees to 2e-3 as in the fp32-output mode
    sg, rsg = sigma.cpu().numpy(), r_sigma.numpy()
    np.testing.assert_allclose(sg, rsg, rtol=9e-3, atol=1e-6)
    assert (np.abs(sg - rsg) > 2e-3 * np.abs(rsg) + 1e-6).mean() < 5e-3
    # the outputs ARE fp16 values, and the mode differs from the fp32-output default by at most half an fp16 ulp
    s_np = sem.cpu().numpy()
    np.testing.assert_array_equal(s_np, s_np.astype(np.float16).astype(np.float32))
    np.testing.assert_allclose(s_np, sem32.cpu().numpy(), rtol=2 ** -11 * 1.01, atol=1e-7)
    assert np.abs(s_np - sem32.cpu().numpy()).max() > 0


@pytest.mark.parametrize("prob", [False, True])
def test_render_baseline_config2_29_classes(prob):
    """BASELINE config 2 as written: scene 102344250, 256x256 view geometry, 4 x 64 base MLP, 29 classes."""
    from apnrf_amd import render as RD
    from oracle import render as R
    sc = H.make_scene("102344250", neurons=64, layers=4, C=29, seed=5)
    hip, orc, est = H.hip_field(sc), H.oracle_field(sc), H.hip_estimator(sc)
    o, d = H.view_rays(sc, 2, width=256, height=256, h=20, w=20)
    bk = torch.tensor([0.2, 0.7, 0.4])
    fn = R.render_prob_test if prob else R.render_test
    ref = fn(1024, orc, sc["occ"], sc["aabb"][None], o, d, render_bkgd=bk, **H.RENDER_KW)
    out = RD.render_views(hip, est, o.to(DEV), d.to(DEV), o.shape[0], 1024, render_bkgd=bk, probabilistic=prob, **H.RENDER_KW)
    assert out["sem"].shape == (400, 29) and ref["total_samples"] > 4000
    _check_render(out, ref, prob, max_tie_rays=2)


def test_render_from_pose_drivers_match_oracle(scene, fields):
    """`Dataset.render_image_from_pose` / `render_probablistic_image_from_pose` (habitat_to_data.py:304-549) called
    directly: float64 host arrays of the reference's shapes, values equal to the oracle's per-pose renders."""
    from apnrf_amd import render as RD
    from apnrf_amd.dataset import Dataset
    from oracle import render as R
    hip, orc = fields
    est = H.hip_estimator(scene)
    poses = scene["poses"][[2, 5]]
    W = Hh = 640
    focal = 0.5 * W / np.tan(np.pi / 4)
    scale = 0.025                                     # 16 x 16 rays per pose
    args = (hip, est, poses, W, Hh, focal, 0.1, 1e-3, scale, 0.004, 0.01, None, DEV)
    images, depths, accs, sems = RD.render_image_from_pose(*args)
    p_images, p_var, p_depths, p_dvar, p_accs, p_sems = RD.render_probablistic_image_from_pose(*args)
    for arr, shp in ((images, (2, 16, 16, 3)), (depths, (2, 16, 16)), (accs, (2, 16, 16)), (sems, (2, 16, 16, 29)),
                     (p_var, (2, 16, 16, 3)), (p_dvar, (2, 16, 16))):
        assert isinstance(arr, np.ndarray) and arr.dtype == np.float64 and arr.shape == shp, (arr.dtype, arr.shape, shp)
    assert Dataset.render_image_from_pose is not None         # the static methods of the reference class resolve to the same drivers
    idx = R.subsample_indices(W * Hh, 256)
    for k, p in enumerate(poses):
        o, d = R.generate_image_rays(R.pose_to_c2w(p), W, Hh, focal, idx)
        ref = R.render_prob_test(1024, orc, scene["occ"], scene["aabb"][None], o, d, render_bkgd=torch.zeros(3), **H.RENDER_KW)
        np.testing.assert_allclose(p_images[k].reshape(-1, 3), ref["rgb"].numpy(), atol=1e-3)
        np.testing.assert_allclose(p_depths[k].reshape(-1), ref["depth"].numpy()[:, 0], atol=1e-3, rtol=1e-3)
        np.testing.assert_allclose(p_accs[k].reshape(-1), ref["acc"].numpy()[:, 0], atol=1e-3)
        np.testing.assert_allclose(p_sems[k].reshape(-1, 29), ref["sem"].numpy(), atol=1e-3)
        np.testing.assert_allclose(p_var[k].reshape(-1, 3), ref["rgb_var"].numpy(), atol=1e-3)
        np.testing.assert_allclose(p_dvar[k].reshape(-1), ref["depth_var"].numpy()[:, 0], atol=2e-3, rtol=2e-3)
        # deterministic driver == probabilistic driver on the shared outputs
        np.testing.assert_allclose(images[k], p_images[k], atol=1e-6)
        np.testing.assert_allclose(sems[k], p_sems[k], atol=1e-5)


def test_train_step_baseline_config5():
    """BASELINE config 5: scene 102344280, 8192-ray train batches.  (1) a 256-ray subset of the batch through the
    differentiable train render + loss + backward against the oracle under CPU autograd; (2) the full 8192-ray step through
    `render.train_step`: size-independent properties (finite loss, every ray's opacity in [0,1], sample accounting, gradients
    only on touched table entries, parameters move, a second identical call after the update lowers the loss on the same batch)."""
    import torch.nn.functional as F
    from apnrf_amd import render as RD
    from apnrf_amd.optim import FusedAdam
    from oracle import render as R
    sc = H.make_scene("102344280", log2_hashmap_size=16, seed=2, n_poses=8)
    hip, orc, est = H.hip_field(sc), H.oracle_field(sc, requires_grad=True), H.hip_estimator(sc)
    c2w = RD.pose_to_c2w(sc["poses"][1]).astype(np.float32)[None]
    K = np.array([[320.0, 0, 320], [0, 320.0, 320], [0, 0, 1.0]])
    rng = np.random.default_rng(4)
    idx = rng.integers(0, 640 * 640, 8192)
    rays = RD.generate_image_rays(torch.from_numpy(c2w), 640, 640, K, DEV, idx)
    pix = torch.from_numpy(rng.random((8192, 3)).astype(np.float32)).to(DEV)
    dep = torch.from_numpy(rng.uniform(0.5, 4.0, 8192).astype(np.float32)).to(DEV)
    lab = torch.from_numpy(rng.integers(0, 29, 8192)).to(DEV)
    bk = torch.tensor([0.1, 0.6, 0.3], device=DEV)

    def loss_fn(rgb, depth, sem, pix_, dep_, lab_):
        return F.smooth_l1_loss(rgb, pix_) * 10 + F.smooth_l1_loss(depth, dep_.unsqueeze(1)) / 5 + F.cross_entropy(sem, lab_) / 2

    # (1) 256-ray subset vs the oracle
    sub = torch.arange(0, 8192, 32, device=DEV)
    hip.eval()                                                     # no stratified jitter; gradients still flow
    o_s, d_s = rays.origins[sub].contiguous(), rays.viewdirs[sub].contiguous()
    rgb, acc, depth, sem, n = RD.render_image_with_occgrid_with_depth_guide(hip, est, RD.Rays(o_s, d_s), render_bkgd=bk, **H.RENDER_KW)
    loss = loss_fn(rgb, depth, sem, pix[sub], dep[sub], lab[sub])
    hip.zero_grad(); loss.backward()
    ref = R.render_train(orc, sc["occ"], sc["aabb"][None], float(est.occs.mean().item()), o_s.cpu(), d_s.cpu(), torch.full((256,), 0.1),
                         render_bkgd=bk.cpu(), render_step_size=1e-3, cone_angle=0.004, alpha_thre=0.01)
    r_loss = loss_fn(ref[0], ref[2], ref[3], pix[sub].cpu(), dep[sub].cpu(), lab[sub].cpu())
    r_loss.backward()
    assert n > 5000 and abs(n - ref[4]) <= max(3, 0.002 * ref[4])
    np.testing.assert_allclose(loss.item(), r_loss.item(), rtol=2e-4)
    n_mlp = sum(o_ * i_ for o_, i_ in orc.shapes["base"])
    _grad_close(hip.mlp_base.params.grad[:n_mlp], orc.p_base.grad[:n_mlp], "base mlp", rel=3e-2, cos=0.999)
    _grad_close(hip.mlp_base.params.grad[n_mlp:], orc.p_base.grad[n_mlp:], "hash table", rel=3e-2, cos=0.999)
    _grad_close(hip.mlp_head.params.grad, orc.p_head.grad, "rgb head", rel=3e-2, cos=0.999)
    _grad_close(hip.mlp_sem.params.grad, orc.p_sem.grad, "sem head", rel=3e-2, cos=0.999)
    # (2) the full batch through train_step
    hip.zero_grad()
    opt = FusedAdam(hip.parameters(), lr=1e-3, eps=1e-15)
    before = [p.detach().clone() for p in hip.parameters()]
    out1 = RD.train_step(hip, est, opt, rays, pix, dep, lab, bk, step=1, **H.RENDER_KW)      # step 1: no occupancy refresh
    assert not out1["skipped"] and np.isfinite(float(out1["loss"])) and out1["n_rendering_samples"] > 8192 * 10
    assert est.last_sampling["n_marched"] >= out1["n_rendering_samples"]
    g_tab = hip.mlp_base.params.grad[n_mlp:]
    assert 0 < int((g_tab != 0).sum()) < g_tab.numel() and torch.isfinite(g_tab).all()
    assert all((p.detach() != b).any() for p, b in zip(hip.parameters(), before) if p.numel())
    hip.eval()
    with torch.no_grad():
        rgb_a, acc_a, _, _, _ = RD.render_image_with_occgrid_with_depth_guide(hip, est, rays, render_bkgd=bk, **H.RENDER_KW)
    assert float(acc_a.min()) >= 0 and float(acc_a.max()) <= 1 + 1e-5 and rgb_a.shape == (8192, 3)
    losses = [float(out1["loss"])]
    for s_ in range(2, 8):
        losses.append(float(RD.train_step(hip, est, opt, rays, pix, dep, lab, bk, step=s_, **H.RENDER_KW)["loss"]))
    assert losses[-1] < losses[0], losses


def test_render_views_sharded_single_rank_and_row_tiles(scene, fields):
    """`distributed.render_views_sharded` without a process group is `render_views`; its row-tile mode (fewer views than
    ranks) renders each tile as a reference call of its own, which is what calling `render_views` on the tile gives."""
    from apnrf_amd import distributed as DD
    from apnrf_amd import render as RD
    hip, _ = fields
    est = H.hip_estimator(scene)
    o, d = H.view_rays(scene, 1, h=16, w=16)
    o, d = o.to(DEV), d.to(DEV)
    bk = torch.zeros(3)
    a = DD.render_views_sharded(hip, est, o, d, 256, probabilistic=True, max_samples=1024, render_bkgd=bk, **H.RENDER_KW)
    b = RD.render_views(hip, est, o, d, 256, 1024, render_bkgd=bk, probabilistic=True, **H.RENDER_KW)
    for k in ("rgb", "acc", "depth", "sem", "rgb_var", "depth_var", "total"):
        assert torch.equal(a[k], b[k]), k
    sh = DD.shard_rays(1, 256, 4, 2)                       # one view over four ranks: rank 2 renders rays 128..191 as its own call
    assert (sh["tiles_per_view"], sh["lo"], sh["hi"], sh["unit_rays"]) == (4, 2, 3, 64)
    tile = RD.render_views(hip, est, o[128:192].contiguous(), d[128:192].contiguous(), 64, 1024, render_bkgd=bk, **H.RENDER_KW)
    np.testing.assert_allclose(tile["rgb"].cpu().numpy(), b["rgb"][128:192].cpu().numpy(), atol=2e-3)   # schedule differs, values agree


def test_slot_compositing_equals_general_path():
    """The render epilogue composites tiles whose ray slots are aligned runs of 4 / 8 / 16 columns with DPP butterflies and a quad
    reduce-scatter (csrc/composite_dev.h), every other tile with segmented scans.  Same arithmetic per sample, different summation
    order: outputs agree to fp32 rounding.  The two knobs that expose the comparison (MNF_MIN_SAMPLES: a diagnostic schedule, not
    the reference's, that makes slots of 8 and 16 occur in every early round; MNF_COMPOSITE_GENERAL=1: every tile down the general
    path) exist only in the -DMNF_DIAG build of the library, so the comparison runs in a child process that loads
    libmi355nerf_diag.so (tests/diag_compositing.py); the product library in this process ignores both variables."""
    import os
    import subprocess
    import sys
    from apnrf_amd import build as B
    here = os.path.dirname(os.path.abspath(__file__))
    assert os.path.exists(B.LIB_DIAG), "libmi355nerf_diag.so missing: run `python __graft_entry__.py build`"
    env = dict(os.environ, MNF_LIB_PATH=B.LIB_DIAG)
    r = subprocess.run([sys.executable, os.path.join(here, "diag_compositing.py")], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "DIAG_COMPOSITING_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_product_library_ignores_diagnostic_environment(scene, fields, monkeypatch):
    """no environment variable changes what the shipped library computes: the same render with the
    results-changing knobs of the diag build set is bit-identical (they are compiled out, csrc/common.h diag_env)."""
    from apnrf_amd import render as RD
    hip, _ = fields
    est = H.hip_estimator(scene)
    o, d = H.view_rays(scene, 2, h=32, w=32)
    o, d = o.to(DEV), d.to(DEV)
    bk = torch.zeros(3)
    a = RD.render_views(hip, est, o, d, o.shape[0], 1024, render_bkgd=bk, probabilistic=True, **H.RENDER_KW)
    for k, v in (("MNF_MIN_SAMPLES", "16"), ("MNF_COMPOSITE_GENERAL", "1"), ("MNF_FIELD_SPLIT", "1"), ("MNF_FIELD_PIPE", "4,20,4,768")):
        monkeypatch.setenv(k, v)
    b = RD.render_views(hip, est, o, d, o.shape[0], 1024, render_bkgd=bk, probabilistic=True, **H.RENDER_KW)
    for k in ("rgb", "acc", "depth", "sem", "rgb_var", "depth_var", "total"):
        assert torch.equal(a[k], b[k]), k


def test_checkpoint_fixture_hand_computed_density():
    """f2: a checkpoint in the reference's key layout written by tests/golden/make_checkpoint_fixture.py (which imports neither
    the product nor the oracle) loads through `dataset.load_checkpoint`, and the density it yields is the HAND-COMPUTED one
    (exp(-0.5) inside the box, 0 outside): pins the in-vector layout the library assumes for tcnn's flat `params`
    (network before table, [out][in] row-major, density = output row 0) against an independent statement of it."""
    import os
    from apnrf_amd import dataset as DS
    from apnrf_amd.nerfacc import OccGridEstimator
    from apnrf_amd.ngp import NGPRadianceField
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "checkpoint_fixture.pth")
    ck = torch.load(path, map_location="cpu")
    assert {"occ_grid", "model", "optimizer_state_dict"} <= set(ck)
    cfg = ck["expected"]["config"]
    field = NGPRadianceField(aabb=ck["model"]["aabb"], **cfg).to(DEV)
    est = OccGridEstimator(ck["model"]["aabb"], resolution=list(ck["occ_grid"].shape[1:])).to(DEV)
    opt = torch.optim.Adam(field.parameters(), lr=1e-3, eps=1e-15)
    DS.load_checkpoint(path, est, field, opt, map_location=DEV)
    assert torch.equal(est.binaries.cpu(), ck["occ_grid"]) and int(est.binaries.sum()) == 6
    dens = field.query_density(ck["expected"]["points"].to(DEV)).cpu()[:, 0]
    np.testing.assert_allclose(dens.numpy(), ck["expected"]["density"].numpy(), rtol=2e-3, atol=1e-7)     # 1/(j+1) is rounded to fp16
    with torch.no_grad():
        rgb, sigma, sem = field(ck["expected"]["points"].to(DEV), torch.tensor([[0.0, 0.0, 1.0]] * 3, device=DEV))
    np.testing.assert_allclose(sigma.cpu().numpy()[:, 0], ck["expected"]["density"].numpy(), rtol=2e-3, atol=1e-7)
    assert sem.shape == (3, 5) and bool(torch.isfinite(rgb).all())
    # the oracle's reading of the same vectors agrees (it is the other consumer of this layout)
    sc = dict(aabb=ck["model"]["aabb"].numpy(), params={k.split(".")[0]: v.numpy() for k, v in ck["model"].items() if k.endswith("params") and v.numel()},
              C=cfg["num_semantic_classes"], **{k: cfg[k] for k in ("neurons", "layers", "log2_hashmap_size")})
    o_d = H.oracle_field(sc).query_density(ck["expected"]["points"])[:, 0]
    np.testing.assert_allclose(o_d.numpy(), ck["expected"]["density"].numpy(), rtol=2e-3, atol=1e-7)


def test_ray_major_density_prepass_gives_same_samples(scene):
    """`OccGridEstimator.sampling` with the ray-major density pass (mnf_field_density_rays: the part of a ray behind
    T < early_stop_eps / 2 is never evaluated) returns exactly the sample set of the reference formulation (sigma_fn on
    every marched sample, then render_visibility_from_density), on a field dense enough that most rays saturate."""
    from apnrf_amd.ngp import RaySigmaFn
    sc = H.make_scene(log2_hashmap_size=15, seed=3)
    sc["params"] = H.S.make_field_params(seed=3, log2_hashmap_size=15, density_gain=24.0)      # opaque quickly: long invisible tails
    hip, est = H.hip_field(sc), H.hip_estimator(sc)
    o, d = H.view_rays(sc, 2, h=40, w=40)
    o, d = o.to(DEV), d.to(DEV)
    fast = RaySigmaFn(hip, o, d)
    plain = lambda ts, te, ri: hip.forward_samples(o, d, ri, ts, te, density_only=True)[0]
    for eps, thre in ((1e-4, 0.01), (1e-2, 0.0), (1e-4, 0.0)):
        a = est.sampling(o, d, sigma_fn=fast, near_plane=0.1, render_step_size=1e-3, cone_angle=0.004, alpha_thre=thre, early_stop_eps=eps)
        b = est.sampling(o, d, sigma_fn=plain, near_plane=0.1, render_step_size=1e-3, cone_angle=0.004, alpha_thre=thre, early_stop_eps=eps)
        for x, y in zip(a, b):
            np.testing.assert_array_equal(x.cpu().numpy(), y.cpu().numpy())
        assert a[0].shape[0] > 1000
    # the pass really skips: densities behind the cut are the zeros of the initial fill, and there are many of them
    n = o.shape[0]
    near, far = torch.full((n,), 0.1, device=DEV), torch.full((n,), 1e10, device=DEV)
    ri, ts, te, info = est._sample_single_pass(o, d, near, far, 1e-3, 0.004)
    s_fast, s_all = fast.ray_major(ts, te, ri, info, 1e-4), plain(ts, te, ri)
    skipped = (s_fast == 0) & (s_all > 0)
    assert 0.03 < float(skipped.float().mean()) < 0.99
    np.testing.assert_array_equal(s_fast[~skipped].cpu().numpy(), s_all[~skipped].cpu().numpy())
    # early_stop_eps = 0 disables the cut
    np.testing.assert_array_equal(fast.ray_major(ts, te, ri, info, 0.0).cpu().numpy(), s_all.cpu().numpy())


# ------------------------------------------------------------------ g1: bf16 matrix-core operands (BASELINE config 5)
def test_bf16_field_forward_backward_and_render_match_oracle():
    """`mfma_bf16=True`: the same kernels with v_mfma_f32_32x32x16_bf16 (weights, MLP inputs, hidden activations and the
    backward's activation gradients in bfloat16; hash table fp16; fp32 accumulate) against the oracle's precision="bf16".
    bf16 keeps 8 significand bits (fp16: 11), so one rounding flip moves a value by 2^-8 relative: tolerances are 8x the
    fp16 mode's (outputs 8e-3 abs + 1.6e-2 rel on logits, density 1.6e-2 rel, gradients 6e-2 relative L2, cosine > 0.998)."""
    from apnrf_amd import render as RD
    from oracle import render as R
    sc = H.make_scene(neurons=128, layers=2, C=29, log2_hashmap_size=14, head_gain=2.0)
    hip, orc = H.hip_field(sc, mfma_bf16=True).train(), H.oracle_field(sc, precision="bf16", requires_grad=True)
    rng = np.random.default_rng(7)
    n = 3000 + 21
    a = sc["aabb"]
    pos = (rng.random((n, 3)) * (a[3:] - a[:3]) * 0.98 + a[:3] + 0.01 * (a[3:] - a[:3])).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32); d /= np.linalg.norm(d, axis=-1, keepdims=True)
    g_rgb = (rng.normal(size=(n, 3)) * 1e-3).astype(np.float32)
    g_sig = (rng.normal(size=(n, 1)) * 1e-5).astype(np.float32)
    g_sem = (rng.normal(size=(n, 29)) * 1e-3).astype(np.float32)
    rgb, sigma, sem = hip(_cu(pos), _cu(d))
    torch.autograd.backward([rgb, sigma, sem], [_cu(g_rgb), _cu(g_sig), _cu(g_sem)])
    r_rgb, r_sigma, r_sem = orc(torch.from_numpy(pos), torch.from_numpy(d))
    torch.autograd.backward([r_rgb, r_sigma, r_sem], [torch.from_numpy(g_rgb), torch.from_numpy(g_sig), torch.from_numpy(g_sem)])
    np.testing.assert_allclose(rgb.detach().cpu().numpy(), r_rgb.detach().numpy(), atol=8e-3)
    np.testing.assert_allclose(sem.detach().cpu().numpy(), r_sem.detach().numpy(), atol=8e-3, rtol=1.6e-2)
    np.testing.assert_allclose(sigma.detach().cpu().numpy(), r_sigma.detach().numpy(), rtol=1.6e-2, atol=1e-6)
    n_mlp = sum(o * i for o, i in orc.shapes["base"])
    _grad_close(hip.mlp_base.params.grad[:n_mlp], orc.p_base.grad[:n_mlp], "base mlp", rel=6e-2, cos=0.998)
    _grad_close(hip.mlp_base.params.grad[n_mlp:], orc.p_base.grad[n_mlp:], "hash table", rel=6e-2, cos=0.998)
    _grad_close(hip.mlp_head.params.grad, orc.p_head.grad, "rgb head", rel=6e-2, cos=0.998)
    _grad_close(hip.mlp_sem.params.grad, orc.p_sem.grad, "sem head", rel=6e-2, cos=0.998)
    # the bf16 mode is really a different arithmetic from the fp16 one (and both are deterministic)
    hip16 = H.hip_field(sc)
    with torch.no_grad():
        s16 = hip16(_cu(pos), _cu(d))[2]
        s_again = hip.eval()(_cu(pos), _cu(d))[2]
    assert (s16 - sem.detach()).abs().max() > 1e-4 and torch.equal(s_again, sem.detach())
    # fused test-time renderer in bf16 against the oracle's bf16 field
    est = H.hip_estimator(sc)
    o, dd = H.view_rays(sc, 1, h=24, w=24)
    orc_ng = H.oracle_field(sc, precision="bf16")
    ref = R.render_test(1024, orc_ng, sc["occ"], sc["aabb"][None], o, dd, render_bkgd=torch.zeros(3), **H.RENDER_KW)
    out = RD.render_views(hip, est, o.to(DEV), dd.to(DEV), 576, 1024, render_bkgd=torch.zeros(3), **H.RENDER_KW)
    for k, tol in (("rgb", 8e-3), ("acc", 8e-3), ("sem", 8e-3)):
        bad = (np.abs(out[k].cpu().numpy() - ref[k].numpy()) > tol).reshape(576, -1).any(1)
        assert bad.sum() <= 3, (k, int(bad.sum()))              # a few alpha-threshold ties at bf16 resolution
    mse = float(((out["rgb"].cpu() - ref["rgb"]) ** 2).mean())
    assert 10.0 * np.log10(1.0 / max(mse, 1e-20)) > 40.0


def test_bf16_train_step_config5_shape():
    """BASELINE config 5 as written (bf16 MFMA path, 8192-ray train batches, fused backward): `train_step` on scene 102344280 in
    bf16 mode lowers the loss on a fixed batch and produces finite gradients on touched entries only."""
    from apnrf_amd import render as RD
    from apnrf_amd.optim import FusedAdam
    sc = H.make_scene("102344280", log2_hashmap_size=16, seed=2, n_poses=8)
    hip, est = H.hip_field(sc, mfma_bf16=True), H.hip_estimator(sc)
    c2w = RD.pose_to_c2w(sc["poses"][1]).astype(np.float32)[None]
    K = np.array([[320.0, 0, 320], [0, 320.0, 320], [0, 0, 1.0]])
    rng = np.random.default_rng(4)
    rays = RD.generate_image_rays(torch.from_numpy(c2w), 640, 640, K, DEV, rng.integers(0, 640 * 640, 8192))
    pix = torch.from_numpy(rng.random((8192, 3)).astype(np.float32)).to(DEV)
    dep = torch.from_numpy(rng.uniform(0.5, 4.0, 8192).astype(np.float32)).to(DEV)
    lab = torch.from_numpy(rng.integers(0, 29, 8192)).to(DEV)
    opt = FusedAdam(hip.parameters(), lr=1e-3, eps=1e-15)
    losses = []
    for s_ in range(1, 9):
        out = RD.train_step(hip, est, opt, rays, pix, dep, lab, torch.zeros(3, device=DEV), step=s_, **H.RENDER_KW)
        assert not out["skipped"] and out["n_rendering_samples"] > 8192 * 10
        losses.append(float(out["loss"]))
    assert np.isfinite(losses).all() and losses[-1] < losses[0], losses
    n_mlp = 128 * 64 + 128 * 128 + 16 * 128
    g_tab = hip.mlp_base.params.grad[n_mlp:]
    assert 0 < int((g_tab != 0).sum()) < g_tab.numel() and bool(torch.isfinite(g_tab).all())


# ------------------------------------------------------------------ single-call C entry points (mnf_train_step, mnf_score_poses)
def test_fused_train_step_equals_call_by_call_path():
    """`mnf_train_step` (render + loss + backward in one C call) against the differentiable Python surface
    (`render_image_with_occgrid_with_depth_guide` + torch's smooth_l1 / cross_entropy + autograd) on the same batch without
    stratified jitter: identical sample counts, losses to fp32 rounding, gradients to the atomics' summation-order noise."""
    import torch.nn.functional as F
    from apnrf_amd import render as RD
    sc = H.make_scene(log2_hashmap_size=15, seed=6)
    est = H.hip_estimator(sc)
    o, d = H.view_rays(sc, 4, h=24, w=24)
    rays = RD.Rays(o.to(DEV), d.to(DEV))
    rng = np.random.default_rng(3)
    pix = torch.from_numpy(rng.random((576, 3)).astype(np.float32)).to(DEV)
    dep = torch.from_numpy(rng.uniform(0.5, 4.0, 576).astype(np.float32)).to(DEV)
    lab = torch.from_numpy(rng.integers(0, 29, 576)).to(DEV)
    dep[:10] += 3.0                                            # some |depth error| > 1: the linear branch of smooth_l1
    bk = torch.tensor([0.5, 0.2, 0.9], device=DEV)
    a, b = H.hip_field(sc), H.hip_field(sc)
    a.eval()
    rgb, acc, depth, sem, n = RD.render_image_with_occgrid_with_depth_guide(a, est, rays, render_bkgd=bk, **H.RENDER_KW)
    l_rgb, l_dep, l_sem = F.smooth_l1_loss(rgb, pix), F.smooth_l1_loss(depth, dep.unsqueeze(1)), F.cross_entropy(sem, lab)
    (l_rgb * 10 + l_dep / 5 + l_sem / 2).backward()
    out = RD.fused_forward_backward(b, est, rays, pix, dep, lab, bk, stratified=False, **H.RENDER_KW)
    assert out is not None and out["n_rendering_samples"] == n and out["n_marched"] >= n
    for got, want in ((out["loss_rgb"], l_rgb), (out["loss_dep"], l_dep), (out["loss_sem"], l_sem), (out["loss"], l_rgb * 10 + l_dep / 5 + l_sem / 2)):
        np.testing.assert_allclose(float(got.detach()), float(want.detach()), rtol=2e-5)
    for pa, pb, name in zip(a.parameters(), b.parameters(), ("dir", "base", "head", "sem")):
        if pa.numel():
            _grad_close(pb.grad, pa.grad.cpu(), name, rel=2e-3, cos=0.99999)
    # and through train_step: both routes take the same optimizer step from the same state (no jitter in eval-less mode is not
    # available there, so only the plumbing is checked: a finite loss, parameters moved, counts reported)
    from apnrf_amd.optim import FusedAdam
    opt = FusedAdam(b.parameters(), lr=1e-3, eps=1e-15)
    before = b.mlp_head.params.detach().clone()
    r = RD.train_step(b, est, opt, rays, pix, dep, lab, bk, step=3, **H.RENDER_KW)
    assert not r["skipped"] and np.isfinite(float(r["loss"])) and r["n_rendering_samples"] > 0 and (b.mlp_head.params.detach() != before).any()
    r2 = RD.train_step(b, est, opt, rays, pix, dep, lab, bk, step=5, fused=False, **H.RENDER_KW)
    assert not r2["skipped"] and abs(float(r2["loss"]) - float(r["loss"])) < 0.5


def test_score_poses_single_call_equals_python_route(scene):
    """`mnf_score_poses` (poses -> rays -> probabilistic renders of both members -> terms, one C call) gives the terms of
    `render.score_views` (the same steps call by call) bit for bit."""
    import ctypes
    from apnrf_amd import _lib as L
    from apnrf_amd import render as RD
    sc2 = dict(scene); sc2["params"] = H.S.make_field_params(seed=1)
    fields = [H.hip_field(scene), H.hip_field(sc2)]
    ests = [H.hip_estimator(scene), H.hip_estimator(scene)]
    poses = scene["poses"][[1, 4, 6]]
    terms_py, score_py = RD.score_views(fields, ests, poses, 640, 640, 320.0, 0.1, 1e-3, 0.025, 0.004, 0.01, DEV)
    terms_c, score_c = RD.score_poses(fields, ests, poses, 640, 640, 320.0, 0.1, 1e-3, 0.025, 0.004, 0.01, DEV)
    assert torch.equal(terms_py, terms_c) and float(score_py) == float(score_c)
