"""Round 5: the HIP path against goldens recorded from the RUNNING reference glue (tests/golden/glue.npz, glue_ngp.npz,
scorer.npz; generator tests/golden/make_golden.py, substitutions tests/golden/ref_shim.py).

  * analytic field (glue.npz): the product's `OccGridEstimator.sampling` and compositing kernels are fed the same closed-form
    field the reference's functions were driven with -> sample sets bit-exact, composited values and gradients 1e-5.
  * NGP field (glue_ngp.npz, scorer.npz): the reference's glue drove the oracle's NGP field with seeded parameters; the
    product's fused renderers, per-pose drivers and scorer run the same parameters -> north star 1e-3 abs."""
import os
import sys

import numpy as np
import pytest
import torch

import helpers as H
from test_glue_golden_cpu import AnalyticField, multilevel_of, pipeline_loss, scene_of, scorer_stacks

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _cu(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    return (t if dtype is None else t.to(dtype)).to(DEV)


def _hip_estimator(sc):
    from apnrf_amd.nerfacc import OccGridEstimator
    est = OccGridEstimator(torch.from_numpy(sc["aabb"]), resolution=sc["res"], levels=1)
    est.binaries = torch.from_numpy(sc["occ"])
    est.occs = torch.from_numpy(sc["occs"])
    return est.to(DEV).eval()


def _hip_ngp_field(g, seed):
    from apnrf_amd import synthetic as S
    lh = int(g["log2_hashmap_size"])
    params = S.make_field_params(128, 2, 29, seed=int(seed), log2_hashmap_size=lh)
    scene = dict(aabb=g["aabb"], neurons=128, layers=2, C=29, log2_hashmap_size=lh, params=params)
    return H.hip_field(scene), params


# ------------------------------------------------------------------ analytic field: sampling (a4) and compositing (a10-a13)
def test_sampling_equals_reference_estimator_golden(golden):
    """occ_grid.py:80-238 as the reference ran it: marched set, density pre-pass with the analytic sigma_fn, visibility filter."""
    g = golden("glue")
    sc = scene_of(g)
    kw = sc["kw"]
    est = _hip_estimator(sc)
    f = AnalyticField(29, seed=int(g["field_seed"]), device=DEV).eval()
    o, d = _cu(g["rays_o"]), _cu(g["rays_d"])

    def sigma_fn(ts, te, ri):
        pos = o[ri] + d[ri] * (ts + te)[:, None] / 2.0
        return f.query_density(pos).squeeze(-1)

    ri, ts, te = est.sampling(o, d, sigma_fn=sigma_fn, stratified=False, **kw)
    assert ri.dtype == torch.int64
    np.testing.assert_array_equal(ri.cpu().numpy(), g["samp_ri"])
    np.testing.assert_array_equal(ts.cpu().numpy(), g["samp_ts"])
    np.testing.assert_array_equal(te.cpu().numpy(), g["samp_te"])
    ri, ts, te = est.sampling(o, d, sigma_fn=None, **kw)
    assert len(ri) == int(g["samp_all_n"]) and int(ri.sum().item()) == int(g["samp_all_ri_sum"])
    np.testing.assert_array_equal(torch.bincount(ri, minlength=len(o)).cpu().numpy(), g["samp_all_cnt"])
    assert float(ts.double().sum().item()) == pytest.approx(float(g["samp_all_ts_sum"]), rel=1e-12)
    # the reference's stratified draw replayed through t_min (near = clamp(near, min=t_min): occ_grid.py:152-159)
    near_st = torch.full((len(o),), kw["near_plane"], device=DEV) + _cu(g["samp_st_draw"]) * kw["render_step_size"]
    ri, ts, te = est.sampling(o, d, sigma_fn=sigma_fn, t_min=near_st, stratified=False, **kw)
    np.testing.assert_array_equal(ri.cpu().numpy(), g["samp_st_ri"])
    np.testing.assert_array_equal(ts.cpu().numpy(), g["samp_st_ts"])
    np.testing.assert_array_equal(te.cpu().numpy(), g["samp_st_te"])


def test_compositing_kernels_equal_reference_sem_rendering_golden(golden):
    """utils.py:362-461 + the loss of pipeline.py:506-511 + autograd, on the reference's own sample set and per-sample field outputs."""
    from apnrf_amd import nerfacc as NA
    g = golden("glue")
    f = AnalyticField(29, seed=int(g["field_seed"])).train()
    o, d = torch.from_numpy(g["rays2_o"]), torch.from_numpy(g["rays2_d"])
    ri, ts, te = (torch.from_numpy(g[k]) for k in ("semr_ri", "semr_ts", "semr_te"))
    with torch.no_grad():
        rgbs, dens, sems = f(o[ri] + d[ri] * (ts + te)[:, None] / 2.0, d[ri])
    np.testing.assert_array_equal(dens.squeeze(-1).numpy(), g["semr_sigmas"])
    n_rays = len(o)
    rgbs, sig, sems = (t.to(DEV).requires_grad_(True) for t in (rgbs, dens.squeeze(-1), sems))
    packed = NA.pack_info_grouped(ri.to(DEV), n_rays)
    np.testing.assert_array_equal(packed[:, 1].cpu().numpy(), np.bincount(g["semr_ri"], minlength=n_rays))
    colors, opac, depths, sem, w, tr, al = NA._CompositeTrain.apply(packed[:, 0].contiguous(), packed[:, 1].contiguous(), ts.to(DEV), te.to(DEV),
                                                                    sig, rgbs, sems, _cu(g["bkgd"]))
    tol = dict(atol=2e-6, rtol=2e-6)
    for got, k in ((colors, "colors"), (opac, "opac"), (depths, "depths"), (sem, "sem"), (w, "weights"), (tr, "trans"), (al, "alphas")):
        np.testing.assert_allclose(got.detach().cpu().numpy(), g["semr_" + k], err_msg=k, atol=1e-5 if k == "sem" else 2e-6, rtol=2e-6)
    loss = pipeline_loss(colors, depths, sem, _cu(g["pix"]), _cu(g["dep"]), _cu(g["lab"]))
    loss.backward()
    np.testing.assert_allclose(loss.item(), float(g["semr_loss"]), rtol=2e-6)
    np.testing.assert_allclose(rgbs.grad.cpu().numpy(), g["semr_g_rgbs"], atol=1e-8, rtol=1e-4)
    np.testing.assert_allclose(sig.grad.cpu().numpy(), g["semr_g_sigmas"], atol=6e-9, rtol=1e-4)      # suffix sums of terms of both signs: 8e-5 of the largest entry
    np.testing.assert_allclose(sems.grad.cpu().numpy()[::4], g["semr_g_sems_every4"], atol=1e-8, rtol=1e-4)
    # no samples at all (utils.py:403-407): background colour, zeros
    z = torch.zeros(4, dtype=torch.int64, device=DEV)
    e = torch.empty(0, device=DEV)
    c0, a0, d0, s0, *_ = NA._CompositeTrain.apply(z, z, e, e, e, torch.empty(0, 3, device=DEV), torch.empty(0, 29, device=DEV), _cu(g["bkgd"]))
    for got, k in ((c0, "colors"), (a0, "opac"), (d0, "depths"), (s0, "sem")):
        np.testing.assert_allclose(got.cpu().numpy(), g["semr0_" + k], atol=1e-7, err_msg=k)


def test_train_render_golden_through_compositing_with_recorded_jitter(golden):
    """utils.py:63-219 in train mode as the reference ran it (its stratified draw replayed): sampling -> analytic field -> compositing
    kernels -> pipeline loss -> backward, against the reference's outputs, loss and per-sample gradients."""
    from apnrf_amd import nerfacc as NA
    g = golden("glue")
    sc = scene_of(g)
    kw = sc["kw"]
    est = _hip_estimator(sc)
    f = AnalyticField(29, seed=int(g["field_seed"]), device=DEV).train()
    o, d = _cu(g["rays2_o"]), _cu(g["rays2_d"])

    def sigma_fn(ts, te, ri):
        return f.query_density(o[ri] + d[ri] * (ts + te)[:, None] / 2.0).squeeze(-1)

    near_st = torch.full((len(o),), kw["near_plane"], device=DEV) + _cu(g["trt_draw"]) * kw["render_step_size"]
    ri, ts, te = est.sampling(o, d, sigma_fn=sigma_fn, t_min=near_st, stratified=False, **kw)
    assert len(ri) == int(g["trt_n"])
    with torch.no_grad():
        rgbs, dens, sems = f(o[ri] + d[ri] * (ts + te)[:, None] / 2.0, d[ri])
    rgbs, sig, sems = (t.detach().requires_grad_(True) for t in (rgbs, dens.squeeze(-1), sems))
    packed = NA.pack_info_grouped(ri, len(o))
    rgb, acc, depth, sem, *_ = NA._CompositeTrain.apply(packed[:, 0].contiguous(), packed[:, 1].contiguous(), ts, te, sig, rgbs, sems, _cu(g["bkgd"]))
    for got, k in ((rgb, "rgb"), (acc, "acc"), (depth, "depth"), (sem, "sem")):
        np.testing.assert_allclose(got.detach().cpu().numpy(), g["trt_" + k], atol=1e-5, rtol=2e-6, err_msg=k)
    loss = pipeline_loss(rgb, depth, sem, _cu(g["pix"]), _cu(g["dep"]), _cu(g["lab"]))
    loss.backward()
    np.testing.assert_allclose(loss.item(), float(g["trt_loss"]), rtol=2e-6)
    np.testing.assert_allclose(rgbs.grad.cpu().numpy(), g["trt_g_rgbs"], atol=1e-8, rtol=1e-4)
    np.testing.assert_allclose(sig.grad.cpu().numpy(), g["trt_g_sigmas"], atol=6e-9, rtol=1e-4)
    np.testing.assert_allclose(sems.grad.cpu().numpy()[::4], g["trt_g_sems_every4"], atol=1e-8, rtol=1e-4)


def test_device_side_visibility_selection_equals_mask_indexing():
    """`sampling()`'s tail on the device (`mnf_visible_samples`: count pass, prefix sum, write pass) against the reference's own sequence — `render_visibility_from_density`
    (volrend.py:424-483) and three boolean-mask selections (occ_grid.py:229-236) — on ragged packed samples: empty rays, a ray longer than a wave, every threshold
    combination `sampling` can be called with, nothing visible at all, no rays."""
    from apnrf_amd import nerfacc as NA
    g = torch.Generator().manual_seed(3)
    cnts = torch.tensor([0, 5, 64, 65, 0, 200, 1, 33, 0], dtype=torch.int64)
    starts = torch.cumsum(cnts, 0) - cnts
    packed = torch.stack([starts, cnts], -1).to(DEV)
    n = int(cnts.sum())
    ts = torch.rand(n, generator=g).to(DEV)
    te = ts + 0.01 + 0.05 * torch.rand(n, generator=g).to(DEV)
    ri = torch.repeat_interleave(torch.arange(len(cnts)), cnts).to(DEV)
    for scale, eps, thre in ((40.0, 1e-4, 0.01), (40.0, 0.0, 0.05), (40.0, 1e-2, 0.0), (0.0, 1e-4, 0.01), (400.0, 0.5, 0.0)):
        sig = (torch.rand(n, generator=g) * scale).to(DEV)
        mask = NA.render_visibility_from_density(ts, te, sig, packed_info=packed, early_stop_eps=eps, alpha_thre=thre)
        got = NA._select_visible(ts, te, sig, packed, eps, torch.tensor([thre], device=DEV))
        for a, b, name in zip(got, (ri[mask], ts[mask], te[mask]), ("ray_indices", "t_starts", "t_ends")):
            assert a.dtype == b.dtype and torch.equal(a, b), (scale, eps, thre, name)
        if scale == 0.0:
            assert got[0].numel() == 0
    e = torch.empty(0, device=DEV)
    got = NA._select_visible(e, e, e, torch.zeros((0, 2), dtype=torch.int64, device=DEV), 1e-4, torch.tensor([0.01], device=DEV))
    assert all(t.numel() == 0 for t in got)


def test_differentiable_render_forms_positions_in_the_kernel_like_the_closure():
    """utils.py:122-137 inside `mnf_field_forward_train_samples` (what `sem_rendering` now calls under autograd) against the closure typed in torch
    in front of `NGPRadianceField.forward`: same per-sample outputs, same rendered values, parameter gradients within the scatter's atomic order."""
    from apnrf_amd import render as RD
    sc = H.make_scene(log2_hashmap_size=15)
    f, est = H.hip_field(sc).train(), H.hip_estimator(sc).train()
    g = torch.Generator().manual_seed(5)
    o = (torch.from_numpy(sc["aabb"][:3]) + (torch.rand(300, 3, generator=g) * 0.6 + 0.2) * torch.from_numpy(sc["aabb"][3:] - sc["aabb"][:3])).float().to(DEV)
    d = torch.nn.functional.normalize(torch.randn(300, 3, generator=g), dim=-1).to(DEV)
    ri, ts, te = est.sampling(o, d, sigma_fn=None, render_step_size=0.02, near_plane=0.05, stratified=False)
    assert len(ri) > 1000
    bk = torch.tensor([0.2, 0.5, 0.7], device=DEV)

    def closure_route():
        t_dirs = d[ri]
        positions = o[ri] + t_dirs * (ts + te)[:, None] / 2.0
        return f(positions, t_dirs)

    outs, grads = [], []
    for route in (closure_route, lambda: f.forward_samples_grad(o, d, ri, ts, te)):
        f.zero_grad()
        rgbs, sig, sems = route()
        (rgbs.sum() * 0.5 + sig.clamp(max=50).sum() * 0.01 + (sems * sems).sum() * 0.1).backward()
        outs.append([t.detach().clone() for t in (rgbs, sig, sems)])
        grads.append([p.grad.detach().clone() for p in (f.mlp_base.params, f.mlp_head.params, f.mlp_sem.params)])
    for a, b, name in zip(outs[0], outs[1], ("rgb", "sigma", "sem")):
        assert torch.equal(a, b), name
    for a, b, name in zip(grads[0], grads[1], ("base", "head", "sem")):
        np.testing.assert_allclose(b.cpu().numpy(), a.cpu().numpy(), rtol=2e-3, atol=2e-4 * float(a.abs().max()), err_msg=name)
    # and sem_rendering takes that route by itself
    f.zero_grad()
    rgb, acc, dep, sem, extra = RD.sem_rendering(f, RD.Rays(o, d), ts, te, ri, len(o), bk)
    assert torch.equal(extra["rgbs"].detach(), outs[0][0]) and torch.equal(extra["sigmas"].detach(), outs[0][1].squeeze(-1))
    (rgb.sum() + dep.sum() + sem.sum()).backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in (f.mlp_base.params, f.mlp_head.params, f.mlp_sem.params))


# ------------------------------------------------------------------ NGP field: the fused renderers (a14-a17) and the scorer (a18)
def _check(got, want, name, atol=1e-3, rtol=0.0):
    np.testing.assert_allclose(np.asarray(got), np.asarray(want), atol=atol, rtol=rtol, err_msg=name)


def test_fused_renderers_equal_reference_glue_golden(golden):
    from apnrf_amd import render as RD
    g = golden("glue_ngp")
    sc = scene_of(g)
    kw = sc["kw"]
    hip, _ = _hip_ngp_field(g, g["param_seed"])
    est = _hip_estimator(sc)
    rays = RD.Rays(_cu(g["rays_o"]), _cu(g["rays_d"]))
    bk = _cu(g["bkgd"])
    rgb, acc, depth, sem, tot = RD.render_image_with_occgrid_test(1024, hip, est, rays, render_bkgd=bk, **kw)
    for got, k in ((rgb, "rgb"), (acc, "acc"), (depth, "depth"), (sem, "sem")):
        _check(got.cpu().numpy(), g["test_" + k], k, rtol=1e-3 if k == "depth" else 0.0)
    assert abs(tot - int(g["test_total"])) <= 3
    rgb, rgb_var, acc, depth, depth_var, sem, tot = RD.render_probablistic_image_with_occgrid_test(1024, hip, est, rays, render_bkgd=bk, **kw)
    for got, k in ((rgb, "rgb"), (acc, "acc"), (depth, "depth"), (sem, "sem"), (rgb_var, "rgb_var"), (depth_var, "depth_var")):
        _check(got.cpu().numpy(), g["prob_" + k], k, atol=2e-3 if k == "depth_var" else 1e-3, rtol=2e-3 if k == "depth_var" else (1e-3 if k == "depth" else 0.0))
    assert abs(tot - int(g["prob_total"])) <= 3
    hip.eval()
    with torch.no_grad():
        rgb, acc, depth, sem, n = RD.render_image_with_occgrid_with_depth_guide(hip, est, rays, render_bkgd=bk, **kw)
    assert abs(n - int(g["tre_n"])) <= 3
    for got, k in ((rgb, "rgb"), (acc, "acc"), (depth, "depth"), (sem, "sem")):
        _check(got.cpu().numpy(), g["tre_" + k], k, rtol=1e-3 if k == "depth" else 0.0)


def test_fused_renderer_two_occupancy_levels_equals_reference_glue_golden(golden):
    """The reference's probabilistic render loop over an estimator with TWO occupancy levels (glue_ngp.npz `ml_*`) against the fused renderer."""
    from apnrf_amd import render as RD
    from apnrf_amd import synthetic as S
    from apnrf_amd.nerfacc import OccGridEstimator
    g = golden("glue_ngp")
    kw = scene_of(g)["kw"]
    occ, aabbs = multilevel_of(g)
    est = OccGridEstimator(torch.from_numpy(g["ml_roi"]), resolution=[50, 12, 50], levels=2)
    est.binaries = torch.from_numpy(occ)
    est = est.to(DEV).eval()
    np.testing.assert_array_equal(est.aabbs.cpu().numpy(), aabbs)
    lh = int(g["log2_hashmap_size"])
    scene = dict(aabb=aabbs[-1].astype(np.float32), neurons=128, layers=2, C=29, log2_hashmap_size=lh,
                 params=S.make_field_params(128, 2, 29, seed=int(g["param_seed"]), log2_hashmap_size=lh))
    hip = H.hip_field(scene)
    rgb, rgb_var, acc, depth, depth_var, sem, tot = RD.render_probablistic_image_with_occgrid_test(
        1024, hip, est, RD.Rays(_cu(g["ml_rays_o"]), _cu(g["ml_rays_d"])), render_bkgd=_cu(g["bkgd"]), **kw)
    assert abs(tot - int(g["ml_total"])) <= 3
    for got, k in ((rgb, "rgb"), (acc, "acc"), (depth, "depth"), (sem, "sem"), (rgb_var, "rgb_var"), (depth_var, "depth_var")):
        _check(got.cpu().numpy(), g["ml_" + k], k, atol=2e-3 if k == "depth_var" else 1e-3, rtol=2e-3 if k == "depth_var" else (1e-3 if k == "depth" else 0.0))


def test_pose_drivers_equal_reference_dataset_golden(golden):
    """habitat_to_data.py:304-549 as the reference ran them: [P,h,w,.] float64 stacks."""
    from apnrf_amd import render as RD
    g = golden("glue_ngp")
    sc = scene_of(g)
    kw = sc["kw"]
    hip, _ = _hip_ngp_field(g, g["param_seed"])
    est = _hip_estimator(sc)
    W, Hh, focal = (float(x) for x in g["pose_whf"])
    poses = g["poses"][g["pose_idx"]]
    args = (hip, est, poses, int(W), int(Hh), focal, kw["near_plane"], kw["render_step_size"], 0.1, kw["cone_angle"], kw["alpha_thre"], 4, DEV)
    images, depths, accs, sems = RD.render_image_from_pose(*args)
    for got, k in ((images, "images"), (depths, "depths"), (accs, "accs"), (sems, "sems")):
        assert got.dtype == np.float64 and got.shape == g["pose_" + k].shape
        _check(got, g["pose_" + k], k, rtol=1e-3 if k == "depths" else 0.0)
    out = RD.render_probablistic_image_from_pose(*args)
    for got, k in zip(out, ("images", "images_var", "depths", "depths_var", "accs", "sems")):
        assert got.dtype == np.float64 and got.shape == g["ppose_" + k].shape
        _check(got, g["ppose_" + k], k, atol=2e-3 if k == "depths_var" else 1e-3, rtol=2e-3 if k == "depths_var" else (1e-3 if k == "depths" else 0.0))


def test_scorer_equals_reference_probablistic_uncertainty_golden(golden):
    """scripts/pipeline.py:666-798 as the reference ran it on 40 views x 2 members: (i) `mnf_score_views` on the reference's own render
    stacks == its four terms to 1e-9; (ii) poses -> terms end to end (`score_views`, `score_poses`) against the same terms."""
    from apnrf_amd import render as RD
    g = golden("scorer")
    sc = scene_of(g)
    kw = sc["kw"]
    rv, dv, ac, sm = scorer_stacks(g)                       # [M,1,V,h,w,.]
    M, V = 2, 40
    terms = RD.score_view_terms(_cu(rv.reshape(M, V, 25, 3), torch.float32), _cu(dv.reshape(M, V, 25), torch.float32),
                                _cu(ac.reshape(M, V, 25), torch.float32), _cu(sm.reshape(M, V, 25, 29), torch.float32))
    t = terms.cpu().numpy()
    np.testing.assert_allclose(t.mean(0) * [1, 1, 3, 2], g["terms"], rtol=1e-9)
    np.testing.assert_allclose(float(RD.trajectory_score(terms).item()), float(g["pi"]), rtol=1e-9)
    hips = [_hip_ngp_field(g, s)[0] for s in g["param_seeds"]]
    ests = [_hip_estimator(sc), _hip_estimator(sc)]
    W, Hh, focal = (float(x) for x in g["whf"])
    poses = g["trajectory"][g["unc_idx"]]
    args = (hips, ests, poses, int(W), int(Hh), focal, kw["near_plane"], kw["render_step_size"], 0.1, kw["cone_angle"], kw["alpha_thre"], DEV)
    terms2, score = RD.score_views(*args)
    t2 = terms2.cpu().numpy().mean(0) * [1, 1, 3, 2]
    np.testing.assert_allclose(t2, g["terms"], rtol=5e-3, atol=2e-4)
    assert abs(float(score.item()) - float(g["pi"])) <= 5e-3 * abs(float(g["pi"])) + 5e-4
    terms3, score3 = RD.score_poses(*args)
    np.testing.assert_array_equal(terms3.cpu().numpy(), terms2.cpu().numpy())


# ------------------------------------------------------------------ repeatability of batched small-view renders
def test_batched_small_views_are_repeatable_and_batch_independent():
    """30 repetitions of a 24-view probabilistic batch (the scorer's shape) are bit-identical, and a view rendered alone equals the same view inside the
    batch: what lets candidate views shard over GPUs with bit-identical scores.  (Written for the view-queue renderer experiment of round 5,
    tools/experiments/viewq.patch, whose cross-workgroup hand-offs it guarded; it holds for the per-round renderer as well and stays.)"""
    from apnrf_amd import render as RD
    scene = H.make_scene(log2_hashmap_size=15)
    hip, est = H.hip_field(scene), H.hip_estimator(scene)
    os_, ds_ = [], []
    for p in range(24):
        o, d = H.view_rays(scene, p % 8, h=48 + (p % 3), w=48 + (p % 3))
        os_.append(o[:2304]); ds_.append(d[:2304])
    o, d = torch.cat(os_).to(DEV), torch.cat(ds_).to(DEV)
    bk = torch.zeros(3)
    ref = RD.render_views(hip, est, o, d, 2304, 1024, render_bkgd=bk, probabilistic=True, **H.RENDER_KW)
    assert int(ref["total"][1]) > 24 * 2304 * 20
    for _ in range(30):
        out = RD.render_views(hip, est, o, d, 2304, 1024, render_bkgd=bk, probabilistic=True, **H.RENDER_KW)
        for k in ("rgb", "acc", "depth", "sem", "rgb_var", "depth_var", "total"):
            assert torch.equal(out[k], ref[k]), k
    one = RD.render_views(hip, est, o[5 * 2304:6 * 2304], d[5 * 2304:6 * 2304], 2304, 1024, render_bkgd=bk, probabilistic=True, **H.RENDER_KW)
    for k in ("rgb", "acc", "depth", "sem", "rgb_var", "depth_var"):
        assert torch.equal(one[k], ref[k][5 * 2304:6 * 2304]), k


# ------------------------------------------------------------------ multi-GPU paths with real processes on the one GPU a test box has
def test_ticket_tile_order_renders_the_same_bits():
    """The field kernel of a large render round takes its tiles in arrival order (tickets, csrc/field.hip).  The order must not change a bit of the result: child
    process on the diagnostic library (tests/diag_tile_order.py), the same view with tickets twice and with the fixed stride."""
    import subprocess
    from apnrf_amd import build as B
    here = os.path.dirname(os.path.abspath(__file__))
    assert os.path.exists(B.LIB_DIAG), "libmi355nerf_diag.so missing: run `python __graft_entry__.py build`"
    r = subprocess.run([sys.executable, os.path.join(here, "diag_tile_order.py")], env=dict(os.environ, MNF_LIB_PATH=B.LIB_DIAG), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "DIAG_TILE_ORDER_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_two_processes_on_one_gpu_end_to_end():
    """Stand-in hand-over without a cache file, view-sharded scoring, view-sharded rendering and ray-data-parallel training with TWO processes and the real kernels
    (tests/two_ranks_one_gpu.py; gloo, both ranks on cuda:0 — RCCL needs one device per rank, which a one-GPU box cannot offer)."""
    import subprocess
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    procs = [subprocess.Popen([sys.executable, os.path.join(here, "two_ranks_one_gpu.py"), str(r), "2"], env=env, stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = []
    for p in procs:
        try:
            outs.append(p.communicate(timeout=600)[0])
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and f"TWO_RANKS_ONE_GPU_OK {r}" in o, o[-3000:]


def test_bench_runs_as_one_rank_job_under_torch_distributed_run():
    """The driver's N > 1 launch line with N = 1: `python -m torch.distributed.run --nproc-per-node 1 bench.py --gpus 1` forms an RCCL process group of one, takes
    every distributed branch of bench.py (barriers, the max over ranks, the sharded scoring pass with its all-gather) and prints ONE JSON line whose `n_gpus` is the
    size of the group RCCL formed."""
    import json
    import subprocess
    import tempfile
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    detail = os.path.join(tempfile.mkdtemp(), "detail.json")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(repo, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--workload", "score256", "--no-cpu-baseline", "--standin-steps", "200",
           "--detail-file", detail]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=repo)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and len(lines[0].encode()) < 4096, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["scaling"] in ("weak", "strong") and d["score256_ms"] > 0
    sc = json.load(open(detail))["score256"]                  # (the stdout line is the < 4 KB summary; every leg's full record is in the detail file)
    assert sc["n_gpus"] == 1
    assert len(sc["per_rank_compute_ms"]) == 1 and len(sc["per_rank_gather_ms"]) == 1 and sc["views"] == 256
    assert sc["ms_per_pass"] > 0 and np.isfinite(sc["score"])
