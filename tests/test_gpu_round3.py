"""Round-3 GPU tests: the train step that never waits for the host, the device-side skip decision and the optimizer's fp16
table mirror; parity on TRAINED weights (the benchmarked workload's kind of scene); a multi-step training trajectory against
the oracle (autograd + torch.optim.Adam + the reference's CyclicLR); ray-data-parallel training with an empty rank."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

import helpers as H

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _train_batch(sc, pose, h, w, seed):
    o, d = H.view_rays(sc, pose, h=h, w=w)
    rng = np.random.default_rng(seed)
    n = o.shape[0]
    pix = torch.from_numpy(rng.random((n, 3)).astype(np.float32))
    dep = torch.from_numpy(rng.uniform(0.5, 4.0, n).astype(np.float32))
    lab = torch.from_numpy(rng.integers(0, sc["C"], n))
    return o, d, pix, dep, lab


# ------------------------------------------------------------------ async train step
def test_train_step_without_host_sync_equals_synchronous_step():
    """sync=False (no host round trip, device-side skip decision) and sync=True (one read-back at the end) are the same kernels
    with the same arguments: same sample counts, same losses, same parameters after the optimizer (float atomics aside)."""
    from apnrf_amd import render as RD
    from apnrf_amd.optim import FusedAdam
    sc = H.make_scene(log2_hashmap_size=15)
    o, d, pix, dep, lab = (t.to(DEV) for t in _train_batch(sc, 3, 24, 24, 0))
    bk = torch.tensor([0.2, 0.4, 0.6], device=DEV)
    res = []
    for sync in (True, False):
        hip, est = H.hip_field(sc), H.hip_estimator(sc)
        opt = FusedAdam(hip.parameters(), lr=1e-3, eps=1e-15).bind_field(hip)
        outs = [RD.train_step(hip, est, opt, RD.Rays(o, d), pix, dep, lab, bk, step=s_, sync=sync, stratified=False, **H.RENDER_KW) for s_ in (1, 2, 3)]
        if not sync:
            assert all(isinstance(x["n_rendering_samples"], torch.Tensor) and isinstance(x["skipped"], torch.Tensor) for x in outs)
        res.append(([int(x["n_rendering_samples"]) for x in outs], [float(x["loss"]) for x in outs], [int(x["skipped"]) for x in outs],
                    [p.detach().clone() for p in hip.parameters()], float(opt.state[hip.mlp_head.params]["step"])))
    (n_a, l_a, s_a, p_a, st_a), (n_b, l_b, s_b, p_b, st_b) = res
    assert n_a[0] == n_b[0] and n_a[0] > 3000 and s_a == s_b == [0, 0, 0] and st_a == st_b == 3.0
    assert all(abs(x - y) <= max(3, 2e-3 * x) for x, y in zip(n_a, n_b))          # later steps: parameters differ by atomics order
    np.testing.assert_allclose(l_a, l_b, rtol=2e-3)
    for a, b in zip(p_a, p_b):
        if a.numel():
            assert torch.nn.functional.cosine_similarity(a, b, dim=0) > 0.9999


def test_fused_adam_skip_flag_and_table_mirror():
    """`FusedAdam.step(skip=flag)`: a raised device flag leaves parameters, moments and the step count untouched (pipeline.py:520-529
    decided on the device); `bind_field`: the hash table the kernels read after the step is the rounded new table (no conversion
    pass), bit-identical to a handle that re-loads the parameters."""
    from apnrf_amd.optim import FusedAdam
    sc = H.make_scene(log2_hashmap_size=14)
    a, b = H.hip_field(sc), H.hip_field(sc)
    g = torch.Generator().manual_seed(1)
    x = (torch.rand(5000, 3, generator=g) * 2 - 1).to(DEV) * torch.tensor([8.0, 1.5, 8.0], device=DEV) + torch.tensor([-9.0, 1.5, -9.0], device=DEV)
    dirs = torch.nn.functional.normalize(torch.randn(5000, 3, generator=g), dim=-1).to(DEV)
    oa, ob = FusedAdam(a.parameters(), lr=1e-2, eps=1e-15).bind_field(a), FusedAdam(b.parameters(), lr=1e-2, eps=1e-15)
    a(x, dirs); b(x, dirs)                                        # both handles hold the initial parameters
    for step in range(3):
        for pa, pb in zip(a.parameters(), b.parameters()):
            if pa.numel():
                gr = torch.randn(pa.shape, generator=g).to(DEV) * 1e-3
                pa.grad, pb.grad = gr.clone(), gr.clone()
        oa.step(); ob.step()
        with torch.no_grad():
            ra, rb = a(x, dirs), b(x, dirs)                       # a: mirrored table + refreshed fragments; b: full reload
        for u, v in zip(ra, rb):
            assert torch.equal(u, v)
        for pa, pb in zip(a.parameters(), b.parameters()):
            assert torch.equal(pa, pb)
    before = [p.detach().clone() for p in a.parameters()]
    m_before = oa.state[a.mlp_base.params]["exp_avg"].clone()
    skip = torch.ones((), dtype=torch.int32, device=DEV)
    oa.step(skip=skip)
    assert all(torch.equal(p, q) for p, q in zip(a.parameters(), before))
    assert torch.equal(oa.state[a.mlp_base.params]["exp_avg"], m_before) and float(oa.state[a.mlp_base.params]["step"]) == 3.0
    oa.step(skip=torch.zeros((), dtype=torch.int32, device=DEV))
    base_before = next(q for p, q in zip(a.parameters(), before) if p is a.mlp_base.params)
    assert float(oa.state[a.mlp_base.params]["step"]) == 4.0 and not torch.equal(a.mlp_base.params.detach(), base_before)


def test_train_step_flags_bad_labels_and_recovers_from_small_bounds():
    """a class id outside [0, C) must not be read past the logits row: the step is flagged on the device (status bit
    8, skip raised) and surfaces as an error.  Sample bounds that are too small end the step on the device with zero gradients;
    the synchronous path repeats it with larger bounds, the asynchronous one skips it and grows the bounds for the next call."""
    from apnrf_amd import _lib as L
    from apnrf_amd import render as RD
    sc = H.make_scene(log2_hashmap_size=14)
    hip, est = H.hip_field(sc), H.hip_estimator(sc)
    hip.train()
    o, d, pix, dep, lab = (t.to(DEV) for t in _train_batch(sc, 2, 16, 16, 1))
    bad = lab.clone(); bad[7] = 29; bad[100] = -100
    with pytest.raises(L.MnfError, match="class id"):
        RD.fused_forward_backward(hip, est, RD.Rays(o, d), pix, dep, bad, None, stratified=False, **H.RENDER_KW)
    out = RD.fused_forward_backward(hip, est, RD.Rays(o, d), pix, dep, lab, None, stratified=False, **H.RENDER_KW)
    n_ok, g_ok = out["n_rendering_samples"], hip.mlp_head.params.grad.clone()
    assert n_ok > 1000 and int(out["skip"]) == 0
    key = id(hip)
    R_ = o.shape[0]
    small = dict(by_R={R_: (2048, 1024)}, abs_m=0, abs_k=0, per_m=0.0, per_k=0.0)
    RD._TRAIN_STATE[key].update(small)                            # far too small
    out = RD.fused_forward_backward(hip, est, RD.Rays(o, d), pix, dep, lab, None, stratified=False, **H.RENDER_KW)
    assert out["n_rendering_samples"] == n_ok and RD._caps_for(RD._TRAIN_STATE[key], o.shape[0])[1] >= n_ok       # repeated with larger bounds
    np.testing.assert_allclose(hip.mlp_head.params.grad.cpu().numpy(), g_ok.cpu().numpy(), rtol=2e-2, atol=1e-6)
    RD._TRAIN_STATE[key].update(dict(small, by_R={R_: (2048, 1024)}))
    lazy = RD.fused_forward_backward(hip, est, RD.Rays(o, d), pix, dep, lab, None, stratified=False, sync=False, **H.RENDER_KW)
    assert int(lazy["skip"]) > 0 and int(lazy["counts"][3]) & 1 and float(hip.mlp_head.params.grad.abs().max()) == 0.0
    # WITHOUT any synchronisation by the caller the bounds are corrected at most two calls late (per bound: marched, then surviving)
    skips = []
    for _ in range(6):
        lazy = RD.fused_forward_backward(hip, est, RD.Rays(o, d), pix, dep, lab, None, stratified=False, sync=False, **H.RENDER_KW)
        skips.append(lazy["skip"])
    skips = [int(x) for x in skips]
    assert skips[-1] == 0 and skips[-2] == 0 and int(lazy["n_rendering_samples"]) == n_ok, skips
    # rays that miss the grid: no sample, skip raised with status bit 16, zero gradients (pipeline.py:491 `continue`)
    up = torch.zeros_like(d); up[:, 1] = 1.0
    far_o = o + torch.tensor([0.0, 100.0, 0.0], device=DEV)
    miss = RD.fused_forward_backward(hip, est, RD.Rays(far_o, up), pix, dep, lab, None, stratified=False, **H.RENDER_KW)
    assert miss["n_rendering_samples"] == 0 and int(miss["skip"]) > 0 and int(miss["counts"][3]) & 16


# ------------------------------------------------------------------ parity on trained weights
@pytest.fixture(scope="module")
def trained():
    """A small trained stand-in (300 iterations of the product's own train step on the analytic rooms target): sharp densities,
    saturated opacities, an occupancy grid that came out of update_every_n_steps — what the benchmarked scenes look like."""
    from apnrf_amd import standin as SI
    sc = H.make_scene("102344250", n_poses=40)
    field, est, info = SI.train_standin(sc, DEV, steps=300, seed=21)
    sc = dict(sc)
    sc["params"] = {"mlp_base": field.mlp_base.params.detach().cpu().numpy(), "mlp_head": field.mlp_head.params.detach().cpu().numpy(),
                    "mlp_sem": field.mlp_sem.params.detach().cpu().numpy()}
    sc["occ"] = est.binaries.cpu().numpy()
    return sc, field, est, info


def _render_errors(out, ref, prob):
    errs = {}
    for k in ("rgb", "acc", "depth", "sem") + (("rgb_var", "depth_var") if prob else ()):
        a, b = out[k].cpu().numpy().reshape(ref[k].shape[0], -1), ref[k].numpy().reshape(ref[k].shape[0], -1)
        errs[k] = np.abs(a - b).max(axis=1)
    return errs


@pytest.mark.parametrize("prob", [False, True])
def test_render_trained_weights_matches_oracle(trained, prob):
    """utils.py:555-779 / :782-1032 on TRAINED weights: north-star tolerance 1e-3 abs on every output; rays whose
    alpha-threshold / termination decision flips on an fp32 tie get the tie budget (<= 2 rays up to 5e-2)."""
    from apnrf_amd import render as RD
    from oracle import render as R
    sc, field, est, info = trained
    orc = H.oracle_field(sc)
    bk = torch.tensor([0.1, 0.3, 0.6])
    tie_rays = 0
    for pose in (0, 13):
        o, d = H.view_rays(sc, pose, h=24, w=24)
        fn = R.render_prob_test if prob else R.render_test
        ref = fn(1024, orc, sc["occ"], sc["aabb"][None], o, d, render_bkgd=bk, **H.RENDER_KW)
        out = RD.render_views(field, est, o.to(DEV), d.to(DEV), o.shape[0], 1024, render_bkgd=bk, probabilistic=prob, **H.RENDER_KW)
        errs = _render_errors(out, ref, prob)
        worst = np.max(np.stack([errs[k] / (1.0 if k != "depth" else 1.0) for k in errs]), axis=0)
        tie = worst > 1e-3
        tie_rays += int(tie.sum())
        assert worst[~tie].max() <= 1e-3 and (worst[tie] <= 5e-2).all(), {k: float(v.max()) for k, v in errs.items()}
        tot = int(out["total"][0])
        assert abs(tot - ref["total_samples"]) <= max(4, 2e-3 * ref["total_samples"]), (tot, ref["total_samples"])
        mse = float(((out["rgb"].cpu() - ref["rgb"]) ** 2).mean())
        assert 10 * np.log10(1.0 / max(mse, 1e-20)) > 50.0
    assert tie_rays <= 2, tie_rays


def test_train_gradients_trained_weights_match_oracle(trained):
    """One train-render + loss + backward on TRAINED weights against oracle autograd (the round-2 checks ran on random-init weights)."""
    import torch.nn.functional as F
    from apnrf_amd import render as RD
    from oracle import render as R
    sc, field, est, info = trained
    orc = H.oracle_field(sc, requires_grad=True)
    o, d, pix, dep, lab = _train_batch(sc, 5, 12, 12, 4)
    bk = torch.tensor([0.5, 0.2, 0.9])
    field.train()
    out = RD.fused_forward_backward(field, est, RD.Rays(o.to(DEV), d.to(DEV)), pix.to(DEV), dep.to(DEV), lab.to(DEV), bk, stratified=False, **H.RENDER_KW)
    field.eval()
    ref = R.render_train(orc, sc["occ"], sc["aabb"][None], float(est.occs.mean().item()), o, d, torch.full((o.shape[0],), 0.1),
                         render_bkgd=bk, render_step_size=1e-3, cone_angle=0.004, alpha_thre=0.01)
    r_loss = F.smooth_l1_loss(ref[0], pix) * 10 + F.smooth_l1_loss(ref[2], dep.unsqueeze(1)) / 5 + F.cross_entropy(ref[3], lab) / 2
    r_loss.backward()
    assert abs(out["n_rendering_samples"] - ref[4]) <= max(3, 2e-3 * ref[4]) and ref[4] > 500
    np.testing.assert_allclose(float(out["loss"]), r_loss.item(), rtol=5e-4)
    n_mlp = sum(o_ * i_ for o_, i_ in orc.shapes["base"])
    for name, g, r in [("base mlp", field.mlp_base.params.grad[:n_mlp], orc.p_base.grad[:n_mlp]), ("hash table", field.mlp_base.params.grad[n_mlp:], orc.p_base.grad[n_mlp:]),
                       ("rgb head", field.mlp_head.params.grad, orc.p_head.grad), ("sem head", field.mlp_sem.params.grad, orc.p_sem.grad)]:
        g, r = g.detach().cpu().double(), r.double()
        rel = float((g - r).norm() / r.norm().clamp_min(1e-30))
        cos = float(torch.dot(g, r) / (g.norm() * r.norm()).clamp_min(1e-30))
        assert rel < 5e-2 and cos > 0.998, (name, rel, cos)


# ------------------------------------------------------------------ multi-step training trajectory vs the oracle
def test_training_trajectory_matches_oracle():
    """VERDICT r02 missing 5: the reference trains thousands of steps per phase (pipeline.py:398-535).  Same initial parameters, the
    same 30 ray batches, no jitter, no occupancy refresh inside the window: the HIP train step (fused forward / loss / backward,
    device-side guard, FusedAdam with the fp16 table mirror) against the oracle (autograd through the fp16-rounded forward,
    torch.optim.Adam), both behind the reference's CyclicLR (pipeline.py:183-193)."""
    import torch.nn.functional as F
    from apnrf_amd import render as RD
    from apnrf_amd.optim import FusedAdam
    from oracle import render as R
    sc = H.make_scene(log2_hashmap_size=15)
    hip, orc = H.hip_field(sc), H.oracle_field(sc, requires_grad=True)
    est = H.hip_estimator(sc)
    occs_mean = float(est.occs.mean().item())
    n_steps = 30
    sched_kw = dict(base_lr=1e-4, max_lr=1e-3, step_size_up=int(n_steps / 4), mode="exp_range", gamma=1.0, cycle_momentum=False)
    opt_h = FusedAdam(hip.parameters(), lr=1e-3, eps=1e-15).bind_field(hip)
    opt_o = torch.optim.Adam([orc.p_base, orc.p_head, orc.p_sem], lr=1e-3, eps=1e-15)
    sch_h = torch.optim.lr_scheduler.ChainedScheduler([torch.optim.lr_scheduler.CyclicLR(opt_h, **sched_kw)])
    sch_o = torch.optim.lr_scheduler.ChainedScheduler([torch.optim.lr_scheduler.CyclicLR(opt_o, **sched_kw)])
    bk = torch.tensor([0.3, 0.3, 0.3])
    steps = [s for s in range(1, 40) if s % 16][:n_steps]          # no occupancy refresh (random cell draws) inside the window
    loss_h, loss_o, n_h, n_o = [], [], [], []
    for k, step in enumerate(steps):
        o, d, pix, dep, lab = _train_batch(sc, k % 6, 10, 10, 100 + k % 6)      # six batches with fixed targets, visited five times each
        out = RD.train_step(hip, est, opt_h, RD.Rays(o.to(DEV), d.to(DEV)), pix.to(DEV), dep.to(DEV), lab.to(DEV), bk.to(DEV), step=step,
                            scheduler=sch_h, stratified=False, **H.RENDER_KW)
        assert not out["skipped"]
        loss_h.append(float(out["loss"])); n_h.append(out["n_rendering_samples"])
        ref = R.render_train(orc, sc["occ"], sc["aabb"][None], occs_mean, o, d, torch.full((o.shape[0],), 0.1), render_bkgd=bk,
                             render_step_size=1e-3, cone_angle=0.004, alpha_thre=0.01)
        r_loss = F.smooth_l1_loss(ref[0], pix) * 10 + F.smooth_l1_loss(ref[2], dep.unsqueeze(1)) / 5 + F.cross_entropy(ref[3], lab) / 2
        opt_o.zero_grad(); r_loss.backward(); opt_o.step(); sch_o.step()
        orc._derive()
        loss_o.append(r_loss.item()); n_o.append(ref[4])
        assert abs(opt_h.param_groups[0]["lr"] - opt_o.param_groups[0]["lr"]) < 1e-12
    loss_h, loss_o = np.array(loss_h), np.array(loss_o)
    rel = np.abs(loss_h - loss_o) / loss_o
    print("trajectory: loss first/last", loss_o[0], loss_o[-1], "max rel diff", rel.max(), "samples", n_h[-1], n_o[-1])
    assert loss_o[-6:].mean() < loss_o[:6].mean(), (loss_o[:6], loss_o[-6:])       # it does train (same six batches, fifth visit vs first)
    assert rel.max() < 1e-3, rel
    assert all(abs(a - b) <= max(3, 3e-3 * b) for a, b in zip(n_h, n_o))
    n_mlp = sum(o_ * i_ for o_, i_ in orc.shapes["base"])
    init = sc["params"]
    for name, p, q, p0 in [("base mlp", hip.mlp_base.params[:n_mlp], orc.p_base[:n_mlp], init["mlp_base"][:n_mlp]),
                           ("hash table", hip.mlp_base.params[n_mlp:], orc.p_base[n_mlp:], init["mlp_base"][n_mlp:]),
                           ("rgb head", hip.mlp_head.params, orc.p_head, init["mlp_head"]), ("sem head", hip.mlp_sem.params, orc.p_sem, init["mlp_sem"])]:
        p, q, p0 = p.detach().cpu().double(), q.detach().double(), torch.from_numpy(p0).double()
        cos = float(torch.dot(p, q) / (p.norm() * q.norm()))
        dp, dq = p - p0, q - p0
        cos_d = float(torch.dot(dp, dq) / (dp.norm() * dq.norm()).clamp_min(1e-30))
        print(f"trajectory: {name}: parameter cosine {cos:.6f}, update cosine {cos_d:.4f}")
        assert cos > 0.999, (name, cos)
        assert cos_d > 0.9, (name, cos_d)


# ------------------------------------------------------------------ ray-data-parallel step with an empty rank
def test_data_parallel_step_with_an_empty_rank_does_not_hang():
    """Two ranks on this GPU (gloo): rank 1's rays miss the grid.  Both ranks must reach the gradient all-reduce, take the same
    skip decision and finish (round 2 returned early on the empty rank and left the other one inside the collective)."""
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29641", HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, os.path.join(here, "dp_empty_rank.py"), str(r), "2"], env=env, stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = []
    for p in procs:
        try:
            outs.append(p.communicate(timeout=150)[0])
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            pytest.fail("data-parallel step hung")
    assert all(p.returncode == 0 for p in procs) and all("DP_EMPTY_RANK_OK" in o for o in outs), "\n".join(o[-1500:] for o in outs)


def test_deterministic_training_is_bitwise_reproducible():
    """`train_step(deterministic=True)` (64-bit fixed-point table-gradient atomics, ordered weight-gradient partial sums) with
    seeded generators: two trainings of the stand-in protocol give bit-identical parameters and occupancy grids — what makes the
    benchmark's trained scenes the same on every box.  The default mode (float atomics) agrees with it to
    rounding."""
    from apnrf_amd import render as RD
    from apnrf_amd import standin as SI
    from apnrf_amd.optim import FusedAdam
    sc = H.make_scene("102344250", n_poses=40, log2_hashmap_size=17)
    runs = []
    for _ in range(2):
        field, est, info = SI.train_standin(sc, DEV, steps=120, seed=33, cache_dir=os.path.join("/tmp", f"mnf_nocache_{os.getpid()}_{len(runs)}"))
        assert not info["cached"]
        runs.append(([p.detach().clone() for p in field.parameters()], est.binaries.clone(), est.occs.clone()))
    for a, b in zip(runs[0][0], runs[1][0]):
        assert torch.equal(a, b)
    assert torch.equal(runs[0][1], runs[1][1]) and torch.equal(runs[0][2], runs[1][2])
    # one step, both accumulation modes, same inputs: the gradients agree to float-atomics rounding
    hip, est = H.hip_field(sc), H.hip_estimator(sc)
    hip.train()
    o, d, pix, dep, lab = (t.to(DEV) for t in _train_batch(sc, 3, 24, 24, 0))
    grads = []
    for det in (True, False, True):
        RD.fused_forward_backward(hip, est, RD.Rays(o, d), pix, dep, lab, None, stratified=False, deterministic=det, **H.RENDER_KW)
        grads.append([p.grad.clone() for p in hip.parameters() if p.numel()])
    for a, b, c in zip(*grads):
        assert torch.equal(a, c)                                   # deterministic mode: bit-identical from call to call
        assert float((a - b).norm() / a.norm().clamp_min(1e-30)) < 1e-4


def test_field_kernels_are_bitwise_repeatable():
    """Guard against the class of fault behind -DMNF_PK=1 (profiles/r03_pk_rootcause.txt: a timing-dependent miscompute in lanes 48..63 whose
    hazard was not identified; the packed instruction itself was exonerated).  The inference kernels are deterministic, so any such fault shows up as
    a run-to-run difference: every instantiation the benchmarks use (explicit positions, packed samples, density only, the renderer with fused
    compositing, fp16 and bf16 operands) is run 40 times on the same inputs and must give bit-identical outputs."""
    from apnrf_amd import render as RD
    sc = H.make_scene()
    rng = np.random.default_rng(1)
    n = 5037
    a = sc["aabb"]
    pos = torch.from_numpy((rng.random((n, 3)) * (a[3:] - a[:3]) * 1.1 + a[:3] - 0.05 * (a[3:] - a[:3])).astype(np.float32)).to(DEV)
    d = rng.normal(size=(n, 3)).astype(np.float32); d /= np.linalg.norm(d, axis=-1, keepdims=True)
    d = torch.from_numpy(d).to(DEV)
    o_v, d_v = H.view_rays(sc, 2, h=48, w=48)
    o_v, d_v = o_v.to(DEV), d_v.to(DEV)
    est = H.hip_estimator(sc)
    for bf16 in (False, True):
        f = H.hip_field(sc, mfma_bf16=bf16)
        with torch.no_grad():
            ref_f = [t.clone() for t in f(pos, d)]
            ref_d = f.query_density(pos).clone()
            ref_r = RD.render_views(f, est, o_v, d_v, o_v.shape[0], 1024, render_bkgd=torch.zeros(3), probabilistic=True, **H.RENDER_KW)
            for rep in range(40):
                got = f(pos, d)
                assert all(torch.equal(x, y) for x, y in zip(got, ref_f)), (bf16, rep, "forward")
                assert torch.equal(f.query_density(pos), ref_d), (bf16, rep, "density")
                if rep % 4 == 0:
                    r = RD.render_views(f, est, o_v, d_v, o_v.shape[0], 1024, render_bkgd=torch.zeros(3), probabilistic=True, **H.RENDER_KW)
                    assert all(torch.equal(r[k], ref_r[k]) for k in ("rgb", "acc", "depth", "sem", "rgb_var", "depth_var", "total")), (bf16, rep, "render")


@pytest.mark.parametrize("levels,prob", [(2, False), (3, True)])
def test_multi_level_occupancy_render_matches_oracle(levels, prob):
    """VERDICT r02 missing 6: occupancy grids with several levels (occ_grid.py:37-55: level l covers the roi enlarged 2^l times;
    perception/models/utils.py:637-644 hands all of them to traverse_grids) in the fused test-time renderer: a ray's segments are taken level by
    level as grid.cu:125-151 does.  Sample totals equal, outputs within 1e-3 of the oracle, and the masks of a one-round call bit-exact."""
    from apnrf_amd import render as RD
    from apnrf_amd.nerfacc import OccGridEstimator
    from oracle import render as R
    sc = H.make_scene(log2_hashmap_size=15)
    roi = np.array([-16.0, 0.0, -16.0, -6.0, 2.4, -6.0], np.float32)              # inner region; level l is 2^l times larger around its centre
    est = OccGridEstimator(torch.from_numpy(roi), resolution=[50, 12, 50], levels=levels)
    rng = np.random.default_rng(7)
    occ = rng.random((levels, 50, 12, 50)) < np.array([0.12, 0.08, 0.05])[:levels, None, None, None]
    est.binaries = torch.from_numpy(occ)
    est = est.to(DEV).eval()
    aabbs = est.aabbs.cpu().numpy()
    field_scene = dict(sc); field_scene["aabb"] = aabbs[-1].astype(np.float32)   # the field covers the largest level (pipeline.py:167-172: aabb = estimator.aabbs[-1])
    hip, orc = H.hip_field(field_scene), H.oracle_field(field_scene)
    o, d = H.view_rays(sc, 3, h=24, w=24)
    o = o + torch.tensor([3.0, 0.0, 3.0])                                          # inside the roi
    bk = torch.tensor([0.2, 0.1, 0.4])
    fn = R.render_prob_test if prob else R.render_test
    ref = fn(1024, orc, occ, aabbs, o, d, render_bkgd=bk, **H.RENDER_KW)
    out = RD.render_views(hip, est, o.to(DEV), d.to(DEV), o.shape[0], 1024, render_bkgd=bk, probabilistic=prob, **H.RENDER_KW)
    assert ref["total_samples"] > 5000 and len(ref["rounds"]) > 3
    assert abs(int(out["total"][0]) - ref["total_samples"]) <= max(3, 2e-3 * ref["total_samples"])
    for k in ("rgb", "acc", "depth", "sem") + (("rgb_var", "depth_var") if prob else ()):
        assert bool(torch.isfinite(out[k]).all()) and bool(torch.isfinite(ref[k]).all()), (k, "non-finite", int((~torch.isfinite(out[k])).sum()), int((~torch.isfinite(ref[k])).sum()),
                                                                                          (~torch.isfinite(out[k].cpu().reshape(o.shape[0], -1))).any(1).nonzero().flatten().tolist()[:8])
        err = (out[k].cpu() - ref[k]).abs().reshape(o.shape[0], -1).max(dim=1).values
        assert int((err > 1e-3).sum()) <= 2 and float(err.max()) < 5e-2, (k, float(err.max()), int((err > 1e-3).sum()))
    # samples come from more than one level: rays that start inside level 0 and leave it keep marching through level 1
    far = RD.render_views(hip, est, o.to(DEV), d.to(DEV), o.shape[0], 1024, render_bkgd=bk, **H.RENDER_KW)
    one = OccGridEstimator(torch.from_numpy(roi), resolution=[50, 12, 50], levels=1)
    one.binaries = torch.from_numpy(occ[:1]); one = one.to(DEV).eval()
    near_only = RD.render_views(H.hip_field(field_scene), one, o.to(DEV), d.to(DEV), o.shape[0], 1024, render_bkgd=bk, **H.RENDER_KW)
    assert int(far["total"][1]) > int(near_only["total"][1])


def test_results_do_not_depend_on_stale_workspace_contents():
    """The cached scratch buffers are never cleared between calls and their layout changes with the call's size, so an unused column can hold
    anything: all-ones bytes are NaN as floats and -1 as ray ids.  Found by the multi-level test when it ran after the rest of the suite: a depth of
    0 * NaN.  Render (both modes), scoring and the fused train step (deterministic accumulation) are run, every cached workspace is overwritten
    with 0xFF, and the same calls must give bit-identical, finite results."""
    from apnrf_amd import render as RD
    sc = H.make_scene(log2_hashmap_size=15)
    hip, est = H.hip_field(sc), H.hip_estimator(sc)
    o, d = H.view_rays(sc, 1, h=40, w=40)
    o, d = o.to(DEV), d.to(DEV)
    o3 = torch.cat([H.view_rays(sc, i, h=20, w=20)[0] for i in range(3)]).to(DEV)
    d3 = torch.cat([H.view_rays(sc, i, h=20, w=20)[1] for i in range(3)]).to(DEV)
    tb = [t.to(DEV) for t in _train_batch(sc, 3, 24, 24, 0)]

    def run():
        res = []
        for prob in (False, True):
            out = RD.render_views(hip, est, o, d, o.shape[0], 1024, probabilistic=prob, **H.RENDER_KW)
            res += [out[k].clone() for k in ("rgb", "acc", "depth", "sem") + (("rgb_var", "depth_var") if prob else ())]
        out = RD.render_views(hip, est, o3, d3, 400, 1024, probabilistic=True, **H.RENDER_KW)
        res += [out[k].clone() for k in ("rgb", "depth", "depth_var")]
        hip.train()
        fb = RD.fused_forward_backward(hip, est, RD.Rays(tb[0], tb[1]), tb[2], tb[3], tb[4], None, stratified=False, deterministic=True, **H.RENDER_KW)
        hip.eval()
        res += [fb["loss"].detach().clone()] + [p.grad.clone() for p in hip.parameters() if p.numel()]
        return res

    first = run()
    assert len(RD._WORKSPACES) > 0
    for ws in RD._WORKSPACES.values():
        ws.fill_(0xFF)
    second = run()
    for i, (a, b) in enumerate(zip(first, second)):
        assert bool(torch.isfinite(b).all()), i
        assert torch.equal(a, b), (i, float((a - b).abs().max()))


def test_binned_scatter_equals_walk():
    """The table gradient of the fine hashed levels goes through per-bin item lists and LDS sums (csrc/train.hip bin_items_kernel /
    bin_accumulate_kernel) instead of memory-side float atomics.  Same contributions, 21 significant bits per item, another summation order: the
    result equals the walk's to 2e-6 relative L2, also when the lists are made so short that nearly every item takes the full-list route.  The knobs
    that select the routes exist only in the diag build: child process (tests/diag_bins.py)."""
    import subprocess
    import sys
    from apnrf_amd import build as B
    here = os.path.dirname(os.path.abspath(__file__))
    assert os.path.exists(B.LIB_DIAG), "libmi355nerf_diag.so missing: run `python __graft_entry__.py build`"
    env = dict(os.environ, MNF_LIB_PATH=B.LIB_DIAG)
    r = subprocess.run([sys.executable, os.path.join(here, "diag_bins.py")], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "DIAG_BINS_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_field_backward_matches_oracle_binned_levels():
    """test_field_backward_matches_oracle at a batch large enough (>= 8192 samples) for the product library to send levels 11..15 through the bins:
    dL/d(params) against torch autograd through the oracle, same tolerances."""
    from test_gpu_parity import _grad_close
    sc = H.make_scene(neurons=128, layers=2, C=29, log2_hashmap_size=15, head_gain=2.0)
    hip = H.hip_field(sc).train()
    orc = H.oracle_field(sc, requires_grad=True)
    rng = np.random.default_rng(17)
    n = 12000 + 37
    a = sc["aabb"]
    pos = (rng.random((n, 3)) * (a[3:] - a[:3]) * 0.98 + a[:3] + 0.01 * (a[3:] - a[:3])).astype(np.float32)
    pos[:7] = a[:3] - 1.0
    d = rng.normal(size=(n, 3)).astype(np.float32); d /= np.linalg.norm(d, axis=-1, keepdims=True)
    g_rgb = (rng.normal(size=(n, 3)) * 1e-3).astype(np.float32)
    g_sig = (rng.normal(size=(n, 1)) * 1e-5).astype(np.float32)
    g_sem = (rng.normal(size=(n, 29)) * 1e-3).astype(np.float32)
    cu = lambda x: torch.from_numpy(x).to(DEV)
    rgb, sigma, sem = hip(cu(pos), cu(d))
    torch.autograd.backward([rgb, sigma, sem], [cu(g_rgb), cu(g_sig), cu(g_sem)])
    r_rgb, r_sigma, r_sem = orc(torch.from_numpy(pos), torch.from_numpy(d))
    torch.autograd.backward([r_rgb, r_sigma, r_sem], [torch.from_numpy(g_rgb), torch.from_numpy(g_sig), torch.from_numpy(g_sem)])
    n_mlp = sum(o * i for o, i in orc.shapes["base"])
    _grad_close(hip.mlp_base.params.grad[:n_mlp], orc.p_base.grad[:n_mlp], "base mlp")
    _grad_close(hip.mlp_base.params.grad[n_mlp:], orc.p_base.grad[n_mlp:], "hash table")
    # per level of the table: the binned levels are as close as the walked ones
    got, want = hip.mlp_base.params.grad[n_mlp:].double().cpu().view(-1, 4), orc.p_base.grad[n_mlp:].double().view(-1, 4)
    size = 1 << 15
    for first in (got.shape[0] - size, got.shape[0] - 3 * size):             # levels 15 and 13
        e = float((got[first:first + size] - want[first:first + size]).norm() / want[first:first + size].norm())
        assert e < 2e-2, (first, e)
    untouched = (orc.p_base.grad[n_mlp:] == 0).numpy()
    assert (hip.mlp_base.params.grad[n_mlp:].cpu().numpy()[untouched] == 0).all()


def _multi_level_estimator(levels, seed=7):
    """Occupancy grid with `levels` nested levels around an inner region of the scene (occ_grid.py:37-55), random cells, occs = 0.05 where occupied."""
    from apnrf_amd.nerfacc import OccGridEstimator
    roi = np.array([-16.0, 0.0, -16.0, -6.0, 2.4, -6.0], np.float32)
    est = OccGridEstimator(torch.from_numpy(roi), resolution=[50, 12, 50], levels=levels)
    rng = np.random.default_rng(seed)
    occ = rng.random((levels, 50, 12, 50)) < np.array([0.12, 0.08, 0.05, 0.04])[:levels, None, None, None]
    est.binaries = torch.from_numpy(occ)
    est.occs = torch.from_numpy(occ.reshape(-1).astype(np.float32)) * 0.05
    return est.to(DEV), occ


def test_multi_level_single_pass_sampler_bit_exact():
    """The single-pass sampler over several occupancy levels (mnf_sample_rays_levels: march_dev.h march_levels) gives the samples of the two-pass
    `traverse_grids` of the same library and of the oracle marcher (grid.cu:125-151), bit for bit."""
    from apnrf_amd import nerfacc as NA
    from oracle import marcher as M
    sc = H.make_scene(log2_hashmap_size=15)
    for levels in (2, 3):
        est, occ = _multi_level_estimator(levels)
        o, d = H.view_rays(sc, 3, h=24, w=24)
        o = (o + torch.tensor([3.0, 0.0, 3.0])).to(DEV); d = d.to(DEV)
        near = torch.full((o.shape[0],), 0.1, device=DEV) + torch.rand(o.shape[0], generator=torch.Generator().manual_seed(5)).to(DEV) * 1e-3
        far = torch.full_like(near, 1e10)
        got = est._sample_single_pass(o, d, near, far, 1e-3, 0.004)
        assert got is not None
        ri, ts, te, packed = got
        iv, sm, _ = NA.traverse_grids(o, d, est.binaries, est.aabbs, near_planes=near, far_planes=far, step_size=1e-3, cone_angle=0.004)
        np.testing.assert_array_equal(ts.cpu().numpy(), iv.vals[iv.is_left].cpu().numpy())
        np.testing.assert_array_equal(te.cpu().numpy(), iv.vals[iv.is_right].cpu().numpy())
        np.testing.assert_array_equal(ri.cpu().numpy(), sm.ray_indices.cpu().numpy())
        ref = M.traverse_grids(o.cpu().numpy(), d.cpu().numpy(), occ, est.aabbs.cpu().numpy(), near_planes=near.cpu().numpy(), far_planes=far.cpu().numpy(),
                               step_size=1e-3, cone_angle=0.004)
        np.testing.assert_array_equal(ts.cpu().numpy(), ref[0].vals[ref[0].is_left])
        assert ts.shape[0] > 5000


def test_multi_level_fused_train_step_and_scoring():
    """VERDICT r02 missing 6, second half: the fused train step and `mnf_score_poses` with a two-level occupancy grid.  Train step: identical sample
    count, losses to fp32 rounding and gradients to summation-order noise against the call-by-call autograd route (whose sampler is checked against the
    oracle above; occ_grid.py:192 takes the mean of `occs` over every level).  Scoring: the single C call equals `score_views` bit for bit."""
    import torch.nn.functional as F
    from apnrf_amd import render as RD
    sc = H.make_scene(log2_hashmap_size=15, seed=6)
    est, occ = _multi_level_estimator(2)
    fs = dict(sc); fs["aabb"] = est.aabbs[-1].cpu().numpy().astype(np.float32)
    o, d = H.view_rays(sc, 3, h=24, w=24)
    rays = RD.Rays((o + torch.tensor([3.0, 0.0, 3.0])).to(DEV), d.to(DEV))
    rng = np.random.default_rng(3)
    pix = torch.from_numpy(rng.random((576, 3)).astype(np.float32)).to(DEV)
    dep = torch.from_numpy(rng.uniform(0.5, 4.0, 576).astype(np.float32)).to(DEV)
    lab = torch.from_numpy(rng.integers(0, 29, 576)).to(DEV)
    bk = torch.tensor([0.5, 0.2, 0.9], device=DEV)
    a, b = H.hip_field(fs), H.hip_field(fs)
    a.eval()
    rgb, acc, depth, sem, n = RD.render_image_with_occgrid_with_depth_guide(a, est, rays, render_bkgd=bk, **H.RENDER_KW)
    assert n > 3000
    l_rgb, l_dep, l_sem = F.smooth_l1_loss(rgb, pix), F.smooth_l1_loss(depth, dep.unsqueeze(1)), F.cross_entropy(sem, lab)
    (l_rgb * 10 + l_dep / 5 + l_sem / 2).backward()
    out = RD.fused_forward_backward(b, est, rays, pix, dep, lab, bk, stratified=False, **H.RENDER_KW)
    assert out is not None and out["n_rendering_samples"] == n
    for got, want in ((out["loss_rgb"], l_rgb), (out["loss_dep"], l_dep), (out["loss_sem"], l_sem)):
        np.testing.assert_allclose(float(got.detach()), float(want.detach()), rtol=2e-5)
    from test_gpu_parity import _grad_close
    for pa, pb, name in zip(a.parameters(), b.parameters(), ("dir", "base", "head", "sem")):
        if pa.numel():
            _grad_close(pb.grad, pa.grad.cpu(), name, rel=2e-3, cos=0.99999)
    # a one-level estimator of the inner region gives fewer samples: the outer level contributes
    one, _ = _multi_level_estimator(1)
    out1 = RD.fused_forward_backward(H.hip_field(fs), one, rays, pix, dep, lab, bk, stratified=False, **H.RENDER_KW)
    assert out1 is None or out1["n_marched"] < out["n_marched"]
    # scoring
    sc2 = dict(fs); sc2["params"] = H.S.make_field_params(seed=1, log2_hashmap_size=15)
    fields = [b.eval(), H.hip_field(sc2)]
    ests = [est.eval(), _multi_level_estimator(2, seed=8)[0].eval()]
    poses = sc["poses"][[1, 3]].copy(); poses[:, :3] += np.array([3.0, 0.0, 3.0])
    terms_py, score_py = RD.score_views(fields, ests, poses, 640, 640, 320.0, 0.1, 1e-3, 0.025, 0.004, 0.01, DEV)
    terms_c, score_c = RD.score_poses(fields, ests, poses, 640, 640, 320.0, 0.1, 1e-3, 0.025, 0.004, 0.01, DEV)
    assert torch.equal(terms_py, terms_c) and float(score_py) == float(score_c) and np.isfinite(float(score_c))


def test_nan_targets_skip_the_step_through_the_binned_scatter():
    """pipeline.py:520-529: a step whose gradients hold a NaN is dropped.  Here the NaN arrives through the whole fused chain at a batch large enough for the fine
    levels to go through the binned scatter (items keep non-finite values non-finite, the LDS sums are doubles): one NaN target pixel -> the loss and the table
    gradient are non-finite, the guard raises the skip flag on the device, `FusedAdam` leaves parameters, moments and step count untouched; the same batch without
    the NaN then steps normally."""
    from apnrf_amd import render as RD
    from apnrf_amd.optim import FusedAdam
    sc = H.make_scene(log2_hashmap_size=15, seed=6)
    hip, est = H.hip_field(sc).train(), H.hip_estimator(sc).train()
    o, d, pix, dep, lab = (t.to(DEV) for t in _train_batch(sc, 3, 48, 48, 0))
    opt = FusedAdam(hip.parameters(), lr=1e-3, eps=1e-15).bind_field(hip)
    before = [p.detach().clone() for p in hip.parameters()]
    bad = pix.clone(); bad[17, 1] = float("nan")
    bk = torch.tensor([0.3, 0.3, 0.3], device=DEV)
    r = RD.train_step(hip, est, opt, RD.Rays(o, d), bad, dep, lab, bk, step=3, **H.RENDER_KW)
    assert r["n_rendering_samples"] >= 8192                         # the product library uses the bins from 8192 samples
    assert r["skipped"] and not np.isfinite(float(r["loss"]))
    n_mlp = hip.mlp_base.params.numel() - 4 * hip._table_entries()
    assert not bool(torch.isfinite(hip.mlp_base.params.grad[n_mlp:]).all())       # the NaN reached the table gradient (fine levels: through the bins)
    for p, b in zip(hip.parameters(), before):
        assert torch.equal(p.detach(), b)
    assert all(float(opt.state[p]["step"]) == 0 for p in hip.parameters() if p.numel() and p in opt.state)
    r = RD.train_step(hip, est, opt, RD.Rays(o, d), pix, dep, lab, bk, step=5, **H.RENDER_KW)
    assert not r["skipped"] and np.isfinite(float(r["loss"]))
    assert any(not torch.equal(p.detach(), b) for p, b in zip(hip.parameters(), before) if p.numel())
    assert bool(torch.isfinite(hip.mlp_base.params.detach()).all())
