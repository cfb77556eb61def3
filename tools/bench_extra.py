"""The legs of `bench.py --full`: everything that is measured but is not the headline — records go to bench_detail.json only.

  render_extras        the headline views with round 1's random-init weights and with tiny-cuda-nn's fp16 hash blend
  train_extras         host-synchronous and next-batch-presampled forms of the train legs, the reference's dynamic ray-count schedule, the UNCHANGED
                       caller (scripts/pipeline.py:472-532 on the drop-in surface), an ensemble of two stepped side by side
  pose_driver_extras   Dataset.render_image_from_pose / render_probablistic_image_from_pose with their float64 host stacks (pipeline.py:960-974, :697-711)
  cpu_extras           BASELINE.md §4's thread sweep of the headline sample, shape BL-1 (frequency-PE field), one scoring view and a train step through the oracle
"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import bench as B      # noqa: E402  (the harness: Ctx, RenderLeg, TrainLeg, constants)


# ------------------------------------------------------------------ renders
def render_extras(cx, line, leg, scene529, field, est):
    from apnrf_amd import scenes as SC
    args, world = cx.args, cx.world
    V, n_per_view = args.views, leg.n_per_view
    if args.weights != "trained":
        return
    # the same views with round 1's engineered random-init weights, reported beside the headline value, never as it
    leg.field, leg.est = SC.hip_field(scene529, cx.dev), SC.hip_estimator(scene529, cx.dev)
    dtr, sr = leg.run(V, args.steps, 2, False)
    leg.field, leg.est = field, est
    line["render_random_weights"] = {"value": n_per_view * V * world * args.steps / dtr, "unit": "rays/s", "ms_per_step": 1e3 * dtr / args.steps, "views_per_step": V,
                                     "samples_per_ray": sr / (n_per_view * V * args.steps), "samples_per_s": sr * world / dtr,
                                     "note": "synthetic.make_field_params seed 0, procedural occupancy grid: round 1's headline configuration"}
    if args.no_kernel_timing:
        return
    # the same trained weights evaluated with tiny-cuda-nn's fp16 hash blend (mnf_field_config.blend_fp16): never the headline
    bfield = SC.hip_field(scene529, cx.dev, tcnn_blend_fp16=True)
    bfield.load_state_dict(field.state_dict())
    leg.field = bfield.eval()
    dtb, sb = leg.run(V, args.steps, 2, False, n_split=args.render_jobs)
    dtb2, sb2 = leg.run(V, args.steps, 1, True, n_split=1)
    bms, bl = cx.prof("field_render")
    leg.field = field
    line["render_blend_fp16"] = {"value": n_per_view * V * world * args.steps / dtb, "unit": "rays/s", "ms_per_step": 1e3 * dtb / args.steps,
                                 "samples_per_ray": sb / (n_per_view * V * args.steps), "samples_per_s": sb * world / dtb,
                                 "field_kernel_avg_launch_ms": bms / max(bl, 1), "field_kernel_samples_per_launch": sb2 / max(bl, 1),
                                 "field_kernel_frac_of_hbm_peak": B.ALGO_BYTES_PER_SAMPLE * sb2 / max(bms * 1e-3, 1e-9) / 1e9 / B.HBM_PEAK_GBS,
                                 "note": "the headline scene and weights with the hash levels' 8-corner blend as fp16 fused multiply-adds (tcnn's T = __half arithmetic "
                                         "as published); the stand-in was trained with the fp32 blend, so the sample counts differ slightly"}


# ------------------------------------------------------------------ train steps
def _presampled_factory(cx):
    """The batch fetched one iteration early (render.presample): the march of batch i + 1 is enqueued in front of step i and runs beside it on the library's
    side stream; step i adopts the march made in front of step i - 1.  Not across an occupancy refresh (steps 1008 + 16 j): those steps march themselves."""
    from apnrf_amd import render as RD
    from apnrf_amd import scenes as SC

    def factory(tf, te, opt, batches, bkd):
        pre = {"tok": None, "adopted": 0}

        def tstep(i):
            r, pix, dep_, lab = batches[i % 8]
            s_ = 1000 + i
            nxt = RD.presample(tf, te, batches[(i + 1) % 8][0], **SC.RENDER_KW) if s_ % 16 and (s_ + 1) % 16 else None
            tok = pre["tok"]
            out = RD.train_step(tf, te, opt, r, pix, dep_, lab, bkd, step=s_, sync=False, occ_thre=1e-2, presampled=tok, **SC.RENDER_KW)
            pre["adopted"] += int(tok is not None and tok.adopted)
            pre["tok"] = nxt
            return out

        def finish(res, outs, n_all, tf_):
            res["marches_adopted_of_steps"] = [pre["adopted"], n_all]
        return tstep, finish
    return factory


def _dynamic_factory(cx, R_cap, target):
    """scripts/pipeline.py:494-504 without a host round trip: the ray count of the next step follows the latest sample count that has ARRIVED on the host
    (RD.latest_step_counts: one or two steps old), capped at 2000 as the reference caps it; pipeline.py:418 starts the data set at 1024 rays."""
    from apnrf_amd import render as RD
    from apnrf_amd import scenes as SC

    def factory(tf, te, opt, batches, bkd):
        dyn = {"R": 1024, "seen": []}

        def tstep(i):
            r, pix, dep_, lab = batches[i % 8]
            n = dyn["R"]
            out = RD.train_step(tf, te, opt, RD.Rays(r.origins[:n], r.viewdirs[:n]), pix[:n], dep_[:n], lab[:n], bkd, step=1000 + i, sync=False, occ_thre=1e-2, **SC.RENDER_KW)
            dyn["seen"].append(n)
            c = RD.latest_step_counts(tf)
            if c is not None and c[2] > 0:
                dyn["R"] = int(min(R_cap, max(64, c[0] * target / c[2])))
            return out

        def finish(res, outs, n_all, tf_):
            seen = dyn["seen"][-res["steps"]:]
            res.update({"rays_per_step_mean": float(np.mean(seen)), "rays_per_step_min_max": [int(min(seen)), int(max(seen))], "distinct_ray_counts": len(set(seen)),
                        "target_sample_batch_size": target, "overflowed_steps": RD._TRAIN_STATE[id(tf_)].get("overflowed_steps", 0)})
        return tstep, finish
    return factory


def _ensemble_leg(cx, tl, R_, steps):
    """Two ensemble members (the reference trains an ensemble of two, one member after the other inside every iteration: pipeline.py:398-412) stepped in turn on one
    stream and side by side on one stream each (`render.train_step_ensemble`).  Both members start from the stand-in's state."""
    from apnrf_amd import render as RD
    from apnrf_amd import scenes as SC
    torch = cx.torch
    mem = [tl.fresh_member() for _ in range(2)]
    batches = tl.make_batches(R_)
    bkd = torch.rand(3, generator=torch.Generator().manual_seed(7)).to(cx.dev)
    res = {"rays_per_member_step": R_, "members": 2, "steps": steps}

    def turn(i):
        return [RD.train_step(tf, te, opt, *batches[(i + 3 * m) % 8], bkd, step=1000 + i, sync=False, occ_thre=1e-2, **SC.RENDER_KW) for m, (tf, te, opt) in enumerate(mem)]

    def side(i):
        return RD.train_step_ensemble(mem, [tuple(batches[(i + 3 * m) % 8]) + (bkd,) for m in range(2)], step=1000 + i, occ_thre=1e-2, **SC.RENDER_KW)
    pre = {"tok": None}

    def side_presampled(i):
        s_ = 1000 + i
        nxt = [RD.presample(tf, te, batches[(i + 1 + 3 * m) % 8][0], **SC.RENDER_KW) for m, (tf, te, _) in enumerate(mem)] if s_ % 16 and (s_ + 1) % 16 else None
        out = RD.train_step_ensemble(mem, [tuple(batches[(i + 3 * m) % 8]) + (bkd,) for m in range(2)], step=s_, occ_thre=1e-2, presampled=pre["tok"], **SC.RENDER_KW)
        pre["tok"] = nxt
        return out
    for label, fn in (("one_stream", turn), ("stream_per_member", side), ("stream_per_member_next_batch_presampled", side_presampled)):
        outs = []
        dt_e = cx.timed(fn, steps, 6, False, outs.append)
        res[label] = {"ms_per_iteration": 1e3 * dt_e / steps, "ms_per_member_step": 1e3 * dt_e / steps / 2,
                      "rendering_samples_per_member_step": float(np.mean([int(o["n_rendering_samples"]) for pair in outs for o in pair])),
                      "skipped_steps": int(sum(int(o["skipped"]) for pair in outs for o in pair))}
    res["speedup"] = res["one_stream"]["ms_per_iteration"] / res["stream_per_member"]["ms_per_iteration"]
    return res


def _dropin_leg(cx, tl, R_, steps):
    """The UNCHANGED caller: scripts/pipeline.py:472-532 typed against the drop-in names only — `render_image_with_occgrid_with_depth_guide` (autograd), torch losses,
    `loss.backward()`, the per-parameter `torch.isnan` loop with its host round trips, `torch.optim.Adam.step()` and the reference's scheduler — on the same scene,
    start state and batches as `train_refyaml`.  What NOT editing pipeline.py costs against `render.train_step` (one fused C call + FusedAdam)."""
    import torch.nn.functional as F
    from apnrf_amd import nerfacc as NA
    from apnrf_amd import render as RD
    torch = cx.torch
    tf, te, _ = tl.fresh_member(optimizer="torch")
    optimizer = torch.optim.Adam(tf.parameters(), lr=2e-4, eps=1e-15, weight_decay=0.0)                      # pipeline.py:173-178
    scheduler = torch.optim.lr_scheduler.ChainedScheduler([torch.optim.lr_scheduler.CyclicLR(
        optimizer, base_lr=1e-4, max_lr=2e-4, step_size_up=250, mode="exp_range", gamma=1.0, cycle_momentum=False)])   # pipeline.py:183-193's form
    occ_eval_fn = NA.FieldDensityOcc(tf, 1e-3)                                                               # pipeline.py:376-378
    batches = tl.make_batches(R_)
    bkd = torch.rand(3, generator=torch.Generator().manual_seed(7)).to(cx.dev)
    stats = {"n": [], "jumped": 0}

    def step(i):
        rays_, pixels, dep_, sem_ = batches[i % 8]
        te.update_every_n_steps(step=1000 + i, occ_eval_fn=occ_eval_fn, occ_thre=1e-2)
        rgb, acc, depth, semantic, n_rendering_samples = RD.render_image_with_occgrid_with_depth_guide(
            tf, te, rays_, near_plane=0.1, render_step_size=1e-3, render_bkgd=bkd, cone_angle=0.004, alpha_thre=0.01, depth=dep_)
        if n_rendering_samples == 0:
            return None
        loss_rgb = F.smooth_l1_loss(rgb, pixels)
        loss_dep = F.smooth_l1_loss(depth, dep_.unsqueeze(1))
        loss_sem = F.cross_entropy(semantic, sem_)
        loss = loss_rgb * 10 + loss_dep / 5 + loss_sem / 2
        host_losses = (loss_rgb.detach().cpu().item(), loss_dep.detach().cpu().item() / 50, loss_sem.detach().cpu().item() / 2)   # pipeline.py:513-515
        optimizer.zero_grad()
        loss.backward()
        flag = False
        for name, param in tf.named_parameters():
            if param.grad is not None and torch.sum(torch.isnan(param.grad)) > 0:
                flag = True
                break
        if flag:
            optimizer.zero_grad()
            stats["jumped"] += 1
            return None
        optimizer.step()
        scheduler.step()
        stats["n"].append(n_rendering_samples)
        return host_losses
    dt_d = cx.timed(step, steps, max(cx.args.warmup, 4), False)
    return {"ms_per_step": 1e3 * dt_d / steps, "steps": steps, "rays_per_step": R_, "rendering_samples_per_step": float(np.mean(stats["n"][-steps:])),
            "steps_jumped": stats["jumped"], "host_round_trips_per_step": "sample count (inside sampling) + n_rendering_samples + 3 losses + one per parameter vector",
            "what": "pipeline.py:472-532 unchanged on the drop-in surface: autograd route (same kernels call by call), torch smooth_l1 / cross_entropy, loss.backward(), "
                    "per-parameter isnan round trips, torch.optim.Adam + CyclicLR"}


def train_extras(cx, line, tl, tsteps, dtypes):
    args = cx.args
    keep = ("ms_per_step", "rendering_samples_per_step", "marches_adopted_of_steps", "skipped_steps", "host_round_trips_per_step")
    for dt_ in dtypes:
        line["train"][dt_]["host_synchronous"] = {k: v for k, v in tl.run(dt_, args.train_rays, True, tsteps, False).items() if k in keep}
        line["train"][dt_]["next_batch_presampled"] = {k: v for k, v in tl.run(dt_, args.train_rays, False, tsteps, False, _presampled_factory(cx)).items() if k in keep}
    line["train"]["presample"] = ("next_batch_presampled: the same steps with the batch fetched one iteration early and its march (occ_grid.py:181-208: reads rays and grid, "
                                  "not the model) enqueued in front of the current step on a library side stream (render.presample / mnf_train_presample); bit-identical results")
    ry = line["train_refyaml"]
    ry["next_batch_presampled"] = {k: v for k, v in tl.run("f16", 2000, False, max(tsteps, 40), False, _presampled_factory(cx)).items() if k in keep}
    dy = tl.run("f16", 2000, False, max(tsteps, 40), False, _dynamic_factory(cx, 2000, 1 << 18))
    dy["workload"] = ("the reference's own schedule (scripts/pipeline.py:494-504, config_102344250.yaml:3-4): num_rays starts at 1024 and is recomputed after "
                      "every iteration to hold 262 144 samples, capped at 2000; asynchronous steps, the count used is the latest that has arrived on the host")
    line["train_dynamic"] = dy
    cx.log("train: the drop-in surface (autograd route + torch.optim.Adam)")
    dropin = _dropin_leg(cx, tl, 2000, max(tsteps, 20))
    dropin["fused_async_ms_per_step"] = ry["ms_per_step"]
    dropin["fused_host_synchronous_ms_per_step"] = ry["host_synchronous_ms_per_step"]
    dropin["cost_of_not_editing_pipeline_py"] = dropin["ms_per_step"] / ry["host_synchronous_ms_per_step"]
    line["train_dropin"] = dropin
    line["train_ensemble2"] = {"workload": "an ensemble of two members (the reference's), both stepped in every iteration: in turn on one stream (the reference's loop) and side by "
                                           "side, one stream per member (render.train_step_ensemble); asynchronous steps, same scene and start state as the train legs",
                               "refyaml_2000_rays": _ensemble_leg(cx, tl, 2000, max(tsteps, 20)), "config5_8192_rays": _ensemble_leg(cx, tl, args.train_rays, tsteps)}


# ------------------------------------------------------------------ the per-pose drivers as scripts/pipeline.py calls them
def pose_driver_extras(cx, line, f0, e0, f1, e1, poses256):
    from apnrf_amd import render as RD
    from apnrf_amd.dataset import Dataset
    dev = cx.dev
    p1, p40 = poses256[:1], poses256[:40]
    a_full = (f0, e0, p1, 640, 640, 320.0, 0.1, 1e-3, 1, 0.004, 0.01, 1, dev)
    a_40 = lambda f_, e_: (f_, e_, p40, 640, 640, 320.0, 0.1, 1e-3, 0.1, 0.004, 0.01, 4, dev)
    dt_full = cx.timed(lambda i: Dataset.render_image_from_pose(*a_full), 3, 1, False)
    dt_40 = cx.timed(lambda i: [Dataset.render_probablistic_image_from_pose(*a_40(f_, e_)) for f_, e_ in ((f0, e0), (f1, e1))], 3, 1, False)
    dt_40s = cx.timed(lambda i: RD.score_views([f0, f1], [e0, e1], p40, 640, 640, 320.0, 0.1, 1e-3, 0.1, 0.004, 0.01, dev, group=False)[1].item(), 3, 1, False)
    line["render_from_pose"] = {
        "full_view_640x640": {"ms_per_call": 1e3 * dt_full / 3, "rays_per_s": 640 * 640 * 3 / dt_full, "poses": 1,
                              "what": "Dataset.render_image_from_pose as pipeline.py:960-974 calls it (scale 1): -> numpy float64 [1,640,640,.] stacks"},
        "uncertainty_40_poses_x2_members": {"ms_per_trajectory": 1e3 * dt_40 / 3, "rays_per_s": 2 * 40 * 4096 * 3 / dt_40, "poses": 40, "members": 2,
                                            "what": "Dataset.render_probablistic_image_from_pose as pipeline.py:697-711 calls it (scale 0.1, once per ensemble "
                                                    "member): -> six numpy float64 [40,64,64,.] stacks per member, the scorer's numpy then runs on the host"},
        "same_40_poses_on_device_scorer_ms": 1e3 * dt_40s / 3,
        "note": "the second entry against the third is what the host stacks cost a trajectory score: the renders are the same kernels"}


# ------------------------------------------------------------------ CPU baselines beyond the headline sample
def _thread_sweep(fn, unit_count, set_threads, budget_s=25.0):
    """3 warm-ups + 20 timed iterations, median, threads swept over {1, 8, 32, 64}, best reported — bounded in wall time: every thread count gets
    budget_s / 4; a thread count whose first pass is more than 3x slower than the best so far is recorded from that one pass."""
    cores = os.cpu_count() or 1
    sweep, best_t = {}, None
    for th in sorted({1, min(8, cores), min(32, cores), min(64, cores)}):
        set_threads(th)
        t0 = time.perf_counter(); fn(); first = time.perf_counter() - t0
        if best_t is not None and first > 3.0 * best_t:
            sweep[th] = unit_count / first; continue
        n = int(max(0, min(23, (budget_s / 4 - first) / max(first, 1e-4))))
        warm, iters = (3, 20) if n >= 23 else (min(1, max(n - 1, 0)), max(n - 1, 0))
        t = B._timed(fn, warm, iters) if iters > 0 else first
        sweep[th] = unit_count / t
        best_t = t if best_t is None else min(best_t, t)
    best = max(sweep, key=sweep.get)
    set_threads(cores)
    return sweep[best], best, {str(k): v for k, v in sweep.items()}


def cpu_extras(cx, line, scene, pose, width, height, focal):
    import torch
    import torch.nn.functional as F
    from apnrf_amd import scenes as SC
    from oracle import render as R
    from oracle import vanilla as V
    set_threads = B.cpu_threads_setter()
    out = line["cpu_baseline"]
    # the headline sample, threads swept
    orc = B.oracle_field(scene)
    S_ = 24
    idx = R.subsample_indices(width * height, S_ * S_)
    o, d = R.generate_image_rays(R.pose_to_c2w(pose), width, height, focal, idx)
    bk = torch.zeros(3)
    best, th, sweep = _thread_sweep(lambda: R.render_test(1024, orc, scene["occ"], scene["aabb"][None], o, d, render_bkgd=bk, **SC.RENDER_KW), S_ * S_, set_threads, 30.0)
    out["thread_sweep"] = {"best_rays_per_s": best, "threads": th, "sweep_rays_per_s": sweep}
    # BL-1: 64x64 rays x 32 samples, frequency-PE field (numpy)
    rng = np.random.default_rng(0)
    sd = {}

    def lin(name, o_, i_):
        lim = np.sqrt(6.0 / (o_ + i_)); sd[name + ".weight"] = rng.uniform(-lim, lim, (o_, i_)).astype(np.float32); sd[name + ".bias"] = np.zeros(o_, np.float32)
    lin("mlp.base.hidden_layers.0", 64, 63); lin("mlp.base.hidden_layers.1", 64, 64); lin("mlp.sigma_layer.output_layer", 1, 64)
    lin("mlp.bottleneck_layer.output_layer", 64, 64); lin("mlp.rgb_layer.hidden_layers.0", 64, 91); lin("mlp.rgb_layer.output_layer", 3, 64)
    vf = V.VanillaField(sd, net_depth=2, net_depth_condition=1)
    o1, d1 = R.generate_image_rays(torch.eye(4), 64, 64, 32.0)
    o1, d1 = o1.numpy(), d1.numpy()
    edges = np.linspace(0.1, 3.3, 33, dtype=np.float32)
    ts, te = np.broadcast_to(edges[:-1], (4096, 32)), np.broadcast_to(edges[1:], (4096, 32))
    pos = o1[:, None, :] + d1[:, None, :] * ((ts + te) / 2)[..., None]
    cond = np.broadcast_to(d1[:, None, :], pos.shape)

    def bl1():
        rgb, sig = vf.forward(pos.reshape(-1, 3), cond.reshape(-1, 3))
        V.render_batched(rgb.reshape(4096, 32, 3), sig.reshape(4096, 32), ts, te)
    best, th, sweep = _thread_sweep(bl1, 4096, set_threads, 16.0)
    out["bl1_vanilla_64x64x32"] = {"rays_per_s": best, "threads": th, "sweep_rays_per_s": sweep, "what": "oracle/vanilla.py forward + batched compositing (numpy fp32)"}
    # one 2000/8-ray train step through oracle autograd + torch.optim.Adam
    set_threads(int(out["cores"]))
    orc_t = B.oracle_field(scene, requires_grad=True)
    g = torch.Generator().manual_seed(5)
    TR = 250
    idx = torch.randint(0, width * height, (TR,), generator=g).numpy()
    o, d = R.generate_image_rays(R.pose_to_c2w(pose), width, height, focal, idx)
    pix, dep, lab = torch.rand(TR, 3, generator=g), torch.rand(TR, generator=g) * 4, torch.randint(0, 29, (TR,), generator=g)
    opt = torch.optim.Adam([orc_t.p_base, orc_t.p_head, orc_t.p_sem], lr=1e-3, eps=1e-15)
    n_s = {}

    def tstep():
        rr = R.render_train(orc_t, scene["occ"], scene["aabb"][None], 0.05, o, d, torch.full((TR,), 0.1), render_bkgd=bk, render_step_size=1e-3, cone_angle=0.004, alpha_thre=0.01)
        loss = F.smooth_l1_loss(rr[0], pix) * 10 + F.smooth_l1_loss(rr[2], dep.unsqueeze(1)) / 5 + F.cross_entropy(rr[3], lab) / 2
        opt.zero_grad(); loss.backward(); opt.step(); orc_t._derive()
        n_s["n"] = rr[4]
    t = B._timed(tstep, 0, 1)
    out["train_step_refyaml_eighth"] = {"ms": 1e3 * t, "rays": TR, "rendering_samples": int(n_s["n"]), "threads": int(out["cores"]), "ms_scaled_to_2000_rays": 1e3 * t * 2000 / TR,
                                        "sample": f"one train step of {TR} rays (1/8 of the reference yaml's 2000) through oracle autograd + torch.optim.Adam, 1 iteration"}
    set_threads(os.cpu_count() or 1)
