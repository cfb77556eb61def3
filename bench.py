#!/usr/bin/env python3
"""bench.py — BASELINE.json's metric on MI355X: rendered rays/s (RGB + depth + 29-class semantics, 800x800) and train-step ms.

  python bench.py --gpus N --steps K --warmup W

N = 1 runs in this process.  N > 1 without a torch.distributed environment starts
`python -m torch.distributed.run --nproc-per-node N bench.py ...` as a CHILD process (before anything here touches the GPU)
and relays its JSON line; under torch.distributed.run it is one rank per GPU over RCCL.

What one invocation measures (ONE JSON line, rank 0):

  value / ms_per_step   BASELINE config 3: a step = `--views` (default 4) full-resolution 800x800 views of the trained
                        stand-in of Habitat scene 102344529 rendered in ONE batched call of the reference's test-time renderer
                        semantics (perception/models/utils.py:555-779) — rays, weights and occupancy grid resident in HBM.
                        With N GPUs every rank renders its own views (independent units, no data-path collective: weak
                        scaling); value = whole-job rays/s.
  render_views1         the same at one view per call (the reference's own call granularity)
  render_random_weights the same views with round 1's engineered random-init weights: identical work on every box and in every round
  roofline              dominant kernel (fused hash-gather + MLP + compositing): algorithmic bytes / hipEvent time
  train                 BASELINE config 5: train step on scene 102344280, 8192 rays (sampling + density pre-pass + forward +
                        loss + backward + fused Adam), ms per step, per-kernel times and its own roofline
  score256              BASELINE config 4: 256 candidate views x 4096 rays x 2 ensemble members on scene 102344250,
                        probabilistic renders + on-device scorer, views sharded over the ranks, ONE all-gather of the [V,4]
                        terms; with N > 1 rank 0 re-computes all views alone afterwards and the gathered terms must be
                        bit-identical
  cpu_baseline          the oracle (CPU port of the same path) on this box's host cores, bounded samples

Weights: trained stand-ins (`apnrf_amd.standin`): the product's own `train_step` for 2000 iterations on an
analytic target built from the procedural occupancy grid, cached under /tmp.  `--workload` restricts the run to one part.
"""
import argparse
import ctypes
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))

ALGO_BYTES_PER_SAMPLE = 1036          # 16 levels x 8 corners x 8 B hash features + 12 B sample record
TRAIN_BYTES_KEPT, TRAIN_BYTES_MARCHED = 4100, 1024   # §8d: ~4.1 KB per surviving sample + 1 KB per pre-pass sample
HBM_PEAK_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="all", choices=["all", "render800", "score256", "train"])
    ap.add_argument("--views", type=int, default=4, choices=[1, 2, 4, 8],
                    help="render800: 800x800 views per step, rendered in one batched call (the reference renders pose lists, "
                         "habitat_to_data.py:304-549); every view keeps its own per-round sample budget")
    ap.add_argument("--train-rays", type=int, default=8192, help="rays per train step (BASELINE config 5: 8192)")
    ap.add_argument("--standin-steps", type=int, default=2000, help="training iterations of the stand-in scenes (SURVEY 8d: 2000)")
    ap.add_argument("--weights", default="trained", choices=["trained", "random"],
                    help="random = round 1's random-init weights with an engineered density gain (continuity only)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true", help="skip the second, hipEvent-instrumented pass")
    ap.add_argument("--no-views1", action="store_true", help="skip the one-view-per-call pass (profiling runs: every field-kernel launch then belongs to the headline workload)")
    return ap.parse_args()


def spawn_ranks(args):
    """--gpus N > 1 outside torch.distributed.run: start the N ranks as a child job and relay its output.  Nothing in this
    process has touched the GPU yet (never re-exec a process that has)."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd)


# ------------------------------------------------------------------ CPU baselines (the oracle, timed on this box's cores)
def _median_time(fn, warm, iters):
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(iters):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    return float(np.median(ts))


def cpu_baselines(scene, poses, width, height, focal):
    """(i) BASELINE config 1 (64x64 rays x 32 samples, frequency-PE MLP) through oracle/vanilla.py, 3 warm-ups +
    20 timed iterations, median, all cores and one core; (ii) the headline path (hash grid + 128x2 MLP + 29-class head,
    occupancy marching, test-time renderer) through oracle/render.py on a bounded sub-sample of one 800x800 view."""
    import torch
    import helpers as H
    from oracle import render as R
    from oracle import vanilla as V
    try:
        from threadpoolctl import threadpool_limits
    except Exception:                                   # pragma: no cover
        threadpool_limits = None
    cores = os.cpu_count() or 1
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip(); break
    except OSError:
        pass
    out = {"cores": min(cores, 32), "host_cpus": cores, "cpu_model": model, "kind": "port"}
    # (i) BL-1
    rng = np.random.default_rng(0)
    sd = {}
    def lin(name, o, i):
        lim = np.sqrt(6.0 / (o + i)); sd[name + ".weight"] = rng.uniform(-lim, lim, (o, i)).astype(np.float32); sd[name + ".bias"] = np.zeros(o, np.float32)
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
    t_all = _median_time(bl1, 3, 20)
    if threadpool_limits is not None:
        with threadpool_limits(limits=1):
            t_one = _median_time(bl1, 3, 20)
    else:
        t_one = None
    out["bl1_vanilla_64x64x32"] = {"rays_per_s": 4096 / t_all, "ms": 1e3 * t_all, "threads": cores,
                                  "rays_per_s_1thread": None if t_one is None else 4096 / t_one,
                                  "protocol": "3 warm-ups + 20 iterations, median; oracle/vanilla.py forward + batched compositing (numpy fp32)"}
    # (ii) headline path, bounded sample
    S_ = 24
    orc = H.oracle_field(scene)
    idx = R.subsample_indices(width * height, S_ * S_)
    o, d = R.generate_image_rays(R.pose_to_c2w(poses[0]), width, height, focal, idx)

    def headline():
        return R.render_test(1024, orc, scene["occ"], scene["aabb"][None], o, d, render_bkgd=torch.zeros(3), **H.RENDER_KW)
    # the oracle issues thousands of small torch ops per render round: beyond a few dozen threads they only add wake-up
    # cost, so the "all cores" figure uses at most 32 of them (reported as `threads`); the sample is bounded in wall time
    threads = min(cores, 32)
    torch.set_num_threads(threads)
    t0 = time.perf_counter(); headline(); t_first = time.perf_counter() - t0
    iters = int(max(1, min(3, 20.0 / max(t_first, 1e-3))))
    t_all = _median_time(headline, 0, iters) if t_first < 30.0 else t_first
    t_one = None
    if t_all < 15.0:
        torch.set_num_threads(1)
        t_one = _median_time(headline, 0, 1)
    torch.set_num_threads(cores)
    out.update({"value": S_ * S_ / t_all, "unit": "rays/s", "threads": threads, "value_1thread": None if t_one is None else S_ * S_ / t_one,
                "sample": f"{S_}x{S_} linspace sub-sample of one 800x800 view of the same scene and weights, oracle.render.render_test "
                          f"(fp32 torch-CPU hash grid + MLPs + occupancy marching); 1 warm-up + {iters} iteration(s) median on {threads} threads "
                          f"({1e3 * t_all:.0f} ms each)" + ("" if t_one is None else f", 1 iteration on 1 thread ({1e3 * t_one:.0f} ms)")})
    return out


T0 = time.perf_counter()


def log(msg):
    print(f"[bench {time.perf_counter() - T0:7.1f}s] {msg}", file=sys.stderr, flush=True)


def main():
    args = parse()
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(spawn_ranks(args))
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    distributed = world > 1 or ("RANK" in os.environ and "MASTER_PORT" in os.environ)   # launched by torch.distributed.run
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"))
        world = dist.get_world_size()                     # the size RCCL actually formed
    torch.cuda.set_device(local_rank)
    dev = f"cuda:{local_rank}"

    import __graft_entry__ as G
    if rank == 0:
        G.build()
    if distributed:
        dist.barrier()
    import helpers as H
    from apnrf_amd import _lib as L
    from apnrf_amd import render as RD
    from apnrf_amd import standin as SI

    lib = L.load_library()
    want = lambda w: args.workload in ("all", w)
    standin_info = {}

    def scene_model(name, seed=9, steps=None):
        """(scene dict, field, estimator): the trained stand-in (rank 0 trains or loads the cache, the others load it)."""
        scene = H.make_scene(name, n_poses=40)
        if args.weights == "random":
            return scene, H.hip_field(scene, dev), H.hip_estimator(scene, dev)
        steps = args.standin_steps if steps is None else steps
        if rank == 0:
            log(f"stand-in {name} seed {seed}: training / loading")
            field, est, info = SI.train_standin(scene, dev, steps=steps, seed=seed)
            log(f"stand-in {name} seed {seed}: {info}")
            standin_info[f"{name}/seed{seed}"] = {k: info[k] for k in ("steps", "seconds", "loss_first", "loss_last", "skipped_steps",
                                                                      "occupied_cells", "cells", "cached")}
        if distributed:
            dist.barrier()
        if rank != 0:
            field, est, _ = SI.train_standin(scene, dev, steps=steps, seed=seed)        # cache hit
        return scene, field, est

    def timed(step_fn, steps, warmup, with_events, collect=None):
        """W warm-up steps, then exactly K steps bracketed by barrier + synchronize on both sides -> seconds (max over ranks)."""
        for i in range(warmup):
            step_fn(i)
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()
        if with_events:
            L.check(lib.mnf_profile_begin())
        t0 = time.perf_counter()
        for i in range(steps):
            r = step_fn(warmup + i)
            if collect is not None:
                collect(r)
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if with_events:
            ms, n = ctypes.c_double(0), ctypes.c_int64(0)
            L.check(lib.mnf_profile_end(ctypes.byref(ms), ctypes.byref(n)))
        if distributed:
            t = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    def prof(label):
        ms, n = ctypes.c_double(0), ctypes.c_int64(0)
        L.check(lib.mnf_profile_query(label.encode(), ctypes.byref(ms), ctypes.byref(n)))
        return ms.value, n.value

    width = height = 800
    focal = 0.5 * width / np.tan(np.pi / 4)
    line = {"metric": "rendered rays/sec (RGB+depth+semantic, 800x800)", "value": None, "unit": "rays/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": None, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f16", "data": "synthetic",
            "config": {"workload": None, "max_samples": 1024, "render_step_size": 1e-3, "cone_angle": 0.004, "alpha_thre": 0.01,
                       "arithmetic": "fp16 hash entries / weights / activations, fp32 accumulate and outputs",
                       "weights": ("trained stand-in (SURVEY 8d): apnrf_amd.standin.train_standin, the product's train_step on an "
                                   "analytic target (opaque procedural rooms, colour fract(xyz), class = cell hash mod 29), "
                                   f"{args.standin_steps} iterations, FusedAdam lr 2e-3 decayed to 2e-4 over the second half; occupancy grid from update_every_n_steps")
                       if args.weights == "trained" else "random-init (hash U(-0.5,0.5), xavier MLPs, |density row| x 8), procedural occupancy"}}

    # ------------------------------------------------------------------ BASELINE config 3: 800x800 renders (the headline value)
    scene529 = None
    if want("render800"):
        scene529, field, est = scene_model("102344529")
        poses = scene529["poses"][[(5 * k + rank) % 40 for k in range(8)]]       # 8 views of the sweep per rank
        c2w = np.stack([RD.pose_to_c2w(p) for p in poses]).astype(np.float32)
        K = np.array([[focal, 0, width / 2], [0, focal, height / 2], [0, 0, 1.0]])
        rays = RD.generate_image_rays(torch.from_numpy(c2w), width, height, K, dev)
        n_per_view = width * height
        bk = torch.zeros(3)

        process_samples = torch.zeros((), dtype=torch.int64, device=dev)      # every evaluated sample of this process (PMC sums cover all launches)

        def render_pass(V, steps, warmup, with_events):
            batches = [(rays.origins[k:k + V].reshape(-1, 3).contiguous(), rays.viewdirs[k:k + V].reshape(-1, 3).contiguous())
                       for k in range(0, 8, V)]
            evaluated = torch.zeros((), dtype=torch.int64, device=dev)

            def step(i):
                o, d = batches[i % len(batches)]
                r = RD.render_views(field, est, o, d, n_per_view, 1024, render_bkgd=bk, image_hw=(height, width), **H.RENDER_KW)
                process_samples.add_(r["total"][1])
                return r

            def collect(r):
                evaluated.add_(r["total"][1])
            dt = timed(step, steps, warmup, with_events, collect)
            return dt, int(evaluated.item())

        V = args.views
        log("render800: timed pass")
        dt, samples = render_pass(V, args.steps, args.warmup, False)                # the reported value: no instrumentation
        log(f"render800: {1e3 * dt / args.steps:.2f} ms/step, {samples / (n_per_view * V * args.steps):.1f} samples/ray")
        line["value"] = n_per_view * V * world * args.steps / dt
        line["ms_per_step"] = 1e3 * dt / args.steps
        line["config"].update({"workload": f"BASELINE config 3: scene 102344529 (trained synthetic stand-in), 800x800 RGB+depth+29-class "
                                           f"semantic render, {V} view(s) per step in one batched call, hash-grid 16x4 T=2^19 + MLP 128x2 + "
                                           "64x2 heads", "views_per_step": V, "rays_per_step_per_gpu": n_per_view * V,
                               "ms_per_view": 1e3 * dt / args.steps / V, "samples_per_ray": samples / (n_per_view * V * args.steps),
                               "samples_per_s": samples * world / dt,
                               "march_order": "8x8 pixel blocks inside every view (mnf_render_opts.view_order); per-ray results do not depend on it"})
        if not args.no_kernel_timing:
            # second pass of the same K steps with hipEvent pairs around every field-kernel launch (hipEventRecord between
            # dependent launches costs up to ~0.15 ms each on this stack, so it is kept out of the pass that yields `value`)
            _, samples2 = render_pass(V, args.steps, 0, True)
            field_ms, launches = prof("field_render")
            if launches and samples2:
                achieved = ALGO_BYTES_PER_SAMPLE * samples2 / (field_ms * 1e-3) / 1e9
                pmc = None
                pj = os.path.join(REPO, "profiles", "r02_pmc.json")
                if os.path.exists(pj):
                    pmc = json.load(open(pj))["field_kernel"]["hbm_bytes_per_sample"] * samples2 / launches
                line["roofline"] = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                                    "traffic": pmc, "kernel": "mnf::field_kernel<128,2,2,false> (hash gather + MLPs + fused compositing)",
                                    "avg_launch_ms": field_ms / launches, "launches": int(launches), "samples_per_launch": samples2 / launches,
                                    "algorithmic_bytes_per_sample": ALGO_BYTES_PER_SAMPLE,
                                    "algorithmic_bytes_per_launch": ALGO_BYTES_PER_SAMPLE * samples2 / launches,
                                    "field_kernel_share_of_step": field_ms * 1e-3 / dt,
                                    "timing": "second pass of the same K steps, hipEvent pair around each launch on the launch stream; "
                                              "traffic from the committed rocprofv3 --pmc passes (profiles/r02_pmc.json), scaled to this run"}
        if V != 1 and not args.no_views1:
            dt1, s1 = render_pass(1, args.steps, 2, False)
            line["render_views1"] = {"value": n_per_view * world * args.steps / dt1, "unit": "rays/s", "ms_per_view": 1e3 * dt1 / args.steps,
                                     "samples_per_ray": s1 / (n_per_view * args.steps)}
        if args.weights == "trained" and not args.no_views1:
            # the same views with round 1's engineered random-init weights: a workload that is the same on every box and in every
            # round (the trained stand-in is not: training is non-deterministic), reported beside the headline value, never as it
            field_t, est_t = field, est
            field, est = H.hip_field(scene529, dev), H.hip_estimator(scene529, dev)
            dtr, sr = render_pass(V, args.steps, 2, False)
            field, est = field_t, est_t
            line["render_random_weights"] = {"value": n_per_view * V * world * args.steps / dtr, "unit": "rays/s",
                                             "ms_per_step": 1e3 * dtr / args.steps, "views_per_step": V,
                                             "samples_per_ray": sr / (n_per_view * V * args.steps), "samples_per_s": sr * world / dtr,
                                             "note": "deterministic workload (synthetic.make_field_params seed 0, procedural occupancy grid): "
                                                     "round 1's headline configuration, for comparisons across boxes and rounds"}
        line["samples"] = {"timed": int(samples), "process_total": int(process_samples.item())}
        del rays

    # ------------------------------------------------------------------ BASELINE config 5: train step
    if want("train"):
        from apnrf_amd.optim import FusedAdam
        scene280, tfield, test_ = scene_model("102344280", seed=11)
        tfield.train(); test_.train()
        opt = FusedAdam(tfield.parameters(), lr=1e-3, eps=1e-15)
        R_ = args.train_rays
        proc = SI._procedural_estimator(scene280, dev)
        c2w = np.stack([RD.pose_to_c2w(p) for p in scene280["poses"][:8]]).astype(np.float32)
        K6 = np.array([[320.0, 0, 320], [0, 320.0, 320], [0, 0, 1.0]])
        g = torch.Generator(device="cpu").manual_seed(100 + rank)
        batches = []
        for k in range(8):
            idx = torch.randint(0, 640 * 640, (R_,), generator=g).numpy()
            ys, xs = idx // 640, idx % 640                      # grouped by 32x32 image block, as dataset.Dataset.fetch_data does
            idx = idx[np.argsort((ys // 32) * 20 + xs // 32, kind="stable")]
            r = RD.generate_image_rays(torch.from_numpy(c2w[k:k + 1]), 640, 640, K6, dev, idx)
            batches.append((r,) + SI.analytic_targets(proc, scene280["aabb"], r.origins, r.viewdirs))
        counts = {"kept": 0, "marched": 0, "steps": 0}

        def tstep(i):
            r, pix, dep_, lab = batches[i % 8]
            out = RD.train_step(tfield, test_, opt, r, pix, dep_, lab, torch.rand(3, device=dev), step=1000 + i, **H.RENDER_KW)
            return out

        def tcollect(out):
            counts["kept"] += out["n_rendering_samples"]; counts["marched"] += test_.last_sampling["n_marched"]; counts["steps"] += 1
        tsteps = max(args.steps, 10)
        log("train: timed pass")
        dt_t = timed(tstep, tsteps, max(args.warmup, 3), False, tcollect)
        log(f"train: {1e3 * dt_t / tsteps:.2f} ms/step")
        kept, marched = counts["kept"] / counts["steps"], counts["marched"] / counts["steps"]
        train = {"ms_per_step": 1e3 * dt_t / tsteps, "steps": tsteps, "rays_per_step": R_, "rendering_samples_per_step": kept,
                 "marched_samples_per_step": marched,
                 "workload": "BASELINE config 5 shape: scene 102344280 (trained stand-in, training continued), 8192-ray batches of one "
                             "640x640 view, occupancy sampling + density pre-pass + differentiable render + loss (pipeline.py:506-511) + "
                             "backward + NaN guard + FusedAdam; occupancy refresh every 16th step"}
        if not args.no_kernel_timing:
            counts.update(kept=0, marched=0, steps=0)
            dt_e = timed(tstep, tsteps, 0, True, tcollect)
            kept_e, marched_e = counts["kept"] / counts["steps"], counts["marched"] / counts["steps"]
            kernels, total_ms = {}, 0.0
            per_sample = {"field_density": ("marched", 1036), "field_train_forward": ("kept", 1036), "hash_scatter": ("kept", 2048)}
            for label in ("sample_rays", "field_density", "field_train_forward", "composite_train_forward", "composite_train_backward",
                          "dgrad", "wgrad", "hash_scatter"):
                ms, n = prof(label)
                if not n:
                    continue
                e = {"ms_per_step": ms / tsteps, "launches_per_step": n / tsteps}
                if label in per_sample:
                    which, b = per_sample[label]
                    nbytes = b * (kept_e if which == "kept" else marched_e)
                    gbs = nbytes / (ms / tsteps * 1e-3) / 1e9
                    e.update({"algorithmic_bytes_per_step": nbytes, "achieved_GBps": gbs, "frac_of_hbm_peak": gbs / HBM_PEAK_GBS})
                kernels[label] = e
                total_ms += ms / tsteps
            algo = TRAIN_BYTES_KEPT * kept_e + TRAIN_BYTES_MARCHED * marched_e
            step_ms = 1e3 * dt_t / tsteps
            train["kernels"] = kernels
            train["roofline"] = {"bound": "hbm", "achieved": algo / (step_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                 "frac": algo / (step_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None,
                                 "algorithmic_bytes_per_step": algo,
                                 "definition": "(4.1 KB x surviving samples + 1 KB x pre-pass samples) / un-instrumented step time (SURVEY 8d)",
                                 "timed_kernels_ms_per_step": total_ms, "instrumented_step_ms": 1e3 * dt_e / tsteps}
        line["train"] = train
        if not want("render800"):
            line.update({"metric": "train-step ms", "value": train["ms_per_step"], "unit": "ms", "higher_is_better": False,
                         "ms_per_step": train["ms_per_step"]})
            line["config"]["workload"] = train["workload"]
        del tfield, test_, opt, batches

    # ------------------------------------------------------------------ BASELINE config 4: candidate-view scoring, views sharded over ranks
    if want("score256"):
        scene250, f0, e0 = scene_model("102344250", seed=9)
        _, f1, e1 = scene_model("102344250", seed=10)
        poses256 = SI._free_space_poses(scene250, 256, seed=9)                  # 8 trajectories x 32 views inside the free space
        group = dist.group.WORLD if distributed else None

        def sstep(i):
            return RD.score_views([f0, f1], [e0, e1], poses256, 640, 640, 320.0, 0.1, 1e-3, 0.1, 0.004, 0.01, dev, group=group)
        ssteps = max(3, min(args.steps, 5))
        log("score256: timed pass")
        dt_s = timed(sstep, ssteps, 1, False)
        log(f"score256: {1e3 * dt_s / ssteps:.2f} ms/pass")
        terms, score = sstep(0)
        evaluated = float(sum(int(t[1]) for t in RD.LAST_SCORE_TOTALS))          # this rank's share of the views, all members
        sc = {"ms_per_pass": 1e3 * dt_s / ssteps, "rays_per_s": 256 * 4096 * 2 * ssteps / dt_s, "views": 256, "rays_per_view": 4096,
              "ensemble_members": 2, "n_gpus": world, "scaling": "strong", "score": float(score),
              "samples_per_ray_rank0": evaluated / max(1, (256 // world) * 4096 * 2),
              "samples_per_s_rank0": evaluated * ssteps / dt_s,
              "collective": "one all_gather_into_tensor of [V/N,4] float64 per pass" if world > 1 else "none (single rank)",
              "workload": "BASELINE config 4: scene 102344250 (two trained stand-ins: seeds 9 and 10, the same protocol as the render scene), 256 candidate poses in free "
                          "space, 64x64 rays each (linspace sub-sample of 640x640), probabilistic render + predictive-information terms"}
        if world > 1:
            full, _ = RD.score_views([f0, f1], [e0, e1], poses256, 640, 640, 320.0, 0.1, 1e-3, 0.1, 0.004, 0.01, dev, group=False)
            same = bool(torch.equal(full, terms))
            sc["bit_identical_to_single_gpu"] = same
            if not same:
                raise SystemExit("score256: gathered terms differ from the single-rank computation")
        line["score256"] = sc
        if not want("render800") and not want("train"):
            line.update({"metric": "candidate-view scoring rays/s", "value": sc["rays_per_s"], "ms_per_step": sc["ms_per_pass"], "scaling": "strong"})
            line["config"]["workload"] = sc["workload"]

    if rank == 0:
        if standin_info:
            line["config"]["standin_training"] = standin_info
        if world == 1 and not args.no_cpu_baseline and scene529 is not None:
            base_scene = dict(scene529)
            base_scene["params"] = {"mlp_base": field.mlp_base.params.detach().cpu().numpy(), "mlp_head": field.mlp_head.params.detach().cpu().numpy(),
                                    "mlp_sem": field.mlp_sem.params.detach().cpu().numpy()}
            base_scene["occ"] = est.binaries.cpu().numpy()
            log("cpu baseline")
            line["cpu_baseline"] = cpu_baselines(base_scene, scene529["poses"][[0]], width, height, focal)
            log("cpu baseline done")
        print(json.dumps(line))
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
