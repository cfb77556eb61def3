#!/usr/bin/env python3
"""bench.py — headline metric of BASELINE.json: rendered rays/s (RGB + depth + 29-class semantics).

  python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run)

A step renders one full-resolution 800x800 view (640 000 rays) of the synthetic stand-in for Habitat scene
102344529 (BASELINE config 3: occupancy-grid ray marching, hash-grid field with the reference-yaml MLP shape
128x2 + 64x2 heads, 29 classes) through the reference's test-time renderer semantics
(perception/models/utils.py:555-779) on libmi355nerf.so.  Rays, weights and the occupancy grid are resident in
HBM before the timed region.  With N GPUs every rank renders its own views (the path shards over independent
views with no data-path collective: weak scaling); `value` is the whole-job rays/s.

`--workload score256` instead times BASELINE config 4 (256 candidate views x 4096 rays, two ensemble members,
probabilistic renders + on-device scorer, views sharded over ranks with one all-gather of the [V,4] terms).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))

ALGO_BYTES_PER_SAMPLE = 1036   # 16 levels x 8 corners x 8 B hash features + 12 B sample record


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="render800", choices=["render800", "score256", "train"])
    ap.add_argument("--views", type=int, default=4, choices=[1, 2, 4, 8],
                    help="render800: 800x800 views per step, rendered in one batched call (the reference renders pose lists, "
                         "habitat_to_data.py:304-549); every view keeps its own per-round sample budget")
    ap.add_argument("--train-rays", type=int, default=8192, help="rays per train step (BASELINE config 5: 8192)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true", help="do not bracket field-kernel launches with hipEvents")
    ap.add_argument("--cpu-sample", type=int, default=80, help="cpu baseline renders a SxS sub-sample of one view")
    return ap.parse_args()


def cpu_baseline(scene, poses, width, height, focal, S):
    """The oracle (a CPU port of the reference path, kind='port') timed on this box's host cores on a bounded
    sample of the same workload: an SxS linspace sub-sample of the first 800x800 view."""
    import helpers as H
    from oracle import render as R
    orc = H.oracle_field(scene)
    idx = R.subsample_indices(width * height, S * S)
    o, d = R.generate_image_rays(R.pose_to_c2w(poses[0]), width, height, focal, idx)
    t0 = time.perf_counter()
    out = R.render_test(1024, orc, scene["occ"], scene["aabb"][None], o, d, render_bkgd=torch.zeros(3), **H.RENDER_KW)
    dt = time.perf_counter() - t0
    return {"value": S * S / dt, "unit": "rays/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{S}x{S} linspace sub-sample of one 800x800 view ({out['total_samples']} kept samples), "
                      f"oracle.render.render_test, fp32 torch-CPU, {dt:.1f} s"}


def pmc_traffic(samples_per_launch):
    """HBM bytes per launch of the field kernel from the committed rocprofv3 --pmc passes (profiles/r01_pmc.json:
    FETCH_SIZE and WRITE_SIZE collected in separate passes, KB units, summed over launches), scaled to this run's
    samples per launch.  PMC counters cannot be collected from inside this process, so this is read, not measured live;
    None when the file is absent."""
    path = os.path.join(REPO, "profiles", "r01_pmc.json")
    if not os.path.exists(path):
        return None
    p = json.load(open(path))["field_kernel"]
    return p["hbm_bytes_per_sample"] * samples_per_launch


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch.distributed as dist
    distributed = world > 1 or ("RANK" in os.environ and "MASTER_PORT" in os.environ)   # launched by torch.distributed.run
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"))
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

    lib = L.load_library()
    width = height = 800
    focal = 0.5 * width / np.tan(np.pi / 4)
    if args.workload == "render800":
        scene = H.make_scene("102344529", n_poses=8 * max(world, 1))
        field, est = H.hip_field(scene, dev), H.hip_estimator(scene, dev)
        my_poses = scene["poses"][rank::world][:8]
        c2w = np.stack([RD.pose_to_c2w(p) for p in my_poses]).astype(np.float32)
        K = np.array([[focal, 0, width / 2], [0, focal, height / 2], [0, 0, 1.0]])
        rays = RD.generate_image_rays(torch.from_numpy(c2w), width, height, K, dev)
        V = args.views
        n_per_view = width * height
        bk = torch.zeros(3)
        view_batches = [(rays.origins[k:k + V].reshape(-1, 3).contiguous(), rays.viewdirs[k:k + V].reshape(-1, 3).contiguous())
                        for k in range(0, 8, V)]

        def step(i):
            o, d = view_batches[i % len(view_batches)]
            return RD.render_views(field, est, o, d, n_per_view, 1024, render_bkgd=bk, image_hw=(height, width), **H.RENDER_KW)
        units_per_step = n_per_view * V
        workload = (f"scene 102344529 (synthetic stand-in), 800x800 RGB+depth+29-class semantic render, {V} view(s) per step in one "
                    "batched call, hash-grid 16x4 T=2^19 + MLP 128x2 + 64x2 heads")
    elif args.workload == "train":
        # BASELINE config 5 shape: scene 102344280, 8192-ray train batches; targets are synthetic (no Habitat data offline)
        scene = H.make_scene("102344280", n_poses=8)
        field, est = H.hip_field(scene, dev), H.hip_estimator(scene, dev)
        from apnrf_amd.optim import FusedAdam
        opt = FusedAdam(field.parameters(), lr=1e-3, eps=1e-15)     # torch.optim.Adam's update as one kernel per parameter
        R_ = args.train_rays
        c2w = np.stack([RD.pose_to_c2w(p) for p in scene["poses"]]).astype(np.float32)
        K = np.array([[320.0, 0, 320], [0, 320.0, 320], [0, 0, 1.0]])
        g = torch.Generator(device="cpu").manual_seed(rank)
        batches = []
        for k in range(8):
            idx = torch.randint(0, 640 * 640, (R_,), generator=g).numpy()
            ys, xs = idx // 640, idx % 640                      # grouped by 32x32 image block, as dataset.Dataset.fetch_data does
            idx = idx[np.argsort((ys // 32) * 20 + xs // 32, kind="stable")]
            r = RD.generate_image_rays(torch.from_numpy(c2w[k:k + 1]), 640, 640, K, dev, idx)
            batches.append((r, torch.rand(R_, 3, generator=g).to(dev), (torch.rand(R_, generator=g) * 4 + 0.5).to(dev),
                            torch.randint(0, 29, (R_,), generator=g).to(dev)))
        state = {"samples": 0}

        def step(i):
            r, pix, dep_, lab = batches[i % 8]
            out = RD.train_step(field, est, opt, r, pix, dep_, lab, torch.rand(3, device=dev), step=i, **H.RENDER_KW)
            state["samples"] += out["n_rendering_samples"]
            return None
        units_per_step = R_
        workload = f"train step, scene 102344280 (synthetic stand-in), {R_} rays/step, hash-grid + MLP 128x2, loss (torch) + backward + fused Adam"
    else:
        scene = H.make_scene("102344250", n_poses=256)
        sc2 = dict(scene); sc2["params"] = H.S.make_field_params(seed=1)
        fields = [H.hip_field(scene, dev), H.hip_field(sc2, dev)]
        ests = [H.hip_estimator(scene, dev), H.hip_estimator(scene, dev)]

        def step(i):
            return RD.score_views(fields, ests, scene["poses"], 640, 640, 320.0, 0.1, 1e-3, 0.1, 0.004, 0.01, dev)
        units_per_step = 256 * 4096 * 2 / world   # rays rendered per rank per step (2 ensemble members)
        workload = "256 candidate views x 4096 rays x 2 ensemble members, probabilistic render + predictive-information scorer"

    def timed_pass(with_events):
        """K steps bracketed by barrier + synchronize on both sides; returns (seconds, evaluated samples, field ms, launches)."""
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()
        if with_events:
            L.check(lib.mnf_profile_begin())
        evaluated = torch.zeros((), dtype=torch.int64, device=dev)
        t0 = time.perf_counter()
        for i in range(args.steps):
            out = step(args.warmup + i)
            if isinstance(out, dict):
                evaluated += out["total"][1]
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        field_ms, launches = ctypes.c_double(0), ctypes.c_int64(0)
        if with_events:
            L.check(lib.mnf_profile_end(ctypes.byref(field_ms), ctypes.byref(launches)))
        return dt, int(evaluated.item()), field_ms.value, launches.value

    import ctypes
    warm = torch.zeros((), dtype=torch.int64, device=dev)
    for i in range(args.warmup):
        out = step(i)
        if isinstance(out, dict):
            warm += out["total"][1]     # also warms up the tiny torch ops the timed loop uses
    # Pass 1 (the reported value): exactly K steps, no instrumentation.
    dt, samples, _, _ = timed_pass(False)
    # Pass 2 (roofline only): the same K steps again with every field-kernel launch bracketed by a hipEvent pair on
    # the launch stream.  Kept out of pass 1 because hipEventRecord between dependent launches costs up to ~0.15 ms
    # each on this stack (+45 % step time); the per-kernel durations agree with the rocprofv3 trace (profiles/).
    field_ms = launches = 0
    if not args.no_kernel_timing:
        _, samples, field_ms, launches = timed_pass(True)
    if distributed:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)     # the slowest rank defines the step time
        dt = float(t.item())

    if rank == 0:
        value = units_per_step * world * args.steps / dt
        line = {
            "metric": "rendered rays/sec (RGB+depth+semantic)", "value": value, "unit": "rays/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f16 operands / f32 accumulate", "data": "synthetic",
            "config": {"workload": workload, "rays_per_step_per_gpu": int(units_per_step), "max_samples": 1024,
                       "render_step_size": 1e-3, "cone_angle": 0.004, "alpha_thre": 0.01,
                       "weights": "random-init (hash U(-0.5,0.5), xavier MLPs, |density row| x 8), procedural occupancy"},
        }
        if args.workload == "render800":
            # evaluated samples of the timed steps and of the warm-up (PMC sums cover both): tools/reduce_pmc.py
            line["samples"] = {"timed": int(samples), "warmup": int(warm.item())}
            line["config"]["views_per_step"] = args.views
            line["config"]["march_order"] = "8x8 pixel blocks inside every view (mnf_render_opts.view_order); per-ray results do not depend on it"
            line["config"]["ms_per_view"] = 1e3 * dt / args.steps / args.views
        if launches and samples:
            achieved = ALGO_BYTES_PER_SAMPLE * samples / (field_ms * 1e-3) / 1e9
            line["roofline"] = {"bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0,
                                "traffic": pmc_traffic(samples / launches), "kernel": "mnf::field_kernel<128,2,2,false>",
                                "avg_launch_ms": field_ms / launches, "launches": int(launches),
                                "samples_per_launch": samples / launches, "algorithmic_bytes_per_launch": ALGO_BYTES_PER_SAMPLE * samples / launches,
                                "algorithmic_bytes_per_sample": ALGO_BYTES_PER_SAMPLE,
                                "samples_per_ray": samples / (units_per_step * args.steps),
                                "field_kernel_share_of_step": field_ms * 1e-3 / dt,
                                "timing": "second pass of the same K steps with hipEvent pairs around each launch"}
        if args.workload == "train":
            line["metric"] = "train-step ms"
            line["train"] = {"ms_per_step": 1e3 * dt / args.steps, "rays_per_step": args.train_rays,
                             "rendering_samples_per_step": state["samples"] / max(1, 2 * args.steps + args.warmup) }
        if world == 1 and not args.no_cpu_baseline and args.workload == "render800":
            line["cpu_baseline"] = cpu_baseline(scene, scene["poses"], width, height, focal, args.cpu_sample)
        print(json.dumps(line))
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
