#!/usr/bin/env python3
"""bench.py — BASELINE.json's metric on MI355X: rendered rays/s (RGB + depth + 29-class semantics, 800x800) and train-step ms.

  python bench.py --gpus N --steps K --warmup W

N = 1 runs in this process.  N > 1 without a torch.distributed environment starts
`python -m torch.distributed.run --nproc-per-node N bench.py ...` as a CHILD process (before anything here touches the GPU)
and relays its output; under torch.distributed.run it is one rank per GPU over RCCL.

Output.  Rank 0 prints ONE JSON line of < 4 KB on stdout (`headline_line`: metric, value, config.workload, dtype, `roofline`,
`cpu_baseline`, scalar summaries of the other legs) and writes everything else — every leg's full record — to
`bench_detail.json` next to this file (`--detail-file`); progress goes to stderr.

Legs of the default run (`--workload all`):

  value / ms_per_step   BASELINE config 3: a step = `--views` (default 4) full-resolution 800x800 views of the trained
                        stand-in of Habitat scene 102344529 rendered in ONE batched call of the reference's test-time renderer
                        semantics (perception/models/utils.py:555-779) — rays, weights and occupancy grid resident in HBM.
                        With N GPUs every rank renders its own views (independent units, no data-path collective: weak
                        scaling); value = whole-job rays/s.
  roofline              the dominant kernel (fused hash gather + MLPs + compositing): algorithmic bytes / hipEvent time of its
                        launches in a second pass of the same steps with ONE render job in flight (the timed pass runs two
                        jobs side by side, whose launches overlap: a per-launch duration is only meaningful for the serial form)
  render_views1         the headline workload at one view per call (the reference's own call granularity)
  train                 BASELINE config 5: train step on scene 102344280, 8192 rays — fp16 (the reference's tcnn arithmetic) AND
                        bf16 matrix-core operands, without host round trips (`sync=False`); per-kernel times and rooflines;
                        `train_refyaml`: the reference yaml's own shape (2000 rays: scripts/config_102344250.yaml:3-4, pipeline.py:494-504)
  config2               BASELINE config 2: scene 102344250 with the reference class default 4x64 base MLP (ngp.py:77-78), 256x256 views:
                        render (8 views per call) and train step (2000 rays), per-kernel times and the 8d fraction
  score256              BASELINE config 4: 256 candidate views x 4096 rays x 2 ensemble members on scene 102344250,
                        probabilistic renders + on-device scorer, views sharded over the ranks, ONE all-gather of the [V,4]
                        terms; with N > 1 rank 0 re-computes all views alone and the gathered terms must be bit-identical.
                        `score256_shard8`: 32 views on this GPU = one rank's share of an 8-GPU run
  cpu_baseline          the oracle (CPU port of the same path) on this box's host cores, on a bounded sample of the headline workload
                        (24x24 rays of one benchmark view, <= ~12 s), and `bench_parity`: the HIP render of the same rays against it.
                        The run FAILS (exit code 3) above the north-star tolerance.

`--full` adds the legs of tools/bench_extra.py (the drop-in surface as the reference calls it, ensemble steps, presampled and
dynamic-schedule train steps, random-weight / fp16-blend renders, the BL-1 and per-shape CPU baselines).

Weights: trained stand-ins (`apnrf_amd.standin`): the product's own `train_step` for 2000 iterations on an
analytic target built from the procedural occupancy grid — bitwise reproducible (seeded draws, deterministic gradient
accumulation), so the scenes are the same on every box.  `--workload` restricts the run to one part.
"""
import argparse
import gc
import ctypes
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

ALGO_BYTES_PER_SAMPLE = 1036          # 16 levels x 8 corners x 8 B hash features + 12 B sample record
TRAIN_BYTES_KEPT, TRAIN_BYTES_MARCHED = 4100, 1024   # §8d: ~4.1 KB per surviving sample + 1 KB per pre-pass sample
HBM_PEAK_GBS = 8000.0
FIELD_SOURCES = ("field.hip", "field_dev.h", "composite_dev.h", "field.h", "common.h")
TRAIN_SOURCES = ("train.hip", "trainstep.hip", "composite_train.hip", "field.hip", "field_dev.h", "field.h", "common.h", "march.hip", "march_dev.h")
MAX_LINE_BYTES = 4096
PMC_ROUNDS = ("r06", "r05")           # profiles/<round>_pmc*.json, newest first; a profile is quoted only for the kernel sources it was measured on (md5)


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="all", choices=["all", "render800", "score256", "train", "config2"])
    ap.add_argument("--views", type=int, default=4, choices=[1, 2, 4, 8],
                    help="render800: 800x800 views per step, rendered in one batched call (the reference renders pose lists, "
                         "habitat_to_data.py:304-549); every view keeps its own per-round sample budget")
    ap.add_argument("--train-rays", type=int, default=8192, help="rays per train step (BASELINE config 5: 8192)")
    ap.add_argument("--standin-steps", type=int, default=2000, help="training iterations of the stand-in scenes (SURVEY 8d: 2000)")
    ap.add_argument("--weights", default="trained", choices=["trained", "random"],
                    help="random = round 1's random-init weights with an engineered density gain (continuity only)")
    ap.add_argument("--full", action="store_true", help="also run the legs of tools/bench_extra.py (detail file only)")
    ap.add_argument("--detail-file", default=os.path.join(REPO, "bench_detail.json"))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true", help="skip the second, hipEvent-instrumented passes")
    ap.add_argument("--no-views1", action="store_true", help="skip the one-view-per-call pass (profiling runs)")
    ap.add_argument("--render-jobs", type=int, default=2, choices=[1, 2, 4],
                    help="render jobs in flight in the timed 800x800 pass (2 = the reported configuration; 1 = launches back to back on one stream, the form "
                         "the roofline pass uses: a rocprofv3 --stats run with 1 gives per-launch durations that can be compared with roofline.avg_launch_ms)")
    ap.add_argument("--train-dtypes", default="f16,bf16", help="matrix-core operand types of the train leg")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"], help="torch.distributed backend of an N > 1 job (nccl = RCCL; gloo: test rigs)")
    ap.add_argument("--one-device", action="store_true", help="test rigs: every rank uses cuda:0 (two ranks on a one-GPU box; RCCL refuses that, so with --dist-backend gloo)")
    return ap.parse_args(argv)


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


# ------------------------------------------------------------------ the one stdout line
def _short(x, digits=6):
    """floats to `digits` significant digits (the line is a summary; bench_detail.json holds full precision)"""
    if isinstance(x, float):
        return float(f"{x:.{digits}g}") if np.isfinite(x) else None
    if isinstance(x, dict):
        return {k: _short(v, digits) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_short(v, digits) for v in x]
    return x


def headline_line(d, detail_file="bench_detail.json"):
    """The single JSON line the driver parses, built from the full record `d`: the contract's fields, `roofline` and `cpu_baseline` whole
    (minus their prose), scalar summaries of the other legs.  Always below MAX_LINE_BYTES (asserted): round 5's 21.6 KB line was not parsed."""
    get = lambda *ks: _dig(d, ks)
    line = {k: d.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                  "vs_baseline", "dtype", "data")}
    cfg = d.get("config", {})
    line["config"] = {k: cfg[k] for k in ("workload", "views_per_step", "rays_per_step_per_gpu", "samples_per_ray", "samples_per_s", "max_samples", "near_plane",
                                          "render_step_size", "cone_angle", "alpha_thre", "weights", "render_jobs_in_flight") if k in cfg}
    for k in ("workload", "weights"):
        if isinstance(line["config"].get(k), str):
            line["config"][k] = line["config"][k][:320]
    r = d.get("roofline")
    if r:
        line["roofline"] = {k: r[k] for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "avg_launch_ms", "launches", "samples_per_launch",
                                              "algorithmic_bytes_per_sample", "field_kernel_share_of_serial_step") if k in r}
    b = d.get("cpu_baseline")
    if b:
        line["cpu_baseline"] = {k: b[k] for k in ("value", "unit", "cores", "kind", "cpu_model", "host_cpus", "seconds") if k in b}
        line["cpu_baseline"]["sample"] = str(b.get("sample", ""))[:260]
    scal = {
        "render_views1_rays_per_s": get("render_views1", "value"),
        "train_ms": get("train", "ms_per_step"), "train_bf16_ms": get("train", "bf16", "ms_per_step"),
        "train_samples": get("train", "rendering_samples_per_step"),
        "train_frac": get("train", "roofline", "frac"), "train_traffic_over_algorithmic": get("train", "roofline", "traffic_over_algorithmic"),
        "train_refyaml_ms": get("train_refyaml", "ms_per_step"), "train_refyaml_frac": get("train_refyaml", "roofline", "frac"),
        "config2_render_ms_per_view": get("config2", "render", "ms_per_view"), "config2_render_rays_per_s": get("config2", "render", "rays_per_s"),
        "config2_field_frac": get("config2", "render", "roofline", "frac"),
        "config2_train_ms": get("config2", "train", "ms_per_step"), "config2_train_frac": get("config2", "train", "roofline", "frac"),
        "score256_ms": get("score256", "ms_per_pass"), "score256_rays_per_s": get("score256", "rays_per_s"),
        "score256_shard8_ms": get("score256_shard8", "ms_per_pass"), "score256_bit_identical_to_single_gpu": get("score256", "bit_identical_to_single_gpu"),
        "bench_parity_ok": get("bench_parity", "ok"), "bench_parity_max_abs": get("bench_parity", "max_abs"),
        "bench_parity_psnr_db": get("bench_parity", "psnr_db"),
    }
    line.update({k: v for k, v in scal.items() if v is not None})
    line["detail"] = os.path.basename(detail_file) if detail_file else None
    line = _short(line)
    s = json.dumps(line)
    assert len(s.encode()) < MAX_LINE_BYTES, f"bench line is {len(s.encode())} bytes (limit {MAX_LINE_BYTES})"
    return s


def _dig(d, ks):
    for k in ks:
        if not isinstance(d, dict) or k not in d:
            return None
        d = d[k]
    return d


# ------------------------------------------------------------------ counter traffic from committed profiles
def _md5_of(files):
    h = hashlib.md5()
    for f in files:
        with open(os.path.join(REPO, "active-perception-using-neural-radiance-fields_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:12]


def field_source_id():
    """Hash of the sources of the dominant kernel: PMC traffic measured on another build is not quoted for this one."""
    return _md5_of(FIELD_SOURCES)


def field_traffic_per_sample(key="field_kernel"):
    """HBM bytes per evaluated sample of the render field kernel from the newest committed counter passes of THIS kernel build -> (value or None, note)"""
    for rnd in PMC_ROUNDS:
        pj = os.path.join(REPO, "profiles", f"{rnd}_pmc.json")
        if not os.path.exists(pj):
            continue
        pm = json.load(open(pj))
        if pm.get("field_sources_md5") != field_source_id():
            return None, f"profiles/{rnd}_pmc.json was measured on a different build of the kernel sources: not quoted"
        if key not in pm:
            continue
        return pm[key]["hbm_bytes_per_sample"], (f"rocprofv3 --pmc passes of this kernel build (profiles/{rnd}_pmc.json: FETCH_SIZE + WRITE_SIZE per evaluated sample), "
                                                "scaled to this run's samples per launch")
    return None, "no PMC profile of this kernel build is committed"


def train_traffic(rays, dtype="f16", model="128x2"):
    """HBM bytes of one train step from the committed counter passes (profiles/rNN_pmc_train*.json: rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE, tools/pmc_train.sh)
    — quoted only for the kernel sources they were measured on (md5) and the shape they were measured at.  -> (bytes or None, note)"""
    suffix = ("" if dtype == "f16" else "_" + dtype) + ("" if model == "128x2" else "_" + model)
    for rnd in PMC_ROUNDS:
        cands = [os.path.join(REPO, "profiles", f"{rnd}_pmc_train{suffix}_{int(rays)}.json"), os.path.join(REPO, "profiles", f"{rnd}_pmc_train{suffix}.json")]
        for pj in cands:
            if not os.path.exists(pj):
                continue
            pm = json.load(open(pj))
            if pm.get("train_sources_md5") != _md5_of(TRAIN_SOURCES):
                return None, "profiles/" + os.path.basename(pj) + " was measured on a different build of the train kernels: not quoted"
            if int(pm.get("rays_per_step", 0)) != int(rays):
                continue
            return pm["per_step"]["hbm_bytes"], ("rocprofv3 --pmc passes of these kernel sources (profiles/" + os.path.basename(pj) + ": FETCH_SIZE + WRITE_SIZE per train "
                                                "step at %s surviving samples; copied from the profile, not measured in this run)" % pm.get("surviving_samples_per_step"))
    return None, "no PMC profile of the train step at this shape is committed"


# ------------------------------------------------------------------ CPU baseline (the oracle, timed on this box's cores) + parity of the benchmarked scene
def _timed(fn, warm, iters):
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(iters):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    return float(np.median(ts))


def oracle_field(scene, requires_grad=False, accum="whole"):
    from oracle.field import FieldConfig, OracleField
    cfg = FieldConfig(aabb=tuple(float(x) for x in scene["aabb"]), neurons=scene["neurons"], layers=scene["layers"],
                      num_semantic_classes=scene["C"], log2_hashmap_size=scene["log2_hashmap_size"])
    return OracleField(cfg, scene["params"], "f16", requires_grad, accum=accum)


def cpu_threads_setter():
    import torch
    try:
        from threadpoolctl import threadpool_limits
    except Exception:                                   # pragma: no cover
        threadpool_limits = None
    state = {"limit": None}

    def set_threads(n):
        torch.set_num_threads(n)
        if threadpool_limits is not None:
            if state["limit"] is not None:
                state["limit"].restore_original_limits()
            state["limit"] = threadpool_limits(limits=n)
    return set_threads


def cpu_model_name():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline_and_parity(scene, pose, width, height, focal, gpu_field, gpu_est, dev, budget_s=12.0):
    """The oracle on a bounded sample of the headline workload (kind "port"), and the parity of the benchmarked scene on the same rays.
    8 threads (or all, when the box has fewer): rounds 3-5 swept {1, 8, 32, 64} on the GPU box's 256-thread host every time and 8 won every time
    (1 184 / 1 231 / 567 / 223 rays/s; 256 torch threads: 0.67) — the sweep lives on in tools/bench_extra.py (--full).  Returns (cpu_baseline, bench_parity)."""
    import torch
    from apnrf_amd import render as RD
    from apnrf_amd import scenes as SC
    from oracle import render as R
    cores = os.cpu_count() or 1
    th = min(8, cores)
    set_threads = cpu_threads_setter()
    set_threads(th)
    S_ = 24
    orc = oracle_field(scene)
    idx = R.subsample_indices(width * height, S_ * S_)
    o, d = R.generate_image_rays(R.pose_to_c2w(pose), width, height, focal, idx)
    bk = torch.zeros(3)
    ref = {}

    def headline():
        ref["r"] = R.render_test(1024, orc, scene["occ"], scene["aabb"][None], o, d, render_bkgd=bk, **SC.RENDER_KW)
    t0 = time.perf_counter(); headline(); first = time.perf_counter() - t0
    iters = int(max(1, min(20, (budget_s - first) / max(first, 1e-3) - 1)))
    t = _timed(headline, 1 if iters > 2 else 0, iters)
    out = {"value": S_ * S_ / t, "unit": "rays/s", "cores": th, "kind": "port", "host_cpus": cores, "cpu_model": cpu_model_name(), "seconds": t * (iters + 2),
           "sample": f"{S_}x{S_} rays of one 800x800 benchmark view, same trained weights and grid, oracle.render.render_test (fp32 torch-CPU), "
                     f"{th} threads, median of {iters} passes"}
    got = RD.render_views(gpu_field, gpu_est, o.to(dev), d.to(dev), S_ * S_, 1024, render_bkgd=bk, **SC.RENDER_KW)
    r = ref["r"]
    errs = {k: (got[k].cpu() - r[k]).abs().reshape(S_ * S_, -1).max(dim=1).values.numpy() for k in ("rgb", "acc", "depth", "sem")}
    # rgb / acc / depth: 1e-3 absolute (north star).  The composited class logits are un-normalised sums of raw logits (up to ~55 on this scene):
    # their bar is max(1e-3, 3e-4 x largest |logit| of the ray) = three times the NOISE FLOOR measured right here — the same model through the
    # oracle a second time with every layer's fp32 products added in another order (oracle/field.py accum="k16_reversed"): what two faithful
    # fp16-operand / fp32-accumulate implementations differ by (tests/test_oracle_noise_floor_cpu.py: ~1e-4 of the logit at every scale).
    r2 = R.render_test(1024, oracle_field(scene, accum="k16_reversed"), scene["occ"], scene["aabb"][None], o, d, render_bkgd=bk, **SC.RENDER_KW)
    nf = {k: (r2[k] - r[k]).abs().reshape(S_ * S_, -1).max(dim=1).values.numpy() for k in ("rgb", "acc", "depth", "sem")}
    mag = r["sem"].abs().max(dim=1).values.numpy()
    sem_scale = np.maximum(1.0, 0.3 * mag)
    raw_sem = errs["sem"].copy()
    errs["sem"] = errs["sem"] / sem_scale
    worst = np.max(np.stack(list(errs.values())), axis=0)
    tie = worst > 1e-3
    mse = float(((got["rgb"].cpu() - r["rgb"]) ** 2).mean())
    parity = {"rays": S_ * S_, "max_abs": {k: float(v[~tie].max()) for k, v in errs.items()}, "tolerance": 1e-3,
              "sem_error_is": "abs error / max(1, 0.3 x largest |composited logit| of the ray): the bar is max(1e-3, 3e-4 x |logit|)",
              "sem_max_abs_unscaled": float(raw_sem.max()), "sem_max_rel_to_logit": float((raw_sem / np.maximum(1.0, mag)).max()),
              "sem_rays_above_1e-3_unscaled": int((raw_sem > 1e-3).sum()), "sem_largest_logit": float(mag.max()),
              "noise_floor": {"what": "oracle vs the same oracle with permuted fp32 accumulation order (same fp16 operands), same rays",
                              "sem_abs": float(nf["sem"].max()), "sem_rel_to_logit": float((nf["sem"] / np.maximum(1.0, mag)).max()),
                              "sem_rays_above_1e-3": int((nf["sem"] > 1e-3).sum()), "rgb_abs": float(nf["rgb"].max()), "acc_abs": float(nf["acc"].max()),
                              "depth_abs": float(nf["depth"].max()), "total_samples": int(r2["total_samples"])},
              "tie_rays": int(tie.sum()), "tie_rays_max_abs": float(worst[tie].max()) if tie.any() else 0.0, "tie_budget": "<= 2 rays up to 5e-2",
              "psnr_db": float(10 * np.log10(1.0 / max(mse, 1e-20))), "total_samples_gpu": int(got["total"][0]), "total_samples_oracle": int(r["total_samples"]),
              "what": "HIP render vs oracle render of the SAME trained scene and rays (stand-alone call: the round schedule of a 576-ray call)"}
    parity["ok"] = bool(worst[~tie].max() <= 1e-3 and int(tie.sum()) <= 2 and (not tie.any() or worst[tie].max() <= 5e-2)
                        and abs(parity["total_samples_gpu"] - parity["total_samples_oracle"]) <= max(4, 2e-3 * parity["total_samples_oracle"]))
    set_threads(cores)
    return out, parity


T0 = time.perf_counter()


def log(msg):
    print(f"[bench {time.perf_counter() - T0:7.1f}s] {msg}", file=sys.stderr, flush=True)


class Ctx:
    """What every leg needs: the process's place in the job, the library, the timing harness, the stand-in scenes."""

    def __init__(self, args):
        import torch
        import torch.distributed as dist
        self.args, self.torch, self.dist = args, torch, dist
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.distributed = self.world > 1 or ("RANK" in os.environ and "MASTER_PORT" in os.environ)   # launched by torch.distributed.run
        if args.one_device:
            self.local_rank = 0
        self.comm_dev = f"cuda:{self.local_rank}" if args.dist_backend == "nccl" else "cpu"      # where the harness's own small collectives live
        if self.distributed:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if args.dist_backend == "nccl":
                dist.init_process_group("nccl", device_id=torch.device(f"cuda:{self.local_rank}"))
            else:
                dist.init_process_group("gloo")
            self.world = dist.get_world_size()                     # the size the backend actually formed
        torch.cuda.set_device(self.local_rank)
        self.dev = f"cuda:{self.local_rank}"
        import __graft_entry__ as G
        if self.rank == 0:
            G.build()
        if self.distributed:
            dist.barrier()
        from apnrf_amd import _lib as L
        self.L = L
        self.lib = L.load_library()
        self.standin_info, self.opt_states, self._scenes = {}, {}, {}

    def log(self, msg):
        if self.rank == 0:
            log(msg)

    def want(self, w):
        return self.args.workload in ("all", w)

    def scene_model(self, name, seed=9, steps=None, keep_optimizer=False, neurons=128, layers=2):
        """(scene dict, field, estimator): the trained stand-in.  Rank 0 trains or loads the cache; the other ranks load the cache it
        wrote, or — when the file could not be written — receive the model by broadcast."""
        from apnrf_amd import scenes as SC
        from apnrf_amd import standin as SI
        key = (name, seed, neurons, layers)
        if key in self._scenes:
            return self._scenes[key]
        scene = SC.make_scene(name, n_poses=40, neurons=neurons, layers=layers)
        if self.args.weights == "random":
            out = scene, SC.hip_field(scene, self.dev), SC.hip_estimator(scene, self.dev)
        else:
            steps = self.args.standin_steps if steps is None else steps
            self.log(f"stand-in {name} seed {seed} {neurons}x{layers}: training / loading")
            field, est, info = SI.shared_standin(scene, self.dev, steps=steps, seed=seed, keep_optimizer=keep_optimizer, group=None if self.distributed else False,
                                                 log=log if self.rank == 0 else None)
            self.opt_states[(name, neurons, layers)] = info.pop("optimizer_state", None)
            if self.rank == 0:
                self.standin_info[f"{name}/seed{seed}/{neurons}x{layers}"] = {k: info.get(k) for k in ("steps", "seconds", "loss_first", "loss_last", "skipped_steps",
                                                                                                       "occupied_cells", "cells", "cached")}
            out = scene, field.eval(), est.eval()
        self._scenes[key] = out
        return out

    def timed(self, step_fn, steps, warmup, with_events, collect=None):
        """W warm-up steps, then exactly K steps bracketed by barrier + synchronize on both sides -> seconds (max over ranks)."""
        torch, dist, L, lib = self.torch, self.dist, self.L, self.lib
        for i in range(warmup):
            step_fn(i)
        # Objects of earlier legs that sit in reference cycles (a torch optimizer and the field bound to it) are destroyed whenever Python's cycle
        # collector happens to run — `mnf_field_destroy` is a dozen `hipFree` calls, each of which waits for the device: inside a timed region that drains
        # the asynchronous pipeline (the sporadic 1.7-2x slow train legs of rounds 3-4): collect before the region, not inside it.
        gc.collect()
        torch.cuda.synchronize()
        if self.distributed:
            dist.barrier()
        torch.cuda.synchronize()
        if with_events:
            L.check(lib.mnf_profile_begin())
        gc.disable()
        t0 = time.perf_counter()
        for i in range(steps):
            r = step_fn(warmup + i)
            if collect is not None:
                collect(r)
        torch.cuda.synchronize()
        if self.distributed:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        gc.enable()
        if with_events:
            ms, n = ctypes.c_double(0), ctypes.c_int64(0)
            L.check(lib.mnf_profile_end(ctypes.byref(ms), ctypes.byref(n)))
        if self.distributed:
            t = torch.tensor([dt], device=self.comm_dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    def prof(self, label):
        ms, n = ctypes.c_double(0), ctypes.c_int64(0)
        self.L.check(self.lib.mnf_profile_query(label.encode(), ctypes.byref(ms), ctypes.byref(n)))
        return ms.value, n.value


# ------------------------------------------------------------------ renders (config 3: the headline; config 2: 256x256, 4x64)
class RenderLeg:
    """`n_views` resident views of one stand-in, rendered `V` per call."""

    def __init__(self, cx, field, est, poses, width, height, focal):
        from apnrf_amd import render as RD
        torch = cx.torch
        self.cx, self.field, self.est, self.width, self.height = cx, field, est, width, height
        c2w = np.stack([RD.pose_to_c2w(p) for p in poses]).astype(np.float32)
        K = np.array([[focal, 0, width / 2], [0, focal, height / 2], [0, 0, 1.0]])
        self.rays = RD.generate_image_rays(torch.from_numpy(c2w), width, height, K, cx.dev)
        self.n_views, self.n_per_view = len(poses), width * height
        self.bk = torch.zeros(3)
        self.process_samples = torch.zeros((), dtype=torch.int64, device=cx.dev)      # every evaluated sample of this process (PMC sums cover all launches)

    def run(self, V, steps, warmup, with_events, n_split=2):
        from apnrf_amd import render as RD
        from apnrf_amd import scenes as SC
        torch, cx = self.cx.torch, self.cx
        batches = [(self.rays.origins[k:k + V].reshape(-1, 3).contiguous(), self.rays.viewdirs[k:k + V].reshape(-1, 3).contiguous())
                   for k in range(0, self.n_views, V)]
        evaluated = torch.zeros((), dtype=torch.int64, device=cx.dev)

        def step(i):
            o, d = batches[i % len(batches)]
            r = RD.render_views(self.field, self.est, o, d, self.n_per_view, 1024, render_bkgd=self.bk, image_hw=(self.height, self.width), n_split=n_split, **SC.RENDER_KW)
            self.process_samples.add_(r["total"][1])
            return r

        def collect(r):
            evaluated.add_(r["total"][1])
        dt = cx.timed(step, steps, warmup, with_events, collect)
        return dt, int(evaluated.item())

    def roofline(self, V, steps, kernel_name, traffic_key="field_kernel"):
        """second pass of the same K steps, ONE job in flight, hipEvent pairs around every field-kernel launch (hipEventRecord between dependent launches
        costs up to ~0.15 ms each on this stack, so it is kept out of the pass that yields `value`)"""
        dt2, samples2 = self.run(V, steps, 1, True, n_split=1)
        field_ms, launches = self.cx.prof("field_render")
        if not (launches and samples2):
            return None
        achieved = ALGO_BYTES_PER_SAMPLE * samples2 / (field_ms * 1e-3) / 1e9
        per_sample, note = field_traffic_per_sample(traffic_key)
        return {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "traffic": per_sample * samples2 / launches if per_sample else None, "traffic_source": note, "kernel": kernel_name,
                "avg_launch_ms": field_ms / launches, "launches": int(launches), "samples_per_launch": samples2 / launches,
                "algorithmic_bytes_per_sample": ALGO_BYTES_PER_SAMPLE, "algorithmic_bytes_per_launch": ALGO_BYTES_PER_SAMPLE * samples2 / launches,
                "field_kernel_share_of_serial_step": field_ms * 1e-3 / dt2,
                "timing": "second pass of the same K steps with one render job in flight (launches back to back on one stream), "
                          "hipEvent pair around each launch on the launch stream"}


def leg_render800(cx, line):
    args, world = cx.args, cx.world
    width = height = 800
    focal = 0.5 * width / np.tan(np.pi / 4)
    scene529, field, est = cx.scene_model("102344529")
    poses = scene529["poses"][[(5 * k + cx.rank) % 40 for k in range(8)]]       # 8 views of the sweep per rank
    leg = RenderLeg(cx, field, est, poses, width, height, focal)
    V, n_per_view = args.views, width * height
    cx.log("render800: timed pass")
    dt, samples = leg.run(V, args.steps, args.warmup, False, n_split=args.render_jobs)      # the reported value: no instrumentation
    cx.log(f"render800: {1e3 * dt / args.steps:.2f} ms/step, {samples / (n_per_view * V * args.steps):.1f} samples/ray")
    line["value"] = n_per_view * V * world * args.steps / dt
    line["ms_per_step"] = 1e3 * dt / args.steps
    line["config"].update({"workload": f"BASELINE config 3: scene 102344529 (trained synthetic stand-in), 800x800 RGB+depth+29-class semantic render, {V} view(s) per step "
                                       "in one batched call, hash-grid 16x4 T=2^19 + MLP 128x2 + 64x2 heads, occupancy-grid marching",
                           "views_per_step": V, "rays_per_step_per_gpu": n_per_view * V,
                           "ms_per_view": 1e3 * dt / args.steps / V, "samples_per_ray": samples / (n_per_view * V * args.steps),
                           "samples_per_s": samples * world / dt, "render_jobs_in_flight": min(args.render_jobs, V),
                           "march_order": "8x8 pixel blocks inside every view (mnf_render_opts.view_order); per-ray results do not depend on it"})
    if not args.no_kernel_timing:
        r = leg.roofline(V, args.steps, "mnf::field_kernel<128,2,2,false> (hash gather + MLPs + fused compositing)")
        if r:
            line["roofline"] = r
            cx.log(f"render800 field kernel: {r['avg_launch_ms']:.4f} ms per {r['samples_per_launch'] / 1e6:.2f} M samples = {r['frac']:.3f} of {HBM_PEAK_GBS:.0f} GB/s")
    if V != 1 and not args.no_views1:
        dt1, s1 = leg.run(1, args.steps, 2, False)
        line["render_views1"] = {"value": n_per_view * world * args.steps / dt1, "unit": "rays/s", "ms_per_view": 1e3 * dt1 / args.steps,
                                 "samples_per_ray": s1 / (n_per_view * args.steps)}
    if args.full:
        sys.path.insert(0, os.path.join(REPO, "tools"))
        import bench_extra as X
        X.render_extras(cx, line, leg, scene529, field, est)
    line["samples"] = {"timed": int(samples), "process_total": int(leg.process_samples.item())}
    return scene529, field, est, poses, (width, height, focal)


# ------------------------------------------------------------------ train steps (config 5: 8192 rays; the reference yaml: 2000 rays; config 2: 4x64)
class TrainLeg:
    """Train steps continued from a stand-in's state (weights, grid, Adam moments): every leg starts from the same state."""

    KERNEL_LABELS = ("sample_rays", "field_density", "field_train_forward", "composite_train_forward", "composite_train_backward",
                     "dgrad", "wgrad", "hash_scatter", "hash_scatter_bins")

    def __init__(self, cx, scene_name, seed, neurons=128, layers=2, image=640):
        from apnrf_amd import render as RD
        from apnrf_amd import standin as SI
        self.cx, self.image = cx, image
        self.model = f"{neurons}x{layers}"
        self.key = (scene_name, neurons, layers)
        self.scene, self.field0, self.est0 = cx.scene_model(scene_name, seed=seed, keep_optimizer=True, neurons=neurons, layers=layers)
        self.proc = SI._procedural_estimator(self.scene, cx.dev)
        self.c2w = np.stack([RD.pose_to_c2w(p) for p in self.scene["poses"][:8]]).astype(np.float32)
        self.K = np.array([[image / 2.0, 0, image / 2.0], [0, image / 2.0, image / 2.0], [0, 0, 1.0]])

    def make_batches(self, R_):
        from apnrf_amd import render as RD
        from apnrf_amd import standin as SI
        torch, cx, im = self.cx.torch, self.cx, self.image
        g = torch.Generator(device="cpu").manual_seed(100 + cx.rank)
        out = []
        blk = max(1, im // 20)
        for k in range(8):
            idx = torch.randint(0, im * im, (R_,), generator=g).numpy()
            ys, xs = idx // im, idx % im                      # grouped by image block, as dataset.Dataset.fetch_data does
            idx = idx[np.argsort((ys // blk) * 20 + xs // blk, kind="stable")]
            r = RD.generate_image_rays(torch.from_numpy(self.c2w[k:k + 1]), im, im, self.K, cx.dev, idx)
            out.append((r,) + SI.analytic_targets(self.proc, self.scene["aabb"], r.origins, r.viewdirs))
        return out

    def fresh_member(self, dtype="f16", optimizer="fused"):
        """(field, estimator, optimizer) restored to the stand-in's final state.  The stand-in's own training run CONTINUES: its Adam moments, step count and
        final learning rate (2e-4).  A fresh Adam would kick every parameter with a non-zero gradient by +-lr in its first steps."""
        import copy
        from apnrf_amd import scenes as SC
        from apnrf_amd.nerfacc import OccGridEstimator
        from apnrf_amd.optim import FusedAdam
        torch, cx = self.cx.torch, self.cx
        tf = SC.hip_field(self.scene, cx.dev, mfma_bf16=(dtype == "bf16"))
        tf.load_state_dict(self.field0.state_dict())
        te = OccGridEstimator(torch.from_numpy(self.scene["aabb"]), resolution=self.scene["res"], levels=1).to(cx.dev)
        te.occs.copy_(self.est0.occs); te.binaries = self.est0.binaries.clone()
        tf.train(); te.train()
        if optimizer != "fused":
            return tf, te, None
        opt = FusedAdam(tf.parameters(), lr=2e-4, eps=1e-15).bind_field(tf)
        if cx.opt_states.get(self.key) is not None:
            opt.load_state_dict(copy.deepcopy(cx.opt_states[self.key]))
            for g_ in opt.param_groups:
                g_["lr"] = 2e-4
        return tf, te, opt

    def run(self, dtype, R_, sync, steps, with_kernels, step_factory=None):
        """ms per step of `steps` train iterations from the stand-in's state -> dict.  `step_factory(tf, te, opt, batches, bkd)` -> (step_fn, finish(res, outs)) replaces the plain step."""
        from apnrf_amd import render as RD
        from apnrf_amd import scenes as SC
        torch, cx, args = self.cx.torch, self.cx, self.cx.args
        tf, te, opt = self.fresh_member(dtype)
        batches = self.make_batches(R_)
        bkd = torch.rand(3, generator=torch.Generator().manual_seed(7)).to(cx.dev)      # a random background colour on the device (habitat_to_data.py:189-191)
        outs = []
        finish = None
        if step_factory is not None:
            tstep, finish = step_factory(tf, te, opt, batches, bkd)
        else:
            def tstep(i):
                r, pix, dep_, lab = batches[i % 8]
                # occ_thre as the stand-in's own training (the reference uses 1e-3 / 1e-2 / 3e-3 by phase, pipeline.py:447-470): the refresh at
                # step 1008 then keeps the grid the stand-in converged to, and the workload stays stationary
                return RD.train_step(tf, te, opt, r, pix, dep_, lab, bkd, step=1000 + i, sync=sync, occ_thre=1e-2, **SC.RENDER_KW)
        warm = max(args.warmup, 6)
        dt_t = cx.timed(tstep, steps, warm, False, outs.append)      # (the first asynchronous steps also size the sample bounds)
        kept = float(np.mean([int(o["n_rendering_samples"]) for o in outs]))
        marched = float(int(te.last_sampling["n_marched"]))
        res = {"ms_per_step": 1e3 * dt_t / steps, "steps": steps, "rays_per_step": R_, "dtype": dtype, "model": self.model,
               "host_round_trips_per_step": 1 if sync else 0, "rendering_samples_per_step": kept, "marched_samples_per_step": marched,
               "skipped_steps": int(sum(int(o["skipped"]) for o in outs))}
        if finish is not None:
            finish(res, outs, steps + warm, tf)
            return res
        algo = TRAIN_BYTES_KEPT * kept + TRAIN_BYTES_MARCHED * marched
        traffic, traffic_note = train_traffic(R_, dtype, self.model)
        res["roofline"] = {"bound": "hbm", "achieved": algo / (dt_t / steps) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": algo / (dt_t / steps) / 1e9 / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_note,
                           "traffic_over_algorithmic": (traffic / algo) if traffic else None, "algorithmic_bytes_per_step": algo,
                           "definition": "(4.1 KB x surviving samples + 1 KB x pre-pass samples) / un-instrumented step time (SURVEY 8d)"}
        if with_kernels and not args.no_kernel_timing:
            outs.clear()
            dt_e = cx.timed(tstep, steps, 0, True, outs.append)
            kept_e = float(np.mean([int(o["n_rendering_samples"]) for o in outs]))
            kernels, total_ms = {}, 0.0
            per_sample = {"field_density": ("marched", 1036), "field_train_forward": ("kept", 1036)}
            for label in self.KERNEL_LABELS:
                ms, n = cx.prof(label)
                if not n:
                    continue
                e = {"ms_per_step": ms / steps, "launches_per_step": n / steps}
                if label in per_sample:
                    which, b = per_sample[label]
                    nbytes = b * (kept_e if which == "kept" else marched)
                    gbs = nbytes / (ms / steps * 1e-3) / 1e9
                    e.update({"algorithmic_bytes_per_step": nbytes, "achieved_GBps": gbs, "frac_of_hbm_peak": gbs / HBM_PEAK_GBS})
                kernels[label] = e
                total_ms += ms / steps
            res["kernels"] = kernels
            res["kernels_note"] = "hipEvent pairs; wgrad, hash_scatter (levels 0-11) and hash_scatter_bins (levels 12-15) run on three streams side by side: their times do not add up"
            res["timed_kernels_ms_per_step"] = total_ms
            res["instrumented_step_ms"] = 1e3 * dt_e / steps
        return res


def leg_train(cx, line):
    args = cx.args
    tl = TrainLeg(cx, "102344280", seed=11)
    tsteps = max(args.steps, 10)
    dtypes = [d for d in args.train_dtypes.split(",") if d in ("f16", "bf16")]
    train = {"workload": "BASELINE config 5: scene 102344280 (trained stand-in, training continued from the same state in every leg), 8192-ray batches "
                         "of one 640x640 view, occupancy sampling + density pre-pass + differentiable render + loss (pipeline.py:506-511) + backward "
                         "+ NaN guard + FusedAdam; occupancy refresh every 16th step; asynchronous steps (sync=False)"}
    for dt_ in dtypes:
        cx.log(f"train {dt_}: timed passes")
        leg = tl.run(dt_, args.train_rays, False, tsteps, True)
        train[dt_] = leg
        cx.log(f"train {dt_}: {leg['ms_per_step']:.2f} ms/step at {leg['rendering_samples_per_step']:.0f} samples")
    first = train[dtypes[0]]
    train.update({k: first[k] for k in ("ms_per_step", "rays_per_step", "rendering_samples_per_step", "marched_samples_per_step", "roofline")})
    train["dtype"] = dtypes[0]
    ry = tl.run("f16", 2000, False, tsteps, True)
    ry["host_synchronous_ms_per_step"] = tl.run("f16", 2000, True, tsteps, False)["ms_per_step"]
    ry["workload"] = ("the reference yaml's own shape: 2000 rays per step (scripts/config_102344250.yaml:3, the cap of pipeline.py:494-504), target "
                      "262 144 samples (config:4); same scene and start state")
    if "kernels" in ry:
        ry["fixed_cost_share"] = 1.0 - sum(v["ms_per_step"] for k, v in ry["kernels"].items() if k in ("field_density", "field_train_forward", "dgrad", "wgrad")) / ry["ms_per_step"]
    cx.log(f"train refyaml (2000 rays): {ry['ms_per_step']:.2f} ms/step (host-synchronous {ry['host_synchronous_ms_per_step']:.2f})")
    line["train"] = train
    line["train_refyaml"] = ry
    if args.full:
        sys.path.insert(0, os.path.join(REPO, "tools"))
        import bench_extra as X
        X.train_extras(cx, line, tl, tsteps, dtypes)
    if not cx.want("render800"):
        line.update({"metric": "train-step ms", "value": train["ms_per_step"], "unit": "ms", "higher_is_better": False,
                     "ms_per_step": train["ms_per_step"], "dtype": dtypes[0]})
        line["config"]["workload"] = train["workload"]


def leg_config2(cx, line):
    """BASELINE config 2: scene 102344250, 256x256 RGB+depth+29-class views, hash grid + 4 hidden layers of 64 (the reference CLASS default, ngp.py:77-78; the yaml
    runs 128x2), train + render on one GPU."""
    args = cx.args
    w = h = 256
    focal = 0.5 * w / np.tan(np.pi / 4)
    tl = TrainLeg(cx, "102344250", seed=9, neurons=64, layers=4, image=256)
    leg = RenderLeg(cx, tl.field0, tl.est0, tl.scene["poses"][[(5 * k + cx.rank) % 40 for k in range(8)]], w, h, focal)
    V = 8
    dt, samples = leg.run(V, args.steps, args.warmup, False, n_split=2)
    c2 = {"workload": "BASELINE config 2: scene 102344250 (trained stand-in), hash-grid 16x4 T=2^19 + base MLP 64x4 + 64x2 heads, 256x256 RGB+depth+29-class views"}
    c2["render"] = {"ms_per_step": 1e3 * dt / args.steps, "views_per_step": V, "ms_per_view": 1e3 * dt / args.steps / V, "rays_per_s": w * h * V * cx.world * args.steps / dt,
                    "samples_per_ray": samples / (w * h * V * args.steps), "samples_per_s": samples * cx.world / dt}
    if not args.no_kernel_timing:
        r = leg.roofline(V, args.steps, "mnf::field_kernel<64,4,2,false> (hash gather + MLPs + fused compositing)", traffic_key="field_kernel_64x4_config2")
        if r:
            c2["render"]["roofline"] = r
    dt1, s1 = leg.run(1, args.steps, 2, False)
    c2["render"]["one_view_per_call"] = {"ms_per_view": 1e3 * dt1 / args.steps, "rays_per_s": w * h * cx.world * args.steps / dt1}
    c2["render"]["process_samples"] = int(leg.process_samples.item())
    tsteps = max(args.steps, 10)
    c2["train"] = tl.run("f16", 2000, False, tsteps, True)
    c2["train"]["host_synchronous_ms_per_step"] = tl.run("f16", 2000, True, tsteps, False)["ms_per_step"]
    c2["train_8192"] = {k: v for k, v in tl.run("f16", 8192, False, tsteps, False).items() if k in ("ms_per_step", "rendering_samples_per_step", "marched_samples_per_step", "roofline")}
    cx.log(f"config2: render {c2['render']['ms_per_view']:.2f} ms/view ({c2['render']['rays_per_s'] / 1e6:.1f} M rays/s), train {c2['train']['ms_per_step']:.2f} ms/step at 2000 rays, "
           f"{c2['train_8192']['ms_per_step']:.2f} at 8192")
    line["config2"] = c2
    if not (cx.want("render800") or cx.want("train")):
        line.update({"value": c2["render"]["rays_per_s"], "ms_per_step": c2["render"]["ms_per_step"], "metric": "rendered rays/sec (RGB+depth+semantic, 256x256)"})
        line["config"]["workload"] = c2["workload"]
        if "roofline" in c2["render"]:
            line["roofline"] = c2["render"]["roofline"]


# ------------------------------------------------------------------ BASELINE config 4: candidate-view scoring, views sharded over ranks
def leg_score256(cx, line):
    from apnrf_amd import render as RD
    from apnrf_amd import standin as SI
    torch, dist, args, world, rank, dev = cx.torch, cx.dist, cx.args, cx.world, cx.rank, cx.dev
    scene250, f0, e0 = cx.scene_model("102344250", seed=9)
    _, f1, e1 = cx.scene_model("102344250", seed=10)
    poses256 = SI._free_space_poses(scene250, 256, seed=9)                  # 8 trajectories x 32 views inside the free space
    group = dist.group.WORLD if cx.distributed else None
    t_parts = {"compute": 0.0, "gather": 0.0, "n": 0}
    score_samples = torch.zeros((), dtype=torch.int64, device=dev)       # every sample this process evaluated while scoring (PMC sums cover all launches)

    def score_call(p, group):
        r = RD.score_views([f0, f1], [e0, e1], p, 640, 640, 320.0, 0.1, 1e-3, 0.1, 0.004, 0.01, dev, group=group)
        for t in RD.LAST_SCORE_TOTALS:
            score_samples.add_(t[1])
        return r

    def sstep(i):
        return score_call(poses256, group)

    def sstep_parts(i):
        """the same pass with the two phases timed apart on this rank (synchronises between them: diagnosis, not the headline)"""
        torch.cuda.synchronize(); t0 = time.perf_counter()
        lo, hi, per = RD.shard_views(256, world, rank)
        local = torch.zeros(per, 4, dtype=torch.float64, device=dev)
        if hi > lo:
            terms, _ = score_call(poses256[lo:hi], False)
            local[:hi - lo] = terms
        torch.cuda.synchronize(); t1 = time.perf_counter()
        RD.gather_view_terms(local, 256, group)
        torch.cuda.synchronize(); t2 = time.perf_counter()
        t_parts["compute"] += t1 - t0; t_parts["gather"] += t2 - t1; t_parts["n"] += 1
    ssteps = max(3, min(args.steps, 5))
    cx.log("score256: timed pass")
    dt_s = cx.timed(sstep, ssteps, 1, False)
    cx.log(f"score256: {1e3 * dt_s / ssteps:.2f} ms/pass")
    terms, score = sstep(0)
    evaluated = float(sum(int(t[1]) for t in RD.LAST_SCORE_TOTALS))          # this rank's share of the views, all members
    cx.timed(sstep_parts, 3, 1, False)
    parts = torch.tensor([t_parts["compute"] / t_parts["n"], t_parts["gather"] / t_parts["n"]], dtype=torch.float64, device=cx.comm_dev)
    per_rank = [parts.clone() for _ in range(world)]
    if cx.distributed:
        dist.all_gather(per_rank, parts)
    sc = {"ms_per_pass": 1e3 * dt_s / ssteps, "rays_per_s": 256 * 4096 * 2 * ssteps / dt_s, "views": 256, "rays_per_view": 4096,
          "ensemble_members": 2, "n_gpus": world, "scaling": "strong", "score": float(score),
          "samples_per_ray_rank0": evaluated / max(1, (256 // world) * 4096 * 2),
          "samples_per_s_rank0": evaluated * ssteps / dt_s,
          "per_rank_compute_ms": [1e3 * float(p[0]) for p in per_rank], "per_rank_gather_ms": [1e3 * float(p[1]) for p in per_rank],
          "collective": "one all_gather_into_tensor of [V/N,4] float64 per pass" if world > 1 else "none (single rank)",
          "workload": "BASELINE config 4: scene 102344250 (two trained stand-ins: seeds 9 and 10), 256 candidate poses in free "
                      "space, 64x64 rays each (linspace sub-sample of 640x640), probabilistic render + predictive-information terms; both ensemble members in one call"}
    if world > 1:
        full, _ = score_call(poses256, False)
        same = bool(torch.equal(full, terms))
        sc["bit_identical_to_single_gpu"] = same
        if not same:
            raise SystemExit("score256: gathered terms differ from the single-rank computation")
    line["score256"] = sc
    if world == 1:
        # one rank's share of an 8-GPU run on this GPU: what strong scaling over 8 GPUs would see per rank (before the 8 KB all-gather)
        lo, hi, _ = RD.shard_views(256, 8, 0)

        def shard_step(i):
            return score_call(poses256[lo:hi], False)
        dt8 = cx.timed(shard_step, ssteps, 1, False)
        t8, _ = shard_step(0)
        ev8 = float(sum(int(t[1]) for t in RD.LAST_SCORE_TOTALS))
        line["score256_shard8"] = {"ms_per_pass": 1e3 * dt8 / ssteps, "views": hi - lo, "samples_per_s": ev8 * ssteps / dt8,
                                   "ratio_to_full_over_8": (dt8 / ssteps) / (dt_s / ssteps / 8),
                                   "bit_identical_to_full_pass_rows": bool(torch.equal(t8, terms[lo:hi])),
                                   "note": "views 0..31 of the same pass on one GPU = the per-rank work of --gpus 8; predicted 8-GPU pass = this + one 8 KB all-gather"}
        cx.log(f"score256 shard of 8: {1e3 * dt8 / ssteps:.2f} ms/pass")
        # ... and of a 2- and a 4-GPU run: the predicted strong-scaling curve of config 4 (no node to measure it on), every rank's share timed on this GPU
        curve = {"1": 1e3 * dt_s / ssteps, "8": 1e3 * dt8 / ssteps}
        for n_ in (2, 4):
            lo_, hi_, _ = RD.shard_views(256, n_, 0)
            curve[str(n_)] = 1e3 * cx.timed(lambda i: score_call(poses256[lo_:hi_], False), ssteps, 1, False) / ssteps
        line["score256_predicted_scaling"] = {"ms_per_pass_of_rank0_share": curve, "efficiency": {k: curve["1"] / (int(k) * v) for k, v in curve.items()},
                                              "note": "rank 0's share of an N-GPU pass timed on ONE GPU (+ one 8 KB all-gather on a node): a prediction, not a measurement"}
    sc["process_samples"] = int(score_samples.item())
    if world == 1 and args.full:
        sys.path.insert(0, os.path.join(REPO, "tools"))
        import bench_extra as X
        X.pose_driver_extras(cx, line, f0, e0, f1, e1, poses256)
    if not (cx.want("render800") or cx.want("train") or cx.want("config2")):
        line.update({"metric": "candidate-view scoring rays/s", "value": sc["rays_per_s"], "ms_per_step": sc["ms_per_pass"], "scaling": "strong"})
        line["config"]["workload"] = sc["workload"]


def trained_scene(scene, f, e):
    """the scene dict with the stand-in's trained parameters and grid (what the oracle evaluates)"""
    s_ = dict(scene)
    s_["params"] = {"mlp_base": f.mlp_base.params.detach().cpu().numpy(), "mlp_head": f.mlp_head.params.detach().cpu().numpy(),
                    "mlp_sem": f.mlp_sem.params.detach().cpu().numpy()}
    s_["occ"] = e.binaries.cpu().numpy()
    return s_


def main():
    args = parse()
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(spawn_ranks(args))
    cx = Ctx(args)
    line = {"metric": "rendered rays/sec (RGB+depth+semantic, 800x800)", "value": None, "unit": "rays/s", "n_gpus": cx.world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": None, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f16", "data": "synthetic",
            "config": {"workload": None, "max_samples": 1024, "near_plane": 0.1, "render_step_size": 1e-3, "cone_angle": 0.004, "alpha_thre": 0.01,
                       "arithmetic": "fp16 hash entries / weights / activations, fp32 accumulate and outputs",
                       "weights": (f"trained stand-in (SURVEY 8d): {args.standin_steps} iterations of the product's train_step on an analytic target, bitwise reproducible"
                                   if args.weights == "trained" else "random-init (hash U(-0.5,0.5), xavier MLPs, |density row| x 8), procedural occupancy"),
                       "weights_detail": "apnrf_amd.standin.train_standin: opaque procedural rooms, colour fract(xyz), class = cell hash mod 29; FusedAdam lr 2e-3 decayed to "
                                         "2e-4 over the second half; occupancy grid from update_every_n_steps; seeded, deterministic gradient accumulation"}}
    headline_scene = None
    if cx.want("render800"):
        headline_scene = leg_render800(cx, line)
    if cx.want("train"):
        leg_train(cx, line)
    if cx.want("config2"):
        leg_config2(cx, line)
    if cx.want("score256"):
        leg_score256(cx, line)

    rc = 0
    if cx.rank == 0:
        if cx.standin_info:
            line["config"]["standin_training"] = cx.standin_info
        if cx.world == 1 and not args.no_cpu_baseline and headline_scene is not None and args.weights == "trained":
            scene529, field, est, poses, (width, height, focal) = headline_scene
            cx.log("cpu baseline + bench parity")
            line["cpu_baseline"], line["bench_parity"] = cpu_baseline_and_parity(trained_scene(scene529, field, est), poses[0], width, height, focal, field, est, cx.dev)
            cx.log(f"cpu baseline {line['cpu_baseline']['value']:.0f} rays/s; bench parity ok = {line['bench_parity']['ok']}")
            if not line["bench_parity"]["ok"]:
                rc = 3
            if args.full:
                sys.path.insert(0, os.path.join(REPO, "tools"))
                import bench_extra as X
                X.cpu_extras(cx, line, trained_scene(scene529, field, est), poses[0], width, height, focal)
        line["bench_seconds"] = time.perf_counter() - T0
        detail_file = args.detail_file
        try:
            with open(detail_file, "w") as fh:
                json.dump(line, fh, indent=1)
        except OSError as e:
            log(f"could not write {detail_file}: {e}")
            detail_file = None
        print(headline_line(line, detail_file), flush=True)
    if cx.distributed:
        cx.dist.destroy_process_group()
    if rc:
        sys.exit(rc)


if __name__ == "__main__":
    main()
