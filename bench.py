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
  roofline              the dominant kernel (fused hash gather + MLPs + compositing): algorithmic bytes / hipEvent time of its
                        launches in a second pass of the same steps with ONE render job in flight (the timed pass runs two
                        jobs side by side, whose launches overlap: a per-launch duration is only meaningful for the serial form)
  bench_parity          the same trained scene: 576 sub-sampled rays of a benchmark view rendered stand-alone by the HIP path and by
                        the oracle (the CPU baseline's pass): max abs error per output, PSNR, tie rays, sample totals.
                        The run FAILS (exit code 3) above the north-star tolerance.
  render_views1         the headline workload at one view per call (the reference's own call granularity)
  render_random_weights the same views with round 1's engineered random-init weights (continuity across rounds)
  train                 BASELINE config 5: train step on scene 102344280, 8192 rays — fp16 (the reference's tcnn arithmetic) AND
                        bf16 matrix-core operands, each without host round trips (`sync=False`) and in the reference's
                        host-synchronous form; per-kernel times and rooflines; `train_refyaml`: the reference yaml's own shape
                        (2000 rays, ~262 144 samples: scripts/config_102344250.yaml:3-4, pipeline.py:494-504)
  score256              BASELINE config 4: 256 candidate views x 4096 rays x 2 ensemble members on scene 102344250,
                        probabilistic renders + on-device scorer, views sharded over the ranks, ONE all-gather of the [V,4]
                        terms (per-rank compute and gather times reported apart); with N > 1 rank 0 re-computes all views alone
                        and the gathered terms must be bit-identical.  `score256_shard8`: 32 views on this GPU = one rank's
                        share of an 8-GPU run (predicts strong scaling without a node)
  train_dropin          scripts/pipeline.py:472-532 UNCHANGED on the drop-in surface (autograd route, torch losses, per-parameter isnan,
                        torch.optim.Adam) at the reference yaml's 2000 rays, beside the fused step's numbers
  render_from_pose      Dataset.render_image_from_pose (one 640x640 pose, pipeline.py:960-974) and render_probablistic_image_from_pose
                        (40 poses x 2 members at scale 0.1, pipeline.py:697-711) with their float64 host stacks
  cpu_baseline          the oracle (CPU port of the same path) on this box's host cores: BASELINE.md §4 protocol (3 warm-ups + 20
                        iterations, threads swept over {1, 8, 32, all}, best reported), shapes (i) BL-1 and (ii) headline sample,
                        one 4096-ray scoring view, one 2000-ray train step

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
ATOMIC_PEAK_GREQ = 21.0               # memory-side atomic requests per second, x1e9 (tools/atomic_bench.hip, profiles/r03_atomic_microbench.txt)
FIELD_SOURCES = ("field.hip", "field_dev.h", "composite_dev.h", "field.h", "common.h")


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
    ap.add_argument("--no-kernel-timing", action="store_true", help="skip the second, hipEvent-instrumented passes")
    ap.add_argument("--no-views1", action="store_true", help="skip the one-view-per-call and random-weight passes (profiling runs)")
    ap.add_argument("--render-jobs", type=int, default=2, choices=[1, 2, 4],
                    help="render jobs in flight in the timed 800x800 pass (2 = the reported configuration; 1 = launches back to back on one stream, the form "
                         "the roofline pass uses: a rocprofv3 --stats run with 1 gives per-launch durations that can be compared with roofline.avg_launch_ms)")
    ap.add_argument("--train-dtypes", default="f16,bf16", help="matrix-core operand types of the train leg")
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


TRAIN_SOURCES = ("train.hip", "trainstep.hip", "composite_train.hip", "field.hip", "field_dev.h", "field.h", "common.h", "march.hip", "march_dev.h")


def train_traffic(rays):
    """HBM bytes of one train step from the committed counter passes (profiles/r05_pmc_train.json: rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE | TCC_EA0_ATOMIC_sum,
    tools/r05_pmc_train.sh) — quoted only for the kernel sources they were measured on (md5) and the shape they were measured at.  -> (bytes or None, note)"""
    pj = os.path.join(REPO, "profiles", "r05_pmc_train.json")
    pj_shape = os.path.join(REPO, "profiles", f"r05_pmc_train_{int(rays)}.json")      # (a second pass set at another batch size, e.g. the reference yaml's 2000 rays)
    if os.path.exists(pj_shape):
        pj = pj_shape
    if not os.path.exists(pj):
        return None, "no PMC profile of the train step is committed"
    pm = json.load(open(pj))
    h = hashlib.md5()
    for f in TRAIN_SOURCES:
        with open(os.path.join(REPO, "active-perception-using-neural-radiance-fields_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    if pm.get("train_sources_md5") != h.hexdigest()[:12]:
        return None, "profiles/" + os.path.basename(pj) + " was measured on a different build of the train kernels: not quoted"
    if int(pm.get("rays_per_step", 0)) != int(rays):
        return None, f"profiles/r05_pmc_train.json was measured at {pm.get('rays_per_step')} rays per step: not quoted for {rays}"
    return pm["per_step"]["hbm_bytes"], ("rocprofv3 --pmc passes of these kernel sources (profiles/" + os.path.basename(pj) + ": FETCH_SIZE corrected x2 for the 16-byte-per-lane streaming kernels "
                                        "+ WRITE_SIZE, per train step at %s surviving samples; copied from the profile, not measured in this run)" % pm.get("surviving_samples_per_step"))


def field_source_id():
    """Hash of the sources of the dominant kernel: PMC traffic measured on another build is not quoted for this one."""
    h = hashlib.md5()
    for f in FIELD_SOURCES:
        with open(os.path.join(REPO, "active-perception-using-neural-radiance-fields_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:12]


# ------------------------------------------------------------------ CPU baselines (the oracle, timed on this box's cores)
def _timed(fn, warm, iters):
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(iters):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    return float(np.median(ts))


def _oracle_field(scene, requires_grad=False, accum="whole"):
    from oracle.field import FieldConfig, OracleField
    cfg = FieldConfig(aabb=tuple(float(x) for x in scene["aabb"]), neurons=scene["neurons"], layers=scene["layers"],
                      num_semantic_classes=scene["C"], log2_hashmap_size=scene["log2_hashmap_size"])
    return OracleField(cfg, scene["params"], "f16", requires_grad, accum=accum)


def _thread_sweep(fn, unit_count, set_threads, budget_s=25.0):
    """3 warm-ups + 20 timed iterations, median, threads swept over {1, 8, 32, all}, best reported — bounded in wall time:
    every thread count gets budget_s / 4; when 3 + 20 passes do not fit, fewer are run (at least one), and a thread count whose first pass is
    more than 3x slower than the best so far is recorded from that one pass (256 spinning threads on small ops).
    -> (best units/s, threads of the best, {threads: units/s}, protocol string)."""
    cores = os.cpu_count() or 1
    sweep, notes, best_t = {}, [], None
    # "all" is capped at 64 threads: on the 256-thread host of the GPU box one 24x24-ray oracle pass took 860 s with 256 torch threads
    # (0.67 rays/s against 1218 rays/s on 8: profiles/r03_bench_line_first.json), thousands of small ops spinning on one another
    for th in sorted({1, min(8, cores), min(32, cores), min(64, cores)}):
        set_threads(th)
        t0 = time.perf_counter(); fn(); first = time.perf_counter() - t0
        if best_t is not None and first > 3.0 * best_t:
            sweep[th] = unit_count / first; notes.append(f"{th} threads: 1 pass"); continue
        n = int(max(0, min(23, (budget_s / 4 - first) / max(first, 1e-4))))
        warm, iters = (3, 20) if n >= 23 else (min(1, max(n - 1, 0)), max(n - 1, 0))
        t = _timed(fn, warm, iters) if iters > 0 else first
        sweep[th] = unit_count / t
        notes.append(f"{th} threads: {warm} warm-up(s) + {iters} iterations" if iters else f"{th} threads: 1 pass")
        best_t = t if best_t is None else min(best_t, t)
    best = max(sweep, key=sweep.get)
    # the reported figure itself follows the protocol in full (3 warm-ups + 20 timed iterations) when that fits ~25 s: the sweep above only picks the thread count
    per_pass = unit_count / sweep[best]
    if f"{best} threads: 3 warm-up(s) + 20 iterations" not in notes and 23 * per_pass <= 25.0:
        set_threads(best)
        sweep[best] = unit_count / _timed(fn, 3, 20)
        notes.append(f"reported value: {best} threads re-timed with 3 warm-up(s) + 20 iterations")
    set_threads(cores)
    return sweep[best], best, {str(k): v for k, v in sweep.items()}, "median of timed passes, best thread count reported; " + "; ".join(notes)


def cpu_baselines(scene, scene_score, poses, score_pose, width, height, focal, gpu_field, gpu_est, dev):
    """BASELINE.md §4 through the oracle, plus the parity of the benchmarked scene (bench_parity).  Returns (cpu_baseline, bench_parity)."""
    import torch
    from apnrf_amd import render as RD
    from apnrf_amd import scenes as SC
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
    out = {"host_cpus": cores, "cpu_model": model, "kind": "port"}
    state = {"limit": None}

    def set_threads(n):
        torch.set_num_threads(n)
        if threadpool_limits is not None:
            if state["limit"] is not None:
                state["limit"].restore_original_limits()
            state["limit"] = threadpool_limits(limits=n)
    # (i) BL-1: 64x64 rays x 32 samples, frequency-PE field (numpy)
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
    best, th, sweep, proto = _thread_sweep(bl1, 4096, set_threads, 16.0)
    out["bl1_vanilla_64x64x32"] = {"rays_per_s": best, "threads": th, "sweep_rays_per_s": sweep,
                                  "protocol": proto + "; oracle/vanilla.py forward + batched compositing (numpy fp32)"}
    # (ii-a) the headline path on a bounded sub-sample of one benchmark view — and its parity against the GPU render of the same rays
    S_ = 24
    orc = _oracle_field(scene)
    idx = R.subsample_indices(width * height, S_ * S_)
    o, d = R.generate_image_rays(R.pose_to_c2w(poses[0]), width, height, focal, idx)
    bk = torch.zeros(3)
    ref = {}

    def headline():
        ref["r"] = R.render_test(1024, orc, scene["occ"], scene["aabb"][None], o, d, render_bkgd=bk, **SC.RENDER_KW)
    best, th, sweep, proto = _thread_sweep(headline, S_ * S_, set_threads, 30.0)
    out.update({"value": best, "unit": "rays/s", "cores": th, "sweep_rays_per_s": sweep,
                "sample": f"{S_}x{S_} linspace sub-sample of one 800x800 benchmark view, same trained weights and occupancy grid, "
                          f"oracle.render.render_test (fp32 torch-CPU hash grid + MLPs + occupancy marching); " + proto})
    got = RD.render_views(gpu_field, gpu_est, o.to(dev), d.to(dev), S_ * S_, 1024, render_bkgd=bk, **SC.RENDER_KW)
    r = ref["r"]
    errs = {k: (got[k].cpu() - r[k]).abs().reshape(S_ * S_, -1).max(dim=1).values.numpy() for k in ("rgb", "acc", "depth", "sem")}
    # rgb / acc / depth: 1e-3 absolute (north star).  The composited class logits are un-normalised sums of raw logits (up to ~55 on this scene):
    # their bar is max(1e-3, 3e-4 x largest |logit| of the ray) = three times the NOISE FLOOR measured right here — the same model through the
    # oracle a second time with every layer's fp32 products added in another order (oracle/field.py accum="k16_reversed"): what two faithful
    # fp16-operand / fp32-accumulate implementations differ by (tests/test_oracle_noise_floor_cpu.py: ~1e-4 of the logit at every scale).
    set_threads(int(th))       # (the sweep leaves its LAST thread count set: the oracle crawls with 64+ torch threads — one 576-ray pass took 20 minutes)
    r2 = R.render_test(1024, _oracle_field(scene, accum="k16_reversed"), scene["occ"], scene["aabb"][None], o, d, render_bkgd=bk, **SC.RENDER_KW)
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
              "noise_floor_abs": float(nf["sem"].max()),
              "tie_rays": int(tie.sum()), "tie_rays_max_abs": float(worst[tie].max()) if tie.any() else 0.0, "tie_budget": "<= 2 rays up to 5e-2",
              "psnr_db": float(10 * np.log10(1.0 / max(mse, 1e-20))), "total_samples_gpu": int(got["total"][0]), "total_samples_oracle": int(r["total_samples"]),
              "what": "HIP render vs oracle render of the SAME trained scene and rays (stand-alone call: the round schedule of a 576-ray call)"}
    parity["ok"] = bool(worst[~tie].max() <= 1e-3 and int(tie.sum()) <= 2 and (not tie.any() or worst[tie].max() <= 5e-2)
                        and abs(parity["total_samples_gpu"] - parity["total_samples_oracle"]) <= max(4, 2e-3 * parity["total_samples_oracle"]))
    # (ii-b) BASELINE.md §4 shape (ii): one 4096-ray scoring view (probabilistic) and one 2000-ray train step through the oracle
    set_threads(int(out["cores"]))
    orc_s = _oracle_field(scene_score)
    idx = R.subsample_indices(640 * 640, 4096)
    o, d = R.generate_image_rays(R.pose_to_c2w(score_pose), 640, 640, 320.0, idx)
    t = _timed(lambda: R.render_prob_test(1024, orc_s, scene_score["occ"], scene_score["aabb"][None], o, d, render_bkgd=bk, **SC.RENDER_KW), 0, 1)
    out["score_view_4096"] = {"rays_per_s": 4096 / t, "seconds": t, "threads": int(out["cores"]),
                              "sample": "one candidate view of BASELINE config 4 (64x64 linspace sub-sample of 640x640), probabilistic render, 1 iteration"}
    import torch.nn.functional as F
    orc_t = _oracle_field(scene_score, requires_grad=True)
    g = torch.Generator().manual_seed(5)
    TR = 250                            # an eighth of the reference yaml's 2000 rays: bounds the CPU time; samples scale with the rays
    idx = torch.randint(0, 640 * 640, (TR,), generator=g).numpy()
    o, d = R.generate_image_rays(R.pose_to_c2w(score_pose), 640, 640, 320.0, idx)
    pix, dep, lab = torch.rand(TR, 3, generator=g), torch.rand(TR, generator=g) * 4, torch.randint(0, 29, (TR,), generator=g)
    opt = torch.optim.Adam([orc_t.p_base, orc_t.p_head, orc_t.p_sem], lr=1e-3, eps=1e-15)
    n_s = {}

    def tstep():
        rr = R.render_train(orc_t, scene_score["occ"], scene_score["aabb"][None], 0.05, o, d, torch.full((TR,), 0.1), render_bkgd=bk,
                            render_step_size=1e-3, cone_angle=0.004, alpha_thre=0.01)
        loss = F.smooth_l1_loss(rr[0], pix) * 10 + F.smooth_l1_loss(rr[2], dep.unsqueeze(1)) / 5 + F.cross_entropy(rr[3], lab) / 2
        opt.zero_grad(); loss.backward(); opt.step(); orc_t._derive()
        n_s["n"] = rr[4]
    t = _timed(tstep, 0, 1)
    out["train_step_refyaml_eighth"] = {"ms": 1e3 * t, "rays": TR, "rendering_samples": int(n_s["n"]), "threads": int(out["cores"]),
                                        "ms_scaled_to_2000_rays": 1e3 * t * 2000 / TR,
                                        "sample": f"one train step of {TR} rays (1/8 of the reference yaml's 2000: bounded CPU time) through oracle autograd + "
                                                  "torch.optim.Adam, 1 iteration; the cost is linear in the samples"}
    set_threads(cores)
    return out, parity


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
    from apnrf_amd import _lib as L
    from apnrf_amd import distributed as DD
    from apnrf_amd import render as RD
    from apnrf_amd import scenes as SC
    from apnrf_amd import standin as SI

    lib = L.load_library()
    want = lambda w: args.workload in ("all", w)
    standin_info = {}

    opt_states = {}

    def scene_model(name, seed=9, steps=None, keep_optimizer=False):
        """(scene dict, field, estimator): the trained stand-in.  Rank 0 trains or loads the cache; the other ranks load the cache it
        wrote, or — when the file could not be written — receive the model by broadcast."""
        scene = SC.make_scene(name, n_poses=40)
        if args.weights == "random":
            return scene, SC.hip_field(scene, dev), SC.hip_estimator(scene, dev)
        steps = args.standin_steps if steps is None else steps
        if rank == 0:
            log(f"stand-in {name} seed {seed}: training / loading")
        field, est, info = SI.shared_standin(scene, dev, steps=steps, seed=seed, keep_optimizer=keep_optimizer, group=None if distributed else False,
                                             log=log if rank == 0 else None)
        opt_states[name] = info.pop("optimizer_state", None)
        if rank == 0:
            standin_info[f"{name}/seed{seed}"] = {k: info.get(k) for k in ("steps", "seconds", "loss_first", "loss_last", "skipped_steps",
                                                                          "occupied_cells", "cells", "cached")}
        return scene, field.eval(), est.eval()

    def timed(step_fn, steps, warmup, with_events, collect=None):
        """W warm-up steps, then exactly K steps bracketed by barrier + synchronize on both sides -> seconds (max over ranks)."""
        for i in range(warmup):
            step_fn(i)
        # Objects of earlier legs that sit in reference cycles (a torch optimizer and the field bound to it) are destroyed whenever Python's cycle
        # collector happens to run — `mnf_field_destroy` is a dozen `hipFree` calls, each of which waits for the device: inside a timed region that drains
        # the asynchronous pipeline.  This is what the sporadic 1.7-2x slow train legs were (profiles/r03_bench_line.json bf16 6.81 ms, a round-4 run
        # 8.20 ms f16 right behind the in-process stand-in training, per-kernel times normal both times): collect before the region, not inside it.
        gc.collect()
        torch.cuda.synchronize()
        if distributed:
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
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        gc.enable()
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
                                   f"{args.standin_steps} iterations, FusedAdam lr 2e-3 decayed to 2e-4 over the second half; occupancy grid from "
                                   "update_every_n_steps; seeded and trained with deterministic gradient accumulation: the same scene on every box")
                       if args.weights == "trained" else "random-init (hash U(-0.5,0.5), xavier MLPs, |density row| x 8), procedural occupancy"}}

    # ------------------------------------------------------------------ BASELINE config 3: 800x800 renders (the headline value)
    scene529 = field = est = None
    if want("render800"):
        scene529, field, est = scene_model("102344529")
        poses = scene529["poses"][[(5 * k + rank) % 40 for k in range(8)]]       # 8 views of the sweep per rank
        c2w = np.stack([RD.pose_to_c2w(p) for p in poses]).astype(np.float32)
        K = np.array([[focal, 0, width / 2], [0, focal, height / 2], [0, 0, 1.0]])
        rays = RD.generate_image_rays(torch.from_numpy(c2w), width, height, K, dev)
        n_per_view = width * height
        bk = torch.zeros(3)
        process_samples = torch.zeros((), dtype=torch.int64, device=dev)      # every evaluated sample of this process (PMC sums cover all launches)
        rfield, rest = field, est

        def render_pass(V, steps, warmup, with_events, n_split=2):
            batches = [(rays.origins[k:k + V].reshape(-1, 3).contiguous(), rays.viewdirs[k:k + V].reshape(-1, 3).contiguous())
                       for k in range(0, 8, V)]
            evaluated = torch.zeros((), dtype=torch.int64, device=dev)

            def step(i):
                o, d = batches[i % len(batches)]
                r = RD.render_views(rfield, rest, o, d, n_per_view, 1024, render_bkgd=bk, image_hw=(height, width), n_split=n_split, **SC.RENDER_KW)
                process_samples.add_(r["total"][1])
                return r

            def collect(r):
                evaluated.add_(r["total"][1])
            dt = timed(step, steps, warmup, with_events, collect)
            return dt, int(evaluated.item())

        V = args.views
        log("render800: timed pass")
        dt, samples = render_pass(V, args.steps, args.warmup, False, n_split=args.render_jobs)      # the reported value: no instrumentation
        log(f"render800: {1e3 * dt / args.steps:.2f} ms/step, {samples / (n_per_view * V * args.steps):.1f} samples/ray")
        line["value"] = n_per_view * V * world * args.steps / dt
        line["ms_per_step"] = 1e3 * dt / args.steps
        line["config"].update({"workload": f"BASELINE config 3: scene 102344529 (trained synthetic stand-in), 800x800 RGB+depth+29-class "
                                           f"semantic render, {V} view(s) per step in one batched call, hash-grid 16x4 T=2^19 + MLP 128x2 + "
                                           "64x2 heads", "views_per_step": V, "rays_per_step_per_gpu": n_per_view * V,
                               "ms_per_view": 1e3 * dt / args.steps / V, "samples_per_ray": samples / (n_per_view * V * args.steps),
                               "samples_per_s": samples * world / dt,
                               "render_jobs_in_flight": min(args.render_jobs, V),
                               "march_order": "8x8 pixel blocks inside every view (mnf_render_opts.view_order); per-ray results do not depend on it"})
        if not args.no_kernel_timing:
            # second pass of the same K steps, ONE job in flight, hipEvent pairs around every field-kernel launch (hipEventRecord
            # between dependent launches costs up to ~0.15 ms each on this stack, so it is kept out of the pass that yields `value`)
            dt2, samples2 = render_pass(V, args.steps, 1, True, n_split=1)
            field_ms, launches = prof("field_render")
            if launches and samples2:
                achieved = ALGO_BYTES_PER_SAMPLE * samples2 / (field_ms * 1e-3) / 1e9
                traffic, traffic_note = None, "no PMC profile of this kernel build is committed (profiles/r05_pmc.json)"
                pj = os.path.join(REPO, "profiles", "r05_pmc.json")
                if os.path.exists(pj):
                    pm = json.load(open(pj))
                    if pm.get("field_sources_md5") == field_source_id():
                        traffic = pm["field_kernel"]["hbm_bytes_per_sample"] * samples2 / launches
                        traffic_note = "rocprofv3 --pmc passes of this kernel build (profiles/r05_pmc.json: FETCH_SIZE + WRITE_SIZE per evaluated sample), scaled to this run's samples per launch"
                    else:
                        traffic_note = "profiles/r05_pmc.json was measured on a different build of the kernel sources: not quoted"
                line["roofline"] = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                                    "traffic": traffic, "traffic_source": traffic_note,
                                    "kernel": "mnf::field_kernel<128,2,2,false> (hash gather + MLPs + fused compositing)",
                                    "avg_launch_ms": field_ms / launches, "launches": int(launches), "samples_per_launch": samples2 / launches,
                                    "algorithmic_bytes_per_sample": ALGO_BYTES_PER_SAMPLE,
                                    "algorithmic_bytes_per_launch": ALGO_BYTES_PER_SAMPLE * samples2 / launches,
                                    "field_kernel_share_of_serial_step": field_ms * 1e-3 / dt2,
                                    "timing": "second pass of the same K steps with one render job in flight (launches back to back on one stream), "
                                              "hipEvent pair around each launch on the launch stream"}
        if V != 1 and not args.no_views1:
            dt1, s1 = render_pass(1, args.steps, 2, False)
            line["render_views1"] = {"value": n_per_view * world * args.steps / dt1, "unit": "rays/s", "ms_per_view": 1e3 * dt1 / args.steps,
                                     "samples_per_ray": s1 / (n_per_view * args.steps)}
        if args.weights == "trained" and not args.no_views1:
            # the same views with round 1's engineered random-init weights, reported beside the headline value, never as it
            rfield, rest = SC.hip_field(scene529, dev), SC.hip_estimator(scene529, dev)
            dtr, sr = render_pass(V, args.steps, 2, False)
            rfield, rest = field, est
            line["render_random_weights"] = {"value": n_per_view * V * world * args.steps / dtr, "unit": "rays/s",
                                             "ms_per_step": 1e3 * dtr / args.steps, "views_per_step": V,
                                             "samples_per_ray": sr / (n_per_view * V * args.steps), "samples_per_s": sr * world / dtr,
                                             "note": "synthetic.make_field_params seed 0, procedural occupancy grid: round 1's headline configuration"}
        if args.weights == "trained" and not args.no_views1 and not args.no_kernel_timing:
            # the same trained weights evaluated with tiny-cuda-nn's fp16 hash blend (mnf_field_config.blend_fp16): never the headline
            bfield = SC.hip_field(scene529, dev, tcnn_blend_fp16=True)
            bfield.load_state_dict(field.state_dict())
            rfield = bfield.eval()
            dtb, sb = render_pass(V, args.steps, 2, False, n_split=args.render_jobs)
            dtb2, sb2 = render_pass(V, args.steps, 1, True, n_split=1)
            bms, bl = prof("field_render")
            rfield = field
            line["render_blend_fp16"] = {"value": n_per_view * V * world * args.steps / dtb, "unit": "rays/s", "ms_per_step": 1e3 * dtb / args.steps,
                                         "samples_per_ray": sb / (n_per_view * V * args.steps), "samples_per_s": sb * world / dtb,
                                         "field_kernel_avg_launch_ms": bms / max(bl, 1), "field_kernel_samples_per_launch": sb2 / max(bl, 1),
                                         "field_kernel_frac_of_hbm_peak": ALGO_BYTES_PER_SAMPLE * sb2 / max(bms * 1e-3, 1e-9) / 1e9 / HBM_PEAK_GBS,
                                         "note": "the headline scene and weights with the hash levels' 8-corner blend as fp16 fused multiply-adds "
                                                 "(tcnn's T = __half arithmetic as published); the stand-in was trained with the fp32 blend, so the sample "
                                                 "counts differ slightly"}
        line["samples"] = {"timed": int(samples), "process_total": int(process_samples.item())}
        del rays

    # ------------------------------------------------------------------ BASELINE config 5: train step (fp16 and bf16 operands) + the reference yaml's shape
    if want("train"):
        from apnrf_amd.optim import FusedAdam
        scene280, tfield0, test0 = scene_model("102344280", seed=11, keep_optimizer=True)
        proc = SI._procedural_estimator(scene280, dev)
        c2w = np.stack([RD.pose_to_c2w(p) for p in scene280["poses"][:8]]).astype(np.float32)
        K6 = np.array([[320.0, 0, 320], [0, 320.0, 320], [0, 0, 1.0]])

        def make_batches(R_):
            g = torch.Generator(device="cpu").manual_seed(100 + rank)
            out = []
            for k in range(8):
                idx = torch.randint(0, 640 * 640, (R_,), generator=g).numpy()
                ys, xs = idx // 640, idx % 640                      # grouped by 32x32 image block, as dataset.Dataset.fetch_data does
                idx = idx[np.argsort((ys // 32) * 20 + xs // 32, kind="stable")]
                r = RD.generate_image_rays(torch.from_numpy(c2w[k:k + 1]), 640, 640, K6, dev, idx)
                out.append((r,) + SI.analytic_targets(proc, scene280["aabb"], r.origins, r.viewdirs))
            return out

        def train_leg(dtype, R_, sync, steps, with_kernels, dynamic_target=0, presample=False):
            """ms per step of `steps` train iterations from the SAME start state (the stand-in's weights and grid are restored before every leg;
            the optimizer continues the stand-in's run); returns a dict."""
            tf = SC.hip_field(scene280, dev, mfma_bf16=(dtype == "bf16"))
            tf.load_state_dict(tfield0.state_dict())
            from apnrf_amd.nerfacc import OccGridEstimator
            te = OccGridEstimator(torch.from_numpy(scene280["aabb"]), resolution=scene280["res"], levels=1).to(dev)
            te.occs.copy_(test0.occs); te.binaries = test0.binaries.clone()
            tf.train(); te.train()
            # the stand-in's own training run CONTINUES: its Adam moments, step count and final learning rate (2e-4).  A fresh Adam would
            # kick every parameter with a non-zero gradient by +-lr in its first steps (the sample count tripled within the timed window)
            opt = FusedAdam(tf.parameters(), lr=2e-4, eps=1e-15).bind_field(tf)
            if opt_states.get("102344280") is not None:
                import copy
                opt.load_state_dict(copy.deepcopy(opt_states["102344280"]))
                for g_ in opt.param_groups:
                    g_["lr"] = 2e-4
            batches = make_batches(R_)
            bkd = torch.rand(3, generator=torch.Generator().manual_seed(7)).to(dev)      # a random background colour on the device (habitat_to_data.py:189-191)
            outs = []

            dyn = {"R": 1024, "seen": []}                                        # pipeline.py:418 starts the data set at 1024 rays

            def tstep_dynamic(i):
                """scripts/pipeline.py:494-504 without a host round trip: the ray count of the next step follows the latest sample count that has
                ARRIVED on the host (RD.latest_step_counts: one or two steps old), capped at 2000 as the reference caps it."""
                r, pix, dep_, lab = batches[i % 8]
                n = dyn["R"]
                rr = RD.Rays(r.origins[:n], r.viewdirs[:n])
                out = RD.train_step(tf, te, opt, rr, pix[:n], dep_[:n], lab[:n], bkd, step=1000 + i, sync=False, occ_thre=1e-2, **SC.RENDER_KW)
                dyn["seen"].append(n)
                c = RD.latest_step_counts(tf)
                if c is not None and c[2] > 0:
                    dyn["R"] = int(min(R_, max(64, c[0] * dynamic_target / c[2])))
                return out

            pre = {"tok": None, "adopted": 0}

            def tstep_presampled(i):
                """The batch fetched one iteration early (render.presample): the march of batch i + 1 is enqueued in front of step i and runs beside it on the library's
                side stream; step i adopts the march made in front of step i - 1.  Not across an occupancy refresh (steps 1008 + 16 j): those steps march themselves."""
                r, pix, dep_, lab = batches[i % 8]
                s_ = 1000 + i
                nxt = RD.presample(tf, te, batches[(i + 1) % 8][0], **SC.RENDER_KW) if s_ % 16 and (s_ + 1) % 16 else None
                tok = pre["tok"]
                out = RD.train_step(tf, te, opt, r, pix, dep_, lab, bkd, step=s_, sync=False, occ_thre=1e-2, presampled=tok, **SC.RENDER_KW)
                pre["adopted"] += int(tok is not None and tok.adopted)
                pre["tok"] = nxt
                return out

            def tstep(i):
                if dynamic_target:
                    return tstep_dynamic(i)
                if presample:
                    return tstep_presampled(i)
                r, pix, dep_, lab = batches[i % 8]
                # occ_thre as the stand-in's own training (the reference uses 1e-3 / 1e-2 / 3e-3 by phase, pipeline.py:447-470): the refresh at
                # step 1008 then keeps the grid the stand-in converged to, and the workload stays stationary
                return RD.train_step(tf, te, opt, r, pix, dep_, lab, bkd, step=1000 + i, sync=sync, occ_thre=1e-2, **SC.RENDER_KW)
            dt_t = timed(tstep, steps, max(args.warmup, 6), False, outs.append)      # (the first asynchronous steps also size the sample bounds)
            kept = float(np.mean([int(o["n_rendering_samples"]) for o in outs]))
            marched = float(int(te.last_sampling["n_marched"]))
            if dynamic_target:
                seen = dyn["seen"][-steps:]
                return {"ms_per_step": 1e3 * dt_t / steps, "steps": steps, "dtype": dtype, "host_round_trips_per_step": 0,
                        "rays_per_step_mean": float(np.mean(seen)), "rays_per_step_min_max": [int(min(seen)), int(max(seen))],
                        "distinct_ray_counts": len(set(seen)), "target_sample_batch_size": dynamic_target,
                        "rendering_samples_per_step": kept, "skipped_steps": int(sum(int(o["skipped"]) for o in outs)),
                        "train_states_of_the_field": 1, "overflowed_steps": RD._TRAIN_STATE[id(tf)].get("overflowed_steps", 0)}
            res = {"ms_per_step": 1e3 * dt_t / steps, "steps": steps, "rays_per_step": R_, "dtype": dtype,
                   "host_round_trips_per_step": 1 if sync else 0, "rendering_samples_per_step": kept, "marched_samples_per_step": marched,
                   "skipped_steps": int(sum(int(o["skipped"]) for o in outs))}
            if presample:
                res["marches_adopted_of_steps"] = [pre["adopted"], steps + max(args.warmup, 6)]
                return res
            # how sparse the hash-table gradient of one step is (decides whether a touched-rows exchange could beat the dense all-reduce of
            # ray-data-parallel training): fraction of the table's entries with a non-zero gradient after the last step
            n_mlp = tf.mlp_base.params.numel() - 4 * tf._table_entries()
            g_tab = tf.mlp_base.params.grad[n_mlp:].view(-1, 4)
            res["table_entries_touched_fraction"] = float((g_tab != 0).any(dim=1).float().mean())
            algo = TRAIN_BYTES_KEPT * kept + TRAIN_BYTES_MARCHED * marched
            traffic, traffic_note = train_traffic(R_) if dtype == "f16" else (None, "the counter passes were taken with fp16 operands")
            res["roofline"] = {"bound": "hbm", "achieved": algo / (dt_t / steps) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": algo / (dt_t / steps) / 1e9 / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_note,
                               "traffic_over_algorithmic": (traffic / algo) if traffic else None, "algorithmic_bytes_per_step": algo,
                               "definition": "(4.1 KB x surviving samples + 1 KB x pre-pass samples) / un-instrumented step time (SURVEY 8d)"}
            if with_kernels and not args.no_kernel_timing:
                outs.clear()
                dt_e = timed(tstep, steps, 0, True, outs.append)
                kept_e = float(np.mean([int(o["n_rendering_samples"]) for o in outs]))
                kernels, total_ms = {}, 0.0
                per_sample = {"field_density": ("marched", 1036), "field_train_forward": ("kept", 1036)}
                for label in ("sample_rays", "field_density", "field_train_forward", "composite_train_forward", "composite_train_backward",
                              "dgrad", "wgrad", "hash_scatter", "hash_scatter_bins"):
                    ms, n = prof(label)
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
                if "hash_scatter" in kernels:
                    kernels["hash_scatter"]["note"] = ("levels 0-11: the walk, bound by the memory-side atomic unit: ~21 G 64-byte requests/s whatever they carry "
                                                       "(profiles/r03_atomic_microbench.txt), on a second stream beside wgrad; levels 12-15 (hash_scatter_bins): per-bin "
                                                       "item lists through HBM + LDS sums, on a third stream; the three overlap, so their times do not add up")
                res["kernels"] = kernels
                res["timed_kernels_ms_per_step"] = total_ms
                res["instrumented_step_ms"] = 1e3 * dt_e / steps
            return res

        def ensemble_leg(R_, steps):
            """Two ensemble members (the reference trains an ensemble of two, one member after the other inside every iteration: pipeline.py:398-412) stepped in turn on one
            stream and side by side on one stream each (`render.train_step_ensemble`): ms per iteration (= one step of each member) and per member step.  Both members start
            from the stand-in's state (the second stand-in would cost another 12 s of the bench; the timing does not depend on the weights being different)."""
            from apnrf_amd.nerfacc import OccGridEstimator
            import copy
            mem = []
            for _ in range(2):
                tf = SC.hip_field(scene280, dev)
                tf.load_state_dict(tfield0.state_dict())
                te = OccGridEstimator(torch.from_numpy(scene280["aabb"]), resolution=scene280["res"], levels=1).to(dev)
                te.occs.copy_(test0.occs); te.binaries = test0.binaries.clone()
                tf.train(); te.train()
                opt = FusedAdam(tf.parameters(), lr=2e-4, eps=1e-15).bind_field(tf)
                if opt_states.get("102344280") is not None:
                    opt.load_state_dict(copy.deepcopy(opt_states["102344280"]))
                    for g_ in opt.param_groups:
                        g_["lr"] = 2e-4
                mem.append((tf, te, opt))
            batches = make_batches(R_)
            bkd = torch.rand(3, generator=torch.Generator().manual_seed(7)).to(dev)
            res = {"rays_per_member_step": R_, "members": 2, "steps": steps}

            def turn(i):
                return [RD.train_step(tf, te, opt, *((batches[(i + 3 * m) % 8][0],) + tuple(batches[(i + 3 * m) % 8][1:])), bkd, step=1000 + i, sync=False, occ_thre=1e-2, **SC.RENDER_KW)
                        for m, (tf, te, opt) in enumerate(mem)]

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
                dt_e = timed(fn, steps, 6, False, outs.append)
                res[label] = {"ms_per_iteration": 1e3 * dt_e / steps, "ms_per_member_step": 1e3 * dt_e / steps / 2,
                              "rendering_samples_per_member_step": float(np.mean([int(o["n_rendering_samples"]) for pair in outs for o in pair])),
                              "skipped_steps": int(sum(int(o["skipped"]) for pair in outs for o in pair))}
            res["speedup"] = res["one_stream"]["ms_per_iteration"] / res["stream_per_member"]["ms_per_iteration"]
            res["speedup_presampled"] = res["one_stream"]["ms_per_iteration"] / res["stream_per_member_next_batch_presampled"]["ms_per_iteration"]
            return res

        def dropin_leg(R_, steps):
            """The UNCHANGED caller: scripts/pipeline.py:472-532 typed against the drop-in names only — `render_image_with_occgrid_with_depth_guide` (autograd), torch losses,
            `loss.backward()`, the per-parameter `torch.isnan` loop with its host round trips, `torch.optim.Adam.step()` and the reference's scheduler — on the same scene,
            start state and batches as `train_refyaml`.  What NOT editing pipeline.py costs against `render.train_step` (one fused C call + FusedAdam)."""
            import torch.nn.functional as F
            from apnrf_amd.nerfacc import OccGridEstimator
            from apnrf_amd import nerfacc as NA
            tf = SC.hip_field(scene280, dev)
            tf.load_state_dict(tfield0.state_dict())
            te = OccGridEstimator(torch.from_numpy(scene280["aabb"]), resolution=scene280["res"], levels=1).to(dev)
            te.occs.copy_(test0.occs); te.binaries = test0.binaries.clone()
            tf.train(); te.train()
            optimizer = torch.optim.Adam(tf.parameters(), lr=2e-4, eps=1e-15, weight_decay=0.0)                      # pipeline.py:173-178
            scheduler = torch.optim.lr_scheduler.ChainedScheduler([torch.optim.lr_scheduler.CyclicLR(
                optimizer, base_lr=1e-4, max_lr=2e-4, step_size_up=250, mode="exp_range", gamma=1.0, cycle_momentum=False)])   # pipeline.py:183-193's form
            occ_eval_fn = NA.FieldDensityOcc(tf, 1e-3)                                                               # pipeline.py:376-378
            batches = make_batches(R_)
            bkd = torch.rand(3, generator=torch.Generator().manual_seed(7)).to(dev)
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
            dt_d = timed(step, steps, max(args.warmup, 4), False)
            return {"ms_per_step": 1e3 * dt_d / steps, "steps": steps, "rays_per_step": R_, "rendering_samples_per_step": float(np.mean(stats["n"][-steps:])),
                    "steps_jumped": stats["jumped"], "host_round_trips_per_step": "sample count (inside sampling) + n_rendering_samples + 3 losses + one per parameter vector",
                    "what": "pipeline.py:472-532 unchanged on the drop-in surface: autograd route (same kernels call by call), torch smooth_l1 / cross_entropy, loss.backward(), "
                            "per-parameter isnan round trips, torch.optim.Adam + CyclicLR"}

        tsteps = max(args.steps, 10)
        dtypes = [d for d in args.train_dtypes.split(",") if d in ("f16", "bf16")]
        presample_note = ("next_batch_presampled: the same steps with the batch fetched one iteration early and its march (occ_grid.py:181-208: reads rays and grid, not the model) "
                          "enqueued in front of the current step on a library side stream (render.presample / mnf_train_presample); bit-identical results "
                          "(tests/test_gpu_round4.py), every march inside the timed region; steps next to an occupancy refresh march themselves")
        train = {"presample": presample_note, "workload": "BASELINE config 5: scene 102344280 (trained stand-in, training continued from the same state in every leg), 8192-ray batches "
                             "of one 640x640 view, occupancy sampling + density pre-pass + differentiable render + loss (pipeline.py:506-511) + backward "
                             "+ NaN guard + FusedAdam; occupancy refresh every 16th step"}
        for dt_ in dtypes:
            log(f"train {dt_}: timed passes")
            leg = train_leg(dt_, args.train_rays, False, tsteps, True)
            leg["host_synchronous"] = {k: v for k, v in train_leg(dt_, args.train_rays, True, tsteps, False).items()
                                       if k in ("ms_per_step", "rendering_samples_per_step", "host_round_trips_per_step")}
            leg["next_batch_presampled"] = {k: v for k, v in train_leg(dt_, args.train_rays, False, tsteps, False, presample=True).items()
                                            if k in ("ms_per_step", "rendering_samples_per_step", "marches_adopted_of_steps", "skipped_steps")}
            train[dt_] = leg
            log(f"train {dt_}: {leg['ms_per_step']:.2f} ms/step at {leg['rendering_samples_per_step']:.0f} samples (host-synchronous {leg['host_synchronous']['ms_per_step']:.2f}, "
                f"next batch presampled {leg['next_batch_presampled']['ms_per_step']:.2f})")
        first = train[dtypes[0]]
        train.update({k: first[k] for k in ("ms_per_step", "rays_per_step", "rendering_samples_per_step", "marched_samples_per_step", "roofline")})
        train["ms_per_step_next_batch_presampled"] = first["next_batch_presampled"]["ms_per_step"]
        train["dtype"] = dtypes[0]
        ry = train_leg("f16", 2000, False, tsteps, True)
        ry["host_synchronous_ms_per_step"] = train_leg("f16", 2000, True, tsteps, False)["ms_per_step"]
        ry["next_batch_presampled"] = {k: v for k, v in train_leg("f16", 2000, False, max(tsteps, 40), False, presample=True).items()
                                       if k in ("ms_per_step", "rendering_samples_per_step", "marches_adopted_of_steps", "skipped_steps")}
        ry["workload"] = ("the reference yaml's own shape: 2000 rays per step (scripts/config_102344250.yaml:3, the cap of pipeline.py:494-504), target "
                          "262 144 samples (config:4); same scene and start state")
        if "kernels" in ry:
            ry["fixed_cost_share"] = 1.0 - sum(v["ms_per_step"] for k, v in ry["kernels"].items() if k in ("field_density", "field_train_forward", "dgrad", "wgrad")) / ry["ms_per_step"]
        dy = train_leg("f16", 2000, False, max(tsteps, 40), False, dynamic_target=1 << 18)
        dy["workload"] = ("the reference's own schedule (scripts/pipeline.py:494-504, config_102344250.yaml:3-4): num_rays starts at 1024 and is recomputed after "
                          "every iteration to hold 262 144 samples, capped at 2000; asynchronous steps, the count used is the latest that has arrived on the host")
        log("train: the drop-in surface (autograd route + torch.optim.Adam)")
        dropin = dropin_leg(2000, max(tsteps, 20))
        dropin["fused_async_ms_per_step"] = ry["ms_per_step"]
        dropin["fused_host_synchronous_ms_per_step"] = ry["host_synchronous_ms_per_step"]
        dropin["cost_of_not_editing_pipeline_py"] = dropin["ms_per_step"] / ry["host_synchronous_ms_per_step"]
        line["train_dropin"] = dropin
        line["train"] = train
        line["train_refyaml"] = ry
        line["train_dynamic"] = dy
        line["train_ensemble2"] = {"workload": "an ensemble of two members (the reference's), both stepped in every iteration: in turn on one stream (the reference's loop) and side by side, "
                                               "one stream per member (render.train_step_ensemble); asynchronous steps, same scene and start state as the train legs",
                                   "refyaml_2000_rays": ensemble_leg(2000, max(tsteps, 20)), "config5_8192_rays": ensemble_leg(args.train_rays, tsteps)}
        if not want("render800"):
            line.update({"metric": "train-step ms", "value": train["ms_per_step"], "unit": "ms", "higher_is_better": False,
                         "ms_per_step": train["ms_per_step"], "dtype": dtypes[0]})
            line["config"]["workload"] = train["workload"]
        del tfield0, test0

    # ------------------------------------------------------------------ BASELINE config 4: candidate-view scoring, views sharded over ranks
    scene250 = None
    if want("score256"):
        scene250, f0, e0 = scene_model("102344250", seed=9)
        _, f1, e1 = scene_model("102344250", seed=10)
        poses256 = SI._free_space_poses(scene250, 256, seed=9)                  # 8 trajectories x 32 views inside the free space
        group = dist.group.WORLD if distributed else None
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
        log("score256: timed pass")
        dt_s = timed(sstep, ssteps, 1, False)
        log(f"score256: {1e3 * dt_s / ssteps:.2f} ms/pass")
        terms, score = sstep(0)
        evaluated = float(sum(int(t[1]) for t in RD.LAST_SCORE_TOTALS))          # this rank's share of the views, all members
        timed(sstep_parts, 3, 1, False)
        parts = torch.tensor([t_parts["compute"] / t_parts["n"], t_parts["gather"] / t_parts["n"]], dtype=torch.float64, device=dev)
        per_rank = [parts.clone() for _ in range(world)]
        if distributed:
            dist.all_gather(per_rank, parts)
        sc = {"ms_per_pass": 1e3 * dt_s / ssteps, "rays_per_s": 256 * 4096 * 2 * ssteps / dt_s, "views": 256, "rays_per_view": 4096,
              "ensemble_members": 2, "n_gpus": world, "scaling": "strong", "score": float(score),
              "samples_per_ray_rank0": evaluated / max(1, (256 // world) * 4096 * 2),
              "samples_per_s_rank0": evaluated * ssteps / dt_s,
              "per_rank_compute_ms": [1e3 * float(p[0]) for p in per_rank], "per_rank_gather_ms": [1e3 * float(p[1]) for p in per_rank],
              "render_jobs_in_flight": 4,
              "collective": "one all_gather_into_tensor of [V/N,4] float64 per pass" if world > 1 else "none (single rank)",
              "workload": "BASELINE config 4: scene 102344250 (two trained stand-ins: seeds 9 and 10), 256 candidate poses in free "
                          "space, 64x64 rays each (linspace sub-sample of 640x640), probabilistic render + predictive-information terms; the two "
                          "ensemble members advance side by side, each cut into two groups of views: four render jobs of one call (the caller's stream + three shared side streams)"}
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
            dt8 = timed(shard_step, ssteps, 1, False)
            t8, _ = shard_step(0)
            ev8 = float(sum(int(t[1]) for t in RD.LAST_SCORE_TOTALS))
            line["score256_shard8"] = {"ms_per_pass": 1e3 * dt8 / ssteps, "views": hi - lo, "samples_per_s": ev8 * ssteps / dt8,
                                       "ratio_to_full_over_8": (dt8 / ssteps) / (dt_s / ssteps / 8),
                                       "bit_identical_to_full_pass_rows": bool(torch.equal(t8, terms[lo:hi])),
                                       "note": "views 0..31 of the same pass on one GPU = the per-rank work of --gpus 8; predicted 8-GPU pass = this + one 8 KB all-gather"}
        sc["process_samples"] = int(score_samples.item())
        if world == 1:
            # The drop-in surface scripts/pipeline.py calls today, float64 host stacks and D2H copies included:
            #   pipeline.py:960-974   Dataset.render_image_from_pose(field, est, traj, img_w, img_h, focal, near, step, 1, cone, alpha, 1, device): full 640x640 views
            #   pipeline.py:697-711   Dataset.render_probablistic_image_from_pose(member, est, trajectory[unc_idx] (40 poses), ..., scale 0.1, ..., 4, device), once per member
            from apnrf_amd.dataset import Dataset
            p1 = poses256[:1]
            p40 = poses256[:40]
            a_full = (f0, e0, p1, 640, 640, 320.0, 0.1, 1e-3, 1, 0.004, 0.01, 1, dev)
            a_40 = lambda f_, e_: (f_, e_, p40, 640, 640, 320.0, 0.1, 1e-3, 0.1, 0.004, 0.01, 4, dev)
            dt_full = timed(lambda i: Dataset.render_image_from_pose(*a_full), 3, 1, False)
            dt_40 = timed(lambda i: [Dataset.render_probablistic_image_from_pose(*a_40(f_, e_)) for f_, e_ in ((f0, e0), (f1, e1))], 3, 1, False)
            dt_40s = timed(lambda i: RD.score_views([f0, f1], [e0, e1], p40, 640, 640, 320.0, 0.1, 1e-3, 0.1, 0.004, 0.01, dev, group=False)[1].item(), 3, 1, False)
            line["render_from_pose"] = {
                "full_view_640x640": {"ms_per_call": 1e3 * dt_full / 3, "rays_per_s": 640 * 640 * 3 / dt_full, "poses": 1,
                                      "what": "Dataset.render_image_from_pose as pipeline.py:960-974 calls it (scale 1): -> numpy float64 [1,640,640,.] stacks"},
                "uncertainty_40_poses_x2_members": {"ms_per_trajectory": 1e3 * dt_40 / 3, "rays_per_s": 2 * 40 * 4096 * 3 / dt_40, "poses": 40, "members": 2,
                                                    "what": "Dataset.render_probablistic_image_from_pose as pipeline.py:697-711 calls it (scale 0.1, once per ensemble "
                                                            "member): -> six numpy float64 [40,64,64,.] stacks per member, the scorer's numpy then runs on the host"},
                "same_40_poses_on_device_scorer_ms": 1e3 * dt_40s / 3,
                "note": "the second entry against the third is what the host stacks cost a trajectory score: the renders are the same kernels"}
            line["render_from_pose"]["source"] = "render.render_image_from_pose / render_probablistic_image_from_pose: pose -> rays -> render -> .cpu().numpy() float64 stacks"
        if not want("render800") and not want("train"):
            line.update({"metric": "candidate-view scoring rays/s", "value": sc["rays_per_s"], "ms_per_step": sc["ms_per_pass"], "scaling": "strong"})
            line["config"]["workload"] = sc["workload"]

    rc = 0
    if rank == 0:
        if standin_info:
            line["config"]["standin_training"] = standin_info
        if world == 1 and not args.no_cpu_baseline and scene529 is not None and args.weights == "trained":
            def trained_scene(scene, f, e):
                s_ = dict(scene)
                s_["params"] = {"mlp_base": f.mlp_base.params.detach().cpu().numpy(), "mlp_head": f.mlp_head.params.detach().cpu().numpy(),
                                "mlp_sem": f.mlp_sem.params.detach().cpu().numpy()}
                s_["occ"] = e.binaries.cpu().numpy()
                return s_
            if scene250 is None:
                scene250, f0, e0 = scene_model("102344250", seed=9)
                poses256 = SI._free_space_poses(scene250, 256, seed=9)
            log("cpu baseline + bench parity")
            line["cpu_baseline"], line["bench_parity"] = cpu_baselines(trained_scene(scene529, field, est), trained_scene(scene250, f0, e0), scene529["poses"][[0]],
                                                                      poses256[0], width, height, focal, field, est, dev)
            log(f"cpu baseline done; bench parity ok = {line['bench_parity']['ok']}")
            if not line["bench_parity"]["ok"]:
                rc = 3
        print(json.dumps(line))
    if distributed:
        dist.destroy_process_group()
    if rc:
        sys.exit(rc)


if __name__ == "__main__":
    main()
