"""Generate the golden vectors under tests/golden/ by IMPORTING THE REFERENCE.

Run in the build container only (it needs /root/reference, which does not exist on the GPU box):

    python tests/golden/make_golden.py

What is captured (all from the reference's own, CPU-runnable code; nothing is re-typed):

  aabb.npz      nerfacc.grid._ray_aabb_intersect            (perception/nerfacc/nerfacc/grid.py:54-90)
  volrend.npz   nerfacc.volrend.render_weight_from_density / render_visibility_from_density /
                rendering, batched [R,S] branch + autograd grads (volrend.py:20-161, :315-365, :424-483)
  occgrid.npz   nerfacc.OccGridEstimator.__init__/_update trajectories with a deterministic
                occ_eval_fn; the torch RNG draws are recorded (occ_grid.py:28-78, :328-437);
                mark_invisible_cells counts (tests/test_grid.py:207-233 known answer 77660 / 53412)
  query.npz     nerfacc.grid._query on sample positions produced by the ORACLE marcher
                (grid.py:201-237) — the reference's own "samples lie in occupied cells" property
                (tests/test_grid.py:39-68) evaluated by the reference's function.
  vanilla.npz   radiance_fields.mlp.VanillaNeRFRadianceField(2, 64, None, 1, 64) + nerfacc.rendering
                on BASELINE config 1 (64x64 rays x 32 samples): state_dict, outputs, loss, grads
                (perception/models/radiance_fields/mlp.py:168-245)
  dataset.npz   Dataset.update_data + evaluation-mode fetch_data/preprocess on a tiny synthetic set
                (perception/data_proc/habitat_to_data.py:89-272), same placeholder-module note as raygen.npz
  raygen.npz    Dataset.generate_image_rays + the linspace sub-sampler
                (perception/data_proc/habitat_to_data.py:274-301, :462-467).  The module imports
                imageio / cv2 / skimage at top level (unused by this function, absent here); empty
                placeholder modules are registered for those three names so the import succeeds.
"""
import os
import sys
import types

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
OUT = os.path.join(REPO, "tests", "golden")
REF = "/root/reference"
sys.path.insert(0, REPO)


def _enter_reference():
    os.chdir(REF)  # the fork does sys.path.append("perception/nerfacc/nerfacc") relative to cwd
    for p in ("perception/nerfacc", "perception/models", "perception/data_proc"):
        sys.path.insert(0, os.path.join(REF, p))


def gen_aabb():
    from nerfacc.grid import _ray_aabb_intersect
    torch.manual_seed(42)
    n_rays, n_aabbs = 256, 16
    rays_o = torch.rand((n_rays, 3)) * 3 - 1
    rays_d = torch.randn((n_rays, 3))
    rays_d = rays_d / rays_d.norm(dim=-1, keepdim=True)
    # a few axis-parallel rays and origins inside boxes
    rays_d[0] = torch.tensor([1.0, 0.0, 0.0]); rays_d[1] = torch.tensor([0.0, -1.0, 0.0]); rays_d[2] = torch.tensor([0.0, 0.0, 1.0])
    aabb_min = torch.rand((n_aabbs, 3))
    aabb_max = aabb_min + torch.rand_like(aabb_min)
    aabbs = torch.cat([aabb_min, aabb_max], -1)
    rays_o[3] = (aabb_min[0] + aabb_max[0]) / 2
    out = {}
    for tag, (near, far) in {"inf": (-float("inf"), float("inf")), "clip": (0.1, 1.5)}.items():
        t0, t1, h = _ray_aabb_intersect(rays_o, rays_d, aabbs, near, far)
        out[f"t_mins_{tag}"], out[f"t_maxs_{tag}"], out[f"hits_{tag}"] = t0.numpy(), t1.numpy(), h.numpy()
    np.savez_compressed(os.path.join(OUT, "aabb.npz"), rays_o=rays_o.numpy(), rays_d=rays_d.numpy(), aabbs=aabbs.numpy(), **out)


def gen_volrend():
    from nerfacc.volrend import render_weight_from_density, render_visibility_from_density, rendering
    torch.manual_seed(7)
    R, S = 64, 48
    t_starts = torch.sort(torch.rand(R, S) * 4 + 0.1, -1).values
    t_ends = t_starts + torch.rand(R, S) * 0.05 + 1e-3
    sigmas = (torch.rand(R, S) * 8).requires_grad_(True)
    rgbs = torch.rand(R, S, 3, requires_grad=True)
    w, tr, al = render_weight_from_density(t_starts, t_ends, sigmas)
    vis = render_visibility_from_density(t_starts, t_ends, sigmas.detach(), early_stop_eps=1e-2, alpha_thre=0.05)
    bkgd = torch.tensor([0.2, 0.5, 0.9])
    colors, opac, depths, _ = rendering(t_starts, t_ends, rgb_sigma_fn=lambda a, b, c: (rgbs, sigmas), render_bkgd=bkgd)
    loss = (colors ** 2).sum() + depths.sum() + 0.5 * opac.sum()
    loss.backward()
    np.savez_compressed(os.path.join(OUT, "volrend.npz"), t_starts=t_starts.numpy(), t_ends=t_ends.numpy(),
                        sigmas=sigmas.detach().numpy(), rgbs=rgbs.detach().numpy(), bkgd=bkgd.numpy(),
                        weights=w.detach().numpy(), trans=tr.detach().numpy(), alphas=al.detach().numpy(), vis=vis.numpy(),
                        colors=colors.detach().numpy(), opacities=opac.detach().numpy(), depths=depths.detach().numpy(),
                        grad_sigmas=sigmas.grad.numpy(), grad_rgbs=rgbs.grad.numpy())


def gen_occgrid():
    from nerfacc import OccGridEstimator
    import nerfacc.estimators.occ_grid as og
    aabb = torch.tensor([-19.1, -0.2, -19.1, 0.5, 3.2, 0.5])
    res = [12, 5, 9]
    est = OccGridEstimator(roi_aabb=aabb, resolution=res, levels=1)
    est.train()
    rec = {"aabbs": est.aabbs.numpy(), "grid_coords": est.grid_coords.numpy()}

    def occ_eval_fn(x):  # deterministic "density * step"
        return (torch.sin(x[:, :1] * 1.3) * torch.cos(x[:, 2:3] * 0.7) + 0.2 * x[:, 1:2]).clamp_min(0) * 0.02

    draws = []
    real_rand_like, real_randint = torch.rand_like, torch.randint

    def rand_like(t, **kw):
        r = real_rand_like(t, **kw); draws.append(("rand_like", r.numpy().copy())); return r

    def randint(*a, **kw):
        kw.pop("device", None)
        r = real_randint(*a, **kw); draws.append(("randint", r.numpy().copy())); return r

    og.torch.rand_like, og.torch.randint = rand_like, randint
    try:
        torch.manual_seed(3)
        steps = [0, 16, 256, 272, 288]
        for k, step in enumerate(steps):
            draws.clear()
            est._update(step=step, occ_eval_fn=occ_eval_fn, occ_thre=1e-2)
            rec[f"s{k}_step"] = np.int64(step)
            rec[f"s{k}_occs"] = est.occs.numpy().copy()
            rec[f"s{k}_binaries"] = est.binaries.numpy().copy()
            for j, (kind, arr) in enumerate(draws):
                rec[f"s{k}_draw{j}_{kind}"] = arr
            rec[f"s{k}_ndraws"] = np.int64(len(draws))
    finally:
        og.torch.rand_like, og.torch.randint = real_rand_like, real_randint

    # tests/test_grid.py:207-233 known answer
    g = OccGridEstimator(roi_aabb=torch.tensor([-1.0, -1.0, -1.0, 1.0, 1.0, 1.0]), resolution=32, levels=4)
    K = torch.tensor([[[100.0, 0, 50.0], [0, 100.0, 50.0], [0, 0, 1]]])
    pose = torch.tensor([[[-1.0, 0.0, 0.0, 0.0], [0.0, 1.0, 0.0, 0.0], [0.0, 0.0, -1.0, 2.5]]])
    g.mark_invisible_cells(K, pose, 100, 100)
    rec["mark_invisible_neg1"] = np.int64((g.occs == -1).sum().item())
    rec["mark_invisible_zero"] = np.int64((g.occs == 0).sum().item())
    rec["levels4_aabbs"] = g.aabbs.numpy()
    np.savez_compressed(os.path.join(OUT, "occgrid.npz"), resolution=np.asarray(res), roi_aabb=aabb.numpy(), **rec)


def gen_query():
    from nerfacc.grid import _enlarge_aabb, _query
    from oracle import marcher as M
    torch.manual_seed(42)
    n_rays, n_aabbs = 10, 4
    rays_o = torch.randn((n_rays, 3))
    rays_d = torch.randn((n_rays, 3))
    rays_d = rays_d / rays_d.norm(dim=-1, keepdim=True)
    base = torch.tensor([-1.0, -1.0, -1.0, 1.0, 1.0, 1.0])
    aabbs = torch.stack([_enlarge_aabb(base, 2 ** i) for i in range(n_aabbs)])
    binaries = torch.rand((n_aabbs, 32, 32, 32)) > 0.5
    iv, sm, _ = M.traverse_grids(rays_o.numpy(), rays_d.numpy(), binaries.numpy(), aabbs.numpy())
    ts = torch.from_numpy(iv.vals[iv.is_left]); te = torch.from_numpy(iv.vals[iv.is_right])
    ri = torch.from_numpy(sm.ray_indices)
    pos = rays_o[ri] + rays_d[ri] * (ts + te)[:, None] / 2.0
    occs, selector = _query(pos, binaries, base)
    np.savez_compressed(os.path.join(OUT, "query.npz"), rays_o=rays_o.numpy(), rays_d=rays_d.numpy(), aabbs=aabbs.numpy(),
                        binaries=np.packbits(binaries.numpy()), binaries_shape=np.asarray(binaries.shape),
                        n_samples=np.int64(len(ts)), ref_query_all_occupied=np.bool_(bool(occs.all())),
                        ref_query_all_selected=np.bool_(bool(selector.all())),
                        chunk_cnts=sm.packed_info[:, 1], t_starts_sum=np.float64(ts.double().sum()), t_ends_sum=np.float64(te.double().sum()))


def gen_vanilla():
    from radiance_fields.mlp import VanillaNeRFRadianceField
    from nerfacc.volrend import rendering
    import torch.nn.functional as F
    torch.manual_seed(0)
    field = VanillaNeRFRadianceField(net_depth=2, net_width=64, skip_layer=None, net_depth_condition=1, net_width_condition=64)
    W = H = 64
    focal = 0.5 * W / np.tan(np.pi / 4)
    x, y = torch.meshgrid(torch.arange(W), torch.arange(H), indexing="xy")
    cam = torch.stack([(x.flatten() - W / 2 + 0.5) / focal, -(y.flatten() - H / 2 + 0.5) / focal, -torch.ones(W * H)], -1).float()
    rays_d = cam / cam.norm(dim=-1, keepdim=True)
    rays_o = torch.zeros_like(rays_d)
    S = 32
    edges = torch.linspace(0.1, 3.3, S + 1)
    t_starts = edges[:-1].expand(W * H, S).contiguous()
    t_ends = edges[1:].expand(W * H, S).contiguous()
    target = torch.rand(W * H, 3)

    def rgb_sigma_fn(ts, te, ri):
        pos = rays_o[:, None, :] + rays_d[:, None, :] * ((ts + te) / 2.0)[..., None]
        rgb, sigma = field(pos, rays_d)
        return rgb, sigma.squeeze(-1)

    colors, opac, depths, _ = rendering(t_starts, t_ends, rgb_sigma_fn=rgb_sigma_fn, render_bkgd=torch.zeros(3))
    loss = F.smooth_l1_loss(colors, target)
    loss.backward()
    sd = {k: v.detach().numpy() for k, v in field.state_dict().items()}
    grads = {"grad." + k: p.grad.numpy() for k, p in field.named_parameters()}
    enc = field.posi_encoder(rays_d[:5] * 1.7).detach().numpy()
    np.savez_compressed(os.path.join(OUT, "vanilla.npz"), rays_o=rays_o.numpy(), rays_d=rays_d.numpy(),
                        t_edges=edges.numpy(), target=target.numpy(), colors=colors.detach().numpy(),
                        opacities=opac.detach().numpy(), depths=depths.detach().numpy(), loss=np.float32(loss.item()),
                        posenc_in=(rays_d[:5] * 1.7).numpy(), posenc_out=enc,
                        **{"sd." + k: v for k, v in sd.items()}, **grads)


def gen_raygen():
    for name in ("imageio", "cv2", "skimage", "skimage.io", "skimage.color"):
        if name not in sys.modules:
            try:
                __import__(name)
            except Exception:
                sys.modules[name] = types.ModuleType(name)
    sk = sys.modules["skimage"]
    if not hasattr(sk, "io"):
        sk.io = sys.modules["skimage.io"]; sk.color = sys.modules["skimage.color"]
    import habitat_to_data as h2d
    from scipy.spatial.transform import Rotation as R
    rec = {}
    poses = np.array([[-14.79389263, 1.5, -10.6045085, 0.0, 0.0, 0.0, 1.0],
                      [-3.0, 1.2, 2.0, 0.0, 0.38268343, 0.0, 0.92387953],
                      [1.0, 0.7, -4.0, 0.1, -0.7, 0.2, 0.678]])
    poses[2, 3:] /= np.linalg.norm(poses[2, 3:])
    for k, (W, H, scale) in enumerate([(64, 64, 1.0), (640, 640, 0.1), (800, 800, 0.05)]):
        p = poses[k]
        pose = np.eye(4); pose[:3, :3] = R.from_quat(p[3:]).as_matrix(); pose[:3, 3] = p[:3]
        pose_t = torch.from_numpy(pose).unsqueeze(0).float()
        focal = 0.5 * W / np.tan(np.pi / 4)
        K = np.array([[focal, 0.0, W / 2], [0.0, focal, H / 2], [0.0, 0.0, 1.0]])
        rs = h2d.Dataset.generate_image_rays(pose_t, W, H, K, "cpu")
        idx = np.round(np.linspace(0, len(rs.origins) - 1, int(H * scale) * int(W * scale))).astype(int)
        rec[f"c{k}_pose"] = p; rec[f"c{k}_whs"] = np.array([W, H, scale]); rec[f"c{k}_focal"] = np.float64(focal)
        rec[f"c{k}_c2w"] = pose_t.numpy(); rec[f"c{k}_idx"] = idx
        rec[f"c{k}_origins"] = rs.origins[idx].numpy(); rec[f"c{k}_viewdirs"] = rs.viewdirs[idx].numpy()
    np.savez_compressed(os.path.join(OUT, "raygen.npz"), **rec)


def _stub_modules():
    for name in ("imageio", "cv2", "skimage", "skimage.io", "skimage.color"):
        if name not in sys.modules:
            try:
                __import__(name)
            except Exception:
                sys.modules[name] = types.ModuleType(name)
    sk = sys.modules["skimage"]
    if not hasattr(sk, "io"):
        sk.io = sys.modules["skimage.io"]; sk.color = sys.modules["skimage.color"]


def gen_dataset():
    """Dataset.update_data / fetch_data (evaluation branch) / preprocess of the reference on a tiny synthetic set
    (perception/data_proc/habitat_to_data.py:89-272), CPU device."""
    _stub_modules()
    import habitat_to_data as h2d
    from scipy.spatial.transform import Rotation as R
    rng = np.random.default_rng(5)
    N, H, W = 3, 6, 8
    images = rng.integers(0, 256, size=(N, H, W, 3)).astype(np.uint8)
    depths = rng.random((N, H, W)).astype(np.float32) * 5
    sems = rng.integers(0, 29, size=(N, H, W))
    c2w = np.stack([np.eye(4) for _ in range(N)])
    for i in range(N):
        c2w[i, :3, :3] = R.from_euler("xyz", rng.random(3) * 2).as_matrix()
        c2w[i, :3, 3] = rng.random(3) * 4
    ds = h2d.Dataset(training=False, save_fp="/tmp/_golden_ds", num_rays=None, num_models=2, device="cpu")
    np.random.seed(0)
    ds.update_data(images[:2], depths[:2], sems[:2], c2w[:2])
    ds.update_data(images[2:], depths[2:], sems[2:], c2w[2:])
    d = ds[2]
    np.savez_compressed(os.path.join(OUT, "dataset.npz"), images=images, depths=depths, sems=sems, c2w=c2w,
                        K=ds.K.numpy(), size=np.int64(ds.size), pixels=d["pixels"].numpy(), dep=d["dep"].numpy(),
                        sem=d["sem"].numpy(), origins=d["rays"].origins.numpy(), viewdirs=d["rays"].viewdirs.numpy(),
                        color_bkgd=d["color_bkgd"].numpy(), boot0=np.asarray(ds.bootstrap(0)), boot1=np.asarray(ds.bootstrap(1)))


if __name__ == "__main__":
    which = sys.argv[1:] or ["aabb", "volrend", "occgrid", "query", "vanilla", "raygen", "dataset"]
    _enter_reference()
    for w in which:
        globals()["gen_" + w]()
        print("wrote", w)
