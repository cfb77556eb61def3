"""`Dataset` with the reference's interface (perception/data_proc/habitat_to_data.py:30-302): device-resident
images / depths / semantics / poses, random-pixel training batches, whole-image evaluation batches, and the static
render drivers.  Rays come from the HIP ray generator (`mnf_generate_rays`), everything else is indexing.

Also the two §8(f) hand-offs that sit next to it: checkpoints in the reference's `.pth` layout
(scripts/pipeline.py:630-635, visualization/vis_nerf_habitat.py:124-126) and the planner's 2-D path-finding map
(pipeline.py:1043-1049, planning/planning_funcs.py:243-266)."""
import os

import numpy as np
import torch

from . import _lib as L
from . import render as RD
from .render import Rays


class Dataset(torch.utils.data.Dataset):
    """Gathered dataset (habitat_to_data.py:30-302)."""

    def __init__(self, training: bool, save_fp: str, num_rays: int = None, batch_over_images: bool = True,
                 num_models: int = 1, device: str = "cpu", packed: bool = False, use_bootstrap_index: bool = False):
        """`packed=True` keeps depths as fp16 and class ids as uint8 on the device (3 + 2 + 1 = 6 B per pixel instead of
        3 + 4 + 8 = 15 B; Habitat class ids are < 256 and depths are metres with ~1e-3 relative fp16 error) — the batches it
        hands out keep the reference dtypes (f32 / f32 / i64).  `use_bootstrap_index=True` makes a TRAINING fetch use the
        image `index` it is given (what scripts/pipeline.py:423-434 computes from the member's bootstrap set) instead of
        drawing an image from the whole set, which is what the reference does and which makes its bootstrap sets
        ineffective (habitat_to_data.py:209-215); default False = reference behaviour."""
        super().__init__()
        self.packed, self.use_bootstrap_index = packed, use_bootstrap_index
        self.num_rays = num_rays
        self.batch_over_images = batch_over_images
        self.num_models = num_models
        self.bootstrap_indices = [np.array([]).astype(int) for _ in range(self.num_models - 1)]
        self.images = self.depths = self.semantics = self.camtoworlds = None
        self.training = training
        self.device = device
        self.save_fp = save_fp
        self.boot_scale = 0.7
        self.saved_batch = 0
        self.downsampled_end = None
        self.size = 0
        self.save_batch_size = 5000
        if save_fp and not os.path.exists(save_fp):
            os.makedirs(save_fp)

    def update_data(self, images, depths, semantics, camtoworlds):
        """habitat_to_data.py:89-153 (uint8 images [N,H,W,C], f32 depths, int64 semantics, f32 poses, on device)."""
        img_np = np.asarray(images)
        if img_np.ndim != 4 or img_np.shape[-1] != 3:
            # the pixel gather reads 3 bytes per pixel (mnf_gather_pixels); an RGBA capture (Habitat's colour sensor before pipeline.py's
            # [..., :3] slice) would be read with the wrong stride and give garbage colours without any error
            raise ValueError(f"Dataset.update_data expects uint8 images [N,H,W,3] (got shape {tuple(img_np.shape)}): slice RGBA captures with [..., :3]")
        new_images = torch.from_numpy(img_np).to(torch.uint8).to(self.device)
        new_depths = torch.from_numpy(np.asarray(depths)).to(torch.float16 if self.packed else torch.float32).to(self.device)
        sem_np = np.asarray(semantics)
        if self.packed and (sem_np.min() < 0 or sem_np.max() > 255):
            raise ValueError("packed dataset layout stores class ids as uint8: ids must be in [0, 255]")
        new_sems = torch.from_numpy(sem_np.astype(np.uint8 if self.packed else np.int64)).to(self.device)
        new_c2w = torch.from_numpy(np.asarray(camtoworlds)).to(torch.float32).to(self.device)
        for i, arr in enumerate(self.bootstrap_indices):
            ids = np.random.choice(len(images), size=(int(len(images) * self.boot_scale),), replace=True)
            self.bootstrap_indices[i] = np.concatenate([arr, self.size + ids], axis=0)
        if self.images is None:
            self.images, self.depths, self.semantics, self.camtoworlds = new_images, new_depths, new_sems, new_c2w
            self.height, self.width = self.images.shape[1:3]
            focal = 0.5 * self.width / np.tan(np.pi / 2 / 2)
            self.K = torch.tensor([[focal, 0.0, self.width / 2], [0.0, focal, self.height / 2], [0.0, 0.0, 1.0]],
                                  dtype=torch.float64).to(torch.float32).to(self.device)
            self._focal = float(np.float32(focal))
        else:
            self.images = torch.cat([self.images, new_images], dim=0)
            self.depths = torch.cat([self.depths, new_depths], dim=0)
            self.semantics = torch.cat([self.semantics, new_sems], dim=0)
            self.camtoworlds = torch.cat([self.camtoworlds, new_c2w], dim=0)
        self.size = self.size + len(images)

    def __len__(self):
        return self.size

    @torch.no_grad()
    def __getitem__(self, index):
        return self.preprocess(self.fetch_data(index))

    def save(self):
        np.savez(self.save_fp + "/data" + str(self.saved_batch) + ".npz", images=self.images.cpu().numpy(),
                 depths=self.depths.cpu().numpy(), semantics=self.semantics.cpu().numpy(),
                 camtoworlds=self.camtoworlds.cpu().numpy(), K=self.K.cpu().numpy(),
                 bootstrap_indices=np.array(self.bootstrap_indices, dtype=object))

    def update_num_rays(self, num_rays):
        self.num_rays = num_rays

    def bootstrap(self, model_idx):
        return np.arange(self.size) if model_idx == 0 else self.bootstrap_indices[model_idx - 1]

    def preprocess(self, data):
        """habitat_to_data.py:184-203: random background while training, white otherwise."""
        color_bkgd = torch.rand(3, device=self.device) if self.training else torch.ones(3, device=self.device)
        out = {"pixels": data["rgb"], "dep": data["dep"], "sem": data["sem"], "rays": data["rays"], "color_bkgd": color_bkgd}
        out.update({k: v for k, v in data.items() if k not in ("rgb", "rays", "dep", "sem")})
        return out

    def fetch_data(self, index):
        """habitat_to_data.py:205-272.  Training: ONE random image (the `index` argument is ignored, as in the
        reference) and `num_rays` random pixels; evaluation: image `index`, all pixels in 'xy' order."""
        W, H = self.width, self.height
        if self.training:
            num_rays = self.num_rays
            image_id = torch.randint(0, self.size, size=(1,), device=self.device)
            if self.use_bootstrap_index:
                image_id = torch.tensor([int(index)], device=self.device)
            x = torch.randint(0, W, size=(num_rays,), device=self.device)
            y = torch.randint(0, H, size=(num_rays,), device=self.device)
            # same random pixels, grouped by 32x32 image block: the batch is a set (every loss is a mean over rays), and
            # neighbouring rays share hash-table lines in the sampler's pre-pass, the forward and the gradient scatter
            order = torch.argsort((y // 32) * ((W + 31) // 32) + x // 32)
            x, y = x[order], y[order]
        else:
            image_id = torch.tensor([index], device=self.device)
            x, y = torch.meshgrid(torch.arange(W, device=self.device), torch.arange(H, device=self.device), indexing="xy")
            x, y = x.flatten(), y.flatten()
        c2w = self.camtoworlds[image_id][:, :3, :4].contiguous()
        pix = (y * W + x).to(torch.int64).contiguous()
        n = pix.shape[0]
        # pixel gather (habitat_to_data.py:229-232) in one kernel, from either storage layout
        L.require_gpu(self.images, self.depths, self.semantics)
        rgb = torch.empty(n, 3, device=self.device)
        dep = torch.empty(n, device=self.device)
        sem = torch.empty(n, dtype=torch.int64, device=self.device)
        L.launch(L.load_library().mnf_gather_pixels, L.ptr(self.images), L.ptr(self.depths), int(self.depths.dtype == torch.float16),
                 L.ptr(self.semantics), int(self.semantics.dtype == torch.uint8), H * W, L.ptr(image_id.to(torch.int64)), L.ptr(pix), n,
                 L.ptr(rgb), L.ptr(dep), L.ptr(sem))
        origins = torch.empty(1, n, 3, device=self.device)
        viewdirs = torch.empty(1, n, 3, device=self.device)
        L.require_gpu(c2w, pix)
        L.launch(L.load_library().mnf_generate_rays, L.ptr(c2w), 1, W, H, self._focal, L.ptr(pix), n, L.ptr(origins), L.ptr(viewdirs))
        shape = (n,) if self.training else (H, W)
        rays = Rays(origins=origins.reshape(*shape, 3), viewdirs=viewdirs.reshape(*shape, 3))
        return {"rgb": rgb.reshape(*shape, rgb.shape[-1]), "dep": dep.reshape(shape), "sem": sem.reshape(shape), "rays": rays,
                "image_id": image_id, "x": x, "y": y}

    generate_image_rays = staticmethod(RD.generate_image_rays)
    render_image_from_pose = staticmethod(RD.render_image_from_pose)
    render_probablistic_image_from_pose = staticmethod(RD.render_probablistic_image_from_pose)


# ------------------------------------------------------------------ checkpoints (pipeline.py:630-635)
def save_checkpoint(path, estimator, radiance_field, optimizer=None):
    d = {"occ_grid": estimator.binaries, "model": radiance_field.state_dict()}
    if optimizer is not None:
        d["optimizer_state_dict"] = optimizer.state_dict()
    torch.save(d, path)


def load_checkpoint(path, estimator, radiance_field, optimizer=None, map_location=None):
    """visualization/vis_nerf_habitat.py:124-126 (+ optimizer state when present)."""
    ck = torch.load(path, map_location=map_location)
    estimator.binaries = ck["occ_grid"].to(estimator.binaries.device)
    radiance_field.load_state_dict(ck["model"])
    if optimizer is not None and "optimizer_state_dict" in ck:
        optimizer.load_state_dict(ck["optimizer_state_dict"])
    return ck


# ------------------------------------------------------------------ planner hand-off
@torch.no_grad()
def planner_path_finding_map(estimators, current_state_xzy=None, aabb_xzy=None, voxel_grid_size=0.2, y_slice=8):
    """The 2-D map `sample_traj` builds from the ensemble's occupancy grids (pipeline.py:1043-1049,
    planning_funcs.py:243-266), computed on the device and returned as a host int32 array [X, Z].
    `current_state_xzy` / `aabb_xzy` are the axis-swapped state and box pipeline.py:1053-1061 passes; when given, the
    five cells around the vehicle are cleared exactly as planning_funcs.py:262-266 does."""
    b = torch.stack([e.binaries[0] for e in estimators]).contiguous()
    L.require_gpu(b)
    bu = b.view(torch.uint8) if b.dtype == torch.bool else b.to(torch.uint8)
    M, X, Y, Z = bu.shape
    out = torch.empty(X, Z, dtype=torch.int32, device=b.device)
    L.launch(L.load_library().mnf_planner_map, L.ptr(bu), M, X, Y, Z, y_slice, L.ptr(out))
    m = out.cpu().numpy()
    if current_state_xzy is not None:
        v = np.array((np.asarray(current_state_xzy)[:3] - np.asarray(aabb_xzy)[:3]) // voxel_grid_size, dtype=int)
        for dy, dx in ((0, 0), (1, 0), (-1, 0), (0, 1), (0, -1)):
            m[v[1] + dy, v[0] + dx] = 0
    return m
