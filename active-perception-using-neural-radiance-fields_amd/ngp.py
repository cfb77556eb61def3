"""`NGPRadianceField` with the reference's constructor and call surface
(perception/models/radiance_fields/ngp.py:69-238), evaluated by the fused HIP kernel of
csrc/field.hip instead of tiny-cuda-nn.

Parameters live in three flat fp32 tensors named like the reference's state_dict
(`mlp_base.params` = base-MLP weights followed by the hash table, `mlp_head.params`,
`mlp_sem.params`), so optimizers and checkpoints see the same objects.  The layout inside each
vector restates tiny-cuda-nn's (un-pinned, see oracle/field.py); the C library converts them to
fp16 hash entries and MFMA-fragment-ordered fp16 weights whenever they change.
"""
import ctypes
import math
from typing import List, Union

import numpy as np
import torch

from . import _lib as L


class _Params(torch.nn.Module):
    """Stands in for a tcnn module: one flat `params` Parameter."""

    def __init__(self, n: int):
        super().__init__()
        self.params = torch.nn.Parameter(torch.zeros(n, dtype=torch.float32))


def _xavier_uniform_(flat: torch.Tensor, shapes, gen: torch.Generator):
    k = 0
    for o, i in shapes:
        lim = math.sqrt(6.0 / (o + i))
        flat[k:k + o * i] = (torch.rand(o * i, generator=gen) * 2 - 1) * lim
        k += o * i
    return k


class _FieldFunction(torch.autograd.Function):
    """Differentiable field evaluation: what `loss.backward()` reaches inside tiny-cuda-nn in the reference
    (scripts/pipeline.py:518).  Forward keeps the activations in a workspace; backward returns dL/d(params) for the
    three flat parameter vectors (positions and directions get no gradient, as in the reference's use)."""

    @staticmethod
    def forward(ctx, module, pos, dirs, p_base, p_head, p_sem):
        lib = L.load_library()
        h = module._ensure_handle()
        n = pos.shape[0]
        dev = pos.device
        rgb = torch.empty(n, 3, device=dev); sigma = torch.empty(n, 1, device=dev)
        sem = torch.empty(n, module.num_semantic_classes, device=dev)
        nbytes = lib.mnf_field_train_workspace_bytes(h, n)
        ws = torch.empty(max(int(nbytes), 1), dtype=torch.uint8, device=dev)
        if n:
            L.launch(lib.mnf_field_forward_train, h, L.ptr(pos), L.ptr(dirs), n, L.ptr(rgb), L.ptr(sigma), L.ptr(sem), L.ptr(ws),
                                                nbytes)
        ctx.module, ctx.ws, ctx.nbytes, ctx.n = module, ws, nbytes, n
        ctx.save_for_backward(pos, rgb, sigma)
        return rgb, sigma, sem

    @staticmethod
    def backward(ctx, g_rgb, g_sigma, g_sem):
        lib = L.load_library()
        module = ctx.module
        pos, rgb, sigma = ctx.saved_tensors
        dev = pos.device
        h = module._handle
        g_base = torch.empty_like(module.mlp_base.params)
        g_head = torch.empty_like(module.mlp_head.params)
        g_s = torch.empty_like(module.mlp_sem.params)
        zeros = lambda g, shape: torch.zeros(shape, device=dev) if g is None else L.contig(g, torch.float32)
        g_rgb, g_sigma = zeros(g_rgb, rgb.shape), zeros(g_sigma, sigma.shape)
        g_sem = zeros(g_sem, (ctx.n, module.num_semantic_classes))
        L.launch(lib.mnf_field_backward, h, L.ptr(pos), ctx.n, L.ptr(g_rgb), L.ptr(g_sigma), L.ptr(g_sem), L.ptr(rgb), L.ptr(sigma),
                                       L.ptr(ctx.ws), ctx.nbytes, float(module.loss_scale), L.ptr(g_base), L.ptr(g_head), L.ptr(g_s))
        ctx.ws = None
        return None, None, None, g_base, g_head, g_s


class _FieldSamplesFunction(torch.autograd.Function):
    """`_FieldFunction` for packed samples given as (ray, t_start, t_end): the closure of utils.py:122-137 (ray gathers, `origins + dirs * (t_starts + t_ends) / 2`)
    happens inside the forward kernel (`mnf_field_forward_train_samples`), which also leaves the positions the backward's hash-table scatter reads."""

    @staticmethod
    def forward(ctx, module, rays_o, rays_d, ray_indices, t_starts, t_ends, p_base, p_head, p_sem):
        lib = L.load_library()
        h = module._ensure_handle()
        n = t_starts.shape[0]
        dev = t_starts.device
        rgb = torch.empty(n, 3, device=dev); sigma = torch.empty(n, 1, device=dev)
        sem = torch.empty(n, module.num_semantic_classes, device=dev)
        pos = torch.empty(n, 3, device=dev)
        nbytes = lib.mnf_field_train_workspace_bytes(h, n)
        ws = torch.empty(max(int(nbytes), 1), dtype=torch.uint8, device=dev)
        if n:
            L.launch(lib.mnf_field_forward_train_samples, h, L.ptr(rays_o), L.ptr(rays_d), L.ptr(ray_indices), L.ptr(t_starts), L.ptr(t_ends), n,
                     L.ptr(rgb), L.ptr(sigma), L.ptr(sem), L.ptr(pos), L.ptr(ws), nbytes)
        ctx.module, ctx.ws, ctx.nbytes, ctx.n = module, ws, nbytes, n
        ctx.save_for_backward(pos, rgb, sigma)
        return rgb, sigma, sem

    @staticmethod
    def backward(ctx, g_rgb, g_sigma, g_sem):
        return (None, None, None, None, None, None) + _FieldFunction.backward(ctx, g_rgb, g_sigma, g_sem)[3:]


class RaySigmaFn:
    """The `sigma_fn` closure of perception/models/utils.py:89-101 (`radiance_field.query_density(origins + dirs * t_mid)`) as
    an object: called like the closure it evaluates every sample; `OccGridEstimator.sampling` recognises it and asks for the
    ray-major form instead (`mnf_field_density_rays`), which leaves out the samples behind opaque surfaces — they cannot
    pass the visibility test, so the returned sample set is the same."""

    def __init__(self, radiance_field, rays_o, rays_d):
        self.field, self.rays_o, self.rays_d = radiance_field, rays_o, rays_d

    def __call__(self, t_starts, t_ends, ray_indices):
        return self.field.forward_samples(self.rays_o, self.rays_d, ray_indices, t_starts, t_ends, density_only=True)[0]

    @torch.no_grad()
    def ray_major(self, t_starts, t_ends, ray_indices, packed_info, early_stop_eps: float):
        f = self.field
        h = f._ensure_handle()
        L.require_gpu(self.rays_o, self.rays_d, ray_indices, t_starts, t_ends, packed_info)
        o, d = L.contig(self.rays_o, torch.float32), L.contig(self.rays_d, torch.float32)
        ri, ts, te = L.contig(ray_indices, torch.int64), L.contig(t_starts, torch.float32), L.contig(t_ends, torch.float32)
        starts, cnts = (L.contig(x, torch.int64) for x in packed_info.unbind(-1))
        sigma = torch.empty(ts.shape[0], device=o.device, dtype=torch.float32)
        L.launch(L.load_library().mnf_field_density_rays, h, L.ptr(o), L.ptr(d), L.ptr(ri), L.ptr(ts), L.ptr(te), L.ptr(starts), L.ptr(cnts),
                 starts.shape[0], ts.shape[0], float(early_stop_eps), L.ptr(sigma))
        return sigma


class NGPRadianceField(torch.nn.Module):
    """Instant-NGP radiance field with a semantic head (ngp.py:69-169)."""

    def __init__(self, aabb: Union[torch.Tensor, List[float]], num_dim: int = 3, use_viewdirs: bool = True,
                 neurons: int = 128, layers: int = 4, density_activation=None, unbounded: bool = False,
                 base_resolution: int = 16, max_resolution: int = 4096, geo_feat_dim: int = 15, n_levels: int = 16,
                 log2_hashmap_size: int = 19, num_semantic_classes: int = 0, seed: int = 0,
                 tcnn_output_rounding: bool = False, mfma_bf16: bool = False, tcnn_blend_fp16: bool = False) -> None:
        super().__init__()
        if not isinstance(aabb, torch.Tensor):
            aabb = torch.tensor(aabb, dtype=torch.float32)
        if num_dim != 3 or not use_viewdirs or unbounded or geo_feat_dim != 15 or density_activation is not None:
            raise NotImplementedError("mi355nerf supports the configuration scripts/pipeline.py uses: num_dim=3, "
                                      "use_viewdirs=True, unbounded=False, geo_feat_dim=15, trunc_exp(x-1) density")
        if num_semantic_classes <= 0:
            raise NotImplementedError("mi355nerf requires num_semantic_classes > 0 (pipeline.py always passes --sem-num)")
        self.register_buffer("aabb", aabb.detach().clone().float())
        self.num_dim, self.use_viewdirs, self.unbounded = num_dim, use_viewdirs, unbounded
        self.base_resolution, self.max_resolution = base_resolution, max_resolution
        self.geo_feat_dim, self.n_levels, self.log2_hashmap_size = geo_feat_dim, n_levels, log2_hashmap_size
        self.num_semantic_classes = num_semantic_classes
        self.neurons, self.layers = neurons, layers
        self.loss_scale = 128.0   # fp16 activation-gradient scale of the backward kernels (tcnn's default)

        lib = L.load_library()
        cfg = L.FieldConfig()
        for i, v in enumerate(aabb.detach().cpu().float().tolist()):
            cfg.aabb[i] = v
        cfg.neurons, cfg.layers, cfg.num_semantic_classes = neurons, layers, num_semantic_classes
        cfg.n_levels, cfg.n_features, cfg.log2_hashmap_size = n_levels, 4, log2_hashmap_size
        cfg.base_resolution, cfg.max_resolution = base_resolution, max_resolution
        # tiny-cuda-nn returns fp16 network outputs which ngp.py:181-220 widen with `.to(x)`; True reproduces that rounding
        # (default False: fp32 outputs, the more precise of the two; DESIGN.md §2 quantifies the difference)
        cfg.output_fp16 = 1 if tcnn_output_rounding else 0
        # matrix-core operand type: fp16 (the reference's tcnn arithmetic) or bf16 (BASELINE config 5); the hash table stays fp16
        cfg.mfma_bf16 = 1 if mfma_bf16 else 0
        # precision of the 8-corner hash blend: fp32 with one rounding (default) or tcnn's fp16 fused multiply-adds (include/mi355nerf.h: blend_fp16)
        cfg.blend_fp16 = 1 if tcnn_blend_fp16 else 0
        self._cfg = cfg
        self._handle = ctypes.c_void_p()
        self._handle_device = None
        self._loaded_versions = None

        # parameter counts do not need a device; compute them like the C side does
        W, Wh = neurons, neurons // 2
        sem_pad = ((num_semantic_classes + 15) // 16) * 16
        self._shapes = {
            "base": [(W, 64)] + [(W, W)] * (layers - 1) + [(16, W)],
            "head": [(Wh, 32), (Wh, Wh), (16, Wh)],
            "sem": [(Wh, 16), (Wh, Wh), (sem_pad, Wh)],
        }
        table = self._table_entries()
        n_base_mlp = sum(o * i for o, i in self._shapes["base"])
        self.direction_encoding = _Params(0)
        self.mlp_base = _Params(n_base_mlp + table * 4)
        self.mlp_head = _Params(sum(o * i for o, i in self._shapes["head"]))
        self.mlp_sem = _Params(sum(o * i for o, i in self._shapes["sem"]))
        # tcnn initialisation: xavier-uniform MLPs, U(-1e-4, 1e-4) grid
        gen = torch.Generator().manual_seed(seed)
        with torch.no_grad():
            k = _xavier_uniform_(self.mlp_base.params, self._shapes["base"], gen)
            self.mlp_base.params[k:] = (torch.rand(table * 4, generator=gen) * 2 - 1) * 1e-4
            _xavier_uniform_(self.mlp_head.params, self._shapes["head"], gen)
            _xavier_uniform_(self.mlp_sem.params, self._shapes["sem"], gen)

    def _table_entries(self) -> int:
        pls = math.exp((math.log(self.max_resolution) - math.log(self.base_resolution)) / (self.n_levels - 1))
        total = 0
        for l in range(self.n_levels):
            scale = np.float32(2.0 ** (l * math.log2(pls)) * self.base_resolution - 1.0)
            res = int(math.ceil(float(scale))) + 1
            total += min(((res ** 3 + 7) // 8) * 8, 1 << self.log2_hashmap_size)
        return total

    # ---- handle management ------------------------------------------------------------
    def _ensure_handle(self):
        dev = self.mlp_base.params.device
        if dev.type != "cuda":
            raise L.MnfError("NGPRadianceField must be on a GPU (`.to('cuda')`): libmi355nerf has no CPU fallback")
        lib = L.load_library()
        if self._handle_device != dev:
            if self._handle:
                lib.mnf_field_destroy(self._handle)
                self._handle = ctypes.c_void_p()
            with torch.cuda.device(dev):
                L.check(lib.mnf_field_create(ctypes.byref(self._cfg), ctypes.byref(self._handle)))
            assert lib.mnf_field_param_count(self._handle, 0) == self.mlp_base.params.numel()
            assert lib.mnf_field_param_count(self._handle, 1) == self.mlp_head.params.numel()
            assert lib.mnf_field_param_count(self._handle, 2) == self.mlp_sem.params.numel()
            self._handle_device = dev
            self._loaded_versions = None
        versions = (self.mlp_base.params._version, self.mlp_head.params._version, self.mlp_sem.params._version,
                    self.mlp_base.params.data_ptr())
        if versions != self._loaded_versions:
            L.launch(lib.mnf_field_set_params, self._handle, L.ptr(self.mlp_base.params), L.ptr(self.mlp_head.params),
                                             L.ptr(self.mlp_sem.params))
            self._loaded_versions = versions
        return self._handle

    def _refresh_after_optimizer(self):
        """Called by `optim.FusedAdam` (bound with `bind_field`) after it updated the parameters: its kernel has already written the
        rounded hash-table values into the handle's fp16 table, so only the MLP weight fragments are re-derived (no conversion pass
        over the 25 M table entries), and the handle counts as current for the new parameter versions."""
        lib = L.load_library()
        L.launch(lib.mnf_field_refresh_weights, self._handle, L.ptr(self.mlp_base.params), L.ptr(self.mlp_head.params), L.ptr(self.mlp_sem.params))
        self._mark_current()

    def _mark_current(self):
        """The handle's fp16 table and weight fragments match the parameters as they are now."""
        self._loaded_versions = (self.mlp_base.params._version, self.mlp_head.params._version, self.mlp_sem.params._version,
                                 self.mlp_base.params.data_ptr())

    def __del__(self):
        try:
            if self._handle:
                L.load_library().mnf_field_destroy(self._handle)
        except Exception:
            pass

    def grid_meta(self):
        """(scale f32, res, size, offset, hashed) per level, as computed by the C library."""
        h = self._ensure_handle()
        n = self.n_levels
        scale = (ctypes.c_float * n)(); res = (ctypes.c_int32 * n)(); size = (ctypes.c_int32 * n)()
        off = (ctypes.c_int64 * n)(); hashed = (ctypes.c_int32 * n)()
        L.check(L.load_library().mnf_field_grid_meta_host(h, scale, res, size, off, hashed))
        return list(scale), list(res), list(size), list(off), list(hashed)

    # ---- reference call surface -------------------------------------------------------
    @torch.no_grad()
    def query_density(self, x, return_feat: bool = False):
        """ngp.py:171-200."""
        if return_feat:
            raise NotImplementedError("return_feat=True is internal to the reference's forward(); use forward()")
        h = self._ensure_handle()
        L.require_gpu(x)
        shp = x.shape[:-1]
        pos = L.contig(x.reshape(-1, 3), torch.float32)
        out = torch.empty(pos.shape[0], 1, device=pos.device, dtype=torch.float32)
        L.launch(L.load_library().mnf_field_density, h, L.ptr(pos), pos.shape[0], L.ptr(out))
        return out.view(*shp, 1)

    def forward(self, positions: torch.Tensor, directions: torch.Tensor = None):
        """ngp.py:222-238 -> (rgb [N,3], density [N,1], sem_logits [N,C])."""
        if directions is None:
            raise NotImplementedError("forward() without directions is not reachable from pipeline.py")
        assert positions.shape == directions.shape, f"{positions.shape} v.s. {directions.shape}"
        h = self._ensure_handle()
        L.require_gpu(positions, directions)
        shp = positions.shape[:-1]
        pos = L.contig(positions.detach().reshape(-1, 3), torch.float32)
        dirs = L.contig(directions.detach().reshape(-1, 3), torch.float32)
        n = pos.shape[0]
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            rgb, sigma, sem = _FieldFunction.apply(self, pos, dirs, self.mlp_base.params, self.mlp_head.params, self.mlp_sem.params)
            return rgb.view(*shp, 3), sigma.view(*shp, 1), sem.view(*shp, self.num_semantic_classes)
        rgb = torch.empty(n, 3, device=pos.device, dtype=torch.float32)
        sigma = torch.empty(n, 1, device=pos.device, dtype=torch.float32)
        sem = torch.empty(n, self.num_semantic_classes, device=pos.device, dtype=torch.float32)
        L.launch(L.load_library().mnf_field_forward, h, L.ptr(pos), L.ptr(dirs), n, L.ptr(rgb), L.ptr(sigma), L.ptr(sem))
        return rgb.view(*shp, 3), sigma.view(*shp, 1), sem.view(*shp, self.num_semantic_classes)

    def forward_samples_grad(self, rays_o, rays_d, ray_indices, t_starts, t_ends):
        """`forward(origins[ray_indices] + viewdirs[ray_indices] * (t_starts + t_ends)[:, None] / 2, viewdirs[ray_indices])` (utils.py:122-137) with the
        gathers and the positions formed inside the kernel; differentiable w.r.t. the three parameter vectors -> (rgb [N,3], density [N,1], sem [N,C])."""
        L.require_gpu(rays_o, rays_d, ray_indices, t_starts, t_ends)
        o, d = L.contig(rays_o.detach(), torch.float32), L.contig(rays_d.detach(), torch.float32)
        ri, ts, te = L.contig(ray_indices, torch.int64), L.contig(t_starts.detach(), torch.float32), L.contig(t_ends.detach(), torch.float32)
        return _FieldSamplesFunction.apply(self, o, d, ri, ts, te, self.mlp_base.params, self.mlp_head.params, self.mlp_sem.params)

    @torch.no_grad()
    def forward_samples(self, rays_o, rays_d, ray_indices, t_starts, t_ends, density_only: bool = False):
        """The closures `sigma_fn` / `rgb_sigma_sem_fn` of perception/models/utils.py:89-137 fused with the field:
        positions are formed in-kernel from (ray, t_start, t_end)."""
        h = self._ensure_handle()
        L.require_gpu(rays_o, rays_d, ray_indices, t_starts, t_ends)
        o, d = L.contig(rays_o, torch.float32), L.contig(rays_d, torch.float32)
        ri, ts, te = L.contig(ray_indices, torch.int64), L.contig(t_starts, torch.float32), L.contig(t_ends, torch.float32)
        n = ts.shape[0]
        sigma = torch.empty(n, device=o.device, dtype=torch.float32)
        rgb = sem = None
        if not density_only:
            rgb = torch.empty(n, 3, device=o.device, dtype=torch.float32)
            sem = torch.empty(n, self.num_semantic_classes, device=o.device, dtype=torch.float32)
        L.launch(L.load_library().mnf_field_forward_samples, h, L.ptr(o), L.ptr(d), L.ptr(ri), L.ptr(ts), L.ptr(te), n,
                                                           L.ptr(rgb), L.ptr(sigma), L.ptr(sem))
        return (sigma,) if density_only else (rgb, sigma, sem)
