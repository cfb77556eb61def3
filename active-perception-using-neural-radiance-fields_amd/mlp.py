"""`VanillaNeRFRadianceField` with the reference's constructor, parameter names and call surface
(perception/models/radiance_fields/mlp.py:206-245), evaluated by csrc/vanilla.hip: frequency positional encoding
(mlp.py:168-203) and the biased Linear + ReLU stack (mlp.py:14-165) in exact fp32 on the matrix cores, forward and
backward.  BASELINE config 1 (64x64 view, 32 samples per ray, 2-layer-64 MLP) runs through this module.

The module tree only HOLDS the parameters (same `state_dict()` keys as the reference, so its checkpoints load); the
arithmetic is in the C library, which sees them as one flat fp32 vector in `named_parameters()` order.
"""
import ctypes
import math
from typing import Optional

import torch
import torch.nn as nn

from . import _lib as L


class _Encoder(nn.Module):
    """`SinusoidalEncoder` as a parameter-free holder of the reference's `scales` buffer (mlp.py:171-182)."""

    def __init__(self, x_dim: int, min_deg: int, max_deg: int, use_identity: bool = True):
        super().__init__()
        self.x_dim, self.min_deg, self.max_deg, self.use_identity = x_dim, min_deg, max_deg, use_identity
        self.register_buffer("scales", torch.tensor([2 ** i for i in range(min_deg, max_deg)]))

    @property
    def latent_dim(self) -> int:
        return (int(self.use_identity) + (self.max_deg - self.min_deg) * 2) * self.x_dim


class _Layers(nn.Module):
    """`MLP` as a holder of its Linear layers: `hidden_layers.<i>` and optionally `output_layer` (mlp.py:43-63)."""

    def __init__(self, input_dim: int, output_dim: Optional[int], net_depth: int, net_width: int, skip_layer: Optional[int]):
        super().__init__()
        self.hidden_layers = nn.ModuleList()
        width_in = input_dim
        for i in range(net_depth):
            self.hidden_layers.append(nn.Linear(width_in, net_width))
            skip_here = skip_layer is not None and i % skip_layer == 0 and i > 0       # mlp.py:52-57
            width_in = net_width + input_dim if skip_here else net_width
        self.output_dim = width_in
        if output_dim is not None:
            self.output_layer = nn.Linear(width_in, output_dim)
            self.output_dim = output_dim
        for m in self.modules():                                                       # mlp.py:67-84: xavier-uniform weights, zero biases
            if isinstance(m, nn.Linear):
                nn.init.xavier_uniform_(m.weight)
                nn.init.zeros_(m.bias)


class _NerfMLP(nn.Module):
    """Parameter tree of `NerfMLP` (mlp.py:113-151): base, sigma_layer, bottleneck_layer, rgb_layer."""

    def __init__(self, input_dim, condition_dim, net_depth, net_width, skip_layer, net_depth_condition, net_width_condition):
        super().__init__()
        self.base = _Layers(input_dim, None, net_depth, net_width, skip_layer)
        hidden = self.base.output_dim
        self.sigma_layer = _Layers(hidden, 1, 0, 0, None)
        self.bottleneck_layer = _Layers(hidden, net_width, 0, 0, None)
        self.rgb_layer = _Layers(net_width + condition_dim, 3, net_depth_condition, net_width_condition, None)


class _VanillaFunction(torch.autograd.Function):
    """forward keeps the layer inputs in a workspace; backward returns one gradient per parameter tensor."""

    @staticmethod
    def forward(ctx, module, pos, dirs, spd, *params):
        lib = L.load_library()
        h = module._ensure_handle()
        n, dev = pos.shape[0], pos.device
        rgb, sigma = torch.empty(n, 3, device=dev), torch.empty(n, device=dev)
        nbytes = int(lib.mnf_vanilla_train_workspace_bytes(h, n))
        ws = torch.empty(max(nbytes, 8), dtype=torch.uint8, device=dev)
        if n:
            L.launch(lib.mnf_vanilla_forward, h, L.ptr(pos), L.ptr(dirs), n, spd, L.ptr(rgb), L.ptr(sigma), L.ptr(ws), nbytes)
        ctx.module, ctx.ws, ctx.nbytes, ctx.n = module, ws, nbytes, n
        ctx.save_for_backward(rgb, sigma)
        return rgb, sigma

    @staticmethod
    def backward(ctx, g_rgb, g_sigma):
        lib = L.load_library()
        module = ctx.module
        rgb, sigma = ctx.saved_tensors
        g_rgb = torch.zeros_like(rgb) if g_rgb is None else L.contig(g_rgb, torch.float32)
        g_sigma = torch.zeros_like(sigma) if g_sigma is None else L.contig(g_sigma, torch.float32)
        flat = torch.empty(module._n_params, device=rgb.device)
        L.launch(lib.mnf_vanilla_backward, module._handle, L.ptr(g_rgb), L.ptr(g_sigma), L.ptr(rgb), L.ptr(sigma), ctx.n, L.ptr(ctx.ws),
                 ctx.nbytes, L.ptr(flat))
        ctx.ws = None
        grads = [flat[o:o + r * c].view(p.shape) for (o, r, c), p in zip(module._layout, module.parameters())]
        return (None, None, None, None, *grads)


class VanillaNeRFRadianceField(nn.Module):
    def __init__(self, net_depth: int = 8, net_width: int = 256, skip_layer: int = 4, net_depth_condition: int = 1,
                 net_width_condition: int = 128) -> None:
        super().__init__()
        self.posi_encoder = _Encoder(3, 0, 10, True)
        self.view_encoder = _Encoder(3, 0, 4, True)
        self.mlp = _NerfMLP(self.posi_encoder.latent_dim, self.view_encoder.latent_dim, net_depth, net_width, skip_layer,
                            net_depth_condition, net_width_condition)
        cfg = L.VanillaConfig()
        cfg.net_depth, cfg.net_width, cfg.skip_layer = net_depth, net_width, skip_layer if skip_layer else 0
        cfg.net_depth_condition, cfg.net_width_condition = net_depth_condition, net_width_condition
        self._cfg, self._handle, self._handle_device, self._loaded = cfg, ctypes.c_void_p(), None, None
        self._n_params = sum(p.numel() for p in self.parameters())
        self._layout = None

    # ---- handle management ------------------------------------------------------------
    def _ensure_handle(self):
        params = list(self.parameters())
        dev = params[0].device
        if dev.type != "cuda":
            raise L.MnfError("VanillaNeRFRadianceField must be on a GPU (`.to('cuda')`): libmi355nerf has no CPU fallback")
        lib = L.load_library()
        if self._handle_device != dev:
            if self._handle:
                lib.mnf_vanilla_destroy(self._handle)
                self._handle = ctypes.c_void_p()
            with torch.cuda.device(dev):
                L.check(lib.mnf_vanilla_create(ctypes.byref(self._cfg), ctypes.byref(self._handle)))
            n = ctypes.c_int32(0)
            off, rows, cols = (ctypes.c_int64 * 64)(), (ctypes.c_int32 * 64)(), (ctypes.c_int32 * 64)()
            L.check(lib.mnf_vanilla_param_layout_host(self._handle, 64, ctypes.byref(n), off, rows, cols))
            self._layout = [(int(off[i]), int(rows[i]), int(cols[i])) for i in range(n.value)]
            assert n.value == len(params) and lib.mnf_vanilla_param_count(self._handle) == self._n_params
            for (o, r, c), p in zip(self._layout, params):          # the C side's layout IS named_parameters() order
                assert r * c == p.numel(), (r, c, tuple(p.shape))
            self._handle_device, self._loaded = dev, None
        versions = tuple((p._version, p.data_ptr()) for p in params)
        if versions != self._loaded:
            flat = torch.cat([p.detach().reshape(-1).float() for p in params])
            L.launch(lib.mnf_vanilla_set_params, self._handle, L.ptr(flat))
            self._loaded = versions
        return self._handle

    def __del__(self):
        try:
            if self._handle:
                L.load_library().mnf_vanilla_destroy(self._handle)
        except Exception:
            pass

    # ---- reference call surface (mlp.py:226-245) ---------------------------------------
    def query_opacity(self, x, step_size):
        return self.query_density(x) * step_size

    @torch.no_grad()
    def query_density(self, x):
        h = self._ensure_handle()
        L.require_gpu(x)
        pos = L.contig(x.reshape(-1, 3), torch.float32)
        out = torch.empty(pos.shape[0], device=pos.device)
        L.launch(L.load_library().mnf_vanilla_density, h, L.ptr(pos), pos.shape[0], L.ptr(out))
        return out.view(*x.shape[:-1], 1)

    def forward(self, x, condition=None):
        """-> (rgb [...,3] after sigmoid, sigma [...,1] after relu).  `condition` is either shaped like `x` or
        [num_rays, 3] with x = [num_rays, S, 3] (broadcast over a ray's samples, mlp.py:154-160)."""
        if condition is None:
            raise NotImplementedError("forward() without a view direction is not reachable from the reference's scripts")
        h = self._ensure_handle()
        L.require_gpu(x, condition)
        lead = x.shape[:-1]
        pos = L.contig(x.detach().reshape(-1, 3), torch.float32)
        if condition.shape[:-1] == lead:
            dirs, spd = L.contig(condition.detach().reshape(-1, 3), torch.float32), 1
        else:
            assert condition.dim() == 2 and condition.shape[0] == x.shape[0], f"{condition.shape} v.s. {x.shape}"
            dirs, spd = L.contig(condition.detach(), torch.float32), int(math.prod(x.shape[1:-1]))
        n = pos.shape[0]
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            rgb, sigma = _VanillaFunction.apply(self, pos, dirs, spd, *self.parameters())
        else:
            rgb, sigma = torch.empty(n, 3, device=pos.device), torch.empty(n, device=pos.device)
            L.launch(L.load_library().mnf_vanilla_forward, h, L.ptr(pos), L.ptr(dirs), n, spd, L.ptr(rgb), L.ptr(sigma), None, 0)
        return rgb.view(*lead, 3), sigma.view(*lead, 1)
