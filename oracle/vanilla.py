"""ORACLE (test infrastructure only): the reference's pure-PyTorch field used by BASELINE
config 1 — perception/models/radiance_fields/mlp.py:168-203 (`SinusoidalEncoder`),
:14-101 (`MLP`), :113-165 (`NerfMLP`), :206-245 (`VanillaNeRFRadianceField`) — restated with
numpy.  Pinned against the reference's own module through tests/golden/vanilla_*.npz.

Weights are passed as the reference's ``state_dict`` (name -> array), so golden vectors
captured from the reference load directly.
"""
import math

import numpy as np


def sinusoidal_encode(x, min_deg, max_deg, use_identity=True):
    """mlp.py:184-203."""
    x = np.asarray(x, np.float32)
    if max_deg == min_deg:
        return x
    scales = np.asarray([2 ** i for i in range(min_deg, max_deg)], np.float32)
    xb = (x[..., None, :] * scales[:, None]).reshape(*x.shape[:-1], (max_deg - min_deg) * x.shape[-1])
    latent = np.sin(np.concatenate([xb, xb + np.float32(0.5 * math.pi)], -1))
    if use_identity:
        latent = np.concatenate([x, latent], -1)
    return latent.astype(np.float32)


def _linear(x, sd, prefix):
    return x @ sd[prefix + ".weight"].T + sd[prefix + ".bias"]


def _sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


class VanillaField:
    """VanillaNeRFRadianceField with skip_layer=None (the BASELINE config-1 shape)."""

    def __init__(self, state_dict, net_depth=2, net_depth_condition=1):
        self.sd = {k: np.asarray(v, np.float32) for k, v in state_dict.items()}
        self.net_depth = net_depth
        self.net_depth_condition = net_depth_condition

    def _base(self, x):
        h = sinusoidal_encode(x, 0, 10)
        for i in range(self.net_depth):
            h = np.maximum(_linear(h, self.sd, f"mlp.base.hidden_layers.{i}"), 0)
        return h

    def query_density(self, x):
        h = self._base(x)
        return np.maximum(_linear(h, self.sd, "mlp.sigma_layer.output_layer"), 0)

    def forward(self, x, condition):
        h = self._base(x)
        raw_sigma = _linear(h, self.sd, "mlp.sigma_layer.output_layer")
        cond = sinusoidal_encode(condition, 0, 4)
        bott = _linear(h, self.sd, "mlp.bottleneck_layer.output_layer")
        z = np.concatenate([bott, cond], -1)
        for i in range(self.net_depth_condition):
            z = np.maximum(_linear(z, self.sd, f"mlp.rgb_layer.hidden_layers.{i}"), 0)
        raw_rgb = _linear(z, self.sd, "mlp.rgb_layer.output_layer")
        return _sigmoid(raw_rgb).astype(np.float32), np.maximum(raw_sigma, 0).astype(np.float32)


def render_batched(rgbs, sigmas, t_starts, t_ends, render_bkgd=None):
    """nerfacc `rendering` on batched [R,S] inputs (volrend.py:20-161 with ray_indices=None;
    the exclusive sum falls back to torch.cumsum, scan.py:85-88)."""
    sdt = sigmas * (t_ends - t_starts)
    alphas = 1.0 - np.exp(-sdt)
    excl = np.cumsum(np.concatenate([np.zeros_like(sdt[..., :1]), sdt[..., :-1]], -1), -1, dtype=np.float32)
    trans = np.exp(-excl)
    w = trans * alphas
    colors = (w[..., None] * rgbs).sum(-2)
    opac = w.sum(-1, keepdims=True)
    depths = (w * (t_starts + t_ends) / 2.0).sum(-1, keepdims=True)
    depths = depths / np.maximum(opac, np.finfo(np.float32).eps)
    if render_bkgd is not None:
        colors = colors + render_bkgd * (1.0 - opac)
    return colors.astype(np.float32), opac.astype(np.float32), depths.astype(np.float32), dict(weights=w, trans=trans, alphas=alphas)
