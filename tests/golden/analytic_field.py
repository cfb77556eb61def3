"""A closed-form radiance field with the call surface of NGPRadianceField (perception/models/radiance_fields/ngp.py:171-238:
`.num_semantic_classes`, `.training`, `.train()/.eval()`, `query_density(x[...,3]) -> [...,1]`, `forward(x[N,3], d[N,3]) ->
(rgb [N,3], density [N,1], sem [N,C])`).  INPUT for the glue goldens: tests/golden/make_golden.py hands it to the REFERENCE's
utils.py / occ_grid.py functions, the tests hand the same object to the oracle's restatement (CPU) and to the product's
sampling / compositing entry points (GPU).  It computes no result that is ever compared with itself.

The density uses only elementwise +, -, *, clamp on fp32 tensors, one torch op at a time (no fused multiply-add), so the CPU
and the GPU evaluate it to the same bits and the threshold decisions that depend on it (alpha >= alpha_thre, T >= 1e-4) see
identical densities everywhere.  rgb (sigmoid) and the class logits (a small matrix product) may differ in the last ulps
between devices: they only enter sums.
"""
import numpy as np
import torch


class AnalyticField:
    def __init__(self, num_semantic_classes=29, seed=11, device="cpu"):
        rng = np.random.default_rng(seed)
        self.num_semantic_classes = num_semantic_classes
        self.training = False
        self.device = device
        self.sem_w = torch.from_numpy(rng.normal(size=(7, num_semantic_classes)).astype(np.float32)).to(device)
        self.rgb_w = torch.from_numpy(rng.normal(size=(7, 3)).astype(np.float32)).to(device)
        self.last = None          # (rgb, density, sem) of the latest forward, with retained gradients when autograd is on

    def train(self, mode=True):
        self.training = mode
        return self

    def eval(self):
        return self.train(False)

    def parameters(self):
        return []

    def _density(self, x):
        u = x[..., 0] * 0.37
        v = x[..., 1] * 1.9
        w = x[..., 2] * 0.41
        s = u * w
        s = s + v * v
        s = s - u * u * 0.5
        s = s + w * 0.25
        s = s * 4.0
        s = s - 1.0
        return torch.clamp(s, min=0.0, max=14.0)

    def query_density(self, x):
        return self._density(x)[..., None]

    def forward(self, x, d):
        feat = torch.cat([x * 0.3, d, torch.ones_like(x[..., :1])], -1)
        rgb = torch.sigmoid(feat @ self.rgb_w)
        sem = (feat @ self.sem_w) * 2.0
        density = self.query_density(x)
        if torch.is_grad_enabled():
            rgb.requires_grad_(True)
            density.requires_grad_(True)
            sem.requires_grad_(True)
        self.last = (rgb, density, sem)
        return rgb, density, sem

    __call__ = forward
