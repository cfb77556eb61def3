"""ORACLE (test infrastructure only): CPU restatement of the reference radiance field
``NGPRadianceField`` (perception/models/radiance_fields/ngp.py:69-238).

PARITY UNPINNED for everything below the ngp.py call sites.  The arithmetic lives in
tiny-cuda-nn (module ``tinycudann``), installed by the reference from GitHub HEAD with no
commit/tag pin (README.md:47-48, perception/models/requirements.txt:1) and absent from
/root/reference.  This file restates tiny-cuda-nn's published algorithm (Müller et al.,
"Instant Neural Graphics Primitives", and the public tcnn sources as of the reference's era,
Oct 2023):

  * multiresolution hash grid  (tcnn ``GridEncoding``; call site ngp.py:123-133)
      scale_l = 2^(l*log2(per_level_scale)) * base_resolution - 1,  res_l = ceil(scale_l) + 1
      params_l = min(round_up(res_l^3, 8), 2^log2_hashmap_size)
      pos = fma(scale_l, x, 0.5); cell = floor(pos); frac = pos - cell
      index = dense (x + y*res + z*res^2) when res^3 <= params_l, else
              (x*1 ^ y*2654435761 ^ z*805459861) mod params_l          (uint32 arithmetic)
      feature = sum over 8 corners of prod_d(bit_d ? frac_d : 1-frac_d) * table[index]
  * degree-4 real spherical harmonics on 2*u-1 (tcnn ``SphericalHarmonics``; ngp.py:108-121, :205)
  * bias-free ReLU MLPs (tcnn ``FullyFusedMLP``; ngp.py:134-169): input width padded to a
    multiple of 16 with the constant 1.0, output width padded to a multiple of 16,
    weights stored [out][in] row-major, layer order input -> hidden... -> output, and in
    ``NetworkWithInputEncoding`` the network parameters precede the encoding parameters.
  * ``trunc_exp(x - 1)`` density activation (in-tree: ngp.py:23-39, :79), aabb selector
    (ngp.py:171-200), sigmoid on rgb (ngp.py:210-212), raw semantic logits (ngp.py:215-220).

Precision model (``precision="f16"``, the product default, mirrors tcnn's fp16 storage):
parameters are rounded to fp16; hash features, SH values and every hidden activation are
rounded to fp16 where they enter a matrix product; products accumulate in fp32; network
outputs stay fp32.  ``precision="tcnn"`` additionally rounds every network output to fp16, which is
what tiny-cuda-nn hands back (the product's `tcnn_output_rounding=True` / `mnf_field_config.output_fp16`);
``precision="bf16"`` is the product's `mfma_bf16` mode (BASELINE config 5): weights, MLP inputs and hidden
activations rounded to bfloat16 instead of fp16, hash table still fp16, fp32 accumulation and outputs.
tcnn's fp16 ACCUMULATION inside a layer is not emulated (its summation order is not part of any published
contract).  ``precision="f32"`` does no rounding.
"""
import math
from dataclasses import dataclass, field as dc_field
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

PRIMES = (1, 2654435761, 805459861)


@dataclass
class FieldConfig:
    aabb: Tuple[float, ...]
    neurons: int = 128          # ngp.py:77 (reference yaml: 128)
    layers: int = 2             # ngp.py:78 (reference yaml: 2) == tcnn n_hidden_layers
    num_semantic_classes: int = 29
    geo_feat_dim: int = 15
    n_levels: int = 16
    n_features: int = 4
    log2_hashmap_size: int = 19
    base_resolution: int = 16
    max_resolution: int = 4096

    @property
    def head_neurons(self) -> int:
        return self.neurons // 2  # ngp.py:153, :166

    @property
    def sem_out_pad(self) -> int:
        return ((self.num_semantic_classes + 15) // 16) * 16


def grid_levels(cfg: FieldConfig):
    """Per-level (scale f32, resolution, n_params, offset, hashed?).  Scales are computed in
    double and rounded once to fp32 (tcnn uses exp2f/log2f in fp32; a 1-ulp difference in a
    level scale is below every tolerance used here)."""
    pls = math.exp((math.log(cfg.max_resolution) - math.log(cfg.base_resolution)) / (cfg.n_levels - 1))
    log2_pls = math.log2(pls)
    out, offset = [], 0
    for l in range(cfg.n_levels):
        scale = np.float32(2.0 ** (l * log2_pls) * cfg.base_resolution - 1.0)
        res = int(math.ceil(float(scale))) + 1
        dense = res ** 3
        n = min(((dense + 7) // 8) * 8, 1 << cfg.log2_hashmap_size)
        out.append(dict(scale=scale, res=res, n=n, offset=offset, hashed=dense > n))
        offset += n
    return out, offset


def mlp_shapes(cfg: FieldConfig) -> Dict[str, List[Tuple[int, int]]]:
    """[out, in] shapes per layer, padded the way tcnn pads (see module docstring)."""
    W, Wh = cfg.neurons, cfg.head_neurons
    base = [(W, cfg.n_levels * cfg.n_features)] + [(W, W)] * (cfg.layers - 1) + [(16, W)]
    head = [(Wh, 32), (Wh, Wh), (16, Wh)]
    sem = [(Wh, 16), (Wh, Wh), (cfg.sem_out_pad, Wh)]
    return {"base": base, "head": head, "sem": sem}


def param_counts(cfg: FieldConfig) -> Dict[str, int]:
    shapes = mlp_shapes(cfg)
    _, table = grid_levels(cfg)
    return {
        "mlp_base": sum(o * i for o, i in shapes["base"]) + table * cfg.n_features,
        "mlp_head": sum(o * i for o, i in shapes["head"]),
        "mlp_sem": sum(o * i for o, i in shapes["sem"]),
    }


def init_params(cfg: FieldConfig, seed: int = 0) -> Dict[str, np.ndarray]:
    """Random-init flat fp32 parameter vectors in the reference state_dict layout
    (``mlp_base.params`` = base MLP then hash table, ``mlp_head.params``, ``mlp_sem.params``).
    tcnn: grid U(-1e-4, 1e-4), MLP xavier-uniform."""
    rng = np.random.default_rng(seed)
    shapes = mlp_shapes(cfg)
    _, table = grid_levels(cfg)

    def mlp(sh):
        parts = []
        for o, i in sh:
            lim = math.sqrt(6.0 / (o + i))
            parts.append(rng.uniform(-lim, lim, size=o * i).astype(np.float32))
        return np.concatenate(parts)

    grid = rng.uniform(-1e-4, 1e-4, size=table * cfg.n_features).astype(np.float32)
    return {
        "mlp_base": np.concatenate([mlp(shapes["base"]), grid]),
        "mlp_head": mlp(shapes["head"]),
        "mlp_sem": mlp(shapes["sem"]),
    }


def _q(x: torch.Tensor, precision: str) -> torch.Tensor:
    """fp16 (or, precision="bf16", bfloat16) rounding with a straight-through gradient (d round(x)/dx := 1), so that torch
    autograd through this oracle yields the fp32 gradient of the rounded forward — the reference the HIP backward is checked
    against."""
    if precision not in ("f16", "tcnn", "bf16"):
        return x
    r = (lambda t: t.bfloat16().float()) if precision == "bf16" else (lambda t: t.half().float())
    if x.requires_grad:
        return x + (r(x.detach()) - x.detach())
    return r(x)


def _split_mlp(flat: torch.Tensor, shapes) -> List[torch.Tensor]:
    ws, k = [], 0
    for o, i in shapes:
        ws.append(flat[k:k + o * i].view(o, i))
        k += o * i
    return ws


class OracleField:
    """forward / query_density with the ngp.py call surface, torch CPU fp32."""

    def __init__(self, cfg: FieldConfig, params: Dict[str, np.ndarray], precision: str = "f16", requires_grad: bool = False,
                 blend: str = "f32", accum: str = "whole"):
        """blend: precision of the 8-corner interpolation of a hash level.  "f32" (default): fp32 weight x fp16 entry summed in fp32 (one rounding
        when the feature enters the network).  "f16": tiny-cuda-nn's published kernel with T = __half (grid.h `kernel_grid`): the weight is cast to
        half and `result = fma((T)weight, value, result)` runs in half precision, corners in index order.  Which of the two the reference's
        un-pinned tinycudann computes cannot be checked here (parity unpinned); the product's mirror is mnf_field_config.blend_fp16."""
        self.cfg = cfg
        self.blend = blend
        # accum: the ORDER in which a layer's fp32 products are added up.  "whole": one torch matmul per layer (the parity reference).
        # "k16_reversed": the contraction cut into the 16-wide k blocks a matrix-core instruction consumes, block sums added last block
        # first.  Both are correct fp32 accumulations of the same fp16 operands; they differ by fp32 rounding, which the fp16 rounding of
        # the next layer's input turns into occasional one-ulp (2^-11 relative) flips.  Used only to MEASURE that noise floor
        # (tests/test_oracle_noise_floor_cpu.py, bench_parity.noise_floor): what two faithful implementations may differ by.
        self.accum = accum
        self.precision = precision
        self.num_semantic_classes = cfg.num_semantic_classes
        self.aabb = torch.tensor(cfg.aabb, dtype=torch.float32)
        self.levels, self.table_entries = grid_levels(cfg)
        self.shapes = mlp_shapes(cfg)
        # leaf parameter vectors in the reference state_dict layout
        self.p_base = torch.from_numpy(np.asarray(params["mlp_base"], np.float32).copy()).requires_grad_(requires_grad)
        self.p_head = torch.from_numpy(np.asarray(params["mlp_head"], np.float32).copy()).requires_grad_(requires_grad)
        self.p_sem = torch.from_numpy(np.asarray(params["mlp_sem"], np.float32).copy()).requires_grad_(requires_grad)
        self._derive()

    def _derive(self):
        """(re)build the fp16-rounded weight views from the leaf vectors (call again after changing them)"""
        precision = self.precision
        n_mlp = sum(o * i for o, i in self.shapes["base"])
        self.w_base = [_q(w, precision) for w in _split_mlp(self.p_base[:n_mlp], self.shapes["base"])]
        # the hash table is stored in fp16 in every mode ("bf16" only changes the matrix-core operands)
        self.table = _q(self.p_base[n_mlp:].view(self.table_entries, self.cfg.n_features), "f16" if precision == "bf16" else precision)
        self.w_head = [_q(w, precision) for w in _split_mlp(self.p_head, self.shapes["head"])]
        self.w_sem = [_q(w, precision) for w in _split_mlp(self.p_sem, self.shapes["sem"])]

    # ---- encodings -------------------------------------------------------------------
    def hash_encode(self, x: torch.Tensor) -> torch.Tensor:
        """x: [N,3] in aabb-normalised coordinates -> [N, n_levels*n_features] fp32."""
        N = x.shape[0]
        F = self.cfg.n_features
        outs = []
        xd = x.double()
        for l, lv in enumerate(self.levels):
            pos = (xd * float(lv["scale"]) + 0.5).float()          # fmaf(scale, x, 0.5)
            cell = torch.floor(pos)
            frac = pos - cell
            cell = cell.to(torch.int32).to(torch.int64) & 0xFFFFFFFF  # (uint32)(int)floorf
            acc = torch.zeros(N, F, dtype=torch.float32)
            acc16 = np.zeros((N, F), np.float16) if self.blend == "f16" else None
            for corner in range(8):
                w = torch.ones(N, dtype=torch.float32)
                idx3 = []
                for d in range(3):
                    bit = (corner >> d) & 1
                    w = w * (frac[:, d] if bit else (1.0 - frac[:, d]))
                    idx3.append((cell[:, d] + bit) & 0xFFFFFFFF)
                if lv["hashed"]:
                    h = torch.zeros(N, dtype=torch.int64)
                    for d in range(3):
                        h = h ^ ((idx3[d] * PRIMES[d]) & 0xFFFFFFFF)
                    index = h % lv["n"]
                else:
                    res = lv["res"]
                    index = ((idx3[0] + idx3[1] * res + idx3[2] * res * res) & 0xFFFFFFFF) % lv["n"]
                acc = acc + w[:, None] * self.table[lv["offset"] + index]
                if acc16 is not None:
                    # half-precision fused multiply-add: the product of two halves and the half sum are exact in float64; ONE rounding to half (numpy's
                    # float64 -> float16 conversion rounds directly)
                    w16 = w.detach().numpy().astype(np.float16).astype(np.float64)
                    val = self.table[lv["offset"] + index].detach().numpy().astype(np.float64)
                    acc16 = (w16[:, None] * val + acc16.astype(np.float64)).astype(np.float16)
            if acc16 is not None:     # the half-precision value with the fp32 blend's gradient (straight-through, as every rounding of this oracle)
                acc = acc + (torch.from_numpy(acc16.astype(np.float32)) - acc.detach())
            outs.append(acc)
        return torch.cat(outs, -1)

    @staticmethod
    def sh4(dirs01: torch.Tensor) -> torch.Tensor:
        """tcnn SphericalHarmonics degree 4; input in [0,1] (ngp.py:203-205), mapped to [-1,1]."""
        x = dirs01[:, 0] * 2.0 - 1.0
        y = dirs01[:, 1] * 2.0 - 1.0
        z = dirs01[:, 2] * 2.0 - 1.0
        xy, xz, yz = x * y, x * z, y * z
        x2, y2, z2 = x * x, y * y, z * z
        o = [
            torch.full_like(x, 0.28209479177387814),
            -0.48860251190291987 * y,
            0.48860251190291987 * z,
            -0.48860251190291987 * x,
            1.0925484305920792 * xy,
            -1.0925484305920792 * yz,
            0.94617469575755997 * z2 - 0.31539156525251999,
            -1.0925484305920792 * xz,
            0.54627421529603959 * x2 - 0.54627421529603959 * y2,
            0.59004358992664352 * y * (-3.0 * x2 + y2),
            2.8906114426405538 * xy * z,
            0.45704579946446572 * y * (1.0 - 5.0 * z2),
            0.3731763325901154 * z * (5.0 * z2 - 3.0),
            0.45704579946446572 * x * (1.0 - 5.0 * z2),
            1.4453057213202769 * z * (x2 - y2),
            0.59004358992664352 * x * (-x2 + 3.0 * y2),
        ]
        return torch.stack(o, -1)

    def _mm(self, h: torch.Tensor, w: torch.Tensor) -> torch.Tensor:
        if self.accum == "whole":
            return h @ w.t()
        assert self.accum == "k16_reversed", self.accum
        out = None
        for k0 in reversed(range(0, h.shape[1], 16)):
            part = h[:, k0:k0 + 16] @ w[:, k0:k0 + 16].t()
            out = part if out is None else out + part
        return out

    def _mlp(self, h: torch.Tensor, ws: List[torch.Tensor]) -> torch.Tensor:
        h = _q(h, self.precision)
        for w in ws[:-1]:
            h = _q(torch.relu(self._mm(h, w)), self.precision)
        out = self._mm(h, ws[-1])
        # "tcnn": the network hands its outputs over in fp16 (ngp.py:181-200, :210-220 widen them with `.to(x)`)
        return _q(out, "f16") if self.precision == "tcnn" else out

    # ---- ngp.py call surface ---------------------------------------------------------
    def _base(self, positions: torch.Tensor):
        aabb_min, aabb_max = self.aabb[:3], self.aabb[3:]
        x = (positions - aabb_min) / (aabb_max - aabb_min)            # ngp.py:177-178
        selector = ((x > 0.0) & (x < 1.0)).all(dim=-1)               # ngp.py:179
        out = self._mlp(self.hash_encode(x), self.w_base)             # [N,16]
        density = torch.exp(out[:, :1] - 1.0) * selector[:, None]     # ngp.py:79, :193-195
        return density, out[:, 1:1 + self.cfg.geo_feat_dim]

    def query_density(self, positions: torch.Tensor) -> torch.Tensor:
        shp = positions.shape[:-1]
        d, _ = self._base(positions.reshape(-1, 3).float())
        return d.view(*shp, 1)

    def forward(self, positions: torch.Tensor, directions: torch.Tensor):
        positions = positions.reshape(-1, 3).float()
        directions = directions.reshape(-1, 3).float()
        density, geo = self._base(positions)
        one = torch.ones(geo.shape[0], 1)
        sh = self.sh4((directions + 1.0) / 2.0)                       # ngp.py:205-206
        rgb = torch.sigmoid(self._mlp(torch.cat([sh, geo, one], -1), self.w_head)[:, :3])   # ngp.py:207-212
        sem = self._mlp(torch.cat([geo, one], -1), self.w_sem)[:, :self.cfg.num_semantic_classes]  # ngp.py:215-220
        return rgb, density, sem

    __call__ = forward
