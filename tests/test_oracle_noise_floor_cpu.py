"""What two faithful fp16 implementations of the field may differ by in the COMPOSITED class logits.

The north star asks for RGB / depth / semantic within 1e-3 absolute.  `sem` is an un-normalised sum of raw logits
(perception/models/radiance_fields/ngp.py:215-220, perception/models/utils.py:450-455), so its size is unbounded: on the trained
stand-in it reaches +-55.  This test evaluates ONE model through the oracle twice — the parity reference, and the same fp16
operands with every layer's fp32 products added in a different order (16-wide k blocks, last block first: what a matrix-core
kernel does differently from a BLAS call) — and measures the composited difference per ray.  Both are correct evaluations of the
reference's arithmetic (fp16 operands, fp32 accumulate); they differ because fp32 rounding moves some hidden activations across
an fp16 rounding boundary.  Measured: the difference is ~1.0e-4 of the ray's largest |logit| at every logit scale, i.e. 4e-3
absolute at |logit| 43 with 20 of 256 rays above 1e-3 — the oracle cannot meet 1e-3 absolute against ITSELF there.  Hence the
bar used for `sem` by bench_parity and the trained-scene GPU tests: max(1e-3, 3e-4 x largest |logit| of the ray) (three times this
floor), 1e-3 absolute wherever logits stay below 3.3; rgb / acc / depth keep 1e-3 absolute (their floor is 1e-4 and below)."""
import numpy as np
import pytest
import torch

import helpers as H


def noise_floor(scene, h=16, w=16, pose=1):
    """-> dict of per-ray max |difference| between the two oracle evaluations, the rays' largest |logit|, and both renders"""
    from oracle import render as R
    from oracle.field import FieldConfig, OracleField
    cfg = FieldConfig(aabb=tuple(float(x) for x in scene["aabb"]), neurons=scene["neurons"], layers=scene["layers"],
                      num_semantic_classes=scene["C"], log2_hashmap_size=scene["log2_hashmap_size"])
    a = OracleField(cfg, scene["params"], "f16")
    b = OracleField(cfg, scene["params"], "f16", accum="k16_reversed")
    o, d = H.view_rays(scene, pose, h=h, w=w)
    bk = torch.zeros(3)
    ra = R.render_test(1024, a, scene["occ"], scene["aabb"][None], o, d, render_bkgd=bk, **H.RENDER_KW)
    rb = R.render_test(1024, b, scene["occ"], scene["aabb"][None], o, d, render_bkgd=bk, **H.RENDER_KW)
    out = {k: (ra[k] - rb[k]).abs().reshape(h * w, -1).max(1).values.numpy() for k in ("rgb", "acc", "depth", "sem")}
    out["mag"] = ra["sem"].abs().max(1).values.numpy()
    out["totals"] = (ra["total_samples"], rb["total_samples"])
    return out


@pytest.mark.parametrize("head_gain", [1.0, 160.0])
def test_composited_logit_noise_floor_scales_with_the_logits(head_gain):
    sc = H.make_scene(log2_hashmap_size=14, head_gain=head_gain)
    nf = noise_floor(sc)
    rel = nf["sem"] / np.maximum(1.0, nf["mag"])
    print(f"head gain {head_gain}: largest |logit| {nf['mag'].max():.2f}; oracle vs oracle with permuted accumulation: sem abs {nf['sem'].max():.2e} "
          f"({int((nf['sem'] > 1e-3).sum())} of {len(rel)} rays above 1e-3), relative {rel.max():.2e}; rgb {nf['rgb'].max():.1e} acc {nf['acc'].max():.1e} "
          f"depth {nf['depth'].max():.1e}; samples {nf['totals']}")
    assert nf["totals"][0] == nf["totals"][1]
    assert nf["rgb"].max() < 5e-4 and nf["acc"].max() < 1e-4 and nf["depth"].max() < 1e-4      # 1e-3 absolute is a meaningful bar for these
    assert rel.max() < 3e-4                                                                     # the bar the product is held to (3x the measured floor)
    if head_gain == 1.0:
        assert nf["mag"].max() < 1.0 and nf["sem"].max() < 1e-4                                 # small logits: 1e-3 absolute holds with margin
    else:
        assert nf["mag"].max() > 30.0
        assert nf["sem"].max() > 1e-3 and (nf["sem"] > 1e-3).sum() >= 5                         # large logits: the oracle misses 1e-3 absolute against itself
        assert rel.max() > 3e-5
