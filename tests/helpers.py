"""Shared fixtures for the parity tests: one small synthetic scene evaluated by both the HIP path
(through the C ABI) and the oracle on identical inputs.  The scene builder itself lives in the package
(`apnrf_amd.scenes`, also used by bench.py and the tools); what is oracle-side stays here."""
import numpy as np

import apnrf_amd  # noqa: F401  (registers the package alias)
from apnrf_amd import synthetic as S  # noqa: F401
from apnrf_amd.scenes import RENDER_KW, hip_estimator, hip_field, make_scene  # noqa: F401


def oracle_field(scene, precision="f16", requires_grad=False):
    from oracle.field import FieldConfig, OracleField
    cfg = FieldConfig(aabb=tuple(float(x) for x in scene["aabb"]), neurons=scene["neurons"], layers=scene["layers"],
                      num_semantic_classes=scene["C"], log2_hashmap_size=scene["log2_hashmap_size"])
    return OracleField(cfg, scene["params"], precision, requires_grad)


def view_rays(scene, pose_idx, width=640, height=640, h=32, w=32):
    """Oracle-side rays of one sub-sampled view (bit-exact against the reference, tests/golden/raygen.npz)."""
    from oracle import render as R
    c2w = R.pose_to_c2w(scene["poses"][pose_idx])
    idx = R.subsample_indices(width * height, h * w)
    focal = 0.5 * width / np.tan(np.pi / 4)
    o, d = R.generate_image_rays(c2w, width, height, focal, idx)
    return o, d
