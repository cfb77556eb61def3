"""Shared fixtures for the parity tests: one small synthetic scene evaluated by both the HIP path
(through the C ABI) and the oracle on identical inputs."""
import numpy as np
import torch

import apnrf_amd  # noqa: F401  (registers the package alias)
from apnrf_amd import synthetic as S

RENDER_KW = dict(near_plane=0.1, render_step_size=1e-3, cone_angle=0.004, alpha_thre=0.01)


def make_scene(scene="102344250", neurons=128, layers=2, C=29, seed=0, log2_hashmap_size=19, head_gain=1.0, n_poses=8):
    sc = S.SCENES[scene]
    res = S.grid_resolution(sc["aabb"])
    poses = S.camera_poses(sc["origin"], n_poses)
    occ = S.make_occupancy(res, aabb=sc["aabb"], free_at=[sc["origin"]])
    params = S.make_field_params(neurons, layers, C, seed=seed, log2_hashmap_size=log2_hashmap_size, head_gain=head_gain)
    return dict(name=scene, aabb=np.asarray(sc["aabb"], np.float32), res=res, occ=occ, params=params, poses=poses,
                neurons=neurons, layers=layers, C=C, log2_hashmap_size=log2_hashmap_size)


def oracle_field(scene, precision="f16", requires_grad=False):
    from oracle.field import FieldConfig, OracleField
    cfg = FieldConfig(aabb=tuple(float(x) for x in scene["aabb"]), neurons=scene["neurons"], layers=scene["layers"],
                      num_semantic_classes=scene["C"], log2_hashmap_size=scene["log2_hashmap_size"])
    return OracleField(cfg, scene["params"], precision, requires_grad)


def hip_field(scene, device="cuda:0", tcnn_output_rounding=False, mfma_bf16=False):
    from apnrf_amd.ngp import NGPRadianceField
    f = NGPRadianceField(aabb=torch.from_numpy(scene["aabb"]), neurons=scene["neurons"], layers=scene["layers"],
                         num_semantic_classes=scene["C"], log2_hashmap_size=scene["log2_hashmap_size"],
                         tcnn_output_rounding=tcnn_output_rounding, mfma_bf16=mfma_bf16)
    with torch.no_grad():
        f.mlp_base.params.copy_(torch.from_numpy(scene["params"]["mlp_base"]))
        f.mlp_head.params.copy_(torch.from_numpy(scene["params"]["mlp_head"]))
        f.mlp_sem.params.copy_(torch.from_numpy(scene["params"]["mlp_sem"]))
    return f.to(device).eval()


def hip_estimator(scene, device="cuda:0"):
    from apnrf_amd.nerfacc import OccGridEstimator
    est = OccGridEstimator(torch.from_numpy(scene["aabb"]), resolution=scene["res"], levels=1)
    est.binaries = torch.from_numpy(scene["occ"])
    est.occs = torch.from_numpy(scene["occ"].reshape(-1).astype(np.float32)) * 0.05
    return est.to(device).eval()


def view_rays(scene, pose_idx, width=640, height=640, h=32, w=32):
    """Oracle-side rays of one sub-sampled view (bit-exact against the reference, tests/golden/raygen.npz)."""
    from oracle import render as R
    c2w = R.pose_to_c2w(scene["poses"][pose_idx])
    idx = R.subsample_indices(width * height, h * w)
    focal = 0.5 * width / np.tan(np.pi / 4)
    o, d = R.generate_image_rays(c2w, width, height, focal, idx)
    return o, d
