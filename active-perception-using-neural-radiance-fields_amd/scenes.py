"""Synthetic scene builder shared by bench.py, __graft_entry__.smoke(), the tools and the tests: the scene boxes and grid
resolutions of the reference's yaml files with a procedural occupancy grid and seeded random-init parameters
(`synthetic.py`), plus the product-side objects built from such a scene dict.  Inputs only: nothing here
computes a result."""
import numpy as np
import torch

from . import synthetic as S

# scripts/config_*.yaml render settings (near plane, step, cone angle, alpha threshold)
RENDER_KW = dict(near_plane=0.1, render_step_size=1e-3, cone_angle=0.004, alpha_thre=0.01)


def make_scene(scene="102344250", neurons=128, layers=2, C=29, seed=0, log2_hashmap_size=19, head_gain=1.0, n_poses=8):
    sc = S.SCENES[scene]
    res = S.grid_resolution(sc["aabb"])
    poses = S.camera_poses(sc["origin"], n_poses)
    occ = S.make_occupancy(res, aabb=sc["aabb"], free_at=[sc["origin"]])
    params = S.make_field_params(neurons, layers, C, seed=seed, log2_hashmap_size=log2_hashmap_size, head_gain=head_gain)
    return dict(name=scene, aabb=np.asarray(sc["aabb"], np.float32), res=res, occ=occ, params=params, poses=poses,
                neurons=neurons, layers=layers, C=C, log2_hashmap_size=log2_hashmap_size)


def hip_field(scene, device="cuda:0", tcnn_output_rounding=False, mfma_bf16=False, tcnn_blend_fp16=False):
    from .ngp import NGPRadianceField
    f = NGPRadianceField(aabb=torch.from_numpy(scene["aabb"]), neurons=scene["neurons"], layers=scene["layers"],
                         num_semantic_classes=scene["C"], log2_hashmap_size=scene["log2_hashmap_size"],
                         tcnn_output_rounding=tcnn_output_rounding, mfma_bf16=mfma_bf16, tcnn_blend_fp16=tcnn_blend_fp16)
    with torch.no_grad():
        f.mlp_base.params.copy_(torch.from_numpy(scene["params"]["mlp_base"]))
        f.mlp_head.params.copy_(torch.from_numpy(scene["params"]["mlp_head"]))
        f.mlp_sem.params.copy_(torch.from_numpy(scene["params"]["mlp_sem"]))
    return f.to(device).eval()


def hip_estimator(scene, device="cuda:0"):
    from .nerfacc import OccGridEstimator
    est = OccGridEstimator(torch.from_numpy(scene["aabb"]), resolution=scene["res"], levels=1)
    est.binaries = torch.from_numpy(scene["occ"])
    est.occs = torch.from_numpy(scene["occ"].reshape(-1).astype(np.float32)) * 0.05
    return est.to(device).eval()
