"""Child process of test_ticket_tile_order_renders_the_same_bits: runs under MNF_LIB_PATH=libmi355nerf_diag.so (the only build that reads
MNF_FIELD_STATIC_TILES) and renders one 800 x 800 view — 40 000 tiles in the early rounds, so the field kernel hands its tiles out by tickets — twice in arrival
order and once with the fixed stride of rounds 1-4.  A ray's samples never leave one tile, so the order the tiles are taken in must not move a bit."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
import helpers as H  # noqa: E402
from apnrf_amd import _lib as L  # noqa: E402
from apnrf_amd import render as RD  # noqa: E402

assert L.lib_path().endswith("_diag.so"), L.lib_path()
DEV = "cuda:0"
bk = torch.zeros(3)
for neurons, layers in ((128, 2), (64, 4)):      # the reference yaml's field and BASELINE config 2's
    scene = H.make_scene("102344529", n_poses=8, head_gain=4.0, neurons=neurons, layers=layers)
    hip, est = H.hip_field(scene), H.hip_estimator(scene)
    c2w = np.stack([RD.pose_to_c2w(p) for p in scene["poses"][:1]]).astype(np.float32)
    K = np.array([[400.0, 0, 400], [0, 400.0, 400], [0, 0, 1.0]])
    rays = RD.generate_image_rays(torch.from_numpy(c2w), 800, 800, K, DEV)
    o, d = rays.origins.reshape(-1, 3).contiguous(), rays.viewdirs.reshape(-1, 3).contiguous()

    def render():
        r = RD.render_views(hip, est, o, d, 640000, 1024, render_bkgd=bk, probabilistic=True, image_hw=(800, 800), n_split=1, **H.RENDER_KW)
        torch.cuda.synchronize()
        return {k: v.clone() for k, v in r.items()}

    os.environ.pop("MNF_FIELD_STATIC_TILES", None)
    a, a2 = render(), render()
    os.environ["MNF_FIELD_STATIC_TILES"] = "1"
    b = render()
    assert float(a["total"][1]) > 8 * 640000 * 4, a["total"]      # the first rounds' launches were above the ticket threshold (16 384 tiles)
    for k in ("rgb", "acc", "depth", "sem", "rgb_var", "depth_var", "total"):
        assert torch.equal(a[k], a2[k]), (neurons, layers, "tickets, twice", k)
        assert torch.equal(a[k], b[k]), (neurons, layers, "tickets against fixed stride", k)
    print("field", neurons, "x", layers, "ok", flush=True)
print("DIAG_TILE_ORDER_OK")
