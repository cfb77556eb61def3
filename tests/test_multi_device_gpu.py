"""A model that lives on cuda:1 must run on GPU 1 whatever the caller's current device is (the reference's
DEVICE_GUARD, perception/nerfacc/nerfacc/cuda/csrc/include/utils_cuda.cuh:23-24).  Needs two GPUs: skipped on the
single-GPU boxes."""
import numpy as np
import pytest
import torch

import helpers as H

pytestmark = pytest.mark.gpu


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs")
def test_model_on_second_gpu_with_current_device_zero():
    from apnrf_amd import render as RD
    sc = H.make_scene(log2_hashmap_size=14)
    o, d = H.view_rays(sc, 1, h=16, w=16)
    torch.cuda.set_device(0)
    outs = []
    for dev in ("cuda:0", "cuda:1"):
        hip, est = H.hip_field(sc, dev), H.hip_estimator(sc, dev)
        assert torch.cuda.current_device() == 0
        r = RD.render_views(hip, est, o.to(dev), d.to(dev), 256, 1024, render_bkgd=torch.zeros(3), **H.RENDER_KW)
        assert r["rgb"].device == torch.device(dev)
        outs.append({k: v.cpu().numpy() for k, v in r.items()})
    for k in ("rgb", "acc", "depth", "sem"):
        np.testing.assert_array_equal(outs[0][k], outs[1][k])
