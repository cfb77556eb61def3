"""CPU-side checks of the product boundary: the C-ABI library builds, loads, exports every symbol
declared in include/mi355nerf.h, and refuses to compute without a GPU (no CPU fallback)."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

import apnrf_amd
from apnrf_amd import _lib as L

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    text = open(os.path.join(REPO, "include", "mi355nerf.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mnf_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as G
    G.build()
    lib = ctypes.CDLL(L.lib_path())
    syms = _header_symbols()
    assert len(syms) >= 18
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/mi355nerf.h but not exported"
    assert set(L.SIGNATURES) == set(syms), "ctypes binding table and header disagree"
    assert apnrf_amd.load_library().mnf_version() == 1


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU behaviour")
def test_no_cpu_fallback():
    from apnrf_amd import nerfacc as NA
    from apnrf_amd.ngp import NGPRadianceField
    with pytest.raises(L.MnfError):
        NA.ray_aabb_intersect(torch.zeros(4, 3), torch.ones(4, 3), torch.tensor([[0.0, 0, 0, 1, 1, 1]]))
    f = NGPRadianceField([0.0, 0, 0, 1, 1, 1], neurons=64, layers=2, num_semantic_classes=29, log2_hashmap_size=12)
    with pytest.raises(L.MnfError):
        f.query_density(torch.rand(8, 3))
    with pytest.raises(NotImplementedError):  # reference behaviour: pack_info is CUDA-only (pack.py:48)
        NA.pack_info(torch.tensor([0, 1, 1]))


def test_field_parameter_layout_matches_oracle():
    from apnrf_amd.ngp import NGPRadianceField
    from apnrf_amd import synthetic as S
    from oracle.field import FieldConfig, param_counts
    for neurons, layers, C in [(128, 2, 29), (64, 4, 29), (128, 4, 13)]:
        f = NGPRadianceField([0.0, 0, 0, 1, 1, 1], neurons=neurons, layers=layers, num_semantic_classes=C)
        pc = param_counts(FieldConfig(aabb=(0, 0, 0, 1, 1, 1), neurons=neurons, layers=layers, num_semantic_classes=C))
        assert f.mlp_base.params.numel() == pc["mlp_base"]
        assert f.mlp_head.params.numel() == pc["mlp_head"]
        assert f.mlp_sem.params.numel() == pc["mlp_sem"]
        p = S.make_field_params(neurons, layers, C)
        assert {k: v.size for k, v in p.items()} == pc
    assert set(f.state_dict().keys()) == {"aabb", "direction_encoding.params", "mlp_base.params", "mlp_head.params", "mlp_sem.params"}


def test_estimator_state_matches_oracle(golden):
    from apnrf_amd.nerfacc import OccGridEstimator
    g = golden("occgrid")
    est = OccGridEstimator(torch.from_numpy(g["roi_aabb"]), resolution=g["resolution"].tolist(), levels=1)
    np.testing.assert_array_equal(est.aabbs.numpy(), g["aabbs"])
    np.testing.assert_array_equal(est.grid_coords.numpy(), g["grid_coords"])
    est.eval()
    with pytest.raises(RuntimeError):
        est.update_every_n_steps(step=0, occ_eval_fn=lambda x: x[:, :1])


def test_mark_invisible_cells_known_answer(golden):
    """The reference's own known-answer test (perception/nerfacc/tests/test_grid.py:207-233) on the mirror estimator
    (pure torch: runs on CPU); the counts are also held by tests/golden/occgrid.npz, captured from the reference."""
    import torch
    from apnrf_amd.nerfacc import OccGridEstimator
    g = OccGridEstimator(roi_aabb=torch.tensor([-1.0, -1.0, -1.0, 1.0, 1.0, 1.0]), resolution=32, levels=4)
    K = torch.tensor([[[100.0, 0, 50.0], [0, 100.0, 50.0], [0, 0, 1]]])
    pose = torch.tensor([[[-1.0, 0.0, 0.0, 0.0], [0.0, 1.0, 0.0, 0.0], [0.0, 0.0, -1.0, 2.5]]])
    g.mark_invisible_cells(K, pose, 100, 100)
    gold = golden("occgrid")
    assert int((g.occs == -1).sum()) == 77660 == int(gold["mark_invisible_neg1"])
    assert int((g.occs == 0).sum()) == 53412 == int(gold["mark_invisible_zero"])


def test_vanilla_field_state_dict_matches_reference(golden):
    """The frequency-PE MLP mirror holds its parameters under the reference's own state_dict keys and shapes
    (tests/golden/vanilla.npz was captured from the reference module), in `named_parameters()` order."""
    from apnrf_amd.mlp import VanillaNeRFRadianceField
    g = golden("vanilla")
    f = VanillaNeRFRadianceField(net_depth=2, net_width=64, skip_layer=None, net_depth_condition=1, net_width_condition=64)
    sd = {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd.")}
    f.load_state_dict(sd, strict=True)
    assert [n for n, _ in f.named_parameters()] == [k[5:] for k in g.files if k.startswith("grad.")]
    assert sum(p.numel() for p in f.parameters()) == 18564
    with pytest.raises(L.MnfError):
        f(torch.zeros(4, 3), torch.zeros(4, 3))                     # no CPU fallback
    dflt = VanillaNeRFRadianceField()                                # reference defaults: depth 8, width 256, skip 4
    assert dflt.mlp.base.hidden_layers[5].in_features == 256 + 63 and dflt.mlp.rgb_layer.hidden_layers[0].in_features == 256 + 27


def test_option_structs_carry_their_size_and_stale_callers_are_refused():
    """include/mi355nerf.h MNF_INIT: by-pointer structs start with `struct_size`; a caller built against an older (shorter) header, or one that
    did not initialise the struct, is refused at the boundary before anything is read from the struct's tail.  No GPU needed:
    the check comes first."""
    lib = apnrf_amd.load_library()
    for cls in (L.FieldConfig, L.TrainOpts, L.RenderOpts, L.VanillaConfig, L.RenderJob):
        s = cls()
        assert s.struct_size == ctypes.sizeof(cls) and cls._fields_[0][0] == "struct_size"
    cfg = L.FieldConfig()
    cfg.neurons, cfg.layers, cfg.num_semantic_classes, cfg.n_levels, cfg.n_features, cfg.log2_hashmap_size = 128, 2, 29, 16, 4, 12
    cfg.base_resolution, cfg.max_resolution = 16, 4096
    h = ctypes.c_void_p()
    for bad in (0, ctypes.sizeof(L.FieldConfig) - 4):
        cfg.struct_size = bad
        assert lib.mnf_field_create(ctypes.byref(cfg), ctypes.byref(h)) == -1          # MNF_ERR_INVALID
        assert b"struct_size" in lib.mnf_last_error()
    vc = L.VanillaConfig()
    vc.net_depth, vc.net_width, vc.skip_layer, vc.net_depth_condition, vc.net_width_condition = 2, 64, 0, 1, 64
    vc.struct_size = ctypes.sizeof(L.VanillaConfig) - 4                                # a caller built against the round-5 header (no size field)
    assert lib.mnf_vanilla_create(ctypes.byref(vc), ctypes.byref(h)) == -1 and b"struct_size" in lib.mnf_last_error()
    jobs = (L.RenderJob * 2)()
    jobs[0].struct_size = ctypes.sizeof(L.RenderJob)                                  # the second element left at 0: every element of the array is checked
    ro = L.RenderOpts()
    assert lib.mnf_render_jobs(jobs, 2, 8, 8, 8, (ctypes.c_float * 6)(0, 0, 0, 1, 1, 1), ctypes.byref(ro), None) == -1
    assert b"jobs[1].struct_size" in lib.mnf_last_error()
    header = open(os.path.join(REPO, "include", "mi355nerf.h")).read()
    assert header.count("typedef struct {") == 5                                       # ... and these five are ALL the structs of the header
    for name in ("mnf_field_config", "mnf_vanilla_config", "mnf_train_opts", "mnf_render_opts", "mnf_render_job"):   # every struct the library takes by pointer declares it first
        body = header[:header.index("} " + name + ";")]
        body = body[body.rindex("typedef struct {"):]
        assert body.split("\n")[1].strip().startswith("uint32_t struct_size;"), name
