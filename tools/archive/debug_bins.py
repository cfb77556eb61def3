"""Debug (diag library): table gradient through the bins vs through the walk, per level.  MNF_LIB_PATH=.../libmi355nerf_diag.so python tools/debug_bins.py [lh]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import helpers as H
lh = int(sys.argv[1]) if len(sys.argv) > 1 else 14
n = int(sys.argv[2]) if len(sys.argv) > 2 else 3021
sc = H.make_scene(neurons=128, layers=2, C=29, log2_hashmap_size=lh, head_gain=2.0)
hip = H.hip_field(sc).train()
rng = np.random.default_rng(7)
a = sc["aabb"]
pos = (rng.random((n, 3)) * (a[3:] - a[:3]) * 0.98 + a[:3] + 0.01 * (a[3:] - a[:3])).astype(np.float32)
d = rng.normal(size=(n, 3)).astype(np.float32); d /= np.linalg.norm(d, axis=-1, keepdims=True)
g_rgb = (rng.normal(size=(n, 3)) * 1e-3).astype(np.float32)
g_sig = (rng.normal(size=(n, 1)) * 1e-5).astype(np.float32)
g_sem = (rng.normal(size=(n, 29)) * 1e-3).astype(np.float32)
cu = lambda x: torch.from_numpy(x).cuda()
grads = {}
for b0 in (16, 8, 16, 15):
    os.environ["MNF_BIN_LEVEL0"] = str(b0)
    hip.zero_grad()
    rgb, sigma, sem = hip(cu(pos), cu(d))
    torch.autograd.backward([rgb, sigma, sem], [cu(g_rgb), cu(g_sig), cu(g_sem)])
    torch.cuda.synchronize()
    grads[b0] = hip.mlp_base.params.grad.clone()
n_tab = hip._table_entries() * 4
tab = {k: v[-n_tab:].view(-1, 4) for k, v in grads.items()}
# level offsets: 16 levels, sizes from the field
import ctypes
ref = tab[16]
print("walk vs walk (noise):", float((ref - tab[16]).norm() / ref.norm()))
for k in (8, 15):
    diff = (tab[k] - ref)
    print(f"bins from {k}: rel err {float(diff.norm() / ref.norm()):.3e}; entries differing > 1e-3 rel: {int(((diff.abs() > 1e-3 * ref.abs().max())).any(1).sum())} of {ref.shape[0]}")
    bad = (diff.abs() > 1e-3 * ref.abs().max()).any(1).nonzero().flatten()
    if len(bad):
        print("  first bad entries", bad[:10].tolist(), " last", bad[-5:].tolist())
        e = int(bad[0]); print("  walk", ref[e].tolist(), "bins", tab[k][e].tolist())
