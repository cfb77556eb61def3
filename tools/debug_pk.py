import sys, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import helpers as H
sc = H.make_scene(neurons=128, layers=1, C=5, log2_hashmap_size=12, head_gain=4.0)
hip, orc = H.hip_field(sc), H.oracle_field(sc)
rng = np.random.default_rng(1)
n = 5037
a = sc["aabb"]
pos = (rng.random((n, 3)) * (a[3:] - a[:3]) * 1.1 + a[:3] - 0.05 * (a[3:] - a[:3])).astype(np.float32)
d = rng.normal(size=(n, 3)).astype(np.float32); d /= np.linalg.norm(d, axis=-1, keepdims=True)
with torch.no_grad():
    rgb, sigma, sem = hip(torch.from_numpy(pos).cuda(), torch.from_numpy(d).cuda())
r_rgb, r_sigma, r_sem = orc(torch.from_numpy(pos), torch.from_numpy(d))
s, r = sigma.cpu().numpy()[:, 0], r_sigma.numpy()[:, 0]
bad = np.where(np.abs(s - r) > 2e-3 * np.abs(r) + 1e-6)[0]
xn = (pos - a[:3]) / (a[3:] - a[:3])
print("bad", len(bad))
from oracle.field import grid_levels, FieldConfig
cfg = FieldConfig(aabb=tuple(float(x) for x in a), neurons=128, layers=1, num_semantic_classes=5, log2_hashmap_size=12)
lv, _ = grid_levels(cfg)
for i in bad[:12]:
    print(i, i % 64, xn[i], s[i], r[i])
    for l in (0, 1, 2, 3):
        p = xn[i] * lv[l]["scale"] + 0.5
        print("   level", l, "res", lv[l]["res"], "hashed", lv[l]["hashed"], "n", lv[l]["n"], "pos", p, "cell", np.floor(p))
