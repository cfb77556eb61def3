import sys, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import helpers as H
sc = H.make_scene(neurons=128, layers=2, C=5, log2_hashmap_size=12, head_gain=4.0)
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
print("bad", len(bad), "of", n, "| lanes of the bad samples:", sorted(set(int(i) % 64 for i in bad)))
from oracle.field import grid_levels, FieldConfig
cfg = FieldConfig(aabb=tuple(float(x) for x in a), neurons=128, layers=2, num_semantic_classes=5, log2_hashmap_size=12)
lv, _ = grid_levels(cfg)
for i in bad[:12]:
    print(i, i % 64, xn[i], s[i], r[i])
    for l in (0, 1, 2, 3):
        p = xn[i] * lv[l]["scale"] + 0.5
        print("   level", l, "res", lv[l]["res"], "hashed", lv[l]["hashed"], "n", lv[l]["n"], "pos", p, "cell", np.floor(p))

# root-cause builds (-DMNF_PK=4 / 5): in-kernel comparison of the packed instruction with scalar-operand v_fma_f32
import ctypes, struct
from apnrf_amd import _lib as L
lib = ctypes.CDLL(L.lib_path())
if hasattr(lib, "mnf_debug_pk_read"):
    buf = (ctypes.c_uint32 * (8 * 512 + 8))()
    torch.cuda.synchronize()
    assert lib.mnf_debug_pk_read(buf, 1) == 0
    f = lambda u: struct.unpack("f", struct.pack("I", u))[0]
    print("in-kernel mismatches of v_pk_fma_f32 vs v_fma_f32:", buf[0])
    lanes = {}
    for k in range(min(buf[0], 512)):
        r = buf[8 + 8 * k: 16 + 8 * k]
        lanes[r[0]] = lanes.get(r[0], 0) + 1
        if k < 10:
            sc, x, y = f(r[1]), f(r[2]), f(r[3])
            print(f"   lane {r[0]} wg {r[7]} scale {sc:.6g} x {x:.6g} y {y:.6g}: packed ({f(r[4]):.6g}, {f(r[5]):.6g}) scalar hi {f(r[6]):.6g} expected ({sc * x + 0.5:.6g}, {sc * y + 0.5:.6g})"
                  f"  -> packed hi = a * y + 0.5 with a = {(f(r[5]) - 0.5) / y if y else float('nan'):.6g} (bits {struct.unpack('I', struct.pack('f', (f(r[5]) - 0.5) / y if y else 0.0))[0]:#010x})")
    print("   lanes:", sorted(lanes.items()))
