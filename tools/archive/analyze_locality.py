"""Offline locality analysis of the hash-grid gather (CPU, numpy; tools/, not product): which levels' table sectors a render
round touches, and how many distinct 64-B sectors / 128-B lines per sample at three scopes:
  tile   - the 64 samples of one field-kernel tile (16 rays of an 8x8 pixel block x 4 samples): what L1 can merge
  round  - all samples of one round of one view: compulsory misses of a round if nothing survives between rounds
  all    - all rounds analysed: compulsory misses with an infinite cache
Usage: python tools/analyze_locality.py [n_rounds]"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import helpers as H
from oracle import render as R, marcher as M
from oracle.field import FieldConfig, grid_levels

n_rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 4
W = Hh = 800
scene = H.make_scene("102344529", n_poses=40)
c2w = R.pose_to_c2w(scene["poses"][0])
focal = 0.5 * W / np.tan(np.pi / 4)
idx = np.arange(W * Hh)
ys, xs = idx // W, idx % W
key = ((ys // 8) * (W // 8) + xs // 8) * 64 + (ys % 8) * 8 + xs % 8
order = np.argsort(key, kind="stable")
crop = (xs[order] >= 272) & (xs[order] < 528) & (ys[order] >= 272) & (ys[order] < 528)
order = order[crop]      # a 256 x 256 pixel crop in block order (unique-counting 10 M samples x 8 corners x 16 levels is slow)
o, d = R.generate_image_rays(c2w, W, Hh, focal, order)
o, d = np.asarray(o, np.float32), np.asarray(d, np.float32)
n = o.shape[0]
iv, sm, _ = M.traverse_grids(o, d, scene["occ"], scene["aabb"][None], near_planes=np.full(n, 0.1, np.float32),
                             far_planes=np.full(n, 1e10, np.float32), step_size=1e-3, cone_angle=0.004,
                             traverse_steps_limit=4 * n_rounds, over_allocate=True)
left = iv.is_left if hasattr(iv, "is_left") else None
vals = iv.vals
info = sm.packed_info            # [n,2] start,count of samples
vals = np.asarray(vals); ts = vals[np.asarray(iv.is_left)]; te = vals[np.asarray(iv.is_right)]
info = np.asarray(info)
ray_of = np.repeat(np.arange(n), info[:, 1])
k_in_ray = np.arange(ts.shape[0]) - np.repeat(info[:, 0], info[:, 1])
print("rays", n, "samples", ts.shape[0], "per ray", ts.shape[0] / n)
pos = o[ray_of] + d[ray_of] * ((ts + te) / 2)[:, None]
a = scene["aabb"]
xn = (pos - a[:3]) / (a[3:] - a[:3])
cfg = FieldConfig(aabb=tuple(float(x) for x in a), neurons=128, layers=2, num_semantic_classes=29, log2_hashmap_size=19)
levels, _ = grid_levels(cfg)
rnd = k_in_ray // 4
tile = (ray_of // 16).astype(np.int64) * n_rounds + rnd          # one tile = 16 consecutive rays x one round (all alive)
print("%5s %6s %8s | %9s %9s %9s | %9s %9s %9s" % ("level", "res", "hashed", "sec/tile", "sec/round", "sec/all", "line/tile", "line/rnd", "line/all"))
tot = np.zeros(6)
for li, lv in enumerate(levels):
    p = np.asarray(xn, np.float32) * np.float32(lv["scale"]) + np.float32(0.5)
    c0 = np.floor(p).astype(np.int64)
    secs = []
    for corner in range(8):
        c = c0 + np.array([corner & 1, (corner >> 1) & 1, corner >> 2])
        if lv["hashed"]:
            i = (c[:, 0] ^ (c[:, 1] * 2654435761) ^ (c[:, 2] * 805459861)) & 0xFFFFFFFF & (lv["n"] - 1)
        else:
            i = (c[:, 0] + c[:, 1] * lv["res"] + c[:, 2] * lv["res"] ** 2) % lv["n"]
        secs.append((i + lv["offset"]) >> 3)                       # 8-byte entries: 8 per 64-B sector
    S = np.stack(secs, 1)                                          # [N,8]
    row = []
    for shift in (0, 1):                                           # sectors, then 128-B lines
        s = S >> shift
        per_tile = np.unique(np.repeat(tile, 8) * (1 << 26) + s.reshape(-1)).shape[0]
        per_round = sum(np.unique(s[rnd == r]).shape[0] for r in range(n_rounds))
        allr = np.unique(s).shape[0]
        row += [per_tile / S.shape[0], per_round / S.shape[0], allr / S.shape[0]]
    tot += np.array(row)
    print("%5d %6d %8s | %9.3f %9.3f %9.3f | %9.3f %9.3f %9.3f" % (li, lv["res"], lv["hashed"], *row))
print("%21s | %9.3f %9.3f %9.3f | %9.3f %9.3f %9.3f   (per sample; corner accesses per sample: 128)" % ("sum", *tot))
