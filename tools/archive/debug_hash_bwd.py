import os, subprocess, sys
import numpy as np
if len(sys.argv) > 1:
    sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
    import torch
    import helpers as H
    sc = H.make_scene(neurons=128, layers=2, C=29, log2_hashmap_size=14, head_gain=2.0)
    hip = H.hip_field(sc).train()
    rng = np.random.default_rng(7)
    n = 3000 + 21
    a = sc["aabb"]
    pos = (rng.random((n, 3)) * (a[3:] - a[:3]) * 0.98 + a[:3] + 0.01 * (a[3:] - a[:3])).astype(np.float32)
    pos[:5] = a[:3] - 1.0
    d = rng.normal(size=(n, 3)).astype(np.float32); d /= np.linalg.norm(d, axis=-1, keepdims=True)
    g_rgb = (rng.normal(size=(n, 3)) * 1e-3).astype(np.float32)
    g_sem = (rng.normal(size=(n, 29)) * 1e-3).astype(np.float32)
    cu = lambda x: torch.from_numpy(x).cuda()
    rgb, sigma, sem = hip(cu(pos), cu(d))
    torch.autograd.backward([rgb, sem], [cu(g_rgb), cu(g_sem)])
    np.save(sys.argv[1], hip.mlp_base.params.grad.cpu().numpy())
    print(hip.grid_meta()[3])
else:
    env = dict(os.environ)
    subprocess.check_call([sys.executable, __file__, '/tmp/g_pre.npy'], env=env)
    env['MNF_HASH_BWD_SIMPLE'] = '1'
    subprocess.check_call([sys.executable, __file__, '/tmp/g_simple.npy'], env=env)
    a, b = np.load('/tmp/g_pre.npy'), np.load('/tmp/g_simple.npy')
    n_mlp = 128 * 64 + 128 * 128 + 16 * 128
    ta, tb = a[n_mlp:].reshape(-1, 4), b[n_mlp:].reshape(-1, 4)
    print('mlp diff', np.abs(a[:n_mlp] - b[:n_mlp]).max())
    diff = np.abs(ta - tb).max(1)
    idx = np.nonzero(diff > 1e-7 * np.abs(tb).max())[0]
    print('entries differing', len(idx), 'of', (np.abs(tb).max(1) > 0).sum(), 'norm ratio', np.linalg.norm(ta - tb) / np.linalg.norm(tb))
    print(idx[:40])
    for i in idx[:10]:
        print(i, ta[i], tb[i])
