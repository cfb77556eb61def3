"""ORACLE (test infrastructure only): occupancy-grid state and EMA update,
perception/nerfacc/nerfacc/estimators/occ_grid.py:28-78 (state), :328-375 (cell selection),
:377-437 (`_update`), grid.py:195-198 (`_enlarge_aabb`), pipeline.py:113-120 (resolution).

The reference draws cell indices and in-cell jitter from the torch RNG on the CUDA device;
that stream is not reproducible anywhere else, so the update takes the drawn `indices`
and `jitter` as explicit inputs.  numpy fp32.
"""
import numpy as np


def grid_resolution(aabb, cell_size):
    """pipeline.py:113-120: float32 subtraction and division, then int truncation
    (this is what yields Y=21 rather than 22 for scenes 102344529 / 102344280)."""
    aabb = np.asarray(aabb, np.float32)
    return ((aabb[3:] - aabb[:3]) / cell_size).astype(int).tolist()


def enlarge_aabb(aabb, factor):
    aabb = np.asarray(aabb, np.float32)
    center = (aabb[:3] + aabb[3:]) / 2
    extent = (aabb[3:] - aabb[:3]) / 2
    return np.concatenate([center - extent * np.float32(factor), center + extent * np.float32(factor)]).astype(np.float32)


def grid_coords(resolution):
    """occ_grid.py:440-455 `_meshgrid3d(...).reshape(cells, 3)`: 'ij' order, z fastest."""
    r = np.asarray(resolution, int)
    g = np.stack(np.meshgrid(np.arange(r[0]), np.arange(r[1]), np.arange(r[2]), indexing="ij"), -1)
    return g.reshape(-1, 3).astype(np.int64)


def cell_sample_points(indices, jitter, resolution, aabb):
    """occ_grid.py:395-401: x = (coord + U[0,1)) / res ; world = aabb_min + x * (aabb_max - aabb_min)."""
    coords = grid_coords(resolution)[indices].astype(np.float32)
    x = (coords + np.asarray(jitter, np.float32)) / np.asarray(resolution, np.float32)
    aabb = np.asarray(aabb, np.float32)
    return (aabb[:3] + x * (aabb[3:] - aabb[:3])).astype(np.float32)


def ema_update(occs, indices, occ, ema_decay=0.95):
    """occ_grid.py:407-434: occs[ids] = max(occs[ids]*decay, occ) then NaN roll-back.
    (Duplicate indices: last write wins, as in torch index assignment.)"""
    occs = np.asarray(occs, np.float32).copy()
    backup = occs.copy()
    occs[indices] = np.maximum(occs[indices] * np.float32(ema_decay), np.asarray(occ, np.float32))
    nan = np.isnan(occs)
    occs[nan] = backup[nan]
    return occs


def binarize(occs, occ_thre):
    """occ_grid.py:436-437."""
    occs = np.asarray(occs, np.float32)
    thre = min(np.float32(occs[occs >= 0].mean(dtype=np.float32)), np.float32(occ_thre))
    return occs > thre, thre


def planner_path_finding_map(binaries_list, current_state_xzy=None, aabb_xzy=None, voxel_grid_size=0.2):
    """scripts/pipeline.py:1043-1049 + planning/planning_funcs.py:243-266 restated with numpy/scipy:
    swap axes 2,3 of every estimator's [1,X,Y,Z] grid, stack, squeeze, slice [:, :, 8], merge, 3x3 'symm' dilation."""
    from scipy import signal
    vg = np.squeeze(np.array([np.swapaxes(np.asarray(b), 2, 3) for b in binaries_list]))      # [M, X, Z, Y]
    v_merge = sum(vg[m, :, :, 8].astype(np.int32) for m in range(vg.shape[0]))
    path = (v_merge > 1e-4).astype(np.int32)
    path = np.array(signal.convolve2d(path, np.ones((3, 3), int), boundary="symm", mode="same") > 1e-4).astype(np.int32)
    if current_state_xzy is not None:
        v = np.array((np.asarray(current_state_xzy)[:3] - np.asarray(aabb_xzy)[:3]) // voxel_grid_size, dtype=int)
        path[v[1], v[0]] = 0; path[v[1] + 1, v[0]] = 0; path[v[1] - 1, v[0]] = 0
        path[v[1], v[0] + 1] = 0; path[v[1], v[0] - 1] = 0
    return path


def query(x, data, base_aabb):
    """grid.py:201-237 `_query`: occupancy value and mip selector of world points in a 2x-nested multi-level grid.
    x [N,3], data [L,X,Y,Z], base_aabb [6] -> (values * selector, selector).  Pinned against the reference function through
    tests/golden/query.npz (tests/test_oracle_golden.py::test_marcher_samples_in_occupied_cells)."""
    x = np.asarray(x, np.float32)
    data = np.asarray(data)
    a = np.asarray(base_aabb, np.float32)
    x_norm = (x - a[:3]) / (a[3:] - a[:3])
    maxval = np.maximum(np.abs(x_norm - np.float32(0.5)).max(-1), np.float32(0.1))
    exponent = np.frexp(maxval)[1].astype(np.int64)
    mip = np.maximum(exponent + 1, 0)
    selector = mip < data.shape[0]
    scale = (2.0 ** mip).astype(np.float32)
    x_unit = (x_norm - np.float32(0.5)) / scale[:, None] + np.float32(0.5)
    res = np.asarray(data.shape[1:], np.int64)
    ix = np.minimum((x_unit * res.astype(np.float32)).astype(np.int64), res - 1)
    mip = np.minimum(mip, data.shape[0] - 1)
    return data[mip, ix[:, 0], ix[:, 1], ix[:, 2]] * selector, selector
