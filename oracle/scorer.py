"""ORACLE (test infrastructure only): the predictive-information scorer,
scripts/pipeline.py:666-798 (`ActiveNeRFMapper.probablistic_uncertainty`), numpy float64.

Inputs are the stacked per-ensemble-member renders exactly as pipeline.py:720-725 builds them:
  rgb_var   [M, 1, V, h, w, 3]   depth_var [M, 1, V, h, w]
  acc       [M, 1, V, h, w]      sem       [M, 1, V, h, w, C]   (raw composited logits)
with M ensemble members (2 in the reference) and V views (40).
"""
import numpy as np


def _softmax(x, axis=-1):
    x = x - x.max(axis=axis, keepdims=True)
    e = np.exp(x)
    return e / e.sum(axis=axis, keepdims=True)


def predictive_information_terms(rgb_var, depth_var, acc, sem):
    rgb_var = np.asarray(rgb_var, np.float64)
    depth_var = np.asarray(depth_var, np.float64)
    acc = np.asarray(acc, np.float64)
    sem = np.asarray(sem, np.float64)

    # pipeline.py:727-735
    rgb_ce = np.log(2 * np.pi * np.e * rgb_var + 1e-4) / 2
    rgb_ens_var = np.sum(rgb_var, axis=0) / 2
    rgb_pi = np.mean(np.log(2 * np.pi * np.e * rgb_ens_var + 1e-4) / 2 - np.mean(rgb_ce, axis=0))
    # pipeline.py:737-746
    d_ce = np.log(2 * np.pi * np.e * depth_var + 1e-4) / 2
    d_ens_var = np.sum(depth_var, axis=0) / 2
    d_pi = np.mean(np.log(2 * np.pi * np.e * d_ens_var + 1e-4) / 2 - np.mean(d_ce, axis=0))
    # pipeline.py:748-760
    p = _softmax(sem, -1)
    s_ce = -np.sum((p + 1e-4) * np.log(p + 1e-4), axis=-1)
    p_ens = np.mean(p, axis=0)
    s_ent = -np.sum((p_ens + 1e-4) * np.log(p_ens + 1e-4), axis=-1)
    s_pi = np.mean(s_ent - np.mean(s_ce, axis=0))
    # pipeline.py:762-773
    o_ce = -(acc + 1e-4) * np.log(acc + 1e-4) - (1 - acc + 1e-4) * np.log(1 - acc + 1e-4)
    a_ens = np.mean(acc, axis=0)
    o_ent = -(a_ens + 1e-4) * np.log(a_ens + 1e-4) - (1 - a_ens + 1e-4) * np.log(1 - a_ens + 1e-4)
    o_pi = np.mean(o_ent - np.mean(o_ce, axis=0))
    return rgb_pi, d_pi, s_pi, o_pi


def predictive_information(rgb_var, depth_var, acc, sem) -> float:
    """pipeline.py:775-781: rgb + depth + 3*sem + 2*occ."""
    r, d, s, o = predictive_information_terms(rgb_var, depth_var, acc, sem)
    return float(r + d + 3 * s + 2 * o)


def per_view_terms(rgb_var, depth_var, acc, sem):
    """Per-view means of the four per-pixel terms, shape [V, 4].  Because every view has the
    same pixel count, the trajectory-level means of pipeline.py:735/746/760/773 are the mean over
    views of these rows — this is what lets the scorer shard over views."""
    V = np.asarray(acc).shape[2]
    rows = []
    for v in range(V):
        sl = (slice(None), slice(None), slice(v, v + 1))
        rows.append(predictive_information_terms(np.asarray(rgb_var)[sl], np.asarray(depth_var)[sl],
                                                 np.asarray(acc)[sl], np.asarray(sem)[sl]))
    return np.asarray(rows, np.float64)
