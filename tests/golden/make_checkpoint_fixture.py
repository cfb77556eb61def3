"""Writes tests/golden/checkpoint_fixture.pth: a checkpoint with the reference's key set (scripts/pipeline.py:630-635:
`occ_grid`, `model` = NGPRadianceField.state_dict() with tcnn's flat `params` vectors, `optimizer_state_dict`) whose
density at any point is computable BY HAND, so that loading it tests the in-vector layout the library assumes
(network parameters before the hash table inside `mlp_base.params`; every matrix [out][in] row-major; layer order
input -> hidden -> output; output row 0 = density logit) without comparing the library with itself.

Construction (neurons 64, 1 hidden layer, 5 classes, T = 2^8):
  hash table      every entry = 0.25            -> the 64 encoded features are 0.25 wherever the point is
  layer 1 [64x64] W1[j][k] = (j + 1) / 2048     -> hidden_j = relu(64 * 0.25 * (j + 1) / 2048) = (j + 1) / 128
  output [16x64]  row 0: W2[0][j] = 1 / (j + 1), rows 1..15 = 0   -> logit = sum_j (j + 1) / 128 / (j + 1) = 0.5
  density = exp(logit - 1) = exp(-0.5) = 0.60653066 inside the box, 0 outside (ngp.py:171-200)
A transposed matrix, a table-first layout or a different output row would give a different number:
  [in][out] layer 1 -> hidden_j = sum_k 0.25 (k + 1) / 2048 = 0.2539 for every j -> logit = 0.2539 * H_64 = 1.2045
The `.pth` is plain torch.save of CPU tensors; nothing here imports the product or the reference.
"""
import os

import numpy as np
import torch

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "checkpoint_fixture.pth")


def table_entries(log2_T=8, n_levels=16, base=16, maxres=4096):
    import math
    pls = math.exp((math.log(maxres) - math.log(base)) / (n_levels - 1))
    total = 0
    for l in range(n_levels):
        scale = np.float32(2.0 ** (l * math.log2(pls)) * base - 1.0)
        res = int(math.ceil(float(scale))) + 1
        total += min(((res ** 3 + 7) // 8) * 8, 1 << log2_T)
    return total


def main():
    W, C = 64, 5
    W1 = (np.arange(1, W + 1, dtype=np.float32)[:, None] / 2048.0) * np.ones((1, 64), np.float32)         # [out j][in k]
    W2 = np.zeros((16, W), np.float32)
    W2[0] = 1.0 / np.arange(1, W + 1, dtype=np.float32)
    table = np.full(table_entries() * 4, 0.25, np.float32)
    rng = np.random.default_rng(0)
    Wh = W // 2
    head = rng.uniform(-0.2, 0.2, Wh * 32 + Wh * Wh + 16 * Wh).astype(np.float32)
    sem = rng.uniform(-0.2, 0.2, Wh * 16 + Wh * Wh + 16 * Wh).astype(np.float32)
    model = {"aabb": torch.tensor([-1.0, -2.0, -3.0, 1.0, 2.0, 3.0]),
             "direction_encoding.params": torch.zeros(0),
             "mlp_base.params": torch.from_numpy(np.concatenate([W1.reshape(-1), W2.reshape(-1), table])),
             "mlp_head.params": torch.from_numpy(head), "mlp_sem.params": torch.from_numpy(sem)}
    occ = torch.zeros(1, 6, 5, 4, dtype=torch.bool)
    occ[0, 1:4, 2, 1:3] = True
    opt = {"state": {}, "param_groups": [{"lr": 1e-3, "betas": (0.9, 0.999), "eps": 1e-15, "weight_decay": 0, "amsgrad": False,
                                          "params": [0, 1, 2, 3]}]}
    torch.save({"occ_grid": occ, "model": model, "optimizer_state_dict": opt,
                "expected": {"points": torch.tensor([[0.0, 0.0, 0.0], [0.9, -1.9, 2.9], [1.5, 0.0, 0.0]]),
                             "density": torch.tensor([0.6065306597, 0.6065306597, 0.0]),
                             "config": {"neurons": W, "layers": 1, "num_semantic_classes": C, "log2_hashmap_size": 8}}}, OUT)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
