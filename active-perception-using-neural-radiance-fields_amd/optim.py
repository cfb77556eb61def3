"""Optimizer for the field's three flat parameter vectors.

`FusedAdam` is `torch.optim.Adam(params, lr, betas, eps)` as the reference constructs it (scripts/pipeline.py:173-178:
weight_decay 0, amsgrad off) with the update of a parameter done by ONE HIP kernel (csrc/train.hip adam_kernel) instead
of the six foreach passes over the 25 M hash-table entries.  State keys (`step`, `exp_avg`, `exp_avg_sq`) and
`state_dict()` layout are torch's, so checkpoints written through either optimizer load into the other
(pipeline.py:630-635 saves `optimizer.state_dict()`)."""
import torch

from . import _lib as L


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False):
        if weight_decay != 0 or amsgrad:
            raise NotImplementedError("FusedAdam covers the reference configuration: weight_decay=0, amsgrad=False")
        if not 0.0 <= lr or not 0.0 <= eps or not (0.0 <= betas[0] < 1.0 and 0.0 <= betas[1] < 1.0):
            raise ValueError("invalid Adam hyper-parameters")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=0, amsgrad=False))

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = L.load_library()
        for group in self.param_groups:
            beta1, beta2 = group["betas"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                if not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous():
                    raise L.MnfError("FusedAdam needs contiguous fp32 parameters on the GPU")
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = torch.tensor(0.0)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["step"] += 1
                g = p.grad.contiguous()
                L.launch(lib.mnf_adam_step, L.ptr(p), L.ptr(g), L.ptr(st["exp_avg"]), L.ptr(st["exp_avg_sq"]), p.numel(),
                                          float(group["lr"]), float(beta1), float(beta2), float(group["eps"]), int(st["step"].item()))
                torch.autograd.graph.increment_version(p)      # in-place update outside autograd's view: the handle reloads
        return loss


def count_nan_gradients(parameters) -> torch.Tensor:
    """Number of NaN gradient entries over `parameters` as a device int32 scalar (one launch per parameter, no host sync):
    the guard of pipeline.py:520-529."""
    lib = L.load_library()
    count = None
    for p in parameters:
        if p.grad is None:
            continue
        if count is None:
            count = torch.zeros((), dtype=torch.int32, device=p.grad.device)
        g = p.grad.contiguous()
        L.launch(lib.mnf_count_nan, L.ptr(g), g.numel(), L.ptr(count))
    return count if count is not None else torch.zeros((), dtype=torch.int32)
