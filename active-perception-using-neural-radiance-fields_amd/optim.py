"""Optimizer for the field's three flat parameter vectors.

`FusedAdam` is `torch.optim.Adam(params, lr, betas, eps)` as the reference constructs it (scripts/pipeline.py:173-178:
weight_decay 0, amsgrad off) with the update of a parameter done by ONE HIP kernel (csrc/train.hip adam_guarded_kernel) instead
of the six foreach passes over the 25 M hash-table entries.  State keys (`step`, `exp_avg`, `exp_avg_sq`) and
`state_dict()` layout are torch's, so checkpoints written through either optimizer load into the other
(pipeline.py:630-635 saves `optimizer.state_dict()`).

The step never synchronises with the host: `state["step"]` is a DEVICE scalar (what torch's own `capturable=True` Adam keeps)
advanced by the kernel, and `step(skip=flag)` takes a device int32 flag — the whole update, including the step count, is
left out on the device when the flag is non-zero.  That is the reference's NaN guard (pipeline.py:520-529: `optimizer.zero_grad();
continue` when any gradient is NaN) without the `.item()` per parameter.

`bind_field(field)`: the optimizer kernel of `mlp_base.params` also writes the rounded new hash-table values into the field
handle's fp16 table, and after the step only the MLP weight fragments are re-derived (a few KB) — the handle does not re-convert
the 25 M table entries on the next forward."""
import ctypes

import torch

from . import _lib as L


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False):
        if weight_decay != 0 or amsgrad:
            raise NotImplementedError("FusedAdam covers the reference configuration: weight_decay=0, amsgrad=False")
        if not 0.0 <= lr or not 0.0 <= eps or not (0.0 <= betas[0] < 1.0 and 0.0 <= betas[1] < 1.0):
            raise ValueError("invalid Adam hyper-parameters")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=0, amsgrad=False))
        self._field = None
        self._scratch = {}          # per parameter: 4 device floats (derived step size, bias correction, skip) — not optimizer state

    def bind_field(self, field):
        """The NGPRadianceField whose parameters this optimizer updates (optional; see the module docstring)."""
        self._field = field
        return self

    def _state_for(self, p):
        st = self.state[p]
        if len(st) == 0:
            st["step"] = torch.zeros((), dtype=torch.float32, device=p.device)
            st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
        if not st["step"].is_cuda or st["step"].device != p.device:      # a state_dict loaded from torch.optim.Adam keeps it on the host
            st["step"] = st["step"].to(device=p.device, dtype=torch.float32)
        return st

    def _step_field(self, field, skip, count_nonfinite, report=None):
        """The bound field's three vectors in ONE C call (`mnf_field_optimizer_step`): guard, three updates, fragment refresh."""
        lib = L.load_library()
        ps = [field.mlp_base.params, field.mlp_head.params, field.mlp_sem.params]
        group = next(g for g in self.param_groups if any(q is ps[0] for q in g["params"]))
        handle = field._ensure_handle()                          # holds the CURRENT parameters before they change
        sts = [self._state_for(p) for p in ps]
        grads = [p.grad.contiguous() for p in ps]
        dev = ps[0].device
        hy = self._scratch.get("field")
        if hy is None or hy.device != dev:
            hy = self._scratch["field"] = torch.zeros(12, dtype=torch.float32, device=dev)
        arr = lambda ts: (ctypes.c_void_p * 3)(*[t.data_ptr() for t in ts])
        beta1, beta2 = group["betas"]
        common = (handle, arr(ps), arr(grads), arr([s_["exp_avg"] for s_ in sts]), arr([s_["exp_avg_sq"] for s_ in sts]),
                  arr([s_["step"] for s_ in sts]), float(group["lr"]), float(beta1), float(beta2), float(group["eps"]), L.ptr(skip), int(count_nonfinite), L.ptr(hy))
        if report is not None:      # (device counters or None, pinned host int64[5]): the step-count kernel writes counts + final skip flag into host memory
            counts, host = report
            L.launch(lib.mnf_field_optimizer_step_report, *common, L.ptr(counts), ctypes.c_void_p(host.data_ptr()))
            self.reported = True
        else:
            L.launch(lib.mnf_field_optimizer_step, *common)
        for p in ps:
            torch.autograd.graph.increment_version(p)
        field._mark_current()

    @torch.no_grad()
    def step(self, closure=None, skip=None, count_nonfinite=False, report=None):
        """`skip`: optional device int32 scalar; non-zero = leave every parameter, moment and step count untouched.
        `count_nonfinite=True` (needs `skip`): the NaN / Inf count of the gradients is added to `skip` first (pipeline.py:520-529).
        `report=(counts_dev or None, pinned int64[5])`: with a bound field the call also writes the train step's four counters and the final skip flag into the
        pinned buffer (`mnf_field_optimizer_step_report`); `self.reported` says whether it did."""
        self.reported = False
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = L.load_library()
        field = self._field
        if field is not None and field.mlp_base.params.is_cuda and len(self.param_groups) == 1 and all(
                p.grad is not None and p.dtype == torch.float32 and p.is_contiguous() for p in (field.mlp_base.params, field.mlp_head.params, field.mlp_sem.params)) and all(
                q.numel() == 0 or any(q is p for p in (field.mlp_base.params, field.mlp_head.params, field.mlp_sem.params)) for q in self.param_groups[0]["params"]):
            self._step_field(field, skip, count_nonfinite, report)
            return loss
        if count_nonfinite:
            count_nan_gradients([p for g_ in self.param_groups for p in g_["params"]], out=skip)
        mirror, mirror_from, touched_field = None, 0, False
        if field is not None and field.mlp_base.params.is_cuda:
            handle = field._ensure_handle()                      # the handle holds the CURRENT parameters before they change
            first = ctypes.c_int64(0)
            mirror = lib.mnf_field_table_mirror(handle, ctypes.byref(first))
            mirror_from = int(first.value)
        for group in self.param_groups:
            beta1, beta2 = group["betas"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                if not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous():
                    raise L.MnfError("FusedAdam needs contiguous fp32 parameters on the GPU")
                st = self._state_for(p)
                hy = self._scratch.get(p)
                if hy is None or hy.device != p.device:
                    hy = self._scratch[p] = torch.zeros(4, dtype=torch.float32, device=p.device)
                g = p.grad.contiguous()
                is_table = field is not None and mirror and p is field.mlp_base.params
                L.launch(lib.mnf_adam_step_guarded, L.ptr(p), L.ptr(g), L.ptr(st["exp_avg"]), L.ptr(st["exp_avg_sq"]), p.numel(),
                         float(group["lr"]), float(beta1), float(beta2), float(group["eps"]), L.ptr(st["step"]), L.ptr(skip), L.ptr(hy),
                         ctypes.c_void_p(mirror) if is_table else None, mirror_from if is_table else 0)
                torch.autograd.graph.increment_version(p)      # in-place update outside autograd's view
                touched_field = touched_field or (field is not None and any(p is q for q in field.parameters()))
        if field is not None and mirror and touched_field:
            field._refresh_after_optimizer()
        return loss


def count_nan_gradients(parameters, out=None) -> torch.Tensor:
    """Number of NaN / Inf gradient entries over `parameters` as a device int32 scalar (one launch per parameter, no host sync):
    the guard of pipeline.py:520-529.  `out`: an existing device int32 scalar to add to (e.g. the skip flag of a train step)."""
    lib = L.load_library()
    count = out
    for p in parameters:
        if p.grad is None:
            continue
        if count is None:
            count = torch.zeros((), dtype=torch.int32, device=p.grad.device)
        g = p.grad.contiguous()
        L.launch(lib.mnf_count_nan, L.ptr(g), g.numel(), L.ptr(count))
    return count if count is not None else torch.zeros((), dtype=torch.int32)
