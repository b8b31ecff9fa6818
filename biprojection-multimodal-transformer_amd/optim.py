"""Fused flat Adam for the BPMulT trunk.

The reference trains with `torch.optim.Adam(model.parameters(), lr=...)` (train.py:123-125), wraps it in
`ReduceLROnPlateau` (train.py:128-136), steps it after the backward pass (train.py:396-398) and stores
`optimizer.state_dict()` in its checkpoints (train.py:372-379).  Here every trunk parameter is a view into one flat fp32
master buffer and its gradient a view into one flat gradient buffer (engine.ParamStore), so the optimizer step for ~all
of the model is ONE streaming kernel (`bpm_adam_step`) instead of a foreach loop over ~1700 tensors; the few parameters
outside the trunk (final GMU, head, front-ends) go through an ordinary torch.optim.Adam with the same hyper-parameters.

`FusedAdam` IS a `torch.optim.Optimizer` (one param group holding every model parameter), so LR schedulers and the
reference's checkpoint code accept it; `param_groups[0]["lr"]` is read at every step.
"""
from __future__ import annotations

from typing import Optional

import torch

from . import _lib


class FusedAdam(torch.optim.Optimizer):
    """Drop-in for `torch.optim.Adam(model.parameters(), lr, betas, eps, weight_decay)` on a `bpmult_amd` model.

    Differences from torch.optim.Adam, all deliberate: trunk parameters that never receive a gradient keep a zero
    gradient instead of `None` (their update is exactly zero unless weight_decay > 0); `zero_grad()` clears the flat
    gradient buffer in place (fused into the step when `fused_zero_grad=True`); the moments are two flat buffers
    (`state_dict()["flat"]`), not per-parameter tensors."""

    def __init__(self, model, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0,
                 fused_zero_grad: bool = False):
        params = [p for p in model.parameters() if p.requires_grad]
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay))
        self.model = model
        self.fused_zero_grad = fused_zero_grad
        self.step_count = 0
        self.pending_grad_scale: Optional[float] = None      # set by distributed.GradSync.finish(): 1 / world_size
        self._m: Optional[torch.Tensor] = None
        self._v: Optional[torch.Tensor] = None
        self._store_id = None
        self._tail_opt = None

    # hyper-parameters live in the param group (what schedulers write)
    @property
    def lr(self) -> float:
        return self.param_groups[0]["lr"]

    # -- plumbing ---------------------------------------------------------------
    def _store(self):
        st = self.model._ensure_store()
        if self._store_id != id(st):                       # (re)built after .to()/.cuda(): restart the moments
            self._m = torch.zeros_like(st.master)
            self._v = torch.zeros_like(st.master)
            self._store_id = id(st)
            g = self.param_groups[0]
            tail = [p for n, p in self.model.named_parameters() if n not in st.params and p.requires_grad]
            self._tail_opt = torch.optim.Adam(tail, lr=g["lr"], betas=g["betas"], eps=g["eps"],
                                              weight_decay=g["weight_decay"]) if tail else None
        return st

    def zero_grad(self, set_to_none: bool = False) -> None:
        st = self._store()
        st.gflat.zero_()
        if self._tail_opt is not None:
            self._tail_opt.zero_grad(set_to_none=set_to_none)

    @torch.no_grad()
    def step(self, closure=None, grad_scale: Optional[float] = None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        st = self._store()
        g = self.param_groups[0]
        if grad_scale is None:
            grad_scale = self.pending_grad_scale if self.pending_grad_scale is not None else 1.0
        self.pending_grad_scale = None
        self.step_count += 1
        n = st.master.numel()
        if n % 4:
            raise RuntimeError("flat parameter buffer is not a multiple of 4 elements")
        # one launch: the update of every trunk parameter AND the CT shadows of the plain weight matrices, written from
        # the updated masters as they are stored (no second pass over the flat master for the next forward's operands)
        st.adam_step(self._m, self._v, g["lr"], g["betas"][0], g["betas"][1], g["eps"], g["weight_decay"], self.step_count,
                     grad_scale, self.fused_zero_grad)
        if self._tail_opt is not None:
            for tg in self._tail_opt.param_groups:
                tg["lr"], tg["betas"], tg["eps"], tg["weight_decay"] = g["lr"], g["betas"], g["eps"], g["weight_decay"]
            if grad_scale != 1.0:
                for p in self._tail_opt.param_groups[0]["params"]:
                    if p.grad is not None:
                        p.grad.mul_(grad_scale)
            self._tail_opt.step()
        return loss

    def state_dict(self):
        self._store()                                      # the moments exist (zeros) even before the first step
        g = self.param_groups[0]
        return {"step": self.step_count, "flat": {"exp_avg": self._m, "exp_avg_sq": self._v},
                "tail": self._tail_opt.state_dict() if self._tail_opt is not None else None,
                "dropout_step": int(getattr(self.model, "dropout_step", 0)),
                "param_groups": [{k: v for k, v in g.items() if k != "params"}]}

    def load_state_dict(self, sd) -> None:
        self._store()
        self.step_count = sd["step"]
        self._m.copy_(sd["flat"]["exp_avg"])
        self._v.copy_(sd["flat"]["exp_avg_sq"])
        if self._tail_opt is not None and sd.get("tail") is not None:
            self._tail_opt.load_state_dict(sd["tail"])
        if hasattr(self.model, "dropout_step"):
            self.model.dropout_step = int(sd.get("dropout_step", 0))
        for k, v in sd["param_groups"][0].items():
            self.param_groups[0][k] = tuple(v) if k == "betas" else v
