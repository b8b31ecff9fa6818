"""Fused flat Adam for the BPMulT trunk.

The reference trains with `torch.optim.Adam(model.parameters(), lr=...)` (train.py:123-125) and steps it after the
backward pass (train.py:396-398).  Here every trunk parameter is a view into one flat fp32 master buffer and its
gradient a view into one flat gradient buffer (engine.ParamStore), so the optimizer step for ~all of the model is ONE
streaming kernel (`bpm_adam_step`) instead of a foreach loop over ~1700 tensors; the few parameters outside the trunk
(final GMU, head, front-ends) go through an ordinary torch.optim.Adam with the same hyper-parameters.
"""
from __future__ import annotations

from typing import Optional

import torch

from . import _lib


class FusedAdam:
    """Drop-in for `torch.optim.Adam(model.parameters(), lr, betas, eps, weight_decay)` on a `bpmult_amd` model.

    Differences from torch.optim.Adam, all deliberate: trunk parameters that never receive a gradient keep a zero
    gradient instead of `None` (their update is exactly zero unless weight_decay > 0); `zero_grad()` clears the flat
    gradient buffer in place (fused into the step when `fused_zero_grad=True`)."""

    def __init__(self, model, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0,
                 fused_zero_grad: bool = False):
        self.model = model
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.fused_zero_grad = fused_zero_grad
        self.step_count = 0
        self._m: Optional[torch.Tensor] = None
        self._v: Optional[torch.Tensor] = None
        self._store_id = None
        self._tail_opt = None

    # -- plumbing ---------------------------------------------------------------
    def _store(self):
        st = self.model._ensure_store()
        if self._store_id != id(st):                       # (re)built after .to()/.cuda(): restart the moments
            self._m = torch.zeros_like(st.master)
            self._v = torch.zeros_like(st.master)
            self._store_id = id(st)
            tail = [p for n, p in self.model.named_parameters() if n not in st.params and p.requires_grad]
            self._tail_opt = torch.optim.Adam(tail, lr=self.lr, betas=self.betas, eps=self.eps,
                                              weight_decay=self.weight_decay) if tail else None
        return st

    @property
    def param_groups(self):                                # ReduceLROnPlateau & friends read / write lr here
        return [self.__dict__]

    def zero_grad(self, set_to_none: bool = False) -> None:
        st = self._store()
        st.gflat.zero_()
        if self._tail_opt is not None:
            self._tail_opt.zero_grad(set_to_none=set_to_none)

    @torch.no_grad()
    def step(self, grad_scale: float = 1.0) -> None:
        st = self._store()
        self.step_count += 1
        n = st.master.numel()
        pad = (-n) % 4
        if pad:
            raise RuntimeError("flat parameter buffer is not a multiple of 4 elements")
        _lib.check(_lib.lib().bpm_adam_step(st.master.data_ptr(), st.gflat.data_ptr(), self._m.data_ptr(), self._v.data_ptr(), n,
                                            self.lr, self.betas[0], self.betas[1], self.eps, self.weight_decay,
                                            self.step_count, grad_scale, int(self.fused_zero_grad),
                                            torch.cuda.current_stream().cuda_stream), "bpm_adam_step")
        if self._tail_opt is not None:
            for g in self._tail_opt.param_groups:
                g["lr"] = self.lr
            if grad_scale != 1.0:
                for p in self._tail_opt.param_groups[0]["params"]:
                    if p.grad is not None:
                        p.grad.mul_(grad_scale)
            self._tail_opt.step()

    def state_dict(self):
        return {"step": self.step_count, "exp_avg": self._m, "exp_avg_sq": self._v,
                "tail": self._tail_opt.state_dict() if self._tail_opt is not None else None,
                "lr": self.lr, "betas": self.betas, "eps": self.eps, "weight_decay": self.weight_decay}

    def load_state_dict(self, sd) -> None:
        self._store()
        self.step_count = sd["step"]
        self._m.copy_(sd["exp_avg"])
        self._v.copy_(sd["exp_avg_sq"])
        if self._tail_opt is not None and sd.get("tail") is not None:
            self._tail_opt.load_state_dict(sd["tail"])
        self.lr, self.betas, self.eps, self.weight_decay = sd["lr"], tuple(sd["betas"]), sd["eps"], sd["weight_decay"]
