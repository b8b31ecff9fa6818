"""Data parallelism for the BPMulT hot path: one process per GPU, gradients
all-reduced by RCCL (torch.distributed backend "nccl") over xGMI.

Replaces the reference's single-process nn.DataParallel (train.py:354-356),
which re-replicates the module and reduces every gradient onto GPU 0 each
step.  Here every rank holds a replica; the batch axis is sharded; the only
exchange per optimizer step is a sum of gradients followed by 1/world_size
(equal to the reference's mean loss over the global batch for equal shards).

The trunk's gradients live in ONE flat fp32 buffer laid out in reverse
execution order (Fusion-GMU, level-2 encoders, level-1 encoders, projections),
so each section can be all-reduced, as a few large messages, on a side stream
as soon as backward has finished it, overlapping with the rest of backward.
The [B,d]-sized tail's parameters (a few hundred KB) are reduced as one
flattened message at the end.
"""
from __future__ import annotations

from typing import List, Optional

import torch
import torch.distributed as dist


class GradSync:
    def __init__(self, model, bucket_bytes: int = 128 << 20, process_group=None):
        self.model, self.pg = model, process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.bucket = bucket_bytes // 4
        self.comm: Optional[torch.cuda.Stream] = None
        self.handles: List = []
        self.active = True           # set False on non-final gradient-accumulation micro-steps
        model._grad_ready_hook = self._on_ready

    def _on_ready(self, flat: torch.Tensor, lo: int, hi: int, events=None) -> None:
        """Called by the trunk's backward when gflat[lo:hi] is final for this step: once `events` have passed
        (a layer's slice: its main- and side-stream work), or everything launched on the current stream so far."""
        if self.world == 1 or not self.active:
            return
        if self.comm is None:
            self.comm = torch.cuda.Stream(device=flat.device)
        if events:
            for ev in events:
                self.comm.wait_event(ev)
        else:
            self.comm.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self.comm):
            for a in range(lo, hi, self.bucket):
                b = min(hi, a + self.bucket)
                self.handles.append(dist.all_reduce(flat[a:b], op=dist.ReduceOp.SUM, group=self.pg, async_op=True))

    def finish(self) -> None:
        """After loss.backward(): reduce the tail, wait for everything, scale by 1/world."""
        if self.world == 1 or not self.active:
            return
        st = self.model._store
        tail = [p.grad for n, p in self.model.named_parameters() if p.grad is not None and n not in st.params]
        for h in self.handles:
            h.wait()
        self.handles.clear()
        if self.comm is not None:
            torch.cuda.current_stream().wait_stream(self.comm)
        inv = 1.0 / self.world
        if tail:
            flat = torch._utils._flatten_dense_tensors(tail)
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.pg)
            flat.mul_(inv)
            for g, f in zip(tail, torch._utils._unflatten_dense_tensors(flat, tail)):
                g.copy_(f)
        st.gflat.mul_(inv)


def reduce_gradients_cpu(params, world: int, group=None) -> None:
    """Reference all-reduce-mean of a parameter list on any backend (used by the gloo tests
    that check shard-sum == global-batch gradient)."""
    grads = [p.grad for p in params if p.grad is not None]
    flat = torch._utils._flatten_dense_tensors(grads)
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    flat.div_(world)
    for g, f in zip(grads, torch._utils._unflatten_dense_tensors(flat, grads)):
        g.copy_(f)
