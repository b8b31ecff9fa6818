"""Data parallelism for the BPMulT hot path: one process per GPU, gradients
all-reduced by RCCL (torch.distributed backend "nccl") over xGMI.

Replaces the reference's single-process nn.DataParallel (train.py:354-356),
which re-replicates the module and reduces every gradient onto GPU 0 each
step.  Here every rank holds a replica; the batch axis is sharded; the only
exchange per optimizer step is a sum of gradients followed by 1/world_size
(equal to the reference's mean loss over the global batch for equal shards).
With gradient accumulation (train.py:390-398) only the last micro-step
exchanges: set `sync.active = False` on the others.

The trunk's gradients live in ONE flat fp32 buffer laid out in reverse
execution order (Fusion-GMU, level-2 encoders, level-1 encoders, projections)
at LAYER granularity, so each slice is all-reduced, as a few large messages, on
a communication stream as soon as backward has finished it, overlapping with
the rest of backward.  The [B,d]-sized tail's parameters (a few hundred KB) are
reduced as one flattened message at the end.

* 1/world is NOT a separate pass over the gradient buffer when a FusedAdam is
  attached (`GradSync(model, optimizer=opt)`): finish() leaves the SUM in the
  flat buffer and hands 1/world to the optimizer, whose kernel multiplies the
  gradient as it reads it (`bpm_adam_step(grad_scale)`).  Without an optimizer
  the buffer is scaled in place so that `.grad` holds the mean.
* `compress`: "none" (the library default) exchanges the fp32 gradients, as the
  reference's DataParallel reduction does.  Opt-in: "bf16" exchanges bf16 copies
  of the slices (one rounding of each rank's summand; RCCL accumulates the sum
  in bf16: the error against the fp32 sum is bounded in
  tests/test_gradsync_gloo_cpu.py at world 8), "auto" does so when the fp32
  gradient buffer is larger than AUTO_BF16_BYTES (1 GiB) -- what bench.py asks
  for and states in its JSON line (`grad_exchange.compress`): a ring all-reduce
  moves 2 (w-1)/w x bytes over ONE xGMI link per GPU (~153 GB/s, SURVEY.md
  section 5), so the headline model's 2.7 GB of fp32 gradients are ~31 ms of
  ring time against a 30 ms step -- nothing to hide them behind -- while bf16
  halves that for one rounding of each rank's summand (RCCL accumulates the sum
  in bf16).  hidden 300 (0.44 GB, ~5 ms) stays fp32.  "none" forces fp32,
  "bf16" forces bf16.
"""
from __future__ import annotations

import collections
from typing import List, Optional

import torch
import torch.distributed as dist


AUTO_BF16_BYTES = 1 << 30


class GradSync:
    def __init__(self, model, bucket_bytes: int = 128 << 20, process_group=None, optimizer=None, compress: str = "none"):
        if compress not in ("auto", "none", "bf16"):
            raise ValueError("compress must be 'auto', 'none' or 'bf16'")
        self.model, self.pg = model, process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.bucket = bucket_bytes // 4
        self.optimizer = optimizer
        self.compress_arg = compress
        self.compress = compress if compress != "auto" else None      # resolved at the first exchange (needs the store)
        self.comm: Optional[torch.cuda.Stream] = None
        self.handles: List = []
        self._half: List = []        # (bf16 staging buffer, fp32 slice) pairs of the step in flight
        self.active = True           # set False on non-final gradient-accumulation micro-steps
        # (event before waiting for the exchange, event after) of the most recent steps: bounded, a long training run
        # must not accumulate live hipEvents
        self._exposed = collections.deque(maxlen=64)
        # per message of the step in flight: (event when its slice was ready on the communication stream, bytes); finish()
        # adds the event behind its all-reduce -> `slice_latency_ms` in stats(): how long each slice took from "gradients
        # final" to "reduced", in issue order (the first real multi-GPU run then shows where the exchange queues up)
        self._ready = []
        self._lat = collections.deque(maxlen=16)
        self._bytes = 0
        self._msgs = 0
        self._steps = 0
        model._grad_ready_hook = self._on_ready

    def _resolve(self, flat: torch.Tensor) -> None:
        if self.compress is None:
            self.compress = "bf16" if flat.numel() * 4 > AUTO_BF16_BYTES else "none"

    def _exchange(self, flat: torch.Tensor, lo: int, hi: int) -> None:
        self._resolve(flat)
        for a in range(lo, hi, self.bucket):
            b = min(hi, a + self.bucket)
            if flat.is_cuda:
                ev = torch.cuda.Event(enable_timing=True)
                ev.record(torch.cuda.current_stream())
                self._ready.append((ev, (2 if self.compress == "bf16" else 4) * (b - a)))
            if self.compress == "bf16":
                half = flat[a:b].to(torch.bfloat16)
                self._half.append((half, flat[a:b]))
                self.handles.append(dist.all_reduce(half, op=dist.ReduceOp.SUM, group=self.pg, async_op=True))
                self._bytes += 2 * (b - a)
                self._msgs += 1
            else:
                self.handles.append(dist.all_reduce(flat[a:b], op=dist.ReduceOp.SUM, group=self.pg, async_op=True))
                self._bytes += 4 * (b - a)
                self._msgs += 1

    def _on_ready(self, flat: torch.Tensor, lo: int, hi: int, events=None) -> None:
        """Called by the trunk's backward when gflat[lo:hi] is final for this step: once `events` have passed
        (a layer's slice: its main- and side-stream work), or everything launched on the current stream so far."""
        if self.world == 1 or not self.active:
            return
        if not flat.is_cuda:                 # host tensors (gloo rehearsal of the section protocol): no streams involved
            self._exchange(flat, lo, hi)
            return
        if self.comm is None:
            self.comm = torch.cuda.Stream(device=flat.device)
        if events:
            for ev in events:
                self.comm.wait_event(ev)
        else:
            self.comm.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self.comm):
            self._exchange(flat, lo, hi)

    def finish(self) -> None:
        """After loss.backward(): reduce the tail, wait for everything, apply (or hand on) 1/world."""
        if self.world == 1 or not self.active:
            return
        st = self.model._store
        tail = [p.grad for n, p in self.model.named_parameters() if p.grad is not None and n not in st.params]
        cuda = st.gflat.is_cuda
        if cuda:
            main = torch.cuda.current_stream()
            e0 = torch.cuda.Event(enable_timing=True)
            e0.record(main)
        if cuda and self.comm is not None:
            with torch.cuda.stream(self.comm):          # the communication stream waits for RCCL, the main stream for it
                done = []
                for h in self.handles:
                    h.wait()
                    ev = torch.cuda.Event(enable_timing=True)
                    ev.record(self.comm)
                    done.append(ev)
                for half, dst in self._half:
                    dst.copy_(half)
            main.wait_stream(self.comm)
            if len(done) == len(self._ready):
                self._lat.append([(r, d, n) for (r, n), d in zip(self._ready, done)])
        else:
            for h in self.handles:
                h.wait()
            for half, dst in self._half:
                dst.copy_(half)
        self._half.clear()
        self.handles.clear()
        self._ready = []
        inv = 1.0 / self.world
        if tail:
            flat = torch._utils._flatten_dense_tensors(tail)
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.pg)
            if self.optimizer is None:
                flat.mul_(inv)
            for g, f in zip(tail, torch._utils._unflatten_dense_tensors(flat, tail)):
                g.copy_(f)
        if cuda:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record(main)
            self._exposed.append((e0, e1))
        self._steps += 1
        if self.optimizer is not None:
            self.optimizer.pending_grad_scale = inv      # consumed (and reset) by the next FusedAdam.step()
        else:
            st.gflat.mul_(inv)

    # -- measurement aid for bench.py ---------------------------------------------
    def reset_stats(self) -> None:
        self._exposed.clear()
        self._lat.clear()
        self._bytes = 0
        self._msgs = 0
        self._steps = 0

    def stats(self) -> dict:
        """Exposed exchange time per step: how long the main stream sat between the end of backward and the end of
        the last all-reduce (the part that did not overlap backward)."""
        if not self._steps:
            return {"world": self.world, "steps": 0, "compress": self.compress or self.compress_arg}
        n = self._steps
        out = {"world": self.world, "steps": n, "bytes_per_step": self._bytes // n, "messages_per_step": self._msgs // n,
               "bucket_bytes": self.bucket * 4, "compress": self.compress or self.compress_arg}
        if self._exposed:
            torch.cuda.synchronize()
            ms = [a.elapsed_time(b) for a, b in self._exposed]
            out["exposed_ms_per_step"] = round(sum(ms) / len(ms), 3)
        if self._lat:
            torch.cuda.synchronize()
            steps = [st for st in self._lat if len(st) == len(self._lat[-1])]
            per = [[r.elapsed_time(d) for r, d, _ in st] for st in steps]
            out["slice_latency_ms"] = [round(sum(col) / len(col), 3) for col in zip(*per)]     # ready -> reduced, in issue order
            out["slice_bytes"] = [n for _, _, n in self._lat[-1]]
        return out


def reduce_gradients_cpu(params, world: int, group=None) -> None:
    """Reference all-reduce-mean of a parameter list on any backend (used by the gloo tests
    that check shard-sum == global-batch gradient)."""
    grads = [p.grad for p in params if p.grad is not None]
    flat = torch._utils._flatten_dense_tensors(grads)
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    flat.div_(world)
    for g, f in zip(grads, torch._utils._unflatten_dense_tensors(flat, grads)):
        g.copy_(f)
