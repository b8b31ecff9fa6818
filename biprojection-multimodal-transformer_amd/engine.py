"""Host-side orchestration of the BPMulT hot path on one MI355X.

Nothing here computes: it owns device buffers (allocated through torch), builds
the per-layer problem tables once per batch shape, and replays them through the
C ABI (ops.py) in forward and in a hand-ordered backward.  The design follows
the hardware, not the reference's call order:

* ParamStore -- every trunk parameter is a view into ONE flat fp32 master
  buffer, its gradient a view into ONE flat fp32 gradient buffer (what the
  RCCL all-reduce buckets and a fused optimizer want), and every 2-D weight has
  a CT (f32 / bf16) "shadow" with a 32-padded leading dimension, refreshed by a
  single table-driven launch per step.
* EncoderGroupPlan -- the independent encoders of one level (six in BPMulT,
  SURVEY.md 3.2) advance layer by layer in lock-step; every kernel launch of a
  layer serves all of them (grouped GEMM / attention / row kernels), so small
  per-encoder problems still fill 256 CUs and launches drop 6x.
* Activations needed by backward are kept in per-layer buffers in the layout
  the backward GEMMs consume (CT row-major for wgrad operands, head-major for
  attention); the [T,S] attention matrix is never stored (only row LSE).

Reference semantics implemented (file:line): transformer.py:52-93 (encoder),
141-195 (layer variants), multihead_attention.py:52-135, mmtr.py:189-195 (GMU).
"""
from __future__ import annotations

import ctypes as C
import math
import os
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import torch

from . import ops
from ._lib import (BPM_BF16, AdamSeg, AddnProblem, ExpandProblem, F_CT_NARROW, F_ACCUM, F_BACKGROUND, F_KPAD, F_RELU, GEMM_NN, GEMM_NT, GEMM_TN, OUT_CT, OUT_HEADS,
                   AttnProblem, CastProblem, FoldDesc, GemmProblem, LnProblem, PackDesc,
                   UnfoldDesc)
from .ops import pad32

# dropout site ids (unique per encoder / layer / op; the seed changes per step)
S_EMB_Q, S_EMB_K, S_EMB_V, S_ATTN, S_RES1, S_RELU, S_RES2, S_ATTN_SELF, S_RES0 = range(9)
SITE_TEXT = 1 << 20


def site(enc_id: int, layer: int, op: int) -> int:
    return (enc_id << 12) | (layer << 4) | op


def dhp_for(dh: int) -> int:
    if dh <= 32:
        return 32
    if dh <= 64:
        return 64
    if dh <= 128:
        return 128
    raise ValueError(f"head_dim {dh} > 128 is not supported by the attention kernels")


_TABLES: Dict[Tuple[int, int, str], torch.Tensor] = {}


def sinusoid_table(n_pos: int, d: int, device) -> torch.Tensor:
    """fp32 [n_pos, d] table of position_embedding.py:44-60, built once on the
    host with the same torch CPU ops as the reference (bit-identical rows) and
    kept resident on the device."""
    key = (d, str(device))
    t = _TABLES.get(key)
    if t is None or t.shape[0] < n_pos:
        n = max(n_pos, 513)
        half = d // 2
        step = math.log(10000.0) / (half - 1)
        freq = torch.exp(torch.arange(half, dtype=torch.float32) * -step)
        ang = torch.arange(n, dtype=torch.float32)[:, None] * freq[None, :]
        tab = torch.cat([ang.sin(), ang.cos()], dim=1)
        if d % 2 == 1:
            tab = torch.cat([tab, torch.zeros(n, 1)], dim=1)
        tab[0].zero_()
        t = tab.contiguous().to(device)
        _TABLES[key] = t
    return t


# ----------------------------------------------------------------------------
# parameters
# ----------------------------------------------------------------------------
class ParamStore:
    """Flat fp32 master + gradient buffers and CT weight shadows for a set of
    nn.Parameters (all on one CUDA device)."""

    ALIGN = 64  # elements

    def __init__(self, named_params: Sequence[Tuple[str, torch.nn.Parameter]], dtype: int, x3: bool = False):
        self.dtype = dtype
        self.x3 = bool(x3) and dtype != BPM_BF16      # bf16x3 mode: fp32 storage, large GEMMs as three split-bf16 products
        self.side_low = True                            # side-stream priority: set by the plan that launches over this store
        self.names = [n for n, _ in named_params]
        self.params = {n: p for n, p in named_params}
        dev = named_params[0][1].device
        if dev.type != "cuda" and not ops._DRY_RUN:
            raise RuntimeError("BPMulT hot path: parameters must live on a CUDA (HIP) device; there is no CPU path")
        self.device = dev
        self.off: Dict[str, int] = {}
        total = 0
        for n, p in named_params:
            self.off[n] = total
            total += (p.numel() + self.ALIGN - 1) // self.ALIGN * self.ALIGN
        self.total = total
        self.master = torch.zeros(total, device=dev, dtype=torch.float32)
        self.gflat = torch.zeros(total, device=dev, dtype=torch.float32)
        with torch.no_grad():
            for n, p in named_params:
                v = self.master[self.off[n]: self.off[n] + p.numel()].view(p.shape)
                v.copy_(p.data)
                p.data = v
        self._gviews = {n: self.gflat[self.off[n]: self.off[n] + p.numel()].view(p.shape) for n, p in named_params}
        self._shadow_specs: List[Tuple[str, int, int, int, int, int, int, int]] = []
        self._shadow_off: Dict[str, int] = {}
        self._shadow_total = 0
        self._fold_specs: List[tuple] = []
        self._fold_off: Dict[str, int] = {}
        self._fold_total = 0
        self._fold_table = None
        self.shadow_flat: Optional[torch.Tensor] = None
        self._table = None
        self._master_ptr = self.master.data_ptr()

    def __del__(self):
        r = getattr(self, "_x3_range", None)
        if r is not None:
            try:
                ops.x3_drop_static(*r)
            except Exception:              # noqa: BLE001 -- interpreter shutdown
                pass

    # -- masters / grads ------------------------------------------------------
    def p(self, name: str) -> torch.Tensor:
        return self.params[name].data

    def g(self, name: str) -> torch.Tensor:
        return self._gviews[name]

    def gptr(self, name: str, elem_off: int = 0) -> int:
        return self._gviews[name].data_ptr() + 4 * elem_off

    def still_flat(self) -> bool:
        """False after anything re-pointed a parameter's storage (model.to()/.cuda()/.float(), `p.data = ...`, a
        child module that built a ParamStore of its own): the launch tables cache raw pointers into `master`, so
        every parameter is checked (~1700 integer comparisons)."""
        base = self._master_ptr
        return all(p.data_ptr() == base + 4 * self.off[n] for n, p in self.params.items())

    def _fresh(self) -> bool:
        """Gradients are unset (backward starts from zero) iff the first TRAINABLE parameter has no .grad: a frozen
        first parameter never gets one and must not make every micro-step look fresh."""
        for n in self.names:
            p = self.params[n]
            if p.requires_grad:
                return p.grad is None
        return True

    def set_store_written(self, names) -> None:
        """`names`: the parameters whose gradient the backward launch tables WRITE (plain store by their first
        weight-gradient launch of a step) when the gradients start from zero -- the large encoder matrices.  Everything
        else (biases, LayerNorm affines, Fusion-GMU / projection / unused parameters: sums of several launches, or
        never written) is what begin_backward(stores=True) clears, through one table-driven launch."""
        skip = set(names)
        segs, cur = [], None
        for n in self.names:                               # flat order: runs of consecutive other parameters (padding included)
            a, b = self.off[n], self.off[n] + self.params[n].numel()
            if n in skip:
                if cur is not None:
                    segs.append(tuple(cur))
                cur = None
            elif cur is None:
                cur = [a, b]
            else:
                cur[1] = b
        if cur is not None:
            segs.append(tuple(cur))
        base = self.gflat.data_ptr()
        self._zero_table = ops.zero_table([(base + 4 * a, b - a) for a, b in segs]) if segs else None
        self._store_written = skip

    def begin_backward(self, stores: bool = False) -> bool:
        """Gradients accumulate into gflat like autograd accumulates into .grad: a parameter whose .grad is None starts
        from zero.  stores=True (the caller's launch tables have a first-writer-stores variant, set_store_written): when
        the gradients are unset, only the small tensors are cleared and True is returned -- the caller must then run the
        storing tables."""
        fresh = self._fresh()
        if stores and fresh and getattr(self, "_zero_table", None) is not None:
            ops.zero_segments(*self._zero_table)
            return True
        if fresh:
            self.gflat.zero_()
        return False

    def end_backward(self) -> None:
        for n in self.names:
            p = self.params[n]
            if p.grad is None and p.requires_grad:
                p.grad = self._gviews[n]

    # -- shadows --------------------------------------------------------------
    def add_shadow(self, key: str, name: str, rows: int, cols: int, *, src_col0: int = 0, src_ld: Optional[int] = None,
                   dst_ld: Optional[int] = None, dst_col0: int = 0, base_key: Optional[str] = None, src_row0: int = 0,
                   colscale: Optional[str] = None, dst_row0: int = 0) -> None:
        """Register a CT shadow [rows, dst_ld] of master `name` viewed as [.., src_ld][src_row0:src_row0+rows,
        src_col0:src_col0+cols].  base_key: write into an already registered shadow (block at row dst_row0, column
        dst_col0) instead of a new one.  colscale: name of a [cols] parameter multiplied into the columns (LayerNorm gain
        folding)."""
        ld = pad32(cols)
        src_ld = cols if src_ld is None else src_ld
        dst_ld = ld if dst_ld is None else dst_ld
        if base_key is None:
            self._shadow_off[key] = self._shadow_total
            off = self._shadow_total
            self._shadow_total += rows * dst_ld
        else:
            off = self._shadow_off[base_key] + dst_row0 * dst_ld
            self._shadow_off[key] = off + dst_col0
        self._shadow_specs.append((name, rows, cols, ld, src_ld, dst_ld, src_col0 + src_row0 * src_ld, off + dst_col0, colscale))

    def add_blank_shadow(self, key: str, rows: int, ld: int) -> None:
        """Reserve a zero-filled CT region [rows, ld] that later add_shadow(base_key=key, ...) calls fill block by block."""
        self._shadow_off[key] = self._shadow_total
        self._shadow_total += rows * ld

    def add_fold(self, key: str, wname: str, row0: int, rows: int, cols: int, beta: str, bias: str, bias_off: int) -> None:
        """Register a folded bias out[rows] = bias[bias_off:] + W[row0:row0+rows, :cols] . beta (fp32)."""
        self._fold_off[key] = self._fold_total
        self._fold_specs.append((wname, row0, rows, cols, beta, bias, bias_off, self._fold_total))
        self._fold_total += (rows + 63) // 64 * 64

    def fold(self, key: str, elem_off: int, n: int) -> torch.Tensor:
        o = self._fold_off[key] + elem_off
        return self.fold_flat[o:o + n]

    def finalize_shadows(self) -> None:
        ct = ops.ct_torch(self.dtype)
        self.shadow_flat = torch.zeros(max(self._shadow_total, 32), device=self.device, dtype=ct)
        esz = self.shadow_flat.element_size()
        descs, blk = [], 0
        for (name, rows, cols, ld, src_ld, dst_ld, src_off, off, colscale) in self._shadow_specs:
            d = PackDesc()
            d.src = self.params[name].data_ptr() + 4 * src_off
            d.dst = self.shadow_flat.data_ptr() + esz * off
            d.rows, d.cols, d.ld, d.src_ld, d.dst_ld, d.blk0 = rows, cols, ld, src_ld, dst_ld, blk
            d.colscale = self.params[colscale].data_ptr() if colscale else None
            blk += (rows * ld + 1023) // 1024
            descs.append(d)
        if self.x3:                                    # the weight shadows are the static operands of the bf16x3 products
            self._x3_range = (self.shadow_flat.data_ptr(), self.shadow_flat.data_ptr() + self.shadow_flat.numel() * esz)
            ops.x3_register_static(*self._x3_range)
        self._ndesc, self._nblk = len(descs), blk
        if descs:
            arr = (PackDesc * len(descs))(*descs)
            raw = bytes(memoryview(arr))
            self._table = torch.frombuffer(bytearray(raw), dtype=torch.uint8)
            self._table = self._table if ops._DRY_RUN else self._table.to(self.device)
        self._build_adam_table()
        self.fold_flat = torch.zeros(max(self._fold_total, 64), device=self.device, dtype=torch.float32)
        fd, blk = [], 0
        for (wname, row0, rows, cols, beta, bias, bias_off, off) in self._fold_specs:
            w = self.params[wname]
            ldw = w.shape[1]
            f = FoldDesc()
            f.W = w.data_ptr() + 4 * row0 * ldw
            f.beta = self.params[beta].data_ptr()
            f.b = self.params[bias].data_ptr() + 4 * bias_off
            f.out = self.fold_flat.data_ptr() + 4 * off
            f.rows, f.cols, f.ldw, f.blk0 = rows, cols, ldw, blk
            blk += (rows + 3) // 4
            fd.append(f)
        self._nfold, self._fold_blk = len(fd), blk
        self._fold_table = ops.device_table(fd) if fd else None

    def _build_adam_table(self) -> None:
        """Segment table of the fused optimizer step (bpm_adam_step_table) and the pack table of what it leaves over.
        A shadow is written BY THE OPTIMIZER KERNEL when it is the plain CT copy of a whole parameter matrix (the large
        encoder matrices, the projections, the GMU hidden maps, the time maps); shadows that mix two parameters (the K / V
        projection weights with their LayerNorm gain folded in) or re-arrange columns (the x_gate halves) stay with a second,
        small pack_weights launch (`_rest_table`), as do the folded biases."""
        esz = 2 if self.dtype == BPM_BF16 else 4
        plain, rest = {}, []
        for spec in self._shadow_specs:
            (name, rows, cols, ld, src_ld, dst_ld, src_off, off, colscale) = spec
            p = self.params[name]
            if (colscale is None and src_off == 0 and src_ld == cols and rows * cols == p.numel() and cols % 4 == 0
                    and name not in plain and (off * esz) % 16 == 0 and dst_ld % 4 == 0):
                plain[name] = (rows, cols, dst_ld, off)
            else:
                rest.append(spec)
        descs, blk = [], 0
        for (name, rows, cols, ld, src_ld, dst_ld, src_off, off, colscale) in rest:
            d = PackDesc()
            d.src = self.params[name].data_ptr() + 4 * src_off
            d.dst = self.shadow_flat.data_ptr() + esz * off
            d.rows, d.cols, d.ld, d.src_ld, d.dst_ld, d.blk0 = rows, cols, ld, src_ld, dst_ld, blk
            d.colscale = self.params[colscale].data_ptr() if colscale else None
            blk += (rows * ld + 1023) // 1024
            descs.append(d)
        self._rest_table = (ops.device_table(descs), len(descs), blk) if descs else None
        nblk = ops.adam_blocks
        segs, blk, run = [], 0, None              # run = [off, end) of consecutive parameters without a plain shadow
        for n in self.names:
            a = self.off[n]
            b = a + (self.params[n].numel() + self.ALIGN - 1) // self.ALIGN * self.ALIGN
            if n in plain:
                if run is not None:
                    segs.append((run[0], run[1], None))
                    run = None
                segs.append((a, b, plain[n]))
            elif run is None:
                run = [a, b]
            else:
                run[1] = b
        if run is not None:
            segs.append((run[0], run[1], None))
        assert segs and segs[0][0] == 0 and segs[-1][1] == self.total and all(x[1] == y[0] for x, y in zip(segs, segs[1:]))
        out = []
        for a, b, sh in segs:
            sg = AdamSeg()
            sg.off4, sg.n4, sg.blk0 = a // 4, (b - a) // 4, blk
            if sh is not None:
                rows, cols, dst_ld, off = sh
                sg.dst, sg.rows, sg.cols, sg.dst_ld = self.shadow_flat.data_ptr() + esz * off, rows, cols, dst_ld
            blk += nblk((b - a) // 4)
            out.append(sg)
        self._adam_table = (ops.device_table(out), len(out), blk)
        self._adam_plain = plain

    def adam_step(self, exp_avg: torch.Tensor, exp_avg_sq: torch.Tensor, lr, beta1, beta2, eps, weight_decay, step, grad_scale,
                  zero_grad: bool) -> None:
        """One launch: torch.optim.Adam's update of every trunk parameter (flat master / gradient / moments) AND the CT
        shadows of the plain weight matrices, written from the updated values as they are stored.  What is left for the
        next forward's refresh_shadows is the small rest (K / V weights with the LayerNorm gain folded in, folded biases)."""
        tab, nseg, nblk = self._adam_table
        ops.adam_step_table(self.dtype, tab, nseg, nblk, self.master, self.gflat, exp_avg, exp_avg_sq, lr, beta1, beta2, eps,
                            weight_decay, step, grad_scale, zero_grad)
        # every plain shadow now holds the CT image of its updated master (whatever was pending before the step); the rest
        # (shadows that mix parameters, folded biases) is stale until the next refresh_shadows
        self._dirty, self._dirty_rest, self._shadow_sig = False, True, self._versions()

    def sptr(self, key: str, elem_off: int = 0) -> int:
        return self.shadow_flat.data_ptr() + self.shadow_flat.element_size() * (self._shadow_off[key] + elem_off)

    def mark_dirty(self) -> None:
        """The master weights were changed through raw pointers (bpm_adam_step) or through a `p.data` alias (neither
        bumps a version counter torch lets us see): the CT shadows are stale.  Writers of `p.data` MUST call this."""
        self._dirty = True

    invalidate = mark_dirty

    def broadcast(self, src: int = 0, group=None) -> None:
        """Replicate rank `src`'s flat master on every rank (one message) and invalidate the shadows."""
        import torch.distributed as dist
        dist.broadcast(self.master, src, group=group)
        self.mark_dirty()

    def _versions(self) -> int:
        # in-place ops on a parameter bump p._version; in-place ops on the flat master itself (dist.broadcast(master),
        # master.mul_(..), EMA / averaging written on the master) bump master._version only
        return self.master._version + sum(p._version for p in self.params.values())

    def refresh_shadows(self, force: bool = False) -> None:
        """Re-derive the CT weight shadows and folded biases from the fp32 masters -- only when the masters changed
        since the last refresh (an optimizer step, load_state_dict, any in-place edit: torch's per-tensor version
        counters, or mark_dirty() for raw-pointer writers).  In a training loop that is once per optimizer step."""
        sig = self._versions()
        full = force or getattr(self, "_dirty", True) or sig != getattr(self, "_shadow_sig", None)
        if not full and not getattr(self, "_dirty_rest", False):
            return
        self._dirty, self._dirty_rest, self._shadow_sig = False, False, sig
        if full:
            if self._table is not None:
                ops.pack_weights(self.dtype, self._table, self._ndesc, self._nblk)
        elif self._rest_table is not None:         # after a fused optimizer step: it wrote the plain shadows itself
            ops.pack_weights(self.dtype, *self._rest_table)
        if self._fold_table is not None:
            ops.fold_bias(self._fold_table, self._nfold, self._fold_blk)
        if self.x3:
            ops.x3_refresh_static()                # split images of the shadows that just changed (outside any graph)


# ----------------------------------------------------------------------------
# encoder group
# ----------------------------------------------------------------------------
@dataclass
class EncoderDesc:
    """One encoder of a lock-step group.  `prefix` selects its parameters in the
    ParamStore (reference state_dict names, e.g. 'trans_l_with_a.')."""
    prefix: str
    enc_id: int
    T: int
    S: int
    attn_dropout: float
    # the T query rows are rows q_pos0 + i*q_stride of a length-T_full sequence (a gathered subset: SURVEY A.10);
    # positions and the attention mask follow the ORIGINAL time steps
    q_pos0: int = 0
    q_stride: int = 1
    T_full: Optional[int] = None
    # biprojection encoders only: the consumer reads rows {0, T-1} of the output, and under the causal
    # self-attention a row never sees a later one, so the LAST layer's query side (self-attention queries, cross
    # attention, FFN) and the final LayerNorm run on those two rows; its self-attention keys / values and every
    # earlier layer stay dense.  The output is then [2, B, d].
    tail_rows: bool = False


@dataclass
class GroupCfg:
    d: int
    H: int
    layers: int
    relu_dropout: float
    res_dropout: float
    embed_dropout: float
    attn_mask: bool
    biprojection: bool


SIDE, JOIN, MARK, WAIT, SIDE2 = "side", "join", "mark", "wait", "side2"
_SIDE = os.environ.get("BPMULT_SIDE", "1") != "0"
# dK/dV attention pass: "0" main stream, "1" side stream, "2" a third stream, "auto": side stream.
# dK / dV feed only side-stream work (weight gradients, key/value dgrad).  At hidden 768 the main stream is the longer one
# (41 ms against 23 in round 2) and the pass on the side stream saved ~4 ms.  At hidden 300 the side stream used to be the
# longer one (round 2: +1 ms/step with the pass there); since the dead-row schedule and the LDS-DMA kernel for its NT / NN
# products the main stream is (46 against 31 ms busy at batch 64): configs[1] 8.77 -> 8.60, configs[3] 50.1 -> 49.7 ms.
_DKV_SIDE_ENV = os.environ.get("BPMULT_DKV_SIDE", "auto")
_side_streams: Dict[Tuple[int, int, bool], "torch.cuda.Stream"] = {}
# Priority of the side stream: "low" (the dispatcher fills CUs from the main stream first), "normal", or "auto": low
# below hidden 512, normal from there on.  Measured on MI355X: at hidden 300 low wins (16.9 -> 16.6 ms/step); at hidden
# 768 every GEMM workgroup owns a CU for 50-300 us, the side stream's weight gradients are a third of the step's work,
# and starving them only lengthens the tail (37.97 ms/step low, 37.53 normal).
_SIDE_PRIORITY_ENV = os.environ.get("BPMULT_SIDE_PRIORITY", "auto")
# Low-rank key side for groups with a handful of query time steps (EncoderGroupPlan._lowrank; "0": dK / dV as everywhere else)
_LOWRANK = os.environ.get("BPMULT_LOWRANK", "1") != "0"
# (Measured in round 3 and removed: a side stream restricted to 160-224 CUs by hipExtStreamCreateWithCUMask, so that the
# main stream's row kernels never queue behind weight-gradient workgroups: 47-51 ms/step against 32.5.  Also without
# effect: d(LayerNorm output) written as bf16 by the data-gradient GEMMs and read as bf16 by the LayerNorm backward --
# 75 MB less per launch pair, 31.2 -> 31.3 ms/step.)


# Side streams that hold work the main stream has not waited for yet (handle -> stream).  Inside a graph capture every such
# fork must be joined back into the capturing stream before the capture ends -- an unjoined one invalidates the capture
# (and crashed inside hipStreamEndCapture in round 3 with a second stream pair).  _run() adds on SIDE / SIDE2 and clears on
# JOIN; the graph code checks that nothing is left (open_forks) before it lets a capture end.
_OPEN_FORKS: Dict[int, "torch.cuda.Stream"] = {}


def open_forks() -> List["torch.cuda.Stream"]:
    return list(_OPEN_FORKS.values())


def _side_stream(device, which: int = 1, low: bool = True) -> "torch.cuda.Stream":
    """Side streams per device, created through the C ABI (low: at the device's LOWEST priority)."""
    dev_i = device.index if device.index is not None else torch.cuda.current_device()
    if _SIDE_PRIORITY_ENV in ("low", "normal"):
        low = _SIDE_PRIORITY_ENV == "low"
    key = (dev_i, which, bool(low))
    if key not in _side_streams:
        from . import _lib
        out = C.c_void_p()
        with torch.cuda.device(dev_i):
            _lib.check(_lib.lib().bpm_stream_create(int(bool(low)), C.byref(out)), "bpm_stream_create")
        _side_streams[key] = torch.cuda.ExternalStream(out.value, device=torch.device("cuda", dev_i))
    return _side_streams[key]


class EncoderGroupPlan:
    """Buffers + launch tables for G encoders x L layers at batch size B."""

    def __init__(self, store: ParamStore, cfg: GroupCfg, encs: Sequence[EncoderDesc], B: int):
        self.store, self.cfg, self.encs, self.B = store, cfg, list(encs), B
        self.dtype = store.dtype
        d, H = cfg.d, cfg.H
        if d % H:
            raise ValueError("embed_dim must be divisible by num_heads")
        self.dh = d // H
        self.dhp = dhp_for(self.dh)
        # LOW-RANK KEY SIDE.  With T query time steps dK = dS^T Q and dV = Pd^T dO have rank T per (batch element, head), and
        # every key / value-side product of the backward factors through the [H T, S] matrices dS, Pd (written by the dQ
        # pass) instead of the [S B, d] matrices dK, dV:
        #   W_k' gradient  = Qexp^T (dS khat)           (was dK^T khat: d x d x S B -- now H T B x S x d, then d x d x H T B)
        #   d(khat)        = sum_layers dS^T (Qexp W_k') (was dK W_k' over K = layers d -- now K = layers H T)
        # and the same with Pd, dOexp, vhat, W_v' for the value side (Qexp / dOexp: the heads' vectors in their own column
        # block of otherwise-zero rows, bpm_expand_heads, so that all heads travel in one product).  Level 2 under dead-row
        # elimination has T = 2: at hidden 768 this removes 1.5 of the step's 19.4 ms (the key / value weight gradients over
        # 4096 rows, the merged K = 6144 data gradient, the dK / dV pass).  Equal to the dK / dV route in real arithmetic; the
        # roundings differ (dS instead of dK is rounded to CT), fixtures F7 / F9 / F11 hold both.  Crossmodal groups only.
        self._lowrank = (_LOWRANK and not cfg.biprojection and all(e.T * H * 4 <= d and e.S % 4 == 0 for e in self.encs)
                         and self.dh <= 128)
        self.ld, self.ld4 = pad32(d), pad32(4 * d)
        self.scale = self.dh ** -0.5
        dev, ct = store.device, ops.ct_torch(self.dtype)
        z = lambda *s, dt=torch.float32: torch.zeros(*s, device=dev, dtype=dt)
        L = cfg.layers
        self.buf: List[dict] = []
        # accumulators that every backward starts from zero (d(khat), d(vhat), folded bias sums): ONE buffer, one fill
        nacc = sum(L * 2 * d + 16 for e in self.encs)
        self._acc0 = torch.zeros(nacc, device=dev, dtype=torch.float32)
        acc_off = [0]

        def carve(*shape):
            n = 1
            for v in shape:
                n *= v
            t = self._acc0[acc_off[0]: acc_off[0] + n].view(*shape)
            acc_off[0] += (n + 15) // 16 * 16
            return t

        for e in self.encs:
            R, Rk = e.T * B, e.S * B
            tailp = bool(e.tail_rows)
            if tailp and (not cfg.biprojection or e.T < 2 or e.T_full is not None):
                raise ValueError("tail_rows: biprojection encoders with at least two query rows (crossmodal encoders gather their rows)")
            # query-side rows / time steps of every layer (the last one shrinks to rows {0, T-1} with tail_rows)
            Rl, Tl = [R] * L, [e.T] * L
            if tailp:
                Rl[-1], Tl[-1] = 2 * B, 2
            b = dict(R=R, Rk=Rk, Rl=Rl, Tl=Tl, tailp=tailp)
            b["x"] = [z(R, d) for _ in range(L)] + [z(Rl[-1], d)]
            b["ke"], b["ve"] = z(Rk, d), z(Rk, d)
            b["out"] = z(Tl[-1], B, d)
            b["stf"] = (z(Rl[-1]), z(Rl[-1]))
            # key / value source, normalised ONCE without affine (the per-layer LayerNorm gain and bias are folded
            # into the K / V projection weights, see register_encoder_shadows)
            b["khat"], b["vhat"] = z(Rk, self.ld, dt=ct), z(Rk, self.ld, dt=ct)
            b["stk"], b["stv"] = (z(Rk), z(Rk)), (z(Rk), z(Rk))
            # d(khat), d(vhat) = sum over the layers of dK_i Wk'_i, dV_i Wv'_i: ONE product over K = L ld per encoder at the
            # end of backward (dK_i / dV_i of every layer are kept side by side in dkall / dvall) instead of L products
            # accumulating into the same fp32 [Rk, d] tensor (8 x 300 MB of read-modify-write per level at hidden 768)
            b["Gk"], b["Gv"] = z(Rk, d), z(Rk, d)
            if self._lowrank:
                HT, Sp = H * e.T, (e.S + 63) // 64 * 64
                b["Sp"] = Sp
                # dS / Pd of every layer, [layer][h*T + t][b][key] with zero key padding (never written); the same row
                # order (h*T + t)*B + b for the expanded head rows and everything computed from them
                b["dSall"], b["Pdall"] = z(L, HT, B, Sp, dt=ct), z(L, HT, B, Sp, dt=ct)
                b["qkall"], b["daall"] = z(L, HT * B, self.ld, dt=ct), z(L, HT * B, self.ld, dt=ct)
                for nm in ("qexp", "doexp", "U", "Av"):
                    b[nm] = [z(HT * B, self.ld, dt=ct) for _ in range(2)]
                b["dkall"] = b["dvall"] = None
            else:
                b["dkall"], b["dvall"] = z(Rk, L * self.ld, dt=ct), z(Rk, L * self.ld, dt=ct)
            b["dWf"] = [z(2 * d, d) for _ in range(L)]              # folded K/V weight gradients (per backward)
            b["dbf"] = carve(L, 2 * d)                              # folded K/V bias gradients (column sums)
            # per-layer activations: shape(i) -- query-side tensors follow Rl / Tl, key / value-side ones stay full
            for nm, shape, dt in (("xn", lambda i: (R, self.ld), ct),
                                  ("qh", lambda i: (B, H, Tl[i], self.dhp), ct), ("kh", lambda i: (B, H, e.S, self.dhp), ct),
                                  ("vh", lambda i: (B, H, e.S, self.dhp), ct), ("ao", lambda i: (Rl[i], self.ld), ct),
                                  ("lse", lambda i: (B, H, Tl[i]), torch.float32),
                                  ("xmid", lambda i: (Rl[i], d), torch.float32), ("xn2", lambda i: (Rl[i], self.ld), ct),
                                  ("h1", lambda i: (Rl[i], self.ld4), ct),
                                  ("st0m", lambda i: (R,), torch.float32), ("st0r", lambda i: (R,), torch.float32),
                                  ("st1m", lambda i: (Rl[i],), torch.float32), ("st1r", lambda i: (Rl[i],), torch.float32)):
                b[nm] = [z(*shape(i), dt=dt) for i in range(L)]
            if cfg.biprojection:
                for nm, shape, dt in (("qs", lambda i: (B, H, Tl[i], self.dhp), ct), ("ks", lambda i: (B, H, e.T, self.dhp), ct),
                                      ("vs", lambda i: (B, H, e.T, self.dhp), ct), ("aos", lambda i: (Rl[i], self.ld), ct),
                                      ("lses", lambda i: (B, H, Tl[i]), torch.float32), ("xmid0", lambda i: (Rl[i], d), torch.float32),
                                      ("xq", lambda i: (Rl[i], self.ld), ct), ("st2m", lambda i: (Rl[i],), torch.float32),
                                      ("st2r", lambda i: (Rl[i],), torch.float32)):
                    b[nm] = [z(*shape(i), dt=dt) for i in range(L)]
            if tailp:
                # last layer: rows {0, T-1} of its input (fp32) and of LN0(input) (CT), their gradients, and the residual
                # gradient scattered back into an otherwise-zero [R, d] buffer (only the two row blocks are ever written)
                b["xg"], b["xng"] = z(2 * B, d), z(2 * B, self.ld, dt=ct)
                b["dxg"], b["dxng"], b["dxs"] = z(2 * B, d), z(2 * B, d), z(R, d)
            # backward temporaries (shared by all layers)
            b["dx"], b["dxn"] = z(R, d), z(R, d)
            # Off-critical-path work (weight gradients, the key/value-side dgrad + LayerNorm backward) runs on a
            # side stream up to two layers behind the main chain, so every operand it reads has its own buffer
            # within a layer (dyf: FFN, dy: attention, dy0/dqs/dks/dvs: biprojection self-attention half) and is
            # double-buffered by layer parity.
            two = lambda *shape: [z(*shape, dt=ct), z(*shape, dt=ct)]
            b["dy"], b["dh1"] = two(R, self.ld), two(R, self.ld4)
            # dyf[i % 3]: written one layer early (fused into the LayerNorm backward that produces dx)
            b["dyf"] = [z(R, self.ld, dt=ct) for _ in range(3)]
            b["dq"] = two(R, self.ld)
            if cfg.biprojection:
                b["dy0"] = two(R, self.ld)
                # dQ | dK | dV of the self-attention half side by side in one [R, 3 ld] buffer: without column padding
                # (ld == d) that is the [R, 3d] operand of ONE d(xn) = [dq dk dv] in_proj_weight product (K = 3d) instead
                # of three K = d launches accumulating into the same output
                # (one spare row: the bounded loaders size their descriptors rows x ld from each VIEW's first element, so
                # the dK / dV views' ranges reach up to 2 ld elements past row R - 1)
                b["dqkvs"] = two(R + 1, 3 * self.ld)
                b["dqs"], b["dks"], b["dvs"] = ([t[:R, w * self.ld:(w + 1) * self.ld] for t in b["dqkvs"]] for w in range(3))
            # read by the side-stream dK/dV pass of the cross attention: by layer parity like dq/dk/dv
            b["dao"] = [z(B, H, e.T, self.dhp, dt=ct) for _ in range(2)]
            b["delta"] = [z(B, H, e.T) for _ in range(2)]
            if cfg.biprojection:
                b["dao0"], b["delta0"] = z(B, H, e.T, self.dhp, dt=ct), z(B, H, e.T)
            b["dke"], b["dve"] = z(Rk, d), z(Rk, d)
            b["dxq"], b["dxk"], b["dxv"] = z(e.T, B, d), z(e.S, B, d), z(e.S, B, d)
            self.buf.append(b)
        if cfg.biprojection and any(e.T_full is not None for e in self.encs):
            raise ValueError("a gathered query subset is only exact for crossmodal (non-biprojection) encoders")
        self.table = sinusoid_table(max(max(e.T_full or e.T, e.S) for e in self.encs) + 2, d, dev)
        self._ones, self._zeros = torch.ones(d, device=dev), torch.zeros(d, device=dev)
        # table of the launch that turns folded K/V gradients into in_proj / LayerNorm parameter gradients
        lnK = 1 if cfg.biprojection else 0
        self._unfold = []                                  # one table per layer: its gradients are final with it
        for i in range(L):
            ud, blk = [], 0
            for e, b in zip(self.encs, self.buf):
                pn = lambda leaf: self._pn(e, i, leaf)
                u = UnfoldDesc()
                u.dWf, u.dbf = b["dWf"][i].data_ptr(), b["dbf"][i].data_ptr()
                u.W = store.p(pn("self_attn.in_proj_weight")).data_ptr() + 4 * d * d
                u.gamma, u.beta = store.p(pn(f"layer_norms.{lnK}.weight")).data_ptr(), store.p(pn(f"layer_norms.{lnK}.bias")).data_ptr()
                u.dW, u.dbias = store.gptr(pn("self_attn.in_proj_weight"), d * d), store.gptr(pn("self_attn.in_proj_bias"), d)
                u.dgamma, u.dbeta = store.gptr(pn(f"layer_norms.{lnK}.weight")), store.gptr(pn(f"layer_norms.{lnK}.bias"))
                u.rows, u.cols, u.ldw, u.blk0 = 2 * d, d, d, blk
                blk += (2 * d + 15) // 16
                ud.append(u)
            self._unfold.append((ops.device_table(ud), len(ud), blk))
        self._dkv_side = _DKV_SIDE_ENV if _DKV_SIDE_ENV != "auto" else "1"
        # a group whose query side is a handful of rows (level 2 under dead-row elimination) is bound by its SIDE stream
        # (key / value projections, their weight gradients): its dK / dV pass goes back to the main stream, which idles
        if _DKV_SIDE_ENV == "auto" and max(e.T for e in self.encs) * B <= 64:
            self._dkv_side = "0"
        self._side_low = d < 512
        self.store.side_low = self._side_low
        self._fwd = {True: self._build_fwd(True), False: self._build_fwd(False)}
        # backward tables by (training, stores): stores = the first weight-gradient launch of each large matrix writes
        # instead of accumulating (the flat gradient buffer was not cleared: ParamStore.begin_backward(stores=True))
        self._bwd = {(t, f): self._build_bwd(t, f) for t in (True, False) for f in (True, False)}

    # -- helpers ----------------------------------------------------------------
    def _mask_off(self, T: int, S: int) -> int:
        return 1 + abs(S - T) if self.cfg.attn_mask else 0

    def _pn(self, e: EncoderDesc, i: int, leaf: str) -> str:
        return f"{e.prefix}layers.{i}.{leaf}"

    def _gemm(self, variant, probs, background=False, presplit=()):
        # every operand of the encoder GEMMs is a CT buffer written by this library (LayerNorm / cast / epilogue /
        # attention outputs into zero-initialised padded rows, packed weight shadows): the k padding is zero
        # background: weight gradients only.  The side stream's K/V projections and K/V dgrads measured better WITH the
        # critical-path issue priority (16.79 -> 16.73 ms/step): the forward ones gate the next attention
        for p in probs:
            p.flags |= F_KPAD | (F_BACKGROUND if background else 0)
        arr = ops.array(GemmProblem, probs)
        arr.x3 = self.store.x3                       # bf16x3 mode: ops.gemm_grouped splits the operands of eligible launches
        # ... except operands whose split image an earlier launch of the same step has left: the forward activations a
        # weight gradient reads again, and gradients the main stream's data-gradient product of this layer has just split
        arr.x3_presplit = frozenset(t.data_ptr() for t in presplit)
        return (ops.gemm_grouped, self.dtype, variant, arr)

    # -- forward tables ---------------------------------------------------------
    def _build_fwd(self, training: bool):
        c, st, B, d, H = self.cfg, self.store, self.B, self.cfg.d, self.cfg.H
        ld, ld4, dh, dhp = self.ld, self.ld4, self.dh, self.dhp
        pr = (lambda p: p) if training else (lambda p: 0.0)
        A = ops.array
        hat = []
        for e, b in zip(self.encs, self.buf):
            hat += [ops.ln_problem(b["ke"], self._ones, self._zeros, b["stk"][0], b["stk"][1], b["Rk"], out=b["khat"], ldo=ld),
                    ops.ln_problem(b["ve"], self._ones, self._zeros, b["stv"][0], b["stv"][1], b["Rk"], out=b["vhat"], ldo=ld)]
        steps, kv_steps = [], [(SIDE, (ops.ln_fwd, self.dtype, A(LnProblem, hat), d)), (MARK, "hat")]
        for i in range(c.layers):
            ln, qkv, att, outp, ln2, fc1, fc2 = [], [], [], [], [], [], []
            kvp = []
            pre = dict(ln=[], gather=[], qkv=[], att=[], outp=[], cast=[])      # biprojection self-attention half
            for e, b in zip(self.encs, self.buf):
                R, Rk = b["R"], b["Rk"]
                Rq, Tq = b["Rl"][i], b["Tl"][i]                        # query-side rows / time steps of this layer
                tail = b["tailp"] and i == c.layers - 1                # rows {0, T-1} only (EncoderDesc.tail_rows)
                qpos = dict(q_pos0=0, q_stride=e.T - 1) if tail else dict(q_pos0=e.q_pos0, q_stride=e.q_stride)
                P = lambda leaf: st.p(self._pn(e, i, leaf))
                ipw = self._pn(e, i, "self_attn.in_proj_weight")
                ipb = P("self_attn.in_proj_bias")
                wo, w1, w2 = (self._pn(e, i, n) for n in ("self_attn.out_proj.weight", "fc1.weight", "fc2.weight"))
                g0, b0 = P("layer_norms.0.weight"), P("layer_norms.0.bias")
                g1, b1 = P("layer_norms.1.weight"), P("layer_norms.1.bias")

                def proj(Ain, rows, which, Cout, Tlen):
                    return ops.gemm_problem(Ain, st.sptr(ipw, which * d * ld), Cout, rows, d, d, ld, ld, 0,
                                            bias_n=ipb[which * d:(which + 1) * d], alpha=self.scale if which == 0 else 1.0,
                                            out_kind=OUT_HEADS, heads=(B, H, Tlen, dh, dhp))

                def proj_kv(Ain, which, Cout):       # folded: khat (W_k * gamma)^T + (W_k beta + b_k)
                    kvf = self._pn(e, i, KVF)
                    return ops.gemm_problem(Ain, st.sptr(kvf + (".k" if which == 1 else ".v")), Cout, Rk, d, d, ld, ld, 0,
                                            bias_n=st.fold(kvf, (which - 1) * d, d), out_kind=OUT_HEADS,
                                            heads=(B, H, e.S, dh, dhp))

                x_in = b["x"][i]
                if c.biprojection:
                    g2, b2 = P("layer_norms.2.weight"), P("layer_norms.2.bias")
                    pre["ln"].append(ops.ln_problem(x_in, g0, b0, b["st0m"][i], b["st0r"][i], R, out=b["xn"][i], ldo=ld))
                    xq_in, res_in = b["xn"][i], x_in                   # self-attention query operand, its residual
                    if tail:
                        # time steps 0 and T-1 are the first and the last B rows of a [T, B, .] tensor: two block copies
                        for j, r0 in ((0, 0), (1, R - B)):
                            pre["gather"].append(ops.cast_problem(x_in[r0:r0 + B], d, B, d, dst_f32=b["xg"][j * B:(j + 1) * B], ldf=d))
                            pre["gather"].append(ops.cast_problem(b["xn"][i][r0:r0 + B], ld, B, d, a_is_ct=True,
                                                                  dst_ct=b["xng"][j * B:(j + 1) * B], ldd=ld))
                        xq_in, res_in = b["xng"], b["xg"]
                    pre["qkv"] += [proj(xq_in, Rq, 0, b["qs"][i], Tq), proj(b["xn"][i], R, 1, b["ks"][i], e.T),
                                   proj(b["xn"][i], R, 2, b["vs"][i], e.T)]
                    pre["att"].append(ops.attn_problem(b["qs"][i], b["ks"][i], b["vs"][i], b["aos"][i], ld, b["lses"][i], B, H,
                                                       Tq, e.T, dh, dhp, self._mask_off(e.T, e.T),
                                                       drop_p=pr(e.attn_dropout), drop_site=site(e.enc_id, i, S_ATTN_SELF),
                                                       **(qpos if tail else {})))
                    pre["outp"].append(ops.gemm_problem(b["aos"][i], st.sptr(wo), b["xmid0"][i], Rq, d, d, ld, ld, d,
                                                        bias_n=P("self_attn.out_proj.bias"), resid=res_in, ldr=d,
                                                        drop_p=pr(c.res_dropout), drop_site=site(e.enc_id, i, S_RES0)))
                    pre["cast"].append(ops.cast_problem(b["xmid0"][i], d, Rq, d, dst_ct=b["xq"][i], ldd=ld))
                    q_src, resid_src = b["xq"][i], b["xmid0"][i]
                    gf, bf = g2, b2
                    stf = (b["st2m"][i], b["st2r"][i])
                else:
                    ln.append(ops.ln_problem(x_in, g0, b0, b["st0m"][i], b["st0r"][i], R, out=b["xn"][i], ldo=ld))
                    q_src, resid_src = b["xn"][i], x_in
                    gf, bf = g1, b1
                    stf = (b["st1m"][i], b["st1r"][i])
                qkv.append(proj(q_src, Rq, 0, b["qh"][i], Tq))
                kvp.append(proj_kv(b["khat"], 1, b["kh"][i]))
                kvp.append(proj_kv(b["vhat"], 2, b["vh"][i]))
                att.append(ops.attn_problem(b["qh"][i], b["kh"][i], b["vh"][i], b["ao"][i], ld, b["lse"][i], B, H, Tq, e.S, dh, dhp,
                                            self._mask_off(e.T_full or e.T, e.S), drop_p=pr(e.attn_dropout),
                                            drop_site=site(e.enc_id, i, S_ATTN), **qpos))
                outp.append(ops.gemm_problem(b["ao"][i], st.sptr(wo), b["xmid"][i], Rq, d, d, ld, ld, d,
                                             bias_n=P("self_attn.out_proj.bias"), resid=resid_src, ldr=d,
                                             drop_p=pr(c.res_dropout), drop_site=site(e.enc_id, i, S_RES1)))
                ln2.append(ops.ln_problem(b["xmid"][i], gf, bf, stf[0], stf[1], Rq, out=b["xn2"][i], ldo=ld))
                fc1.append(ops.gemm_problem(b["xn2"][i], st.sptr(w1), b["h1"][i], Rq, 4 * d, d, ld, ld, ld4, bias_n=P("fc1.bias"),
                                            flags=F_RELU, drop_p=pr(c.relu_dropout), drop_site=site(e.enc_id, i, S_RELU),
                                            out_kind=OUT_CT))
                fc2.append(ops.gemm_problem(b["h1"][i], st.sptr(w2), b["x"][i + 1], Rq, d, 4 * d, ld4, ld4, d, bias_n=P("fc2.bias"),
                                            resid=b["xmid"][i], ldr=d, drop_p=pr(c.res_dropout),
                                            drop_site=site(e.enc_id, i, S_RES2)))
            if c.biprojection:
                steps += [(ops.ln_fwd, self.dtype, A(LnProblem, pre["ln"]), d)] + \
                         ([(ops.rows_cast, self.dtype, A(CastProblem, pre["gather"]))] if pre["gather"] else []) + \
                         [self._gemm(GEMM_NT, pre["qkv"]),
                          (ops.attn_fwd, self.dtype, A(AttnProblem, pre["att"])),
                          self._gemm(GEMM_NT, pre["outp"]),
                          (ops.rows_cast, self.dtype, A(CastProblem, pre["cast"]))]
            # K/V side of every layer depends only on the (embedded) key/value sources: the side stream runs it
            # ahead of the query chain; the main stream waits for layer i's K/V heads just before attention i.
            kv_steps += [(SIDE, self._gemm(GEMM_NT, kvp)), (MARK, i)]
            if ln:
                steps.append((ops.ln_fwd, self.dtype, A(LnProblem, ln), d))
            steps += [self._gemm(GEMM_NT, qkv),
                      (WAIT, i),
                      (ops.attn_fwd, self.dtype, A(AttnProblem, att)),
                      self._gemm(GEMM_NT, outp)]
            steps += [(ops.ln_fwd, self.dtype, A(LnProblem, ln2), d),
                      self._gemm(GEMM_NT, fc1),
                      self._gemm(GEMM_NT, fc2)]
        fin = [ops.ln_problem(b["x"][c.layers], st.p(e.prefix + "layer_norm.weight"), st.p(e.prefix + "layer_norm.bias"),
                              b["stf"][0], b["stf"][1], b["Rl"][-1], out=b["out"], ldo=d, out_f32=True)
               for e, b in zip(self.encs, self.buf)]
        steps.append((ops.ln_fwd, self.dtype, A(LnProblem, fin), d))
        return kv_steps + steps + [JOIN]

    def _exec(self, s, seed: int) -> None:
        fn = s[0]
        if fn is ops.gemm_grouped:
            fn(s[1], s[2], s[3], seed)
        elif fn in (ops.attn_fwd, ops.attn_bwd, ops.attn_bwd_dq, ops.attn_bwd_dkv, ops.rows_cast):
            fn(s[1], s[2], seed)
        elif fn is ops.expand_heads:
            fn(s[1], s[2])
        elif fn is ops.ln_fwd:
            fn(s[1], s[2], s[3])
        elif fn is ops.ln_bwd:
            fn(s[1], s[2], self.dtype, seed)
        elif fn is ops.unfold_grads:
            fn(s[1], s[2], s[3], s[4])
        elif fn is ops.add_n:
            fn(s[1])
        else:
            raise RuntimeError("unknown step")

    def _run(self, steps, seed: int, on_mark=None) -> None:
        """Launch a step table.  Plain steps go to the current ("main") stream.  (SIDE, step) goes to the side
        stream, ordered behind everything the main stream has launched so far; (MARK, k) records an event on the
        side stream; (WAIT, k) makes the main stream wait for mark k (no-op if it was never recorded); JOIN
        makes the main stream wait for all side work.  With BPMULT_SIDE=0 everything runs on the main stream."""
        if not _SIDE:
            for s in steps:
                if s is not JOIN and s[0] is MARK and on_mark is not None:
                    ev = torch.cuda.Event()
                    ev.record(torch.cuda.current_stream())
                    on_mark(s[1], [ev])
                if s is JOIN or s[0] is MARK or s[0] is WAIT:
                    continue
                self._exec(s[1] if s[0] in (SIDE, SIDE2) else s, seed)
            return
        main = torch.cuda.current_stream()
        side = _side_stream(main.device, 1, self._side_low)
        marks: Dict[int, torch.cuda.Event] = {}
        main_dirty, side_dirty = True, False
        for s in steps:
            if s is JOIN:
                if side_dirty:
                    ev = torch.cuda.Event()
                    ev.record(side)
                    main.wait_event(ev)
                    side_dirty = False
                    _OPEN_FORKS.pop(side.cuda_stream, None)
            elif s[0] is SIDE2:                     # third stream: starts right behind the main stream's last launch
                side2 = _side_stream(main.device, 2, self._side_low)
                ev = torch.cuda.Event()
                ev.record(main)
                side2.wait_event(ev)
                _OPEN_FORKS[side2.cuda_stream] = side2
                with torch.cuda.stream(side2):
                    self._exec(s[1], seed)
                ev2 = torch.cuda.Event()
                ev2.record(side2)
                side.wait_event(ev2)                # the side stream's later steps consume its output: joined through it
                _OPEN_FORKS.pop(side2.cuda_stream, None)
                _OPEN_FORKS[side.cuda_stream] = side
                side_dirty = True
            elif s[0] is SIDE:
                if main_dirty:
                    ev = torch.cuda.Event()
                    ev.record(main)
                    side.wait_event(ev)
                    main_dirty = False
                _OPEN_FORKS[side.cuda_stream] = side
                with torch.cuda.stream(side):
                    self._exec(s[1], seed)
                side_dirty = True
            elif s[0] is MARK:
                if side_dirty:
                    marks[s[1]] = torch.cuda.Event()
                    marks[s[1]].record(side)
                if on_mark is not None:             # everything launched for this mark so far, on both streams
                    evm = torch.cuda.Event()
                    evm.record(main)
                    on_mark(s[1], [evm] + ([marks[s[1]]] if s[1] in marks else []))
            elif s[0] is WAIT:
                if s[1] in marks:
                    main.wait_event(marks.pop(s[1]))
            else:
                self._exec(s, seed)
                main_dirty = True

    def forward(self, xq: Sequence[torch.Tensor], xk: Sequence[torch.Tensor], xv: Sequence[torch.Tensor], seed: int,
                training: bool) -> List[torch.Tensor]:
        """xq[e]: fp32 [T_e,B,d]; xk[e], xv[e]: fp32 [S_e,B,d] key / value sources (the same tensor at every
        reference call site, mmtr.py:779-791; they still get independent embedding dropout, transformer.py:73-79)."""
        c, B, d = self.cfg, self.B, self.cfg.d
        p = c.embed_dropout if training else 0.0
        emb = []
        for e, b, q, k, v in zip(self.encs, self.buf, xq, xk, xv):
            for t, n in ((q, e.T), (k, e.S), (v, e.S)):
                if tuple(t.shape) != (n, B, d) or not t.is_contiguous() or t.dtype != torch.float32:
                    raise ValueError(f"encoder {e.prefix}: expected contiguous fp32 [{n},{B},{d}], got {tuple(t.shape)} {t.dtype}")
            emb += [ops.embed_problem(q, b["x"][0], e.T, B, drop_p=p, drop_site=site(e.enc_id, 0, S_EMB_Q), pos0=e.q_pos0,
                                      pos_stride=e.q_stride),
                    ops.embed_problem(k, b["ke"], e.S, B, drop_p=p, drop_site=site(e.enc_id, 0, S_EMB_K)),
                    ops.embed_problem(v, b["ve"], e.S, B, drop_p=p, drop_site=site(e.enc_id, 0, S_EMB_V))]
        ops.embed_pos_fwd(emb, self.table, d, math.sqrt(d), seed)
        self._last = (seed, training)
        self._run(self._fwd[training], seed)
        return [b["out"] for b in self.buf]

    # -- backward tables --------------------------------------------------------
    def _build_bwd(self, training: bool, stores: bool = False):
        c, st, B, d, H = self.cfg, self.store, self.B, self.cfg.d, self.cfg.H
        ACC1 = 0 if stores else F_ACCUM          # flags of the FIRST writer of a large weight gradient in a step
        ld, ld4, dh, dhp = self.ld, self.ld4, self.dh, self.dhp
        pr = (lambda p: p) if training else (lambda p: 0.0)
        A = ops.array
        G = len(self.encs)
        steps = []
        inv_relu = 1.0 / (1.0 - pr(c.relu_dropout))
        for i in reversed(range(c.layers)):
            wg_ffn, dg_fc2, dg_fc1, lnf = [], [], [], []
            wg_att, dg_out, att, dg_q, lnq = [], [], [], [], []
            s_cast0, s_dgout0, s_att0, s_wg0, s_dg0a, s_dg0b, s_dg0c, s_ln0 = [], [], [], [], [], [], [], []
            s_dgq, s_scatter = [], []                 # tail_rows (last layer): d(LN0 rows {0, T-1}) and the scatter back to [R, d]
            lr = self._lowrank
            lr_exp, lr_qk = [], []                    # low-rank key side: head expansion; Qexp W' and dS khat / Pd vhat products
            pre_ffn, pre_att = [], []                 # bf16x3: operands of the weight gradients whose split image already exists
            for e, b in zip(self.encs, self.buf):
                R, Rk = b["R"], b["Rk"]
                Rq, Tq = b["Rl"][i], b["Tl"][i]                        # query-side rows / time steps of this layer
                tail = b["tailp"] and i == c.layers - 1
                qpos = dict(q_pos0=0, q_stride=e.T - 1) if tail else dict(q_pos0=e.q_pos0, q_stride=e.q_stride)
                P = lambda leaf: st.p(self._pn(e, i, leaf))
                GP = lambda leaf, off=0: st.gptr(self._pn(e, i, leaf), off)
                ipw = self._pn(e, i, "self_attn.in_proj_weight")
                wo, w1, w2 = (self._pn(e, i, n) for n in ("self_attn.out_proj.weight", "fc1.weight", "fc2.weight"))
                lnF = 2 if c.biprojection else 1      # FFN LayerNorm index
                lnK = 1 if c.biprojection else 0      # key/value LayerNorm index
                stF = (b["st2m"][i], b["st2r"][i]) if c.biprojection else (b["st1m"][i], b["st1r"][i])
                # residual-stream gradient of this layer's query rows: the two gathered rows in a tail_rows last layer
                # (its LayerNorm-0 backward over all rows then writes the dense dx the layers below continue from)
                dx = b["dxg"] if tail else b["dx"]
                par = i & 1
                dh1, dy, dq, dao, delta = (b[n][par] for n in ("dh1", "dy", "dq", "dao", "delta"))
                ldk = c.layers * ld                       # layer i's dK / dV: column block i of dkall / dvall
                dk, dv = (None, None) if lr else (b["dkall"][:, i * ld:(i + 1) * ld], b["dvall"][:, i * ld:(i + 1) * ld])
                dyf = b["dyf"][i % 3]
                # hand-off to the next layer down (i-1): its FFN-output gradient dyf = dropmask(dx) and fc2.bias
                # gradient are produced by whichever LayerNorm backward finishes this layer's dx
                nxt = {} if i == 0 else dict(cast=b["dyf"][(i - 1) % 3], ldc=ld, cast_colsum=st.gptr(self._pn(e, i - 1, "fc2.bias")),
                                             drop_p=pr(c.res_dropout), drop_site=site(e.enc_id, i - 1, S_RES2))
                if c.biprojection:
                    dy0, dqs, dks, dvs = (b[n][par] for n in ("dy0", "dqs", "dks", "dvs"))
                pre_ffn += [dyf, dh1, b["h1"][i], b["xn2"][i]]
                pre_att += [dy, dq, b["ao"][i], b["xq"][i] if c.biprojection else b["xn"][i]] + ([] if lr else [b["khat"], b["vhat"]])
                # ---- FFN
                wg_ffn.append(ops.gemm_problem(dyf, b["h1"][i], GP("fc2.weight"), d, 4 * d, Rq, ld, ld4, 4 * d,
                                               flags=ACC1))
                dg_fc2.append(ops.gemm_problem(dyf, st.sptr(w2), dh1, Rq, 4 * d, d, ld, ld4, ld4, gate=b["h1"][i], ldg=ld4,
                                               gate_scale=inv_relu, colsum=GP("fc1.bias"), out_kind=OUT_CT))
                wg_ffn.append(ops.gemm_problem(dh1, b["xn2"][i], GP("fc1.weight"), 4 * d, d, Rq, ld4, ld, d,
                                               flags=ACC1))
                dg_fc1.append(ops.gemm_problem(dh1, st.sptr(w1), b["dxn"], Rq, d, 4 * d, ld4, ld, d))
                lnf.append(ops.ln_problem(b["xmid"][i], P(f"layer_norms.{lnF}.weight"), None, stF[0], stF[1], Rq, dy=b["dxn"], ldy=d,
                                          add=dx, dx=dx, dgamma=GP(f"layer_norms.{lnF}.weight"), dbeta=GP(f"layer_norms.{lnF}.bias"),
                                          cast=dy, ldc=ld, cast_colsum=GP("self_attn.out_proj.bias"), drop_p=pr(c.res_dropout),
                                          drop_site=site(e.enc_id, i, S_RES1)))
                # ---- (cross) attention block
                # (cross-attention half: the first writer of out_proj.weight / in_proj_weight rows [0, d) on the side stream; the
                # biprojection self-attention half below comes second and accumulates)
                wg_att.append(ops.gemm_problem(dy, b["ao"][i], GP("self_attn.out_proj.weight"), d, d, Rq, ld, ld, d,
                                               flags=ACC1))
                dg_out.append(ops.gemm_problem(dy, st.sptr(wo), dao, Rq, d, d, ld, ld, 0, out_kind=OUT_HEADS,
                                               heads=(B, H, Tq, dh, dhp)))
                lrx = {}
                if lr:
                    HT, Sp = H * Tq, b["Sp"]
                    rows = HT * B                        # row (h*T + t)*B + b everywhere below
                    lrx = dict(dS=b["dSall"][i], Pd=b["Pdall"][i], xs=(Sp, Tq * B * Sp, B * Sp))
                att.append(ops.attn_problem(b["qh"][i], b["kh"][i], b["vh"][i], b["ao"][i], ld, b["lse"][i], B, H, Tq, e.S, dh, dhp,
                                            self._mask_off(e.T_full or e.T, e.S), dO=dao, delta=delta, dQ=dq, lddq=ld,
                                            dK=dk, lddk=ldk, dV=dv, lddv=ldk, dq_scale=self.scale,
                                            drop_p=pr(e.attn_dropout), drop_site=site(e.enc_id, i, S_ATTN), **qpos, **lrx))
                ipb_g = self._pn(e, i, "self_attn.in_proj_bias")
                # query projection: gradients go straight to the parameters.  Key / value projections ran with the
                # LayerNorm folded in: their bias column sums and weight gradients (against khat / vhat) land in
                # per-layer scratch and are unfolded into in_proj / LayerNorm gradients by one launch at the end.
                # (the bias column sums ride on the weight-gradient GEMMs: colsum_a, one extra MFMA against ones)
                q_src = b["xq"][i] if c.biprojection else b["xn"][i]
                wg_att.append(ops.gemm_problem(dq, q_src, st.gptr(ipw, 0), d, d, Rq, ld, ld, d, flags=ACC1,
                                               colsum_a=st.gptr(ipb_g, 0)))
                if lr:
                    qexp, doexp, U, Av = (b[n][par] for n in ("qexp", "doexp", "U", "Av"))
                    # heads' q / dO vectors as block rows; the folded value-bias gradient = sum rowsum(Pd) dO (the folded key
                    # bias gets none: the rows of dS sum to zero -- dbf's key half stays at the zero every backward starts from)
                    lr_exp.append(ops.expand_problem(b["qh"][i], dao, qexp, doexp, B, H, Tq, dh, dhp, ld, Pd=b["Pdall"][i], S=Sp,
                                                     dbias=b["dbf"][i][d:]))
                    for src, stack, dst in ((qexp, KSTACK, b["qkall"][i]), (doexp, VSTACK, b["daall"][i])):
                        lr_qk.append(ops.gemm_problem(src, st.sptr(e.prefix + stack, i * ld * ld), dst, rows, d, d, ld, ld, ld,
                                                      out_kind=OUT_CT))
                    # per batch element: [H T, S] x [S, d]; the rows of a batch element are B rows apart in all three tensors
                    for mat, hat_, dst in ((b["dSall"][i], b["khat"], U), (b["Pdall"][i], b["vhat"], Av)):
                        lr_qk.append(ops.gemm_problem(mat, hat_, dst, HT, d, e.S, B * Sp, B * ld, B * ld, out_kind=OUT_CT,
                                                      flags=F_CT_NARROW, batch=(B, Sp, ld, ld)))
                    wg_att.append(ops.gemm_problem(qexp, U, b["dWf"][i][:d], d, d, rows, ld, ld, d))
                    wg_att.append(ops.gemm_problem(doexp, Av, b["dWf"][i][d:], d, d, rows, ld, ld, d))
                else:
                    wg_att.append(ops.gemm_problem(dk, b["khat"], b["dWf"][i][:d], d, d, Rk, ldk, ld, d, colsum_a=b["dbf"][i][:d]))
                    wg_att.append(ops.gemm_problem(dv, b["vhat"], b["dWf"][i][d:], d, d, Rk, ldk, ld, d, colsum_a=b["dbf"][i][d:]))
                if c.biprojection:   # query was not normalised: its gradient joins the residual stream directly
                    dg_q.append(ops.gemm_problem(dq, st.sptr(ipw, 0), dx, Rq, d, d, ld, ld, d, flags=F_ACCUM))
                else:
                    dg_q.append(ops.gemm_problem(dq, st.sptr(ipw, 0), b["dxn"], Rq, d, d, ld, ld, d))
                    lnq.append(ops.ln_problem(b["x"][i], P("layer_norms.0.weight"), None, b["st0m"][i], b["st0r"][i], Rq, dy=b["dxn"],
                                              ldy=d, add=dx, dx=dx, dgamma=GP("layer_norms.0.weight"), dbeta=GP("layer_norms.0.bias"),
                                              **nxt))
                if c.biprojection:
                    # ---- self-attention half (same attention parameters)
                    s_cast0.append(ops.cast_problem(dx, d, Rq, d, dst_ct=dy0, ldd=ld, colsum=GP("self_attn.out_proj.bias"),
                                                    drop_p=pr(c.res_dropout), drop_site=site(e.enc_id, i, S_RES0)))
                    s_wg0.append(ops.gemm_problem(dy0, b["aos"][i], GP("self_attn.out_proj.weight"), d, d, Rq, ld, ld, d,
                                                  flags=F_ACCUM))
                    s_dgout0.append(ops.gemm_problem(dy0, st.sptr(wo), b["dao0"], Rq, d, d, ld, ld, 0, out_kind=OUT_HEADS,
                                                     heads=(B, H, Tq, dh, dhp)))
                    s_att0.append(ops.attn_problem(b["qs"][i], b["ks"][i], b["vs"][i], b["aos"][i], ld, b["lses"][i], B, H, Tq, e.T,
                                                   dh, dhp, self._mask_off(e.T, e.T), dO=b["dao0"], delta=b["delta0"], dQ=dqs,
                                                   lddq=3 * ld, dK=dks, lddk=3 * ld, dV=dvs, lddv=3 * ld, dq_scale=self.scale,
                                                   drop_p=pr(e.attn_dropout), drop_site=site(e.enc_id, i, S_ATTN_SELF),
                                                   **(qpos if tail else {})))
                    # in_proj_weight rows [d, 3d): this launch is their first writer (unfold_grads comes after it and adds)
                    xq_in = b["xng"] if tail else b["xn"][i]
                    for w, src, xin, rows in ((0, dqs, xq_in, Rq), (1, dks, b["xn"][i], R), (2, dvs, b["xn"][i], R)):
                        s_wg0.append(ops.gemm_problem(src, xin, st.gptr(ipw, w * d * d), d, d, rows, 3 * ld, ld, d,
                                                      flags=F_ACCUM if w == 0 else ACC1, colsum_a=st.gptr(ipb_g, w * d)))
                    # d(xn) = dq Wq + dk Wk + dv Wv: three launches (plain store, then two +=) -- one owner per
                    # output tile in each launch, no atomics (per-lane-scattered float atomics run ~17x below store rate)
                    if tail:
                        # keys / values come from every row, the query only from rows {0, T-1}: d(xn) over all rows is the
                        # K / V part; the query part is a [2B, d] product whose row blocks are added into it, and the
                        # gathered rows' residual gradient goes to the same two row blocks of the otherwise-zero dxs
                        if ld == d:
                            s_dg0a.append(ops.gemm_problem(b["dqkvs"][par][:R, ld:], st.sptr(ipw, d * ld), b["dxn"], R, d, 2 * d,
                                                           3 * ld, ld, d))
                        else:
                            s_dg0a.append(ops.gemm_problem(dks, st.sptr(ipw, d * ld), b["dxn"], R, d, d, 3 * ld, ld, d))
                            s_dg0b.append(ops.gemm_problem(dvs, st.sptr(ipw, 2 * d * ld), b["dxn"], R, d, d, 3 * ld, ld, d, flags=F_ACCUM))
                        s_dgq.append(ops.gemm_problem(dqs, st.sptr(ipw, 0), b["dxng"], Rq, d, d, 3 * ld, ld, d))
                        for j, r0 in ((0, 0), (1, R - B)):
                            blk = slice(j * B, (j + 1) * B)
                            s_scatter.append(ops.addn_problem(b["dxn"][r0:r0 + B], [b["dxn"][r0:r0 + B], b["dxng"][blk]]))
                            s_scatter.append(ops.addn_problem(b["dxs"][r0:r0 + B], [b["dxg"][blk]]))
                    elif ld == d:                                   # one product over K = 3d (see the dqkvs buffer)
                        s_dg0a.append(ops.gemm_problem(b["dqkvs"][par], st.sptr(ipw, 0), b["dxn"], R, d, 3 * d, 3 * ld, ld, d))
                    else:
                        s_dg0a.append(ops.gemm_problem(dqs, st.sptr(ipw, 0), b["dxn"], R, d, d, 3 * ld, ld, d))
                        s_dg0b.append(ops.gemm_problem(dks, st.sptr(ipw, d * ld), b["dxn"], R, d, d, 3 * ld, ld, d, flags=F_ACCUM))
                        s_dg0c.append(ops.gemm_problem(dvs, st.sptr(ipw, 2 * d * ld), b["dxn"], R, d, d, 3 * ld, ld, d, flags=F_ACCUM))
                    s_ln0.append(ops.ln_problem(b["x"][i], P("layer_norms.0.weight"), None, b["st0m"][i], b["st0r"][i], R,
                                                dy=b["dxn"], ldy=d, add=b["dxs"] if tail else dx, dx=b["dx"],
                                                dgamma=GP("layer_norms.0.weight"), dbeta=GP("layer_norms.0.bias"), **nxt))
            for group in (lnf, lnq, s_ln0):               # each LayerNorm-backward launch owns its gradient rows exactly once
                ops.check_ln_rows(group, d)
            # Ownership of parameter-gradient words (bpm_ln_bwd_ws adds its row sums with a plain read-modify-write):
            # layer_norms.* / out_proj.bias / fc2.bias gradients are written by the MAIN stream's LayerNorm backward
            # launches only, except layer_norms.{lnK} which unfold_grads (side stream) also adds to -- that launch is
            # ordered behind lnq(i) / s_ln0(i) by the main_dirty event recorded before every SIDE step, and the next
            # main-stream writer of the same words is the NEXT step's backward (behind the JOIN).  Keep it that way.
            # Side stream (SIDE): weight gradients and the key/value-side dgrad + LayerNorm backward -- nothing on
            # the backward critical path consumes them.  Temporaries are double-buffered by layer parity, so the
            # main chain only waits (WAIT) for the side work of two layers ago before overwriting them.
            x3 = st.x3
            wg_ffn_step = (SIDE, self._gemm(GEMM_TN, wg_ffn, background=True, presplit=pre_ffn if x3 else ()))
            wg_att_step = (SIDE, self._gemm(GEMM_TN, wg_att, background=True, presplit=pre_att if x3 else ()))
            # (bf16x3: the weight gradients are launched BEHIND the data-gradient products that split the same gradients --
            # dg_fc1 splits dh1, dg_q splits dq -- so that the side stream finds those images instead of splitting again)
            steps += [(WAIT, i + 2),
                      self._gemm(GEMM_NN, dg_fc2)] + ([] if x3 else [wg_ffn_step]) + \
                     [self._gemm(GEMM_NN, dg_fc1)] + ([wg_ffn_step] if x3 else []) + \
                     [(ops.ln_bwd, A(LnProblem, lnf), d),
                      self._gemm(GEMM_NN, dg_out),
                      (ops.attn_bwd_dq, self.dtype, A(AttnProblem, att)),
                      # dK / dV feed only side work: beside the main chain where the side stream has slack (see _DKV_SIDE_ENV)
                      ((SIDE if self._dkv_side == "1" else SIDE2, (ops.attn_bwd_dkv, self.dtype, A(AttnProblem, att)))
                       if self._dkv_side in ("1", "2") else (ops.attn_bwd_dkv, self.dtype, A(AttnProblem, att)))][:3 if lr else 4] + \
                     ([] if x3 or lr else [wg_att_step]) + [self._gemm(GEMM_NN, dg_q)] + ([wg_att_step] if x3 and not lr else [])
            if lnq:
                steps.append((ops.ln_bwd, A(LnProblem, lnq), d))
            if lr:
                # low-rank key side: no dK / dV pass; the side stream continues from the dS / Pd the dQ pass wrote.  Issued
                # BEHIND the rest of the layer's main chain
                steps += [(SIDE, (ops.expand_heads, self.dtype, A(ExpandProblem, lr_exp))), (SIDE, self._gemm(GEMM_NN, lr_qk)),
                          wg_att_step]
            if c.biprojection:
                steps += [(ops.rows_cast, self.dtype, A(CastProblem, s_cast0)),
                          self._gemm(GEMM_NN, s_dgout0),
                          (ops.attn_bwd, self.dtype, A(AttnProblem, s_att0)),
                          (SIDE, self._gemm(GEMM_TN, s_wg0, background=True)),
                          self._gemm(GEMM_NN, s_dg0a)] + \
                         ([self._gemm(GEMM_NN, s_dg0b)] if s_dg0b else []) + ([self._gemm(GEMM_NN, s_dg0c)] if s_dg0c else []) + \
                         ([self._gemm(GEMM_NN, s_dgq), (ops.add_n, A(AddnProblem, s_scatter))] if s_dgq else []) + \
                         [(ops.ln_bwd, A(LnProblem, s_ln0), d)]
            # folded K/V gradients of this layer -> in_proj / LayerNorm parameter gradients; with it every gradient of
            # layer i is final once the side stream reaches MARK i and the main stream this point (all-reduce hook)
            steps += [(SIDE, (ops.unfold_grads,) + self._unfold[i] + (stores and not c.biprojection,)), (MARK, i)]
        # d(khat), d(vhat): all layers' dK / dV against the stacked projection weights, one product over K = L ld per
        # encoder; then -> d(embedded key / value source): LayerNorm backward without affine
        hat, dg_kv = [], []
        for e, b in zip(self.encs, self.buf):
            if self._lowrank:        # d(khat) of batch element bb = dS_all[:, bb]^T (Qexp W_k')_all[:, bb], K = layers H T
                KK, Sp = c.layers * H * e.T, b["Sp"]
                for mat, prod, G_ in ((b["dSall"], b["qkall"], b["Gk"]), (b["Pdall"], b["daall"], b["Gv"])):
                    dg_kv.append(ops.gemm_problem(mat, prod, G_, e.S, d, KK, B * Sp, B * ld, B * d, batch=(B, Sp, ld, d)))
            else:
                dg_kv += [ops.gemm_problem(b["dkall"], st.sptr(e.prefix + KSTACK), b["Gk"], b["Rk"], d, c.layers * ld, c.layers * ld, ld, d),
                          ops.gemm_problem(b["dvall"], st.sptr(e.prefix + VSTACK), b["Gv"], b["Rk"], d, c.layers * ld, c.layers * ld, ld, d)]
            hat += [ops.ln_problem(b["ke"], self._ones, None, b["stk"][0], b["stk"][1], b["Rk"], dy=b["Gk"], ldy=d, dx=b["dke"]),
                    ops.ln_problem(b["ve"], self._ones, None, b["stv"][0], b["stv"][1], b["Rk"], dy=b["Gv"], ldy=d, dx=b["dve"])]
        steps += [(SIDE, self._gemm(GEMM_TN if self._lowrank else GEMM_NN, dg_kv)), (SIDE, (ops.ln_bwd, A(LnProblem, hat), d)), JOIN]
        return steps

    @staticmethod
    def store_written(prefix: str, layers: int):
        """Names of the parameters whose gradient the `stores` tables write with a plain store (ParamStore.set_store_written)."""
        return [f"{prefix}layers.{i}.{leaf}" for i in range(layers)
                for leaf in ("self_attn.in_proj_weight", "self_attn.out_proj.weight", "fc1.weight", "fc2.weight")]

    def backward(self, douts: Sequence[Optional[torch.Tensor]], on_layer=None, stores: bool = False):
        """douts[e]: fp32 [T_e,B,d] gradient of encoder e's output (None = zero).  Returns the plan-owned
        gradients w.r.t. each encoder's query, key and value sources (three lists), and accumulates
        parameter gradients into the ParamStore's flat gradient buffer (stores=True: the large weight gradients are
        WRITTEN by their first launch -- the buffer was not cleared, ParamStore.begin_backward(stores=True))."""
        seed, training = self._last
        c, st, B, d = self.cfg, self.store, self.B, self.cfg.d
        fin, keep = [], []
        top = c.layers - 1
        self._acc0.zero_()
        for e, b, g in zip(self.encs, self.buf, douts):
            dx_top = b["dxg"] if b["tailp"] else b["dx"]          # gradient of the top layer's output rows
            if g is None:
                dx_top.zero_()
                b["dyf"][top % 3].zero_()
                continue
            if tuple(g.shape) != (b["Tl"][-1], B, d) or g.dtype != torch.float32:
                raise ValueError(f"encoder {e.prefix}: output gradient must be fp32 [{b['Tl'][-1]},{B},{d}], got {tuple(g.shape)} {g.dtype}")
            g = g.contiguous()
            keep.append(g)
            fin.append(ops.ln_problem(b["x"][c.layers], st.p(e.prefix + "layer_norm.weight"), None, b["stf"][0], b["stf"][1], b["Rl"][-1],
                                      dy=g, ldy=d, dx=dx_top, dgamma=st.gptr(e.prefix + "layer_norm.weight"),
                                      dbeta=st.gptr(e.prefix + "layer_norm.bias"),
                                      cast=b["dyf"][top % 3], ldc=self.ld, cast_colsum=st.gptr(self._pn(e, top, "fc2.bias")),
                                      drop_p=c.res_dropout if training else 0.0, drop_site=site(e.enc_id, top, S_RES2)))
        if fin:
            ops.ln_bwd(fin, d, self.dtype, seed)
        # on_layer(i, events): every parameter gradient of layer i (and, with the first call, of the final LayerNorm)
        # is complete once `events` have passed -- the data-parallel exchange starts there (distributed.GradSync)
        self._run(self._bwd[(training, stores)], seed, on_layer)
        p = c.embed_dropout if training else 0.0
        emb = []
        for e, b in zip(self.encs, self.buf):
            emb.append(ops.embed_problem(b["dx"], b["dxq"], e.T, B, drop_p=p, drop_site=site(e.enc_id, 0, S_EMB_Q)))
            emb.append(ops.embed_problem(b["dke"], b["dxk"], e.S, B, drop_p=p, drop_site=site(e.enc_id, 0, S_EMB_K)))
            emb.append(ops.embed_problem(b["dve"], b["dxv"], e.S, B, drop_p=p, drop_site=site(e.enc_id, 0, S_EMB_V)))
        ops.embed_pos_bwd(emb, d, math.sqrt(d), seed)
        return [b["dxq"] for b in self.buf], [b["dxk"] for b in self.buf], [b["dxv"] for b in self.buf]


KVF = "self_attn.in_proj_weight#kvf"      # key suffix of the folded key/value shadow and bias


KSTACK, VSTACK = "#kstack", "#vstack"     # per-encoder stacks of the folded key / value projection weights


def register_encoder_shadows(store: ParamStore, prefix: str, d: int, layers: int, biprojection: bool = False) -> None:
    lnK = 1 if biprojection else 0        # LayerNorm applied to the key / value source (transformer.py:167-172)
    ld = pad32(d)
    for i in range(layers):
        p = f"{prefix}layers.{i}."
        store.add_shadow(p + "self_attn.in_proj_weight", p + "self_attn.in_proj_weight", 3 * d, d)
        # key / value projections with that LayerNorm folded in: W' = W * gamma (columns), b' = W beta + b.  The layers' W'
        # are STACKED per encoder (block i = rows [i ld, i ld + d) of a [layers ld, ld] shadow, pad rows zero): layer i's
        # projection reads its block, and the key / value-side data gradient of ALL layers is one product over K = layers ld
        if i == 0:
            store.add_blank_shadow(prefix + KSTACK, layers * ld, ld)
            store.add_blank_shadow(prefix + VSTACK, layers * ld, ld)
        store.add_shadow(p + KVF + ".k", p + "self_attn.in_proj_weight", d, d, src_row0=d, colscale=p + f"layer_norms.{lnK}.weight",
                         base_key=prefix + KSTACK, dst_row0=i * ld)
        store.add_shadow(p + KVF + ".v", p + "self_attn.in_proj_weight", d, d, src_row0=2 * d, colscale=p + f"layer_norms.{lnK}.weight",
                         base_key=prefix + VSTACK, dst_row0=i * ld)
        store.add_fold(p + KVF, p + "self_attn.in_proj_weight", d, 2 * d, d, p + f"layer_norms.{lnK}.bias",
                       p + "self_attn.in_proj_bias", d)
        store.add_shadow(p + "self_attn.out_proj.weight", p + "self_attn.out_proj.weight", d, d)
        store.add_shadow(p + "fc1.weight", p + "fc1.weight", 4 * d, d)
        store.add_shadow(p + "fc2.weight", p + "fc2.weight", d, 4 * d)
