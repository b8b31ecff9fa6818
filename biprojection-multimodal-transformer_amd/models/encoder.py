"""`TransformerEncoder` with the reference's constructor, forward signature and
state_dict layout (bpmult/models/transformer.py:9-99), executed by the grouped
HIP engine.  The sub-modules below are parameter containers that reproduce the
reference's parameter names; none of them has a PyTorch forward.
"""
from __future__ import annotations

import math
from typing import Optional

import torch
from torch import nn

from .. import config
from ..engine import EncoderDesc, EncoderGroupPlan, GroupCfg, ParamStore, register_encoder_shadows


class MultiheadAttention(nn.Module):
    """Packed in-projection attention parameters (multihead_attention.py:17-50)."""

    def __init__(self, embed_dim: int, num_heads: int, attn_dropout: float = 0.0):
        super().__init__()
        self.embed_dim, self.num_heads, self.attn_dropout = embed_dim, num_heads, attn_dropout
        self.head_dim = embed_dim // num_heads
        if self.head_dim * num_heads != embed_dim:
            raise ValueError("embed_dim must be divisible by num_heads")
        self.in_proj_weight = nn.Parameter(torch.empty(3 * embed_dim, embed_dim))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * embed_dim))
        self.out_proj = nn.Linear(embed_dim, embed_dim, bias=True)
        nn.init.xavier_uniform_(self.in_proj_weight)          # fan computed on the packed shape (:41-46)
        nn.init.xavier_uniform_(self.out_proj.weight)
        nn.init.constant_(self.out_proj.bias, 0.0)


def _xavier_linear(i: int, o: int) -> nn.Linear:
    m = nn.Linear(i, o)
    nn.init.xavier_uniform_(m.weight)                         # transformer.py:219-224
    nn.init.constant_(m.bias, 0.0)
    return m


class TransformerEncoderLayer(nn.Module):
    def __init__(self, embed_dim: int, num_heads: int, attn_dropout: float, biprojection: bool):
        super().__init__()
        self.self_attn = MultiheadAttention(embed_dim, num_heads, attn_dropout)
        self.fc1 = _xavier_linear(embed_dim, 4 * embed_dim)
        self.fc2 = _xavier_linear(4 * embed_dim, embed_dim)
        self.layer_norms = nn.ModuleList([nn.LayerNorm(embed_dim) for _ in range(3 if biprojection else 2)])


class SinusoidalPositionalEmbedding(nn.Module):
    """Only the reference's buffer (position_embedding.py:42); the table itself is
    built by engine.sinusoid_table and gathered inside the embedding kernel."""

    def __init__(self, embedding_dim: int):
        super().__init__()
        self.embedding_dim = embedding_dim
        self.register_buffer("_float_tensor", torch.zeros(1))


class _EncoderFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, anchor, x_q, x_k, x_v, enc):
        plan = enc._plan_for(x_q, x_k)
        enc._store.refresh_shadows()
        if enc._store.x3:
            from .. import ops
            ops.x3_new_step()
        out = plan.forward([x_q.detach().contiguous()], [x_k.detach().contiguous()], [x_v.detach().contiguous()],
                           enc._next_seed(), enc.training)[0]
        ctx.enc, ctx.plan = enc, plan
        ctx.same_kv = x_k is x_v
        return out.detach().clone()

    @staticmethod
    def backward(ctx, dout):
        enc, plan = ctx.enc, ctx.plan
        enc._store.begin_backward()
        dq, dk, dv = plan.backward([dout])
        enc._store.end_backward()
        return None, dq[0].clone(), dk[0].clone(), dv[0].clone(), None


class TransformerEncoder(nn.Module):
    """Drop-in for bpmult.models.transformer.TransformerEncoder.

    forward(x_in, x_in_k, x_in_v) on [T,B,d] / [S,B,d] fp32 CUDA tensors.  The
    self-attention-only call forward(x_in) of the reference (used only by its
    broken `hybrid` branch) is not part of the hot path."""

    def __init__(self, embed_dim, num_heads, layers, attn_dropout=0.0, relu_dropout=0.0, res_dropout=0.0,
                 embed_dropout=0.0, attn_mask=False, biprojection=False):
        super().__init__()
        self.dropout = embed_dropout
        self.attn_dropout, self.relu_dropout, self.res_dropout = attn_dropout, relu_dropout, res_dropout
        self.embed_dim, self.num_heads = embed_dim, num_heads
        self.embed_scale = math.sqrt(embed_dim)
        self.embed_positions = SinusoidalPositionalEmbedding(embed_dim)
        self.attn_mask, self.biprojection = attn_mask, biprojection
        self.layers = nn.ModuleList([TransformerEncoderLayer(embed_dim, num_heads, attn_dropout, biprojection)
                                     for _ in range(layers)])
        self.register_buffer("version", torch.Tensor([2]))
        self.normalize = True
        self.layer_norm = nn.LayerNorm(embed_dim)
        self.precision: Optional[str] = None          # None -> config.precision() at first use
        self._store: Optional[ParamStore] = None
        self._plans = {}
        self._step = 0

    def group_cfg(self) -> GroupCfg:
        return GroupCfg(self.embed_dim, self.num_heads, len(self.layers), self.relu_dropout, self.res_dropout, self.dropout,
                        self.attn_mask, self.biprojection)

    # -- standalone execution (a group of one) ----------------------------------
    def _apply(self, fn, *a, **k):
        r = super()._apply(fn, *a, **k)
        self._store, self._plans = None, {}
        return r

    def _ensure_store(self) -> ParamStore:
        if self._store is None or not self._store.still_flat():
            prec = self.precision or config.precision()
            self._store = ParamStore(list(self.named_parameters()), config.dtype_code(prec), x3=config.is_x3(prec))
            register_encoder_shadows(self._store, "", self.embed_dim, len(self.layers), biprojection=self.biprojection)
            self._store.finalize_shadows()
            self._plans = {}
            self._anchor = torch.zeros(1, device=self._store.device, requires_grad=True)
        return self._store

    def _plan_for(self, x_q, x_k) -> EncoderGroupPlan:
        st = self._ensure_store()
        key = (x_q.shape[0], x_k.shape[0], x_q.shape[1])
        if key not in self._plans:
            desc = EncoderDesc("", 0, x_q.shape[0], x_k.shape[0], self.attn_dropout)
            self._plans[key] = EncoderGroupPlan(st, self.group_cfg(), [desc], x_q.shape[1])
        return self._plans[key]

    def _next_seed(self) -> int:
        self._step += 1
        return (torch.initial_seed() * 1000003 + self._step) & 0x7FFFFFFFFFFFFFFF      # 63 bits: bit 63 marks a device-resident seed

    def forward(self, x_in, x_in_k=None, x_in_v=None):
        if x_in_k is None or x_in_v is None:
            raise NotImplementedError("self-attention-only TransformerEncoder.forward(x) is outside the BPMulT hot path "
                                      "(reference uses it only in the broken --hybrid branch)")
        self._ensure_store()
        return _EncoderFn.apply(self._anchor, x_in, x_in_k, x_in_v, self)
