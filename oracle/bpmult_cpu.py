"""CPU oracle for the BPMulT forward/backward hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import this file, and there only as the checker / CPU
baseline -- never as the thing shipped or measured as "GPU".  The product path
(``biprojection-multimodal-transformer_amd``) raises if its HIP library is
missing; it never falls back to this code.

What it is: a functional (state_dict in, tensors out) restatement in plain
``torch`` CPU fp32 ops of the arithmetic of

  * bpmult/models/position_embedding.py   (sinusoid table, positions)
  * bpmult/models/multihead_attention.py  (packed in-proj MHA)
  * bpmult/models/transformer.py          (pre-LN encoder, crossmodal /
                                           biprojection / self variants, mask)
  * bpmult/models/mmtr.py                 (GMU family, 3- and 4-modal graphs)

written from SURVEY.md Appendix A.  Every function cites the reference
file:line it follows.  Gradients come from torch autograd over these ops.

Parity pin: ``tests/golden/*.npz`` were produced by importing the real
reference from /root/reference (tests/golden/make_golden.py, run in the build
container only) and ``tests/test_oracle_golden.py`` checks this file against
every one of them (<= 1e-5).  The reference repository has no tests, golden
vectors or known-answer fixtures of its own.

State-dict keys are the reference's own, so a
reference checkpoint can be fed to these functions unchanged.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
SD = Dict[str, Tensor]


# ----------------------------------------------------------------------------
# positional embedding  (position_embedding.py:8-27, 44-76)
# ----------------------------------------------------------------------------
def sinusoid_table(n_pos: int, d: int) -> Tensor:
    """Row p = [sin(p f_i) | cos(p f_i)], f_i = exp(-i ln(1e4)/(d//2 - 1));
    odd d gets a trailing zero column; row 0 (padding) is zero.
    position_embedding.py:44-60."""
    half = d // 2
    step = math.log(10000.0) / (half - 1)
    freq = torch.exp(torch.arange(half, dtype=torch.float32) * -step)
    ang = torch.arange(n_pos, dtype=torch.float32)[:, None] * freq[None, :]
    tab = torch.cat([ang.sin(), ang.cos()], dim=1)
    if d % 2 == 1:
        tab = torch.cat([tab, torch.zeros(n_pos, 1)], dim=1)
    tab[0].zero_()
    return tab


def positions_from_channel0(x_tbd: Tensor) -> Tensor:
    """pos[t,b] = t+1 where x[t,b,0] != 0 else 0 (padding row).
    position_embedding.py:8-27 applied to x.transpose(0,1)[:,:,0]
    (transformer.py:68); left_pad = 0."""
    T = x_tbd.shape[0]
    ar = torch.arange(1, T + 1, dtype=torch.long)[:, None]
    return torch.where(x_tbd[:, :, 0] != 0, ar, torch.zeros_like(ar))


def pos_embedding(x_tbd: Tensor) -> Tensor:
    """[T,B,d] positional term, detached.  position_embedding.py:62-76."""
    T, B, d = x_tbd.shape
    tab = sinusoid_table(T + 1, d)
    pos = positions_from_channel0(x_tbd)
    return tab.index_select(0, pos.reshape(-1)).view(T, B, d).detach()


# ----------------------------------------------------------------------------
# mask  (transformer.py:204-216)
# ----------------------------------------------------------------------------
def future_mask(T: int, S: int) -> Tensor:
    """Additive [T,S] mask: -inf where j - i >= 1 + |S - T|, else 0."""
    i = torch.arange(T)[:, None]
    j = torch.arange(S)[None, :]
    m = torch.zeros(T, S)
    m[(j - i) >= 1 + abs(S - T)] = float("-inf")
    return m


# ----------------------------------------------------------------------------
# multi-head attention  (multihead_attention.py:52-158)
# ----------------------------------------------------------------------------
def mha(sd: SD, pfx: str, q_in: Tensor, k_in: Tensor, v_in: Tensor, H: int,
        mask: Optional[Tensor], p_attn: float = 0.0, training: bool = False) -> Tensor:
    """q_in [T,B,d], k_in/v_in [S,B,d] -> [T,B,d].
    Packed in_proj rows [0:d]=Q, [d:2d]=K, [2d:3d]=V (:152-158); q scaled by
    dh^-0.5 after the bias (:86); head h = channels [h*dh,(h+1)*dh), batch-head
    index b*H+h (:95-99); fp32 softmax (:121); dropout on the weights (:124);
    out_proj (:130).  The head-averaged weights (:133-134) are discarded by
    every caller and not computed."""
    W, b = sd[pfx + "in_proj_weight"], sd[pfx + "in_proj_bias"]
    T, B, d = q_in.shape
    S = k_in.shape[0]
    dh = d // H
    q = F.linear(q_in, W[:d], b[:d]) * dh ** -0.5
    k = F.linear(k_in, W[d:2 * d], b[d:2 * d])
    v = F.linear(v_in, W[2 * d:], b[2 * d:])
    q = q.reshape(T, B * H, dh).transpose(0, 1)
    k = k.reshape(S, B * H, dh).transpose(0, 1)
    v = v.reshape(S, B * H, dh).transpose(0, 1)
    s = torch.bmm(q, k.transpose(1, 2))
    if mask is not None:
        s = s + mask.unsqueeze(0)
    a = F.softmax(s.float(), dim=-1)
    a = F.dropout(a, p=p_attn, training=training)
    o = torch.bmm(a, v).transpose(0, 1).reshape(T, B, d)
    return F.linear(o, sd[pfx + "out_proj.weight"], sd[pfx + "out_proj.bias"])


def _ln(sd: SD, pfx: str, x: Tensor) -> Tensor:
    """nn.LayerNorm(d), eps 1e-5, affine (transformer.py:227-229)."""
    return F.layer_norm(x, (x.shape[-1],), sd[pfx + "weight"], sd[pfx + "bias"], 1e-5)


@dataclass
class EncCfg:
    """Per-encoder hyper-parameters (transformer.py:23-24; mmtr.py:398-429)."""
    num_heads: int
    layers: int
    attn_dropout: float = 0.0
    relu_dropout: float = 0.0
    res_dropout: float = 0.0
    embed_dropout: float = 0.0
    attn_mask: bool = True
    biprojection: bool = False


# ----------------------------------------------------------------------------
# encoder layer / stack  (transformer.py:52-93, 141-195)
# ----------------------------------------------------------------------------
def encoder_layer(sd: SD, pfx: str, c: EncCfg, x: Tensor, k_e: Optional[Tensor],
                  v_e: Optional[Tensor], training: bool) -> Tensor:
    T = x.shape[0]
    att = pfx + "self_attn."
    ln = lambda i, t: _ln(sd, f"{pfx}layer_norms.{i}.", t)
    drop = lambda t, p: F.dropout(t, p=p, training=training)
    if k_e is None:                                        # self-only (A.5; :158-159)
        xn = ln(0, x)
        m = future_mask(T, T) if c.attn_mask else None
        x = x + drop(mha(sd, att, xn, xn, xn, c.num_heads, m, c.attn_dropout, training), c.res_dropout)
        ffn_ln = 1
    elif c.biprojection:                                   # A.4; :154-169
        S = k_e.shape[0]
        xn = ln(0, x)
        m = future_mask(T, T) if c.attn_mask else None
        x = x + drop(mha(sd, att, xn, xn, xn, c.num_heads, m, c.attn_dropout, training), c.res_dropout)
        m = future_mask(T, S) if c.attn_mask else None
        # same attention weights, query NOT normalised (:165-169)
        a = mha(sd, att, x, ln(1, k_e), ln(1, v_e), c.num_heads, m, c.attn_dropout, training)
        x = x + drop(a, c.res_dropout)
        ffn_ln = 2
    else:                                                  # crossmodal, A.3; :170-175
        S = k_e.shape[0]
        m = future_mask(T, S) if c.attn_mask else None
        a = mha(sd, att, ln(0, x), ln(0, k_e), ln(0, v_e), c.num_heads, m, c.attn_dropout, training)
        x = x + drop(a, c.res_dropout)
        ffn_ln = 1
    h = F.relu(F.linear(ln(ffn_ln, x), sd[pfx + "fc1.weight"], sd[pfx + "fc1.bias"]))
    h = drop(h, c.relu_dropout)
    h = F.linear(h, sd[pfx + "fc2.weight"], sd[pfx + "fc2.bias"])
    return x + drop(h, c.res_dropout)                      # :186-190


def encoder(sd: SD, pfx: str, c: EncCfg, x_in: Tensor, x_in_k: Optional[Tensor] = None,
            x_in_v: Optional[Tensor] = None, training: bool = False) -> Tensor:
    """TransformerEncoder.forward (transformer.py:52-93): sqrt(d) scale + PE,
    three independent embedding dropouts, layer loop, final LayerNorm."""
    d = x_in.shape[-1]
    sc = math.sqrt(d)
    emb = lambda t: F.dropout(sc * t + pos_embedding(t), p=c.embed_dropout, training=training)
    x = emb(x_in)
    k_e = v_e = None
    if x_in_k is not None and x_in_v is not None:
        k_e, v_e = emb(x_in_k), emb(x_in_v)
    for i in range(c.layers):
        x = encoder_layer(sd, f"{pfx}layers.{i}.", c, x, k_e, v_e, training)
    return _ln(sd, pfx + "layer_norm.", x)


# ----------------------------------------------------------------------------
# Fusion-GMU family  (mmtr.py:179-247)
# ----------------------------------------------------------------------------
def gmu_features(sd: SD, pfx: str, x1: Tensor, x2: Tensor) -> Tuple[Tensor, Tensor]:
    """GatedMultimodalLayerFeatures.forward (mmtr.py:189-195)."""
    h1 = torch.tanh(F.linear(x1, sd[pfx + "hidden1.weight"]))
    h2 = torch.tanh(F.linear(x2, sd[pfx + "hidden2.weight"]))
    z = torch.sigmoid(F.linear(torch.cat([x1, x2], -1), sd[pfx + "x_gate.weight"]))
    return z * h1 * x1 + (1 - z) * h2 * x2, torch.cat([z, 1 - z], -1)


def text_shifting(sd: SD, pfx: str, xs: Sequence[Tensor]) -> Tuple[Tensor, Tensor]:
    """TextShifting{3,4}Layer.forward (mmtr.py:210-219, 236-247): sum_i z_i *
    tanh(W_i x_i), z_i = sigmoid(G_i [x_1|..|x_n]); returns (fused, cat(z_i))."""
    cat = torch.cat(list(xs), -1)
    out, zs = 0, []
    for i, x in enumerate(xs, 1):
        h = torch.tanh(F.linear(x, sd[f"{pfx}hidden{i}.weight"]))
        z = torch.sigmoid(F.linear(cat, sd[f"{pfx}x{i}_gate.weight"]))
        out = out + z * h
        zs.append(z)
    return out, torch.cat(zs, -1)


# ----------------------------------------------------------------------------
# model graphs  (mmtr.py:735-866 three-modal, 444-583 four-modal)
# ----------------------------------------------------------------------------
@dataclass
class ModelCfg:
    """The args fields the models read (mmtr.py:284-302, 594-613)."""
    hidden_sz: int
    num_heads: int
    layers: int
    n_classes: int
    orig_d_l: int = 768
    orig_d_v: int = 35
    orig_d_a: int = 74
    orig_d_p: int = 4096
    attn_dropout: float = 0.0
    attn_dropout_v: float = 0.0
    attn_dropout_a: float = 0.0
    relu_dropout: float = 0.0
    res_dropout: float = 0.0
    out_dropout: float = 0.0
    embed_dropout: float = 0.0
    attn_mask: bool = True
    # sequence lengths are source constants in the reference (mmtr.py:371-377,
    # 664-670); exposed here so small shapes can be tested
    num_vectors_l: int = 512
    num_vectors_a: int = 512
    num_vectors_v: int = 512


# tag -> key/value source modality decides the attention dropout (A.8;
# mmtr.py:398-404 / 692-697): the LAST letter of the tag.
ENC_TAGS = {
    "trans_l_with_a": "la", "trans_l_with_v": "lv", "trans_l_with_v2a": "lv2a", "trans_l_with_a2v": "la2v",
    "trans_v_with_l": "vl", "trans_v_with_a": "va", "trans_v_with_l2a": "vl2a", "trans_v_with_a2l": "va2l",
    "trans_a_with_l": "al", "trans_a_with_v": "av", "trans_a_with_v2l": "av2l", "trans_a_with_l2v": "al2v",
}
LEVEL2 = ("trans_l_with_v2a", "trans_l_with_a2v", "trans_v_with_l2a", "trans_v_with_a2l",
          "trans_a_with_v2l", "trans_a_with_l2v")


def enc_cfg(m: ModelCfg, name: str, four_modal: bool) -> EncCfg:
    last = ENC_TAGS[name][-1]
    p = {"l": m.attn_dropout, "a": m.attn_dropout_a, "v": m.attn_dropout_v}[last]
    return EncCfg(m.num_heads, m.layers, p, m.relu_dropout, m.res_dropout, m.embed_dropout,
                  m.attn_mask, biprojection=four_modal and name in LEVEL2)


def _pad_time(x: Tensor, n: int) -> Tensor:
    """Zero-pad the time axis of [T,B,d] to n rows (mmtr.py:431-441)."""
    if x.shape[0] == n:
        return x
    return torch.cat([x, x.new_zeros(n - x.shape[0], *x.shape[1:])], 0)


def _stage(sd: SD, m: ModelCfg, name: str, x_bct: Tensor, orig_d: int, n: int) -> Tensor:
    """[B,orig_d,T] -> conv1x1 (skipped when orig_d == d) -> [T,B,d] -> pad.
    mmtr.py:456-469 / 748-761."""
    if orig_d != m.hidden_sz:
        x_bct = F.conv1d(x_bct, sd[name + ".weight"])
    return _pad_time(x_bct.permute(2, 0, 1), n)


def _time_linear(sd: SD, name: str, h: Tensor) -> Tensor:
    """nn.Linear over the TIME axis on a permute(2,1,0) view (mmtr.py:507-508)."""
    return F.linear(h.permute(2, 1, 0), sd[name + ".weight"], sd[name + ".bias"]).permute(2, 1, 0)


def _head(sd: SD, m: ModelCfg, h: Tensor, training: bool) -> Tensor:
    """Residual MLP head (mmtr.py:577-583 / 860-866)."""
    y = F.linear(F.dropout(F.relu(F.linear(h, sd["proj1.weight"], sd["proj1.bias"])),
                           p=m.out_dropout, training=training), sd["proj2.weight"], sd["proj2.bias"]) + h
    return F.linear(y, sd["out_layer.weight"], sd["out_layer.bias"])


def _fuse_branch(sd, gm_mid, gm_top, lvl2_a, lvl1_a, lvl2_b, lvl1_b, mid_x1, mid_x2):
    """One target modality's fusion (A.9): middle GMU on two level-1 outputs,
    level 1->2 residuals, top GMU, level 1->3 residual, first+last timestep."""
    mid, _ = gmu_features(sd, gm_mid, mid_x1, mid_x2)
    top, _ = gmu_features(sd, gm_top, lvl2_a + lvl1_a, lvl2_b + lvl1_b)
    tot = top + mid
    return tot[0] + tot[-1]


def bpmult3_forward(sd: SD, m: ModelCfg, x_l_feat: Tensor, img: Tensor, audio: Tensor,
                    training: bool = False) -> Tuple[Tensor, Tensor]:
    """MultiprojectionMMTransformer3DGMUClf.forward (mmtr.py:735-866).
    x_l_feat [B,L,orig_d_l] is the text encoder's output (BERT is upstream of
    the path); img [B,V,orig_d_v]; audio [B,A,orig_d_a].  Returns (logits, z)."""
    E = lambda n, q, kv: encoder(sd, n + ".", enc_cfg(m, n, False), q, kv, kv, training)
    x_l = F.dropout(x_l_feat.transpose(1, 2), p=m.embed_dropout, training=training)   # :741
    pl = _stage(sd, m, "proj_l", x_l, m.orig_d_l, m.num_vectors_l)
    pa = _stage(sd, m, "proj_a", audio.transpose(1, 2), m.orig_d_a, m.num_vectors_a)
    pv = _stage(sd, m, "proj_v", img.transpose(1, 2), m.orig_d_v, m.num_vectors_v)
    # level 1 (:779-786)
    v_a, a_v = E("trans_v_with_a", pv, pa), E("trans_a_with_v", pa, pv)
    v_l, l_v = E("trans_v_with_l", pv, pl), E("trans_l_with_v", pl, pv)
    a_l, l_a = E("trans_a_with_l", pa, pl), E("trans_l_with_a", pl, pa)
    # level 2 + fusion, argument order exactly as :790-852
    l_v2a, l_a2v = E("trans_l_with_v2a", pl, a_v), E("trans_l_with_a2v", pl, v_a)
    last_l = _fuse_branch(sd, "gmu_l_m.", "gmu_l.", l_a2v, v_a, l_v2a, a_v, v_a, a_v)
    a_v2l, a_l2v = E("trans_a_with_v2l", pa, l_v), E("trans_a_with_l2v", pa, v_l)
    last_a = _fuse_branch(sd, "gmu_a_m.", "gmu_a.", a_v2l, l_v, a_l2v, v_l, l_v, v_l)
    v_a2l, v_l2a = E("trans_v_with_a2l", pv, l_a), E("trans_v_with_l2a", pv, a_l)
    last_v = _fuse_branch(sd, "gmu_v_m.", "gmu_v.", v_a2l, l_a, v_l2a, a_l, l_a, a_l)
    h, z = text_shifting(sd, "gmu.", [last_l, last_v, last_a])                         # :857
    return _head(sd, m, h, training), z


def audio_encoder(sd: SD, audio: Tensor) -> Tensor:
    """AudioEncoder (mmtr.py:93-108): 2x Conv1d(96,96,k=128,s=2) + AdaptiveAvgPool1d(200)."""
    x = F.conv1d(audio, sd["audio_enc.conv_layers.0.weight"], sd["audio_enc.conv_layers.0.bias"], stride=2)
    x = F.conv1d(x, sd["audio_enc.conv_layers.1.weight"], sd["audio_enc.conv_layers.1.bias"], stride=2)
    return F.adaptive_avg_pool1d(x, 200)


def bpmult4_forward(sd: SD, m: ModelCfg, x_l_feat: Tensor, img: Tensor, audio_feat: Tensor,
                    poster: Tensor, training: bool = False) -> Tuple[Tensor, Tensor]:
    """MultiprojectionMMTransformerGMUClf.forward (mmtr.py:444-583).
    audio_feat [B,orig_d_a,A] is the AudioEncoder output (front-end, SURVEY
    8(f) "next"); poster [B,orig_d_p]."""
    E = lambda n, q, kv: encoder(sd, n + ".", enc_cfg(m, n, True), q, kv, kv, training)
    x_l = F.dropout(x_l_feat.transpose(1, 2), p=m.embed_dropout, training=training)   # :449
    pl = _stage(sd, m, "proj_l", x_l, m.orig_d_l, m.num_vectors_l)
    pa = _stage(sd, m, "proj_a", audio_feat, m.orig_d_a, m.num_vectors_a)
    pv = _stage(sd, m, "proj_v", img.transpose(1, 2), m.orig_d_v, m.num_vectors_v)
    post = F.linear(poster, sd["proj_poster.weight"])                                  # :486
    v_a, a_v = E("trans_v_with_a", pv, pa), E("trans_a_with_v", pa, pv)
    v_l, l_v = E("trans_v_with_l", pv, pl), E("trans_l_with_v", pl, pv)
    a_l, l_a = E("trans_a_with_l", pa, pl), E("trans_l_with_a", pl, pa)
    # l  (:501-522)
    l_v2a, l_a2v = E("trans_l_with_v2a", pl, a_v), E("trans_l_with_a2v", pl, v_a)
    t_a_v, t_v_a = _time_linear(sd, "transfm_a2l", a_v), _time_linear(sd, "transfm_v2l", v_a)
    last_l = _fuse_branch(sd, "gmu_l_m.", "gmu_l.", l_a2v, t_v_a, l_v2a, t_a_v, t_v_a, t_a_v)
    # a  (:524-545)
    a_v2l, a_l2v = E("trans_a_with_v2l", pa, l_v), E("trans_a_with_l2v", pa, v_l)
    t_l_v = _time_linear(sd, "transfm_l2a", l_v)
    last_a = _fuse_branch(sd, "gmu_a_m.", "gmu_a.", a_v2l, t_l_v, a_l2v, v_l, t_l_v, v_l)
    # v  (:547-568)
    v_a2l, v_l2a = E("trans_v_with_a2l", pv, l_a), E("trans_v_with_l2a", pv, a_l)
    t_l_a = _time_linear(sd, "transfm_l2v", l_a)
    last_v = _fuse_branch(sd, "gmu_v_m.", "gmu_v.", v_a2l, t_l_a, v_l2a, a_l, t_l_a, a_l)
    h, z = text_shifting(sd, "gmu.", [last_l, last_v, last_a, post])                   # :574
    return _head(sd, m, h, training), z


# ----------------------------------------------------------------------------
# parameter shapes (the reference's state_dict ABI)
# ----------------------------------------------------------------------------
def encoder_param_shapes(pfx: str, d: int, layers: int, biprojection: bool) -> Dict[str, Tuple[int, ...]]:
    s: Dict[str, Tuple[int, ...]] = {}
    for i in range(layers):
        p = f"{pfx}layers.{i}."
        s[p + "self_attn.in_proj_weight"] = (3 * d, d)
        s[p + "self_attn.in_proj_bias"] = (3 * d,)
        s[p + "self_attn.out_proj.weight"] = (d, d)
        s[p + "self_attn.out_proj.bias"] = (d,)
        s[p + "fc1.weight"] = (4 * d, d)
        s[p + "fc1.bias"] = (4 * d,)
        s[p + "fc2.weight"] = (d, 4 * d)
        s[p + "fc2.bias"] = (d,)
        for j in range(3 if biprojection else 2):
            s[p + f"layer_norms.{j}.weight"] = (d,)
            s[p + f"layer_norms.{j}.bias"] = (d,)
    s[pfx + "layer_norm.weight"] = (d,)
    s[pfx + "layer_norm.bias"] = (d,)
    return s


def model_param_shapes(m: ModelCfg, four_modal: bool) -> Dict[str, Tuple[int, ...]]:
    """Trainable parameters of the hot path, reference key names, in the
    reference's registration order is NOT guaranteed -- use as a dict."""
    d = m.hidden_sz
    s: Dict[str, Tuple[int, ...]] = {}
    if four_modal:
        s["audio_enc.conv_layers.0.weight"] = (96, 96, 128)
        s["audio_enc.conv_layers.0.bias"] = (96,)
        s["audio_enc.conv_layers.1.weight"] = (96, 96, 128)
        s["audio_enc.conv_layers.1.bias"] = (96,)
        s["proj_poster.weight"] = (d, m.orig_d_p)
    for g in ("gmu_l_m", "gmu_v_m", "gmu_a_m", "gmu_l", "gmu_v", "gmu_a"):
        s[g + ".hidden1.weight"] = (d, d)
        s[g + ".hidden2.weight"] = (d, d)
        s[g + ".x_gate.weight"] = (d, 2 * d)
    s["proj_l.weight"] = (d, m.orig_d_l, 1)
    s["proj_v.weight"] = (d, m.orig_d_v, 1)
    s["proj_a.weight"] = (d, m.orig_d_a, 1)
    for n in ENC_TAGS:
        s.update(encoder_param_shapes(n + ".", d, m.layers, four_modal and n in LEVEL2))
    for n in ("proj1", "proj2"):
        s[n + ".weight"] = (d, d)
        s[n + ".bias"] = (d,)
    s["out_layer.weight"] = (m.n_classes, d)
    s["out_layer.bias"] = (m.n_classes,)
    n_in = 4 if four_modal else 3
    for i in range(1, n_in + 1):
        s[f"gmu.hidden{i}.weight"] = (d, d)
        s[f"gmu.x{i}_gate.weight"] = (d, n_in * d)
    L, A, V = m.num_vectors_l, m.num_vectors_a, m.num_vectors_v
    s["transfm_a2l.weight"], s["transfm_a2l.bias"] = (L, A), (L,)
    s["transfm_v2l.weight"], s["transfm_v2l.bias"] = (L, V), (L,)
    s["transfm_l2a.weight"], s["transfm_l2a.bias"] = (A, L), (A,)
    s["transfm_l2v.weight"], s["transfm_l2v.bias"] = (V, L), (V,)
    return s
