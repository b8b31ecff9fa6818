#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the REAL reference (Damorgal/BPMulT).

Runs only in the build container, where the reference is mounted read-only at
/root/reference; the fixtures it writes are what travels to the GPU box.  The
reference does not import as shipped; this script
applies the five arithmetic-neutral shims documented there:

  (i)   make_positions -> .contiguous()            (position_embedding.py:76 on torch >= 2)
  (ii)  in_proj_qkv clones its chunks              (in-place q *= scaling, multihead_attention.py:86)
  (iii) transfm_2dim -> device-agnostic zero pad   (hard-coded .cuda(), mmtr.py:431-441)
  (iv)  BertEncoder -> returns supplied features   (no pretrained weights offline)
  (v)   TextShifting3Layer 4-arg constructor       (arity bug, mmtr.py:663 vs :199)

Weights / inputs are regenerated from names by tests/golden/detgen.py, so the
fixtures only carry reference OUTPUTS (logits, gates, gradients).

usage:  python tests/golden/make_golden.py [--only f7,f9]
"""
import argparse
import importlib.machinery
import os
import sys
import types
from argparse import Namespace

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference")

from transformers import BertModel  # noqa: F401,E402  (must precede the torchvision stub)


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, None)
    m.__path__ = []
    m.__dict__.update(attrs)
    sys.modules[name] = m


class _D:
    pass


for _n, _a in [("torchvision", {}), ("torchvision.models", {}),
               ("detectron2", {"model_zoo": _D}), ("detectron2.modeling", {"build_model": _D}),
               ("detectron2.config", {"get_cfg": _D}), ("detectron2.checkpoint", {"DetectionCheckpointer": _D}),
               ("detectron2.structures", {}), ("detectron2.structures.image_list", {"ImageList": _D}),
               ("pytorch_pretrained_bert", {}), ("pytorch_pretrained_bert.modeling", {"BertModel": _D})]:
    _stub(_n, **_a)

import torch  # noqa: E402
from torch import nn  # noqa: E402

import bpmult.models.mmtr as mmtr  # noqa: E402
import bpmult.models.position_embedding as pe  # noqa: E402
import bpmult.models.transformer as tr  # noqa: E402
from bpmult.models.multihead_attention import MultiheadAttention as MHA  # noqa: E402

from detgen import det, det_param  # noqa: E402

_mp = pe.make_positions
pe.make_positions = lambda t, p, l: _mp(t, p, l).contiguous()                                  # (i)
MHA.in_proj_qkv = lambda s, q: tuple(c.clone() for c in s._in_proj(q).chunk(3, dim=-1))         # (ii)


def _pad(self, x, dim, n):                                                                      # (iii)
    if x.size(dim) != n:
        shp = list(x.shape)
        shp[dim] = n - x.size(dim)
        x = torch.cat((x, x.new_zeros(shp)), dim)
    return x


for _C in (mmtr.MultiprojectionMMTransformer3DGMUClf, mmtr.MultiprojectionMMTransformerGMUClf):
    _C.transfm_2dim = _pad


class FeatText(nn.Module):                                                                      # (iv)
    """Stands in for BertEncoder: returns the feature tensor parked on it."""

    def __init__(self, args):
        super().__init__()
        self.feat = None

    def forward(self, txt, mask, segment):
        return self.feat


mmtr.BertEncoder = FeatText
_T3 = mmtr.TextShifting3Layer


class TS3(_T3):                                                                                 # (v)
    def __init__(self, a, b, c, o):
        nn.Module.__init__(self)
        self.hidden1, self.hidden2, self.hidden3 = (nn.Linear(i, o, bias=False) for i in (a, b, c))
        self.x1_gate, self.x2_gate, self.x3_gate = (nn.Linear(a + b + c, o, bias=False) for _ in range(3))


mmtr.TextShifting3Layer = TS3

T = torch.from_numpy
SKIP = ("version", "_float_tensor")


def load_det(mod, prefix=""):
    """Overwrite every parameter of `mod` with det_param(prefix+name)."""
    with torch.no_grad():
        for k, p in mod.named_parameters():
            p.copy_(T(det_param(prefix + k, p.shape)))
    return {prefix + k: tuple(p.shape) for k, p in mod.named_parameters()}


def leaf(name, shape, scale=1.0):
    return T(det(name, shape, scale)).requires_grad_(True)


def grads_of(mod, prefix=""):
    return {"g." + prefix + k: p.grad.numpy() for k, p in mod.named_parameters() if p.grad is not None}


def save(name, **arrs):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **{k: np.asarray(v) for k, v in arrs.items()})
    print(f"  wrote {name}.npz  ({os.path.getsize(path) / 1024:.1f} KiB, {len(arrs)} arrays)")


def zero_some_channel0(x, name):
    """Make a few channel-0 entries exactly 0 so the padding-row rule fires."""
    with torch.no_grad():
        m = T(det(name + ".z", x.shape[:2])) > 1.0
        x[:, :, 0][m] = 0.0
    return x


# ---------------------------------------------------------------------------
def f1_posemb():
    out = {}
    for d, (Tn, B) in ((24, (9, 3)), (25, (7, 2)), (300, (512, 1))):
        x = zero_some_channel0(T(det(f"f1.x.{d}", (Tn, B, d))), f"f1.x.{d}")
        emb = pe.SinusoidalPositionalEmbedding(d)
        y = emb(x.transpose(0, 1)[:, :, 0]).transpose(0, 1).contiguous().numpy()
        if d == 300:
            rows = np.array([0, 1, 2, 255, 510, 511])
            out["rows.300"] = rows
            y = y[rows]
        out[f"y.{d}"] = y
    save("f1_posemb", **out)


def f2_mask():
    out = {}
    for (a, b) in ((4, 4), (3, 6), (6, 3), (512, 200), (200, 512), (512, 400), (50, 50), (2, 512)):
        m = tr.buffered_future_mask(torch.zeros(a, 1, 1), torch.zeros(b, 1, 1))
        assert m.shape == (a, b)
        out[f"m.{a}.{b}"] = np.packbits((m == float("-inf")).numpy(), axis=1)
    save("f2_mask", **out)


def f3_mha():
    out = {}
    for tag, d, H, Tn, S, B, same in (("a", 24, 4, 5, 7, 2, False), ("b", 300, 12, 6, 4, 2, False),
                                      ("s", 24, 4, 6, 6, 3, True)):
        pfx = f"f3{tag}."
        m = MHA(d, H)
        load_det(m, pfx)
        q = leaf(pfx + "q", (Tn, B, d))
        if same:
            k = v = q
        else:
            k, v = leaf(pfx + "k", (S, B, d)), leaf(pfx + "v", (S, B, d))
        mask = tr.buffered_future_mask(q, k)
        y, _ = m(q, k, v, attn_mask=mask)
        (y * T(det(pfx + "w", y.shape))).sum().backward()
        out[f"{tag}.y"] = y.detach().numpy()
        out[f"{tag}.gq"] = q.grad.numpy()
        if not same:
            out[f"{tag}.gk"], out[f"{tag}.gv"] = k.grad.numpy(), v.grad.numpy()
        for kk, g in grads_of(m).items():
            out[f"{tag}.{kk}"] = g
    save("f3_mha", **out)


def f4_layer():
    out = {}
    d, H, B = 24, 4, 2
    for tag, bi, Tn, S in (("x", False, 6, 9), ("b", True, 5, 8), ("xs", False, 9, 9)):
        pfx = f"f4{tag}."
        m = tr.TransformerEncoderLayer(d, num_heads=H, attn_dropout=0., relu_dropout=0., res_dropout=0.,
                                       attn_mask=True, biprojection=bi)
        load_det(m, pfx)
        x, k, v = leaf(pfx + "x", (Tn, B, d)), leaf(pfx + "k", (S, B, d)), leaf(pfx + "v", (S, B, d))
        y = m(x, k, v)
        (y * T(det(pfx + "w", y.shape))).sum().backward()
        out.update({f"{tag}.y": y.detach().numpy(), f"{tag}.gx": x.grad.numpy(),
                    f"{tag}.gk": k.grad.numpy(), f"{tag}.gv": v.grad.numpy()})
        for kk, g in grads_of(m).items():
            out[f"{tag}.{kk}"] = g
    save("f4_layer", **out)


def f5_encoder():
    out = {}
    d, H, B, Ly = 24, 4, 2, 2
    for tag, bi, Tn, S, mask in (("x", False, 7, 5, True), ("b", True, 5, 8, True), ("s", False, 6, 0, True),
                                 ("xn", False, 6, 6, False), ("x25", False, 8, 11, True)):
        dd, HH = (50, 2) if tag == "x25" else (d, H)          # head_dim 25 like hidden 300 / 12 heads
        pfx = f"f5{tag}."
        m = tr.TransformerEncoder(dd, HH, Ly, attn_mask=mask, biprojection=bi)
        load_det(m, pfx)
        x = zero_some_channel0(T(det(pfx + "x", (Tn, B, dd))), pfx + "x")
        with torch.no_grad():
            x[-2:] = 0.0                                       # zero-padded tail rows
        x.requires_grad_(True)
        if S:
            kv = zero_some_channel0(T(det(pfx + "kv", (S, B, dd))), pfx + "kv").requires_grad_(True)
            y = m(x, kv, kv)
        else:
            kv = None
            y = m(x)
        (y * T(det(pfx + "w", y.shape))).sum().backward()
        out.update({f"{tag}.y": y.detach().numpy(), f"{tag}.gx": x.grad.numpy()})
        if kv is not None:
            out[f"{tag}.gkv"] = kv.grad.numpy()
        for kk, g in grads_of(m).items():
            out[f"{tag}.{kk}"] = g
    save("f5_encoder", **out)


def f6_gmu():
    out = {}
    d = 24
    g = mmtr.GatedMultimodalLayerFeatures(d, d, d)
    load_det(g, "f6g.")
    x1, x2 = leaf("f6g.x1", (5, 2, d)), leaf("f6g.x2", (5, 2, d))
    y, z = g([x1, x2])
    ((y * T(det("f6g.w", y.shape))).sum() + (z * T(det("f6g.wz", z.shape))).sum() * 0).backward()
    out.update({"g.y": y.detach().numpy(), "g.z": z.detach().numpy(), "g.gx1": x1.grad.numpy(), "g.gx2": x2.grad.numpy()})
    for kk, gg in grads_of(g).items():
        out["g." + kk] = gg
    for n, cls in ((3, TS3), (4, mmtr.TextShifting4Layer)):
        pfx = f"f6t{n}."
        m = cls(*([d] * (n + 1)))
        load_det(m, pfx)
        xs = [leaf(pfx + f"x{i}", (3, d)) for i in range(n)]
        y, z = m(xs)
        (y * T(det(pfx + "w", y.shape))).sum().backward()
        out.update({f"t{n}.y": y.detach().numpy(), f"t{n}.z": z.detach().numpy()})
        for i, x in enumerate(xs):
            out[f"t{n}.gx{i}"] = x.grad.numpy()
        for kk, gg in grads_of(m).items():
            out[f"t{n}.{kk}"] = gg
    save("f6_gmu", **out)


def _args(**kw):
    a = dict(orig_d_l=768, orig_d_v=35, orig_d_a=74, orig_d_p=4096, hidden_sz=300, vonly=True, lonly=True,
             aonly=True, num_heads=12, layers=8, attn_dropout=0., attn_dropout_v=0., attn_dropout_a=0.,
             relu_dropout=0., res_dropout=0., out_dropout=0., embed_dropout=0., attn_mask=True, hybrid=False,
             n_classes=6, bert_model="unused")
    a.update(kw)
    return Namespace(**a)


FULL_GRADS_3 = ["out_layer.weight", "out_layer.bias", "proj1.weight", "proj2.bias", "proj_l.weight",
                "proj_v.weight", "proj_a.weight", "gmu.hidden1.weight", "gmu.x3_gate.weight",
                "gmu_l.x_gate.weight", "gmu_a_m.hidden2.weight", "gmu_v.hidden1.weight",
                "trans_l_with_a.layers.0.self_attn.in_proj_weight", "trans_l_with_a.layers.0.self_attn.in_proj_bias",
                "trans_v_with_a.layers.1.fc1.weight", "trans_a_with_v.layers.0.fc2.bias",
                "trans_l_with_v2a.layers.1.self_attn.out_proj.weight", "trans_v_with_l2a.layers.0.layer_norms.0.weight",
                "trans_a_with_l2v.layers.1.layer_norms.1.bias", "trans_a_with_v2l.layer_norm.weight",
                "trans_v_with_l.layers.0.self_attn.out_proj.bias"]


def _model_fixture(name, model, pfx, call, inputs, n_classes, full_grads, small_only=False):
    shapes = load_det(model, pfx)
    logits, z = call()
    tgt = (T(det(pfx + "tgt", logits.shape)) > 0).float()
    loss = nn.BCEWithLogitsLoss()(logits, tgt)
    loss.backward()
    out = {"logits": logits.detach().numpy(), "z": z.detach().numpy(), "loss": loss.detach().numpy(),
           "param_names": np.array(sorted(k[len(pfx):] for k in shapes)),
           "param_shapes": np.array([",".join(map(str, shapes[pfx + k])) for k in sorted(k[len(pfx):] for k in shapes)])}
    nograd = []
    for k, p in model.named_parameters():
        if p.grad is None:
            nograd.append(k)
            continue
        g = p.grad.double()
        out["gn." + k] = np.array([g.norm().item(), g.sum().item()])
        if (k in full_grads and (not small_only or p.numel() <= 12000)) or (not small_only and p.numel() <= 64):
            out["g." + k] = p.grad.numpy()
    out["nograd"] = np.array(nograd)
    for k, t in inputs.items():
        if t.grad is not None:
            out["gin." + k] = t.grad.numpy()
    save(name, **out)


def _three_modal(name, pfx, d, H, Ly, B, L, V, A, odl, small_only=False, nv=None):
    torch.manual_seed(0)
    args = _args(hidden_sz=d, num_heads=H, layers=Ly, orig_d_l=odl)
    model = mmtr.MultiprojectionMMTransformer3DGMUClf(args)
    if nv is not None:                                     # the padded lengths are source constants (mmtr.py:664-670): set on the instance
        model.num_vectors_l = model.num_vectors_a = model.num_vectors_v = nv
    model.train()                                          # all dropout rates are 0
    xl = leaf(pfx + "xl", (B, L, odl))
    img, aud = leaf(pfx + "img", (B, V, 35)), leaf(pfx + "aud", (B, A, 74))
    model.enc.feat = xl
    call = lambda: model(None, None, None, img, aud, output_gate=True)
    _model_fixture(name, model, pfx, call, {"xl": xl, "img": img, "aud": aud}, 6, FULL_GRADS_3, small_only)


def f7_mmtrvat():
    _three_modal("f7_mmtrvat", "f7.", 24, 4, 2, 2, 50, 500, 375, 32)


def f9_cfg1():
    # IEMOCAP shape of BASELINE.json configs[0/1] at B=2: d=300, 12 heads (head_dim 25), 8 layers
    _three_modal("f9_cfg1", "f9.", 300, 12, 8, 2, 20, 500, 400, 768, small_only=True)


def f11_h768():
    """The bench's headline workload (BASELINE.json metric: hidden 768, 3-modal unaligned): `mmtrvat` at d=768, 12 heads
    (head_dim 64), 8 layers, orig_d 768/35/74 (proj_l skipped: mmtr.py:748), L/V/A = 20/500/400 -> 512, B=1.  680 M
    parameters: logits, gates, loss, every parameter's gradient norm, the small full gradients, the input gradients."""
    _three_modal("f11_h768", "f11.", 768, 12, 8, 1, 20, 500, 400, 768, small_only=True)


def f12_k768():
    """The north-star kernel-point model (BASELINE.json north_star: hidden 768 / seq_len 50; bench.py `k768`): `mmtrvat` at
    d=768, 6 heads (head_dim 128), 5 layers, every modality 50 time steps (num_vectors_* = 50: T = S = 50 in all twelve
    encoders, strictly causal masks), B=2.  The unused time-axis maps keep the reference's 512 x 512 shapes (no gradient)."""
    _three_modal("f12_k768", "f12.", 768, 6, 5, 2, 50, 50, 50, 768, small_only=True, nv=50)


def f8_mmtrvapt():
    pfx = "f8."
    d, H, Ly, B = 24, 4, 2, 2
    args = _args(hidden_sz=d, num_heads=H, layers=Ly, orig_d_l=32, orig_d_v=40, orig_d_a=96, orig_d_p=64, n_classes=13)
    model = mmtr.MultiprojectionMMTransformerGMUClf(args)
    model.train()
    xl, img = leaf(pfx + "xl", (B, 60, 32)), leaf(pfx + "img", (B, 150, 40))
    aud, post = leaf(pfx + "aud", (B, 96, 1000)), leaf(pfx + "post", (B, 64))
    model.enc.feat = xl
    feat = {}
    def _keep(mod, inp, o):
        o.retain_grad()
        feat["a"] = o

    hook = model.audio_enc.register_forward_hook(_keep)
    call = lambda: model(None, None, None, img, aud, post, output_gate=True)
    full = [k.replace("trans_l_with_v2a.layers.1.self_attn.out_proj.weight", "trans_l_with_v2a.layers.1.layer_norms.2.weight")
            for k in FULL_GRADS_3] + ["gmu.hidden4.weight", "gmu.x4_gate.weight", "proj_poster.weight",
                                      "transfm_a2l.weight", "transfm_l2v.bias", "trans_l_with_v2a.layers.0.self_attn.in_proj_weight",
                                      "trans_v_with_a2l.layers.1.fc2.weight"]
    _model_fixture("f8_mmtrvapt", model, pfx, call, {"xl": xl, "img": img, "post": post}, 13, full)
    hook.remove()
    # the AudioEncoder output and its gradient (front-end boundary of the hot path)
    path = os.path.join(HERE, "f8_mmtrvapt.npz")
    old = dict(np.load(path))
    old["audio_feat"] = feat["a"].detach().numpy()
    old["gin.audio_feat"] = feat["a"].grad.numpy()
    np.savez_compressed(path, **old)


def f10_cfg3():
    """BASELINE.json configs[2] dims: Moviescope-shape 4-modal `mmtrvapt`, d=768, 6 heads (head_dim 128), 5 layers,
    orig_d_v=4096, L=512 / V=A=200 (biprojection level-2 encoders, time-axis maps, AudioEncoder front-end), B=1.
    462 M parameters: the fixture keeps logits, gates, loss, every parameter's gradient norm and the full gradients of
    the small tensors only (weights / inputs are regenerated from names)."""
    pfx = "f10."
    args = _args(hidden_sz=768, num_heads=6, layers=5, orig_d_l=768, orig_d_v=4096, orig_d_a=96, orig_d_p=4096, n_classes=13)
    model = mmtr.MultiprojectionMMTransformerGMUClf(args)
    model.train()
    B = 1
    xl, img = leaf(pfx + "xl", (B, 512, 768)), leaf(pfx + "img", (B, 200, 4096))
    aud, post = leaf(pfx + "aud", (B, 96, 1000)), leaf(pfx + "post", (B, 4096))
    model.enc.feat = xl
    feat = {}

    def _keep(mod, inp, o):
        o.retain_grad()
        feat["a"] = o

    hook = model.audio_enc.register_forward_hook(_keep)
    call = lambda: model(None, None, None, img, aud, post, output_gate=True)
    full = ["out_layer.weight", "out_layer.bias", "proj2.bias", "transfm_l2v.bias", "transfm_a2l.bias",
            "trans_l_with_a.layers.0.self_attn.in_proj_bias", "trans_a_with_v.layers.0.fc2.bias",
            "trans_v_with_l2a.layers.0.layer_norms.0.weight", "trans_a_with_l2v.layers.1.layer_norms.1.bias",
            "trans_l_with_v2a.layers.4.layer_norms.2.weight", "trans_a_with_v2l.layer_norm.weight",
            "trans_v_with_l.layers.0.self_attn.out_proj.bias", "trans_l_with_a2v.layers.2.fc1.bias",
            "trans_v_with_a2l.layers.3.self_attn.in_proj_bias", "audio_enc.conv_layers.0.bias", "audio_enc.conv_layers.1.bias"]
    _model_fixture("f10_cfg3", model, pfx, call, {"post": post}, 13, full, small_only=True)
    hook.remove()
    path = os.path.join(HERE, "f10_cfg3.npz")
    old = dict(np.load(path))
    # input-side checks that stay small: gradient norms of the big inputs, the poster gradient in full, and a strided
    # sample of the AudioEncoder output / its gradient
    for k, t in (("xl", xl), ("img", img), ("aud", aud)):
        g = t.grad.double()
        old["ginn." + k] = np.array([g.norm().item(), g.sum().item()])
    a = feat["a"]
    old["audio_feat_s"] = a.detach().numpy()[:, ::8, ::5]
    old["gin.audio_feat_s"] = a.grad.numpy()[:, ::8, ::5]
    old["ginn.audio_feat"] = np.array([a.grad.double().norm().item(), a.grad.double().sum().item()])
    np.savez_compressed(path, **old)
    print(f"  f10_cfg3.npz now {os.path.getsize(path) / 1024:.1f} KiB")


ALL = dict(f1=f1_posemb, f2=f2_mask, f3=f3_mha, f4=f4_layer, f5=f5_encoder, f6=f6_gmu,
           f7=f7_mmtrvat, f8=f8_mmtrvapt, f9=f9_cfg1, f10=f10_cfg3, f11=f11_h768, f12=f12_k768)

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    torch.set_num_threads(8)
    for k, fn in ALL.items():
        if a.only and k not in a.only.split(","):
            continue
        print(k)
        fn()
