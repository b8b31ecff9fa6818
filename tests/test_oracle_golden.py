"""The CPU oracle (oracle/bpmult_cpu.py) against every golden vector produced
from the real reference (tests/golden/make_golden.py).  Tolerance 1e-5 abs on
O(1) values (fp32, same torch build; differences are summation order only)."""
import os

import numpy as np
import pytest
import torch

from detgen import det, det_param
from oracle import bpmult_cpu as O

G = os.path.join(os.path.dirname(__file__), "golden")
T = torch.from_numpy


def load(name):
    return dict(np.load(os.path.join(G, name + ".npz")))


def sd_for(shapes, pfx):
    return {k: T(det_param(pfx + k, s)).requires_grad_(True) for k, s in shapes.items()}


def close(a, b, tol=1e-5, what=""):
    a = a.detach().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    scale = max(1.0, float(np.abs(b).max()))
    err = float(np.abs(a - b).max())
    assert a.shape == b.shape, (what, a.shape, b.shape)
    assert err <= tol * scale, f"{what}: max err {err:.3e} (scale {scale:.3g})"


def zero_some_channel0(x, name):
    m = T(det(name + ".z", x.shape[:2])) > 1.0
    x[:, :, 0][m] = 0.0
    return x


def test_f1_posemb():
    g = load("f1_posemb")
    for d, (Tn, B) in ((24, (9, 3)), (25, (7, 2)), (300, (512, 1))):
        x = zero_some_channel0(T(det(f"f1.x.{d}", (Tn, B, d))), f"f1.x.{d}")
        y = O.pos_embedding(x)
        if d == 300:
            y = y[g["rows.300"]]
        close(y, g[f"y.{d}"], 1e-6, f"posemb d={d}")


def test_f2_mask():
    g = load("f2_mask")
    for k, packed in g.items():
        _, a, b = k.split(".")
        a, b = int(a), int(b)
        ref = np.unpackbits(packed, axis=1)[:, :b].astype(bool)
        mine = (O.future_mask(a, b) == float("-inf")).numpy()
        assert (ref == mine).all(), k


def _mha_shapes(d):
    return {"in_proj_weight": (3 * d, d), "in_proj_bias": (3 * d,), "out_proj.weight": (d, d), "out_proj.bias": (d,)}


@pytest.mark.parametrize("tag,d,H,Tn,S,B,same", [("a", 24, 4, 5, 7, 2, False), ("b", 300, 12, 6, 4, 2, False),
                                                   ("s", 24, 4, 6, 6, 3, True)])
def test_f3_mha(tag, d, H, Tn, S, B, same):
    g = load("f3_mha")
    pfx = f"f3{tag}."
    sd = sd_for(_mha_shapes(d), pfx)
    q = T(det(pfx + "q", (Tn, B, d))).requires_grad_(True)
    if same:
        k = v = q
    else:
        k, v = (T(det(pfx + n, (S, B, d))).requires_grad_(True) for n in "kv")
    y = O.mha(sd, "", q, k, v, H, O.future_mask(Tn, k.shape[0]))
    (y * T(det(pfx + "w", y.shape))).sum().backward()
    close(y, g[f"{tag}.y"], what="y")
    close(q.grad, g[f"{tag}.gq"], what="gq")
    if not same:
        close(k.grad, g[f"{tag}.gk"], what="gk")
        close(v.grad, g[f"{tag}.gv"], what="gv")
    for n, p in sd.items():
        close(p.grad, g[f"{tag}.g.{n}"], what=n)


def _layer_shapes(d, bi):
    s = O.encoder_param_shapes("", d, 1, bi)
    return {k[len("layers.0."):]: v for k, v in s.items() if k.startswith("layers.0.")}


@pytest.mark.parametrize("tag,bi,Tn,S", [("x", False, 6, 9), ("b", True, 5, 8), ("xs", False, 9, 9)])
def test_f4_layer(tag, bi, Tn, S):
    g = load("f4_layer")
    d, H, B = 24, 4, 2
    pfx = f"f4{tag}."
    sd = sd_for(_layer_shapes(d, bi), pfx)
    x, k, v = (T(det(pfx + n, (t, B, d))).requires_grad_(True) for n, t in (("x", Tn), ("k", S), ("v", S)))
    y = O.encoder_layer(sd, "", O.EncCfg(H, 1, biprojection=bi), x, k, v, False)
    (y * T(det(pfx + "w", y.shape))).sum().backward()
    close(y, g[f"{tag}.y"], what="y")
    for n, t in (("gx", x), ("gk", k), ("gv", v)):
        close(t.grad, g[f"{tag}.{n}"], what=n)
    for n, p in sd.items():
        close(p.grad, g[f"{tag}.g.{n}"], what=n)


@pytest.mark.parametrize("tag,bi,Tn,S,mask", [("x", False, 7, 5, True), ("b", True, 5, 8, True), ("s", False, 6, 0, True),
                                             ("xn", False, 6, 6, False), ("x25", False, 8, 11, True)])
def test_f5_encoder(tag, bi, Tn, S, mask):
    g = load("f5_encoder")
    d, H = (50, 2) if tag == "x25" else (24, 4)
    B, Ly = 2, 2
    pfx = f"f5{tag}."
    sd = sd_for(O.encoder_param_shapes("", d, Ly, bi), pfx)
    x = zero_some_channel0(T(det(pfx + "x", (Tn, B, d))), pfx + "x")
    x[-2:] = 0.0
    x.requires_grad_(True)
    kv = None
    if S:
        kv = zero_some_channel0(T(det(pfx + "kv", (S, B, d))), pfx + "kv").requires_grad_(True)
    y = O.encoder(sd, "", O.EncCfg(H, Ly, attn_mask=mask, biprojection=bi), x, kv, kv)
    (y * T(det(pfx + "w", y.shape))).sum().backward()
    close(y, g[f"{tag}.y"], what="y")
    close(x.grad, g[f"{tag}.gx"], what="gx")
    if kv is not None:
        close(kv.grad, g[f"{tag}.gkv"], what="gkv")
    for n, p in sd.items():
        close(p.grad, g[f"{tag}.g.{n}"], what=n)


def test_f6_gmu():
    g = load("f6_gmu")
    d = 24
    sd = sd_for({"hidden1.weight": (d, d), "hidden2.weight": (d, d), "x_gate.weight": (d, 2 * d)}, "f6g.")
    x1, x2 = (T(det(f"f6g.x{i}", (5, 2, d))).requires_grad_(True) for i in (1, 2))
    y, z = O.gmu_features(sd, "", x1, x2)
    (y * T(det("f6g.w", y.shape))).sum().backward()
    close(y, g["g.y"]); close(z, g["g.z"]); close(x1.grad, g["g.gx1"]); close(x2.grad, g["g.gx2"])
    for n, p in sd.items():
        close(p.grad, g[f"g.g.{n}"], what=n)
    for n in (3, 4):
        pfx = f"f6t{n}."
        shapes = {}
        for i in range(1, n + 1):
            shapes[f"hidden{i}.weight"] = (d, d)
            shapes[f"x{i}_gate.weight"] = (d, n * d)
        sd = sd_for(shapes, pfx)
        xs = [T(det(pfx + f"x{i}", (3, d))).requires_grad_(True) for i in range(n)]
        y, z = O.text_shifting(sd, "", xs)
        (y * T(det(pfx + "w", y.shape))).sum().backward()
        close(y, g[f"t{n}.y"]); close(z, g[f"t{n}.z"])
        for i, x in enumerate(xs):
            close(x.grad, g[f"t{n}.gx{i}"])
        for k, p in sd.items():
            close(p.grad, g[f"t{n}.g.{k}"], what=k)


def _check_model(g, sd, logits, z, inputs, pfx, tol=2e-5):
    tgt = (T(det(pfx + "tgt", logits.shape)) > 0).float()
    loss = torch.nn.functional.binary_cross_entropy_with_logits(logits, tgt)
    loss.backward()
    close(logits, g["logits"], tol, "logits")
    close(z, g["z"], tol, "z")
    close(loss, g["loss"], tol, "loss")
    nograd = set(g["nograd"].tolist())
    for k, p in sd.items():
        if k in nograd:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, k
            continue
        gn = g["gn." + k]
        n = p.grad.double().norm().item()
        assert abs(n - gn[0]) <= 1e-4 * max(gn[0], 1e-6) + 1e-7, (k, n, gn[0])
        if "g." + k in g:
            close(p.grad, g["g." + k], tol, k)
    for k, t in inputs.items():
        close(t.grad, g["gin." + k], tol, "gin." + k)


def test_f7_mmtrvat_and_state_dict_abi():
    g = load("f7_mmtrvat")
    pfx = "f7."
    m = O.ModelCfg(24, 4, 2, 6, orig_d_l=32)
    shapes = O.model_param_shapes(m, False)
    # the oracle's key/shape table IS the reference's named_parameters()
    ref = dict(zip(g["param_names"].tolist(), g["param_shapes"].tolist()))
    assert {k: ",".join(map(str, v)) for k, v in shapes.items()} == ref
    sd = sd_for(shapes, pfx)
    xl, img, aud = (T(det(pfx + n, s)).requires_grad_(True) for n, s in
                    (("xl", (2, 50, 32)), ("img", (2, 500, 35)), ("aud", (2, 375, 74))))
    logits, z = O.bpmult3_forward(sd, m, xl, img, aud)
    _check_model(g, sd, logits, z, {"xl": xl, "img": img, "aud": aud}, pfx)


def test_f8_mmtrvapt():
    g = load("f8_mmtrvapt")
    pfx = "f8."
    m = O.ModelCfg(24, 4, 2, 13, orig_d_l=32, orig_d_v=40, orig_d_a=96, orig_d_p=64,
                   num_vectors_a=200, num_vectors_v=200)
    shapes = O.model_param_shapes(m, True)
    ref = dict(zip(g["param_names"].tolist(), g["param_shapes"].tolist()))
    assert {k: ",".join(map(str, v)) for k, v in shapes.items()} == ref
    sd = sd_for(shapes, pfx)
    xl, img, post = (T(det(pfx + n, s)).requires_grad_(True) for n, s in
                     (("xl", (2, 60, 32)), ("img", (2, 150, 40)), ("post", (2, 64))))
    aud = T(det(pfx + "aud", (2, 96, 1000)))
    af = O.audio_encoder(sd, aud)
    close(af, g["audio_feat"], 2e-5, "audio_feat")
    af.retain_grad()
    logits, z = O.bpmult4_forward(sd, m, xl, img, af, post)
    _check_model(g, sd, logits, z, {"xl": xl, "img": img, "post": post, "audio_feat": af}, pfx)


@pytest.mark.skipif(not os.path.exists(os.path.join(G, "f9_cfg1.npz")), reason="f9 fixture not generated")
def test_f9_cfg1_shape_logits():
    """BASELINE.json configs[0] shape (d=300, 12 heads -> head_dim 25, 8 layers,
    lengths padded to 512) at B=2: forward only here (backward is covered on
    the GPU against the stored gradient norms)."""
    g = load("f9_cfg1")
    pfx = "f9."
    m = O.ModelCfg(300, 12, 8, 6)
    sd = {k: T(det_param(pfx + k, s)) for k, s in O.model_param_shapes(m, False).items()}
    xl, img, aud = (T(det(pfx + n, s)) for n, s in
                    (("xl", (2, 20, 768)), ("img", (2, 500, 35)), ("aud", (2, 400, 74))))
    with torch.no_grad():
        logits, z = O.bpmult3_forward(sd, m, xl, img, aud)
    close(logits, g["logits"], 5e-5, "logits")
    close(z, g["z"], 5e-5, "z")


@pytest.mark.skipif(not os.path.exists(os.path.join(G, "f10_cfg3.npz")), reason="f10 fixture not generated")
def test_f10_cfg3_shape_logits():
    """BASELINE.json configs[2] dims (Moviescope 4-modal: d=768, 6 heads -> head_dim 128, 5 layers, orig_d_v=4096,
    L=512 / V=A=200, biprojection level 2) at B=1: forward only here (462 M parameters; the backward is compared on
    the GPU against the stored gradient norms and small gradients)."""
    g = load("f10_cfg3")
    pfx = "f10."
    m = O.ModelCfg(768, 6, 5, 13, orig_d_l=768, orig_d_v=4096, orig_d_a=96, orig_d_p=4096, num_vectors_a=200, num_vectors_v=200)
    shapes = O.model_param_shapes(m, True)
    ref = dict(zip(g["param_names"].tolist(), g["param_shapes"].tolist()))
    assert {k: ",".join(map(str, v)) for k, v in shapes.items()} == ref
    sd = {k: T(det_param(pfx + k, s)) for k, s in shapes.items()}
    xl, img, post = (T(det(pfx + n, s)) for n, s in (("xl", (1, 512, 768)), ("img", (1, 200, 4096)), ("post", (1, 4096))))
    aud = T(det(pfx + "aud", (1, 96, 1000)))
    with torch.no_grad():
        af = O.audio_encoder(sd, aud)
        close(af[:, ::8, ::5], g["audio_feat_s"], 2e-5, "audio_feat sample")
        logits, z = O.bpmult4_forward(sd, m, xl, img, af, post)
    close(logits, g["logits"], 1e-4, "logits")
    close(z, g["z"], 1e-4, "z")


@pytest.mark.skipif(not os.path.exists(os.path.join(G, "f11_h768.npz")), reason="f11 fixture not generated")
def test_f11_headline_shape_logits():
    """The bench's headline workload (BASELINE.json metric: hidden 768, 3-modal unaligned; mmtr.py:587-866): `mmtrvat`
    at d=768, 12 heads (head_dim 64), 8 layers, L/V/A = 20/500/400 -> 512, B=1: forward only here (680 M parameters; the
    backward is compared on the GPU against the stored gradient norms, small gradients and input gradients)."""
    g = load("f11_h768")
    pfx = "f11."
    m = O.ModelCfg(768, 12, 8, 6, orig_d_l=768)
    shapes = O.model_param_shapes(m, False)
    ref = dict(zip(g["param_names"].tolist(), g["param_shapes"].tolist()))
    assert {k: ",".join(map(str, v)) for k, v in shapes.items()} == ref
    sd = {k: T(det_param(pfx + k, s)) for k, s in shapes.items()}
    xl, img, aud = (T(det(pfx + n, s)) for n, s in (("xl", (1, 20, 768)), ("img", (1, 500, 35)), ("aud", (1, 400, 74))))
    with torch.no_grad():
        logits, z = O.bpmult3_forward(sd, m, xl, img, aud)
    close(logits, g["logits"], 1e-4, "logits")
    close(z, g["z"], 1e-4, "z")


@pytest.mark.skipif(not os.path.exists(os.path.join(G, "f12_k768.npz")), reason="f12 fixture not generated")
def test_f12_kernel_point_model():
    """The north-star kernel-point model from the reference (BASELINE.json north_star: hidden 768 / seq_len 50; bench.py
    `k768`): `mmtrvat` at d=768, 6 heads (head_dim 128), 5 layers, 50 time steps per modality (num_vectors_* = 50, set on
    the reference instance), B=2 -- logits, gates, loss and a few gradient norms through the oracle's backward."""
    g = load("f12_k768")
    pfx = "f12."
    m = O.ModelCfg(768, 6, 5, 6, orig_d_l=768, num_vectors_l=50, num_vectors_a=50, num_vectors_v=50)
    shapes = O.model_param_shapes(m, False)
    ref = dict(zip(g["param_names"].tolist(), g["param_shapes"].tolist()))
    mine = {k: ",".join(map(str, v)) for k, v in shapes.items()}
    assert mine.keys() == ref.keys()
    # (the unused time-axis maps keep the reference's source-constant 512 x 512 shapes there)
    assert {k: v for k, v in mine.items() if not k.startswith("transfm_")} == {k: v for k, v in ref.items() if not k.startswith("transfm_")}
    sd = {k: T(det_param(pfx + k, s)).requires_grad_(True) for k, s in shapes.items()}
    xl, img, aud = (T(det(pfx + n, s)) for n, s in (("xl", (2, 50, 768)), ("img", (2, 50, 35)), ("aud", (2, 50, 74))))
    logits, z = O.bpmult3_forward(sd, m, xl, img, aud)
    close(logits, g["logits"], 1e-4, "logits")
    close(z, g["z"], 1e-4, "z")
    tgt = (T(det(pfx + "tgt", tuple(logits.shape))) > 0).float()
    loss = torch.nn.functional.binary_cross_entropy_with_logits(logits, tgt)
    close(loss, g["loss"], 1e-5, "loss")
    loss.backward()
    for k in ("trans_l_with_a.layers.0.fc1.weight", "trans_v_with_l2a.layers.4.self_attn.in_proj_weight", "gmu_l.x_gate.weight",
              "proj_v.weight", "out_layer.bias"):
        n = sd[k].grad.double().norm().item()
        assert abs(n - g["gn." + k][0]) <= 1e-3 * g["gn." + k][0], (k, n, g["gn." + k][0])
