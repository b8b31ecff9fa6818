"""TransformerEncoder (HIP engine, through the nn.Module surface) against the
golden vectors produced by the reference's TransformerEncoder (f5) on the GPU.
f32 mode: <= 2e-4 (exact f32 MFMA; differences are summation order and
exp/tanh intrinsics).  bf16 mode: per kind of tensor, <= 2x the measured errors (see `close`)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from detgen import det, det_param  # noqa: E402

import bpmult_amd  # noqa: E402
from bpmult_amd.models.encoder import TransformerEncoder  # noqa: E402

G = os.path.join(os.path.dirname(__file__), "golden")
T = torch.from_numpy


def load(name):
    return dict(np.load(os.path.join(G, name + ".npz")))


def zero_some_channel0(x, name):
    m = T(det(name + ".z", x.shape[:2])) > 1.0
    x[:, :, 0][m] = 0.0
    return x


_MEASURED = {}
_LOG = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "r04_encoder_errors.json")


def _note(tag, what, rel):
    """Largest relative-L2 error per (case, tensor kind), written to gpurun_out/ (copied to profiles/ when committed): the
    bf16 limits below are held to <= 2x these."""
    import json
    kind = what if what in ("y", "gx", "gkv") else ("layer_norm" if "layer_norm" in what else "bias" if "bias" in what else "weight")
    key = f"{tag}.{kind}"
    _MEASURED[key] = max(_MEASURED.get(key, 0.0), rel)
    try:
        os.makedirs(os.path.dirname(_LOG), exist_ok=True)
        json.dump(_MEASURED, open(_LOG, "w"), indent=1, sort_keys=True)
    except OSError:
        pass


def close(a, b, tol, what, tag=""):
    a = a.detach().float().cpu().numpy()
    assert a.shape == b.shape, (what, a.shape, b.shape)
    assert np.isfinite(a).all(), what
    scale = max(1.0, float(np.abs(b).max()))
    err = float(np.abs(a - b).max())
    if tol >= 1e-2:
        # bf16 mode: stated tolerance is on the relative L2 error of the whole tensor
        # (max-norm of a bf16 gradient is dominated by a few cancelling entries), per kind of tensor, <= 2x the errors
        # measured against the reference fixture (profiles/r04_encoder_errors.json: y 3.1e-3, gx / gkv 4.8e-2,
        # weights / biases 7.0e-2, LayerNorm affines 1.0e-1)
        rel = float(np.linalg.norm((a - b).ravel()) / max(np.linalg.norm(b.ravel()), 1e-6))
        _note(tag, what, rel)
        tol = 1e-2 if what == "y" else 1e-1 if what in ("gx", "gkv") else 2e-1 if "layer_norm" in what else 1.4e-1
        assert rel <= tol and err <= 4 * tol * scale, f"{what}: rel-L2 {rel:.3e}, max err {err:.3e} (scale {scale:.3g})"
    else:
        assert err <= tol * scale, f"{what}: max err {err:.3e} > {tol * scale:.3e}"


@pytest.mark.parametrize("prec,tol", [("f32", 2e-4), ("bf16", 2e-1)])
@pytest.mark.parametrize("tag,bi,Tn,S,mask", [("x", False, 7, 5, True), ("xn", False, 6, 6, False), ("x25", False, 8, 11, True),
                                             ("b", True, 5, 8, True)])
def test_f5_encoder(prec, tol, tag, bi, Tn, S, mask):
    g = load("f5_encoder")
    d, H = (50, 2) if tag == "x25" else (24, 4)
    B, Ly = 2, 2
    pfx = f"f5{tag}."
    enc = TransformerEncoder(d, H, Ly, attn_mask=mask, biprojection=bi)
    enc.precision = prec
    with torch.no_grad():
        for k, p in enc.named_parameters():
            p.copy_(T(det_param(pfx + k, p.shape)))
    enc = enc.cuda().train()
    x = zero_some_channel0(T(det(pfx + "x", (Tn, B, d))), pfx + "x")
    x[-2:] = 0.0
    x = x.cuda().requires_grad_(True)
    kv = zero_some_channel0(T(det(pfx + "kv", (S, B, d))), pfx + "kv").cuda().requires_grad_(True)
    y = enc(x, kv, kv)
    w = T(det(pfx + "w", tuple(y.shape))).cuda()
    (y * w).sum().backward()
    close(y, g[f"{tag}.y"], tol, "y", tag)
    close(x.grad, g[f"{tag}.gx"], tol, "gx", tag)
    close(kv.grad, g[f"{tag}.gkv"], tol, "gkv", tag)
    for k, p in enc.named_parameters():
        assert p.grad is not None, k
        close(p.grad, g[f"{tag}.g.{k}"], tol, k, tag)


def test_state_dict_keys_match_reference_layout():
    enc = TransformerEncoder(24, 4, 2, biprojection=True)
    keys = set(enc.state_dict().keys())
    for k in ("version", "embed_positions._float_tensor", "layer_norm.weight", "layers.1.self_attn.in_proj_weight",
              "layers.0.self_attn.out_proj.bias", "layers.1.fc1.weight", "layers.0.layer_norms.2.bias"):
        assert k in keys
    assert enc.state_dict()["layers.0.self_attn.in_proj_weight"].shape == (72, 24)

