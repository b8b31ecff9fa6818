"""TransformerEncoder (HIP engine, through the nn.Module surface) against the
golden vectors produced by the reference's TransformerEncoder (f5) on the GPU.
f32 mode: <= 2e-4 (exact f32 MFMA; differences are summation order and
exp/tanh intrinsics).  bf16 mode: stated tolerance 5e-2 on O(1) outputs."""
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


def close(a, b, tol, what):
    a = a.detach().float().cpu().numpy()
    assert a.shape == b.shape, (what, a.shape, b.shape)
    assert np.isfinite(a).all(), what
    scale = max(1.0, float(np.abs(b).max()))
    err = float(np.abs(a - b).max())
    if tol >= 1e-2:
        # bf16 mode: stated tolerance is on the relative L2 error of the whole tensor
        # (max-norm of a bf16 gradient is dominated by a few cancelling entries)
        rel = float(np.linalg.norm((a - b).ravel()) / max(np.linalg.norm(b.ravel()), 1e-6))
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
    close(y, g[f"{tag}.y"], tol, "y")
    close(x.grad, g[f"{tag}.gx"], tol, "gx")
    close(kv.grad, g[f"{tag}.gkv"], tol, "gkv")
    for k, p in enc.named_parameters():
        assert p.grad is not None, k
        close(p.grad, g[f"{tag}.g.{k}"], tol, k)


def test_state_dict_keys_match_reference_layout():
    enc = TransformerEncoder(24, 4, 2, biprojection=True)
    keys = set(enc.state_dict().keys())
    for k in ("version", "embed_positions._float_tensor", "layer_norm.weight", "layers.1.self_attn.in_proj_weight",
              "layers.0.self_attn.out_proj.bias", "layers.1.fc1.weight", "layers.0.layer_norms.2.bias"):
        assert k in keys
    assert enc.state_dict()["layers.0.self_attn.in_proj_weight"].shape == (72, 24)


@pytest.mark.parametrize("bi,Tn,S", [(False, 50, 50), (False, 33, 64), (True, 40, 21)])
def test_fused_short_block_schedule_equals_separate_kernels(bi, Tn, S):
    """engine.FUSE_SHORT_BLOCKS: with T, S <= 64 and head_dim 128 (bf16) the crossmodal attention block of every layer
    runs as one bpm_xblock_fwd launch that also writes what the (unchanged) backward kernels read.  Same weights, inputs
    and dropout seed through both schedules: outputs and every gradient must agree to bf16 rounding of re-ordered sums."""
    from bpmult_amd import engine
    d, H, B, Ly = 256, 2, 3, 2
    res = []
    for fuse in (False, True):
        engine.FUSE_SHORT_BLOCKS = fuse
        try:
            torch.manual_seed(11)
            enc = TransformerEncoder(d, H, Ly, attn_dropout=0.1, relu_dropout=0.1, res_dropout=0.1, embed_dropout=0.1, attn_mask=True,
                                     biprojection=bi)
            enc.precision = "bf16"
            with torch.no_grad():
                for k, p in enc.named_parameters():
                    p.copy_(T(det_param("fz." + k, p.shape)))
            enc = enc.cuda().train()
            x = T(det("fz.x", (Tn, B, d))).cuda().requires_grad_(True)
            kv = T(det("fz.kv", (S, B, d))).cuda().requires_grad_(True)
            y = enc(x, kv, kv)
            plan = next(iter(enc._plans.values()))
            assert plan.fused_block == fuse
            (y * T(det("fz.w", tuple(y.shape))).cuda()).sum().backward()
            res.append((y.detach().float().cpu(), x.grad.float().cpu(), kv.grad.float().cpu(),
                        {k: p.grad.float().cpu() for k, p in enc.named_parameters()}))
        finally:
            engine.FUSE_SHORT_BLOCKS = False
    (y0, gx0, gk0, gp0), (y1, gx1, gk1, gp1) = res
    rel = lambda a, b: float((a - b).norm() / b.norm().clamp_min(1e-9))
    assert rel(y1, y0) < 2e-2 and rel(gx1, gx0) < 5e-2 and rel(gk1, gk0) < 5e-2, (rel(y1, y0), rel(gx1, gx0), rel(gk1, gk0))
    for k in gp0:
        assert rel(gp1[k], gp0[k]) < 8e-2, (k, rel(gp1[k], gp0[k]))
