"""Whole-model parity on the MI355X against golden vectors produced by the
reference models (f7: mmtrvat toy, f8: mmtrvapt toy, f9: BASELINE configs[0]
shape d=300 / 12 heads / 8 layers at B=2).  Inputs and weights are regenerated
from names (tests/golden/detgen.py); the fixtures hold reference outputs only.

Tolerances: f32 mode -- logits/gates 1e-4 abs on O(1) values (north-star bar is
1e-3 relative), gradients 2e-3 relative to the tensor's max; bf16 mode -- <= 2x
the errors measured against the same fixtures (profiles/r03_parity_errors.json)."""
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from detgen import det, det_param  # noqa: E402

import bpmult_amd  # noqa: E402
from bpmult_amd.models import get_model  # noqa: E402

G = os.path.join(os.path.dirname(__file__), "golden")
T = torch.from_numpy


def load(name):
    return dict(np.load(os.path.join(G, name + ".npz")))


def args_for(model, **kw):
    a = dict(model=model, orig_d_l=768, orig_d_v=35, orig_d_a=74, orig_d_p=4096, hidden_sz=300, vonly=True, lonly=True,
             aonly=True, num_heads=12, layers=8, attn_dropout=0., attn_dropout_v=0., attn_dropout_a=0., relu_dropout=0.,
             res_dropout=0., out_dropout=0., embed_dropout=0., attn_mask=True, hybrid=False, n_classes=6,
             bert_model="unused", text_features=True)
    a.update(kw)
    return SimpleNamespace(**a)


def err(a, b):
    a = a.detach().float().cpu().numpy()
    assert a.shape == b.shape, (a.shape, b.shape)
    assert np.isfinite(a).all()
    return float(np.abs(a - b).max()), max(float(np.abs(b).max()), 1e-6), \
        float(np.linalg.norm((a - b).ravel()) / max(np.linalg.norm(b.ravel()), 1e-12))


# bf16-mode tolerances, held to <= 2x what profiles/r03_parity_errors.json records against the reference fixtures
# (relative L2 of the whole tensor).  Measured: logits 0.8-1.15e-2 (F7-F11; max-abs / max-abs 1.0-1.6e-2), gates 2.7-4.5e-3,
# loss <= 3.2e-3; worst per-tensor gradient 0.15-0.17 at the BASELINE shapes (F9, F11: a LayerNorm gain / in_proj bias
# of a level-2 encoder, 1e-3-sized sums of cancelling terms), 0.25 on the d=24 toys; worst gradient NORM 2.3e-2 (F11),
# 6.3e-2 (F9), 9e-2 (F10, five biprojection layers).  The f32 mode is the one held to the north star's 1e-3
# (measured 4-6e-6 on logits); the reference itself moves 1.1e-2 under bf16 autocast (hard parts).
BF16_FWD, BF16_GRAD, BF16_GRAD_TOY = 2.5e-2, 3e-1, 4e-1
BF16_GNORM_BIG = 5e-2


# bf16x3 mode (fp32 storage, the large linear-layer products as three split-bf16 MFMA products): the north star's bar --
# logits / gates / loss within 1e-3 of the reference (measured ~1e-5: profiles/r04_parity_errors.json), gradients 5e-3
X3_FWD, X3_GRAD = 1e-3, 5e-3


def check(a, b, prec, what, f32_rel=2e-3, f32_abs=None, bf16_rel=BF16_FWD):
    e, scale, rel = err(a, b)
    if prec == "bf16x3":
        lim = (X3_FWD if f32_abs is not None else X3_GRAD) * max(scale, 1e-3)
        assert e <= lim, f"{what}: max err {e:.3e} > {lim:.3e} (scale {scale:.3g})"
    elif prec == "f32":
        lim = f32_abs if f32_abs is not None else f32_rel * max(scale, 1e-3)
        assert e <= lim, f"{what}: max err {e:.3e} > {lim:.3e} (scale {scale:.3g})"
    else:
        assert rel <= bf16_rel, f"{what}: rel-L2 {rel:.3e} (max err {e:.3e}, scale {scale:.3g})"


# measured errors of every whole-model comparison against the reference fixtures, written to
# gpurun_out/r04_parity_errors.json (copied to profiles/ when committed): the tolerances above are held to <= 2x these
PARITY_LOG = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "r04_parity_errors.json")
_PARITY = {}


def _record(tag, prec, rec):
    import json
    _PARITY[f"{tag}{prec}"] = rec
    try:
        os.makedirs(os.path.dirname(PARITY_LOG), exist_ok=True)
        old = {}
        if os.path.exists(PARITY_LOG):
            with open(PARITY_LOG) as f:
                old = json.load(f)
        old.update(_PARITY)
        with open(PARITY_LOG, "w") as f:
            json.dump(old, f, indent=1, sort_keys=True)
    except OSError:
        pass


SCHEDULES = [pytest.param(True, id="pruned"), pytest.param(False, id="dense")]


def run_model(g, model, pfx, inputs, call, prec, BF16_GRAD=BF16_GRAD_TOY, BF16_GNORM=1e-1, prune=False):
    """Both schedules are held to the SAME reference fixture and the same limits: `prune` = exact dead-row elimination
    (the library's default), dense = the reference's own schedule."""
    model.set_prune_unused_rows(prune)
    with torch.no_grad():
        for k, p in model.named_parameters():
            p.copy_(T(det_param(pfx + k, p.shape)))
    model.precision = prec
    model = model.cuda().train()
    dev = {k: v.cuda().requires_grad_(True) for k, v in inputs.items()}
    logits, z = call(model, dev)
    tgt = (T(det(pfx + "tgt", tuple(logits.shape))) > 0).float().cuda()
    loss = torch.nn.functional.binary_cross_entropy_with_logits(logits, tgt)
    loss.backward()
    rec = {}
    for nm, t in (("logits", logits), ("z", z), ("loss", loss)):
        e, scale, rel = err(t, g[nm])
        rec[nm] = {"max_abs": e, "max_abs_over_max_abs": e / scale, "rel_l2": rel}
    worst_g = (0.0, "")
    nograd = set(g["nograd"].tolist())
    worst = (0.0, "")
    for k, p in model.named_parameters():
        if k in nograd:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, k
            continue
        assert p.grad is not None, k
        gn = g["gn." + k]
        n = p.grad.double().norm().item()
        rel = abs(n - gn[0]) / max(gn[0], 1e-9)
        worst = max(worst, (rel, k))
        if "g." + k in g:
            worst_g = max(worst_g, (err(p.grad, g["g." + k])[2], k))
    for k, t in dev.items():
        if "gin." + k in g:
            worst_g = max(worst_g, (err(t.grad, g["gin." + k])[2], "gin." + k))
    rec["worst_grad_norm_rel"] = {"value": worst[0], "param": worst[1]}
    rec["worst_grad_rel_l2"] = {"value": worst_g[0], "tensor": worst_g[1]}
    _record(pfx + ("pruned." if prune else "dense."), prec, rec)
    assert model._trunks[logits.shape[0]].prune == prune
    check(logits, g["logits"], prec, "logits", f32_abs=1e-4 * max(1.0, float(np.abs(g["logits"]).max())))
    check(z, g["z"], prec, "z", f32_abs=1e-4)
    check(loss, g["loss"], prec, "loss", f32_abs=1e-4)
    for k, p in model.named_parameters():
        if k in nograd:
            continue
        gn = g["gn." + k]
        n = p.grad.double().norm().item()
        rel = abs(n - gn[0]) / max(gn[0], 1e-9)
        assert rel <= (5e-3 if prec in ("f32", "bf16x3") else BF16_GNORM), f"grad norm of {k}: {n:.6e} vs reference {gn[0]:.6e}"
        if "g." + k in g:
            check(p.grad, g["g." + k], prec, "grad " + k, bf16_rel=BF16_GRAD)
    for k, t in dev.items():
        if "gin." + k in g:
            check(t.grad, g["gin." + k], prec, "gin." + k, bf16_rel=BF16_GRAD)
    return worst


@pytest.mark.parametrize("prune", SCHEDULES)
@pytest.mark.parametrize("prec", ["f32", "bf16"])
def test_f7_mmtrvat(prec, prune):
    g = load("f7_mmtrvat")
    model = get_model(args_for("mmtrvat", hidden_sz=24, num_heads=4, layers=2, orig_d_l=32))
    assert sorted(k for k, _ in model.named_parameters()) == sorted(g["param_names"].tolist())
    inputs = {"xl": T(det("f7.xl", (2, 50, 32))), "img": T(det("f7.img", (2, 500, 35))), "aud": T(det("f7.aud", (2, 375, 74)))}
    run_model(g, model, "f7.", inputs, lambda m, d: m(d["xl"], None, None, d["img"], d["aud"], output_gate=True), prec,
              prune=prune)


@pytest.mark.parametrize("prune", SCHEDULES)
@pytest.mark.parametrize("prec", ["f32", "bf16"])
def test_f8_mmtrvapt(prec, prune):
    g = load("f8_mmtrvapt")
    model = get_model(args_for("mmtrvapt", hidden_sz=24, num_heads=4, layers=2, orig_d_l=32, orig_d_v=40, orig_d_a=96,
                               orig_d_p=64, n_classes=13))
    assert sorted(k for k, _ in model.named_parameters()) == sorted(g["param_names"].tolist())
    inputs = {"xl": T(det("f8.xl", (2, 60, 32))), "img": T(det("f8.img", (2, 150, 40))),
              "aud": T(det("f8.aud", (2, 96, 1000))), "post": T(det("f8.post", (2, 64)))}
    run_model(g, model, "f8.", inputs,
              lambda m, d: m(d["xl"], None, None, d["img"], d["aud"], d["post"], output_gate=True), prec, prune=prune)


@pytest.mark.parametrize("prune", SCHEDULES)
@pytest.mark.parametrize("prec", ["f32", "bf16", "bf16x3"])
def test_f9_cfg1_shape(prec, prune):
    """d=300, 12 heads (head_dim 25), 8 layers, lengths padded to 512, B=2."""
    g = load("f9_cfg1")
    model = get_model(args_for("mmtrvat"))
    inputs = {"xl": T(det("f9.xl", (2, 20, 768))), "img": T(det("f9.img", (2, 500, 35))), "aud": T(det("f9.aud", (2, 400, 74)))}
    run_model(g, model, "f9.", inputs, lambda m, d: m(d["xl"], None, None, d["img"], d["aud"], output_gate=True), prec,
              BF16_GRAD=BF16_GRAD, prune=prune)


@pytest.mark.skipif(not os.path.exists(os.path.join(G, "f10_cfg3.npz")), reason="f10 fixture not generated")
@pytest.mark.parametrize("prune", SCHEDULES)
@pytest.mark.parametrize("prec", ["f32", "bf16"])        # (bf16x3: its split products need >= 256 rows per problem -- F9 / F11)
def test_f10_cfg3_shape(prec, prune):
    """BASELINE.json configs[2] dims: 4-modal mmtrvapt, d=768, 6 heads (head_dim 128: the multi-tile attention forward /
    dQ / dK-dV kernels at T=512 / S=200 and 200 / 512), 5 layers, orig_d_v=4096 (K=4096 projection GEMM), biprojection
    level 2, time-axis maps, B=1.  Logits, gates, loss, every parameter's gradient norm and the small gradients."""
    g = load("f10_cfg3")
    model = get_model(args_for("mmtrvapt", hidden_sz=768, num_heads=6, layers=5, orig_d_l=768, orig_d_v=4096, orig_d_a=96,
                               orig_d_p=4096, n_classes=13))
    assert sorted(k for k, _ in model.named_parameters()) == sorted(g["param_names"].tolist())
    inputs = {"xl": T(det("f10.xl", (1, 512, 768))), "img": T(det("f10.img", (1, 200, 4096))),
              "aud": T(det("f10.aud", (1, 96, 1000))), "post": T(det("f10.post", (1, 4096)))}
    dev = {}

    def call(m, d):
        dev.update(d)
        return m(d["xl"], None, None, d["img"], d["aud"], d["post"], output_gate=True)

    # bf16 mode, stated: gradient norms within 1.5e-1 here (five biprojection layers at head_dim 128: measured worst
    # 1.1e-1 on one level-2 in_proj_weight); the f32 mode holds every norm to 5e-3
    run_model(g, model, "f10.", inputs, call, prec, BF16_GRAD=1.5e-1, BF16_GNORM=1.5e-1, prune=prune)
    for k in ("xl", "img", "aud"):                      # big inputs: gradient norms only
        n = dev[k].grad.double().norm().item()
        ref = float(g["ginn." + k][0])
        assert abs(n - ref) <= (5e-3 if prec in ("f32", "bf16x3") else 1e-1) * ref, (k, n, ref)


@pytest.mark.skipif(not os.path.exists(os.path.join(G, "f11_h768.npz")), reason="f11 fixture not generated")
@pytest.mark.parametrize("prune", SCHEDULES)
@pytest.mark.parametrize("prec", ["f32", "bf16", "bf16x3"])
def test_f11_headline(prec, prune):
    """The bench's headline workload against the reference itself (mmtr.py:587-866): 3-modal mmtrvat, d=768, 12 heads
    (head_dim 64), 8 layers, orig_d 768/35/74, L/V/A = 20/500/400 -> 512, B=1.  Logits, gates, loss, every parameter's
    gradient norm, the small gradients and the input gradients."""
    g = load("f11_h768")
    model = get_model(args_for("mmtrvat", hidden_sz=768, num_heads=12, layers=8))
    assert sorted(k for k, _ in model.named_parameters()) == sorted(g["param_names"].tolist())
    inputs = {"xl": T(det("f11.xl", (1, 20, 768))), "img": T(det("f11.img", (1, 500, 35))), "aud": T(det("f11.aud", (1, 400, 74)))}
    run_model(g, model, "f11.", inputs, lambda m, d: m(d["xl"], None, None, d["img"], d["aud"], output_gate=True), prec,
              BF16_GRAD=BF16_GRAD, BF16_GNORM=BF16_GNORM_BIG, prune=prune)


@pytest.mark.skipif(not os.path.exists(os.path.join(G, "f12_k768.npz")), reason="f12 fixture not generated")
@pytest.mark.parametrize("prune", SCHEDULES)
@pytest.mark.parametrize("prec", ["f32", "bf16"])        # (bf16x3: its split products need >= 256 rows per problem -- F9 / F11)
def test_f12_kernel_point_model(prec, prune):
    """The north-star kernel-point model against the reference itself (BASELINE.json north_star: hidden 768 / seq_len 50;
    bench.py `k768` and `kernel_point`): `mmtrvat` at d=768, 6 heads (head_dim 128), 5 layers, T = S = 50 in all twelve
    encoders (num_vectors_* = 50: single-tile attention, strictly causal masks), B=2."""
    g = load("f12_k768")
    model = get_model(args_for("mmtrvat", hidden_sz=768, num_heads=6, layers=5, num_vectors_l=50, num_vectors_a=50,
                               num_vectors_v=50))
    assert sorted(k for k, _ in model.named_parameters()) == sorted(g["param_names"].tolist())
    inputs = {"xl": T(det("f12.xl", (2, 50, 768))), "img": T(det("f12.img", (2, 50, 35))), "aud": T(det("f12.aud", (2, 50, 74)))}
    run_model(g, model, "f12.", inputs, lambda m, d: m(d["xl"], None, None, d["img"], d["aud"], output_gate=True), prec,
              BF16_GRAD=BF16_GRAD, BF16_GNORM=1e-1, prune=prune)


def test_fused_adam_matches_torch_adam():
    """bpmult_amd.optim.FusedAdam (one kernel over the flat trunk buffers) == torch.optim.Adam, three steps,
    including the hand-off through refreshed weight shadows (the loss sequence must match too)."""
    import copy
    from bpmult_amd.optim import FusedAdam
    torch.manual_seed(7)
    a = args_for("mmtrvat", hidden_sz=24, num_heads=4, layers=2, orig_d_l=32)
    m1 = get_model(a)
    m2 = copy.deepcopy(m1)
    m1.precision = m2.precision = "f32"
    m1, m2 = m1.cuda().train(), m2.cuda().train()
    x = [torch.randn(2, 50, 32, device="cuda"), torch.randn(2, 500, 35, device="cuda"), torch.randn(2, 375, 74, device="cuda")]
    tgt = (torch.randn(2, 6, device="cuda") > 0).float()
    o1 = torch.optim.Adam(m1.parameters(), lr=1e-2, weight_decay=0.0)
    o2 = FusedAdam(m2, lr=1e-2)
    for it in range(3):
        losses = []
        for m, o in ((m1, o1), (m2, o2)):
            o.zero_grad()
            loss = torch.nn.functional.binary_cross_entropy_with_logits(m(x[0], None, None, x[1], x[2]), tgt)
            loss.backward()
            o.step()
            losses.append(float(loss.detach()))
        assert abs(losses[0] - losses[1]) <= 1e-5 * max(1.0, abs(losses[0])), (it, losses)
    # Adam normalises every element's step to ~lr, so elements whose gradient is rounding noise (atomics order
    # differs between the two runs) may move differently: bound by a fraction of the 3 * lr they can travel
    worst = 0.0
    for (k, p1), (_, p2) in zip(m1.named_parameters(), m2.named_parameters()):
        worst = max(worst, float((p1.detach() - p2.detach()).abs().max()))
    assert worst <= 3e-3, worst


@pytest.mark.parametrize("prec,d", [("bf16", 64), ("bf16", 40), ("f32", 40)])
@pytest.mark.parametrize("four", [False, True])
def test_fused_adam_writes_the_weight_shadows(prec, d, four):
    """The optimizer kernel stores the CT shadows of the plain weight matrices as it stores the updated masters; the small
    rest (K / V weights with the LayerNorm gain folded in, x_gate halves, folded biases) follows at the next forward's
    refresh.  Together they must be bit-equal to a full re-derivation from the masters (pack_weights + fold_bias), with
    and without column padding (d = 40: leading dimension 64), for both models; and the next forward uses them."""
    from bpmult_amd.optim import FusedAdam
    torch.manual_seed(5)
    kw = dict(hidden_sz=d, num_heads=4, layers=2, orig_d_l=32, num_vectors_l=48, num_vectors_a=48, num_vectors_v=48)
    if four:
        m = get_model(args_for("mmtrvapt", orig_d_v=40, orig_d_a=96, orig_d_p=64, n_classes=13, **kw))
        m.audio_enc.conv_layers[2] = torch.nn.AdaptiveAvgPool1d(48)
        x = [torch.randn(2, 17, 32), torch.randn(2, 30, 40), torch.randn(2, 96, 600), torch.randn(2, 64)]
    else:
        m = get_model(args_for("mmtrvat", **kw))
        x = [torch.randn(2, 17, 32), torch.randn(2, 48, 35), torch.randn(2, 31, 74)]
    m.precision = prec
    m = m.cuda().train()
    x = [t.cuda() for t in x]
    opt = FusedAdam(m, lr=1e-2, weight_decay=0.01)
    for it in range(2):
        opt.zero_grad()
        m(x[0], None, None, *x[1:]).square().mean().backward()
        opt.step()
        st = m._store
        assert st._dirty_rest and not st._dirty and len(st._adam_plain) > 50
        st.refresh_shadows()                                    # the rest pass only
        got, gotf = st.shadow_flat.clone(), st.fold_flat.clone()
        st.refresh_shadows(force=True)                          # full re-derivation from the masters
        assert torch.equal(got, st.shadow_flat), f"step {it}: optimizer-written shadows differ from pack_weights"
        assert torch.equal(gotf, st.fold_flat)
    # a write to a parameter between steps still forces the full refresh (version counters), and the optimizer then
    # rewrites every plain shadow from the updated master
    with torch.no_grad():
        m.trans_l_with_a.layers[0].fc1.weight.mul_(0.5)
    opt.zero_grad()
    m(x[0], None, None, *x[1:]).square().mean().backward()
    opt.step()
    m._store.refresh_shadows()
    got = m._store.shadow_flat.clone()
    m._store.refresh_shadows(force=True)
    assert torch.equal(got, m._store.shadow_flat)


def test_fused_adam_kernel_exact():
    """bpm_adam_step on given gradients == torch.optim.Adam on the same gradients (weight decay, grad_scale, fused zero)."""
    from bpmult_amd import _lib
    n = 4096 + 64
    g = torch.Generator().manual_seed(3)
    p0 = torch.randn(n, generator=g)
    grads = [torch.randn(n, generator=g) * (10.0 ** float(torch.randint(-6, 1, (1,), generator=g))) for _ in range(4)]
    ref = p0.clone().cuda().requires_grad_(True)
    opt = torch.optim.Adam([ref], lr=3e-3, betas=(0.9, 0.98), eps=1e-8, weight_decay=0.01)
    p = p0.clone().cuda()
    m, v = torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    for it, gr in enumerate(grads, 1):
        ref.grad = (gr * 0.5).cuda()
        opt.step()
        gd = gr.clone().cuda()
        _lib.check(_lib.lib().bpm_adam_step(p.data_ptr(), gd.data_ptr(), m.data_ptr(), v.data_ptr(), n, 3e-3, 0.9, 0.98, 1e-8,
                                            0.01, it, 0.5, 1, torch.cuda.current_stream().cuda_stream), "bpm_adam_step")
        assert float(gd.abs().max()) == 0.0
        d = float((p - ref.detach()).abs().max())
        assert d <= 2e-6, (it, d)


@pytest.mark.parametrize("prec,B", [("f32", 3), ("bf16", 2), ("bf16x3", 2)])
def test_low_rank_key_side_equals_the_dk_dv_route(prec, B):
    """Level 2 of the pruned 3-modal model, TRAINING mode with every dropout on (attention dropout included: the folded
    value-bias gradient then carries rowsum(Pd) != 1): the low-rank key side (dS / Pd from the dQ pass, bpm_expand_heads,
    batched products; engine.EncoderGroupPlan._lowrank) against the dK / dV route on the same weights, inputs and seeds --
    the same dropout masks, so the two may differ by rounding only.  Key lengths that are not whole 64-key tiles (96)."""
    import copy
    from bpmult_amd import engine
    torch.manual_seed(5)
    a = args_for("mmtrvat", hidden_sz=48, num_heads=4, layers=3, orig_d_l=32, num_vectors_l=96, num_vectors_a=96,
                 num_vectors_v=96, attn_dropout=0.2, attn_dropout_a=0.1, attn_dropout_v=0.15)
    xs = [torch.randn(B, 40, 32), torch.randn(B, 96, 35), torch.randn(B, 77, 74)]
    m1 = get_model(a)
    with torch.no_grad():
        for p in m1.parameters():
            if p.dim() == 1:
                p.add_(0.1 * torch.randn_like(p))
    m2 = copy.deepcopy(m1)
    tgt = (torch.randn(B, a.n_classes) > 0).float().cuda()
    outs = []
    keep = engine._LOWRANK
    try:
        for m, lowrank in ((m1, False), (m2, True)):
            engine._LOWRANK = lowrank                    # read when the trunk's launch tables are built (first forward)
            m.precision = prec
            m = m.cuda().train()
            m.set_prune_unused_rows(True)
            x = [t.clone().cuda().requires_grad_(True) for t in xs]
            logits, z = m(x[0], None, None, *x[1:], output_gate=True)
            assert m._trunks[B].plan2._lowrank == lowrank and not m._trunks[B].plan1._lowrank
            torch.nn.functional.binary_cross_entropy_with_logits(logits, tgt).backward()
            outs.append((logits.detach(), z.detach(), {k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None},
                         [t.grad.detach().clone() for t in x]))
    finally:
        engine._LOWRANK = keep
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]), "forward is the same launch sequence"
    tol_g = {"f32": 5e-6, "bf16x3": 5e-6, "bf16": 1.5e-2}[prec]      # measured 5.7e-7 / 5.2e-7 / 4.5e-3 (profiles/r04_parity_errors.json)
    rel = lambda u, v: float((u.double() - v.double()).norm() / max(float(v.double().norm()), 1e-12))
    worst = ("", 0.0)
    for k in outs[0][2]:
        g0, g1 = outs[0][2][k], outs[1][2][k]
        if float(g0.norm()) < 1e-9:                      # (the key-projection bias: rounding noise on one route, zero on the other)
            assert float(g1.norm()) < 1e-6, k
            continue
        r = rel(g1, g0)
        worst = max(worst, (k, r), key=lambda t: t[1])
        assert r <= tol_g, (k, r)
    for g0, g1 in zip(outs[0][3], outs[1][3]):
        assert rel(g1, g0) <= tol_g
    _record("lowrank_vs_dkdv/", prec, {"worst_gradient": worst[0], "rel_l2": worst[1]})


@pytest.mark.parametrize("model", ["mmtrvat", "mmtrvapt", "mmtrvapt_1layer"])
@pytest.mark.parametrize("prec", ["f32", "bf16"])
def test_pruned_schedule_equals_dense_schedule(prec, model):
    """SURVEY A.10 on random weights with non-trivial LayerNorm affines (the fixtures' are the reference's initial 1 / 0):
    the pruned schedule -- 3-modal: level-2 encoders and Fusion-GMUs on query rows {0, N-1}; 4-modal: the last
    biprojection layer's query side, the Fusion-GMUs and the time-map outputs on those rows -- gives the same logits,
    gates and gradients (parameters and inputs) as the dense schedule (dropout off: the two draw different masks)."""
    import copy
    torch.manual_seed(11)
    four = model.startswith("mmtrvapt")
    kw = dict(hidden_sz=48, num_heads=4, layers=1 if model.endswith("1layer") else 3, orig_d_l=32)
    if four:
        a = args_for("mmtrvapt", orig_d_v=40, orig_d_a=96, orig_d_p=64, n_classes=13, num_vectors_l=96, num_vectors_a=56,
                     num_vectors_v=56, **kw)
        xs = [torch.randn(2, 40, 32), torch.randn(2, 50, 40), torch.randn(2, 96, 700), torch.randn(2, 64)]
    else:
        a = args_for("mmtrvat", num_vectors_l=96, num_vectors_a=96, num_vectors_v=96, **kw)
        xs = [torch.randn(2, 40, 32), torch.randn(2, 96, 35), torch.randn(2, 77, 74)]
    m1 = get_model(a)
    if four:
        m1.audio_enc.conv_layers[2] = torch.nn.AdaptiveAvgPool1d(56)
    with torch.no_grad():
        for p in m1.parameters():
            if p.dim() == 1:
                p.add_(0.1 * torch.randn_like(p))           # non-trivial LayerNorm gains / biases
    m2 = copy.deepcopy(m1)
    m1.precision = m2.precision = prec
    m1, m2 = m1.cuda().train(), m2.cuda().train()
    m1.set_prune_unused_rows(False)
    m2.set_prune_unused_rows(True)
    tgt = (torch.randn(2, a.n_classes) > 0).float().cuda()
    outs = []
    for m in (m1, m2):
        x = [t.clone().cuda().requires_grad_(True) for t in xs]
        logits, z = m(x[0], None, None, *x[1:], output_gate=True)
        loss = torch.nn.functional.binary_cross_entropy_with_logits(logits, tgt)
        loss.backward()
        outs.append((logits.detach(), z.detach(), {k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None},
                     [t.grad.detach().clone() for t in x]))
    assert m2._trunks[2].prune and not m1._trunks[2].prune
    tol_o, tol_g = (2e-5, 2e-4) if prec == "f32" else (3e-2, 2.5e-1)
    rel = lambda u, v: float((u.double() - v.double()).norm() / max(float(v.double().norm()), 1e-12))
    assert rel(outs[1][0], outs[0][0]) <= tol_o and rel(outs[1][1], outs[0][1]) <= tol_o
    assert outs[0][2].keys() == outs[1][2].keys()
    for k in outs[0][2]:
        g0, g1 = outs[0][2][k], outs[1][2][k]
        if float(g0.norm()) < 1e-10:
            assert float(g1.norm()) < 1e-8, k
            continue
        assert rel(g1, g0) <= tol_g, (k, rel(g1, g0))
    for g0, g1 in zip(outs[0][3], outs[1][3]):
        assert rel(g1, g0) <= tol_g


def test_stale_backward_raises():
    m = get_model(args_for("mmtrvat", hidden_sz=24, num_heads=4, layers=1, orig_d_l=32, num_vectors_l=32, num_vectors_a=32,
                           num_vectors_v=32)).cuda().train()
    x = [torch.randn(2, 10, 32, device="cuda"), torch.randn(2, 20, 35, device="cuda"), torch.randn(2, 30, 74, device="cuda")]
    l1 = m(x[0], None, None, x[1], x[2]).sum()
    l2 = m(x[0], None, None, x[1], x[2]).sum()
    with pytest.raises(RuntimeError, match="one step at a time"):
        l1.backward()
    l2.backward()


def _toy(seed=3, **kw):
    torch.manual_seed(seed)
    a = args_for("mmtrvat", hidden_sz=24, num_heads=4, layers=2, orig_d_l=32, num_vectors_l=48, num_vectors_a=48,
                 num_vectors_v=48, **kw)
    m = get_model(a)
    with torch.no_grad():
        for p in m.parameters():
            if p.dim() == 1:
                p.add_(0.1 * torch.randn_like(p))
    m.precision = "f32"
    return m


def _toy_inputs(B=2, seed=4):
    g = torch.Generator().manual_seed(seed)
    return [torch.randn(B, 17, 32, generator=g).cuda(), torch.randn(B, 48, 35, generator=g).cuda(), torch.randn(B, 31, 74, generator=g).cuda()]


def test_eval_mode_no_grad_and_state_dict_round_trip():
    """eval() under no_grad (train.py:174) == train() with all dropout rates 0; a state_dict round trip into a fresh
    model (the reference checkpoint format) reproduces the logits; .to() after a forward rebuilds the flat store."""
    m = _toy().cuda()
    x = _toy_inputs()
    m.train()
    ref = m(x[0], None, None, x[1], x[2]).detach()
    m.eval()
    with torch.no_grad():
        out = m(x[0], None, None, x[1], x[2])
    assert float((out - ref).detach().abs().max()) <= 1e-5
    sd = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    m2 = _toy(seed=99)
    m2.load_state_dict(sd)
    m2 = m2.cuda().eval()
    with torch.no_grad():
        out2 = m2(x[0], None, None, x[1], x[2])
    assert float((out2 - ref).abs().max()) <= 1e-5
    m2 = m2.cpu().cuda()                                    # storages replaced: the store must be rebuilt
    with torch.no_grad():
        out3 = m2(x[0], None, None, x[1], x[2])
    assert float((out3 - ref).abs().max()) <= 1e-5


@pytest.mark.skipif(os.environ.get("BPMULT_GRAPH", "1") == "0", reason="graph replay switched off by BPMULT_GRAPH=0")
@pytest.mark.parametrize("prec", ["f32", "bf16"])
def test_graph_replay_equals_eager_launches(prec):
    """The step captured into hipGraphs (forward graph + backward graph per key, dropout seed read from device memory at
    execution time) gives bit-for-bit the logits, gates and gradients of the eager launch sequence: same dropout masks
    (README rates on), fresh and accumulating micro-steps, a second input shape (its own key), eval mode."""
    import copy
    drop = dict(attn_dropout=0.1, relu_dropout=0.1, res_dropout=0.1, embed_dropout=0.25, out_dropout=0.1)
    m1 = _toy(**drop)
    m1.precision = prec
    m2 = copy.deepcopy(m1)
    m1, m2 = m1.cuda().train(), m2.cuda().train()
    m1.use_graphs, m2.use_graphs = False, True
    tgt = (torch.randn(2, 6, generator=torch.Generator().manual_seed(1)) > 0).float().cuda()
    lossf = torch.nn.functional.binary_cross_entropy_with_logits
    xa, xb = _toy_inputs(seed=4), _toy_inputs(seed=5)
    xb[1] = xb[1][:, :40].contiguous()                          # another video length: another graph key
    plan = [(xa, True), (xa, True), (xa, True), (xa, False), (xb, True), (xb, True), (xb, True), (xa, True), (xb, False)]
    for step, (x, clear) in enumerate(plan):
        outs = []
        for m in (m1, m2):
            if clear:
                for p in m.parameters():
                    p.grad = None
            xs = [t.clone().requires_grad_(True) for t in x]
            logits, z = m(xs[0], None, None, xs[1], xs[2], output_gate=True)
            lossf(logits, tgt).backward()
            outs.append((logits.detach().clone(), z.detach().clone(), [t.grad.clone() for t in xs],
                         {k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None}))
        a, b = outs
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]), f"step {step}: logits / gates differ"
        for u, v in zip(a[2], b[2]):
            assert torch.equal(u, v), f"step {step}: input gradients differ"
        assert a[3].keys() == b[3].keys()
        for k in a[3]:
            if "gmu." in k or k.startswith(("proj1", "proj2", "out_layer")) or "layer_norm" in k or "bias" in k:
                # float atomics / two-stream arrival order in the tail and the column sums: same values up to rounding
                assert float((a[3][k] - b[3][k]).abs().max()) <= 1e-5 * max(1.0, float(a[3][k].abs().max())), (step, k)
            else:
                assert torch.equal(a[3][k], b[3][k]), f"step {step}: gradient of {k} differs"
    t2 = m2._trunks[2]
    assert len(t2._fg) >= 2 and sum("graph" in e for e in t2._fg.values()) >= 2, "both input shapes were captured"
    assert any("graph" in e for e in t2._bg.values())
    assert getattr(m1._trunks[2], "_fg", None) in (None, {})
    m1.eval(), m2.eval()
    with torch.no_grad():
        for _ in range(4):
            assert torch.equal(m1(xa[0], None, None, xa[1], xa[2]), m2(xa[0], None, None, xa[1], xa[2]))


@pytest.mark.skipif(os.environ.get("BPMULT_GRAPH", "1") == "0", reason="graph replay switched off by BPMULT_GRAPH=0")
def test_graph_cache_is_bounded_under_varying_lengths():
    """The reference collate pads text to the batch's longest sentence and trims audio to its shortest clip
    (data/helpers.py:83-102): training sees many (L, V, A) shapes.  At most MAX_GRAPHS of them are ever captured (with their
    backward graphs); the others run as eager launches, device memory stops growing, and results stay equal to the eager
    launch sequence throughout.  (No eviction: destroying captured graphs crashes later launches on this stack,
    tools/graph_cache_probe.py.)"""
    import copy
    m1 = _toy()
    m2 = copy.deepcopy(m1)
    m1, m2 = m1.cuda().train(), m2.cuda().train()
    m1.use_graphs, m2.use_graphs = False, True
    tgt = (torch.randn(2, 6, generator=torch.Generator().manual_seed(1)) > 0).float().cuda()
    lossf = torch.nn.functional.binary_cross_entropy_with_logits
    base = _toy_inputs(seed=4)
    lengths = [10, 11, 12, 13, 14, 15, 16]

    def step(m, L):
        for p in m.parameters():
            p.grad = None
        x = [base[0][:, :L].contiguous(), base[1], base[2]]
        out = m(x[0], None, None, x[1], x[2])
        lossf(out, tgt).backward()
        return out.detach().clone(), m.trans_l_with_a.layers[0].fc1.weight.grad.detach().clone()

    step(m2, lengths[0])
    t2 = m2._trunks[2]
    t2.MAX_GRAPHS = 3
    mem = []
    for rnd_ in range(3):
        for L in lengths:
            for _ in range(4):                      # 4 calls per shape: the first MAX_GRAPHS shapes are captured on their 3rd
                a, b = step(m1, L), step(m2, L)
                assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]), (rnd_, L)
            assert sum("graph" in e for e in t2._fg.values()) <= 3
        torch.cuda.synchronize()
        mem.append(torch.cuda.memory_allocated())
    captured = sorted(k[1][0][1] for k, e in t2._fg.items() if "graph" in e)
    assert captured == lengths[:3], captured
    assert t2.graph_stats["captured"] == 3 and t2.graph_stats["failed"] == 0
    assert mem[2] <= mem[1] + (1 << 20), f"device memory keeps growing with the number of shapes seen: {mem}"
    t2.MAX_TRACKED = 2
    step(m2, 17)
    assert sum("graph" not in e for e in t2._fg.values()) <= 2 and sum("graph" in e for e in t2._fg.values()) == 3


@pytest.mark.skipif(os.environ.get("BPMULT_GRAPH", "1") == "0", reason="graph replay switched off by BPMULT_GRAPH=0")
def test_dropped_trunks_park_their_graphs_and_new_ones_replay():
    """A trunk is dropped when a third batch size shows up (MAX_TRUNKS = 2), on .to() / set_prune_unused_rows(): its captured
    graphs are parked in _RETIRED_GRAPHS (destroying them is what crashes later launches on this stack) and the static
    tensors they pinned are released.  New trunks then capture and replay their own graphs with the eager results."""
    import copy
    from bpmult_amd.models import bpmult as BM
    m1 = _toy()
    m2 = copy.deepcopy(m1)
    m1, m2 = m1.cuda().train(), m2.cuda().train()
    m1.use_graphs, m2.use_graphs = False, True
    lossf = torch.nn.functional.binary_cross_entropy_with_logits
    n0 = len(BM._RETIRED_GRAPHS)
    for rnd_ in range(2):
        for B in (2, 3, 1):                                     # three batch sizes: every switch past the second drops a trunk
            x = _toy_inputs(B=B, seed=20 + B)
            tgt = (torch.randn(B, 6, generator=torch.Generator().manual_seed(B)) > 0).float().cuda()
            for it in range(4):
                outs = []
                for m in (m1, m2):
                    for p in m.parameters():
                        p.grad = None
                    out = m(x[0], None, None, x[1], x[2])
                    lossf(out, tgt).backward()
                    outs.append((out.detach().clone(), m.trans_v_with_a.layers[1].fc2.weight.grad.detach().clone()))
                assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]), (rnd_, B, it)
            assert any("graph" in e for e in m2._trunks[B]._fg.values())
    assert len(m2._trunks) == 2
    assert len(BM._RETIRED_GRAPHS) >= n0 + 2 * 4 - 2, "dropped trunks must park their forward + backward graphs"


@pytest.mark.skipif(os.environ.get("BPMULT_GRAPH", "1") == "0", reason="graph replay switched off by BPMULT_GRAPH=0")
def test_unjoined_side_stream_fails_the_capture_in_python_and_falls_back_to_eager():
    """A launch table that forks the side stream and never joins it back (here: the level-2 forward table -- the last one
    to use the side stream -- with its JOIN removed) must not reach hipStreamEndCapture unjoined: the capture joins the stream, ends, reports the key as
    non-capturable (a warning), and the step runs as eager launches with the right result."""
    import copy
    m1 = _toy()
    m2 = copy.deepcopy(m1)
    m1, m2 = m1.cuda().eval(), m2.cuda().eval()
    m1.use_graphs, m2.use_graphs = False, True
    x = _toy_inputs()
    with torch.no_grad():
        ref = m1(x[0], None, None, x[1], x[2])
        m2(x[0], None, None, x[1], x[2])
        t2 = m2._trunks[2]
        from bpmult_amd import engine
        assert t2.plan2._fwd[False][-1] is engine.JOIN
        t2.plan2._fwd[False] = t2.plan2._fwd[False][:-1]
        m2(x[0], None, None, x[1], x[2])
        with pytest.warns(UserWarning, match="graph capture of the forward pass failed"):
            out = m2(x[0], None, None, x[1], x[2])          # third call of the key: capture attempted
        assert t2.graph_stats["failed"] == 1 and not any("graph" in e for e in t2._fg.values())
        assert torch.equal(out, ref)
        assert torch.equal(m2(x[0], None, None, x[1], x[2]), ref)      # and stays eager, silently
    engine._OPEN_FORKS.clear()


def test_master_level_writes_refresh_the_weight_shadows():
    """The bf16 / f32 GEMM operands are shadows of the flat fp32 master, re-derived when the masters change.  A write
    through the flat master itself (dist.broadcast(master), EMA on the master) bumps no parameter's version counter:
    the store watches master._version too; writers of a `p.data` alias call ParamStore.invalidate()."""
    m = _toy().cuda().eval()
    x = _toy_inputs()
    with torch.no_grad():
        ref = m(x[0], None, None, x[1], x[2]).clone()
        st = m._store
        st.master.mul_(0.5)                                  # in place on the flat buffer: only master._version moves
        out = m(x[0], None, None, x[1], x[2]).clone()
        assert float((out - ref).abs().max()) > 1e-3, "halved trunk weights must change the logits"
        st.master.mul_(2.0)
        back = m(x[0], None, None, x[1], x[2]).clone()
        assert float((back - ref).abs().max()) <= 1e-5
        p = m.trans_l_with_a.layers[0].fc1.weight
        p.data.mul_(0.0)                                     # alias write: invisible to every version counter
        st.invalidate()
        out2 = m(x[0], None, None, x[1], x[2])
        assert float((out2 - ref).abs().max()) > 1e-6


def test_gradient_accumulation_and_batch_of_one():
    """Two backward passes without clearing .grad accumulate (train.py:390-398 gradient accumulation); batch size 1."""
    m = _toy().cuda().train()
    x = _toy_inputs(B=2)
    tgt = (torch.randn(2, 6, generator=torch.Generator().manual_seed(1)) > 0).float().cuda()
    lossf = torch.nn.functional.binary_cross_entropy_with_logits

    def grads(sl, clear):
        if clear:
            for p in m.parameters():
                p.grad = None
        lossf(m(x[0][sl], None, None, x[1][sl], x[2][sl]), tgt[sl], reduction="sum").backward()
        return {k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None}

    full = grads(slice(0, 2), True)
    grads(slice(0, 1), True)
    acc = grads(slice(1, 2), False)                         # accumulates onto the first sample's gradients
    for k in full:
        n = float(full[k].abs().max())
        if n < 1e-9:
            continue
        assert float((acc[k] - full[k]).abs().max()) <= 2e-4 * n + 1e-7, k


def test_input_longer_than_num_vectors_is_rejected():
    """The reference zero-pads to fixed lengths (mmtr.py:431-441) and would silently mis-shape longer inputs; here it
    is an error before anything is launched."""
    m = _toy().cuda().eval()
    x = _toy_inputs()
    too_long = torch.randn(2, 49, 35, device="cuda")          # num_vectors_v = 48
    with torch.no_grad(), pytest.raises(ValueError, match="exceeds num_vectors"):
        m(x[0], None, None, too_long, x[2])
    with torch.no_grad():                                      # exactly the maximum is fine
        m(x[0], None, None, torch.randn(2, 48, 35, device="cuda"), x[2])


def test_text_encoder_from_local_directory_through_the_hip_trunk(tmp_path):
    """SURVEY 8(f) rank 4 (mmtr.py:144-158, 740; collate formats helpers.py:78-137): token ids -> HF BertModel loaded from a
    LOCAL directory (`args.bert_model`, `text_features=False`) -> the HIP trunk, through `model.forward` and through
    training.model_forward's batch tuple; logits equal the same model fed the encoder's features directly, and the
    BERT parameters receive gradients through the one autograd node of the path."""
    from transformers import BertConfig, BertModel
    from bpmult_amd import training as TR
    cfg = BertConfig(vocab_size=60, hidden_size=32, num_hidden_layers=2, num_attention_heads=2, intermediate_size=64,
                     max_position_embeddings=32, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    torch.manual_seed(0)
    BertModel(cfg).save_pretrained(tmp_path / "tiny_bert")
    kw = dict(hidden_sz=24, num_heads=4, layers=2, orig_d_l=32, num_vectors_l=48, num_vectors_a=48, num_vectors_v=48)
    torch.manual_seed(3)
    m_tok = get_model(args_for("mmtrvat", bert_model=str(tmp_path / "tiny_bert"), text_features=False, **kw))
    m_tok.precision = "f32"
    m_feat = get_model(args_for("mmtrvat", **kw))
    m_feat.precision = "f32"
    m_feat.load_state_dict({k: v for k, v in m_tok.state_dict().items() if not k.startswith("enc.bert.")})
    m_tok, m_feat = m_tok.cuda().train(), m_feat.cuda().train()
    g = torch.Generator().manual_seed(5)
    txt = torch.randint(1, 60, (2, 17), generator=g)
    txt[1, 12:] = 0                                            # padded tail of the shorter sentence
    mask = (txt != 0).long()
    seg = torch.zeros_like(txt)
    img, aud = torch.randn(2, 40, 35, generator=g), torch.randn(2, 31, 74, generator=g)
    tgt = (torch.randn(2, 6, generator=g) > 0).float()
    txt, mask, seg, img, aud, tgt = (t.cuda() for t in (txt, mask, seg, img, aud, tgt))
    logits = m_tok(txt, mask, seg, img, aud)
    feats = m_tok.enc(txt, mask, seg).detach()
    assert feats.shape == (2, 17, 32)
    ref = m_feat(feats, None, None, img, aud)
    assert float((logits - ref).detach().abs().max()) <= 1e-5
    # the reference's batch tuple (helpers.py:129-133) through training.model_forward (train.py:283-338)
    loss, out, _ = TR.model_forward(m_tok, torch.nn.BCEWithLogitsLoss(), (txt, seg, mask, img, tgt, aud), "mmtrvat")
    assert float((out - ref).detach().abs().max()) <= 1e-5
    loss.backward()
    gw = m_tok.enc.bert.embeddings.word_embeddings.weight.grad
    assert gw is not None and torch.isfinite(gw).all() and float(gw.abs().max()) > 0
    # d(loss)/d(features) through the HIP path == the gradient reaching BERT's output
    feats2 = feats.clone().requires_grad_(True)
    torch.nn.functional.binary_cross_entropy_with_logits(m_feat(feats2, None, None, img, aud), tgt).backward()
    assert feats2.grad is not None and float(feats2.grad.abs().max()) > 0


def _prop_args(model, **kw):
    a = args_for(model, **kw)
    a.text_features = True
    return a


def _randn(*s, seed):
    return torch.randn(*s, generator=torch.Generator().manual_seed(seed))


@pytest.mark.parametrize("name", ["cfg4_mosei_b64", "cfg5_stress_d1536"])
def test_full_size_configs_properties(name):
    """BASELINE.json configs[3] (CMU-MOSEI shape: 3-modal, d=300, 12 heads, 8 layers, L/V/A = 50/500/500 -> 512, the
    per-GPU batch of 64) and configs[4] (stress: 4-modal, d=1536, 12 heads -> head_dim 128, seq_len 512 per modality,
    5 layers, per-GPU batch 8) at FULL size through the HIP path, bf16, default (pruned) schedule.  The reference cannot be run at these sizes in a test, so the
    checks are size-independent properties of the model: finite outputs of the right shape; train mode with every
    dropout rate 0 equals eval mode exactly (the only train/eval difference is dropout); gradients accumulate (two
    backward passes without clearing == twice one pass); parameters the graph never touches get no gradient; and rows of
    the batch are independent (the first sample alone gives the same logits as inside the batch)."""
    if name == "cfg4_mosei_b64":
        a = _prop_args("mmtrvat")                          # d=300, H=12, 8 layers, lengths padded to 512
        B = 64
        ins = [_randn(B, 50, 768, seed=1), _randn(B, 500, 35, seed=2), _randn(B, 500, 74, seed=3)]
        nc, unused = 6, ("transfm_",)
    else:
        a = _prop_args("mmtrvapt", hidden_sz=1536, num_heads=12, layers=5, orig_d_l=768, orig_d_v=4096, orig_d_a=96, orig_d_p=4096,
                       n_classes=13, num_vectors_l=512, num_vectors_a=512, num_vectors_v=512)
        B = 8                                                 # the configuration's per-GPU batch
        ins = [_randn(B, 512, 768, seed=1), _randn(B, 512, 4096, seed=2), _randn(B, 96, 2600, seed=3), _randn(B, 4096, seed=4)]
        nc, unused = 13, ()
    torch.manual_seed(3)
    model = get_model(a)
    if name == "cfg5_stress_d1536":
        # AudioEncoder ends in AdaptiveAvgPool1d(200) (mmtr.py:93-108): the 512-vector override needs a 512-long audio
        # feature sequence, so the stress shape pools to num_vectors_a
        model.audio_enc.conv_layers[2] = torch.nn.AdaptiveAvgPool1d(512)
    model.precision = "bf16"
    model = model.cuda()
    dev = [t.cuda() for t in ins]
    call = lambda m, xs: m(xs[0], None, None, *xs[1:])
    tgt = (_randn(B, nc, seed=9) > 0).float().cuda()

    model.train()
    out_train = call(model, dev)
    assert out_train.shape == (B, nc) and torch.isfinite(out_train).all()
    model.eval()
    with torch.no_grad():
        out_eval = call(model, dev)
        one = call(model, [t[:1] for t in dev])
    assert torch.equal(out_train.detach(), out_eval), "dropout 0: train == eval"
    e1 = float((one[0] - out_eval[0]).abs().max() / out_eval.abs().max().clamp_min(1e-6))
    assert e1 < 2e-2, f"batch rows are independent: {e1:.3e}"

    model.train()
    for p in model.parameters():
        p.grad = None
    torch.nn.functional.binary_cross_entropy_with_logits(call(model, dev), tgt).backward()
    g1 = {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}
    for k, p in model.named_parameters():
        if k.startswith(unused) or (k == "proj_l.weight" and a.orig_d_l == a.hidden_sz):
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, k
        else:
            assert p.grad is not None and torch.isfinite(p.grad).all(), k
    torch.nn.functional.binary_cross_entropy_with_logits(call(model, dev), tgt).backward()      # accumulate on top
    worst = 0.0
    for k, p in model.named_parameters():
        if k not in g1 or float(g1[k].abs().max()) == 0.0:
            continue
        worst = max(worst, float((p.grad - 2 * g1[k]).abs().max() / (2 * g1[k].abs().max())))
    assert worst < 2e-3, f"gradient accumulation identity: {worst:.3e}"
