"""Host logic without a GPU: the launch tables of the whole hot path (problem structs of every grouped launch of the
forward and of both backward variants, weight-shadow / fold / unfold / zero-segment descriptor tables) build for the
3-modal and the 4-modal model from HOST tensors (ops._DRY_RUN).  Nothing is launched -- this catches shape, flag and
naming mistakes in the table builders before a GPU box is involved, and checks a few structural invariants."""
from types import SimpleNamespace

import pytest
import torch

import bpmult_amd  # noqa: F401
from bpmult_amd import engine, ops
from bpmult_amd._lib import F_ACCUM, GEMM_TN
from bpmult_amd.models import get_model


def _args(model, **kw):
    a = dict(model=model, orig_d_l=32, orig_d_v=35, orig_d_a=74, orig_d_p=64, hidden_sz=64, vonly=True, lonly=True, aonly=True,
             num_heads=4, layers=2, attn_dropout=0.1, attn_dropout_v=0., attn_dropout_a=0., relu_dropout=0.1, res_dropout=0.1,
             out_dropout=0., embed_dropout=0.25, attn_mask=True, hybrid=False, n_classes=6, bert_model="unused",
             text_features=True, precision="bf16", num_vectors_l=48, num_vectors_a=48, num_vectors_v=48)
    a.update(kw)
    return SimpleNamespace(**a)


@pytest.fixture
def dry_run():
    ops._DRY_RUN = True
    try:
        yield
    finally:
        ops._DRY_RUN = False


@pytest.mark.parametrize("prune", [True, False])
@pytest.mark.parametrize("model,kw", [("mmtrvat", {}), ("mmtrvat", {"hidden_sz": 40, "precision": "f32"}),
                                      ("mmtrvapt", {"orig_d_a": 96, "num_vectors_a": 40, "num_vectors_v": 40}),
                                      ("mmtrvapt", {"orig_d_a": 96, "hidden_sz": 40, "num_vectors_a": 40, "num_vectors_v": 40})])
def test_launch_tables_build_from_host_tensors(dry_run, model, kw, prune):
    """Both schedules: dense (the reference's) and exact dead-row elimination (the default)."""
    m = get_model(_args(model, prune_unused_rows=prune, **kw))
    st = m._ensure_store()
    trunk = m._trunk_for(2)
    assert trunk.prune == prune
    L = m.layers
    if prune:                                    # level 2 / GMU / time-map outputs shrink to rows {0, N-1}
        for n, b in zip(trunk.out2, trunk.plan2.buf):
            assert trunk.out2[n].shape[0] == 2 and b["Rl"][-1] == 2 * 2
            assert b["tailp"] == (model == "mmtrvapt") and (b["Rl"][0] == b["R"] or L == 1 or model == "mmtrvat")
        assert all(t["out"].shape[0] == 2 for t in trunk.tmap.values())
        assert all(g["R"] == 2 * 2 for g in trunk.g.values())
    for plan in (trunk.plan1, trunk.plan2):
        assert set(plan._bwd) == {(t, f) for t in (True, False) for f in (True, False)}
        for training in (True, False):
            acc, sto = plan._bwd[(training, False)], plan._bwd[(training, True)]
            assert len(acc) == len(sto)
            n_first = 0
            for a, s in zip(acc, sto):                       # same steps; only first-writer flags / the unfold mode differ
                a, s = (a[1] if isinstance(a, tuple) and a[0] in (engine.SIDE, engine.SIDE2) else a), \
                       (s[1] if isinstance(s, tuple) and s[0] in (engine.SIDE, engine.SIDE2) else s)
                if isinstance(a, tuple) and a[0] is ops.gemm_grouped and a[2] == GEMM_TN:
                    for pa, ps in zip(a[3], s[3]):
                        assert pa.C == ps.C and (pa.flags & ~F_ACCUM) == (ps.flags & ~F_ACCUM)
                        n_first += bool(pa.flags & F_ACCUM) and not (ps.flags & F_ACCUM)
            # per encoder and layer: fc1.weight, fc2.weight, out_proj.weight, in_proj rows [0,d) (+ rows [d,3d) twice when a
            # biprojection self-attention half writes them before unfold_grads)
            per = 6 if plan.cfg.biprojection else 4
            assert n_first == per * L * len(plan.encs), (n_first, per, L)
        # every layer's dK / dV block sits inside the stacked buffer the merged data-gradient product reads
        # (low-rank key side -- a handful of query time steps, level 2 under dead-row elimination: dS / Pd of every layer
        # instead, [layer][h*T + t][b][padded keys], and the merged product runs over K = layers * H * T)
        assert plan._lowrank == (prune and model == "mmtrvat" and plan is trunk.plan2)
        for e, b in zip(plan.encs, plan.buf):
            assert b["Gk"].shape == (b["Rk"], plan.cfg.d)
            if plan._lowrank:
                HT = plan.cfg.H * e.T
                assert b["dkall"] is None and b["dSall"].shape == b["Pdall"].shape == (L, HT, 2, b["Sp"]) and b["Sp"] % 64 == 0
                assert b["qkall"].shape == b["daall"].shape == (L, HT * 2, plan.ld)
            else:
                assert b["dkall"].shape == (b["Rk"], L * plan.ld)
    # the zero list covers everything but the large encoder matrices, without overlapping them
    tab, nd, nblk = st._zero_table
    assert nd > 0 and nblk >= nd
    big = sum(st.params[n].numel() for n in st._store_written)
    assert big > 0.5 * sum(p.numel() for p in st.params.values())
    assert all(n in st.params for n in st._store_written)


def test_low_rank_key_side_tables(dry_run):
    """Level 2 of the pruned 3-modal model (two query rows): its backward table has no dK / dV pass; per layer the dQ pass
    exports dS / Pd in the [h*T + t][b][padded keys] layout, bpm_expand_heads and ONE NN launch (Qexp W' products + the
    per-batch-element products as BATCHED problems) follow, and the key / value-source gradient is one batched TN product
    over K = layers * H * T at the end (engine.EncoderGroupPlan._lowrank)."""
    from bpmult_amd._lib import F_BATCHED, F_CT_NARROW, GEMM_NN
    m = get_model(_args("mmtrvat", prune_unused_rows=True))
    m._ensure_store()
    B = 2
    trunk = m._trunk_for(B)
    p1, p2 = trunk.plan1, trunk.plan2
    assert p2._lowrank and not p1._lowrank
    H, L, d, ld = p2.cfg.H, p2.cfg.layers, p2.cfg.d, p2.ld
    G = len(p2.encs)
    un = lambda s: s[1] if isinstance(s, tuple) and s[0] in (engine.SIDE, engine.SIDE2) else s
    steps = [un(s) for s in p2._bwd[(True, True)] if s is not engine.JOIN]
    fns = [s[0] for s in steps if isinstance(s, tuple) and callable(s[0])]
    assert ops.attn_bwd_dkv not in fns and fns.count(ops.expand_heads) == L and fns.count(ops.attn_bwd_dq) == L
    assert ops.attn_bwd_dkv in [un(s)[0] for s in p1._bwd[(True, True)] if s is not engine.JOIN and callable(un(s)[0])]
    for s in steps:
        if not (isinstance(s, tuple) and callable(s[0])):
            continue
        if s[0] is ops.attn_bwd_dq:
            for e, b, a in zip(p2.encs, p2.buf, s[2]):
                Sp = b["Sp"]
                assert a.dS and a.Pd and (a.xs_b, a.xs_h, a.xs_q) == (Sp, e.T * B * Sp, B * Sp) and Sp % 64 == 0 and Sp >= e.S
        if s[0] is ops.expand_heads:
            assert len(s[2]) == G and all(x.B == B and x.H == H and x.T == e.T and x.ld == ld and x.dbias for x, e in zip(s[2], p2.encs))
        if s[0] is ops.gemm_grouped and s[2] == GEMM_NN and any(q.flags & F_BATCHED for q in s[3]):
            plain = [q for q in s[3] if not q.flags & F_BATCHED]
            bat = [q for q in s[3] if q.flags & F_BATCHED]
            assert len(plain) == 2 * G and len(bat) == 2 * G          # Qexp W_k', dOexp W_v'  |  dS khat, Pd vhat
            for q, e in zip(bat[::2], p2.encs):
                Sp = p2.buf[p2.encs.index(e)]["Sp"]
                assert (q.M, q.N, q.K) == (H * e.T, d, e.S) and q.batch == B and q.flags & F_CT_NARROW
                assert (q.lda, q.ldb, q.ldc) == (B * Sp, B * ld, B * ld) and (q.batch_stride_a, q.batch_stride_b, q.batch_stride_c) == (Sp, ld, ld)
            for q, e in zip(plain[::2], p2.encs):
                assert (q.M, q.N, q.K) == (H * e.T * B, d, d)
    last_tn = [s for s in steps if isinstance(s, tuple) and s[0] is ops.gemm_grouped and s[2] == GEMM_TN][-1]
    assert len(last_tn[3]) == 2 * G
    for q, e in zip(last_tn[3][::2], p2.encs):
        assert q.flags & F_BATCHED and q.batch == B and (q.M, q.N, q.K) == (e.S, d, L * H * e.T) and q.ldc == B * d and q.batch_stride_c == d
