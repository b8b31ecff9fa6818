"""No-GPU checks of the drop-in boundary: the C-ABI library builds for gfx950,
loads, and exports every symbol include/bpmult_hip.h declares; the ctypes
structure layouts match the header; argument validation rejects bad calls
before anything is launched; the product path refuses to run without CUDA."""
import ctypes as C
import os
import re

import pytest
import torch

import bpmult_amd  # noqa: F401
from bpmult_amd import _lib, ops

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = open(os.path.join(ROOT, "include", "bpmult_hip.h")).read()


@pytest.fixture(scope="module")
def lib():
    _lib.build()
    return _lib.lib()


def test_every_declared_symbol_is_exported_and_bound(lib):
    declared = set(re.findall(r"^\s*(?:int|size_t|const char\*)\s+(bpm_\w+)\s*\(", HEADER, flags=re.M))
    assert len(declared) >= 17
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.bpm_version() == int(re.search(r"#define BPM_ABI_VERSION (\d+)", HEADER).group(1))
    assert _lib.MAX_GROUP == int(re.search(r"#define BPM_MAX_GROUP (\d+)", HEADER).group(1))
    assert _lib.GEMM_MAX_GROUP == int(re.search(r"#define BPM_GEMM_MAX_GROUP (\d+)", HEADER).group(1))


def _c_fields(struct_name):
    body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (struct_name, struct_name), HEADER, flags=re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        parts = decl.split(",")
        first = parts[0].split()
        names.append(first[-1].lstrip("*"))
        names += [p.strip().lstrip("*") for p in parts[1:]]
    return [re.sub(r"\[\d+\]$", "", n) for n in names]


@pytest.mark.parametrize("cname,cls", [("bpm_gemm_problem", _lib.GemmProblem), ("bpm_attn_problem", _lib.AttnProblem),
                                       ("bpm_pack_problem", _lib.PackProblem), ("bpm_pack_desc", _lib.PackDesc),
                                       ("bpm_embed_problem", _lib.EmbedProblem), ("bpm_ln_problem", _lib.LnProblem),
                                       ("bpm_cast_problem", _lib.CastProblem), ("bpm_gmu_problem", _lib.GmuProblem),
                                       ("bpm_fold_desc", _lib.FoldDesc), ("bpm_unfold_desc", _lib.UnfoldDesc),
                                       ("bpm_tail_desc", _lib.TailDesc), ("bpm_tail_grads", _lib.TailGrads),
                                       ("bpm_addn_problem", _lib.AddnProblem), ("bpm_zero_desc", _lib.ZeroDesc)])
def test_ctypes_structs_mirror_the_header(cname, cls):
    assert _c_fields(cname) == [f[0] for f in cls._fields_]


def test_integration_example_binds_the_current_abi():
    """INTEGRATION.md's ctypes example (what a maintainer of the reference would copy) lists exactly the fields of
    bpm_attn_problem, in order, and names the current BPM_ABI_VERSION -- a shorter struct would make the library read the
    optional dS / Pd pointers past the caller's allocation."""
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    block = doc[doc.index("class AttnProblem(C.Structure)"):]
    block = block[:block.index("lib.bpm_attn_fwd.argtypes")]
    names = re.findall(r'\("(\w+)", C\.c_', block)
    assert names == [n for n, _ in _lib.AttnProblem._fields_], names
    ver = int(re.search(r"#define BPM_ABI_VERSION (\d+)", HEADER).group(1))
    assert f"bpm_version() == {ver}" in block and f"`BPM_ABI_VERSION` is {ver}" in doc


def test_error_strings_and_argument_validation(lib):
    assert lib.bpm_error_string(0) == b"ok"
    assert b"invalid argument" in lib.bpm_error_string(-1)
    assert b"aligned" in lib.bpm_error_string(-2)
    p = _lib.GemmProblem()                      # all-zero problem: rejected on the host, nothing launched
    assert lib.bpm_gemm_grouped(_lib.BPM_BF16, _lib.GEMM_NT, C.byref(p), 1, 0, None) == -1
    assert lib.bpm_gemm_grouped(_lib.BPM_BF16, _lib.GEMM_NT, C.byref(p), 0, 0, None) == -1
    a = _lib.AttnProblem()
    assert lib.bpm_attn_fwd(_lib.BPM_F32, C.byref(a), 1, 0, None) == -1
    ln = _lib.LnProblem()
    assert lib.bpm_ln_fwd(_lib.BPM_F32, C.byref(ln), 1, 300, 1e-5, None) == -1
    p.A, p.B, p.C, p.M, p.N, p.K, p.lda, p.ldb = 16, 16, 16, 4, 4, 4, 3, 32   # lda*2 bytes not a multiple of 16
    assert lib.bpm_gemm_grouped(_lib.BPM_BF16, _lib.GEMM_NT, C.byref(p), 1, 0, None) == -2
    # later additions: every entry validates on the host before touching the device
    assert lib.bpm_attn_bwd_dq(_lib.BPM_BF16, C.byref(a), 1, 0, None) == -1
    assert lib.bpm_attn_bwd_dkv(_lib.BPM_BF16, C.byref(a), 1, 0, None) == -1
    assert lib.bpm_ln_bwd(_lib.BPM_F32, C.byref(ln), 1, 300, 0, None) == -1
    assert lib.bpm_fold_bias(None, 1, 1, None) == -1 and lib.bpm_unfold_grads(None, 1, 1, 0, None) == -1
    assert lib.bpm_adam_step(None, None, None, None, 16, 1e-3, .9, .999, 1e-8, 0., 1, 1., 0, None) == -1
    assert lib.bpm_adam_step(16, 16, 16, 16, 6, 1e-3, .9, .999, 1e-8, 0., 1, 1., 0, None) == -1     # n % 4
    assert lib.bpm_adam_step(16, 16, 16, 20, 8, 1e-3, .9, .999, 1e-8, 0., 1, 1., 0, None) == -2     # alignment
    assert lib.bpm_adam_step(16, 16, 16, 16, 8, 1e-3, .9, .999, 1e-8, 0., 0, 1., 0, None) == -1     # step >= 1
    assert lib.bpm_stream_create(1, None) == -1
    e = _lib.EmbedProblem()
    assert lib.bpm_embed_pos_bwd(C.byref(e), 1, 24, 1.0, 0, None) == -1


def test_no_cpu_fallback():
    with pytest.raises(ValueError, match="no CPU fallback"):
        ops.gemm_problem(torch.zeros(4, 32), torch.zeros(4, 32), torch.zeros(4, 4), 4, 4, 4, 32, 32, 4)
    from bpmult_amd.models.encoder import TransformerEncoder
    enc = TransformerEncoder(24, 4, 1)
    x = torch.zeros(3, 2, 24)
    with pytest.raises(RuntimeError, match="no CPU path"):
        enc(x, x, x)


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.HipLibraryError, match="no CPU/PyTorch fallback"):
        _lib.lib()


def test_model_surface_matches_reference_registry():
    from types import SimpleNamespace
    from bpmult_amd.models import MODELS, get_model
    assert set(MODELS) == {"mmtrvat", "mmtrvapt"}
    a = SimpleNamespace(model="mmtrvat", orig_d_l=32, orig_d_v=35, orig_d_a=74, orig_d_p=64, hidden_sz=24, vonly=True, lonly=True,
                        aonly=True, num_heads=4, layers=2, attn_dropout=.1, attn_dropout_v=0., attn_dropout_a=0., relu_dropout=.1,
                        res_dropout=.1, out_dropout=0., embed_dropout=.25, attn_mask=True, hybrid=False, n_classes=6,
                        bert_model="unused", text_features=True)
    m = get_model(a)
    from oracle import bpmult_cpu as O
    shapes = O.model_param_shapes(O.ModelCfg(24, 4, 2, 6, orig_d_l=32), False)      # == the reference's named_parameters (f7 fixture)
    mine = {k: tuple(p.shape) for k, p in m.named_parameters()}
    assert mine == shapes
    sd = m.state_dict()
    assert "trans_l_with_a.version" in sd and "trans_l_with_a.embed_positions._float_tensor" in sd
    # attention dropout is keyed by the key/value source modality (reference get_network tags)
    assert m.trans_a_with_l.attn_dropout == 0.1 and m.trans_l_with_a.attn_dropout == 0.0 and m.trans_l_with_v2a.attn_dropout == 0.0
    assert m.trans_v_with_a2l.attn_dropout == 0.1


def test_fused_adam_is_a_torch_optimizer_for_the_reference_loop():
    """The reference wraps its optimizer in ReduceLROnPlateau (train.py:128-136) and checkpoints optimizer.state_dict()
    (train.py:372-379): FusedAdam must be accepted by both (it is a torch.optim.Optimizer whose param group carries lr)."""
    import torch
    from types import SimpleNamespace
    from bpmult_amd.models import get_model
    from bpmult_amd.optim import FusedAdam
    a = SimpleNamespace(model="mmtrvat", orig_d_l=32, orig_d_v=35, orig_d_a=74, orig_d_p=64, hidden_sz=24, vonly=True, lonly=True,
                        aonly=True, num_heads=4, layers=1, attn_dropout=0., attn_dropout_v=0., attn_dropout_a=0., relu_dropout=0.,
                        res_dropout=0., out_dropout=0., embed_dropout=0., attn_mask=True, hybrid=False, n_classes=6,
                        bert_model="unused", text_features=True)
    model = get_model(a)
    opt = FusedAdam(model, lr=1e-3)
    assert isinstance(opt, torch.optim.Optimizer)
    sched = torch.optim.lr_scheduler.ReduceLROnPlateau(opt, "max", patience=0, factor=0.5)
    sched.step(1.0)
    sched.step(0.5)             # no improvement -> lr halves
    assert abs(opt.param_groups[0]["lr"] - 5e-4) < 1e-12 and abs(opt.lr - 5e-4) < 1e-12
    assert len(opt.param_groups) == 1 and len(opt.param_groups[0]["params"]) == len([p for p in model.parameters() if p.requires_grad])
