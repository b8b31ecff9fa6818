"""Host-side loop plumbing on CPU: reference checkpoint formats incl. the DataParallel
`module.` prefix, the plateau / early-stop / resume loop, batch collation, and the local-directory BERT plug point."""
import os
from types import SimpleNamespace

import pytest
import torch
from torch import nn

import bpmult_amd  # noqa: F401
from bpmult_amd import training as TR
from bpmult_amd.models import get_model


def _args(**kw):
    a = dict(model="mmtrvat", orig_d_l=32, orig_d_v=35, orig_d_a=74, orig_d_p=64, hidden_sz=24, vonly=True, lonly=True, aonly=True,
             num_heads=4, layers=1, attn_dropout=0., attn_dropout_v=0., attn_dropout_a=0., relu_dropout=0., res_dropout=0.,
             out_dropout=0., embed_dropout=0., attn_mask=True, hybrid=False, n_classes=6, bert_model="unused", text_features=True)
    a.update(kw)
    return SimpleNamespace(**a)


def test_reference_checkpoint_with_module_prefix_round_trips(tmp_path):
    """train.py:419-430 saves {"state_dict": DataParallel(model).state_dict(), ...}: every key starts with `module.`."""
    torch.manual_seed(0)
    src, dst = get_model(_args()), get_model(_args())
    wrapped = {"module." + k: v.clone() for k, v in src.state_dict().items()}
    TR.save_checkpoint({"epoch": 3, "state_dict": wrapped, "optimizer": {}, "scheduler": {}, "n_no_improve": 0, "best_metric": 0.5},
                       True, str(tmp_path))
    assert os.path.exists(tmp_path / "checkpoint.pt") and os.path.exists(tmp_path / "model_best.pt")
    missing, unexpected = TR.load_reference_checkpoint(dst, str(tmp_path / "model_best.pt"))
    assert missing == [] and unexpected == []
    for (k, a), (_, b) in zip(src.state_dict().items(), dst.state_dict().items()):
        assert torch.equal(a, b), k
    # bare state_dict, no prefix, via the reference's own helper name
    dst2 = get_model(_args())
    torch.save({"state_dict": src.state_dict()}, tmp_path / "plain.pt")
    TR.load_checkpoint(dst2, str(tmp_path / "plain.pt"))
    assert all(torch.equal(a, b) for a, b in zip(src.state_dict().values(), dst2.state_dict().values()))
    # a key that does not belong is reported under strict loading
    bad = dict(wrapped)
    bad["module.not_a_parameter"] = torch.zeros(1)
    with pytest.raises(RuntimeError):
        TR.load_reference_checkpoint(dst, {"state_dict": bad})


def test_fit_plateau_early_stop_and_resume(tmp_path):
    """train.py:370-439: ReduceLROnPlateau on the tuning metric, checkpoint on improvement (>=), stop after `patience`
    epochs without one, resume from checkpoint.pt."""
    torch.manual_seed(0)
    model = nn.Linear(4, 1)
    opt = torch.optim.Adam(model.parameters(), lr=1e-2)
    sched = TR.get_scheduler(opt, lr_patience=0, lr_factor=0.5, mode="max")
    x, y = torch.randn(16, 4), torch.randn(16, 1)
    metrics = iter([0.1, 0.3, 0.2, 0.25, 0.2, 0.9])
    steps = []

    def batches():
        for i in range(4):
            yield x[4 * i: 4 * i + 4], y[4 * i: 4 * i + 4]

    def fl(m, b):
        steps.append(1)
        return ((m(b[0]) - b[1]) ** 2).mean()

    r = TR.fit(model, opt, sched, batches, fl, lambda m: next(metrics), str(tmp_path), max_epochs=10, patience=3,
               gradient_accumulation_steps=2)
    # epochs: 0.1 (improve), 0.3 (improve), 0.2, 0.25, 0.2 -> three epochs without improvement -> stop after 5 epochs
    assert r["epochs_run"] == 5 and abs(r["best_metric"] - 0.3) < 1e-12
    assert [h["improved"] for h in r["history"]] == [True, True, False, False, False]
    assert r["global_step"] == 20 and len(steps) == 20
    assert r["history"][-1]["lr"] < 1e-2                       # the plateau scheduler cut the rate
    ck = torch.load(tmp_path / "checkpoint.pt", weights_only=False)
    assert ck["epoch"] == 2 and set(ck) == {"epoch", "state_dict", "optimizer", "scheduler", "n_no_improve", "best_metric"}
    # resume: starts at the stored epoch with the stored counters
    model2 = nn.Linear(4, 1)
    opt2 = torch.optim.Adam(model2.parameters(), lr=1e-2)
    r2 = TR.fit(model2, opt2, TR.get_scheduler(opt2, 0, 0.5), batches, fl, lambda m: 0.0, str(tmp_path), max_epochs=3, patience=5)
    assert r2["history"][0]["epoch"] == 2 and r2["epochs_run"] == 1
    assert TR.run_seeds(lambda s: s * s, from_seed=3) == {3: 9, 4: 16, 5: 25}
    assert list(TR.run_seeds(lambda s: s, from_seed=1, inverse_seed=True)) == [5, 4, 3, 2, 1]


def test_collate_and_forward_argument_order():
    """data/helpers.py:78-133 batch tuple; train.py:313 passes (txt, mask, segment, ...) to the model."""
    rows = []
    for n, ta in ((5, 12), (3, 9)):
        rows.append((torch.arange(1, n + 1), torch.ones(n, dtype=torch.long), torch.randn(7, 35), torch.ones(6),
                     torch.randn(74, ta), torch.randn(64)))
    text, segment, mask, img, tgt, audio = TR.collate_fn(rows, "mmtrvat")
    assert text.shape == (2, 5) and text[1].tolist() == [1, 2, 3, 0, 0] and mask[1].tolist() == [1, 1, 1, 0, 0]
    assert segment[1].tolist() == [1, 1, 1, 0, 0] and img.shape == (2, 7, 35) and tgt.shape == (2, 6)
    assert audio.shape == (2, 74, 9)                            # cropped to the batch minimum
    b4 = TR.collate_fn(rows, "mmtrvapt")
    assert len(b4) == 7 and b4[6].shape == (2, 64)

    seen = {}

    class Probe(nn.Module):
        def __init__(self):
            super().__init__()
            self.w = nn.Parameter(torch.zeros(1))

        def forward(self, txt, mask, segment, img, audio, gate=False):
            seen.update(txt=txt, mask=mask, segment=segment)
            return torch.zeros(txt.shape[0], 6) + self.w

    loss, out, t = TR.model_forward(Probe(), nn.BCEWithLogitsLoss(), (text, segment * 7, mask, img, tgt, audio), "mmtrvat")
    assert torch.equal(seen["mask"], mask) and torch.equal(seen["segment"], segment * 7) and out.shape == (2, 6)


def test_text_encoder_from_a_local_directory(tmp_path):
    """mmtr.py:144-158: BertEncoder = HF BertModel.from_pretrained(args.bert_model); offline only a LOCAL directory can
    be given.  A tiny randomly initialised BERT is saved and loaded back through the plug point; its sequence output
    feeds the text projection (orig_d_l = BERT hidden size)."""
    from transformers import BertConfig, BertModel
    from bpmult_amd.models.bpmult import BertEncoder
    cfg = BertConfig(vocab_size=50, hidden_size=32, num_hidden_layers=1, num_attention_heads=2, intermediate_size=64,
                     max_position_embeddings=16)
    torch.manual_seed(0)
    ref = BertModel(cfg).eval()
    ref.save_pretrained(tmp_path / "tiny_bert")
    enc = BertEncoder(_args(bert_model=str(tmp_path / "tiny_bert"), text_features=False)).eval()
    txt = torch.tensor([[2, 5, 7, 0], [3, 4, 0, 0]])
    mask = (txt != 0).long()
    seg = torch.zeros_like(txt)
    with torch.no_grad():
        out = enc(txt, mask, seg)
        want = ref(input_ids=txt, token_type_ids=seg, attention_mask=mask, return_dict=False)[0]
    assert out.shape == (2, 4, 32) and torch.allclose(out, want, atol=1e-6)
    # the same state_dict prefix as the reference: enc.bert.*
    model = get_model(_args(bert_model=str(tmp_path / "tiny_bert"), text_features=False))
    assert any(k.startswith("enc.bert.embeddings.word_embeddings") for k in model.state_dict())
