"""Training-loop plumbing around the hot path, mirroring the reference's train.py so that a run can swap the model in
unchanged.  Data loading and metrics stay the caller's (out of scope): the loop takes an iterable
of batches, a `forward_loss(model, batch)` callable and an `evaluate(model)` callable that returns the tuning metric.

Reference behaviour reproduced (file:line in /root/reference/bpmult):
* checkpoints: `save_checkpoint(state, is_best, path)` writes `checkpoint.pt` and copies it to `model_best.pt`
  (utils/utils.py:21-25); the dict holds epoch / state_dict / optimizer / scheduler / n_no_improve / best_metric
  (train.py:419-430); `load_checkpoint(model, path)` reads `["state_dict"]` (utils/utils.py:28-30).  A checkpoint saved
  from the reference's `nn.DataParallel` wrapper (train.py:354-356) prefixes every key with `module.`: stripped here.
* resume: `checkpoint.pt` in the save directory restores epoch, counters, model (strict=False), optimizer and scheduler
  (train.py:372-379).
* epoch loop: zero_grad, loss / accumulation steps, backward, step every `gradient_accumulation_steps` (train.py:382-398);
  `scheduler.step(tuning_metric)` with ReduceLROnPlateau (train.py:128-136, 410); improvement is `>=` (`<=` when the
  metric is minimised, train.py:411-414); a checkpoint is written on improvement only; stop when
  `n_no_improve >= patience` (train.py:432-439).
* the five-seed outer loop `for i in range(from_seed, 6)` with `inverse_seed` (train.py:490-503).
"""
from __future__ import annotations

import os
import shutil
from typing import Callable, Dict, Iterable, Optional, Tuple

import torch


def strip_module_prefix(state_dict: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """Keys of a DataParallel / DistributedDataParallel wrapper -> keys of the wrapped model."""
    if state_dict and all(k.startswith("module.") for k in state_dict):
        return {k[len("module."):]: v for k, v in state_dict.items()}
    return dict(state_dict)


def load_reference_checkpoint(model, src, strict: bool = True, map_location="cpu") -> Tuple[list, list]:
    """Load a reference `checkpoint.pt` / `model_best.pt` (a dict with "state_dict"), a bare state_dict, or a path to
    either, into a bpmult_amd model.  Returns (missing_keys, unexpected_keys)."""
    if isinstance(src, (str, os.PathLike)):
        src = torch.load(src, map_location=map_location, weights_only=False)
    sd = src["state_dict"] if isinstance(src, dict) and "state_dict" in src else src
    res = model.load_state_dict(strip_module_prefix(sd), strict=strict)
    return list(res.missing_keys), list(res.unexpected_keys)


def load_checkpoint(model, path) -> None:
    """utils/utils.py:28-30."""
    load_reference_checkpoint(model, path)


def save_checkpoint(state: dict, is_best: bool, checkpoint_path: str, filename: str = "checkpoint.pt") -> None:
    """utils/utils.py:21-25."""
    os.makedirs(checkpoint_path, exist_ok=True)
    filename = os.path.join(checkpoint_path, filename)
    torch.save(state, filename)
    if is_best:
        shutil.copyfile(filename, os.path.join(checkpoint_path, "model_best.pt"))


def get_scheduler(optimizer, lr_patience: int = 2, lr_factor: float = 0.5, mode: str = "max"):
    """train.py:128-136."""
    return torch.optim.lr_scheduler.ReduceLROnPlateau(optimizer, mode, patience=lr_patience, factor=lr_factor)


def fit(model, optimizer, scheduler, train_batches: Callable[[], Iterable], forward_loss: Callable, evaluate: Callable,
        savedir: str, max_epochs: int, patience: int, gradient_accumulation_steps: int = 1, minimise: bool = False,
        grad_sync=None, log: Optional[Callable[[str], None]] = None) -> dict:
    """The reference's train() epoch loop (train.py:370-439) around any model / optimizer pair.

    train_batches() yields the batches of one epoch; forward_loss(model, batch) returns the scalar loss;
    evaluate(model) returns the tuning metric of the validation split.  grad_sync: an optional
    distributed.GradSync -- only the last micro-step of an accumulation group exchanges gradients."""
    log = log or (lambda s: None)
    start_epoch, global_step, n_no_improve = 0, 0, 0
    best_metric = float("inf") if minimise else -float("inf")
    ck = os.path.join(savedir, "checkpoint.pt")
    if os.path.exists(ck):
        c = torch.load(ck, map_location="cpu", weights_only=False)
        start_epoch, n_no_improve, best_metric = c["epoch"], c["n_no_improve"], c["best_metric"]
        model.load_state_dict(strip_module_prefix(c["state_dict"]), strict=False)
        optimizer.load_state_dict(c["optimizer"])
        scheduler.load_state_dict(c["scheduler"])
    history = []
    for i_epoch in range(start_epoch, max_epochs):
        losses = []
        model.train()
        optimizer.zero_grad()
        for batch in train_batches():
            loss = forward_loss(model, batch)
            if gradient_accumulation_steps > 1:
                loss = loss / gradient_accumulation_steps
            losses.append(float(loss.detach()))
            last = (global_step + 1) % gradient_accumulation_steps == 0
            if grad_sync is not None:
                grad_sync.active = last
            loss.backward()
            if grad_sync is not None:
                grad_sync.finish()
            global_step += 1
            if last:
                optimizer.step()
                optimizer.zero_grad()
        model.eval()
        with torch.no_grad():
            tuning_metric = float(evaluate(model))
        log(f"epoch {i_epoch}: train loss {sum(losses) / max(len(losses), 1):.4f}, tuning metric {tuning_metric:.4f}")
        scheduler.step(tuning_metric)
        is_improvement = tuning_metric <= best_metric if minimise else tuning_metric >= best_metric
        if is_improvement:
            best_metric, n_no_improve = tuning_metric, 0
        else:
            n_no_improve += 1
        history.append({"epoch": i_epoch, "loss": sum(losses) / max(len(losses), 1), "metric": tuning_metric,
                        "lr": optimizer.param_groups[0]["lr"], "improved": is_improvement})
        if is_improvement:
            save_checkpoint({"epoch": i_epoch + 1, "state_dict": model.state_dict(), "optimizer": optimizer.state_dict(),
                             "scheduler": scheduler.state_dict(), "n_no_improve": n_no_improve, "best_metric": best_metric},
                            is_improvement, savedir)
        if n_no_improve >= patience:
            log("No improvement. Breaking out of loop.")
            break
    return {"best_metric": best_metric, "epochs_run": len(history), "history": history, "global_step": global_step}


def run_seeds(run_one: Callable[[int], object], from_seed: int = 1, inverse_seed: bool = False) -> Dict[int, object]:
    """train.py:490-503: seeds from_seed..5 (or 6 - i when inverse_seed)."""
    out = {}
    for i in range(from_seed, 6):
        seed = 6 - i if inverse_seed else i
        out[seed] = run_one(seed)
    return out


# ----------------------------------------------------------------------------
# batch formats (data/helpers.py:78-137, train.py:283-321): the loaders themselves are out of scope
# ----------------------------------------------------------------------------
def collate_fn(batch, model: str, task_type: str = "multilabel", with_poster: bool = True):
    """Rows as the reference's datasets yield them -- (tokens, segment, img, tgt, audio[, poster]) -- to the batch tuple
    of data/helpers.py:129-133: (text, segment, mask, img, tgt, audio[, poster]).  Text is right-padded with zeros to the
    longest row (mask = 1 on real tokens); audio is cropped to the batch-minimum length along its last axis
    (helpers.py:100-102, 106-110)."""
    bsz = len(batch)
    lens = [len(row[0]) for row in batch]
    max_len = max(lens)
    text = torch.zeros(bsz, max_len, dtype=torch.long)
    segment = torch.zeros(bsz, max_len, dtype=torch.long)
    mask = torch.zeros(bsz, max_len, dtype=torch.long)
    for i, (row, n) in enumerate(zip(batch, lens)):
        text[i, :n], segment[i, :n], mask[i, :n] = row[0], row[1], 1
    img = torch.stack([row[2] for row in batch])
    tgt = torch.stack([row[3] for row in batch]) if task_type == "multilabel" else torch.cat([row[3] for row in batch]).long()
    min_len = min(row[4].shape[1] for row in batch)
    audio = torch.stack([row[4][..., :min_len] for row in batch])
    if model == "mmtrvapt":
        poster = torch.stack([row[5] for row in batch]) if with_poster else None
        return text, segment, mask, img, tgt, audio, poster
    return text, segment, mask, img, tgt, audio


def model_forward(model, criterion, batch, model_name: str, gmu_gate: bool = False):
    """train.py:283-335 for the two hot-path models: note the argument order (txt, MASK, SEGMENT, ...) against the
    batch order (text, SEGMENT, MASK, ...).  Returns (loss, out, tgt[, gates])."""
    dev = next(model.parameters()).device
    if model_name == "mmtrvapt":
        txt, segment, mask, img, tgt, audio, poster = batch
        args = (txt.to(dev), mask.to(dev), segment.to(dev), img.to(dev), audio.to(dev), poster.to(dev))
    else:
        txt, segment, mask, img, tgt, audio = batch
        args = (txt.to(dev), mask.to(dev), segment.to(dev), img.to(dev), audio.to(dev))
    if gmu_gate:
        out, gates = model(*args, True)
    else:
        out, gates = model(*args), None
    tgt = tgt.to(dev)
    loss = criterion(out, tgt)
    return (loss, out, tgt, gates) if gmu_gate else (loss, out, tgt)
