"""Data-parallel gradient exchange of the HIP path: two processes share cuda:0 and talk over gloo (the box has one
GPU; the collective backend is irrelevant to what is checked).  Each rank runs forward/backward of the SAME model on its
half of a batch with bpmult_amd.distributed.GradSync hooked in (bucketed all-reduces on the communication stream as
the trunk finishes each gradient section, then x 1/world); afterwards every parameter's gradient must equal the
gradient of the mean loss over the global batch, computed by the same process without any exchange."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out, backend="gloo", compress="none"):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if backend == "nccl":                       # one GPU per rank over RCCL: the production path
        torch.cuda.set_device(rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    import bpmult_amd  # noqa: F401
    from bpmult_amd.distributed import GradSync
    from bpmult_amd.models import get_model
    from test_model_gpu import args_for
    torch.manual_seed(5)
    model = get_model(args_for("mmtrvat", hidden_sz=24, num_heads=4, layers=2, orig_d_l=32, num_vectors_l=64, num_vectors_a=64,
                               num_vectors_v=64))
    with torch.no_grad():
        for p in model.parameters():
            if p.dim() == 1:
                p.add_(0.1 * torch.randn_like(p))
    model.precision = "f32"
    model = model.cuda().train()
    g = torch.Generator().manual_seed(9)
    B = 4
    xs = [torch.randn(B, 30, 32, generator=g).cuda(), torch.randn(B, 64, 35, generator=g).cuda(), torch.randn(B, 50, 74, generator=g).cuda()]
    tgt = (torch.randn(B, 6, generator=g) > 0).float().cuda()
    sync = GradSync(model, bucket_bytes=1 << 16, compress=compress)   # small buckets: several all-reduces per section

    def run(sl):
        for p in model.parameters():
            p.grad = None
        loss = torch.nn.functional.binary_cross_entropy_with_logits(model(xs[0][sl], None, None, xs[1][sl], xs[2][sl]), tgt[sl])
        loss.backward()

    sync.active = False
    run(slice(0, B))
    full = {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}
    sync.active = True
    per = B // world
    if backend == "nccl":
        # with a gradient-accumulation micro-step in front (no exchange), each micro-step on half of the rank's shard:
        # sum of the two half-shard mean-loss gradients / 2 == the shard's mean-loss gradient
        half = per // 2
        lo = rank * per
        for p in model.parameters():
            p.grad = None
        for j, active in ((0, False), (1, True)):
            sync.active = active
            sl = slice(lo + j * half, lo + (j + 1) * half)
            loss = torch.nn.functional.binary_cross_entropy_with_logits(model(xs[0][sl], None, None, xs[1][sl], xs[2][sl]), tgt[sl])
            (loss / 2).backward()
            sync.finish()
    else:
        run(slice(rank * per, (rank + 1) * per))
        sync.finish()
    torch.cuda.synchronize()
    worst = 0.0
    for k, p in model.named_parameters():
        if k not in full:
            continue
        ref = full[k]
        if float(ref.abs().max()) < 1e-9:
            continue
        worst = max(worst, float((p.grad - ref).abs().max() / ref.abs().max()))
    out[rank] = worst
    dist.destroy_process_group()


def test_gradsync_two_ranks_equals_global_batch_gradient():
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    assert len(out) == world
    for r, e in out.items():
        assert e < 2e-4, (r, e)


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (RCCL over xGMI); the one-GPU box runs the gloo variant above")
@pytest.mark.parametrize("compress", ["none", "bf16"])
def test_gradsync_rccl_two_gpus_with_accumulation(compress):
    """backend "nccl" (= RCCL), one GPU per rank: per-layer slice all-reduces on the communication stream gated by main- and
    side-stream events, first-writer-stores launch tables beside in-flight exchanges, one accumulation micro-step; fp32 slices
    and bf16 copies (conftest.py runs these first wherever two GPUs are visible)."""
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, out, "nccl", compress), nprocs=world, join=True)
    assert len(out) == world
    for r, e in out.items():
        assert e < (2e-4 if compress == "none" else 2e-2), (r, e)


def test_gradsync_two_ranks_bf16_slices():
    """The same exchange with bf16 copies of the slices (the default above 1 GiB of gradients), two ranks on this GPU."""
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, out, "gloo", "bf16"), nprocs=world, join=True)
    assert len(out) == world
    for r, e in out.items():
        assert e < 2e-2, (r, e)
