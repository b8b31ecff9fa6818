"""Data-parallel correctness of the gradient exchange on CPU (gloo, world_size 2).

The hot path shards on the batch axis with one all-reduce(sum) of gradients
followed by 1/world.  Two ranks each run the CPU oracle on
half of a batch; after bpmult_amd.distributed.reduce_gradients_cpu every rank
must hold the gradient of the mean loss over the GLOBAL batch."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from detgen import det, det_param
    import bpmult_amd  # noqa: F401
    from bpmult_amd.distributed import reduce_gradients_cpu
    from oracle import bpmult_cpu as O
    m = O.ModelCfg(16, 2, 1, 3, orig_d_l=8, orig_d_v=5, orig_d_a=6, num_vectors_l=12, num_vectors_a=12, num_vectors_v=12)
    shapes = {k: s for k, s in O.model_param_shapes(m, False).items() if not k.startswith("transfm_")}
    sd = {k: torch.from_numpy(det_param("dp." + k, s)).requires_grad_(True) for k, s in shapes.items()}
    B = 4
    xl, img, aud = (torch.from_numpy(det("dp." + n, s)) for n, s in (("xl", (B, 7, 8)), ("img", (B, 11, 5)), ("aud", (B, 9, 6))))
    tgt = (torch.from_numpy(det("dp.tgt", (B, 3))) > 0).float()

    def grads(sl):
        for p in sd.values():
            p.grad = None
        logits, _ = O.bpmult3_forward(sd, m, xl[sl], img[sl], aud[sl])
        torch.nn.functional.binary_cross_entropy_with_logits(logits, tgt[sl]).backward()
        return {k: p.grad.clone() for k, p in sd.items()}

    full = grads(slice(0, B))
    per = B // world
    grads(slice(rank * per, (rank + 1) * per))
    reduce_gradients_cpu(list(sd.values()), world)
    err = max(float((sd[k].grad - full[k]).abs().max() / (full[k].abs().max() + 1e-12)) for k in sd)
    out[rank] = err
    dist.destroy_process_group()


def test_shard_allreduce_mean_equals_global_batch_gradient():
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    assert len(out) == world
    for r, e in out.items():
        assert e < 1e-5, (r, e)
