#!/usr/bin/env python3
"""BPMulT hot-path benchmark on MI355X.

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is one training pass of the hot path over one synthetic batch that is
already resident in HBM: zero-grad, forward, BCE-with-logits loss, backward and,
for N > 1, the RCCL all-reduce of the gradients (the optimizer
step is reported separately as `optimizer_ms`, it is not part of the metric).

Default workload = the configuration BASELINE.json's metric is quoted on:
3-modal unaligned BPMulT (`mmtrvat`) at hidden 768 -- IEMOCAP shape, 12 heads,
8 layers, L/V/A = 20/500/400 zero-padded to 512, per-GPU batch 8, README dropout
rates, bf16 MFMA operands with f32 accumulation and an f32 residual stream.
Weak scaling: every rank processes its own batch of 8.

Rank 0 prints ONE JSON line.
* `roofline`: the kernel family with the largest share of a step, measured in a
  SEPARATE profiled pass after the timed region (bpm_prof_*: HIP events on the
  stream each launch goes to); achieved = algorithmic FLOPs / summed launch
  time; `bound` from its flop-per-byte against the 312 flop/B ridge; `traffic`
  = HBM bytes per launch from the committed PMC passes of this same command
  (profiles/r02_hbm_traffic_h768.json, tools/pmc_traffic.sh).
* `cpu_baseline`: the CPU oracle (oracle/bpmult_cpu.py, a port of the reference
  arithmetic) on this box's host cores, same model, 1 warm-up + 3 timed steps.
* `f32_mode` / `bf16_logits_relerr`: throughput of the exact-f32 mode (the one
  that meets the 1e-3 logit tolerance) and the measured logit error of the bf16
  mode against it at dropout 0, same weights and inputs.
* `kernel_point`: the north-star kernel point (d=768, H=6, T=S=50, B=64), timed stand-alone.
* secondary workloads (never `value`): `pruned_schedule` (exact dead-row
  elimination), `cfg1_hidden300` (BASELINE configs[1]).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time
from types import SimpleNamespace

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CONFIGS = {
    # name: (model, hidden, heads, layers, classes, orig dims l/v/a/p, raw lengths L/V/A, num_vectors l/a/v)
    "h768": dict(model="mmtrvat", hidden_sz=768, num_heads=12, layers=8, n_classes=6, orig_d_l=768, orig_d_v=35, orig_d_a=74,
                 orig_d_p=4096, L=20, V=500, A=400, nv=(512, 512, 512), batch=8,
                 desc="BPMulT hidden=768 3-modal unaligned: IEMOCAP-shape synthetic mmtrvat d=768 H=12 layers=8 L/V/A=20/500/400->512"),
    "cfg1": dict(model="mmtrvat", hidden_sz=300, num_heads=12, layers=8, n_classes=6, orig_d_l=768, orig_d_v=35, orig_d_a=74,
                 orig_d_p=4096, L=20, V=500, A=400, nv=(512, 512, 512), batch=8,
                 desc="IEMOCAP-shape synthetic 3-modal mmtrvat d=300 H=12 layers=8 L/V/A=20/500/400->512"),
    "cfg3": dict(model="mmtrvapt", hidden_sz=768, num_heads=6, layers=5, n_classes=13, orig_d_l=768, orig_d_v=4096, orig_d_a=96,
                 orig_d_p=4096, L=512, V=200, A=1000, nv=(512, 200, 200), batch=8,
                 desc="Moviescope-shape synthetic 4-modal mmtrvapt d=768 H=6 layers=5 L=512 V=A=200"),
    "k768": dict(model="mmtrvat", hidden_sz=768, num_heads=6, layers=5, n_classes=6, orig_d_l=768, orig_d_v=35, orig_d_a=74,
                 orig_d_p=4096, L=50, V=50, A=50, nv=(50, 50, 50), batch=64,
                 desc="kernel point: 3-modal d=768 H=6 layers=5 seq_len=50"),
}
BASELINE_CONFIG = {"h768": "metric string (hidden=768, 3-modal unaligned); shape of configs[1]", "cfg1": "configs[1]", "cfg3": "configs[2]"}
PEAK_BF16_TFLOPS = 2500.0       # MI355X dense bf16 MFMA (MI355X_MICROARCH.md, chip-level parameters)
PEAK_F32_TFLOPS = 157.3         # f32-input MFMA = the f32 vector rate
PEAK_HBM_GBS = 8000.0           # HBM3E spec
RIDGE_FLOP_PER_BYTE = PEAK_BF16_TFLOPS * 1e12 / (PEAK_HBM_GBS * 1e9)


def model_args(c, precision, dropout=True):
    p = (lambda v: v) if dropout else (lambda v: 0.0)
    return SimpleNamespace(model=c["model"], hidden_sz=c["hidden_sz"], num_heads=c["num_heads"], layers=c["layers"],
                           n_classes=c["n_classes"], orig_d_l=c["orig_d_l"], orig_d_v=c["orig_d_v"], orig_d_a=c["orig_d_a"],
                           orig_d_p=c["orig_d_p"], vonly=True, lonly=True, aonly=True, attn_dropout=p(0.1), attn_dropout_v=0.0,
                           attn_dropout_a=0.0, relu_dropout=p(0.1), res_dropout=p(0.1), out_dropout=0.0, embed_dropout=p(0.25),
                           attn_mask=True, hybrid=False, bert_model="unused", text_features=True, precision=precision,
                           num_vectors_l=c["nv"][0], num_vectors_a=c["nv"][1], num_vectors_v=c["nv"][2])


def synth_batch(c, B, seed, device):
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.randn(*s, generator=g)
    b = dict(xl=r(B, c["L"], c["orig_d_l"]), img=r(B, c["V"], c["orig_d_v"]),
             tgt=(torch.rand(B, c["n_classes"], generator=g) > 0.5).float())
    if c["model"] == "mmtrvapt":
        b["aud"] = r(B, 96, c["A"])
        b["post"] = r(B, c["orig_d_p"])
    else:
        b["aud"] = r(B, c["A"], c["orig_d_a"])
    return {k: v.to(device) for k, v in b.items()}


def run_model(model, b):
    if "post" in b:
        return model(b["xl"], None, None, b["img"], b["aud"], b["post"])
    return model(b["xl"], None, None, b["img"], b["aud"])


def usable_cores() -> int:
    """Host cores this process may actually use: affinity mask and cgroup CPU quota, capped at 16 (the GPU box's share)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, 16))


def cpu_model_name() -> str:
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(c, sample_B, warm, steps):
    """The CPU oracle (port of the reference arithmetic) on this box's host cores: fwd + bwd, train mode."""
    from oracle import bpmult_cpu as O
    torch.manual_seed(1234)
    four = c["model"] == "mmtrvapt"
    m = O.ModelCfg(c["hidden_sz"], c["num_heads"], c["layers"], c["n_classes"], orig_d_l=c["orig_d_l"], orig_d_v=c["orig_d_v"],
                   orig_d_a=c["orig_d_a"], orig_d_p=c["orig_d_p"], attn_dropout=0.1, relu_dropout=0.1, res_dropout=0.1,
                   embed_dropout=0.25, num_vectors_l=c["nv"][0], num_vectors_a=c["nv"][1], num_vectors_v=c["nv"][2])
    sd = {k: (torch.randn(s) * 0.05).requires_grad_(True) for k, s in O.model_param_shapes(m, four).items()}
    b = synth_batch(c, sample_B, 1234, "cpu")
    cores = usable_cores()
    torch.set_num_threads(cores)
    print(f"[bench] cpu_baseline: {cores} threads, batch {sample_B}, {warm} warm-up + {steps} timed step(s) ...", file=sys.stderr, flush=True)

    def step():
        for p in sd.values():
            p.grad = None
        if four:
            logits, _ = O.bpmult4_forward(sd, m, b["xl"], b["img"], O.audio_encoder(sd, b["aud"]), b["post"], training=True)
        else:
            logits, _ = O.bpmult3_forward(sd, m, b["xl"], b["img"], b["aud"], training=True)
        torch.nn.functional.binary_cross_entropy_with_logits(logits, b["tgt"]).backward()

    for _ in range(warm):
        step()
    t0 = time.perf_counter()
    for i in range(steps):
        step()
        print(f"[bench] cpu_baseline step {i + 1}/{steps}: {time.perf_counter() - t0:.1f} s elapsed", file=sys.stderr, flush=True)
    dt = (time.perf_counter() - t0) / steps
    return dict(value=round(sample_B / dt, 4), unit="samples/s", cores=cores, kind="port",
                sample=f"{warm} warm-up + {steps} timed fwd+bwd steps of the same model ({c['desc']}) at batch {sample_B} "
                       f"(fp32, train-mode dropout), {dt:.2f} s/step, torch {torch.__version__} CPU ops, {cpu_model_name()}")


def baseline_metric():
    """BASELINE.json's metric string, verbatim; the default workload (h768) is the configuration it names."""
    try:
        with open(os.path.join(ROOT, "BASELINE.json")) as f:
            return json.load(f)["metric"]
    except (OSError, KeyError, ValueError):
        return "training samples/sec (fwd+bwd) BPMulT hidden=768, 3-modal unaligned, at 1/2/4/8 MI355X"


KIND_TO_FAMILY = {"gemm_nt": "gemm<NT>", "gemm_nn": "gemm<NN>", "gemm_tn": "gemm<TN>",
                  "attn_fwd": "attn_fwd_kernel", "attn_bwd_dq": "attn_bwd_dq_kernel", "attn_bwd_dkv": "attn_bwd_dkv_kernel"}
TRAFFIC_FILES = {"h768": "r02_hbm_traffic_h768.json", "cfg1": "r01_hbm_traffic.json"}


def hbm_traffic(config, kind):
    """HBM bytes per launch of a kernel family, from the committed PMC passes (tools/pmc_traffic.sh: rocprofv3 --pmc
    FETCH_SIZE / WRITE_SIZE in separate runs of this same workload, FETCH_SIZE doubled as MI355X_MICROARCH.md
    prescribes for gfx950).  None when no measurement of this configuration is committed."""
    path = os.path.join(ROOT, "profiles", TRAFFIC_FILES.get(config, "_none_"))
    try:
        with open(path) as f:
            ks = json.load(f)["kernels"]
        fam = KIND_TO_FAMILY[kind]
        tot, n = 0.0, 0
        for name, k in ks.items():            # a family may run as several kernels (tile configurations)
            if name == fam or name.startswith(fam + ":") or (fam.startswith("gemm<") and fam[5:7] in name and "gemm" in name):
                tot += k["hbm_bytes_per_launch"] * k.get("launches", 1)
                n += k.get("launches", 1)
        return round(tot / n) if n else None
    except (OSError, KeyError, ValueError):
        return None


def timed(step, n, world, dev):
    import torch.distributed as dist
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        out = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt, out


def kernel_point(dev):
    """North-star kernel point: crossmodal attention blocks at d=768, H=6 (head_dim 128), T=S=50, B=64, six encoders per
    launch -- projections + attention + output projection of one layer, forward and backward, timed stand-alone."""
    try:
        from tools.kernel_point import measure
    except ImportError:
        return None
    return measure(dev)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="h768", choices=sorted(CONFIGS))
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch (default: the config's)")
    ap.add_argument("--precision", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip pruned / f32-mode / cfg1 / kernel-point figures")
    ap.add_argument("--grad-compress", default="none", choices=["none", "bf16"], help="N > 1: all-reduce bf16 copies of the gradient sections")
    ap.add_argument("--cpu-batch", type=int, default=1)
    ap.add_argument("--cpu-steps", type=int, default=3)
    a = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            sys.exit("bench.py --gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)")
    import torch.distributed as dist
    # BENCH_REHEARSAL=1: all ranks share cuda:0 and talk over gloo -- only to exercise the N > 1 code path
    # (sectioned all-reduce on a side stream, rank-max timing) on a one-GPU box; never a measurement.
    rehearsal = os.environ.get("BENCH_REHEARSAL") == "1"
    if rehearsal:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    backend = None
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        backend = "gloo" if rehearsal else "nccl"
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    import bpmult_amd
    from bpmult_amd import _lib
    from bpmult_amd.distributed import GradSync
    from bpmult_amd.models import get_model
    from bpmult_amd.optim import FusedAdam

    c = CONFIGS[a.config]
    B = a.batch or c["batch"]
    secondary = world == 1 and not a.no_secondary and not a.batch
    torch.manual_seed(1234 + rank)
    model = get_model(model_args(c, a.precision)).to(dev).train()
    if world > 1:                                     # same replica everywhere: the flat trunk master in one message, the tail per tensor
        st = model._ensure_store()
        dist.broadcast(st.master, 0)
        for n, p in model.named_parameters():
            if n not in st.params:
                dist.broadcast(p.data, 0)
    opt = FusedAdam(model, lr=1e-3)          # one kernel over the flat trunk buffers (reported as optimizer_ms, not timed)
    sync = GradSync(model, optimizer=opt, compress=a.grad_compress)   # 1/world is folded into the optimizer's grad_scale
    batch = synth_batch(c, B, 1234 + rank, dev)
    crit = torch.nn.BCEWithLogitsLoss()

    def step():
        for p in model.parameters():
            p.grad = None
        loss = crit(run_model(model, batch), batch["tgt"])
        loss.backward()
        sync.finish()
        return loss

    for _ in range(max(a.warmup, 1)):
        step()
    sync.reset_stats()
    dt, loss = timed(step, a.steps, world, dev)
    comm = sync.stats()

    # ---- profiled pass (NOT the timed region): per-kernel-kind launch times with HIP events on the launch streams
    L = _lib.lib()
    kinds = _lib.PROF_KINDS
    L.bpm_prof_enable(sum(1 << k for k in kinds.values()))
    nprof = 3
    for _ in range(nprof):
        step()
    torch.cuda.synchronize()
    tot = {}
    for name, k in kinds.items():
        ms, work, n = C.c_double(), C.c_double(), C.c_int()
        _lib.check(L.bpm_prof_collect(k, C.byref(ms), C.byref(work), C.byref(n)), "bpm_prof_collect")
        tot[name] = (ms.value, work.value, n.value)
    L.bpm_prof_enable(0)
    dom = max(tot, key=lambda k: tot[k][0])
    dms, dwork, dn = tot[dom]

    # optimizer step, reported separately (not part of the fwd+bwd metric); one untimed step allocates the moments
    # (the bf16 weight shadows are re-derived when the masters change, i.e. once per optimizer step: counted here)
    opt.step()
    model._store.refresh_shadows()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(3):
        opt.step()
        model._store.refresh_shadows()
    torch.cuda.synchronize()
    opt_ms = (time.perf_counter() - t1) / 3 * 1e3

    extra = {}
    if secondary and c["model"] == "mmtrvat":
        # same step with the exact dead-row elimination of SURVEY A.10 (level-2 encoders and Fusion-GMUs on query rows
        # {0, N-1} only; logits and gradients identical, tests/test_model_gpu.py)
        model.set_prune_unused_rows(True)
        for _ in range(2):
            step()
        dtp, _ = timed(step, min(a.steps, 10), world, dev)
        extra["pruned_schedule"] = {"value": round(B * min(a.steps, 10) / dtp, 3), "unit": "samples/s",
                                    "ms_per_step": round(dtp / min(a.steps, 10) * 1e3, 3),
                                    "what": "same step, level-2 encoders + Fusion-GMUs on query rows {0, N-1} only (exact; SURVEY A.10)"}
        model.set_prune_unused_rows(False)
    if secondary and a.precision == "bf16":
        # both modes in one record: the bf16 mode's logit error against the f32 mode (the one held to 1e-3 against the
        # reference, tests/test_model_gpu.py) at dropout 0 on the same weights and inputs, and the f32 mode's throughput
        sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
        del model, opt, sync
        torch.cuda.empty_cache()
        logits = {}
        for prec in ("bf16", "f32"):
            m0 = get_model(model_args(c, prec, dropout=False)).to(dev)
            m0.load_state_dict(sd)
            m0.train()
            with torch.no_grad():
                logits[prec] = run_model(m0, batch).float()
            if prec == "f32":
                mf = get_model(model_args(c, "f32")).to(dev).train()
                mf.load_state_dict(sd)

                def stepf():
                    for p in mf.parameters():
                        p.grad = None
                    crit(run_model(mf, batch), batch["tgt"]).backward()

                stepf()
                nf = 3
                dtf, _ = timed(stepf, nf, 1, dev)
                extra["f32_mode"] = {"value": round(B * nf / dtf, 3), "unit": "samples/s", "ms_per_step": round(dtf / nf * 1e3, 2),
                                     "what": "same workload with exact f32 MFMA end to end (--precision f32): the mode that meets the "
                                             "1e-3 logit tolerance against the reference"}
                del mf
            del m0
            torch.cuda.empty_cache()
        ref = logits["f32"]
        extra["bf16_logits_relerr"] = {"max_abs_over_max_abs": float((logits["bf16"] - ref).abs().max() / ref.abs().max()),
                                       "rel_l2": float((logits["bf16"] - ref).norm() / ref.norm()),
                                       "what": "bf16-mode logits vs f32-mode HIP logits, dropout 0, same weights and inputs"}
        model = None
    if secondary and a.config == "h768":
        kp = kernel_point(dev)
        if kp is not None:
            extra["kernel_point"] = kp
        c1 = CONFIGS["cfg1"]
        m1 = get_model(model_args(c1, a.precision)).to(dev).train()
        b1 = synth_batch(c1, c1["batch"], 1234, dev)

        def step1():
            for p in m1.parameters():
                p.grad = None
            crit(run_model(m1, b1), b1["tgt"]).backward()

        for _ in range(3):
            step1()
        d1, _ = timed(step1, 10, 1, dev)
        extra["cfg1_hidden300"] = {"value": round(c1["batch"] * 10 / d1, 3), "unit": "samples/s", "ms_per_step": round(d1 / 10 * 1e3, 3),
                                   "workload": c1["desc"], "baseline_config": "configs[1]"}
        del m1, b1
        torch.cuda.empty_cache()

    if rank == 0:
        peak = PEAK_BF16_TFLOPS if a.precision == "bf16" else PEAK_F32_TFLOPS
        ach = dwork / (dms * 1e-3) / 1e12 if dms > 0 else 0.0
        traffic = hbm_traffic(a.config, dom) if (a.precision == "bf16" and not a.batch) else None
        roof = {"kernel": dom, "launches_profiled": dn, "avg_launch_ms": round(dms / max(dn, 1), 4),
                "flops_per_launch": dwork / max(dn, 1), "traffic": traffic}
        fpb = (dwork / max(dn, 1)) / traffic if traffic else None
        if fpb is not None and fpb < RIDGE_FLOP_PER_BYTE and a.precision == "bf16":
            # under the ridge: the binding roof is HBM; achieved = measured bytes / time
            gbs = traffic / (dms / max(dn, 1) * 1e-3) / 1e9
            roof.update({"bound": "hbm", "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(gbs / PEAK_HBM_GBS, 4),
                         "flop_per_byte": round(fpb, 1), "mfma_tflops": round(ach, 2), "mfma_frac": round(ach / peak, 4)})
        else:
            roof.update({"bound": "mfma", "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4),
                         "flop_per_byte": round(fpb, 1) if fpb else None})
        out = {
            "metric": baseline_metric(),
            "value": round(world * B * a.steps / dt, 3), "unit": "samples/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(dt / a.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": a.precision, "data": "synthetic",
            "config": {"workload": c["desc"], "baseline_config": BASELINE_CONFIG.get(a.config, a.config), "per_gpu_batch": B,
                       "global_batch": B * world, "parallelism": f"dp{world}", "world": world, "backend": backend,
                       "dropout": "README rates (attn .1/0/0, relu .1, res .1, embed .25)",
                       "loss": float(loss.detach()), "optimizer_ms": round(opt_ms, 3),
                       "kernel_ms_per_step_profiled": {k: round(v[0] / nprof, 3) for k, v in tot.items()}},
            "roofline": roof,
        }
        if world > 1:
            out["config"]["grad_exchange"] = comm
        out.update(extra)
        print("[bench] gpu part done: " + json.dumps({k: out[k] for k in ("value", "ms_per_step", "roofline")}), file=sys.stderr, flush=True)
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(c, a.cpu_batch, 1, a.cpu_steps)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
