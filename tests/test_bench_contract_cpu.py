"""bench.py host-side contract pieces that need no GPU: the metric string is BASELINE.json's verbatim, the committed
HBM-traffic measurement resolves for the dominant kernel families, every workload names its shape, and the CLI keeps
the driver's flags."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _bench():
    argv, sys.argv = sys.argv, ["bench.py"]
    try:
        import bench
    finally:
        sys.argv = argv
    return bench


def test_metric_is_baseline_json_verbatim():
    b = _bench()
    with open(os.path.join(ROOT, "BASELINE.json")) as f:
        assert b.baseline_metric() == json.load(f)["metric"]


def test_traffic_file_covers_the_profiled_kernel_families():
    b = _bench()
    for cfg in ("h768", "cfg1"):
        for kind in b.KIND_TO_FAMILY:
            t = b.hbm_traffic(cfg, kind)
            assert t is None or t > 0
    assert b.hbm_traffic("cfg1", "gemm_tn") and b.hbm_traffic("cfg1", "gemm_tn") > 5e7      # committed PMC measurement present
    if os.path.exists(os.path.join(ROOT, "profiles", b.TRAFFIC_FILES["h768"])):
        assert b.hbm_traffic("h768", "gemm_dma_tn") > 5e7 and b.hbm_traffic("h768", "gemm_dma_nt") > 5e7


def test_default_workload_is_the_metric_configuration():
    """BASELINE.json's metric is quoted on hidden 768, 3-modal unaligned: that is what `value` must describe."""
    b = _bench()
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True, timeout=120)
    assert "default: h768" in out.stdout.replace("(default", "default").replace("'", "") or "h768" in out.stdout
    c = b.CONFIGS["h768"]
    assert c["hidden_sz"] == 768 and c["model"] == "mmtrvat" and (c["L"], c["V"], c["A"]) == (20, 500, 400)
    import inspect
    assert 'default="h768"' in inspect.getsource(b.main)


def test_workloads_and_cli():
    b = _bench()
    assert {"cfg1", "cfg3", "cfg4", "cfg5", "h768", "k768"} <= set(b.CONFIGS)      # every BASELINE.json config has a line
    for c in b.CONFIGS.values():
        assert c["desc"] and c["batch"] >= 1 and len(c["nv"]) == 3
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0
    for flag in ("--gpus", "--steps", "--warmup"):
        assert flag in out.stdout
