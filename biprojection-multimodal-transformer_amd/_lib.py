"""Build + load libbpmult_hip.so (the C ABI in include/bpmult_hip.h) and bind it
with ctypes.  There is deliberately NO fallback: if the library is missing or a
symbol does not resolve, importing the product path raises.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

# torch must be imported BEFORE the library is dlopen'ed: libbpmult_hip.so then binds to the
# HIP runtime torch has already loaded (one runtime, one device context per process).  Loaded
# the other way round, the process ends up with two HIP runtimes and launches fail with
# hipErrorNoDevice.
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.environ.get("BPMULT_LIB", os.path.join(_HERE, "libbpmult_hip.so"))   # override: kernel-variant experiments
SOURCES = ("gemm.hip", "attention.hip", "rowops.hip", "tail.hip", "frontend.hip", "prof.hip")
HEADERS = ("bpm_common.h", "bpm_prof.h", "gemm_dma.h")
ARCH = "gfx950"
PROF_KINDS = {"gemm_nt": 0, "gemm_nn": 1, "gemm_tn": 2, "attn_fwd": 3, "attn_bwd_dq": 4, "attn_bwd_dkv": 5,
              "gemm_dma_nt": 13, "gemm_dma_nn": 14, "gemm_dma_tn": 15}      # gemm_*: the 128 x 64 kernel; gemm_dma_*: the LDS-DMA kernel

ABI_VERSION = 3                   # == BPM_ABI_VERSION of include/bpmult_hip.h; lib() refuses any other library
# -DBPM_LAB build: the same kernels plus the two process-global tuning hooks (bpm_debug_gemm_force / bpm_debug_attn_pair)
# that tools/gemm_lab.py, tools/attn_lab.py and three kernel tests use; never loaded by the product path
LAB_LIB_PATH = os.path.join(_HERE, "..", "build", "lab", "libbpmult_hip_lab.so")
BPM_F32, BPM_BF16, BPM_BF16X3 = 0, 1, 2      # BPM_BF16X3: bpm_gemm_grouped only (split-bf16 operands, three products)
GEMM_NT, GEMM_NN, GEMM_TN = 0, 1, 2
OUT_F32, OUT_CT, OUT_HEADS = 0, 1, 2
F_ACCUM, F_RELU, F_ATOMIC, F_KPAD, F_BACKGROUND = 1, 2, 4, 8, 16     # F_KPAD = BPM_GEMM_KPAD_ZERO
F_A_OVERLAP, F_B_OVERLAP = 32, 64                                     # BPM_GEMM_A_OVERLAP / _B_OVERLAP
F_CT_NARROW, F_BATCHED = 128, 256                                     # BPM_GEMM_CT_NARROW / _BATCHED
LN_OUT_F32 = 2
MAX_GROUP = 18
SEED_INDIRECT = 1 << 63          # seed = SEED_INDIRECT | device address of a uint64 (include/bpmult_hip.h)
GEMM_MAX_GROUP = 24


def build(force: bool = False, verbose: bool = False, out: str | None = None, flags: tuple = (), objdir: str | None = None) -> str:
    """hipcc --offload-arch=gfx950: every csrc/*.hip to an object file (in parallel; only the stale ones), then one shared
    library in-tree.  `out` / `flags` / `objdir`: variant builds for the lab tools (e.g. -DBPM_DMA_ABLATE=4).  Safe when
    several ranks call it at once: one builder at a time per object directory (file lock), objects and the library are
    written under a temporary name and renamed into place."""
    import fcntl
    import hashlib
    from concurrent.futures import ThreadPoolExecutor
    out = out or LIB_PATH
    # object directory named from a STABLE digest of the flags (str hashes are randomised per process)
    tag = ("_" + hashlib.sha1(" ".join(flags).encode()).hexdigest()[:10]) if flags else ""
    objdir = objdir or os.path.join(_HERE, "..", "build", "obj" + tag)
    srcs = [os.path.join(CSRC, s) for s in SOURCES]
    hdrs = [os.path.join(CSRC, h) for h in HEADERS] + [os.path.join(_HERE, "..", "include", "bpmult_hip.h")]
    newest_hdr = max(os.path.getmtime(h) for h in hdrs)

    def fresh():
        return os.path.exists(out) and all(os.path.getmtime(out) >= os.path.getmtime(d) for d in srcs + hdrs)

    if not force and fresh():
        return out
    os.makedirs(objdir, exist_ok=True)
    os.makedirs(os.path.dirname(os.path.abspath(out)), exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    common = [hipcc, f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC"] + list(flags)

    def compile_one(src):
        obj = os.path.join(objdir, os.path.basename(src)[:-4] + ".o")
        if not force and os.path.exists(obj) and os.path.getmtime(obj) >= max(os.path.getmtime(src), newest_hdr):
            return obj
        tmp = f"{obj}.{os.getpid()}.tmp"
        cmd = common + ["-c", src, "-o", tmp]
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True)
        os.replace(tmp, obj)
        return obj

    with open(os.path.join(objdir, ".lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and fresh():              # another rank built it while we waited
                return out
            with ThreadPoolExecutor(max_workers=min(4, os.cpu_count() or 1)) as ex:
                objs = list(ex.map(compile_one, srcs))
            tmp = f"{out}.{os.getpid()}.tmp"
            cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", tmp] + objs
            if verbose:
                print(" ".join(cmd))
            subprocess.run(cmd, check=True)
            os.replace(tmp, out)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return out


def build_lab(force: bool = False) -> str:
    """The -DBPM_LAB variant (tuning hooks exported) for tools/ and the kernel tests that pin a tile configuration."""
    return build(force=force, out=LAB_LIB_PATH, flags=("-DBPM_LAB",))


class GemmProblem(C.Structure):
    _fields_ = [("A", C.c_void_p), ("B", C.c_void_p), ("C", C.c_void_p),
                ("M", C.c_int), ("N", C.c_int), ("K", C.c_int),
                ("lda", C.c_int), ("ldb", C.c_int), ("ldc", C.c_int),
                ("bias_n", C.c_void_p), ("bias_m", C.c_void_p),
                ("resid", C.c_void_p), ("ldr", C.c_int),
                ("gate", C.c_void_p), ("ldg", C.c_int), ("gate_scale", C.c_float),
                ("alpha", C.c_float), ("drop_p", C.c_float), ("drop_site", C.c_uint32),
                ("colsum", C.c_void_p),
                ("flags", C.c_int), ("out_kind", C.c_int), ("splitk", C.c_int),
                ("heads_B", C.c_int), ("heads_H", C.c_int), ("heads_T", C.c_int),
                ("heads_dh", C.c_int), ("heads_dhp", C.c_int), ("colsum_a", C.c_void_p),
                ("batch", C.c_int), ("batch_stride_a", C.c_int), ("batch_stride_b", C.c_int), ("batch_stride_c", C.c_int)]


class AttnProblem(C.Structure):
    _fields_ = [("Q", C.c_void_p), ("K", C.c_void_p), ("V", C.c_void_p),
                ("O", C.c_void_p), ("ldo", C.c_int), ("lse", C.c_void_p),
                ("dO", C.c_void_p), ("delta", C.c_void_p),
                ("dQ", C.c_void_p), ("lddq", C.c_int),
                ("dK", C.c_void_p), ("lddk", C.c_int),
                ("dV", C.c_void_p), ("lddv", C.c_int),
                ("B", C.c_int), ("H", C.c_int), ("T", C.c_int), ("S", C.c_int),
                ("dh", C.c_int), ("dhp", C.c_int), ("mask_off", C.c_int),
                ("dq_scale", C.c_float), ("drop_p", C.c_float), ("drop_site", C.c_uint32),
                ("q_pos0", C.c_int), ("q_stride", C.c_int),
                ("dS", C.c_void_p), ("Pd", C.c_void_p), ("xs_b", C.c_int), ("xs_h", C.c_int), ("xs_q", C.c_int)]


class PackProblem(C.Structure):
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("g", C.c_void_p), ("ldg", C.c_int), ("dsrc", C.c_void_p),
                ("B", C.c_int), ("T", C.c_int), ("C", C.c_int), ("ld", C.c_int),
                ("drop_p", C.c_float), ("drop_site", C.c_uint32)]


class PackDesc(C.Structure):
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("rows", C.c_int), ("cols", C.c_int),
                ("ld", C.c_int), ("src_ld", C.c_int), ("dst_ld", C.c_int), ("blk0", C.c_uint),
                ("colscale", C.c_void_p)]


class FoldDesc(C.Structure):
    _fields_ = [("W", C.c_void_p), ("beta", C.c_void_p), ("b", C.c_void_p), ("out", C.c_void_p),
                ("rows", C.c_int), ("cols", C.c_int), ("ldw", C.c_int), ("blk0", C.c_uint)]


class UnfoldDesc(C.Structure):
    _fields_ = [("dWf", C.c_void_p), ("dbf", C.c_void_p), ("W", C.c_void_p), ("gamma", C.c_void_p), ("beta", C.c_void_p),
                ("dW", C.c_void_p), ("dbias", C.c_void_p), ("dgamma", C.c_void_p), ("dbeta", C.c_void_p),
                ("rows", C.c_int), ("cols", C.c_int), ("ldw", C.c_int), ("blk0", C.c_uint)]


class ZeroDesc(C.Structure):
    _fields_ = [("p", C.c_void_p), ("n", C.c_uint), ("blk0", C.c_uint)]


class AdamSeg(C.Structure):
    _fields_ = [("off4", C.c_size_t), ("n4", C.c_uint), ("blk0", C.c_uint), ("dst", C.c_void_p),
                ("rows", C.c_int), ("cols", C.c_int), ("dst_ld", C.c_int), ("pad_", C.c_int)]


class SplitProblem(C.Structure):
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("R", C.c_int), ("C", C.c_int), ("ld", C.c_int), ("ldp", C.c_int)]


class EmbedProblem(C.Structure):
    _fields_ = [("x", C.c_void_p), ("out", C.c_void_p), ("T", C.c_int), ("B", C.c_int), ("accumulate", C.c_int),
                ("drop_p", C.c_float), ("drop_site", C.c_uint32), ("pos0", C.c_int), ("pos_stride", C.c_int)]


class LnProblem(C.Structure):
    _fields_ = [("x", C.c_void_p), ("gamma", C.c_void_p), ("beta", C.c_void_p),
                ("out", C.c_void_p), ("ldo", C.c_int), ("out_f32", C.c_int),
                ("mean", C.c_void_p), ("rstd", C.c_void_p), ("R", C.c_int),
                ("dy", C.c_void_p), ("ldy", C.c_int), ("add", C.c_void_p), ("dx", C.c_void_p),
                ("dgamma", C.c_void_p), ("dbeta", C.c_void_p),
                ("cast", C.c_void_p), ("ldc", C.c_int), ("cast_colsum", C.c_void_p),
                ("drop_p", C.c_float), ("drop_site", C.c_uint32)]


class CastProblem(C.Structure):
    _fields_ = [("a", C.c_void_p), ("lda", C.c_int), ("a_is_ct", C.c_int),
                ("b", C.c_void_p), ("ldb", C.c_int),
                ("dst_ct", C.c_void_p), ("ldd", C.c_int), ("ct_cols", C.c_int),
                ("dst_f32", C.c_void_p), ("ldf", C.c_int),
                ("colsum", C.c_void_p), ("R", C.c_int), ("C", C.c_int),
                ("drop_p", C.c_float), ("drop_site", C.c_uint32)]


class GmuProblem(C.Structure):
    _fields_ = [("a1", C.c_void_p), ("a2", C.c_void_p), ("ag", C.c_void_p), ("x1", C.c_void_p), ("x2", C.c_void_p),
                ("out", C.c_void_p), ("dout", C.c_void_p), ("da1", C.c_void_p), ("da2", C.c_void_p), ("dag", C.c_void_p),
                ("ldg", C.c_int), ("dx1", C.c_void_p), ("dx2", C.c_void_p), ("R", C.c_int)]


class AddnProblem(C.Structure):
    _fields_ = [("out", C.c_void_p), ("src", C.c_void_p * 8), ("n_in", C.c_int), ("count", C.c_size_t)]


class ExpandProblem(C.Structure):
    _fields_ = [("q", C.c_void_p), ("dO", C.c_void_p), ("qexp", C.c_void_p), ("dOexp", C.c_void_p), ("Pd", C.c_void_p),
                ("dbias", C.c_void_p), ("B", C.c_int), ("H", C.c_int), ("T", C.c_int), ("S", C.c_int), ("dh", C.c_int),
                ("dhp", C.c_int), ("ld", C.c_int)]


class TailDesc(C.Structure):
    _fields_ = [("B", C.c_int), ("d", C.c_int), ("n", C.c_int), ("C", C.c_int), ("N", C.c_int * 3),
                ("top", C.c_void_p * 3), ("mid", C.c_void_p * 3), ("extra", C.c_void_p),
                ("Wh", C.c_void_p * 4), ("Wg", C.c_void_p * 4),
                ("W1", C.c_void_p), ("b1", C.c_void_p), ("W2", C.c_void_p), ("b2", C.c_void_p), ("Wo", C.c_void_p), ("bo", C.c_void_p),
                ("out_dropout", C.c_float), ("drop_site", C.c_uint32),
                ("x", C.c_void_p), ("z", C.c_void_p), ("t", C.c_void_p), ("h", C.c_void_p), ("p1", C.c_void_p), ("y", C.c_void_p),
                ("logits", C.c_void_p)]


class TailGrads(C.Structure):
    _fields_ = [("dlogits", C.c_void_p), ("dz", C.c_void_p), ("dWh", C.c_void_p * 4), ("dWg", C.c_void_p * 4),
                ("dW1", C.c_void_p), ("db1", C.c_void_p), ("dW2", C.c_void_p), ("db2", C.c_void_p), ("dWo", C.c_void_p), ("dbo", C.c_void_p),
                ("dtop", C.c_void_p * 3), ("dmid", C.c_void_p * 3), ("dextra", C.c_void_p),
                ("dy", C.c_void_p), ("dp1", C.c_void_p), ("dh", C.c_void_p), ("dzp", C.c_void_p), ("dtp", C.c_void_p), ("dx", C.c_void_p)]


_P, _I, _F, _U64, _U32 = C.c_void_p, C.c_int, C.c_float, C.c_uint64, C.c_uint32

# every symbol include/bpmult_hip.h declares: name -> argtypes
SIGNATURES = {
    "bpm_version": [],
    "bpm_error_string": [_I],
    "bpm_gemm_grouped": [_I, _I, C.POINTER(GemmProblem), _I, _U64, _P],
    "bpm_attn_fwd": [_I, C.POINTER(AttnProblem), _I, _U64, _P],
    "bpm_attn_bwd": [_I, C.POINTER(AttnProblem), _I, _U64, _P],
    "bpm_attn_bwd_dq": [_I, C.POINTER(AttnProblem), _I, _U64, _P],
    "bpm_attn_bwd_dkv": [_I, C.POINTER(AttnProblem), _I, _U64, _P],
    "bpm_pack_rows_fwd": [_I, C.POINTER(PackProblem), _I, _U64, _P],
    "bpm_pack_rows_bwd": [C.POINTER(PackProblem), _I, _U64, _P],
    "bpm_pack_weights": [_I, _P, _I, C.c_uint, _P],
    "bpm_fold_bias": [C.c_void_p, _I, C.c_uint, _P],
    "bpm_unfold_grads": [C.c_void_p, _I, C.c_uint, _I, _P],
    "bpm_zero_segment_blocks": [C.c_uint],
    "bpm_zero_segments": [C.c_void_p, _I, C.c_uint, _P],
    "bpm_embed_pos_fwd": [C.POINTER(EmbedProblem), _I, _P, _I, _I, _F, _U64, _P],
    "bpm_embed_pos_bwd": [C.POINTER(EmbedProblem), _I, _I, _F, _U64, _P],
    "bpm_ln_fwd": [_I, C.POINTER(LnProblem), _I, _I, _F, _P],
    "bpm_ln_bwd": [_I, C.POINTER(LnProblem), _I, _I, _U64, _P],
    "bpm_ln_bwd_ws": [_I, C.POINTER(LnProblem), _I, _I, _U64, _P, C.c_size_t, _P],
    "bpm_ln_bwd_ws_bytes": [_I, _I],
    "bpm_rows_cast": [_I, C.POINTER(CastProblem), _I, _U64, _P],
    "bpm_add_n": [C.POINTER(AddnProblem), _I, _P],
    "bpm_expand_heads": [_I, C.POINTER(ExpandProblem), _I, _P],
    "bpm_split_rows": [C.POINTER(SplitProblem), _I, _P],
    "bpm_gmu2_fwd": [C.POINTER(GmuProblem), _I, _I, _P],
    "bpm_gmu2_bwd": [_I, C.POINTER(GmuProblem), _I, _I, _P],
    "bpm_signal_pack": [_I, _P, _P, _I, _I, _I, C.c_int64, C.c_int64, C.c_int64, _I, _I, C.c_int64, _I, _P],
    "bpm_signal_unpack": [_P, _P, _I, _I, _I, C.c_int64, C.c_int64, C.c_int64, _I, _I, _I, _P],
    "bpm_adaptive_pool1d_fwd": [_P, _P, _I, _I, _I, _I, _P],
    "bpm_adaptive_pool1d_bwd": [_P, _P, _I, _I, _I, _I, _P],
    "bpm_tail_fwd": [C.POINTER(TailDesc), _U64, _P],
    "bpm_tail_bwd": [C.POINTER(TailDesc), C.POINTER(TailGrads), _P],
    "bpm_adam_step": [_P, _P, _P, _P, C.c_size_t, _F, _F, _F, _F, _F, _I, _F, _I, _P],
    "bpm_adam_blocks": [C.c_size_t],
    "bpm_adam_step_table": [_I, _P, _I, C.c_uint, _P, _P, _P, _P, _F, _F, _F, _F, _F, _I, _F, _I, _P],
    "bpm_stream_create": [_I, C.POINTER(C.c_void_p)],
    "bpm_stream_priority_range": [C.POINTER(_I), C.POINTER(_I)],
    "bpm_prof_enable": [C.c_uint],
    "bpm_prof_collect": [_I, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(_I)],
    "bpm_prof_collect2": [_I, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(_I)],
}

_lib = None


class HipLibraryError(RuntimeError):
    pass


def load(path: str) -> C.CDLL:
    """dlopen + bind one build of the library; raises (never falls back) on a missing file, a missing symbol or another
    ABI version (a stale build would be called with shifted arguments)."""
    if not os.path.exists(path):
        raise HipLibraryError(
            f"{path} not found: run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). The BPMulT hot path has no CPU/PyTorch fallback.")
    L = C.CDLL(path)
    for name, args in SIGNATURES.items():
        fn = getattr(L, name)            # AttributeError if the symbol is missing
        fn.argtypes = args
        fn.restype = C.c_char_p if name == "bpm_error_string" else C.c_size_t if name == "bpm_ln_bwd_ws_bytes" else C.c_int
    v = L.bpm_version()
    if v != ABI_VERSION:
        raise HipLibraryError(f"{path}: ABI version {v}, this package binds version {ABI_VERSION} (include/bpmult_hip.h): rebuild it")
    return L


def lib() -> C.CDLL:
    """The loaded library; raises (never falls back) when it is absent."""
    global _lib
    if _lib is None:
        _lib = load(LIB_PATH)
    return _lib


class lab_library:
    """Context manager for tools/ and tests: route this process's launches through the -DBPM_LAB build (which exports
    bpm_debug_gemm_force / bpm_debug_attn_pair) and back.  Raises HipLibraryError when that build is absent."""

    def __enter__(self) -> C.CDLL:
        global _lib
        self._prev = _lib
        L = load(LAB_LIB_PATH)
        L.bpm_debug_gemm_force.argtypes = [C.c_int]
        L.bpm_debug_attn_pair.argtypes = [C.c_int]
        _lib = L
        return L

    def __exit__(self, *exc) -> None:
        global _lib
        try:
            _lib.bpm_debug_gemm_force(-1)
            _lib.bpm_debug_attn_pair(7)
        finally:
            _lib = self._prev


_prof_mask = 0


def prof_enable(mask: int) -> None:
    """bpm_prof_enable; remembered host-side: while the launch profiler records, steps run eagerly (a graph replay
    launches nothing from the host, so there would be nothing to bracket with events)."""
    global _prof_mask
    lib().bpm_prof_enable(mask)
    _prof_mask = mask


def prof_enabled() -> bool:
    return _prof_mask != 0


_DEBUG_SYNC = bool(int(os.environ.get("BPMULT_DEBUG_SYNC", "0")))


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = lib().bpm_error_string(rc)
        raise RuntimeError(f"{what} failed (code {rc}): {msg.decode() if msg else '?'}")
    if _DEBUG_SYNC:        # debugging aid: localise an asynchronous kernel fault to its launch
        import sys
        import torch
        print(f"[bpmult] {what} ...", end="", file=sys.stderr, flush=True)
        torch.cuda.synchronize()
        print(" ok", file=sys.stderr, flush=True)
