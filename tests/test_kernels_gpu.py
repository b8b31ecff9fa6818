"""Per-kernel parity on the MI355X, through the C ABI (ctypes), against fp64/fp32
torch CPU restatements of the same op.  f32 mode must agree to ~1e-5 (exact f32
MFMA); bf16 mode is compared against the same maths on bf16-rounded operands.
"""
import math
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from bpmult_amd import ops  # noqa: E402
from bpmult_amd.ops import (BPM_BF16, BPM_F32, F_ACCUM, F_ATOMIC, F_RELU, GEMM_NN, GEMM_NT, GEMM_TN, OUT_CT, OUT_F32, OUT_HEADS,  # noqa: E402
                            pad32)

DEV = "cuda"
DT = [BPM_F32, BPM_BF16]


class lab_library:
    """The -DBPM_LAB build of the library (tile-configuration / block-pairing overrides; never the product): skip when
    `__graft_entry__.build()` has not produced it."""

    def __enter__(self):
        from bpmult_amd import _lib
        if not os.path.exists(_lib.LAB_LIB_PATH):
            pytest.skip("build/lab/libbpmult_hip_lab.so not built (python -c 'import __graft_entry__ as g; g.build()')")
        self.cm = _lib.lab_library()
        return self.cm.__enter__()

    def __exit__(self, *exc):
        return self.cm.__exit__(*exc)


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def to_ct(x, dtype, ld=None):
    """CPU fp32 [R,C] -> device CT [R,ld] zero padded; also the CT-rounded fp32 copy."""
    R, Cn = x.shape
    ld = pad32(Cn) if ld is None else ld
    buf = torch.zeros(R, ld, dtype=ops.ct_torch(dtype))
    buf[:, :Cn] = x.to(ops.ct_torch(dtype))
    return buf.to(DEV), buf[:, :Cn].float()


def tol(dtype):
    return 2e-5 if dtype == BPM_F32 else 1.5e-2


def close(got, ref, t, what=""):
    got, ref = got.detach().cpu().double(), ref.double()
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    assert torch.isfinite(got).all(), what + ": non-finite"
    scale = max(1.0, ref.abs().max().item())
    err = (got - ref).abs().max().item()
    assert err <= t * scale, f"{what}: max err {err:.3e} vs tol {t * scale:.3e}"


# numpy restatement of the dropout hash (bpm_common.h) -- test infrastructure
def host_key(seed, site):
    M = (1 << 64) - 1
    z = (seed + 0x9E3779B97F4A7C15 * (site + 1)) & M
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M
    z ^= z >> 31
    return (z ^ (z >> 32)) & 0xFFFFFFFF


def drop_mult(shape, p, seed, site):
    if p <= 0:
        return torch.ones(shape)
    n = int(np.prod(shape))
    idx = np.arange(n, dtype=np.uint64)
    M = 0xFFFFFFFF
    a = ((idx >> 2) + host_key(seed, site)) & M
    a ^= a >> 16
    a = (a * 0x85EBCA6B) & M
    a ^= a >> 13
    w0 = (a * 0xC2B2AE35) & M
    w0 ^= w0 >> 16
    w1 = (a * 0x27D4EB2F) & M
    w1 ^= w1 >> 15
    w = np.where((idx & 2).astype(bool), w1, w0)
    bits = np.where((idx & 1).astype(bool), w >> 16, w & 0xFFFF)
    thresh = int(p * 65536.0 + 0.5)
    keep = bits >= thresh
    return torch.from_numpy(np.where(keep, 1.0 / (1.0 - p), 0.0).astype(np.float32)).reshape(shape)


# ---------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("M,N,K", [(200, 300, 300), (520, 70, 35), (129, 65, 1200), (8, 6, 24)])
def test_gemm_nt_bias_resid(dtype, M, N, K):
    A, Ar = to_ct(rnd(M, K, seed=1), dtype)
    W, Wr = to_ct(rnd(N, K, seed=2, scale=K ** -0.5), dtype)
    bias = rnd(N, seed=3).to(DEV)
    resid = rnd(M, N, seed=4).to(DEV)
    out = torch.full((M, N), float("nan"), device=DEV)
    p = ops.gemm_problem(A, W, out, M, N, K, A.shape[1], W.shape[1], N, bias_n=bias, resid=resid, ldr=N)
    ops.gemm_grouped(dtype, GEMM_NT, [p])
    ref = Ar.double() @ Wr.double().T + bias.cpu().double() + resid.cpu().double()
    close(out, ref, tol(dtype) if dtype == BPM_F32 else 2e-3, "nt")


@pytest.mark.parametrize("variant", [GEMM_NT, GEMM_NN, GEMM_TN])
@pytest.mark.parametrize("M,N,K", [(1000, 768, 768), (520, 1536, 320), (4096, 3072, 768), (768, 3072, 1100), (300, 260, 512)])
def test_gemm_large_tile_paths_bf16(variant, M, N, K):
    """Shapes that the dispatcher sends to the LDS-DMA kernel (256 x 256 tiles; ragged M / N / K, K not a multiple of
    the 64-k stage with zero-padded rows) and neighbours that stay on the 128 x 64 kernel: bias + residual epilogue,
    bf16 operands, against fp64 on the bf16-rounded operands; in f32 mode the same shapes must agree too."""
    for dtype in (BPM_BF16, BPM_F32):
        pad64 = lambda n: (n + 63) // 64 * 64
        if variant == GEMM_NT:
            A, Ar = to_ct(rnd(M, K, seed=1), dtype, pad64(K)); Bm, Br = to_ct(rnd(N, K, seed=2, scale=K ** -0.5), dtype, pad64(K))
            ref = Ar.double() @ Br.double().T
        elif variant == GEMM_NN:
            A, Ar = to_ct(rnd(M, K, seed=1), dtype, pad64(K)); Bm, Br = to_ct(rnd(K, N, seed=2, scale=K ** -0.5), dtype)
            ref = Ar.double() @ Br.double()
        else:
            A, Ar = to_ct(rnd(K, M, seed=1), dtype); Bm, Br = to_ct(rnd(K, N, seed=2, scale=K ** -0.5), dtype)
            ref = Ar.double().T @ Br.double()
        bias = rnd(N, seed=3).to(DEV)
        resid = rnd(M, N, seed=4).to(DEV)
        out = torch.full((M, N), float("nan"), device=DEV)
        cs = torch.zeros(M, device=DEV)
        kw = dict(colsum_a=cs) if variant == GEMM_TN else {}
        p = ops.gemm_problem(A, Bm, out, M, N, K, A.shape[1], Bm.shape[1], N, bias_n=bias, resid=resid, ldr=N, flags=ops.F_KPAD, **kw)
        ops.gemm_grouped(dtype, variant, [p, p] if variant != GEMM_TN else [p])
        close(out, ref + bias.cpu().double() + resid.cpu().double(), tol(dtype) if dtype == BPM_F32 else 2e-3, f"large tile v{variant}")
        if variant == GEMM_TN:
            close(cs, Ar.double().sum(0), 1e-4 if dtype == BPM_F32 else 2e-3, "colsum_a")


@pytest.mark.parametrize("variant,M,N,K", [(GEMM_NT, 1000, 520, 328), (GEMM_NN, 1000, 520, 328), (GEMM_NT, 1300, 768, 768), (GEMM_NN, 960, 1024, 3072)])
def test_gemm_tall_tile_ragged(variant, M, N, K):
    """The 320 x 256 configuration (picked when it saves a round of workgroups: gemm.hip, tile choice) forced on ragged
    shapes -- rows past M in the last 320-row tile, a k tail, a column tail -- against fp64 on the bf16-rounded operands."""
    from bpmult_amd import _lib
    pad64 = lambda n: (n + 63) // 64 * 64
    A, Ar = to_ct(rnd(M, K, seed=21), BPM_BF16, pad64(K))
    if variant == GEMM_NT:
        Bm, Br = to_ct(rnd(N, K, seed=22, scale=K ** -0.5), BPM_BF16, pad64(K))
        ref = Ar.double() @ Br.double().T
    else:
        Bm, Br = to_ct(rnd(K, N, seed=22, scale=K ** -0.5), BPM_BF16)
        ref = Ar.double() @ Br.double()
    bias, resid = rnd(N, seed=23).to(DEV), rnd(M, N, seed=24).to(DEV)
    out = torch.full((M, N), float("nan"), device=DEV)
    p = ops.gemm_problem(A, Bm, out, M, N, K, A.shape[1], Bm.shape[1], N, bias_n=bias, resid=resid, ldr=N, flags=ops.F_KPAD)
    with lab_library() as L:                   # -DBPM_LAB build: the product library has no configuration override
        _lib.check(L.bpm_debug_gemm_force(5), "force")
        ops.gemm_grouped(BPM_BF16, variant, [p, p, p])
        torch.cuda.synchronize()
    close(out, ref + bias.cpu().double() + resid.cpu().double(), 2e-3, f"tall tile v{variant}")


def test_gemm_large_tile_epilogues_match_small_tile():
    """Every fused epilogue through the LDS-DMA kernel (wide LDS-transposed stores) against the 128 x 64 kernel on the
    same operands: relu + dropout -> CT, gate + column sums -> CT, head-major scatter (head_dim 128 and 64), += into f32.
    The arithmetic per element is the same, so bf16 outputs may differ by one rounding of a differently ordered sum."""
    from bpmult_amd import _lib
    M, N, K = 1024, 768, 768
    B_, ctt = 8, torch.bfloat16
    A, _ = to_ct(rnd(M, K, seed=11), BPM_BF16, K)
    W, _ = to_ct(rnd(N, K, seed=12, scale=K ** -0.5), BPM_BF16, K)
    gate, _ = to_ct(rnd(M, N, seed=14), BPM_BF16, N)
    bias = rnd(N, seed=13).to(DEV)
    base = rnd(M, N, seed=15).to(DEV)
    res = {}
    with lab_library() as L:
        for cfg in (-2, 3, 2, 0, 5, 6):
            _lib.check(L.bpm_debug_gemm_force(cfg), "force")
            o1 = torch.full((M, N), float("nan"), device=DEV).to(ctt)
            o2 = torch.full((M, N), float("nan"), device=DEV).to(ctt)
            o3 = torch.zeros(B_, 6, M // B_, 128, device=DEV, dtype=ctt)
            o4 = torch.zeros(B_, 12, M // B_, 64, device=DEV, dtype=ctt)
            o5 = base.clone()
            cs = torch.zeros(N, device=DEV)
            ps = [ops.gemm_problem(A, W, o1, M, N, K, K, K, N, bias_n=bias, flags=F_RELU | ops.F_KPAD, drop_p=0.3, drop_site=5, out_kind=OUT_CT),
                  ops.gemm_problem(A, W, o2, M, N, K, K, K, N, gate=gate, ldg=N, gate_scale=1.25, colsum=cs, flags=ops.F_KPAD, out_kind=OUT_CT),
                  ops.gemm_problem(A, W, o3, M, N, K, K, K, 0, bias_n=bias, alpha=0.2, out_kind=OUT_HEADS, heads=(B_, 6, M // B_, 128, 128), flags=ops.F_KPAD),
                  ops.gemm_problem(A, W, o4, M, N, K, K, K, 0, bias_n=bias, out_kind=OUT_HEADS, heads=(B_, 12, M // B_, 64, 64), flags=ops.F_KPAD),
                  ops.gemm_problem(A, W, o5, M, N, K, K, K, N, flags=F_ACCUM | ops.F_KPAD)]
            ops.gemm_grouped(BPM_BF16, GEMM_NT, ps, seed=77)
            torch.cuda.synchronize()
            res[cfg] = [t.float().cpu() for t in (o1, o2, o3, o4, o5, cs)]
    for cfg in (3, 2, 0, 5, 6):
        for i, nm in enumerate(("relu+drop CT", "gate CT", "heads 128", "heads 64", "accum f32", "colsum")):
            close(res[cfg][i], res[-2][i], 1e-2 if i < 4 else 2e-4, f"cfg {cfg} {nm}")
        assert ((res[cfg][0] == 0) == (res[-2][0] == 0)).all(), "dropout / relu zero pattern must be identical"


@pytest.mark.parametrize("T", [1001, 640])
def test_audio_encoder_frontend_lengths_that_leave_unused_frames(T):
    """Sequence lengths at which the last frame of a layer's input is read by no window ((L - 128) % 2 == 1 in layer 1 at
    T = 1001, in layer 2 at both): the implicit product trims those positions, their input gradient is exactly zero, and
    everything else equals fp64 torch.  Batch 3, so virtual rows straddle two batch boundaries."""
    from detgen import det, det_param
    from bpmult_amd.models.bpmult import AudioEncoder
    enc = AudioEncoder()
    with torch.no_grad():
        for k, p in enc.named_parameters():
            p.copy_(torch.from_numpy(det_param("f8.audio_enc." + k, p.shape)))
    aud = torch.from_numpy(det("t.aud%d" % T, (3, 96, T)))
    ref_leaves = [p.detach().double().requires_grad_(True) for p in enc.parameters()]
    w1, b1, w2, b2 = ref_leaves
    xr = aud.double().requires_grad_(True)
    yr = torch.nn.functional.adaptive_avg_pool1d(torch.nn.functional.conv1d(torch.nn.functional.conv1d(xr, w1, b1, stride=2), w2, b2, stride=2), 50)
    wgt = torch.from_numpy(det("t.audw%d" % T, (3, 96, 50))).double()
    (yr * wgt).sum().backward()
    enc.conv_layers[2] = torch.nn.AdaptiveAvgPool1d(50)
    enc = enc.cuda()
    x = aud.cuda().requires_grad_(True)
    y = enc.encode(x, "f32")
    (y * wgt.float().cuda().transpose(1, 2)).sum().backward()
    close(y.transpose(1, 2), yr.detach(), 2e-5, "audio_feat")
    close(x.grad, xr.grad, 2e-4, "d(audio)")
    if T % 2 == 1:
        assert float(x.grad[..., -1].abs().max()) == 0.0, "the frame no window reads must get a zero gradient"
    for (k, p), r in zip(enc.named_parameters(), ref_leaves):
        close(p.grad, r.grad, 2e-4, "d(" + k + ")")


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("N", [96, 512])
def test_gemm_operands_with_overlapping_rows(dtype, N):
    """BPM_GEMM_A_OVERLAP / _B_OVERLAP: an operand whose leading dimension is smaller than its row length -- the sliding
    windows of a strided convolution over a channels-last signal (frontend.py).  NT with the overlapping A (N = 96: the
    128 x 64 kernel; N = 512: the LDS-DMA kernel in bf16) and TN with the same buffer as the overlapping B, against fp64 on
    the explicitly unfolded windows; the plain-pointer loaders refuse the flags."""
    from bpmult_amd._lib import F_A_OVERLAP, F_B_OVERLAP
    ct = torch.bfloat16 if dtype == BPM_BF16 else torch.float32
    C, taps, stride, L = 32, 16, 2, 1030
    K, ld = taps * C, stride * C                       # 512-element windows every 64 elements
    M = (L - taps) // stride + 1
    sig = rnd(L * C, seed=31).to(ct).to(DEV)
    win = torch.stack([sig[r * ld:r * ld + K] for r in range(M)]).double().cpu()          # [M, K]
    W = rnd(N, K, seed=32, scale=K ** -0.5).to(ct).to(DEV)
    bias = rnd(N, seed=33).to(DEV)
    out = torch.full((M, N), float("nan"), device=DEV)
    ops.gemm_grouped(dtype, GEMM_NT, [ops.gemm_problem(sig, W, out, M, N, K, ld, K, N, bias_n=bias, flags=ops.F_KPAD | F_A_OVERLAP)])
    torch.cuda.synchronize()
    close(out, win @ W.double().cpu().T + bias.double().cpu(), 1e-4 if dtype == BPM_F32 else 2e-3, "NT, overlapping A")
    # TN: dW[n, (k,c)] = sum_r dy[r, n] window_r -- k extent padded to 64 rows of zeros in dy; the signal readable behind them
    Mp = (M + 63) // 64 * 64
    sig2 = torch.zeros((Mp - 1) * ld + K, device=DEV, dtype=ct)
    sig2[:L * C] = sig
    dy = torch.zeros(Mp, N, device=DEV, dtype=ct)
    dy[:M] = rnd(M, N, seed=34).to(ct).to(DEV)
    dW = torch.full((N, K), float("nan"), device=DEV)
    cs = torch.zeros(N, device=DEV)
    ops.gemm_grouped(dtype, GEMM_TN, [ops.gemm_problem(dy, sig2, dW, N, K, Mp, N, ld, K, flags=ops.F_KPAD | F_B_OVERLAP, colsum_a=cs)])
    torch.cuda.synchronize()
    close(dW, dy[:M].double().cpu().T @ win, 1e-4 if dtype == BPM_F32 else 3e-3, "TN, overlapping B")
    close(cs, dy[:M].double().cpu().sum(0), 1e-4 if dtype == BPM_F32 else 2e-3, "TN column sums")
    with pytest.raises(RuntimeError):                  # without the zero-padding promise there is no hardware-bounded loader
        ops.gemm_grouped(dtype, GEMM_NT, [ops.gemm_problem(sig, W, out, M, N, K, ld, K, N, flags=F_A_OVERLAP)])


@pytest.mark.parametrize("dtype", DT)
def test_gemm_nn_tn_and_splitk(dtype):
    M, N, K = 300, 140, 200          # dgrad: dx[M,K'] = dy[M,N'] W[N',K']; here generic names
    A, Ar = to_ct(rnd(M, K, seed=5), dtype)
    Bm, Br = to_ct(rnd(K, N, seed=6, scale=K ** -0.5), dtype)
    out = torch.full((M, N), float("nan"), device=DEV)
    ops.gemm_grouped(dtype, GEMM_NN, [ops.gemm_problem(A, Bm, out, M, N, K, A.shape[1], Bm.shape[1], N)])
    close(out, Ar.double() @ Br.double(), tol(dtype) if dtype == BPM_F32 else 2e-3, "nn")
    # TN: C[M,N] = A[K,M]^T B[K,N], contraction over rows, split-K with atomics into zeros
    Kc = 1000
    At, Atr = to_ct(rnd(Kc, M, seed=7), dtype)
    Bt, Btr = to_ct(rnd(Kc, N, seed=8, scale=Kc ** -0.5), dtype)
    for splitk in (1, 4):
        out = torch.zeros(M, N, device=DEV)
        ops.gemm_grouped(dtype, GEMM_TN, [ops.gemm_problem(At, Bt, out, M, N, Kc, At.shape[1], Bt.shape[1], N,
                                                           flags=F_ATOMIC, splitk=splitk)])
        close(out, Atr.double().T @ Btr.double(), tol(dtype) if dtype == BPM_F32 else 2e-3, f"tn splitk={splitk}")


@pytest.mark.parametrize("dtype", DT)
def test_gemm_batched_problems_with_interleaved_rows(dtype):
    """BPM_GEMM_BATCHED + BPM_GEMM_CT_NARROW: the products of the engine's low-rank key side -- per batch element
    [H T, S] x [S, d] with the rows of a batch element B rows apart in both operands and the output (NN, CT output that must
    not touch its neighbours' rows), and the transposed product back (TN, fp32) -- beside a plain problem in the same launch."""
    from bpmult_amd._lib import F_CT_NARROW
    F_KPAD = ops.F_KPAD
    B_, HT, S, d = 3, 10, 72, 96
    Sp, ld = 128, pad32(d)
    ctt = ops.ct_torch(dtype)
    dS = torch.zeros(HT, B_, Sp, dtype=ctt)
    dS[..., :S] = rnd(HT, B_, S, seed=41).to(ctt)
    khat = torch.zeros(S, B_, ld, dtype=ctt)
    khat[..., :d] = rnd(S, B_, d, seed=42, scale=S ** -0.5).to(ctt)
    U = torch.full((HT, B_, ld), 9.0, dtype=ctt).to(DEV)
    dSd, khd = dS.to(DEV), khat.to(DEV)
    A2, A2r = to_ct(rnd(40, 64, seed=43), dtype)
    W2, W2r = to_ct(rnd(64, 52, seed=44), dtype)
    o2 = torch.zeros(40, 52, device=DEV)
    ops.gemm_grouped(dtype, GEMM_NN, [
        ops.gemm_problem(dSd, khd, U, HT, d, S, B_ * Sp, B_ * ld, B_ * ld, out_kind=OUT_CT, flags=F_CT_NARROW | F_KPAD,
                         batch=(B_, Sp, ld, ld)),
        ops.gemm_problem(A2, W2, o2, 40, 52, 64, A2.shape[1], W2.shape[1], 52)])
    torch.cuda.synchronize()
    ref = torch.einsum("kbs,sbn->kbn", dS[..., :S].double(), khat[..., :d].double())
    t = tol(dtype) if dtype == BPM_F32 else 1e-2
    close(U[..., :d].float(), ref, t, "batched NN")
    assert (U[..., d:].float() == 9.0).all(), "CT_NARROW: pad columns stay untouched"
    close(o2, A2r.double() @ W2r.double(), t, "plain problem beside it")
    # back: G[(s, b), :] = sum_k dS[k, b, s] * U[k, b, :]
    G = torch.full((S, B_, d), float("nan"), device=DEV)
    Uc = U.clone()
    Uc[..., d:] = 0
    ops.gemm_grouped(dtype, GEMM_TN, [ops.gemm_problem(dSd, Uc, G, S, d, HT, B_ * Sp, B_ * ld, B_ * d, flags=F_KPAD,
                                                       batch=(B_, Sp, ld, d))])
    torch.cuda.synchronize()
    refG = torch.einsum("kbs,kbn->sbn", dS[..., :S].double(), Uc[..., :d].float().cpu().double())
    close(G, refG, t, "batched TN")


@pytest.mark.parametrize("dtype", DT)
def test_gemm_epilogues_grouped(dtype):
    """relu+dropout -> CT (fc1), gate (fc2 dgrad), heads scatter (Q proj), three problems in one launch."""
    M, N, K = 70, 100, 64
    B_, H, dh, dhp = 5, 4, 25, 32
    T_ = M // B_
    ctt = ops.ct_torch(dtype)
    A, Ar = to_ct(rnd(M, K, seed=11), dtype)
    W, Wr = to_ct(rnd(N, K, seed=12, scale=K ** -0.5), dtype)
    bias = rnd(N, seed=13).to(DEV)
    gate, gater = to_ct(rnd(M, N, seed=14), dtype)
    o1 = torch.full((M, pad32(N)), float("nan"), device=DEV).to(ctt)
    o2 = torch.full((M, N), float("nan"), device=DEV)
    o3 = torch.zeros(B_, H, T_, dhp, device=DEV, dtype=ctt)
    cs1 = torch.ones(N, device=DEV)
    p1 = ops.gemm_problem(A, W, o1, M, N, K, A.shape[1], W.shape[1], pad32(N), bias_n=bias, flags=F_RELU,
                          drop_p=0.3, drop_site=5, out_kind=OUT_CT, colsum=cs1)
    p2 = ops.gemm_problem(A, W, o2, M, N, K, A.shape[1], W.shape[1], N, gate=gate, ldg=gate.shape[1], gate_scale=1.25)
    p3 = ops.gemm_problem(A, W, o3, M, N, K, A.shape[1], W.shape[1], 0, bias_n=bias, alpha=0.2, out_kind=OUT_HEADS,
                          heads=(B_, H, T_, dh, dhp))
    ops.gemm_grouped(dtype, GEMM_NT, [p1, p2, p3], seed=77)
    acc = Ar.double() @ Wr.double().T
    t = tol(dtype) if dtype == BPM_F32 else 1e-2
    ref1 = torch.relu(acc + bias.cpu().double()) * drop_mult((M, N), 0.3, 77, 5).double()
    close(o1[:, :N].float(), ref1, t, "relu+dropout CT")
    close(cs1, 1.0 + ref1.sum(0), t, "gemm colsum")
    assert (o1[:, N:].float() == 0).all(), "pad columns must be zero"
    frac = (ref1 == 0).double().mean().item()
    assert 0.3 < frac < 0.9
    close(o2, torch.where(gater.double() > 0, acc * 1.25, torch.zeros_like(acc)), t, "gate")
    ref3 = ((acc + bias.cpu().double()) * 0.2).reshape(T_, B_, H, dh).permute(1, 2, 0, 3)
    close(o3[..., :dh].float(), ref3, t, "heads")
    assert (o3[..., dh:].float() == 0).all()


# ---------------------------------------------------------------------------
def attn_ref(q, k, v, off, pmask):
    """q [B,H,T,dh] (already scaled), k,v [B,H,S,dh]; pmask [B,H,T,S] dropout multipliers."""
    T_, S_ = q.shape[2], k.shape[2]
    s = q @ k.transpose(-1, -2)
    if off > 0:
        i = torch.arange(T_)[:, None]
        j = torch.arange(S_)[None, :]
        s = s.masked_fill((j - i) >= off, float("-inf"))
    p = torch.softmax(s, -1)
    return (p * pmask) @ v, torch.logsumexp(s, -1)


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("B,H,T,S,dh,masked,pdrop", [(2, 3, 70, 100, 25, True, 0.0), (1, 2, 130, 130, 25, True, 0.2),
                                                      (2, 2, 100, 70, 6, True, 0.0), (1, 2, 50, 50, 128, True, 0.0),
                                                      (1, 1, 33, 65, 64, False, 0.1), (1, 2, 2, 200, 25, True, 0.0),
                                                      (1, 1, 1, 1, 25, True, 0.0), (3, 1, 1, 5, 8, True, 0.0),
                                                      (1, 2, 513, 512, 25, True, 0.0),
                                                      # head_dim 128 (hidden 768 / 6 heads, BASELINE configs[2]): multi-tile
                                                      # forward / dQ / dK-dV paths, T > S (no mask), T < S (band), T = S + 1
                                                      (1, 2, 512, 200, 128, True, 0.0), (1, 2, 200, 512, 128, True, 0.0),
                                                      (1, 1, 513, 512, 128, True, 0.0), (1, 2, 512, 200, 128, True, 0.1),
                                                      (1, 2, 200, 512, 128, True, 0.1), (1, 2, 512, 512, 64, True, 0.1)])
def test_attention_fwd_bwd(dtype, B, H, T, S, dh, masked, pdrop):
    dhp = 32 if dh <= 32 else (64 if dh <= 64 else 128)
    ctt = ops.ct_torch(dtype)
    d = H * dh
    ld = pad32(d)

    def heads(x):        # [B,H,L,dh] cpu -> device CT [B,H,L,dhp], and the CT-rounded values
        buf = torch.zeros(*x.shape[:3], dhp, dtype=ctt)
        buf[..., :dh] = x.to(ctt)
        return buf.to(DEV), buf[..., :dh].double()

    Q, q = heads(rnd(B, H, T, dh, seed=21) * dh ** -0.5)
    K, k = heads(rnd(B, H, S, dh, seed=22))
    V, v = heads(rnd(B, H, S, dh, seed=23))
    dO, do = heads(rnd(B, H, T, dh, seed=24))
    q.requires_grad_(True); k.requires_grad_(True); v.requires_grad_(True)
    off = 1 + abs(S - T) if masked else 0
    pm = drop_mult((B, H, T, S), pdrop, 9, 3).double()
    o_ref, lse_ref = attn_ref(q, k, v, off, pm)
    (o_ref * do).sum().backward()

    O = torch.zeros(T * B, ld, device=DEV, dtype=ctt)
    lse = torch.zeros(B, H, T, device=DEV)
    delta = torch.zeros(B, H, T, device=DEV)
    dQ, dK, dV = (torch.zeros(L * B, ld, device=DEV, dtype=ctt) for L in (T, S, S))
    p = ops.attn_problem(Q, K, V, O, ld, lse, B, H, T, S, dh, dhp, off, dO=dO, delta=delta, dQ=dQ, lddq=ld, dK=dK, lddk=ld,
                         dV=dV, lddv=ld, dq_scale=1.0, drop_p=pdrop, drop_site=3)
    ops.attn_fwd(dtype, [p], seed=9)
    t = 3e-5 if dtype == BPM_F32 else 2e-2

    def rows(x, L):      # row-major [(l*B+b), h*dh+c] -> [B,H,L,dh]
        return x[:, :d].float().reshape(L, B, H, dh).permute(1, 2, 0, 3)

    close(rows(O, T), o_ref.detach(), t, "O")
    close(lse, lse_ref.detach(), t, "lse")
    assert (O[:, d:].float() == 0).all()
    ops.attn_bwd(dtype, [p], seed=9)
    tb = 1e-4 if dtype == BPM_F32 else 4e-2
    close(rows(dQ, T), q.grad, tb, "dQ")
    close(rows(dK, S), k.grad, tb, "dK")
    close(rows(dV, S), v.grad, tb, "dV")



@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("B,H,T,S,dh,pdrop,qpos", [(2, 3, 2, 200, 25, 0.0, (0, 199)), (3, 2, 2, 72, 64, 0.2, (0, 71)),
                                                    (2, 2, 4, 132, 128, 0.1, (0, 1)), (1, 2, 2, 512, 64, 0.1, (0, 511))])
def test_attention_dq_pass_exports_score_gradient_and_dropped_probabilities(dtype, B, H, T, S, dh, pdrop, qpos):
    """bpm_attn_bwd_dq with dS / Pd: the [rows, S] factors of dK = dS^T Q and dV = Pd^T dO (the engine's low-rank key side),
    in the [h*T + t][b][padded keys] layout the engine uses, against fp64 torch with the library's dropout masks; then
    bpm_expand_heads on the same operands against torch."""
    dhp = 32 if dh <= 32 else (64 if dh <= 64 else 128)
    ctt = ops.ct_torch(dtype)
    d, ld, Sp = H * dh, pad32(H * dh), (S + 63) // 64 * 64

    def heads(x):
        buf = torch.zeros(*x.shape[:3], dhp, dtype=ctt)
        buf[..., :dh] = x.to(ctt)
        return buf.to(DEV), buf[..., :dh].double()

    Q, q = heads(rnd(B, H, T, dh, seed=31) * dh ** -0.5)
    K, k = heads(rnd(B, H, S, dh, seed=32))
    V, v = heads(rnd(B, H, S, dh, seed=33))
    dO, do = heads(rnd(B, H, T, dh, seed=34))
    pos0, stride = qpos[0], (qpos[1] - qpos[0]) if T == 2 else 1
    off = 1                                              # T_full = S: key j visible to the query at time step i iff j <= i
    pm = drop_mult((B, H, T, S), pdrop, 9, 3).double()
    sc = q @ k.transpose(-1, -2)
    tpos = pos0 + stride * torch.arange(T)
    sc = sc.masked_fill((torch.arange(S)[None, :] - tpos[:, None]) >= off, float("-inf"))
    pr = torch.softmax(sc, -1)
    pd_ref = pr * pm
    o = pd_ref @ v
    dp = do @ v.transpose(-1, -2)
    ds_ref = pr * (pm * dp - (do * o).sum(-1, keepdim=True))

    O = torch.zeros(T * B, ld, device=DEV, dtype=ctt)
    lse, delta = torch.zeros(B, H, T, device=DEV), torch.zeros(B, H, T, device=DEV)
    dQ = torch.zeros(T * B, ld, device=DEV, dtype=ctt)
    dS = torch.full((H * T, B, Sp), 7.0, device=DEV, dtype=ctt)
    Pd = torch.full((H * T, B, Sp), 7.0, device=DEV, dtype=ctt)
    p = ops.attn_problem(Q, K, V, O, ld, lse, B, H, T, S, dh, dhp, off, dO=dO, delta=delta, dQ=dQ, lddq=ld, dq_scale=1.0,
                         drop_p=pdrop, drop_site=3, q_pos0=pos0, q_stride=stride, dS=dS, Pd=Pd, xs=(Sp, T * B * Sp, B * Sp))
    ops.attn_fwd(dtype, [p], seed=9)
    ops.attn_bwd_dq(dtype, [p], seed=9)
    torch.cuda.synchronize()
    t = 1e-4 if dtype == BPM_F32 else 3e-2
    got_ds = dS.float().reshape(H, T, B, Sp).permute(2, 0, 1, 3)
    got_pd = Pd.float().reshape(H, T, B, Sp).permute(2, 0, 1, 3)
    # keys beyond the last visible one of the block's last query are never visited: they keep what the buffer held
    vis = int(min(S, tpos.max().item() + off))
    nt = (vis + 63) // 64 * 64
    close(got_ds[..., :min(nt, S)], ds_ref[..., :min(nt, S)], t, "dS")
    close(got_pd[..., :min(nt, S)], pd_ref[..., :min(nt, S)], t, "Pd")
    assert (got_ds[..., S:] == 7.0).all() and (got_pd[..., S:] == 7.0).all(), "key padding must not be written"
    # ---- expand_heads
    qexp = torch.full((H * T * B, ld), 5.0, device=DEV, dtype=ctt)
    doexp = torch.full((H * T * B, ld), 5.0, device=DEV, dtype=ctt)
    dbias = torch.full((d,), 3.0, device=DEV)
    Pz = Pd.clone()
    Pz[..., S:] = 0                                      # the engine's buffers are zero there
    ops.expand_heads(dtype, [ops.expand_problem(Q, dO, qexp, doexp, B, H, T, dh, dhp, ld, Pd=Pz, S=Sp, dbias=dbias)])
    torch.cuda.synchronize()
    ref_q = torch.zeros(H, T, B, ld, dtype=torch.float64)
    ref_do = torch.zeros(H, T, B, ld, dtype=torch.float64)
    for h in range(H):
        ref_q[h, :, :, h * dh:(h + 1) * dh] = q[:, h].permute(1, 0, 2)
        ref_do[h, :, :, h * dh:(h + 1) * dh] = do[:, h].permute(1, 0, 2)
    assert torch.equal(qexp.float().cpu().double().reshape(H, T, B, ld), ref_q)
    assert torch.equal(doexp.float().cpu().double().reshape(H, T, B, ld), ref_do)
    rs = Pz.float().cpu().double().reshape(H, T, B, Sp).sum(-1)                    # [H, T, B]
    ref_b = torch.einsum("htb,bhtj->hj", rs, do).reshape(d)
    close(dbias, ref_b, 1e-5 if dtype == BPM_F32 else 1e-3, "value-bias gradient")


@pytest.mark.parametrize("T,S,dh", [(320, 320, 64), (65, 200, 25), (513, 512, 64), (64, 64, 128)])
def test_attention_block_pairing_is_bitwise_neutral(T, S, dh):
    """A workgroup of the attention kernels takes two 64-row blocks (b, nblk-1-b); the tuning hook switches every kernel
    back to one block per workgroup.  Both schedules run the same per-block code, so O, LSE, dQ, dK, dV must be
    bit-identical -- odd block counts (the middle block is alone), one block, masked and with dropout."""
    from bpmult_amd import _lib
    B, H = 2, 2
    dhp = 32 if dh <= 32 else (64 if dh <= 64 else 128)
    ctt, d = torch.bfloat16, H * dh
    ld = pad32(d)
    mk = lambda L_, seed: torch.cat([rnd(B, H, L_, dh, seed=seed) * 0.5, torch.zeros(B, H, L_, dhp - dh)], -1).to(ctt).to(DEV)
    Q, K, V, dO = mk(T, 41), mk(S, 42), mk(S, 43), mk(T, 44)
    res = {}
    with lab_library() as L:
        for mode in (7, 0):
            _lib.check(L.bpm_debug_attn_pair(mode), "bpm_debug_attn_pair")
            O = torch.zeros(T * B, ld, device=DEV, dtype=ctt)
            lse, delta = torch.zeros(B, H, T, device=DEV), torch.zeros(B, H, T, device=DEV)
            dQ, dK, dV = (torch.zeros(n * B, ld, device=DEV, dtype=ctt) for n in (T, S, S))
            p = ops.attn_problem(Q, K, V, O, ld, lse, B, H, T, S, dh, dhp, 1 + abs(S - T), dO=dO, delta=delta, dQ=dQ, lddq=ld,
                                 dK=dK, lddk=ld, dV=dV, lddv=ld, dq_scale=dh ** -0.5, drop_p=0.1, drop_site=4)
            ops.attn_fwd(BPM_BF16, [p], seed=11)
            ops.attn_bwd(BPM_BF16, [p], seed=11)
            torch.cuda.synchronize()
            res[mode] = [t.float().cpu() for t in (O, lse, dQ, dK, dV)]
    for a, b, nm in zip(res[7], res[0], ("O", "lse", "dQ", "dK", "dV")):
        assert torch.isfinite(a).all(), nm
        assert torch.equal(a, b), f"{nm}: paired and unpaired schedules differ"

# ---------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", DT)
def test_pack_rows_and_embed_pos(dtype):
    B, T, Cn = 3, 7, 35
    ld = pad32(Cn)
    src = rnd(B, T, Cn, seed=31)
    src2 = rnd(B, 4, 74, seed=30)
    dst = torch.full((T * B, ld), float("nan"), device=DEV).to(ops.ct_torch(dtype))
    dst2 = torch.full((4 * B, 96), float("nan"), device=DEV).to(ops.ct_torch(dtype))
    srcd, src2d = src.to(DEV), src2.to(DEV)      # problem structs hold raw pointers: keep the tensors alive
    ops.pack_rows_fwd(dtype, [ops.pack_problem(B, T, Cn, ld, src=srcd, dst=dst, drop_p=0.25, drop_site=2),
                              ops.pack_problem(B, 4, 74, 96, src=src2d, dst=dst2)], seed=5)
    ref = (src * drop_mult((B, T, Cn), 0.25, 5, 2)).permute(1, 0, 2).reshape(T * B, Cn)
    close(dst[:, :Cn].float(), ref.to(ops.ct_torch(dtype)).float(), 1e-6, "pack fwd")
    assert (dst[:, Cn:].float() == 0).all()
    close(dst2[:, :74].float(), src2.permute(1, 0, 2).reshape(4 * B, 74).to(ops.ct_torch(dtype)).float(), 1e-6, "pack fwd 2")
    g = rnd(T * B, ld, seed=32).to(DEV)
    dsrc = torch.full((B, T, Cn), float("nan"), device=DEV)
    ops.pack_rows_bwd([ops.pack_problem(B, T, Cn, 0, g=g, ldg=ld, dsrc=dsrc, drop_p=0.25, drop_site=2)], seed=5)
    refb = g.cpu()[:, :Cn].reshape(T, B, Cn).permute(1, 0, 2) * drop_mult((B, T, Cn), 0.25, 5, 2)
    close(dsrc, refb, 1e-6, "pack bwd")

    from oracle import bpmult_cpu as O
    T2, B2, d = 9, 3, 24
    x = rnd(T2, B2, d, seed=33)
    x[2, 1, 0] = 0.0
    x[-2:] = 0.0
    x3 = rnd(5, B2, d, seed=35)
    table = O.sinusoid_table(T2 + 1, d).to(DEV)
    out = torch.empty(T2, B2, d, device=DEV)
    out3 = torch.empty(5, B2, d, device=DEV)
    xd, x3d = x.to(DEV), x3.to(DEV)
    ops.embed_pos_fwd([ops.embed_problem(xd, out, T2, B2, drop_p=0.25, drop_site=4),
                       ops.embed_problem(x3d, out3, 5, B2)], table, d, math.sqrt(d), seed=1)
    ref = (math.sqrt(d) * x + O.pos_embedding(x)) * drop_mult((T2, B2, d), 0.25, 1, 4)
    close(out, ref, 1e-6, "embed_pos fwd")
    close(out3, math.sqrt(d) * x3 + O.pos_embedding(x3), 1e-6, "embed_pos fwd 2")
    dy = rnd(T2, B2, d, seed=34)
    dx = torch.ones(T2, B2, d, device=DEV)
    dyd = dy.to(DEV)
    ops.embed_pos_bwd([ops.embed_problem(dyd, dx, T2, B2, accumulate=True, drop_p=0.25, drop_site=4)], d, math.sqrt(d), seed=1)
    close(dx, 1.0 + math.sqrt(d) * dy * drop_mult((T2, B2, d), 0.25, 1, 4), 1e-6, "embed_pos bwd")


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("R,d", [(37, 300), (10, 24), (9, 50), (5, 768), (3, 1536)])
def test_layernorm(dtype, R, d):
    x = (rnd(R, d, seed=41) * 2 + 0.5).requires_grad_(True)
    x.data[1] = 0.0                                  # an all-zero (padded) row: LN(0) = beta
    gamma = (1 + 0.1 * rnd(d, seed=42)).requires_grad_(True)
    beta = (0.1 * rnd(d, seed=43)).requires_grad_(True)
    y = torch.nn.functional.layer_norm(x, (d,), gamma, beta, 1e-5)
    dy = rnd(R, d, seed=44)
    (y * dy).sum().backward()
    ld = pad32(d)
    out = torch.full((R, ld), float("nan"), device=DEV).to(ops.ct_torch(dtype))
    outf = torch.empty(R, d, device=DEV)
    mean, rstd, mean2, rstd2 = (torch.empty(R, device=DEV) for _ in range(4))
    xd, gd, bd = x.detach().to(DEV), gamma.detach().to(DEV), beta.detach().to(DEV)
    ops.ln_fwd(dtype, [ops.ln_problem(xd, gd, bd, mean, rstd, R, out=out, ldo=ld),
                       ops.ln_problem(xd, gd, bd, mean2, rstd2, R, out=outf, ldo=d, out_f32=True)], d)
    close(out[:, :d].float(), y.detach(), 2e-5 if dtype == BPM_F32 else 1e-2, "ln fwd")
    assert (out[:, d:].float() == 0).all()
    close(outf, y.detach(), 2e-5, "ln fwd f32")
    add = rnd(R, d, seed=45).to(DEV)
    dx, dx2 = torch.empty(R, d, device=DEV), torch.empty(R, d, device=DEV)
    dgam, dbet = torch.zeros(d, device=DEV), torch.zeros(d, device=DEV)
    dyd = dy.to(DEV)
    ops.ln_bwd([ops.ln_problem(xd, gd, None, mean, rstd, R, dy=dyd, ldy=d, add=add, dx=dx, dgamma=dgam, dbeta=dbet),
                ops.ln_problem(xd, gd, None, mean, rstd, R, dy=dyd, ldy=d, dx=dx2)], d)
    close(dx, x.grad + add.cpu(), 1e-4, "ln dx")
    close(dx2, x.grad, 1e-4, "ln dx (no add, no param grads)")
    close(dgam, gamma.grad, 1e-4, "ln dgamma")
    close(dbet, beta.grad, 1e-4, "ln dbeta")
    # fused hand-off: CT copy of dropmask(dx) with zero pad + its column sums (== bpm_rows_cast on dx)
    dx3 = torch.empty(R, d, device=DEV)
    cast = torch.full((R, ld), float("nan"), device=DEV).to(ops.ct_torch(dtype))
    cs = torch.ones(d, device=DEV)
    ops.ln_bwd([ops.ln_problem(xd, gd, None, mean, rstd, R, dy=dyd, ldy=d, add=add, dx=dx3, cast=cast, ldc=ld, cast_colsum=cs,
                               drop_p=0.2, drop_site=9)], d, dtype, 1234)
    assert torch.equal(dx3, dx)
    want = dx.cpu() * drop_mult((R, d), 0.2, 1234, 9)
    close(cast[:, :d].float(), want, 1e-6 if dtype == BPM_F32 else 1e-2, "ln fused cast")
    assert (cast[:, d:].float() == 0).all()
    close(cs, 1 + want.sum(0), 1e-4, "ln fused colsum")


@pytest.mark.parametrize("dtype", DT)
def test_rows_cast_and_gmu(dtype):
    R, Cn = 77, 300
    ld = pad32(Cn)
    ctt = ops.ct_torch(dtype)
    a, b = rnd(R, Cn, seed=51), rnd(R, Cn, seed=52)
    dct = torch.full((R, ld), float("nan"), device=DEV).to(ctt)
    df = torch.empty(R, Cn, device=DEV)
    cs = torch.ones(Cn, device=DEV)
    act, actr = to_ct(rnd(40, 70, seed=53), dtype)
    cs2 = torch.zeros(70, device=DEV)
    ad, bd_ = a.to(DEV), b.to(DEV)
    ops.rows_cast(dtype, [ops.cast_problem(ad, Cn, R, Cn, b=bd_, ldb=Cn, dst_ct=dct, ldd=ld, dst_f32=df, ldf=Cn,
                                           colsum=cs, drop_p=0.1, drop_site=9),
                          ops.cast_problem(act, act.shape[1], 40, 70, a_is_ct=True, colsum=cs2)], seed=3)
    ref = (a + b) * drop_mult((R, Cn), 0.1, 3, 9)
    close(df, ref, 1e-6, "cast f32")
    close(dct[:, :Cn].float(), ref.to(ctt).float(), 1e-6, "cast ct")
    assert (dct[:, Cn:].float() == 0).all()
    close(cs, 1.0 + ref.sum(0), 1e-4, "colsum")
    close(cs2, actr.sum(0), 1e-4, "colsum of a CT input")

    d = 24
    ts = [rnd(R, d, seed=60 + i).requires_grad_(True) for i in range(5)]
    a1, a2, ag, x1, x2 = ts
    z = torch.sigmoid(ag)
    y = z * torch.tanh(a1) * x1 + (1 - z) * torch.tanh(a2) * x2
    dout = rnd(R, d, seed=66)
    (y * dout).sum().backward()
    dv = [t.detach().to(DEV) for t in ts]
    out = torch.empty(R, d, device=DEV)
    ops.gmu2_fwd([ops.gmu_problem(*dv, R, out=out)], d)
    close(out, y.detach(), 1e-5, "gmu fwd")
    ldg = pad32(d)
    das = [torch.full((R, ldg), float("nan"), device=DEV).to(ctt) for _ in range(3)]
    dx1, dx2 = torch.empty(R, d, device=DEV), torch.empty(R, d, device=DEV)
    doutd = dout.to(DEV)
    ops.gmu2_bwd(dtype, [ops.gmu_problem(*dv, R, dout=doutd, da1=das[0], da2=das[1], dag=das[2], ldg=ldg, dx1=dx1, dx2=dx2)], d)
    t = 1e-5 if dtype == BPM_F32 else 1e-2
    for got, ref_, n in zip(das, (a1, a2, ag), ("da1", "da2", "dag")):
        close(got[:, :d].float(), ref_.grad, t, n)
        assert (got[:, d:].float() == 0).all()
    close(dx1, x1.grad, 1e-5, "dx1")
    close(dx2, x2.grad, 1e-5, "dx2")


# ---------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("variant", [GEMM_NT, GEMM_NN, GEMM_TN])
def test_gemm_hardware_bounded_loader(dtype, variant):
    """BPM_GEMM_KPAD_ZERO (buffer loads, no k-tail masking) == the masked loader, on ragged M / N / K with
    zero-padded rows, grouped with a second problem of another size."""
    from bpmult_amd.ops import F_KPAD
    outs = {}
    for flag in (0, F_KPAD):
        probs, keep, refs = [], [], []
        for (M, N, K, s0) in ((203, 77, 300, 60), (64, 300, 45, 70)):
            if variant == GEMM_NT:
                A, Ar = to_ct(rnd(M, K, seed=s0), dtype)
                Bm, Br = to_ct(rnd(N, K, seed=s0 + 1, scale=K ** -0.5), dtype)
                ref = Ar.double() @ Br.double().T
            elif variant == GEMM_NN:
                A, Ar = to_ct(rnd(M, K, seed=s0), dtype)
                Bm, Br = to_ct(rnd(K, N, seed=s0 + 1, scale=K ** -0.5), dtype)
                ref = Ar.double() @ Br.double()
            else:
                A, Ar = to_ct(rnd(K, M, seed=s0), dtype)
                Bm, Br = to_ct(rnd(K, N, seed=s0 + 1, scale=K ** -0.5), dtype)
                ref = Ar.double().T @ Br.double()
            out = torch.full((M, N), float("nan"), device=DEV)
            keep += [A, Bm, out]
            refs.append(ref)
            probs.append(ops.gemm_problem(A, Bm, out, M, N, K, A.shape[1], Bm.shape[1], N, flags=flag))
        ops.gemm_grouped(dtype, variant, probs)
        outs[flag] = [keep[2].cpu(), keep[5].cpu()]
        for o, r in zip(outs[flag], refs):
            close(o, r, tol(dtype) if dtype == BPM_F32 else 2e-3, f"gemm variant {variant} flag {flag}")
    for a, b in zip(outs[0], outs[F_KPAD]):
        assert torch.equal(a, b), "the two loaders must agree bit for bit"


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("variant,M,N,K", [(GEMM_TN, 200, 96, 333), (GEMM_NN, 333, 96, 200), (GEMM_TN, 768, 768, 1000), (GEMM_NN, 1000, 768, 768)])
def test_gemm_operand_column_view_at_the_end_of_its_allocation(dtype, variant, M, N, K):
    """The hardware-bounded loaders size their buffer descriptors from each operand VIEW: (rows - 1) leading dimensions
    plus the last row's width.  Operand A here is the LAST column block of a [rows, 3 ld] buffer (the dQ | dK | dV layout
    of engine.py) whose allocation ends exactly with that view, followed by poison: everything the kernels may touch past
    the view's last row must come back as zeros from the range check, never as the neighbour's bytes."""
    from bpmult_amd.ops import F_KPAD
    ct = ops.ct_torch(dtype)
    rows, cols = (K, M) if variant == GEMM_TN else (M, K)          # A is [K, M] (TN) or [M, K] (NN)
    ld = pad32(cols)
    A32 = rnd(rows, cols, seed=3)
    flat = torch.full((rows * 3 * ld + 4096,), float("nan"), dtype=ct)    # the view ends where the poison starts
    wide = flat[:rows * 3 * ld].view(rows, 3 * ld)
    wide.zero_()
    wide[:, 2 * ld:2 * ld + cols] = A32.to(ct)
    flat = flat.to(DEV)
    A = flat[:rows * 3 * ld].view(rows, 3 * ld)[:, 2 * ld:]
    Ar = A32.to(ct).float()
    Bm, Br = to_ct(rnd(K, N, seed=4, scale=K ** -0.5), dtype)
    ref = (Ar.double().T if variant == GEMM_TN else Ar.double()) @ Br.double()
    out = torch.full((M, N), float("nan"), device=DEV)
    ops.gemm_grouped(dtype, variant, [ops.gemm_problem(A, Bm, out, M, N, K, 3 * ld, Bm.shape[1], N, flags=F_KPAD)])
    close(out, ref, tol(dtype) if dtype == BPM_F32 else 2e-3, f"column-view operand, variant {variant}")


def test_indirect_seed_draws_the_same_masks():
    """BPM_SEED_INDIRECT | device pointer (the seed is read by the kernels when they run: captured graphs) == the same
    seed passed by value, for every kernel family with a dropout site: GEMM epilogues (tiled and LDS-DMA kernels),
    attention forward / backward, LayerNorm backward's fused cast, rows_cast, embed_pos, pack_rows."""
    from bpmult_amd._lib import SEED_INDIRECT
    from bpmult_amd.ops import F_KPAD
    dtype, seed = BPM_BF16, 0x1234_5678_9ABC_DEF
    sd = torch.tensor([seed], dtype=torch.int64, device=DEV)
    ind = ops.DeviceSeed(sd)
    assert int(ind) == SEED_INDIRECT | sd.data_ptr()
    with pytest.raises(ValueError, match="63-bit"):          # a by-value seed with bit 63 set is refused, not dereferenced
        ops.rows_cast(BPM_F32, [ops.cast_problem(sd.float(), 1, 1, 1, dst_f32=torch.zeros(1, device=DEV), ldf=1)], SEED_INDIRECT | 5)

    def both(fn):
        outs = []
        for s in (seed, ind):
            outs.append(fn(s))
        for a, b in zip(outs[0], outs[1]):
            assert torch.equal(a, b)
        return outs[0]

    for (M, N, K) in ((200, 96, 64), (1024, 768, 768)):          # tiled kernel, LDS-DMA kernel
        A, _ = to_ct(rnd(M, K, seed=1), dtype)
        W, _ = to_ct(rnd(N, K, seed=2, scale=K ** -0.5), dtype)

        def gemm(s):
            out = torch.zeros(M, N, device=DEV)
            ops.gemm_grouped(dtype, GEMM_NT, [ops.gemm_problem(A, W, out, M, N, K, A.shape[1], W.shape[1], N, drop_p=0.3, drop_site=5, flags=F_KPAD)], s)
            return [out]
        o = both(gemm)[0]
        assert 0.2 < float((o == 0).float().mean()) < 0.4

    B, H, T, S, dh, dhp = 2, 2, 70, 90, 64, 64
    q, k, v, do = (torch.randn(B, H, n, dhp, device=DEV).to(torch.bfloat16) for n in (T, S, S, T))

    def attn(s):
        o = torch.zeros(T * B, H * dhp, device=DEV, dtype=torch.bfloat16)
        lse, delta = torch.zeros(B, H, T, device=DEV), torch.zeros(B, H, T, device=DEV)
        dq, dk, dv = (torch.zeros(n * B, H * dhp, device=DEV, dtype=torch.bfloat16) for n in (T, S, S))
        p = ops.attn_problem(q, k, v, o, H * dhp, lse, B, H, T, S, dh, dhp, 1 + abs(S - T), dO=do, delta=delta, dQ=dq, lddq=H * dhp,
                             dK=dk, lddk=H * dhp, dV=dv, lddv=H * dhp, dq_scale=0.125, drop_p=0.2, drop_site=9)
        ops.attn_fwd(dtype, [p], s)
        ops.attn_bwd(dtype, [p], s)
        return [o, dq, dk, dv]
    both(attn)

    R, d = 300, 768
    x, dy = torch.randn(R, d, device=DEV), torch.randn(R, d, device=DEV)
    mean, rstd = x.mean(1), (x.var(1, unbiased=False) + 1e-5).rsqrt()
    gam = torch.ones(d, device=DEV)

    def ln(s):
        dx = torch.zeros(R, d, device=DEV)
        cast = torch.zeros(R, d, device=DEV, dtype=torch.bfloat16)
        cs = torch.zeros(d, device=DEV)
        ops.ln_bwd([ops.ln_problem(x, gam, None, mean, rstd, R, dy=dy, ldy=d, dx=dx, cast=cast, ldc=d, cast_colsum=cs, drop_p=0.1, drop_site=3)],
                   d, dtype, s)
        return [dx, cast]
    both(ln)

    def cast_embed_pack(s):
        dst = torch.zeros(R, d, device=DEV, dtype=torch.bfloat16)
        ops.rows_cast(dtype, [ops.cast_problem(x, d, R, d, dst_ct=dst, ldd=d, drop_p=0.1, drop_site=4)], s)
        xe = torch.randn(10, 3, d, device=DEV)
        oe = torch.zeros(10, 3, d, device=DEV)
        from bpmult_amd.engine import sinusoid_table
        ops.embed_pos_fwd([ops.embed_problem(xe, oe, 10, 3, drop_p=0.25, drop_site=6)], sinusoid_table(16, d, torch.device(DEV)), d, d ** 0.5, s)
        src = torch.randn(3, 10, 40, device=DEV)
        pk = torch.zeros(30, 64, device=DEV, dtype=torch.bfloat16)
        ops.pack_rows_fwd(dtype, [ops.pack_problem(3, 10, 40, 64, src=src, dst=pk, drop_p=0.25, drop_site=7)], s)
        return [dst, oe, pk]
    torch.manual_seed(0)
    a = cast_embed_pack(seed)
    torch.manual_seed(0)
    b = cast_embed_pack(ind)
    for u, w in zip(a, b):
        assert torch.equal(u, w)
    # a new value stored in the same word changes the masks of the next launch (nothing was baked in at launch time)
    o1 = gemm(ind)[0]
    sd.fill_(seed + 1)
    o2 = gemm(ind)[0]
    assert not torch.equal(o1, o2)


@pytest.mark.parametrize("dtype", DT)
def test_kv_layernorm_folding(dtype):
    """pack_weights(colscale) + fold_bias + unfold_grads reproduce y = LN(x; gamma, beta) W^T + b and its gradients
    from the un-affined normalised source (the engine's key / value path)."""
    import ctypes as C
    from bpmult_amd._lib import FoldDesc, PackDesc, UnfoldDesc
    R, d, n = 50, 300, 2 * 300
    ld = pad32(d)
    x = rnd(R, d, seed=81) * 1.5 + 0.2
    W = (rnd(n, d, seed=82) * d ** -0.5).requires_grad_(True)
    b = (0.1 * rnd(n, seed=83)).requires_grad_(True)
    gamma = (1 + 0.2 * rnd(d, seed=84)).requires_grad_(True)
    beta = (0.2 * rnd(d, seed=85)).requires_grad_(True)
    y = torch.nn.functional.layer_norm(x, (d,), gamma, beta, 1e-5) @ W.T + b
    dy = rnd(R, n, seed=86)
    (y * dy).sum().backward()
    xhat = torch.nn.functional.layer_norm(x, (d,))

    Wd, bd, gd, btd = (t.detach().to(DEV).contiguous() for t in (W, b, gamma, beta))
    shadow = torch.full((n, ld), float("nan"), device=DEV).to(ops.ct_torch(dtype))
    pd = PackDesc()
    pd.src, pd.dst, pd.rows, pd.cols, pd.ld, pd.src_ld, pd.dst_ld, pd.blk0 = Wd.data_ptr(), shadow.data_ptr(), n, d, ld, d, ld, 0
    pd.colscale = gd.data_ptr()
    ops.pack_weights(dtype, ops.device_table([pd]), 1, (n * ld + 1023) // 1024)
    close(shadow[:, :d].float(), (W * gamma).detach(), 1e-6 if dtype == BPM_F32 else 1e-2, "folded weight")
    assert (shadow[:, d:].float() == 0).all()

    bf = torch.empty(n, device=DEV)
    fd = FoldDesc()
    fd.W, fd.beta, fd.b, fd.out, fd.rows, fd.cols, fd.ldw, fd.blk0 = Wd.data_ptr(), btd.data_ptr(), bd.data_ptr(), bf.data_ptr(), n, d, d, 0
    ops.fold_bias(ops.device_table([fd]), 1, (n + 3) // 4)
    close(bf, (W @ beta + b).detach(), 1e-5, "folded bias")

    # forward through the folded operands (fp32 check of the algebra)
    close((xhat @ shadow[:, :d].float().cpu().T + bf.cpu()), y.detach(), 1e-4 if dtype == BPM_F32 else 5e-2, "folded forward")

    # backward: dWf = dy^T xhat, dbf = colsum(dy)  ->  dW, db, dgamma, dbeta
    dWf = (dy.T @ xhat).contiguous().to(DEV)
    dbf = dy.sum(0).contiguous().to(DEV)
    dW, db = torch.ones(n, d, device=DEV), torch.ones(n, device=DEV)
    dg, dbt = torch.ones(d, device=DEV), torch.ones(d, device=DEV)
    ud = UnfoldDesc()
    ud.dWf, ud.dbf, ud.W, ud.gamma, ud.beta = dWf.data_ptr(), dbf.data_ptr(), Wd.data_ptr(), gd.data_ptr(), btd.data_ptr()
    ud.dW, ud.dbias, ud.dgamma, ud.dbeta = dW.data_ptr(), db.data_ptr(), dg.data_ptr(), dbt.data_ptr()
    ud.rows, ud.cols, ud.ldw, ud.blk0 = n, d, d, 0
    ops.unfold_grads(ops.device_table([ud]), 1, (n + 15) // 16)
    close(dW, 1 + W.grad, 2e-4, "unfolded dW")
    close(db, 1 + b.grad, 2e-4, "unfolded dbias")
    close(dg, 1 + gamma.grad, 2e-4, "unfolded dgamma")
    close(dbt, 1 + beta.grad, 2e-4, "unfolded dbeta")


def test_attention_backward_halves_and_side_stream():
    """bpm_attn_bwd_dq + bpm_attn_bwd_dkv == bpm_attn_bwd; bpm_stream_create gives a usable stream."""
    import ctypes as C
    from bpmult_amd import _lib
    dtype, B, H, T, S, dh, dhp = BPM_BF16, 2, 2, 96, 80, 25, 32
    ct = ops.ct_torch(dtype)
    mk = lambda L, seed: (torch.randn(B, H, L, dhp, generator=torch.Generator().manual_seed(seed)) * 0.5).to(ct).to(DEV)
    Q, K, V, dO = mk(T, 1), mk(S, 2), mk(S, 3), mk(T, 4)
    ld = pad32(H * dh)
    O = torch.zeros(T * B, ld, device=DEV, dtype=ct)
    lse = torch.zeros(B, H, T, device=DEV)
    res = []
    for split in (False, True):
        delta = torch.zeros(B, H, T, device=DEV)
        dQ, dK, dV = (torch.zeros(n * B, ld, device=DEV, dtype=ct) for n in (T, S, S))
        p = ops.attn_problem(Q, K, V, O, ld, lse, B, H, T, S, dh, dhp, 1 + abs(S - T), dO=dO, delta=delta, dQ=dQ, lddq=ld,
                             dK=dK, lddk=ld, dV=dV, lddv=ld, dq_scale=dh ** -0.5, drop_p=0.1, drop_site=3)
        ops.attn_fwd(dtype, [p], 11)
        if split:
            out = C.c_void_p()
            _lib.check(_lib.lib().bpm_stream_create(1, C.byref(out)), "bpm_stream_create")
            side = torch.cuda.ExternalStream(out.value)
            ops.attn_bwd_dq(dtype, [p], 11)
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                ops.attn_bwd_dkv(dtype, [p], 11)
            torch.cuda.current_stream().wait_stream(side)
        else:
            ops.attn_bwd(dtype, [p], 11)
        torch.cuda.synchronize()
        res.append([t.float().cpu() for t in (dQ, dK, dV, delta)])
    for a, b in zip(*res):
        assert torch.equal(a, b)
    assert res[0][0].abs().sum() > 0 and res[0][1].abs().sum() > 0


def test_gathered_query_rows_keep_positions():
    """q_pos0 / q_stride (attention) and pos0 / pos_stride (embedding): rows {0, T-1} computed alone == the same
    rows of the full computation."""
    from oracle import bpmult_cpu as O
    dtype, B, H, T, S, dh, dhp = BPM_F32, 2, 2, 40, 40, 8, 32
    ct = ops.ct_torch(dtype)
    gen = torch.Generator().manual_seed(5)
    mk = lambda L: torch.cat([torch.randn(B, H, L, dh, generator=gen) * 0.5, torch.zeros(B, H, L, dhp - dh)], -1)
    Q, K, V, dO = mk(T), mk(S), mk(S), mk(T)
    ld = pad32(H * dh)
    idx = torch.tensor([0, T - 1])

    def run(Qx, dOx, Tn, **pos):
        O_ = torch.zeros(Tn * B, ld, device=DEV, dtype=ct)
        lse, delta = torch.zeros(B, H, Tn, device=DEV), torch.zeros(B, H, Tn, device=DEV)
        dQ, dK, dV = (torch.zeros(n * B, ld, device=DEV, dtype=ct) for n in (Tn, S, S))
        keep = [t.contiguous().to(ct).to(DEV) for t in (Qx, K, V, dOx)]
        p = ops.attn_problem(keep[0], keep[1], keep[2], O_, ld, lse, B, H, Tn, S, dh, dhp, 1 + abs(S - T), dO=keep[3], delta=delta,
                             dQ=dQ, lddq=ld, dK=dK, lddk=ld, dV=dV, lddv=ld, dq_scale=1.0, **pos)
        ops.attn_fwd(dtype, [p], 0)
        ops.attn_bwd(dtype, [p], 0)
        torch.cuda.synchronize()
        return O_.view(Tn, B, ld).cpu(), dQ.view(Tn, B, ld).cpu(), dK.cpu(), dV.cpu()

    # dense run with only rows {0, T-1} of dO non-zero, vs the 2-row run with original positions
    dO_sparse = torch.zeros_like(dO)
    dO_sparse[:, :, idx] = dO[:, :, idx]
    Of, dQf, dKf, dVf = run(Q, dO_sparse, T)
    Og, dQg, dKg, dVg = run(Q[:, :, idx], dO[:, :, idx], 2, q_pos0=0, q_stride=T - 1)
    close(Og, Of[idx], 1e-5, "gathered O")
    close(dQg, dQf[idx], 1e-5, "gathered dQ")
    close(dKg, dKf, 1e-5, "gathered dK")
    close(dVg, dVf, 1e-5, "gathered dV")
    assert float(Og[0].abs().sum()) > 0 and not torch.allclose(Og[0], Og[1])      # row 0 sees one key, row T-1 all

    d, T2 = 24, 9
    x = rnd(T2, 3, d, seed=91)
    table = O.sinusoid_table(T2 + 1, d).to(DEV)
    full, part = torch.empty(T2, 3, d, device=DEV), torch.empty(2, 3, d, device=DEV)
    xg = x[[0, T2 - 1]].contiguous().to(DEV)
    ops.embed_pos_fwd([ops.embed_problem(x.to(DEV), full, T2, 3)], table, d, math.sqrt(d), seed=0)
    ops.embed_pos_fwd([ops.embed_problem(xg, part, 2, 3, pos0=0, pos_stride=T2 - 1)], table, d, math.sqrt(d), seed=0)
    close(part, full[[0, T2 - 1]].cpu(), 1e-6, "gathered embed_pos")


@pytest.mark.parametrize("dtype", DT)
def test_gemm_tn_colsum_of_first_operand(dtype):
    """colsum_a: the bias gradient beside a weight gradient (column sums of A over the contraction rows), += semantics,
    ragged M / N, two problems in one launch."""
    probs, keep, refs = [], [], []
    for (M, N, K, s0) in ((300, 140, 1000, 70), (77, 300, 333, 72)):
        A, Ar = to_ct(rnd(K, M, seed=s0), dtype)
        Bm, Br = to_ct(rnd(K, N, seed=s0 + 1, scale=K ** -0.5), dtype)
        out = torch.zeros(M, N, device=DEV)
        cs = torch.ones(M, device=DEV)
        keep += [A, Bm, out, cs]
        refs.append((Ar.double().T @ Br.double(), 1 + Ar.double().sum(0)))
        probs.append(ops.gemm_problem(A, Bm, out, M, N, K, A.shape[1], Bm.shape[1], N, flags=F_ACCUM if s0 == 70 else 0, colsum_a=cs))
    ops.gemm_grouped(dtype, GEMM_TN, probs)
    for i, (rc, rs) in enumerate(refs):
        close(keep[4 * i + 2], rc, tol(dtype) if dtype == BPM_F32 else 2e-3, "tn product")
        close(keep[4 * i + 3], rs, 1e-4 if dtype == BPM_F32 else 2e-3, "colsum_a")


@pytest.mark.parametrize("n,B,d,Cn,Ns,pdrop", [(3, 2, 24, 6, (5, 7, 1), 0.0), (4, 5, 40, 13, (6, 3, 4), 0.0), (3, 11, 768, 6, (2, 2, 2), 0.3)])
def test_tail_fwd_bwd_against_torch(n, B, d, Cn, Ns, pdrop):
    """bpm_tail_fwd / bpm_tail_bwd (token pick + n-way gated fusion + residual head, mmtr.py:197-247, 806-808, 860-866)
    against the same arithmetic in torch fp64 with autograd, including the gradient of the returned gates."""
    from bpmult_amd._lib import TailDesc, TailGrads
    g = torch.Generator().manual_seed(3)
    r = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc)
    top = [r(N, B, d) for N in Ns]
    mid = [r(N, B, d) for N in Ns]
    extra = r(B, d) if n == 4 else None
    Wh = [r(d, d, sc=d ** -0.5) for _ in range(n)]
    Wg = [r(d, n * d, sc=(n * d) ** -0.5) for _ in range(n)]
    W1, b1, W2, b2, Wo, bo = r(d, d, sc=d ** -0.5), r(d), r(d, d, sc=d ** -0.5), r(d), r(Cn, d, sc=d ** -0.5), r(Cn)
    dlog, dzv = r(B, Cn), r(B, n * d, sc=0.3)
    leaves = [t.double().requires_grad_(True) for t in top + mid + ([extra] if n == 4 else []) + Wh + Wg + [W1, b1, W2, b2, Wo, bo]]
    it = iter(leaves)
    topr = [next(it) for _ in range(3)]
    midr = [next(it) for _ in range(3)]
    exr = next(it) if n == 4 else None
    Whr = [next(it) for _ in range(n)]
    Wgr = [next(it) for _ in range(n)]
    W1r, b1r, W2r, b2r, Wor, bor = (next(it) for _ in range(6))
    xs = [(topr[i] + midr[i])[0] + (topr[i] + midr[i])[-1] for i in range(3)] + ([exr] if n == 4 else [])
    cat = torch.cat(xs, -1)
    zs = [torch.sigmoid(cat @ Wgr[i].T) for i in range(n)]
    h = sum(zs[i] * torch.tanh(xs[i] @ Whr[i].T) for i in range(n))
    dm = drop_mult((B, d), pdrop, 21, (1 << 20) + 1).double()
    p1 = torch.relu(h @ W1r.T + b1r) * dm
    y = p1 @ W2r.T + b2r + h
    logits = y @ Wor.T + bor
    zcat = torch.cat(zs, -1)
    ((logits * dlog.double()).sum() + (zcat * dzv.double()).sum()).backward()

    dev = lambda t: t.to(DEV).contiguous()
    keep = dict(top=[dev(t) for t in top], mid=[dev(t) for t in mid], extra=dev(extra) if n == 4 else None,
                Wh=[dev(t) for t in Wh], Wg=[dev(t) for t in Wg], misc=[dev(t) for t in (W1, b1, W2, b2, Wo, bo)])
    z_ = lambda *s: torch.zeros(*s, device=DEV)
    buf = {k: z_(B, n * d) for k in ("x", "z", "t", "dzp", "dtp", "dx")}
    buf.update({k: z_(B, d) for k in ("h", "p1", "y", "dy", "dp1", "dh")})
    buf["logits"] = z_(B, Cn)
    t = TailDesc()
    t.B, t.d, t.n, t.C = B, d, n, Cn
    for i in range(3):
        t.N[i], t.top[i], t.mid[i] = Ns[i], keep["top"][i].data_ptr(), keep["mid"][i].data_ptr()
    t.extra = keep["extra"].data_ptr() if n == 4 else None
    for i in range(n):
        t.Wh[i], t.Wg[i] = keep["Wh"][i].data_ptr(), keep["Wg"][i].data_ptr()
    t.W1, t.b1, t.W2, t.b2, t.Wo, t.bo = (x.data_ptr() for x in keep["misc"])
    t.out_dropout, t.drop_site = pdrop, (1 << 20) + 1
    for k in ("x", "z", "t", "h", "p1", "y", "logits"):
        setattr(t, k, buf[k].data_ptr())
    ops.tail_fwd(t, 21)
    tl = 2e-5
    close(buf["logits"], logits.detach(), tl, "logits")
    close(buf["z"], zcat.detach(), tl, "gates")
    gr = TailGrads()
    dl, dzd = dev(dlog), dev(dzv)
    gr.dlogits, gr.dz = dl.data_ptr(), dzd.data_ptr()
    gW = dict(Wh=[1 + z_(d, d) for _ in range(n)], Wg=[1 + z_(d, n * d) for _ in range(n)],
              misc=[1 + z_(*s) for s in ((d, d), (d,), (d, d), (d,), (Cn, d), (Cn,))])       # += semantics: start from ones
    for i in range(n):
        gr.dWh[i], gr.dWg[i] = gW["Wh"][i].data_ptr(), gW["Wg"][i].data_ptr()
    gr.dW1, gr.db1, gr.dW2, gr.db2, gr.dWo, gr.dbo = (x.data_ptr() for x in gW["misc"])
    dtop, dmid = [z_(N, B, d) for N in Ns], [z_(N, B, d) for N in Ns]
    dex = z_(B, d) if n == 4 else None
    for i in range(3):
        gr.dtop[i], gr.dmid[i] = dtop[i].data_ptr(), dmid[i].data_ptr()
    gr.dextra = dex.data_ptr() if n == 4 else None
    for k in ("dy", "dp1", "dh", "dzp", "dtp", "dx"):
        setattr(gr, k, buf[k].data_ptr())
    ops.tail_bwd(t, gr)
    tg = 1e-4
    for i in range(3):
        close(dtop[i], topr[i].grad, tg, f"dtop{i}")
        close(dmid[i], midr[i].grad, tg, f"dmid{i}")
    if n == 4:
        close(dex, exr.grad, tg, "dextra")
    for i in range(n):
        close(gW["Wh"][i] - 1, Whr[i].grad, tg, f"dWh{i}")
        close(gW["Wg"][i] - 1, Wgr[i].grad, tg, f"dWg{i}")
    for nm, got, ref in zip(("dW1", "db1", "dW2", "db2", "dWo", "dbo"), gW["misc"], (W1r, b1r, W2r, b2r, Wor, bor)):
        close(got - 1, ref.grad, tg, nm)


@pytest.mark.parametrize("prec", ["f32", "bf16"])
def test_audio_encoder_frontend_against_torch_and_reference_fixture(prec):
    """AudioEncoder on the HIP path (window gather + grouped GEMM + adaptive pooling; mmtr.py:93-108) against
    torch.nn.functional.conv1d / adaptive_avg_pool1d in fp64 with autograd (outputs, weight / bias / input gradients), and
    against the reference's own output stored in the f8 fixture (`audio_feat`, [B, 96, 200])."""
    import os
    import numpy as np
    from detgen import det, det_param
    from bpmult_amd.models.bpmult import AudioEncoder
    enc = AudioEncoder()
    with torch.no_grad():
        for k, p in enc.named_parameters():
            p.copy_(torch.from_numpy(det_param("f8.audio_enc." + k, p.shape)))
    aud = torch.from_numpy(det("f8.aud", (2, 96, 1000)))
    ref_leaves = [p.detach().double().requires_grad_(True) for p in enc.parameters()]
    w1, b1, w2, b2 = ref_leaves
    xr = aud.double().requires_grad_(True)
    yr = torch.nn.functional.adaptive_avg_pool1d(torch.nn.functional.conv1d(torch.nn.functional.conv1d(xr, w1, b1, stride=2), w2, b2, stride=2), 200)
    wgt = torch.from_numpy(det("f8.audw", (2, 96, 200))).double()
    (yr * wgt).sum().backward()
    enc = enc.cuda()
    x = aud.cuda().requires_grad_(True)
    y = enc.encode(x, prec)                                   # [B, 200, 96]
    (y * wgt.float().cuda().transpose(1, 2)).sum().backward()
    t_out, t_g = (2e-5, 2e-4) if prec == "f32" else (2e-2, 4e-2)
    close(y.transpose(1, 2), yr.detach(), t_out, "audio_feat vs torch")
    g = dict(np.load(os.path.join(os.path.dirname(__file__), "golden", "f8_mmtrvapt.npz")))
    close(y.transpose(1, 2), torch.from_numpy(g["audio_feat"]).double(), t_out, "audio_feat vs reference fixture")
    close(x.grad, xr.grad, t_g, "d(audio)")
    for (k, p), r in zip(enc.named_parameters(), ref_leaves):
        close(p.grad, r.grad, t_g, "d(" + k + ")")


@pytest.mark.parametrize("variant,M,N,K", [(GEMM_NT, 1024, 768, 768), (GEMM_NT, 1000, 3072, 768), (GEMM_NN, 1024, 768, 3072),
                                          (GEMM_NN, 900, 1024, 1000), (GEMM_TN, 768, 3072, 4096), (GEMM_TN, 768, 768, 1000)])
def test_gemm_bf16x3_products(variant, M, N, K):
    """The parity-grade fast products (ops.gemm_grouped(..., x3=True) on fp32 operands: bpm_split_rows + BPM_BF16X3 on the
    LDS-DMA kernel, x y ~ hi hi + hi lo + lo hi) against fp64 on the SAME fp32 operands: <= 5e-5 of the result's scale
    (plain bf16 operands: ~4e-3), with the fused epilogues the encoder uses -- bias + residual + dropout -> f32,
    relu -> fp32 CT, bias gradient beside a weight gradient (column sums of X), += -- and ragged M / K."""
    from bpmult_amd.ops import F_KPAD
    g = torch.Generator().manual_seed(31)
    dev = DEV
    if variant == GEMM_NT:
        A, Bm = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=K ** -0.5)
        ref = A.double() @ Bm.double().T
    elif variant == GEMM_NN:
        A, Bm = rnd(M, K, seed=1), rnd(K, N, seed=2, scale=K ** -0.5)
        ref = A.double() @ Bm.double()
    else:
        A, Bm = rnd(K, M, seed=1), rnd(K, N, seed=2, scale=K ** -0.5)
        ref = A.double().T @ Bm.double()
    pad = lambda t: torch.nn.functional.pad(t, (0, pad32(t.shape[1]) - t.shape[1])).contiguous().to(dev)
    Ad, Bd = pad(A), pad(Bm)
    scale = float(ref.abs().max())
    # plain store
    out = torch.full((M, N), float("nan"), device=dev)
    p = ops.gemm_problem(Ad, Bd, out, M, N, K, Ad.shape[1], Bd.shape[1], N, flags=F_KPAD)
    tiles = ((M + 255) // 256) * ((N + 255) // 256)
    arr = ops.array(ops.GemmProblem, [p] * max(3, -(-100 // tiles)))      # (a weight-gradient launch takes the split path when it fills ~a third of the chip)
    ops.gemm_grouped(BPM_F32, variant, arr, x3=True)
    assert arr._x3_plan.ok, "this launch must take the split-bf16 path, not the exact fp32 fallback"
    e3 = float((out.cpu().double() - ref).abs().max()) / scale
    assert e3 <= 5e-5, f"bf16x3 max err / scale = {e3:.2e}"
    # exact f32 kernel on the same operands: the x3 product must be within a few 1e-5 of it too
    out32 = torch.zeros(M, N, device=dev)
    ops.gemm_grouped(BPM_F32, variant, [ops.gemm_problem(Ad, Bd, out32, M, N, K, Ad.shape[1], Bd.shape[1], N, flags=F_KPAD)])
    assert float((out - out32).abs().max()) / scale <= 5e-5
    if variant == GEMM_TN:                      # weight gradient: += and the bias gradient (column sums of X) beside it
        base = rnd(M, N, seed=5).to(dev)
        acc = base.clone()
        cs = torch.zeros(M, device=dev)
        scratch = [torch.zeros(M, N, device=dev) for _ in range(2)]
        ops.gemm_grouped(BPM_F32, variant, [ops.gemm_problem(Ad, Bd, acc, M, N, K, Ad.shape[1], Bd.shape[1], N, flags=F_KPAD | F_ACCUM,
                                                            colsum_a=cs)] +
                         [ops.gemm_problem(Ad, Bd, scratch[i % 2], M, N, K, Ad.shape[1], Bd.shape[1], N, flags=F_KPAD)
                          for i in range(max(2, -(-100 // tiles)))], x3=True)
        close(acc, ref + base.cpu().double(), 5e-5, "accumulate")
        close(cs, A.double().sum(0), 5e-5, "colsum_a")
    else:                                       # forward / data-gradient epilogues
        bias, resid = rnd(N, seed=6).to(dev), rnd(M, N, seed=7).to(dev)
        o1 = torch.full((M, N), float("nan"), device=dev)
        o2 = torch.full((M, pad32(N)), float("nan"), device=dev)
        ps = [ops.gemm_problem(Ad, Bd, o1, M, N, K, Ad.shape[1], Bd.shape[1], N, bias_n=bias, resid=resid, ldr=N, drop_p=0.25, drop_site=9, flags=F_KPAD),
              ops.gemm_problem(Ad, Bd, o2, M, N, K, Ad.shape[1], Bd.shape[1], pad32(N), bias_n=bias, flags=F_KPAD | F_RELU, out_kind=OUT_CT)]
        ops.gemm_grouped(BPM_F32, variant, ps, seed=77, x3=True)
        dm = drop_mult((M, N), 0.25, 77, 9).double()
        close(o1, (ref + bias.cpu().double()) * dm + resid.cpu().double(), 5e-5 * 2, "bias + dropout + residual")
        close(o2[:, :N], (ref + bias.cpu().double()).clamp_min(0), 5e-5, "relu -> CT (fp32)")
        assert float(o2[:, N:].abs().max()) == 0.0 if pad32(N) > N else True


@pytest.mark.parametrize("variant", [GEMM_NT, GEMM_NN, GEMM_TN])
def test_gemm_bf16x3_at_the_model_launch_shapes(variant):
    """The bf16x3 products at the launch shapes of the headline model (hidden 768, six problems of 4096 rows: the 320 x 256
    tile configuration for N = 768 products; 24 weight gradients of 768 x 768 x 4096 with their bias column sums in one
    launch) -- every problem of the launch against fp64 on the same fp32 operands."""
    from bpmult_amd.ops import F_KPAD
    dev = DEV
    if variant == GEMM_TN:
        G, M, N, K = 24, 768, 768, 4096
    else:
        G, M, N, K = 6, 4096, 768, 3072
    probs, refs, outs, sums, keep = [], [], [], [], []
    for gi in range(G):
        if variant == GEMM_NT:
            A, Bm = rnd(M, K, seed=100 + gi), rnd(N, K, seed=200 + gi, scale=K ** -0.5)
            ref = A.double() @ Bm.double().T
        elif variant == GEMM_NN:
            A, Bm = rnd(M, K, seed=100 + gi), rnd(K, N, seed=200 + gi, scale=K ** -0.5)
            ref = A.double() @ Bm.double()
        else:
            A, Bm = rnd(K, M, seed=100 + gi), rnd(K, N, seed=200 + gi, scale=K ** -0.5)
            ref = A.double().T @ Bm.double()
        Ad, Bd = A.to(dev), Bm.to(dev)
        out = torch.full((M, N), float("nan"), device=dev)
        cs = torch.zeros(M, device=dev) if variant == GEMM_TN else None
        probs.append(ops.gemm_problem(Ad, Bd, out, M, N, K, Ad.shape[1], Bd.shape[1], N, flags=F_KPAD, colsum_a=cs))
        refs.append(ref); outs.append(out); sums.append((cs, A.double().sum(0) if cs is not None else None)); keep += [Ad, Bd]
    arr = ops.array(ops.GemmProblem, probs)
    ops.gemm_grouped(BPM_F32, variant, arr, x3=True)
    assert arr._x3_plan.ok
    torch.cuda.synchronize()
    for gi in range(G):
        close(outs[gi], refs[gi], 5e-5, f"problem {gi}")
        if sums[gi][0] is not None:
            close(sums[gi][0], sums[gi][1], 5e-5, f"column sums of X, problem {gi}")


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("variant,M,N,K", [(GEMM_NT, 16, 768, 768), (GEMM_NT, 2, 3072, 768), (GEMM_NT, 9, 300, 300),
                                          (GEMM_NN, 16, 768, 3072), (GEMM_NN, 4, 3072, 768), (GEMM_NN, 16, 300, 1200)])
def test_gemm_skinny_rows(dtype, variant, M, N, K):
    """Problems of at most 16 rows (the level-2 query side under dead-row elimination: rows {0, N-1} x batch) take the
    skinny kernel -- K split over the four waves of a workgroup, operands straight from global memory, partial tiles
    summed through LDS -- with the shared epilogue: every fused epilogue the encoder uses on such launches, against fp64
    on the CT-rounded operands; dropout masks equal to the other kernels' (host restatement of the hash)."""
    from bpmult_amd.ops import F_KPAD
    dev = DEV
    A, Ar = to_ct(rnd(M, K, seed=51), dtype)
    if variant == GEMM_NT:
        Bm, Br = to_ct(rnd(N, K, seed=52, scale=K ** -0.5), dtype)
        ref = Ar.double() @ Br.double().T
    else:
        Bm, Br = to_ct(rnd(K, N, seed=52, scale=K ** -0.5), dtype)
        ref = Ar.double() @ Br.double()
    t = tol(dtype) if dtype == BPM_F32 else 4e-3
    bias, resid = rnd(N, seed=53).to(dev), rnd(M, N, seed=54).to(dev)
    gate, gr = to_ct(rnd(M, N, seed=55), dtype)
    base = rnd(M, N, seed=56).to(dev)
    o1 = torch.full((M, N), float("nan"), device=dev)
    o2 = torch.full((M, pad32(N)), float("nan"), device=dev).to(ops.ct_torch(dtype))
    o3 = torch.full((M, pad32(N)), float("nan"), device=dev).to(ops.ct_torch(dtype))
    o5 = base.clone()
    cs = torch.zeros(N, device=dev)
    ps = [ops.gemm_problem(A, Bm, o1, M, N, K, A.shape[1], Bm.shape[1], N, bias_n=bias, resid=resid, ldr=N, drop_p=0.25, drop_site=9, flags=F_KPAD),
          ops.gemm_problem(A, Bm, o2, M, N, K, A.shape[1], Bm.shape[1], pad32(N), bias_n=bias, flags=F_KPAD | F_RELU, drop_p=0.1, drop_site=3, out_kind=OUT_CT),
          ops.gemm_problem(A, Bm, o3, M, N, K, A.shape[1], Bm.shape[1], pad32(N), gate=gate, ldg=gate.shape[1], gate_scale=1.25, colsum=cs, flags=F_KPAD, out_kind=OUT_CT),
          ops.gemm_problem(A, Bm, o5, M, N, K, A.shape[1], Bm.shape[1], N, flags=F_KPAD | F_ACCUM)]
    heads = None
    if N % 12 == 0 and M % 2 == 0:                       # head-major scatter: rows are (t, b) with B = M / 2
        Hn, dh = 12, N // 12
        dhp = 32 if dh <= 32 else 64 if dh <= 64 else 128 if dh <= 128 else 256
        if dhp <= 128:
            heads = torch.zeros(M // 2, Hn, 2, dhp, device=dev, dtype=ops.ct_torch(dtype))
            ps.append(ops.gemm_problem(A, Bm, heads, M, N, K, A.shape[1], Bm.shape[1], 0, bias_n=bias, alpha=0.5, out_kind=OUT_HEADS,
                                       heads=(M // 2, Hn, 2, dh, dhp), flags=F_KPAD))
    ops.gemm_grouped(dtype, variant, ps, seed=77)
    torch.cuda.synchronize()
    rb = ref + bias.cpu().double()
    close(o1, rb * drop_mult((M, N), 0.25, 77, 9).double() + resid.cpu().double(), 2 * t, "bias + dropout + residual")
    close(o2[:, :N], rb.clamp_min(0) * drop_mult((M, N), 0.1, 77, 3).double(), 2 * t, "relu + dropout -> CT")
    if pad32(N) > N:
        assert float(o2[:, N:].float().abs().max()) == 0.0 and float(o3[:, N:].float().abs().max()) == 0.0
    g3 = torch.where(gr.double() > 0, ref * 1.25, torch.zeros_like(ref))
    close(o3[:, :N], g3, 2 * t, "gate -> CT")
    close(cs, g3.sum(0), 4 * t, "column sums")
    close(o5, ref + base.cpu().double(), t, "+=")
    if heads is not None:
        want = ((rb * 0.5).reshape(2, M // 2, 12, N // 12)).permute(1, 2, 0, 3)      # [B, H, T, dh]
        close(heads[..., :N // 12], want, 2 * t, "head-major")


@pytest.mark.parametrize("variant,N,K", [(GEMM_NT, 300, 300), (GEMM_NT, 300, 1200), (GEMM_NN, 300, 1200), (GEMM_NN, 300, 300)])
def test_gemm_hidden_300_products_take_the_two_resident_tiles(variant, N, K):
    """Hidden 300: 300 columns fill 78 % of 128-wide tiles but 59 % of 256-wide ones -- such launches go to the LDS-DMA
    kernel's 256 x 128 configuration instead of the 128 x 64 kernel (the dispatcher's `narrow` rule; BASELINE configs[1] /
    configs[3]: 8.97 -> 8.79 and 56.8 -> 50.6 ms per step).  Six problems of 4000 rows with the encoder's epilogues against
    fp64, and the launch profiler's tally says which kernel ran."""
    import ctypes as C
    from bpmult_amd import _lib
    M, G = 4000, 6
    ld = lambda n: pad32(n)
    ps, checks = [], []
    for gi in range(G):
        A, Ar = to_ct(rnd(M, K, seed=60 + gi), BPM_BF16, ld(K))
        if variant == GEMM_NT:
            Bm, Br = to_ct(rnd(N, K, seed=70 + gi, scale=K ** -0.5), BPM_BF16, ld(K))
            ref = Ar.double() @ Br.double().T
        else:
            Bm, Br = to_ct(rnd(K, N, seed=70 + gi, scale=K ** -0.5), BPM_BF16, ld(N))
            ref = Ar.double() @ Br.double()
        bias, resid = rnd(N, seed=80 + gi).to(DEV), rnd(M, N, seed=90 + gi).to(DEV)
        out = torch.full((M, N), float("nan"), device=DEV)
        ps.append(ops.gemm_problem(A, Bm, out, M, N, K, A.shape[1], Bm.shape[1], N, bias_n=bias, resid=resid, ldr=N, flags=ops.F_KPAD))
        checks.append((out, ref + bias.cpu().double() + resid.cpu().double(), A, Bm, bias, resid))   # (the problems hold raw pointers)
    L = _lib.lib()
    kinds = _lib.PROF_KINDS
    _lib.prof_enable(sum(1 << k for k in kinds.values()))
    try:
        ops.gemm_grouped(BPM_BF16, variant, ps)
        torch.cuda.synchronize()
        n = {}
        for name, k in kinds.items():
            ms, work, cnt = C.c_double(), C.c_double(), C.c_int()
            _lib.check(L.bpm_prof_collect(k, C.byref(ms), C.byref(work), C.byref(cnt)), "bpm_prof_collect")
            n[name] = cnt.value
    finally:
        _lib.prof_enable(0)
    dma = "gemm_dma_nt" if variant == GEMM_NT else "gemm_dma_nn"
    assert n[dma] == 1 and n["gemm_nt"] == 0 and n["gemm_nn"] == 0, n
    for gi, chk in enumerate(checks):
        close(chk[0], chk[1], 4e-3, f"problem {gi}")


@pytest.mark.parametrize("variant,M,N,K", [(GEMM_NT, 1000, 520, 328), (GEMM_NN, 1000, 520, 328), (GEMM_TN, 520, 1000, 1300),
                                          (GEMM_NT, 4096, 768, 768), (GEMM_TN, 768, 3072, 4096)])
def test_gemm_two_resident_config(variant, M, N, K):
    """The 256 x 128 tile with 32-k stages (three stages = 72 KB, <= 128 registers: two workgroups per CU; picked for the
    FFN weight gradients and the twelve-problem K / V projections) forced on ragged shapes -- rows past M, a k tail that
    is not a whole stage, a column tail -- and on the model's shapes, all three operand arrangements, against fp64."""
    from bpmult_amd import _lib
    pad64 = lambda n: (n + 63) // 64 * 64
    if variant == GEMM_TN:
        A, Ar = to_ct(rnd(K, M, seed=21), BPM_BF16, pad64(M))
        Bm, Br = to_ct(rnd(K, N, seed=22, scale=K ** -0.5), BPM_BF16, pad64(N))
        ref = Ar.double().T @ Br.double()
    else:
        A, Ar = to_ct(rnd(M, K, seed=21), BPM_BF16, pad64(K))
        if variant == GEMM_NT:
            Bm, Br = to_ct(rnd(N, K, seed=22, scale=K ** -0.5), BPM_BF16, pad64(K))
            ref = Ar.double() @ Br.double().T
        else:
            Bm, Br = to_ct(rnd(K, N, seed=22, scale=K ** -0.5), BPM_BF16)
            ref = Ar.double() @ Br.double()
    base = rnd(M, N, seed=24).to(DEV)
    out, acc = torch.full((M, N), float("nan"), device=DEV), base.clone()
    ps = [ops.gemm_problem(A, Bm, out, M, N, K, A.shape[1], Bm.shape[1], N, flags=ops.F_KPAD),
          ops.gemm_problem(A, Bm, acc, M, N, K, A.shape[1], Bm.shape[1], N, flags=ops.F_KPAD | F_ACCUM)]
    with lab_library() as L:
        _lib.check(L.bpm_debug_gemm_force(6), "force")
        ops.gemm_grouped(BPM_BF16, variant, ps)
        torch.cuda.synchronize()
    close(out, ref, 2e-3, f"two-resident v{variant}")
    close(acc, ref + base.cpu().double(), 2e-3, f"two-resident += v{variant}")
