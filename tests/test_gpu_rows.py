"""The single-pass, register-resident kernel (prl_policy_rows_fwd_bwd) through the
C ABI: against the CPU oracle, and against the two-kernel path it stands in for.

Tolerances as in test_gpu_parity.py: 1e-4 for fp32 per-token values, one bf16
unit in the last place for the gradient."""

import ctypes
import math

import pytest
import torch

import prime_rl_cse_amd as prl
from oracle import ref_cpu as O
from prime_rl_cse_amd import _capi as C
from tests.helpers import bf16_ulp_close, make_packed_row

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def close(a, b, atol=1e-4, rtol=1e-4):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    assert a.shape == b.shape, (a.shape, b.shape)
    ok = ((a - b).abs() <= atol + rtol * b.abs()) | (a.isnan() & b.isnan()) | (a == b)
    assert bool(ok.all()), f"max abs diff {(a - b).abs().nan_to_num().max().item():.3e}"


def rows_call(logits, ids, adv, old, mask, temp, clip, scale, skip, stride=None):
    """Direct C-ABI call; logits [T, >=V] bf16 on the device (row stride = logits.stride(0))."""
    lib = C.load()
    t = ids.numel()
    v = logits.shape[1] if stride is None else stride[1]
    rs = logits.stride(0)
    assert lib.prl_policy_rows_supported(C.PRL_BF16, t, v, rs, C.PRL_BF16, v) == 1
    lp = torch.full((t,), 7.0, device=DEV)
    ent = torch.full((t,), 7.0, device=DEV)
    lse = torch.full((t,), 7.0, device=DEV)
    d = torch.full((t, v), 7.0, dtype=torch.bfloat16, device=DEV)
    C.check(
        lib.prl_policy_rows_fwd_bwd(
            C.ptr(logits), C.PRL_BF16, t, v, rs, C.ptr(ids), C.ptr(adv), C.ptr(old), C.ptr(mask), int(skip),
            temp, clip, float(scale), C.ptr(lp), C.ptr(ent), C.ptr(lse), C.ptr(d), C.PRL_BF16, v,
            C.stream_ptr(torch.device(DEV)),
        ),
        "prl_policy_rows_fwd_bwd",
    )
    torch.cuda.synchronize()
    return lp, ent, lse, d


def seeded_case(seed, lengths, pad, vocab, temp, mask_p=0.7, spread=0.2):
    gen = torch.Generator().manual_seed(seed)
    ids, pos, mask = make_packed_row(gen, lengths, pad, vocab, mask_p=mask_p)
    t = ids.numel()
    logits = (torch.randn(1, t, vocab, generator=gen) * 3).to(torch.bfloat16)
    adv = torch.randn(t, generator=gen)
    with torch.no_grad():
        lp0 = O.selective_log_softmax(O.shift_logits(logits.float()) / temp, ids.unsqueeze(0))[0]
    old = lp0 + spread * torch.randn(t, generator=gen)
    return logits, ids, pos, mask, adv, old


# one case per register-array size the kernel is built for (2, 4, 8, 12, 16, 19 chunks per lane),
# with vocabularies that end in the middle of a slab and in the middle of a wave
@pytest.mark.parametrize(
    "vocab,temp,skip",
    [
        (8, 1.0, False),
        (1000, 0.7, False),
        (16384, 1.0, True),
        (32000, 0.6, False),
        (50264, 1.0, True),
        (65536 + 8, 1.3, False),
        (98304, 1.0, False),
        (128256, 0.9, True),
        (151936, 1.0, False),
        (152064, 0.6, True),
        (155648, 1.0, False),
    ],
)
def test_rows_kernel_against_oracle(vocab, temp, skip):
    lengths, pad = [7, 5, 9], 2
    logits, ids, pos, mask, adv, old = seeded_case(vocab, lengths, pad, vocab, temp)
    old[3] -= 5.0  # a ratio far above the clip: that token's gradient is zero
    t = ids.numel()
    cfg = prl.LossConfig()
    scale = int(mask.sum())
    b = lambda x: x.unsqueeze(0)
    want = O.policy_epilogue(logits, b(ids), b(pos), b(adv), b(old), b(mask), temp, cfg, scale)
    lp, ent, lse, d = rows_call(
        logits[0].to(DEV), ids.to(DEV), adv.to(DEV), old.to(DEV), mask.to(torch.uint8).to(DEV), temp, cfg.clip_ratio,
        scale, skip,
    )
    sel = mask if skip else torch.ones_like(mask)
    close(lp.cpu()[sel], want["logprobs"][0][sel])
    close(ent.cpu()[sel], want["entropy"][0][sel])
    if skip:  # unscored tokens report zeros
        assert not lp.cpu()[~mask].any() and not ent.cpu()[~mask].any()
    assert bf16_ulp_close(d.cpu(), want["dlogits"][0], ulps=1, atol=1e-9)
    assert not d[t - 1].any()
    assert abs(lse[0].item() - (0.0 if skip and not mask[0] else math.log(vocab))) < 1e-5


@pytest.mark.parametrize("vocab,t", [(151936, 96), (32768, 300)])
def test_rows_kernel_equals_two_kernel_path(vocab, t):
    """policy_loss with the single-pass kernel on and off: same numbers to the
    stated tolerances, zero rows in the same places."""
    import os

    lengths = [t // 3, t // 3, t - 2 * (t // 3)]
    logits, ids, pos, mask, adv, old = seeded_case(5, lengths, 0, vocab, 0.8)
    b = lambda x: x.unsqueeze(0).to(DEV)
    cfg = prl.LossConfig()
    out = {}
    for flag in ("1", "0"):
        os.environ["PRL_FUSED_ROWS"] = flag
        try:
            x = logits.to(DEV).requires_grad_(True)
            loss, aux = prl.policy_loss(x, b(ids), b(pos), b(adv), b(old), b(mask), 0.8, cfg, int(mask.sum()),
                                        skip_masked=True)
            loss.backward()
            out[flag] = (loss.detach(), aux, x.grad)
        finally:
            os.environ.pop("PRL_FUSED_ROWS", None)
    (l1, a1, g1), (l0, a0, g0) = out["1"], out["0"]
    close(l1, l0, atol=1e-6)
    for key in ("logprobs", "entropy", "importance_ratio", "clipped_importance_ratio"):
        close(a1[key], a0[key], atol=2e-5)
    assert torch.equal(a1["is_clipped"], a0["is_clipped"])
    n = int(a1["n_seg"].item())
    assert n == int(a0["n_seg"].item()) == 3
    assert torch.equal(a1["seg_start"][: n + 1], a0["seg_start"][: n + 1])
    assert bf16_ulp_close(g1.cpu(), g0.cpu(), ulps=1, atol=1e-9)
    assert torch.equal(g1 == 0, g0 == 0)


def test_rows_kernel_row_stride_and_single_token():
    vocab, pad = 4096, 64
    logits, ids, pos, mask, adv, old = seeded_case(9, [6, 6], 0, vocab, 1.0)
    t = ids.numel()
    wide = torch.zeros(t, vocab + pad, dtype=torch.bfloat16)
    wide[:, :vocab] = logits[0]
    wide[:, vocab:] = 50.0  # must never be read
    wd = wide.to(DEV)
    cfg = prl.LossConfig()
    want = O.policy_epilogue(logits, ids[None], pos[None], adv[None], old[None], mask[None], 1.0, cfg, 3)
    lp, ent, _, d = rows_call(wd, ids.to(DEV), adv.to(DEV), old.to(DEV), mask.to(torch.uint8).to(DEV), 1.0,
                              cfg.clip_ratio, 3, False, stride=(t, vocab))
    close(lp, want["logprobs"][0])
    close(ent, want["entropy"][0])
    assert bf16_ulp_close(d.cpu(), want["dlogits"][0], ulps=1, atol=1e-9)
    # T = 1: the only token is scored by the all-zero row, the only gradient row is zero
    one = torch.ones(1, device=DEV)
    lp, ent, lse, d = rows_call(wd[:1], ids[:1].to(DEV), one, one, one.to(torch.uint8), 1.0, 8.0, 1, False,
                                stride=(1, vocab))
    assert abs(lp.item() + math.log(vocab)) < 1e-6 and abs(ent.item() - math.log(vocab)) < 1e-6
    assert not d.any()


def test_rows_kernel_refuses_what_it_cannot_hold():
    lib = C.load()
    assert lib.prl_policy_rows_supported(C.PRL_F32, 16, 4096, 4096, C.PRL_F32, 4096) == 0  # fp32 rows
    assert lib.prl_policy_rows_supported(C.PRL_BF16, 16, 1001, 1001, C.PRL_BF16, 1001) == 0  # V not a multiple of 8
    assert lib.prl_policy_rows_supported(C.PRL_BF16, 16, 155656, 155656, C.PRL_BF16, 155656) == 0  # too wide
    assert lib.prl_policy_rows_supported(C.PRL_BF16, 16, 4096, 4100, C.PRL_BF16, 4096) == 0  # rows not 16-byte apart
    x = torch.zeros(16, 1001, dtype=torch.bfloat16, device=DEV)
    z = torch.zeros(16, device=DEV)
    rc = lib.prl_policy_rows_fwd_bwd(C.ptr(x), C.PRL_BF16, 16, 1001, 1001, C.ptr(z), C.ptr(z), C.ptr(z), C.ptr(z), 0,
                                     1.0, 8.0, 1.0, C.ptr(z), None, None, C.ptr(x), C.PRL_BF16, 1001, None)
    assert rc != 0 and b"not supported" in lib.prl_last_error()
    # and policy_loss quietly takes the two-kernel path there
    xl = torch.randn(1, 6, 1001, device=DEV).to(torch.bfloat16).requires_grad_(True)
    i = torch.randint(0, 1001, (1, 6), device=DEV)
    loss, _ = prl.policy_loss(xl, i, torch.arange(6, device=DEV)[None], torch.ones(1, 6, device=DEV),
                              torch.full((1, 6), -7.0, device=DEV), torch.ones(1, 6, dtype=torch.bool, device=DEV),
                              1.0, prl.LossConfig(), 6)
    loss.backward()
    assert xl.grad is not None and bool(xl.grad.isfinite().all())


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("scale", [1.0, 0.375, -2.0])
def test_scale_rows_if_not_one(dtype, scale):
    lib = C.load()
    gen = torch.Generator().manual_seed(3)
    t, v, stride = 37, 1003, 1010  # odd sizes: rows not 16-byte aligned
    buf = torch.randn(t, stride, generator=gen).to(dtype).to(DEV)
    before = buf.clone()
    s = torch.tensor([scale], device=DEV)
    C.check(lib.prl_scale_rows_if_not_one(C.ptr(buf), C.dtype_code(buf), t, v, stride, C.ptr(s),
                                          C.stream_ptr(torch.device(DEV))), "prl_scale_rows_if_not_one")
    torch.cuda.synchronize()
    want = (before[:, :v].float() * scale).to(dtype)
    assert torch.equal(buf[:, :v], want)  # both factors are exact in bf16, so the product rounds once
    assert torch.equal(buf[:, v:], before[:, v:])  # the gap between rows is untouched


def test_incoming_gradient_other_than_one_and_second_backward():
    logits, ids, pos, mask, adv, old = seeded_case(21, [10, 14], 0, 32000, 1.0)
    b = lambda x: x.unsqueeze(0).to(DEV)
    cfg = prl.LossConfig()
    args = (b(ids), b(pos), b(adv), b(old), b(mask), 1.0, cfg, int(mask.sum()))
    x1 = logits.to(DEV).requires_grad_(True)
    prl.policy_loss(x1, *args)[0].backward()
    x2 = logits.to(DEV).requires_grad_(True)
    loss2, _ = prl.policy_loss(x2, *args)
    (loss2 * 0.3).backward(retain_graph=True)
    want = (x1.grad.float() * 0.3).to(torch.bfloat16)
    assert bf16_ulp_close(x2.grad.cpu(), want.cpu(), ulps=1, atol=1e-12)
    with pytest.raises(RuntimeError, match="second backward"):
        (loss2 * 0.3).backward()
    # the constant from unit_gradient() is recognised by its storage: same result, no rescale launch
    x4 = logits.to(DEV).requires_grad_(True)
    loss4, _ = prl.policy_loss(x4, *args)
    timer = C.KernelTimer()
    C.set_timer(timer)
    try:
        loss4.backward(prl.unit_gradient(DEV))
    finally:
        C.set_timer(None)
    torch.cuda.synchronize()
    assert torch.equal(x4.grad, x1.grad) and "prl_scale_rows_if_not_one" not in timer.summary()
    # without a gradient to compute, nothing is written for one
    x3 = logits.to(DEV)
    loss3, aux3 = prl.policy_loss(x3, *args)
    assert not loss3.requires_grad
    close(loss3, loss2, atol=1e-6)


def test_rows_kernel_special_values_follow_the_two_kernel_path():
    """Rows that make the sampled reference point useless (a huge logit far from every first chunk), rows with
    -inf, +inf and NaN logits, and a label outside the vocabulary: same numbers and the same NaN pattern as the
    two-kernel path, which follows the reference's arithmetic."""
    import os

    vocab, t = 32768, 12
    gen = torch.Generator().manual_seed(77)
    logits = (torch.randn(1, t, vocab, generator=gen) * 2).to(torch.bfloat16)
    logits[0, 1] = -200.0
    logits[0, 1, 20000] = 200.0      # exp2 of the difference overflows against any sampled reference: exact fallback
    logits[0, 2, 30000] = 90.0       # large but summable
    logits[0, 3, 100:200] = float("-inf")  # masked-out vocabulary entries: entropy becomes NaN (0 * -inf), as in torch
    logits[0, 4, 5] = float("inf")
    logits[0, 5, 7] = float("nan")
    logits[0, 6] = -300.0            # a constant row far below zero
    ids = torch.randint(0, vocab, (t,), generator=gen)
    ids[2] = 20000                   # the label of the overflow row (token 2 is scored by row 1)
    ids[8] = vocab + 3               # out of range
    pos = torch.arange(t)
    mask = torch.ones(t, dtype=torch.bool)
    adv = torch.randn(t, generator=gen)
    old = torch.full((t,), -10.0)
    old[2] = 0.0
    b = lambda x: x.unsqueeze(0).to(DEV)
    out = {}
    for flag in ("1", "0"):
        os.environ["PRL_FUSED_ROWS"] = flag
        try:
            x = logits.to(DEV).requires_grad_(True)
            loss, aux = prl.policy_loss(x, b(ids), b(pos), b(adv), b(old), b(mask), 1.0, prl.LossConfig(clip_ratio=1e9), t)
            loss.backward()
            out[flag] = (aux["logprobs"][0].cpu(), aux["entropy"][0].cpu(), x.grad[0].float().cpu())
        finally:
            os.environ.pop("PRL_FUSED_ROWS", None)
    (lp1, en1, g1), (lp0, en0, g0) = out["1"], out["0"]
    assert torch.equal(lp1.isnan(), lp0.isnan()) and torch.equal(en1.isnan(), en0.isnan())
    assert lp1.isnan().tolist() == [False] * 5 + [True, True, False, True, False, False, False]  # +inf, NaN rows, bad label
    assert en1.isnan()[4] and not en1.isnan()[2]
    close(lp1, lp0, atol=2e-5)
    close(en1, en0, atol=2e-5)
    assert abs(lp1[2].item()) < 1e-5          # the 200 logit takes all the probability
    assert abs(lp1[7].item() + math.log(vocab)) < 1e-4  # constant row: uniform
    assert torch.equal(g1.isnan(), g0.isnan())
    fin = ~g1.isnan()
    assert bf16_ulp_close(g1[fin].to(torch.bfloat16), g0[fin].to(torch.bfloat16), ulps=1, atol=1e-9)


# --------------------------------------------------------------------------- sequence mode in one pass


def _seq_call(logits, ids, pos, adv, old, mask, temp, length_norm, clip, scale, skip, max_len, spins):
    """prl_policy_rows_seq_fwd_bwd, prl_policy_tail and prl_logprob_bwd_selected through the C ABI, as
    `policy_loss` chains them.  Returns logprob, entropy, dlogits, loss, per-sample ratios, n_seg, pending."""
    lib = C.load()
    t, v = logits.shape
    st = C.stream_ptr(torch.device(DEV))
    lp, ent, lse, g = (torch.full((t,), 7.0, device=DEV) for _ in range(4))
    d = torch.full((t, v), 7.0, dtype=torch.bfloat16, device=DEV)
    pending = torch.full((t,), 9, dtype=torch.uint8, device=DEV)
    ws_bytes = int(lib.prl_policy_rows_seq_workspace_bytes(t))
    ws = torch.full((ws_bytes,), 0xAB, dtype=torch.uint8, device=DEV)  # the call clears what it needs
    m8 = mask.to(torch.uint8)
    C.check(lib.prl_policy_rows_seq_fwd_bwd(
        C.ptr(logits), C.PRL_BF16, t, v, logits.stride(0), C.ptr(ids), C.ptr(pos), C.ptr(adv), C.ptr(old), C.ptr(m8),
        int(skip), temp, int(length_norm), clip, float(scale), C.ptr(lp), C.ptr(ent), C.ptr(lse), C.ptr(d), C.PRL_BF16,
        v, C.ptr(pending), max_len, spins, C.ptr(ws), ws_bytes, st), "prl_policy_rows_seq_fwd_bwd")
    n_max = t // 2 + 1
    loss = torch.zeros(1, device=DEV)
    ratio, clipped, is_clipped, seg_loss = (torch.zeros(n_max, device=DEV) for _ in range(4))
    seg_start = torch.zeros(t + 1, dtype=torch.int64, device=DEV)
    n_seg = torch.zeros(1, dtype=torch.int32, device=DEV)
    C.check(lib.prl_policy_tail(
        C.ptr(pos), C.ptr(lp), C.ptr(old), C.ptr(adv), C.ptr(m8), t, C.RATIO_SEQUENCE, int(length_norm), clip,
        float(scale), C.ptr(loss), C.ptr(ratio), C.ptr(clipped), C.ptr(is_clipped), C.ptr(g), C.ptr(seg_loss), None,
        C.ptr(seg_start), C.ptr(n_seg), st), "prl_policy_tail")
    C.check(lib.prl_logprob_bwd_selected(
        C.ptr(logits), C.PRL_BF16, t, v, logits.stride(0), C.ptr(ids), C.ptr(lse), C.ptr(g), None, temp, 1, C.ptr(d),
        C.PRL_BF16, v, C.ptr(pending), st), "prl_logprob_bwd_selected")
    torch.cuda.synchronize()
    return lp, ent, d, loss, ratio, int(n_seg.item()), pending


@pytest.mark.parametrize("max_len,spins", [(224, 4096), (0, 4096), (224, 0), (16, 3)])
@pytest.mark.parametrize("skip", [False, True])
def test_sequence_rows_kernel_against_oracle(max_len, spins, skip):
    """One ratio per sample in one pass over the logits, against the CPU oracle: with the workgroups of a sample
    waiting for each other (224 / 4096), with every row left to the second pass (a longest sample of 0), with
    a wait of a single sweep (which rows then take the second pass depends on timing; the result must not), and
    with both limits small."""
    lengths = [40, 3, 64, 2, 100, 31, 16]
    vocab, temp = 32768, 0.8
    logits, ids, pos, mask, adv, old = seeded_case(11, lengths, 3, vocab, temp, mask_p=0.6)
    t = ids.numel()
    cfg = O.LossConfig("sequence", True, 1.02)
    b = lambda x: x.unsqueeze(0)
    want = O.policy_epilogue(logits, b(ids), b(pos), b(adv), b(old), b(mask), temp, cfg, len(lengths))
    lp, ent, d, loss, ratio, n, pending = _seq_call(
        logits[0].to(DEV), ids.to(DEV), pos.to(DEV), adv.to(DEV), old.to(DEV), mask.to(DEV), temp, True, 1.02,
        len(lengths), skip, max_len, spins)
    assert n == len(want["lengths"])
    assert bool((pending <= 1).all())  # written for every row
    if max_len == 0:  # every row with a gradient to write was left to the second pass
        scored = mask.clone()
        scored[0] = False
        assert pending.cpu()[:-1].bool().tolist() == scored[1:].tolist()
    sel = mask if skip else torch.ones_like(mask)
    close(lp.cpu()[sel], want["logprobs"][0][sel])
    close(ent.cpu()[sel], want["entropy"][0][sel])
    if skip:
        assert not lp.cpu()[~mask].any() and not ent.cpu()[~mask].any()
    close(loss[0], want["loss"])
    close(ratio[:n], want["importance_ratio"])
    assert bf16_ulp_close(d.cpu(), want["dlogits"][0], ulps=1, atol=1e-9)
    assert int((want["is_clipped"] > 0).sum()) > 0 and int((want["is_clipped"] == 0).sum()) > 0


def test_sequence_rows_kernel_full_size_through_policy_loss(monkeypatch):
    """The benchmark shape (2048 tokens, 151936 entries) with ragged samples, two of them (256 and 1024 tokens) too
    long to wait for: `policy_loss` in sequence mode against the oracle, and the rows that took the second pass
    are exactly the scored rows of the long samples."""
    from prime_rl_cse_amd import loss as L

    t, v = 2048, 151936
    gen = torch.Generator().manual_seed(35)
    lengths = [100, 156, 128, 128, 256, 64, 64, 128, 1024]
    ids, pos, mask = make_packed_row(gen, lengths, 0, v, mask_p=0.5)
    logits = torch.empty(1, t, v, dtype=torch.bfloat16)
    for r in range(0, t, 256):
        logits[0, r : r + 256] = (torch.randn(256, v, generator=gen) * 2.0).to(torch.bfloat16)
    adv = torch.randn(t, generator=gen)
    b = lambda x: x.unsqueeze(0)
    with torch.no_grad():
        lp0 = O.selective_log_softmax(O.shift_logits(logits.float()), b(ids))[0]
    old = lp0 + 0.01 * torch.randn(t, generator=gen)
    cfg = prl.LossConfig(ratio_type="sequence", ratio_length_norm=True, clip_ratio=1.05)
    want = O.policy_epilogue(logits, b(ids), b(pos), b(adv), b(old), b(mask), 1.0, cfg, len(lengths))
    dl = logits.to(DEV).requires_grad_(True)
    loss, aux = prl.policy_loss(dl, b(ids).to(DEV), b(pos).to(DEV), b(adv).to(DEV), b(old).to(DEV), b(mask).to(DEV),
                                1.0, cfg, len(lengths), skip_masked=True)
    pending = aux["seq_pending"].cpu().bool()
    loss.backward()
    n = len(lengths)
    assert int(aux["n_seg"].item()) == n
    close(loss, want["loss"])
    close(aux["importance_ratio"][:n], want["importance_ratio"])
    close(aux["is_clipped"][:n], want["is_clipped"])
    assert bf16_ulp_close(dl.grad.cpu(), want["dlogits"], ulps=1, atol=1e-9)
    # token u is written by row u - 1; the samples of 256 and 1024 tokens are longer than the 224 a workgroup waits for
    starts = [0]
    for n_tok in lengths:
        starts.append(starts[-1] + n_tok)
    long_rows = torch.zeros(t, dtype=torch.bool)
    for a, z in zip(starts, starts[1:]):
        if z - a > 224:
            lo = max(a, 1)
            long_rows[lo - 1 : z - 1] = mask[lo:z]
    assert pending[long_rows].all()
    # (a short sample's row may take the second pass too if its wait ran out; on an idle GPU none does)
    assert int(pending[~long_rows].sum()) == 0


def test_sequence_rows_kernel_beside_other_work_on_the_gpu():
    """The same call while another stream keeps the compute units busy with matrix products: workgroups of a
    sample may then not all be resident together, waits may run out and rows go to the second pass; whichever
    rows do, the results are the oracle's."""
    lengths = [120, 60, 200, 33, 90, 150, 10, 180]
    vocab, temp = 32768, 1.0
    logits, ids, pos, mask, adv, old = seeded_case(23, lengths, 0, vocab, temp, mask_p=0.8)
    cfg = O.LossConfig("sequence", False, 8.0)
    b = lambda x: x.unsqueeze(0)
    want = O.policy_epilogue(logits, b(ids), b(pos), b(adv), b(old), b(mask), temp, cfg, len(lengths))
    args = (logits[0].to(DEV), ids.to(DEV), pos.to(DEV), adv.to(DEV), old.to(DEV), mask.to(DEV), temp, False, 8.0,
            len(lengths), True)
    side = torch.cuda.Stream()
    a = torch.randn(4096, 4096, device=DEV, dtype=torch.bfloat16)
    torch.cuda.synchronize()
    seen = []
    for spins in (4096, 64, 2):
        with torch.cuda.stream(side):
            for _ in range(40):
                a @ a
        lp, ent, d, loss, ratio, n, pending = _seq_call(*args, 224, spins)
        side.synchronize()
        seen.append(int(pending.sum()))
        close(lp.cpu()[mask], want["logprobs"][0][mask])
        close(loss[0], want["loss"])
        close(ratio[:n], want["importance_ratio"])
        assert bf16_ulp_close(d.cpu(), want["dlogits"][0], ulps=1, atol=1e-9)
    print("rows left to the second pass beside a busy stream, by wait length:", seen)


def test_sequence_mode_with_a_long_sample_announced_takes_the_two_passes():
    """`max_sample_len` beyond what workgroups wait for: `policy_loss` does not try the one-pass kernel (whose rows
    would all be left to the second pass); results are the oracle's either way."""
    from prime_rl_cse_amd import loss as L

    lengths = [300, 40, 72]
    vocab = 32768
    logits, ids, pos, mask, adv, old = seeded_case(29, lengths, 0, vocab, 1.0, mask_p=0.8)
    cfg = prl.LossConfig(ratio_type="sequence", ratio_length_norm=True, clip_ratio=1.2)
    b = lambda x: x.unsqueeze(0)
    want = O.policy_epilogue(logits, b(ids), b(pos), b(adv), b(old), b(mask), 1.0, cfg, len(lengths))
    for hint, one_pass in ((300, False), (None, True), (200, True)):
        dl = logits.to(DEV).requires_grad_(True)
        loss, aux = prl.policy_loss(dl, b(ids).to(DEV), b(pos).to(DEV), b(adv).to(DEV), b(old).to(DEV),
                                    b(mask).to(DEV), 1.0, cfg, len(lengths), max_sample_len=hint)
        loss.backward()
        assert ("seq_pending" in aux) == one_pass
        close(loss, want["loss"])
        close(aux["importance_ratio"][: len(lengths)], want["importance_ratio"])
        assert bf16_ulp_close(dl.grad.cpu(), want["dlogits"], ulps=1, atol=1e-9)


def test_kernel_stamped_timing_of_the_row_kernel():
    """`prl_policy_rows_time_next`: the next launch stamps two events with the kernel's own start and end; results
    are those of an ordinary launch, the request is spent by one launch, and the time is a plausible kernel time
    (below what events recorded around the call see)."""
    lib = C.load()
    logits, ids, pos, mask, adv, old = seeded_case(3, [64] * 8, 0, 32768, 1.0, mask_p=1.0)
    t = ids.numel()
    x = logits[0].to(DEV)
    args = (x, ids.to(DEV), adv.to(DEV), old.to(DEV), mask.to(torch.uint8).to(DEV), 1.0, 8.0, t, False)
    plain = rows_call(*args)
    timer = C.KernelTimer(kernel_stamps=2, device=torch.device(DEV))
    (a, b), (c, d) = timer._stamps
    C.check(lib.prl_policy_rows_time_next(C.c_void_p(a.cuda_event), C.c_void_p(b.cuda_event)), "time_next")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    stamped = rows_call(*args)
    e1.record()
    again = rows_call(*args)  # no request pending any more: an ordinary launch
    torch.cuda.synchronize()
    for u, v, w in zip(plain, stamped, again):
        assert torch.equal(u, v) and torch.equal(u, w)
    kernel_ms, around_ms = a.elapsed_time(b), e0.elapsed_time(e1)
    assert 0.0 < kernel_ms < around_ms, (kernel_ms, around_ms)
    with pytest.raises(RuntimeError, match="two events"):
        C.check(lib.prl_policy_rows_time_next(C.c_void_p(c.cuda_event), None), "time_next")
