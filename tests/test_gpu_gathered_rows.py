"""Gathered rows (prl_policy_rows_gathered_fwd_bwd): the LM head run only at the positions whose next token is
scored.  Held bit for bit to the whole-row call with skip_masked=True, which the other tests hold to the oracle:
same per-token values, same loss and logged tensors, and the gradient rows of the kept positions; every row left
out has a zero gradient in the whole-row call."""

import pytest
import torch

import prime_rl_cse_amd as prl
from prime_rl_cse_amd import _capi as C
from tests.test_gpu_rows import seeded_case

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def both_ways(logits, ids, pos, mask, adv, old, temp, cfg, scale, multiple=1):
    dev = torch.device(DEV)
    t = ids.numel()
    on = lambda x: x.to(dev)
    full = on(logits).requires_grad_(True)
    args = (on(ids).view(1, t), on(pos).view(1, t), on(adv).view(1, t), on(old).view(1, t), on(mask).view(1, t))
    loss_f, aux_f = prl.policy_loss(full, *args, temp, cfg, scale, skip_masked=True)
    loss_f.backward()
    keep = prl.scored_positions(mask, multiple=multiple)
    gathered = on(logits)[:, on(keep), :].clone().requires_grad_(True)
    loss_g, aux_g = prl.policy_loss(gathered, *args, temp, cfg, scale, positions=on(keep))
    loss_g.backward()
    torch.cuda.synchronize()
    return keep, (loss_f, aux_f, full.grad), (loss_g, aux_g, gathered.grad)


@pytest.mark.parametrize(
    "vocab,temp,mask_p,multiple",
    [(1000, 0.7, 0.5, 1), (32000, 1.0, 0.3, 1), (151936, 1.0, 0.5, 1), (152064, 0.6, 0.6, 16), (50264, 1.3, 0.05, 64)],
)
def test_gathered_rows_equal_the_whole_row_with_masked_tokens_skipped(vocab, temp, mask_p, multiple):
    logits, ids, pos, mask, adv, old = seeded_case(100 + vocab % 97, [9, 1, 30, 17, 5], 3, vocab, temp, mask_p=mask_p)
    cfg = prl.LossConfig(clip_ratio=1.15)  # low enough that some tokens are clipped
    scale = int(mask.sum())
    keep, (loss_f, aux_f, grad_f), (loss_g, aux_g, grad_g) = both_ways(logits, ids, pos, mask, adv, old, temp, cfg,
                                                                      scale, multiple)
    t = ids.numel()
    assert int(mask[1:].sum()) <= keep.numel() <= t - 1
    if multiple > 1 and int(mask[1:].sum()) > 0:
        assert keep.numel() % multiple == 0 or keep.numel() == t - 1
    assert torch.equal(loss_f.cpu(), loss_g.cpu())
    for k in ("logprobs", "entropy", "importance_ratio", "clipped_importance_ratio", "is_clipped", "n_seg"):
        assert torch.equal(aux_f[k].cpu(), aux_g[k].cpu()), k
    n = int(aux_f["n_seg"][0]) + 1  # (the table is filled up to the last sample's end only)
    assert torch.equal(aux_f["seg_start"][:n].cpu(), aux_g["seg_start"][:n].cpu())
    assert grad_g.shape == (1, keep.numel(), vocab)
    assert torch.equal(grad_f[0, keep.to(DEV)].cpu(), grad_g[0].cpu())
    left_out = torch.ones(t, dtype=torch.bool)
    left_out[keep] = False
    assert not bool(grad_f[0, left_out.to(DEV)].any())  # what was not gathered had nothing to learn from
    assert bool(grad_g.any())


def test_no_scored_token_and_only_the_first_token_scored():
    logits, ids, pos, mask, adv, old = seeded_case(5, [6, 10], 0, 1000, 1.0)
    cfg = prl.LossConfig()
    for first in (False, True):
        mask = torch.zeros_like(mask)
        mask[0] = first
        keep, (loss_f, aux_f, grad_f), (loss_g, aux_g, grad_g) = both_ways(logits, ids, pos, mask, adv, old, 1.0, cfg, 1)
        assert keep.numel() == 0 and grad_g.shape == (1, 0, 1000)
        assert torch.equal(loss_f.cpu(), loss_g.cpu())
        for k in ("logprobs", "entropy", "importance_ratio", "clipped_importance_ratio", "is_clipped"):
            assert torch.equal(aux_f[k].cpu(), aux_g[k].cpu()), k
        assert not bool(grad_f.any())
        if first:  # token 0 is scored against an all-zero row: log-probability -ln V
            assert abs(float(aux_g["logprobs"][0, 0]) + torch.log(torch.tensor(1000.0)).item()) < 1e-6


def test_positions_outside_the_row_are_not_read_and_unordered_positions_work():
    dev = torch.device(DEV)
    lib = C.load()
    logits, ids, pos, mask, adv, old = seeded_case(11, [12, 20], 0, 1000, 1.0, mask_p=1.0)
    t, v = ids.numel(), 1000
    scale = float(mask.sum())
    x = logits[0].to(dev)
    ids_d, adv_d, old_d, mask_d = ids.to(dev), adv.to(dev), old.to(dev), mask.to(torch.uint8).to(dev)

    def call(positions):
        r = positions.numel()
        rows = torch.zeros(r, v, dtype=torch.bfloat16, device=dev)
        ok = (positions >= 0) & (positions < t - 1)
        rows[ok.to(dev)] = x[positions[ok].to(dev)]
        lp, ent, lse = (torch.full((t,), 7.0, device=dev) for _ in range(3))
        d = torch.full((r, v), 7.0, dtype=torch.bfloat16, device=dev)
        C.check(lib.prl_policy_rows_gathered_fwd_bwd(
            C.ptr(rows), C.PRL_BF16, r, v, v, C.ptr(positions.to(dev)), t, C.ptr(ids_d), C.ptr(adv_d), C.ptr(old_d),
            C.ptr(mask_d), 1.0, 8.0, scale, C.ptr(lp), C.ptr(ent), C.ptr(lse), C.ptr(d), C.PRL_BF16, v,
            C.stream_ptr(dev)), "prl_policy_rows_gathered_fwd_bwd")
        torch.cuda.synchronize()
        return lp.cpu(), ent.cpu(), lse.cpu(), d.cpu()

    ascending = torch.arange(0, t - 1, dtype=torch.int64)
    shuffled = ascending[torch.randperm(t - 1, generator=torch.Generator().manual_seed(3))]
    a, b = call(ascending), call(shuffled)
    for u, w in zip(a[:3], b[:3]):
        assert torch.equal(u, w)
    assert torch.equal(a[3][shuffled], b[3])
    assert bool(a[3].any())
    # positions outside [0, T - 2]: zero gradient rows, their would-be tokens stay unscored, the others are unchanged
    wild = torch.tensor([3, -1, t - 1, 5, 1 << 40, -(1 << 40)], dtype=torch.int64)
    lp, ent, lse, d = call(wild)
    assert torch.equal(d[[0, 3]], a[3][[3, 5]])
    assert not bool(d[[1, 2, 4, 5]].any())
    scored = torch.zeros(t, dtype=torch.bool)
    scored[[0, 4, 6]] = True  # token 0 by the call itself, tokens 4 and 6 by positions 3 and 5
    assert torch.equal(lp[scored], a[0][scored]) and not bool(lp[~scored].any())
    assert not bool(ent[~scored].any()) and not bool(lse[~scored].any())


def test_gathered_rows_refuse_what_the_kernel_cannot_do():
    dev = torch.device(DEV)
    logits, ids, pos, mask, adv, old = seeded_case(2, [8], 0, 1000, 1.0)
    t = ids.numel()
    on = lambda x: x.to(dev).view(1, t)
    keep = prl.scored_positions(mask).to(dev)
    rows = logits.to(dev)[:, keep, :]
    with pytest.raises(ValueError, match="gathered rows"):
        prl.policy_loss(rows.clone().requires_grad_(True), on(ids), on(pos), on(adv), on(old), on(mask), 1.0,
                        prl.LossConfig(ratio_type="sequence"), 1, positions=keep)
    with pytest.raises(ValueError, match="gathered rows"):
        prl.policy_loss(rows.float().requires_grad_(True), on(ids), on(pos), on(adv), on(old), on(mask), 1.0,
                        prl.LossConfig(), 1, positions=keep)
    with pytest.raises(ValueError, match="one position per row"):
        prl.policy_loss(rows.clone().requires_grad_(True), on(ids), on(pos), on(adv), on(old), on(mask), 1.0,
                        prl.LossConfig(), 1, positions=keep[:-1])


@pytest.mark.parametrize("vocab,temp,mask_p,multiple", [(1000, 0.7, 0.5, 1), (151936, 1.0, 0.4, 16), (152064, 0.6, 0.7, 1)])
def test_gathered_rows_that_need_no_gradient_are_only_scored(vocab, temp, mask_p, multiple):
    """Under `torch.no_grad()` (forward-only passes): the same per-token values, loss and logged tensors as the
    gathered call that also differentiates, bit for bit, and nothing to call backward on."""
    dev = torch.device(DEV)
    logits, ids, pos, mask, adv, old = seeded_case(7 + vocab % 89, [11, 2, 25, 14], 2, vocab, temp, mask_p=mask_p)
    cfg = prl.LossConfig(clip_ratio=1.15)
    t = ids.numel()
    on = lambda x: x.to(dev).view(1, t)
    args = (on(ids), on(pos), on(adv), on(old), on(mask))
    keep = prl.scored_positions(mask, multiple=multiple).to(dev)
    rows = logits.to(dev)[:, keep, :].clone()
    loss_g, aux_g = prl.policy_loss(rows.clone().requires_grad_(True), *args, temp, cfg, int(mask.sum()), positions=keep)
    with torch.no_grad():
        loss_n, aux_n = prl.policy_loss(rows, *args, temp, cfg, int(mask.sum()), positions=keep)
    # ... and without autograd switched off, for logits that simply need no gradient
    loss_p, aux_p = prl.policy_loss(rows, *args, temp, cfg, int(mask.sum()), positions=keep)
    torch.cuda.synchronize()
    assert not loss_n.requires_grad and not loss_p.requires_grad
    for loss, aux in ((loss_n, aux_n), (loss_p, aux_p)):
        assert torch.equal(loss.cpu(), loss_g.detach().cpu())
        for k in ("logprobs", "entropy", "importance_ratio", "clipped_importance_ratio", "is_clipped", "n_seg"):
            assert torch.equal(aux[k].cpu(), aux_g[k].cpu()), k
    assert bool(aux_n["logprobs"].any())


def test_score_only_call_through_the_c_abi_leaves_the_gradient_alone():
    """dlogits = NULL: logprob, entropy and lse as with a gradient buffer, whatever out_dtype / out_row_stride say."""
    dev = torch.device(DEV)
    lib = C.load()
    logits, ids, pos, mask, adv, old = seeded_case(21, [16, 9, 30], 0, 32000, 0.8, mask_p=0.6)
    t, v = ids.numel(), 32000
    keep = prl.scored_positions(mask, multiple=8).to(dev)
    r = keep.numel()
    rows = logits[0].to(dev)[keep].contiguous()
    ids_d, adv_d, old_d, mask_d = ids.to(dev), adv.to(dev), old.to(dev), mask.to(torch.uint8).to(dev)
    out = []
    for with_gradient in (True, False):
        lp, ent, lse = (torch.full((t,), 7.0, device=dev) for _ in range(3))
        d = torch.full((r, v), 7.0, dtype=torch.bfloat16, device=dev)
        C.check(lib.prl_policy_rows_gathered_fwd_bwd(
            C.ptr(rows), C.PRL_BF16, r, v, v, C.ptr(keep), t, C.ptr(ids_d), C.ptr(adv_d), C.ptr(old_d), C.ptr(mask_d),
            0.8, 8.0, float(mask.sum()), C.ptr(lp), C.ptr(ent), C.ptr(lse), C.ptr(d) if with_gradient else None,
            C.PRL_BF16 if with_gradient else -5, v if with_gradient else 3, C.stream_ptr(dev)),
            "prl_policy_rows_gathered_fwd_bwd")
        torch.cuda.synchronize()
        out.append((lp.cpu(), ent.cpu(), lse.cpu(), d.cpu()))
    for a, b in zip(out[0][:3], out[1][:3]):
        assert torch.equal(a, b)
    assert bool(out[0][0].any()) and not bool((out[0][3] == 7.0).any())
    assert bool((out[1][3] == 7.0).all())  # the buffer the second call never saw
