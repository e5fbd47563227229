"""Host logic of the side-by-side attention layout, and its arithmetic on CPU tensors against a masked
attention over the whole row."""

import torch
import torch.nn.functional as F

from prime_rl_cse_amd import attention


def block_causal_mask(lengths):
    t = sum(lengths)
    seg = torch.repeat_interleave(torch.arange(len(lengths)), torch.tensor(lengths))
    i = torch.arange(t)
    return (seg[:, None] == seg[None, :]) & (i[:, None] >= i[None, :])


def test_sample_starts_follow_position_restarts():
    pos = torch.tensor([[0, 1, 2, 0, 1, 0, 0, 1, 2, 3]])
    assert attention.sample_starts(pos).tolist() == [0, 3, 5, 6]
    assert attention.sample_starts(torch.zeros(1, 0, dtype=torch.long)).tolist() == []
    assert attention.sample_starts(torch.arange(7)[None]).tolist() == [0]


def test_plan_indices_of_a_ragged_row():
    lengths = [3, 1, 4, 2, 4, 3, 4, 4]
    pos = torch.cat([torch.arange(n) for n in lengths])[None]
    plan = attention.plan_from_position_ids(pos, "cpu")
    assert plan is not None and not plan.equal and (plan.n, plan.lmax, plan.total) == (8, 4, 25)
    g = plan.gather.view(8, 4)
    assert g[0].tolist() == [0, 1, 2, 2]  # padding repeats the sample's last token
    assert g[1].tolist() == [3, 3, 3, 3]
    assert g[2].tolist() == [4, 5, 6, 7]
    # scatter undoes gather on the real tokens
    assert torch.equal(plan.gather[plan.scatter], torch.arange(25))


def run_case(lengths, hq, hk, d=16):
    t = sum(lengths)
    gen = torch.Generator().manual_seed(t)
    q = torch.randn(1, t, hq, d, generator=gen).transpose(1, 2)
    k = torch.randn(1, t, hk, d, generator=gen).transpose(1, 2)
    v = torch.randn(1, t, hk, d, generator=gen).transpose(1, 2)
    pos = torch.cat([torch.arange(n) for n in lengths])[None]
    plan = attention.plan_from_position_ids(pos, "cpu", max_work_ratio=10.0)
    rep = hq // hk
    want = F.scaled_dot_product_attention(q, k.repeat_interleave(rep, 1), v.repeat_interleave(rep, 1),
                                          attn_mask=block_causal_mask(lengths)[None, None])
    want = want.transpose(1, 2)  # [1, t, hq, d], as the attention interface returns it
    if not attention._HF_SDPA:
        attention._HF_SDPA.append(lambda *a, **k: (_ for _ in ()).throw(AssertionError("fallback taken")))
    with attention.use_plan(plan):
        got, _ = attention.packed_sdpa_attention(None, q, k, v, None, scaling=None)
    assert got.shape == want.shape
    assert torch.allclose(got, want, rtol=1e-5, atol=1e-5)


def test_side_by_side_equals_masked_attention_equal_lengths():
    run_case([8] * 5, 4, 4)
    run_case([8] * 5, 4, 2)  # grouped-query heads


def test_side_by_side_equals_masked_attention_ragged():
    run_case([5, 1, 9, 3, 7, 2], 4, 2)
    run_case([2, 2, 6], 2, 2)
