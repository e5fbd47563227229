"""The short-sample attention kernels (prl_attn_short_fwd / _bwd) against a float32 causal attention per
sample, and inside a model against the masked attention over the whole row."""

import pytest
import torch
import torch.nn.functional as F

from prime_rl_cse_amd import _capi as C
from prime_rl_cse_amd import attention

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def reference(q, k, v, lengths, scale):
    """float32, sample by sample; q [1, Hq, T, D], k / v [1, Hkv, T, D] -> out [1, T, Hq, D]."""
    rep = q.shape[1] // k.shape[1]
    outs, a = [], 0
    for n in lengths:
        qs, ks, vs = (x[:, :, a : a + n].float() for x in (q, k, v))
        o = F.scaled_dot_product_attention(qs, ks.repeat_interleave(rep, 1), vs.repeat_interleave(rep, 1),
                                           is_causal=True, scale=scale)
        outs.append(o)
        a += n
    return torch.cat(outs, dim=2).transpose(1, 2)


def make(lengths, hq, hkv, seed, layout):
    t, d = sum(lengths), 128
    gen = torch.Generator(device=DEV).manual_seed(seed)

    def one(h):
        if layout == "projection":  # what the attention module hands over: [1, T, H, D] viewed head-major
            x = torch.randn(1, t, h, d, generator=gen, device=DEV).to(torch.bfloat16).transpose(1, 2)
        elif layout == "fused":  # a slice of a wider row, as after one q/k/v projection
            x = torch.randn(1, t, h * d + 256, generator=gen, device=DEV).to(torch.bfloat16)[..., 128 : 128 + h * d]
            x = x.unflatten(2, (h, d)).transpose(1, 2)
        else:  # head-major memory
            x = torch.randn(1, h, t, d, generator=gen, device=DEV).to(torch.bfloat16)
        return x.detach().requires_grad_(True)

    pos = torch.cat([torch.arange(n) for n in lengths])[None]
    plan = attention.plan_from_position_ids(pos, DEV)
    return one(hq), one(hkv), one(hkv), plan


CASES = [
    ([128] * 16, 16, 8, "projection"),  # the reverse_text row
    ([1, 31, 32, 33, 128, 100, 64, 2, 97], 16, 8, "projection"),
    ([5, 128, 17], 4, 4, "head_major"),
    ([40, 90, 3, 77], 8, 2, "fused"),
    ([128], 2, 1, "projection"),
    ([7], 4, 2, "head_major"),
    # 128 (sample, key/value head) groups or more: the backward is the one kernel per group
    ([1, 31, 32, 33, 128, 100, 64, 2, 97, 65, 127, 96, 5, 128, 77, 40, 12, 63], 16, 8, "fused"),
    ([128, 3] * 16 + [50], 8, 4, "projection"),
    ([17, 128, 64, 99] * 8, 16, 4, "head_major"),  # four query heads per key/value head
    ([128, 1, 45, 80] * 4, 8, 8, "projection"),  # one
]


@pytest.mark.parametrize("lengths,hq,hkv,layout", CASES)
def test_short_attention_forward_and_backward(lengths, hq, hkv, layout):
    q, k, v, plan = make(lengths, hq, hkv, seed=sum(lengths) + hq, layout=layout)
    assert plan is not None and plan.cu is not None and plan.cu.tolist()[-1] == sum(lengths)
    assert attention.short_attention_supported(q, k, plan)
    scale = 128 ** -0.5
    out = attention.short_attention(q, k, v, plan, scale)
    assert out.shape == (1, sum(lengths), hq, 128) and out.is_contiguous()
    gen = torch.Generator(device=DEV).manual_seed(1)
    dout = torch.randn(out.shape, generator=gen, device=DEV).to(torch.bfloat16)
    out.backward(dout)
    got = [out.float(), q.grad.float(), k.grad.float(), v.grad.float()]
    for x in (q, k, v):
        x.grad = None

    want_out = reference(q, k, v, lengths, scale)
    want_out.backward(dout.float())
    want = [want_out, q.grad.float(), k.grad.float(), v.grad.float()]
    for x in (q, k, v):
        x.grad = None

    # what bf16 storage of the same quantities costs: PyTorch's own bf16 attention, sample by sample
    a, errs = 0, [0.0, 0.0, 0.0, 0.0]
    rep = hq // hkv
    sd_out = []
    for n in lengths:
        o = F.scaled_dot_product_attention(q[:, :, a : a + n], k[:, :, a : a + n].repeat_interleave(rep, 1),
                                           v[:, :, a : a + n].repeat_interleave(rep, 1), is_causal=True, scale=scale)
        sd_out.append(o)
        a += n
    sd = torch.cat(sd_out, dim=2).transpose(1, 2)
    sd.backward(dout)
    base = [sd.float(), q.grad.float(), k.grad.float(), v.grad.float()]
    for name, g, w, b in zip(("out", "dq", "dk", "dv"), got, want, base):
        err = (g - w).norm().item() / (w.norm().item() + 1e-12)
        base_err = (b - w).norm().item() / (w.norm().item() + 1e-12)
        assert err <= max(2.0 * base_err, 4e-3), (name, err, base_err)
        assert (g - w).abs().max().item() <= 0.05 * w.abs().max().item() + 1e-3, name


def test_short_attention_rejects_what_it_cannot_take():
    lib = C.load()
    assert lib.prl_attn_short_supported(C.PRL_BF16 if hasattr(C, "PRL_BF16") else 1, 128, 128, 16, 8)
    for args in ((1, 64, 128, 16, 8), (1, 128, 129, 16, 8), (0, 128, 128, 16, 8), (1, 128, 128, 16, 3)):
        assert not lib.prl_attn_short_supported(*args)
    q = torch.zeros(1, 16, 133, 128, dtype=torch.bfloat16, device=DEV)
    k = torch.zeros(1, 8, 133, 128, dtype=torch.bfloat16, device=DEV)
    pos = torch.cat([torch.arange(n) for n in (129, 4)])[None]
    plan = attention.plan_from_position_ids(pos, DEV)  # a sample of 129 tokens: not for the short-sample kernels
    assert not attention.short_attention_supported(q, k, plan) and not plan.side_by_side
    assert attention.varlen_attention_supported(q, plan, 128 ** -0.5)
    assert not attention.varlen_attention_supported(q, plan, 0.1)  # another score scale than 1 / sqrt(d)
    assert not attention.varlen_attention_supported(q.float(), plan, None)
    pos = torch.cat([torch.arange(n) for n in (129, 100, 120, 90)])[None]
    plan = attention.plan_from_position_ids(pos, DEV)
    assert plan.side_by_side and plan.cu.tolist() == [0, 129, 229, 349, 439]


@pytest.mark.parametrize("lengths", [[128] * 4, [50, 7, 31, 64, 2, 40, 60, 33, 48, 21, 128], [90]])
def test_model_with_short_attention_matches_the_masked_row(lengths, monkeypatch):
    from prime_rl_cse_amd import trainer as T

    cfg = T.model_config("Qwen/Qwen3-0.6B", num_layers=2)
    model = T.get_model(cfg, DEV, torch.bfloat16, seed=2)
    t = sum(lengths)
    gen = torch.Generator().manual_seed(t)
    ids = torch.randint(0, 1000, (1, t), generator=gen)
    pos = torch.cat([torch.arange(n) for n in lengths])[None]
    plan = attention.plan_from_position_ids(pos, DEV)
    assert plan is not None and plan.cu is not None
    want = T.forward(model, ids.to(DEV), pos.to(DEV))  # transformers' SDPA under the block-diagonal mask
    want.float().square().mean().backward()
    grads = {n: p.grad.clone() for n, p in model.named_parameters()}
    model.zero_grad(set_to_none=True)
    calls = []
    real = attention.short_attention
    monkeypatch.setattr(attention, "short_attention", lambda *a, **k: (calls.append(1), real(*a, **k))[1])
    got = T.forward(model, ids.to(DEV), pos.to(DEV), plan)
    got.float().square().mean().backward()
    assert len(calls) == 2  # once per layer: the kernel ran, not a fallback
    scale = want.float().abs().max().item()
    assert (got.float() - want.float()).abs().max().item() <= 0.03 * scale
    for n, p in model.named_parameters():
        a, b = p.grad.float(), grads[n].float()
        assert (a - b).norm().item() <= 0.03 * b.norm().item() + 1e-6, n


def test_forward_without_a_mask_is_the_forward_with_one(monkeypatch):
    """Rows whose samples are laid side by side never read transformers' [T, T] mask, so `trainer.forward` tells
    the model not to build it; the logits are bit for bit those of the same kernels with the mask built.  A row
    of one sample keeps its mask (the masked-row path may read it), and so does a model without the mapping."""
    from prime_rl_cse_amd import trainer as T

    cfg = T.model_config("Qwen/Qwen3-0.6B", num_layers=2)
    model = T.get_model(cfg, DEV, torch.bfloat16, seed=2)
    lengths = [50, 7, 31, 64, 2, 40, 60, 33, 48, 21, 128]
    t = sum(lengths)
    ids = torch.randint(0, 1000, (1, t), generator=torch.Generator().manual_seed(1)).to(DEV)
    pos = torch.cat([torch.arange(n) for n in lengths])[None]
    plan = attention.plan_from_position_ids(pos, DEV)
    attention.install()
    assert not plan.side_by_side and attention.maskless(plan, model)  # bf16: varlen_attn takes any row
    wide = attention.plan_from_position_ids(torch.cat([torch.arange(16)] * 8)[None], DEV)
    single = attention.plan_from_position_ids(torch.arange(90)[None], DEV)
    fp32 = T.get_model(T.model_config("Qwen/Qwen3-0.6B", num_layers=1, hidden_size=64, intermediate_size=128,
                                      num_attention_heads=4, num_key_value_heads=2, head_dim=16, vocab_size=512),
                       DEV, torch.float32, seed=1)
    assert wide.side_by_side and attention.maskless(wide, fp32)
    assert not attention.maskless(single, fp32) and not attention.maskless(plan, fp32)
    assert not attention.maskless(None, model)
    built = []
    import transformers.models.qwen3.modeling_qwen3 as Q

    real = Q.create_causal_mask
    monkeypatch.setattr(Q, "create_causal_mask", lambda **kw: (built.append(1), real(**kw))[1])
    with torch.no_grad():
        got = T.forward(model, ids, pos.to(DEV), plan)
        assert not built
        with attention.use_plan(plan):
            want = model(input_ids=ids, position_ids=pos.to(DEV)).logits
        assert built == [1]
    assert torch.equal(got, want)
    # a layer that needs the mask after all says so instead of attending across samples
    q = torch.randn(1, 16, t, 128, device=DEV, dtype=torch.bfloat16)
    with attention.use_plan(plan), pytest.raises(RuntimeError, match="mask"):
        attention.packed_sdpa_attention(None, q, q[:, :8], q[:, :8], None, dropout=0.1)
