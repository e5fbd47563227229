"""The tiled attention kernels for samples of any length (prl_attn_long_fwd / _bwd) against a float32 causal
attention per sample."""

import pytest
import torch
import torch.nn.functional as F

from prime_rl_cse_amd import attention
from test_gpu_attention import DEV, make, reference

pytestmark = pytest.mark.gpu

LONG_CASES = [
    ([129], 2, 1, "projection"),
    ([300, 129, 1000], 4, 2, "projection"),
    ([128, 256, 384, 5], 4, 4, "head_major"),
    ([257, 1, 640], 8, 2, "fused"),
    ([2048, 100], 4, 2, "projection"),
    ([4096], 2, 1, "projection"),
    ([77, 130], 14, 2, "projection"),  # seven query heads per key/value head
]


@pytest.mark.parametrize("lengths,hq,hkv,layout", LONG_CASES)
def test_long_attention_forward_and_backward(lengths, hq, hkv, layout):
    q, k, v, plan = make(lengths, hq, hkv, seed=sum(lengths) + hq, layout=layout)
    assert plan is not None and plan.cu is not None and plan.lmax == max(lengths)
    assert attention.long_attention_supported(q, k, plan)
    scale = 128 ** -0.5
    out = attention.long_attention(q, k, v, plan, scale)
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
    a, rep, sd_out = 0, hq // hkv, []
    for n in lengths:
        o = F.scaled_dot_product_attention(q[:, :, a : a + n], k[:, :, a : a + n].repeat_interleave(rep, 1),
                                           v[:, :, a : a + n].repeat_interleave(rep, 1), is_causal=True, scale=scale)
        sd_out.append(o)
        a += n
    sd = torch.cat(sd_out, dim=2).transpose(1, 2)
    sd.backward(dout)
    base = [sd.float(), q.grad.float(), k.grad.float(), v.grad.float()]
    for name, g, w, b in zip(("out", "dq", "dk", "dv"), got, want, base):
        assert torch.isfinite(g).all(), name
        err = (g - w).norm().item() / (w.norm().item() + 1e-12)
        base_err = (b - w).norm().item() / (w.norm().item() + 1e-12)
        assert err <= max(2.0 * base_err, 4e-3), (name, err, base_err)
        assert (g - w).abs().max().item() <= 0.05 * w.abs().max().item() + 1e-3, name


def test_long_attention_is_what_a_model_forward_takes(monkeypatch):
    """Samples of more than 128 tokens go to the tiled kernels, not to torch's varlen_attn, unless switched off."""
    q, k, v, plan = make([200, 131], 4, 2, seed=5, layout="projection")
    assert not attention.short_attention_supported(q, k, plan)
    assert attention.long_attention_supported(q, k, plan)
    attention.install()
    with attention.use_plan(plan):
        out, _ = attention.packed_sdpa_attention(None, q, k, v, None, scaling=None)
    monkeypatch.setenv("PRL_ATTN_LONG", "0")
    assert not attention.long_attention_supported(q, k, plan)
    with attention.use_plan(plan):
        ref, _ = attention.packed_sdpa_attention(None, q, k, v, None, scaling=None)
    assert out.shape == ref.shape
    assert (out.float() - ref.float()).abs().max().item() <= 2e-2


def test_long_attention_refuses_what_it_cannot_take():
    from prime_rl_cse_amd import _capi as C

    lib = C.load()
    assert lib.prl_attn_long_supported(C.PRL_BF16, 128, 4096, 16, 8)
    for args in ((C.PRL_BF16, 64, 4096, 16, 8), (C.PRL_F32, 128, 4096, 16, 8), (C.PRL_BF16, 128, 4096, 16, 3),
                 (C.PRL_BF16, 128, 0, 16, 8)):
        assert not lib.prl_attn_long_supported(*args)
    q = torch.zeros(1, 4, 300, 128, dtype=torch.float32, device=DEV)
    pos = torch.arange(300)[None]
    plan = attention.plan_from_position_ids(pos, DEV)
    assert not attention.long_attention_supported(q, q, plan)  # fp32: left to torch
