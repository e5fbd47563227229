"""Fused transformer-block kernels against the torch modules they stand in for."""

import pytest
import torch

import prime_rl_cse_amd as prl  # noqa: F401
from prime_rl_cse_amd import fused
from tests.helpers import bf16_ulp_close

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def hf_rmsnorm(x, w, eps):
    """The forward of transformers' Qwen2/Qwen3 RMSNorm module, verbatim arithmetic."""
    dt = x.dtype
    h = x.to(torch.float32)
    var = h.pow(2).mean(-1, keepdim=True)
    h = h * torch.rsqrt(var + eps)
    return w * h.to(dt)


@pytest.mark.parametrize("shape", [(1, 2048, 1024), (1, 2048, 16, 128), (1, 37, 896), (3, 5, 3584), (2, 4096), (1, 1, 8)])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_rmsnorm_forward_and_backward(shape, dtype):
    gen = torch.Generator().manual_seed(sum(shape))
    x = (torch.randn(shape, generator=gen) * 2).to(dtype).to(DEV)
    w = (1 + 0.1 * torch.randn(shape[-1], generator=gen)).to(dtype).to(DEV)
    gy = torch.randn(shape, generator=gen).to(dtype).to(DEV)
    eps = 1e-6
    assert fused.rms_norm_supported(x, w)

    x1, w1 = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    y1 = fused.rms_norm(x1, w1, eps)
    y1.backward(gy)
    x0, w0 = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    y0 = hf_rmsnorm(x0, w0, eps)
    y0.backward(gy)
    # truth in float64 from the same (rounded) inputs
    xd, wd, gd = x.double().requires_grad_(True), w.double().requires_grad_(True), gy.double()
    yd = wd * (xd * torch.rsqrt(xd.pow(2).mean(-1, keepdim=True) + eps))
    yd.backward(gd)

    if dtype == torch.bfloat16:
        assert bf16_ulp_close(y1.detach().cpu(), y0.detach().cpu(), ulps=1, atol=1e-6)
        # input gradient: no further from the float64 value than torch's own bf16 chain, plus one rounding
        err1 = (x1.grad.double() - xd.grad).abs()
        err0 = (x0.grad.double() - xd.grad).abs()
        tol = xd.grad.abs() * 2.0**-7 + 1e-6
        assert bool((err1 <= torch.maximum(err0, tol)).all())
        # weight gradient: as a vector, no further from the float64 value than torch's own bf16 chain (+ 50 %)
        werr1 = (w1.grad.double() - wd.grad).norm().item()
        werr0 = (w0.grad.double() - wd.grad).norm().item()
        assert werr1 <= 1.5 * werr0 + 2.0**-8 * wd.grad.norm().item()
    else:
        assert torch.allclose(y1, y0, rtol=1e-6, atol=1e-6)
        assert torch.allclose(x1.grad.double(), xd.grad, rtol=1e-4, atol=1e-5)
        assert torch.allclose(w1.grad.double(), wd.grad, rtol=1e-4, atol=1e-3)
    # same inputs, same bits
    x2, w2 = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    fused.rms_norm(x2, w2, eps).backward(gy)
    assert torch.equal(x2.grad, x1.grad) and torch.equal(w2.grad, w1.grad)


@pytest.mark.parametrize("shape", [(1, 2048, 1024), (1, 16384, 1024), (1, 37, 896), (3, 5, 3584), (1, 300, 4096)])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_residual_addition_inside_the_norm_kernels_is_the_separate_addition(shape, dtype):
    """add_rms_norm = an addition kernel + rms_norm forward, rms_norm backward + an addition kernel: bit for bit,
    since the fused kernels round the sum and the input gradient to the storage type where the separate ones
    store them."""
    gen = torch.Generator().manual_seed(sum(shape) + 1)
    mk = lambda *sh: torch.randn(*sh, generator=gen).to(dtype).to(DEV)
    x, branch, gy, gs = mk(*shape), mk(*shape), mk(*shape), mk(*shape)
    w = (1 + 0.1 * torch.randn(shape[-1], generator=gen)).to(dtype).to(DEV)
    assert fused.add_rms_norm_supported(x, branch, w)

    x1, b1, w1 = (t.clone().requires_grad_(True) for t in (x, branch, w))
    s1, y1 = fused.add_rms_norm(x1, b1, w1, 1e-6)
    torch.autograd.backward([s1, y1], [gs, gy])
    x0, b0, w0 = (t.clone().requires_grad_(True) for t in (x, branch, w))
    s0 = x0 + b0
    y0 = fused.rms_norm(s0, w0, 1e-6)
    torch.autograd.backward([s0, y0], [gs, gy])
    assert torch.equal(s1, s0) and torch.equal(y1, y0)
    assert torch.equal(x1.grad, x0.grad)
    assert torch.equal(b1.grad, b0.grad)
    assert torch.equal(w1.grad, w0.grad)
    # one output unused: the gradient of the other alone
    x2, b2, w2 = (t.clone().requires_grad_(True) for t in (x, branch, w))
    fused.add_rms_norm(x2, b2, w2, 1e-6)[1].backward(gy)
    x3, w3 = (x + branch).requires_grad_(True), w.clone().requires_grad_(True)
    fused.rms_norm(x3, w3, 1e-6).backward(gy)
    assert torch.equal(x2.grad, x3.grad) and torch.equal(b2.grad, x3.grad) and torch.equal(w2.grad, w3.grad)
    x4, b4, w4 = (t.clone().requires_grad_(True) for t in (x, branch, w))
    fused.add_rms_norm(x4, b4, w4, 1e-6)[0].backward(gs)
    assert torch.equal(x4.grad, gs) and torch.equal(b4.grad, gs) and w4.grad is None


def test_layers_that_carry_their_last_addition_give_the_model_without():
    from prime_rl_cse_amd import trainer as T

    cfg = T.model_config("Qwen/Qwen3-0.6B", num_layers=3)
    model = T.get_model(cfg, DEV, torch.bfloat16, seed=0)
    ids = torch.randint(0, 1000, (1, 96), device=DEV)
    pos = torch.arange(32, device=DEV).repeat(3)[None]
    done = fused.fuse_model(model)
    assert done["layer"] == 3

    def run():
        model.zero_grad(set_to_none=True)
        out = T.forward(model, ids, pos)
        out.float().square().mean().backward()
        return out, {n: p.grad.clone() for n, p in model.named_parameters()}

    want, grads = run()  # every layer adds its own MLP output (the addition after the attention is in the norm)
    assert fused.enable_residual_carry(model) == 3
    got, carried = run()
    assert torch.equal(got, want)
    for n, g in grads.items():
        assert torch.equal(carried[n], g), n
    with torch.no_grad():
        assert torch.equal(T.forward(model, ids, pos), want)
    # a reader handed something else than the carrying layer's output fails instead of dropping the branch
    layer = model.model.layers[1]
    hidden = torch.randn(1, 96, cfg.hidden_size, device=DEV).to(torch.bfloat16)
    cos_sin = model.model.rotary_emb(hidden, pos)
    out = layer(hidden, position_embeddings=cos_sin, position_ids=pos)
    with pytest.raises(RuntimeError, match="residual carry"):
        model.model.norm(out.clone())
    assert fused.enable_residual_carry(model, False) == 0
    assert torch.equal(T.forward(model, ids, pos), want)
    fused.unfuse_model(model)
    assert fused.describe(model) == 0 and not hasattr(model.model.layers[0], "_prl_carry")


def test_rmsnorm_rows_with_a_stride_and_unsupported_shapes():
    gen = torch.Generator().manual_seed(1)
    wide = torch.randn(64, 1024 + 64, generator=gen).to(torch.bfloat16).to(DEV)
    x = wide[:, :1024]  # rows further apart than H elements
    w = torch.ones(1024, dtype=torch.bfloat16, device=DEV)
    assert fused.rms_norm_supported(x, w)
    assert bf16_ulp_close(fused.rms_norm(x, w, 1e-6).cpu(), hf_rmsnorm(x, w, 1e-6).cpu(), ulps=1, atol=1e-6)
    assert not fused.rms_norm_supported(torch.randn(4, 1001, device=DEV).to(torch.bfloat16),
                                        torch.ones(1001, dtype=torch.bfloat16, device=DEV))
    assert not fused.rms_norm_supported(x, w.float())
    assert not fused.rms_norm_supported(x.cpu(), w.cpu())


def test_fuse_model_rebinds_every_norm_and_keeps_the_numbers():
    from prime_rl_cse_amd import trainer as T

    cfg = T.model_config("Qwen/Qwen3-0.6B", num_layers=2)
    model = T.get_model(cfg, DEV, torch.bfloat16, seed=0)
    ids = torch.randint(0, 1000, (1, 96), device=DEV)
    pos = torch.arange(32, device=DEV).repeat(3)[None]
    want = T.forward(model, ids, pos)
    want.float().square().mean().backward()
    grads = {n: p.grad.clone() for n, p in model.named_parameters()}
    model.zero_grad(set_to_none=True)
    done = fused.fuse_model(model)
    assert done["rmsnorm"] == 2 * 4 + 1 and done["swiglu"] == 2  # input, post-attention, query, key norms per block + final
    got = T.forward(model, ids, pos)
    got.float().square().mean().backward()
    assert (got.float() - want.float()).abs().max().item() <= 0.05 * want.float().abs().max().item() + 1e-3
    for n, p in model.named_parameters():
        a, b = p.grad.float(), grads[n].float()
        assert (a - b).norm().item() <= 0.05 * b.norm().item() + 1e-6, n
    fused.unfuse_model(model)
    assert fused.describe(model) == 0
    assert torch.equal(T.forward(model, ids, pos), want)


def hf_rope(q, k, cos, sin):
    """transformers' apply_rotary_pos_emb with unsqueeze_dim = 1, verbatim arithmetic."""
    def rotate_half(x):
        return torch.cat((-x[..., x.shape[-1] // 2:], x[..., : x.shape[-1] // 2]), dim=-1)

    cos, sin = cos.unsqueeze(1), sin.unsqueeze(1)
    return (q * cos) + (rotate_half(q) * sin), (k * cos) + (rotate_half(k) * sin)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("b,t,hq,hk,d", [(1, 2048, 16, 8, 128), (2, 37, 14, 2, 64), (1, 5, 4, 4, 32)])
def test_rope_forward_and_backward(dtype, b, t, hq, hk, d):
    gen = torch.Generator().manual_seed(b * t + d)
    # the layout the attention module produces: projections viewed as [b, t, heads, d], then transposed
    q = torch.randn(b, t, hq, d, generator=gen).to(dtype).to(DEV).transpose(1, 2)
    k = torch.randn(b, t, hk, d, generator=gen).to(dtype).to(DEV).transpose(1, 2)
    ang = torch.rand(b, t, d // 2, generator=gen) * 6.28
    ang = torch.cat((ang, ang), dim=-1)
    cos, sin = ang.cos().to(dtype).to(DEV), ang.sin().to(dtype).to(DEV)
    gq = torch.randn(b, hq, t, d, generator=gen).to(dtype).to(DEV)
    gk = torch.randn(b, hk, t, d, generator=gen).to(dtype).to(DEV)
    assert fused.rope_supported(q, k, cos, sin)

    q1, k1 = q.detach().clone().requires_grad_(True), k.detach().clone().requires_grad_(True)
    # keep the transposed layout for the leaf as well
    q1 = q.detach().requires_grad_(True)
    k1 = k.detach().requires_grad_(True)
    qo1, ko1 = fused.apply_rotary_pos_emb(q1, k1, cos, sin)
    torch.autograd.backward([qo1, ko1], [gq, gk])
    qd, kd = q.double().requires_grad_(True), k.double().requires_grad_(True)
    qod, kod = hf_rope(qd, kd, cos.double(), sin.double())
    torch.autograd.backward([qod, kod], [gq.double(), gk.double()])
    assert qo1.shape == q.shape and qo1.stride() == q.stride()
    if dtype == torch.bfloat16:
        for got, want in ((qo1, qod), (ko1, kod), (q1.grad, qd.grad), (k1.grad, kd.grad)):
            assert bf16_ulp_close(got.detach().cpu(), want.detach().to(torch.bfloat16).cpu(), ulps=1, atol=1e-6)
        # and no worse than the torch chain it replaces
        q0, k0 = q.detach().requires_grad_(True), k.detach().requires_grad_(True)
        qo0, ko0 = hf_rope(q0, k0, cos, sin)
        assert (qo1.double() - qod).abs().max().item() <= (qo0.double() - qod).abs().max().item() + 1e-6
    else:
        for got, want in ((qo1, qod), (ko1, kod), (q1.grad, qd.grad), (k1.grad, kd.grad)):
            assert torch.allclose(got.double(), want, rtol=1e-5, atol=1e-6)


def test_fuse_model_patches_rope_and_whole_model_agrees():
    from prime_rl_cse_amd import trainer as T

    cfg = T.model_config("Qwen/Qwen3-0.6B", num_layers=2)
    model = T.get_model(cfg, DEV, torch.bfloat16, seed=1)
    ids = torch.randint(0, 1000, (1, 64), device=DEV)
    pos = torch.arange(32, device=DEV).repeat(2)[None]
    want = T.forward(model, ids, pos).float()
    done = fused.fuse_model(model)
    assert done["rope"] == 1
    timer = prl._capi.KernelTimer()
    prl._capi.set_timer(timer)
    try:
        got = T.forward(model, ids, pos).float()
    finally:
        prl._capi.set_timer(None)
    torch.cuda.synchronize()
    seen = timer.summary()
    assert seen["prl_rope"]["launches"] == 2
    # per block: input, query and key norms alone, the post-attention norm together with the addition before it
    assert seen["prl_rmsnorm_fwd"]["launches"] == 7 and seen["prl_add_rmsnorm_fwd"]["launches"] == 2
    assert (got - want).abs().max().item() <= 0.05 * want.abs().max().item() + 1e-3
    fused.unfuse_model(model)
    assert torch.equal(T.forward(model, ids, pos).float(), want)


@pytest.mark.parametrize("path", ["best", "side_by_side"])
@pytest.mark.parametrize("lengths", [[32] * 6, [50, 7, 31, 64, 2, 40, 60, 33, 48, 21], [200], [200, 150, 129, 140],
                                     [300, 20, 20]])
def test_side_by_side_attention_gives_the_masked_row_attention(lengths, path, monkeypatch):
    from prime_rl_cse_amd import attention
    from prime_rl_cse_amd import trainer as T

    cfg = T.model_config("Qwen/Qwen3-0.6B", num_layers=2)
    model = T.get_model(cfg, DEV, torch.bfloat16, seed=2)
    t = sum(lengths)
    gen = torch.Generator().manual_seed(t)
    ids = torch.randint(0, 1000, (1, t), generator=gen)
    pos = torch.cat([torch.arange(n) for n in lengths])[None]
    if path == "side_by_side":  # neither the short-sample kernels nor varlen: SDPA over the padded layout
        monkeypatch.setenv("PRL_ATTN_SHORT", "0")
        monkeypatch.setenv("PRL_ATTN_VARLEN", "0")
    plan = attention.plan_from_position_ids(pos, DEV)
    assert plan.n == len(lengths) and plan.lmax == max(lengths) and plan.cu.tolist()[-1] == t
    if plan.side_by_side:
        assert plan.equal == (len(set(lengths)) == 1)
    else:
        assert len(lengths) == 1 or lengths == [300, 20, 20]  # one sample (dominates): no padded layout
    want = T.forward(model, ids.to(DEV), pos.to(DEV))
    want.float().square().mean().backward()
    grads = {n: p.grad.clone() for n, p in model.named_parameters()}
    model.zero_grad(set_to_none=True)
    got = T.forward(model, ids.to(DEV), pos.to(DEV), plan)
    got.float().square().mean().backward()
    scale = want.float().abs().max().item()
    assert (got.float() - want.float()).abs().max().item() <= 0.03 * scale
    for n, p in model.named_parameters():
        a, b = p.grad.float(), grads[n].float()
        assert (a - b).norm().item() <= 0.03 * b.norm().item() + 1e-6, n
    # a different row afterwards, without a plan, still takes the masked path
    again = T.forward(model, ids.to(DEV), pos.to(DEV))
    assert torch.equal(again, want)


def test_plan_declines_when_padding_would_not_pay():
    from prime_rl_cse_amd import attention

    pos = torch.cat([torch.arange(900), torch.arange(50), torch.arange(50)])[None]
    plan = attention.plan_from_position_ids(pos, "cpu")  # one sample dominates: 3 * 900^2 > T^2 / 2
    assert not plan.side_by_side and plan.gather is None and plan.cu.tolist() == [0, 900, 950, 1000]
    pos = torch.cat([torch.arange(128)] * 16)[None]
    plan = attention.plan_from_position_ids(pos, "cpu")
    assert plan.equal and plan.n == 16 and plan.gather is None


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("shape", [(1, 2048, 3072), (3, 5, 8), (1, 37, 4864)])
def test_swiglu_forward_and_backward(dtype, shape):
    gen = torch.Generator().manual_seed(sum(shape))
    g = (torch.randn(shape, generator=gen) * 2).to(dtype).to(DEV)
    u = torch.randn(shape, generator=gen).to(dtype).to(DEV)
    d = torch.randn(shape, generator=gen).to(dtype).to(DEV)
    assert fused.swiglu_supported(g, u)
    g1, u1 = g.clone().requires_grad_(True), u.clone().requires_grad_(True)
    o1 = fused.swiglu(g1, u1)
    o1.backward(d)
    g0, u0 = g.clone().requires_grad_(True), u.clone().requires_grad_(True)
    o0 = torch.nn.functional.silu(g0) * u0
    o0.backward(d)
    gd, ud = g.double().requires_grad_(True), u.double().requires_grad_(True)
    od = torch.nn.functional.silu(gd) * ud
    od.backward(d.double())
    if dtype == torch.bfloat16:
        assert bf16_ulp_close(o1.detach().cpu(), o0.detach().cpu(), ulps=1, atol=1e-6)
        for got, eager, want in ((g1.grad, g0.grad, gd.grad), (u1.grad, u0.grad, ud.grad)):
            err1, err0 = (got.double() - want).abs(), (eager.double() - want).abs()
            assert bool((err1 <= torch.maximum(err0, want.abs() * 2.0**-6 + 1e-6)).all())
    else:
        assert torch.allclose(o1.double(), od, rtol=1e-5, atol=1e-6)
        assert torch.allclose(g1.grad.double(), gd.grad, rtol=1e-4, atol=1e-5)
        assert torch.allclose(u1.grad.double(), ud.grad, rtol=1e-5, atol=1e-6)
    assert not fused.swiglu_supported(g[..., :-1], u[..., :-1])  # not contiguous: torch's path


def test_fuse_model_rebinds_the_gated_mlps():
    from prime_rl_cse_amd import trainer as T

    cfg = T.model_config("Qwen/Qwen3-0.6B", num_layers=3)
    model = T.get_model(cfg, DEV, torch.bfloat16, seed=3)
    done = fused.fuse_model(model)
    assert done == {"rmsnorm": 13, "rope": 1, "swiglu": 3, "qkv": 3, "linear": 12, "layer": 3, "lm_head": 1}
    ids = torch.randint(0, 1000, (1, 64), device=DEV)
    pos = torch.arange(64, device=DEV)[None]
    timer = prl._capi.KernelTimer()
    prl._capi.set_timer(timer)
    try:
        T.forward(model, ids, pos).float().mean().backward()
    finally:
        prl._capi.set_timer(None)
    torch.cuda.synchronize()
    s = timer.summary()
    assert s["prl_swiglu_fwd"]["launches"] == 3 and s["prl_swiglu_bwd"]["launches"] == 3
    fused.unfuse_model(model)


def test_rmsnorm_over_the_heads_of_a_wider_row():
    """The query block of a stacked q/k/v projection: [token][head][128] inside rows of 4096."""
    gen = torch.Generator().manual_seed(4)
    for dtype in (torch.bfloat16, torch.float32):
        wide = torch.randn(1, 70, 4096, generator=gen).to(dtype).to(DEV)
        w = (1.0 + 0.1 * torch.randn(128, generator=gen)).to(dtype).to(DEV)
        x = wide[..., 1024:3072].view(1, 70, 16, 128)
        assert not x.is_contiguous() and fused.rms_norm_supported(x, w)
        x1 = x.detach().clone().requires_grad_(True)  # contiguous copy: the plain-rows path
        w1 = w.clone().requires_grad_(True)
        gy = torch.randn(x.shape, generator=gen).to(dtype).to(DEV)
        y1 = fused.rms_norm(x1, w1, 1e-6)
        y1.backward(gy)
        wide2 = wide.clone().requires_grad_(True)
        w2 = w.clone().requires_grad_(True)
        y2 = fused.rms_norm(wide2[..., 1024:3072].view(1, 70, 16, 128), w2, 1e-6)
        y2.backward(gy)
        assert torch.equal(y2, y1) and torch.equal(w2.grad, w1.grad)
        assert torch.equal(wide2.grad[..., 1024:3072].reshape(x.shape), x1.grad)
        assert not wide2.grad[..., :1024].any() and not wide2.grad[..., 3072:].any()


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_stacked_products_over_adjacent_weights_match_the_separate_ones(dtype):
    """With the parameters laid back to back (replica.FlatState) the q/k/v and gate/up products run stacked;
    logits and every gradient agree with the same model run product by product."""
    import copy

    from prime_rl_cse_amd import attention
    from prime_rl_cse_amd import trainer as T
    from prime_rl_cse_amd.replica import FlatState

    cfg = T.model_config("Qwen/Qwen3-0.6B", num_layers=2, vocab_size=4096)
    plain = T.get_model(cfg, DEV, dtype, seed=5)
    stacked = copy.deepcopy(plain)
    fused.fuse_model(stacked)
    state = FlatState(stacked, compute_dtype=dtype)
    layer = stacked.model.layers[0]
    assert fused.adjacent_rows(layer.self_attn.q_proj.weight, layer.self_attn.k_proj.weight,
                               layer.self_attn.v_proj.weight).shape == (4096, 1024)
    assert fused.adjacent_rows(layer.mlp.gate_proj.weight, layer.mlp.up_proj.weight).shape == (6144, 1024)
    assert fused.adjacent_rows(plain.model.layers[0].mlp.gate_proj.weight,
                               plain.model.layers[0].mlp.up_proj.weight) is None  # separate allocations
    lengths = [40, 24, 64]
    t = sum(lengths)
    gen = torch.Generator().manual_seed(6)
    ids = torch.randint(0, 4096, (1, t), generator=gen).to(DEV)
    pos = torch.cat([torch.arange(n) for n in lengths])[None]
    plan = attention.plan_from_position_ids(pos, DEV)
    calls = []
    real = fused._QKVNormRope.apply
    try:
        fused._QKVNormRope.apply = staticmethod(lambda *a: (calls.append(1), real(*a))[1])
        got = T.forward(stacked, ids, pos.to(DEV), plan)
    finally:
        fused._QKVNormRope.apply = real
    assert len(calls) == 2  # per layer: one stacked q/k/v product, then query / key norm + rotary in one kernel
    want = T.forward(plain, ids, pos.to(DEV), plan)
    got.float().square().mean().backward()
    want.float().square().mean().backward()
    tol = 1e-4 if dtype == torch.float32 else 3e-2
    scale = want.float().abs().max().item()
    assert (got.float() - want.float()).abs().max().item() <= tol * scale
    for (name, a), b in zip(plain.named_parameters(), state.params):
        assert b.grad is not None and b.grad.shape == a.grad.shape, name
        assert (b.grad.float() - a.grad.float()).norm().item() <= tol * a.grad.float().norm().item() + 1e-9, name
    fused.unfuse_model(stacked)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_qk_norm_rope_function_against_the_separate_steps(dtype):
    """`_QKVNormRope` against projection, per-head RMSNorm and rotary embedding spelled out in torch (float64 for
    the fp32 case), outputs and every gradient; a ragged head split (12 query heads, 4 key / value heads)."""
    from prime_rl_cse_amd.fused import _QKVNormRope

    t, hid, hq, hk, d = 77, 256, 12, 4, 128
    gen = torch.Generator().manual_seed(8)
    ref_dtype = torch.float64 if dtype == torch.float32 else torch.float32
    x = torch.randn(1, t, hid, generator=gen)
    flat = torch.randn((hq + 2 * hk) * d, hid, generator=gen) * hid ** -0.5
    nq, nk = 1.0 + 0.1 * torch.randn(d, generator=gen), 1.0 + 0.1 * torch.randn(d, generator=gen)
    ang = torch.rand(1, t, d // 2, generator=gen) * 6.28
    cos, sin = torch.cat([ang.cos(), ang.cos()], -1), torch.cat([ang.sin(), ang.sin()], -1)
    gq, gk, gv = (torch.randn(1, h, t, d, generator=gen) for h in (hq, hk, hk))

    def run(fused_path: bool):
        dt = dtype if fused_path else ref_dtype
        xs = x.to(DEV).to(dtype).to(dt).requires_grad_(True)
        w = flat.to(DEV).to(dtype).to(dt)  # one storage: the three weights are adjacent rows of it
        wq, wk, wv = (p.detach().requires_grad_(True) for p in w.split([hq * d, hk * d, hk * d]))
        a, b = nq.to(DEV).to(dtype).to(dt).requires_grad_(True), nk.to(DEV).to(dtype).to(dt).requires_grad_(True)
        c, s_ = cos.to(DEV).to(dtype).to(dt), sin.to(DEV).to(dtype).to(dt)
        if fused_path:
            stacked = fused.adjacent_rows(wq, wk, wv)
            assert stacked is not None and stacked.shape == w.shape
            q, k, v = _QKVNormRope.apply(xs, stacked, wq, wk, wv, a, b, c, s_, 1e-6)
        else:
            def norm(z, weight):
                return weight * (z * torch.rsqrt(z.pow(2).mean(-1, keepdim=True) + 1e-6))

            def rot(z):
                return torch.cat([-z[..., d // 2:], z[..., : d // 2]], -1)

            q = norm((xs @ wq.t()).view(1, t, hq, d), a).transpose(1, 2)
            k = norm((xs @ wk.t()).view(1, t, hk, d), b).transpose(1, 2)
            v = (xs @ wv.t()).view(1, t, hk, d).transpose(1, 2)
            q, k = q * c[:, None] + rot(q) * s_[:, None], k * c[:, None] + rot(k) * s_[:, None]
        loss = (q * gq.to(DEV).to(dt)).sum() + (k * gk.to(DEV).to(dt)).sum() + (v * gv.to(DEV).to(dt)).sum()
        loss.backward()
        return [z.detach().double() for z in (q, k, v, xs.grad, wq.grad, wk.grad, wv.grad, a.grad, b.grad)]

    got, want = run(True), run(False)
    tol = 2e-5 if dtype == torch.float32 else 2e-2
    for name, g, w in zip("q k v dx dwq dwk dwv dnq dnk".split(), got, want):
        assert g.shape == w.shape, name
        assert (g - w).norm().item() <= tol * w.norm().item() + 1e-9, (name, (g - w).norm().item(), w.norm().item())


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_qk_norm_rope_backward_in_place_on_gradients_that_are_one_matrix(dtype):
    """The attention kernels hand dq, dk, dv over as the blocks of one [T, (hq + 2 hk) d] matrix; `_QKVNormRope`
    then runs its kernel in place and copies nothing.  Same bits as with three separate gradient tensors."""
    from prime_rl_cse_amd import attention
    from prime_rl_cse_amd.fused import _QKVNormRope, _as_one_matrix

    t, hid, hq, hk, d = 200, 256, 8, 4, 128
    gen = torch.Generator(device=DEV).manual_seed(5)
    rnd = lambda *sh: torch.randn(*sh, generator=gen, device=DEV)
    x = rnd(1, t, hid).to(dtype)
    w = (rnd((hq + 2 * hk) * d, hid) * hid ** -0.5).to(dtype)
    nq, nk = (1.0 + 0.1 * rnd(d)).to(dtype), (1.0 + 0.1 * rnd(d)).to(dtype)
    ang = torch.rand(1, t, d // 2, generator=gen, device=DEV) * 6.28
    cos, sin = torch.cat([ang.cos(), ang.cos()], -1).to(dtype), torch.cat([ang.sin(), ang.sin()], -1).to(dtype)
    grads = [rnd(1, h, t, d).to(dtype) for h in (hq, hk, hk)]
    together = attention._stacked_grads(t, hq, hk, d, dtype, torch.device(DEV))
    for dst, src in zip(together, grads):
        dst.copy_(src)
    assert _as_one_matrix(*together, t, hq, hk, d, torch.empty(t, (hq + 2 * hk) * d, dtype=dtype, device=DEV)) is not None
    assert _as_one_matrix(*grads, t, hq, hk, d, torch.empty(t, (hq + 2 * hk) * d, dtype=dtype, device=DEV)) is None
    assert _as_one_matrix(together[0], together[2], together[1], t, hq, hk, d,
                          torch.empty(t, (hq + 2 * hk) * d, dtype=dtype, device=DEV)) is None

    def run(gs, hook=False, retain=False):
        xs = x.clone().requires_grad_(True)
        wq, wk, wv = (p.detach().requires_grad_(True) for p in w.split([hq * d, hk * d, hk * d]))
        a, b = nq.clone().requires_grad_(True), nk.clone().requires_grad_(True)
        out = fused.qkv_norm_rope(xs, fused.adjacent_rows(wq, wk, wv), wq, wk, wv, a, b, cos, sin, 1e-6)
        seen = []
        if hook:
            out[0].register_hook(lambda g: seen.append(g))  # a holder of the very tensor the backward is handed
        if retain:
            out[1].retain_grad()
        torch.autograd.backward(list(out), list(gs))
        return [z.grad for z in (xs, wq, wk, wv, a, b)], seen, out[1].grad if retain else None

    want, _, _ = run(grads)
    before = [g.clone() for g in together]
    got, _, _ = run(together)
    for a_, b_ in zip(got, want):
        assert torch.equal(a_, b_)
    assert not torch.equal(together[0], before[0])  # nobody was watching: the kernel worked in place
    # with a hook on an output the gradients are left as they arrived, and the hook's tensor keeps dq
    for dst, src in zip(together, grads):
        dst.copy_(src)
    got, seen, kept = run(together, hook=True, retain=True)
    for a_, b_ in zip(got, want):
        assert torch.equal(a_, b_)
    assert torch.equal(seen[0], grads[0]) and torch.equal(kept, grads[1])
    for g, src in zip(together, grads):
        assert torch.equal(g, src)
    # the bare function (no outputs named) never works in place
    out = _QKVNormRope.apply(x.clone().requires_grad_(True), w, *w.split([hq * d, hk * d, hk * d]), nq, nk, cos, sin,
                             1e-6)
    torch.autograd.backward(list(out), list(together))
    for g, src in zip(together, grads):
        assert torch.equal(g, src)


def test_weight_gradient_split_over_the_tokens(monkeypatch):
    """`weight_grad` at a row long enough to be split, against the float32 product; and the short row left alone."""
    monkeypatch.setenv("PRL_LAYER_GEMMS", "lib")  # (this is about the products that stay with the BLAS library)
    gen = torch.Generator(device=DEV).manual_seed(12)
    for rows, n_out, n_in in ((16384, 1024, 2048), (16384, 4096, 1024), (8192, 6144, 1024), (2048, 1024, 3072)):
        gy = torch.randn(rows, n_out, generator=gen, device=DEV).to(torch.bfloat16)
        x = torch.randn(rows, n_in, generator=gen, device=DEV).to(torch.bfloat16)
        want = gy.float().t() @ x.float()
        got = fused.weight_grad(gy, x)
        one = gy.t() @ x
        assert got.shape == want.shape and got.dtype == torch.bfloat16
        err = (got.float() - want).norm().item() / want.norm().item()
        base = (one.float() - want).norm().item() / want.norm().item()
        assert err <= 1.6 * base + 1e-4, (rows, n_out, n_in, err, base)
        if rows < 8192:
            assert torch.equal(got, one)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("nparts,n,stride", [(4, 1024 * 2048, 1024 * 2048), (2, 6144 * 1024, 6144 * 1024), (3, 40, 48),
                                             (1, 8, 8), (5, 0, 0)])
def test_partial_results_added_in_order_and_rounded_once(dtype, nparts, n, stride):
    from prime_rl_cse_amd import _capi as C

    gen = torch.Generator(device=DEV).manual_seed(nparts + n)
    parts = torch.randn(nparts, max(stride, 8), generator=gen, device=DEV).to(dtype)
    out = torch.full((max(n, 8),), 7.0, dtype=torch.bfloat16, device=DEV)
    C.check(C.load().prl_sum_parts(C.ptr(parts), C.dtype_code(parts), stride, nparts, n, C.ptr(out), C.stream_ptr(out.device)),
            "prl_sum_parts")
    want = torch.zeros(n, dtype=torch.float32, device=DEV)
    for p in range(nparts):  # this order, fp32, one rounding at the end
        want = want + parts[p, :n].float()
    assert torch.equal(out[:n], want.to(torch.bfloat16))
    assert bool((out[n:] == 7.0).all())  # nothing past the end is written
    lib = C.load()
    assert lib.prl_sum_parts(C.ptr(parts), C.dtype_code(parts), stride, nparts, 12, C.ptr(out), C.stream_ptr(out.device)) != 0
    assert lib.prl_sum_parts(C.ptr(parts), 7, stride, nparts, n, C.ptr(out), C.stream_ptr(out.device)) != 0


def test_block_products_on_the_mfma_kernels(monkeypatch):
    """The three products of a decoder block's linear layer through `mm_nt`, `mm_nn` and `weight_grad` with the
    kernels switched on for every shape they take: the LM-head kernels' results (so they did run), and the fp32
    product of the same bf16 operands to bf16 rounding."""
    from prime_rl_cse_amd import lm_head as L

    gen = torch.Generator(device=DEV).manual_seed(21)
    for rows, n_in, n_out in ((4096, 1024, 4096), (4096, 3072, 1024), (8192, 2048, 1024), (2048, 1024, 6144)):
        x = torch.randn(rows, n_in, generator=gen, device=DEV).to(torch.bfloat16)
        w = (torch.randn(n_out, n_in, generator=gen, device=DEV) * 0.05).to(torch.bfloat16)
        gy = (torch.randn(rows, n_out, generator=gen, device=DEV) * 0.1).to(torch.bfloat16)
        monkeypatch.setenv("PRL_LAYER_GEMMS", "hip")
        got = fused.mm_nt(x, w), fused.mm_nn(gy, w), fused.weight_grad(gy, x)
        ours = L.lm_head_fwd(x, w), L.lm_head_bwd_input(gy, w), L.lm_head_bwd_weight(gy, x)
        want = x.float() @ w.float().t(), gy.float() @ w.float(), gy.float().t() @ x.float()
        monkeypatch.setenv("PRL_LAYER_GEMMS", "lib")
        lib = fused.mm_nt(x, w), fused.mm_nn(gy, w), fused.weight_grad(gy, x)
        for g, o, w32, l, k in zip(got, ours, want, lib, (n_in, n_out, rows)):
            assert torch.equal(g, o)
            tol = 2.0 ** -8 * w32.abs() + 1e-6 * k ** 0.5 * w32.abs().max()
            assert not bool(((g.float() - w32).abs() > tol).any())
            # and no further from the exact product than the library's result is
            assert (g.float() - w32).norm().item() <= 1.05 * (l.float() - w32).norm().item() + 1e-6


def test_blocks_with_their_products_on_the_kernels_give_the_library_blocks_results(monkeypatch):
    """Two real-size decoder blocks over a row of 8192 tokens: with the layers' products on the kernels where the
    measured table puts them, logits and gradients are the library-only run's to bf16 accuracy, and the kernels ran."""
    import copy

    from prime_rl_cse_amd import attention, lm_head as L
    from prime_rl_cse_amd import trainer as T
    from prime_rl_cse_amd.replica import FlatState

    cfg = T.model_config("Qwen/Qwen3-0.6B", num_layers=2, vocab_size=4096)
    models = [T.get_model(cfg, DEV, torch.bfloat16, seed=5) for _ in range(2)]
    lengths = [128] * 64
    t = sum(lengths)
    gen = torch.Generator().manual_seed(6)
    ids = torch.randint(0, 4096, (1, t), generator=gen).to(DEV)
    pos = torch.cat([torch.arange(n) for n in lengths])[None]
    plan = attention.plan_from_position_ids(pos, DEV)
    assert any(L.layer_plan(t, 1024, 4096))
    calls = {"fwd": 0, "dx": 0, "dw": 0}
    real = L.lm_head_fwd, L.lm_head_bwd_input, L.lm_head_bwd_weight

    def counted(name, fn):
        def run(*a):
            calls[name] += 1
            return fn(*a)
        return run

    outs = []
    for model, mode in zip(models, ("auto", "lib")):
        monkeypatch.setenv("PRL_LAYER_GEMMS", mode)
        monkeypatch.setenv("PRL_LM_HEAD", "lib")
        fused.fuse_model(model)
        state = FlatState(model, compute_dtype=torch.bfloat16)
        try:
            L.lm_head_fwd, L.lm_head_bwd_input, L.lm_head_bwd_weight = (counted(n, f) for n, f in zip(calls, real))
            before = dict(calls)
            logits = T.forward(model, ids, pos.to(DEV), plan)
            logits.float().square().mean().backward()
        finally:
            L.lm_head_fwd, L.lm_head_bwd_input, L.lm_head_bwd_weight = real
        ran = {k: calls[k] - before[k] for k in calls}
        assert (sum(ran.values()) > 0) == (mode == "auto"), ran
        outs.append((logits.detach().float(), [p.grad.float() for p in state.params]))
        fused.unfuse_model(model)
    (la, ga), (lb, gb) = outs
    assert (la - lb).norm().item() <= 2e-2 * lb.norm().item()
    for a, b in zip(ga, gb):
        assert (a - b).norm().item() <= 3e-2 * b.norm().item() + 1e-9


def test_linear_with_a_bias_keeps_its_forward_and_moves_its_gradients(monkeypatch):
    """A layer with a bias (Qwen2.5's q, k, v): the forward stays `F.linear` with the bias, bit for bit; the input and
    weight gradients come from the kernels (every product switched to them here) and the bias's gradient is the
    column sum, all to bf16 accuracy of what autograd gives for `nn.Linear`."""
    gen = torch.Generator(device=DEV).manual_seed(31)
    lin = torch.nn.Linear(1024, 512, bias=True, device=DEV, dtype=torch.bfloat16)
    x = torch.randn(1, 4096, 1024, generator=gen, device=DEV).to(torch.bfloat16).requires_grad_()
    gy = (torch.randn(1, 4096, 512, generator=gen, device=DEV) * 0.1).to(torch.bfloat16)
    want = lin(x)
    want.backward(gy)
    ref = x.grad.clone(), lin.weight.grad.clone(), lin.bias.grad.clone()
    x.grad = lin.weight.grad = lin.bias.grad = None
    monkeypatch.setenv("PRL_LAYER_GEMMS", "hip")
    lin._prl_torch_forward = lin.forward
    got = fused._split_dw_linear_forward(lin, x)
    assert got.grad_fn is not None and "Linear" in type(got.grad_fn).__name__ and torch.equal(got, want)
    got.backward(gy)
    for g, r in zip((x.grad, lin.weight.grad, lin.bias.grad), ref):
        assert g.shape == r.shape and g.dtype == r.dtype
        assert (g.float() - r.float()).norm().item() <= 1e-2 * r.float().norm().item()
