"""PyTorch-ROCm's varlen_attn on packed rows of long samples against the masked row and the side-by-side layout."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch, torch.nn.functional as F
from torch.nn.attention.varlen import varlen_attn
import inspect
print(inspect.signature(varlen_attn))
dev = torch.device("cuda:0")
for lengths in ([2000, 1500, 300, 296], [512] * 8, [700, 650, 600, 550, 500, 450, 400, 246], [300] * 13 + [196]):

    t = sum(lengths); hq, hk, d = 16, 8, 128
    gen = torch.Generator(device=dev).manual_seed(0)
    q = torch.randn(t, hq, d, generator=gen, device=dev).to(torch.bfloat16).requires_grad_(True)
    k = torch.randn(t, hk, d, generator=gen, device=dev).to(torch.bfloat16).requires_grad_(True)
    v = torch.randn(t, hk, d, generator=gen, device=dev).to(torch.bfloat16).requires_grad_(True)
    cu = torch.tensor([0] + list(torch.tensor(lengths).cumsum(0)), dtype=torch.int32, device=dev)
    dout = torch.randn(t, hq, d, generator=gen, device=dev).to(torch.bfloat16)
    try:
        out = varlen_attn(q, k, v, cu, cu, max(lengths), max(lengths), is_causal=True)
        print("varlen_attn with 8 kv heads:", out.shape)
    except Exception as e:
        print("varlen_attn GQA failed:", repr(e)[:300])
        kk, vv = k.repeat_interleave(2, 1), v.repeat_interleave(2, 1)
        out = varlen_attn(q, kk, vv, cu, cu, max(lengths), max(lengths), is_causal=True)
        print("varlen_attn with repeated kv heads:", out.shape)
    out.backward(dout)
    # reference per sample
    a = 0; refs = []
    for n in lengths:
        qs, ks, vs = (x[a:a+n].transpose(0, 1)[None].float() for x in (q, k, v))
        refs.append(F.scaled_dot_product_attention(qs, ks.repeat_interleave(2, 1), vs.repeat_interleave(2, 1), is_causal=True)[0].transpose(0, 1))
        a += n
    ref = torch.cat(refs)
    print("max abs diff vs fp32 per-sample attention:", (out.float() - ref).abs().max().item())
    def timed(fn, n=10):
        for _ in range(3): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): fn()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
    def run_varlen():
        o = varlen_attn(q, k, v, cu, cu, max(lengths), max(lengths), is_causal=True); o.backward(dout)
    from prime_rl_cse_amd import attention
    pos = torch.cat([torch.arange(n) for n in lengths])[None]
    plan = attention.plan_from_position_ids(pos, dev)
    attention._HF_SDPA[:] = [None]
    q4, k4, v4 = (x.detach().transpose(0, 1)[None].requires_grad_(True) for x in (q, k, v))
    def run_side():
        with attention.use_plan(plan):
            o, _ = attention.packed_sdpa_attention(None, q4, k4, v4, None, scaling=None)
        o.backward(dout[None])
    def run_masked():
        seg = torch.repeat_interleave(torch.arange(len(lengths), device=dev), torch.tensor(lengths, device=dev))
        i = torch.arange(t, device=dev)
        mask = (seg[:, None] == seg[None, :]) & (i[:, None] >= i[None, :])
        o = F.scaled_dot_product_attention(q4, k4, v4, attn_mask=mask[None, None], enable_gqa=True)
        o.transpose(1, 2).backward(dout[None])
    line = f"lengths {lengths}: varlen fwd+bwd {timed(run_varlen):.0f} us; masked row {timed(run_masked):.0f} us"
    if plan is not None and plan.side_by_side:
        line += f"; side by side {timed(run_side):.0f} us"
    print(line)
