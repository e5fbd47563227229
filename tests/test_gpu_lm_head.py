"""The LM head's three products on the MFMA kernels (csrc/lm_head.hip) against fp32 products of the same bf16
inputs.  The reference leaves these products to its BLAS library (src/prime_rl/trainer/model.py:251-254 for the
forward, autograd behind src/prime_rl/trainer/rl/train.py:310 for the two gradients), so what pins them is the
mathematical product: exactly on small-integer data, where every partial sum is an integer that fp32 holds
whatever the order of the additions, and to bf16 rounding of the result on random data (tolerance: half a unit in
the last place of bf16, 2^-9 relative, plus the fp32 summation error of a K-long sum)."""

import pytest
import torch

pytestmark = pytest.mark.gpu

import prime_rl_cse_amd  # noqa: E402,F401
from prime_rl_cse_amd import lm_head as L  # noqa: E402


def _ints(shape, lo, hi, gen, dev):
    return torch.randint(lo, hi + 1, shape, generator=gen, device=dev).to(torch.bfloat16)


# ragged edges on every side: rows and columns that are not multiples of the 256-wide tile, one K-tile, odd and
# even numbers of K-tiles, a K split over workgroups with an empty last split (1024 x 256 x 4160), workgroups that
# walk several tiles with edge tiles among them (1088 x 128 x 16448) and several fp32 slabs of different K splits
# (5120 x 3840 x 2048); many tokens and a small weight, where the WEIGHT gradient is split over ranges of the tokens: a
# short last split (4160 tokens: 65 K-tiles in 8 splits of 9), ragged tiles in every slab (8192 x 320 x 448), and
# a decoder block's q + k + v at a joined row's length (16384 x 1024 x 4096); layers that narrow (V < H), whose weight
# gradient transposes dlogits instead of hidden: split (8192 x 448 x 320, 16384 x 3072 x 1024) and whole (512 x 1088 x 576)
SHAPES = [(256, 64, 256), (512, 128, 1088), (192, 192, 832), (1024, 256, 4160), (2048, 1024, 2560), (64, 64, 64),
          (320, 448, 1984), (1088, 128, 16448), (5120, 3840, 2048), (4160, 256, 1024), (8192, 320, 448),
          (16384, 1024, 4096), (8192, 448, 320), (16384, 3072, 1024), (512, 1088, 576)]


@pytest.mark.parametrize("t,h,v", SHAPES)
def test_products_are_exact_on_integer_data(t, h, v):
    dev = torch.device("cuda:0")
    gen = torch.Generator(device=dev).manual_seed(t * 7 + h * 3 + v)
    x, w, g = _ints((t, h), -2, 2, gen, dev), _ints((v, h), -2, 2, gen, dev), _ints((t, v), -1, 1, gen, dev)
    assert torch.equal(L.lm_head_fwd(x, w), (x.float() @ w.float().t()).to(torch.bfloat16))
    assert torch.equal(L.lm_head_bwd_input(g, w), (g.float() @ w.float()).to(torch.bfloat16))
    assert torch.equal(L.lm_head_bwd_weight(g, x), (g.float().t() @ x.float()).to(torch.bfloat16))


@pytest.mark.parametrize("form", ["both", "transpose"])
@pytest.mark.parametrize("t,h,v", [(1024, 256, 4160), (2048, 1024, 2560), (4160, 256, 1024), (8192, 448, 320),
                                   (512, 1088, 576), (320, 448, 1984)])
def test_weight_gradient_is_exact_in_either_form(t, h, v, form, monkeypatch):
    """The weight gradient reads both operands along the tokens as they lie, or makes a transposed copy of the
    narrower one first (the library picks by the width of the weight); either form forced here, whole and split over
    the tokens, ragged tiles, V < H and V > H."""
    monkeypatch.setenv("PRL_LM_HEAD_DW_FORM", form)
    dev = torch.device("cuda:0")
    gen = torch.Generator(device=dev).manual_seed(t + h + v)
    x, g = _ints((t, h), -2, 2, gen, dev), _ints((t, v), -1, 1, gen, dev)
    for _ in range(3):  # (a stale operand piece would be a matter of timing)
        assert torch.equal(L.lm_head_bwd_weight(g, x), (g.float().t() @ x.float()).to(torch.bfloat16))


def test_products_are_exact_on_random_small_shapes():
    """Thirty shapes drawn from the multiples of 64 (tile counts 1 .. 7 per side, ragged strips of one to three tile
    columns, one to 24 K-tiles, K splits of the input gradient with a short last split)."""
    import random

    dev = torch.device("cuda:0")
    rng = random.Random(20240607)
    for case in range(30):
        t, h, v = (64 * rng.randint(1, 24) for _ in range(3))
        gen = torch.Generator(device=dev).manual_seed(case)
        x, w, g = _ints((t, h), -2, 2, gen, dev), _ints((v, h), -2, 2, gen, dev), _ints((t, v), -1, 1, gen, dev)
        assert torch.equal(L.lm_head_fwd(x, w), (x.float() @ w.float().t()).to(torch.bfloat16)), (t, h, v, "fwd")
        assert torch.equal(L.lm_head_bwd_input(g, w), (g.float() @ w.float()).to(torch.bfloat16)), (t, h, v, "dx")
        assert torch.equal(L.lm_head_bwd_weight(g, x), (g.float().t() @ x.float()).to(torch.bfloat16)), (t, h, v, "dw")


def _close(got, want32, k):
    # bf16 rounding of the result (2^-9 relative) plus fp32 accumulation over k terms of magnitude |a||b|
    tol = 2.0 ** -8 * want32.abs() + 1e-6 * k ** 0.5 * want32.abs().max()
    bad = (got.float() - want32).abs() > tol
    assert not bool(bad.any()), f"{int(bad.sum())} elements off; worst {float((got.float() - want32).abs().max())}"


@pytest.mark.parametrize("t,h,v", [(2048, 1024, 151936), (256, 2048, 152064), (1024, 3584, 8192)])
def test_products_on_random_data_at_recipe_vocabularies(t, h, v):
    dev = torch.device("cuda:0")
    gen = torch.Generator(device=dev).manual_seed(5)
    x = torch.randn((t, h), generator=gen, device=dev).to(torch.bfloat16)
    w = (torch.randn((v, h), generator=gen, device=dev) * 0.05).to(torch.bfloat16)
    g = (torch.randn((t, v), generator=gen, device=dev) * 0.01).to(torch.bfloat16)
    rows = min(t, 512)
    _close(L.lm_head_fwd(x, w)[:rows], x[:rows].float() @ w.float().t(), h)
    _close(L.lm_head_bwd_input(g, w), g.float() @ w.float(), v)
    cols = min(v, 8192)
    _close(L.lm_head_bwd_weight(g, x)[:cols], g[:, :cols].float().t() @ x.float(), t)


def test_strided_inputs_and_views():
    dev = torch.device("cuda:0")
    gen = torch.Generator(device=dev).manual_seed(9)
    big = _ints((256, 320), -2, 2, gen, dev)
    x = big[:, 64:192]  # row stride 320, offset 128 bytes
    w = _ints((512, 128), -2, 2, gen, dev)
    assert torch.equal(L.lm_head_fwd(x, w), (x.float() @ w.float().t()).to(torch.bfloat16))
    x3 = x.reshape(2, 128, 128)
    out = L.lm_head(x3.contiguous(), w)
    assert out.shape == (2, 128, 512)


def test_autograd_function_matches_torch_linear(monkeypatch):
    dev = torch.device("cuda:0")
    gen = torch.Generator(device=dev).manual_seed(11)
    x = _ints((1, 256, 128), -2, 2, gen, dev).requires_grad_(True)
    w = _ints((1088, 128), -2, 2, gen, dev).requires_grad_(True)
    gy = _ints((1, 256, 1088), -1, 1, gen, dev)
    torch.nn.functional.linear(x, w).backward(gy)
    want = (x.grad.clone(), w.grad.clone())
    for mode in ("hip", "auto", "lib"):
        monkeypatch.setenv("PRL_LM_HEAD", mode)
        x.grad = w.grad = None
        out = L.linear(x, w)
        assert torch.equal(out, torch.nn.functional.linear(x.detach(), w.detach()))
        out.backward(gy)
        assert torch.equal(x.grad, want[0]) and torch.equal(w.grad, want[1]), mode


def test_plan_follows_the_switches(monkeypatch):
    monkeypatch.delenv("PRL_LM_HEAD", raising=False)
    assert L.plan(2048, 1024, 151936) == (True, True, True)
    assert L.plan(2048, 5120, 152064) == (True, True, True)
    assert L.plan(2048, 3584, 152064) == (False, True, True)  # the library's forward is 1-2 % ahead there
    assert L.plan(100, 1024, 151936) == (False, False, False)  # not a multiple of 64: the library's
    monkeypatch.setenv("PRL_LM_HEAD", "hip")
    assert L.plan(2048, 1024, 151936) == (True, True, True)
    with pytest.raises(RuntimeError):
        L.plan(100, 1024, 151936)
    monkeypatch.setenv("PRL_LM_HEAD", "lib")
    monkeypatch.setenv("PRL_LM_HEAD_DW", "1")
    assert L.plan(2048, 1024, 151936) == (False, False, True)


def test_unsupported_shapes_are_refused_by_the_library():
    dev = torch.device("cuda:0")
    x = torch.zeros((100, 128), dtype=torch.bfloat16, device=dev)
    w = torch.zeros((256, 128), dtype=torch.bfloat16, device=dev)
    assert not L.supported(x, w)
    with pytest.raises(RuntimeError):
        L.lm_head_fwd(x, w)
    with pytest.raises(RuntimeError):
        L.lm_head_fwd(x.cpu(), w.cpu())
