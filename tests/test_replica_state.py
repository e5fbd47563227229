"""The flat parameter / gradient / master-weight buffers behind `Replica` and the end of a step over them:
against torch's own clip_grad_norm_ + AdamW on separate fp32 parameters, on one rank and on two (gloo)."""

import os

import torch
from torch import nn

from prime_rl_cse_amd.replica import ALIGN, FlatState


class Tiny(nn.Module):
    def __init__(self):
        super().__init__()
        self.a = nn.Linear(37, 19)  # sizes that are no multiple of anything
        self.b = nn.Embedding(11, 37)
        self.c = nn.Linear(19, 11, bias=False)
        self.c.weight = nn.Parameter(self.c.weight)  # untied
        self.tied = nn.Linear(37, 11, bias=False)
        self.tied.weight = self.b.weight  # one parameter in two places, as embed_tokens / lm_head


def _grads(seed: int, model: nn.Module) -> list[torch.Tensor]:
    gen = torch.Generator().manual_seed(seed)
    return [torch.randn(p.shape, generator=gen) * 3.0 for p in model.parameters()]


def _expected(grads_per_rank: list[list[torch.Tensor]], steps: int, max_norm: float):
    """Separate fp32 parameters, mean gradient over ranks, torch's clipping and AdamW."""
    torch.manual_seed(0)
    ref = Tiny()
    params = list(ref.parameters())
    opt = torch.optim.AdamW(params, lr=1e-2, weight_decay=0.01, betas=(0.9, 0.999))
    norms = []
    for s in range(steps):
        for k, p in enumerate(params):
            p.grad = sum(g[k] for g in grads_per_rank) / len(grads_per_rank) * (s + 1)
        norms.append(float(torch.nn.utils.clip_grad_norm_(params, max_norm=max_norm)))
        opt.step()
    return [p.detach() for p in params], norms


def _run_steps(state: FlatState, model: nn.Module, my_grads: list[torch.Tensor], steps: int, max_norm: float):
    opt = torch.optim.AdamW(state.holder().parameters(), lr=1e-2, weight_decay=0.01, betas=(0.9, 0.999))
    norms = []
    for s in range(steps):
        for g, o, n in zip(my_grads, state.offsets, state.numel):
            state.acc32[o : o + n] += g.reshape(-1) * (s + 1)
        norms.append(float(state.finish_step(opt, max_norm)))
        assert not state.acc32.any()  # cleared for the next step
    return norms


def test_flat_layout_and_one_rank_step():
    torch.manual_seed(0)
    model = Tiny()
    before = [p.detach().clone() for p in model.parameters()]
    assert len(before) == 4  # the tied weight counts once
    state = FlatState(model)
    assert state.world == 1 and state.total % ALIGN == 0 and all(o % ALIGN == 0 for o in state.offsets)
    for p, b, o in zip(model.parameters(), before, state.offsets):
        assert p.dtype == torch.bfloat16 and torch.equal(p, b.to(torch.bfloat16))
        assert p.data_ptr() == state.flat16.data_ptr() + 2 * o  # a view into the flat buffer
        assert torch.equal(state.master[o : o + b.numel()], b.reshape(-1))  # master weights keep fp32
    assert model.tied.weight is model.b.weight
    grads = _grads(1, model)
    norms = _run_steps(state, model, grads, steps=3, max_norm=1.0)
    want, want_norms = _expected([grads], steps=3, max_norm=1.0)
    assert norms == __import__("pytest").approx(want_norms, rel=1e-5)
    for p, w, o in zip(model.parameters(), want, state.offsets):
        torch.testing.assert_close(state.master[o : o + w.numel()].view(w.shape), w, rtol=1e-5, atol=1e-6)
        assert torch.equal(p, state.master[o : o + w.numel()].view(w.shape).to(torch.bfloat16))
    # padding between parameters never moves
    pad = torch.ones(state.total, dtype=torch.bool)
    for o, n in zip(state.offsets, state.numel):
        pad[o : o + n] = False
    assert not state.master.detach()[pad].any() and not state.flat16[pad].any()


def _rank(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(0)
        model = Tiny()
        state = FlatState(model)
        assert state.world == world and state.shard * world == state.total
        norms = _run_steps(state, model, _grads(10 + rank, model), steps=2, max_norm=0.5)
        q.put((rank, norms, [p.detach().float().tolist() for p in model.parameters()],
               state.master.detach().tolist()))
    finally:
        dist.destroy_process_group()


def test_two_ranks_shard_master_weights_and_agree_over_gloo():
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rank, args=(r, 2, 29763, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=120) for _ in procs), key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    torch.manual_seed(0)
    model = Tiny()
    want, want_norms = _expected([_grads(10, model), _grads(11, model)], steps=2, max_norm=0.5)
    import pytest

    for rank, norms, params, master in res:
        assert norms == pytest.approx(want_norms, rel=1e-5)
        for got, w in zip(params, want):  # every rank ends with the same full bf16 copy
            assert torch.equal(torch.tensor(got), w.to(torch.bfloat16).float())
    # the two master shards, end to end, are the fp32 parameters
    full = torch.tensor(res[0][3] + res[1][3])
    state = FlatState(model)
    for w, o in zip(want, state.offsets):
        torch.testing.assert_close(full[o : o + w.numel()].view(w.shape), w, rtol=1e-5, atol=1e-6)


# --------------------------------------------------------------------------- the exchange in buckets


class Blocks(nn.Module):
    """Named like a causal LM: an embedding, numbered blocks, a final norm, a tied head."""

    def __init__(self, n_layers: int = 5):
        super().__init__()
        self.model = nn.Module()
        self.model.embed_tokens = nn.Embedding(53, 24)
        self.model.layers = nn.ModuleList()
        for _ in range(n_layers):
            block = nn.Module()
            block.attn = nn.Linear(24, 24, bias=False)
            block.mlp = nn.Linear(24, 31)
            block.norm = nn.LayerNorm(24)
            self.model.layers.append(block)
        self.model.norm = nn.LayerNorm(24)
        self.lm_head = nn.Linear(24, 53, bias=False)
        self.lm_head.weight = self.model.embed_tokens.weight


def test_bucket_plan_keeps_blocks_whole():
    from prime_rl_cse_amd.replica import bucket_plan

    model = Blocks()
    names = [n for n, _ in model.named_parameters()]
    numel = [p.numel() for _, p in model.named_parameters()]
    per_block = 24 * 24 + 24 * 31 + 31 + 48
    assert bucket_plan(names, numel, None) == [(0, len(names))]
    plan = bucket_plan(names, numel, 2 * per_block)
    # the embedding table (53 * 24 elements) with the first block, two blocks, two blocks with the final norm
    # (too small to open a bucket of its own)
    assert plan == [(0, 6), (6, 16), (16, 28)]
    # a bucket size smaller than anything: every block, and every parameter outside the blocks, by itself
    assert [b - a for a, b in bucket_plan(names, numel, 1)] == [1, 5, 5, 5, 5, 5, 1, 1]
    # the dimensions of Qwen3-0.6B at 256 MB of fp32 sums per bucket: the table, then four blocks at a time
    names = ["model.embed_tokens.weight"] + [f"model.layers.{i}.w{j}" for i in range(28) for j in range(3)] \
        + ["model.norm.weight"]
    numel = [151936 * 1024] + [15_000_000, 700_000, 30_000] * 28 + [1024]
    assert [b - a for a, b in bucket_plan(names, numel, 2**26)] == [1] + [12] * 6 + [13]


def test_weight_arrival_hooks_sit_on_modules_whose_forward_runs():
    """The gather / forward overlap waits for a bucket in a forward pre-hook.  A hook on an `nn.ModuleList` never
    fires (the parent indexes into it), so with the table in one bucket and ALL blocks in another, the blocks
    themselves must carry the hooks; every bucket must be waited for by some module that is really called, and
    every module with parameters must find its bucket on a hook at or above it."""
    from prime_rl_cse_amd.replica import arrival_hooks

    model = Blocks()
    params = list(model.parameters())  # the tied head shares the table's parameter
    names = [n for n, _ in model.named_parameters()]
    for layout in ("table_and_blocks", "every_block_alone", "one_bucket"):
        if layout == "table_and_blocks":
            bucket_of = [0 if n == "model.embed_tokens.weight" else 1 for n in names]
        elif layout == "every_block_alone":
            bucket_of = [0 if n == "model.embed_tokens.weight" else
                         1 + int(n.split(".")[2]) if n.startswith("model.layers.") else 6 for n in names]
        else:
            bucket_of = [0] * len(names)
        hooks = arrival_hooks(model, params, bucket_of)
        assert hooks and not any(isinstance(m, (nn.ModuleList, nn.ModuleDict)) for m, _ in hooks), layout
        assert {k for _, which in hooks for k in which} == set(bucket_of), layout
        hooked = {id(m): set(which) for m, which in hooks}
        index = {id(p): bucket_of[i] for i, p in enumerate(params)}

        def covered(module: nn.Module, above: set[int]) -> None:
            above = above | hooked.get(id(module), set())
            for p in module.parameters(recurse=False):
                assert index[id(p)] in above, (layout, type(module).__name__)
            for child in module.children():
                covered(child, above)

        covered(model, set())
        if layout == "table_and_blocks":
            blocks = [m for m, which in hooks if which == [1] and hasattr(m, "attn")]
            assert len(blocks) == 5  # each block waits for the blocks' bucket itself


def test_every_rank_issues_its_buckets_in_one_order():
    """Collectives have to be issued in the same order on every rank.  Whatever order a busy rank's gradients
    land in, with parameters that get no gradient at all, and on a rank that runs no pass, the buckets go to
    the exchange in `ExchangeOrder.sequence()`: not-held buckets back to front, the held one(s) last."""
    import random

    from prime_rl_cse_amd.replica import ExchangeOrder

    rnd = random.Random(5)
    for trial in range(200):
        n = rnd.randint(1, 9)
        held = [k == 0 and rnd.random() < 0.8 for k in range(n)]  # the table's bucket is the first one
        sizes = [rnd.randint(1, 6) for _ in range(n)]
        idle = ExchangeOrder(held).sequence()
        assert sorted(idle) == list(range(n))
        order = ExchangeOrder(held)
        order.begin(sizes)
        # gradients of every bucket in random order; some parameters never get one
        events = [k for k in range(n) for _ in range(sizes[k]) if rnd.random() < 0.85]
        rnd.shuffle(events)
        issued = []
        for k in events:
            now = order.landed(k)
            assert all(not held[j] for j in now)
            issued += now
        rest, last = order.finish()
        assert issued + rest + last == idle, (trial, held, sizes, events)
    # every gradient landed, back to front: everything but the held bucket leaves from inside backward
    order = ExchangeOrder([True, False, False])
    order.begin([1, 2, 1])
    assert order.landed(2) == [2] and order.landed(1) == [] and order.landed(0) == [] and order.landed(1) == [1]
    assert order.finish() == ([], [0])


def _bucketed_rank(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        out = {}
        for max_norm in (1e9, 0.5):  # without and with clipping
            states = []
            for bucket_bytes in (None, 4 * 1600):
                torch.manual_seed(0)
                model = Blocks()
                state = FlatState(model, compute_dtype=torch.float32, bucket_bytes=bucket_bytes)
                opt = torch.optim.AdamW(state.holder().parameters(), lr=1e-2, weight_decay=0.01)
                grads = _grads(20 + rank, model)
                norms, means = [], []
                # ranks differ in how many passes make up their step (1, 2 or 3 accumulations of a share of the
                # gradient), and with more than two ranks the last one has nothing to learn from at all
                passes = rank % 3 + 1
                idle = world > 2 and rank == world - 1
                for s in range(3):
                    if not idle:
                        for _ in range(passes):
                            for g, o, n in zip(grads, state.offsets, state.numel):
                                state.acc32[o : o + n] += g.reshape(-1) * ((s + 1) / passes)
                    if bucket_bytes is not None and s == 1:
                        # as backward's hooks do it, and as a rank without a pass does it: the order of
                        # `ExchangeOrder` (buckets handed over early, back to front, the table's last)
                        from prime_rl_cse_amd.replica import ExchangeOrder

                        held = [k == 0 for k in range(len(state.buckets))]
                        for k in ExchangeOrder(held).sequence()[:-1]:
                            state.reduce_bucket(state.buckets[k])
                    state.wait_reduced()
                    mean = torch.zeros(state.total)
                    for lo, hi, at in state.owned_ranges():
                        mean[lo:hi] = state._shard_grad[at : at + hi - lo]
                    means.append(mean)
                    norms.append(float(state.finish_step(opt, max_norm)))
                states.append((state, model, norms, means))
            (one, m1, norms1, means1), (many, m2, norms2, means2) = states
            assert len(one.buckets) == 1 and len(many.buckets) > 3
            assert many.shard * world == many.total and sum(b.shard for b in many.buckets) == many.shard
            for a, b in zip(means1, means2):  # this rank's elements of the mean gradient, wherever they live
                for (o1, n1), (o2, n2) in zip(zip(one.offsets, one.numel), zip(many.offsets, many.numel)):
                    x, y = a[o1 : o1 + n1], b[o2 : o2 + n2]
                    both = (x != 0) & (y != 0)  # (each layout gives this rank other elements)
                    assert torch.equal(x[both], y[both])
            same_weights = all(torch.equal(p1, p2) for p1, p2 in zip(m1.parameters(), m2.parameters()))
            # "to rounding": another order of adding the ranks' shares moves a mean gradient in its last bit, and
            # AdamW turns that into a visible step only where the shares nearly cancel (update = lr * g / |g|
            # early on): all but a handful of elements within 1e-6, none further than a hundredth of one step
            diff = torch.cat([(p1 - p2).abs().reshape(-1) for p1, p2 in zip(m1.parameters(), m2.parameters())])
            close = bool(diff.max() < 1e-4) and float((diff > 1e-6).float().mean()) < 5e-3
            # master shard against the compute copy (fp32 here): the same numbers
            for lo, hi, at in many.owned_ranges():
                assert torch.equal(many.master.detach()[at : at + hi - lo], many.flat16[lo:hi])
            out[max_norm] = (norms1, norms2, same_weights, close)
        q.put((rank, out))
    except BaseException as e:  # the parent should see the failure, not a time-out
        q.put((rank, repr(e)))
        raise
    finally:
        dist.destroy_process_group()


@__import__("pytest").mark.parametrize("world", [2, 4, 8])
def test_bucketed_exchange_gives_the_step_of_the_single_exchange_over_gloo(world):
    """Two, four and eight ranks (each with its own number of passes per step, one of them idle): the flat state cut into buckets (each sharded over the ranks by itself, some reduced early)
    against the one-bucket layout.  With two ranks and no clipping the updated weights agree bit for bit (every
    element sees the same mean gradient, a + b in either order, and the same AdamW arithmetic); the gradient norm
    is a sum in another order, so with clipping the weights agree to rounding.  With four ranks the collective
    adds the ranks' contributions in an order that depends on where an element lies in the buffer it reduces,
    so there the two layouts agree to rounding throughout."""
    import pytest
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_bucketed_rank, args=(r, world, 29764 + world, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, out in res:
        assert isinstance(out, dict), out
        norms1, norms2, same, close = out[1e9]
        assert norms2 == pytest.approx(norms1, rel=1e-6) and (same if world == 2 else close)
        norms1, norms2, same, close = out[0.5]
        assert norms2 == pytest.approx(norms1, rel=1e-6) and close


class _Stack(nn.Module):
    """Parameters with the names a transformer has (an embedding, numbered blocks, a final norm), odd sizes."""

    def __init__(self, blocks: int, width: int):
        super().__init__()
        self.embed_tokens = nn.Embedding(53, width)
        self.layers = nn.ModuleList(
            nn.ModuleDict({"attn": nn.Linear(width, width + 3), "norm": nn.LayerNorm(width), "mlp": nn.Linear(width, 7)})
            for _ in range(blocks))
        self.norm = nn.LayerNorm(width)


@__import__("pytest").mark.parametrize("world", [2, 3, 8])
@__import__("pytest").mark.parametrize("bucket_bytes", [None, 4 * 700, 4 * 5000])
def test_layout_of_rank_counts_that_never_run_here(world, bucket_bytes, monkeypatch):
    """What every rank of a W-rank job lays out, built one rank after the other in this process (the layout is
    made without any collective): the buckets tile the flat buffers, every bucket splits evenly over the ranks,
    the ranks' master shards put together are exactly the parameters, and nobody's shard overlaps another's."""
    import torch.distributed as dist

    states, models = [], []
    for rank in range(world):
        monkeypatch.setattr(dist, "is_initialized", lambda: True)
        monkeypatch.setattr(dist, "get_world_size", lambda group=None: world)
        monkeypatch.setattr(dist, "get_rank", lambda group=None, rank=rank: rank)
        torch.manual_seed(3)
        model = _Stack(blocks=5, width=29)
        if rank == 0:
            values = [p.detach().clone() for p in model.parameters()]
        states.append(FlatState(model, None, torch.float32, bucket_bytes))
        models.append(model)
    first = states[0]
    assert all(s.total == first.total and s.offsets == first.offsets and s.shard == first.shard for s in states)
    assert first.total == world * first.shard
    # buckets: back to back, whole blocks, every one a multiple of ALIGN elements per rank
    assert first.buckets[0].lo == 0 and first.buckets[-1].hi == first.total
    for a, b in zip(first.buckets, first.buckets[1:]):
        assert a.hi == b.lo and a.last == b.first
    for b in first.buckets:
        assert (b.hi - b.lo) == world * b.shard and b.shard % ALIGN == 0
        block_of = {n.split(".")[1] for n in first.names[b.first : b.last] if n.startswith("layers.")}
        for other in first.buckets:
            if other is not b:
                assert not block_of & {n.split(".")[1] for n in first.names[other.first : other.last]
                                       if n.startswith("layers.")}
    if bucket_bytes is not None and bucket_bytes < 4 * 5000:
        assert len(first.buckets) > 2
    # the shards of all ranks, put where `owned_ranges` says, give the flat fp32 parameters exactly once
    flat = torch.full((first.total,), float("nan"))
    covered = torch.zeros(first.total, dtype=torch.int32)
    for s in states:
        for lo, hi, m_lo in s.owned_ranges():
            flat[lo:hi] = s.master.detach()[m_lo : m_lo + hi - lo]
            covered[lo:hi] += 1
    assert bool((covered == 1).all())
    for v, o, n in zip(values, first.offsets, first.numel):
        assert torch.equal(flat[o : o + n], v.reshape(-1))
    pad = torch.ones(first.total, dtype=torch.bool)
    for o, n in zip(first.offsets, first.numel):
        pad[o : o + n] = False
    assert not flat[pad].any()
