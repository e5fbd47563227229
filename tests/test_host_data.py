"""Host side of the path: packing rollouts into micro-batches (against the
reference's vectors), the rollout file format and the loaders, and the
world_size-2 sharding over gloo."""

import os
from pathlib import Path

import pytest
import torch

import prime_rl_cse_amd as prl
from prime_rl_cse_amd import batch as B
from prime_rl_cse_amd import data as D
from prime_rl_cse_amd.world import get_world, reset_world
from tests.helpers import load_npz

KEYS = ("input_ids", "advantages", "loss_mask", "position_ids", "logprobs")


def rollouts_of_case(d, case):
    n_roll, mbs, seq_len = d[f"c{case}__meta"].tolist()
    out = []
    for j in range(n_roll):
        p = d[f"c{case}__r{j}__prompt"].tolist()
        out.append(
            B.Rollout(j, p, [0] * len(p), d[f"c{case}__r{j}__completion"].tolist(), d[f"c{case}__r{j}__cmask"].tolist(),
                      d[f"c{case}__r{j}__clogp"].tolist(), False, 0.0, float(d[f"c{case}__r{j}__adv"]))
        )
    return out, n_roll, mbs, seq_len


@pytest.mark.parametrize("case", range(4))
@pytest.mark.parametrize("w", (1, 2, 4))
def test_prepare_batch_matches_reference_vectors(case, w):
    d = load_npz("batches.npz")
    rollouts, n_roll, mbs, seq_len = rollouts_of_case(d, case)
    got = prl.prepare_batch(rollouts, 0.7, None, n_roll, mbs, seq_len, w)
    assert len(got) == w
    for rank in range(w):
        assert len(got[rank]) == int(d[f"c{case}__w{w}__rank{rank}__rows"])
        for m, row in enumerate(got[rank]):
            assert row["temperature"] == 0.7
            for k in KEYS:
                want = d[f"c{case}__w{w}__rank{rank}__mb{m}__{k}"]
                assert row[k].dtype == want.dtype and torch.equal(row[k], want), (rank, m, k)


def test_too_long_rollout_is_refused():
    r = B.Rollout(0, [1] * 5, [0] * 5, [2] * 5, [1] * 5, [-0.1] * 5, False, 0.0, 0.0)
    with pytest.raises(ValueError, match="greater than sequence length"):
        prl.prepare_sample(r, 9)


@pytest.fixture
def clean_world(monkeypatch):
    reset_world()
    yield monkeypatch
    reset_world()


def test_world_reads_torchrun_env(clean_world):
    for k, v in dict(RANK="3", WORLD_SIZE="8", LOCAL_RANK="3", LOCAL_WORLD_SIZE="8").items():
        clean_world.setenv(k, v)
    w = get_world()
    assert (w.rank, w.world_size, w.local_rank, w.num_nodes, w.is_master) == (3, 8, 3, 1, False)
    assert get_world() is w


def test_fake_loader_shapes_and_sharding(clean_world):
    # src/prime_rl/trainer/rl/data.py:30: micro-batches per rank = batch // micro_batch // world
    clean_world.setenv("WORLD_SIZE", "2")
    clean_world.setenv("LOCAL_WORLD_SIZE", "2")
    clean_world.setenv("RANK", "1")
    clean_world.setenv("LOCAL_RANK", "1")
    loader = prl.FakeDataLoader(prl.FakeDataLoaderConfig(batch_size=128, micro_batch_size=16, seq_len=128))
    rows = loader.get_batch()
    assert len(rows) == 4
    r = rows[0]
    assert r["input_ids"].shape == (1, 2048) and r["input_ids"].dtype == torch.int64
    assert r["position_ids"][0, :130].tolist() == list(range(128)) + [0, 1]
    assert r["loss_mask"].dtype == torch.bool and bool(r["loss_mask"].all()) and r["temperature"] == 1.0
    assert int(r["input_ids"].max()) < 100
    with pytest.raises(ValueError):
        prl.FakeDataLoaderConfig(batch_size=10, micro_batch_size=4)


def test_rollout_files_round_trip(tmp_path, clean_world):
    d = load_npz("batches.npz")
    rollouts, n_roll, mbs, seq_len = rollouts_of_case(d, 0)
    per_rank = prl.prepare_batch(rollouts, 1.0, None, n_roll, mbs, seq_len, 2)
    paths = prl.save_rollout_batches(per_rank, tmp_path, step=3)
    assert [p.name for p in paths] == ["rank_0.pt", "rank_1.pt"] and not list(tmp_path.rglob("*.tmp"))
    clean_world.setenv("WORLD_SIZE", "2")
    clean_world.setenv("LOCAL_WORLD_SIZE", "2")
    clean_world.setenv("RANK", "1")
    clean_world.setenv("LOCAL_RANK", "1")
    loader = prl.DataLoader(tmp_path, start_step=3)
    assert loader.get_rollout_path() == tmp_path / "rollouts" / "step_3" / "rank_1.pt"
    loader.wait_for_batch(interval=0.01, timeout=1)
    got = loader.get_batch()
    assert loader.current_step == 4 and len(got) == len(per_rank[1])
    for a, b in zip(got, per_rank[1]):
        assert a["temperature"] == b["temperature"]
        for k in KEYS:
            assert torch.equal(a[k], b[k])
    with pytest.raises(TimeoutError):
        loader.wait_for_batch(interval=0.01, timeout=0.05)


def _rank_main(rank, world, port, root, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), LOCAL_WORLD_SIZE=str(world))
    import torch.distributed as dist

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        reset_world()
        loader = prl.DataLoader(Path(root), start_step=0)
        loader.wait_for_batch(interval=0.01, timeout=30)
        rows = loader.get_batch()
        n = torch.tensor([len(rows)])
        counts = [torch.zeros(1, dtype=torch.long) for _ in range(world)]
        dist.all_gather(counts, n)
        # the loss normaliser is local to the rank (src/prime_rl/trainer/rl/train.py:263-267): no collective
        scale = sum(int(r["loss_mask"].sum()) for r in rows)
        toks = torch.tensor([sum(r["input_ids"].numel() for r in rows)])
        dist.all_reduce(toks)
        q.put((rank, [int(c) for c in counts], scale, int(toks), [r["input_ids"][0].tolist() for r in rows]))
    finally:
        dist.destroy_process_group()


def test_two_ranks_each_load_their_own_shard_over_gloo(tmp_path):
    import torch.multiprocessing as mp

    d = load_npz("batches.npz")
    rollouts, n_roll, mbs, seq_len = rollouts_of_case(d, 1)
    per_rank = prl.prepare_batch(rollouts, 1.0, None, n_roll, mbs, seq_len, 2)
    prl.save_rollout_batches(per_rank, tmp_path, step=0)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rank_main, args=(r, 2, 29741, str(tmp_path), q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, c0, s0, t0, ids0), (r1, c1, s1, t1, ids1) = res
    assert c0 == c1 and c0[0] == c0[1] == len(per_rank[0])  # same number of rows on every rank
    assert t0 == t1 == sum(r["input_ids"].numel() for rows in per_rank for r in rows)
    assert ids0 == [r["input_ids"][0].tolist() for r in per_rank[0]]
    assert ids1 == [r["input_ids"][0].tolist() for r in per_rank[1]]
    assert s0 == sum(int(r["loss_mask"].sum()) for r in per_rank[0])


def test_scored_positions_are_the_positions_whose_next_token_is_scored():
    import prime_rl_cse_amd as prl

    mask = torch.tensor([[True, False, True, True, False, False, True, False]])
    assert prl.scored_positions(mask).tolist() == [1, 2, 5]  # tokens 2, 3 and 6; token 0 has no position
    assert prl.scored_positions(torch.zeros(1, 5, dtype=torch.bool)).numel() == 0
    assert prl.scored_positions(torch.ones(1, 5, dtype=torch.bool)).tolist() == [0, 1, 2, 3]
    # rounded up with positions nobody scores, never beyond the row, never when nothing is scored
    padded = prl.scored_positions(mask, multiple=4)
    assert padded.tolist() == [0, 1, 2, 5] and set([1, 2, 5]) <= set(padded.tolist())
    assert prl.scored_positions(mask, multiple=16).tolist() == list(range(7))
    assert prl.scored_positions(torch.zeros(1, 5, dtype=torch.bool), multiple=4).numel() == 0
    assert prl.scored_positions(mask, multiple=3).tolist() == [1, 2, 5]


def test_scored_rows_engage_only_where_they_apply(monkeypatch):
    """`trainer.scored_rows` (host logic): token mode, a model computing bf16 logits over a vocabulary the row
    kernel takes, and at most 0.9 of a row's positions left after rounding up to a multiple of 256."""
    import types

    import prime_rl_cse_amd as prl
    from prime_rl_cse_amd import trainer as T

    def model(dtype, vocab):
        m = torch.nn.Linear(2, 2).to(dtype)
        m.config = types.SimpleNamespace(vocab_size=vocab)
        return m

    t = 2048
    mask = torch.zeros(1, t, dtype=torch.bool)
    mask[0, 600:1500] = True  # 900 scored tokens
    mb = {"loss_mask": mask}
    tok, seq = prl.LossConfig(), prl.LossConfig(ratio_type="sequence")
    keep = T.scored_rows(mb, tok, model(torch.bfloat16, 151936), "cpu")
    assert keep is not None and keep.numel() == 1024 and keep.dtype == torch.int64
    assert set(range(599, 1499)) <= set(keep.tolist())  # position p feeds token p + 1
    assert T.scored_rows(mb, seq, model(torch.bfloat16, 151936), "cpu") is None
    assert T.scored_rows(mb, tok, model(torch.float32, 151936), "cpu") is None
    assert T.scored_rows(mb, tok, model(torch.bfloat16, 151937), "cpu") is None  # not a multiple of 8
    assert T.scored_rows(mb, tok, model(torch.bfloat16, 200000), "cpu") is None  # wider than a row's registers
    fsdp_like = model(torch.float32, 151936)
    fsdp_like._prl_compute_dtype = torch.bfloat16  # fp32 parameters, bf16 compute: what `setup_fsdp` records
    assert T.scored_rows(mb, tok, fsdp_like, "cpu").numel() == 1024
    dense = {"loss_mask": torch.ones(1, t, dtype=torch.bool)}
    assert T.scored_rows(dense, tok, model(torch.bfloat16, 151936), "cpu") is None
    nearly = {"loss_mask": torch.rand(1, t, generator=torch.Generator().manual_seed(0)) < 0.89}
    assert T.scored_rows(nearly, tok, model(torch.bfloat16, 151936), "cpu") is None  # ~1820, rounded up: over 0.9 of 2047
    most = {"loss_mask": torch.rand(1, t, generator=torch.Generator().manual_seed(0)) < 0.8}
    assert T.scored_rows(most, tok, model(torch.bfloat16, 151936), "cpu").numel() == 1792  # ~1640 -> 1792 of 2047
    assert T.scored_rows({"loss_mask": torch.zeros(1, t, dtype=torch.bool)}, tok, model(torch.bfloat16, 151936), "cpu") is None
    # long rows of narrow models: multiples of 2048 rows, and only up to three quarters of the row
    # the LM head's products on the BLAS library: coarse steps for long rows of narrow models
    assert T.gather_rule(16384, 1024, True) == (2048, 0.76) and T.gather_rule(16384, 3584, True) == (256, 0.9)
    assert T.gather_rule(8192, 1024, True) == (256, 0.9) and T.gather_rule(16384, 0, True) == (256, 0.9)
    # on this package's kernels the cost follows the row count everywhere
    assert T.gather_rule(16384, 1024) == (256, 0.9) and T.gather_rule(16384, 2048, False) == (256, 0.9)
    narrow = model(torch.bfloat16, 151936)
    narrow.config.hidden_size = 1024
    long_mask = torch.zeros(1, 16384, dtype=torch.bool)
    long_mask[0, 1000:12000] = True  # 11000 scored tokens
    long_mask[0, 12000:13000] = True  # 12000
    long_mask[0, 13000:13400] = True  # 12400
    monkeypatch.setenv("PRL_LM_HEAD", "lib")  # the library's products: 12400 would round up to 14336: the whole row
    assert T.scored_rows({"loss_mask": long_mask}, tok, narrow, "cpu") is None
    long_mask[0, 12000:] = False  # 11000 -> 12288 rows
    assert T.scored_rows({"loss_mask": long_mask}, tok, narrow, "cpu").numel() == 12288
    monkeypatch.delenv("PRL_LM_HEAD")  # this package's kernels: a multiple of 256, up to 0.9 of the row
    assert T.scored_rows({"loss_mask": long_mask}, tok, narrow, "cpu").numel() == 11008
    long_mask[0, 12000:13400] = True  # 12400 -> 12544
    assert T.scored_rows({"loss_mask": long_mask}, tok, narrow, "cpu").numel() == 12544
    wide = model(torch.bfloat16, 152064)
    wide.config.hidden_size = 3584
    assert T.scored_rows({"loss_mask": long_mask}, tok, wide, "cpu").numel() == 12544  # 12400 -> a multiple of 256
    monkeypatch.setenv("PRL_GATHER_SCORED", "0")
    assert T.scored_rows(mb, tok, model(torch.bfloat16, 151936), "cpu") is None


def test_split_by_gradient_sorts_samples_by_whether_anything_can_be_learned_from_them(monkeypatch):
    from prime_rl_cse_amd import trainer as T

    monkeypatch.setenv("PRL_SPLIT_NO_GRADIENT", "1")  # (the default only separates steps of 16 K tokens and more)

    def row(lens, adv_per_sample, prompt=4, first_scored=False):
        t = sum(lens)
        pos = torch.cat([torch.arange(n) for n in lens]).unsqueeze(0)
        adv = torch.cat([torch.full((n,), float(a)) for n, a in zip(lens, adv_per_sample)]).unsqueeze(0)
        mask = pos >= prompt
        if first_scored:
            mask[0, lens[0]] = True  # the second sample's first token is scored: it depends on its neighbour
        return {"input_ids": torch.arange(t).unsqueeze(0), "position_ids": pos, "advantages": adv,
                "logprobs": torch.zeros(1, t), "loss_mask": mask, "temperature": 1.0}

    mixed, dead, alive = row([8, 8, 8], [0.5, 0.0, -0.5]), row([10, 6], [0.0, 0.0]), row([16], [1.0])
    stuck = row([8, 8], [0.0, 1.0], first_scored=True)
    unscored_sample = row([8, 3, 8], [1.0, 0.0, 0.0])  # the middle sample is all prompt: dropped
    grad, grad_row, fwd, fwd_row = T.split_by_gradient([mixed, dead, alive, stuck, unscored_sample])
    assert grad_row == [0, 0, 2, 3, 4] and fwd_row == [0, 1, 4]
    assert [m["input_ids"][0].tolist() for m in grad[:2]] == [list(range(0, 8)), list(range(16, 24))]
    assert fwd[0]["input_ids"][0].tolist() == list(range(8, 16)) and fwd[0]["position_ids"][0].tolist() == list(range(8))
    assert fwd[1] is dead and grad[2] is alive and grad[3] is stuck  # rows of one kind, or not movable: left whole
    assert grad[4]["input_ids"][0].tolist() == list(range(0, 8)) and fwd[2]["input_ids"][0].tolist() == list(range(11, 19))
    assert all(m["temperature"] == 1.0 for m in grad + fwd)
    assert T.split_by_gradient([alive, row([8, 8], [1.0, 2.0])]) is None  # nothing to separate
    nan_adv = row([8, 8], [float("nan"), 0.0])
    g, _, f, _ = T.split_by_gradient([nan_adv])
    assert len(g) == 1 and len(f) == 1  # a NaN advantage is not "nothing to learn": it has to reach the loss
    # per-row losses from per-piece losses
    out = T.losses_per_given_row([torch.tensor([1.0, 2.0, 3.0]), torch.tensor([4.0])], [0, 2, 0, 3], 5, "cpu")
    assert out.tolist() == [4.0, 0.0, 2.0, 4.0, 0.0]
    assert T.losses_per_given_row([], [], 3, "cpu").tolist() == [0.0, 0.0, 0.0]
    # by default short steps stay as they are, long ones are separated
    monkeypatch.delenv("PRL_SPLIT_NO_GRADIENT")
    assert T.split_by_gradient([mixed, dead, alive]) is None
    long_dead, long_alive = row([8192], [0.0]), row([8192], [1.0])
    g, gr, f, fr = T.split_by_gradient([long_dead, long_alive])
    assert gr == [1] and fr == [0] and g[0] is long_alive and f[0] is long_dead
    assert T.split_by_gradient([row([2048], [0.0]), row([16384], [1.0])]) is None  # too little to take out
    monkeypatch.setenv("PRL_SPLIT_NO_GRADIENT", "0")
    assert T.split_by_gradient([long_dead, long_alive]) is None
