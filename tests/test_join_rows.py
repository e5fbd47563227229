"""Joining a step's micro-batches into longer rows (host logic of trainer.join_micro_batches)."""

import torch

from prime_rl_cse_amd import trainer as T


def row(n, temperature=1.0, seed=0):
    gen = torch.Generator().manual_seed(seed + n)
    return {
        "input_ids": torch.randint(0, 50, (1, n), generator=gen),
        "position_ids": torch.arange(n)[None],
        "advantages": torch.randn(1, n, generator=gen),
        "logprobs": torch.randn(1, n, generator=gen),
        "loss_mask": torch.ones(1, n, dtype=torch.bool),
        "temperature": temperature,
    }


def test_rows_are_joined_up_to_the_token_limit_and_only_at_one_temperature():
    mbs = [row(10), row(20), row(30), row(25, temperature=0.6), row(5, temperature=0.6), row(7)]
    joined = T.join_micro_batches(mbs, max_tokens=40)
    assert [r.lengths for r in joined] == [[10, 20], [30], [25, 5], [7]]
    assert [r.starts for r in joined] == [[0, 10], [0], [0, 25], [0]]
    first = joined[0]
    assert first.micro_batch["input_ids"].shape == (1, 30)
    assert torch.equal(first.micro_batch["input_ids"], torch.cat([mbs[0]["input_ids"], mbs[1]["input_ids"]], dim=1))
    assert first.micro_batch["position_ids"][0, 10] == 0  # the second row restarts: attention keeps them apart
    # the first token of every row but the first is masked out of the joined row, not out of the original mask
    assert first.loss_mask.all() and not first.micro_batch["loss_mask"][0, 10]
    assert int(first.micro_batch["loss_mask"].sum()) == 29 and mbs[1]["loss_mask"].all()
    assert joined[2].micro_batch["temperature"] == 0.6
    # nothing to join: the micro-batch itself, mask untouched
    assert joined[1].micro_batch["loss_mask"].all() and joined[1].micro_batch["input_ids"] is not None


def test_rows_with_a_batch_dimension_stay_alone_and_pieces_follow_the_rows():
    two = row(8)
    two = {k: (torch.cat([v, v]) if isinstance(v, torch.Tensor) else v) for k, v in two.items()}
    mbs = [row(4), two, row(5), row(6)]
    joined = T.join_micro_batches(mbs, max_tokens=100)
    assert [r.lengths for r in joined] == [[4], [8], [5, 6]]
    pieces = [torch.full((1, n), float(i)) for i, n in enumerate((4, 8, 5, 6))]
    out = T.join_per_row(pieces, joined)
    assert [p.shape[1] for p in out] == [4, 8, 11] and out[2][0, 4] == 2.0 and out[2][0, 5] == 3.0
    assert T.join_micro_batches([], 10) == []


def test_fillers_are_recognised_by_their_mask_alone():
    """A filler is a copy of the first micro-batch with an all-False mask (orchestrator/batch.py:153-161); a
    micro-batch with a single unmasked token is not one; with nothing to score at all, every micro-batch stays."""
    def mb(mask):
        t = len(mask)
        return {"input_ids": torch.zeros(1, t, dtype=torch.int64), "position_ids": torch.arange(t).unsqueeze(0),
                "advantages": torch.zeros(1, t), "logprobs": torch.zeros(1, t),
                "loss_mask": torch.tensor([mask], dtype=torch.bool), "temperature": 1.0}

    batch = [mb([True, False, True]), mb([False, False, False]), mb([False, False, True]), mb([False] * 5)]
    assert T.scored(batch) == [0, 2]
    assert T.scored([batch[1], batch[3]]) == [0, 1]
    assert T.scored([]) == []


def test_joined_rows_are_rounded_up_with_tokens_nobody_scores():
    def mb(n, seed):
        g = torch.Generator().manual_seed(seed)
        return {"input_ids": torch.randint(0, 100, (1, n), generator=g), "position_ids": torch.arange(n).unsqueeze(0),
                "advantages": torch.randn(1, n, generator=g), "logprobs": torch.randn(1, n, generator=g),
                "loss_mask": torch.rand(1, n, generator=g) < 0.5, "temperature": 1.0}

    batch = [mb(300, 1), mb(211, 2)]
    (row,) = T.join_micro_batches(batch, 16384, pad_to=512)
    assert row.padding == 1 and row.micro_batch["input_ids"].shape == (1, 512) and row.lengths == [300, 211]
    (row,) = T.join_micro_batches(batch, 16384, pad_to=384)
    t = row.micro_batch["input_ids"].shape[1]
    assert t == 768 and row.padding == 257 and row.loss_mask.shape == (1, 768)
    assert not row.micro_batch["loss_mask"][0, 511:].any() and not row.loss_mask[0, 511:].any()
    assert (row.micro_batch["advantages"][0, 511:] == 0).all()
    pos = row.micro_batch["position_ids"][0, 511:]
    assert pos[0] == 0 and int(pos.max()) == 63  # its own samples, short ones
    assert torch.equal(row.micro_batch["input_ids"][0, :511], torch.cat([b["input_ids"][0] for b in batch]))
    # per-row pieces follow
    (piece,) = T.join_per_row([torch.zeros(1, 300), torch.zeros(1, 211)], [row])
    assert piece.shape == (1, 768)
    # no room, no rounding; a multiple already, nothing added; 0 switches it off
    assert T.join_micro_batches(batch, 511, pad_to=512)[0].padding == 0
    assert T.join_micro_batches([mb(512, 3)], 16384, pad_to=512)[0].padding == 0
    assert T.join_micro_batches(batch, 16384, pad_to=0)[0].padding == 0
