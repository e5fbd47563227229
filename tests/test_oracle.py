"""The CPU oracle against the committed golden vectors (made from the
reference's own functions by oracle/make_golden.py) and against the known
answers the reference states in its docstrings."""

import numpy as np
import pytest
import torch

from oracle import ref_cpu as O
from tests.helpers import epilogue_cases, full_vocabulary_cases, load_npz, GOLDEN


def same(a, b):
    a, b = torch.as_tensor(a), torch.as_tensor(b)
    if a.is_floating_point():
        return bool(((a == b) | (a.isnan() & b.isnan())).all())
    return bool((a == b).all())


@pytest.fixture(autouse=True)
def _one_thread():
    # the vectors were made with one thread; summation order follows the thread count
    n = torch.get_num_threads()
    torch.set_num_threads(1)
    yield
    torch.set_num_threads(n)


@pytest.mark.parametrize("name", epilogue_cases() + full_vocabulary_cases())
def test_epilogue_matches_reference_vectors(name):
    d = load_npz(name)
    cfg = O.LossConfig(d["ratio_type"], bool(d["ratio_length_norm"]), float(d["clip_ratio"]))
    b = lambda x: x.unsqueeze(0)
    out = O.policy_epilogue(
        b(d["logits"]), b(d["input_ids"]), b(d["position_ids"]), b(d["advantages"]), b(d["old_logprobs"]),
        b(d["loss_mask"]), float(d["temperature"]), cfg, d["loss_scale"],
    )
    assert out["lengths"] == d["exp_lengths"].tolist()
    assert same(out["logprobs"][0], d["exp_logprobs"])
    assert same(out["entropy"][0], d["exp_entropy"])
    assert same(out["loss"], torch.as_tensor(d["exp_loss"]))
    for k in ("importance_ratio", "clipped_importance_ratio", "is_clipped"):
        assert same(out[k], d["exp_" + k]), k
    assert same(out["dlogits"][0], d["exp_dlogits"])


@pytest.mark.parametrize("name", full_vocabulary_cases())
def test_full_vocabulary_vectors_say_how_far_the_reference_is_from_float64(name):
    """The float64 side of the fixture is what the generating script says it is (recomputed here from the
    stored logits), and the reference's fp32 entropy is 1e-4 .. 4e-4 away from it at this size while its
    log-probabilities are within 1e-5: the entropy bound of the GPU test rests on these two facts."""
    d = load_npz(name)
    # the fp32 quotient the reference's entropy sees (train.py:285-287), then everything in float64
    x = (O.shift_logits(d["logits"].float().unsqueeze(0)) / float(d["temperature"])).double()
    ent = (torch.logsumexp(x, -1) - (torch.softmax(x, -1) * x).sum(-1))[0]
    assert torch.allclose(ent, d["f64_entropy"], rtol=0, atol=1e-12)
    err = (d["exp_entropy"].double() - d["f64_entropy"]).abs().max().item()
    assert 1e-4 < err < 4e-4
    assert (d["exp_logprobs"].double() - d["f64_logprobs"]).abs().max().item() < 1e-5


@pytest.mark.parametrize("name", sorted(p.name for p in GOLDEN.glob("pieces_*.npz")))
def test_pieces(name):
    d = load_npz(name)
    x = d["logits"].unsqueeze(0)
    idx = d["index"].unsqueeze(0)
    xs = O.shift_logits(x) / float(d["temperature"])
    assert same(O.selective_log_softmax(xs, idx)[0], d["exp_logprobs"])
    assert same(O.selective_log_softmax(xs.to(torch.bfloat16), idx)[0].float(), d["exp_logprobs_bf16in"])
    assert same(O.compute_entropy(xs)[0], d["exp_entropy"])


@pytest.mark.parametrize("name", sorted(p.name for p in GOLDEN.glob("loss_*.npz")))
def test_loss_alone(name):
    d = load_npz(name)
    lengths = d["lengths"].tolist()
    lp = [x.clone().requires_grad_(True) for x in d["logprobs"].split(lengths)]
    cfg = O.LossConfig(d["ratio_type"], bool(d["ratio_length_norm"]), float(d["clip_ratio"]))
    loss, aux = O.compute_loss(
        lp, d["old_logprobs"].split(lengths), d["advantages"].split(lengths), d["loss_mask"].split(lengths), cfg,
        float(d["loss_scale"]),
    )
    loss.backward()
    g = torch.cat([x.grad if x.grad is not None else torch.zeros_like(x) for x in lp])
    assert same(loss.detach(), torch.as_tensor(d["exp_loss"]))
    assert same(g, d["exp_grad_logprobs"])
    for k in ("importance_ratio", "clipped_importance_ratio", "is_clipped"):
        assert same(aux[k].detach(), d["exp_" + k])


def test_response_lengths_vectors():
    d = load_npz("response_lengths.npz")
    names = sorted(k[: -len("__pos")] for k in d if k.endswith("__pos"))
    assert len(names) > 50
    for n in names:
        assert O.get_response_lengths(d[n + "__pos"].unsqueeze(0)) == d[n + "__len"].tolist(), n


def test_tensor_stats_vectors():
    """`Tensors.compute_stats` of the reference (src/prime_rl/trainer/utils.py:235-256, one rank) on stored
    values: the oracle's statistics are the reference's, bit for bit (NaN where the reference has NaN)."""
    d = load_npz("tensor_stats.npz")
    names = sorted(k[: -len("__x")] for k in d if k.endswith("__x"))
    assert len(names) >= 9
    for n in names:
        got = O.tensor_stats(d[n + "__x"].float())
        for stat, want in zip(("mean", "median", "std", "min", "max"), d[n + "__stats"].tolist()):
            assert got[stat] == want or (got[stat] != got[stat] and want != want), (n, stat, got[stat], want)


def test_advantage_vectors_and_docstring_answers():
    # src/prime_rl/orchestrator/advantage.py:13-16 and :27-30
    assert O.compute_advantages([0.0, 0.0, 1.0, 1.0], [1] * 4, 4, "drgrpo") == [-0.5, -0.5, 0.5, 0.5]
    assert O.compute_advantages([0.0] * 4, [1] * 4, 4, "drgrpo") == [0.0] * 4
    assert O.compute_advantages([1.0] * 4, [1] * 4, 4, "drgrpo") == [0.0] * 4
    assert O.compute_advantages([0.0, 0.0, 1.0, 1.0], [1] * 4, 4, "drgrpo-negclipped") == [0.0, 0.0, 0.5, 0.5]
    d = load_npz("advantages.npz")
    keys = sorted(k[: -len("__adv")] for k in d if k.endswith("__adv"))
    assert len(keys) == 28
    for k in keys:
        got = O.compute_advantages(d[k + "__rewards"].tolist(), d[k + "__lengths"].tolist(), int(d[k + "__g"]),
                                   k.split("__")[1])
        assert same(torch.tensor(got, dtype=torch.float64), d[k + "__adv"]), k


def test_batch_vectors():
    d = load_npz("batches.npz")
    for case in range(4):
        n_roll, mbs, seq_len = d[f"c{case}__meta"].tolist()
        rollouts = []
        for j in range(n_roll):
            p = d[f"c{case}__r{j}__prompt"].tolist()
            rollouts.append(
                O.Rollout(j, p, [0] * len(p), d[f"c{case}__r{j}__completion"].tolist(),
                          d[f"c{case}__r{j}__cmask"].tolist(), d[f"c{case}__r{j}__clogp"].tolist(), False, 0.0,
                          float(d[f"c{case}__r{j}__adv"]))
            )
        for w in (1, 2, 4):
            got = O.prepare_batch(rollouts, 0.7, n_roll, mbs, seq_len, w)
            assert len(got) == w
            for rank in range(w):
                assert len(got[rank]) == int(d[f"c{case}__w{w}__rank{rank}__rows"])
                for m, row in enumerate(got[rank]):
                    for k in O.TOKEN_KEYS:
                        assert same(row[k], d[f"c{case}__w{w}__rank{rank}__mb{m}__{k}"]), (case, w, rank, m, k)


def test_extra_fillers_when_row_count_divides():
    # W - n % W fillers: a whole extra W when n already divides (src/prime_rl/orchestrator/batch.py:153)
    r = [O.Rollout(i, [1, 2], [0, 0], [3, 4], [1, 1], [-0.1, -0.2], False, 1.0, 0.5) for i in range(4)]
    got = O.prepare_batch(r, 1.0, 4, 1, 4, 2)  # 4 rows of 4 tokens, 2 ranks -> 2 fillers
    assert [len(x) for x in got] == [3, 3]
    assert not got[1][-1]["loss_mask"].any() and (got[1][-1]["advantages"] == 0).all()
