"""The trainer step on a small random-init model: the HIP loss path under
autograd against the same model on the CPU with the oracle's loss."""

import copy

import pytest
import torch

import prime_rl_cse_amd as prl
from oracle import ref_cpu as O
from prime_rl_cse_amd import trainer as T

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def tiny_config():
    return T.model_config("Qwen/Qwen3-0.6B", num_layers=2, vocab_size=2048, hidden_size=64, intermediate_size=128,
                          num_attention_heads=4, num_key_value_heads=2, head_dim=16)


def rows(gen, n_rows, lens, vocab):
    out = []
    for _ in range(n_rows):
        t = sum(lens)
        out.append({
            "input_ids": torch.randint(0, vocab, (1, t), generator=gen),
            "position_ids": torch.cat([torch.arange(n) for n in lens]).unsqueeze(0),
            "advantages": torch.randn(1, t, generator=gen),
            "logprobs": -8.0 + 0.3 * torch.randn(1, t, generator=gen),
            "loss_mask": torch.rand(1, t, generator=gen) < 0.7,
            "temperature": 0.9,
        })
    return out


@pytest.mark.parametrize("ratio_type", ["token", "sequence"])
def test_train_step_matches_cpu_model_with_oracle_loss(ratio_type):
    cfg = tiny_config()
    cpu_model = T.get_model(cfg, "cpu", torch.float32, seed=3)
    gpu_model = copy.deepcopy(cpu_model).to(DEV)
    gen = torch.Generator().manual_seed(5)
    mbs = rows(gen, 3, [20, 12, 32], cfg.vocab_size)
    # a filler as the orchestrator appends them (batch.py:153-161): the reference runs it, this trainer leaves it
    # out and logs its loss of 0
    filler = copy.deepcopy(mbs[0])
    filler["advantages"], filler["loss_mask"] = torch.zeros_like(filler["advantages"]), torch.zeros_like(filler["loss_mask"])
    mbs.insert(2, filler)
    loss_cfg = prl.LossConfig(ratio_type=ratio_type, clip_ratio=4.0)

    # CPU: reference call order with the oracle
    scale = T.local_loss_scale(mbs, loss_cfg)
    cpu_losses, cpu_logged = [], {}
    for mb in mbs:
        logits = T.forward(cpu_model, mb["input_ids"], mb["position_ids"])
        x = O.shift_logits(logits.float().contiguous()) / mb["temperature"]
        lp = O.selective_log_softmax(x, mb["input_ids"])
        lens = O.get_response_lengths(mb["position_ids"])
        loss, logged = O.compute_loss(lp.squeeze().split(lens), mb["logprobs"].squeeze().split(lens),
                                      mb["advantages"].squeeze().split(lens), mb["loss_mask"].squeeze().split(lens),
                                      O.LossConfig(ratio_type, False, 4.0), scale)
        loss.backward()
        cpu_losses.append(loss.item())
        for k, v in logged.items():  # src/prime_rl/trainer/rl/train.py:327-333
            cpu_logged.setdefault(k, []).append(v.detach() if ratio_type == "sequence"
                                                else v.detach()[mb["loss_mask"].squeeze()])
    cpu_norm = torch.nn.utils.clip_grad_norm_(cpu_model.parameters(), 1e9).item()
    cpu_grads = {n: p.grad.clone() for n, p in cpu_model.named_parameters()}

    opt = torch.optim.SGD(gpu_model.parameters(), lr=0.0)
    res = T.train_step(gpu_model, opt, mbs, loss_cfg, max_norm=1e9, device=DEV)
    assert abs(res.grad_norm - cpu_norm) <= 2e-4 * cpu_norm
    assert abs(res.stats["loss/mean"] - sum(cpu_losses) / len(cpu_losses)) <= 1e-4
    # token mode leaves the filler out; with one ratio per sample the filler's ratios are logged, so it runs
    assert res.tokens == (3 if ratio_type == "token" else 4) * 64 and res.samples == 4 and cpu_losses[2] == 0.0
    for k, parts in cpu_logged.items():
        v = torch.cat([p.float().reshape(-1) for p in parts])
        assert res.stats[f"{k}/mean"] == pytest.approx(v.mean().item(), rel=1e-4, abs=1e-6), k
        assert res.stats[f"{k}/max"] == pytest.approx(v.max().item(), rel=1e-4, abs=1e-6), k
    assert res.passes == (1 if ratio_type == "token" else 4)

    # gradients themselves (recomputed, because train_step clears them)
    for mb in mbs:
        d = prl.to_device(mb, DEV)
        logits = T.forward(gpu_model, d["input_ids"], d["position_ids"])
        loss, _ = prl.policy_loss(logits, d["input_ids"], d["position_ids"], d["advantages"], d["logprobs"],
                                  d["loss_mask"], d["temperature"], loss_cfg, scale)
        loss.backward()
    for n, p in gpu_model.named_parameters():
        ref = cpu_grads[n]
        assert torch.allclose(p.grad.cpu(), ref, rtol=2e-3, atol=2e-4 * ref.abs().max().item() + 1e-9), n


def test_checkpointed_blocks_take_the_same_attention_path_when_they_run_again():
    """`[model.ac]` (src/prime_rl/trainer/model.py:195-214): blocks re-run their forward inside backward, where the
    packed-row attention layout must still be in force; the step is the one of the model without checkpointing."""
    from torch.distributed.algorithms._checkpoint.checkpoint_wrapper import checkpoint_wrapper

    cfg = tiny_config()
    gen = torch.Generator().manual_seed(15)
    mbs = rows(gen, 2, [20, 12, 32], cfg.vocab_size) + rows(gen, 1, [16, 16, 16, 16], cfg.vocab_size)
    results = []
    for ac in (False, True):
        model = T.get_model(cfg, DEV, torch.float32, seed=3)
        if ac:
            for name, block in list(model.model.layers.named_children()):
                model.model.layers.register_module(name, checkpoint_wrapper(block, preserve_rng_state=False))
        opt = torch.optim.SGD(model.parameters(), lr=0.0)
        for join in (0, 16384):
            results.append(T.train_step(model, opt, copy.deepcopy(mbs), prl.LossConfig(), max_norm=1e9, device=DEV,
                                        join_tokens=join))
    for plain, checkpointed in zip(results[:2], results[2:]):
        assert checkpointed.grad_norm == pytest.approx(plain.grad_norm, rel=1e-5)
        for k in plain.stats:
            assert checkpointed.stats[k] == pytest.approx(plain.stats[k], rel=1e-5, abs=1e-7), k


def test_bf16_model_step_runs_and_learns_direction():
    """bf16 parameters (what FSDP puts on the wire): one SGD step along the
    gradient lowers the loss on the same batch."""
    cfg = tiny_config()
    model = T.get_model(cfg, DEV, torch.bfloat16, seed=4)
    gen = torch.Generator().manual_seed(6)
    mbs = rows(gen, 2, [40, 24], cfg.vocab_size)
    for mb in mbs:
        mb["loss_mask"] = torch.ones_like(mb["loss_mask"])
        mb["logprobs"] = torch.full_like(mb["logprobs"], -7.6)
    opt = torch.optim.SGD(model.parameters(), lr=0.05)
    before = T.train_step(model, opt, mbs, prl.LossConfig(), max_norm=1e9, device=DEV)
    after = T.train_step(model, opt, mbs, prl.LossConfig(), max_norm=1e9, device=DEV)
    assert after.stats["loss/mean"] < before.stats["loss/mean"]
    # no log-probability model: the reference logs the error of a tensor of ones (train.py:215, 316-318)
    assert before.stats["recomputed_logprob_error/mean"] == 1.0 and before.stats["recomputed_logprob_error/std"] == 0.0
    assert set(k.split("/")[0] for k in before.stats) >= {"probs", "old_probs", "entropy", "loss", "importance_ratio",
                                                         "clipped_importance_ratio", "is_clipped"}


def test_recompute_logprobs_matches_reference_chain():
    """src/prime_rl/trainer/rl/train.py:236-245 with the oracle on the CPU copy of the model."""
    cfg = tiny_config()
    cpu_model = T.get_model(cfg, "cpu", torch.float32, seed=8)
    gpu_model = copy.deepcopy(cpu_model).to(DEV)
    gen = torch.Generator().manual_seed(9)
    mbs = rows(gen, 2, [16, 24], cfg.vocab_size)
    filler = copy.deepcopy(mbs[0])  # not scored: keeps its log-probabilities, its (masked) error reads 1
    filler["loss_mask"] = torch.zeros_like(filler["loss_mask"])
    kept = filler["logprobs"].clone()
    want_lp, want_err = [], []
    with torch.no_grad():
        for mb in mbs:
            logits = T.forward(cpu_model, mb["input_ids"], mb["position_ids"]).float().contiguous()
            lp = O.selective_log_softmax(O.shift_logits(logits) / mb["temperature"], mb["input_ids"])
            want_lp.append(lp)
            want_err.append(torch.exp(lp - mb["logprobs"]))
    given = [mbs[0], filler, mbs[1]]
    errs = T.recompute_logprobs(gpu_model, given, DEV)
    assert len(errs) == 3 and torch.equal(filler["logprobs"], kept) and torch.equal(errs[1].cpu(), torch.ones_like(kept))
    for mb, lp, err, got in zip(mbs, want_lp, want_err, [errs[0], errs[2]]):
        assert mb["logprobs"].device.type == "cpu"
        assert torch.allclose(mb["logprobs"], lp, atol=1e-4, rtol=1e-4)
        assert torch.allclose(got.cpu(), err, atol=1e-4, rtol=1e-3)


def _single_rank_env(monkeypatch, port):
    from prime_rl_cse_amd.world import reset_world

    for k, v in dict(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
                     LOCAL_WORLD_SIZE="1").items():
        monkeypatch.setenv(k, v)
    reset_world()


def test_trainer_entry_point_on_fake_data(tmp_path, monkeypatch):
    """The counterpart of the reference's `trainer @ configs/debug/rl/train.toml` smoke run
    (tests/integration/test_trainer.py:10-19): bench mode, fake data, weights published per step."""
    from prime_rl_cse_amd.settings import RLTrainerConfig, parse_argv
    from prime_rl_cse_amd.train import train

    _single_rank_env(monkeypatch, 29761)
    toml = tmp_path / "train.toml"
    toml.write_text("max_steps = 5\nasync_level = 5\n\n[data.fake]\nbatch_size = 8\nmicro_batch_size = 2\nseq_len = 64\n"
                    "\n[scheduler]\ntype = \"linear\"\nwarmup_steps = 2\ndecay_steps = 0\n")
    cfg = parse_argv(RLTrainerConfig, ["@", str(toml), "--bench", "--model.debug.num-layers", "2", "--output-dir",
                                       str(tmp_path / "out"), "--recompute-logprobs"], use_env=False)
    history = train(cfg)
    assert len(history) == 4 and all(h["time/step"] > 0 for h in history)  # bench: 1 warm-up + 3 measured steps
    assert {"loss/mean", "entropy/mean", "importance_ratio/mean", "recomputed_logprob_error/mean", "optim/grad_norm",
            "perf/throughput", "perf/throughput_per_gpu", "time/wait_for_batch"} <= set(history[-1])
    # the rate a step was taken with, not the one the scheduler has moved on to (train.py:346-347 of the reference)
    lrs = [h["optim/lr"] for h in history]
    assert lrs[0] < lrs[1] < lrs[2] == pytest.approx(1e-6) and lrs[3] == pytest.approx(1e-6)
    for step in (1, 2, 3, 4):
        assert (tmp_path / "out" / "weights" / f"step_{step}" / "pytorch_model.bin").exists()
    assert not list((tmp_path / "out").rglob("*.tmp"))
    state = torch.load(tmp_path / "out" / "weights" / "step_4" / "pytorch_model.bin", weights_only=True)
    assert state["model.embed_tokens.weight"].dtype == torch.bfloat16
    assert state["model.embed_tokens.weight"].shape == (151936, 1024)


def test_trainer_entry_point_with_model_compile_takes_whole_model_mode(tmp_path, monkeypatch):
    """`[model.compile]` selects the whole-model-per-GPU mode (its passes run eagerly: graph replay is off, a
    captured pass holds the memset nodes of library calls); the run logs what the FSDP2 run of the same
    configuration logs."""
    monkeypatch.delenv("PRL_GRAPHS", raising=False)
    from prime_rl_cse_amd.settings import RLTrainerConfig, parse_argv
    from prime_rl_cse_amd.train import train

    base = ("max_steps = 3\nasync_level = 3\nrecompute_logprobs = true\n\n[data.fake]\nbatch_size = 8\n"
            "micro_batch_size = 2\nseq_len = 64\n\n[model]\nimpl = \"liger_kernel\"\n\n[model.debug]\nnum_layers = 2\n")
    runs = {}
    for name, extra, port in (("fsdp", "", 29771), ("graphs", "\n[model.compile]\nfullgraph = false\n", 29772)):
        _single_rank_env(monkeypatch, port)
        monkeypatch.setenv("PRL_TRAINER", "fsdp" if name == "fsdp" else "")
        toml = tmp_path / f"{name}.toml"
        toml.write_text(base + extra)
        cfg = parse_argv(RLTrainerConfig, ["@", str(toml), "--output-dir", str(tmp_path / name)], use_env=False)
        assert (cfg.model.compile is not None) == (name == "graphs")
        torch.manual_seed(0)  # the fake loader draws from the global generator
        runs[name] = train(cfg)
    for a, b in zip(runs["fsdp"], runs["graphs"]):
        for key in ("loss/mean", "entropy/mean", "probs/mean", "importance_ratio/mean", "optim/grad_norm",
                    "recomputed_logprob_error/mean"):
            # the loss is a sum of terms of either sign that nearly cancel: compare it on the scale of its terms
            assert b[key] == pytest.approx(a[key], rel=2e-2, abs=5e-4 if key == "loss/mean" else 1e-6), key
    state = torch.load(tmp_path / "graphs" / "weights" / "step_3" / "pytorch_model.bin", weights_only=True)
    want = torch.load(tmp_path / "fsdp" / "weights" / "step_3" / "pytorch_model.bin", weights_only=True)
    assert state.keys() == want.keys()
    for k in want:
        assert state[k].dtype == want[k].dtype and state[k].shape == want[k].shape


def test_trainer_entry_point_on_rollout_files(tmp_path, monkeypatch):
    """Fed by `rollouts/step_N/rank_0.pt` files as the orchestrator writes them
    (model: tests/integration/test_trainer_path.py:18-82 of the reference)."""
    from prime_rl_cse_amd.settings import RLTrainerConfig
    from prime_rl_cse_amd.train import train

    _single_rank_env(monkeypatch, 29762)
    gen = torch.Generator().manual_seed(3)
    for step in range(2):
        rollouts = []
        for pid in range(6):
            p, c = int(torch.randint(4, 20, (1,), generator=gen)), int(torch.randint(4, 40, (1,), generator=gen))
            rollouts.append(prl.Rollout(pid, torch.randint(0, 1000, (p,), generator=gen).tolist(), [0] * p,
                                        torch.randint(0, 1000, (c,), generator=gen).tolist(), [1] * c,
                                        (-11.9 + 0.05 * torch.randn(c, generator=gen)).tolist(), False, 1.0,  # near ln V of a random model
                                        float(torch.randn(1, generator=gen))))
        prl.save_rollout_batches(prl.prepare_batch(rollouts, 1.0, None, 6, 2, 64, 1), tmp_path, step)
    cfg = RLTrainerConfig(max_steps=2, output_dir=tmp_path, model={"debug": {"num_layers": 1}},
                          loss={"ratio_type": "sequence"})
    history = train(cfg)
    assert len(history) == 2 and history[0]["optim/grad_norm"] > 0
    # one ratio per sample, logged for every sample (src/prime_rl/trainer/rl/train.py:327-333)
    assert 0.0 < history[0]["importance_ratio/min"] <= history[0]["importance_ratio/mean"] < 8.0
    assert history[0]["is_clipped/max"] in (0.0, 1.0)
    assert (tmp_path / "weights" / "step_2" / "pytorch_model.bin").exists()


def test_trainer_entry_point_on_the_reference_tests_dummy_batches(tmp_path, monkeypatch):
    """The reference's trainer-path test with its own dummy data (tests/integration/test_trainer_path.py:18-82):
    files written with plain torch.save in the orchestrator's layout, 16 samples of 16 tokens in micro-batches of
    8, position ids that are ALL zero (attention then keeps every token to itself, the loss sees one sample per
    row), then `trainer @ configs/debug/rl/train.toml --data.fake None`: five steps without an error."""
    from prime_rl_cse_amd.settings import RLTrainerConfig, parse_argv
    from prime_rl_cse_amd.train import train

    _single_rank_env(monkeypatch, 29764)
    gen = torch.Generator().manual_seed(11)
    steps, batch_size, micro_batch_size, seq_len = 5, 16, 8, 16
    for step in range(steps):
        batches = []
        for _ in range(batch_size // micro_batch_size):
            n = micro_batch_size * seq_len
            batches.append({
                "input_ids": torch.randint(0, 100, (1, n), generator=gen),
                "position_ids": torch.zeros(1, n, dtype=torch.long),
                "advantages": torch.randn(1, n, generator=gen),
                "loss_mask": torch.ones(1, n, dtype=torch.bool),
                "logprobs": torch.randn(1, n, generator=gen),
                "temperature": 1.0,
                "total_tokens": n,
            })
        step_dir = tmp_path / "rollouts" / f"step_{step}"
        step_dir.mkdir(parents=True)
        torch.save(batches, step_dir / "rank_0.tmp")
        (step_dir / "rank_0.tmp").rename(step_dir / "rank_0.pt")
    toml = tmp_path / "train.toml"  # configs/debug/rl/train.toml of the reference
    toml.write_text("max_steps = 5\nasync_level = 5\n\n[data.fake]\nbatch_size = 16\nmicro_batch_size = 1\nseq_len = 128\n")
    cfg = parse_argv(RLTrainerConfig, ["@", str(toml), "--output-dir", str(tmp_path), "--data.fake", "None",
                                       "--log.level", "debug", "--model.debug.num-layers", "2"], use_env=False)
    assert cfg.data.fake is None
    history = train(cfg)
    assert len(history) == steps
    for h in history:
        assert h["optim/grad_norm"] > 0 and h["loss/mean"] == h["loss/mean"]  # finite, not NaN
        assert h["entropy/mean"] > 0


def test_rollout_files_through_both_trainers_agree(tmp_path, monkeypatch):
    """Ragged rollouts (prompts masked, completions of many lengths, several rows per step) from
    `rollouts/step_N/rank_0.pt` through the FSDP2 trainer and through the whole-model trainer, fused
    blocks on in both: the same logged numbers step by step."""
    monkeypatch.delenv("PRL_GRAPHS", raising=False)
    from prime_rl_cse_amd.settings import RLTrainerConfig
    from prime_rl_cse_amd.train import train

    gen = torch.Generator().manual_seed(17)
    for step in range(3):
        rollouts = []
        for pid in range(24):
            p, c = int(torch.randint(3, 30, (1,), generator=gen)), int(torch.randint(1, 90, (1,), generator=gen))
            rollouts.append(prl.Rollout(pid, torch.randint(0, 1000, (p,), generator=gen).tolist(), [0] * p,
                                        torch.randint(0, 1000, (c,), generator=gen).tolist(), [1] * c,
                                        (-11.9 + 0.05 * torch.randn(c, generator=gen)).tolist(), False, 1.0,
                                        float(torch.randn(1, generator=gen))))
        batches = prl.prepare_batch(rollouts, 1.0, None, 24, 4, 128, 1)
        for root in ("fsdp", "graphs"):
            prl.save_rollout_batches(batches, tmp_path / root, step)
    runs = {}
    for name, port in (("fsdp", 29781), ("graphs", 29782)):
        _single_rank_env(monkeypatch, port)
        model = {"debug": {"num_layers": 2}, "impl": "liger_kernel"}
        if name == "graphs":
            model["compile"] = {}
        else:
            model["reshard_after_forward"] = True  # a recipe that cares about FSDP2's handling gets FSDP2
        cfg = RLTrainerConfig(max_steps=3, output_dir=tmp_path / name, model=model)
        runs[name] = train(cfg)
    assert len(runs["fsdp"]) == len(runs["graphs"]) == 3
    for a, b in zip(runs["fsdp"], runs["graphs"]):
        assert a["optim/grad_norm"] > 0
        for key in ("loss/mean", "loss/max", "entropy/mean", "probs/mean", "importance_ratio/mean",
                    "importance_ratio/max", "is_clipped/mean", "optim/grad_norm"):
            assert b[key] == pytest.approx(a[key], rel=2e-2, abs=5e-4 if key.startswith("loss") else 1e-6), key


def test_statistics_exchange_runs_over_rccl(monkeypatch):
    """The between-pass all-reduces of `Tensors` (int32 histograms, int64 counts, float64 moments, int32
    extrema with MAX) through the RCCL backend.  One rank is all this box has, so the group has one member and
    the package is told the world is larger: every collective is issued for real and must leave the numbers
    of the single-rank path unchanged."""
    import torch.distributed as dist

    from prime_rl_cse_amd import utils as U

    _single_rank_env(monkeypatch, 29763)
    dist.init_process_group("nccl", device_id=torch.device("cuda:0"))
    try:
        gen = torch.Generator().manual_seed(5)
        xs = [torch.randn(5000, generator=gen).to("cuda:0") for _ in range(3)]
        ms = [(torch.rand(5000, generator=gen) < 0.6).to("cuda:0") for _ in range(3)]

        def run():
            t = prl.Tensors()
            for x, m in zip(xs, ms):
                t.add("a", x, m)
                t.add("b", x, m, exp=True)
                t.add("c", x[:7])
            return t.compute_stats()

        alone = run()

        class LargerWorld:
            def __getattr__(self, name):
                return getattr(dist, name)

            def get_world_size(self, *a, **k):
                return 2

        monkeypatch.setattr(U, "dist", LargerWorld())
        for stream_choice in ("auto", "0"):  # with ranks to talk to, "auto" is a stream of the statistics' own
            monkeypatch.setenv("PRL_STATS_STREAM", stream_choice)
            exchanged = run()
            assert exchanged.keys() == alone.keys()
            for k in alone:
                assert exchanged[k] == alone[k], (stream_choice, k)
        assert U._STATS_STREAMS  # the stream was really made
    finally:
        dist.barrier()
        dist.destroy_process_group()


def test_logprob_model_serves_the_weights_it_was_last_given(monkeypatch):
    """The forward-only model of the recompute pass gets new weights every step (train.py:205-214 of the
    reference); what it computes must follow them even though nothing ever runs a backward through it."""
    import torch.distributed as dist

    from prime_rl_cse_amd import trainer as T
    from prime_rl_cse_amd.settings import RLTrainerConfig
    from prime_rl_cse_amd.train import build_model

    _single_rank_env(monkeypatch, 29764)
    dev = torch.device("cuda:0")
    mesh = T.init_distributed(dev)
    try:
        cfg = RLTrainerConfig(model={"debug": {"num_layers": 1}})
        model, _ = build_model(cfg, dev, mesh, dtype=torch.bfloat16, for_training=False)
        model.requires_grad_(False)
        mb = prl.FakeDataLoader(prl.FakeDataLoaderConfig(batch_size=2, micro_batch_size=2, seq_len=16)).get_batch()[0]
        first = T.recompute_logprobs(model, [dict(mb)], dev)[0].clone()
        state = {k: (v.detach().clone() * 0.5) for k, v in model.state_dict().items()}
        model.load_state_dict(state)
        second = T.recompute_logprobs(model, [dict(mb)], dev)[0]
        assert not torch.equal(first, second)
    finally:
        dist.barrier()
        dist.destroy_process_group()


def test_once_per_step_collectives_give_the_per_micro_batch_gradients(monkeypatch):
    """Parameters kept gathered and gradients reduce-scattered once per step (this package's arrangement for
    models that fit) against resharding after every forward and reducing after every backward (the
    reference's): same gradient norm, same updated weights."""
    import torch.distributed as dist

    _single_rank_env(monkeypatch, 29765)
    dev = torch.device(DEV)
    mesh = T.init_distributed(dev)
    try:
        cfg = tiny_config()
        base = T.get_model(cfg, dev, torch.float32, seed=11)
        gen = torch.Generator().manual_seed(6)
        mbs = rows(gen, 4, [20, 12, 32], cfg.vocab_size)
        out = {}
        for name, reshard in (("per_micro_batch", True), ("once_per_step", None)):
            model = copy.deepcopy(base)
            T.setup_fsdp(model, mesh, reshard_after_forward=reshard, blocks_per_group=1 if reshard else None)
            assert model._prl_keep_gathered == (reshard is None)
            opt = torch.optim.SGD(model.parameters(), lr=0.5)
            res = T.train_step(model, opt, mbs, prl.LossConfig(), max_norm=1e9, device=dev)
            res2 = T.train_step(model, opt, mbs, prl.LossConfig(), max_norm=1e9, device=dev)  # weights must be fresh
            out[name] = (res.grad_norm, res2.grad_norm,
                         {k: v.full_tensor().clone() if hasattr(v, "full_tensor") else v.clone()
                          for k, v in model.state_dict().items()})
        a, b = out["per_micro_batch"], out["once_per_step"]
        assert a[0] > 0 and abs(a[0] - b[0]) <= 1e-3 * a[0]
        assert abs(a[1] - b[1]) <= 2e-3 * a[1] and abs(a[1] - a[0]) > 1e-6 * a[0]  # the second step saw new weights
        for k in a[2]:
            assert torch.allclose(a[2][k], b[2][k], rtol=1e-3, atol=1e-5), k
    finally:
        dist.barrier()
        dist.destroy_process_group()


def test_gathered_lm_head_rows_in_the_plain_trainer_step(monkeypatch):
    """`trainer.train_step` on a bf16 model whose rows have their prompts masked out: with the LM head run only
    where the next token is scored the step's statistics and gradients are those of the run over every position
    (to bf16 rounding of products of another shape); an fp32 model never gathers."""
    import prime_rl_cse_amd.trainer as TT

    cfg = tiny_config()
    gen = torch.Generator().manual_seed(8)
    mbs = rows(gen, 3, [40, 24, 64], cfg.vocab_size)
    for mb in mbs:
        mb["loss_mask"] = mb["loss_mask"] & (mb["position_ids"] >= 16)
    seen = []
    real = TT.policy_loss

    def spy(logits, *args, **kw):
        seen.append(None if kw.get("positions") is None else int(kw["positions"].numel()))
        return real(logits, *args, **kw)

    monkeypatch.setattr(TT, "policy_loss", spy)
    out = []
    for gather, dtype in (("0", torch.bfloat16), ("1", torch.bfloat16), ("1", torch.float32)):
        monkeypatch.setenv("PRL_GATHER_SCORED", gather)
        model = T.get_model(cfg, DEV, dtype, seed=4)
        captured = {}
        hooks = [p.register_post_accumulate_grad_hook(lambda q, n=n: captured.__setitem__(n, q.grad.float().clone()))
                 for n, p in model.named_parameters()]
        res = T.train_step(model, torch.optim.SGD(model.parameters(), lr=0.0), copy.deepcopy(mbs), prl.LossConfig(),
                           max_norm=1e9, device=DEV)
        for h in hooks:
            h.remove()
        out.append((res, captured))
    assert seen == [None, 256, None]  # 384 tokens joined, rounded up to 512; 256 gathered rows
    (r0, g0), (r1, g1), _ = out
    assert r1.grad_norm == pytest.approx(r0.grad_norm, rel=2e-2)
    for k in r0.stats:
        assert r1.stats[k] == pytest.approx(r0.stats[k], rel=5e-3, abs=1e-5), k
    for n in g0:
        assert (g1[n] - g0[n]).norm().item() <= 3e-2 * g0[n].norm().item() + 1e-9, n


def test_zero_advantage_samples_run_forward_only_in_the_plain_trainer_step(monkeypatch):
    """`trainer.train_step`: samples with advantage 0 are taken out of the backward pass; gradients, gradient norm
    and statistics are those of the step that runs every row forward and backward."""
    cfg = tiny_config()
    gen = torch.Generator().manual_seed(9)
    mbs = rows(gen, 3, [32, 32, 32, 32], cfg.vocab_size)
    for r, mb in enumerate(mbs):
        mb["loss_mask"] = mb["loss_mask"] & (mb["position_ids"] >= 8)
        adv = torch.randn(4, generator=gen).repeat_interleave(32)
        for s in ((1, 3), (0, 1, 2, 3), ())[r]:
            adv[32 * s : 32 * (s + 1)] = 0.0
        mb["advantages"] = adv.unsqueeze(0)
    out = []
    for split in ("0", "1"):  # ("1": also for steps as short as these)
        monkeypatch.setenv("PRL_SPLIT_NO_GRADIENT", split)
        model = T.get_model(cfg, DEV, torch.float32, seed=4)
        captured = {}
        hooks = [p.register_post_accumulate_grad_hook(lambda q, n=n: captured.__setitem__(n, q.grad.clone()))
                 for n, p in model.named_parameters()]
        res = T.train_step(model, torch.optim.SGD(model.parameters(), lr=0.0), copy.deepcopy(mbs), prl.LossConfig(),
                           max_norm=1e9, device=DEV)
        for h in hooks:
            h.remove()
        out.append((res, captured))
    (r0, g0), (r1, g1) = out
    assert (r0.passes, r1.passes) == (1, 2) and r0.tokens == r1.tokens == 384 and r0.samples == r1.samples == 3
    assert r1.grad_norm == pytest.approx(r0.grad_norm, rel=2e-5)
    assert r1.stats.keys() == r0.stats.keys()
    for k in r0.stats:
        assert r1.stats[k] == pytest.approx(r0.stats[k], rel=2e-5, abs=1e-6), k
    for n in g0:
        torch.testing.assert_close(g1[n], g0[n], rtol=1e-4, atol=1e-6 * float(g0[n].abs().max()) + 1e-12)
