"""The trainer's configuration surface: TOML recipes, `@ file` syntax, dotted CLI overrides, inheritance,
environment, validators (model: tests/unit/test_config.py:51-112 of the reference, which loads every recipe)."""

from pathlib import Path

import pytest

from prime_rl_cse_amd.settings import RLTrainerConfig, load_config, parse_argv

REFERENCE_CONFIGS = Path("/root/reference")


def write(tmp_path, name, text):
    p = tmp_path / name
    p.write_text(text)
    return p


def test_defaults_match_reference():
    c = RLTrainerConfig()
    assert c.model.name == "Qwen/Qwen3-0.6B" and c.model.attn == "flash_attention_2" and c.model.impl == "hf"
    assert c.model.optimization_dtype == "float32" and c.model.reduce_dtype == "float32"
    assert (c.optim.type, c.optim.lr, c.optim.weight_decay, c.optim.max_norm) == ("adamw", 1e-6, 0.01, 1.0)
    assert (c.optim.betas1, c.optim.betas2) == (0.9, 0.999) and c.scheduler.type == "constant"
    assert (c.loss.ratio_type, c.loss.clip_ratio) == ("token", 8.0)
    assert c.async_level == 2 and c.max_steps is None and c.ckpt is None and c.weights.save_async
    assert c.output_dir == Path("outputs") and c.dist_timeout_seconds == 600 and not c.recompute_logprobs


def test_recipe_shaped_toml_with_at_syntax(tmp_path):
    f = write(tmp_path, "train.toml", """
max_steps = 20
async_level = 5

[model]
name = "PrimeIntellect/Qwen3-0.6B-Reverse-Text-SFT"

[optim]
lr = 3e-6

[data.fake]
batch_size = 16
micro_batch_size = 1
seq_len = 128

[loss]
ratio_type = "sequence"
clip_ratio = 4.0
""")
    c = parse_argv(RLTrainerConfig, ["@", str(f), "--optim.max-norm", "0.5", "--model.attn=sdpa", "--recompute_logprobs"],
                   use_env=False)
    assert c.max_steps == 20 and c.async_level == 5 and c.model.name.endswith("Reverse-Text-SFT")
    assert c.optim.lr == 3e-6 and c.optim.max_norm == 0.5 and c.model.attn == "sdpa" and c.recompute_logprobs
    assert c.data.fake.batch_size == 16 and c.loss.ratio_type == "sequence" and c.loss.clip_ratio == 4.0
    # `@file` glued form and a file loaded under a key
    g = write(tmp_path, "optim.toml", 'type = "sgd"\nlr = 0.1\n')
    c = parse_argv(RLTrainerConfig, [f"@{f}", "--optim", "@", str(g)], use_env=False)
    assert c.optim.type == "sgd" and c.optim.lr == 0.1 and c.optim.nesterov and c.max_steps == 20


def test_inheritance_and_precedence(tmp_path, monkeypatch):
    write(tmp_path, "base.toml", 'max_steps = 7\nasync_level = 1\n[optim]\nlr = 1e-5\n')
    main = write(tmp_path, "main.toml", 'toml_files = ["base.toml"]\nmax_steps = 3\n[optim]\nweight_decay = 0.1\n')
    c = load_config(RLTrainerConfig, main)
    # the files a file names take precedence over it; untouched keys survive the merge
    assert c.max_steps == 7 and c.async_level == 1 and c.optim.lr == 1e-5 and c.optim.weight_decay == 0.1
    monkeypatch.setenv("PRIME_ASYNC_LEVEL", "9")
    monkeypatch.setenv("PRIME_OPTIM__LR", "0.5")
    monkeypatch.setenv("PRIME_UNRELATED_THING", "x")
    c = parse_argv(RLTrainerConfig, ["@", str(main), "--max-steps", "11"])
    assert c.max_steps == 11 and c.async_level == 1 and c.optim.lr == 1e-5  # CLI > TOML > environment
    c = parse_argv(RLTrainerConfig, [])
    assert c.async_level == 9 and c.optim.lr == 0.5


def test_booleans_none_strings_and_optional_sections(tmp_path):
    c = parse_argv(RLTrainerConfig, ["--bench"], use_env=False)
    assert c.bench and c.max_steps == 4 and c.data.fake is not None and c.ckpt is None
    c = parse_argv(RLTrainerConfig, ["--ckpt", "--ckpt.interval", "5", "--weights.no-save-async"], use_env=False)
    assert c.ckpt.interval == 5 and c.weights.save_async is False
    f = write(tmp_path, "n.toml", '[data]\nfake = "None"\n[wandb]\nproject = "x"\n')
    c = load_config(RLTrainerConfig, f)
    assert c.data.fake is None and c.wandb.project == "x" and c.wandb.log_extras.samples is False


def test_validation_errors(tmp_path):
    with pytest.raises(Exception):
        parse_argv(RLTrainerConfig, ["--not-a-field", "1"], use_env=False)
    with pytest.raises(Exception):
        parse_argv(RLTrainerConfig, ["--optim.type", "adagrad"], use_env=False)
    with pytest.raises(Exception, match="max_steps"):
        RLTrainerConfig(trace_path="t")
    with pytest.raises(Exception, match="massive"):
        RLTrainerConfig(trace_path="t", max_steps=10)
    with pytest.raises(Exception, match="meta"):
        RLTrainerConfig(model={"debug": {"random_init": True}})
    with pytest.raises(FileNotFoundError):
        parse_argv(RLTrainerConfig, ["@", str(tmp_path / "missing.toml")], use_env=False)


@pytest.mark.skipif(not REFERENCE_CONFIGS.exists(), reason="the reference's recipes exist in the build container only")
def test_every_reference_rl_trainer_recipe_loads():
    files = sorted(REFERENCE_CONFIGS.glob("configs/**/rl/train.toml")) + sorted(
        REFERENCE_CONFIGS.glob("examples/**/rl/train.toml"))
    assert len(files) >= 15
    for f in files:
        c = load_config(RLTrainerConfig, f)
        assert c.model.name
        if "A3B" not in c.model.name:   # the two mixture-of-experts recipes need their weights' own config
            from prime_rl_cse_amd.train import model_config_of
            assert model_config_of(c).vocab_size >= 151936


def test_trainer_mode_and_block_kernels_follow_the_recipe(monkeypatch):
    from prime_rl_cse_amd.settings import RLTrainerConfig
    from prime_rl_cse_amd.train import fused_blocks, trainer_mode

    monkeypatch.delenv("PRL_TRAINER", raising=False)
    monkeypatch.delenv("PRL_BLOCKS", raising=False)
    plain = RLTrainerConfig()
    assert trainer_mode(plain) == "auto" and fused_blocks(plain)
    assert trainer_mode(RLTrainerConfig(model={"compile": {}})) == "replica"
    from prime_rl_cse_amd.train import replay_graphs

    monkeypatch.delenv("PRL_GRAPHS", raising=False)
    # the recipe's switch picks the whole-model trainer but never graph replay
    assert not replay_graphs(RLTrainerConfig(model={"compile": {}})) and not replay_graphs(RLTrainerConfig())
    monkeypatch.setenv("PRL_GRAPHS", "0")
    assert not replay_graphs(RLTrainerConfig(model={"compile": {}}))
    monkeypatch.setenv("PRL_GRAPHS", "1")
    assert replay_graphs(RLTrainerConfig())
    monkeypatch.delenv("PRL_GRAPHS")
    assert trainer_mode(RLTrainerConfig(model={"reshard_after_forward": True})) == "fsdp"
    assert not fused_blocks(RLTrainerConfig(model={"impl": "hf"}))
    assert fused_blocks(RLTrainerConfig(model={"impl": "liger_kernel"}))
    monkeypatch.setenv("PRL_TRAINER", "fsdp")
    monkeypatch.setenv("PRL_BLOCKS", "eager")
    assert trainer_mode(RLTrainerConfig(model={"compile": {}})) == "fsdp" and not fused_blocks(plain)


def test_recipe_model_names_resolve_to_known_architectures(tmp_path):
    """The names in the reference's recipes (configs/**/train.toml) build a config without the hub: from a local
    checkpoint when there is one, else from literal dimensions of the same architecture."""
    import pytest

    from prime_rl_cse_amd import trainer as T
    from prime_rl_cse_amd.settings import RLTrainerConfig
    from prime_rl_cse_amd.train import model_config_of

    cfg = model_config_of(RLTrainerConfig(model={"name": "PrimeIntellect/Qwen3-0.6B-Reverse-Text-SFT"}))
    assert (cfg.hidden_size, cfg.num_hidden_layers, cfg.vocab_size, cfg.head_dim) == (1024, 28, 151936, 128)
    cfg = model_config_of(RLTrainerConfig(model={"name": "willcb/DeepSeek-R1-Distill-Qwen-7B", "debug": {"num_layers": 3}}))
    assert (cfg.hidden_size, cfg.num_hidden_layers, cfg.vocab_size) == (3584, 3, 152064)
    assert all(T.known_architecture(n) in T.MODEL_DIMS for n in T.MODEL_ALIASES)
    with pytest.raises(ValueError, match="no weights on this machine"):
        model_config_of(RLTrainerConfig(model={"name": "nobody/unknown-model"}))
    # a directory in HF layout wins over the literal dimensions
    T.model_config("Qwen/Qwen3-0.6B", num_layers=2, hidden_size=64, intermediate_size=128, num_attention_heads=4,
                   num_key_value_heads=2, head_dim=16, vocab_size=512).save_pretrained(tmp_path)
    assert T.local_checkpoint(str(tmp_path)) == str(tmp_path)
    cfg = model_config_of(RLTrainerConfig(model={"name": str(tmp_path)}))
    assert (cfg.hidden_size, cfg.vocab_size, cfg.num_hidden_layers) == (64, 512, 2)
