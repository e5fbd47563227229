"""Known answers of the reference's FLOP accounting (tests/unit/train/test_perf.py:8-11)
from literal model dimensions (no hub access), and the MI355X peaks."""

from types import SimpleNamespace

import pytest

from prime_rl_cse_amd import perf

QWEN3_0_6B = SimpleNamespace(vocab_size=151936, hidden_size=1024, intermediate_size=3072, head_dim=128,
                             num_attention_heads=16, num_key_value_heads=8, num_hidden_layers=28)


def test_qwen3_0_6b_known_answers():
    assert perf.get_active_mm_params(QWEN3_0_6B) == 595_984_384
    assert perf.flop_per_token(QWEN3_0_6B, 1024) == 3_928_227_840


def test_every_known_model_has_a_flop_count():
    # Qwen2 configs carry no head_dim: the hidden size split over the heads stands in (7B: 3584 / 28 = 128)
    from prime_rl_cse_amd import trainer as T

    for name in T.MODEL_DIMS:
        cfg = T.model_config(name)
        assert perf.get_active_mm_params(cfg) > 0 and perf.flop_per_token(cfg, 128) > 6 * perf.get_active_mm_params(cfg)
    assert perf.get_active_mm_params(T.model_config("Qwen/Qwen2.5-7B")) == 7070285824


def test_mi355x_peaks_and_unknown_devices():
    assert perf.peak_bf16_flops("AMD Instinct MI355X") == 2.5e15
    assert perf.peak_bf16_flops("AMD Radeon Graphics gfx950:sramecc+:xnack-") == 2.5e15
    assert perf.peak_hbm_bytes("AMD Instinct MI355X") == 8.0e12
    assert perf.peak_bf16_flops("NVIDIA H100 80GB HBM3") == 989e12
    with pytest.raises(KeyError):
        perf.peak_bf16_flops("Unknown GPU")  # no silent A100 fallback


def test_epilogue_bytes():
    # fp32 logits and gradient: 12 V bytes per token; bf16: 6 V
    assert perf.epilogue_bytes_per_token(151936, 4) == 12 * 151936 == 1_823_232
    assert perf.epilogue_bytes_per_token(151936, 2) == 6 * 151936


def test_counter_window():
    c = perf.PerfCounter(QWEN3_0_6B, 1024, window_size=3, device_name="AMD Instinct MI355X", world_size=8)
    assert c.get_tokens_per_second() is None and c.get_mfu() is None
    for _ in range(5):
        c.count_tokens(1000)
    assert len(c.tokens) == 3 and c.get_tokens_per_second() > 0 and c.get_mfu() > 0
