"""Throughput, MFU and HBM-roofline accounting for the trainer
(counterpart of src/prime_rl/trainer/perf.py:11-180, which knows NVIDIA parts
only and falls back to the A100 figure on anything else)."""

from __future__ import annotations

import time
from typing import Any

# Dense bf16 peak FLOP/s by a substring of the device name.  MI355X: 2.5 PFLOP/s dense
# (the 5 PFLOP/s headline figure assumes 2:1 structured sparsity).
PEAK_BF16_FLOPS = {
    "gfx950": 2.5e15,  # some driver stacks report MI355X only as "AMD Radeon Graphics"; the architecture string is reliable
    "MI355": 2.5e15,
    "MI350": 2.3e15,
    "MI325": 1.307e15,
    "MI300X": 1.307e15,
    "A100": 312e12,
    "H100 NVL": 835e12,
    "H100 PCIe": 756e12,
    "H100": 989e12,
    "H200": 989e12,
    "B200": 2.25e15,
}
# Peak HBM bandwidth in bytes/s by a substring of the device name (MI355X: 8 TB/s HBM3E).
PEAK_HBM_BYTES = {"gfx950": 8.0e12, "MI355": 8.0e12, "MI350": 8.0e12, "MI325": 6.0e12, "MI300X": 5.3e12}


def describe_device(device) -> str:
    """Marketing name plus architecture string of a torch device, the text the peak tables are matched against."""
    import torch

    props = torch.cuda.get_device_properties(device)
    return f"{props.name} {getattr(props, 'gcnArchName', '')}".strip()


def peak_bf16_flops(device_name: str) -> float:
    for key, val in PEAK_BF16_FLOPS.items():
        if key in device_name:
            return val
    raise KeyError(f"no bf16 peak known for device {device_name!r}; add it to PEAK_BF16_FLOPS")


def peak_hbm_bytes(device_name: str) -> float:
    for key, val in PEAK_HBM_BYTES.items():
        if key in device_name:
            return val
    raise KeyError(f"no HBM peak known for device {device_name!r}; add it to PEAK_HBM_BYTES")


def get_active_mm_params(config: Any) -> int:
    """Parameters per token that take part in matmuls: attention projections,
    MLP (dense, shared and routed experts, router) and the LM head
    (src/prime_rl/trainer/perf.py:80-139)."""
    h, layers = config.hidden_size, config.num_hidden_layers
    heads = config.num_attention_heads
    if hasattr(config, "q_lora_rank") and hasattr(config, "kv_lora_rank"):  # multi-head latent attention
        q = layers * (h * config.q_lora_rank + config.q_lora_rank * heads * config.qk_head_dim)
        kv = layers * (
            h * (config.kv_lora_rank + config.qk_rope_head_dim)
            + config.kv_lora_rank * heads * (config.qk_nope_head_dim + config.v_head_dim)
        )
        o = layers * heads * config.v_head_dim * h
    else:  # grouped-query attention; configs without an explicit head_dim (Qwen2) split the hidden size evenly
        head_dim = getattr(config, "head_dim", None) or h // heads
        q = layers * h * heads * head_dim
        kv = 2 * layers * h * config.num_key_value_heads * head_dim
        o = layers * h * heads * head_dim
    if hasattr(config, "first_k_dense_replace"):
        dense_layers = config.first_k_dense_replace
    elif hasattr(config, "num_experts_per_tok"):
        dense_layers = 0
    else:
        dense_layers = layers
    sparse_layers = layers - dense_layers
    mlp = dense_layers * 3 * config.intermediate_size * h
    sparse = 0
    if hasattr(config, "num_shared_experts"):
        sparse += sparse_layers * config.num_shared_experts * 3 * config.moe_intermediate_size * h
    if hasattr(config, "num_experts_per_tok"):
        sparse += sparse_layers * config.num_experts_per_tok * 3 * config.moe_intermediate_size * h
    if hasattr(config, "n_routed_experts"):
        sparse += sparse_layers * config.n_routed_experts * h
    elif hasattr(config, "num_experts"):
        sparse += sparse_layers * config.num_experts * h
    else:
        sparse = 0
    return q + kv + o + mlp + sparse + config.vocab_size * h


def flop_per_token(config: Any, seq_len: int) -> int:
    """6 x active matmul parameters + 12 x layers x heads x (hidden / heads) x seq_len
    for attention, causal sparsity not credited (src/prime_rl/trainer/perf.py:141-160)."""
    l, nh = config.num_hidden_layers, config.num_attention_heads  # noqa: E741
    return 6 * get_active_mm_params(config) + 12 * l * nh * (config.hidden_size // nh) * seq_len


def epilogue_bytes_per_token(vocab_size: int, logits_bytes: int, grad_bytes: int | None = None) -> int:
    """Bytes the loss epilogue must move per token: the logits are read once by the
    forward and once by the backward, the gradient is written once."""
    grad_bytes = logits_bytes if grad_bytes is None else grad_bytes
    return vocab_size * (2 * logits_bytes + grad_bytes)


class PerfCounter:
    """Rolling-window tokens/s and MFU (src/prime_rl/trainer/perf.py:11-53)."""

    def __init__(self, config: Any, seq_len: int, window_size: int = 10, device_name: str | None = None,
                 world_size: int = 1):
        self.window_size = window_size
        self.tokens: list[int] = []
        self.times: list[float] = []
        self.world_size = world_size
        self.num_flop_per_token = flop_per_token(config, seq_len)
        self.gpu_peak_flops = peak_bf16_flops(device_name) if device_name else 0.0

    def count_tokens(self, tokens: int) -> None:
        self.tokens.append(tokens)
        self.times.append(time.perf_counter())
        if len(self.tokens) > self.window_size:
            self.tokens.pop(0)
            self.times.pop(0)

    def get_tokens_per_second(self) -> float | None:
        if len(self.tokens) < 2:
            return None
        return sum(self.tokens[1:]) / (self.times[-1] - self.times[0])

    def get_mfu(self) -> float | None:
        tps = self.get_tokens_per_second()
        if tps is None or not self.gpu_peak_flops:
            return None
        return 100 * self.num_flop_per_token * tps / self.gpu_peak_flops / self.world_size
