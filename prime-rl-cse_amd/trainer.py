"""The RL trainer's step on MI355X: PyTorch-ROCm runs the transformer body,
FSDP2 shards it over RCCL, and everything after the LM head is the HIP path of
this package (counterpart of src/prime_rl/trainer/rl/train.py:259-364 and of the
model set-up in src/prime_rl/trainer/model.py:55-135, :251-254).

What differs from the reference loop, on purpose:
  * the model's bf16 logits go straight into `policy_loss` (no fp32 copy, no
    shifted copy, no divided copy, no softmax tensor);
  * sample boundaries, loss and statistics never leave the device inside the
    loop: the only host synchronisations of a step are the gradient-norm read
    and the final 5-floats-per-key statistics copy;
  * statistics over ranks are all-reduced histograms, not gathered vectors.
"""

from __future__ import annotations

import time
from dataclasses import dataclass, field
from typing import Any

import torch
import torch.distributed as dist
from torch import Tensor, nn

from .batch import host_to_device, to_device
from .config import LossConfig
from .data import MicroBatch
from .loss import policy_loss, unit_gradient
from .utils import Tensors

# Literal architecture dimensions (there is no hub access where this runs).  The
# Qwen3-0.6B row is confirmed by the reference's own FLOP known-answer test
# (tests/unit/train/test_perf.py:10); the others are the public model cards'.
MODEL_DIMS: dict[str, dict[str, Any]] = {
    "Qwen/Qwen3-0.6B": dict(family="qwen3", vocab_size=151936, hidden_size=1024, intermediate_size=3072,
                            num_hidden_layers=28, num_attention_heads=16, num_key_value_heads=8, head_dim=128,
                            tie_word_embeddings=True, rope_theta=1e6, rms_norm_eps=1e-6, max_position_embeddings=40960),
    "Qwen/Qwen3-1.7B": dict(family="qwen3", vocab_size=151936, hidden_size=2048, intermediate_size=6144,
                            num_hidden_layers=28, num_attention_heads=16, num_key_value_heads=8, head_dim=128,
                            tie_word_embeddings=True, rope_theta=1e6, rms_norm_eps=1e-6, max_position_embeddings=40960),
    "Qwen/Qwen2.5-0.5B": dict(family="qwen2", vocab_size=151936, hidden_size=896, intermediate_size=4864,
                              num_hidden_layers=24, num_attention_heads=14, num_key_value_heads=2,
                              tie_word_embeddings=True, rope_theta=1e6, rms_norm_eps=1e-6,
                              max_position_embeddings=32768),
    "Qwen/Qwen2.5-7B": dict(family="qwen2", vocab_size=152064, hidden_size=3584, intermediate_size=18944,
                            num_hidden_layers=28, num_attention_heads=28, num_key_value_heads=4,
                            tie_word_embeddings=False, rope_theta=1e6, rms_norm_eps=1e-6,
                            max_position_embeddings=131072),
}


MODEL_DIMS.update({
    # public model cards' dimensions for the other architectures the reference's recipes train
    "Qwen/Qwen2.5-1.5B": dict(family="qwen2", vocab_size=151936, hidden_size=1536, intermediate_size=8960,
                              num_hidden_layers=28, num_attention_heads=12, num_key_value_heads=2,
                              tie_word_embeddings=True, rope_theta=1e6, rms_norm_eps=1e-6,
                              max_position_embeddings=131072),
    "Qwen/Qwen2.5-3B": dict(family="qwen2", vocab_size=151936, hidden_size=2048, intermediate_size=11008,
                            num_hidden_layers=36, num_attention_heads=16, num_key_value_heads=2,
                            tie_word_embeddings=True, rope_theta=1e6, rms_norm_eps=1e-6,
                            max_position_embeddings=32768),
    "Qwen/Qwen2.5-14B": dict(family="qwen2", vocab_size=152064, hidden_size=5120, intermediate_size=13824,
                             num_hidden_layers=48, num_attention_heads=40, num_key_value_heads=8,
                             tie_word_embeddings=False, rope_theta=1e6, rms_norm_eps=1e-5,
                             max_position_embeddings=131072),
    "Qwen/Qwen2.5-32B": dict(family="qwen2", vocab_size=152064, hidden_size=5120, intermediate_size=27648,
                             num_hidden_layers=64, num_attention_heads=40, num_key_value_heads=8,
                             tie_word_embeddings=False, rope_theta=1e6, rms_norm_eps=1e-5,
                             max_position_embeddings=131072),
})

# Names the reference's recipes use (configs/**/train.toml, examples/**) for fine-tunes of these architectures.
# They only matter when no weights are found on the machine and a random initialisation of the same shape
# stands in (benchmarks, smoke runs); with weights present the checkpoint's own config is read.
MODEL_ALIASES: dict[str, str] = {
    "PrimeIntellect/Qwen3-0.6B": "Qwen/Qwen3-0.6B",
    "PrimeIntellect/Qwen3-0.6B-Reverse-Text-SFT": "Qwen/Qwen3-0.6B",
    "PrimeIntellect/Qwen3-1.7B": "Qwen/Qwen3-1.7B",
    "PrimeIntellect/Qwen3-1.7B-Wordle-SFT": "Qwen/Qwen3-1.7B",
    "deepseek-ai/DeepSeek-R1-Distill-Qwen-1.5B": "Qwen/Qwen2.5-1.5B",
    "willcb/DeepSeek-R1-Distill-Qwen-1.5B": "Qwen/Qwen2.5-1.5B",
    "deepseek-ai/DeepSeek-R1-Distill-Qwen-7B": "Qwen/Qwen2.5-7B",
    "willcb/DeepSeek-R1-Distill-Qwen-7B": "Qwen/Qwen2.5-7B",
    "willcb/DeepSeek-R1-Distill-Qwen-14B": "Qwen/Qwen2.5-14B",
    "willcb/DeepSeek-R1-Distill-Qwen-32B": "Qwen/Qwen2.5-32B",
}


def known_architecture(name: str) -> str | None:
    """The key of `MODEL_DIMS` that has the architecture of model `name`, or None."""
    for candidate in (name, name.removesuffix("-Instruct"), name.removesuffix("-Base")):   # same shapes
        if candidate in MODEL_DIMS:
            return candidate
    return MODEL_ALIASES.get(name)


def local_checkpoint(name: str) -> str | None:
    """Where the weights of `name` are on this machine: the directory itself, or its snapshot in the Hugging Face
    cache (never the network: `local_files_only`), or None."""
    from pathlib import Path

    if Path(name).is_dir():
        return name
    try:
        from huggingface_hub import snapshot_download

        return snapshot_download(name, local_files_only=True)
    except Exception:
        return None


def model_config(name: str, attn: str = "sdpa", num_layers: int | None = None, **overrides: Any):
    """A transformers config from literal dimensions; `num_layers` trims the depth
    (the reference's `model.debug.num_layers`, src/prime_rl/trainer/model.py:64-68)."""
    from transformers import Qwen2Config, Qwen3Config

    dims = dict(MODEL_DIMS[known_architecture(name) or name])
    dims.update(overrides)
    family = dims.pop("family")
    if num_layers is not None:
        dims["num_hidden_layers"] = min(num_layers, dims["num_hidden_layers"])
    cfg = (Qwen3Config if family == "qwen3" else Qwen2Config)(**dims)
    cfg._attn_implementation = attn
    cfg.use_cache = False
    return cfg


def get_model(cfg: Any, device: torch.device | str, dtype: torch.dtype = torch.float32, seed: int = 0) -> nn.Module:
    """A randomly initialised causal LM in `dtype` on `device` (the counterpart of
    the reference's `random_init` debug path, src/prime_rl/trainer/model.py:144-146)."""
    from transformers import AutoModelForCausalLM

    torch.manual_seed(seed)
    with torch.device(device):
        model = AutoModelForCausalLM.from_config(cfg, dtype=dtype)
    return model


def blocks_per_fsdp_group(model: nn.Module, target_bytes: int = 1 << 30) -> int:
    """How many transformer blocks share one FSDP group so that a group's bf16 parameters come to about
    `target_bytes`.  The reference makes one group per block (src/prime_rl/trainer/model.py:106-113), sized
    for 80 GB parts; with 288 GB of HBM per GPU and point-to-point xGMI links, fewer and larger all-gathers /
    reduce-scatters use the links better and cost the host fewer hooks: Qwen3-0.6B (31 MB per block) becomes
    one group of 28 blocks, a 7B model two blocks per group, a 32B model one."""
    blocks = list(model.model.layers)
    per_block = sum(p.numel() for p in blocks[0].parameters()) * 2
    return max(1, min(len(blocks), target_bytes // max(per_block, 1)))


KEEP_GATHERED_BYTES = 32 << 30  # bf16 parameters up to this size stay gathered for a whole optimizer step


def keeps_parameters_gathered(model: nn.Module) -> bool:
    """Whether the whole bf16 model is small enough (32 GB of the 288 GB) to stay gathered from the first
    micro-batch of a step to the last.  Then a step costs ONE all-gather and ONE reduce-scatter per FSDP group
    instead of two and one per micro-batch (the reference reshards after every forward and reduces after every
    backward, src/prime_rl/trainer/model.py:110-134, src/prime_rl/trainer/rl/train.py:271-310): with the
    gradients accumulated unsharded in fp32 the sums are the same, and xGMI carries 1/8 .. 1/16 of the bytes."""
    return sum(p.numel() for p in model.parameters()) * 2 <= KEEP_GATHERED_BYTES


def setup_fsdp(model: nn.Module, mesh: Any, reduce_dtype: torch.dtype = torch.float32,
               reshard_after_forward: bool | None = None, blocks_per_group: int | None = None) -> nn.Module:
    """FSDP2 over a 1-D data-parallel mesh as the reference sets it up
    (src/prime_rl/trainer/model.py:102-135): bf16 parameters on the wire, fp32
    gradient reduction, transformer blocks in groups (`blocks_per_group`, default
    `blocks_per_fsdp_group`; 1 reproduces the reference's one group per block);
    for untied models the embedding and [lm_head, final norm] get groups of their own."""
    from torch.distributed.fsdp import MixedPrecisionPolicy, fully_shard

    if reshard_after_forward is None:  # not asked for either way: keep small models gathered
        reshard_after_forward = not keeps_parameters_gathered(model)
    model._prl_keep_gathered = not reshard_after_forward
    model._prl_compute_dtype = torch.bfloat16  # what the blocks and the LM head compute in (the policy below)
    mp = MixedPrecisionPolicy(param_dtype=torch.bfloat16, reduce_dtype=reduce_dtype)
    blocks = list(model.model.layers)
    per = blocks_per_group if blocks_per_group else blocks_per_fsdp_group(model)
    for a in range(0, len(blocks), per):
        group = blocks[a : a + per]
        fully_shard(group if len(group) > 1 else group[0], mesh=mesh, mp_policy=mp,
                    reshard_after_forward=reshard_after_forward)
    if not model.config.tie_word_embeddings:
        fully_shard(model.model.embed_tokens, mesh=mesh, mp_policy=mp, reshard_after_forward=reshard_after_forward)
        fully_shard([model.lm_head, model.model.norm], mesh=mesh, mp_policy=mp, reshard_after_forward=False)
    fully_shard(model, mesh=mesh, mp_policy=mp, reshard_after_forward=reshard_after_forward)
    return model


def forward(model: nn.Module, input_ids: Tensor | None, position_ids: Tensor, plan: Any = None,
            inputs_embeds: Tensor | None = None, positions: Tensor | None = None) -> Tensor:
    """Logits [batch, seq, vocab] of a packed row (src/prime_rl/trainer/model.py:251-254).  With a
    `attention.PackedPlan` of that row, attention runs over the samples side by side instead of over the
    whole row under a [T, T] mask; the logits are the same.  `inputs_embeds` (the rows of the embedding table,
    already looked up) takes the place of `input_ids`.  `positions` (int64 [R], on the device): the LM head runs
    at these positions only and the logits are [batch, R, vocab] (`scored_rows`)."""
    tokens: dict[str, Any] = {"input_ids": input_ids} if inputs_embeds is None else {"inputs_embeds": inputs_embeds}
    if positions is not None:
        tokens["logits_to_keep"] = positions  # transformers' causal LMs index the hidden states with it
    if plan is None:
        return model(position_ids=position_ids, **tokens).logits
    from . import attention

    attention.install()
    with attention.use_plan(plan):
        if _maskless_enabled() and attention.maskless(plan, model):  # no [T, T] mask to build: nothing reads it
            tokens["attention_mask"] = {"full_attention": None}
        return model(position_ids=position_ids, **tokens).logits


def _maskless_enabled() -> bool:
    """PRL_ATTN_MASKLESS=0 lets transformers build its mask for every forward (for A/B measurements)."""
    import os

    return os.environ.get("PRL_ATTN_MASKLESS", "1") != "0"


def plan_scope(plan: Any) -> Any:
    """`with plan_scope(plan):` keeps `plan` in force for everything inside, backward included.  Activation
    checkpointing re-runs a block's forward during backward, and the re-run must take the same attention path
    (and save the same tensors) as the first run did inside `forward`."""
    import contextlib

    if plan is None:
        return contextlib.nullcontext()
    from . import attention

    attention.install()
    return attention.use_plan(plan)


def attention_plan(host_micro_batch: MicroBatch, device: torch.device | str) -> Any:
    """The side-by-side attention layout of one micro-batch, from its position ids while they are still on the
    host (no device synchronisation); None when there is nothing to gain."""
    from . import attention

    pos = host_micro_batch["position_ids"]
    if pos.is_cuda:
        return None
    return attention.plan_from_position_ids(pos, device)


@dataclass
class StepResult:
    loss: float
    grad_norm: float
    stats: dict[str, float]
    seconds: float
    tokens: int
    samples: int
    timings: dict[str, float] = field(default_factory=dict)
    passes: int = 0  # forward / backward passes the step took (fewer than micro-batches when rows were joined)


def local_loss_scale(micro_batches: list[MicroBatch], loss_config: LossConfig) -> int:
    """The normaliser of src/prime_rl/trainer/rl/train.py:263-267: unmasked tokens of
    THIS rank's micro-batches in token mode, rows x row batch size in sequence
    mode.  The masks are still on the host here, so counting them costs no
    device synchronisation."""
    if loss_config.ratio_type == "token":
        return int(sum(int(mb["loss_mask"].sum()) for mb in micro_batches))
    return int(micro_batches[0]["input_ids"].shape[0] * len(micro_batches))


def add_ratio_statistics(tensors: Tensors, aux: dict[str, Tensor], mask: Tensor, loss_config: LossConfig) -> None:
    """The three logged tensors of `compute_loss` into `tensors` as the reference selects them
    (src/prime_rl/trainer/rl/train.py:327-333): per token under the loss mask, or, with one ratio per sample,
    every sample's value.  `policy_loss` leaves room for T // 2 + 1 samples and fills the first `n_seg`, a count
    that stays on the device: the selection is made there too."""
    keys = ("importance_ratio", "clipped_importance_ratio", "is_clipped")
    if loss_config.ratio_type == "sequence":
        n = aux[keys[0]].numel()
        mask = torch.arange(n, device=aux[keys[0]].device) < aux["n_seg"].reshape(())
    for key in keys:
        tensors.add(key, aux[key], mask)


def ones_as_recomputed_error(stats: dict[str, float]) -> None:
    """Without a log-probability model the reference still logs `recomputed_logprob_error`: of a tensor of ones
    under the loss mask (src/prime_rl/trainer/rl/train.py:215, 316-318).  Those five numbers need no kernel."""
    if "probs/mean" in stats:  # at least one unmasked token, as for every other masked key
        for name, value in (("mean", 1.0), ("median", 1.0), ("std", 0.0), ("min", 1.0), ("max", 1.0)):
            stats.setdefault(f"recomputed_logprob_error/{name}", value)


def scored(micro_batches: list[MicroBatch]) -> list[int]:
    """Indices of the micro-batches that can change anything: those with at least one unmasked token.

    The orchestrator pads every step with filler micro-batches, copies of the first one with advantage 0 and an
    all-False mask (src/prime_rl/orchestrator/batch.py:153-161: with W > 1 ranks `W - n % W` of them, so a whole
    extra round when the count already divides), because FSDP's collectives need the same number of passes on every rank.  A
    micro-batch without an unmasked token has loss 0 and gradient 0, and every logged statistic is taken under the
    mask (src/prime_rl/trainer/rl/train.py:313-333), so where the number of passes is a rank's own business such a
    pass only costs time: the caller runs the others and logs a loss of 0 for each one left out.  The masks are
    still on the host here.  If no micro-batch has an unmasked token, all are kept (a step needs a pass)."""
    keep = [i for i, mb in enumerate(micro_batches) if bool(mb["loss_mask"].any())]
    return keep if keep else list(range(len(micro_batches)))


GATHER_MULTIPLE = 256  # the number of gathered rows is rounded up to a multiple of this
GATHER_AT_MOST = 0.9  # gather only when at most this share of a row's positions is left


def gather_rule(tokens: int, hidden: int, library_products: bool = False) -> tuple[int, float]:
    """(multiple the number of gathered rows is rounded up to, largest share of a row's positions worth gathering).

    On this package's LM-head kernels (`lm_head.plan`; the default) the three products cost the same per row at
    every row count (0.705-0.74 ms per 1024 rows at H = 1024, R = 8192 .. 16384 in steps of 512,
    `tools/lm_head_rows_sweep.py`), so the plain rule holds.  With the products on the BLAS library
    (`library_products`: PRL_LM_HEAD=lib, or shapes the kernels do not take) long rows of narrow models get a
    coarser rule: the input gradient, [R, V] x [V, H], has few output tiles and a very long sum, and with H = 1024 or
    2048 the library serves it at 1.2-1.3 PFLOP/s when R is 8192, 10240, 12288 or 16384 and at 0.6-0.9 at every
    other multiple of 256 (14336 included), so that 11776 rows cost more than 12288 and 14336 as much as 16384; with
    H = 3584 the library's cost follows R at every size."""
    if library_products and tokens > 8192 and 0 < hidden <= 2048:
        return 2048, 0.76
    return GATHER_MULTIPLE, GATHER_AT_MOST


def scored_rows(host_micro_batch: MicroBatch, loss_config: LossConfig, model: nn.Module,
                device: torch.device | str) -> Tensor | None:
    """The positions of a packed row at which the LM head has to run, on the device, or None for "all of them".

    The reference runs the LM head, the soft-max and their backward over every position of a row
    (src/prime_rl/trainer/model.py:251-254, src/prime_rl/trainer/rl/train.py:285-310) and then uses the results
    only under the loss mask (loss.py:93, :104; train.py:313-315); prompts are masked out
    (src/prime_rl/orchestrator/batch.py:21-62), so a good part of a real row is computed for nothing.  A position
    is needed iff the NEXT token is scored; the others are left out of the LM head's three products (forward,
    input gradient, weight gradient) and of the pass over the logits.  Loss, gradients and logged statistics are
    unchanged.  The mask is still on the host here, so picking the positions costs no synchronisation.

    Only in token mode (with one ratio per sample the reference logs unmasked samples' ratios too), for a model
    that computes bf16 logits over a vocabulary the one-pass row kernel handles, and only when at most
    `GATHER_AT_MOST` of the positions is left (the count is rounded up to a multiple of `GATHER_MULTIPLE` with
    positions nobody scores, so that the LM head's products meet few distinct sizes; `gather_rule` has the
    numbers for long rows of narrow models).  PRL_GATHER_SCORED=0 switches it off."""
    import os

    from . import _capi as C
    from .loss import scored_positions

    if os.environ.get("PRL_GATHER_SCORED", "1") == "0" or loss_config.ratio_type != "token":
        return None
    mask = host_micro_batch["loss_mask"]
    if mask.is_cuda or mask.ndim != 2 or mask.shape[0] != 1 or mask.shape[1] < 2:
        return None
    dtype = getattr(model, "_prl_compute_dtype", None)
    if dtype is None:
        dtype = next(model.parameters()).dtype
    vocab = int(getattr(getattr(model, "config", None), "vocab_size", 0))
    if dtype != torch.bfloat16 or vocab < 8 or \
            not C.load().prl_policy_rows_supported(C.PRL_BF16, 1, vocab, vocab, C.PRL_BF16, vocab):
        return None
    import numpy as np

    from . import lm_head

    hidden = int(getattr(getattr(model, "config", None), "hidden_size", 0))
    on_kernels = hidden > 0 and hidden % 64 == 0 and lm_head.plan(GATHER_MULTIPLE, hidden, vocab)[1]
    multiple, at_most = gather_rule(int(mask.shape[1]), hidden, library_products=not on_kernels)
    n_scored = int(np.count_nonzero(mask.numpy()[0, 1:]))  # (cheap, and settles the rows where every token is scored)
    if n_scored == 0 or n_scored > at_most * (mask.shape[1] - 1):
        return None
    keep = scored_positions(mask, multiple=multiple)
    if keep.numel() > at_most * (mask.shape[1] - 1):
        return None
    return host_to_device(keep, device, torch.int64)


def split_by_gradient(micro_batches: list[MicroBatch]
                      ) -> tuple[list[MicroBatch], list[int], list[MicroBatch], list[int]] | None:
    """The step's rows sorted into what needs a backward pass and what needs a forward pass only.

    A token can only have a gradient if it is scored AND its advantage is not zero
    (src/prime_rl/trainer/rl/loss.py:98-104: the loss term is -clipped_ratio * advantage under the mask).  With
    group-relative advantages (src/prime_rl/orchestrator/advantage.py:18) every sample of a group whose rollouts
    all earned the same reward has advantage 0, so whole samples contribute exact zeros to the loss and to every
    gradient, and the reference still runs them through forward AND backward.  Their log-probabilities, entropies
    and ratios are logged (train.py:313-333), so they need the forward pass, and nothing else: two thirds of their
    cost goes away.

    Returns (rows to train on, the given row each came from, rows for a forward pass only, the given row each came
    from), or None when there is nothing to separate.  A row whose samples are all of one kind stays whole.  A
    mixed row is cut into its samples (the attention's samples: position ids restart) only if no sample's first
    token is scored: a first token is predicted from the token in front of it in the packed row
    (loss.py:51-58), which moving the sample would change; prompts are masked out, so real rollouts qualify.
    Samples without any scored token are dropped.  Host tensors only.

    Worth it only where the passes are long enough to run at full rate either way: with 150 K tokens per step and
    half of the problems settled a 0.6B step takes 0.49 s instead of 0.71 s, with 10 K tokens per step
    (one short pass becoming two shorter ones, of lengths no product has met before) 0.051-0.058 s instead of
    0.051-0.053 s.  So by default rows are separated only when at least `SPLIT_MIN_FORWARD_TOKENS` tokens go
    forward-only out of at least `SPLIT_MIN_TOKENS`; PRL_SPLIT_NO_GRADIENT=1 separates always, =0 never."""
    import os

    import numpy as np

    from .attention import sample_starts

    mode = os.environ.get("PRL_SPLIT_NO_GRADIENT", "auto")
    if mode == "0" or not micro_batches:
        return None
    if any(mb["loss_mask"].is_cuda or mb["loss_mask"].ndim != 2 or mb["loss_mask"].shape[0] != 1 for mb in micro_batches):
        return None
    grad: list[MicroBatch] = []
    fwd: list[MicroBatch] = []
    grad_row: list[int] = []
    fwd_row: list[int] = []
    changed = False

    def piece(mb: MicroBatch, a: int, z: int) -> MicroBatch:
        return {k: (v[:, a:z] if isinstance(v, Tensor) and v.ndim == 2 else v) for k, v in mb.items()}

    for r, mb in enumerate(micro_batches):
        # (the common case, settled first and through numpy: this runs with the GPU waiting for the step)
        adv = mb["advantages"][0].numpy()
        if np.count_nonzero(adv) == adv.size:
            grad.append(mb)
            grad_row.append(r)
            continue
        mask = mb["loss_mask"][0].numpy().astype(bool)
        live = mask & (adv != 0)  # (a NaN advantage counts: its NaN must reach the loss)
        n_scored, n_live = int(np.count_nonzero(mask)), int(np.count_nonzero(live))
        if n_scored == 0 or n_live == n_scored:  # nobody scored (the caller's business), or all live
            grad.append(mb)
            grad_row.append(r)
            continue
        if n_live == 0:  # scored, but nothing to learn from: the whole row forward only
            fwd.append(mb)
            fwd_row.append(r)
            changed = True
            continue
        starts = sample_starts(mb["position_ids"])
        n = int(mask.size)
        any_live = np.add.reduceat(live.astype(np.int64), starts) > 0
        if bool(any_live.all()) or bool(mask[starts].any()):
            grad.append(mb)  # every sample has live tokens, or a sample cannot be moved
            grad_row.append(r)
            continue
        any_scored = np.add.reduceat(mask.astype(np.int64), starts) > 0
        ends = np.append(starts[1:], n)
        for a, z, lv, sc in zip(starts.tolist(), ends.tolist(), any_live.tolist(), any_scored.tolist()):
            if lv:
                grad.append(piece(mb, a, z))
                grad_row.append(r)
            elif sc:
                fwd.append(piece(mb, a, z))
                fwd_row.append(r)
        changed = True
    if not changed:
        return None
    if mode != "1":
        moved = sum(int(mb["input_ids"].numel()) for mb in fwd)
        total = moved + sum(int(mb["input_ids"].numel()) for mb in grad)
        if moved < SPLIT_MIN_FORWARD_TOKENS or total < SPLIT_MIN_TOKENS:
            return None
    return grad, grad_row, fwd, fwd_row


SPLIT_MIN_FORWARD_TOKENS = 4096  # `split_by_gradient`: fewer tokens than this are not worth a pass of their own
SPLIT_MIN_TOKENS = 16384  # ... and a step shorter than this is one pass, which splitting would only make two of


def losses_per_given_row(parts: list[Tensor], rows: list[int], n_given: int, device: torch.device | str) -> Tensor:
    """The loss the reference logs per micro-batch (train.py:327-333) from the losses of the pieces the rows were
    run as: `parts` concatenated has one entry per piece, `rows[i]` is the given row piece i came from.  Rows that
    were not run at all (nothing scored) have loss 0.  Pieces of one row are added in the order they ran."""
    out = torch.zeros(n_given, dtype=torch.float32, device=device)
    if not rows:
        return out
    slot, seen = [], {}
    for r in rows:
        slot.append(seen.get(r, 0))
        seen[r] = slot[-1] + 1
    table = torch.zeros((n_given, max(seen.values())), dtype=torch.float32, device=device)
    vals = torch.cat([p.reshape(-1).float() for p in parts])
    table[host_to_device(rows, device, torch.int64), host_to_device(slot, device, torch.int64)] = vals
    return table.sum(1)


MERGE_MAX_TOKENS = 16384  # longest joined row (the one-launch loss kernel's limit, 5 GB of bf16 logits at V = 152 K)


@dataclass
class JoinedRows:
    """Several packed rows of a step joined end to end into one, for ONE forward / backward over all of them."""

    micro_batch: MicroBatch  # the joined row; the first token of every row but the first is masked out (see below)
    starts: list[int]  # first token of every original row
    lengths: list[int]
    loss_mask: Tensor  # the original masks, joined ([1, T], host)
    padding: int = 0  # unscored tokens appended after the last row (the joined row's length is rounded up)


PAD_TOKENS = 512  # joined rows are rounded up to a multiple of this many tokens


def _pad_tokens() -> int:
    """PRL_PAD_TOKENS overrides the rounding of joined rows (0: none)."""
    import os

    try:
        return max(0, int(os.environ.get("PRL_PAD_TOKENS", PAD_TOKENS)))
    except ValueError:
        return PAD_TOKENS


def _padding(n: int, like: MicroBatch) -> MicroBatch:
    """`n` tokens nobody scores, as samples of at most 64 tokens (short enough for every attention path): mask
    False, advantage 0.  They follow the last real row, so every real token keeps its place."""
    return {"input_ids": torch.zeros(1, n, dtype=like["input_ids"].dtype),
            "position_ids": (torch.arange(n, dtype=like["position_ids"].dtype) % 64).unsqueeze(0),
            "advantages": torch.zeros(1, n, dtype=like["advantages"].dtype),
            "logprobs": torch.zeros(1, n, dtype=like["logprobs"].dtype),
            "loss_mask": torch.zeros(1, n, dtype=torch.bool)}


def join_micro_batches(micro_batches: list[MicroBatch], max_tokens: int = MERGE_MAX_TOKENS,
                       pad_to: int | None = None) -> list[JoinedRows]:
    """Consecutive micro-batches with the same temperature joined into rows of at most `max_tokens` tokens.

    The reference runs one forward / backward per micro-batch because of memory (gradient accumulation,
    src/prime_rl/trainer/rl/train.py:271-310); with 288 GB a whole step's rows fit at once, and at 16 K tokens
    the GEMMs and element-wise kernels of a 0.6B model run at twice the rate they reach at 2 K.  In token mode
    the result is the same sum: samples never attend across a position restart, the loss is a sum over tokens
    under one normaliser for the whole step.  One token per row needs care: the reference scores the FIRST token
    of a row against all-zero logits (loss.py:51-58), while in a joined row it would be scored against the
    previous row's last logits.  Those tokens are masked out here and put back by `restore_row_starts`."""
    out: list[JoinedRows] = []
    group: list[MicroBatch] = []
    pad_to = _pad_tokens() if pad_to is None else pad_to

    def flush() -> None:
        if not group:
            return
        lengths = [int(mb["input_ids"].shape[1]) for mb in group]
        starts = [0]
        for n in lengths[:-1]:
            starts.append(starts[-1] + n)
        # Rows of real rollouts never have the same length twice, and every new length costs the GEMM library a
        # kernel selection per product and the allocator fresh blocks: the joined row is rounded up with tokens
        # nobody scores, so that a run sees a handful of lengths instead of a new one per step.
        pieces = list(group)
        total = sum(lengths)
        pad = (-total) % pad_to if pad_to > 0 and all(mb["input_ids"].shape[0] == 1 for mb in group) else 0
        if pad and total + pad > max_tokens:
            pad = 0
        if pad:
            pieces.append(_padding(pad, group[0]))
        joined = {k: torch.cat([mb[k] for mb in pieces], dim=1) for k in
                  ("input_ids", "position_ids", "advantages", "logprobs", "loss_mask")}
        joined["temperature"] = group[0]["temperature"]
        original = joined["loss_mask"]
        if len(group) > 1:
            joined["loss_mask"] = original.clone()
            joined["loss_mask"][0, starts[1:]] = False
        out.append(JoinedRows(joined, starts, lengths, original, pad))
        group.clear()

    tokens = 0
    for mb in micro_batches:
        n = int(mb["input_ids"].shape[1])
        fits = mb["input_ids"].shape[0] == 1 and group and group[0]["temperature"] == mb["temperature"] \
            and tokens + n <= max_tokens
        if not fits:
            flush()
            tokens = 0
        group.append(mb)
        tokens += n
        if mb["input_ids"].shape[0] != 1:
            flush()
            tokens = 0
    flush()
    return out


def join_per_row(pieces: list[Tensor], rows: list[JoinedRows]) -> list[Tensor]:
    """Per-micro-batch [1, T_i] tensors joined the way `rows` joined the micro-batches."""
    out, a = [], 0
    for r in rows:
        k = len(r.lengths)
        parts = list(pieces[a : a + k])
        if r.padding:
            parts.append(torch.ones((1, r.padding), dtype=parts[0].dtype, device=parts[0].device))
        out.append(parts[0] if len(parts) == 1 else torch.cat(parts, dim=1))
        a += k
    return out


def restore_row_starts(rows: JoinedRows, mb: dict[str, Any], aux: dict[str, Tensor], loss_config: LossConfig,
                       vocab: int, inv_scale: float) -> Tensor:
    """Give the first token of every joined row but the first what the reference gives it: log-probability
    -ln V and entropy ln V (all-zero logits), the ratio and loss term that follow from those, no gradient (a
    constant has none).  `mb` is the joined row on the device, `aux` what `policy_loss` returned for it (fixed in
    place).  Returns the losses of the original rows [n_rows], each under the step's normaliser `1 / inv_scale`
    (what the reference logs as one `loss` per micro-batch)."""
    dev = aux["logprobs"].device
    mask = host_to_device(rows.loss_mask, dev).reshape(-1)
    if len(rows.starts) > 1:
        idx = host_to_device(rows.starts[1:], dev, torch.int64)
        ln_v = float(torch.log(torch.tensor(float(vocab), dtype=torch.float32)))
        old = mb["logprobs"].reshape(-1)[idx].float()
        ratio = torch.exp(-ln_v - old)
        # (index_fill_ takes the number as a kernel argument.  `x[idx] = number` keeps the host until the stream
        # has drained, `tools/setitem_sync_probe.py`: once per pass, that was 15 % of a step of ten passes)
        aux["logprobs"].reshape(-1).index_fill_(0, idx, -ln_v)
        aux["entropy"].reshape(-1).index_fill_(0, idx, ln_v)
        aux["importance_ratio"][idx] = ratio
        aux["clipped_importance_ratio"][idx] = torch.clamp(ratio, max=float(loss_config.clip_ratio))
        aux["is_clipped"][idx] = (ratio > float(loss_config.clip_ratio)).float()
    term = -(aux["clipped_importance_ratio"] * mb["advantages"].reshape(-1).float()) * mask * inv_scale
    # one launch for all rows (a joined row of ragged rollouts has a hundred pieces, and this runs between passes)
    lengths = list(rows.lengths) + ([term.numel() - sum(rows.lengths)] if term.numel() > sum(rows.lengths) else [])
    # (`unsafe`: the lengths are not checked, which would read them back and keep the host until the stream drains)
    sums = torch.segment_reduce(term, "sum", lengths=host_to_device(lengths, dev, torch.int64), unsafe=True)
    return sums[: len(rows.lengths)]


def train_step(model: nn.Module, optimizer: torch.optim.Optimizer, micro_batches: list[MicroBatch],
               loss_config: LossConfig, max_norm: float = 1.0, device: torch.device | str = "cuda",
               scheduler: Any = None, samples: int | None = None,
               recomputed_errors: list[Tensor] | None = None, join_tokens: int = MERGE_MAX_TOKENS,
               skip_unscored: bool = True) -> StepResult:
    """One optimizer step over this rank's micro-batches
    (src/prime_rl/trainer/rl/train.py:259-364).  `recomputed_errors` are the
    per-micro-batch tensors `recompute_logprobs` returned; they are logged as
    `recomputed_logprob_error` (train.py:316-318).  In token mode consecutive
    micro-batches are joined into rows of up to `join_tokens` tokens and run as
    one (`join_micro_batches`; 0 keeps them apart).  Micro-batches without an
    unmasked token (the orchestrator's fillers) are not run in token mode where
    that is this rank's own business (`scored`; `skip_unscored=False` runs them),
    and samples whose advantage is 0 wherever they are scored take a forward
    pass only (`split_by_gradient`)."""
    t0 = time.perf_counter()
    n_given = len(micro_batches)
    loss_scale = local_loss_scale(micro_batches, loss_config)
    tensors = Tensors()
    tokens = 0
    # Ranks may end up with different numbers of joined rows (their rows differ in length).  That is harmless when
    # a step's collectives do not depend on the number of passes (parameters kept gathered: one all-gather with the
    # first pass, one reduce-scatter with the last); a model that reshards after every pass must run the same
    # number of passes on every rank, so there the micro-batches stay apart.
    one_rank = not dist.is_initialized() or dist.get_world_size() == 1
    passes_are_local = one_rank or getattr(model, "_prl_keep_gathered", False) or not hasattr(model, "unshard")
    token_mode = loss_config.ratio_type == "token"
    given_row = list(range(n_given))  # the given row every micro-batch still in the list came from
    # (token mode only: with one ratio per sample the reference logs every sample's ratio unmasked, a filler's too)
    if passes_are_local and skip_unscored and token_mode:
        keep = scored(micro_batches)
        micro_batches = [micro_batches[i] for i in keep]
        given_row = keep
        if recomputed_errors is not None:
            recomputed_errors = [recomputed_errors[i] for i in keep]
    # samples nothing can be learned from (advantage 0 wherever they are scored) take a forward pass only
    forward_only: list[MicroBatch] = []
    forward_row: list[int] = []
    if passes_are_local and skip_unscored and token_mode and recomputed_errors is None:
        split = split_by_gradient(micro_batches)
        if split is not None:
            micro_batches, rows_a, forward_only, rows_b = split
            if not micro_batches and not one_rank and hasattr(model, "unshard"):
                # FSDP2 issues the step's reduce-scatters from inside a backward pass: a rank whose samples all
                # have advantage 0 still runs one (over gradients of zero), or the other ranks would wait for it
                micro_batches, rows_a, forward_only, rows_b = forward_only[:1], rows_b[:1], forward_only[1:], rows_b[1:]
            forward_row = [given_row[i] for i in rows_b]
            given_row = [given_row[i] for i in rows_a]
    inv_scale = 1.0 / max(float(loss_scale), 1.0)
    loss_parts: list[Tensor] = []
    loss_rows: list[int] = []
    passes = 0
    vocab = int(getattr(getattr(model, "config", None), "vocab_size", 0))

    def units_of(mbs: list[MicroBatch]) -> tuple[list[JoinedRows] | None, list[MicroBatch]]:
        rows = join_micro_batches(mbs, join_tokens) if token_mode and join_tokens > 0 and passes_are_local else None
        return rows, ([r.micro_batch for r in rows] if rows is not None else mbs)

    def pieces_of(rows: list[JoinedRows] | None, k: int, came_from: list[int]) -> list[int]:
        n = len(rows[k].lengths) if rows is not None else 1
        at = sum(len(r.lengths) for r in rows[:k]) if rows is not None else k
        return came_from[at : at + n]

    def book(joined: JoinedRows | None, mb: dict[str, Any], loss: Tensor, aux: dict[str, Tensor], came_from: list[int],
             error: Tensor | None) -> None:
        nonlocal tokens, passes
        if joined is not None:
            losses = restore_row_starts(joined, mb, aux, loss_config, vocab, inv_scale)
            mask = host_to_device(joined.loss_mask, device)
        else:
            losses, mask = loss.detach().reshape(1), mb["loss_mask"]
        tensors.add("probs", aux["logprobs"], mask, exp=True)
        tensors.add("old_probs", mb["logprobs"], mask, exp=True)
        tensors.add("entropy", aux["entropy"], mask)
        loss_parts.append(losses)
        loss_rows.extend(came_from)
        if error is not None:
            tensors.add("recomputed_logprob_error", error, mask)
        add_ratio_statistics(tensors, aux, mask, loss_config)
        tokens += int(mb["input_ids"].numel()) - (joined.padding if joined is not None else 0)
        passes += 1

    rows_b, units_b = units_of(forward_only)
    for k, host_mb in enumerate(units_b):
        mb = to_device(host_mb, device)
        plan = attention_plan(host_mb, device)
        keep = scored_rows(host_mb, loss_config, model, device)
        with torch.no_grad(), plan_scope(plan):
            logits = forward(model, mb["input_ids"], mb["position_ids"], plan, positions=keep)
            vocab = int(logits.shape[-1])
            loss, aux = policy_loss(logits, mb["input_ids"], mb["position_ids"], mb["advantages"], mb["logprobs"],
                                    mb["loss_mask"], mb["temperature"], loss_config, loss_scale, skip_masked=True,
                                    positions=keep, max_sample_len=None if plan is None else plan.lmax)
            del logits
        book(rows_b[k] if rows_b is not None else None, mb, loss, aux, pieces_of(rows_b, k, forward_row), None)

    rows, units = units_of(micro_batches)
    if rows is not None and recomputed_errors is not None:
        recomputed_errors = join_per_row(recomputed_errors, rows)
    for step_idx, host_mb in enumerate(units):
        last = step_idx == len(units) - 1
        if hasattr(model, "set_requires_all_reduce"):
            model.set_requires_all_reduce(last)
        if getattr(model, "_prl_keep_gathered", False) and hasattr(model, "set_requires_gradient_sync"):
            # gradients accumulate unsharded (in the fp32 reduce dtype) and are reduce-scattered once, after the
            # last micro-batch; the parameters stay gathered until then
            model.set_requires_gradient_sync(last)
            model.set_reshard_after_backward(last)
        mb = to_device(host_mb, device)
        plan = attention_plan(host_mb, device)
        keep = scored_rows(host_mb, loss_config, model, device)  # the LM head only where the next token is scored
        with plan_scope(plan):  # backward included: checkpointed blocks run their forward again in there
            logits = forward(model, mb["input_ids"], mb["position_ids"], plan, positions=keep)
            vocab = int(logits.shape[-1])
            loss, aux = policy_loss(logits, mb["input_ids"], mb["position_ids"], mb["advantages"], mb["logprobs"],
                                    mb["loss_mask"], mb["temperature"], loss_config, loss_scale, skip_masked=True,
                                    positions=keep, max_sample_len=None if plan is None else plan.lmax)
            del logits
            loss.backward(unit_gradient(loss.device))
        book(rows[step_idx] if rows is not None else None, mb, loss, aux, pieces_of(rows, step_idx, given_row),
             recomputed_errors[step_idx] if recomputed_errors is not None else None)
    # one loss per given micro-batch, as the reference logs it; 0 for those not run (nothing scored)
    tensors.add("loss", losses_per_given_row(loss_parts, loss_rows, n_given, device))
    if not units:  # nothing to learn from: the optimizer still takes its step, on gradients of zero
        for prm in model.parameters():
            if prm.requires_grad and prm.grad is None:
                prm.grad = torch.zeros_like(prm)
    grad_norm = torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm=max_norm)
    if hasattr(grad_norm, "full_tensor"):
        grad_norm = grad_norm.full_tensor()
    optimizer.step()
    optimizer.zero_grad(set_to_none=True)
    if scheduler is not None:
        scheduler.step()
    stats = tensors.compute_stats()
    if recomputed_errors is None:
        ones_as_recomputed_error(stats)
    seconds = time.perf_counter() - t0
    if samples is None:
        # a packed row carries as many rollouts as it has position resets; for the
        # synthetic loader that is the micro-batch size
        samples = n_given
    return StepResult(loss=stats.get("loss/mean", float("nan")), grad_norm=float(grad_norm), stats=stats,
                      seconds=seconds, tokens=tokens, samples=samples, passes=passes)


@torch.no_grad()
def recompute_logprobs(logprob_model: nn.Module, micro_batches: list[MicroBatch],
                       device: torch.device | str = "cuda", join_tokens: int = MERGE_MAX_TOKENS) -> list[Tensor]:
    """Replace every micro-batch's sampler log-probabilities by the ones
    `logprob_model` assigns (the trainer's optional recompute pass,
    src/prime_rl/trainer/rl/train.py:227-245).  Returns, per micro-batch, the
    device tensor exp(recomputed - sampler) that the trainer logs as
    `recomputed_logprob_error`.  Same single pass over the logits as the training
    forward; nothing is kept for a backward.  Rows are joined as in `train_step`
    when the model's forward needs no collective per pass (`join_tokens` = 0 keeps
    them apart); the first token of every row then gets the -ln V of the
    all-zero logits the reference scores it against."""
    from .loss import selective_log_softmax, shift_logits

    one_rank = not dist.is_initialized() or dist.get_world_size() == 1
    joinable = join_tokens > 0 and (one_rank or not hasattr(logprob_model, "unshard"))
    given = micro_batches
    if joinable:  # passes are this rank's own business: fillers are not scored (their error is logged under an
        keep = scored(given)  # all-False mask, their log-probabilities are never used)
        micro_batches = [given[i] for i in keep]
    rows = join_micro_batches(micro_batches, join_tokens) if joinable else \
        [JoinedRows(mb, [0], [int(mb["input_ids"].shape[1])], mb["loss_mask"]) for mb in micro_batches]
    errors, k = [], 0
    for r in rows:
        mb = r.micro_batch
        ids = host_to_device(mb["input_ids"], device)
        logits = forward(logprob_model, ids, host_to_device(mb["position_ids"], device), attention_plan(mb, device))
        lp = selective_log_softmax(shift_logits(logits) / mb["temperature"], ids)
        del logits
        if len(r.starts) > 1:
            idx = host_to_device(r.starts[1:], device, torch.int64)
            ln_v = float(torch.log(torch.tensor(float(logprob_model.config.vocab_size), dtype=torch.float32)))
            lp.index_fill_(1, idx, -ln_v)  # (not `lp[:, idx] = number`, which waits for the stream)
        for a, n in zip(r.starts, r.lengths):  # back to the micro-batches the caller holds
            piece = lp[:, a : a + n]
            errors.append(torch.exp(piece - host_to_device(micro_batches[k]["logprobs"], device)))
            micro_batches[k]["logprobs"] = piece.cpu()
            k += 1
    if len(micro_batches) != len(given):
        done = {id(mb): e for mb, e in zip(micro_batches, errors)}
        errors = [done[id(mb)] if id(mb) in done else torch.ones(mb["logprobs"].shape, dtype=torch.float32, device=device)
                  for mb in given]
    return errors


def setup_optimizer(model: nn.Module, lr: float = 1e-6, weight_decay: float = 0.01,
                    betas: tuple[float, float] = (0.9, 0.999)) -> torch.optim.Optimizer:
    """AdamW with the reference's defaults (src/prime_rl/trainer/optim.py:19-25,
    src/prime_rl/trainer/config.py:299-315)."""
    first = next(iter(model.parameters()), None)
    fused = bool(first is not None and first.is_cuda)  # one kernel instead of a dozen launches per step on a GPU
    return torch.optim.AdamW(model.parameters(), lr=lr, weight_decay=weight_decay, betas=betas, fused=fused)


def rccl_environment() -> None:
    """What a process must have in its environment before RCCL builds a communicator between processes on the
    MI355X nodes this package runs on: their host driver shares device memory between processes through dmabuf
    handles only, and with the HSA runtime's legacy IPC mode RCCL's set-up (and any sharing of a device tensor
    between processes) fails in `hipIpcGetMemHandle: invalid argument`.  The nodes' images export
    HSA_ENABLE_IPC_MODE_LEGACY=0 for that reason; it is set here as well, unless the caller has set it, so that
    a rank started with a scrubbed environment behaves the same.  Call before the first HIP call of the process
    when possible (the runtime reads it at start-up); `bench.rank_commands` puts it into the ranks' environment
    for that reason."""
    import os

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def init_distributed(device: torch.device) -> Any:
    """RCCL process group (backend "nccl" on ROCm) and the 1-D data-parallel mesh
    (src/prime_rl/trainer/utils.py:27-33; dp_shard = world size as in
    src/prime_rl/trainer/parallel_dims.py:251-258 with every other degree at 1)."""
    from torch.distributed.device_mesh import init_device_mesh

    if not dist.is_initialized():
        rccl_environment()
        dist.init_process_group("nccl", device_id=device)
    return init_device_mesh("cuda", (dist.get_world_size(),), mesh_dim_names=("dp_shard",))
