"""The micro-batch schema and the two loaders of the RL trainer
(src/prime_rl/trainer/rl/data.py:13-74): synthetic batches for debugging and
benchmarks, and the rollout files the orchestrator writes per trainer rank at
`<output_dir>/rollouts/step_<N>/rank_<R>.pt`."""

from __future__ import annotations

import time
from pathlib import Path
from typing import TypedDict

import torch
from torch import Tensor

from .config import FakeDataLoaderConfig
from .world import get_world


class MicroBatch(TypedDict, total=False):
    # token level, each [1, T] for a packed row
    input_ids: Tensor  # int64
    position_ids: Tensor  # int64
    advantages: Tensor  # fp32
    logprobs: Tensor  # fp32, the sampler's log-probabilities
    loss_mask: Tensor  # bool
    # batch level
    temperature: float
    total_tokens: int  # declared by the schema; the producer does not write it


def get_rollout_dir(output_dir: Path) -> Path:
    return Path(output_dir) / "rollouts"


def wait_for_path(path: Path, interval: float = 1.0, timeout: float | None = None) -> None:
    """Poll until `path` exists (src/prime_rl/utils/utils.py:119-130).  The writer
    publishes with write-to-.tmp + rename, so existence means complete."""
    waited = 0.0
    while not Path(path).exists():
        if timeout is not None and waited >= timeout:
            raise TimeoutError(f"{path} did not appear within {timeout} s")
        time.sleep(interval)
        waited += interval


def synthetic_row(micro_batch_size: int, seq_len: int, generator: torch.Generator) -> MicroBatch:
    """One packed row of `micro_batch_size` full-length samples with random content: token ids below 100,
    standard-normal advantages and sampler log-probabilities, every token in the loss, temperature 1
    (the distribution of src/prime_rl/trainer/rl/data.py:39-54)."""
    t = micro_batch_size * seq_len
    row: MicroBatch = {"temperature": 1.0}
    row["input_ids"] = torch.randint(0, 100, (1, t), generator=generator)
    row["position_ids"] = torch.arange(seq_len).repeat(micro_batch_size).unsqueeze(0)
    for key in ("advantages", "logprobs"):
        row[key] = torch.randn(1, t, generator=generator)
    row["loss_mask"] = torch.ones(1, t, dtype=torch.bool)
    return row


class FakeDataLoader:
    """Synthetic batches of fixed shape for debugging and benchmarks; each rank gets
    batch_size / micro_batch_size / world_size rows per step (src/prime_rl/trainer/rl/data.py:26-37).
    Unlike the reference the generator is seeded (1234 + rank), so runs repeat."""

    def __init__(self, config: FakeDataLoaderConfig, seed: int | None = None):
        world = get_world()
        self.config = config
        self.num_micro_batches = config.batch_size // config.micro_batch_size // world.world_size
        self.generator = torch.Generator().manual_seed(1234 + world.rank if seed is None else seed)

    def wait_for_batch(self) -> None:
        """Nothing to wait for."""

    def ready(self) -> bool:
        return True

    def get_batch(self) -> list[MicroBatch]:
        c = self.config
        return [synthetic_row(c.micro_batch_size, c.seq_len, self.generator) for _ in range(self.num_micro_batches)]


class DataLoader:
    """Loads the micro-batches the orchestrator serialised for this rank
    (src/prime_rl/trainer/rl/data.py:57-74)."""

    def __init__(self, output_dir: Path, start_step: int):
        self.rollout_dir = get_rollout_dir(output_dir)
        self.current_step = start_step
        self.world = get_world()

    def get_rollout_path(self) -> Path:
        return self.rollout_dir / f"step_{self.current_step}" / f"rank_{self.world.rank}.pt"

    def wait_for_batch(self, interval: float = 1.0, timeout: float | None = None) -> None:
        wait_for_path(self.get_rollout_path(), interval, timeout)

    def ready(self) -> bool:
        """Whether the next step's file is there already (the orchestrator renames it into place when it is
        complete, src/prime_rl/orchestrator/orchestrator.py:356-363): `get_batch` would not have to wait."""
        return self.get_rollout_path().exists()

    def get_batch(self) -> list[MicroBatch]:
        # a list of dicts of tensors and floats: nothing in it needs unpickling of code
        batches = torch.load(self.get_rollout_path(), weights_only=True)
        self.current_step += 1
        return batches


def save_rollout_batches(batches_per_rank: list[list[MicroBatch]], output_dir: Path, step: int) -> list[Path]:
    """Publish one file per trainer rank the way the orchestrator does
    (src/prime_rl/orchestrator/orchestrator.py:356-363): write `rank_i.tmp`,
    then rename to `rank_i.pt`, so a reader never sees a partial file."""
    step_dir = get_rollout_dir(output_dir) / f"step_{step}"
    step_dir.mkdir(parents=True, exist_ok=True)
    out = []
    for i, batches in enumerate(batches_per_rank):
        tmp = step_dir / f"rank_{i}.tmp"
        torch.save(batches, tmp)
        final = step_dir / f"rank_{i}.pt"
        tmp.rename(final)
        out.append(final)
    return out
