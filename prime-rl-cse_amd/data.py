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


class FakeDataLoader:
    """Random micro-batches of fixed shape (src/prime_rl/trainer/rl/data.py:26-54).
    Unlike the reference the generator is seeded (1234 + rank) so runs repeat."""

    def __init__(self, config: FakeDataLoaderConfig, seed: int | None = None):
        self.batch_size = config.batch_size
        self.micro_batch_size = config.micro_batch_size
        self.num_micro_batches = self.batch_size // self.micro_batch_size // get_world().world_size
        self.seq_len = config.seq_len
        self.generator = torch.Generator().manual_seed(1234 + get_world().rank if seed is None else seed)

    def wait_for_batch(self) -> None:
        return

    def get_batch(self) -> list[MicroBatch]:
        return [self._get_micro_batch() for _ in range(self.num_micro_batches)]

    def _get_micro_batch(self) -> MicroBatch:
        t = self.micro_batch_size * self.seq_len
        g = self.generator
        return {
            "input_ids": torch.randint(0, 100, (1, t), generator=g),
            "position_ids": torch.cat([torch.arange(self.seq_len)] * self.micro_batch_size).unsqueeze(0),
            "advantages": torch.randn(t, generator=g).unsqueeze(0),
            "logprobs": torch.randn(t, generator=g).unsqueeze(0),
            "temperature": 1.0,
            "loss_mask": torch.ones(t, dtype=torch.bool).unsqueeze(0),
        }


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
