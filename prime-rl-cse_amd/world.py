"""Where this process sits in the job, read once from the variables torchrun exports
(the role of src/prime_rl/trainer/world.py and src/prime_rl/trainer/envs.py in the reference)."""

from __future__ import annotations

import os
from dataclasses import dataclass

_ENV_DEFAULTS = {"RANK": 0, "WORLD_SIZE": 1, "LOCAL_RANK": 0, "LOCAL_WORLD_SIZE": 1}


@dataclass(frozen=True)
class World:
    rank: int
    world_size: int
    local_rank: int
    local_world_size: int

    @classmethod
    def from_env(cls) -> "World":
        vals = {k: int(os.environ.get(k, d)) for k, d in _ENV_DEFAULTS.items()}
        return cls(vals["RANK"], vals["WORLD_SIZE"], vals["LOCAL_RANK"], vals["LOCAL_WORLD_SIZE"])

    def __post_init__(self) -> None:
        problems = []
        if not 0 <= self.rank < self.world_size:
            problems.append(f"rank {self.rank} outside [0, {self.world_size})")
        if not 0 <= self.local_rank < self.local_world_size:
            problems.append(f"local rank {self.local_rank} outside [0, {self.local_world_size})")
        if self.local_world_size > self.world_size or self.world_size % self.local_world_size:
            problems.append(f"{self.world_size} ranks do not split into nodes of {self.local_world_size}")
        if problems:
            raise AssertionError("inconsistent torchrun environment: " + "; ".join(problems))

    @property
    def num_nodes(self) -> int:
        return self.world_size // self.local_world_size

    @property
    def is_master(self) -> bool:
        return self.rank == 0


_current: World | None = None


def get_world() -> World:
    """The process-wide World; built from the environment the first time it is asked for."""
    global _current
    if _current is None:
        _current = World.from_env()
    return _current


def reset_world() -> None:
    """Forget the cached World (tests change the environment between cases)."""
    global _current
    _current = None


if __name__ == "__main__":
    # under torchrun every rank prints where it sits (the reference runs its world.py the same way,
    # tests/unit/train/test_world.py:39-52)
    print(get_world())
