"""Process topology from the environment torchrun sets, as a singleton
(counterpart of src/prime_rl/trainer/world.py:4-44 and src/prime_rl/trainer/envs.py)."""

from __future__ import annotations

import os


class World:
    def __init__(self):
        self.rank = int(os.environ.get("RANK", "0"))
        self.world_size = int(os.environ.get("WORLD_SIZE", "1"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.local_world_size = int(os.environ.get("LOCAL_WORLD_SIZE", "1"))
        assert 0 <= self.local_rank < self.local_world_size
        assert 0 <= self.rank < self.world_size
        assert self.local_world_size <= self.world_size
        assert self.world_size % self.local_world_size == 0
        self.num_nodes = self.world_size // self.local_world_size

    @property
    def is_master(self) -> bool:
        return self.rank == 0

    def __repr__(self) -> str:
        return (
            f"World(world_size={self.world_size}, rank={self.rank}, local_rank={self.local_rank}, "
            f"local_world_size={self.local_world_size}, num_nodes={self.num_nodes})"
        )


_WORLD: World | None = None


def get_world() -> World:
    """The World, created from the environment on first use."""
    global _WORLD
    if _WORLD is None:
        _WORLD = World()
    return _WORLD


def reset_world() -> None:
    global _WORLD
    _WORLD = None
