"""World (rank bookkeeping from torchrun's variables) on the cases the reference tests for its own
(tests/unit/train/test_world.py): defaults, valid and inconsistent environments, a real torchrun start."""

import os
import subprocess
import sys
from pathlib import Path

import pytest

from prime_rl_cse_amd.world import World, get_world, reset_world

ENV_VARS = ["RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE"]
REPO = Path(__file__).resolve().parent.parent


@pytest.fixture(autouse=True)
def clean_environment(monkeypatch):
    for k in ENV_VARS:
        monkeypatch.delenv(k, raising=False)
    reset_world()
    yield
    reset_world()


def test_defaults_are_one_process_on_one_node():
    world = get_world()
    assert world.world_size == world.local_world_size == 1
    assert world.rank == world.local_rank == 0
    assert world.num_nodes == 1 and world.is_master
    assert world is get_world()  # built once


@pytest.mark.parametrize("world_size,local_world_size", [(1, 1), (2, 1), (2, 2), (16, 8)])
def test_valid_environment(monkeypatch, world_size, local_world_size):
    for k, v in dict(RANK=0, WORLD_SIZE=world_size, LOCAL_RANK=0, LOCAL_WORLD_SIZE=local_world_size).items():
        monkeypatch.setenv(k, str(v))
    world = get_world()
    assert (world.world_size, world.local_world_size) == (world_size, local_world_size)
    assert world.rank == world.local_rank == 0
    assert world.num_nodes == world_size // local_world_size
    assert world == get_world()


def test_last_rank_of_the_second_node(monkeypatch):
    for k, v in dict(RANK=15, WORLD_SIZE=16, LOCAL_RANK=7, LOCAL_WORLD_SIZE=8).items():
        monkeypatch.setenv(k, str(v))
    world = get_world()
    assert (world.rank, world.local_rank, world.num_nodes, world.is_master) == (15, 7, 2, False)


def test_more_local_ranks_than_ranks_is_refused(monkeypatch):
    monkeypatch.setenv("WORLD_SIZE", "1")
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "2")
    with pytest.raises(AssertionError):
        get_world()


def test_nodes_of_unequal_size_are_refused():
    with pytest.raises(AssertionError):
        World(rank=0, world_size=6, local_rank=0, local_world_size=4)


@pytest.mark.parametrize("rank,world_size", [(1, 1), (-1, 1), (8, 8)])
def test_rank_outside_the_job_is_refused(monkeypatch, rank, world_size):
    monkeypatch.setenv("RANK", str(rank))
    monkeypatch.setenv("WORLD_SIZE", str(world_size))
    monkeypatch.setenv("LOCAL_WORLD_SIZE", str(world_size))
    with pytest.raises(AssertionError):
        get_world()


@pytest.mark.parametrize("local_rank,local_world_size", [(1, 1), (-1, 1)])
def test_local_rank_outside_the_node_is_refused(monkeypatch, local_rank, local_world_size):
    monkeypatch.setenv("LOCAL_RANK", str(local_rank))
    monkeypatch.setenv("LOCAL_WORLD_SIZE", str(local_world_size))
    with pytest.raises(AssertionError):
        get_world()


@pytest.mark.parametrize("ranks", [1, 2])
def test_under_torchrun_every_rank_reads_its_own_place(ranks):
    port = 29700 + ranks + os.getpid() % 200
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={ranks}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), "-m", "prime_rl_cse_amd.world"]
    env = {**os.environ, "PYTHONPATH": str(REPO) + os.pathsep + os.environ.get("PYTHONPATH", "")}
    for k in ENV_VARS:
        env.pop(k, None)
    done = subprocess.run(cmd, cwd=REPO, env=env, capture_output=True, text=True, timeout=240)
    assert done.returncode == 0, done.stderr[-2000:] or done.stdout[-2000:]
    for r in range(ranks):
        assert f"World(rank={r}, world_size={ranks}, local_rank={r}, local_world_size={ranks})" in done.stdout
