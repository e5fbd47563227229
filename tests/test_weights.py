"""Weight publication: file layout and atomic rename of the reference, the keep / clean rule,
the gather over ranks (gloo, world size 2) and, on the GPU, the side-stream snapshot."""

import os
from pathlib import Path

import pytest
import torch
from torch import nn

from prime_rl_cse_amd.weights import (
    WeightPublisher, get_weight_ckpt_model_path, get_weights_dir, wait_for_weight_checkpoint,
)
from prime_rl_cse_amd.world import reset_world


class Tiny(nn.Module):
    def __init__(self):
        super().__init__()
        self.a = nn.Linear(6, 5)
        self.b = nn.Embedding(7, 3)
        self.register_buffer("steps", torch.tensor([3], dtype=torch.int64))


def test_publish_writes_reference_layout(tmp_path):
    reset_world()
    torch.manual_seed(0)
    model = Tiny()
    pub = WeightPublisher(tmp_path, async_level=1)
    pending = pub.publish(model, step=4)
    path = pending.wait(timeout=30)
    assert path == tmp_path / "weights" / "step_4" / "pytorch_model.bin" == get_weight_ckpt_model_path(
        get_weights_dir(tmp_path), 4)
    assert path.exists() and not list(tmp_path.rglob("*.tmp"))
    wait_for_weight_checkpoint(get_weights_dir(tmp_path), 4, interval=0.01, timeout=1)
    state = torch.load(path, weights_only=True)
    assert set(state) == set(model.state_dict())
    for k, v in model.state_dict().items():
        if v.is_floating_point():
            assert state[k].dtype == torch.bfloat16 and torch.equal(state[k], v.to(torch.bfloat16))
        else:
            assert torch.equal(state[k], v)


def test_crc32_switch_is_explicit_and_put_back(tmp_path):
    """torch has one process-wide switch for the zip records' CRC32.  A publisher leaves it alone unless asked
    (`crc32=False`), and `close()` restores what it found."""
    reset_world()
    before = torch.serialization.get_crc32_options()
    pub = WeightPublisher(tmp_path / "a", async_level=1)
    assert torch.serialization.get_crc32_options() == before
    pub.close()
    fast = WeightPublisher(tmp_path / "b", async_level=1, crc32=False)
    assert torch.serialization.get_crc32_options() is False
    path = fast.publish(Tiny(), step=1).wait(timeout=30)
    assert set(torch.load(path, weights_only=True)) == set(Tiny().state_dict())  # torch.load does not need it
    fast.close()
    assert torch.serialization.get_crc32_options() == before


def test_keep_and_clean_rule(tmp_path):
    # src/prime_rl/trainer/weights.py:211-228: consider step - (async_level + 1); keep evaluation steps and the
    # async_level + 1 steps that lead up to a full checkpoint
    reset_world()
    pub = WeightPublisher(tmp_path, interval=10, ckpt_interval=8, async_level=2)
    assert pub.keep_step(0) and pub.keep_step(10) and pub.keep_step(20)  # evaluation steps
    assert pub.keep_step(6) and pub.keep_step(7) and pub.keep_step(8) and pub.keep_step(14)  # 8-2 .. 8, 16-2
    assert not pub.keep_step(5) and not pub.keep_step(9) and not pub.keep_step(11)
    for s in (3, 5, 6):
        (tmp_path / "weights" / f"step_{s}").mkdir(parents=True)
    assert pub.maybe_clean(8) == tmp_path / "weights" / "step_5" and not (tmp_path / "weights" / "step_5").exists()
    assert pub.maybe_clean(9) is None and (tmp_path / "weights" / "step_6").exists()  # kept for the checkpoint at 8
    no_rules = WeightPublisher(tmp_path, async_level=2)
    assert no_rules.maybe_clean(6, wait=False) == tmp_path / "weights" / "step_3"  # on a thread of its own
    no_rules.wait_all()
    assert not (tmp_path / "weights" / "step_3").exists()


def test_publications_in_flight_are_bounded(tmp_path):
    reset_world()
    pub = WeightPublisher(tmp_path, async_level=8, max_in_flight=2)
    big = nn.Sequential(*[nn.Linear(1024, 1024) for _ in range(24)])
    pendings = [pub.publish(big, step=s) for s in range(5)]
    alive = [p for p in pendings if p._thread.is_alive()]
    assert len(alive) <= 2 and not pendings[0]._thread.is_alive()  # the early ones were waited for
    pub.wait_all()
    assert all((tmp_path / "weights" / f"step_{s}" / "pytorch_model.bin").exists() for s in range(5))


def test_clean_waits_for_a_publication_still_being_written(tmp_path):
    """Steps faster than the disk: the directory about to be removed may still be receiving its file."""
    reset_world()
    pub = WeightPublisher(tmp_path, async_level=0)
    big = nn.Sequential(*[nn.Linear(1024, 1024) for _ in range(24)])
    pending = pub.publish(big, step=0)
    assert pub.maybe_clean(1) == tmp_path / "weights" / "step_0"  # step 1 considers step 0
    assert pending.error is None and not (tmp_path / "weights" / "step_0").exists()
    pending.wait(timeout=60)


def _rank(rank, world, port, root, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), LOCAL_WORLD_SIZE=str(world))
    import torch.distributed as dist
    from torch.distributed.device_mesh import init_device_mesh
    from torch.distributed.tensor import Shard, distribute_tensor

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        reset_world()
        mesh = init_device_mesh("cpu", (world,))
        gen = torch.Generator().manual_seed(1)
        full = {"w_even": torch.randn(8, 3, generator=gen), "w_odd": torch.randn(5, 4, generator=gen),
                "w_one_row": torch.randn(1, 6, generator=gen), "plain": torch.arange(4.0)}
        state = {k: (distribute_tensor(v, mesh, [Shard(0)]) if k != "plain" else v) for k, v in full.items()}
        pub = WeightPublisher(Path(root), async_level=1)
        pending = pub.publish(None, step=2, state_dict=state)
        pending.wait(timeout=60)
        dist.barrier()
        q.put((rank, {k: v.tolist() for k, v in full.items()}))
    finally:
        dist.destroy_process_group()


def test_two_ranks_gather_sharded_weights_over_gloo(tmp_path):
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rank, args=(r, 2, 29751, str(tmp_path), q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    state = torch.load(get_weight_ckpt_model_path(get_weights_dir(tmp_path), 2), weights_only=True)
    want = {k: torch.tensor(v) for k, v in res[0][1].items()}
    assert set(state) == set(want)
    for k, v in want.items():
        assert state[k].shape == v.shape and torch.equal(state[k], v.to(torch.bfloat16)), k


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_side_stream_snapshot_is_not_disturbed_by_the_next_update(tmp_path, dtype):
    """The file holds the weights as they were at publish(); an optimizer-like in-place update issued right
    afterwards (behind guard_parameters) must not leak into it.  bf16 parameters (the whole-model trainer's)
    need a copy just as fp32 ones need the cast."""
    reset_world()
    dev = torch.device("cuda:0")
    torch.manual_seed(1)
    model = nn.Sequential(nn.Linear(4096, 4096), nn.Linear(4096, 4096)).to(dev).to(dtype)
    before = {k: v.detach().to(torch.bfloat16).cpu() for k, v in model.state_dict().items()}
    pub = WeightPublisher(tmp_path, async_level=1, device=dev)
    seen = {}
    pub.add_device_subscriber(lambda state, step: seen.update(step=step, keys=set(state), dev=next(iter(state.values())).device))
    pending = pub.publish(model, step=1)
    pub.guard_parameters()
    with torch.no_grad():
        for p in model.parameters():
            p.add_(1.0)
    path = pending.wait(timeout=60)
    state = torch.load(path, weights_only=True)
    for k, v in before.items():
        assert torch.equal(state[k], v), k
    assert seen["step"] == 1 and seen["keys"] == set(before) and seen["dev"].type == "cuda"
    # a second publication sees the update
    path2 = pub.publish(model, step=2).wait(timeout=60)
    state2 = torch.load(path2, weights_only=True)
    assert torch.equal(state2["0.weight"], (model.state_dict()["0.weight"]).to(torch.bfloat16).cpu())
    assert not torch.equal(state2["0.weight"], state["0.weight"])


@pytest.mark.gpu
def test_whole_tensors_go_out_through_one_staging_buffer_and_stay_tied(tmp_path):
    """Tensors that share their memory (tied embeddings) share their slice of the staging buffer and come back
    tied from the file; integer buffers keep their type; four publications in a row reuse the pinned buffers and
    every file holds the weights of its own step."""
    reset_world()
    dev = torch.device("cuda:0")

    class Tied(nn.Module):
        def __init__(self):
            super().__init__()
            self.embed = nn.Embedding(1000, 64)
            self.body = nn.Linear(64, 64)
            self.head = nn.Linear(64, 1000, bias=False)
            self.head.weight = self.embed.weight
            self.register_buffer("steps", torch.arange(5))
            self.register_buffer("nothing", torch.zeros(0))  # an empty tensor ahead of the parameters' slices

    torch.manual_seed(3)
    model = Tied().to(dev).to(torch.bfloat16)
    pub = WeightPublisher(tmp_path, async_level=8, device=dev)
    want = []
    for step in range(1, 5):
        want.append({k: v.detach().cpu().clone() for k, v in model.state_dict().items()})
        pub.publish(model, step=step)
        pub.guard_parameters()
        with torch.no_grad():
            for p in model.parameters():
                p.mul_(1.5)
    pub.wait_all()
    assert len(pub._slots) <= pub.max_in_flight + 1
    for step, w in zip(range(1, 5), want):
        # read the way the reference's inference worker does (src/prime_rl/inference/vllm/worker.py:14-30)
        state = torch.load(tmp_path / "weights" / f"step_{step}" / "pytorch_model.bin", weights_only=True, mmap=True)
        assert set(state) == set(w) and state["steps"].dtype == torch.int64
        for k, v in w.items():
            assert torch.equal(state[k], v), (step, k)
        assert state["embed.weight"].data_ptr() == state["head.weight"].data_ptr()
    size = (tmp_path / "weights" / "step_1" / "pytorch_model.bin").stat().st_size
    assert size < 2 * (1000 * 64 + 64 * 64 + 64) + 16384  # the tied table is in the file once


# --------------------------------------------------------------------------- device-to-device hand-off (gloo, CPU tensors)


def _broadcast_rank(rank, world, port, q):
    import os

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist

    import prime_rl_cse_amd as prl
    from prime_rl_cse_amd.weights import WeightBroadcast

    dist.init_process_group("gloo", rank=rank, world_size=world)
    prl.reset_world()
    try:
        # ranks 0 and 1 are trainers (0 is the source), rank 2 is the "inference" side: group = {0, 2}
        group = dist.new_group([0, 2])
        ch = WeightBroadcast(group if rank in (0, 2) else None, src=0, device="cpu")
        if rank == 1:
            ch.is_member = False
            ch.send({"w": torch.zeros(1)}, 0)  # a trainer rank outside the group: returns at once
            q.put((rank, "skipped"))
        elif rank == 0:
            for step in (3, 4, 5):
                g = torch.Generator().manual_seed(step)
                state = {"a.weight": torch.randn(5, 7, generator=g).to(torch.bfloat16),
                         "b.bias": torch.randn(3, generator=g).to(torch.bfloat16),
                         "c.steps": torch.tensor([step], dtype=torch.int64)}
                ch.send(state, step)
            ch.close()
            q.put((rank, "sent"))
        else:
            got = []
            kept = None
            while True:
                r = ch.receive()
                if r is None:
                    break
                step, state = r
                g = torch.Generator().manual_seed(step)
                want_a = torch.randn(5, 7, generator=g).to(torch.bfloat16)
                want_b = torch.randn(3, generator=g).to(torch.bfloat16)
                ok = torch.equal(state["a.weight"], want_a) and torch.equal(state["b.bias"], want_b) \
                    and int(state["c.steps"]) == step and state["a.weight"].shape == (5, 7)
                if kept is not None:  # the previous publication is still intact while this one is in hand
                    pstep, pstate = kept
                    pg = torch.Generator().manual_seed(pstep)
                    ok = ok and torch.equal(pstate["a.weight"], torch.randn(5, 7, generator=pg).to(torch.bfloat16))
                kept = (step, state)
                got.append((step, ok))
            q.put((rank, got))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_weight_broadcast_between_trainer_and_inference_ranks_over_gloo():
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29000 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_broadcast_rank, args=(r, 3, port, q)) for r in range(3)]
    for p in procs:
        p.start()
    out = dict(q.get(timeout=120) for _ in range(3))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert out[0] == "sent" and out[1] == "skipped"
    assert out[2] == [(3, True), (4, True), (5, True)]


def _broadcast_rank_of_eight(rank, world, port, q):
    import os

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist

    import prime_rl_cse_amd as prl
    from prime_rl_cse_amd.weights import WeightBroadcast

    dist.init_process_group("gloo", rank=rank, world_size=world)
    prl.reset_world()
    try:
        # a node of eight: ranks 0-3 train (0 publishes), ranks 4-7 serve; the channel is {0, 4, 5, 6, 7}
        members = [0, 4, 5, 6, 7]
        group = dist.new_group(members)  # every rank makes the call, members or not
        ch = WeightBroadcast(group if rank in members else None, src=0, device="cpu")

        def weights(step):
            g = torch.Generator().manual_seed(step)
            return {"embed.weight": torch.randn(33, 16, generator=g).to(torch.bfloat16),
                    "layers.0.w": torch.randn(16, 16, generator=g).to(torch.bfloat16),
                    "norm.weight": torch.randn(16, generator=g).to(torch.bfloat16)}

        if rank in (1, 2, 3):
            ch.is_member = False
            ch.send(weights(1), 1)  # a trainer rank outside the channel: returns at once
            q.put((rank, "skipped"))
        elif rank == 0:
            for step in (1, 2, 3, 4):
                ch.send(weights(step), step)
            ch.close()
            q.put((rank, "sent"))
        else:
            got = []
            while True:
                r = ch.receive()
                if r is None:
                    break
                step, state = r
                want = weights(step)
                got.append((step, set(state) == set(want) and all(torch.equal(state[k], want[k]) for k in want)))
            q.put((rank, got))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_weight_broadcast_to_four_inference_ranks_of_an_eight_rank_node_over_gloo():
    """Four trainer ranks and four receiving ranks in one job: the publishing rank's flat buckets reach every
    receiver, publication after publication, and the other trainer ranks are not held up."""
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31000 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_broadcast_rank_of_eight, args=(r, 8, port, q)) for r in range(8)]
    for p in procs:
        p.start()
    out = dict(q.get(timeout=240) for _ in range(8))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert out[0] == "sent" and all(out[r] == "skipped" for r in (1, 2, 3))
    for r in (4, 5, 6, 7):
        assert out[r] == [(1, True), (2, True), (3, True), (4, True)], (r, out[r])


@pytest.mark.gpu
def test_weight_broadcast_as_a_publisher_subscriber_over_rccl(tmp_path, monkeypatch):
    """The sender half on the GPU through RCCL: a one-member group is all this box has, so the broadcasts have
    no receiver, but the bucket is packed on the publication stream from the gathered bf16 tensors and every
    collective is issued for real.  The bucket must hold exactly what the published file holds."""
    import torch.distributed as dist

    from prime_rl_cse_amd.weights import WeightBroadcast

    for k, v in dict(MASTER_ADDR="127.0.0.1", MASTER_PORT="29771", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0").items():
        monkeypatch.setenv(k, v)
    reset_world()
    dev = torch.device("cuda:0")
    dist.init_process_group("nccl", device_id=dev)
    try:
        torch.manual_seed(2)
        model = nn.Sequential(nn.Linear(256, 512), nn.Linear(512, 64)).to(dev)
        pub = WeightPublisher(tmp_path, async_level=1, device=dev)
        ch = WeightBroadcast(None, src=0, device=dev)
        pub.add_device_subscriber(ch.send)
        for step in (1, 2):
            with torch.no_grad():
                for p in model.parameters():
                    p.add_(0.5)
            path = pub.publish(model, step=step).wait(timeout=60)
            torch.cuda.synchronize()
            state = torch.load(path, weights_only=True)
            views = ch._views(ch._buckets[0])
            assert set(views) == set(state)
            for k, v in state.items():
                assert torch.equal(views[k].cpu(), v), (step, k)
            pub.guard_parameters()
        ch.close()
    finally:
        dist.barrier()
        dist.destroy_process_group()
        reset_world()
