"""RL trainer entry point:  torchrun ... -m prime_rl_cse_amd.train @ train.toml

The loop of src/prime_rl/trainer/rl/train.py:55-466 on MI355X: FSDP2 over RCCL
for the transformer body, this package's HIP path for everything after the LM
head, weight publication on a side stream.  Left out on purpose: W&B, full
(resumable) checkpoints, LoRA, MoE load-balance statistics, the torch profiler
hooks.
"""

from __future__ import annotations

import os
import statistics
import time
from pathlib import Path
from typing import Any

import torch
import torch.distributed as dist

from . import perf
from . import trainer as T
from .data import DataLoader, FakeDataLoader
from .optim import setup_optimizer, setup_scheduler
from .settings import RLTrainerConfig, parse_argv
from .weights import WeightPublisher
from .world import get_world

DTYPES = {"bfloat16": torch.bfloat16, "float32": torch.float32}


def log(msg: str) -> None:
    if get_world().is_master:
        print(msg, flush=True)


def trainer_mode(config: RLTrainerConfig) -> str:
    """"replica" (the whole model on every GPU, replica.py), "fsdp" (FSDP2 as the reference sets it
    up) or "auto" (replica where the model and optimizer allow it, else FSDP2).  A recipe asks for replica with
    `[model.compile]`, the reference's switch for its tracing compiler, and for FSDP2 by setting
    `model.reshard_after_forward` (either way: it then cares how FSDP2 treats its parameters); PRL_TRAINER
    overrides both.  Without any of these: auto."""

    env = os.environ.get("PRL_TRAINER", "").lower()
    if env in ("replica", "fsdp", "auto"):
        return env
    if config.model.compile is not None:
        return "replica"
    if "reshard_after_forward" in config.model.model_fields_set:
        return "fsdp"
    return "auto"


def replay_graphs(config: RLTrainerConfig) -> bool:
    """Whether the whole-model trainer replays its micro-batches from HIP graphs: only under PRL_GRAPHS=1, and
    then `Replica` looks at every captured pass before it runs and raises if it holds a memset node (library
    calls record them under capture: hipBLASLt's stream-K products clear their flags that way, about 130 per
    pass of the 0.6B model, and replays of such nodes have gone wrong on MI355X boxes).  `[model.compile]` in a
    recipe, the reference's opt-in for its tracing compiler, selects the whole-model trainer
    (`trainer_mode`) but no longer graph replay: with a step's rows joined into one pass replay saved 1 %."""

    if os.environ.get("PRL_GRAPHS", "") == "1":
        return True
    if config.model.compile is not None:
        log("[model.compile]: whole model per GPU, passes run eagerly (graph replay is off: captured library "
            "calls hold memset nodes; PRL_GRAPHS=1 captures, checks for them and refuses a graph that has any)")
    return False


def fused_blocks(config: RLTrainerConfig) -> bool:
    """Whether the transformer blocks run on this package's HIP kernels: when the recipe says
    `model.impl = "liger_kernel"` (the reference's switch for fused block kernels) and also when it says nothing;
    an explicit `impl = "hf"`, or PRL_BLOCKS=eager, keeps the transformers modules as they are."""

    env = os.environ.get("PRL_BLOCKS", "").lower()
    if env in ("eager", "fused"):
        return env == "fused"
    return config.model.impl == "liger_kernel" or "impl" not in config.model.model_fields_set


def model_config_of(config: RLTrainerConfig) -> Any:
    """The transformers config of `config.model`: read from the checkpoint where one is on this machine (a
    directory in HF layout, or the model's snapshot in the Hugging Face cache), else made from literal dimensions
    for the architectures the reference's recipes name (there is no hub access here)."""
    mc = config.model
    attn = "sdpa"  # flash-attn is a CUDA package; packed rows go through attention.py or SDPA's block-diagonal mask
    where = T.local_checkpoint(mc.name)
    if where is not None:
        from transformers import AutoConfig

        cfg = AutoConfig.from_pretrained(where)
        cfg._attn_implementation, cfg.use_cache = attn, False
        if mc.debug.num_layers is not None:
            cfg.num_hidden_layers = min(mc.debug.num_layers, cfg.num_hidden_layers)
        return cfg
    if T.known_architecture(mc.name) is not None:
        return T.model_config(mc.name, attn=attn, num_layers=mc.debug.num_layers)
    raise ValueError(f"model {mc.name!r}: no weights on this machine (directory or Hugging Face cache) and not one "
                     f"of the known architectures {sorted(T.MODEL_DIMS) + sorted(T.MODEL_ALIASES)}")


def build_model(config: RLTrainerConfig, device: torch.device, mesh: Any, dtype: torch.dtype | None = None,
                for_training: bool = True, shard: bool = True):
    """The model of `config.model` on `device`; `shard=False` leaves it whole (for `replica.Replica`, or as the
    log-probability model beside one) instead of wrapping it in FSDP2."""
    mc = config.model
    cfg = model_config_of(config)
    where = T.local_checkpoint(mc.name)
    if where is not None:
        from transformers import AutoModelForCausalLM

        model = AutoModelForCausalLM.from_pretrained(where, config=cfg, dtype=dtype or DTYPES[mc.optimization_dtype])
        model.to(device)
    else:
        model = T.get_model(cfg, device, dtype or DTYPES[mc.optimization_dtype])
        log(f"{mc.name}: no weights on this machine, training from a random initialisation")
    if fused_blocks(config):
        from .fused import fuse_model

        log(f"fused block kernels: {fuse_model(model)}")
    if not shard:
        return model, cfg
    if mc.ac is not None:
        from torch.distributed.algorithms._checkpoint.checkpoint_wrapper import checkpoint_wrapper

        for i, (name, block) in enumerate(list(model.model.layers.named_children())):
            if i % mc.ac.freq == 0:
                model.model.layers.register_module(name, checkpoint_wrapper(block, preserve_rng_state=False))
    # `reshard_after_forward = true` is the reference's default and means "not asked for": models that fit stay
    # gathered for a whole step (trainer.keeps_parameters_gathered); `false` is honoured as given
    # A model that only ever runs forwards (the logprob model) must reshard after each one: its weights are replaced
    # between steps, and parameters left gathered would go on serving the old ones.
    reshard = True if not for_training else (None if mc.reshard_after_forward else False)
    T.setup_fsdp(model, mesh, DTYPES[mc.reduce_dtype], reshard)
    return model, cfg


def print_benchmark(history: list[dict[str, float]]) -> dict[str, dict[str, float]]:
    """Mean, std, min, max of throughput and step time over all steps but the first
    (src/prime_rl/trainer/utils.py:133-192)."""
    rows = history[1:]
    out = {}
    for key in ("perf/mfu", "perf/throughput", "time/step", "perf/peak_memory"):
        vals = [r[key] for r in rows if key in r]
        if vals:
            out[key] = {"mean": statistics.fmean(vals), "std": statistics.stdev(vals) if len(vals) > 1 else 0.0,
                        "min": min(vals), "max": max(vals)}
    log("Benchmark (first step excluded)")
    for key, v in out.items():
        log(f"  {key:18s} {v['mean']:.4g} +- {v['std']:.2g} [{v['min']:.4g}, {v['max']:.4g}]")
    return out


def train(config: RLTrainerConfig) -> list[dict[str, float]]:
    world = get_world()
    torch.cuda.set_device(world.local_rank)
    device = torch.device("cuda", world.local_rank)
    if config.model.cp > 1:
        raise ValueError("CP is not supported for RL. No reason it shouldn't, we just didn't test it.")
    mesh = T.init_distributed(device)
    try:
        return _train(config, world, device, mesh)
    finally:
        if dist.is_initialized():
            dist.barrier()
            dist.destroy_process_group()


def freeze_setup_objects() -> None:
    """Takes everything built so far out of the garbage collector's sight (`gc.freeze`).

    The modules, the transformers package and the configs are millions of objects that live until the process
    ends.  A full collection walks all of them: 100-180 ms on the host, every few steps, with the GPU running
    dry meanwhile, to find next to nothing (`tools/gc_pauses.py`).  Frozen, they are skipped, and a full
    collection only looks at what the steps themselves have made."""
    import gc

    gc.collect()
    gc.freeze()


def _train(config: RLTrainerConfig, world: Any, device: torch.device, mesh: Any) -> list[dict[str, float]]:
    log(f"Starting RL trainer on {world}")
    replica = None
    mode = trainer_mode(config)
    if mode != "fsdp":
        from .replica import Replica, eligible

        # a pass is as long as the joined rows of a step get (trainer.MERGE_MAX_TOKENS), or one micro-batch if longer;
        # rollout files say nothing about their rows' length ahead of time: twice the joined length is assumed
        fake = config.data.fake
        tokens = max(T.MERGE_MAX_TOKENS, fake.micro_batch_size * fake.seq_len) if fake else 2 * T.MERGE_MAX_TOKENS
        meta = T.get_model(model_config_of(config), "meta")
        second_model = 2 * sum(p.numel() for p in meta.parameters()) if config.recompute_logprobs else 0
        reason = eligible(meta, config.optim.type, config.model.ac is not None, tokens_per_pass=tokens,
                          world=world.world_size, hbm_bytes=torch.cuda.get_device_properties(device).total_memory,
                          extra_bytes=second_model)
        if reason is None and config.model.ac is not None:
            log("activation checkpointing was asked for and is not needed: a step without it fits this GPU")
        if reason is None:
            model, model_cfg = build_model(config, device, mesh, shard=False)
            graphs = replay_graphs(config)
            replica = Replica(model, lambda m: setup_optimizer(config.optim, m, flat=True), graphs=graphs)
            log("whole model per GPU, master weights and optimizer state sharded"
                + (", micro-batches from HIP graphs" if graphs else ""))
        elif mode == "replica" or config.model.compile is not None:
            log(f"whole-model mode not possible ({reason}); FSDP2")
    if replica is None:
        model, model_cfg = build_model(config, device, mesh)
        optimizer = setup_optimizer(config.optim, model)
    else:
        optimizer = replica.optimizer
    scheduler = setup_scheduler(optimizer, config.scheduler, config.max_steps, config.optim.lr)
    # the weight files are a hand-off that lives for async_level + 1 steps and is read by torch.load only: their
    # zip checksums are left out (PRL_SAVE_CRC32=1 keeps them)
    publisher = WeightPublisher(config.output_dir, config.weights.interval,
                                config.ckpt.interval if config.ckpt else None, config.async_level, device,
                                crc32=os.environ.get("PRL_SAVE_CRC32", "0") == "1")
    loader = FakeDataLoader(config.data.fake) if config.data.fake else DataLoader(config.output_dir, 0)
    logprob_model = None
    if config.recompute_logprobs:
        logprob_model, _ = build_model(config, device, mesh, dtype=torch.bfloat16, for_training=False,
                                       shard=replica is None)
        logprob_model.requires_grad_(False)

    publish = os.environ.get("PRL_PUBLISH", "1") != "0"  # 0: no weight files at all (to time the loop alone)
    snapshots: dict[int, dict[str, torch.Tensor]] = {}
    device_name = perf.describe_device(device)
    counter = None
    history: list[dict[str, float]] = []
    ahead: list[tuple[list[Any], Any]] = []  # at most one entry: the next step's rollouts and their passes
    published = 0  # the last step whose weights have been handed to the publisher
    step = 0
    while config.max_steps is None or step < config.max_steps:
        if step in (0, 2):  # after the set-up, and once more when the first steps have filled every cache
            freeze_setup_objects()
        t_step = time.perf_counter()
        torch.cuda.reset_peak_memory_stats(device)  # the step's own peak, as src/prime_rl/trainer/rl/train.py:159
        t_save = 0.0
        if step > 0 and publish and published < step:
            # weights of step N go to the inference side while step N + 1 trains (the whole-model trainer has
            # started this already, from inside step N: see `when_queued` below)
            t0 = time.perf_counter()
            if replica is not None:
                # the snapshot reads every parameter: all-gathers of the last step's weights may still be in
                # flight.  (A step that publishes nothing does not wait here: its forward waits bucket by bucket,
                # in the modules' pre-hooks, `replica.arrival_hooks`.)
                replica.wait_parameters()
            publisher.publish(model, step)
            published = step
            t_save = time.perf_counter() - t0
        if logprob_model is not None:
            if replica is not None:
                replica.wait_parameters()  # the snapshot reads every parameter
            snapshots[step] = {k: v.detach().clone() for k, v in model.state_dict().items()}
        prepared = None
        if ahead:  # loaded and sorted while the last step's kernels were running
            micro_batches, prepared = ahead.pop()
            t_wait = t_data = 0.0
        else:
            t0 = time.perf_counter()
            loader.wait_for_batch()
            t_wait = time.perf_counter() - t0
            t0 = time.perf_counter()
            micro_batches = loader.get_batch()
            t_data = time.perf_counter() - t0
        t_recompute = 0.0
        errors = None
        if logprob_model is not None:
            t0 = time.perf_counter()
            infer_step = max(step - config.async_level, 0)
            logprob_model.load_state_dict(snapshots[infer_step])
            if infer_step == step - config.async_level:
                del snapshots[infer_step]
            errors = T.recompute_logprobs(logprob_model, micro_batches, device)
            t_recompute = time.perf_counter() - t0
        publisher.guard_parameters()
        samples = (config.data.fake.batch_size // world.world_size) if config.data.fake else None
        lr = optimizer.param_groups[0]["lr"]  # the rate this step is taken with (the scheduler moves on inside)
        if replica is not None:
            more = config.max_steps is None or step + 1 < config.max_steps

            def when_queued() -> None:
                # The step, its optimizer kernels included, has been handed to the GPU and the host is about to
                # wait for its statistics.  What the host would otherwise do with the GPU idle between two steps
                # happens now, beside the GPU's work: the new weights start on their way out (3-4 ms of host work
                # for a 0.6B model's 311 tensors; the copies are ordered behind the optimizer on the stream), and
                # if the next step's rollouts are there already (the orchestrator runs ahead of the trainer,
                # `async_level`), they are read and sorted into passes.
                nonlocal published, t_save
                if publish:
                    t0 = time.perf_counter()
                    replica.wait_parameters()
                    publisher.publish(model, step + 1)
                    published = step + 1
                    t_save += time.perf_counter() - t0
                if more and logprob_model is None and loader.ready():
                    nxt = loader.get_batch()
                    ahead.append((nxt, replica.prepare_step(nxt, config.loss)))

            res = replica.train_step(micro_batches, config.loss, config.optim.max_norm, scheduler, samples=samples,
                                     recomputed_errors=errors, prepared=prepared, before_statistics=when_queued)
        else:
            res = T.train_step(model, optimizer, micro_batches, config.loss, config.optim.max_norm, device, scheduler,
                               samples=samples, recomputed_errors=errors)
        publisher.maybe_clean(step, wait=False)
        seq_len = micro_batches[0]["input_ids"].shape[1]
        if counter is None:
            counter = perf.PerfCounter(model_cfg, seq_len, device_name=device_name, world_size=world.world_size)
        counter.count_tokens(res.tokens * world.world_size)
        step_time = time.perf_counter() - t_step
        throughput = counter.get_tokens_per_second() or 0.0
        row = {"step": step, "time/step": step_time, "time/forward_backward": res.seconds,
               "time/wait_for_batch": t_wait, "time/load_data": t_data,
               "time/save_weights": t_save, "time/compute_logprobs": t_recompute,
               "perf/throughput": throughput, "perf/throughput_per_gpu": throughput / world.world_size,
               "perf/mfu": counter.get_mfu() or 0.0,
               "perf/peak_memory": torch.cuda.max_memory_allocated(device) / 1024**3,
               "optim/lr": lr, "optim/grad_norm": res.grad_norm, **res.stats}
        history.append(row)
        log(f"Step {step} | Time: {step_time:.2f}s | Loss: {res.stats.get('loss/mean', float('nan')):.4f} | "
            f"Entropy: {res.stats.get('entropy/mean', float('nan')):.4f} | "
            f"Importance Ratio: {res.stats.get('importance_ratio/mean', float('nan')):.4f} | "
            f"Grad. Norm: {res.grad_norm:.4f} | LR: {row['optim/lr']:.2e} | "
            f"Throughput: {row['perf/throughput']:.0f} tokens/s | MFU: {row['perf/mfu']:.1f}% | "
            f"Peak Mem.: {row['perf/peak_memory']:.1f} GiB")
        step += 1
    if replica is not None:
        replica.wait_parameters()
    if step > 0 and publish and published < step:  # the final weights, as the orchestrator waits for them
        publisher.publish(model, step)
    publisher.close()
    if config.bench and len(history) > 1:
        print_benchmark(history)
    return history


def main() -> None:
    train(parse_argv(RLTrainerConfig))


if __name__ == "__main__":
    main()
