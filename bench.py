"""Benchmark of the policy-update hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

One step = one pass of the hot path over one training batch of the Qwen3-0.6B
reverse_text GRPO recipe (128 rollouts of 128 tokens, 16 rollouts per
micro-batch: 8 packed rows of T = 2048 tokens, vocabulary V = 151936).  Per
packed row the path takes the model's logits, already resident in HBM, to the
loss, the logged tensors and the gradient with respect to the logits:

    per-token logprob + entropy + dL/dlogits in ONE pass over the logits (bf16; a row is held in a
    compute unit's registers between its read and the write of its gradient)
    -> sample boundaries, ratio-clip loss and logged tensors -> statistics

(fp32 logits, --logits-dtype f32, take two passes: logprob + entropy, then dL/dlogits.)

Every rank runs the same amount of work on its own rollouts (data parallel, no
collective on the data path); the reported value is the job total.  Rank 0
prints ONE JSON line.
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import threading
import time
from pathlib import Path

REPO = Path(__file__).resolve().parent
if str(REPO) not in sys.path:
    sys.path.insert(0, str(REPO))

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

# Qwen3-0.6B reverse_text GRPO (configs/reverse_text/rl/orch.toml of the reference)
VOCAB = 151936
SEQ_LEN = 128
MICRO_BATCH = 16
BATCH = 128
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak
MFMA_PEAK_TFLOPS = 2500.0  # MI355X dense bf16 matrix peak
EXIT_TRAINER_EXTRA_HUNG = 3  # exit code when the whole-trainer-step extra (or the barrier behind it) never returned


# recipe -> (TOML under tools/recipes, what it stands for)
RECIPES = {
    "reverse_text": ("reverse_text_fake.toml", "Qwen3-0.6B reverse_text GRPO, rows of 16 x 128 tokens"),
    "wordle": ("wordle_1p7b_fake.toml", "Qwen3-1.7B wordle GRPO, rows of 1 x 4096 tokens, 32 rollouts per rank and step"),
    "hendrycks_7b": ("hendrycks_7b_fake.toml", "Qwen2.5-7B hendrycks_math GRPO, rows of 2 x 2048 tokens, activation "
                                               "checkpointing asked for (the trainer drops it where a step without "
                                               "it fits the GPU), 32 rollouts per rank and step"),
    "hendrycks_32b_8_layers": ("hendrycks_32b_8_layers_fake.toml",
                               "Qwen2.5-32B hendrycks_math GRPO cut to 8 of its 64 blocks (what one GPU holds), rows of "
                               "8 x 2048 tokens, activation checkpointing, 32 rollouts per rank and step"),
}


def parse(argv: list[str] | None = None) -> argparse.Namespace:
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=10)
    p.add_argument("--warmup", type=int, default=2)
    p.add_argument("--logits-dtype", choices=["bf16", "f32"], default="bf16",
                   help="storage type of the logits and of their gradient (arithmetic is fp32 either way)")
    p.add_argument("--data", choices=["near", "fake"], default="near",
                   help="near: old logprobs close to the new ones, advantages +-0.5 (every token has a gradient); "
                        "fake: the reference's FakeDataLoader distribution")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU time budget of the baseline sample")
    p.add_argument("--no-stats", action="store_true", help="leave the logged statistics out of the step")
    p.add_argument("--event-pairs", action="store_true",
                   help="time the dominant kernel with an event recorded on either side of its call instead of with "
                        "events stamped by the launch itself")
    p.add_argument("--workload", choices=["epilogue", "trainer"], default="epilogue",
                   help="epilogue: the hot path from resident logits (the headline line); trainer: the whole "
                        "optimizer step with a random-init Qwen3-0.6B body under FSDP2")
    p.add_argument("--no-trainer-extra", action="store_true",
                   help="do not append the short whole-trainer-step measurement to the line")
    p.add_argument("--trainer-steps", type=int, default=3)
    p.add_argument("--trainer-limit", type=float, default=180.0,
                   help="with --gpus N > 1: seconds after which the line is printed without the whole-trainer-step "
                        "extra and the process ends")
    p.add_argument("--recipe", choices=sorted(RECIPES), default="reverse_text",
                   help="reverse_text: the headline (the hot path of the Qwen3-0.6B recipe from resident logits, plus "
                        "its whole trainer step as an extra). wordle / hendrycks_7b / hendrycks_32b_8_layers: the whole trainer step of that "
                        "recipe's row shape through the trainer entry point (tools/recipes/*.toml: synthetic "
                        "rollouts, 32 rollouts per rank and step, weights published every step); the line then "
                        "reports that step and names the recipe in config.workload")
    p.add_argument("--launch-check", action="store_true",
                   help="ranks only meet, all-reduce one number and rank 0 prints what it saw: a test of the launch "
                        "path that needs no GPU (with PRL_BENCH_BACKEND=gloo)")
    p.add_argument("--trainer-impl", choices=["hf", "fused", "replica"], default="replica",
                   help="whole-trainer-step workload: hf = the transformers blocks as they are under FSDP2; fused = "
                        "their RMSNorm / rotary / SwiGLU on this package's HIP kernels (the reference's model.impl = "
                        "liger_kernel); replica = fused, the whole model on every GPU with sharded master weights. "
                        "(Replaying the passes from HIP graphs is not offered: library calls record memset nodes "
                        "into a captured pass, and such nodes have given wrong results on replay.)")
    p.add_argument("--no-paths", action="store_true",
                   help="leave out the short measurements of the other product paths (fp32 logits, two passes, one "
                        "ratio per sample, the reference's fake-data distribution) and of the LM head's products")
    return p.parse_args(argv)


def make_batch(dev, dtype, kind: str, seed: int):
    """Synthetic micro-batches shaped like the reference's FakeDataLoader
    (src/prime_rl/trainer/rl/data.py:39-54), generated on the device."""
    import prime_rl_cse_amd as prl

    g = torch.Generator(device=dev).manual_seed(seed)
    t = MICRO_BATCH * SEQ_LEN
    rows = []
    for _ in range(BATCH // MICRO_BATCH):
        logits = torch.empty(1, t, VOCAB, dtype=dtype, device=dev)
        for r in range(0, t, 512):
            logits[0, r : r + 512] = (torch.randn(512, VOCAB, generator=g, device=dev) * 2.5).to(dtype)
        ids = torch.randint(0, 100, (1, t), generator=g, device=dev)
        pos = torch.cat([torch.arange(SEQ_LEN, device=dev)] * MICRO_BATCH).unsqueeze(0)
        mask = torch.ones(1, t, dtype=torch.bool, device=dev)
        if kind == "fake":
            adv = torch.randn(1, t, generator=g, device=dev)
            old = torch.randn(1, t, generator=g, device=dev)
        else:
            adv = torch.where(torch.rand(1, t, generator=g, device=dev) < 0.5, -0.5, 0.5)
            with torch.no_grad():
                lp = prl.selective_log_softmax(prl.shift_logits(logits), ids)
            old = lp + 0.05 * torch.randn(1, t, generator=g, device=dev)
        rows.append({"logits": logits, "input_ids": ids, "position_ids": pos, "advantages": adv,
                     "logprobs": old, "loss_mask": mask, "temperature": 1.0})
    return rows


def run_step(rows, cfg, with_stats: bool):
    """The hot loop of src/prime_rl/trainer/rl/train.py:264-333 and :364 from the
    logits onward."""
    import prime_rl_cse_amd as prl
    from prime_rl_cse_amd.trainer import add_ratio_statistics

    loss_scale = sum(int(r["loss_mask"].numel()) for r in rows)  # masks are all ones: no sync needed to count them
    tensors = prl.Tensors()
    out = None
    for r in rows:
        logits = r["logits"].requires_grad_(True)
        logits.grad = None
        loss, aux = prl.policy_loss(logits, r["input_ids"], r["position_ids"], r["advantages"], r["logprobs"],
                                    r["loss_mask"], r["temperature"], cfg, loss_scale)
        loss.backward(prl.unit_gradient(loss.device))
        out = logits.grad
        if with_stats:
            m = r["loss_mask"]
            tensors.add("probs", aux["logprobs"], m, exp=True)
            tensors.add("old_probs", r["logprobs"], m, exp=True)
            tensors.add("entropy", aux["entropy"], m)
            tensors.add("loss", loss.detach().reshape(1))
            add_ratio_statistics(tensors, aux, m, cfg)  # per token, or per sample with one ratio per sample
    # the statistics are reduced on the device inside the step; their 5 floats per key reach the host
    # asynchronously and are read one step later, as a trainer that logs step N during step N + 1 would
    stats = tensors.compute_stats_async() if with_stats else None
    return out, stats


def measure_path(rows, cfg, with_stats: bool, steps: int, dominant: str, bytes_per_launch: float, env: dict | None = None
                 ) -> dict:
    """A short measurement of one product path over the same 8 packed rows: `steps` timed steps after one warm-up
    step, then one more step with an event on either side of every library call for the kernel times.  `env`
    holds switches of the package set for the duration (they are read at every call)."""
    from prime_rl_cse_amd import _capi

    old = {k: os.environ.get(k) for k in (env or {})}
    os.environ.update(env or {})
    try:
        _, fut = run_step(rows, cfg, with_stats)
        if fut is not None:
            fut.result()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            _, fut = run_step(rows, cfg, with_stats)
            if fut is not None:
                fut.result()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / steps * 1e3
        side = _capi.KernelTimer()
        _capi.set_timer(side)
        _, fut = run_step(rows, cfg, with_stats)
        if fut is not None:
            fut.result()
        torch.cuda.synchronize()
        _capi.set_timer(None)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    k = side.summary()
    out = {"ms_per_step": ms, "samples_per_s": BATCH / (ms * 1e-3), "steps": steps,
           "calls": {name: round(v["mean_ms"], 4) for name, v in sorted(k.items())}}
    if dominant in k:
        gbs = bytes_per_launch / (k[dominant]["mean_ms"] * 1e-3) / 1e9
        out["dominant"] = {"call": dominant, "mean_ms": k[dominant]["mean_ms"], "GBps": gbs, "frac": gbs / HBM_PEAK_GBS,
                           "bytes_per_launch": bytes_per_launch,
                           "timed_by": "HIP events recorded on the launch stream on either side of the call"}
    return out


def lm_head_measure(dev, t: int = 16384, h: int = 1024, v: int = VOCAB, reps: int = 3, rounds: int = 3) -> dict:
    """The LM head's three products at the shape the whole-model trainer step runs them (the step's 8 rows joined:
    16384 x 1024 against the 151936 x 1024 table): this package's MFMA kernels and the BLAS library's product
    through torch, interleaved in random order, HIP events around `reps` back-to-back calls, medians over the
    rounds.  Random operands (zero-filled ones run at a higher clock)."""
    import random

    from prime_rl_cse_amd import lm_head as L

    gen = torch.Generator(device=dev).manual_seed(0)
    x = torch.randn((t, h), generator=gen, device=dev).to(torch.bfloat16)
    w = (torch.randn((v, h), generator=gen, device=dev) * 0.05).to(torch.bfloat16)
    g = torch.empty((t, v), dtype=torch.bfloat16, device=dev)
    for r in range(0, t, 2048):
        g[r:r + 2048] = (torch.randn((min(2048, t - r), v), generator=gen, device=dev) * 0.01).to(torch.bfloat16)
    ours = {"forward": lambda: L.lm_head_fwd(x, w), "input_gradient": lambda: L.lm_head_bwd_input(g, w),
            "weight_gradient": lambda: L.lm_head_bwd_weight(g, x)}
    lib = {"forward": lambda: torch.nn.functional.linear(x, w), "input_gradient": lambda: g @ w,
           "weight_gradient": lambda: g.t() @ x}

    def ms(fn) -> float:
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        b.record()
        b.synchronize()
        return a.elapsed_time(b) / reps

    for k in ours:
        ours[k]()
        lib[k]()
    torch.cuda.synchronize()
    times = {(k, who): [] for k in ours for who in ("kernel", "library")}
    rnd = random.Random(0)
    for _ in range(rounds):
        order = list(times)
        rnd.shuffle(order)
        for k, who in order:
            times[(k, who)].append(ms(ours[k] if who == "kernel" else lib[k]))
    flop = 2.0 * t * h * v
    products = {}
    for k in ours:
        a, b = sorted(times[(k, "kernel")]), sorted(times[(k, "library")])
        ma, mb = a[len(a) // 2], b[len(b) // 2]
        products[k] = {"mean_ms": ma, "tflops": flop / ma / 1e9, "frac": flop / ma / 1e9 / MFMA_PEAK_TFLOPS,
                       "library_ms": mb, "library_tflops": flop / mb / 1e9, "kernel_over_library": mb / ma}
    plan = L.plan(t, h, v)
    best = max(products, key=lambda k: products[k]["frac"])
    return {"bound": "mfma", "kernel": "k_lm_gemm", "unit": "TFLOP/s", "peak": MFMA_PEAK_TFLOPS,
            "flops": flop, "achieved": products[best]["tflops"], "frac": products[best]["frac"],
            "mean_ms": products[best]["mean_ms"], "product": best, "products": products,
            "shape": {"T": t, "H": h, "V": v},
            "in_the_trainer_by_default": {"forward": plan[0], "input_gradient": plan[1], "weight_gradient": plan[2]},
            "timed_by": f"HIP events around {reps} back-to-back calls, kernel and library interleaved in random "
                        f"order, median of {rounds} rounds, random bf16 operands"}


def other_paths(dev, rows, cfg, with_stats, args, ms_per_step, dom, achieved, dom_bytes) -> dict:
    """The product paths the headline does not take, a few steps each over the same shape (8 packed rows of
    T = 2048, V = 151936): two passes over bf16 logits (what the one-pass kernel replaces; also what logits it does
    not take fall back to), one ratio per sample (src/prime_rl/trainer/rl/loss.py:92-96, 107-108), fp32 logits
    (the reference's own storage type, loss.py:26-30 after train.py:285)."""
    import prime_rl_cse_amd as prl

    t = MICRO_BATCH * SEQ_LEN
    paths = {"token_bf16": {"ms_per_step": ms_per_step, "samples_per_s": BATCH / (ms_per_step * 1e-3),
                            "dominant": {"call": "prl_policy_rows_fwd_bwd", "mean_ms": dom["mean_ms"], "GBps": achieved,
                                         "frac": achieved / HBM_PEAK_GBS, "bytes_per_launch": dom_bytes},
                            "note": "the headline"}}
    bwd2 = (t - 1) * VOCAB * 2 + t * VOCAB * 2
    paths["token_bf16_two_pass"] = measure_path(rows, cfg, with_stats, 3, "prl_logprob_bwd", bwd2,
                                                env={"PRL_FUSED_ROWS": "0"})
    seq_cfg = prl.LossConfig(ratio_type="sequence")
    paths["sequence_bf16"] = measure_path(rows, seq_cfg, with_stats, 3, "prl_policy_rows_seq_fwd_bwd", bwd2 + 12 * t)
    # fp32 logits: the same values widened (8 rows of 1.24 GB); the bf16 rows are not needed meanwhile
    wide = []
    for r in rows:
        w = dict(r)
        w["logits"] = r["logits"].detach().float()
        wide.append(w)
    paths["token_f32"] = measure_path(wide, cfg, with_stats, 3, "prl_logprob_bwd", (t - 1) * VOCAB * 4 + t * VOCAB * 4)
    del wide
    torch.cuda.empty_cache()
    return paths


def cpu_baseline(rows, seconds: float) -> dict:
    """The CPU oracle (a restatement of the reference's torch code, which runs
    the same torch CPU kernels) on a bounded sample of the same workload."""
    from oracle import ref_cpu as O

    threads = torch.get_num_threads()
    host = [{k: (v.detach().cpu() if torch.is_tensor(v) else v) for k, v in rows[i % len(rows)].items()}
            for i in range(2)]
    scale = sum(int(r["loss_mask"].numel()) for r in rows)
    done, t0 = 0, time.perf_counter()
    while True:
        r = host[done % len(host)]
        O.policy_epilogue(r["logits"], r["input_ids"], r["position_ids"], r["advantages"], r["logprobs"],
                          r["loss_mask"], r["temperature"], O.LossConfig(), scale)
        done += 1
        el = time.perf_counter() - t0
        if el >= seconds or done >= 8:
            break
    return {
        "value": done * MICRO_BATCH / el,
        "unit": "samples/s",
        "cores": threads,
        "kind": "port",
        "sample": f"{done} packed rows of T={MICRO_BATCH * SEQ_LEN}, V={VOCAB} (logits -> loss, entropy, dlogits) "
                  f"in {el:.1f} s with torch CPU kernels on {threads} threads",
    }


def pmc_traffic(kernel: str, logits_dtype: str) -> tuple[float | None, str | None]:
    """HBM bytes per launch of `kernel` from the newest committed PMC summary whose shape matches."""
    best = None
    for path in sorted((REPO / "profiles").glob("*pmc_traffic.json")):
        try:
            d = json.loads(path.read_text())
        except Exception:
            continue
        sh = d.get("shape", {})
        if sh.get("T") == MICRO_BATCH * SEQ_LEN and sh.get("V") == VOCAB and sh.get("logits") == logits_dtype:
            for name, vals in d.get("kernels", {}).items():
                if name.startswith(kernel):
                    best = (float(vals["hbm_bytes_per_launch"]), f"profiles/{path.name}")
    return best if best else (None, None)


def trainer_measure(dev, world: int, steps: int, warmup: int, cpu: bool, cpu_seconds: float,
                    impl: str = "replica") -> dict:
    """The whole optimizer step of the Qwen3-0.6B reverse_text recipe on synthetic rollouts: transformer body by
    PyTorch-ROCm (random init, bf16 compute under FSDP2, fp32 master weights and gradient reduction), everything
    after the LM head by this package, gradient clipping and AdamW (src/prime_rl/trainer/rl/train.py:259-352)."""
    import prime_rl_cse_amd as prl
    from prime_rl_cse_amd import trainer as T

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29577")
    os.environ.setdefault("RANK", "0")
    os.environ.setdefault("WORLD_SIZE", "1")
    mesh = T.init_distributed(dev) if impl in ("hf", "fused") else None  # the replica reduces over the default group
    cfg = T.model_config("Qwen/Qwen3-0.6B")
    model = T.get_model(cfg, dev, torch.float32, seed=0)
    fused_modules = {}
    if impl != "hf":
        from prime_rl_cse_amd import fused

        fused_modules = fused.fuse_model(model)
    fsdp_groups = T.blocks_per_fsdp_group(model)
    replica = None
    if impl == "replica":  # whole model per GPU
        from prime_rl_cse_amd.replica import Replica

        from prime_rl_cse_amd.optim import FlatAdamW

        replica = Replica(model, lambda m: FlatAdamW(m.parameters(), lr=1e-6, weight_decay=0.01), graphs=False)
        opt = replica.optimizer

        def step(mbs):
            return replica.train_step(mbs, loss_cfg, samples=BATCH)
    else:
        T.setup_fsdp(model, mesh)
        opt = T.setup_optimizer(model)

        def step(mbs):
            return T.train_step(model, opt, mbs, loss_cfg, device=dev, samples=BATCH)
    prl.reset_world()
    loader = prl.FakeDataLoader(prl.FakeDataLoaderConfig(batch_size=BATCH * world, micro_batch_size=MICRO_BATCH,
                                                         seq_len=SEQ_LEN))
    loss_cfg = prl.LossConfig()
    batches = [loader.get_batch() for _ in range(2)]
    for i in range(warmup):
        step(batches[i % 2])
    from prime_rl_cse_amd.train import freeze_setup_objects

    freeze_setup_objects()  # as the trainer's entry point does after its first steps
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    per_step = []
    t0 = time.perf_counter()
    for i in range(steps):
        res = step(batches[i % 2])
        per_step.append(res.seconds)  # train_step ends with the statistics read, a host wait for the whole step
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64,
                      device=dev if world == 1 or dist.get_backend() == "nccl" else "cpu")  # gloo: host tensors only
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    sec = float(el.item()) / steps
    flop = prl.perf.flop_per_token(cfg, SEQ_LEN) * BATCH * SEQ_LEN
    out = {
        "model": "Qwen3-0.6B (random init, literal dimensions)",
        "ms_per_step": sec * 1e3,
        "ms_each_step": [round(1e3 * x, 1) for x in per_step],
        "samples_per_s": world * BATCH / sec,
        "tokens_per_s": world * BATCH * SEQ_LEN / sec,
        "mfu_percent_of_2.5PF": 100 * flop / sec / 2.5e15,
        "steps": steps,
        "loss": res.loss,
        "grad_norm": res.grad_norm,
        "precision": "bf16 compute, fp32 master weights and gradient reduction, fp32 loss math",
        "attention": "sdpa" if os.environ.get("PRL_ATTN_SHORT") == "0" else "short-sample MFMA kernels (csrc/attn_short.hip)",
        "micro_batches_per_step": len(batches[0]),
        "forward_backward_passes_per_step": res.passes,
        "blocks": "transformers Qwen3" + (f", modules on HIP kernels: {fused_modules}" if fused_modules else ""),
    }
    if fused_modules:
        # which matrix products of the blocks' linear layers ran on the package's MFMA kernels (the rest: BLAS library)
        from prime_rl_cse_amd import lm_head

        rows = MICRO_BATCH * SEQ_LEN * len(batches[0]) // max(res.passes, 1)
        names = ("forward", "input_gradient", "weight_gradient")
        hq = cfg.num_attention_heads * cfg.head_dim
        qkv = hq + 2 * cfg.num_key_value_heads * cfg.head_dim
        out["layer_products_on_mfma_kernels"] = {
            f"{what} {n_in}->{n_out}": [n for n, on in zip(names, lm_head.layer_plan(rows, n_in, n_out)) if on]
            for what, n_in, n_out in (("q+k+v", cfg.hidden_size, qkv), ("o", hq, cfg.hidden_size),
                                      ("gate+up", cfg.hidden_size, 2 * cfg.intermediate_size),
                                      ("down", cfg.intermediate_size, cfg.hidden_size))}
    if replica is None:
        out["fsdp_blocks_per_group"] = fsdp_groups
    else:
        out["data_parallel"] = ("whole bf16 model and fp32 gradient sum per GPU, fp32 master weights and AdamW state "
                                "sharded; per step one reduce-scatter and one all-gather per bucket, the "
                                "reduce-scatters started from inside the last backward pass")
        out["exchange"] = {"buckets": len(replica.state.buckets), "overlapped_with_backward": replica.overlap,
                           "buckets_reduced_inside_backward": replica.bucket_launches}
        out["micro_batches"] = {"replayed_from_hip_graphs": replica.replays, "run_eagerly": replica.eager_runs}
        if replica.capture_errors:
            out["capture_errors"] = list(replica.capture_errors.values())
    del model, opt
    torch.cuda.empty_cache()
    if cpu and world == 1:
        out["cpu_baseline"] = trainer_cpu_baseline(batches[0], cpu_seconds)
    return out


def recipe_measure(recipe: str, world: int, steps: int, warmup: int) -> dict:
    """`steps` timed optimizer steps of another recipe's shape through the trainer entry point itself
    (prime_rl_cse_amd.train, the loop of src/prime_rl/trainer/rl/train.py:55-466): random-init model of the
    recipe's architecture, FakeDataLoader rollouts, weight publication every step as in a real run."""
    import statistics
    import tempfile

    from prime_rl_cse_amd.settings import RLTrainerConfig, parse_argv
    from prime_rl_cse_amd.train import train

    toml, what = RECIPES[recipe]
    base = parse_argv(RLTrainerConfig, ["@", str(REPO / "tools" / "recipes" / toml)], use_env=False)
    per_rank = base.data.fake.batch_size
    with tempfile.TemporaryDirectory(prefix="prl_bench_") as out:
        cfg = parse_argv(RLTrainerConfig, ["@", str(REPO / "tools" / "recipes" / toml), "--max-steps",
                                           str(steps + warmup), "--output-dir", out, "--data.fake.batch-size",
                                           str(per_rank * world)], use_env=False)
        import contextlib

        with contextlib.redirect_stdout(sys.stderr):  # the entry point logs every step; stdout is for the one line
            history = train(cfg)  # starts and ends its own process group
    rows = history[warmup:]
    step_s = [r["time/step"] for r in rows]
    fb_s = [r["time/forward_backward"] for r in rows]
    sec = statistics.fmean(step_s)
    tokens = per_rank * world * cfg.data.fake.seq_len
    return {
        "recipe": recipe, "what": what, "model": cfg.model.name, "samples_per_s": per_rank * world / sec,
        "ms_per_step": 1e3 * sec, "ms_each_step": [round(1e3 * x, 1) for x in step_s],
        "ms_forward_backward": 1e3 * statistics.fmean(fb_s), "tokens_per_s": tokens / sec,
        "mfu_percent": statistics.fmean(r["perf/mfu"] for r in rows),
        "peak_memory_gib": max(r["perf/peak_memory"] for r in rows),
        "rollouts_per_rank_and_step": per_rank, "micro_batch_size": cfg.data.fake.micro_batch_size,
        "seq_len": cfg.data.fake.seq_len, "loss": rows[-1].get("loss/mean"), "grad_norm": rows[-1]["optim/grad_norm"],
        "timing": "mean of the entry point's own time/step over the timed steps (each ends with the host reading "
                  "the step's statistics, so it spans the step's GPU work); the first steps are left out as warm-up",
    }


def trainer_cpu_baseline(batch, seconds: float) -> dict:
    """The same step on the host cores: fp32 model on the CPU, the oracle's loss, AdamW; a bounded sample of
    micro-batches (the reference trainer itself refuses to run without CUDA)."""
    from oracle import ref_cpu as O
    from prime_rl_cse_amd import trainer as T

    cfg = T.model_config("Qwen/Qwen3-0.6B")
    model = T.get_model(cfg, "cpu", torch.float32, seed=0)
    opt = T.setup_optimizer(model)
    scale = sum(int(mb["loss_mask"].sum()) for mb in batch)
    done, t0 = 0, time.perf_counter()
    for mb in batch:
        logits = T.forward(model, mb["input_ids"], mb["position_ids"]).float().contiguous()
        x = O.shift_logits(logits) / mb["temperature"]
        lp = O.selective_log_softmax(x, mb["input_ids"])
        lens = O.get_response_lengths(mb["position_ids"])
        loss, _ = O.compute_loss(lp.squeeze().split(lens), mb["logprobs"].squeeze().split(lens),
                                 mb["advantages"].squeeze().split(lens), mb["loss_mask"].squeeze().split(lens),
                                 O.LossConfig(), scale)
        with torch.no_grad():
            O.compute_entropy(x)
        del logits, x
        loss.backward()
        done += 1
        if time.perf_counter() - t0 >= seconds:
            break
    torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
    opt.step()
    el = time.perf_counter() - t0
    return {"value": done * MICRO_BATCH / el, "unit": "samples/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{done} micro-batches of {MICRO_BATCH} x {SEQ_LEN} tokens through an fp32 Qwen3-0.6B on the CPU "
                      f"(forward, oracle loss, backward) plus one clip + AdamW step, {el:.1f} s"}


def free_port() -> int:
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return int(s.getsockname()[1])


def rank_commands(n: int, argv: list[str], port: int, environ: dict[str, str] | None = None
                  ) -> list[tuple[list[str], dict[str, str]]]:
    """(argv, environment) of the N rank processes of `python bench.py --gpus N ...` on this node: this same
    script under a fresh interpreter, one per GPU, with the rendezvous variables torch.distributed.run would
    set (the reference starts its trainer ranks through torchrun from its launcher,
    src/prime_rl/rl.py:589-623)."""
    base = dict(os.environ if environ is None else environ)
    out = []
    for r in range(n):
        env = {**base, "RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n), "LOCAL_WORLD_SIZE": str(n),
               "GROUP_RANK": "0", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "PYTHONUNBUFFERED": "1",
               "PRL_BENCH_LAUNCHED": "1"}
        # the host driver of these nodes shares device memory between processes through dmabuf only: without this
        # RCCL's set-up fails in hipIpcGetMemHandle (kept as it is when the caller has set it)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        out.append(([sys.executable, "-u", str(Path(__file__).resolve()), *argv], env))
    return out


def launch_ranks(n: int, argv: list[str], grace: float = 30.0) -> int:
    """Starts the N ranks as children, passes rank 0's JSON line through to stdout (everything else the ranks
    print goes to stderr) and returns the worst exit code.  When a rank fails the others get `grace` seconds
    to end by themselves and are then killed, as the reference's launcher ends every process once one has
    failed (src/prime_rl/rl.py:641-651).  This process makes no GPU call: the ranks are fresh processes."""
    import subprocess

    procs = []
    try:
        for r, (cmd, env) in enumerate(rank_commands(n, argv, free_port())):
            procs.append(subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE if r == 0 else sys.stderr,
                                          stderr=sys.stderr, text=True))
    except BaseException:  # a rank that could not be started: the ones that were must not be left waiting for it
        for p in procs:
            p.kill()
        raise

    def relay() -> None:
        for text in procs[0].stdout:
            is_line = False
            if text.startswith("{"):
                try:
                    is_line = "metric" in json.loads(text)
                except ValueError:
                    pass
            (sys.stdout if is_line else sys.stderr).write(text)
            (sys.stdout if is_line else sys.stderr).flush()

    reader = threading.Thread(target=relay, daemon=True)
    reader.start()
    failed_at = None
    while any(p.poll() is None for p in procs):
        if failed_at is None and any(p.poll() not in (None, 0) for p in procs):
            failed_at = time.monotonic()
        if failed_at is not None and time.monotonic() - failed_at > grace:
            for p in procs:
                if p.poll() is None:
                    p.kill()
        time.sleep(0.2)
    reader.join(timeout=10)
    codes = [p.returncode for p in procs]
    worst = max((128 - c if c < 0 else c) for c in codes)  # a rank ended by signal S counts as 128 + S
    if worst:
        print(f"bench.py: rank exit codes {codes}", file=sys.stderr, flush=True)
    return worst


def launch_check(rank: int, world: int, backend: str, local: int) -> None:
    """The ranks meet, add up their ranks and rank 0 prints a line.  With RCCL the number travels through the
    GPUs; with gloo nothing touches one."""
    if backend == "nccl":
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        x = torch.tensor([float(rank)], device=torch.device("cuda", local))
    else:
        dist.init_process_group(backend)
        x = torch.tensor([float(rank)])
    dist.all_reduce(x)
    names = [None] * world
    dist.all_gather_object(names, f"rank {rank}: " + (torch.cuda.get_device_name(local) if backend == "nccl" else "cpu"))
    if rank == 0:
        print(json.dumps({"metric": "launch_check", "value": float(x.item()), "unit": "sum of ranks",
                          "n_gpus": world, "backend": dist.get_backend(), "ranks": dist.get_world_size(),
                          "rccl_ranks": dist.get_world_size() if backend == "nccl" else 0, "devices": names,
                          "gpu_initialised": torch.cuda.is_initialized()}), file=LINE_OUT, flush=True)
    dist.barrier()
    dist.destroy_process_group()


LINE_OUT = sys.stdout  # where the one JSON line goes (main() points it at the real stdout)


def keep_stdout_for_the_line() -> None:
    """From here on file descriptor 1 of this process is its stderr, and `LINE_OUT` is the stdout it had: the
    contract is ONE JSON line on stdout, and libraries print there too (RCCL writes a five-line version banner
    to stdout when its first communicator is made, i.e. in every run with more than one rank)."""
    global LINE_OUT
    sys.stdout.flush()
    LINE_OUT = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)


def main(argv: list[str] | None = None) -> int:
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` by itself: this process only starts the ranks and waits for them
        return launch_ranks(args.gpus, argv)
    keep_stdout_for_the_line()
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # one rank per GPU over RCCL.  PRL_BENCH_BACKEND=gloo is a rehearsal aid for boxes with fewer GPUs than
    # ranks: ranks then share devices round-robin and exchange through host memory.
    backend = os.environ.get("PRL_BENCH_BACKEND", "nccl")
    if args.launch_check:
        launch_check(rank, world, backend, local)
        return 0
    if args.recipe != "reverse_text":
        for k, v in dict(MASTER_ADDR="127.0.0.1", MASTER_PORT="29579", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
                         LOCAL_WORLD_SIZE="1").items():
            os.environ.setdefault(k, v)
        info = recipe_measure(args.recipe, world, args.steps, max(args.warmup, 2))
        if rank == 0:
            print(json.dumps({
                "metric": "policy_update_samples_per_sec", "value": info["samples_per_s"], "unit": "samples/s",
                "n_gpus": world, "steps": args.steps, "warmup": max(args.warmup, 2), "ms_per_step": info["ms_per_step"],
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
                "config": {"workload": f"whole trainer step through the entry point, recipe {args.recipe}: {info['what']}",
                           "global_batch": world * info["rollouts_per_rank_and_step"], "seq_len": info["seq_len"],
                           "micro_batch": info["micro_batch_size"], "parallelism": f"dp{world}"},
                "trainer_step": info}), file=LINE_OUT, flush=True)
        return 0
    if backend == "nccl" and local >= torch.cuda.device_count():
        raise SystemExit(f"bench.py: rank {rank} wants GPU {local}, but this process sees {torch.cuda.device_count()} "
                         f"(--gpus {args.gpus} needs one GPU per rank; PRL_BENCH_BACKEND=gloo shares GPUs for rehearsals)")
    local = local % torch.cuda.device_count() if backend != "nccl" else local
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    import prime_rl_cse_amd as prl
    from prime_rl_cse_amd import _capi, perf  # noqa: F401

    if args.workload == "trainer":
        info = trainer_measure(dev, world, args.steps, args.warmup, not args.no_cpu_baseline, args.cpu_seconds,
                               args.trainer_impl)
        if rank == 0:
            line = {"metric": "policy_update_samples_per_sec", "value": info["samples_per_s"], "unit": "samples/s",
                    "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": info["ms_per_step"],
                    "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16",
                    "data": "synthetic",
                    "config": {"workload": "whole trainer step, Qwen3-0.6B reverse_text GRPO, synthetic rollouts",
                               "global_batch": world * BATCH, "seq_len": SEQ_LEN, "micro_batch": MICRO_BATCH,
                               "parallelism": f"fsdp{world}"},
                    "trainer_step": info}
            print(json.dumps(line), file=LINE_OUT, flush=True)
        if dist.is_initialized():
            dist.barrier()
            dist.destroy_process_group()
        return 0

    dtype = torch.bfloat16 if args.logits_dtype == "bf16" else torch.float32
    es = 2 if dtype == torch.bfloat16 else 4
    cfg = prl.LossConfig()
    rows = make_batch(dev, dtype, args.data, 1234 + rank)
    with_stats = not args.no_stats

    for _ in range(args.warmup):
        _, pending = run_step(rows, cfg, with_stats)
        if pending is not None:
            pending.result()

    # inside the timed region only the dominant kernel carries events (every event pair costs a few
    # microseconds of queue time); the other kernels are timed in a short untimed pass afterwards
    # ... and of its launches one in seven (with 8 per step, a sample that walks over all 8 rows; an event pair
    # holds the queue up for ~11 us, which at one in three was 1.6 % of the step)
    # The single-pass kernel is timed by its own start and end on the GPU (events the launch itself stamps:
    # the duration a profiler's kernel trace shows); --event-pairs records an event on either side of the call
    # instead, which is also how the two-pass backward is timed.
    stamps = 0 if args.event_pairs else args.steps * len(rows) // 7 + 2
    timer = _capi.KernelTimer(only={"prl_logprob_bwd", "prl_policy_rows_fwd_bwd"}, every=7, kernel_stamps=stamps,
                              device=dev)
    _capi.set_timer(timer)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pending, last_stats = None, {}
    for _ in range(args.steps):
        _, fut = run_step(rows, cfg, with_stats)
        if pending is not None:
            last_stats = pending.result()  # step N-1's numbers, while step N runs
        pending = fut
    if pending is not None:
        last_stats = pending.result()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    side = _capi.KernelTimer()
    _capi.set_timer(side)
    for _ in range(3):
        _, fut = run_step(rows, cfg, with_stats)
        if fut is not None:
            fut.result()
    torch.cuda.synchronize()
    _capi.set_timer(None)

    el = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    elapsed = float(el.item())
    ms_per_step = elapsed / args.steps * 1e3
    value = world * BATCH / (elapsed / args.steps)
    # who took part: one entry per rank, and the size of the RCCL communicator the timing all-reduce went through
    devices = [f"rank {rank}: cuda:{local} {torch.cuda.get_device_name(dev)}"]
    if world > 1:
        devices = [None] * world
        dist.all_gather_object(devices, f"rank {rank}: cuda:{local} {torch.cuda.get_device_name(dev)}")
    rccl_ranks = dist.get_world_size() if dist.is_initialized() and dist.get_backend() == "nccl" else 0

    line = None
    if rank == 0:
        t = MICRO_BATCH * SEQ_LEN
        k = side.summary()
        timed = timer.summary()
        single_pass = "prl_policy_rows_fwd_bwd" in timed
        # algorithmic bytes per launch: V logits read for every row that predicts a token (all but the
        # one dropped by the shift) and V gradient elements written for every row; the single-pass kernel
        # adds the 12 bytes of logprob, entropy and lse per token, the two-pass forward reads the rows once more
        bwd_bytes = (t - 1) * VOCAB * es + t * VOCAB * es
        fwd_bytes = (t - 1) * VOCAB * es + 12 * t
        if single_pass:
            dom_name, dom_call, dom = "k_policy_rows", "prl_policy_rows_fwd_bwd", timed["prl_policy_rows_fwd_bwd"]
            dom_bytes = bwd_bytes + 12 * t
        else:
            dom_name, dom_call, dom = "k_logprob_bwd", "prl_logprob_bwd", timed["prl_logprob_bwd"]
            dom_bytes = bwd_bytes
        achieved = dom_bytes / (dom["mean_ms"] * 1e-3) / 1e9
        traffic, traffic_src = pmc_traffic(dom_name, args.logits_dtype)
        line = {
            "metric": "policy_update_samples_per_sec",
            "value": value,
            "unit": "samples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",  # the arithmetic; the logits and their gradient are stored as `storage_dtype`
            "storage_dtype": args.logits_dtype,
            "data": "synthetic",
            "rccl_ranks": rccl_ranks,
            "backend": dist.get_backend() if dist.is_initialized() else "none (one rank)",
            "devices": devices,
            "config": {
                "workload": "Qwen3-0.6B reverse_text GRPO policy update from resident logits: per step 8 packed rows "
                            f"of T={t} tokens x V={VOCAB} {args.logits_dtype} logits -> sample boundaries, logprob, "
                            "entropy, ratio-clip loss, dlogits" + (", logged statistics" if with_stats else ""),
                "global_batch": world * BATCH,
                "seq_len": SEQ_LEN,
                "micro_batch": MICRO_BATCH,
                "logits_storage": args.logits_dtype,
                "inputs": args.data,
                "parallelism": f"dp{world}",
            },
            "roofline": {
                "bound": "hbm",
                "kernel": dom_name,
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "traffic_source": traffic_src,
                "bytes_per_launch": dom_bytes,
                "mean_ms": dom["mean_ms"],
                "launches": dom["launches"],
                "timed_by": ("HIP events stamped by the kernel launch itself (hipExtLaunchKernelGGL), one launch in seven"
                             if single_pass and not args.event_pairs else
                             "HIP events recorded on the launch stream before and after the call, one launch in seven"),
            },
            "kernels": {
                dom_call: {"mean_ms": dom["mean_ms"], "GBps": achieved},
                # two-pass path only: one library call holding k_logprob_fwd and the bookkeeping kernel
                **({"prl_policy_forward": {"mean_ms": k["prl_policy_forward"]["mean_ms"],
                                           "GBps": fwd_bytes / (k["prl_policy_forward"]["mean_ms"] * 1e-3) / 1e9}}
                   if "prl_policy_forward" in k else {}),
                **({name: {"mean_ms": k[name]["mean_ms"]} for name in ("prl_policy_tail", "prl_scale_rows_if_not_one")
                    if name in k}),
                **({"prl_stats(all keys)": {"mean_ms": k["prl_stats"]["mean_ms"]}} if "prl_stats" in k else {}),
            },
        }
        if last_stats:
            line["last_step_stats"] = {k: last_stats[k] for k in ("loss/mean", "entropy/mean", "importance_ratio/mean")
                                       if k in last_stats}
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(rows, args.cpu_seconds)
        if world == 1 and not args.no_paths and dtype == torch.bfloat16:
            line["paths"] = other_paths(dev, rows, cfg, with_stats, args, ms_per_step, dom, achieved, dom_bytes)
            if args.data != "fake":
                del rows
                torch.cuda.empty_cache()
                fake = make_batch(dev, dtype, "fake", 1234 + rank)
                m = measure_path(fake, cfg, with_stats, 3, "prl_policy_rows_fwd_bwd", dom_bytes)
                line["value_fake_data"] = m["samples_per_s"]
                line["paths"]["token_bf16_fake_data"] = {
                    **m, "inputs": "the reference's FakeDataLoader distribution (src/prime_rl/trainer/rl/data.py:39-54): "
                                   "sampler log-probabilities ~ N(0, 1), so most ratios hit the clip and their rows "
                                   "get a zero gradient without being read"}
                rows = fake
    del rows
    torch.cuda.empty_cache()
    if rank == 0 and world == 1 and not args.no_paths:
        try:
            line["roofline_lm_head"] = lm_head_measure(dev)
        except Exception as exc:
            line["roofline_lm_head"] = {"error": f"{type(exc).__name__}: {exc}"}
        torch.cuda.empty_cache()

    # The whole-trainer-step extra runs on every rank (with more than one, the end of its step is a reduce-scatter
    # and an all-gather over RCCL).  The headline line must survive whatever happens in there: an exception is
    # reported inside the line, and with several ranks a watchdog prints the line without the extra and ends the
    # process if the extra, or the final barrier after a failure on another rank, has not come back in time.
    state = {"printed": False}
    lock = threading.Lock()

    def emit() -> None:
        with lock:
            if rank == 0 and not state["printed"]:
                state["printed"] = True
                print(json.dumps(line), file=LINE_OUT, flush=True)

    def bail() -> None:
        # a collective that never came back: the headline still goes out, the exit code says that the run hung
        if rank == 0:
            line.setdefault("trainer_step", {"error": f"no result within {args.trainer_limit:.0f} s"})
        emit()
        os._exit(EXIT_TRAINER_EXTRA_HUNG)

    watchdog = None
    if world > 1 and not args.no_trainer_extra:
        watchdog = threading.Timer(args.trainer_limit, bail)
        watchdog.daemon = True
        watchdog.start()
    if not args.no_trainer_extra:
        try:
            info = trainer_measure(dev, world, args.trainer_steps, 2, world == 1 and not args.no_cpu_baseline,
                                   args.cpu_seconds, args.trainer_impl)
        except Exception as exc:  # e.g. a transformers install that lacks the model
            info = {"error": f"{type(exc).__name__}: {exc}"}
        if rank == 0:
            line["trainer_step"] = info
    emit()
    if dist.is_initialized():
        dist.barrier()
        if watchdog is not None:
            watchdog.cancel()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
