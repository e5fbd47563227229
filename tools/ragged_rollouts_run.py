"""Ragged rollouts, as an orchestrator would write them, through the trainer entry point at full model size:
random prompt / completion lengths, prompt tokens masked, group advantages, first-fit-decreasing packing into rows
(`prepare_batch`), `rollouts/step_N/rank_0.pt` files, then `train()` reading them.

usage: python tools/ragged_rollouts_run.py [--model NAME] [--seq-len 128] [--micro-batch 16] [--batch 128]
                                            [--steps 4] [--max-completion N] [--settled SHARE]"""
import argparse, os, sys, tempfile
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
for k, v in dict(MASTER_ADDR="127.0.0.1", MASTER_PORT="29795", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
                 LOCAL_WORLD_SIZE="1").items():
    os.environ.setdefault(k, v)
import numpy as np
import torch

import prime_rl_cse_amd as prl
from prime_rl_cse_amd.batch import Rollout, prepare_batch
from prime_rl_cse_amd.data import save_rollout_batches
from prime_rl_cse_amd.settings import RLTrainerConfig
from prime_rl_cse_amd.train import train

p = argparse.ArgumentParser()
p.add_argument("--model", default="PrimeIntellect/Qwen3-0.6B-Reverse-Text-SFT")
p.add_argument("--seq-len", type=int, default=128)
p.add_argument("--micro-batch", type=int, default=16)
p.add_argument("--batch", type=int, default=128)
p.add_argument("--group", type=int, default=16)
p.add_argument("--steps", type=int, default=4)
p.add_argument("--max-completion", type=int, default=None)
p.add_argument("--temperature", type=float, default=1.0)
p.add_argument("--settled", type=float, default=None,
               help="rewards are 0 or 1, and this share of the problems is always solved or never solved (every "
                    "rollout of such a group has the same reward, so its group-relative advantage is 0)")
args = p.parse_args()

rng = np.random.default_rng(0)
out = Path(tempfile.mkdtemp())
for step in range(args.steps):
    if args.settled is None:
        rewards = rng.random(args.batch).round(1).tolist()
    else:
        n_prob = args.batch // args.group
        p_solve = np.where(rng.random(n_prob) < args.settled, rng.integers(0, 2, n_prob).astype(float), 0.5)
        rewards = (rng.random(args.batch) < np.repeat(p_solve, args.group)).astype(float).tolist()
    rollouts, lengths = [], []
    for i in range(args.batch):
        n_prompt = int(rng.integers(10, max(11, args.seq_len // 3)))
        room = args.seq_len - n_prompt
        n_comp = int(rng.integers(1, min(room, args.max_completion or room) + 1))
        lengths.append(n_comp)
        rollouts.append(Rollout(problem_id=i // args.group, prompt_tokens=rng.integers(0, 150000, n_prompt).tolist(),
                                prompt_mask=[0] * n_prompt, completion_tokens=rng.integers(0, 150000, n_comp).tolist(),
                                completion_mask=[1] * n_comp,
                                completion_logprobs=(-0.1 - np.abs(rng.normal(0, 1.5, n_comp))).tolist(),
                                is_truncated=False, reward=rewards[i], advantage=0.0))
    adv = prl.compute_advantages(rewards, lengths, args.group, "drgrpo")
    for r, a in zip(rollouts, adv):
        r.advantage = a
    per_rank = prepare_batch(rollouts, args.temperature, None, args.batch, args.micro_batch, args.seq_len, 1)
    save_rollout_batches(per_rank, out, step)
    rows = per_rank[0]
    print(f"step {step}: {len(rows)} rows, tokens {[int(r['input_ids'].shape[1]) for r in rows]}, "
          f"samples per row {[int((r['position_ids'] == 0).sum()) for r in rows]}", flush=True)

cfg = RLTrainerConfig(max_steps=args.steps, output_dir=str(out), model={"name": args.model})
hist = train(cfg)
for h in hist:
    print({k: (round(v, 5) if isinstance(v, float) else v) for k, v in h.items()
           if k in ("step", "time/step", "time/forward_backward", "time/load_data", "time/save_weights", "loss/mean",
                    "entropy/mean", "importance_ratio/mean", "is_clipped/mean", "optim/grad_norm",
                    "perf/throughput", "perf/mfu", "perf/peak_memory")}, flush=True)
