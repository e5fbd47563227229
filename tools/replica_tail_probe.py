"""How far ahead of the GPU is the host through one training step of `Replica` (Qwen3-0.6B, 128 samples of 128
tokens joined into one row)?  An event is recorded at marked places of the step; whether it has already
completed when the host gets there says who waits for whom, and the host clock says what each stretch costs.

    python tools/replica_tail_probe.py
"""
import sys, time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch

import prime_rl_cse_amd as prl
from prime_rl_cse_amd import fused, replica as R, trainer as T
from prime_rl_cse_amd.optim import FlatAdamW

dev = torch.device("cuda:0")
cfg = T.model_config("Qwen/Qwen3-0.6B")
model = T.get_model(cfg, dev, torch.float32, seed=0)
fused.fuse_model(model)
rep = R.Replica(model, lambda m: FlatAdamW(m.parameters(), lr=1e-6, weight_decay=0.01), graphs=False)
loader = prl.FakeDataLoader(prl.FakeDataLoaderConfig(batch_size=128, micro_batch_size=16, seq_len=128))
loss_cfg = prl.LossConfig()
marks = []


def mark(name):
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    marks.append((name, time.perf_counter(), e, e.query()))


orig_fb = rep._forward_backward


def fb(*a, **k):
    mark("pass begins")
    out = orig_fb(*a, **k)
    mark("backward returned")
    return out


rep._forward_backward = fb
orig_acc = rep._accumulate


def acc(*a, **k):
    mark("accumulate launch")
    return orig_acc(*a, **k)


rep._accumulate = acc
orig_finish = rep.state.finish_step


def finish(*a, **k):
    mark("finish_step begins")
    out = orig_finish(*a, **k)
    mark("finish_step launched")
    return out


rep.state.finish_step = finish
for step in range(4):
    mbs = loader.get_batch()
    marks.clear()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    mark("step begins")
    res = rep.train_step(mbs, loss_cfg)
    mark("step returned")
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    if step < 2:
        continue
    print(f"step {step}: {1e3 * (t1 - t0):.2f} ms")
    first = marks[0][2]
    for name, t, e, done_then in marks:
        print(f"  {1e3 * (t - t0):8.3f} ms host  {first.elapsed_time(e):8.3f} ms GPU  {name:24s} "
              f"GPU had {'caught up' if done_then else 'work queued'}", flush=True)
