import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import prime_rl_cse_amd as prl
from prime_rl_cse_amd import fused, trainer as T
from prime_rl_cse_amd.replica import Replica
dev = torch.device("cuda:0")
cfg = T.model_config("Qwen/Qwen3-0.6B")
model = T.get_model(cfg, dev, torch.float32, seed=0)
fused.fuse_model(model)
rep = Replica(model, lambda m: torch.optim.AdamW(m.parameters(), lr=1e-6, weight_decay=0.01, fused=True))
loss_cfg = prl.LossConfig()
for mbs in (16, 32, 64, 128):
    loader = prl.FakeDataLoader(prl.FakeDataLoaderConfig(batch_size=128, micro_batch_size=mbs, seq_len=128))
    batches = [loader.get_batch() for _ in range(2)]
    for i in range(3):
        rep.train_step(batches[i % 2], loss_cfg, samples=128)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(3):
        res = rep.train_step(batches[i % 2], loss_cfg, samples=128)
    torch.cuda.synchronize()
    print(f"micro_batch_size {mbs:4d} (T = {mbs * 128:6d}, {len(batches[0])} per step): {(time.perf_counter() - t0) / 3 * 1e3:7.1f} ms per step, "
          f"peak memory {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB, errors {list(rep.capture_errors.values())[:1]}", flush=True)
