"""Host-side profile of one whole trainer step (random-init Qwen3-0.6B, FSDP2 over a one-rank RCCL group)."""
import cProfile, os, pstats, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import prime_rl_cse_amd as prl
from prime_rl_cse_amd import trainer as T

os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29588")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
dev = torch.device("cuda:0"); torch.cuda.set_device(0)
mesh = T.init_distributed(dev)
layers = int(sys.argv[1]) if len(sys.argv) > 1 else 28
blocks_per_group = int(sys.argv[2]) if len(sys.argv) > 2 else 1
cfg = T.model_config("Qwen/Qwen3-0.6B", num_layers=layers)
model = T.get_model(cfg, dev, torch.float32, seed=0)
if len(sys.argv) > 3 and sys.argv[3] == "fused":
    from prime_rl_cse_amd import fused
    print("fused:", fused.fuse_model(model))
try:
    T.setup_fsdp(model, mesh, blocks_per_group=blocks_per_group)
except TypeError:
    T.setup_fsdp(model, mesh)
opt = T.setup_optimizer(model)
prl.reset_world()
loader = prl.FakeDataLoader(prl.FakeDataLoaderConfig(batch_size=128, micro_batch_size=16, seq_len=128))
batch = loader.get_batch()
for _ in range(2):
    T.train_step(model, opt, batch, prl.LossConfig(), device=dev, samples=128)
torch.cuda.synchronize()
t0 = time.perf_counter()
T.train_step(model, opt, batch, prl.LossConfig(), device=dev, samples=128)
torch.cuda.synchronize()
print(f"step {1e3 * (time.perf_counter() - t0):.1f} ms ({layers} layers, {blocks_per_group} blocks per FSDP group)")
pr = cProfile.Profile(); pr.enable()
T.train_step(model, opt, batch, prl.LossConfig(), device=dev, samples=128)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(int(os.environ.get("PROFILE_LINES", "28")))
import torch.distributed as dist
dist.destroy_process_group()
