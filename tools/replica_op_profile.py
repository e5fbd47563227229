"""Where do the small element-wise kernels of one micro-batch come from?  torch.profiler over one eager
micro-batch of `Replica` (Qwen3-0.6B, fused block kernels), grouped by operator and Python call site.

    python tools/replica_op_profile.py [layers]
"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from torch.profiler import ProfilerActivity, profile

import prime_rl_cse_amd as prl
from prime_rl_cse_amd import fused, trainer as T
from prime_rl_cse_amd.replica import Replica

dev = torch.device("cuda:0")
layers = int(sys.argv[1]) if len(sys.argv) > 1 else 4
cfg = T.model_config("Qwen/Qwen3-0.6B", num_layers=layers)
model = T.get_model(cfg, dev, torch.float32, seed=0)
fused.fuse_model(model)
rep = Replica(model, lambda m: torch.optim.AdamW(m.parameters(), lr=1e-6, fused=True), graphs=False)
loader = prl.FakeDataLoader(prl.FakeDataLoaderConfig(batch_size=32, micro_batch_size=16, seq_len=128))
mbs = loader.get_batch()
loss_cfg = prl.LossConfig()
rep.train_step(mbs, loss_cfg)
torch.cuda.synchronize()
host = mbs[0]
mb = prl.to_device(host, dev)
plan = T.attention_plan(host, dev)
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    rep.micro_batch(mb, plan, loss_cfg)
    torch.cuda.synchronize()
print(prof.key_averages(group_by_input_shape=True).table(sort_by="cuda_time_total", row_limit=45, max_name_column_width=60,
                                                         max_shapes_column_width=70))
print(prof.key_averages(group_by_stack_n=6).table(sort_by="cuda_time_total", row_limit=40, max_name_column_width=40,
                                                  max_src_column_width=110))
