"""Two layers of every known model shape through the trainer entry point, whole-model mode and FSDP2: the logged
numbers must agree between the modes."""
import os, sys, tempfile
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
for k, v in dict(MASTER_ADDR="127.0.0.1", MASTER_PORT="29791", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", LOCAL_WORLD_SIZE="1").items():
    os.environ[k] = v
import torch
from prime_rl_cse_amd.settings import RLTrainerConfig
from prime_rl_cse_amd.train import train
for name in ("Qwen/Qwen2.5-7B", "Qwen/Qwen2.5-0.5B", "Qwen/Qwen3-1.7B"):  # two layers of each: shapes, not speed
    for mode in ("auto", "fsdp"):
        os.environ["PRL_TRAINER"] = mode
        out = tempfile.mkdtemp()
        cfg = RLTrainerConfig(max_steps=5, output_dir=out, model={"name": name, "debug": {"num_layers": 2}},
                              data={"fake": {"batch_size": 16, "micro_batch_size": 4, "seq_len": 128}})
        hist = train(cfg)
        print(name, mode, [round(h["optim/grad_norm"], 6) for h in hist], [round(h["entropy/mean"], 4) for h in hist],
              "step", [round(h["time/step"], 3) for h in hist], "fwd/bwd", [round(h["time/forward_backward"], 3) for h in hist],
              "save", [round(h["time/save_weights"], 3) for h in hist], flush=True)
        torch.cuda.empty_cache()
