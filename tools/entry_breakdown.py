"""Mean of the entry point's time/* columns over a run of a recipe (after the first five steps)."""
import os, sys, tempfile
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
for k, v in dict(MASTER_ADDR="127.0.0.1", MASTER_PORT="29796", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
                 LOCAL_WORLD_SIZE="1").items():
    os.environ.setdefault(k, v)
from prime_rl_cse_amd.settings import RLTrainerConfig, parse_argv
from prime_rl_cse_amd.train import train

cfg = parse_argv(RLTrainerConfig, ["@", sys.argv[1], "--output-dir", tempfile.mkdtemp(), *sys.argv[2:]], use_env=False)
hist = train(cfg)[5:]
for key in sorted(k for k in hist[0] if k.startswith("time/")):
    vals = [h[key] for h in hist]
    print(f"{key:28s} mean {1e3 * sum(vals) / len(vals):7.1f} ms   max {1e3 * max(vals):7.1f} ms")
