"""How much of a bench step is host time?  Enqueue K steps without synchronising, note when
the host is done, then synchronise.  If 'host done' is close to 'all done' the step is
host-bound."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import bench
import prime_rl_cse_amd as prl

dev = torch.device("cuda:0")
torch.cuda.set_device(0)
rows = bench.make_batch(dev, torch.bfloat16, "near", 1234)
cfg = prl.LossConfig()
for with_stats in (True, False):
    for _ in range(3):
        bench.run_step(rows, cfg, with_stats)
    torch.cuda.synchronize()
    k = 20
    t0 = time.perf_counter()
    for _ in range(k):
        bench.run_step(rows, cfg, with_stats)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"stats={with_stats}: host enqueue {1e3 * (t1 - t0) / k:.3f} ms/step, all done {1e3 * (t2 - t0) / k:.3f} ms/step")
# cost of pieces of the host path, per micro-batch
r = rows[0]
import cProfile, pstats
pr = cProfile.Profile()
pr.enable()
for _ in range(10):
    bench.run_step(rows, cfg, True)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(25)
