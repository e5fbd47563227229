"""Where a weight publication's time goes: the caller's part of `publish` and the writer thread's parts, for a
Qwen3-0.6B-sized bf16 model published every 60 ms (the step time of the trainer)."""
import sys, tempfile, time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch

import prime_rl_cse_amd as prl
from prime_rl_cse_amd import trainer as T
from prime_rl_cse_amd.weights import WeightPublisher

prl.reset_world()
dev = torch.device("cuda:0")
model = T.get_model(T.model_config("Qwen/Qwen3-0.6B"), dev, torch.bfloat16, seed=0)
in_flight = int(sys.argv[1]) if len(sys.argv) > 1 else 2
if len(sys.argv) > 2 and sys.argv[2] == "nocrc":
    torch.serialization.set_crc32_options(False)
pub = WeightPublisher(Path(tempfile.mkdtemp()), async_level=2, device=dev, max_in_flight=in_flight)
real_save = torch.save
spans = []


def timed_save(obj, f, *a, **k):
    t0 = time.perf_counter()
    real_save(obj, f, *a, **k)
    spans.append(time.perf_counter() - t0)


torch.save = timed_save
t_all = time.perf_counter()
for step in range(1, 13):
    t0 = time.perf_counter()
    sd = model.state_dict()
    t1 = time.perf_counter()
    pending = pub.publish(model, step, state_dict=sd)
    t2 = time.perf_counter()
    pub.guard_parameters()
    pub.maybe_clean(step)
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    print(f"step {step}: state_dict {1e3 * (t1 - t0):.1f} ms, publish {1e3 * (t2 - t1):.1f} ms, "
          f"until the device is idle {1e3 * (t3 - t2):.1f} ms", flush=True)
    time.sleep(0.06)
print(f"in flight {in_flight}: {1e3 * (time.perf_counter() - t_all) / 12:.1f} ms per step of 60 ms", flush=True)
pub.wait_all()
print("torch.save spans (ms):", [round(1e3 * s) for s in spans])

import cProfile, pstats

prof = cProfile.Profile()
sd = model.state_dict()
prof.enable()
for step in range(20, 24):
    pub.publish(model, step, state_dict=sd)
    pub.guard_parameters()
prof.disable()
pub.wait_all()
pstats.Stats(prof).sort_stats("cumulative").print_stats(18)
