"""Host time to enqueue one RMSNorm forward + backward: HIP kernel behind ctypes vs torch's eager chain."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from prime_rl_cse_amd import fused

dev = "cuda:0"
for shape in [(1, 2048, 1024), (1, 2048, 16, 128)]:
    x = torch.randn(shape, device=dev).to(torch.bfloat16).requires_grad_(True)
    w = torch.ones(shape[-1], device=dev, dtype=torch.bfloat16, requires_grad=True)
    gy = torch.randn(shape, device=dev).to(torch.bfloat16)

    def eager():
        h = x.to(torch.float32)
        v = h.pow(2).mean(-1, keepdim=True)
        return w * (h * torch.rsqrt(v + 1e-6)).to(torch.bfloat16)

    for name, fn in (("eager", eager), ("hip", lambda: fused.rms_norm(x, w, 1e-6))):
        for _ in range(20):
            fn().backward(gy)
        torch.cuda.synchronize()
        n = 300
        t0 = time.perf_counter()
        for _ in range(n):
            y = fn()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        for _ in range(n):
            fn().backward(gy)
        t3 = time.perf_counter()
        torch.cuda.synchronize()
        t4 = time.perf_counter()
        print(f"{shape} {name:6s} fwd: host {1e6 * (t1 - t0) / n:6.1f} us, done {1e6 * (t2 - t0) / n:6.1f} us | "
              f"fwd+bwd: host {1e6 * (t3 - t2) / n:6.1f} us, done {1e6 * (t4 - t2) / n:6.1f} us")
