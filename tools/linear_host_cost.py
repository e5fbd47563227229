"""Host and device time of one bf16 nn.Linear forward + backward at the trainer's shapes, per BLAS library."""
import sys, time
import torch

dev = "cuda:0"
for lib in ("hipblaslt", "hipblas"):
    try:
        torch.backends.cuda.preferred_blas_library(lib)
    except Exception as exc:
        print(lib, "not selectable:", exc); continue
    for (t, k, n) in ((2048, 1024, 3072), (2048, 3072, 1024), (2048, 1024, 2048)):
        x = torch.randn(1, t, k, device=dev, dtype=torch.bfloat16, requires_grad=True)
        lin = torch.nn.Linear(k, n, bias=False, device=dev, dtype=torch.bfloat16)
        g = torch.randn(1, t, n, device=dev, dtype=torch.bfloat16)
        for _ in range(20):
            lin(x).backward(g)
        torch.cuda.synchronize()
        it = 300
        t0 = time.perf_counter()
        for _ in range(it):
            y = lin(x)
        t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        for _ in range(it):
            lin(x).backward(g)
        t3 = time.perf_counter(); torch.cuda.synchronize(); t4 = time.perf_counter()
        print(f"{lib:10s} [{t}x{k}]x[{k}x{n}] fwd host {1e6*(t1-t0)/it:6.1f} us done {1e6*(t2-t0)/it:6.1f} us | fwd+bwd host {1e6*(t3-t2)/it:6.1f} us done {1e6*(t4-t2)/it:6.1f} us")
