"""Which ways of writing a number into device tensor elements make the host wait for the GPU?  Behind ~50 ms of
queued matrix products, the host time of each statement (a statement that only enqueues work takes microseconds).

    python3 tools/setitem_sync_probe.py
"""
import time

import torch

dev = torch.device("cuda:0")
a = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
x = torch.zeros(16384, device=dev)
x2 = torch.zeros(1, 16384, device=dev)
idx = torch.arange(0, 16384, 512, device=dev)
val = torch.full((idx.numel(),), 2.5, device=dev)


def busy() -> None:
    for _ in range(40):
        a @ a


cases = {
    "x[idx] = 1.5": lambda: x.__setitem__(idx, 1.5),
    "x2[:, idx] = 1.5": lambda: x2.__setitem__((slice(None), idx), 1.5),
    "x.index_fill_(0, idx, 1.5)": lambda: x.index_fill_(0, idx, 1.5),
    "x[idx] = device tensor": lambda: x.__setitem__(idx, val),
    "x[3] = 1.5": lambda: x.__setitem__(3, 1.5),
    "x[idx] (read)": lambda: x[idx],
}
for _ in range(2):
    for name, fn in cases.items():
        torch.cuda.synchronize()
        busy()
        t0 = time.perf_counter()
        fn()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"{name:28s} host {1e3 * (t1 - t0):8.3f} ms, queue drained after another {1e3 * (t2 - t1):7.3f} ms", flush=True)
