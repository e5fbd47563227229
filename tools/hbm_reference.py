"""Reference HBM rates on the box, measured the same way as the kernels (HIP events,
interleaved rounds): a device-to-device copy (read + write), a fill (write only) and a
sum (read only) over a buffer the size of one micro-batch of logits."""

import statistics
import sys

import torch

n_bytes = int(sys.argv[1]) if len(sys.argv) > 1 else 2048 * 151936 * 2
dev = torch.device("cuda:0")
src = torch.empty(n_bytes // 4, dtype=torch.float32, device=dev).normal_()
dst = torch.empty_like(src)
other = torch.empty_like(src)
res = {"copy": [], "fill": [], "sum": [], "copy_after_copy": []}
for _ in range(30):
    for name in res:
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        if name == "copy_after_copy":
            other.copy_(dst)
        a.record()
        if name in ("copy", "copy_after_copy"):
            dst.copy_(src)
        elif name == "fill":
            dst.zero_()
        else:
            src.sum()
        b.record()
        b.synchronize()
        res[name].append(a.elapsed_time(b))
for name, ms in res.items():
    m = statistics.median(ms)
    moved = n_bytes * (1 if name in ("fill", "sum") else 2)
    print(f"{name:16s} {m:.4f} ms  {moved / m / 1e6:7.0f} GB/s (bytes moved {moved / 1e6:.0f} MB)")
