"""The forward LM-head product on several builds of the library side by side (each loaded with ctypes), timed in
random order in one process:   python tools/lm_head_variants.py name=path.so [name=path.so ...] [--t --h --v]"""
import argparse
import ctypes
import random
import statistics

import torch

ap = argparse.ArgumentParser()
ap.add_argument("libs", nargs="+")
ap.add_argument("--t", type=int, default=16384)
ap.add_argument("--h", type=int, default=1024)
ap.add_argument("--v", type=int, default=151936)
ap.add_argument("--rounds", type=int, default=9)
args = ap.parse_args()
dev = torch.device("cuda:0")
gen = torch.Generator(device=dev).manual_seed(0)
x = torch.randn((args.t, args.h), generator=gen, device=dev).to(torch.bfloat16)
w = (torch.randn((args.v, args.h), generator=gen, device=dev) * 0.05).to(torch.bfloat16)
out = torch.empty((args.t, args.v), dtype=torch.bfloat16, device=dev)
libs = {}
for spec in args.libs:
    name, path = spec.split("=", 1)
    lib = ctypes.CDLL(path)
    lib.prl_lm_head_fwd.restype = ctypes.c_int
    libs[name] = lib
stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def run(lib):
    rc = lib.prl_lm_head_fwd(ctypes.c_void_p(x.data_ptr()), ctypes.c_int64(args.h), ctypes.c_void_p(w.data_ptr()),
                             ctypes.c_int64(args.h), ctypes.c_void_p(out.data_ptr()), ctypes.c_int64(args.v),
                             ctypes.c_int64(args.t), ctypes.c_int64(args.h), ctypes.c_int64(args.v), stream)
    assert rc == 0


ref = None
for name, lib in libs.items():
    run(lib)
    torch.cuda.synchronize()
    if ref is None:
        ref = out.clone()
    else:
        assert torch.equal(out, ref), f"{name} differs from the first build"
times = {n: [] for n in libs}
order = list(libs)
for _ in range(args.rounds):
    random.shuffle(order)
    for n in order:
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(3):
            run(libs[n])
        b.record()
        torch.cuda.synchronize()
        times[n].append(a.elapsed_time(b) / 3)
for n, ts in times.items():
    print(f"{n:12s} median {statistics.median(ts):.3f} ms  min {min(ts):.3f} ms", flush=True)
