"""Build-time tunables of the two vocabulary-wide kernels, measured against each
other in ONE process with interleaved rounds.

    python tools/tune_logprob.py build        # in the build container: compiles every variant
    python tools/tune_logprob.py run [--dtype bf16|f32] [--rounds 20]   # on the MI355X box

Variants are whole copies of libprl_hip.so that differ only in the -D flags
logprob.hip was compiled with; each is loaded with ctypes from its own path.
"""

from __future__ import annotations

import argparse
import ctypes
import itertools
import statistics
import subprocess
import sys
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
CSRC = REPO / "prime-rl-cse_amd" / "csrc"
LIB = REPO / "prime-rl-cse_amd" / "lib"
TUNE = LIB / "tune"

VARIANTS: dict[str, dict[str, int]] = {"base": {}}
for pipe, th, ld in itertools.product((0, 1), (256, 512, 1024), (2, 4, 8)):
    VARIANTS[f"p{pipe}_t{th}_l{ld}"] = {"PRL_BWD_PIPE": pipe, "PRL_FWD_THREADS": th, "PRL_BWD_THREADS": th,
                                        "PRL_FWD_UNR": ld, "PRL_BWD_LOADS": ld}
VARIANTS["nts1"] = {"PRL_NT_STORE": 1}
VARIANTS["ntl0"] = {"PRL_NT_LOAD": 0}


def build() -> None:
    subprocess.run(["make", "-C", str(CSRC), "-j8"], check=True)
    TUNE.mkdir(parents=True, exist_ok=True)
    others = [str(LIB / "obj" / f"{n}.o") for n in ("capi", "segments", "loss", "advantage", "stats", "policy")]
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Xclang", "-target-feature", "-Xclang",
             "-packed-fp32-ops", f"-I{REPO / 'include'}", f"-I{CSRC}"]
    for tag, defs in VARIANTS.items():
        obj = TUNE / f"logprob_{tag}.o"
        dflags = [f"-D{k}={v}" for k, v in defs.items()]
        subprocess.run(["/opt/rocm/bin/hipcc", *flags, *dflags, "-c", str(CSRC / "logprob.hip"), "-o", str(obj)],
                       check=True, stderr=subprocess.DEVNULL)
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o",
                        str(TUNE / f"libprl_{tag}.so"), str(obj), *others], check=True)
        obj.unlink()
        print("built", tag, flush=True)


def run(dtype_name: str, rounds: int, t: int, v: int) -> None:
    import torch

    sys.path.insert(0, str(REPO))
    from prime_rl_cse_amd import _capi as C

    dev = torch.device("cuda:0")
    dtype = torch.bfloat16 if dtype_name == "bf16" else torch.float32
    code = C.PRL_BF16 if dtype_name == "bf16" else C.PRL_F32
    es = 2 if dtype_name == "bf16" else 4
    g = torch.Generator(device=dev).manual_seed(0)
    logits = torch.empty(t, v, dtype=dtype, device=dev)
    for r in range(0, t, 512):
        logits[r : r + 512] = (torch.randn(min(512, t - r), v, generator=g, device=dev) * 2.5).to(dtype)
    ids = torch.randint(0, v, (t,), generator=g, device=dev)
    lp = torch.empty(t, device=dev)
    ent = torch.empty(t, device=dev)
    lse = torch.empty(t, device=dev)
    gvec = torch.randn(t, generator=g, device=dev) * 1e-3
    out = torch.empty(t, v, dtype=dtype, device=dev)
    st = C.stream_ptr(dev)

    libs = {}
    for tag in VARIANTS:
        path = TUNE / f"libprl_{tag}.so"
        if not path.exists():
            continue
        lib = ctypes.CDLL(str(path))
        for name in ("prl_logprob_entropy_fwd", "prl_logprob_bwd"):
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = C.SIGNATURES[name]
        libs[tag] = lib

    def fwd(lib):
        return lib.prl_logprob_entropy_fwd(C.ptr(logits), code, t, v, v, C.ptr(ids), 1.0, 1, C.ptr(lp), C.ptr(ent),
                                           C.ptr(lse), None, 0, st)

    def bwd(lib):
        return lib.prl_logprob_bwd(C.ptr(logits), code, t, v, v, C.ptr(ids), C.ptr(lse), C.ptr(gvec), None, 1.0, 1,
                                   C.ptr(out), code, v, st)

    times = {tag: {"fwd": [], "bwd": []} for tag in libs}
    for tag, lib in libs.items():  # warm up, and check every variant agrees with the first
        assert fwd(lib) == 0 and bwd(lib) == 0
    torch.cuda.synchronize()
    for _ in range(rounds):
        for tag, lib in libs.items():
            for name, fn in (("fwd", fwd), ("bwd", bwd)):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                fn(lib)
                b.record()
                b.synchronize()
                times[tag][name].append(a.elapsed_time(b))
    fwd_bytes = (t - 1) * v * es
    bwd_bytes = (2 * t - 1) * v * es
    print(f"T={t} V={v} {dtype_name}: median (min) ms and GB/s at the median")
    for tag in libs:
        f, b = times[tag]["fwd"], times[tag]["bwd"]
        fm, bm = statistics.median(f), statistics.median(b)
        print(f"{tag:16s} fwd {fm:.4f} ({min(f):.4f}) {fwd_bytes / fm / 1e6:7.0f} GB/s | "
              f"bwd {bm:.4f} ({min(b):.4f}) {bwd_bytes / bm / 1e6:7.0f} GB/s", flush=True)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("mode", choices=["build", "run"])
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--rounds", type=int, default=20)
    ap.add_argument("--t", type=int, default=2048)
    ap.add_argument("--v", type=int, default=151936)
    a = ap.parse_args()
    build() if a.mode == "build" else run(a.dtype, a.rounds, a.t, a.v)
