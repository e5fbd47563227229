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

VARIANTS: dict[str, dict[str, int]] = {"base": {}, "base_again": {}}  # two copies of one code: the noise floor
VARIANTS["rows"] = {"PRL_BWD_FLAT": 0}
for upt, grid in itertools.product((1, 2, 4), (1024, 2048, 4096, 8192)):
    VARIANTS[f"flat_u{upt}_g{grid}"] = {"PRL_BWD_FLAT": 1, "PRL_BWD_FLAT_UPT": upt, "PRL_BWD_FLAT_GRID": grid}
VARIANTS["flat_u1_g2048_t512"] = {"PRL_BWD_FLAT_UPT": 1, "PRL_BWD_FLAT_GRID": 2048, "PRL_BWD_THREADS": 512}
VARIANTS["flat_u1_g1024_t512"] = {"PRL_BWD_FLAT_UPT": 1, "PRL_BWD_FLAT_GRID": 1024, "PRL_BWD_THREADS": 512}
VARIANTS["flat_u1_g2048_nts1"] = {"PRL_BWD_FLAT_UPT": 1, "PRL_BWD_FLAT_GRID": 2048, "PRL_NT_STORE": 1}
VARIANTS["flat_u1_g2048_bnt0"] = {"PRL_BWD_FLAT_UPT": 1, "PRL_BWD_FLAT_GRID": 2048, "PRL_BWD_NT_LOAD": 0}


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


def run(dtype_name: str, rounds: int, t: int, v: int, producer: bool = True) -> None:
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
    src = logits.clone()
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
    import random

    rng = random.Random(0)
    order = list(libs.items())
    for _ in range(rounds):
        rng.shuffle(order)  # a fresh order every round, so that no variant always follows the same one
        for tag, lib in order:
            # the producer of the logits (the LM-head GEMM in the trainer) writes them just before the
            # forward reads them; a device copy stands in for it so that the caches start from that state
            if producer:
                logits.copy_(src)
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
            ev[0].record()
            fwd(lib)
            ev[1].record()
            bwd(lib)
            ev[2].record()
            ev[2].synchronize()
            times[tag]["fwd"].append(ev[0].elapsed_time(ev[1]))
            times[tag]["bwd"].append(ev[1].elapsed_time(ev[2]))
    fwd_bytes = (t - 1) * v * es
    bwd_bytes = (2 * t - 1) * v * es
    print(f"T={t} V={v} {dtype_name}: median (min) ms and GB/s at the median")
    for tag in libs:
        f, b = times[tag]["fwd"], times[tag]["bwd"]
        fm, bm = statistics.median(f), statistics.median(b)
        pair = statistics.median([x + y for x, y in zip(f, b)])
        print(f"{tag:22s} fwd {fm:.4f} ({min(f):.4f}) {fwd_bytes / fm / 1e6:6.0f} GB/s | "
              f"bwd {bm:.4f} ({min(b):.4f}) {bwd_bytes / bm / 1e6:6.0f} GB/s | pair {pair:.4f}", flush=True)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("mode", choices=["build", "run"])
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--rounds", type=int, default=20)
    ap.add_argument("--t", type=int, default=2048)
    ap.add_argument("--v", type=int, default=151936)
    ap.add_argument("--no-producer", action="store_true", help="do not rewrite the logits before each forward")
    a = ap.parse_args()
    build() if a.mode == "build" else run(a.dtype, a.rounds, a.t, a.v, not a.no_producer)
