"""Variants of the single-pass row kernel measured against each other in ONE process, rounds in random order.

    python tools/tune_rows.py build      # build container: compiles every variant
    python tools/tune_rows.py run        # MI355X box
"""
from __future__ import annotations

import ctypes
import random
import statistics
import subprocess
import sys
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
CSRC = REPO / "prime-rl-cse_amd" / "csrc"
LIB = REPO / "prime-rl-cse_amd" / "lib"
TUNE = LIB / "tune"

VARIANTS = {
    "base": {}, "base_again": {},
    "plain_store": {"PRL_ROWS_NT_STORE": 0},
    "plain_load": {"PRL_ROWS_NT_LOAD": 0},
    "rows_in_launch_order": {"PRL_ROWS_XCD_BLOCKS": 0},
    "walking_workgroups": {"PRL_ROWS_WALK": 1},
    "walking_reads_in_the_write_stream": {"PRL_ROWS_WALK": 1, "PRL_ROWS_INTERLEAVE": 1},
    "trace": {"PRL_ROWS_TRACE": 1},
    # cache-policy bits of the gradient stores: nt + sc1 (written through the L2 at device scope), sc1 alone,
    # sc0 + sc1 (system scope), nt + sc0 + sc1
    "store_nt_sc1": {"PRL_ROWS_STORE_AUX": 18},
    "store_sc1": {"PRL_ROWS_STORE_AUX": 16},
    "store_sc0_sc1": {"PRL_ROWS_STORE_AUX": 17},
    "store_nt_sc0_sc1": {"PRL_ROWS_STORE_AUX": 19},
}


def build() -> None:
    subprocess.run(["make", "-C", str(CSRC), "-j8"], check=True)
    TUNE.mkdir(parents=True, exist_ok=True)
    others = [str(LIB / "obj" / f"{n}.o") for n in ("capi", "logprob", "segments", "loss", "advantage", "stats", "policy",
                                                      "norm", "accum", "attn_short", "attn_long", "optim", "lm_head")]
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Xclang", "-target-feature", "-Xclang",
             "-packed-fp32-ops", f"-I{REPO / 'include'}", f"-I{CSRC}"]
    for tag, defs in VARIANTS.items():
        obj = TUNE / f"rows_{tag}.o"
        dflags = [f"-D{k}={v}" for k, v in defs.items()]
        subprocess.run(["/opt/rocm/bin/hipcc", *flags, *dflags, "-c", str(CSRC / "policy_rows.hip"), "-o", str(obj)],
                       check=True, stderr=subprocess.DEVNULL)
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o",
                        str(TUNE / f"librows_{tag}.so"), str(obj), *others], check=True)
        obj.unlink()
        print("built", tag, flush=True)


def run(rounds: int = 15, t: int = 2048, v: int = 151936) -> None:
    import torch

    sys.path.insert(0, str(REPO))
    from prime_rl_cse_amd import _capi as C

    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(0)
    logits = torch.empty(t, v, dtype=torch.bfloat16, device=dev)
    for r in range(0, t, 512):
        logits[r : r + 512] = (torch.randn(min(512, t - r), v, generator=g, device=dev) * 2.5).to(torch.bfloat16)
    # the calls of a round walk over several copies of the logits and several gradient buffers, as the benchmark's
    # eight micro-batches do: with one buffer a good part of it stays in the 256 MB Infinity Cache between calls
    sets = 5
    logits_sets = [logits] + [logits.clone() for _ in range(sets - 1)]
    ids = torch.randint(0, v, (t,), generator=g, device=dev)
    adv = torch.where(torch.rand(t, generator=g, device=dev) < 0.5, -0.5, 0.5)
    old = torch.full((t,), -12.0, device=dev)
    mask = torch.ones(t, dtype=torch.uint8, device=dev)
    lp, ent, lse = (torch.empty(t, device=dev) for _ in range(3))
    out_sets = [torch.empty(t, v, dtype=torch.bfloat16, device=dev) for _ in range(sets)]
    out = out_sets[0]
    turn = [0]
    st = C.stream_ptr(dev)
    libs = {}
    for path in sorted(TUNE.glob("librows_*.so")):  # every build in the directory, also hand-placed ones
        lib = ctypes.CDLL(str(path))
        fn = lib.prl_policy_rows_fwd_bwd
        fn.restype, fn.argtypes = C.SIGNATURES["prl_policy_rows_fwd_bwd"]
        libs[path.stem[len("librows_"):]] = fn

    def call(fn):
        k = turn[0] = (turn[0] + 1) % sets
        rc = fn(C.ptr(logits_sets[k]), C.PRL_BF16, t, v, v, C.ptr(ids), C.ptr(adv), C.ptr(old), C.ptr(mask), 0, 1.0,
                1e30, float(t), C.ptr(lp), C.ptr(ent), C.ptr(lse), C.ptr(out_sets[k]), C.PRL_BF16, v, st)
        assert rc == 0

    # old = lp so that every token has a gradient: one warm call per variant fills lp
    call(next(iter(libs.values())))
    torch.cuda.synchronize()
    old.copy_(lp)
    times = {tag: [] for tag in libs}
    order = [tag for tag in libs if tag != "trace"]
    for _ in range(rounds):
        random.shuffle(order)
        for tag in order:
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(5):
                call(libs[tag])
            b.record()
            torch.cuda.synchronize()
            times[tag].append(a.elapsed_time(b) / 5)
    if "trace" in libs:
        call(libs["trace"])
        torch.cuda.synchronize()
        for name, arr in (("launch -> first chunk here", lp), ("sums pass and merge", ent), ("gradient pass", lse)):
            x = arr[1:].double() / 100.0
            print(f"trace: {name:30s} mean {x.mean():6.2f} us  min {x.min():6.2f}  max {x.max():6.2f}")
        del libs["trace"], times["trace"]
    nbytes = (2 * t - 1) * v * 2
    for tag, ts in sorted(times.items(), key=lambda kv: statistics.median(kv[1])):
        med = statistics.median(ts)
        print(f"{tag:28s} median {med * 1e3:7.1f} us  min {min(ts) * 1e3:7.1f} us  {nbytes / med / 1e6:7.0f} GB/s", flush=True)


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build()
    else:
        run()
