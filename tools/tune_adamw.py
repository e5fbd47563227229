"""The flat AdamW kernel with different numbers of vectors in flight per thread, cache hints and grid caps,
timed with HIP events in one process (random order) on the 0.6B model's shard size.

    python tools/tune_adamw.py build | run
"""
import ctypes, random, statistics, subprocess, sys
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
CSRC = REPO / "prime-rl-cse_amd" / "csrc"
LIB = REPO / "prime-rl-cse_amd" / "lib"
TUNE = LIB / "tune"
# (vectors per thread and trip, non-temporal streams, grid cap)
VARIANTS = ((1, 0, 16384), (4, 1, 16384), (2, 1, 65536), (1, 0, 1 << 22), (1, 1, 1 << 22), (2, 0, 1 << 22),
            (2, 1, 1 << 22), (1, 0, 1 << 18))
N = 596_049_920  # parameters of Qwen3-0.6B with tied embeddings


def name(v):
    return f"u{v[0]}_nt{v[1]}_b{v[2]}"


def build():
    subprocess.run(["make", "-C", str(CSRC), "-j8"], check=True)
    TUNE.mkdir(parents=True, exist_ok=True)
    others = [str(LIB / "obj" / f"{n}.o") for n in ("capi",)]
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Xclang", "-target-feature", "-Xclang",
             "-packed-fp32-ops", f"-I{REPO / 'include'}", f"-I{CSRC}"]
    for v in VARIANTS:
        obj = TUNE / f"adam_{name(v)}.o"
        subprocess.run(["/opt/rocm/bin/hipcc", *flags, f"-DPRL_ADAM_UNROLL={v[0]}", f"-DPRL_ADAM_NT={v[1]}",
                        f"-DPRL_ADAM_MAX_BLOCKS={v[2]}", "-c", str(CSRC / "optim.hip"), "-o", str(obj)],
                       check=True, stderr=subprocess.DEVNULL)
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o",
                        str(TUNE / f"libadam_{name(v)}.so"), str(obj), *others], check=True)
        obj.unlink()
        print("built", name(v), flush=True)


def run():
    import torch
    sys.path.insert(0, str(REPO))
    from prime_rl_cse_amd import _capi as C
    dev = torch.device("cuda:0")
    st = C.stream_ptr(dev)
    p = torch.randn(N, device=dev)
    g = torch.randn(N, device=dev) * 1e-3
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    copy = torch.empty(N, device=dev, dtype=torch.bfloat16)
    coef = torch.ones(1, device=dev)
    libs = {}
    for var in VARIANTS:
        lib = ctypes.CDLL(str(TUNE / f"libadam_{name(var)}.so"))
        lib.prl_adamw_flat.restype, lib.prl_adamw_flat.argtypes = C.SIGNATURES["prl_adamw_flat"]
        libs[var] = lib
    steps = [0]

    def step(var):
        steps[0] += 1
        rc = libs[var].prl_adamw_flat(C.ptr(p), C.ptr(g), C.ptr(m), C.ptr(v), C.ptr(copy), C.PRL_BF16, N, C.ptr(coef), 0,
                                      1e-6, 0.9, 0.999, 1e-8, 0.01, steps[0], st)
        assert rc == 0

    # every variant gives the same numbers (one step each from the same start)
    ref = None
    for var in VARIANTS:
        p0, m0, v0 = p.clone(), m.clone(), v.clone()
        libs[var].prl_adamw_flat(C.ptr(p0), C.ptr(g), C.ptr(m0), C.ptr(v0), C.ptr(copy), C.PRL_BF16, N, C.ptr(coef), 0,
                                 1e-6, 0.9, 0.999, 1e-8, 0.01, 1, st)
        got = (p0, m0, v0, copy.clone())
        if ref is None:
            ref = got
        else:
            assert all(torch.equal(a, b) for a, b in zip(ref, got)), name(var)
        del p0, m0, v0
    del ref, got
    print("all variants agree bit for bit", flush=True)
    times = {var: [] for var in VARIANTS}
    order = list(VARIANTS)
    for _ in range(8):
        random.shuffle(order)
        for var in order:
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(3):
                step(var)
            b.record()
            torch.cuda.synchronize()
            times[var].append(a.elapsed_time(b) / 3)
    nbytes = N * 30
    for var in VARIANTS:
        ms = statistics.median(times[var])
        print(f"{name(var):>18}: {ms:6.3f} ms  {nbytes / ms / 1e9:5.2f} TB/s", flush=True)


if __name__ == "__main__":
    build() if sys.argv[1] == "build" else run()
