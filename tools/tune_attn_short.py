"""The short-sample attention backward built in variants (one fused kernel per group against the dQ and dK/dV
pair, or any other build-time knob), timed side by side through the C-ABI on rows of 128-token samples and on a
ragged row (16 query / 8 key-value heads of 128, bf16), HIP events, random order; every variant is first held to
the default's results.

    python tools/tune_attn_short.py build | run  ["-DPRL_ATTN_...=.. -D..." ...]   (the same variants for both)
"""
import ctypes, random, statistics, subprocess, sys
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
CSRC = REPO / "prime-rl-cse_amd" / "csrc"
LIB = REPO / "prime-rl-cse_amd" / "lib"
TUNE = LIB / "tune"
VARIANTS = {
    "default": [],
    "fused_always": ["-DPRL_ATTN_SHORT_FUSED_MIN=0"],
    "fused_never": ["-DPRL_ATTN_SHORT_FUSED_MIN=1073741824"],
}
if len(sys.argv) > 2:  # variants named on the command line: one argument each, its -D flags separated by spaces
    VARIANTS = {"default": [], **{e.replace("-DPRL_ATTN_", "").replace("=", "").replace(" ", "_"): e.split()
                                  for e in sys.argv[2:]}}


def build():
    subprocess.run(["make", "-C", str(CSRC), "-j8"], check=True)
    TUNE.mkdir(parents=True, exist_ok=True)
    others = [str(LIB / "obj" / f"{n}.o") for n in ("capi",)]
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Xclang", "-target-feature", "-Xclang",
             "-packed-fp32-ops", f"-I{REPO / 'include'}", f"-I{CSRC}"]
    for name, defs in VARIANTS.items():
        obj = TUNE / f"attns_{name}.o"
        subprocess.run(["/opt/rocm/bin/hipcc", *flags, *defs, "-c", str(CSRC / "attn_short.hip"), "-o", str(obj)],
                       check=True, stderr=subprocess.DEVNULL)
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o",
                        str(TUNE / f"libattns_{name}.so"), str(obj), *others], check=True)
        obj.unlink()
        print("built", name, flush=True)


def run():
    import torch
    sys.path.insert(0, str(REPO))
    from prime_rl_cse_amd import _capi as C

    dev = torch.device("cuda:0")
    hq, hk, d = 16, 8, 128
    scale = d ** -0.5
    st = C.stream_ptr(dev)
    s2 = lambda x: (C.c_int64 * 2)(x.stride(0), x.stride(1))
    libs = {}
    for name in VARIANTS:
        lib = ctypes.CDLL(str(TUNE / f"libattns_{name}.so"))
        for fn in ("prl_attn_short_fwd", "prl_attn_short_bwd"):
            getattr(lib, fn).restype, getattr(lib, fn).argtypes = C.SIGNATURES[fn]
        libs[name] = lib
    gen = torch.Generator(device=dev).manual_seed(0)
    rng = random.Random(1)
    cases = {"128 x 128": [128] * 128, "16 x 128": [128] * 16, "64 x 128": [128] * 64,
             "ragged 1..128 (16 K tokens)": None}
    for what, lens in cases.items():
        if lens is None:
            lens, left = [], 16384
            while left:
                lens.append(min(left, rng.randint(1, 128)))
                left -= lens[-1]
        n, t = len(lens), sum(lens)
        mk = lambda h: torch.randn(t, h, d, generator=gen, device=dev).to(torch.bfloat16)
        q, k, v, do = mk(hq), mk(hk), mk(hk), mk(hq)
        cu = torch.tensor([0] + list(torch.tensor(lens).cumsum(0)), dtype=torch.int32, device=dev)
        o, lse = torch.empty_like(q), torch.empty(hq, t, device=dev)
        assert libs["default"].prl_attn_short_fwd(C.ptr(q), s2(q), C.ptr(k), s2(k), C.ptr(v), s2(v), C.ptr(o), s2(o),
                                                  C.ptr(lse), C.ptr(cu), n, t, hq, hk, scale, st) == 0
        outs = {name: dict(delta=torch.empty(hq, t, device=dev), dq=torch.zeros_like(q), dk=torch.zeros_like(k),
                           dv=torch.zeros_like(v)) for name in VARIANTS}

        def bwd(name):
            b = outs[name]
            assert libs[name].prl_attn_short_bwd(C.ptr(q), s2(q), C.ptr(k), s2(k), C.ptr(v), s2(v), C.ptr(o), s2(o),
                                                 C.ptr(do), s2(do), C.ptr(lse), C.ptr(b["delta"]), C.ptr(b["dq"]),
                                                 s2(b["dq"]), C.ptr(b["dk"]), s2(b["dk"]), C.ptr(b["dv"]), s2(b["dv"]),
                                                 C.ptr(cu), n, t, hq, hk, scale, st) == 0

        for name in VARIANTS:
            bwd(name)
        torch.cuda.synchronize()
        for name in VARIANTS:
            for key in ("dq", "dk", "dv"):
                a, b = outs["default"][key].float(), outs[name][key].float()
                worst = (a - b).abs().max().item()
                assert worst <= 2e-2 * a.abs().max().item(), (what, name, key, worst)
        times = {name: [] for name in VARIANTS}
        order = list(VARIANTS)
        for _ in range(10):
            random.shuffle(order)
            for name in order:
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for _ in range(5):
                    bwd(name)
                b.record()
                torch.cuda.synchronize()
                times[name].append(a.elapsed_time(b) / 5 * 1e3)
        print(f"{what}: variants agree; backward " +
              "  ".join(f"{name} {statistics.median(ts):6.1f} us" for name, ts in times.items()), flush=True)


if __name__ == "__main__":
    build() if sys.argv[1] == "build" else run()
