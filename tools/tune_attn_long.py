"""The long-sample attention kernels built with different LDS row pitches and dK/dV splits, timed side by side
on four samples of 4096 tokens (16 query / 8 key-value heads of 128, bf16), HIP events, random order; every
variant is first held to the default's results.

    python tools/tune_attn_long.py build | run  ["-DPRL_ATTN_...=.. -D..." ...]   (the same variants for both)
"""
import ctypes, random, statistics, subprocess, sys
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
CSRC = REPO / "prime-rl-cse_amd" / "csrc"
LIB = REPO / "prime-rl-cse_amd" / "lib"
TUNE = LIB / "tune"
VARIANTS = {
    "default": [],
    "pitch144": ["-DPRL_ATTN_PITCH=144"],
    "pitch152": ["-DPRL_ATTN_PITCH=152"],
    "dkv_split2": ["-DPRL_ATTN_DKV_SPLIT=2"],
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
        obj = TUNE / f"attn_{name}.o"
        subprocess.run(["/opt/rocm/bin/hipcc", *flags, *defs, "-c", str(CSRC / "attn_long.hip"), "-o", str(obj)],
                       check=True, stderr=subprocess.DEVNULL)
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o",
                        str(TUNE / f"libattn_{name}.so"), str(obj), *others], check=True)
        obj.unlink()
        print("built", name, flush=True)


def run():
    import torch
    sys.path.insert(0, str(REPO))
    from prime_rl_cse_amd import _capi as C

    dev = torch.device("cuda:0")
    n, L, hq, hk, d = 4, 4096, 16, 8, 128
    t = n * L
    gen = torch.Generator(device=dev).manual_seed(0)
    mk = lambda h: torch.randn(t, h, d, generator=gen, device=dev).to(torch.bfloat16)
    q, k, v, do = mk(hq), mk(hk), mk(hk), mk(hq)
    cu = torch.arange(0, t + 1, L, dtype=torch.int32, device=dev)
    st = C.stream_ptr(dev)
    s2 = lambda x: (C.c_int64 * 2)(x.stride(0), x.stride(1))
    libs, outs = {}, {}
    for name in VARIANTS:
        lib = ctypes.CDLL(str(TUNE / f"libattn_{name}.so"))
        for fn in ("prl_attn_long_fwd", "prl_attn_long_bwd"):
            getattr(lib, fn).restype, getattr(lib, fn).argtypes = C.SIGNATURES[fn]
        libs[name] = lib
        outs[name] = dict(o=torch.empty_like(q), lse=torch.empty(hq, t, device=dev), delta=torch.empty(hq, t, device=dev),
                          dq=torch.empty_like(q), dk=torch.empty_like(k), dv=torch.empty_like(v))
    scale = d ** -0.5

    def fwd(name):
        b = outs[name]
        assert libs[name].prl_attn_long_fwd(C.ptr(q), s2(q), C.ptr(k), s2(k), C.ptr(v), s2(v), C.ptr(b["o"]), s2(b["o"]),
                                            C.ptr(b["lse"]), C.ptr(cu), n, L, t, hq, hk, scale, st) == 0

    def bwd(name):
        b = outs[name]
        assert libs[name].prl_attn_long_bwd(C.ptr(q), s2(q), C.ptr(k), s2(k), C.ptr(v), s2(v), C.ptr(b["o"]), s2(b["o"]),
                                            C.ptr(do), s2(do), C.ptr(b["lse"]), C.ptr(b["delta"]), C.ptr(b["dq"]),
                                            s2(b["dq"]), C.ptr(b["dk"]), s2(b["dk"]), C.ptr(b["dv"]), s2(b["dv"]),
                                            C.ptr(cu), n, L, t, hq, hk, scale, st) == 0

    for name in VARIANTS:
        fwd(name), bwd(name)
    torch.cuda.synchronize()
    for name in VARIANTS:
        for key in ("o", "dq", "dk", "dv"):
            a, b = outs["default"][key].float(), outs[name][key].float()
            assert (a - b).abs().max().item() <= 2e-2 * a.abs().max().item(), (name, key)
    print("all variants agree", flush=True)
    flop = n * hq * (2 * 2 * L * L * d) / 2
    for what, fn, units in (("forward", fwd, 1.0), ("backward", bwd, 2.5)):
        times = {name: [] for name in VARIANTS}
        order = list(VARIANTS)
        for _ in range(8):
            random.shuffle(order)
            for name in order:
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for _ in range(3):
                    fn(name)
                b.record()
                torch.cuda.synchronize()
                times[name].append(a.elapsed_time(b) / 3 * 1e3)
        print(f"{what}: " + "  ".join(f"{name} {statistics.median(ts):7.0f} us ({units * flop / statistics.median(ts) / 1e6:5.0f} TF/s)"
                                       for name, ts in times.items()), flush=True)


if __name__ == "__main__":
    build() if sys.argv[1] == "build" else run()
