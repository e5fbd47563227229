"""RMSNorm kernels with different grid caps, timed with HIP events in one process (random order).

    python tools/tune_norm.py build | run
"""
import ctypes, random, statistics, subprocess, sys
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
CSRC = REPO / "prime-rl-cse_amd" / "csrc"
LIB = REPO / "prime-rl-cse_amd" / "lib"
TUNE = LIB / "tune"
CAPS = (128, 256, 512, 1024, 2048, 4096)


def build():
    subprocess.run(["make", "-C", str(CSRC), "-j8"], check=True)
    TUNE.mkdir(parents=True, exist_ok=True)
    others = [str(LIB / "obj" / f"{n}.o") for n in ("capi",)]
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Xclang", "-target-feature", "-Xclang",
             "-packed-fp32-ops", f"-I{REPO / 'include'}", f"-I{CSRC}"]
    for cap in CAPS:
        obj = TUNE / f"norm_{cap}.o"
        subprocess.run(["/opt/rocm/bin/hipcc", *flags, f"-DPRL_NORM_MAX_BLOCKS={cap}", "-c", str(CSRC / "norm.hip"), "-o", str(obj)],
                       check=True, stderr=subprocess.DEVNULL)
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o",
                        str(TUNE / f"libnorm_{cap}.so"), str(obj), *others], check=True)
        obj.unlink()
        print("built", cap, flush=True)


def run():
    import torch
    sys.path.insert(0, str(REPO))
    from prime_rl_cse_amd import _capi as C
    dev = torch.device("cuda:0")
    st = C.stream_ptr(dev)
    for rows, h in ((16384, 1024), (2048, 1024), (32768, 128), (16384, 128), (262144, 128)):
        x = torch.randn(rows, h, device=dev).to(torch.bfloat16)
        w = torch.ones(h, device=dev, dtype=torch.bfloat16)
        gy = torch.randn(rows, h, device=dev).to(torch.bfloat16)
        y, gx = torch.empty_like(x), torch.empty_like(x)
        gw = torch.empty_like(w)
        rstd = torch.empty(rows, device=dev)
        libs = {}
        for cap in CAPS:
            lib = ctypes.CDLL(str(TUNE / f"libnorm_{cap}.so"))
            for name in ("prl_rmsnorm_fwd", "prl_rmsnorm_bwd", "prl_rmsnorm_bwd_workspace_bytes"):
                getattr(lib, name).restype, getattr(lib, name).argtypes = C.SIGNATURES[name]
            ws = torch.empty(lib.prl_rmsnorm_bwd_workspace_bytes(C.PRL_BF16, rows, h), dtype=torch.uint8, device=dev)
            libs[cap] = (lib, ws)

        def fwd(cap):
            lib, _ = libs[cap]
            assert lib.prl_rmsnorm_fwd(C.ptr(x), h, C.ptr(w), C.ptr(y), h, C.ptr(rstd), C.PRL_BF16, rows, h, 1e-6, 0, 0, st) == 0

        def bwd(cap):
            lib, ws = libs[cap]
            assert lib.prl_rmsnorm_bwd(C.ptr(x), h, C.ptr(w), C.ptr(rstd), C.ptr(gy), h, C.ptr(gx), h, C.ptr(gw), C.PRL_BF16,
                                       rows, h, C.ptr(ws), ws.numel(), 0, 0, st) == 0

        for name, fn in (("fwd", fwd), ("bwd", bwd)):
            times = {cap: [] for cap in CAPS}
            order = list(CAPS)
            for _ in range(12):
                random.shuffle(order)
                for cap in order:
                    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    a.record()
                    for _ in range(10):
                        fn(cap)
                    b.record()
                    torch.cuda.synchronize()
                    times[cap].append(a.elapsed_time(b) / 10 * 1e3)
            print(f"[{rows} x {h}] {name}: " + "  ".join(f"cap {cap}: {statistics.median(times[cap]):5.1f} us" for cap in CAPS), flush=True)


if __name__ == "__main__":
    build() if sys.argv[1] == "build" else run()
