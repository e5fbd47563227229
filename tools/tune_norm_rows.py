"""The RMSNorm backward with 1, 2 or 4 rows in flight per row slot (`-DPRL_NORM_BWD_ROWS`), with and without the
residual gradient added in the same pass, timed side by side with HIP events in random order.

    python tools/tune_norm_rows.py build | run
"""
import ctypes, random, statistics, subprocess, sys
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
CSRC = REPO / "prime-rl-cse_amd" / "csrc"
LIB = REPO / "prime-rl-cse_amd" / "lib"
TUNE = LIB / "tune"
ROWS = (1, 2, 4)


def build():
    subprocess.run(["make", "-C", str(CSRC), "-j8"], check=True)
    TUNE.mkdir(parents=True, exist_ok=True)
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Xclang", "-target-feature", "-Xclang",
             "-packed-fp32-ops", f"-I{REPO / 'include'}", f"-I{CSRC}"]
    for r in ROWS:
        obj = TUNE / f"normrows_{r}.o"
        subprocess.run(["/opt/rocm/bin/hipcc", *flags, f"-DPRL_NORM_BWD_ROWS={r}", "-c", str(CSRC / "norm.hip"), "-o", str(obj)],
                       check=True, stderr=subprocess.DEVNULL)
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o",
                        str(TUNE / f"libnormrows_{r}.so"), str(obj), str(LIB / "obj" / "capi.o")], check=True)
        obj.unlink()
        print("built", r, flush=True)


def run():
    import torch
    sys.path.insert(0, str(REPO))
    from prime_rl_cse_amd import _capi as C
    dev = torch.device("cuda:0")
    st = C.stream_ptr(dev)
    for rows, h in ((16384, 1024), (16384, 2048), (2048, 1024), (16384, 3584)):
        mk = lambda: torch.randn(rows, h, device=dev).to(torch.bfloat16)
        x, gy, ga = mk(), mk(), mk()
        w = torch.ones(h, device=dev, dtype=torch.bfloat16)
        rstd = torch.rand(rows, device=dev) + 0.5
        libs, outs = {}, {}
        for r in ROWS:
            lib = ctypes.CDLL(str(TUNE / f"libnormrows_{r}.so"))
            for name in ("prl_rmsnorm_bwd", "prl_add_rmsnorm_bwd", "prl_rmsnorm_bwd_workspace_bytes"):
                getattr(lib, name).restype, getattr(lib, name).argtypes = C.SIGNATURES[name]
            ws = torch.empty(lib.prl_rmsnorm_bwd_workspace_bytes(C.PRL_BF16, rows, h), dtype=torch.uint8, device=dev)
            libs[r] = (lib, ws)
            outs[r] = (torch.empty_like(x), torch.empty_like(w))

        def plain(r):
            (lib, ws), (gx, gw) = libs[r], outs[r]
            assert lib.prl_rmsnorm_bwd(C.ptr(x), h, C.ptr(w), C.ptr(rstd), C.ptr(gy), h, C.ptr(gx), h, C.ptr(gw), C.PRL_BF16,
                                       rows, h, C.ptr(ws), ws.numel(), 0, 0, st) == 0

        def added(r):
            (lib, ws), (gx, gw) = libs[r], outs[r]
            assert lib.prl_add_rmsnorm_bwd(C.ptr(x), h, C.ptr(w), C.ptr(rstd), C.ptr(gy), h, C.ptr(ga), h, C.ptr(gx), h,
                                           C.ptr(gw), C.PRL_BF16, rows, h, C.ptr(ws), ws.numel(), st) == 0

        for what, fn, nbytes in (("backward", plain, 3), ("backward + residual gradient", added, 4)):
            for r in ROWS:
                fn(r)
            torch.cuda.synchronize()
            assert all(torch.equal(outs[r][0], outs[1][0]) and torch.equal(outs[r][1], outs[1][1]) for r in ROWS)
            times = {r: [] for r in ROWS}
            order = list(ROWS)
            for _ in range(12):
                random.shuffle(order)
                for r in order:
                    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    a.record()
                    for _ in range(4):
                        fn(r)
                    b.record()
                    torch.cuda.synchronize()
                    times[r].append(a.elapsed_time(b) / 4 * 1e3)
            gb = nbytes * rows * h * 2 / 1e3
            print(f"{rows} x {h} {what} (with the column-sum finish): " + "  ".join(
                f"{r} row(s) {statistics.median(t):6.1f} us ({gb / statistics.median(t) / 1e3:4.2f} TB/s)" for r, t in times.items()), flush=True)


if __name__ == "__main__":
    build() if sys.argv[1] == "build" else run()
