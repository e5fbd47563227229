"""Launches the single-pass row kernel (prl_policy_rows_fwd_bwd) and nothing else of this package: the target of
`rocprofv3 --pmc` / `--kernel-trace` runs about that kernel (T = 2048, V = 151936, every token with a gradient).

    python3 tools/rows_only.py [launches] [library]
"""
from __future__ import annotations

import ctypes
import sys
from pathlib import Path

import torch

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
from prime_rl_cse_amd import _capi as C  # noqa: E402


def main() -> None:
    launches = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    path = sys.argv[2] if len(sys.argv) > 2 else str(C.LIB_PATH)
    t, v = 2048, 151936
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(0)
    # three logits / gradient buffers in turn (3.7 GB): no launch finds its rows in the 256 MiB Infinity Cache
    logits = [torch.empty(t, v, dtype=torch.bfloat16, device=dev) for _ in range(3)]
    for x in logits:
        for r in range(0, t, 512):
            x[r : r + 512] = (torch.randn(512, v, generator=g, device=dev) * 2.5).to(torch.bfloat16)
    out = [torch.empty(t, v, dtype=torch.bfloat16, device=dev) for _ in range(3)]
    ids = torch.randint(0, v, (t,), generator=g, device=dev)
    adv = torch.where(torch.rand(t, generator=g, device=dev) < 0.5, -0.5, 0.5)
    old = torch.full((t,), -12.0, device=dev)
    mask = torch.ones(t, dtype=torch.uint8, device=dev)
    lp, ent, lse = (torch.empty(t, device=dev) for _ in range(3))
    lib = ctypes.CDLL(path)
    fn = lib.prl_policy_rows_fwd_bwd
    fn.restype, fn.argtypes = C.SIGNATURES["prl_policy_rows_fwd_bwd"]
    st = C.stream_ptr(dev)

    def call(k: int) -> None:
        rc = fn(C.ptr(logits[k % 3]), C.PRL_BF16, t, v, v, C.ptr(ids), C.ptr(adv), C.ptr(old), C.ptr(mask), 0, 1.0, 1e30,
                float(t), C.ptr(lp), C.ptr(ent), C.ptr(lse), C.ptr(out[k % 3]), C.PRL_BF16, v, st)
        assert rc == 0

    call(0)
    torch.cuda.synchronize()
    old.copy_(lp)  # ratio 1 for every token of buffer 0, near 1 for the others: every row gets a gradient
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    call(1), call(2)
    a.record()
    for k in range(launches):
        call(k)
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / launches
    nbytes = (2 * t - 1) * v * 2 + 12 * t
    print(f"rows_only: {launches} launches, {ms * 1e3:.1f} us each, {nbytes / ms / 1e6:.0f} GB/s algorithmic", flush=True)


if __name__ == "__main__":
    main()
