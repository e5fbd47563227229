"""Does the single-pass row kernel cost the same per row whatever the number of rows one launch covers?  The same
16384 rows (V = 151936, bf16, every token with a gradient; two sets of buffers in turn, 20 GB in all) are run as one
launch of 16384 rows, two of 8192, four of 4096, eight of 2048 (what the headline benchmark launches), in random
order over several rounds; printed per form: the median time per 2048 rows.

    python3 tools/rows_by_length.py [rounds]
"""
from __future__ import annotations

import ctypes
import random
import statistics
import sys
from pathlib import Path

import torch

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
from prime_rl_cse_amd import _capi as C  # noqa: E402


def main() -> None:
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 7
    t_all, v = 16384, 151936
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(0)
    sets = 2
    logits = [torch.empty(t_all, v, dtype=torch.bfloat16, device=dev) for _ in range(sets)]
    for x in logits:
        for r in range(0, t_all, 512):
            x[r : r + 512] = (torch.randn(512, v, generator=g, device=dev) * 2.5).to(torch.bfloat16)
    out = [torch.empty(t_all, v, dtype=torch.bfloat16, device=dev) for _ in range(sets)]
    ids = torch.randint(0, v, (t_all,), generator=g, device=dev)
    adv = torch.where(torch.rand(t_all, generator=g, device=dev) < 0.5, -0.5, 0.5)
    old = [torch.full((t_all,), -12.0, device=dev) for _ in range(sets)]
    mask = torch.ones(t_all, dtype=torch.uint8, device=dev)
    lp, ent, lse = (torch.empty(t_all, device=dev) for _ in range(3))
    lib = ctypes.CDLL(str(C.LIB_PATH))
    fn = lib.prl_policy_rows_fwd_bwd
    fn.restype, fn.argtypes = C.SIGNATURES["prl_policy_rows_fwd_bwd"]
    st = C.stream_ptr(dev)
    es = 2  # bytes per logit

    def sweep(k: int, t: int) -> None:
        """All 16384 rows of buffer set k, `t` rows per launch (the first row of a launch plays the dropped one)."""
        for lo in range(0, t_all, t):
            hi = lo + t
            rc = fn(C.ptr(logits[k][lo:hi]), C.PRL_BF16, t, v, v, C.ptr(ids[lo:hi]), C.ptr(adv[lo:hi]),
                    C.ptr(old[k][lo:hi]), C.ptr(mask[lo:hi]), 0, 1.0, 1e30, float(t), C.ptr(lp[lo:hi]),
                    C.ptr(ent[lo:hi]), C.ptr(lse[lo:hi]), C.ptr(out[k][lo:hi]), C.PRL_BF16, v, st)
            assert rc == 0, rc

    for k in range(sets):  # old logprobs = the new ones of the 2048-row form: ratio 1, every row has a gradient
        sweep(k, 2048)
        torch.cuda.synchronize()
        old[k].copy_(lp)
    forms = [16384, 8192, 4096, 2048]
    times: dict[int, list[float]] = {t: [] for t in forms}
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    rng = random.Random(1)
    for r in range(rounds + 1):
        order = forms[:]
        rng.shuffle(order)
        for t in order:
            a.record()
            for k in range(sets):
                sweep(k, t)
            b.record()
            torch.cuda.synchronize()
            if r:
                times[t].append(a.elapsed_time(b) * 1e3 / (sets * t_all / 2048))
    for t in forms:
        med = statistics.median(times[t])
        nbytes = (2 * 2048 - 2048 / t) * v * es + 12 * 2048
        print(f"rows_by_length: {t:5d} rows per launch: median {med:6.1f} us per 2048 rows (min {min(times[t]):6.1f}), "
              f"{nbytes / med / 1e3:.0f} GB/s algorithmic", flush=True)


if __name__ == "__main__":
    main()
