"""The LM head's three products on this package's MFMA kernels against torch's (hipBLASLt) products.

    python tools/lm_head_bench.py [--t 16384] [--h 1024] [--v 151936] [--rounds 7] [--check-only]

1. exactness: small-integer operands (every partial sum is an integer fp32 holds exactly, whatever the order of the
   additions), so the kernels' bf16 results must equal the rounded fp32 product bit for bit, on shapes with ragged
   edges;
2. random operands at the asked shape: largest difference from the fp32 product, next to the library's;
3. time per product, both implementations interleaved in random order in one process (medians over the rounds).
"""

from __future__ import annotations

import argparse
import json
import random
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import prime_rl_cse_amd  # noqa: E402,F401
from prime_rl_cse_amd import lm_head as L  # noqa: E402


def ints(shape, lo, hi, gen, dev):
    return torch.randint(lo, hi + 1, shape, generator=gen, device=dev).to(torch.bfloat16)


def exact_cases(dev):
    gen = torch.Generator(device=dev).manual_seed(1)
    bad = 0
    for t, h, v in [(256, 64, 256), (512, 128, 1088), (192, 192, 832), (1024, 256, 4160), (2048, 1024, 2560),
                    (1088, 128, 16448), (5120, 3840, 2048)]:
        x = ints((t, h), -2, 2, gen, dev)
        w = ints((v, h), -2, 2, gen, dev)
        g = ints((t, v), -1, 1, gen, dev)
        want = {
            "fwd": (x.float() @ w.float().t()).to(torch.bfloat16),
            "dx": (g.float() @ w.float()).to(torch.bfloat16),
            "dw": (g.float().t() @ x.float()).to(torch.bfloat16),
        }
        got = {"fwd": L.lm_head_fwd(x, w), "dx": L.lm_head_bwd_input(g, w), "dw": L.lm_head_bwd_weight(g, x)}
        torch.cuda.synchronize()
        for k in want:
            same = torch.equal(got[k], want[k])
            nbad = 0 if same else int((got[k] != want[k]).sum())
            print(f"exact T={t} H={h} V={v} {k}: {'ok' if same else f'{nbad} of {want[k].numel()} differ'}", flush=True)
            if not same:
                bad += 1
                idx = (got[k] != want[k]).nonzero()[:5].tolist()
                print("   first differing (row, col):", idx,
                      [(float(got[k][i, j]), float(want[k][i, j])) for i, j in idx], flush=True)
    return bad


def timed_ms(fn, reps):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    b.synchronize()
    return a.elapsed_time(b) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--t", type=int, default=16384)
    ap.add_argument("--h", type=int, default=1024)
    ap.add_argument("--v", type=int, default=151936)
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--check-only", action="store_true")
    ap.add_argument("--skip-exact", action="store_true")
    ap.add_argument("--json", type=str, default="")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    print(torch.cuda.get_device_name(0), flush=True)
    if not args.skip_exact:
        bad = exact_cases(dev)
        if bad:
            print(f"{bad} exactness cases FAILED", flush=True)
            sys.exit(1)
    if args.check_only:
        return
    t, h, v = args.t, args.h, args.v
    gen = torch.Generator(device=dev).manual_seed(0)
    x = (torch.randn((t, h), generator=gen, device=dev) * 1.0).to(torch.bfloat16)
    w = (torch.randn((v, h), generator=gen, device=dev) * 0.05).to(torch.bfloat16)
    g = (torch.randn((t, v), generator=gen, device=dev) * 0.01).to(torch.bfloat16)
    ours = {"fwd": lambda: L.lm_head_fwd(x, w), "dx": lambda: L.lm_head_bwd_input(g, w),
            "dw": lambda: L.lm_head_bwd_weight(g, x)}
    lib = {"fwd": lambda: torch.nn.functional.linear(x, w), "dx": lambda: g @ w, "dw": lambda: g.t() @ x}
    # numerics on a slice of rows (the fp32 product of everything would need T x V x 4 bytes more)
    rows = min(t, 1024)
    ref = {"fwd": x[:rows].float() @ w.float().t(), "dx": None, "dw": None}
    report = {"shape": {"T": t, "H": h, "V": v}, "device": torch.cuda.get_device_name(0)}
    o, l_ = ours["fwd"](), lib["fwd"]()
    e_o = float((o[:rows].float() - ref["fwd"]).abs().max())
    e_l = float((l_[:rows].float() - ref["fwd"]).abs().max())
    print(f"fwd: max |ours - fp32| {e_o:.4g}, max |library - fp32| {e_l:.4g}, max |ref| {float(ref['fwd'].abs().max()):.4g}",
          flush=True)
    report["fwd_err"] = {"ours": e_o, "library": e_l}
    del o, l_, ref
    r32 = g.float() @ w.float()
    o, l_ = ours["dx"](), lib["dx"]()
    e_o, e_l = float((o.float() - r32).abs().max()), float((l_.float() - r32).abs().max())
    print(f"dx: max |ours - fp32| {e_o:.4g}, max |library - fp32| {e_l:.4g}, max |ref| {float(r32.abs().max()):.4g}",
          flush=True)
    report["dx_err"] = {"ours": e_o, "library": e_l}
    del o, l_, r32
    r32 = (g[:, :4096].float().t() @ x.float())
    o, l_ = ours["dw"](), lib["dw"]()
    e_o = float((o[:4096].float() - r32).abs().max())
    e_l = float((l_[:4096].float() - r32).abs().max())
    print(f"dw: max |ours - fp32| {e_o:.4g}, max |library - fp32| {e_l:.4g}, max |ref| {float(r32.abs().max()):.4g}",
          flush=True)
    report["dw_err"] = {"ours": e_o, "library": e_l}
    del o, l_, r32
    torch.cuda.synchronize()

    flop = 2.0 * t * h * v
    times = {(k, who): [] for k in ours for who in ("ours", "library")}
    for k in ours:  # warm up (kernel selection inside the library, allocator)
        for _ in range(2):
            ours[k]()
            lib[k]()
    torch.cuda.synchronize()
    rnd = random.Random(0)
    for _ in range(args.rounds):
        order = list(times)
        rnd.shuffle(order)
        for k, who in order:
            fn = ours[k] if who == "ours" else lib[k]
            times[(k, who)].append(timed_ms(fn, args.reps))
    report["products"] = {}
    for k in ours:
        a = sorted(times[(k, "ours")])
        b = sorted(times[(k, "library")])
        ma, mb = a[len(a) // 2], b[len(b) // 2]
        report["products"][k] = {"ours_ms": ma, "library_ms": mb, "ours_tflops": flop / ma / 1e9,
                                 "library_tflops": flop / mb / 1e9, "ours_min_ms": a[0], "library_min_ms": b[0]}
        print(f"{k}: ours {ma:.3f} ms ({flop / ma / 1e9:.0f} TFLOP/s, min {a[0]:.3f})   "
              f"library {mb:.3f} ms ({flop / mb / 1e9:.0f} TFLOP/s, min {b[0]:.3f})   ratio {mb / ma:.3f}", flush=True)
    if args.json:
        Path(args.json).parent.mkdir(parents=True, exist_ok=True)
        Path(args.json).write_text(json.dumps(report, indent=1))


if __name__ == "__main__":
    main()
