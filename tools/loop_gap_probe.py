"""What one micro-batch of the headline loop costs beyond its row kernel, layer by layer (T = 2048, V = 151936):
the row kernel alone back to back, row kernel + tail through the C ABI, `policy_loss` + backward, and the same
with the statistics bookkeeping of `bench.py`.  HIP events around 24 iterations each, rotating buffers.

    python3 tools/loop_gap_probe.py
"""
from __future__ import annotations

import statistics
import sys
from pathlib import Path

import torch

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
import prime_rl_cse_amd as prl  # noqa: E402
from prime_rl_cse_amd import _capi as C  # noqa: E402

N = 24


def main() -> None:
    t, v = 2048, 151936
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(0)
    logits = [torch.empty(1, t, v, dtype=torch.bfloat16, device=dev) for _ in range(3)]
    for x in logits:
        for r in range(0, t, 512):
            x[0, r : r + 512] = (torch.randn(512, v, generator=g, device=dev) * 2.5).to(torch.bfloat16)
    out = [torch.empty(t, v, dtype=torch.bfloat16, device=dev) for _ in range(3)]
    ids = torch.randint(0, v, (1, t), generator=g, device=dev)
    pos = torch.arange(128, device=dev).repeat(t // 128).reshape(1, t)
    adv = torch.where(torch.rand(1, t, generator=g, device=dev) < 0.5, -0.5, 0.5)
    mask = torch.ones(1, t, dtype=torch.bool, device=dev)
    mask8 = mask.view(torch.uint8)
    cfg = prl.LossConfig()
    with torch.no_grad():
        old = prl.selective_log_softmax(prl.shift_logits(logits[0]), ids).clone()
    f = torch.empty(9 * t + 1, device=dev)
    ia = torch.empty(t + 2, dtype=torch.int64, device=dev)
    lib = C.load()
    st = C.stream_ptr(dev)
    fp, ip = f.data_ptr(), ia.data_ptr()
    at = lambda k: C.c_void_p(fp + 4 * t * k)

    def rows(k):
        assert lib.prl_policy_rows_fwd_bwd(C.ptr(logits[k % 3]), C.PRL_BF16, t, v, v, C.ptr(ids), C.ptr(adv), C.ptr(old),
                                           C.ptr(mask8), 0, 1.0, 8.0, float(t), at(0), at(1), at(2), C.ptr(out[k % 3]),
                                           C.PRL_BF16, v, st) == 0

    def tail(k):
        assert lib.prl_policy_tail(C.ptr(pos), at(0), C.ptr(old), C.ptr(adv), C.ptr(mask8), t, C.RATIO_TOKEN, 0, 8.0,
                                   float(t), at(8), at(4), at(5), at(6), at(3), at(7), None, C.c_void_p(ip),
                                   C.c_void_p(ip + 8 * (t + 1)), st) == 0

    def rows_tail(k):
        rows(k), tail(k)

    unit = prl.unit_gradient(dev)

    def loss_bwd(k):
        x = logits[k % 3].requires_grad_(True)
        x.grad = None
        loss, aux = prl.policy_loss(x, ids, pos, adv, old, mask, 1.0, cfg, float(t))
        loss.backward(unit)
        return loss, aux

    tensors = [None]
    futures = []

    def loss_bwd_stats(k):
        if k % 8 == 0 or tensors[0] is None:
            if tensors[0] is not None:
                futures.append(tensors[0].compute_stats_async())
                if len(futures) > 1:
                    futures.pop(0).result()
            tensors[0] = prl.Tensors()
        loss, aux = loss_bwd(k)
        ts = tensors[0]
        ts.add("probs", aux["logprobs"], mask, exp=True)
        ts.add("old_probs", old, mask, exp=True)
        ts.add("entropy", aux["entropy"], mask)
        ts.add("loss", loss.detach().reshape(1))
        for key in ("importance_ratio", "clipped_importance_ratio", "is_clipped"):
            ts.add(key, aux[key], mask)

    def loss_bwd_add_only(k):
        if k % 8 == 0 or tensors[0] is None:
            tensors[0] = prl.Tensors()
        loss, aux = loss_bwd(k)
        ts = tensors[0]
        ts.add("probs", aux["logprobs"], mask, exp=True)
        ts.add("old_probs", old, mask, exp=True)
        ts.add("entropy", aux["entropy"], mask)
        ts.add("loss", loss.detach().reshape(1))
        for key in ("importance_ratio", "clipped_importance_ratio", "is_clipped"):
            ts.add(key, aux[key], mask)

    import time
    lead = []

    def loss_bwd_stats_marked(k):
        if k % 8 == 0 and tensors[0] is not None:
            e = torch.cuda.Event()
            e.record()
            h0 = time.perf_counter()
            tensors[0].compute_stats_async()
            h1 = time.perf_counter()
            lead.append((e.query(), (h1 - h0) * 1e6))
            tensors[0] = None
        loss_bwd_add_only(k)

    for name, fn in (("row kernel alone", rows), ("row kernel + tail (C ABI)", rows_tail),
                     ("policy_loss + backward", loss_bwd), ("the same + Tensors.add, no reduction", loss_bwd_add_only),
                     ("the same + statistics every 8", loss_bwd_stats),
                     ("the same, marked", loss_bwd_stats_marked)):
        times = []
        for _ in range(5):
            fn(1), fn(2)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for k in range(N):
                fn(k)
            b.record()
            torch.cuda.synchronize()
            times.append(a.elapsed_time(b) / N * 1e3)
        print(f"{name:36s} {statistics.median(times):7.1f} us per micro-batch (min {min(times):.1f})", flush=True)
    print("at compute_stats_async: (GPU already idle?, host us in the call):", lead[-6:], flush=True)


if __name__ == "__main__":
    main()
