"""The LM-head kernels on the shapes of a decoder block's linear layers (they are plain y = x W^T products), against
what the blocks use today: the BLAS library through torch for the forward and the input gradient, `fused.weight_grad`
(batched products over ranges of the tokens + `prl_sum_parts`) for the weight gradient.  Kernel and library calls of
a product are interleaved in random order; prints the median ms of each and the ratio.

    python tools/body_gemms_ab.py [tokens] [--json FILE]

With --json the ratios (library ms / kernel ms per product) are merged into FILE under the token count, keeping the
smaller ratio where FILE holds the shape already: that file, three runs each at 4096, 8192 and 16384 tokens, is the
package's `gemm_table.json`, from which `lm_head.layer_plan` decides
which products of a decoder block's layers run on the kernels.
"""
import json
import random
import statistics
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import prime_rl_cse_amd  # noqa: E402,F401
from prime_rl_cse_amd import fused, lm_head as L  # noqa: E402

import os  # noqa: E402

os.environ["PRL_LAYER_GEMMS"] = "lib"  # `fused.weight_grad` below is the library's side of the comparison
T = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 16384
JSON_OUT = sys.argv[sys.argv.index("--json") + 1] if "--json" in sys.argv else None
ratios: dict[str, list[float]] = {}
MODELS = {
    "Qwen3-0.6B": [("q+k+v", 1024, 4096), ("o", 2048, 1024), ("gate+up", 1024, 6144), ("down", 3072, 1024)],
    "Qwen3-1.7B": [("q+k+v", 2048, 4096), ("o", 2048, 2048), ("gate+up", 2048, 12288), ("down", 6144, 2048)],
    "Qwen2.5-7B": [("q+k+v", 3584, 4608), ("o", 3584, 3584), ("gate+up", 3584, 37888), ("down", 18944, 3584)],
    "Qwen2.5-32B": [("q+k+v", 5120, 7168), ("o", 5120, 5120), ("gate+up", 5120, 55296), ("down", 27648, 5120)],
    # the same models' layers one by one, as they run where their weights do not lie back to back in one buffer
    # (FSDP2) or carry a bias (Qwen2.5's q, k, v): gate or up alone, q alone, k or v alone
    "one by one": [("gate|up", 1024, 3072), ("gate|up", 2048, 6144), ("gate|up", 3584, 18944), ("gate|up", 5120, 27648),
                   ("q", 1024, 2048), ("k|v", 1024, 1024), ("k|v", 3584, 512), ("k|v", 5120, 1024)],
}
dev = torch.device("cuda:0")
gen = torch.Generator(device=dev).manual_seed(0)


def ms(fn, reps=5):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


total = {"lib": 0.0, "hip": 0.0, "best": 0.0}
for model, layers in MODELS.items():
    print(model, f"T={T}")
    for name, n_in, n_out in layers:
        x = torch.randn((T, n_in), generator=gen, device=dev).to(torch.bfloat16)
        w = (torch.randn((n_out, n_in), generator=gen, device=dev) * 0.05).to(torch.bfloat16)
        gy = torch.randn((T, n_out), generator=gen, device=dev).to(torch.bfloat16)
        products = {
            "forward": (lambda: x @ w.t(), lambda: L.lm_head_fwd(x, w)),
            "input gradient": (lambda: gy @ w, lambda: L.lm_head_bwd_input(gy, w)),
            "weight gradient": (lambda: fused.weight_grad(gy, x), lambda: L.lm_head_bwd_weight(gy, x)),
        }
        for what, (lib, hip) in products.items():
            for f in (lib, hip):
                f()
            times = {"lib": [], "hip": []}
            for _ in range(5):
                order = ["lib", "hip"]
                random.shuffle(order)
                for side in order:
                    times[side].append(ms(lib if side == "lib" else hip))
            a, b = statistics.median(times["lib"]), statistics.median(times["hip"])
            flops = 2.0 * T * n_in * n_out
            if model == "Qwen3-0.6B":
                total["lib"] += a
                total["hip"] += b
                total["best"] += min(a, b)
            ratios.setdefault(f"{n_in}x{n_out}", []).append(round(a / b, 3))
            print(f"  {name:8s} {n_in:5d} -> {n_out:5d}  {what:15s} library {a:7.3f} ms ({flops / a * 1e-9:6.0f} TFLOP/s)  "
                  f"kernels {b:7.3f} ms ({flops / b * 1e-9:6.0f})  library/kernels {a / b:5.2f}")
        del x, w, gy
print(f"Qwen3-0.6B, one block: library {total['lib']:.3f} ms, kernels {total['hip']:.3f} ms, the faster of the two per "
      f"product {total['best']:.3f} ms")
if JSON_OUT:
    path = Path(JSON_OUT)
    table = json.loads(path.read_text()) if path.exists() else {}
    table.setdefault("what", "library ms / kernel ms of (forward, input gradient, weight gradient) for y = x W^T with W "
                             "[out, in], keyed by 'in x out' and the number of rows of x; tools/body_gemms_ab.py on an MI355X")
    # a shape measured before keeps, product by product, the SMALLER ratio: the library's time for one and the same
    # shape moves by up to 10 % from run to run, and a product should move to the kernels only if they win every time
    for key, r in ratios.items():
        old = table.setdefault("ratios", {}).setdefault(key, {}).get(str(T))
        table["ratios"][key][str(T)] = r if old is None else [min(a, b) for a, b in zip(old, r)]
    path.write_text(json.dumps(table, indent=1) + "\n")
