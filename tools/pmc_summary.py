"""Fold two rocprofv3 counter runs (one with --pmc FETCH_SIZE, one with --pmc WRITE_SIZE) into the HBM bytes
per launch of the vocabulary-wide kernels.

    python tools/pmc_summary.py <dir of the FETCH_SIZE run> <dir of the WRITE_SIZE run> <out.json>

Corrections for gfx950: both counters are in KiB; FETCH_SIZE tallies the 128-byte requests of wide coalesced
streaming reads at 64 bytes, so reads are doubled; WRITE_SIZE is exact.
"""
from __future__ import annotations

import csv
import json
import sys
from collections import defaultdict
from pathlib import Path

KERNELS = ("k_policy_rows", "k_logprob_bwd_flat", "k_logprob_bwd", "k_logprob_fwd")


def collect(root: Path, counter: str) -> dict[str, list[float]]:
    per_dispatch: dict[tuple[str, str], float] = defaultdict(float)
    for path in root.rglob("*counter_collection.csv"):
        with open(path, newline="") as fh:
            for row in csv.DictReader(fh):
                if row.get("Counter_Name") != counter:
                    continue
                name = row["Kernel_Name"]
                short = next((k for k in KERNELS if k in name), None)
                if short is None:
                    continue
                per_dispatch[(short, row["Dispatch_Id"])] += float(row["Counter_Value"])
    out: dict[str, list[float]] = defaultdict(list)
    for (short, _), v in per_dispatch.items():
        out[short].append(v)
    return out


def main() -> None:
    fetch_dir, write_dir, out_path = Path(sys.argv[1]), Path(sys.argv[2]), Path(sys.argv[3])
    fetch, write = collect(fetch_dir, "FETCH_SIZE"), collect(write_dir, "WRITE_SIZE")
    kernels = {}
    for k in KERNELS:
        if k in fetch and k in write:
            f = sum(fetch[k]) / len(fetch[k])
            w = sum(write[k]) / len(write[k])
            kernels[k] = {
                "FETCH_SIZE_KB_mean": f, "FETCH_SIZE_launches": len(fetch[k]),
                "WRITE_SIZE_KB_mean": w, "WRITE_SIZE_launches": len(write[k]),
                "hbm_read_bytes_per_launch": f * 1024 * 2, "hbm_write_bytes_per_launch": w * 1024,
                "hbm_bytes_per_launch": f * 1024 * 2 + w * 1024,
            }
    doc = {
        "shape": {"T": 2048, "V": 151936, "logits": sys.argv[4] if len(sys.argv) > 4 else "bf16"},
        "command": "rocprofv3 --pmc FETCH_SIZE (and, in a run of its own, --pmc WRITE_SIZE) --output-format csv -- "
                   "python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-trainer-extra --no-stats",
        "corrections": "FETCH_SIZE (KiB) x 1024 x 2 (gfx950 tallies 128-byte requests at 64 bytes for wide coalesced "
                       "streaming reads); WRITE_SIZE (KiB) x 1024",
        "kernels": kernels,
    }
    out_path.write_text(json.dumps(doc, indent=1))
    print(json.dumps(kernels, indent=1))


if __name__ == "__main__":
    main()
