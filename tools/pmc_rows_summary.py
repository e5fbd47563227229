"""Mean per launch of every counter that rocprofv3 --pmc runs collected for one kernel.

    python tools/pmc_rows_summary.py <kernel substring> <out.json> <run dir> [<run dir> ...]

SQ_WAVE_CYCLES, SQ_WAIT_* and SQ_ACTIVE_INST_* count quad-cycles summed over all waves; SQ_BUSY_CYCLES counts per
shader engine.  The derived shares are of SQ_WAVE_CYCLES (WAIT_ANY: parked on s_waitcnt or a barrier;
WAIT_INST_ANY: ready but not issued; ACTIVE_INST_ANY: issuing)."""
from __future__ import annotations

import csv
import json
import sys
from collections import defaultdict
from pathlib import Path


def main() -> None:
    kernel, out_path, roots = sys.argv[1], Path(sys.argv[2]), [Path(p) for p in sys.argv[3:]]
    sums: dict[str, dict[str, float]] = defaultdict(lambda: defaultdict(float))
    for root in roots:
        for path in root.rglob("*counter_collection.csv"):
            with open(path, newline="") as fh:
                for row in csv.DictReader(fh):
                    if kernel in row["Kernel_Name"]:
                        sums[row["Counter_Name"]][row["Dispatch_Id"]] += float(row["Counter_Value"])
    means = {name: sum(d.values()) / len(d) for name, d in sorted(sums.items())}
    launches = {name: len(d) for name, d in sums.items()}
    derived = {}
    wc = means.get("SQ_WAVE_CYCLES")
    if wc:
        for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_VMEM"):
            if k in means:
                derived[f"{k}/SQ_WAVE_CYCLES"] = means[k] / wc
    if "SQ_INSTS_VALU" in means and "SQ_WAVES" in means:
        derived["VALU instructions per wave"] = means["SQ_INSTS_VALU"] / means["SQ_WAVES"]
    for k in ("SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR"):
        if k in means and "SQ_WAVES" in means:
            derived[f"{k} per wave"] = means[k] / means["SQ_WAVES"]
    doc = {"kernel": kernel, "mean_per_launch": means, "launches_seen": launches, "derived": derived}
    out_path.write_text(json.dumps(doc, indent=1))
    print(json.dumps(doc, indent=1))


if __name__ == "__main__":
    main()
