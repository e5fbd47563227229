"""One line per kernel from a tools/pmc_rows_summary.py JSON: every counter as a share of SQ_WAVE_CYCLES.

    python3 tools/pmc_print.py <summary.json>
"""
import json, sys

d = json.load(open(sys.argv[1]))
m = d["mean_per_launch"]
wc = m["SQ_WAVE_CYCLES"]
print(d["kernel"], {k: round(v / wc, 3) for k, v in m.items() if k != "SQ_WAVE_CYCLES"})
