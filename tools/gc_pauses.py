"""How long Python's garbage collector stops a script: runs it with a callback on every collection and prints, at
the end, the collections per generation, their total time and the ones longer than 5 ms (with the time since start).

    python3 tools/gc_pauses.py script.py [args ...]
"""
import gc
import runpy
import sys
import time

t_start = time.perf_counter()
began = {}
pauses = []


def callback(phase: str, info: dict) -> None:
    if phase == "start":
        began[info["generation"]] = time.perf_counter()
    else:
        t0 = began.pop(info["generation"], None)
        if t0 is not None:
            pauses.append((info["generation"], time.perf_counter() - t0, t0 - t_start, info.get("collected", 0)))


gc.callbacks.append(callback)
sys.argv = sys.argv[1:]
try:
    runpy.run_path(sys.argv[0], run_name="__main__")
finally:
    for gen in (0, 1, 2):
        sel = [p for p in pauses if p[0] == gen]
        print(f"gc_pauses: generation {gen}: {len(sel)} collections, {1e3 * sum(p[1] for p in sel):.1f} ms in all")
    for gen, dur, at, n in pauses:
        if dur > 0.005:
            print(f"gc_pauses: generation {gen} at {at:8.2f} s: {1e3 * dur:7.1f} ms, {n} objects collected")
