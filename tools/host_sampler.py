"""Where the host threads are while a script runs: a sampler thread looks at every thread's Python frame every
2 ms and counts (thread, innermost frames) pairs; printed at the end, most frequent first.  For finding what
the trainer's main thread does (or waits for) between the GPU's kernels.

    python3 tools/host_sampler.py [--skip SECONDS] script.py [args ...]
"""
from __future__ import annotations

import collections
import runpy
import sys
import threading
import time


def main() -> None:
    argv = sys.argv[1:]
    skip = 0.0
    if argv and argv[0] == "--skip":
        skip, argv = float(argv[1]), argv[2:]
    counts: dict[tuple[str, str], int] = collections.Counter()
    stop = threading.Event()
    t_start = time.perf_counter()

    def sample() -> None:
        me = threading.get_ident()
        names = {}
        while not stop.wait(0.002):
            if time.perf_counter() - t_start < skip:
                continue
            for t in threading.enumerate():
                names[t.ident] = t.name
            for ident, frame in sys._current_frames().items():
                if ident == me:
                    continue
                where = []
                f = frame
                while f is not None and len(where) < 4:
                    where.append(f"{f.f_code.co_filename.rsplit('/', 1)[-1]}:{f.f_lineno}:{f.f_code.co_name}")
                    f = f.f_back
                counts[(names.get(ident, str(ident)), " < ".join(where))] += 1

    th = threading.Thread(target=sample, name="sampler", daemon=True)
    th.start()
    sys.argv = argv
    try:
        runpy.run_path(argv[0], run_name="__main__")
    finally:
        stop.set()
        th.join()
        total = collections.Counter()
        for (name, _), n in counts.items():
            total[name] += n
        print(f"host_sampler: samples per thread {dict(total)} (2 ms apart)")
        for (name, where), n in sorted(counts.items(), key=lambda kv: -kv[1])[:70]:
            print(f"{n:6d}  {name[:24]:24s} {where}")


if __name__ == "__main__":
    main()
