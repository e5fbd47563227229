"""`python bench.py --gpus N` by itself: the parent starts N fresh rank processes, passes rank 0's one JSON line
through and returns the worst exit code, and never makes a GPU call of its own (what the reference's launcher
does for its trainer ranks, src/prime_rl/rl.py:589-651).  All on the CPU: the ranks meet over gloo."""

import json
import os
import subprocess
import sys
import time
from pathlib import Path

import pytest
import torch

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
import bench  # noqa: E402


def test_rank_commands_carry_the_rendezvous_of_one_node():
    cmds = bench.rank_commands(4, ["--gpus", "4", "--steps", "7"], 29999, environ={"HOME": "/x", "WORLD_SIZE": "9"})
    assert len(cmds) == 4
    for r, (argv, env) in enumerate(cmds):
        assert argv[0] == sys.executable and Path(argv[2]) == REPO / "bench.py"
        assert argv[3:] == ["--gpus", "4", "--steps", "7"]
        assert env["RANK"] == env["LOCAL_RANK"] == str(r)
        assert env["WORLD_SIZE"] == env["LOCAL_WORLD_SIZE"] == "4"
        assert env["MASTER_ADDR"] == "127.0.0.1" and env["MASTER_PORT"] == "29999"
        assert env["HOME"] == "/x" and env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


@pytest.mark.parametrize("n", [2, 8])
def test_ranks_meet_over_gloo_and_one_line_comes_out(n):
    """`python bench.py --gpus N --launch-check` as the driver starts it for N = 2 and for the whole node's 8:
    the parent starts the ranks, they rendezvous (gloo here: no GPU), all-reduce one number, and exactly one
    line reaches stdout."""
    env = {**os.environ, "PRL_BENCH_BACKEND": "gloo"}
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, str(REPO / "bench.py"), "--gpus", str(n), "--launch-check"], env=env,
                         capture_output=True, text=True, timeout=400)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [x for x in out.stdout.splitlines() if x.strip()]
    assert len(lines) == 1  # rank 0's line and nothing else on stdout
    line = json.loads(lines[0])
    assert line["ranks"] == n and line["n_gpus"] == n and line["value"] == float(n * (n - 1) // 2)  # sum of ranks
    assert line["devices"] == [f"rank {r}: cpu" for r in range(n)]
    assert line["gpu_initialised"] is False


def _fake_ranks(monkeypatch, scripts):
    def commands(n, argv, port, environ=None):
        assert n == len(scripts)
        return [([sys.executable, "-c", code], dict(os.environ)) for code in scripts]

    monkeypatch.setattr(bench, "rank_commands", commands)


def test_parent_relays_only_the_json_line_and_never_touches_the_gpu(monkeypatch, capfd):
    _fake_ranks(monkeypatch, [
        'print("noise from rank 0"); print(\'{"metric": "m", "value": 1}\'); print(\'{"other": 1}\')',
        'print("noise from rank 1")',
    ])
    assert bench.launch_ranks(2, []) == 0
    out, err = capfd.readouterr()
    assert out.strip() == '{"metric": "m", "value": 1}'
    assert "noise from rank 0" in err and "noise from rank 1" in err and '{"other": 1}' in err
    assert not torch.cuda.is_initialized()


def test_a_failing_rank_ends_the_others_and_sets_the_exit_code(monkeypatch, capfd):
    _fake_ranks(monkeypatch, ["import time; time.sleep(600)", "import sys; sys.exit(7)"])
    t0 = time.monotonic()
    assert bench.launch_ranks(2, [], grace=1.0) == 137  # rank 0 was killed (128 + SIGKILL), rank 1 left with 7
    assert time.monotonic() - t0 < 60
    assert "rank exit codes" in capfd.readouterr().err


def test_the_hung_extra_has_a_nonzero_exit_code():
    assert bench.EXIT_TRAINER_EXTRA_HUNG != 0
    src = (REPO / "bench.py").read_text()
    assert "os._exit(EXIT_TRAINER_EXTRA_HUNG)" in src and "os._exit(0)" not in src


def test_only_the_line_reaches_stdout_whatever_libraries_print():
    """RCCL prints a version banner to file descriptor 1 when its first communicator is made; after
    `keep_stdout_for_the_line` such output (and Python's own prints) lands on stderr and stdout carries the line."""
    import subprocess
    import sys
    from pathlib import Path

    code = ("import os, bench; bench.keep_stdout_for_the_line(); os.write(1, b'RCCL version : x\\n'); "
            "print('progress'); print('{\"metric\": \"m\"}', file=bench.LINE_OUT, flush=True)")
    done = subprocess.run([sys.executable, "-c", code], cwd=Path(__file__).resolve().parent.parent,
                          capture_output=True, text=True, timeout=120)
    assert done.returncode == 0, done.stderr
    assert done.stdout == '{"metric": "m"}\n'
    assert "RCCL version : x" in done.stderr and "progress" in done.stderr
