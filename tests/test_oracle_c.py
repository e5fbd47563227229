"""The C restatement (oracle/ref_c.c) against the golden vectors and against the
torch oracle: integers bit for bit, doubles to 1e-6."""

import ctypes
import subprocess
from pathlib import Path

import numpy as np
import pytest
import torch

from oracle import ref_cpu as O
from tests.helpers import load_npz

REPO = Path(__file__).resolve().parent.parent
SO = REPO / "oracle" / "_build" / "libprl_oracle.so"
i64p = ctypes.POINTER(ctypes.c_int64)
f32p = ctypes.POINTER(ctypes.c_float)
f64p = ctypes.POINTER(ctypes.c_double)


@pytest.fixture(scope="module")
def lib():
    if not SO.exists():
        subprocess.run(["make", "-C", str(REPO / "oracle")], check=True)
    lib = ctypes.CDLL(str(SO))
    lib.prl_oracle_response_lengths.restype = ctypes.c_int64
    lib.prl_oracle_response_lengths.argtypes = [i64p, ctypes.c_int64, i64p]
    lib.prl_oracle_first_fit_decreasing.restype = ctypes.c_int64
    lib.prl_oracle_first_fit_decreasing.argtypes = [i64p, ctypes.c_int64, ctypes.c_int64, i64p]
    lib.prl_oracle_logprob_entropy_f64.restype = None
    lib.prl_oracle_logprob_entropy_f64.argtypes = [f32p, ctypes.c_int64, ctypes.c_int64, i64p, ctypes.c_double,
                                                   ctypes.c_int, f64p, f64p, f64p]
    lib.prl_oracle_logprob_bwd_f64.restype = None
    lib.prl_oracle_logprob_bwd_f64.argtypes = [f32p, ctypes.c_int64, ctypes.c_int64, i64p, f64p, f64p,
                                               ctypes.c_double, ctypes.c_int, f64p]
    lib.prl_oracle_group_advantage.restype = None
    lib.prl_oracle_group_advantage.argtypes = [f64p, i64p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int, f64p]
    return lib


def p(a, t):
    return a.ctypes.data_as(t)


def test_response_lengths(lib):
    d = load_npz("response_lengths.npz")
    for n in sorted(k[: -len("__pos")] for k in d if k.endswith("__pos")):
        pos = np.ascontiguousarray(d[n + "__pos"].numpy())
        out = np.zeros(max(len(pos), 1), dtype=np.int64)
        k = lib.prl_oracle_response_lengths(p(pos, i64p), len(pos), p(out, i64p))
        assert out[:k].tolist() == d[n + "__len"].tolist(), n


def test_first_fit_decreasing(lib):
    gen = np.random.default_rng(0)
    for _ in range(50):
        n = int(gen.integers(1, 60))
        cap = int(gen.integers(16, 200))
        lens = gen.integers(1, cap + 1, n).astype(np.int64)
        bin_of = np.zeros(n, dtype=np.int64)
        nb = lib.prl_oracle_first_fit_decreasing(p(lens, i64p), n, cap, p(bin_of, i64p))
        want = O.first_fit_decreasing(lens.tolist(), cap)
        assert nb == len(want)
        for b, members in enumerate(want):
            assert all(bin_of[i] == b for i in members)


@pytest.mark.parametrize("name", ["epilogue_tok_pad.npz", "epilogue_tok_mixed.npz", "epilogue_odd_vocab.npz"])
def test_logprob_entropy_and_gradient(lib, name):
    d = load_npz(name)
    x = np.ascontiguousarray(d["logits"].float().numpy())
    t, v = x.shape
    ids = np.ascontiguousarray(d["input_ids"].numpy())
    temp = float(d["temperature"])
    lp, ent, lse = (np.zeros(t) for _ in range(3))
    lib.prl_oracle_logprob_entropy_f64(p(x, f32p), t, v, p(ids, i64p), temp, 1, p(lp, f64p), p(ent, f64p), p(lse, f64p))
    assert np.abs(lp - d["exp_logprobs"].numpy()).max() < 1e-5
    assert np.abs(ent - d["exp_entropy"].numpy()).max() < 1e-4  # the fp32 reference itself is only this close to double
    # gradient seed of the token-mode loss in closed form, then the logits gradient
    old, adv = d["old_logprobs"].double().numpy(), d["advantages"].double().numpy()
    mask = d["loss_mask"].numpy()
    ratio = np.exp(lp - old)
    g = np.where(mask & (ratio <= float(d["clip_ratio"])), -adv * ratio / max(d["loss_scale"], 1), 0.0)
    out = np.zeros((t, v))
    lib.prl_oracle_logprob_bwd_f64(p(x, f32p), t, v, p(ids, i64p), p(lse, f64p), p(g, f64p), temp, 1, p(out, f64p))
    want = d["exp_dlogits"].double().numpy()
    assert np.abs(out - want).max() < 1e-7 + 1e-4 * np.abs(want).max()


def test_group_advantage(lib):
    d = load_npz("advantages.npz")
    kinds = {"drgrpo": 0, "drgrpo-negclipped": 1, "rloo": 2, "opo": 3}
    for k in sorted(k[: -len("__adv")] for k in d if k.endswith("__adv")):
        r = np.ascontiguousarray(d[k + "__rewards"].numpy())
        n = np.ascontiguousarray(d[k + "__lengths"].numpy())
        out = np.zeros_like(r)
        g = int(d[k + "__g"])
        with np.errstate(all="ignore"):
            lib.prl_oracle_group_advantage(p(r, f64p), p(n, i64p), len(r), g, kinds[k.split("__")[1]], p(out, f64p))
        want = d[k + "__adv"].numpy()
        ok = (np.abs(out - want) < 1e-6) | (np.isnan(out) & np.isnan(want))
        assert ok.all(), k
