"""Shared helpers for the parity tests."""

from __future__ import annotations

from pathlib import Path

import numpy as np
import torch

GOLDEN = Path(__file__).resolve().parent / "golden"


def load_npz(name: str) -> dict:
    """A fixture as a dict of torch tensors / Python scalars.  Arrays stored as
    `<key>__bf16` hold the raw bits of bfloat16 values."""
    out = {}
    with np.load(GOLDEN / name, allow_pickle=False) as z:
        for k in z.files:
            a = z[k]
            if k.endswith("__bf16"):
                out[k[: -len("__bf16")]] = torch.from_numpy(a.copy()).view(torch.bfloat16)
            elif a.ndim == 0:
                out[k] = a.item()
            else:
                out[k] = torch.from_numpy(a.copy())
    return out


def epilogue_cases() -> list[str]:
    return sorted(p.name for p in GOLDEN.glob("epilogue_*.npz"))


def full_vocabulary_cases() -> list[str]:
    """Whole-epilogue vectors at V = 151936 / 152064, temperature 0.6, bf16 logits (a few MB each)."""
    return sorted(p.name for p in GOLDEN.glob("full_*.npz"))


def make_packed_row(gen: torch.Generator, lengths, pad=0, vocab=1000, mask_p=0.6):
    pos = torch.cat([torch.arange(n) for n in lengths] + [torch.zeros(pad, dtype=torch.long)])
    t = pos.shape[0]
    ids = torch.randint(0, vocab, (t,), generator=gen)
    mask = torch.rand(t, generator=gen) < mask_p
    return ids, pos, mask


def bf16_ulp_close(a: torch.Tensor, b: torch.Tensor, ulps: int = 1, atol: float = 1e-30) -> bool:
    """Two bf16 tensors equal up to `ulps` units in the last place."""
    af, bf = a.float(), b.float()
    tol = ulps * torch.maximum(af.abs(), bf.abs()) * 2.0**-7 + atol
    return bool(((af - bf).abs() <= tol).all())
