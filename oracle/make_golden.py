"""Pin the CPU oracle to the reference and write the golden vectors.

Run in the build container only (it reads /root/reference, which does not exist
on the GPU box):

    python oracle/make_golden.py            # checks oracle == reference, writes tests/golden/*.npz

What it does
  1. loads the reference's own loss, advantage, packing and sample-length code
     from the files where they lie, with identity stand-ins for the
     type-annotation packages the container lacks (jaxtyping, beartype) and for
     the two config modules whose import chain needs pydantic_settings;
  2. runs reference and oracle on the same seeded inputs and demands exact
     equality (floats compared with ==, NaNs allowed to match NaNs);
  3. stores inputs and the REFERENCE's outputs as small .npz fixtures.

The fixtures are data only; no reference source text is kept.
"""

from __future__ import annotations

import ast
import importlib.util
import sys
import types
from dataclasses import dataclass
from pathlib import Path

import numpy as np
import torch

HERE = Path(__file__).resolve().parent
REPO = HERE.parent
REF = Path("/root/reference/src/prime_rl")
OUT = REPO / "tests" / "golden"

sys.path.insert(0, str(REPO))
from oracle import ref_cpu as O  # noqa: E402


# --------------------------------------------------------------------------- loading the reference


def _install_stubs() -> None:
    class _Sub:
        def __class_getitem__(cls, item):
            return cls

    jt = types.ModuleType("jaxtyping")
    jt.Float = jt.Int = jt.Bool = _Sub
    jt.jaxtyped = lambda typechecker=None: (lambda f: f)
    sys.modules["jaxtyping"] = jt

    bt = types.ModuleType("beartype")
    bt.beartype = lambda f: f
    sys.modules["beartype"] = bt

    for name in ("prime_rl", "prime_rl.trainer", "prime_rl.trainer.rl", "prime_rl.orchestrator"):
        sys.modules.setdefault(name, types.ModuleType(name))

    cfg = types.ModuleType("prime_rl.trainer.rl.config")

    @dataclass
    class LossConfig:
        ratio_type: str = "token"
        ratio_length_norm: bool = False
        clip_ratio: float = 8.0

    cfg.LossConfig = LossConfig
    sys.modules["prime_rl.trainer.rl.config"] = cfg

    buf = types.ModuleType("prime_rl.orchestrator.buffer")
    buf.Rollout = O.Rollout
    sys.modules["prime_rl.orchestrator.buffer"] = buf

    data = types.ModuleType("prime_rl.trainer.rl.data")
    data.MicroBatch = dict
    sys.modules["prime_rl.trainer.rl.data"] = data

    try:
        import transformers.tokenization_utils  # noqa: F401
        from transformers.tokenization_utils import PreTrainedTokenizer  # noqa: F401
    except Exception:
        tu = types.ModuleType("transformers.tokenization_utils")
        tu.PreTrainedTokenizer = object
        sys.modules.setdefault("transformers", types.ModuleType("transformers"))
        sys.modules["transformers.tokenization_utils"] = tu


def _load(name: str, path: Path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def _load_function(path: Path, func: str):
    """Compile one top-level function of a reference file on its own (its
    module imports packages the container lacks)."""
    tree = ast.parse(path.read_text())
    node = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == func)
    ns = {"torch": torch}
    exec(compile(ast.Module(body=[node], type_ignores=[]), str(path), "exec"), ns)
    return ns[func]


def _load_compute_stats():
    """`Tensors.compute_stats` of the reference (src/prime_rl/trainer/utils.py:235-256) compiled on its own and
    bound to a plain defaultdict(list): the class's module needs packages the container lacks and its __init__
    wants a process group.  Two stand-ins, neither touching the arithmetic: `flexible_all_gather` returns its
    argument, which is what the reference's does on one rank (:203-204), and the device name the body moves the
    values to is "cpu" here (no GPU in the build container; the constant is swapped in the syntax tree)."""
    from collections import defaultdict

    path = REF / "trainer/utils.py"
    tree = ast.parse(path.read_text())
    cls = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "Tensors")
    fn = next(n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name == "compute_stats")

    class _Cpu(ast.NodeTransformer):
        def visit_Constant(self, node):
            return ast.copy_location(ast.Constant("cpu"), node) if node.value == "cuda" else node

    fn = ast.fix_missing_locations(_Cpu().visit(fn))
    ns = {"torch": torch, "flexible_all_gather": lambda t: t}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), str(path), "exec"), ns)

    def compute_stats(pieces: dict[str, list[torch.Tensor]]) -> dict[str, float]:
        store = defaultdict(list)
        for key, parts in pieces.items():
            store[key].extend(parts)
        return ns["compute_stats"](store)

    return compute_stats


def load_reference():
    _install_stubs()
    loss = _load("prime_rl.trainer.rl.loss", REF / "trainer/rl/loss.py")
    adv = _load("prime_rl.orchestrator.advantage", REF / "orchestrator/advantage.py")
    batch = _load("prime_rl.orchestrator.batch", REF / "orchestrator/batch.py")
    grl = _load_function(REF / "trainer/utils.py", "get_response_lengths")
    return loss, adv, batch, grl, sys.modules["prime_rl.trainer.rl.config"].LossConfig


# --------------------------------------------------------------------------- helpers


def same(a, b) -> bool:
    a = torch.as_tensor(a)
    b = torch.as_tensor(b)
    if a.shape != b.shape or a.dtype != b.dtype:
        return False
    if a.is_floating_point():
        return bool(((a == b) | (a.isnan() & b.isnan())).all())
    return bool((a == b).all())


def must(cond: bool, what: str) -> None:
    if not cond:
        raise SystemExit(f"ORACLE != REFERENCE: {what}")


def packed_row(gen: torch.Generator, lengths: list[int], pad: int = 0, vocab: int = 1000):
    """A packed row as the orchestrator lays it out: samples back to back,
    positions restarting at 0, optional trailing padding zeros."""
    pos = torch.cat([torch.arange(n) for n in lengths] + [torch.zeros(pad, dtype=torch.long)])
    t = pos.shape[0]
    ids = torch.randint(0, vocab, (t,), generator=gen)
    mask = torch.rand(t, generator=gen) < 0.6
    return ids, pos, mask


def reference_epilogue(R, logits, ids, pos, adv, old, mask, temperature, cfg, scale, grl):
    """The trainer's call order (src/prime_rl/trainer/rl/train.py:285-310) run
    with the reference's own functions."""
    leaf = logits.detach().clone().requires_grad_(True)
    x = leaf.float().contiguous()
    shifted = R.shift_logits(x)
    shifted = shifted / temperature
    lp = R.selective_log_softmax(shifted, ids)
    lengths = grl(pos)
    loss, aux = R.compute_loss(
        logprobs=lp.squeeze().split(lengths),
        old_logprobs=old.squeeze().split(lengths),
        advantages=adv.squeeze().split(lengths),
        loss_mask=mask.squeeze().split(lengths),
        loss_config=cfg,
        loss_scale=scale,
    )
    with torch.no_grad():
        ent = R.compute_entropy(shifted)
    if loss.requires_grad:
        loss.backward()
        grad = leaf.grad
    else:
        grad = torch.zeros_like(leaf)
    return {
        "logprobs": lp.detach(),
        "entropy": ent,
        "lengths": lengths,
        "loss": loss.detach(),
        "importance_ratio": aux["importance_ratio"].detach(),
        "clipped_importance_ratio": aux["clipped_importance_ratio"].detach(),
        "is_clipped": aux["is_clipped"].detach(),
        "dlogits": grad,
    }


def npify(d: dict) -> dict:
    out = {}
    for k, v in d.items():
        if isinstance(v, torch.Tensor):
            if v.dtype == torch.bfloat16:
                out[k + "__bf16"] = v.view(torch.int16).numpy()
            else:
                out[k] = v.numpy()
        elif isinstance(v, (list, tuple)):
            out[k] = np.asarray(v)
        else:
            out[k] = np.asarray(v)
    return out


# --------------------------------------------------------------------------- the vectors


def golden_epilogue(R, grl, LossConfig) -> int:
    """Whole-epilogue vectors: logits -> logprobs, entropy, loss, aux, dlogits."""
    cases = [
        # name, sample lengths, pad, V, temperature, logits dtype, cfg, clip-forcing
        ("tok_small", [3, 4], 0, 1000, 1.0, torch.float32, ("token", False, 8.0)),
        ("tok_mixed", [17, 9, 1, 22, 15], 0, 1000, 1.0, torch.float32, ("token", False, 8.0)),
        ("tok_pad", [30, 20], 14, 4096, 0.6, torch.float32, ("token", False, 8.0)),
        ("tok_clip2", [16, 16, 16, 16], 0, 4096, 1.0, torch.float32, ("token", False, 2.0)),
        ("tok_bf16", [40, 56, 32], 0, 4096, 0.6, torch.bfloat16, ("token", False, 8.0)),
        ("tok_t512", [128] * 4, 0, 1000, 1.0, torch.bfloat16, ("token", False, 8.0)),
        ("seq_plain", [12, 20, 32], 0, 1000, 1.0, torch.float32, ("sequence", False, 8.0)),
        ("seq_norm", [12, 20, 32], 0, 1000, 0.6, torch.float32, ("sequence", True, 8.0)),
        ("seq_cap10", [40, 24], 0, 1000, 1.0, torch.float32, ("sequence", False, 1e9)),
        ("seq_bf16", [50, 78, 128], 0, 1000, 1.0, torch.bfloat16, ("sequence", True, 3.0)),
        ("one_sample", [7], 0, 1000, 1.0, torch.float32, ("token", False, 8.0)),
        ("odd_vocab", [9, 11], 3, 1003, 0.7, torch.float32, ("token", False, 8.0)),
    ]
    n = 0
    for i, (name, lengths, pad, vocab, temp, dt, (rtype, lnorm, clip)) in enumerate(cases):
        gen = torch.Generator().manual_seed(1000 + i)
        ids, pos, mask = packed_row(gen, lengths, pad, vocab)
        t = ids.shape[0]
        logits = (torch.randn(1, t, vocab, generator=gen) * 3.0).to(dt)
        adv = torch.randn(t, generator=gen)
        # old logprobs close to the new ones so that ratios are near 1, with a
        # few pushed far out so that the clip (and the log-10 cap) fire
        with torch.no_grad():
            lp0 = O.selective_log_softmax(O.shift_logits(logits.float()) / temp, ids.unsqueeze(0)).squeeze(0)
        old = lp0 + 0.3 * torch.randn(t, generator=gen)
        far = torch.rand(t, generator=gen) < 0.1
        old = torch.where(far, lp0 - 3.0, old)
        if name == "seq_cap10":
            old = lp0 - 1.0  # masked sum of log ratios far above 10
        if name == "tok_mixed":
            mask[17:26] = False  # a sample with nothing to learn from
        cfg = LossConfig(ratio_type=rtype, ratio_length_norm=lnorm, clip_ratio=clip)
        scale = int(mask.sum().item()) if rtype == "token" else len(lengths)
        b = lambda x: x.unsqueeze(0)
        ref = reference_epilogue(R, logits, b(ids), b(pos), b(adv), b(old), b(mask), temp, cfg, scale, grl)
        ocl = O.policy_epilogue(
            logits, b(ids), b(pos), b(adv), b(old), b(mask), temp, O.LossConfig(rtype, lnorm, clip), scale
        )
        for k in ref:
            if k == "lengths":
                must(ref[k] == ocl[k], f"{name}.{k}")
            else:
                must(same(ref[k], ocl[k]), f"{name}.{k}")
        np.savez_compressed(
            OUT / f"epilogue_{name}.npz",
            **npify(
                {
                    "logits": logits[0],
                    "input_ids": ids,
                    "position_ids": pos,
                    "advantages": adv,
                    "old_logprobs": old,
                    "loss_mask": mask,
                    "temperature": temp,
                    "ratio_type": rtype,
                    "ratio_length_norm": lnorm,
                    "clip_ratio": clip,
                    "loss_scale": scale,
                    "exp_logprobs": ref["logprobs"][0],
                    "exp_entropy": ref["entropy"][0],
                    "exp_lengths": ref["lengths"],
                    "exp_loss": ref["loss"],
                    "exp_importance_ratio": ref["importance_ratio"],
                    "exp_clipped_importance_ratio": ref["clipped_importance_ratio"],
                    "exp_is_clipped": ref["is_clipped"],
                    "exp_dlogits": ref["dlogits"][0],
                }
            ),
        )
        n += 1
    return n


def golden_full_vocabulary(R, grl, LossConfig) -> int:
    """Whole-epilogue vectors at the vocabularies of the recipes' models (151936: Qwen3 / Qwen2.5 small;
    152064: the 7B and 32B distillations) and the sampling temperature of the long-context recipe (0.6), bf16
    logits as the model emits them.  Few rows, so that a case stays a few MB.  Next to the reference's fp32
    entropy the file holds a float64 evaluation of the same formula: at this size the reference's own fp32 sums
    are a few 1e-4 away from it, which is what a test needs to know to judge a third implementation."""
    n = 0
    for i, (name, vocab, lengths) in enumerate([("v151936", 151936, [3, 5]), ("v152064", 152064, [6, 2])]):
        gen = torch.Generator().manual_seed(7000 + i)
        temp = 0.6
        t = sum(lengths)
        pos = torch.cat([torch.arange(k) for k in lengths])
        ids = torch.randint(0, vocab, (t,), generator=gen)
        ids[-1] = vocab - 1  # the last column of the matrix is somebody's label
        mask = torch.ones(t, dtype=torch.bool)
        mask[0] = mask[4] = False
        logits = (torch.randn(1, t, vocab, generator=gen) * 2.5).to(torch.bfloat16)
        adv = torch.randn(t, generator=gen)
        with torch.no_grad():
            x = O.shift_logits(logits.float()) / temp
            lp0 = O.selective_log_softmax(x, ids.unsqueeze(0)).squeeze(0)
            x64 = x.double()
            ent64 = (torch.logsumexp(x64, -1) - (torch.softmax(x64, -1) * x64).sum(-1))[0]
            lp64 = (x64.gather(-1, ids.view(1, t, 1)).squeeze(-1) - torch.logsumexp(x64, -1))[0]
        old = lp0 + 0.2 * torch.randn(t, generator=gen)
        cfg = LossConfig(ratio_type="token", ratio_length_norm=False, clip_ratio=8.0)
        scale = int(mask.sum().item())
        b = lambda v: v.unsqueeze(0)
        ref = reference_epilogue(R, logits, b(ids), b(pos), b(adv), b(old), b(mask), temp, cfg, scale, grl)
        ocl = O.policy_epilogue(logits, b(ids), b(pos), b(adv), b(old), b(mask), temp,
                                O.LossConfig("token", False, 8.0), scale)
        for k in ref:
            must(ref[k] == ocl[k] if k == "lengths" else same(ref[k], ocl[k]), f"full_{name}.{k}")
        print(f"  full_{name}: reference fp32 vs float64: entropy {(ref['entropy'][0].double() - ent64).abs().max():.2e}, "
              f"logprob {(ref['logprobs'][0].double() - lp64).abs().max():.2e}")
        np.savez_compressed(
            OUT / f"full_{name}.npz",
            **npify({
                "logits": logits[0], "input_ids": ids, "position_ids": pos, "advantages": adv, "old_logprobs": old,
                "loss_mask": mask, "temperature": temp, "ratio_type": "token", "ratio_length_norm": False,
                "clip_ratio": 8.0, "loss_scale": scale,
                "exp_logprobs": ref["logprobs"][0], "exp_entropy": ref["entropy"][0], "exp_lengths": ref["lengths"],
                "exp_loss": ref["loss"], "exp_importance_ratio": ref["importance_ratio"],
                "exp_clipped_importance_ratio": ref["clipped_importance_ratio"], "exp_is_clipped": ref["is_clipped"],
                "exp_dlogits": ref["dlogits"][0],
                "f64_entropy": ent64, "f64_logprobs": lp64,
            }),
        )
        n += 1
    return n


def golden_pieces(R, LossConfig) -> int:
    """Function-level vectors: shift, selective_log_softmax (fp32 and bf16
    branches), entropy and compute_loss on its own inputs."""
    n = 0
    for i, (t, v, temp) in enumerate([(7, 1000, 1.0), (64, 4096, 0.6), (512, 1000, 1.0), (5, 37, 1.3)]):
        gen = torch.Generator().manual_seed(2000 + i)
        x = torch.randn(1, t, v, generator=gen) * 4.0
        idx = torch.randint(0, v, (1, t), generator=gen)
        sh_r, sh_o = R.shift_logits(x), O.shift_logits(x)
        must(same(sh_r, sh_o), f"shift {t}x{v}")
        xs = sh_r / temp
        lp_r, lp_o = R.selective_log_softmax(xs, idx), O.selective_log_softmax(xs, idx)
        must(same(lp_r, lp_o), f"lsm f32 {t}x{v}")
        xb = xs.to(torch.bfloat16)
        lpb_r, lpb_o = R.selective_log_softmax(xb, idx), O.selective_log_softmax(xb, idx)
        must(same(lpb_r, lpb_o), f"lsm bf16 {t}x{v}")
        en_r, en_o = R.compute_entropy(xs), O.compute_entropy(xs)
        must(same(en_r, en_o), f"entropy {t}x{v}")
        np.savez_compressed(
            OUT / f"pieces_{t}x{v}.npz",
            **npify(
                {
                    "logits": x[0],
                    "index": idx[0],
                    "temperature": temp,
                    "exp_logprobs": lp_r[0],
                    "exp_logprobs_bf16in": lpb_r[0].float(),
                    "exp_entropy": en_r[0],
                }
            ),
        )
        n += 1

    # compute_loss alone, on arbitrary (not log-softmax derived) inputs, as the
    # reference's own shape tests feed it (tests/unit/train/rl/test_loss.py:10-42)
    for i, (lengths, rtype, lnorm, clip) in enumerate(
        [
            ([50, 30], "token", False, 10.0),
            ([40, 60], "sequence", False, 10.0),
            ([40, 60], "sequence", True, 10.0),
            ([5, 1, 1, 9], "token", False, 0.5),
            ([3], "sequence", True, 8.0),
        ]
    ):
        gen = torch.Generator().manual_seed(3000 + i)
        lp = [torch.randn(k, generator=gen) for k in lengths]
        old = [torch.randn(k, generator=gen) for k in lengths]
        adv = [torch.randn(k, generator=gen) for k in lengths]
        msk = [torch.rand(k, generator=gen) < 0.7 for k in lengths]
        if i == 3:
            msk[1] = torch.zeros(1, dtype=torch.bool)
        lpg = [x.clone().requires_grad_(True) for x in lp]
        l_r, a_r = R.compute_loss(lpg, old, adv, msk, LossConfig(rtype, lnorm, clip), 7.0)
        l_r.backward()
        g_r = torch.cat([x.grad if x.grad is not None else torch.zeros_like(x) for x in lpg])
        lpo = [x.clone().requires_grad_(True) for x in lp]
        l_o, a_o = O.compute_loss(lpo, old, adv, msk, O.LossConfig(rtype, lnorm, clip), 7.0)
        l_o.backward()
        g_o = torch.cat([x.grad if x.grad is not None else torch.zeros_like(x) for x in lpo])
        must(same(l_r.detach(), l_o.detach()), f"loss {i}")
        must(same(g_r, g_o), f"loss grad {i}")
        for k in a_r:
            must(same(a_r[k].detach(), a_o[k].detach()), f"loss {i} {k}")
        np.savez_compressed(
            OUT / f"loss_{i}_{rtype}.npz",
            **npify(
                {
                    "lengths": lengths,
                    "logprobs": torch.cat(lp),
                    "old_logprobs": torch.cat(old),
                    "advantages": torch.cat(adv),
                    "loss_mask": torch.cat(msk),
                    "ratio_type": rtype,
                    "ratio_length_norm": lnorm,
                    "clip_ratio": clip,
                    "loss_scale": 7.0,
                    "exp_loss": l_r.detach(),
                    "exp_grad_logprobs": g_r,
                    "exp_importance_ratio": a_r["importance_ratio"].detach(),
                    "exp_clipped_importance_ratio": a_r["clipped_importance_ratio"].detach(),
                    "exp_is_clipped": a_r["is_clipped"].detach(),
                }
            ),
        )
        n += 1
    return n


def golden_lengths(grl) -> int:
    rows = {
        "all_zero": [0] * 12,
        "one_seq": list(range(9)),
        "many": list(range(5)) + list(range(3)) + list(range(7)),
        "trailing_pad": list(range(6)) + list(range(4)) + [0, 0, 0],
        "single_token_mid": list(range(4)) + [0] + list(range(5)),
        "zero_zero_one": [0, 0, 1, 2, 0, 1],
        "starts_with_pad": [0, 0, 0, 1, 2, 0, 1, 2],
        "boundary_at_end": list(range(5)) + [0],
        "boundary_then_one_last": list(range(5)) + [0, 1],
        "empty": [],  # boundaries = [0] and nothing else: ONE sample of length 0
        "len1": [0],
        "len2": [0, 1],
        "two_tokens_each": [0, 1, 0, 1, 0, 1],
        "nonmonotone": [0, 1, 2, 0, 2, 0, 1, 5, 0, 1],
    }
    gen = torch.Generator().manual_seed(4000)
    for k in range(40):
        t = int(torch.randint(1, 80, (1,), generator=gen))
        # random rows over {0,1,2,3} hit every branch of the rule densely
        rows[f"rand{k}"] = torch.randint(0, 4, (t,), generator=gen).tolist()
    for k in range(10):
        lens = torch.randint(1, 40, (int(torch.randint(1, 9, (1,), generator=gen)),), generator=gen).tolist()
        pad = int(torch.randint(0, 6, (1,), generator=gen))
        rows[f"packed{k}"] = [p for n in lens for p in range(n)] + [0] * pad
    store = {}
    for name, row in rows.items():
        pos = torch.tensor(row, dtype=torch.long)
        r = grl(pos.unsqueeze(0))
        o = O.get_response_lengths(pos.unsqueeze(0))
        must(r == o, f"lengths {name}: {r} vs {o}")
        store[f"{name}__pos"] = pos.numpy()
        store[f"{name}__len"] = np.asarray(r, dtype=np.int64)
    np.savez_compressed(OUT / "response_lengths.npz", **store)
    return len(rows)


def golden_tensor_stats(compute_stats) -> int:
    """Logged statistics: the reference's `Tensors.compute_stats` on one rank against the oracle's `tensor_stats`
    over the same values (a key's pieces concatenated)."""
    gen = torch.Generator().manual_seed(7000)
    cases: dict[str, list[torch.Tensor]] = {
        "one_value": [torch.tensor([0.25])],  # std of one value is NaN
        "two_values": [torch.tensor([3.0, -1.0])],  # even count: the LOWER middle value is the median
        "ties": [torch.tensor([1.0, 1.0, 2.0, 2.0, 2.0, 1.0])],
        "pieces": [torch.randn(n, generator=gen) for n in (5, 1, 300, 77)],
        "probabilities": [torch.rand(2048, generator=gen), torch.rand(1999, generator=gen)],
        "wide_range": [torch.randn(4097, generator=gen) * 1e4, torch.randn(3, generator=gen) * 1e-4],
        "clipped_flags": [(torch.rand(1000, generator=gen) < 0.1).float()],
        "with_inf": [torch.tensor([1.0, float("inf"), -2.0, 0.5])],
        "negative_zero": [torch.tensor([-0.0, 0.0, -0.0])],
    }
    want = compute_stats({k: [p.clone() for p in v] for k, v in cases.items()})
    store = {}
    for name, parts in cases.items():
        got = O.tensor_stats(torch.cat(parts))
        for stat in ("mean", "median", "std", "min", "max"):
            r, o = want[f"{name}/{stat}"], got[stat]
            must(r == o or (r != r and o != o), f"tensor stats {name}/{stat}: {r} vs {o}")
        store[f"{name}__sizes"] = np.asarray([p.numel() for p in parts], dtype=np.int64)
        store[f"{name}__x"] = torch.cat(parts).numpy()
        store[f"{name}__stats"] = np.asarray([want[f"{name}/{stat}"] for stat in ("mean", "median", "std", "min", "max")],
                                             dtype=np.float64)
    np.savez_compressed(OUT / "tensor_stats.npz", **store)
    return len(cases)


def golden_advantages(A) -> int:
    store = {}
    n = 0
    gen = torch.Generator().manual_seed(5000)
    sets = [
        ("doc_a", [0.0, 0.0, 1.0, 1.0], [3, 4, 5, 6], 4),
        ("doc_b", [0.0] * 4, [1, 1, 1, 1], 4),
        ("doc_c", [1.0] * 4, [2, 9, 4, 4], 4),
        ("g1", [0.3, 0.9, -1.0], [5, 6, 7], 1),
        ("g16", torch.rand(128, generator=gen).tolist(), torch.randint(1, 128, (128,), generator=gen).tolist(), 16),
        ("g8", torch.randn(64, generator=gen).tolist(), torch.randint(1, 8192, (64,), generator=gen).tolist(), 8),
        ("g3", torch.randn(12, generator=gen).tolist(), torch.randint(1, 50, (12,), generator=gen).tolist(), 3),
    ]
    for name, rewards, lengths, g in sets:
        for kind in ("drgrpo", "drgrpo-negclipped", "rloo", "opo"):
            r = A.compute_advantages(rewards, lengths, g, kind)
            o = O.compute_advantages(rewards, lengths, g, kind)
            must(same(torch.tensor(r), torch.tensor(o)), f"adv {name} {kind}")
            key = f"{name}__{kind}"
            store[key + "__rewards"] = np.asarray(rewards, dtype=np.float64)
            store[key + "__lengths"] = np.asarray(lengths, dtype=np.int64)
            store[key + "__g"] = np.asarray(g)
            store[key + "__adv"] = np.asarray(r, dtype=np.float64)
            n += 1
    np.savez_compressed(OUT / "advantages.npz", **store)
    return n


def golden_batches(B) -> int:
    """Packing vectors: rank count 1, 2, 4, including the case where the row
    count already divides the rank count (a whole extra set of fillers)."""
    store = {}
    n = 0
    for case, (n_roll, mbs, seq_len, seed) in enumerate([(12, 2, 32, 0), (16, 4, 16, 1), (8, 1, 64, 2), (5, 3, 24, 3)]):
        gen = torch.Generator().manual_seed(6000 + seed)
        rollouts = []
        for pid in range(n_roll):
            p = int(torch.randint(1, seq_len // 2, (1,), generator=gen))
            c = int(torch.randint(1, seq_len - p + 1, (1,), generator=gen))
            rollouts.append(
                O.Rollout(
                    problem_id=pid,
                    prompt_tokens=torch.randint(0, 1000, (p,), generator=gen).tolist(),
                    prompt_mask=[0] * p,
                    completion_tokens=torch.randint(0, 1000, (c,), generator=gen).tolist(),
                    completion_mask=(torch.rand(c, generator=gen) < 0.9).long().tolist(),
                    completion_logprobs=(-torch.rand(c, generator=gen)).tolist(),
                    is_truncated=False,
                    reward=float(torch.rand(1, generator=gen)),
                    advantage=float(torch.randn(1, generator=gen)),
                )
            )
        store[f"c{case}__meta"] = np.asarray([n_roll, mbs, seq_len])
        for j, r in enumerate(rollouts):
            store[f"c{case}__r{j}__prompt"] = np.asarray(r.prompt_tokens, dtype=np.int64)
            store[f"c{case}__r{j}__completion"] = np.asarray(r.completion_tokens, dtype=np.int64)
            store[f"c{case}__r{j}__cmask"] = np.asarray(r.completion_mask, dtype=np.int64)
            store[f"c{case}__r{j}__clogp"] = np.asarray(r.completion_logprobs, dtype=np.float64)
            store[f"c{case}__r{j}__adv"] = np.asarray(r.advantage, dtype=np.float64)
        for w in (1, 2, 4):
            ref = B.prepare_batch(rollouts, 0.7, None, n_roll, mbs, seq_len, w)
            ocl = O.prepare_batch(rollouts, 0.7, n_roll, mbs, seq_len, w)
            must(len(ref) == len(ocl) == w, f"batch c{case} w{w} ranks")
            for rank in range(w):
                must(len(ref[rank]) == len(ocl[rank]), f"batch c{case} w{w} rank{rank} rows")
                store[f"c{case}__w{w}__rank{rank}__rows"] = np.asarray(len(ref[rank]))
                for m, (a, b) in enumerate(zip(ref[rank], ocl[rank])):
                    must(a["temperature"] == b["temperature"], "temperature")
                    for k in O.TOKEN_KEYS:
                        must(same(a[k], b[k]), f"batch c{case} w{w} rank{rank} mb{m} {k}")
                        store[f"c{case}__w{w}__rank{rank}__mb{m}__{k}"] = a[k].numpy()
            n += 1
    np.savez_compressed(OUT / "batches.npz", **store)
    return n


def main() -> None:
    if not REF.exists():
        raise SystemExit("the reference is not mounted here; golden vectors are made in the build container only")
    OUT.mkdir(parents=True, exist_ok=True)
    torch.set_num_threads(1)  # one summation order, whatever machine regenerates these
    R, A, B, grl, LossConfig = load_reference()
    must(A.compute_advantages([0.0, 0.0, 1.0, 1.0], [3, 4, 5, 6], 4, "drgrpo") == [-0.5, -0.5, 0.5, 0.5], "docstring example")
    counts = {
        "epilogue": golden_epilogue(R, grl, LossConfig),
        "full_vocabulary": golden_full_vocabulary(R, grl, LossConfig),
        "pieces": golden_pieces(R, LossConfig),
        "lengths": golden_lengths(grl),
        "advantages": golden_advantages(A),
        "batches": golden_batches(B),
        "tensor_stats": golden_tensor_stats(_load_compute_stats()),
    }
    print("oracle == reference on every case; fixtures written:", counts)


if __name__ == "__main__":
    main()
