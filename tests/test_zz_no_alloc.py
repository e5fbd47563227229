"""`prl_logprob_entropy_fwd_static`, the entry point that allocates nothing (its partial soft-max states live in
a fixed device array of the library), against `prl_logprob_entropy_fwd` with a workspace of the caller's: the
same kernels, so the three outputs must agree bit for bit.  Two calls in a row, the second with a smaller
input, so that the second finds the first call's leftovers in the array."""

import pytest
import torch

from prime_rl_cse_amd import _capi as C

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _both(lib, logits, ids, temperature, shift):
    t, v = logits.shape
    outs = []
    for static in (False, True):
        lp, ent, lse = (torch.full((t,), float("nan"), device=DEV) for _ in range(3))
        if static:
            rc = lib.prl_logprob_entropy_fwd_static(C.ptr(logits), C.dtype_code(logits), t, v, logits.stride(0),
                                                    C.ptr(ids), temperature, shift, C.ptr(lp), C.ptr(ent), C.ptr(lse),
                                                    C.stream_ptr(torch.device(DEV)))
        else:
            need = int(lib.prl_logprob_workspace_bytes(t, v))
            ws = torch.empty(max(need, 16), dtype=torch.uint8, device=DEV)
            rc = lib.prl_logprob_entropy_fwd(C.ptr(logits), C.dtype_code(logits), t, v, logits.stride(0), C.ptr(ids),
                                             temperature, shift, C.ptr(lp), C.ptr(ent), C.ptr(lse), C.ptr(ws), need,
                                             C.stream_ptr(torch.device(DEV)))
        C.check(rc, "logprob_entropy_fwd" + ("_static" if static else ""))
        torch.cuda.synchronize()
        outs.append((lp, ent, lse))
    return outs


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_static_scratch_entry_point_matches_the_workspace_one_twice_in_a_row(dtype):
    lib = C.load()
    assert int(lib.prl_logprob_static_scratch_bytes()) == 49152
    gen = torch.Generator().manual_seed(77)
    vocab = 151936 if dtype == torch.bfloat16 else 40000
    # rows split over many workgroups (scratch in use), then fewer rows, then enough rows that none is split
    for t, temperature, shift in ((96, 0.6, 1), (7, 1.0, 1), (33, 1.3, 0), (1100, 1.0, 1)):
        v = vocab if t < 1000 else 8200
        logits = (torch.randn(t, v, generator=gen) * 2.5).to(dtype).to(DEV)
        ids = torch.randint(0, v, (t,), generator=gen).to(DEV)
        need = int(lib.prl_logprob_workspace_bytes(t, v))
        assert (need > 0) == (t < 1000) and need <= 49152
        (lp0, ent0, lse0), (lp1, ent1, lse1) = _both(lib, logits, ids, temperature, shift)
        assert not lp1.isnan().any() and not ent1.isnan().any() and not lse1.isnan().any()
        assert torch.equal(lp0, lp1) and torch.equal(ent0, ent1) and torch.equal(lse0, lse1), (t, v)
        # and the numbers are the right ones, not merely the same ones
        x = logits.float() / temperature
        if shift:
            x = torch.cat([torch.zeros(1, v, device=DEV), x[:-1]])
        want_lse = torch.logsumexp(x.double(), -1)
        want_lp = x.double().gather(-1, ids[:, None])[:, 0] - want_lse
        assert (lse1.double() - want_lse).abs().max().item() < 1e-4
        assert (lp1.double() - want_lp).abs().max().item() < 1e-4


def test_static_scratch_entry_point_takes_no_device_memory():
    """Free device memory as the HIP runtime reports it (hipMemGetInfo, which sees every allocation of the process,
    the library's included, not only torch's) does not move over 100 calls: rows split over workgroups (the static
    scratch in use), rows that are not split (T >= 1024), and calls that are refused.  Every tensor exists before
    the first reading, and one call has been made on the device (it looks the array's address up)."""
    lib = C.load()
    dev = torch.device(DEV)
    gen = torch.Generator().manual_seed(3)
    cases = []
    for t, v in ((96, 151936), (7, 40000), (1100, 8200)):
        logits = (torch.randn(t, v, generator=gen) * 2.5).to(torch.bfloat16).to(DEV)
        ids = torch.randint(0, v, (t,), generator=gen).to(DEV)
        outs = [torch.empty(t, device=DEV) for _ in range(3)]
        cases.append((logits, ids, outs))
    stream = C.stream_ptr(dev)

    def call(logits, ids, outs, dtype_code=None, temperature=1.0):
        t, v = logits.shape
        return lib.prl_logprob_entropy_fwd_static(C.ptr(logits), C.dtype_code(logits) if dtype_code is None else
                                                  dtype_code, t, v, logits.stride(0), C.ptr(ids), temperature, 1,
                                                  C.ptr(outs[0]), C.ptr(outs[1]), C.ptr(outs[2]), stream)

    for c in cases:  # warm-up: code objects loaded, the array's address looked up
        assert call(*c) == 0
    torch.cuda.synchronize()
    free_before, _ = torch.cuda.mem_get_info(dev)
    for i in range(100):
        assert call(*cases[i % 3]) == 0
        if i % 10 == 0:
            assert call(*cases[0], dtype_code=7) == -3  # refused: nothing launched
            assert call(*cases[1], temperature=-1.0) == -1
    torch.cuda.synchronize()
    free_after, _ = torch.cuda.mem_get_info(dev)
    assert free_after >= free_before, (free_before, free_after)
    for _, _, outs in cases:
        assert not any(o.isnan().any() for o in outs)


def test_static_scratch_entry_point_refuses_what_the_other_refuses():
    lib = C.load()
    x = torch.zeros(4, 64, device=DEV)
    ids = torch.zeros(4, dtype=torch.int64, device=DEV)
    out = torch.zeros(4, device=DEV)
    assert lib.prl_logprob_entropy_fwd_static(C.ptr(x), 7, 4, 64, 64, C.ptr(ids), 1.0, 1, C.ptr(out), None, None,
                                              None) == -3
    assert lib.prl_logprob_entropy_fwd_static(C.ptr(x), 0, 4, 64, 64, C.ptr(ids), -1.0, 1, C.ptr(out), None, None,
                                              None) == -1
    assert b"temperature" in lib.prl_last_error()
