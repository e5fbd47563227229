"""The reference's own unit tests of its loss module (tests/unit/train/rl/test_loss.py), call for call, against this
package's callables of the same names; each also holds the value to the CPU oracle (1e-4, as the other parity tests)."""

import pytest
import torch

from oracle import ref_cpu as O
from prime_rl_cse_amd import LossConfig, compute_entropy, compute_loss

pytestmark = [pytest.mark.gpu]


def _lists(lengths, seed):
    g = torch.Generator().manual_seed(seed)
    logprobs = [torch.randn(n, dtype=torch.float32, generator=g) for n in lengths]
    old_logprobs = [torch.randn(n, dtype=torch.float32, generator=g) for n in lengths]
    advantages = [torch.randn(n, generator=g) for n in lengths]
    loss_mask = [torch.ones(n, dtype=torch.bool) for n in lengths]
    return logprobs, old_logprobs, advantages, loss_mask


def _both(lengths, seed, config):
    host = _lists(lengths, seed)
    device = [[t.cuda() for t in part] for part in host]
    loss, tensors = compute_loss(*device, loss_config=config, loss_scale=1.0)
    want, want_tensors = O.compute_loss(*host, config, 1.0)
    return loss, tensors, want, want_tensors


def test_grpo_loss():  # tests/unit/train/rl/test_loss.py:10-24
    loss, tensors, want, want_tensors = _both([50, 30], 1, LossConfig(ratio_type="token", clip_ratio=10.0))
    assert loss.shape == ()
    assert abs(loss.item() - float(want)) <= 1e-4 * max(1.0, abs(float(want)))
    assert set(tensors) == set(want_tensors)


def test_gspo_loss():  # tests/unit/train/rl/test_loss.py:27-42
    loss, tensors, want, want_tensors = _both([40, 60], 2, LossConfig(ratio_type="sequence", clip_ratio=10.0))
    assert loss.shape == ()
    assert abs(loss.item() - float(want)) <= 1e-4 * max(1.0, abs(float(want)))
    assert set(tensors) == set(want_tensors)


def test_entropy_loss():  # tests/unit/train/rl/test_loss.py:45-48
    shifted_logits = torch.randn(10, 10, 10, dtype=torch.float32, generator=torch.Generator().manual_seed(3))
    entropy = compute_entropy(shifted_logits.cuda())
    assert entropy.shape == (10, 10)
    assert torch.allclose(entropy.cpu(), O.compute_entropy(shifted_logits), atol=1e-4, rtol=0)
