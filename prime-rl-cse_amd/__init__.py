"""MI355X-native policy-update hot path (HIP kernels behind a C ABI), exposed
under the names of the reference trainer's functions.

The directory name carries a hyphen; import it as `prime_rl_cse_amd` through the
loader module of that name at the repository root.
"""

from . import perf  # noqa: F401
from .advantage import compute_advantages, group_advantage  # noqa: F401
from .batch import Rollout, prepare_batch, prepare_sample, to_device  # noqa: F401
from .config import FakeDataLoaderConfig, LossConfig  # noqa: F401
from .data import DataLoader, FakeDataLoader, MicroBatch, save_rollout_batches  # noqa: F401
from .loss import (  # noqa: F401
    ShiftedLogits,
    compute_entropy,
    compute_loss,
    policy_loss,
    scored_positions,
    segments_from_position_ids,
    selective_log_softmax,
    shift_logits,
    unit_gradient,
)
from .world import get_world, reset_world  # noqa: F401
from .utils import Tensors, flexible_all_gather, get_response_lengths, masked_stats  # noqa: F401
