"""Loss configuration with the reference's field names and defaults
(src/prime_rl/trainer/rl/config.py:19-27)."""

from __future__ import annotations

from typing import Literal

from pydantic import BaseModel, Field
from typing_extensions import Annotated


class LossConfig(BaseModel):
    """Base config for loss."""

    ratio_type: Annotated[Literal["token", "sequence"], Field(description="Type of importance ratio to use.")] = "token"
    ratio_length_norm: Annotated[
        bool, Field(description="Whether to normalize the importance ratio by the sequence length.")
    ] = False
    clip_ratio: Annotated[float, Field(ge=0)] = 8.0


class FakeDataLoaderConfig(BaseModel):
    """Synthetic micro-batches for debugging and benchmarks
    (src/prime_rl/trainer/rl/config.py:30-43)."""

    micro_batch_size: Annotated[int, Field(ge=1)] = 1
    batch_size: Annotated[int, Field(ge=1)] = 2
    seq_len: Annotated[int, Field(ge=1)] = 128

    def model_post_init(self, _ctx) -> None:
        if self.batch_size % self.micro_batch_size != 0:
            raise ValueError("Batch size must be divisible by micro batch size")
        if self.batch_size < self.micro_batch_size:
            raise ValueError("Batch size must be greater than or equal to micro batch size")
