"""The two small schemas the hot path reads: how the loss is formed, and the shape of synthetic micro-batches.
Field names, bounds and defaults are the reference's (src/prime_rl/trainer/rl/config.py:19-43), because its TOML
recipes are loaded as they are."""

from __future__ import annotations

from typing import Literal

from pydantic import BaseModel, Field, model_validator


class LossConfig(BaseModel):
    """`ratio_type`: one importance ratio per token, or one per sample (its tokens' summed log-ratio);
    `ratio_length_norm`: divide that sum by the sample's number of scored tokens (sequence mode only);
    `clip_ratio`: upper clamp of the ratio, not below 0."""

    ratio_type: Literal["token", "sequence"] = "token"
    ratio_length_norm: bool = False
    clip_ratio: float = Field(default=8.0, ge=0)


class FakeDataLoaderConfig(BaseModel):
    """Synthetic rollouts: `batch_size` samples of `seq_len` tokens per step, `micro_batch_size` of them packed
    into a row (all at least 1; a step holds whole rows)."""

    micro_batch_size: int = Field(default=1, ge=1)
    batch_size: int = Field(default=2, ge=1)
    seq_len: int = Field(default=128, ge=1)

    @model_validator(mode="after")
    def _whole_rows(self) -> "FakeDataLoaderConfig":
        if self.batch_size < self.micro_batch_size or self.batch_size % self.micro_batch_size:
            raise ValueError(f"batch_size {self.batch_size} is not a positive multiple of micro_batch_size "
                             f"{self.micro_batch_size}")
        return self
