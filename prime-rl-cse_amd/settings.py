"""The RL trainer's configuration surface: the reference's field names, defaults
and TOML / CLI conventions, so its `configs/**/rl/train.toml` recipes load
unchanged (counterparts: src/prime_rl/trainer/config.py:15-379,
src/prime_rl/trainer/rl/config.py:19-191, src/prime_rl/utils/config.py:21-98,
src/prime_rl/utils/pydantic_config.py:18-280).

The reference builds on pydantic-settings and tomli_w; neither is needed here:
TOML is read with tomli, sources are merged by this module and the result is
validated by plain pydantic models.

Conventions kept
  * `@ file.toml` on the command line loads a file; `--key @ file.toml` loads it
    under `key` (dotted keys nest);
  * a file may name others in `toml_files` (paths relative to it); those take
    precedence over the file that names them;
  * `--a.b value`, `--a.b=value`, `--flag` / `--no-flag` for booleans, snake or
    kebab case; unknown keys are errors (`extra="forbid"`);
  * environment variables `PRIME_A__B=value`;
  * the string "None" means None;
  * precedence: command line > TOML files > environment > defaults.
"""

from __future__ import annotations

import json
import os
import sys
from pathlib import Path
from typing import Annotated, Any, Literal, Type, TypeVar, Union

import tomli
from pydantic import BaseModel, ConfigDict, Field, field_validator, model_validator

from .config import FakeDataLoaderConfig, LossConfig


class BaseConfig(BaseModel):
    model_config = ConfigDict(extra="forbid")

    @field_validator("*", mode="before")
    @classmethod
    def _none_string(cls, v):
        return None if v == "None" else v


class _Strict(BaseModel):
    """Unknown keys are errors; no wildcard validator, so the class can sit in a discriminated union."""

    model_config = ConfigDict(extra="forbid")


# --------------------------------------------------------------------------- model


class ActivationCheckpointConfig(BaseConfig):
    freq: Annotated[int, Field(ge=1)] = 1


class CompileConfig(BaseConfig):
    fullgraph: bool = False


class DebugModelConfig(BaseConfig):
    num_layers: int | None = None
    random_init: bool = False


class LoRAConfig(BaseConfig):
    rank: Annotated[int, Field(ge=1)] = 16
    alpha: Annotated[float, Field(ge=0)] = 16.0
    dropout: Annotated[float, Field(ge=0, le=1)] = 0.0
    target_modules: list[str] = [r".*\.q_proj$", r".*\.k_proj$", r".*\.v_proj$", r".*\.o_proj$", r".*\.gate_proj$",
                                 r".*\.up_proj$", r".*\.down_proj$"]
    modules_to_save: list[str] = [r".*embed_tokens$", r".*norm$", r".*layernorm$", r"lm_head$"]


class ExperimentalConfig(BaseConfig):
    lora: LoRAConfig | None = None


MOE_MODEL_MAPS = {
    "Qwen/Qwen3-30B-A3B": "Jackmin108/Qwen3-30B-A3B-Fast",
    "moonshotai/Moonlight-16B-A3B-Instruct": "Jackmin108/Moonlight-16B-A3B-Instruct-Fast",
}


class ModelConfig(BaseConfig):
    name: str = "Qwen/Qwen3-0.6B"
    attn: Literal["sdpa", "flash_attention_2"] = "flash_attention_2"
    compile: CompileConfig | None = None
    ac: ActivationCheckpointConfig | None = None
    reshard_after_forward: bool = True
    trust_remote_code: bool = False
    dp_replicate: int = 1
    ep: int = 1
    tp: int = 1
    cp: int = 1
    impl: Literal["hf", "liger_kernel", "custom"] = "hf"
    log_signature: bool = False
    load_using_meta: bool = False
    optimization_dtype: Literal["bfloat16", "float32"] = "float32"
    reduce_dtype: Literal["bfloat16", "float32"] = "float32"
    moe_use_grouped_mm: bool = True
    debug: DebugModelConfig = DebugModelConfig()
    experimental: ExperimentalConfig = ExperimentalConfig()

    @model_validator(mode="after")
    def _checks(self):
        if self.name in MOE_MODEL_MAPS:
            self.name = MOE_MODEL_MAPS[self.name]
        if self.trust_remote_code and self.impl != "hf":
            raise ValueError("Trust remote code is only supported with the HF implementation.")
        if self.debug.random_init and not self.load_using_meta:
            raise ValueError("Random initialize is only supported when loading with meta.")
        return self


# --------------------------------------------------------------------------- optimisation


class ConstantSchedulerConfig(_Strict):
    type: Literal["constant"] = "constant"


class LinearSchedulerConfig(_Strict):
    type: Literal["linear"] = "linear"
    warmup_steps: Annotated[int, Field(ge=0)] = 10
    decay_steps: Annotated[int, Field(ge=0)] = 10
    min_lr: Annotated[float, Field(ge=0)] = 0.0


class CosineSchedulerConfig(_Strict):
    type: Literal["cosine"] = "cosine"
    warmup_steps: Annotated[int, Field(ge=0)] = 10
    min_lr: Annotated[float, Field(ge=0)] = 0.0


SchedulerConfigType = Union[ConstantSchedulerConfig, LinearSchedulerConfig, CosineSchedulerConfig]


class _OptimizerBase(_Strict):
    lr: Annotated[float, Field(ge=0)] = 1e-6
    weight_decay: Annotated[float, Field(ge=0)] = 0.01
    max_norm: Annotated[float, Field(ge=0)] = 1.0


class SGDConfig(_OptimizerBase):
    type: Literal["sgd"] = "sgd"
    nesterov: bool = True
    momentum: float = 0.9


class AdamWConfig(_OptimizerBase):
    type: Literal["adamw"] = "adamw"
    betas1: Annotated[float, Field(ge=0)] = 0.9
    betas2: Annotated[float, Field(ge=0)] = 0.999


class MuonConfig(_OptimizerBase):
    type: Literal["muon"] = "muon"
    betas1: Annotated[float, Field(ge=0)] = 0.9
    betas2: Annotated[float, Field(ge=0)] = 0.999


OptimizerConfigType = Union[SGDConfig, AdamWConfig, MuonConfig]


# --------------------------------------------------------------------------- checkpoints, logging


class CheckpointConfig(BaseConfig):
    interval: Annotated[int | None, Field(ge=1)] = None
    resume_step: Annotated[int | None, Field(ge=-1)] = None
    keep: Annotated[int | None, Field(ge=1)] = None
    skip_dataloader: bool = False


class WeightCheckpointConfig(BaseConfig):
    interval: Annotated[int | None, Field(ge=1)] = None
    save_async: bool = True


class LogConfig(BaseConfig):
    level: str = "info"
    file: bool = True
    log_data: bool = False


class LogExtrasConfig(BaseConfig):
    samples: bool = True
    distributions: bool = True
    interval: Annotated[int, Field(ge=1)] = 10


class WandbMonitorConfig(BaseConfig):
    project: str = "prime-rl"
    name: str | None = None
    offline: bool = False
    id: str | None = None
    log_extras: LogExtrasConfig | None = LogExtrasConfig()


class HuggingFaceConfig(BaseConfig):
    repo_id: str
    organization: str | None = None
    private: bool = False
    optimizer_save_mode: str | None = None


class DataLoaderConfig(BaseConfig):
    fake: FakeDataLoaderConfig | None = None


# --------------------------------------------------------------------------- the trainer


class RLTrainerConfig(BaseConfig):
    """Configures the RL trainer (src/prime_rl/trainer/rl/config.py:81-191)."""

    model: ModelConfig = ModelConfig()
    data: DataLoaderConfig = DataLoaderConfig()
    loss: LossConfig = LossConfig()
    optim: Annotated[OptimizerConfigType, Field(discriminator="type")] = AdamWConfig()
    scheduler: Annotated[SchedulerConfigType, Field(discriminator="type")] = ConstantSchedulerConfig()
    ckpt: CheckpointConfig | None = None
    weights: WeightCheckpointConfig = WeightCheckpointConfig()
    log: LogConfig = LogConfig()
    wandb: WandbMonitorConfig | None = None
    output_dir: Path = Path("outputs")
    max_steps: int | None = None
    async_level: Annotated[int, Field(ge=0)] = 2
    memory_profiler_path: Path | None = None
    recompute_logprobs: bool = False
    bench: bool = False
    hf: HuggingFaceConfig | None = None
    trace_path: Path | None = None
    dist_timeout_seconds: int = 600
    toml_files: Annotated[list[str] | None, Field(exclude=True)] = None

    @model_validator(mode="before")
    @classmethod
    def _partial_sections(cls, data):
        # a recipe may set `[optim] lr = ...` without naming the type: the default kind is updated in place
        if isinstance(data, dict):
            for key, default in (("optim", "adamw"), ("scheduler", "constant")):
                if isinstance(data.get(key), dict) and "type" not in data[key]:
                    data = {**data, key: {"type": default, **data[key]}}
        return data

    @model_validator(mode="after")
    def _post(self):
        if self.bench:  # 1 warm-up + 3 measured steps on fake data, no extras, no checkpoints
            self.max_steps = 4
            if not self.data.fake:
                self.data.fake = FakeDataLoaderConfig()
            if self.wandb:
                self.wandb.log_extras = None
            self.ckpt = None
        if self.wandb and self.wandb.log_extras:
            self.wandb.log_extras.samples = False
        if self.trace_path:
            if self.max_steps is None:
                raise ValueError("Must specify max_steps when tracing")
            if self.max_steps >= 10:
                raise ValueError("Tracing more than 10 steps is not recommended as your trace will be massive.")
        return self


# --------------------------------------------------------------------------- sources


def _deep_merge(base: dict, over: dict) -> dict:
    out = dict(base)
    for k, v in over.items():
        out[k] = _deep_merge(out[k], v) if isinstance(v, dict) and isinstance(out.get(k), dict) else v
    return out


def _nest(keys: list[str], value: Any) -> dict:
    for k in reversed(keys):
        value = {k: value}
    return value


def load_toml(path: Path, _seen: set[Path] | None = None) -> dict:
    """One TOML file with the files it names in `toml_files` merged over it."""
    path = Path(path)
    seen = _seen if _seen is not None else set()
    if path.resolve() in seen:
        return {}
    if not path.exists():
        raise FileNotFoundError(f"TOML file {path} does not exist")
    seen.add(path.resolve())
    with open(path, "rb") as f:
        data = tomli.load(f)
    for extra in data.pop("toml_files", None) or []:
        if str(extra).endswith(".toml"):
            data = _deep_merge(data, load_toml(path.parent / extra, seen))
    return data


def extract_toml_paths(args: list[str]) -> tuple[list[tuple[str | None, str]], list[str]]:
    """Split `@ file.toml` (optionally preceded by `--key`) out of the argument list."""
    files: list[tuple[str | None, str]] = []
    rest: list[str] = []
    i = 0
    while i < len(args):
        a = args[i]
        if a == "@" or (a.startswith("@") and len(a) > 1):
            path = args[i + 1] if a == "@" else a[1:]
            key = None
            if rest and rest[-1].startswith("--") and "=" not in rest[-1]:
                key = rest.pop()[2:].replace("-", "_")
            files.append((key, path))
            i += 2 if a == "@" else 1
        else:
            rest.append(a)
            i += 1
    return files, rest


def _field_type(cls: Type[BaseModel], dotted: list[str]) -> Any:
    """Annotation of a dotted field path, or None when it cannot be followed statically."""
    ann: Any = cls
    for part in dotted:
        models = [m for m in _models_in(ann) if part in m.model_fields]
        if not models:
            return None
        ann = models[0].model_fields[part].annotation
    return ann


def _models_in(ann: Any) -> list[Type[BaseModel]]:
    import typing

    if isinstance(ann, type) and issubclass(ann, BaseModel):
        return [ann]
    out: list[Type[BaseModel]] = []
    for arg in typing.get_args(ann):
        out += _models_in(arg)
    return out


def _is_bool(ann: Any) -> bool:
    import typing

    return ann is bool or bool in typing.get_args(ann)


def _parse_value(text: str) -> Any:
    """Numbers, booleans, lists and quoted strings as JSON; anything else stays a string for pydantic to coerce."""
    try:
        return json.loads(text)
    except (ValueError, TypeError):
        low = text.lower()
        if low in ("true", "false"):
            return low == "true"
        return text


def parse_cli(cls: Type[BaseModel], args: list[str]) -> dict:
    """`--a.b value`, `--a.b=value`, `--flag`, `--no-flag` into a nested dict."""
    out: dict = {}
    i = 0
    while i < len(args):
        a = args[i]
        if not a.startswith("--"):
            raise ValueError(f"unexpected argument {a!r}")
        body = a[2:]
        value: Any
        if "=" in body:
            body, text = body.split("=", 1)
            value = _parse_value(text)
            i += 1
        else:
            keys = body.replace("-", "_").split(".")
            ann = _field_type(cls, keys)
            has_value = i + 1 < len(args) and not args[i + 1].startswith("--")
            if ann is None and keys[-1].startswith("no_") and _is_bool(_field_type(cls, keys[:-1] + [keys[-1][3:]])):
                body, value = ".".join(keys[:-1] + [keys[-1][3:]]), False
                i += 1
            elif has_value:
                value = _parse_value(args[i + 1])
                i += 2
            elif ann is None or _is_bool(ann):
                value = True
                i += 1
            else:  # `--ckpt` alone switches an optional sub-config on with its defaults
                value = {}
                i += 1
        out = _deep_merge(out, _nest(body.replace("-", "_").split("."), value))
    return out


def parse_env(prefix: str = "PRIME_") -> dict:
    out: dict = {}
    for k, v in os.environ.items():
        if k.startswith(prefix) and len(k) > len(prefix):
            out = _deep_merge(out, _nest(k[len(prefix):].lower().split("__"), _parse_value(v)))
    return out


T = TypeVar("T", bound=BaseModel)


def parse_argv(config_cls: Type[T], argv: list[str] | None = None, use_env: bool = True) -> T:
    """Build `config_cls` from TOML files, the command line and the environment
    (src/prime_rl/utils/pydantic_config.py:250-273)."""
    args = list(sys.argv[1:] if argv is None else argv)
    files, rest = extract_toml_paths(args)
    data: dict = {}
    if use_env:
        known = set(config_cls.model_fields)
        data = {k: v for k, v in parse_env().items() if k in known}
    for key, path in files:
        loaded = load_toml(Path(path))
        data = _deep_merge(data, _nest(key.split("."), loaded) if key else loaded)
    data = _deep_merge(data, parse_cli(config_cls, rest))
    return config_cls(**data)


def load_config(config_cls: Type[T], *toml_paths: str | Path, **overrides: Any) -> T:
    """Programmatic form: files in order (later ones win), then keyword overrides."""
    data: dict = {}
    for p in toml_paths:
        data = _deep_merge(data, load_toml(Path(p)))
    return config_cls(**_deep_merge(data, overrides))
