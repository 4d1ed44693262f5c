"""Model configuration for the FlowLM + Mimi-decode hot path.

Reads the same YAML files as the reference (`pocket_tts/utils/config.py:111-135`,
schema `pocket_tts/config/english.yaml:7-61`) and rejects unknown keys the way the
reference's pydantic `extra="forbid"` does (`utils/config.py:11-12`).  Only plain
dataclasses + PyYAML are used so the host side has no pydantic dependency.
"""

from __future__ import annotations

import dataclasses
from dataclasses import dataclass, field
from pathlib import Path

import yaml

CONFIGS_DIR = Path(__file__).parent / "config"


def _strict(cls, d: dict, where: str):
    names = {f.name for f in dataclasses.fields(cls)}
    extra = set(d) - names
    if extra:
        raise ValueError(f"{where}: extra fields not permitted: {sorted(extra)}")
    missing = [
        f.name
        for f in dataclasses.fields(cls)
        if f.default is dataclasses.MISSING
        and f.default_factory is dataclasses.MISSING
        and f.name not in d
    ]
    if missing:
        raise ValueError(f"{where}: missing fields: {missing}")
    return cls(**d)


@dataclass
class FlowConfig:
    dim: int
    depth: int


@dataclass
class FlowLMTransformerConfig:
    hidden_scale: int
    max_period: int
    d_model: int
    num_heads: int
    num_layers: int


@dataclass
class LookupTable:
    dim: int
    n_bins: int
    tokenizer: str
    tokenizer_path: str


@dataclass
class FlowLMConfig:
    dtype: str
    flow: FlowConfig
    transformer: FlowLMTransformerConfig
    lookup_table: LookupTable
    weights_path: str | None = None
    insert_bos_before_voice: bool = False


@dataclass
class SEANetConfig:
    dimension: int
    channels: int
    n_filters: int
    n_residual_layers: int
    ratios: list
    kernel_size: int
    residual_kernel_size: int
    last_kernel_size: int
    dilation_base: int
    pad_mode: str
    compress: int


@dataclass
class MimiTransformerConfig:
    d_model: int
    input_dimension: int
    output_dimensions: tuple
    num_heads: int
    num_layers: int
    layer_scale: float
    context: int
    dim_feedforward: int
    max_period: float = 10000.0


@dataclass
class QuantizerConfig:
    dimension: int
    output_dimension: int


@dataclass
class MimiConfig:
    dtype: str
    sample_rate: int
    channels: int
    frame_rate: float
    seanet: SEANetConfig
    transformer: MimiTransformerConfig
    quantizer: QuantizerConfig
    weights_path: str | None = None
    inner_dim: int | None = None
    outer_dim: int | None = None


@dataclass
class Config:
    flow_lm: FlowLMConfig
    mimi: MimiConfig
    weights_path: str | None = None
    weights_path_without_voice_cloning: str | None = None
    pad_with_spaces_for_short_inputs: bool = False
    remove_semicolons: bool = False
    model_recommended_frames_after_eos: int | None = None

    # ---- derived quantities used all over the hot path -------------------
    @property
    def hop_length(self) -> int:
        h = 1
        for r in self.mimi.seanet.ratios:
            h *= int(r)
        return h

    @property
    def encoder_frame_rate(self) -> float:
        return self.mimi.sample_rate / self.hop_length

    @property
    def upsample_stride(self) -> int:
        """`int(encoder_frame_rate / frame_rate)` (reference `mimi.py:53-66`)."""
        return int(self.encoder_frame_rate / self.mimi.frame_rate)

    @property
    def frame_samples(self) -> int:
        return self.upsample_stride * self.hop_length


def config_from_dict(d: dict) -> Config:
    d = dict(d)
    fl = dict(d["flow_lm"])
    fl["flow"] = _strict(FlowConfig, fl["flow"], "flow_lm.flow")
    fl["transformer"] = _strict(FlowLMTransformerConfig, fl["transformer"], "flow_lm.transformer")
    fl["lookup_table"] = _strict(LookupTable, fl["lookup_table"], "flow_lm.lookup_table")
    d["flow_lm"] = _strict(FlowLMConfig, fl, "flow_lm")
    mi = dict(d["mimi"])
    mi["seanet"] = _strict(SEANetConfig, mi["seanet"], "mimi.seanet")
    tr = dict(mi["transformer"])
    tr["output_dimensions"] = tuple(tr["output_dimensions"])
    mi["transformer"] = _strict(MimiTransformerConfig, tr, "mimi.transformer")
    mi["quantizer"] = _strict(QuantizerConfig, mi["quantizer"], "mimi.quantizer")
    d["mimi"] = _strict(MimiConfig, mi, "mimi")
    return _strict(Config, d, "config")


def load_config(yaml_path: str | Path) -> Config:
    """Same error behaviour as the reference loader (`utils/config.py:121-135`)."""
    yaml_path = Path(yaml_path)
    if not yaml_path.exists():
        if yaml_path.is_relative_to(CONFIGS_DIR):
            raise FileNotFoundError(
                f"Config file not found: {yaml_path}. Did you make a typo? "
                f"Available languages: {[p.stem for p in CONFIGS_DIR.glob('*.yaml')]}"
            )
        raise FileNotFoundError(f"Config file not found: {yaml_path}. Did you make a typo?")
    with open(yaml_path, "r") as f:
        return config_from_dict(yaml.safe_load(f))


def config_to_dict(cfg: Config) -> dict:
    d = dataclasses.asdict(cfg)
    d["mimi"]["transformer"]["output_dimensions"] = list(
        d["mimi"]["transformer"]["output_dimensions"]
    )
    return d


def make_config(
    num_layers: int = 6,
    d_model: int = 1024,
    num_heads: int = 16,
    flow_dim: int = 512,
    flow_depth: int = 6,
    n_bins: int = 4000,
    seanet_dimension: int = 512,
    n_filters: int = 64,
    mimi_d_model: int = 512,
    mimi_heads: int = 8,
    mimi_ff: int = 2048,
    mimi_layers: int = 2,
    context: int = 250,
    tokenizer_path: str = "synthetic",
) -> Config:
    """Programmatic config with the 100M-English defaults (`english.yaml:7-61`).

    `num_layers=24` gives the `*_24l` variants (`italian_24l.yaml:18`); the small
    dims are used by the exhaustive per-op parity tests.
    """
    return config_from_dict(
        dict(
            flow_lm=dict(
                insert_bos_before_voice=True,
                dtype="float32",
                flow=dict(depth=flow_depth, dim=flow_dim),
                transformer=dict(
                    d_model=d_model,
                    hidden_scale=4,
                    max_period=10000,
                    num_heads=num_heads,
                    num_layers=num_layers,
                ),
                lookup_table=dict(
                    dim=d_model, n_bins=n_bins, tokenizer="sentencepiece", tokenizer_path=tokenizer_path
                ),
            ),
            mimi=dict(
                dtype="float32",
                sample_rate=24000,
                inner_dim=32,
                outer_dim=seanet_dimension,
                channels=1,
                frame_rate=12.5,
                seanet=dict(
                    dimension=seanet_dimension,
                    channels=1,
                    n_filters=n_filters,
                    n_residual_layers=1,
                    ratios=[6, 5, 4],
                    kernel_size=7,
                    residual_kernel_size=3,
                    last_kernel_size=3,
                    dilation_base=2,
                    pad_mode="constant",
                    compress=2,
                ),
                transformer=dict(
                    d_model=mimi_d_model,
                    num_heads=mimi_heads,
                    num_layers=mimi_layers,
                    layer_scale=0.01,
                    context=context,
                    dim_feedforward=mimi_ff,
                    input_dimension=seanet_dimension,
                    output_dimensions=[seanet_dimension],
                ),
                quantizer=dict(dimension=32, output_dimension=seanet_dimension),
            ),
        )
    )


NAMED_CONFIGS = {
    # 100M English model: BASELINE.json configs #2/#3/#5
    "en100m": dict(),
    # 24-layer variant: BASELINE.json config #4
    "24l": dict(num_layers=24),
    # tiny variant for exhaustive per-op golden vectors
    "tiny": dict(
        num_layers=2,
        d_model=128,
        num_heads=2,
        flow_dim=64,
        flow_depth=2,
        n_bins=64,
        seanet_dimension=256,
        n_filters=32,
        mimi_d_model=256,
        mimi_heads=4,
        mimi_ff=512,
        mimi_layers=2,
        context=40,
    ),
}


def named_config(name: str) -> Config:
    return make_config(**NAMED_CONFIGS[name])
