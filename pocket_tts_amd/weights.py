"""Weight inventory of the hot path + a deterministic synthetic-weight generator.

There is no network here or on the GPU box, so the shipped `hf://` checkpoints
(`pocket_tts/config/english.yaml:3-4`) cannot be fetched.  Parity and benchmarks
therefore run on synthetic weights that are a pure function of
`(seed, tensor name, shape)` -- integer hashing only, so the values are bit-identical
on every machine.  The same generator feeds the reference modules when golden vectors
are produced (tests/golden/gen_*.py) and feeds the HIP engine on the GPU box.

`state_dict_spec` lists the tensors of `TTSModel.state_dict()` that the decode hot
path reads (names as in the reference checkpoint), so a real
safetensors checkpoint can be dropped in unchanged.
"""

from __future__ import annotations

import hashlib
import math

import numpy as np

from .config import Config

_MASK = np.uint64(0xFFFFFFFFFFFFFFFF)


def _name_key(seed: int, name: str) -> int:
    h = hashlib.sha256(f"{seed}:{name}".encode()).digest()
    return int.from_bytes(h[:8], "little")


def uniform_pm1(seed: int, name: str, n: int) -> np.ndarray:
    """n float32 values in [-1, 1), splitmix64 of (key(name) + index)."""
    key = np.uint64(_name_key(seed, name))
    with np.errstate(over="ignore"):
        z = np.arange(n, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15) + key
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    u24 = (z >> np.uint64(40)).astype(np.float32)  # 24 random bits, exact in fp32
    return u24 * np.float32(2.0 / (1 << 24)) - np.float32(1.0)


# --------------------------------------------------------------------------
# inventory
# --------------------------------------------------------------------------
def _transformer_layer(prefix: str, d: int, ff: int, layer_scale: bool) -> dict:
    s = {
        f"{prefix}.self_attn.in_proj.weight": (3 * d, d),
        f"{prefix}.self_attn.out_proj.weight": (d, d),
        f"{prefix}.norm1.weight": (d,),
        f"{prefix}.norm1.bias": (d,),
        f"{prefix}.norm2.weight": (d,),
        f"{prefix}.norm2.bias": (d,),
        f"{prefix}.linear1.weight": (ff, d),
        f"{prefix}.linear2.weight": (d, ff),
    }
    if layer_scale:
        s[f"{prefix}.layer_scale_1.scale"] = (d,)
        s[f"{prefix}.layer_scale_2.scale"] = (d,)
    return s


def flow_lm_spec(cfg: Config) -> dict:
    """Tensors of `flow_lm.*` (reference `flow_lm.py:74-90`, `mlp.py:134-186`)."""
    t = cfg.flow_lm.transformer
    d, L = t.d_model, t.num_layers
    ff = d * t.hidden_scale
    ldim = cfg.mimi.quantizer.dimension
    fd, depth = cfg.flow_lm.flow.dim, cfg.flow_lm.flow.depth
    s: dict = {}
    p = "flow_lm."
    s[p + "emb_std"] = (ldim,)
    s[p + "emb_mean"] = (ldim,)
    s[p + "bos_emb"] = (ldim,)
    if cfg.flow_lm.insert_bos_before_voice:
        s[p + "bos_before_voice"] = (1, 1, d)
    s[p + "speaker_proj_weight"] = (d, cfg.mimi.inner_dim or cfg.mimi.seanet.dimension)
    s[p + "conditioner.embed.weight"] = (cfg.flow_lm.lookup_table.n_bins + 1, cfg.flow_lm.lookup_table.dim)
    s[p + "input_linear.weight"] = (d, ldim)
    for i in range(L):
        s.update(_transformer_layer(f"{p}transformer.layers.{i}", d, ff, False))
    s[p + "out_norm.weight"] = (d,)
    s[p + "out_norm.bias"] = (d,)
    s[p + "out_eos.weight"] = (1, d)
    s[p + "out_eos.bias"] = (1,)
    f = p + "flow_net."
    for i in range(2):
        s[f"{f}time_embed.{i}.freqs"] = (128,)
        s[f"{f}time_embed.{i}.mlp.0.weight"] = (fd, 256)
        s[f"{f}time_embed.{i}.mlp.0.bias"] = (fd,)
        s[f"{f}time_embed.{i}.mlp.2.weight"] = (fd, fd)
        s[f"{f}time_embed.{i}.mlp.2.bias"] = (fd,)
        s[f"{f}time_embed.{i}.mlp.3.alpha"] = (fd,)
    s[f + "cond_embed.weight"] = (fd, d)
    s[f + "cond_embed.bias"] = (fd,)
    s[f + "input_proj.weight"] = (fd, ldim)
    s[f + "input_proj.bias"] = (fd,)
    for i in range(depth):
        r = f"{f}res_blocks.{i}."
        s[r + "in_ln.weight"] = (fd,)
        s[r + "in_ln.bias"] = (fd,)
        s[r + "mlp.0.weight"] = (fd, fd)
        s[r + "mlp.0.bias"] = (fd,)
        s[r + "mlp.2.weight"] = (fd, fd)
        s[r + "mlp.2.bias"] = (fd,)
        s[r + "adaLN_modulation.1.weight"] = (3 * fd, fd)
        s[r + "adaLN_modulation.1.bias"] = (3 * fd,)
    s[f + "final_layer.linear.weight"] = (ldim, fd)
    s[f + "final_layer.linear.bias"] = (ldim,)
    s[f + "final_layer.adaLN_modulation.1.weight"] = (2 * fd, fd)
    s[f + "final_layer.adaLN_modulation.1.bias"] = (2 * fd,)
    return s


def seanet_decoder_layers(cfg: Config) -> list:
    """Structure of `SEANetDecoder.model` (reference `seanet.py:141-172`).

    Returns a list of (index-in-ModuleList, kind, cin, cout, kernel, stride); ELU entries
    are implicit.  Only `n_residual_layers == 1` (every shipped config) is supported.
    """
    sn = cfg.mimi.seanet
    assert sn.n_residual_layers == 1, "hot path supports n_residual_layers=1 (all shipped configs)"
    mult = 2 ** len(sn.ratios)
    layers = [(0, "conv", sn.dimension, mult * sn.n_filters, sn.kernel_size, 1)]
    idx = 1
    for r in sn.ratios:
        cin = mult * sn.n_filters
        cout = cin // 2
        layers.append((idx + 1, "convtr", cin, cout, 2 * r, r))  # idx = ELU
        hid = cout // sn.compress
        layers.append((idx + 2, "res", cout, hid, sn.residual_kernel_size, 1))
        idx += 3
        mult //= 2
    layers.append((idx + 1, "conv", sn.n_filters, sn.channels, sn.last_kernel_size, 1))
    return layers


def mimi_decode_spec(cfg: Config) -> dict:
    """Decode-side tensors of `mimi.*` (reference `mimi.py:89-94`)."""
    m = cfg.mimi
    s: dict = {}
    q = m.quantizer
    s["mimi.quantizer.output_proj.weight"] = (q.output_dimension, q.dimension, 1)
    st = cfg.upsample_stride
    s["mimi.upsample.convtr.convtr.weight"] = (m.seanet.dimension, 1, 2 * st)
    tr = m.transformer
    assert tr.d_model == tr.input_dimension and tuple(tr.output_dimensions) == (tr.d_model,), (
        "hot path assumes no input/output projection in the Mimi decoder transformer "
        "(true for every shipped config)"
    )
    for i in range(tr.num_layers):
        s.update(
            _transformer_layer(
                f"mimi.decoder_transformer.transformer.layers.{i}", tr.d_model, tr.dim_feedforward, True
            )
        )
    for idx, kind, cin, cout, k, stride in seanet_decoder_layers(cfg):
        p = f"mimi.decoder.model.{idx}"
        if kind == "conv":
            s[f"{p}.conv.weight"] = (cout, cin, k)
            s[f"{p}.conv.bias"] = (cout,)
        elif kind == "convtr":
            s[f"{p}.convtr.weight"] = (cin, cout, k)
            s[f"{p}.convtr.bias"] = (cout,)
        else:  # res block: ELU, conv k (cin->hid), ELU, conv 1 (hid->cin)
            hid = cout
            s[f"{p}.block.1.conv.weight"] = (hid, cin, k)
            s[f"{p}.block.1.conv.bias"] = (hid,)
            s[f"{p}.block.3.conv.weight"] = (cin, hid, 1)
            s[f"{p}.block.3.conv.bias"] = (cin,)
    return s


def seanet_encoder_layers(cfg: Config) -> list:
    """Structure of `SEANetEncoder.model` (reference `seanet.py:63-104`): list of
    (index-in-ModuleList, kind, cin, cout, kernel, stride) with kind in conv / res / down."""
    sn = cfg.mimi.seanet
    assert sn.n_residual_layers == 1
    mult = 1
    layers = [(0, "conv", sn.channels, mult * sn.n_filters, sn.kernel_size, 1)]
    idx = 1
    for r in reversed(sn.ratios):
        dim = mult * sn.n_filters
        layers.append((idx, "res", dim, dim // sn.compress, sn.residual_kernel_size, 1))
        layers.append((idx + 2, "down", dim, 2 * dim, 2 * r, r))  # idx + 1 = ELU
        idx += 3
        mult *= 2
    layers.append((idx + 1, "conv", mult * sn.n_filters, sn.dimension, sn.last_kernel_size, 1))
    return layers


def mimi_encode_spec(cfg: Config) -> dict:
    """Encode-side tensors of `mimi.*` used by the voice-prompt path (reference `mimi.py:96-119`)."""
    m = cfg.mimi
    s: dict = {}
    for idx, kind, cin, cout, k, stride in seanet_encoder_layers(cfg):
        p = f"mimi.encoder.model.{idx}"
        if kind == "res":
            hid = cout
            s[f"{p}.block.1.conv.weight"] = (hid, cin, k)
            s[f"{p}.block.1.conv.bias"] = (hid,)
            s[f"{p}.block.3.conv.weight"] = (cin, hid, 1)
            s[f"{p}.block.3.conv.bias"] = (cin,)
        else:
            s[f"{p}.conv.weight"] = (cout, cin, k)
            s[f"{p}.conv.bias"] = (cout,)
    tr = m.transformer
    for i in range(tr.num_layers):
        s.update(_transformer_layer(f"mimi.encoder_transformer.transformer.layers.{i}", tr.d_model,
                                    tr.dim_feedforward, True))
    st = cfg.upsample_stride
    inner = m.inner_dim or m.seanet.dimension
    s["mimi.downsample.conv.conv.weight"] = (inner, m.seanet.dimension, 2 * st)
    return s


def state_dict_spec(cfg: Config) -> dict:
    s = flow_lm_spec(cfg)
    s.update(mimi_decode_spec(cfg))
    s.update(mimi_encode_spec(cfg))
    return s


# --------------------------------------------------------------------------
# generator
# --------------------------------------------------------------------------
def _fan_in(name: str, shape: tuple) -> int:
    if ".convtr.weight" in name and "upsample" not in name:
        # ConvTranspose1d [cin, cout, k]: every output position sums cin * k/stride taps
        return shape[0] * 2
    if len(shape) == 3:
        return shape[1] * shape[2]
    return shape[-1]


def generate_tensor(name: str, shape: tuple, seed: int = 0) -> np.ndarray:
    """Synthetic value of one tensor: unit-gain uniform for matrices, ~1 for norm gains."""
    n = int(np.prod(shape))
    if name.endswith(".freqs"):
        # deterministic buffer of TimestepEmbedder (reference `mlp.py:75-77`)
        half = shape[0]
        k = np.arange(half, dtype=np.float32) * np.float32(-math.log(10000.0))
        return np.exp(k / np.float32(half)).astype(np.float32)
    u = uniform_pm1(seed, name, n).reshape(shape)
    leaf = name.rsplit(".", 1)[-1]
    if name.endswith("out_eos.weight"):
        # large enough that the EOS logit straddles the -4 threshold over a run
        return (u * np.float32(6.0 * math.sqrt(3.0 / shape[-1]))).astype(np.float32)
    if name.endswith("out_eos.bias"):
        return np.full(shape, -4.5, np.float32)
    if name.endswith("emb_std"):
        return (np.float32(1.0) + np.float32(0.25) * u).astype(np.float32)
    if name.endswith("emb_mean"):
        return (np.float32(0.1) * u).astype(np.float32)
    if ".layer_scale_" in name:
        return (np.float32(0.1) + np.float32(0.05) * u).astype(np.float32)
    if leaf == "alpha" or (leaf == "weight" and len(shape) == 1):
        return (np.float32(1.0) + np.float32(0.1) * u).astype(np.float32)
    if leaf == "bias":
        return (np.float32(0.1) * u).astype(np.float32)
    if len(shape) == 1 or name.endswith("bos_before_voice") or name.endswith("embed.weight"):
        return u.astype(np.float32)
    bound = math.sqrt(3.0 / _fan_in(name, shape))
    if name.endswith("final_layer.linear.weight"):
        # keeps the emitted latents O(1) like the normalised latents of a trained model;
        # unit gain here makes the autoregressive map needlessly chaotic (|latent| ~ 50)
        bound *= 0.05
    return (u * np.float32(bound)).astype(np.float32)


def generate_state_dict(cfg: Config, seed: int = 0, spec: dict | None = None) -> dict:
    spec = state_dict_spec(cfg) if spec is None else spec
    return {name: generate_tensor(name, tuple(shape), seed) for name, shape in spec.items()}


def count_params(spec: dict) -> int:
    return int(sum(int(np.prod(s)) for s in spec.values()))
