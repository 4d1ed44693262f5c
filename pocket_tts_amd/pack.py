"""Offline packer: checkpoint -> packed engine files, once, ahead of deployment.

    python -m pocket_tts_amd.pack --config english.yaml [--quantize | --lm-bf16] [--codec-bf16 | --codec-fp8 | --codec-split] --out model.ptts

Writes `model.ptts` (everything `ptts_create_ex` builds on the device: MFMA-fragment-ordered weights, int8 / bf16
/ e4m3 / split-bf16 images with their scales, LayerNorm-fold vectors), `model.ptts.aux.safetensors` (embedding table, bos_before_voice) and `model.ptts.yaml`
(config + weight format).  `Engine.from_packed("model.ptts")` then starts without the fp32 checkpoint and without
packing or quantising again (the reference quantises at every load: quantization.py:60-88, tts_model.py:312-313).
Needs the GPU: the packing kernels are device kernels."""

from __future__ import annotations

import argparse
from pathlib import Path


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--config", required=True, help="model YAML (weights_path: local .safetensors, or null = synthetic)")
    ap.add_argument("--quantize", action="store_true", help="int8 weights for the FlowLM attention + FFN layers")
    ap.add_argument("--codec-bf16", action="store_true", help="bf16 Mimi decoder (weights + activations)")
    ap.add_argument("--codec-fp8", action="store_true", help="SEANet convolutions on the fp8 MFMA (e4m3, calibrated activation scales)")
    ap.add_argument("--codec-split", action="store_true", help="codec GEMMs on error-compensated bf16 (fp32 accuracy)")
    ap.add_argument("--lm-bf16", action="store_true", help="bf16 weights + operands for the FlowLM Linear layers")
    ap.add_argument("--device", default="cuda:0")
    ap.add_argument("--out", required=True)
    a = ap.parse_args(argv)
    from .config import load_config
    from .engine import Engine
    from .tts_model import _load_weights

    cfg = load_config(Path(a.config))
    groups = (({"attention", "ffn"} if a.quantize else set()) | ({"codec_bf16"} if a.codec_bf16 else set())
              | ({"codec_fp8"} if a.codec_fp8 else set()) | ({"codec_split"} if a.codec_split else set())
              | ({"lm_bf16"} if a.lm_bf16 else set()))
    eng = Engine(cfg, _load_weights(cfg), a.device, quantize_groups=groups or None)
    eng.save_packed(a.out)
    eng.close()
    print(f"wrote {a.out} (+ .aux.safetensors, .yaml)")


if __name__ == "__main__":
    main()
