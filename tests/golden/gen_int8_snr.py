#!/usr/bin/env python3
"""What SNR does the REFERENCE's own int8 path reach against its fp32 path?  (dev container only)

    python tests/golden/gen_int8_snr.py     # writes tests/golden/int8_reference_snr.json

Runs the reference's `quantization.apply_dynamic_int8(flow_lm, {"attention", "ffn"})` (quantization.py:60-128,
the groups `load_model(quantize=True)` uses) on the reference's FlowLM modules holding the build's synthetic
en100m weights, with the torch.ao backend (torchao is not installed here), and compares 8 greedy (temp 0)
autoregressive latents with the fp32 modules on the inputs tests/test_gpu_quant.py::test_int8_snr_against_fp32
uses.  The fixture holds two numbers; the GPU test requires the build's weight-only int8 path to be no worse
(minus 3 dB)."""

import json
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent))
import gen_golden as G  # noqa: E402

from pocket_tts_amd.config import named_config  # noqa: E402


def snr_db(ref, x):
    ref, x = np.asarray(ref, np.float64), np.asarray(x, np.float64)
    return float(10 * np.log10((ref ** 2).mean() / ((ref - x) ** 2).mean()))


@torch.no_grad()
def run(lm, mods, emb, ns):
    B, Tp, _ = emb.shape
    ldim = lm.bos_emb.shape[0]
    state = mods["stateful_module"].init_states(lm, B, Tp + ns)
    G._flow_lm_step(lm, state, emb, torch.zeros(B, 0, ldim), mods, 1, None, -4.0)
    x = torch.full((B, 1, ldim), float("nan"))
    lat = []
    for _ in range(ns):
        cur, _, _, _ = G._flow_lm_step(lm, state, torch.zeros(B, 0, emb.shape[2]), x, mods, 1, torch.zeros(B, ldim), -4.0)
        lat.append(cur.numpy().copy())
        x = cur[:, None, :]
    return np.stack(lat)


def main():
    import importlib

    mods = G.import_reference_modules()
    quantization = importlib.import_module("pocket_tts.quantization")
    cfg = named_config("en100m")
    rng = np.random.default_rng(9)
    emb = torch.from_numpy((rng.standard_normal((2, 24, 1024)) * 0.5).astype(np.float32))
    lm, _, _ = G.build_reference(cfg, mods, 0)
    fp32 = run(lm, mods, emb, 8)
    quantization.apply_dynamic_int8(lm, {"attention", "ffn"})
    int8 = run(lm, mods, emb, 8)
    out = {
        "latent_snr_db": snr_db(fp32, int8),
        "first_step_latent_snr_db": snr_db(fp32[0], int8[0]),
        "backend": quantization._get_backend(), "engine": torch.backends.quantized.engine,
        "what": "reference apply_dynamic_int8({attention, ffn}) vs reference fp32, synthetic en100m weights seed 0, "
                "B=2, 24 prefill positions N(0,0.25) seed 9, 8 AR steps at temp 0",
    }
    print(out)
    with open(Path(__file__).resolve().parent / "int8_reference_snr.json", "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
