"""Error-compensated ("split") bf16 codec GEMMs (PTTS_CODEC_SPLIT; VERDICT r2 next #8: hi*hi + hi*lo + lo*hi of bf16 halves on
the bf16 MFMA, fp32 accumulation, fp32 buffers) must pass the fp32 codec's parity checks AT THE fp32 TOLERANCE: every codec
golden (reference-generated per-stage taps of frames 0-2 and the PCM of every frame, tiny / en100m / 24-layer configs), the
real-size ring-wrap run against the numpy oracle, and the batch-64 shapes.  Tolerance: max-abs <= 2e-4 (test_gpu_parity.ATOL).
The mode is an experiment reported beside the fp32 headline; these tests are what "same results" means for it.  `-m gpu`."""

import numpy as np
import pytest
import torch

from conftest import synth_weights
from test_gpu_parity import ATOL, _maxerr, dev

CASES = ["en100m_b1", "en100m_b2_noise", "24l_b1"]  # the tiny config has 16-channel layers (odd fragment counts): refused by the mode

pytestmark = pytest.mark.gpu
_ENG = {}


def split_engine(cfg_name, seed=0):
    from pocket_tts_amd.engine import Engine

    key = (cfg_name, seed)
    if key not in _ENG:
        for e in _ENG.values():
            e.close()
        _ENG.clear()
        cfg, W = synth_weights(cfg_name, seed)
        _ENG[key] = Engine(cfg, W, "cuda:0", quantize_groups={"codec_split"})
    return _ENG[key]


@pytest.mark.parametrize("case", CASES)
def test_split_codec_vs_reference_goldens(golden, case):
    g = golden(case)
    m = g["meta"]
    eng = split_engine(m["config"], m["seed"])
    B = m["B"]
    ms = eng.new_mimi_state(B)
    worst = 0.0
    for f in range(m["n_frames"]):
        pcm = eng.mimi_decode(ms, dev(g["mimi_latents"][f]))
        torch.cuda.synchronize()
        if f < 3:
            for k in [k for k in g if k.startswith("tap_")]:
                name = k[4:]
                ref = g[k][f]
                if name == "seanet11":
                    continue
                got = eng.debug_read(ms, name).cpu().numpy()
                got = got.reshape(B, ref.shape[2], ref.shape[1]).transpose(0, 2, 1)
                if name in ("seanet0", "seanet3", "seanet6", "seanet9"):
                    ref = np.where(ref > 0, ref, np.expm1(np.minimum(ref, 0)))
                assert _maxerr(got, ref) < ATOL, f"{name} frame {f}"
        e = _maxerr(pcm.cpu().numpy(), g["pcm"][f])
        worst = max(worst, e)
        assert e < ATOL, f"pcm frame {f}"
    print(f"{case}: split-bf16 codec worst PCM error {worst:.2e} (tolerance {ATOL:.0e})")
    ms.close()


@pytest.mark.parametrize("B,nf,tuned", [(2, 20, False), (64, 2, True)])
def test_split_codec_vs_oracle_real_size(B, nf, tuned):
    """en100m: 20 frames (the 272-slot ring wraps, the 250-key window is active) at batch 2, and the BASELINE config #3
    shapes (batch 64, tuned tiles) against the numpy oracle"""
    from oracle import np_oracle as O

    cfg, W = synth_weights("en100m")
    eng = split_engine("en100m")
    if tuned:
        eng.tune(B)
    dec = O.MimiDecoder(cfg, W)
    rng = np.random.default_rng(77 + B)
    lat = rng.standard_normal((nf, B, eng.ldim)).astype(np.float32)
    ost, ms = dec.init_state(B, nf), eng.new_mimi_state(B)
    worst = 0.0
    for f in range(nf):
        ref = dec.decode(ost, lat[f])
        got = eng.mimi_decode(ms, dev(lat[f]))
        torch.cuda.synchronize()
        e = _maxerr(got.cpu().numpy(), ref)
        worst = max(worst, e)
        assert e < ATOL, f"frame {f}"
    print(f"en100m B={B}: split-bf16 codec worst PCM error {worst:.2e} over {nf} frames (tolerance {ATOL:.0e})")
    ms.close()


def test_split_engine_teardown():
    for e in _ENG.values():
        e.close()
    _ENG.clear()
