"""fp8 codec convolutions (PTTS_CODEC_FP8: the SEANet decoder convs on v_mfma_f32_16x16x32_fp8_fp8 with e4m3 weights,
per-output-channel weight scales, static per-tensor activation scales; Mimi transformer bf16; FlowLM fp32) against this
build's fp32 codec and against the numpy oracle.  BASELINE.json configs[4] names "fp8 MFMA codec convs"; the reference has
NO counterpart (docs/quantization.md:67-76: Mimi is never quantised), so parity is UNPINNED and quality is reported the way
the reference's quantisation harness does: SNR against the full-precision output (scripts/evaluate_quantization.py:215-228).
EOS decisions / frame counts come from the FlowLM, which is untouched: identical by construction, checked end to end.
`-m gpu`."""

from pathlib import Path

import numpy as np
import pytest
import torch

from conftest import synth_weights
from test_gpu_bf16 import snr_db
from test_gpu_parity import dev

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("B,nf", [(2, 12), (64, 3)])
def test_fp8_codec_snr_vs_fp32_and_bf16(B, nf):
    """The number that decides whether the format is worth using: SNR of the fp8-conv codec against the fp32 codec (and the
    bf16 codec's SNR beside it).  e4m3 carries 3 mantissa bits (relative step 2^-4 .. 2^-3), ten convolutions deep."""
    from oracle import np_oracle as O
    from pocket_tts_amd.engine import Engine

    cfg, W = synth_weights("en100m")
    rng = np.random.default_rng(8)
    lat = rng.standard_normal((nf, B, cfg.mimi.quantizer.dimension)).astype(np.float32)
    outs = {}
    for name, groups in (("fp32", None), ("bf16", {"codec_bf16"}), ("fp8", {"codec_fp8"})):
        eng = Engine(cfg, W, "cuda:0", quantize_groups=groups)
        ms = eng.new_mimi_state(B)
        pcm = [eng.mimi_decode(ms, dev(lat[f])).cpu().numpy() for f in range(nf)]
        ms.reset_row(B - 1)  # continuous batching: the per-sequence carries are cleared in every buffer format
        pcm += [eng.mimi_decode(ms, dev(lat[f])).cpu().numpy() for f in range(2)]
        outs[name] = np.stack(pcm)
        if name == "fp8":
            assert eng.mimi_weight_bytes() < bf16_bytes  # the transformer (bf16) and conv0 dominate the codec's weight bytes
        if name == "bf16":
            bf16_bytes = eng.mimi_weight_bytes()
        eng.close()
    assert np.isfinite(outs["fp8"]).all()
    s8, s16 = snr_db(outs["fp32"], outs["fp8"]), snr_db(outs["fp32"], outs["bf16"])
    print(f"B={B}: fp8-conv codec SNR {s8:.1f} dB vs fp32 ({nf + 2} frames); bf16 codec {s16:.1f} dB")
    if B == 2:  # and against the ORACLE's fp32 codec (the build's fp32 path is within 2e-4 of it)
        d = O.MimiDecoder(cfg, W)
        st = d.init_state(B, nf)
        ref = np.stack([d.decode(st, lat[f]) for f in range(nf)])
        print(f"      vs the numpy oracle: {snr_db(ref, outs['fp8'][:nf]):.1f} dB")
    assert s8 > 12.0, s8  # a sanity floor, not a quality claim: the measured value is what DESIGN.md reports
    assert s16 > s8
    # after the row reset, the reset row reproduces its own first frames (carries really cleared)
    assert snr_db(outs["fp8"][0][B - 1], outs["fp8"][nf][B - 1]) > 60.0


def test_fp8_codec_end_to_end_frame_counts():
    """TTSModel(codec_fp8=True): same EOS decisions / frame count as the fp32 model, waveform SNR reported"""
    from pocket_tts_amd import TTSModel

    G = Path(__file__).parent / "golden"
    text = "Hello world. This is a test."
    wavs = []
    for flag in (False, True):
        m = TTSModel.load_model(config=G / "e2e2_en100m.yaml", temp=0.0, codec_fp8=flag)
        st = m.get_state_for_conditioning(torch.randn(1, 12, 1024, generator=torch.Generator().manual_seed(3)) * 0.1)
        wavs.append(m.generate_audio(st, text, frames_after_eos=2).numpy())
        m.engine.close()
    assert wavs[0].shape == wavs[1].shape and wavs[0].shape[0] >= 1920
    print(f"end to end ({wavs[0].shape[0] // 1920} frames): fp8-conv codec SNR {snr_db(wavs[0], wavs[1]):.1f} dB vs fp32")


def test_fp8_and_bf16_codec_flags_are_exclusive():
    from pocket_tts_amd._lib import PttsError
    from pocket_tts_amd.engine import Engine

    cfg, W = synth_weights("en100m")
    with pytest.raises(PttsError, match="exclusive"):
        Engine(cfg, W, "cuda:0", quantize_groups={"codec_bf16", "codec_fp8"})
    with pytest.raises(PttsError, match="exclusive"):
        Engine(cfg, W, "cuda:0", quantize_groups={"lm_bf16", "attention"})
