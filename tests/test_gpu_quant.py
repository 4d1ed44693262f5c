"""int8 weight path (BASELINE config #5: `load_model(quantize=True)`): the HIP kernels with int8 per-channel
weights against the numpy oracle running the same dequantised weights, plus the reference's quality metric
(SNR against the fp32 model, scripts/evaluate_quantization.py:215-228).  The reference's own int8 arithmetic is
torch.ao / torchao CPU code outside /root/reference (dynamic activation quantisation): parity against THAT is
unpinned; tests/golden/int8_reference_snr.json holds the SNR it reaches on the same synthetic weights."""

import json
import os

import numpy as np
import pytest
import torch

from conftest import synth_weights
from oracle import np_oracle as O

pytestmark = pytest.mark.gpu
ATOL = 2e-4
GROUPS = {"attention", "ffn"}


def _dev(x):
    return torch.from_numpy(np.ascontiguousarray(x)).to("cuda:0")


def _snr_db(ref, x):
    ref, x = np.asarray(ref, np.float64), np.asarray(x, np.float64)
    return float(10 * np.log10((ref ** 2).mean() / max(((ref - x) ** 2).mean(), 1e-300)))


def _run_gpu(eng, emb, ns, tune_batch=None):
    B, Tp = emb.shape[:2]
    if tune_batch:
        eng.tune(tune_batch)
    st, ms = eng.new_lm_state(B, Tp + ns), eng.new_mimi_state(B)
    eng.lm_prefill(st, _dev(emb))
    lat, logit, pcm = [], [], []
    for _ in range(ns):
        o, lg, _ = eng.lm_decode_step(st, None, None, 1, -4.0)
        p = eng.mimi_decode(ms, o)
        torch.cuda.synchronize()
        lat.append(o.cpu().numpy().copy()); logit.append(lg.cpu().numpy().copy()); pcm.append(p.cpu().numpy().copy())
    kv = st.export_layer(eng.L - 1, Tp).cpu().numpy()
    st.close(); ms.close()
    return np.stack(lat), np.stack(logit), np.stack(pcm), kv


@pytest.mark.parametrize("cfg_name,B,tuned", [("tiny", 3, False), ("en100m", 2, False), ("en100m", 20, True)])
def test_int8_weights_match_oracle(cfg_name, B, tuned):
    """prefill (GEMM tiles) + decode steps (K-split tiles, LN-folded and plain) with int8 weights"""
    from pocket_tts_amd.engine import Engine

    cfg, W = synth_weights(cfg_name, 0)
    eng = Engine(cfg, W, "cuda:0", quantize_groups=GROUPS)
    try:
        rng = np.random.default_rng(3)
        Tp, ns = 21, 4
        emb = (rng.standard_normal((B, Tp, eng.D)) * 0.5).astype(np.float32)
        lat, logit, _, kv = _run_gpu(eng, emb, ns, tune_batch=B if tuned else None)
        lm = O.FlowLM(cfg, O.quantized_weights(W, GROUPS))
        st = lm.init_state(B, Tp + ns)
        lm.prefill(st, emb)
        assert np.abs(st[-1]["cache"][:, :, :Tp] - kv).max() < ATOL
        x = np.full((B, lm.ldim), np.nan, np.float32)
        for i in range(ns):
            x, lg, _ = lm.decode_step(st, x, None, 1, -4.0)
            assert np.abs(x - lat[i]).max() < ATOL, i
            assert np.abs(np.asarray(lg).reshape(-1) - logit[i].reshape(-1)).max() < 1e-3, i
    finally:
        eng.close()


def test_int8_snr_against_fp32():
    """SNR of the int8-weight model against the fp32 model on the same inputs (latents of 8 AR steps and the
    decoded PCM), the reference's quality metric.  Must not be worse than what the reference's own int8 path
    reaches on these weights (golden, measured with torch.ao dynamic int8 on CPU) minus 3 dB."""
    from pocket_tts_amd.engine import Engine

    cfg, W = synth_weights("en100m", 0)
    rng = np.random.default_rng(9)
    emb = (rng.standard_normal((2, 24, 1024)) * 0.5).astype(np.float32)
    outs = []
    for groups in (None, GROUPS):
        eng = Engine(cfg, W, "cuda:0", quantize_groups=groups)
        assert eng.lm_weight_bytes() < (200e6 if groups else 400e6)
        outs.append(_run_gpu(eng, emb, 8))
        eng.close()
    snr_lat = _snr_db(outs[0][0], outs[1][0])
    snr_pcm = _snr_db(outs[0][2], outs[1][2])
    print(f"int8 weight-only vs fp32: latent SNR {snr_lat:.1f} dB, pcm SNR {snr_pcm:.1f} dB")
    floor = 20.0
    path = os.path.join(os.path.dirname(__file__), "golden", "int8_reference_snr.json")
    if os.path.exists(path):
        with open(path) as f:
            floor = json.load(f)["latent_snr_db"] - 3.0
    assert snr_lat > floor and snr_pcm > 10.0


@pytest.mark.parametrize("cfg_name,B,tuned", [("en100m", 2, False), ("en100m", 64, True)])
def test_bf16_lm_weights_vs_oracle_rounding_model(cfg_name, B, tuned):
    """PTTS_LM_BF16 (bf16 weights, bf16-rounded operands on the bf16 MFMA, everything else fp32) against
    `oracle.np_oracle.FlowLMBF16`, the CPU statement of where the build rounds.  HIP and model differ by summation order
    plus the occasional operand that rounds the other way next to a bf16 boundary, so the check is an SNR (thresholds below) next to the format's own price, the SNR of either against the fp32 ORACLE (the
    reference's metric, scripts/evaluate_quantization.py:215-228)."""
    from pocket_tts_amd.engine import Engine

    cfg, W = synth_weights(cfg_name, 0)
    eng = Engine(cfg, W, "cuda:0", quantize_groups={"lm_bf16"})
    try:
        assert eng.lm_weight_bytes() < 230e6  # transformer weights at 2 bytes (fp32: 338 MB)
        rng = np.random.default_rng(5)
        Tp, ns = 21, 4
        emb = (rng.standard_normal((B, Tp, eng.D)) * 0.5).astype(np.float32)
        lat, logit, _, kv = _run_gpu(eng, emb, ns, tune_batch=B if tuned else None)
        outs = {}
        for name, lm in (("fp32", O.FlowLM(cfg, W)), ("model", O.FlowLMBF16(cfg, W))):
            st = lm.init_state(B, Tp + ns)
            lm.prefill(st, emb)
            x = np.full((B, lm.ldim), np.nan, np.float32)
            ls, lgs = [], []
            for i in range(ns):
                x, lg, _ = lm.decode_step(st, x, None, 1, -4.0)
                ls.append(x.copy()); lgs.append(np.asarray(lg).reshape(-1).copy())
            outs[name] = (np.stack(ls), np.stack(lgs), st[-1]["cache"][:, :, :Tp].copy())
        hip_vs_model = _snr_db(outs["model"][0], lat)
        model_vs_fp32, hip_vs_fp32 = _snr_db(outs["fp32"][0], outs["model"][0]), _snr_db(outs["fp32"][0], lat)
        kv_snr = _snr_db(outs["model"][2], kv)
        print(f"B={B}: HIP bf16-LM vs rounding model {hip_vs_model:.1f} dB (KV after prefill {kv_snr:.1f} dB); "
              f"model vs fp32 oracle {model_vs_fp32:.1f} dB, HIP vs fp32 oracle {hip_vs_fp32:.1f} dB")
        # one pass through the layers (the KV cache after prefill) agrees to ~50 dB; four autoregressive steps feed every
        # rounding flip back through the whole model, so there "same arithmetic" shows as: the HIP path is closer to the
        # model than the model is to fp32, and both pay the same price against fp32 (tests/test_oracle_bf16_model.py
        # measures the same sensitivity between two CPU evaluations of one model)
        assert kv_snr > 46.0 and hip_vs_model > model_vs_fp32 + 2.0
        assert abs(hip_vs_fp32 - model_vs_fp32) < 2.0
        print(f"      EOS logits: max |HIP - model| {np.abs(outs['model'][1] - logit.reshape(ns, -1)).max():.3f} "
              f"(|model - fp32| {np.abs(outs['model'][1] - outs['fp32'][1]).max():.3f})")
    finally:
        eng.close()


def test_load_model_quantize_flag_end_to_end(tmp_path):
    """The reference's tests/test_quantization.py on this build: `load_model(quantize=True)` produces valid audio
    (not silent, finite), really switches the FlowLM attention/FFN weights to int8, and the CLI accepts --quantize."""
    import subprocess
    import sys
    from pathlib import Path

    from pocket_tts_amd import TTSModel

    G = Path(__file__).parent / "golden"
    mq = TTSModel.load_model(config=G / "e2e_tiny.yaml", temp=0.0, quantize=True)
    mb = TTSModel.load_model(config=G / "e2e_tiny.yaml", temp=0.0, quantize=False)
    try:
        assert mq.engine.quantize_groups == {"attention", "ffn"} and mb.engine.quantize_groups == frozenset()
        assert mq.engine.lm_weight_bytes() < mb.engine.lm_weight_bytes()
        state = mq.get_state_for_audio_prompt(G / "e2e_voice.safetensors")
        audio = mq.generate_audio(state, "Hello, this is a test.")
        ref = mb.generate_audio(mb.get_state_for_audio_prompt(G / "e2e_voice.safetensors"), "Hello, this is a test.")
        assert len(audio) > 0 and torch.isfinite(audio).all() and audio.abs().max() > 0
        n = min(len(audio), len(ref))
        assert n > 0 and _snr_db(ref[:n].numpy(), audio[:n].numpy()) > 5.0
    finally:
        mq.engine.close()
        mb.engine.close()
    out = tmp_path / "q.wav"
    r = subprocess.run([sys.executable, "-m", "pocket_tts_amd", "generate", "--quantize", "--config", str(G / "e2e_tiny.yaml"),
                        "--voice", str(G / "e2e_voice.safetensors"), "--text", "Hello, this is a test.",
                        "--output-path", str(out), "-q"], capture_output=True, text=True, timeout=600,
                       cwd=str(Path(__file__).parent.parent))
    assert r.returncode == 0, r.stderr[-2000:]
    assert out.exists() and out.stat().st_size > 1000
