"""Pins the oracle's two hot loops against the reference's `FlowLMModel._sample_next_latent` and
`TTSModel.generate_audio` end to end (fixtures: tests/golden/gen_golden_e2e.py).  CPU only."""

import ast
from pathlib import Path

import numpy as np
import pytest
import safetensors.numpy

from oracle import np_oracle as O
from pocket_tts_amd.config import load_config
from pocket_tts_amd.text import estimate_max_gen_len, prepare_text_prompt, split_into_best_sentences
from pocket_tts_amd.weights import generate_state_dict

G = Path(__file__).parent / "golden"
ATOL = 2e-4


@pytest.fixture(scope="module")
def fx():
    z = np.load(G / "e2e_tiny.npz", allow_pickle=False)
    d = {k: z[k] for k in z.files}
    d["meta"] = ast.literal_eval(str(d["meta"]))
    cfg = load_config(G / "e2e_tiny.yaml")
    W = generate_state_dict(cfg, 0)
    return d, cfg, W


def test_sample_next_latent_sequence(fx):
    d, cfg, W = fx
    lm = O.FlowLM(cfg, W)
    st = lm.init_state(1, 6 + 5 + 10)
    lm.prefill(st, d["sl_voice"])
    lm.prefill(st, lm.embed_text(d["sl_tokens"]))
    x = np.full((1, lm.ldim), np.nan, np.float32)
    for i in range(8):
        x, logit, is_eos = lm.decode_step(st, x, None, 1, -4.0)
        assert np.abs(x - d["sl_latents"][i]).max() < ATOL, i
        if abs(float(logit[0]) + 4.0) > 1e-3:
            assert bool(is_eos[0]) == bool(d["sl_eos"][i].reshape(-1)[0]), i


def oracle_generate(cfg, W, voice, text, frames_after_eos, eos_threshold=-4.0, noise_fn=None):
    """The reference's generate_audio flow (tts_model.py:603-779) on the oracle."""
    import sentencepiece

    sp = sentencepiece.SentencePieceProcessor(str(G / "e2e_sp.model"))
    enc = lambda s: sp.encode(s, out_type=int)  # noqa: E731
    lm, dec = O.FlowLM(cfg, W), O.MimiDecoder(cfg, W)
    out = []
    for chunk in split_into_best_sentences(enc, sp, text, 50, False, False):
        _, guess = prepare_text_prompt(chunk, False, False)
        fae = frames_after_eos if frames_after_eos is not None else guess + 2
        toks = np.array(enc(chunk))[None]
        gen = estimate_max_gen_len(toks.shape[1], cfg.mimi.frame_rate)
        T = voice["transformer.layers.0.self_attn/cache"].shape[2]
        st = lm.init_state(1, T + toks.shape[1] + gen)
        for i, s in enumerate(st):
            s["cache"][:, :, :T] = voice[f"transformer.layers.{i}.self_attn/cache"]
            s["offset"] = T
        lm.prefill(st, lm.embed_text(toks))
        noise = None if noise_fn is None else [noise_fn() for _ in range(gen)]
        lat, _, _ = O.autoregressive_generation(lm, st, gen, fae, noise, 1, eos_threshold)
        ms = dec.init_state(1, gen)
        for z in lat:
            out.append(dec.decode(ms, z)[0])
    return np.concatenate(out)


def test_generate_audio_end_to_end(fx):
    d, cfg, W = fx
    voice = safetensors.numpy.load_file(str(G / "e2e_voice.safetensors"))
    wav = oracle_generate(cfg, W, voice, d["meta"]["text"], 2)
    assert wav.shape == d["e2e_wav_temp0"].shape  # exact frame count (EOS decisions)
    assert np.abs(wav - d["e2e_wav_temp0"]).max() < ATOL


def test_generate_audio_without_eos_hits_max_len(fx):
    d, cfg, W = fx
    voice = safetensors.numpy.load_file(str(G / "e2e_voice.safetensors"))
    wav = oracle_generate(cfg, W, voice, "ok", 1, eos_threshold=1e9)
    assert wav.shape == d["e2e_wav_noeos"].shape
    assert np.abs(wav - d["e2e_wav_noeos"]).max() < 5e-4  # 50 autoregressive frames
