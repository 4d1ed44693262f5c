#!/usr/bin/env python3
"""End-to-end golden vectors from the reference's own `TTSModel` / `FlowLMModel` (dev container only).

    python tests/golden/gen_golden_e2e.py     # writes tests/golden/e2e_*.{npz,json,safetensors,model}

`pocket_tts/__init__.py`, `models/flow_lm.py` and `data/audio.py` import `beartype`, a runtime type
checker that performs no arithmetic and is not installed here.  This script
registers three EMPTY in-memory module objects under the names `beartype`, `beartype.claw` and
`beartype.typing` (no-op `beartype_this_package`, `typing` re-exports) so that the UNMODIFIED reference
source imports; no file pretending to be that library is written anywhere.  The module-level fixtures of
gen_golden.py do not depend on this and stand on their own.

What is recorded (data only):
  * text pipeline: inputs -> outputs of the reference's `prepare_text_prompt`, `split_into_best_sentences`,
    `TTSModel._estimate_max_gen_len` with a sentencepiece model trained here on a built-in corpus;
  * `FlowLMModel._sample_next_latent` driven like `_run_flow_lm_and_increment_step`: latents, EOS flags;
  * `TTSModel.generate_audio` end to end on synthetic weights: a voice state exported with the
    reference's `export_model_state`, the waveform at temp 0 and at temp 0.7 (seeded), frame counts.
"""

from __future__ import annotations

import json
import sys
import types
import typing
from pathlib import Path

import numpy as np
import torch
import yaml

REPO = Path(__file__).resolve().parents[2]
REF = Path("/root/reference")
OUT = Path(__file__).parent
sys.path.insert(0, str(REPO))

from pocket_tts_amd.config import config_to_dict, named_config  # noqa: E402
from pocket_tts_amd.weights import generate_tensor  # noqa: E402

CORPUS = """Hello world. I am a pocket sized text to speech system, and I run on small machines!
The quick brown fox jumps over the lazy dog; then it sleeps: quietly, calmly, happily.
How are you today? Fine, thanks... This is a longer sentence, with several clauses, to test splitting.
Numbers like 1, 2 and 3 appear too. Short one. Another short one! Is this a question? Yes it is.
We are testing the tokenizer with enough different words so that the vocabulary can be trained."""

TEXTS = [
    "hello world",
    "Hello world. I am a pocket sized text to speech system, and I run on small machines!",
    "  this has\nnewlines and   spaces; and a semicolon",
    "Short one. Another short one! Is this a question? Yes it is. " * 3,
    "A very long sentence without any full stop, but with commas, so that it must be split on commas, "
    "because it exceeds the maximum number of tokens in a chunk, which is quite small here, right",
    "ok",
    "one two three four five six",
]


def install_noop_beartype():
    bt = types.ModuleType("beartype")

    class BeartypeConf:
        def __init__(self, **kw):
            pass

    bt.BeartypeConf = BeartypeConf
    claw = types.ModuleType("beartype.claw")
    claw.beartype_this_package = lambda conf=None: None
    typ = types.ModuleType("beartype.typing")
    for k in dir(typing):
        if not k.startswith("_"):
            setattr(typ, k, getattr(typing, k))
    sys.modules.update({"beartype": bt, "beartype.claw": claw, "beartype.typing": typ})


def train_tokenizer(n_bins: int, path: Path):
    import sentencepiece as spm

    corpus = OUT / "_corpus.txt"
    corpus.write_text((CORPUS + "\n") * 20)
    spm.SentencePieceTrainer.train(input=str(corpus), model_prefix=str(path.with_suffix("")), vocab_size=n_bins,
                                   model_type="bpe", character_coverage=1.0, hard_vocab_limit=False,
                                   minloglevel=2)
    corpus.unlink()
    path.with_suffix(".vocab").unlink(missing_ok=True)


@torch.no_grad()
def main():
    install_noop_beartype()
    sys.path.insert(0, str(REF))
    torch.set_num_threads(4)
    cfg = named_config("tiny")
    n_bins = 128
    d = config_to_dict(cfg)
    d["flow_lm"]["lookup_table"]["n_bins"] = n_bins
    sp_path = OUT / "e2e_sp.model"
    train_tokenizer(n_bins, sp_path)
    import sentencepiece

    vocab = sentencepiece.SentencePieceProcessor(str(sp_path)).vocab_size()
    d["flow_lm"]["lookup_table"]["n_bins"] = vocab
    d["flow_lm"]["lookup_table"]["tokenizer_path"] = str(sp_path)
    yml = OUT / "e2e_tiny.yaml"
    rel = dict(d)
    yml.write_text(yaml.safe_dump(d))

    from pocket_tts.models import tts_model as R  # the reference, unmodified

    model = R.TTSModel.load_model(config=str(yml), temp=0.0)
    sd = model.state_dict()
    new = {k: torch.from_numpy(generate_tensor(k, tuple(v.shape), 0)) for k, v in sd.items()}
    model.load_state_dict(new, strict=True)
    model.eval()

    # ---- text pipeline fixtures
    tok = model.flow_lm.conditioner.tokenizer
    text_fx = []
    for t in TEXTS:
        for pad in (False, True):
            for semi in (False, True):
                p, guess = R.prepare_text_prompt(t, pad, semi)
                chunks = R.split_into_best_sentences(tok, t, 12, pad, semi)
                chunks50 = R.split_into_best_sentences(tok, t, 50, pad, semi)
                text_fx.append(dict(text=t, pad=pad, semi=semi, prepared=p, guess=guess, chunks12=chunks,
                                    chunks50=chunks50, tokens=tok(p).tokens[0].tolist()))
    gen_len = {str(n): model._estimate_max_gen_len(n) for n in (1, 3, 7, 20, 50, 58)}
    try:
        R.prepare_text_prompt("   ", False, False)
        empty_raises = False
    except ValueError:
        empty_raises = True
    (OUT / "e2e_text.json").write_text(json.dumps(dict(cases=text_fx, gen_len=gen_len, empty_raises=empty_raises,
                                                       vocab=vocab), indent=1))

    # ---- FlowLMModel._sample_next_latent (pins the glue of flow_lm.py:96-157)
    out = {}
    g = torch.Generator().manual_seed(5)
    B = 1
    dm = cfg.flow_lm.transformer.d_model
    voice = torch.randn(B, 6, dm, generator=g) * 0.1
    text_tokens = torch.randint(0, vocab, (B, 5), generator=g)
    from pocket_tts.modules.stateful_module import init_states

    st = init_states(model.flow_lm, B, 6 + 5 + 10)
    model._run_flow_lm_and_increment_step(model_state=st, audio_conditioning=voice)
    model._run_flow_lm_and_increment_step(model_state=st, text_tokens=text_tokens)
    x = torch.full((B, 1, 32), float("nan"))
    lat, eos = [], []
    for i in range(8):
        x, is_eos = model._run_flow_lm_and_increment_step(model_state=st, backbone_input_latents=x)
        lat.append(x[:, 0].numpy().copy())
        eos.append(is_eos.numpy().copy())
    out["sl_voice"], out["sl_tokens"] = voice.numpy(), text_tokens.numpy()
    out["sl_latents"], out["sl_eos"] = np.stack(lat), np.stack(eos)

    # ---- end to end: voice state from an audio tensor, exported, then generate_audio
    audio = torch.randn(1, 24000 * 1, generator=g) * 0.1  # 1 s of noise as the "voice"
    out["e2e_audio"] = audio.numpy()
    vstate = model.get_state_for_audio_prompt(audio)
    R.export_model_state(vstate, OUT / "e2e_voice.safetensors")
    text = "Hello world. This is a test, of the pocket system!"
    wav0 = model.generate_audio(vstate, text, frames_after_eos=2)
    out["e2e_wav_temp0"] = wav0.numpy()
    model.temp = 0.7
    torch.manual_seed(1234)
    wav7 = model.generate_audio(vstate, text, frames_after_eos=None)
    out["e2e_wav_temp07"] = wav7.numpy()
    model.temp = 0.0
    model.eos_threshold = 1e9  # never EOS -> max_gen_len frames, warning path (tts_model.py:770-775)
    wavn = model.generate_audio(vstate, "ok", frames_after_eos=1)
    out["e2e_wav_noeos"] = wavn.numpy()
    out["meta"] = np.array(repr(dict(text=text, seed_temp07=1234, vocab=vocab, frames0=len(wav0) // 1920,
                                     frames7=len(wav7) // 1920, framesn=len(wavn) // 1920)))
    np.savez_compressed(OUT / "e2e_tiny.npz", **out)
    # the yaml is a fixture too: make the tokenizer path relative to the fixture directory
    rel["flow_lm"]["lookup_table"]["tokenizer_path"] = "e2e_sp.model"
    rel["flow_lm"]["lookup_table"]["n_bins"] = vocab
    yml.write_text(yaml.safe_dump(rel))
    print("frames temp0", len(wav0) // 1920, "temp0.7", len(wav7) // 1920, "noeos", len(wavn) // 1920, "vocab", vocab)


if __name__ == "__main__":
    main()
