#!/usr/bin/env python3
"""Round-2 additions to the end-to-end golden vectors (dev container only; same import arrangement as
gen_golden_e2e.py: the UNMODIFIED reference source, three empty in-memory `beartype` module objects).

    python tests/golden/gen_golden_e2e2.py     # writes tests/golden/e2e2_*.npz / .yaml

What is recorded (data only):
  * en100m (full-size, 6 layers) `FlowLMModel._sample_next_latent` through the reference's own
    `TTSModel._run_flow_lm_and_increment_step` (voice conditioning prefill, text-token prefill, 8 chained steps,
    temp 0 and temp 0.7 seeded): VERDICT r1 weak #1 - the full-size FlowLM latents of gen_golden.py go through
    builder-written glue, these do not;
  * tiny `TTSModel.generate_audio` with `noise_clamp=0.8`, temp 0.7, seeded (flow_lm.py:136-137 trunc_normal_).
"""

from __future__ import annotations

import sys
from pathlib import Path

import numpy as np
import torch
import yaml

HERE = Path(__file__).parent
sys.path.insert(0, str(HERE))
import gen_golden_e2e as G1  # noqa: E402  (install_noop_beartype, REPO / REF paths, sys.path for the build's package)

from pocket_tts_amd.config import config_to_dict, named_config  # noqa: E402
from pocket_tts_amd.weights import generate_tensor  # noqa: E402

OUT = HERE


def load_ref_model(R, yml, **kw):
    model = R.TTSModel.load_model(config=str(yml), **kw)
    sd = model.state_dict()
    model.load_state_dict({k: torch.from_numpy(generate_tensor(k, tuple(v.shape), 0)) for k, v in sd.items()}, strict=True)
    model.eval()
    return model


@torch.no_grad()
def main():
    G1.install_noop_beartype()
    sys.path.insert(0, str(G1.REF))
    torch.set_num_threads(8)
    import sentencepiece

    from pocket_tts.models import tts_model as R  # the reference, unmodified
    from pocket_tts.modules.stateful_module import init_states

    vocab = sentencepiece.SentencePieceProcessor(str(OUT / "e2e_sp.model")).vocab_size()

    # ---- en100m: the real FlowLMModel glue at full size
    if not (OUT / "e2e2_en100m.npz").exists() or "--force" in sys.argv:
        gen_en100m(R, init_states, vocab)
    gen_noise_clamp(R)


def gen_en100m(R, init_states, vocab):
    d = config_to_dict(named_config("en100m"))
    d["flow_lm"]["lookup_table"]["n_bins"] = vocab
    d["flow_lm"]["lookup_table"]["tokenizer_path"] = str(OUT / "e2e_sp.model")
    yml = OUT / "e2e2_en100m.yaml"
    yml.write_text(yaml.safe_dump(d))
    out = {}
    for tag, temp in (("t0", 0.0), ("t07", 0.7)):
        model = load_ref_model(R, yml, temp=temp)
        g = torch.Generator().manual_seed(9)
        B, Tv, Tt, ns = 1, 9, 7, 8
        dm = 1024
        voice = torch.randn(B, Tv, dm, generator=g) * 0.1
        tokens = torch.randint(0, vocab, (B, Tt), generator=g)
        st = init_states(model.flow_lm, B, Tv + Tt + ns + 2)
        torch.manual_seed(4321)
        model._run_flow_lm_and_increment_step(model_state=st, audio_conditioning=voice)
        model._run_flow_lm_and_increment_step(model_state=st, text_tokens=tokens)
        x = torch.full((B, 1, 32), float("nan"))
        lat, eos = [], []
        for i in range(ns):
            x, is_eos = model._run_flow_lm_and_increment_step(model_state=st, backbone_input_latents=x)
            lat.append(x[:, 0].numpy().copy())
            eos.append(is_eos.numpy().copy())
        out["voice"], out["tokens"] = voice.numpy(), tokens.numpy()
        out[f"latents_{tag}"], out[f"eos_{tag}"] = np.stack(lat), np.stack(eos)
        del model
    # the temp-0.7 run drew its noise from torch's global generator seeded with 4321: one draw per forward call,
    # including the two prefill calls (flow_lm.py:131-137 runs for every forward)
    out["meta"] = np.array(repr(dict(config="en100m", vocab=vocab, Tv=9, Tt=7, n_steps=8, seed_noise=4321)))
    np.savez_compressed(OUT / "e2e2_en100m.npz", **out)
    rel = dict(d)
    rel["flow_lm"]["lookup_table"]["tokenizer_path"] = "e2e_sp.model"
    yml.write_text(yaml.safe_dump(rel))


def gen_noise_clamp(R):
    # ---- tiny: noise_clamp end to end
    model = load_ref_model(R, OUT / "e2e_tiny.yaml", temp=0.7, noise_clamp=0.8)
    # e2e_tiny.yaml holds a tokenizer path relative to the fixture directory
    vstate = model.get_state_for_audio_prompt(str(OUT / "e2e_voice.safetensors"))
    text = "Hello world. This is a test, of the pocket system!"
    torch.manual_seed(777)
    wav = model.generate_audio(vstate, text, frames_after_eos=10)
    np.savez_compressed(OUT / "e2e2_noise_clamp.npz", wav=wav.numpy(),
                        meta=np.array(repr(dict(text=text, seed=777, temp=0.7, noise_clamp=0.8, frames_after_eos=10,
                                                frames=len(wav) // 1920))))
    print("noise_clamp frames", len(wav) // 1920)


if __name__ == "__main__":
    main()
