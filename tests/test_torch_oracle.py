"""Pin the torch-CPU restatement (oracle/torch_oracle.py, the `cpu_baseline` engine of bench.py) to the same
reference-generated golden vectors as the numpy oracle.  CPU only.  Tolerance as in test_oracle_golden.py:
max-abs <= 2e-4 on latents / PCM, EOS decisions exact where the golden logit is not within 1e-3 of the threshold."""

import numpy as np
import pytest
import torch

from conftest import synth_weights
from oracle import torch_oracle as T

ATOL = 2e-4


def _maxerr(a, b):
    return float(np.max(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64))))


@pytest.mark.parametrize("case", ["tiny_b3_noise_lsd2", "en100m_b2_noise"])
def test_torch_flow_lm_matches_reference(golden, case):
    g = golden(case)
    m = g["meta"]
    cfg, W = synth_weights(m["config"], m["seed"])
    lm = T.FlowLM(cfg, W)
    B = m["B"]
    st = lm.init_state(B, m["Tv"] + m["Tt"] + m["n_steps"])
    lm.prefill(st, g["voice_emb"])
    lm.prefill(st, g["text_emb"])
    assert _maxerr(st[0]["cache"][:, :, : m["Tv"] + m["Tt"]].numpy(), g["kv_after_prefill_l0"]) < ATOL
    x = torch.full((B, lm.ldim), float("nan"))
    lat, logits = [], []
    for i in range(m["n_steps"]):
        x, logit, _ = lm.decode_step(st, x, g["noise"][i] if m["with_noise"] else None, m["lsd_steps"], -4.0)
        lat.append(x.numpy().copy())
        logits.append(logit.numpy().copy())
    lat, logits = np.stack(lat), np.stack(logits)
    assert _maxerr(lat, g["latents"]) < ATOL
    sure = np.abs(g["eos_logits"] + 4.0) > 1e-3
    assert np.array_equal((logits > -4.0)[sure], (g["eos_logits"] > -4.0)[sure])


@pytest.mark.parametrize("case", ["tiny_b2", "en100m_b1"])
def test_torch_mimi_matches_reference(golden, case):
    g = golden(case)
    m = g["meta"]
    cfg, W = synth_weights(m["config"], m["seed"])
    dec = T.MimiDecoder(cfg, W)
    ms = dec.init_state(m["B"], m["n_frames"])
    for f in range(m["n_frames"]):
        pcm = dec.decode(ms, g["mimi_latents"][f]).numpy()
        assert _maxerr(pcm, g["pcm"][f]) < ATOL, f
