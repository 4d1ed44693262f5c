"""Pin the numpy oracle (oracle/np_oracle.py) to the golden vectors produced by the reference's
own modules (tests/golden/gen_golden.py).  CPU only.

Tolerance: both sides are fp32 with different summation orders (ATen/mkldnn vs numpy/OpenBLAS);
we require max-abs error <= 2e-4 on O(1) quantities after up to 12 autoregressive steps and
EXACT equality of the EOS decisions at the default threshold and at the median logit.
"""

import numpy as np
import pytest

from conftest import synth_weights
from oracle import np_oracle as O

ATOL = 2e-4

CASES = ["tiny_b2", "tiny_b3_noise_lsd2", "en100m_b1", "en100m_b2_noise", "24l_b1"]


def _maxerr(a, b):
    return float(np.max(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64))))


def _run_lm(g):
    m = g["meta"]
    cfg, W = synth_weights(m["config"], m["seed"])
    lm = O.FlowLM(cfg, W)
    B = m["B"]
    st = lm.init_state(B, m["Tv"] + m["Tt"] + m["n_steps"])
    lm.prefill(st, g["voice_emb"])
    lm.prefill(st, g["text_emb"])
    kv0 = st[0]["cache"][:, :, : m["Tv"] + m["Tt"]].copy()
    kvl = st[-1]["cache"][:, :, : m["Tv"] + m["Tt"]].copy()
    x = np.full((B, lm.ldim), np.nan, np.float32)
    lat, logits = [], []
    for i in range(m["n_steps"]):
        noise = g["noise"][i] if m["with_noise"] else None
        x, logit, _ = lm.decode_step(st, x, noise, m["lsd_steps"], -4.0)
        lat.append(x.copy())
        logits.append(logit.copy())
    return lm, st, kv0, kvl, np.stack(lat), np.stack(logits)


@pytest.mark.parametrize("case", CASES)
def test_flow_lm_matches_reference(golden, case):
    g = golden(case)
    m = g["meta"]
    lm, st, kv0, kvl, lat, logits = _run_lm(g)
    assert _maxerr(kv0, g["kv_after_prefill_l0"]) < ATOL
    assert _maxerr(kvl, g["kv_after_prefill_last"]) < ATOL
    assert _maxerr(lat, g["latents"]) < ATOL
    assert _maxerr(logits, g["eos_logits"]) < 5e-4
    # EOS decision sequence: exact at the reference default and at the median logit
    for thr in (-4.0, float(np.median(g["eos_logits"]))):
        margin = np.abs(g["eos_logits"] - thr).min()
        if margin > 1e-3:
            assert np.array_equal(logits > thr, g["eos_logits"] > thr)
    assert st[0]["offset"] == int(g["offset_final"])
    last = m["Tv"] + m["Tt"] + m["n_steps"] - 1
    assert _maxerr(st[0]["cache"][:, :, last], g["kv_final_l0_last_pos"]) < ATOL


@pytest.mark.parametrize("case", ["tiny_b2", "en100m_b1"])
def test_transformer_stack_single_call(golden, case):
    g = golden(case)
    m = g["meta"]
    cfg, W = synth_weights(m["config"], m["seed"])
    lm = O.FlowLM(cfg, W)
    st = lm.init_state(m["B"], 8)
    x = g["tr_in"].astype(np.float32)
    for i in range(lm.L):
        x = O.transformer_layer(x, st[i], W, f"flow_lm.transformer.layers.{i}", lm.H, None, lm.max_period)
    assert _maxerr(x, g["tr_out"]) < ATOL


@pytest.mark.parametrize("case", ["tiny_b2", "en100m_b1"])
def test_flow_net_single_call(golden, case):
    g = golden(case)
    m = g["meta"]
    cfg, W = synth_weights(m["config"], m["seed"])
    lm = O.FlowLM(cfg, W)
    B = m["B"]
    for tag, (s, t) in {"01": (0.0, 1.0), "0h": (0.0, 0.5), "h1": (0.5, 1.0)}.items():
        o = lm.flow_net(g["fn_c"], np.full((B, 1), s, np.float32), np.full((B, 1), t, np.float32), g["fn_x"])
        assert _maxerr(o, g[f"fn_out_{tag}"]) < ATOL, tag


@pytest.mark.parametrize("case", CASES)
def test_mimi_decode_matches_reference(golden, case):
    g = golden(case)
    m = g["meta"]
    cfg, W = synth_weights(m["config"], m["seed"])
    dec = O.MimiDecoder(cfg, W)
    st = dec.init_state(m["B"], m["n_frames"])
    for f in range(m["n_frames"]):
        taps = {}
        pcm = dec.decode(st, g["mimi_latents"][f], taps)
        assert pcm.shape == (m["B"], cfg.frame_samples)
        assert _maxerr(pcm, g["pcm"][f]) < ATOL, f"frame {f}"
        if f < 3:
            for k, v in taps.items():
                gk = "tap_" + k
                if gk in g:
                    assert _maxerr(v, g[gk][f]) < ATOL, f"{k} frame {f}"


def test_golden_covers_first_frame_zero_state(golden):
    """Frame 0 exercises the zero-initialised `previous`/`partial` carries."""
    g = golden("tiny_b2")
    assert "tap_upsample" in g and "tap_seanet0" in g and g["tap_seanet0"].shape[0] == 3


@pytest.mark.parametrize("case", ["encode_tiny", "encode_en100m"])
def test_voice_encoder_matches_reference(golden, case):
    """SEANet encoder + encoder transformer + replicate-padded downsample + speaker projection against
    the reference's `MimiModel.encode_to_latent` (mimi.py:96-119)."""
    g = golden(case)
    m = g["meta"]
    cfg, W = synth_weights(m["config"], m["seed"])
    enc = O.VoiceEncoder(cfg, W)
    taps = {}
    lat = enc.encode_to_latent(g["audio"], taps)
    assert lat.shape == g["latent"].shape
    for k, v in taps.items():
        if "tap_" + k in g:
            assert _maxerr(v, g["tap_" + k]) < ATOL, k
    assert _maxerr(lat, g["latent"]) < ATOL
    assert _maxerr(enc.conditioning(g["audio"]), g["conditioning"]) < ATOL


def test_int8_weight_restatement_properties():
    """oracle/np_oracle.py::quantized_weights (the scheme the GPU int8 path implements): per-row symmetric
    int8, error <= scale / 2, idempotent, touches only the reference's "attention" and "ffn" Linear weights
    (quantization.py:60-128), and the quantised model stays close to the fp32 one."""
    cfg, W = synth_weights("tiny", 0)
    Wq = O.quantized_weights(W)
    changed = sorted(k for k in W if not np.array_equal(W[k], Wq[k]))
    L = cfg.flow_lm.transformer.num_layers
    assert len(changed) == 4 * L and all(k.startswith("flow_lm.transformer.layers.") for k in changed)
    for k in changed:
        w, q = W[k], Wq[k]
        scale = np.abs(w).max(axis=1, keepdims=True) / 127.0
        assert np.all(np.abs(w - q) <= scale * 0.5 * (1 + 1e-5))
        assert np.array_equal(O.quantize_dequantize_int8(q), q)
        assert len(np.unique(np.rint(q[0] / scale[0]))) <= 255
    only_attn = O.quantized_weights(W, ("attention",))
    assert np.array_equal(only_attn["flow_lm.transformer.layers.0.linear1.weight"], W["flow_lm.transformer.layers.0.linear1.weight"])
    # end to end: a few AR steps, fp32 vs int8 weights
    rng = np.random.default_rng(0)
    emb = (rng.standard_normal((2, 9, cfg.flow_lm.transformer.d_model)) * 0.5).astype(np.float32)
    outs = []
    for w in (W, Wq):
        lm = O.FlowLM(cfg, w)
        st = lm.init_state(2, 16)
        lm.prefill(st, emb)
        x = np.full((2, lm.ldim), np.nan, np.float32)
        seq = []
        for _ in range(4):
            x, _, _ = lm.decode_step(st, x, None, 1, -4.0)
            seq.append(x.copy())
        outs.append(np.stack(seq))
    snr = 10 * np.log10((outs[0] ** 2).mean() / ((outs[0] - outs[1]) ** 2).mean())
    assert snr > 15.0


def test_reference_int8_snr_fixture():
    """tests/golden/int8_reference_snr.json: what the reference's own dynamic-int8 path reaches against its fp32
    path on the synthetic en100m weights (made by tests/golden/gen_int8_snr.py with the reference's modules)."""
    import json
    import os

    with open(os.path.join(os.path.dirname(__file__), "golden", "int8_reference_snr.json")) as f:
        d = json.load(f)
    assert d["backend"] in ("torch.ao", "torchao") and 0 < d["latent_snr_db"] < d["first_step_latent_snr_db"] < 60
