"""Round-2 parity additions, all through the C ABI on the GPU (`-m gpu`):

  * the Mimi KV ring (272 slots) and the 250-position window at the REAL size: en100m, 20 frames (320 positions)
    against the numpy oracle, which keeps the reference's linear cache + mask;
  * en100m FlowLM latents against `FlowLMModel._sample_next_latent` driven by the reference's own `TTSModel`
    (fixture tests/golden/e2e2_en100m.npz; no builder-written glue on the reference side), temp 0 and seeded temp 0.7;
  * `noise_clamp` (`trunc_normal_`, flow_lm.py:136-137): seeded end-to-end waveform of the reference;
  * batch 64 (BASELINE config #3) x 2 steps and 24-layer batch 32 (config #4, per GPU) against the oracle / through
    properties;
  * `generate_audio_batch` and `ContinuousBatcher` against `oracle.autoregressive_generation` + the oracle codec
    (not against the build's own single-utterance path);
  * tests/hip/test_kernels.hip (each kernel against plain C++ loops) built and run as a child process.

Tolerances as in test_gpu_parity.py: latents / PCM max-abs <= 2e-4, EOS logits <= 1e-3, EOS decisions and frame counts exact.
"""

import ast
import shutil
import subprocess
from pathlib import Path

import numpy as np
import pytest
import torch

from conftest import synth_weights
from test_gpu_parity import ATOL, _maxerr, dev, get_engine

pytestmark = pytest.mark.gpu
G = Path(__file__).parent / "golden"


def _npz(name):
    z = np.load(G / name, allow_pickle=False)
    d = {k: z[k] for k in z.files}
    d["meta"] = ast.literal_eval(str(d["meta"]))
    return d


def test_mimi_ring_wrap_and_window_at_real_size():
    """en100m codec, B = 2, 20 frames = 320 positions: the 272-slot ring wraps after frame 17 and the 250-key window
    drops keys from frame 16 on; the oracle keeps a linear cache of 320 positions and the reference mask."""
    from oracle import np_oracle as O

    cfg, W = synth_weights("en100m")
    eng = get_engine("en100m")
    dec = O.MimiDecoder(cfg, W)
    B, nf = 2, 20
    rng = np.random.default_rng(77)
    lat = rng.standard_normal((nf, B, eng.ldim)).astype(np.float32)
    ost, ms = dec.init_state(B, nf), eng.new_mimi_state(B)
    for f in range(nf):
        ref = dec.decode(ost, lat[f])
        got = eng.mimi_decode(ms, dev(lat[f]))
        torch.cuda.synchronize()
        assert _maxerr(got.cpu().numpy(), ref) < ATOL, f"frame {f}"
    ms.close()


def _e2e2_engine():
    from pocket_tts_amd.config import load_config
    from pocket_tts_amd.engine import Engine
    from pocket_tts_amd.weights import generate_state_dict

    cfg = load_config(G / "e2e2_en100m.yaml")  # en100m with the fixture tokenizer's vocabulary
    return cfg, Engine(cfg, generate_state_dict(cfg, 0), "cuda:0")


def test_en100m_latents_match_reference_sample_next_latent():
    d = _npz("e2e2_en100m.npz")
    m = d["meta"]
    cfg, eng = _e2e2_engine()
    try:
        for tag, temp in (("t0", 0.0), ("t07", 0.7)):
            st = eng.new_lm_state(1, m["Tv"] + m["Tt"] + m["n_steps"] + 2)
            torch.manual_seed(m["seed_noise"])
            draw = lambda: torch.nn.init.normal_(torch.empty(1, eng.ldim), mean=0.0, std=temp ** 0.5)  # noqa: E731
            eng.lm_prefill(st, dev(d["voice"]))
            eng.lm_prefill(st, eng.embed_text(torch.from_numpy(d["tokens"])))
            if temp > 0:
                draw(), draw()  # the reference's two prefill forwards each consume one draw (flow_lm.py:131-137)
            for i in range(m["n_steps"]):
                noise = draw().to("cuda:0") if temp > 0 else None
                o, lg, fl = eng.lm_decode_step(st, None, noise, 1, -4.0)
                torch.cuda.synchronize()
                assert _maxerr(o.cpu().numpy(), d[f"latents_{tag}"][i]) < ATOL, (tag, i)
                if abs(float(lg[0]) + 4.0) > 1e-3:
                    assert bool(fl[0]) == bool(d[f"eos_{tag}"][i].reshape(-1)[0]), (tag, i)
            st.close()
    finally:
        eng.close()


def test_noise_clamp_seeded_end_to_end():
    """TTSModel(noise_clamp=0.8, temp=0.7), seed 777: the reference's waveform, exact frame count."""
    from pocket_tts_amd import TTSModel

    d = _npz("e2e2_noise_clamp.npz")
    m = d["meta"]
    model = TTSModel.load_model(config=G / "e2e_tiny.yaml", temp=m["temp"], noise_clamp=m["noise_clamp"])
    try:
        state = model.get_state_for_audio_prompt(G / "e2e_voice.safetensors")
        torch.manual_seed(m["seed"])
        wav = model.generate_audio(state, m["text"], frames_after_eos=m["frames_after_eos"]).numpy()
        assert wav.shape == d["wav"].shape
        assert _maxerr(wav, d["wav"]) < 5e-4  # 11 autoregressive frames with noise
    finally:
        model.engine.close()


def test_batch64_two_steps_vs_oracle():
    """BASELINE config #3 shapes (batch 64, 64 distinct rows) against the numpy oracle: FlowLM latents, EOS logits,
    PCM of two steps after a short prefill."""
    from oracle import np_oracle as O

    cfg, W = synth_weights("en100m")
    eng = get_engine("en100m")
    eng.tune(64)
    B, Tp, ns = 64, 24, 2
    rng = np.random.default_rng(64)
    emb = (rng.standard_normal((B, Tp, eng.D)) * 0.3).astype(np.float32)
    noise = (rng.standard_normal((ns, B, eng.ldim)) * 0.8).astype(np.float32)
    lm, dec = O.FlowLM(cfg, W), O.MimiDecoder(cfg, W)
    ost, oms = lm.init_state(B, Tp + ns), dec.init_state(B, ns)
    lm.prefill(ost, emb)
    st, ms = eng.new_lm_state(B, Tp + ns), eng.new_mimi_state(B)
    eng.lm_prefill(st, dev(emb))
    xo = np.full((B, eng.ldim), np.nan, np.float32)
    for i in range(ns):
        xo, lo, _ = lm.decode_step(ost, xo, noise[i], 1, -4.0)
        po = dec.decode(oms, xo)
        xg, lg, _ = eng.lm_decode_step(st, None, dev(noise[i]), 1, -4.0)
        pg = eng.mimi_decode(ms, xg)
        torch.cuda.synchronize()
        assert _maxerr(xg.cpu().numpy(), xo) < ATOL, i
        assert _maxerr(lg.cpu().numpy(), lo) < 1e-3, i
        assert _maxerr(pg.cpu().numpy(), po) < ATOL, i
    st.close(); ms.close()


def test_24l_batch32_properties():
    """BASELINE config #4 per GPU (24-layer model, 32 utterances): permutation equivariance is bitwise, row 0 agrees
    with a batch-1 run, three steps each (the per-layer golden for this model is golden_24l_b1)."""
    eng = get_engine("24l")
    rng = np.random.default_rng(24)
    B, Tp, ns = 32, 40, 3
    emb = (rng.standard_normal((B, Tp, eng.D)) * 0.3).astype(np.float32)

    def run(e):
        b = e.shape[0]
        st, ms = eng.new_lm_state(b, Tp + ns + 1), eng.new_mimi_state(b)
        eng.lm_prefill(st, dev(e))
        outs = []
        for _ in range(ns):
            o, lg, _ = eng.lm_decode_step(st, None, None, 1, -4.0)
            p = eng.mimi_decode(ms, o)
            torch.cuda.synchronize()
            outs.append((o.cpu().numpy().copy(), lg.cpu().numpy().reshape(-1).copy(), p.cpu().numpy().copy()))
        assert not st.error()
        st.close(); ms.close()
        return outs

    perm = rng.permutation(B)
    a, b, one = run(emb), run(emb[perm]), run(emb[:1])
    for (o, lg, p), (o2, lg2, p2), (o1, lg1, p1) in zip(a, b, one):
        assert np.isfinite(o).all() and np.abs(o[0] - o[1]).max() > 1e-4
        assert np.array_equal(o[perm], o2) and np.array_equal(p[perm], p2) and np.array_equal(lg[perm], lg2)
        assert _maxerr(o[0], o1[0]) < ATOL and _maxerr(p[0], p1[0]) < ATOL and _maxerr(lg[0], lg1[0]) < 1e-3


# ---- batched generation against the ORACLE's generation loop -----------------------------------------------------
@pytest.fixture(scope="module")
def tiny_model():
    from pocket_tts_amd import TTSModel

    m = TTSModel.load_model(config=G / "e2e_tiny.yaml", temp=0.0)
    yield m
    m.engine.close()


def _oracle_waveforms(texts, frames_after_eos):
    import safetensors.numpy

    from pocket_tts_amd.config import load_config
    from pocket_tts_amd.weights import generate_state_dict
    from test_oracle_e2e import oracle_generate

    cfg = load_config(G / "e2e_tiny.yaml")
    W = generate_state_dict(cfg, 0)
    voice = safetensors.numpy.load_file(str(G / "e2e_voice.safetensors"))
    return [oracle_generate(cfg, W, voice, t, frames_after_eos) for t in texts]


TEXTS = ["Hello world.", "This is a test, of the pocket system!", "one two three four five six", "ok"]


def test_generate_audio_batch_vs_oracle_generation(tiny_model):
    """rows of different text lengths decoded in lock-step == the oracle's `autoregressive_generation` per utterance
    (reference loop tts_model.py:744-779): exact frame counts, waveform within tolerance"""
    state = tiny_model.get_state_for_audio_prompt(G / "e2e_voice.safetensors")
    want = _oracle_waveforms(TEXTS, 2)
    got = tiny_model.generate_audio_batch(state, TEXTS, frames_after_eos=2)
    for w, g in zip(want, got):
        assert g.shape[0] == w.shape[0]
        assert _maxerr(g.numpy(), w) < 5e-4


def test_continuous_batcher_vs_oracle_generation(tiny_model):
    """7 requests through 3 slots (joins and leaves mid-flight) == the oracle per request"""
    from pocket_tts_amd.batching import ContinuousBatcher

    state = tiny_model.get_state_for_audio_prompt(G / "e2e_voice.safetensors")
    texts = (TEXTS * 2)[:7]
    want = _oracle_waveforms(texts, 2)
    cb = ContinuousBatcher(tiny_model, slots=3, capacity=256)
    try:
        reqs = [cb.submit(state, t, frames_after_eos=2) for t in texts]
        cb.run_until_idle()
        for r, w in zip(reqs, want):
            g = r.result().numpy()
            assert g.shape[0] == w.shape[0]
            assert _maxerr(g, w) < 5e-4
    finally:
        cb.close()


def test_batcher_failure_paths(tiny_model):
    """a request that fails during admission (malformed voice state) gets an exception instead of blocking
    its consumer forever, the scheduler keeps serving the others, and close() releases whatever is still pending."""
    from pocket_tts_amd.batching import ContinuousBatcher

    state = tiny_model.get_state_for_audio_prompt(G / "e2e_voice.safetensors")
    bad = {k: dict(v) for k, v in state.items()}
    first = next(iter(bad))
    bad[first]["cache"] = bad[first]["cache"][:, :, :, :1]  # wrong head count: the import must reject it
    cb = ContinuousBatcher(tiny_model, slots=2, capacity=256)
    cb.start()
    try:
        r_bad = cb.submit(bad, "Hello world.", frames_after_eos=1)
        r_ok = cb.submit(state, "Hello world.", frames_after_eos=1)
        with pytest.raises((ValueError, RuntimeError, KeyError, IndexError, TypeError)):
            r_bad.result()
        assert r_ok.result().shape[0] > 0
        # pending work when the batcher is closed: the consumer is released with an error
        cb.stop()
        r_late = cb.submit(state, "This request is never scheduled.", frames_after_eos=1)
    finally:
        cb.close()
    with pytest.raises(RuntimeError):
        r_late.result()
    with pytest.raises(RuntimeError):
        cb.submit(state, "after close", frames_after_eos=1)


def test_hip_kernel_unit_tests(tmp_path):
    """tests/hip/test_kernels.hip: every kernel against plain C++ loops (built with hipcc here, run as a child)"""
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    src = Path(__file__).parent / "hip" / "test_kernels.hip"
    exe = tmp_path / "test_kernels"
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O2", "-std=c++17", "-ffp-contract=on", "-o", str(exe), str(src)],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-2000:])


def test_pipeline_streams_run_concurrently():
    """HIP maps streams round-robin onto a few hardware queues; a FlowLM / codec stream pair on ONE queue serialises the
    pipeline.  `Engine.concurrent_stream` must return a stream that overlaps with the FlowLM stream, whatever the number of
    streams created before (here: eight pipelines' worth), and the probe itself must see both outcomes' timing scale."""
    eng = get_engine("tiny")
    assert not eng.streams_overlap(eng.stream, eng.stream)
    seen = []
    keep = []
    for _ in range(8):
        s = eng.concurrent_stream(eng.stream)
        keep.append(s)
        seen.append(eng.streams_overlap(eng.stream, s))
    assert all(seen), seen
