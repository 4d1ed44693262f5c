"""End-to-end drop-in check on the GPU: `pocket_tts_amd.TTSModel` against waveforms produced by the
reference's `TTSModel.generate_audio` on the same synthetic weights, voice-state file, tokenizer and
text (fixtures: tests/golden/gen_golden_e2e.py).  Frame counts (= EOS decision sequence) must be EXACT;
waveform max-abs error <= 5e-4."""

import ast
from pathlib import Path

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
G = Path(__file__).parent / "golden"


@pytest.fixture(scope="module")
def fx():
    z = np.load(G / "e2e_tiny.npz", allow_pickle=False)
    d = {k: z[k] for k in z.files}
    d["meta"] = ast.literal_eval(str(d["meta"]))
    return d


@pytest.fixture(scope="module")
def model():
    from pocket_tts_amd import TTSModel

    m = TTSModel.load_model(config=G / "e2e_tiny.yaml", temp=0.0)
    yield m
    m.engine.close()


def test_public_api():
    import pocket_tts_amd

    assert pocket_tts_amd.__all__ == ["TTSModel", "export_model_state"]
    for name in ("load_model", "generate_audio", "generate_audio_stream", "get_state_for_audio_prompt"):
        assert hasattr(pocket_tts_amd.TTSModel, name)


def test_generate_audio_temp0(model, fx):
    state = model.get_state_for_audio_prompt(G / "e2e_voice.safetensors")
    wav = model.generate_audio(state, fx["meta"]["text"], frames_after_eos=2)
    assert wav.dtype == torch.float32 and wav.dim() == 1
    assert wav.shape[0] == fx["e2e_wav_temp0"].shape[0]
    assert np.abs(wav.numpy() - fx["e2e_wav_temp0"]).max() < 5e-4
    assert model.sample_rate == 24000 and model.device.type == "cuda"


def test_generate_audio_stream_chunks_and_seeded_noise(model, fx):
    """temp 0.7: noise is drawn from torch's global CPU generator exactly like the reference CPU path,
    so a seeded run reproduces the reference waveform."""
    state = model.get_state_for_audio_prompt(G / "e2e_voice.safetensors")
    model.temp = 0.7
    torch.manual_seed(fx["meta"]["seed_temp07"])
    chunks = list(model.generate_audio_stream(state, fx["meta"]["text"]))
    model.temp = 0.0
    assert all(c.shape == (1920,) for c in chunks)
    wav = torch.cat(chunks).numpy()
    assert wav.shape == fx["e2e_wav_temp07"].shape
    assert np.abs(wav - fx["e2e_wav_temp07"]).max() < 5e-4


def test_max_length_without_eos(model, fx):
    state = model.get_state_for_audio_prompt(G / "e2e_voice.safetensors")
    thr = model.eos_threshold
    model.eos_threshold = 1e9
    try:
        wav = model.generate_audio(state, "ok", frames_after_eos=1)
    finally:
        model.eos_threshold = thr
    assert wav.shape[0] == fx["e2e_wav_noeos"].shape[0]
    assert np.abs(wav.numpy() - fx["e2e_wav_noeos"]).max() < 1e-3


def test_state_is_not_mutated_with_copy_state(model, fx):
    state = model.get_state_for_audio_prompt(G / "e2e_voice.safetensors")
    before = {k: v["cache"].clone() for k, v in state.items()}
    model.generate_audio(state, "hello world", frames_after_eos=1)
    for k, v in state.items():
        assert torch.equal(v["cache"], before[k])


def test_export_import_voice_state_roundtrip(model, tmp_path):
    from pocket_tts_amd import export_model_state

    cond = torch.randn(1, 7, model.engine.D) * 0.1
    st = model.get_state_for_conditioning(cond)
    assert int(st["transformer.layers.0.self_attn"]["offset"][0]) == 8  # bos_before_voice + 7
    export_model_state(st, tmp_path / "v.safetensors")
    st2 = model.get_state_for_audio_prompt(tmp_path / "v.safetensors")
    for k in st:
        assert torch.equal(st[k]["cache"].cpu(), st2[k]["cache"].cpu())
        assert torch.equal(st[k]["offset"].cpu(), st2[k]["offset"].cpu())


def test_errors_mirror_reference(model):
    from pocket_tts_amd import TTSModel

    with pytest.raises(ValueError):
        TTSModel.load_model(language="english", config="x.yaml")
    with pytest.raises(ValueError):
        TTSModel.load_model(language="french")
    with pytest.raises(FileNotFoundError):
        TTSModel.load_model(config="/nonexistent/none.yaml")
    state = model.get_state_for_audio_prompt(G / "e2e_voice.safetensors")
    with pytest.raises(ValueError):
        model.generate_audio(state, "   ")


def test_cli_generate_writes_wav(tmp_path):
    """drop-in CLI surface: exit code 0, mono 24 kHz 16-bit WAV, non-empty (reference tests/test_cli_generate.py)"""
    import wave

    from pocket_tts_amd.main import cli_app

    out = tmp_path / "o.wav"
    rc = cli_app(["generate", "--config", str(G / "e2e_tiny.yaml"), "--voice", str(G / "e2e_voice.safetensors"),
                  "--text", "Hello world. This is a test.", "--temperature", "0", "--output-path", str(out), "-q"])
    assert rc == 0 and out.exists()
    with wave.open(str(out), "rb") as w:
        assert w.getnchannels() == 1 and w.getframerate() == 24000 and w.getsampwidth() == 2
        n = w.getnframes()
    assert n >= 1920 + 4800  # at least one frame + 200 ms of silence


def test_voice_state_from_audio_matches_reference(model, fx):
    """`get_state_for_audio_prompt(audio tensor)`: Mimi encoder + speaker projection + bos_before_voice +
    prefill on the GPU against the voice state the reference produced from the same audio and exported with
    `export_model_state` (tts_model.py:874-899)."""
    import safetensors.torch

    ref = safetensors.torch.load_file(str(G / "e2e_voice.safetensors"))
    state = model.get_state_for_audio_prompt(torch.from_numpy(fx["e2e_audio"]))
    for name, st in state.items():
        rc = ref[f"{name}/cache"]
        T = int(st["offset"][0])
        assert T == int(ref[f"{name}/offset"][0]) == 14  # 13 frames + bos_before_voice
        assert np.abs(st["cache"][:, :, :T].cpu().numpy() - rc[:, :, :T].numpy()).max() < 2e-4
    # and the state drives generation exactly like the file-loaded one
    wav = model.generate_audio(state, fx["meta"]["text"], frames_after_eos=2)
    assert wav.shape[0] == fx["e2e_wav_temp0"].shape[0]
    assert np.abs(wav.numpy() - fx["e2e_wav_temp0"]).max() < 5e-4


def test_voice_state_from_wav_file(model, fx, tmp_path):
    import wave

    pcm = (np.clip(fx["e2e_audio"][0], -1, 1) * 32767).astype(np.int16)
    with wave.open(str(tmp_path / "v.wav"), "wb") as w:
        w.setnchannels(1); w.setsampwidth(2); w.setframerate(24000); w.writeframes(pcm.tobytes())
    state = model.get_state_for_audio_prompt(tmp_path / "v.wav")
    assert int(state["transformer.layers.0.self_attn"]["offset"][0]) == 14


TEXTS = ["Hello world. This is a test.", "ok", "This is a longer sentence, with several clauses, to test it.",
         "How are you today?", "Short one.", "Another request arrives while the others are running.", "Yes."]


def test_generate_audio_batch_matches_single_utterances(model, fx):
    """Mixed-length batch (different texts, hence different prompt lengths and per-row cache offsets, different
    EOS steps and frame counts) against one-by-one generation at temp 0."""
    state = model.get_state_for_audio_prompt(G / "e2e_voice.safetensors")
    texts = ["Hello world. This is a test.", "ok", "This is a longer sentence, with several clauses, to test it.",
             "How are you today?", "Short one."]
    singles = [model.generate_audio(state, t) for t in texts]
    batch = model.generate_audio_batch(state, texts)
    assert len(batch) == len(texts)
    lens = [w.shape[0] for w in singles]
    assert len(set(lens)) > 1  # the rows really end at different frames
    for t, a, b in zip(texts, singles, batch):
        assert a.shape == b.shape, (t, a.shape, b.shape)
        assert np.abs(a.numpy() - b.numpy()).max() < 5e-4, t


def test_two_voices_interleaved_share_prefixes_and_match_single_utterances(model, fx):
    """20 requests alternating between TWO voices, in one `generate_audio_batch` call and through 16 batcher slots:
    rows of a voice are placed next to each other, borrow their voice's keys (KvPrefix) and the decode steps run the
    cascade attention over groups of 4 rows - including the group where the two voices meet (per-row path for the
    stranger) and, in the batcher, groups whose rows were re-admitted at different positions.  Every waveform must
    reproduce the one-by-one result at temp 0, in the caller's order."""
    from pocket_tts_amd.batching import ContinuousBatcher

    d = model.engine.D
    va = model.get_state_for_audio_prompt(G / "e2e_voice.safetensors")
    vb = model.get_state_for_conditioning(torch.randn(1, 45, d, generator=torch.Generator().manual_seed(5)) * 0.1)
    texts = [TEXTS[i % len(TEXTS)] for i in range(20)]
    voices = [va if i % 2 == 0 else vb for i in range(20)]
    singles = [model.generate_audio(v, t) for v, t in zip(voices, texts)]
    batch = model.generate_audio_batch(voices, texts)
    for i, (a, b) in enumerate(zip(singles, batch)):
        assert a.shape == b.shape, (i, a.shape, b.shape)
        assert np.abs(a.numpy() - b.numpy()).max() < 5e-4, i
    cb = ContinuousBatcher(model, slots=16, capacity=512)
    try:
        reqs = [cb.submit(v, t) for v, t in zip(voices, texts)]
        cb.run_until_idle()
        outs = [r.result() for r in reqs]
    finally:
        cb.close()
    for i, (a, b) in enumerate(zip(singles, outs)):
        assert a.shape == b.shape, (i, a.shape, b.shape)
        assert np.abs(a.numpy() - b.numpy()).max() < 5e-4, i


def test_generate_audio_batch_validates_inputs(model):
    state = model.get_state_for_audio_prompt(G / "e2e_voice.safetensors")
    with pytest.raises(ValueError):
        model.generate_audio_batch([state], ["a", "b"])
    with pytest.raises(ValueError):
        model.generate_audio_batch(state, ["   "])




def test_continuous_batching_join_leave_matches_single(model, fx):
    """7 requests through 3 slots: requests join as slots free up (different positions per row, codec carries of
    a joining row zeroed, parked rows in between), each must reproduce the one-by-one result at temp 0."""
    from pocket_tts_amd.batching import ContinuousBatcher

    state = model.get_state_for_audio_prompt(G / "e2e_voice.safetensors")
    singles = [model.generate_audio(state, t) for t in TEXTS]
    cb = ContinuousBatcher(model, slots=3, capacity=512)
    try:
        reqs = [cb.submit(state, t) for t in TEXTS[:5]]
        for _ in range(6):  # let the first ones run for a while, then two more arrive mid-flight
            cb.step()
        reqs += [cb.submit(state, t) for t in TEXTS[5:]]
        cb.run_until_idle()
        outs = [r.result() for r in reqs]
    finally:
        cb.close()
    for t, a, b in zip(TEXTS, singles, outs):
        assert a.shape == b.shape, (t, a.shape, b.shape)
        assert np.abs(a.numpy() - b.numpy()).max() < 5e-4, t


def test_continuous_batching_background_thread_i16_and_long_text(model, fx):
    """Background scheduler thread, 16-bit PCM written by the codec's last kernel, and a multi-chunk text (its
    chunks run one after the other from the voice state, tts_model.py:618-631)."""
    from pocket_tts_amd.batching import ContinuousBatcher

    state = model.get_state_for_audio_prompt(G / "e2e_voice.safetensors")
    long_text = "Hello world. This is a test. How are you today? This is a longer sentence, with several clauses."
    ref_long = model.generate_audio(state, long_text, max_tokens=12)
    ref_short = model.generate_audio(state, "Short one.")
    cb = ContinuousBatcher(model, slots=2, capacity=512, pcm_format="i16")
    cb.start()
    try:
        r1 = cb.submit(state, long_text, max_tokens=12)
        r2 = cb.submit(state, "Short one.")
        chunks = list(r2)
        assert all(c.dtype == torch.int16 and c.shape == (1920,) for c in chunks)
        o2 = torch.cat(chunks)
        o1 = r1.result()
    finally:
        cb.close()
    for ref, got in ((ref_long, o1), (ref_short, o2)):
        assert ref.shape == got.shape
        want = (ref.clamp(-1, 1) * 32767).short()   # StreamingWAVWriter.write_pcm_data (data/audio.py:79)
        assert (want.int() - got.int()).abs().max() <= 16  # 5e-4 * 32767


def test_continuous_batching_rejects_oversized_request(model):
    from pocket_tts_amd.batching import ContinuousBatcher

    state = model.get_state_for_audio_prompt(G / "e2e_voice.safetensors")
    cb = ContinuousBatcher(model, slots=2, capacity=64)
    try:
        with pytest.raises(ValueError):
            cb.submit(state, "This request cannot fit the slot capacity because it needs many frames.")
        with pytest.raises(ValueError):
            cb.submit(state, "   ")
    finally:
        cb.close()


def test_continuous_batching_stress_with_noise(model, fx):
    """40 requests of mixed length through 8 slots at temp 0.7 (device noise, independent per row), submitted from
    two threads while the scheduler thread runs: every request completes, chunk shapes are right, lengths respect the
    per-request maximum, samples are finite, and the slots end up parked."""
    import threading

    from pocket_tts_amd.batching import ContinuousBatcher
    from pocket_tts_amd.text import estimate_max_gen_len

    state = model.get_state_for_audio_prompt(G / "e2e_voice.safetensors")
    words = "hello world this is a test of the continuous batching scheduler with many short requests".split()
    rng = np.random.default_rng(4)
    texts = [" ".join(rng.choice(words, size=int(rng.integers(1, 9)))) + "." for _ in range(40)]
    model.temp = 0.7
    cb = ContinuousBatcher(model, slots=8, capacity=512, noise_seed=11)
    cb.start()
    reqs = [None] * len(texts)
    try:
        def feed(lo, hi):
            for i in range(lo, hi):
                reqs[i] = cb.submit(state, texts[i])

        th = [threading.Thread(target=feed, args=(0, 20)), threading.Thread(target=feed, args=(20, 40))]
        for t in th:
            t.start()
        for t in th:
            t.join()
        outs = [r.result() for r in reqs]
        assert all(j is None for j in cb.slot) and not cb.waiting
    finally:
        model.temp = 0.0
        cb.close()
    for t, o in zip(texts, outs):
        n_tok = len(model.tokenizer.encode(t)) + 8
        assert o.dim() == 1 and o.shape[0] % 1920 == 0 and torch.isfinite(o).all()
        assert 0 < o.shape[0] // 1920 <= estimate_max_gen_len(n_tok, model.config.mimi.frame_rate) + 1
