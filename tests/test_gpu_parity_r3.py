"""Round-3 parity additions, through the C ABI on the GPU (`-m gpu`):

  * the FULL-SIZE FlowLM (en100m) against the numpy oracle at the contexts `bench.py` runs (159 - 283 keys), batch 64
    and batch 3: voice prefill of 126 positions + text prefill of 32 (the bench's shapes), decode steps at contexts
    159-161, a further prefill up to 280 positions, decode steps at contexts 281-283.  Until this round the en100m model
    met the oracle at <= 26 keys only; longer contexts were covered on `tiny` and through HIP-vs-HIP properties.
    Reference: `StreamingMultiheadAttention.forward` transformer.py:135-158, `FlowLMModel.forward` flow_lm.py:96-139.
  * the cooperative-kernel error word is read-and-clear: one transient timeout must not poison later chunks.

Tolerances as in test_gpu_parity.py: latents / PCM max-abs <= 2e-4, EOS logits <= 1e-3, EOS decisions exact wherever the
oracle's logit is further than 1e-3 from the threshold.
"""

import shutil
import subprocess
from pathlib import Path

import numpy as np
import pytest
import torch

from conftest import synth_weights
from test_gpu_parity import ATOL, _maxerr, dev, get_engine

pytestmark = pytest.mark.gpu


def _run_long_context(B, graph, shared_voice=False):
    from oracle import np_oracle as O

    cfg, W = synth_weights("en100m")
    eng = get_engine("en100m")
    if B >= 16:
        eng.tune(B)  # the tile table the bench uses for this batch (tuned shapes differ from the heuristic's)
    Tv, Tt, n1, Tx, n2 = 126, 32, 3, 119, 3  # 126 + 32 = 158 -> steps at 159..161; + 119 = 280 -> steps at 281..283
    rng = np.random.default_rng(1000 + B)
    voice = (rng.standard_normal((B, Tv, eng.D)) * 0.1).astype(np.float32)   # bench.py: N(0, 1) * 0.1 conditioning
    if shared_voice:
        voice[:] = voice[:1]
    text = (rng.standard_normal((B, Tt, eng.D)) * 0.3).astype(np.float32)
    extra = (rng.standard_normal((B, Tx, eng.D)) * 0.3).astype(np.float32)
    noise = (rng.standard_normal((n1 + n2, B, eng.ldim)) * 0.7 ** 0.5).astype(np.float32)  # temp 0.7
    cap = Tv + Tt + n1 + Tx + n2 + 1
    lm, dec = O.FlowLM(cfg, W), O.MimiDecoder(cfg, W)
    ost, oms = lm.init_state(B, cap), dec.init_state(B, n1 + n2)
    st, ms = eng.new_lm_state(B, cap), eng.new_mimi_state(B)
    lat, logit = torch.zeros(B, eng.ldim, device="cuda:0"), torch.zeros(B, device="cuda:0")
    flag = torch.zeros(B, dtype=torch.uint8, device="cuda:0")
    nz = torch.zeros(B, eng.ldim, device="cuda:0")
    g = eng.capture_lm_step(st, nz, 1, -4.0, lat, logit, flag) if graph else None
    try:
        lm.prefill(ost, voice)
        if shared_voice:  # the bench's / the API's way: ONE voice state, cloned (tts_model.py:637-638) - the clones borrow its
            vs = eng.new_lm_state(1, Tv + 1)  # first 112 keys (KvPrefix) and the decode steps run attn_cascade_kernel
            eng.lm_prefill(vs, dev(voice[:1]))
            st.copy_from(vs)
            vs.close()  # the clones keep the memory alive
        else:
            eng.lm_prefill(st, dev(voice))
        lm.prefill(ost, text)
        eng.lm_prefill(st, dev(text))
        xo = np.full((B, eng.ldim), np.nan, np.float32)
        step = 0
        for phase, n in ((0, n1), (1, n2)):
            if phase == 1:
                lm.prefill(ost, extra)
                eng.lm_prefill(st, dev(extra))
            for _ in range(n):
                ctx = int(ost[0]["offset"]) + 1
                xo, lo, eo = lm.decode_step(ost, xo, noise[step], 1, -4.0)
                po = dec.decode(oms, xo)
                if graph:
                    nz.copy_(dev(noise[step]))
                    torch.cuda.synchronize()
                    eng.graph_launch(g)
                    eng.sync()
                    xg, lg, fg = lat, logit, flag
                else:
                    xg, lg, fg = eng.lm_decode_step(st, None, dev(noise[step]), 1, -4.0)
                pg = eng.mimi_decode(ms, xg)
                torch.cuda.synchronize()
                assert _maxerr(xg.cpu().numpy(), xo) < ATOL, (ctx, "latent")
                assert _maxerr(lg.cpu().numpy(), lo) < 1e-3, (ctx, "eos logit")
                sure = np.abs(lo + 4.0) > 1e-3
                assert np.array_equal((fg.cpu().numpy() > 0)[sure], eo[sure]), (ctx, "eos decision")
                assert _maxerr(pg.cpu().numpy(), po) < ATOL, (ctx, "pcm")
                step += 1
        assert list(st.offsets()) == [Tv + Tt + n1 + Tx + n2] * B
        assert not st.error()
    finally:
        if g is not None:
            eng.graph_destroy(g)
        st.close()
        ms.close()


def test_en100m_batch64_vs_oracle_at_bench_contexts():
    """BASELINE config #3: batch 64, contexts 159-161 and 281-283, through the captured step graph the bench replays"""
    _run_long_context(64, graph=True)


@pytest.mark.parametrize("B,graph", [(64, True), (21, False)])
def test_en100m_cloned_voice_vs_oracle_at_bench_contexts(B, graph):
    """the same, with every sequence cloned from ONE voice state as bench.py and the API do: shared prefix keys
    (pointer select in the prefill attention) and the cascade decode attention (MFMA prefix tiles + per-sequence suffix),
    against the oracle, which knows nothing of sharing; 21 = a group of 4 sequences that overhangs the batch"""
    _run_long_context(B, graph, shared_voice=True)


def test_en100m_batch3_vs_oracle_at_bench_contexts():
    """a partially filled row tile (3 of 16 rows), eager launches"""
    _run_long_context(3, graph=False)


def test_state_error_word_is_read_and_clear():
    """`ptts_lm_state_error` returns the cooperative kernels' timeout word and clears it, so one transient timeout is
    reported once instead of failing every later chunk on a cached state"""
    from pocket_tts_amd import _lib

    eng = get_engine("tiny")
    st = eng.new_lm_state(2, 8)
    try:
        assert not st.error()
        _lib.check(eng.lib.ptts_debug_set_error(st.handle, 1, eng._sp))
        assert st.error()
        assert not st.error()
    finally:
        st.close()


def test_permlane_swap_inline_asm(tmp_path):
    """tests/hip/xrow_test.hip: the inline-asm v_permlane16/32_swap used by the attention kernels' cross-row reductions
    (ptts_kernels.h: xrow_swap16 / xrow_swap32) against the lane arithmetic worked out by hand; the program also prints
    whether this compiler's builtin still returns its first result twice (the reason for the asm)."""
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    src = Path(__file__).parent / "hip" / "xrow_test.hip"
    exe = tmp_path / "xrow_test"
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O2", "-std=c++17", "-Wno-unused-value", "-o", str(exe), str(src)],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    print(r.stdout)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-2000:])


def test_24l_batch32_vs_oracle():
    """BASELINE config #4 per GPU (24-layer model, 32 utterances) against the numpy ORACLE (until now
    this model met the oracle at batch 1 only and batch 32 through HIP-vs-HIP properties): voice + text prefill of the
    bench's shapes (126 + 32), three decode steps with noise, latents / EOS logits / PCM."""
    from oracle import np_oracle as O

    cfg, W = synth_weights("24l")
    eng = get_engine("24l")
    B, Tv, Tt, ns = 32, 126, 32, 3
    rng = np.random.default_rng(2432)
    voice = (rng.standard_normal((B, Tv, eng.D)) * 0.1).astype(np.float32)
    text = (rng.standard_normal((B, Tt, eng.D)) * 0.3).astype(np.float32)
    noise = (rng.standard_normal((ns, B, eng.ldim)) * 0.7 ** 0.5).astype(np.float32)
    lm, dec = O.FlowLM(cfg, W), O.MimiDecoder(cfg, W)
    ost, oms = lm.init_state(B, Tv + Tt + ns), dec.init_state(B, ns)
    st, ms = eng.new_lm_state(B, Tv + Tt + ns), eng.new_mimi_state(B)
    try:
        for e in (voice, text):
            lm.prefill(ost, e)
            eng.lm_prefill(st, dev(e))
        xo = np.full((B, eng.ldim), np.nan, np.float32)
        for i in range(ns):
            xo, lo, eo = lm.decode_step(ost, xo, noise[i], 1, -4.0)
            po = dec.decode(oms, xo)
            xg, lg, fg = eng.lm_decode_step(st, None, dev(noise[i]), 1, -4.0)
            pg = eng.mimi_decode(ms, xg)
            torch.cuda.synchronize()
            assert _maxerr(xg.cpu().numpy(), xo) < ATOL, (i, "latent")
            assert _maxerr(lg.cpu().numpy(), lo) < 1e-3, (i, "eos logit")
            sure = np.abs(lo + 4.0) > 1e-3
            assert np.array_equal((fg.cpu().numpy() > 0)[sure], eo[sure]), (i, "eos decision")
            assert _maxerr(pg.cpu().numpy(), po) < ATOL, (i, "pcm")
        assert not st.error()
    finally:
        st.close()
        ms.close()


@pytest.mark.parametrize("B", [3, 64])
def test_fused_last_conv_matches_separate_conv(B):
    """"fuse_pcm" (SEANet's last conv inside the last stage's fused residual block: per-tile partial sums + a carry for the
    first two samples of the next 64-row tile, across tiles, frames and a row reset) against the separate last conv reading
    the stored block output: same PCM to fp32 summation order, fp32 and int16, and a stage-3 tap refuses to be read while
    it stays on chip.  The goldens / oracle comparisons of the other tests run on the fused path (the default)."""
    from pocket_tts_amd._lib import PttsError
    from pocket_tts_amd.engine import Engine

    cfg, W = synth_weights("en100m")
    eng = Engine(cfg, W, "cuda:0")
    rng = np.random.default_rng(21)
    nf = 5
    lat = rng.standard_normal((nf, B, cfg.mimi.quantizer.dimension)).astype(np.float32)
    outs = []
    try:
        for fuse in (0, 1):
            eng.set_option("fuse_pcm", fuse)
            ms = eng.new_mimi_state(B)
            i16 = torch.zeros((B, eng.frame_samples), dtype=torch.int16, device="cuda:0")
            ms.set_pcm_i16(i16)
            got = []
            for f in range(nf):
                if f == 3:
                    ms.reset_row(B - 1)  # continuous batching: the row's carries (conv inputs AND tile carries) are cleared
                pcm = eng.mimi_decode(ms, dev(lat[f % 3 if f >= 3 else f]))
                torch.cuda.synchronize()
                got.append((pcm.cpu().numpy().copy(), i16.cpu().numpy().copy()))
            if fuse:
                with pytest.raises(PttsError, match="debug_taps"):
                    eng.debug_read(ms, "seanet9")
            ms.reset()  # a new utterance batch on the same state: every carry cleared, frame 0 reproduces itself
            again = eng.mimi_decode(ms, dev(lat[0])).cpu().numpy()
            assert np.abs(again - got[0][0]).max() < 2e-6, ("reset", fuse)
            outs.append(got)
            ms.close()
    finally:
        eng.close()
    worst = 0.0
    for (p0, i0), (p1, i1) in zip(*outs):
        worst = max(worst, float(np.abs(p0 - p1).max()))
        assert np.abs(i0.astype(np.int32) - i1.astype(np.int32)).max() <= 1
    print(f"B={B}: fused vs separate last conv, max abs PCM difference {worst:.2e}")
    assert worst < 2e-6, worst
    # after the reset, the reset row reproduces its own first frames (tile carries really cleared)
    assert np.abs(outs[1][3][0][B - 1] - outs[1][0][0][B - 1]).max() < 2e-6


def test_en100m_states_reused_across_utterances():
    """the streaming and batch paths keep their states, scratch and graphs between calls: a second utterance (and a second
    batch) on the reused states must equal the first - every carry of the previous utterance (conv inputs, KV ring, the fused
    last conv's tile carries, borrowed voice prefixes) cleared or re-established"""
    from pocket_tts_amd import TTSModel

    G = Path(__file__).parent / "golden"
    m = TTSModel.load_model(config=G / "e2e2_en100m.yaml", temp=0.0)
    try:
        st = m.get_state_for_conditioning(torch.randn(1, 12, 1024, generator=torch.Generator().manual_seed(3)) * 0.1)
        text = "Hello world. This is a test."
        a = m.generate_audio(st, text, frames_after_eos=2).numpy()
        other = m.generate_audio(st, "Another one.", frames_after_eos=2).numpy()
        b = m.generate_audio(st, text, frames_after_eos=2).numpy()
        assert a.shape == b.shape and a.shape[0] >= 1920 and other.shape[0] >= 1920
        assert np.abs(a - b).max() < 1e-6
        texts = [text, "Another one.", text]
        w1 = m.generate_audio_batch(st, texts, frames_after_eos=2)
        w2 = m.generate_audio_batch(st, texts, frames_after_eos=2)
        for x, y in zip(w1, w2):
            assert x.shape == y.shape and np.abs(x.numpy() - y.numpy()).max() < 1e-6
        assert w1[0].shape == a.shape and np.abs(w1[0].numpy() - a).max() < 5e-4  # batch row == single utterance
    finally:
        m.engine.close()


def test_embedding_gather_is_the_table_lookup():
    """`ptts_embed_tokens` (LUTConditioner._get_condition, reference text.py:74-76) == indexing the table, bit for bit; ids
    outside the table raise on the host like torch.nn.Embedding"""
    eng = get_engine("tiny")
    n = eng.embed.shape[0]
    tok = torch.randint(0, n, (3, 37), generator=torch.Generator().manual_seed(0))
    got = eng.embed_text(tok)
    torch.cuda.synchronize()
    assert got.shape == (3, 37, eng.D) and torch.equal(got, eng.embed[tok.to("cuda:0")])
    assert torch.equal(eng.embed_text(tok.to("cuda:0")), got)  # device-resident ids (bench.py)
    with pytest.raises(IndexError):
        eng.embed_text(torch.tensor([[0, n]]))
