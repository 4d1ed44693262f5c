"""Shared voice-prefix attention ("share_prefix"): clones of a one-sequence state read its first T & ~15 keys / values in
place instead of holding copies.  The reference deep-copies the voice state per generation (tts_model.py:637-638); what a
clone computes must not depend on where its keys live, so the storage checks here are BITWISE against the same run with
the option off (full copies), on top of the oracle parity the other GPU tests establish for the unshared layout.  The
cascade decode attention ("prefix_cascade": prefix scores as MFMA tiles shared by 4 sequences) sums in another order: it
is checked against the unshared run at the fp32 tolerance here and against the oracle in test_gpu_parity_r3.py.  `-m gpu`."""

import numpy as np
import pytest
import torch

from conftest import synth_weights
from test_gpu_parity import dev

pytestmark = pytest.mark.gpu


def _run(eng, cfg, share, B, T0, T1, ns, graph, events=None, cascade=0):
    """voice state (T0 positions) -> batch state of B clones -> text prefill (T1) -> ns decode steps"""
    D, L = cfg.flow_lm.transformer.d_model, cfg.flow_lm.transformer.num_layers
    rng = np.random.default_rng(5)
    eng.set_option("share_prefix", int(share))
    eng.set_option("prefix_cascade", int(cascade))
    voice = eng.new_lm_state(1, T0 + 4)
    eng.lm_prefill(voice, dev((rng.standard_normal((1, T0, D)) * 0.5).astype(np.float32)))
    st = eng.new_lm_state(B, T0 + T1 + ns + 2)
    st.copy_from(voice)
    text = dev((rng.standard_normal((B, T1, D)) * 0.5).astype(np.float32))
    eng.lm_prefill(st, text)
    out = []
    o = torch.empty((B, cfg.mimi.quantizer.dimension), device="cuda:0")
    lg, fl = torch.empty((B,), device="cuda:0"), torch.empty((B,), dtype=torch.uint8, device="cuda:0")
    g = eng.capture_lm_step(st, None, 1, -4.0, o, lg, fl) if graph else None
    for i in range(ns):
        if events and i in events:
            events[i](st, voice)
        if g is not None:
            eng.graph_launch(g)
        else:
            eng.lm_decode_step(st, None, None, 1, -4.0, out_latent=o, out_logit=lg, out_eos=fl)
        eng.sync()
        out.append((o.cpu().numpy().copy(), lg.cpu().numpy().copy()))
    off = st.offsets()
    kv = [st.export_layer(l, int(off.max())).cpu().numpy() for l in (0, L - 1)]
    if g is not None:
        eng.graph_destroy(g)
    st.close()
    if voice.handle is not None:
        voice.close()
    return out, kv, off


@pytest.mark.parametrize("cfg_name,B,T0,T1,graph", [("tiny", 3, 37, 5, False), ("tiny", 4, 16, 1, True),
                                                    ("en100m", 5, 141, 23, True), ("en100m", 64, 125, 18, True)])
def test_shared_prefix_is_bitwise_the_full_copy(cfg_name, B, T0, T1, graph):
    from pocket_tts_amd.engine import Engine

    cfg, W = synth_weights(cfg_name)
    eng = Engine(cfg, W, "cuda:0")
    try:
        ns = 6
        ref = _run(eng, cfg, False, B, T0, T1, ns, graph)
        got = _run(eng, cfg, True, B, T0, T1, ns, graph)
    finally:
        eng.close()
    for (o, lg), (o2, lg2) in zip(ref[0], got[0]):
        assert np.array_equal(o, o2) and np.array_equal(lg, lg2)
    assert np.array_equal(ref[2], got[2])
    for a, b in zip(ref[1], got[1]):  # the export materialises the borrowed positions: same reference-layout cache
        assert np.array_equal(a, b)


def test_owner_destroyed_rows_parked_and_lent_state_guarded():
    """life-cycle around a borrowed prefix: the voice state is destroyed while its clones still decode (the memory must
    stay until they let go), a row is parked and re-admitted from another clone (the prefix travels with it), and the
    owner refuses to be rewritten while lent"""
    from pocket_tts_amd._lib import PttsError
    from pocket_tts_amd.engine import Engine

    cfg, W = synth_weights("tiny")
    eng = Engine(cfg, W, "cuda:0")
    D = cfg.flow_lm.transformer.d_model
    try:
        def close_voice(st, voice):
            voice.close()

        def park_and_readmit(st, voice):
            st.set_row_active(1, False)
            st.copy_row_from(1, st2[0], 0)  # row 1 <- row 0 of a second clone of the same voice (shares the same owner)

        st2 = [None]

        def scenario(share):
            eng.set_option("share_prefix", int(share))
            rng = np.random.default_rng(9)
            v2 = eng.new_lm_state(1, 64)
            eng.lm_prefill(v2, dev((rng.standard_normal((1, 40, D)) * 0.5).astype(np.float32)))
            st2[0] = eng.new_lm_state(2, 80)
            st2[0].copy_from(v2)
            eng.lm_prefill(st2[0], dev((rng.standard_normal((2, 3, D)) * 0.5).astype(np.float32)))
            if share:
                with pytest.raises(PttsError, match="shared prefix"):
                    v2.reset()
            r = _run(eng, cfg, share, 3, 37, 5, 8, False, events={2: close_voice, 4: park_and_readmit})
            v2.close()  # still lent to st2 (and, through copy_row_from, was lent to the closed batch state)
            o, lg, _ = eng.lm_decode_step(st2[0], None, None, 1, -4.0)
            tail = (o.cpu().numpy(), lg.cpu().numpy())
            st2[0].close()
            return r, tail

        (ref, rt), (got, gt) = scenario(False), scenario(True)
        for (o, lg), (o2, lg2) in zip(ref[0], got[0]):
            assert np.array_equal(o, o2) and np.array_equal(lg, lg2)
        assert np.array_equal(rt[0], gt[0]) and np.array_equal(rt[1], gt[1])
        assert np.array_equal(ref[2], got[2])
    finally:
        eng.close()


@pytest.mark.parametrize("cfg_name,B,T0,T1,shape", [("tiny", 16, 37, 5, 44), ("tiny", 19, 100, 3, 42), ("en100m", 64, 126, 32, 44),
                                                    ("en100m", 33, 126, 32, 84), ("en100m", 18, 47, 9, 22), ("en100m", 21, 126, 32, 442),
                                                    ("tiny", 17, 50, 4, 222), ("tiny", 16, 523, 41, 1), ("en100m", 16, 600, 50, 1)])
def test_cascade_attention_matches_per_sequence_attention(cfg_name, B, T0, T1, shape):
    """decode steps with the prefix as shared MFMA tiles == every sequence on its own, to fp32 summation order; groups
    that overhang the batch, a parked row and a row re-admitted from ANOTHER voice (its group falls back to the
    per-sequence path for that row, the other rows keep the shared tiles)"""
    from pocket_tts_amd.engine import Engine

    cfg, W = synth_weights(cfg_name)
    eng = Engine(cfg, W, "cuda:0")
    D = cfg.flow_lm.transformer.d_model
    other = [None]

    def park(st, voice):
        st.set_row_active(2, False)

    def readmit_other_voice(st, voice):
        st.copy_row_from(5, other[0], 0)

    try:
        outs = []
        for share, casc in ((False, 0), (True, shape)):
            eng.set_option("share_prefix", int(share))
            other[0] = eng.new_lm_state(1, 64)
            eng.lm_prefill(other[0], dev((np.random.default_rng(2).standard_normal((1, 33, D)) * 0.5).astype(np.float32)))
            outs.append(_run(eng, cfg, share, B, T0, T1, 8, True, events={3: park, 5: readmit_other_voice}, cascade=casc))
            other[0].close()
    finally:
        eng.close()
    ref, got = outs
    worst = 0.0
    live = np.arange(B) != 2  # the parked row's outputs are nobody's (and its stale keys are not part of its state)
    for (o, lg), (o2, lg2) in zip(ref[0], got[0]):
        worst = max(worst, float(np.abs(o - o2)[live].max()), float(np.abs(lg - lg2)[live].max()))
    first = float(np.abs(ref[0][0][0] - got[0][0][0])[live].max())
    print(f"{cfg_name} B={B} shape {shape}: cascade vs per-sequence attention, max abs difference {first:.2e} after one step, "
          f"{worst:.2e} over 8 steps (fed back)")
    assert first < 2e-5 and worst < 5e-4, (first, worst)
    assert np.array_equal(ref[2], got[2])
    for a, b in zip(ref[1], got[1]):  # [2, B, T, H, 64]; positions behind a row's offset are stale, not state
        valid = (np.arange(a.shape[2])[None, :] < ref[2][:, None]) & live[:, None]
        assert np.abs(a - b)[:, valid].max() < 2e-4
