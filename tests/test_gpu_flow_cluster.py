"""The single-launch flow MLP (csrc/ptts_flow.h): parity with the one-launch-per-layer path, determinism of the
in-launch hand-offs under load, and engine isolation (no process-global state in libptts).  `-m gpu`.

The hand-off protocol (write-through stores + flag, polled by the consumers) must never deliver stale bytes.  A stale
read would show as (a) a difference between two runs of the same trajectory, (b) a deviation from the multi-launch
path far above fp32 summation-order noise.  Both are checked over many steps, with a second stream keeping the chip
busy (uneven load is what exposes visibility bugs: MI355X_MICROARCH.md, "Test every hand-off under UNEVEN load").
"""

import threading

import numpy as np
import pytest
import torch

from conftest import synth_weights
from test_gpu_parity import ATOL, _maxerr, dev, get_engine

pytestmark = pytest.mark.gpu


def _trajectory(eng, B, steps, lsd, seed, busy=None, ctx=20):
    rng = np.random.default_rng(seed)
    st = eng.new_lm_state(B, ctx + steps + 1)
    eng.lm_prefill(st, dev((rng.standard_normal((B, ctx, eng.D)) * 0.3).astype(np.float32)))
    out = []
    for i in range(steps):
        noise = dev((rng.standard_normal((B, eng.ldim)) * 0.8).astype(np.float32))
        if busy is not None:
            busy()
        o, lg, _ = eng.lm_decode_step(st, None, noise, lsd, -4.0)
        out.append((o.clone(), lg.clone()))
    torch.cuda.synchronize()
    assert not st.error()
    lat = np.stack([o.cpu().numpy() for o, _ in out])
    lg = np.stack([l.cpu().numpy() for _, l in out])
    st.close()
    return lat, lg


@pytest.mark.parametrize("cfg_name,B,lsd", [("tiny", 3, 2), ("tiny", 37, 1), ("en100m", 1, 1), ("en100m", 5, 2),
                                            ("en100m", 64, 1), ("en100m", 100, 1)])
def test_cluster_matches_per_layer_launches(cfg_name, B, lsd):
    eng = get_engine(cfg_name)
    steps = 6
    eng.set_option("flow_cluster", 0)
    ref, ref_lg = _trajectory(eng, B, steps, lsd, 11)
    eng.set_option("flow_cluster", 1)
    got, got_lg = _trajectory(eng, B, steps, lsd, 11)
    assert np.isfinite(got).all()
    assert _maxerr(got, ref) < ATOL, np.abs(got - ref).reshape(steps, -1).max(1)
    assert _maxerr(got_lg, ref_lg) < 1e-3


def test_cluster_is_deterministic_under_load():
    """120 steps at batch 64, twice, while a second stream hammers HBM and the CUs: bit-equal latents."""
    eng = get_engine("en100m")
    eng.set_option("flow_cluster", 1)
    side = torch.cuda.Stream()
    big = torch.randn(64 * 1024 * 1024, device="cuda:0")
    mats = torch.randn(2048, 2048, device="cuda:0")
    k = [0]

    def busy():
        k[0] += 1
        with torch.cuda.stream(side):
            if k[0] % 3 == 0:
                big.mul_(1.0000001)      # streaming load on every CU
            elif k[0] % 3 == 1:
                torch.mm(mats, mats)     # compute-heavy blocks holding CUs
            # every third step: idle chip

    a, _ = _trajectory(eng, 64, 120, 1, 5, busy)
    b, _ = _trajectory(eng, 64, 120, 1, 5, None)
    c, _ = _trajectory(eng, 64, 120, 1, 5, busy)
    side.synchronize()
    assert np.isfinite(a).all()
    assert np.array_equal(a, b) and np.array_equal(a, c)


def test_cluster_vs_golden_chain_many_steps(golden):
    """the en100m golden trajectory (12 chained steps) through graph replay of the cluster path"""
    g = golden("en100m_b2_noise")
    m = g["meta"]
    eng = get_engine(m["config"], m["seed"])
    eng.set_option("flow_cluster", 1)
    B, Tv, Tt, ns = m["B"], m["Tv"], m["Tt"], m["n_steps"]
    st = eng.new_lm_state(B, Tv + Tt + ns)
    eng.lm_prefill(st, dev(g["voice_emb"]))
    eng.lm_prefill(st, dev(g["text_emb"]))
    noise = torch.zeros(B, eng.ldim, device="cuda:0")
    out = torch.zeros(B, eng.ldim, device="cuda:0")
    lg = torch.zeros(B, device="cuda:0")
    fl = torch.zeros(B, dtype=torch.uint8, device="cuda:0")
    eng.sync()
    gr = eng.capture_lm_step(st, noise, m["lsd_steps"], -4.0, out, lg, fl)
    for i in range(ns):
        with torch.cuda.stream(eng.stream):
            noise.copy_(dev(g["noise"][i]))
        eng.graph_launch(gr)
        eng.sync()
        assert _maxerr(out.cpu().numpy(), g["latents"][i]) < ATOL, i
    eng.graph_destroy(gr)
    assert not st.error()


def test_two_engines_two_threads_are_independent():
    """One process, two engines (different models), each driven by its own thread at the same time: results are
    bit-equal to the same trajectories run alone (no process-global state)."""
    from pocket_tts_amd.engine import Engine

    cfg_a, W_a = synth_weights("tiny", 0)
    cfg_b, W_b = synth_weights("tiny", 3)
    ea, eb = Engine(cfg_a, W_a, "cuda:0"), Engine(cfg_b, W_b, "cuda:0")
    try:
        def run(eng, seed, mimi):
            lat, _ = _trajectory(eng, 4, 25, 1, seed)
            pcm = None
            if mimi:
                ms = eng.new_mimi_state(4)
                pcm = np.stack([eng.mimi_decode(ms, dev(lat[i])).cpu().numpy() for i in range(6)])
                ms.close()
            return lat, pcm

        alone_a, alone_b = run(ea, 1, True), run(eb, 2, True)
        res = {}
        for _ in range(3):
            ts = [threading.Thread(target=lambda: res.__setitem__("a", run(ea, 1, True))),
                  threading.Thread(target=lambda: res.__setitem__("b", run(eb, 2, True)))]
            [t.start() for t in ts]
            [t.join() for t in ts]
            for got, want in ((res["a"], alone_a), (res["b"], alone_b)):
                assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
        # profiler state is per engine: profiling one engine records nothing of the other
        ea.profile_start()
        run(eb, 2, False)
        assert ea.profile_stop() == []
    finally:
        ea.close()
        eb.close()


# ---- the transformer stack as one launch (csrc/ptts_lm.h) ----------------------------------------------------------
@pytest.mark.parametrize("cfg_name,B,ctx", [("tiny", 3, 20), ("tiny", 37, 5), ("en100m", 1, 40), ("en100m", 5, 170),
                                            ("en100m", 64, 33), ("en100m", 100, 20), ("24l", 32, 24)])
def test_lm_cluster_matches_per_layer_launches(cfg_name, B, ctx):
    """latents, EOS logits AND the KV rows written by the single-launch transformer stack against five launches per
    layer (fp32 summation order differs: K-split over 8 waves instead of 4)"""
    eng = get_engine(cfg_name)
    steps = 5
    out = {}
    for opt in (0, 1):
        eng.set_option("lm_cluster", opt)
        rng = np.random.default_rng(17)
        st = eng.new_lm_state(B, ctx + steps + 1)
        eng.lm_prefill(st, dev((rng.standard_normal((B, ctx, eng.D)) * 0.3).astype(np.float32)))
        lat = []
        for i in range(steps):
            noise = dev((rng.standard_normal((B, eng.ldim)) * 0.8).astype(np.float32))
            o, lg, _ = eng.lm_decode_step(st, None, noise, 1, -4.0)
            lat.append((o.cpu().numpy(), lg.cpu().numpy()))
        assert not st.error()
        kv = [st.export_layer(l, ctx + steps).cpu().numpy()[:, :, ctx:] for l in (0, eng.L - 1)]
        out[opt] = (lat, kv, list(st.offsets()))
        st.close()
    eng.set_option("lm_cluster", 0)
    assert out[0][2] == out[1][2] == [ctx + steps] * B
    for (o0, l0), (o1, l1) in zip(out[0][0], out[1][0]):
        assert np.isfinite(o1).all()
        assert _maxerr(o0, o1) < ATOL and _maxerr(l0, l1) < 1e-3
    for k0, k1 in zip(out[0][1], out[1][1]):
        assert _maxerr(k0, k1) < ATOL


def test_lm_cluster_mixed_positions_and_parked_rows():
    """per-row offsets (rows prefilled to different lengths and copied into a batch; one row parked): the cluster
    kernel reads every row's own position, like the per-layer path"""
    eng = get_engine("en100m")
    rng = np.random.default_rng(23)
    lens = [7, 40, 19, 33, 12]
    res = {}
    for opt in (0, 1):
        eng.set_option("lm_cluster", opt)
        big = eng.new_lm_state(len(lens), 64)
        r2 = np.random.default_rng(5)
        for b, T in enumerate(lens):
            one = eng.new_lm_state(1, 64)
            eng.lm_prefill(one, dev((r2.standard_normal((1, T, eng.D)) * 0.3).astype(np.float32)))
            big.copy_row_from(b, one)
            eng.sync()
            one.close()
        big.set_row_active(3, False)
        outs = []
        for _ in range(4):
            o, lg, _ = eng.lm_decode_step(big, None, None, 1, -4.0)
            outs.append(o.cpu().numpy())
        res[opt] = (outs, list(big.offsets()))
        big.close()
    eng.set_option("lm_cluster", 0)
    assert res[0][1] == res[1][1] == [l + 4 if b != 3 else 0 for b, l in enumerate(lens)]
    for a, b in zip(res[0][0], res[1][0]):
        keep = [0, 1, 2, 4]
        assert _maxerr(a[keep], b[keep]) < ATOL


def test_lm_cluster_deterministic_under_load_and_two_states():
    """bit-equal trajectories with a busy second stream, and two states stepped from two threads at once (their
    cooperative launches are chained on the GPU, never starving each other)"""
    eng = get_engine("en100m")
    eng.set_option("lm_cluster", 1)
    side = torch.cuda.Stream()
    big = torch.randn(64 * 1024 * 1024, device="cuda:0")
    mats = torch.randn(2048, 2048, device="cuda:0")
    k = [0]

    def busy():
        k[0] += 1
        with torch.cuda.stream(side):
            if k[0] % 3 == 0:
                big.mul_(1.0000001)
            elif k[0] % 3 == 1:
                torch.mm(mats, mats)

    a, _ = _trajectory(eng, 64, 60, 1, 5, busy)
    b, _ = _trajectory(eng, 64, 60, 1, 5, None)
    side.synchronize()
    assert np.isfinite(a).all() and np.array_equal(a, b)
    res = {}
    ts = [threading.Thread(target=lambda: res.__setitem__(0, _trajectory(eng, 64, 60, 1, 5)[0])),
          threading.Thread(target=lambda: res.__setitem__(1, _trajectory(eng, 48, 60, 1, 9)[0]))]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert np.array_equal(res[0], a)
    assert np.array_equal(res[1], _trajectory(eng, 48, 60, 1, 9)[0])
    eng.set_option("lm_cluster", 0)
