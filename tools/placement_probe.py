"""Does a sequence's result depend on WHERE in the batch it sits?  Runs the batch-64 step on 64 distinct utterances and on
a row permutation of them and reports the first intermediate (codec transformer taps) whose rows differ.
PTTS_FORCE_CFG=NT:MT:cfg pins one GEMM shape to one tile configuration (this is how the FMA-contraction issue behind
-ffp-contract=on was found: see pocket_tts_amd/_lib.py)."""
import sys, numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import test_gpu_parity as T
eng = T.get_engine("en100m")
rng = np.random.default_rng(21)
B, Tp, ns = 64, 126 + 32, 4
one = (rng.standard_normal((1, Tp, eng.D)) * 0.3).astype(np.float32)
def run(emb, tune):
    b = emb.shape[0]
    if tune:
        eng.tune(b)
    st, ms = eng.new_lm_state(b, Tp + ns + 1), eng.new_mimi_state(b)
    eng.lm_prefill(st, T.dev(emb))
    outs = []
    for _ in range(ns):
        o, lg, _ = eng.lm_decode_step(st, None, None, 1, -4.0)
        p = eng.mimi_decode(ms, o)
        torch.cuda.synchronize()
        taps = {n: eng.debug_read(ms, n).cpu().numpy().copy() for n in ('upsample', 'tr_attn', 'tr_resid', 'tr_ff', 'dec_tr')} if b == 64 else {}
        outs.append((o.cpu().numpy().copy(), lg.cpu().numpy().reshape(-1).copy(), p.cpu().numpy().copy(), taps))
    st.close(); ms.close()
    return outs
same = run(np.repeat(one, B, axis=0), True)
single = run(one, True)
emb = (rng.standard_normal((B, Tp, eng.D)) * 0.3).astype(np.float32)
perm = rng.permutation(B)
a = run(emb, False)
b = run(emb[perm], False)
a2 = run(emb, False)
for i, ((o, lg, p, t1), (o2, lg2, p2, t2), (o3, lg3, p3, t3)) in enumerate(zip(a, b, a2)):
    d = np.abs(p[perm] - p2)
    print("step", i, "lat", np.abs(o[perm] - o2).max(), "pcm", d.max(), "rows", np.nonzero(d.max(axis=1) > 0)[0][:10], "cols", np.nonzero(d.max(axis=0) > 0)[0][:10], "| repeat pcm", np.abs(p - p3).max(), "lg", np.abs(lg[perm]-lg2).max())
    for n in t1:
        v = t1[n].reshape(64, 16, -1)[perm]; w = t2[n].reshape(64, 16, -1)
        dd = np.abs(v - w)
        if dd.max() > 0:
            r = np.nonzero(dd.reshape(64, -1).max(axis=1) > 0)[0]
            tt = np.nonzero(dd[r[0]].max(axis=1) > 0)[0]
            print("    tap", n, "max", dd.max(), "rows", r[:6], "positions", tt[:16])
