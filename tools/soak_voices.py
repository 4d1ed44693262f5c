"""One-off soak of the shared-prefix life cycle (run on the GPU box): more voices than the device-resident voice cache holds
(8), requests of all of them interleaved through a 16-slot ContinuousBatcher and through generate_audio_batch, so voice
states are evicted (destroyed) while rows still borrow their keys, rows are re-admitted next to other voices, and groups of 4
rows mix voices.  temp 0: every result must equal the one-by-one result.  python tools/soak_voices.py [rounds]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pathlib import Path
from pocket_tts_amd import TTSModel
from pocket_tts_amd.batching import ContinuousBatcher

G = Path(__file__).parent.parent / "tests" / "golden"
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
m = TTSModel.load_model(config=G / "e2e_tiny.yaml", temp=0.0)
d = m.engine.D
gen = torch.Generator().manual_seed(11)
voices = [m.get_state_for_conditioning(torch.randn(1, 20 + 7 * i, d, generator=gen) * 0.1) for i in range(12)]
texts = ["Hello world. This is a test.", "ok", "This is a longer sentence, with several clauses, to test it.", "How are you today?",
         "Short one.", "Another request arrives while the others are running.", "Yes."]
rng = np.random.default_rng(0)
ref = {}
def single(v, t):
    k = (v, t)
    if k not in ref:
        ref[k] = m.generate_audio(voices[v], texts[t]).numpy()
    return ref[k]
t0 = time.time()
bad = 0
for r in range(rounds):
    reqs = [(int(rng.integers(0, 12)), int(rng.integers(0, len(texts)))) for _ in range(40)]
    cb = ContinuousBatcher(m, slots=16, capacity=512)
    try:
        hs = []
        for i, (v, t) in enumerate(reqs):
            hs.append(cb.submit(voices[v], texts[t]))
            if i % 5 == 4:
                for _ in range(3):
                    cb.step()
        cb.run_until_idle()
        outs = [h.result().numpy() for h in hs]
    finally:
        cb.close()
    for (v, t), o in zip(reqs, outs):
        s = single(v, t)
        if o.shape != s.shape or np.abs(o - s).max() > 5e-4:
            bad += 1
            print("MISMATCH batcher", r, v, t, o.shape, s.shape)
    sel = reqs[:24]
    outs = m.generate_audio_batch([voices[v] for v, _ in sel], [texts[t] for _, t in sel])
    for (v, t), o in zip(sel, outs):
        s = single(v, t)
        if o.shape != s.shape or np.abs(o.numpy() - s).max() > 5e-4:
            bad += 1
            print("MISMATCH batch", r, v, t)
    print(f"round {r}: {len(reqs)} batcher + {len(sel)} batch requests ok so far, bad={bad}, {time.time() - t0:.1f} s", flush=True)
m.engine.close()
print("SOAK", "FAILED" if bad else "OK")
sys.exit(1 if bad else 0)
