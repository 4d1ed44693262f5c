import sys, time, torch
sys.path.insert(0, "/root/repo")
from pocket_tts_amd.config import named_config
from pocket_tts_amd.weights import generate_state_dict
from pocket_tts_amd.engine import Engine
cfg = named_config("en100m")
W = generate_state_dict(cfg, 0)
eng = Engine(cfg, W, "cuda:0")
for B in (64, 1, 16):
    t0 = time.time()
    log = eng.tune(B)
    print(f"==== B={B} tuned in {time.time()-t0:.2f}s")
    print(log if B != 16 else "\n".join(l for l in log.splitlines()[-200:]))
    eng.lib.ptts_tune_clear(eng.handle)
eng.close()
