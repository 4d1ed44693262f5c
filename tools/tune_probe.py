"""Prints the tuner's log (every configuration's time with PTTS_TUNE_VERBOSE=1).  python tools/tune_probe.py [batch ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("PTTS_TUNE_VERBOSE", "1")
from pocket_tts_amd.config import named_config
from pocket_tts_amd.weights import generate_state_dict
from pocket_tts_amd.engine import Engine
cfg = named_config(os.environ.get("PTTS_PROBE_CONFIG", "en100m"))
eng = Engine(cfg, generate_state_dict(cfg, 0), "cuda:0")
for B in [int(x) for x in sys.argv[1:]] or [64, 1]:
    t0 = time.time()
    log = eng.tune(B, force=True)
    print(f"==== B={B} tuned in {time.time()-t0:.2f}s")
    print(log)
    eng.lib.ptts_tune_clear(eng.handle)
eng.close()
