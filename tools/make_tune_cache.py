"""Measures the tile table for the bench configurations and writes it to one cache file (commit it as
profiles/tune_cache_mi355x.txt).  python tools/make_tune_cache.py <out-file>"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
out = os.path.abspath(sys.argv[1])
if os.path.exists(out):
    os.remove(out)
os.environ["PTTS_TUNE_CACHE"] = out
from pocket_tts_amd.config import named_config
from pocket_tts_amd.weights import generate_state_dict
from pocket_tts_amd.engine import Engine

for cfg_name, groups, batches in (("en100m", None, (64, 1)), ("en100m", {"attention", "ffn"}, (64, 1)),
                                  ("en100m", {"codec_bf16"}, (64,)), ("24l", None, (32,))):
    cfg = named_config(cfg_name)
    eng = Engine(cfg, generate_state_dict(cfg, 0), "cuda:0", quantize_groups=groups)
    for B in batches:
        log = eng.tune(B)
        print(f"{cfg_name} {sorted(groups) if groups else 'fp32'} B={B}: {len(log.splitlines())} shapes measured", flush=True)
    eng.close()
print(open(out).read().count("\n"), "lines ->", out)
