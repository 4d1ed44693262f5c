"""Where does the time of an API-level batch call go?  cProfile of TTSModel.generate_audio_batch / ContinuousBatcher
(64 fixed-length requests, as bench.py's api_batch leg).  python tools/api_probe.py [batch]"""
import cProfile, io, os, pstats, sys, time, logging
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("PTTS_TUNE_CACHE", "profiles/tune_cache_mi355x.txt")
import torch
import bench
from pocket_tts_amd.batching import ContinuousBatcher
from pocket_tts_amd.config import named_config
from pocket_tts_amd.engine import Engine
from pocket_tts_amd.tts_model import TTSModel, _export_lm_state
from pocket_tts_amd.weights import generate_state_dict

logging.getLogger("pocket_tts_amd").setLevel(logging.ERROR)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
cfg = named_config("en100m")
eng = Engine(cfg, generate_state_dict(cfg, 0), "cuda:0")
model = TTSModel(eng, cfg, bench._CharTokenizer(cfg.flow_lm.lookup_table.n_bins), 0.7, 1, None, float("inf"))
v = eng.new_lm_state(1, 128)
eng.lm_prefill(v, (torch.randn(1, 126, eng.D) * 0.1).cuda())
voice = _export_lm_state(eng, v, 126)
texts = [f"The quick brown fox jumps {i:04d}." for i in range(B)]
for i in range(2):
    t0 = time.perf_counter(); model.generate_audio_batch(voice, texts); print("call", i, time.perf_counter() - t0, flush=True)
pr = cProfile.Profile(); pr.enable()
t0 = time.perf_counter(); model.generate_audio_batch(voice, texts); dt = time.perf_counter() - t0
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(25); print("generate_audio_batch", dt); print(s.getvalue()[:6000])
model._drop_batch_contexts()
cb = ContinuousBatcher(model, slots=B, capacity=512)
for i in range(2):
    t0 = time.perf_counter(); reqs = [cb.submit(voice, t) for t in texts]; cb.run_until_idle(); [r.result() for r in reqs]; print("cb", i, time.perf_counter() - t0, flush=True)
pr = cProfile.Profile(); pr.enable()
t0 = time.perf_counter(); reqs = [cb.submit(voice, t) for t in texts]; cb.run_until_idle(); n = sum(c.shape[0] for r in reqs for c in r); dt = time.perf_counter() - t0
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(25); print("batcher", dt, n); print(s.getvalue()[:6000])
cb.close()
