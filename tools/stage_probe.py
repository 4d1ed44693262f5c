"""FlowLM step graph alone, codec frame graph alone and both streams, per batch size (run on the GPU box):
python tools/stage_probe.py [batch ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("PTTS_TUNE_CACHE", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "tune_cache_mi355x.txt"))
os.environ.setdefault("PTTS_TUNE_CACHE_OUT", "/tmp/tune_stage_probe.txt")
import torch
import bench
from pocket_tts_amd.config import named_config
from pocket_tts_amd.engine import Engine
from pocket_tts_amd.weights import generate_state_dict

cfg = named_config("en100m")
eng = Engine(cfg, generate_state_dict(cfg, 0), "cuda:0")
for B in [int(x) for x in sys.argv[1:]] or [1, 64]:
    args = bench.parse(["--batch", str(B)])
    job = bench.Job(eng, B, args, 0)
    job.start_utterances()
    for _ in range(40):
        job.step()
    job.pipe.flush(); job.sync(); torch.cuda.synchronize()
    P = job.pipe
    n = 30

    def timed(fn):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n):
            fn(i)
        job.sync(); eng.sync(); torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    lm = timed(lambda i: eng.graph_launch(P.g_first[i % P.nb], P.s1))          # positions 199 .. 228
    codec = timed(lambda i: eng.graph_launch(P.g_last[i % P.nb], P.s2))
    job.start_utterances()
    for _ in range(40):
        job.step()
    job.pipe.flush(); job.sync()
    both = timed(lambda i: job.step())
    print(f"batch {B}: FlowLM graph alone {lm:.4f} ms, codec graph alone {codec:.4f} ms, both streams {both:.4f} ms per step", flush=True)
    job = None
eng.close()
