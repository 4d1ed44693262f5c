"""fp32 codec vs bf16 codec: codec graph alone, FlowLM alone, pipeline.  python tools/ab_bf16.py [batch ...]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from pocket_tts_amd.config import named_config
from pocket_tts_amd.weights import generate_state_dict
from pocket_tts_amd.engine import Engine

cfg = named_config("en100m")
W = generate_state_dict(cfg, 0)
for B in [int(x) for x in sys.argv[1:]] or [64, 1]:
    for groups in (None, {"codec_bf16"}, {"codec_bf16", "attention", "ffn"}):
        eng = Engine(cfg, W, "cuda:0", quantize_groups=groups)
        args = bench.parse([]); args.batch = B
        job = bench.Job(eng, B, args, 0)
        P = job.pipe
        def timed(fn, n=80):
            job.start_utterances()
            for _ in range(30):
                job.step()
            job.sync(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(n):
                fn(i)
            eng.sync(); P.s2.synchronize(); torch.cuda.synchronize()
            return (time.perf_counter() - t0) / n * 1e3
        lm = timed(lambda i: eng.graph_launch(P.g_first[i & 1]))
        mi = timed(lambda i: eng.graph_launch(P.g_last[i & 1], P.s2))
        pi = timed(lambda i: job.step())
        print(f"B={B} {sorted(groups) if groups else 'fp32'}: lm alone {lm:.3f} ms | codec alone {mi:.3f} | pipeline {pi:.3f} -> {B*0.08/pi*1e3:.0f} audio-s/s", flush=True)
        job = None
        eng.close()
