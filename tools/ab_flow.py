"""A/B of the single-launch flow MLP against one launch per layer: FlowLM step graph alone, codec graph alone, both
on two streams, the pipeline.  python tools/ab_flow.py [batch ...]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from pocket_tts_amd.config import named_config
from pocket_tts_amd.weights import generate_state_dict
from pocket_tts_amd.engine import Engine

batches = [int(x) for x in sys.argv[1:]] or [64, 1]
sys.argv = sys.argv[:1]
cfg = named_config("en100m")
W = generate_state_dict(cfg, 0)
for B in batches:
    for opt in (0, 1):
        eng = Engine(cfg, W, "cuda:0")
        eng.set_option("flow_cluster", opt)
        args = bench.parse()
        args.batch = B
        job = bench.Job(eng, B, args, 0)
        P = job.pipe

        def timed(fn, n=60):
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
        def both(i):
            eng.graph_launch(P.g_first[i & 1]); eng.graph_launch(P.g_last[i & 1], P.s2)
        bo = timed(both)
        pi = timed(lambda i: job.step())
        print(f"B={B} flow_cluster={opt}: lm alone {lm:.3f} ms | codec alone {mi:.3f} | two streams unordered {bo:.3f} | pipeline {pi:.3f} ({P.mode}) err={job.st.error()}", flush=True)
        job = None
        eng.close()
