"""Where does the step time go at a given batch?  Times, with hipGraphs already captured:
(a) FlowLM step graphs alone, (b) codec graphs alone, (c) both on two streams with no ordering between them,
(d) the StepPipeline ("events").  Run on the GPU box:  python tools/stream_probe.py [batch]"""
import sys, time, torch
sys.path.insert(0, "/root/repo")
import argparse
import bench
from pocket_tts_amd.config import named_config
from pocket_tts_amd.weights import generate_state_dict
from pocket_tts_amd.engine import Engine

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
sys.argv = sys.argv[:1]
args = bench.parse()
args.batch = B
cfg = named_config("en100m")
eng = Engine(cfg, generate_state_dict(cfg, 0), "cuda:0")
job = bench.Job(eng, B, args, 0)
P = job.pipe
N = 60

def timed(fn, n=N):
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
def serial(i):
    eng.graph_launch(P.g_first[i & 1]); eng.graph_launch(P.g_last[i & 1])
se = timed(serial)
pi = timed(lambda i: job.step())
print(f"B={B}: lm alone {lm:.3f} ms | mimi alone {mi:.3f} ms | serial one stream {se:.3f} | two streams unordered {bo:.3f} | pipeline {pi:.3f} ({P.mode})")
eng.close()
