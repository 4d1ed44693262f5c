"""Throughput of a fixed total batch split into K independent sub-batches (own states, graphs and stream pairs),
stepped round-robin from one host thread.  python tools/subbatch_probe.py [total] [K ...]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from pocket_tts_amd.config import named_config
from pocket_tts_amd.weights import generate_state_dict
from pocket_tts_amd.engine import Engine

total = int(sys.argv[1]) if len(sys.argv) > 1 else 64
Ks = [int(x) for x in sys.argv[2:]] or [1, 2, 4]
cfg = named_config("en100m")
eng = Engine(cfg, generate_state_dict(cfg, 0), "cuda:0")
for K in Ks:
    args = bench.parse([]); args.batch = total // K
    jobs = [bench.Job(eng, total // K, args, s) for s in range(K)]
    for j in jobs:
        j.pipe.mode = "events"  # throughput mode also for small sub-batches
        if K > 1:
            j.pipe.s1 = torch.cuda.Stream()  # own FlowLM stream per sub-batch
    def run(n):
        for _ in range(n):
            for j in jobs:
                j.step()
        for j in jobs:
            j.sync()
        torch.cuda.synchronize()
    run(30)
    t0 = time.perf_counter()
    n = 125
    run(n)
    dt = time.perf_counter() - t0
    print(f"total {total} as {K} x {total // K}: {dt / n * 1e3:.3f} ms per step of all -> {total * n * 0.08 / dt:.0f} audio-s/s", flush=True)
    for j in jobs:
        j.pipe.close()
    jobs = None
eng.close()
