"""Replays the FlowLM step graph alone (for rocprofv3 --kernel-trace --stats).  python tools/lm_only.py [batch] [n]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from pocket_tts_amd.config import named_config
from pocket_tts_amd.weights import generate_state_dict
from pocket_tts_amd.engine import Engine

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
n = int(sys.argv[2]) if len(sys.argv) > 2 else 60
which = sys.argv[3] if len(sys.argv) > 3 else "lm"
sys.argv = sys.argv[:1]
cfg = named_config("en100m")
eng = Engine(cfg, generate_state_dict(cfg, 0), "cuda:0")
args = bench.parse()
args.batch = B
job = bench.Job(eng, B, args, 0)
job.start_utterances()
for _ in range(30):
    job.step()
job.sync()
for i in range(n):
    if which in ("lm", "both"):
        eng.graph_launch(job.pipe.g_first[i & 1])
    if which in ("codec", "both"):
        eng.graph_launch(job.pipe.g_last[i & 1], job.pipe.s2)
eng.sync(); job.pipe.s2.synchronize()
print("done", job.st.error())
