"""What limits the overlap of the two stages?  Runs pairs of graphs of INDEPENDENT states on two streams:
FlowLM || FlowLM, codec || codec, FlowLM || codec, against each alone.  python tools/overlap_probe.py [batch]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from pocket_tts_amd.config import named_config
from pocket_tts_amd.weights import generate_state_dict
from pocket_tts_amd.engine import Engine

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
cfg = named_config("en100m")
eng = Engine(cfg, generate_state_dict(cfg, 0), "cuda:0")
eng.set_option("flow_cluster", int(os.environ.get("FLOW_CLUSTER", "0")))  # 0: no cooperative kernels, no event chaining
args = bench.parse([]); args.batch = B
ja, jb = bench.Job(eng, B, args, 0), bench.Job(eng, B, args, 1)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
N = 60

def prep():
    for j in (ja, jb):
        j.start_utterances()
        for _ in range(20):
            j.step()
        j.sync()
    torch.cuda.synchronize()

def timed(fn):
    prep()
    t0 = time.perf_counter()
    for i in range(N):
        fn(i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / N * 1e3

lm = timed(lambda i: eng.graph_launch(ja.pipe.g_first[i & 1], s1))
co = timed(lambda i: eng.graph_launch(ja.pipe.g_last[i & 1], s1))
lmlm = timed(lambda i: (eng.graph_launch(ja.pipe.g_first[i & 1], s1), eng.graph_launch(jb.pipe.g_first[i & 1], s2)))
coco = timed(lambda i: (eng.graph_launch(ja.pipe.g_last[i & 1], s1), eng.graph_launch(jb.pipe.g_last[i & 1], s2)))
lmco = timed(lambda i: (eng.graph_launch(ja.pipe.g_first[i & 1], s1), eng.graph_launch(jb.pipe.g_last[i & 1], s2)))
print(f"B={B}: lm {lm:.3f} | codec {co:.3f} | lm||lm {lmlm:.3f} (x{lmlm/lm:.2f}) | codec||codec {coco:.3f} (x{coco/co:.2f}) | lm||codec {lmco:.3f} (sum {lm+co:.3f})")
eng.close()
