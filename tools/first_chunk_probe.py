"""Where does the batch-1 first-chunk latency go?  (state clone, text prefill, codec reset, first FlowLM step + frame)"""
import sys, time, torch, numpy as np
sys.path.insert(0, "/root/repo")
import bench
from pocket_tts_amd.config import named_config
from pocket_tts_amd.weights import generate_state_dict
from pocket_tts_amd.engine import Engine
sys.argv = sys.argv[:1]
args = bench.parse(); args.batch = 1
cfg = named_config("en100m")
eng = Engine(cfg, generate_state_dict(cfg, 0), "cuda:0")
job = bench.Job(eng, 1, args, 7)
emb = eng.embed_text(job.tokens)
def t(fn, n=50):
    xs = []
    for _ in range(n):
        job.sync(); eng.sync(); torch.cuda.synchronize()
        t0 = time.perf_counter(); fn(); eng.sync(); job.pipe.s2.synchronize(); xs.append((time.perf_counter() - t0) * 1e3)
    return float(np.median(xs))
print("copy_from voice state   %.3f ms" % t(lambda: job.st.copy_from(job.voice)))
def pre():
    job.st.copy_from(job.voice); eng.lm_prefill(job.st, emb)
print("copy + text prefill(32) %.3f ms" % t(pre))
print("pipe.restart            %.3f ms" % t(lambda: job.pipe.restart()))
def first():
    job.start_utterances(); job.pipe.step(); f = job.pipe.flush(); job.pipe.done_event(f).synchronize()
print("whole first chunk       %.3f ms" % t(first))
def steps():
    job.pipe.step()
job.start_utterances()
print("one more step           %.3f ms" % t(steps, 20))
eng.close()
