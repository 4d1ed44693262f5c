"""Does giving the FlowLM stream and the codec stream DISJOINT CU sets (the same slice of every XCD) make them overlap?
Both as graph launches on CU-masked streams.  python tools/cu_mask_probe.py [batch] [codec CUs per XCD ...]"""
import ctypes as C, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from pocket_tts_amd import _lib
from pocket_tts_amd.config import named_config
from pocket_tts_amd.weights import generate_state_dict
from pocket_tts_amd.engine import Engine

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
shares = [int(x) for x in sys.argv[2:]] or [32, 28, 24, 20, 16]
cfg = named_config("en100m")
eng = Engine(cfg, generate_state_dict(cfg, 0), "cuda:0")
args = bench.parse([]); args.batch = B
job = bench.Job(eng, B, args, 0)
P = job.pipe
lib, H = eng.lib, eng.handle
N = 80

def masked(lo, hi):
    s = C.c_void_p()
    _lib.check(lib.ptts_stream_create_masked(H, lo, hi, C.byref(s)))
    return s, torch.cuda.ExternalStream(s.value, device="cuda:0")

def run(s_lm, s_co, label):
    def t(fn):
        job.start_utterances()
        for _ in range(20):
            job.step()
        job.sync(); s_lm.synchronize(); s_co.synchronize(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(N):
            fn(i)
        eng.sync(); s_lm.synchronize(); s_co.synchronize(); torch.cuda.synchronize()
        return (time.perf_counter() - t0) / N * 1e3
    co = t(lambda i: eng.graph_launch(P.g_last[i & 1], s_co))
    lm = t(lambda i: eng.graph_launch(P.g_first[i & 1], s_lm))
    def both(i):
        eng.graph_launch(P.g_first[i & 1], s_lm); eng.graph_launch(P.g_last[i & 1], s_co)
    bo = t(both)
    print(f"B={B} {label}: codec alone {co:.3f} ms | lm alone {lm:.3f} | both unordered {bo:.3f}", flush=True)

run(eng.stream, P.s2, "unmasked")
for n in shares:
    h1, s_co = masked(0, n)
    if n < 32:
        h2, s_lm = masked(n, 32)
        run(s_lm, s_co, f"codec CUs [0,{n}) of each XCD, lm CUs [{n},32)")
        run(eng.stream, s_co, f"codec CUs [0,{n}) of each XCD, lm unmasked")
        torch.cuda.synchronize(); lib.ptts_stream_destroy(h2)
    else:
        run(eng.stream, s_co, "codec on a full mask")
    torch.cuda.synchronize(); lib.ptts_stream_destroy(h1)
eng.close()
