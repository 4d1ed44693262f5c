"""CU partition of the two pipeline stages WITH per-partition tile tuning: the FlowLM stream gets `k` CUs of every
shader engine of every XCD, the codec stream the other 8 - k (tools/../tests/hip/cu_map.hip: mask bit i = XCD i % 8,
SE (i / 8) % 4, CU (i / 8) / 4, so the CU range [4 n, 32) of ptts_stream_create_masked is CUs n..7 of all four SEs).
python tools/partition_probe.py [batch] [k ...]"""
import ctypes as C, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from pocket_tts_amd import _lib
from pocket_tts_amd.config import named_config
from pocket_tts_amd.weights import generate_state_dict
from pocket_tts_amd.engine import Engine

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
ks = [int(x) for x in sys.argv[2:]] or [0, 2, 3, 4]
cfg = named_config("en100m")
N = 80

def measure(k):
    eng = Engine(cfg, generate_state_dict(cfg, 0), "cuda:0")
    lib, H = eng.lib, eng.handle
    def masked(lo, hi):
        s = C.c_void_p()
        _lib.check(lib.ptts_stream_create_masked(H, lo, hi, C.byref(s)))
        return s, torch.cuda.ExternalStream(s.value, device="cuda:0")
    if k:
        h_co, s_co = masked(0, 32 - 4 * k)
        h_lm, s_lm = masked(32 - 4 * k, 32)
        eng.set_option("flow_max_cus", 32 * k)
        lib.ptts_tune_clear(H)
        torch.cuda.synchronize()
        _lib.check(lib.ptts_tune_streams(H, B, h_lm, h_co))
        eng._tuned.add(B)
    args = bench.parse([]); args.batch = B
    job = bench.Job(eng, B, args, 0)
    P = job.pipe
    if not k:
        s_lm, s_co = eng.stream, P.s2
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
    print(f"B={B} FlowLM on {32 * k if k else 256} CUs, codec on {256 - 32 * k}: codec alone {co:.3f} ms | lm alone {lm:.3f} | "
          f"both {bo:.3f}", flush=True)
    torch.cuda.synchronize()
    eng.close()

for k in ks:
    measure(k)
