import torch, sys
sys.path.insert(0, "/root/repo")
from tests.conftest import synth_weights
from pocket_tts_amd.engine import Engine
cfg, W = synth_weights("tiny", 0)
eng = Engine(cfg, W, "cuda:0")
ss = [torch.cuda.Stream() for _ in range(12)]
print("overlap of eng.stream with 12 fresh torch streams:", [int(eng.streams_overlap(eng.stream, s)) for s in ss])
print("pairs (i, i+1):", [int(eng.streams_overlap(ss[i], ss[i + 1])) for i in range(11)])
print("pairs (i, i+4):", [int(eng.streams_overlap(ss[i], ss[i + 4])) for i in range(8)])
