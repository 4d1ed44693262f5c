"""N > 1 path on CPU: world_size 2 over gloo.  Checks the utterance sharding (every utterance decoded by
exactly one rank, no data-path collective needed) and the whole-job aggregation used by bench.py
(sum of units over ranks / max wall over ranks)."""

import json
import subprocess
import sys
from pathlib import Path

import numpy as np

from pocket_tts_amd import parallel

HERE = Path(__file__).parent


def test_shard_is_a_partition():
    for n in (0, 1, 7, 64, 256):
        for world in (1, 2, 3, 8):
            parts = [parallel.shard(n, r, world) for r in range(world)]
            flat = sorted(i for p in parts for i in p)
            assert flat == list(range(n))
            assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1


def test_world_size_2_gloo(tmp_path):
    out = tmp_path / "res.json"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", "29533", str(HERE / "_multirank_worker.py"), str(out)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    res = json.loads(out.read_text())
    assert res["world"] == 2
    frames = {}
    for d in res["frames"]:
        for k, v in d.items():
            assert k not in frames  # each utterance has exactly one owner
            frames[int(k)] = v
    assert sorted(frames) == list(range(7))
    expect = {i: int(np.random.default_rng(i).integers(3, 9)) for i in range(7)}
    assert frames == expect
    assert res["wall"] >= 0.1  # max over ranks (rank 1 sleeps 0.1 s)
    assert abs(res["value"] - sum(expect.values()) * 0.08 / res["wall"]) < 1e-9


def test_single_process_path():
    v, w = parallel.job_throughput(10.0, 2.0, None)
    assert v == 5.0 and w == 2.0
