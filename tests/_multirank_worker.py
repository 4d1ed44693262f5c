"""Worker for tests/test_multirank_cpu.py: run under torch.distributed.run with the gloo backend."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from pocket_tts_amd import parallel  # noqa: E402


def main():
    out = sys.argv[1]
    rank, local, world = parallel.env_ranks()
    dist = parallel.init_distributed("gloo")
    n_utt = 7
    mine = parallel.shard(n_utt, rank, world)
    # each "utterance" is decoded independently by its owner (stand-in workload: seeded by its id)
    frames = {i: int(np.random.default_rng(i).integers(3, 9)) for i in mine}
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    time.sleep(0.05 * (rank + 1))  # rank 1 is slower: the job time is the max
    wall = time.perf_counter() - t0
    value, job_wall = parallel.job_throughput(sum(frames.values()) * 0.08, wall, dist)
    gathered = [None] * world
    if dist is not None:
        dist.all_gather_object(gathered, frames)
    else:
        gathered = [frames]
    if rank == 0:
        json.dump(dict(world=world, value=value, wall=job_wall, frames=gathered), open(out, "w"))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
