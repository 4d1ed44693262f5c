#!/bin/bash
# bench lines of the secondary configurations (run on the GPU box): tools/bench_matrix.sh <out-dir>
out=${1:-gpurun_out/matrix}; mkdir -p $out
export PTTS_TUNE_CACHE=$PWD/profiles/tune_cache_mi355x.txt
export PTTS_TUNE_CACHE_OUT=$PWD/$out/tune_additions.txt
Q="--quick"
python bench.py --preset config4 $Q > $out/config4_24l_b32.json 2> $out/config4.err
python bench.py --preset int8 $Q > $out/int8.json 2> $out/int8.err
python bench.py --preset bf16codec $Q > $out/bf16codec.json 2> $out/bf16codec.err
python bench.py --preset config5 $Q > $out/config5_int8_bf16codec.json 2> $out/config5.err
for b in 1 4 16 32 128 256; do python bench.py --batch $b $Q > $out/b$b.json 2> $out/b$b.err; done
python - <<'P'
import json,glob,os
for f in sorted(glob.glob(os.environ.get("OUT","gpurun_out/matrix")+"/*.json")):
    try:
        d=json.load(open(f)); print(os.path.basename(f), round(d["value"]), "audio-s/s", round(d["ms_per_step"],3), "ms/step", d["dtype"][:40])
    except Exception as e: print(f, "FAILED", e)
P
