#!/bin/bash
# bench lines of the secondary configurations, all on ONE box (run through gpurun): tools/bench_matrix.sh <out-dir>
out=${1:-gpurun_out/matrix}; mkdir -p $out
export PTTS_TUNE_CACHE=$PWD/profiles/tune_cache_mi355x.txt
export PTTS_TUNE_CACHE_OUT=$PWD/$out/tune_additions.txt
Q="--quick"
python bench.py $Q > $out/headline.json 2> $out/headline.err
for p in config4 int8 bf16codec config5 fp8codec config5fp8 lmbf16 split; do python bench.py --preset $p $Q > $out/$p.json 2> $out/$p.err; done
for b in 1 4 16 32 128 256; do python bench.py --batch $b $Q > $out/b$b.json 2> $out/b$b.err; done
# every utterance cloned from ONE voice (shared prefix keys + cascade attention): headline shape, 24-layer model, larger batches
python bench.py $Q --voices one > $out/onevoice.json 2> $out/onevoice.err
python bench.py --preset config4 $Q --voices one > $out/onevoice_config4.json 2> $out/onevoice_config4.err
for b in 16 32 128 256; do python bench.py --batch $b $Q --voices one > $out/onevoice_b$b.json 2> $out/onevoice_b$b.err; done
for p in bf16codec config5 lmbf16; do python bench.py --preset $p $Q --voices one > $out/onevoice_$p.json 2> $out/onevoice_$p.err; done
python bench.py $Q > $out/headline_again.json 2> $out/headline_again.err
OUT=$out python - <<'P'
import json,glob,os
rows = {}
for f in sorted(glob.glob(os.environ["OUT"]+"/*.json")):
    try:
        d=json.loads(open(f).read().strip().split("\n")[-1]); rows[os.path.basename(f)[:-5]] = dict(value=round(d["value"],1), ms_per_step=round(d["ms_per_step"],4), dtype=d["dtype"], workload=d["config"]["workload"][:60])
        print(os.path.basename(f), round(d["value"]), "audio-s/s", round(d["ms_per_step"],3), "ms/step", d["dtype"][:50])
    except Exception as e: print(f, "FAILED", e)
json.dump(rows, open(os.environ["OUT"]+"/matrix.json","w"), indent=1)
P
