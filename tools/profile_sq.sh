#!/bin/bash
# SQ counters of one bench configuration (run on the GPU box through gpurun):  tools/profile_sq.sh <tag> [bench.py args ...]
# One rocprofv3 --pmc pass (program directly after `--`), kernel trace in the same run for the durations; writes
# gpurun_out/<tag>_sq/ and prints the per-kernel table (tools/pmc_sq.py); copy the JSON into profiles/ to keep it.
set -o pipefail
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out
export PTTS_TUNE_CACHE=$root/profiles/tune_cache_mi355x.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA \
  --kernel-trace -d $out/${tag}_sq -o run --output-format csv -- python3 $root/bench.py --steps 12 --warmup 3 --min-seconds 0 --min-utterances 1 --quick "$@" > $out/${tag}_sq.log 2>&1 || exit 3
tid=$(python3 -c "import json,sys; print(json.loads(open('$out/${tag}_sq.log').read().strip().split('\n')[-1])['tune_table_id'])" 2>/dev/null)
python3 $root/tools/pmc_sq.py $out/${tag}_sq $out/${tag}_pmc_sq.json "rocprofv3 --pmc SQ_* --kernel-trace, bench.py --steps 12 --warmup 3 $*" "$tid" | tee $out/${tag}_pmc_sq.txt
find $out/${tag}_sq -name "*.csv" -size +20M -delete
du -sh $out/${tag}_sq
