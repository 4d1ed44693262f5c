#!/bin/bash
# rocprofv3 evidence of one bench configuration (run on the GPU box through gpurun):
#   tools/profile_round.sh <tag> [bench.py args ...]
# writes gpurun_out/<tag>_{stats,fetch,write}/ (kernel trace + stats; PMC FETCH_SIZE; PMC WRITE_SIZE, separate passes as
# MI355X_MICROARCH.md prescribes) and gpurun_out/<tag>_bench.json (the un-profiled bench line of the same build).
set -o pipefail
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out
export PTTS_TUNE_CACHE=$root/profiles/tune_cache_mi355x.txt
cd /tmp && export TMPDIR=/tmp
python3 $root/bench.py "$@" > $out/${tag}_bench.json 2> $out/${tag}_bench.err || exit 1
rocprofv3 --kernel-trace --stats -d $out/${tag}_stats -o run --output-format csv -- python3 $root/bench.py --quick "$@" > $out/${tag}_stats.log 2>&1 || exit 2
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $out/${tag}_fetch -o run --output-format csv -- python3 $root/bench.py --steps 12 --warmup 3 --min-seconds 0 --min-utterances 1 --quick "$@" > $out/${tag}_fetch.log 2>&1 || exit 3
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $out/${tag}_write -o run --output-format csv -- python3 $root/bench.py --steps 12 --warmup 3 --min-seconds 0 --min-utterances 1 --quick "$@" > $out/${tag}_write.log 2>&1 || exit 4
ls $out/${tag}_stats $out/${tag}_fetch | head -20
# keep only what is needed (the raw traces are large)
find $out/${tag}_fetch $out/${tag}_write -name "*kernel_trace.csv" -delete
du -sh $out/${tag}_*
