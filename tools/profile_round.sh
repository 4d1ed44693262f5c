#!/bin/bash
# rocprofv3 evidence of one bench configuration (run on the GPU box through gpurun):
#   tools/profile_round.sh <tag> [bench.py args ...]
# writes into gpurun_out/ (copy what is to be judged into profiles/):
#   <tag>_bench.json          the un-profiled bench line of the same build
#   <tag>_kernel_stats.csv    rocprofv3 --kernel-trace --stats of the same command (per-kernel calls / total / average)
#   <tag>_pmc_traffic.json    separate --pmc FETCH_SIZE and --pmc WRITE_SIZE passes (MI355X_MICROARCH.md: FETCH_SIZE doubled)
#   <tag>_onevoice_kernel_stats.csv / _onevoice_pmc_traffic.json   the same two for `--voices one` (shared prefix keys)
#   <tag>_pmc_sq.json         one --pmc pass of SQ counters (MFMA busy, wave cycles, waits), tools/pmc_sq.py
# every rocprofv3 run has the program directly after `--`; counters are never combined with other trace domains.
set -o pipefail
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out
export PTTS_TUNE_CACHE=$root/profiles/tune_cache_mi355x.txt
cd /tmp && export TMPDIR=/tmp
python3 $root/bench.py "$@" > $out/${tag}_bench.json 2> $out/${tag}_bench.err || exit 1
tid=$(python3 -c "import json; print(json.loads(open('$out/${tag}_bench.json').read().strip().split('\n')[-1])['tune_table_id'])")
rocprofv3 --kernel-trace --stats -d $out/${tag}_stats -o run --output-format csv -- python3 $root/bench.py --quick "$@" > $out/${tag}_stats.log 2>&1 || exit 2
cp $(find $out/${tag}_stats -name "*kernel_stats.csv" | head -1) $out/${tag}_kernel_stats.csv
Q="--steps 12 --warmup 3 --min-seconds 0 --min-utterances 1 --quick"
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $out/${tag}_fetch -o run --output-format csv -- python3 $root/bench.py $Q "$@" > $out/${tag}_fetch.log 2>&1 || exit 3
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $out/${tag}_write -o run --output-format csv -- python3 $root/bench.py $Q "$@" > $out/${tag}_write.log 2>&1 || exit 4
python3 $root/tools/pmc_traffic.py $out/${tag}_fetch $out/${tag}_write $out/${tag}_pmc_traffic.json "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), bench.py $Q $*; one calibration + one timed utterance, contexts 159-283" "$tid" > $out/${tag}_pmc_traffic.txt
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA \
  --kernel-trace -d $out/${tag}_sq -o run --output-format csv -- python3 $root/bench.py $Q "$@" > $out/${tag}_sq.log 2>&1 || exit 5
python3 $root/tools/pmc_sq.py $out/${tag}_sq $out/${tag}_pmc_sq.json "rocprofv3 --pmc SQ_* --kernel-trace, bench.py $Q $*" "$tid" > $out/${tag}_pmc_sq.txt
# the one-voice case (rows share the voice's keys: attn_cascade_kernel): kernel stats + traffic passes of its own
rocprofv3 --kernel-trace --stats -d $out/${tag}_stats1 -o run --output-format csv -- python3 $root/bench.py --quick --voices one "$@" > $out/${tag}_stats1.log 2>&1 || exit 6
cp $(find $out/${tag}_stats1 -name "*kernel_stats.csv" | head -1) $out/${tag}_onevoice_kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $out/${tag}_fetch1 -o run --output-format csv -- python3 $root/bench.py $Q --voices one "$@" > $out/${tag}_fetch1.log 2>&1 || exit 7
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $out/${tag}_write1 -o run --output-format csv -- python3 $root/bench.py $Q --voices one "$@" > $out/${tag}_write1.log 2>&1 || exit 8
python3 $root/tools/pmc_traffic.py $out/${tag}_fetch1 $out/${tag}_write1 $out/${tag}_onevoice_pmc_traffic.json "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), bench.py $Q --voices one $*; one calibration + one timed utterance, contexts 159-283, every row cloned from one voice state" "$tid" > $out/${tag}_onevoice_pmc_traffic.txt
rm -rf $out/${tag}_stats $out/${tag}_fetch $out/${tag}_write $out/${tag}_sq $out/${tag}_stats1 $out/${tag}_fetch1 $out/${tag}_write1   # the raw traces are large; the summaries above are what is kept
head -30 $out/${tag}_pmc_sq.txt; head -12 $out/${tag}_pmc_traffic.txt; head -8 $out/${tag}_onevoice_pmc_traffic.txt
