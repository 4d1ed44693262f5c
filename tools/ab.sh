#!/bin/bash
# A/B runs of the whole-utterance bench on ONE box (boxes differ by +-3 %, arms of one call by +-0.2 %): every arm runs
# twice, interleaved.  One script for what used to be a dozen five-liners (round 2's ab_env / ab_lib / ab_libs /
# ab_retune / ab_merge_cache / ab_attn_nw / ab_ldspad_tuned / ab_lib_b1):
#   tools/ab.sh env   VAR "v1 v2 .."         [bench args]   one environment knob (e.g. PTTS_CODEC_LDS_TARGET "0 57344")
#   tools/ab.sh lib   "<so1> <so2> .."       [bench args]   builds of the library ("-" = the in-tree libptts.so)
#   tools/ab.sh cache "<file1> <file2> .."   [bench args]   tile tables (PTTS_TUNE_CACHE; "" = tune live)
# `AB_B1=1` also prints the batch-1 latency block (runs the latency leg).
mode=$1; shift
case $mode in
  env) var=$1; vals=$2; shift 2 ;;
  lib) var=PTTS_LIB_PATH; vals=$1; shift ;;
  cache) var=PTTS_TUNE_CACHE; vals=$1; shift ;;
  *) echo "usage: tools/ab.sh env|lib|cache ..." >&2; exit 2 ;;
esac
flags="--quick"; [ -n "$AB_B1" ] && flags="--no-cpu-baseline --no-api --no-profile"
for i in 1 2; do
  for v in $vals; do
    p=$v; [ "$v" = "-" ] && p=""
    env $var=$p timeout -k 10 400 python bench.py $flags "$@" 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().split('\n')[-1])
l = d.get('latency_b1')
print('$var=$v', round(d['value'], 1), 'audio-s/s', round(d['ms_per_step'], 4), 'ms/step',
      *(['| b1 step', round(l['b1_ms_per_step'], 4), 'first chunk', round(l['first_chunk_ms_p50'], 4)] if l else []))"
  done
done
