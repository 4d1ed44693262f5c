#!/bin/bash
# A/B of one environment knob on one box: tools/ab_env.sh VAR "v1 v2" [bench args...]
var=$1; vals=$2; shift 2
for i in 1 2; do
  for v in $vals; do
    env $var=$v timeout -k 10 300 python bench.py --no-cpu-baseline --no-latency "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$var=$v', round(d['value'],1), round(d['ms_per_step'],4))"
  done
done
