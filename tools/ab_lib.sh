#!/bin/bash
# A/B of two builds of the library on one box: tools/ab_lib.sh <other .so> [bench args...]
other=$1; shift
for i in 1 2; do
  for v in "" "$other"; do
    PTTS_LIB_PATH=$v timeout -k 10 300 python bench.py --no-cpu-baseline --no-latency "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('lib=${v:-default}', round(d['value'],1), round(d['ms_per_step'],4))"
  done
done
