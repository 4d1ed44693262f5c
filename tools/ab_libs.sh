#!/bin/bash
# A/B/C of several builds of the library on one box: tools/ab_libs.sh "<lib1> <lib2> ..." [bench args...]   ("-" = default build)
libs=$1; shift
for i in 1 2; do
  for v in $libs; do
    p=$v; [ "$v" = "-" ] && p=""
    PTTS_LIB_PATH=$p timeout -k 10 300 python bench.py --no-cpu-baseline --no-latency "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('lib=$v', round(d['value'],1), round(d['ms_per_step'],4))"
  done
done
