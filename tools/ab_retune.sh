timeout -k 10 600 python tools/make_tune_cache.py gpurun_out/tune_cache_new.txt 2>&1 | grep -v amdgpu.ids | tail -6
for i in 1 2; do
  for c in profiles/tune_cache_mi355x.txt gpurun_out/tune_cache_new.txt; do
    PTTS_TUNE_CACHE=$c timeout -k 10 300 python bench.py --no-cpu-baseline --no-latency --steps 375 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('cache=$c', round(d['value'],1), round(d['ms_per_step'],4))"
  done
done
