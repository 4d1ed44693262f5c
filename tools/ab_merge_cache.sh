timeout -k 10 600 python tools/make_tune_cache.py gpurun_out/tune_cache_fresh.txt 2>&1 | grep -v amdgpu.ids | tail -3
python tools/merge_tune_cache.py profiles/tune_cache_mi355x.txt gpurun_out/tune_cache_fresh.txt gpurun_out/tune_cache_merged.txt
for i in 1 2; do
  for c in profiles/tune_cache_mi355x.txt gpurun_out/tune_cache_merged.txt; do
    PTTS_TUNE_CACHE=$c timeout -k 10 300 python bench.py --no-cpu-baseline --steps 125 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); l=d['latency_b1']; print('cache=$c', round(d['value'],1), 'b1 step', round(l['b1_ms_per_step'],4), 'first chunk', round(l['first_chunk_ms_p50'],4))"
  done
done
