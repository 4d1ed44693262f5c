for i in 1 2; do
  for v in "" pocket_tts_amd/libptts_u2.so; do
    PTTS_LIB_PATH=$v timeout -k 10 300 python bench.py --no-cpu-baseline --steps 125 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); l=d['latency_b1']; print('lib=${v:-default}', round(d['value'],1), 'b1 step', round(l['b1_ms_per_step'],4), 'first chunk', round(l['first_chunk_ms_p50'],4), 'engine-level', round(l['engine_level_first_chunk_ms_p50'],4))"
  done
done
