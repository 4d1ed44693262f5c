for i in 1 2; do
  for v in 0 45056 57344; do
    PTTS_TUNE_CACHE= PTTS_CODEC_LDS_TARGET=$v timeout -k 10 300 python bench.py --no-cpu-baseline --no-latency --steps 375 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('live-tuned target=$v', round(d['value'],1), round(d['ms_per_step'],4))"
  done
done
