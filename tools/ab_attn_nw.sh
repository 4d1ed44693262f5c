for i in 1 2; do
  for nw in 1 2; do
    PTTS_ATTN_KERNEL_NW=$nw timeout -k 10 200 python bench.py --no-cpu-baseline --no-latency --steps 375 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('nw=$nw', round(d['value'],1), round(d['ms_per_step'],4))"
  done
done
