export PTTS_TUNE_CACHE=profiles/tune_cache_mi355x.txt
for v in 0 45056 57344; do
  echo "== PTTS_CODEC_LDS_TARGET=$v"
  FLOW_CLUSTER=1 PTTS_CODEC_LDS_TARGET=$v timeout -k 10 300 python tools/overlap_probe.py 64 2>&1 | grep -v amdgpu.ids
done
