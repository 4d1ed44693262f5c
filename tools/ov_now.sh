export PTTS_TUNE_CACHE=profiles/tune_cache_mi355x.txt
FLOW_CLUSTER=1 timeout -k 10 300 python tools/overlap_probe.py 64 2>&1 | grep -v amdgpu.ids
