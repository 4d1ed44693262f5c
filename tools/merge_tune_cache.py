"""Merges a freshly measured tile table into the committed one: shapes of the batch-64 pipeline (row tiles 4 / 64 / 384 /
1920 / 7680, fp32) keep the committed entries - several of those were set by tools/pipeline_tune.py, which judges a tile by
the pipelined step, not by its isolated time - every other shape takes the fresh measurement.
python tools/merge_tune_cache.py <committed> <fresh> <out>"""
import sys

def load(p):
    head, d = [], {}
    for ln in open(p):
        if ln.startswith("#") or not ln.strip():
            head.append(ln)
            continue
        f = ln.split()
        d[tuple(int(x) for x in f[:13])] = int(f[13])
    return head, d

hold, old = load(sys.argv[1])
_, new = load(sys.argv[2])
keep = {4, 64, 384, 1920, 7680}
out = dict(old)
changed = 0
for k, v in new.items():
    if k in old and k[4] in keep and k[12] == 0:
        continue
    if out.get(k) != v:
        changed += 1
    out[k] = v
with open(sys.argv[3], "w") as f:
    f.write("# ptts-tune-version 2\n")
    for k in sorted(out):
        f.write(" ".join(str(x) for x in k) + f" {out[k]}\n")
print(f"{len(out)} entries, {changed} changed or added")
