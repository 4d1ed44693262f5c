"""Per-kernel duration and inter-kernel gap statistics from a rocprofv3 --kernel-trace CSV.
usage: python tools/trace_gaps.py <dir-with-*_kernel_trace.csv> [skip_fraction]"""
import csv, glob, sys, collections
d = sys.argv[1]
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
rows = rows[int(len(rows) * skip):]
byq = collections.defaultdict(list)
for r in rows:
    byq[r.get("Queue_Id", "0")].append(r)
for q, rs in byq.items():
    if len(rs) < 50:
        continue
    dur = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rs)
    span = int(rs[-1]["End_Timestamp"]) - int(rs[0]["Start_Timestamp"])
    gaps = [int(b["Start_Timestamp"]) - int(a["End_Timestamp"]) for a, b in zip(rs, rs[1:])]
    small = [g for g in gaps if g < 20000]
    print(f"queue {q}: {len(rs)} kernels, busy {dur/1e3:.0f} us of span {span/1e3:.0f} us; gaps<20us: n={len(small)} mean {sum(small)/max(1,len(small))/1e3:.2f} us")
    agg = collections.defaultdict(lambda: [0, 0, 0])
    for i, r in enumerate(rs):
        k = r["Kernel_Name"][:70] + " g" + r.get("Grid_Size_X", "?") + "x" + r.get("Grid_Size_Y", "?")
        a = agg[k]
        a[0] += 1
        a[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        if i + 1 < len(rs):
            g = int(rs[i + 1]["Start_Timestamp"]) - int(r["End_Timestamp"])
            a[2] += g if g < 20000 else 0
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
        print(f"  {a[0]:6d} x {a[1]/a[0]/1e3:7.2f} us  (+gap after {a[2]/a[0]/1e3:5.2f})  total {a[1]/1e3:9.0f}  {k}")
