"""rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes (separate runs) -> profiles/*_pmc_traffic.json.

    python tools/pmc_traffic.py <dir of the FETCH_SIZE run> <dir of the WRITE_SIZE run> <out.json> ["note"] [tune_table_id]

`tune_table_id` (bench.py prints it) stamps the file with the tile table the kernels were chosen from: bench.py only
replays these numbers into `roofline.traffic` when its own run uses the same table.

Per kernel (named as bench.py's profiler names them): launches, HBM-side read / write bytes per launch and their sum.
Units and the gfx950 correction follow MI355X_MICROARCH.md: both counters are in KiB; FETCH_SIZE reports exactly
half of the bytes of wide coalesced reads on gfx950 (calibrated here too: tests/hip/fetch_calib.hip ->
profiles/r01_fetch_size_calibration.csv), so it is doubled; WRITE_SIZE is exact."""
import collections, csv, glob, json, re, sys

PRE = {0: "", 1: "+elu", 2: "+addsilu", 3: "+ln", 4: "+lnmod"}


def norm(name):
    # 7th template argument = weight format (round 2: bool Q8; round 3: int WF: 0 fp32, 1 int8, 2 bf16)
    m = re.search(r"gemm_kernel<(\d+), (\d+), (\d+), (\d+), (\d+), (\d+)(?:, (true|false|\d+))?>", name)
    if m:
        a = [int(x) for x in m.groups()[:6]]
        wf = {"true": "+q8", "1": "+q8", "2": "+b16"}.get(m.group(7) or "0", "")
        return "gemm<%d,%d,%d,%d,%d>%s%s" % (*a[:5], PRE.get(a[5], ""), wf)
    m = re.search(r"gemm_lds_kernel<(\d+), (\d+), (\d+), (\d+)(?:, (\d+))?(?:, (\d+))?>", name)
    if m:
        if m.group(6) and int(m.group(6)) > 0:  # fused residual block: bench.py's profiler calls it resblock<BNT,NT2>
            return "resblock<%s,%s>%s" % (m.group(2), m.group(6), PRE.get(int(m.group(4)), ""))
        return "gemm_lds<%s,%s,%s>%s" % (m.group(1), m.group(2), m.group(3), PRE.get(int(m.group(4)), ""))
    if "attn_decode" in name:
        return "attn_decode"
    m = re.search(r"gemm_h_kernel<(\d+), (\d+), (\d+), (\d+), (\d+)>", name)
    if m:
        return "gemm_h<%s,%s,%s,%s>%s" % (*m.groups()[:4], "+ln" if m.group(5) == "3" else "")
    m = re.match(r"(?:void )?(\w+?)(?:_kernel)?[<(]", name)
    return m.group(1) if m else name


def collect(d, counter):
    acc = collections.defaultdict(lambda: [0, 0.0])
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            name = norm(r["Kernel_Name"])
            if name.startswith(("gemm", "attn", "flow_cluster", "lm_cluster", "resblock")) and not name.startswith("attn_combine"):
                name += "@" + r["Grid_Size"]  # as bench.py's profiler labels them: one label = one grid = one shape class
            a = acc[name]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
    return acc


def main():
    fd, wd, out = sys.argv[1:4]
    note = sys.argv[4] if len(sys.argv) > 4 else ""
    table_id = sys.argv[5] if len(sys.argv) > 5 else None
    F, W = collect(fd, "FETCH_SIZE"), collect(wd, "WRITE_SIZE")
    ks = {}
    for k in sorted(set(F) | set(W)):
        n = max(F[k][0], W[k][0]) or 1
        rd = 2.0 * 1024.0 * F[k][1] / max(F[k][0], 1)
        wr = 1024.0 * W[k][1] / max(W[k][0], 1)
        ks[k] = dict(launches=n, hbm_read_bytes_per_launch=rd, hbm_write_bytes_per_launch=wr, traffic_bytes_per_launch=rd + wr)
    json.dump(dict(note=note, tune_table_id=table_id, kernels=ks), open(out, "w"), indent=1)
    for k, v in sorted(ks.items(), key=lambda kv: -kv[1]["traffic_bytes_per_launch"] * kv[1]["launches"])[:25]:
        print(f"{k:34s} n={v['launches']:5d}  read {v['hbm_read_bytes_per_launch']/1e6:9.2f} MB  write {v['hbm_write_bytes_per_launch']/1e6:8.2f} MB")


if __name__ == "__main__":
    main()
