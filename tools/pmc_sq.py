"""SQ counters per kernel from a `rocprofv3 --pmc SQ_... --kernel-trace --output-format csv` run -> profiles/*_pmc_sq.json.

    python tools/pmc_sq.py <run dir> [out.json] ["note"] [tune_table_id]

Per kernel (labelled as bench.py's profiler labels them, see tools/pmc_traffic.py): launches, mean duration from the
kernel trace of the SAME run, the raw counters per launch, and
  * mfma_busy_frac  = SQ_VALU_MFMA_BUSY_CYCLES / (duration x 2.4 GHz x 1024 SIMDs): share of the chip's matrix-pipe
    cycles the kernel used (MI355X_MICROARCH.md: the counter is in shader cycles, 32 per v_mfma_f32_16x16x4_f32; the
    clock under load is below 2.4 GHz, so this is a lower bound of the in-kernel utilisation);
  * wait_any / wait_inst_any / active_inst_any as fractions of SQ_WAVE_CYCLES (all three in quad-cycles): a wave is
    parked at s_waitcnt / a barrier, stalls at issue (behind the matrix pipe: the useful stall), or issues.
Counters of a profiled run are perturbed by the profiler (serialised dispatches): durations here are NOT the bench's."""
import collections, csv, glob, json, os, sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_traffic import norm  # noqa: E402

CLK, SIMDS = 2.4e9, 1024


def label(r):
    name = norm(r["Kernel_Name"])
    if name.startswith(("gemm", "attn", "flow_cluster", "lm_cluster", "resblock", "stage3")) and not name.startswith("attn_combine"):
        name += "@" + r["Grid_Size"]
    return name


def main():
    d = sys.argv[1]
    out = sys.argv[2] if len(sys.argv) > 2 else None
    note = sys.argv[3] if len(sys.argv) > 3 else ""
    table_id = sys.argv[4] if len(sys.argv) > 4 else None
    ctr = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(set)
    dur = collections.defaultdict(lambda: [0, 0.0])
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = label(r)
            ctr[k][r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Dispatch_Id"] not in disp[k]:  # the counter file repeats a dispatch's timestamps on each of its counter rows
                disp[k].add(r["Dispatch_Id"])
                dur[k][0] += 1
                dur[k][1] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    ks = {}
    for k, c in ctr.items():
        n = len(disp[k]) or 1
        per = {name: v / n for name, v in c.items()}
        avg_ns = dur[k][1] / dur[k][0] if dur[k][0] else None
        wc = per.get("SQ_WAVE_CYCLES") or None
        e = dict(launches=n, avg_us=None if avg_ns is None else avg_ns / 1e3, counters_per_launch=per)
        if avg_ns and "SQ_VALU_MFMA_BUSY_CYCLES" in per:
            e["mfma_busy_frac"] = per["SQ_VALU_MFMA_BUSY_CYCLES"] / (avg_ns * 1e-9 * CLK * SIMDS)
        if wc:
            for src, dst in (("SQ_WAIT_ANY", "wait_any"), ("SQ_WAIT_INST_ANY", "wait_inst_any"), ("SQ_ACTIVE_INST_ANY", "active_inst_any"),
                             ("SQ_ACTIVE_INST_VALU", "active_inst_valu"), ("SQ_ACTIVE_INST_LDS", "active_inst_lds"),
                             ("SQ_INSTS_VALU_MFMA_MOPS_F32", None)):
                if dst and src in per:
                    e[dst + "_of_wave_cycles"] = per[src] / wc
        ks[k] = e
    res = dict(note=note, tune_table_id=table_id, clock_hz_assumed=CLK, simds=SIMDS, kernels=ks)
    if out:
        json.dump(res, open(out, "w"), indent=1)
    tot = sum((v["avg_us"] or 0) * v["launches"] for v in ks.values()) or 1
    for k, v in sorted(ks.items(), key=lambda kv: -(kv[1]["avg_us"] or 0) * kv[1]["launches"])[:30]:
        print(f"{k:36s} n={v['launches']:5d} {v['avg_us'] or 0:8.1f} us  mfma {100 * v.get('mfma_busy_frac', 0):5.1f}%  "
              f"wait {100 * v.get('wait_any_of_wave_cycles', 0):5.1f}%  stall {100 * v.get('wait_inst_any_of_wave_cycles', 0):5.1f}%  "
              f"issue {100 * v.get('active_inst_any_of_wave_cycles', 0):5.1f}%  ({100 * (v['avg_us'] or 0) * v['launches'] / tot:4.1f}% of kernel time)")


if __name__ == "__main__":
    main()
