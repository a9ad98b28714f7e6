"""Per-queue timeline of the last prove in a rocprofv3 kernel trace (tools/collect: --kernel-trace csv).
usage: python tools/prove_timeline.py <kernel_trace.csv> [min_ms]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
min_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 0.1
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last prove = kernels after the last abc_rows_kernel start (chain start), minus a little
starts = [int(r["Start_Timestamp"]) for r in rows if "abc_rows_kernel" in r["Kernel_Name"]]
t0 = starts[-1] - 3_000_000
sel = [r for r in rows if int(r["Start_Timestamp"]) >= t0]
t0 = int(sel[0]["Start_Timestamp"])
qs = {}
for r in sel:
    qs.setdefault(r["Queue_Id"], []).append(r)
for q, rs in qs.items():
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rs) / 1e6
    print("queue %s: %d kernels, busy %.2f ms, span %.2f .. %.2f ms" % (
        q, len(rs), busy, (int(rs[0]["Start_Timestamp"]) - t0) / 1e6, (int(rs[-1]["End_Timestamp"]) - t0) / 1e6))
    for r in rs:
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
        if d >= min_ms:
            print("    %7.2f +%6.2f  %s" % ((int(r["Start_Timestamp"]) - t0) / 1e6, d,
                                         r["Kernel_Name"].replace("zkpoa::", "")[:70]))
