"""Timeline of the LAST proof in a rocprofv3 kernel trace made with tools/trace_serial_prove.py: per kernel start
(relative), duration and the idle gap before it; totals of kernel time and gap time."""
import csv, glob, sys
path = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(path)))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows))
# a serialised proof runs its stages in the order A, B1, B2, C, chain + H, and the A and B stages each start with one
# gather32_kernel: the last proof starts at the second-to-last gather
names = [e[2] for e in ev]
gathers = [i for i, n in enumerate(names) if "gather32" in n]
start = gathers[-2]
t0 = ev[start][0]
busy = gap = 0
prev_end = t0
short = lambda n: n.replace("zkpoa::", "").replace("void ", "").split("(")[0][:58]
for s, e, n in ev[start:]:
    g = max(0, s - prev_end)
    print("%9.1f us  +%7.1f gap  %8.1f us  %s" % ((s - t0) / 1e3, g / 1e3, (e - s) / 1e3, short(n)))
    busy += e - s
    gap += g
    prev_end = max(prev_end, e)
print("kernels %.2f ms, gaps %.2f ms, span %.2f ms" % (busy / 1e6, gap / 1e6, (prev_end - t0) / 1e6))
