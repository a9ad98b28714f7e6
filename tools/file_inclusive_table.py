"""Markdown table of DESIGN.md section 7 "File to file" from the JSON tools/file_inclusive.py writes."""
import json
import re
import sys

d = json.load(open(sys.argv[1] if len(sys.argv) > 1 else "profiles/r04_file_inclusive_final.json"))
REF = {"21": "2.56 s (rapidsnark, 1-sig L1, same 2^21 domain: tests/1_sigs_1_batches_5_height/logs/layers_one_two_prove_batch_0.log:16-18)",
       "25": "26.7 s (…batch_0.log:54-56)", "26_l3": '"1 m" (tests/4_sigs_2_batches_12_height/benchmarks.txt:62)'}


def total(run):
    """what the caller waited: the wall time around the process (r04 until the worker process: the client's own clock from
    main to exit, which for keys of 13 GB and more left out 130-165 ms of process dismantling)"""
    return run["wall_s"] * 1e3


print("| shape | files | HBM-resident proof | one-shot `prover` (upload overlapped) | one-shot, `ZKPOA_OVERLAP=0` | server: first call / second call (no tables yet) | server steady, 1 client | 2 clients at a time (per proof) | reference's log (other hardware) |")
print("|---|---|---|---|---|---|---|---|---|")
for spec, r in d["records"].items():
    if "one_shot" not in r:
        continue
    over = sorted(total(x) for x in r["one_shot"] if "OVERLAP" not in x["what"])
    seq = [total(x) for x in r["one_shot"] if "OVERLAP" in x["what"]]
    srv = [total(x) for x in r["server"]]
    steady = sorted(srv[3:])
    conc = r.get("server_concurrent", {})
    res = r["hbm_resident_ms"]
    print("| %s | %.2f GB zkey, %.0f MB wtns | %.1f ms | %.0f–%.0f ms (first run of the box: %.0f) | %.0f ms | %.2f / %.2f s | **%.1f ms** (%.2f ×) | %s | %s |" % (
        r["shape"], r["files"]["zkey_gb"], r["files"]["wtns_gb"] * 1e3, res, over[0], over[-2] if len(over) > 2 else over[-1], over[-1],
        seq[0] if seq else float("nan"), srv[0] / 1e3, srv[1] / 1e3, steady[len(steady) // 2], steady[len(steady) // 2] / res,
        ("**%.1f ms** (%.2f ×)" % (conc["2"]["ms_per_proof"], conc["2"]["ms_per_proof"] / res)) if "2" in conc else "–", REF.get(spec, "")))
