"""Readable summary of a bench.py JSON line (headline + 'also' legs). usage: python tools/show_bench.py <file>"""
import json, sys
l = json.loads([x for x in open(sys.argv[1]).read().strip().splitlines() if x.startswith("{")][-1])
def show(e, name):
    r = e["roofline"]
    print("%-26s %12.4g %-9s ms/step %8.3f steps %d" % (name, e["value"], e["unit"], e["ms_per_step"], e["steps"]), end=" ")
    if "modmul_per_hash" in r.get("valu", {}):
        print("| tree %.2f ms, %.1f M hashes/s, valu frac %.3f" % (r["kernel_ms"], r["valu"]["hashes_per_s"] / 1e6, r["valu"]["frac"]))
    elif "kernel_ms" in r:
        print("| accum solo %.3f ms (overlapped %.3f) hbm frac %.4f valu %.3f c=%d W=%d msm solo %.2f" % (
            r["kernel_ms"], r["kernel_ms_overlapped"], r["frac"], r["valu"]["frac"], r["valu"]["window_bits"],
            r["valu"]["windows"], r["msm_device_ms_solo"]))
    else:
        v = r.get("valu")
        print("| hbm frac %.4f" % r["frac"], "sum solo %.1f wall %.1f ratio %.2f" % (v["sum_solo_ms"], v["wall_ms"], v["ratio"]) if v else "")
        if v:
            print("     solo:", {k: round(x, 1) for k, x in v["stage_ms_solo"].items()})
        print("     overlapped:", {k: round(x, 1) for k, x in r["phase_ms_overlapped"].items()})
        print("     ", e["config"]["checked"], "| tables GB", e["config"].get("fixed_base_tables_GB"))
show(l, "headline")
if "cpu_baseline" in l:
    c = l["cpu_baseline"]; print("   cpu_baseline %.4g %s on %d cores (%.1f s)" % (c["value"], c["unit"], c["cores"], c["seconds"]))
for e in l.get("also", []):
    show(e, e["workload"]); print("     leg_seconds %.1f" % e["leg_seconds"])
