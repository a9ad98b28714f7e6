"""File-to-file timings of the `prover` executable -- what scripts/g16_prove.sh:248-252 actually sees -- at the
reference's three circuit shapes (SURVEY.md 8 "Sizes"): L1(2) 2^21, L2(2,12) 2^25, L3(2) 2^26 / 13 public.
bench.py's `value` has key and witness resident in HBM; these numbers are NEVER `value`. For each shape:

  * the synthetic key and witness are written as real .zkey / .wtns files (streamed out of HBM);
  * HBM-resident reference: prove with the key and the witness already on the device (tables built), as bench.py;
  * one-shot `prover`: with the staged upload overlapped with the compute (default) and with ZKPOA_OVERLAP=0;
  * resident server (ZKPOA_SERVER): first call (start + upload), second (tables), then the steady state;
  every run with ZKPOA_VERBOSE=1 so the stage breakdown (file map, witness -> HBM, chain, MSMs) is in the record.
  Every proof.json must equal the HBM-resident proof (r = s = 0).

  python tools/file_inclusive.py [21 25 26_l3] [--dir /dev/shm/x] [--out gpurun_out/file_inclusive.json]
"""
import argparse
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

os.environ["ZKPOA_SELFCHECK"] = "0"     # the synthetic key is not a valid trusted setup: its proofs do not verify
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

SHAPES = {"16": (16, 60000, 2), "21": (21, 2083343, 1), "25": (25, 21356921, 2), "26_l3": (26, 52367163, 13),
          "26": (26, 61197000, 1), "27_l3": (27, 100845225, 13), "28_l3": (28, 197801349, 13)}


def run_cli(z, paths, env, tag):
    t0 = time.perf_counter()
    m0 = time.monotonic() * 1e3
    rc = subprocess.run([z.PROVER_BIN] + paths, env=env, capture_output=True, text=True)
    m1 = time.monotonic() * 1e3
    dt = time.perf_counter() - t0
    lines = [l for l in rc.stderr.strip().splitlines() if "WARNING" not in l and "amdgpu.ids" not in l]
    rec = {"what": tag, "wall_s": dt, "rc": rc.returncode, "stderr": lines}
    for l in lines:   # the process's own stamps (same clock): what the caller waited for before main and after it
        if "main entered at" in l:
            a, b = l.split("main entered at")[1].split(", leaving at")
            rec["before_main_ms"] = float(a) - m0
            rec["after_main_ms"] = m1 - float(b.split(")")[0])
            rec["worker_process"] = "worker process" in l
    return rec


def settle(gb):
    """The driver wipes device memory a process gives back, at tens of GB/s, and the next process's large allocations wait
    for it (seen as 4-5 s inside one hipMalloc right after this script freed its ~120 GB of tables: that is this script's
    own footprint, not the prover's). Give the wipe time before the next timed process starts."""
    time.sleep(min(20.0, 1.0 + gb / 12.0))


def pick_dir(need_bytes, want):
    for d in ([want] if want else []) + ["/dev/shm", tempfile.gettempdir()]:
        try:
            os.makedirs(d, exist_ok=True)
            if shutil.disk_usage(d).free > need_bytes * 1.1 + (1 << 30):
                return tempfile.mkdtemp(dir=d, prefix="zkpoa_files_")
        except OSError:
            continue
    return None


def one_shape(z, spec, args):
    import torch
    from zkpoa_amd.synthetic import SyntheticCircuit
    k, m, npub = SHAPES[spec]
    n = 1 << k
    rec = {"shape": "2^%d domain, %d wires, %d public" % (k, m, npub)}
    ctx = z.Context(0)
    t0 = time.perf_counter()
    circ = SyntheticCircuit(z, ctx, k, m, n_public=npub, seed=0x5EED0010, witness_like=True)
    rec["generate_s"] = time.perf_counter() - t0
    zkey_bytes = 4 + circ.n_coef * 44 + m * (64 + 64 + 128) + (m - npub - 1) * 64 + n * 64 + 1024
    d = pick_dir(zkey_bytes + m * 32, args.dir)
    if d is None:
        rec["skipped"] = "no directory with %.1f GB free" % ((zkey_bytes + m * 32) / 1e9)
        circ.close()
        ctx.close()
        return rec
    try:
        zp, wp = os.path.join(d, "c.zkey"), os.path.join(d, "w.wtns")
        t0 = time.perf_counter()
        circ.write_zkey(zp)
        circ.write_wtns(wp)
        rec["files"] = {"dir": d, "fs_free_gb_before": None, "zkey_gb": os.path.getsize(zp) / 1e9,
                        "wtns_gb": os.path.getsize(wp) / 1e9, "write_s": time.perf_counter() - t0}
        # HBM-resident reference (what bench.py's `value` measures): first proof, tables, steady state
        want, _ = circ.prove(0, 0)
        tb = circ.key.precompute()
        circ.prove(0, 0)
        steps = 20 if k <= 21 else 4
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            pts, _ = circ.prove(0, 0)
        rec["hbm_resident_ms"] = (time.perf_counter() - t0) / steps * 1e3
        rec["table_gb"] = tb / 1e9
        assert pts == want
        want_json = z.proof_to_json(want)
        circ.close()
        del circ
        ctx.close()
        torch.cuda.empty_cache()
        settle(rec["table_gb"] + zkey_bytes / 1e9 + 40)
        paths = [zp, wp, os.path.join(d, "proof.json"), os.path.join(d, "public.json")]
        env = dict(os.environ, ZKPOA_R="0", ZKPOA_S="0", ZKPOA_VERBOSE="1")
        env.pop("ZKPOA_SERVER", None)
        runs = []
        for i in range(3):
            runs.append(run_cli(z, paths, env, "one-shot, upload overlapped with the prove (run %d)" % i))
            assert runs[-1]["rc"] == 0, runs[-1]
            assert open(paths[2]).read() == want_json, "one-shot proof differs from the HBM-resident proof"
            settle(zkey_bytes / 1e9 * 2.5)
        runs.append(run_cli(z, paths, dict(env, ZKPOA_OVERLAP="0"), "one-shot, ZKPOA_OVERLAP=0 (upload, then prove)"))
        settle(zkey_bytes / 1e9 * 2.5)
        assert runs[-1]["rc"] == 0, runs[-1]
        assert open(paths[2]).read() == want_json
        rec["one_shot"] = runs
        # resident server
        sock = os.path.join(d, "prover.sock")
        senv = dict(env, ZKPOA_SERVER=sock, ZKPOA_SERVER_IDLE_S="120")
        sruns = []
        try:
            for i in range(7):
                tag = (["server start + key upload", "second call, at once: no tables yet (they are built from idle time)",
                        "third call, after the idle-time table build"][i] if i < 3 else "steady state")
                sruns.append(run_cli(z, paths, senv, "via the resident server, call %d (%s)" % (i, tag)))
                assert sruns[-1]["rc"] == 0, sruns[-1]
                assert open(paths[2]).read() == want_json, "server proof differs from the HBM-resident proof"
                if i == 1:       # leave the server alone until its log says the tables are complete
                    t0 = time.perf_counter()
                    while time.perf_counter() - t0 < 60:
                        time.sleep(0.25)
                        try:
                            if "warm-up proof" in open(sock + ".log").read():
                                break
                        except OSError:
                            pass
                    rec["idle_table_build_s"] = time.perf_counter() - t0
            # two clients at a time, as the reference's parallel batch jobs (scripts/full_workflow.sh:552): the server stages
            # one request's witness while it proves the other's -- the period per proof is what a workflow sees
            import threading
            per_client = 4

            def client(idx, out):
                p2 = [zp, wp, os.path.join(d, "proof_c%d.json" % idx), os.path.join(d, "public_c%d.json" % idx)]
                for _ in range(per_client):
                    r = subprocess.run([z.PROVER_BIN] + p2, env=dict(senv, ZKPOA_VERBOSE=""), capture_output=True, text=True)
                    out.append(r.returncode)
                assert open(p2[2]).read() == want_json
            for nclients in (1, 2, 3):
                outs = [[] for _ in range(nclients)]
                th = [threading.Thread(target=client, args=(i, outs[i])) for i in range(nclients)]
                t0 = time.perf_counter()
                for t in th:
                    t.start()
                for t in th:
                    t.join()
                dt = time.perf_counter() - t0
                assert all(rc == 0 for o in outs for rc in o), outs
                rec.setdefault("server_concurrent", {})[str(nclients)] = {
                    "clients": nclients, "proofs": nclients * per_client, "ms_per_proof": dt / (nclients * per_client) * 1e3}
        finally:
            subprocess.run([z.PROVER_BIN, "--stop-server"], env=senv)
            settle(rec["table_gb"] * 1.5 + zkey_bytes / 1e9 + 40)
        try:
            rec["server_log"] = [l for l in open(sock + ".log").read().splitlines() if "WARNING" not in l][-40:]
        except OSError:
            pass
        rec["server"] = sruns
        steady = sorted(r["wall_s"] for r in sruns[3:])
        rec["server_steady_ms"] = steady[len(steady) // 2] * 1e3
        rec["server_steady_over_resident"] = rec["server_steady_ms"] / rec["hbm_resident_ms"]
    finally:
        shutil.rmtree(d, ignore_errors=True)
    return rec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("shapes", nargs="*", default=["21", "25", "26_l3"])
    ap.add_argument("--dir", default=None)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "file_inclusive.json"))
    args = ap.parse_args()
    import torch  # noqa: F401  (before the library: one HSA runtime)
    from __graft_entry__ import load_package
    z = load_package()
    out = {"cpus": len(os.sched_getaffinity(0)), "records": {}}
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    for spec in args.shapes:
        print("== shape", spec, flush=True)
        rec = one_shape(z, spec, args)
        out["records"][spec] = rec
        with open(args.out, "w") as f:
            json.dump(out, f, indent=1)
        brief = {kk: v for kk, v in rec.items() if kk in ("hbm_resident_ms", "server_steady_ms", "server_steady_over_resident", "skipped")}
        brief["one_shot_s"] = [round(r["wall_s"], 3) for r in rec.get("one_shot", [])]
        brief["server_s"] = [round(r["wall_s"], 3) for r in rec.get("server", [])]
        brief["server_concurrent_ms_per_proof"] = {k: round(v["ms_per_proof"], 1) for k, v in rec.get("server_concurrent", {}).items()}
        print(json.dumps(brief), flush=True)


if __name__ == "__main__":
    main()
