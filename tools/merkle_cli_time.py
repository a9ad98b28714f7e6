"""File-to-file time of the `merkle-tree` executable (scripts/full_workflow.sh:371-380) on a 10 M-line anonymity set --
the size the reference's Rust binary quotes "2.5 hrs" for (scripts/merkle_tree.rs:3-5).
  python tools/merkle_cli_time.py [--rows 10000000] [--dir /dev/shm]"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=10_000_000)
    ap.add_argument("--dir", default="/dev/shm")
    args = ap.parse_args()
    import numpy as np
    from __graft_entry__ import load_package
    z = load_package()
    d = os.path.join(args.dir, "zkpoa_merkle_time")
    os.makedirs(d, exist_ok=True)
    csv = os.path.join(d, "anonymity_set.csv")
    nr = np.random.default_rng(7)
    t0 = time.perf_counter()
    owned = {}
    with open(csv, "w") as f:
        f.write("address,eth_balance\n")
        step = 1 << 18
        for lo in range(0, args.rows, step):
            cnt = min(step, args.rows - lo)
            hi64 = nr.integers(0, 1 << 63, size=(cnt, 3), dtype=np.uint64)
            bal = nr.integers(0, 1 << 62, size=cnt, dtype=np.uint64)
            out = []
            for i in range(cnt):
                a = (int(hi64[i, 0]) << 96) | (int(hi64[i, 1]) << 33) | int(hi64[i, 2] >> 30)
                a &= (1 << 160) - 1
                out.append("0x%040x,%d\n" % (a, int(bal[i])))
                if (lo + i) in (5, args.rows // 2, args.rows - 1):
                    owned[lo + i] = (a, int(bal[i]))
            f.write("".join(out))
    poa = {"accountAttestations": [{"accountData": {"address": {"__bigint__": str(a)}, "balance": {"__bigint__": str(b)}}}
                                   for _, (a, b) in sorted(owned.items())]}
    json.dump(poa, open(os.path.join(d, "poa.json"), "w"))
    print("wrote %d rows, %.0f MB in %.0f s" % (args.rows, os.path.getsize(csv) / 1e6, time.perf_counter() - t0), flush=True)
    try:
        for threads in (None, "1"):
            env = dict(os.environ, ZKPOA_VERBOSE="1")
            if threads:
                env["ZKPOA_MERKLE_THREADS"] = threads
            for i in range(2):
                t0 = time.perf_counter()
                rc = subprocess.run([z.MERKLE_BIN, "-a", csv, "-p", os.path.join(d, "poa.json"), "-o", d], env=env,
                                    capture_output=True, text=True)
                dt = time.perf_counter() - t0
                assert rc.returncode == 0, rc.stderr
                print("merkle-tree, %d rows, %s: %.2f s  (%s)" % (args.rows, "parser on 1 thread" if threads else "parser on all threads",
                                                                    dt, rc.stdout.strip().splitlines()[-1][:60]), flush=True)
                for l in rc.stderr.splitlines():
                    if l.startswith("merkle-tree:"):
                        print("    " + l, flush=True)
    finally:
        for f in os.listdir(d):
            os.remove(os.path.join(d, f))
        os.rmdir(d)


if __name__ == "__main__":
    main()
