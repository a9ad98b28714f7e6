"""Kernel timeline of ONE serialised proof (every stage alone on the chip): where a stage's solo time goes --
kernels vs gaps (host round trips, launch latency). Run under `rocprofv3 --kernel-trace --output-format csv`;
tools/trace_timeline.py prints the timeline from the trace."""
import os, sys
os.environ["ZKPOA_SELFCHECK"] = "0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from __graft_entry__ import load_package
z = load_package()
from zkpoa_amd.synthetic import SyntheticCircuit
k = int(sys.argv[1]) if len(sys.argv) > 1 else 21
m = {21: 2083343, 25: 33000000, 26: 66000000}.get(k, (1 << k) - 1000)
ctx = z.Context(0)
circ = SyntheticCircuit(z, ctx, k, m, n_public=1, seed=0x5EED0010, witness_like=True)
if os.environ.get("TRACE_TABLES", "1") == "1":
    circ.prove(0, 0)          # measures the witness's digit density: the A / B / C tables are sized with it
    print("tables: %.2f GB" % (circ.key.precompute() / 1e9))
for _ in range(3):
    circ.prove(0, 0)
ctx.set_option("prove_serial", 1)
for _ in range(3):
    circ.prove(0, 0)
print("stage ms (serial):", "chain %.2f" % ctx.last_ms(3), " ".join("%s %.2f" % (x, ctx.last_ms_lane(i, 0)) for i, x in enumerate("H A B1 B2 C".split())))
circ.close()
