import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, numpy as np
from __graft_entry__ import load_package
z = load_package(); ctx = z.Context(0)
for k in (20, 22, 24, 26):
    n = 1 << k
    x = torch.randint(0, 1 << 30, (n * 8,), dtype=torch.int32, device="cuda")
    for inv in (False, True):
        ctx.ntt_device(x.data_ptr(), k, inv)
        ts = []
        for _ in range(3):
            ctx.ntt_device(x.data_ptr(), k, inv); ts.append(ctx.last_ms(2))
        print("ntt 2^%d inverse=%s: %.3f ms (natural order in/out incl. permute)" % (k, inv, min(ts)), flush=True)
