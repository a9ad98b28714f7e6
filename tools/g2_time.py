import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, numpy as np
from __graft_entry__ import load_package
z = load_package(); ctx = z.Context(0)
for group, logn in ((2, 18), (2, 20), (2, 22), (1, 22)):
    n = 1 << logn
    size = 64 if group == 1 else 128
    d_bases = torch.empty(n * size, dtype=torch.uint8, device="cuda")
    (ctx.gen_bases_g1_device if group == 1 else ctx.gen_bases_g2_device)(123, 456, 0, n, d_bases.data_ptr())
    nr = np.random.default_rng(1)
    limbs = nr.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64); limbs[:, 3] &= np.uint64((1 << 59) - 1)
    d_sc = torch.from_numpy(limbs.view(np.uint8).reshape(-1).copy()).cuda()
    fn = ctx.msm_g1_device if group == 1 else ctx.msm_g2_device
    fn(d_bases.data_ptr(), d_sc.data_ptr(), n)
    best = (1e9, 0)
    for _ in range(3):
        fn(d_bases.data_ptr(), d_sc.data_ptr(), n)
        best = min(best, (ctx.last_ms(0), ctx.last_ms(1)))
    print("G%d MSM 2^%d uniform: device %.2f ms, accumulate %.2f ms" % (group, logn, best[0], best[1]), flush=True)
