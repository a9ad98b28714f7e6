"""First GPU validation of field / group / MSM kernels against the Python oracle (scratch tool)."""
import os, sys, time, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from __graft_entry__ import load_package
from oracle.py import bn254 as bn
from oracle.py import groth16 as g16
z = load_package()
Q, R = bn.Q, bn.R
rng = random.Random(7)
ctx = z.Context(0)
def le(x): return int(x).to_bytes(32, "little")
def rd(b, i=0): return int.from_bytes(b[32*i:32*i+32], "little")
ok_all = True
def check(name, cond):
    global ok_all
    print(("PASS " if cond else "FAIL ") + name, flush=True)
    ok_all = ok_all and cond

# 1. field ops
for field, p in ((0, Q), (1, R)):
    vals = [0, 1, 2, p-1, p-2, (1<<253), rng.randrange(p)] + [rng.randrange(p) for _ in range(300)]
    a = [rng.choice(vals) for _ in range(1000)]; b = [rng.choice(vals) for _ in range(1000)]
    A = b"".join(le(x) for x in a); B = b"".join(le(x) for x in b)
    Rinv = pow(1<<256, -1, p)
    out = ctx.field_op(field, 0, A, B); check("field%d mul" % field, all(rd(out,i) == a[i]*b[i]*Rinv % p for i in range(1000)))
    out = ctx.field_op(field, 1, A, B); check("field%d add" % field, all(rd(out,i) == (a[i]+b[i]) % p for i in range(1000)))
    out = ctx.field_op(field, 2, A, B); check("field%d sub" % field, all(rd(out,i) == (a[i]-b[i]) % p for i in range(1000)))
    out = ctx.field_op(field, 4, A); check("field%d to_mont" % field, all(rd(out,i) == (a[i]<<256) % p for i in range(1000)))
    out = ctx.field_op(field, 5, A); check("field%d from_mont" % field, all(rd(out,i) == a[i]*Rinv % p for i in range(1000)))
    nz = [x if x else 5 for x in a[:200]]
    out = ctx.field_op(field, 3, b"".join(le(x) for x in nz))
    # inverse in Montgomery domain: inv(aR) = a^-1 R  -> x^-1 * R^2
    check("field%d inv" % field, all(rd(out,i) == pow(nz[i], -1, p) * pow(1<<256, 2, p) % p for i in range(200)))

# 2. group add incl. exceptional cases
def g1_rand(): return bn.g1_mul(bn.G1_GEN, rng.randrange(1, R))
def g2_rand(): return bn.g2_mul(bn.G2_GEN, rng.randrange(1, R))
P = [g1_rand() for _ in range(6)]
pairs = [(P[0], P[1]), (P[2], P[2]), (P[3], bn.ec_neg(P[3], bn.FQ)), (None, P[4]), (P[5], None), (None, None)]
A = b"".join(g16.g1_to_bytes(x) for x, _ in pairs); B = b"".join(g16.g1_to_bytes(y) for _, y in pairs)
out = ctx.group_add(1, A, B)
check("g1 add cases", all(g16.g1_from_bytes(out, 64*i) == bn.g1_add(x, y) for i, (x, y) in enumerate(pairs)))
P2 = [g2_rand() for _ in range(6)]
pairs2 = [(P2[0], P2[1]), (P2[2], P2[2]), (P2[3], bn.ec_neg(P2[3], bn.FQ2)), (None, P2[4]), (P2[5], None), (None, None)]
A = b"".join(g16.g2_to_bytes(x) for x, _ in pairs2); B = b"".join(g16.g2_to_bytes(y) for _, y in pairs2)
out = ctx.group_add(2, A, B)
check("g2 add cases", all(g16.g2_from_bytes(out, 128*i) == bn.g2_add(x, y) for i, (x, y) in enumerate(pairs2)))

# 3. small MSMs vs naive oracle
for n in (0, 1, 2, 33, 200):
    pts = [g1_rand() if rng.random() > 0.1 else None for _ in range(n)]
    special = [0, 1, 2, R-1, R-2, (R-1)//2, (R+1)//2, 1 << 16, (1 << 16) - 1, 1 << 15]
    sc = [rng.choice(special) if rng.random() < 0.4 else rng.randrange(R) for _ in range(n)]
    out = ctx.msm_g1(b"".join(g16.g1_to_bytes(x) for x in pts), b"".join(le(x) for x in sc), n)
    exp = bn.msm_naive(pts, sc, bn.FQ)
    check("msm g1 n=%d" % n, g16.g1_from_bytes(out) == exp)
for n in (1, 17):
    pts = [g2_rand() for _ in range(n)]
    sc = [rng.randrange(R) for _ in range(n)]
    out = ctx.msm_g2(b"".join(g16.g2_to_bytes(x) for x in pts), b"".join(le(x) for x in sc), n)
    check("msm g2 n=%d" % n, g16.g2_from_bytes(out) == bn.msm_naive(pts, sc, bn.FQ2))

# 4. gen_bases + known-dlog MSM at scale
import numpy as np
def dlog_check(group, logn, dist, force_c=0):
    n = 1 << logn
    a, b = rng.randrange(R), rng.randrange(R)
    size = 64 if group == 1 else 128
    d_bases = torch.empty(n * size, dtype=torch.uint8, device="cuda")
    t0 = time.time()
    (ctx.gen_bases_g1_device if group == 1 else ctx.gen_bases_g2_device)(a, b, 0, n, d_bases.data_ptr())
    tgen = time.time() - t0
    if logn <= 12:
        hb = bytes(d_bases[:3*size].cpu().numpy())
        for i in range(3):
            got = g16.g1_from_bytes(hb, 64*i) if group == 1 else g16.g2_from_bytes(hb, 128*i)
            exp = bn.g1_mul(bn.G1_GEN, (a + i*b) % R) if group == 1 else bn.g2_mul(bn.G2_GEN, (a + i*b) % R)
            check("gen_bases g%d i=%d" % (group, i), got == exp)
    nrng = np.random.default_rng(1234 + logn)
    limbs = nrng.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64) * 2 + nrng.integers(0, 2, size=(n, 4), dtype=np.uint64)
    limbs[:, 3] &= np.uint64((1 << 60) - 1)       # < 2^252 < r
    if dist == "witness":
        u = nrng.random(n)
        limbs[u < 0.55, 1:] = 0
        limbs[u < 0.55, 0] = nrng.integers(0, 2, size=int((u < 0.55).sum()), dtype=np.uint64)
        m = (u >= 0.55) & (u < 0.9)
        limbs[m, 1:] = 0
    sc_bytes = limbs.tobytes()
    # expected dlog = sum k_i (a + i b)
    ks = [int.from_bytes(sc_bytes[32*i:32*i+32], "little") for i in range(n)] if n <= (1 << 16) else None
    if ks is None:
        # vectorised: sum k_i and sum i*k_i with python ints over limb columns
        S0 = 0; S1 = 0
        idx = np.arange(n, dtype=object)
        for j in range(4):
            col = limbs[:, j].astype(object)
            S0 += int(col.sum()) << (64*j)
            S1 += int((col * idx).sum()) << (64*j)
    else:
        S0 = sum(ks); S1 = sum(i*k for i, k in enumerate(ks))
    d = (a * S0 + b * S1) % R
    d_sc = torch.from_numpy(np.frombuffer(sc_bytes, dtype=np.uint8).copy()).cuda()
    ctx.set_option("msm_c", force_c)
    fn = ctx.msm_g1_device if group == 1 else ctx.msm_g2_device
    out = fn(d_bases.data_ptr(), d_sc.data_ptr(), n)      # warm (allocates workspace)
    t0 = time.time(); out = fn(d_bases.data_ptr(), d_sc.data_ptr(), n); wall = time.time() - t0
    exp = bn.g1_mul(bn.G1_GEN, d) if group == 1 else bn.g2_mul(bn.G2_GEN, d)
    got = g16.g1_from_bytes(out) if group == 1 else g16.g2_from_bytes(out)
    check("msm g%d 2^%d %s c=%d  dev_ms=%.3f accum_ms=%.3f wall_ms=%.3f gen_s=%.2f" % (
        group, logn, dist, force_c, ctx.last_ms(0), ctx.last_ms(1), wall*1e3, tgen), got == exp)

dlog_check(1, 10, "uniform")
dlog_check(1, 14, "uniform")
dlog_check(1, 14, "witness")
dlog_check(1, 16, "uniform")
dlog_check(1, 16, "uniform", force_c=8)     # many points per bucket -> multi-level path
dlog_check(1, 20, "uniform")
dlog_check(1, 20, "witness")
dlog_check(1, 20, "uniform", force_c=14)
dlog_check(1, 20, "uniform", force_c=13)
dlog_check(2, 10, "uniform")
dlog_check(2, 16, "witness")
dlog_check(2, 18, "uniform")
print("ALL OK" if ok_all else "SOME FAILED")
sys.exit(0 if ok_all else 1)
