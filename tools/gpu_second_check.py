"""Second GPU validation: NTT, h_scalars, full prove, CLI -- against the Python oracle (scratch tool)."""
import os, sys, time, random, json, subprocess, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
from oracle.py import bn254 as bn
from oracle.py import groth16 as g16
z = load_package()
R = bn.R
rng = random.Random(11)
ctx = z.Context(0)
def le(x): return int(x).to_bytes(32, "little")
def rd(b, i=0): return int.from_bytes(b[32*i:32*i+32], "little")
ok_all = True
def check(name, cond):
    global ok_all
    print(("PASS " if cond else "FAIL ") + name, flush=True)
    ok_all = ok_all and bool(cond)
M = bn.MONT_R
# NTT
for k in (0, 1, 2, 5, 10, 11, 12, 13, 14):
    n = 1 << k
    x = [rng.randrange(R) for _ in range(n)]
    xm = b"".join(le(v * M % R) for v in x)
    t0 = time.time(); exp = bn.ntt(x); tpy = time.time() - t0
    out = ctx.ntt(xm, k, False)
    check("ntt fwd k=%d (ms=%.3f)" % (k, ctx.last_ms(2)), all(rd(out, i) == exp[i] * M % R for i in range(n)))
    back = ctx.ntt(out, k, True)
    check("ntt inv k=%d" % k, back == xm)
for k in (16, 20, 22):
    n = 1 << k
    import numpy as np
    nr = np.random.default_rng(k)
    limbs = nr.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64); limbs[:, 3] &= np.uint64((1 << 59) - 1)
    xm = limbs.tobytes()
    out = ctx.ntt(xm, k, False); ms = ctx.last_ms(2)
    back = ctx.ntt(out, k, True)
    okrt = back == xm
    # spot check 2 outputs by direct evaluation
    xs = [rd(xm, i) for i in range(n)] if k <= 16 else None
    good = True
    if xs is not None:
        w = bn.fr_root_of_unity(k); Minv = pow(M, -1, R)
        for idx in (1, n - 3):
            wi = pow(w, idx, R); acc = 0; p = 1
            for v in xs:
                acc = (acc + v * p) % R; p = p * wi % R
            good = good and rd(out, idx) == acc
    check("ntt k=%d roundtrip+spot fwd_ms=%.3f" % (k, ms), okrt and good)

# h_scalars + prove on small random circuits
def make(nVars, nPublic, nCons, seed):
    r = random.Random(seed)
    cons, w = g16.random_circuit(r, nVars, nPublic, nCons)
    tox = {k: r.randrange(1, R) for k in ("tau", "alpha", "beta", "gamma", "delta")}
    zk, vk = g16.synthetic_setup(nVars, nPublic, cons, tox)
    return zk, vk, g16.write_wtns(w), w
for (nVars, nPublic, nCons, seed) in ((12, 2, 5, 1), (40, 1, 60, 2), (300, 3, 250, 3)):
    t0 = time.time()
    zk, vk, wt, w = make(nVars, nPublic, nCons, seed)
    zo = g16.read_zkey(zk)
    secs = g16.read_binfile(zk, "zkey", 2)
    p4, l4 = secs[4][0]
    exp = g16.h_scalars(zo, w)
    power = zo.domainSize.bit_length() - 1
    out = ctx.h_scalars(zk[p4:p4 + l4], b"".join(le(v) for v in w), nVars, power)
    check("h_scalars nVars=%d domain=2^%d" % (nVars, power), all(rd(out, i) == exp[i] for i in range(zo.domainSize)))
    r_, s_ = rng.randrange(R), rng.randrange(R)
    for (rr, ss) in ((0, 0), (r_, s_)):
        proof, pub = g16.prove(zk, wt, rr, ss)
        key = ctx.load_zkey(zk)
        pts, pubb = ctx.prove(key, wt, rr, ss)
        key.close()
        got = {"pi_a": g16.g1_from_bytes(pts, 0), "pi_b": g16.g2_from_bytes(pts, 64), "pi_c": g16.g1_from_bytes(pts, 192)}
        check("prove nVars=%d r=%s bit-exact" % (nVars, "0" if rr == 0 else "rand"), got == proof and [rd(pubb, i) for i in range(nPublic)] == pub)
        check("  json rapidsnark", z.proof_to_json(pts, "rapidsnark") == g16.proof_json_rapidsnark(proof) and z.public_to_json(pubb, "rapidsnark") == g16.public_json_rapidsnark(pub))
        check("  json snarkjs", z.proof_to_json(pts, "snarkjs") == g16.proof_json_snarkjs(proof) and z.public_to_json(pubb, "snarkjs") == g16.public_json_snarkjs(pub))
    check("  verifies", g16.verify(vk, pub, g16.proof_to_obj(proof)))
    print("   (%.1fs)" % (time.time() - t0), flush=True)

# CLI
d = tempfile.mkdtemp()
zk, vk, wt, w = make(40, 1, 60, 5)
open(d + "/c.zkey", "wb").write(zk); open(d + "/w.wtns", "wb").write(wt)
env = dict(os.environ, ZKPOA_R="12345", ZKPOA_S="67890", ZKPOA_VERBOSE="1")
rc = subprocess.run([z.PROVER_BIN, d + "/c.zkey", d + "/w.wtns", d + "/proof.json", d + "/public.json"], env=env, capture_output=True, text=True)
print(rc.stderr.strip())
proof, pub = g16.prove(zk, wt, 12345, 67890)
check("cli exit 0", rc.returncode == 0)
check("cli proof.json bytes", open(d + "/proof.json").read() == g16.proof_json_rapidsnark(proof))
check("cli public.json bytes", open(d + "/public.json").read() == g16.public_json_rapidsnark(pub))
rc = subprocess.run([z.PROVER_BIN, d + "/c.zkey", d + "/w.wtns", d + "/p2.json", d + "/u2.json"], capture_output=True, text=True)
check("cli random r,s verifies", rc.returncode == 0 and g16.verify(vk, json.load(open(d + "/u2.json")), json.load(open(d + "/p2.json"))))
# bad witness length
wt_bad = g16.write_wtns(w[:-1]); open(d + "/bad.wtns", "wb").write(wt_bad)
rc = subprocess.run([z.PROVER_BIN, d + "/c.zkey", d + "/bad.wtns", d + "/p3.json", d + "/u3.json"], capture_output=True, text=True)
check("cli bad witness -> nonzero exit, no output (%s)" % rc.stderr.strip(), rc.returncode != 0 and not os.path.exists(d + "/p3.json"))
print("ALL OK" if ok_all else "SOME FAILED")
sys.exit(0 if ok_all else 1)
