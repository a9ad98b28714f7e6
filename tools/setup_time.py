"""Timing of the phase-2 setup arithmetic (zkpoa_setup_accumulate) at the reference's layer-one shape: 2^21
constraints, 2.08 M signals, 6.3 M coefficients (3 per constraint, as the synthetic prove workload has), with an R1CS-like
coefficient mix (1, -1, small constants, a few powers of two and full-width values) and a hot signal (the constant one).
Points are (a + i b) G from the device generator -- for timing only (parity: tests/test_gpu_setup.py)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from __graft_entry__ import load_package
z = load_package()
R = 21888242871839275222246405745257275088548364400416034343698204186575808495617
k = int(sys.argv[1]) if len(sys.argv) > 1 else 21
n = 1 << k
m = {21: 2083343, 25: 21400000}.get(k, n - 1000)
ctx = z.Context(0)
rng = np.random.default_rng(3)
nnz_a, nnz_b = 2 * n, n
def coefs(cnt):
    u = rng.random(cnt)
    out = np.zeros((cnt, 4), dtype=np.uint64)
    out[:, 0] = 1
    neg = (u >= 0.45) & (u < 0.70)                      # r - 1
    rl = [(R - 1 >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)]
    out[neg] = np.array(rl, dtype=np.uint64)
    small = (u >= 0.70) & (u < 0.90)
    out[small, 0] = rng.integers(2, 1 << 16, size=int(small.sum()), dtype=np.uint64)
    p2 = (u >= 0.90) & (u < 0.97)                       # 2^j, j < 252
    j = rng.integers(0, 252, size=int(p2.sum()))
    t = np.zeros((int(p2.sum()), 4), dtype=np.uint64)
    t[np.arange(len(j)), j // 64] = np.uint64(1) << (j % 64).astype(np.uint64)
    out[p2] = t
    full = u >= 0.97
    f = rng.integers(0, 1 << 63, size=(int(full.sum()), 4), dtype=np.uint64)
    f[:, 3] &= np.uint64((1 << 60) - 1)
    out[full] = f
    return torch.from_numpy(out.view(np.uint8).reshape(-1)).cuda()
def sigs(cnt, hot):
    s = rng.integers(0, m, size=cnt, dtype=np.int64)
    s[rng.random(cnt) < hot] = 0
    return torch.from_numpy(s.astype(np.int32)).cuda()
def pidx(cnt, npts):
    return torch.from_numpy(rng.integers(0, npts, size=cnt, dtype=np.int64).astype(np.int32)).cuda()
g1 = torch.empty(3 * n * 64, dtype=torch.uint8, device="cuda")
g2 = torch.empty(n * 128, dtype=torch.uint8, device="cuda")
ctx.gen_bases_g1_device(12345, 67890, 0, 3 * n, g1.data_ptr())
ctx.gen_bases_g2_device(12345, 67890, 0, n, g2.data_ptr())
cases = [("A  (section 5, G1)", 1, g1, n, nnz_a, 0.02), ("B1 (section 6, G1)", 1, g1, n, nnz_b, 0.02),
         ("B2 (section 7, G2)", 2, g2, n, nnz_b, 0.02), ("IC + C (sections 3 + 8, G1 over 3n points)", 1, g1, 3 * n, 2 * nnz_a + nnz_b, 0.05)]
total = 0.0
for name, grp, pts, npts, nnz, hot in cases:
    c, s, p = coefs(nnz), sigs(nnz, hot), pidx(nnz, npts)
    out = torch.empty(m * (64 if grp == 1 else 128), dtype=torch.uint8, device="cuda")
    best = 1e9
    for _ in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        ctx.setup_accumulate(grp, pts.data_ptr(), npts, c.data_ptr(), p.data_ptr(), s.data_ptr(), nnz, m, out.data_ptr())
        torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    total += best
    print("%-46s %9d entries -> %8d signals: %8.1f ms" % (name, nnz, m, best * 1e3))
print("domain 2^%d: point sections of `snarkjs zkey new` in %.2f s on the device" % (k, total))

# ---- the whole command on files: zkpoa-setup <r1cs> <ptau> <zkey> at the same shape (format-valid inputs: the "ceremony"
# points come from the device generator, the constraints are the synthetic prove workload's: 2 A terms, 1 B term, 1 C term)
if "--files" in sys.argv:
    import struct, subprocess, tempfile
    d = tempfile.mkdtemp()
    n_cons = n - 2
    npub = 1
    t0 = time.perf_counter()
    le32 = lambda x: int(x).to_bytes(32, "little")
    # r1cs: constraint c = (w[a] + k w[b]) * w[d] = w[e]
    a = rng.integers(1, m, size=n_cons, dtype=np.uint32); b = rng.integers(1, m, size=n_cons, dtype=np.uint32)
    b[a == b] = 0
    dd = rng.integers(0, m, size=n_cons, dtype=np.uint32); e = rng.integers(0, m, size=n_cons, dtype=np.uint32)
    rec = np.zeros((n_cons, 4 + 36 + 36 + 4 + 36 + 4 + 36), dtype=np.uint8)
    def put_u32(col, arr): rec[:, col:col + 4] = arr.astype("<u4").view(np.uint8).reshape(-1, 4)
    put_u32(0, np.full(n_cons, 2, dtype=np.uint32)); put_u32(4, a); rec[:, 8] = 1
    put_u32(40, b); rec[:, 44:76] = coefs(n_cons).cpu().numpy().reshape(n_cons, 32)
    put_u32(76, np.full(n_cons, 1, dtype=np.uint32)); put_u32(80, dd); rec[:, 84] = 1
    put_u32(116, np.full(n_cons, 1, dtype=np.uint32)); put_u32(120, e); rec[:, 124] = 1
    hdr = struct.pack("<I", 32) + le32(R) + struct.pack("<IIIIQI", m, 0, npub, m - npub - 1, m, n_cons)
    body = rec.tobytes()
    with open(d + "/c.r1cs", "wb") as f:
        f.write(b"r1cs" + struct.pack("<II", 1, 2))
        f.write(struct.pack("<IQ", 1, len(hdr)) + hdr)
        f.write(struct.pack("<IQ", 2, len(body))); f.write(body)
    # ptau of power k: only what the setup reads carries data (sections 4-6 heads, 12-15); 2, 3, 7 are present but empty
    Q = 21888242871839275222246405745257275088696311157297823662689037894645226208583
    def g1pts(cnt, seed):
        t = torch.empty(cnt * 64, dtype=torch.uint8, device="cuda"); ctx.gen_bases_g1_device(seed, seed + 7, 0, cnt, t.data_ptr()); return t.cpu().numpy().tobytes()
    def g2pts(cnt, seed):
        t = torch.empty(cnt * 128, dtype=torch.uint8, device="cuda"); ctx.gen_bases_g2_device(seed, seed + 7, 0, cnt, t.data_ptr()); return t.cpu().numpy().tobytes()
    secs = [(1, struct.pack("<I", 32) + le32(Q) + struct.pack("<II", k, k)), (2, b""), (3, b""), (4, g1pts(1, 11)),
            (5, g1pts(1, 12)), (6, g2pts(1, 13)), (7, struct.pack("<I", 0)), (12, g1pts((4 << k) - 1, 14)),
            (13, g2pts((2 << k) - 1, 15)), (14, g1pts((2 << k) - 1, 16)), (15, g1pts((2 << k) - 1, 17))]
    with open(d + "/pot.ptau", "wb") as f:
        f.write(b"ptau" + struct.pack("<II", 1, len(secs)))
        for sid, payload in secs:
            f.write(struct.pack("<IQ", sid, len(payload))); f.write(payload)
    print("inputs written in %.1f s: r1cs %.2f GB, ptau %.2f GB" % (time.perf_counter() - t0, os.path.getsize(d + "/c.r1cs") / 1e9, os.path.getsize(d + "/pot.ptau") / 1e9))
    # (at 2^26 the inputs are 62 GB and the two keys 69 GB, all in /dev/shm, i.e. in host memory: give back what this
    # script itself still holds -- the box allows one command 270 GB)
    del secs, body, rec, a, b, dd, e
    import gc; gc.collect()
    g1 = g2 = None; torch.cuda.empty_cache()
    for i in range(2):
        t0 = time.perf_counter()
        rc = subprocess.run([z.SETUP_BIN, "zkey", "new", d + "/c.r1cs", d + "/pot.ptau", d + "/c_0.zkey"], capture_output=True, text=True, env=dict(os.environ, ZKPOA_VERBOSE="1"))
        print("zkpoa-setup zkey new, run %d: %.2f s wall, rc=%d, zkey %.2f GB  %s" % (i, time.perf_counter() - t0, rc.returncode, os.path.getsize(d + "/c_0.zkey") / 1e9 if rc.returncode == 0 else 0, rc.stderr.strip().splitlines()[-1] if rc.stderr.strip() else ""))
        if i == 1:
            print("\n".join("    " + l for l in rc.stderr.splitlines() if "zkey new:" in l))
    t0 = time.perf_counter()
    rc = subprocess.run([z.SETUP_BIN, "zkey", "contribute", d + "/c_0.zkey", d + "/c_final.zkey", "--name=First contributor", "-e=random text for entropy"], capture_output=True, text=True, env=dict(os.environ, ZKPOA_VERBOSE="1"))
    print("zkpoa-setup zkey contribute: %.2f s wall, rc=%d  %s" % (time.perf_counter() - t0, rc.returncode, rc.stderr.strip().splitlines()[-1] if rc.stderr.strip() else ""))
    # the key it wrote must load (section sizes, coordinate and coefficient range checks of the prover's loader)
    key = ctx.load_zkey(open(d + "/c_0.zkey", "rb").read()); key.close()
    print("the prover loads the key")
    import shutil; shutil.rmtree(d)
