"""Writers of the two input files of `snarkjs zkey new` for the tests of csrc/setup.hip: an iden3 .r1cs from a list of
constraints, and a .ptau "prepared for phase 2" built from KNOWN toxic waste (tau, alpha, beta) with the oracle's
fixed-base products -- sections 1-7 as `snarkjs powersoftau` lays them out, 12-15 the Lagrange-form points per level.
Test infrastructure only (the reference has no .r1cs / .ptau fixtures small enough to commit)."""
import struct

from oracle import c_oracle as co
from oracle.py import bn254 as bn
from oracle.py import groth16 as g16

R, Q = bn.R, bn.Q


def _le32(x):
    return int(x).to_bytes(32, "little")


def write_r1cs(n_wires, n_public, constraints, n_outputs=0):
    hdr = struct.pack("<I", 32) + _le32(R) + struct.pack("<IIIIQI", n_wires, n_outputs, n_public - n_outputs,
                                                        n_wires - n_public - 1, n_wires, len(constraints))
    body = bytearray()
    for lcs in constraints:
        for lc in lcs:
            body += struct.pack("<I", len(lc))
            for s, v in lc.items():                    # file order = insertion order (not sorted)
                body += struct.pack("<I", s) + _le32(v % R)
    w2l = b"".join(struct.pack("<Q", i) for i in range(n_wires))
    return g16.write_binfile("r1cs", 1, [(1, hdr), (2, bytes(body)), (3, w2l)])


def write_ptau(power, tau, alpha, beta, threads=8):
    g1 = lambda ks: co.fixed_base_g1(b"".join(_le32(k % R) for k in ks), threads)
    g2 = lambda ks: co.fixed_base_g2(b"".join(_le32(k % R) for k in ks), threads)
    n = 1 << power
    pw = [1]
    for _ in range(2 * n - 2):
        pw.append(pw[-1] * tau % R)
    hdr = struct.pack("<I", 32) + _le32(Q) + struct.pack("<II", power, power)
    lag = {lvl: g16.fr_lagrange_at(tau, 1 << lvl) for lvl in range(power + 2)}
    sec12 = b"".join(g1(lag[lvl]) for lvl in range(power + 2))
    sec13 = b"".join(g2(lag[lvl]) for lvl in range(power + 1))
    sec14 = b"".join(g1([alpha * x for x in lag[lvl]]) for lvl in range(power + 1))
    sec15 = b"".join(g1([beta * x for x in lag[lvl]]) for lvl in range(power + 1))
    secs = [(1, hdr), (2, g1(pw)), (3, g2(pw[:n])), (4, g1([alpha * x for x in pw[:n]])),
            (5, g1([beta * x for x in pw[:n]])), (6, g2([beta])), (7, struct.pack("<I", 0)),
            (12, sec12), (13, sec13), (14, sec14), (15, sec15)]
    return g16.write_binfile("ptau", 1, secs)


# ---- full-shape inputs with KNOWN discrete logs (tests/test_gpu_setup.py: zkey new at the layer-one shape) -----------
# At 2^21 constraints no CPU setup is affordable, so the "ceremony" is an arithmetic progression of multiples of the
# generators made by the device generator: section sec, point index i (counted over the whole section, all levels) =
# (a_sec + i * b_sec) * G. Every point of the key `zkey new` must produce is then a known multiple of G as well.
PTAU_PROGRESSIONS = {4: (11, 18), 5: (12, 19), 6: (13, 20), 12: (14, 21), 13: (15, 22), 14: (16, 23), 15: (17, 24)}


def r1cs_coefficient_mix(nr, cnt):
    """R1CS-like coefficients as uint64 [cnt, 4] (LE limbs): 45 % one, 25 % r - 1, 20 % small, 7 % powers of two, 3 %
    full width."""
    import numpy as np
    u = nr.random(cnt)
    out = np.zeros((cnt, 4), dtype=np.uint64)
    out[:, 0] = 1
    out[(u >= 0.45) & (u < 0.70)] = np.array([((R - 1) >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)], dtype=np.uint64)
    small = (u >= 0.70) & (u < 0.90)
    out[small, 0] = nr.integers(2, 1 << 16, size=int(small.sum()), dtype=np.uint64)
    p2 = (u >= 0.90) & (u < 0.97)
    j = nr.integers(0, 252, size=int(p2.sum()))
    t = np.zeros((int(p2.sum()), 4), dtype=np.uint64)
    t[np.arange(len(j)), j // 64] = np.uint64(1) << (j % 64).astype(np.uint64)
    out[p2] = t
    full = u >= 0.97
    f = nr.integers(0, 1 << 63, size=(int(full.sum()), 4), dtype=np.uint64)
    f[:, 3] &= np.uint64((1 << 60) - 1)
    out[full] = f
    return out


def write_full_shape_inputs(ctx, k, m, directory, seed=3, n_public=1):
    """Writes <directory>/c.r1cs (2^k - 2 constraints (w[a] + coef * w[b]) * w[d] = w[e] over m wires) and
    <directory>/pot.ptau (power k, prepared for phase 2, points = PTAU_PROGRESSIONS). Returns the term lists the
    expectation needs: dict matrix -> (constraint index, signal, coefficient limbs) numpy arrays."""
    import numpy as np
    import torch
    nr = np.random.default_rng(seed)
    n = 1 << k
    n_cons = n - 2
    a = nr.integers(1, m, size=n_cons, dtype=np.uint32)
    b = nr.integers(1, m, size=n_cons, dtype=np.uint32)
    b[a == b] = 0
    d = nr.integers(0, m, size=n_cons, dtype=np.uint32)
    e = nr.integers(0, m, size=n_cons, dtype=np.uint32)
    kb = r1cs_coefficient_mix(nr, n_cons)
    one = np.zeros((n_cons, 4), dtype=np.uint64)
    one[:, 0] = 1
    rec = np.zeros((n_cons, 4 + 36 + 36 + 4 + 36 + 4 + 36), dtype=np.uint8)

    def put_u32(col, arr):
        rec[:, col:col + 4] = arr.astype("<u4").view(np.uint8).reshape(-1, 4)
    put_u32(0, np.full(n_cons, 2, dtype=np.uint32)); put_u32(4, a); rec[:, 8] = 1
    put_u32(40, b); rec[:, 44:76] = kb.view(np.uint8).reshape(n_cons, 32)
    put_u32(76, np.full(n_cons, 1, dtype=np.uint32)); put_u32(80, d); rec[:, 84] = 1
    put_u32(116, np.full(n_cons, 1, dtype=np.uint32)); put_u32(120, e); rec[:, 124] = 1
    hdr = struct.pack("<I", 32) + _le32(R) + struct.pack("<IIIIQI", m, 0, n_public, m - n_public - 1, m, n_cons)
    body = rec.tobytes()
    with open(directory + "/c.r1cs", "wb") as f:
        f.write(b"r1cs" + struct.pack("<II", 1, 2))
        f.write(struct.pack("<IQ", 1, len(hdr)) + hdr)
        f.write(struct.pack("<IQ", 2, len(body)))
        f.write(body)

    def pts(group, cnt, sec):
        size = 64 if group == 1 else 128
        t = torch.empty(cnt * size, dtype=torch.uint8, device="cuda")
        (ctx.gen_bases_g1_device if group == 1 else ctx.gen_bases_g2_device)(*PTAU_PROGRESSIONS[sec], 0, cnt, t.data_ptr())
        return t.cpu().numpy().tobytes()
    secs = [(1, struct.pack("<I", 32) + _le32(Q) + struct.pack("<II", k, k)), (2, b""), (3, b""), (4, pts(1, 1, 4)),
            (5, pts(1, 1, 5)), (6, pts(2, 1, 6)), (7, struct.pack("<I", 0)), (12, pts(1, (4 << k) - 1, 12)),
            (13, pts(2, (2 << k) - 1, 13)), (14, pts(1, (2 << k) - 1, 14)), (15, pts(1, (2 << k) - 1, 15))]
    with open(directory + "/pot.ptau", "wb") as f:
        f.write(b"ptau" + struct.pack("<II", 1, len(secs)))
        for sid, payload in secs:
            f.write(struct.pack("<IQ", sid, len(payload)))
            f.write(payload)
    cons = np.arange(n_cons, dtype=np.int64)
    return {"n_cons": n_cons,
            "A": (np.concatenate([cons, cons]), np.concatenate([a, b]).astype(np.int64), np.concatenate([one, kb])),
            "B": (cons, d.astype(np.int64), one), "C": (cons, e.astype(np.int64), one)}
