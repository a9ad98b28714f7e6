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
