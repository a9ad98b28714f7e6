"""The .r1cs / prepared .ptau writers the GPU setup tests feed to zkpoa_zkey_new (tests/setup_files.py) are themselves
checked here on the CPU: layout of the iden3 containers, section sizes `snarkjs zkey new` relies on (level p of a
Lagrange section starts at point 2^p - 1; tau*G1 carries one level more than the others), and the points against the
oracle's curve arithmetic for a tiny ceremony."""
import random
import struct

from oracle import c_oracle as co
from oracle.py import bn254 as bn
from oracle.py import groth16 as g16
from setup_files import write_ptau, write_r1cs

R = bn.R


def test_r1cs_writer_round_trip():
    rng = random.Random(3)
    cons, w = g16.random_circuit(rng, 20, 2, 30)
    buf = write_r1cs(20, 2, cons)
    secs = g16.read_binfile(buf, "r1cs", 1)
    (hp, hl), (cp, cl) = secs[1][0], secs[2][0]
    assert struct.unpack_from("<I", buf, hp)[0] == 32 and int.from_bytes(buf[hp + 4:hp + 36], "little") == R
    n_wires, n_out, n_pub_in, n_prv, n_labels, n_cons = struct.unpack_from("<IIIIQI", buf, hp + 36)
    assert (n_wires, n_out + n_pub_in, n_cons) == (20, 2, 30) and n_prv == 20 - 2 - 1 and hl == 64
    pos, got = cp, []
    for _ in range(n_cons):
        lcs = []
        for _m in range(3):
            nt = struct.unpack_from("<I", buf, pos)[0]
            pos += 4
            lc = {}
            for _t in range(nt):
                lc[struct.unpack_from("<I", buf, pos)[0]] = int.from_bytes(buf[pos + 4:pos + 36], "little")
                pos += 36
            lcs.append(lc)
        got.append(tuple(lcs))
    assert pos == cp + cl
    assert got == [tuple({s: v % R for s, v in lc.items()} for lc in c) for c in cons]
    # the witness of that circuit satisfies what was written
    dot = lambda lc: sum(v * w[s] for s, v in lc.items()) % R
    assert all(dot(a) * dot(b) % R == dot(c) for a, b, c in got)


def test_ptau_writer_sections_and_points():
    power, tau, alpha, beta = 3, 0x1234567, 0x89ABCDE, 0xF012345
    buf = write_ptau(power, tau, alpha, beta, threads=2)
    secs = {t: lst[0] for t, lst in g16.read_binfile(buf, "ptau", 1).items()}
    n = 1 << power
    assert secs[2][1] == (2 * n - 1) * 64 and secs[3][1] == n * 128 and secs[4][1] == n * 64 and secs[6][1] == 128
    assert secs[12][1] == ((4 << power) - 1) * 64          # levels 0 .. power + 1
    assert secs[13][1] == ((2 << power) - 1) * 128 and secs[14][1] == secs[15][1] == ((2 << power) - 1) * 64
    g1 = lambda ks: co.fixed_base_g1(b"".join(int(k % R).to_bytes(32, "little") for k in ks), 1)
    for lvl in range(power + 2):
        L = g16.fr_lagrange_at(tau, 1 << lvl)
        off = secs[12][0] + ((1 << lvl) - 1) * 64
        assert buf[off:off + 64 * (1 << lvl)] == g1(L)
        assert sum(L) % R == 1                               # a Lagrange basis sums to one
    L = g16.fr_lagrange_at(tau, n)
    off = secs[15][0] + (n - 1) * 64
    assert buf[off:off + 64 * n] == g1([beta * x for x in L])
    assert buf[secs[4][0]:secs[4][0] + 64] == g1([alpha])    # alpha * tau^0 * G1: the key's alpha1
    assert buf[secs[2][0] + 64:secs[2][0] + 128] == g1([tau])
