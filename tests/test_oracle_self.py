"""Oracle self-consistency (CPU): Python big-int oracle internals, and the C restatement
(oracle/c/zkpoa_oracle.c) against the Python oracle and the committed golden vectors."""
import random

import pytest

from conftest import golden_case, le, rd
from oracle import c_oracle as co
from oracle.py import bn254 as bn
from oracle.py import groth16 as g16

Q, R, M = bn.Q, bn.R, bn.MONT_R


def test_constants():
    assert bn.g1_is_on_curve(bn.G1_GEN) and bn.g2_is_on_curve(bn.G2_GEN)
    assert bn.g1_mul(bn.G1_GEN, R) is None and bn.g2_mul(bn.G2_GEN, R) is None
    w = bn.fr_root_of_unity(28)
    assert pow(w, 1 << 28, R) == 1 and pow(w, 1 << 27, R) == R - 1
    assert w == 19103219067921713944291392827692070036145651957329286315305642004821462161904
    assert M % R == 6350874878119819312338956282401532410528162663560392320966563075034087161851
    assert M % Q == 6350874878119819312338956282401532409788428879151445726012394534686998597021
    assert bn.FR_SHIFT == 25


def test_ntt_matches_definition():
    rng = random.Random(1)
    for k in (0, 1, 3, 5):
        x = [rng.randrange(R) for _ in range(1 << k)]
        assert bn.ntt(x) == bn.ntt_naive(x)
        assert bn.ntt(bn.ntt(x), inverse=True) == x


def test_pairing_bilinear():
    a, b = 1234567, 7654321
    e1 = bn.pairing(bn.g2_mul(bn.G2_GEN, b), bn.g1_mul(bn.G1_GEN, a))
    e2 = bn.FQ12.pow(bn.pairing(bn.G2_GEN, bn.G1_GEN), a * b)
    assert bn.FQ12.eq(e1, e2) and not bn.FQ12.eq(e1, bn.FQ12.one)


@pytest.mark.parametrize("tag", ["n8", "n128"])
def test_golden_proof_regenerates_and_verifies(tag):
    import json
    g = golden_case(tag)
    rs = json.loads(g["rs.json"])
    proof, pub = g16.prove(g["circuit.zkey"], g["witness.wtns"], int(rs["r"]), int(rs["s"]))
    assert g16.proof_json_rapidsnark(proof) == g["proof_rapidsnark.json"]
    assert g16.public_json_rapidsnark(pub) == g["public_rapidsnark.json"]
    assert g16.proof_json_snarkjs(proof) == g["proof_snarkjs.json"]
    assert g16.public_json_snarkjs(pub) == g["public_snarkjs.json"]
    vkey = json.loads(g["vkey.json"])
    assert g16.verify(vkey, pub, g16.proof_to_obj(proof))
    pub_bad = [(pub[0] + 1) % R] + pub[1:]
    assert not g16.verify(vkey, pub_bad, g16.proof_to_obj(proof))


def test_prove_r_s_zero_is_deterministic_part():
    """r = s = 0: pi_a = alpha + sum w A, pi_b = beta + sum w B2."""
    g = golden_case("n8")
    zk = g16.read_zkey(g["circuit.zkey"])
    _, w = g16.read_wtns(g["witness.wtns"])
    proof, _ = g16.prove(g["circuit.zkey"], g["witness.wtns"], 0, 0)
    assert proof["pi_a"] == bn.g1_add(bn.msm_naive(zk.A, w, bn.FQ), zk.alpha1)
    assert proof["pi_b"] == bn.g2_add(bn.msm_naive(zk.B2, w, bn.FQ2), zk.beta2)


def test_prove_rejects_wrong_witness_length():
    g = golden_case("n8")
    _, w = g16.read_wtns(g["witness.wtns"])
    with pytest.raises(ValueError, match="Invalid witness length"):
        g16.prove(g["circuit.zkey"], g16.write_wtns(w[:-1]), 0, 0)


# ---- C restatement vs golden vectors / Python oracle ------------------------------------------------
def _cat(hexes):
    return b"".join(bytes.fromhex(h) for h in hexes)


@pytest.mark.parametrize("name,field", [("fq", 0), ("fr", 1)])
def test_c_field_golden(vectors, name, field):
    v = vectors[name]
    a, b = _cat(v["a"]), _cat(v["b"])
    assert co.field_op(field, 0, a, b) == _cat(v["mont_mul"])
    assert co.field_op(field, 1, a, b) == _cat(v["add"])
    assert co.field_op(field, 2, a, b) == _cat(v["sub"])
    assert co.field_op(field, 4, a) == _cat(v["to_mont"])
    assert co.field_op(field, 5, a) == _cat(v["from_mont"])
    # inverse of zero is zero under Fermat inversion; the vectors encode that
    assert co.field_op(field, 3, a) == _cat(v["mont_inv"])


def test_c_group_add_golden(vectors):
    for key, grp in (("g1_add", 1), ("g2_add", 2)):
        v = vectors[key]
        assert co.group_add(grp, _cat(v["a"]), _cat(v["b"])) == _cat(v["sum"])


def test_c_msm_golden(vectors):
    for m in vectors["msm"]:
        fn = co.msm_g1 if m["group"] == 1 else co.msm_g2
        for threads in (1, 3):
            assert fn(bytes.fromhex(m["bases"]), bytes.fromhex(m["scalars"]), m["n"], threads).hex() == m["result"]


def test_c_ntt_golden(vectors):
    for t in vectors["ntt"]:
        assert co.ntt(bytes.fromhex(t["in"]), t["k"]).hex() == t["fwd"]
        assert co.ntt(bytes.fromhex(t["in"]), t["k"], inverse=True).hex() == t["inv"]


@pytest.mark.parametrize("tag", ["n8", "n128"])
def test_c_prove_golden(tag):
    import json
    g = golden_case(tag)
    rs = json.loads(g["rs.json"])
    zk = g16.read_zkey(g["circuit.zkey"])
    secs = g16.read_binfile(g["circuit.zkey"], "zkey", 2)
    p4, l4 = secs[4][0]
    _, w = g16.read_wtns(g["witness.wtns"])
    k = zk.domainSize.bit_length() - 1
    assert co.h_scalars(g["circuit.zkey"][p4:p4 + l4], b"".join(le(x) for x in w), zk.nVars, k) == g["h_scalars.bin"]
    pts, pub = co.prove(g["circuit.zkey"], g["witness.wtns"], int(rs["r"]), int(rs["s"]), nthreads=2)
    proof = {"pi_a": g16.g1_from_bytes(pts, 0), "pi_b": g16.g2_from_bytes(pts, 64), "pi_c": g16.g1_from_bytes(pts, 192)}
    assert g16.proof_json_rapidsnark(proof) == g["proof_rapidsnark.json"]
    assert g16.public_json_rapidsnark([rd(pub, i) for i in range(zk.nPublic)]) == g["public_rapidsnark.json"]


def test_c_fixed_base_and_mid_size_circuit():
    """A 2^10-constraint circuit built with the C fixed-base generator: C prove == Python prove with
    C MSMs plugged in, and the proof verifies under the fixture-pinned verifier."""
    rng = random.Random(77)
    ks = [0, 1, R - 1] + [rng.randrange(R) for _ in range(5)]
    out = co.fixed_base_g1(b"".join(le(k) for k in ks))
    assert all(g16.g1_from_bytes(out, 64 * i) == bn.g1_mul(bn.G1_GEN, k) for i, k in enumerate(ks))
    out = co.fixed_base_g2(b"".join(le(k) for k in ks[:4]))
    assert all(g16.g2_from_bytes(out, 128 * i) == bn.g2_mul(bn.G2_GEN, k) for i, k in enumerate(ks[:4]))
    nVars, nPublic, nCons = 900, 2, 1000
    cons, w = g16.random_circuit(rng, nVars, nPublic, nCons)
    tox = {k: rng.randrange(1, R) for k in ("tau", "alpha", "beta", "gamma", "delta")}
    zk, vk = g16.synthetic_setup(nVars, nPublic, cons, tox,
                                 g1_batch=lambda s: co.fixed_base_g1(b"".join(le(k) for k in s), 4),
                                 g2_batch=lambda s: co.fixed_base_g2(b"".join(le(k) for k in s), 4))
    wt = g16.write_wtns(w)
    r_, s_ = rng.randrange(R), rng.randrange(R)
    pts, pub = co.prove(zk, wt, r_, s_, nthreads=4)
    proof = {"pi_a": g16.g1_from_bytes(pts, 0), "pi_b": g16.g2_from_bytes(pts, 64), "pi_c": g16.g1_from_bytes(pts, 192)}
    assert g16.verify(vk, [rd(pub, i) for i in range(nPublic)], g16.proof_to_obj(proof))


# ---- quotient identity (full-size pi_c checks): pinned on the golden H scalars ----------------------------------
@pytest.mark.parametrize("tag,k", [("n8", 3), ("n128", 7)])
def test_quotient_identity_on_golden_h_scalars(tag, k):
    """orc_quotient_check accepts the golden H scalars (made by the big-int Python oracle through iNTT / coset /
    NTT) at random points, on 1 and 3 threads, and rejects a one-bit change and the scalars of another witness."""
    from conftest import golden_case
    g = golden_case(tag)
    z = g["circuit.zkey"]
    secs = g16.read_binfile(z, "zkey", 2)
    p4, l4 = secs[4][0]
    coeffs = z[p4:p4 + l4]
    _, wit = g16.read_wtns(g["witness.wtns"])
    wb = b"".join(w.to_bytes(32, "little") for w in wit)
    h = g["h_scalars.bin"]
    rng = random.Random(1)
    R = bn.R
    for nt in (1, 3):
        assert co.quotient_check(coeffs, wb, len(wit), k, h, rng.randrange(R), nt)
    bad = bytearray(h)
    bad[33] ^= 1
    assert not co.quotient_check(coeffs, wb, len(wit), k, bytes(bad), rng.randrange(R), 2)
    # an unsatisfying witness still has a quotient (C_T = A_T o B_T by construction): its own scalars pass, the
    # other witness's do not
    wit2 = list(wit)
    wit2[3] = (wit2[3] + 5) % R
    wb2 = b"".join(w.to_bytes(32, "little") for w in wit2)
    h2 = co.h_scalars(coeffs, wb2, len(wit), k)
    assert h2 != h
    assert co.quotient_check(coeffs, wb2, len(wit), k, h2, rng.randrange(R), 2)
    assert not co.quotient_check(coeffs, wb2, len(wit), k, h, rng.randrange(R), 2)
    # numpy buffers are taken without copying
    import numpy as np
    assert co.quotient_check(np.frombuffer(coeffs, dtype=np.uint8), np.frombuffer(wb, dtype=np.uint64).reshape(-1, 4),
                             len(wit), k, np.frombuffer(h, dtype=np.uint64).reshape(-1, 4), rng.randrange(R), 2)
