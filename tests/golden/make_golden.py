"""Generates tests/golden/gen/* from the Python big-int oracle (oracle/py). Committed together with
its outputs so every vector can be regenerated:  python tests/golden/make_golden.py

Vectors (SURVEY.md 8c "golden vectors to commit"): field mul/inv/Montgomery conversions, group-law
edge cases, MSMs incl. 0 / 1 / r-1 scalars and infinity bases, NTTs, H-scalar chains, and two complete
synthetic (zkey, wtns, r, s) -> proof.json / public.json cases in both JSON styles (domain 2^3, 2^7)."""
import json
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.py import bn254 as bn          # noqa: E402
from oracle.py import groth16 as g16       # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "gen")
Q, R, M = bn.Q, bn.R, bn.MONT_R


def hx(b):
    return bytes(b).hex()


def le(x):
    return int(x).to_bytes(32, "little")


def main():
    os.makedirs(OUT, exist_ok=True)
    rng = random.Random(0x5EED)
    vec = {}
    # ---- fields: operands and results as 32-byte LE hex; mul is the Montgomery product a*b/R
    for name, p in (("fq", Q), ("fr", R)):
        special = [0, 1, 2, p - 1, p - 2, (p - 1) // 2, (p + 1) // 2, M % p, (M * M) % p, 1 << 253]
        a = special + [rng.randrange(p) for _ in range(54)]
        b = list(reversed(special)) + [rng.randrange(p) for _ in range(54)]
        Ri = pow(M, -1, p)
        vec[name] = {
            "a": [hx(le(x)) for x in a], "b": [hx(le(x)) for x in b],
            "mont_mul": [hx(le(x * y * Ri % p)) for x, y in zip(a, b)],
            "add": [hx(le((x + y) % p)) for x, y in zip(a, b)],
            "sub": [hx(le((x - y) % p)) for x, y in zip(a, b)],
            "to_mont": [hx(le(x * M % p)) for x in a],
            "from_mont": [hx(le(x * Ri % p)) for x in a],
            # inverse inside the Montgomery domain: inv(x) = x^-1 * R^2
            "mont_inv": [hx(le(pow(x, -1, p) * M * M % p)) if x else hx(le(0)) for x in a],
        }
    # ---- group law edge cases (wire format: affine Montgomery, inf = zeros)
    def g1r(): return bn.g1_mul(bn.G1_GEN, rng.randrange(1, R))
    def g2r(): return bn.g2_mul(bn.G2_GEN, rng.randrange(1, R))
    P = [g1r() for _ in range(4)]
    cases1 = [(P[0], P[1]), (P[2], P[2]), (P[3], bn.ec_neg(P[3], bn.FQ)), (None, P[0]), (P[1], None), (None, None)]
    vec["g1_add"] = {"a": [hx(g16.g1_to_bytes(x)) for x, _ in cases1], "b": [hx(g16.g1_to_bytes(y)) for _, y in cases1],
                     "sum": [hx(g16.g1_to_bytes(bn.g1_add(x, y))) for x, y in cases1]}
    P2 = [g2r() for _ in range(4)]
    cases2 = [(P2[0], P2[1]), (P2[2], P2[2]), (P2[3], bn.ec_neg(P2[3], bn.FQ2)), (None, P2[0]), (P2[1], None), (None, None)]
    vec["g2_add"] = {"a": [hx(g16.g2_to_bytes(x)) for x, _ in cases2], "b": [hx(g16.g2_to_bytes(y)) for _, y in cases2],
                     "sum": [hx(g16.g2_to_bytes(bn.g2_add(x, y))) for x, y in cases2]}
    # ---- MSMs
    special_k = [0, 1, 2, R - 1, R - 2, (R - 1) // 2, (R + 1) // 2, 1 << 15, (1 << 16) - 1, 1 << 16, (1 << 128) - 1]
    msms = []
    for n in (1, 2, 33, 100):
        pts = [g1r() if rng.random() > 0.12 else None for _ in range(n)]
        ks = [rng.choice(special_k) if rng.random() < 0.45 else rng.randrange(R) for _ in range(n)]
        msms.append({"group": 1, "n": n, "bases": hx(b"".join(g16.g1_to_bytes(x) for x in pts)),
                     "scalars": hx(b"".join(le(k) for k in ks)),
                     "result": hx(g16.g1_to_bytes(bn.msm_naive(pts, ks, bn.FQ)))})
    for n in (1, 9):
        pts = [g2r() if rng.random() > 0.12 else None for _ in range(n)]
        ks = [rng.choice(special_k) if rng.random() < 0.45 else rng.randrange(R) for _ in range(n)]
        msms.append({"group": 2, "n": n, "bases": hx(b"".join(g16.g2_to_bytes(x) for x in pts)),
                     "scalars": hx(b"".join(le(k) for k in ks)),
                     "result": hx(g16.g2_to_bytes(bn.msm_naive(pts, ks, bn.FQ2)))})
    vec["msm"] = msms
    # ---- NTTs (Montgomery form in and out, natural order; definitional O(n^2) DFT)
    ntts = []
    for k in (1, 3, 6):
        x = [rng.randrange(R) for _ in range(1 << k)]
        ntts.append({"k": k, "in": hx(b"".join(le(v * M % R) for v in x)),
                     "fwd": hx(b"".join(le(v * M % R) for v in bn.ntt_naive(x))),
                     "inv": hx(b"".join(le(v * M % R) for v in bn.ntt_naive(x, inverse=True)))})
    vec["ntt"] = ntts
    with open(os.path.join(OUT, "vectors.json"), "w") as f:
        json.dump(vec, f, indent=0)

    # ---- complete synthetic proofs
    for tag, nVars, nPublic, nCons in (("n8", 10, 2, 4), ("n128", 100, 3, 110)):
        cons, w = g16.random_circuit(rng, nVars, nPublic, nCons)
        tox = {k: rng.randrange(1, R) for k in ("tau", "alpha", "beta", "gamma", "delta")}
        zk, vk = g16.synthetic_setup(nVars, nPublic, cons, tox)
        wt = g16.write_wtns(w)
        r_, s_ = rng.randrange(R), rng.randrange(R)
        proof, pub = g16.prove(zk, wt, r_, s_)
        assert g16.verify(vk, pub, g16.proof_to_obj(proof))
        zo = g16.read_zkey(zk)
        d = os.path.join(OUT, tag)
        os.makedirs(d, exist_ok=True)
        open(os.path.join(d, "circuit.zkey"), "wb").write(zk)
        open(os.path.join(d, "witness.wtns"), "wb").write(wt)
        open(os.path.join(d, "h_scalars.bin"), "wb").write(b"".join(le(v) for v in g16.h_scalars(zo, w)))
        open(os.path.join(d, "proof_rapidsnark.json"), "w").write(g16.proof_json_rapidsnark(proof))
        open(os.path.join(d, "public_rapidsnark.json"), "w").write(g16.public_json_rapidsnark(pub))
        open(os.path.join(d, "proof_snarkjs.json"), "w").write(g16.proof_json_snarkjs(proof))
        open(os.path.join(d, "public_snarkjs.json"), "w").write(g16.public_json_snarkjs(pub))
        json.dump(vk, open(os.path.join(d, "vkey.json"), "w"), indent=1)
        json.dump({"r": str(r_), "s": str(s_)}, open(os.path.join(d, "rs.json"), "w"))
    print("golden vectors written to", OUT)


if __name__ == "__main__":
    main()
