"""Pins the oracle against the reference's own committed fixtures:
  * proof.json / public.json byte formats (rapidsnark style: tests/*/layer_*/**; snarkjs style:
    experiments/scripts/groth16_input_prep) -- copied as data under tests/golden/ref/
  * verifier KATs: the 5 proofs of tests/4_sigs_2_batches_12_height verify against *_vkey.json;
    the 1_sigs L1/L2 proofs verify against the vkey pieces embedded in their sanitized_proof.json
  * pairing KATs: negalfa1xbeta2 (sanitize_groth16_proof.py:31-37,63) and vk_alphabeta_12.
The reference's prove call site is scripts/g16_prove.sh:248-259, its verify call site
scripts/g16_verify.sh:213-216."""
import json
import os

import pytest

from conftest import GOLDEN
from oracle.py import bn254 as bn
from oracle.py import groth16 as g16

REF = os.path.join(GOLDEN, "ref")
CASES_4SIG = [
    ("4_sigs_2_batches_12_height__layer_one__batch_0", "layer_one_vkey.json"),
    ("4_sigs_2_batches_12_height__layer_one__batch_1", "layer_one_vkey.json"),
    ("4_sigs_2_batches_12_height__layer_two__batch_0", "layer_two_vkey.json"),
    ("4_sigs_2_batches_12_height__layer_two__batch_1", "layer_two_vkey.json"),
    ("4_sigs_2_batches_12_height__layer_three", "layer_three_vkey.json"),
]
ALL_DIRS = sorted(d for d in os.listdir(REF) if os.path.isdir(os.path.join(REF, d)) and d not in ("snarkjs_style", "merkle"))


def _load(d, name):
    with open(os.path.join(REF, d, name)) as f:
        return f.read()


def _proof_points(obj):
    return {"pi_a": g16.g1_from_obj(obj["pi_a"]), "pi_b": g16.g2_from_obj(obj["pi_b"]),
            "pi_c": g16.g1_from_obj(obj["pi_c"])}


@pytest.mark.parametrize("d", ALL_DIRS)
def test_rapidsnark_json_bytes(d):
    raw = _load(d, "proof.json")
    assert g16.proof_json_rapidsnark(_proof_points(json.loads(raw))) == raw
    rawp = _load(d, "public.json")
    assert g16.public_json_rapidsnark(json.loads(rawp)) == rawp
    assert not raw.endswith("\n") and not rawp.endswith("\n")


def test_snarkjs_json_bytes():
    raw = _load("snarkjs_style", "proof.json")
    assert g16.proof_json_snarkjs(_proof_points(json.loads(raw))) == raw
    rawp = _load("snarkjs_style", "public.json")
    assert g16.public_json_snarkjs(json.loads(rawp)) == rawp


@pytest.mark.parametrize("d", ALL_DIRS)
def test_fixture_points_on_curve(d):
    pts = _proof_points(json.loads(_load(d, "proof.json")))
    assert bn.g1_is_on_curve(pts["pi_a"]) and bn.g1_is_on_curve(pts["pi_c"]) and bn.g2_is_on_curve(pts["pi_b"])


@pytest.mark.parametrize("d,vk", CASES_4SIG)
def test_verifier_kat(d, vk):
    vkey = json.load(open(os.path.join(REF, vk)))
    proof = json.loads(_load(d, "proof.json"))
    public = json.loads(_load(d, "public.json"))
    assert g16.verify(vkey, public, proof)
    bad = list(public)
    bad[0] = str((int(bad[0]) + 1) % bn.R)
    assert not g16.verify(vkey, bad, proof)


def _limbs(v):          # 6 x 43-bit limbs (sanitize_groth16_proof.py:113-114)
    return sum(int(x) << (43 * i) for i, x in enumerate(v))


def _fq12_from_sanitized(arr):
    """negalfa1xbeta2 layout: 6 Fq2 elements [c_i + 9 c_{i+6}, c_{i+6}], i = 0..5."""
    c = [0] * 12
    for i in range(6):
        a, b = _limbs(arr[i][0]), _limbs(arr[i][1])
        c[i + 6] = b % bn.Q
        c[i] = (a - 9 * b) % bn.Q
    return tuple(c)


SANITIZED = [d for d in ALL_DIRS if os.path.exists(os.path.join(REF, d, "sanitized_proof.json"))]


@pytest.mark.parametrize("d", SANITIZED)
def test_sanitized_proof_recombines_and_verifies(d):
    """sanitized_proof.json = (vkey pieces, -A, B, C, public) in 43-bit limbs: recombine, check against
    proof.json, and verify e(-A,B) e(vk_x,gamma) e(C,delta) == negalfa1xbeta2^-1... i.e. product with
    e(alpha,beta) is one, using only what the file holds."""
    s = json.loads(_load(d, "sanitized_proof.json"))
    pts = _proof_points(json.loads(_load(d, "proof.json")))
    negA = (_limbs(s["negpa"][0]), _limbs(s["negpa"][1]))
    assert negA == bn.ec_neg(pts["pi_a"], bn.FQ)
    B = ((_limbs(s["pb"][0][0]), _limbs(s["pb"][0][1])), (_limbs(s["pb"][1][0]), _limbs(s["pb"][1][1])))
    assert B == pts["pi_b"]
    C = (_limbs(s["pc"][0]), _limbs(s["pc"][1]))
    assert C == pts["pi_c"]
    gamma2 = tuple((_limbs(s["gamma2"][i][0]), _limbs(s["gamma2"][i][1])) for i in range(2))
    delta2 = tuple((_limbs(s["delta2"][i][0]), _limbs(s["delta2"][i][1])) for i in range(2))
    IC = [(_limbs(p[0]), _limbs(p[1])) for p in s["IC"]]
    pub = [int(v) for v in s["pubInput"]]
    assert [str(v) for v in pub] == json.loads(_load(d, "public.json"))
    vk_x = IC[0]
    for v, P in zip(pub, IC[1:]):
        vk_x = bn.g1_add(vk_x, bn.g1_mul(P, v))
    f = bn.FQ12.one
    for P1, Q2 in ((negA, B), (vk_x, gamma2), (C, delta2)):
        f = bn.FQ12.mul(f, bn.miller_loop(Q2, P1))
    lhs = bn.final_exponentiation(f)
    assert bn.FQ12.eq(lhs, _fq12_from_sanitized(s["negalfa1xbeta2"]))


def test_pairing_kat_negalfa1xbeta2_and_alphabeta12():
    """negalfa1xbeta2 == e(-alpha1, beta2) (plain final exponentiation) and
    vk_alphabeta_12 == e(alpha1, beta2)^(2x(6x^2+3x+1)) in the 2-3-2 tower layout."""
    vkey = json.load(open(os.path.join(REF, "layer_one_vkey.json")))
    s = json.loads(_load("4_sigs_2_batches_12_height__layer_one__batch_0", "sanitized_proof.json"))
    alpha1 = g16.g1_from_obj(vkey["vk_alpha_1"])
    beta2 = g16.g2_from_obj(vkey["vk_beta_2"])
    e_neg = bn.pairing(beta2, bn.ec_neg(alpha1, bn.FQ))
    assert bn.FQ12.eq(e_neg, _fq12_from_sanitized(s["negalfa1xbeta2"]))
    x = bn.BN_X
    e = bn.FQ12.pow(bn.pairing(beta2, alpha1), 2 * x * (6 * x * x + 3 * x + 1))
    ab = vkey["vk_alphabeta_12"]
    c = [0] * 12
    for k in range(2):
        for j in range(3):
            a, b = int(ab[k][j][0]), int(ab[k][j][1])
            c[2 * j + 6 + k] = b % bn.Q
            c[2 * j + k] = (a - 9 * b) % bn.Q
    assert bn.FQ12.eq(e, tuple(c))
