"""Self-verifying prover: the reference verifies every proof right after proving it
(scripts/g16_verify.sh:213-216, scripts/full_workflow.sh:503-504). A .zkey carries its own verification key
(section 2: alpha1, beta2, gamma2, delta2; section 3: IC), so the prover checks its first proof per key against
it before writing anything -- a zkey whose conventions differ from SURVEY.md 8c fails on first contact.

CPU part: the wire-format verifier entry point (zkpoa_groth16_verify_points) is pinned on the REFERENCE's own
vkey / proof / public fixtures, converted to the zkey wire format here. GPU part: the prover's self-check passes
on golden keys and fails when section 9, one section-4 coefficient's R^2 scaling, or the witness is corrupted."""
import json
import os
import struct
import subprocess

import pytest

from conftest import GOLDEN, golden_case
from oracle.py import groth16 as g16

Q = 21888242871839275222246405745257275088696311157297823662689037894645226208583
R = 21888242871839275222246405745257275088548364400416034343698204186575808495617
REF = os.path.join(GOLDEN, "ref")
CASES = [
    ("4_sigs_2_batches_12_height__layer_one__batch_0", "layer_one_vkey.json"),
    ("4_sigs_2_batches_12_height__layer_two__batch_1", "layer_two_vkey.json"),
    ("4_sigs_2_batches_12_height__layer_three", "layer_three_vkey.json"),
]


def _mont(dec):
    return ((int(dec) << 256) % Q).to_bytes(32, "little")


def _g1(o):
    assert o[2] == "1"
    return _mont(o[0]) + _mont(o[1])


def _g2(o):
    assert o[2] == ["1", "0"]
    return _mont(o[0][0]) + _mont(o[0][1]) + _mont(o[1][0]) + _mont(o[1][1])


def _wire(vkey, proof, public):
    vk = _g1(vkey["vk_alpha_1"]) + _g2(vkey["vk_beta_2"]) + _g2(vkey["vk_gamma_2"]) + _g2(vkey["vk_delta_2"])
    vk += b"".join(_g1(p) for p in vkey["IC"])
    pts = _g1(proof["pi_a"]) + _g2(proof["pi_b"]) + _g1(proof["pi_c"])
    pub = b"".join(int(x).to_bytes(32, "little") for x in public)
    return vk, pts, pub


@pytest.mark.parametrize("d,vkf", CASES)
def test_verify_points_on_reference_fixtures(zk, d, vkf):
    rd = lambda *p: json.load(open(os.path.join(REF, *p)))
    vk, pts, pub = _wire(rd(vkf), rd(d, "proof.json"), rd(d, "public.json"))
    assert zk.groth16_verify_points(vk, pts, pub) is True
    bad = bytearray(pub)
    bad[0] ^= 1
    assert zk.groth16_verify_points(vk, pts, bytes(bad)) is False
    assert zk.groth16_verify_points(vk, pts[192:] + pts[64:192] + pts[:64], pub) is False   # pi_a <-> pi_c
    off = bytearray(pts)
    off[0] ^= 1                                                                             # off the curve
    assert zk.groth16_verify_points(vk, bytes(off), pub) is False
    assert zk.groth16_verify_points(vk, pts, R.to_bytes(32, "little") + pub[32:]) is False    # public input >= r
    with pytest.raises(zk.ZkpoaError):
        zk.groth16_verify_points(vk[:-64], pts, pub)                                        # IC length != nPublic + 1


@pytest.mark.parametrize("tag", ["n8", "n128"])
def test_zkey_sections_2_3_are_the_vkey(zk, tag):
    """The vkey bytes taken from sections 2-3 of a golden zkey verify that key's golden proof (host only)."""
    g = golden_case(tag)
    z = g["circuit.zkey"]
    secs = g16.read_binfile(z, "zkey", 2)
    p2, _ = secs[2][0]
    p3, l3 = secs[3][0]
    h = p2 + 4 + 32 + 4 + 32 + 12
    vk = z[h:h + 64] + z[h + 128:h + 256] + z[h + 256:h + 384] + z[h + 448:h + 576] + z[p3:p3 + l3]
    proof = json.loads(g["proof_rapidsnark.json"])
    pub = json.loads(g["public_rapidsnark.json"])
    pts = _g1(proof["pi_a"]) + _g2(proof["pi_b"]) + _g1(proof["pi_c"])
    assert zk.groth16_verify_points(vk, pts, b"".join(int(x).to_bytes(32, "little") for x in pub)) is True


# ---- verifier input hardening ------------------------------------------------------------------
def test_verifier_rejects_hostile_json(zk):
    d, vkf = CASES[0]
    rd = lambda *p: open(os.path.join(REF, *p)).read()
    vkey, public, proof = rd(vkf), rd(d, "public.json"), rd(d, "proof.json")
    vk = json.loads(vkey)
    for np in ("-1", "1e3", "", "99999999999999999999"):
        vk2 = dict(vk, nPublic=np)
        with pytest.raises(zk.ZkpoaError):
            zk.groth16_verify(json.dumps(vk2), public, proof)
    with pytest.raises(zk.ZkpoaError):
        zk.groth16_verify(json.dumps(dict(vk, IC=[], nPublic=0)), "[]", proof)
    deep = "[" * 100000 + "]" * 100000
    with pytest.raises(zk.ZkpoaError, match="deep"):
        zk.groth16_verify(vkey, deep, proof)
    # z not in {0, 1}: snarkjs (ffjavascript fromObject) reads Jacobian coordinates (x/z^2, y/z^3)
    pr = json.loads(proof)
    z = 7
    x, y = int(pr["pi_a"][0]), int(pr["pi_a"][1])
    pr["pi_a"] = [str(x * z * z % Q), str(y * z ** 3 % Q), str(z)]
    assert zk.groth16_verify(vkey, public, json.dumps(pr)) is True
    pr["pi_a"] = [str(x * z % Q), str(y * z % Q), str(z)]            # homogeneous projective: not what snarkjs reads
    assert zk.groth16_verify(vkey, public, json.dumps(pr)) is False


# ---- GPU: the prover checks its own first proof ---------------------------------------------------------------
def _patch(z, sid, off, data):
    secs = g16.read_binfile(z, "zkey", 2)
    p, _ = secs[sid][0]
    out = bytearray(z)
    out[p + off:p + off + len(data)] = data
    return bytes(out)


@pytest.mark.gpu
def test_selfcheck_passes_and_is_timed(ctx, zk, monkeypatch):
    monkeypatch.delenv("ZKPOA_SELFCHECK", raising=False)
    g = golden_case("n128")
    rs = json.loads(g["rs.json"])
    key = ctx.load_zkey(g["circuit.zkey"])
    try:
        assert key.vkey_points() is not None
        pts, pub = ctx.prove(key, g["witness.wtns"], int(rs["r"]), int(rs["s"]))
        assert zk.proof_to_json(pts) == g["proof_rapidsnark.json"]
        first = ctx.last_ms(6)
        assert 0 < first < 1000, "the first proof of a key must be self-checked"
        assert zk.groth16_verify_points(key.vkey_points(), pts, pub) is True
        ctx.prove(key, g["witness.wtns"])
        assert ctx.last_ms(6) == 0, "default mode checks only the first proof of a key"
        monkeypatch.setenv("ZKPOA_SELFCHECK", "all")
        ctx.prove(key, g["witness.wtns"])
        assert ctx.last_ms(6) > 0
        monkeypatch.setenv("ZKPOA_SELFCHECK", "0")
        ctx.prove(key, g["witness.wtns"])
        assert ctx.last_ms(6) == 0
        print("self-check cost: %.2f ms (host pairing check)" % first)
    finally:
        key.close()


@pytest.mark.gpu
@pytest.mark.parametrize("what", ["h_points_swapped", "coef_not_r2_scaled", "c_section_shifted", "witness_value"])
def test_selfcheck_catches_broken_conventions(ctx, zk, what, monkeypatch):
    monkeypatch.delenv("ZKPOA_SELFCHECK", raising=False)
    g = golden_case("n128")
    z, w = g["circuit.zkey"], g["witness.wtns"]
    secs = g16.read_binfile(z, "zkey", 2)
    if what == "h_points_swapped":            # section 9 in another order (e.g. a different coset / root of unity)
        p9, _ = secs[9][0]
        z = _patch(z, 9, 0, z[p9 + 64:p9 + 128] + z[p9:p9 + 64])
    elif what == "coef_not_r2_scaled":         # one coefficient stored as coef*R instead of coef*R^2
        p4, _ = secs[4][0]
        ncoef = struct.unpack_from("<I", z, p4)[0]
        _, wit = g16.read_wtns(w)
        for i in range(ncoef):                   # a coefficient of a wire whose value is non-zero
            o = p4 + 4 + 44 * i
            sig = struct.unpack_from("<I", z, o + 8)[0]
            if wit[sig] != 0:
                v = int.from_bytes(z[o + 12:o + 44], "little")
                v = v * pow(1 << 256, -1, R) % R
                z = _patch(z, 4, 4 + 44 * i + 12, v.to_bytes(32, "little"))
                break
    elif what == "c_section_shifted":          # C points rotated by one wire (wrong section-8 offset)
        p8, l8 = secs[8][0]
        z = _patch(z, 8, 0, z[p8 + 64:p8 + l8] + z[p8:p8 + 64])
    else:                                       # a witness that does not satisfy the circuit
        _, wit = g16.read_wtns(w)
        wit[3] = (wit[3] + 1) % R
        w = g16.write_wtns(wit)
    key = ctx.load_zkey(z)
    try:
        with pytest.raises(zk.ZkpoaError, match="self-check failed"):
            ctx.prove(key, w, 5, 6)
        monkeypatch.setenv("ZKPOA_SELFCHECK", "0")   # escape hatch: the (invalid) proof is returned as before
        pts, _ = ctx.prove(key, w, 5, 6)
        assert len(pts) == 256
    finally:
        key.close()


@pytest.mark.gpu
def test_cli_selfcheck_exit_code_and_no_partial_output(zk, tmp_path):
    """Under the reference's `set -eE` workflow a failed self-check must look like any prover failure:
    non-zero exit, message on stderr, no proof.json left behind (scripts/lib/error_handling.sh:14-41)."""
    g = golden_case("n128")
    z = g["circuit.zkey"]
    secs = g16.read_binfile(z, "zkey", 2)
    p9, _ = secs[9][0]
    (tmp_path / "c_final.zkey").write_bytes(_patch(z, 9, 0, z[p9 + 64:p9 + 128] + z[p9:p9 + 64]))
    (tmp_path / "witness.wtns").write_bytes(g["witness.wtns"])
    env = {k: v for k, v in os.environ.items() if k != "ZKPOA_SELFCHECK"}
    rc = subprocess.run([zk.PROVER_BIN, str(tmp_path / "c_final.zkey"), str(tmp_path / "witness.wtns"),
                         str(tmp_path / "proof.json"), str(tmp_path / "public.json")], env=env,
                        capture_output=True, text=True)
    assert rc.returncode != 0 and "self-check failed" in rc.stderr
    assert not (tmp_path / "proof.json").exists() and not (tmp_path / "public.json").exists()


# ---- `snarkjs zkey export verificationkey` from zkey sections 1-3, pinned on the reference's *_vkey.json -------------
def _zkey_with_vkey(vkey):
    """A zkey image holding just what the export reads: section 1 (protocol), 2 (header), 3 (IC), built from a vkey."""
    R_ = R
    hdr = struct.pack("<I", 32) + Q.to_bytes(32, "little") + struct.pack("<I", 32) + R_.to_bytes(32, "little")
    hdr += struct.pack("<III", 1000, int(vkey["nPublic"]), 1024)
    dummy_g1 = _g1(vkey["vk_alpha_1"])
    hdr += _g1(vkey["vk_alpha_1"]) + dummy_g1 + _g2(vkey["vk_beta_2"]) + _g2(vkey["vk_gamma_2"]) + dummy_g1 + _g2(vkey["vk_delta_2"])
    ic = b"".join(_g1(p) for p in vkey["IC"])
    return g16.write_binfile("zkey", 1, [(1, struct.pack("<I", 1)), (2, hdr), (3, ic)])


@pytest.mark.parametrize("vkf", ["layer_one_vkey.json", "layer_two_vkey.json", "layer_three_vkey.json"])
def test_export_vkey_reproduces_reference_files(zk, vkf, tmp_path):
    """Byte for byte the reference's committed <circuit>_vkey.json (snarkjs output), vk_alphabeta_12 included."""
    text = open(os.path.join(REF, vkf)).read()
    z = _zkey_with_vkey(json.loads(text))
    assert zk.export_vkey(z) == text
    (tmp_path / "c.zkey").write_bytes(z)
    rc = subprocess.run([zk.VERIFY_BIN, "--export-vkey", str(tmp_path / "c.zkey"), str(tmp_path / "vkey.json")],
                        capture_output=True, text=True)
    assert rc.returncode == 0, rc.stderr
    assert (tmp_path / "vkey.json").read_text() == text
    with pytest.raises(zk.ZkpoaError):
        zk.export_vkey(z[:40])


@pytest.mark.parametrize("tag", ["n8", "n128"])
def test_export_vkey_of_golden_zkeys(zk, tag):
    g = golden_case(tag)
    got = json.loads(zk.export_vkey(g["circuit.zkey"]))
    want = json.loads(g["vkey.json"])
    assert {k: v for k, v in got.items() if k != "vk_alphabeta_12"} == want
    # and the exported key verifies the golden proof
    assert zk.groth16_verify(json.dumps(got), g["public_rapidsnark.json"], g["proof_rapidsnark.json"])
