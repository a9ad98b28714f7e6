"""Oracle for the Poseidon Merkle tree of the anonymity set (scripts/merkle_tree.rs), pinned on the
reference's own fixture: the committed anonymity set of tests/1_sigs_1_batches_5_height must hash to the Merkle root the
reference logged (logs/merkle_tree.log:13) and proved against (layer_two/batch_0/public.json, second public signal)."""
import csv
import json
import os
import random

import pytest

from conftest import GOLDEN, le
from oracle import c_oracle as co
from oracle.py import poseidon as P

REF_ROOT = 4980353021834912512710796692386145127886467347162150588171360986794629731619


def anon_set():
    rows = list(csv.reader(open(os.path.join(GOLDEN, "ref", "merkle", "anonymity_set_10.csv"))))[1:]
    return [int(a[2:], 16) for a, _ in rows], [int(b) for _, b in rows]


def test_parameters_match_circomlib():
    C, M = P.params(3)
    assert len(C) == 195 and C[0] == 0x0ee9a592ba9a9518d05986d656f40c2114c4993c11bb29938d21d47304cd8e6e
    # circomlib test vector (test/poseidoncircuit.js): poseidon([1, 2])
    assert P.poseidon([1, 2]) == 7853200120776062878684798364095072458815029376092732009249414926327459813530


def test_reference_merkle_root_python_oracle():
    addr, bal = anon_set()
    levels = P.merkle_levels(addr, bal)
    assert len(levels[0]) == 16 and levels[-1][0] == REF_ROOT
    # the root is also what the reference's layer-two proof exposes as its second public signal
    pub = json.load(open(os.path.join(GOLDEN, "ref", "1_sigs_1_batches_5_height__layer_two__batch_0", "public.json")))
    assert int(pub[1]) == REF_ROOT
    elems, idx = P.merkle_path(levels, 3)
    node = levels[0][3]
    for e, bit in zip(elems, idx):
        node = P.poseidon([e, node] if bit else [node, e])
    assert node == REF_ROOT


def ref_path_fixture():
    """The Merkle inputs of the reference's layer-two circuit for this set (tests/1_sigs_1_batches_5_height/layer_two/
    batch_0/layer_two_batch_0_input.json: leaf_addresses, leaf_balances, merkle_root, path_elements, path_indices)."""
    d = json.load(open(os.path.join(GOLDEN, "ref", "merkle", "layer_two_batch_0_merkle_inputs.json")))
    return (int(d["leaf_addresses"][0]), int(d["leaf_balances"][0]), int(d["merkle_root"]),
            [int(x) for x in d["path_elements"][0]], [int(x) for x in d["path_indices"][0]])


def test_reference_merkle_path_fixture_oracles():
    """The sibling path the reference fed to its layer-two circuit (produced by its Rust binary, merkle_tree.rs:354-376)
    equals the Python oracle's and the C oracle's path for that leaf, and folds to the logged root."""
    leaf_addr, leaf_bal, root, elems, bits = ref_path_fixture()
    addr, bal = anon_set()
    assert root == REF_ROOT
    idx = addr.index(leaf_addr)
    assert idx == 3 and bal[idx] == leaf_bal
    levels = P.merkle_levels(addr, bal)
    assert P.merkle_path(levels, idx) == (elems, bits)
    lv = co.merkle_levels(b"".join(le(a) for a in addr), b"".join(le(b) for b in bal), 4, 2)
    node_at = lambda level, i: int.from_bytes(lv[32 * ((32 - (32 >> level)) + i):][:32], "little")   # levels 16, 8, 4, 2, 1
    i, got_e, got_b = idx, [], []
    for level in range(4):
        got_e.append(node_at(level, i ^ 1))
        got_b.append(i & 1)
        i >>= 1
    assert (got_e, got_b) == (elems, bits)
    node = P.poseidon([leaf_addr, leaf_bal])
    for e, bit in zip(elems, bits):
        node = P.poseidon([e, node] if bit else [node, e])
    assert node == root


def height12_paths():
    """The four sibling paths of the reference's height-12 test (tests/4_sigs_2_batches_12_height/layer_two/batch_{0,1}/
    layer_two_batch_*_input.json): (leaf address, leaf balance, root, path elements, path indices) per owned leaf."""
    out = []
    for b in (0, 1):
        d = json.load(open(os.path.join(GOLDEN, "ref", "merkle", "height12_layer_two_batch_%d_merkle_inputs.json" % b)))
        for i in range(len(d["leaf_addresses"])):
            out.append((int(d["leaf_addresses"][i]), int(d["leaf_balances"][i]), int(d["merkle_root"]),
                        [int(x) for x in d["path_elements"][i]], [int(x) for x in d["path_indices"][i]]))
    return out


def test_reference_height12_paths_fold_to_the_root_in_both_oracles():
    """No anonymity set is committed for the height-12 test, but its four sibling paths are: every one must fold, leaf
    hash upwards, to the committed merkle_root -- which is also public[1] of both layer-two proofs of that test -- under
    the Python oracle's Poseidon and under the C oracle's."""
    paths = height12_paths()
    assert len(paths) == 4 and all(len(e) == 11 and len(b) == 11 for _, _, _, e, b in paths)
    for b in (0, 1):
        pub = json.load(open(os.path.join(GOLDEN, "ref", "4_sigs_2_batches_12_height__layer_two__batch_%d" % b, "public.json")))
        assert int(pub[1]) == paths[2 * b][2] == paths[2 * b + 1][2]
    for addr, bal, root, elems, bits in paths:
        node = P.poseidon([addr, bal])
        c_node = int.from_bytes(co.poseidon2(le(addr), le(bal), 1), "little")
        assert c_node == node
        for e, bit in zip(elems, bits):
            left, right = (e, node) if bit else (node, e)
            node = P.poseidon([left, right])
            c_node = int.from_bytes(co.poseidon2(le(left), le(right), 1), "little")
            assert c_node == node
        assert node == root


def test_c_oracle_equals_python_oracle():
    rng = random.Random(7)
    xs = [0, 1, P.R - 1, 2 ** 160 - 1] + [rng.randrange(P.R) for _ in range(12)]
    ys = [0, 2, P.R - 1, 5] + [rng.randrange(P.R) for _ in range(12)]
    got = co.poseidon2(b"".join(le(x) for x in xs), b"".join(le(y) for y in ys), 3)
    assert [int.from_bytes(got[32 * i:32 * i + 32], "little") for i in range(len(xs))] == [P.poseidon([x, y]) for x, y in zip(xs, ys)]
    addr, bal = anon_set()
    lv = co.merkle_levels(b"".join(le(a) for a in addr), b"".join(le(b) for b in bal), 4, 4)
    flat = [v for level in P.merkle_levels(addr, bal) for v in level]
    assert [int.from_bytes(lv[32 * i:32 * i + 32], "little") for i in range(31)] == flat
    assert int.from_bytes(lv[-32:], "little") == REF_ROOT


def test_library_parameters_equal_the_oracle(zk):
    """The product generates the Poseidon parameters itself (csrc/poseidon.hip, host side, Grain LFSR): all 195 round
    constants and the MDS matrix must equal the oracle's, i.e. circomlib's. Host only: no GPU needed."""
    C, M = zk.poseidon_params()
    oc, om = P.params(3)
    assert C == oc and M == om


def test_merkle_cli_fails_loudly_without_gpu(zk, tmp_path):
    import subprocess
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the gpu-marked tests")
    (tmp_path / "poa.json").write_text('{"accountAttestations": []}')
    rc = subprocess.run([zk.MERKLE_BIN, "--anon-set", os.path.join(GOLDEN, "ref", "merkle", "anonymity_set_10.csv"),
                         "--poa-input-data", str(tmp_path / "poa.json"), "--output-dir", str(tmp_path)],
                        capture_output=True, text=True, timeout=120)
    assert rc.returncode != 0 and "no HIP device" in rc.stderr
    assert not (tmp_path / "merkle_root.json").exists()
    rc = subprocess.run([zk.MERKLE_BIN], capture_output=True, text=True)
    assert rc.returncode == 2 and "--anon-set" in rc.stderr
