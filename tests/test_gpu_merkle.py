"""GPU parity for the anonymity-set Poseidon Merkle tree (reference: scripts/merkle_tree.rs, exec'd at
scripts/full_workflow.sh:371-380): hashes and trees bit-exact against the oracle, the reference's committed anonymity
set must give the Merkle root the reference logged, and the drop-in `merkle-tree` executable must write the files
merkle_tree.rs writes."""
import csv
import json
import os
import random
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, le
from oracle import c_oracle as co
from oracle.py import poseidon as P
from test_poseidon_oracle import REF_ROOT, anon_set

pytestmark = pytest.mark.gpu
R = P.R


def _ints(b):
    return [int.from_bytes(b[32 * i:32 * i + 32], "little") for i in range(len(b) // 32)]


def test_poseidon2_vs_oracles(ctx):
    rng = random.Random(11)
    xs = [0, 1, R - 1, 1, 2 ** 160 - 1] + [rng.randrange(R) for _ in range(3000)]
    ys = [0, 2, R - 1, R - 1, 7] + [rng.randrange(R) for _ in range(3000)]
    l, r = b"".join(le(x) for x in xs), b"".join(le(y) for y in ys)
    got = ctx.poseidon2(l, r)
    assert got == co.poseidon2(l, r, 8)
    assert _ints(got)[:40] == [P.poseidon([x, y]) for x, y in zip(xs[:40], ys[:40])]
    assert _ints(got)[1] == 7853200120776062878684798364095072458815029376092732009249414926327459813530   # circomlib
    assert ctx.poseidon2(b"", b"") == b""


def test_reference_anonymity_set_root(ctx):
    """The reference's own fixture: tests/1_sigs_1_batches_5_height/anonymity_set_10.csv -> logs/merkle_tree.log:13."""
    addr, bal = anon_set()
    tree = ctx.merkle_build(b"".join(le(a) for a in addr), b"".join(le(b) for b in bal))
    try:
        assert tree.info() == (10, 4, 31)
        assert tree.root() == REF_ROOT
        levels = P.merkle_levels(addr, bal)
        assert _ints(tree.leaves()) == levels[0]                      # incl. the six zero-valued padding leaves
        for i in range(16):
            elems, bits = tree.path(i)
            assert (elems, bits) == P.merkle_path(levels, i)
            node = levels[0][i]
            for e, b in zip(elems, bits):
                node = P.poseidon([e, node] if b else [node, e])
            assert node == REF_ROOT
    finally:
        tree.close()


def test_reference_height12_paths_on_the_device(ctx):
    """The four height-12 sibling paths of tests/4_sigs_2_batches_12_height (layer_two_batch_{0,1}_input.json) folded with
    the DEVICE hash: every level's four hashes in one zkpoa_poseidon2 call, ending at the committed merkle_root."""
    from test_poseidon_oracle import height12_paths
    paths = height12_paths()
    nodes = _ints(ctx.poseidon2(b"".join(le(a) for a, _, _, _, _ in paths), b"".join(le(b) for _, b, _, _, _ in paths)))
    assert nodes == [P.poseidon([a, b]) for a, b, _, _, _ in paths]
    for lvl in range(11):
        left = [p[3][lvl] if p[4][lvl] else nd for p, nd in zip(paths, nodes)]
        right = [nd if p[4][lvl] else p[3][lvl] for p, nd in zip(paths, nodes)]
        nodes = _ints(ctx.poseidon2(b"".join(le(x) for x in left), b"".join(le(x) for x in right)))
    assert nodes == [p[2] for p in paths]


def test_reference_merkle_path_fixture(ctx):
    """The sibling path the reference's Rust binary produced for leaf 3 of that set and fed to its layer-two circuit
    (tests/1_sigs_1_batches_5_height/layer_two/batch_0/layer_two_batch_0_input.json path_elements / path_indices)."""
    from test_poseidon_oracle import ref_path_fixture
    leaf_addr, leaf_bal, root, elems, bits = ref_path_fixture()
    addr, bal = anon_set()
    idx = addr.index(leaf_addr)
    assert bal[idx] == leaf_bal
    tree = ctx.merkle_build(b"".join(le(a) for a in addr), b"".join(le(b) for b in bal))
    try:
        assert tree.root() == root
        assert tree.path(idx) == (elems, bits)
        assert _ints(tree.leaves())[idx] == P.poseidon([leaf_addr, leaf_bal])
    finally:
        tree.close()


@pytest.mark.parametrize("n", [1, 2, 3, 17, 1000, 4097])
def test_tree_vs_c_oracle(ctx, n):
    rng = random.Random(n)
    addr = b"".join(le(rng.randrange(1 << 160)) for _ in range(n))
    bal = b"".join(le(rng.randrange(1 << 90)) for _ in range(n))
    k = max(0, (n - 1).bit_length())
    want = co.merkle_levels(addr, bal, k, 8)
    tree = ctx.merkle_build(addr, bal)
    try:
        assert tree.info() == (n, k, (2 << k) - 1)
        assert tree.root() == int.from_bytes(want[-32:], "little")
        assert tree.leaves() == want[:32 << k]
        for i in {0, n - 1, (1 << k) - 1, rng.randrange(1 << k)}:
            elems, bits = tree.path(i)
            off, j = 0, i
            for l in range(k):                                        # sibling at every level of the oracle's array
                assert elems[l] == int.from_bytes(want[32 * (off + (j ^ 1)):32 * (off + (j ^ 1)) + 32], "little")
                assert bits[l] == (j & 1)
                off += 1 << (k - l)
                j >>= 1
    finally:
        tree.close()


def test_tree_2p20_device_inputs(ctx):
    """2^20 leaves generated in HBM (no padding), root against the threaded C oracle."""
    import torch
    n = 1 << 20
    nr = np.random.default_rng(5)
    a = np.zeros((n, 4), dtype=np.uint64)
    b = np.zeros((n, 4), dtype=np.uint64)
    a[:, 0] = nr.integers(0, 1 << 63, size=n, dtype=np.uint64)
    a[:, 1] = nr.integers(0, 1 << 63, size=n, dtype=np.uint64)
    a[:, 2] = nr.integers(0, 1 << 32, size=n, dtype=np.uint64)       # 160-bit addresses
    b[:, 0] = nr.integers(0, 1 << 63, size=n, dtype=np.uint64)
    b[:, 1] = nr.integers(0, 1 << 26, size=n, dtype=np.uint64)       # 90-bit balances
    da, db = torch.from_numpy(a.view(np.uint8).reshape(-1)).cuda(), torch.from_numpy(b.view(np.uint8).reshape(-1)).cuda()
    tree = ctx.merkle_build(da.data_ptr(), db.data_ptr(), device=True, n=n)
    try:
        want = co.merkle_levels(a.tobytes(), b.tobytes(), 20, min(16, os.cpu_count() or 1))
        assert tree.root() == int.from_bytes(want[-32:], "little")
        assert tree.leaves(12345, 100) == want[32 * 12345:32 * 12445]
        print("2^20-leaf tree (2^21 - 1 hashes): %.2f ms on the GPU" % ctx.last_ms(7))
    finally:
        tree.close()


def test_merkle_tree_cli_drop_in(zk, tmp_path):
    """`merkle-tree --anon-set --poa-input-data --output-dir` (scripts/full_workflow.sh:375-379): merkle_root.json /
    merkle_proofs.json in merkle_tree.rs's shapes for the reference's anonymity set and three owned addresses."""
    addr, bal = anon_set()
    owned = [1, 3, 9]                                                 # in set order, as merkle_tree.rs requires
    poa = {"accountAttestations": [
        {"signature": {"r": {"__bigint__": "1"}, "s": {"__bigint__": "2"}},
         "accountData": {"address": {"__bigint__": str(addr[i])}, "balance": {"__bigint__": str(bal[i])}}} for i in owned]}
    (tmp_path / "parsed_sigs.json").write_text(json.dumps(poa, indent=2))
    out = tmp_path / "build"
    out.mkdir()
    argv = [zk.MERKLE_BIN, "--anon-set", os.path.join(GOLDEN, "ref", "merkle", "anonymity_set_10.csv"),
            "--poa-input-data", str(tmp_path / "parsed_sigs.json"), "--output-dir", str(out)]
    rc = subprocess.run(argv, capture_output=True, text=True, timeout=120)
    assert rc.returncode == 0, rc.stderr
    assert "Done creating Merkle tree of height 5" in rc.stdout        # the reference's log line for this set
    assert "Root hash %d written to file" % REF_ROOT in rc.stdout
    assert (out / "merkle_root.json").read_text() == '{\n  "__bigint__": "%d"\n}' % REF_ROOT   # serde_json pretty
    proofs = json.loads((out / "merkle_proofs.json").read_text())
    levels = P.merkle_levels(addr, bal)
    assert [int(l["address"]["__bigint__"]) for l in proofs["leaves"]] == [addr[i] for i in owned]
    assert [int(l["hash"]["__bigint__"]) for l in proofs["leaves"]] == [levels[0][i] for i in owned]
    for k, i in enumerate(owned):
        elems, bits = P.merkle_path(levels, i)
        assert [int(e["__bigint__"]) for e in proofs["path_elements"][k]] == elems
        assert proofs["path_indices"][k] == bits
    # the same text a serde_json pretty printer produces (2-space indent, no trailing newline)
    assert (out / "merkle_proofs.json").read_text() == json.dumps(proofs, indent=2)
    # an owned address that is not in the set (or out of order): the Rust binary panics, this one exits non-zero
    poa["accountAttestations"].reverse()
    (tmp_path / "parsed_sigs.json").write_text(json.dumps(poa))
    rc = subprocess.run(argv, capture_output=True, text=True, timeout=120)
    assert rc.returncode != 0 and "does not exist in the anonymity set" in rc.stderr
    # a balance that is not a field element
    bad = tmp_path / "bad.csv"
    bad.write_text("address,eth_balance\n0x01,%d\n" % R)
    rc = subprocess.run([zk.MERKLE_BIN, "-a", str(bad), "-p", str(tmp_path / "parsed_sigs.json"), "-o", str(out)],
                        capture_output=True, text=True, timeout=120)
    assert rc.returncode != 0 and "not below the BN254 scalar modulus" in rc.stderr


@pytest.mark.parametrize("threads", ["1", "7"])
def test_merkle_tree_cli_reads_a_messy_csv_in_parallel(zk, tmp_path, threads):
    """The CLI maps the anonymity set and parses it in pieces cut at line starts, one per thread (a 10 M-line set is ~650 MB
    of text). 5000 rows with everything the line-at-a-time reader accepted -- blank lines, CRLF endings, quoted fields,
    padding blanks, a last line without a newline -- read on 1 and on 7 threads must give the root the C oracle computes
    from the same rows; a malformed row in the middle is reported with its text."""
    rng = random.Random(99)
    rows = [(rng.randrange(1 << 160), rng.randrange(1 << 90)) for _ in range(5000)]
    lines = ["address,eth_balance"]
    for i, (a, b) in enumerate(rows):
        s = ("0x%040x" % a, str(b))
        form = i % 5
        line = {0: "%s,%s", 1: '"%s","%s"', 2: "  %s , %s  ", 3: "%s,%s\r", 4: "0X%s,%s"}[form] % (
            (s[0][2:], s[1]) if form == 4 else s)
        lines.append(line)
        if i % 97 == 0:
            lines.append("")                                           # blank lines are skipped
    text = "\n\n" + "\n".join(lines)                                   # leading blank lines, no final newline
    (tmp_path / "set.csv").write_text(text, newline="")
    own = [17, 2048, 4999]
    poa = {"accountAttestations": [{"accountData": {"address": {"__bigint__": str(rows[i][0])},
                                                       "balance": {"__bigint__": str(rows[i][1])}}} for i in own]}
    (tmp_path / "poa.json").write_text(json.dumps(poa))
    out = tmp_path / "out"
    out.mkdir()
    env = dict(os.environ, ZKPOA_MERKLE_THREADS=threads)
    argv = [zk.MERKLE_BIN, "-a", str(tmp_path / "set.csv"), "-p", str(tmp_path / "poa.json"), "-o", str(out)]
    rc = subprocess.run(argv, env=env, capture_output=True, text=True, timeout=120)
    assert rc.returncode == 0, rc.stderr
    assert "Done creating 5000 leaves" in rc.stdout
    want = co.merkle_levels(b"".join(le(a) for a, _ in rows), b"".join(le(b) for _, b in rows), 13, 8)
    root = int.from_bytes(want[-32:], "little")
    assert (out / "merkle_root.json").read_text() == '{\n  "__bigint__": "%d"\n}' % root
    proofs = json.loads((out / "merkle_proofs.json").read_text())
    assert [int(l["address"]["__bigint__"]) for l in proofs["leaves"]] == [rows[i][0] for i in own]
    # fold every path with the oracle's hash
    for k, i in enumerate(own):
        node = int(proofs["leaves"][k]["hash"]["__bigint__"])
        assert node == P.poseidon([rows[i][0], rows[i][1]])
        for e, bit in zip(proofs["path_elements"][k], proofs["path_indices"][k]):
            e = int(e["__bigint__"])
            node = P.poseidon([e, node] if bit else [node, e])
        assert node == root
    bad = text.replace("0x%040x" % rows[3000][0], "0x%039xg" % (rows[3000][0] >> 4), 1)
    (tmp_path / "set.csv").write_text(bad, newline="")
    rc = subprocess.run(argv, env=env, capture_output=True, text=True, timeout=120)
    assert rc.returncode == 1 and "malformed address" in rc.stderr and ("%039xg" % (rows[3000][0] >> 4)) in rc.stderr
