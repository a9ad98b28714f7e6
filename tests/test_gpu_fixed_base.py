"""Fixed-base form of the MSMs (csrc/msm.hip.h MsmTable; C ABI zkpoa_msm_table_*, zkpoa_zkey_precompute):
2^(c*j) * P_i precomputed for every window, all windows sharing one bucket set. It must be a pure re-arrangement:
every result bit-identical to the classic form and to the oracle -- G1 and G2, every sort-pass boundary of the
window width, infinity bases, special scalars, hot buckets, and whole proofs with tables on some or all sections."""
import json
import random

import numpy as np
import pytest

from conftest import golden_case, le
from oracle import c_oracle as co
from oracle.py import bn254 as bn
from oracle.py import groth16 as g16
from test_gpu_kernels import _dlog_expected, _dlog_setup, _np_scalars, _rand_scalars

pytestmark = pytest.mark.gpu
R = bn.R


def _dev(b):
    import torch
    return torch.frombuffer(bytearray(b), dtype=torch.uint8).cuda()


@pytest.mark.parametrize("group,n,dist,c", [
    (1, 1, "uniform", 0), (1, 33, "special", 0), (1, 3000, "witness", 0), (1, 3000, "uniform", 4),
    (1, 20000, "witness", 9), (1, 20000, "uniform", 10), (1, 40000, "witness", 17), (1, 40000, "special", 18),
    (1, 5000, "witness", 24), (2, 700, "uniform", 0), (2, 2000, "witness", 12), (2, 2000, "special", 17),
])
def test_table_msm_equals_oracle(ctx, group, n, dist, c):
    rng = random.Random(1000 * group + n + c)
    size = 64 if group == 1 else 128
    fb = co.fixed_base_g1 if group == 1 else co.fixed_base_g2
    bases = bytearray(fb(b"".join(le(rng.randrange(R)) for _ in range(n)), 8))
    for i in range(0, n, 17):                                   # infinity bases, as unused wires have in a zkey
        if n > 1:
            bases[size * i:size * i + size] = bytes(size)
    sc = _rand_scalars(rng, n, dist)
    d_b, d_s = _dev(bases), _dev(sc)
    table = ctx.msm_table(group, d_b.data_ptr(), n, c)
    try:
        tn, tc, tw, tbytes = table.info()
        assert tn == n and tw == (254 + tc - 1) // tc and tbytes == n * tw * size and (c == 0 or tc == c)
        want = (co.msm_g1 if group == 1 else co.msm_g2)(bytes(bases), sc, n, 8)
        assert ctx.msm_table_run(table, d_s.data_ptr()) == want
        assert ctx.msm_table_run(table, d_s.data_ptr(), lane=3) == want       # any lane, repeatable
        classic = (ctx.msm_g1_device if group == 1 else ctx.msm_g2_device)(d_b.data_ptr(), d_s.data_ptr(), n)
        assert classic == want
    finally:
        table.close()


def test_table_rows_are_the_shifted_bases(ctx):
    """Row j of a table is 2^(c*j) * P_i in the zkey wire format (checked against the big-int oracle)."""
    import torch
    n, c = 5, 13
    rng = random.Random(4)
    ks = [rng.randrange(R) for _ in range(n)]
    bases = bytearray(co.fixed_base_g1(b"".join(le(k) for k in ks), 1))
    bases[64:128] = bytes(64)
    ks[1] = 0
    d_b = _dev(bases)
    # the table itself is not exposed; a one-hot digit selects one row: scalar 2^(c*j) picks table[j][i]
    table = ctx.msm_table(1, d_b.data_ptr(), n, c)
    try:
        for j in (0, 1, 7, 19):
            for i in (0, 1, 4):
                sc = bytearray(32 * n)
                sc[32 * i:32 * i + 32] = le(1 << (c * j))
                got = g16.g1_from_bytes(ctx.msm_table_run(table, _dev(sc).data_ptr()))
                assert got == bn.g1_mul(bn.G1_GEN, ks[i] * (1 << (c * j)) % R)
    finally:
        table.close()


@pytest.mark.parametrize("dist", ["uniform", "witness"])
def test_table_msm_full_size_known_dlog(ctx, dist):
    """BASELINE.json configs[1] shape through the fixed-base form: 2^20 points, known discrete log."""
    import torch
    n = 1 << 20
    a, b, d_bases = _dlog_setup(ctx, n, 99)
    limbs = _np_scalars(n, 7, dist)
    d_sc = torch.from_numpy(limbs.view(np.uint8).reshape(-1).copy()).cuda()
    table = ctx.msm_table(1, d_bases.data_ptr(), n)
    try:
        out = ctx.msm_table_run(table, d_sc.data_ptr())
    finally:
        table.close()
    assert g16.g1_from_bytes(out) == bn.g1_mul(bn.G1_GEN, _dlog_expected(limbs, a, b))
    assert out == ctx.msm_g1_device(d_bases.data_ptr(), d_sc.data_ptr(), n)


@pytest.mark.parametrize("tag", ["n8", "n128"])
def test_prove_with_tables_is_bit_identical(ctx, zk, tag):
    g = golden_case(tag)
    rs = json.loads(g["rs.json"])
    key = ctx.load_zkey(g["circuit.zkey"])
    try:
        used = key.precompute()
        assert used > 0
        for _ in range(2):
            pts, pub = ctx.prove(key, g["witness.wtns"], int(rs["r"]), int(rs["s"]))
            assert zk.proof_to_json(pts) == g["proof_rapidsnark.json"]
            assert zk.public_to_json(pub) == g["public_rapidsnark.json"]
    finally:
        key.close()


def test_prove_with_partial_tables_and_shards(ctx, zk):
    """2^16 synthetic key: tables on every section, on a budget that only fits some, and with the handle re-pointed
    to shards (tables are bypassed there) -- always the same proof, and it matches the known-dlog expectation."""
    from zkpoa_amd.synthetic import SyntheticCircuit
    circ = SyntheticCircuit(zk, ctx, 16, 60000, n_public=2, seed=77, witness_like=True)
    try:
        rng = random.Random(8)
        r_, s_ = rng.randrange(R), rng.randrange(R)
        want, _ = circ.prove(r_, s_)
        P = circ.h_scalars()
        assert P.tobytes() == co.h_scalars(circ.coeff_section_bytes(), circ.witness_bytes(), circ.m, 16)
        assert circ.check(want, r_, s_, P)
        full = circ.key.precompute()
        assert full > 0
        assert circ.prove(r_, s_)[0] == want
        some = circ.key.precompute(full // 3)             # only the first table(s) fit
        assert 0 < some <= full // 3
        assert circ.prove(r_, s_)[0] == want
        assert circ.key.precompute(1) == 0                # nothing fits: classic form everywhere
        assert circ.prove(r_, s_)[0] == want
        circ.key.precompute()
        header = circ.key.header()
        parts = []
        for rank in range(3):
            circ.key.set_shard(rank, 3)
            parts.append(ctx.prove_partials_device(circ.key, circ.d_witness.data_ptr()))
        circ.key.set_shard(0, 1)
        assert zk.prove_assemble(header, zk.sum_partials(parts), r_, s_) == want
        assert circ.prove(r_, s_)[0] == want              # whole key again: tables back in use
    finally:
        circ.close()


@pytest.mark.parametrize("pattern", ["w0_only", "all_ones", "plus_minus_one", "limbs64", "mixed"])
def test_sparse_first_pass_extreme_witnesses(ctx, zk, pattern):
    """The witness MSMs through tables take a compact entry list instead of the dense digit array when the witness is
    sparse in digits (csrc/msm_sort.hip.h msm_entries_kernel). Witnesses at the edges of that path -- nothing but w[0],
    all ones (one hot bucket holds everything), alternating 1 / r - 1 (every digit negative or positive one), 64-bit
    limbs, and a mix with full-width values -- must give the proof the classic (dense, table-free) form gives, and that
    proof must match the known discrete logs with H scalars the C oracle confirms."""
    import torch
    from zkpoa_amd.synthetic import SyntheticCircuit
    k, m = 12, 3000
    circ = SyntheticCircuit(zk, ctx, k, m, n_public=2, seed=5, witness_like=True)
    try:
        w = np.zeros((m, 4), dtype=np.uint64)
        rl = [(R >> (64 * j)) & 0xFFFFFFFFFFFFFFFF for j in range(4)]
        if pattern == "all_ones":
            w[:, 0] = 1
        elif pattern == "plus_minus_one":
            w[:, 0] = 1
            w[1::2] = [rl[0] - 1, rl[1], rl[2], rl[3]]                 # r - 1
        elif pattern == "limbs64":
            w[:, 0] = np.random.default_rng(3).integers(1, 1 << 63, size=m, dtype=np.uint64) * 2 + 1
        elif pattern == "mixed":
            g = np.random.default_rng(4)
            w[:, 0] = g.integers(0, 2, size=m, dtype=np.uint64)
            full = g.integers(0, 1 << 62, size=(m // 50 + 1, 4), dtype=np.uint64)
            full[:, 3] &= np.uint64((1 << 60) - 1)                     # < 2^252 < r: canonical field elements
            w[::50] = full[: len(w[::50])]
        w[0] = (1, 0, 0, 0)
        circ.w_limbs = w
        circ.d_witness = torch.from_numpy(w.view(np.uint8).reshape(-1).copy()).cuda()
        r_, s_ = 11, 22
        want, _ = circ.prove(r_, s_)                                    # no tables: classic form, dense digits
        P = circ.h_scalars()
        assert P.tobytes() == co.h_scalars(circ.coeff_section_bytes(), circ.witness_bytes(), m, k)
        a, b, c = circ.expected_dlogs(r_, s_, P)
        assert g16.g1_from_bytes(want, 0) == bn.g1_mul(bn.G1_GEN, a)
        assert g16.g2_from_bytes(want, 64) == bn.g2_mul(bn.G2_GEN, b)
        assert g16.g1_from_bytes(want, 192) == bn.g1_mul(bn.G1_GEN, c)
        assert circ.key.precompute() > 0                                 # tables sized with this witness's digit density
        for _ in range(2):
            assert circ.prove(r_, s_)[0] == want
    finally:
        circ.close()
