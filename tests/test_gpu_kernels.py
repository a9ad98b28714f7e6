"""GPU parity tests for the kernels of the path, through the C ABI (libzkpoa_prover.so), against
the oracle (golden vectors from the Python big-int oracle; the C restatement at sizes it finishes in
seconds; size-independent properties at BASELINE.json's full size 2^20).
Bar: bit-exact (integer arithmetic)."""
import random

import numpy as np
import pytest

from conftest import le, rd
from oracle import c_oracle as co
from oracle.py import bn254 as bn
from oracle.py import groth16 as g16

pytestmark = pytest.mark.gpu
Q, R, M = bn.Q, bn.R, bn.MONT_R


def _cat(hexes):
    return b"".join(bytes.fromhex(h) for h in hexes)


# ---- field layer ---------------------------------------------------------------
@pytest.mark.parametrize("name,field", [("fq", 0), ("fr", 1)])
def test_field_golden(ctx, vectors, name, field):
    v = vectors[name]
    a, b = _cat(v["a"]), _cat(v["b"])
    assert ctx.field_op(field, 0, a, b) == _cat(v["mont_mul"])
    assert ctx.field_op(field, 1, a, b) == _cat(v["add"])
    assert ctx.field_op(field, 2, a, b) == _cat(v["sub"])
    assert ctx.field_op(field, 3, a) == _cat(v["mont_inv"])
    assert ctx.field_op(field, 4, a) == _cat(v["to_mont"])
    assert ctx.field_op(field, 5, a) == _cat(v["from_mont"])


@pytest.mark.parametrize("field,p", [(0, Q), (1, R)])
def test_field_random_vs_c_oracle(ctx, field, p):
    rng = np.random.default_rng(field + 5)
    n = 50000
    limbs = rng.integers(0, 1 << 63, size=(2, n, 4), dtype=np.uint64) * 2 + 1
    limbs[:, :, 3] &= np.uint64((1 << 60) - 1)
    a, b = limbs[0].tobytes(), limbs[1].tobytes()
    for op in (0, 1, 2):
        assert ctx.field_op(field, op, a, b) == co.field_op(field, op, a, b)
    assert ctx.field_op(field, 0, b"", b"") == b""          # empty input


# ---- group layer -----------------------------------------------------------------------------------
def test_group_add_golden(ctx, vectors):
    for key, grp in (("g1_add", 1), ("g2_add", 2)):
        v = vectors[key]
        assert ctx.group_add(grp, _cat(v["a"]), _cat(v["b"])) == _cat(v["sum"])


# ---- MSM ---------------------------------------------------------------------
def test_msm_golden(ctx, vectors):
    for m in vectors["msm"]:
        fn = ctx.msm_g1 if m["group"] == 1 else ctx.msm_g2
        assert fn(bytes.fromhex(m["bases"]), bytes.fromhex(m["scalars"]), m["n"]).hex() == m["result"]


def test_msm_empty(ctx):
    assert ctx.msm_g1(b"", b"", 0) == bytes(64)
    assert ctx.msm_g2(b"", b"", 0) == bytes(128)


def _rand_scalars(rng, n, dist):
    special = [0, 1, 2, R - 1, R - 2, (R - 1) // 2, (R + 1) // 2, 1 << 16, (1 << 16) - 1, 1 << 15]
    out = []
    for _ in range(n):
        u = rng.random()
        if dist == "witness":
            out.append(rng.randrange(2) if u < 0.55 else rng.randrange(1 << 64) if u < 0.9 else rng.randrange(R))
        elif dist == "special":
            out.append(rng.choice(special))
        else:
            out.append(rng.randrange(R))
    return b"".join(le(k) for k in out)


@pytest.mark.parametrize("n,dist", [(3000, "uniform"), (3000, "witness"), (500, "special"), (20000, "witness")])
def test_msm_g1_vs_c_oracle(ctx, n, dist):
    rng = random.Random(n + len(dist))
    bases = bytearray(co.fixed_base_g1(b"".join(le(rng.randrange(R)) for _ in range(n)), 8))
    for i in range(0, n, 17):                      # infinity bases, as unused wires have in a zkey
        bases[64 * i:64 * i + 64] = bytes(64)
    sc = _rand_scalars(rng, n, dist)
    assert ctx.msm_g1(bytes(bases), sc, n) == co.msm_g1(bytes(bases), sc, n, 8)


@pytest.mark.parametrize("n,dist", [(700, "uniform"), (2000, "witness")])
def test_msm_g2_vs_c_oracle(ctx, n, dist):
    rng = random.Random(n)
    bases = bytearray(co.fixed_base_g2(b"".join(le(rng.randrange(R)) for _ in range(n)), 8))
    for i in range(0, n, 13):
        bases[128 * i:128 * i + 128] = bytes(128)
    sc = _rand_scalars(rng, n, dist)
    assert ctx.msm_g2(bytes(bases), sc, n) == co.msm_g2(bytes(bases), sc, n, 8)


def test_msm_repeated_and_opposite_bases(ctx):
    """All bases equal (bucket accumulation hits the doubling case) and P / -P pairs (hits P + (-P))."""
    n = 512
    G = g16.g1_to_bytes(bn.G1_GEN)
    negG = g16.g1_to_bytes(bn.ec_neg(bn.G1_GEN, bn.FQ))
    sc = le(5) * n
    assert g16.g1_from_bytes(ctx.msm_g1(G * n, sc, n)) == bn.g1_mul(bn.G1_GEN, 5 * n)
    bases = (G + negG) * (n // 2)
    assert ctx.msm_g1(bases, sc, n) == bytes(64)
    sc2 = b"".join(le(7 if i % 2 == 0 else 3) for i in range(n))
    assert g16.g1_from_bytes(ctx.msm_g1(bases, sc2, n)) == bn.g1_mul(bn.G1_GEN, 4 * (n // 2))


# window widths on both sides of every sort-pass boundary (c - 1 key bits: 1 pass up to 8, 2 up to 16, 3 up to 21)
# and of the odd / even splits of the bucket matrix in the reduction
@pytest.mark.parametrize("c", [4, 5, 8, 9, 10, 11, 14, 16, 17, 18, 20, 22])
def test_msm_forced_windows(ctx, c):
    rng = random.Random(c)
    n = 4096 if c < 16 else 40000
    bases = co.fixed_base_g1(b"".join(le(rng.randrange(R)) for _ in range(n)), 8)
    sc = _rand_scalars(rng, n, "witness")
    ctx.set_option("msm_c", c)
    try:
        got = ctx.msm_g1(bases, sc, n)
    finally:
        ctx.set_option("msm_c", 0)
    assert got == co.msm_g1(bases, sc, n, 8)


def test_scan_look_back_gives_up_with_an_error_not_a_hang(zk):
    """A tile prefix that is never published (r03: a host-side ordering mistake wiped live status words and the look-back
    span for ever) must come back as an error: with tile 0 withholding its prefix (test option) and the poll limit
    lowered, the MSM fails with the scan's message within seconds, and the same context computes correctly afterwards."""
    rng = random.Random(11)
    n = 40000
    bases = co.fixed_base_g1(b"".join(le(rng.randrange(R)) for _ in range(n)), 8)
    sc = _rand_scalars(rng, n, "witness")
    c = zk.Context(0)
    try:
        c.set_option("msm_c", 14)                  # 19 windows x 8192 buckets: 76 scan tiles
        c.set_option("scan_poll_limit_log2", 12)
        c.set_option("scan_test_withhold", 1)
        with pytest.raises(zk.ZkpoaError, match="scan look-back gave up"):
            c.msm_g1(bases, sc, n)
        c.set_option("scan_test_withhold", 0)
        c.set_option("scan_poll_limit_log2", 24)
        assert c.msm_g1(bases, sc, n) == co.msm_g1(bases, sc, n, 8)
    finally:
        c.close()


def test_msm_chunked(ctx):
    """An MSM over more points than one bucket sort may index (n x windows < 2^32: 2^27 points by default) runs
    in chunks added on the host; "msm_max_points" forces that path at a size the oracle can check."""
    rng = random.Random(31)
    n = 3001
    b1 = co.fixed_base_g1(b"".join(le(rng.randrange(R)) for _ in range(n)), 8)
    b2 = co.fixed_base_g2(b"".join(le(rng.randrange(R)) for _ in range(700)), 8)
    sc = _rand_scalars(rng, n, "witness")
    ctx.set_option("msm_max_points", 1000)           # 3001 = 3 full chunks + 1 point
    try:
        got1 = ctx.msm_g1(b1, sc, n)
        got2 = ctx.msm_g2(b2, sc[:700 * 32], 700)
    finally:
        ctx.set_option("msm_max_points", 0)
    assert got1 == co.msm_g1(b1, sc, n, 8)
    assert got2 == co.msm_g2(b2, sc[:700 * 32], 700, 8)


def test_msm_under_memory_pressure_goes_in_pieces(zk):
    """A lane's workspace grows with the points sorted at once (~0.6-0.9 KB per point). When it does not fit -- here:
    over a cap set with lane_workspace_max_mb, on a 2^27 key: HBM itself -- the MSM is not failed: it halves the piece
    it sorts until the workspace fits, the context remembers that size, and the sum is the same point (known discrete
    log, 3 x 2^20 + 5 points so that the last piece is ragged)."""
    import torch
    c = zk.Context(0)
    try:
        n = 3 * (1 << 20) + 5
        a, b, d_bases = _dlog_setup(c, n, 41)
        limbs = _np_scalars(n, 9, "witness")
        d_sc = torch.from_numpy(limbs.view(np.uint8).reshape(-1).copy()).cuda()
        want = bn.g1_mul(bn.G1_GEN, _dlog_expected(limbs, a, b))
        assert c.msm_points_limit() == 1 << 27
        c.set_option("lane_workspace_max_mb", 300)
        assert g16.g1_from_bytes(c.msm_g1_device(d_bases.data_ptr(), d_sc.data_ptr(), n)) == want
        lim = c.msm_points_limit()
        assert (1 << 16) <= lim <= (1 << 19) and lim & (lim - 1) == 0      # 2^20 points need ~0.6 GB: at least 2 halvings
        assert g16.g1_from_bytes(c.msm_g1_device(d_bases.data_ptr(), d_sc.data_ptr(), n)) == want   # remembered: no retry
        assert c.msm_points_limit() == lim
        c2 = zk.Context(0)                                                     # not even 2^16 points fit: an error, not a loop
        try:
            c2.set_option("lane_workspace_max_mb", 8)
            with pytest.raises(zk.ZkpoaError, match="over the lane limit"):
                c2.msm_g1_device(d_bases.data_ptr(), d_sc.data_ptr(), n)
        finally:
            c2.close()
        c.set_option("lane_workspace_max_mb", 0)                               # cap lifted: whole MSMs again
        assert c.msm_points_limit() == 1 << 27
        assert g16.g1_from_bytes(c.msm_g1_device(d_bases.data_ptr(), d_sc.data_ptr(), n)) == want
    finally:
        c.close()


def _dlog_setup(ctx, n, seed, group=1):
    import torch
    rng = random.Random(seed)
    a, b = rng.randrange(R), rng.randrange(R)
    size = 64 if group == 1 else 128
    d_bases = torch.empty(n * size, dtype=torch.uint8, device="cuda")
    (ctx.gen_bases_g1_device if group == 1 else ctx.gen_bases_g2_device)(a, b, 0, n, d_bases.data_ptr())
    return a, b, d_bases


def _np_scalars(n, seed, dist="uniform"):
    nr = np.random.default_rng(seed)
    limbs = nr.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64) * 2 + nr.integers(0, 2, size=(n, 4), dtype=np.uint64)
    limbs[:, 3] &= np.uint64((1 << 60) - 1)
    if dist == "witness":
        u = nr.random(n)
        small = u < 0.55
        limbs[small, 1:] = 0
        limbs[small, 0] = nr.integers(0, 2, size=int(small.sum()), dtype=np.uint64)
        limbs[(u >= 0.55) & (u < 0.9), 1:] = 0
    return limbs


def _dlog_expected(limbs, a, b):
    """(sum k_i (a + i b)) mod r from the limb columns, O(n) integer work."""
    n = limbs.shape[0]
    idx = np.arange(n, dtype=object)
    s0 = s1 = 0
    for j in range(4):
        col = limbs[:, j].astype(object)
        s0 += int(col.sum()) << (64 * j)
        s1 += int((col * idx).sum()) << (64 * j)
    return (a * s0 + b * s1) % R


@pytest.mark.parametrize("dist", ["uniform", "witness"])
def test_msm_g1_full_size_known_dlog(ctx, dist):
    """BASELINE.json configs[1]: 2^20 points. Bases (a + i b) G are generated on the device; the
    expected result is a field-only O(n) computation."""
    import torch
    n = 1 << 20
    a, b, d_bases = _dlog_setup(ctx, n, 99)
    hb = bytes(d_bases[:128].cpu().numpy())
    assert g16.g1_from_bytes(hb, 0) == bn.g1_mul(bn.G1_GEN, a)
    assert g16.g1_from_bytes(hb, 64) == bn.g1_mul(bn.G1_GEN, (a + b) % R)
    limbs = _np_scalars(n, 7, dist)
    d_sc = torch.from_numpy(limbs.view(np.uint8).reshape(-1).copy()).cuda()
    out = ctx.msm_g1_device(d_bases.data_ptr(), d_sc.data_ptr(), n)
    assert g16.g1_from_bytes(out) == bn.g1_mul(bn.G1_GEN, _dlog_expected(limbs, a, b))


def test_msm_linearity_full_size(ctx, zk):
    """MSM(k) + MSM(k') == MSM(k + k') at 2^20 (k + k' < r by construction)."""
    import torch
    n = 1 << 20
    _, _, d_bases = _dlog_setup(ctx, n, 5)
    k1 = _np_scalars(n, 1)
    k2 = _np_scalars(n, 2, "witness")
    k1[:, 3] >>= np.uint64(1)
    k2[:, 3] >>= np.uint64(1)
    ks = np.zeros_like(k1)
    carry = np.zeros(n, dtype=np.uint64)
    for j in range(4):
        s = k1[:, j] + k2[:, j]
        c1 = s < k1[:, j]
        s2 = s + carry
        c2 = s2 < s
        ks[:, j] = s2
        carry = (c1 | c2).astype(np.uint64)
    outs = []
    for arr in (k1, k2, ks):
        d = torch.from_numpy(arr.view(np.uint8).reshape(-1).copy()).cuda()
        outs.append(ctx.msm_g1_device(d_bases.data_ptr(), d.data_ptr(), n))
    assert zk.g1_sum(outs[0] + outs[1]) == outs[2]


def test_msm_g2_known_dlog(ctx):
    import torch
    n = 1 << 16
    a, b, d_bases = _dlog_setup(ctx, n, 123, group=2)
    limbs = _np_scalars(n, 3, "witness")
    d_sc = torch.from_numpy(limbs.view(np.uint8).reshape(-1).copy()).cuda()
    out = ctx.msm_g2_device(d_bases.data_ptr(), d_sc.data_ptr(), n)
    assert g16.g2_from_bytes(out) == bn.g2_mul(bn.G2_GEN, _dlog_expected(limbs, a, b))


# ---- NTT ----------------------------------------------------------------------------
def test_ntt_golden(ctx, vectors):
    for t in vectors["ntt"]:
        assert ctx.ntt(bytes.fromhex(t["in"]), t["k"]).hex() == t["fwd"]
        assert ctx.ntt(bytes.fromhex(t["in"]), t["k"], inverse=True).hex() == t["inv"]


@pytest.mark.parametrize("k", [0, 1, 2, 7, 10, 11, 12, 13, 15, 17])
def test_ntt_vs_c_oracle(ctx, k):
    """Covers the single-pass (k <= 11), two-pass (12..19) plans of the LDS transform."""
    nr = np.random.default_rng(k)
    limbs = nr.integers(0, 1 << 62, size=(1 << k, 4), dtype=np.uint64)
    limbs[:, 3] &= np.uint64((1 << 59) - 1)
    x = limbs.tobytes()
    assert ctx.ntt(x, k) == co.ntt(x, k)
    assert ctx.ntt(x, k, inverse=True) == co.ntt(x, k, inverse=True)


def test_ntt_roundtrip_and_linearity_large(ctx):
    """2^22 (three passes): ifft(fft(x)) == x and fft(x + y) == fft(x) + fft(y)."""
    k = 22
    nr = np.random.default_rng(0)
    xs = []
    for _ in range(2):
        limbs = nr.integers(0, 1 << 62, size=(1 << k, 4), dtype=np.uint64)
        limbs[:, 3] &= np.uint64((1 << 58) - 1)
        xs.append(limbs)
    x, y = xs[0].tobytes(), xs[1].tobytes()
    fx = ctx.ntt(x, k)
    assert ctx.ntt(fx, k, inverse=True) == x
    fy = ctx.ntt(y, k)
    s = (xs[0].astype(object) + 0)  # limb-wise sum without carries is not a field sum: use the device add
    xy = ctx.field_op(1, 1, x, y)
    assert ctx.ntt(xy, k) == ctx.field_op(1, 1, fx, fy)


def test_msm_concurrent_lanes(ctx):
    """Three MSMs in flight from three host threads on three lanes (what bench.py and the prover do)."""
    import torch
    from concurrent.futures import ThreadPoolExecutor
    n = 1 << 16
    a, b, d_bases = _dlog_setup(ctx, n, 31)
    inputs = []
    for lane in range(3):
        limbs = _np_scalars(n, 100 + lane, "witness" if lane == 1 else "uniform")
        inputs.append((limbs, torch.from_numpy(limbs.view(np.uint8).reshape(-1).copy()).cuda()))
    with ThreadPoolExecutor(3) as pool:
        for _ in range(3):
            futs = [pool.submit(ctx.msm_g1_device_lane, lane, d_bases.data_ptr(), inputs[lane][1].data_ptr(), n)
                    for lane in range(3)]
            for lane, f in enumerate(futs):
                assert g16.g1_from_bytes(f.result()) == bn.g1_mul(bn.G1_GEN, _dlog_expected(inputs[lane][0], a, b))
