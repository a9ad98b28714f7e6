"""Parity at the shapes BASELINE.json names, all three proof points checked (VERDICT r01 "pi_c is never checked
above 2^16"):

* configs[2]: full prove at the layer_one(2 sigs) shape -- n = 2^21, 2,083,343 wires, 1 public
  (tests/4_sigs_2_batches_12_height/benchmarks.txt:17-23). pi_a, pi_b, pi_c against the known-dlog expectation;
  the H scalars the GPU chain produced (buildABC -> 6 NTTs -> joinABC) are compared bit for bit with the C
  oracle's and, independently of any transform, through the oracle's quotient identity.
* G2 MSM at 2^20 (three-pass sort + short G2 pieces), known discrete log.
* configs[3] shapes (N = 1): layer_two(2, 12) -- n = 2^25, 21,356,921 wires, 2 public -- and layer_three(2) --
  n = 2^26, 52,367,163 wires, 13 public (tests/4_sigs_2_batches_12_height/benchmarks.txt:33-39, 49-55), same checks.
* configs[4] (N = 1): 2^26 MSM and the synthetic layer_one(128 sigs) prove, same checks (about a minute on the GPU
  box, ~25 GB of host memory; ZKPOA_SKIP_2P26=1 skips the 2^25 / 2^26 cases; bench.py runs the same checks on its
  2^26 lines).

Every expected group element is computed with the ORACLE's scalar multiplication (oracle/py/bn254.py), never with the
product's own host code.
"""
import os
import random
import time

import numpy as np
import pytest

from oracle import c_oracle as co
from oracle.py import bn254 as bn
from oracle.py import groth16 as g16
from test_gpu_kernels import _dlog_expected, _dlog_setup, _np_scalars

pytestmark = pytest.mark.gpu
R = bn.R
THREADS = min(16, os.cpu_count() or 1)


def _prove_and_check_all(ctx, zk, k, m, n_public, seed, oracle_h):
    from zkpoa_amd.synthetic import SyntheticCircuit
    circ = SyntheticCircuit(zk, ctx, k, m, n_public=n_public, seed=seed, witness_like=True)
    try:
        rng = random.Random(seed)
        r_, s_ = rng.randrange(R), rng.randrange(R)
        pts, pub = circ.prove(r_, s_)
        P = circ.h_scalars()                                   # what the H MSM consumed, read back from HBM
        assert P.shape == (1 << k, 4)
        t0 = time.time()
        for _ in range(2):                                     # two independent random points
            assert co.quotient_check(circ.coeff_section(), circ.w_limbs, m, k, P, rng.randrange(R), THREADS), \
                "GPU H scalars fail the quotient identity A(z)B(z) - C(z) = H(z)(z^n - 1)"
        t_q = time.time() - t0
        bad = P.copy()
        bad[rng.randrange(1 << k), 0] ^= np.uint64(1)
        assert not co.quotient_check(circ.coeff_section(), circ.w_limbs, m, k, bad, rng.randrange(R), THREADS)
        if oracle_h:                                           # bit-exact against the oracle's transform-based chain
            want = co.h_scalars(circ.coeff_section_bytes(), circ.witness_bytes(), m, k)
            assert P.tobytes() == want
        a, b, c = circ.expected_dlogs(r_, s_, P)               # discrete logs: integer arithmetic only
        assert g16.g1_from_bytes(pts, 0) == bn.g1_mul(bn.G1_GEN, a), "pi_a differs from the known-dlog expectation"
        assert g16.g2_from_bytes(pts, 64) == bn.g2_mul(bn.G2_GEN, b), "pi_b differs from the known-dlog expectation"
        assert g16.g1_from_bytes(pts, 192) == bn.g1_mul(bn.G1_GEN, c), "pi_c differs from the known-dlog expectation"
        assert circ.check(pts, r_, s_, P)                      # the product-side check bench.py's N > 1 path uses agrees
        wrong = bytearray(pts)
        wrong[192] ^= 1
        assert not circ.check(bytes(wrong), r_, s_, P)         # ... and does look at pi_c
        assert pub == circ.witness_bytes()[32:32 * (1 + n_public)]
        print("2^%d prove: quotient identity x2 in %.1f s on %d threads" % (k, t_q, THREADS))
    finally:
        circ.close()


def test_prove_layer_one_shape_all_points(ctx, zk):
    """BASELINE.json configs[2]."""
    _prove_and_check_all(ctx, zk, 21, 2083343, 1, 0x5EED0010, oracle_h=True)


def test_prove_layer_one_shape_under_memory_pressure(zk, capfd, monkeypatch):
    """The same proof with every MSM lane's workspace capped at 500 MB (a whole 2 M-point sort needs ~1 GB): before the
    proof starts the stages' workspaces are added up from their plans (csrc/prover.hip budget_lane_workspaces), the A, C
    and H MSMs go through their points in pieces, the B query cannot be sorted once for B1 and B2 -- each sorts its own
    pieces -- and all three proof points meet the known-dlog expectation. On a key of 2^27 constraints it is HBM itself
    that runs out, and the same path runs (include/zkpoa_prover.h, "Memory pressure")."""
    monkeypatch.setenv("ZKPOA_VERBOSE", "1")
    c = zk.Context(0)
    try:
        c.set_option("lane_workspace_max_mb", 500)
        _prove_and_check_all(c, zk, 21, 2083343, 1, 0x5EED0010, oracle_h=False)
        assert c.msm_points_limit() <= 1 << 20
    finally:
        c.close()
    err = capfd.readouterr().err
    assert "HBM budget" in err, err     # decided before the proof from the stages' plans, not by failed reservations
    assert "out of device memory" not in err and "over the lane limit" not in err, err


@pytest.mark.parametrize("dist", ["uniform", "witness"])
def test_msm_g2_full_size_known_dlog(ctx, dist):
    import torch
    n = 1 << 20
    a, b, d_bases = _dlog_setup(ctx, n, 321, group=2)
    limbs = _np_scalars(n, 11, dist)
    d_sc = torch.from_numpy(limbs.view(np.uint8).reshape(-1).copy()).cuda()
    out = ctx.msm_g2_device(d_bases.data_ptr(), d_sc.data_ptr(), n)
    assert g16.g2_from_bytes(out) == bn.g2_mul(bn.G2_GEN, _dlog_expected(limbs, a, b))


needs_2p26 = pytest.mark.skipif(os.environ.get("ZKPOA_SKIP_2P26") == "1", reason="ZKPOA_SKIP_2P26=1")


@needs_2p26
def test_msm_g1_2p26_known_dlog(ctx, zk):
    """BASELINE.json configs[4], N = 1: the standalone 2^26 MSM."""
    import torch
    from zkpoa_amd.synthetic import dlog_sums
    n = 1 << 26
    a, b, d_bases = _dlog_setup(ctx, n, 2026)
    limbs = _np_scalars(n, 26)
    d_sc = torch.from_numpy(limbs.view(np.uint8).reshape(-1)).cuda()
    out = ctx.msm_g1_device(d_bases.data_ptr(), d_sc.data_ptr(), n)
    s0, s1 = dlog_sums(limbs)
    assert g16.g1_from_bytes(out) == bn.g1_mul(bn.G1_GEN, (a * s0 + b * s1) % R)


@needs_2p26
def test_prove_layer_two_shape_all_points(ctx, zk):
    """BASELINE.json configs[3], layer two: L2(2, 12), tests/4_sigs_2_batches_12_height/benchmarks.txt:33-39."""
    _prove_and_check_all(ctx, zk, 25, 21356921, 2, 0x5EED0025, oracle_h=False)


@needs_2p26
def test_prove_layer_three_shape_all_points(ctx, zk):
    """BASELINE.json configs[3], layer three: L3(2), 13 public signals, tests/4_sigs_2_batches_12_height/benchmarks.txt:49-55."""
    _prove_and_check_all(ctx, zk, 26, 52367163, 13, 0x5EED0026, oracle_h=False)


@needs_2p26
def test_prove_2p26_all_points(ctx, zk):
    """BASELINE.json configs[4], N = 1: synthetic layer_one(128 sigs) shape (tests/old/128_sigs/benchmarks.txt:4-10)."""
    _prove_and_check_all(ctx, zk, 26, 61197000, 1, 0x5EED0010, oracle_h=False)


@pytest.mark.skipif(os.environ.get("ZKPOA_TEST_2P27") != "1", reason="opt-in (ZKPOA_TEST_2P27=1): ~2 minutes, ~60 GB of host memory")
def test_prove_2p27_layer_three_of_four_batches_shape(zk):
    """Beyond the reference's own runs: layer three grows by 24.2 M wires per batch, so four batches need a 2^27 domain
    (100.8 M wires). Key, chain buffers and five whole-MSM workspaces are then ~250 GB: the lanes' HBM budget
    (csrc/prover.hip budget_lane_workspaces) decides which MSMs go in pieces. Same checks as every other shape.
    (bench.py --workload prove_2p27_l3 / prove_2p28_l3 run the same checks: profiles/r04_bench_prove_2p2{7,8}_l3.json.log.)"""
    c = zk.Context(0)
    try:
        _prove_and_check_all(c, zk, 27, 100845225, 13, 0x5EED0027, oracle_h=False)
    finally:
        c.close()


def test_layer_one_shape_through_the_file_boundary_on_several_ranks_vs_c_oracle(ctx, zk, tmp_path):
    """BASELINE.json configs[3], layer-one leg, through the boundary the reference really has
    (scripts/g16_prove.sh:248-252 execs `prover <zkey> <wtns> <proof.json> <public.json>`; one such process per batch,
    scripts/full_workflow.sh:552): the L1(2 sigs) shape -- n = 2^21, 2,083,343 wires -- written as a real 1.08 GB .zkey and
    a 67 MB .wtns, proved by the executable on 1, 2 and 4 ranks (ZKPOA_DEVICES: real block-cyclic shards of 2^16 / 2^15
    wires read from the file's byte ranges, cyclic section 9, split chain with its two exchanges, the witness uploaded
    in slices). Every proof.json must hold exactly the points the C ORACLE's orc_prove computes from the same two files
    (~8 s on 16 host threads), and public.json its public signal."""
    import subprocess
    from zkpoa_amd.synthetic import SyntheticCircuit
    zp, wp = str(tmp_path / "circuit_final.zkey"), str(tmp_path / "witness.wtns")
    circ = SyntheticCircuit(zk, ctx, 21, 2083343, n_public=1, seed=0x5EED0021, witness_like=True)
    try:
        circ.write_zkey(zp)
        circ.write_wtns(wp)
    finally:
        circ.close()
    r_, s_ = 0x1234567890abcdef1234567, 0xfedcba0987654321
    t0 = time.time()
    want_pts, want_pub = co.prove(open(zp, "rb").read(), open(wp, "rb").read(), r_, s_, THREADS, n_public=1)
    t_cpu = time.time() - t0
    want = zk.proof_to_json(want_pts, "rapidsnark")
    base = dict(os.environ, ZKPOA_R=str(r_), ZKPOA_S=str(s_), ZKPOA_VERBOSE="1", ZKPOA_SELFCHECK="0")   # no valid vkey inside
    base.pop("ZKPOA_SERVER", None)
    for devices in ("0", "0,0", "0,0,0,0"):
        out = str(tmp_path / ("proof_%d.json" % len(devices)))
        rc = subprocess.run([zk.PROVER_BIN, zp, wp, out, str(tmp_path / "public.json")],
                            env=dict(base, ZKPOA_DEVICES=devices), capture_output=True, text=True, timeout=600)
        assert rc.returncode == 0, rc.stderr
        assert open(out).read() == want, "ranks %s: proof differs from the C oracle's" % devices
        assert open(tmp_path / "public.json").read() == zk.public_to_json(want_pub, "rapidsnark")
        if "," in devices:
            assert "H-scalar chain split" in rc.stderr and "block-cyclic" in rc.stderr
    # memory pressure on the one-shot path, whose stages learn their sizes as the file arrives: with every lane's workspace
    # capped at 500 MB (ZKPOA_LANE_WORKSPACE_MAX_MB; a whole 2 M-point sort needs ~1 GB) each reservation that does not
    # fit is answered by halving that MSM's piece, the B query falls back from its shared sort -- same proof
    out = str(tmp_path / "proof_capped.json")
    rc = subprocess.run([zk.PROVER_BIN, zp, wp, out, str(tmp_path / "public.json")],
                        env=dict(base, ZKPOA_DEVICES="0", ZKPOA_LANE_WORKSPACE_MAX_MB="500"), capture_output=True, text=True, timeout=600)
    assert rc.returncode == 0, rc.stderr
    assert open(out).read() == want, "capped workspaces: proof differs from the C oracle's"
    assert "B1 and B2 sort their own pieces" in rc.stderr and "continuing with at most" in rc.stderr, rc.stderr
    rc = subprocess.run([zk.PROVER_BIN, zp, wp, out, str(tmp_path / "public.json")],
                        env=dict(base, ZKPOA_DEVICES="0", ZKPOA_LANE_WORKSPACE_MAX_MB="lots"), capture_output=True, text=True, timeout=600)
    assert rc.returncode != 0 and "ZKPOA_LANE_WORKSPACE_MAX_MB" in rc.stderr
    # the card nearly full (another tenant: here a ballast tensor of this process): the key, its temporaries and five
    # whole-MSM workspaces want ~6 GB. Whatever is left, the command either writes the right proof -- the lanes halve their
    # pieces rather than take what the key being loaded still needs -- or fails with an out-of-memory message and no
    # output file; never a wrong proof, never a partial one.
    import torch
    outcomes = []
    for left_gb in (4.5, 5.25, 6.0):
        torch.cuda.empty_cache()
        free_b = torch.cuda.mem_get_info()[0]
        ballast = torch.empty(free_b - int(left_gb * 1e9), dtype=torch.uint8, device="cuda")
        try:
            out = str(tmp_path / ("proof_%.2f_left.json" % left_gb))
            rc = subprocess.run([zk.PROVER_BIN, zp, wp, out, str(tmp_path / "public.json")], env=dict(base, ZKPOA_DEVICES="0"),
                                capture_output=True, text=True, timeout=600)
            if rc.returncode == 0:
                assert open(out).read() == want, "%.2f GB left: proof differs from the C oracle's" % left_gb
                outcomes.append("proved" + (" in pieces" if "continuing with at most" in rc.stderr else ""))
            else:
                assert "memory" in rc.stderr and not os.path.exists(out), rc.stderr
                outcomes.append("out of memory")
        finally:
            del ballast
            torch.cuda.empty_cache()
    assert "proved" in " ".join(outcomes), outcomes
    print("card nearly full (4.5 / 5.25 / 6.0 GB left):", outcomes)
    print("L1 shape through the file boundary on 1 / 2 / 4 ranks == C oracle (orc_prove %.1f s on %d threads)" % (t_cpu, THREADS))
