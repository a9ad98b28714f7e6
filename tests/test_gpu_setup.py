"""Phase-2 setup arithmetic (csrc/setup.hip, C ABI zkpoa_setup_accumulate; `snarkjs zkey new`, g16_setup.sh:243-246):
out[s] = sum over the entries of signal s of coef * point. The points are k_i * G with known k_i (the oracle's
fixed-base products), so the expected output is (sum coef * k_i mod r) * G from the oracle -- bit-exact, G1 and G2,
with the coefficient shapes an R1CS has (1, -1 = r - 1, small constants, powers of two, full-width), a hot signal
(the constant-one wire), empty signals, single-entry signals, and the documented rejections."""
import os
import random

import numpy as np
import pytest

from conftest import le
from oracle import c_oracle as co
from oracle.py import bn254 as bn

pytestmark = pytest.mark.gpu
R = bn.R


def _dev(b):
    import torch
    return torch.frombuffer(bytearray(b), dtype=torch.uint8).cuda()


def _coef(rng):
    u = rng.random()
    if u < 0.35: return 1
    if u < 0.55: return R - 1
    if u < 0.70: return rng.randrange(2, 1 << 16)
    if u < 0.80: return 1 << rng.randrange(0, 253)
    if u < 0.85: return R - (1 << rng.randrange(0, 200))
    if u < 0.87: return 0
    return rng.randrange(R)


def _instance(rng, n_points, n_signals, nnz, hot=0.4):
    ks = [rng.randrange(R) for _ in range(n_points)]
    sig, pidx, coefs = [], [], []
    used = list(range(0, n_signals, 1))
    empties = set(rng.sample(used, max(1, n_signals // 7))) - {0} if n_signals > 3 else set()
    cand = [s for s in used if s not in empties]
    for _ in range(nnz):
        s = 0 if rng.random() < hot else rng.choice(cand)
        sig.append(s); pidx.append(rng.randrange(n_points)); coefs.append(_coef(rng))
    want = [0] * n_signals
    for s, i, c in zip(sig, pidx, coefs):
        want[s] = (want[s] + c * ks[i]) % R
    return ks, sig, pidx, coefs, want


@pytest.mark.parametrize("group,n_points,n_signals,nnz", [(1, 1, 1, 1), (1, 50, 40, 0), (1, 700, 300, 4000),
                                                          (1, 3000, 5000, 60000), (2, 200, 150, 1500),
                                                          (2, 900, 1000, 9000)])
def test_setup_accumulate_equals_oracle(ctx, group, n_points, n_signals, nnz):
    import torch
    rng = random.Random(77 * group + nnz)
    size = 64 if group == 1 else 128
    fb = co.fixed_base_g1 if group == 1 else co.fixed_base_g2
    ks, sig, pidx, coefs, want = _instance(rng, n_points, n_signals, nnz)
    pts = bytearray(fb(b"".join(le(k) for k in ks), 8))
    if n_points > 10:                       # a point at infinity among the inputs (an unused Lagrange slot)
        pts[size * 3:size * 4] = bytes(size)
        for j, (s, i, c) in enumerate(zip(sig, pidx, coefs)):
            if i == 3:
                want[s] = (want[s] - c * ks[3]) % R
    d_pts = _dev(pts)
    d_coef = _dev(b"".join(le(c) for c in coefs) or b"\0")
    d_pidx = torch.tensor(pidx or [0], dtype=torch.int64).to(torch.int32).cuda()
    d_sig = torch.tensor(sig or [0], dtype=torch.int64).to(torch.int32).cuda()
    d_out = torch.full((n_signals * size,), 0xAB, dtype=torch.uint8, device="cuda")
    ctx.setup_accumulate(group, d_pts.data_ptr(), n_points, d_coef.data_ptr(), d_pidx.data_ptr(), d_sig.data_ptr(),
                         nnz, n_signals, d_out.data_ptr())
    got = d_out.cpu().numpy().tobytes()
    assert got == fb(b"".join(le(w) for w in want), 8)
    # same entries in another order: same bytes
    if nnz > 1:
        perm = list(range(nnz)); rng.shuffle(perm)
        d_coef2 = _dev(b"".join(le(coefs[i]) for i in perm))
        d_pidx2 = torch.tensor([pidx[i] for i in perm], dtype=torch.int64).to(torch.int32).cuda()
        d_sig2 = torch.tensor([sig[i] for i in perm], dtype=torch.int64).to(torch.int32).cuda()
        d_out2 = torch.zeros_like(d_out)
        ctx.setup_accumulate(group, d_pts.data_ptr(), n_points, d_coef2.data_ptr(), d_pidx2.data_ptr(),
                             d_sig2.data_ptr(), nnz, n_signals, d_out2.data_ptr())
        assert d_out2.cpu().numpy().tobytes() == got


def test_setup_accumulate_rejections(ctx, zk):
    import torch
    pts = _dev(co.fixed_base_g1(le(5) + le(7), 1))
    out = torch.zeros(3 * 64, dtype=torch.uint8, device="cuda")
    i32 = lambda vals: torch.tensor(vals, dtype=torch.int64).to(torch.int32).cuda()
    ok_coef, bad_coef = _dev(le(1) + le(2)), _dev(le(1) + le(R))
    i01, i02, i03 = i32([0, 1]), i32([0, 2]), i32([0, 3])      # kept alive: the calls take raw pointers

    def call(group, coef, pidx, sig):
        ctx.setup_accumulate(group, pts.data_ptr(), 2, coef.data_ptr(), pidx.data_ptr(), sig.data_ptr(), 2, 3,
                             out.data_ptr())
    call(1, ok_coef, i01, i01)
    with pytest.raises(zk.ZkpoaError, match="out of range"):
        call(1, ok_coef, i02, i01)            # point index 2 of 2 points
    with pytest.raises(zk.ZkpoaError, match="out of range"):
        call(1, ok_coef, i01, i03)            # signal 3 of 3 signals
    with pytest.raises(zk.ZkpoaError, match="field element"):
        call(1, bad_coef, i01, i01)           # coefficient r
    with pytest.raises(zk.ZkpoaError, match="group"):
        call(3, ok_coef, i01, i01)
    call(1, ok_coef, i01, i01)                # the context is still usable


# ---- `snarkjs zkey new` on files: zkpoa_zkey_new / the zkpoa-setup executable --------------------------------------
def _setup_case(rng, n_vars, n_public, n_cons):
    from oracle.py import groth16 as g16
    from setup_files import write_ptau, write_r1cs
    cons, w = g16.random_circuit(rng, n_vars, n_public, n_cons)
    tox = {"tau": rng.randrange(2, R), "alpha": rng.randrange(2, R), "beta": rng.randrange(2, R), "gamma": 1, "delta": 1}
    want, vk = g16.synthetic_setup(n_vars, n_public, cons, tox,
                                   g1_batch=lambda s: co.fixed_base_g1(b"".join(le(k) for k in s), 8),
                                   g2_batch=lambda s: co.fixed_base_g2(b"".join(le(k) for k in s), 8))
    power = g16.read_zkey(want).domainSize.bit_length() - 1
    return cons, w, tox, want, vk, write_r1cs(n_vars, n_public, cons), power


@pytest.mark.parametrize("n_vars,n_public,n_cons,extra_power", [(12, 1, 5, 0), (40, 3, 60, 1), (300, 2, 500, 0),
                                                               (2000, 0, 4000, 2)])
def test_zkey_new_equals_the_setup_with_known_toxic_waste(ctx, zk, tmp_path, n_vars, n_public, n_cons, extra_power):
    """r1cs + ptau files -> zkey, byte for byte the key that the oracle derives directly from tau, alpha, beta (delta =
    gamma = 1, as `zkey new` leaves them); the key then proves (self-check against its own vkey on) and verifies."""
    from oracle.py import groth16 as g16
    from setup_files import write_ptau
    rng = random.Random(n_vars * 31 + n_cons)
    cons, w, tox, want, vk, r1cs, power = _setup_case(rng, n_vars, n_public, n_cons)
    (tmp_path / "c.r1cs").write_bytes(r1cs)
    (tmp_path / "pot.ptau").write_bytes(write_ptau(power + extra_power, tox["tau"], tox["alpha"], tox["beta"]))
    ctx.zkey_new(tmp_path / "c.r1cs", tmp_path / "pot.ptau", tmp_path / "c_0.zkey")
    got = (tmp_path / "c_0.zkey").read_bytes()
    assert got == want
    key = ctx.load_zkey(got)
    try:
        pts, pub = ctx.prove(key, g16.write_wtns(w), 11, 13)       # default self-check: first proof of a key
        assert zk.groth16_verify_points(key.vkey_points(), pts, pub)
    finally:
        key.close()


def test_zkpoa_setup_cli_and_rejections(ctx, zk, tmp_path):
    import subprocess
    from setup_files import write_ptau
    rng = random.Random(5)
    cons, w, tox, want, vk, r1cs, power = _setup_case(rng, 30, 2, 40)
    (tmp_path / "c.r1cs").write_bytes(r1cs)
    (tmp_path / "pot.ptau").write_bytes(write_ptau(power, tox["tau"], tox["alpha"], tox["beta"]))
    for words in ([], ["zkey", "new"], ["groth16", "setup"]):
        out = tmp_path / ("o%d.zkey" % len(words))
        rc = subprocess.run([zk.SETUP_BIN] + words + ["c.r1cs", "pot.ptau", str(out)], cwd=tmp_path,
                            capture_output=True, text=True, timeout=300)
        assert rc.returncode == 0, rc.stderr
        assert out.read_bytes() == want
    # a ceremony too small for the circuit
    (tmp_path / "small.ptau").write_bytes(write_ptau(power - 1, tox["tau"], tox["alpha"], tox["beta"]))
    with pytest.raises(zk.ZkpoaError, match="too small"):
        ctx.zkey_new(tmp_path / "c.r1cs", tmp_path / "small.ptau", tmp_path / "x.zkey")
    # not an r1cs; a coefficient that is not a field element; a ptau that was not prepared for phase 2
    (tmp_path / "bad.r1cs").write_bytes(b"r1cx" + r1cs[4:])
    with pytest.raises(zk.ZkpoaError, match="magic"):
        ctx.zkey_new(tmp_path / "bad.r1cs", tmp_path / "pot.ptau", tmp_path / "x.zkey")
    i = r1cs.index(le(1), 100)
    (tmp_path / "big.r1cs").write_bytes(r1cs[:i] + le(R) + r1cs[i + 32:])
    with pytest.raises(zk.ZkpoaError, match="field element"):
        ctx.zkey_new(tmp_path / "big.r1cs", tmp_path / "pot.ptau", tmp_path / "x.zkey")
    from oracle.py import groth16 as g16
    ptau = (tmp_path / "pot.ptau").read_bytes()
    secs = [(t, ptau[pos:pos + ln]) for t, lst in g16.read_binfile(ptau, "ptau", 1).items() for pos, ln in lst if t < 12]
    (tmp_path / "raw.ptau").write_bytes(g16.write_binfile("ptau", 1, secs))
    with pytest.raises(zk.ZkpoaError, match="prepared for phase 2"):
        ctx.zkey_new(tmp_path / "c.r1cs", tmp_path / "raw.ptau", tmp_path / "x.zkey")
    rc = subprocess.run([zk.SETUP_BIN, "only-one-arg"], capture_output=True, text=True)
    assert rc.returncode == 2 and "usage" in rc.stderr


def test_zkey_new_survives_mutated_inputs(ctx, zk, tmp_path):
    """Bytes of the two input files flipped, cut or overwritten with extreme counts: every call either writes a key
    or fails with a ZkpoaError -- no crash, no hang -- and the context stays usable."""
    from setup_files import write_ptau
    rng = random.Random(99)
    cons, w, tox, want, vk, r1cs, power = _setup_case(rng, 25, 2, 30)
    ptau = write_ptau(power, tox["tau"], tox["alpha"], tox["beta"])
    (tmp_path / "ok.r1cs").write_bytes(r1cs)
    (tmp_path / "ok.ptau").write_bytes(ptau)
    extremes = [b"\xff\xff\xff\xff", b"\0\0\0\0", b"\xff\xff\xff\x7f", b"\x01\0\0\x80"]
    outcomes = {"ok": 0, "rejected": 0}
    for it in range(80):
        which = it % 2
        b = bytearray(r1cs if which == 0 else ptau)
        for _ in range(rng.choice([1, 1, 2, 3])):
            op, i = rng.randrange(4), rng.randrange(min(len(b), 400 if rng.random() < 0.7 else len(b)))
            if op == 0: b[i] = rng.randrange(256)
            elif op == 1: b[i:i + 4] = rng.choice(extremes)
            elif op == 2: b = b[:max(1, rng.randrange(len(b)))]
            else: b[i:i + 8] = rng.randrange(1 << 64).to_bytes(8, "little")
        name = "m.r1cs" if which == 0 else "m.ptau"
        (tmp_path / name).write_bytes(bytes(b))
        try:
            ctx.zkey_new(tmp_path / ("m.r1cs" if which == 0 else "ok.r1cs"),
                         tmp_path / ("m.ptau" if which == 1 else "ok.ptau"), tmp_path / "m.zkey")
            outcomes["ok"] += 1
        except zk.ZkpoaError:
            outcomes["rejected"] += 1
    assert outcomes["rejected"] > 10
    ctx.zkey_new(tmp_path / "ok.r1cs", tmp_path / "ok.ptau", tmp_path / "final.zkey")
    assert (tmp_path / "final.zkey").read_bytes() == want


# ---- the arithmetic of `snarkjs zkey contribute`: delta <- d * delta, C and H <- C / d, H / d -----------------------
def _zkey_sections(buf):
    from oracle.py import groth16 as g16
    return {t: buf[lst[0][0]:lst[0][0] + lst[0][1]] for t, lst in g16.read_binfile(buf, "zkey", 1).items()}


def test_zkey_contribute_equals_the_setup_with_that_delta(ctx, zk, tmp_path):
    """new -> contribute(d1) -> contribute(d2): every section equals the oracle's key for delta = d1, then d1 * d2
    (gamma = 1), byte for byte; the contributed key proves (self-check on) and verifies, and the proof made under the
    old key does not verify under the new one."""
    import subprocess
    from oracle.py import groth16 as g16
    from setup_files import write_ptau
    rng = random.Random(404)
    n_vars, n_public, n_cons = 60, 2, 90
    cons, w, tox, want0, vk, r1cs, power = _setup_case(rng, n_vars, n_public, n_cons)
    (tmp_path / "c.r1cs").write_bytes(r1cs)
    (tmp_path / "pot.ptau").write_bytes(write_ptau(power, tox["tau"], tox["alpha"], tox["beta"]))
    ctx.zkey_new(tmp_path / "c.r1cs", tmp_path / "pot.ptau", tmp_path / "c_0.zkey")
    d1, d2 = rng.randrange(1, R), R - 5
    fb1 = lambda s: co.fixed_base_g1(b"".join(le(k) for k in s), 8)
    fb2 = lambda s: co.fixed_base_g2(b"".join(le(k) for k in s), 8)
    ctx.zkey_contribute(tmp_path / "c_0.zkey", tmp_path / "c_1.zkey", d1)
    want1, _ = g16.synthetic_setup(n_vars, n_public, cons, dict(tox, delta=d1), g1_batch=fb1, g2_batch=fb2)
    assert (tmp_path / "c_1.zkey").read_bytes() == want1
    # second contribution through the executable, snarkjs' command line (options ignored), secret from ZKPOA_DELTA
    rc = subprocess.run([zk.SETUP_BIN, "zkey", "contribute", "c_1.zkey", "c_final.zkey", "--name=First contributor",
                         "-e=random text for entropy"], cwd=tmp_path, capture_output=True, text=True, timeout=300,
                        env=dict(os.environ, ZKPOA_DELTA=str(d2)))
    assert rc.returncode == 0 and "WARNING" in rc.stderr, rc.stderr
    want2, _ = g16.synthetic_setup(n_vars, n_public, cons, dict(tox, delta=d1 * d2 % R), g1_batch=fb1, g2_batch=fb2)
    final = (tmp_path / "c_final.zkey").read_bytes()
    assert final == want2
    wt = g16.write_wtns(w)
    k0, k2 = ctx.load_zkey(want0), ctx.load_zkey(final)
    try:
        pts0, pub0 = ctx.prove(k0, wt, 3, 4)
        pts2, pub2 = ctx.prove(k2, wt, 3, 4)                         # self-check against the new delta
        assert pub0 == pub2 and pts0 != pts2
        assert zk.groth16_verify_points(k2.vkey_points(), pts2, pub2)
        assert not zk.groth16_verify_points(k2.vkey_points(), pts0, pub0)
    finally:
        k0.close(); k2.close()
    # a random secret (no ZKPOA_DELTA): a different, working key
    rc = subprocess.run([zk.SETUP_BIN, "zkey", "contribute", "c_0.zkey", "c_r.zkey"], cwd=tmp_path, capture_output=True,
                        text=True, timeout=300, env={k: v for k, v in os.environ.items() if k != "ZKPOA_DELTA"})
    assert rc.returncode == 0, rc.stderr
    rnd = (tmp_path / "c_r.zkey").read_bytes()
    s0, sr = _zkey_sections(want0), _zkey_sections(rnd)
    assert all(sr[t] == s0[t] for t in (1, 3, 4, 5, 6, 7, 10)) and sr[8] != s0[8] and sr[9] != s0[9] and sr[2] != s0[2]
    kr = ctx.load_zkey(rnd)
    try:
        ptsr, pubr = ctx.prove(kr, wt, 1, 2)
        assert zk.groth16_verify_points(kr.vkey_points(), ptsr, pubr)
    finally:
        kr.close()
    with pytest.raises(zk.ZkpoaError, match="delta"):
        ctx.zkey_contribute(tmp_path / "c_0.zkey", tmp_path / "x.zkey", 0)
    with pytest.raises(zk.ZkpoaError, match="delta"):
        ctx.zkey_contribute(tmp_path / "c_0.zkey", tmp_path / "x.zkey", R)


def test_setup_outputs_are_atomic_and_inputs_range_checked(ctx, zk, tmp_path):
    """(1) zkey new / contribute write <path>.tmp.<pid> and rename: a failing run leaves nothing under the
    final name (and no temporary); contributing IN PLACE (in == out) works; (2) a ptau / zkey coordinate >= q is
    rejected instead of flowing into a well-formed but wrong key."""
    from oracle.py import groth16 as g16
    from setup_files import write_ptau
    rng = random.Random(77)
    n_vars, n_public, n_cons = 40, 1, 50
    cons, w, tox, want0, vk, r1cs, power = _setup_case(rng, n_vars, n_public, n_cons)
    ptau = write_ptau(power, tox["tau"], tox["alpha"], tox["beta"])
    (tmp_path / "c.r1cs").write_bytes(r1cs)
    (tmp_path / "pot.ptau").write_bytes(ptau)
    # unwritable destination: error, nothing left behind
    with pytest.raises(zk.ZkpoaError):
        ctx.zkey_new(tmp_path / "c.r1cs", tmp_path / "pot.ptau", tmp_path / "no_such_dir" / "c.zkey")
    ctx.zkey_new(tmp_path / "c.r1cs", tmp_path / "pot.ptau", tmp_path / "c.zkey")
    assert (tmp_path / "c.zkey").read_bytes() == want0
    # in place: the output replaces the input atomically
    d1 = rng.randrange(1, R)
    ctx.zkey_contribute(tmp_path / "c.zkey", tmp_path / "c.zkey", d1)
    fb1 = lambda s: co.fixed_base_g1(b"".join(le(k) for k in s), 8)
    fb2 = lambda s: co.fixed_base_g2(b"".join(le(k) for k in s), 8)
    want1, _ = g16.synthetic_setup(n_vars, n_public, cons, dict(tox, delta=d1), g1_batch=fb1, g2_batch=fb2)
    assert (tmp_path / "c.zkey").read_bytes() == want1
    assert sorted(x.name for x in tmp_path.iterdir()) == ["c.r1cs", "c.zkey", "pot.ptau"]      # no *.tmp.* anywhere
    # a Lagrange-form ceremony point with x = q (not canonical): rejected, and no key appears
    secs = {t: lst[0] for t, lst in g16.read_binfile(ptau, "ptau", 1).items()}
    Q = bn.Q
    for sec_id in (12, 13, 14, 15):
        off, ln = secs[sec_id]
        bad = bytearray(ptau)
        unit = 128 if sec_id == 13 else 64
        at = off + ((1 << power) - 1) * unit                                   # first point of the circuit's level
        bad[at:at + 32] = le(Q)
        (tmp_path / "bad.ptau").write_bytes(bytes(bad))
        with pytest.raises(zk.ZkpoaError, match="field element"):
            ctx.zkey_new(tmp_path / "c.r1cs", tmp_path / "bad.ptau", tmp_path / "bad.zkey")
        assert not (tmp_path / "bad.zkey").exists()
    # ... and the same for a point of section 9 of a key handed to contribute
    zsecs = {t: lst[0] for t, lst in g16.read_binfile(want0, "zkey", 1).items()}
    badk = bytearray(want0)
    badk[zsecs[9][0]:zsecs[9][0] + 32] = le(Q + 1)
    (tmp_path / "badk.zkey").write_bytes(bytes(badk))
    with pytest.raises(zk.ZkpoaError, match="field element"):
        ctx.zkey_contribute(tmp_path / "badk.zkey", tmp_path / "out.zkey", 3)
    assert not (tmp_path / "out.zkey").exists()


# ---- zkey new at the reference's layer-one shape, every point section checked --------------------
def _weighted_dlog(rho, terms, dl_of_constraint):
    """sum over the terms (c, s, coef) of rho[s] * coef * dl(c) mod r, in vectorised 16-bit pieces (exact)."""
    c, s, coef = terms
    x = rho[s].astype(np.uint64) * dl_of_constraint(c).astype(np.uint64)        # < 2^29 * 2^35
    x16 = [((x >> np.uint64(16 * j)) & np.uint64(0xFFFF)) for j in range(4)]
    total = 0
    for i in range(16):
        ci = (coef[:, i // 4] >> np.uint64(16 * (i % 4))) & np.uint64(0xFFFF)
        if not ci.any():
            continue
        for j in range(4):
            total += int(np.dot(ci, x16[j])) << (16 * (i + j))                   # < 2^32 * 2^23 terms: no overflow
    return total % R


def test_zkey_new_layer_one_shape_known_dlog(ctx, zk, tmp_path):
    """`zkpoa-setup zkey new` at the layer_one(2 sigs) shape (2^21 domain, 2,083,343 wires, 8.4 M terms; shape from
    tests/4_sigs_2_batches_12_height/benchmarks.txt:17-23) on format-valid files whose ceremony points are known
    multiples of the generators. Every point section of the key is checked in full: for random 29-bit weights rho_s,
    sum_s rho_s * Section[s] (one MSM over the section, G1 or G2) must equal (sum_terms rho_s * coef * dlog(point)) * G
    computed with integer arithmetic and ONE oracle scalar multiplication -- a wrong point passes with probability
    2^-29. Section 9 (H) must be the odd points of the 2n-point Lagrange level; the key must load in the prover."""
    import subprocess
    import time
    from oracle.py import groth16 as g16
    from setup_files import PTAU_PROGRESSIONS, write_full_shape_inputs
    k, m, n_pub = 21, 2083343, 1
    n = 1 << k
    t0 = time.time()
    terms = write_full_shape_inputs(ctx, k, m, str(tmp_path), seed=3, n_public=n_pub)
    t_in = time.time() - t0
    t0 = time.time()
    rc = subprocess.run([zk.SETUP_BIN, "zkey", "new", "c.r1cs", "pot.ptau", "c_0.zkey"], cwd=tmp_path, capture_output=True,
                        text=True, timeout=600)
    assert rc.returncode == 0, rc.stderr
    t_new = time.time() - t0
    zkey = (tmp_path / "c_0.zkey").read_bytes()
    secs = {t: lst[0] for t, lst in g16.read_binfile(zkey, "zkey", 1).items()}
    sec = lambda t: zkey[secs[t][0]:secs[t][0] + secs[t][1]]
    assert secs[5][1] == m * 64 and secs[7][1] == m * 128 and secs[8][1] == (m - n_pub - 1) * 64 and secs[9][1] == n * 64
    nr = np.random.default_rng(99)
    rho = nr.integers(1, 1 << 29, size=m, dtype=np.int64)
    rho_bytes = np.zeros((m, 4), dtype=np.uint64)
    rho_bytes[:, 0] = rho.astype(np.uint64)
    rho_bytes = rho_bytes.tobytes()
    lvl = (1 << k) - 1                                    # first point of the circuit's Lagrange level in a section
    dl = lambda s: (lambda c: PTAU_PROGRESSIONS[s][0] + PTAU_PROGRESSIONS[s][1] * (lvl + c))
    pub_rows = (np.arange(terms["n_cons"], terms["n_cons"] + n_pub + 1, dtype=np.int64), np.arange(n_pub + 1, dtype=np.int64),
                np.tile(np.array([1, 0, 0, 0], dtype=np.uint64), (n_pub + 1, 1)))
    A_terms = tuple(np.concatenate([x, y]) for x, y in zip(terms["A"], pub_rows))
    # section 5 (A over tau*G1), 6 / 7 (B over tau*G1 / tau*G2)
    assert g16.g1_from_bytes(ctx.msm_g1(sec(5), rho_bytes, m)) == bn.g1_mul(bn.G1_GEN, _weighted_dlog(rho, A_terms, dl(12)))
    eB = _weighted_dlog(rho, terms["B"], dl(12))
    assert g16.g1_from_bytes(ctx.msm_g1(sec(6), rho_bytes, m)) == bn.g1_mul(bn.G1_GEN, eB)
    eB2 = _weighted_dlog(rho, terms["B"], dl(13))
    assert g16.g2_from_bytes(ctx.msm_g2(sec(7), rho_bytes, m)) == bn.g2_mul(bn.G2_GEN, eB2)
    # sections 3 + 8: IC and C are one vector K[s] = A over beta*tau*G1 + B over alpha*tau*G1 + C over tau*G1
    eK = (_weighted_dlog(rho, A_terms, dl(15)) + _weighted_dlog(rho, terms["B"], dl(14)) +
          _weighted_dlog(rho, terms["C"], dl(12))) % R
    assert g16.g1_from_bytes(ctx.msm_g1(sec(3) + sec(8), rho_bytes, m)) == bn.g1_mul(bn.G1_GEN, eK)
    # section 9: H[i] = point 2 i + 1 of the 2n-point level of section 12
    a12, b12 = PTAU_PROGRESSIONS[12]
    h = sec(9)
    for i in (0, 1, 12345, n // 2, n - 1):
        assert g16.g1_from_bytes(h, 64 * i) == bn.g1_mul(bn.G1_GEN, a12 + b12 * ((2 << k) - 1 + 2 * i + 1))
    rho_h = np.zeros((n, 4), dtype=np.uint64)
    rho_h[:, 0] = nr.integers(1, 1 << 20, size=n, dtype=np.uint64)         # 2^20 * 2^21 * 2^21 terms < 2^63
    idx = np.arange(n, dtype=np.uint64)
    eH = (int(rho_h[:, 0].sum()) * (a12 + b12 * ((2 << k) - 1 + 1)) + 2 * b12 * int(np.dot(rho_h[:, 0], idx))) % R
    assert g16.g1_from_bytes(ctx.msm_g1(h, rho_h.tobytes(), n)) == bn.g1_mul(bn.G1_GEN, eH)
    key = ctx.load_zkey(zkey)                             # section sizes, coordinate / coefficient range checks
    key.close()
    print("layer-one shape: inputs written in %.1f s, zkpoa-setup zkey new %.2f s, 1.08 GB key checked" % (t_in, t_new))


# ---- `snarkjs wtns check` -------------------------------------------------------------------------------------------
def test_wtns_check(ctx, zk, tmp_path):
    import subprocess
    from oracle.py import groth16 as g16
    from setup_files import write_r1cs
    rng = random.Random(8)
    n_vars, n_public, n_cons = 200, 3, 700
    cons, w = g16.random_circuit(rng, n_vars, n_public, n_cons)
    (tmp_path / "c.r1cs").write_bytes(write_r1cs(n_vars, n_public, cons))
    (tmp_path / "w.wtns").write_bytes(g16.write_wtns(w))
    assert ctx.wtns_check(tmp_path / "c.r1cs", tmp_path / "w.wtns") == (0, None)
    rc = subprocess.run([zk.SETUP_BIN, "wtns", "check", "c.r1cs", "w.wtns"], cwd=tmp_path, capture_output=True, text=True)
    assert rc.returncode == 0 and "WITNESS IS CORRECT" in rc.stdout, rc.stderr
    # break one wire: exactly the constraints that mention it (with a non-zero effect) fail; the oracle counts them
    bad_w = list(w)
    victim = n_vars - 5
    bad_w[victim] = (bad_w[victim] + 1) % R
    lc = lambda d, ww: sum(v * ww[s] for s, v in d.items()) % R
    failing = [c for c, (a, b, cc) in enumerate(cons) if lc(a, bad_w) * lc(b, bad_w) % R != lc(cc, bad_w)]
    assert failing
    (tmp_path / "bad.wtns").write_bytes(g16.write_wtns(bad_w))
    assert ctx.wtns_check(tmp_path / "c.r1cs", tmp_path / "bad.wtns") == (len(failing), failing[0])
    rc = subprocess.run([zk.SETUP_BIN, "wtns", "check", "c.r1cs", "bad.wtns"], cwd=tmp_path, capture_output=True, text=True)
    assert rc.returncode == 1 and ("#%d" % failing[0]) in rc.stderr
    # malformed: wrong number of values; a value >= r
    (tmp_path / "short.wtns").write_bytes(g16.write_wtns(w[:-1]))
    with pytest.raises(zk.ZkpoaError, match="wires"):
        ctx.wtns_check(tmp_path / "c.r1cs", tmp_path / "short.wtns")
    raw = bytearray(g16.write_wtns(w))
    raw[-32:] = le(R)
    (tmp_path / "big.wtns").write_bytes(bytes(raw))
    with pytest.raises(zk.ZkpoaError, match="field element"):
        ctx.wtns_check(tmp_path / "c.r1cs", tmp_path / "big.wtns")
