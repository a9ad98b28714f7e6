"""ORACLE (test infrastructure, not product code) -- Groth16 over BN254 in plain Python ints.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.

Restates what runs at the reference's prove call site scripts/g16_prove.sh:246-260
(`prover <zkey> <wtns> <proof.json> <public.json>`, rapidsnark, or `snarkjs groth16 prove`)
and the acceptance check scripts/g16_verify.sh:213-216 (`snarkjs groth16 verify`).
The algorithm lives in snarkjs 0.7.2 (pnpm-lock.yaml:2231) / ffjavascript 0.2.62 /
@iden3/binfileutils 0.0.11 (pnpm-lock.yaml:493-497), none of which is under /root/reference;
this file follows their published algorithm (groth16_prove.js: buildABC1, ifft/batchApplyKey/fft,
joinABC, five multiExpAffine, randomised assembly) as specified in SURVEY.md 3.2 and 8c.

Pinning: verify() + the two JSON writers are pinned by reference fixtures (see
tests/test_oracle_fixtures.py). prove() is pinned only indirectly: its proofs for synthetic
zkeys must verify under that pinned verifier ("prover parity unpinned" otherwise).
"""
import json
import struct

from . import bn254 as bn
from .bn254 import Q, R, FQ, FQ2, MONT_R


# ----------------------------------------------------------------------------- binfile container
def write_binfile(magic, version, sections):
    """sections: list of (id, bytes). Layout: magic[4] u32 version u32 nSections {u32 id u64 len payload}*"""
    out = bytearray()
    out += magic.encode("ascii")
    out += struct.pack("<II", version, len(sections))
    for sid, payload in sections:
        out += struct.pack("<IQ", sid, len(payload))
        out += payload
    return bytes(out)


def read_binfile(buf, magic, max_version):
    if buf[:4] != magic.encode("ascii"):
        raise ValueError("%s: invalid file format (magic)" % magic)
    version, nsec = struct.unpack_from("<II", buf, 4)
    if version > max_version:
        raise ValueError("version not supported")
    pos = 12
    sections = {}
    for _ in range(nsec):
        sid, ln = struct.unpack_from("<IQ", buf, pos)
        pos += 12
        sections.setdefault(sid, []).append((pos, ln))
        pos += ln
    if pos != len(buf):
        raise ValueError("binfile: trailing or truncated data")
    return sections


def _le32(x): return int(x).to_bytes(32, "little")
def _rd32(buf, off): return int.from_bytes(buf[off:off + 32], "little")


# point <-> zkey bytes (affine, Montgomery, LE, infinity = zeros)
def g1_to_bytes(P):
    if P is None:
        return bytes(64)
    return _le32(bn.to_mont(P[0], Q)) + _le32(bn.to_mont(P[1], Q))


def g2_to_bytes(P):
    if P is None:
        return bytes(128)
    (x0, x1), (y0, y1) = P
    return b"".join(_le32(bn.to_mont(v, Q)) for v in (x0, x1, y0, y1))


def g1_from_bytes(buf, off=0):
    x, y = _rd32(buf, off), _rd32(buf, off + 32)
    if x == 0 and y == 0:
        return None
    return (bn.from_mont(x, Q), bn.from_mont(y, Q))


def g2_from_bytes(buf, off=0):
    v = [_rd32(buf, off + 32 * i) for i in range(4)]
    if not any(v):
        return None
    v = [bn.from_mont(x, Q) for x in v]
    return ((v[0], v[1]), (v[2], v[3]))


# ----------------------------------------------------------------------------- wtns
def write_wtns(witness):
    sec1 = struct.pack("<I", 32) + _le32(R) + struct.pack("<I", len(witness))
    sec2 = b"".join(_le32(w % R) for w in witness)
    return write_binfile("wtns", 2, [(1, sec1), (2, sec2)])


def read_wtns(buf):
    secs = read_binfile(buf, "wtns", 2)
    p, _ = secs[1][0]
    n8 = struct.unpack_from("<I", buf, p)[0]
    assert n8 == 32
    prime = _rd32(buf, p + 4)
    nwit = struct.unpack_from("<I", buf, p + 36)[0]
    p2, l2 = secs[2][0]
    assert l2 == nwit * 32
    return prime, [_rd32(buf, p2 + 32 * i) for i in range(nwit)]


# ----------------------------------------------------------------------------- zkey
class ZKey:
    pass


def read_zkey(buf):
    secs = read_binfile(buf, "zkey", 2)
    zk = ZKey()
    p, _ = secs[1][0]
    zk.protocol = struct.unpack_from("<I", buf, p)[0]
    if zk.protocol != 1:
        raise ValueError("zkey file is not groth16")
    p, _ = secs[2][0]
    n8q = struct.unpack_from("<I", buf, p)[0]; p += 4
    zk.q = int.from_bytes(buf[p:p + n8q], "little"); p += n8q
    n8r = struct.unpack_from("<I", buf, p)[0]; p += 4
    zk.r = int.from_bytes(buf[p:p + n8r], "little"); p += n8r
    zk.nVars, zk.nPublic, zk.domainSize = struct.unpack_from("<III", buf, p); p += 12
    zk.alpha1 = g1_from_bytes(buf, p); p += 64
    zk.beta1 = g1_from_bytes(buf, p); p += 64
    zk.beta2 = g2_from_bytes(buf, p); p += 128
    zk.gamma2 = g2_from_bytes(buf, p); p += 128
    zk.delta1 = g1_from_bytes(buf, p); p += 64
    zk.delta2 = g2_from_bytes(buf, p); p += 128
    p, l = secs[3][0]
    zk.IC = [g1_from_bytes(buf, p + 64 * i) for i in range(l // 64)]
    p, l = secs[4][0]
    ncoef = struct.unpack_from("<I", buf, p)[0]
    assert l == 4 + ncoef * 44
    zk.coefs = []
    for i in range(ncoef):
        o = p + 4 + 44 * i
        m, c, s = struct.unpack_from("<III", buf, o)
        zk.coefs.append((m, c, s, _rd32(buf, o + 12)))          # raw value (= coef * R^2 mod r)
    def pts(sid, size, rd):
        p, l = secs[sid][0]
        return [rd(buf, p + size * i) for i in range(l // size)]
    zk.A = pts(5, 64, g1_from_bytes)
    zk.B1 = pts(6, 64, g1_from_bytes)
    zk.B2 = pts(7, 128, g2_from_bytes)
    zk.C = pts(8, 64, g1_from_bytes)
    zk.H = pts(9, 64, g1_from_bytes)
    return zk


def fr_lagrange_at(tau, n):
    """[L_c(tau)] for the size-n domain {w^c}, w = w[log2 n]."""
    k = n.bit_length() - 1
    w = bn.fr_root_of_unity(k)
    zt = (pow(tau, n, R) - 1) % R
    ninv = pow(n, -1, R)
    out = []
    wc = 1
    for _ in range(n):
        out.append(zt * ninv % R * wc % R * pow((tau - wc) % R, -1, R) % R)
        wc = wc * w % R
    return out


def synthetic_setup(nVars, nPublic, constraints, toxic, g1_batch=None, g2_batch=None):
    """Build a real-format .zkey for an R1CS from known toxic waste.

    constraints: list of (lcA, lcB, lcC), each lc a dict {signal: coef}.
    toxic: dict tau, alpha, beta, gamma, delta (ints mod r).
    g1_batch / g2_batch: optional callables [scalars] -> concatenated wire-format bytes of k*G
    (e.g. the C oracle's fixed-base routine, for circuits too large for Python scalar muls).
    Returns (zkey_bytes, vkey_dict). Mirrors snarkjs `zkey new` output conventions
    (SURVEY.md 8c): nPublic+1 extra A rows, coefficient values scaled by R^2, H = odd
    points of the size-2n Lagrange basis divided by delta.
    """
    g1_batch = g1_batch or (lambda ks: b"".join(g1_to_bytes(bn.g1_mul(bn.G1_GEN, k)) for k in ks))
    g2_batch = g2_batch or (lambda ks: b"".join(g2_to_bytes(bn.g2_mul(bn.G2_GEN, k)) for k in ks))
    tau, alpha, beta, gamma, delta = (toxic[k] % R for k in ("tau", "alpha", "beta", "gamma", "delta"))
    ncons = len(constraints)
    n = 1
    while n < ncons + nPublic + 1:
        n <<= 1
    L = fr_lagrange_at(tau, n)
    At = [0] * nVars
    Bt = [0] * nVars
    Ct = [0] * nVars
    coefs = []
    for c, (la, lb, lc) in enumerate(constraints):
        for s, v in sorted(la.items()):
            At[s] = (At[s] + v * L[c]) % R
            coefs.append((0, c, s, v % R))
        for s, v in sorted(lb.items()):
            Bt[s] = (Bt[s] + v * L[c]) % R
            coefs.append((1, c, s, v % R))
        for s, v in lc.items():
            Ct[s] = (Ct[s] + v * L[c]) % R
    for i in range(nPublic + 1):
        At[i] = (At[i] + L[ncons + i]) % R
        coefs.append((0, ncons + i, i, 1))
    dinv = pow(delta, -1, R)
    ginv = pow(gamma, -1, R)
    L2 = fr_lagrange_at(tau, 2 * n)
    r2 = MONT_R * MONT_R % R

    K = [(beta * At[i] + alpha * Bt[i] + Ct[i]) % R for i in range(nVars)]
    hdr1 = g1_batch([alpha, beta, delta])
    hdr2 = g2_batch([beta, gamma, delta])
    alpha1_b, beta1_b, delta1_b = hdr1[0:64], hdr1[64:128], hdr1[128:192]
    beta2_b, gamma2_b, delta2_b = hdr2[0:128], hdr2[128:256], hdr2[256:384]
    sec2 = struct.pack("<I", 32) + _le32(Q) + struct.pack("<I", 32) + _le32(R)
    sec2 += struct.pack("<III", nVars, nPublic, n)
    sec2 += alpha1_b + beta1_b + beta2_b + gamma2_b + delta1_b + delta2_b
    sec3 = g1_batch([K[i] * ginv % R for i in range(nPublic + 1)])
    sec4 = struct.pack("<I", len(coefs)) + b"".join(
        struct.pack("<III", m, c, s) + _le32(v * r2 % R) for (m, c, s, v) in coefs)
    sec5 = g1_batch(At)
    sec6 = g1_batch(Bt)
    sec7 = g2_batch(Bt)
    sec8 = g1_batch([K[i] * dinv % R for i in range(nPublic + 1, nVars)])
    sec9 = g1_batch([L2[2 * i + 1] * dinv % R for i in range(n)])
    sec10 = bytes(64) + struct.pack("<I", 0)
    zkey = write_binfile("zkey", 1, [(1, struct.pack("<I", 1)), (2, sec2), (3, sec3), (4, sec4), (5, sec5),
                                     (6, sec6), (7, sec7), (8, sec8), (9, sec9), (10, sec10)])
    IC = [g1_from_bytes(sec3, 64 * i) for i in range(nPublic + 1)]
    vkey = {"protocol": "groth16", "curve": "bn128", "nPublic": nPublic,
            "vk_alpha_1": g1_to_obj(g1_from_bytes(alpha1_b)), "vk_beta_2": g2_to_obj(g2_from_bytes(beta2_b)),
            "vk_gamma_2": g2_to_obj(g2_from_bytes(gamma2_b)), "vk_delta_2": g2_to_obj(g2_from_bytes(delta2_b)),
            "IC": [g1_to_obj(P) for P in IC]}
    return zkey, vkey


# ----------------------------------------------------------------------------- JSON object forms
def g1_to_obj(P):
    if P is None:
        return ["0", "1", "0"]
    return [str(P[0]), str(P[1]), "1"]


def g2_to_obj(P):
    if P is None:
        return [["0", "0"], ["1", "0"], ["0", "0"]]
    return [[str(P[0][0]), str(P[0][1])], [str(P[1][0]), str(P[1][1])], ["1", "0"]]


def g1_from_obj(o):
    if int(o[2]) == 0:
        return None
    return (int(o[0]), int(o[1]))


def g2_from_obj(o):
    if int(o[2][0]) == 0 and int(o[2][1]) == 0:
        return None
    return ((int(o[0][0]), int(o[0][1])), (int(o[1][0]), int(o[1][1])))


def proof_json_rapidsnark(proof):
    """Byte format of rapidsnark's proof.json as committed in the reference fixtures
    (tests/4_sigs_2_batches_12_height/layer_one/batch_0/proof.json): one line, no spaces,
    keys pi_a, pi_b, pi_c, protocol, no trailing newline."""
    obj = {"pi_a": g1_to_obj(proof["pi_a"]), "pi_b": g2_to_obj(proof["pi_b"]),
           "pi_c": g1_to_obj(proof["pi_c"]), "protocol": "groth16"}
    return json.dumps(obj, separators=(",", ":"))


def public_json_rapidsnark(public):
    return json.dumps([str(v) for v in public], separators=(",", ":"))


def proof_json_snarkjs(proof):
    """Byte format of snarkjs's proof.json (JSON.stringify(obj, null, 1)) as committed in
    experiments/scripts/groth16_input_prep/proof.json."""
    obj = {"pi_a": g1_to_obj(proof["pi_a"]), "pi_b": g2_to_obj(proof["pi_b"]),
           "pi_c": g1_to_obj(proof["pi_c"]), "protocol": "groth16", "curve": "bn128"}
    return json.dumps(obj, indent=1)


def public_json_snarkjs(public):
    return json.dumps([str(v) for v in public], indent=1)


# ----------------------------------------------------------------------------- prove
def build_abc(zk, witness):
    """groth16_prove.js::buildABC1 -- returns A_T, B_T, C_T in Montgomery form (ints)."""
    n = zk.domainSize
    out = [[0] * n, [0] * n]
    for m, c, s, val in zk.coefs:
        out[m][c] = (out[m][c] + bn.mont_mul(val, witness[s], R)) % R
    C = [bn.mont_mul(a, b, R) for a, b in zip(out[0], out[1])]
    return out[0], out[1], C


def coset_inc(power):
    return bn.FR_SHIFT if power == bn.FR_S else bn.fr_root_of_unity(power + 1)


def to_odd_coset(evals):
    """ifft -> batchApplyKey(1, inc) -> fft (all linear, so Montgomery form is preserved)."""
    n = len(evals)
    power = n.bit_length() - 1
    coefs = bn.ntt(evals, inverse=True)
    inc = coset_inc(power)
    x = 1
    for j in range(n):
        coefs[j] = coefs[j] * x % R
        x = x * inc % R
    return bn.ntt(coefs)


def h_scalars(zk, witness):
    """joinABC output: (A_odd*B_odd - C_odd) from Montgomery to standard form."""
    A, B, C = build_abc(zk, witness)
    Ao, Bo, Co = to_odd_coset(A), to_odd_coset(B), to_odd_coset(C)
    return [bn.from_mont((bn.mont_mul(a, b, R) - c) % R, R) for a, b, c in zip(Ao, Bo, Co)]


def prove(zkey_bytes, wtns_bytes, r=0, s=0, msm_g1=None, msm_g2=None):
    """(zkey, wtns, r, s) -> (proof dict of affine points, public signal list)."""
    msm_g1 = msm_g1 or (lambda pts, sc: bn.msm_naive(pts, sc, FQ))
    msm_g2 = msm_g2 or (lambda pts, sc: bn.msm_naive(pts, sc, FQ2))
    zk = read_zkey(zkey_bytes)
    prime, w = read_wtns(wtns_bytes)
    if prime != zk.r:
        raise ValueError("Curve of the witness does not match the curve of the proving key")
    if len(w) != zk.nVars:
        raise ValueError("Invalid witness length. Circuit: %d, witness: %d" % (zk.nVars, len(w)))
    P = h_scalars(zk, w)
    pi_a = msm_g1(zk.A, w)
    pib1 = msm_g1(zk.B1, w)
    pi_b = msm_g2(zk.B2, w)
    pi_c = msm_g1(zk.C, w[zk.nPublic + 1:])
    resH = msm_g1(zk.H, P)
    r %= R
    s %= R
    pi_a = bn.g1_add(bn.g1_add(pi_a, zk.alpha1), bn.g1_mul(zk.delta1, r))
    pi_b = bn.g2_add(bn.g2_add(pi_b, zk.beta2), bn.g2_mul(zk.delta2, s))
    pib1 = bn.g1_add(bn.g1_add(pib1, zk.beta1), bn.g1_mul(zk.delta1, s))
    pi_c = bn.g1_add(pi_c, resH)
    pi_c = bn.g1_add(pi_c, bn.g1_mul(pi_a, s))
    pi_c = bn.g1_add(pi_c, bn.g1_mul(pib1, r))
    pi_c = bn.g1_add(pi_c, bn.g1_mul(zk.delta1, (-(r * s)) % R))
    proof = {"pi_a": pi_a, "pi_b": pi_b, "pi_c": pi_c}
    public = [w[i] for i in range(1, zk.nPublic + 1)]
    return proof, public


# ----------------------------------------------------------------------------- verify
def verify(vkey, public, proof):
    """snarkjs groth16 verify: e(-A,B) e(alpha,beta) e(vk_x,gamma) e(C,delta) == 1.
    vkey/proof are the JSON objects (decimal strings), public a list of decimal strings/ints."""
    IC = [g1_from_obj(o) for o in vkey["IC"]]
    pub = [int(v) for v in public]
    if len(pub) + 1 != len(IC):
        return False
    if any(v >= R for v in pub):
        return False
    A = g1_from_obj(proof["pi_a"])
    B = g2_from_obj(proof["pi_b"])
    C = g1_from_obj(proof["pi_c"])
    if not (bn.g1_is_on_curve(A) and bn.g2_is_on_curve(B) and bn.g1_is_on_curve(C)):
        return False
    vk_x = IC[0]
    for v, P in zip(pub, IC[1:]):
        vk_x = bn.g1_add(vk_x, bn.g1_mul(P, v))
    alpha1 = g1_from_obj(vkey["vk_alpha_1"])
    beta2 = g2_from_obj(vkey["vk_beta_2"])
    gamma2 = g2_from_obj(vkey["vk_gamma_2"])
    delta2 = g2_from_obj(vkey["vk_delta_2"])
    return bn.pairing_product_is_one([
        (bn.ec_neg(A, FQ), B), (alpha1, beta2), (vk_x, gamma2), (C, delta2)])


def proof_to_obj(proof):
    return {"pi_a": g1_to_obj(proof["pi_a"]), "pi_b": g2_to_obj(proof["pi_b"]),
            "pi_c": g1_to_obj(proof["pi_c"]), "protocol": "groth16"}


# ----------------------------------------------------------------------------- small random circuits
def random_circuit(rng, nVars, nPublic, nConstraints):
    """Random satisfiable R1CS: (w[a]+k*w[b]) * w[d] = w[e], witness filled forward.
    Returns (constraints, witness). Signal 0 is the constant 1."""
    assert nVars > nPublic + 3
    w = [1] + [rng.randrange(R) for _ in range(nVars - 1)]
    cons = []
    # the first free "output" wire index; constraints define wires from the back
    first_out = max(nPublic + 1, nVars - nConstraints)
    for c in range(nConstraints):
        e = first_out + (c % (nVars - first_out))
        a, b, d = (rng.randrange(0, first_out) for _ in range(3))
        k = rng.choice([1, 2, R - 1, rng.randrange(R)])
        if c >= nVars - first_out:
            # wire e already fixed: make a constraint that is satisfied with a constant on the C side
            lhs = (w[a] + k * w[b]) % R * w[d] % R
            cons.append(({a: 1, b: k} if a != b else {a: (1 + k) % R}, {d: 1}, {0: lhs}))
        else:
            w[e] = (w[a] + k * w[b]) % R * w[d] % R
            cons.append(({a: 1, b: k} if a != b else {a: (1 + k) % R}, {d: 1}, {e: 1}))
    return cons, w
