"""ORACLE (test infrastructure, not product code) -- BN254 big-integer arithmetic.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.

CPU restatement, in plain Python integers, of the arithmetic that the reference delegates
to third-party code that is NOT vendored under /root/reference:
  * snarkjs 0.7.2      (reference pin: pnpm-lock.yaml:2231-2244, package.json:18)
  * ffjavascript 0.2.62 (pnpm-lock.yaml:1406-1411)   -- F1Field / F2Field / EC / bn128
  * wasmcurves 0.2.2    (pnpm-lock.yaml:2459-2462)   -- pairing, Montgomery form R = 2^256
The reference's own call sites of that arithmetic are scripts/g16_prove.sh:248-259 (prove)
and scripts/g16_verify.sh:213-216 (verify).

Pinning status: the pairing/verifier in this file is pinned by the reference's committed
fixtures (tests/4_sigs_2_batches_12_height/layer_*/*_vkey.json + proof.json/public.json,
and every sanitized_proof.json's `negalfa1xbeta2`), see tests/test_oracle_fixtures.py.
The (zkey, wtns) -> proof map itself has no golden vector in the reference:
"prover parity unpinned" except through "oracle proof verifies under the pinned verifier".
"""

# ----------------------------------------------------------------------------- constants
Q = 21888242871839275222246405745257275088696311157297823662689037894645226208583
R = 21888242871839275222246405745257275088548364400416034343698204186575808495617
MONT_R = 1 << 256                  # Montgomery radix used by the zkey point/coef encoding
FR_S = 28                          # two-adicity of R-1
FR_NQR = 5                         # smallest quadratic non-residue mod R (ffjavascript F1Field)
G1_GEN = (1, 2)
G2_GEN = (
    (10857046999023057135944570762232829481370756359578518086990519993285655852781,
     11559732032986387107991004021392285783925812861821192530917403151452391805634),
    (8495653923123431417604973247489272438418190587263600148770280649306958101930,
     4082367875863433681332203403145435568316851327593401208105741076214120093531),
)
ATE_LOOP_COUNT = 29793968203157093288     # 6x+2, x = 4965661367192848881
LOG_ATE_LOOP_COUNT = 63
BN_X = 4965661367192848881


def fr_root_of_unity(k):
    """w[k]: primitive 2^k-th root of unity of Fr, w[28] = 5^((r-1)/2^28), w[k] = w[k+1]^2."""
    assert 0 <= k <= FR_S
    w = pow(FR_NQR, (R - 1) >> FR_S, R)
    for _ in range(FR_S - k):
        w = w * w % R
    return w


FR_SHIFT = FR_NQR * FR_NQR % R     # Fr.shift in ffjavascript (= 25)


# ----------------------------------------------------------------------------- field "ops" objects
class FqOps:
    """Prime field of order p with plain ints as elements."""

    def __init__(self, p):
        self.p = p
        self.zero = 0
        self.one = 1

    def add(self, a, b): return (a + b) % self.p
    def sub(self, a, b): return (a - b) % self.p
    def neg(self, a): return (-a) % self.p
    def mul(self, a, b): return a * b % self.p
    def sqr(self, a): return a * a % self.p
    def inv(self, a): return pow(a, -1, self.p)
    def eq(self, a, b): return (a - b) % self.p == 0
    def is_zero(self, a): return a % self.p == 0
    def muli(self, a, k): return a * k % self.p


class Fq2Ops:
    """Fq2 = Fq[u]/(u^2+1); elements are (c0, c1) tuples."""

    def __init__(self, p):
        self.p = p
        self.zero = (0, 0)
        self.one = (1, 0)

    def add(self, a, b): return ((a[0] + b[0]) % self.p, (a[1] + b[1]) % self.p)
    def sub(self, a, b): return ((a[0] - b[0]) % self.p, (a[1] - b[1]) % self.p)
    def neg(self, a): return ((-a[0]) % self.p, (-a[1]) % self.p)

    def mul(self, a, b):
        p = self.p
        return ((a[0] * b[0] - a[1] * b[1]) % p, (a[0] * b[1] + a[1] * b[0]) % p)

    def sqr(self, a): return self.mul(a, a)

    def inv(self, a):
        p = self.p
        d = pow(a[0] * a[0] + a[1] * a[1], -1, p)
        return (a[0] * d % p, (-a[1]) * d % p)

    def eq(self, a, b): return (a[0] - b[0]) % self.p == 0 and (a[1] - b[1]) % self.p == 0
    def is_zero(self, a): return a[0] % self.p == 0 and a[1] % self.p == 0
    def muli(self, a, k): return (a[0] * k % self.p, a[1] * k % self.p)


class Fq12Ops:
    """Fq12 = Fq[t]/(t^12 - 18 t^6 + 82); elements are 12-tuples c0..c11 (t^6 = 9 + u)."""

    def __init__(self, p):
        self.p = p
        self.zero = (0,) * 12
        self.one = (1,) + (0,) * 11

    def add(self, a, b): return tuple((x + y) % self.p for x, y in zip(a, b))
    def sub(self, a, b): return tuple((x - y) % self.p for x, y in zip(a, b))
    def neg(self, a): return tuple((-x) % self.p for x in a)
    def muli(self, a, k): return tuple(x * k % self.p for x in a)

    def mul(self, a, b):
        p = self.p
        t = [0] * 23
        for i, x in enumerate(a):
            if x:
                for j, y in enumerate(b):
                    t[i + j] += x * y
        # t^12 = 18 t^6 - 82
        for k in range(22, 11, -1):
            v = t[k]
            if v:
                t[k - 6] += 18 * v
                t[k - 12] -= 82 * v
        return tuple(x % p for x in t[:12])

    def sqr(self, a): return self.mul(a, a)
    def eq(self, a, b): return all((x - y) % self.p == 0 for x, y in zip(a, b))
    def is_zero(self, a): return all(x % self.p == 0 for x in a)

    def pow(self, a, e):
        res = self.one
        base = a
        while e:
            if e & 1:
                res = self.mul(res, base)
            base = self.mul(base, base)
            e >>= 1
        return res

    def inv(self, a):
        # extended Euclid on polynomials over Fq (degree-12 modulus)
        p = self.p
        mod = [82, 0, 0, 0, 0, 0, (-18) % p, 0, 0, 0, 0, 0, 1]
        lm, hm = [1] + [0] * 12, [0] * 13
        low, high = list(a) + [0], mod[:]

        def deg(poly):
            d = len(poly) - 1
            while d and poly[d] % p == 0:
                d -= 1
            return d

        while deg(low):
            # r = high // low (polynomial rounded division)
            dl, dh = deg(low), deg(high)
            temp = high[:]
            quo = [0] * 13
            inv_lead = pow(low[dl], -1, p)
            for i in range(dh - dl, -1, -1):
                quo[i] = temp[dl + i] * inv_lead % p
                for c in range(dl + 1):
                    temp[c + i] = (temp[c + i] - low[c] * quo[i]) % p
            nm, new = hm[:], high[:]
            for i in range(13):
                for j in range(13 - i):
                    nm[i + j] = (nm[i + j] - lm[i] * quo[j]) % p
                    new[i + j] = (new[i + j] - low[i] * quo[j]) % p
            lm, low, hm, high = nm, new, lm, low
        inv0 = pow(low[0], -1, p)
        return tuple(x * inv0 % p for x in lm[:12])


FQ = FqOps(Q)
FR = FqOps(R)
FQ2 = Fq2Ops(Q)
FQ12 = Fq12Ops(Q)

B1 = 3
# twist coefficient b' = 3/(9+u)
B2 = FQ2.mul((3, 0), FQ2.inv((9, 1)))
B12 = (3,) + (0,) * 11


# ----------------------------------------------------------------------------- elliptic curve (affine, None = infinity)
def ec_is_on_curve(P, F, b):
    if P is None:
        return True
    x, y = P
    return F.eq(F.sqr(y), F.add(F.mul(F.sqr(x), x), b))


def ec_neg(P, F):
    if P is None:
        return None
    return (P[0], F.neg(P[1]))


def ec_double(P, F):
    if P is None:
        return None
    x, y = P
    if F.is_zero(y):
        return None
    m = F.mul(F.muli(F.sqr(x), 3), F.inv(F.muli(y, 2)))
    nx = F.sub(F.sqr(m), F.muli(x, 2))
    ny = F.sub(F.mul(m, F.sub(x, nx)), y)
    return (nx, ny)


def ec_add(P, S, F):
    if P is None:
        return S
    if S is None:
        return P
    x1, y1 = P
    x2, y2 = S
    if F.eq(x1, x2):
        if F.eq(y1, y2):
            return ec_double(P, F)
        return None
    m = F.mul(F.sub(y2, y1), F.inv(F.sub(x2, x1)))
    nx = F.sub(F.sub(F.sqr(m), x1), x2)
    ny = F.sub(F.mul(m, F.sub(x1, nx)), y1)
    return (nx, ny)


# Jacobian variants (much faster for scalar multiplication: no inversion per step)
def jac_from_affine(P, F):
    if P is None:
        return (F.one, F.one, F.zero)
    return (P[0], P[1], F.one)


def jac_to_affine(P, F):
    X, Y, Z = P
    if F.is_zero(Z):
        return None
    zi = F.inv(Z)
    zi2 = F.sqr(zi)
    return (F.mul(X, zi2), F.mul(Y, F.mul(zi2, zi)))


def jac_double(P, F):
    X, Y, Z = P
    if F.is_zero(Z) or F.is_zero(Y):
        return (F.one, F.one, F.zero)
    A = F.sqr(X)
    Bv = F.sqr(Y)
    C = F.sqr(Bv)
    D = F.muli(F.sub(F.sub(F.sqr(F.add(X, Bv)), A), C), 2)
    E = F.muli(A, 3)
    Fv = F.sqr(E)
    X3 = F.sub(Fv, F.muli(D, 2))
    Y3 = F.sub(F.mul(E, F.sub(D, X3)), F.muli(C, 8))
    Z3 = F.muli(F.mul(Y, Z), 2)
    return (X3, Y3, Z3)


def jac_add(P, S, F):
    X1, Y1, Z1 = P
    X2, Y2, Z2 = S
    if F.is_zero(Z1):
        return S
    if F.is_zero(Z2):
        return P
    Z1Z1 = F.sqr(Z1)
    Z2Z2 = F.sqr(Z2)
    U1 = F.mul(X1, Z2Z2)
    U2 = F.mul(X2, Z1Z1)
    S1 = F.mul(Y1, F.mul(Z2, Z2Z2))
    S2 = F.mul(Y2, F.mul(Z1, Z1Z1))
    if F.eq(U1, U2):
        if F.eq(S1, S2):
            return jac_double(P, F)
        return (F.one, F.one, F.zero)
    H = F.sub(U2, U1)
    Rr = F.sub(S2, S1)
    HH = F.sqr(H)
    HHH = F.mul(H, HH)
    V = F.mul(U1, HH)
    X3 = F.sub(F.sub(F.sqr(Rr), HHH), F.muli(V, 2))
    Y3 = F.sub(F.mul(Rr, F.sub(V, X3)), F.mul(S1, HHH))
    Z3 = F.mul(F.mul(Z1, Z2), H)
    return (X3, Y3, Z3)


def ec_mul(P, k, F, order=R):
    """k*P for affine P (None = infinity), double-and-add in Jacobian coordinates."""
    k %= order
    if P is None or k == 0:
        return None
    acc = (F.one, F.one, F.zero)
    base = jac_from_affine(P, F)
    for bit in bin(k)[2:]:
        acc = jac_double(acc, F)
        if bit == "1":
            acc = jac_add(acc, base, F)
    return jac_to_affine(acc, F)


def msm_naive(points, scalars, F):
    """sum_i scalars[i] * points[i]; affine in, affine out. Definitional (no windows)."""
    acc = (F.one, F.one, F.zero)
    for P, k in zip(points, scalars):
        if P is None or k % R == 0:
            continue
        acc = jac_add(acc, jac_from_affine(ec_mul(P, k, F), F), F)
    return jac_to_affine(acc, F)


def g1_mul(P, k): return ec_mul(P, k, FQ)
def g2_mul(P, k): return ec_mul(P, k, FQ2)
def g1_add(P, S): return ec_add(P, S, FQ)
def g2_add(P, S): return ec_add(P, S, FQ2)
def g1_is_on_curve(P): return ec_is_on_curve(P, FQ, B1)
def g2_is_on_curve(P): return ec_is_on_curve(P, FQ2, B2)


# ----------------------------------------------------------------------------- pairing (optimal ate, textbook form)
def _fq2_to_fq12(a):
    # a0 + a1*u with u = t^6 - 9
    c = [0] * 12
    c[0] = (a[0] - 9 * a[1]) % Q
    c[6] = a[1] % Q
    return tuple(c)


_W = (0, 1) + (0,) * 10
_W2 = FQ12.mul(_W, _W)
_W3 = FQ12.mul(_W2, _W)


def twist(P):
    """G2 point over Fq2 -> point on y^2 = x^3 + 3 over Fq12."""
    if P is None:
        return None
    return (FQ12.mul(_fq2_to_fq12(P[0]), _W2), FQ12.mul(_fq2_to_fq12(P[1]), _W3))


def cast_g1_to_fq12(P):
    if P is None:
        return None
    return ((P[0] % Q,) + (0,) * 11, (P[1] % Q,) + (0,) * 11)


def _linefunc(P1, P2, T):
    F = FQ12
    x1, y1 = P1
    x2, y2 = P2
    xt, yt = T
    if not F.eq(x1, x2):
        m = F.mul(F.sub(y2, y1), F.inv(F.sub(x2, x1)))
        return F.sub(F.mul(m, F.sub(xt, x1)), F.sub(yt, y1))
    if F.eq(y1, y2):
        m = F.mul(F.muli(F.sqr(x1), 3), F.inv(F.muli(y1, 2)))
        return F.sub(F.mul(m, F.sub(xt, x1)), F.sub(yt, y1))
    return F.sub(xt, x1)


def miller_loop(Q2, P1):
    """Miller loop f_{6x+2,Q}(P) * two Frobenius lines; Q2 in G2 (Fq2 affine), P1 in G1 (affine).
    No final exponentiation."""
    if Q2 is None or P1 is None:
        return FQ12.one
    F = FQ12
    Qt = twist(Q2)
    Pt = cast_g1_to_fq12(P1)
    Rp = Qt
    f = F.one
    for i in range(LOG_ATE_LOOP_COUNT, -1, -1):
        f = F.mul(F.sqr(f), _linefunc(Rp, Rp, Pt))
        Rp = ec_double(Rp, F)
        if ATE_LOOP_COUNT & (1 << i):
            f = F.mul(f, _linefunc(Rp, Qt, Pt))
            Rp = ec_add(Rp, Qt, F)
    Q1 = (F.pow(Qt[0], Q), F.pow(Qt[1], Q))
    nQ2 = (F.pow(Q1[0], Q), F.neg(F.pow(Q1[1], Q)))
    f = F.mul(f, _linefunc(Rp, Q1, Pt))
    Rp = ec_add(Rp, Q1, F)
    f = F.mul(f, _linefunc(Rp, nQ2, Pt))
    return f


FINAL_EXP = (Q ** 12 - 1) // R


def final_exponentiation(f):
    return FQ12.pow(f, FINAL_EXP)


def pairing(Q2, P1):
    """e(P1, Q2) = miller(Q2, P1)^((q^12-1)/r)  (plain final exponentiation)."""
    return final_exponentiation(miller_loop(Q2, P1))


def pairing_product_is_one(pairs):
    """prod e(P_i, Q_i) == 1 with one shared final exponentiation; pairs = [(P1, Q2), ...]."""
    f = FQ12.one
    for P1, Q2 in pairs:
        f = FQ12.mul(f, miller_loop(Q2, P1))
    return FQ12.eq(final_exponentiation(f), FQ12.one)


# ----------------------------------------------------------------------------- Montgomery helpers (zkey encoding)
def to_mont(x, p): return x * MONT_R % p
def from_mont(x, p): return x * pow(MONT_R, -1, p) % p


def mont_mul(a, b, p):
    """Montgomery product a*b/R mod p -- what Fr.mul does on raw buffers in ffjavascript."""
    return a * b * pow(MONT_R, -1, p) % p


# ----------------------------------------------------------------------------- NTT over Fr (definitional + radix-2)
def ntt_naive(a, inverse=False):
    n = len(a)
    k = n.bit_length() - 1
    assert 1 << k == n
    w = fr_root_of_unity(k)
    if inverse:
        w = pow(w, -1, R)
    out = []
    for i in range(n):
        wi = pow(w, i, R)
        acc, x = 0, 1
        for j in range(n):
            acc = (acc + a[j] * x) % R
            x = x * wi % R
        out.append(acc)
    if inverse:
        ninv = pow(n, -1, R)
        out = [v * ninv % R for v in out]
    return out


def ntt(a, inverse=False):
    """Iterative radix-2 NTT, natural order in and out (Fr.fft / Fr.ifft semantics)."""
    n = len(a)
    k = n.bit_length() - 1
    assert 1 << k == n
    a = list(a)
    j = 0
    for i in range(1, n):
        bit = n >> 1
        while j & bit:
            j ^= bit
            bit >>= 1
        j ^= bit
        if i < j:
            a[i], a[j] = a[j], a[i]
    w_n = fr_root_of_unity(k)
    if inverse:
        w_n = pow(w_n, -1, R)
    length = 2
    while length <= n:
        wl = pow(w_n, n // length, R)
        for start in range(0, n, length):
            w = 1
            half = length >> 1
            for t in range(half):
                u = a[start + t]
                v = a[start + t + half] * w % R
                a[start + t] = (u + v) % R
                a[start + t + half] = (u - v) % R
                w = w * wl % R
        length <<= 1
    if inverse:
        ninv = pow(n, -1, R)
        a = [v * ninv % R for v in a]
    return a
