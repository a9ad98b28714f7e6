"""Big-int model of the three local stages of the split H-scalar chain (include/zkpoa_prover.h,
zkpoa_split_stage1/2/3), written against the Python oracle's NTT. Test infrastructure: the CPU gloo test runs
the product's orchestration (zkpoa_amd.sharding.split_h_chain) over these stages, and the result must equal
the oracle's unsplit chain (oracle/py/groth16.py::h_scalars, after snarkjs groth16_prove.js steps 2-4)."""
from oracle.py import bn254 as bn
from oracle.py import groth16 as g16

R = bn.R


def _rev(x, bits):
    r = 0
    for i in range(bits):
        r |= ((x >> i) & 1) << (bits - 1 - i)
    return r


def _dif(x, inverse):
    """natural -> bit-reversed, unscaled (what the engine's DIF half produces)"""
    n = len(x)
    L = n.bit_length() - 1
    y = bn.ntt(x, inverse=inverse)
    if inverse:
        y = [v * n % R for v in y]
    return [y[_rev(p, L)] for p in range(n)]


def _dit(x):
    """bit-reversed -> natural, forward root"""
    n = len(x)
    L = n.bit_length() - 1
    return bn.ntt([x[_rev(j, L)] for j in range(n)])


def to_buf(polys, world):
    """[3 polynomials][M values] -> exchange-buffer layout [world][3][Q * 32 bytes] (include/zkpoa_prover.h)"""
    import torch
    flat = torch.frombuffer(bytearray(b"".join(int(v).to_bytes(32, "little") for p in polys for v in p)),
                            dtype=torch.uint8).view(len(polys), world, -1)
    return flat.permute(1, 0, 2).contiguous()


def from_buf(t):
    """exchange-buffer layout [world][3][Q * 32] -> [3][M values] (blocks ordered by rank)"""
    t = t.permute(1, 0, 2).contiguous()
    raw = bytes(t.view(-1).numpy().tobytes())
    per = t.shape[1] * t.shape[2]
    return [[int.from_bytes(raw[x * per + 32 * i:x * per + 32 * i + 32], "little") for i in range(per // 32)]
            for x in range(t.shape[0])]


class SplitRank:
    def __init__(self, zkey, witness, rank, world):
        self.G, self.g = world, rank
        self.n = zkey.domainSize
        self.k = self.n.bit_length() - 1
        self.M = self.n // world
        A, B, C = g16.build_abc(zkey, witness)
        self.rows = [[X[rank + world * t] for t in range(self.M)] for X in (A, B, C)]   # rows c = rank (mod G)
        self.h = None

    def stage1(self, out):
        out.copy_(to_buf([_dif(x, True) for x in self.rows], self.G))

    def stage2(self, recv, out):
        G, M, n, k = self.G, self.M, self.n, self.k
        Q, L = M // G, M.bit_length() - 1
        w = bn.fr_root_of_unity(k)
        winv = pow(w, -1, R)
        inc = g16.coset_inc(k)
        ninv = pow(n, -1, R)
        res = []
        for poly in from_buf(recv):
            o = [0] * M
            for pl in range(Q):
                k1 = _rev(self.g * Q + pl, L)
                v = [poly[g * Q + pl] * pow(winv, g * k1, R) % R for g in range(G)]
                a = [sum(v[g] * pow(winv, M * g * k2, R) for g in range(G)) % R for k2 in range(G)]
                a = [a[k2] * pow(inc, k1 + M * k2, R) % R * ninv % R for k2 in range(G)]
                for i2 in range(G):
                    b = sum(a[k2] * pow(w, M * k2 * i2, R) for k2 in range(G)) % R
                    o[i2 * Q + pl] = b * pow(w, k1 * i2, R) % R
            res.append(o)
        out.copy_(to_buf(res, self.G))

    def stage3(self, recv):
        Ao, Bo, Co = [_dit(p) for p in from_buf(recv)]
        self.h = [bn.from_mont((bn.mont_mul(a, b, R) - c) % R, R) for a, b, c in zip(Ao, Bo, Co)]
