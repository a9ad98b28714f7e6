"""Synthetic Groth16 workloads generated directly in HBM.

The reference's circuits cannot be compiled offline (no circom; submodules empty), so the
measurement configs use a circuit of the same *shape* (BASELINE.json configs[2..4]): n_vars wires,
domain 2^k, constraints (w[a] + w[b]) * w[d] = ..., 3 coefficients per constraint plus the
nPublic+1 public rows -- with every base point a KNOWN multiple of the generator:

    section X, index i:   P_i = (a_X + i * b_X) * G        (B1 and B2 share a_B, b_B)
    header:               alpha1 = al*G1, beta1 = be*G1, beta2 = be*G2, delta1 = de*G1, delta2 = de*G2

so the expected proof is a discrete-log computation in Fr (O(n) integer work, no CPU MSM):
    a = sum w_i A_i + al + r de,  b = sum w_i B_i + be + s de,
    c = sum w_i C_i + sum P_j H_j + s a + r b - r s de;     proof = (a*G1, b*G2, c*G1).
As in a real zkey, a wire that does not occur in a matrix has the point at infinity (all-zero bytes) in that
matrix's query: sections 5 (A) and 6 / 7 (B) are zeroed for the wires the random R1CS never uses there, and
the expectation sums only over the wires that are present.
Such a key is not a valid trusted setup (proofs do not verify); it exercises exactly the same
kernels on the same data volumes, which is what the timing configs need, and every group element of
the output is still checked exactly.
"""
import random

R_MOD = 21888242871839275222246405745257275088548364400416034343698204186575808495617
Q_MOD = 21888242871839275222246405745257275088696311157297823662689037894645226208583


def dlog_sums(limbs, i0=0):
    """(sum k_i, sum (i0 + i) * k_i) as exact Python ints; limbs = uint64 array [n, 4] (LE limbs)."""
    import numpy as np
    n = limbs.shape[0]
    s0 = 0
    s1 = 0
    chunk = 1 << 20
    for start in range(0, n, chunk):
        blk = limbs[start:start + chunk]
        idx = np.arange(i0 + start, i0 + start + blk.shape[0], dtype=np.uint64)
        for j in range(4):
            col = blk[:, j]
            for piece in range(4):
                part = (col >> np.uint64(16 * piece)) & np.uint64(0xFFFF)
                sh = 64 * j + 16 * piece
                s0 += int(part.sum(dtype=np.uint64)) << sh
                # idx < 2^28, part < 2^16, <= 2^20 terms: sum < 2^64
                s1 += int(np.dot(idx, part)) << sh
    return s0, s1


def _mont_g1_generator():
    one = (1 << 256) % Q_MOD
    two = (2 << 256) % Q_MOD
    return one.to_bytes(32, "little") + two.to_bytes(32, "little")


_G2_STD = (
    10857046999023057135944570762232829481370756359578518086990519993285655852781,
    11559732032986387107991004021392285783925812861821192530917403151452391805634,
    8495653923123431417604973247489272438418190587263600148770280649306958101930,
    4082367875863433681332203403145435568316851327593401208105741076214120093531,
)


def _mont_g2_generator():
    return b"".join(((v << 256) % Q_MOD).to_bytes(32, "little") for v in _G2_STD)


class SyntheticCircuit:
    """Device-resident synthetic proving key + witness with known discrete logs."""

    def __init__(self, zk, ctx, log_domain, n_vars, n_public=1, seed=0x5EED0010, witness_like=False, device=None,
                 shard=None):
        """shard = (rank, world, split[, block_log]): generate and keep only that rank's part of the key -- its index
        ranges of the point sections (sharding.shard_range) or, with block_log = L > 0, its blocks of 2^L items dealt
        round-robin (sharding.block_cyclic_indices), with `split` the cyclic H shard and the coefficient records of its
        own constraints -- and load it with zkpoa_zkey_load_device_shard. Same key as the unsharded circuit of the
        same seed, so the N partial results add up to its proof. The witness is whole on every rank."""
        import numpy as np
        import torch
        from .sharding import block_cyclic_blocks, shard_range
        self.zk, self.ctx = zk, ctx
        self.k, self.n, self.m, self.n_public = log_domain, 1 << log_domain, n_vars, n_public
        dev = device if device is not None else torch.device("cuda", torch.cuda.current_device())
        rng = random.Random(seed)
        self.par = {x: (rng.randrange(R_MOD), rng.randrange(R_MOD)) for x in ("A", "B", "C", "H")}
        self.hdr = {x: rng.randrange(1, 1 << 64) for x in ("alpha", "beta", "delta")}
        m, n = self.m, self.n
        nC = m - n_public - 1
        rank, world, split = (tuple(shard) + (0,))[:3] if shard is not None else (0, 1, False)
        block_log = shard[3] if shard is not None and len(shard) > 3 and world > 1 else 0
        self.shard = (rank, world, bool(split)) if shard is not None else None
        self.block_log = block_log
        (wlo, whi), (clo, chi), (hlo, hhi) = (shard_range(x, rank, world) for x in (m, nC, n))
        # the pieces of sections 5-8 this rank holds, as (global start, count) runs: one range, or its blocks
        wruns = block_cyclic_blocks(m, rank, world, block_log) if block_log else [(wlo, whi - wlo)]
        cruns = block_cyclic_blocks(nC, rank, world, block_log) if block_log else [(clo, chi - clo)]
        wcnt, ccnt, hcnt = sum(c for _, c in wruns), sum(c for _, c in cruns), hhi - hlo
        self.d_A = torch.empty(max(wcnt, 1) * 64, dtype=torch.uint8, device=dev)
        self.d_B1 = torch.empty(max(wcnt, 1) * 64, dtype=torch.uint8, device=dev)
        self.d_B2 = torch.empty(max(wcnt, 1) * 128, dtype=torch.uint8, device=dev)
        self.d_C = torch.empty(max(ccnt, 1) * 64, dtype=torch.uint8, device=dev)
        at = 0
        for start, cnt in wruns:
            ctx.gen_bases_g1_device(*self.par["A"], start, cnt, self.d_A.data_ptr() + at * 64)
            ctx.gen_bases_g1_device(*self.par["B"], start, cnt, self.d_B1.data_ptr() + at * 64)
            ctx.gen_bases_g2_device(*self.par["B"], start, cnt, self.d_B2.data_ptr() + at * 128)
            at += cnt
        at = 0
        for start, cnt in cruns:
            ctx.gen_bases_g1_device(*self.par["C"], start, cnt, self.d_C.data_ptr() + at * 64)
            at += cnt
        if split:
            # H[t * world + rank] = (a + rank * b + t * (world * b)) G: the same generator with shifted parameters
            ha, hb = self.par["H"]
            hcnt = n // world
            self.d_H = torch.empty(hcnt * 64, dtype=torch.uint8, device=dev)
            ctx.gen_bases_g1_device((ha + rank * hb) % R_MOD, (world * hb) % R_MOD, 0, hcnt, self.d_H.data_ptr())
        else:
            self.d_H = torch.empty(max(hcnt, 1) * 64, dtype=torch.uint8, device=dev)
            ctx.gen_bases_g1_device(*self.par["H"], hlo, hcnt, self.d_H.data_ptr())
        # coefficient records: constraint c has A-terms (a_c, 1), (b_c, 1) and B-term (d_c, 1); then public rows
        n_cons = n - n_public - 1
        g = torch.Generator(device="cpu")
        g.manual_seed(seed & 0x7FFFFFFF)
        sig = torch.randint(0, m, (n_cons, 3), generator=g, dtype=torch.int32)
        cidx = torch.arange(n_cons, dtype=torch.int32)
        r2 = (1 << 512) % R_MOD                           # coefficient 1 is stored as 1 * R^2
        val = torch.tensor([(r2 >> (32 * i)) & 0xFFFFFFFF for i in range(8)], dtype=torch.int64).to(torch.int32)
        n_coef = 3 * n_cons + n_public + 1
        # host image of zkey section 4 (u32 count + 44-byte records) in one buffer; `recs` is a view of the records.
        # A split shard other than rank 0 builds only the records of its own constraints c = rank (mod world): the full
        # image (8.9 GB at 2^26) is needed on the device by nobody and on the host only by rank 0 (bench.py's checks).
        own_only = bool(split) and rank != 0
        if own_only:
            cons = torch.arange(rank, n_cons, world, dtype=torch.int64)
            pubs = torch.arange(n_cons, n_cons + n_public + 1, dtype=torch.int64)
            pubs = pubs[(pubs % world) == rank]
            n_host = 3 * cons.numel() + pubs.numel()
        else:
            cons = torch.arange(n_cons, dtype=torch.int64)
            pubs = torch.arange(n_cons, n_cons + n_public + 1, dtype=torch.int64)
            n_host = n_coef
        self._sec4 = np.empty(4 + n_host * 44, dtype=np.uint8)
        self._sec4[:4] = np.frombuffer(int(n_host).to_bytes(4, "little"), dtype=np.uint8)
        recs = torch.from_numpy(self._sec4[4:].view(np.int32).reshape(n_host, 11))
        recs[:, 3:] = val
        nc = cons.numel()
        for t, mat in enumerate((0, 0, 1)):
            blk = recs[t * nc:(t + 1) * nc]
            blk[:, 0] = mat
            blk[:, 1] = cons.to(torch.int32)
            blk[:, 2] = sig[cons, t]
        pub = recs[3 * nc:]
        pub[:, 0] = 0
        pub[:, 1] = pubs.to(torch.int32)
        pub[:, 2] = (pubs - n_cons).to(torch.int32)
        self.sec4_complete = not own_only
        # wires present in A (both A-terms and the public rows) and in B: the others get the point at infinity
        in_a = torch.zeros(m, dtype=torch.bool)
        in_b = torch.zeros(m, dtype=torch.bool)
        in_a[sig[:, 0].long()] = True
        in_a[sig[:, 1].long()] = True
        in_a[:n_public + 1] = True
        in_b[sig[:, 2].long()] = True
        self.in_a, self.in_b = in_a.numpy(), in_b.numpy()
        if wcnt:
            mine_a = torch.cat([in_a[s0:s0 + c] for s0, c in wruns])      # presence flags in the order of this rank's points
            mine_b = torch.cat([in_b[s0:s0 + c] for s0, c in wruns])
            self.d_A[:wcnt * 64].view(wcnt, 64)[(~mine_a).to(dev)] = 0
            self.d_B1[:wcnt * 64].view(wcnt, 64)[(~mine_b).to(dev)] = 0
            self.d_B2[:wcnt * 128].view(wcnt, 128)[(~mine_b).to(dev)] = 0
        self.n_coef = n_coef
        self.recs_host = recs                              # [n_coef, 11] int32 == 44-byte records
        if split and not own_only:                         # only the records of this rank's constraints go to HBM
            mine = recs[(recs[:, 1] % world) == rank].contiguous()
            self.d_recs, n_dev = mine.to(dev), mine.shape[0]
            del mine
        else:
            self.d_recs, n_dev = recs.to(dev), n_host
        # witness: w[0] = 1, rest uniform 252-bit (or witness-like: 55% bits, 35% < 2^64, 10% uniform)
        nr = np.random.default_rng(seed + 1)
        limbs = nr.integers(0, 1 << 63, size=(m, 4), dtype=np.uint64) * 2 + nr.integers(0, 2, size=(m, 4), dtype=np.uint64)
        limbs[:, 3] &= np.uint64((1 << 60) - 1)
        if witness_like:
            u = nr.random(m)
            small = u < 0.55
            limbs[small, 1:] = 0
            limbs[small, 0] = nr.integers(0, 2, size=int(small.sum()), dtype=np.uint64)
            limbs[(u >= 0.55) & (u < 0.9), 1:] = 0
        limbs[0] = (1, 0, 0, 0)
        self.w_limbs = limbs
        self.d_witness = torch.from_numpy(limbs.view(np.uint8).reshape(-1).copy()).to(dev)
        G1, G2 = _mont_g1_generator(), _mont_g2_generator()
        hp = (zk.g1_mul(G1, self.hdr["alpha"]) + zk.g1_mul(G1, self.hdr["beta"]) + zk.g2_mul(G2, self.hdr["beta"]) +
              zk.g1_mul(G1, self.hdr["delta"]) + zk.g2_mul(G2, self.hdr["delta"]))
        if shard is None:
            self.key = ctx.load_zkey_device(m, n_public, log_domain, self.d_A.data_ptr(), self.d_B1.data_ptr(),
                                            self.d_B2.data_ptr(), self.d_C.data_ptr(), self.d_H.data_ptr(),
                                            self.d_recs.data_ptr(), n_coef, hp)
        else:
            self.key = ctx.load_zkey_device_shard(m, n_public, log_domain, rank, world, split, self.d_A.data_ptr(),
                                                  self.d_B1.data_ptr(), self.d_B2.data_ptr(), self.d_C.data_ptr(),
                                                  self.d_H.data_ptr(), self.d_recs.data_ptr(), n_dev, hp,
                                                  block_log=block_log)
        self.header_points = hp
        self._G1, self._G2 = G1, G2

    def coeff_section(self):
        """Payload of zkey section 4 (u32 count + records) as a numpy uint8 array (no copy)."""
        if not self.sec4_complete:
            raise RuntimeError("this rank's synthetic circuit holds only its own coefficient records")
        return self._sec4

    def coeff_section_bytes(self):
        """Same, as bytes (a copy), for the host-buffer entry points."""
        return self._sec4.tobytes()

    def h_scalars(self):
        """H-MSM scalars of the last prove on this key, read back from HBM: numpy uint64 [n, 4] (standard form)."""
        return self.ctx.read_h_scalars(self.key, self.n)

    def witness_bytes(self):
        return self.w_limbs.tobytes()

    @staticmethod
    def _binfile(magic, version, sections):
        """iden3 binfile container from (id, bytes-like) pairs, as one bytes object."""
        out = [magic, int(version).to_bytes(4, "little"), len(sections).to_bytes(4, "little")]
        for sid, payload in sections:
            out += [int(sid).to_bytes(4, "little"), len(payload).to_bytes(8, "little"), payload]
        return b"".join(out)

    def zkey_image(self):
        """The whole (unsharded) synthetic key as a .zkey file image: the sections are copied out of HBM, gamma2 is
        set to beta2 (the prover never reads it) and section 3 (IC) to points at infinity, so the image has no valid
        verification key -- it is for provers (this one through its file entry points, the C oracle), not verifiers."""
        if self.shard is not None:
            raise RuntimeError("zkey_image needs the unsharded circuit")
        m, npub, hp = self.m, self.n_public, self.header_points          # alpha1 beta1 beta2 delta1 delta2
        sec2 = (b"".join([(32).to_bytes(4, "little"), Q_MOD.to_bytes(32, "little"), (32).to_bytes(4, "little"),
                          R_MOD.to_bytes(32, "little"), m.to_bytes(4, "little"), npub.to_bytes(4, "little"),
                          self.n.to_bytes(4, "little")]) + hp[0:64] + hp[64:128] + hp[128:256] + hp[128:256] +
                hp[256:320] + hp[320:448])
        dev = lambda t, count, size: t[:count * size].cpu().numpy().tobytes()
        return self._binfile(b"zkey", 1, [
            (1, (1).to_bytes(4, "little")), (2, sec2), (3, bytes(64 * (npub + 1))), (4, self.coeff_section_bytes()),
            (5, dev(self.d_A, m, 64)), (6, dev(self.d_B1, m, 64)), (7, dev(self.d_B2, m, 128)),
            (8, dev(self.d_C, m - npub - 1, 64)), (9, dev(self.d_H, self.n, 64))])

    def wtns_image(self):
        """The witness as a .wtns file image."""
        return self._binfile(b"wtns", 2, [
            (1, (32).to_bytes(4, "little") + R_MOD.to_bytes(32, "little") + self.m.to_bytes(4, "little")),
            (2, self.witness_bytes())])

    def write_zkey(self, path, chunk_bytes=256 << 20):
        """zkey_image() written straight to `path`, the point sections streamed out of HBM in chunks: the layer-two /
        layer-three shapes are 13 / 30 GB files, which zkey_image() would hold three times over in host memory."""
        if self.shard is not None:
            raise RuntimeError("write_zkey needs the unsharded circuit")
        m, npub, hp = self.m, self.n_public, self.header_points
        sec2 = (b"".join([(32).to_bytes(4, "little"), Q_MOD.to_bytes(32, "little"), (32).to_bytes(4, "little"),
                          R_MOD.to_bytes(32, "little"), m.to_bytes(4, "little"), npub.to_bytes(4, "little"),
                          self.n.to_bytes(4, "little")]) + hp[0:64] + hp[64:128] + hp[128:256] + hp[128:256] +
                hp[256:320] + hp[320:448])
        small = [(1, (1).to_bytes(4, "little")), (2, sec2), (3, bytes(64 * (npub + 1)))]
        tensors = [(5, self.d_A, m * 64), (6, self.d_B1, m * 64), (7, self.d_B2, m * 128),
                   (8, self.d_C, (m - npub - 1) * 64), (9, self.d_H, self.n * 64)]
        with open(path, "wb") as f:
            f.write(b"zkey" + (1).to_bytes(4, "little") + (4 + len(tensors)).to_bytes(4, "little"))
            for sid, payload in small:
                f.write(int(sid).to_bytes(4, "little") + len(payload).to_bytes(8, "little") + payload)
            sec4 = self.coeff_section()
            f.write((4).to_bytes(4, "little") + int(sec4.nbytes).to_bytes(8, "little"))
            f.write(memoryview(sec4))
            for sid, t, nbytes in tensors:
                f.write(int(sid).to_bytes(4, "little") + int(nbytes).to_bytes(8, "little"))
                for off in range(0, nbytes, chunk_bytes):
                    f.write(memoryview(t[off:min(off + chunk_bytes, nbytes)].cpu().numpy()))
        return path

    def write_wtns(self, path):
        """wtns_image() written to `path` without an intermediate copy of the values."""
        with open(path, "wb") as f:
            hdr = (32).to_bytes(4, "little") + R_MOD.to_bytes(32, "little") + self.m.to_bytes(4, "little")
            f.write(b"wtns" + (2).to_bytes(4, "little") + (2).to_bytes(4, "little"))
            f.write((1).to_bytes(4, "little") + len(hdr).to_bytes(8, "little") + hdr)
            f.write((2).to_bytes(4, "little") + int(self.w_limbs.nbytes).to_bytes(8, "little"))
            f.write(memoryview(self.w_limbs).cast("B"))
        return path

    def prove(self, r=0, s=0):
        return self.ctx.prove_device(self.key, self.d_witness.data_ptr(), r, s)

    def expected_dlogs(self, r, s, h_scalars_bytes=None):
        """(a, b, c) with c = None when the H scalars are not supplied (bytes or a numpy uint64 [n, 4] array).
        The caller supplies H scalars it has validated independently of this package (the tests and bench.py do:
        a CPU restatement of the chain at test sizes, a polynomial-identity check at full size)."""
        import numpy as np
        sums = {}
        for x, present in (("A", self.in_a), ("B", self.in_b)):
            limbs = self.w_limbs.copy()
            limbs[~present] = 0                             # absent wire: point at infinity in that query
            s0, s1 = dlog_sums(limbs)
            sums[x] = (self.par[x][0] * s0 + self.par[x][1] * s1) % R_MOD
        de = self.hdr["delta"]
        a = (sums["A"] + self.hdr["alpha"] + r * de) % R_MOD
        b = (sums["B"] + self.hdr["beta"] + s * de) % R_MOD
        c = None
        if h_scalars_bytes is not None:
            wc = self.w_limbs[self.n_public + 1:]
            c0, c1 = dlog_sums(wc)
            csum = (self.par["C"][0] * c0 + self.par["C"][1] * c1) % R_MOD
            P = (h_scalars_bytes if isinstance(h_scalars_bytes, np.ndarray)
                 else np.frombuffer(h_scalars_bytes, dtype=np.uint64).reshape(-1, 4))
            h0, h1 = dlog_sums(P)
            hsum = (self.par["H"][0] * h0 + self.par["H"][1] * h1) % R_MOD
            c = (csum + hsum + s * a + r * b - r * s % R_MOD * de) % R_MOD
        return a, b, c

    def check(self, proof_points, r, s, h_scalars_bytes=None):
        """True iff pi_a, pi_b (and pi_c when H scalars are given) equal the known-dlog expectation."""
        a, b, c = self.expected_dlogs(r, s, h_scalars_bytes)
        ok = proof_points[0:64] == self.zk.g1_mul(self._G1, a) and proof_points[64:192] == self.zk.g2_mul(self._G2, b)
        if c is not None:
            ok = ok and proof_points[192:256] == self.zk.g1_mul(self._G1, c)
        return ok

    def close(self):
        self.key.close()
