"""ORACLE (test infrastructure) -- ctypes loader for oracle/_build/liboracle.so (oracle/c/zkpoa_oracle.c).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this."""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle.so")
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            subprocess.check_call(["make", "-C", os.path.join(_HERE, "c")])
        L = ctypes.CDLL(_SO)
        vp, u64, ci = ctypes.c_void_p, ctypes.c_uint64, ctypes.c_int
        L.orc_msm_g1.argtypes = [vp, vp, u64, vp, ci]
        L.orc_msm_g2.argtypes = [vp, vp, u64, vp, ci]
        L.orc_fixed_base_g1.argtypes = [vp, u64, vp, ci]
        L.orc_fixed_base_g2.argtypes = [vp, u64, vp, ci]
        L.orc_ntt.argtypes = [vp, ctypes.c_uint, ci]
        L.orc_h_scalars.argtypes = [vp, u64, vp, u64, ctypes.c_uint, vp]
        L.orc_quotient_check.argtypes = [vp, u64, vp, u64, ctypes.c_uint, vp, ctypes.c_char_p, ci]
        L.orc_poseidon2.argtypes = [vp, vp, u64, vp, ci]
        L.orc_merkle_levels.argtypes = [vp, vp, u64, ctypes.c_uint, vp, ci]
        L.orc_prove.argtypes = [vp, u64, vp, u64, ctypes.c_char_p, ctypes.c_char_p, vp, vp, ci]
        L.orc_field_op.argtypes = [ci, ci, vp, vp, vp, u64]
        L.orc_group_add.argtypes = [ci, vp, vp, vp, u64]
        _lib = L
    return _lib


def _in(b):
    return ctypes.cast(ctypes.create_string_buffer(bytes(b), len(b)), ctypes.c_void_p) if not isinstance(b, ctypes.Array) else b


def msm_windows(n):
    """Number of Pippenger windows the C oracle uses for n points (= its maximum useful thread count)."""
    best, bc = 1, None
    for c in range(1, 17):
        cost = ((254 + c - 1) // c) * (n + 2.0 * (1 << c))
        if bc is None or cost < bc:
            best, bc = c, cost
    return (254 + best - 1) // best


def msm_g1(bases, scalars, n=None, nthreads=1):
    n = len(scalars) // 32 if n is None else n
    out = ctypes.create_string_buffer(64)
    lib().orc_msm_g1(bytes(bases), bytes(scalars), n, out, nthreads)
    return out.raw


def msm_g2(bases, scalars, n=None, nthreads=1):
    n = len(scalars) // 32 if n is None else n
    out = ctypes.create_string_buffer(128)
    lib().orc_msm_g2(bytes(bases), bytes(scalars), n, out, nthreads)
    return out.raw


def fixed_base_g1(scalars, nthreads=1):
    n = len(scalars) // 32
    out = ctypes.create_string_buffer(max(1, 64 * n))
    lib().orc_fixed_base_g1(bytes(scalars), n, out, nthreads)
    return out.raw[:64 * n]


def fixed_base_g2(scalars, nthreads=1):
    n = len(scalars) // 32
    out = ctypes.create_string_buffer(max(1, 128 * n))
    lib().orc_fixed_base_g2(bytes(scalars), n, out, nthreads)
    return out.raw[:128 * n]


def ntt(data, k, inverse=False):
    buf = ctypes.create_string_buffer(bytes(data), 32 << k)
    lib().orc_ntt(buf, k, 1 if inverse else 0)
    return buf.raw


def h_scalars(coeffs_section, witness, n_vars, k):
    out = ctypes.create_string_buffer(32 << k)
    rc = lib().orc_h_scalars(bytes(coeffs_section), len(coeffs_section), bytes(witness), n_vars, k, out)
    if rc:
        raise ValueError("orc_h_scalars rc=%d" % rc)
    return out.raw


def _ptr(b):
    """bytes or a C-contiguous numpy array -> (address, keepalive), without copying."""
    if isinstance(b, (bytes, bytearray)):
        if isinstance(b, bytes):
            return ctypes.cast(ctypes.c_char_p(b), ctypes.c_void_p), b
        arr = (ctypes.c_char * len(b)).from_buffer(b)
        return ctypes.cast(arr, ctypes.c_void_p), arr
    return ctypes.c_void_p(b.ctypes.data), b          # numpy


def quotient_check(coeffs_section, witness, n_vars, k, h_scalars, z, nthreads=1):
    """True iff the 2^k standard-form H scalars satisfy A(z) B(z) - C(z) == Hq(z) (z^n - 1) at the point z
    (A, B, C from buildABC1 over coeffs_section + witness): a Schwartz-Zippel check of the COMPLETE vector with
    no NTT (oracle/c/zkpoa_oracle.c: orc_quotient_check). Buffers: bytes or numpy arrays (not copied)."""
    pc, kc = _ptr(coeffs_section)
    pw, kw = _ptr(witness)
    ph, kh = _ptr(h_scalars)
    size = len(coeffs_section) if isinstance(coeffs_section, (bytes, bytearray)) else coeffs_section.nbytes
    rc = lib().orc_quotient_check(pc, size, pw, n_vars, k, ph, int(z).to_bytes(32, "little"), nthreads)
    if rc not in (0, 1):
        raise ValueError("orc_quotient_check rc=%d" % rc)
    return rc == 0


def poseidon2(left, right, nthreads=1):
    """n independent circomlib Poseidon(2) hashes; left, right: n x 32 B LE standard form -> n x 32 B."""
    n = len(left) // 32
    out = ctypes.create_string_buffer(max(1, 32 * n))
    lib().orc_poseidon2(bytes(left), bytes(right), n, out, nthreads)
    return out.raw[:32 * n]


def merkle_levels(addresses, balances, log_leaves, nthreads=1):
    """All levels of the anonymity-set tree (leaves first, root last) as one byte string of (2^(k+1) - 1) x 32 B."""
    n = len(addresses) // 32
    out = ctypes.create_string_buffer(32 * ((2 << log_leaves) - 1))
    rc = lib().orc_merkle_levels(bytes(addresses), bytes(balances), n, log_leaves, out, nthreads)
    if rc:
        raise ValueError("orc_merkle_levels rc=%d" % rc)
    return out.raw


def prove(zkey, wtns, r=0, s=0, nthreads=1, n_public=None):
    """-> (proof_points[256], public bytes) ; raises on error (rc 3 = witness length mismatch)."""
    pts = ctypes.create_string_buffer(256)
    pub = ctypes.create_string_buffer(1 << 16)
    rc = lib().orc_prove(bytes(zkey), len(zkey), bytes(wtns), len(wtns), int(r).to_bytes(32, "little"),
                         int(s).to_bytes(32, "little"), pts, pub, nthreads)
    if rc:
        raise ValueError("orc_prove rc=%d" % rc)
    if n_public is None:
        import struct
        # header: section 2 of the zkey; locate nPublic
        from .py import groth16 as g16
        secs = g16.read_binfile(zkey, "zkey", 2)
        p, _ = secs[2][0]
        n_public = struct.unpack_from("<I", zkey, p + 76)[0]
    return pts.raw, pub.raw[:32 * n_public]


def field_op(field, op, a, b=None):
    n = len(a) // 32
    out = ctypes.create_string_buffer(max(1, 32 * n))
    lib().orc_field_op(field, op, bytes(a), bytes(b) if b is not None else bytes(a), out, n)
    return out.raw[:32 * n]


def group_add(group, a, b):
    size = 64 if group == 1 else 128
    n = len(a) // size
    out = ctypes.create_string_buffer(max(1, size * n))
    lib().orc_group_add(group, bytes(a), bytes(b), out, n)
    return out.raw[:size * n]
