"""zkpoa_amd -- Python host layer over libzkpoa_prover.so (the MI355X Groth16 prover).

Mirrors the reference's interface for the prove step: the reference calls an external prover
as `prover <zkey> <wtns> <proof.json> <public.json>` (scripts/g16_prove.sh:248-252) or
`snarkjs groth16 prove` with the same four arguments (scripts/g16_prove.sh:255-259); here
that is `groth16_prove(zkey_path, wtns_path, proof_path, public_path)`, plus the stages of
the path (`msm_g1`, `msm_g2`, `ntt`, `h_scalars`) as the C ABI exposes them.

This package never computes on the CPU: if the shared library is missing, or no HIP device
is usable, calls raise. (The directory name has a hyphen, so the package is loaded through
`__graft_entry__.load_package()` / tests/conftest.py under the module name `zkpoa_amd`.)
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libzkpoa_prover.so")
PROVER_BIN = os.path.join(_HERE, "prover")
MERKLE_BIN = os.path.join(_HERE, "merkle-tree")
SETUP_BIN = os.path.join(_HERE, "zkpoa-setup")

PROVER_OK = 0
PROVER_ERROR = 1
PROVER_ERROR_SHORT_BUFFER = 2
PROVER_INVALID_WITNESS_LENGTH = 3
PROVER_ERROR_RUNTIME = 4

# every symbol include/zkpoa_prover.h declares
EXPORTS = [
    "groth16_prover", "groth16_prover_zkey_file",
    "zkpoa_context_create", "zkpoa_context_destroy", "zkpoa_last_error",
    "zkpoa_zkey_load", "zkpoa_zkey_free", "zkpoa_zkey_info", "zkpoa_prove",
    "zkpoa_zkey_load_device", "zkpoa_zkey_load_device_shard", "zkpoa_prove_device", "zkpoa_setup_accumulate", "zkpoa_zkey_new", "zkpoa_zkey_contribute", "zkpoa_wtns_check",
    "zkpoa_groth16_prover_files", "zkpoa_set_thread_options", "zkpoa_clear_thread_options", "zkpoa_idle_work", "zkpoa_zkey_load_shard", "zkpoa_zkey_load_shard_ex", "zkpoa_zkey_set_shard", "zkpoa_zkey_header",
    "zkpoa_prove_partials", "zkpoa_prove_partials_device", "zkpoa_prove_assemble",
    "zkpoa_zkey_load_shard_split", "zkpoa_zkey_set_shard_split", "zkpoa_witness_load",
    "zkpoa_split_stage1", "zkpoa_split_stage2", "zkpoa_split_stage3",
    "zkpoa_proof_to_json", "zkpoa_public_to_json",
    "zkpoa_msm_g1", "zkpoa_msm_g2", "zkpoa_ntt", "zkpoa_h_scalars",
    "zkpoa_msm_g1_device", "zkpoa_msm_g2_device", "zkpoa_ntt_device",
    "zkpoa_msm_g1_device_lane", "zkpoa_last_ms_lane",
    "zkpoa_gen_bases_g1_device", "zkpoa_gen_bases_g2_device",
    "zkpoa_g1_sum", "zkpoa_g2_sum", "zkpoa_g1_mul", "zkpoa_g2_mul",
    "zkpoa_setup_defer_host_frees", "zkpoa_last_ms", "zkpoa_set_option", "zkpoa_msm_points_limit", "zkpoa_field_op", "zkpoa_group_add",
    "zkpoa_groth16_verify", "zkpoa_sanitize_proof", "zkpoa_groth16_verify_points", "zkpoa_zkey_vkey", "zkpoa_zkey_export_vkey",
    "zkpoa_zkey_read_h_scalars", "zkpoa_zkey_precompute",
    "zkpoa_context_stream", "zkpoa_context_synchronize",
    "zkpoa_poseidon_params", "zkpoa_poseidon2", "zkpoa_poseidon2_device", "zkpoa_merkle_build", "zkpoa_merkle_build_device", "zkpoa_merkle_free",
    "zkpoa_merkle_info", "zkpoa_merkle_root", "zkpoa_merkle_leaves", "zkpoa_merkle_path",
    "zkpoa_msm_table_build", "zkpoa_msm_table_free", "zkpoa_msm_table_info", "zkpoa_msm_table_run_lane",
]


class ZkpoaError(RuntimeError):
    pass


def shard_flags(split=False, block_log=0):
    """include/zkpoa_prover.h: ZKPOA_SHARD_SPLIT_CHAIN | ZKPOA_SHARD_BLOCK_CYCLIC(block_log)."""
    return (1 if split else 0) | ((int(block_log) & 0xff) << 8)


_lib = None


def lib():
    """The loaded shared library; raises loudly when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ZkpoaError("libzkpoa_prover.so not built (%s): run `python -c 'import __graft_entry__ as g; "
                             "g.build()'` -- there is no CPU fallback" % LIB_PATH)
        # PyTorch-ROCm wheels bundle their own libamdhip64 / libhsa-runtime64. If our library (linked
        # against /opt/rocm) is loaded first, torch's HSA copy later fails to open the device ("No HIP GPUs
        # are available"); loaded in this order both HIP runtimes share torch's HSA runtime (same SONAME).
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = ctypes.CDLL(LIB_PATH)
        c_void_pp = ctypes.POINTER(ctypes.c_void_p)
        ul_p = ctypes.POINTER(ctypes.c_ulong)
        L.zkpoa_context_create.argtypes = [ctypes.c_int, c_void_pp, ctypes.c_char_p, ctypes.c_ulong]
        L.zkpoa_context_destroy.argtypes = [ctypes.c_void_p]
        L.zkpoa_context_destroy.restype = None
        L.zkpoa_context_stream.argtypes = [ctypes.c_void_p, ctypes.c_int]
        L.zkpoa_context_stream.restype = ctypes.c_void_p
        L.zkpoa_context_synchronize.argtypes = [ctypes.c_void_p]
        for name in ("zkpoa_poseidon2", "zkpoa_poseidon2_device"):
            getattr(L, name).argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p]
        for name in ("zkpoa_merkle_build", "zkpoa_merkle_build_device"):
            getattr(L, name).argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, c_void_pp]
        L.zkpoa_merkle_free.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        L.zkpoa_merkle_free.restype = None
        L.zkpoa_merkle_info.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint64)]
        L.zkpoa_merkle_root.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        L.zkpoa_merkle_leaves.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_void_p]
        L.zkpoa_merkle_path.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_void_p]
        L.zkpoa_last_error.argtypes = [ctypes.c_void_p]
        L.zkpoa_last_error.restype = ctypes.c_char_p
        L.zkpoa_last_ms.argtypes = [ctypes.c_void_p, ctypes.c_int]
        L.zkpoa_last_ms.restype = ctypes.c_float
        L.zkpoa_set_option.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_long]
        L.zkpoa_msm_points_limit.argtypes = [ctypes.c_void_p]
        L.zkpoa_msm_points_limit.restype = ctypes.c_uint64
        for name in ("zkpoa_msm_g1", "zkpoa_msm_g2", "zkpoa_msm_g1_device", "zkpoa_msm_g2_device"):
            getattr(L, name).argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64,
                                         ctypes.c_void_p]
        L.zkpoa_msm_g1_device_lane.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                               ctypes.c_uint64, ctypes.c_void_p]
        L.zkpoa_last_ms_lane.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
        L.zkpoa_last_ms_lane.restype = ctypes.c_float
        L.zkpoa_ntt.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint, ctypes.c_int]
        L.zkpoa_ntt_device.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint, ctypes.c_int]
        L.zkpoa_h_scalars.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_ulong, ctypes.c_void_p,
                                      ctypes.c_uint64, ctypes.c_uint, ctypes.c_void_p]
        for name in ("zkpoa_gen_bases_g1_device", "zkpoa_gen_bases_g2_device"):
            getattr(L, name).argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_uint64,
                                         ctypes.c_uint64, ctypes.c_void_p]
        for name in ("zkpoa_g1_sum", "zkpoa_g2_sum"):
            getattr(L, name).argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p]
        for name in ("zkpoa_g1_mul", "zkpoa_g2_mul"):
            getattr(L, name).argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_void_p]
        L.zkpoa_field_op.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                     ctypes.c_void_p, ctypes.c_uint64]
        L.zkpoa_group_add.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                      ctypes.c_void_p, ctypes.c_uint64]
        L.zkpoa_zkey_load.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_ulong, c_void_pp]
        L.zkpoa_zkey_free.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        L.zkpoa_zkey_free.restype = None
        L.zkpoa_zkey_info.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint64)]
        L.zkpoa_prove.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_ulong,
                                  ctypes.c_char_p, ctypes.c_char_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_ulong]
        L.zkpoa_zkey_load_device.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint] + \
            [ctypes.c_void_p] * 6 + [ctypes.c_uint64, ctypes.c_char_p, c_void_pp]
        L.zkpoa_setup_accumulate.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_uint64,
                                             ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64,
                                             ctypes.c_uint64, ctypes.c_void_p]
        L.zkpoa_zkey_new.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p]
        L.zkpoa_wtns_check.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_char_p,
                                       ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64)]
        L.zkpoa_zkey_contribute.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p]
        L.zkpoa_zkey_load_device_shard.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint,
                                                   ctypes.c_uint64, ctypes.c_uint64, ctypes.c_int] + \
            [ctypes.c_void_p] * 6 + [ctypes.c_uint64, ctypes.c_char_p, c_void_pp]
        L.zkpoa_prove_device.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_char_p,
                                         ctypes.c_char_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_ulong]
        L.zkpoa_zkey_load_shard_ex.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_ulong, ctypes.c_uint64,
                                               ctypes.c_uint64, ctypes.c_int, c_void_pp]
        L.zkpoa_zkey_load_shard.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_ulong, ctypes.c_uint64,
                                            ctypes.c_uint64, c_void_pp]
        L.zkpoa_zkey_set_shard.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint64]
        L.zkpoa_zkey_load_shard_split.argtypes = L.zkpoa_zkey_load_shard.argtypes
        L.zkpoa_zkey_set_shard_split.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint64]
        L.zkpoa_witness_load.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_ulong,
                                         ctypes.c_void_p, ctypes.c_ulong]
        L.zkpoa_split_stage1.argtypes = [ctypes.c_void_p] * 4
        L.zkpoa_split_stage2.argtypes = [ctypes.c_void_p] * 4
        L.zkpoa_split_stage3.argtypes = [ctypes.c_void_p] * 3
        L.zkpoa_zkey_header.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        L.zkpoa_prove_partials.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_ulong,
                                           ctypes.c_void_p, ctypes.c_void_p, ctypes.c_ulong]
        L.zkpoa_prove_partials_device.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        L.zkpoa_prove_assemble.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p,
                                           ctypes.c_void_p]
        L.zkpoa_groth16_verify.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_void_p,
                                           ctypes.c_ulong]
        L.zkpoa_sanitize_proof.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_void_p, ul_p,
                                           ctypes.c_void_p, ctypes.c_ulong]
        L.zkpoa_groth16_verify_points.argtypes = [ctypes.c_void_p, ctypes.c_ulong, ctypes.c_void_p, ctypes.c_void_p,
                                                  ctypes.c_ulong, ctypes.c_void_p, ctypes.c_ulong]
        L.zkpoa_zkey_vkey.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ul_p]
        L.zkpoa_zkey_precompute.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64,
                                            ctypes.POINTER(ctypes.c_uint64)]
        L.zkpoa_msm_table_build.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_int,
                                            c_void_pp]
        L.zkpoa_msm_table_free.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        L.zkpoa_msm_table_free.restype = None
        L.zkpoa_msm_table_info.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint64)]
        L.zkpoa_msm_table_run_lane.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                               ctypes.c_void_p]
        L.zkpoa_zkey_read_h_scalars.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_ulong]
        L.zkpoa_proof_to_json.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ul_p]
        L.zkpoa_public_to_json.argtypes = [ctypes.c_void_p, ctypes.c_ulong, ctypes.c_int, ctypes.c_void_p, ul_p]
        L.groth16_prover.argtypes = [ctypes.c_void_p, ctypes.c_ulong, ctypes.c_void_p, ctypes.c_ulong,
                                     ctypes.c_void_p, ul_p, ctypes.c_void_p, ul_p, ctypes.c_void_p, ctypes.c_ulong]
        L.groth16_prover_zkey_file.argtypes = [ctypes.c_char_p, ctypes.c_void_p, ctypes.c_ulong,
                                               ctypes.c_void_p, ul_p, ctypes.c_void_p, ul_p, ctypes.c_void_p,
                                               ctypes.c_ulong]
        _lib = L
    return _lib


def _buf(b):
    """read-only bytes-like -> (ctypes pointer, keepalive); no copy (a 2^20 MSM hands over 96 MiB per call)."""
    if isinstance(b, bytes):
        return ctypes.cast(ctypes.c_char_p(b), ctypes.c_void_p), b
    if isinstance(b, bytearray):
        arr = (ctypes.c_char * len(b)).from_buffer(b)
        return ctypes.cast(arr, ctypes.c_void_p), arr
    # numpy array or anything exposing the buffer protocol
    mv = memoryview(b)
    if mv.readonly:
        raw = mv.tobytes()
        return ctypes.cast(ctypes.c_char_p(raw), ctypes.c_void_p), raw
    arr = (ctypes.c_char * mv.nbytes).from_buffer(b)
    return ctypes.cast(arr, ctypes.c_void_p), arr


class Context:
    """A device context (one GPU). Raises if no HIP device is usable."""

    def __init__(self, device=0):
        self._h = ctypes.c_void_p()
        err = ctypes.create_string_buffer(512)
        rc = lib().zkpoa_context_create(device, ctypes.byref(self._h), err, 512)
        if rc != PROVER_OK:
            self._h = None
            raise ZkpoaError("zkpoa_context_create: " + err.value.decode())

    def close(self):
        if self._h:
            lib().zkpoa_context_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc != PROVER_OK:
            raise ZkpoaError("%s failed (%d): %s" % (what, rc, lib().zkpoa_last_error(self._h).decode()))

    def stream(self, lane=0):
        """The HIP stream of a lane as an integer handle (wrap with torch.cuda.ExternalStream to order torch work /
        RCCL collectives with the library's kernels without host synchronisation)."""
        h = int(lib().zkpoa_context_stream(self._h, lane) or 0)
        if h == 0:      # 0 would silently become the legacy default stream in torch.cuda.ExternalStream
            raise ZkpoaError("zkpoa_context_stream: no stream for lane %d" % lane)
        return h

    def synchronize(self):
        """Wait for lane 0's stream (the split-chain stages only enqueue)."""
        self._check(lib().zkpoa_context_synchronize(self._h), "zkpoa_context_synchronize")

    def set_option(self, key, value):
        self._check(lib().zkpoa_set_option(self._h, key.encode(), int(value)), "zkpoa_set_option")

    def msm_points_limit(self):
        """Points one MSM of this context sorts at once right now: 2^27 until a lane's workspace did not fit in HBM (or
        under option lane_workspace_max_mb), then the size that did (include/zkpoa_prover.h, "Memory pressure")."""
        return int(lib().zkpoa_msm_points_limit(self._h))

    def last_ms(self, ident):
        return float(lib().zkpoa_last_ms(self._h, ident))

    # ---- stages, host buffers -------------------------------------------------------------
    def msm_g1(self, bases, scalars, n=None):
        n = len(scalars) // 32 if n is None else n
        pb, kb = _buf(bases)
        ps, ks = _buf(scalars)
        out = ctypes.create_string_buffer(64)
        self._check(lib().zkpoa_msm_g1(self._h, pb, ps, n, out), "zkpoa_msm_g1")
        return out.raw

    def msm_g2(self, bases, scalars, n=None):
        n = len(scalars) // 32 if n is None else n
        pb, kb = _buf(bases)
        ps, ks = _buf(scalars)
        out = ctypes.create_string_buffer(128)
        self._check(lib().zkpoa_msm_g2(self._h, pb, ps, n, out), "zkpoa_msm_g2")
        return out.raw

    def ntt(self, data, log_n, inverse=False):
        buf = bytearray(data)
        p, k = _buf(buf)
        self._check(lib().zkpoa_ntt(self._h, p, log_n, 1 if inverse else 0), "zkpoa_ntt")
        return bytes(buf)

    def h_scalars(self, coeffs_section, witness, n_vars, log_domain):
        pc, kc = _buf(coeffs_section)
        pw, kw = _buf(witness)
        out = ctypes.create_string_buffer(32 << log_domain)
        self._check(lib().zkpoa_h_scalars(self._h, pc, len(coeffs_section), pw, n_vars, log_domain, out),
                    "zkpoa_h_scalars")
        return out.raw

    # ---- stages, device pointers (ints) ---------------------------------------------------
    def msm_g1_device(self, d_bases, d_scalars, n):
        out = ctypes.create_string_buffer(64)
        self._check(lib().zkpoa_msm_g1_device(self._h, d_bases, d_scalars, n, out), "zkpoa_msm_g1_device")
        return out.raw

    def msm_g1_device_lane(self, lane, d_bases, d_scalars, n):
        """G1 MSM on lane `lane` (own stream + workspace); thread-safe across different lanes."""
        out = ctypes.create_string_buffer(64)
        if lib().zkpoa_msm_g1_device_lane(self._h, lane, d_bases, d_scalars, n, out) != PROVER_OK:
            raise ZkpoaError("zkpoa_msm_g1_device_lane failed")
        return out.raw

    def last_ms_lane(self, lane, ident):
        return float(lib().zkpoa_last_ms_lane(self._h, lane, ident))

    # ---- the anonymity-set Merkle tree (scripts/merkle_tree.rs) ---------------------------------------------------
    def poseidon2(self, left, right):
        """n independent circomlib Poseidon(2) hashes; left, right: n x 32 B LE standard form -> n x 32 B."""
        n = len(left) // 32
        pl, kl = _buf(left)
        pr, kr = _buf(right)
        out = ctypes.create_string_buffer(max(1, 32 * n))
        self._check(lib().zkpoa_poseidon2(self._h, pl, pr, n, out), "zkpoa_poseidon2")
        return out.raw[:32 * n]

    def poseidon2_device(self, d_left, d_right, n, d_out):
        self._check(lib().zkpoa_poseidon2_device(self._h, d_left, d_right, n, d_out), "zkpoa_poseidon2_device")

    def merkle_build(self, addresses, balances, device=False, n=None):
        """Tree over (address, balance) pairs: host buffers (n x 32 B LE each) or, with device=True, device pointers."""
        return MerkleTree(self, addresses, balances, device, n)

    def msm_table(self, group, d_bases, n, window_bits=0):
        """Fixed-base table over n device-resident bases (group 1 = G1, 2 = G2) -> MsmTable."""
        return MsmTable(self, group, d_bases, n, window_bits)

    def msm_table_run(self, table, d_scalars, lane=0):
        """sum k_i P_i over a table's bases, fixed-base form; thread-safe across different lanes."""
        out = ctypes.create_string_buffer(64 if table.group == 1 else 128)
        if lib().zkpoa_msm_table_run_lane(self._h, lane, table._h, d_scalars, out) != PROVER_OK:
            raise ZkpoaError("zkpoa_msm_table_run_lane failed")
        return out.raw

    def msm_g2_device(self, d_bases, d_scalars, n):
        out = ctypes.create_string_buffer(128)
        self._check(lib().zkpoa_msm_g2_device(self._h, d_bases, d_scalars, n, out), "zkpoa_msm_g2_device")
        return out.raw

    def ntt_device(self, d_data, log_n, inverse=False):
        self._check(lib().zkpoa_ntt_device(self._h, d_data, log_n, 1 if inverse else 0), "zkpoa_ntt_device")

    def gen_bases_g1_device(self, a, b, i0, n, d_out):
        self._check(lib().zkpoa_gen_bases_g1_device(self._h, int(a).to_bytes(32, "little"),
                                                    int(b).to_bytes(32, "little"), i0, n, d_out),
                    "zkpoa_gen_bases_g1_device")

    def gen_bases_g2_device(self, a, b, i0, n, d_out):
        self._check(lib().zkpoa_gen_bases_g2_device(self._h, int(a).to_bytes(32, "little"),
                                                    int(b).to_bytes(32, "little"), i0, n, d_out),
                    "zkpoa_gen_bases_g2_device")

    # ---- element-wise hooks ------------------------------------------------------------------
    def field_op(self, field, op, a, b=None):
        n = len(a) // 32
        pa, ka = _buf(a)
        pb, kb = _buf(b) if b is not None else (None, None)
        out = ctypes.create_string_buffer(max(1, 32 * n))
        self._check(lib().zkpoa_field_op(self._h, field, op, pa, pb, out, n), "zkpoa_field_op")
        return out.raw[:32 * n]

    def group_add(self, group, a, b):
        size = 64 if group == 1 else 128
        n = len(a) // size
        pa, ka = _buf(a)
        pb, kb = _buf(b)
        out = ctypes.create_string_buffer(max(1, size * n))
        self._check(lib().zkpoa_group_add(self._h, group, pa, pb, out, n), "zkpoa_group_add")
        return out.raw[:size * n]

    # ---- proving key + prove -----------------------------------------------------------------
    def load_zkey(self, zkey_bytes):
        return ZKey(self, zkey_bytes)

    def load_zkey_device(self, n_vars, n_public, log_domain, d_A, d_B1, d_B2, d_C, d_H, d_coefs, n_coefs,
                         header_points):
        """Proving key from device-resident sections (device pointers as ints); the caller keeps them alive."""
        key = ZKey.__new__(ZKey)
        key._ctx = self
        key._h = ctypes.c_void_p()
        self._check(lib().zkpoa_zkey_load_device(self._h, n_vars, n_public, log_domain, d_A, d_B1, d_B2, d_C, d_H,
                                                 d_coefs, n_coefs, bytes(header_points), ctypes.byref(key._h)),
                    "zkpoa_zkey_load_device")
        return key

    def setup_accumulate(self, group, d_points, n_points, d_coefs, d_point_index, d_signal, nnz, n_signals, d_out):
        """out[s] = sum of coefs[e] * points[point_index[e]] over the entries of signal s (device pointers as ints):
        one zkey point section of `snarkjs zkey new` (include/zkpoa_prover.h: zkpoa_setup_accumulate)."""
        self._check(lib().zkpoa_setup_accumulate(self._h, group, d_points, n_points, d_coefs, d_point_index, d_signal,
                                                 nnz, n_signals, d_out), "zkpoa_setup_accumulate")

    def zkey_new(self, r1cs_path, ptau_path, zkey_path):
        """`snarkjs zkey new` on files (include/zkpoa_prover.h: zkpoa_zkey_new)."""
        self._check(lib().zkpoa_zkey_new(self._h, os.fsencode(r1cs_path), os.fsencode(ptau_path),
                                         os.fsencode(zkey_path)), "zkpoa_zkey_new")

    def wtns_check(self, r1cs_path, wtns_path):
        """`snarkjs wtns check`: -> (number of violated constraints, smallest violated index or None)."""
        bad, first = ctypes.c_uint64(0), ctypes.c_uint64(0)
        self._check(lib().zkpoa_wtns_check(self._h, os.fsencode(r1cs_path), os.fsencode(wtns_path), ctypes.byref(bad),
                                           ctypes.byref(first)), "zkpoa_wtns_check")
        return int(bad.value), (int(first.value) if bad.value else None)

    def zkey_contribute(self, zkey_in_path, zkey_out_path, delta=None):
        """The arithmetic of `snarkjs zkey contribute` (delta: int in [1, r), None = random)."""
        d = None if delta is None else int(delta).to_bytes(32, "little")
        self._check(lib().zkpoa_zkey_contribute(self._h, os.fsencode(zkey_in_path), os.fsencode(zkey_out_path), d),
                    "zkpoa_zkey_contribute")

    def load_zkey_device_shard(self, n_vars, n_public, log_domain, rank, world, split, d_A, d_B1, d_B2, d_C, d_H,
                               d_coefs, n_coefs, header_points, block_log=0):
        """Shard `rank` of `world` from device-resident sections that hold only this rank's ranges -- or, with
        block_log = L > 0, its blocks of 2^L items (block b to rank b mod world) concatenated
        (include/zkpoa_prover.h: zkpoa_zkey_load_device_shard, ZKPOA_SHARD_*); the caller keeps the buffers alive."""
        key = ZKey.__new__(ZKey)
        key._ctx = self
        key._h = ctypes.c_void_p()
        self._check(lib().zkpoa_zkey_load_device_shard(self._h, n_vars, n_public, log_domain, rank, world,
                                                       shard_flags(split, block_log), d_A, d_B1, d_B2, d_C, d_H, d_coefs, n_coefs,
                                                       bytes(header_points), ctypes.byref(key._h)),
                    "zkpoa_zkey_load_device_shard")
        return key

    def load_zkey_shard(self, zkey_bytes, rank, world):
        """Shard `rank` of `world` of a proving key: only that byte range of sections 5-9 is uploaded."""
        key = ZKey.__new__(ZKey)
        key._ctx = self
        key._h = ctypes.c_void_p()
        p, k = _buf(zkey_bytes)
        self._check(lib().zkpoa_zkey_load_shard(self._h, p, len(zkey_bytes), rank, world, ctypes.byref(key._h)),
                    "zkpoa_zkey_load_shard")
        return key

    def load_zkey_shard_ex(self, zkey_bytes, rank, world, split=False, block_log=0):
        """Shard `rank` of `world` with shard flags (include/zkpoa_prover.h ZKPOA_SHARD_*): `split` = the H-scalar chain
        is split too; block_log = L > 0 = sections 5-8 dealt out in blocks of 2^L items instead of one range per rank
        (what the prover's own multi-GPU path, env ZKPOA_DEVICES, loads)."""
        key = ZKey.__new__(ZKey)
        key._ctx = self
        key._h = ctypes.c_void_p()
        p, k = _buf(zkey_bytes)
        self._check(lib().zkpoa_zkey_load_shard_ex(self._h, p, len(zkey_bytes), rank, world, shard_flags(split, block_log),
                                                   ctypes.byref(key._h)), "zkpoa_zkey_load_shard_ex")
        return key

    def load_zkey_shard_split(self, zkey_bytes, rank, world):
        """Like load_zkey_shard, with the H-scalar chain split too: this rank's constraint rows (c = rank mod
        world) of the coefficient list and the cyclic H-point shard H[t*world + rank]."""
        key = ZKey.__new__(ZKey)
        key._ctx = self
        key._h = ctypes.c_void_p()
        p, k = _buf(zkey_bytes)
        self._check(lib().zkpoa_zkey_load_shard_split(self._h, p, len(zkey_bytes), rank, world, ctypes.byref(key._h)),
                    "zkpoa_zkey_load_shard_split")
        return key

    def witness_load(self, zkey, wtns_bytes):
        """Parse a .wtns and upload it into the key's witness buffer -> public bytes."""
        pw, kw = _buf(wtns_bytes)
        npub = zkey.info()[1]
        pub = ctypes.create_string_buffer(max(1, 32 * npub))
        self._check(lib().zkpoa_witness_load(self._h, zkey._h, pw, len(wtns_bytes), pub, 32 * npub), "zkpoa_witness_load")
        return pub.raw[:32 * npub]

    def split_stage1(self, zkey, d_witness, d_exchange):
        self._check(lib().zkpoa_split_stage1(self._h, zkey._h, d_witness, d_exchange), "zkpoa_split_stage1")

    def split_stage2(self, zkey, d_received, d_exchange):
        self._check(lib().zkpoa_split_stage2(self._h, zkey._h, d_received, d_exchange), "zkpoa_split_stage2")

    def split_stage3(self, zkey, d_received):
        self._check(lib().zkpoa_split_stage3(self._h, zkey._h, d_received), "zkpoa_split_stage3")

    def prove_partials(self, zkey, wtns_bytes):
        """-> (partials[384] = A|B1|B2|C|H MSM results of the key's shard, public bytes)"""
        pw, kw = _buf(wtns_bytes)
        parts = ctypes.create_string_buffer(384)
        npub = zkey.info()[1]
        pub = ctypes.create_string_buffer(max(1, 32 * npub))
        self._check(lib().zkpoa_prove_partials(self._h, zkey._h, pw, len(wtns_bytes), parts, pub, 32 * npub),
                    "zkpoa_prove_partials")
        return parts.raw, pub.raw[:32 * npub]

    def prove_partials_device(self, zkey, d_witness):
        parts = ctypes.create_string_buffer(384)
        self._check(lib().zkpoa_prove_partials_device(self._h, zkey._h, d_witness, parts), "zkpoa_prove_partials_device")
        return parts.raw

    def read_h_scalars(self, zkey, count):
        """H-MSM scalars of the last prove on `zkey` (count x 32 B standard form) as a numpy uint64 [count, 4]."""
        import numpy as np
        out = np.empty((count, 4), dtype=np.uint64)
        self._check(lib().zkpoa_zkey_read_h_scalars(self._h, zkey._h, out.ctypes.data, out.nbytes),
                    "zkpoa_zkey_read_h_scalars")
        return out

    def prove_device(self, zkey, d_witness, r=None, s=None):
        """Prove with the witness already in HBM -> (proof_points[256], public bytes)."""
        rb = None if r is None else int(r).to_bytes(32, "little")
        sb = None if s is None else int(s).to_bytes(32, "little")
        proof = ctypes.create_string_buffer(256)
        npub = zkey.info()[1]
        pub = ctypes.create_string_buffer(max(1, 32 * npub))
        self._check(lib().zkpoa_prove_device(self._h, zkey._h, d_witness, rb, sb, proof, pub, 32 * npub),
                    "zkpoa_prove_device")
        return proof.raw, pub.raw[:32 * npub]

    def prove(self, zkey, wtns_bytes, r=None, s=None):
        """-> (proof_points[256 bytes], public[nPublic*32 bytes])"""
        pw, kw = _buf(wtns_bytes)
        rb = None if r is None else int(r).to_bytes(32, "little")
        sb = None if s is None else int(s).to_bytes(32, "little")
        proof = ctypes.create_string_buffer(256)
        npub = zkey.info()[1]
        pub = ctypes.create_string_buffer(max(1, 32 * npub))
        rc = lib().zkpoa_prove(self._h, zkey._h, pw, len(wtns_bytes), rb, sb, proof, pub, 32 * npub)
        self._check(rc, "zkpoa_prove")
        return proof.raw, pub.raw[:32 * npub]


class MerkleTree:
    """The anonymity-set Poseidon Merkle tree resident in HBM (zkpoa_merkle_build)."""

    def __init__(self, ctx, addresses, balances, device=False, n=None):
        self._ctx = ctx
        self._h = ctypes.c_void_p()
        if device:
            ctx._check(lib().zkpoa_merkle_build_device(ctx._h, addresses, balances, n, ctypes.byref(self._h)),
                       "zkpoa_merkle_build_device")
        else:
            n = len(addresses) // 32 if n is None else n
            pa, ka = _buf(addresses)
            pb, kb = _buf(balances)
            ctx._check(lib().zkpoa_merkle_build(ctx._h, pa, pb, n, ctypes.byref(self._h)), "zkpoa_merkle_build")

    def info(self):
        """(n, path length, nodes)"""
        out = (ctypes.c_uint64 * 3)()
        lib().zkpoa_merkle_info(self._h, out)
        return tuple(int(v) for v in out)

    def root(self):
        out = ctypes.create_string_buffer(32)
        self._ctx._check(lib().zkpoa_merkle_root(self._ctx._h, self._h, out), "zkpoa_merkle_root")
        return int.from_bytes(out.raw, "little")

    def leaves(self, first=0, count=None):
        count = (1 << self.info()[1]) - first if count is None else count
        out = ctypes.create_string_buffer(max(1, 32 * count))
        self._ctx._check(lib().zkpoa_merkle_leaves(self._ctx._h, self._h, first, count, out), "zkpoa_merkle_leaves")
        return out.raw[:32 * count]

    def path(self, index):
        """(sibling hashes from the leaves up as ints, index bits)"""
        k = self.info()[1]
        elems = ctypes.create_string_buffer(max(1, 32 * k))
        bits = ctypes.create_string_buffer(max(1, k))
        self._ctx._check(lib().zkpoa_merkle_path(self._ctx._h, self._h, index, elems, bits), "zkpoa_merkle_path")
        return ([int.from_bytes(elems.raw[32 * i:32 * i + 32], "little") for i in range(k)], list(bits.raw[:k]))

    def close(self):
        if self._h and self._ctx._h:
            lib().zkpoa_merkle_free(self._ctx._h, self._h)
        self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class MsmTable:
    """Fixed-base table 2^(c*j) * P_i of a resident base array (zkpoa_msm_table_build)."""

    def __init__(self, ctx, group, d_bases, n, window_bits=0):
        self._ctx, self.group = ctx, group
        self._h = ctypes.c_void_p()
        ctx._check(lib().zkpoa_msm_table_build(ctx._h, group, d_bases, n, window_bits, ctypes.byref(self._h)),
                   "zkpoa_msm_table_build")

    def info(self):
        """(n, window_bits, windows, bytes)"""
        out = (ctypes.c_uint64 * 4)()
        lib().zkpoa_msm_table_info(self._h, out)
        return tuple(int(v) for v in out)

    def close(self):
        if self._h and self._ctx._h:
            lib().zkpoa_msm_table_free(self._ctx._h, self._h)
        self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ZKey:
    """A proving key resident in HBM (zkey sections 4-9 uploaded once)."""

    def __init__(self, ctx, zkey_bytes):
        self._ctx = ctx
        self._h = ctypes.c_void_p()
        p, k = _buf(zkey_bytes)
        ctx._check(lib().zkpoa_zkey_load(ctx._h, p, len(zkey_bytes), ctypes.byref(self._h)), "zkpoa_zkey_load")

    def info(self):
        out = (ctypes.c_uint64 * 4)()
        lib().zkpoa_zkey_info(self._h, out)
        return tuple(int(v) for v in out)

    def set_shard(self, rank, world):
        """Restrict a fully resident key to shard `rank` of `world` (for zkpoa_prove_partials)."""
        if lib().zkpoa_zkey_set_shard(self._h, rank, world) != PROVER_OK:
            raise ZkpoaError("zkpoa_zkey_set_shard failed")

    def set_shard_split(self, rank, world):
        """set_shard + the H-scalar chain split over the same ranks (world in 2, 4, 8)."""
        self._ctx._check(lib().zkpoa_zkey_set_shard_split(self._ctx._h, self._h, rank, world),
                         "zkpoa_zkey_set_shard_split")

    def precompute(self, budget_bytes=0):
        """Build the key's fixed-base tables (H, C, A, B while they fit; 0 = half of the free HBM) -> bytes used."""
        used = ctypes.c_uint64(0)
        self._ctx._check(lib().zkpoa_zkey_precompute(self._ctx._h, self._h, budget_bytes, ctypes.byref(used)),
                         "zkpoa_zkey_precompute")
        return int(used.value)

    def vkey_points(self):
        """The verification key the zkey carries (sections 2-3): alpha1(64) beta2(128) gamma2(128) delta2(128) +
        IC[(nPublic+1) x 64], wire format; None for keys assembled from device buffers."""
        size = ctypes.c_ulong(0)
        if lib().zkpoa_zkey_vkey(self._h, None, ctypes.byref(size)) != PROVER_ERROR_SHORT_BUFFER:
            return None
        buf = ctypes.create_string_buffer(size.value)
        if lib().zkpoa_zkey_vkey(self._h, buf, ctypes.byref(size)) != PROVER_OK:
            raise ZkpoaError("zkpoa_zkey_vkey failed")
        return buf.raw

    def header(self):
        """alpha1(64) beta1(64) beta2(128) delta1(64) delta2(128), wire format."""
        out = ctypes.create_string_buffer(448)
        lib().zkpoa_zkey_header(self._h, out)
        return out.raw

    def close(self):
        if self._h and self._ctx._h:
            lib().zkpoa_zkey_free(self._ctx._h, self._h)
        self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _json_call(fn, *args):
    size = ctypes.c_ulong(0)
    rc = fn(*args, None, ctypes.byref(size))
    if rc not in (PROVER_OK, PROVER_ERROR_SHORT_BUFFER):
        raise ZkpoaError("json conversion failed")
    buf = ctypes.create_string_buffer(size.value)
    rc = fn(*args, buf, ctypes.byref(size))
    if rc != PROVER_OK:
        raise ZkpoaError("json conversion failed")
    return buf.value.decode()


def proof_to_json(proof_points, style="rapidsnark"):
    p, k = _buf(proof_points)
    return _json_call(lib().zkpoa_proof_to_json, p, 0 if style == "rapidsnark" else 1)


def public_to_json(public_le, style="rapidsnark"):
    p, k = _buf(public_le if len(public_le) else b"\0")
    return _json_call(lib().zkpoa_public_to_json, p, len(public_le) // 32, 0 if style == "rapidsnark" else 1)


VERIFY_BIN = os.path.join(_HERE, "zkpoa-verify")
SANITIZE_BIN = os.path.join(_HERE, "zkpoa-sanitize")


def groth16_verify(vkey_json, public_json, proof_json):
    """`snarkjs groth16 verify` on the three JSON texts (host only) -> True / False; raises on malformed input."""
    err = ctypes.create_string_buffer(512)
    rc = lib().zkpoa_groth16_verify(vkey_json.encode(), public_json.encode(), proof_json.encode(), err, 512)
    if rc == PROVER_OK:
        return True
    if rc == 0x10:
        return False
    raise ZkpoaError("zkpoa_groth16_verify: " + err.value.decode())


def export_vkey(zkey_bytes):
    """`snarkjs zkey export verificationkey`: the text of <circuit>_vkey.json from a zkey image (host only)."""
    L = lib()
    L.zkpoa_zkey_export_vkey.argtypes = [ctypes.c_void_p, ctypes.c_ulong, ctypes.c_void_p,
                                         ctypes.POINTER(ctypes.c_ulong), ctypes.c_void_p, ctypes.c_ulong]
    p, k = _buf(zkey_bytes)
    size = ctypes.c_ulong(0)
    err = ctypes.create_string_buffer(512)
    rc = L.zkpoa_zkey_export_vkey(p, len(zkey_bytes), None, ctypes.byref(size), err, 512)
    if rc != PROVER_ERROR_SHORT_BUFFER:
        raise ZkpoaError("zkpoa_zkey_export_vkey: " + err.value.decode())
    buf = ctypes.create_string_buffer(size.value)
    if L.zkpoa_zkey_export_vkey(p, len(zkey_bytes), buf, ctypes.byref(size), err, 512) != PROVER_OK:
        raise ZkpoaError("zkpoa_zkey_export_vkey: " + err.value.decode())
    return buf.value.decode()


def poseidon_params():
    """(round constants, MDS rows) the library generates (host only), as Python ints."""
    buf = ctypes.create_string_buffer(204 * 32)
    lib().zkpoa_poseidon_params.argtypes = [ctypes.c_void_p]
    if lib().zkpoa_poseidon_params(buf) != PROVER_OK:
        raise ZkpoaError("zkpoa_poseidon_params failed")
    vals = [int.from_bytes(buf.raw[32 * i:32 * i + 32], "little") for i in range(204)]
    return vals[:195], [vals[195 + 3 * i:198 + 3 * i] for i in range(3)]


def groth16_verify_points(vkey_points, proof_points, public_le):
    """The same check on wire-format points (ZKey.vkey_points(), Context.prove() outputs); host only."""
    err = ctypes.create_string_buffer(512)
    pub = bytes(public_le)
    rc = lib().zkpoa_groth16_verify_points(bytes(vkey_points), len(vkey_points), bytes(proof_points),
                                           pub if pub else None, len(pub) // 32, err, 512)
    if rc == PROVER_OK:
        return True
    if rc == 0x10:
        return False
    raise ZkpoaError("zkpoa_groth16_verify_points: " + err.value.decode())


def sanitize_proof(vkey_json, public_json, proof_json):
    """The text sanitize_groth16_proof.py writes to sanitized_proof.json (host only)."""
    err = ctypes.create_string_buffer(512)
    size = ctypes.c_ulong(1 << 16)
    buf = ctypes.create_string_buffer(size.value)
    rc = lib().zkpoa_sanitize_proof(vkey_json.encode(), public_json.encode(), proof_json.encode(), buf,
                                    ctypes.byref(size), err, 512)
    if rc == PROVER_ERROR_SHORT_BUFFER:
        buf = ctypes.create_string_buffer(size.value)
        rc = lib().zkpoa_sanitize_proof(vkey_json.encode(), public_json.encode(), proof_json.encode(), buf,
                                        ctypes.byref(size), err, 512)
    if rc != PROVER_OK:
        raise ZkpoaError("zkpoa_sanitize_proof: " + err.value.decode())
    return buf.value.decode()


def prove_assemble(header_points, partial_sums, r=None, s=None):
    """Host-only randomised assembly: header (448 B) + summed partials (384 B) + r, s -> proof_points[256]."""
    rb = None if r is None else int(r).to_bytes(32, "little")
    sb = None if s is None else int(s).to_bytes(32, "little")
    out = ctypes.create_string_buffer(256)
    if lib().zkpoa_prove_assemble(bytes(header_points), bytes(partial_sums), rb, sb, out) != PROVER_OK:
        raise ZkpoaError("zkpoa_prove_assemble failed")
    return out.raw


def sum_partials(partials_list):
    """Component-wise sum of per-rank partials (each 384 B = A|B1|B2|C|H) -> 384 B."""
    g1 = lambda lo: g1_sum(b"".join(p[lo:lo + 64] for p in partials_list))
    b2 = g2_sum(b"".join(p[128:256] for p in partials_list))
    return g1(0) + g1(64) + b2 + g1(256) + g1(320)


def g1_sum(points):
    p, k = _buf(points)
    out = ctypes.create_string_buffer(64)
    lib().zkpoa_g1_sum(p, len(points) // 64, out)
    return out.raw


def g2_sum(points):
    p, k = _buf(points)
    out = ctypes.create_string_buffer(128)
    lib().zkpoa_g2_sum(p, len(points) // 128, out)
    return out.raw


def g1_mul(point, k):
    out = ctypes.create_string_buffer(64)
    lib().zkpoa_g1_mul(point, int(k).to_bytes(32, "little"), out)
    return out.raw


def g2_mul(point, k):
    out = ctypes.create_string_buffer(128)
    lib().zkpoa_g2_mul(point, int(k).to_bytes(32, "little"), out)
    return out.raw


def groth16_prove(zkey_path, wtns_path, proof_path, public_path):
    """The reference's prove step (scripts/g16_prove.sh:248-252) through the C ABI:
    same four arguments as the `prover` executable."""
    with open(wtns_path, "rb") as f:
        wtns = f.read()
    pw, kw = _buf(wtns)
    psz = ctypes.c_ulong(1 << 12)
    usz = ctypes.c_ulong(1 << 20)
    proof = ctypes.create_string_buffer(psz.value)
    public = ctypes.create_string_buffer(usz.value)
    err = ctypes.create_string_buffer(1024)
    rc = lib().groth16_prover_zkey_file(os.fsencode(zkey_path), pw, len(wtns), proof, ctypes.byref(psz), public,
                                        ctypes.byref(usz), err, 1024)
    if rc == PROVER_ERROR_SHORT_BUFFER:
        proof = ctypes.create_string_buffer(psz.value)
        public = ctypes.create_string_buffer(usz.value)
        rc = lib().groth16_prover_zkey_file(os.fsencode(zkey_path), pw, len(wtns), proof, ctypes.byref(psz), public,
                                            ctypes.byref(usz), err, 1024)
    if rc != PROVER_OK:
        raise ZkpoaError("groth16_prover failed (%d): %s" % (rc, err.value.decode()))
    for path, text in ((proof_path, proof.value), (public_path, public.value)):
        tmp = "%s.tmp.%d" % (path, os.getpid())
        with open(tmp, "wb") as f:
            f.write(text)
        os.replace(tmp, path)


def groth16_prove_files(zkey_path, wtns_path, proof_path, public_path):
    """The same step with both inputs as paths (zkpoa_groth16_prover_files: what the `prover` executable calls; the
    witness goes from the page cache straight into the upload's pinned buffers, never mapped or copied on the host)."""
    psz = ctypes.c_ulong(1 << 12)
    usz = ctypes.c_ulong(1 << 20)
    proof = ctypes.create_string_buffer(psz.value)
    public = ctypes.create_string_buffer(usz.value)
    err = ctypes.create_string_buffer(1024)
    fn = lib().zkpoa_groth16_prover_files
    rc = fn(os.fsencode(zkey_path), os.fsencode(wtns_path), proof, ctypes.byref(psz), public, ctypes.byref(usz), err, 1024)
    if rc == PROVER_ERROR_SHORT_BUFFER:
        proof = ctypes.create_string_buffer(psz.value)
        public = ctypes.create_string_buffer(usz.value)
        rc = fn(os.fsencode(zkey_path), os.fsencode(wtns_path), proof, ctypes.byref(psz), public, ctypes.byref(usz), err, 1024)
    if rc != PROVER_OK:
        raise ZkpoaError("zkpoa_groth16_prover_files failed (%d): %s" % (rc, err.value.decode()))
    for path, text in ((proof_path, proof.value), (public_path, public.value)):
        tmp = "%s.tmp.%d" % (path, os.getpid())
        with open(tmp, "wb") as f:
            f.write(text)
        os.replace(tmp, path)
