"""The C-ABI boundary without a GPU: libzkpoa_prover.so loads, exports every symbol that
include/zkpoa_prover.h declares, host-only entry points work, and anything that needs the GPU
fails loudly instead of falling back to the CPU."""
import ctypes
import os
import re
import subprocess

import pytest

from conftest import ROOT, golden_case
from oracle.py import bn254 as bn
from oracle.py import groth16 as g16


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "zkpoa_prover.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = re.findall(r"\b((?:zkpoa|groth16)_\w+)\s*\(", text)
    return sorted(set(names))


def test_header_symbols_all_exported(zk):
    L = zk.lib()
    declared = _declared_symbols()
    assert len(declared) >= 25
    missing = [s for s in declared if not hasattr(L, s)]
    assert missing == []
    assert sorted(zk.EXPORTS) == declared


def test_host_only_group_helpers(zk):
    G = g16.g1_to_bytes(bn.G1_GEN)
    assert g16.g1_from_bytes(zk.g1_mul(G, 123456789)) == bn.g1_mul(bn.G1_GEN, 123456789)
    G2 = g16.g2_to_bytes(bn.G2_GEN)
    assert g16.g2_from_bytes(zk.g2_mul(G2, bn.R - 5)) == bn.g2_mul(bn.G2_GEN, bn.R - 5)
    pts = [bn.g1_mul(bn.G1_GEN, k) for k in (3, 5, 7)] + [None]
    assert g16.g1_from_bytes(zk.g1_sum(b"".join(g16.g1_to_bytes(p) for p in pts))) == bn.g1_mul(bn.G1_GEN, 15)
    # P + (-P) and doubling through the sum path
    P = bn.g1_mul(bn.G1_GEN, 99)
    assert g16.g1_from_bytes(zk.g1_sum(g16.g1_to_bytes(P) + g16.g1_to_bytes(bn.ec_neg(P, bn.FQ)))) is None
    assert g16.g1_from_bytes(zk.g1_sum(g16.g1_to_bytes(P) * 2)) == bn.g1_mul(bn.G1_GEN, 198)
    pts2 = [bn.g2_mul(bn.G2_GEN, k) for k in (2, 9)]
    assert g16.g2_from_bytes(zk.g2_sum(b"".join(g16.g2_to_bytes(p) for p in pts2))) == bn.g2_mul(bn.G2_GEN, 11)


@pytest.mark.parametrize("tag", ["n8", "n128"])
def test_json_writers_match_reference_formats(zk, tag):
    """zkpoa_proof_to_json / zkpoa_public_to_json (host-only) reproduce the golden JSON bytes, which
    follow the reference's committed rapidsnark- and snarkjs-style files."""
    import json
    g = golden_case(tag)
    obj = json.loads(g["proof_rapidsnark.json"])
    pts = (g16.g1_to_bytes(g16.g1_from_obj(obj["pi_a"])) + g16.g2_to_bytes(g16.g2_from_obj(obj["pi_b"])) +
           g16.g1_to_bytes(g16.g1_from_obj(obj["pi_c"])))
    pub = b"".join(int(v).to_bytes(32, "little") for v in json.loads(g["public_rapidsnark.json"]))
    assert zk.proof_to_json(pts, "rapidsnark") == g["proof_rapidsnark.json"]
    assert zk.proof_to_json(pts, "snarkjs") == g["proof_snarkjs.json"]
    assert zk.public_to_json(pub, "rapidsnark") == g["public_rapidsnark.json"]
    assert zk.public_to_json(pub, "snarkjs") == g["public_snarkjs.json"]


def test_json_writer_on_reference_fixture(zk):
    """Round-trip one of the reference's own proof.json files through the product's writer."""
    import json
    d = os.path.join(ROOT, "tests", "golden", "ref", "4_sigs_2_batches_12_height__layer_three")
    raw = open(os.path.join(d, "proof.json")).read()
    obj = json.loads(raw)
    pts = (g16.g1_to_bytes(g16.g1_from_obj(obj["pi_a"])) + g16.g2_to_bytes(g16.g2_from_obj(obj["pi_b"])) +
           g16.g1_to_bytes(g16.g1_from_obj(obj["pi_c"])))
    assert zk.proof_to_json(pts, "rapidsnark") == raw
    rawp = open(os.path.join(d, "public.json")).read()
    pub = b"".join(int(v).to_bytes(32, "little") for v in json.loads(rawp))
    assert zk.public_to_json(pub, "rapidsnark") == rawp


def _no_gpu():
    import torch
    return not torch.cuda.is_available()


@pytest.mark.skipif(not _no_gpu(), reason="only meaningful on a box without a GPU")
def test_no_gpu_fails_loudly(zk, tmp_path):
    with pytest.raises(zk.ZkpoaError):
        zk.Context(0)
    g = golden_case("n8")
    (tmp_path / "c.zkey").write_bytes(g["circuit.zkey"])
    (tmp_path / "w.wtns").write_bytes(g["witness.wtns"])
    with pytest.raises(zk.ZkpoaError):
        zk.groth16_prove(str(tmp_path / "c.zkey"), str(tmp_path / "w.wtns"), str(tmp_path / "p.json"),
                         str(tmp_path / "u.json"))
    rc = subprocess.run([zk.PROVER_BIN, str(tmp_path / "c.zkey"), str(tmp_path / "w.wtns"),
                         str(tmp_path / "p.json"), str(tmp_path / "u.json")], capture_output=True, text=True)
    assert rc.returncode != 0 and "Error" in rc.stderr
    assert not (tmp_path / "p.json").exists()


def test_cli_usage_error(zk):
    rc = subprocess.run([zk.PROVER_BIN, "only-one-arg"], capture_output=True, text=True)
    assert rc.returncode != 0 and "Usage" in rc.stderr


def test_cli_worker_process_reports_a_failure(zk, tmp_path):
    """`prover` with a worker process (ZKPOA_DETACH_EXIT=always; by default for keys of 2 GB or more): an input error in
    the worker is this command's exit status and message, and nothing is left behind. No GPU needed to get that far."""
    (tmp_path / "c.zkey").write_bytes(b"zkey" + bytes(40))
    rc = subprocess.run([zk.PROVER_BIN, str(tmp_path / "c.zkey"), str(tmp_path / "missing.wtns"), str(tmp_path / "p.json"),
                         str(tmp_path / "q.json")], env=dict(os.environ, ZKPOA_DETACH_EXIT="always"), capture_output=True,
                        text=True, timeout=60)
    assert rc.returncode == 1 and "cannot read witness file" in rc.stderr
    assert not (tmp_path / "p.json").exists() and not (tmp_path / "q.json").exists()


def test_product_does_not_reference_oracle():
    """The product tree must not import, link or execute anything under oracle/."""
    bad = []
    for base, _, files in os.walk(os.path.join(ROOT, "zk-proof-of-assets_amd")):
        if os.sep + "build" in base:
            continue
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".hpp", ".cpp", "Makefile")):
                text = open(os.path.join(base, f), errors="ignore").read()
                if re.search(r"oracle[/.]|liboracle|c_oracle|import oracle|from oracle", text):
                    bad.append(os.path.join(base, f))
    assert bad == []


def test_prover_server_lifecycle_without_gpu(zk, tmp_path):
    """Server mode of the CLI on a box without a GPU: the resident process starts, answers with the library's
    loud "no HIP device" failure (exit 1, no outputs), and stops on request. On a GPU box the same sequence
    is the parity test tests/test_gpu_prove.py::test_prover_cli_server_mode."""
    import subprocess
    import time
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the gpu-marked server test")
    g = golden_case("n8")
    (tmp_path / "c.zkey").write_bytes(g["circuit.zkey"])
    (tmp_path / "w.wtns").write_bytes(g["witness.wtns"])
    sock = str(tmp_path / "p.sock")
    env = dict(os.environ, ZKPOA_SERVER=sock, ZKPOA_SERVER_IDLE_S="30")
    try:
        rc = subprocess.run([zk.PROVER_BIN, "c.zkey", "w.wtns", "proof.json", "public.json"], env=env, cwd=tmp_path,
                            capture_output=True, text=True, timeout=120)
        # a HIP runtime that cannot come up is a runtime failure (PROVER_ERROR_RUNTIME), not an input error: the server
        # says so and leaves (the next call starts a fresh one), the client logs it, tries in its own process and
        # fails as loudly
        assert rc.returncode == 1 and "no HIP device" in rc.stderr and "reported a GPU runtime failure" in rc.stderr
        assert not (tmp_path / "proof.json").exists() and not (tmp_path / "public.json").exists()
        rc = subprocess.run([zk.PROVER_BIN, "c.zkey", "missing.wtns", "proof.json", "public.json"], env=env,
                            cwd=tmp_path, capture_output=True, text=True, timeout=120)
        assert rc.returncode == 1 and "cannot read witness file" in rc.stderr
        assert os.path.exists(sock)                                  # an input error: the (new) server answered and stays
    finally:
        rc = subprocess.run([zk.PROVER_BIN, "--stop-server"], env=env, cwd=tmp_path, timeout=60)
    assert rc.returncode == 0
    for _ in range(100):
        if not os.path.exists(sock):
            break
        time.sleep(0.05)
    assert not os.path.exists(sock)


def test_prover_server_failure_paths_without_gpu(zk, tmp_path):
    """(1) a server that dies with a request in hand does not fail the call -- the client proves in its own
    process (here: reaches the library, which has no GPU on this box); (2) a client that connects and says nothing
    does not hold the single-threaded server. On a GPU box tests/test_gpu_prove.py checks that (1) yields a proof."""
    import socket
    import subprocess
    import time
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the gpu-marked server tests")
    g = golden_case("n8")
    (tmp_path / "c.zkey").write_bytes(g["circuit.zkey"])
    (tmp_path / "w.wtns").write_bytes(g["witness.wtns"])
    argv = [zk.PROVER_BIN, "c.zkey", "w.wtns", "proof.json", "public.json"]
    # (1) the server exits on receipt of the request
    sock = str(tmp_path / "crash.sock")
    env = dict(os.environ, ZKPOA_SERVER=sock, ZKPOA_SERVER_IDLE_S="30", ZKPOA_SERVER_TEST_CRASH="1")
    rc = subprocess.run(argv, env=env, cwd=tmp_path, capture_output=True, text=True, timeout=120)
    assert "went away without answering; proving in-process" in rc.stderr
    assert rc.returncode == 1 and "no HIP device" in rc.stderr       # the in-process attempt ran (and has no GPU here)
    # (2) a silent connection is dropped after the receive timeout and the next request is served
    sock = str(tmp_path / "quiet.sock")
    env = dict(os.environ, ZKPOA_SERVER=sock, ZKPOA_SERVER_IDLE_S="30", ZKPOA_SERVER_RCV_TIMEOUT_S="1")
    # (requests that fail on their INPUT -- a missing witness -- so that the server stays up on this GPU-less box: a HIP
    # runtime that cannot start makes it leave, see test_prover_server_lifecycle_without_gpu)
    argv_bad = [zk.PROVER_BIN, "c.zkey", "missing.wtns", "proof.json", "public.json"]
    try:
        rc = subprocess.run(argv_bad, env=env, cwd=tmp_path, capture_output=True, text=True, timeout=120)
        assert os.path.exists(sock)
        quiet = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
        quiet.connect(sock)                                           # ... and never sends a byte
        t0 = time.time()
        rc = subprocess.run(argv_bad, env=env, cwd=tmp_path, capture_output=True, text=True, timeout=60)
        assert rc.returncode == 1 and "cannot read witness file" in rc.stderr and "in-process" not in rc.stderr
        assert time.time() - t0 < 20
        quiet.close()
    finally:
        subprocess.run([zk.PROVER_BIN, "--stop-server"], env=env, cwd=tmp_path, timeout=60)


def test_automatic_gpu_choice_of_the_multi_gpu_drop_in(zk):
    """csrc/multi_device.hip.h auto_pick_devices (through a test hook; no GPU needed): what `prover` does on a node with
    several GPUs when neither ZKPOA_DEVICES nor ZKPOA_DEVICE is set -- the policy behind "full_workflow.sh unchanged":
    small keys take ONE free GPU, the search starting at pid mod #GPUs (parallel batch jobs spread out); keys of
    2^24 constraints or more take every free GPU, cut to 2 / 4 / 8; nothing free -> queue behind GPU pid mod #GPUs."""
    import ctypes
    L = zk.lib()
    f = L.zkpoa_test_auto_pick_devices
    f.argtypes = [ctypes.c_int, ctypes.c_uint, ctypes.c_uint, ctypes.c_ulong, ctypes.c_uint, ctypes.POINTER(ctypes.c_int)]

    def pick(count, power, pid, busy=0, min_power=24):
        out = (ctypes.c_int * 8)()
        n = f(count, power, min_power, pid, busy, out)
        return list(out[:n])
    # layer-one / layer-two sized keys: one GPU each, different processes start at different GPUs, busy ones are skipped
    assert pick(8, 21, pid=1000) == [0] and pick(8, 21, pid=1003) == [3] and pick(8, 23, pid=1007) == [7]
    assert pick(8, 21, pid=1003, busy=0b00001000) == [4]
    assert pick(8, 21, pid=1007, busy=0b10000000) == [0]                       # wraps around
    assert pick(4, 21, pid=6, busy=0b1111) == [2]                              # all taken: wait for pid mod count
    assert sorted({tuple(pick(8, 21, pid=p)) for p in range(100, 108)}) == [(d,) for d in range(8)]
    # a layer-three sized key (2^26) alone on the node: all 8; with some GPUs busy: the free ones, cut to a power of two
    assert pick(8, 26, pid=5) == list(range(8))
    assert pick(8, 26, pid=5, busy=0b00000001) == [1, 2, 3, 4]                 # 7 free -> 4
    assert pick(8, 26, pid=5, busy=0b11110000) == [0, 1, 2, 3]
    assert pick(8, 26, pid=5, busy=0b11111010) == [0, 2]
    assert pick(8, 26, pid=5, busy=0b11111110) == [0]
    assert pick(8, 26, pid=13, busy=0b11111111) == [5]                         # nothing free: queue behind pid mod 8
    assert pick(2, 25, pid=9) == [0, 1] and pick(3, 25, pid=9) == [0, 1] and pick(16, 26, pid=1) == list(range(8))
    assert pick(8, 23, pid=2, min_power=23) == list(range(8))                  # ZKPOA_MULTI_MIN_POWER


def test_block_size_of_the_block_cyclic_shards(zk):
    """csrc/multi_device.hip.h multi_block_log_default (test hook, no GPU): sections 5-8 are dealt out in blocks of 2^16
    items, fewer for small keys so that every rank still holds at least eight blocks, never fewer than 2^4."""
    import ctypes
    f = zk.lib().zkpoa_test_multi_block_log
    f.argtypes = [ctypes.c_uint64, ctypes.c_uint]
    f.restype = ctypes.c_uint
    # the reference's shapes: layer one (2.08 M wires: 31 blocks of 2^16), two (21.4 M), three (52.4 M)
    assert [f(2083343, g) for g in (1, 2, 4, 8)] == [16, 16, 15, 14]
    assert [f(21356921, g) for g in (1, 2, 4, 8)] == [16] * 4 and [f(52367163, g) for g in (2, 4, 8)] == [16] * 3
    for n, g in [(0, 1), (1, 8), (15, 2), (16, 1), (127, 1), (128, 1), (60000, 2), (60000, 8), (7000, 4), (1 << 19, 1), ((1 << 19) - 1, 1),
                 (1 << 32, 8)]:
        L = f(n, g)
        assert 4 <= L <= 16
        assert L == 4 or (n >> L) >= 8 * g            # at least eight blocks per rank ...
        assert L == 16 or (n >> (L + 1)) < 8 * g       # ... and the largest block size that still gives that many
    assert f(0, 1) == 4 and f(128, 1) == 4 and f(1 << 19, 1) == 16 and f((1 << 19) - 1, 1) == 15


def test_malformed_device_environment_is_an_input_error(zk, tmp_path):
    """ZKPOA_DEVICE / ZKPOA_SHARD_BLOCK_LOG / ZKPOA_MULTI_MIN_POWER are parsed with range checks: a bad value
    fails the call with a message naming the variable (no GPU needed: nothing is proved on this box anyway, but the
    input error must come first wherever a device count is not needed to see it)."""
    import subprocess
    g = golden_case("n8")
    (tmp_path / "c.zkey").write_bytes(g["circuit.zkey"])
    (tmp_path / "w.wtns").write_bytes(g["witness.wtns"])
    argv = [zk.PROVER_BIN, "c.zkey", "w.wtns", "proof.json", "public.json"]
    import torch
    for var, bad in (("ZKPOA_DEVICE", "7x"), ("ZKPOA_DEVICE", "-1"), ("ZKPOA_DEVICE", "99")):
        env = dict(os.environ, **{var: bad})
        env.pop("ZKPOA_SERVER", None)
        rc = subprocess.run(argv, env=env, cwd=tmp_path, capture_output=True, text=True, timeout=120)
        assert rc.returncode == 1 and not (tmp_path / "proof.json").exists()
        if torch.cuda.is_available():
            assert var in rc.stderr, rc.stderr
        else:
            assert "no HIP device" in rc.stderr or var in rc.stderr


def test_server_pool_answers_concurrent_requests_without_gpu(zk, tmp_path):
    """The resident prover serves from a pool of threads (r04). Sixteen clients at once, each with its own missing witness
    (an INPUT error, answered before any GPU work, so the server stays up on this box): every client gets its own message
    and exit code 1, nothing is written, and the server still stops on request. (The same sixteen under ThreadSanitizer
    -- the `prover` executable is host-only code: g++ -fsanitize=thread -- report nothing)"""
    import subprocess
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by tests/test_gpu_prove.py::test_server_overlaps_requests_without_mixing_them_up")
    g = golden_case("n8")
    (tmp_path / "c.zkey").write_bytes(g["circuit.zkey"])
    env = dict(os.environ, ZKPOA_SERVER=str(tmp_path / "p.sock"), ZKPOA_SERVER_IDLE_S="30", ZKPOA_SERVER_WORKERS="4")
    try:
        procs = [subprocess.Popen([zk.PROVER_BIN, "c.zkey", "missing%d.wtns" % i, "p%d.json" % i, "u%d.json" % i], env=env,
                                  cwd=tmp_path, stderr=subprocess.PIPE, text=True) for i in range(16)]
        for i, pr in enumerate(procs):
            _, err = pr.communicate(timeout=120)
            assert pr.returncode == 1 and ("missing%d.wtns" % i) in err and "cannot read witness file" in err, err
            assert not (tmp_path / ("p%d.json" % i)).exists()
        assert os.path.exists(tmp_path / "p.sock")
    finally:
        rc = subprocess.run([zk.PROVER_BIN, "--stop-server"], env=env, cwd=tmp_path, timeout=60)
    assert rc.returncode == 0
