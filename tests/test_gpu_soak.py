"""Soak of the resident prover (ZKPOA_SERVER): many proofs of one key from three clients at a time, each with the
prover's own random blinding (r, s from /dev/urandom: no two proofs are alike), EVERY proof checked by the native
verifier against the key's verification key, and the card's memory compared before and after -- a leak of a witness
buffer, a temporary or a lane workspace per request would show as growth. The key is a VALID setup (the synthetic keys
of the big shapes are not), made through the C oracle's fixed-base generator.

By default 90 proofs (seconds); ZKPOA_SOAK_PROOFS=3000 is the run recorded in DESIGN.md section 7."""
import json
import os
import random
import subprocess
import threading
import time

import pytest

from conftest import le
from oracle import c_oracle as co
from oracle.py import bn254 as bn
from oracle.py import groth16 as g16

pytestmark = pytest.mark.gpu


def test_server_soak_every_proof_verified_no_memory_growth(zk, tmp_path):
    import torch
    total = int(os.environ.get("ZKPOA_SOAK_PROOFS", "90"))
    clients = 3
    rng = random.Random(77)
    nVars, nPublic, nCons = 30000, 2, 32000                       # 2^15 domain
    cons, w = g16.random_circuit(rng, nVars, nPublic, nCons)
    tox = {k: rng.randrange(1, bn.R) for k in ("tau", "alpha", "beta", "gamma", "delta")}
    zkey, vk = g16.synthetic_setup(nVars, nPublic, cons, tox,
                                   g1_batch=lambda s: co.fixed_base_g1(b"".join(le(k) for k in s), 8),
                                   g2_batch=lambda s: co.fixed_base_g2(b"".join(le(k) for k in s), 8))
    (tmp_path / "c.zkey").write_bytes(zkey)
    (tmp_path / "w.wtns").write_bytes(g16.write_wtns(w))
    (tmp_path / "vkey.json").write_text(json.dumps(vk))
    sock = str(tmp_path / "prover.sock")
    env = dict(os.environ, ZKPOA_SERVER=sock, ZKPOA_SERVER_IDLE_S="120", ZKPOA_SERVER_WORKERS="3")
    for k in ("ZKPOA_R", "ZKPOA_S", "ZKPOA_JSON", "ZKPOA_VERBOSE"):
        env.pop(k, None)

    def prove(tag):
        rc = subprocess.run([zk.PROVER_BIN, "c.zkey", "w.wtns", "p_%s.json" % tag, "u_%s.json" % tag], env=env, cwd=tmp_path,
                            capture_output=True, text=True, timeout=120)
        assert rc.returncode == 0, rc.stderr
        rc = subprocess.run([zk.VERIFY_BIN, "vkey.json", "u_%s.json" % tag, "p_%s.json" % tag], cwd=tmp_path,
                            capture_output=True, text=True, timeout=60)
        assert rc.returncode == 0, "proof %s does not verify: %s %s" % (tag, rc.stdout, rc.stderr)
        return (tmp_path / ("p_%s.json" % tag)).read_text()

    try:
        for i in range(3):                                           # start, load, tables (idle time), warm-up
            prove("warm%d" % i)
            time.sleep(0.5)
        time.sleep(1.5)
        free0 = torch.cuda.mem_get_info()[0]
        seen, errors = set(), []

        def client(idx):
            try:
                for j in range(total // clients):
                    seen.add(prove("c%d" % idx))                    # (files reused per client: the content is what counts)
            except BaseException as e:                               # noqa: BLE001 -- reported by the main thread
                errors.append(e)

        t0 = time.perf_counter()
        th = [threading.Thread(target=client, args=(i,)) for i in range(clients)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        dt = time.perf_counter() - t0
        assert not errors, errors[0]
        done = (total // clients) * clients
        assert len(seen) == done, "two requests were handed the same proof (the blinding is random per request)"
        time.sleep(1.0)
        free1 = torch.cuda.mem_get_info()[0]
        grown = (free0 - free1) / 1e6
        print("soak: %d proofs, all verified, in %.1f s (%.1f ms per proof incl. the client process and the verifier); "
              "HBM in use grew by %.1f MB" % (done, dt, dt / done * 1e3, grown))
        assert grown < 64.0, "HBM in use grew by %.1f MB over %d proofs" % (grown, done)
        log = open(sock + ".log").read() if os.path.exists(sock + ".log") else ""
        assert "failure" not in log.lower() and "fault" not in log.lower(), log[-2000:]
    finally:
        subprocess.run([zk.PROVER_BIN, "--stop-server"], env=env, cwd=tmp_path, timeout=60)
