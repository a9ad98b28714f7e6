"""BASELINE.json configs[0] -- plumbing through the REFERENCE's own prove script (build container only).

Runs /root/reference/scripts/g16_prove.sh (never copied into this repo; skipped where the reference is absent, e.g.
on the GPU box) with `-b` against this repo's `prover`, using the stand-ins SURVEY.md 8(b) describes for the tools
the container lacks: a stub witness generator at <build>/<circuit>_cpp/<circuit> that copies a ready-made .wtns, a
stub file named `node`, and a `time` shim on PATH (scripts/lib/cmd_executor.sh:17 calls `\\time --quiet`).
What it pins: the argv order the script execs (g16_prove.sh:246-252), the `prover` basename rule (:195-199), the
`<build>/<circuit>_final.zkey` default (lib/g16_utils.sh:39), and the failure convention -- on a box without a GPU
the library refuses to compute ("no HIP device"), the exit code is non-zero, the script's ERR trap prints
"ERROR GENERATING PROOF USING RAPIDSNARK" (lib/error_handling.sh:14-41) and no proof.json is left behind. With a
GPU (and the reference present) the same run must produce the golden proof bytes."""
import json
import os
import shutil
import stat
import subprocess

import pytest

from conftest import golden_case

REF_SCRIPT = "/root/reference/scripts/g16_prove.sh"
pytestmark = pytest.mark.skipif(not os.path.exists(REF_SCRIPT), reason="reference checkout not present")


def _exe(path, text):
    path.write_text(text)
    path.chmod(path.stat().st_mode | stat.S_IXUSR | stat.S_IXGRP | stat.S_IXOTH)


def _layout(tmp_path, zk, name="layer_one"):
    g = golden_case("n128")
    build = tmp_path / "build"
    (build / (name + "_cpp")).mkdir(parents=True)
    (build / (name + "_final.zkey")).write_bytes(g["circuit.zkey"])          # lib/g16_utils.sh:39 naming
    (tmp_path / "ready.wtns").write_bytes(g["witness.wtns"])
    # circom's C++ witness calculator is invoked as `<gen> <input.json> <witness.wtns>` (g16_prove.sh:230-232)
    _exe(build / (name + "_cpp") / name, '#!/bin/bash\nset -e\n[ -f "$1" ]\ncp "%s" "$2"\n' % (tmp_path / "ready.wtns"))
    (tmp_path / (name + ".circom")).write_text("// contents unused by the prove step\n")
    (tmp_path / "input.json").write_text("{}")
    shim = tmp_path / "shim"
    shim.mkdir()
    _exe(shim / "time", '#!/bin/bash\n[ "$1" = "--quiet" ] && shift\nexec "$@"\n')   # GNU time is not installed here
    (tmp_path / "node").write_text("")                                       # verify_patched_node_path: a FILE named node
    env = {k: v for k, v in os.environ.items() if not k.startswith("ZKPOA_")}
    env["PATH"] = str(shim) + os.pathsep + env.get("PATH", "")
    rs = json.loads(g["rs.json"])
    env.update(ZKPOA_R=rs["r"], ZKPOA_S=rs["s"])
    argv = ["bash", REF_SCRIPT, "-b", "-B", str(build), "-p", str(tmp_path / "batch_0"), "-n", str(tmp_path / "node"),
            "-r", zk.PROVER_BIN, str(tmp_path / (name + ".circom")), str(tmp_path / "input.json")]
    return g, env, argv


def test_reference_script_reaches_the_prover_and_fails_loudly_without_gpu(zk, tmp_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: see the gpu-marked variant")
    g, env, argv = _layout(tmp_path, zk)
    rc = subprocess.run(argv, env=env, capture_output=True, text=True, timeout=120)
    out = rc.stdout + rc.stderr
    assert "GENERATING WITNESS USING C++ CODE" in out
    assert (tmp_path / "batch_0" / "witness.wtns").read_bytes() == g["witness.wtns"]   # the stub generator ran
    assert "GENERATING PROOF USING RAPIDSNARK" in out                                  # the script reached the exec
    assert rc.returncode != 0
    assert "no HIP device" in rc.stderr                                                # our binary: loud, no CPU fallback
    assert "ERROR GENERATING PROOF USING RAPIDSNARK" in out                            # the reference's ERR trap fired
    assert "DONE G16 PROVE" not in out
    assert not (tmp_path / "batch_0" / "proof.json").exists()
    assert not (tmp_path / "batch_0" / "public.json").exists()
    assert not [f for f in os.listdir(tmp_path / "batch_0") if ".tmp." in f]           # no partial outputs either


def test_reference_script_with_a_device_list_in_the_environment(zk, tmp_path):
    """The multi-GPU drop-in is configured by environment only (ZKPOA_DEVICES), so it passes through the unchanged
    script like ZKPOA_R / ZKPOA_S do: without a GPU the same loud failure, exit code and ERR trap, no partial outputs."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: see the gpu-marked variant")
    g, env, argv = _layout(tmp_path, zk)
    rc = subprocess.run(argv, env=dict(env, ZKPOA_DEVICES="0,1,2,3"), capture_output=True, text=True, timeout=120)
    out = rc.stdout + rc.stderr
    assert rc.returncode != 0 and "no HIP device" in rc.stderr and "ERROR GENERATING PROOF USING RAPIDSNARK" in out
    assert not (tmp_path / "batch_0" / "proof.json").exists()
    assert not [f for f in os.listdir(tmp_path / "batch_0") if ".tmp." in f]


def test_reference_script_rejects_a_prover_with_another_basename(zk, tmp_path):
    """g16_prove.sh:195-199: the binary must be called `prover` -- which is why ours is."""
    g, env, argv = _layout(tmp_path, zk)
    other = tmp_path / "zkpoa_prover"
    shutil.copy(zk.PROVER_BIN, other)
    argv[argv.index("-r") + 1] = str(other)
    rc = subprocess.run(argv, env=env, capture_output=True, text=True, timeout=120)
    assert rc.returncode != 0 and "must point to a file with name 'prover'" in rc.stdout + rc.stderr
    assert os.path.basename(zk.PROVER_BIN) == "prover"


@pytest.mark.gpu
@pytest.mark.parametrize("devices", [None, "0,0", "0,0,0,0"])
def test_reference_script_end_to_end_on_gpu(zk, tmp_path, devices):
    """Same run where both the reference and a GPU exist: golden proof bytes through the unchanged script -- on one
    device, and with one proof over 2 / 4 ranks (ZKPOA_DEVICES; ranks share the device on a one-GPU box)."""
    g, env, argv = _layout(tmp_path, zk)
    if devices:
        env = dict(env, ZKPOA_DEVICES=devices)
    rc = subprocess.run(argv, env=env, capture_output=True, text=True, timeout=300)
    assert rc.returncode == 0, rc.stdout + rc.stderr
    assert "DONE G16 PROVE" in rc.stdout
    assert (tmp_path / "batch_0" / "proof.json").read_text() == g["proof_rapidsnark.json"]
    assert (tmp_path / "batch_0" / "public.json").read_text() == g["public_rapidsnark.json"]
