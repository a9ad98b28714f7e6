import json
import os
import random
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def zk():
    """The product package (zk-proof-of-assets_amd/, loaded as module zkpoa_amd)."""
    import __graft_entry__ as entry
    mod = entry.load_package()
    if not all(os.path.exists(p) for p in (mod.LIB_PATH, mod.PROVER_BIN, mod.VERIFY_BIN, mod.MERKLE_BIN)):
        entry.build()           # fresh checkout: compile the HIP library + CLIs (hipcc cross-compiles without a GPU)
    return mod


@pytest.fixture(scope="session")
def ctx(zk):
    """A device context; creating it without a GPU raises (no CPU fallback)."""
    c = zk.Context(0)
    yield c
    c.close()


@pytest.fixture(scope="session")
def vectors():
    with open(os.path.join(GOLDEN, "gen", "vectors.json")) as f:
        return json.load(f)


def golden_case(tag):
    d = os.path.join(GOLDEN, "gen", tag)
    out = {}
    for name in os.listdir(d):
        mode = "rb" if name.endswith((".zkey", ".wtns", ".bin")) else "r"
        with open(os.path.join(d, name), mode) as f:
            out[name] = f.read()
    return out


@pytest.fixture(scope="session")
def rng():
    return random.Random(20240)


def le(x):
    return int(x).to_bytes(32, "little")


def rd(b, i=0):
    return int.from_bytes(b[32 * i:32 * i + 32], "little")
