"""The N>1 path on CPU: world_size-2 gloo process group, each rank computes the MSM of its slice
(the C oracle stands in for the GPU kernel here), partial points are all-gathered and summed with
the product's host-side group sum -- the same code path bench.py runs per GPU over RCCL."""
import os
import random
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, le


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, n, seed, q):
    import sys
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from __graft_entry__ import load_package
    from oracle import c_oracle as co
    from oracle.py import bn254 as bn
    zk = load_package()
    sharding = __import__("zkpoa_amd.sharding", fromlist=["x"])
    rng = random.Random(seed)
    dl = [rng.randrange(1, bn.R) for _ in range(n)]
    bases = co.fixed_base_g1(b"".join(le(k) for k in dl))
    scalars = b"".join(le(rng.choice([0, 1, bn.R - 1, rng.randrange(bn.R)])) for _ in range(n))
    full = sharding.sharded_msm(
        lambda lo, hi: co.msm_g1(bases[64 * lo:64 * hi], scalars[32 * lo:32 * hi], hi - lo),
        zk.g1_sum, n, dist)
    q.put((rank, full.hex(), co.msm_g1(bases, scalars, n).hex()))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_range_partitions():
    from __graft_entry__ import load_package
    load_package()
    from zkpoa_amd import sharding
    for n in (0, 1, 7, 8, 1000003):
        for world in (1, 2, 3, 8):
            spans = [sharding.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


@pytest.mark.parametrize("n", [5, 301])
def test_two_rank_sharded_msm_gloo(n):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, 4242, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert len({r[1] for r in results}) == 1          # identical on every rank
    assert results[0][1] == results[0][2]             # equals the unsharded MSM


def _prove_worker(rank, world, port, q):
    """Sharded prove host logic on CPU: each rank's partial MSMs come from the Python oracle on its index
    ranges; all-gather (gloo) + component-wise sum + zkpoa_prove_assemble must equal the oracle's proof."""
    import sys
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from __graft_entry__ import load_package
    from conftest import golden_case
    from oracle.py import bn254 as bn
    from oracle.py import groth16 as g16
    zk = load_package()
    from zkpoa_amd import sharding
    g = golden_case("n128")
    zkey = g16.read_zkey(g["circuit.zkey"])
    _, w = g16.read_wtns(g["witness.wtns"])
    import json
    rs = json.loads(g["rs.json"])
    P = g16.h_scalars(zkey, w)
    m, npub, n = zkey.nVars, zkey.nPublic, zkey.domainSize

    def partials():
        lo, hi = sharding.shard_range(m, rank, world)
        clo, chi = sharding.shard_range(m - npub - 1, rank, world)
        hlo, hhi = sharding.shard_range(n, rank, world)
        A = bn.msm_naive(zkey.A[lo:hi], w[lo:hi], bn.FQ)
        B1 = bn.msm_naive(zkey.B1[lo:hi], w[lo:hi], bn.FQ)
        B2 = bn.msm_naive(zkey.B2[lo:hi], w[lo:hi], bn.FQ2)
        C = bn.msm_naive(zkey.C[clo:chi], w[npub + 1 + clo:npub + 1 + chi], bn.FQ)
        H = bn.msm_naive(zkey.H[hlo:hhi], P[hlo:hhi], bn.FQ)
        return (g16.g1_to_bytes(A) + g16.g1_to_bytes(B1) + g16.g2_to_bytes(B2) + g16.g1_to_bytes(C) +
                g16.g1_to_bytes(H))

    header = (g16.g1_to_bytes(zkey.alpha1) + g16.g1_to_bytes(zkey.beta1) + g16.g2_to_bytes(zkey.beta2) +
              g16.g1_to_bytes(zkey.delta1) + g16.g2_to_bytes(zkey.delta2))
    pts = sharding.sharded_prove(partials, header, zk.sum_partials, zk.prove_assemble, int(rs["r"]), int(rs["s"]), dist)
    q.put((rank, zk.proof_to_json(pts, "rapidsnark"), g["proof_rapidsnark.json"]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharded_prove_gloo():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_prove_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for _, got, want in results:
        assert got == want


def _split_worker(rank, world, port, q):
    """The product's orchestration of the split H-scalar chain (sharding.split_h_chain: stage, all-to-all,
    stage, all-to-all, stage) over gloo, with the big-int stage model standing in for the HIP stages."""
    import sys
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from __graft_entry__ import load_package
    from conftest import golden_case
    from oracle.py import groth16 as g16
    from split_model import SplitRank
    load_package()
    from zkpoa_amd import sharding
    g = golden_case("n128")
    zkey = g16.read_zkey(g["circuit.zkey"])
    _, w = g16.read_wtns(g["witness.wtns"])
    assert sharding.split_chain_supported(world, zkey.domainSize)
    model = SplitRank(zkey, w, rank, world)
    a, b = sharding.exchange_buffers(zkey.domainSize, world, torch.device("cpu"))
    sharding.split_h_chain(model.stage1, model.stage2, model.stage3, a, b, dist)
    q.put((rank, model.h, g16.h_scalars(zkey, w)[rank::world]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_split_h_chain_gloo(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_split_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(r[0] for r in results) == list(range(world))
    for _, got, want in results:
        assert got == want          # rank g ends with the H scalars of the odd-coset indices i = g (mod G)


def test_split_chain_supported():
    from __graft_entry__ import load_package
    load_package()
    from zkpoa_amd import sharding
    assert sharding.split_chain_supported(8, 1 << 26) and sharding.split_chain_supported(2, 4)
    assert not sharding.split_chain_supported(3, 1 << 20)      # replicated chain instead
    assert not sharding.split_chain_supported(8, 32)
    assert not sharding.split_chain_supported(1, 1 << 20)


def test_block_cyclic_blocks_partition_the_section(zk):
    """sharding.block_cyclic_blocks (what bench.py --gpus N and the prover's own multi-GPU path deal out, include/
    zkpoa_prover.h ZKPOA_SHARD_BLOCK_CYCLIC): the ranks' blocks cover [0, n) exactly once, every block but the globally
    last is full, rank sizes differ by at most one block, and the flag word packs as the header's macros do."""
    import random
    from zkpoa_amd.sharding import block_cyclic_blocks
    rng = random.Random(5)
    for _ in range(300):
        n, world, L = rng.randrange(0, 100000), rng.randrange(1, 9), rng.randrange(4, 13)
        seen, sizes = [], []
        for rank in range(world):
            runs = block_cyclic_blocks(n, rank, world, L)
            assert all(start % (1 << L) == 0 and (start >> L) % world == rank for start, _ in runs)
            assert all(cnt == (1 << L) for _, cnt in runs[:-1])
            seen += [(s, c) for s, c in runs]
            sizes.append(sum(c for _, c in runs))
        seen.sort()
        assert sum(c for _, c in seen) == n
        pos = 0
        for s, c in seen:
            assert s == pos and c > 0
            pos += c
        assert max(sizes) - min(sizes) <= (1 << L)
    assert zk.shard_flags(True, 16) == 0x1001 and zk.shard_flags(False, 0) == 0 and zk.shard_flags(False, 7) == 0x700
