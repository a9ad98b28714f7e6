#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X Groth16 prove path (contract: see the build prompt).

Default workload = BASELINE.json configs[1]: standalone BN254 G1 Pippenger MSM, 2^20 random
points / scalars, inputs resident in HBM. One "step" = one complete MSM over the rank's 2^20-point
slice (weak scaling: with N ranks the job is one N*2^20-point MSM whose per-rank partial points are
all-gathered over RCCL and summed on every rank -- SURVEY.md 8e).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload msm_g1_2p20|msm_g1_2pXX|prove_2pXX]

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

R_MOD = 21888242871839275222246405745257275088548364400416034343698204186575808495617
VALU_PEAK_GADDS = 12.1      # XYZZ mixed additions/s: register-only loop of the same addition on one MI355X at its best
                            # occupancy (tools/microbench.hip: 11.6 at 3 waves/SIMD, 12.1 at 4)
HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
G1_MSM_BYTES_PER_POINT = 96      # 64 B base + 32 B scalar, each read once


def pmc_traffic(workload, kernel_substr):
    """HBM-side bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes
    (FETCH_SIZE + WRITE_SIZE, separate passes, calibrated on this access pattern:
    profiles/r01_pmc_hbm_traffic.json). None when this workload has not been profiled."""
    path = os.path.join(ROOT, "profiles", "r01_pmc_hbm_traffic.json")
    try:
        with open(path) as f:
            prof = json.load(f)
        for name, v in prof["workloads"].get(workload, {}).items():
            if kernel_substr in name:
                return v["bytes_per_launch"]
    except (OSError, ValueError, KeyError):
        pass
    return None


def np_scalars(n, seed):
    import numpy as np
    nr = np.random.default_rng(seed)
    limbs = nr.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64) * 2 + nr.integers(0, 2, size=(n, 4), dtype=np.uint64)
    limbs[:, 3] &= np.uint64((1 << 60) - 1)       # < 2^252 < r: uniform 252-bit scalars
    return limbs


def dlog_expected(limbs, a, b, i0):
    import numpy as np
    n = limbs.shape[0]
    idx = np.arange(i0, i0 + n, dtype=object)
    s0 = s1 = 0
    for j in range(4):
        col = limbs[:, j].astype(object)
        s0 += int(col.sum()) << (64 * j)
        s1 += int((col * idx).sum()) << (64 * j)
    return (a * s0 + b * s1) % R_MOD


# circuit shapes of the reference's own test runs (SURVEY.md 8 "Sizes at BASELINE.json configs")
PROVE_SHAPES = {
    16: (60000, 2, "test-size synthetic"),
    21: (2083343, 1, "layer_one(2 sigs) shape, tests/4_sigs_2_batches_12_height/benchmarks.txt:17-23"),
    25: (21356921, 2, "layer_two(2,12) shape, tests/4_sigs_2_batches_12_height/benchmarks.txt:33-39"),
    26: (61197000, 1, "synthetic layer_one(128 sigs) shape, tests/old/128_sigs/benchmarks.txt:4-10"),
}


def bench_prove(args, zk, dist, rank, world, local_rank, dev):
    """Full Groth16 prove against a key + witness resident in HBM. N>1: the proof's five MSMs are sharded
    over the ranks (BASELINE.json configs[3..4]); the H-scalar chain is replicated."""
    import torch
    from zkpoa_amd.synthetic import SyntheticCircuit
    k = int(args.workload[len("prove_2p"):])
    m, n_pub, what = PROVE_SHAPES[k]
    from zkpoa_amd import sharding
    ctx = zk.Context(local_rank)
    # N > 1: ONE proof sharded over the N GPUs (strong scaling): same circuit on every rank, each rank
    # owns index range rank/N of the five MSMs; partial points are all-gathered over RCCL and summed.
    circ = SyntheticCircuit(zk, ctx, k, m, n_public=n_pub, seed=0x5EED0010, witness_like=True)
    header = circ.key.header()
    split = world > 1 and not args.replicated_chain and sharding.split_chain_supported(world, 1 << k)
    if split:
        # the H-scalar chain is split over the ranks too: rows c = rank (mod N), two all-to-alls per polynomial
        circ.key.set_shard_split(rank, world)
        xbufs = sharding.exchange_buffers(1 << k, world, dev)
    elif world > 1:
        circ.key.set_shard(rank, world)

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier(**args.barrier_kw)
        torch.cuda.synchronize()

    def one_proof():
        if world == 1:
            return circ.prove(0, 0)[0]
        if split:
            return sharding.sharded_prove_split(ctx, circ.key, circ.d_witness.data_ptr(), header, zk.sum_partials,
                                                zk.prove_assemble, 0, 0, dist, args.cdev, xbufs)
        return sharding.sharded_prove(lambda: ctx.prove_partials_device(circ.key, circ.d_witness.data_ptr()),
                                      header, zk.sum_partials, zk.prove_assemble, 0, 0, dist, args.cdev)

    for _ in range(args.warmup):
        one_proof()
    sync()
    t0 = time.perf_counter()
    acc = {"h_chain": 0.0, "msm_phase": 0.0, "prove": 0.0, "h_msm_accum": 0.0}
    names = ("H", "A", "B1", "B2", "C")
    acc.update({"msm_%s" % x: 0.0 for x in names})
    for i in range(args.steps):
        pts = one_proof()
        acc["h_chain"] += ctx.last_ms(3)
        acc["msm_phase"] += ctx.last_ms(4)
        acc["prove"] += ctx.last_ms(5)
        acc["h_msm_accum"] += ctx.last_ms(1)
        for lane, x in enumerate(names):                 # device time of each MSM on its own lane (they overlap)
            acc["msm_%s" % x] += ctx.last_ms_lane(lane, 0)
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=args.cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if not circ.check(pts, 0, 0):
        raise SystemExit("bench.py: proof failed the known-dlog check (pi_a / pi_b)")
    if rank == 0:
        n = 1 << k
        ncoef = circ.n_coef
        # SURVEY.md 8d full-prove formula
        alg = 96 * (3 * m - n_pub - 1) + 160 * m + 96 * n + 6 * 64 * n + 76 * ncoef + 96 * n + 128 * n
        sec = elapsed / args.steps
        line = {
            "metric": "Groth16 proofs/sec", "value": args.steps / elapsed, "unit": "proofs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": sec * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "u32x8 (254-bit modular integer)", "data": "synthetic" + (" (REHEARSAL: ranks share a GPU, gloo "
                                                                              "collectives; not a measurement)" if args.rehearse else ""),
            "config": {"workload": "full Groth16 prove, domain 2^%d, %d wires, %d public (%s); key and witness "
                                   "resident in HBM; witness-like scalar distribution" % (k, m, n_pub, what),
                       "n_coefs": ncoef, "parallelism": ("one proof, five MSMs sharded over %d GPUs by index range, all-gather of partial "
                                       "points; H-scalar chain %s" % (world, "split (four-step NTTs, 2 all-to-alls per "
                                       "polynomial)" if split else "replicated")) if world > 1 else "single GPU"},
            "roofline": {"bound": "hbm", "kernel": "whole prove (5 MSMs + H chain)",
                         "achieved": alg / sec / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": alg / sec / 1e9 / HBM_PEAK_GBS, "traffic": None,
                         "algorithmic_bytes": alg,
                         "phase_ms": {kk: v / args.steps for kk, v in acc.items()}},
        }
        print(json.dumps(line), flush=True)
    if dist.is_initialized():
        dist.barrier(**args.barrier_kw)
        dist.destroy_process_group()
    circ.close()
    ctx.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="msm_g1_2p20")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--replicated-chain", action="store_true",
                    help="prove workloads, N > 1: run the whole H-scalar chain on every rank instead of splitting it")
    ap.add_argument("--inflight", type=int, default=6, help="MSMs kept in flight on separate HIP streams (1..6)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist
    from __graft_entry__ import load_package
    zk = load_package()
    from zkpoa_amd import sharding

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run for N>1)"
                         % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py: no GPU visible; the hot path is HIP-only (no CPU fallback)")
    # ZKPOA_BENCH_REHEARSE=1: rehearsal of the N > 1 code path on a box with fewer GPUs than ranks -- the ranks
    # share the visible GPU(s) and the collectives run over gloo through the host. Never a measurement.
    rehearse = os.environ.get("ZKPOA_BENCH_REHEARSE") == "1"
    if rehearse:
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    args.cdev = torch.device("cpu") if rehearse else dev           # where collective payloads live
    args.barrier_kw = {} if rehearse else {"device_ids": [local_rank]}
    args.rehearse = rehearse
    force_dist = os.environ.get("ZKPOA_BENCH_FORCE_DIST") == "1"   # exercise the RCCL path with one rank
    if world > 1 or force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if rehearse:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    if args.workload.startswith("prove_2p"):
        return bench_prove(args, zk, dist, rank, world, local_rank, dev)
    if args.workload.startswith("msm_g1_2p"):
        logn = int(args.workload[len("msm_g1_2p"):])
    else:
        raise SystemExit("unknown workload " + args.workload)
    n_local = 1 << logn
    ctx = zk.Context(local_rank)
    if os.environ.get("ZKPOA_MSM_C"):        # experiments only: force the Pippenger window width
        ctx.set_option("msm_c", int(os.environ["ZKPOA_MSM_C"]))
    if logn > 27:
        raise SystemExit("msm workload limited to 2^27 points per GPU")

    # ---- synthetic inputs, resident in HBM: bases (a + i*b)*G with known discrete logs, uniform scalars
    seeds = random.Random(0x5EED0001)
    a, b = seeds.randrange(R_MOD), seeds.randrange(R_MOD)
    i0 = rank * n_local
    d_bases = torch.empty(n_local * 64, dtype=torch.uint8, device=dev)
    ctx.gen_bases_g1_device(a, b, i0, n_local, d_bases.data_ptr())
    limbs = np_scalars(n_local, 0x5EED0002 + rank)
    d_scalars = torch.from_numpy(limbs.view(np.uint8).reshape(-1).copy()).to(dev)

    # `inflight` MSMs are kept in flight, each on its own lane (HIP stream + workspace) driven by its own
    # host thread: the sort / bucket-reduction / read-back / host-Horner phases of one MSM are small or
    # latency-bound and overlap with the accumulation kernel of the next (the prover does the same with
    # its five MSMs). Collectives stay on the main thread, in step order.
    from collections import deque
    from concurrent.futures import ThreadPoolExecutor
    inflight = max(1, min(args.inflight, 6))
    pool = ThreadPoolExecutor(inflight)

    def msm_on(lane):
        part = ctx.msm_g1_device_lane(lane, d_bases.data_ptr(), d_scalars.data_ptr(), n_local)
        return part, ctx.last_ms_lane(lane, 1), ctx.last_ms_lane(lane, 0)

    def combine(part):
        if world == 1 and not force_dist:
            return part
        return sharding.combine_partials(zk.g1_sum, sharding.all_gather_bytes(part, dist, args.cdev))

    def run(steps):
        """-> (last result, sum of accumulate-kernel ms, sum of whole-MSM device ms)"""
        q = deque(pool.submit(msm_on, i % inflight) for i in range(min(inflight, steps)))
        res, k_ms, d_ms = None, 0.0, 0.0
        for i in range(steps):
            part, k1, k0 = q.popleft().result()
            if i + inflight < steps:
                q.append(pool.submit(msm_on, i % inflight))      # lane i % inflight is free again: refill it
            res = combine(part)                                  # ... before the (blocking) collective
            k_ms += k1     # HIP events on the MSM's own stream, inside the library
            d_ms += k0
        return res, k_ms, d_ms

    def sync():
        torch.cuda.synchronize()
        if world > 1 or force_dist:
            dist.barrier(**args.barrier_kw)
        torch.cuda.synchronize()

    run(max(args.warmup, inflight))      # also sizes every lane's workspace outside the timed region
    sync()
    t0 = time.perf_counter()
    result, kernel_ms, msm_dev_ms = run(args.steps)
    sync()
    elapsed = time.perf_counter() - t0
    pool.shutdown()
    if world > 1 or force_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=args.cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- correctness of what was timed (outside the timed region): known-dlog check
    d_loc = dlog_expected(limbs, a, b, i0)
    if world > 1 or force_dist:
        parts = [None] * world
        dist.all_gather_object(parts, d_loc)
        d_all = sum(parts) % R_MOD
    else:
        d_all = d_loc
    G = (1).to_bytes(32, "little") + (2).to_bytes(32, "little")
    one_m = (1 << 256) % 21888242871839275222246405745257275088696311157297823662689037894645226208583
    two_m = (2 << 256) % 21888242871839275222246405745257275088696311157297823662689037894645226208583
    Gm = one_m.to_bytes(32, "little") + two_m.to_bytes(32, "little")
    expected = zk.g1_mul(Gm, d_all)
    if result != expected:
        raise SystemExit("bench.py: MSM result failed the known-dlog check")

    if rank == 0:
        pts_total = n_local * world * args.steps
        k_ms = kernel_ms / args.steps
        achieved = G1_MSM_BYTES_PER_POINT * n_local / (k_ms * 1e-3) / 1e9
        # the bound that actually binds: one XYZZ mixed addition per (point, window) entry; window width from the
        # library's cost model (csrc/msm.hip.h msm_make_plan), ceiling = register-only loop of the same addition
        # measured on this chip (tools/microbench2.hip)
        c_win = int(os.environ.get("ZKPOA_MSM_C") or
                    min(range(4, 23), key=lambda c: ((254 + c - 1) // c) * (n_local + 8.0 * (1 << (c - 1)))))
        adds = n_local * ((254 + c_win - 1) // c_win)
        gadds = adds / (k_ms * 1e-3) / 1e9
        line = {
            "metric": "G1-MSM throughput", "value": pts_total / elapsed, "unit": "pts/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u32x8 (254-bit modular integer)",
            "data": "synthetic" + (" (REHEARSAL: ranks share a GPU, gloo collectives; not a measurement)"
                                   if args.rehearse else ""),
            "config": {"workload": "BN254 G1 Pippenger MSM, 2^%d points per GPU, uniform 252-bit scalars, "
                                   "bases (a+i*b)*G resident in HBM (BASELINE.json configs[1])" % logn,
                       "points_per_gpu": n_local, "sharding": "index ranges, all-gather of partial points",
                       "msms_in_flight": inflight},
            "roofline": {"bound": "hbm", "kernel": "msm_accum0_kernel<Fq> (bucket accumulation)",
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS,
                         "traffic": pmc_traffic(args.workload, "msm_accum0_kernel"),
                         "traffic_source": "profiles/r01_pmc_hbm_traffic.json (rocprofv3 --pmc FETCH_SIZE / "
                                           "WRITE_SIZE, bytes per launch; each base is re-read once per window)",
                         "kernel_ms": k_ms, "msm_device_ms": msm_dev_ms / args.steps,
                         "valu": {"unit": "G mixed additions/s", "achieved": gadds, "peak": VALU_PEAK_GADDS,
                                  "frac": gadds / VALU_PEAK_GADDS, "additions_per_launch": adds, "window_bits": c_win,
                                  "note": "kernel_ms is per launch with %d MSMs in flight, so launches overlap "
                                          "other kernels; --inflight 1 gives the kernel alone" % inflight},
                         "note": "algorithmic bytes = 96 B/point x points per launch; the kernel is "
                                 "integer-VALU-bound (v_mad_u64_u32), not HBM-bound: see DESIGN.md"},
        }
        if world == 1 and not args.no_cpu_baseline:
            from oracle import c_oracle as co
            cores = os.cpu_count() or 1
            sample_log = min(logn, 20)      # ~10-20 s of CPU work spread over the host cores
            ns = 1 << sample_log
            hb = bytes(d_bases[:ns * 64].cpu().numpy())
            hs = limbs[:ns].tobytes()
            windows = co.msm_windows(ns)     # the C oracle threads by window: threads used = min(cores, windows)
            threads = min(cores, windows)
            reps = 3                         # ~0.5 s each on 16 threads: ~25 core-seconds in total
            tc = time.perf_counter()
            for _ in range(reps):
                ref = co.msm_g1(hb, hs, ns, threads)
            tcpu = (time.perf_counter() - tc) / reps
            chk = ctx.msm_g1_device(d_bases.data_ptr(), d_scalars.data_ptr(), ns)
            if chk != ref:
                raise SystemExit("bench.py: GPU and CPU-oracle MSM disagree on the baseline sample")
            line["cpu_baseline"] = {"value": ns / tcpu, "unit": "pts/s", "cores": threads, "kind": "port",
                                    "sample": "first 2^%d points of the same workload, mean of %d MSMs, C oracle "
                                              "(oracle/c, Pippenger threaded by window)" % (sample_log, reps),
                                    "seconds": tcpu * reps}
        print(json.dumps(line), flush=True)
    if world > 1 or force_dist:
        dist.barrier(**args.barrier_kw)
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
