#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X Groth16 prove path (contract: see the build prompt).

Default workload = BASELINE.json configs[1]: standalone BN254 G1 Pippenger MSM, 2^20 random
points / scalars, inputs resident in HBM. One "step" = one complete MSM over the rank's 2^20-point
slice (weak scaling: with N ranks the job is one N*2^20-point MSM whose per-rank partial points are
all-gathered over RCCL and summed on every rank -- SURVEY.md 8e).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload msm_g1_2p20|msm_g1_2pXX|prove_2pXX|merkle_<leaves>] [--no-also]

Prints ONE JSON line on rank 0. BASELINE.json's metric is "Groth16 proofs/sec + G1-MSM pts/s at 2^20 / 2^26", so
with N = 1 and the default workload the same line carries, under "also", the other shapes of that metric measured
in the same process: the 2^26 MSM, the 2^20 MSM in fixed-base form, and full proves at the layer_one(2 sigs)
2^21 shape, the layer_two(2, 12) 2^25 shape, the synthetic layer_one(128 sigs) 2^26 shape and the layer_three(2)
2^26 / 13-public shape (prove_2p26_l3) -- each with its own step count, roofline object and correctness check (pi_c
included). With --gpus N > 1 the default workload is ONE 2^26 proof over the N GPUs (strong scaling). `value` is always the headline workload alone.
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
Q_MOD = 21888242871839275222246405745257275088696311157297823662689037894645226208583
VALU_PEAK_GADDS = 14.2      # XYZZ mixed additions/s the integer pipe allows on one MI355X, from the measured rates of its parts
                            # (tools/microbench.hip, profiles/r03_microbench.txt, 8 waves/SIMD): 6 products at 139.8 G/s + one
                            # sum of two products under a single reduction (y3: Fq::dot2, half an Fq2 product: 90.6 G/s) + 2
                            # squares at 170.2 G/s + 7 additions / subtractions / negations of ~24 instructions each (0.09 of
                            # a product's 272) = 70.3 ps per addition. Until r04's y3 change the addition had 8 products
                            # (73.4 ps: 13.6 G/s, the peak of every earlier line). The register-only loop of the addition
                            # was r02's "peak" -- the kernel, whose base loads hide under the arithmetic, runs above it, so it
                            # was not a ceiling
G2_VALU_PEAK_GADDS = 4.67   # the same bound for the G2 mixed addition: 8 Fq2 products at 45.3 G/s (176.6 ps) + 2 Fq2 squares =
                            # 4 Fq products at 139.8 G/s (28.6 ps) + 14 Fq additions / subtractions of 24 instructions, 0.088
                            # of a product each (8.9 ps) = 214 ps (profiles/r03_microbench.txt; VERDICT r03 quoted 3.74 from a
                            # mis-priced addition term, and 2.92 G/s for the kernel from a loop that was compiled for ONE
                            # wave per SIMD: at the kernel's two the register-only loop runs 4.06, profiles/r04_microbench4.txt)
HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
G1_MSM_BYTES_PER_POINT = 96      # 64 B base + 32 B scalar, each read once
DTYPE = "u32x8 (254-bit modular integer)"


def oracle_g1(k):
    """k * G1 generator in the wire format, by the ORACLE's scalar multiplication (oracle/py: big-int double-and-add) --
    the checker of every known-discrete-log expectation below; the product's own host code is not consulted."""
    from oracle.py import bn254 as obn
    from oracle.py import groth16 as og
    return og.g1_to_bytes(obn.g1_mul(obn.G1_GEN, k % R_MOD))


def oracle_g2(k):
    from oracle.py import bn254 as obn
    from oracle.py import groth16 as og
    return og.g2_to_bytes(obn.g2_mul(obn.G2_GEN, k % R_MOD))


class BenchError(Exception):
    """A correctness check of something that was timed failed."""


def pmc_traffic(workload, kernel_substr):
    """HBM-side bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes
    (FETCH_SIZE + WRITE_SIZE, separate passes, calibrated on this access pattern:
    profiles/rNN_pmc_hbm_traffic.json, newest round first). None when this workload has not been profiled."""
    for rnd in ("r04", "r03", "r02", "r01"):
        path = os.path.join(ROOT, "profiles", "%s_pmc_hbm_traffic.json" % rnd)
        try:
            with open(path) as f:
                prof = json.load(f)
            for name, v in prof["workloads"].get(workload, {}).items():
                if kernel_substr in name:
                    return v["bytes_per_launch"], "profiles/%s_pmc_hbm_traffic.json" % rnd
        except (OSError, ValueError, KeyError):
            continue
    return None, None


def np_scalars(n, seed):
    import numpy as np
    nr = np.random.default_rng(seed)
    limbs = nr.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64) * 2 + nr.integers(0, 2, size=(n, 4), dtype=np.uint64)
    limbs[:, 3] &= np.uint64((1 << 60) - 1)       # < 2^252 < r: uniform 252-bit scalars
    return limbs


def host_cores():
    """Cores this process may run on (the GPU box gives a job a share of the host, not all of it)."""
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    try:                                   # cgroup v2 CPU quota: "max 100000" or "<quota> <period>"
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(-(-int(quota) // int(period)))))
    except (OSError, ValueError):
        pass
    if os.environ.get("ZKPOA_BENCH_THREADS"):
        n = int(os.environ["ZKPOA_BENCH_THREADS"])
    return n


def box_cores():
    """Logical CPUs of the GPU box itself (the job's share of them is host_cores())."""
    return os.cpu_count() or 1


def classic_window(n):
    """Window width the library's cost model picks for a classic n-point MSM (csrc/msm.hip.h msm_make_plan)."""
    def cost(c):
        w = (254 + c - 1) // c
        return w * n * (1.0 + 0.03 * ((c - 1 + 7) // 8)) + 3.0 * w * (1 << (c - 1))
    return min(range(4, 23), key=cost)


# circuit shapes of the reference's own test runs (SURVEY.md 8 "Sizes at BASELINE.json configs")
PROVE_SHAPES = {
    16: (60000, 2, "test-size synthetic"),
    21: (2083343, 1, "layer_one(2 sigs) shape, tests/4_sigs_2_batches_12_height/benchmarks.txt:17-23"),
    25: (21356921, 2, "layer_two(2,12) shape, tests/4_sigs_2_batches_12_height/benchmarks.txt:33-39"),
    26: (61197000, 1, "synthetic layer_one(128 sigs) shape, tests/old/128_sigs/benchmarks.txt:4-10"),
    "26_l3": (52367163, 13, "layer_three(2) shape, tests/4_sigs_2_batches_12_height/benchmarks.txt:49-55"),
    # beyond the reference's own runs: layer three grows by 24.2 M wires per batch (28.1 M at 1 batch, 52.4 M at 2:
    # tests/{1_sigs_1_batches_5_height,4_sigs_2_batches_12_height}/benchmarks.txt), so 4 batches need a 2^27 domain
    "27_l3": (100845225, 13, "layer_three(4 batches) shape, extrapolated from the reference's 1- and 2-batch runs"),
    # the largest domain the proving system has (Fr has 2^28-th roots of unity only): layer three over 8 batches
    "28_l3": (197801349, 13, "layer_three(8 batches) shape, extrapolated likewise; 2^28 = the largest Groth16 domain over BN254"),
}


def prove_shape(spec):
    """'21' | '25' | '26' | '26_l3' (workload prove_2p<spec>) -> (log2 domain, wires, public signals, description)."""
    key = int(spec) if str(spec).isdigit() else str(spec)
    if key not in PROVE_SHAPES:
        raise SystemExit("unknown prove workload prove_2p%s (known: %s)" % (spec, ", ".join(str(k) for k in PROVE_SHAPES)))
    m, n_pub, what = PROVE_SHAPES[key]
    return int(str(spec).split("_")[0]), m, n_pub, what


class Env:
    """Everything a leg needs: the package, one device context, the process group."""

    def __init__(self, args, zk, dist, rank, world, local_rank, dev):
        self.args, self.zk, self.dist = args, zk, dist
        self.rank, self.world, self.local_rank, self.dev = rank, world, local_rank, dev
        self.ctx = zk.Context(local_rank)
        self.force_dist = os.environ.get("ZKPOA_BENCH_FORCE_DIST") == "1"   # exercise the RCCL path with one rank
        self.multi = world > 1 or self.force_dist

    def sync(self):
        import torch
        torch.cuda.synchronize()
        if self.multi:
            self.dist.barrier(**self.args.barrier_kw)
        torch.cuda.synchronize()

    def max_over_ranks(self, seconds):
        import torch
        if not self.multi:
            return seconds
        t = torch.tensor([seconds], dtype=torch.float64, device=self.args.cdev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def agree(self, ok, what):
        """Every rank takes the same exit path: a failed local check fails the job everywhere (no rank is left
        waiting in a collective)."""
        import torch
        if self.multi:
            t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=self.args.cdev)
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MIN)
            ok = bool(int(t.item()))
        if not ok:
            raise BenchError(what)


# ---------------------------------------------------------------------------------------------------------------
def msm_leg(env, logn, steps, warmup, inflight, fixed_base=False, cpu_baseline=False):
    """G1 MSM over 2^logn points per rank, `inflight` MSMs kept in flight on separate lanes. -> line dict."""
    import numpy as np
    import torch
    from collections import deque
    from concurrent.futures import ThreadPoolExecutor
    from zkpoa_amd import sharding
    from zkpoa_amd.synthetic import dlog_sums
    zk, ctx, dist, args = env.zk, env.ctx, env.dist, env.args
    n_local = 1 << logn
    if logn > 27:
        raise SystemExit("msm workload limited to 2^27 points per GPU")

    # ---- synthetic inputs, resident in HBM: bases (a + i*b)*G with known discrete logs, uniform scalars
    seeds = random.Random(0x5EED0001)
    a, b = seeds.randrange(R_MOD), seeds.randrange(R_MOD)
    i0 = env.rank * n_local
    d_bases = torch.empty(n_local * 64, dtype=torch.uint8, device=env.dev)
    ctx.gen_bases_g1_device(a, b, i0, n_local, d_bases.data_ptr())
    limbs = np_scalars(n_local, 0x5EED0002 + env.rank)
    d_scalars = torch.from_numpy(limbs.view(np.uint8).reshape(-1)).to(env.dev)
    table = None
    if fixed_base:
        table = ctx.msm_table(1, d_bases.data_ptr(), n_local, int(os.environ.get("ZKPOA_MSM_C") or 0))

    # `inflight` MSMs are kept in flight, each on its own lane (HIP stream + workspace) driven by its own
    # host thread: the sort / bucket-reduction / read-back / host-Horner phases of one MSM are small or
    # latency-bound and overlap with the accumulation kernel of the next (the prover does the same with
    # its five MSMs). Collectives stay on the main thread, in step order.
    inflight = max(1, min(inflight, 12))
    pool = ThreadPoolExecutor(inflight)

    def msm_on(lane):
        if table is not None:
            part = ctx.msm_table_run(table, d_scalars.data_ptr(), lane)
        else:
            part = ctx.msm_g1_device_lane(lane, d_bases.data_ptr(), d_scalars.data_ptr(), n_local)
        return part, ctx.last_ms_lane(lane, 1), ctx.last_ms_lane(lane, 0)

    def combine(part):
        if not env.multi:
            return part
        return sharding.combine_partials(zk.g1_sum, sharding.all_gather_bytes(part, dist, args.cdev))

    def run(count):
        """-> (last result, sum of accumulate-kernel ms, sum of whole-MSM device ms)"""
        q = deque(pool.submit(msm_on, i % inflight) for i in range(min(inflight, count)))
        res, k_ms, d_ms = None, 0.0, 0.0
        for i in range(count):
            part, k1, k0 = q.popleft().result()
            if i + inflight < count:
                q.append(pool.submit(msm_on, i % inflight))      # lane i % inflight is free again: refill it
            res = combine(part)                                  # ... before the (blocking) collective
            k_ms += k1     # HIP events on the MSM's own stream, inside the library
            d_ms += k0
        return res, k_ms, d_ms

    try:
        run(max(warmup, inflight))      # also sizes every lane's workspace outside the timed region
        env.sync()
        t0 = time.perf_counter()
        result, kernel_ms, msm_dev_ms = run(steps)
        env.sync()
        elapsed = env.max_over_ranks(time.perf_counter() - t0)
        # the dominant kernel ALONE (nothing else in flight on the chip): three more MSMs, one at a time, after the
        # timed region; the roofline fractions below use this duration, never an overlapped one
        solo_k, solo_d = 0.0, 0.0
        for _ in range(3):
            _, k1, k0 = msm_on(0)
            solo_k += k1 / 3
            solo_d += k0 / 3
    finally:
        pool.shutdown()

    # ---- correctness of what was timed (outside the timed region): known-dlog check
    s0, s1 = dlog_sums(limbs, i0)
    d_loc = (a * s0 + b * s1) % R_MOD
    if env.multi:
        parts = [None] * env.world
        dist.all_gather_object(parts, d_loc)
        d_all = sum(parts) % R_MOD
    else:
        d_all = d_loc
    env.agree(result == oracle_g1(d_all), "MSM result failed the known-dlog check")

    line = None
    if env.rank == 0:
        if table is not None:
            _, c_win, windows, table_bytes = table.info()
        else:
            c_win = int(os.environ.get("ZKPOA_MSM_C") or classic_window(n_local))
            windows, table_bytes = (254 + c_win - 1) // c_win, 0
        adds = n_local * windows
        achieved = G1_MSM_BYTES_PER_POINT * n_local / (solo_k * 1e-3) / 1e9
        gadds = adds / (solo_k * 1e-3) / 1e9
        wl = "msm_g1_2p%d" % logn
        traffic, tsrc = pmc_traffic(wl + ("_fixed_base" if fixed_base else ""), "msm_accum0_kernel")
        line = {
            "metric": "G1-MSM throughput", "value": n_local * env.world * steps / elapsed, "unit": "pts/s",
            "n_gpus": env.world, "steps": steps, "warmup": warmup,
            "ms_per_step": elapsed / steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": DTYPE,
            "data": "synthetic" + (" (REHEARSAL: ranks share a GPU, gloo collectives; not a measurement)"
                                   if args.rehearse else ""),
            "config": {"workload": "BN254 G1 Pippenger MSM, 2^%d points per GPU, uniform 252-bit scalars, bases "
                                   "(a+i*b)*G resident in HBM%s" % (logn, " (BASELINE.json configs[1])" if logn == 20 else
                                                                   " (BASELINE.json configs[4] MSM part)" if logn == 26 else ""),
                       "points_per_gpu": n_local, "sharding": "index ranges, all-gather of partial points",
                       "msms_in_flight": inflight,
                       "form": ("fixed-base: 2^(c*j)*P_i precomputed once per base array (%.2f GB table), all windows "
                                "in one bucket set" % (table_bytes / 1e9)) if fixed_base else
                               "classic: arbitrary bases, nothing precomputed",
                       "checked": "known discrete log of the result (O(n) field arithmetic; the expected point from the "
                                  "oracle's scalar multiplication)"},
            "roofline": {"bound": "hbm", "kernel": "msm_accum0_kernel<Fq> (bucket accumulation)",
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": tsrc,
                         "kernel_ms": solo_k, "kernel_ms_overlapped": kernel_ms / steps,
                         "msm_device_ms_solo": solo_d, "msm_device_ms_overlapped": msm_dev_ms / steps,
                         "valu": {"unit": "G mixed additions/s", "achieved": gadds, "peak": VALU_PEAK_GADDS,
                                  "frac": gadds / VALU_PEAK_GADDS, "additions_per_launch": adds, "window_bits": c_win,
                                  "windows": windows},
                         "note": "achieved = 96 B/point x points per launch / kernel_ms, kernel_ms = the accumulation "
                                 "kernel alone on the chip (HIP events on its stream, 3 launches after the timed "
                                 "region; kernel_ms <= ms_per_step). The kernel is integer-VALU-bound (v_mad_u64_u32), "
                                 "not HBM-bound: `valu` is the bound that binds (DESIGN.md section 4)"},
        }
        if cpu_baseline:
            line["cpu_baseline"] = cpu_msm_baseline(env, d_bases, limbs, logn)
    if table is not None:
        table.close()
    del d_bases, d_scalars
    torch.cuda.empty_cache()
    return line


def cpu_msm_baseline(env, d_bases, limbs, logn):
    """The C oracle's Pippenger (oracle/c, -O3, threaded over (window, point-chunk) tasks) on ALL host cores, on a
    bounded sample of the same workload; checked against the GPU on that sample."""
    from oracle import c_oracle as co
    cores = host_cores()
    sample_log = min(logn, 20)
    ns = 1 << sample_log
    hb = bytes(d_bases[:ns * 64].cpu().numpy())
    hs = limbs[:ns].tobytes()
    reps = 3
    tc = time.perf_counter()
    for _ in range(reps):
        ref = co.msm_g1(hb, hs, ns, cores)
    tcpu = (time.perf_counter() - tc) / reps
    import torch
    d_s = torch.frombuffer(bytearray(hs), dtype=torch.uint8).to(env.dev)
    chk = env.ctx.msm_g1_device(d_bases.data_ptr(), d_s.data_ptr(), ns)
    if chk != ref:
        raise BenchError("GPU and CPU-oracle MSM disagree on the baseline sample")
    return {"value": ns / tcpu, "unit": "pts/s", "cores": cores, "nproc": box_cores(), "kind": "port",
            "sample": "first 2^%d points of the same workload, mean of %d MSMs; C oracle (oracle/c: plain-C "
                      "Pippenger, unsigned windows, Jacobian, u128 Montgomery, gcc -O3 -march=x86-64-v3 -madx), "
                      "%d threads = every core of this job's share of the GPU box (the box has nproc = %d logical CPUs); "
                      "a stand-in for rapidsnark (absent here), not a tuned CPU prover" % (sample_log, reps, cores, box_cores()),
            "seconds": tcpu * reps}


# ---------------------------------------------------------------------------------------------------------------
def gather_h_scalars(env, circ, split):
    """The complete H-scalar vector (numpy uint64 [n, 4]) of the last proof, on rank 0 (None elsewhere when N > 1).
    N = 1: read back from HBM. N > 1: every rank contributes what ITS H MSM consumed -- with the split chain its cyclic
    shard (odd-coset indices i = rank mod N, in order of i div N), with the replicated chain its index range -- and
    rank 0 puts the vector together."""
    import numpy as np
    import torch
    from zkpoa_amd.sharding import shard_range
    n, world, rank = circ.n, env.world, env.rank
    if world == 1 and not env.multi:      # (ZKPOA_BENCH_FORCE_DIST=1: one rank still goes through the collective)
        return circ.h_scalars()
    rows = -(-n // world)
    mine = np.zeros((rows, 4), dtype=np.uint64)
    if split:
        mine[:n // world] = circ.ctx.read_h_scalars(circ.key, n // world)
    else:
        lo, hi = shard_range(n, rank, world)
        mine[:hi - lo] = circ.ctx.read_h_scalars(circ.key, n)[lo:hi]
    t = torch.from_numpy(mine.view(np.uint8).reshape(-1)).to(env.args.cdev)
    if t.is_cuda:
        out = torch.empty(world * t.numel(), dtype=torch.uint8, device=t.device)
        env.dist.all_gather_into_tensor(out, t)
        parts = out.cpu().numpy().view(np.uint64).reshape(world, rows, 4) if rank == 0 else None
    else:
        outs = [torch.empty_like(t) for _ in range(world)]
        env.dist.all_gather(outs, t)
        parts = np.stack([o.numpy().view(np.uint64).reshape(rows, 4) for o in outs]) if rank == 0 else None
    if rank != 0:
        return None
    P = np.empty((n, 4), dtype=np.uint64)
    for g in range(world):
        if split:
            P[g::world] = parts[g][:n // world]
        else:
            lo, hi = shard_range(n, g, world)
            P[lo:hi] = parts[g][:hi - lo]
    return P


def probe_reference_provers(zkey, wtns):
    """if the box happens to have the reference's own provers -- `snarkjs` (with node >= 16) or
    a rapidsnark `prover` binary (on PATH under another directory than ours, or named by $RAPIDSNARK_REF_PATH) -- time
    them on the same key and witness. Never installs anything; on the build pool neither exists and this reports so."""
    import shutil
    import subprocess
    import tempfile
    ours = os.path.realpath(os.path.join(ROOT, "zk-proof-of-assets_amd", "prover"))
    cands = {}
    sj = shutil.which("snarkjs")
    node = shutil.which("node")
    if sj and node:
        try:
            major = int(subprocess.run([node, "--version"], capture_output=True, text=True, timeout=10).stdout.strip().lstrip("v").split(".")[0])
        except (ValueError, OSError, subprocess.SubprocessError):
            major = 0
        if major >= 16:
            cands["snarkjs"] = [sj, "groth16", "prove"]
    rs = os.environ.get("RAPIDSNARK_REF_PATH") or shutil.which("prover")
    if rs and os.path.realpath(rs) != ours and os.access(rs, os.X_OK):
        cands["rapidsnark"] = [rs]
    out = {"probed": "snarkjs + node >= 16 on PATH; a rapidsnark `prover` on PATH or $RAPIDSNARK_REF_PATH", "found": sorted(cands)}
    if not cands:
        return out
    with tempfile.TemporaryDirectory() as d:
        zp, wp = os.path.join(d, "c.zkey"), os.path.join(d, "w.wtns")
        with open(zp, "wb") as f:
            f.write(zkey)
        with open(wp, "wb") as f:
            f.write(wtns)
        for name, argv in cands.items():
            t0 = time.perf_counter()
            try:
                rc = subprocess.run(argv + [zp, wp, os.path.join(d, name + "_proof.json"), os.path.join(d, name + "_public.json")],
                                    capture_output=True, text=True, timeout=600)
                out[name] = {"seconds": time.perf_counter() - t0, "rc": rc.returncode}
            except (OSError, subprocess.SubprocessError) as e:
                out[name] = {"error": str(e)[:200]}
    return out


def cpu_prove_baseline(env, circ, gpu_points):
    """The same proof on the host: the synthetic key's sections are copied out of HBM into a .zkey image and the C
    oracle (oracle/c: orc_prove -- buildABC, 6 NTTs, joinABC, the five Pippenger MSMs, assembly) proves it on the
    job's cores with r = s = 0. Its 256 proof bytes must equal the GPU's: a whole-proof oracle parity at this shape."""
    from oracle import c_oracle as co
    cores = host_cores()
    npub = circ.n_public
    zkey, wtns = circ.zkey_image(), circ.wtns_image()
    tc = time.perf_counter()
    ref, _ = co.prove(zkey, wtns, 0, 0, cores, n_public=npub)
    tcpu = time.perf_counter() - tc
    if ref != gpu_points:
        raise BenchError("GPU proof and C-oracle proof differ (r = s = 0)")
    probe = probe_reference_provers(zkey, wtns)
    return {"reference_provers": probe, "value": 1.0 / tcpu, "unit": "proofs/s", "cores": cores, "nproc": box_cores(), "kind": "port",
            "sample": "ONE complete proof of the same key and witness (%.2f GB zkey image copied out of HBM), r = s = 0; C "
                      "oracle orc_prove (oracle/c: single-threaded buildABC + NTT chain, then five Pippenger MSMs threaded "
                      "over (window, chunk) tasks) on %d threads = this job's share of the box (nproc = %d); proof bytes equal "
                      "to the GPU's. Reference log, other hardware: rapidsnark 2.56 s at the 1-sig layer-one shape (same 2^21 domain) on ~22 busy x86 cores "
                      "(tests/1_sigs_1_batches_5_height/logs/layers_one_two_prove_batch_0.log:16-18)"
                      % (len(zkey) / 1e9, cores, box_cores()),
            "seconds": tcpu}


def prove_leg(env, k, steps, warmup, precompute=True, cpu_baseline=False, serial=False):
    """Full Groth16 prove against a key + witness resident in HBM. N>1: the proof's five MSMs are sharded
    over the ranks (BASELINE.json configs[3..4]); the H-scalar chain is split or replicated."""
    import torch
    from zkpoa_amd import sharding
    from zkpoa_amd.synthetic import SyntheticCircuit
    zk, ctx, dist, args = env.zk, env.ctx, env.dist, env.args
    spec = k
    k, m, n_pub, what = prove_shape(spec)
    world, rank = env.world, env.rank
    # N > 1: ONE proof sharded over the N GPUs (strong scaling): same circuit on every rank, each rank
    # owns index range rank/N of the five MSMs; partial points are all-gathered over RCCL and summed.
    # Each rank generates ONLY its own part of the key (its index ranges of the point sections, with the split chain
    # its cyclic H shard and its constraints' records: zkpoa_zkey_load_device_shard) and, like the single-GPU key,
    # gets the fixed-base tables of what it holds -- 1/N of the memory.
    split = world > 1 and not args.replicated_chain and sharding.split_chain_supported(world, 1 << k)
    # sections 5-8 are dealt out block-cyclically (blocks of 2^16 wires; fewer for the test sizes so that every rank
    # still holds several): a real witness clusters, equal index ranges would be unequal work
    block_log = 16
    while block_log > 4 and (m >> block_log) < 8 * world:
        block_log -= 1
    circ = SyntheticCircuit(zk, ctx, k, m, n_public=n_pub, seed=0x5EED0010, witness_like=True,
                            shard=(rank, world, split, block_log) if world > 1 else None)
    header = circ.key.header()
    xbufs = sharding.exchange_buffers(1 << k, world, env.dev) if split else None
    table_bytes = 0

    def one_proof():
        if world == 1:
            return circ.prove(0, 0)[0]
        if split:
            return sharding.sharded_prove_split(ctx, circ.key, circ.d_witness.data_ptr(), header, zk.sum_partials,
                                                zk.prove_assemble, 0, 0, dist, args.cdev, xbufs)
        return sharding.sharded_prove(lambda: ctx.prove_partials_device(circ.key, circ.d_witness.data_ptr()),
                                      header, zk.sum_partials, zk.prove_assemble, 0, 0, dist, args.cdev)

    names = ("H", "A", "B1", "B2", "C")
    proofs_run = 0
    try:
        if serial:      # profiling runs (rocprofv3 --pmc): one stage at a time, so per-kernel counters are solo values
            ctx.set_option("prove_serial", 1)
        if precompute:
            # fixed-base tables, once, outside the timed region -- after one proof, as the prover's own key cache does
            # (second use of a key): the A / B / C tables are then sized for the digit density of a real witness
            one_proof()
            table_bytes = circ.key.precompute()
            proofs_run += 1
        for _ in range(warmup):
            one_proof()
        proofs_run += warmup + steps
        env.sync()
        t0 = time.perf_counter()
        acc = {"h_chain": 0.0, "msm_phase": 0.0, "prove": 0.0, "h_msm_accum": 0.0}
        acc.update({"msm_%s" % x: 0.0 for x in names})
        for i in range(steps):
            pts = one_proof()
            acc["h_chain"] += ctx.last_ms(3)
            acc["msm_phase"] += ctx.last_ms(4)
            acc["prove"] += ctx.last_ms(5)
            acc["h_msm_accum"] += ctx.last_ms(1)
            for lane, x in enumerate(names):                 # device time of each MSM on its own lane (they overlap)
                acc["msm_%s" % x] += ctx.last_ms_lane(lane, 0)
        env.sync()
        elapsed = env.max_over_ranks(time.perf_counter() - t0)

        # ---- every stage ALONE (nothing else on the chip): two proofs with the stages serialised. Their sum against
        # the overlapped wall time says how much of a proof is arithmetic and how much is scheduling.
        solo = None
        if world == 1:
            ctx.set_option("prove_serial", 1)
            proofs_run += 2
            try:
                solo = {"h_chain": 0.0}
                solo.update({"msm_%s" % x: 0.0 for x in names})
                accum = {x: [0.0, 0.0] for x in names}    # accumulation kernel alone: ms, millions of mixed additions
                for _ in range(2):
                    if circ.prove(0, 0)[0] != pts:
                        raise BenchError("serialised prove differs from the overlapped one")
                    solo["h_chain"] += ctx.last_ms(3) / 2
                    for lane, x in enumerate(names):
                        solo["msm_%s" % x] += ctx.last_ms_lane(lane, 0) / 2
                        accum[x][0] += ctx.last_ms_lane(lane, 1) / 2
                        accum[x][1] += ctx.last_ms_lane(lane, 2) / 2
            finally:
                ctx.set_option("prove_serial", 1 if serial else 0)

        # ---- correctness of what was timed: pi_a, pi_b AND pi_c against the known-dlog expectation. The H scalars
        # the GPU produced are read back and validated independently of any transform by the oracle's quotient
        # identity A(z)B(z) - C(z) = H(z)(z^n - 1) at a random point (oracle/c: orc_quotient_check).
        # N > 1: every rank's H scalars (its cyclic shard after the split chain, its index range of the replicated
        # chain -- what its H MSM actually consumed) are gathered and checked on rank 0, so pi_c is covered on the
        # real multi-process path too.
        from oracle import c_oracle as co
        P = gather_h_scalars(env, circ, split)
        ok, tq, threads = True, 0.0, min(32, host_cores())
        if rank == 0:
            zpt = random.Random(0xC0FFEE + k).randrange(R_MOD)
            tq = time.perf_counter()
            ok = co.quotient_check(circ.coeff_section(), circ.w_limbs, m, k, P, zpt, threads)
            tq = time.perf_counter() - tq
            ea, eb, ec = circ.expected_dlogs(0, 0, P)      # exponents in Fr; the group elements come from the oracle
            ok = ok and pts[0:64] == oracle_g1(ea) and pts[64:192] == oracle_g2(eb) and pts[192:256] == oracle_g1(ec)
        checked = ("pi_a, pi_b, pi_c by known discrete log (expected points from the oracle's scalar multiplication); the 2^%d H scalars%s by the oracle's quotient identity at a "
                   "random point (%.1f s on %d host threads)"
                   % (k, " (gathered from the %d ranks)" % world if world > 1 else "", tq, threads))
        env.agree(ok, "proof failed the known-dlog / quotient-identity check")
        cpu = cpu_prove_baseline(env, circ, pts) if (cpu_baseline and world == 1) else None
    except BaseException:
        circ.close()
        raise
    finally:
        if serial:
            ctx.set_option("prove_serial", 0)

    line = None
    if rank == 0:
        n = 1 << k
        ncoef = circ.n_coef
        # SURVEY.md 8d full-prove formula
        alg = 96 * (3 * m - n_pub - 1) + 160 * m + 96 * n + 6 * 64 * n + 76 * ncoef + 96 * n + 128 * n
        sec = elapsed / steps
        traffic, tsrc = pmc_traffic("prove_2p%s" % spec, "per_proof") if world == 1 else (None, None)
        roof = {"bound": "hbm", "kernel": "whole prove (5 MSMs + H chain)",
                "achieved": alg / sec / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": alg / sec / 1e9 / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": tsrc,
                "algorithmic_bytes": alg,
                "phase_ms_overlapped": {kk: v / steps for kk, v in acc.items()}}
        if solo is not None:
            # the accumulation kernels of the five MSMs against the integer pipe: G1 against the
            # 14.2 G additions/s its parts allow, the G2 one (B2) against 4.67 G/s (8 Fq2 products at 45.3 G/s + 2 Fq2
            # squares + 14 Fq additions, tools/microbench.hip) -- it runs at 2 waves per SIMD, where one wave's issue
            # rate, not the pipe, sets the pace
            roof["valu_accum"] = {
                x: {"kernel_ms_solo": accum[x][0], "mixed_additions_M": accum[x][1],
                    "achieved_Gadds": accum[x][1] / accum[x][0] if accum[x][0] > 0 else None,   # millions per ms = G/s
                    "peak_Gadds": G2_VALU_PEAK_GADDS if x == "B2" else VALU_PEAK_GADDS,
                    "frac": (accum[x][1] / accum[x][0] / (G2_VALU_PEAK_GADDS if x == "B2" else VALU_PEAK_GADDS))
                            if accum[x][0] > 0 else None}
                for x in names}
            tot = sum(solo.values())
            roof["valu"] = {"stage_ms_solo": solo, "sum_solo_ms": tot, "wall_ms": sec * 1e3, "ratio": tot / (sec * 1e3),
                            "note": "every stage timed alone on the chip (prove_serial) vs the overlapped proof: "
                                    "ratio > 1 = the lanes overlap that much work; the stages are integer-VALU-bound "
                                    "(MSMs) or VALU + HBM (NTT chain), see DESIGN.md section 4"}
        line = {
            "metric": "Groth16 proofs/sec", "value": steps / elapsed, "unit": "proofs/s",
            "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": sec * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": DTYPE, "data": "synthetic" + (" (REHEARSAL: ranks share a GPU, gloo "
                                                   "collectives; not a measurement)" if args.rehearse else ""),
            "config": {"workload": "full Groth16 prove, domain 2^%d, %d wires, %d public (%s); key and witness "
                                   "resident in HBM; witness-like scalar distribution" % (k, m, n_pub, what),
                       "n_coefs": ncoef, "proofs_run_in_process": proofs_run,
                       "fixed_base_tables_GB": table_bytes / 1e9,
                       "parallelism": ("one proof, five MSMs sharded over %d GPUs (sections 5-8 block-cyclic, blocks of 2^%d "
                                       "items; H cyclic or by range), all-gather of partial points; H-scalar chain %s"
                                       % (world, block_log, "split (four-step NTTs, 2 all-to-alls per "
                                       "proof, all three polynomials in each)" if split else "replicated")) if world > 1 else "single GPU",
                       "checked": checked},
            "roofline": roof,
        }
        if cpu is not None:
            line["cpu_baseline"] = cpu
    circ.close()
    del circ, xbufs
    torch.cuda.empty_cache()
    return line


# ---------------------------------------------------------------------------------------------------------------
MODMUL_PEAK_G = 140.0       # 254-bit Montgomery products/s, register-only loop on one MI355X (tools/microbench.hip, r03: a * a
                            # 139.8 G/s at 8 waves/SIMD, the dedicated square 170.2 G/s; r02 quoted 130 from 4 waves/SIMD)
# One hash as the kernel computes it (poseidon.hip), in multiply-accumulate pairs (v_mad_u64_u32 + v_addc): a product is
# 64 + 72 (the Montgomery reduction), a square 36 + 72, a row of the 3 x 3 MDS product ONE sum of three products under a
# single reduction (r04: Fr::dot3, 192 + 72 -- it was three products, 408), the 2 x 2 product of the form change two sums
# of two (200 each). Full round: 3 S-boxes (2 squares + 1 product) + 3 rows; sparse partial round: 1 S-box + 1 row + 2
# products; + 3 products of form changes. Expressed in products of 136 pairs: the VALU ceiling is MODMUL_PEAK_G / that.
POSEIDON_MAC_PAIRS = 8 * (3 * (2 * 108 + 136) + 3 * 264) + 2 * 200 + 57 * ((2 * 108 + 136) + 264 + 2 * 136) + 3 * 136
POSEIDON_MODMULS = POSEIDON_MAC_PAIRS / 136.0   # 486.8 (r03: 607 products, every one with its own reduction)


def merkle_leg(env, n_leaves, steps, warmup, cpu_baseline=True):
    """The anonymity-set Poseidon Merkle tree (scripts/merkle_tree.rs): n (address, balance) pairs
    resident in HBM -> all levels of the tree. One step = one complete tree."""
    import numpy as np
    import torch
    ctx = env.ctx
    nr = np.random.default_rng(0x5EED0020)
    a = np.zeros((n_leaves, 4), dtype=np.uint64)
    b = np.zeros((n_leaves, 4), dtype=np.uint64)
    a[:, 0] = nr.integers(0, 1 << 63, size=n_leaves, dtype=np.uint64)
    a[:, 1] = nr.integers(0, 1 << 63, size=n_leaves, dtype=np.uint64)
    a[:, 2] = nr.integers(0, 1 << 32, size=n_leaves, dtype=np.uint64)       # 160-bit addresses
    b[:, 0] = nr.integers(0, 1 << 63, size=n_leaves, dtype=np.uint64)
    b[:, 1] = nr.integers(0, 1 << 26, size=n_leaves, dtype=np.uint64)       # 90-bit balances (wei)
    da = torch.from_numpy(a.view(np.uint8).reshape(-1)).to(env.dev)
    db = torch.from_numpy(b.view(np.uint8).reshape(-1)).to(env.dev)
    k = max(0, (n_leaves - 1).bit_length())
    hashes = n_leaves + (1 << k) - 1

    def build():
        t = ctx.merkle_build(da.data_ptr(), db.data_ptr(), device=True, n=n_leaves)
        ms = ctx.last_ms(7)
        return t, ms

    for _ in range(warmup):
        build()[0].close()
    env.sync()
    t0 = time.perf_counter()
    dev_ms = 0.0
    tree = None
    for _ in range(steps):
        if tree is not None:
            tree.close()
        tree, ms = build()
        dev_ms += ms
    env.sync()
    elapsed = time.perf_counter() - t0
    # correctness of what was timed: the root folds from a sampled leaf through its path with the CPU oracle's hash,
    # and a 2^12-leaf prefix tree equals the oracle's tree
    from oracle import c_oracle as co
    root = tree.root()
    ok = True
    for idx in (0, n_leaves - 1, n_leaves // 3):
        elems, bits = tree.path(idx)
        node = co.poseidon2(a[idx].tobytes(), b[idx].tobytes(), 1)
        for e, bit in zip(elems, bits):
            eb = int(e).to_bytes(32, "little")
            node = co.poseidon2(eb, node, 1) if bit else co.poseidon2(node, eb, 1)
        ok = ok and int.from_bytes(node, "little") == root
    tree.close()
    env.agree(ok, "Merkle root does not fold from sampled leaves with the oracle's Poseidon")
    sec = elapsed / steps
    kernel_s = dev_ms / steps * 1e-3
    line = {
        "metric": "Poseidon Merkle tree build", "value": n_leaves / sec, "unit": "leaves/s",
        "n_gpus": env.world, "steps": steps, "warmup": warmup, "ms_per_step": sec * 1e3, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": DTYPE, "data": "synthetic",
        "config": {"workload": "Poseidon(2) Merkle tree of %d (address, balance) leaves padded to 2^%d (scripts/merkle_tree.rs; "
                               "its header quotes 2.5 hours for a 10 M set on the CPU), inputs resident in HBM" % (n_leaves, k),
                   "hashes_per_tree": hashes,
                   "checked": "root folded from 3 sampled leaves through their paths with the CPU oracle's Poseidon"},
        "roofline": {"bound": "hbm", "kernel": "poseidon2_kernel (all levels of one tree)",
                     "achieved": 96.0 * hashes / kernel_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": 96.0 * hashes / kernel_s / 1e9 / HBM_PEAK_GBS, "traffic": None, "kernel_ms": kernel_s * 1e3,
                     "valu": {"unit": "G modmul/s", "achieved": POSEIDON_MODMULS * hashes / kernel_s / 1e9,
                              "peak": MODMUL_PEAK_G, "frac": POSEIDON_MODMULS * hashes / kernel_s / 1e9 / MODMUL_PEAK_G,
                              "hashes_per_s": hashes / kernel_s, "modmul_per_hash": POSEIDON_MODMULS},
                     "note": "algorithmic bytes = 64 B in + 32 B out per hash; the hash is %d multiply-accumulate pairs = %.1f "
                             "Montgomery products' worth (sums of products share one reduction): integer-VALU-bound"
                             % (POSEIDON_MAC_PAIRS, POSEIDON_MODMULS)},
    }
    if cpu_baseline:
        cores = host_cores()
        ns = 1 << 16
        tc = time.perf_counter()
        co.poseidon2(a[:ns].tobytes(), b[:ns].tobytes(), cores)
        tcpu = time.perf_counter() - tc
        line["cpu_baseline"] = {"value": ns / tcpu, "unit": "hashes/s", "cores": cores, "kind": "port",
                                "sample": "2^16 leaf hashes of the same workload, C oracle (oracle/c) on every host core of "
                                          "the job; the reference's Rust binary is single-threaded", "seconds": tcpu}
    del da, db
    torch.cuda.empty_cache()
    return line


# ---------------------------------------------------------------------------------------------------------------
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default=None,
                    help="default: msm_g1_2p20 with --gpus 1 (BASELINE.json configs[1]); prove_2p26 with --gpus N > 1 "
                         "(configs[4]: ONE synthetic layer_one 2^26 proof sharded over the N GPUs, strong scaling)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-also", action="store_true", help="headline workload only (no 'also' legs)")
    ap.add_argument("--fixed-base", action="store_true", help="msm workloads: fixed-base form (precomputed table)")
    ap.add_argument("--no-precompute", action="store_true", help="prove workloads: no fixed-base tables")
    ap.add_argument("--replicated-chain", action="store_true",
                    help="prove workloads, N > 1: run the whole H-scalar chain on every rank instead of splitting it")
    ap.add_argument("--inflight", type=int, default=6, help="MSMs kept in flight on separate HIP streams (1..12)")
    ap.add_argument("--serial", action="store_true",
                    help="prove workloads: run the stages of every proof one after the other (profiling: per-kernel "
                         "counters are then solo values); never a throughput measurement")
    args = ap.parse_args()
    default_workload = args.workload is None
    # the two workloads of BASELINE.json's metric; ZKPOA_BENCH_SMALL=1 (tests only: the multi-process rehearsal on one
    # GPU) swaps in test-size stand-ins so that the default code path -- headline choice, `also`, the curve objects --
    # runs in seconds; never a measurement
    small = os.environ.get("ZKPOA_BENCH_SMALL") == "1"
    weak_wl, strong_wl = ("msm_g1_2p14", "prove_2p16") if small else ("msm_g1_2p20", "prove_2p26")
    weak_log = int(weak_wl[len("msm_g1_2p"):])
    if default_workload:
        args.workload = weak_wl if args.gpus == 1 else strong_wl

    import torch
    import torch.distributed as dist
    from __graft_entry__ import load_package
    zk = load_package()

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
    force_dist = os.environ.get("ZKPOA_BENCH_FORCE_DIST") == "1"
    if world > 1 or force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if rehearse:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    rc = 0
    env = None
    try:
        env = Env(args, zk, dist, rank, world, local_rank, dev)
        if os.environ.get("ZKPOA_MSM_C"):        # experiments only: force the Pippenger window width
            env.ctx.set_option("msm_c", int(os.environ["ZKPOA_MSM_C"]))
        if os.environ.get("ZKPOA_MSM_K0"):       # experiments only: force the level-0 piece length
            env.ctx.set_option("msm_k0", int(os.environ["ZKPOA_MSM_K0"]))
        if args.workload.startswith("prove_2p"):
            line = prove_leg(env, args.workload[len("prove_2p"):], args.steps, args.warmup,
                             precompute=not args.no_precompute, serial=args.serial,
                             cpu_baseline=(world == 1 and not args.no_cpu_baseline and args.workload == "prove_2p21"))
        elif args.workload.startswith("merkle_"):
            spec = args.workload[len("merkle_"):]            # merkle_10000000, merkle_1e7, merkle_10M, merkle_64k
            mult = {"k": 10**3, "K": 10**3, "M": 10**6}.get(spec[-1:], 1)
            line = merkle_leg(env, int(float(spec[:-1] if mult > 1 else spec) * mult), args.steps, args.warmup)
        elif args.workload.startswith("msm_g1_2p"):
            line = msm_leg(env, int(args.workload[len("msm_g1_2p"):]), args.steps, args.warmup, args.inflight,
                           fixed_base=args.fixed_base, cpu_baseline=(world == 1 and not args.no_cpu_baseline))
        else:
            raise SystemExit("unknown workload " + args.workload)
        # ---- the rest of BASELINE.json's metric, same process, N = 1 only (bounded: about two minutes in total)
        if world == 1 and not force_dist and not args.no_also and default_workload and small:
            leg = prove_leg(env, strong_wl[len("prove_2p"):], args.steps, args.warmup)
            entry = {"workload": strong_wl}
            entry.update({kk: leg[kk] for kk in ("metric", "value", "unit", "steps", "warmup", "ms_per_step", "config", "roofline")})
            line["also"] = [entry]
        elif world == 1 and not force_dist and not args.no_also and args.workload == "msm_g1_2p20" and not args.fixed_base:
            also = []
            for name, fn in (
                    ("msm_g1_2p20_fixed_base", lambda: msm_leg(env, 20, args.steps, args.warmup, args.inflight, fixed_base=True)),
                    ("msm_g1_2p26", lambda: msm_leg(env, 26, 5, 2, 3)),
                    ("msm_g1_2p26_fixed_base", lambda: msm_leg(env, 26, 5, 2, 3, fixed_base=True)),
                    ("prove_2p21", lambda: prove_leg(env, 21, 20, 3, cpu_baseline=not args.no_cpu_baseline)),
                    ("prove_2p25", lambda: prove_leg(env, 25, 4, 1)),
                    ("prove_2p26", lambda: prove_leg(env, 26, 3, 1)),
                    ("prove_2p26_l3", lambda: prove_leg(env, "26_l3", 3, 1)),
                    ("merkle_10M", lambda: merkle_leg(env, 10_000_000, 3, 1))):
                t0 = time.perf_counter()
                leg = fn()
                entry = {"workload": name}
                entry.update({kk: leg[kk] for kk in ("metric", "value", "unit", "steps", "warmup", "ms_per_step", "config",
                                                      "roofline", "cpu_baseline") if kk in leg})
                entry["leg_seconds"] = time.perf_counter() - t0
                also.append(entry)
            line["also"] = also
        # N > 1 with the driver's command line: the headline is the north-star's strong-scaling proof; the weak-scaling
        # MSM of BASELINE.json configs[1] (2^20 points per GPU, all-gather of the partial points) rides along
        if world > 1 and default_workload and not args.no_also:
            t0 = time.perf_counter()
            leg = msm_leg(env, weak_log, args.steps, args.warmup, args.inflight)
            if rank == 0:
                entry = {"workload": "%s (weak scaling, 2^%d points per GPU)" % (weak_wl, weak_log)}
                entry.update({kk: leg[kk] for kk in ("metric", "value", "unit", "steps", "warmup", "ms_per_step", "scaling",
                                                      "config", "roofline") if kk in leg})
                entry["leg_seconds"] = time.perf_counter() - t0
                line["also"] = [entry]
        if rank == 0 and default_workload:
            # The two curves of BASELINE.json's metric under the SAME keys at every N, so that a plot of `curve.value`
            # (or `curve_weak.value`) over N never mixes metrics: `value` is the headline of this N (N = 1: configs[1],
            # the 2^20 MSM; N > 1: the north-star's one 2^26 proof over N GPUs), the curves are N-independent.
            def curve_of(src, workload, scaling):
                if src is None:
                    return None
                return {"workload": workload, "value": src["value"], "unit": src["unit"], "scaling": scaling,
                        "ms_per_step": src["ms_per_step"], "n_gpus": world}
            legs = {e["workload"].split(" ")[0]: e for e in line.get("also", [])}
            strong = line if world > 1 else legs.get(strong_wl)
            weak = line if world == 1 else legs.get(weak_wl)
            line["curve"] = curve_of(strong, strong_wl, "strong")              # proofs/s, ONE 2^26 proof over N GPUs
            line["curve_weak"] = curve_of(weak, weak_wl, "weak")               # pts/s, 2^20 points per GPU
        if rank == 0:
            if world > 1 and default_workload:
                line["scaling_note"] = ("strong scaling of ONE synthetic layer_one 2^26 proof (BASELINE.json configs[4]); the N = 1 "
                                        "point of this curve is the `prove_2p26` entry under `also` of the N = 1 line (whose "
                                        "headline is configs[1], the 2^20 MSM); the weak-scaling MSM of every N is under `also` here. Both curves "
                                        "are also top-level objects with the same keys at every N: `curve` (strong, proofs/s) and "
                                        "`curve_weak` (weak, pts/s)")
            line["n_ranks_seen"] = dist.get_world_size() if dist.is_initialized() else 1
            if dist.is_initialized():
                line["collectives"] = dist.get_backend()
                if dist.get_backend() == "nccl":
                    try:
                        line["rccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version())
                    except Exception:
                        pass
            print(json.dumps(line), flush=True)
    except BenchError as e:
        sys.stderr.write("bench.py: %s\n" % e)
        rc = 1
    finally:
        # no barrier here: a rank that failed must not leave its peers waiting; tearing the group down is enough
        if dist.is_initialized():
            try:
                dist.destroy_process_group()
            except Exception:
                pass
        if env is not None:
            try:
                env.ctx.close()
            except Exception:
                pass
    sys.exit(rc)


if __name__ == "__main__":
    main()
