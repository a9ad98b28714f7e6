#!/bin/bash
# Runs on the GPU box (gpurun): collects the evidence committed under profiles/ for this round.
# kernel stats and PMC counters are separate rocprofv3 runs (never combined), one PMC counter per pass.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/profiles_r01
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $O/bench_default.json.log 2>&1; echo "bench default rc=$?"
python3 $R/bench.py --inflight 1 --no-cpu-baseline > $O/bench_msm_g1_2p20_inflight1.json.log 2>&1
python3 $R/bench.py --workload msm_g1_2p26 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_msm_g1_2p26.json.log 2>&1
for k in 21 25 26; do python3 $R/bench.py --workload prove_2p$k --steps 5 --warmup 2 > $O/bench_prove_2p$k.json.log 2>&1; done
echo "bench lines done"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_default -- python3 $R/bench.py --no-cpu-baseline > $O/stats_default.log 2>&1; echo "stats default rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_inflight1 -- python3 $R/bench.py --inflight 1 --no-cpu-baseline > $O/stats_inflight1.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_msm26 -- python3 $R/bench.py --workload msm_g1_2p26 --steps 2 --warmup 1 --no-cpu-baseline > $O/stats_msm26.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_prove25 -- python3 $R/bench.py --workload prove_2p25 --steps 2 --warmup 1 > $O/stats_prove25.log 2>&1
echo "stats done"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_calib_$c -- $R/tools/gather_calib > $O/pmc_calib_$c.log 2>&1
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_msm20_$c -- python3 $R/bench.py --inflight 1 --steps 4 --warmup 1 --no-cpu-baseline > $O/pmc_msm20_$c.log 2>&1
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_msm26_$c -- python3 $R/bench.py --workload msm_g1_2p26 --inflight 1 --steps 1 --warmup 1 --no-cpu-baseline > $O/pmc_msm26_$c.log 2>&1
done
echo "pmc done"
# keep the merged output small: drop the per-dispatch traces of the big runs
find $O -name '*kernel_trace.csv' -size +4M -delete
ls $O
