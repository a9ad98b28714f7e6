#!/bin/bash
# Runs on the GPU box (gpurun): collects the evidence committed under profiles/ for this round (ROUND=r02 ...).
# kernel stats and PMC counters are separate rocprofv3 runs (never combined), one PMC counter per pass.
set -u
ROUND=${ROUND:-r04}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/profiles_$ROUND
PART=${PART:-all}      # 1 = bench lines + kernel stats, 2 = PMC passes (a gpurun call is at most 20 minutes)
[ "$PART" != 2 ] && rm -rf $O
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py"
if [ "$PART" != 2 ]; then
$B > $O/bench_default.json.log 2>$O/bench_default.err; echo "bench default (headline + also legs) rc=$?"
$B --no-also --no-cpu-baseline --inflight 1 > $O/bench_msm_g1_2p20_inflight1.json.log 2>/dev/null
$B --no-also --no-cpu-baseline --fixed-base --inflight 1 > $O/bench_msm_g1_2p20_fixed_base_inflight1.json.log 2>/dev/null
$B --workload prove_2p25 --steps 5 --warmup 2 > $O/bench_prove_2p25.json.log 2>/dev/null
$B --workload prove_2p21 --steps 20 --warmup 3 --no-precompute > $O/bench_prove_2p21_no_tables.json.log 2>/dev/null
$B --workload prove_2p26 --steps 3 --warmup 1 --no-precompute > $O/bench_prove_2p26_no_tables.json.log 2>/dev/null
echo "bench lines done"
S="rocprofv3 --kernel-trace --stats --output-format csv"
$S -d $O/stats_default -- $B --no-also --no-cpu-baseline > $O/stats_default.log 2>&1; echo "stats default rc=$?"
$S -d $O/stats_inflight1 -- $B --no-also --no-cpu-baseline --inflight 1 > $O/stats_inflight1.log 2>&1
$S -d $O/stats_fixed_inflight1 -- $B --no-also --no-cpu-baseline --fixed-base --inflight 1 > $O/stats_fixed_inflight1.log 2>&1
$S -d $O/stats_msm26 -- $B --workload msm_g1_2p26 --steps 2 --warmup 1 --inflight 1 --no-cpu-baseline > $O/stats_msm26.log 2>&1
$S -d $O/stats_msm26_fixed -- $B --workload msm_g1_2p26 --fixed-base --steps 2 --warmup 1 --inflight 1 --no-cpu-baseline > $O/stats_msm26_fixed.log 2>&1
$S -d $O/stats_prove26 -- $B --workload prove_2p26 --steps 2 --warmup 1 > $O/stats_prove26.log 2>&1
$S -d $O/stats_prove21 -- $B --workload prove_2p21 --steps 10 --warmup 2 > $O/stats_prove21.log 2>&1
$S -d $O/stats_ntt -- python3 $R/tools/ntt_time.py > $O/stats_ntt.log 2>&1
$S -d $O/stats_prove25 -- $B --workload prove_2p25 --steps 3 --warmup 1 > $O/stats_prove25.log 2>&1
$S -d $O/stats_prove21_serial -- $B --workload prove_2p21 --steps 6 --warmup 1 --serial --no-cpu-baseline > $O/stats_prove21_serial.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $O/trace_serial21 -- python3 $R/tools/trace_serial_prove.py 21 > $O/trace_serial21.log 2>&1
python3 $R/tools/trace_timeline.py $O/trace_serial21 > $O/timeline_prove21_serial.txt 2>&1
echo "stats done"
fi
if [ "$PART" != 1 ]; then
for c in FETCH_SIZE WRITE_SIZE; do
  P="rocprofv3 --pmc $c --kernel-trace --output-format csv"
  $P -d $O/pmc_calib_$c -- $R/tools/gather_calib > $O/pmc_calib_$c.log 2>&1
  $P -d $O/pmc_msm20_$c -- $B --no-also --inflight 1 --steps 4 --warmup 1 --no-cpu-baseline > $O/pmc_msm20_$c.log 2>&1
  $P -d $O/pmc_msm20fb_$c -- $B --no-also --fixed-base --inflight 1 --steps 4 --warmup 1 --no-cpu-baseline > $O/pmc_msm20fb_$c.log 2>&1
  $P -d $O/pmc_msm26_$c -- $B --workload msm_g1_2p26 --inflight 1 --steps 1 --warmup 1 --no-cpu-baseline > $O/pmc_msm26_$c.log 2>&1
  $P -d $O/pmc_msm26fb_$c -- $B --workload msm_g1_2p26 --fixed-base --inflight 1 --steps 1 --warmup 1 --no-cpu-baseline > $O/pmc_msm26fb_$c.log 2>&1
  # the same 2^20 MSM with 128-entry pieces instead of 32: a quarter of the per-piece metadata gathers and of the
  # lane-strided walks over the sorted index array -- tells how much of FETCH beyond the 64-B bases those are
  ZKPOA_MSM_K0=128 $P -d $O/pmc_msm20k128_$c -- $B --no-also --inflight 1 --steps 4 --warmup 1 --no-cpu-baseline > $O/pmc_msm20k128_$c.log 2>&1
  # whole proofs, one stage at a time (--serial): every kernel's counters are solo values; bench_prove21_pmc.json.log
  # says how many proofs the process ran (config.proofs_run_in_process)
  $P -d $O/pmc_prove21_$c -- $B --workload prove_2p21 --steps 4 --warmup 1 --serial --no-cpu-baseline > $O/pmc_prove21_$c.log 2>&1
  $P -d $O/pmc_prove26_$c -- $B --workload prove_2p26 --steps 1 --warmup 0 --serial --no-cpu-baseline > $O/pmc_prove26_$c.log 2>&1
  $P -d $O/pmc_prove25_$c -- $B --workload prove_2p25 --steps 1 --warmup 0 --serial --no-cpu-baseline > $O/pmc_prove25_$c.log 2>&1
done
echo "pmc done"
fi
# keep the merged output small: drop the per-dispatch traces of the big runs
find $O -name '*kernel_trace.csv' -size +4M -delete
find $O -name '*counter_collection.csv' -size +8M -delete
ls $O
