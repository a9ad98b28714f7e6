#!/bin/bash
# r03 A/B on one GPU box: dedicated sqr / dot2 are compiled in; the scan and the NTT radix switch at run time.
# usage (gpurun): bash tools/ab_r03.sh   -> gpurun_out/r03_ab_*.txt
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
B="python3 $R/bench.py --no-cpu-baseline"
run() { name=$1; shift; timeout -k 10 300 env "$@" > $O/r03_ab_$name.json 2> $O/r03_ab_$name.err || echo "FAILED $name"; python3 $R/tools/show_bench.py $O/r03_ab_$name.json 2>/dev/null | head -8; }
echo "== ntt radix 4 (default)"; timeout -k 10 120 python3 $R/tools/ntt_time.py
echo "== ntt radix 2"; ZKPOA_NTT_RADIX=2 timeout -k 10 120 python3 $R/tools/ntt_time.py
echo "== msm 2^20 default";   run msm_default $B --no-also
echo "== msm 2^20 scan3";     run msm_scan3 ZKPOA_SCAN=3 $B --no-also
echo "== prove 2^21 default"; run p21_default $B --workload prove_2p21 --steps 20 --warmup 3
echo "== prove 2^21 scan3";   run p21_scan3 ZKPOA_SCAN=3 $B --workload prove_2p21 --steps 20 --warmup 3
echo "== prove 2^21 radix2";  run p21_radix2 ZKPOA_NTT_RADIX=2 $B --workload prove_2p21 --steps 20 --warmup 3
echo "== prove 2^26 default"; run p26_default $B --workload prove_2p26 --steps 3 --warmup 1
echo "== prove 2^26 radix2";  run p26_radix2 ZKPOA_NTT_RADIX=2 $B --workload prove_2p26 --steps 3 --warmup 1
