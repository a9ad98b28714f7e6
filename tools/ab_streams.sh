# GPU box: same-box A/B of the lanes' stream priorities (experiments)
run() {
  label=$1; shift
  for w in ${WORKLOADS:-prove_2p21 msm_g1_2p20}; do
    vals=""
    for rep in 1 2 3; do
      st=30; [ $w = prove_2p25 ] && st=5
      v=$(env "$@" python bench.py --no-also --no-cpu-baseline --workload $w --steps $st --warmup 3 2>/dev/null | python -c "import json,sys; l=json.loads(sys.stdin.read()); print(round(l['ms_per_step'],3))")
      vals="$vals $v"
    done
    echo "$label $w:$vals"
  done
}
run "default (ladder2, lanes 0,3,2,5,1,4)" A=1
run "ladder2, lanes in order             " ZKPOA_BENCH_LANES=0,1,2,3,4,5
run "ladder2, lanes 0,3,1,4,2,5 (H,H,L,L,M,M)" ZKPOA_BENCH_LANES=0,3,1,4,2,5
run "ladder,  lanes in order             " ZKPOA_LANE_PRIO=ladder ZKPOA_BENCH_LANES=0,1,2,3,4,5
run "flat,    lanes in order             " ZKPOA_LANE_PRIO=flat ZKPOA_BENCH_LANES=0,1,2,3,4,5
run "default (ladder2, lanes 0,3,2,5,1,4)" A=1
