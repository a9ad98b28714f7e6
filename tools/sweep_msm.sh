#!/bin/bash
# GPU box: sweep of window width / piece length for the 2^20 MSM, classic and fixed-base (experiments; results in gpurun_out/)
# usage: tools/sweep_msm.sh <logn> "<c list>" "<k0 list>" [classic|fixed|both]
LOGN=${1:-20}; CS=${2:-"0"}; KS=${3:-"0"}; FORMS=${4:-both}
OUT=gpurun_out/sweep_msm_2p${LOGN}.txt
: > $OUT
for form in classic fixed; do
  [ "$FORMS" != both ] && [ "$FORMS" != $form ] && continue
  for c in $CS; do for k in $KS; do
    flag=""; [ $form = fixed ] && flag="--fixed-base"
    for inf in 1 6; do
      line=$(ZKPOA_MSM_C=$( [ $c = 0 ] && echo "" || echo $c ) ZKPOA_MSM_K0=$( [ $k = 0 ] && echo "" || echo $k ) python bench.py --workload msm_g1_2p${LOGN} --no-also --no-cpu-baseline --inflight $inf --steps ${STEPS:-20} $flag 2>/dev/null | tail -1)
      echo "$form c=$c k0=$k inflight=$inf $(echo "$line" | python -c 'import json,sys; l=json.loads(sys.stdin.read()); r=l["roofline"]; print("ms/step %.3f  accum_solo %.3f  msm_solo %.3f  valu %.3f  W=%d" % (l["ms_per_step"], r["kernel_ms"], r["msm_device_ms_solo"], r["valu"]["frac"], r["valu"]["windows"]))')" | tee -a $OUT
    done
  done; done
done
