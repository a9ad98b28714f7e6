run() { python bench.py --no-also --no-cpu-baseline $2 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d['roofline'].get('kernel_ms'))"; }
for i in 1 2 3; do
  ZKPOA_ACCUM_GATE=0 run gate0
  ZKPOA_ACCUM_GATE=1 run gate1
done
ZKPOA_ACCUM_GATE=1 run gate1_fixed --fixed-base
ZKPOA_ACCUM_GATE=0 run gate0_fixed --fixed-base
ZKPOA_ACCUM_GATE=1 run gate1_inflight8 "--inflight 8"
ZKPOA_ACCUM_GATE=1 run gate1_inflight4 "--inflight 4"
ZKPOA_ACCUM_GATE=1 run gate1_inflight3 "--inflight 3"
for g in 0 1; do ZKPOA_ACCUM_GATE=$g python bench.py --no-also --workload prove_2p21 --steps 20 --warmup 3 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('prove21 gate$g', d['ms_per_step'])"; done
