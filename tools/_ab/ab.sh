set -e
cd zk-proof-of-assets_amd
cp libzkpoa_prover.so /tmp/new.so
run() { python ../bench.py --no-also --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d['roofline'].get('kernel_ms'))"; }
for i in 1 2 3; do
  cp ../tools/_ab/libzkpoa_prover_old.so libzkpoa_prover.so; run old
  cp /tmp/new.so libzkpoa_prover.so; run new
done
