run() { python bench.py --no-also --no-cpu-baseline $2 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d['roofline'].get('kernel_ms'))"; }
for i in 1 2 3; do
  run sdma_default
  HSA_ENABLE_SDMA=0 run sdma_off
done
