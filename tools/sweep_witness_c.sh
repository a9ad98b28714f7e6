for c in 14 15 16 17 18 19 20; do
  ZKPOA_WITNESS_TABLE_C=$c python bench.py --no-also --workload prove_2p21 --steps 20 --warmup 3 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; s=r['valu']['stage_ms_solo']
print('c=$c wall %.2f ms  tables %.2f GB  solo: '%(d['ms_per_step'], d['config']['fixed_base_tables_GB'])+' '.join('%s %.2f'%(k,v) for k,v in s.items()))"
done
python bench.py --no-also --workload prove_2p21 --steps 20 --warmup 3 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; s=r['valu']['stage_ms_solo']
print('model wall %.2f ms  tables %.2f GB  solo: '%(d['ms_per_step'], d['config']['fixed_base_tables_GB'])+' '.join('%s %.2f'%(k,v) for k,v in s.items()))"
