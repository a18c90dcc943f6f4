#!/bin/bash
# round 3, call F: prediction kernels with the new epilogue (ring vs staged at configs[2] in full), then the default bench line with its sub-records
O=gpurun_out/r3f
mkdir -p $O
( time timeout 900 python -m pytest tests/test_step1_gpu.py tests/test_reference_gpu.py -x -q -m gpu ) > $O/pytest.log 2>&1
grep "passed\|failed\|error" $O/pytest.log | tail -3
( time timeout 600 python bench.py --samples 500000 --snps 500000 --phenos 10 --steps 1 --warmup 1 --no-cpu --oracle-check ) > $O/config3_ring.log 2>&1
grep '^{' $O/config3_ring.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], {k:round(v['ms'],1) for k,v in d['kernels'].items()}, d['cpu_baseline']['full_config_vs_oracle'])"
( time RG_PRED_STAGED=1 timeout 600 python bench.py --samples 500000 --snps 500000 --phenos 10 --steps 1 --warmup 1 --no-cpu ) > $O/config3_staged.log 2>&1
grep '^{' $O/config3_staged.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], {k:round(v['ms'],1) for k,v in d['kernels'].items()})"
( time timeout 900 python bench.py ) > $O/bench_default.log 2>&1
tail -1 $O/bench_default.log | cut -c1-400
tail -1 $O/bench_default.log | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('e2e', d['end_to_end_from_files'] and d['end_to_end_from_files'].get('walls_s'))
print('config3', {k: d['config3_single_gpu'].get(k) for k in ('ms_per_step','value','error')})
s2=d['step2']; print('step2', s2.get('error') or {k:(round(v['ms_per_block'],3), round(v['variants_per_s']/1e6,2), v['parity_vs_oracle']) for k,v in s2['cases'].items()})
"
