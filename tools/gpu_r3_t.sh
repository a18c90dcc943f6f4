#!/bin/bash
# round 3, call T: masked-sample sums of the integer-dosage Step-2 route on the matrix cores (16 variants per workgroup) against one
# workgroup per (variant, phenotype); the driver after its split into translation units
O=gpurun_out/r3t
mkdir -p $O
( time timeout 900 python -m pytest tests/test_step2_qt_gpu.py tests/test_step2_bt_gpu.py -x -q -m gpu ) > $O/pytest_s2.log 2>&1
grep "passed\|failed\|error" $O/pytest_s2.log | tail -3
grep -E "^E " $O/pytest_s2.log | head -10 | cut -c1-300
for v in 1 0; do
  ( RG_S2_MASKED_OLD=$v timeout 600 python tools/step2_record.py ) > $O/step2_record_old_$v.json 2> $O/step2_record_old_$v.err
  python -c "
import json
d=json.load(open('$O/step2_record_old_$v.json'))
print('RG_S2_MASKED_OLD=$v', d.get('error') or {k:(round(x['ms_per_block'],3), round(x['variants_per_s']/1e6,3), x['parity_vs_oracle']) for k,x in d['cases'].items()})"
done
( time timeout 1500 python -m pytest tests/test_cli_gpu.py tests/test_reference_gpu.py -x -q -m gpu ) > $O/pytest_driver.log 2>&1
grep "passed\|failed\|error" $O/pytest_driver.log | tail -3
